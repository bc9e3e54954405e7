"""warpedganspace_amd — MI355X-native WarpedGANSpace training inner loop (warp -> G -> R -> loss).

Host-side mirror of the reference's Python interface (lib/support_sets.py, lib/reconstructor.py,
lib/trainer.py, models/gan_load.py) on top of hand-written HIP kernels reached through the C ABI of
include/wgs.h (libwgs_hip.so).
"""
__version__ = "0.1.0"
