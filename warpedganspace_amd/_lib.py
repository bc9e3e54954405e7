"""ctypes binding of libwgs_hip.so (the C ABI declared in include/wgs.h).

The product path has NO CPU / PyTorch fallback: if the HIP library is missing, or a tensor is not
a contiguous fp32 device tensor, the call raises.  PyTorch only provides device memory and streams.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WGS_LIB", os.path.join(_HERE, "libwgs_hip.so"))   # WGS_LIB: development override
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "wgs.h")

_lib = None


class WgsError(RuntimeError):
    pass


def header_symbols(header_path=HEADER_PATH):
    """Names of every function declared in include/wgs.h (used by the CPU export test)."""
    src = open(header_path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wgs_[a-z0-9_]+)\s*\(", src)))


def source_fingerprint():
    """sha256 over csrc/*.hip, *.h, *.inc and include/wgs.h in the order build.sh hashes them (shell glob order = sorted names)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for pat in ("*.hip", "*.h", "*.inc"):
        for f in sorted(glob.glob(os.path.join(csrc, pat))):
            h.update(open(f, "rb").read())
    h.update(open(HEADER_PATH, "rb").read())
    return h.hexdigest()


def library_matches_sources():
    """True / False: libwgs_hip.so was built from the sources beside it (fingerprint written by csrc/build.sh); None: no fingerprint."""
    fp = LIB_PATH + ".sources.sha256"
    if not os.path.isfile(fp):
        return None
    return open(fp).read().strip() == source_fingerprint()


def lib():
    """Load (once) and return the ctypes handle; raises WgsError when the extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise WgsError(
                "libwgs_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or warpedganspace_amd/csrc/build.sh. There is no CPU fallback." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.wgs_last_error.restype = ctypes.c_char_p
        _lib.wgs_rbf_ws_floats.restype = ctypes.c_int64
        _lib.wgs_dev_launch_count.restype = ctypes.c_int64
        _lib.wgs_dev_last_kernel.restype = ctypes.c_char_p
        for name in ("wgs_conv_ws_bytes",):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = ctypes.c_int64
    return _lib


def check(rc, what="wgs"):
    if rc != 0:
        raise WgsError("%s failed (rc=%d): %s" % (what, rc, lib().wgs_last_error().decode()))


_RAW_STREAM = None if os.environ.get("WGS_TORCH_STREAM") else getattr(torch._C, "_cuda_getCurrentRawStream", None)    # WGS_TORCH_STREAM=1: development A/B
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream_id(device_index=None):
    """hipStream_t (as an integer) of torch's current stream on the device (default: the current device).  Through torch's raw
    accessors where this build has them: torch.cuda.current_stream() costs ~7 us per call in device-index resolution, and a training
    step asks ~280 times (tools/host_cprofile.py)."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return _RAW_STREAM(_RAW_DEVICE() if device_index is None else device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def stream():
    return ctypes.c_void_p(stream_id())


class StagedPass:
    """A generator pass enqueued in stages (a Python generator that yields at its pause points).  The pass captured its launch stream
    when it was created, so every stage must be resumed under THAT stream: advance() checks it instead of silently enqueueing one
    stage's launches on another stream than the tensors they read were produced on."""

    def __init__(self, gen):
        self.stream_id = stream_id()
        self.gen = gen
        next(gen)                      # the first stage

    def advance(self):
        """Enqueue the next stage: None while the pass is paused again, the image when it is complete."""
        if stream_id() != self.stream_id:
            raise WgsError("a staged generator pass must be resumed under the stream it was begun on (begun on %#x, resumed on %#x)"
                           % (self.stream_id, stream_id()))
        try:
            next(self.gen)
        except StopIteration as e:
            return e.value[0]
        return None

    def finish(self):
        """Enqueue everything that is left: the image."""
        while True:
            img = self.advance()
            if img is not None:
                return img


def ptr(t, dtype=torch.float32, name="tensor"):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise WgsError("%s must be a GPU tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise WgsError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise WgsError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def rawptr(t):
    """Device pointer without layout checks (caller guarantees the memory layout)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


c_int = ctypes.c_int
c_float = ctypes.c_float
c_int64 = ctypes.c_int64
