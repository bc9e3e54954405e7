#!/usr/bin/env bash
# Build development variants of the library -> tools/_bin/libwgs_<tag><n>.so
#   tools/build_abl.sh abl 3 4      : -DWGS_ABL=<n>  on conv_igemm_bf16.hip  (register-staged kernel ablations)
#   tools/build_abl.sh pabl 1 2 3   : -DWGS_PABL=<n> on conv_igemm_patch.hip (patch kernel ablations)
#   tools/build_abl.sh uabl 1 2 3   : -DWGS_UABL=<n> on conv_upfused.hip     (fused up-conv kernel ablations)
#   tools/build_abl.sh qabl 1 2 4   : -DWGS_QABL=<n> on conv_patch_dma.hip   (all-DMA patch kernel ablations)
#   tools/build_abl.sh w16abl 1 2 : -DWGS_W16ABL=<n> on conv_wino_bf16.hip (split-bf16 F(2,3) kernel ablations)
#   tools/build_abl.sh dabl 1       : -DWGS_DABL=<n> on conv_igemm_dma.hip   (two-stage pipeline with full drains)
set -euo pipefail
cd "$(dirname "$0")/../warpedganspace_amd/csrc"
mkdir -p ../../tools/_bin
tag=$1; shift
if [ "$tag" = pabl ]; then src=conv_igemm_patch; def=WGS_PABL; elif [ "$tag" = qabl ]; then src=conv_patch_dma; def=WGS_QABL; elif [ "$tag" = dabl ]; then src=conv_igemm_dma; def=WGS_DABL; elif [ "$tag" = w16abl ]; then src=conv_wino_bf16; def=WGS_W16ABL; elif [ "$tag" = uabl ]; then src=conv_upfused; def=WGS_UABL; else src=conv_igemm_bf16; def=WGS_ABL; fi
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -D$def=$n -c $src.hip -o /tmp/$tag$n.o &
done
wait
for n in "$@"; do
  objs=$(ls build/*.o | grep -v "build/$src.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_bin/libwgs_$tag$n.so $objs /tmp/$tag$n.o
done
