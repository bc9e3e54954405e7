#!/usr/bin/env bash
# Build development variants of the library with -DWGS_ABL=<n> for the split-bf16 conv kernel -> tools/_bin/libwgs_abl<n>.so
set -euo pipefail
cd "$(dirname "$0")/../warpedganspace_amd/csrc"
mkdir -p ../../tools/_bin
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DWGS_ABL=$n -c conv_igemm_bf16.hip -o /tmp/abl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls build/*.o | grep -v conv_igemm_bf16)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_bin/libwgs_abl$n.so $objs /tmp/abl$n.o
done
