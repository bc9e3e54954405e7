#!/usr/bin/env bash
# Build libwgs_hip.so for gfx950 (MI355X) in-tree.  No torch headers, no pybind: a plain C-ABI library.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
OBJS=()
for src in *.hip; do
  obj="build/${src%.hip}.o"
  mkdir -p build
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ wgs_common.h -nt "$obj" ] || [ conv_args.h -nt "$obj" ] || [ fir4.h -nt "$obj" ] || [ conv_epilogue.h -nt "$obj" ] || [ conv_scheme.h -nt "$obj" ] || [ conv_nt_kernel.inc -nt "$obj" ] || [ conv_nt_launch.inc -nt "$obj" ] || [ ../../include/wgs.h -nt "$obj" ] \
     || { [ -f "${src%.hip}.inc" ] && [ "${src%.hip}.inc" -nt "$obj" ]; }; then
    echo "[hipcc] $src"
    $HIPCC $FLAGS -c "$src" -o "$obj" &
  fi
  OBJS+=("$obj")
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libwgs_hip.so "${OBJS[@]}"
# fingerprint of the sources this library was built from (checked by __graft_entry__.build() and warpedganspace_amd._lib: a prebuilt
# .so that travels with a snapshot must match the sources beside it)
LC_ALL=C; export LC_ALL; cat *.hip *.h *.inc ../../include/wgs.h | sha256sum | cut -d' ' -f1 > ../libwgs_hip.so.sources.sha256
echo "built $(cd .. && pwd)/libwgs_hip.so"
