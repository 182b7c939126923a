#!/bin/bash
# Build librltime_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# Objects are rebuilt only when their source (or any header) is newer; the
# translation units compile in parallel.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../librltime_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function ${MIRL_EXTRA_HIPCC_FLAGS:-}"
UNITS="replay qmath lstm lstm_seq convert nnops acting actnet conv_in conv_mid gemm3 conv3 conv_wrw optim"
newest_hdr=$(ls -t "$HERE"/*.h "$HERE"/*.hpp "$HERE"/../../include/*.h "$HERE/build.sh" | head -1)
pids=()
objs=()
for u in $UNITS; do
  [ -f "$HERE/$u.hip" ] || continue
  objs+=("$HERE/$u.o")
  if [ ! -f "$HERE/$u.o" ] || [ "$HERE/$u.hip" -nt "$HERE/$u.o" ] || [ "$newest_hdr" -nt "$HERE/$u.o" ]; then
    "$HIPCC" $FLAGS -c "$HERE/$u.hip" -o "$HERE/$u.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
# plain-C consumer of the C-ABI (gcc, no Python / torch): proves the boundary is self-contained
ROCM="${ROCM_PATH:-/opt/rocm}"
gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -I "$ROCM/include" -I "$HERE/../../include" "$HERE/../../examples/mirl_demo.c" \
  -L "$HERE/.." -lrltime_hip -L "$ROCM/lib" -lamdhip64 -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,"$ROCM/lib" -o "$HERE/mirl_demo"
echo "built $OUT"
