#!/bin/bash
# Build librltime_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../librltime_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
"$HIPCC" $FLAGS -c "$HERE/replay.hip" -o "$HERE/replay.o"
"$HIPCC" $FLAGS -c "$HERE/qmath.hip" -o "$HERE/qmath.o"
"$HIPCC" $FLAGS -c "$HERE/lstm.hip" -o "$HERE/lstm.o"
"$HIPCC" $FLAGS -c "$HERE/convert.hip" -o "$HERE/convert.o"
"$HIPCC" $FLAGS -c "$HERE/nnops.hip" -o "$HERE/nnops.o"
"$HIPCC" $FLAGS -c "$HERE/acting.hip" -o "$HERE/acting.o"
"$HIPCC" $FLAGS -c "$HERE/conv_in.hip" -o "$HERE/conv_in.o"
"$HIPCC" $FLAGS -c "$HERE/conv_mid.hip" -o "$HERE/conv_mid.o"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$HERE/replay.o" "$HERE/qmath.o" "$HERE/lstm.o" "$HERE/convert.o" "$HERE/nnops.o" "$HERE/acting.o" "$HERE/conv_in.o" "$HERE/conv_mid.o" -o "$OUT"
# plain-C consumer of the C-ABI (gcc, no Python / torch): proves the boundary is self-contained
ROCM="${ROCM_PATH:-/opt/rocm}"
gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -I "$ROCM/include" -I "$HERE/../../include" "$HERE/../../examples/mirl_demo.c" \
  -L "$HERE/.." -lrltime_hip -L "$ROCM/lib" -lamdhip64 -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,"$ROCM/lib" -o "$HERE/mirl_demo"
echo "built $OUT"
