#!/bin/bash
# Builds libldpc_erasure_amd.so for gfx950 (cross-compiles without a GPU).  Called by __graft_entry__.build().
# Every translation unit is compiled to its own object (in parallel, and only when a source it depends on is newer), then linked:
# kernels.hip -- the one big device TU -- takes minutes, the host TUs seconds.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
python3 "$ROOT/tools/gen_builtin_codes.py" > /dev/null
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -Wall -Wno-unused-function ${EXTRA_HIPCC_FLAGS:-}"
OUT="${LDPC_AMD_OUT:-$ROOT/ldpc_erasure_codes_amd/libldpc_erasure_amd.so}"
OBJ="${LDPC_AMD_OBJDIR:-$HERE/obj}"
mkdir -p "$OBJ"
echo "$FLAGS" > "$OBJ/flags.new"
if ! cmp -s "$OBJ/flags.new" "$OBJ/flags" 2>/dev/null; then rm -f "$OBJ"/*.o; mv "$OBJ/flags.new" "$OBJ/flags"; fi
COMMON="$HERE/internal.h $ROOT/include/ldpc_erasure_amd.h $ROOT/include/ldpc_erasure_amd_synth.h $ROOT/include/ldpc_erasure_amd_wire.h $ROOT/include/ldpc_erasure_amd_multi.h"
declare -A DEPS=(
  [kernels.hip]="$HERE/gf256_dev.h $HERE/peel_relax.inc $HERE/ml_kernel.inc $HERE/ml_pi.inc $HERE/rs_kernels.inc $HERE/fpga_kernels.inc"
  [api.cpp]="$HERE/builtin_codes_gen.inc"
  [wire.cpp]=""
  [multi.hip]=""
)
pids=()
for src in kernels.hip api.cpp wire.cpp multi.hip; do
  o="$OBJ/${src%.*}.o"
  stale=0
  [ -f "$o" ] || stale=1
  for d in "$HERE/$src" $COMMON ${DEPS[$src]}; do
    if [ "$stale" = 0 ] && [ "$d" -nt "$o" ]; then stale=1; fi
  done
  if [ "$stale" = 1 ]; then
    ( "$HIPCC" $FLAGS -c -o "$o.tmp" "$HERE/$src" && mv "$o.tmp" "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ/kernels.o" "$OBJ/api.o" "$OBJ/wire.o" "$OBJ/multi.o"
echo "$OUT"
