#!/bin/bash
# Builds libldpc_erasure_amd.so for gfx950 (cross-compiles without a GPU).  Called by __graft_entry__.build().
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
python3 "$ROOT/tools/gen_builtin_codes.py" > /dev/null
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -Wall -Wno-unused-function"
OUT="$ROOT/ldpc_erasure_codes_amd/libldpc_erasure_amd.so"
"$HIPCC" $FLAGS ${EXTRA_HIPCC_FLAGS:-} -shared -o "$OUT" "$HERE/kernels.hip" "$HERE/api.cpp" "$HERE/wire.cpp"
echo "$OUT"
