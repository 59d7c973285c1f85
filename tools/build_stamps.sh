#!/bin/bash
# Diagnostic build of the library with -DLDPC_AMD_STAMPS (per-phase cycle counters) -> tools/bin/ (git-ignored; travels with gpurun).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/ldpc_erasure_codes_amd/csrc"
mkdir -p "$ROOT/tools/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -DLDPC_AMD_STAMPS ${EXTRA_HIPCC_FLAGS:-} -shared \
    -o "$ROOT/tools/bin/libldpc_erasure_amd_stamps.so" "$SRC/kernels.hip" "$SRC/api.cpp" "$SRC/wire.cpp"
echo "$ROOT/tools/bin/libldpc_erasure_amd_stamps.so"
