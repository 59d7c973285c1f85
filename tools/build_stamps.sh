#!/bin/bash
# Diagnostic builds of the library -> tools/bin/ (git-ignored; travels with gpurun):
#   libldpc_erasure_amd_stamps.so   -DLDPC_AMD_STAMPS  per-phase cycle counters (tools/stamp_*.py)
#   libldpc_erasure_amd_mldbg.so    -DLDPC_AMD_MLDBG   timing-only variants with WRONG bytes (tools/bound_cfg3.py, tools/sens_ml.py)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "$ROOT/tools/bin"
which="${1:-stamps}"
if [ "$which" = "stamps" ]; then FLAG=-DLDPC_AMD_STAMPS; else FLAG=-DLDPC_AMD_MLDBG; fi
EXTRA_HIPCC_FLAGS=$FLAG LDPC_AMD_OUT="$ROOT/tools/bin/libldpc_erasure_amd_$which.so" LDPC_AMD_OBJDIR="/tmp/obj_$which" bash "$ROOT/ldpc_erasure_codes_amd/csrc/build.sh"
