#!/bin/bash
# The runs of tools/peel_lab.hip behind profiles/round4_s1_relaxation.txt (on the GPU box, through gpurun):
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/peel_lab tools/peel_lab.hip
# arguments of peel_lab: code.csr.bin coef_seed frames per|bursty|parity max_sweeps [frames per workgroup] [tables in global memory 0/1] [1]
#                        [evaluations of a chunk per visit: 1 (shipped), 0 = until it stops improving, -K = again at once while <= K lanes improve]
D=ldpc_erasure_codes_amd/data
A=$D/code_n2040_k1530.csr.bin; C=$D/code_n4080_k3060.csr.bin; B=$D/code_n4000_k2000.csr.bin; DD=$D/code_n2000_k1000.csr.bin
run() { echo "== $*"; timeout -k 5 120 tools/bin/peel_lab "$@" | grep -E "frames,|mismatch|evaluations|kernel"; }
run $A 2040 4096 0.10 10 16 0 1 1
run $A 2040 65536 0.10 10 9 0 1 1
run $A 2040 65536 0.10 10 12 1 1 1
run $C 4080 65536 0.10 10 6 1 1 1
run $B 4000 32768 0.10 10 16 1 1 1
run $A 7 4096 bursty 10 16 0 1 1
run $DD 11 2048 0.38 50 16 0 1 1
run $A 3 2048 parity 10 16 0 1 1
run $A 6 512 0.9 3 16 0 1 1
run $A 5 512 0.0 10 16 0 1 1
for im in 0 -1 -2 -4; do run $A 2040 4096 0.10 10 16 0 1 $im; run $A 3 2048 parity 10 16 0 1 $im; run $A 7 4096 bursty 10 16 0 1 $im; done
