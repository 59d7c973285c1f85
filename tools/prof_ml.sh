#!/bin/bash
# One-off: hardware counters of the ML kernel on the cfg 3 batch (S = 1024 factor + solve), one counter per pass.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/mlprof && mkdir -p gpurun_out/mlprof
for c in LDSBankConflict SALUBusy VALUBusy SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/mlprof/$c -o p -- python3 bench.py --config 3 --no-s1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/mlprof/$c.err || echo "$c failed"
  f=$(find gpurun_out/mlprof/$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$c" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        k = r["Kernel_Name"].replace("ldpc_amd::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    if "ldpc" in k:
        print(f"{sys.argv[2]:24s} {k:50s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
  fi
  find gpurun_out/mlprof/$c -type f -size +1M -delete
done
