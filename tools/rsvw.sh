for vw in 4 2 1; do
  LDPC_AMD_RS_VW=$vw timeout -k 10 300 python bench.py --config 4 --S 1024 --no-cpu-baseline 2>/dev/null > gpurun_out/r3_rsvw_$vw.json
  python - <<PY
import json
l=json.loads(open('gpurun_out/r3_rsvw_$vw.json').read().strip().splitlines()[-1]); r=l["configs"]["cfg4_S1024"]["rs"]
print("RS_VW", $vw, r["blocks_per_s"], r["kernel_ms"], r["roofline_frac"], r["verified"])
PY
done
