#!/bin/bash
# cfg 3 at S = 1024 under the ML-stage knobs (fast path on/off, factorisation beside / behind the packet kernel)
cd $GRAFT_REPO_ROOT
for env in ${MODES:-"LDPC_AMD_ML_PI=0" "LDPC_AMD_ML_PI=1,LDPC_AMD_ML_OVERLAP=0" "LDPC_AMD_ML_PI=1" "LDPC_AMD_ML_PI=2"}; do
  env=${env//,/ }
  env $env python3 bench.py --config 3 --no-s1 --steps 8 --warmup 2 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; }
  python3 -c "
import json;d=json.load(open('/tmp/b.json'));c=d['configs']['cfg3_S1024'] if 'configs' in d else d
print('$env', round(c['ms_per_step'],3), {k:round(v,3) for k,v in c['kernel_ms'].items()}, c.get('verified'))"
done
