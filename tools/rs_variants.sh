#!/bin/bash
# RS(255,223) on 1 KB packets: dwords per lane of the streaming kernel (knob RS_VW), one bench run each
for vw in 1 2 4; do
  LDPC_AMD_RS_VW=$vw timeout -k 10 300 python bench.py --config 4 --S 1024 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = l['configs']['cfg4_S1024']['rs']
print('RS_VW', $vw, 'blocks/s %.3g' % r['blocks_per_s'], 'kernel ms %.2f' % r['kernel_ms']['rs_decode'], 'of 8 TB/s %.3f' % r['roofline_frac'], r['verified'])"
done
