#!/bin/bash
# Batch and packet size sweep of the headline workload: bench.py --frames F --S S --no-configs, one line each.
for fs in "1024 1024" "4096 1024" "16384 1024" "4096 256" "4096 4096" "4096 64"; do
  set -- $fs
  timeout -k 10 200 python bench.py --frames $1 --S $2 --no-configs --no-cpu-baseline --no-s1 --sustain-seconds 0 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('frames', $1, 'S', $2, 'frames/s %.3g' % l['value'], 'recovered GB/s %.0f' % l['recovered_GBps'], 'ms/step %.3f' % l['ms_per_step'], 'packet kernel frac %.3f' % l['roofline']['frac'], l['roofline']['kernel'], l['verified_bit_exact'])"
done
