#!/bin/bash
# rocprofv3 passes whose summaries are committed under profiles/round4_* (run on the GPU box through gpurun): kernel trace + stats
# and the FETCH_SIZE / WRITE_SIZE / VALUBusy counters in SEPARATE passes (never combined with a trace), the program directly after
# `--`, for the headline batch (cfg 2), cfg 3 at S = 1024 and the (4080,3060) code + RS(255,223) in packet mode (cfg4p).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r3prof
rm -rf $O && mkdir -p $O
A2="--steps 10 --warmup 3 --no-cpu-baseline --no-configs --sustain-seconds 0"
P2="--steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0"
A3="--config 3 --no-s1 --steps 10 --warmup 2 --no-cpu-baseline"
P3="--config 3 --no-s1 --steps 3 --warmup 2 --no-cpu-baseline"
A4="--config 4 --S 1024 --steps 5 --no-cpu-baseline"
P4="--config 4 --S 1024 --steps 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o cfg2 -- python3 bench.py $A2 > $O/bench_cfg2_stats.json 2> $O/bench_cfg2_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch2 -o cfg2 -- python3 bench.py $P2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write2 -o cfg2 -- python3 bench.py $P2 > /dev/null 2>&1
rocprofv3 --pmc VALUBusy --output-format csv -d $O/valu2 -o cfg2 -- python3 bench.py $P2 > /dev/null 2>&1 || true
echo cfg2 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o cfg3 -- python3 bench.py $A3 > $O/bench_cfg3_stats.json 2> $O/bench_cfg3_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch3 -o cfg3 -- python3 bench.py $P3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write3 -o cfg3 -- python3 bench.py $P3 > /dev/null 2>&1
rocprofv3 --pmc VALUBusy --output-format csv -d $O/valu3 -o cfg3 -- python3 bench.py $P3 > /dev/null 2>&1 || true
echo cfg3 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -o cfg4p -- python3 bench.py $A4 > $O/bench_cfg4p_stats.json 2> $O/bench_cfg4p_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch4 -o cfg4p -- python3 bench.py $P4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write4 -o cfg4p -- python3 bench.py $P4 > /dev/null 2>&1
rocprofv3 --pmc VALUBusy --output-format csv -d $O/valu4 -o cfg4p -- python3 bench.py $P4 > /dev/null 2>&1 || true
echo cfg4p done
find $O -name "*kernel_trace.csv" -size +20M -delete
S=$O/summ
python3 tools/summarize_profiles.py --tag round4 --workload cfg2 --stats $O/stats2 --fetch $O/fetch2 --write $O/write2 --outdir $S --cmd "python3 bench.py $P2" || true
python3 tools/summarize_profiles.py --tag round4_cfg3 --workload cfg3 --stats $O/stats3 --fetch $O/fetch3 --write $O/write3 --outdir $S --cmd "python3 bench.py $P3" || true
python3 tools/summarize_profiles.py --tag round4_cfg4p --workload cfg4p --stats $O/stats4 --fetch $O/fetch4 --write $O/write4 --outdir $S --cmd "python3 bench.py $P4" || true
python3 tools/summarize_profiles.py --tag round4 --valu $O/valu2 --outdir $S || true
python3 tools/summarize_profiles.py --tag round4_cfg3 --valu $O/valu3 --outdir $S || true
python3 tools/summarize_profiles.py --tag round4_cfg4p --valu $O/valu4 --outdir $S || true
cp $O/bench_cfg2_stats.json $S/round4_cfg2_bench_line_profiled.json; cp $O/bench_cfg3_stats.json $S/round4_cfg3_bench_line_profiled.json; cp $O/bench_cfg4p_stats.json $S/round4_cfg4p_bench_line_profiled.json
find $O -type f -size +2M -delete
du -sh $O
