#!/bin/bash
# rocprofv3 passes whose summaries are committed under profiles/ (run on the GPU box through gpurun): kernel trace + stats, and the
# FETCH_SIZE / WRITE_SIZE / VALUBusy counters in separate passes, for the headline batch (cfg 2) and for cfg 3 at S = 1024.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/r2prof && mkdir -p gpurun_out/r2prof
# cfg 2 (headline): kernel trace + the two PMC passes
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2prof/stats2 -o cfg2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --sustain-seconds 0 > gpurun_out/r2prof/bench_cfg2_stats.json 2> gpurun_out/r2prof/bench_cfg2_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2prof/fetch2 -o cfg2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2prof/write2 -o cfg2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 > /dev/null 2>&1
# cfg 3, S = 1024
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2prof/stats3 -o cfg3 -- python3 bench.py --config 3 --no-s1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2prof/bench_cfg3_stats.json 2> gpurun_out/r2prof/bench_cfg3_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2prof/fetch3 -o cfg3 -- python3 bench.py --config 3 --no-s1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2prof/write3 -o cfg3 -- python3 bench.py --config 3 --no-s1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc VALUBusy --output-format csv -d gpurun_out/r2prof/valu3 -o cfg3 -- python3 bench.py --config 3 --no-s1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || true
rocprofv3 --pmc VALUBusy --output-format csv -d gpurun_out/r2prof/valu2 -o cfg2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0 > /dev/null 2>&1 || true
find gpurun_out/r2prof -type f | xargs ls -la | head -60
# keep only the small CSVs (traces can be large)
find gpurun_out/r2prof -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/r2prof
# summaries (small) next to the raw CSVs; the raw traces are dropped when large
python3 tools/summarize_profiles.py --tag round2 --stats gpurun_out/r2prof/stats2 --fetch gpurun_out/r2prof/fetch2 --write gpurun_out/r2prof/write2 --outdir gpurun_out/r2prof/summ --cmd "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sustain-seconds 0" || true
python3 tools/summarize_profiles.py --tag round2_cfg3 --stats gpurun_out/r2prof/stats3 --fetch gpurun_out/r2prof/fetch3 --write gpurun_out/r2prof/write3 --outdir gpurun_out/r2prof/summ --cmd "python3 bench.py --config 3 --no-s1 --steps 3 --warmup 1 --no-cpu-baseline" || true
python3 tools/summarize_profiles.py --tag round2 --valu gpurun_out/r2prof/valu2 --outdir gpurun_out/r2prof/summ || true
python3 tools/summarize_profiles.py --tag round2_cfg3 --valu gpurun_out/r2prof/valu3 --outdir gpurun_out/r2prof/summ || true
find gpurun_out/r2prof -type f -size +2M -delete
du -sh gpurun_out/r2prof
