#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of a gpurun call into the small, committed summaries under profiles/.

    python tools/summarize_profiles.py --tag round1 --stats gpurun_out/prof_r1b --fetch gpurun_out/pmc_fetch \
        --write gpurun_out/pmc_write

* <tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (ldpc_* kernels only)
* <tag>_pmc_summary.json      per kernel: mean FETCH_SIZE / WRITE_SIZE per launch (KB as reported) and the
                              HBM bytes per launch after the gfx950 correction of
                              /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts a
                              16-B-per-lane coalesced stream at exactly half its bytes -> doubled; WRITE_SIZE
                              is exact for 16-B-per-lane streaming stores.  Separate --pmc passes.
bench.py reads <tag>_pmc_summary.json (if present) for roofline.traffic.
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.replace("ldpc_amd::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def pmc_mean(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "ldpc_amd" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v), max(v)) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--note", default="")
    ap.add_argument("--workload", default="cfg2", help="bench.py workload the passes ran (cfg2, cfg3, cfg4p): bench.py replays traffic per workload")
    ap.add_argument("--valu", help="directory of a --pmc VALUBusy pass -> <tag>_valubusy.json")
    ap.add_argument("--outdir", default=None, help="where the summaries go (default: profiles/)")
    ap.add_argument("--cmd", default="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline", help="the profiled command (for the note)")
    a = ap.parse_args()
    out = a.outdir or os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if a.stats:
        src = find(a.stats, "*kernel_stats.csv")
        rows = [r for r in csv.reader(open(src))]
        keep = [rows[0]] + [r for r in rows[1:] if "ldpc_amd" in r[0]]   # (every kernel of the library lives in namespace ldpc_amd, RS included)
        with open(os.path.join(out, f"{a.tag}_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f).writerows(keep)
        print("stats:", len(keep) - 1, "kernels")
    if a.fetch and a.write:
        fe = pmc_mean(find(a.fetch, "*counter_collection.csv"), "FETCH_SIZE")
        wr = pmc_mean(find(a.write, "*counter_collection.csv"), "WRITE_SIZE")
        summ = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `" + a.cmd + "`; values per launch. FETCH_SIZE/WRITE_SIZE are reported in KB; "
                        "fetch_bytes = 2 * FETCH_SIZE * 1024 (gfx950: wide coalesced reads are tallied at half), "
                        "write_bytes = WRITE_SIZE * 1024. For kernels launched with different batch shapes in one run "
                        "(peel: S=1 and packet batches) the mean mixes them; the max is the largest launch. " + a.note,
                "workload": a.workload, "kernels": {}}
        for k in sorted(set(fe) | set(wr)):
            f_kb, f_n, f_max = fe.get(k, (0.0, 0, 0.0))
            w_kb, w_n, w_max = wr.get(k, (0.0, 0, 0.0))
            summ["kernels"][k] = {"FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "launches": [f_n, w_n],
                                  "fetch_bytes": 2.0 * f_kb * 1024.0, "write_bytes": w_kb * 1024.0,
                                  "traffic_bytes": 2.0 * f_kb * 1024.0 + w_kb * 1024.0,
                                  "FETCH_SIZE_KB_max": f_max, "WRITE_SIZE_KB_max": w_max}
        with open(os.path.join(out, f"{a.tag}_pmc_summary.json"), "w") as f:
            json.dump(summ, f, indent=1)
        for k, v in summ["kernels"].items():
            print(f"{k:45s} traffic/launch {v['traffic_bytes'] / 1e9:8.3f} GB")
    if a.valu:
        vb = pmc_mean(find(a.valu, "*counter_collection.csv"), "VALUBusy")
        summ = {"note": "rocprofv3 --pmc VALUBusy (own pass), mean over the launches of each kernel, percent of cycles the vector "
                        "ALUs were busy", "kernels": {k: {"VALUBusy_pct": v[0], "launches": v[1], "max": v[2]} for k, v in sorted(vb.items())}}
        with open(os.path.join(out, f"{a.tag}_valubusy.json"), "w") as f:
            json.dump(summ, f, indent=1)
        for k, v in summ["kernels"].items():
            print(f"{k:45s} VALUBusy {v['VALUBusy_pct']:6.1f} %")


if __name__ == "__main__":
    main()
