#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configs[1]:

    decoded frames/sec + recovered GB/s, n=2040 k=1530 GF(256) LDPC, 10 % uniform random erasures,
    batch = 4096 frames per GPU, hybrid MP + ML decoder (max 10 sweeps, ML on).

A "step" is one pass of the hot path (ldpc_amd_decode_batch: peel -> apply -> ML on residual frames) over
one batch of synthetic frames that are already resident in HBM when the timed region starts.  Symbols are
S-byte packets (--S, default 1024 = the reference's FPGA packet, OpenCL/host/src/main.cpp:42-47); the
Matlab-exact scalar case S = 1 is measured in the same run and reported under "s1".

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; frames are sharded with no data-path collective (weak scaling: 4096 frames per GPU);
RCCL is used once, for the final gather of the per-frame status words.  Rank 0 prints ONE JSON line.

The oracle (oracle/) is used here only (a) to time the CPU baseline and (b) to spot-check a few decoded
frames after the timed region; the measured path is the HIP library behind the C ABI.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CODE, K_CODE, CODE_IND = 2040, 1530, 1
PER = 0.10
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
SEED_SRC, SEED_ERA = 20261004, 20261005


SCATTER_KERNEL = "ldpc_scatter_kernel<16, 2, true, 8, false>"  # LPR=16 (256-byte row pieces), 2 pieces in flight, nt, 8 waves/SIMD, out of place
PEEL_S1_KERNEL = "ldpc_peel_kernel<14, true, false>"  # 14 neighbours per check, fused S = 1 apply, code tables in LDS


def pmc_traffic(kernel, frames, S):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (separate --pmc FETCH_SIZE /
    WRITE_SIZE passes of this same command, gfx950 correction applied: tools/summarize_profiles.py).  The counters
    cannot be read from inside this process, so the value is the one measured for the default batch
    (4096 frames, S = 1024 / 1); None when there is no summary or the shape differs."""
    path = os.path.join(ROOT, "profiles", "round1_pmc_summary.json")
    if frames != 4096 or S not in (1, 1024) or not os.path.exists(path):
        return None
    try:
        ks = json.load(open(path))["kernels"]
        k = ks.get(kernel) or next(v for name, v in ks.items() if name.startswith(kernel.split("<")[0] + "<")
                                   and (S == 1) == ("peel" in name and name.split(",")[1].strip().startswith("true")))
    except (KeyError, ValueError, StopIteration):
        return None
    if S == 1:  # the S = 1 launches are the large ones of that kernel in the profiled run
        return 2.0 * k["FETCH_SIZE_KB_max"] * 1024.0 + k["WRITE_SIZE_KB_max"] * 1024.0
    return k["traffic_bytes"]


def alg_bytes_per_frame(n, S):
    # symbols in + erasure flags in + Msg out + sweeps/residual words
    return 2 * n * S + n + 8


# ----------------------------------------------------------------------------------------------------
# CPU baseline ("port": the oracle's lane-vectorised restatement of the Matlab decoder), run BEFORE the
# GPU is touched so that forked workers never inherit a HIP context.
# ----------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    wid, S, nframes, chunk = args
    from ldpc_erasure_codes_amd import codes, synth
    from oracle import oracle_py
    code = codes.load_builtin(CODE_IND)
    oc = oracle_py.OracleCode(code)
    busy = 0.0
    done = 0
    frame0 = 1_000_000 + wid * nframes  # frames disjoint from the GPU batch, same generator
    while done < nframes:
        c = min(chunk, nframes - done)
        src = synth.source(SEED_SRC, frame0 + done, c, code.k, S)
        era = synth.erasures_uniform(SEED_ERA, frame0 + done, c, code.n, PER)
        if S == 1:
            cw = np.stack([oc.encode(src[f, :, 0]) for f in range(c)])
            t0 = time.perf_counter()
            out, sw, res, st = oc.decode_batch_s1(cw, era)
            busy += time.perf_counter() - t0
            assert np.array_equal(out, cw)
        else:
            cws = [oc.encode(src[f]) for f in range(c)]
            t0 = time.perf_counter()
            outs = [oc.decode_packets(cws[f], era[f]) for f in range(c)]
            busy += time.perf_counter() - t0
            assert all(np.array_equal(outs[f][0], cws[f]) for f in range(c))
        done += c
    return done, busy


def cpu_baseline(S, cores, frames_per_core):
    chunk = 256 if S == 1 else 8
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(w, S, frames_per_core, chunk) for w in range(cores)])
    wall = time.perf_counter() - t0
    rate = sum(d / b for d, b in res)  # workers run concurrently: aggregate = sum of per-worker rates
    return {"value": rate, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{sum(d for d, _ in res)} frames of the same workload (S={S}) decoded by oracle/oracle.c, "
                      f"{frames_per_core} per core on {cores} cores, decode time only ({wall:.1f} s wall incl. input generation)",
            "per_core": rate / cores}


def cpu_cfg1(reps=200):
    """BASELINE configs[0] (the reference's own CPU-runnable case): ONE binary (2040,1530) frame, message passing only
    (Matlab/My_LDPC_Erasure_Decoder.m, 50 sweeps max), FPGA data_in erasures at PER 9/64 -- the oracle on one core."""
    from ldpc_erasure_codes_amd import codes, synth
    from oracle import oracle_py
    code = codes.load_builtin(CODE_IND, 0)  # coefficient seed 0 = the binary code (all ones)
    oc = oracle_py.OracleCode(code)
    era = synth.fpga_erasures(1, 9, reps, code.n)
    busy, ok, sweeps = 0.0, 0, 0
    for f in range(reps):
        recv = np.zeros(code.n, dtype=np.int16)  # the all-zero codeword, as the FPGA source sends (data_in)
        recv[era[f] != 0] = -1
        t0 = time.perf_counter()
        msg, it = oc.binary_mp(recv, itenum=50)
        busy += time.perf_counter() - t0
        ok += int(not (msg < 0).any())
        sweeps += it
    return {"workload": "BASELINE cfg1: 1 binary (2040,1530) frame, MP only (<= 50 sweeps), PER 9/64, oracle on 1 core",
            "us_per_frame": busy / reps * 1e6, "frames": reps, "decoded": ok, "mean_sweeps": sweeps / reps}


# ----------------------------------------------------------------------------------------------------
def run_gpu(args, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    from ldpc_erasure_codes_amd import api, codes, sharding

    # LDPC_BENCH_BACKEND=gloo is only for rehearsing the N > 1 launch path on a box with fewer GPUs than ranks
    backend = os.environ.get("LDPC_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    ctx = api.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # torch events and our kernels share one stream
    ctx.selftest()
    h = ctx.load_builtin_code(CODE_IND, codes.DEFAULT_COEF_SEED[CODE_IND])
    n, k, _ = ctx.code_info(h)
    F = args.frames
    frame0 = rank * F  # every rank decodes its own, different frames

    def make_batch(S):
        src = torch.empty((F, k, S), dtype=torch.uint8, device=dev)
        ctx.synth_source(SEED_SRC, frame0, F, k, S, src)
        cw = ctx.encode(h, src if S > 1 else src.reshape(F, k))
        del src
        era = torch.empty((F, n), dtype=torch.uint8, device=dev)
        ctx.synth_erasures_uniform(SEED_ERA, frame0, F, n, PER, era)
        sym = cw.clone()
        sym[era.bool()] = 0x5A  # the payload of an erased symbol is garbage, never the true value
        return cw, sym, era

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(S, steps, warmup):
        cw, sym, era = make_batch(S)
        out = torch.empty_like(sym)
        sw = torch.empty(F, dtype=torch.int32, device=dev)
        res = torch.empty(F, dtype=torch.int32, device=dev)
        st = torch.empty(F, dtype=torch.int32, device=dev)
        counts = [F] * world
        for _ in range(warmup):
            ctx.decode(h, sym, era, out=out, sweeps=sw, residual=res, status=st)
        ctx.get_profile()
        ctx.set_profiling(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.decode(h, sym, era, out=out, sweeps=sw, residual=res, status=st)
        # the one collective of the job: final gather of the status words (12 B/frame) over RCCL/xGMI
        parts = sharding.gather_status(torch.stack([sw, res, st]), counts)
        barrier()
        dt = time.perf_counter() - t0
        ctx.set_profiling(False)
        prof = ctx.get_profile()
        dt = sharding.max_over_ranks(dt, dev)
        assert len(parts) == world
        # correctness of what was just timed: every frame of cfg 2 decodes to its codeword
        ok = bool(torch.equal(out, cw)) and int(st.max()) == 0
        hist = torch.bincount(sw, minlength=12).cpu().numpy().tolist()
        ml_rate = float((res > 0).float().mean())
        sample = (sym[:2].cpu().numpy(), era[:2].cpu().numpy(), out[:2].cpu().numpy(), sw[:2].cpu().numpy())
        inplace = None
        copy_gbps = None
        if S >= 16:
            # the copy rate this box reaches (same buffers, same streaming loads/stores), quoted next to
            # the nominal HBM peak; bytes moved = read + write
            nb = sym.numel()
            copy_ms = ctx.copy_probe(sym, out, reps=5)
            copy_gbps = 2.0 * nb / (copy_ms * 1e-3) / 1e9
        if S >= 16 and world == 1:
            # extension, reported separately and never as `value`: LDPC_AMD_INPLACE decodes inside the caller's frame
            # buffer and writes only the erased symbols (the reference always returns a copy)
            buf = sym.clone()
            for _ in range(2):
                ctx.decode(h, buf, era, sweeps=sw, residual=res, status=st, inplace=True)
            ctx.get_profile()
            ctx.set_profiling(True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                ctx.decode(h, buf, era, sweeps=sw, residual=res, status=st, inplace=True)
            torch.cuda.synchronize()
            dti = time.perf_counter() - t1
            ctx.set_profiling(False)
            pi = ctx.get_profile()
            inplace = {"ms_per_step": dti / steps * 1e3, "frames_per_s": F * steps / dti, "verified": bool(torch.equal(buf, cw)),
                       "kernel_ms": {kk: (v[0] / max(v[1], 1)) for kk, v in pi.items()}}
            del buf
        del cw, sym, era, out
        torch.cuda.empty_cache()
        return dt, prof, ok, hist, ml_rate, sample, inplace, copy_gbps

    result = {}
    for S in ([args.S, 1] if args.S != 1 else [1]):
        steps = args.steps if S == args.S else max(args.steps, 20)
        dt, prof, ok, hist, ml_rate, sample, inplace, copy_gbps = measure(S, steps, args.warmup)
        fps = world * F * steps / dt
        kind = "apply" if S > 1 else "peel"
        kms, kcnt = prof[kind]
        kavg = kms / max(kcnt, 1)  # ms per launch of the dominant kernel, HIP events on its own stream
        ab = alg_bytes_per_frame(n, S) * F
        ach = ab / (kavg * 1e-3) / 1e9 if kavg > 0 else 0.0
        result[S] = {
            "value": fps, "ms_per_step": dt / steps * 1e3, "recovered_GBps": fps * k * S / 1e9, "steps": steps,
            "verified": ok, "sweeps_hist": hist, "ml_trigger_rate": ml_rate,
            "kernel_ms": {kk: (v[0] / max(v[1], 1)) for kk, v in prof.items()},
            "roofline": {"bound": "hbm", "kernel": SCATTER_KERNEL if S > 1 else PEEL_S1_KERNEL,
                         "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic(SCATTER_KERNEL if S > 1 else PEEL_S1_KERNEL, F, S),
                         "alg_bytes_per_launch": ab, "avg_launch_ms": kavg,
                         "copy_kernel_GBps": copy_gbps,
                         "frac_of_copy": (ach / copy_gbps) if copy_gbps else None},
            "sample": sample, "inplace": inplace,
        }
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return result, n, k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step (BASELINE cfg 2: 4096)")
    ap.add_argument("--S", type=int, default=1024, help="bytes per symbol (1 = Matlab model, 1024 = FPGA packet)")
    ap.add_argument("--cpu-frames", type=int, default=None, help="CPU-baseline frames per core (default: sized for ~10 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    cpu = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))
        # ~3 ms/frame/core at S=1024 and ~0.2 ms at S=1 on a current x86 core -> about 10 s of CPU work per core
        cpu[args.S] = cpu_baseline(args.S, cores, args.cpu_frames or (1024 if args.S > 1 else 65536))
        if args.S != 1:
            cpu[1] = cpu_baseline(1, cores, args.cpu_frames or 65536)
        cpu["cfg1"] = cpu_cfg1()

    result, n, k = run_gpu(args, rank, world, local_rank)
    if rank != 0:
        return

    # spot-check of the timed output against the oracle (after the timed region)
    from ldpc_erasure_codes_amd import codes
    from oracle import oracle_py
    oc = oracle_py.OracleCode(codes.load_builtin(CODE_IND))
    for S, r in result.items():
        sym, era, out, sw = r.pop("sample")
        for f in range(sym.shape[0]):
            if S == 1:
                o, osw, _, _ = oc.decode_batch_s1(sym[f:f + 1], era[f:f + 1])
                r["verified"] = r["verified"] and bool(np.array_equal(o[0], out[f])) and int(osw[0]) == int(sw[f])
            else:
                o, _, it, _, _ = oc.decode_packets(sym[f], era[f])
                r["verified"] = r["verified"] and bool(np.array_equal(o, out[f])) and it == int(sw[f])

    main_r = result[args.S]
    line = {
        "metric": "decoded frames/sec (+ recovered GB/s), n=2040 k=1530 GF(256) LDPC hybrid MP+ML erasure decode",
        "value": main_r["value"], "unit": "frames/s", "recovered_GBps": main_r["recovered_GBps"],
        "n_gpus": world, "steps": main_r["steps"], "warmup": args.warmup, "ms_per_step": main_r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE cfg2: n=2040,k=1530 GF(256) LDPC (H_nb seed {2040}), 10% uniform random erasures, "
                               f"batch={args.frames} frames per GPU, S={args.S} bytes/symbol, max_sweeps=10, ML on",
                   "frames_per_gpu": args.frames, "S": args.S, "per": PER, "code": "n2040_k1530",
                   "sharding": f"{world} x {args.frames} independent frames, status gather over RCCL"},
        "verified_bit_exact": main_r["verified"], "sweeps_hist": main_r["sweeps_hist"],
        "ml_trigger_rate": main_r["ml_trigger_rate"], "kernel_ms": main_r["kernel_ms"],
        "roofline": main_r["roofline"],
    }
    if main_r.get("inplace"):
        ip = main_r["inplace"]
        ach_ip = alg_bytes_per_frame(n, args.S) * args.frames / (ip["kernel_ms"]["apply"] * 1e-3) / 1e9
        line["inplace_extension"] = {
            "note": "LDPC_AMD_INPLACE (out == sym, only erased symbols written; the reference always returns a copy). Reported "
                    "against the same algorithmic bytes as SURVEY.md 8(d) prescribes; not the headline value.",
            "frames_per_s": ip["frames_per_s"], "ms_per_step": ip["ms_per_step"], "verified_bit_exact": ip["verified"],
            "kernel_ms": ip["kernel_ms"], "roofline_frac_vs_algorithmic_bytes": ach_ip / HBM_PEAK_GBPS}
    if args.S in cpu:
        line["cpu_baseline"] = cpu[args.S]
        line["gpu_over_cpu"] = main_r["value"] / cpu[args.S]["value"]
    if "cfg1" in cpu:
        line["cfg1_cpu_reference_row"] = cpu["cfg1"]
    if 1 in result and args.S != 1:
        s1 = result[1]
        line["s1"] = {"note": "same batch with S=1 (one GF(256) element per symbol: the Matlab model, bit-exact incl. iterations); "
                              "latency/LDS-bound by construction, HBM fraction reported for completeness",
                      "value": s1["value"], "unit": "frames/s", "ms_per_step": s1["ms_per_step"], "steps": s1["steps"],
                      "verified_bit_exact": s1["verified"], "roofline": s1["roofline"]}
        if 1 in cpu:
            line["s1"]["cpu_baseline"] = cpu[1]
            line["s1"]["gpu_over_cpu"] = s1["value"] / cpu[1]["value"]
    print(json.dumps(line))


if __name__ == "__main__":
    main()
