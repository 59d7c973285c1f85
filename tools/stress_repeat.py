#!/usr/bin/env python3
"""Run-to-run determinism of the whole decode step on the batches bench.py times: the same batch decoded N times must give the
same bytes, sweep counts, residuals and status words every time (accumulation order varies -- LDS atomics, dynamic hand-out of
work items, two ML systems per CU, matrices in the L2-backed scratch -- the result must not).  cfg 3 (hybrid ML, rank-deficient
frames included) at S = 1024 / 64 / 1 and cfg 2; exits non-zero on the first difference."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench

    class A:
        pass
    g = bench.Gpu(A(), 0, 1, 0)
    n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    bad = 0
    for cfg, S in (("cfg3", 1024), ("cfg3", 64), ("cfg3", 1), ("cfg2", 1024), ("cfg2", 16)):
        h, n, k = g.code(1)
        cw, sym, era, _ = g.make_batch(cfg, 1, S, frame0=0, nframes=4096)
        ref = None
        for rep in range(n_rep):
            out, sw, res, st = g.ctx.decode(h, sym, era)
            torch.cuda.synchronize()
            cur = (out.clone(), sw.clone(), res.clone(), st.clone())
            if ref is None:
                ref = cur
                ok = st <= 1
                assert torch.equal(out[ok], cw[ok]), (cfg, S)
            else:
                same = all(torch.equal(a, b) for a, b in zip(cur, ref))
                if not same:
                    bad += 1
                    diff = (cur[0] != ref[0]).reshape(cur[0].shape[0], -1).any(dim=1).nonzero().flatten()[:8].tolist()
                    print(f"{cfg} S={S} repeat {rep}: DIFFERS from the first run; frames {diff}, status {[int(st[f]) for f in diff]}")
        print(f"{cfg} S={S}: {n_rep} repeats, ML frames {int((ref[2] > 0).sum())}, rank deficient {int((ref[3] == 2).sum())}: "
              f"{'identical' if bad == 0 else 'NOT identical'}", flush=True)
        del cw, sym, era, ref
        torch.cuda.empty_cache()
    g.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
