#!/usr/bin/env python3
"""Compute side of a P-GPU row-sharded re-ranking, measured on ONE GPU: the phases of mpreid/distributed.py are run for
P virtual ranks one after the other; per phase the slowest rank is what a P-GPU run would wait for.  The all-gathers
are reported as bytes (they are host-side concatenations here): at ~150 GB/s per xGMI link they add well under a
millisecond each at these sizes.  Usage: python tools/sharded_rerank_estimate.py [N nq D] [world ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import distributed as D, ops, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else N // 5
d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
worlds = [int(x) for x in sys.argv[4:]] or [1, 2, 4, 8]
g = torch.Generator(device="cuda")
g.manual_seed(1234)
cent = torch.randn((N // 20, d), generator=g, device="cuda")
pid = torch.randint(0, N // 20, (N,), generator=g, device="cuda")
f = ops.l2_normalize(cent[pid] + 3.0 * torch.randn((N, d), generator=g, device="cuda"))
q, ga = f[:nq], f[nq:]
ref, st = ops.re_ranking(q, ga, 50, 15, 0.3, timing=True)
print(f"single call: {st['ms_total']:.2f} ms (algo {st['algo']})")
for algo, name in ((ops.RERANK_SPARSE, "sparse"), (ops.RERANK_SPARSE_SPLIT3, "split3"), (ops.RERANK_DENSE, "dense")):
    for w in worlds:
        D.re_ranking_virtual(q, ga, 50, 15, 0.3, w, algo=algo)          # warm
        tm = {}
        out = D.re_ranking_virtual(q, ga, 50, 15, 0.3, w, algo=algo, timings=tm)
        assert torch.equal(out, ref) if algo != ops.RERANK_SPARSE_SPLIT3 else float((out - ref).abs().max()) <= 1e-6
        phases = {k: round(max(v), 2) for k, v in tm.items() if isinstance(v, list)}
        if "phase4_jaccard_total" in phases:   # column-sharded index build: phase 4 = slowest rank's rows + count + fill + Jaccard
            sub = {k: phases.pop(k) for k in ("phase4_rows", "phase4_index_count", "phase4_index_fill", "phase4_jaccard")}
            phases["phase4 (rows + index count + index fill + jaccard)"] = phases.pop("phase4_jaccard_total")
            phases["phase4_steps_slowest"] = sub
            total = sum(v for k, v in phases.items() if not isinstance(v, dict))
            print(f"{name:6s} P={w}: per-phase slowest rank {phases} -> {total:.2f} ms compute; all-gather bytes {tm['all_gather_bytes']}",
                  flush=True)
            del out
            continue
        total = sum(phases.values())
        print(f"{name:6s} P={w}: per-phase slowest rank {phases} -> {total:.2f} ms compute; all-gather bytes {tm['all_gather_bytes']}",
              flush=True)
        del out
