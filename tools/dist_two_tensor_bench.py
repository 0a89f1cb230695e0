#!/usr/bin/env python3
"""Stored 20k x 20k x 768 distance matrix of TWO tensors (q != g: the evaluator's real shape, utils/metrics.py:7-13), one-pass
fp16 and 3-term split: the 256 x 256 one-workgroup-per-CU kernel against the two-workgroups-per-CU 256 x 128 kernel
(MPREID_TUNE dist_p2_full).  The tuning string is latched per process: this script re-runs itself per setting, alternating, on one
device.   Usage: python tools/dist_two_tensor_bench.py [n d rounds]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("_DTT_CHILD"):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
    import torch
    from mpreid import ops, synth
    n, d = int(sys.argv[1]), int(sys.argv[2])
    f, _ = synth.clustered_features(2 * n, d, 3.0, seed=1234)
    q, g = torch.from_numpy(f[:n]).cuda(), torch.from_numpy(f[n:]).cuda()
    out = torch.empty((n, n), device="cuda")
    res = []
    for mode, name in ((ops.GEMM_F16_FAST, "fp16"), (ops.GEMM_F16_SPLIT3, "split3")):
        for _ in range(3):
            ops.euclidean_distance(q, g, mode=mode, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.euclidean_distance(q, g, mode=mode, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(f"{name} {ms:.3f} ms ({2.0 * n * n * d / ms / 1e9:.0f} TF/s algorithmic)")
    print(os.environ.get("MPREID_TUNE", "(default)"), "|", " | ".join(res), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "20000"
d = sys.argv[2] if len(sys.argv) > 2 else "768"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for _ in range(rounds):
    for tune in ("dist_p2_full=0", "dist_p2_full=1"):
        subprocess.run([sys.executable, os.path.abspath(__file__), n, d], env=dict(os.environ, _DTT_CHILD="1", MPREID_TUNE=tune), check=True)
