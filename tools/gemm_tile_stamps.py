#!/usr/bin/env python3
"""Per-tile phase timeline of the persistent 256 x 256 GEMM on the stored distance matrix (ablation build only:
MPREID_ABLATION=1 python mp-reid_amd/mpreid/build.py -> tools/ablation_lib/libmpreid_hip_abl.so; run with
MPREID_LIB=tools/ablation_lib/libmpreid_hip_abl.so MPREID_ALLOW_ABLATION=1).  Every workgroup stamps the 100 MHz real-time counter
after a tile's first barrier (operands of stage 0 landed), after the k-loop and after the epilogue.
Usage: python tools/gemm_tile_stamps.py [n d]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "enc":
    # the encoder's GEMM classes instead (M = 65536): python tools/gemm_tile_stamps.py enc sqkv|sout|sfc1|sfc2|qkv|out|fc1|fc2
    import ctypes as C
    os.environ["MPREID_GEMM_DBG"] = "32"
    os.environ["MPREID_TUNE"] = "gemm_big=2"
    from mpreid import _lib
    L = _lib.load()
    dev = _lib.require_gpu()
    SH = {"qkv": (2304, 768, 1), "out": (768, 768, 2), "fc1": (3072, 768, 3), "fc2": (768, 3072, 2),
          "sqkv": (2304, 768, 10), "sout": (768, 768, 11), "sfc1": (3072, 768, 12), "sfc2": (768, 3072, 11),
          # crossed shapes: which of N and the epilogue sets the k-loop time?
          "sqkv_n3072": (3072, 768, 10), "sfc1_n2304": (2304, 768, 12), "sqkv_n768": (768, 768, 10)}
    M = 65536
    for name in sys.argv[2:]:
        N, K, epi = SH[name]
        A = torch.rand((M, K), device=dev) * 2 - 1
        W = (torch.rand((N, K), device=dev) * 2 - 1) * 0.05
        bias = torch.randn(N, device=dev)
        if epi >= 10:
            def pair(x, scale):
                y = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=torch.float16, device=dev)
                _lib.check(L.mpreid_split_pack_f32(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], scale, C.c_void_p(y.data_ptr()),
                                                   _lib.stream_ptr()), "pack")
                return y
            A, W = pair(A.contiguous(), 1.0), pair(W.contiguous(), 2.0 ** 13)
            out = torch.zeros((M, 2 * N), device=dev, dtype=torch.float16) if epi == 12 else torch.zeros((M, N), device=dev)
        else:
            A, W = A.half(), W.half()
            out = torch.zeros((M, N), device=dev, dtype=torch.float16 if epi in (1, 3) else torch.float32)
        stamps = torch.zeros((256, 32, 8), dtype=torch.int64, device=dev)
        os.environ["MPREID_GEMM_STAMPS"] = "%x" % stamps.data_ptr()
        for _ in range(3):
            stamps.zero_()
            if epi >= 10:
                _lib.check(L.mpreid_gemm_f16_split_nt(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()),
                                                      C.c_void_p(bias.data_ptr()), M, N, K, 2.0 ** -13, epi, _lib.stream_ptr()), "gemm")
            else:
                _lib.check(L.mpreid_gemm_f16_nt_ex(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(bias.data_ptr()), M, N, K, epi, _lib.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().astype(np.int64)
        ok = s[:, :, 0] > 0
        kl = (s[:, :, 1] - s[:, :, 0])[ok] / 100.0
        ep = (s[:, :, 2] - s[:, :, 1])[ok] / 100.0
        nx = (s[:, 1:, 0] - s[:, :-1, 2])[ok[:, 1:]] / 100.0
        span = (s[:, :, 2].max() - s[:, :, 0][ok].min()) / 100.0
        mult = 3 if epi >= 10 else 1
        print("%-5s tiles/wg %d-%d | k-loop %.1f us (%.2f PF executed) | epilogue %.1f us | restart %.1f us | span %.0f us | idle share %.1f %%"
              % (name, ok.sum(1).min(), ok.sum(1).max(), kl.mean(), 256 * 256 * K * 2.0 * mult / kl.mean() / 1e6 * 256 / 1e3, ep.mean(),
                 nx.mean(), span, 100.0 * (ep.mean() + nx.mean()) / (kl.mean() + ep.mean() + nx.mean())), flush=True)
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
os.environ.setdefault("MPREID_GEMM_DBG", "32")
from mpreid import ops, synth  # noqa: E402

f, _ = synth.clustered_features(n, d, 3.0, seed=1234)
ft = torch.from_numpy(f).cuda()
out = torch.empty((n, n), device="cuda")
stamps = torch.zeros((256, 32, 8), dtype=torch.int64, device="cuda")
os.environ["MPREID_GEMM_STAMPS"] = "%x" % stamps.data_ptr()
for _ in range(3):
    stamps.zero_()
    ops.euclidean_distance(ft, ft, mode=ops.GEMM_F16_FAST, out=out)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64)
ok = s[:, :, 0] > 0
t0 = s[:, :, 0][ok].min()
tiles = ok.sum(1)
nwg = int((tiles > 0).sum())
print("tiles per workgroup: min %d max %d" % (tiles.min(), tiles.max()))
kl = (s[:, :, 1] - s[:, :, 0])[ok] / 100.0          # us
ep = (s[:, :, 2] - s[:, :, 1])[ok] / 100.0
nxt = (s[:, 1:, 0] - s[:, :-1, 2])[ok[:, 1:]] / 100.0
print("k-loop   us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f" % (kl.mean(), *np.percentile(kl, [10, 50, 90])))
print("epilogue us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f" % (ep.mean(), *np.percentile(ep, [10, 50, 90])))
print("restart  us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  (end of epilogue -> stage 0 of the next tile landed)"
      % (nxt.mean(), *np.percentile(nxt, [10, 50, 90])))
for name, a, b in (("tables ready (k-loop end -> first pass written to LDS)", 1, 3), ("pass 0 (reads, math, 4 stores issued)", 3, 4),
                   ("passes 1-3", 4, 6), ("passes 4-7", 6, 5), ("trailing barrier", 5, 2)):
    x = (s[:, :, b] - s[:, :, a])[ok & (s[:, :, 3] > 0)] / 100.0
    if x.size:
        print("  epilogue part %-58s us: mean %.2f  p50 %.2f  p90 %.2f" % (name, x.mean(), *np.percentile(x, [50, 90])))
end = s[:, :, 2].max()
print("kernel span %.1f us; first tile start spread %.2f us; last tile end spread %.2f us"
      % ((end - t0) / 100.0, (s[:nwg, 0, 0].max() - s[:nwg, 0, 0].min()) / 100.0,
         (np.array([s[w, tiles[w] - 1, 2] for w in range(nwg)]).max() - np.array([s[w, tiles[w] - 1, 2] for w in range(nwg)]).min()) / 100.0))
for w in [w for w in (0, 1, 8, 100) if w < nwg]:
    row = " ".join("%.1f/%.1f" % ((s[w, t, 1] - s[w, t, 0]) / 100.0, (s[w, t, 2] - s[w, t, 1]) / 100.0) for t in range(min(8, tiles[w])))
    print("wg %3d start %.1f us: k-loop/epilogue us per tile: %s" % (w, (s[w, 0, 0] - t0) / 100.0, row))
# how synchronised are the epilogues chip-wide?  fraction of workgroups inside an epilogue, sampled over time
ts = np.linspace(t0, end, 2000)
inside = np.zeros_like(ts)
for w in range(nwg):
    for t in range(tiles[w]):
        inside += (ts >= s[w, t, 1]) & (ts < s[w, t, 2])
print("workgroups inside an epilogue at a time: mean %.1f  max %d (of 256)" % (inside.mean(), inside.max()))
