#!/usr/bin/env python3
"""GEMM micro-benchmark on random data: the encoder's GEMM shapes through the C ABI
(mpreid_gemm_f16_nt_ex), interleaved rounds in one process.  MPREID_TUNE=gemm_big=0/1/2 selects the
kernel.  Usage: python tools/gemm_bench.py [--m 65536] [--reps 20] [--only fc1]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import _lib  # noqa: E402

SHAPES = {"qkv": (2304, 768, 1), "out": (768, 768, 2), "fc1": (3072, 768, 3), "fc2": (768, 3072, 2), "f32": (768, 768, 0),
          # the `split` precision mode (fp16 operand pairs, 3 products per multiply-add): epilogues 10 / 11 / 12
          "sqkv": (2304, 768, 10), "sout": (768, 768, 11), "sfc1": (3072, 768, 12), "sfc2": (768, 3072, 11)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    L = _lib.load()
    dev = _lib.require_gpu()
    names = [n for n in SHAPES if (a.only and n in a.only.split(",")) or (not a.only and not n.startswith("s"))]
    if a.only == "split":
        names = ["sqkv", "sout", "sfc1", "sfc2"]
    bufs = {}
    for n in names:
        N, K, epi = SHAPES[n]
        A = (torch.rand((a.m, K), device=dev) * 2 - 1)
        W = (torch.rand((N, K), device=dev) * 2 - 1) * 0.05
        bias = torch.randn(N, device=dev)
        if epi >= 10:
            def pair(x, scale):
                y = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=torch.float16, device=dev)
                _lib.check(L.mpreid_split_pack_f32(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], scale, C.c_void_p(y.data_ptr()),
                                                   _lib.stream_ptr()), "pack")
                return y
            A, W = pair(A.contiguous(), 1.0), pair(W.contiguous(), 2.0 ** 13)
            out = torch.zeros((a.m, 2 * N), device=dev, dtype=torch.float16) if epi == 12 else torch.zeros((a.m, N), device=dev)
        else:
            A, W = A.half(), W.half()
            out = torch.zeros((a.m, N), device=dev, dtype=torch.float16 if epi in (1, 3) else torch.float32)
        bufs[n] = (A, W, bias, out, N, K, epi)
    s = _lib.stream_ptr()
    res = {n: [] for n in names}
    for rnd in range(a.rounds + 1):
        for n in names:
            A, W, bias, out, N, K, epi = bufs[n]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                if epi >= 10:
                    _lib.check(L.mpreid_gemm_f16_split_nt(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()),
                                                          C.c_void_p(out.data_ptr()), C.c_void_p(bias.data_ptr()), a.m, N, K,
                                                          2.0 ** -13, epi, s), "gemm")
                else:
                    _lib.check(L.mpreid_gemm_f16_nt_ex(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()),
                                                       C.c_void_p(out.data_ptr()), C.c_void_p(bias.data_ptr()), a.m, N, K,
                                                       epi, s), "gemm")
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[n].append(e0.elapsed_time(e1) / a.reps)
    for n in names:
        _, _, _, _, N, K, epi = bufs[n]
        ms = sorted(res[n])[len(res[n]) // 2]
        mult = 3.0 if epi >= 10 else 1.0   # executed products per logical multiply-add
        print(f"{n:4s} M={a.m} N={N} K={K} epi={epi} tune={os.environ.get('MPREID_TUNE', '')}: {ms*1e3:8.1f} us  "
              f"{mult*2.0*a.m*N*K/ms/1e9:7.1f} TFLOP/s executed (min {mult*2.0*a.m*N*K/min(res[n])/1e9:.1f})")


if __name__ == "__main__":
    main()
