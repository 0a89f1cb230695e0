#!/usr/bin/env python3
"""Race screen for the persistent GEMM's epilogues (counted-vmcnt LDS queues, ring reuse across tiles): the big
kernel is run many times per shape -- alone and with a second stream hammering HBM beside it -- and every FULL result
is compared bit for bit with the 128x128 kernel's (computed by a child process under MPREID_TUNE=gemm_big=0: the kernel
choice is latched per process).  Usage: python tools/gemm_stress.py [--iters 40]"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import _lib  # noqa: E402


def run(L, A, W, bias, out, epi, stream):
    if epi >= 10:   # split precision: A, W are fp16 pairs [hi | lo]; kseg = half the row length
        _lib.check(L.mpreid_gemm_f16_split_nt(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()),
                                              C.c_void_p(bias.data_ptr()), A.shape[0], W.shape[0], A.shape[1] // 2, 2.0 ** -13,
                                              epi, stream), "gemm")
        return
    _lib.check(L.mpreid_gemm_f16_nt_ex(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()),
                                       C.c_void_p(bias.data_ptr()), A.shape[0], W.shape[0], A.shape[1], epi, stream), "gemm")


SHAPES = [("qkv", 2304, 768, 1), ("out", 768, 768, 2), ("fc1", 3072, 768, 3), ("fc2", 768, 3072, 2),
          ("conv_relu", 512, 2048, 7), ("conv_add_relu", 2048, 512, 8), ("conv_add_relu_n256", 256, 64, 8),
          # split precision (round 3): k-block-major stage order with operand reuse, counted waits of 2 / 4 pieces
          ("s_qkv", 2304, 768, 10), ("s_out", 768, 768, 11), ("s_fc1", 3072, 768, 12), ("s_fc2", 768, 3072, 11),
          ("s_k64", 768, 64, 11), ("s_k128", 512, 128, 10)]
MS = (16384, 32768 + 256, 50944)   # whole sweeps, a ragged tile count on the blocked walk (129 tile rows), one on the owned walk (199)


def inputs(name, N, K, epi, Mb, dev):
    g = torch.Generator(device="cpu").manual_seed(zlib.crc32(f"{name}:{Mb}".encode()))   # same in parent and child
    A = (torch.rand((Mb, K), generator=g) * 2 - 1)
    W = ((torch.rand((N, K), generator=g) * 2 - 1) * 0.05)
    bias = torch.randn(N, generator=g).to(dev)
    if epi >= 10:
        def pair(x, scale):
            x = x.to(dev).contiguous()
            y = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=torch.float16, device=dev)
            _lib.check(_lib.load().mpreid_split_pack_f32(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], scale,
                                                         C.c_void_p(y.data_ptr()), _lib.stream_ptr()), "pack")
            return y
        init = (torch.randn((Mb, 2 * N), generator=g).half() if epi == 12 else torch.randn((Mb, N), generator=g)).to(dev)
        return pair(A, 1.0), pair(W, 2.0 ** 13), bias, init
    A, W = A.half().to(dev), W.half().to(dev)
    dt = torch.float16 if epi in (1, 3, 7, 8) else torch.float32
    init = torch.randn((Mb, N), generator=g).to(dt).to(dev)
    return A, W, bias, init


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--ref-dir", default="")
    a = ap.parse_args()
    L = _lib.load()
    dev = _lib.require_gpu()
    if a.ref_dir:   # child: reference outputs with the 128x128 kernel only
        assert os.environ.get("MPREID_TUNE") == "gemm_big=0"
        for name, N, K, epi in SHAPES:
            for Mb in MS:
                A, W, bias, init = inputs(name, N, K, epi, Mb, dev)
                run(L, A, W, bias, init, epi, _lib.stream_ptr())
                torch.cuda.synchronize()
                torch.save(init.cpu(), os.path.join(a.ref_dir, f"{name}_{Mb}.pt"))
        return
    assert os.environ.get("MPREID_TUNE") == "gemm_big=2", "run with MPREID_TUNE=gemm_big=2"
    tmp = tempfile.mkdtemp(prefix="gemm_stress_")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--ref-dir", tmp],
                          env=dict(os.environ, MPREID_TUNE="gemm_big=0"))
    side = torch.cuda.Stream()
    noise_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    noise_b = torch.empty_like(noise_a)
    bad = 0
    for name, N, K, epi in SHAPES:
        for Mb in MS:
            A, W, bias, init = inputs(name, N, K, epi, Mb, dev)
            ref = torch.load(os.path.join(tmp, f"{name}_{Mb}.pt")).to(dev)
            fails = 0
            for it in range(a.iters):
                out = init.clone()
                if it % 2:   # a copy stream saturating HBM beside the GEMM: different timing of DMAs and stores
                    with torch.cuda.stream(side):
                        noise_b.copy_(noise_a, non_blocking=True)
                run(L, A, W, bias, out, epi, _lib.stream_ptr())
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    fails += 1
            print(f"{name:20s} M={Mb:6d} N={N:5d} K={K:5d}: {a.iters - fails}/{a.iters} identical to the 128x128 kernel", flush=True)
            bad += fails
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
