#!/usr/bin/env python3
"""Exhaustive bit comparison of the oracle's np.exp restatement (include/mpreid_numerics.h: mpreid_np_expf) with
np.exp(float32) over EVERY float32 in [-2, -0] (2^30 + 1 arguments; the re-ranking only feeds -O with O in [0, 1]).
Takes ~2 minutes on 1 core.  Run in the build container 2026-10 (numpy 2.2.6, AVX512F dispatch): 0 mismatches."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc  # noqa: E402

L = orc.lib()
fp = C.POINTER(C.c_float)
L.orc_expf_array.argtypes = [fp, fp, C.c_long]
L.orc_expf_array.restype = None
bad = tot = 0
step = 1 << 24
for s in range(0x80000000, 0xC0000001, step):
    x = np.arange(s, min(s + step, 0xC0000001), dtype=np.uint64).astype(np.uint32).view(np.float32)
    a = np.exp(x)
    b = np.empty_like(x)
    L.orc_expf_array(x.ctypes.data_as(fp), b.ctypes.data_as(fp), x.size)
    bad += int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)))
    tot += x.size
print(f"checked {tot} float32 arguments in [-2, -0]: {bad} mismatches")
sys.exit(1 if bad else 0)
