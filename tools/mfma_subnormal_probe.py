"""Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs on gfx950?  (decides whether the hi/lo operand split of the
encoder's `split` precision mode needs per-row power-of-two scaling to keep `lo` normal.)

A[m][k] = 2^-20 (fp16 subnormal), B[n][k] = 1.0  ->  C = K * 2^-20 when honoured, 0 when flushed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]

import torch  # noqa: E402
from mpreid import ops  # noqa: E402

m = n = 256
k = 64
for val in (2.0 ** -20, 2.0 ** -24, 2.0 ** -15, 2.0 ** -14):
    a = torch.full((m, k), val, dtype=torch.float16, device="cuda")
    b = torch.ones((n, k), dtype=torch.float16, device="cuda")
    c = ops.gemm_f16_nt(a, b)
    c2 = ops.gemm_f16_nt(b, a)
    print(f"a = {val:.3e} (fp16 {float(a[0, 0]):.3e}): A-side sum {float(c[0, 0]):.6e}  B-side sum {float(c2[0, 0]):.6e}  "
          f"expected {k * float(a[0, 0]):.6e}")
# subnormal x subnormal-ish product magnitude: 2^-20 * 2^-10 = 2^-30 (fp32 normal) accumulates?
a = torch.full((m, k), 2.0 ** -20, dtype=torch.float16, device="cuda")
b = torch.full((n, k), 2.0 ** -10, dtype=torch.float16, device="cuda")
print("2^-20 x 2^-10 x 64 =", float(ops.gemm_f16_nt(a, b)[0, 0]), "expected", 64 * 2.0 ** -30)
