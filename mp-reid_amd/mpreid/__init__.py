"""mpreid — MI355X-native implementation of MP-ReID's evaluation hot path (host side).

Layout:  csrc/ (HIP kernels + C ABI, include/mpreid.h) -> mpreid/_lib.py (ctypes) -> mpreid/ops.py
(tensor API) -> utils/, model/, processor/ (the reference's own module paths and signatures).
"""
__all__ = ["synth"]
