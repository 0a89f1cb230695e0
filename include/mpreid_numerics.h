/* mpreid_numerics.h — scalar numerics shared by the HIP kernels and by the CPU oracle.
 *
 * The re-ranking path of the reference (utils/reranking.py:29-100) is defined by a SEQUENCE OF
 * ROUNDINGS (points 1-11).  To make "HIP result == oracle result, bit for
 * bit" a testable statement, every scalar step whose result depends on rounding is written once,
 * here, from IEEE-754 primitives only (+, -, *, /, fmaf, rintf, integer ops), so that gcc on the
 * host and hipcc on gfx950 evaluate the very same operation sequence.  Both sides are compiled
 * with -ffp-contract=off; every fused multiply-add is an explicit fmaf().
 *
 *   mpreid_f32_to_f16 / mpreid_f16_to_f32   IEEE binary16 <-> binary32, round-to-nearest-even,
 *                                           subnormals kept (numpy's float16 cast semantics)
 *   mpreid_h_add/sub/mul/div                numpy float16 arithmetic: operate in fp32, round the
 *                                           fp32 result to fp16 (double rounding included, as numpy)
 *   mpreid_expf                             exp() in fp32, < 1 ulp, branch-free core
 *
 * Plain C99; also valid HIP device code (MPREID_HD expands to __host__ __device__ under hipcc).
 */
#ifndef MPREID_NUMERICS_H
#define MPREID_NUMERICS_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define MPREID_HD __host__ __device__ __forceinline__
#else
#define MPREID_HD static inline
#endif

MPREID_HD uint32_t mpreid_f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
MPREID_HD float mpreid_bits_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* binary32 -> binary16 bits, round to nearest even, subnormals and overflow to inf handled. */
MPREID_HD uint16_t mpreid_f32_to_f16(float f) {
    uint32_t x = mpreid_f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x0200u | ((ax >> 13) & 0x3ffu)) : 0u));
    }
    if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax < 0x38800000u) { /* < 2^-14: subnormal half (or zero) */
        if (ax < 0x33000000u) { /* < 2^-25: rounds to zero (2^-25 itself ties to even = 0) */
            return (uint16_t)sign;
        }
        /* value = m * 2^(e-150) with m 24-bit incl. hidden one; half subnormal unit is 2^-24 */
        uint32_t e = ax >> 23;
        uint32_t m = (ax & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126u - e; /* e in [102,112] -> shift in [14,24] */
        uint32_t q = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (q & 1u))) q += 1u;
        return (uint16_t)(sign | q);
    }
    /* normal half: rebias exponent 127 -> 15, keep 10 mantissa bits, RNE on the 13 dropped */
    {
        uint32_t r = ax - 0x38000000u; /* exponent rebias */
        uint32_t q = r >> 13;
        uint32_t rem = r & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q += 1u; /* may carry into exponent: fine */
        return (uint16_t)(sign | q);
    }
}

MPREID_HD float mpreid_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t out;
    if (e == 0u) {
        if (m == 0u) {
            out = sign;
        } else { /* subnormal: m * 2^-24, exactly representable in fp32 */
            float v = (float)m * 5.9604644775390625e-08f; /* 2^-24, exact product */
            out = sign | mpreid_f32_bits(v);
        }
    } else if (e == 31u) {
        out = sign | 0x7f800000u | (m << 13);
    } else {
        out = sign | ((e + 112u) << 23) | (m << 13);
    }
    return mpreid_bits_f32(out);
}

/* numpy float16 binary ops: npy_half a,b -> float, op in float, -> npy_half */
MPREID_HD uint16_t mpreid_h_add(uint16_t a, uint16_t b) {
    return mpreid_f32_to_f16(mpreid_f16_to_f32(a) + mpreid_f16_to_f32(b));
}
MPREID_HD uint16_t mpreid_h_sub(uint16_t a, uint16_t b) {
    return mpreid_f32_to_f16(mpreid_f16_to_f32(a) - mpreid_f16_to_f32(b));
}
MPREID_HD uint16_t mpreid_h_mul(uint16_t a, uint16_t b) {
    return mpreid_f32_to_f16(mpreid_f16_to_f32(a) * mpreid_f16_to_f32(b));
}
MPREID_HD uint16_t mpreid_h_div(uint16_t a, uint16_t b) {
    return mpreid_f32_to_f16(mpreid_f16_to_f32(a) / mpreid_f16_to_f32(b));
}
/* np.minimum on two finite non-negative halves == min of the bit patterns */
MPREID_HD uint16_t mpreid_h_min_nonneg(uint16_t a, uint16_t b) { return a < b ? a : b; }

/* exp(x) in fp32.  n = rint(x*log2(e)); r = x - n*ln2 (Cody-Waite, two fmaf); degree-7 Horner
 * polynomial for e^r on |r| <= ln2/2 (truncation error 5e-9 relative); scale by 2^n through the
 * exponent field (two-step so that results in the subnormal range are still produced by one
 * correctly rounded multiply).  Max error measured against double exp(): see tests/test_oracle.py. */
MPREID_HD float mpreid_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return mpreid_bits_f32(0x7f800000u);
    if (x < -103.97208404541015625f) return 0.0f;
    {
        float n = rintf(x * 1.44269502162933349609375f);
        float r = fmaf(n, -0.693145751953125f, x);            /* ln2 high part: 12 significant bits */
        r = fmaf(n, -1.42860676533018704e-06f, r);            /* ln2 low part */
        float p = 1.98412701138295233249664306640625e-4f;     /* 1/5040 */
        p = fmaf(p, r, 1.388888922519981861114501953125e-3f); /* 1/720 */
        p = fmaf(p, r, 8.3333337679505348205566406250e-3f);   /* 1/120 */
        p = fmaf(p, r, 4.16666679084300994873046875e-2f);     /* 1/24 */
        p = fmaf(p, r, 0.16666667163372039794921875f);        /* 1/6 */
        p = fmaf(p, r, 0.5f);
        p = fmaf(p, r, 1.0f);
        p = fmaf(p, r, 1.0f);
        {
            int ni = (int)n; /* in [-150, 128] */
            int n1 = ni / 2, n2 = ni - n1;
            float s1 = mpreid_bits_f32((uint32_t)(n1 + 127) << 23);
            float s2 = mpreid_bits_f32((uint32_t)(n2 + 127) << 23);
            return (p * s1) * s2;
        }
    }
}

/* Python: int(np.around(k1 / 2)) + 1 with round-half-to-even (utils/reranking.py:60,62) */
MPREID_HD int mpreid_half_k1(int k1) {
    int q = k1 / 2;
    if (k1 & 1) { /* x.5 -> nearest even */
        q = (q & 1) ? q + 1 : q;
    }
    return q + 1;
}

#endif /* MPREID_NUMERICS_H */
