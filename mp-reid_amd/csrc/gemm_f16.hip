// gemm_f16.hip — see gemm_f16.h for the design.  Roofline: MFMA fp16 (2.5 PFLOP/s dense peak).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gemm_f16.h"

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;

__device__ __forceinline__ void dma16(const void *gsrc, unsigned char *lds_dst_wave_base) {
    // LDS destination = wave-uniform base + lane*16; the global source is per lane.
    __builtin_amdgcn_global_load_lds((glb_ptr_t *)gsrc, (lds_ptr_t *)lds_dst_wave_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// 16-byte LDS-DMA / store with the address split as wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane offset.
// hipcc materialises one 64-bit VGPR pointer per access instead (32 of them in the residual epilogue: spills);
// written out, the base is scalar arithmetic and the lane offset one register.  M0 (LDS destination of the DMA)
// is saved and restored: the compiler tracks its own M0 values across the asm.
__device__ __forceinline__ void dma16_sv(uint64_t sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %1 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(sbase), "v"(voff), "s"(lds_dst)
                 : "memory");
}
// (the trailing s_nop 1 is required: a 128-bit store reads its data registers over several cycles and nothing inside an
// asm string is padded by hipcc's hazard recognizer -- cdna_hip_programming.md 5.7 item 1.  Without it the split-mode
// instance, whose schedule puts a v_pk_fma_f32 into the same registers right behind the store, wrote garbage into
// 1 % of the residual stream; the fp16 instance happened to have scalar address arithmetic in that slot.)
__device__ __forceinline__ void store16_sv(uint64_t sbase, unsigned voff, const f32x4 &v) {
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store8_sv(uint64_t sbase, unsigned voff, const u32x2 &v) {   // 8 bytes per lane, same addressing
    asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void store16_nt_sv(uint64_t sbase, unsigned voff, const f32x4 &v) {   // streaming variant
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
template <int F>   // (ablation builds) cache-policy flavours of the same store
__device__ __forceinline__ void store16_flav_sv(uint64_t sbase, unsigned voff, const f32x4 &v) {
    if constexpr (F == 1) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
    else if constexpr (F == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
    else if constexpr (F == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
    else if constexpr (F == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
    else if constexpr (F == 5) asm volatile("global_store_dwordx4 %0, %1, %2 sc0\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
    else store16_nt_sv(sbase, voff, v);
}

// s_waitcnt takes an immediate: callers pass a value that is a constant after unrolling, the switch folds away
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ float quick_gelu(float h) {
    // model/clip/model.py:159-161  x * sigmoid(1.702 x)
    // v_exp + v_rcp (1 ulp) instead of the ~15-instruction IEEE divide: the result is rounded to fp16
    return h * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896340736f * h));
}

// streaming (non-temporal) 16-byte store: the big outputs (hundreds of MB per launch) are read again only by a
// later kernel, long after they have left the 4 MB L2 -- written with the default policy they evict the A / W
// panels the MFMA loop is re-reading and back-pressure the store path at the end of every tile
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(void *p, const uint4 &v) {
    const u32x4_t nv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t *>(p));
}
__device__ __forceinline__ void store_nt(float *p, const float4 &v) {
    const f32x4_t nv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(nv, reinterpret_cast<f32x4_t *>(p));
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, 8, tm, tn);
    const int m0 = tm * GBM, n0 = tn * GBN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int K = g.K;
    // split precision mode: rows are [hi | lo] (K = 2 * kseg halfs) and the loop walks hi.hi', lo.hi', hi.lo'
    constexpr bool SPLIT = gemm_epi_is_split(EPI);
    const int nseg = SPLIT ? g.kseg / GBK : 0;

    // ---- staging: wave w moves rows [32w, 32w+32) of both tiles, 8 rows per DMA instruction ----
    const int srow = lane >> 3;
    const int gchunk = (lane & 7) ^ srow; // source chunk that lands in LDS slot (lane & 7) of row srow
    const _Float16 *a_src = g.A + (int64_t)(m0 + wave * 32 + srow) * K + gchunk * 8;
    const _Float16 *b_src = g.W + (int64_t)(n0 + wave * 32 + srow) * K + gchunk * 8;
    auto stage = [&](int s, int kt) {
        unsigned char *abase = smem + s * G_STAGE_BYTES + wave * 4096;
        unsigned char *bbase = abase + G_TILE_BYTES;
        const int koff = kt * GBK;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dma16(a_src + (int64_t)t * 8 * K + koff, abase + t * 1024);
            dma16(b_src + (int64_t)t * 8 * K + koff, bbase + t * 1024);
        }
    };

    // ---- fragment addresses (bytes inside a tile) ----
    const int frow = lane & 15, fq = lane >> 4;
    const int a_row_off = (wm * 64 + frow) * 128;
    const int b_row_off = (wn * 64 + frow) * 128;
    int ksw[2];
    ksw[0] = ((0 + fq) ^ (lane & 7)) << 4;
    ksw[1] = ((4 + fq) ^ (lane & 7)) << 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int s) {
        const unsigned char *at = smem + s * G_STAGE_BYTES;
        const unsigned char *bt = at + G_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f16x8 *>(at + a_row_off + i * 2048 + ksw[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8 *>(bt + b_row_off + j * 2048 + ksw[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (SPLIT) {
        // Split precision: per 32-wide k block the three products hi.hi', lo.hi', hi.lo' in THIS order -- the accumulation
        // order of the persistent 256 x 256 kernel, so that a row gives the same bits whichever kernel its batch selects.
        // Plain synchronous form (this kernel only sees the small shapes of the split mode: the CLS-only tail, reduced test
        // models): the hi and lo parts of a 64-wide k block of both operands are staged into two stage buffers
        // (buffer 0: hi | hi', buffer 1: lo | lo'), then consumed.
        for (int kb = 0; kb < nseg; ++kb) {
            if (kb) __syncthreads();   // every wave is done reading the buffers
            {
                unsigned char *abase = smem + wave * 4096;
#pragma unroll
                for (int part = 0; part < 2; ++part) {       // 0: hi, 1: lo
                    const int koff = part * g.kseg + kb * GBK;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        dma16(a_src + (int64_t)t * 8 * K + koff, abase + part * G_STAGE_BYTES + t * 1024);
                        dma16(b_src + (int64_t)t * 8 * K + koff, abase + part * G_STAGE_BYTES + G_TILE_BYTES + t * 1024);
                    }
                }
            }
            __syncthreads();   // drains the DMA and publishes the four parts
            const unsigned char *ah = smem, *bh = smem + G_TILE_BYTES, *al = smem + G_STAGE_BYTES,
                                *bl = smem + G_STAGE_BYTES + G_TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 afh[4], afl[4], bfh[4], bfl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    afh[i] = *reinterpret_cast<const f16x8 *>(ah + a_row_off + i * 2048 + ksw[ks]);
                    afl[i] = *reinterpret_cast<const f16x8 *>(al + a_row_off + i * 2048 + ksw[ks]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bfh[j] = *reinterpret_cast<const f16x8 *>(bh + b_row_off + j * 2048 + ksw[ks]);
                    bfl[j] = *reinterpret_cast<const f16x8 *>(bl + b_row_off + j * 2048 + ksw[ks]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afh[i], bfh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afl[i], bfh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afh[i], bfl[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    } else {
    const int nkt = K / GBK;
    stage(0, 0);
    __syncthreads(); // drains the DMA (vmcnt(0)) and publishes tile 0
    int cur = 0;
    for (int kt = 0; kt < nkt - 1; ++kt) {
        stage(cur ^ 1, kt + 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);
    }

    // ---- epilogue.  C/D map of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg ----
    if constexpr (EPI == GE_BIAS_F16 || EPI == GE_BIAS_GELU || EPI == GE_BIAS_RELU) {
        __syncthreads(); // every wave is done reading the staging buffers
        _Float16 *wreg = reinterpret_cast<_Float16 *>(smem) + wave * (64 * 72);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bias = g.bias[n0 + wn * 64 + j * 16 + frow];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bias;
                    if (EPI == GE_BIAS_GELU) v = quick_gelu(v);
                    if (EPI == GE_BIAS_RELU) v = v < 0.f ? 0.f : v;   // relu commutes with the rounding below
                    wreg[(i * 16 + fq * 4 + r) * 72 + j * 16 + frow] = (_Float16)v;
                }
        }
        __syncthreads();
        _Float16 *out = reinterpret_cast<_Float16 *>(g.out);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int lr = it * 8 + (lane >> 3), ch = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4 *>(wreg + lr * 72 + ch * 8);
            *reinterpret_cast<uint4 *>(out + (int64_t)(m0 + wm * 64 + lr) * g.ldo + n0 + wn * 64 + ch * 8) = v;
        }
    } else if constexpr (EPI == GE_S_BIAS_GELU || EPI == GE_S_BIAS_RELU_PAIR) {
        // fp16 pair out [M][2N] (RELU_PAIR: [M][2 pair_c], columns >= pair_c dropped): the hi parts, then the lo parts, through the same patch
        _Float16 *wreg = reinterpret_cast<_Float16 *>(smem) + wave * (64 * 72);
        _Float16 *out = reinterpret_cast<_Float16 *>(g.out);
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bias = g.bias[n0 + wn * 64 + j * 16 + frow];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int rp = 0; rp < 4; rp += 2) {   // (on element pairs, as in the persistent kernel: the same instructions)
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                        const f32x2 a = {acc[i][j][rp], acc[i][j][rp + 1]};
                        f32x2 h;
                        h = __builtin_elementwise_fma(a, f32x2{g.oscale, g.oscale}, f32x2{bias, bias});
                        f32x2 v;
                        if constexpr (EPI == GE_S_BIAS_GELU) {
                            const f32x2 t = h * (-1.702f * 1.44269504088896340736f);
                            const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
                            v = h * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};   // quick_gelu
                        } else {
                            v = f32x2{h[0] < 0.f ? 0.f : h[0], h[1] < 0.f ? 0.f : h[1]};               // relu
                        }
                        f16x2 hi, lo;
                        split_pair2(v[0], v[1], hi, lo);
                        wreg[(i * 16 + fq * 4 + rp) * 72 + j * 16 + frow] = part == 0 ? hi[0] : lo[0];
                        wreg[(i * 16 + fq * 4 + rp + 1) * 72 + j * 16 + frow] = part == 0 ? hi[1] : lo[1];
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int lr = it * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4 *>(wreg + lr * 72 + ch * 8);
                if constexpr (EPI == GE_S_BIAS_RELU_PAIR) {
                    if (n0 + wn * 64 + ch * 8 < g.pair_c)
                        *reinterpret_cast<uint4 *>(out + (int64_t)(m0 + wm * 64 + lr) * g.ldo + part * g.pair_c + n0 + wn * 64 + ch * 8) = v;
                } else {
                    *reinterpret_cast<uint4 *>(out + (int64_t)(m0 + wm * 64 + lr) * g.ldo + part * g.N + n0 + wn * 64 + ch * 8) = v;
                }
            }
        }
    } else if constexpr (EPI == GE_BIAS_ADD_RELU) {
        // out fp16 = relu(acc + bias + identity): the sum stays fp32 through the patch and is rounded ONCE (the
        // arithmetic of conv_f16.hip's epilogue, so either kernel may serve a layer); 8-byte pieces per lane
        __syncthreads();
        typedef _Float16 h4v __attribute__((ext_vector_type(4)));
        float *wreg = reinterpret_cast<float *>(smem) + wave * (32 * 68);
        _Float16 *outh = reinterpret_cast<_Float16 *>(g.out);
        const int nbase = n0 + wn * 64, c4 = (lane & 15) * 4;
        const float4 bias4 = *reinterpret_cast<const float4 *>(g.bias + nbase + c4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        wreg[(ii * 16 + fq * 4 + r) * 68 + j * 16 + frow] = acc[half * 2 + ii][j][r];
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int lr = it * 4 + (lane >> 4);
                const int64_t off = (int64_t)(m0 + wm * 64 + half * 32 + lr) * g.ldo + nbase + c4;
                const float4 a = *reinterpret_cast<const float4 *>(wreg + lr * 68 + c4);
                const h4v idv = *reinterpret_cast<const h4v *>(g.identity + off);
                float v[4] = {a.x + bias4.x, a.y + bias4.y, a.z + bias4.z, a.w + bias4.w};
                h4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] + (float)idv[e];
                    o[e] = (_Float16)(v[e] < 0.f ? 0.f : v[e]);
                }
                *reinterpret_cast<h4v *>(outh + off) = o;
            }
            __syncthreads();
        }
    } else if constexpr (EPI == GE_S_BIAS_F32 || EPI == GE_S_BIAS_RES || EPI == GE_S_BIAS_RES_PAIR) {
        // split mode, fp32 outputs: out = acc * oscale + bias, or x += that (RES_PAIR: and relu(x) as fp16 pairs); whole 256-byte row pieces, each lane four
        // consecutive columns (the layout of the persistent kernel's epilogues)
        __syncthreads();
        float *wreg = reinterpret_cast<float *>(smem) + wave * (32 * 68);
        float *outp = reinterpret_cast<float *>(g.out);
        const int nbase = n0 + wn * 64, c4 = (lane & 15) * 4;
        const float4 bias4 = *reinterpret_cast<const float4 *>(g.bias + nbase + c4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        wreg[(ii * 16 + fq * 4 + r) * 68 + j * 16 + frow] = acc[half * 2 + ii][j][r];
            __syncthreads();
            const int mbase = m0 + wm * 64 + half * 32;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int lr = it * 4 + (lane >> 4);
                const int64_t m = mbase + lr;
                const float4 a = *reinterpret_cast<const float4 *>(wreg + lr * 68 + c4);
                float *dst = outp + m * g.ldo + nbase + c4;
                float o[4];
                o[0] = fmaf(a.x, g.oscale, bias4.x);
                o[1] = fmaf(a.y, g.oscale, bias4.y);
                o[2] = fmaf(a.z, g.oscale, bias4.z);
                o[3] = fmaf(a.w, g.oscale, bias4.w);
                if (EPI == GE_S_BIAS_RES || EPI == GE_S_BIAS_RES_PAIR) {
                    float4 x = *reinterpret_cast<const float4 *>(dst);
                    if (g.relu_x) {   // the destination holds a pre-activation: its ReLU is applied here (see gemm_f16.h)
                        x.x = x.x < 0.f ? 0.f : x.x;
                        x.y = x.y < 0.f ? 0.f : x.y;
                        x.z = x.z < 0.f ? 0.f : x.z;
                        x.w = x.w < 0.f ? 0.f : x.w;
                    }
                    o[0] = x.x + o[0];
                    o[1] = x.y + o[1];
                    o[2] = x.z + o[2];
                    o[3] = x.w + o[3];
                }
                *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                if constexpr (EPI == GE_S_BIAS_RES_PAIR) {   // relu(x) as the pair operand of the next block's first convolution
                    typedef _Float16 h4p __attribute__((ext_vector_type(4)));
                    h4p hi, lo;
                    {
                        const float y[4] = {o[0] < 0.f ? 0.f : o[0], o[1] < 0.f ? 0.f : o[1], o[2] < 0.f ? 0.f : o[2],
                                            o[3] < 0.f ? 0.f : o[3]};
                        split_pairs<4>(y, hi, lo);
                    }
                    _Float16 *po = g.pair_out + m * 2 * g.pair_c + nbase + c4;
                    *reinterpret_cast<h4p *>(po) = hi;
                    *reinterpret_cast<h4p *>(po + g.pair_c) = lo;
                }
            }
            __syncthreads();
        }
    } else if constexpr (EPI == GE_BIAS_RES || EPI == GE_EUCLID || EPI == GE_F32) {
        // fp32 outputs: transpose each wave's 64x64 accumulator block through LDS in two 32-row
        // halves so that global accesses are whole 256-byte row pieces (16 B per lane when aligned)
        __syncthreads();
        float *wreg = reinterpret_cast<float *>(smem) + wave * (32 * 68);
        float *outp = reinterpret_cast<float *>(g.out);
        const bool vec_ok = (g.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(outp) & 15) == 0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        wreg[(ii * 16 + fq * 4 + r) * 68 + j * 16 + frow] = acc[half * 2 + ii][j][r];
            __syncthreads();
            const int mbase = m0 + wm * 64 + half * 32, nbase = n0 + wn * 64;
            if (vec_ok) {
                const int c4 = (lane & 15) * 4;
                float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), bn4 = bias4;
                if (EPI == GE_BIAS_RES) bias4 = *reinterpret_cast<const float4 *>(g.bias + nbase + c4);
                if (EPI == GE_EUCLID) {
                    bn4.x = (nbase + c4 + 0 < g.n_valid) ? g.aux2[nbase + c4 + 0] : 0.f;
                    bn4.y = (nbase + c4 + 1 < g.n_valid) ? g.aux2[nbase + c4 + 1] : 0.f;
                    bn4.z = (nbase + c4 + 2 < g.n_valid) ? g.aux2[nbase + c4 + 2] : 0.f;
                    bn4.w = (nbase + c4 + 3 < g.n_valid) ? g.aux2[nbase + c4 + 3] : 0.f;
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int lr = it * 4 + (lane >> 4);
                    const int m = mbase + lr;
                    const float4 a = *reinterpret_cast<const float4 *>(wreg + lr * 68 + c4);
                    float *dst = outp + (int64_t)m * g.ldo + nbase + c4;
                    if (EPI == GE_F32) {
                        *reinterpret_cast<float4 *>(dst) = a;
                    } else if (EPI == GE_BIAS_RES) {
                        float4 x = *reinterpret_cast<const float4 *>(dst);
                        x.x = x.x + (a.x + bias4.x);
                        x.y = x.y + (a.y + bias4.y);
                        x.z = x.z + (a.z + bias4.z);
                        x.w = x.w + (a.w + bias4.w);
                        *reinterpret_cast<float4 *>(dst) = x;
                    } else { // GE_EUCLID
                        if (m < g.m_valid) {
                            const float am = g.aux[m];
                            float4 o, s4 = make_float4(1.f, 1.f, 1.f, 1.f);
                            if (g.rscale) {
                                const float rs = g.rscale[m];
                                s4 = make_float4(rs * g.cscale[nbase + c4 + 0], rs * g.cscale[nbase + c4 + 1],
                                                 rs * g.cscale[nbase + c4 + 2], rs * g.cscale[nbase + c4 + 3]);
                            }
                            o.x = fmaf(-2.0f, a.x * s4.x, am + bn4.x);
                            o.y = fmaf(-2.0f, a.y * s4.y, am + bn4.y);
                            o.z = fmaf(-2.0f, a.z * s4.z, am + bn4.z);
                            o.w = fmaf(-2.0f, a.w * s4.w, am + bn4.w);
                            if (nbase + c4 + 3 < g.n_valid) {
                                *reinterpret_cast<float4 *>(dst) = o;
                            } else {
                                if (nbase + c4 + 0 < g.n_valid) dst[0] = o.x;
                                if (nbase + c4 + 1 < g.n_valid) dst[1] = o.y;
                                if (nbase + c4 + 2 < g.n_valid) dst[2] = o.z;
                            }
                        }
                    }
                }
            } else {
                // unaligned rows (e.g. ng = 15913): one full 256-byte row per wave instruction
                const int n = nbase + lane;
                float bias = 0.f, bnv = 0.f;
                if (EPI == GE_BIAS_RES) bias = g.bias[n];
                if (EPI == GE_EUCLID) bnv = (n < g.n_valid) ? g.aux2[n] : 0.f;
#pragma unroll 4
                for (int lr = 0; lr < 32; ++lr) {
                    const int m = mbase + lr;
                    const float a = wreg[lr * 68 + lane];
                    float *dst = outp + (int64_t)m * g.ldo + n;
                    if (EPI == GE_F32) {
                        *dst = a;
                    } else if (EPI == GE_BIAS_RES) {
                        *dst = *dst + (a + bias);
                    } else if (m < g.m_valid && n < g.n_valid) {
                        const float sc = g.rscale ? g.rscale[m] * g.cscale[n] : 1.0f;
                        *dst = fmaf(-2.0f, a * sc, g.aux[m] + bnv);
                    }
                }
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + frow;
            float bnv = 0.f;
            if (EPI == GE_COSINE) bnv = (n < g.n_valid) ? g.aux2[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + fq * 4 + r;
                    const float a = (EPI == GE_S_PATCH) ? acc[i][j][r] * g.oscale : acc[i][j][r];
                    if (EPI == GE_PATCH || EPI == GE_S_PATCH) {
                        if (m < g.m_valid) {
                            const int b = m / g.P, p = m - b * g.P;
                            reinterpret_cast<float *>(g.out)[((int64_t)b * g.L + 1 + p) * g.ldo + n] =
                                a + g.aux[(int64_t)(1 + p) * g.N + n];
                        }
                    } else if (EPI == GE_COSINE) {
                        if (m < g.m_valid && n < g.n_valid) {
                            const float sc = g.rscale ? g.rscale[m] * g.cscale[n] : 1.0f;
                            float c = (a * sc) * __fdiv_rn(1.0f, g.aux[m] * bnv);
                            const float lo = (float)(-1.0 + 0.00001), hi = (float)(1.0 - 0.00001);
                            c = c < lo ? lo : (c > hi ? hi : c);
                            reinterpret_cast<float *>(g.out)[(int64_t)m * g.ldo + n] = acosf(c);
                        }
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// "big" kernel: 256 x 256 tile, 8 waves (2 x 4, 128 x 64 per wave), K consumed in 32-wide stages
// held in a 4-slot LDS ring (4 x 32 KB).  LDS-DMA runs 2-3 stages ahead and is never drained inside
// the loop: counted s_waitcnt vmcnt(4) + raw s_barrier per stage (cdna_hip_programming.md §5
// "Pipelining across barriers").  MFMA operand fragments are double-buffered in registers, so the
// ds_read_b128 of stage t+1 are issued before the 32 MFMAs of stage t.  One workgroup per CU.
//
// LDS stage image: A part [256 rows][64 B], then B part [256 rows][64 B]; one DMA instruction
// writes 16 rows x 64 B.  Bank swizzle for 64-byte rows read by ds_read_b128 (four non-contiguous
// 16-lane groups): 16-byte chunk index ^= 3 * ((row >> 3) & 1) -- conflict-free for all groups;
// applied to the DMA source address and to the read address (rule 21).
// ---------------------------------------------------------------------------------------------
constexpr int BBM = 256, BBN = 256, BBK = 32, B_NSTAGE = 4;
constexpr int B_PART_BYTES = BBM * BBK * 2;          // 16 KB
constexpr int B_STAGE_BYTES = 2 * B_PART_BYTES;      // 32 KB
constexpr int B_LDS_BYTES = B_NSTAGE * B_STAGE_BYTES; // 128 KB ring
constexpr int B_LDS_TOTAL = B_LDS_BYTES + 8 * 4096;   // + 32 KB of epilogue patches = the CU's whole 160 KB

// DBG (timing-only ablation builds, never used by the product path): 1 no DMA in the steady loop,
// 2 no MFMA, 4 no fragment reads in the steady loop, 8 no epilogue
template <int EPI, int DBG = 0, bool SYM = false>   // SYM (GE_EUCLID only): the instance with the mirrored stores, see SYM_STORE
__global__ __launch_bounds__(512) void gemm_f16_big_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int K = g.K;
    // split precision mode (GE_S_*): rows are [hi | lo], K = 2 * kseg halfs; the stages walk hi.hi', lo.hi', hi.lo'
    constexpr bool SPLIT = gemm_epi_is_split(EPI);
    // epilogues that know symmetric problems (GemmArgs::sym: A == W, square): only tiles on or above the diagonal are
    // computed; GE_CAND tests the mirrored pair, GE_EUCLID stores every off-diagonal tile a second time, transposed
    // (the mirrored stores live in their own instance, <GE_EUCLID, 0, true>, so the plain distance kernel carries none of
    // its registers)
    constexpr bool SYM_STORE = (EPI == GE_EUCLID) && SYM;
    constexpr bool SYM_EPI = (EPI == GE_CAND) || SYM_STORE;
    const int nseg = SPLIT ? g.kseg / BBK : 0;
    const int nst = SPLIT ? 3 * nseg : K / BBK; // even (K, kseg are multiples of 64)
    // SPLIT stage order is k-block major: stage 3r + s works on the 32-wide k block r with
    //   s = 0: hi . hi'      s = 1: lo . hi'      s = 2: hi . lo'
    // so consecutive stages SHARE an operand: stage 3r+1 keeps the W fragments (hi') of stage 3r in registers and stage
    // 3r+2 the A fragments (hi) -- their LDS parts are neither staged nor read a second time: 4 operand parts of 16 KB per
    // k block instead of 6 (one third less L2 -> LDS traffic and one third fewer fragment reads per matrix instruction).
    // k offset of a stage's A / W part inside a row [hi(kseg) | lo(kseg)], -1 = the stage has no such part:
    auto koff_a = [&](int st) -> int {
        if constexpr (!SPLIT) return st * BBK;
        const int r = st / 3, sidx = st - 3 * r;
        return sidx == 0 ? r * BBK : (sidx == 1 ? g.kseg + r * BBK : -1);
    };
    auto koff_b = [&](int st) -> int {
        if constexpr (!SPLIT) return st * BBK;
        const int r = st / 3, sidx = st - 3 * r;
        return sidx == 0 ? r * BBK : (sidx == 2 ? g.kseg + r * BBK : -1);
    };

    // PERSISTENT: gridDim.x workgroups (one per CU) walk the tiles.  L2 is private to an XCD, so each XCD
    // (blockIdx % 8 under round-robin dispatch; locality only, never correctness) OWNS whole groups of 8
    // tile rows and sweeps all their tile columns back to back: the group's A panel (8 x 256 rows) stays in
    // that XCD's L2 while the weight panels stream past it.  (The plain strided walk re-fetched A 4x: 442 MB
    // FETCH_SIZE vs 105 MB algorithmic on the FC1 shape.)
    const int ntiles = tiles_m * tiles_n;
    const int nb = (int)gridDim.x;
    // (only when the groups divide evenly among the eight XCDs: with tiles_m = 16 -- the 4000 x 16 000 distance rows of
    // the re-ranking -- two XCDs owned everything and the other six idled: 1.12 ms instead of 0.4)
    // g.walk: 0 = groups of 8 tile rows, row-fastest inside a group; 1 = column-fastest inside a group; 3 / 4 = groups of
    // 4 / 16 tile rows, row-fastest (32 CUs then cover 4 x 8 or 16 x 2 tiles at a time); | 8 = groups of 4 rows
    const int GR = ((g.walk & 7) == 3 || (g.walk & 8)) ? 4 : ((g.walk & 7) == 4 ? 16 : 8);
    const bool colfast = (g.walk & 7) == 1;
    // RAGGED row counts (round 6): when the groups do not divide among the XCDs but there are at least three per XCD, the
    // XCDs still own whole groups -- one XCD gets one group less, the last group may be short (its missing rows are unused
    // slots) -- instead of falling to the blocked walk below, whose 8 x 4 blocks leave a quarter of the slots idle on a
    // 3- or 9-column output (patch embedding, M = 254 tile rows: 0.298 -> 0.23 ms; the last, shorter encode group).
    // Symmetric problems keep the exact rule (their block lists are balanced by construction).
    const int ngroups = (tiles_m + GR - 1) / GR;
    const bool owned = (nb % 8 == 0) && (tiles_m % (8 * GR) == 0 || (ngroups >= 24 && !(g.walk & 16) && !(SYM_EPI && g.sym)));
    // Any other shape on a full grid (the distance GEMMs: 79 x 79 tiles at N = 20 000): "blocked" walk.  In round r
    // XCD x works on block r*8 + x of the tile grid cut into blocks of 8 tile rows x (CUs per XCD / 8) tile columns
    // (8 x 4 on 256 CUs), one tile per CU: 12 operand panels feed 32 tiles.  (The plain grouped walk gave an XCD ONE
    // tile row x 32 columns per round: 33 panels per 32 tiles, every B panel fetched by all eight L2s -- measured
    // 2.66 GB of fabric reads against 62 MB of operands on the 20k x 20k x 768 distance GEMM.)  Edge blocks have
    // unused slots, which are skipped.
    const bool blocked = !owned && (nb % 64 == 0) && ntiles >= nb;
    const int xcd = (int)(blockIdx.x & 7), per_xcd = nb >> 3, per_group = GR * tiles_n;
    const int bc_w = per_xcd >> 3;                                   // tile columns per block (blocked mode)
    const int nbc = blocked ? (tiles_n + bc_w - 1) / bc_w : 1, nbr = (tiles_m + 7) >> 3;
    // SYM_STORE on the blocked walk: COMPACTED.  The blocks that straddle the diagonal (and the edge blocks) have unused
    // slots; a CU tied to one slot of every block then gets 11-14 tiles at N = 20 000 (79 x 79 tiles) where the mean is 12.3.
    // Instead the XCD's blocks form one list of VALID tiles (block order, slot order inside a block) and the CU in slot s
    // takes items s, s + 32, s + 64, ...: 12 or 13 tiles each, and 32 consecutive items still share one or two blocks'
    // operand panels.  (GE_CAND too: its candidate lists are appended through atomics and sorted by the refinement, so
    // the order in which tiles are visited never reaches a result.)
    const bool symc = SYM_EPI && blocked && g.sym != 0;
    int pos = (owned || symc) ? (int)(blockIdx.x >> 3) : (blocked ? 0 : (int)blockIdx.x); // position in this CU's tile list
    const int pos_step = (owned || symc) ? per_xcd : (blocked ? 1 : nb);
    int sy_p = 0, sy_cum = 0;   // symc: cursor into the XCD's block list and the number of valid tiles before it
    auto sym_tile_at = [&](int k) -> int {   // k-th valid tile of this XCD's list (k never decreases), -1 when exhausted
        for (;;) {
            int b = sy_p * 8 + xcd, br = 0, bcol = 0;
            for (;; ++br) {   // block row br holds the block columns from ceil((8 br - bc_w + 1) / bc_w) on
                if (br >= nbr) return -1;
                int fb = (br * 8 - bc_w + 1 + bc_w - 1) / bc_w;
                fb = fb < 0 ? 0 : fb;
                const int cnt = nbc - fb;
                if (cnt > 0 && b < cnt) {
                    bcol = fb + b;
                    break;
                }
                b -= cnt > 0 ? cnt : 0;
            }
            const int c_lo = bcol * bc_w, c_hi = min(tiles_n, c_lo + bc_w);
            int cnt = 0;
            for (int r = 0; r < 8; ++r) {
                const int tm = br * 8 + r;
                if (tm < tiles_m) cnt += max(0, c_hi - max(tm, c_lo));
            }
            if (k < sy_cum + cnt) {
                int want = k - sy_cum;
                for (int sl = 0; sl < 8 * bc_w; ++sl) {
                    const int tm = br * 8 + (sl & 7), tn = c_lo + (sl >> 3);
                    if (tm < tiles_m && tn < tiles_n && tn >= tm && want-- == 0) return tm * tiles_n + tn;
                }
            }
            sy_cum += cnt;
            ++sy_p;
        }
    };
    auto tile_at_raw = [&](int p) -> int { // -1 when the list is exhausted, -2 for an unused slot of an edge block
        if (blocked) {
            int b = p * 8 + xcd, br, bcol;
            if (SYM_EPI && g.sym) {
                // symmetric problem: only blocks with a tile on or above the diagonal are enumerated (block row br
                // starts at block column ceil((8 br - bc_w + 1) / bc_w)), so the eight XCDs get the same number of
                // blocks to within one
                br = 0;
                for (;; ++br) {
                    if (br >= nbr) return -1;
                    int fb = (br * 8 - bc_w + 1 + bc_w - 1) / bc_w;
                    fb = fb < 0 ? 0 : fb;
                    const int cnt = nbc - fb;
                    if (cnt > 0 && b < cnt) {
                        bcol = fb + b;
                        break;
                    }
                    b -= cnt > 0 ? cnt : 0;
                }
            } else {
                br = b / nbc;
                bcol = b - br * nbc;
            }
            if (br >= nbr) return -1;
            const int slot = (int)(blockIdx.x >> 3);
            const int tm = br * 8 + (slot & 7), tn = bcol * bc_w + (slot >> 3);
            if (SYM_EPI && g.sym && tn < tm) return -2;   // symmetric problem: upper-triangular tiles only
            return (tm < tiles_m && tn < tiles_n) ? tm * tiles_n + tn : -2;
        }
        if (!owned) {
            if (p >= ntiles) return -1;
            if (SYM_EPI && g.sym) {
                int tm, tn;
                tile_coords((unsigned)p, tiles_m, tiles_n, 8, tm, tn);
                if (tn < tm) return -2;
            }
            return p;
        }
        const int grp = xcd + 8 * (p / per_group);
        if (grp >= ngroups) return -1;
        const int within = p % per_group;
        // order inside a group of 8 tile rows: row fastest (32 CUs = 8 rows x 4 columns at a time), or -- g.walk, narrow
        // outputs -- column fastest: with tiles_n = 3 (N = 768: out-proj, FC2) 32 slots then cover ~11 whole tile rows, each A
        // panel read by its three CUs at the same time, instead of 8 rows x 3 columns plus one column of the next group
        // whose A panels come back a round later
        const int wrow = colfast ? within / tiles_n : within % GR, wcol = colfast ? within % tiles_n : within / GR;
        if (grp * GR + wrow >= tiles_m) return -2;   // a short last group
        if (SYM_EPI && g.sym && wcol < grp * GR + wrow) return -2;
        return (grp * GR + wrow) * tiles_n + wcol; // row-major tile id
    };
    auto tile_at = [&](int &p) -> int {    // advances p past unused slots
        if constexpr (SYM_EPI) {
            if (symc) return sym_tile_at(p);
        }
        int t = tile_at_raw(p);
        while (t == -2) {
            p += pos_step;
            t = tile_at_raw(p);
        }
        return t;
    };

    // ---- DMA: wave w moves rows [32w, 32w+32) of the A part and of the B part of every stage ----
    const int drow = lane >> 2;                                     // row inside a 16-row DMA piece
    const int dchunk = (lane & 3) ^ (((drow >> 3) & 1) * 3);        // source chunk for LDS slot lane&3
    const _Float16 *a_src = nullptr, *b_src = nullptr;
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int tile) {
        int tm, tn;
        if (owned || blocked) {
            tm = tile / tiles_n;
            tn = tile - tm * tiles_n;
        } else {
            tile_coords((unsigned)tile, tiles_m, tiles_n, 8, tm, tn);
        }
        m0 = tm * BBM;
        n0 = tn * BBN;
        // (ablation DBG 64, wrong results: every tile reads operand panels 0-1 -- 2 x 2 panels stay resident in each XCD's
        // L2, so nothing is fetched from beyond it: the upper bound of what a better tile order could win)
        const int ma = (DBG == 64) ? (tm & 1) * BBM : m0, na = (DBG == 64) ? (tn & 1) * BBN : n0;
        a_src = g.A + (int64_t)(ma + wave * 32 + drow) * K + dchunk * 8;
        b_src = g.W + (int64_t)(na + wave * 32 + drow) * K + dchunk * 8;
    };
    auto dma_stage = [&](int st) {
        unsigned char *abase = smem + (st & (B_NSTAGE - 1)) * B_STAGE_BYTES + wave * 2048;
        unsigned char *bbase = abase + B_PART_BYTES;
        const int koff = koff_a(st), koffb = koff_b(st);
        if (!SPLIT || koff >= 0) {
            dma16(a_src + koff, abase);
            dma16(a_src + (int64_t)16 * K + koff, abase + 1024);
        }
        if (!SPLIT || koffb >= 0) {
            dma16(b_src + koffb, bbase);
            dma16(b_src + (int64_t)16 * K + koffb, bbase + 1024);
        }
    };
    // PAIRED DMA (non-split instances; DBG bit 16 = the unpaired schedule of rounds 1-2, for A/B timing).  A stage is a
    // 64-byte k slab of every row = HALF a 128-byte line; fetched one stage at a time, every line crosses the L1 -> L2
    // path twice, ~0.5 us apart -- by then the first fill has left the 32 KB L1.  Measured on the DMA path alone
    // (20k x 20k x 768, matrix instructions and epilogue compiled out): 12 TB/s with half-line requests, 18 TB/s with
    // whole lines.  The stages 2D and 2D+1 (the two halves of the same lines) are therefore requested back to back,
    // piece by piece, so that the second request meets the first one's fill: odd steps issue two stages, even steps none
    // (see step_steady); the prologue fills all four slots.
    constexpr bool PAIR = !SPLIT && !(DBG & 16);
    auto dma_pair = [&](int st) {   // stages st (even) and st + 1, pieces interleaved
        unsigned char *a0 = smem + (st & (B_NSTAGE - 1)) * B_STAGE_BYTES + wave * 2048;
        unsigned char *a1 = smem + ((st + 1) & (B_NSTAGE - 1)) * B_STAGE_BYTES + wave * 2048;
        const int k0 = st * BBK, k1 = k0 + BBK;
        dma16(a_src + k0, a0);
        dma16(a_src + k1, a1);
        dma16(a_src + (int64_t)16 * K + k0, a0 + 1024);
        dma16(a_src + (int64_t)16 * K + k1, a1 + 1024);
        dma16(b_src + k0, a0 + B_PART_BYTES);
        dma16(b_src + k1, a1 + B_PART_BYTES);
        dma16(b_src + (int64_t)16 * K + k0, a0 + B_PART_BYTES + 1024);
        dma16(b_src + (int64_t)16 * K + k1, a1 + B_PART_BYTES + 1024);
    };
    auto dma_prologue = [&]() {
        if constexpr (PAIR) {
            dma_pair(0);               // nst is even and >= 2
            if (2 < nst) dma_pair(2);
        } else {
            dma_stage(0);
            if (1 < nst) dma_stage(1);
            if (2 < nst) dma_stage(2);
        }
    };

    // ---- fragment addresses ----
    const int frow = lane & 15, fq = lane >> 4;
    const int fsw = (fq ^ (((lane >> 3) & 1) * 3)) << 4;
    const int a_off = (wr * 128 + frow) * 64 + fsw;                 // + i * 1024
    const int b_off = B_PART_BYTES + (wc * 64 + frow) * 64 + fsw;   // + j * 1024
    auto load_frags = [&](int st, f16x8 (&fa)[8], f16x8 (&fb)[4]) {
        const unsigned char *sb = smem + (st & (B_NSTAGE - 1)) * B_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const f16x8 *>(sb + b_off + j * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const f16x8 *>(sb + a_off + i * 1024);
    };

    // GE_CAND: wave-private candidate list in the 4 KB patch above the ring (SoA: row, col, d), kept across tiles and
    // flushed to the global per-row lists only when it fills up (and at the end): a tile's epilogue then touches
    // no global memory at all, so the next tile's first DMA wait does not sit behind store / atomic round trips
    constexpr int CL_CAP = 336;
    int cl_n = 0;
    unsigned *cl_row = reinterpret_cast<unsigned *>(smem + B_LDS_BYTES + wave * 4096);
    unsigned *cl_col = cl_row + CL_CAP;
    unsigned *cl_d = cl_col + CL_CAP;
    auto cl_flush = [&]() {
        if constexpr (EPI == GE_CAND) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (tried: all slot reservations of a flush issued before the first is consumed -- fewer round trips, but
            // the 30 extra live registers spill the accumulators and the replicated body triples the kernel's ISA:
            // 727 us instead of 540 at N = 20 000)
            for (int e = lane; e < cl_n; e += 64) {
                const unsigned rw = cl_row[e], cw = cl_col[e], dv = cl_d[e];
                const unsigned fl = rw >> 28, row = rw & 0x0fffffffu;
                if (fl & 1u) {
                    const unsigned p = atomicAdd(&g.cnt_lo[row], 1u);
                    if (p < (unsigned)g.cap_lo) g.list_lo[(size_t)row * g.cap_lo + p] = make_uint2(cw, dv);
                }
                if (fl & 2u) {
                    const unsigned p = atomicAdd(&g.cnt_hi[row], 1u);
                    if (p < (unsigned)g.cap_hi) g.list_hi[(size_t)row * g.cap_hi + p] = make_uint2(cw, dv);
                }
                if (fl & 4u) {
                    const unsigned p = atomicAdd(&g.cnt_lo[cw], 1u);
                    if (p < (unsigned)g.cap_lo) g.list_lo[(size_t)cw * g.cap_lo + p] = make_uint2(row, dv);
                }
                if (fl & 8u) {
                    const unsigned p = atomicAdd(&g.cnt_hi[cw], 1u);
                    if (p < (unsigned)g.cap_hi) g.list_hi[(size_t)cw * g.cap_hi + p] = make_uint2(row, dv);
                }
            }
            __builtin_amdgcn_wave_barrier();
            cl_n = 0;
        }
    };

    int tile = tile_at(pos);
    if (tile < 0) return;
    if (g.stagger > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = g.stagger_mode ? (unsigned long long)g.stagger * ((blockIdx.x >> 3) & 7u) / 8u
                                                       : (unsigned long long)g.stagger * (blockIdx.x & 255u) / 256u;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
    set_tile(tile);
    dma_prologue();
    [[maybe_unused]] int stamp_tile = 0;
    for (;;) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // stage 0 of this tile has landed everywhere (its DMA was issued before the previous tile's last
    // stores drained, or at kernel start); in-order vmcnt also retires every older store
    if constexpr (SPLIT) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");   // stages 1 (2 pieces) + 2 (2) may be in flight
    else if constexpr (PAIR) {   // the pair (0, 1) has landed, the pair (2, 3) may be in flight
        if (nst > 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (nst > 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");

    auto stamp = [&](int which) {
        if constexpr ((DBG & 32) != 0) {
            if (g.stamps && tid == 0 && stamp_tile < 32)
                g.stamps[((size_t)blockIdx.x * 32 + stamp_tile) * 8 + which] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    f16x8 fa0[8], fb0[4], fa1[8], fb1[4];
    load_frags(0, fa0, fb0);

    // one pipeline step: stage t is in (fa, fb); make stage t+1 visible, start the DMA of stage t+3
    // into the slot stage t-1 used, fetch the fragments of stage t+1, then 32 MFMAs on stage t
    auto mfma32 = [&](f16x8 (&fa)[8], f16x8 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    // steady state (no conditions, one basic block): the 4 DMA pieces and the 12 fragment reads are
    // spread between the MFMAs (1 DMA + 3 ds_read per 8 MFMA) so that the CU's load path works
    // beside the matrix pipe instead of in a burst after the barrier
    // PAIR: even steps (ODD_ = 0) wait for the pair that ends with stage t+1 (the next pair, 8 pieces, stays in flight) and
    // issue nothing; odd steps wait for stage t+1 = the first of its pair (pieces are interleaved: only the last piece
    // of stage t+2 may still fly), then issue stages t+3 | t+4 into the slots of stages t-1 and t.  Stage t's fragments were
    // read in step t-1: lgkmcnt(0) before the barrier makes every wave's reads of that slot complete before any wave's
    // DMA can land in it.
    auto step_steady = [&](auto ODD_, int t, f16x8 (&fa)[8], f16x8 (&fb)[4], f16x8 (&na)[8], f16x8 (&nb)[4]) {
        constexpr bool ODD = decltype(ODD_)::value != 0;
        constexpr bool ISSUE = !PAIR || ODD;   // this step issues DMA
        if constexpr (!PAIR) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else if constexpr (ODD) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        unsigned char *dbase = smem + ((t + 3) & (B_NSTAGE - 1)) * B_STAGE_BYTES + wave * 2048;
        unsigned char *dbase2 = smem + ((t + 4) & (B_NSTAGE - 1)) * B_STAGE_BYTES + wave * 2048;   // PAIR only
        const unsigned char *sb = smem + ((t + 1) & (B_NSTAGE - 1)) * B_STAGE_BYTES;
        const int koff = koff_a(t + 3), koffb = koff_b(t + 3);
#define MPREID_FRAG_B(j) nb[j] = *reinterpret_cast<const f16x8 *>(sb + b_off + (j) * 1024)
#define MPREID_FRAG_A(i) na[i] = *reinterpret_cast<const f16x8 *>(sb + a_off + (i) * 1024)
#define MPREID_MFMA_ROWS(i0)                                                                             \
    _Pragma("unroll") for (int ii = (i0); ii < (i0) + 2; ++ii) _Pragma("unroll") for (int j = 0; j < 4; ++j) { \
        if constexpr (!(DBG & 2))                                                                        \
            acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ii], fb[j], acc[ii][j], 0, 0, 0);       \
        else                                                                                             \
            asm volatile("" ::"v"(fa[ii]), "v"(fb[j]));                                                  \
    }
        // group 0
        if constexpr (!(DBG & 1) && ISSUE) {
            dma16(a_src + koff, dbase);
            if constexpr (PAIR) dma16(a_src + koff + BBK, dbase2);
        }
        if constexpr (!(DBG & 4)) { MPREID_FRAG_B(0); MPREID_FRAG_B(1); MPREID_FRAG_B(2); }
        MPREID_MFMA_ROWS(0);
        __builtin_amdgcn_sched_barrier(0);
        // group 1
        if constexpr (!(DBG & 1) && ISSUE) {
            dma16(a_src + (int64_t)16 * K + koff, dbase + 1024);
            if constexpr (PAIR) dma16(a_src + (int64_t)16 * K + koff + BBK, dbase2 + 1024);
        }
        if constexpr (!(DBG & 4)) { MPREID_FRAG_B(3); MPREID_FRAG_A(0); MPREID_FRAG_A(1); }
        MPREID_MFMA_ROWS(2);
        __builtin_amdgcn_sched_barrier(0);
        // group 2
        if constexpr (!(DBG & 1) && ISSUE) {
            dma16(b_src + koffb, dbase + B_PART_BYTES);
            if constexpr (PAIR) dma16(b_src + koffb + BBK, dbase2 + B_PART_BYTES);
        }
        if constexpr (!(DBG & 4)) { MPREID_FRAG_A(2); MPREID_FRAG_A(3); MPREID_FRAG_A(4); }
        MPREID_MFMA_ROWS(4);
        __builtin_amdgcn_sched_barrier(0);
        // group 3
        if constexpr (!(DBG & 1) && ISSUE) {
            dma16(b_src + (int64_t)16 * K + koffb, dbase + B_PART_BYTES + 1024);
            if constexpr (PAIR) dma16(b_src + (int64_t)16 * K + koffb + BBK, dbase2 + B_PART_BYTES + 1024);
        }
        if constexpr (!(DBG & 4)) { MPREID_FRAG_A(5); MPREID_FRAG_A(6); MPREID_FRAG_A(7); }
        MPREID_MFMA_ROWS(6);
        __builtin_amdgcn_sched_barrier(0);
#undef MPREID_FRAG_A
#undef MPREID_FRAG_B
#undef MPREID_MFMA_ROWS
    };
    auto step_tail = [&](int t, f16x8 (&fa)[8], f16x8 (&fb)[4], f16x8 (&na)[8], f16x8 (&nb)[4]) {
        if (t + 1 < nst) {
            if constexpr (PAIR) {
                // every stage has been issued (by the prologue or the steady loop's last odd step): an even step with a
                // whole pair behind stage t+1 lets that pair fly, everything else drains
                if (!(t & 1) && t + 3 < nst) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                if (t + 2 < nst) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (t + 3 < nst) dma_stage(t + 3);
            }
            load_frags(t + 1, na, nb);
        }
        mfma32(fa, fb);
    };
    // (tried in round 2: waves 4-7 half a stage late -- MI355X_MICROARCH.md "Two waves per SIMD" item 9 -- by moving
    // their MFMA groups across the barrier: rows 4-7 of stage t after barrier t, rows 0-3 of stage t+1 before barrier
    // t+1; LDS traffic unchanged, same bits.  Both formulations that hipcc accepts -- two copies of the loop selected
    // by the wave number, or one copy with a wave-uniform branch per MFMA group -- make the register allocator spill
    // 300-1000 VGPRs around the accumulators; it needs an assembly k-loop.)
    if constexpr (SPLIT) {
        // Six steps = two k blocks form the period of the register roles (S = t mod 6):
        //   S   stage     uses        meanwhile fetches (fragments of stage t+1)
        //   0   hi.hi'    fa0, fb0    fa1 <- lo
        //   1   lo.hi'    fa1, fb0    fb1 <- lo'
        //   2   hi.lo'    fa0, fb1    fa1 <- hi, fb0 <- hi'   (next k block)
        //   3   hi.hi'    fa1, fb0    fa0 <- lo
        //   4   lo.hi'    fa0, fb0    fb1 <- lo'
        //   5   hi.lo'    fa1, fb1    fa0 <- hi, fb0 <- hi'
        // DMA of stage t+3 (the parts it has) goes out in step t; the wait at the start of step t lets the pieces of stage
        // t+2 stay in flight (4 when that stage has both parts, else 2).
        auto sstep = [&](auto S_, auto TAIL_, int t) {
            constexpr int S = decltype(S_)::value;
            constexpr bool TAIL = decltype(TAIL_)::value;
            constexpr int s0 = S % 3;                     // this stage: 0 hi.hi', 1 lo.hi', 2 hi.lo'
            constexpr bool ldA = s0 != 1, ldB = s0 != 0;  // stage t+1 has an A part unless it is hi.lo' (s0+1 == 2); a W part unless lo.hi'
            constexpr bool dmA = s0 != 2, dmB = s0 != 1;  // stage t+3 has the same kind as stage t
            f16x8 (&fa)[8] = (S & 1) ? fa1 : fa0;
            f16x8 (&fb)[4] = (s0 == 2) ? fb1 : fb0;
            f16x8 (&na)[8] = (S < 3) ? fa1 : fa0;
            f16x8 (&nb)[4] = (s0 == 1) ? fb1 : fb0;
            // the tail group is the last six stages (nst is a multiple of six: kseg % 64 == 0), so what is left to fetch is
            // known at compile time: stage t+1 exists for S < 5, stage t+2 for S < 4, stage t+3 for S < 3
            constexpr bool more1 = !TAIL || S < 5, more3 = !TAIL || S < 3;
            if constexpr (more1) {
                if constexpr (TAIL && S >= 4) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                else if constexpr (s0 == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");   // stage t+2 is hi.hi': 4 pieces
                else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            }
            unsigned char *dbase = smem + ((t + 3) & (B_NSTAGE - 1)) * B_STAGE_BYTES + wave * 2048;
            const unsigned char *sb = smem + ((t + 1) & (B_NSTAGE - 1)) * B_STAGE_BYTES;
            const int r3 = (t + 3) / 3;
            const int koff = (s0 == 1 ? g.kseg : 0) + r3 * BBK, koffb = (s0 == 2 ? g.kseg : 0) + r3 * BBK;
            auto piece = [&](int q) {   // q-th DMA piece of stage t+3 (A pieces first)
                if (!more3) return;
                if (dmA && q < 2) dma16(a_src + (int64_t)(q * 16) * K + koff, dbase + q * 1024);
                else if (dmB && q - (dmA ? 2 : 0) >= 0 && q - (dmA ? 2 : 0) < 2)
                    dma16(b_src + (int64_t)((q - (dmA ? 2 : 0)) * 16) * K + koffb, dbase + B_PART_BYTES + (q - (dmA ? 2 : 0)) * 1024);
            };
            auto rd_a = [&](int i) { if (more1) na[i] = *reinterpret_cast<const f16x8 *>(sb + a_off + i * 1024); };
            auto rd_b = [&](int j) { if (more1) nb[j] = *reinterpret_cast<const f16x8 *>(sb + b_off + j * 1024); };
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                piece(gq);
                if constexpr (ldA && ldB) {          // 12 reads: B0 B1 B2 | B3 A0 A1 | A2 A3 A4 | A5 A6 A7
                    if (gq == 0) { rd_b(0); rd_b(1); rd_b(2); }
                    if (gq == 1) { rd_b(3); rd_a(0); rd_a(1); }
                    if (gq == 2) { rd_a(2); rd_a(3); rd_a(4); }
                    if (gq == 3) { rd_a(5); rd_a(6); rd_a(7); }
                } else if constexpr (ldA) {          // 8 reads, two per group
                    rd_a(2 * gq);
                    rd_a(2 * gq + 1);
                } else {                             // 4 reads, one per group
                    rd_b(gq);
                }
#pragma unroll
                for (int ii = 2 * gq; ii < 2 * gq + 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ii], fb[j], acc[ii][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using std::integral_constant;
        int t = 0;
        for (; t + 6 < nst; t += 6) {
            sstep(integral_constant<int, 0>{}, integral_constant<bool, false>{}, t);
            sstep(integral_constant<int, 1>{}, integral_constant<bool, false>{}, t + 1);
            sstep(integral_constant<int, 2>{}, integral_constant<bool, false>{}, t + 2);
            sstep(integral_constant<int, 3>{}, integral_constant<bool, false>{}, t + 3);
            sstep(integral_constant<int, 4>{}, integral_constant<bool, false>{}, t + 4);
            sstep(integral_constant<int, 5>{}, integral_constant<bool, false>{}, t + 5);
        }
        sstep(integral_constant<int, 0>{}, integral_constant<bool, true>{}, t);
        sstep(integral_constant<int, 1>{}, integral_constant<bool, true>{}, t + 1);
        sstep(integral_constant<int, 2>{}, integral_constant<bool, true>{}, t + 2);
        sstep(integral_constant<int, 3>{}, integral_constant<bool, true>{}, t + 3);
        sstep(integral_constant<int, 4>{}, integral_constant<bool, true>{}, t + 4);
        sstep(integral_constant<int, 5>{}, integral_constant<bool, true>{}, t + 5);
    } else {
    int t = 0;
    for (; t + 4 < nst; t += 2) {
        step_steady(std::integral_constant<int, 0>{}, t, fa0, fb0, fa1, fb1);
        step_steady(std::integral_constant<int, 1>{}, t + 1, fa1, fb1, fa0, fb0);
    }
    for (; t < nst; t += 2) {
        step_tail(t, fa0, fb0, fa1, fb1);
        step_tail(t + 1, fa1, fb1, fa0, fb0);
    }
    }

    // ---- epilogue.  The accumulators leave through wave-private patches in the 32 KB of LDS above the
    // ring, so the ring itself is free for the next tile's first stages as soon as every wave has passed
    // this barrier ----
    __syncthreads();
    stamp(1);
    const int cur_m0 = m0, cur_n0 = n0;
    pos += pos_step;
    const int next_tile = tile_at(pos);
    unsigned char *patch = smem + B_LDS_BYTES;
    if constexpr (DBG & 8) {
        // keep EVERY accumulator live (rule 17: a skipped consumer lets the compiler delete the MFMAs too)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if constexpr (EPI == GE_CAND) {
        // Two phases per 16-row block i, both in the accumulator layout (lane: rows fq*4 + r, column frow of each
        // 16-column block j):
        //   A  straight-line: d for the lane's 16 elements, four comparisons each, a 16-bit hit mask per lane
        //      (~1.6 % of the elements pass; the block's values are also laid into a [16][64] LDS patch)
        //   B  while any lane still has a hit: every such lane takes its lowest one -- up to 64 appends per iteration
        //      -- re-reads the value from the patch (dynamic index), recomputes d and its flags, and appends
        //      (row, column, d, flags) to the wave's LDS candidate list.  Typically 1-2 iterations.
        // No global traffic unless the list fills up.
        const bool mirror = g.sym != 0 && cur_m0 != cur_n0;   // off-diagonal tile of a symmetric problem
        float *tp = reinterpret_cast<float *>(smem + wave * 16384);   // patch in the idle k-loop ring
        const int rb = cur_m0 + wr * 128, cb = cur_n0 + wc * 64;
        const float NEG = -__builtin_huge_valf(), POS = __builtin_huge_valf();
        // the wave's 128 rows x (norm, tlo, thi) and 64 columns x (norm, tlo, thi) are fetched ONCE per tile into a
        // wave-private LDS table (one global round trip instead of one per 16-row pass); arrays are padded to M
        // entries (0 / -inf / +inf past m_valid)
        float *rt = tp + 1024;                 // [3][128] rows, then [3][64] columns (2304 B, behind the 4 KB patch)
        float *ct = rt + 384;
        {
            const float a0 = g.aux[rb + lane], a1 = g.aux[rb + 64 + lane];
            const float l0 = g.tlo[rb + lane], l1 = g.tlo[rb + 64 + lane];
            const float h0 = g.thi[rb + lane], h1 = g.thi[rb + 64 + lane];
            const int c = cb + lane;
            const bool cv = c < g.n_valid;
            const float cn = cv ? g.aux2[c] : 0.f;
            const float cl = (mirror && cv) ? g.tlo[c] : NEG, chh = (mirror && cv) ? g.thi[c] : POS;
            rt[lane] = a0; rt[64 + lane] = a1;
            rt[128 + lane] = l0; rt[192 + lane] = l1;
            rt[256 + lane] = h0; rt[320 + lane] = h1;
            ct[lane] = cn; ct[64 + lane] = cl; ct[128 + lane] = chh;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float bnv[4], tlc[4], thc[4];
        unsigned cvm = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cl_ = j * 16 + frow;
            cvm |= (cb + cl_ < g.n_valid) ? (0xfu << (4 * j)) : 0u;
            bnv[j] = ct[cl_];
            tlc[j] = ct[64 + cl_];
            thc[j] = ct[128 + cl_];
        }
        // (tried: the eight passes rolled into one copy of the code, accumulators moved into place by a switch -- the
        // ISA shrinks from 6 k to 5.6 k lines but 40-80 VGPRs spill and the kernel is slower, 592 vs 540 us)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = i * 16 + fq * 4;
            const int r0 = rb + rl;
            const float4 an4 = *reinterpret_cast<const float4 *>(rt + rl);
            const float4 tl4 = *reinterpret_cast<const float4 *>(rt + 128 + rl);
            const float4 th4 = *reinterpret_cast<const float4 *>(rt + 256 + rl);
            const float anr[4] = {an4.x, an4.y, an4.z, an4.w}, tlr[4] = {tl4.x, tl4.y, tl4.z, tl4.w},
                        thr[4] = {th4.x, th4.y, th4.z, th4.w};
            unsigned rvm = 0u;   // rows inside the problem (the mirrored tests need it: a padded row has d = |g|^2)
#pragma unroll
            for (int r = 0; r < 4; ++r) rvm |= (r0 + r < g.m_valid) ? (0x1111u << r) : 0u;
            unsigned hit = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = acc[i][j][r];
                    tp[(fq * 4 + r) * 64 + j * 16 + frow] = a;
                    const float d = fmaf(-2.0f, a, anr[r] + bnv[j]);
                    const bool h = (d <= tlr[r]) | (d >= thr[r]) | (d <= tlc[j]) | (d >= thc[j]);
                    hit |= h ? (1u << (j * 4 + r)) : 0u;
                }
            hit &= cvm & rvm;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unsigned long long bal = __ballot(hit != 0u);
            if (g.stagger & 2) bal = 0;   // (timing experiment only: no appends)
            while (bal) {
                const int add = __popcll(bal);
                if (cl_n + add > CL_CAP) cl_flush();
                if (hit) {
                    const int e = __ffs((int)hit) - 1;
                    hit &= hit - 1u;
                    const int j = e >> 2, r = e & 3;
                    const float a = tp[(fq * 4 + r) * 64 + j * 16 + frow];
                    const float anx = r == 0 ? anr[0] : (r == 1 ? anr[1] : (r == 2 ? anr[2] : anr[3]));
                    const float tlx = r == 0 ? tlr[0] : (r == 1 ? tlr[1] : (r == 2 ? tlr[2] : tlr[3]));
                    const float thx = r == 0 ? thr[0] : (r == 1 ? thr[1] : (r == 2 ? thr[2] : thr[3]));
                    const float bnx = j == 0 ? bnv[0] : (j == 1 ? bnv[1] : (j == 2 ? bnv[2] : bnv[3]));
                    const float tcx = j == 0 ? tlc[0] : (j == 1 ? tlc[1] : (j == 2 ? tlc[2] : tlc[3]));
                    const float hcx = j == 0 ? thc[0] : (j == 1 ? thc[1] : (j == 2 ? thc[2] : thc[3]));
                    const float d = fmaf(-2.0f, a, anx + bnx);
                    const unsigned fl = (d <= tlx ? 1u : 0u) | (d >= thx ? 2u : 0u) | (d <= tcx ? 4u : 0u) | (d >= hcx ? 8u : 0u);
                    const int p = cl_n + __popcll(bal & ((1ull << lane) - 1ull));
                    cl_row[p] = (unsigned)(r0 + r) | (fl << 28);
                    cl_col[p] = (unsigned)(cb + j * 16 + frow);
                    cl_d[p] = __float_as_uint(d);
                }
                cl_n += add;
                bal = __ballot(hit != 0u);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (next_tile < 0) cl_flush();
        // the transposition patches alias stage slots the next tile's prologue is about to fill
        asm volatile("s_barrier" ::: "memory");
    } else if constexpr (EPI == GE_BIAS_F16 || EPI == GE_BIAS_GELU || EPI == GE_BIAS_RELU) {
        // one 16-row MFMA tile row per pass: patch [16][72] halfs (2304 B) -> two 16-byte row pieces per lane
        _Float16 *wreg = reinterpret_cast<_Float16 *>(patch + wave * 4096);
        _Float16 *out = reinterpret_cast<_Float16 *>(g.out);
        float bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = g.bias[cur_n0 + wc * 64 + j * 16 + frow];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bias[j];
                    if (EPI == GE_BIAS_GELU) v = quick_gelu(v);
                    if (EPI == GE_BIAS_RELU) v = v < 0.f ? 0.f : v;
                    wreg[(fq * 4 + r) * 72 + j * 16 + frow] = (_Float16)v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int lr = it * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4 *>(wreg + lr * 72 + ch * 8);
                _Float16 *dsth = out + (int64_t)(cur_m0 + wr * 128 + i * 16 + lr) * g.ldo + cur_n0 + wc * 64 + ch * 8;
                if (EPI == GE_BIAS_RELU) *reinterpret_cast<uint4 *>(dsth) = v;   // conv activations: read again at once
                else store_nt(dsth, v);
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else if constexpr (EPI == GE_S_BIAS_GELU || EPI == GE_S_BIAS_RELU_PAIR) {
        // fp16 PAIR out [M][2N] (hi | lo of the fp32 value; RELU_PAIR: [M][2 pair_c]): two [16][72] patches per pass -- hi in the wave's 4 KB
        // above the ring, lo in the idle k-loop ring -- so that both leave with one LDS round trip, each as whole
        // 128-byte row pieces
        _Float16 *whi = reinterpret_cast<_Float16 *>(patch + wave * 4096);
        _Float16 *wlo = reinterpret_cast<_Float16 *>(smem + wave * 4096);
        _Float16 *out = reinterpret_cast<_Float16 *>(g.out);
        float bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = g.bias[cur_n0 + wc * 64 + j * 16 + frow];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // (the epilogue is vector-ALU bound -- 128 elements per lane and tile, two transcendentals each; written on
            // element PAIRS so that the fused multiply-add, the products, the sum, the difference and both roundings to
            // fp16 are packed instructions (v_pk_fma_f32 ... v_cvt_pk_f16_f32): 11 instead of 20 per pair, same bits)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                    const f32x2 a = {acc[i][j][rp], acc[i][j][rp + 1]};
                    const f32x2 h = __builtin_elementwise_fma(a, f32x2{g.oscale, g.oscale}, f32x2{bias[j], bias[j]});
                    f32x2 v;
                    if constexpr (EPI == GE_S_BIAS_GELU) {
                        const f32x2 t = h * (-1.702f * 1.44269504088896340736f);
                        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
                        v = h * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};   // quick_gelu
                    } else {
                        v = f32x2{h[0] < 0.f ? 0.f : h[0], h[1] < 0.f ? 0.f : h[1]};               // relu
                    }
                    f16x2 hi, lo;
                    split_pair2(v[0], v[1], hi, lo);   // (common.h: v_cvt_pk_f16_f32 + two v_fma_mix*_f16, same bits)
                    whi[(fq * 4 + rp) * 72 + j * 16 + frow] = hi[0];
                    whi[(fq * 4 + rp + 1) * 72 + j * 16 + frow] = hi[1];
                    wlo[(fq * 4 + rp) * 72 + j * 16 + frow] = lo[0];
                    wlo[(fq * 4 + rp + 1) * 72 + j * 16 + frow] = lo[1];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int lr = it * 8 + (lane >> 3), ch = lane & 7;
                const uint4 vh = *reinterpret_cast<const uint4 *>(whi + lr * 72 + ch * 8);
                const uint4 vl = *reinterpret_cast<const uint4 *>(wlo + lr * 72 + ch * 8);
                _Float16 *dsth = out + (int64_t)(cur_m0 + wr * 128 + i * 16 + lr) * g.ldo + cur_n0 + wc * 64 + ch * 8;
                if constexpr (EPI == GE_S_BIAS_RELU_PAIR) {   // (read again at once by the next convolution: plain stores)
                    if (cur_n0 + wc * 64 + ch * 8 < g.pair_c) {
                        *reinterpret_cast<uint4 *>(dsth) = vh;
                        *reinterpret_cast<uint4 *>(dsth + g.pair_c) = vl;
                    }
                } else {
                    store_nt(dsth, vh);
                    store_nt(dsth + g.N, vl);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the lo patches alias stage 0 of the ring, which the next tile's prologue fills: all waves done first
        asm volatile("s_barrier" ::: "memory");
    } else if constexpr (EPI == GE_BIAS_ADD_RELU) {
        // out fp16 = relu(acc + bias + identity) with ONE rounding: the identity must meet the fp32 accumulators in
        // THEIR layout (one column, four rows per lane), but it should be read from memory as whole rows.  So each
        // pass's 16 x 64 identity block is loaded coalesced (16 B per lane, one pass ahead), laid into a second
        // wave-private LDS patch (in the idle k-loop ring) and picked up per element from there; the result leaves
        // through the usual fp16 patch as whole 128-byte rows.
        _Float16 *wreg = reinterpret_cast<_Float16 *>(patch + wave * 4096);          // result patch [16][72]
        _Float16 *pid = reinterpret_cast<_Float16 *>(smem + wave * 4096);            // identity patch [16][72]
        _Float16 *outh = reinterpret_cast<_Float16 *>(g.out);
        const int nb0 = cur_n0 + wc * 64;
        float bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = g.bias[nb0 + j * 16 + frow];
        const int lr0 = lane >> 3, ch = lane & 7;
        auto id_load = [&](int i, int it) -> uint4 {
            return *reinterpret_cast<const uint4 *>(g.identity + (int64_t)(cur_m0 + wr * 128 + i * 16 + it * 8 + lr0) * g.ldo +
                                                    nb0 + ch * 8);
        };
        uint4 idn0 = id_load(0, 0), idn1 = id_load(0, 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<uint4 *>(pid + lr0 * 72 + ch * 8) = idn0;
            *reinterpret_cast<uint4 *>(pid + (8 + lr0) * 72 + ch * 8) = idn1;
            if (i + 1 < 8) {
                idn0 = id_load(i + 1, 0);
                idn1 = id_load(i + 1, 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = (fq * 4 + r) * 72 + j * 16 + frow;
                    float v = acc[i][j][r] + bias[j];
                    v = v + (float)pid[e];
                    wreg[e] = (_Float16)(v < 0.f ? 0.f : v);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int lr = it * 8 + lr0;
                const uint4 v = *reinterpret_cast<const uint4 *>(wreg + lr * 72 + ch * 8);
                *reinterpret_cast<uint4 *>(outh + (int64_t)(cur_m0 + wr * 128 + i * 16 + lr) * g.ldo + nb0 + ch * 8) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the identity patches alias stage 0 of the ring, which the next tile's prologue fills: all waves done first
        asm volatile("s_barrier" ::: "memory");
    } else if ((EPI == GE_BIAS_RES || EPI == GE_S_BIAS_RES || EPI == GE_S_BIAS_RES_PAIR) && (g.ldo % 4 == 0) &&
               ((reinterpret_cast<uintptr_t>(g.out) & 15) == 0)) {
        // x[m][n] += acc + bias, x fp32 (residual stream).  The x values are the expensive part: fetched pass by pass
        // into VGPRs every pass pays a full HBM round trip (8 per tile, matrix pipe idle), and there are no registers
        // to fetch them ahead.  So they are fetched ahead into LDS instead: the k-loop's ring is idle during the
        // epilogue, each wave owns 16 KB of it as a 4-slot queue of x blocks ([16 rows][64 floats] = one pass),
        // filled by LDS-DMA (no VGPRs) four passes ahead and read back lane-linearly.  vmcnt retires in order, so
        // the wait for pass i's block counts the younger DMAs and stores (constants below).  The patch and queue
        // reads are inline asm: hipcc guards LDS loads it knows about with vmcnt(0) while an LDS-DMA is in flight.
        // (per-lane constants are derived from an opaque copy of the lane id HERE: hoisted to kernel start they
        // would be spilled across the k-loop and reloaded -- scratch loads count on vmcnt -- in every pass)
        int el = lane;
        asm volatile("" : "+v"(el));
        const int efrow = el & 15, efq = el >> 4;
        float *outp = reinterpret_cast<float *>(g.out);
        const int nbase = cur_n0 + wc * 64;
        const int c4 = efrow * 4;
        float4 bias4 = *reinterpret_cast<const float4 *>(g.bias + nbase + c4);
        // "use" the bias now: hipcc then waits for it HERE, before any DMA goes out, not with a vmcnt(0) in pass 0
        asm volatile("" : "+v"(bias4.x), "+v"(bias4.y), "+v"(bias4.z), "+v"(bias4.w)::"memory");
        float *wreg = reinterpret_cast<float *>(patch + wave * 4096);
        const unsigned x_loff = ((unsigned)efq * (unsigned)g.ldo + (unsigned)c4) * 4u;
        [[maybe_unused]] const unsigned p_loff = ((unsigned)efq * 2u * (unsigned)g.pair_c + (unsigned)c4) * 2u;   // RES_PAIR: lane offset in the pair tensor
        auto x_base = [&](int row16) -> uint64_t {   // row16: first row of a 4-row piece inside the wave's 128 rows
            return reinterpret_cast<uint64_t>(outp) + (uint64_t)(((int64_t)(cur_m0 + wr * 128 + row16) * g.ldo + nbase) * 4);
        };
        unsigned char *xq = smem + wave * 16384;
        const unsigned xq_lds = __builtin_amdgcn_readfirstlane(lds_addr(xq));
        const unsigned p_addr = lds_addr(wreg) + el * 16, q_addr = lds_addr(xq) + el * 16;
        auto x_dma = [&](int pass) {   // 16 rows x 256 B: four 1 KB pieces, lane l -> row l >> 4, 16 B at column c4
            const int slot = pass & 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) dma16_sv(x_base(pass * 16 + q * 4), x_loff, xq_lds + slot * 4096 + q * 1024);
        };
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) x_dma(pq);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) wreg[(efq * 4 + r) * 64 + j * 16 + efrow] = acc[i][j][r];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // younger than pass i's block: 4 DMAs per queued pass behind it + 4 stores per finished pass since:
            // i <= 3: 12 + 4 i, i >= 4: 12 + 4 (7 - i)
            // (RES_PAIR: 12 stores per pass -- 4 fp32 + 4 hi + 4 lo: i <= 3: 12 + 12 i; i >= 4: 36 + 4 (7 - i))
            if constexpr (EPI == GE_S_BIAS_RES_PAIR) wait_vmcnt(i < 4 ? 12 + 12 * i : 36 + 4 * (7 - i));   // 12 24 36 48 | 48 44 40 36
            else wait_vmcnt(i < 4 ? 12 + 4 * i : 24 - 4 * (i - 4));                                        // 12 16 20 24 | 24 20 16 12
            f32x4_t a[4], x[4];
            asm volatile("ds_read_b128 %0, %8\n\t"
                         "ds_read_b128 %1, %8 offset:1024\n\t"
                         "ds_read_b128 %2, %8 offset:2048\n\t"
                         "ds_read_b128 %3, %8 offset:3072\n\t"
                         "ds_read_b128 %4, %9\n\t"
                         "ds_read_b128 %5, %9 offset:1024\n\t"
                         "ds_read_b128 %6, %9 offset:2048\n\t"
                         "ds_read_b128 %7, %9 offset:3072\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]),
                           "=&v"(x[3])
                         : "v"(p_addr), "v"(q_addr + (i & 3) * 4096)
                         : "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                f32x4_t o;
                if constexpr (EPI == GE_S_BIAS_RES || EPI == GE_S_BIAS_RES_PAIR) {
                    if (g.relu_x) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[it][e] = x[it][e] < 0.f ? 0.f : x[it][e];
                    }
                    o[0] = x[it][0] + fmaf(a[it][0], g.oscale, bias4.x);
                    o[1] = x[it][1] + fmaf(a[it][1], g.oscale, bias4.y);
                    o[2] = x[it][2] + fmaf(a[it][2], g.oscale, bias4.z);
                    o[3] = x[it][3] + fmaf(a[it][3], g.oscale, bias4.w);
                } else {
                    o[0] = x[it][0] + (a[it][0] + bias4.x);
                    o[1] = x[it][1] + (a[it][1] + bias4.y);
                    o[2] = x[it][2] + (a[it][2] + bias4.z);
                    o[3] = x[it][3] + (a[it][3] + bias4.w);
                }
                store16_sv(x_base(i * 16 + it * 4), x_loff, o);
                if constexpr (EPI == GE_S_BIAS_RES_PAIR) {   // relu(x) as fp16 pairs: 8 bytes of hi and of lo per lane
                    typedef _Float16 h4p __attribute__((ext_vector_type(4)));
                    h4p hi, lo;
                    {
                        const float y[4] = {o[0] < 0.f ? 0.f : o[0], o[1] < 0.f ? 0.f : o[1], o[2] < 0.f ? 0.f : o[2],
                                            o[3] < 0.f ? 0.f : o[3]};
                        split_pairs<4>(y, hi, lo);
                    }
                    const uint64_t pb = reinterpret_cast<uint64_t>(g.pair_out) +
                                        (uint64_t)(((int64_t)(cur_m0 + wr * 128 + i * 16 + it * 4) * 2 * g.pair_c + nbase) * 2);
                    store8_sv(pb, p_loff, __builtin_bit_cast(u32x2, hi));
                    store8_sv(pb + (uint64_t)g.pair_c * 2u, p_loff, __builtin_bit_cast(u32x2, lo));
                }
            }
            if (i + 4 < 8) x_dma(i + 4);
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        }
        // the queues alias the stages the next tile's prologue is about to fill: every wave must be done reading
        asm volatile("s_barrier" ::: "memory");
    } else {
        // fp32 outputs: patch [16][64] floats (4 KB) per pass
        float *wreg = reinterpret_cast<float *>(patch + wave * 4096);
        float *outp = reinterpret_cast<float *>(g.out);
        const bool vec_ok = (g.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(outp) & 15) == 0);
        const int nbase = cur_n0 + wc * 64;
        const int c4 = (lane & 15) * 4;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), bn4 = bias4, cs4 = make_float4(1.f, 1.f, 1.f, 1.f);
        float bias1 = 0.f, bnv1 = 0.f, cs1 = 1.f;
        if (EPI == GE_EUCLID && g.rscale) {   // buffers are padded to the tile size: no bounds needed
            cs4 = *reinterpret_cast<const float4 *>(g.cscale + nbase + c4);
            cs1 = g.cscale[nbase + lane];
        }
        if (vec_ok) {
            if (EPI == GE_BIAS_RES || EPI == GE_S_BIAS_RES || EPI == GE_S_BIAS_RES_PAIR || EPI == GE_S_BIAS_F32)
                bias4 = *reinterpret_cast<const float4 *>(g.bias + nbase + c4);
            if (EPI == GE_EUCLID) {
                bn4.x = (nbase + c4 + 0 < g.n_valid) ? g.aux2[nbase + c4 + 0] : 0.f;
                bn4.y = (nbase + c4 + 1 < g.n_valid) ? g.aux2[nbase + c4 + 1] : 0.f;
                bn4.z = (nbase + c4 + 2 < g.n_valid) ? g.aux2[nbase + c4 + 2] : 0.f;
                bn4.w = (nbase + c4 + 3 < g.n_valid) ? g.aux2[nbase + c4 + 3] : 0.f;
            }
        } else {
            if (EPI == GE_BIAS_RES || EPI == GE_S_BIAS_RES || EPI == GE_S_BIAS_F32) bias1 = g.bias[nbase + lane];
            if (EPI == GE_EUCLID) bnv1 = (nbase + lane < g.n_valid) ? g.aux2[nbase + lane] : 0.f;
        }
        // (round-3 ablation of the stored 20k x 20k x 768 distance GEMM, one device: 0.848 ms = k-loop + operand casts 0.618,
        // epilogue instructions 0.138, stores 0.092.  Tried and NOT kept: the matrix instruction with swapped operands, so
        // that a lane owns four consecutive output columns and stores 16 bytes straight from the accumulators with no LDS
        // pass -- bit-identical, but 0.909 ms: a store instruction then covers 16 rows x 64 bytes, i.e. half lines, and the
        // memory system takes those slower than whole 256-byte row pieces.  A start stagger of 3-30 us: no change.)
        // GE_EUCLID: the wave's 128 row norms (and row scales) are fetched ONCE per tile into a wave-private table in the
        // idle k-loop ring.  Fetched per pass -- as this epilogue did until round 3 -- every load sat behind the stores of
        // the pass before it (vmcnt retires in order and a load is the youngest operation, so its wait is a vmcnt(0)):
        // 32 dependent store round trips per tile, ~10 us of an ~31 us tile with the matrix pipe idle.
        // fast path (below): the ring -- idle until the next prologue -- holds four patches per wave and the tables sit in
        // the wave's 4 KB above the ring; general path: one patch above the ring, the tables in the ring
        bool fast_path = false;
        if constexpr (EPI == GE_EUCLID || EPI == GE_S_BIAS_F32)
            fast_path = vec_ok && g.ldo < (1 << 24) &&
                        (EPI == GE_S_BIAS_F32 || (cur_m0 + BBM <= g.m_valid && cur_n0 + BBN <= g.n_valid));
        float *rt = fast_path ? reinterpret_cast<float *>(patch + wave * 4096)
                              : reinterpret_cast<float *>(smem + wave * 16384);   // [128] |q|^2, then [128] row scales
        if constexpr (EPI == GE_EUCLID) {
            const int r0 = cur_m0 + wr * 128;
            const float t0 = (r0 + lane < g.m_valid) ? g.aux[r0 + lane] : 0.f;
            const float t1 = (r0 + 64 + lane < g.m_valid) ? g.aux[r0 + 64 + lane] : 0.f;
            float t2 = 1.f, t3 = 1.f;
            if (g.rscale) {
                t2 = g.rscale[r0 + lane];           // (padded to the tile size)
                t3 = g.rscale[r0 + 64 + lane];
            }
            rt[lane] = t0;
            rt[64 + lane] = t1;
            rt[128 + lane] = t2;
            rt[192 + lane] = t3;
        }
        // Fast path: GE_EUCLID on an interior tile (every tile but the last tile row / column) and GE_S_BIAS_F32 (the
        // split mode's q | k | v output, never ragged).  The general loop below spends ~45 instructions per store on 64-bit
        // address arithmetic, row / column bounds and exec masking; here: wave-uniform 64-bit base (SGPR pair) + four
        // per-lane 32-bit offsets computed once per tile, no bounds, the scale test hoisted.  Same arithmetic in the same
        // order: same bits.  MPREID_FAST_NBLK 16-row blocks are laid into the idle ring per LDS round trip; measured on
        // one device in one run (20k x 20k x 768 stored distances): 1 block 0.803 ms, 2 blocks 0.813, 4 blocks 0.844 --
        // batching the round trips only adds live registers (per-tile stamps: epilogue 8.5 -> 10.2 us), the time goes to
        // the stores themselves (docs/HISTORY_r01-r03.md).
        bool fast_done = false;
        if constexpr (EPI == GE_EUCLID || EPI == GE_S_BIAS_F32) {
            if (fast_path) {
                fast_done = true;
                float *tp4 = reinterpret_cast<float *>(smem + wave * 16384);   // four [16][64] patches
                const uint64_t obase = reinterpret_cast<uint64_t>(outp + (int64_t)(cur_m0 + wr * 128) * g.ldo + nbase);
                const unsigned ldo_b = (unsigned)g.ldo * 4u;
                unsigned voff[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) voff[it] = (unsigned)(it * 4 + (lane >> 4)) * ldo_b + (unsigned)c4 * 4u;
                const bool scaled = g.rscale != nullptr;
                #ifndef MPREID_FAST_NBLK
#define MPREID_FAST_NBLK 1
#endif
                constexpr int NBLK = MPREID_FAST_NBLK;
                if constexpr (((DBG >> 7) & 7) == 6) {   // (timing experiment, wrong data) the 32 stores straight from the accumulators
#pragma unroll
                    for (int e = 0; e < 32; ++e)
                        store16_nt_sv(obase + (uint64_t)(e >> 2) * 16u * ldo_b, voff[e & 3], acc[e >> 2][e & 3]);
                } else
#pragma unroll
                for (int h = 0; h < 8 / NBLK; ++h) {   // NBLK 16-row blocks per round trip
#pragma unroll
                    for (int ii = 0; ii < NBLK; ++ii)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                tp4[ii * 1024 + (fq * 4 + r) * 64 + j * 16 + frow] = acc[h * NBLK + ii][j][r];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (h == 0) stamp(3);
                    float4 av[4 * NBLK];
                    float amv[4 * NBLK], rsv[4 * NBLK];
#pragma unroll
                    for (int e = 0; e < 4 * NBLK; ++e) {
                        const int ii = e >> 2, lr = (e & 3) * 4 + (lane >> 4);
                        av[e] = *reinterpret_cast<const float4 *>(tp4 + ii * 1024 + lr * 64 + c4);
                        if constexpr (EPI == GE_EUCLID) {
                            amv[e] = rt[(h * NBLK + ii) * 16 + lr];
                            rsv[e] = scaled ? rt[128 + (h * NBLK + ii) * 16 + lr] : 1.0f;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4 * NBLK; ++e) {
                        const int ii = e >> 2, it = e & 3;
                        const uint64_t pbase = obase + (uint64_t)(h * NBLK + ii) * 16u * ldo_b;
                        const float4 a = av[e];
                        f32x4 o;
                        if constexpr (EPI == GE_S_BIAS_F32) {
                            o[0] = fmaf(a.x, g.oscale, bias4.x);
                            o[1] = fmaf(a.y, g.oscale, bias4.y);
                            o[2] = fmaf(a.z, g.oscale, bias4.z);
                            o[3] = fmaf(a.w, g.oscale, bias4.w);
                        } else if (scaled) {
                            const float rs = rsv[e], am = amv[e];
                            o[0] = fmaf(-2.0f, a.x * (rs * cs4.x), am + bn4.x);
                            o[1] = fmaf(-2.0f, a.y * (rs * cs4.y), am + bn4.y);
                            o[2] = fmaf(-2.0f, a.z * (rs * cs4.z), am + bn4.z);
                            o[3] = fmaf(-2.0f, a.w * (rs * cs4.w), am + bn4.w);
                        } else {
                            const float am = amv[e];
                            o[0] = fmaf(-2.0f, a.x, am + bn4.x);
                            o[1] = fmaf(-2.0f, a.y, am + bn4.y);
                            o[2] = fmaf(-2.0f, a.z, am + bn4.z);
                            o[3] = fmaf(-2.0f, a.w, am + bn4.w);
                        }
                        store16_flav_sv<(DBG >> 7) & 7>(pbase, voff[it], o);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (h == 0) stamp(6);
                    if (h == 8 / NBLK - 1) stamp(5);
                }
            }
        }
        if (!fast_done) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) wreg[(fq * 4 + r) * 64 + j * 16 + frow] = acc[i][j][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int mbase = cur_m0 + wr * 128 + i * 16;
            if (vec_ok) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int lr = it * 4 + (lane >> 4);
                    const int m = mbase + lr;
                    const float4 a = *reinterpret_cast<const float4 *>(wreg + lr * 64 + c4);
                    float *dst = outp + (int64_t)m * g.ldo + nbase + c4;
                    if (EPI == GE_F32) {
                        *reinterpret_cast<float4 *>(dst) = a;
                    } else if (EPI == GE_S_BIAS_F32) {
                        // q | k | v for the split attention kernel: read once, by a later kernel
                        store_nt(dst, make_float4(fmaf(a.x, g.oscale, bias4.x), fmaf(a.y, g.oscale, bias4.y),
                                                  fmaf(a.z, g.oscale, bias4.z), fmaf(a.w, g.oscale, bias4.w)));
                    } else if (EPI == GE_S_BIAS_RES) {
                        float4 x = *reinterpret_cast<const float4 *>(dst);
                        if (g.relu_x) {
                            x.x = x.x < 0.f ? 0.f : x.x;
                            x.y = x.y < 0.f ? 0.f : x.y;
                            x.z = x.z < 0.f ? 0.f : x.z;
                            x.w = x.w < 0.f ? 0.f : x.w;
                        }
                        x.x = x.x + fmaf(a.x, g.oscale, bias4.x);
                        x.y = x.y + fmaf(a.y, g.oscale, bias4.y);
                        x.z = x.z + fmaf(a.z, g.oscale, bias4.z);
                        x.w = x.w + fmaf(a.w, g.oscale, bias4.w);
                        *reinterpret_cast<float4 *>(dst) = x;
                    } else if (EPI == GE_PATCH || EPI == GE_S_PATCH) {
                        // patch embedding: row m = b*P + p goes to token 1 + p of image b, + positional embedding
                        if (m < g.m_valid) {
                            const int b = m / g.P, pp = m - b * g.P;
                            const float os = (EPI == GE_S_PATCH) ? g.oscale : 1.0f;   // (a * 1 is exact: same bits as before)
                            const float4 pe = *reinterpret_cast<const float4 *>(g.aux + (int64_t)(1 + pp) * g.N + nbase + c4);
                            *reinterpret_cast<float4 *>(outp + ((int64_t)b * g.L + 1 + pp) * g.ldo + nbase + c4) =
                                (EPI == GE_S_PATCH) ? make_float4(a.x * os + pe.x, a.y * os + pe.y, a.z * os + pe.z, a.w * os + pe.w)
                                                    : make_float4(a.x + pe.x, a.y + pe.y, a.z + pe.z, a.w + pe.w);
                        }
                    } else if (EPI == GE_BIAS_RES) {
                        const f32x4_t xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(dst));
                        float4 x = make_float4(xv[0], xv[1], xv[2], xv[3]);
                        x.x = x.x + (a.x + bias4.x);
                        x.y = x.y + (a.y + bias4.y);
                        x.z = x.z + (a.z + bias4.z);
                        x.w = x.w + (a.w + bias4.w);
                        *reinterpret_cast<float4 *>(dst) = x;   // (streaming stores measured neutral here)
                    } else if (m < g.m_valid) { // GE_EUCLID
                        const float am = rt[i * 16 + lr];
                        float4 o, s4 = make_float4(1.f, 1.f, 1.f, 1.f);
                        if (g.rscale) {
                            const float rs = rt[128 + i * 16 + lr];
                            s4 = make_float4(rs * cs4.x, rs * cs4.y, rs * cs4.z, rs * cs4.w);
                        }
                        o.x = fmaf(-2.0f, a.x * s4.x, am + bn4.x);
                        o.y = fmaf(-2.0f, a.y * s4.y, am + bn4.y);
                        o.z = fmaf(-2.0f, a.z * s4.z, am + bn4.z);
                        o.w = fmaf(-2.0f, a.w * s4.w, am + bn4.w);
                        if (nbase + c4 + 3 < g.n_valid) {
                            store_nt(dst, o);   // N x N distances: written once, far larger than any cache
                        } else {
                            if (nbase + c4 + 0 < g.n_valid) dst[0] = o.x;
                            if (nbase + c4 + 1 < g.n_valid) dst[1] = o.y;
                            if (nbase + c4 + 2 < g.n_valid) dst[2] = o.z;
                        }
                    }
                }
            } else {
                const int n = nbase + lane;
#pragma unroll 4
                for (int lr = 0; lr < 16; ++lr) {
                    const int m = mbase + lr;
                    const float a = wreg[lr * 64 + lane];
                    float *dst = outp + (int64_t)m * g.ldo + n;
                    if (EPI == GE_F32) {
                        *dst = a;
                    } else if (EPI == GE_S_BIAS_F32) {
                        *dst = fmaf(a, g.oscale, bias1);
                    } else if (EPI == GE_S_BIAS_RES) {
                        const float xv = *dst;
                        *dst = ((g.relu_x && xv < 0.f) ? 0.f : xv) + fmaf(a, g.oscale, bias1);
                    } else if (EPI == GE_PATCH || EPI == GE_S_PATCH) {
                        if (m < g.m_valid) {
                            const int b = m / g.P, pp = m - b * g.P;
                            outp[((int64_t)b * g.L + 1 + pp) * g.ldo + n] =
                                ((EPI == GE_S_PATCH) ? a * g.oscale : a) + g.aux[(int64_t)(1 + pp) * g.N + n];
                        }
                    } else if (EPI == GE_BIAS_RES) {
                        *dst = *dst + (a + bias1);
                    } else if (m < g.m_valid && n < g.n_valid) {
                        const float sc = g.rscale ? rt[128 + i * 16 + lr] * cs1 : 1.0f;
                        *dst = fmaf(-2.0f, a * sc, rt[i * 16 + lr] + bnv1);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        }
        if constexpr (SYM_STORE) {
            if (g.sym && cur_m0 != cur_n0) {
                // SYMMETRIC problem (all-pairs distances of one set), off-diagonal tile: the tile (tn, tm) is never computed --
                // its values are this tile's, transposed (same fp16 products in the same k order, |q_m|^2 + |q_n|^2 commutes:
                // the bits the other tile would have produced, tests/test_gpu_distance.py).  The wave's 128 x 64 block leaves
                // as 64 rows x 128 columns in two halves of 64 x 64: the finished values go into the wave's 16 KB of the idle
                // ring TRANSPOSED ([n][m], 16-byte chunks XOR-swizzled by n & 15: the b128 writes of a 16-lane group and the
                // b128 reads of a row both touch 16 different chunk banks) and leave as whole 256-byte row pieces, like the
                // direct copy.  (Straight from the accumulators a store instruction would cover 16 rows x 64 bytes: half
                // lines, measured slower in round 3.)
                float *T = reinterpret_cast<float *>(smem + wave * 16384);
                float *rtm = reinterpret_cast<float *>(patch + wave * 4096);
                if (!fast_path) {   // the general path kept its row table in the ring slice: move it above the ring
                    const float t0 = rt[lane], t1 = rt[64 + lane], t2 = rt[128 + lane], t3 = rt[192 + lane];
                    __builtin_amdgcn_wave_barrier();
                    rtm[lane] = t0;
                    rtm[64 + lane] = t1;
                    rtm[128 + lane] = t2;
                    rtm[192 + lane] = t3;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                const bool scaledT = g.rscale != nullptr;
                float bnT[4], csT[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = nbase + j * 16 + frow;
                    bnT[j] = (n < g.n_valid) ? g.aux2[n] : 0.f;
                    csT[j] = scaledT ? g.cscale[n] : 1.0f;
                }
                const bool interior = vec_ok && cur_m0 + BBM <= g.m_valid && cur_n0 + BBN <= g.n_valid;
                // the accumulators become the finished values IN PLACE, one 16-row block at a time (short live ranges: the
                // direct copy above is done with the raw sums)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 am4 = *reinterpret_cast<const float4 *>(rtm + i * 16 + fq * 4);
                    const float amr[4] = {am4.x, am4.y, am4.z, am4.w};
                    if (scaledT) {
                        const float4 rs4 = *reinterpret_cast<const float4 *>(rtm + 128 + i * 16 + fq * 4);
                        const float rsr[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[i][j][r] = fmaf(-2.0f, acc[i][j][r] * (rsr[r] * csT[j]), amr[r] + bnT[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(-2.0f, acc[i][j][r], amr[r] + bnT[j]);
                    }
                    asm volatile("" ::: "memory");
                }
                float *Trow = T + frow * 64;                      // + j * 1024: row n = j * 16 + frow of the transposed patch
                const int cx = fq ^ frow;                         // chunk (ii * 4 + fq) ^ frow = (ii * 4) ^ cx  (ii * 4 has no low bits)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<f32x4 *>(Trow + j * 1024 + ((((ii * 4) ^ cx) & 15) << 2)) = acc[hf * 4 + ii][j];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int l16 = lane & 15;
                    const int mcol = cur_m0 + wr * 128 + hf * 64 + l16 * 4;
                    // wave-uniform 64-bit base + one 32-bit lane offset (as the direct fast path): row nl = e * 4 + (lane >> 4)
                    const uint64_t tbase = reinterpret_cast<uint64_t>(outp + (int64_t)nbase * g.ldo + cur_m0 + wr * 128 + hf * 64);
                    const unsigned ldo_bT = (unsigned)g.ldo * 4u;
                    const unsigned voffT = (unsigned)(lane >> 4) * ldo_bT + (unsigned)l16 * 16u;
                    if (interior) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int nl = e * 4 + (lane >> 4);
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(T + nl * 64 + ((l16 ^ (nl & 15)) << 2));
                            store16_nt_sv(tbase + (uint64_t)(e * 4) * ldo_bT, voffT, v);
                        }
                    } else {   // last tile row / column: bounds per row and per 16-byte piece (rolled: 158 of 3160 tiles at N = 20 000)
#pragma unroll 1
                        for (int e = 0; e < 16; ++e) {
                            const int nl = e * 4 + (lane >> 4);
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(T + nl * 64 + ((l16 ^ (nl & 15)) << 2));
                            const int n = nbase + nl;
                            float *dst = outp + (int64_t)n * g.ldo + mcol;
                            if (n < g.n_valid) {
                                if (vec_ok && mcol + 3 < g.m_valid) {
                                    store_nt(dst, make_float4(v[0], v[1], v[2], v[3]));
                                } else {
                                    if (mcol + 0 < g.m_valid) dst[0] = v[0];
                                    if (mcol + 1 < g.m_valid) dst[1] = v[1];
                                    if (mcol + 2 < g.m_valid) dst[2] = v[2];
                                    if (mcol + 3 < g.m_valid) dst[3] = v[3];
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        // the fast path's patches alias the ring, which the next tile's prologue fills: all waves done first
        if constexpr (EPI == GE_EUCLID || EPI == GE_S_BIAS_F32) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // next tile: its first three DMA stages go out right behind this tile's stores (the ring was released by
    // the barrier above), so their cold latency overlaps the store drain
    stamp(2);
    if constexpr ((DBG & 32) != 0) ++stamp_tile;
    tile = next_tile;
    if (tile < 0) break;
    set_tile(tile);
    dma_prologue();
    } // tiles
}

// ---------------------------------------------------------------------------------------------
// Symmetric stored distance matrix (all-pairs distances of ONE set: A == W), "p2" schedule: PERSISTENT, TWO independent
// 4-wave workgroups per CU walking tiles out of phase.
//   Why: in the one-workgroup-per-CU kernel above ~45 % of a tile of this problem is its epilogue -- 2 x 256 KB of fp32
//   leaving through the CU's ~16 B/clk store path with the matrix pipe idle (docs/HISTORY_r04.md).  That kernel owns the
//   CU's whole LDS and half of its registers, so nothing else can run beside its stores.  Here a workgroup is half the
//   size (tile 256 x 128, waves 2 x 2 of 128 x 64 -- the same fragment code -- and a 3-slot ring of 24 KB stages = 72 KB),
//   two of them share a CU, and while one drains its stores the other runs its k-loop.
//   Tiles: only those that reach the upper triangle (tn >= 2 tm in units of the 256 x 128 tile) are computed; a tile
//   entirely above the diagonal (tn >= 2 tm + 2) also stores its transpose, through the same swizzled LDS transposition
//   as the kernel above.  Order: strips of 8 tile rows, column-major inside a strip, so that 64 consecutive tiles -- the
//   64 workgroups of an XCD at one time -- are an 8 x 8 block sharing 8 + 8 operand panels in that XCD's L2; the list is
//   cut into eight EQUAL contiguous ranges, one per XCD (closed-form position -> tile map, no list in memory).
//   Same products in the same k order as the kernels above: same bits (tests/test_gpu_distance.py).
// ---------------------------------------------------------------------------------------------
constexpr int PBM = 256, PBN = 128, PBK = 32, P_NSTAGE = 3;
constexpr int P_A_BYTES = PBM * PBK * 2;                 // 16 KB
constexpr int P_B_BYTES = PBN * PBK * 2;                 // 8 KB
constexpr int P_STAGE_BYTES = P_A_BYTES + P_B_BYTES;     // 24 KB
constexpr int P_RING_BYTES = P_NSTAGE * P_STAGE_BYTES;   // 72 KB
constexpr int P_LDS_BYTES = P_RING_BYTES + 4 * 1024;     // + the four waves' row tables: 76 KB, two workgroups per CU

// number of tiles of strip br (tile rows 8 br .. 8 br + R - 1, tile columns 16 br .. tiles_n - 1): column j of the strip
// holds min(R, j / 2 + 1) tiles
__device__ __forceinline__ int p2_strip_size(int R, int ncol, int &ramp) {
    ramp = min(ncol, 2 * (R - 1));
    const int p = ramp >> 1;
    return p * (p + 1) + ((ramp & 1) ? (p + 1) : 0) + (ncol - ramp) * R;
}
__device__ __forceinline__ int p2_total_tiles(int tiles_m, int tiles_n, int SH) {   // SH = tile rows per strip
    int T = 0, ramp;
    for (int br = 0; SH * br < tiles_m; ++br) T += p2_strip_size(min(SH, tiles_m - SH * br), tiles_n - 2 * SH * br, ramp);
    return T;
}
__device__ __forceinline__ void p2_tile_at(int k, int tiles_m, int tiles_n, int SH, int &tm, int &tn) {   // k < total
    int br = 0, R, ramp;
    for (;; ++br) {
        R = min(SH, tiles_m - SH * br);
        const int S = p2_strip_size(R, tiles_n - 2 * SH * br, ramp);
        if (k < S) break;
        k -= S;
    }
    int j = 0;
    for (; j < ramp; ++j) {
        const int c = (j >> 1) + 1;
        if (k < c) break;
        k -= c;
    }
    if (j == ramp) {
        j += k / R;
        k = k % R;
    }
    tm = SH * br + k;
    tn = 2 * SH * br + j;
}

// FULL (round 5): the same two-workgroups-per-CU schedule for TWO tensors (q != g, the evaluator's real shape,
// utils/metrics.py:7-13): every tile of the tiles_m x tiles_n grid is computed, nothing is mirrored.  Walk: strips of SH tile
// rows, column-major inside a strip, cut into eight equal contiguous ranges (one per XCD) like the symmetric walk.
template <int abl, bool FULL = false>   // ablation variants (MPREID_ABLATION builds): bit 0 no stores, 1 no k-loop, 2 no mirrored stores, 3 operands aliased
                     // onto two L2-resident panels, 4 stores straight from the accumulators (wrong data)
__global__ __launch_bounds__(256, 2) void dist_sym_p2_kernel(GemmArgs g, int tiles_m, int tiles_n, int naps, int SH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int K = g.K;
    const int nst = (abl & 2) ? 1 : K / PBK;
    const int per_xcd = (int)gridDim.x >> 3;                      // the grid is a multiple of 8
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    const int total = FULL ? tiles_m * tiles_n : p2_total_tiles(tiles_m, tiles_n, SH);
    const int lo = (int)((int64_t)xcd * total / 8), hi = (int)((int64_t)(xcd + 1) * total / 8);

    // (experiments) the second workgroup of every CU -- blocks b and b + gridDim/2 share a CU, tools/probes/hwid_probe.hip --
    // can be started late.  Neither that nor a per-CU token that lets only one of the two store at a time (perfect
    // alternation, checked with per-tile stamps) helps: a tile's 256 KB take ~20 us to leave while the neighbour runs its
    // k-loop, 12 us when nothing else uses the CU's vector-memory path.
    if (slot >= (per_xcd >> 1))
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(8);

    // DMA geometry: one piece = 16 rows x 64 B.  A part: 16 pieces, wave w takes 4w..4w+3; B part: 8 pieces, wave w takes
    // 2w, 2w+1.  Rows 8..15 of a piece hold their four 16-byte chunks reversed (the read side undoes it): the b128
    // fragment reads of a 16-lane group then touch all banks
    const int drow = lane >> 2;
    const int dchunk = (lane & 3) ^ (((drow >> 3) & 1) * 3);
    const int frow = lane & 15, fq = lane >> 4;
    const int fsw = (fq ^ (((lane >> 3) & 1) * 3)) << 4;
    const int a_off = (wr * 128 + frow) * 64 + fsw;                // + i * 1024
    const int b_off = P_A_BYTES + (wc * 64 + frow) * 64 + fsw;     // + j * 1024
    float *outp = reinterpret_cast<float *>(g.out);
    const bool vec_ok = (g.ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(outp) & 15) == 0) && g.ldo < (1 << 24);
    const bool scaled = g.rscale != nullptr;
    float *rt = reinterpret_cast<float *>(smem + P_RING_BYTES + wave * 1024);   // [128] |x_m|^2, [128] row scales
    float *T = reinterpret_cast<float *>(smem + wave * (P_RING_BYTES / 4));     // 18 KB of the idle ring per wave

    [[maybe_unused]] int stamp_tile = 0;
    auto stamp = [&](int which) {   // (ablation builds) per-tile phase stamps of the 100 MHz counter, tools/p2_stamps.py
#ifdef MPREID_ABLATION
        if (g.stamps && tid == 0 && stamp_tile < 32)
            g.stamps[((size_t)blockIdx.x * 32 + stamp_tile) * 8 + which] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    for (int it = lo + slot; it < hi; it += per_xcd) {
        int tm, tn;
        if constexpr (FULL) {   // strip br = SH tile rows x all tile columns, column-major inside the strip
            const int per_strip = SH * tiles_n;
            const int br = it / per_strip, kk = it - br * per_strip;
            const int R = min(SH, tiles_m - SH * br);
            tn = kk / R;
            tm = SH * br + (kk - tn * R);
        } else {
            p2_tile_at(it, tiles_m, tiles_n, SH, tm, tn);
        }
        const int m0 = tm * PBM, n0 = tn * PBN;
        const int ma = (abl & 8) ? (tm & 1) * PBM : m0, na = (abl & 8) ? (tn & 1) * PBN : n0;   // (experiment) L2-resident operands
        const _Float16 *a_src = g.A + (int64_t)(ma + wave * 64 + drow) * K + dchunk * 8;
        const _Float16 *b_src = g.W + (int64_t)(na + wave * 32 + drow) * K + dchunk * 8;
        auto dma_stage = [&](int st) {
            unsigned char *sb = smem + (st % P_NSTAGE) * P_STAGE_BYTES;
            const int koff = st * PBK;
            if constexpr ((abl & 32) != 0) {   // (timing experiment, wrong data) the same bytes requested as WHOLE 128-byte lines:
                // even stages the first half of the rows, odd stages the second half, both k slabs of a line at once
                const _Float16 *aw = g.A + (int64_t)(ma + (st & 1) * 128 + wave * 32 + (lane >> 3)) * K + (st >> 1) * 64 + (lane & 7) * 8;
                const _Float16 *bw = g.W + (int64_t)(na + (st & 1) * 64 + wave * 16 + (lane >> 3)) * K + (st >> 1) * 64 + (lane & 7) * 8;
#pragma unroll
                for (int t = 0; t < 4; ++t) dma16(aw + (int64_t)t * 8 * K, sb + wave * 4096 + t * 1024);
#pragma unroll
                for (int t = 0; t < 2; ++t) dma16(bw + (int64_t)t * 8 * K, sb + P_A_BYTES + wave * 2048 + t * 1024);
                return;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) dma16(a_src + (int64_t)t * 16 * K + koff, sb + wave * 4096 + t * 1024);
#pragma unroll
            for (int t = 0; t < 2; ++t) dma16(b_src + (int64_t)t * 16 * K + koff, sb + P_A_BYTES + wave * 2048 + t * 1024);
        };

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // the previous tile's epilogue is done with the ring (LDS reads only: its stores drain behind the first stages)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        dma_stage(0);
        if (nst > 1) dma_stage(1);
        // One barrier per stage: stage t has landed everywhere (stage t+1 may stay in flight), every wave is done with stage
        // t-1, whose slot takes the DMA of stage t+2; then 12 fragment reads and 32 matrix instructions.  The other
        // workgroup of the CU fills the gaps.  (A software-pipelined version -- fragments of stage t+1 and the DMA of stage
        // t+3 spread between the matrix instructions of stage t, as in the 256 x 256 kernel -- runs the k-loop ALONE 7 %
        // faster and the whole kernel slower, 0.62 against 0.54-0.59 ms at N = 20 000 on one device: this kernel's
        // k-loop is bound by the CU's vector-memory path, which its stores share; docs/HISTORY_r04.md.)
        for (int t = 0; t < nst; ++t) {
            if (t + 1 < nst) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (t == 0) stamp(0);
            if (t + 2 < nst) dma_stage(t + 2);
            const unsigned char *sb = smem + (t % P_NSTAGE) * P_STAGE_BYTES;
            f16x8 fa[8], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const f16x8 *>(sb + b_off + j * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const f16x8 *>(sb + a_off + i * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }

        // ---- epilogue ----
        stamp(1);
        const int r0 = m0 + wr * 128, nbase = n0 + wc * 64;
        {   // row tables of the wave (above the ring: no hazard with the other waves' last fragment reads)
            const float t0 = (r0 + lane < g.m_valid) ? g.aux[r0 + lane] : 0.f;
            const float t1 = (r0 + 64 + lane < g.m_valid) ? g.aux[r0 + 64 + lane] : 0.f;
            float t2 = 1.f, t3 = 1.f;
            if (scaled) {
                t2 = g.rscale[r0 + lane];           // (padded to the tile size)
                t3 = g.rscale[r0 + 64 + lane];
            }
            rt[lane] = t0;
            rt[64 + lane] = t1;
            rt[128 + lane] = t2;
            rt[192 + lane] = t3;
        }
        float bnT[4], csT[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nbase + j * 16 + frow;
            bnT[j] = (n < g.n_valid) ? g.aux2[n] : 0.f;
            csT[j] = scaled ? g.cscale[n] : 1.0f;
        }
        // every wave is past its last fragment reads: the ring becomes the waves' patches
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(2);
        // the accumulators become the finished values in place, one 16-row block at a time
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 am4 = *reinterpret_cast<const float4 *>(rt + i * 16 + fq * 4);
            const float amr[4] = {am4.x, am4.y, am4.z, am4.w};
            if (scaled) {
                const float4 rs4 = *reinterpret_cast<const float4 *>(rt + 128 + i * 16 + fq * 4);
                const float rsr[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[i][j][r] = fmaf(-2.0f, acc[i][j][r] * (rsr[r] * csT[j]), amr[r] + bnT[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(-2.0f, acc[i][j][r], amr[r] + bnT[j]);
            }
            asm volatile("" ::: "memory");
        }
        const bool interior = vec_ok && m0 + PBM <= g.m_valid && n0 + PBN <= g.n_valid;
        const int l16 = lane & 15;
        if constexpr (!(abl & 1)) {
        // (1) the block itself: 16 rows at a time through a [16][68] patch (row stride 68 floats: the four row groups of
        // a b32 write land in different banks), out as whole 256-byte row pieces
        {
            const uint64_t obase = reinterpret_cast<uint64_t>(outp + (int64_t)r0 * g.ldo + nbase);
            const unsigned ldo_b = (unsigned)g.ldo * 4u;
            const unsigned voff = (unsigned)(lane >> 4) * ldo_b + (unsigned)l16 * 16u;
            if constexpr ((abl & 16) != 0) {   // (timing experiment, wrong data) straight from the accumulators, no LDS round trips
                if (interior)
#pragma unroll
                for (int e = 0; e < 32; ++e) store16_nt_sv(obase + (uint64_t)(e * 4) * ldo_b, voff, acc[e >> 2][e & 3]);
                if (interior && tn >= 2 * tm + 2) {
                    const uint64_t tb = reinterpret_cast<uint64_t>(outp + (int64_t)nbase * g.ldo + r0);
#pragma unroll
                    for (int e = 0; e < 32; ++e)
                        store16_nt_sv(tb + (uint64_t)((e & 15) * 4) * ldo_b + (e >> 4) * 256, voff, acc[e >> 2][e & 3]);
                }
            } else
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[(fq * 4 + r) * 68 + j * 16 + frow] = acc[i][j][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (interior) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(T + (e * 4 + (lane >> 4)) * 68 + l16 * 4);
                        store16_nt_sv(obase + (uint64_t)(i * 16 + e * 4) * ldo_b, voff, v);
                    }
                } else {
#pragma unroll 1
                    for (int e = 0; e < 4; ++e) {
                        const int lr = e * 4 + (lane >> 4);
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(T + lr * 68 + l16 * 4);
                        const int m = r0 + i * 16 + lr, n = nbase + l16 * 4;
                        float *dst = outp + (int64_t)m * g.ldo + n;
                        if (m < g.m_valid) {
                            if (vec_ok && n + 3 < g.n_valid) {
                                store_nt(dst, make_float4(v[0], v[1], v[2], v[3]));
                            } else {
                                if (n + 0 < g.n_valid) dst[0] = v[0];
                                if (n + 1 < g.n_valid) dst[1] = v[1];
                                if (n + 2 < g.n_valid) dst[2] = v[2];
                                if (n + 3 < g.n_valid) dst[3] = v[3];
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // (2) its transpose, when the tile lies entirely above the diagonal: 64 rows x 128 columns in two halves, each
        // through a [64 n][64 m] patch whose 16-byte chunks are XOR-swizzled by n & 15 (b128 writes of a 16-lane group
        // and b128 reads of a row both touch 16 different chunk banks)
        if (!FULL && tn >= 2 * tm + 2 && !(abl & (4 | 16))) {
            float *Trow = T + frow * 64;                      // + j * 1024: row n = j * 16 + frow
            const int cx = fq ^ frow;                         // chunk (ii * 4 + fq) ^ frow = (ii * 4) ^ cx
            const unsigned ldo_bT = (unsigned)g.ldo * 4u;
            const unsigned voffT = (unsigned)(lane >> 4) * ldo_bT + (unsigned)l16 * 16u;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<f32x4 *>(Trow + j * 1024 + ((((ii * 4) ^ cx) & 15) << 2)) = acc[hf * 4 + ii][j];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int mcol = r0 + hf * 64 + l16 * 4;
                const uint64_t tbase = reinterpret_cast<uint64_t>(outp + (int64_t)nbase * g.ldo + r0 + hf * 64);
                if (interior) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int nl = e * 4 + (lane >> 4);
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(T + nl * 64 + ((l16 ^ (nl & 15)) << 2));
                        store16_nt_sv(tbase + (uint64_t)(e * 4) * ldo_bT, voffT, v);
                    }
                } else {
#pragma unroll 1
                    for (int e = 0; e < 16; ++e) {
                        const int nl = e * 4 + (lane >> 4);
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(T + nl * 64 + ((l16 ^ (nl & 15)) << 2));
                        const int n = nbase + nl;
                        float *dst = outp + (int64_t)n * g.ldo + mcol;
                        if (n < g.n_valid) {
                            if (vec_ok && mcol + 3 < g.m_valid) {
                                store_nt(dst, make_float4(v[0], v[1], v[2], v[3]));
                            } else {
                                if (mcol + 0 < g.m_valid) dst[0] = v[0];
                                if (mcol + 1 < g.m_valid) dst[1] = v[1];
                                if (mcol + 2 < g.m_valid) dst[2] = v[2];
                                if (mcol + 3 < g.m_valid) dst[3] = v[3];
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        }
        stamp(3);
#ifdef MPREID_ABLATION
        if (g.stamps) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        stamp(4);
        ++stamp_tile;
    }
}

// ---------------------------------------------------------------------------------------------
// optional per-launch event timing (bench.py's roofline leg): hipEvents recorded on the launch
// stream around every GEMM launch while enabled, aggregated per (epilogue, M, N, K) class.
// ---------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
namespace {
struct ProfClass {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double flops_total = 0.0;
    int64_t m = 0; // largest M seen in the class
};
bool g_prof_on = false;
// MPREID_TUNE gemm_big: 0 = never use the 256x256 kernel, 1 = when the grid fills the chip (default),
// 2 = whenever the shape is divisible (tests)
int big_mode() {
    static const int mode = mpreid_tune("gemm_big", 1);
    return mode;
}
std::mutex g_prof_mu;
std::map<std::tuple<int, int, int, int>, ProfClass> g_prof;   // (epilogue, M, N, K)
} // namespace

extern "C" int mpreid_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return MPREID_OK;
}

extern "C" int mpreid_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &kv : g_prof)
        for (auto &e : kv.second.ev) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    g_prof.clear();
    return MPREID_OK;
}

// Fills up to `cap` entries, sorted by total time descending; returns the number of classes.
extern "C" int mpreid_profile_query(mpreid_profile_entry *out, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::vector<mpreid_profile_entry> v;
    for (auto &kv : g_prof) {
        mpreid_profile_entry e{};
        e.epilogue = std::get<0>(kv.first);
        e.n = std::get<2>(kv.first);
        e.k = std::get<3>(kv.first);
        e.m = kv.second.m;
        e.launches = (int64_t)kv.second.ev.size();
        e.flops_total = kv.second.flops_total;
        double tot = 0.0;
        for (auto &p : kv.second.ev) {
            if (hipEventSynchronize(p.second) != hipSuccess) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) tot += ms;
        }
        e.total_ms = tot;
        v.push_back(e);
    }
    std::sort(v.begin(), v.end(), [](const mpreid_profile_entry &a, const mpreid_profile_entry &b) {
        return a.total_ms > b.total_ms;
    });
    for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

struct ProfToken {
    hipEvent_t e0;
};
void *mpreid_prof_begin(hipStream_t stream) {
    if (!g_prof_on) return nullptr;
    ProfToken *t = new ProfToken{};
    if (hipEventCreate(&t->e0) != hipSuccess) {
        delete t;
        return nullptr;
    }
    if (hipEventRecord(t->e0, stream) != hipSuccess) {
        (void)hipEventDestroy(t->e0);
        delete t;
        return nullptr;
    }
    return t;
}
void mpreid_prof_end(void *token, hipStream_t stream, int cls, int64_t m, int n, int k, double work) {
    if (!token) return;
    ProfToken *t = static_cast<ProfToken *>(token);
    hipEvent_t e1 = nullptr;
    if (hipEventCreate(&e1) == hipSuccess && hipEventRecord(e1, stream) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ProfClass &pc = g_prof[std::make_tuple(cls, (int)m, n, k)];
        pc.ev.emplace_back(t->e0, e1);
        pc.m = std::max<int64_t>(pc.m, m);
        pc.flops_total += work;
    } else {   // not filed: both events are ours to destroy
        if (e1) (void)hipEventDestroy(e1);
        (void)hipEventDestroy(t->e0);
    }
    delete t;
}

// CU count of the current device (cached per device ordinal; benign race: every writer stores the same value)
static int current_device_cus(int *cus) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    static int cus_of[64] = {0};
    if (dev >= 0 && dev < 64 && cus_of[dev] > 0) {
        *cus = cus_of[dev];
        return MPREID_OK;
    }
    int n = 0;
    HIP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cus_of[dev] = n;
    *cus = n;
    return MPREID_OK;
}

template <int EPI>
static int launch_one(const GemmArgs &a_in, hipStream_t stream) {
    const GemmArgs &a = a_in;
    static PerDeviceOnce attr_once;
    {
        const int rc = attr_once.run([]() -> int {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_kernel<EPI>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES));
            return MPREID_OK;
        });
        if (rc) return rc;
    }
    constexpr bool HAS_BIG = (EPI == GE_F32 || EPI == GE_BIAS_F16 || EPI == GE_BIAS_RES || EPI == GE_BIAS_GELU ||
                              EPI == GE_EUCLID || EPI == GE_PATCH || EPI == GE_BIAS_RELU || EPI == GE_BIAS_ADD_RELU ||
                              EPI == GE_CAND || gemm_epi_is_split(EPI));
    if constexpr (gemm_epi_is_split(EPI)) {
        if (a.kseg <= 0 || a.kseg % GBK || a.K != 2 * a.kseg) {
            mpreid_set_error("gemm_f16: split epilogue %d needs K == 2 * kseg, kseg a multiple of %d (K=%d kseg=%d)", EPI, GBK,
                             a.K, a.kseg);
            return MPREID_ERR_ARG;
        }
    }
    if constexpr (EPI == GE_CAND) {
        if (a.M % BBM || a.N % BBN) {
            mpreid_set_error("gemm_f16: the candidate epilogue needs M, N multiples of %d", BBM);
            return MPREID_ERR_ARG;
        }
    }
    const int bm = big_mode();
    const bool use_big = HAS_BIG && (EPI == GE_CAND || (bm > 0 && (a.M % BBM == 0) && (a.N % BBN == 0) &&
                                                        (bm >= 2 || (int64_t)(a.M / BBM) * (a.N / BBN) >= 128)));
    if constexpr (HAS_BIG) {
        static PerDeviceOnce big_attr_once;
        const int rc = big_attr_once.run([]() -> int {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));
            return MPREID_OK;
        });
        if (rc) return rc;
    }
    const int tiles_m = use_big ? a.M / BBM : a.M / GBM, tiles_n = use_big ? a.N / BBN : a.N / GBN;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_prof_on) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, stream));
    }
    // MPREID_TUNE dist_sym_p2: 0 = never, 1 = symmetric stored distance problems of at least 16 tile rows (default),
    // 2 = every symmetric problem whose padded size allows it (tests)
    bool use_p2 = false, p2_full = false;
    if constexpr (EPI == GE_EUCLID) {
        static const int p2_mode = mpreid_tune("dist_sym_p2", 1);
        use_p2 = a.sym && p2_mode > 0 && a.M == a.N && (a.A == a.W || p2_mode >= 3) && (a.M % PBM == 0) && (a.K % PBK == 0) &&
                 (p2_mode >= 2 || a.M / PBM >= 16);   // (3: also the 3-term split operands, A != W: measured slower, see DESIGN)
        // MPREID_TUNE dist_p2_full: the two-workgroups-per-CU kernel for TWO tensors (stored distances, one-pass fp16 and
        // 3-term split operands): 0 never (default: measured slower), 1 problems of at least 16 x 32 tiles, 2 whenever
        // the padded sizes allow (tests)
        static const int full_mode = mpreid_tune("dist_p2_full", 0);
        if (!use_p2 && !a.sym && full_mode > 0 && a.out && (a.M % PBM == 0) && (a.N % PBN == 0) && (a.K % PBK == 0) &&
            (full_mode >= 2 || (a.M / PBM >= 16 && a.N / PBN >= 32)))
            use_p2 = p2_full = true;
    }
    if (use_p2) {
        if constexpr (EPI == GE_EUCLID) {
            static PerDeviceOnce p2_once;
            const int rc = p2_once.run([]() -> int {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(dist_sym_p2_kernel<0>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES));
                return MPREID_OK;
            });
            if (rc) return rc;
            int cus = 0;
            if (const int rcc = current_device_cus(&cus)) return rcc;
            const int grid = std::max(8, (2 * cus) & ~7);
            static const int naps_tune = mpreid_tune("dist_sym_p2_naps", -1);
            const int naps = naps_tune >= 0 ? naps_tune : 0;   // (experiments) late start of every CU's second workgroup, units of s_sleep 8
            [[maybe_unused]] static const int abl = mpreid_tune("dist_sym_p2_abl", 0);
            GemmArgs a = a_in;
            static const int gridt = mpreid_tune("dist_sym_p2_grid", 0);
            static const int strip = std::max(1, mpreid_tune("dist_sym_p2_strip", 8));   // tile rows per strip of the walk
            const dim3 gdim((unsigned)(gridt ? gridt : grid));
#ifdef MPREID_ABLATION
            if (const char *sp = getenv("MPREID_GEMM_STAMPS")) a.stamps = reinterpret_cast<unsigned long long *>(strtoull(sp, nullptr, 16));
#define MPREID_P2_CASE(V)                                                                                      \
    case V:                                                                                                    \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(dist_sym_p2_kernel<V>),                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES));                 \
        hipLaunchKernelGGL(dist_sym_p2_kernel<V>, gdim, dim3(256), P_LDS_BYTES, stream, a, a.M / PBM, a.N / PBN, naps, strip); \
        break;
            if (p2_full) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(dist_sym_p2_kernel<0, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES));
                hipLaunchKernelGGL((dist_sym_p2_kernel<0, true>), gdim, dim3(256), P_LDS_BYTES, stream, a, a.M / PBM, a.N / PBN, naps, strip);
            } else
            switch (abl) {
                MPREID_P2_CASE(1) MPREID_P2_CASE(2) MPREID_P2_CASE(3) MPREID_P2_CASE(4) MPREID_P2_CASE(8) MPREID_P2_CASE(9)
                MPREID_P2_CASE(16) MPREID_P2_CASE(24) MPREID_P2_CASE(32) MPREID_P2_CASE(33)
            default:
                hipLaunchKernelGGL(dist_sym_p2_kernel<0>, gdim, dim3(256), P_LDS_BYTES, stream, a, a.M / PBM, a.N / PBN, naps, strip);
            }
#undef MPREID_P2_CASE
#else
            if (p2_full) {
                static PerDeviceOnce p2f_once;
                const int rcf = p2f_once.run([]() -> int {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(dist_sym_p2_kernel<0, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES));
                    return MPREID_OK;
                });
                if (rcf) return rcf;
                hipLaunchKernelGGL((dist_sym_p2_kernel<0, true>), gdim, dim3(256), P_LDS_BYTES, stream, a, a.M / PBM, a.N / PBN, naps, strip);
            } else {
                hipLaunchKernelGGL(dist_sym_p2_kernel<0>, gdim, dim3(256), P_LDS_BYTES, stream, a, a.M / PBM, a.N / PBN, naps, strip);
            }
#endif
        }
    } else if (use_big) {
        if constexpr (HAS_BIG) {
            static const int dbg = mpreid_ablation_env("MPREID_GEMM_DBG");
            // persistent: one workgroup per CU (the kernel owns the CU's whole LDS), each walking tiles
            int big_cus = 0;
            if (const int rcc = current_device_cus(&big_cus)) return rcc;
            static const int stag_all = mpreid_tune("gemm_stagger_all", 0);
            // tile order inside an XCD's row group (bit-identical results; tools/walk_ab.sh on one device, M = 65 536, split
            // operands): column-fastest takes FC2 (N = 768, 12 KB operand rows) from 711 / 709 to 693 / 693 us and leaves
            // out-proj (3 KB rows) 0.5-1 % SLOWER (219.6 / 217.7 -> 221.5 / 220.7); groups of 4 tile rows (an A panel that fits
            // the 4 MB L2) change nothing (QKV 552.6 / 554.0, FC1 755 / 756), groups of 16 lose 1-3 %.  auto (-1) = column-fastest
            // for narrow outputs with operand rows of >= 8 KB, i.e. the split FC2.
            static const int walk_tune = mpreid_tune("gemm_walk", -1);   // 0 row-fastest; 1 column-fastest when tiles_n <= 4; 2 always; 3 / 4: groups of 4 / 16 rows
            GemmArgs a = a_in;
            if (stag_all > 0) {
                a.stagger = stag_all;
                a.stagger_mode = 1;
            }
            a.walk = walk_tune >= 3 ? walk_tune
                   : (walk_tune == 2 || (walk_tune == 1 && tiles_n <= 4) || (walk_tune < 0 && tiles_n <= 4 && (int64_t)a.K * 2 >= 8192)) ? 1 : 0;
            // row groups of 4 tile rows when groups of 8 do not divide among the XCDs but groups of 4 do (e.g. tiles_m = 224):
            // bit 3 of walk
            if (walk_tune < 3 && tiles_m % 64 != 0 && tiles_m % 32 == 0) a.walk |= 8;
            static const int ragged_tune = mpreid_tune("gemm_ragged", 1);   // 0: round 5's rule (owned walk only for exact divisions)
            if (!ragged_tune) a.walk |= 16;
            const unsigned total_tiles = (unsigned)tiles_m * (unsigned)tiles_n;
#ifdef MPREID_ABLATION
            if (const char *gg = getenv("MPREID_GEMM_GRID")) big_cus = atoi(gg);   // (ablation) fewer workgroups than CUs
#endif
            static const int grid_tune = mpreid_tune("gemm_grid", 0);   // experiment: persistent workgroups on fewer CUs (same bits)
            if (grid_tune > 0 && grid_tune < big_cus) big_cus = grid_tune;
            const dim3 grid(total_tiles < (unsigned)big_cus ? total_tiles : (unsigned)big_cus);
#ifdef MPREID_ABLATION
            if (dbg == 64 && gemm_epi_is_split(EPI)) {   // operand panels aliased onto two (L2-resident): see set_tile
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI, 64>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));
                hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, 64>), grid, dim3(512), B_LDS_TOTAL, stream, a, tiles_m, tiles_n);
            } else if (dbg == 32 && EPI != GE_EUCLID && EPI != GE_BIAS_GELU) {   // per-tile phase stamps for any epilogue (tools/gemm_tile_stamps.py)
                GemmArgs as = a;
                const char *sp = getenv("MPREID_GEMM_STAMPS");
                as.stamps = sp ? reinterpret_cast<unsigned long long *>(strtoull(sp, nullptr, 16)) : nullptr;
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI, 32>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));
                hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, 32>), grid, dim3(512), B_LDS_TOTAL, stream, as, tiles_m, tiles_n);
            } else
#endif
            if constexpr (EPI == GE_BIAS_F16) {
#define MPREID_DBG_CASE(D)                                                                                  \
    case D: {                                                                                               \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI, D>),            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));              \
        hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, D>), grid, dim3(512), B_LDS_TOTAL, stream, a, tiles_m, tiles_n); \
        break;                                                                                              \
    }
                switch (dbg) {
                    MPREID_DBG_CASE(1) MPREID_DBG_CASE(2) MPREID_DBG_CASE(3) MPREID_DBG_CASE(4) MPREID_DBG_CASE(5)
                    MPREID_DBG_CASE(7) MPREID_DBG_CASE(8) MPREID_DBG_CASE(9) MPREID_DBG_CASE(15) MPREID_DBG_CASE(16)
                    MPREID_DBG_CASE(24)
                default:
                    hipLaunchKernelGGL(gemm_f16_big_kernel<EPI>, grid, dim3(512), B_LDS_TOTAL, stream, a, tiles_m, tiles_n);
                }
#undef MPREID_DBG_CASE
            } else if constexpr (EPI == GE_EUCLID || EPI == GE_BIAS_GELU) {
                if (dbg == 800) {   // stamps + raw stores
                    GemmArgs as = a;
                    const char *sp = getenv("MPREID_GEMM_STAMPS");
                    as.stamps = sp ? reinterpret_cast<unsigned long long *>(strtoull(sp, nullptr, 16)) : nullptr;
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI, 800>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));
                    hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, 800>), grid, dim3(512), B_LDS_TOTAL, stream, as, tiles_m, tiles_n);
                } else if (dbg == 32) {   // (ablation builds) per-tile phase stamps; MPREID_GEMM_STAMPS = device pointer (hex)
                    GemmArgs as = a;
                    const char *sp = getenv("MPREID_GEMM_STAMPS");
                    as.stamps = sp ? reinterpret_cast<unsigned long long *>(strtoull(sp, nullptr, 16)) : nullptr;
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI, 32>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));
                    hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, 32>), grid, dim3(512), B_LDS_TOTAL, stream, as, tiles_m, tiles_n);
                } else if (dbg >= 128 && dbg <= 6 * 128 && dbg % 128 == 0) {   // (ablation builds) store cache policies
#define MPREID_FLAV_CASE(F)                                                                                         \
    case F * 128:                                                                                                    \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI, F * 128>),               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));                       \
        hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, F * 128>), grid, dim3(512), B_LDS_TOTAL, stream, a, tiles_m, tiles_n); \
        break;
                    switch (dbg) { MPREID_FLAV_CASE(1) MPREID_FLAV_CASE(2) MPREID_FLAV_CASE(3) MPREID_FLAV_CASE(4) MPREID_FLAV_CASE(5) MPREID_FLAV_CASE(6) }
#undef MPREID_FLAV_CASE
                } else if (dbg == 16) {   // (ablation builds) the unpaired DMA schedule
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI, 16>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));
                    hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, 16>), grid, dim3(512), B_LDS_TOTAL, stream, a, tiles_m, tiles_n);
                } else if (EPI == GE_EUCLID && a.sym) {   // symmetric problem: the instance with the mirrored stores
                    if constexpr (EPI == GE_EUCLID) {
                        static PerDeviceOnce sym_once;
                        const int rc = sym_once.run([]() -> int {
                            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_big_kernel<EPI, 0, true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_TOTAL));
                            return MPREID_OK;
                        });
                        if (rc) return rc;
                        hipLaunchKernelGGL((gemm_f16_big_kernel<EPI, 0, true>), grid, dim3(512), B_LDS_TOTAL, stream, a, tiles_m, tiles_n);
                    }
                } else {
                    hipLaunchKernelGGL(gemm_f16_big_kernel<EPI>, grid, dim3(512), B_LDS_TOTAL, stream, a, tiles_m, tiles_n);
                }
            } else {
                hipLaunchKernelGGL(gemm_f16_big_kernel<EPI>, grid, dim3(512), B_LDS_TOTAL, stream, a, tiles_m, tiles_n);
            }
        }
    } else {
        hipLaunchKernelGGL(gemm_f16_kernel<EPI>, dim3((unsigned)tiles_m * (unsigned)tiles_n), dim3(256), G_LDS_BYTES,
                               stream, a, tiles_m, tiles_n);
    }
    LAUNCH_CHECK();
    if (e0) {
        HIP_TRY(hipEventRecord(e1, stream));
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ProfClass &pc = g_prof[std::make_tuple(EPI, a.M, a.N, a.K)];
        pc.ev.emplace_back(e0, e1);
        pc.m = std::max<int64_t>(pc.m, a.M);
        // split mode: three products per logical multiply-add are EXECUTED on the matrix cores
        pc.flops_total += 2.0 * (double)a.M * (double)a.N * (gemm_epi_is_split(EPI) ? 3.0 * (double)a.kseg : (double)a.K);
    }
    return MPREID_OK;
}

int launch_gemm_f16(const GemmArgs &a, int epi, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.M % GBM || a.N % GBN || a.K % GBK) {
        mpreid_set_error("gemm_f16: M=%d N=%d K=%d must be positive multiples of %d/%d/%d", a.M, a.N, a.K, GBM, GBN,
                         GBK);
        return MPREID_ERR_ARG;
    }
    if (epi == GE_S_BIAS_RELU_PAIR && (a.pair_c <= 0 || a.pair_c % 64 || a.ldo != 2 * (int64_t)a.pair_c)) {
        mpreid_set_error("gemm_f16: the pair epilogue needs pair_c %% 64 == 0 and ldo == 2 * pair_c (pair_c=%d ldo=%lld)", a.pair_c,
                         (long long)a.ldo);
        return MPREID_ERR_ARG;
    }
    if (epi == GE_S_BIAS_RES_PAIR && (!a.pair_out || a.pair_c != a.N || a.pair_c % 64 || a.ldo % 4 || (reinterpret_cast<uintptr_t>(a.out) & 15) ||
                                      (reinterpret_cast<uintptr_t>(a.pair_out) & 15))) {
        mpreid_set_error("gemm_f16: the residual + pair epilogue needs pair_out, pair_c == N, ldo %% 4 == 0, 16-byte aligned tensors");
        return MPREID_ERR_ARG;
    }
    switch (epi) {
    case GE_F32: return launch_one<GE_F32>(a, stream);
    case GE_BIAS_F16: return launch_one<GE_BIAS_F16>(a, stream);
    case GE_BIAS_RES: return launch_one<GE_BIAS_RES>(a, stream);
    case GE_BIAS_GELU: return launch_one<GE_BIAS_GELU>(a, stream);
    case GE_PATCH: return launch_one<GE_PATCH>(a, stream);
    case GE_EUCLID: return launch_one<GE_EUCLID>(a, stream);
    case GE_COSINE: return launch_one<GE_COSINE>(a, stream);
    case GE_BIAS_RELU: return launch_one<GE_BIAS_RELU>(a, stream);
    case GE_BIAS_ADD_RELU: return launch_one<GE_BIAS_ADD_RELU>(a, stream);
    case GE_CAND: return launch_one<GE_CAND>(a, stream);
    case GE_S_BIAS_F32: return launch_one<GE_S_BIAS_F32>(a, stream);
    case GE_S_BIAS_RES: return launch_one<GE_S_BIAS_RES>(a, stream);
    case GE_S_BIAS_GELU: return launch_one<GE_S_BIAS_GELU>(a, stream);
    case GE_S_PATCH: return launch_one<GE_S_PATCH>(a, stream);
    case GE_S_BIAS_RELU_PAIR: return launch_one<GE_S_BIAS_RELU_PAIR>(a, stream);
    case GE_S_BIAS_RES_PAIR: return launch_one<GE_S_BIAS_RES_PAIR>(a, stream);
    }
    mpreid_set_error("gemm_f16: unknown epilogue %d", epi);
    return MPREID_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// fp32 -> fp16 helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float *__restrict__ x, _Float16 *__restrict__ y,
                                                           int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i);
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 o = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<h4 *>(y + i) = o;
    } else {
        for (int64_t k = i; k < n; ++k) y[k] = (_Float16)x[k];
    }
}

// x [n][d] fp32 -> y [n_pad][d_pad] fp16, zero padded
__global__ __launch_bounds__(256) void cast_pad_kernel(const float *__restrict__ x, int64_t n, int d,
                                                       _Float16 *__restrict__ y, int64_t n_pad, int d_pad) {
    const int64_t row = blockIdx.x;
    for (int k = threadIdx.x; k < d_pad; k += 256) {
        float v = 0.f;
        if (row < n && k < d) v = x[row * (int64_t)d + k];
        y[row * (int64_t)d_pad + k] = (_Float16)v;
    }
}

extern "C" int mpreid_cast_f32_to_f16(const float *x, void *y, int64_t n, mpreid_stream_t stream) {
    ARG_CHECK(x && y && n >= 0);
    if (n == 0) return MPREID_OK;
    ARG_CHECK(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0));
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x,
                       (_Float16 *)y, n);
    LAUNCH_CHECK();
    return MPREID_OK;
}

extern "C" int mpreid_gemm_f16_nt(const void *a, const void *b, float *c, int64_t m, int64_t n, int64_t k,
                                  mpreid_stream_t stream) {
    ARG_CHECK(a && b && c);
    GemmArgs g{};
    g.A = (const _Float16 *)a;
    g.W = (const _Float16 *)b;
    g.M = (int)m;
    g.N = (int)n;
    g.K = (int)k;
    g.out = c;
    g.ldo = n;
    return launch_gemm_f16(g, GE_F32, (hipStream_t)stream);
}

// GEMM with a selectable fused epilogue (0 f32, 1 +bias->f16, 2 +bias+residual (out fp32, in place),
// 3 +bias,QuickGELU->f16): unit tests and the GEMM micro-benchmark (tools/gemm_bench.py).
extern "C" int mpreid_gemm_f16_nt_ex(const void *a, const void *b, void *out, const float *bias, int64_t m, int64_t n,
                                     int64_t k, int epilogue, mpreid_stream_t stream) {
    ARG_CHECK(a && b && out && ((epilogue >= 0 && epilogue <= 3) || epilogue == GE_BIAS_RELU || epilogue == GE_BIAS_ADD_RELU) &&
              (epilogue == 0 || bias));
    GemmArgs g{};
    g.identity = (const _Float16 *)out;   // epilogue 8: the residual is what out holds on entry (updated in place)
    g.A = (const _Float16 *)a;
    g.W = (const _Float16 *)b;
    g.M = (int)m;
    g.N = (int)n;
    g.K = (int)k;
    g.out = out;
    g.ldo = n;
    g.bias = bias;
    return launch_gemm_f16(g, epilogue, (hipStream_t)stream);
}

// Split-precision GEMM (the linear layers of the encoder's `split` mode), exposed for unit tests and tools/gemm_bench.py:
// a2 [M][2*kseg], b2 [N][2*kseg] fp16 pairs [hi | lo]; epilogue 10 = out fp32 [M][N] = acc*oscale + bias, 11 = out fp32
// += acc*oscale + bias, 12 = out fp16 pair [M][2N] = hi | lo of quickgelu(acc*oscale + bias).
extern "C" int mpreid_gemm_f16_split_nt(const void *a2, const void *b2, void *out, const float *bias, int64_t m, int64_t n,
                                        int64_t kseg, float oscale, int epilogue, mpreid_stream_t stream) {
    ARG_CHECK(a2 && b2 && out && bias && epilogue >= GE_S_BIAS_F32 && epilogue <= GE_S_BIAS_GELU);
    GemmArgs g{};
    g.A = (const _Float16 *)a2;
    g.W = (const _Float16 *)b2;
    g.M = (int)m;
    g.N = (int)n;
    g.K = (int)(2 * kseg);
    g.kseg = (int)kseg;
    g.oscale = oscale;
    g.out = out;
    g.ldo = epilogue == GE_S_BIAS_GELU ? 2 * n : n;
    g.bias = bias;
    return launch_gemm_f16(g, epilogue, (hipStream_t)stream);
}

// x [rows][cols] fp32 -> y [rows][2*cols] fp16 pair: hi = fp16(x * scale), lo = fp16(x * scale - hi)   (scale: a power of two)
__global__ __launch_bounds__(256) void split_pack_kernel(const float *__restrict__ x, int64_t rows, int cols, float scale,
                                                         _Float16 *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (int64_t)cols) return;
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const float v = x[i] * scale;
    const _Float16 hi = (_Float16)v;
    y[r * 2 * cols + c] = hi;
    y[r * 2 * cols + cols + c] = (_Float16)(v - (float)hi);
}

extern "C" int mpreid_split_pack_f32(const float *x, int64_t rows, int cols, float scale, void *y, mpreid_stream_t stream) {
    ARG_CHECK(x && y && rows >= 0 && cols > 0);
    if (rows == 0) return MPREID_OK;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       cols, scale, (_Float16 *)y);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// ---------------------------------------------------------------------------------------------
// 3-term fp16 split: x * 2^e = hi + lo (both fp16, e per row so that |x * 2^e| < 2^10),
// q.g = 2^-(eq+eg) * (hi.hi' + lo.hi' + hi.lo') + O(2^-22) -- ONE ordinary GEMM over operands concatenated along K:
//   A' = [hi | lo | hi]   B' = [hi' | hi' | lo']   (K' = 3 * K)
// fp16 x fp16 products are exact in the fp32 accumulator, so the error is the dropped lo.lo' term, the rounding of lo
// and the fp32 accumulation: max |delta| <= 1e-6 on unit-norm rows (tests/test_gpu_distance.py), the level of the
// exact fp32 chain's own rounding error, at 3/16 of the fp32-MFMA time.  The scaling keeps lo out of the fp16
// subnormal range; it is undone exactly (powers of two) by rscale[m] * cscale[n] in the epilogue.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split3_pack_kernel(const float *__restrict__ x, const float *__restrict__ sqn,
                                                          int norm_is_sqrt, int64_t n, int d, _Float16 *__restrict__ y,
                                                          int d_pad, float *__restrict__ unscale, int is_b) {
    const int64_t row = blockIdx.x;
    float sc = 1.0f, un = 1.0f;
    if (row < n) {
        const float nrm = norm_is_sqrt ? sqn[row] : sqrtf(sqn[row]);
        if (nrm > 0.0f && nrm < 3.0e38f) {
            int ex;
            (void)frexpf(nrm, &ex);          // nrm < 2^ex, so |x_k| * 2^(10-ex) < 2^10
            sc = ldexpf(1.0f, 10 - ex);
            un = ldexpf(1.0f, ex - 10);
        }
    }
    if (threadIdx.x == 0) unscale[row] = un;
    _Float16 *yr = y + row * (int64_t)(3 * d_pad);
    for (int k = threadIdx.x; k < d_pad; k += 256) {
        float v = 0.f;
        if (row < n && k < d) v = x[row * (int64_t)d + k] * sc;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        yr[k] = hi;
        yr[d_pad + k] = is_b ? hi : lo;
        yr[2 * d_pad + k] = is_b ? lo : hi;
    }
}

// start stagger of the persistent distance GEMM (see GemmArgs::stagger); MPREID_TUNE gemm_stagger = ticks (default 0 = off)
static int euclid_stagger_ticks(int M, int N, int K) {
    static const int ticks = mpreid_tune("gemm_stagger", 0);
    (void)M; (void)N; (void)K;
    return ticks;
}

size_t mpreid_distance_split3_ws_bytes(int64_t nq, int64_t ng, int d) {
    const size_t dp = align_up((size_t)d, GBK);
    const size_t mp = align_up((size_t)nq, 256), np = align_up((size_t)ng, 256);
    return align_up(mp * 3 * dp * 2, 256) + align_up(np * 3 * dp * 2, 256) + align_up(mp * 4, 256) + align_up(np * 4, 256);
}

int mpreid_distance_f16_split3(const float *q, const float *g, int64_t nq, int64_t ng, int d, const float *qn,
                               const float *gn, float *out, int64_t ldo, int epi, void *ws, size_t ws_bytes,
                               hipStream_t stream) {
    if (ws_bytes < mpreid_distance_split3_ws_bytes(nq, ng, d)) {
        mpreid_set_error("split3 distance workspace too small");
        return MPREID_ERR_WORKSPACE;
    }
    const int dp = (int)align_up((size_t)d, GBK);
    const int64_t mp = (int64_t)align_up((size_t)nq, BBM), np = (int64_t)align_up((size_t)ng, BBN);
    char *p = (char *)ws;
    _Float16 *qh = (_Float16 *)p;
    p += align_up((size_t)mp * 3 * dp * 2, 256);
    _Float16 *gh = (_Float16 *)p;
    p += align_up((size_t)np * 3 * dp * 2, 256);
    float *qs = (float *)p;
    p += align_up((size_t)mp * 4, 256);
    float *gs = (float *)p;
    // qn / gn are squared norms for the Euclidean epilogue and norms for the cosine one (distance_common)
    hipLaunchKernelGGL(split3_pack_kernel, dim3((unsigned)mp), dim3(256), 0, stream, q, qn, epi != 0, nq, d, qh, dp, qs, 0);
    hipLaunchKernelGGL(split3_pack_kernel, dim3((unsigned)np), dim3(256), 0, stream, g, gn, epi != 0, ng, d, gh, dp, gs, 1);
    LAUNCH_CHECK();
    GemmArgs a{};
    a.A = qh;
    a.W = gh;
    a.M = (int)mp;
    a.N = (int)np;
    a.K = 3 * dp;
    a.out = out;
    a.ldo = ldo;
    a.aux = qn;
    a.aux2 = gn;
    a.m_valid = (int)nq;
    a.n_valid = (int)ng;
    a.rscale = qs;
    a.cscale = gs;
    // all-pairs distances of ONE set (same pointer, utils.metrics.euclidean_distance(f, f)): the persistent kernel computes
    // the tiles on or above the diagonal and stores the others as transposed copies -- half the matrix work (the 3-term sum
    // is not bit-symmetric computed both ways; either value is within the mode's 1e-6, and the copies are what they copy)
    a.sym = (epi == 0 && q == g && nq == ng) ? 1 : 0;
    a.stagger = euclid_stagger_ticks(a.M, a.N, a.K);
    return launch_gemm_f16(a, epi == 0 ? GE_EUCLID : GE_COSINE, stream);
}

size_t mpreid_distance_f16_ws_bytes(int64_t nq, int64_t ng, int d) {
    const size_t dp = align_up((size_t)d, GBK);
    return align_up(align_up((size_t)nq, 256) * dp * 2, 256) + align_up(align_up((size_t)ng, 256) * dp * 2, 256);
}

int mpreid_distance_f16_fast(const float *q, const float *g, int64_t nq, int64_t ng, int d, const float *qn,
                             const float *gn, float *out, int64_t ldo, int epi, void *ws, size_t ws_bytes,
                             hipStream_t stream) {
    if (ws_bytes < mpreid_distance_f16_ws_bytes(nq, ng, d)) {
        mpreid_set_error("fp16 distance workspace too small");
        return MPREID_ERR_WORKSPACE;
    }
    const int dp = (int)align_up((size_t)d, GBK);
    const int64_t mp = (int64_t)align_up((size_t)nq, BBM), np = (int64_t)align_up((size_t)ng, BBN);
    // all-pairs distances of ONE set (same pointer): one cast serves both operands and the kernel computes the tiles on or
    // above the diagonal only, storing each off-diagonal tile twice (the fp16 dot products are bit-symmetric: same bits
    // as the full computation)
    const bool sym = epi == 0 && q == g && nq == ng;
    _Float16 *qh = (_Float16 *)ws;
    _Float16 *gh = sym ? qh : (_Float16 *)((char *)ws + align_up((size_t)mp * dp * 2, 256));
    hipLaunchKernelGGL(cast_pad_kernel, dim3((unsigned)mp), dim3(256), 0, stream, q, nq, d, qh, mp, dp);
    if (!sym) hipLaunchKernelGGL(cast_pad_kernel, dim3((unsigned)np), dim3(256), 0, stream, g, ng, d, gh, np, dp);
    LAUNCH_CHECK();
    GemmArgs a{};
    a.A = qh;
    a.W = gh;
    a.M = (int)mp;
    a.N = (int)np;
    a.K = dp;
    a.out = out;
    a.ldo = ldo;
    a.aux = qn;
    a.aux2 = gn;
    a.m_valid = (int)nq;
    a.n_valid = (int)ng;
    a.sym = sym ? 1 : 0;
    a.stagger = euclid_stagger_ticks(a.M, a.N, a.K);
    return launch_gemm_f16(a, epi == 0 ? GE_EUCLID : GE_COSINE, stream);
}
