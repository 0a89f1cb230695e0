// gemm_f16.h — fp16-in / fp32-accumulate MFMA GEMM for gfx950:  C[M][N] = A[M][K] x W[N][K]^T
// Both operands are K-contiguous (activations row-major, weights in torch Linear layout), which is
// the natural MFMA operand order: no transposes anywhere.
//
//   tile      128 x 128 per 256-thread workgroup (2x2 waves, 64x64 per wave = 4x4 MFMA 16x16x32)
//   K step    64 halfs (128-byte LDS rows); A and B tiles double-buffered: 2 x 32 KB = 64 KB LDS,
//             so two workgroups share a CU and overlap each other's barriers
//   staging   global_load_lds_dwordx4 (LDS-DMA): one wave-instruction = 8 rows x 128 B, no VGPRs;
//             LDS image is lane-linear, the bank-conflict swizzle (16-byte chunk ^= row & 7) is
//             applied to the per-lane SOURCE address and again on the ds_read_b128 address
//             (cdna_hip_programming.md §5.4 rule 21, T2)
//   loop      issue DMA for tile t+1 -> ds_read + 32 MFMA on tile t -> vmcnt(0) + barrier
//   epilogue  fused (bias / QuickGELU / residual / positional embedding / distance), fp16
//             outputs are transposed through LDS so that every store is a full 128-byte row piece
//
// Requirements: M, N multiples of 128 and K a multiple of 64 at the buffer level (callers pad);
// logical bounds are passed separately where the epilogue needs them.
#pragma once
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int G_TILE_BYTES = GBM * GBK * 2;        // 16 KB per operand tile
constexpr int G_STAGE_BYTES = 2 * G_TILE_BYTES;    // A + B
constexpr int G_LDS_BYTES = 2 * G_STAGE_BYTES;     // double buffered: 64 KB

enum GemmEpi {
    GE_F32 = 0,        // C fp32 = acc
    GE_BIAS_F16 = 1,   // out fp16 = acc + bias[n]
    GE_BIAS_RES = 2,   // x fp32 += acc + bias[n]                       (out-proj, MLP c_proj)
    GE_BIAS_GELU = 3,  // out fp16 = quickgelu(acc + bias[n])           (MLP c_fc)
    GE_PATCH = 4,      // x fp32[b*L + 1 + p][n] = acc + pos[1 + p][n]  (patch embed; m = b*P + p)
    GE_EUCLID = 5,     // out fp32 = fmaf(-2, acc, an[m] + bn[n])       (bounds checked)
    GE_COSINE = 6,     // out fp32 = acos(clip(acc / (an[m]*bn[n])))    (bounds checked)
    GE_BIAS_RELU = 7,  // out fp16 = relu(acc + bias[n])                (RN50 1x1 conv + folded BN + ReLU)
    GE_BIAS_ADD_RELU = 8, // out fp16 = relu(acc + bias[n] + identity[m][n]), one rounding (Bottleneck conv3)
    GE_CAND = 9,       // nothing is stored: d = fmaf(-2, acc, an[m] + bn[n]) is compared with per-row thresholds and the
                       // few (row, col, d) that pass are appended to per-row candidate lists (re-ranking: the N x N
                       // distance matrix is never materialised; csrc/rerank2.hip)
    // ---- "split" precision mode of the encoder (MODEL.ENCODER_PRECISION: split): both operands are fp16 PAIRS
    // x = hi + lo stored as rows [hi(kseg) | lo(kseg)], and the k-loop runs the three products hi.hi' + lo.hi' + hi.lo'
    // (3 * kseg deep) into the one fp32 accumulator: fp32-grade linear layers on the fp16 matrix cores.  acc is
    // multiplied by GemmArgs::oscale (the exact power of two that undoes the weight scaling) in the epilogue.
    GE_S_BIAS_F32 = 10,  // out fp32 = acc * oscale + bias[n]                        (in_proj: q | k | v for the split attention)
    GE_S_BIAS_RES = 11,  // x fp32 += acc * oscale + bias[n]                         (out-proj, MLP c_proj)
    GE_S_BIAS_GELU = 12, // out fp16 pair [M][2N]: hi | lo of quickgelu(acc * oscale + bias[n])   (MLP c_fc)
    GE_S_PATCH = 13,     // x fp32[b*L + 1 + p][n] = acc * oscale + pos[1 + p][n]    (patch embed)
    GE_S_BIAS_RELU_PAIR = 14   // out fp16 pair [M][2 * pair_c]: hi | lo of relu(acc * oscale + bias[n]) for the columns n < pair_c (the
                               // others are padding and are not stored): the A operand of the NEXT pair convolution, written by
                               // its producer (RN50 split tower: conv1 -> conv2); ldo = 2 * pair_c
    , GE_S_BIAS_RES_PAIR = 15  // GE_S_BIAS_RES (x fp32 += acc * oscale + bias[n], relu_x honoured) that ALSO writes relu(x) as fp16
                               // pairs [M][2 * pair_c] to pair_out (pair_c == N): a ResNet block output, stored once as the fp32
                               // identity of the next block and once as the pair operand of its first convolution
};
constexpr bool gemm_epi_is_split(int epi) { return epi >= GE_S_BIAS_F32 && epi <= GE_S_BIAS_RES_PAIR; }

struct GemmArgs {
    const _Float16 *A;   // [M][K]
    const _Float16 *W;   // [N][K]
    int M, N, K;         // padded sizes (multiples of 128 / 128 / 64); K is also the row stride of A and W in halfs
    int kseg;            // GE_S_* only: logical K of one operand half (K == 2 * kseg: rows are [hi | lo]); else 0
    float oscale;        // GE_S_* only: acc is multiplied by this before the epilogue arithmetic
    void *out;           // fp16 or fp32 output, row stride ldo elements
    int64_t ldo;
    const float *bias;   // [N]                       (GE_BIAS_*)
    const float *aux;    // pos emb [L][N] (GE_PATCH) / an [M] (distance)
    const float *aux2;   // bn [N] (distance)
    int m_valid, n_valid; // logical bounds for the bounds-checked epilogues
    int P, L;            // GE_PATCH: patches per image, tokens per image
    const _Float16 *identity; // GE_BIAS_ADD_RELU: fp16 [M][ldo] residual
    const float *rscale;  // distance epilogues, optional: acc is multiplied by rscale[m] * cscale[n] (exact powers of
    const float *cscale;  // two undoing the per-row operand scaling of the 3-term split mode) before the epilogue
    // GE_CAND: row m keeps column n when d <= tlo[m] (list_lo) or d >= thi[m] (list_hi); with sym != 0 (A == W, only
    // tiles with tn >= tm are computed) the transposed pair is tested as well: column n keeps row m.  tlo / thi are
    // padded to M (= N) entries with -inf / +inf.  Lists: uint2 {partner index, float bits of d}, row stride cap_*;
    // cnt_* count every append (entries past the capacity are dropped: the consumer sees cnt > cap and falls back).
    const float *tlo, *thi;
    unsigned *cnt_lo, *cnt_hi;
    uint2 *list_lo, *list_hi;
    int cap_lo, cap_hi, sym;
    int stagger;          // persistent kernel: workgroup b starts (b mod 256) / 256 * stagger ticks of the 100 MHz
                          // real-time counter late, so that the CUs reach their store-heavy epilogues at different
                          // times instead of all at once (0 = off)
    int stagger_mode;     // 1: eight phases by the CU's slot inside its XCD instead ((b >> 3) mod 8) / 8
    int walk;             // persistent kernel, XCD-owned row groups: 1 = column-fastest tile order inside a group (set by the launcher)
    int pair_c;           // GE_S_BIAS_RELU_PAIR / GE_S_BIAS_RES_PAIR: half the row length of the pair output (hi at n, lo at pair_c + n); % 64 == 0
    _Float16 *pair_out;   // GE_S_BIAS_RES_PAIR: the pair copy (out stays the fp32 tensor)
    int relu_x;           // GE_S_BIAS_RES only: the destination holds a PRE-activation (a ResNet block input whose ReLU is pending):
                          // x = max(x, 0) + acc * oscale + bias -- saves the producer-side pass that would write the ReLU back
    unsigned long long *stamps;   // ablation builds (DBG bit 32): [workgroup][32 tiles][8] real-time stamps, else null
};

int launch_gemm_f16(const GemmArgs &a, int epi, hipStream_t stream);
