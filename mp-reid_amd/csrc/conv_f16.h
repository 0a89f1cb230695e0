// conv_f16.h — implicit-GEMM convolution over NHWC fp16 (see conv_f16.hip)
#pragma once
#include "common.h"

struct ConvArgs {
    const _Float16 *act;       // [B][H][W][C] NHWC, C % 64 == 0
    const _Float16 *wgt;       // [Npad][taps * C], k order (tap = kh*3+kw, c); BatchNorm folded in
    const float *bias;         // [Npad] folded BatchNorm shift
    const _Float16 *identity;  // [M][ldo] residual added before the ReLU, or nullptr
    _Float16 *out;             // [M][ldo]
    const _Float16 *zero_page; // >= 128 bytes of zeros (padding pixels, rows past M)
    int H, W, C;               // input = output spatial size (stride 1, pad taps/2), input channels
    int M;                     // B*H*W output pixels
    int N, Npad;               // output channels, padded to a multiple of 128 in wgt / bias
    int ldo;                   // row stride of out / identity in elements
    int taps;                  // 1 (1x1) or 9 (3x3, pad 1)
    int relu;
    // SPLIT form (the RN50 split tower's 3x3 convolutions, round 4): act = fp16 PAIRS [pixel][hi(C) | lo(C)] (row stride 2 C),
    // wgt = [Npad][taps][C / 64][hi(64) | lo(64)] of W * 2^e, and every (tap, 64 channels) block runs the three products
    // hi.hi' + lo.hi' + hi.lo' into the fp32 accumulators;
    // out32 fp32 [M][ldo] = acc * oscale + bias (no ReLU: the consumer's pack applies it)
    int split;
    float oscale;
    float *out32;
    _Float16 *out_pairs;       // SPLIT, optional (then out32 is not written): relu(acc * oscale + bias) as fp16 pairs
    int pair_c;                // [M][hi(pair_c) | lo(pair_c)] -- the A operand of the next layer's pair GEMM; pair_c % 64 == 0, >= N
};

int launch_conv_f16(const ConvArgs &a, hipStream_t stream);
