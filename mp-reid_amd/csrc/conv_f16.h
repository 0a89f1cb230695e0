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
};

int launch_conv_f16(const ConvArgs &a, hipStream_t stream);
