// preprocess.hip — the data-format steps on either side of the encoder:
//
//   mpreid_resize_bilinear_u8   T.Resize(cfg.INPUT.SIZE_TEST) of val_transforms (datasets/make_dataloader.py:57-58).
//       torchvision 0.19.1 hands the PIL image to Image.resize(size[::-1], BILINEAR); the arithmetic is Pillow's
//       8-bit two-pass resample (src/libImaging/Resample.c, pinned 10.4.0 in requirements.txt:122): triangle
//       filter whose support widens with the down-scale factor, coefficients normalised in double and rounded to
//       22-bit fixed point, horizontal pass into an 8-bit image, then the vertical pass.  Integer work after the
//       coefficients, which are computed here in IEEE double exactly as on the host -> bit-exact with Pillow
//       (tests/golden/resize.npz, oracle orc_resize_bilinear_u8).
//       Ragged batch: images packed back to back in one buffer, (offset, h, w) per image.  HBM-bound byte work
//       (a 128x64 image is 24 KB in, 96 KB out, against 21 GFLOP of encoding): one thread per output pixel,
//       coalesced along the output row, no LDS.
//
//   mpreid_tta_mean_f32   processor/processor_uniprompt_stage2.py:636-640: torch.stack(feat_list).mean(0), then
//       F.normalize(p=2, dim=1) when TEST.FEAT_NORM is set.
#include "common.h"
#include "../../include/mpreid.h"

namespace {

constexpr int RS_BITS = 22;

__device__ __forceinline__ double rs_tri(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

struct RsTaps {
    int lo, cnt;
    double center, ss, ww;
};

// Resample.c precompute_coeffs() for one output position (in0 = 0, in1 = in_size)
__device__ __forceinline__ RsTaps rs_taps(int in_size, int out_size, int xx) {
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    RsTaps t;
    t.center = 0.0 + (xx + 0.5) * scale;
    t.ss = 1.0 / filterscale;
    int lo = (int)(t.center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(t.center + support + 0.5);
    if (hi > in_size) hi = in_size;
    t.lo = lo;
    t.cnt = hi - lo;
    double ww = 0.0;
    for (int x = 0; x < t.cnt; ++x) ww += rs_tri((x + lo - t.center + 0.5) * t.ss);
    t.ww = ww;
    return t;
}

// normalize_coeffs_8bpc(): the triangle weights are never negative
__device__ __forceinline__ int rs_coeff(const RsTaps &t, int x) {
    double w = rs_tri((x + t.lo - t.center + 0.5) * t.ss);
    if (t.ww != 0.0) w = w / t.ww;
    return (int)(0.5 + w * (double)(1 << RS_BITS));
}

__device__ __forceinline__ unsigned char rs_clip8(int acc) {
    const int v = acc >> RS_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src image b [h][w][3] -> tmp [b][max_in_h][out_w][3]
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char *__restrict__ src,
                                                       const int64_t *__restrict__ offsets,
                                                       const int32_t *__restrict__ hw, int max_in_h, int out_w,
                                                       unsigned char *__restrict__ tmp) {
    const int b = blockIdx.y;
    const int h = hw[2 * b], w = hw[2 * b + 1];
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)h * out_w) return;
    const int y = (int)(gid / out_w), xx = (int)(gid % out_w);
    const RsTaps t = rs_taps(w, out_w, xx);
    const unsigned char *row = src + offsets[b] + ((int64_t)y * w + t.lo) * 3;
    int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < t.cnt; ++x) {
        const int k = rs_coeff(t, x);
        a0 += (int)row[x * 3 + 0] * k;
        a1 += (int)row[x * 3 + 1] * k;
        a2 += (int)row[x * 3 + 2] * k;
    }
    unsigned char *o = tmp + (((int64_t)b * max_in_h + y) * out_w + xx) * 3;
    o[0] = rs_clip8(a0);
    o[1] = rs_clip8(a1);
    o[2] = rs_clip8(a2);
}

// vertical pass: tmp -> dst [b][out_h][out_w][3]
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char *__restrict__ tmp,
                                                       const int32_t *__restrict__ hw, int max_in_h, int out_h,
                                                       int out_w, unsigned char *__restrict__ dst) {
    const int b = blockIdx.y;
    const int h = hw[2 * b];
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)out_h * out_w) return;
    const int yy = (int)(gid / out_w), xx = (int)(gid % out_w);
    const RsTaps t = rs_taps(h, out_h, yy);
    const unsigned char *col = tmp + (((int64_t)b * max_in_h + t.lo) * out_w + xx) * 3;
    int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < t.cnt; ++y) {
        const int k = rs_coeff(t, y);
        const unsigned char *p = col + (int64_t)y * out_w * 3;
        a0 += (int)p[0] * k;
        a1 += (int)p[1] * k;
        a2 += (int)p[2] * k;
    }
    unsigned char *o = dst + (((int64_t)b * out_h + yy) * out_w + xx) * 3;
    o[0] = rs_clip8(a0);
    o[1] = rs_clip8(a1);
    o[2] = rs_clip8(a2);
}

// one wave per row: out = (((f0 + f1) + f2) + ...) / n, optionally / max(||.||_2, 1e-12)
// The squared norm uses the same order as sqnorm_kernel / l2_normalize_kernel (distance.hip): 64 lane-strided
// fmaf chains, butterfly 32..1.
__global__ __launch_bounds__(256) void tta_mean_kernel(const float *__restrict__ feats, int n_views, int64_t rows,
                                                       int dim, int normalize, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t vstride = rows * dim;
    const float *f = feats + row * dim;
    float *o = out + row * dim;
    float acc = 0.f;
    for (int d = lane; d < dim; d += 64) {
        float s = f[d];
        for (int v = 1; v < n_views; ++v) s = s + f[v * vstride + d];
        s = __fdiv_rn(s, (float)n_views);
        o[d] = s;
        acc = fmaf(s, s, acc);
    }
    if (!normalize) return;
    acc = wave_bfly_add(acc);
    const float nrm = fmaxf(sqrtf(acc), 1e-12f);
    for (int d = lane; d < dim; d += 64) o[d] = __fdiv_rn(o[d], nrm);
}

} // namespace

extern "C" size_t mpreid_resize_workspace_bytes(int batch, int max_in_h, int out_w) {
    if (batch <= 0 || max_in_h <= 0 || out_w <= 0) return 0;
    return align_up((size_t)batch * max_in_h * out_w * 3, 256);
}

extern "C" int mpreid_resize_bilinear_u8(const uint8_t *src, const int64_t *offsets, const int32_t *hw, int batch,
                                         int max_in_h, int out_h, int out_w, uint8_t *dst, void *ws, size_t ws_bytes,
                                         mpreid_stream_t stream_) {
    ARG_CHECK(src && offsets && hw && dst && batch > 0 && max_in_h > 0 && out_h > 0 && out_w > 0);
    const size_t need = mpreid_resize_workspace_bytes(batch, max_in_h, out_w);
    if (!ws || ws_bytes < need) {
        mpreid_set_error("resize workspace too small: %zu < %zu", ws_bytes, need);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    unsigned char *tmp = (unsigned char *)ws;
    for (int b0 = 0; b0 < batch; b0 += 65535) { // gridDim.y limit
        const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
        const unsigned gx_h = (unsigned)(((int64_t)max_in_h * out_w + 255) / 256);
        const unsigned gx_v = (unsigned)(((int64_t)out_h * out_w + 255) / 256);
        hipLaunchKernelGGL(resize_h_kernel, dim3(gx_h, nb), dim3(256), 0, stream, src, offsets + b0, hw + 2 * b0, max_in_h,
                           out_w, tmp + (size_t)b0 * max_in_h * out_w * 3);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(resize_v_kernel, dim3(gx_v, nb), dim3(256), 0, stream,
                           tmp + (size_t)b0 * max_in_h * out_w * 3, hw + 2 * b0, max_in_h, out_h, out_w,
                           dst + (size_t)b0 * out_h * out_w * 3);
        LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int mpreid_tta_mean_f32(const float *feats, int n_views, int64_t rows, int dim, int normalize, float *out,
                                   mpreid_stream_t stream_) {
    ARG_CHECK(feats && out && n_views >= 1 && rows > 0 && dim > 0);
    hipLaunchKernelGGL(tta_mean_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, feats,
                       n_views, rows, dim, normalize, out);
    LAUNCH_CHECK();
    return 0;
}
