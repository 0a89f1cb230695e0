// Is the fp16 pair split through v_cvt_pk_f16_f32 + v_fma_mixlo_f16 / v_fma_mixhi_f16 (lo = fp16(fma(hi, -1, x)): ONE rounding of
// the exact difference) bit-identical to the plain form hi = fp16(x), lo = fp16(x - float(hi)) (x - hi is exact in fp32, then
// one rounding)?  Random operands over the whole range the encoder sees: normal, tiny (lo subnormal or zero), large, negative.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probes/mix_probe.hip -o tools/probes/mix_probe && tools/probes/mix_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_mix(float x0, float x1, unsigned &hi, unsigned &lo) {
    const h2_t h = __builtin_convertvector(f2_t{x0, x1}, h2_t);
    hi = __builtin_bit_cast(unsigned, h);
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(x1));
    lo = d;
}
__device__ __forceinline__ unsigned short bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }
__global__ void k(const float *x, unsigned *bad, unsigned *sub, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const float a = x[i], b = x[i + 1];
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    unsigned hi, lo;
    split_pair_mix(a, b, hi, lo);
    if ((hi & 0xffffu) != bits(ha) || (hi >> 16) != bits(hb)) atomicAdd(&bad[0], 1u);
    if ((lo & 0xffffu) != bits(la) || (lo >> 16) != bits(lb)) atomicAdd(&bad[1], 1u);
    if (((bits(la) & 0x7c00u) == 0 && (bits(la) & 0x3ffu)) || ((bits(lb) & 0x7c00u) == 0 && (bits(lb) & 0x3ffu))) atomicAdd(sub, 1u);
}
int main() {
    const int n = 1 << 26;
    float *h = new float[n];
    uint32_t seed = 4242;
    for (int i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        // exponents 2^-30 .. 2^17 (beyond fp16's range on both sides), random sign and mantissa
        const uint32_t e = 127 - 30 + ((seed >> 23) % 48);
        const uint32_t bitsv = (seed & 0x807fffffu) | (e << 23);
        memcpy(&h[i], &bitsv, 4);
    }
    float *d;
    unsigned *bad, hb[3];
    hipMalloc(&d, (size_t)n * 4);
    hipMemcpy(d, h, (size_t)n * 4, hipMemcpyHostToDevice);
    hipMalloc(&bad, 12);
    hipMemset(bad, 0, 12);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, d, bad, bad + 2, n);
    hipMemcpy(hb, bad, 12, hipMemcpyDeviceToHost);
    printf("%d operand pairs: hi mismatches %u, lo mismatches %u (pairs with a subnormal lo: %u)\n", n / 2, hb[0], hb[1], hb[2]);
    return hb[0] || hb[1];
}
