// Are the packed fp32 instructions of gfx950 bit-identical to their scalar forms?  (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32
// against v_mul_f32, v_add_f32, v_fma_f32 on random operands, incl. small magnitudes)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/pk_probe.hip -o tools/probes/pk_probe && tools/probes/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, const float *y, const float *z, unsigned *bad, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = y[i], c = z[i];
    float m, s, f;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(a), "v"(c));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f) : "v"(a), "v"(b), "v"(c));
    f2 av = {a, a}, bv = {b, b}, cv = {c, c}, pm, ps, pf;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(av), "v"(bv));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(ps) : "v"(av), "v"(cv));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf) : "v"(av), "v"(bv), "v"(cv));
    if (__float_as_uint(m) != __float_as_uint(pm[0]) || __float_as_uint(m) != __float_as_uint(pm[1])) atomicAdd(&bad[0], 1u);
    if (__float_as_uint(s) != __float_as_uint(ps[0]) || __float_as_uint(s) != __float_as_uint(ps[1])) atomicAdd(&bad[1], 1u);
    if (__float_as_uint(f) != __float_as_uint(pf[0]) || __float_as_uint(f) != __float_as_uint(pf[1])) atomicAdd(&bad[2], 1u);
    // the QuickGELU chain both ways
    const float h = a * 4.0f, cst = -1.702f * 1.44269504088896340736f;
    const float e1 = __builtin_amdgcn_exp2f(cst * h), v1 = h * __builtin_amdgcn_rcpf(1.0f + e1);
    f2 hv = {h, h}, tv;
    f2 cc = {cst, cst};
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(tv) : "v"(hv), "v"(cc));
    const float e2 = __builtin_amdgcn_exp2f(tv[0]);
    f2 ev = {e2, e2}, one = {1.0f, 1.0f}, dv, vv;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(dv) : "v"(ev), "v"(one));
    const float r2 = __builtin_amdgcn_rcpf(dv[0]);
    f2 rv = {r2, r2};
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(vv) : "v"(hv), "v"(rv));
    if (__float_as_uint(v1) != __float_as_uint(vv[0])) atomicAdd(&bad[3], 1u);
}
int main() {
    const int n = 1 << 22;
    float *h[3];
    uint32_t seed = 777;
    for (int t = 0; t < 3; ++t) {
        h[t] = new float[n];
        for (int i = 0; i < n; ++i) {
            seed = seed * 1664525u + 1013904223u;
            uint32_t bits = (seed & 0x807fffffu) | ((127 - 12 + ((seed >> 23) % 16)) << 23);
            memcpy(&h[t][i], &bits, 4);
        }
    }
    float *d[3];
    unsigned *bad, hb[4];
    for (int t = 0; t < 3; ++t) {
        hipMalloc(&d[t], n * 4);
        hipMemcpy(d[t], h[t], n * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&bad, 16);
    hipMemset(bad, 0, 16);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d[0], d[1], d[2], bad, n);
    hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
    printf("of %d random operand triples: v_pk_mul_f32 != v_mul_f32: %u, v_pk_add_f32 != v_add_f32: %u, v_pk_fma_f32 != v_fma_f32: %u, "
           "QuickGELU chain (compiler scalar vs packed asm): %u\n", n, hb[0], hb[1], hb[2], hb[3]);
    return 0;
}
