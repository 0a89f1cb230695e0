// v_cvt_f16_f32 against v_cvt_pk_f16_f32 (gfx950) on exact ties and their neighbours: do both round to nearest even?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cvt_probe.hip -o tools/probes/cvt_probe && tools/probes/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, unsigned short *s, unsigned short *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    _Float16 a;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(v));
    unsigned pk;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(v));
    unsigned short au;
    memcpy(&au, &a, 2);
    s[i] = au;
    p[i] = (unsigned short)(pk & 0xffffu);
}
int main() {
    const int n = 1 << 20;
    float *hx = new float[n];
    uint32_t seed = 12345;
    for (int i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        // random sign / exponent in [-20, 3] / 10 mantissa bits, then a tail of: exactly half | half +- 1 ulp32 | random
        uint32_t sign = (seed >> 31) << 31, ex = 127 - 20 + ((seed >> 8) % 24), m10 = (seed >> 13) & 0x3ffu;
        uint32_t tail;
        switch (i & 3) {
        case 0: tail = 0x1000u; break;
        case 1: tail = 0x1001u; break;
        case 2: tail = 0x0fffu; break;
        default: tail = (seed >> 3) & 0x1fffu; break;
        }
        uint32_t bits = sign | (ex << 23) | (m10 << 13) | tail;
        memcpy(&hx[i], &bits, 4);
    }
    float *dx;
    unsigned short *ds, *dp;
    hipMalloc(&dx, n * 4);
    hipMalloc(&ds, n * 2);
    hipMalloc(&dp, n * 2);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dp, n);
    unsigned short *hs = new unsigned short[n], *hp = new unsigned short[n];
    hipMemcpy(hs, ds, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hp, dp, n * 2, hipMemcpyDeviceToHost);
    long diff[4] = {0, 0, 0, 0}, s_even = 0, p_even = 0, ties = 0, shown = 0;
    for (int i = 0; i < n; ++i) {
        if (hs[i] != hp[i]) {
            ++diff[i & 3];
            if (shown++ < 6) {
                uint32_t b;
                memcpy(&b, &hx[i], 4);
                printf("x = %.9g (bits %08x, class %d): v_cvt_f16_f32 -> %04x, v_cvt_pk_f16_f32 -> %04x\n", hx[i], b, i & 3, hs[i], hp[i]);
            }
        }
        if ((i & 3) == 0) {
            ++ties;
            s_even += !(hs[i] & 1);
            p_even += !(hp[i] & 1);
        }
    }
    printf("differences: exact ties %ld, tie+1ulp %ld, tie-1ulp %ld, random %ld (of %d each)\n", diff[0], diff[1], diff[2], diff[3], n / 4);
    printf("exact ties rounded to an even fp16 mantissa: v_cvt_f16_f32 %ld / %ld, v_cvt_pk_f16_f32 %ld / %ld\n", s_even, ties, p_even, ties);
    return 0;
}
