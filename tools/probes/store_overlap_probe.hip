// Can a wave's tile stores drain BEHIND its next k-loop if the wave never waits on vmcnt?  (round 6)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/store_overlap_probe.hip -o tools/probes/store_overlap_probe && tools/probes/store_overlap_probe
// The persistent GEMM's tile = a k-loop (matrix instructions only, stand-in below) + an epilogue that stores 256 KB per CU
// (32 x 1 KB store instructions per wave, the epilogue's shape: 4 rows x 256 B, nt).  In the product kernel every wave also
// issues LDS-DMA and waits for it with a counted vmcnt -- which retires in order, so each wave's stores must drain before its
// next tile's first DMA wait.  If only PRODUCER waves issued DMA (and waited), the other waves would never execute a vmcnt wait
// in the k-loop: would their stores then leave in the shadow of the next k-loop?  Variants, 24 tiles per CU, all CUs:
//   0  k-loop only (no stores)                       1  stores, every wave drains (vmcnt(0)) before the next k-loop
//   2  stores, NO wave waits                         3  stores, waves 0-1 drain, waves 2-7 do not (the producer scheme)
// Per variant: kernel time, and the mean time a wave spends ISSUING its 32 stores (no wait included).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(512) void probe(float *out, int ld, int tiles_n, int mfma_iters, unsigned long long *issue_ticks) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) smem[0] = 1;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)((lane + i) & 7);
        b[i] = (_Float16)((lane * 3 + i) & 7);
    }
    unsigned long long ticks = 0;
    for (int t = 0; t < 24; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        // "k-loop": matrix instructions only
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        __syncthreads();
        if (VAR != 0) {
            float *base = out + (size_t)tm * 256 * ld + tn * 256;
            const int wr = wave >> 2, wc = wave & 3;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 8
            for (int s = 0; s < 32; ++s) {
                float *p = base + (size_t)(wr * 128 + s * 4 + (lane >> 4)) * ld + wc * 64 + (lane & 15) * 4;
                f32x4 v = acc[s & 7];
                v[0] += (float)s;
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
            }
            ticks += __builtin_amdgcn_s_memrealtime() - t0;
            if (VAR == 1 || (VAR == 3 && wave < 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
    }
    if (lane == 0) issue_ticks[blockIdx.x * 8 + wave] = ticks;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.678f) out[0] = s;
}

template <int VAR>
static int run(const char *name, float *d, int ld, int tiles_n, int cus, int iters, unsigned long long *dt) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<VAR>, dim3(cus), dim3(512), 160 * 1024, 0, d, ld, tiles_n, iters, dt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    static unsigned long long h[256 * 8];
    hipMemcpy(h, dt, sizeof(unsigned long long) * cus * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < cus * 8; ++i) mean += (double)h[i];
    mean = mean / (cus * 8) / 24.0 / 100.0;   // 100 MHz counter -> us per tile
    printf("%-58s %8.3f ms = %6.2f us per tile; a wave issues its 32 stores in %5.2f us\n", name, best, best * 1e3 / 24.0, mean);
    return 0;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;   // 400 x 8 MFMA per wave ~ 51 us at 2 waves per SIMD
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int tiles_n = 79, ld = tiles_n * 256;
    float *d;
    unsigned long long *dt;
    if (hipMalloc(&d, (size_t)ld * ld * 4) != hipSuccess || hipMalloc(&dt, sizeof(unsigned long long) * 256 * 8) != hipSuccess) return 1;
    printf("%d CUs, 24 tiles per CU, %d x 8 matrix instructions per wave and tile, 256 KB stored per CU and tile\n", cus, iters);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("0 k-loop only", d, ld, tiles_n, cus, iters, dt);
        run<1>("1 stores, every wave drains before the next k-loop", d, ld, tiles_n, cus, iters, dt);
        run<2>("2 stores, no wave waits", d, ld, tiles_n, cus, iters, dt);
        run<3>("3 stores, waves 0-1 drain, waves 2-7 do not", d, ld, tiles_n, cus, iters, dt);
    }
    return 0;
}
