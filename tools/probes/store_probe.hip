// How fast can one CU retire fp32 tile stores, and does the shape of a store instruction matter?
// 256 workgroups x 512 threads (160 KB LDS: one per CU) each write 24 tiles of 256 x 256 fp32 of a 20224-wide matrix from
// registers (no LDS pass, no arithmetic), in the persistent GEMM's tile order.  Patterns per store instruction (1 KB each):
//   0: 4 rows x 256 B (the GEMM epilogue's: wave = 128 x 64 sub-tile)     1: 1 row x 1 KB     2: 2 rows x 512 B
//   3: 16 rows x 64 B (accumulator layout, half lines)                      +8: without the nt hint
//   hipcc --offload-arch=gfx950 -O2 tools/probes/store_probe.hip -o tools/probes/store_probe && tools/probes/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool NT>
__global__ __launch_bounds__(512) void probe(float *out, int ld, int tiles_n, int ntiles, int xcd_mask) {
    extern __shared__ unsigned char smem[];
    if (!((xcd_mask >> (blockIdx.x & 7)) & 1)) return;   // only workgroups on the selected XCDs work
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) smem[0] = 1;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        float *base = out + (size_t)tm * 256 * ld + tn * 256;
        for (int s = 0; s < 32; ++s) {   // 32 store instructions per wave per tile
            float *p;
            if (PAT == 0) {
                const int wr = wave >> 2, wc = wave & 3;
                p = base + (size_t)(wr * 128 + s * 4 + (lane >> 4)) * ld + wc * 64 + (lane & 15) * 4;
            } else if (PAT == 1) {
                p = base + (size_t)(wave * 32 + s) * ld + lane * 4;
            } else if (PAT == 2) {
                const int wr = wave >> 1, wc = wave & 1;
                p = base + (size_t)(wr * 64 + s * 2 + (lane >> 5)) * ld + wc * 128 + (lane & 31) * 4;
            } else {
                const int wr = wave >> 2, wc = wave & 3;
                p = base + (size_t)(wr * 128 + (s >> 2) * 16 + (lane & 15)) * ld + wc * 64 + (s & 3) * 16 + (lane >> 4) * 4;
            }
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
            else *reinterpret_cast<f32x4 *>(p) = v;
            v[1] += 1.f;
        }
    }
}

// stores on `n_store` CUs of XCD 0 while the other CUs of that XCD stream L2-resident operands (what the k-loops of the
// neighbours do during a CU's epilogue): each reader wave keeps 8 x 1 KB loads in flight over a 3 MB buffer
__global__ __launch_bounds__(512) void probe_mixed(float *out, int ld, int tiles_n, int n_store, const f32x4 *rd, int rd_elems,
                                                   unsigned long long *t_store, float *sink, int reader_iters) {
    extern __shared__ unsigned char smem[];
    if (blockIdx.x & 7) return;   // XCD 0 only
    const int cu = blockIdx.x >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) smem[0] = 1;
    if (cu < n_store) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
        for (int tile = cu; tile < n_store * 24; tile += n_store) {
            const int tm = tile / tiles_n, tn = tile % tiles_n;
            float *base = out + (size_t)tm * 256 * ld + tn * 256;
            const int wr = wave >> 2, wc = wave & 3;
            for (int s = 0; s < 32; ++s) {
                float *p = base + (size_t)(wr * 128 + s * 4 + (lane >> 4)) * ld + wc * 64 + (lane & 15) * 4;
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
                v[1] += 1.f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) t_store[cu] = __builtin_amdgcn_s_memrealtime() - t0;
    } else {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int idx = (cu * 8 + wave) * 64 * 97 + lane;
        for (int it = 0; it < reader_iters; ++it) {
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = rd[idx % rd_elems];
                idx += 64 * 13;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += x[u];
        }
        if (acc[0] == 12345.678f) sink[threadIdx.x] = acc[1];
    }
}

template <int PAT, bool NT>
static void run(float *d, int ld, int tiles, const char *name, int grid = 256, int xcd_mask = 255) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<PAT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int ntiles = tiles * tiles;
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<PAT, NT>), dim3(grid), dim3(512), 160 * 1024, 0, d, ld, tiles, grid * 24, xcd_mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    (void)ntiles;
    int nx = 0;
    for (int x = 0; x < 8; ++x) nx += (xcd_mask >> x) & 1;
    const int active = grid * nx / 8;
    const double bytes = (double)active * 24 * 256 * 256 * 4;   // 24 tiles per working workgroup
    grid = active;
    printf("%-36s %3d CUs: %.3f ms  %5.0f GB/s  = %.1f GB/s per CU, %.2f us per 256 KB tile per CU\n", name, grid, best,
           bytes / best / 1e6, bytes / best / 1e6 / grid, best * 1e3 / 24.0);
}

int main() {
    const int tiles = 79, ld = tiles * 256;
    float *d;
    if (hipMalloc(&d, (size_t)ld * ld * 4) != hipSuccess) return 1;
    run<0, true>(d, ld, tiles, "4 rows x 256 B, nt (GEMM epilogue)");
    run<1, true>(d, ld, tiles, "1 row x 1 KB, nt");
    run<2, true>(d, ld, tiles, "2 rows x 512 B, nt");
    run<3, true>(d, ld, tiles, "16 rows x 64 B, nt");
    run<0, false>(d, ld, tiles, "4 rows x 256 B");
    run<1, false>(d, ld, tiles, "1 row x 1 KB");
    // fewer CUs storing at once (HBM no longer the limit): what ONE CU's store path sustains
    for (int grid : {8, 32, 64, 128})
        run<0, true>(d, ld, tiles, "4 rows x 256 B, nt", grid);
    // the same number of CUs, all on ONE XCD / on two / on four: the per-XCD write port
    run<0, true>(d, ld, tiles, "32 CUs of one XCD", 256, 0x01);
    run<0, true>(d, ld, tiles, "16 CUs of one XCD", 128, 0x01);
    run<0, true>(d, ld, tiles, "8 CUs of one XCD", 64, 0x01);
    run<0, true>(d, ld, tiles, "4 CUs of one XCD", 32, 0x01);
    run<0, true>(d, ld, tiles, "64 CUs of two XCDs", 256, 0x03);
    run<0, true>(d, ld, tiles, "128 CUs of four XCDs", 256, 0x0f);
    {
        f32x4 *rd;
        unsigned long long *ts;
        float *sink;
        const int rd_elems = 3 * 1024 * 1024 / 16;
        hipMalloc(&rd, 3 * 1024 * 1024);
        hipMemset(rd, 0, 3 * 1024 * 1024);
        hipMalloc(&ts, 32 * 8);
        hipMalloc(&sink, 4096);
        hipFuncSetAttribute(reinterpret_cast<const void *>(probe_mixed), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int n_store : {4, 8, 16}) {
            for (int iters : {0, 20000}) {
                unsigned long long h[32];
                for (int r = 0; r < 2; ++r) {
                    hipLaunchKernelGGL(probe_mixed, dim3(256), dim3(512), 160 * 1024, 0, d, ld, tiles, n_store, rd, rd_elems, ts, sink, iters);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h, ts, 32 * 8, hipMemcpyDeviceToHost);
                double m = 0;
                for (int i = 0; i < n_store; ++i) m += (double)h[i];
                m /= n_store;
                printf("%2d CUs of XCD 0 store, the other %2d %s: %.2f us per 256 KB tile per storing CU\n", n_store, 32 - n_store,
                       iters ? "stream 1 KB loads from a 3 MB L2-resident buffer" : "idle", m / 100.0 / 24.0);
            }
        }
    }
    run<1, true>(d, ld, tiles, "1 row x 1 KB, nt", 32);
    run<3, true>(d, ld, tiles, "16 rows x 64 B, nt", 32);
    return 0;
}
