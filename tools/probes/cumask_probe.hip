// Does hipExtStreamCreateWithCUMask partition the chip the way the two-partition encoder schedule needs (round 6)?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o tools/probes/cumask_probe && tools/probes/cumask_probe [G]
// Stream G gets the first G mask bits (default 224), stream X the remaining 256 - G.  Checks:
//  1. where the workgroups of a launch on each stream land (XCC_ID / HW_ID): disjoint CU sets, how many CUs per XCC;
//  2. a 160 KB-LDS persistent-style kernel with G workgroups on stream G is fully co-resident (one round), and the same
//     launch with 256 workgroups takes two rounds;
//  3. both streams busy at once: the X kernel's duration does not depend on whether G is busy and vice versa.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                              \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

__global__ __launch_bounds__(256) void where(unsigned *out, int spin_ticks) {
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2 + 0] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
}

static unsigned cu_key(unsigned hw, unsigned xcc) {
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    return ((xcc & 15) << 16) | (se << 8) | (sh << 4) | cu;
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 224;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device: %s, %d CUs; G = %d, X = %d\n", p.name, ncu, G, ncu - G);
    std::vector<uint32_t> mg((ncu + 31) / 32, 0), mx((ncu + 31) / 32, 0);
    for (int i = 0; i < ncu; ++i) (i < G ? mg : mx)[i / 32] |= 1u << (i % 32);
    hipStream_t sg, sx;
    CK(hipExtStreamCreateWithCUMask(&sg, (uint32_t)mg.size(), mg.data()));
    CK(hipExtStreamCreateWithCUMask(&sx, (uint32_t)mx.size(), mx.data()));
    unsigned *dg, *dx;
    const int nb = 2048;
    CK(hipMalloc(&dg, nb * 8));
    CK(hipMalloc(&dx, nb * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(where), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // 1. placement
    hipLaunchKernelGGL(where, dim3(nb), dim3(256), 1024, sg, dg, 200);
    hipLaunchKernelGGL(where, dim3(nb), dim3(256), 1024, sx, dx, 200);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> hg(nb * 2), hx(nb * 2);
    CK(hipMemcpy(hg.data(), dg, nb * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx.data(), dx, nb * 8, hipMemcpyDeviceToHost));
    std::set<unsigned> cg, cx;
    std::map<unsigned, int> per_xcc_g, per_xcc_x;
    for (int b = 0; b < nb; ++b) {
        cg.insert(cu_key(hg[b * 2], hg[b * 2 + 1]));
        cx.insert(cu_key(hx[b * 2], hx[b * 2 + 1]));
    }
    for (unsigned k : cg) per_xcc_g[k >> 16]++;
    for (unsigned k : cx) per_xcc_x[k >> 16]++;
    int overlap = 0;
    for (unsigned k : cx) overlap += cg.count(k);
    printf("stream G: %zu distinct CUs; stream X: %zu distinct CUs; common: %d\n", cg.size(), cx.size(), overlap);
    printf("CUs per XCC  G:");
    for (auto &kv : per_xcc_g) printf(" %u:%d", kv.first, kv.second);
    printf("   X:");
    for (auto &kv : per_xcc_x) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    // 2. whole-CU workgroups (160 KB of LDS, 256 threads): G of them on stream G = one round; 256 = two rounds
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    auto timed = [&](hipStream_t s, int blocks, size_t lds, int ticks, float *ms) -> int {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(where, dim3(blocks), dim3(256), lds, s, dg, ticks);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(ms, e0, e1));
        return 0;
    };
    float a = 0, b2 = 0, c = 0;
    if (timed(sg, G, 160 * 1024, 20000, &a)) return 1;          // 200 us spin per workgroup (100 MHz counter)
    if (timed(sg, G, 160 * 1024, 20000, &a)) return 1;
    if (timed(sg, ncu, 160 * 1024, 20000, &b2)) return 1;
    if (timed((hipStream_t)0, ncu, 160 * 1024, 20000, &c)) return 1;
    printf("whole-CU workgroups, 200 us each: %d on G: %.3f ms; %d on G: %.3f ms (two rounds expected); %d on the null stream: %.3f ms\n",
           G, a, ncu, b2, ncu, c);
    // 3. both partitions busy at once
    const int xb = (ncu - G) * 2;
    CK(hipEventRecord(e0, sg));
    hipLaunchKernelGGL(where, dim3(G), dim3(256), 160 * 1024, sg, dg, 100000);     // 1 ms on G
    CK(hipEventRecord(e1, sg));
    CK(hipEventRecord(f0, sx));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(where, dim3(xb), dim3(256), 64 * 1024, sx, dx, 20000);   // 4 x 200 us on X
    CK(hipEventRecord(f1, sx));
    CK(hipDeviceSynchronize());
    float tg = 0, tx = 0;
    CK(hipEventElapsedTime(&tg, e0, e1));
    CK(hipEventElapsedTime(&tx, f0, f1));
    printf("concurrent: G kernel (1 ms spin) %.3f ms; four X kernels (200 us spin each, %d workgroups) %.3f ms -> %s\n", tg, xb, tx,
           (tg < 1.3f && tx < 1.1f) ? "the partitions run side by side" : "NOT concurrent");
    return 0;
}
