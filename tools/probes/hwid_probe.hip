// Where do the workgroups of a 2-per-CU launch land?  512 workgroups x 256 threads x 72 KB LDS (the twin GEMM's shape):
// every workgroup records HW_ID / XCC_ID of its first wave and spins ~50 us so that all are co-resident.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(unsigned *out) {
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < 5000) __builtin_amdgcn_s_sleep(32);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 0] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}

int main() {
    const int nb = 512;
    unsigned *d;
    hipMalloc(&d, nb * 4 * 2 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 72 * 1024, 0, d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4 * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 15;
        if (b < 24 || b % 64 == 0) {
            printf("block %3d: xcc %u se %u sh %u cu %2u tg %u | waves:", b, xcc, se, sh, cu, tg);
            for (int w = 0; w < 4; ++w) printf(" simd %u slot %u", (h[(b * 4 + w) * 2] >> 4) & 3, h[(b * 4 + w) * 2] & 15);
            printf("\n");
        }
        by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    int hist[8] = {0};
    for (auto &kv : by_cu) hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
    printf("distinct (xcc, se, sh, cu): %zu; workgroups per CU histogram:", by_cu.size());
    for (int i = 0; i < 8; ++i) printf(" %d:%d", i, hist[i]);
    printf("\n");
    int shown = 0;
    for (auto &kv : by_cu) {
        if (shown++ >= 12) break;
        printf("cu key %06x:", kv.first);
        for (int b : kv.second) printf(" block %d (tg %u)", b, (h[b * 8] >> 16) & 15);
        printf("\n");
    }
    return 0;
}
