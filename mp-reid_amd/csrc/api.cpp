// api.cpp — error reporting and device queries of libmpreid_hip.so
#include "common.h"

#include <cstring>
#include <string>
#include <utility>
#include <vector>

static thread_local char g_err[512] = "";

void mpreid_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// MPREID_TUNE="key=value,...": parsed once; see common.h / include/mpreid.h
namespace {
struct TuneTable {
    std::vector<std::pair<std::string, int>> kv;
    TuneTable() {
        static const char *known[] = {"gemm_big", "gemm_stagger", "gemm_stagger_all", "jaccard_wave", "jaccard_wave_rows",
                                      "jaccard_table", "csc_atomic", "rerank_overlap", "dist_sym_p2", "dist_sym_p2_naps", "dist_sym_p2_abl", "dist_sym_p2_grid", "dist_sym_p2_strip", "dist_p2_full", "gemm_walk", "gemm_grid", "gemm_ragged", "verbose"};
        const char *e = getenv("MPREID_TUNE");
        if (!e) return;
        std::string s(e);
        size_t pos = 0;
        while (pos <= s.size()) {
            size_t end = s.find(',', pos);
            if (end == std::string::npos) end = s.size();
            std::string item = s.substr(pos, end - pos);
            pos = end + 1;
            if (item.empty()) continue;
            const size_t eq = item.find('=');
            const std::string key = item.substr(0, eq);
            const int val = eq == std::string::npos ? 1 : atoi(item.c_str() + eq + 1);
            bool ok = false;
            for (const char *k : known) ok = ok || key == k;
            if (!ok) fprintf(stderr, "[mpreid] MPREID_TUNE: unknown key '%s' ignored\n", key.c_str());
            else kv.emplace_back(key, val);
        }
    }
};
} // namespace

int mpreid_tune(const char *key, int dflt) {
    static const TuneTable t;
    for (const auto &p : t.kv)
        if (p.first == key) return p.second;
    return dflt;
}

extern "C" int mpreid_version(void) { return 100; } /* 0.1.0 */

// 1 for a library compiled with -DMPREID_ABLATION (timing-ablation switches honoured: WRONG RESULTS by design), else 0
extern "C" int mpreid_is_ablation_build(void) {
#ifdef MPREID_ABLATION
    return 1;
#else
    return 0;
#endif
}

extern "C" const char *mpreid_last_error(void) { return g_err; }

extern "C" int mpreid_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int mpreid_device_info(char *name, int name_len, int *cu_count, size_t *hbm_bytes) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return MPREID_OK;
}

