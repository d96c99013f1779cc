// tools/exp_alloc_kind.cpp — what is a "slow" input allocation, and can the allocation call control it?
// Build (on any box):  hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude -o tools/bin/exp_alloc_kind tools/exp_alloc_kind.cpp \
//                        -Lfastq-rs_amd -lfastq_hip -Wl,-rpath,'$ORIGIN/../../fastq-rs_amd'
// Run (GPU box):       tools/bin/exp_alloc_kind [methods] [instances] [gib]
//   methods: comma list of  malloc | contig | uncached | vmm2m | vmm1g | vmm_one | vmm_rec
//     malloc    hipMalloc
//     contig    hipExtMallocWithFlags(hipDeviceMallocContiguous)
//     vmm_one   hipMemAddressReserve + ONE hipMemCreate handle of the whole size + hipMemMap
//     vmm1g     ... handles of 1 GiB each, mapped back to back
//     vmm2m     ... handles of 2 MiB each (the minimum granularity's multiple), mapped back to back
//     vmm_rec   ... handles of the RECOMMENDED granularity each
// For every method: `instances` allocations of `gib` GiB, all alive together; each is filled with the synthetic file and
// measured: the bare read (fqh_read_ceiling, min of 4) and the scan's index kernel (fqh_scan + fqh_last_timing, min of 5,
// ONE context for everything, FQH_OPT_ADAPT_LINES off so that the line buffer is the same allocation throughout).
// With FQH_EXP_PMC=1 every input gets exactly 2 scans and 1 bare read (a rocprofv3 --pmc pass matches rows by order).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fastq_hip.h"

#define CK(x)                                                                               \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                        \
        }                                                                                   \
    } while (0)

struct Alloc {
    void *p = nullptr;
    size_t bytes = 0;
    bool vmm = false;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    size_t reserved = 0;
};

static bool vmm_alloc(Alloc &a, size_t bytes, size_t chunk, int dev) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    if (chunk == 1) chunk = grec;
    if (chunk == 0) chunk = (bytes + gmin - 1) / gmin * gmin;
    chunk = (chunk + gmin - 1) / gmin * gmin;
    const size_t total = (bytes + chunk - 1) / chunk * chunk;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) return false;
    a.p = va;
    a.reserved = total;
    a.vmm = true;
    a.bytes = bytes;
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) return false;
        a.handles.push_back(h);
        if (hipMemMap((char *)va + off, chunk, 0, h, 0) != hipSuccess) return false;
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) return false;
    static bool said = false;
    if (!said) {
        printf("# VMM granularity: minimum %zu, recommended %zu\n", gmin, grec);
        said = true;
    }
    return true;
}

static void release(Alloc &a) {
    if (!a.p) return;
    if (a.vmm) {
        (void)hipMemUnmap(a.p, a.reserved);
        for (auto h : a.handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(a.p, a.reserved);
    } else {
        (void)hipFree(a.p);
    }
    a = Alloc{};
}

static bool make(Alloc &a, const std::string &m, size_t bytes) {
    a = Alloc{};
    a.bytes = bytes;
    if (m == "malloc") return hipMalloc(&a.p, bytes) == hipSuccess;
    if (m == "contig") return hipExtMallocWithFlags(&a.p, bytes, hipDeviceMallocContiguous) == hipSuccess;
    if (m == "uncached") return hipExtMallocWithFlags(&a.p, bytes, hipDeviceMallocUncached) == hipSuccess;
    if (m == "vmm_one") return vmm_alloc(a, bytes, 0, 0);
    if (m == "vmm1g") return vmm_alloc(a, bytes, (size_t)1 << 30, 0);
    if (m == "vmm2m") return vmm_alloc(a, bytes, (size_t)2 << 20, 0);
    if (m == "vmm_rec") return vmm_alloc(a, bytes, 1, 0);
    fprintf(stderr, "unknown method %s\n", m.c_str());
    return false;
}

int main(int argc, char **argv) {
    std::string methods = argc > 1 ? argv[1] : "malloc,contig,vmm_one,vmm1g,vmm2m";
    const int inst = argc > 2 ? atoi(argv[2]) : 3;
    const double gib = argc > 3 ? atof(argv[3]) : 16.0;
    const bool pmc = getenv("FQH_EXP_PMC") != nullptr;
    const uint64_t n = (uint64_t)(gib * (1ull << 30)) / 330 * 330;
    CK(hipSetDevice(0));
    fqh_ctx *ctx = nullptr;
    if (fqh_create(0, &ctx) != FQH_OK) return 2;
    fqh_set_option(ctx, FQH_OPT_ADAPT_LINES, 0);
    fqh_set_option(ctx, FQH_OPT_SPIN_WAIT, 20000);
    const uint64_t cap = n / 300 + 16;
    uint64_t *rs = nullptr;
    CK(hipMalloc((void **)&rs, cap * sizeof(uint64_t)));
    size_t pos = 0;
    while (pos <= methods.size()) {
        size_t e = methods.find(',', pos);
        if (e == std::string::npos) e = methods.size();
        const std::string m = methods.substr(pos, e - pos);
        pos = e + 1;
        if (m.empty()) continue;
        std::vector<Alloc> as(inst);
        int made = 0;
        for (int i = 0; i < inst; ++i) {
            (void)hipGetLastError();
            if (!make(as[i], m, n + 4096)) {
                printf("%-8s instance %d: allocation failed (%s)\n", m.c_str(), i, hipGetErrorString(hipGetLastError()));
                release(as[i]);
                break;
            }
            ++made;
            if (fqh_synth_fill(ctx, (uint8_t *)as[i].p, 0, n, 0) != FQH_OK) {
                printf("%-8s instance %d: fill failed: %s\n", m.c_str(), i, fqh_last_error(ctx));
                return 2;
            }
        }
        for (int rnd = 0; rnd < (pmc ? 1 : 2); ++rnd)
            for (int i = 0; i < made; ++i) {
                const uint8_t *b = (const uint8_t *)as[i].p;
                float rd = 1e9f, best = 1e9f;
                for (int k = 0; k < (pmc ? 1 : 4); ++k) {
                    uint64_t sum = 0;
                    float ms = 0;
                    if (fqh_read_ceiling(ctx, b, n, &sum, &ms) != FQH_OK) return 2;
                    rd = std::min(rd, ms);
                }
                for (int k = 0; k < (pmc ? 2 : 7); ++k) {
                    fqh_summary s;
                    if ((fqh_scan(ctx, b, n, 1, nullptr, rs, cap, &s, nullptr) != FQH_OK || s.n_records != n / 330) && !getenv("FQH_EXP_IGNORE")) {
                        printf("scan failed: %s\n", fqh_last_error(ctx));
                        return 2;
                    }
                    fqh_timing t;
                    fqh_last_timing(ctx, &t);
                    if (pmc || k >= 2) best = std::min(best, t.index_ms);
                }
                // the same kernel WITHOUT its line stores, on the first 4 GiB of this input (FQH_OPT_PLACE_TRIES's yardstick: a fresh
                // context's first scan times it, fqh_placement reports it): is an input of the slow kind slow without the stores too?
                float nostore = 0, cand0 = 0;
                if (!pmc && rnd == 0) {
                    fqh_ctx *c2 = nullptr;
                    if (fqh_create(0, &c2) == FQH_OK) {
                        fqh_set_option(c2, FQH_OPT_ADAPT_LINES, 0);
                        fqh_set_option(c2, FQH_OPT_PLACE_TRIES, 2);
                        fqh_summary s;
                        if (fqh_scan(c2, b, n, 1, nullptr, rs, cap, &s, nullptr) == FQH_OK) {
                            int nc = 0;
                            float ms[10];
                            if (fqh_placement(c2, &nc, ms) == FQH_OK && nc) { nostore = ms[9]; cand0 = ms[0]; }
                        }
                        fqh_destroy(c2);
                    }
                }
                printf("%-8s round %d instance %d @ %p: bare read %.3f ms, index kernel %.3f ms (fast path %d); first 4 GiB without stores %.3f, with %.3f\n",
                       m.c_str(), rnd, i, as[i].p, rd, best, fqh_last_scan_fast(ctx), nostore, cand0);
                fflush(stdout);
            }
        for (auto &a : as) release(a);
    }
    fqh_destroy(ctx);
    return 0;
}
