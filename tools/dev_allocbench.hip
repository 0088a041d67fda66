// dev_allocbench — what a one-shot run pays for its consensus workspace: hipMalloc of 10^2 GB in one piece, in several pieces on several threads, and
// through the virtual-memory API (one address range, physical chunks created and mapped by several threads), and what a concurrent 1 GB upload sees of it.
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/allocbench tools/dev_allocbench.hip -lpthread ; run: /tmp/allocbench [GB]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void touch(char* p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }

static void upload_probe(const char* what, std::atomic<int>& stop) {   // 256 MB host -> device copies while something else allocates
    std::vector<char> h(256u << 20, 1);
    char* d = nullptr;
    CK(hipSetDevice(0));
    int n = 0; double worst = 0, t0 = now();
    while (!stop.load()) {
        const double a = now();
        CK(hipMalloc((void**)&d, h.size())); CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipFree(d));
        worst = std::max(worst, now() - a); n++;
    }
    printf("    [%s] upload probe: %d x (hipMalloc + 256 MB copy + hipFree) in %.3f s, worst %.3f s\n", what, n, now() - t0, worst);
}

int main(int argc, char** argv) {
    const size_t GB = 1ull << 30;
    const size_t total = (argc > 1 ? (size_t)atoi(argv[1]) : 208) * GB;
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    size_t fr, tt; CK(hipMemGetInfo(&fr, &tt));
    printf("free %.1f GB of %.1f GB; workspace %.0f GB\n", fr / 1e9, tt / 1e9, total / 1e9);
    for (int rep = 0; rep < 2; rep++) {   // ---- one piece
        std::atomic<int> stop{0};
        std::thread pr(upload_probe, "one hipMalloc", std::ref(stop));
        char* p = nullptr;
        double t0 = now(); CK(hipMalloc((void**)&p, total)); double t1 = now();
        stop.store(1); pr.join();
        touch<<<(unsigned)((total / (2u << 20) + 255) / 256), 256>>>(p, total, 2u << 20); CK(hipDeviceSynchronize()); double t2 = now();
        CK(hipFree(p)); double t3 = now();
        printf("one hipMalloc(%.0f GB): %.3f s, first touch of every 2 MB page %.3f s, hipFree %.3f s\n", total / 1e9, t1 - t0, t2 - t1, t3 - t2);
    }
    for (int nt : {4, 8, 16}) {   // ---- several pieces, several threads
        std::vector<char*> ps((size_t)nt, nullptr);
        std::vector<std::thread> th;
        std::atomic<int> stop{0};
        std::thread pr(upload_probe, "parallel hipMalloc", std::ref(stop));
        double t0 = now();
        for (int t = 0; t < nt; t++) th.emplace_back([&, t]() { CK(hipSetDevice(0)); CK(hipMalloc((void**)&ps[(size_t)t], total / nt)); });
        for (auto& x : th) x.join();
        double t1 = now();
        stop.store(1); pr.join();
        for (char* p : ps) CK(hipFree(p));
        printf("%d threads x hipMalloc(%.1f GB): %.3f s, frees %.3f s\n", nt, total / nt / 1e9, t1 - t0, now() - t1);
    }
    // ---- virtual memory API: one range, chunks created + mapped + made accessible by T threads
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("no VMM API\n"); return 0; }
    printf("VMM granularity %zu\n", gran);
    for (size_t chunk : {1 * GB, 4 * GB, 16 * GB})
        for (int nt : {1, 4, 8}) {
            void* va = nullptr;
            double t0 = now();
            if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("hipMemAddressReserve failed\n"); return 0; }
            const size_t nchunk = total / chunk;
            std::vector<hipMemGenericAllocationHandle_t> hs(nchunk);
            std::atomic<size_t> next{0};
            std::atomic<int> stop{0}, bad{0};
            std::thread pr(upload_probe, "VMM", std::ref(stop));
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++)
                th.emplace_back([&]() {
                    CK(hipSetDevice(0));
                    hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
                    for (;;) {
                        const size_t k = next.fetch_add(1);
                        if (k >= nchunk) break;
                        if (hipMemCreate(&hs[k], chunk, &prop, 0) != hipSuccess || hipMemMap((char*)va + k * chunk, chunk, 0, hs[k], 0) != hipSuccess ||
                            hipMemSetAccess((char*)va + k * chunk, chunk, &ad, 1) != hipSuccess) { bad.store(1); break; }
                    }
                });
            for (auto& x : th) x.join();
            double t1 = now();
            stop.store(1); pr.join();
            if (bad.load()) { printf("VMM chunk %zu GB x %d threads: failed (%s)\n", chunk / GB, nt, hipGetErrorString(hipGetLastError())); }
            else {
                touch<<<(unsigned)((total / (2u << 20) + 255) / 256), 256>>>((char*)va, total, 2u << 20);
                hipError_t e = hipDeviceSynchronize();
                printf("VMM %zu chunks of %zu GB on %d threads: %.3f s (touch: %s)\n", nchunk, chunk / GB, nt, t1 - t0, hipGetErrorString(e));
            }
            double t2 = now();
            for (size_t k = 0; k < nchunk; k++) { (void)hipMemUnmap((char*)va + k * chunk, chunk); (void)hipMemRelease(hs[k]); }
            (void)hipMemAddressFree(va, total);
            printf("    unmap + release %.3f s\n", now() - t2);
        }
    return 0;
}
