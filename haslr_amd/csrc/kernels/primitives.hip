// primitives.hip — device-wide exclusive scan and a stable LSD radix sort, hand-written for wave64.
// (rocPRIM/hipCUB are deliberately not used on the product path.)
#include <algorithm>

#include "kernels.h"

namespace hxk {

namespace {
constexpr int SCAN_T = 256, SCAN_I = 4, SCAN_TILE = SCAN_T * SCAN_I;

__device__ __forceinline__ uint64_t block_scan_excl_u64(uint64_t v, uint64_t* total, uint64_t* lds /* SCAN_T/64 */) {
    // inclusive wave scan
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(x, d, 64);
        if (lane >= d) x += o;
    }
    if (lane == 63) lds[w] = x;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (int i = 0; i < SCAN_T / 64; i++) { if (i < w) base += lds[i]; tot += lds[i]; }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

template <typename TIN>
__global__ void scan_block_sums(const TIN* in, uint64_t n, uint64_t* sums) {
    __shared__ uint64_t lds[SCAN_T / 64];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE, acc = 0;
    for (int i = 0; i < SCAN_I; i++) {
        uint64_t k = base + (uint64_t)i * SCAN_T + threadIdx.x;
        if (k < n) acc += in[k];
    }
    uint64_t tot;
    block_scan_excl_u64(acc, &tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

template <typename TIN>
__global__ void scan_apply(const TIN* in, uint64_t n, const uint64_t* block_prefix, uint64_t* out) {
    __shared__ uint64_t lds[SCAN_T / 64];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    // blocked arrangement: thread t owns items [t*SCAN_I, (t+1)*SCAN_I)
    uint64_t v[SCAN_I], acc = 0;
    for (int i = 0; i < SCAN_I; i++) {
        uint64_t k = base + (uint64_t)threadIdx.x * SCAN_I + i;
        v[i] = k < n ? (uint64_t)in[k] : 0;
        acc += v[i];
    }
    uint64_t tot;
    uint64_t ex = block_scan_excl_u64(acc, &tot, lds) + (block_prefix ? block_prefix[blockIdx.x] : 0);
    for (int i = 0; i < SCAN_I; i++) {
        uint64_t k = base + (uint64_t)threadIdx.x * SCAN_I + i;
        if (k < n) out[k] = ex;
        ex += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_T - 1) out[n] = ex;   // grand total
}

template <typename TIN>
void scan_impl(const TIN* in, uint64_t* out, uint64_t n, hipStream_t s, Workspace& ws) {
    if (n == 0) { hipMemsetAsync(out, 0, sizeof(uint64_t), s); return; }
    uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb == 1) {
        scan_apply<TIN><<<1, SCAN_T, 0, s>>>(in, n, nullptr, out);
        return;
    }
    uint64_t* sums = (uint64_t*)ws.take(nb * sizeof(uint64_t));
    uint64_t* prefix = (uint64_t*)ws.take((nb + 1) * sizeof(uint64_t));
    if (!sums || !prefix) return;   // (out of device memory: Workspace::oom is set, the operator reports it before it reads any result)
    scan_block_sums<TIN><<<(unsigned)nb, SCAN_T, 0, s>>>(in, n, sums);
    scan_impl<uint64_t>(sums, prefix, nb, s, ws);
    scan_apply<TIN><<<(unsigned)nb, SCAN_T, 0, s>>>(in, n, prefix, out);
}
}  // namespace

void exclusive_scan_u32(const uint32_t* in, uint64_t* out, uint64_t n, hipStream_t s, Workspace& ws) { scan_impl<uint32_t>(in, out, n, s, ws); }

void* Workspace::take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (!blocks.empty() && blocks.back().cap - blocks.back().used >= bytes) {
        void* r = blocks.back().p + blocks.back().used;
        blocks.back().used += bytes;
        return r;
    }
    size_t cap = std::max<size_t>(bytes, blocks.empty() ? (size_t)1 << 20 : blocks.back().cap * 2);
    char* p = nullptr;
    if (hipMalloc((void**)&p, cap) != hipSuccess) { (void)hipGetLastError(); oom = true; return nullptr; }
    blocks.push_back(Block{p, cap, bytes});
    return p;
}
void Workspace::reset(hipStream_t s) {
    if (blocks.size() > 1) {   // grew during the last use: one block of the total size from now on
        size_t total = 0;
        for (const Block& b : blocks) total += b.cap;
        (void)hipStreamSynchronize(s);
        release();
        char* p = nullptr;
        if (hipMalloc((void**)&p, total) == hipSuccess) blocks.push_back(Block{p, total, 0}); else { (void)hipGetLastError(); oom = true; }
    }
    for (Block& b : blocks) b.used = 0;
}
void Workspace::release() {
    for (Block& b : blocks) (void)hipFree(b.p);
    blocks.clear();
}

// ------------------------------------------------------------------------------------------------
// Stable LSD radix sort, 8 bits per pass. Each block owns a tile of RS_TILE consecutive elements.
//   pass 1  histogram per tile            -> hist[digit * nblocks + block]
//   scan    exclusive scan (digit-major)   -> global base of every (digit, block)
//   pass 2  stable scatter: within a tile elements are visited in rounds of RS_T consecutive elements;
//           inside a round a wave ranks equal digits with ballots (match-any), waves are ordered through
//           per-wave digit counts in LDS, rounds through a running per-digit counter.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int RS_T = 256, RS_ROUNDS = 8, RS_TILE = RS_T * RS_ROUNDS, RS_W = RS_T / 64;

__device__ __forceinline__ uint32_t digit_of(uint64_t k, int shift) { return (uint32_t)(k >> shift) & 255u; }

__global__ void rs_hist(const uint64_t* key, uint64_t n, int shift, uint32_t* hist, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ROUNDS; r++) {
        uint64_t i = base + (uint64_t)r * RS_T + threadIdx.x;
        if (i < n) atomicAdd(&h[digit_of(key[i], shift)], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ void rs_scatter(const uint64_t* key, const uint32_t* val, uint64_t n, int shift, const uint64_t* base_off,
                           uint32_t nblocks, uint64_t* key_out, uint32_t* val_out) {
    __shared__ uint32_t cnt[RS_W][256];
    __shared__ uint32_t run[256];
    run[threadIdx.x] = 0;
    for (int w = 0; w < RS_W; w++) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint64_t tile = (uint64_t)blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ROUNDS; r++) {
        uint64_t i = tile + (uint64_t)r * RS_T + threadIdx.x;
        bool valid = i < n;
        uint64_t k = valid ? key[i] : 0;
        uint32_t v = valid ? val[i] : 0;
        uint32_t d = digit_of(k, shift);
        // lanes of this wave holding the same digit
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            uint64_t bb = __ballot((d >> b) & 1);
            m &= ((d >> b) & 1) ? bb : ~bb;
        }
        uint32_t rank_in_wave = __popcll(m & lt);
        if (valid && rank_in_wave == 0) cnt[w][d] = __popcll(m);
        __syncthreads();
        if (valid) {
            uint32_t before = run[d];
            for (int ww = 0; ww < w; ww++) before += cnt[ww][d];
            uint64_t dst = base_off[(uint64_t)d * nblocks + blockIdx.x] + before + rank_in_wave;
            key_out[dst] = k;
            val_out[dst] = v;
        }
        __syncthreads();
        uint32_t add = 0;
        for (int ww = 0; ww < RS_W; ww++) { add += cnt[ww][threadIdx.x]; cnt[ww][threadIdx.x] = 0; }
        run[threadIdx.x] += add;
        __syncthreads();
    }
}
}  // namespace

void radix_sort_pairs(uint64_t* key, uint32_t* val, uint64_t* key_tmp, uint32_t* val_tmp, uint64_t n, int bits_lo, int bits_hi, hipStream_t s, Workspace& ws) {
    if (n == 0) return;
    uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    uint32_t* hist = (uint32_t*)ws.take((uint64_t)256 * nblocks * sizeof(uint32_t));
    uint64_t* base = (uint64_t*)ws.take(((uint64_t)256 * nblocks + 1) * sizeof(uint64_t));
    if (!hist || !base) return;
    uint64_t *ki = key, *ko = key_tmp;
    uint32_t *vi = val, *vo = val_tmp;
    int shifts[16], ns = 0;
    for (int b = 0; b < bits_lo; b += 8) shifts[ns++] = b;
    for (int b = 0; b < bits_hi; b += 8) shifts[ns++] = 32 + b;
    for (int p = 0; p < ns; p++) {
        rs_hist<<<nblocks, RS_T, 0, s>>>(ki, n, shifts[p], hist, nblocks);
        exclusive_scan_u32(hist, base, (uint64_t)256 * nblocks, s, ws);
        rs_scatter<<<nblocks, RS_T, 0, s>>>(ki, vi, n, shifts[p], base, nblocks, ko, vo);
        uint64_t* tk = ki; ki = ko; ko = tk;
        uint32_t* tv = vi; vi = vo; vo = tv;
    }
    if (ki != key) {   // odd number of passes: result sits in the temp buffers
        hipMemcpyAsync(key, ki, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
        hipMemcpyAsync(val, vi, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
    }
}

// Consensus collection (hx_api.hip collect_batch): the kernels leave every edge's string at its own offset of a buffer sized by the node ESTIMATES (a
// hundred megabytes for 13 000 edges); the finished strings are moved side by side before they cross PCIe. One workgroup per string;
// desc = (source offset lo / hi, destination offset lo / hi, length).
__global__ void __launch_bounds__(256) k_gather_bytes(const char* __restrict__ src, const uint32_t* __restrict__ desc, char* __restrict__ dst) {
    const uint32_t* d = desc + (size_t)blockIdx.x * 5;
    const uint64_t so = (uint64_t)d[0] | (uint64_t)d[1] << 32, to = (uint64_t)d[2] | (uint64_t)d[3] << 32;
    const uint32_t n = d[4];
    for (uint32_t i = threadIdx.x; i < n; i += 256) dst[to + i] = src[so + i];
}
void gather_bytes(const char* src, const uint32_t* desc, uint32_t n_items, char* dst, hipStream_t s) {
    if (n_items) k_gather_bytes<<<n_items, 256, 0, s>>>(src, desc, dst);
}

// One wave with as much private memory per lane as the largest k_poa instance spills (kernels/poa.hip: 288 to 928 bytes per lane): run once on a
// stream, it takes the stream's hardware queue to that scratch size. See hx_api.hip (poa_scratch_warm) for why.
__global__ void __launch_bounds__(64) k_scratch_warm(uint32_t* out, uint32_t n) {
    volatile uint32_t a[256];
    for (uint32_t i = 0; i < 256; i++) a[i] = i * n;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < n; i++) sum += a[(i * 7u) & 255u];
    if (out) *out = sum;
}
void scratch_warm(hipStream_t s) { k_scratch_warm<<<1, 64, 0, s>>>(nullptr, 3u); }

}  // namespace hxk
