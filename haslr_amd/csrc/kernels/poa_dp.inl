// poa_dp.inl - part of kernels/poa.hip: the DP over (rank, column) - block / wave scans, row stores (nibbles through a buffer resource), the keys' constants, the mailboxes of the
// wave pipeline, and dp_rows (the row loop, its pruned form and the fast-row switch).
__device__ __forceinline__ int block_excl_scan_max(int v, int* lds /* blockDim/64 */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_scan_max(v);
    if (lane == 63) lds[w] = inc;
    barrier_lds_only();
    // wave totals (<= 16): one LDS read per lane, a 16-lane DPP row scan, and a scalar read of entry w-1
    const int nw = blockDim.x >> 6;
    int tot = (lane & 15) < nw ? lds[lane & 15] : NEG;
    int x = tot;
    x = max(x, __builtin_amdgcn_update_dpp(NEG, tot, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(NEG, tot, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(NEG, tot, 0x113, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(NEG, x, 0x114, 0xf, 0xe, false));
    x = max(x, __builtin_amdgcn_update_dpp(NEG, x, 0x118, 0xf, 0xc, false));
    const int base = w == 0 ? NEG : __builtin_amdgcn_readlane(x, w - 1);
    return max(base, wave_shift_up1(inc, NEG));
}

// a pointer every lane holds the same value of, as a scalar
template <class T> __device__ __forceinline__ T* uptr(T* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<T*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}
// inclusive prefix sum over the 64 lanes of a wave: the DPP sequence of wave_scan_max with an addition (no LDS round trips)
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);   // row_shr:3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, false);   // row_shr:4 bank_mask:0xe
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, false);   // row_shr:8 bank_mask:0xc
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 row_mask:0xa
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 row_mask:0xc
    return x;
}
// Exclusive prefix sum over the workgroup. Its barriers wait for LDS only: the loops of the graph phases call it once per block of ranks / bases, between
// their stores - with __syncthreads (which drains the wave's outstanding global stores first, ~2 us under load) the scans WERE those loops' time.
// A one-wave workgroup meets no barrier at all. Nothing here orders global memory: callers that hand data to other lanes through it synchronise themselves.
__device__ __forceinline__ uint32_t block_excl_scan_add(uint32_t v, uint32_t* lds /* blockDim/64 */, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t inc = wave_incl_add(v);
    if (nw == 1) { *total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63); return inc - v; }
    if (lane == 63) lds[w] = inc;
    barrier_lds_only();
    uint32_t base = 0, tot = 0;
    for (uint32_t i = 0; i < nw; i++) { const uint32_t x = lds[i]; if (i < w) base += x; tot += x; }
    barrier_lds_only();
    *total = tot;
    return base + inc - v;
}



// contiguous per-lane chunk stores/loads as single wide memory instructions (rows are padded to 16 columns, chunks are CM-aligned)
template <int CM> __device__ __forceinline__ void store_chunk_i32(int32_t* p, const int (&v)[CM]) {
    if constexpr (CM >= 4) {
#pragma unroll
        for (int q = 0; q < CM / 4; q++) reinterpret_cast<int4*>(p)[q] = make_int4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else if constexpr (CM == 2) *reinterpret_cast<int2*>(p) = make_int2(v[0], v[1]);
    else p[0] = v[0];
}
template <int CM> __device__ __forceinline__ void load_chunk_i32(const int32_t* p, int (&v)[CM]) {
    if constexpr (CM >= 4) {
#pragma unroll
        for (int q = 0; q < CM / 4; q++) { const int4 x = reinterpret_cast<const int4*>(p)[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
    } else if constexpr (CM == 2) { const int2 x = *reinterpret_cast<const int2*>(p); v[0] = x.x; v[1] = x.y; }
    else v[0] = p[0];
}
template <int CM> __device__ __forceinline__ void store_chunk_u8(uint8_t* p, const uint32_t (&v)[CM]) {
    if constexpr (CM >= 4) {
        uint32_t w[CM / 4];
#pragma unroll
        for (int q = 0; q < CM / 4; q++) w[q] = (v[4 * q] & 0xffu) | ((v[4 * q + 1] & 0xffu) << 8) | ((v[4 * q + 2] & 0xffu) << 16) | (v[4 * q + 3] << 24);
        if constexpr (CM == 4) *reinterpret_cast<uint32_t*>(p) = w[0];
        else if constexpr (CM == 8) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
        else {
#pragma unroll
            for (int q = 0; q < CM / 16; q++) reinterpret_cast<uint4*>(p)[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
        }
    } else if constexpr (CM == 2) *reinterpret_cast<uint16_t*>(p) = (uint16_t)((v[0] & 0xffu) | (v[1] << 8));
    else p[0] = (uint8_t)v[0];
}

// ---------------------------------------------------------------------------------------------------
// DP over (rank, column) for one sequence against the current graph — rows live in registers, the waves of an edge form a pipeline.
//
// Lane t of wave w owns the CM contiguous columns [(64 w + t) CM, (64 w + t + 1) CM) (w counts through all workgroups that share
// the edge) and keeps the CURRENT row there. The common predecessor of row i is row i-1: the lane's own registers, plus the value left
// of its first column, which falls out of the prefix scan that row i-1 needed anyway. So the usual row costs no LDS row traffic and
// exactly ONE cross-lane operation: the 64-lane DPP prefix-max scan of (chunk end - column*gap) that resolves the horizontal
// recurrence H[j] = max(T[j], H[j-1]+g) inside the wave.
//
// What crosses a wave boundary is one number per row: the prefix maximum through the wave's last column. Round 1 exchanged it with a
// workgroup barrier per row (all waves in lock step: 65 % of the wave cycles were spent parked). Now every wave runs at its own pace
// and is a stage of a pipeline: it publishes the carry of every finished row to a mailbox — a small tagged ring in LDS towards the
// next wave of the workgroup, a tagged word per row in HBM towards the first wave of the next workgroup ("member") — and takes its
// own carries from the wave on its left, 32 rows at a time (one coalesced read, lane r = row r of the batch, broadcast per row with
// v_readlane like the row records). A wave therefore runs one batch behind its left neighbour, polls once per 32 rows and never meets
// a barrier inside the DP. Tags (a row counter that runs through all DPs of the edge) make every entry self-validating; the consumer
// reports how far it has read so that the producer never laps it.
// Nothing else is shared: the LDS ring of kept rows and the rows kept in HBM are private to the wave (each with a copy of the value
// left of its first column), so a wave that is ahead can never pull a row from under one that is behind.
//
// Cells are "keys": 64 x score + 6 low bits = move type * 16 + 15 - predecessor slot. The low bits make one max() do the
// reference's tie-breaking: type 3 diagonal > 2 vertical > 1 horizontal (its traceback tries them in this order and takes a
// horizontal move only when nothing else reaches the score), and among moves of one type the first predecessor in in-edge order
// wins. Scores stay below 2^24 in magnitude (8*(V+L) with V+L < 2^21, checked by the host), so keys fit 32 bits. The 6 bits of the
// winning move ARE the direction byte written to HBM for the traceback (in-degrees above 16 send the edge back to the host, which
// retries it with the score-matrix traceback).
// Round 4: the score in a key is DE-RAMPED, X[i][j] = H[i][j] - gap * j. A horizontal move then keeps the key's score, so the row's
// horizontal recurrence is a plain prefix maximum: the scan input is the chunk's largest key, nothing is subtracted before the scan or
// added after it, and the chunk's own recurrence runs once, after the carry is known - each finished key (score x 64 + KHC, the one
// horizontal code of every row format) is the next column's horizontal candidate and the row as later rows read it, in one register.
// A diagonal move adds (substitution score - gap), a vertical one gap; only the sink scores and the score-matrix flavour's HBM rows
// (plain scores for its traceback) put the ramp back.
//
// Rows that a later row needs as a NON-adjacent predecessor are flagged by the CSR build ("kept") and copied to an LDS
// ring in the order they are produced (per wave: CM planes of 65 words, column t*CM+k at word 65*CM*w + 65*k + 1 + t, conflict-free; word 0
// of the wave's LAST plane holds the value left of the wave's first column, so "the column left of my chunk" is one load
// at lane offset 0 for EVERY lane), or to HBM when the ring has wrapped; predecessor references carry that location (0 registers,
// 1..14 ring slot+1, 15 HBM). Row metadata travels in registers: every wave loads the records of 64 rows with one coalesced load (one
// batch ahead) and broadcasts the current row's words with v_readlane.
// Columns beyond L are computed like real ones and never read by a real column, so the loop has no column predicates.
// ---------------------------------------------------------------------------------------------------
constexpr int32_t NEGK = -(1 << 30);   // "minus infinity" key
constexpr int KD = 63, KV = 47;           // low 6 bits of a key of a wide row = move type * 16 + 15 - predecessor slot: diagonal 3, vertical 2
constexpr int KHC = 4;                    // ... and the horizontal move's code in EVERY row format (4-bit rows: type 1 * 4 + 3 - 3; wide rows: below every other code, the
                                          // traceback takes type 0 and 1 alike): a finished key (score x 64 + KHC) is the horizontal candidate of the next column as it is

constexpr uint32_t CARRY_BATCH = 32;       // rows whose carries a wave takes at a time at most
constexpr uint32_t CARRY_MIN = 4;          // ... and at least (int32 rows; = how far a wave that keeps up runs behind its left neighbour)
constexpr uint32_t WAVE_MBOX = 64;         // entries of the LDS mailbox between two waves of a workgroup (a power of two >= 2 * CARRY_BATCH)
constexpr uint32_t MAX_WAVES = 16;         // waves per workgroup at most
constexpr uint32_t WG_POLL_LIMIT = 1u << 24;   // polls of a wave for another wave of its own workgroup (resident by construction) before it flags an internal error

template <int NWAVES> struct WaveMailT {   // LDS
    unsigned long long box[(NWAVES > 1 ? NWAVES - 1 : 1) * WAVE_MBOX];   // boundary b (between waves b and b + 1): entry of row i at [b][i % WAVE_MBOX] = {tag, carry}
    uint32_t consumed[NWAVES];                                           // boundary b: tag up to which wave b + 1 has taken the carries
};
using WaveMail = WaveMailT<MAX_WAVES>;     // (dp_rows addresses box[] and consumed[] through their own pointers: the layout of the largest serves all)

// ---- an edge shared by several workgroups ("members", one CU each): member m owns the waves [m*NW, (m+1)*NW) of the pipeline; the carry
// of its last wave travels through a tagged 64-bit word per row in HBM (relaxed device-scope atomics; the tag validates the word, no
// fences inside the DP).
struct DpCl {
    uint32_t mem, members;            // this member / members of the edge (1: no cluster)
    uint32_t stride;                  // rows per member in mbox (vcap + 1)
    uint32_t tag0;                    // tag of row i = tag0 + i (rows of all DPs of the edge numbered consecutively)
    unsigned long long* mbox;         // edge base
    uint32_t* err;                    // device-visible error word of the edge (set when a poll gives up)
    uint32_t poll_limit;              // polls before a waiter gives up and flags the edge instead of hanging the GPU (the host then redoes it unshared)
    uint32_t lanes;                   // lanes of the workgroup that take part in the DP (a "wide" member has more: they work in the graph phases only)
};
__device__ __forceinline__ uint32_t ld_dev(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_dev64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_wg64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_wg64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t ld_wg(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_wg(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// inclusive prefix maximum over the 64 lanes of a wave: the classic DPP sequence with the max fused into the DPP instruction
// (lanes without a source keep their value). s_nop 1 = the two wait states a DPP read needs after a VALU write of its source.
__device__ __forceinline__ int wave_incl_max(int v) {
    int x;
    asm volatile(
        "v_mov_b32 %0, %1\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(x) : "v"(v));
    return x;
}
// The same scan with the wait states a DPP read needs (two after a VALU write of its source) spent on scalar work of the row instead of s_nop - a lone
// wave pays ~8 cycles per `s_nop 1` (tools/dev_lonebench.hip). Scalar outputs: the LDS address of mailbox entry i and its tag, the row's ring slot (4 bits of
// its record) and the slot's byte offset, the record's rare-case bits; the nibble row pointer (dlo, dhi) moves on by dstep.
__device__ __forceinline__ int wave_incl_max_fill(int v, uint32_t i, uint32_t tag0, uint32_t mb_lds, uint32_t meta, uint32_t ring_w4, uint32_t dstep,
                                                  uint32_t& mb_addr, uint32_t& mb_tag, uint32_t& slot, uint32_t& rare, uint32_t& roff, uint32_t& dlo, uint32_t& dhi) {
    int x;
    // (every scalar operand through readfirstlane: a no-op for a value that already sits in a scalar register, and the only way to tell the compiler so)
    i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i); tag0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tag0); mb_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb_lds);
    meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)meta); ring_w4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_w4); dstep = (uint32_t)__builtin_amdgcn_readfirstlane((int)dstep);
    dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)dlo); dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)dhi);
    asm volatile(
        "v_mov_b32 %[x], %[v]\n\t"
        "s_and_b32 %[a], %[i], 63\n\t"
        "s_lshl_b32 %[a], %[a], 3\n\t"
        "v_max_i32_dpp %[x], %[v], %[v] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %[x], %[v], %[x] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %[x], %[v], %[x] row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
        "s_add_i32 %[a], %[a], %[mb]\n\t"
        "s_add_i32 %[t], %[i], %[tag0]\n\t"
        "v_max_i32_dpp %[x], %[x], %[x] row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
        "s_bfe_u32 %[slot], %[meta], 0x40008\n\t"
        "s_and_b32 %[rare], %[meta], 44\n\t"
        "v_max_i32_dpp %[x], %[x], %[x] row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_mul_i32 %[roff], %[slot], %[rw]\n\t"
        "s_add_u32 %[dlo], %[dlo], %[dstep]\n\t"
        "v_max_i32_dpp %[x], %[x], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_addc_u32 %[dhi], %[dhi], 0\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %[x], %[x], %[x] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : [x] "=&v"(x), [a] "=&s"(mb_addr), [t] "=&s"(mb_tag), [slot] "=&s"(slot), [rare] "=&s"(rare), [roff] "=&s"(roff), [dlo] "+s"(dlo), [dhi] "+s"(dhi)
        : [v] "v"(v), [i] "s"(i), [tag0] "s"(tag0), [mb] "s"(mb_lds), [meta] "s"(meta), [rw] "s"(ring_w4), [dstep] "s"(dstep) : "scc");
    // (... and back: the compiler takes what an asm statement writes for divergent - a v_cmp for every test of it, a waterfall loop around the buffer store)
    mb_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb_addr); mb_tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb_tag); slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    rare = (uint32_t)__builtin_amdgcn_readfirstlane((int)rare); roff = (uint32_t)__builtin_amdgcn_readfirstlane((int)roff);
    dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)dlo); dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)dhi);
    return x;
}
// 1 if a >= b, else 0, both wave-uniform: a scalar compare and select
__device__ __forceinline__ uint32_t s_ge_i32(int a, int b) {
    uint32_t r;
    asm("s_cmp_ge_i32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
// 1 if the lane mask has a bit set, else 0
__device__ __forceinline__ uint32_t s_nz_u64(unsigned long long m) {
    uint32_t r;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(m) : "scc");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
// byte 0 of four registers -> one dword
__device__ __forceinline__ uint32_t pack_b0(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u), cd = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
}
template <int CM> __device__ __forceinline__ void store_dirs(uint8_t* p, const uint32_t (&v)[CM], uint32_t keep) {
    if constexpr (CM >= 4) {
        uint32_t w[CM / 4];
#pragma unroll
        for (int q = 0; q < CM / 4; q++) w[q] = pack_b0(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]) & keep;
        if constexpr (CM == 4) *reinterpret_cast<uint32_t*>(p) = w[0];
        else if constexpr (CM == 8) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
        else {
#pragma unroll
            for (int q = 0; q < CM / 16; q++) reinterpret_cast<uint4*>(p)[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < CM; k++) p[k] = (uint8_t)(v[k] & keep);
    }
}

// low nibble of CM registers -> CM / 2 bytes (cell k in the low half of byte k / 2 for even k, the high half for odd k)
template <int CM> __device__ __forceinline__ void store_nibbles(uint8_t* p, const uint32_t (&v)[CM]) {
    static_assert(CM >= 4 && CM % 4 == 0, "4, 8, 16 or 32 columns per lane");
    uint32_t b[CM / 2];   // byte 0 of b[q] = the two cells 2q, 2q + 1 (higher bits are dropped by the byte packing)
#pragma unroll
    for (int q = 0; q < CM / 2; q++) b[q] = (v[2 * q] & 15u) | (v[2 * q + 1] << 4);
    if constexpr (CM == 4) *reinterpret_cast<uint16_t*>(p) = (uint16_t)__builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u);
    else {
        uint32_t w[CM / 8];
#pragma unroll
        for (int q = 0; q < CM / 8; q++) w[q] = pack_b0(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
        if constexpr (CM == 8) *reinterpret_cast<uint32_t*>(p) = w[0];
        else if constexpr (CM == 16) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
        else *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// the same nibbles through a raw buffer resource (base = the row, range = its bytes): offsets beyond the range are not written.
// Word 3 of the resource (0x00020000: raw, dword data format) and the rule the row loop relies on - "an access whose offset lies beyond num_records
// is dropped" - are the gfx9 family's; gfx10+ / gfx12 encode the word differently and check ranges per format. This file is written for gfx950 only:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "kernels/poa.hip stores its nibble rows through a gfx9 raw buffer resource (written for gfx950): another target needs the predicated store_nibbles()"
#endif
// (the row's resource covers W / 2 bytes = exactly the pitch of the nibble matrix: dp_rows steps its row pointer by the same W >> 1)
template <int CM> __device__ __forceinline__ void store_nibbles_buf(__amdgpu_buffer_rsrc_t r, const uint32_t off, const uint32_t (&v)[CM]) {
    static_assert(CM == 2 || (CM >= 4 && CM % 4 == 0), "2, 4, 8, 16 or 32 columns per lane");
    uint32_t b[CM / 2];
#pragma unroll
    for (int q = 0; q < CM / 2; q++) b[q] = (v[2 * q] & 15u) | (v[2 * q + 1] << 4);
    if constexpr (CM == 2) __builtin_amdgcn_raw_buffer_store_b8((char)b[0], r, (int)off, 0, 0);   // (two columns per lane - the members of the few-edge regime's shared edges: a byte per lane)
    else if constexpr (CM == 4) __builtin_amdgcn_raw_buffer_store_b16((short)__builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u), r, (int)off, 0, 0);
    else {
        uint32_t w[CM / 8];
#pragma unroll
        for (int q = 0; q < CM / 8; q++) w[q] = pack_b0(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
        // (8 columns per lane = the instances of the many-edge regime: the nibble rows leave as NON-TEMPORAL stores. A 13 000-edge call writes 0.45 TB of them, of
        // which the traceback reads back one byte in a few thousand; streamed past the L2 they leave it to the graph arrays and the far rows, whose round trips
        // are what the serial phases are made of - wave cycles of all workgroups of the 140 Mb call: -2.8 %. aux 2 = nt on gfx940/gfx950.)
#ifndef HX_NIB_AUX
#define HX_NIB_AUX 2
#endif
        if constexpr (CM == 8) __builtin_amdgcn_raw_buffer_store_b32(w[0], r, (int)off, 0, HX_NIB_AUX);
        else if constexpr (CM == 16) { typedef uint32_t u32x2 __attribute__((ext_vector_type(2))); __builtin_amdgcn_raw_buffer_store_b64((u32x2){w[0], w[1]}, r, (int)off, 0, 0); }
        else { typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[0], w[1], w[2], w[3]}, r, (int)off, 0, 0); }
    }
}

#ifdef HX_DP_PROF3   // development: per member of a shared edge, cycles inside the DP and cycles of them spent waiting for carries (phase slots 6 + member)
#define HX_DP_PROF
#define HX_DP_PROF2
#endif
#if defined(HX_DP_PROF) && !defined(HX_DP_PROF2)
#define DP_T(k) do { if (tid == 0) { const long long _n = clock64(); prof[k] += (unsigned long long)(_n - tprev); tprev = _n; } } while (0)
#else
#define DP_T(k) do { } while (0)
#endif
// The row loop is written for a lone wavefront's latency: on this hardware a VALU instruction costs ~5 cycles whether or not it depends
// on its predecessor, a taken scalar branch ~35, an LDS round trip ~75, the DPP scan ~90. So a row is ONE dispatch on where its
// predecessor lives (registers / LDS ring / anything else), then straight-line code: rare events (row spilled to HBM, sink row) share
// one not-taken branch, only rows with a non-adjacent reader are copied to the LDS ring (slot from the row's record), selects are
// arithmetic.
#ifdef HX_FARREAD_STORE   // (development: dead far-read rows store "nothing" and need a look, flags or not - as before the sticky far bit)
#define HX_FARREAD_RISKY(fb) true
#else
#define HX_FARREAD_RISKY(fb) (!(fb))
#endif
template <int CM, bool DIR, bool PRUNE, bool ONEW /* the workgroup is one wave (the 64-lane instances): no LDS mailbox on either side, no relay - known at compile time, the row loses its tests of them */>
__device__ __forceinline__ void dp_rows(const G& g, int32_t* __restrict__ H, uint8_t* __restrict__ D, uint8_t* __restrict__ Dwide, const uint32_t W, const uint32_t WH, const uint8_t* __restrict__ seq,
                        const uint32_t L_, const uint32_t V_, int32_t* ring, const uint32_t R_, const uint32_t ring_w_, const int match, const int mismatch, const int gap,
                        unsigned long long* wm_box, uint32_t* wm_cons, uint32_t* sink_row, int* sink_score, const uint32_t sink_cap, uint32_t& nSinkOut, const DpCl& cl, unsigned long long* prof,
                        const int thrT /* PRUNE: score threshold T of this alignment (PRUNE_OFF: nothing real is below it) */, const uint32_t lazy_on /* PRUNE: skipped waves poll rarely */, unsigned long long* pstat /* PRUNE: wave-rows, wave-rows skipped */,
                        const uint32_t far_n /* PRUNE: rows of H (far-read rows) of the edge */) {
    static_assert(!PRUNE || DIR, "pruned rows: direction-byte flavour only");
#if defined(HX_DP_PROF) && !defined(HX_DP_PROF2)
    long long tprev = clock64();
#endif
    // wave-uniform values the compiler cannot know to be uniform (they come through LDS / integer division / the thread index): in scalar
    // registers they turn the loop control, the ring slot arithmetic and the carry hand-over into scalar instructions and branches
    const uint32_t L = (uint32_t)__builtin_amdgcn_readfirstlane((int)L_), V = (uint32_t)__builtin_amdgcn_readfirstlane((int)V_);
    const uint32_t R = (uint32_t)__builtin_amdgcn_readfirstlane((int)R_), ring_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_w_);
    const uint32_t tid = threadIdx.x, NT = ONEW ? 64u : blockDim.x, lane = tid & 63u, wv = ONEW ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), NW = ONEW ? 1u : cl.lanes >> 6;
    const uint32_t ncol = L + 1;
    // A wide member (1024 lanes, NW of its 16 waves in the DP) other than the first lets a spare wave RELAY the carries that arrive through HBM
    // into an LDS mailbox: its first DP wave then takes them like any wave takes its left neighbour's (an LDS round trip per batch of rows
    // instead of a device-scope load, ~1.7 us, that it would sit through - and the pipeline runs at the speed of its slowest wave).
    constexpr uint32_t RELAY_BOX = MAX_WAVES - 2;                             // mailbox / consumed word of the relay (the 1024-lane instances have them; boundaries 0 .. NW-2 are the DP's)
    const bool relay_mode = !ONEW && NT == 1024u && NW + 2u <= 16u && cl.mem > 0;
    if (wv >= NW) {                                                           // (a wave of a wide member that sits the DP out)
        if (relay_mode && wv == NW + 1u && (uint64_t)(cl.mem * NW) * 64u * CM < ncol) {   // (wave NW + 1: not the SIMD of the wave it feeds)
            const unsigned long long* src = cl.mbox + (uint64_t)(cl.mem - 1) * cl.stride;
            unsigned long long* dst = wm_box + (size_t)RELAY_BOX * WAVE_MBOX;
            const uint32_t* cons = wm_cons + RELAY_BOX;
            bool dead = false;
            for (uint32_t ib = 0; ib < V; ib += 64) {
                const uint32_t ie = min(64u, V - ib);
                uint32_t nb = 0;
                for (uint32_t rb = 0; rb < ie; rb += nb) {
                    const uint32_t want = min(CARRY_BATCH, ie - rb), i0 = ib + rb + 1;   // (as many rows as have arrived, at least CARRY_MIN: like the DP waves)
                    nb = want;
                    unsigned long long v = (unsigned long long)(cl.tag0 + i0 + lane);   // (a relay that gave up still hands out tagged entries: the edge is flagged and redone)
                    for (uint32_t spin = 0; !dead; spin++) {
                        bool ok = true;
                        if (lane < want) { v = ld_dev64(src + i0 + lane); ok = (uint32_t)v == cl.tag0 + i0 + lane; }
                        const unsigned long long okm = __ballot(ok);
                        const uint32_t run = okm == ~0ull ? want : (uint32_t)__builtin_ctzll(~okm);
                        if (run >= min(want, CARRY_MIN)) { nb = run; break; }
                        if (spin > cl.poll_limit) { if (lane == 0) st_dev(cl.err, 1u); dead = true; v = (unsigned long long)(cl.tag0 + i0 + lane); break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    const uint32_t need = i0 + nb - 1 > WAVE_MBOX ? cl.tag0 + i0 + nb - 1 - WAVE_MBOX : 0;   // the entries overwritten must have been taken
                    for (uint32_t spin = 0; need; spin++) {
                        const uint32_t got = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_wg(cons));
                        if ((int32_t)(got - need) >= 0) break;
                        if (spin > WG_POLL_LIMIT) { if (lane == 0) st_dev(cl.err, 2u); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (lane < nb) st_wg64(dst + ((i0 + lane) & (WAVE_MBOX - 1)), v);
                }
            }
        }
        return;
    }
    const uint32_t gw = cl.mem * NW + wv;                                     // this wave's place in the edge's pipeline
    if ((uint64_t)gw * 64u * CM >= ncol) return;                              // the wave owns no real column of this sequence
    const uint32_t gt = gw * 64u + lane;                                      // lane index over all waves
    const bool has_in = gw > 0;                                               // a wave on the left feeds the horizontal carry ...
    const bool in_lds = !ONEW && (wv > 0 || relay_mode);                                 // ... through the workgroup's LDS mailbox, or (first wave of a member without a relay wave) through HBM
    const bool has_out = (uint64_t)(gw + 1) * 64u * CM < ncol;                // a wave on the right owns real columns (the host sized the pipeline for the longest sequence)
    const bool out_lds = !ONEW && wv + 1 < NW;
    // (wave-uniform tests of the row loop as 32-bit scalars: a test of a lane-mask boolean is `s_andn2 vcc` + a vcc branch, ~32 cycles for a lone
    // wave against ~15 for `s_cmp` + an scc branch: tools/dev_lonebench.hip)
    const uint32_t out_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(has_out && out_lds)), out_h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(has_out && !out_lds));
    const unsigned long long* mb_in_h = cl.mbox + (uint64_t)(cl.mem ? cl.mem - 1 : 0) * cl.stride;
    unsigned long long* mb_out_h = cl.mbox + (uint64_t)cl.mem * cl.stride;
    const unsigned long long* mb_in_l = wm_box + (size_t)(wv ? wv - 1 : relay_mode ? RELAY_BOX : 0) * WAVE_MBOX;
    unsigned long long* mb_out_l = wm_box + (size_t)wv * WAVE_MBOX;
    uint32_t* cons_in = wm_cons + (wv ? wv - 1 : relay_mode ? RELAY_BOX : 0);   // what this wave has taken from the boundary on its left
    const uint32_t* cons_out = wm_cons + wv;                                  // what the wave on the right has taken from this wave's mailbox
    if (has_in && in_lds && lane == 0) st_wg(cons_in, cl.tag0);               // everything of earlier DPs counts as taken (a wave may have sat out a short sequence)
    const uint32_t* farslot = reinterpret_cast<const uint32_t*>(g.pred);      // per rank: row of H that holds the far-read row (consensus scratch, free during the DP)
    const uint32_t* wideslot = g.wslot;                                       // per rank: row of the wide-row pool (rows with more than 4 predecessors)
    const uint32_t j0 = gt * CM;
    const bool live = j0 <= L;                       // the chunk holds at least one real column: only such chunks touch HBM
    const bool owns_last = live && L < j0 + CM;
    const uint32_t klast = owns_last ? L - j0 : 0;
    const uint32_t hleft = W + gw;                   // H rows end with one word per wave: the value left of the wave's first column (its own copy)
    // bases under the lane's columns, 2 bits per column (bit pair k); columns without a base (column 0, padding) never match
    using mask_t = typename std::conditional<(CM <= 16), uint32_t, unsigned long long>::type;
    static_assert(CM <= 32, "at most 32 columns per lane");
    mask_t bases = 0, nobase = 0;
    uint32_t onehot = 0;   // up to 8 columns per lane: bit 4 k + letter of the base under column k - a row's match bits are ONE shift by its letter (no base: no bit)
#pragma unroll
    for (int k = 0; k < CM; k++) {
        const uint32_t j = j0 + k;
        if (j >= 1 && j < ncol) { bases |= (mask_t)seq[j - 1] << (2 * k); if constexpr (CM <= 8) onehot |= 1u << (4 * k + seq[j - 1]); } else nobase |= (mask_t)1 << (2 * k);
    }
    const int mm64 = mismatch * 64, g64 = gap * 64, m64 = match * 64;
    const int jg0 = (int)j0 * g64;
    const int mdN = m64 - g64 + 15 - KHC, mdW = m64 - g64 + KD - KHC, gvN = g64 + 11 - KHC, gvW = g64 + KV - KHC;   // diagonal (match) / vertical constants of 4-bit and wide rows
    // The previous row is still in the registers of the lanes that own its columns (tp, lnp): a successor that follows it immediately
    // reads it there - no LDS round trip on the most common dependency. Every row a NON-adjacent successor reads ("kept") lives in the LDS
    // ring, R slots in the order they are produced; rows nobody else reads are not written at all. A predecessor reference is a code from
    // the CSR build (1 + ring slot; 13 = the previous row; 14 = the virtual row 0 that source nodes start from; 15 = a kept row that left
    // the ring: HBM), the row's own slot sits in its record. Per wave: CM planes of 65 words, column t*CM+k at word 65*CM*wv + 65*k + 1 + t;
    // word 0 of the wave's LAST plane holds the value left of the wave's first column, so "the column left of my chunk" is word
    // 65*(CM-1) + t for EVERY lane: one load, no select. Plane offsets are instruction offsets of ONE address register per slot.
    constexpr uint32_t PW = 65u;
    int32_t* const ring_me = ring + wv * (65u * CM) + lane;
    if (!DIR && live) {                                 // the score-matrix traceback reads row 0 like any other row
        int pl[CM];
#pragma unroll
        for (int k = 0; k < CM; k++) pl[k] = (jg0 + k * g64) >> 6;
        store_chunk_i32<CM>(H + j0, pl);
    }
    uint32_t nsink = 0;
    // row records of 64 rows per register: the current batch (C), the next one (N, complete with the third and fourth predecessor entries
    // of the rows that have them - a gather that needs the records first), and the one after it (F) in flight
    uint32_t mC = 0, aC = 0, bC = 0, oC = 0, cC = 0, dC = 0, fC = 0, mN = 0, aN = 0, bN = 0, oN = 0, cN = 0, dN = 0, fN = 0, mF = 0, aF = 0, bF = 0, oF = 0;
    auto fetch = [&](uint32_t base, uint32_t& m, uint32_t& a, uint32_t& b, uint32_t& o) {
        const uint32_t r = base + lane;
        if (r < V) { m = g.row_meta[r]; a = g.row_pred0[r]; b = g.row_pred1[r]; o = g.row_pred_off[r]; }
    };
    // Second stage, a batch ahead of its use. Everything that would otherwise be a DEPENDENT load on the row's own path is gathered here:
    // the third and fourth predecessor entries, and (direction-byte flavour) the H slots of far rows - a far first / second predecessor
    // entry gets its slot in place of the rank (the row loop never needs the rank), a row that is stored for a far reader its own slot.
    auto fetch_more = [&](uint32_t base, uint32_t m, uint32_t o, uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t& f) {
        if (base + lane < V) {
            if ((m >> META_NP) > 2u) c = g.pred_rank[o + 2];
            if ((m >> META_NP) > 3u) d = g.pred_rank[o + 3];
            if (DIR) {
                if ((a >> 28) == 15u) a = 0xf0000000u | farslot[a & 0x0fffffffu];
                if ((b >> 28) == 15u && (m >> META_NP) > 1u) b = 0xf0000000u | farslot[b & 0x0fffffffu];
                if (m & 8u) f = farslot[base + lane];
            }
        }
    };
    fetch(0, mN, aN, bN, oN);
    fetch(64, mF, aF, bF, oF);
    fetch_more(0, mN, oN, aN, bN, cN, dN, fN);
    int32_t* hrow = H;
    uint32_t dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)D), dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)D >> 32));   // the row of direction nibbles (a scalar pointer in two halves: wave_incl_max_fill moves it)
    const uint32_t mb_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)mb_out_l);   // LDS byte address of this wave's mailbox towards the right
    const uint32_t ring_w4 = ring_w * 4u, tag0_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)cl.tag0), dstep_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(DIR ? W >> 1 : 0u));
    // predecessor row `ent` (slot << 28 | rank): its columns under this lane and the value left of them
    int tp[CM], lnp = NEGK;                  // the previous row's finished keys under this lane, and the key left of the wave's first column (lane 0's is used)
#pragma unroll
    for (int k = 0; k < CM; k++) tp[k] = NEGK;
    // ---- PRUNE: exact score-bound pruning at the granularity this pipeline works at, (row, wave). U(i, j) = H[i][j] + match x (L - j) bounds the
    // final score of every path through cell (i, j) and never grows along a path, so with a threshold T <= the final score S no cell with U < T
    // lies on an optimal path, every cell of an optimal path keeps its exact value whatever stands in the dead cells (anything <= their true
    // value), and the traceback - which compares the candidates of optimal cells only - is unchanged. In de-ramped keys (X = H - gap j, never
    // decreasing along a row) U = X + match L - (match - gap) j: a lane's columns are all dead when the key of its LAST column, taken at its
    // FIRST, is below T (`thr_lane`), the carry entering the wave is dead below `thr_cin`. A wave SKIPS a row (no predecessor reads, no cells, no
    // scan, no ring copy, no nibbles: it forwards the incoming carry under the row's tag) when the carry is dead and none of its predecessor
    // rows was FLAGGED by this wave; a computed row is flagged when one of its lanes or its carry-in is live. The flags of the rows a successor
    // can name live in one scalar word FM, bit = the location code of a predecessor entry: 1 + ring slot, 13 the previous row, 14 the virtual
    // row 0, 15 a row read back from HBM (always set: a skipped row with a far reader stores "nothing" there, so it may be read). Predecessors
    // whose flag is clear are not read at all (their registers / ring slot hold an older row). T is the caller's: poa_edge checks S >= T
    // afterwards and repeats the alignment otherwise. (kernels.h: PRUNE_OFF; oracle.cpp prune_sim = this rule on the CPU, a statistic.)
    int thr_lane = 0; uint32_t FM = 0xffffu, n_dead = 0, n_bulk = 0, lazy = 0; int thr_cin = 0;
    if constexpr (PRUNE) {
        const int mg = match - gap, thr_base = thrT - match * (int)L;
        const int c0 = (int)(gw * 64u * CM);
        // (padding lanes - no real column: the last wave's - never flag a row. They read the FIRST chunk of a far row, another wave's, and what stands there when
        // that wave had no reason to store the row is whatever the slot held before: harmless for the cells, but a flag from it made the pruning counters differ
        // from run to run)
        thr_lane = live ? (thr_base + mg * (int)j0) * 64 : INT32_MAX;
        thr_cin = __builtin_amdgcn_readfirstlane((thr_base + mg * (c0 - 1)) * 64);
        const uint32_t f0 = (uint32_t)(match * (int)L - mg * max(c0 - 1, 0) >= thrT);
        // (bit 15, "a row in HBM": set for good where the far rows have no flags of their own; else it is raised by the first far row this wave stores LIVE - until
        // then every far row it could name is dead, and above the band of the matrix a record that names one is as dead as its neighbours)
        FM = (uint32_t)__builtin_amdgcn_readfirstlane((int)((far_n <= 2048u ? 0u : 0x8000u) | (f0 << 14)));   // nothing in the ring, no previous row yet
    }
    // PRUNE: the flags of the rows kept in HBM ("far" rows, code 15), one bit per row of H in ONE register of the wave (lane = slot / 32: 2 048 slots; an edge with
    // more has every bit set for good - its far rows are read as they always were). A far row whose flag is clear is not fetched: its reader takes "nothing",
    // like the readers of an unflagged ring row do - in a dead region of the matrix that was an HBM round trip (~2 us under load) on the path of a row that came
    // out dead anyway, one row in forty.
    const bool far_bits = PRUNE && far_n <= 2048u;
    uint32_t farbits = far_bits ? 0u : 0xffffffffu;
    auto far_set = [&](const uint32_t slot, const uint32_t fl) {   // (slot, fl: wave-uniform)
        if (far_bits && lane == ((slot >> 5) & 63u)) farbits = (farbits & ~(1u << (slot & 31u))) | (fl << (slot & 31u));
    };
#ifdef HX_RING_PREFETCH
    // (development, round 6: the ring rows the NEXT row names as its first / second predecessor are requested while the current row is in its scan - an LDS round trip
    // is ~100 cycles for a lone wave and a row makes 1.14 of them. pf_row = the row they were requested for.)
    constexpr bool PF = !PRUNE && DIR;
    int q0[CM], q0l = NEGK, q1[CM], q1l = NEGK;
    uint32_t pf_row = 0xffffffffu, pf_s0 = 0xffu, pf_s1 = 0xffu;   // ... and the ring slots (location codes) they were read from
#pragma unroll
    for (int k = 0; k < CM; k++) { q0[k] = NEGK; q1[k] = NEGK; }
#else
    constexpr bool PF = false;
#endif
    auto pred_row = [&](const uint32_t ent, int (&hp)[CM], int& left, const bool slot_known) {
        const uint32_t loc = ent >> 28;
        if (__builtin_expect(loc == 13u, 1)) {   // the previous row: registers (the likely case falls through: a taken scalar branch costs a lone wave ~35 cycles)
#pragma unroll
            for (int k = 0; k < CM; k++) hp[k] = tp[k];
            left = wave_shift_up1(tp[CM - 1], lnp);
        } else if (__builtin_expect(loc < 13u, 1)) {    // in the LDS ring
            const int32_t* S = ring_me + (size_t)(loc - 1) * ring_w;
#pragma unroll
            for (int k = 0; k < CM; k++) hp[k] = S[k * PW + 1];
            left = S[(CM - 1) * PW];
        } else if (loc == 14u) {                 // a source node starts from the virtual row 0
#pragma unroll
            for (int k = 0; k < CM; k++) hp[k] = KHC;   // (row 0 is the gap ramp itself)
            left = gt > 0 ? KHC : NEGK;
        } else {                                 // kept row that fell out of the ring: HBM
            // (no divergent branch in here: with one, the compiler structurises the whole dispatch and every row pays a flag test. Padding lanes
            // - columns beyond the sequence, last wave only - read the row's first chunk instead: their keys reach no real column)
            // with direction bytes only the rows a far successor reads are in HBM, in the slots the CSR build gave them
            const uint32_t hr = DIR ? (slot_known ? ent & 0x0fffffffu : farslot[ent & 0x0fffffffu]) : (ent & 0x0fffffffu) + 1;
            if constexpr (PRUNE) {
                const uint32_t hs = (uint32_t)__builtin_amdgcn_readfirstlane((int)hr);
                if ((((uint32_t)__builtin_amdgcn_readlane((int)farbits, (int)((hs >> 5) & 63u)) >> (hs & 31u)) & 1u) == 0u) {   // an unflagged far row: not read
#pragma unroll
                    for (int k = 0; k < CM; k++) hp[k] = NEGK;
                    left = NEGK;
                    return;
                }
            }
            const int32_t* Grow = H + (uint64_t)hr * WH;
            const uint32_t jl = live ? j0 : 0u;
            const int32_t* Gp = Grow + jl;
            load_chunk_i32<CM>(Gp, hp);
            // the key left of the chunk: the neighbour's last column - lane 0: the wave's own copy (the column belongs to a wave that may be far ahead)
            const int32_t* lp = (lane > 0 && live) ? Gp - 1 : has_in ? Grow + hleft : Grow;
            left = *lp;
            if (!DIR) {                          // the score matrix holds plain scores
#pragma unroll
                for (int k = 0; k < CM; k++) hp[k] = (hp[k] << 6) - (jg0 + k * g64) + KHC;
                left = (left << 6) - (jg0 - g64) + KHC;
            }
            left = gt > 0 ? left : NEGK;
            // the loaded values are consumed HERE: otherwise the wait for them is placed where the three sources of a predecessor row
            // join - on the path of every row - and waits for the previous rows' direction stores as well (vmcnt counts them)
#pragma unroll
            for (int k = 0; k < CM; k++) asm volatile("" : "+v"(hp[k]));
            asm volatile("" : "+v"(left));
        }
    };
    // (a one-wave workgroup has no LDS mailbox on either side - its carries come from and go to HBM, window by window - and takes the 64 rows of a record batch
    // at once: half as many round trips for the carries, which is what a dead batch costs now that its rows leave in runs)
    const uint32_t cbatch = (uint32_t)__builtin_amdgcn_readfirstlane((int)(NW == 1u ? 64u : CARRY_BATCH));
    for (uint32_t ib = 0; ib < V; ib += 64) {
        // the batches move up (the only waits for these loads: everything was requested at least 64 rows ago), another one goes in flight
        mC = mN; aC = aN; bC = bN; oC = oN; cC = cN; dC = dN; fC = fN;
        mN = mF; aN = aF; bN = bF; oN = oF;
        fetch(ib + 128, mF, aF, bF, oF);
        fetch_more(ib + 64, mN, oN, aN, bN, cN, dN, fN);
        const uint32_t ie = min(64u, V - ib);
        uint32_t nb = 0;
        for (uint32_t rb = 0; rb < ie; rb += nb) {
            // rows i0 .. i0 + nb - 1: up to CARRY_BATCH rows of the record batch - as many as have their carries in the mailbox, at least CARRY_MIN (a
            // wave follows its left neighbour at that distance when it keeps up, and the mailbox's 64 entries still absorb a neighbour's hiccup)
            const uint32_t want = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(cbatch, ie - rb)), i0 = ib + rb + 1;
            nb = want;
            int cinV = NEGK;     // lane r: carry into this wave for row i0 + r
            if (has_in) {
#ifdef HX_DP_PROF3
                const long long tw0 = clock64();
#endif
                for (uint32_t spin = 0;; spin++) {
                    unsigned long long v = 0;
                    bool ok = true;
                    if (lane < want) {
                        v = in_lds ? ld_wg64(mb_in_l + ((i0 + lane) & (WAVE_MBOX - 1))) : ld_dev64(mb_in_h + i0 + lane);
                        ok = (uint32_t)v == cl.tag0 + i0 + lane;
                    }
                    const unsigned long long okm = __ballot(ok);
                    const uint32_t run = okm == ~0ull ? want : (uint32_t)__builtin_ctzll(~okm);   // leading rows whose carries have arrived
                    // (PRUNE: a wave whose last batch was skipped whole is AHEAD of the band - it is not what its edge waits for, but a poll every ~130 cycles
                    // takes issue slots from the waves that are: it waits for whole batches and sleeps 16 times as long between polls)
                    if (run >= min(want, PRUNE && lazy ? want : CARRY_MIN)) { nb = run; cinV = (int)(uint32_t)(v >> 32); break; }
                    if (spin > (in_lds ? WG_POLL_LIMIT : cl.poll_limit)) { if (lane == 0) st_dev(cl.err, in_lds ? 2u : 1u); break; }
                    if (PRUNE && lazy) __builtin_amdgcn_s_sleep(32);
                    else if (in_lds) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(8);
                }
                if (in_lds && lane == 0) st_wg(cons_in, cl.tag0 + i0 + nb - 1);   // the entries of these rows may be written again
#ifdef HX_DP_PROF3
                if (tid == 0) prof[0] += (unsigned long long)(clock64() - tw0);
#endif
            }
            if (has_out && out_lds) {   // the rows of this batch overwrite the entries of the rows WAVE_MBOX earlier: the wave on the right must have taken those
                const uint32_t need = i0 + nb - 1 > WAVE_MBOX ? cl.tag0 + i0 + nb - 1 - WAVE_MBOX : 0;
                for (uint32_t spin = 0; need; spin++) {
                    const uint32_t got = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld_wg(cons_out));
                    if ((int32_t)(got - need) >= 0) break;
                    if (spin > WG_POLL_LIMIT) { if (lane == 0) st_dev(cl.err, 2u); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            // (a row's record words are read out of their lane during the row BEFORE: a scalar instruction that consumes a readlane's result at once
            // waits ~14 cycles for it)
            // PRUNE: rows skipped in RUNS. `bad` = the rows of the batch that cannot be skipped without a look at them (bit r = row i0 + r): a live carry, or
            // a record that names a predecessor outside the ring and the previous row (the virtual row 0 where this wave has it flagged, a row in HBM, a
            // fifth predecessor) or that is itself read back from HBM (it has to store "nothing" there) - all of it read off the 64 row records in their
            // lanes, once per batch. Wherever nothing in the ring or the previous row is flagged (FM), the rows up to the next bad one are dead and leave
            // together: their carries go out as one vector store. (Round 5, second half: the first version skipped whole batches only, and one risky row in
            // fourteen left 45 % of the dead rows to the row-by-row path at ~750 cycles each under load.)
            unsigned long long bad = 0, farref = 0;
            if constexpr (PRUNE) {
                const uint32_t np_l = mC >> META_NP;
                // (codes 14 and 15 - the virtual row 0, a row in HBM - name something live only where this wave has their bit of FM set. A row that is itself read back
                // from HBM needs a look only where the far rows have no flags: with them its bit is clear until somebody stores it live, and nobody fetches it)
                auto outside = [&](const uint32_t ent) -> bool { const uint32_t c = ent >> 28; return c >= 14u && ((FM >> c) & 1u) != 0u; };
                const bool risky = outside(aC) || (np_l > 1u && outside(bC)) || (np_l > 2u && outside(cC)) || (np_l > 3u && outside(dC)) || np_l > 4u || (HX_FARREAD_RISKY(far_bits) && (mC & 8u) != 0u);
                const bool clive = lane < nb && cinV >= thr_cin;
                bad = (__builtin_amdgcn_ballot_w64(risky) >> rb) | __builtin_amdgcn_ballot_w64(clive);
                if ((FM & 0x8000u) == 0u) {   // the rows that name a far row: they need a look from the moment one is stored live (below)
                    const bool refs_far = (aC >> 28) == 15u || (np_l > 1u && (bC >> 28) == 15u) || (np_l > 2u && (cC >> 28) == 15u) || (np_l > 3u && (dC >> 28) == 15u);
                    farref = __builtin_amdgcn_ballot_w64(refs_far) >> rb;
                }
                if (nb < 64u) { bad &= (1ull << nb) - 1ull; farref &= (1ull << nb) - 1ull; }   // (a bit beyond the batch would send a run past its end)
            }
            // (the run is looked for where one can begin - at the batch's first row and behind a row that was skipped by itself - not on the path of a live row: as a
            // test at the head of every row it cost ten scalar instructions, and a 13 000-edge call 4 %)
            auto skip_run = [&](const uint32_t from) -> uint32_t {
                if ((FM & 0x3ffeu) != 0u || from >= nb) return 0u;
                const unsigned long long rest = bad >> from;
                const uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : nb - from;
                if (run != 0u) {
                    if (lane >= from && lane < from + run) {
                        const unsigned long long ent = (unsigned long long)(tag0_s + i0 + lane) | ((unsigned long long)(uint32_t)cinV << 32);
                        if (out_l != 0u) st_wg64(mb_out_l + ((i0 + lane) & (WAVE_MBOX - 1)), ent);
                        if (out_h != 0u) st_dev64(mb_out_h + i0 + lane, ent);
                    }
                    n_dead += run; n_bulk += run;
                    const unsigned long long dp_ = (((unsigned long long)dhi << 32) | dlo) + (unsigned long long)dstep_s * run;
                    dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dp_); dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dp_ >> 32));
                }
                return run;
            };
            const uint32_t dead_before = n_dead;
            uint32_t rj0 = 0;
            if constexpr (PRUNE) rj0 = skip_run(0u);
            uint32_t meta_nx = __builtin_amdgcn_readlane(mC, (rb + rj0) & 63u), p0_nx = __builtin_amdgcn_readlane(aC, (rb + rj0) & 63u);
            // The row, in two forms of one body. FAST (direction bytes, no pruning: the instances of the few-edge regime, where ONE wave's instruction count per row is
            // what the call waits for) = a row whose record names ONE predecessor, the previous row: three rows in five. Such rows run in a loop of their own
            // (below): the cells come straight out of the previous row's registers (as one of three sources joined in one set of registers the compiler copies them:
            // six v_mov), the row format's constants are the 4-bit ones, and neither the dispatch on the first predecessor's location nor the branch around the
            // later predecessors exists. (Round 5 tried the same cells as a block inside the one loop: + 14 %, through the two taken branches around it.)
            uint32_t rj = rj0;
            auto row = [&](auto fast_tag) __attribute__((always_inline)) {
                constexpr bool FAST = decltype(fast_tag)::value;
                const uint32_t ri = rb + rj, i = ib + ri + 1;
                const uint32_t meta = meta_nx, p0 = p0_nx;
                const uint32_t npred = meta >> META_NP;
                uint32_t cin_live = 0, fl0 = 1, flB = 1, flC = 1, flD = 1;
                if constexpr (PRUNE) {
                    // does anything this wave can read for the row still reach T? (all scalar: flags of the predecessor entries, the carry's test)
                    const int cin_e = __builtin_amdgcn_readlane(cinV, rj);
                    cin_live = s_ge_i32(cin_e, thr_cin);   // (as a C comparison the flag became a lane mask and the whole test vector code: v_cndmask, v_or, v_cmp_ne, a vcc branch)
                    fl0 = (FM >> (p0 >> 28)) & 1u;
                    uint32_t act = cin_live | fl0;
                    if (npred > 1) {
                        flB = (FM >> ((uint32_t)__builtin_amdgcn_readlane(bC, ri) >> 28)) & 1u; act |= flB;
                        if (npred > 2) {
                            flC = (FM >> ((uint32_t)__builtin_amdgcn_readlane(cC, ri) >> 28)) & 1u; act |= flC;
                            if (npred > 3) { flD = (FM >> ((uint32_t)__builtin_amdgcn_readlane(dC, ri) >> 28)) & 1u; act |= flD | ((npred + 3u) >> 3); }   // (non-zero for a fifth predecessor)
                        }
                    }
                    if (act == 0u) {
                        // ---- a skipped row: the carry passes through, the row's flags are cleared, a far reader finds "nothing"
                        n_dead++;
                        const uint32_t slot_d = (meta >> META_SLOT) & 15u;
                        FM &= ~(0x2000u | (2u << slot_d));   // (slot 15 = not kept: bit 16, which nobody reads)
                        meta_nx = __builtin_amdgcn_readlane(mC, (ri + 1) & 63u); p0_nx = __builtin_amdgcn_readlane(aC, (ri + 1) & 63u);
                        {
                            const unsigned long long ent = (unsigned long long)(tag0_s + i) | ((unsigned long long)(uint32_t)cin_e << 32);
                            if (out_l != 0u) { if (lane == 63) *(volatile __attribute__((address_space(3))) unsigned long long*)(uintptr_t)(mb_lds + ((i & (WAVE_MBOX - 1)) << 3)) = ent; }
                            if (out_h != 0u) { if (lane == 63) st_dev64(mb_out_h + i, ent); }
                        }
                        {   // the nibble row pointer moves on (live rows: inside the scan)
                            const unsigned long long dp_ = (((unsigned long long)dhi << 32) | dlo) + dstep_s;
                            dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dp_); dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dp_ >> 32));
                        }
                        if (__builtin_expect((meta & 8u) != 0u, 0)) {   // a far successor will read this row from HBM: keys of "nothing" (its flag is always taken for set)
                            const uint32_t fslot = __builtin_amdgcn_readlane(fC, ri);
                            if (HX_FARREAD_RISKY(far_bits) && live) {   // (with flags: the row's bit is clear - every DP begins with none set - and nobody fetches an unflagged row)
                                int32_t* F = H + (uint64_t)fslot * WH;
                                int ng[CM];
#pragma unroll
                                for (int k = 0; k < CM; k++) ng[k] = NEGK;
                                store_chunk_i32<CM>(F + j0, ng);
                                if (lane == 0 && has_in) F[hleft] = NEGK;
                            }
                        }
                        {   // the rows behind this one, up to the next that needs a look
                            const uint32_t run = skip_run(rj + 1u);
                            if (run != 0u) { rj += run; meta_nx = __builtin_amdgcn_readlane(mC, (rb + rj + 1u) & 63u); p0_nx = __builtin_amdgcn_readlane(aC, (rb + rj + 1u) & 63u); }
                        }
                        return;
                    }
                }
                // mismatch bits of this row: bit 2k set <=> the base under column k differs from the row's letter
                mask_t mis = 0;
                uint32_t hit = 0;     // (CM <= 8) bit 4 k set <=> the base under column k is the row's letter
                if constexpr (CM <= 8) hit = onehot >> (meta & 3u);
                else {
                    const mask_t x = bases ^ ((mask_t)(meta & 3u) * (mask_t)0x5555555555555555ull);
                    mis = x | (x >> 1) | nobase;
                }
                // Move codes (the low 6 bits of a key while a row is computed; they are masked off before the row is used as a predecessor, so the
                // format is the row's own). A row with at most 4 predecessors uses 4 bits - type * 4 + 3 - predecessor slot - which are its
                // traceback nibble as they are; a "wide" row (rare) uses type * 16 + 15 - slot and stores a byte per cell in a side pool.
                const bool wide = !FAST && (!DIR || (meta & 32u));   // (a fast row has one predecessor: never the wide format)
                // (a diagonal move leaves the ramp of column j - 1 for that of column j; a finished key carries KHC; both formats' constants wait in
                // scalar registers: one bit test and three selects per row)
                const int md = wide ? mdW : mdN, gv = wide ? gvW : gvN, mmd = wide ? mdW + (mm64 - m64) : mdN + (mm64 - m64);
                auto score_of = [&](int k) -> int {   // 64 x substitution score of column k + the diagonal move code
                    if constexpr (CM <= 8 && FAST) {   // (the bit field through an asm statement: with constant terms around it the compiler turns one of the cells into v_and + v_cmp + two v_mov + v_cndmask)
                        int bit;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(bit) : "v"(hit), "n"((4 * k) & 31));
                        return mmd + ((m64 - mm64) & bit);
                    }
                    if constexpr (CM <= 8) return mmd + ((m64 - mm64) & __builtin_amdgcn_sbfe((int)hit, (4 * k) & 31, 1));   // (-1 on a match)
                    int neg;   // -1 on a mismatch, 0 on a match
                    if constexpr (CM <= 16) neg = __builtin_amdgcn_sbfe((int)mis, 2 * k, 1);
                    else neg = 2 * k < 32 ? __builtin_amdgcn_sbfe((int)(uint32_t)mis, (2 * k) & 31, 1) : __builtin_amdgcn_sbfe((int)(uint32_t)(mis >> 32), (2 * k) & 31, 1);
                    return md + ((mm64 - m64) & neg);
                };
                hrow += WH;
                DP_T(0);   // row decode
                int m[CM];
                if constexpr (FAST) {
                    // (see the loops below: ONE predecessor, the previous row - its cells straight from the registers they are in, the 4-bit row format's constants,
                    // no dispatch, nothing behind the cells to skip)
                    const int left = wave_shift_up1(tp[CM - 1], lnp);
#pragma unroll
                    for (int k = 0; k < CM; k++) m[k] = max((k == 0 ? left : tp[k - 1]) + score_of(k), tp[k] + gv);
                } else
                if (!PRUNE || __builtin_expect(fl0 != 0u, 1)) {   // the first predecessor (or row 0): diagonal and vertical move
#ifndef HX_NO_PREV_DIRECT
                    if (PRUNE && __builtin_expect((p0 >> 28) == 13u, 1)) {   // (PRUNE = the instances of the many-edge regime; the row of the 4-column instances a lone wave runs got 14 % SLOWER with this block: 292 -> 332 M cycles on the longest 12 Mb edge)
                        // the previous row: its cells straight from the registers they are in. (Through pred_row the three sources of a predecessor row join
                        // in ONE set of registers and the compiler copies the previous row into them - ten v_mov per row of the 8-column instances. The
                        // statement at the end keeps this block from being merged with the general one below.)
                        const int left = wave_shift_up1(tp[CM - 1], lnp);
#pragma unroll
                        for (int k = 0; k < CM; k++) m[k] = max((k == 0 ? left : tp[k - 1]) + score_of(k), tp[k] + gv);
                        asm volatile("" : "+v"(m[CM - 1]));
                    } else
#endif
                    {
                        int hp[CM], left;
#ifdef HX_RING_PREFETCH
                        if (PF && pf_row == i && pf_s0 == (p0 >> 28)) {
#pragma unroll
                            for (int k = 0; k < CM; k++) hp[k] = q0[k];
                            left = q0l;
                        } else
#endif
                        pred_row(p0, hp, left, true);
#pragma unroll
                        for (int k = 0; k < CM; k++) m[k] = max((k == 0 ? left : hp[k - 1]) + score_of(k), hp[k] + gv);
                    }
                } else {                                          // (PRUNE: a skipped row is not read)
#pragma unroll
                    for (int k = 0; k < CM; k++) m[k] = NEGK;
                }
                if (!FAST && npred > 1) {   // (two rows in five at 25-45x)
                    // the maximum over the predecessors: the low bits carry the move type and 15 - p, so ONE running maximum does it all
                    // (a diagonal beats a vertical move of the same score, the first predecessor in in-edge order beats the later ones).
                    // The second, third and fourth predecessor are spelled out - their entries come with the row records, one readlane each, and most of
                    // them live in the LDS ring, which is read here without the general dispatch; a loop that picks the entry by its index and then
                    // dispatches compiles into a dozen flag tests per predecessor (~300 cycles for a lone wave).
                    auto more = [&](const uint32_t ent, const int ps, const bool slot_known) {
                        int hp[CM], left;
                        const uint32_t loc = ent >> 28;
                        const bool inring = loc - 1u < 12u;
                        // the ring row is requested FIRST, whatever the entry is (another kind asks for slot 0 and is overwritten below): one not-taken
                        // test on the usual path instead of an if / else whose join the compiler guards with a flag test
                        const int32_t* S = ring_me + (size_t)(inring ? loc - 1u : 0u) * ring_w;
#pragma unroll
                        for (int k = 0; k < CM; k++) hp[k] = S[k * PW + 1];
                        left = S[(CM - 1) * PW];
                        if (__builtin_expect(!inring, 0)) pred_row(ent, hp, left, slot_known);
                        // (diagonal: ONE three-operand add per cell - predecessor key + the cell's substitution term + the slot's code offset)
                        const int mps = -ps, gvp = gv - ps;
#pragma unroll
                        for (int k = 0; k < CM; k++) {
                            int dg;
                            asm("v_add3_u32 %0, %1, %2, %3" : "=v"(dg) : "v"(k == 0 ? left : hp[k - 1]), "v"(score_of(k)), "s"(mps));
                            m[k] = max(m[k], max(dg, hp[k] + gvp));
                        }
                    };
#ifdef HX_RING_PREFETCH
                    const uint32_t entB_ = __builtin_amdgcn_readlane(bC, ri);
                    if (PF && pf_row == i && pf_s1 == (entB_ >> 28)) {
                        const int gvp = gv - 1;
#pragma unroll
                        for (int k = 0; k < CM; k++) {
                            int dg;
                            asm("v_add3_u32 %0, %1, %2, %3" : "=v"(dg) : "v"(k == 0 ? q1l : q1[k - 1]), "v"(score_of(k)), "s"(-1));
                            m[k] = max(m[k], max(dg, q1[k] + gvp));
                        }
                    } else if (!PRUNE || flB != 0u) more(entB_, DIR ? 1 : 0, true);
#else
                    if (!PRUNE || flB != 0u) more(__builtin_amdgcn_readlane(bC, ri), DIR ? 1 : 0, true);
#endif
                    if (npred > 2) {
                        if (!PRUNE || flC != 0u) more(__builtin_amdgcn_readlane(cC, ri), DIR ? 2 : 0, false);
                        if (npred > 3) {
                            if (!PRUNE || flD != 0u) more(__builtin_amdgcn_readlane(dC, ri), DIR ? 3 : 0, false);
                            if (__builtin_expect(npred > 4, 0)) {   // a fifth and later ones are fetched here (direction bytes exist only while in-degrees stay <= 16: the CSR build checks)
                                const uint32_t po = __builtin_amdgcn_readlane(oC, ri);
                                for (uint32_t p = 4; p < npred; p++) {
                                    const uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.pred_rank[po + p]);
                                    if (!PRUNE || ((FM >> (ent >> 28)) & 1u) != 0u) more(ent, DIR ? (int)p : 0, false);
                                }
                            }
                        }
                    }
                }
                // what this chunk hands to the right whatever comes in from the left: its largest key (the horizontal move of de-ramped keys is a
                // plain prefix maximum, so the chunk's own recurrence can wait for the carry and run ONCE, after the scan)
                int lm = m[0];
#pragma unroll
                for (int k = 1; k < CM; k++) lm = max(lm, m[k]);
                DP_T(1);   // predecessor rows + cells
                // prefix maximum over the lanes to the left; the wait states of its DPP steps do the row's scalar chores (mailbox entry address and tag,
                // ring slot and its offset, the test word of the rare cases, the step of the nibble row pointer)
                uint32_t mb_addr, mb_tag, slot, rare, roff;
                const int inc = wave_incl_max_fill((lm & ~63) | KHC, (uint32_t)__builtin_amdgcn_readfirstlane((int)i), tag0_s, mb_lds, meta, ring_w4, dstep_s, mb_addr, mb_tag, slot, rare, roff, dlo, dhi);
                int ex = wave_shift_up1(inc, NEGK);
                DP_T(2);   // wave scan
                const int cin = __builtin_amdgcn_readlane(cinV, rj);   // NEGK without a wave on the left
                meta_nx = __builtin_amdgcn_readlane(mC, (ri + 1) & 63u);   // (the next row's record; beyond the batch: unused)
                if constexpr (!FAST) p0_nx = __builtin_amdgcn_readlane(aC, (ri + 1) & 63u);   // (inside a run of fast rows the first predecessor is known; read again behind the run)
#ifdef HX_RING_PREFETCH
                if constexpr (PF) {   // the next row's ring predecessors (within the batch): requested now, used when the row comes
                    pf_row = 0xffffffffu;
                    if (rj + 1 < nb) {
                        const uint32_t l0 = p0_nx >> 28, npn = meta_nx >> META_NP;
                        const uint32_t entN = __builtin_amdgcn_readlane(bC, (ri + 1) & 63u), l1 = npn > 1u ? entN >> 28 : 0xffu;
                        const bool r0 = l0 - 1u < 12u, r1 = l1 - 1u < 12u;
                        if (r0 | r1) {
                            const int32_t* S0 = ring_me + (size_t)(r0 ? l0 - 1u : 0u) * ring_w;
                            const int32_t* S1 = ring_me + (size_t)(r1 ? l1 - 1u : 0u) * ring_w;
#pragma unroll
                            for (int k = 0; k < CM; k++) { q0[k] = S0[k * PW + 1]; q1[k] = S1[k * PW + 1]; }
                            q0l = S0[(CM - 1) * PW]; q1l = S1[(CM - 1) * PW];
                            pf_row = i + 1; pf_s0 = r0 ? l0 : 0xffu; pf_s1 = r1 ? l1 : 0xffu;
                        }
                    }
                }
#endif
                // the carry of this row for the wave on the right: the prefix maximum through this wave's last column (lane 63 holds it)
                if (out_l != 0u) { if (lane == 63) *(volatile __attribute__((address_space(3))) unsigned long long*)(uintptr_t)mb_addr = (unsigned long long)mb_tag | ((unsigned long long)(uint32_t)max(cin, inc) << 32); }
                if (out_h != 0u) { if (lane == 63) st_dev64(mb_out_h + i, (unsigned long long)mb_tag | ((unsigned long long)(uint32_t)max(cin, inc) << 32)); }
                ex = max(ex, cin);
                DP_T(3);   // carry in / out
                // the horizontal recurrence from the finished key left of this chunk (the exclusive prefix; it carries the horizontal code, which
                // loses every tie) through the chunk: each finished key t[k] is both the next column's horizontal candidate and the row as a predecessor
                m[0] = max(m[0], ex);
                int t[CM];
                t[0] = (m[0] & ~63) | KHC;
#pragma unroll
                for (int k = 1; k < CM; k++) { m[k] = max(m[k], t[k - 1]); t[k] = (m[k] & ~63) | KHC; }
                const int left_now = ex;              // key of column j0 - 1
                if (slot != 15u) {   // a kept row goes to its ring slot (the CSR build counted the kept rows)
                    int32_t* S = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ring_me) + roff);
#pragma unroll
                    for (int k = 0; k < CM; k++) S[k * PW + 1] = t[k];
                    if (lane == 0) S[(CM - 1) * PW] = left_now;   // (first wave of the edge: "minus infinity")
                }
#pragma unroll
                for (int k = 0; k < CM; k++) tp[k] = t[k];
                lnp = left_now;
                uint32_t row_fl = 1u;
                if constexpr (PRUNE) {   // the row's flag for its successors: a lane whose last key, taken at its first column, reaches T - or a live carry-in (the column left of the wave)
                    const uint32_t fl = s_nz_u64(__builtin_amdgcn_ballot_w64(t[CM - 1] >= thr_lane)) | cin_live;   // (as a C comparison: s_cselect_b64, v_cndmask, v_readfirstlane)
                    row_fl = fl;
                    { const uint32_t mk = 0x2000u | (2u << slot); FM = (FM & ~mk) | (mk * fl); }   // (the previous row's bit and the ring slot's: both the row's flag)
                }
                DP_T(4);   // carry applied, ring copy
                if (DIR) {
                    // the move code of every cell: type * 4 + 3 - predecessor slot. The row is stored through a buffer resource of ITS bytes: chunks
                    // beyond the row (the padding lanes of the last wave) fail the range check and are dropped - no exec mask, no branch
                    uint32_t dc[CM];
#pragma unroll
                    for (int k = 0; k < CM; k++) dc[k] = (uint32_t)m[k];
                    store_nibbles_buf<CM>(__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t*>(((uintptr_t)dhi << 32) | dlo), 0, (int)(W >> 1), 0x00020000), j0 >> 1, dc);
                }
                if (!DIR && __builtin_expect(live, 1)) {
                    {
                        int pl[CM];
#pragma unroll
                        for (int k = 0; k < CM; k++) pl[k] = (t[k] + jg0 + k * g64) >> 6;
                        store_chunk_i32<CM>(hrow + j0, pl);
                        if (lane == 0 && has_in) hrow[hleft] = (left_now + jg0 - g64) >> 6;   // the wave's own copy of the column on its left
                    }
                }
                DP_T(5);   // stores
                if (__builtin_expect(rare != 0u, 0)) {   // ONE test for everything rare: wide row, far reader, sink (meta & 44)
                    if (DIR && (meta & 32u) && live) {   // wide row: type * 16 + 15 - slot, a byte per cell
                        uint32_t dc[CM];
#pragma unroll
                        for (int k = 0; k < CM; k++) dc[k] = (uint32_t)m[k];
                        store_dirs<CM>(Dwide + (uint64_t)wideslot[i - 1] * W + j0, dc, 0x3f3f3f3fu);
                    }
                    if (meta & 8u) {   // a far successor reads this row back from HBM (keys; with the score matrix it is there already)
                        // (the slot is read out of its lane HERE, where every lane is active: inside the divergent block below a register
                        // that was spilled is reloaded for the active lanes only, and lane ri need not be one of them)
                        const uint32_t fslot = DIR ? __builtin_amdgcn_readlane(fC, ri) : 0u;
                        if constexpr (PRUNE) { far_set(fslot, row_fl); if (row_fl != 0u && (FM & 0x8000u) == 0u) { FM |= 0x8000u; bad |= farref; } }
                        if (DIR && live
                            ) {
                            int32_t* F = H + (uint64_t)fslot * WH;
                            store_chunk_i32<CM>(F + j0, t);
                            if (lane == 0 && has_in) F[hleft] = left_now;
                        }
                        // (no wait: only this wave reads these words back, and a wave's memory instructions reach the cache in program order)
                    }
                    if (owns_last && (meta & 4u)) {   // sink node: candidate end of the global alignment
                        int v = NEGK;
#pragma unroll
                        for (int k = 0; k < CM; k++) if ((uint32_t)k == klast) v = t[k];
                        if (nsink < sink_cap) { sink_row[nsink] = i; sink_score[nsink] = (v >> 6) + (int)L * gap; }
                        nsink++;
                    }
                }
            };
            // Measured (round 6, 12 Mb, A/B of two builds in one GPU call): SLOWER - the longest edge's DP 284 -> 322 M cycles, the step 164 -> 180 ms. Its 303 916 rows
            // are 42 % rows with several predecessors, 30 % rows whose one predecessor sits in the ring (bubbles interleave in rank order), and only 28 % fast rows, in
            // runs of 1.3: what a run costs to set up (mask shift, count, the first predecessor's entry read again: ~100 cycles) and what every other row pays for the
            // test is more than the ~25 instructions a fast row saves. Kept behind -DHX_FAST_ROWS (off) as the measured alternative it is.
#ifdef HX_FAST_ROWS
            constexpr bool FAST_OK = DIR && !PRUNE;
#else
            constexpr bool FAST_OK = false;
#endif
            // fast <=> one predecessor (meta >> META_NP == 1) with the location code 13 (the previous row). Which rows of the batch are is read off the 64 records in
            // their lanes ONCE, as a mask (bit r = row i0 + r, nothing beyond the batch); a run of fast rows is then a counted loop - its back edge is s_sub + s_cmp +
            // one branch (as a test of the next row's record after every row it was a flag-guarded pair of branches and five scalar instructions).
            unsigned long long fastm = 0;
            if constexpr (FAST_OK) {
                fastm = __builtin_amdgcn_ballot_w64(((mC >> META_NP) << 4 | (aC >> 28)) == (1u << 4 | 13u)) >> rb;
                if (nb < 64u) fastm &= (1ull << nb) - 1ull;
            }
            while (rj < nb) {
                if constexpr (FAST_OK) {
                    // (bit nb - rj of the complement is set, so the run ends with the batch at the latest - except for a batch of 64 fast rows seen from its first
                    // row: the complement is zero there, and the count of trailing zeros of zero is not 64 but whatever the instruction leaves)
                    const unsigned long long inv = ~(fastm >> rj);
                    uint32_t run = inv ? (uint32_t)__builtin_ctzll(inv) : 64u;
                    if (__builtin_expect(run != 0u, 1)) {
                        do { row(std::true_type{}); rj++; } while (--run != 0u);
                        p0_nx = __builtin_amdgcn_readlane(aC, (rb + rj) & 63u);
                        continue;
                    }
                }
                row(std::false_type{});
                rj++;
            }
            if constexpr (PRUNE) lazy = (uint32_t)(n_dead - dead_before == nb) & lazy_on;
        }
    }
    if (owns_last) nSinkOut = nsink;
    if constexpr (PRUNE) { if (lane == 0 && pstat) { atomicAdd(&pstat[0], (unsigned long long)V); atomicAdd(&pstat[1], (unsigned long long)n_dead); atomicAdd(&pstat[8], (unsigned long long)n_bulk); } }
}

