// Planning step of the deterministic row-sparse embedding backward, ONE-KERNEL form for near-unique id batches over mid-size tables
// (C2: 26 tables x 1 M rows, 65 536 uniform ids each).  Reference behaviour being replaced: autograd of nn.Embedding over every lookup
// feature (src/model/BaseModel/base_model.py:262-308) -- same role as nrx_sparse_plan_place (nrx_sparse.hip), different method.
//
// Why not a sort: on uniform ids 97 % of the rows of such a launch are looked up ONCE -- their gradient row is the lookup's upstream row,
// no order needed -- and 3 % twice (a + b == b + a: no order needed either).  Only the rows looked up three times or more (and the padding
// row) need their lookups in a fixed order.  What every lookup needs is its row's UNIQUE INDEX, and that is a rank in a bitmap.
// Why not a global bitmap: measured first (tools/bitmap_plan_probe.hip, profiles/r05_bitmap_plan_probe.txt) -- every scattered L2 access,
// atomic or plain, is one L2 request, and the chip does ~22 G returning atomics/s: mark (78 us) + emit (36 us) lose to two coalesced radix
// passes.  So the bitmaps live in LDS and nothing is scattered:
//   a block owns a 131 072-row RANGE of one table -- three 16 KB bitmaps (looked up / twice / three times or more: a saturating per-row
//   count from cascaded LDS atomics; no result depends on the order of arrival) -- and SCANS every lookup of its table (coalesced 16-byte
//   loads; the ranges of a table read the same ids from L2), keeping the lookups of its range in an LDS cache;
//   popcount prefix of `looked up` = the rank of every row inside the block; the blocks' totals are chained in table-major order (a block
//   takes its number from a ticket, publishes its totals, and sums those of the earlier tickets -- which have all started): unique index
//   of a row = rows before the block + rank, ascending by (table, row) as the sorted plan has it;
//   per cached lookup: dest[p] = u (once); -1 and the block's list (the other rows);
//   the list is sorted by (row, lookup) in LDS (in place in global memory beyond 4096 entries): the two lookups of a row looked up twice end up
//   side by side -> one PAIR record {u, first, second}; the rows looked up 3+ times (and the padding row) -> order / seg_start / walk;
//   the unique keys leave in row order through LDS as whole lines.
// C2: 43 us in one launch against 74-80 us in six (profiles/r05_plan_lds.txt).  A skewed batch (many rows looked up 3+ times) is correct but
// slow here (one block sorts its range's whole list): the caller chooses the planner from the previous batch's duplicate statistics
// (stats[]; ops.py) -- the radix planner stays the default for anything not near-unique.
#include "nrx_common.h"
#include <cstring>
#include <cstdlib>

namespace {

constexpr int PL_THREADS = 1024;
constexpr int PL_RLOG = 17;                         // rows per block
constexpr int PL_WORDS = 1 << (PL_RLOG - 5);        // 4096 words per bitmap
constexpr int PL_CACHE = 9728;                      // lookups of its range a block keeps in LDS (beyond: the later phases re-scan the ids)
constexpr int PL_PAIRS = 2048;                      // pair rows a block matches through LDS slots (beyond: through the sorted list)
constexpr int PL_SORT = 4096;                       // 3+ lookups a block sorts in LDS (beyond: in place in global memory)
constexpr int PL_MAX_BLOCKS = 4096;
constexpr size_t PL_LDS = (size_t)4 * PL_WORDS * 4 + (size_t)PL_WORDS * 2 + (size_t)PL_PAIRS * 4 + (size_t)(PL_CACHE + 64) * 8;      // (+ the spare entries)
constexpr int PL_AGG = 4;                            // totals a block publishes
constexpr size_t PL_STATE_BYTES = 256 + (size_t)PL_MAX_BLOCKS * PL_AGG * 8;

struct PlanLdsArgs {
    const void* ids[NRX_MAX_FEATURES];          // per feature
    int64_t off[NRX_MAX_FEATURES + 1];          // flat lookup offset of feature f
    int64_t rows[NRX_MAX_FEATURES];             // per table
    int32_t feat_first[NRX_MAX_FEATURES + 1];   // table t's features: feat_of[feat_first[t] .. feat_first[t + 1])
    int32_t feat_of[NRX_MAX_FEATURES];
    int32_t blk_first[NRX_MAX_FEATURES + 1];    // table t's work items (row ranges)
    int32_t n_tables, n_blocks;
    int32_t xcd_order, pad_;
    int64_t batch, n_total;
    uint32_t* ctl;                              // [0] ticket  [1] blocks done  [2] epoch  (zero before the first call; the kernel re-arms it)
    unsigned long long* agg;                    // [n_blocks][4]: (epoch + 1) << 32 | {unique rows, walk rows, their lookups, pair rows} of the block
    int64_t *uniq_keys, *counts, *order, *seg_start, *n_walk, *n_pairs, *stats;
    int32_t *dest, *walk, *pairs;
    unsigned long long* list3;
};
static_assert(sizeof(PlanLdsArgs) <= 3584, "kernarg budget");

#ifdef NRX_PL_TIMING            // dev builds (tools/plan_lds_phases.py): wall-clock stamps of every block at the phase boundaries
__device__ unsigned long long pl_stamps[PL_MAX_BLOCKS * 8];
#define PL_T(k) do { if (threadIdx.x == 0) pl_stamps[8 * w + (k)] = wall_clock64(); } while (0)
#else
#define PL_T(k) do { } while (0)
#endif

template <bool IDX64>
__device__ __forceinline__ uint64_t pl_id(const void* p, int64_t i) {
    return IDX64 ? (uint64_t)nrx_gconst<int64_t>(p)[i] : (uint64_t)(int64_t)nrx_gconst<int32_t>(p)[i];
}

// every lookup of the block's row range: from the LDS cache, or -- a range that took more lookups than the cache holds -- by scanning again
template <bool IDX64, typename F>
__device__ __forceinline__ void pl_for_each(const NRX_CONST PlanLdsArgs* a, int t, uint32_t rb, bool cached, uint32_t ncache,
                                            const unsigned long long* s_cache, F fn) {
    if (cached) {
        for (uint32_t i = threadIdx.x; i < ncache; i += PL_THREADS) {
            const unsigned long long e = s_cache[i];
            fn((uint32_t)(e >> 32), (uint32_t)e);
        }
        return;
    }
    const uint64_t rows = (uint64_t)a->rows[t];
    for (int fi = a->feat_first[t]; fi < a->feat_first[t + 1]; ++fi) {
        const int f = a->feat_of[fi];
        const void* idp = a->ids[f];
        const int64_t po = a->off[f];
        for (int64_t b = threadIdx.x; b < a->batch; b += PL_THREADS) {
            uint64_t r = pl_id<IDX64>(idp, b);
            r = r >= rows ? 0ull : r;
            if ((uint32_t)(r >> PL_RLOG) == rb) fn((uint32_t)r & ((1u << PL_RLOG) - 1), (uint32_t)(po + b));
        }
    }
}

// IDX64: id width.  VEC: the id arrays are 16-byte aligned (16-byte loads: 2 / 4 ids); else one id per load.
template <bool IDX64, bool VEC>
__global__ __launch_bounds__(PL_THREADS) void plan_lds_kernel(const PlanLdsArgs args_in_kernarg) {
    const NRX_CONST PlanLdsArgs* a = nrx_kernarg<PlanLdsArgs>();
    extern __shared__ __attribute__((aligned(16))) uint32_t pl_smem[];
    uint32_t* s_t = pl_smem;                       // looked up
    uint32_t* s_m = s_t + PL_WORDS;                // ... at least twice
    uint32_t* s_h = s_m + PL_WORDS;                // ... at least three times
    uint32_t* s_pre = s_h + PL_WORDS;              // rows of the range before each word
    uint16_t* s_pre2 = reinterpret_cast<uint16_t*>(s_pre + PL_WORDS);      // pair rows (looked up exactly twice) of the range before each word
    uint32_t* s_slot = reinterpret_cast<uint32_t*>(s_pre2 + PL_WORDS);     // per pair row of the range: the lookup that came first
    unsigned long long* s_cache = reinterpret_cast<unsigned long long*>(s_slot + PL_PAIRS);      // (row in range << 32 | flat lookup)
    __shared__ uint32_t s_ticket, s_epoch, s_ncache, s_nlist;
    __shared__ unsigned long long s_red[4][PL_THREADS / 64];
    __shared__ uint32_t s_wsum[PL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        // Which row range this block takes.  XCD order (launches of at most one block per compute unit: every block is resident, the chain below
        // cannot wait for a block that has not started): block b runs on XCD b % 8, and XCD x takes the x-th eighth of the table-major list -- the
        // ranges of a table sit on ONE XCD and its ids are fetched into ONE L2 (in ticket order the eight ranges of a table ran on eight XCDs: 8 x
        // 13.6 MB of fabric reads for C2's ids, the scan ran at the fabric's rate).  Larger launches: ticket order (always safe).
        if (a->xcd_order) {
            const unsigned b = blockIdx.x, x = b & 7u, i = b >> 3, qt = (unsigned)a->n_blocks >> 3, rt = (unsigned)a->n_blocks & 7u;
            s_ticket = x * qt + (x < rt ? x : rt) + i;
        } else {
            s_ticket = atomicAdd(&a->ctl[0], 1u);
        }
        s_epoch = __hip_atomic_load(&a->ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        s_ncache = 0;
        s_nlist = 0;
    }
    for (int i = tid; i < 3 * PL_WORDS; i += PL_THREADS) s_t[i] = 0;
    __syncthreads();
    const int w = (int)s_ticket;
    const uint32_t mark = s_epoch;
    int t = 0;
    for (int i = 1; i < a->n_tables; ++i) t += w >= a->blk_first[i] ? 1 : 0;
    const uint32_t rb = (uint32_t)(w - a->blk_first[t]);
    const uint64_t rows = (uint64_t)a->rows[t];
    PL_T(0);
    constexpr uint32_t RMASK = (1u << PL_RLOG) - 1;
    // ---- scan the table's lookups: keep those of this range.  A round = K ids per thread taken together: range tests, K unconditional LDS writes; the next round's ids are requested before this
    // round is looked at (a block is 16 wavefronts on a compute unit of its own: nothing else hides the latency).
    constexpr int PER = VEC ? (IDX64 ? 2 : 4) : 1;     // ids per load
    constexpr int U = VEC ? (IDX64 ? 4 : 2) : 8;       // loads per thread and round
    constexpr int K = PER * U;
    typedef long long pl_ll2 __attribute__((ext_vector_type(2)));
    typedef int pl_i4 __attribute__((ext_vector_type(4)));
    for (int fi = a->feat_first[t]; fi < a->feat_first[t + 1]; ++fi) {
        const int f = a->feat_of[fi];
        const void* idp = a->ids[f];
        const int64_t po = a->off[f];
        const int64_t nv = a->batch / PER;                // whole loads
        uint64_t bufa[K], bufb[K];
        auto fetch = [&](int64_t v0, uint64_t* dst) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int64_t v = v0 + (int64_t)j * PL_THREADS + tid;
                if (VEC && IDX64) {
                    pl_ll2 x = {-1, -1};
                    if (v < nv) x = nrx_gconst<pl_ll2>(idp)[v];          // (plain loads: the other ranges of the table read the same lines from L2)
                    dst[2 * j] = (uint64_t)x.x; dst[2 * j + 1] = (uint64_t)x.y;
                } else if (VEC) {
                    pl_i4 x = {-1, -1, -1, -1};
                    if (v < nv) x = nrx_gconst<pl_i4>(idp)[v];
                    dst[4 * j] = (uint64_t)(int64_t)x.x; dst[4 * j + 1] = (uint64_t)(int64_t)x.y;
                    dst[4 * j + 2] = (uint64_t)(int64_t)x.z; dst[4 * j + 3] = (uint64_t)(int64_t)x.w;
                } else {
                    dst[j] = v < nv ? pl_id<IDX64>(idp, v) : ~0ull;
                }
            }
        };
        auto process = [&](const uint64_t* cur, int64_t v0) {
            // cache places without a cross-lane scan: per id one ballot (a scalar), the lane's rank among the set lanes (mbcnt), the running
            // total in a scalar register; ONE LDS atomic per wavefront and round reserves the places (six dependent cross-lane steps per round
            // were ~700 cycles of latency with four wavefronts per SIMD to hide them)
            uint32_t lr[K], rank[K];
            bool in[K];
            uint32_t tot = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int64_t v = v0 + (int64_t)(k / PER) * PL_THREADS + tid;
                uint64_t ur = cur[k];
                ur = ur >= rows ? 0ull : ur;                   // (negative ids are huge here): out-of-range ids fall on the padding row
                in[k] = v < nv && (uint32_t)(ur >> PL_RLOG) == rb;
                lr[k] = (uint32_t)ur & RMASK;
                const unsigned long long bal = __ballot(in[k]);
                rank[k] = tot + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                tot += (uint32_t)__popcll(bal);
            }
            uint32_t base = 0;
            if (lane == 0 && tot != 0) base = atomicAdd(&s_ncache, tot);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            // no branch per id: a lookup of another range (7 of 8) writes its lane's spare entry behind the cache
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int64_t v = v0 + (int64_t)(k / PER) * PL_THREADS + tid;
                const uint32_t pos = base + rank[k];
                const uint32_t at = in[k] && pos < PL_CACHE ? pos : (uint32_t)PL_CACHE + lane;      // (a spare entry per lane: 64 lanes on one address are 64 conflicts)
                s_cache[at] = ((unsigned long long)lr[k] << 32) | (unsigned long long)(uint32_t)(po + v * PER + (k % PER));
            }
        };
        // two buffers, the loop body written twice: a round's ids are requested a full round ahead, and nothing is moved between registers
        const int64_t step = (int64_t)U * PL_THREADS;
        fetch(0, bufa);
        for (int64_t v0 = 0; v0 < nv; v0 += 2 * step) {
            fetch(v0 + step, bufb);
            process(bufa, v0);
            if (v0 + step >= nv) break;
            fetch(v0 + 2 * step, bufa);
            process(bufb, v0 + step);
        }
        for (int64_t b = nv * PER + tid; b < a->batch; b += PL_THREADS) {           // the last ids past the whole loads
            uint64_t ur = pl_id<IDX64>(idp, b);
            ur = ur >= rows ? 0ull : ur;
            if ((uint32_t)(ur >> PL_RLOG) == rb) {
                const uint32_t l = (uint32_t)ur & RMASK;
                const uint32_t pos = atomicAdd(&s_ncache, 1u);
                if (pos < PL_CACHE) s_cache[pos] = ((unsigned long long)l << 32) | (unsigned long long)(uint32_t)(po + b);
            }
        }
    }
    __syncthreads();
    PL_T(1);
    const uint32_t ncache = s_ncache;
    const bool cached = ncache <= PL_CACHE;
    // ---- mark: every lookup of the range ORs its row's bit into `looked up`; one that finds it set ORs `twice`; one that finds that set ORs `3+`
    // (from the cache: dense wavefronts -- inside the scan the same atomics ran with one lane in eight active, eight times per round)
    pl_for_each<IDX64>(a, t, rb, cached, ncache, s_cache, [&](uint32_t lr, uint32_t) {
        const uint32_t bit = 1u << (lr & 31);
        const uint32_t o = atomicOr(&s_t[lr >> 5], bit);
        if (o & bit) {
            const uint32_t o2 = atomicOr(&s_m[lr >> 5], bit);
            if (o2 & bit) atomicOr(&s_h[lr >> 5], bit);
        }
    });
    __syncthreads();
    if (tid == 0 && rb == 0 && (s_t[0] & 1u)) {              // the padding row never trains and is never placed: it goes to the list
        s_m[0] |= 1u;
        s_h[0] |= 1u;
    }
    __syncthreads();
    PL_T(2);
    // ---- ranks inside the block, the block's totals: unique rows, walk rows (looked up 3+ times, padding), their lookups, pair rows (twice)
    uint32_t tw[4], pw[4], ct = 0, c3 = 0, c2 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        tw[j] = s_t[4 * tid + j];
        const uint32_t hw = s_h[4 * tid + j];
        pw[j] = s_m[4 * tid + j] & ~hw;
        ct += __popc(tw[j]);
        c3 += __popc(hw);
        c2 += __popc(pw[j]);
    }
    uint32_t c3l = 0;
    pl_for_each<IDX64>(a, t, rb, cached, ncache, s_cache, [&](uint32_t lr, uint32_t) { c3l += (s_h[lr >> 5] >> (lr & 31)) & 1u; });
    // both running counts in one word: unique rows in the low 18 bits (<= 2^17 rows per range), pair rows above (the field wraps beyond 2^14 pair
    // rows -- harmlessly for the low field; the pair ranks are only used while the block has <= PL_PAIRS of them, T_2 is summed separately)
    const uint32_t mine = ct | (c2 << 18);
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c3 += __shfl_xor(c3, off, 64);
        c3l += __shfl_xor(c3l, off, 64);
        c2 += __shfl_xor(c2, off, 64);
    }
    if (lane == 63) s_wsum[wid] = incl;
    if (lane == 0) {
        s_red[1][wid] = c3;
        s_red[2][wid] = c3l;
        s_red[3][wid] = c2;
    }
    __syncthreads();
    uint32_t wbase = 0, wtot = 0, T_3r = 0, T_3l = 0, T_2 = 0;
#pragma unroll
    for (int i = 0; i < PL_THREADS / 64; ++i) {
        if (i < wid) wbase += s_wsum[i];
        wtot += s_wsum[i];
        T_3r += (uint32_t)s_red[1][i];
        T_3l += (uint32_t)s_red[2][i];
        T_2 += (uint32_t)s_red[3][i];
    }
    const uint32_t T_u = wtot & 0x3ffffu;
    const bool pair_slots = T_2 <= PL_PAIRS;          // the two lookups of a pair row meet in an LDS slot; else: side by side in the sorted list
    {
        const uint32_t excl = wbase + incl - mine;
        uint32_t run = excl & 0x3ffffu, run2 = excl >> 18;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_pre[4 * tid + j] = run;
            s_pre2[4 * tid + j] = (uint16_t)run2;
            run += __popc(tw[j]);
            run2 += __popc(pw[j]);
        }
    }
    for (int i = tid; i < PL_PAIRS; i += PL_THREADS) s_slot[i] = 0xffffffffu;
    PL_T(3);
    if (tid < PL_AGG) {
        const uint32_t v = tid == 0 ? T_u : (tid == 1 ? T_3r : (tid == 2 ? T_3l : T_2));
        __hip_atomic_store(&a->agg[PL_AGG * w + tid], ((unsigned long long)mark << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the blocks before this one in table-major order = the earlier tickets (all of them running or done): their totals
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int j = tid; j < PL_AGG * w; j += PL_THREADS) {
        unsigned long long x;
        do { x = __hip_atomic_load(&a->agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(x >> 32) != mark);
        const uint32_t v = (uint32_t)x;
        const int k = j & (PL_AGG - 1);
        s0 += k == 0 ? v : 0; s1 += k == 1 ? v : 0; s2 += k == 2 ? v : 0; s3 += k == 3 ? v : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); s3 += __shfl_xor(s3, off, 64);
    }
    __syncthreads();
    if (lane == 0) { s_red[0][wid] = s0; s_red[1][wid] = s1; s_red[2][wid] = s2; s_red[3][wid] = s3; }
    __syncthreads();
    uint32_t base_u = 0, base_w3 = 0, base_o3 = 0, base_p = 0;
#pragma unroll
    for (int i = 0; i < PL_THREADS / 64; ++i) {
        base_u += (uint32_t)s_red[0][i]; base_w3 += (uint32_t)s_red[1][i]; base_o3 += (uint32_t)s_red[2][i]; base_p += (uint32_t)s_red[3][i];
    }
    PL_T(4);
    // ---- per lookup of the range: its place.  A lookup of a row looked up once is placed (dest = unique index); the others go to the block's
    // list as (unique index << 33 | walk row ? 1 : 0) << 32 | lookup) -- sorted below, the two lookups of a pair row end up side by side
    const uint32_t m = T_3l + (pair_slots ? 0u : 2 * T_2);      // the block's list
    const bool sort_lds = m <= PL_SORT;
    NRX_GLOBAL unsigned long long* g3 = nrx_gmut<unsigned long long>(a->list3) + base_o3 + 2 * (size_t)base_p;
    NRX_GLOBAL int32_t* dest = nrx_gmut<int32_t>(a->dest);
    typedef int nrx_i32x4p __attribute__((ext_vector_type(4)));
    NRX_GLOBAL nrx_i32x4p* pair_out = nrx_gmut<nrx_i32x4p>(a->pairs);
    const int64_t key_hi = ((int64_t)t << 40) | ((int64_t)rb << PL_RLOG);
    pl_for_each<IDX64>(a, t, rb, cached, ncache, s_cache, [&](uint32_t lr, uint32_t p) {
        const uint32_t wd = lr >> 5, bit = 1u << (lr & 31);
        const uint32_t u = base_u + s_pre[wd] + __popc(s_t[wd] & (bit - 1));
        const bool mm = (s_m[wd] & bit) != 0, h = (s_h[wd] & bit) != 0;
        dest[p] = mm ? -1 : (int32_t)u;
        if (mm && !h && pair_slots) {                   // the second lookup to arrive at the row's slot writes the record: {u, first, second}
            const uint32_t r2 = (uint32_t)s_pre2[wd] + __popc(s_m[wd] & ~s_h[wd] & (bit - 1));
            const uint32_t o = atomicExch(&s_slot[r2], p);
            if (o != 0xffffffffu) {
                nrx_i32x4p rec;
                rec.x = (int32_t)u; rec.y = (int32_t)(o < p ? o : p); rec.z = (int32_t)(o < p ? p : o); rec.w = 0;
                pair_out[base_p + r2] = rec;
            }
        } else if (mm) {
            g3[atomicAdd(&s_nlist, 1u)] = ((unsigned long long)u << 33) | ((unsigned long long)(h ? 1u : 0u) << 32) | p;
        }
    });
    if (tid == 0) {
        if (rb == 0) a->counts[1 + t] = base_u;
        if (w == a->n_blocks - 1) {
            a->counts[0] = base_u + T_u;
            a->counts[1 + a->n_tables] = base_u + T_u;
            a->n_walk[0] = base_w3 + T_3r;
            a->n_pairs[0] = base_p + T_2;
            if (a->stats != nullptr) {
                a->stats[0] = base_u + T_u;         // unique rows
                a->stats[1] = base_w3 + T_3r;       // rows on the walk list (3+ lookups, padding rows)
                a->stats[2] = base_o3 + T_3l;       // their lookups
                a->stats[3] = a->n_total;
            }
        }
    }
    __syncthreads();
    PL_T(5);
    // ---- the block's list in (row, lookup) order -> pair records; order / seg_start / walk for the walk rows
    if (m != 0) {
        unsigned long long* s_keys = s_cache;
        uint32_t N = 1;
        while (N < m) N <<= 1;
        if (sort_lds) {
            for (uint32_t i = tid; i < N; i += PL_THREADS) s_keys[i] = i < m ? g3[i] : ~0ull;
            __syncthreads();
            if (m <= PL_THREADS) {           // a key's place = the number of smaller keys (all distinct)
                const unsigned long long k = (uint32_t)tid < m ? s_keys[tid] : ~0ull;
                uint32_t r = 0;
                for (uint32_t i = 0; i < m; ++i) r += s_keys[i] < k ? 1u : 0u;
                __syncthreads();
                if ((uint32_t)tid < m) s_keys[r] = k;
                __syncthreads();
            } else {
                // normalised bitonic network: every comparator leaves the smaller key at the lower index (first step of a stage: the mirror partner)
                for (uint32_t k = 2; k <= N; k <<= 1)
                    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                        for (uint32_t i = tid; i < N; i += PL_THREADS) {
                            const uint32_t x = j == (k >> 1) ? i ^ (k - 1) : i ^ j;
                            if (x > i) {
                                const unsigned long long ki = s_keys[i], kx = s_keys[x];
                                if (ki > kx) { s_keys[i] = kx; s_keys[x] = ki; }
                            }
                        }
                        __syncthreads();
                    }
            }
        } else {                             // in place in global memory; entries past m count as +inf and are never stored
            for (uint32_t k = 2; k <= N; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = tid; i < N; i += PL_THREADS) {
                        const uint32_t x = j == (k >> 1) ? i ^ (k - 1) : i ^ j;
                        if (x > i && x < m) {
                            const unsigned long long ki = g3[i], kx = g3[x];
                            if (ki > kx) { g3[i] = kx; g3[x] = ki; }
                        }
                    }
                    __syncthreads();
                }
        }
        // the sorted list, 1024 entries at a time: an entry of a walk row goes to `order` (packed: the pair rows' entries do not count), the
        // first one of a row also names it in `walk` and starts its segment, the last one closes it; the first entry of a pair row writes
        // the pair's record {unique index, first lookup, second lookup}
        uint32_t run_e = 0, run_h = 0, run_p = 0;
        NRX_GLOBAL int32_t* pairs = nrx_gmut<int32_t>(a->pairs);
        for (uint32_t i0 = 0; i0 < m; i0 += PL_THREADS) {
            const uint32_t i = i0 + tid;
            const bool live = i < m;
            unsigned long long k = 0, kp = ~0ull, kn = ~0ull;
            if (live) {
                k = sort_lds ? s_keys[i] : g3[i];
                if (i > 0) kp = sort_lds ? s_keys[i - 1] : g3[i - 1];
                if (i + 1 < m) kn = sort_lds ? s_keys[i + 1] : g3[i + 1];
            }
            const uint32_t u = (uint32_t)(k >> 33);
            const bool isw = live && ((k >> 32) & 1ull) != 0;
            const bool first = live && (uint32_t)(kp >> 33) != u, last = live && (uint32_t)(kn >> 33) != u;      // (the sentinels' index is 2^31 - 1: no row's)
            const bool head = isw && first, tail = isw && last, phead = live && !isw && first;
            const unsigned long long be = __ballot(isw), bh = __ballot(head), bp = __ballot(phead);
            if (lane == 0) {
                s_wsum[wid] = (uint32_t)__popcll(be);
                s_red[0][wid] = (uint32_t)__popcll(bh);
                s_red[1][wid] = (uint32_t)__popcll(bp);
            }
            __syncthreads();
            uint32_t e_before = run_e, h_before = run_h, p_before = run_p, e_tot = 0, h_tot = 0, p_tot = 0;
            for (int ww = 0; ww < PL_THREADS / 64; ++ww) {
                if (ww < wid) { e_before += s_wsum[ww]; h_before += (uint32_t)s_red[0][ww]; p_before += (uint32_t)s_red[1][ww]; }
                e_tot += s_wsum[ww]; h_tot += (uint32_t)s_red[0][ww]; p_tot += (uint32_t)s_red[1][ww];
            }
            const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
            const uint32_t e_at = base_o3 + e_before + (uint32_t)__popcll(be & lt);
            if (isw) a->order[e_at] = (int64_t)(uint32_t)k;
            if (head) {
                a->walk[base_w3 + h_before + (uint32_t)__popcll(bh & lt)] = (int32_t)u;
                a->seg_start[u] = (int64_t)e_at;
            }
            if (tail) a->seg_start[u + 1] = (int64_t)e_at + 1;
            if (phead) {
                const size_t r = (size_t)(base_p + p_before + (uint32_t)__popcll(bp & lt)) * 4;
                pairs[r] = (int32_t)u; pairs[r + 1] = (int32_t)(uint32_t)k; pairs[r + 2] = (int32_t)(uint32_t)kn; pairs[r + 3] = 0;
            }
            run_e += e_tot; run_h += h_tot; run_p += p_tot;
            __syncthreads();
        }
    }
    PL_T(6);
    // ---- the unique keys of the range in row order (= unique-index order), staged through LDS and written as whole lines
    {
        __syncthreads();
        unsigned long long* s_stage = s_cache;
        constexpr uint32_t CH = PL_CACHE;
        const uint32_t pre = s_pre[4 * tid];
        for (uint32_t c0 = 0; c0 < T_u; c0 += CH) {
            uint32_t run = pre;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t bits = tw[j];
                while (bits) {
                    const int bp = __ffs(bits) - 1;
                    bits &= bits - 1;
                    if (run >= c0 && run < c0 + CH) s_stage[run - c0] = (unsigned long long)(key_hi | (int64_t)((4 * tid + j) * 32 + bp));
                    ++run;
                }
            }
            __syncthreads();
            const uint32_t cnt = T_u - c0 < CH ? T_u - c0 : CH;
            for (uint32_t i = tid; i < cnt; i += PL_THREADS) a->uniq_keys[base_u + c0 + i] = (int64_t)s_stage[i];
            __syncthreads();
        }
    }
    PL_T(7);
    // ---- the LAST block of the chain re-arms the state: it has read the published totals of every other block, so every block has read the epoch
    // (a block publishes with it) and taken its ticket.  (Round 5 counted the finished blocks in ctl[1]: n_blocks device-scope atomics on ONE address,
    // which the memory side serialises at ~40 ns each -- blocks that finish together queued for up to 8 us of the launch's tail.)
    if (tid == 0 && w == a->n_blocks - 1) {
        a->ctl[0] = 0;
        __hip_atomic_store(&a->ctl[2], mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void plan_stats_kernel(const int64_t* __restrict__ counts, const int64_t* __restrict__ n_walk, int64_t n, int64_t* __restrict__ stats) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stats[0] = counts[0];
        stats[1] = n_walk != nullptr ? n_walk[0] : -1;
        stats[2] = -1;                  // the sorted planners do not count the lookups of the walk rows
        stats[3] = n;
    }
}

}  // namespace

// The duplicate statistics of a nrx_sparse_plan_place plan in nrx_sparse_plan_lds's stats format ({unique rows, walk rows, -1, n}), so that a caller
// that alternates between the planners by the previous batch's statistics has them from either.  stats may be mapped host memory.
extern "C" int nrx_sparse_plan_stats(const int64_t* counts, const int64_t* n_walk, int64_t n_lookups, int64_t* stats, void* stream) {
    NRX_REQUIRE(counts != nullptr && stats != nullptr, "nrx_sparse_plan_stats: null buffer");
    hipLaunchKernelGGL(plan_stats_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), counts, n_walk, n_lookups, stats);
    NRX_LAUNCH_CHECK("nrx_sparse_plan_stats");
    return NRX_OK;
}

extern "C" int64_t nrx_sparse_plan_lds_state_bytes(void) { return (int64_t)PL_STATE_BYTES; }

extern "C" int64_t nrx_sparse_plan_lds_workspace(int64_t n_lookups) {
    if (n_lookups < 0 || n_lookups >= 0x7fffffffLL) return -1;
    return (n_lookups + 8) * 8 + 256;
}

// the launch shape, or why not: > 0 = blocks
static int plan_lds_shape(const int64_t* lens, const int32_t* table_of, const int64_t* rows, int32_t n_feats, int32_t n_tables,
                          int64_t* rows_t /* [n_tables] */, int64_t* look_t /* [n_tables] */) {
    if (n_feats < 1 || n_feats > NRX_MAX_FEATURES || n_tables < 1 || n_tables > NRX_MAX_FEATURES) return 0;
    const int64_t batch = lens[0];
    if (batch < 1) return 0;
    for (int t = 0; t < n_tables; ++t) rows_t[t] = look_t[t] = 0;
    int64_t n = 0;
    for (int f = 0; f < n_feats; ++f) {
        if (lens[f] != batch || table_of[f] < 0 || table_of[f] >= n_tables || rows[f] < 1) return 0;      // single-valued features only
        if (rows_t[table_of[f]] != 0 && rows_t[table_of[f]] != rows[f]) return 0;
        rows_t[table_of[f]] = rows[f];
        look_t[table_of[f]] += batch;
        n += batch;
    }
    if (n >= 0x7fffffffLL) return 0;
    int64_t blocks = 0, visits = 0;
    for (int t = 0; t < n_tables; ++t) {
        if (rows_t[t] >= (1ll << 31)) return 0;
        const int64_t nb = rows_t[t] == 0 ? 1 : (rows_t[t] + (1ll << PL_RLOG) - 1) >> PL_RLOG;      // (a table nobody reads: one idle block keeps the numbering simple)
        blocks += nb;
        visits += nb * look_t[t];
    }
    // every range of a table scans all of the table's lookups: worth it while the redundancy stays near C2's 8x
    if (blocks > PL_MAX_BLOCKS || visits > 16 * n) return 0;
    return (int)blocks;
}

extern "C" int nrx_sparse_plan_lds_ok(const int64_t* lens, const int32_t* table_of, const int64_t* rows, int32_t n_feats, int32_t n_tables) {
    if (lens == nullptr || table_of == nullptr || rows == nullptr) return 0;
    int64_t rows_t[NRX_MAX_FEATURES], look_t[NRX_MAX_FEATURES];
    return plan_lds_shape(lens, table_of, rows, n_feats, n_tables, rows_t, look_t) > 0 ? 1 : 0;
}

extern "C" int nrx_sparse_plan_lds(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                                   int32_t n_feats, int32_t index_bits, int32_t n_tables, int64_t* order, int64_t* uniq_keys,
                                   int64_t* seg_start, int64_t* counts, int32_t* dest, int32_t* walk, int64_t* n_walk, int32_t* pairs,
                                   int64_t* n_pairs, int64_t* stats, void* state, void* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(ids && lens && table_of && rows, "nrx_sparse_plan_lds: null argument array");
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_sparse_plan_lds: index_bits must be 32 or 64");
    NRX_REQUIRE(order && uniq_keys && seg_start && counts && dest && walk && n_walk && pairs && n_pairs && state && workspace, "nrx_sparse_plan_lds: null buffer");
    NRX_REQUIRE(nrx_aligned16(pairs), "nrx_sparse_plan_lds: pairs must be 16-byte aligned");
    int64_t rows_t[NRX_MAX_FEATURES], look_t[NRX_MAX_FEATURES];
    const int blocks = plan_lds_shape(lens, table_of, rows, n_feats, n_tables, rows_t, look_t);
    if (blocks <= 0) {
        nrx_set_error("nrx_sparse_plan_lds: launch outside the one-kernel planner's shapes (single-valued features of one length, <= %d row ranges, "
                      "redundancy <= 16): use nrx_sparse_plan_place", PL_MAX_BLOCKS);
        return NRX_ERR_UNSUPPORTED;
    }
    PlanLdsArgs a;
    memset(&a, 0, sizeof(a));
    bool vec = true;
    int64_t off = 0;
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(ids[f] != nullptr, "nrx_sparse_plan_lds: feature %d: null ids", f);
        a.ids[f] = ids[f];
        a.off[f] = off;
        off += lens[f];
        vec = vec && nrx_aligned16(ids[f]);
    }
    a.off[n_feats] = off;
    int nb = 0, q = 0;
    for (int t = 0; t < n_tables; ++t) {
        a.rows[t] = rows_t[t] > 0 ? rows_t[t] : 1;
        a.feat_first[t] = q;
        for (int f = 0; f < n_feats; ++f)
            if (table_of[f] == t) a.feat_of[q++] = f;
        a.blk_first[t] = nb;
        nb += rows_t[t] == 0 ? 1 : (int)((rows_t[t] + (1ll << PL_RLOG) - 1) >> PL_RLOG);
    }
    a.feat_first[n_tables] = q;
    a.blk_first[n_tables] = nb;
    a.n_tables = n_tables;
    a.n_blocks = nb;
    {
        static int cus = 0;
        if (cus == 0) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
            else cus = 1;
        }
        const char* e = getenv("NRX_PLAN_LDS_XCD");
        a.xcd_order = nb <= cus && !(e && e[0] == '0');
    }
    a.batch = lens[0];
    a.n_total = off;
    a.ctl = reinterpret_cast<uint32_t*>(state);
    a.agg = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(state) + 256);
    a.uniq_keys = uniq_keys; a.counts = counts; a.order = order; a.seg_start = seg_start; a.n_walk = n_walk; a.stats = stats;
    a.dest = dest; a.walk = walk; a.pairs = pairs; a.n_pairs = n_pairs;
    a.list3 = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_lds_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_lds_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_lds_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&plan_lds_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL_LDS);
        attr_done = true;
    }
    if (index_bits == 64) {
        if (vec) hipLaunchKernelGGL((plan_lds_kernel<true, true>), dim3(nb), dim3(PL_THREADS), PL_LDS, st, a);
        else hipLaunchKernelGGL((plan_lds_kernel<true, false>), dim3(nb), dim3(PL_THREADS), PL_LDS, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((plan_lds_kernel<false, true>), dim3(nb), dim3(PL_THREADS), PL_LDS, st, a);
        else hipLaunchKernelGGL((plan_lds_kernel<false, false>), dim3(nb), dim3(PL_THREADS), PL_LDS, st, a);
    }
    NRX_LAUNCH_CHECK("nrx_sparse_plan_lds");
    return NRX_OK;
}

#ifdef NRX_PL_TIMING
extern "C" __attribute__((visibility("default"))) int nrx_plan_lds_stamps(unsigned long long* out, int n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pl_stamps), (size_t)n_blocks * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
