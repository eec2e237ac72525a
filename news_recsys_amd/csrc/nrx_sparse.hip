// Planning step of the deterministic row-sparse embedding backward: group all lookups of a launch by
// (table, row).  Reference behaviour being replaced: autograd of nn.Embedding over every lookup feature
// (src/model/BaseModel/base_model.py:262-308), here producing nn.Embedding(sparse=True)-style COO grads.
//
// keys = (table << row_bits) | row are built COMPACT -- only table_bits + row_bits significant bits, 32-bit
// when they fit (C2: 5 + 20 bits) -- with payload = the 32-bit flat lookup index, and sorted STABLY, so the lookups of one
// row stay in (feature, sample) order and the segmented reduction that follows (nrx_embed_bwd_sorted) is bit-reproducible.
// The sort is the table-segmented LSD radix sort below (the table of a lookup is known from its feature, so the pairs are
// laid out table-major by the key kernel and only the row bits are sorted, inside each table's segment: C2 plan 96 us vs
// 171 us with the general sort); rocPRIM's radix sort limited to the significant bits is the fallback for more than 64
// tables and under NRX_PLAN_SORT=rocprim.  Two more launches (head count per tile, then rank + emit) give
// the unique (table,row) list, the segment starts, the number of unique rows and the per-table split, all left on
// the device: the host reads n_tables + 2 integers once (or nothing, in the fused-optimizer mode).
#include "nrx_common.h"
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

struct PlanArgs {
    const void* ids[NRX_MAX_FEATURES];
    int64_t off[NRX_MAX_FEATURES + 1];
    int64_t rows[NRX_MAX_FEATURES];
    int32_t table_of[NRX_MAX_FEATURES];
    int32_t n_feats;
    int32_t idx64;
    int32_t row_bits;
    int64_t n_total;
};
static_assert(sizeof(PlanArgs) <= 3584, "kernarg budget");

template <typename KeyT>
__global__ __launch_bounds__(NRX_BLOCK) void plan_keys_kernel(const PlanArgs args_in_kernarg, KeyT* __restrict__ keys, uint32_t* __restrict__ payload) {
    const NRX_CONST PlanArgs* a = nrx_kernarg<PlanArgs>();
    for (int64_t p = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; p < a->n_total; p += (int64_t)gridDim.x * NRX_BLOCK) {
        int lo = 0, hi = a->n_feats;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a->off[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t i = p - a->off[lo];
        int64_t id = a->idx64 ? nrx_gconst<int64_t>(a->ids[lo])[i] : (int64_t)nrx_gconst<int32_t>(a->ids[lo])[i];
        if (id < 0 || id >= a->rows[lo]) id = 0;           // out-of-range ids were reported by the forward; the
        keys[p] = ((KeyT)a->table_of[lo] << a->row_bits) | (KeyT)id;     // padding row never trains
        payload[p] = (uint32_t)p;
    }
}

// Unique rows, segment starts and per-table bounds from the sorted keys in TWO launches (was: head flags, rocPRIM scan
// (+ its init kernel), finalize, bounds -- five launches of a few microseconds of work each):
//   plan_count_kernel   block c counts the segment heads among its PLAN_TILE sorted entries      -> block_heads[c]
//   plan_emit_kernel    block c sums block_heads[0..c) itself (a few KB, L2-resident), ranks its heads with one block
//                       scan and writes order / uniq_keys / seg_start; a head whose table differs from its
//                       predecessor's also writes that table's first-unique bound (counts[1 + t] for every table id in
//                       between, so tables without lookups get empty ranges); the last entry closes counts / seg_start.
constexpr int PLAN_TILE = 1024;        // entries per block = 4 per thread

// PAIR (32-bit keys from the segmented sort): the sorted list is ONE array of {key, payload} pairs
template <typename KeyT, bool PAIR>
__device__ __forceinline__ KeyT plan_key_at(const KeyT* __restrict__ skeys, int64_t e) {
    if (PAIR) return (KeyT)reinterpret_cast<const uint2*>(skeys)[e].x;
    return skeys[e];
}

template <typename KeyT, bool PAIR = false>
__device__ __forceinline__ bool plan_is_head(const KeyT* __restrict__ skeys, int64_t e) {
    return e == 0 || plan_key_at<KeyT, PAIR>(skeys, e) != plan_key_at<KeyT, PAIR>(skeys, e - 1);
}

// Placement (nrx_sparse_plan_place): a unique row that is looked up exactly ONCE in the launch, by a single-valued feature,
// needs no reduction -- its gradient row is the lookup's upstream row.  The plan then also says, per lookup p, where that row
// goes (dest[p] = its unique index, or -1) and lists the remaining unique rows (several lookups, a bag feature's lookup, or the
// padding row) as `walk`: the backward streams the upstream rows sample-major and places them (embed_bwd_place_kernel) and
// walks only the listed rows.  PlaceInfo: which features' lookups may be placed, by flat offset.
struct PlaceInfo {
    int64_t off[NRX_MAX_FEATURES + 1];
    uint64_t feats;                 // bit f: lookups of feature f may be placed
    int32_t n;
    int32_t all;                    // every feature may be placed: no feature lookup needed
    // SEG (segment-local keys): the sorted pairs carry the ROW only -- a launch whose table + row bits exceed 32 (C5: 40 tables up to 500 M rows)
    // still sorts 8-byte {row, lookup} pairs instead of 8-byte keys + 4-byte payloads; the table of a sorted position is the run it lies in
    int64_t seg_off[NRX_MAX_FEATURES + 1];      // first sorted position of table t's run (table-major; [n_seg] = n)
    int32_t n_seg, pad_;
};
// SEG: the table of sorted position e = the last run that starts at or before e.  t_lo / b1: the table of the tile's first position and where the next
// run starts (block-uniform, found once per tile); an entry past b1 walks on (rare: a tile that holds the end of a run).
struct PlanSeg {
    int t_lo;
    int64_t b1;
};
__device__ __forceinline__ PlanSeg plan_seg_of_tile(const NRX_CONST PlaceInfo* pi, int64_t e_first) {
    PlanSeg ps;
    // one lane per run start: ONE vector load, a compare, a ballot (a scalar loop over 40 starts was 40 dependent loads: the count kernel 8 -> 18 us)
    const int lane = threadIdx.x & 63;
    const bool hit = lane >= 1 && lane < pi->n_seg && pi->seg_off[lane < pi->n_seg ? lane : 0] <= e_first;
    const int t = __builtin_amdgcn_readfirstlane((int)__popcll(__ballot(hit)));
    ps.t_lo = t;
    ps.b1 = t + 1 < pi->n_seg ? pi->seg_off[t + 1] : 0x7fffffffffffffffLL;
    return ps;
}
__device__ __forceinline__ int plan_table_at(const NRX_CONST PlaceInfo* pi, const PlanSeg& ps, int64_t e) {
    if (e < ps.b1) return e < 0 ? -1 : ps.t_lo;
    int t = ps.t_lo + 1;
    while (t + 1 < pi->n_seg && e >= pi->seg_off[t + 1]) ++t;
    return t;
}
__device__ __forceinline__ bool plan_placeable(const NRX_CONST PlaceInfo* pi, uint32_t p) {
    if (pi->all) return true;
    int f = 0;
    for (int i = 1; i < pi->n; ++i) f += (int64_t)p >= pi->off[i] ? 1 : 0;      // uniform loop, scalar loads
    return (pi->feats >> f) & 1ull;
}
// entry e (key k, payload p) is PLACED iff it is a whole segment by itself, not the padding row, and of a placeable feature
template <typename KeyT>
__device__ __forceinline__ bool plan_is_placed(bool head, bool next_head, KeyT key, uint64_t rmask, const NRX_CONST PlaceInfo* pi, uint32_t p) {
    return head && next_head && ((uint64_t)key & rmask) != 0 && plan_placeable(pi, p);
}

template <typename KeyT, bool PAIR = false, bool PLACE = false, bool SEG = false, bool PAIRS = false>
__global__ __launch_bounds__(NRX_BLOCK) void plan_count_kernel(const PlaceInfo place_in_kernarg /* first: read through nrx_kernarg */,
                                                               const KeyT* __restrict__ skeys, int64_t n, uint32_t* __restrict__ block_heads,
                                                               const uint32_t* __restrict__ spayload, int row_bits) {
    constexpr bool want_pairs = PLACE && PAIRS;      // (a template flag: the extra loads and the third count were 2 us of C5's plan even when off)
    __shared__ uint32_t s_cnt[3 * (NRX_BLOCK / 64)];
    const int64_t e0 = (int64_t)blockIdx.x * PLAN_TILE + threadIdx.x;
    uint32_t c = 0, cw = 0, cp = 0;
    const uint64_t rmask = (1ull << row_bits) - 1;
    // (the tile's keys are requested before the run lookup: the two do not depend on each other, and a block is one short chain of round trips)
    constexpr int ROUNDS = PLAN_TILE / NRX_BLOCK;
    KeyT key[ROUNDS], prev[ROUNDS];
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t e = e0 + j * NRX_BLOCK;
        const int64_t ec = e < n ? e : n - 1;
        key[j] = plan_key_at<KeyT, PAIR>(skeys, ec);
        prev[j] = plan_key_at<KeyT, PAIR>(skeys, ec > 0 ? ec - 1 : 0);
    }
    PlanSeg ps = {0, 0};
    if (SEG) ps = plan_seg_of_tile(nrx_kernarg<PlaceInfo>(), (int64_t)blockIdx.x * PLAN_TILE > 0 ? (int64_t)blockIdx.x * PLAN_TILE - 1 : 0);
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t e = e0 + j * NRX_BLOCK;
        bool head = e < n && (e == 0 || key[j] != prev[j]);
        if (SEG && e < n && e > 0)
            head = head || plan_table_at(nrx_kernarg<PlaceInfo>(), ps, e) != plan_table_at(nrx_kernarg<PlaceInfo>(), ps, e - 1);
        c += (uint32_t)__popcll(__ballot(head));      // wave-uniform count
        if (PLACE) {
            const NRX_CONST PlaceInfo* pi = nrx_kernarg<PlaceInfo>();
            bool walked = false, pairh = false;
            if (head) {
                bool next_head = e + 1 >= n || plan_key_at<KeyT, PAIR>(skeys, e + 1) != key[j];
                if (SEG && e + 1 < n) next_head = next_head || plan_table_at(pi, ps, e + 1) != plan_table_at(pi, ps, e);
                const uint32_t pay = PAIR ? reinterpret_cast<const uint2*>(skeys)[e].y : spayload[e];
                walked = !plan_is_placed<KeyT>(true, next_head, key[j], rmask, pi, pay);
                if (want_pairs && !next_head && ((uint64_t)key[j] & rmask) != 0) {      // a row looked up exactly twice: a pair record, not a walk row
                    bool next2_head = e + 2 >= n || plan_key_at<KeyT, PAIR>(skeys, e + 2) != key[j];
                    if (SEG && e + 2 < n) next2_head = next2_head || plan_table_at(pi, ps, e + 2) != plan_table_at(pi, ps, e);
                    pairh = next2_head;
                }
                walked = walked && !pairh;
            }
            cw += (uint32_t)__popcll(__ballot(walked));
            cp += (uint32_t)__popcll(__ballot(pairh));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_cnt[threadIdx.x >> 6] = c;
        s_cnt[NRX_BLOCK / 64 + (threadIdx.x >> 6)] = cw;
        s_cnt[2 * (NRX_BLOCK / 64) + (threadIdx.x >> 6)] = cp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_heads[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (PLACE) block_heads[gridDim.x + blockIdx.x] = s_cnt[4] + s_cnt[5] + s_cnt[6] + s_cnt[7];
        if (PLACE && want_pairs) block_heads[2 * gridDim.x + blockIdx.x] = s_cnt[8] + s_cnt[9] + s_cnt[10] + s_cnt[11];
    }
}

// entry e of the tile is handled by thread (e % 256) in round (e / 256): coalesced key / payload / order accesses; the
// rank of a head = heads of earlier blocks + heads of earlier (round, wave) cells + heads of lower lanes in its cell
template <typename KeyT, bool PAIR = false, bool PLACE = false, bool SEG = false, bool PAIRS = false>
__global__ __launch_bounds__(NRX_BLOCK) void plan_emit_kernel(const PlaceInfo place_in_kernarg /* first: read through nrx_kernarg */,
                                                              const KeyT* __restrict__ skeys, const uint32_t* __restrict__ spayload,
                                                              const uint32_t* __restrict__ block_heads, int64_t n, int row_bits,
                                                              int32_t n_tables, int64_t* __restrict__ order,
                                                              int64_t* __restrict__ uniq_keys, int64_t* __restrict__ seg_start,
                                                              int64_t* __restrict__ counts, int32_t* __restrict__ dest,
                                                              int32_t* __restrict__ walk, int64_t* __restrict__ n_walk,
                                                              int32_t* __restrict__ pairs = nullptr, int64_t* __restrict__ n_pairs = nullptr) {
    constexpr int ROUNDS = PLAN_TILE / NRX_BLOCK, WAVES = NRX_BLOCK / 64;
    __shared__ uint32_t s_cell[ROUNDS * WAVES + 1];
    __shared__ uint32_t s_wcell[ROUNDS * WAVES + 1];
    __shared__ uint32_t s_pcell[ROUNDS * WAVES + 1];
    __shared__ uint32_t s_part[3 * WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // PLACE: the dest words of a tile (one table's lookups) all fall into that feature's stretch of dest -- B x 4 bytes.  Blocks
    // land on XCD (block % 8), each with its own L2: in launch order the eight XCDs would each hold PARTIAL lines of every
    // stretch and write them back piecemeal.  Giving XCD x the x-th eighth of the tiles (a bijection of the grid) keeps a
    // stretch inside one L2 until its lines are complete.
    unsigned tile = blockIdx.x;
    if (PLACE) {
        const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3, qt = gridDim.x >> 3, rt = gridDim.x & 7u;
        tile = x * qt + (x < rt ? x : rt) + i;
    }
    const int64_t e0 = (int64_t)tile * PLAN_TILE + tid;
    KeyT key[ROUNDS], prev[ROUNDS], next[ROUNDS], next2[ROUNDS];
    uint32_t pay[ROUNDS], pay1[ROUNDS];
    constexpr bool want_pairs = PLACE && PAIRS;
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {            // all of the tile's loads are issued before anything waits
        const int64_t e = e0 + j * NRX_BLOCK;
        const int64_t ec = e < n ? e : n - 1;
        if (PAIR) {
            const uint2 kp = reinterpret_cast<const uint2*>(skeys)[ec];
            key[j] = (KeyT)kp.x;
            pay[j] = kp.y;
        } else {
            key[j] = skeys[ec];
            pay[j] = spayload[ec];
        }
        prev[j] = plan_key_at<KeyT, PAIR>(skeys, ec > 0 ? ec - 1 : 0);
        if (PLACE) next[j] = plan_key_at<KeyT, PAIR>(skeys, ec + 1 < n ? ec + 1 : ec);
        next2[j] = key[j];
        pay1[j] = 0;
        if (want_pairs) {                             // the entry behind the next one, and the next one's lookup: a pair record is {u, this lookup, the next}
            next2[j] = plan_key_at<KeyT, PAIR>(skeys, ec + 2 < n ? ec + 2 : ec);
            const int64_t e1 = ec + 1 < n ? ec + 1 : ec;
            pay1[j] = PAIR ? reinterpret_cast<const uint2*>(skeys)[e1].y : spayload[e1];
        }
    }
    uint32_t acc = 0, wacc = 0, pacc = 0;
    for (uint32_t i = tid; i < tile; i += NRX_BLOCK) {
        acc += block_heads[i];
        if (PLACE) wacc += block_heads[gridDim.x + i];
        if (want_pairs) pacc += block_heads[2 * gridDim.x + i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off, 64);
        if (PLACE) wacc += __shfl_xor(wacc, off, 64);
        if (want_pairs) pacc += __shfl_xor(pacc, off, 64);
    }
    if (lane == 0) {
        s_part[wid] = acc;
        s_part[WAVES + wid] = wacc;
        s_part[2 * WAVES + wid] = pacc;
    }
    const uint64_t rmask = (1ull << row_bits) - 1;
    bool head[ROUNDS], walked[ROUNDS], placed[ROUNDS], able[ROUNDS], pairh[ROUNDS];
    unsigned long long mask[ROUNDS], wmask[ROUNDS], pmask[ROUNDS];
    int tab[ROUNDS], tabp[ROUNDS];                 // SEG: table of the entry / of its predecessor (-1 before the first)
    PlanSeg ps = {0, 0};
    if (SEG) ps = plan_seg_of_tile(nrx_kernarg<PlaceInfo>(), (int64_t)tile * PLAN_TILE > 0 ? (int64_t)tile * PLAN_TILE - 1 : 0);
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t e = e0 + j * NRX_BLOCK;
        head[j] = e < n && (e == 0 || key[j] != prev[j]);
        tab[j] = tabp[j] = 0;
        if (SEG && e < n) {
            tab[j] = plan_table_at(nrx_kernarg<PlaceInfo>(), ps, e);
            tabp[j] = e > 0 ? plan_table_at(nrx_kernarg<PlaceInfo>(), ps, e - 1) : -1;
            head[j] = head[j] || tab[j] != tabp[j];
        }
        mask[j] = __ballot(head[j]);
        if (lane == 0) s_cell[j * WAVES + wid] = (uint32_t)__popcll(mask[j]);
        if (PLACE) {
            bool next_head = e + 1 >= n || next[j] != key[j];
            if (SEG && e + 1 < n) next_head = next_head || plan_table_at(nrx_kernarg<PlaceInfo>(), ps, e + 1) != tab[j];
            able[j] = plan_placeable(nrx_kernarg<PlaceInfo>(), pay[j]);
            placed[j] = head[j] && next_head && ((uint64_t)key[j] & rmask) != 0 && able[j];
            pairh[j] = false;
            if (want_pairs && head[j] && !next_head && ((uint64_t)key[j] & rmask) != 0) {
                bool next2_head = e + 2 >= n || next2[j] != key[j];
                if (SEG && e + 2 < n) next2_head = next2_head || plan_table_at(nrx_kernarg<PlaceInfo>(), ps, e + 2) != tab[j];
                pairh[j] = next2_head;
            }
            walked[j] = head[j] && !placed[j] && !pairh[j];
            wmask[j] = __ballot(walked[j]);
            pmask[j] = want_pairs ? __ballot(pairh[j]) : 0ull;
            if (lane == 0) {
                s_wcell[j * WAVES + wid] = (uint32_t)__popcll(wmask[j]);
                if (want_pairs) s_pcell[j * WAVES + wid] = (uint32_t)__popcll(pmask[j]);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {                                // exclusive scan of the 16 cells, seeded with the earlier blocks' heads
        uint32_t run = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        for (int c = 0; c < ROUNDS * WAVES; ++c) {
            const uint32_t v = s_cell[c];
            s_cell[c] = run;
            run += v;
        }
        s_cell[ROUNDS * WAVES] = run;
    }
    if (PLACE && tid == 64) {
        uint32_t run = s_part[WAVES] + s_part[WAVES + 1] + s_part[WAVES + 2] + s_part[WAVES + 3];
        for (int c = 0; c < ROUNDS * WAVES; ++c) {
            const uint32_t v = s_wcell[c];
            s_wcell[c] = run;
            run += v;
        }
        s_wcell[ROUNDS * WAVES] = run;
    }
    if (want_pairs && tid == 128) {
        uint32_t run = s_part[2 * WAVES] + s_part[2 * WAVES + 1] + s_part[2 * WAVES + 2] + s_part[2 * WAVES + 3];
        for (int c = 0; c < ROUNDS * WAVES; ++c) {
            const uint32_t v = s_pcell[c];
            s_pcell[c] = run;
            run += v;
        }
        s_pcell[ROUNDS * WAVES] = run;
    }
    __syncthreads();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t e = e0 + j * NRX_BLOCK;
        if (e >= n) continue;
        order[e] = (int64_t)pay[j];
        const uint32_t u = s_cell[j * WAVES + wid] + (uint32_t)__popcll(mask[j] & lt);
        if (PLACE) {
            if (able[j]) dest[pay[j]] = placed[j] ? (int32_t)u : -1;      // every placeable lookup gets an answer: no fill pass
                                                                          // (the words of other features' lookups are never read)
            const uint32_t wr = s_wcell[j * WAVES + wid] + (uint32_t)__popcll(wmask[j] & lt);
            if (walked[j]) walk[wr] = (int32_t)u;
            if (e == n - 1) n_walk[0] = (int64_t)(wr + (walked[j] ? 1u : 0u));
            if (want_pairs) {
                const uint32_t pr = s_pcell[j * WAVES + wid] + (uint32_t)__popcll(pmask[j] & lt);
                if (pairh[j]) {
                    typedef int nrx_i32x4e __attribute__((ext_vector_type(4)));
                    nrx_i32x4e rec;
                    rec.x = (int32_t)u; rec.y = (int32_t)pay[j]; rec.z = (int32_t)pay1[j]; rec.w = 0;
                    reinterpret_cast<nrx_i32x4e*>(pairs)[pr] = rec;
                }
                if (e == n - 1) n_pairs[0] = (int64_t)pr;          // (the last entry starts no pair)
            }
        }
        if (head[j]) {
            const uint64_t k = (uint64_t)key[j];
            const int64_t t = SEG ? (int64_t)tab[j] : (int64_t)(k >> row_bits);
            uniq_keys[u] = (t << 40) | (int64_t)(k & rmask);
            seg_start[u] = e;
            const int64_t tprev = e == 0 ? -1 : (SEG ? (int64_t)tabp[j] : (int64_t)((uint64_t)prev[j] >> row_bits));
            for (int64_t tt = tprev + 1; tt <= t; ++tt) counts[1 + tt] = u;     // first unique entry of tables tprev+1 .. t
        }
        if (e == n - 1) {
            const uint32_t nu = u + (head[j] ? 1u : 0u);
            counts[0] = nu;
            seg_start[nu] = n;
            const int64_t tl = SEG ? (int64_t)tab[j] : (int64_t)((uint64_t)key[j] >> row_bits);
            for (int64_t tt = tl + 1; tt <= n_tables; ++tt) counts[1 + tt] = nu;
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Table-segmented LSD radix sort of the compact keys (the default planner sort; rocPRIM's onesweep stays as the fallback for
// shapes outside its limits and under NRX_PLAN_SORT=rocprim).
// The general-purpose sort has to order all table_bits + row_bits bits (C2: 25 bits = four 8-bit onesweep passes, each with its
// own memset + look-back chain: 155 us for 1.7 M pairs).  But the TABLE of a lookup is known from its feature, and the number
// of lookups per table is known on the host: the key kernel writes the pairs table-major to begin with (table t's lookups,
// in (feature, sample) order, at seg_off[t]), so only the ROW bits are left to sort, inside each table's segment, with
// digits as wide as a block ranks comfortably in LDS (up to 10 bits; every segment splits ITS row bits evenly over the
// passes): two passes for C2 (20 row bits = 2 x 10), three for a 10 M-row table.  One pass =
//   seg_hist_kernel      one block per 4096-entry tile (tiles never straddle a segment): LDS histogram of the digit
//                        (pass 0: fused into the key kernel)                                      -> hist[tile][bin]
//   seg_scan_chunks      (only when a segment has more than 32 tiles) per (32-tile chunk, bin): running sum over the
//                        chunk's tiles in place, chunk totals -> ctot[chunk][bin]
//   seg_scan_bins        one block per segment: per bin the running sum over the segment's tiles (or chunks), then the
//                        exclusive scan over the bins + the segment's base                         -> bin_base[seg][bin]
//   seg_scatter_kernel   re-reads the tile; each wavefront ranks its 1024 contiguous entries round by round (equal-digit
//                        peers by ballots, running per-wave bin counts in LDS); a cross-wave prefix + a block scan over
//                        the bins give the stable rank inside the tile; the pairs are first permuted INSIDE LDS and
//                        then written out in tile order, so that the entries of one bin leave as one contiguous run.
// Every step is order-preserving, so equal rows stay in (feature, sample) order: the result is the same permutation the
// stable rocPRIM sort produces (tests compare both against the oracle's plan).
// ---------------------------------------------------------------------------------------------------
#ifndef NRX_SEG_TILE
#define NRX_SEG_TILE 4096
#endif
#ifndef NRX_SEG_THREADS
#define NRX_SEG_THREADS 512
#endif
constexpr int SEG_TILE = NRX_SEG_TILE;              // entries per block
constexpr int SEG_THREADS = NRX_SEG_THREADS;        // threads of the tile kernels: 8 wavefronts x 512 contiguous entries (the
                                                    // kernels are latency-bound at these sizes: short serial chains, more of them)
constexpr int SEG_PER_THREAD = SEG_TILE / SEG_THREADS;
constexpr int SEG_MAX_DB = 10;                      // widest digit
constexpr int SEG_CHUNK = 32;                       // tiles per scan chunk
constexpr int SEG_DIRECT_CHUNKS = 128;              // longest segment (in chunks) whose chunk totals every scatter block sums itself

struct SegArgs {
    const void* ids[NRX_MAX_FEATURES];
    int64_t qoff[NRX_MAX_FEATURES + 1];      // table-major start of the q-th feature slot (slots ordered by table, then feature)
    int64_t poff[NRX_MAX_FEATURES];          // feature-major start (the payload) of the slot's feature
    int64_t rows[NRX_MAX_FEATURES];          // per slot
    int64_t seg_off[NRX_MAX_FEATURES + 1];   // per table: first entry
    int32_t seg_tile[NRX_MAX_FEATURES + 1];  // per table: first tile
    int32_t seg_chunk[NRX_MAX_FEATURES + 1]; // per table: first chunk
    // (dwords, bytes / nibbles packed: a byte array indexed by a wave-uniform number is still read with a VECTOR load -- the scalar unit
    // of this target loads dwords only -- and every tile kernel began with such a load and a full wait for it before its first useful one)
    uint32_t table_w[NRX_MAX_FEATURES / 4];  // per slot: table number, a byte each
    uint32_t seg_db_w[NRX_MAX_FEATURES / 8]; // per table: digit width, a nibble each (<= SEG_MAX_DB = 10)
    uint32_t seg_p0_w[NRX_MAX_FEATURES / 8]; // per table (LSD form): the first pass the segment takes part in, a nibble each.  A segment whose row bits
                                             // need fewer passes than the launch's widest table sits out the FIRST passes (C4: the 200 k-row news table
                                             // -- 97 % of the lookups -- needs two 9-bit passes where the 10 M-row user table needs three): the key
                                             // kernel leaves its pairs in the buffer its first pass reads, every pass kernel returns at once for it
    int32_t n_slots, n_seg, idx64, row_bits, nb;     // nb = 1 << (widest digit) = row stride of hist / ctot / bin_base
    int32_t xcd;                                      // 1: blocks take their tiles in XCD order (seg_block_tile)
    // MSD form (round 4; launches whose LSD sort would take three or more passes): ONE scatter pass on each segment's HIGH digit (its top
    // seg_db bits, i.e. key >> seg_shift), then the bins -- a few dozen entries each on uniform ids -- are sorted where they lie by
    // seg_binsort_kernel (comparison ranks, any number of low bits); bins too large for that go to a work list
    uint32_t seg_shift_w[NRX_MAX_FEATURES / 4];      // per table: low bits under the MSD digit, a byte each (0: the one pass sorts the segment)
    int32_t msd;
    int32_t small_max;                                // largest bin the rank sort takes
    uint32_t* bin_start;                              // [n_seg][nb]: global position of every bin's first entry (written by the scatter pass)
    int32_t* work;                                    // [4 + 4 * work_cap]: work[0] = items, then {start, size, low bits, 0} per large bin
    int32_t work_cap;
    // Padding split (launches with bag features; seg_split_kernel): the lookups of the padding row (id 0, out-of-range ids) -- half of a
    // padded history -- have nothing to sort: the key kernel writes them, in lookup order, to the FRONT of their table's segment in the buffer the
    // last pass leaves its result in, and the live pairs compacted behind them; every pass then sorts [seg_off + seg_lo, seg_end) only.
    int32_t segkey;                                   // 1: the pairs carry the row only (PlaceInfo::seg_off names the table of a sorted position)
    int32_t final_b;                                  // the buffer that holds the sorted pairs after the last pass: 0 = first, 1 = second
    uint32_t* seg_lo;                                 // [n_seg]: padding lookups of the segment (null: no split, every pass sorts whole segments)
    uint32_t* padcnt;                                 // [tiles]: padding lookups per input tile
    const uint32_t* payload_src;                      // NRX_PLAN_PAYLOAD: the sorted payload of lookup p is payload_src[p] instead of p (null: p)
};
static_assert(sizeof(SegArgs) <= 3584, "kernarg budget");
__device__ __forceinline__ int seg_db_of(const NRX_CONST SegArgs* a, int seg) {                 // seg wave-uniform: scalar loads and shifts
    seg = __builtin_amdgcn_readfirstlane(seg);
    return (int)((a->seg_db_w[seg >> 3] >> ((seg & 7) * 4)) & 15u);
}
__device__ __forceinline__ int seg_p0_of(const NRX_CONST SegArgs* a, int seg) {
    seg = __builtin_amdgcn_readfirstlane(seg);
    return (int)((a->seg_p0_w[seg >> 3] >> ((seg & 7) * 4)) & 15u);
}
__device__ __forceinline__ int seg_shift_of(const NRX_CONST SegArgs* a, int seg) {              // seg wave-uniform
    seg = __builtin_amdgcn_readfirstlane(seg);
    return (int)((a->seg_shift_w[seg >> 2] >> ((seg & 3) * 8)) & 255u);
}
__device__ __forceinline__ uint32_t seg_table_of(const NRX_CONST SegArgs* a, int slot) {        // slot: any lane's
    return (a->table_w[slot >> 2] >> ((slot & 3) * 8)) & 255u;
}

// Segment of a tile: the LAST segment whose first tile is <= tile = the number of entries 1 .. n-1 of the (<= 64-entry) array that are
// <= tile.  One lane per entry: ONE vector load from the argument block, one compare, a ballot and a popcount per wavefront -- a binary
// search is six DEPENDENT round trips to the argument block before the block can load anything, and an unrolled scalar count of all
// 64 entries cost the kernels around it 128 VGPRs.
template <typename T>
__device__ __forceinline__ int seg_count_le(const NRX_CONST T* arr, int n, T x) {
    const int lane = threadIdx.x & 63;
    const bool hit = lane >= 1 && lane < n && arr[lane < n ? lane : 0] <= x;
    return (int)__popcll(__ballot(hit));
}
// Tile of a block.  Workgroups land on XCD (block % 8), each XCD with its own L2: in launch order the 16 tiles of a C2 table are spread over
// all eight L2s, so the 32-byte runs a tile scatters into the table's 1024 bins reach memory as eight partial copies of every line, and
// the next pass (and the histogram rows in between) finds nothing of its input in its own L2.  With xcd set, XCD x takes the x-th eighth of
// the tiles (a bijection of the grid, as in plan_emit_kernel): a table's whole sort -- pairs in, histogram rows, pairs out -- stays inside one L2.
__device__ __forceinline__ int seg_block_tile(const NRX_CONST SegArgs* a) {
    if (!a->xcd) return (int)blockIdx.x;
    const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3, qt = gridDim.x >> 3, rt = gridDim.x & 7u;
    return (int)(x * qt + (x < rt ? x : rt) + i);
}
__device__ __forceinline__ int seg_of_tile(const NRX_CONST SegArgs* a, int tile) {
    return __builtin_amdgcn_readfirstlane(seg_count_le<int32_t>(a->seg_tile, a->n_seg, tile));
}

// First entry of `tile` and the end of its segment.  With the padding split the tiles of a segment cover its LIVE zone (the padding pairs at the
// front are in place already): the last tiles of the segment are then empty (q0 >= qend).
__device__ __forceinline__ void seg_tile_range(const NRX_CONST SegArgs* a, int tile, int seg, int64_t& q0, int64_t& qend) {
    const uint32_t lo = a->seg_lo != nullptr ? (uint32_t)__builtin_amdgcn_readfirstlane((int)a->seg_lo[seg]) : 0u;
    q0 = a->seg_off[seg] + (int64_t)lo + (int64_t)(tile - a->seg_tile[seg]) * SEG_TILE;
    qend = a->seg_off[seg + 1];
}

// PAIR (32-bit keys): {key, payload} travel as one 8-byte element through every pass and into plan_count / plan_emit -- the scatter's
// runs are short (4 entries per bin and tile at 10-bit digits), so one 32-byte piece per run instead of two 16-byte ones.
template <typename KeyT, bool PAIR>
__global__ __launch_bounds__(SEG_THREADS) void seg_keys_kernel(const SegArgs args_in_kernarg, KeyT* __restrict__ keys_a, uint32_t* __restrict__ payload_a,
                                                               uint32_t* __restrict__ hist, KeyT* __restrict__ keys_b, uint32_t* __restrict__ payload_b) {
    extern __shared__ uint32_t s_hist[];
    const NRX_CONST SegArgs* a = nrx_kernarg<SegArgs>();
    const int tile = seg_block_tile(a), seg = seg_of_tile(a, tile);
    // (a segment that starts at an odd pass: its pairs go to the buffer that pass reads)
    const bool odd = !a->msd && (seg_p0_of(a, seg) & 1) != 0;
    KeyT* __restrict__ keys = odd ? keys_b : keys_a;
    uint32_t* __restrict__ payload = odd ? payload_b : payload_a;
    const int nbins = 1 << seg_db_of(a, seg);
    const int hshift = a->msd ? seg_shift_of(a, seg) : 0;            // the digit this kernel histograms: the lowest (LSD) or the segment's highest (MSD)
    if (a->msd && blockIdx.x == 0 && threadIdx.x == 0) a->work[0] = 0;   // the scatter pass (next launch) appends the large bins
    for (int b = threadIdx.x; b < nbins; b += SEG_THREADS) s_hist[b] = 0;
    const int64_t q0 = a->seg_off[seg] + (int64_t)(tile - a->seg_tile[seg]) * SEG_TILE, qend = a->seg_off[seg + 1];
    const uint32_t dmask = (uint32_t)nbins - 1u;
    // first feature slot of the tile (wave-uniform, counted like seg_of_tile); a tile rarely holds a second one: then
    // every per-slot field is a scalar, otherwise the entries past the boundary walk on from there with per-lane indices
    const int lo = __builtin_amdgcn_readfirstlane(seg_count_le<int64_t>(a->qoff, a->n_slots, q0));
    const int64_t lo_end = a->qoff[lo + 1];
    const bool one_slot = lo_end >= qend || lo_end >= q0 + SEG_TILE;
    int sl[SEG_PER_THREAD];
    int64_t id[SEG_PER_THREAD];
    if (one_slot) {
        const int64_t i0 = q0 - a->qoff[lo];
        const void* idp = a->ids[lo];
#pragma unroll
        for (int j = 0; j < SEG_PER_THREAD; ++j) {          // all id loads issued before anything waits
            const int64_t q = q0 + j * SEG_THREADS + threadIdx.x;
            sl[j] = lo;
            id[j] = 0;
            if (q < qend) {
                const int64_t i = i0 + j * SEG_THREADS + threadIdx.x;
                id[j] = a->idx64 ? nrx_gconst<int64_t>(idp)[i] : (int64_t)nrx_gconst<int32_t>(idp)[i];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < SEG_PER_THREAD; ++j) {
            const int64_t q = q0 + j * SEG_THREADS + threadIdx.x;
            int s = lo;
            if (q >= lo_end && q < qend) {
                do ++s; while (a->qoff[s + 1] <= q);
            }
            sl[j] = s;
        }
#pragma unroll
        for (int j = 0; j < SEG_PER_THREAD; ++j) {
            const int64_t q = q0 + j * SEG_THREADS + threadIdx.x;
            id[j] = 0;
            if (q < qend) {
                const int64_t i = q - a->qoff[sl[j]];
                id[j] = a->idx64 ? nrx_gconst<int64_t>(a->ids[sl[j]])[i] : (int64_t)nrx_gconst<int32_t>(a->ids[sl[j]])[i];
            }
        }
    }
    __syncthreads();
    if (one_slot) {
        const int64_t rows = a->rows[lo], pbase = a->poff[lo] + (q0 - a->qoff[lo]);
        const KeyT tkey = a->segkey ? (KeyT)0 : (KeyT)((KeyT)seg_table_of(a, lo) << a->row_bits);
#pragma unroll
        for (int j = 0; j < SEG_PER_THREAD; ++j) {
            const int64_t q = q0 + j * SEG_THREADS + threadIdx.x;
            if (q < qend) {
                int64_t v = id[j];
                if (v < 0 || v >= rows) v = 0;            // out-of-range ids were reported by the forward; the padding row never trains
                uint32_t pv = (uint32_t)(pbase + j * SEG_THREADS + threadIdx.x);
                if (a->payload_src != nullptr) pv = a->payload_src[pv];
                if (PAIR) reinterpret_cast<uint2*>(keys)[q] = make_uint2((uint32_t)(tkey | (KeyT)v), pv);
                else {
                    keys[q] = tkey | (KeyT)v;
                    payload[q] = pv;
                }
                atomicAdd(&s_hist[(uint32_t)(v >> hshift) & dmask], 1u);
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < SEG_PER_THREAD; ++j) {
        const int64_t q = q0 + j * SEG_THREADS + threadIdx.x;
        if (q < qend) {
            const int s = sl[j];
            int64_t v = id[j];
            if (v < 0 || v >= a->rows[s]) v = 0;
            const KeyT tk = a->segkey ? (KeyT)0 : (KeyT)((KeyT)seg_table_of(a, s) << a->row_bits);
            uint32_t pv = (uint32_t)(a->poff[s] + (q - a->qoff[s]));
            if (a->payload_src != nullptr) pv = a->payload_src[pv];
            if (PAIR) reinterpret_cast<uint2*>(keys)[q] = make_uint2((uint32_t)(tk | (KeyT)v), pv);
            else {
                keys[q] = tk | (KeyT)v;
                payload[q] = pv;
            }
            atomicAdd(&s_hist[(uint32_t)(v >> hshift) & dmask], 1u);
        }
    }
    }
    __syncthreads();
    uint32_t* h = hist + (size_t)tile * a->nb;
    for (int b = threadIdx.x; b < nbins; b += SEG_THREADS) h[b] = s_hist[b];
}

// ---- padding split (SegArgs::seg_lo).  Rows and payloads of the entries of INPUT tile `tile` (entry j of a thread: q0 + j * SEG_THREADS + tid, as
// in seg_keys_kernel; row 0 = the padding row: id 0 and out-of-range ids, which the forward reported).
__device__ __forceinline__ void seg_tile_rows(const NRX_CONST SegArgs* a, int64_t q0, int64_t qend, int64_t (&v)[SEG_PER_THREAD],
                                              uint32_t (&pay)[SEG_PER_THREAD], uint32_t (&tab)[SEG_PER_THREAD]) {
    // a wave-uniform walk over the feature slots the tile touches (one, rarely two): every per-slot field is a scalar
    int s = __builtin_amdgcn_readfirstlane(seg_count_le<int64_t>(a->qoff, a->n_slots, q0));
    const int64_t tend = qend < q0 + SEG_TILE ? qend : q0 + SEG_TILE;
#pragma unroll
    for (int j = 0; j < SEG_PER_THREAD; ++j) { v[j] = 0; pay[j] = 0; tab[j] = 0; }
    for (;;) {
        const int64_t s_beg = a->qoff[s], s_end = a->qoff[s + 1], rows = a->rows[s], pb = a->poff[s] - s_beg;
        const void* idp = a->ids[s];
        const uint32_t t = seg_table_of(a, s);
        int64_t id[SEG_PER_THREAD];
#pragma unroll
        for (int j = 0; j < SEG_PER_THREAD; ++j) {          // all id loads issued before anything waits
            const int64_t q = q0 + j * SEG_THREADS + threadIdx.x;
            id[j] = 0;
            if (q >= s_beg && q < s_end && q < tend) id[j] = a->idx64 ? nrx_gconst<int64_t>(idp)[q - s_beg] : (int64_t)nrx_gconst<int32_t>(idp)[q - s_beg];
        }
#pragma unroll
        for (int j = 0; j < SEG_PER_THREAD; ++j) {
            const int64_t q = q0 + j * SEG_THREADS + threadIdx.x;
            if (q >= s_beg && q < s_end && q < tend) {
                v[j] = id[j] < 0 || id[j] >= rows ? 0 : id[j];
                pay[j] = (uint32_t)(pb + q);
                tab[j] = t;
            }
        }
        if (s_end >= tend) break;
        ++s;
    }
}

// padcnt[tile] = the padding lookups among the tile's entries
__global__ __launch_bounds__(SEG_THREADS) void seg_padcount_kernel(const SegArgs args_in_kernarg) {
    const NRX_CONST SegArgs* a = nrx_kernarg<SegArgs>();
    __shared__ uint32_t s_part[SEG_THREADS / 64];
    const int tile = seg_block_tile(a), seg = seg_of_tile(a, tile);
    const int64_t q0 = a->seg_off[seg] + (int64_t)(tile - a->seg_tile[seg]) * SEG_TILE, qend = a->seg_off[seg + 1];
    int64_t v[SEG_PER_THREAD];
    uint32_t pay[SEG_PER_THREAD], tab[SEG_PER_THREAD];
    seg_tile_rows(a, q0, qend, v, pay, tab);
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < SEG_PER_THREAD; ++j) c += (q0 + j * SEG_THREADS + threadIdx.x < qend && v[j] == 0) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < SEG_THREADS / 64; ++w) t += s_part[w];
        a->padcnt[tile] = t;
    }
}

// The key kernel of the split: a stable two-way partition of the segment.  Padding pairs -> [seg_off, seg_off + P) of the buffer the last pass
// writes (they are in their final place: row 0 sorts first, equal keys keep lookup order); live pairs -> [seg_off + P, seg_end) of the buffer the
// segment's first pass reads, in lookup order.  P and the counts of the earlier tiles: summed by the block from padcnt[] (a segment has at most
// SEG_SPLIT_MAX_TILES tiles in this mode).  The first pass's histogram is taken by seg_hist_kernel (the live tiles are not the input tiles).
constexpr int SEG_SPLIT_MAX_TILES = 8192;
template <typename KeyT, bool PAIR>
__global__ __launch_bounds__(SEG_THREADS) void seg_split_kernel(const SegArgs args_in_kernarg, KeyT* __restrict__ keys_a, uint32_t* __restrict__ payload_a,
                                                                KeyT* __restrict__ keys_b, uint32_t* __restrict__ payload_b) {
    constexpr int WAVES = SEG_THREADS / 64, NE = WAVES * SEG_PER_THREAD;
    static_assert(NE <= 64, "one wavefront scans the (round, wavefront) counts");
    __shared__ uint32_t s_cnt[NE], s_pre[NE], s_part[2][WAVES];
    const NRX_CONST SegArgs* a = nrx_kernarg<SegArgs>();
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = seg_block_tile(a), seg = seg_of_tile(a, tile);
    if (a->msd && blockIdx.x == 0 && tid == 0) a->work[0] = 0;       // the scatter pass (a later launch) appends the large bins
    const int t0 = a->seg_tile[seg], t1 = a->seg_tile[seg + 1];
    const int64_t q0 = a->seg_off[seg] + (int64_t)(tile - t0) * SEG_TILE, qend = a->seg_off[seg + 1];
    uint32_t before = 0, all = 0;                    // padding lookups of the segment's earlier tiles / of the whole segment
    for (int t = t0 + tid; t < t1; t += SEG_THREADS) {
        const uint32_t c = a->padcnt[t];
        all += c;
        before += t < tile ? c : 0u;
    }
    int64_t v[SEG_PER_THREAD];
    uint32_t pay[SEG_PER_THREAD], tab[SEG_PER_THREAD];
    seg_tile_rows(a, q0, qend, v, pay, tab);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { before += __shfl_xor(before, off, 64); all += __shfl_xor(all, off, 64); }
    if (lane == 0) { s_part[0][wid] = before; s_part[1][wid] = all; }
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t mine[SEG_PER_THREAD];
    bool pad[SEG_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SEG_PER_THREAD; ++j) {
        pad[j] = q0 + j * SEG_THREADS + tid < qend && v[j] == 0;
        const unsigned long long bal = __ballot(pad[j]);
        mine[j] = (uint32_t)__popcll(bal & lt);
        if (lane == 0) s_cnt[j * WAVES + wid] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (wid == 0) {                                  // exclusive scan of the counts in entry order: round-major, then wavefront
        const uint32_t c = lane < NE ? s_cnt[lane] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t x = __shfl_up(inc, off, 64); if (lane >= off) inc += x; }
        if (lane < NE) s_pre[lane] = inc - c;
    }
    __syncthreads();
    before = 0; all = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { before += s_part[0][w]; all += s_part[1][w]; }
    if (tile == t0 && tid == 0) a->seg_lo[seg] = all;
    const bool odd = !a->msd && (seg_p0_of(a, seg) & 1) != 0;        // (a segment that starts at an odd pass: its pairs go to the buffer that pass reads)
    KeyT* __restrict__ lk = odd ? keys_b : keys_a;
    uint32_t* __restrict__ lp = odd ? payload_b : payload_a;
    KeyT* __restrict__ pk = a->final_b ? keys_b : keys_a;
    uint32_t* __restrict__ pp = a->final_b ? payload_b : payload_a;
    const int64_t pad_base = a->seg_off[seg] + (int64_t)before;
    const int64_t live_base = a->seg_off[seg] + (int64_t)all + ((int64_t)(tile - t0) * SEG_TILE - (int64_t)before);
#pragma unroll
    for (int j = 0; j < SEG_PER_THREAD; ++j) {
        const int idx = j * SEG_THREADS + tid;
        if (q0 + idx >= qend) continue;
        const uint32_t pb = s_pre[j * WAVES + wid] + mine[j];         // padding lookups of the tile before this entry
        const KeyT key = (a->segkey ? (KeyT)0 : (KeyT)((KeyT)tab[j] << a->row_bits)) | (KeyT)v[j];
        const int64_t pos = pad[j] ? pad_base + pb : live_base + ((int64_t)idx - (int64_t)pb);
        KeyT* kk = pad[j] ? pk : lk;
        uint32_t* pq = pad[j] ? pp : lp;
        if (PAIR) reinterpret_cast<uint2*>(kk)[pos] = make_uint2((uint32_t)key, pay[j]);
        else { kk[pos] = key; pq[pos] = pay[j]; }
    }
}

template <typename KeyT, bool PAIR>
__global__ __launch_bounds__(SEG_THREADS) void seg_hist_kernel(const SegArgs args_in_kernarg, const KeyT* __restrict__ keys, int pass,
                                                             uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t s_hist[];
    const NRX_CONST SegArgs* a = nrx_kernarg<SegArgs>();
    const int tile = seg_block_tile(a), seg = seg_of_tile(a, tile);
    const int dpass = a->msd ? 0 : pass - seg_p0_of(a, seg);
    if (dpass < 0) return;                           // the segment sits this pass out
    const int db = seg_db_of(a, seg), nbins = 1 << db, shift = a->msd ? seg_shift_of(a, seg) : dpass * db;
    int64_t q0, qend;
    seg_tile_range(a, tile, seg, q0, qend);
    if (q0 >= qend) {                                // (padding split: a tile past the live zone) -- the scans still sum its row
        uint32_t* hz = hist + (size_t)tile * a->nb;
        for (int b = threadIdx.x; b < nbins; b += SEG_THREADS) hz[b] = 0;
        return;
    }
    for (int b = threadIdx.x; b < nbins; b += SEG_THREADS) s_hist[b] = 0;
    const uint32_t dmask = (uint32_t)nbins - 1u;
    KeyT k[SEG_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SEG_PER_THREAD; ++j) {
        const int64_t q = q0 + j * SEG_THREADS + threadIdx.x;
        k[j] = plan_key_at<KeyT, PAIR>(keys, q < qend ? q : qend - 1);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SEG_PER_THREAD; ++j)
        if (q0 + j * SEG_THREADS + threadIdx.x < qend) atomicAdd(&s_hist[(uint32_t)(k[j] >> shift) & dmask], 1u);
    __syncthreads();
    uint32_t* h = hist + (size_t)tile * a->nb;
    for (int b = threadIdx.x; b < nbins; b += SEG_THREADS) h[b] = s_hist[b];
}

// segments of more than SEG_CHUNK tiles: hist[tile][bin] -> exclusive prefix over the earlier tiles of the tile's chunk
// (in place); ctot[chunk][bin] = the chunk's total.  grid (nb / 256, chunks)
__global__ __launch_bounds__(NRX_BLOCK) void seg_scan_chunks(const SegArgs args_in_kernarg, uint32_t* __restrict__ hist, uint32_t* __restrict__ ctot, int pass) {
    const NRX_CONST SegArgs* a = nrx_kernarg<SegArgs>();
    const int chunk = blockIdx.y;
    const int seg = __builtin_amdgcn_readfirstlane(seg_count_le<int32_t>(a->seg_chunk, a->n_seg, chunk));
    if (!a->msd && pass < seg_p0_of(a, seg)) return;
    const int nb = a->nb;
    const int bin = blockIdx.x * NRX_BLOCK + threadIdx.x;
    if (bin >= (1 << seg_db_of(a, seg))) return;
    const int t0 = a->seg_tile[seg] + (chunk - a->seg_chunk[seg]) * SEG_CHUNK;
    const int t1 = t0 + SEG_CHUNK < a->seg_tile[seg + 1] ? t0 + SEG_CHUNK : a->seg_tile[seg + 1];
    uint32_t* h = hist + (size_t)t0 * nb + bin;
    uint32_t run = 0;
    int t = t0;
    for (; t + 4 <= t1; t += 4, h += (size_t)4 * nb) {       // four independent loads in flight
        const uint32_t v0 = h[0], v1 = h[nb], v2 = h[2 * (size_t)nb], v3 = h[3 * (size_t)nb];
        h[0] = run; h[nb] = run + v0; h[2 * (size_t)nb] = run + v0 + v1; h[3 * (size_t)nb] = run + v0 + v1 + v2;
        run += v0 + v1 + v2 + v3;
    }
    for (; t < t1; ++t, h += nb) { const uint32_t v = h[0]; h[0] = run; run += v; }
    ctot[(size_t)chunk * nb + bin] = run;
}

// One block per segment (launched after seg_scan_chunks, for segments too long for the scatter blocks to sum their chunk totals
// themselves).  Per bin: running sum over the segment's chunks (ctot in place); then the exclusive scan over the bins + the
// segment's first entry -> bin_base[seg][bin].
__global__ __launch_bounds__(NRX_BLOCK) void seg_scan_bins(const SegArgs args_in_kernarg, uint32_t* __restrict__ hist, uint32_t* __restrict__ ctot,
                                                            uint32_t* __restrict__ bin_base, int pass) {
    __shared__ uint32_t s_part[NRX_BLOCK / 64];
    const NRX_CONST SegArgs* a = nrx_kernarg<SegArgs>();
    const int seg = blockIdx.x, nb = a->nb, nbins = 1 << seg_db_of(a, seg);
    if (!a->msd && pass < seg_p0_of(a, seg)) return;
    constexpr int PER = (1 << SEG_MAX_DB) / NRX_BLOCK;            // bins per thread: bin = i * 256 + tid (coalesced rows)
    const int nchunks = a->seg_chunk[seg + 1] - a->seg_chunk[seg];
    uint32_t v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int bin = i * NRX_BLOCK + threadIdx.x;
        uint32_t run = 0;
        if (bin < nbins) {
            // always over the chunk totals: seg_scan_chunks has already turned EVERY segment's histogram rows into in-chunk
            // prefixes, one-chunk segments included (summing those rows again here corrupted the short segments of a launch
            // that also held a long one -- caught by the 83-tile parity case)
            uint32_t* h = ctot + (size_t)a->seg_chunk[seg] * nb + bin;
            const int cnt = nchunks;
            int t = 0;
            for (; t + 8 <= cnt; t += 8, h += (size_t)8 * nb) {      // eight independent loads in flight
                uint32_t x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = h[(size_t)u * nb];
#pragma unroll
                for (int u = 0; u < 8; ++u) { h[(size_t)u * nb] = run; run += x[u]; }
            }
            for (; t < cnt; ++t, h += nb) { const uint32_t x = h[0]; h[0] = run; run += x; }
        }
        v[i] = run;
    }
    // exclusive scan over the bins in bin order: bin = i * 256 + tid -> scan each i-row across the block, carry between rows
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t carry = (uint32_t)a->seg_off[seg] + (a->seg_lo != nullptr ? a->seg_lo[seg] : 0u);
    uint32_t* o = bin_base + (size_t)seg * nb;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (i * NRX_BLOCK >= nbins) break;
        uint32_t inc = v[i];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t x = __shfl_up(inc, off, 64); if (lane >= off) inc += x; }
        __syncthreads();
        if (lane == 63) s_part[wid] = inc;
        __syncthreads();
        uint32_t base = carry;
        for (int w = 0; w < wid; ++w) base += s_part[w];
        const int bin = i * NRX_BLOCK + threadIdx.x;
        if (bin < nbins) o[bin] = base + inc - v[i];
        carry += s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
}

// DIRECT 1 (every segment has at most SEG_CHUNK tiles): no scan launches at all -- the block sums the histogram rows of its
// own segment itself (<= 32 L2-resident rows: the earlier tiles' counts and the segment totals per bin) and scans the totals
// together with its local ones; hist is read-only in this mode.  DIRECT 2 (longer segments, at most SEG_DIRECT_CHUNKS chunks
// each): seg_scan_chunks runs, the block sums the segment's chunk totals the same way -- seg_scan_bins is not launched.
template <typename KeyT, int DIRECT, bool PAIR>
__global__ __launch_bounds__(SEG_THREADS) void seg_scatter_kernel(const SegArgs args_in_kernarg, const KeyT* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ pay_in, int pass,
                                                                const uint32_t* __restrict__ hist, const uint32_t* __restrict__ ctot,
                                                                const uint32_t* __restrict__ bin_base,
                                                                KeyT* __restrict__ keys_out, uint32_t* __restrict__ pay_out) {
    constexpr int WAVES = SEG_THREADS / 64, ROUNDS = SEG_TILE / SEG_THREADS;      // 8 waves x 8 rounds of 64 entries
    extern __shared__ uint32_t s_mem[];
    const NRX_CONST SegArgs* a = nrx_kernarg<SegArgs>();
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = seg_block_tile(a), seg = seg_of_tile(a, tile);
    const int db = seg_db_of(a, seg), nbins = 1 << db, nb = a->nb;
    const int dpass = a->msd ? 0 : pass - seg_p0_of(a, seg);
    if (dpass < 0) return;                           // the segment sits this pass out: its pairs already lie in the buffer its first pass reads
    const int shift = a->msd ? seg_shift_of(a, seg) : dpass * db;
    uint16_t* s_wh = reinterpret_cast<uint16_t*>(s_mem);  // [WAVES][nbins]: running bin counts of each wave's 512-entry chunk (16-bit: a tile
                                                          // holds 4096 entries; halves this area -- a third / fourth resident block per CU)
    uint32_t* s_bin = s_mem + WAVES * nbins / 2;          // [nbins]: the bin's first position inside the tile
    uint32_t* s_gb = s_bin + nbins;                       // [nbins]: global position of the bin's first entry of this tile - s_bin
    uint32_t* s_pay = s_gb + nbins;                       // [SEG_TILE]
    KeyT* s_key = reinterpret_cast<KeyT*>(s_pay + SEG_TILE);      // [SEG_TILE]
    __shared__ uint32_t s_part[WAVES], s_gpart[WAVES];
    int64_t q0, qend;
    seg_tile_range(a, tile, seg, q0, qend);
    if (q0 >= qend) {                                // (padding split: a tile past the live zone)
        if (DIRECT == 1 && a->msd && tile == a->seg_tile[seg])       // a segment of padding lookups only: every bin is empty
            for (int b = tid; b < nbins; b += SEG_THREADS) a->bin_start[(size_t)seg * nb + b] = (uint32_t)qend;
        return;
    }
    for (int b = tid; b < WAVES * nbins / 2; b += SEG_THREADS) s_mem[b] = 0;
    const int count = (int)(qend - q0 < SEG_TILE ? qend - q0 : SEG_TILE);
    const int64_t qw = q0 + (int64_t)wid * (64 * ROUNDS) + lane;         // wave w owns entries [w * 512, (w + 1) * 512) of the tile
    const uint32_t dmask = (uint32_t)nbins - 1u;
    KeyT key[ROUNDS];
    uint32_t pay[ROUNDS], loc[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {            // all loads issued before anything waits
        const int64_t q = qw + r * 64;
        const int64_t qc = q < qend ? q : qend - 1;
        if (PAIR) {
            const uint2 kp = reinterpret_cast<const uint2*>(keys_in)[qc];
            key[r] = (KeyT)kp.x;
            pay[r] = kp.y;
        } else {
            key[r] = keys_in[qc];
            pay[r] = pay_in[qc];
        }
    }
    // the tile's global bases per bin: in flight during the ranking
    constexpr int PER = (1 << SEG_MAX_DB) / SEG_THREADS;
    const int chunk = a->seg_chunk[seg] + (tile - a->seg_tile[seg]) / SEG_CHUNK;
    const bool chunked = a->seg_chunk[seg + 1] - a->seg_chunk[seg] > 1;
    uint32_t gbase[PER], gtot[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = i * SEG_THREADS + tid;
        gbase[i] = 0;
        gtot[i] = 0;
        if (b < nbins) {
            if (DIRECT) {
                // DIRECT 1: the rows are the segment's tile histograms; DIRECT 2 (seg_scan_chunks has run): the rows are its chunk
                // totals, and the tile's own histogram row already holds the counts of the earlier tiles of its chunk
                const int t0 = DIRECT == 1 ? a->seg_tile[seg] : a->seg_chunk[seg], t1 = DIRECT == 1 ? a->seg_tile[seg + 1] : a->seg_chunk[seg + 1];
                const int mine = DIRECT == 1 ? tile : chunk;
                const uint32_t* h = (DIRECT == 1 ? hist : ctot) + (size_t)t0 * nb + b;
                uint32_t before = DIRECT == 1 ? 0u : hist[(size_t)tile * nb + b], all = 0;
                int t = t0;
                for (; t + 16 <= t1; t += 16, h += (size_t)16 * nb) {      // sixteen independent loads in flight (a C2 table is 16 tiles: one round trip, not two)
                    uint32_t v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = h[(size_t)u * nb];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { all += v[u]; before += t + u < mine ? v[u] : 0u; }
                }
                for (; t + 8 <= t1; t += 8, h += (size_t)8 * nb) {         // eight independent loads in flight
                    uint32_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = h[(size_t)u * nb];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { all += v[u]; before += t + u < mine ? v[u] : 0u; }
                }
                for (; t < t1; ++t, h += nb) {
                    const uint32_t v = h[0];
                    all += v;
                    before += t < mine ? v : 0u;
                }
                gbase[i] = before;
                gtot[i] = all;
            } else {
                gbase[i] = bin_base[(size_t)seg * nb + b] + hist[(size_t)tile * nb + b];
                if (chunked) gbase[i] += ctot[(size_t)chunk * nb + b];
            }
        }
    }
    __syncthreads();
    uint16_t* wh = s_wh + wid * nbins;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool valid = qw + r * 64 < qend;
        const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < db; ++b) {             // lanes of this round with the same digit
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt);
        uint32_t base = 0;
        if (valid) {
            base = wh[d];                                         // every peer reads the count before the leader bumps it
            if (before == 0) wh[d] = (uint16_t)(base + (uint32_t)__popcll(peers));
        }
        loc[r] = base + before;
    }
    __syncthreads();
    // per bin: exclusive prefix over the waves (in place) and the tile total; then the block scan of the totals
    uint32_t tot[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = i * SEG_THREADS + tid;
        uint32_t run = 0;
        if (b < nbins) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { const uint32_t v = s_wh[w * nbins + b]; s_wh[w * nbins + b] = (uint16_t)run; run += v; }
        }
        tot[i] = run;
    }
    uint32_t carry = 0, gcarry = DIRECT ? (uint32_t)(q0 - (int64_t)(tile - a->seg_tile[seg]) * SEG_TILE) : 0u;      // the segment's first (live) entry
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (i * SEG_THREADS >= nbins) break;
        uint32_t inc = tot[i], ginc = gtot[i];            // the tile's counts and (DIRECT) the segment's totals, scanned together
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t x = __shfl_up(inc, off, 64);
            if (lane >= off) inc += x;
            if (DIRECT) { const uint32_t y = __shfl_up(ginc, off, 64); if (lane >= off) ginc += y; }
        }
        __syncthreads();
        if (lane == 63) { s_part[wid] = inc; if (DIRECT) s_gpart[wid] = ginc; }
        __syncthreads();
        uint32_t base = carry, gb = gcarry;
        for (int w = 0; w < wid; ++w) { base += s_part[w]; if (DIRECT) gb += s_gpart[w]; }
        const int b = i * SEG_THREADS + tid;
        if (b < nbins) {
            const uint32_t start = base + inc - tot[i];
            s_bin[b] = start;
            s_gb[b] = (DIRECT ? gb + ginc - gtot[i] + gbase[i] : gbase[i]) - start;
            if (DIRECT == 1 && a->msd && tile == a->seg_tile[seg]) {      // the segment's first tile publishes where every bin begins
                const uint32_t bs = gb + ginc - gtot[i];
                a->bin_start[(size_t)seg * nb + b] = bs;
                if (shift > 0 && (int)gtot[i] > a->small_max) {            // too large for the rank sort: onto the work list
                    const int k = atomicAdd(&a->work[0], 1);
                    if (k < a->work_cap) {
                        a->work[4 + 4 * k] = (int32_t)bs;
                        a->work[5 + 4 * k] = (int32_t)gtot[i];
                        a->work[6 + 4 * k] = shift;
                        a->work[7 + 4 * k] = 0;
                    }
                }
            }
        }
        for (int w = 0; w < WAVES; ++w) { carry += s_part[w]; if (DIRECT) gcarry += s_gpart[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (qw + r * 64 < qend) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
            const uint32_t lp = s_bin[d] + wh[d] + loc[r];
            if (PAIR) reinterpret_cast<uint2*>(s_pay)[lp] = make_uint2((uint32_t)key[r], pay[r]);      // s_pay .. s_key are one 8-byte-per-entry area
            else {
                s_key[lp] = key[r];
                s_pay[lp] = pay[r];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SEG_PER_THREAD; ++j) {         // tile order = bin order: the entries of one bin leave as one run
        const int i = j * SEG_THREADS + tid;
        if (i < count) {
            if (PAIR) {
                const uint2 kp = reinterpret_cast<const uint2*>(s_pay)[i];
                const uint32_t pos = s_gb[(kp.x >> shift) & dmask] + (uint32_t)i;
                reinterpret_cast<uint2*>(keys_out)[pos] = kp;
            } else {
                const KeyT k = s_key[i];
                const uint32_t pos = s_gb[(uint32_t)(k >> shift) & dmask] + (uint32_t)i;
                keys_out[pos] = k;
                pay_out[pos] = s_pay[i];
            }
        }
    }
}

// MSD form, second step: the bins of a segment (entries that share table and high digit, in source order) are sorted by their low bits WHERE THEY
// LIE.  grid (nb / 4 + SEG_WORKERS, n_seg).  Blocks x < nb / 4: one wavefront per bin of segment y -- up to 256 entries, 4 per lane; every entry's
// (low bits, position) packs into one word, the bin's words go to the wavefront's 1 KB of LDS, and an entry's rank is the number of smaller words
// (broadcast reads, one compare per pair): a stable order whatever the width of the low bits.  The other blocks walk the work list of the bins
// that are larger (heavily skewed ids): a block sorts such a bin alone by streaming LSD passes through the second pair buffer (histogram by
// all wavefronts, the order-preserving placement by one wavefront, ballots as in seg_scatter_kernel) -- slow but bounded, and exact.
constexpr int SEG_SMALL_BIN = 256;
constexpr int SEG_WORKERS = 2;           // work-list blocks per grid row
#ifndef NRX_SEG_BPW
#define NRX_SEG_BPW 1                    // measured on C5 / C2 (plan alone, us): 1 bin per wavefront 112.9 / 73.3, 2: 114.7 / 73.6, 4: 119.7 / 77.3
#endif
constexpr int SEG_BPW = NRX_SEG_BPW;     // bins per wavefront of the rank sort

template <typename KeyT, bool PAIR>
__device__ __forceinline__ void seg_pair_load(const KeyT* keys, const uint32_t* pay, int64_t i, KeyT& k, uint32_t& p) {
    if (PAIR) { const uint2 v = reinterpret_cast<const uint2*>(keys)[i]; k = (KeyT)v.x; p = v.y; }
    else { k = keys[i]; p = pay[i]; }
}
template <typename KeyT, bool PAIR>
__device__ __forceinline__ void seg_pair_store(KeyT* keys, uint32_t* pay, int64_t i, KeyT k, uint32_t p) {
    if (PAIR) reinterpret_cast<uint2*>(keys)[i] = make_uint2((uint32_t)k, p);
    else { keys[i] = k; pay[i] = p; }
}

template <int NU>
__device__ __forceinline__ void seg_rank_words(const uint32_t* sl, int sz, const uint32_t (&w)[4], uint32_t (&rank)[4]) {
    for (int j = 0; j < sz; j += 4) {                             // broadcast reads: every lane the same four words
        const uint4 q = *reinterpret_cast<const uint4*>(sl + j);
#pragma unroll
        for (int u = 0; u < NU; ++u)
            rank[u] += (uint32_t)(q.x < w[u]) + (uint32_t)(q.y < w[u]) + (uint32_t)(q.z < w[u]) + (uint32_t)(q.w < w[u]);
    }
}

template <typename KeyT, bool PAIR>
__global__ __launch_bounds__(NRX_BLOCK) void seg_binsort_kernel(const SegArgs args_in_kernarg, KeyT* __restrict__ keys, uint32_t* __restrict__ pay,
                                                                KeyT* __restrict__ keys2, uint32_t* __restrict__ pay2) {
    const NRX_CONST SegArgs* a = nrx_kernarg<SegArgs>();
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nb = a->nb, nbx = (nb + 4 * SEG_BPW - 1) / (4 * SEG_BPW);
    __shared__ __attribute__((aligned(16))) uint32_t s_low[NRX_BLOCK / 64][SEG_SMALL_BIN + 4];
    __shared__ uint32_t s_hist[1 << SEG_MAX_DB];
    __shared__ uint32_t s_scan[NRX_BLOCK / 64];
    if ((int)blockIdx.x < nbx) {
        const int seg = blockIdx.y;
        const int shift = seg_shift_of(a, seg);
        if (shift == 0 || a->seg_off[seg + 1] == a->seg_off[seg]) return;      // the scatter pass sorted this segment completely / no lookups
        const int nbins = 1 << seg_db_of(a, seg);
        // a wavefront takes SEG_BPW consecutive bins and requests ALL their entries before it ranks the first: one bin per wavefront was three
        // dependent round trips (bin bounds, entries, stores) for 64 entries -- 41 000 wavefronts of ~7 us each on C5, 48 us for the launch
        const int bin0 = (blockIdx.x * (NRX_BLOCK / 64) + wid) * SEG_BPW;
        if (bin0 >= nbins) return;                               // wave-uniform
        const uint32_t* bs = a->bin_start + (size_t)seg * nb;
        const int64_t seg_end = a->seg_off[seg + 1];
        int64_t bnd = seg_end;                                   // lane l < SEG_BPW + 1: where bin bin0 + l begins (the segment's end past the last bin)
        if (lane <= SEG_BPW && bin0 + lane < nbins) bnd = (int64_t)bs[bin0 + lane];
        int64_t start[SEG_BPW];
        int sz[SEG_BPW];
#pragma unroll
        for (int g = 0; g < SEG_BPW; ++g) {
            start[g] = __shfl(bnd, g, 64);
            sz[g] = (int)(__shfl(bnd, g + 1, 64) - start[g]);
            if (bin0 + g >= nbins || sz[g] > SEG_SMALL_BIN) sz[g] = 0;         // past the segment's bins / the work list's
        }
        const uint32_t lowmask = (1u << shift) - 1u;
        KeyT k[SEG_BPW][4];
        uint32_t p[SEG_BPW][4];
#pragma unroll
        for (int g = 0; g < SEG_BPW; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = lane + 64 * u;
                k[g][u] = 0; p[g][u] = 0;
                if (i < sz[g] && sz[g] > 1) seg_pair_load<KeyT, PAIR>(keys, pay, start[g] + i, k[g][u], p[g][u]);
            }
        uint32_t* sl = s_low[wid];
#pragma unroll
        for (int g = 0; g < SEG_BPW; ++g) {
            if (sz[g] <= 1) continue;                            // wave-uniform: nothing to order
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = lane + 64 * u;
                w[u] = ((((uint32_t)k[g][u]) & lowmask) << 8) | (uint32_t)i;      // (low bits, position): distinct words, <= 21 + 8 bits
                if (i < sz[g]) sl[i] = w[u];
            }
            if (lane < 4) sl[sz[g] + lane] = 0xffffffffu;         // the tail of the last 4-word read
            __builtin_amdgcn_wave_barrier();
            uint32_t rank[4] = {0u, 0u, 0u, 0u};
            // a lane owns entries lane, lane + 64, ...: only the first ceil(sz / 64) of its four slots hold one (wave-uniform) -- ranking all four
            // regardless made the launch VALU-bound (a 64-entry bin did four times the compares it needs: 57 us on C5)
            const int nu = (sz[g] + 63) >> 6;
            if (nu == 1) seg_rank_words<1>(sl, sz[g], w, rank);
            else if (nu == 2) seg_rank_words<2>(sl, sz[g], w, rank);
            else if (nu == 3) seg_rank_words<3>(sl, sz[g], w, rank);
            else seg_rank_words<4>(sl, sz[g], w, rank);
            __builtin_amdgcn_wave_barrier();                      // the next bin reuses the words
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (lane + 64 * u < sz[g]) seg_pair_store<KeyT, PAIR>(keys, pay, start[g] + rank[u], k[g][u], p[g][u]);
        }
        return;
    }
    // ---- work-list blocks: the bins beyond SEG_SMALL_BIN entries, one at a time per block
    const int nwork = a->work[0] < a->work_cap ? a->work[0] : a->work_cap;
    const int worker = ((int)blockIdx.x - nbx) * gridDim.y + blockIdx.y, nworkers = SEG_WORKERS * gridDim.y;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int it = worker; it < nwork; it += nworkers) {
        const int64_t start = a->work[4 + 4 * it];
        const int sz = a->work[5 + 4 * it], lowbits = a->work[6 + 4 * it];
        const int passes = (lowbits + SEG_MAX_DB - 1) / SEG_MAX_DB, dbw = (lowbits + passes - 1) / passes, nbins = 1 << dbw;
        const uint32_t dmask = (uint32_t)nbins - 1u;
        KeyT* sk = keys; uint32_t* sp = pay; KeyT* dk = keys2; uint32_t* dp = pay2;
        for (int pass = 0; pass < passes; ++pass) {
            const int sh = pass * dbw;
            for (int b = tid; b < nbins; b += NRX_BLOCK) s_hist[b] = 0;
            __syncthreads();
            for (int i = tid; i < sz; i += NRX_BLOCK) {
                KeyT kk; uint32_t pp;
                seg_pair_load<KeyT, PAIR>(sk, sp, start + i, kk, pp);
                atomicAdd(&s_hist[((uint32_t)kk >> sh) & dmask], 1u);
            }
            __syncthreads();
            {   // exclusive scan of the <= 1024 counts: 4 consecutive bins per thread
                uint32_t v[4], sum = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int b = tid * 4 + u; v[u] = b < nbins ? s_hist[b] : 0u; sum += v[u]; }
                uint32_t inc = sum;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const uint32_t x = __shfl_up(inc, off, 64); if (lane >= off) inc += x; }
                if (lane == 63) s_scan[wid] = inc;
                __syncthreads();
                uint32_t base = inc - sum;
                for (int w2 = 0; w2 < wid; ++w2) base += s_scan[w2];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int b = tid * 4 + u; if (b < nbins) s_hist[b] = base; base += v[u]; }
            }
            __syncthreads();
            if (wid == 0) {                                                  // order-preserving placement: one wavefront, 64 entries per round
                for (int r0 = 0; r0 < sz; r0 += 64) {
                    const int i = r0 + lane;
                    const bool valid = i < sz;
                    KeyT kk = 0; uint32_t pp = 0;
                    if (valid) seg_pair_load<KeyT, PAIR>(sk, sp, start + i, kk, pp);
                    const uint32_t d = ((uint32_t)kk >> sh) & dmask;
                    unsigned long long peers = __ballot(valid);
                    for (int b = 0; b < dbw; ++b) {
                        const bool bit = (d >> b) & 1u;
                        const unsigned long long bal = __ballot(bit);
                        peers &= bit ? bal : ~bal;
                    }
                    const uint32_t before = (uint32_t)__popcll(peers & lt);
                    uint32_t base = 0;
                    if (valid) {
                        base = s_hist[d];                                     // every peer reads the bin's position before its leader moves it
                        if (before == 0) s_hist[d] = base + (uint32_t)__popcll(peers);
                        seg_pair_store<KeyT, PAIR>(dk, dp, start + base + before, kk, pp);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
            KeyT* tk = sk; sk = dk; dk = tk;
            uint32_t* tp = sp; sp = dp; dp = tp;
        }
        if (sk != keys) {                                                     // an odd number of passes: the result sits in the second buffer
            for (int i = tid; i < sz; i += NRX_BLOCK) {
                KeyT kk; uint32_t pp;
                seg_pair_load<KeyT, PAIR>(sk, sp, start + i, kk, pp);
                seg_pair_store<KeyT, PAIR>(keys, pay, start + i, kk, pp);
            }
        }
        __syncthreads();
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename KeyT>
size_t sort_temp_bytes(int64_t n, int bits) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const KeyT*)nullptr, (KeyT*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (size_t)n, 0u, (unsigned)bits, (hipStream_t) nullptr);
    return bytes;
}

// scratch of the table-segmented sort: per-tile histograms + per-segment totals and bases, at the widest digit
inline size_t seg_work_cap(size_t n) { return n / SEG_SMALL_BIN + 64; }      // bins of more than SEG_SMALL_BIN entries: at most n / 256 of them
inline size_t seg_scratch_bytes(size_t n) {
    const size_t tiles = n / SEG_TILE + NRX_MAX_FEATURES + 1;          // hist rows; ctot has at most as many rows as there are tiles / 32 + segments
    return align256((tiles + tiles / SEG_CHUNK + 2 * (size_t)NRX_MAX_FEATURES + 1) * ((size_t)1 << SEG_MAX_DB) * sizeof(uint32_t)) +
           align256(((size_t)NRX_MAX_FEATURES << SEG_MAX_DB) * sizeof(uint32_t)) + align256((4 + 4 * seg_work_cap(n)) * sizeof(int32_t)) +      // MSD: bin starts, work list
           align256(tiles * sizeof(uint32_t)) + 256;                                                                                               // padding split: padcnt, seg_lo
}

int bits_for(int64_t v) {      // bits needed to represent values 0 .. v-1 (at least 1)
    int b = 1;
    while (((int64_t)1 << b) < v) ++b;
    return b;
}

// Stable LSD sort of n (key, payload) pairs on the low `bits` bits of the keys, from the planner's tile kernels: the whole array is ONE
// segment, digits of up to SEG_MAX_DB bits split evenly over the passes, per pass a histogram launch, the chunk scan (more than 32 tiles)
// and the ranking scatter.  Keys and payloads travel as separate arrays (the consumers -- nrx_route_ids_dedup, nrx_unique_inverse -- read
// them that way).  Returns through src / psrc the buffers that hold the sorted result (the in / out pair ping-pongs once per pass).
// `scratch`: seg_scratch_bytes(n) bytes.  Same permutation as rocprim::radix_sort_pairs on the same bits (every step is order-preserving).
template <typename KeyT>
void seg_sort_generic(KeyT*& src, KeyT*& dst, uint32_t*& psrc, uint32_t*& pdst, int64_t n, int bits, void* scratch, hipStream_t st) {
    SegArgs sa;
    memset(&sa, 0, sizeof(sa));
    const int passes = (bits + SEG_MAX_DB - 1) / SEG_MAX_DB;
    const int db = (bits + passes - 1) / passes;
    const int tiles = (int)((n + SEG_TILE - 1) / SEG_TILE), chunks = (tiles + SEG_CHUNK - 1) / SEG_CHUNK;
    sa.seg_off[0] = 0; sa.seg_off[1] = n;
    sa.seg_tile[0] = 0; sa.seg_tile[1] = tiles;
    sa.seg_chunk[0] = 0; sa.seg_chunk[1] = chunks;
    sa.seg_db_w[0] = (uint32_t)db;
    sa.qoff[0] = 0; sa.qoff[1] = n;
    sa.n_slots = 1; sa.n_seg = 1; sa.idx64 = 0; sa.row_bits = bits; sa.nb = 1 << db;
    { const char* e = getenv("NRX_SEG_XCD"); sa.xcd = e ? atoi(e) : 1; }
    const int nb = sa.nb;
    uint32_t* hist = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* ctot = hist + (size_t)tiles * nb;
    uint32_t* bin_base = ctot + (size_t)chunks * nb;
    const size_t lds_hist = (size_t)nb * 4;
    const size_t lds_scatter = (size_t)nb * (2 * (SEG_THREADS / 64) + 8) + (size_t)SEG_TILE * (4 + sizeof(KeyT));
    static const bool lds_ok = [] {
        const int bytes = (1 << SEG_MAX_DB) * (2 * (SEG_THREADS / 64) + 8) + SEG_TILE * (4 + (int)sizeof(KeyT));
        return hipFuncSetAttribute(reinterpret_cast<const void*>(seg_scatter_kernel<KeyT, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(seg_scatter_kernel<KeyT, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(seg_scatter_kernel<KeyT, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    }();
    (void)lds_ok;
    const dim3 gchunks((unsigned)((nb + NRX_BLOCK - 1) / NRX_BLOCK), (unsigned)chunks);
    for (int pass = 0; pass < passes; ++pass) {
        hipLaunchKernelGGL((seg_hist_kernel<KeyT, false>), dim3((unsigned)tiles), dim3(SEG_THREADS), lds_hist, st, sa, (const KeyT*)src, pass, hist);
        const char* se = getenv("NRX_PLAN_SORT");
        const bool force_bins = se && !strcmp(se, "segmented-bins");      // tests: the seg_scan_bins path of very long arrays
        if (tiles <= SEG_CHUNK) {
            hipLaunchKernelGGL((seg_scatter_kernel<KeyT, 1, false>), dim3((unsigned)tiles), dim3(SEG_THREADS), lds_scatter, st, sa, (const KeyT*)src,
                               (const uint32_t*)psrc, pass, (const uint32_t*)hist, (const uint32_t*)ctot, (const uint32_t*)bin_base, dst, pdst);
        } else if (chunks <= SEG_DIRECT_CHUNKS && !force_bins) {
            hipLaunchKernelGGL(seg_scan_chunks, gchunks, dim3(NRX_BLOCK), 0, st, sa, hist, ctot, pass);
            hipLaunchKernelGGL((seg_scatter_kernel<KeyT, 2, false>), dim3((unsigned)tiles), dim3(SEG_THREADS), lds_scatter, st, sa, (const KeyT*)src,
                               (const uint32_t*)psrc, pass, (const uint32_t*)hist, (const uint32_t*)ctot, (const uint32_t*)bin_base, dst, pdst);
        } else {
            hipLaunchKernelGGL(seg_scan_chunks, gchunks, dim3(NRX_BLOCK), 0, st, sa, hist, ctot, pass);
            hipLaunchKernelGGL(seg_scan_bins, dim3(1), dim3(NRX_BLOCK), 0, st, sa, hist, ctot, bin_base, pass);
            hipLaunchKernelGGL((seg_scatter_kernel<KeyT, 0, false>), dim3((unsigned)tiles), dim3(SEG_THREADS), lds_scatter, st, sa, (const KeyT*)src,
                               (const uint32_t*)psrc, pass, (const uint32_t*)hist, (const uint32_t*)ctot, (const uint32_t*)bin_base, dst, pdst);
        }
        KeyT* tk = src; src = dst; dst = tk;
        uint32_t* tp = psrc; psrc = pdst; pdst = tp;
    }
}

}  // namespace

extern "C" int64_t nrx_sparse_plan_workspace(int64_t n_lookups) {
    if (n_lookups < 0 || n_lookups >= 0xffffffffLL) return -1;
    const size_t n = (size_t)(n_lookups > 0 ? n_lookups : 1);
    const size_t t1 = sort_temp_bytes<uint64_t>(n, 64);
    return (int64_t)(2 * align256(n * 8) + 2 * align256(n * 4) + 2 * align256(n * 4) + align256(t1) + seg_scratch_bytes(n) + 256);
}

static int sparse_plan_impl(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                            int32_t n_feats, int32_t index_bits, int32_t n_tables, int64_t* order, int64_t* uniq_keys,
                            int64_t* seg_start, int64_t* counts, uint64_t place_feats, int32_t* dest, int32_t* walk, int64_t* n_walk,
                            void* workspace, void* stream, uint32_t opt_flags = 0, int32_t* pairs_out = nullptr, int64_t* n_pairs_out = nullptr,
                            const uint32_t* payload_src = nullptr) {
    NRX_REQUIRE(n_feats >= 0 && n_feats <= NRX_MAX_FEATURES && (index_bits == 32 || index_bits == 64) && n_tables >= 1 && n_tables < (1 << 20),
                "nrx_sparse_plan: bad argument");
    NRX_REQUIRE(counts != nullptr, "nrx_sparse_plan: null counts");
    PlanArgs a;
    int64_t off = 0, max_rows = 1;
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(lens[f] >= 0 && (lens[f] == 0 || ids[f] != nullptr) && table_of[f] >= 0 && table_of[f] < n_tables && rows[f] >= 1,
                    "nrx_sparse_plan: bad feature entry");
        a.ids[f] = ids[f];
        a.off[f] = off;
        a.rows[f] = rows[f];
        a.table_of[f] = table_of[f];
        off += lens[f];
        if (rows[f] > max_rows) max_rows = rows[f];
    }
    a.off[n_feats] = off;
    a.n_feats = n_feats;
    a.idx64 = index_bits == 64;
    a.n_total = off;
    PlaceInfo pinfo;
    memset(&pinfo, 0, sizeof(pinfo));
    if (dest != nullptr) {
        for (int f = 0; f <= n_feats; ++f) pinfo.off[f] = a.off[f];
        pinfo.n = n_feats;
        const uint64_t every = n_feats >= 64 ? ~0ull : ((1ull << n_feats) - 1);
        pinfo.feats = place_feats & every;
        pinfo.all = pinfo.feats == every;
    }
    // pair records (rows looked up exactly twice leave the walk list): only when every feature's lookups may be placed
    if (dest == nullptr || !pinfo.all || n_pairs_out == nullptr) pairs_out = nullptr;
    const int want_pairs = pairs_out != nullptr ? 1 : 0;
    const int row_bits = bits_for(max_rows), table_bits = bits_for(n_tables);
    NRX_REQUIRE(row_bits <= 40 && row_bits + table_bits <= 62, "nrx_sparse_plan: table too large for the composite key");
    NRX_REQUIRE(off < 0xffffffffLL, "nrx_sparse_plan: too many lookups for one plan");
    a.row_bits = row_bits;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = off;
    if (n == 0) {
        // counts = {0, 0, ..., 0}; seg_start[0] = 0
        int e = nrx_zero_async(counts, sizeof(int64_t) * (size_t)(n_tables + 2), st);
        if (e == NRX_OK && seg_start) e = nrx_zero_async(seg_start, sizeof(int64_t), st);
        if (e == NRX_OK && n_walk) e = nrx_zero_async(n_walk, sizeof(int64_t), st);
        if (e == NRX_OK && n_pairs_out) e = nrx_zero_async(n_pairs_out, sizeof(int64_t), st);
        if (e != NRX_OK) return NRX_ERR_LAUNCH;
        return NRX_OK;
    }
    NRX_REQUIRE(order && uniq_keys && seg_start && workspace, "nrx_sparse_plan: null buffer");
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    char* keys_in = w;                 w += align256((size_t)n * 8);
    char* keys_out = w;                w += align256((size_t)n * 8);
    uint32_t* pay_in = (uint32_t*)w;   w += align256((size_t)n * 4);
    uint32_t* pay_out = (uint32_t*)w;  w += align256((size_t)n * 4);
    uint32_t* flags = (uint32_t*)w;    w += align256((size_t)n * 4);
    w += align256((size_t)n * 4);
    void* temp = w;
    const int bits = row_bits + table_bits;
    int64_t g = (n + NRX_BLOCK - 1) / NRX_BLOCK;
    const unsigned gtile = (unsigned)((n + PLAN_TILE - 1) / PLAN_TILE);
    if (g > 4096) g = 4096;
    hipError_t err = hipSuccess;
    size_t tb = 0;
    // ---- default: table-segmented LSD sort (see the kernels' header); NRX_PLAN_SORT=rocprim forces the library sort
    const char* sort_env = getenv("NRX_PLAN_SORT");              // read per call: tests switch it between calls
    const bool force_rocprim = sort_env && !strcmp(sort_env, "rocprim");
    if (payload_src != nullptr && (force_rocprim || n_tables > NRX_MAX_FEATURES)) {
        nrx_set_error("nrx_sparse_plan_ex: NRX_PLAN_PAYLOAD is served by the table-segmented sort only (<= %d tables, NRX_PLAN_SORT != rocprim)", NRX_MAX_FEATURES);
        return NRX_ERR_UNSUPPORTED;
    }
    if (!force_rocprim && n_tables <= NRX_MAX_FEATURES) {
        SegArgs sa;
        sa.payload_src = payload_src;
        memset(sa.table_w, 0, sizeof(sa.table_w));
        memset(sa.seg_db_w, 0, sizeof(sa.seg_db_w));
        const bool force_bins = sort_env && !strcmp(sort_env, "segmented-bins");      // tests: the seg_scan_bins path of very long segments
        int digit_cap = SEG_MAX_DB;                                   // NRX_PLAN_DIGIT_BITS narrows the digits (measurement knob)
        if (const char* e = getenv("NRX_PLAN_DIGIT_BITS")) { const int v = atoi(e); if (v >= 4 && v <= SEG_MAX_DB) digit_cap = v; }
        const int passes = (row_bits + digit_cap - 1) / digit_cap;
        // MSD form (see SegArgs::msd): where the LSD sort would need three passes or more (tables beyond 2^20 rows: C3's 100 M-row table, C5),
        // when every segment has at most SEG_CHUNK tiles and its bins come out small on uniform ids.  NRX_PLAN_SORT=msd forces it wherever
        // the structure allows (tests), =lsd forbids it.
        const bool force_msd = sort_env && !strcmp(sort_env, "msd");
        const bool forbid_msd = sort_env && (!strcmp(sort_env, "lsd") || force_bins);
        bool msd = !forbid_msd && (force_msd || passes >= 3) && n < (1ll << 30);
        memset(sa.seg_shift_w, 0, sizeof(sa.seg_shift_w));
        memset(sa.seg_p0_w, 0, sizeof(sa.seg_p0_w));
        const bool skip_off = getenv("NRX_SEG_SKIP") != nullptr && atoi(getenv("NRX_SEG_SKIP")) == 0;      // A/B knob: every segment takes every pass
        if (msd) {
            for (int t = 0; t < n_tables && msd; ++t) {
                int64_t seg_rows = 1, seg_len = 0;
                for (int f = 0; f < n_feats; ++f)
                    if (table_of[f] == t && lens[f] > 0) { seg_len += lens[f]; if (rows[f] > seg_rows) seg_rows = rows[f]; }
                const int rb = bits_for(seg_rows), dbm = rb < SEG_MAX_DB ? rb : SEG_MAX_DB, sh = rb - dbm;
                if ((seg_len + SEG_TILE - 1) / SEG_TILE > SEG_CHUNK || sh > 21) msd = false;                  // DIRECT 1 scatter only; (low bits, position) in 29 bits
                else if (sh > 0 && !force_msd && (seg_len >> dbm) > SEG_SMALL_BIN / 2) msd = false;          // average bin beyond 128 entries: the rank sort would not pay
            }
        }
        int slot = 0, tile = 0, chunk = 0, max_db = 1;
        bool chunked = false;
        int max_chunks = 0;
        int64_t q = 0;
        for (int t = 0; t < n_tables; ++t) {
            sa.seg_off[t] = q;
            sa.seg_tile[t] = tile;
            sa.seg_chunk[t] = chunk;
            int64_t seg_rows = 1;
            for (int f = 0; f < n_feats; ++f) {
                if (table_of[f] != t || lens[f] == 0) continue;
                sa.ids[slot] = ids[f];
                sa.qoff[slot] = q;
                sa.poff[slot] = a.off[f];
                sa.rows[slot] = rows[f];
                sa.table_w[slot >> 2] |= (uint32_t)(t & 255) << ((slot & 3) * 8);
                if (rows[f] > seg_rows) seg_rows = rows[f];
                q += lens[f];
                ++slot;
            }
            // the segment's row bits, split evenly over the passes IT needs (a narrow table next to a wide one sits out the first passes)
            const int passes_t = (bits_for(seg_rows) + digit_cap - 1) / digit_cap > 0 ? (bits_for(seg_rows) + digit_cap - 1) / digit_cap : 1;
            const int p0 = (msd || skip_off) ? 0 : passes - (passes_t < passes ? passes_t : passes);
            int db = (bits_for(seg_rows) + (passes - p0) - 1) / (passes - p0);
            sa.seg_p0_w[t >> 3] |= (uint32_t)(p0 & 15) << ((t & 7) * 4);
            if (msd) {                                                       // ... or its top <= 10 bits, the rest left to the bin sort
                const int rb = bits_for(seg_rows);
                db = rb < SEG_MAX_DB ? rb : SEG_MAX_DB;
                sa.seg_shift_w[t >> 2] |= (uint32_t)((rb - db) & 255) << ((t & 3) * 8);
            }
            sa.seg_db_w[t >> 3] |= (uint32_t)(db & 15) << ((t & 7) * 4);
            if (db > max_db) max_db = db;
            const int tiles = (int)((q - sa.seg_off[t] + SEG_TILE - 1) / SEG_TILE);
            tile += tiles;
            chunk += (tiles + SEG_CHUNK - 1) / SEG_CHUNK;
            chunked |= tiles > SEG_CHUNK;
            if ((tiles + SEG_CHUNK - 1) / SEG_CHUNK > max_chunks) max_chunks = (tiles + SEG_CHUNK - 1) / SEG_CHUNK;
        }
        sa.seg_off[n_tables] = q;
        sa.seg_tile[n_tables] = tile;
        sa.seg_chunk[n_tables] = chunk;
        sa.qoff[slot] = q;
        sa.n_slots = slot;
        sa.n_seg = n_tables;
        sa.idx64 = a.idx64;
        sa.row_bits = row_bits;
        sa.nb = 1 << max_db;
        // tiles in XCD order (seg_block_tile): C2 plan 80.2 -> 73.8 us, C4 / C5 unchanged (profiles/r04_planner_variants.txt); NRX_SEG_XCD=0: launch order
        { const char* e = getenv("NRX_SEG_XCD"); sa.xcd = e ? atoi(e) : 1; }
        const int nb = sa.nb;
        uint32_t* hist = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(temp) + align256(sort_temp_bytes<uint64_t>(n, 64)));
        uint32_t* ctot = hist + (size_t)tile * nb;
        uint32_t* bin_base = ctot + (size_t)chunk * nb;
        {   // MSD outputs of the scatter pass: behind the LSD scratch (seg_scratch_bytes)
            const size_t tiles_cap = (size_t)n / SEG_TILE + NRX_MAX_FEATURES + 1;
            char* m0 = reinterpret_cast<char*>(hist) + align256((tiles_cap + tiles_cap / SEG_CHUNK + 2 * (size_t)NRX_MAX_FEATURES + 1) * ((size_t)1 << SEG_MAX_DB) * sizeof(uint32_t));
            sa.bin_start = reinterpret_cast<uint32_t*>(m0);
            sa.work = reinterpret_cast<int32_t*>(m0 + align256(((size_t)NRX_MAX_FEATURES << SEG_MAX_DB) * sizeof(uint32_t)));
            sa.work_cap = (int32_t)seg_work_cap((size_t)n);
            sa.small_max = SEG_SMALL_BIN;
            sa.msd = msd ? 1 : 0;
            sa.padcnt = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sa.work) + align256((4 + 4 * seg_work_cap((size_t)n)) * sizeof(int32_t)));
            sa.seg_lo = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sa.padcnt) + align256(tiles_cap * sizeof(uint32_t)));
            sa.final_b = msd ? 1 : (passes & 1);
        }
        // Padding split (SegArgs::seg_lo), asked for by the caller (nrx_sparse_plan_ex, NRX_PLAN_SPLIT_PADDING): worth it when a large share of the
        // lookups name the padding row (padded histories: 142 -> 114 us for 3.4 M lookups, half of them padding) -- two more launches (the
        // count, the first pass's histogram), then every pass on the live pairs only; a launch without padding pays ~17 us for nothing, and the
        // planner cannot know: the caller does, from the statistics of its previous batch.  NRX_PLAN_PADSPLIT=1 / 0 forces / forbids it
        // (tests run every planner case both ways).
        bool split = (opt_flags & NRX_PLAN_SPLIT_PADDING) != 0;
        {
            const char* e = getenv("NRX_PLAN_PADSPLIT");
            if (e && e[0] == '1') split = true;
            else if (e && e[0] == '0') split = false;
            for (int t = 0; t < n_tables && split; ++t)
                if (sa.seg_tile[t + 1] - sa.seg_tile[t] > SEG_SPLIT_MAX_TILES) split = false;
            if (!split) { sa.seg_lo = nullptr; sa.padcnt = nullptr; }
            else if (getenv("NRX_SEG_XCD") == nullptr) sa.xcd = 0;         // the live tiles are the FIRST tiles of a segment: in XCD order (an eighth of the tile list per XCD) the empty tiles
                                     // of a half-padding segment would idle whole XCDs; in launch order consecutive tiles alternate over the XCDs
        }
        const size_t lds_hist = (size_t)nb * 4;
        const dim3 gchunks((unsigned)((nb + NRX_BLOCK - 1) / NRX_BLOCK), (unsigned)chunk);
#define NRX_SEGSORT(KeyT, SEG_)                                                                                           \
    {                                                                                                                     \
        constexpr bool PAIR_ = sizeof(KeyT) == 4;                                                                         \
        const size_t lds_scatter = (size_t)nb * (2 * (SEG_THREADS / 64) + 8) + (size_t)SEG_TILE * (4 + sizeof(KeyT));         \
        static const bool lds_ok = [] {                                                                                   \
            const int bytes = (1 << SEG_MAX_DB) * (2 * (SEG_THREADS / 64) + 8) + SEG_TILE * (4 + (int)sizeof(KeyT));            \
            return hipFuncSetAttribute(reinterpret_cast<const void*>(seg_scatter_kernel<KeyT, 0, PAIR_>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess && \
                   hipFuncSetAttribute(reinterpret_cast<const void*>(seg_scatter_kernel<KeyT, 1, PAIR_>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess && \
                   hipFuncSetAttribute(reinterpret_cast<const void*>(seg_scatter_kernel<KeyT, 2, PAIR_>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess; \
        }();                                                                                                              \
        (void)lds_ok;                                                                                                     \
        KeyT* src = (KeyT*)keys_in; KeyT* dst = (KeyT*)keys_out;                                                          \
        uint32_t* psrc = pay_in; uint32_t* pdst = pay_out;                                                                \
        if (split) {                                                                                                      \
            hipLaunchKernelGGL(seg_padcount_kernel, dim3((unsigned)tile), dim3(SEG_THREADS), 0, st, sa);                         \
            hipLaunchKernelGGL((seg_split_kernel<KeyT, PAIR_>), dim3((unsigned)tile), dim3(SEG_THREADS), 0, st, sa, src, psrc, dst, pdst); \
            if (msd) hipLaunchKernelGGL((seg_hist_kernel<KeyT, PAIR_>), dim3((unsigned)tile), dim3(SEG_THREADS), lds_hist, st, sa, (const KeyT*)src, 0, hist); \
        } else {                                                                                                          \
            hipLaunchKernelGGL((seg_keys_kernel<KeyT, PAIR_>), dim3((unsigned)tile), dim3(SEG_THREADS), lds_hist, st, sa, src, psrc, hist, dst, pdst); \
        }                                                                                                                 \
        if (msd) {                                                                                                        \
            hipLaunchKernelGGL((seg_scatter_kernel<KeyT, 1, PAIR_>), dim3((unsigned)tile), dim3(SEG_THREADS), lds_scatter, st, sa, (const KeyT*)src, \
                               (const uint32_t*)psrc, 0, (const uint32_t*)hist, (const uint32_t*)ctot, (const uint32_t*)bin_base, dst, pdst); \
            KeyT* tk = src; src = dst; dst = tk;                                                                          \
            uint32_t* tp = psrc; psrc = pdst; pdst = tp;                                                                  \
            hipLaunchKernelGGL((seg_binsort_kernel<KeyT, PAIR_>), dim3((unsigned)((nb + 4 * SEG_BPW - 1) / (4 * SEG_BPW) + SEG_WORKERS), (unsigned)n_tables), dim3(NRX_BLOCK), 0, st, \
                               sa, src, psrc, dst, pdst);                                                                 \
        }                                                                                                                 \
        for (int pass = 0; pass < (msd ? 0 : passes); ++pass) {                                                           \
            if (pass > 0 || split) hipLaunchKernelGGL((seg_hist_kernel<KeyT, PAIR_>), dim3((unsigned)tile), dim3(SEG_THREADS), lds_hist, st, sa, (const KeyT*)src, pass, hist); \
            if (chunked && (max_chunks > SEG_DIRECT_CHUNKS || force_bins)) {                                              \
                hipLaunchKernelGGL(seg_scan_chunks, gchunks, dim3(NRX_BLOCK), 0, st, sa, hist, ctot, pass);                      \
                hipLaunchKernelGGL(seg_scan_bins, dim3((unsigned)n_tables), dim3(NRX_BLOCK), 0, st, sa, hist, ctot, bin_base, pass); \
                hipLaunchKernelGGL((seg_scatter_kernel<KeyT, 0, PAIR_>), dim3((unsigned)tile), dim3(SEG_THREADS), lds_scatter, st, sa, (const KeyT*)src, \
                                   (const uint32_t*)psrc, pass, (const uint32_t*)hist, (const uint32_t*)ctot, (const uint32_t*)bin_base, dst, pdst); \
            } else if (chunked) {                                                                                         \
                hipLaunchKernelGGL(seg_scan_chunks, gchunks, dim3(NRX_BLOCK), 0, st, sa, hist, ctot, pass);                      \
                hipLaunchKernelGGL((seg_scatter_kernel<KeyT, 2, PAIR_>), dim3((unsigned)tile), dim3(SEG_THREADS), lds_scatter, st, sa, (const KeyT*)src, \
                                   (const uint32_t*)psrc, pass, (const uint32_t*)hist, (const uint32_t*)ctot, (const uint32_t*)bin_base, dst, pdst); \
            } else {                                                                                                      \
                hipLaunchKernelGGL((seg_scatter_kernel<KeyT, 1, PAIR_>), dim3((unsigned)tile), dim3(SEG_THREADS), lds_scatter, st, sa, (const KeyT*)src, \
                                   (const uint32_t*)psrc, pass, (const uint32_t*)hist, (const uint32_t*)ctot, (const uint32_t*)bin_base, dst, pdst); \
            }                                                                                                             \
            KeyT* tk = src; src = dst; dst = tk;                                                                          \
            uint32_t* tp = psrc; psrc = pdst; pdst = tp;                                                                  \
        }                                                                                                                 \
        if (dest != nullptr) {                                                                                            \
            if (want_pairs) {                                                                                             \
                hipLaunchKernelGGL((plan_count_kernel<KeyT, PAIR_, true, SEG_, true>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)src, n, flags, (const uint32_t*)psrc, row_bits); \
                hipLaunchKernelGGL((plan_emit_kernel<KeyT, PAIR_, true, SEG_, true>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)src, \
                                   (const uint32_t*)psrc, (const uint32_t*)flags, n, row_bits, n_tables, order, uniq_keys, \
                                   seg_start, counts, dest, walk, n_walk, pairs_out, n_pairs_out);                        \
            } else {                                                                                                      \
            hipLaunchKernelGGL((plan_count_kernel<KeyT, PAIR_, true, SEG_>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)src, n, flags, (const uint32_t*)psrc, row_bits); \
            hipLaunchKernelGGL((plan_emit_kernel<KeyT, PAIR_, true, SEG_>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)src,     \
                               (const uint32_t*)psrc, (const uint32_t*)flags, n, row_bits, n_tables, order, uniq_keys,     \
                               seg_start, counts, dest, walk, n_walk);                                                    \
            }                                                                                                             \
        } else {                                                                                                          \
            hipLaunchKernelGGL((plan_count_kernel<KeyT, PAIR_, false, SEG_>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)src, n, flags, (const uint32_t*)psrc, row_bits); \
            hipLaunchKernelGGL((plan_emit_kernel<KeyT, PAIR_, false, SEG_>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)src,           \
                               (const uint32_t*)psrc, (const uint32_t*)flags, n, row_bits, n_tables, order, uniq_keys,     \
                               seg_start, counts, (int32_t*)nullptr, (int32_t*)nullptr, (int64_t*)nullptr);               \
        }                                                                                                                 \
    }
        // table + row bits beyond 32 (many tables next to a huge one): the pairs carry the ROW only, the table of a sorted position is the run it
        // lies in (PlaceInfo::seg_off) -- 8 bytes per lookup through every pass instead of 12.  NRX_PLAN_SEGKEY=0: the 64-bit keys.
        const bool segkey = bits > 32 && row_bits <= 32 && !(getenv("NRX_PLAN_SEGKEY") && atoi(getenv("NRX_PLAN_SEGKEY")) == 0);
        sa.segkey = segkey ? 1 : 0;
        for (int t = 0; t <= n_tables; ++t) pinfo.seg_off[t] = sa.seg_off[t];
        pinfo.n_seg = n_tables;
        if (bits <= 32) NRX_SEGSORT(uint32_t, false) else if (segkey) NRX_SEGSORT(uint32_t, true) else NRX_SEGSORT(uint64_t, false)
#undef NRX_SEGSORT
        NRX_LAUNCH_CHECK("nrx_sparse_plan(segmented sort)");
        return NRX_OK;
    }
#define NRX_PLAN(KeyT)                                                                                                    \
    {                                                                                                                     \
        hipLaunchKernelGGL(plan_keys_kernel<KeyT>, dim3((unsigned)g), dim3(NRX_BLOCK), 0, st, a, (KeyT*)keys_in, pay_in);  \
        tb = sort_temp_bytes<KeyT>(n, bits);                                                                              \
        err = rocprim::radix_sort_pairs(temp, tb, (const KeyT*)keys_in, (KeyT*)keys_out, (const uint32_t*)pay_in, pay_out, \
                                        (size_t)n, 0u, (unsigned)bits, st);                                               \
        if (err == hipSuccess) {                                                                                          \
            if (dest != nullptr) {                                                                                        \
                if (want_pairs) {                                                                                         \
                hipLaunchKernelGGL((plan_count_kernel<KeyT, false, true, false, true>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)keys_out, n, flags, (const uint32_t*)pay_out, row_bits); \
                hipLaunchKernelGGL((plan_emit_kernel<KeyT, false, true, false, true>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)keys_out, \
                                   (const uint32_t*)pay_out, (const uint32_t*)flags, n, row_bits, n_tables, order, uniq_keys, \
                                   seg_start, counts, dest, walk, n_walk, pairs_out, n_pairs_out);                        \
                } else {                                                                                                  \
                hipLaunchKernelGGL((plan_count_kernel<KeyT, false, true>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)keys_out, n, flags, (const uint32_t*)pay_out, row_bits); \
                hipLaunchKernelGGL((plan_emit_kernel<KeyT, false, true>), dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)keys_out, \
                                   (const uint32_t*)pay_out, (const uint32_t*)flags, n, row_bits, n_tables, order, uniq_keys, \
                                   seg_start, counts, dest, walk, n_walk);                                                \
                }                                                                                                         \
            } else {                                                                                                      \
                hipLaunchKernelGGL(plan_count_kernel<KeyT>, dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)keys_out, n, flags, (const uint32_t*)pay_out, row_bits); \
                hipLaunchKernelGGL(plan_emit_kernel<KeyT>, dim3(gtile), dim3(NRX_BLOCK), 0, st, pinfo, (const KeyT*)keys_out,      \
                                   (const uint32_t*)pay_out, (const uint32_t*)flags, n, row_bits, n_tables, order, uniq_keys, \
                                   seg_start, counts, (int32_t*)nullptr, (int32_t*)nullptr, (int64_t*)nullptr);           \
            }                                                                                                             \
        }                                                                                                                 \
    }
    if (bits <= 32) NRX_PLAN(uint32_t) else NRX_PLAN(uint64_t)
#undef NRX_PLAN
    if (err != hipSuccess) {
        nrx_set_error("nrx_sparse_plan: rocPRIM call failed: %s", hipGetErrorString(err));
        return NRX_ERR_LAUNCH;
    }
    NRX_LAUNCH_CHECK("nrx_sparse_plan");
    return NRX_OK;
}

extern "C" int nrx_sparse_plan(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                               int32_t n_feats, int32_t index_bits, int32_t n_tables, int64_t* order, int64_t* uniq_keys,
                               int64_t* seg_start, int64_t* counts, void* workspace, void* stream) {
    NRX_TRACE();
    return sparse_plan_impl(ids, lens, table_of, rows, n_feats, index_bits, n_tables, order, uniq_keys, seg_start, counts, 0, nullptr,
                            nullptr, nullptr, workspace, stream);
}

extern "C" int nrx_sparse_plan_place(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                                     int32_t n_feats, int32_t index_bits, int32_t n_tables, uint64_t place_feats, int64_t* order,
                                     int64_t* uniq_keys, int64_t* seg_start, int64_t* counts, int32_t* dest, int32_t* walk,
                                     int64_t* n_walk, void* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(dest != nullptr && walk != nullptr && n_walk != nullptr, "nrx_sparse_plan_place: null placement buffer");
    return sparse_plan_impl(ids, lens, table_of, rows, n_feats, index_bits, n_tables, order, uniq_keys, seg_start, counts, place_feats,
                            dest, walk, n_walk, workspace, stream);
}

namespace {
// {unique rows, walk rows (-1: plan without placement), -1, n, lookups of the padding rows} of a finished plan
__global__ void plan_ex_stats_kernel(const int64_t* __restrict__ uniq, const int64_t* __restrict__ seg_start, const int64_t* __restrict__ counts,
                                     const int64_t* __restrict__ n_walk, int n_tables, int64_t n, int64_t* __restrict__ stats) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int64_t pads = 0;
    for (int t = 0; t < n_tables; ++t) {
        const int64_t first = counts[1 + t], last = counts[2 + t];
        if (first < last && (uniq[first] & ((1ll << 40) - 1)) == 0) pads += seg_start[first + 1] - seg_start[first];
    }
    stats[0] = counts[0];
    stats[1] = n_walk != nullptr ? n_walk[0] : -1;
    stats[2] = -1;
    stats[3] = n;
    stats[4] = pads;
}
}  // namespace

extern "C" int nrx_sparse_plan_ex(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                                  int32_t n_feats, int32_t index_bits, int32_t n_tables, uint64_t place_feats, uint32_t flags, int64_t* order,
                                  int64_t* uniq_keys, int64_t* seg_start, int64_t* counts, int32_t* dest, int32_t* walk,
                                  int64_t* n_walk, int32_t* pairs, int64_t* n_pairs, int64_t* stats, void* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE((dest != nullptr) == (walk != nullptr) && (dest != nullptr) == (n_walk != nullptr),
                "nrx_sparse_plan_ex: dest, walk and n_walk go together (all null: the plan without placement)");
    NRX_REQUIRE((flags & ~(uint32_t)(NRX_PLAN_SPLIT_PADDING | NRX_PLAN_PAIRS | NRX_PLAN_PAYLOAD)) == 0, "nrx_sparse_plan_ex: unknown flag");
    const uint32_t* payload_src = nullptr;
    if (flags & NRX_PLAN_PAYLOAD) {
        NRX_REQUIRE(flags == NRX_PLAN_PAYLOAD && pairs != nullptr && (dest == nullptr || place_feats == 0),
                    "nrx_sparse_plan_ex: NRX_PLAN_PAYLOAD goes alone, with the payload array in `pairs` and nothing placeable (place_feats = 0)");
        payload_src = reinterpret_cast<const uint32_t*>(pairs);
    }
    if (flags & NRX_PLAN_PAIRS) {
        NRX_REQUIRE(dest != nullptr && pairs != nullptr && n_pairs != nullptr && nrx_aligned16(pairs),
                    "nrx_sparse_plan_ex: NRX_PLAN_PAIRS needs the placement outputs, pairs (16-byte aligned) and n_pairs");
        const uint64_t every = n_feats >= 64 ? ~0ull : ((1ull << n_feats) - 1);
        NRX_REQUIRE((place_feats & every) == every, "nrx_sparse_plan_ex: NRX_PLAN_PAIRS needs every feature in place_feats (single-valued features only)");
    }
    const bool wp = (flags & NRX_PLAN_PAIRS) != 0;
    const int rc = sparse_plan_impl(ids, lens, table_of, rows, n_feats, index_bits, n_tables, order, uniq_keys, seg_start, counts,
                                    dest != nullptr ? place_feats : 0, dest, walk, n_walk, workspace, stream, flags & NRX_PLAN_SPLIT_PADDING,
                                    wp ? pairs : nullptr, wp ? n_pairs : nullptr, payload_src);
    if (rc != NRX_OK || stats == nullptr) return rc;
    int64_t n = 0;
    for (int f = 0; f < n_feats; ++f) n += lens[f];
    if (n == 0) return rc;                    // (nothing planned: the caller's statistics stay as they are)
    hipLaunchKernelGGL(plan_ex_stats_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), uniq_keys, seg_start, counts, n_walk,
                       n_tables, n, stats);
    NRX_LAUNCH_CHECK("nrx_sparse_plan_ex(stats)");
    return NRX_OK;
}

// ---------------------------------------------------------------------------------------------------
// Per-destination de-duplicated routing (SURVEY 7 hard part 1b / 8e step 1): the ids of one exchange are grouped by
// (owner, table, local row) with the same bit-limited stable sort as the backward planner; every DISTINCT
// (owner, table, row) is sent once, and slot[] maps every lookup -- duplicates included -- to its unique entry's position
// in the returned-row buffer.  Same block layout as nrx_route_ids with tables in the place of features: inside owner o's
// block the unique rows are ordered by table, then by row, so the owner segments its inbox with counts2d[o][table].
// Pays off on skewed (click-log) ids: a hot row crosses the fabric once per step instead of once per lookup.
// ---------------------------------------------------------------------------------------------------
namespace {

struct DedupArgs {
    const void* ids[NRX_MAX_FEATURES];
    int64_t off[NRX_MAX_FEATURES + 1];
    int64_t local_rows[NRX_MAX_FEATURES];   // rows of the LARGEST local shard of the feature's table (= the out-of-range marker)
    int32_t table_of[NRX_MAX_FEATURES];
    int32_t n_feats;
    int32_t idx64;
    int32_t world;
    int32_t row_bits;
    int32_t table_bits;
    int64_t n_total;
};
static_assert(sizeof(DedupArgs) <= 3584, "kernarg budget");

template <typename KeyT>
__global__ __launch_bounds__(NRX_BLOCK) void dedup_keys_kernel(const DedupArgs args_in_kernarg, KeyT* __restrict__ keys, uint32_t* __restrict__ payload) {
    const NRX_CONST DedupArgs* a = nrx_kernarg<DedupArgs>();
    for (int64_t p = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; p < a->n_total; p += (int64_t)gridDim.x * NRX_BLOCK) {
        int lo = 0, hi = a->n_feats;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a->off[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t i = p - a->off[lo];
        const int64_t id = a->idx64 ? nrx_gconst<int64_t>(a->ids[lo])[i] : (int64_t)nrx_gconst<int32_t>(a->ids[lo])[i];
        uint64_t owner = 0, local = (uint64_t)a->local_rows[lo];           // ids that cannot be rows: rank 0, first row past the shard
        if (id >= 0 && id <= 0x7fffffffLL) {
            const uint32_t u = (uint32_t)id, l = u / (uint32_t)a->world;
            owner = u - l * (uint32_t)a->world;
            local = l < (uint64_t)a->local_rows[lo] ? l : (uint64_t)a->local_rows[lo];
        }
        keys[p] = (KeyT)((owner << (a->table_bits + a->row_bits)) | ((uint64_t)a->table_of[lo] << a->row_bits) | local);
        payload[p] = (uint32_t)p;
    }
}

// global unique rank of every sorted entry (same tile scheme as plan_emit_kernel) + first unique entry of every owner
template <typename KeyT>
__global__ __launch_bounds__(NRX_BLOCK) void dedup_rank_kernel(const KeyT* __restrict__ skeys, const uint32_t* __restrict__ block_heads, int64_t n,
                                                               int owner_shift, int world, uint32_t* __restrict__ urank,
                                                               uint32_t* __restrict__ owner_base /* [world + 1] */,
                                                               int grp_shift = 0, int n_groups = 0, uint32_t* __restrict__ grp_base = nullptr
                                                               /* [n_groups + 1]: first unique entry of every (owner, table) group */) {
    constexpr int ROUNDS = PLAN_TILE / NRX_BLOCK, WAVES = NRX_BLOCK / 64;
    __shared__ uint32_t s_cell[ROUNDS * WAVES + 1];
    __shared__ uint32_t s_part[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t e0 = (int64_t)blockIdx.x * PLAN_TILE + tid;
    KeyT key[ROUNDS], prev[ROUNDS];
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t e = e0 + j * NRX_BLOCK;
        const int64_t ec = e < n ? e : n - 1;
        key[j] = skeys[ec];
        prev[j] = skeys[ec > 0 ? ec - 1 : 0];
    }
    uint32_t acc = 0;
    for (uint32_t i = tid; i < blockIdx.x; i += NRX_BLOCK) acc += block_heads[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) s_part[wid] = acc;
    bool head[ROUNDS];
    unsigned long long mask[ROUNDS];
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t e = e0 + j * NRX_BLOCK;
        head[j] = e < n && (e == 0 || key[j] != prev[j]);
        mask[j] = __ballot(head[j]);
        if (lane == 0) s_cell[j * WAVES + wid] = (uint32_t)__popcll(mask[j]);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        for (int c = 0; c < ROUNDS * WAVES; ++c) {
            const uint32_t v = s_cell[c];
            s_cell[c] = run;
            run += v;
        }
    }
    __syncthreads();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t e = e0 + j * NRX_BLOCK;
        if (e >= n) continue;
        // rank of the entry's unique key = heads before it (+ itself if it is a head) - 1
        const uint32_t before = s_cell[j * WAVES + wid] + (uint32_t)__popcll(mask[j] & lt);
        const uint32_t u = head[j] ? before : before - 1;
        urank[e] = u;
        if (head[j]) {
            const int64_t o = (int64_t)((uint64_t)key[j] >> owner_shift);
            const int64_t oprev = e == 0 ? -1 : (int64_t)((uint64_t)prev[j] >> owner_shift);
            for (int64_t t = oprev + 1; t <= o; ++t) owner_base[t] = u;
            if (grp_base != nullptr) {
                const int64_t g = (int64_t)((uint64_t)key[j] >> grp_shift);
                const int64_t gprev = e == 0 ? -1 : (int64_t)((uint64_t)prev[j] >> grp_shift);
                for (int64_t t = gprev + 1; t <= g; ++t) grp_base[t] = u;
            }
        }
        if (e == n - 1) {
            const int64_t ol = (int64_t)((uint64_t)key[j] >> owner_shift);
            for (int64_t t = ol + 1; t <= world; ++t) owner_base[t] = u + 1;
            if (grp_base != nullptr) {
                const int64_t gl = (int64_t)((uint64_t)key[j] >> grp_shift);
                for (int64_t t = gl + 1; t <= n_groups; ++t) grp_base[t] = u + 1;
            }
        }
    }
}

template <typename KeyT>
__global__ __launch_bounds__(NRX_BLOCK) void dedup_place_kernel(const KeyT* __restrict__ skeys, const uint32_t* __restrict__ spayload,
                                                                const uint32_t* __restrict__ urank, const uint32_t* __restrict__ owner_base,
                                                                int64_t n, int row_bits, int table_bits, int world, int n_tables,
                                                                int64_t cap, int32_t* __restrict__ send_rows, int32_t* __restrict__ slot,
                                                                int64_t* __restrict__ counts2d, int64_t* __restrict__ overflow,
                                                                const uint32_t* __restrict__ grp_base) {
    const int64_t e = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    // per-(owner, table) counts of unique rows = differences of the groups' first unique entries (dedup_rank_kernel).  One atomic per unique
    // row onto world x n_tables counters was 3.6 ms of a 3.7 ms call at the C2 shape (1.66 M heads onto 208 addresses).
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < world * n_tables; i += NRX_BLOCK) {
            const int o = i / n_tables, t = i - o * n_tables, g = (o << table_bits) | t;
            counts2d[i] = (int64_t)grp_base[g + 1] - (int64_t)grp_base[g];
        }
    if (e == 0) {
        int64_t worst = 0;
        for (int o = 0; o < world; ++o) {
            const int64_t c = (int64_t)owner_base[o + 1] - (int64_t)owner_base[o];
            worst = c > worst ? c : worst;
        }
        if (worst > overflow[0]) overflow[0] = worst;          // running maximum since the caller zeroed the word
    }
    if (e >= n) return;
    const uint64_t key = (uint64_t)skeys[e];
    const int o = (int)(key >> (table_bits + row_bits));
    const int t = (int)((key >> row_bits) & ((1ull << table_bits) - 1));
    const int64_t k = (int64_t)urank[e] - (int64_t)owner_base[o];
    const bool head = e == 0 || skeys[e - 1] != skeys[e];
    slot[spayload[e]] = k < cap ? (int32_t)(o * cap + k) : -1;
    if (head && k < cap) send_rows[o * cap + k] = (int32_t)(key & ((1ull << row_bits) - 1));
    (void)t;
}

// np.unique(return_inverse=True) on the device, for the ABI's integer utility: sorted distinct values and, for
// every input element, the index of its value among them
__global__ __launch_bounds__(NRX_BLOCK) void uinv_keys_kernel(const void* __restrict__ ids, int idx64, int64_t n, uint64_t* __restrict__ keys,
                                                              uint32_t* __restrict__ payload) {
    const int64_t p = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    if (p >= n) return;
    const int64_t id = idx64 ? reinterpret_cast<const int64_t*>(ids)[p] : (int64_t)reinterpret_cast<const int32_t*>(ids)[p];
    keys[p] = (uint64_t)id ^ 0x8000000000000000ull;          // order-preserving map of signed to unsigned
    payload[p] = (uint32_t)p;
}

__global__ __launch_bounds__(NRX_BLOCK) void uinv_emit_kernel(const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ spayload,
                                                              const uint32_t* __restrict__ urank, int64_t n, int64_t* __restrict__ unique_out,
                                                              int64_t* __restrict__ inverse_out, int64_t* __restrict__ n_unique) {
    const int64_t e = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    if (e >= n) return;
    const uint32_t u = urank[e];
    inverse_out[spayload[e]] = (int64_t)u;
    if (e == 0 || skeys[e - 1] != skeys[e]) unique_out[u] = (int64_t)(skeys[e] ^ 0x8000000000000000ull);
    if (e == n - 1) n_unique[0] = (int64_t)u + 1;
}

}  // namespace

extern "C" int64_t nrx_route_dedup_workspace(int64_t n_total, int32_t world) {
    if (n_total < 0 || n_total >= 0xffffffffLL || world < 1) return -1;
    const size_t n = (size_t)(n_total > 0 ? n_total : 1);
    const size_t t1 = sort_temp_bytes<uint64_t>(n, 64);
    return (int64_t)(2 * align256(n * 8) + 2 * align256(n * 4) + 2 * align256(n * 4) + align256((size_t)(world + 2) * 4) +
                     align256((((size_t)world << 6) + 2) * 4) + align256(t1) + seg_scratch_bytes(n) + 256);
}

extern "C" int nrx_route_ids_dedup(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* table_local_rows,
                                   int32_t n_feats, int32_t n_tables, int32_t index_bits, int32_t world, int64_t cap,
                                   int32_t* send_rows, int32_t* slot, int64_t* counts2d, int64_t* overflow, void* workspace,
                                   void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(ids && lens && table_of && table_local_rows && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES && n_tables >= 1 &&
                    n_tables <= NRX_MAX_FEATURES, "nrx_route_ids_dedup: bad feature / table count");
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_route_ids_dedup: index_bits must be 32 or 64");
    NRX_REQUIRE(world >= 1 && world <= 64 && cap >= 1 && cap * world <= 0x7fffffffLL, "nrx_route_ids_dedup: bad world / cap");
    NRX_REQUIRE(send_rows && counts2d && overflow && workspace, "nrx_route_ids_dedup: null buffer");
    DedupArgs a;
    int64_t off = 0, max_rows = 1;
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(lens[f] >= 0 && (lens[f] == 0 || ids[f] != nullptr) && table_of[f] >= 0 && table_of[f] < n_tables,
                    "nrx_route_ids_dedup: feature %d: bad entry", f);
        const int64_t lr = table_local_rows[table_of[f]];
        NRX_REQUIRE(lr >= 0 && lr <= 0x7fffffffLL, "nrx_route_ids_dedup: table %d: bad local row count", table_of[f]);
        a.ids[f] = ids[f];
        a.off[f] = off;
        a.local_rows[f] = lr;
        a.table_of[f] = table_of[f];
        off += lens[f];
        if (lr + 1 > max_rows) max_rows = lr + 1;
    }
    a.off[n_feats] = off;
    NRX_REQUIRE(off < 0x7fffffffLL, "nrx_route_ids_dedup: too many ids for one exchange");
    NRX_REQUIRE(slot != nullptr || off == 0, "nrx_route_ids_dedup: null slot buffer");
    a.n_feats = n_feats;
    a.idx64 = index_bits == 64;
    a.world = world;
    a.row_bits = bits_for(max_rows);
    a.table_bits = bits_for(n_tables);
    a.n_total = off;
    const int owner_bits = bits_for(world);
    const int bits = owner_bits + a.table_bits + a.row_bits;
    NRX_REQUIRE(bits <= 62, "nrx_route_ids_dedup: composite key too wide");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = off;
    hipError_t err = nrx_zero_async(counts2d, sizeof(int64_t) * (size_t)world * n_tables, st) == NRX_OK ? hipSuccess : hipErrorLaunchFailure;
    (void)overflow;
    if (err != hipSuccess) {
        nrx_set_error("nrx_route_ids_dedup: memset failed: %s", hipGetErrorString(err));
        return NRX_ERR_LAUNCH;
    }
    if (n == 0) return NRX_OK;
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    char* keys_in = w;                 w += align256((size_t)n * 8);
    char* keys_out = w;                w += align256((size_t)n * 8);
    uint32_t* pay_in = (uint32_t*)w;   w += align256((size_t)n * 4);
    uint32_t* pay_out = (uint32_t*)w;  w += align256((size_t)n * 4);
    uint32_t* heads = (uint32_t*)w;    w += align256((size_t)n * 4);
    uint32_t* urank = (uint32_t*)w;    w += align256((size_t)n * 4);
    uint32_t* obase = (uint32_t*)w;    w += align256((size_t)(world + 2) * 4);
    uint32_t* gbase = (uint32_t*)w;    w += align256((((size_t)world << 6) + 2) * 4);
    const int n_groups = world << a.table_bits;
    void* temp = w;
    int64_t g = (n + NRX_BLOCK - 1) / NRX_BLOCK;
    const unsigned gfull = (unsigned)g, gtile = (unsigned)((n + PLAN_TILE - 1) / PLAN_TILE);
    if (g > 4096) g = 4096;
    size_t tb = 0;
    // the sort: the planner's own tile kernels (seg_sort_generic) -- NRX_PLAN_SORT=rocprim keeps the library sort (A/B, and the reference
    // permutation the tests compare with)
    const char* sort_env = getenv("NRX_PLAN_SORT");
    const bool use_rocprim = sort_env && !strcmp(sort_env, "rocprim");
    void* seg_scratch = reinterpret_cast<char*>(temp) + align256(sort_temp_bytes<uint64_t>(n, 64));
#define NRX_DD(KeyT)                                                                                                       \
    {                                                                                                                      \
        hipLaunchKernelGGL(dedup_keys_kernel<KeyT>, dim3((unsigned)g), dim3(NRX_BLOCK), 0, st, a, (KeyT*)keys_in, pay_in);  \
        KeyT* sk = (KeyT*)keys_out; uint32_t* sp = pay_out;                                                                \
        if (use_rocprim) {                                                                                                 \
            tb = sort_temp_bytes<KeyT>(n, bits);                                                                           \
            err = rocprim::radix_sort_pairs(temp, tb, (const KeyT*)keys_in, (KeyT*)keys_out, (const uint32_t*)pay_in, pay_out, \
                                            (size_t)n, 0u, (unsigned)bits, st);                                            \
        } else {                                                                                                           \
            KeyT* s0 = (KeyT*)keys_in; KeyT* s1 = (KeyT*)keys_out; uint32_t* p0 = pay_in; uint32_t* p1 = pay_out;          \
            seg_sort_generic<KeyT>(s0, s1, p0, p1, n, bits, seg_scratch, st);                                              \
            sk = s0; sp = p0;                                                                                              \
        }                                                                                                                  \
        if (err == hipSuccess) {                                                                                           \
            hipLaunchKernelGGL(plan_count_kernel<KeyT>, dim3(gtile), dim3(NRX_BLOCK), 0, st, PlaceInfo(), (const KeyT*)sk, n, heads, (const uint32_t*)nullptr, 0); \
            hipLaunchKernelGGL(dedup_rank_kernel<KeyT>, dim3(gtile), dim3(NRX_BLOCK), 0, st, (const KeyT*)sk,               \
                               (const uint32_t*)heads, n, a.table_bits + a.row_bits, world, urank, obase, a.row_bits, n_groups, gbase); \
            hipLaunchKernelGGL(dedup_place_kernel<KeyT>, dim3(gfull), dim3(NRX_BLOCK), 0, st, (const KeyT*)sk,              \
                               (const uint32_t*)sp, (const uint32_t*)urank, (const uint32_t*)obase, n, a.row_bits,          \
                               a.table_bits, world, n_tables, cap, send_rows, slot, counts2d, overflow, (const uint32_t*)gbase); \
        }                                                                                                                  \
    }
    if (bits <= 32) NRX_DD(uint32_t) else NRX_DD(uint64_t)
#undef NRX_DD
    if (err != hipSuccess) {
        nrx_set_error("nrx_route_ids_dedup: rocPRIM call failed: %s", hipGetErrorString(err));
        return NRX_ERR_LAUNCH;
    }
    NRX_LAUNCH_CHECK("nrx_route_ids_dedup");
    return NRX_OK;
}

extern "C" int64_t nrx_unique_inverse_workspace(int64_t n) { return nrx_route_dedup_workspace(n, 1); }

extern "C" int nrx_unique_inverse(const void* ids, int32_t index_bits, int64_t n, int64_t* unique_out, int64_t* inverse_out,
                                  int64_t* n_unique, void* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE((index_bits == 32 || index_bits == 64) && n >= 0 && n < 0x7fffffffLL, "nrx_unique_inverse: bad argument");
    NRX_REQUIRE(n_unique != nullptr, "nrx_unique_inverse: null n_unique");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n == 0) {
        if (nrx_zero_async(n_unique, sizeof(int64_t), st) != NRX_OK) return NRX_ERR_LAUNCH;
        return NRX_OK;
    }
    NRX_REQUIRE(ids && unique_out && inverse_out && workspace, "nrx_unique_inverse: null buffer");
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    uint64_t* keys_in = (uint64_t*)w;  w += align256((size_t)n * 8);
    uint64_t* keys_out = (uint64_t*)w; w += align256((size_t)n * 8);
    uint32_t* pay_in = (uint32_t*)w;   w += align256((size_t)n * 4);
    uint32_t* pay_out = (uint32_t*)w;  w += align256((size_t)n * 4);
    uint32_t* heads = (uint32_t*)w;    w += align256((size_t)n * 4);
    uint32_t* urank = (uint32_t*)w;    w += align256((size_t)n * 4);
    uint32_t* obase = (uint32_t*)w;    w += align256((size_t)3 * 4);
    void* temp = w;
    const unsigned gfull = (unsigned)((n + NRX_BLOCK - 1) / NRX_BLOCK), gtile = (unsigned)((n + PLAN_TILE - 1) / PLAN_TILE);
    hipLaunchKernelGGL(uinv_keys_kernel, dim3(gfull), dim3(NRX_BLOCK), 0, st, ids, (int)(index_bits == 64), n, keys_in, pay_in);
    const char* sort_env = getenv("NRX_PLAN_SORT");
    uint64_t* sk = keys_out;
    uint32_t* sp = pay_out;
    if (sort_env && !strcmp(sort_env, "rocprim")) {            // the library sort (A/B knob; the default is the planner's own tile kernels)
        size_t tb = sort_temp_bytes<uint64_t>(n, 64);
        hipError_t err = rocprim::radix_sort_pairs(temp, tb, (const uint64_t*)keys_in, keys_out, (const uint32_t*)pay_in, pay_out, (size_t)n,
                                                   0u, 64u, st);
        if (err != hipSuccess) {
            nrx_set_error("nrx_unique_inverse: rocPRIM call failed: %s", hipGetErrorString(err));
            return NRX_ERR_LAUNCH;
        }
    } else {
        uint64_t* s0 = keys_in; uint64_t* s1 = keys_out; uint32_t* p0 = pay_in; uint32_t* p1 = pay_out;
        seg_sort_generic<uint64_t>(s0, s1, p0, p1, n, 64, reinterpret_cast<char*>(temp) + align256(sort_temp_bytes<uint64_t>(n, 64)), st);
        sk = s0; sp = p0;
    }
    hipLaunchKernelGGL(plan_count_kernel<uint64_t>, dim3(gtile), dim3(NRX_BLOCK), 0, st, PlaceInfo(), (const uint64_t*)sk, n, heads, (const uint32_t*)nullptr, 0);
    hipLaunchKernelGGL(dedup_rank_kernel<uint64_t>, dim3(gtile), dim3(NRX_BLOCK), 0, st, (const uint64_t*)sk, (const uint32_t*)heads, n,
                       64 - 1, 1, urank, obase);      // owner_shift 63: one "owner" (bit 63 may be set: two bases are reserved)
    hipLaunchKernelGGL(uinv_emit_kernel, dim3(gfull), dim3(NRX_BLOCK), 0, st, (const uint64_t*)sk, (const uint32_t*)sp,
                       (const uint32_t*)urank, n, unique_out, inverse_out, n_unique);
    NRX_LAUNCH_CHECK("nrx_unique_inverse");
    return NRX_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fused row-sparse Adam(W) on the unique rows the sorted backward produced.
// Replaces, for the embedding tables only, the reference's dense AdamW over every row of every table
// (configure_optimizers, src/model/sort/deep/model.py:54-65).  Semantics = torch.optim.SparseAdam (moments and
// weights of a row move only in steps that looked the row up; bias correction from the global step) plus an
// optional decoupled weight decay applied to the touched rows; the padding row (row 0) never moves.
// ---------------------------------------------------------------------------------------------------
namespace {

struct SparseAdamArgs {
    float* table[NRX_MAX_FEATURES];
    float* m[NRX_MAX_FEATURES];
    float* v[NRX_MAX_FEATURES];
    const int64_t* keys;        // (table << 40) | row, one per unique row
    const float* grads;         // [n, dim]
    const int64_t* n_dev;       // optional: actual count on the device
    int64_t max_n;
    int32_t n_tables;
    int32_t dim;
    int32_t mom_ld[NRX_MAX_FEATURES];   // per table: floats between consecutive rows of a moment array: dim, or 2 * dim when the row's two moments are adjacent
    float step_size, one_minus_b1, one_minus_b2, eps, decay;
    const float* step_size_dev;   // optional: step size read on the device (graph-captured training loops)
};
static_assert(sizeof(SparseAdamArgs) <= 3584, "kernarg budget");

__device__ __forceinline__ float adam_elem(float g, float& m, float& v, float w, float step_size, const NRX_CONST SparseAdamArgs* a) {
    m = m + (g - m) * a->one_minus_b1;          // torch's update order
    v = v + (g * g - v) * a->one_minus_b2;
    w -= w * a->decay;
    return w - step_size * (m / (sqrtf(v) + a->eps));
}

// VEC: dim % 4 == 0 and 16-byte aligned rows -> one float4 per lane per array; R rows are in flight per lane
// group (key loads, then 4R independent row loads) -- the update is a random 3-array read-modify-write, HBM-bound.
template <int QLOG2, bool VEC>
__global__ __launch_bounds__(NRX_BLOCK) void sparse_adam_kernel(const SparseAdamArgs args_in_kernarg) {
    const NRX_CONST SparseAdamArgs* a = nrx_kernarg<SparseAdamArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    constexpr int R = 4;
    const int q = threadIdx.x & (Q - 1);
    const int D = a->dim;
    const float ss = a->step_size_dev != nullptr ? nrx_gconst<float>(a->step_size_dev)[0] : a->step_size;
    int64_t n = a->max_n;
    if (a->n_dev != nullptr) {
        const int64_t nd = nrx_gconst<int64_t>(a->n_dev)[0];
        n = nd < n ? nd : n;
    }
    const int64_t u0 = ((int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2)) * R;
    if (u0 >= n) return;
    int64_t key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = u0 + r < n ? nrx_gconst<int64_t>(a->keys)[u0 + r] : -1;
    float* p[R];
    float* pm[R];
    float* pv[R];
    bool on[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t t = key[r] >> 40, row = key[r] & ((1ll << 40) - 1);
        on[r] = key[r] >= 0 && row != 0 && t < a->n_tables;        // padding row / filler keys of a merged list
        const int64_t tc = on[r] ? t : 0, rc = on[r] ? row : 0;
        p[r] = a->table[tc] + rc * D;
        pm[r] = a->m[tc] + rc * a->mom_ld[tc];
        pv[r] = a->v[tc] + rc * a->mom_ld[tc];
    }
    if (VEC) {
        for (int k = q * 4; k < D; k += 4 * Q) {
            float4 g[R], w[R], m[R], v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                g[r] = nrx_ldg4(a->grads + (u0 + (on[r] ? r : 0)) * (int64_t)D + k, 0);
                w[r] = nrx_ldg4(p[r] + k, 0);
                m[r] = nrx_ldg4(pm[r] + k, 0);
                v[r] = nrx_ldg4(pv[r] + k, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (on[r]) {
                    float4 o;
                    o.x = adam_elem(g[r].x, m[r].x, v[r].x, w[r].x, ss, a);
                    o.y = adam_elem(g[r].y, m[r].y, v[r].y, w[r].y, ss, a);
                    o.z = adam_elem(g[r].z, m[r].z, v[r].z, w[r].z, ss, a);
                    o.w = adam_elem(g[r].w, m[r].w, v[r].w, w[r].w, ss, a);
                    nrx_stg4(pm[r] + k, 0, m[r]);
                    nrx_stg4(pv[r] + k, 0, v[r]);
                    nrx_stg4(p[r] + k, 0, o);
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!on[r]) continue;
            const float* g = a->grads + (u0 + r) * (int64_t)D;
            for (int k = q; k < D; k += Q) {
                float m = pm[r][k], v = pv[r][k];
                const float o = adam_elem(g[k], m, v, p[r][k], ss, a);
                pm[r][k] = m;
                pv[r][k] = v;
                p[r][k] = o;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// Unique-row gradients -> dense gradient tables (the DEFAULT backward of the gather: nn.Embedding(sparse=False)'s
// [rows, dim] .grad, formed from the deterministic sorted reduction instead of float atomics).
// One lane group per 4 unique rows: key loads, then 4 independent row loads, then the stores.
// --------------------------------------------------------------------------------------------
struct RowsToDenseArgs {
    float* table[NRX_MAX_FEATURES];
    const int64_t* keys;        // (table << 40) | row, one per unique row
    const float* rows;          // [n, dim]
    const int64_t* n_dev;       // optional: actual count on the device
    int64_t max_n;
    int32_t n_tables;
    int32_t dim;
    int32_t accumulate;         // != 0: table row += (a table fed by more than one reduction); 0: plain store
};

template <int QLOG2, bool VEC, bool ACC>
__global__ __launch_bounds__(NRX_BLOCK) void rows_to_dense_kernel(const RowsToDenseArgs args_in_kernarg) {
    const NRX_CONST RowsToDenseArgs* a = nrx_kernarg<RowsToDenseArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    constexpr int R = 4;
    const int q = threadIdx.x & (Q - 1);
    const int D = a->dim;
    int64_t n = a->max_n;
    if (a->n_dev != nullptr) {
        const int64_t nd = nrx_gconst<int64_t>(a->n_dev)[0];
        n = nd < n ? nd : n;
    }
    const int64_t u0 = ((int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2)) * R;
    if (u0 >= n) return;
    int64_t key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = u0 + r < n ? nrx_gconst<int64_t>(a->keys)[u0 + r] : -1;
    float* p[R];
    bool on[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t t = key[r] >> 40, row = key[r] & ((1ll << 40) - 1);
        on[r] = key[r] >= 0 && t < a->n_tables;
        p[r] = a->table[on[r] ? t : 0] + (on[r] ? row : 0) * D;
    }
    if (VEC) {
        for (int k = q * 4; k < D; k += 4 * Q) {
            float4 g[R], w[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                g[r] = nrx_ldg4(a->rows + (u0 + (on[r] ? r : 0)) * (int64_t)D + k, 0);
                if (ACC) w[r] = nrx_ldg4(p[r] + k, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (on[r]) {
                    if (ACC) { g[r].x += w[r].x; g[r].y += w[r].y; g[r].z += w[r].z; g[r].w += w[r].w; }
                    nrx_stg4(p[r] + k, 0, g[r]);
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!on[r]) continue;
            const float* g = a->rows + (u0 + r) * (int64_t)D;
            for (int k = q; k < D; k += Q) p[r][k] = ACC ? p[r][k] + g[k] : g[k];
        }
    }
}

}  // namespace

extern "C" int nrx_rows_to_dense(float* const* tables, int32_t n_tables, int32_t dim, const int64_t* uniq_keys, const float* rows,
                                 int64_t n_unique, const int64_t* n_unique_dev, int32_t accumulate, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(n_tables >= 1 && n_tables <= NRX_MAX_FEATURES && dim >= 1 && n_unique >= 0, "nrx_rows_to_dense: bad argument");
    if (n_unique == 0) return NRX_OK;
    NRX_REQUIRE(tables && uniq_keys && rows, "nrx_rows_to_dense: null buffer");
    RowsToDenseArgs a;
    bool vec = (dim & 3) == 0 && nrx_aligned16(rows);
    for (int t = 0; t < n_tables; ++t) {         // (tables of another width may be listed: the keys of this call never name them)
        NRX_REQUIRE(tables[t] != nullptr, "nrx_rows_to_dense: table %d: null pointer", t);
        a.table[t] = tables[t];
        vec = vec && nrx_aligned16(tables[t]);
    }
    a.keys = uniq_keys;
    a.rows = rows;
    a.n_dev = n_unique_dev;
    a.max_n = n_unique;
    a.n_tables = n_tables;
    a.dim = dim;
    a.accumulate = accumulate;
    int ql = 0;
    while ((4 << ql) < dim && ql < 6) ++ql;
    const int tb = NRX_BLOCK >> ql;
    const int64_t groups = (n_unique + 3) / 4;
    const unsigned grid = (unsigned)((groups + tb - 1) / tb);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define NRX_RD2(QL_, V_) if (accumulate) hipLaunchKernelGGL((rows_to_dense_kernel<QL_, V_, true>), dim3(grid), dim3(NRX_BLOCK), 0, st, a); \
                         else hipLaunchKernelGGL((rows_to_dense_kernel<QL_, V_, false>), dim3(grid), dim3(NRX_BLOCK), 0, st, a)
#define NRX_RD(QL_) if (vec) { NRX_RD2(QL_, true); } else { NRX_RD2(QL_, false); }
    switch (ql) {
        case 0: NRX_RD(0); break; case 1: NRX_RD(1); break; case 2: NRX_RD(2); break; case 3: NRX_RD(3); break;
        case 4: NRX_RD(4); break; case 5: NRX_RD(5); break; default: NRX_RD(6); break;
    }
#undef NRX_RD
#undef NRX_RD2
    NRX_LAUNCH_CHECK("nrx_rows_to_dense");
    return NRX_OK;
}

// ---- exact dense AdamW from row-sparse gradients ("exact-dense mode"): the reference trains every table with one dense
// torch.optim.AdamW (sort/deep/model.py:55) -- EVERY row moves every step (weight decay, decaying moments), so the optimizer is a stream over the
// whole tables however few rows the batch looked up.  This form does that stream ONCE: (p, m, v) of every row read and written, the gradient taken
// from the backward's (key, row) pairs where there is one (a per-table slot map, -1 elsewhere, written by nrx_rows_mark and reset here) and zero
// otherwise -- no dense gradient tensor is formed, zero-filled or read.  torch's single-tensor AdamW arithmetic, fp32.
struct DenseAdamWArgs {
    float* table[NRX_MAX_FEATURES];
    float* m[NRX_MAX_FEATURES];
    float* v[NRX_MAX_FEATURES];
    int32_t* map[NRX_MAX_FEATURES];         // [rows] slot of the row's gradient in `grads`, or -1
    int64_t rows[NRX_MAX_FEATURES];
    int32_t tile0[NRX_MAX_FEATURES + 1];    // first block of every table
    const float* grads;                     // [n, dim]
    int32_t n_tables, dim;
    float decay_keep, one_minus_b1, one_minus_b2, b2, step_size, inv_bc2_sqrt, eps;
    const float* hyper_dev;                 // optional {lr / bias_correction1, 1 / sqrt(bias_correction2)} on the device (captured training loops)
};
static_assert(sizeof(DenseAdamWArgs) <= 3840, "kernarg budget");

__device__ __forceinline__ float adamw_elem(float g, float& m, float& v, float w, const NRX_CONST DenseAdamWArgs* a, float step_size, float inv_bc2_sqrt) {
    w *= a->decay_keep;                                   // param.mul_(1 - lr * weight_decay)
    m = m + (g - m) * a->one_minus_b1;                    // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a->b2 + g * g * a->one_minus_b2;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) * inv_bc2_sqrt + a->eps;
    return w - step_size * (m / denom);                  // param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
}

template <int QLOG2, bool VEC>
__global__ __launch_bounds__(NRX_BLOCK) void dense_adamw_rows_kernel(const DenseAdamWArgs args_in_kernarg) {
    const NRX_CONST DenseAdamWArgs* a = nrx_kernarg<DenseAdamWArgs>();
    constexpr int Q = 1 << QLOG2, TB = NRX_BLOCK / Q, R = 4;
    const int q = threadIdx.x & (Q - 1), g = threadIdx.x >> QLOG2;
    int t = 0;                              // the block's table: tile0[t] <= blockIdx.x < tile0[t + 1] (uniform)
    while (t + 1 < a->n_tables && (int)blockIdx.x >= a->tile0[t + 1]) ++t;
    const int64_t nrows = a->rows[t];
    const int D = a->dim;
    const float ss = a->hyper_dev != nullptr ? nrx_gconst<float>(a->hyper_dev)[0] : a->step_size;
    const float ib = a->hyper_dev != nullptr ? nrx_gconst<float>(a->hyper_dev)[1] : a->inv_bc2_sqrt;
    float* tab = a->table[t];
    float* mt = a->m[t];
    float* vt = a->v[t];
    int32_t* map = a->map[t];
    const int64_t r0 = ((int64_t)((int)blockIdx.x - a->tile0[t]) * TB + g) * R;
    int slot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) slot[r] = r0 + r < nrows ? map[r0 + r] : -1;
    if (VEC) {
        for (int k = q * 4; k < D; k += 4 * Q) {
            float4 gr[R], w[R], m[R], v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = r0 + r < nrows ? r0 + r : 0;
                gr[r] = slot[r] >= 0 ? nrx_ldg4(a->grads + (int64_t)slot[r] * D + k, 0) : make_float4(0.f, 0.f, 0.f, 0.f);
                w[r] = nrx_ldg4(tab + row * D + k, 0);
                m[r] = nrx_ldg4(mt + row * D + k, 0);
                v[r] = nrx_ldg4(vt + row * D + k, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r0 + r >= nrows) continue;
                w[r].x = adamw_elem(gr[r].x, m[r].x, v[r].x, w[r].x, a, ss, ib);
                w[r].y = adamw_elem(gr[r].y, m[r].y, v[r].y, w[r].y, a, ss, ib);
                w[r].z = adamw_elem(gr[r].z, m[r].z, v[r].z, w[r].z, a, ss, ib);
                w[r].w = adamw_elem(gr[r].w, m[r].w, v[r].w, w[r].w, a, ss, ib);
                const int64_t row = r0 + r;
                nrx_stg4(tab + row * D + k, 0, w[r]);
                nrx_stg4(mt + row * D + k, 0, m[r]);
                nrx_stg4(vt + row * D + k, 0, v[r]);
            }
        }
    } else {
        for (int r = 0; r < R; ++r) {
            if (r0 + r >= nrows) continue;
            const int64_t row = r0 + r;
            for (int k = q; k < D; k += Q) {
                const float gk = slot[r] >= 0 ? a->grads[(int64_t)slot[r] * D + k] : 0.f;
                float mk = mt[row * D + k], vk = vt[row * D + k];
                tab[row * D + k] = adamw_elem(gk, mk, vk, tab[row * D + k], a, ss, ib);
                mt[row * D + k] = mk;
                vt[row * D + k] = vk;
            }
        }
    }
    if (q == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (slot[r] >= 0) map[r0 + r] = -1;       // the map is all -1 again when the launch ends
    }
}

struct RowsMarkArgs {
    int32_t* map[NRX_MAX_FEATURES];
    int64_t rows[NRX_MAX_FEATURES];
    const int64_t* keys;
    const int64_t* n_dev;
    int64_t max_n;
    int32_t n_tables;
    int32_t unmark;
};

__global__ __launch_bounds__(NRX_BLOCK) void rows_mark_kernel(const RowsMarkArgs args_in_kernarg) {
    const NRX_CONST RowsMarkArgs* a = nrx_kernarg<RowsMarkArgs>();
    int64_t n = a->max_n;
    if (a->n_dev != nullptr) {
        const int64_t nd = nrx_gconst<int64_t>(a->n_dev)[0];
        n = nd < n ? nd : n;
    }
    const int64_t i = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t key = nrx_gconst<int64_t>(a->keys)[i];
    const int64_t t = key >> 40, row = key & ((1ll << 40) - 1);
    if (key < 0 || t >= a->n_tables || row == 0) return;         // filler keys; the padding row never trains
    if (row < a->rows[t]) a->map[t][row] = a->unmark ? -1 : (int32_t)i;
}

// Two unique-key lists that may name the same row (DSSM's towers share the news table: two backward launches, two sink entries): with list A
// marked in the slot maps, every pair of list B whose row A also has is ADDED to A's row and its key becomes the filler -1; the rest of B stays.
// A row has at most one pair in each list, so every sum is one addition: deterministic.  One lane group per pair of B.
struct RowsMergeArgs {
    int32_t* map[NRX_MAX_FEATURES];
    int64_t rows[NRX_MAX_FEATURES];
    int64_t* keys_b;
    const float* values_b;
    float* values_a;
    const int64_t* n_dev;
    int64_t max_n;
    int32_t n_tables, dim;
};

__global__ __launch_bounds__(NRX_BLOCK) void rows_merge_kernel(const RowsMergeArgs args_in_kernarg) {
    const NRX_CONST RowsMergeArgs* a = nrx_kernarg<RowsMergeArgs>();
    int64_t n = a->max_n;
    if (a->n_dev != nullptr) {
        const int64_t nd = nrx_gconst<int64_t>(a->n_dev)[0];
        n = nd < n ? nd : n;
    }
    const int q = threadIdx.x & 15;                              // 16 lanes per pair
    const int64_t j = ((int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x) >> 4;
    if (j >= n) return;
    const int64_t key = a->keys_b[j];
    const int64_t t = key >> 40, row = key & ((1ll << 40) - 1);
    if (key < 0 || t >= a->n_tables || row == 0 || row >= a->rows[t]) return;
    const int32_t sa = a->map[t][row];
    if (sa < 0) return;
    const int D = a->dim;
    for (int k = q; k < D; k += 16) a->values_a[(int64_t)sa * D + k] += a->values_b[j * D + k];
    if (q == 0) a->keys_b[j] = -1;
}

// slot_maps[t][row] = i for every key i = (t << 40 | row) of the list (negative keys and keys of tables >= n_tables are fillers; row 0 is the
// padding row): the per-table row -> gradient-slot maps nrx_dense_adamw_rows reads.  The maps must be all -1 before (they are again after
// nrx_dense_adamw_rows).  Keys must be unique.  unmark != 0: writes -1 instead (undoes a marking).
extern "C" int nrx_rows_mark(const int64_t* uniq_keys, int64_t n, const int64_t* n_dev, int32_t* const* slot_maps, const int64_t* rows,
                             int32_t n_tables, int32_t unmark, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(n >= 0 && n <= 0x7fffffffll && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES, "nrx_rows_mark: bad argument");
    if (n == 0) return NRX_OK;
    NRX_REQUIRE(uniq_keys && slot_maps && rows, "nrx_rows_mark: null buffer");
    RowsMarkArgs a;
    for (int t = 0; t < n_tables; ++t) {
        NRX_REQUIRE(slot_maps[t] != nullptr && rows[t] >= 1, "nrx_rows_mark: table %d: null map / no rows", t);
        a.map[t] = slot_maps[t];
        a.rows[t] = rows[t];
    }
    a.keys = uniq_keys; a.n_dev = n_dev; a.max_n = n; a.n_tables = n_tables; a.unmark = unmark != 0;
    hipLaunchKernelGGL(rows_mark_kernel, dim3((unsigned)((n + NRX_BLOCK - 1) / NRX_BLOCK)), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), a);
    NRX_LAUNCH_CHECK("nrx_rows_mark");
    return NRX_OK;
}

// keys_b / values_b [n, dim] merged into the list marked in slot_maps (values_a: its [.., dim] rows): see rows_merge_kernel.
extern "C" int nrx_rows_merge(int64_t* keys_b, const float* values_b, int64_t n, const int64_t* n_dev, float* values_a, int32_t* const* slot_maps,
                              const int64_t* rows, int32_t n_tables, int32_t dim, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(n >= 0 && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES && dim >= 1, "nrx_rows_merge: bad argument");
    if (n == 0) return NRX_OK;
    NRX_REQUIRE(keys_b && values_b && values_a && slot_maps && rows, "nrx_rows_merge: null buffer");
    RowsMergeArgs a;
    for (int t = 0; t < n_tables; ++t) {
        NRX_REQUIRE(slot_maps[t] != nullptr && rows[t] >= 1, "nrx_rows_merge: table %d: null map / no rows", t);
        a.map[t] = slot_maps[t];
        a.rows[t] = rows[t];
    }
    a.keys_b = keys_b; a.values_b = values_b; a.values_a = values_a; a.n_dev = n_dev; a.max_n = n; a.n_tables = n_tables; a.dim = dim;
    hipLaunchKernelGGL(rows_merge_kernel, dim3((unsigned)((n * 16 + NRX_BLOCK - 1) / NRX_BLOCK)), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), a);
    NRX_LAUNCH_CHECK("nrx_rows_merge");
    return NRX_OK;
}

// One AdamW step (torch.optim.AdamW's arithmetic: decoupled weight decay, bias corrections of step `step` >= 1) over EVERY row of the n_tables
// [rows[t], dim] tables: the gradient of row r of table t is grads[slot_maps[t][r]] where that slot is >= 0 (nrx_rows_mark), zero elsewhere;
// exp_avg / exp_avg_sq: [rows[t], dim] like the tables (torch's state layout).  Resets every slot it consumed to -1.  hyper_dev (optional, device):
// {lr / bias_correction1, 1 / sqrt(bias_correction2)} read by the kernel instead of the values computed from `step` -- a loop captured in a
// hipGraph advances them between replays.
extern "C" int nrx_dense_adamw_rows(float* const* tables, float* const* exp_avg, float* const* exp_avg_sq, int32_t* const* slot_maps,
                                    const int64_t* rows, int32_t n_tables, int32_t dim, const float* grads, int64_t step, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, const float* hyper_dev, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(n_tables >= 1 && n_tables <= NRX_MAX_FEATURES && dim >= 1 && step >= 1, "nrx_dense_adamw_rows: bad argument");
    NRX_REQUIRE(tables && exp_avg && exp_avg_sq && slot_maps && rows, "nrx_dense_adamw_rows: null buffer");
    DenseAdamWArgs a;
    int ql = 0;
    while ((4 << ql) < dim && ql < 6) ++ql;
    const int tb = (NRX_BLOCK >> ql) * 4;                   // rows per block
    bool vec = (dim & 3) == 0 && (grads == nullptr || nrx_aligned16(grads));
    int64_t blocks = 0;
    for (int t = 0; t < n_tables; ++t) {
        NRX_REQUIRE(tables[t] && exp_avg[t] && exp_avg_sq[t] && slot_maps[t] && rows[t] >= 1, "nrx_dense_adamw_rows: table %d: null pointer / no rows", t);
        a.table[t] = tables[t]; a.m[t] = exp_avg[t]; a.v[t] = exp_avg_sq[t]; a.map[t] = slot_maps[t]; a.rows[t] = rows[t];
        a.tile0[t] = (int32_t)blocks;
        blocks += (rows[t] + tb - 1) / tb;
        vec = vec && nrx_aligned16(tables[t]) && nrx_aligned16(exp_avg[t]) && nrx_aligned16(exp_avg_sq[t]);
    }
    NRX_REQUIRE(blocks <= 0x7fffffffll, "nrx_dense_adamw_rows: too many rows for one launch");
    a.tile0[n_tables] = (int32_t)blocks;
    a.grads = grads;
    a.n_tables = n_tables;
    a.dim = dim;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.decay_keep = 1.0f - lr * weight_decay;
    a.one_minus_b1 = 1.0f - beta1;
    a.one_minus_b2 = 1.0f - beta2;
    a.b2 = beta2;
    a.step_size = (float)((double)lr / bc1);
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    a.eps = eps;
    a.hyper_dev = hyper_dev;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define NRX_DA(QL_) if (vec) hipLaunchKernelGGL((dense_adamw_rows_kernel<QL_, true>), dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, st, a); \
                    else hipLaunchKernelGGL((dense_adamw_rows_kernel<QL_, false>), dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, st, a)
    switch (ql) {
        case 0: NRX_DA(0); break; case 1: NRX_DA(1); break; case 2: NRX_DA(2); break; case 3: NRX_DA(3); break;
        case 4: NRX_DA(4); break; case 5: NRX_DA(5); break; default: NRX_DA(6); break;
    }
#undef NRX_DA
    NRX_LAUNCH_CHECK("nrx_dense_adamw_rows");
    return NRX_OK;
}

extern "C" int nrx_sparse_adam_step(float* const* tables, float* const* exp_avg, float* const* exp_avg_sq, int32_t n_tables,
                                    int32_t dim, const int64_t* uniq_keys, const float* grads, int64_t n_unique,
                                    const int64_t* n_unique_dev, float step_size, const float* step_size_dev, float beta1,
                                    float beta2, float eps, float lr_times_weight_decay, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(n_tables >= 1 && n_tables <= NRX_MAX_FEATURES && dim >= 1 && n_unique >= 0, "nrx_sparse_adam_step: bad argument");
    if (n_unique == 0) return NRX_OK;
    NRX_REQUIRE(tables && exp_avg && exp_avg_sq && uniq_keys && grads, "nrx_sparse_adam_step: null buffer");
    SparseAdamArgs a;
    // Interleaved moments: exp_avg_sq[t] == exp_avg[t] + dim means the two moments of a row of table t sit side by side
    // ([rows, 2, dim]: for dim = 16 ONE 128-byte line per row instead of two half-used ones -- the update is a random
    // read-modify-write and every 64-byte access costs a 128-byte fetch).  Separate arrays can never satisfy this by accident.
    // (The caller may list tables of other widths too -- the keys of this call never name them; their entry is unused.)
    for (int t = 0; t < n_tables; ++t) {
        NRX_REQUIRE(tables[t] && exp_avg[t] && exp_avg_sq[t], "nrx_sparse_adam_step: table %d: null pointer", t);
        a.table[t] = tables[t];
        a.m[t] = exp_avg[t];
        a.v[t] = exp_avg_sq[t];
        a.mom_ld[t] = exp_avg_sq[t] == exp_avg[t] + dim ? 2 * dim : dim;
    }
    a.keys = uniq_keys;
    a.grads = grads;
    a.n_dev = n_unique_dev;
    a.max_n = n_unique;
    a.n_tables = n_tables;
    a.dim = dim;
    a.step_size = step_size;
    a.step_size_dev = step_size_dev;
    a.one_minus_b1 = 1.0f - beta1;
    a.one_minus_b2 = 1.0f - beta2;
    a.eps = eps;
    a.decay = lr_times_weight_decay;
    int ql = 0;
    while ((4 << ql) < dim && ql < 6) ++ql;
    bool vec = (dim & 3) == 0 && nrx_aligned16(grads);
    for (int t = 0; t < n_tables && vec; ++t) vec = nrx_aligned16(tables[t]) && nrx_aligned16(exp_avg[t]) && nrx_aligned16(exp_avg_sq[t]);
    const int tb = NRX_BLOCK >> ql;
    const int64_t groups = (n_unique + 3) / 4;                   // 4 rows per lane group
    const unsigned grid = (unsigned)((groups + tb - 1) / tb);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define NRX_SA(QL_) if (vec) hipLaunchKernelGGL((sparse_adam_kernel<QL_, true>), dim3(grid), dim3(NRX_BLOCK), 0, st, a); \
                    else hipLaunchKernelGGL((sparse_adam_kernel<QL_, false>), dim3(grid), dim3(NRX_BLOCK), 0, st, a)
    switch (ql) {
        case 0: NRX_SA(0); break; case 1: NRX_SA(1); break; case 2: NRX_SA(2); break; case 3: NRX_SA(3); break;
        case 4: NRX_SA(4); break; case 5: NRX_SA(5); break; default: NRX_SA(6); break;
    }
#undef NRX_SA
    NRX_LAUNCH_CHECK("nrx_sparse_adam_step");
    return NRX_OK;
}
