// Per-feature fixed-capacity routing for row-sharded tables (round 6; the reference is single-device -- every trainer is `devices=1`,
// src/model/sort/deep/train.py:38-44 -- so there is no reference counterpart; the definition is oracle/ref_np.py route_feat).
//
// Why another layout.  nrx_route_ids gives every (source, owner) pair ONE block of `cap` slots with the features packed behind each other at
// device-resident offsets: fine for a gather, but nothing on the owner's side is then a plain [batch] id array per feature, and none of the
// single-GPU machinery (the fused forward, the planners, the placement pass, the sorted reduction, the row-sparse optimizer) can be pointed at
// it.  Here every (source, owner, FEATURE) triple gets its own `capf` slots:
//     send_ids[o][f][k]   k-th id of feature f owned by rank o, in sample order, as an OWNER ID: 0 = nothing (an empty slot, or the global
//                         padding id 0), v >= 1 = local row v - 1 of the owner's shard (global row (v - 1) * world + o)
// so that, after one equal-split all-to-all and a transposing copy, the owner holds per feature ONE id array of world * capf pseudo-samples
// [f][s][k] -- a batch like any other for nrx_embed_fwd / nrx_sparse_plan* / nrx_embed_bwd_* over tables that carry a leading dummy row (owner id
// 0 = that row: it reads as zeros and never trains, exactly the padding row of a single-GPU table).  Rows and gradient rows travel as
// [s][k][f][dim] = a [world * capf, n_feats * dim] concat.
//
// ONE launch (nrx_route_ids: four).  A block takes a tile of 4096 ids of one feature (tiles in ticket order: every earlier tile has started),
// ranks its ids per owner with ballots, publishes its per-owner totals and adds up those of the feature's earlier tiles (published words with
// an epoch mark, relaxed agent-scope polling: the construction of nrx_plan_lds.hip's range chain), then places.  The last tile of a feature
// knows the counts: it writes counts[o][f], raises the running overflow maximum, and zero-fills the tails.  No order-dependent atomic: the
// result is the same run to run and equal to its definition bit for bit.
#include <cstring>

#include "nrx_common.h"

namespace {

constexpr int RF_THREADS = 256;
#ifndef NRX_RF_ROUNDS
#define NRX_RF_ROUNDS 16
#endif
constexpr int RF_ROUNDS = NRX_RF_ROUNDS;
constexpr int RF_TILE = RF_THREADS * RF_ROUNDS;      // 4096 ids per block
constexpr int RF_WAVES = RF_THREADS / 64;
constexpr int RF_MAX_WORLD = 64;

struct RouteFeatArgs {
    const void* ids[NRX_MAX_FEATURES];
    int64_t batch;            // ids per feature
    int64_t capf;
    int32_t n_feats, world, idx64, tiles;      // tiles per feature
    int32_t* send_ids;        // [world][n_feats][capf]
    int32_t* send_pos;        // same layout, or null
    int32_t* slot;            // [n_feats][batch]
    int64_t* counts;          // [world][n_feats]
    int64_t* overflow;
    uint32_t* ctl;            // [0] ticket  [1] features done  [2] epoch  [8 + f] tiles of feature f done
    unsigned long long* agg;  // [n_feats * tiles][world]: (mark << 32) | total
    int32_t use_ticket;       // 0: tile = blockIdx.x (every block of the launch is resident: nothing can wait for a block that has not started);
                              // 1: tiles are handed out in start order.  The ticket is ONE address: the memory side serialises device-scope atomics
                              // on it at ~40 ns each -- 416 tickets were 17 of the launch's 33 us (tools/bench_route_bags.py, the timing builds)
};
static_assert(sizeof(RouteFeatArgs) <= 3584, "kernarg budget");

__global__ __launch_bounds__(RF_THREADS) void route_feat_kernel(const RouteFeatArgs args_in_kernarg) {
    const NRX_CONST RouteFeatArgs* a = nrx_kernarg<RouteFeatArgs>();
    __shared__ int s_cell[RF_ROUNDS * RF_WAVES][RF_MAX_WORLD];      // ids of owner o in cell (round, wavefront), then their exclusive prefix
    __shared__ int s_tot[RF_MAX_WORLD], s_base[RF_MAX_WORLD];
    __shared__ uint32_t s_ticket, s_mark;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int W = a->world, n = a->n_feats, T = a->tiles;
    if (W == 1) {
        // one rank: every id stays here, in sample order (k = b): nothing to rank, no chain -- one narrowing pass, as nrx_route_ids has for its layout
        const int f = (int)blockIdx.x / T, tile = (int)blockIdx.x - f * T;
        const int64_t B = a->batch, i0 = (int64_t)tile * RF_TILE, capf = a->capf;
        const void* p = a->ids[f];
#pragma unroll
        for (int j = 0; j < RF_ROUNDS; ++j) {
            const int64_t i = i0 + j * RF_THREADS + tid;
            if (i >= B) continue;
            const int64_t id = a->idx64 ? nrx_gconst<int64_t>(p)[i] : (int64_t)nrx_gconst<int32_t>(p)[i];
            const int32_t v = id < 0 ? -1 : id >= 0x7fffffffLL ? 0x7fffffff : id == 0 ? 0 : (int32_t)(id + 1);
            if (i < capf) {
                a->send_ids[(int64_t)f * capf + i] = v;
                if (a->send_pos != nullptr) a->send_pos[(int64_t)f * capf + i] = (int32_t)i;
                a->slot[(int64_t)f * B + i] = (int32_t)(i * n + f);
            } else {
                a->slot[(int64_t)f * B + i] = -1;
            }
        }
        if (tile == T - 1) {
            if (tid == 0) {
                a->counts[f] = B;
                atomicMax(reinterpret_cast<long long*>(a->overflow), (long long)B);
            }
            for (int64_t k = B + tid; k < capf; k += RF_THREADS) {
                a->send_ids[(int64_t)f * capf + k] = 0;
                if (a->send_pos != nullptr) a->send_pos[(int64_t)f * capf + k] = -1;
            }
        }
        return;
    }
    if (tid == 0) {
        s_ticket = a->use_ticket ? atomicAdd(&a->ctl[0], 1u) : blockIdx.x;
        s_mark = __hip_atomic_load(&a->ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    }
    if (tid < RF_MAX_WORLD) s_base[tid] = 0;
    __syncthreads();
    const int w = (int)s_ticket;
    const uint32_t mark = s_mark;
    const int f = w / T, tile = w - f * T;
    const int64_t B = a->batch, i0 = (int64_t)tile * RF_TILE;
    // ---- the tile's ids -> (owner, owner id)
    int owner[RF_ROUNDS];
    int32_t val[RF_ROUNDS];
    {
        const void* p = a->ids[f];
        int64_t id[RF_ROUNDS];
#pragma unroll
        for (int j = 0; j < RF_ROUNDS; ++j) {
            const int64_t i = i0 + j * RF_THREADS + tid;
            id[j] = -1;
            if (i < B) id[j] = a->idx64 ? nrx_gconst<int64_t>(p)[i] : (int64_t)nrx_gconst<int32_t>(p)[i];
        }
#pragma unroll
        for (int j = 0; j < RF_ROUNDS; ++j) {
            const int64_t i = i0 + j * RF_THREADS + tid;
            if (i >= B) { owner[j] = -1; val[j] = 0; continue; }
            // ids that cannot be rows go to rank 0 as -1 / INT32_MAX: the owner's forward reports them out of range (the reference raises IndexError)
            if (id[j] < 0) { owner[j] = 0; val[j] = -1; }
            else if (id[j] >= 0x7fffffffLL) { owner[j] = 0; val[j] = 0x7fffffff; }
            else {
                const uint32_t u = (uint32_t)id[j], l = u / (uint32_t)W;
                owner[j] = (int)(u - l * (uint32_t)W);
                val[j] = u == 0 ? 0 : (int32_t)(l + 1u);
            }
        }
    }
    // ---- ballot ranks inside (round, wavefront) cells
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank[RF_ROUNDS];
#pragma unroll
    for (int j = 0; j < RF_ROUNDS; ++j) {
        rank[j] = 0;
        for (int t = 0; t < W; ++t) {
            const unsigned long long m = __ballot(owner[j] == t);
            if (owner[j] == t) rank[j] = __popcll(m & lt);
            if (lane == t) s_cell[j * RF_WAVES + wid][t] = __popcll(m);
        }
    }
    __syncthreads();
    // exclusive prefix of every owner's cells and the tile's totals: a wavefront per owner (wavefront w takes owners w, w + 4, ...), lane c = cell c,
    // one shuffle scan -- 64 serial LDS round trips per owner in a single lane were ~3 us of every block's critical path
    static_assert(RF_ROUNDS * RF_WAVES <= 64, "one lane per cell");
    for (int o = wid; o < W; o += RF_WAVES) {
        const int v = lane < RF_ROUNDS * RF_WAVES ? s_cell[lane][o] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane < RF_ROUNDS * RF_WAVES) s_cell[lane][o] = incl - v;
        if (lane == 63) {
            s_tot[o] = incl;
            if (tile + 1 < T)            // (the feature's last tile has no successor)
                __hip_atomic_store(&a->agg[((int64_t)f * T + tile) * W + o], ((unsigned long long)mark << 32) | (unsigned)incl, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- the earlier tiles of this feature: one (tile, owner) word per thread and round, summed per owner with LDS integer atomics
    for (int j = tid; j < tile * W; j += RF_THREADS) {
        const int o = j % W;
        unsigned long long x;
        do { x = __hip_atomic_load(&a->agg[(int64_t)f * T * W + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(x >> 32) != mark);
        atomicAdd(&s_base[o], (int)(uint32_t)x);          // integer sum: the value does not depend on the order
    }
    __syncthreads();
    // ---- place
    const int64_t capf = a->capf;
    int32_t* __restrict__ send_ids = a->send_ids;
    int32_t* __restrict__ send_pos = a->send_pos;
    int32_t* __restrict__ slot = a->slot;
#pragma unroll
    for (int j = 0; j < RF_ROUNDS; ++j) {
        if (owner[j] < 0) continue;
        const int o = owner[j];
        const int64_t i = i0 + j * RF_THREADS + tid;
        const int64_t k = (int64_t)s_base[o] + s_cell[j * RF_WAVES + wid][o] + rank[j];
        if (k < capf) {
            const int64_t d = ((int64_t)o * n + f) * capf + k;
            send_ids[d] = val[j];
            if (send_pos != nullptr) send_pos[d] = (int32_t)i;
            slot[(int64_t)f * B + i] = (int32_t)(((int64_t)o * capf + k) * n + f);
        } else {
            slot[(int64_t)f * B + i] = -1;
        }
    }
    // ---- the feature's last tile: counts, overflow, tails
    if (tile == T - 1) {
        __syncthreads();                 // (s_base is rewritten below: every placement above has read it)
        if (tid < W) {
            const int64_t c = (int64_t)s_base[tid] + s_tot[tid];
            a->counts[(int64_t)tid * n + f] = c;
            s_base[tid] = (int)(c < capf ? c : capf);      // first empty slot of (owner, feature)
        }
        __syncthreads();
        if (tid == 0) {
            long long worst = 0;
            for (int o = 0; o < W; ++o) {
                const long long c = a->counts[(int64_t)o * n + f];
                worst = c > worst ? c : worst;
            }
            atomicMax(reinterpret_cast<long long*>(a->overflow), worst);      // running maximum: order-independent
        }
        for (int o = 0; o < W; ++o) {
            const int64_t base = ((int64_t)o * n + f) * capf;
            for (int64_t k = s_base[o] + tid; k < capf; k += RF_THREADS) {
                send_ids[base + k] = 0;
                if (send_pos != nullptr) send_pos[base + k] = -1;      // (an empty slot: owner id 0 with a position = a lookup of the padding id)
            }
        }
    }
    // ---- re-arm: the block that finishes last advances the epoch and clears the counters (every other block is done polling).  Counted per feature
    // first (T blocks per counter), then once per feature: n x T atomics on one address would serialise the launch's tail
    __syncthreads();
    if (tid == 0) {
        const uint32_t fd = atomicAdd(&a->ctl[8 + f], 1u);
        if (fd == (uint32_t)T - 1u) {
            __hip_atomic_store(&a->ctl[8 + f], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t done = atomicAdd(&a->ctl[1], 1u);
            if (done == (uint32_t)n - 1u) {
                __hip_atomic_store(&a->ctl[2], mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a->ctl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a->ctl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- nrx_route_bags in ONE launch (round 6): the pooled-bag channel's routing -- same outputs, same layout (one block of `cap` slots per owner,
// the features behind each other inside it, lookups with weight 0 not sent) -- by nrx_route_feat's construction with the chain running over ALL
// tiles (tiles are enumerated feature-major, so "the entries of owner o before this tile" IS the position inside o's block): a tile publishes
// its per-owner totals and adds up those of every earlier tile; a feature's last tile adds up the feature's tiles for counts2d; the last tile
// of all raises the running overflow maximum.  (nrx_route_bags: histogram, scan and placement launches: 46 us for C4's 3.3 M lookups.)
struct RouteBagsArgs {
    const void* ids[NRX_MAX_FEATURES];
    const float* weight[NRX_MAX_FEATURES];
    int32_t bag_len[NRX_MAX_FEATURES];
    uint64_t magic_len[NRX_MAX_FEATURES];    // floor(2^64 / bag_len) + 1: sample = umul64hi(entry, magic), exact for entries < 2^32 (a 64-bit division per
                                             // entry was most of this launch's instruction count)
    uint64_t magic_world;                    // the same for id / world
    int32_t use_ticket;                      // as RouteFeatArgs::use_ticket
    int32_t tile0[NRX_MAX_FEATURES + 1];     // first tile of each feature
    int64_t batch, cap;
    int32_t n_feats, world, idx64, tiles;
    int32_t* send_rows;
    int32_t* send_tag;
    float* send_w;
    int64_t* counts2d;                       // [world][n_feats]
    int64_t* overflow;
    uint32_t* ctl;
    unsigned long long* agg;                 // [tiles][world]: a tile's totals
    unsigned long long* gagg;                // [tiles / RB_GROUP][world]: the totals of a whole group of RB_GROUP tiles (published by the group's last tile)
    // the RUNS form (nrx_route_bags_runs): tiles of WHOLE samples, the weights normalised here, run bounds instead of tags
    int32_t tile_len[NRX_MAX_FEATURES];      // entries per tile: RF_TILE, or (RUNS) spt * bag_len with spt = RF_TILE / bag_len samples
    int32_t kind[NRX_MAX_FEATURES];          // (RUNS) NRX_BAG_*: weight[] holds the RAW mask
    float* inv_out[NRX_MAX_FEATURES];        // (RUNS, optional) [batch]: 1 / den of every sample (0 for an empty masked bag)
    int32_t* send_run;                       // (RUNS) [world][n_feats * batch][2]: first and one-past-last slot of the (owner, tag) run inside o's block
};
static_assert(sizeof(RouteBagsArgs) <= 3584, "kernarg budget");
constexpr int RB_GROUP = 32;                 // tiles per group of the two-level chain: a tile adds up <= 31 tile totals + tiles / 32 group totals,
                                             // not every earlier tile's (800 tiles on C4: 320 k uncached polls per launch, the launch was no faster than
                                             // the three it replaced)

// RUNS (nrx_route_bags_runs, round 6): the two launches either side of the routing are folded in.  A tile holds WHOLE samples (spt = 4096 / L of
// them), so (1) the normalised weights are formed here from the raw mask -- nrx_bag_norm_weights' arithmetic and summation order, the sample sums
// in LDS -- and (2) the entries of one (sample, owner) are contiguous inside the tile's part of o's block: the tile writes where that RUN starts
// and ends (send_run) instead of a tag per entry, which is what the owner's pooling launch needs (it spent a launch of its own, pool_mark_kernel
// over every entry behind a memset, finding the runs from the tags).
template <bool RUNS>
__global__ __launch_bounds__(RF_THREADS) __attribute__((amdgpu_waves_per_eu(4)))      // (four blocks per compute unit: C4's ~810 tiles are all resident)
void route_bags_one_kernel(const RouteBagsArgs args_in_kernarg) {
    const NRX_CONST RouteBagsArgs* a = nrx_kernarg<RouteBagsArgs>();
    __shared__ int s_cell[RF_ROUNDS * RF_WAVES][RF_MAX_WORLD];
    __shared__ __attribute__((aligned(16))) uint32_t s_aux[RUNS ? RF_TILE : 1];      // (RUNS) the sample sums (float), then the local sample of every
                                                                                   // entry in owner order (uint16)
    __shared__ int s_pref[RF_MAX_WORLD];
    __shared__ int s_tot[RF_MAX_WORLD], s_base[RF_MAX_WORLD], s_feat[RF_MAX_WORLD];
    __shared__ uint32_t s_ticket, s_mark;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int W = a->world, n = a->n_feats, T = a->tiles;
    if (tid == 0) {
        s_ticket = a->use_ticket ? atomicAdd(&a->ctl[0], 1u) : blockIdx.x;
        s_mark = __hip_atomic_load(&a->ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    }
    if (tid < RF_MAX_WORLD) { s_base[tid] = 0; s_feat[tid] = 0; }
    __syncthreads();
    const int w = (int)s_ticket;
    const uint32_t mark = s_mark;
    int f = 0;
    for (int i = 1; i < n; ++i) f += w >= a->tile0[i] ? 1 : 0;
    const int tile = w - a->tile0[f];
    const int TL = RUNS ? a->tile_len[f] : RF_TILE;
    const int64_t i0 = (int64_t)tile * TL;
    const int64_t len_f = a->batch * a->bag_len[f];
    const int64_t len = RUNS ? (len_f < i0 + TL ? len_f : i0 + TL) : len_f;      // (RUNS: the tile's own end)
    const int L_ = a->bag_len[f];
    const int64_t smp0 = RUNS ? i0 / L_ : 0;                                // the tile's first sample
    const int ns = RUNS ? (int)((len - i0) / L_) : 0;                       // ... and how many it holds
    // (RUNS) the sample of the thread's j-th entry inside the tile: (j * 256 + tid) / L by a float multiply -- exact: the quotient's rounding error
    // (< 4096 / L * 2^-22) is a thousandth of the half-step the + 0.5 keeps it away from an integer (a 64-bit multiply-high per entry and pass was
    // a third of this launch's instructions)
    const float inv_l = 1.0f / (float)L_;
    auto local_sample = [&](int j) -> int { return (int)(((float)(j * RF_THREADS + tid) + 0.5f) * inv_l); };
    if constexpr (RUNS) {
        // the (owner, tag) pairs of this tile's samples without an entry keep the empty run 0..0: written first, the real bounds below overwrite them
        // (same block, behind barriers)
        const int64_t ntag = (int64_t)n * a->batch, t0 = (int64_t)f * a->batch + smp0;
        for (int j = tid; j < W * ns; j += RF_THREADS) {
            const int o = j / ns, b = j - o * ns;
            reinterpret_cast<int2*>(a->send_run)[(int64_t)o * ntag + t0 + b] = make_int2(0, 0);
        }
    }
    int owner[RF_ROUNDS];
    int32_t val[RF_ROUNDS];
    float wt[RF_ROUNDS];
    {
        const void* p = a->ids[f];
        const float* wp = a->weight[f];
        int64_t id[RF_ROUNDS];
#pragma unroll
        for (int j = 0; j < RF_ROUNDS; ++j) {
            const int64_t i = i0 + j * RF_THREADS + tid;
            id[j] = 0;
            wt[j] = 1.0f;
            if (i < len) {
                id[j] = a->idx64 ? nrx_gconst<int64_t>(p)[i] : (int64_t)nrx_gconst<int32_t>(p)[i];
                if (wp != nullptr && !(RUNS && a->kind[f] == NRX_BAG_MEAN)) wt[j] = nrx_gconst<float>(wp)[i];
            }
        }
        if constexpr (RUNS) {
            // the weight that travels = m / den, as nrx_bag_norm_weights writes it.  den of a masked mean: the tile's raw masks go to LDS (read from
            // memory once), lane groups add each sample's up in nrx_bag_norm_weights' order (16 lanes per sample up to 64 entries, else a
            // wavefront: the same partial sums, the same tree) and leave the sum IN PLACE, in the sample's first word.
            float* s_m = reinterpret_cast<float*>(s_aux);
            const int kind = a->kind[f];
            if (kind == NRX_BAG_MASKED_MEAN) {
#pragma unroll
                for (int j = 0; j < RF_ROUNDS; ++j) s_m[j * RF_THREADS + tid] = wt[j];          // (past the tile's end: never read)
                __syncthreads();
                const bool narrow = L_ <= 64;
                const int gl = narrow ? 16 : 64, q = tid & (gl - 1);
                for (int b0 = 0; b0 < ns; b0 += RF_THREADS / gl) {                  // (ns, b0: block-uniform)
                    const int b = b0 + tid / gl;
                    float part = 0.f;
                    if (b < ns)
                        for (int l = q; l < L_; l += gl) part += s_m[b * L_ + l];
                    if (narrow) {
                        part += nrx_dpp<0xB1>(part);
                        part += nrx_dpp<0x4E>(part);
                        part += nrx_dpp<0x141>(part);
                        part += nrx_dpp<0x140>(part);
                    } else {
                        part = nrx_wave_sum(part);
                    }
                    if (b < ns && q == 0) s_m[b * L_] = part + 1e-8f;               // (every lane of the group has read its words: the DPP steps are behind the loads)
                }
                __syncthreads();
            }
            const float den_fixed = kind == NRX_BAG_MEAN ? (float)L_ : 1.0f;
#pragma unroll
            for (int j = 0; j < RF_ROUNDS; ++j) {
                const int64_t i = i0 + j * RF_THREADS + tid;
                if (i < len) {
                    const int b = local_sample(j);
                    const float den = kind == NRX_BAG_MASKED_MEAN ? s_m[b * L_] : den_fixed;
                    wt[j] = wt[j] / den;
                    if (a->inv_out[f] != nullptr && (int64_t)(smp0 + b) * L_ == i)          // the sample's first entry writes 1 / den
                        a->inv_out[f][smp0 + b] = (kind == NRX_BAG_MASKED_MEAN && den <= 1e-8f) ? 0.f : 1.0f / den;
                }
            }
            __syncthreads();         // (s_aux is reused below)
        }
#pragma unroll
        for (int j = 0; j < RF_ROUNDS; ++j) {
            const int64_t i = i0 + j * RF_THREADS + tid;
            if (i >= len || wt[j] == 0.f) { owner[j] = -1; val[j] = 0; continue; }      // a lookup with weight 0 is not sent at all
            if (id[j] < 0) { owner[j] = 0; val[j] = -1; }
            else if (id[j] > 0x7fffffffLL) { owner[j] = 0; val[j] = 0x7fffffff; }
            else {
                const uint32_t u = (uint32_t)id[j], l = W == 1 ? u : (uint32_t)__umul64hi((uint64_t)u, a->magic_world);
                owner[j] = (int)(u - l * (uint32_t)W);
                val[j] = (int32_t)l;
            }
        }
    }
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank[RF_ROUNDS];
#pragma unroll
    for (int j = 0; j < RF_ROUNDS; ++j) {
        rank[j] = 0;
        for (int t = 0; t < W; ++t) {
            const unsigned long long m = __ballot(owner[j] == t);
            if (owner[j] == t) rank[j] = __popcll(m & lt);
            if (lane == t) s_cell[j * RF_WAVES + wid][t] = __popcll(m);
        }
    }
    __syncthreads();
    for (int o = wid; o < W; o += RF_WAVES) {
        const int v = lane < RF_ROUNDS * RF_WAVES ? s_cell[lane][o] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane < RF_ROUNDS * RF_WAVES) s_cell[lane][o] = incl - v;
        if (lane == 63) {
            s_tot[o] = incl;
            __hip_atomic_store(&a->agg[(int64_t)w * W + o], ((unsigned long long)mark << 32) | (unsigned)incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the entries of every owner before tile x (all features: the block layout packs them behind each other) = the totals of the whole groups
    // before x's group + the totals of the tiles of x's group before x; one word per thread and round, summed with LDS integer atomics.
    // Two phases, so that a group's total never waits for an earlier group's: (A) the tiles of the own group -> the group's last tile publishes
    // the group total; (B) the earlier groups' totals.
    // RUNS: which entries open / close a (sample, owner) run -- found HERE, from the places inside the tile alone, while the chain's totals arrive:
    // every entry leaves its local sample at its place in the tile's owner-major order (LDS); an entry opens its run when the entry before it
    // there is another sample's (or there is none: tiles hold whole samples), and closes it when the next one is.
    uint32_t opn = 0, cls = 0;
    if constexpr (RUNS) {
        uint16_t* s_smp = reinterpret_cast<uint16_t*>(s_aux);
        __syncthreads();                         // (s_cell / s_tot of every owner)
        if (tid == 0) {
            int run = 0;
            for (int o = 0; o < W; ++o) { s_pref[o] = run; run += s_tot[o]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RF_ROUNDS; ++j) rank[j] += owner[j] >= 0 ? s_cell[j * RF_WAVES + wid][owner[j]] : 0;      // -> the place among the tile's entries of the owner
#pragma unroll
        for (int j = 0; j < RF_ROUNDS; ++j) {
            if (owner[j] < 0) continue;
            s_smp[s_pref[owner[j]] + rank[j]] = (uint16_t)local_sample(j);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RF_ROUNDS; ++j) {
            if (owner[j] < 0) continue;
            const int o = owner[j];
            const int at = s_pref[o] + rank[j];
            const uint16_t me = (uint16_t)local_sample(j);
            if (rank[j] == 0 || s_smp[at - 1] != me) opn |= 1u << j;
            if (rank[j] == s_tot[o] - 1 || s_smp[at + 1] != me) cls |= 1u << j;
        }
    }
    auto poll = [&](const unsigned long long* p) -> int {
        unsigned long long v;
        do { v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(v >> 32) != mark);
        return (int)(uint32_t)v;
    };
    auto tiles_before_in_group = [&](int x, int* dst) {
        const int g0 = x / RB_GROUP, nt = (x - g0 * RB_GROUP) * W;
        for (int j = tid; j < nt; j += RF_THREADS) atomicAdd(&dst[j % W], poll(&a->agg[(int64_t)g0 * RB_GROUP * W + j]));
    };
    auto groups_before = [&](int x, int* dst) {
        const int ng = (x / RB_GROUP) * W;
        for (int j = tid; j < ng; j += RF_THREADS) atomicAdd(&dst[j % W], poll(&a->gagg[j]));
    };
    const bool last_of_feat = w + 1 == a->tile0[f + 1];
#ifndef NRX_RB_NO_CHAIN          // (dev timing builds leave the chain out: wrong positions, the launch's time without its inter-block waits)
    tiles_before_in_group(w, s_base);
    __syncthreads();
    if ((w % RB_GROUP) == RB_GROUP - 1 && tid < W)
        __hip_atomic_store(&a->gagg[(int64_t)(w / RB_GROUP) * W + tid], ((unsigned long long)mark << 32) | (unsigned)(s_base[tid] + s_tot[tid]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    groups_before(w, s_base);
    if (last_of_feat) {                                          // (where the feature began: counts2d = where it ends - where it began)
        tiles_before_in_group(a->tile0[f], s_feat);
        groups_before(a->tile0[f], s_feat);
    }
#endif
    __syncthreads();
    const int64_t cap = a->cap, B = a->batch;
    const int L = a->bag_len[f];
    const uint64_t magic_l = a->magic_len[f];
    const int32_t tag0 = (int32_t)(f * B);
    const int64_t ntag_r = (int64_t)n * B;
#pragma unroll
    for (int j = 0; j < RF_ROUNDS; ++j) {
        if (owner[j] < 0) continue;
        const int o = owner[j];
        const int64_t i = i0 + j * RF_THREADS + tid;
        const int pt = RUNS ? rank[j] : s_cell[j * RF_WAVES + wid][o] + rank[j];            // place among the tile's entries of owner o
        const int64_t k = (int64_t)s_base[o] + pt;
        const int32_t smp = RUNS ? (int32_t)smp0 + local_sample(j) : (L == 1 ? (int32_t)i : (int32_t)__umul64hi((uint64_t)i, magic_l));
#ifdef NRX_RB_NO_STORE
        if (k < 0) {
#else
        if (k < cap) {
#endif
            const int64_t d = (int64_t)o * cap + k;
            a->send_rows[d] = val[j];
            if (!RUNS) a->send_tag[d] = tag0 + smp;
            a->send_w[d] = wt[j];
        }
        if (RUNS && ((opn | cls) >> j) & 1u) {      // the bounds of the (sample, owner) run this entry opens / closes (clamped to the block)
            int32_t* r = a->send_run + ((int64_t)o * ntag_r + tag0 + smp) * 2;
            if ((opn >> j) & 1u) r[0] = (int32_t)(k < cap ? k : cap);
            if ((cls >> j) & 1u) r[1] = (int32_t)(k + 1 < cap ? k + 1 : cap);
        }
    }
    if (last_of_feat && tid < W) a->counts2d[(int64_t)tid * n + f] = (int64_t)s_base[tid] + s_tot[tid] - s_feat[tid];
    if (w == T - 1 && tid == 0) {
        long long worst = 0;
        for (int o = 0; o < W; ++o) {
            const long long c = (long long)s_base[o] + s_tot[o];
            worst = c > worst ? c : worst;
        }
        atomicMax(reinterpret_cast<long long*>(a->overflow), worst);
    }
    // ---- re-arm: the LAST tile has, by then, read a word of every group and of every tile of its own group: every block of the launch has read the
    // epoch (it published with it) and taken its ticket -- no count of finished blocks is needed (T atomics on one address would serialise the tail)
    __syncthreads();
    if (w == T - 1 && tid == 0) {
        __hip_atomic_store(&a->ctl[2], mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a->ctl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// [world][n][capf] -> [n][world][capf] (the owner's per-feature id arrays), up to two arrays in one launch
__global__ __launch_bounds__(NRX_BLOCK) void inbox_transpose_kernel(const int32_t* __restrict__ a_in, int32_t* __restrict__ a_out,
                                                                   const int32_t* __restrict__ b_in, int32_t* __restrict__ b_out, int world, int n,
                                                                   int64_t capf) {
    const int64_t total = (int64_t)world * n * capf;
    for (int64_t i = ((int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * NRX_BLOCK * 4) {
        // capf % 4 == 0: a 16-byte piece never straddles two (source, feature) runs
        const int64_t run = i / capf, k = i - run * capf;
        const int s = (int)(run / n), f = (int)(run - (int64_t)s * n);
        const int64_t o = ((int64_t)f * world + s) * capf + k;
        typedef int nrx_i32x4t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<nrx_i32x4t*>(a_out + o) = *reinterpret_cast<const nrx_i32x4t*>(a_in + i);
        if (b_in != nullptr) *reinterpret_cast<nrx_i32x4t*>(b_out + o) = *reinterpret_cast<const nrx_i32x4t*>(b_in + i);
    }
}

// One-sided placement in the per-feature layout: the owner's gather writes every row straight into its place in the REQUESTER's concat buffer
// (peer[s] + pos * ld + col[f]) instead of into a [world * capf, n * dim] row buffer that an all-to-all carries back and a final launch re-reads.
// A lane group of Q lanes owns a pseudo-sample b' = s * capf + k and walks the features with GP_R row loads in flight; owner id 0 with a
// position >= 0 is a lookup of the padding id: zeros are written (position < 0: an empty slot, skipped).
#ifndef NRX_GP_R
#define NRX_GP_R 8
#endif
constexpr int GP_R = NRX_GP_R;
struct GatherPlaceArgs {
    const float* table[NRX_MAX_FEATURES];      // arena base (row 0 = the dummy row)
    int64_t rows[NRX_MAX_FEATURES];            // arena rows
    int32_t col[NRX_MAX_FEATURES];
    float* peer[RF_MAX_WORLD];
    const int32_t* oid;                        // [n][bp]
    const int32_t* opos;                       // [n][bp]
    int64_t bp, capf, ld, out_rows;
    int32_t n, world;
    int32_t* status;
};
static_assert(sizeof(GatherPlaceArgs) <= 3584, "kernarg budget");

template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void gather_place_feat_kernel(const GatherPlaceArgs args_in_kernarg) {
    const NRX_CONST GatherPlaceArgs* a = nrx_kernarg<GatherPlaceArgs>();
    constexpr int Q = 1 << QLOG2, TB = NRX_BLOCK / Q;
    const int q = threadIdx.x & (Q - 1);
    const int64_t bp = a->bp;
    const int n = a->n;
    for (int64_t b = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2); b < bp; b += (int64_t)gridDim.x * TB) {
        const int s = (int)(b / a->capf);
        float* __restrict__ out = a->peer[s];
        // the next eight features' {owner id, position} are requested BEFORE the current eight's rows are waited for (two register sets, the loop body
        // written twice: a chunk was two dependent round trips -- ids, then rows -- and a 26-feature sample four chunks)
        auto fetch = [&](int f0, int32_t* id, int32_t* pos) {
#pragma unroll
            for (int r = 0; r < GP_R; ++r) {
                const int f = f0 + r < n ? f0 + r : n - 1;
                id[r] = nrx_gconst<int32_t>(a->oid)[(int64_t)f * bp + b];
                pos[r] = nrx_gconst<int32_t>(a->opos)[(int64_t)f * bp + b];
            }
        };
        auto process = [&](int f0, const int32_t* id, int32_t* pos) {
            float4 v[GP_R];
#pragma unroll
            for (int r = 0; r < GP_R; ++r) {
                const int f = f0 + r < n ? f0 + r : n - 1;
                v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f0 + r >= n || pos[r] < 0) { pos[r] = -1; continue; }
                if (id[r] < 0 || (int64_t)id[r] >= a->rows[f]) {          // cannot be a row of this shard: zeros, reported
                    if (q == 0) nrx_report_oob(a->status, f, b, id[r]);
                    continue;
                }
                if (id[r] != 0) v[r] = nrx_ldg4_nt(a->table[f], (int64_t)id[r] * Q + q);
            }
#pragma unroll
            for (int r = 0; r < GP_R; ++r) {
                if (pos[r] < 0) continue;
                const int f = f0 + r;
                if ((int64_t)pos[r] >= a->out_rows) {                    // a position outside the requester's batch (it came from a peer): dropped, reported
                    if (q == 0) nrx_report_oob(a->status, f, b, pos[r]);
                    continue;
                }
                nrx_stg4(out, ((int64_t)pos[r] * a->ld + a->col[f]) / 4 + q, v[r]);
            }
        };
        int32_t ida[GP_R], posa[GP_R], idb[GP_R], posb[GP_R];
        fetch(0, ida, posa);
        for (int f0 = 0; f0 < n; f0 += 2 * GP_R) {
            if (f0 + GP_R < n) fetch(f0 + GP_R, idb, posb);
            process(f0, ida, posa);
            if (f0 + GP_R >= n) break;
            if (f0 + 2 * GP_R < n) fetch(f0 + 2 * GP_R, ida, posa);
            process(f0 + GP_R, idb, posb);
        }
    }
}

// Where does the gradient row of lookup p = (f, b) go?  The owner's plan says (dest_req: the plan's dest[] of the owners, brought back by an all-to-all
// in send_ids' layout [o][f][k]): to values[u] of owner o when its row is looked up once in the whole exchange (u = the unique index), else to
// this rank's block of the owner's receive buffer [source][k][f].  Both live in ONE arena per owner -- values rows first, the receive buffer from row
// `recv_row0` on -- so the answer is one number: (owner << shift) | row of that arena.  -1: the lookup was dropped (an overflowed block).
__global__ __launch_bounds__(NRX_BLOCK) void shard_dest_combine_kernel(const int32_t* __restrict__ slot, const int32_t* __restrict__ dest_req, int n, int64_t batch,
                                                                      int64_t capf, int64_t recv_row0, int rank, int shift, int32_t* __restrict__ out) {
    // grid.y = the feature; the slot's two divisions in 32 bits (slot < 2^31, n <= 64, capf < 2^31): three 64-bit divisions per lookup were most
    // of this launch's instructions (C2: 8.8 us for 1.7 M lookups)
    const int f = blockIdx.y;
    const uint32_t un = (uint32_t)n, ucapf = (uint32_t)capf;
    for (int64_t b = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; b < batch; b += (int64_t)gridDim.x * NRX_BLOCK) {
        const int64_t p = (int64_t)f * batch + b;
        const int32_t sl = slot[p];
        int32_t r = -1;
        if (sl >= 0) {
            const uint32_t ok = (uint32_t)sl / un, o = ok / ucapf, k = ok - o * ucapf;
            const int32_t d = dest_req[((int64_t)o * n + f) * capf + k];
            const int64_t row = d >= 0 ? (int64_t)d : recv_row0 + ((int64_t)rank * capf + k) * n + f;
            r = (int32_t)(((int64_t)o << shift) | row);
        }
        out[p] = r;
    }
}

// May the launch take tile = blockIdx.x (no ticket)?  When every one of its blocks fits the device at once: then no block can wait for one that
// never gets a slot.  (Blocks are dispatched in index order per XCD -- block i to XCD i % 8 -- so even with part of the device held by another
// stream the smallest unstarted tile's XCD only runs smaller tiles, which do not wait for it; the ticket path needs no such argument and serves
// the larger launches.)  Occupancy is asked of the runtime once per kernel.
bool all_resident(const void* kernel, int64_t blocks) {
    static thread_local const void* cached_k[4] = {nullptr, nullptr, nullptr, nullptr};
    static thread_local int64_t cached_cap[4] = {0, 0, 0, 0};
    int64_t capacity = -1;
    for (int i = 0; i < 4; ++i)
        if (cached_k[i] == kernel) capacity = cached_cap[i];
    if (capacity < 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RF_THREADS, 0) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return false;
        capacity = (int64_t)per_cu * cus;
        for (int i = 0; i < 4; ++i)
            if (cached_k[i] == nullptr) { cached_k[i] = kernel; cached_cap[i] = capacity; break; }
    }
    return blocks <= capacity;
}

}      // namespace

extern "C" int nrx_shard_dest_combine(const int32_t* slot, const int32_t* dest_req, int32_t n_feats, int64_t batch, int64_t capf, int64_t recv_row0,
                                      int32_t rank, int32_t world, int32_t shift, int32_t* dest_out, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(slot && dest_req && dest_out && n_feats >= 1 && batch >= 0 && capf >= 1 && rank >= 0 && rank < world, "nrx_shard_dest_combine: bad argument");
    NRX_REQUIRE(shift >= 1 && shift <= 30 && (int64_t)world <= (1ll << (31 - shift)) && recv_row0 + (int64_t)world * capf * n_feats <= (1ll << shift),
                "nrx_shard_dest_combine: (owner << shift) | row must fit 31 bits");
    if (batch == 0) return NRX_OK;
    NRX_REQUIRE(capf < 0x7fffffffLL && n_feats <= NRX_MAX_FEATURES, "nrx_shard_dest_combine: capf must fit 31 bits, n_feats <= %d", NRX_MAX_FEATURES);
    int64_t blocks = (batch + NRX_BLOCK - 1) / NRX_BLOCK;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(shard_dest_combine_kernel, dim3((unsigned)blocks, (unsigned)n_feats), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), slot, dest_req, (int)n_feats,
                       batch, capf, recv_row0, (int)rank, (int)shift, dest_out);
    NRX_LAUNCH_CHECK("nrx_shard_dest_combine");
    return NRX_OK;
}

extern "C" int nrx_gather_place_feat(const float* const* tables, const int64_t* table_rows, const int32_t* feat_col, int32_t n_feats, int32_t world,
                                     int64_t capf, const int32_t* owner_ids, const int32_t* owner_pos, int32_t dim, float* const* peer_out,
                                     int64_t out_ld, int64_t out_rows, int32_t* status, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(tables && table_rows && feat_col && owner_ids && owner_pos && peer_out, "nrx_gather_place_feat: null argument");
    NRX_REQUIRE(n_feats >= 1 && n_feats <= NRX_MAX_FEATURES && world >= 1 && world <= RF_MAX_WORLD && capf >= 1, "nrx_gather_place_feat: bad sizes");
    if (!(dim == 16 || dim == 32 || dim == 64 || dim == 128 || dim == 256) || (out_ld & 3) != 0) {
        nrx_set_error("nrx_gather_place_feat: dim must be 16 / 32 / 64 / 128 / 256 and out_ld a multiple of 4");
        return NRX_ERR_UNSUPPORTED;
    }
    GatherPlaceArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n_feats; ++i) {
        NRX_REQUIRE(tables[i] != nullptr && nrx_aligned16(tables[i]) && (feat_col[i] & 3) == 0, "nrx_gather_place_feat: feature %d: aligned table and column", i);
        a.table[i] = tables[i];
        a.rows[i] = table_rows[i];
        a.col[i] = feat_col[i];
    }
    for (int s = 0; s < world; ++s) {
        NRX_REQUIRE(peer_out[s] != nullptr && nrx_aligned16(peer_out[s]), "nrx_gather_place_feat: peer %d: null / unaligned buffer", s);
        a.peer[s] = peer_out[s];
    }
    a.oid = owner_ids;
    a.opos = owner_pos;
    a.bp = (int64_t)world * capf;
    a.capf = capf;
    a.ld = out_ld;
    a.out_rows = out_rows;
    a.n = n_feats;
    a.world = world;
    a.status = status;
    const int ql = dim == 16 ? 2 : dim == 32 ? 3 : dim == 64 ? 4 : dim == 128 ? 5 : 6;
    const int64_t tb = NRX_BLOCK >> ql;
    int64_t blocks = (a.bp + tb - 1) / tb;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (ql == 2) hipLaunchKernelGGL((gather_place_feat_kernel<2>), dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, st, a);
    else if (ql == 3) hipLaunchKernelGGL((gather_place_feat_kernel<3>), dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, st, a);
    else if (ql == 4) hipLaunchKernelGGL((gather_place_feat_kernel<4>), dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, st, a);
    else if (ql == 5) hipLaunchKernelGGL((gather_place_feat_kernel<5>), dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((gather_place_feat_kernel<6>), dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, st, a);
    NRX_LAUNCH_CHECK("nrx_gather_place_feat");
    return NRX_OK;
}

extern "C" int64_t nrx_route_feat_state_bytes(int32_t n_feats, int64_t batch, int32_t world) {
    if (n_feats < 1 || n_feats > NRX_MAX_FEATURES || batch < 0 || world < 1 || world > RF_MAX_WORLD) return -1;
    const int64_t tiles = (batch + RF_TILE - 1) / RF_TILE;
    return 512 + (int64_t)n_feats * (tiles > 0 ? tiles : 1) * world * 8;
}

extern "C" int nrx_route_feat(const void* const* ids, int32_t n_feats, int64_t batch, int32_t index_bits, int32_t world, int64_t capf,
                              int32_t* send_ids, int32_t* send_pos, int32_t* slot, int64_t* counts, int64_t* overflow, void* state, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(ids != nullptr && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "nrx_route_feat: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(world >= 1 && world <= RF_MAX_WORLD, "nrx_route_feat: world must be in [1, %d]", RF_MAX_WORLD);
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_route_feat: index_bits must be 32 or 64");
    NRX_REQUIRE(batch >= 0 && batch < 0x7fffffffLL && capf >= 1, "nrx_route_feat: bad batch / capf");
    NRX_REQUIRE((int64_t)world * capf * n_feats < 0x7fffffffLL, "nrx_route_feat: world * capf * n_feats must stay below 2^31 (slot is int32)");
    NRX_REQUIRE(send_ids && slot && counts && overflow && state, "nrx_route_feat: null buffer");
    if (batch == 0) return NRX_OK;
    RouteFeatArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n_feats; ++i) {
        NRX_REQUIRE(ids[i] != nullptr, "nrx_route_feat: feature %d: null ids", i);
        a.ids[i] = ids[i];
    }
    a.batch = batch;
    a.capf = capf;
    a.n_feats = n_feats;
    a.world = world;
    a.idx64 = index_bits == 64;
    a.tiles = (int32_t)((batch + RF_TILE - 1) / RF_TILE);
    a.send_ids = send_ids;
    a.send_pos = send_pos;
    a.slot = slot;
    a.counts = counts;
    a.overflow = overflow;
    a.ctl = reinterpret_cast<uint32_t*>(state);
    a.agg = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(state) + 512);
    a.use_ticket = world > 1 && !all_resident(reinterpret_cast<const void*>(route_feat_kernel), (int64_t)n_feats * a.tiles);
    hipLaunchKernelGGL(route_feat_kernel, dim3((unsigned)(n_feats * a.tiles)), dim3(RF_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
    NRX_LAUNCH_CHECK("nrx_route_feat");
    return NRX_OK;
}

extern "C" int64_t nrx_route_bags_one_state_bytes(const int32_t* bag_lens, int32_t n_feats, int64_t batch, int32_t world) {
    if (bag_lens == nullptr || n_feats < 1 || n_feats > NRX_MAX_FEATURES || batch < 0 || world < 1 || world > RF_MAX_WORLD) return -1;
    int64_t tiles = 0;
    for (int f = 0; f < n_feats; ++f) tiles += (batch * bag_lens[f] + RF_TILE - 1) / RF_TILE;
    return 64 + ((tiles > 0 ? tiles : 1) + tiles / RB_GROUP + 1) * world * 8;
}

extern "C" int nrx_route_bags_one(const void* const* ids, const float* const* weights, const int32_t* bag_lens, int32_t n_feats, int32_t index_bits,
                                  int64_t batch, int32_t world, int64_t cap, int32_t* send_rows, int32_t* send_tag, float* send_w, int64_t* counts2d,
                                  int64_t* overflow, void* state, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(ids && bag_lens && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "nrx_route_bags_one: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_route_bags_one: index_bits must be 32 or 64");
    NRX_REQUIRE(world >= 1 && world <= RF_MAX_WORLD && cap >= 1 && cap * world <= 0x7fffffffLL && batch >= 0, "nrx_route_bags_one: bad world / cap / batch");
    NRX_REQUIRE((int64_t)n_feats * batch <= 0x7fffffffLL, "nrx_route_bags_one: n_feats * batch must fit 31 bits");
    NRX_REQUIRE(send_rows && send_tag && send_w && counts2d && overflow && state, "nrx_route_bags_one: null buffer");
    RouteBagsArgs a;
    memset(&a, 0, sizeof(a));
    int64_t tiles = 0, total = 0;
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(bag_lens[f] >= 1 && (batch == 0 || ids[f] != nullptr), "nrx_route_bags_one: feature %d: bad ids / bag_len", f);
        a.ids[f] = ids[f];
        a.weight[f] = weights ? weights[f] : nullptr;
        a.bag_len[f] = bag_lens[f];
        a.magic_len[f] = bag_lens[f] > 1 ? ~0ull / (uint64_t)bag_lens[f] + 1 : 0;
        a.tile0[f] = (int32_t)tiles;
        tiles += (batch * bag_lens[f] + RF_TILE - 1) / RF_TILE;
        total += batch * bag_lens[f];
    }
    a.tile0[n_feats] = (int32_t)tiles;
    a.magic_world = world > 1 ? ~0ull / (uint64_t)world + 1 : 0;
    NRX_REQUIRE(total <= 0x7fffffffLL && tiles < (1 << 24), "nrx_route_bags_one: too many ids for one exchange");
    if (tiles == 0) return NRX_OK;
    a.batch = batch;
    a.cap = cap;
    a.n_feats = n_feats;
    a.world = world;
    a.idx64 = index_bits == 64;
    a.tiles = (int32_t)tiles;
    a.send_rows = send_rows;
    a.send_tag = send_tag;
    a.send_w = send_w;
    a.counts2d = counts2d;
    a.overflow = overflow;
    a.ctl = reinterpret_cast<uint32_t*>(state);
    a.agg = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(state) + 64);
    a.gagg = a.agg + tiles * world;
    a.use_ticket = !all_resident(reinterpret_cast<const void*>(route_bags_one_kernel<false>), tiles);
    hipLaunchKernelGGL(route_bags_one_kernel<false>, dim3((unsigned)tiles), dim3(RF_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
    NRX_LAUNCH_CHECK("nrx_route_bags_one");
    return NRX_OK;
}

// tiles of whole samples: spt = RF_TILE / L samples (the form needs L <= RF_TILE and spt * world run words per tile <= RF_TILE)
static bool route_bags_runs_ok(const int32_t* bag_lens, int32_t n_feats, int32_t world) {
    for (int f = 0; f < n_feats; ++f)
        if (bag_lens[f] < 1 || bag_lens[f] > RF_TILE || (int64_t)(RF_TILE / bag_lens[f]) * world > RF_TILE) return false;
    return true;
}

extern "C" int64_t nrx_route_bags_runs_state_bytes(const int32_t* bag_lens, int32_t n_feats, int64_t batch, int32_t world) {
    if (bag_lens == nullptr || n_feats < 1 || n_feats > NRX_MAX_FEATURES || batch < 0 || world < 1 || world > RF_MAX_WORLD) return -1;
    if (!route_bags_runs_ok(bag_lens, n_feats, world)) return 0;             // not this form's shape: nrx_route_bags_one + nrx_pool_inbox_fwd
    int64_t tiles = 0;
    for (int f = 0; f < n_feats; ++f) {
        const int64_t spt = RF_TILE / bag_lens[f];
        tiles += (batch + spt - 1) / spt;
    }
    return 64 + ((tiles > 0 ? tiles : 1) + tiles / RB_GROUP + 1) * world * 8;
}

extern "C" int nrx_route_bags_runs(const void* const* ids, const float* const* masks, const int32_t* kinds, const int32_t* bag_lens, int32_t n_feats,
                                   int32_t index_bits, int64_t batch, int32_t world, int64_t cap, int32_t* send_rows, float* send_w, int32_t* send_run,
                                   float* const* inv_out, int64_t* counts2d, int64_t* overflow, void* state, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(ids && bag_lens && kinds && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "nrx_route_bags_runs: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_route_bags_runs: index_bits must be 32 or 64");
    NRX_REQUIRE(world >= 1 && world <= RF_MAX_WORLD && cap >= 1 && cap * world <= 0x7fffffffLL && batch >= 0, "nrx_route_bags_runs: bad world / cap / batch");
    NRX_REQUIRE((int64_t)n_feats * batch * world * 2 <= 0x7fffffffLL, "nrx_route_bags_runs: world * n_feats * batch run words must fit 31 bits");
    NRX_REQUIRE(send_rows && send_w && send_run && counts2d && overflow && state, "nrx_route_bags_runs: null buffer");
    NRX_REQUIRE(route_bags_runs_ok(bag_lens, n_feats, world), "nrx_route_bags_runs: not this form's shape (nrx_route_bags_runs_state_bytes returned 0)");
    RouteBagsArgs a;
    memset(&a, 0, sizeof(a));
    int64_t tiles = 0, total = 0;
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(batch == 0 || ids[f] != nullptr, "nrx_route_bags_runs: feature %d: null ids", f);
        NRX_REQUIRE(kinds[f] == NRX_BAG_SUM || kinds[f] == NRX_BAG_MEAN || kinds[f] == NRX_BAG_MASKED_MEAN, "nrx_route_bags_runs: feature %d: kind must be a bag kind", f);
        NRX_REQUIRE(kinds[f] != NRX_BAG_MASKED_MEAN || (masks && masks[f]) || batch == 0, "nrx_route_bags_runs: feature %d: a masked mean needs its mask", f);
        const int64_t spt = RF_TILE / bag_lens[f];
        a.ids[f] = ids[f];
        a.weight[f] = masks ? masks[f] : nullptr;
        a.kind[f] = kinds[f];
        a.inv_out[f] = inv_out ? inv_out[f] : nullptr;
        a.bag_len[f] = bag_lens[f];
        a.tile_len[f] = (int32_t)(spt * bag_lens[f]);
        a.magic_len[f] = bag_lens[f] > 1 ? ~0ull / (uint64_t)bag_lens[f] + 1 : 0;
        a.tile0[f] = (int32_t)tiles;
        tiles += (batch + spt - 1) / spt;
        total += batch * bag_lens[f];
    }
    a.tile0[n_feats] = (int32_t)tiles;
    a.magic_world = world > 1 ? ~0ull / (uint64_t)world + 1 : 0;
    NRX_REQUIRE(total <= 0x7fffffffLL && tiles < (1 << 24), "nrx_route_bags_runs: too many ids for one exchange");
    if (tiles == 0) return NRX_OK;
    a.batch = batch;
    a.cap = cap;
    a.n_feats = n_feats;
    a.world = world;
    a.idx64 = index_bits == 64;
    a.tiles = (int32_t)tiles;
    a.send_rows = send_rows;
    a.send_w = send_w;
    a.send_run = send_run;
    a.counts2d = counts2d;
    a.overflow = overflow;
    a.ctl = reinterpret_cast<uint32_t*>(state);
    a.agg = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(state) + 64);
    a.gagg = a.agg + tiles * world;
    a.use_ticket = !all_resident(reinterpret_cast<const void*>(route_bags_one_kernel<true>), tiles);
    hipLaunchKernelGGL(route_bags_one_kernel<true>, dim3((unsigned)tiles), dim3(RF_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
    NRX_LAUNCH_CHECK("nrx_route_bags_runs");
    return NRX_OK;
}

extern "C" int nrx_inbox_transpose(const int32_t* inbox_a, int32_t* out_a, const int32_t* inbox_b, int32_t* out_b, int32_t world, int32_t n_feats,
                                   int64_t capf, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(inbox_a != nullptr && out_a != nullptr && (inbox_b == nullptr) == (out_b == nullptr), "nrx_inbox_transpose: null buffer");
    NRX_REQUIRE(world >= 1 && n_feats >= 1 && capf >= 4 && (capf & 3) == 0, "nrx_inbox_transpose: capf must be a positive multiple of 4");
    NRX_REQUIRE(nrx_aligned16(inbox_a) && nrx_aligned16(out_a) && nrx_aligned16(inbox_b) && nrx_aligned16(out_b), "nrx_inbox_transpose: 16-byte aligned buffers");
    const int64_t total = (int64_t)world * n_feats * capf;
    int64_t blocks = (total / 4 + NRX_BLOCK - 1) / NRX_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(inbox_transpose_kernel, dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), inbox_a, out_a, inbox_b,
                       out_b, (int)world, (int)n_feats, capf);
    NRX_LAUNCH_CHECK("nrx_inbox_transpose");
    return NRX_OK;
}
