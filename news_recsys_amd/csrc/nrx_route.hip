// Fixed-capacity, host-sync-free routing for row-sharded tables (row r on rank r % world at local
// row r / world).  New in this build (the reference is single-device); definitions of the results are
// in oracle/ref_np.py (route_ids / gather_inbox).  Everything is deterministic: order comes from
// ballot ranks and prefix sums; the only atomics are integer counters whose final value is
// order-independent.
//
//   route_hist   per 2048-id chunk of one feature (one block): ids of each owner   -> hist[o][chunk]
//                and per-(owner, feature) totals                                   -> counts2d
//   route_scan   one block: exclusive scan of hist over chunks per owner, largest block -> overflow
//   route_place  per chunk: stable rank of each id inside its owner block -> send_rows (int32 local rows), slot (int32)
//   route_single world == 1: one narrowing pass
//   inbox_*      owner side: walk the [world, cap] inbox; a slot's feature (hence table) comes from
//                the prefix sums of the device-resident recv2d row, staged in LDS per block.
#include "nrx_common.h"

namespace {

// ------------------------------------------------------------------------------- source side
// A 256-thread block owns one CHUNK of 2048 ids of ONE feature (chunks are enumerated feature-major, so the chunk
// order IS the source order): the feature -- hence the id array -- is block-uniform, ids are read coalesced
// (position p = chunk start + j * 256 + thread), and the stable rank of an id inside its owner block is
//     (ids of that owner in earlier chunks)  +  (in earlier (round j, wavefront) cells of this chunk)  +  (in lower lanes)
// = scanned histogram + LDS cell prefix + ballot rank.  On the wire a row is an int32 local row (tables have < 2^31 rows).
constexpr int CHUNK = 2048;
constexpr int ROUNDS = CHUNK / NRX_BLOCK;     // 8 ids per thread
constexpr int RWAVES = NRX_BLOCK / 64;

struct RouteArgs {
    const void* ids[NRX_MAX_FEATURES];
    int64_t len[NRX_MAX_FEATURES];
    int64_t off[NRX_MAX_FEATURES + 1];        // flat start of each feature
    int32_t chunk0[NRX_MAX_FEATURES + 1];     // first chunk of each feature
    int32_t n_feats;
    int32_t world;
    int32_t idx64;
    int32_t nchunks;
    int64_t cap;
    int32_t* hist;       // [world][nchunks]  (owner-major: the scan reads it coalesced)
    int64_t* counts2d;   // [world][n_feats]
    int32_t* send_rows;
    int32_t* slot;
    int64_t* overflow;
    // pooled-bag channel (nrx_route_bags): lookups carry the sample they pool into and their normalised weight;
    // zero-weight lookups are not sent at all.  tag = tag0[f] + i / bag_len[f]  (i = position inside feature f)
    const float* weight[NRX_MAX_FEATURES];
    int32_t bag_len[NRX_MAX_FEATURES];
    int32_t tag0[NRX_MAX_FEATURES];
    int32_t* send_tag;
    float* send_w;
    int32_t bags;
    int32_t* send_pos;   // optional (one-sided placement): the position of every sent id inside its feature (= the sample, for [B] features)
};
static_assert(sizeof(RouteArgs) <= 3840, "kernarg budget");

__device__ __forceinline__ int chunk_feature(const NRX_CONST RouteArgs* a, int chunk) {
    int lo = 0, hi = a->n_feats;               // last f with chunk0[f] <= chunk (features without ids own no chunk)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a->chunk0[mid] <= chunk) lo = mid; else hi = mid;
    }
    return lo;
}

// owner and wire form of an id: ids outside [0, 2^31) cannot be rows of any table -- they go to rank 0 as -1 / INT32_MAX
// and are reported there as out of range (the reference raises IndexError for them too)
__device__ __forceinline__ void route_split(int64_t id, int world, int& owner, int32_t& local) {
    if (id < 0) { owner = 0; local = -1; return; }
    if (id > 0x7fffffffLL) { owner = 0; local = 0x7fffffff; return; }
    const uint32_t u = (uint32_t)id;
    const uint32_t l = u / (uint32_t)world;
    owner = (int)(u - l * (uint32_t)world);
    local = (int32_t)l;
}

struct ChunkIds {
    int owner[ROUNDS];
    int32_t local[ROUNDS];
    float w[ROUNDS];
};

__device__ __forceinline__ void load_chunk(const NRX_CONST RouteArgs* a, int f, int64_t i0, int64_t len, int tid, ChunkIds& c) {
    int64_t id[ROUNDS];
    const void* p = a->ids[f];
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t i = i0 + j * NRX_BLOCK + tid;
        id[j] = 0;
        if (i < len) id[j] = a->idx64 ? nrx_gconst<int64_t>(p)[i] : (int64_t)nrx_gconst<int32_t>(p)[i];
    }
    const bool bags = a->bags != 0;
    const float* wp = bags ? a->weight[f] : nullptr;
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        const int64_t i = i0 + j * NRX_BLOCK + tid;
        c.w[j] = 1.0f;
        if (wp != nullptr && i < len) c.w[j] = nrx_gconst<float>(wp)[i];
    }
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        route_split(id[j], a->world, c.owner[j], c.local[j]);
        if (i0 + j * NRX_BLOCK + tid >= len || (bags && c.w[j] == 0.f)) c.owner[j] = -1;
    }
}

__global__ __launch_bounds__(NRX_BLOCK) void route_hist(const RouteArgs args_in_kernarg) {
    const NRX_CONST RouteArgs* a = nrx_kernarg<RouteArgs>();
    __shared__ int s_cnt[64];
    const int tid = threadIdx.x, chunk = blockIdx.x;
    const int f = chunk_feature(a, chunk);
    const int64_t i0 = (int64_t)(chunk - a->chunk0[f]) * CHUNK;
    if (tid < 64) s_cnt[tid] = 0;
    __syncthreads();
    ChunkIds c;
    load_chunk(a, f, i0, a->len[f], tid, c);
    const int lane = tid & 63;
    int cnt = 0;                                                      // lane o: ids of owner o seen by this wavefront
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j)
        for (int t = 0; t < a->world; ++t) {
            const unsigned long long m = __ballot(c.owner[j] == t);
            if (lane == t) cnt += __popcll(m);
        }
    if (lane < a->world && cnt) atomicAdd(&s_cnt[lane], cnt);         // integer counts: the final value is order-independent
    __syncthreads();
    // (the per-(owner, feature) totals come out of route_scan: one device atomic per block on the same few addresses
    // serialised -- 1 600 blocks took 32 us for 26 MB of ids)
    if (tid < a->world) a->hist[(int64_t)tid * a->nchunks + chunk] = s_cnt[tid];
}

// One wavefront per owner: exclusive scan of hist[o][0..nchunks) in place (256 chunks per step, coalesced); the largest
// owner total -> overflow.
// Chunks are feature-uniform, so the ids owner o receives of feature f are a difference of the scanned histogram at the
// feature's chunk range: counts2d[o][f] = excl[chunk0[f + 1]] - excl[chunk0[f]].
__global__ __launch_bounds__(NRX_BLOCK) void route_scan(const RouteArgs args_in_kernarg) {
    const NRX_CONST RouteArgs* a = nrx_kernarg<RouteArgs>();
    int32_t* __restrict__ hist = a->hist;
    const int nchunks = a->nchunks, world = a->world;
    int64_t* __restrict__ overflow = a->overflow;
    __shared__ int s_tot[64];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    for (int o = wid; o < world; o += NRX_BLOCK / 64) {
        int32_t* h = hist + (int64_t)o * nchunks;
        int running = 0;
        for (int c0 = 0; c0 < nchunks; c0 += 256) {
            const int c = c0 + lane * 4;
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (c + k < nchunks) ? h[c + k] : 0;
            const int mine = v[0] + v[1] + v[2] + v[3];
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            int base = running + incl - mine;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (c + k < nchunks) h[c + k] = base;
                base += v[k];
            }
            running += __shfl(incl, 63, 64);
        }
        if (lane == 0) s_tot[o] = running;
        __threadfence();                                   // this wavefront's scanned values, read back below by other lanes
        for (int f = lane; f < a->n_feats; f += 64) {
            const int c0 = a->chunk0[f], c1 = a->chunk0[f + 1];
            const int lo = c0 < nchunks ? h[c0] : running;
            const int hi = c1 < nchunks ? h[c1] : running;
            a->counts2d[o * a->n_feats + f] = (int64_t)(hi - lo);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int worst = 0;
        for (int o = 0; o < world; ++o)
            if (s_tot[o] > worst) worst = s_tot[o];
        if ((int64_t)worst > overflow[0]) overflow[0] = worst;     // running maximum since the caller zeroed the word
    }
}

__global__ __launch_bounds__(NRX_BLOCK) void route_place(const RouteArgs args_in_kernarg) {
    const NRX_CONST RouteArgs* a = nrx_kernarg<RouteArgs>();
    __shared__ int s_cell[ROUNDS * RWAVES][64];          // ids of owner o in cell (round j, wavefront w), then their prefix
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, chunk = blockIdx.x;
    const int world = a->world;
    const int f = chunk_feature(a, chunk);
    const int64_t i0 = (int64_t)(chunk - a->chunk0[f]) * CHUNK;
    const int64_t len = a->len[f];
    ChunkIds c;
    load_chunk(a, f, i0, len, tid, c);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank[ROUNDS];
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        rank[j] = 0;
        for (int t = 0; t < world; ++t) {
            const unsigned long long m = __ballot(c.owner[j] == t);
            if (c.owner[j] == t) rank[j] = __popcll(m & lt);
            if (lane == t) s_cell[j * RWAVES + wid][t] = __popcll(m);
        }
    }
    __syncthreads();
    if (tid < world) {                                   // lane o: exclusive prefix of owner o's cells, seeded with earlier chunks
        int run = a->hist[(int64_t)tid * a->nchunks + chunk];
        for (int cidx = 0; cidx < ROUNDS * RWAVES; ++cidx) {
            const int v = s_cell[cidx][tid];
            s_cell[cidx][tid] = run;
            run += v;
        }
    }
    __syncthreads();
    const int64_t cap = a->cap;
    const int64_t p0 = a->off[f] + i0;
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        if (c.owner[j] < 0) continue;
        const int64_t p = p0 + j * NRX_BLOCK + tid;
        const int64_t k = s_cell[j * RWAVES + wid][c.owner[j]] + rank[j];
        if (a->bags) {
            if (k < cap) {
                const int64_t d = c.owner[j] * cap + k;
                a->send_rows[d] = c.local[j];
                a->send_tag[d] = a->tag0[f] + (int32_t)((i0 + j * NRX_BLOCK + tid) / a->bag_len[f]);
                a->send_w[d] = c.w[j];
            }
        } else if (k < cap) {
            a->slot[p] = (int32_t)(c.owner[j] * cap + k);
            a->send_rows[c.owner[j] * cap + k] = c.local[j];
            if (a->send_pos != nullptr) a->send_pos[c.owner[j] * cap + k] = (int32_t)(i0 + j * NRX_BLOCK + tid);
        } else {
            a->slot[p] = -1;
        }
    }
}

// world == 1: every id stays here, in source order -- one pass, nothing to rank
__global__ __launch_bounds__(NRX_BLOCK) void route_single(const RouteArgs args_in_kernarg) {
    const NRX_CONST RouteArgs* a = nrx_kernarg<RouteArgs>();
    const int tid = threadIdx.x, chunk = blockIdx.x;
    if (chunk == 0) {
        if (tid < a->n_feats) a->counts2d[tid] = a->len[tid];
        if (tid == 0 && a->off[a->n_feats] > a->overflow[0]) a->overflow[0] = a->off[a->n_feats];
    }
    if (chunk >= a->nchunks) return;
    const int f = chunk_feature(a, chunk);
    const int64_t i0 = (int64_t)(chunk - a->chunk0[f]) * CHUNK;
    ChunkIds c;
    load_chunk(a, f, i0, a->len[f], tid, c);
    const int64_t p0 = a->off[f] + i0;
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) {
        if (c.owner[j] < 0) continue;
        const int64_t p = p0 + j * NRX_BLOCK + tid;
        if (p < a->cap) {
            a->slot[p] = (int32_t)p;
            a->send_rows[p] = c.local[j];
            if (a->send_pos != nullptr) a->send_pos[p] = (int32_t)(i0 + j * NRX_BLOCK + tid);
        } else {
            a->slot[p] = -1;
        }
    }
}

// composite (table, row) keys for the sorted backward
struct KeyArgs {
    const void* ids[NRX_MAX_FEATURES];
    int64_t off[NRX_MAX_FEATURES + 1];
    int32_t table_of[NRX_MAX_FEATURES];
    int32_t n_feats;
    int32_t idx64;
    int64_t n_total;
    int64_t* keys;
};

__global__ __launch_bounds__(NRX_BLOCK) void make_keys_kernel(const KeyArgs args_in_kernarg) {
    const NRX_CONST KeyArgs* a = nrx_kernarg<KeyArgs>();
    for (int64_t p = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; p < a->n_total; p += (int64_t)gridDim.x * NRX_BLOCK) {
        int lo = 0, hi = a->n_feats;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a->off[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t i = p - a->off[lo];
        int64_t id = a->idx64 ? nrx_gconst<int64_t>(a->ids[lo])[i] : (int64_t)nrx_gconst<int32_t>(a->ids[lo])[i];
        if (id < 0) id = 0;
        a->keys[p] = ((int64_t)a->table_of[lo] << 40) | (id & ((1ll << 40) - 1));
    }
}

// ------------------------------------------------------------------------------- owner side
struct InboxArgs {
    float* table[NRX_MAX_FEATURES];        // weight tables (gather) or grad tables (scatter)
    int64_t rows[NRX_MAX_FEATURES];
    int32_t feat_table[NRX_MAX_FEATURES];
    int32_t n_feats;
    int32_t world;
    int64_t cap;
    const int64_t* recv2d;
    const int32_t* inbox;
    float* buf;                            // out_rows (gather) / g_rows (scatter, read only)
    int32_t* status;
    int32_t dim;
    int32_t skip_row0;
    // one-sided placement (nrx_gather_inbox_place): a gathered row goes straight into its place in the REQUESTER's concat buffer --
    // peer_out[s] (the buffer of source rank s, mapped into this process: hipIpc / a peer mapping; this rank's own for s == rank) at
    // row inbox_pos[slot] (the sample), column feat_col[feature] -- instead of into a row buffer that travels back and is read again
    float* peer_out[NRX_MAX_FEATURES];     // [world]
    int32_t feat_col[NRX_MAX_FEATURES];    // [n_feats]
    const int32_t* inbox_pos;
    int64_t out_ld;
    int64_t out_rows;                      // rows of a requester's buffer: a position past them (a corrupt word from a peer) is dropped and reported
};
static_assert(sizeof(InboxArgs) <= 3584, "kernarg budget");

constexpr int INBOX_R = 8;   // slots per thread: 8 independent row reads in flight

template <int QLOG2, bool SCATTER>
__global__ __launch_bounds__(NRX_BLOCK) void inbox_kernel(const InboxArgs args_in_kernarg) {
    const NRX_CONST InboxArgs* a = nrx_kernarg<InboxArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t* s_pre = reinterpret_cast<int64_t*>(smem);       // [F + 1] prefix of this block's source rank
    const int F = a->n_feats;
    const int64_t cap = a->cap;
    // blocks are laid out per source rank: blockIdx.y = s, blockIdx.x tiles its cap slots
    const int s = blockIdx.y;
    if (threadIdx.x == 0) {
        int64_t acc = 0;
        for (int f = 0; f < F; ++f) {
            s_pre[f] = acc;
            acc += a->recv2d[s * F + f];
        }
        s_pre[F] = acc;
    }
    __syncthreads();
    const int64_t total = s_pre[F] < cap ? s_pre[F] : cap;
    const int q = threadIdx.x & (Q - 1);
    const int64_t j0 = (int64_t)blockIdx.x * (TB * INBOX_R) + (threadIdx.x >> QLOG2);
    if (j0 >= total) return;
    const int D = a->dim;
    const bool vec = (D & 3) == 0;
    int64_t row[INBOX_R];
    int tab[INBOX_R];
    bool ok[INBOX_R];
#pragma unroll
    for (int r = 0; r < INBOX_R; ++r) {
        const int64_t j = j0 + r * TB;
        ok[r] = j < total;
        row[r] = ok[r] ? (int64_t)nrx_gconst<int32_t>(a->inbox)[s * cap + j] : 0;
    }
#pragma unroll
    for (int r = 0; r < INBOX_R; ++r) {
        const int64_t j = j0 + r * TB;
        int lo = 0, hi = F;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid] <= j) lo = mid; else hi = mid;
        }
        tab[r] = a->feat_table[lo];
        if (ok[r] && (uint64_t)row[r] >= (uint64_t)a->rows[tab[r]]) {
            if (!SCATTER && q == 0) nrx_report_oob(a->status, tab[r], s * cap + j, row[r]);
            if (SCATTER) ok[r] = false;
            row[r] = 0;
        }
        if (SCATTER && a->skip_row0 && row[r] == 0) ok[r] = false;
    }
    for (int k0 = q * 4; k0 < D; k0 += 4 * Q) {
        if (SCATTER) {
#pragma unroll
            for (int r = 0; r < INBOX_R; ++r) {
                if (!ok[r]) continue;
                const int64_t p = s * cap + j0 + r * TB;
                const NRX_GLOBAL float* src = nrx_gconst<float>(a->buf) + p * (int64_t)D + k0;
                float* dst = a->table[tab[r]] + row[r] * (int64_t)D + k0;
                unsafeAtomicAdd(dst, src[0]);
                if (k0 + 1 < D) unsafeAtomicAdd(dst + 1, src[1]);
                if (k0 + 2 < D) unsafeAtomicAdd(dst + 2, src[2]);
                if (k0 + 3 < D) unsafeAtomicAdd(dst + 3, src[3]);
            }
        } else if (vec) {
            float4 v[INBOX_R];
#pragma unroll
            for (int r = 0; r < INBOX_R; ++r)
                v[r] = ok[r] ? nrx_ldg4_nt(a->table[tab[r]] + row[r] * (int64_t)D + k0, 0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < INBOX_R; ++r)
                if (ok[r]) nrx_stg4(a->buf, ((s * cap + j0 + r * TB) * (int64_t)D + k0) >> 2, v[r]);
        } else {
#pragma unroll
            for (int r = 0; r < INBOX_R; ++r) {
                if (!ok[r]) continue;
                const float* src = a->table[tab[r]] + row[r] * (int64_t)D + k0;
                NRX_GLOBAL float* dst = nrx_gmut<float>(a->buf) + (s * cap + j0 + r * TB) * (int64_t)D + k0;
                dst[0] = src[0];
                if (k0 + 1 < D) dst[1] = src[1];
                if (k0 + 2 < D) dst[2] = src[2];
                if (k0 + 3 < D) dst[3] = src[3];
            }
        }
    }
}

// Ring form of the owner-side GATHER for float4-addressable rows of up to 256 floats (the common case; the kernel above keeps
// the scatter-add, odd widths and wider rows).  Same idea as embed_fwd_ring: the R-load burst + R-store burst of the kernel
// above makes every block phase between reading and writing; here a lane group walks NS = R * NC consecutive-stride slots
// with R row reads in flight -- `store slot i; issue the load of slot i + R` -- so reads and writes interleave at row
// granularity, and a persistent grid (a few blocks per CU, each looping over (source, chunk) work items) replaces thousands
// of short blocks whose prologue (per-source feature prefix + binary searches) was paid once per 8 rows per lane.
// C2 at world = 1 (1.7 M rows of 64 B): 85 -> see profiles; results identical (a gather: bit-exact).
constexpr int INBOX_NC = 4;   // chunks of INBOX_R slots per lane group and work item

template <int QLOG2, bool PLACE = false>
__global__ __launch_bounds__(NRX_BLOCK) void inbox_gather_ring_kernel(const InboxArgs args_in_kernarg, int chunks_per_source) {
    const NRX_CONST InboxArgs* a = nrx_kernarg<InboxArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    constexpr int R = INBOX_R, NS = INBOX_R * INBOX_NC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t* s_pre = reinterpret_cast<int64_t*>(smem);       // [F + 1] prefix of the current source rank
    __shared__ const float* s_ptr[NS * TB];                  // the work item's row addresses (null past the source's count)
    __shared__ float* s_dst[PLACE ? NS * TB : 1];            // PLACE: where each row goes in the requester's concat
    const int F = a->n_feats;
    const int64_t cap = a->cap;
    const int q = threadIdx.x & (Q - 1), g = threadIdx.x >> QLOG2;
    const int D4 = a->dim >> 2;                              // float4 per row; lanes with q >= D4 idle
    const int nwork = a->world * chunks_per_source;
    int s_cur = -1;
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int s = work / chunks_per_source;
        const int chunk = work - s * chunks_per_source;
        if (s != s_cur) {                                    // block-uniform: (re)build the feature prefix of source s
            __syncthreads();
            if (threadIdx.x < 64) {                          // one wavefront: F <= 64 counts, inclusive scan by shuffles
                const int f = threadIdx.x;
                int64_t c = f < F ? a->recv2d[s * F + f] : 0;
                int64_t incl = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int64_t t = __shfl_up(incl, off, 64);
                    if (f >= off) incl += t;
                }
                if (f < F) s_pre[f] = incl - c;
                if (f == F - 1) s_pre[F] = incl;
            }
            __syncthreads();
            s_cur = s;
        }
        const int64_t total = s_pre[F] < cap ? s_pre[F] : cap;
        const int64_t base = (int64_t)chunk * (TB * NS);
        if (base >= total) continue;                         // block-uniform: nothing of this chunk is in use
        const int64_t j0 = base + g;
        // the chunk's slots are resolved ONCE, by the whole block in full-width coalesced loads: local row -> bounds check ->
        // owning table (prefix search) -> row address, parked in LDS; the walk below then needs one ds_read per row load
        // (resolving inside the walk put a ~6-step dependent chain in front of every load: 113 us instead of 85)
        __syncthreads();                                     // (the previous work item's readers are done)
        const NRX_GLOBAL int32_t* inb = nrx_gconst<int32_t>(a->inbox) + s * cap + base;
#pragma unroll
        for (int i = 0; i < NS * TB / NRX_BLOCK; ++i) {
            const int pos = i * NRX_BLOCK + threadIdx.x;
            const int64_t j = base + pos;
            const float* ptr = nullptr;
            float* dst = nullptr;
            if (j < total) {
                int lo = 0, hi = F;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_pre[mid] <= j) lo = mid; else hi = mid;
                }
                const int tab = a->feat_table[lo];
                int64_t row = inb[pos];
                if ((uint64_t)row >= (uint64_t)a->rows[tab]) {
                    nrx_report_oob(a->status, tab, s * cap + j, row);
                    row = 0;
                }
                ptr = a->table[tab] + row * (int64_t)a->dim;
                if (PLACE) {
                    // the position came from a peer: a word outside the requester's batch must not become a write into another process's memory
                    const int64_t at = nrx_gconst<int32_t>(a->inbox_pos)[s * cap + j];
                    if ((uint64_t)at < (uint64_t)a->out_rows) dst = a->peer_out[s] + at * a->out_ld + a->feat_col[lo];
                    else nrx_report_oob(a->status, tab, s * cap + j, at);
                }
            }
            s_ptr[pos] = ptr;
            if (PLACE) s_dst[pos] = dst;
        }
        __syncthreads();
        auto fetch = [&](int i) -> float4 {
            const float* ptr = s_ptr[i * TB + g];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ptr != nullptr && q < D4) v = nrx_ldg4_nt(ptr, q);
            return v;
        };
        float4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = fetch(r);
#pragma unroll 1
        for (int c = 0; c < INBOX_NC; ++c) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = c * R + r;
                const int64_t j = j0 + (int64_t)i * TB;
                if (PLACE) {
                    if (j < total && q < D4 && s_dst[i * TB + g] != nullptr) nrx_stg4(s_dst[i * TB + g], q, v[r]);
                } else if (j < total && q < D4) {
                    nrx_stg4(a->buf, (s * cap + j) * D4 + q, v[r]);
                }
                if (c + 1 < INBOX_NC) v[r] = fetch(i + R);
            }
        }
    }
}

// ------------------------------------------------------------------------------- owner-side partial pooling
// Pooled-bag channel: block s of the inbox holds {local row, tag, weight} triples in SOURCE order, so
// all lookups of one (feature, sample) -- one tag -- that this rank owns are CONTIGUOUS: a run.  Pass 1 (one thread per
// entry) records where every run starts and ends; pass 2 puts one Q-lane group on every (source, tag), which walks its run
// with 8 rows in flight -- the fused bag kernel's shape, every lane busy whatever the run lengths -- and writes
//     partial[s][tag][:] = sum_k w_k * table[row_k][:]           (summed in source order: deterministic, no atomics)
// (zeros for a (feature, sample) without a lookup on this rank).  The source then adds the `world` partials of a sample in rank order (a BAG_SUM over the returned
// slabs) -- the normalisation is already inside w (nrx_bag_norm_weights), so
//     sum_o sum_{k in o} (w_k / den) * row_k   ==   array_feature_pooling's  (sum_k w_k row_k) / den      (base_model.py:278-282)
// up to fp32 summation order (stated tolerance rtol 1e-6).
struct PoolArgs {
    float* table[NRX_MAX_FEATURES];        // weight tables (fwd) or grad tables (bwd)
    int64_t rows[NRX_MAX_FEATURES];
    int32_t feat_table[NRX_MAX_FEATURES];
    int32_t n_feats;
    int32_t world;
    int64_t cap;
    int64_t batch;                         // tags of feature f are f*batch .. (f+1)*batch - 1
    const int64_t* recv2d;                 // [world][n_feats] valid entries per (source, feature)
    const int32_t* inbox_rows;
    const int32_t* inbox_tag;
    const float* inbox_w;
    float* partial;                        // fwd: out [world][n_feats*batch][dim]; bwd: g_partial (read only)
    int32_t* status;
    int32_t dim;
    int32_t skip_row0;
};
static_assert(sizeof(PoolArgs) <= 3584, "kernarg budget");

__device__ __forceinline__ int64_t pool_block_total(const NRX_CONST PoolArgs* a, int s) {
    int64_t t = 0;
    for (int f = 0; f < a->n_feats; ++f) t += nrx_gconst<int64_t>(a->recv2d)[s * a->n_feats + f];
    return t < a->cap ? t : a->cap;
}

// pass 1: where does the run of every (source, tag) start and end?  One thread per inbox entry; run[.][0] = first entry,
// run[.][1] = one past the last (the array is zeroed first: tags without a lookup on this rank keep the empty run 0..0).
__global__ __launch_bounds__(NRX_BLOCK) void pool_mark_kernel(const PoolArgs args_in_kernarg, int32_t* __restrict__ run) {
    const NRX_CONST PoolArgs* a = nrx_kernarg<PoolArgs>();
    const int s = blockIdx.y;
    const int64_t total = pool_block_total(a, s);
    const int64_t j = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    if (j >= total) return;
    const NRX_GLOBAL int32_t* tags = nrx_gconst<int32_t>(a->inbox_tag) + (int64_t)s * a->cap;
    const int32_t tag = tags[j];
    const int64_t ntag = (int64_t)a->n_feats * a->batch;
    if ((uint32_t)tag >= (uint64_t)ntag) return;
    int32_t* r = run + ((int64_t)s * ntag + tag) * 2;
    if (j == 0 || tags[j - 1] != tag) r[0] = (int32_t)j;
    if (j == total - 1 || tags[j + 1] != tag) r[1] = (int32_t)(j + 1);
}

// pass 2: one Q-lane group per (source, tag) walks its run with AHEAD rows in flight -- the shape of the fused bag kernel
// The training step's per-entry words from the run bounds (nrx_pool_inbox_runs_words): the tags stayed at the source in the runs form, and what
// the backward's launches over the inbox read per entry follows from the run an entry lies in -- its tag (tag_out: nrx_pool_inbox_expand's
// inbox_tag), or its OWNER ID and PAYLOAD (oid_out / payload_out: exactly nrx_pool_inbox_owner_ids' words).  16 lanes per (source, tag) write the
// run's entries (adjacent tags' runs are adjacent: whole lines leave); the slots past a block's count get owner id 0 / payload 0.  (Written inside
// the pooling launch instead, these 4-byte stores cost it 43 us: partial lines evicted between the row fetches.)
struct PoolEmit {
    int32_t* tag_out;
    int32_t* oid_out;
    uint32_t* payload_out;
};
__global__ __launch_bounds__(NRX_BLOCK) void pool_runs_words_kernel(const PoolArgs args_in_kernarg, const int32_t* __restrict__ run, const PoolEmit em) {
    const NRX_CONST PoolArgs* a = nrx_kernarg<PoolArgs>();
    const int s = blockIdx.y;
    const int q = threadIdx.x & 15;
    const int64_t ntag = (int64_t)a->n_feats * a->batch;
    const int64_t base = (int64_t)s * a->cap;
    if (em.oid_out != nullptr) {
        const int64_t total = pool_block_total(a, s);
        for (int64_t j = total + (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; j < a->cap; j += (int64_t)gridDim.x * NRX_BLOCK) {
            em.oid_out[base + j] = 0;
            if (em.payload_out != nullptr) em.payload_out[base + j] = 0u;
        }
    }
    for (int64_t tag = (int64_t)blockIdx.x * (NRX_BLOCK / 16) + (threadIdx.x >> 4); tag < ntag; tag += (int64_t)gridDim.x * (NRX_BLOCK / 16)) {
        const int2 r = reinterpret_cast<const int2*>(run)[(int64_t)s * ntag + tag];
        for (int e = r.x + q; e < r.y; e += 16) {
            if (em.tag_out != nullptr) em.tag_out[base + e] = (int32_t)tag;
            if (em.oid_out != nullptr) {
                const int32_t row = nrx_gconst<int32_t>(a->inbox_rows)[base + e];
                const bool live = (uint32_t)row < (uint64_t)a->rows[0] && !(a->skip_row0 && row == 0);
                em.oid_out[base + e] = live ? row + 1 : 0;
                if (em.payload_out != nullptr) em.payload_out[base + e] = live ? (uint32_t)((int64_t)s * ntag + tag) : 0u;
            }
        }
    }
}

template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void pool_inbox_fwd_kernel(const PoolArgs args_in_kernarg, const int32_t* __restrict__ run) {
    const NRX_CONST PoolArgs* a = nrx_kernarg<PoolArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
#ifndef NRX_POOL_AHEAD
#define NRX_POOL_AHEAD 16      // rows in flight per lane group (4 / 8 / 12 / 16: C4's sharded forward 136 / 124 / 121 / 120.5 us)
#endif
    constexpr int AHEAD = NRX_POOL_AHEAD;
    const int s = blockIdx.y;
    const int q = threadIdx.x & (Q - 1);
    const int64_t ntag = (int64_t)a->n_feats * a->batch;
    const int64_t tag = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    const int64_t base = (int64_t)s * a->cap;
    if (tag >= ntag) return;
    const int32_t lo = run[((int64_t)s * ntag + tag) * 2], hi = run[((int64_t)s * ntag + tag) * 2 + 1];
    const int D = a->dim;
    const int f = (int)(tag / a->batch);
    const int tix = a->feat_table[f];
    const float* table = a->table[tix];
    const int64_t nrows = a->rows[tix];
    const bool vec = (D & 3) == 0;

    for (int k0 = q * 4; k0 < D; k0 += 4 * Q) {              // one pass when D <= 4Q (the launch picks Q for that)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e0 = lo; e0 < hi; e0 += AHEAD) {
            int32_t row[AHEAD];
            float w[AHEAD];
            // the eight entries' {row, weight}: fetched ONCE per lane group -- lane q loads entries q * EPL .. and the group exchanges them (every
            // lane loading all eight was 16 load instructions of 16 scattered words each: two thirds of this launch's L2 requests)
            constexpr int EPL = Q >= AHEAD ? 1 : AHEAD / Q;
            if constexpr (Q == 1) {
#pragma unroll
                for (int u = 0; u < AHEAD; ++u) {
                    const int e = e0 + u < hi ? e0 + u : lo;
                    row[u] = nrx_gconst<int32_t>(a->inbox_rows)[base + e];
                    w[u] = e0 + u < hi ? nrx_gconst<float>(a->inbox_w)[base + e] : 0.f;
                }
            } else {
                int32_t myrow[EPL];
                float myw[EPL];
#pragma unroll
                for (int x = 0; x < EPL; ++x) {
                    const int u = q * EPL + x;
                    const int e = (u < AHEAD && e0 + u < hi) ? e0 + u : lo;
                    myrow[x] = nrx_gconst<int32_t>(a->inbox_rows)[base + e];
                    myw[x] = (u < AHEAD && e0 + u < hi) ? nrx_gconst<float>(a->inbox_w)[base + e] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < AHEAD; ++u) {
                    row[u] = __shfl(myrow[u % EPL], u / EPL, Q);
                    w[u] = __shfl(myw[u % EPL], u / EPL, Q);
                }
            }
            float4 v[AHEAD];
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) {
                if ((uint32_t)row[u] >= (uint64_t)nrows) {
                    if (q == 0 && k0 == 0 && w[u] != 0.f) nrx_report_oob(a->status, tix, base + e0 + u, row[u]);
                    row[u] = 0;
                    w[u] = 0.f;
                }
                const float* p = table + (int64_t)row[u] * D + k0;
                if (vec) v[u] = nrx_ldg4(p, 0);
                else { v[u] = make_float4(p[0], k0 + 1 < D ? p[1] : 0.f, k0 + 2 < D ? p[2] : 0.f, k0 + 3 < D ? p[3] : 0.f); }
            }
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) {
#pragma clang fp contract(off)
                acc.x += v[u].x * w[u]; acc.y += v[u].y * w[u]; acc.z += v[u].z * w[u]; acc.w += v[u].w * w[u];
            }
        }
        float* dst = a->partial + ((int64_t)s * ntag + tag) * D + k0;
        dst[0] = acc.x;
        if (k0 + 1 < D) dst[1] = acc.y;
        if (k0 + 2 < D) dst[2] = acc.z;
        if (k0 + 3 < D) dst[3] = acc.w;
    }
}

// backward: grad_table[row] += w * g_partial[s][tag][:] for every entry (fp32 atomics; the global padding row is skipped)
template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void pool_inbox_bwd_kernel(const PoolArgs args_in_kernarg) {
    const NRX_CONST PoolArgs* a = nrx_kernarg<PoolArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    const int s = blockIdx.y;
    const int64_t total = pool_block_total(a, s);
    const int q = threadIdx.x & (Q - 1);
    const int64_t j = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    if (j >= total) return;
    const int64_t base = (int64_t)s * a->cap;
    const int32_t tag = nrx_gconst<int32_t>(a->inbox_tag)[base + j];
    const int32_t row = nrx_gconst<int32_t>(a->inbox_rows)[base + j];
    const float w = nrx_gconst<float>(a->inbox_w)[base + j];
    const int D = a->dim;
    const int f = (int)(tag / a->batch);
    const int tix = a->feat_table[f < a->n_feats ? f : 0];
    if ((uint32_t)row >= (uint64_t)a->rows[tix]) return;      // reported by the forward
    if (a->skip_row0 && row == 0) return;
    const float* g = a->partial + ((int64_t)s * a->n_feats * a->batch + tag) * D;
    float* dst = a->table[tix] + (int64_t)row * D;
    for (int k = q; k < D; k += Q) unsafeAtomicAdd(dst + k, nrx_gconst<float>(g)[k] * w);
}

// The pooled channel's backward as the single-GPU engine wants it (round 6): every inbox entry e = s * cap + j becomes ONE pseudo-lookup of the
// pooled table -- its OWNER ID (0 = nothing: past the block's count, the global padding row, a row that cannot be one; else local row + 1: the
// arena row) and its upstream row G[e] = w[e] * g_partial[s][tag[e]] -- so that the planners and the sorted reduction take over: deterministic
// row-sparse (keys, values) instead of pool_inbox_bwd_kernel's float atomics into a dense shard gradient.  `partial` = g_partial (read only).
template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void pool_inbox_expand_kernel(const PoolArgs args_in_kernarg, int32_t* __restrict__ owner_ids,
                                                                     float* __restrict__ g_rows) {
    const NRX_CONST PoolArgs* a = nrx_kernarg<PoolArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    const int s = blockIdx.y;
    const int64_t total = pool_block_total(a, s);
    const int q = threadIdx.x & (Q - 1);
    const int64_t j = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    if (j >= a->cap) return;
    const int64_t e = (int64_t)s * a->cap + j;
    if (j >= total) {
        if (q == 0) owner_ids[e] = 0;
        return;
    }
    const int32_t tag = nrx_gconst<int32_t>(a->inbox_tag)[e];
    const int32_t row = nrx_gconst<int32_t>(a->inbox_rows)[e];
    const float w = nrx_gconst<float>(a->inbox_w)[e];
    const int D = a->dim;
    const bool live = (uint32_t)row < (uint64_t)a->rows[0] && !(a->skip_row0 && row == 0) && (uint32_t)tag < (uint64_t)(a->n_feats * a->batch);
    if (q == 0) owner_ids[e] = live ? row + 1 : 0;
    if (!live || g_rows == nullptr) return;
    const float* g = a->partial + ((int64_t)s * a->n_feats * a->batch + tag) * D;
    float* dst = g_rows + e * D;
    if (D == 4 * Q) {
        float4 v = nrx_ldg4(g, q);
        v.x *= w; v.y *= w; v.z *= w; v.w *= w;
        nrx_stg4(dst, q, v);
    } else {
        for (int k = q; k < D; k += Q) dst[k] = nrx_gconst<float>(g)[k] * w;
    }
}

// normalised bag weights: masked mean w/(sum w + 1e-8) (base_model.py:278-282), plain mean 1/L (:275-276), sum w | 1
__global__ __launch_bounds__(NRX_BLOCK) void bag_norm_weights_kernel(const float* __restrict__ mask, int64_t batch, int L, int kind,
                                                                     float* __restrict__ out, float* __restrict__ inv_out = nullptr) {
    // bags of up to 64 entries: 16 lanes per sample, four samples per wavefront (a wavefront per 50-entry bag left most lanes
    // idle: 14.8 us at the C4 shape); longer bags: one wavefront per sample.  Sum order per sample is fixed either way.
    const bool narrow = L <= 64;
    const int gl = narrow ? 16 : 64;                                                    // lanes per sample
    const int q = threadIdx.x & (gl - 1);
    const int64_t b = ((int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x) / gl;
    if (b >= batch) return;                                                              // whole lane groups leave together
    float den = 1.0f;
    if (kind == NRX_BAG_MASKED_MEAN) {
        float part = 0.f;
        for (int l = q; l < L; l += gl) part += mask[b * L + l];
        if (narrow) {                                                                    // sum over the 16-lane row (DPP)
            part += nrx_dpp<0xB1>(part);
            part += nrx_dpp<0x4E>(part);
            part += nrx_dpp<0x141>(part);
            part += nrx_dpp<0x140>(part);
        } else {
            part = nrx_wave_sum(part);
        }
        den = part + 1e-8f;
    } else if (kind == NRX_BAG_MEAN) {
        den = (float)L;
    }
    for (int l = q; l < L; l += gl) {
        const float m = (kind == NRX_BAG_MEAN || mask == nullptr) ? 1.0f : mask[b * L + l];
        out[b * L + l] = m / den;
    }
    // the weight every live entry of the sample carries when the mask is 0 / 1 (1 / den; 0 for an empty bag): what the pooled channel's
    // backward pre-multiplies the sample's upstream row by (shard_step, binary_masks)
    if (inv_out != nullptr && q == 0) inv_out[b] = (kind == NRX_BAG_MASKED_MEAN && den <= 1e-8f) ? 0.f : 1.0f / den;
}

int log2_ceil(int x) {
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

template <bool SCATTER>
int launch_inbox(InboxArgs& a, hipStream_t st, const char* who) {
    int ql = log2_ceil((a.dim + 3) / 4);
    if (ql > 6) ql = 6;
    const int tb = NRX_BLOCK >> ql;
    const int per_block = tb * INBOX_R;
    dim3 grid((unsigned)((a.cap + per_block - 1) / per_block), (unsigned)a.world);
    const size_t smem = (size_t)(a.n_feats + 1) * sizeof(int64_t);
    if (!SCATTER && (a.dim & 3) == 0 && a.dim <= 256 && ql >= 2) {          // ring form (see inbox_gather_ring_kernel)
        for (int t = 0; t < a.n_feats; ++t)
            if (!nrx_aligned16(a.table[a.feat_table[t]])) goto classic;
        {
            const int per_item = tb * INBOX_R * INBOX_NC;
            const int64_t cps = (a.cap + per_item - 1) / per_item;
            NRX_REQUIRE(cps * a.world <= 0x7fffffffLL, "%s: too many rows for one launch", who);
            int64_t g = cps * a.world;
            if (g < 512) goto classic;            // a small exchange (C4's id features: 64 items) fills the chip better with the short blocks
            const int64_t persistent = 256 * 6;                              // ~6 resident blocks per CU walk the work items
            if (g > persistent) g = persistent;
            switch (ql) {
#define NRX_RCASE(QL_) case QL_: hipLaunchKernelGGL((inbox_gather_ring_kernel<QL_>), dim3((unsigned)g), dim3(NRX_BLOCK), smem, st, a, (int)cps); break;
                NRX_RCASE(2) NRX_RCASE(3) NRX_RCASE(4) NRX_RCASE(5)
                default: hipLaunchKernelGGL((inbox_gather_ring_kernel<6>), dim3((unsigned)g), dim3(NRX_BLOCK), smem, st, a, (int)cps); break;
#undef NRX_RCASE
            }
            NRX_LAUNCH_CHECK(who);
            return NRX_OK;
        }
    }
classic:
    switch (ql) {
#define NRX_CASE(QL_) case QL_: hipLaunchKernelGGL((inbox_kernel<QL_, SCATTER>), grid, dim3(NRX_BLOCK), smem, st, a); break;
        NRX_CASE(0) NRX_CASE(1) NRX_CASE(2) NRX_CASE(3) NRX_CASE(4) NRX_CASE(5)
        default: hipLaunchKernelGGL((inbox_kernel<6, SCATTER>), grid, dim3(NRX_BLOCK), smem, st, a); break;
#undef NRX_CASE
    }
    NRX_LAUNCH_CHECK(who);
    return NRX_OK;
}

int fill_inbox_args(InboxArgs& a, float* const* tables, const int64_t* table_rows, int32_t n_tables,
                    const int32_t* feat_table, int32_t n_feats, int32_t world, int64_t cap, const int64_t* recv2d,
                    const int32_t* inbox_rows, int32_t dim, const char* who) {
    NRX_REQUIRE(tables && table_rows && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES, "%s: n_tables must be in [1, %d]", who, NRX_MAX_FEATURES);
    NRX_REQUIRE(feat_table && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "%s: n_feats must be in [1, %d]", who, NRX_MAX_FEATURES);
    NRX_REQUIRE(world >= 1 && world <= 64 && cap >= 1 && dim >= 1, "%s: bad world/cap/dim", who);
    NRX_REQUIRE(recv2d && inbox_rows, "%s: null buffer", who);
    for (int i = 0; i < n_tables; ++i) {
        NRX_REQUIRE(tables[i] != nullptr, "%s: table %d is null", who, i);
        NRX_REQUIRE((dim & 3) != 0 || nrx_aligned16(tables[i]), "%s: table %d must be 16-byte aligned", who, i);
        a.table[i] = tables[i];
        a.rows[i] = table_rows[i];
    }
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(feat_table[f] >= 0 && feat_table[f] < n_tables, "%s: feat_table[%d] out of range", who, f);
        a.feat_table[f] = feat_table[f];
    }
    a.n_feats = n_feats;
    a.world = world;
    a.cap = cap;
    a.recv2d = recv2d;
    a.inbox = inbox_rows;
    a.dim = dim;
    return NRX_OK;
}

}  // namespace

extern "C" int64_t nrx_route_workspace(int64_t n_total, int32_t world) {
    if (n_total < 0 || world < 1) return -1;
    // int32 histogram [world][chunks]; every feature may end in a partial chunk.  Size in int64 units.
    const int64_t nchunks = (n_total + CHUNK - 1) / CHUNK + NRX_MAX_FEATURES;
    return (nchunks * world + 1) / 2 + 1;
}

static int route_ids_impl(const void* const* ids, const int64_t* lens, int32_t n_feats, int32_t index_bits,
                          int32_t world, int64_t cap, int32_t* send_rows, int32_t* slot, int64_t* counts2d,
                          int64_t* overflow, int64_t* workspace, int32_t* send_pos, void* stream) {
    NRX_REQUIRE(ids && lens && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "nrx_route_ids: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_route_ids: index_bits must be 32 or 64");
    NRX_REQUIRE(world >= 1 && world <= 64 && cap >= 1 && cap * world <= 0x7fffffffLL, "nrx_route_ids: bad world / cap");
    NRX_REQUIRE(send_rows && counts2d && overflow && workspace, "nrx_route_ids: null buffer");
    RouteArgs a;
    int64_t off = 0, chunks = 0;
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(lens[f] >= 0 && (lens[f] == 0 || ids[f] != nullptr), "nrx_route_ids: feature %d: bad ids/len", f);
        a.ids[f] = ids[f];
        a.len[f] = lens[f];
        a.off[f] = off;
        a.chunk0[f] = (int32_t)chunks;
        off += lens[f];
        chunks += (lens[f] + CHUNK - 1) / CHUNK;
    }
    a.off[n_feats] = off;
    a.chunk0[n_feats] = (int32_t)chunks;
    NRX_REQUIRE(off <= 0x7fffffffLL, "nrx_route_ids: too many ids for one exchange");
    NRX_REQUIRE(slot != nullptr || off == 0, "nrx_route_ids: null slot buffer");   // an exchange may carry zero ids
    a.n_feats = n_feats;
    a.world = world;
    a.idx64 = index_bits == 64;
    a.nchunks = (int32_t)chunks;
    a.cap = cap;
    a.hist = reinterpret_cast<int32_t*>(workspace);
    a.counts2d = counts2d;
    a.send_rows = send_rows;
    a.slot = slot;
    a.overflow = overflow;
    a.send_tag = nullptr;
    a.send_w = nullptr;
    a.bags = 0;
    a.send_pos = send_pos;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (world == 1) {       // nothing to bucket: one narrowing pass; block 0 also writes the counts (= the lengths)
        const unsigned g = chunks > 0 ? (unsigned)chunks : 1u;
        hipLaunchKernelGGL(route_single, dim3(g), dim3(NRX_BLOCK), 0, st, a);
        NRX_LAUNCH_CHECK("nrx_route_ids(world 1)");
        return NRX_OK;
    }
    if (chunks > 0) hipLaunchKernelGGL(route_hist, dim3((unsigned)chunks), dim3(NRX_BLOCK), 0, st, a);
    hipLaunchKernelGGL(route_scan, dim3(1), dim3(NRX_BLOCK), 0, st, a);
    if (chunks > 0) hipLaunchKernelGGL(route_place, dim3((unsigned)chunks), dim3(NRX_BLOCK), 0, st, a);
    NRX_LAUNCH_CHECK("nrx_route_ids");
    return NRX_OK;
}

extern "C" int nrx_route_ids(const void* const* ids, const int64_t* lens, int32_t n_feats, int32_t index_bits,
                             int32_t world, int64_t cap, int32_t* send_rows, int32_t* slot, int64_t* counts2d,
                             int64_t* overflow, int64_t* workspace, void* stream) {
    NRX_TRACE();
    return route_ids_impl(ids, lens, n_feats, index_bits, world, cap, send_rows, slot, counts2d, overflow, workspace, nullptr, stream);
}

extern "C" int nrx_route_ids_pos(const void* const* ids, const int64_t* lens, int32_t n_feats, int32_t index_bits,
                                 int32_t world, int64_t cap, int32_t* send_rows, int32_t* send_pos, int32_t* slot, int64_t* counts2d,
                                 int64_t* overflow, int64_t* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(send_pos != nullptr, "nrx_route_ids_pos: null send_pos");
    return route_ids_impl(ids, lens, n_feats, index_bits, world, cap, send_rows, slot, counts2d, overflow, workspace, send_pos, stream);
}

extern "C" int nrx_gather_inbox_place(const float* const* tables, const int64_t* table_rows, int32_t n_tables,
                                      const int32_t* feat_table, int32_t n_feats, int32_t world, int64_t cap,
                                      const int64_t* recv2d, const int32_t* inbox_rows, const int32_t* inbox_pos, int32_t dim,
                                      float* const* peer_out, int64_t out_ld, int64_t out_rows, const int32_t* feat_col, int32_t* status, void* stream) {
    NRX_TRACE();
    InboxArgs a;
    int rc = fill_inbox_args(a, const_cast<float* const*>(tables), table_rows, n_tables, feat_table, n_feats, world, cap,
                             recv2d, inbox_rows, dim, "nrx_gather_inbox_place");
    if (rc != NRX_OK) return rc;
    NRX_REQUIRE(inbox_pos && peer_out && feat_col && out_ld >= dim && out_rows >= 0, "nrx_gather_inbox_place: null / bad placement argument");
    if ((dim & 3) != 0 || dim > 256 || dim < 16 || (out_ld & 3) != 0) {
        nrx_set_error("nrx_gather_inbox_place: rows of 16..256 floats (a multiple of 4) and a row stride that is a multiple of 4 floats");
        return NRX_ERR_UNSUPPORTED;
    }
    for (int s = 0; s < world; ++s) {
        NRX_REQUIRE(peer_out[s] != nullptr && nrx_aligned16(peer_out[s]), "nrx_gather_inbox_place: peer_out[%d] must be a 16-byte aligned mapping", s);
        a.peer_out[s] = peer_out[s];
    }
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(feat_col[f] >= 0 && (feat_col[f] & 3) == 0 && feat_col[f] + dim <= out_ld, "nrx_gather_inbox_place: feat_col[%d] must be a multiple of 4 inside the row", f);
        a.feat_col[f] = feat_col[f];
    }
    a.inbox_pos = inbox_pos;
    a.out_ld = out_ld;
    a.out_rows = out_rows;
    a.buf = nullptr;
    a.status = status;
    a.skip_row0 = 0;
    int ql = log2_ceil((dim + 3) / 4);
    const int tb = NRX_BLOCK >> ql;
    const int per_item = tb * INBOX_R * INBOX_NC;
    const int64_t cps = (cap + per_item - 1) / per_item;
    NRX_REQUIRE(cps * world <= 0x7fffffffLL, "nrx_gather_inbox_place: too many rows for one launch");
    int64_t g = cps * world;
    if (g > 256 * 6) g = 256 * 6;
    const size_t smem = (size_t)(n_feats + 1) * sizeof(int64_t);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (ql) {
        case 2: hipLaunchKernelGGL((inbox_gather_ring_kernel<2, true>), dim3((unsigned)g), dim3(NRX_BLOCK), smem, st, a, (int)cps); break;
        case 3: hipLaunchKernelGGL((inbox_gather_ring_kernel<3, true>), dim3((unsigned)g), dim3(NRX_BLOCK), smem, st, a, (int)cps); break;
        case 4: hipLaunchKernelGGL((inbox_gather_ring_kernel<4, true>), dim3((unsigned)g), dim3(NRX_BLOCK), smem, st, a, (int)cps); break;
        case 5: hipLaunchKernelGGL((inbox_gather_ring_kernel<5, true>), dim3((unsigned)g), dim3(NRX_BLOCK), smem, st, a, (int)cps); break;
        default: hipLaunchKernelGGL((inbox_gather_ring_kernel<6, true>), dim3((unsigned)g), dim3(NRX_BLOCK), smem, st, a, (int)cps); break;
    }
    NRX_LAUNCH_CHECK("nrx_gather_inbox_place");
    return NRX_OK;
}

extern "C" int nrx_gather_inbox(const float* const* tables, const int64_t* table_rows, int32_t n_tables,
                                const int32_t* feat_table, int32_t n_feats, int32_t world, int64_t cap,
                                const int64_t* recv2d, const int32_t* inbox_rows, int32_t dim,
                                float* out_rows, int32_t* status, void* stream) {
    NRX_TRACE();
    InboxArgs a;
    int rc = fill_inbox_args(a, const_cast<float* const*>(tables), table_rows, n_tables, feat_table, n_feats, world, cap,
                             recv2d, inbox_rows, dim, "nrx_gather_inbox");
    if (rc != NRX_OK) return rc;
    NRX_REQUIRE(out_rows != nullptr && ((dim & 3) != 0 || nrx_aligned16(out_rows)), "nrx_gather_inbox: bad out_rows");
    a.buf = out_rows;
    a.status = status;
    a.skip_row0 = 0;
    return launch_inbox<false>(a, reinterpret_cast<hipStream_t>(stream), "nrx_gather_inbox");
}

extern "C" int nrx_scatter_add_inbox(float* const* grad_tables, const int64_t* table_rows, int32_t n_tables,
                                     const int32_t* feat_table, int32_t n_feats, int32_t world, int64_t cap,
                                     const int64_t* recv2d, const int32_t* inbox_rows, int32_t dim,
                                     const float* g_rows, int32_t skip_row0, void* stream) {
    NRX_TRACE();
    InboxArgs a;
    int rc = fill_inbox_args(a, grad_tables, table_rows, n_tables, feat_table, n_feats, world, cap, recv2d, inbox_rows, dim,
                             "nrx_scatter_add_inbox");
    if (rc != NRX_OK) return rc;
    NRX_REQUIRE(g_rows != nullptr, "nrx_scatter_add_inbox: null g_rows");
    a.buf = const_cast<float*>(g_rows);
    a.status = nullptr;
    a.skip_row0 = skip_row0;
    return launch_inbox<true>(a, reinterpret_cast<hipStream_t>(stream), "nrx_scatter_add_inbox");
}

extern "C" int nrx_bag_norm_weights(const float* mask, int64_t batch, int32_t bag_len, int32_t kind, float* out_w, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(out_w && batch >= 0 && bag_len >= 1, "nrx_bag_norm_weights: bad argument");
    NRX_REQUIRE(kind == NRX_BAG_MASKED_MEAN || kind == NRX_BAG_MEAN || kind == NRX_BAG_SUM, "nrx_bag_norm_weights: bad kind");
    NRX_REQUIRE(kind != NRX_BAG_MASKED_MEAN || mask != nullptr, "nrx_bag_norm_weights: masked mean needs a mask");
    if (batch == 0) return NRX_OK;
    const int per_block = NRX_BLOCK / (bag_len <= 64 ? 16 : 64);
    const unsigned grid = (unsigned)((batch + per_block - 1) / per_block);
    hipLaunchKernelGGL(bag_norm_weights_kernel, dim3(grid), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), mask, batch,
                       bag_len, kind, out_w, (float*)nullptr);
    NRX_LAUNCH_CHECK("nrx_bag_norm_weights");
    return NRX_OK;
}

extern "C" int nrx_bag_norm_weights_inv(const float* mask, int64_t batch, int32_t bag_len, int32_t kind, float* out_w, float* out_inv, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(out_w != nullptr && out_inv != nullptr && batch >= 0 && bag_len >= 1, "nrx_bag_norm_weights_inv: bad argument");
    NRX_REQUIRE(kind == NRX_BAG_MASKED_MEAN || kind == NRX_BAG_MEAN || kind == NRX_BAG_SUM, "nrx_bag_norm_weights_inv: not a bag kind");
    NRX_REQUIRE(kind != NRX_BAG_MASKED_MEAN || mask != nullptr, "nrx_bag_norm_weights_inv: masked mean needs a mask");
    if (batch == 0) return NRX_OK;
    const int per_block = NRX_BLOCK / (bag_len <= 64 ? 16 : 64);
    const unsigned grid = (unsigned)((batch + per_block - 1) / per_block);
    hipLaunchKernelGGL(bag_norm_weights_kernel, dim3(grid), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), mask, batch,
                       bag_len, kind, out_w, out_inv);
    NRX_LAUNCH_CHECK("nrx_bag_norm_weights_inv");
    return NRX_OK;
}

// The pooled channel's backward, requester side: the block of sample gradients every owner gets -- column block [col, col + dim) of the upstream
// gradient of the concat, each row times its sample's weight (scale, optional), written `copies` times at a stride (one copy per owner).
__global__ __launch_bounds__(NRX_BLOCK) void bag_upstream_rows_kernel(const float* __restrict__ g_out, int64_t ld, int col, int dim, int64_t batch,
                                                                     const float* __restrict__ scale, int copies, int64_t copy_stride,
                                                                     float* __restrict__ dst) {
    const int64_t total = batch * dim;
    for (int64_t i = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * NRX_BLOCK) {
        const int64_t b = i / dim;
        const int k = (int)(i - b * dim);
        float v = g_out[b * ld + col + k];
        if (scale != nullptr) v *= scale[b];
        for (int c = 0; c < copies; ++c) dst[(int64_t)c * copy_stride + i] = v;
    }
}

extern "C" int nrx_bag_upstream_rows(const float* g_out, int64_t ld, int32_t col, int32_t dim, int64_t batch, const float* scale, int32_t copies,
                                     int64_t copy_stride, float* dst, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(g_out != nullptr && dst != nullptr && dim >= 1 && col >= 0 && ld >= (int64_t)col + dim && batch >= 0 && copies >= 1 &&
                (copies == 1 || copy_stride >= batch * dim), "nrx_bag_upstream_rows: bad argument");
    if (batch == 0) return NRX_OK;
    int64_t blocks = (batch * dim + NRX_BLOCK - 1) / NRX_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bag_upstream_rows_kernel, dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), g_out, ld, (int)col,
                       (int)dim, batch, scale, (int)copies, copy_stride, dst);
    NRX_LAUNCH_CHECK("nrx_bag_upstream_rows");
    return NRX_OK;
}

extern "C" int nrx_route_bags(const void* const* ids, const float* const* weights, const int32_t* bag_lens, int32_t n_feats,
                              int32_t index_bits, int64_t batch, int32_t world, int64_t cap, int32_t* send_rows,
                              int32_t* send_tag, float* send_w, int64_t* counts2d, int64_t* overflow, int64_t* workspace,
                              void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(ids && bag_lens && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "nrx_route_bags: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_route_bags: index_bits must be 32 or 64");
    NRX_REQUIRE(world >= 1 && world <= 64 && cap >= 1 && cap * world <= 0x7fffffffLL && batch >= 0, "nrx_route_bags: bad world / cap / batch");
    NRX_REQUIRE((int64_t)n_feats * batch <= 0x7fffffffLL, "nrx_route_bags: n_feats * batch must fit 31 bits");
    NRX_REQUIRE(send_rows && send_tag && send_w && counts2d && overflow && workspace, "nrx_route_bags: null buffer");
    RouteArgs a;
    int64_t off = 0, chunks = 0;
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(bag_lens[f] >= 1 && (batch == 0 || ids[f] != nullptr), "nrx_route_bags: feature %d: bad ids / bag_len", f);
        const int64_t len = batch * bag_lens[f];
        a.ids[f] = ids[f];
        a.len[f] = len;
        a.off[f] = off;
        a.chunk0[f] = (int32_t)chunks;
        a.weight[f] = weights ? weights[f] : nullptr;
        a.bag_len[f] = bag_lens[f];
        a.tag0[f] = (int32_t)(f * batch);
        off += len;
        chunks += (len + CHUNK - 1) / CHUNK;
    }
    a.off[n_feats] = off;
    a.chunk0[n_feats] = (int32_t)chunks;
    NRX_REQUIRE(off <= 0x7fffffffLL, "nrx_route_bags: too many ids for one exchange");
    a.n_feats = n_feats;
    a.world = world;
    a.idx64 = index_bits == 64;
    a.nchunks = (int32_t)chunks;
    a.cap = cap;
    a.hist = reinterpret_cast<int32_t*>(workspace);
    a.counts2d = counts2d;
    a.send_rows = send_rows;
    a.slot = nullptr;
    a.overflow = overflow;
    a.send_tag = send_tag;
    a.send_w = send_w;
    a.bags = 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (chunks > 0) hipLaunchKernelGGL(route_hist, dim3((unsigned)chunks), dim3(NRX_BLOCK), 0, st, a);
    hipLaunchKernelGGL(route_scan, dim3(1), dim3(NRX_BLOCK), 0, st, a);
    if (chunks > 0) hipLaunchKernelGGL(route_place, dim3((unsigned)chunks), dim3(NRX_BLOCK), 0, st, a);
    NRX_LAUNCH_CHECK("nrx_route_bags");
    return NRX_OK;
}

namespace {
int fill_pool_args(PoolArgs& a, float* const* tables, const int64_t* table_rows, int32_t n_tables, const int32_t* feat_table,
                   int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d, const int32_t* inbox_rows,
                   const int32_t* inbox_tag, const float* inbox_w, int32_t dim, const char* who) {
    NRX_REQUIRE(tables && table_rows && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES, "%s: n_tables must be in [1, %d]", who, NRX_MAX_FEATURES);
    NRX_REQUIRE(feat_table && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "%s: n_feats must be in [1, %d]", who, NRX_MAX_FEATURES);
    NRX_REQUIRE(world >= 1 && world <= 64 && cap >= 1 && dim >= 1 && dim <= 1024 && batch >= 1, "%s: bad world/cap/dim/batch", who);
    NRX_REQUIRE(recv2d && inbox_rows && inbox_tag && inbox_w, "%s: null buffer", who);
    for (int i = 0; i < n_tables; ++i) {
        NRX_REQUIRE(tables[i] != nullptr, "%s: table %d is null", who, i);
        NRX_REQUIRE((dim & 3) != 0 || nrx_aligned16(tables[i]), "%s: table %d must be 16-byte aligned", who, i);
        a.table[i] = tables[i];
        a.rows[i] = table_rows[i];
    }
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(feat_table[f] >= 0 && feat_table[f] < n_tables, "%s: feat_table[%d] out of range", who, f);
        a.feat_table[f] = feat_table[f];
    }
    a.n_feats = n_feats;
    a.world = world;
    a.cap = cap;
    a.batch = batch;
    a.recv2d = recv2d;
    a.inbox_rows = inbox_rows;
    a.inbox_tag = inbox_tag;
    a.inbox_w = inbox_w;
    a.dim = dim;
    return NRX_OK;
}

int pool_ql(int dim) {
    int ql = log2_ceil((dim + 3) / 4);
    return ql > 6 ? 6 : ql;
}
}  // namespace

extern "C" int64_t nrx_pool_inbox_workspace(int32_t n_feats, int64_t batch, int32_t world) {
    if (n_feats < 1 || batch < 0 || world < 1) return -1;
    return (int64_t)world * n_feats * batch * 2 * (int64_t)sizeof(int32_t);      // bytes: run start / end per (source, tag)
}

extern "C" int nrx_pool_inbox_fwd(const float* const* tables, const int64_t* table_rows, int32_t n_tables, const int32_t* feat_table,
                                  int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                                  const int32_t* inbox_rows, const int32_t* inbox_tag, const float* inbox_w, int32_t dim,
                                  float* partial, void* workspace, int32_t* status, void* stream) {
    NRX_TRACE();
    PoolArgs a;
    int rc = fill_pool_args(a, const_cast<float* const*>(tables), table_rows, n_tables, feat_table, n_feats, batch, world, cap, recv2d,
                            inbox_rows, inbox_tag, inbox_w, dim, "nrx_pool_inbox_fwd");
    if (rc != NRX_OK) return rc;
    NRX_REQUIRE(partial != nullptr && workspace != nullptr, "nrx_pool_inbox_fwd: null partial / workspace");
    a.partial = partial;
    a.status = status;
    a.skip_row0 = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int32_t* run = reinterpret_cast<int32_t*>(workspace);
    const int64_t ntag = (int64_t)n_feats * batch;
    if (nrx_zero_async(run, (size_t)world * ntag * 2 * sizeof(int32_t), st) != NRX_OK) return NRX_ERR_LAUNCH;
    hipLaunchKernelGGL(pool_mark_kernel, dim3((unsigned)((cap + NRX_BLOCK - 1) / NRX_BLOCK), (unsigned)world), dim3(NRX_BLOCK), 0, st, a, run);
    const int ql = pool_ql(dim);
    const int tb = NRX_BLOCK >> ql;
    const dim3 grid((unsigned)((ntag + tb - 1) / tb), (unsigned)world);
    switch (ql) {
#define NRX_CASE(QL_) case QL_: hipLaunchKernelGGL((pool_inbox_fwd_kernel<QL_>), grid, dim3(NRX_BLOCK), 0, st, a, (const int32_t*)run); break;
        NRX_CASE(0) NRX_CASE(1) NRX_CASE(2) NRX_CASE(3) NRX_CASE(4) NRX_CASE(5)
        default: hipLaunchKernelGGL((pool_inbox_fwd_kernel<6>), grid, dim3(NRX_BLOCK), 0, st, a, (const int32_t*)run); break;
#undef NRX_CASE
    }
    NRX_LAUNCH_CHECK("nrx_pool_inbox_fwd");
    return NRX_OK;
}

// The owner's pooling launch alone, over run bounds that ARRIVED (nrx_route_bags_runs wrote them at the source; block s of `run` = what source s
// sent): no memset, no marking pass over the entries.
extern "C" int nrx_pool_inbox_fwd_runs(const float* const* tables, const int64_t* table_rows, int32_t n_tables, const int32_t* feat_table,
                                       int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                                       const int32_t* inbox_rows, const float* inbox_w, const int32_t* run, int32_t dim, float* partial,
                                       int32_t* status, void* stream) {
    NRX_TRACE();
    PoolArgs a;
    NRX_REQUIRE(run != nullptr, "nrx_pool_inbox_fwd_runs: null run");
    int rc = fill_pool_args(a, const_cast<float* const*>(tables), table_rows, n_tables, feat_table, n_feats, batch, world, cap, recv2d,
                            inbox_rows, run /* (no tag array: never read) */, inbox_w, dim, "nrx_pool_inbox_fwd_runs");
    if (rc != NRX_OK) return rc;
    NRX_REQUIRE(partial != nullptr, "nrx_pool_inbox_fwd_runs: null partial");
    a.inbox_tag = nullptr;
    a.partial = partial;
    a.status = status;
    a.skip_row0 = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t ntag = (int64_t)n_feats * batch;
    const int ql = pool_ql(dim);
    const int tb = NRX_BLOCK >> ql;
    const dim3 grid((unsigned)((ntag + tb - 1) / tb), (unsigned)world);
    switch (ql) {
#define NRX_CASE(QL_) case QL_: hipLaunchKernelGGL((pool_inbox_fwd_kernel<QL_>), grid, dim3(NRX_BLOCK), 0, st, a, run); break;
        NRX_CASE(0) NRX_CASE(1) NRX_CASE(2) NRX_CASE(3) NRX_CASE(4) NRX_CASE(5)
        default: hipLaunchKernelGGL((pool_inbox_fwd_kernel<6>), grid, dim3(NRX_BLOCK), 0, st, a, run); break;
#undef NRX_CASE
    }
    NRX_LAUNCH_CHECK("nrx_pool_inbox_fwd_runs");
    return NRX_OK;
}

extern "C" int nrx_pool_inbox_runs_words(int64_t table_rows, int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                                         const int32_t* inbox_rows, const int32_t* run, int32_t skip_row0, int32_t* tag_out, int32_t* oid_out,
                                         uint32_t* payload_out, void* stream) {
    NRX_TRACE();
    PoolArgs a;
    NRX_REQUIRE(run != nullptr && (tag_out != nullptr || oid_out != nullptr), "nrx_pool_inbox_runs_words: null run / nothing to write");
    NRX_REQUIRE(payload_out == nullptr || oid_out != nullptr, "nrx_pool_inbox_runs_words: payload_out goes with oid_out");
    float* dummy_table = reinterpret_cast<float*>(const_cast<int32_t*>(run));      // (fill_pool_args wants a table pointer and a weight array: never dereferenced here)
    int32_t ft[NRX_MAX_FEATURES];
    for (int f = 0; f < NRX_MAX_FEATURES; ++f) ft[f] = 0;
    int rc = fill_pool_args(a, &dummy_table, &table_rows, 1, ft, n_feats, batch, world, cap, recv2d, inbox_rows, run, reinterpret_cast<const float*>(run), 4,
                            "nrx_pool_inbox_runs_words");
    if (rc != NRX_OK) return rc;
    a.inbox_tag = nullptr;
    a.partial = nullptr;
    a.status = nullptr;
    a.skip_row0 = skip_row0;
    PoolEmit em;
    em.tag_out = tag_out;
    em.oid_out = oid_out;
    em.payload_out = payload_out;
    const int64_t ntag = (int64_t)n_feats * batch;
    int64_t bx = (ntag + NRX_BLOCK / 16 - 1) / (NRX_BLOCK / 16);
    if (bx > 8192) bx = 8192;
    hipLaunchKernelGGL(pool_runs_words_kernel, dim3((unsigned)bx, (unsigned)world), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), a, run, em);
    NRX_LAUNCH_CHECK("nrx_pool_inbox_runs_words");
    return NRX_OK;
}

extern "C" int nrx_pool_inbox_bwd(float* const* grad_tables, const int64_t* table_rows, int32_t n_tables, const int32_t* feat_table,
                                  int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                                  const int32_t* inbox_rows, const int32_t* inbox_tag, const float* inbox_w, int32_t dim,
                                  const float* g_partial, int32_t skip_row0, void* stream) {
    NRX_TRACE();
    PoolArgs a;
    int rc = fill_pool_args(a, grad_tables, table_rows, n_tables, feat_table, n_feats, batch, world, cap, recv2d, inbox_rows, inbox_tag,
                            inbox_w, dim, "nrx_pool_inbox_bwd");
    if (rc != NRX_OK) return rc;
    NRX_REQUIRE(g_partial != nullptr, "nrx_pool_inbox_bwd: null g_partial");
    a.partial = const_cast<float*>(g_partial);
    a.status = nullptr;
    a.skip_row0 = skip_row0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ql = pool_ql(dim);
    const int tb = NRX_BLOCK >> ql;
    const dim3 grid((unsigned)((cap + tb - 1) / tb), (unsigned)world);
    switch (ql) {
#define NRX_CASE(QL_) case QL_: hipLaunchKernelGGL((pool_inbox_bwd_kernel<QL_>), grid, dim3(NRX_BLOCK), 0, st, a); break;
        NRX_CASE(0) NRX_CASE(1) NRX_CASE(2) NRX_CASE(3) NRX_CASE(4) NRX_CASE(5)
        default: hipLaunchKernelGGL((pool_inbox_bwd_kernel<6>), grid, dim3(NRX_BLOCK), 0, st, a); break;
#undef NRX_CASE
    }
    NRX_LAUNCH_CHECK("nrx_pool_inbox_bwd");
    return NRX_OK;
}

// The owner ids alone, one thread per entry, and -- optionally -- every entry's PAYLOAD for nrx_sparse_plan_ex(NRX_PLAN_PAYLOAD): the row
// s * n_tags + tag of the [world * n_tags, dim] block of sample gradients its upstream row is.
__global__ __launch_bounds__(NRX_BLOCK) void pool_owner_ids_kernel(const PoolArgs args_in_kernarg, int32_t* __restrict__ owner_ids, uint32_t* __restrict__ payload) {
    const NRX_CONST PoolArgs* a = nrx_kernarg<PoolArgs>();
    const int s = blockIdx.y;
    const int64_t total = pool_block_total(a, s);
    const int64_t n_tags = (int64_t)a->n_feats * a->batch;
    for (int64_t j = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; j < a->cap; j += (int64_t)gridDim.x * NRX_BLOCK) {
        const int64_t e = (int64_t)s * a->cap + j;
        int32_t id = 0;
        uint32_t pv = 0;
        if (j < total) {
            const int32_t tag = nrx_gconst<int32_t>(a->inbox_tag)[e];
            const int32_t row = nrx_gconst<int32_t>(a->inbox_rows)[e];
            const bool live = (uint32_t)row < (uint64_t)a->rows[0] && !(a->skip_row0 && row == 0) && (uint32_t)tag < (uint64_t)n_tags;
            if (live) { id = row + 1; pv = (uint32_t)((int64_t)s * n_tags + tag); }
        }
        owner_ids[e] = id;
        if (payload != nullptr) payload[e] = pv;
    }
}

// order[i] names an inbox entry e = s * cap + j; what the reduction wants to fetch for it is the upstream row of its (source, tag): row
// s * n_tags + tag[e] of the [world * n_tags, dim] block of (pre-scaled) sample gradients.  In place.
__global__ __launch_bounds__(NRX_BLOCK) void pool_order_remap_kernel(int64_t* __restrict__ order, int64_t n, const int32_t* __restrict__ tag, int64_t cap,
                                                                    int64_t n_tags, int64_t limit) {
    for (int64_t i = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * NRX_BLOCK) {
        const int64_t e = order[i];
        int64_t r = 0;
        if (e >= 0 && e < n) {
            const int64_t t = tag[e];
            r = (e / cap) * n_tags + (t >= 0 && t < n_tags ? t : 0);      // (entries past a block's count carry stale tags: their row is the padding row, never summed)
        }
        order[i] = r < limit ? r : 0;
    }
}

extern "C" int nrx_pool_inbox_owner_ids(int64_t table_rows, int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                                        const int32_t* inbox_rows, const int32_t* inbox_tag, int32_t skip_row0, int32_t* owner_ids, uint32_t* payload,
                                        void* stream) {
    NRX_TRACE();
    PoolArgs a;
    NRX_REQUIRE(owner_ids != nullptr, "nrx_pool_inbox_owner_ids: null owner_ids");
    float* dummy_table = reinterpret_cast<float*>(owner_ids);      // (fill_pool_args wants a table pointer and a weight array: never dereferenced here)
    int32_t ft[NRX_MAX_FEATURES];
    for (int f = 0; f < NRX_MAX_FEATURES; ++f) ft[f] = 0;
    int rc = fill_pool_args(a, &dummy_table, &table_rows, 1, ft, n_feats, batch, world, cap, recv2d, inbox_rows, inbox_tag,
                            reinterpret_cast<const float*>(inbox_tag), 4, "nrx_pool_inbox_owner_ids");
    if (rc != NRX_OK) return rc;
    a.partial = nullptr;
    a.status = nullptr;
    a.skip_row0 = skip_row0;
    int64_t bx = (cap + NRX_BLOCK - 1) / NRX_BLOCK;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(pool_owner_ids_kernel, dim3((unsigned)bx, (unsigned)world), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), a, owner_ids, payload);
    NRX_LAUNCH_CHECK("nrx_pool_inbox_owner_ids");
    return NRX_OK;
}

extern "C" int nrx_pool_order_remap(int64_t* order, int64_t n_entries, const int32_t* inbox_tag, int64_t cap, int64_t n_tags, int32_t world, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(order != nullptr && inbox_tag != nullptr && n_entries >= 0 && cap >= 1 && n_tags >= 1 && world >= 1, "nrx_pool_order_remap: bad argument");
    if (n_entries == 0) return NRX_OK;
    int64_t blocks = (n_entries + NRX_BLOCK - 1) / NRX_BLOCK;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pool_order_remap_kernel, dim3((unsigned)blocks), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), order, n_entries, inbox_tag,
                       cap, n_tags, (int64_t)world * n_tags);
    NRX_LAUNCH_CHECK("nrx_pool_order_remap");
    return NRX_OK;
}

extern "C" int nrx_pool_inbox_expand(int64_t table_rows, int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                                     const int32_t* inbox_rows, const int32_t* inbox_tag, const float* inbox_w, int32_t dim,
                                     const float* g_partial, int32_t skip_row0, int32_t* owner_ids, float* g_rows, void* stream) {
    NRX_TRACE();
    PoolArgs a;
    float* dummy_table = g_rows;                      // (fill_pool_args wants a table pointer: never dereferenced here)
    int32_t ft[NRX_MAX_FEATURES];
    for (int f = 0; f < NRX_MAX_FEATURES; ++f) ft[f] = 0;
    NRX_REQUIRE(owner_ids != nullptr && (g_rows == nullptr || g_partial != nullptr), "nrx_pool_inbox_expand: null buffer");
    NRX_REQUIRE((dim & 3) != 0 || (nrx_aligned16(g_partial) && nrx_aligned16(g_rows)), "nrx_pool_inbox_expand: 16-byte aligned rows");
    if (dummy_table == nullptr) dummy_table = reinterpret_cast<float*>(owner_ids);
    int rc = fill_pool_args(a, &dummy_table, &table_rows, 1, ft, n_feats, batch, world, cap, recv2d, inbox_rows, inbox_tag, inbox_w, dim,
                            "nrx_pool_inbox_expand");
    if (rc != NRX_OK) return rc;
    a.partial = const_cast<float*>(g_partial);
    a.status = nullptr;
    a.skip_row0 = skip_row0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ql = pool_ql(dim);
    const int tb = NRX_BLOCK >> ql;
    const dim3 grid((unsigned)((cap + tb - 1) / tb), (unsigned)world);
    switch (ql) {
#define NRX_CASE(QL_) case QL_: hipLaunchKernelGGL((pool_inbox_expand_kernel<QL_>), grid, dim3(NRX_BLOCK), 0, st, a, owner_ids, g_rows); break;
        NRX_CASE(0) NRX_CASE(1) NRX_CASE(2) NRX_CASE(3) NRX_CASE(4) NRX_CASE(5)
        default: hipLaunchKernelGGL((pool_inbox_expand_kernel<6>), grid, dim3(NRX_BLOCK), 0, st, a, owner_ids, g_rows); break;
#undef NRX_CASE
    }
    NRX_LAUNCH_CHECK("nrx_pool_inbox_expand");
    return NRX_OK;
}

extern "C" int nrx_make_table_keys(const void* const* ids, const int64_t* lens, const int32_t* table_of,
                                   int32_t n_feats, int32_t index_bits, int64_t* keys, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(ids && lens && table_of && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES,
                "nrx_make_table_keys: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_make_table_keys: index_bits must be 32 or 64");
    KeyArgs a;
    int64_t off = 0;
    for (int f = 0; f < n_feats; ++f) {
        NRX_REQUIRE(lens[f] >= 0 && (lens[f] == 0 || ids[f] != nullptr), "nrx_make_table_keys: feature %d: bad ids/len", f);
        NRX_REQUIRE(table_of[f] >= 0 && table_of[f] < (1 << 20), "nrx_make_table_keys: feature %d: bad table index", f);
        a.ids[f] = ids[f];
        a.off[f] = off;
        a.table_of[f] = table_of[f];
        off += lens[f];
    }
    a.off[n_feats] = off;
    a.n_feats = n_feats;
    a.idx64 = index_bits == 64;
    a.n_total = off;
    a.keys = keys;
    if (off == 0) return NRX_OK;
    NRX_REQUIRE(keys != nullptr, "nrx_make_table_keys: null keys buffer");
    int64_t g = (off + NRX_BLOCK - 1) / NRX_BLOCK;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(make_keys_kernel, dim3((unsigned)g), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), a);
    NRX_LAUNCH_CHECK("nrx_make_table_keys");
    return NRX_OK;
}
