// Dev probe (round 5, asked by the round-4 review before any integration): can an L2-resident row BITMAP replace the two-pass radix sort of the
// backward planner where the tables' bitmaps are small (C2: 26 x 1 M rows = 128 KB per table and bitmap)?  Four kernels, timed one by one:
//   mark   every lookup ORs its row's bit into `touched`; a lookup that finds the bit set ORs `multi`; one that finds that set too ORs `three`
//          (a saturating per-row count 1 / 2 / 3+ from three cascaded atomics; nothing depends on who arrives first)
//   rank   one pass over the bitmap words in ticket order: popcount prefix of `touched` (= the unique index of every row, ascending by
//          (table, row) for free), per-table bounds, number of 3+ rows per table; leaves {prefix, touched, multi, three} per word and clears the bitmaps
//   emit   every lookup reads its word's entry: unique index u; row looked up once -> dest[p] = u; twice -> dest[p] = -2 - u and cand[u] = p (either
//          of the two lookups may win the word: a + b == b + a); 3+ (or row 0) -> appended to its table's list as (u << 32 | p)
//   sort3  one block per table sorts its list (rank sort / bitonic in LDS / bitonic in place beyond 8192 entries), writes order / seg_start / walk
// Validated against a CPU plan.  Build: hipcc --offload-arch=gfx950 -O3 tools/bitmap_plan_probe.hip -o tools/bin/bitmap_plan_probe
// Usage: bitmap_plan_probe [uniform|zipf] [n_tables=26] [rows=1000000] [batch=65536]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int MAXT = 64;
constexpr int TILE = 2048;              // lookups per block of mark / emit
constexpr int CHUNK = 2048;             // bitmap words per block of rank
constexpr int SORT_THREADS = 1024;
constexpr int SORT_LDS = 8192;          // keys a block sorts in LDS

struct Args {
    const int64_t* ids[MAXT];           // per feature (one table per feature in the probe)
    int64_t rows[MAXT];
    int64_t word_off[MAXT + 1];         // first bitmap word of table t (multiple of CHUNK)
    int32_t chunk_off[MAXT + 1];
    int64_t list_off[MAXT + 1];         // first list entry of table t
    int32_t n_tables, n_chunks, xcd;
    int64_t batch;
    uint32_t *touched, *multi, *three;
    uint4* rank;
    unsigned long long* agg;            // [n_chunks]: bit 63 valid | 3+ rows << 32 | touched rows
    uint32_t* ctl;                      // [0] ticket, [8 + t] list fill of table t, [8 + 64 + t] first walk index of table t
    int64_t *uniq_keys, *counts, *order, *seg_start, *n_walk;
    int32_t *dest, *cand, *walk;
    unsigned long long* list3;
};

__device__ __forceinline__ unsigned xcd_tile(unsigned blk, unsigned grid, int on) {
    if (!on) return blk;
    const unsigned x = blk & 7u, i = blk >> 3, qt = grid >> 3, rt = grid & 7u;
    return x * qt + (x < rt ? x : rt) + i;
}

template <int SCOPE>
__global__ __launch_bounds__(256) void mark_kernel(const Args a) {
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x, a.xcd);
    const unsigned tiles_per = (unsigned)((a.batch + TILE - 1) / TILE);
    const int f = tile / tiles_per;
    const int64_t b0 = (int64_t)(tile - f * tiles_per) * TILE + threadIdx.x;
    const int64_t rows = a.rows[f];
    uint32_t* tw = a.touched + a.word_off[f];
    uint32_t* mw = a.multi + a.word_off[f];
    uint32_t* hw = a.three + a.word_off[f];
    int64_t id[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t b = b0 + j * 256;
        id[j] = b < a.batch ? a.ids[f][b] : -1;
    }
    uint32_t old[8], bit[8], w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t b = b0 + j * 256;
        int64_t r = id[j];
        if (r < 0 || r >= rows) r = 0;
        w[j] = (uint32_t)(r >> 5);
        bit[j] = 1u << (r & 31);
        old[j] = 0;
        if (b < a.batch) old[j] = __hip_atomic_fetch_or(&tw[w[j]], bit[j], __ATOMIC_RELAXED, SCOPE);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (old[j] & bit[j]) {
            const uint32_t o2 = __hip_atomic_fetch_or(&mw[w[j]], bit[j], __ATOMIC_RELAXED, SCOPE);
            if (o2 & bit[j]) __hip_atomic_fetch_or(&hw[w[j]], bit[j], __ATOMIC_RELAXED, SCOPE);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 8 + 2 * MAXT) a.ctl[threadIdx.x] = 0;      // the rank pass's ticket, the list fills
}

__global__ __launch_bounds__(256) void rank_kernel(const Args a) {
    __shared__ uint32_t s_scan[2][4];
    __shared__ uint32_t s_ticket;
    __shared__ unsigned long long s_red[4];
    if (threadIdx.x == 0) s_ticket = atomicAdd(&a.ctl[0], 1u);
    __syncthreads();
    const int c = (int)s_ticket;
    int t = 0;
    for (int i = 1; i < a.n_tables; ++i) t += c >= a.chunk_off[i] ? 1 : 0;
    const int64_t w0 = (int64_t)c * CHUNK + threadIdx.x * 8;
    uint4 tv[2], mv[2], hv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        tv[h] = reinterpret_cast<const uint4*>(a.touched + w0)[h];
        mv[h] = reinterpret_cast<const uint4*>(a.multi + w0)[h];
        hv[h] = reinterpret_cast<const uint4*>(a.three + w0)[h];
    }
    uint32_t tw[8] = {tv[0].x, tv[0].y, tv[0].z, tv[0].w, tv[1].x, tv[1].y, tv[1].z, tv[1].w};
    uint32_t mw[8] = {mv[0].x, mv[0].y, mv[0].z, mv[0].w, mv[1].x, mv[1].y, mv[1].z, mv[1].w};
    uint32_t hw[8] = {hv[0].x, hv[0].y, hv[0].z, hv[0].w, hv[1].x, hv[1].y, hv[1].z, hv[1].w};
    if (c == a.chunk_off[t] && threadIdx.x == 0 && (tw[0] & 1u)) {      // the padding row never trains and is never placed: it goes to the list
        mw[0] |= 1u;
        hw[0] |= 1u;
    }
    uint32_t ct = 0, c3 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ct += __popc(tw[j]);
        c3 += __popc(hw[j]);
    }
    // block scan of (ct, c3) packed in one 64-bit word
    unsigned long long v = ((unsigned long long)c3 << 32) | ct, incl = v;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_red[wid] = incl;
    __syncthreads();
    unsigned long long wbase = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < wid) wbase += s_red[i];
        total += s_red[i];
    }
    const unsigned long long excl = wbase + incl - v;
    if (threadIdx.x == 0) __hip_atomic_store(&a.agg[c], total | (1ull << 63), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // every earlier chunk's totals (all of them published by blocks that hold earlier tickets: they have started)
    unsigned long long before = 0;
    for (int j = threadIdx.x; j < c; j += 256) {
        unsigned long long x;
        do { x = __hip_atomic_load(&a.agg[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((x >> 63) == 0);
        before += x & ~(1ull << 63);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    __syncthreads();
    if (lane == 0) s_red[wid] = before;
    __syncthreads();
    before = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const uint32_t ubase = (uint32_t)before, b3 = (uint32_t)(before >> 32);
    uint32_t u = ubase + (uint32_t)excl;
    uint4* out = a.rank + w0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        out[j] = make_uint4(u, tw[j], mw[j], hw[j]);
        u += __popc(tw[j]);
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        reinterpret_cast<uint4*>(a.touched + w0)[h] = z;
        reinterpret_cast<uint4*>(a.multi + w0)[h] = z;
        reinterpret_cast<uint4*>(a.three + w0)[h] = z;
    }
    if (threadIdx.x == 0) {
        if (c == a.chunk_off[t]) {
            int tt = t;                                     // tables without rows would share the chunk: none in the probe
            a.counts[1 + tt] = ubase;
            a.ctl[8 + MAXT + tt] = b3;
        }
        if (c == a.n_chunks - 1) {
            a.counts[0] = ubase + (uint32_t)total;
            a.counts[1 + a.n_tables] = ubase + (uint32_t)total;
            a.n_walk[0] = b3 + (uint32_t)(total >> 32);
        }
    }
}

__global__ __launch_bounds__(256) void emit_kernel(const Args a) {
    const unsigned tile = xcd_tile(blockIdx.x, gridDim.x, a.xcd);
    const unsigned tiles_per = (unsigned)((a.batch + TILE - 1) / TILE);
    const int f = tile / tiles_per;
    const int64_t b0 = (int64_t)(tile - f * tiles_per) * TILE + threadIdx.x;
    const int64_t rows = a.rows[f];
    const uint4* rk = a.rank + a.word_off[f];
    const int64_t pbase = (int64_t)f * a.batch;
    const int lane = threadIdx.x & 63;
    int64_t id[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t b = b0 + j * 256;
        id[j] = b < a.batch ? a.ids[f][b] : -1;
    }
    uint4 e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int64_t r = id[j];
        if (r < 0 || r >= rows) r = 0;
        id[j] = r;
        e[j] = rk[r >> 5];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t b = b0 + j * 256;
        const bool live = b < a.batch;
        const uint32_t bit = 1u << (id[j] & 31);
        const uint32_t u = e[j].x + __popc(e[j].y & (bit - 1));
        const bool m = (e[j].z & bit) != 0, h = (e[j].w & bit) != 0;
        const int64_t p = pbase + b;
        if (live) {
            a.uniq_keys[u] = ((int64_t)f << 40) | id[j];
            a.dest[p] = h ? -1 : (m ? -2 - (int32_t)u : (int32_t)u);
            if (m && !h) a.cand[u] = (int32_t)p;
        }
        const unsigned long long bal = __ballot(live && h);
        if (bal != 0ull) {                                   // wave-uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&a.ctl[8 + f], (uint32_t)__popcll(bal));
            base = __shfl(base, 0, 64);
            if (live && h) {
                const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
                a.list3[a.list_off[f] + base + __popcll(bal & lt)] = ((unsigned long long)u << 32) | (unsigned long long)(uint32_t)p;
            }
        }
    }
}

// one block per table: sort its list of (u << 32 | p), then order / seg_start / walk
__global__ __launch_bounds__(SORT_THREADS) void sort3_kernel(const Args a) {
    extern __shared__ unsigned long long s_keys[];
    __shared__ uint32_t s_cnt[SORT_THREADS / 64 + 1];
    const int t = blockIdx.x, tid = threadIdx.x;
    const uint32_t m = a.ctl[8 + t];
    uint32_t pb = 0;
    for (int i = 0; i < t; ++i) pb += a.ctl[8 + i];
    if (m == 0) return;
    unsigned long long* g = a.list3 + a.list_off[t];
    uint32_t N = 1;
    while (N < m) N <<= 1;
    const bool in_lds = N <= SORT_LDS;
    if (in_lds) {
        for (uint32_t i = tid; i < N; i += SORT_THREADS) s_keys[i] = i < m ? g[i] : ~0ull;
        __syncthreads();
        if (m <= 1024) {                 // rank sort: a key's place = the number of smaller keys (all distinct)
            unsigned long long k = tid < m ? s_keys[tid] : ~0ull;
            uint32_t r = 0;
            for (uint32_t i = 0; i < m; ++i) r += s_keys[i] < k ? 1u : 0u;
            __syncthreads();
            if (tid < m) s_keys[r] = k;
            __syncthreads();
        } else {
            // normalised bitonic network: every comparator leaves the smaller key at the lower index (first step of a stage: the mirror partner)
            for (uint32_t k = 2; k <= N; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = tid; i < N; i += SORT_THREADS) {
                        const uint32_t x = j == (k >> 1) ? i ^ (k - 1) : i ^ j;
                        if (x > i) {
                            const unsigned long long ki = s_keys[i], kx = s_keys[x];
                            if (ki > kx) { s_keys[i] = kx; s_keys[x] = ki; }
                        }
                    }
                    __syncthreads();
                }
        }
    } else {                             // in place in global memory, entries past m taken as +inf (never stored)
        for (uint32_t k = 2; k <= N; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < N; i += SORT_THREADS) {
                    const uint32_t x = j == (k >> 1) ? i ^ (k - 1) : i ^ j;
                    if (x > i && x < m) {                  // a partner past m is +inf: nothing moves
                        const unsigned long long ki = g[i], kx = g[x];
                        if (ki > kx) { g[i] = kx; g[x] = ki; }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
    }
    const unsigned long long* src = in_lds ? s_keys : g;
    const uint32_t wb = a.ctl[8 + MAXT + t];
    uint32_t run = 0;
    for (uint32_t i0 = 0; i0 < m; i0 += SORT_THREADS) {
        const uint32_t i = i0 + tid;
        const bool live = i < m;
        const unsigned long long k = live ? src[i] : 0ull;
        const uint32_t u = (uint32_t)(k >> 32);
        const bool head = live && (i == 0 || (uint32_t)(src[i - 1] >> 32) != u);
        const bool tail = live && (i == m - 1 || (uint32_t)(src[i + 1] >> 32) != u);
        const unsigned long long bal = __ballot(head);
        const int lane = tid & 63, wid = tid >> 6;
        if (lane == 0) s_cnt[wid] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = run, tot = 0;
        for (int w = 0; w < SORT_THREADS / 64; ++w) {
            if (w < wid) before += s_cnt[w];
            tot += s_cnt[w];
        }
        const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if (live) a.order[pb + i] = (int64_t)(uint32_t)k;
        if (head) {
            a.walk[wb + before + __popcll(bal & lt)] = (int32_t)u;
            a.seg_start[u] = pb + i;
        }
        if (tail) a.seg_start[u + 1] = pb + i + 1;
        run += tot;
        __syncthreads();
    }
}


// ================================================================================================================================
// Second form (after the first measurement: every SCATTERED L2 access -- atomic or not -- costs one L2 request, ~22 G atomics/s on the whole chip,
// so two random passes per lookup cannot beat two coalesced radix passes): ONE kernel, no scattered global access at all.  A block owns a
// 131 072-row range of one table (three 16 KB bitmaps in LDS) and SCANS every lookup of its table (coalesced; 8 blocks per 1 M-row table read the
// same 512 KB of ids from L2), keeping the lookups of its range in an LDS cache; bitmaps, ranks, the table-major prefix over the blocks (ticket
// order + published totals) and the sort of the 3+ list all happen inside the block.
// ================================================================================================================================
constexpr int PL_THREADS = 1024;
constexpr int PL_RLOG = 17;                         // rows per block
constexpr int PL_WORDS = 1 << (PL_RLOG - 5);        // 4096 words per bitmap
constexpr int PL_CACHE = 10240;                     // lookups of its range a block keeps in LDS (beyond: the phases re-scan the ids)
constexpr int PL_SORT = 4096;                       // 3+ lookups a block sorts in LDS (beyond: in place in global memory)
#define AS4 __attribute__((address_space(4)))

struct PlArgs {
    const int64_t* ids[MAXT];          // per feature
    int64_t off[MAXT + 1];             // flat lookup offset of feature f
    int32_t feat_first[MAXT + 1];      // table t's features: feat_of[feat_first[t] .. feat_first[t + 1])
    int32_t feat_of[MAXT];
    int32_t blk_first[MAXT + 1];       // table t's work items
    int64_t rows[MAXT];                // per table
    int32_t n_tables, n_blocks;
    int64_t batch, n_total;
    uint32_t* ctl;                     // [0] ticket  [1] blocks done  [2] epoch   (zero before the first call; the kernel leaves it ready for the next)
    unsigned long long* agg;           // [n_blocks][3]: (epoch + 1) << 32 | {unique rows, 3+ rows, 3+ lookups} of the block
    int64_t *uniq_keys, *counts, *order, *seg_start, *n_walk, *stats;
    int32_t *dest, *cand, *walk;
    unsigned long long* list3;
    unsigned long long* tstamp;        // [n_blocks][8] debug: wall_clock64 at the phase boundaries
};
#define PL_T(k) do { if (tid == 0 && a->tstamp) a->tstamp[8 * w + (k)] = wall_clock64(); } while (0)

template <typename F>
__device__ __forceinline__ void pl_for_each(const AS4 PlArgs* a, int t, uint32_t rb, bool cached, uint32_t ncache, const unsigned long long* s_cache, F fn) {
    if (cached) {
        for (uint32_t i = threadIdx.x; i < ncache; i += PL_THREADS) {
            const unsigned long long e = s_cache[i];
            fn((uint32_t)(e >> 32), (uint32_t)e);
        }
        return;
    }
    const int64_t rows = a->rows[t];
    for (int fi = a->feat_first[t]; fi < a->feat_first[t + 1]; ++fi) {
        const int f = a->feat_of[fi];
        const int64_t* idp = a->ids[f];
        const int64_t po = a->off[f];
        for (int64_t b = threadIdx.x; b < a->batch; b += PL_THREADS) {
            int64_t r = idp[b];
            if (r < 0 || r >= rows) r = 0;
            if ((uint32_t)(r >> PL_RLOG) == rb) fn((uint32_t)r & ((1u << PL_RLOG) - 1), (uint32_t)(po + b));
        }
    }
}

__global__ __launch_bounds__(PL_THREADS) void plan_lds_kernel(const PlArgs args_in_kernarg) {
    const AS4 PlArgs* a = (const AS4 PlArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* s_t = smem;
    uint32_t* s_m = s_t + PL_WORDS;
    uint32_t* s_h = s_m + PL_WORDS;
    uint32_t* s_pre = s_h + PL_WORDS;
    unsigned long long* s_cache = reinterpret_cast<unsigned long long*>(s_pre + PL_WORDS);
    __shared__ uint32_t s_ticket, s_epoch, s_ncache, s_nlist;
    __shared__ unsigned long long s_red[3][PL_THREADS / 64];
    __shared__ uint32_t s_wsum[PL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        s_ticket = atomicAdd(&a->ctl[0], 1u);
        s_epoch = __hip_atomic_load(&a->ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        s_ncache = 0;
        s_nlist = 0;
    }
    for (int i = tid; i < 3 * PL_WORDS; i += PL_THREADS) s_t[i] = 0;
    __syncthreads();
    const int w = (int)s_ticket;
    const uint32_t mark = s_epoch;
    PL_T(0);
    int t = 0;
    for (int i = 1; i < a->n_tables; ++i) t += w >= a->blk_first[i] ? 1 : 0;
    const uint32_t rb = (uint32_t)(w - a->blk_first[t]);
    const int64_t rows = a->rows[t];
    // ---- P1: scan the table's lookups; mark the rows of this range, keep their lookups.  16-byte loads (two ids), four per thread and round,
    // the next round's requested before this round is looked at: a block has 16 wavefronts and nothing else to hide the latency behind
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    for (int fi = a->feat_first[t]; fi < a->feat_first[t + 1]; ++fi) {
        const int f = a->feat_of[fi];
        const int64_t* idp = a->ids[f];
        const int64_t po = a->off[f];
        const int64_t nv = a->batch >> 1;                     // whole 16-byte vectors (the pointer is 16-byte aligned: host-checked)
        const ll2* vp = reinterpret_cast<const ll2*>(idp);
        constexpr int U = 4;
        ll2 cur[U], nxt[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t v = (int64_t)j * PL_THREADS + tid;
            cur[j] = v < nv ? __builtin_nontemporal_load(vp + v) : ll2{-1, -1};
        }
        for (int64_t v0 = 0; v0 < nv; v0 += U * PL_THREADS) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int64_t v = v0 + (int64_t)(U + j) * PL_THREADS + tid;
                nxt[j] = v < nv ? __builtin_nontemporal_load(vp + v) : ll2{-1, -1};
            }
            // the round's 2 U ids together: range tests, then all the bitmap atomics in flight at once, then ONE wave scan for the cache places
            uint32_t lr[2 * U], old[2 * U];
            bool in[2 * U];
            uint32_t cnt = 0;
#pragma unroll
            for (int k = 0; k < 2 * U; ++k) {
                const int64_t v = v0 + (int64_t)(k >> 1) * PL_THREADS + tid;
                uint64_t ur = (uint64_t)((k & 1) ? cur[k >> 1].y : cur[k >> 1].x);
                ur = ur >= (uint64_t)rows ? 0ull : ur;                 // (negative ids are huge here)
                in[k] = v < nv && (uint32_t)(ur >> PL_RLOG) == rb;
                lr[k] = (uint32_t)ur & ((1u << PL_RLOG) - 1);
                cnt += in[k] ? 1u : 0u;
            }
#pragma unroll
            for (int k = 0; k < 2 * U; ++k) {
                old[k] = 0;
                if (in[k]) old[k] = atomicOr(&s_t[lr[k] >> 5], 1u << (lr[k] & 31));
            }
            uint32_t incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            uint32_t base = 0;
            if (lane == 63 && incl != 0) base = atomicAdd(&s_ncache, incl);
            base = __shfl(base, 63, 64);
            uint32_t pos = base + incl - cnt;
#pragma unroll
            for (int k = 0; k < 2 * U; ++k) {
                if (in[k]) {
                    const int64_t v = v0 + (int64_t)(k >> 1) * PL_THREADS + tid;
                    if (pos < PL_CACHE) s_cache[pos] = ((unsigned long long)lr[k] << 32) | (unsigned long long)(uint32_t)(po + 2 * v + (k & 1));
                    ++pos;
                    const uint32_t bit = 1u << (lr[k] & 31);
                    if (old[k] & bit) {
                        const uint32_t o2 = atomicOr(&s_m[lr[k] >> 5], bit);
                        if (o2 & bit) atomicOr(&s_h[lr[k] >> 5], bit);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) cur[j] = nxt[j];
        }
        if ((a->batch & 1) != 0 && tid == 0) {
            uint64_t ur = (uint64_t)idp[a->batch - 1];
            ur = ur >= (uint64_t)rows ? 0ull : ur;
            if ((uint32_t)(ur >> PL_RLOG) == rb) {
                const uint32_t l = (uint32_t)ur & ((1u << PL_RLOG) - 1), bit = 1u << (l & 31);
                const uint32_t o = atomicOr(&s_t[l >> 5], bit);
                if (o & bit) { const uint32_t o2 = atomicOr(&s_m[l >> 5], bit); if (o2 & bit) atomicOr(&s_h[l >> 5], bit); }
                const uint32_t pos = atomicAdd(&s_ncache, 1u);
                if (pos < PL_CACHE) s_cache[pos] = ((unsigned long long)l << 32) | (unsigned long long)(uint32_t)(po + a->batch - 1);
            }
        }
    }
    __syncthreads();
    PL_T(1);
    const uint32_t ncache = s_ncache;
    const bool cached = ncache <= PL_CACHE;
    if (tid == 0 && rb == 0 && (s_t[0] & 1u)) {              // the padding row is never placed: it goes to the list
        s_m[0] |= 1u;
        s_h[0] |= 1u;
    }
    __syncthreads();
    // ---- P2: ranks inside the block, the block's totals
    uint32_t tw[4], ct = 0, c3 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        tw[j] = s_t[4 * tid + j];
        ct += __popc(tw[j]);
        c3 += __popc(s_h[4 * tid + j]);
    }
    uint32_t c3l = 0;
    pl_for_each(a, t, rb, cached, ncache, s_cache, [&](uint32_t lr, uint32_t) { c3l += (s_h[lr >> 5] >> (lr & 31)) & 1u; });
    uint32_t incl = ct;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c3 += __shfl_xor(c3, off, 64);
        c3l += __shfl_xor(c3l, off, 64);
    }
    if (lane == 63) s_wsum[wid] = incl;
    if (lane == 0) {
        s_red[1][wid] = c3;
        s_red[2][wid] = c3l;
    }
    __syncthreads();
    uint32_t wbase = 0, T_u = 0, T_3r = 0, T_3l = 0;
#pragma unroll
    for (int i = 0; i < PL_THREADS / 64; ++i) {
        if (i < wid) wbase += s_wsum[i];
        T_u += s_wsum[i];
        T_3r += (uint32_t)s_red[1][i];
        T_3l += (uint32_t)s_red[2][i];
    }
    {
        uint32_t run = wbase + incl - ct;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_pre[4 * tid + j] = run;
            run += __popc(tw[j]);
        }
    }
    PL_T(2);
    if (tid < 3) {
        const uint32_t v = tid == 0 ? T_u : (tid == 1 ? T_3r : T_3l);
        __hip_atomic_store(&a->agg[3 * w + tid], ((unsigned long long)mark << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the blocks before this one (earlier tickets: all started) -- their totals
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    for (int j = tid; j < 3 * w; j += PL_THREADS) {
        unsigned long long x;
        do { x = __hip_atomic_load(&a->agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(x >> 32) != mark);
        const uint32_t v = (uint32_t)x;
        const int k = j % 3;
        s0 += k == 0 ? v : 0; s1 += k == 1 ? v : 0; s2 += k == 2 ? v : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64);
    }
    __syncthreads();
    if (lane == 0) { s_red[0][wid] = s0; s_red[1][wid] = s1; s_red[2][wid] = s2; }
    __syncthreads();
    uint32_t base_u = 0, base_w3 = 0, base_o3 = 0;
#pragma unroll
    for (int i = 0; i < PL_THREADS / 64; ++i) {
        base_u += (uint32_t)s_red[0][i]; base_w3 += (uint32_t)s_red[1][i]; base_o3 += (uint32_t)s_red[2][i];
    }
    PL_T(3);
    // ---- P4: emit
    const bool sort_lds = T_3l <= PL_SORT;
    unsigned long long* g3 = a->list3 + base_o3;
    const int64_t key_hi = ((int64_t)t << 40) | ((int64_t)rb << PL_RLOG);
    pl_for_each(a, t, rb, cached, ncache, s_cache, [&](uint32_t lr, uint32_t p) {
        const uint32_t wd = lr >> 5, bit = 1u << (lr & 31);
        const uint32_t u = base_u + s_pre[wd] + __popc(s_t[wd] & (bit - 1));
        const bool m = (s_m[wd] & bit) != 0, h = (s_h[wd] & bit) != 0;
        a->dest[p] = h ? -1 : (m ? -2 - (int32_t)u : (int32_t)u);
        if (m && !h) a->cand[u] = (int32_t)p;
        if (h) g3[atomicAdd(&s_nlist, 1u)] = ((unsigned long long)u << 32) | p;
    });
    if (tid == 0) {
        if (rb == 0) a->counts[1 + t] = base_u;
        if (w == a->n_blocks - 1) {
            a->counts[0] = base_u + T_u;
            a->counts[1 + a->n_tables] = base_u + T_u;
            a->n_walk[0] = base_w3 + T_3r;
            a->stats[0] = base_u + T_u; a->stats[1] = base_w3 + T_3r; a->stats[2] = base_o3 + T_3l;
        }
    }
    __syncthreads();
    PL_T(4);
    // ---- P5: the block's 3+ list in (row, lookup) order -> order / seg_start / walk
    const uint32_t m = T_3l;
    if (m != 0) {
        unsigned long long* s_keys = s_cache;
        uint32_t N = 1;
        while (N < m) N <<= 1;
        if (sort_lds) {
            for (uint32_t i = tid; i < N; i += PL_THREADS) s_keys[i] = i < m ? g3[i] : ~0ull;
            __syncthreads();
            if (m <= PL_THREADS) {
                const unsigned long long k = tid < m ? s_keys[tid] : ~0ull;
                uint32_t r = 0;
                for (uint32_t i = 0; i < m; ++i) r += s_keys[i] < k ? 1u : 0u;
                __syncthreads();
                if (tid < m) s_keys[r] = k;
                __syncthreads();
            } else {
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
        } else {
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
        const unsigned long long* src = sort_lds ? s_keys : g3;
        uint32_t run = 0;
        for (uint32_t i0 = 0; i0 < m; i0 += PL_THREADS) {
            const uint32_t i = i0 + tid;
            const bool live = i < m;
            const unsigned long long k = live ? src[i] : 0ull;
            const uint32_t u = (uint32_t)(k >> 32);
            const bool head = live && (i == 0 || (uint32_t)(src[i - 1] >> 32) != u);
            const bool tail = live && (i == m - 1 || (uint32_t)(src[i + 1] >> 32) != u);
            const unsigned long long bal = __ballot(head);
            if (lane == 0) s_wsum[wid] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = run, tot = 0;
            for (int ww = 0; ww < PL_THREADS / 64; ++ww) {
                if (ww < wid) before += s_wsum[ww];
                tot += s_wsum[ww];
            }
            const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
            if (live) a->order[base_o3 + i] = (int64_t)(uint32_t)k;
            if (head) {
                a->walk[base_w3 + before + __popcll(bal & lt)] = (int32_t)u;
                a->seg_start[u] = base_o3 + i;
            }
            if (tail) a->seg_start[u + 1] = base_o3 + i + 1;
            run += tot;
            __syncthreads();
        }
    }
    // ---- the unique keys of the range in row order (= unique-index order), staged through LDS and written as whole lines
    {
        __syncthreads();
        unsigned long long* s_stage = s_cache;
        constexpr uint32_t CH = PL_CACHE;
        uint32_t pre = s_pre[4 * tid];
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
    PL_T(5);
    // ---- the last block out re-arms the state
    if (tid == 0) {
        if (a->n_total < 0) __threadfence();
        const uint32_t d = atomicAdd(&a->ctl[1], 1u);
        if (d == (uint32_t)a->n_blocks - 1) {
            a->ctl[0] = 0;
            a->ctl[1] = 0;
            __hip_atomic_store(&a->ctl[2], mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

int main(int argc, char** argv) {
    const bool zipf = argc > 1 && strcmp(argv[1], "zipf") == 0;
    const int T = argc > 2 ? atoi(argv[2]) : 26;
    const int64_t rows = argc > 3 ? atoll(argv[3]) : 1000000;
    const int64_t B = argc > 4 ? atoll(argv[4]) : 65536;
    const int64_t n = (int64_t)T * B;
    std::vector<int64_t> ids((size_t)n);
    std::vector<double> cdf;
    if (zipf) {
        cdf.resize((size_t)rows);
        double s = 0;
        for (int64_t r = 1; r < rows; ++r) { s += pow((double)r, -1.05); cdf[(size_t)r] = s; }
        for (int64_t r = 1; r < rows; ++r) cdf[(size_t)r] /= s;
    }
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t h = mix64((uint64_t)i * 7919 + 12345);
        if (!zipf) ids[(size_t)i] = 1 + (int64_t)(h % (uint64_t)(rows - 1));
        else {
            const double x = (double)(h >> 11) * (1.0 / 9007199254740992.0);
            ids[(size_t)i] = (int64_t)(std::lower_bound(cdf.begin() + 1, cdf.end(), x) - cdf.begin());
            if (ids[(size_t)i] >= rows) ids[(size_t)i] = rows - 1;
        }
    }
    for (int t = 0; t < T; ++t) ids[(size_t)t * B + 5] = 0;      // a padding id per table
    Args a;
    memset(&a, 0, sizeof(a));
    a.n_tables = T; a.batch = B; a.xcd = getenv("PROBE_NO_XCD") ? 0 : 1;
    int64_t words = 0;
    for (int t = 0; t < T; ++t) {
        a.rows[t] = rows;
        a.word_off[t] = words;
        a.chunk_off[t] = (int32_t)(words / CHUNK);
        a.list_off[t] = (int64_t)t * B;
        words += ((rows + 31) / 32 + CHUNK - 1) / CHUNK * CHUNK;
    }
    a.word_off[T] = words; a.chunk_off[T] = (int32_t)(words / CHUNK); a.list_off[T] = n;
    a.n_chunks = (int32_t)(words / CHUNK);
    int64_t* d_ids;
    CK(hipMalloc(&d_ids, n * 8));
    CK(hipMemcpy(d_ids, ids.data(), n * 8, hipMemcpyHostToDevice));
    for (int t = 0; t < T; ++t) a.ids[t] = d_ids + (int64_t)t * B;
    CK(hipMalloc(&a.touched, words * 4)); CK(hipMalloc(&a.multi, words * 4)); CK(hipMalloc(&a.three, words * 4));
    CK(hipMemset(a.touched, 0, words * 4)); CK(hipMemset(a.multi, 0, words * 4)); CK(hipMemset(a.three, 0, words * 4));
    CK(hipMalloc(&a.rank, words * 16));
    CK(hipMalloc(&a.agg, a.n_chunks * 8));
    CK(hipMalloc(&a.ctl, (8 + 2 * MAXT) * 4)); CK(hipMemset(a.ctl, 0, (8 + 2 * MAXT) * 4));
    CK(hipMalloc(&a.uniq_keys, n * 8)); CK(hipMalloc(&a.counts, (T + 2) * 8)); CK(hipMalloc(&a.order, n * 8));
    CK(hipMalloc(&a.seg_start, (n + 1) * 8)); CK(hipMalloc(&a.n_walk, 8));
    CK(hipMalloc(&a.dest, n * 4)); CK(hipMalloc(&a.cand, n * 4)); CK(hipMalloc(&a.walk, n * 4)); CK(hipMalloc(&a.list3, n * 8));
    CK(hipMemset(a.seg_start, 0xff, (n + 1) * 8)); CK(hipMemset(a.cand, 0xff, n * 4));
    const unsigned tiles = (unsigned)(T * ((B + TILE - 1) / TILE));
    printf("%s ids, %d tables x %lld rows, batch %lld: %lld lookups, %lld bitmap words (%.2f MB per bitmap), %d chunks, %u tiles\n",
           zipf ? "zipf" : "uniform", T, (long long)rows, (long long)B, (long long)n, (long long)words, words * 4 / 1e6, a.n_chunks, tiles);
    hipEvent_t ev[6];
    for (auto& e : ev) CK(hipEventCreate(&e));
    const int iters = 30;
    double tsum[5] = {0, 0, 0, 0, 0};
    const bool wg = getenv("PROBE_WG_SCOPE") != nullptr;
    for (int it = 0; it < iters + 5; ++it) {
        CK(hipMemsetAsync(a.agg, 0, a.n_chunks * 8, 0));
        CK(hipEventRecord(ev[0]));
        if (wg) hipLaunchKernelGGL(mark_kernel<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(tiles), dim3(256), 0, 0, a);
        else hipLaunchKernelGGL(mark_kernel<__HIP_MEMORY_SCOPE_AGENT>, dim3(tiles), dim3(256), 0, 0, a);
        CK(hipEventRecord(ev[1]));
        hipLaunchKernelGGL(rank_kernel, dim3(a.n_chunks), dim3(256), 0, 0, a);
        CK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL(emit_kernel, dim3(tiles), dim3(256), 0, 0, a);
        CK(hipEventRecord(ev[3]));
        hipLaunchKernelGGL(sort3_kernel, dim3(T), dim3(SORT_THREADS), SORT_LDS * 8, 0, a);
        CK(hipEventRecord(ev[4]));
        CK(hipEventSynchronize(ev[4]));
        CK(hipGetLastError());
        if (it >= 5) {
            for (int k = 0; k < 4; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); tsum[k] += ms; }
            float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[4])); tsum[4] += ms;
        }
    }
    printf("mark %.1f us | rank %.1f us | emit %.1f us | sort3 %.1f us | all four %.1f us  (mean of %d, events around each launch)\n",
           tsum[0] / iters * 1e3, tsum[1] / iters * 1e3, tsum[2] / iters * 1e3, tsum[3] / iters * 1e3, tsum[4] / iters * 1e3, iters);
    auto validate = [&](const char* what) -> int {
    std::vector<int64_t> uk((size_t)n), cnt(T + 2), ord((size_t)n), seg((size_t)n + 1), nw(1);
    std::vector<int32_t> dest((size_t)n), cand((size_t)n), walk((size_t)n);
    CK(hipMemcpy(uk.data(), a.uniq_keys, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(cnt.data(), a.counts, (T + 2) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ord.data(), a.order, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(seg.data(), a.seg_start, (n + 1) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nw.data(), a.n_walk, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(dest.data(), a.dest, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cand.data(), a.cand, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(walk.data(), a.walk, n * 4, hipMemcpyDeviceToHost));
    int64_t bad = 0, u = 0, n_once = 0, n_twice = 0, n_more = 0, l_more = 0, wi = 0, oi = 0;
    for (int t = 0; t < T && bad < 10; ++t) {
        std::vector<std::pair<int64_t, int64_t>> v((size_t)B);
        for (int64_t b = 0; b < B; ++b) v[(size_t)b] = {ids[(size_t)t * B + b], (int64_t)t * B + b};
        std::stable_sort(v.begin(), v.end(), [](auto& x, auto& y) { return x.first < y.first; });
        if (cnt[1 + t] != u) { printf("counts[%d] %lld != %lld\n", 1 + t, (long long)cnt[1 + t], (long long)u); ++bad; }
        for (size_t i = 0; i < v.size();) {
            size_t j = i;
            while (j < v.size() && v[j].first == v[i].first) ++j;
            const int64_t k = (int64_t)(j - i), row = v[i].first;
            if (uk[(size_t)u] != (((int64_t)t << 40) | row)) { if (bad++ < 10) printf("uniq_keys[%lld] wrong\n", (long long)u); }
            if (row != 0 && k == 1) {
                ++n_once;
                if (dest[(size_t)v[i].second] != (int32_t)u) { if (bad++ < 10) printf("dest once wrong at %lld: %d vs %lld\n", (long long)v[i].second, dest[(size_t)v[i].second], (long long)u); }
            } else if (row != 0 && k == 2) {
                ++n_twice;
                const int64_t p1 = v[i].second, p2 = v[i + 1].second;
                if (dest[(size_t)p1] != -2 - (int32_t)u || dest[(size_t)p2] != -2 - (int32_t)u || (cand[(size_t)u] != p1 && cand[(size_t)u] != p2)) { if (bad++ < 10) printf("twice wrong at u %lld\n", (long long)u); }
            } else {
                ++n_more; l_more += k;
                bool ok = walk[(size_t)wi] == (int32_t)u && seg[(size_t)u] == oi && seg[(size_t)u + 1] == oi + k;
                for (int64_t e = 0; e < k && ok; ++e) ok = ord[(size_t)(oi + e)] == v[i + (size_t)e].second && dest[(size_t)v[i + (size_t)e].second] == -1;
                if (!ok) { if (bad++ < 10) printf("walk row wrong at u %lld (k %lld, walk[%lld] = %d, seg %lld..%lld want %lld)\n", (long long)u, (long long)k, (long long)wi, walk[(size_t)wi], (long long)seg[(size_t)u], (long long)seg[(size_t)u + 1], (long long)oi); }
                ++wi; oi += k;
            }
            ++u;
            i = j;
        }
    }
    if (cnt[0] != u || cnt[1 + T] != u || nw[0] != wi) { printf("totals wrong: n_unique %lld vs %lld, n_walk %lld vs %lld\n", (long long)cnt[0], (long long)u, (long long)nw[0], (long long)wi); ++bad; }
    printf("%s: unique rows %lld: once %lld, twice %lld, 3+/padding %lld (%lld lookups)  ->  %s\n", what, (long long)u, (long long)n_once, (long long)n_twice, (long long)n_more,
           (long long)l_more, bad ? "MISMATCH" : "plan matches the CPU plan");
        return (int)bad;
    };
    int rc = validate("four kernels");
    // ---- second form: one kernel
    {
        PlArgs p;
        memset(&p, 0, sizeof(p));
        int nb = 0;
        for (int t = 0; t < T; ++t) {
            p.ids[t] = d_ids + (int64_t)t * B; p.off[t] = (int64_t)t * B; p.feat_first[t] = t; p.feat_of[t] = t; p.rows[t] = rows;
            p.blk_first[t] = nb;
            nb += (int)((rows + (1 << PL_RLOG) - 1) >> PL_RLOG);
        }
        p.off[T] = n; p.feat_first[T] = T; p.blk_first[T] = nb;
        p.n_tables = T; p.n_blocks = nb; p.batch = B; p.n_total = n;
        CK(hipMalloc(&p.ctl, 64)); CK(hipMemset(p.ctl, 0, 64));
        CK(hipMalloc(&p.agg, (size_t)nb * 3 * 8)); CK(hipMemset(p.agg, 0, (size_t)nb * 3 * 8));
        CK(hipMalloc(&p.stats, 64));
        CK(hipMalloc(&p.tstamp, (size_t)nb * 64)); CK(hipMemset(p.tstamp, 0, (size_t)nb * 64));
        p.uniq_keys = a.uniq_keys; p.counts = a.counts; p.order = a.order; p.seg_start = a.seg_start; p.n_walk = a.n_walk;
        p.dest = a.dest; p.cand = a.cand; p.walk = a.walk; p.list3 = a.list3;
        CK(hipMemset(a.uniq_keys, 0xff, n * 8)); CK(hipMemset(a.dest, 0x7f, n * 4)); CK(hipMemset(a.cand, 0xff, n * 4));
        CK(hipMemset(a.seg_start, 0xff, (n + 1) * 8)); CK(hipMemset(a.order, 0xff, n * 8)); CK(hipMemset(a.walk, 0xff, n * 4));
        const size_t lds = (size_t)4 * PL_WORDS * 4 + (size_t)PL_CACHE * 8;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(plan_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        double ts = 0;
        for (int it = 0; it < iters + 5; ++it) {
            CK(hipEventRecord(ev[0]));
            hipLaunchKernelGGL(plan_lds_kernel, dim3(nb), dim3(PL_THREADS), lds, 0, p);
            CK(hipEventRecord(ev[1]));
            CK(hipEventSynchronize(ev[1]));
            CK(hipGetLastError());
            if (it >= 5) { float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1])); ts += ms; }
        }
        printf("ONE kernel (%d blocks x %d threads, %zu KB of LDS): %.1f us\n", nb, PL_THREADS, lds >> 10, ts / iters * 1e3);
        {
            std::vector<unsigned long long> ts_h((size_t)nb * 8);
            CK(hipMemcpy(ts_h.data(), p.tstamp, (size_t)nb * 64, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < nb; ++b) t0 = std::min(t0, ts_h[(size_t)b * 8]);
            const char* names[6] = {"start", "scan+mark", "rank", "lookback", "emit", "sort3+keys"};
            printf("phase ends relative to the first block's start, in us (100 MHz clock): mean / max over the blocks\n");
            for (int k = 0; k < 6; ++k) {
                double sum = 0, mx = 0;
                for (int b = 0; b < nb; ++b) { const double v = (double)(ts_h[(size_t)b * 8 + k] - t0) / 100.0; sum += v; mx = std::max(mx, v); }
                printf("  %-10s %7.2f / %7.2f\n", names[k], sum / nb, mx);
            }
        }
        rc |= validate("one kernel");
    }
    return rc ? 1 : 0;
}
