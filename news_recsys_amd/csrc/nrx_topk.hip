// Exact inner-product top-k (brute force, like faiss.IndexFlatIP) for the DSSM recall evaluation.
// Reference: src/model/model_utils/TopKSearcher.py:50-84, src/model/recall/DSSM/model.py:182-254.
//
// Shape: Q queries x N items x d (d = 16 for the reference's towers): 2*Q*N*d flop against N*d*4 bytes
// of items that every query re-reads -> compute-bound, GEMM-shaped -> the matrix cores.
// One wavefront owns 32 queries.  v_mfma_f32_32x32x2_f32 with A = a 32-item tile, B = the wave's 32
// queries (kept in registers for the whole scan) leaves, in every lane, 16 scores of ONE query
// (C layout: column = lane & 31 = query, rows = items), so the k-selection is lane-local: each lane keeps a
// sorted top-k of the item rows it sees (the two lanes of a query and the item splits over blockIdx.y are
// merged by a second kernel).  Candidates that beat the lane's (possibly stale) k-th score go to a small
// per-lane LDS queue; the ~100-instruction sorted insertion runs only when some lane's queue could overflow.
// The user's history is a per-query ascending exclusion list walked in step with the ascending item scan.
// VALU work (compare / queue) of one wave overlaps the MFMAs of the others.
//
// Score definition (measured: tools/mfma_f32_probe.hip -> profiles/r01_mfma_f32_semantics.txt, the MFMA is
// the fused chain fma(a1,b1, fma(a0,b0,c)) bit for bit): with P = dim rounded up to 8, 16, 32, 64 or 128 (zero
// padded) and H = P/2,   s = 0;  for j in 0..H-1:  s = fma(it[j], q[j], s);  s = fma(it[H+j], q[H+j], s).
// Ties: lower item index first (total order (score desc, index asc), so the result does not depend on
// how items are split over lanes and blocks).  The oracle restates exactly this.
#include "nrx_common.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int KMAX = 32;
constexpr int FLUSH_AT = 6;       // merge the queues once some lane holds more than this many candidates
constexpr int QCAP = FLUSH_AT + 16;   // one tile can add 16 per lane before the next check
constexpr int WAVES = NRX_BLOCK / 64;

__device__ __forceinline__ int64_t lower_bound(const int64_t* lst, int64_t lo, int64_t hi, int64_t v) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (lst[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// insert into a list sorted by (score desc, index asc), kept in registers (static indexing only)
template <int K, typename I>
__device__ __forceinline__ void topk_insert(float (&s)[K], I (&ix)[K], float score, I idx) {
    float cs = score;
    I ci = idx;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const bool take = cs > s[j] || (cs == s[j] && ci < ix[j]);
        const float ts = s[j];
        const I ti = ix[j];
        s[j] = take ? cs : ts;
        ix[j] = take ? ci : ti;
        cs = take ? ts : cs;
        ci = take ? ti : ci;
    }
}

// H4 = float4 loads per lane per item row (= padded_dim / 8).
// Measured on gfx950 (tools/mfma_peak_probe.hip -> profiles/r01_mfma_f32_valu_overlap_probe.txt): the fp32 MFMA
// reaches 155 TF even as one dependent chain, but VALU instructions do NOT overlap with it -- from the same
// wave or from others, every VALU instruction costs its 4 cycles of matrix time.  So the loop is written
// for the fewest VALU instructions per MFMA: a v_max3 tree (8 per 8 MFMAs) decides whether a tile has any
// candidate, its inner nodes are reused to find which registers do, the queue address doubles as the
// counter, and the two A-fragment register sets ping-pong without copies.
template <int K, int H4, bool PAD, int MINB>
__global__ __launch_bounds__(NRX_BLOCK, MINB) void topk_mfma_kernel(const float* __restrict__ items, int64_t n_items, int dim,
                                                              const float* __restrict__ queries, int64_t n_queries,
                                                              const int64_t* __restrict__ excl_off, const int64_t* __restrict__ excl_items,
                                                              int64_t items_per_split, float* __restrict__ p_score, int* __restrict__ p_idx, float kth0) {
    constexpr int H = 4 * H4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int64_t qi = ((int64_t)blockIdx.x * WAVES + wid) * 32 + l31;
    const bool live = qi < n_queries;
    const int64_t qc = live ? qi : n_queries - 1;
    const int e_base = hi * H;                      // first element of this lane's half of the (padded) row

    float qf[H];
#pragma unroll
    for (int v = 0; v < H4; ++v) {
        const bool ok = live && e_base + 4 * v < dim;
        const float4 t = nrx_ldg4(queries + qc * dim + (ok ? e_base + 4 * v : 0), 0);
        qf[4 * v + 0] = ok ? t.x : 0.f; qf[4 * v + 1] = ok ? t.y : 0.f;
        qf[4 * v + 2] = ok ? t.z : 0.f; qf[4 * v + 3] = ok ? t.w : 0.f;
    }
    float bs[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bs[j] = -FLT_MAX; bi[j] = -1; }

    const int begin = (int)((int64_t)blockIdx.y * items_per_split);
    const int end = (int)(begin + items_per_split < n_items ? begin + items_per_split : n_items);
    int64_t ep = excl_off ? excl_off[qc] : 0;
    const int64_t e1 = excl_off ? excl_off[qc + 1] : 0;
    ep = lower_bound(excl_items, ep, e1, begin);
    int next_ex = ep < e1 ? (int)excl_items[ep] : INT_MAX;
    auto is_excluded = [&](int it) {
        while (next_ex < it) { ++ep; next_ex = ep < e1 ? (int)excl_items[ep] : INT_MAX; }
        return next_ex == it;
    };

    // per-lane candidate queue in LDS: slot c of this lane lives at q_s[c][tid]; `qpos` (a byte address
    // relative to q_s) is the next free slot, so it is also the count
    __shared__ float q_s[QCAP][NRX_BLOCK];
    __shared__ int q_i[QCAP][NRX_BLOCK];
    constexpr int SLOT = NRX_BLOCK * 4;
    const int qbase = threadIdx.x * 4;
    int qpos = qbase;
    const int qflush = qbase + FLUSH_AT * SLOT;
    char* const q_s_b = reinterpret_cast<char*>(&q_s[0][0]);
    char* const q_i_b = reinterpret_cast<char*>(&q_i[0][0]);
    float kth = kth0;
    // queue entries are in ascending item order, so the exclusion walk happens here, once per candidate;
    // inside one lane a later entry has the higher index, so strict '>' keeps ties in index order
    auto flush = [&]() {
#pragma unroll 1
        for (int c = qbase; __builtin_amdgcn_ballot_w64(c < qpos) != 0; c += SLOT) {
            if (c < qpos) {
                const float cs = *reinterpret_cast<const float*>(q_s_b + c);
                const int it = *reinterpret_cast<const int*>(q_i_b + c);
                if (cs > bs[K - 1] && it < end && !is_excluded(it)) {
                    float ns = cs;
                    int ni = it;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const bool take = ns > bs[j];
                        const float ts = bs[j];
                        const int ti = bi[j];
                        bs[j] = take ? ns : ts;
                        bi[j] = take ? ni : ti;
                        ns = take ? ts : ns;
                        ni = take ? ti : ni;
                    }
                }
            }
        }
        qpos = qbase;
        kth = bs[K - 1];
    };

    // A fragment of one tile: lane (l31, hi) holds elements [hi*H, hi*H + H) of item row (i0 + l31).
    // Rows past the end are clamped to the last row (their scores are dropped at flush by `it < end`);
    // only a padded dimension (dim % 8 != 0) needs zero selects.
    const float* const ibase = items + e_base;
    const int last_row = (int)n_items - 1;
    auto load_tile = [&](float (&a)[H], int i0) {
        const int r = min(i0 + l31, last_row);
        const float* rp = ibase + (int64_t)r * dim;
#pragma unroll
        for (int v = 0; v < H4; ++v) {
            const bool ok = !PAD || e_base + 4 * v < dim;
            const float4 t = nrx_ldg4(rp + (ok ? 4 * v : -e_base), 0);
            a[4 * v + 0] = ok ? t.x : 0.f; a[4 * v + 1] = ok ? t.y : 0.f;
            a[4 * v + 2] = ok ? t.z : 0.f; a[4 * v + 3] = ok ? t.w : 0.f;
        }
    };

    auto push = [&](float sc, int it) {
        if (sc > kth) {
            *reinterpret_cast<float*>(q_s_b + qpos) = sc;
            *reinterpret_cast<int*>(q_i_b + qpos) = it;
            qpos += SLOT;
        }
    };

    auto step = [&](const float (&a)[H], int i0) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < H; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], qf[s], acc, 0, 0, 0);
        // v_max3 tree; the first-level nodes are reused below
        const float n0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]), n1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]);
        const float n2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]), n3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]);
        const float n4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]);
        const float m = fmaxf(fmaxf(fmaxf(n0, n1), n2), fmaxf(fmaxf(n3, n4), acc[15]));
        if (__builtin_amdgcn_ballot_w64(m > kth) != 0) {           // wave-uniform
            const int ib = i0 + 4 * hi;                            // item of register r: ib + (r & 3) + 8 * (r >> 2)
#define NRX_ROW(r_) (ib + ((r_) & 3) + 8 * ((r_) >> 2))
            if (n0 > kth) { push(acc[0], NRX_ROW(0)); push(acc[1], NRX_ROW(1)); push(acc[2], NRX_ROW(2)); }
            if (n1 > kth) { push(acc[3], NRX_ROW(3)); push(acc[4], NRX_ROW(4)); push(acc[5], NRX_ROW(5)); }
            if (n2 > kth) { push(acc[6], NRX_ROW(6)); push(acc[7], NRX_ROW(7)); push(acc[8], NRX_ROW(8)); }
            if (n3 > kth) { push(acc[9], NRX_ROW(9)); push(acc[10], NRX_ROW(10)); push(acc[11], NRX_ROW(11)); }
            if (n4 > kth) { push(acc[12], NRX_ROW(12)); push(acc[13], NRX_ROW(13)); push(acc[14], NRX_ROW(14)); }
            push(acc[15], NRX_ROW(15));
#undef NRX_ROW
            if (__builtin_amdgcn_ballot_w64(qpos > qflush) != 0) flush();
        }
    };

    // two A-fragment register sets ping-pong: the loads of tile t+1 are in flight during tile t's MFMAs.
    // (Software-pipelining the selection of tile t-1 under tile t's chain was tried: slower, 4.6 vs 4.2 ms.)
    if (begin < end) {
        float fa[H], fb[H];
        load_tile(fa, begin);
        for (int i0 = begin; i0 < end; i0 += 64) {
            load_tile(fb, i0 + 32);
            step(fa, i0);
            load_tile(fa, i0 + 64);
            if (i0 + 32 < end) step(fb, i0 + 32);
        }
    }
    flush();
    if (live) {
        const int64_t slot = (qi * gridDim.y + blockIdx.y) * 2 + hi;
        float* ps = p_score + slot * K;
        int* pi = p_idx + slot * K;
#pragma unroll
        for (int j = 0; j < K; ++j) { ps[j] = bs[j]; pi[j] = bi[j]; }
    }
}

// merge the partial lists of one query (each sorted by (score desc, index asc))
template <int K>
__global__ __launch_bounds__(NRX_BLOCK) void topk_merge_kernel(const float* __restrict__ p_score, const int* __restrict__ p_idx,
                                                               int64_t n_queries, int n_lists, int k,
                                                               int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
    const int64_t qi = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    if (qi >= n_queries) return;
    float bs[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bs[j] = -FLT_MAX; bi[j] = -1; }
    for (int s = 0; s < n_lists; ++s) {
        const float* ps = p_score + (qi * n_lists + s) * (int64_t)K;
        const int* pi = p_idx + (qi * n_lists + s) * (int64_t)K;
        for (int j = 0; j < K; ++j) {
            const float cs = ps[j];
            const int ci = pi[j];
            if (ci < 0 || !(cs > bs[K - 1] || (cs == bs[K - 1] && ci < bi[K - 1]))) break;   // sorted: nothing further can enter
            topk_insert<K, int>(bs, bi, cs, ci);
        }
    }
    for (int j = 0; j < k; ++j) {
        out_idx[qi * k + j] = (int64_t)bi[j];
        out_score[qi * k + j] = bs[j];
    }
}

// list length instantiated for a requested k (10 is what the reference's hit_rate asks for)
int pick_k(int k) { return k <= 4 ? 4 : (k <= 10 ? 10 : (k <= 16 ? 16 : KMAX)); }

int choose_splits(int64_t n_items, int64_t n_queries) {
    const int64_t qblocks = n_queries > 0 ? (n_queries + 32 * WAVES - 1) / (32 * WAVES) : 1;
    int64_t s = (512 + qblocks - 1) / qblocks;             // aim at >= 512 blocks (2 waves per SIMD: MFMA-bound,
                                                              // and every extra split repeats the top-k warm-up)
    const int64_t max_s = (n_items + 4095) / 4096;          // keep >= 4096 items per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

}  // namespace

extern "C" int64_t nrx_topk_workspace(int64_t n_items, int64_t n_queries, int32_t k) {
    if (n_items < 0 || n_queries < 0 || k < 1) return -1;
    const int K = pick_k(k);
    return 2 * (int64_t)choose_splits(n_items, n_queries) * n_queries * K * (int64_t)(sizeof(float) + sizeof(int)) + 256;
}

extern "C" int nrx_topk_ip(const float* items, int64_t n_items, int32_t dim, const float* queries, int64_t n_queries,
                           int32_t k, const int64_t* excl_offsets, const int64_t* excl_items,
                           int64_t* out_idx, float* out_score, void* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(n_items >= 0 && n_items < 0x7fffffffLL && n_queries >= 0 && k >= 1, "nrx_topk_ip: bad argument");
    if (k > KMAX || (dim & 3) != 0 || dim < 4 || dim > 128) {
        nrx_set_error("nrx_topk_ip: supports k <= %d and dim %% 4 == 0 with 4 <= dim <= 128 (got k=%d dim=%d)", KMAX, k, dim);
        return NRX_ERR_UNSUPPORTED;
    }
    if (n_queries == 0) return NRX_OK;
    NRX_REQUIRE(queries && out_idx && out_score && workspace && (n_items == 0 || items), "nrx_topk_ip: null buffer");
    NRX_REQUIRE(nrx_aligned16(items) && nrx_aligned16(queries), "nrx_topk_ip: items / queries must be 16-byte aligned");
    NRX_REQUIRE(excl_offsets == nullptr || excl_items != nullptr, "nrx_topk_ip: exclusion offsets without items");
    const int K = pick_k(k);
    const int S = choose_splits(n_items, n_queries);
    const int64_t per_split = ((n_items + S - 1) / S + 63) & ~63ll;
    const int H4 = dim <= 8 ? 1 : (dim <= 16 ? 2 : (dim <= 32 ? 4 : (dim <= 64 ? 8 : 16)));
    // the row is padded to 8 H4 elements: every dim that is not exactly that wide needs the zero selects.  (Was `dim % 8 != 0`: a 40-wide row in
    // the 64-wide form read 24 floats past its end unselected -- the next items' values times the query's zeros, harmless, except behind the LAST
    // item, where it is whatever lies past the tensor: a NaN there made that item's score NaN (tests/test_topk_retrieval.py's property sweep
    // failed once in five full-suite runs on N = 1, dim = 40), and the read could fault at the end of a mapping.)
    const bool pad = dim != 8 * H4;
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int* p_idx = reinterpret_cast<int*>(ws);
    float* p_score = reinterpret_cast<float*>(ws + (size_t)2 * S * n_queries * K * sizeof(int));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((n_queries + 32 * WAVES - 1) / (32 * WAVES)), (unsigned)S);
    const float kth0 = -FLT_MAX;      // initial threshold (FLT_MAX here disables selection: how the MFMA-only time in DESIGN.md was taken)
#define NRX_TK2(K_, H4_, PAD_) hipLaunchKernelGGL((topk_mfma_kernel<K_, H4_, PAD_, 2>), grid, dim3(NRX_BLOCK), 0, st, items, n_items, (int)dim, \
                                                  queries, n_queries, excl_offsets, excl_items, per_split, p_score, p_idx, kth0)
#define NRX_TK(K_, H4_) if (pad) NRX_TK2(K_, H4_, true); else NRX_TK2(K_, H4_, false)
#define NRX_TK_D(K_)                                                                \
    switch (H4) {                                                                   \
        case 1: NRX_TK(K_, 1); break; case 2: NRX_TK(K_, 2); break;                 \
        case 4: NRX_TK(K_, 4); break; case 8: NRX_TK(K_, 8); break;                 \
        default: NRX_TK(K_, 16); break;                                             \
    }
    const unsigned mg = (unsigned)((n_queries + NRX_BLOCK - 1) / NRX_BLOCK);
#define NRX_MERGE(K_) hipLaunchKernelGGL(topk_merge_kernel<K_>, dim3(mg), dim3(NRX_BLOCK), 0, st, p_score, p_idx, n_queries, 2 * S, k, out_idx, out_score)
    switch (K) {
        case 4: NRX_TK_D(4) NRX_MERGE(4); break;
        case 10: NRX_TK_D(10) NRX_MERGE(10); break;
        case 16: NRX_TK_D(16) NRX_MERGE(16); break;
        default: NRX_TK_D(32) NRX_MERGE(32); break;
    }
    NRX_LAUNCH_CHECK("nrx_topk_ip");
    return NRX_OK;
}
