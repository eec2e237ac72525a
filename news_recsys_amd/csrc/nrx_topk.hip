// Exact inner-product top-k (brute force, like faiss.IndexFlatIP) for the DSSM recall evaluation.
// Reference: src/model/model_utils/TopKSearcher.py:50-84, src/model/recall/DSSM/model.py:182-254.
//
// Shape: Q queries x N items x d (d = 16 for the reference's towers): 2*Q*N*d flop against N*d*4 bytes
// of items that every query re-reads -> compute-bound.  On gfx950 the fp32 MFMA rate equals the fp32
// VALU rate (157 TF), so the contraction runs on the VALU, where the per-lane k-selection lives anyway:
// one thread owns one query (its d floats in registers), a block shares LDS tiles of items (every lane
// reads the same item row: broadcast ds_read_b128, conflict-free), 4 items in flight per lane for ILP,
// and each lane keeps its running top-k sorted in registers (insertion only when a score beats the
// current k-th, which becomes rare after the first few tiles).  Items are split over blockIdx.y so small
// query counts still fill the chip; a second kernel merges the per-split lists.
#include "nrx_common.h"
#include <float.h>
#include <math.h>

namespace {

constexpr int KMAX = 32;
constexpr int TILE_BYTES = 32 * 1024;

// score = sequential fp32 fma chain over the dimension (k ascending): the oracle restates exactly this
__device__ __forceinline__ float dot4(const float4 q, const float4 v, float a) {
    a = fmaf(q.x, v.x, a);
    a = fmaf(q.y, v.y, a);
    a = fmaf(q.z, v.z, a);
    return fmaf(q.w, v.w, a);
}

__device__ __forceinline__ bool excluded(const int64_t* lst, int64_t n, int64_t item) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t v = lst[mid];
        if (v == item) return true;
        if (v < item) lo = mid + 1; else hi = mid;
    }
    return false;
}

// insert (score, idx) into a descending list of length K kept in registers (static indexing only)
template <int K, typename I>
__device__ __forceinline__ void topk_insert(float (&s)[K], I (&ix)[K], float score, I idx) {
    // ties: an equal score that arrives later (higher index) goes AFTER the existing one
    float cs = score;
    I ci = idx;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const bool take = cs > s[j];
        const float ts = s[j];
        const I ti = ix[j];
        s[j] = take ? cs : ts;
        ix[j] = take ? ci : ti;
        cs = take ? ts : cs;
        ci = take ? ti : ci;
    }
}

template <int K, int D4>
__global__ __launch_bounds__(NRX_BLOCK) void topk_partial_kernel(const float* __restrict__ items, int64_t n_items, const float* __restrict__ queries,
                                                                 int64_t n_queries, int k, const int64_t* __restrict__ excl_off,
                                                                 const int64_t* __restrict__ excl_items, int64_t items_per_split,
                                                                 int tile_items, float* __restrict__ p_score, int* __restrict__ p_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_items = reinterpret_cast<float4*>(smem);          // [tile_items][D4]
    const int64_t qi = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    const bool live = qi < n_queries;
    float4 q[D4];
#pragma unroll
    for (int j = 0; j < D4; ++j) q[j] = live ? reinterpret_cast<const float4*>(queries + qi * (int64_t)(4 * D4))[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    float bs[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bs[j] = -FLT_MAX; bi[j] = -1; }
    const int64_t e0 = (live && excl_off) ? excl_off[qi] : 0;
    const int64_t en = (live && excl_off) ? excl_off[qi + 1] - e0 : 0;
    const int64_t begin = (int64_t)blockIdx.y * items_per_split;
    const int64_t end = begin + items_per_split < n_items ? begin + items_per_split : n_items;
    for (int64_t t0 = begin; t0 < end; t0 += tile_items) {
        const int cur = (int)((end - t0) < tile_items ? (end - t0) : tile_items);
        __syncthreads();
        for (int e = threadIdx.x; e < cur * D4; e += NRX_BLOCK)
            s_items[e] = reinterpret_cast<const float4*>(items + t0 * (int64_t)(4 * D4))[e];
        __syncthreads();
        if (!live) continue;
        int i = 0;
        for (; i + 4 <= cur; i += 4) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int j = 0; j < D4; ++j) {
                const float4 v0 = s_items[(i + 0) * D4 + j], v1 = s_items[(i + 1) * D4 + j];
                const float4 v2 = s_items[(i + 2) * D4 + j], v3 = s_items[(i + 3) * D4 + j];
                a0 = dot4(q[j], v0, a0);
                a1 = dot4(q[j], v1, a1);
                a2 = dot4(q[j], v2, a2);
                a3 = dot4(q[j], v3, a3);
            }
            const float kth = bs[K - 1];
            if (a0 > kth || a1 > kth || a2 > kth || a3 > kth) {
                const float sc[4] = {a0, a1, a2, a3};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t it = t0 + i + u;
                    if (sc[u] > bs[K - 1] && !(en && excluded(excl_items + e0, en, it))) topk_insert<K, int>(bs, bi, sc[u], (int)it);
                }
            }
        }
        for (; i < cur; ++i) {
            float a0 = 0.f;
#pragma unroll
            for (int j = 0; j < D4; ++j) {
                const float4 v0 = s_items[i * D4 + j];
                a0 = dot4(q[j], v0, a0);
            }
            const int64_t it = t0 + i;
            if (a0 > bs[K - 1] && !(en && excluded(excl_items + e0, en, it))) topk_insert<K, int>(bs, bi, a0, (int)it);
        }
    }
    if (live) {
        float* ps = p_score + (qi * gridDim.y + blockIdx.y) * (int64_t)K;
        int* pi = p_idx + (qi * gridDim.y + blockIdx.y) * (int64_t)K;
#pragma unroll
        for (int j = 0; j < K; ++j) { ps[j] = bs[j]; pi[j] = bi[j]; }
    }
}

// merge the per-split lists of one query (each sorted descending; splits cover ascending item ranges)
template <int K>
__global__ __launch_bounds__(NRX_BLOCK) void topk_merge_kernel(const float* __restrict__ p_score, const int* __restrict__ p_idx,
                                                               int64_t n_queries, int n_split, int k,
                                                               int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
    const int64_t qi = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    if (qi >= n_queries) return;
    float bs[K];
    int64_t bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bs[j] = -FLT_MAX; bi[j] = -1; }
    for (int s = 0; s < n_split; ++s) {
        const float* ps = p_score + (qi * n_split + s) * (int64_t)K;
        const int* pi = p_idx + (qi * n_split + s) * (int64_t)K;
        for (int j = 0; j < K; ++j) {
            if (pi[j] < 0 || !(ps[j] > bs[K - 1])) break;      // lists are sorted: nothing further can enter
            topk_insert<K, int64_t>(bs, bi, ps[j], (int64_t)pi[j]);
        }
    }
    for (int j = 0; j < k; ++j) {
        out_idx[qi * k + j] = j < K ? bi[j] : -1;
        out_score[qi * k + j] = j < K ? bs[j] : -FLT_MAX;
    }
}

int choose_splits(int64_t n_items, int64_t n_queries) {
    const int64_t qblocks = n_queries > 0 ? (n_queries + NRX_BLOCK - 1) / NRX_BLOCK : 1;
    int64_t s = (1024 + qblocks - 1) / qblocks;            // aim at >= 1024 blocks (4 per CU)
    const int64_t max_s = (n_items + 4095) / 4096;          // keep >= 4096 items per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

}  // namespace

extern "C" int64_t nrx_topk_workspace(int64_t n_items, int64_t n_queries, int32_t k) {
    if (n_items < 0 || n_queries < 0 || k < 1) return -1;
    const int K = k <= 8 ? 8 : (k <= 16 ? 16 : KMAX);
    return (int64_t)choose_splits(n_items, n_queries) * n_queries * K * (int64_t)(sizeof(float) + sizeof(int)) + 256;
}

extern "C" int nrx_topk_ip(const float* items, int64_t n_items, int32_t dim, const float* queries, int64_t n_queries,
                           int32_t k, const int64_t* excl_offsets, const int64_t* excl_items,
                           int64_t* out_idx, float* out_score, void* workspace, void* stream) {
    NRX_REQUIRE(n_items >= 0 && n_items < 0x7fffffffLL && n_queries >= 0 && k >= 1, "nrx_topk_ip: bad argument");
    if (k > KMAX || (dim & 3) != 0 || dim < 4 || dim > 128) {
        nrx_set_error("nrx_topk_ip: supports k <= %d and dim %% 4 == 0 with 4 <= dim <= 128 (got k=%d dim=%d)", KMAX, k, dim);
        return NRX_ERR_UNSUPPORTED;
    }
    if (n_queries == 0) return NRX_OK;
    NRX_REQUIRE(queries && out_idx && out_score && workspace && (n_items == 0 || items), "nrx_topk_ip: null buffer");
    NRX_REQUIRE(nrx_aligned16(items) && nrx_aligned16(queries), "nrx_topk_ip: items / queries must be 16-byte aligned");
    NRX_REQUIRE((excl_offsets == nullptr) == (excl_items == nullptr) || excl_offsets != nullptr, "nrx_topk_ip: exclusion lists need offsets");
    const int K = k <= 8 ? 8 : (k <= 16 ? 16 : KMAX);
    const int S = choose_splits(n_items, n_queries);
    const int64_t per_split = ((n_items + S - 1) / S + 3) & ~3ll;
    const int D4 = dim / 4;
    const int tile_items = (TILE_BYTES / (dim * 4)) & ~3;
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int* p_idx = reinterpret_cast<int*>(ws);
    float* p_score = reinterpret_cast<float*>(ws + (size_t)S * n_queries * K * sizeof(int));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((n_queries + NRX_BLOCK - 1) / NRX_BLOCK), (unsigned)S);
    const size_t smem = (size_t)tile_items * dim * 4;
#define NRX_TK(K_, D4_) hipLaunchKernelGGL((topk_partial_kernel<K_, D4_>), grid, dim3(NRX_BLOCK), smem, st, items, n_items, queries, \
                                           n_queries, k, excl_offsets, excl_items, per_split, tile_items, p_score, p_idx)
#define NRX_TK_D(K_)                                                                                     \
    switch (D4) {                                                                                        \
        case 1: NRX_TK(K_, 1); break; case 2: NRX_TK(K_, 2); break; case 3: NRX_TK(K_, 3); break;        \
        case 4: NRX_TK(K_, 4); break; case 8: NRX_TK(K_, 8); break; case 16: NRX_TK(K_, 16); break;      \
        case 32: NRX_TK(K_, 32); break;                                                                  \
        default: nrx_set_error("nrx_topk_ip: dim %d not instantiated (4,8,12,16,32,64,128)", dim); return NRX_ERR_UNSUPPORTED; \
    }
    if (K == 8) { NRX_TK_D(8) } else if (K == 16) { NRX_TK_D(16) } else { NRX_TK_D(32) }
    const unsigned mg = (unsigned)((n_queries + NRX_BLOCK - 1) / NRX_BLOCK);
    if (K == 8) hipLaunchKernelGGL(topk_merge_kernel<8>, dim3(mg), dim3(NRX_BLOCK), 0, st, p_score, p_idx, n_queries, S, k, out_idx, out_score);
    else if (K == 16) hipLaunchKernelGGL(topk_merge_kernel<16>, dim3(mg), dim3(NRX_BLOCK), 0, st, p_score, p_idx, n_queries, S, k, out_idx, out_score);
    else hipLaunchKernelGGL(topk_merge_kernel<32>, dim3(mg), dim3(NRX_BLOCK), 0, st, p_score, p_idx, n_queries, S, k, out_idx, out_score);
    NRX_LAUNCH_CHECK("nrx_topk_ip");
    return NRX_OK;
}
