// Exact inner-product top-k (brute force, like faiss.IndexFlatIP) for the DSSM recall evaluation.
// Reference: src/model/model_utils/TopKSearcher.py:50-84, src/model/recall/DSSM/model.py:182-254.
//
// Shape: Q queries x N items x d (d = 16 for the reference's towers): 2*Q*N*d flop against N*d*4 bytes
// of items that every query re-reads -> compute-bound.  On gfx950 the fp32 MFMA rate equals the packed
// fp32 VALU rate (157 TF), so the contraction runs on the VALU, where the per-lane k-selection lives
// anyway: one thread owns one query (its d floats in registers), item rows are wave-uniform and arrive
// through the scalar cache, v_pk_fma_f32 accumulates even / odd dimensions, and each lane keeps its
// running top-k sorted in registers (insertion only when a score beats the current k-th, which becomes
// rare after the first few hundred items).  Items are split over blockIdx.y so small query counts still
// fill the chip; a second kernel merges the per-split lists.
// Score definition (the oracle restates it): s = fl(e + o), e / o = fp32 fma chains over the even / odd
// dimensions in ascending order.
#include "nrx_common.h"
#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <math.h>

namespace {

constexpr int KMAX = 32;
constexpr int QCAP = 8;          // per-lane candidate queue depth (LDS)

__device__ __forceinline__ int64_t lower_bound(const int64_t* lst, int64_t lo, int64_t hi, int64_t v) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (lst[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
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

using f32x2 = __attribute__((ext_vector_type(2))) float;

// One thread = one query; the item row is the same for every lane of the wave, so it is fetched with
// SCALAR loads (constant address space -> s_load_dwordx8/x16 through the scalar cache) and consumed
// straight from SGPRs by packed fp32 FMAs: v_pk_fma_f32 does the even and the odd dimension of one
// (query, item) pair per lane per issue.  UNROLL items are in flight for ILP.  No LDS, no barriers.
template <int K, int D2, int UNROLL>
__global__ __launch_bounds__(NRX_BLOCK) void topk_partial_kernel(const float* __restrict__ items, int64_t n_items, const float* __restrict__ queries,
                                                                 int64_t n_queries, int k, const int64_t* __restrict__ excl_off,
                                                                 const int64_t* __restrict__ excl_items, int64_t items_per_split,
                                                                 float* __restrict__ p_score, int* __restrict__ p_idx) {
    const int64_t qi = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;
    const bool live = qi < n_queries;
    const int64_t qc = live ? qi : n_queries - 1;
    f32x2 q[D2];
#pragma unroll
    for (int j = 0; j < D2; ++j) q[j] = reinterpret_cast<const f32x2*>(queries + qc * (int64_t)(2 * D2))[j];
    float bs[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bs[j] = -FLT_MAX; bi[j] = -1; }
    const int64_t begin = (int64_t)blockIdx.y * items_per_split;
    const int64_t end = begin + items_per_split < n_items ? begin + items_per_split : n_items;
    // exclusion list (ascending) walked in step with the ascending item scan: next_ex = first excluded
    // position >= the last candidate looked at; advanced lazily, only when a candidate is examined
    int64_t ep = excl_off ? excl_off[qc] : 0;
    const int64_t e1 = excl_off ? excl_off[qc + 1] : 0;
    ep = lower_bound(excl_items, ep, e1, begin);
    int next_ex = ep < e1 ? (int)excl_items[ep] : INT_MAX;
    auto is_excluded = [&](int it) {
        while (next_ex < it) { ++ep; next_ex = ep < e1 ? (int)excl_items[ep] : INT_MAX; }
        return next_ex == it;
    };
    const f32x2 NRX_CONST* itc = (const f32x2 NRX_CONST*)(items);
    // Candidates that beat the (possibly stale) k-th score are appended to a small per-lane queue in LDS
    // and merged into the sorted register list only when some lane's queue could overflow: a wave takes
    // the ~100-instruction insertion path once per ~QCAP hits of its busiest lane instead of on every
    // hit of ANY of its 64 lanes.  Appending in item order + strict '>' insertion keeps the result
    // identical to immediate insertion (ties toward the lower index).
    __shared__ float q_s[QCAP][NRX_BLOCK];
    __shared__ int q_i[QCAP][NRX_BLOCK];
    int cnt = 0;
    float kth = -FLT_MAX;
    auto flush = [&]() {
#pragma unroll 1
        for (int c = 0; c < QCAP; ++c) {
            if (c < cnt) {
                const float cs = q_s[c][threadIdx.x];
                if (cs > bs[K - 1]) topk_insert<K, int>(bs, bi, cs, q_i[c][threadIdx.x]);
            }
        }
        cnt = 0;
        kth = bs[K - 1];
    };
    int64_t i = begin;
    for (; i + UNROLL <= end; i += UNROLL) {
        const f32x2 NRX_CONST* row = itc + i * D2;
        f32x2 acc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc[u] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < D2; ++j)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc[u] = __builtin_elementwise_fma(q[j], row[u * D2 + j], acc[u]);
        float sc[UNROLL];
        bool any = false;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { sc[u] = acc[u].x + acc[u].y; any |= sc[u] > kth; }
        if (__builtin_amdgcn_ballot_w64(any) != 0) {      // wave-uniform branch
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t it = i + u;
                if (sc[u] > kth && !is_excluded((int)it)) {
                    q_s[cnt][threadIdx.x] = sc[u];
                    q_i[cnt][threadIdx.x] = (int)it;
                    ++cnt;
                }
            }
            if (__builtin_amdgcn_ballot_w64(cnt > QCAP - UNROLL) != 0) flush();
        }
    }
    for (; i < end; ++i) {
        const f32x2 NRX_CONST* row = itc + i * D2;
        f32x2 acc = (f32x2){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < D2; ++j) acc = __builtin_elementwise_fma(q[j], row[j], acc);
        const float a0 = acc.x + acc.y;
        if (a0 > kth && !is_excluded((int)i)) {
            q_s[cnt][threadIdx.x] = a0;
            q_i[cnt][threadIdx.x] = (int)i;
            ++cnt;
        }
        if (__builtin_amdgcn_ballot_w64(cnt > QCAP - UNROLL) != 0) flush();
    }
    flush();
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
    const char* env = getenv("NRX_TOPK_BLOCKS");
    const int64_t target = env ? atoll(env) : 2048;
    int64_t s = (target + qblocks - 1) / qblocks;            // aim at >= 2048 blocks (8 per CU)
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
    const int D2 = dim / 2;
    char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int* p_idx = reinterpret_cast<int*>(ws);
    float* p_score = reinterpret_cast<float*>(ws + (size_t)S * n_queries * K * sizeof(int));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((n_queries + NRX_BLOCK - 1) / NRX_BLOCK), (unsigned)S);
#define NRX_TK(K_, D2_, U_) hipLaunchKernelGGL((topk_partial_kernel<K_, D2_, U_>), grid, dim3(NRX_BLOCK), 0, st, items, n_items, queries, \
                                               n_queries, k, excl_offsets, excl_items, per_split, p_score, p_idx)
#define NRX_TK_D(K_)                                                                                            \
    switch (D2) {                                                                                               \
        case 2: NRX_TK(K_, 2, 4); break; case 4: NRX_TK(K_, 4, 4); break; case 6: NRX_TK(K_, 6, 4); break;      \
        case 8: NRX_TK(K_, 8, 4); break; case 16: NRX_TK(K_, 16, 2); break; case 32: NRX_TK(K_, 32, 1); break;  \
        case 64: NRX_TK(K_, 64, 1); break;                                                                      \
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
