// Integer utilities of the row-sharded embedding path (new in this build; the reference is
// single-device: every trainer is devices=1, e.g. src/model/sort/deep/train.py:38-44).
// Row r of a table lives on rank (r % world) at local row (r / world).  All results are
// bit-exact and deterministic (no order-dependent atomics): oracle/ref_np.py owner_of /
// bucketize_by_owner / csr_from_mask give the definitions.
#include "nrx_common.h"

namespace {

constexpr int CHUNK = 2048;   // ids per wavefront in the bucketing passes

__device__ __forceinline__ int owner_of(int64_t id, int world) {
    // ids are validated non-negative by the gather; a negative id maps to owner 0 and is
    // reported out-of-range by the owner-side gather.
    return id < 0 ? 0 : (int)(id % world);
}

// pass 1: hist[chunk][o] = #ids of the chunk owned by rank o   (lane o keeps owner o's count)
__global__ __launch_bounds__(NRX_BLOCK) void owner_hist_kernel(const void* __restrict__ ids, bool is64, int64_t n, int world,
                                                               int64_t* __restrict__ hist) {
    const int lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * (NRX_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t begin = chunk * CHUNK;
    if (begin >= n) return;
    const int64_t end = (begin + CHUNK < n) ? begin + CHUNK : n;
    int64_t cnt = 0;
    for (int64_t i0 = begin; i0 < end; i0 += 64) {
        const int64_t i = i0 + lane;
        const int o = (i < end) ? owner_of(nrx_load_id(ids, i, is64), world) : -1;
        for (int t = 0; t < world; ++t) {
            const unsigned long long m = __ballot(o == t);
            if (lane == t) cnt += __popcll(m);
        }
    }
    if (lane < world) hist[chunk * world + lane] = cnt;
}

// pass 2 (one block): per owner exclusive scan over chunks (in place), totals -> counts,
// exclusive scan of totals -> offsets
__global__ __launch_bounds__(NRX_BLOCK) void owner_scan_kernel(int64_t* __restrict__ hist, int64_t nchunks, int world,
                                                               int64_t* __restrict__ counts, int64_t* __restrict__ offsets) {
    __shared__ int64_t s_tot[64];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    for (int o = wid; o < world; o += NRX_BLOCK / 64) {
        int64_t running = 0;
        for (int64_t c0 = 0; c0 < nchunks; c0 += 64) {
            const int64_t c = c0 + lane;
            const int64_t v = (c < nchunks) ? hist[c * world + o] : 0;
            int64_t incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int64_t t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            if (c < nchunks) hist[c * world + o] = running + incl - v;
            running += __shfl(incl, 63, 64);
        }
        if (lane == 0) s_tot[o] = running;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t acc = 0;
        for (int o = 0; o < world; ++o) {
            counts[o] = s_tot[o];
            offsets[o] = acc;
            acc += s_tot[o];
        }
    }
}

// pass 3: stable placement
__global__ __launch_bounds__(NRX_BLOCK) void owner_place_kernel(const void* __restrict__ ids, bool is64, int64_t n, int world,
                                                                const int64_t* __restrict__ hist, const int64_t* __restrict__ offsets,
                                                                int64_t* __restrict__ local_rows, int64_t* __restrict__ slot) {
    const int lane = threadIdx.x & 63;
    const int64_t chunk = (int64_t)blockIdx.x * (NRX_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t begin = chunk * CHUNK;
    if (begin >= n) return;
    const int64_t end = (begin + CHUNK < n) ? begin + CHUNK : n;
    int64_t run = (lane < world) ? hist[chunk * world + lane] + offsets[lane] : 0;   // lane o: next free slot of owner o
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t i0 = begin; i0 < end; i0 += 64) {
        const int64_t i = i0 + lane;
        int64_t id = 0;
        int o = -1;
        if (i < end) {
            id = nrx_load_id(ids, i, is64);
            o = owner_of(id, world);
        }
        int64_t pos = 0;
        for (int t = 0; t < world; ++t) {
            const unsigned long long m = __ballot(o == t);
            const int64_t base = __shfl(run, t, 64);
            if (o == t) pos = base + __popcll(m & lt);
            if (lane == t) run += __popcll(m);
        }
        if (i < end) {
            slot[i] = pos;
            local_rows[pos] = id < 0 ? id : id / world;
        }
    }
}

// Owner-side gather of a flat, table-segmented id list.
struct SegArgs {
    const float* table[NRX_MAX_FEATURES];
    int64_t rows[NRX_MAX_FEATURES];
};

template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void gather_segmented_kernel(const SegArgs a, const int64_t* __restrict__ seg_start,
                                                                     const int32_t* __restrict__ seg_table, int n_seg, int D,
                                                                     const int64_t* __restrict__ local_rows, float* __restrict__ out,
                                                                     int32_t* status, bool vec) {
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t* s_start = reinterpret_cast<int64_t*>(smem);   // [n_seg + 1]
    for (int i = threadIdx.x; i <= n_seg; i += NRX_BLOCK) s_start[i] = seg_start[i];
    __syncthreads();
    const int64_t total = s_start[n_seg];
    const int q = threadIdx.x & (Q - 1);
    const int64_t p = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    if (p >= total) return;
    int lo = 0, hi = n_seg;          // last segment with start <= p
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_start[mid] <= p) lo = mid; else hi = mid;
    }
    const int t = seg_table[lo];
    int64_t row = local_rows[p];
    if ((uint64_t)row >= (uint64_t)a.rows[t]) {
        if (q == 0) nrx_report_oob(status, t, p, row);
        row = 0;
    }
    for (int k0 = q * 4; k0 < D; k0 += 4 * Q) {
        const float* src = a.table[t] + row * (int64_t)D + k0;
        float* dst = out + p * (int64_t)D + k0;
        if (vec) {
            *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        } else {
            dst[0] = src[0];
            if (k0 + 1 < D) dst[1] = src[1];
            if (k0 + 2 < D) dst[2] = src[2];
            if (k0 + 3 < D) dst[3] = src[3];
        }
    }
}

struct SegGradArgs {
    float* table[NRX_MAX_FEATURES];
    int64_t rows[NRX_MAX_FEATURES];
};

template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void scatter_segmented_kernel(const SegGradArgs a, const int64_t* __restrict__ seg_start,
                                                                      const int32_t* __restrict__ seg_table, int n_seg, int D,
                                                                      const int64_t* __restrict__ local_rows,
                                                                      const float* __restrict__ g_rows, bool skip_row0) {
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t* s_start = reinterpret_cast<int64_t*>(smem);
    for (int i = threadIdx.x; i <= n_seg; i += NRX_BLOCK) s_start[i] = seg_start[i];
    __syncthreads();
    const int64_t total = s_start[n_seg];
    const int q = threadIdx.x & (Q - 1);
    const int64_t p = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    if (p >= total) return;
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_start[mid] <= p) lo = mid; else hi = mid;
    }
    const int t = seg_table[lo];
    const int64_t row = local_rows[p];
    if ((uint64_t)row >= (uint64_t)a.rows[t] || (skip_row0 && row == 0)) return;
    for (int k0 = q * 4; k0 < D; k0 += 4 * Q) {
        const float* src = g_rows + p * (int64_t)D + k0;
        float* dst = a.table[t] + row * (int64_t)D + k0;
        unsafeAtomicAdd(dst, src[0]);
        if (k0 + 1 < D) unsafeAtomicAdd(dst + 1, src[1]);
        if (k0 + 2 < D) unsafeAtomicAdd(dst + 2, src[2]);
        if (k0 + 3 < D) unsafeAtomicAdd(dst + 3, src[3]);
    }
}

__global__ __launch_bounds__(NRX_BLOCK) void mask_lengths_kernel(const float* __restrict__ mask, int64_t batch, int L, int64_t* __restrict__ lens) {
    for (int64_t b = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; b < batch; b += (int64_t)gridDim.x * NRX_BLOCK) {
        int64_t c = 0;
        for (int l = 0; l < L; ++l) c += mask[b * (int64_t)L + l] != 0.f;
        lens[b] = c;
    }
}

template <typename T>
__global__ __launch_bounds__(NRX_BLOCK) void csr_to_padded_kernel(const T* __restrict__ values, const int64_t* __restrict__ offsets,
                                                                  const int64_t* __restrict__ rows, int64_t batch, int L,
                                                                  T* __restrict__ ids, float* __restrict__ mask) {
    const int64_t total = batch * (int64_t)L;
    for (int64_t i = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * NRX_BLOCK) {
        const int64_t b = i / L;
        const int l = (int)(i - b * L);
        const int64_t src = rows ? rows[b] : b;         // device-resident dataset: batch row b = dataset row rows[b]
        const int64_t lo = offsets[src];
        const int64_t n = offsets[src + 1] - lo;
        const bool real = l < n;
        ids[i] = real ? values[lo + l] : (T)0;
        mask[i] = real ? 1.0f : 0.0f;
    }
}

// One thread per user segment (segments are short: tens to hundreds of impressions per user in MIND).
__global__ __launch_bounds__(NRX_BLOCK) void user_rank_metrics_kernel(const float* __restrict__ scores, const float* __restrict__ labels,
                                                                      const int64_t* __restrict__ seg_start, int64_t n_users, int k,
                                                                      double* __restrict__ auc, double* __restrict__ ndcg,
                                                                      double* __restrict__ hr, double* __restrict__ mrr) {
    for (int64_t u = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; u < n_users; u += (int64_t)gridDim.x * NRX_BLOCK) {
        const int64_t lo = seg_start[u], hi = seg_start[u + 1];
        double P = 0.0;
        for (int64_t i = lo; i < hi; ++i) P += labels[i] == 1.0f ? 1.0 : 0.0;
        const double N = (double)(hi - lo) - P;
        // AUC with ties at 1/2 (sklearn.roc_auc_score): groups of equal score, descending
        double num = 0.0, neg_above = 0.0;
        for (int64_t i = lo; i < hi;) {
            int64_t j = i;
            double p = 0.0;
            while (j < hi && scores[j] == scores[i]) {
                p += labels[j] == 1.0f ? 1.0 : 0.0;
                ++j;
            }
            const double q = (double)(j - i) - p;
            num += p * (N - neg_above - q + 0.5 * q);
            neg_above += q;
            i = j;
        }
        auc[u] = (P > 0.0 && N > 0.0) ? num / (P * N) : __longlong_as_double(0x7ff8000000000000LL);
        // top-k metrics (base_model.py:381-433)
        double dcg = 0.0, first = 0.0, hit = 0.0;
        const int64_t top = (hi - lo) < k ? (hi - lo) : k;
        for (int64_t r = 1; r <= top; ++r) {
            if (labels[lo + r - 1] == 1.0f) {
                dcg += 1.0 / log2((double)(r + 1));
                if (first == 0.0) first = 1.0 / (double)r;
                hit = 1.0;
            }
        }
        double idcg = 0.0;
        const int64_t ideal = P < (double)k ? (int64_t)P : k;
        for (int64_t r = 1; r <= ideal; ++r) idcg += 1.0 / log2((double)(r + 1));
        const bool any_pos = P > 0.0;
        hr[u] = any_pos ? hit : 0.0;
        ndcg[u] = (any_pos && idcg > 0.0) ? dcg / idcg : 0.0;
        mrr[u] = any_pos ? first : 0.0;
    }
}

int ceil_log2u(int x) {
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

}  // namespace

extern "C" int64_t nrx_bucketize_workspace(int64_t n, int32_t world) {
    if (n < 0 || world < 1) return -1;
    const int64_t nchunks = (n + CHUNK - 1) / CHUNK;
    return nchunks * world + world;
}

extern "C" int nrx_bucketize_by_owner(const void* ids, int32_t index_bits, int64_t n, int32_t world,
                                      int64_t* counts, int64_t* local_rows, int64_t* slot,
                                      int64_t* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(world >= 1 && world <= 64, "nrx_bucketize_by_owner: world must be in [1, 64]");
    NRX_REQUIRE(index_bits == 32 || index_bits == 64, "nrx_bucketize_by_owner: index_bits must be 32 or 64");
    NRX_REQUIRE(n >= 0 && counts && workspace, "nrx_bucketize_by_owner: bad argument");
    NRX_REQUIRE(n == 0 || (ids && local_rows && slot), "nrx_bucketize_by_owner: null buffer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t nchunks = (n + CHUNK - 1) / CHUNK;
    int64_t* hist = workspace;
    int64_t* offsets = workspace + nchunks * world;
    const bool is64 = index_bits == 64;
    const unsigned grid = (unsigned)((nchunks + 3) / 4);
    if (nchunks > 0) hipLaunchKernelGGL(owner_hist_kernel, dim3(grid), dim3(NRX_BLOCK), 0, st, ids, is64, n, world, hist);
    hipLaunchKernelGGL(owner_scan_kernel, dim3(1), dim3(NRX_BLOCK), 0, st, hist, nchunks, world, counts, offsets);
    if (nchunks > 0)
        hipLaunchKernelGGL(owner_place_kernel, dim3(grid), dim3(NRX_BLOCK), 0, st, ids, is64, n, world, hist, offsets, local_rows, slot);
    NRX_LAUNCH_CHECK("nrx_bucketize_by_owner");
    return NRX_OK;
}

extern "C" int nrx_gather_rows_segmented(const float* const* tables, const int64_t* table_rows, int32_t n_tables,
                                         const int64_t* seg_start, const int32_t* seg_table, int32_t n_seg,
                                         int64_t n_rows, int32_t dim, const int64_t* local_rows,
                                         float* out_rows, int32_t* status, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(tables && table_rows && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES,
                "nrx_gather_rows_segmented: n_tables must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(seg_start && seg_table && n_seg >= 1 && n_seg <= 4096, "nrx_gather_rows_segmented: n_seg must be in [1, 4096]");
    NRX_REQUIRE(dim >= 1 && n_rows >= 0, "nrx_gather_rows_segmented: bad argument");
    if (n_rows == 0) return NRX_OK;
    NRX_REQUIRE(local_rows && out_rows, "nrx_gather_rows_segmented: null buffer");
    SegArgs a;
    bool vec = (dim & 3) == 0 && nrx_aligned16(out_rows);
    for (int i = 0; i < n_tables; ++i) {
        NRX_REQUIRE(tables[i] != nullptr, "nrx_gather_rows_segmented: table %d is null", i);
        a.table[i] = tables[i];
        a.rows[i] = table_rows[i];
        vec &= nrx_aligned16(tables[i]);
    }
    int ql = ceil_log2u((dim + 3) / 4);
    if (ql > 6) ql = 6;
    const int tb = NRX_BLOCK >> ql;
    const unsigned grid = (unsigned)((n_rows + tb - 1) / tb);
    const size_t smem = (size_t)(n_seg + 1) * sizeof(int64_t);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (ql) {
#define NRX_CASE(QL_) case QL_: hipLaunchKernelGGL((gather_segmented_kernel<QL_>), dim3(grid), dim3(NRX_BLOCK), smem, st, a, seg_start, seg_table, n_seg, dim, local_rows, out_rows, status, vec); break;
        NRX_CASE(0) NRX_CASE(1) NRX_CASE(2) NRX_CASE(3) NRX_CASE(4) NRX_CASE(5)
        default: hipLaunchKernelGGL((gather_segmented_kernel<6>), dim3(grid), dim3(NRX_BLOCK), smem, st, a, seg_start, seg_table, n_seg, dim, local_rows, out_rows, status, vec); break;
#undef NRX_CASE
    }
    NRX_LAUNCH_CHECK("nrx_gather_rows_segmented");
    return NRX_OK;
}

extern "C" int nrx_scatter_add_rows_segmented(float* const* grad_tables, const int64_t* table_rows, int32_t n_tables,
                                              const int64_t* seg_start, const int32_t* seg_table, int32_t n_seg,
                                              int64_t n_rows, int32_t dim, const int64_t* local_rows,
                                              const float* g_rows, int32_t skip_row0, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(grad_tables && table_rows && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES,
                "nrx_scatter_add_rows_segmented: n_tables must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(seg_start && seg_table && n_seg >= 1 && n_seg <= 4096, "nrx_scatter_add_rows_segmented: n_seg must be in [1, 4096]");
    NRX_REQUIRE(dim >= 1 && n_rows >= 0, "nrx_scatter_add_rows_segmented: bad argument");
    if (n_rows == 0) return NRX_OK;
    NRX_REQUIRE(local_rows && g_rows, "nrx_scatter_add_rows_segmented: null buffer");
    SegGradArgs a;
    for (int i = 0; i < n_tables; ++i) {
        NRX_REQUIRE(grad_tables[i] != nullptr, "nrx_scatter_add_rows_segmented: table %d is null", i);
        a.table[i] = grad_tables[i];
        a.rows[i] = table_rows[i];
    }
    int ql = ceil_log2u((dim + 3) / 4);
    if (ql > 6) ql = 6;
    const int tb = NRX_BLOCK >> ql;
    const unsigned grid = (unsigned)((n_rows + tb - 1) / tb);
    const size_t smem = (size_t)(n_seg + 1) * sizeof(int64_t);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (ql) {
#define NRX_CASE(QL_) case QL_: hipLaunchKernelGGL((scatter_segmented_kernel<QL_>), dim3(grid), dim3(NRX_BLOCK), smem, st, a, seg_start, seg_table, n_seg, dim, local_rows, g_rows, skip_row0 != 0); break;
        NRX_CASE(0) NRX_CASE(1) NRX_CASE(2) NRX_CASE(3) NRX_CASE(4) NRX_CASE(5)
        default: hipLaunchKernelGGL((scatter_segmented_kernel<6>), dim3(grid), dim3(NRX_BLOCK), smem, st, a, seg_start, seg_table, n_seg, dim, local_rows, g_rows, skip_row0 != 0); break;
#undef NRX_CASE
    }
    NRX_LAUNCH_CHECK("nrx_scatter_add_rows_segmented");
    return NRX_OK;
}

extern "C" int nrx_csr_to_padded(const void* values, int32_t value_bits, const int64_t* offsets, const int64_t* rows, int64_t batch,
                                 int32_t bag_len, void* ids_out, float* mask_out, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(value_bits == 32 || value_bits == 64, "nrx_csr_to_padded: value_bits must be 32 or 64");
    NRX_REQUIRE(batch >= 0 && bag_len >= 1, "nrx_csr_to_padded: bad argument");
    if (batch == 0) return NRX_OK;
    NRX_REQUIRE(offsets && ids_out && mask_out, "nrx_csr_to_padded: null buffer");
    int64_t g = (batch * bag_len + NRX_BLOCK - 1) / NRX_BLOCK;
    if (g > 4096) g = 4096;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (value_bits == 64)
        hipLaunchKernelGGL(csr_to_padded_kernel<int64_t>, dim3((unsigned)g), dim3(NRX_BLOCK), 0, st, (const int64_t*)values, offsets,
                           rows, batch, bag_len, (int64_t*)ids_out, mask_out);
    else
        hipLaunchKernelGGL(csr_to_padded_kernel<int32_t>, dim3((unsigned)g), dim3(NRX_BLOCK), 0, st, (const int32_t*)values, offsets,
                           rows, batch, bag_len, (int32_t*)ids_out, mask_out);
    NRX_LAUNCH_CHECK("nrx_csr_to_padded");
    return NRX_OK;
}

extern "C" int nrx_user_rank_metrics(const float* scores, const float* labels, const int64_t* seg_start, int64_t n_users,
                                     int32_t k, double* auc, double* ndcg, double* hr, double* mrr, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(n_users >= 0 && k >= 1, "nrx_user_rank_metrics: bad argument");
    if (n_users == 0) return NRX_OK;
    NRX_REQUIRE(scores && labels && seg_start && auc && ndcg && hr && mrr, "nrx_user_rank_metrics: null buffer");
    int64_t g = (n_users + NRX_BLOCK - 1) / NRX_BLOCK;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(user_rank_metrics_kernel, dim3((unsigned)g), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                       scores, labels, seg_start, n_users, k, auc, ndcg, hr, mrr);
    NRX_LAUNCH_CHECK("nrx_user_rank_metrics");
    return NRX_OK;
}

extern "C" int nrx_mask_lengths(const float* mask, int64_t batch, int32_t bag_len, int64_t* lens, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(batch >= 0 && bag_len >= 1, "nrx_mask_lengths: bad argument");
    if (batch == 0) return NRX_OK;
    NRX_REQUIRE(mask && lens, "nrx_mask_lengths: null buffer");
    int64_t g = (batch + NRX_BLOCK - 1) / NRX_BLOCK;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(mask_lengths_kernel, dim3((unsigned)g), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), mask, batch, bag_len, lens);
    NRX_LAUNCH_CHECK("nrx_mask_lengths");
    return NRX_OK;
}
