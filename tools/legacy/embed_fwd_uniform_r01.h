// Round-1 form of the uniform gather kernel (burst of U row loads per lane, then U stores; 64-bit ids of the current and
// the next group held in registers), kept ONLY as the baseline of tools/c2_ring_sweep.hip -- the library no longer builds
// it (replaced by csrc/nrx_embed_ring.h).  Uses UniformArgs / fm_accumulate / group_sum from nrx_embed_ring.h.
#pragma once
#include "nrx_embed_ring.h"

// One straight-line group of CNT features: CNT id loads, then CNT independent row loads, then CNT
// stores -- no control flow, so all CNT random row reads of a lane are in flight together.
// Out-of-range ids are clamped branch-free and reported once per lane after the feature walk.
template <int CNT, bool IDX64>
__device__ __forceinline__ void uniform_load_ids(const NRX_CONST UniformArgs* a, int f0, int64_t b, int64_t (&id)[CNT]) {
#pragma unroll
    for (int u = 0; u < CNT; ++u)
        id[u] = IDX64 ? nrx_gconst<int64_t>(a->index[f0 + u])[b]
                      : (int64_t)nrx_gconst<int32_t>(a->index[f0 + u])[b];
}

// One straight-line group of CNT features: the ids are already in registers; CNT independent row
// loads are issued, then (software pipeline) the NEXT group's ids are requested before this group's
// stores, so their latency hides behind the row loads instead of queueing behind the stores.
// Out-of-range ids are clamped branch-free and reported once per lane after the feature walk.
template <int Q, int CNT, int NEXT, bool IDX64, bool FM, bool STORE, bool NT>
__device__ __forceinline__ void uniform_group(const NRX_CONST UniformArgs* a, int f0, int64_t b, int q,
                                              int64_t (&id)[CNT], int64_t (&id_next)[NEXT == 0 ? 1 : NEXT],
                                              int& bad_feat, int64_t& bad_id, float& fm_first, float4& fm_s, float4& fm_q) {
    float4 v[CNT];
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
        const bool bad = (uint64_t)id[u] >= (uint64_t)a->rows[f0 + u];
        bad_feat = bad ? f0 + u : bad_feat;
        bad_id = bad ? id[u] : bad_id;
        id[u] = bad ? 0 : id[u];
        v[u] = NT ? nrx_ldg4_nt(a->table[f0 + u], id[u] * Q + q) : nrx_ldg4(a->table[f0 + u], id[u] * Q + q);
    }
    if (NEXT > 0) uniform_load_ids<(NEXT == 0 ? 1 : NEXT), IDX64>(a, f0 + CNT, b, id_next);
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
        if (STORE) nrx_stg4(a->out, b * a->ld4 + a->col4[f0 + u] + q, v[u]);
        if (FM) fm_accumulate(v[u], q * 4, 4 * Q, fm_first, fm_s, fm_q);
    }
}

template <int Q, int U, int R, bool IDX64, bool FM, bool STORE, bool NT>
struct UniformTail {
    static __device__ __forceinline__ void run(const NRX_CONST UniformArgs* a, int f0, int rem, int64_t b, int q, int& bad_feat,
                                               int64_t& bad_id, float& fm_first, float4& fm_s, float4& fm_q) {
        if (rem == R) {
            int64_t id[R], none[1];
            uniform_load_ids<R, IDX64>(a, f0, b, id);
            uniform_group<Q, R, 0, IDX64, FM, STORE, NT>(a, f0, b, q, id, none, bad_feat, bad_id, fm_first, fm_s, fm_q);
        } else {
            UniformTail<Q, U, R + 1, IDX64, FM, STORE, NT>::run(a, f0, rem, b, q, bad_feat, bad_id, fm_first, fm_s, fm_q);
        }
    }
};
template <int Q, int U, bool IDX64, bool FM, bool STORE, bool NT>
struct UniformTail<Q, U, U, IDX64, FM, STORE, NT> {
    static __device__ __forceinline__ void run(const NRX_CONST UniformArgs*, int, int, int64_t, int, int&, int64_t&, float&, float4&, float4&) {}
};

template <int QLOG2, int U, bool IDX64, bool FM, bool STORE, bool NT>
__global__ __launch_bounds__(NRX_BLOCK) void embed_fwd_uniform(const UniformArgs args_in_kernarg_segment) {
    const NRX_CONST UniformArgs* a = nrx_kernarg<UniformArgs>();   // == &args_in_kernarg_segment
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    const int tid = threadIdx.x;
    const int q = tid & (Q - 1);
    const int64_t b = (int64_t)blockIdx.x * TB + (tid >> QLOG2);
    if (b >= a->batch) return;   // the Q lanes of a sample leave together: the FM shuffle stays inside the group

    float fm_first = 0.f;
    float4 fm_s = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 fm_q = make_float4(0.f, 0.f, 0.f, 0.f);
    int bad_feat = -1;
    int64_t bad_id = 0;

    int f0 = 0;
    const int n = a->n;
    if (n >= U) {
        int64_t id[U], id_next[U], none[1];
        uniform_load_ids<U, IDX64>(a, 0, b, id);
        for (; f0 + 2 * U <= n; f0 += U) {       // a full group follows: prefetch its ids
            uniform_group<Q, U, U, IDX64, FM, STORE, NT>(a, f0, b, q, id, id_next, bad_feat, bad_id, fm_first, fm_s, fm_q);
#pragma unroll
            for (int u = 0; u < U; ++u) id[u] = id_next[u];
        }
        uniform_group<Q, U, 0, IDX64, FM, STORE, NT>(a, f0, b, q, id, none, bad_feat, bad_id, fm_first, fm_s, fm_q);
        f0 += U;
    }
    const int rem = n - f0;
    if (rem > 0) UniformTail<Q, U, 1, IDX64, FM, STORE, NT>::run(a, f0, rem, b, q, bad_feat, bad_id, fm_first, fm_s, fm_q);

    if (bad_feat >= 0 && q == 0) nrx_report_oob(a->status, bad_feat, b, bad_id);
    if (FM) {
        float part = 0.5f * ((fm_s.x * fm_s.x - fm_q.x) + (fm_s.y * fm_s.y - fm_q.y) +
                             (fm_s.z * fm_s.z - fm_q.z) + (fm_s.w * fm_s.w - fm_q.w)) + fm_first;
        part = group_sum<Q>(part);
        if (q == 0) nrx_gmut<float>(a->fm_out)[b] = part;
    }
}

