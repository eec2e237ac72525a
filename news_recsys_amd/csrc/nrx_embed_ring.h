// Uniform gather -> concat (+FM epilogue), "ring" form: the dominant kernel of the C2 / C5 shapes.
//
// Reference arithmetic: BaseModel.get_embeddings_from_batch (src/model/BaseModel/base_model.py:284-308) for
// single-valued features of one common dim, + FM.get_inp_embedding / FMModel.forward pre-sigmoid
// (src/model/sort/fm/model.py:18-26,48-59).
//
// Same lane mapping as the first uniform kernel (a sample is owned by Q = D/4 adjacent lanes, lane q holds columns
// [4q, 4q+4); a 256-thread block owns TB = 256/Q samples and walks the features), but built so that reads and
// writes never phase apart:
//   * the block's n x TB ids are fetched up front by all 256 threads in full-width coalesced loads, bounds-checked
//     ONCE per id (not once per lane), narrowed to 32 bits and parked in LDS -- the feature walk then gets an id with a
//     ds_read instead of holding 64-bit ids of the current and the next group in registers (the first kernel: 170
//     VGPRs, 2 waves per SIMD, two lock-stepped rounds of blocks);
//   * the walk is a software ring of R row registers per lane: `store feature f; issue the load of feature f + R` --
//     every completed row load is turned into a store and a new load at once, so each wave keeps ~R random row
//     reads in flight from its first feature to its last and HBM sees reads and writes interleaved at row
//     granularity, not a chip-wide read burst followed by a chip-wide write burst;
// Measured on the C2 / C5 / C3 shapes against the first kernel (profiles/r02_c2_ring_sweep.md): -2 % with a recycled
// output buffer, -10 % with distinct output buffers, -5..-9 % with cache-resident tables, at 52-100 VGPRs.
// This header is shared by the library (nrx_embed.hip) and the sweep harness (tools/c2_ring_sweep.hip).
#pragma once
#include "nrx_common.h"

struct UniformArgs {
    const float* table[NRX_MAX_FEATURES];
    const void* index[NRX_MAX_FEATURES];
    int64_t rows[NRX_MAX_FEATURES];
    int32_t col4[NRX_MAX_FEATURES];   // out column / 4
    uint8_t feat_id[NRX_MAX_FEATURES]; // the feature's index in the caller's list (what an out-of-range report names)
    int64_t batch;
    float4* out;                      // may be null (FM-only inference)
    int64_t ld4;                      // out_ld / 4
    float* fm_out;
    float* fm_sums;                   // optional [batch, sums_ld]: per-sample field sums of the FM epilogue (training)
    int64_t sums_ld;
    int32_t* status;
    int32_t n;
    int32_t idx64;                    // ids are int64 (else int32)
    int32_t stnt;                     // 1: the concat leaves with non-temporal stores (measurement knob NRX_FWD_STNT)
    int32_t unal;                     // 1: some first column (or `out` / out_ld) is not a multiple of 4 floats -- a dense value in the
                                      //    middle of the sorted feature order shifts everything after it: col4[] and ld4 are then in
                                      //    FLOATS and the row leaves as a dword-aligned 16-byte store (global memory needs no more)
};
static_assert(sizeof(UniformArgs) <= 3584, "kernarg budget");
typedef float nrx_ring_f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// FM bookkeeping for one field chunk held by this lane (columns k0..k0+3 of the field):
// column 0 is the first-order weight, columns 1.. are the factor vector.
__device__ __forceinline__ void fm_accumulate(float4 v, int k0, int D, float& first, float4& s, float4& sq) {
    if (k0 == 0) {
        first += v.x;
        v.x = 0.f;
    }
    if (k0 + 1 >= D) v.y = 0.f;
    if (k0 + 2 >= D) v.z = 0.f;
    if (k0 + 3 >= D) v.w = 0.f;
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    // explicit fused multiply-adds: the ring, generic and one-block-per-sample kernels must form the SAME bits (left to the
    // compiler's contraction choices two kernels differed in the last place)
    sq.x = __builtin_fmaf(v.x, v.x, sq.x); sq.y = __builtin_fmaf(v.y, v.y, sq.y);
    sq.z = __builtin_fmaf(v.z, v.z, sq.z); sq.w = __builtin_fmaf(v.w, v.w, sq.w);
}

// This lane's share of 0.5 * sum_k [(sum_f v)^2 - sum_f v^2] + (first-order sum): one definition, explicit operation order.
__device__ __forceinline__ float fm_lane_part(const float4& s, const float4& sq, float first) {
    const float a = __builtin_fmaf(s.x, s.x, -sq.x), b = __builtin_fmaf(s.y, s.y, -sq.y);
    const float c = __builtin_fmaf(s.z, s.z, -sq.z), d = __builtin_fmaf(s.w, s.w, -sq.w);
    return __builtin_fmaf(0.5f, ((a + b) + c) + d, first);
}

// sum over the Q lanes of a sample, result in all of them: DPP moves inside a 16-lane row (no LDS round trip);
// wider groups finish with cross-row shuffles
template <int Q>
__device__ __forceinline__ float group_sum(float v) {
    if (Q >= 2) v += nrx_dpp<0xB1>(v);           // lane ^ 1
    if (Q >= 4) v += nrx_dpp<0x4E>(v);           // lane ^ 2
    if (Q >= 8) v += nrx_dpp<0x141>(v);          // other quad of each 8
    if (Q >= 16) v += nrx_dpp<0x140>(v);         // other half of each 16
#pragma unroll
    for (int off = 16; off < Q; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

namespace nrx_ring {

template <int QLOG2, bool NT>
__device__ __forceinline__ float4 load_row(const float* table /*wave-uniform*/, int32_t id, int q) {
    const NRX_GLOBAL nrx_f32x4* p = (const NRX_GLOBAL nrx_f32x4*)table + (((int64_t)id << QLOG2) + q);
    const nrx_f32x4 t = NT ? __builtin_nontemporal_load(p) : *p;
    return make_float4(t.x, t.y, t.z, t.w);
}

// One step of the walk: hand the finished row of feature f to its consumers.
template <int Q, bool FM, bool STORE>
__device__ __forceinline__ void consume(const NRX_CONST UniformArgs* a, int f, float4 v, int q, int64_t row4 /* b*ld4 + q */,
                                        float& fm_first, float4& fm_s, float4& fm_q) {
    if (STORE) {
        nrx_f32x4 t;
        t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        if (a->unal) *(NRX_GLOBAL nrx_ring_f32x4u*)((NRX_GLOBAL float*)(a->out) + row4 + a->col4[f]) = t;      // row4 = b * ld + 4 q, in floats
        else if (a->stnt) __builtin_nontemporal_store(t, (NRX_GLOBAL nrx_f32x4*)(a->out) + row4 + a->col4[f]);
        else ((NRX_GLOBAL nrx_f32x4*)(a->out))[row4 + a->col4[f]] = t;
    }
    if (FM) fm_accumulate(v, q * 4, 4 * Q, fm_first, fm_s, fm_q);
}

// Wavefront `wave` of the block stages the ids of features wave, wave + 4, ... (wave-uniform feature => table sizes
// and id pointers come from scalar loads); lanes < TB each fetch one id; the loads of a pass are in flight together
// (branch-free up to the rare out-of-range report; a pass past the last feature re-reads it and drops the result).
template <int TB, bool IDX64, typename A = UniformArgs>       // A: any argument block with index[] / rows[] / status
__device__ __forceinline__ void stage_ids(const NRX_CONST A* a, int32_t* s_ids, int n, int64_t b0, int nb, int lane, int wave) {
    const int s = lane < nb ? lane : nb - 1;          // TB <= 64 = wavefront size
    constexpr int PASS = 4;
    for (int k0 = 0; k0 * 4 < n; k0 += PASS) {
        int64_t idv[PASS];
#pragma unroll
        for (int u = 0; u < PASS; ++u) {
            int f = (k0 + u) * 4 + wave;
            f = f < n ? f : n - 1;
            const void* p = a->index[f];
            idv[u] = IDX64 ? nrx_gconst<int64_t>(p)[b0 + s] : (int64_t)nrx_gconst<int32_t>(p)[b0 + s];
        }
#pragma unroll
        for (int u = 0; u < PASS; ++u) {
            const int f = (k0 + u) * 4 + wave;
            const int fc = f < n ? f : n - 1;
            int64_t id = idv[u];
            if ((uint64_t)id >= (uint64_t)a->rows[fc]) {
                if (f < n && lane < nb) nrx_report_oob(a->status, a->feat_id[f], b0 + lane, id);
                id = 0;
            }
            if (f < n && lane < TB) s_ids[f * TB + lane] = (int32_t)id;
        }
    }
}

}  // namespace nrx_ring

// Requires n >= R (the host sends smaller feature counts to a smaller R).  Dynamic LDS: n * TB * 4 bytes.
template <int QLOG2, int R, bool FM, bool STORE, bool NT, int MINW = 4>
__global__ __launch_bounds__(NRX_BLOCK, MINW) void embed_fwd_ring(const UniformArgs args_in_kernarg_segment) {
    using namespace nrx_ring;
    const NRX_CONST UniformArgs* a = nrx_kernarg<UniformArgs>();   // == &args_in_kernarg_segment
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) int32_t s_ids[];   // [n][TB]

    const int tid = threadIdx.x;
    const int n = a->n;
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int nb = (int)((a->batch - b0) < (int64_t)TB ? (a->batch - b0) : (int64_t)TB);

    // ---- stage the block's ids in LDS (bounds-checked, 32-bit)
    {
        const int lane = tid & (NRX_WAVE - 1);
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (a->idx64) stage_ids<TB, true>(a, s_ids, n, b0, nb, lane, wave);
        else stage_ids<TB, false>(a, s_ids, n, b0, nb, lane, wave);
    }
    __syncthreads();

    const int q = tid & (Q - 1);
    const int sb = tid >> QLOG2;
    const int64_t b = b0 + sb;
    if (b >= a->batch) return;   // the Q lanes of a sample leave together: the FM shuffle stays inside the group
    const int32_t* s_my = s_ids + sb;
    const int64_t row4 = a->unal ? b * a->ld4 + 4 * q : b * a->ld4 + q;

    float fm_first = 0.f;
    float4 fm_s = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 fm_q = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- the ring: R row loads in flight per lane from here to the last feature
    float4 v[R];
    {
        int32_t idn[R];
#pragma unroll
        for (int u = 0; u < R; ++u) idn[u] = s_my[u * TB];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = load_row<QLOG2, NT>(a->table[u], idn[u], q);
    }
    int f0 = 0;
    for (; f0 + 2 * R <= n; f0 += R) {
        int32_t idn[R];
#pragma unroll
        for (int u = 0; u < R; ++u) idn[u] = s_my[(f0 + R + u) * TB];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            consume<Q, FM, STORE>(a, f0 + u, v[u], q, row4, fm_first, fm_s, fm_q);
            v[u] = load_row<QLOG2, NT>(a->table[f0 + R + u], idn[u], q);
        }
    }
    // f0 + R <= n < f0 + 2R: drain the ring; the n - f0 - R loads still to be issued sit behind wave-uniform branches
    {
        int32_t idn[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int f = f0 + R + u;
            idn[u] = s_my[(f < n ? f : n - 1) * TB];
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            consume<Q, FM, STORE>(a, f0 + u, v[u], q, row4, fm_first, fm_s, fm_q);
            if (f0 + R + u < n) v[u] = load_row<QLOG2, NT>(a->table[f0 + R + u], idn[u], q);
        }
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (f0 + R + u < n) consume<Q, FM, STORE>(a, f0 + R + u, v[u], q, row4, fm_first, fm_s, fm_q);
    }

    if (FM) {
        float part = fm_lane_part(fm_s, fm_q, fm_first);
        if (a->fm_sums != nullptr) {   // S[b, :] = sum over fields (column 0: the first-order sum); the FM backward needs it
            nrx_f32x4 t;
            t.x = q == 0 ? fm_first : fm_s.x; t.y = fm_s.y; t.z = fm_s.z; t.w = fm_s.w;
            *(NRX_GLOBAL nrx_f32x4*)(nrx_gmut<float>(a->fm_sums) + b * a->sums_ld + 4 * q) = t;
        }
        part = group_sum<Q>(part);
        if (q == 0 && a->fm_out != nullptr) nrx_gmut<float>(a->fm_out)[b] = part;
    }
}
