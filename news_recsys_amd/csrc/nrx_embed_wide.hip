// Fused gather -> concat WITH the Wide&Deep column split, fast path for uniform single-valued features.
// Reference: WideDeep.get_inp_embedding (src/model/sort/widedeep/model.py:53-69): for a feature in
// `wide_feature_names` column 0 of its embedding row goes to the wide tensor and columns 1..D-1 to the deep concat,
// every other feature goes to the deep concat whole.  The deep row therefore has blocks of D and D-1 floats and
// nothing after the first wide feature is 16-byte aligned any more; the generic kernel falls back to one feature at
// a time and scalar stores there (C5's table set: 210 us against 82 us for the same gather without the split).
// Global memory only needs dword alignment for multi-dword accesses, so this kernel keeps the uniform kernel's
// structure -- U independent aligned 16-byte row loads per lane in flight, then the stores -- and writes each
// lane's 4 floats with dword-aligned 12-byte + 4-byte stores at the (shifted) deep position; the 4-byte store of a
// wide feature's lane 0 is redirected to the wide tensor.  Branch-free inside a group (see nrx_embed.hip on why).
#include "nrx_common.h"
#include "nrx_embed_ring.h"   // nrx_ring::stage_ids / load_row (the ring form below)

struct UniformWideArgs {
    const float* table[NRX_MAX_FEATURES];
    const void* index[NRX_MAX_FEATURES];
    int64_t rows[NRX_MAX_FEATURES];
    int32_t col[NRX_MAX_FEATURES];        // first deep column of the feature (float units)
    int32_t wide_col[NRX_MAX_FEATURES];   // column in the wide tensor, -1 = not a wide feature
    uint8_t feat_id[NRX_MAX_FEATURES];    // the feature's index in the caller's list (out-of-range reports; stage_ids of nrx_embed_ring.h reads it)
    int64_t batch;
    float* out;                           // deep concat [batch, ld]
    int64_t ld;
    float* wide;                          // [batch, wide_ld]
    int64_t wide_ld;
    int32_t* status;
    int32_t n;
    int32_t idx64;
    int32_t stnt;                         // 1: the aligned chunks leave with non-temporal stores (rows on whole 128-byte lines: written once, read by the MLP later)
};
static_assert(sizeof(UniformWideArgs) <= 3584, "kernarg budget");

namespace {

typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int Q, int CNT, bool IDX64, bool NT>
__device__ __forceinline__ void wide_group(const NRX_CONST UniformWideArgs* a, int f0, int64_t b, int q, int& bad_feat, int64_t& bad_id) {
    int64_t id[CNT];
    float4 v[CNT];
#pragma unroll
    for (int u = 0; u < CNT; ++u)
        id[u] = IDX64 ? nrx_gconst<int64_t>(a->index[f0 + u])[b] : (int64_t)nrx_gconst<int32_t>(a->index[f0 + u])[b];
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
        const bool bad = (uint64_t)id[u] >= (uint64_t)a->rows[f0 + u];
        bad_feat = bad ? f0 + u : bad_feat;
        bad_id = bad ? id[u] : bad_id;
        id[u] = bad ? 0 : id[u];
        v[u] = NT ? nrx_ldg4_nt(a->table[f0 + u], id[u] * Q + q) : nrx_ldg4(a->table[f0 + u], id[u] * Q + q);
    }
    float* const orow = a->out + b * a->ld;
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
        const int wc = a->wide_col[f0 + u];                     // wave-uniform
        const int shift = wc >= 0 ? 1 : 0;
        float* p = orow + a->col[f0 + u] + 4 * q - shift;       // where this lane's element 0 would land in the deep row
        if (wc >= 0) {                                           // wave-uniform: 4-byte + 12-byte stores, lane 0's first
            float* p0 = q == 0 ? a->wide + b * a->wide_ld + wc : p;          // float goes to the wide tensor
            *p0 = v[u].x;
            f32x3u t;
            t.x = v[u].y; t.y = v[u].z; t.z = v[u].w;
            *reinterpret_cast<f32x3u*>(p + 1) = t;
        } else {                                                 // whole row to the deep concat: one (dword-aligned) 16-byte store
            f32x4u t;
            t.x = v[u].x; t.y = v[u].y; t.z = v[u].z; t.w = v[u].w;
            *reinterpret_cast<f32x4u*>(p) = t;
        }
    }
}

template <int Q, int U, int R, bool IDX64, bool NT>
struct WideTail {
    static __device__ __forceinline__ void run(const NRX_CONST UniformWideArgs* a, int f0, int rem, int64_t b, int q, int& bad_feat, int64_t& bad_id) {
        if (rem == R) wide_group<Q, R, IDX64, NT>(a, f0, b, q, bad_feat, bad_id);
        else WideTail<Q, U, R + 1, IDX64, NT>::run(a, f0, rem, b, q, bad_feat, bad_id);
    }
};
template <int Q, int U, bool IDX64, bool NT>
struct WideTail<Q, U, U, IDX64, NT> {
    static __device__ __forceinline__ void run(const NRX_CONST UniformWideArgs*, int, int, int64_t, int, int&, int64_t&) {}
};

template <int QLOG2, int U, bool IDX64, bool NT>
__global__ __launch_bounds__(NRX_BLOCK) void embed_fwd_uniform_wide(const UniformWideArgs args_in_kernarg_segment) {
    const NRX_CONST UniformWideArgs* a = nrx_kernarg<UniformWideArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    const int q = threadIdx.x & (Q - 1);
    const int64_t b = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    if (b >= a->batch) return;
    int bad_feat = -1;
    int64_t bad_id = 0;
    const int n = a->n;
    int f0 = 0;
    for (; f0 + U <= n; f0 += U) wide_group<Q, U, IDX64, NT>(a, f0, b, q, bad_feat, bad_id);
    if (f0 < n) WideTail<Q, U, 1, IDX64, NT>::run(a, f0, n - f0, b, q, bad_feat, bad_id);
    if (bad_feat >= 0 && q == 0) nrx_report_oob(a->status, bad_feat, b, bad_id);
}

// Ring form (as embed_fwd_ring, nrx_embed_ring.h): ids staged once per block in LDS (bounds-checked, 32-bit), R row loads in
// flight per lane, `store feature f; load feature f + R`.  Requires n >= R.  C5 with the split on the 10 smallest tables:
// burst form 159 us -> see profiles.
template <int Q>
__device__ __forceinline__ void wide_store(const NRX_CONST UniformWideArgs* a, int f, float4 v, int q, float* orow, float* wrow) {
    const int wc = a->wide_col[f];                              // wave-uniform
    float* p = orow + a->col[f] + 4 * q - (wc >= 0 ? 1 : 0);
    if (wc >= 0) {
        float* p0 = q == 0 ? wrow + wc : p;
        *p0 = v.x;
        f32x3u t;
        t.x = v.y; t.y = v.z; t.z = v.w;
        *reinterpret_cast<f32x3u*>(p + 1) = t;
    } else {
        f32x4u t;
        t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        *reinterpret_cast<f32x4u*>(p) = t;
    }
}

// ALIGNED stores (round 4).  The split makes deep blocks of D - 1 floats, so with wide_store above every 16-byte store after the first
// wide feature starts at a dword-aligned address inside a 16-byte slot: each of a sample's 128-byte pieces straddles two lines (the C5 set:
// 152 us against 131 us for the plain concat of the same tables).  When the deep row is gap-free, starts on a 16-byte boundary and the row
// stride is a multiple of 4 floats (the caller pads out_ld: 1270 -> 1272), the row is instead written as the stream of ALIGNED 16-byte chunks it
// is: lane q of feature f stores the chunk that ends inside its own 4 floats -- its first d floats come from the lane below (one DPP move
// per float; lane 0: from the previous feature's last lane, carried in registers), d = the number of floats pending from earlier features
// (wave-uniform, 0..3) -- and whatever is left over at the end of the row leaves as 1-3 dword stores.  Same values in the same places.
template <int Q>
struct WideAl {
    float4 carry;       // lane 0 of each group: the previous feature's last lane's floats (its last `p` are not stored yet)
    int p;              // pending floats of the row so far (== the next deep column mod 4); wave-uniform
    __device__ __forceinline__ void store(const NRX_CONST UniformWideArgs* a, int f, float4 v, int q, float* orow, float* wrow) {
        const int wc = a->wide_col[f];                          // wave-uniform
        const int w = wc >= 0 ? 1 : 0;
        if (w && q == 0) wrow[wc] = v.x;
        float4 lower, nxt;
        lower.x = nrx_dpp<0x111>(v.x); lower.y = nrx_dpp<0x111>(v.y); lower.z = nrx_dpp<0x111>(v.z); lower.w = nrx_dpp<0x111>(v.w);      // lane q - 1
        nxt.x = nrx_dpp<0x100 + Q - 1>(v.x); nxt.y = nrx_dpp<0x100 + Q - 1>(v.y);                                                     // lane q + Q - 1:
        nxt.z = nrx_dpp<0x100 + Q - 1>(v.z); nxt.w = nrx_dpp<0x100 + Q - 1>(v.w);                                                     //   lane 0 <- the group's last lane
        const int d = (p - w) & 3;
        float4 lo = lower, hi = v;
        if (q == 0) {
            // a wide feature's float 0 is not part of the deep row: the slot it would take is the previous feature's last float
            lo = w ? make_float4(0.f, carry.x, carry.y, carry.z) : carry;
            hi = w ? make_float4(carry.w, v.y, v.z, v.w) : v;
        }
        float4 c;
        switch (d) {                                            // wave-uniform
            case 0: c = hi; break;
            case 1: c = make_float4(lo.w, hi.x, hi.y, hi.z); break;
            case 2: c = make_float4(lo.z, lo.w, hi.x, hi.y); break;
            default: c = make_float4(lo.y, lo.z, lo.w, hi.x); break;
        }
        if (!(q == 0 && w && p == 0)) {                         // (nothing pending in front of a wide feature: its lane 0 completes no chunk)
            nrx_f32x4 t;
            t.x = c.x; t.y = c.y; t.z = c.z; t.w = c.w;
            if (a->stnt) __builtin_nontemporal_store(t, (NRX_GLOBAL nrx_f32x4*)(orow + a->col[f] - w + 4 * q - d));
            else *(NRX_GLOBAL nrx_f32x4*)(orow + a->col[f] - w + 4 * q - d) = t;
        }
        carry = nxt;
        p = d;
    }
    __device__ __forceinline__ void flush(int q, float* orow, int end_col) {      // the row's last 1-3 floats
        if (q == 0) {
            if (p >= 3) orow[end_col - 3] = carry.y;
            if (p >= 2) orow[end_col - 2] = carry.z;
            if (p >= 1) orow[end_col - 1] = carry.w;
        }
    }
};

template <int QLOG2, int R, bool NT, bool AL = false>
__global__ __launch_bounds__(NRX_BLOCK) void embed_fwd_ring_wide(const UniformWideArgs args_in_kernarg_segment) {
    using namespace nrx_ring;
    const NRX_CONST UniformWideArgs* a = nrx_kernarg<UniformWideArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) int32_t s_ids[];   // [n][TB]
    const int tid = threadIdx.x;
    const int n = a->n;
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int nb = (int)((a->batch - b0) < (int64_t)TB ? (a->batch - b0) : (int64_t)TB);
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
    if (b >= a->batch) return;
    const int32_t* s_my = s_ids + sb;
    float* const orow = a->out + b * a->ld;
    float* const wrow = a->wide + b * a->wide_ld;
    WideAl<Q> al;
    al.carry = make_float4(0.f, 0.f, 0.f, 0.f);
    al.p = 0;
    auto put = [&](int f, float4 x) {
        if (AL) al.store(a, f, x, q, orow, wrow);
        else wide_store<Q>(a, f, x, q, orow, wrow);
    };
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
            put(f0 + u, v[u]);
            v[u] = load_row<QLOG2, NT>(a->table[f0 + R + u], idn[u], q);
        }
    }
    {
        int32_t idn[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int f = f0 + R + u;
            idn[u] = s_my[(f < n ? f : n - 1) * TB];
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            put(f0 + u, v[u]);
            if (f0 + R + u < n) v[u] = load_row<QLOG2, NT>(a->table[f0 + R + u], idn[u], q);
        }
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (f0 + R + u < n) put(f0 + R + u, v[u]);
    }
    if (AL) al.flush(q, orow, a->col[n - 1] + (4 * Q) - (a->wide_col[n - 1] >= 0 ? 1 : 0));
}

template <int QLOG2>
void launch_ring_wide(const UniformWideArgs& ua, int64_t batch, bool nt, bool al, hipStream_t st) {
    constexpr int TB = NRX_BLOCK >> QLOG2;
    constexpr int R = 8;
    const dim3 grid((unsigned)((batch + TB - 1) / TB)), block(NRX_BLOCK);
    const size_t smem = (size_t)ua.n * TB * 4;
    if (al) {
        if (nt) hipLaunchKernelGGL((embed_fwd_ring_wide<QLOG2, R, true, true>), grid, block, smem, st, ua);
        else hipLaunchKernelGGL((embed_fwd_ring_wide<QLOG2, R, false, true>), grid, block, smem, st, ua);
        return;
    }
    if (nt) hipLaunchKernelGGL((embed_fwd_ring_wide<QLOG2, R, true>), grid, block, smem, st, ua);
    else hipLaunchKernelGGL((embed_fwd_ring_wide<QLOG2, R, false>), grid, block, smem, st, ua);
}

template <int QLOG2, int U>
void launch_wide(const UniformWideArgs& ua, int64_t batch, bool i64, bool nt, hipStream_t st) {
    constexpr int TB = NRX_BLOCK >> QLOG2;
    const dim3 grid((unsigned)((batch + TB - 1) / TB)), block(NRX_BLOCK);
    if (i64) { if (nt) hipLaunchKernelGGL((embed_fwd_uniform_wide<QLOG2, U, true, true>), grid, block, 0, st, ua);
               else hipLaunchKernelGGL((embed_fwd_uniform_wide<QLOG2, U, true, false>), grid, block, 0, st, ua); }
    else     { if (nt) hipLaunchKernelGGL((embed_fwd_uniform_wide<QLOG2, U, false, true>), grid, block, 0, st, ua);
               else hipLaunchKernelGGL((embed_fwd_uniform_wide<QLOG2, U, false, false>), grid, block, 0, st, ua); }
}

}  // namespace

// Returns false when the feature set is not eligible (caller falls back to the generic kernel).
bool nrx_launch_uniform_wide(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, float* out, int64_t out_ld,
                             float* wide_out, int64_t wide_ld, int32_t* status, hipStream_t st) {
    if (out == nullptr || wide_out == nullptr || n_feats < 1) return false;
    const int D0 = feats[0].dim;
    if (D0 != 16 && D0 != 32 && D0 != 64) return false;
    UniformWideArgs ua;
    int64_t table_bytes = 0;
    bool any_wide = false;
    for (int i = 0; i < n_feats; ++i) {
        const nrx_feature_t& s = feats[i];
        if (s.kind != NRX_SPARSE || s.dim != D0 || s.fm_field != 0 || s.index_bits != feats[0].index_bits || s.table == nullptr ||
            s.index == nullptr || !nrx_aligned16(s.table) || s.rows < 1 || s.rows > 0x7fffffffLL || s.out_col < 0)
            return false;
        ua.table[i] = s.table;
        ua.index[i] = s.index;
        ua.rows[i] = s.rows;
        ua.col[i] = s.out_col;
        ua.wide_col[i] = s.wide_col;
        ua.feat_id[i] = (uint8_t)i;
        any_wide |= s.wide_col >= 0;
        table_bytes += s.rows * (int64_t)D0 * 4;
    }
    if (!any_wide) return false;
    ua.batch = batch;
    ua.out = out;
    ua.ld = out_ld;
    ua.wide = wide_out;
    ua.wide_ld = wide_ld;
    ua.status = status;
    ua.n = n_feats;
    const bool i64 = feats[0].index_bits == 64;
    const bool nt = table_bytes > (256ll << 20);
    ua.idx64 = i64;
    { const char* e = getenv("NRX_FWD_STNT"); ua.stnt = e ? atoi(e) : 0; }
    if (n_feats >= 8 && n_feats * (NRX_BLOCK / (D0 / 4)) * 4 <= 48 * 1024) {      // ring form: >= R features, ids fit a modest LDS tile
        // aligned-chunk stores (WideAl): a gap-free deep row in feature order that starts on a 16-byte boundary, row stride % 4 floats == 0
        bool al = nrx_aligned16(out) && (out_ld & 3) == 0 && (feats[0].out_col & 3) == 0;
        for (int i = 0; i + 1 < n_feats && al; ++i)
            al = feats[i + 1].out_col == feats[i].out_col + D0 - (feats[i].wide_col >= 0 ? 1 : 0);
        if (const char* e = getenv("NRX_WIDE_ALIGNED")) al = al && atoi(e) != 0;      // A/B knob
        switch (D0) {
            case 16: launch_ring_wide<2>(ua, batch, nt, al, st); break;
            case 32: launch_ring_wide<3>(ua, batch, nt, al, st); break;
            default: launch_ring_wide<4>(ua, batch, nt, al, st); break;
        }
        return true;
    }
    const int w13 = (13 - n_feats % 13) % 13, w8 = (8 - n_feats % 8) % 8;
    const bool u13 = D0 <= 32 && n_feats >= 13 && w13 <= w8;
    switch (D0) {
        case 16: if (u13) launch_wide<2, 13>(ua, batch, i64, nt, st); else launch_wide<2, 8>(ua, batch, i64, nt, st); break;
        case 32: if (u13) launch_wide<3, 13>(ua, batch, i64, nt, st); else launch_wide<3, 8>(ua, batch, i64, nt, st); break;
        default: launch_wide<4, 8>(ua, batch, i64, nt, st); break;
    }
    return true;
}
