// Standalone pooling / FM / DCN-v1 kernels (forward + backward) on materialised tensors.  gfx950.
//
//   nrx_bag_pool_*  BaseModel.array_feature_pooling        src/model/BaseModel/base_model.py:273-282
//   nrx_fm_*        FM.get_inp_embedding + FMModel.forward  src/model/sort/fm/model.py:18-26,48-59
//   nrx_dcn_v1_*    DCNLayer / DCNNet                       src/model/sort/dcn/dcn_arch.py:14-30,63-70
//
// All three are < 1 flop/byte: HBM-bound streaming kernels.  FM and DCN keep a sample's row in
// registers (FM: Q lanes per sample like the gather kernel; DCN: one wavefront per row, all cross
// layers fused, the per-layer dot product reduced with wave shuffles, w/b staged once per block in LDS).
#include "nrx_common.h"

namespace {

// ------------------------------------------------------------------------------ bag pooling
__global__ __launch_bounds__(NRX_BLOCK) void bag_pool_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ mask,
                                                                 int64_t batch, int L, int D, float* __restrict__ out) {
    const int64_t total = batch * (int64_t)D;
    for (int64_t i = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * NRX_BLOCK) {
        const int64_t b = i / D;
        const int d = (int)(i - b * D);
        const float* e = emb + (b * L) * (int64_t)D + d;
        float acc = 0.f, den = 0.f;
        if (mask != nullptr) {
            const float* m = mask + b * (int64_t)L;
            for (int l = 0; l < L; ++l) {
#pragma clang fp contract(off)
                const float w = m[l];
                den += w;
                acc += e[(int64_t)l * D] * w;
            }
            out[i] = acc / (den + 1e-8f);
        } else {
            for (int l = 0; l < L; ++l) acc += e[(int64_t)l * D];
            out[i] = acc / (float)L;
        }
    }
}

// one block per sample: den once, then the [L, D] slab
// One wavefront per sample (the first version used a 256-thread block per sample with two barriers, scalar stores and a division
// per element: 94.7 us for [65536, 50, 16], 2.35 TB/s; this form: float4 stores of g / den * mask over the sample's L x D floats).
template <bool VEC>
__global__ __launch_bounds__(NRX_BLOCK) void bag_pool_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ mask,
                                                                 int64_t batch, int L, int D, float* __restrict__ g_emb) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (NRX_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NRX_BLOCK / 64);
    const int n = L * D;
    for (int64_t b = wave; b < batch; b += nwaves) {
        float den = (float)L;
        if (mask != nullptr) {
            float part = 0.f;
            for (int l = lane; l < L; l += NRX_WAVE) part += mask[b * (int64_t)L + l];
            den = nrx_wave_sum(part) + 1e-8f;
        }
        const float* gb = g_out + b * (int64_t)D;
        const float* mb = mask ? mask + b * (int64_t)L : nullptr;
        float* ob = g_emb + b * (int64_t)n;
        if (VEC) {
            const unsigned q = (unsigned)D >> 2;            // float4 chunks per bag position
            for (unsigned e4 = lane; e4 < (unsigned)n >> 2; e4 += NRX_WAVE) {
                const unsigned l = e4 / q, d = (e4 - l * q) * 4;
                const float4 g = *reinterpret_cast<const float4*>(gb + d);
                const float m = mb ? mb[l] : 1.0f;
                float4 o = make_float4(g.x / den, g.y / den, g.z / den, g.w / den);
                if (mb) o = make_float4(o.x * m, o.y * m, o.z * m, o.w * m);
                *reinterpret_cast<float4*>(ob + (size_t)e4 * 4) = o;
            }
        } else {
            for (int e = lane; e < n; e += NRX_WAVE) {
                const int l = e / D, d = e - l * D;
                const float g = gb[d] / den;
                ob[e] = mb ? g * mb[l] : g;
            }
        }
    }
}

// ------------------------------------------------------------------------------ FM
__device__ __forceinline__ float4 ld4(const float* p, int k0, int D, bool vec) {
    float4 v;
    if (vec) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        v.x = p[0];
        v.y = (k0 + 1 < D) ? p[1] : 0.f;
        v.z = (k0 + 2 < D) ? p[2] : 0.f;
        v.w = (k0 + 3 < D) ? p[3] : 0.f;
    }
    return v;
}

template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void fm_fwd_kernel(const float* __restrict__ feat, int64_t ld, int F, int D,
                                                           int64_t batch, float* __restrict__ fm_out, bool vec,
                                                           float* __restrict__ sums = nullptr, int64_t sums_ld = 0) {
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    const int q = threadIdx.x & (Q - 1);
    const int64_t b = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    const bool live = b < batch;
    float total = 0.f;
    for (int kc = 0; kc < D; kc += 4 * Q) {     // one pass when D <= 4Q (the normal case)
        const int k0 = kc + q * 4;
        float first = 0.f;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
        if (live && k0 < D) {
            const float* row = feat + b * ld + k0;
            for (int f = 0; f < F; ++f) {
                float4 v = ld4(row + (int64_t)f * D, k0, D, vec);
                if (k0 == 0) { first += v.x; v.x = 0.f; }
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
            }
        }
        total += 0.5f * ((s.x * s.x - sq.x) + (s.y * s.y - sq.y) + (s.z * s.z - sq.z) + (s.w * s.w - sq.w)) + first;
        if (sums != nullptr && live && k0 < D) {      // training form: the field sums the backward folds in (column 0: sum of the first-order weights)
            float* dst = sums + b * sums_ld + k0;
            const float v0 = k0 == 0 ? first : s.x;
            dst[0] = v0;
            if (k0 + 1 < D) dst[1] = s.y;
            if (k0 + 2 < D) dst[2] = s.z;
            if (k0 + 3 < D) dst[3] = s.w;
        }
    }
#pragma unroll
    for (int off = Q / 2; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    if (live && q == 0) fm_out[b] = total;
}

template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void fm_bwd_kernel(const float* __restrict__ feat, int64_t ld, int F, int D, int64_t batch,
                                                           const float* __restrict__ g_fm, const float* g_in, int64_t g_in_ld,
                                                           float* g_feat, int64_t g_ld, bool vec) {
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    constexpr int U = 8;               // field rows in flight per lane
    const int q = threadIdx.x & (Q - 1);
    const int64_t b = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    if (b >= batch) return;
    const float gl = g_fm[b];
    const bool gvec = vec && (g_ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(g_feat) & 15u) == 0) &&
                      (g_in == nullptr || ((g_in_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(g_in) & 15u) == 0));
    for (int kc = 0; kc < D; kc += 4 * Q) {
        const int k0 = kc + q * 4;
        if (k0 >= D) continue;
        const float* row = feat + b * ld + k0;
        const float* irow = g_in ? g_in + b * g_in_ld + k0 : nullptr;
        float* grow = g_feat + b * g_ld + k0;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int f = 0;
        for (; f + U <= F; f += U) {          // pass 1: S = sum over fields (U independent loads in flight)
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld4(row + (int64_t)(f + u) * D, k0, D, vec);
#pragma unroll
            for (int u = 0; u < U; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; f < F; ++f) {
            const float4 v = ld4(row + (int64_t)f * D, k0, D, vec);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        // pass 2: d/dv_f = gl * (S - v_f) (+ upstream); the feature rows are L1/L2-hot.  Full groups are straight-line
        // (2U loads in flight, then U stores); the aligned fast path only -- the tail and odd shapes go field by field
        f = 0;
        if (gvec && k0 + 4 <= D) {
            for (; f + U <= F; f += U) {
                float4 v[U], up[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    v[u] = ld4(row + (int64_t)(f + u) * D, k0, D, vec);
                    up[u] = irow ? *reinterpret_cast<const float4*>(irow + (int64_t)(f + u) * D) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float4 g = make_float4(gl * (s.x - v[u].x), gl * (s.y - v[u].y), gl * (s.z - v[u].z), gl * (s.w - v[u].w));
                    if (k0 == 0) g.x = gl;      // d/dw = 1
                    *reinterpret_cast<float4*>(grow + (int64_t)(f + u) * D) = make_float4(up[u].x + g.x, up[u].y + g.y, up[u].z + g.z, up[u].w + g.w);
                }
            }
        }
        for (; f < F; ++f) {
            const float4 v = ld4(row + (int64_t)f * D, k0, D, vec);
            float g[4] = {gl * (s.x - v.x), gl * (s.y - v.y), gl * (s.z - v.z), gl * (s.w - v.w)};
            if (k0 == 0) g[0] = gl;
            float* gp = grow + (int64_t)f * D;
            const float* ip = irow ? irow + (int64_t)f * D : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < D) gp[j] = (ip ? ip[j] : 0.f) + g[j];
        }
    }
}

// ------------------------------------------------------------------------------ DCN v1
// One wavefront per row; lane holds R chunks of V contiguous floats: element (r, lane, j) is
// column (r*64 + lane)*V + j.  V = 4 when D % 4 == 0 and everything is 16 B aligned, else 1.
template <int R, int V>
struct RowRegs {
    float v[R][V];
};

template <int R, int V>
__device__ __forceinline__ void row_load(RowRegs<R, V>& x, const float* p, int D, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c = (r * 64 + lane) * V;
        if (V == 4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < D) t = *reinterpret_cast<const float4*>(p + c);
            x.v[r][0] = t.x; x.v[r][1 % V] = t.y; x.v[r][2 % V] = t.z; x.v[r][3 % V] = t.w;
        } else {
            x.v[r][0] = (c < D) ? p[c] : 0.f;
        }
    }
}

template <int R, int V>
__device__ __forceinline__ void row_store(const RowRegs<R, V>& x, float* p, int D, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c = (r * 64 + lane) * V;
        if (c < D) {
            if (V == 4) *reinterpret_cast<float4*>(p + c) = make_float4(x.v[r][0], x.v[r][1 % V], x.v[r][2 % V], x.v[r][3 % V]);
            else p[c] = x.v[r][0];
        }
    }
}

// G = lanes per row: 64 (one row per wavefront) or 32 (narrow rows, D <= 32 V: TWO rows per wavefront, lanes 0..31 / 32..63; lane
// here is the lane inside the row's half).  The half-wave sum uses the same DPP steps inside rows of 16 and adds the two row sums
// of its half -- for a row that fits 32 lanes this is bit for bit what the full-wave tree gives ((r0 + r1) + (0 + 0)).
__device__ __forceinline__ float nrx_half_wave_sum(float v, bool upper) {
    v += nrx_dpp<0xB1>(v);
    v += nrx_dpp<0x4E>(v);
    v += nrx_dpp<0x141>(v);
    v += nrx_dpp<0x140>(v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return upper ? r2 + r3 : r0 + r1;
}

template <int R, int V, int G = 64>
__device__ __forceinline__ float row_dot(const RowRegs<R, V>& a, const RowRegs<R, V>& b, bool upper = false) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < V; ++j) s += a.v[r][j] * b.v[r][j];
    return G == 64 ? nrx_wave_sum(s) : nrx_half_wave_sum(s, upper);
}

// x0p: the layer-0 input when the stack starts from a later x_l (DCNLayer.forward(x_l, x_0), dcn_arch.py:14-30); null = x.
template <int R, int V, int G = 64>
__global__ __launch_bounds__(NRX_BLOCK) void dcn_v1_fwd_kernel(const float* __restrict__ x, int64_t x_ld, const float* __restrict__ x0p,
                                                               int64_t x0_ld, int64_t batch, int D, int NL,
                                                               const float* __restrict__ w, const float* __restrict__ bvec,
                                                               float* __restrict__ out, int64_t out_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_w = reinterpret_cast<float*>(smem);          // [NL][Dp]
    const int Dp = (D + 3) & ~3;
    float* s_b = s_w + NL * Dp;
    for (int i = threadIdx.x; i < NL * Dp; i += NRX_BLOCK) {
        const int l = i / Dp, c = i - l * Dp;
        s_w[i] = (c < D) ? w[l * (int64_t)D + c] : 0.f;
        s_b[i] = (c < D) ? bvec[l * (int64_t)D + c] : 0.f;
    }
    __syncthreads();
    constexpr int RPW = 64 / G;                            // rows per wavefront
    const int lane = threadIdx.x & (G - 1);
    const bool upper = G == 32 && (threadIdx.x & 32) != 0;
    const int64_t wave = (int64_t)blockIdx.x * (NRX_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NRX_BLOCK / 64);
    for (int64_t r0 = wave * RPW; r0 < batch; r0 += nwaves * RPW) {
        const int64_t rr = r0 + (upper ? 1 : 0);
        const bool live = rr < batch;                      // an odd batch leaves the last wavefront's upper half without a row
        const int64_t row = live ? rr : batch - 1;
        RowRegs<R, V> x0, xl;
        row_load<R, V>(xl, x + row * x_ld, D, lane);
        if (x0p != nullptr) row_load<R, V>(x0, x0p + row * x0_ld, D, lane);
        else x0 = xl;
        for (int l = 0; l < NL; ++l) {
            RowRegs<R, V> wl, bl;
            row_load<R, V>(wl, s_w + l * Dp, Dp, lane);
            row_load<R, V>(bl, s_b + l * Dp, Dp, lane);
            const float s = row_dot<R, V, G>(xl, wl, upper);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < V; ++j) xl.v[r][j] = x0.v[r][j] * s + bl.v[r][j] + xl.v[r][j];
        }
        if (live) row_store<R, V>(xl, out + row * out_ld, D, lane);
    }
}

// Backward.  For layer l (top down): g = dL/dx_{l+1};  gs = g . x0;
//   gb_l += g;  gw_l += gs * x_l;  gx0 += g * s_l;  g <- g + gs * w_l.      Finally gx = g + gx0.
// x_l and s_l are recomputed from x0 (l forward steps) instead of being stored: the kernel is
// HBM-bound and the extra FMAs are free; gw/gb accumulate in LDS per block, then one global atomic
// per element per block.
// NLR > 0: gw/gb of the wave's rows accumulate in registers (NLR = n_layers, compile time) and are
// flushed to LDS once per wave; NLR == 0: generic path, per-row LDS atomics (any n_layers).
// 1024-thread blocks, at most one per CU: the per-block flush of gw / gb is NL*D global atomics on the SAME addresses
// from every block, and device-scope atomics on one address serialise -- with 2048 blocks of 256 threads that
// tail was ~60 of the kernel's 99 us; 256 blocks cut the atomics eightfold while 16 waves per block keep 4 per SIMD.
// (3+ layers need more than the 128 VGPRs a 1024-thread block allows: 512-thread blocks, two per CU.)
// SEP: the stack starts from x (= x_l) with a separate layer-0 input x0p; its gradient goes to g_x0 (x's to g_x).
template <int R, int V, int NLR, int DCN_BWD_BLOCK, bool SEP = false, int G = 64, bool ATOMIC = false>
__global__ __launch_bounds__(DCN_BWD_BLOCK) void dcn_v1_bwd_kernel(const float* __restrict__ x, int64_t x_ld, int64_t batch, int D, int NL,
                                                               const float* __restrict__ w, const float* __restrict__ bvec,
                                                               const float* __restrict__ g_out, int64_t g_out_ld,
                                                               float* __restrict__ g_x, int64_t g_x_ld,
                                                               float* __restrict__ g_w, float* __restrict__ g_b,
                                                               const float* __restrict__ x0p, int64_t x0_ld,
                                                               float* __restrict__ g_x0, int64_t g_x0_ld, int ordered) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Dp = (D + 3) & ~3;
    float* s_w = reinterpret_cast<float*>(smem);   // [NL][Dp]
    float* s_b = s_w + NL * Dp;
    float* s_gw = s_b + NL * Dp;
    float* s_gb = s_gw + NL * Dp;
    for (int i = threadIdx.x; i < NL * Dp; i += DCN_BWD_BLOCK) {
        const int l = i / Dp, c = i - l * Dp;
        s_w[i] = (c < D) ? w[l * (int64_t)D + c] : 0.f;
        s_b[i] = (c < D) ? bvec[l * (int64_t)D + c] : 0.f;
        s_gw[i] = 0.f;
        s_gb[i] = 0.f;
    }
    float* const s_gslab = s_gb + NL * Dp;                 // NLR == 0: [slabs][NL][2][Dp], one slab per wavefront (half)
    if (NLR == 0 && !ATOMIC)
        for (int i = threadIdx.x; i < (DCN_BWD_BLOCK / 64) * (64 / G) * NL * 2 * Dp; i += DCN_BWD_BLOCK) s_gslab[i] = 0.f;
    __syncthreads();
    static_assert(G == 64 || (G == 32 && R == 1), "two rows per wavefront only for rows that fit 32 lanes");
    constexpr int RPW = 64 / G;                            // rows per wavefront (G = 32: lanes 0..31 / 32..63 hold one row each)
    const int lane = threadIdx.x & (G - 1);
    const bool upper = G == 32 && (threadIdx.x & 32) != 0;
    const int64_t wave = ((int64_t)blockIdx.x * (DCN_BWD_BLOCK / 64) + (threadIdx.x >> 6)) * RPW;      // first row of this wavefront
    const int64_t nwaves = (int64_t)gridDim.x * (DCN_BWD_BLOCK / 64) * RPW;                            // row stride of the walk
    const int gslab = (threadIdx.x >> 6) * RPW + (upper ? 1 : 0);
    constexpr int NA = NLR > 0 ? NLR : 1;
    float acc_w[NA][R][V], acc_b[NA][R][V];
#pragma unroll
    for (int l = 0; l < NA; ++l)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < V; ++j) { acc_w[l][r][j] = 0.f; acc_b[l][r][j] = 0.f; }

    // software pipeline: the next row's x0 / g loads are issued before this row's five reductions and the
    // store, so the memory system is never idle behind the arithmetic of a wave
    // (G = 32: an odd batch leaves the last wavefront's upper half without a row -- it re-reads the last row with a zero
    // upstream gradient, so nothing it computes contributes, and skips its stores)
    auto row_of = [&](int64_t r0) { const int64_t rr = r0 + (upper ? 1 : 0); return rr < batch ? rr : batch - 1; };
    auto load_g = [&](RowRegs<R, V>& gr, int64_t r0) {
        row_load<R, V>(gr, g_out + row_of(r0) * g_out_ld, D, lane);
        if (G == 32 && r0 + (upper ? 1 : 0) >= batch) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < V; ++j) gr.v[r][j] = 0.f;
        }
    };
    RowRegs<R, V> x0n, gn;
    if (wave < batch) {
        row_load<R, V>(x0n, (SEP ? x0p + row_of(wave) * x0_ld : x + row_of(wave) * x_ld), D, lane);
        load_g(gn, wave);
    }
    for (int64_t rw = wave; rw < batch; rw += nwaves) {
        const int64_t row = row_of(rw);
        const bool live = rw + (upper ? 1 : 0) < batch;
        RowRegs<R, V> x0 = x0n, g = gn, gx0, xs;
        if (SEP) row_load<R, V>(xs, x + row * x_ld, D, lane);      // the stack's first input x_l (not prefetched)
        if (rw + nwaves < batch) {
            row_load<R, V>(x0n, (SEP ? x0p + row_of(rw + nwaves) * x0_ld : x + row_of(rw + nwaves) * x_ld), D, lane);
            load_g(gn, rw + nwaves);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < V; ++j) gx0.v[r][j] = 0.f;
        if constexpr (NLR > 0) {
            // forward once, keeping every layer's input x_l and scalar s_l in registers (the same arithmetic, in the same order, as
            // the forward kernel), then top-down: NLR + NLR row reductions per row -- recomputing x_l for every layer took
            // NLR (NLR + 1) / 2 + NLR of them (3 layers: 6 instead of 9)
            RowRegs<R, V> xls[NLR], wl, bl;
            float ss[NLR];
            xls[0] = SEP ? xs : x0;
#pragma unroll
            for (int t = 0; t < NLR; ++t) {
                row_load<R, V>(wl, s_w + t * Dp, Dp, lane);
                ss[t] = row_dot<R, V, G>(xls[t], wl, upper);
                if (t + 1 < NLR) {
                    row_load<R, V>(bl, s_b + t * Dp, Dp, lane);
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int j = 0; j < V; ++j) xls[t + 1 < NLR ? t + 1 : t].v[r][j] = x0.v[r][j] * ss[t] + bl.v[r][j] + xls[t].v[r][j];
                }
            }
#pragma unroll
            for (int l = NLR - 1; l >= 0; --l) {
                row_load<R, V>(wl, s_w + l * Dp, Dp, lane);
                const float gs = row_dot<R, V, G>(g, x0, upper);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        acc_b[l][r][j] += g.v[r][j];
                        acc_w[l][r][j] += gs * xls[l].v[r][j];
                        gx0.v[r][j] += g.v[r][j] * ss[l];
                        g.v[r][j] += gs * wl.v[r][j];
                    }
            }
        } else {
#pragma unroll
        for (int li = 0; li < (NLR > 0 ? NLR : 1); ++li) {
            // generic path walks the runtime layer count with the same body
            for (int l = (NLR > 0 ? NLR - 1 - li : NL - 1); l >= (NLR > 0 ? NLR - 1 - li : 0); --l) {
                RowRegs<R, V> xl = SEP ? xs : x0, wl, bl;
                float s = 0.f;
                for (int t = 0; t <= l; ++t) {          // recompute x_l, s_l from the stack's input
                    row_load<R, V>(wl, s_w + t * Dp, Dp, lane);
                    s = row_dot<R, V, G>(xl, wl, upper);
                    if (t < l) {
                        row_load<R, V>(bl, s_b + t * Dp, Dp, lane);
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int j = 0; j < V; ++j) xl.v[r][j] = x0.v[r][j] * s + bl.v[r][j] + xl.v[r][j];
                    }
                }
                const float gs = row_dot<R, V, G>(g, x0, upper);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        if (NLR > 0) {
                            acc_b[NLR > 0 ? NLR - 1 - li : 0][r][j] += g.v[r][j];
                            acc_w[NLR > 0 ? NLR - 1 - li : 0][r][j] += gs * xl.v[r][j];
                        } else {
                            // generic depth / width: the wavefront (half) owns an LDS slab [NL][2][Dp] and adds into it with plain
                            // loads and stores -- the same lane always touches the same words, nobody else does (this was a pair of
                            // ds_add_f32 per element per row onto addresses shared by the whole block: D=320 L=5 1 090 us)
                            // (ATOMIC: the last resort when NL x D is too large for the slabs)
                            const int c = (r * G + lane) * V + j;
                            if (c < D) {
                                if (ATOMIC) {
                                    atomicAdd(&s_gb[l * Dp + c], g.v[r][j]);
                                    atomicAdd(&s_gw[l * Dp + c], gs * xl.v[r][j]);
                                } else {
                                    float* my = s_gslab + ((size_t)gslab * NL + l) * 2 * Dp;
                                    my[c] += gs * xl.v[r][j];
                                    my[Dp + c] += g.v[r][j];
                                }
                            }
                        }
                        gx0.v[r][j] += g.v[r][j] * s;
                        g.v[r][j] += gs * wl.v[r][j];
                    }
            }
        }
        }
        if (SEP) {
            if (live) row_store<R, V>(gx0, g_x0 + row * g_x0_ld, D, lane);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < V; ++j) g.v[r][j] += gx0.v[r][j];
        }
        if (live) row_store<R, V>(g, g_x + row * g_x_ld, D, lane);
    }
    if (NLR > 0) {
        // Block-level sum of the per-wavefront accumulators, layer by layer: every wavefront (every half, with two rows per
        // wavefront) writes its partial row to its own LDS slab, then the block adds the slabs in a fixed order.  (The first
        // version did this with ds_add_f32 from all 16 wavefronts onto the same addresses: 26 us per launch, whatever the batch --
        // a timestamped run at B = 16 showed 1 us of staging, 1.3 us of rows and 26 us of LDS atomics.)
        constexpr int SLABS = (DCN_BWD_BLOCK / 64) * RPW;
        float* s_slab = s_gb + NL * Dp;                      // [SLABS][2][Dp]
        const int slab = (threadIdx.x >> 6) * RPW + (upper ? 1 : 0);
#pragma unroll
        for (int l = 0; l < NA; ++l) {
            __syncthreads();                                 // the previous layer's slabs have been consumed
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const int c = (r * G + lane) * V + j;
                    if (c < Dp) {
                        s_slab[(slab * 2 + 0) * Dp + c] = acc_w[l][r][j];
                        s_slab[(slab * 2 + 1) * Dp + c] = acc_b[l][r][j];
                    }
                }
            __syncthreads();
            for (int i = threadIdx.x; i < 2 * Dp; i += DCN_BWD_BLOCK) {
                const int which = i >= Dp, c = i - which * Dp;
                float sum = 0.f;
#pragma unroll 8
                for (int sidx = 0; sidx < SLABS; ++sidx) sum += s_slab[(sidx * 2 + which) * Dp + c];
                (which ? s_gb : s_gw)[l * Dp + c] = sum;
            }
        }
    } else if (!ATOMIC) {
        __syncthreads();
        constexpr int GSLABS = (DCN_BWD_BLOCK / 64) * RPW;
        for (int i = threadIdx.x; i < NL * 2 * Dp; i += DCN_BWD_BLOCK) {
            const int l = i / (2 * Dp), rem = i - l * 2 * Dp, which = rem >= Dp, c = rem - which * Dp;
            float sum = 0.f;
            for (int sidx = 0; sidx < GSLABS; ++sidx) sum += s_gslab[((size_t)sidx * NL + l) * 2 * Dp + rem];
            (which ? s_gb : s_gw)[l * Dp + c] = sum;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NL * Dp; i += DCN_BWD_BLOCK) {
        const int l = i / Dp, c = i - l * Dp;
#ifndef NRX_PROBE_NOFLUSH
        if (c < D) {
            if (ordered) {                  // this block's sums to its own slot [block][NL][D]: dcn_v1_reduce_kernel adds the blocks in block order
                g_w[((int64_t)blockIdx.x * NL + l) * D + c] = s_gw[i];
                g_b[((int64_t)blockIdx.x * NL + l) * D + c] = s_gb[i];
            } else {
                unsafeAtomicAdd(&g_w[l * (int64_t)D + c], s_gw[i]);
                unsafeAtomicAdd(&g_b[l * (int64_t)D + c], s_gb[i]);
            }
        }
#endif
    }
}

// ------------------------------------------------------------------------------ fused gather + DCN v1
// One wavefront per sample (S samples in flight per wave); the concat row is laid out over the lanes
// exactly like RowRegs (chunk c = r*64 + lane holds columns 4c..4c+3), so the cross arithmetic and
// its reduction order are those of dcn_v1_fwd_kernel: bit-identical results.  Which feature / row
// offset a (lane, r) chunk belongs to is the same for every sample and is resolved once per lane.
struct EmbedDcnArgs {
    const float* table[NRX_MAX_FEATURES];
    const void* index[NRX_MAX_FEATURES];
    int64_t rows[NRX_MAX_FEATURES];
    int32_t col[NRX_MAX_FEATURES + 1];   // out_col of each feature (sorted ascending), col[n] = width
    int64_t batch;
    float* out;
    int64_t out_ld;
    const float* w;
    const float* b;
    int32_t* status;
    int32_t n;
    int32_t width;
    int32_t n_layers;
    int32_t idx64;
};
static_assert(sizeof(EmbedDcnArgs) <= 3840, "kernarg budget");

template <int R, int S>
__global__ __launch_bounds__(NRX_BLOCK) void embed_dcn_v1_kernel(const EmbedDcnArgs args_in_kernarg) {
    const NRX_CONST EmbedDcnArgs* a = nrx_kernarg<EmbedDcnArgs>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = a->width, NL = a->n_layers;
    float* s_w = reinterpret_cast<float*>(smem);          // [NL][W]
    float* s_b = s_w + NL * W;
    for (int i = threadIdx.x; i < NL * W; i += NRX_BLOCK) {
        s_w[i] = a->w[i];
        s_b[i] = a->b[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // per-lane chunk -> (feature, float4 offset inside the row), resolved once
    const float* tab[R];
    const void* idxp[R];
    int64_t nrows[R];
    int koff[R], feat[R];
    bool on[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int c4 = (r * 64 + lane) * 4;
        on[r] = c4 < W;
        int lo = 0, hi = a->n;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a->col[mid] <= c4) lo = mid; else hi = mid;
        }
        on[r] = on[r] && c4 < a->col[lo + 1];          // (holes between features stay zero)
        feat[r] = lo;
        tab[r] = a->table[lo];
        idxp[r] = a->index[lo];
        nrows[r] = a->rows[lo];
        koff[r] = c4 - a->col[lo];
    }
    const int64_t wave = (int64_t)blockIdx.x * (NRX_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NRX_BLOCK / 64);
    for (int64_t b0 = wave * S; b0 < a->batch; b0 += nwaves * S) {
        RowRegs<R, 4> x0[S];
        int64_t id[S][R];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                id[s][r] = 0;
                if (on[r] && b0 + s < a->batch)
                    id[s][r] = a->idx64 ? nrx_gconst<int64_t>(idxp[r])[b0 + s] : (int64_t)nrx_gconst<int32_t>(idxp[r])[b0 + s];
            }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on[r] && b0 + s < a->batch) {
                    int64_t i = id[s][r];
                    if ((uint64_t)i >= (uint64_t)nrows[r]) {
                        if (koff[r] == 0) nrx_report_oob(a->status, feat[r], b0 + s, i);
                        i = 0;
                    }
                    v = nrx_ldg4(tab[r] + i * (int64_t)(a->col[feat[r] + 1] - a->col[feat[r]]) + koff[r], 0);
                }
                x0[s].v[r][0] = v.x; x0[s].v[r][1] = v.y; x0[s].v[r][2] = v.z; x0[s].v[r][3] = v.w;
            }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (b0 + s >= a->batch) break;
            RowRegs<R, 4> xl = x0[s];
            for (int l = 0; l < NL; ++l) {
                RowRegs<R, 4> wl, bl;
                row_load<R, 4>(wl, s_w + l * W, W, lane);
                row_load<R, 4>(bl, s_b + l * W, W, lane);
                const float dot = row_dot<R, 4>(xl, wl);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) xl.v[r][j] = x0[s].v[r][j] * dot + bl.v[r][j] + xl.v[r][j];
            }
            float* o = a->out + (b0 + s) * a->out_ld;
            row_store<R, 4>(x0[s], o, W, lane);
            row_store<R, 4>(xl, o + W, W, lane);
        }
    }
}

// Grouped variant for uniform 128 / 256-byte rows (D = 32, 64) and N <= 8 features: the Q = D/4 lanes that own a
// row chunk walk ALL N features of one sample, so the whole concat row of a sample sits in the registers of one
// 8- or 16-lane group (4 or 8 samples per wave instruction, exactly the access pattern of embed_fwd_uniform, the
// fastest gather here).  N row loads in flight, then per cross layer a 4N-term partial dot per lane, a DPP
// reduction inside the group (no LDS), the update, and 2N stores (x and cross).  w / b come from LDS.
template <int Q>
__device__ __forceinline__ float group_sum_dpp(float v) {
    v += nrx_dpp<0xB1>(v);                       // lane ^ 1
    v += nrx_dpp<0x4E>(v);                       // lane ^ 2
    if (Q >= 8) v += nrx_dpp<0x141>(v);          // other quad of each 8
    if (Q >= 16) v += nrx_dpp<0x140>(v);         // other half of each 16
    return v;
}

template <int QLOG2, int N, bool IDX64, bool NT>
__global__ __launch_bounds__(NRX_BLOCK) void embed_dcn_v1_group_kernel(const EmbedDcnArgs args_in_kernarg) {
    const NRX_CONST EmbedDcnArgs* a = nrx_kernarg<EmbedDcnArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = a->width, NL = a->n_layers;
    float4* s_w = reinterpret_cast<float4*>(smem);          // [NL][W/4]
    float4* s_b = s_w + NL * (W / 4);
    for (int i = threadIdx.x; i < NL * (W / 4); i += NRX_BLOCK) {
        s_w[i] = reinterpret_cast<const float4*>(a->w)[i];
        s_b[i] = reinterpret_cast<const float4*>(a->b)[i];
    }
    __syncthreads();
    const int q = threadIdx.x & (Q - 1);
    const int64_t b = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    if (b >= a->batch) return;                      // the Q lanes of a sample leave together
    int64_t id[N];
#pragma unroll
    for (int u = 0; u < N; ++u)
        id[u] = IDX64 ? nrx_gconst<int64_t>(a->index[u])[b] : (int64_t)nrx_gconst<int32_t>(a->index[u])[b];
    float4 x0[N];
    int bad_feat = -1;
    int64_t bad_id = 0;
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const bool bad = (uint64_t)id[u] >= (uint64_t)a->rows[u];
        bad_feat = bad ? u : bad_feat;
        bad_id = bad ? id[u] : bad_id;
        x0[u] = NT ? nrx_ldg4_nt(a->table[u], (bad ? 0 : id[u]) * Q + q) : nrx_ldg4(a->table[u], (bad ? 0 : id[u]) * Q + q);
    }
    if (bad_feat >= 0 && q == 0) nrx_report_oob(a->status, bad_feat, b, bad_id);
    float4 xl[N];
#pragma unroll
    for (int u = 0; u < N; ++u) xl[u] = x0[u];
    for (int l = 0; l < NL; ++l) {
        const float4* wl = s_w + l * (W / 4) + q;
        const float4* bl = s_b + l * (W / 4) + q;
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const float4 wv = wl[u * Q];
            part += xl[u].x * wv.x + xl[u].y * wv.y + xl[u].z * wv.z + xl[u].w * wv.w;
        }
        const float dot = group_sum_dpp<Q>(part);
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const float4 bv = bl[u * Q];
            xl[u].x = x0[u].x * dot + bv.x + xl[u].x;
            xl[u].y = x0[u].y * dot + bv.y + xl[u].y;
            xl[u].z = x0[u].z * dot + bv.z + xl[u].z;
            xl[u].w = x0[u].w * dot + bv.w + xl[u].w;
        }
    }
    float4* o = reinterpret_cast<float4*>(a->out + b * a->out_ld);
#pragma unroll
    for (int u = 0; u < N; ++u) o[u * Q + q] = x0[u];
#pragma unroll
    for (int u = 0; u < N; ++u) o[W / 4 + u * Q + q] = xl[u];
}

// Round 4 form of the grouped kernel.  What the one-tile-per-block form above makes every block do before its first useful request: stage
// w / b in LDS (a global round trip + a barrier), THEN load its ids (a second round trip), THEN its rows (a third) -- for 16 samples, 4 096
// times per launch, with the 2 N stores of a sample issued as one burst at the very end.  Here a block lives for the whole launch
// (grid = resident blocks, tiles strided): the ids of the first tile are requested before w / b are staged, and inside the loop the ids of
// tile i + 1 are requested before tile i's rows are waited for, so a wavefront spends ONE exposed round trip per tile (its rows) instead of
// three, and the output -- written once, never re-read by this launch -- leaves with non-temporal stores (STNT).  Same arithmetic in the
// same order as the kernel above: bit-identical results (tests/test_hip_parity.py::test_fused_gather_cross_*).
template <int QLOG2, int N, bool IDX64, bool NT, bool STNT, bool EARLY = false>
__global__ __launch_bounds__(NRX_BLOCK) void embed_dcn_v1_group_persist_kernel(const EmbedDcnArgs args_in_kernarg) {
    const NRX_CONST EmbedDcnArgs* a = nrx_kernarg<EmbedDcnArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = a->width, NL = a->n_layers;
    float4* s_w = reinterpret_cast<float4*>(smem);          // [NL][W/4]
    float4* s_b = s_w + NL * (W / 4);
    const int q = threadIdx.x & (Q - 1);
    const int64_t batch = a->batch, stride = (int64_t)gridDim.x * TB;
    int64_t b = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    auto load_ids = [&](int64_t bb, int64_t (&id)[N]) {
        const int64_t bc = bb < batch ? bb : batch - 1;       // lanes past the batch re-read the last sample's ids (never stored)
#pragma unroll
        for (int u = 0; u < N; ++u)
            id[u] = IDX64 ? nrx_gconst<int64_t>(a->index[u])[bc] : (int64_t)nrx_gconst<int32_t>(a->index[u])[bc];
    };
    int64_t id[N];
    load_ids(b, id);                                          // in flight while w / b are staged
    for (int i = threadIdx.x; i < NL * (W / 4); i += NRX_BLOCK) {
        s_w[i] = reinterpret_cast<const float4*>(a->w)[i];
        s_b[i] = reinterpret_cast<const float4*>(a->b)[i];
    }
    __syncthreads();
    for (; b - (threadIdx.x >> QLOG2) < batch; b += stride) {     // block-uniform trip count (the tile's first sample decides)
        const bool live = b < batch;
        float4 x0[N];
        int bad_feat = -1;
        int64_t bad_id = 0;
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const bool bad = (uint64_t)id[u] >= (uint64_t)a->rows[u];
            bad_feat = bad ? u : bad_feat;
            bad_id = bad ? id[u] : bad_id;
            x0[u] = NT ? nrx_ldg4_nt(a->table[u], (bad ? 0 : id[u]) * Q + q) : nrx_ldg4(a->table[u], (bad ? 0 : id[u]) * Q + q);
        }
        if (bad_feat >= 0 && q == 0 && live) nrx_report_oob(a->status, bad_feat, b, bad_id);
        if (b - (threadIdx.x >> QLOG2) + stride < batch) load_ids(b + stride, id);       // next tile's ids: requested under this tile's rows
        if (EARLY && live) {                                  // x leaves as soon as it has arrived, under the cross arithmetic
            NRX_GLOBAL nrx_f32x4* o = (NRX_GLOBAL nrx_f32x4*)(a->out + b * a->out_ld);
#pragma unroll
            for (int u = 0; u < N; ++u) {
                nrx_f32x4 t;
                t.x = x0[u].x; t.y = x0[u].y; t.z = x0[u].z; t.w = x0[u].w;
                if (STNT) __builtin_nontemporal_store(t, o + u * Q + q);
                else o[u * Q + q] = t;
            }
        }
        float4 xl[N];
#pragma unroll
        for (int u = 0; u < N; ++u) xl[u] = x0[u];
        for (int l = 0; l < NL; ++l) {
            const float4* wl = s_w + l * (W / 4) + q;
            const float4* bl = s_b + l * (W / 4) + q;
            float part = 0.f;
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const float4 wv = wl[u * Q];
                part += xl[u].x * wv.x + xl[u].y * wv.y + xl[u].z * wv.z + xl[u].w * wv.w;
            }
            const float dot = group_sum_dpp<Q>(part);
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const float4 bv = bl[u * Q];
                xl[u].x = x0[u].x * dot + bv.x + xl[u].x;
                xl[u].y = x0[u].y * dot + bv.y + xl[u].y;
                xl[u].z = x0[u].z * dot + bv.z + xl[u].z;
                xl[u].w = x0[u].w * dot + bv.w + xl[u].w;
            }
        }
        if (live) {
            NRX_GLOBAL nrx_f32x4* o = (NRX_GLOBAL nrx_f32x4*)(a->out + b * a->out_ld);
#pragma unroll
            for (int u = EARLY ? N : 0; u < 2 * N; ++u) {
                const float4 v = u < N ? x0[u] : xl[u - N];
                nrx_f32x4 t;
                t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                NRX_GLOBAL nrx_f32x4* dst = o + (u < N ? 0 : W / 4) + (u < N ? u : u - N) * Q + q;
                if (STNT) __builtin_nontemporal_store(t, dst);
                else *dst = t;
            }
        }
    }
}

int ceil_log2i(int x) {
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

unsigned stream_grid(int64_t items, int per_block) {
    int64_t g = (items + per_block - 1) / per_block;
    const int64_t cap = 256 * 8;   // 8 blocks per CU, grid-stride the rest
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

#define NRX_QSWITCH(qlog2, ...)               \
    switch (qlog2) {                           \
        case 0: { constexpr int QL = 0; __VA_ARGS__; } break; \
        case 1: { constexpr int QL = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int QL = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int QL = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int QL = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int QL = 5; __VA_ARGS__; } break; \
        default: { constexpr int QL = 6; __VA_ARGS__; } break; \
    }

// R (chunks per lane) for the DCN row kernels
#define NRX_RSWITCH(R_, V_, ...)                                   \
    if (V_ == 4) {                                                  \
        constexpr int VV = 4;                                       \
        if (R_ <= 1) { constexpr int RR = 1; __VA_ARGS__; }                \
        else if (R_ <= 2) { constexpr int RR = 2; __VA_ARGS__; }           \
        else if (R_ <= 4) { constexpr int RR = 4; __VA_ARGS__; }           \
        else { constexpr int RR = 8; __VA_ARGS__; }                        \
    } else {                                                        \
        constexpr int VV = 1;                                       \
        if (R_ <= 1) { constexpr int RR = 1; __VA_ARGS__; }                \
        else if (R_ <= 2) { constexpr int RR = 2; __VA_ARGS__; }           \
        else if (R_ <= 4) { constexpr int RR = 4; __VA_ARGS__; }           \
        else if (R_ <= 8) { constexpr int RR = 8; __VA_ARGS__; }           \
        else if (R_ <= 16) { constexpr int RR = 16; __VA_ARGS__; }         \
        else { constexpr int RR = 32; __VA_ARGS__; }                       \
    }

}  // namespace

extern "C" int nrx_bag_pool_fwd(const float* emb, const float* mask, int64_t batch, int32_t bag_len,
                                int32_t dim, float* out, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(emb && out && batch >= 0 && bag_len >= 1 && dim >= 1, "nrx_bag_pool_fwd: bad argument");
    if (batch == 0) return NRX_OK;
    hipLaunchKernelGGL(bag_pool_fwd_kernel, dim3(stream_grid(batch * dim, NRX_BLOCK)), dim3(NRX_BLOCK), 0,
                       reinterpret_cast<hipStream_t>(stream), emb, mask, batch, bag_len, dim, out);
    NRX_LAUNCH_CHECK("nrx_bag_pool_fwd");
    return NRX_OK;
}

extern "C" int nrx_bag_pool_bwd(const float* g_out, const float* mask, int64_t batch, int32_t bag_len,
                                int32_t dim, float* g_emb, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(g_out && g_emb && batch >= 0 && bag_len >= 1 && dim >= 1, "nrx_bag_pool_bwd: bad argument");
    if (batch == 0) return NRX_OK;
    const bool vec = (dim & 3) == 0 && nrx_aligned16(g_out) && nrx_aligned16(g_emb);
    if (vec) hipLaunchKernelGGL(bag_pool_bwd_kernel<true>, dim3(stream_grid(batch, NRX_BLOCK / 64)), dim3(NRX_BLOCK), 0,
                                reinterpret_cast<hipStream_t>(stream), g_out, mask, batch, bag_len, dim, g_emb);
    else hipLaunchKernelGGL(bag_pool_bwd_kernel<false>, dim3(stream_grid(batch, NRX_BLOCK / 64)), dim3(NRX_BLOCK), 0,
                            reinterpret_cast<hipStream_t>(stream), g_out, mask, batch, bag_len, dim, g_emb);
    NRX_LAUNCH_CHECK("nrx_bag_pool_bwd");
    return NRX_OK;
}

extern "C" int nrx_fm_fwd(const float* feat, int64_t ld, int32_t n_fields, int32_t dim, int64_t batch,
                          float* fm_out, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feat && fm_out && n_fields >= 1 && dim >= 1 && batch >= 0 && ld >= (int64_t)n_fields * dim,
                "nrx_fm_fwd: bad argument");
    if (batch == 0) return NRX_OK;
    int ql = ceil_log2i((dim + 3) / 4);
    if (ql > 6) ql = 6;
    const bool vec = (dim & 3) == 0 && (ld & 3) == 0 && nrx_aligned16(feat);
    const int tb = NRX_BLOCK >> ql;
    const unsigned grid = (unsigned)((batch + tb - 1) / tb);
    NRX_QSWITCH(ql, { hipLaunchKernelGGL((fm_fwd_kernel<QL>), dim3(grid), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                                         feat, ld, n_fields, dim, batch, fm_out, vec); });
    NRX_LAUNCH_CHECK("nrx_fm_fwd");
    return NRX_OK;
}

extern "C" int nrx_fm_fwd_train(const float* feat, int64_t ld, int32_t n_fields, int32_t dim, int64_t batch,
                                float* fm_out, float* fm_sums, int64_t sums_ld, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feat && fm_out && fm_sums && n_fields >= 1 && dim >= 1 && batch >= 0 && ld >= (int64_t)n_fields * dim && sums_ld >= dim,
                "nrx_fm_fwd_train: bad argument");
    if (batch == 0) return NRX_OK;
    int ql = ceil_log2i((dim + 3) / 4);
    if (ql > 6) ql = 6;
    const bool vec = (dim & 3) == 0 && (ld & 3) == 0 && nrx_aligned16(feat);
    const int tb = NRX_BLOCK >> ql;
    const unsigned grid = (unsigned)((batch + tb - 1) / tb);
    NRX_QSWITCH(ql, { hipLaunchKernelGGL((fm_fwd_kernel<QL>), dim3(grid), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                                         feat, ld, n_fields, dim, batch, fm_out, vec, fm_sums, sums_ld); });
    NRX_LAUNCH_CHECK("nrx_fm_fwd_train");
    return NRX_OK;
}

extern "C" int nrx_fm_bwd(const float* feat, int64_t ld, int32_t n_fields, int32_t dim, int64_t batch,
                          const float* g_fm, const float* g_in, int64_t g_in_ld, float* g_feat, int64_t g_ld, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feat && g_fm && g_feat && n_fields >= 1 && dim >= 1 && batch >= 0, "nrx_fm_bwd: bad argument");
    if (batch == 0) return NRX_OK;
    int ql = ceil_log2i((dim + 3) / 4);
    if (ql > 6) ql = 6;
    const bool vec = (dim & 3) == 0 && (ld & 3) == 0 && nrx_aligned16(feat);
    const int tb = NRX_BLOCK >> ql;
    const unsigned grid = (unsigned)((batch + tb - 1) / tb);
    NRX_QSWITCH(ql, { hipLaunchKernelGGL((fm_bwd_kernel<QL>), dim3(grid), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                                         feat, ld, n_fields, dim, batch, g_fm, g_in, g_in_ld, g_feat, g_ld, vec); });
    NRX_LAUNCH_CHECK("nrx_fm_bwd");
    return NRX_OK;
}

// ---- FM head: sigmoid(bias + logit) (FMModel.forward, sort/fm/model.py:25-26) and its autograd, one launch each.  As torch ops the head was two
// launches forward and three backward (add, sigmoid; sigmoid_backward, the bias gradient's sum, ...): 16 us of a 212 us captured step at B = 65 536.
namespace {
constexpr int FH_BLOCKS = 64;
__global__ __launch_bounds__(NRX_BLOCK) void fm_head_fwd_kernel(const float* __restrict__ logit, const float* __restrict__ bias, float* __restrict__ out,
                                                                int64_t batch) {
    const float b = bias != nullptr ? bias[0] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x; i < batch; i += (int64_t)gridDim.x * NRX_BLOCK) {
        const float x = b + logit[i];
        out[i] = 1.0f / (1.0f + expf(-x));
    }
}
// g_logit = g_out * p (1 - p); g_bias = sum over the batch, DETERMINISTIC: block j sums its own contiguous slice with fixed trees and leaves
// partial[j]; the block that arrives last (a counter in `state`, re-armed by that block) adds the partials in index order.
__global__ __launch_bounds__(NRX_BLOCK) void fm_head_bwd_kernel(const float* __restrict__ g_out, int64_t g_stride, const float* __restrict__ out,
                                                                float* __restrict__ g_logit, float* __restrict__ g_bias, uint32_t* __restrict__ state,
                                                                int64_t batch) {
    __shared__ float s_w[NRX_BLOCK / 64];
    __shared__ uint32_t s_last;
    const int64_t chunk = (batch + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < batch ? lo + chunk : batch;
    float acc = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += NRX_BLOCK) {
        const float p = out[i];
        const float g = g_out[i * g_stride] * (p * (1.0f - p));
        g_logit[i] = g;
        acc += g;
    }
    if (g_bias == nullptr) return;
    acc = nrx_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    float* partial = reinterpret_cast<float*>(state + 8);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        __threadfence();
        s_last = atomicAdd(&state[0], 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
        __threadfence();
        float v = (int)threadIdx.x < (int)gridDim.x ? __hip_atomic_load(&partial[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        v = nrx_wave_sum(v);
        if (threadIdx.x == 0) {
            g_bias[0] = v;
            state[0] = 0;
        }
    }
}
}  // namespace

extern "C" int nrx_fm_head_fwd(const float* logit, const float* bias, float* out, int64_t batch, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(logit && out && batch >= 0, "nrx_fm_head_fwd: bad argument");
    if (batch == 0) return NRX_OK;
    const int64_t g = (batch + NRX_BLOCK - 1) / NRX_BLOCK;
    hipLaunchKernelGGL(fm_head_fwd_kernel, dim3((unsigned)(g < 1024 ? g : 1024)), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), logit, bias, out, batch);
    NRX_LAUNCH_CHECK("nrx_fm_head_fwd");
    return NRX_OK;
}

extern "C" int64_t nrx_fm_head_state_bytes(void) { return 32 + FH_BLOCKS * 4; }

extern "C" int nrx_fm_head_bwd(const float* g_out, int64_t g_stride, const float* out, float* g_logit, float* g_bias, void* state, int64_t batch,
                               void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(g_out && out && g_logit && batch >= 0 && (g_stride == 0 || g_stride == 1), "nrx_fm_head_bwd: bad argument");
    NRX_REQUIRE(g_bias == nullptr || state != nullptr, "nrx_fm_head_bwd: the bias gradient needs the state words (nrx_fm_head_state_bytes, zero before the first use)");
    if (batch == 0) {
        if (g_bias != nullptr && nrx_zero_async(g_bias, 4, reinterpret_cast<hipStream_t>(stream)) != NRX_OK) return NRX_ERR_LAUNCH;
        return NRX_OK;
    }
    const int64_t g = (batch + 1023) / 1024;
    hipLaunchKernelGGL(fm_head_bwd_kernel, dim3((unsigned)(g < FH_BLOCKS ? g : FH_BLOCKS)), dim3(NRX_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), g_out, g_stride,
                       out, g_logit, g_bias, reinterpret_cast<uint32_t*>(state), batch);
    NRX_LAUNCH_CHECK("nrx_fm_head_bwd");
    return NRX_OK;
}

extern "C" int nrx_dcn_v1_fwd(const float* x, int64_t x_ld, const float* x0, int64_t x0_ld, int64_t batch, int32_t dim,
                              int32_t n_layers, const float* w, const float* b, float* out, int64_t out_ld, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(x && out && batch >= 0 && dim >= 1, "nrx_dcn_v1_fwd: bad argument");
    if (x0 == x && x0_ld == x_ld) x0 = nullptr;
    NRX_REQUIRE(n_layers >= 0 && n_layers <= NRX_MAX_DCN_LAYERS, "nrx_dcn_v1_fwd: n_layers must be in [0, %d]", NRX_MAX_DCN_LAYERS);
    NRX_REQUIRE(n_layers == 0 || (w && b), "nrx_dcn_v1_fwd: null cross weights");
    NRX_REQUIRE(dim <= 2048, "nrx_dcn_v1_fwd: dim %d > 2048 unsupported", dim);
    if (batch == 0) return NRX_OK;
    const bool vec = (dim & 3) == 0 && (x_ld & 3) == 0 && (out_ld & 3) == 0 && nrx_aligned16(x) && nrx_aligned16(out) &&
                     (x0 == nullptr || ((x0_ld & 3) == 0 && nrx_aligned16(x0)));
    const int V = vec ? 4 : 1;
    const int R = (dim + 64 * V - 1) / (64 * V);
    const size_t smem = (size_t)2 * n_layers * ((dim + 3) & ~3) * sizeof(float);
    NRX_REQUIRE(smem <= 64 * 1024, "nrx_dcn_v1_fwd: n_layers*dim too large for the LDS stage");
    if (vec && dim <= 128) {          // rows of <= 32 float4 chunks (the reference's own DCN width is 112): two rows per wavefront
        const unsigned grid2 = stream_grid((batch + 1) / 2, NRX_BLOCK / 64);
        hipLaunchKernelGGL((dcn_v1_fwd_kernel<1, 4, 32>), dim3(grid2), dim3(NRX_BLOCK), smem, reinterpret_cast<hipStream_t>(stream),
                           x, x_ld, x0, x0_ld, batch, dim, n_layers, w, b, out, out_ld);
        NRX_LAUNCH_CHECK("nrx_dcn_v1_fwd(half-wave rows)");
        return NRX_OK;
    }
    const unsigned grid = stream_grid(batch, NRX_BLOCK / 64);
    NRX_RSWITCH(R, V, { hipLaunchKernelGGL((dcn_v1_fwd_kernel<RR, VV>), dim3(grid), dim3(NRX_BLOCK), smem, reinterpret_cast<hipStream_t>(stream),
                                           x, x_ld, x0, x0_ld, batch, dim, n_layers, w, b, out, out_ld); });
    NRX_LAUNCH_CHECK("nrx_dcn_v1_fwd");
    return NRX_OK;
}

// ordered_ws (null: float atomics): every block leaves its g_w / g_b sums in its own slot of the scratch, *grid_out = the blocks launched
constexpr int DCN_V1_MAX_BLOCKS = 1024;
static int dcn_v1_bwd_impl(const float* x, int64_t x_ld, const float* x0, int64_t x0_ld, int64_t batch, int32_t dim,
                           int32_t n_layers, const float* w, const float* b, const float* g_out, int64_t g_out_ld,
                           float* g_x, int64_t g_x_ld, float* g_x0, int64_t g_x0_ld, float* g_w_user, float* g_b_user, void* stream,
                           float* ordered_ws, unsigned* grid_out) {
    const int ordered = ordered_ws != nullptr ? 1 : 0;
    float* g_w = ordered ? ordered_ws : g_w_user;
    float* g_b = ordered ? ordered_ws + (size_t)DCN_V1_MAX_BLOCKS * (n_layers > 0 ? n_layers : 1) * dim : g_b_user;
    unsigned last_grid = 0;
    struct GridOut { unsigned* p; unsigned* v; ~GridOut() { if (p) *p = *v; } } grid_guard{grid_out, &last_grid};
    NRX_REQUIRE(x && g_out && g_x && batch >= 0 && dim >= 1, "nrx_dcn_v1_bwd: bad argument");
    if (x0 == x && x0_ld == x_ld && g_x0 == nullptr) x0 = nullptr;
    NRX_REQUIRE((x0 == nullptr) == (g_x0 == nullptr), "nrx_dcn_v1_bwd: x0 and g_x0 go together (both null: x0 is x, one gradient)");
    NRX_REQUIRE(n_layers >= 0 && n_layers <= NRX_MAX_DCN_LAYERS, "nrx_dcn_v1_bwd: n_layers must be in [0, %d]", NRX_MAX_DCN_LAYERS);
    NRX_REQUIRE(n_layers == 0 || (w && b && g_w && g_b), "nrx_dcn_v1_bwd: null cross weights");
    if (ordered && n_layers > 0) {      // (the per-row LDS-atomic body -- n_layers x dim beyond the LDS slabs -- has no fixed order inside a block)
        const size_t Dp_ = (size_t)((dim + 3) & ~3);
        const bool vec_ = (dim & 3) == 0;
        if (((size_t)4 * n_layers + (size_t)(vec_ && dim <= 128 ? 8 : 4) * n_layers * 2) * Dp_ * sizeof(float) > 128 * 1024 &&
            !(n_layers <= 4 && (dim + 64 * (vec_ ? 4 : 1) - 1) / (64 * (vec_ ? 4 : 1)) <= 2)) {
            nrx_set_error("nrx_dcn_v1_bwd_ordered: n_layers * dim too large for the ordered mode");
            return NRX_ERR_UNSUPPORTED;
        }
    }
    NRX_REQUIRE(dim <= 2048, "nrx_dcn_v1_bwd: dim %d > 2048 unsupported", dim);
    if (batch == 0) return NRX_OK;
    const bool vec = (dim & 3) == 0 && (x_ld & 3) == 0 && (g_out_ld & 3) == 0 && (g_x_ld & 3) == 0 &&
                     nrx_aligned16(x) && nrx_aligned16(g_out) && nrx_aligned16(g_x) &&
                     (x0 == nullptr || ((x0_ld & 3) == 0 && (g_x0_ld & 3) == 0 && nrx_aligned16(x0) && nrx_aligned16(g_x0)));
    const int V = vec ? 4 : 1;
    const int R = (dim + 64 * V - 1) / (64 * V);
    // register accumulation of gw/gb when the accumulators fit (n_layers <= 4 and <= 2 chunks per lane)
    const int nlr = (n_layers >= 1 && n_layers <= 4 && R <= 2) ? n_layers : 0;
    // w, b, gw, gb [n_layers][Dp] (+ with register accumulation the per-wavefront slabs of the block-level gw / gb sum:
    // [slabs][2][Dp], one slab per wavefront of a block of up to 1024 threads -- two per wavefront in the two-rows-per-wavefront form)
    const int slabs = (vec && dim <= 128) ? 32 : 16;
    const size_t Dp = (size_t)((dim + 3) & ~3);
    size_t smem = ((size_t)4 * n_layers + (nlr > 0 ? 2 * slabs : 0)) * Dp * sizeof(float);
    NRX_REQUIRE(smem <= 128 * 1024, "nrx_dcn_v1_bwd: n_layers*dim too large for the LDS stage");
    // generic depth / width (no register accumulation): 256-thread blocks whose 4 wavefronts (8 halves) each own an LDS slab
    // [n_layers][2][Dp]; when even that does not fit, the per-row LDS-atomic body at 512 threads
    const size_t gen_smem = ((size_t)4 * n_layers + (size_t)(vec && dim <= 128 ? 8 : 4) * n_layers * 2) * Dp * sizeof(float);
    const bool gen_slabs = gen_smem <= 128 * 1024;
#define NRX_DCN_BWD(NLR_)                                                                                           \
    NRX_RSWITCH(R, V, {                                                                                             \
        constexpr int DCN_BWD_BLOCK = (NLR_ >= 1 && NLR_ <= 2) ? 1024 : 512;                                        \
        unsigned grid = (unsigned)((batch + DCN_BWD_BLOCK / 64 - 1) / (DCN_BWD_BLOCK / 64));                         \
        if (grid > 256u * (1024 / DCN_BWD_BLOCK)) grid = 256u * (1024 / DCN_BWD_BLOCK);                              \
        auto kern = dcn_v1_bwd_kernel<(RR <= 2 ? RR : 2), VV, NLR_, DCN_BWD_BLOCK>;      /* nlr > 0 only with R <= 2 */     \
        if (smem > 64 * 1024)                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        last_grid = grid; hipLaunchKernelGGL(kern, dim3(grid), dim3(DCN_BWD_BLOCK), smem, reinterpret_cast<hipStream_t>(stream), x, x_ld, \
                           batch, dim, n_layers, w, b, g_out, g_out_ld, g_x, g_x_ld, g_w, g_b,                      \
                           (const float*)nullptr, (int64_t)0, (float*)nullptr, (int64_t)0, ordered);                         \
    })
    if (vec && dim <= 128) {          // two rows per wavefront (see nrx_dcn_v1_fwd)
        const int64_t pairs = (batch + 1) / 2;
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define NRX_DCN_BWD_H(NLR_, BLK_, SEP_)                                                                             \
        {                                                                                                           \
            unsigned grid = (unsigned)((pairs + BLK_ / 64 - 1) / (BLK_ / 64));                                      \
            if (grid > 256u * (BLK_ >= 512 ? 1024 / BLK_ : 2)) grid = 256u * (BLK_ >= 512 ? 1024 / BLK_ : 2);                  \
            auto kern = dcn_v1_bwd_kernel<1, 4, NLR_, BLK_, SEP_, 32>;                                              \
            if (smem > 64 * 1024)                                                                                   \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            last_grid = grid; hipLaunchKernelGGL(kern, dim3(grid), dim3(BLK_), smem, st, x, x_ld, batch, dim, n_layers, w, b, g_out, g_out_ld, g_x, g_x_ld, \
                               g_w, g_b, x0, x0_ld, g_x0, g_x0_ld, ordered);                                                 \
        }
        if (x0 != nullptr && n_layers == 1) NRX_DCN_BWD_H(1, 1024, true)      // DCNLayer.forward(x_l, x_0): one layer, register accumulation
        else if (x0 != nullptr) { smem = gen_smem; NRX_DCN_BWD_H(0, 256, true) }
        else switch (nlr) {
            case 1: NRX_DCN_BWD_H(1, 1024, false) break;
            case 2: NRX_DCN_BWD_H(2, 1024, false) break;
            case 3: NRX_DCN_BWD_H(3, 512, false) break;
            case 4: NRX_DCN_BWD_H(4, 512, false) break;
            default: { smem = gen_smem; NRX_DCN_BWD_H(0, 256, false) } break;
        }
#undef NRX_DCN_BWD_H
        NRX_LAUNCH_CHECK("nrx_dcn_v1_bwd(half-wave rows)");
        return NRX_OK;
    }
    if (x0 != nullptr && n_layers == 1 && R <= 2) {      // DCNLayer.forward(x_l, x_0), one layer: the register-accumulation body
        NRX_RSWITCH(R, V, {
            unsigned grid = (unsigned)((batch + 1024 / 64 - 1) / (1024 / 64));
            if (grid > 256u) grid = 256u;
            auto kern = dcn_v1_bwd_kernel<(RR <= 2 ? RR : 2), VV, 1, 1024, true>;
            if (smem > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            last_grid = grid; hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), smem, reinterpret_cast<hipStream_t>(stream), x, x_ld, batch, dim,
                               n_layers, w, b, g_out, g_out_ld, g_x, g_x_ld, g_w, g_b, x0, x0_ld, g_x0, g_x0_ld, ordered);
        });
        NRX_LAUNCH_CHECK("nrx_dcn_v1_bwd(x0, one layer)");
        return NRX_OK;
    }
#define NRX_DCN_BWD_GEN(SEP_)                                                                                       \
    NRX_RSWITCH(R, V, {                                                                                             \
        if (gen_slabs) {                                                                                            \
            unsigned grid = (unsigned)((batch + 256 / 64 - 1) / (256 / 64));                                        \
            if (grid > 512u) grid = 512u;                                                                           \
            auto kern = dcn_v1_bwd_kernel<RR, VV, 0, 256, SEP_, 64, false>;                                         \
            if (gen_smem > 64 * 1024)                                                                               \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gen_smem); \
            last_grid = grid; hipLaunchKernelGGL(kern, dim3(grid), dim3(256), gen_smem, reinterpret_cast<hipStream_t>(stream), x, x_ld, batch, dim, \
                               n_layers, w, b, g_out, g_out_ld, g_x, g_x_ld, g_w, g_b, x0, x0_ld, g_x0, g_x0_ld, ordered);   \
        } else {                                                                                                    \
            unsigned grid = (unsigned)((batch + 512 / 64 - 1) / (512 / 64));                                        \
            if (grid > 512u) grid = 512u;                                                                           \
            auto kern = dcn_v1_bwd_kernel<RR, VV, 0, 512, SEP_, 64, true>;                                          \
            if (smem > 64 * 1024)                                                                                   \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            last_grid = grid; hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, reinterpret_cast<hipStream_t>(stream), x, x_ld, batch, dim, \
                               n_layers, w, b, g_out, g_out_ld, g_x, g_x_ld, g_w, g_b, x0, x0_ld, g_x0, g_x0_ld, ordered);   \
        }                                                                                                           \
    })
    if (x0 != nullptr) {      // separate layer-0 input, more than one layer (or a wide row): the generic body
        NRX_DCN_BWD_GEN(true);
        NRX_LAUNCH_CHECK("nrx_dcn_v1_bwd(x0)");
        return NRX_OK;
    }
    switch (nlr) {
        case 1: NRX_DCN_BWD(1); break;
        case 2: NRX_DCN_BWD(2); break;
        case 3: NRX_DCN_BWD(3); break;
        case 4: NRX_DCN_BWD(4); break;
        default: NRX_DCN_BWD_GEN(false); break;
    }
#undef NRX_DCN_BWD
#undef NRX_DCN_BWD_GEN
    NRX_LAUNCH_CHECK("nrx_dcn_v1_bwd");
    return NRX_OK;
}

namespace {
// ordered mode: g_w / g_b = the blocks' slots added in block order (64 outputs x 16 block groups per block; a fixed association)
__global__ __launch_bounds__(1024) void dcn_v1_reduce_kernel(const float* __restrict__ pw, const float* __restrict__ pb, int blocks, int n_out,
                                                              float* __restrict__ g_w, float* __restrict__ g_b) {
    __shared__ float s_p[16][64];
    const int li = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + li;
    float a0 = 0.f, a1 = 0.f;
    if (i < 2 * n_out) {
        const float* p = i < n_out ? pw + i : pb + (i - n_out);
        int sb = c;
        for (; sb + 16 < blocks; sb += 32) { a0 += p[(int64_t)sb * n_out]; a1 += p[(int64_t)(sb + 16) * n_out]; }
        if (sb < blocks) a0 += p[(int64_t)sb * n_out];
    }
    s_p[c][li] = a0 + a1;
    __syncthreads();
    if (c == 0 && i < 2 * n_out) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += s_p[k][li];
        if (i < n_out) g_w[i] = v;
        else g_b[i - n_out] = v;
    }
}
}  // namespace

extern "C" int nrx_dcn_v1_bwd(const float* x, int64_t x_ld, const float* x0, int64_t x0_ld, int64_t batch, int32_t dim,
                              int32_t n_layers, const float* w, const float* b, const float* g_out, int64_t g_out_ld,
                              float* g_x, int64_t g_x_ld, float* g_x0, int64_t g_x0_ld, float* g_w, float* g_b, void* stream) {
    NRX_TRACE();
    return dcn_v1_bwd_impl(x, x_ld, x0, x0_ld, batch, dim, n_layers, w, b, g_out, g_out_ld, g_x, g_x_ld, g_x0, g_x0_ld, g_w, g_b, stream, nullptr, nullptr);
}

extern "C" int64_t nrx_dcn_v1_bwd_ordered_workspace(int32_t dim, int32_t n_layers) {
    if (dim < 1 || n_layers < 0 || n_layers > NRX_MAX_DCN_LAYERS) return -1;
    return (int64_t)2 * DCN_V1_MAX_BLOCKS * (n_layers > 0 ? n_layers : 1) * dim * (int64_t)sizeof(float) + 512;
}

extern "C" int nrx_dcn_v1_bwd_ordered(const float* x, int64_t x_ld, const float* x0, int64_t x0_ld, int64_t batch, int32_t dim,
                                      int32_t n_layers, const float* w, const float* b, const float* g_out, int64_t g_out_ld,
                                      float* g_x, int64_t g_x_ld, float* g_x0, int64_t g_x0_ld, float* g_w, float* g_b, void* workspace,
                                      void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(workspace != nullptr, "nrx_dcn_v1_bwd_ordered: null workspace");
    NRX_REQUIRE(n_layers < 0 || n_layers > NRX_MAX_DCN_LAYERS || n_layers == 0 || (g_w != nullptr && g_b != nullptr), "nrx_dcn_v1_bwd_ordered: null g_w / g_b");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n_layers == 0 || batch == 0) {
        if (n_layers > 0 && g_w && g_b &&
            (nrx_zero_async(g_w, sizeof(float) * (size_t)n_layers * dim, st) != NRX_OK || nrx_zero_async(g_b, sizeof(float) * (size_t)n_layers * dim, st) != NRX_OK))
            return NRX_ERR_LAUNCH;
        return dcn_v1_bwd_impl(x, x_ld, x0, x0_ld, batch, dim, n_layers, w, b, g_out, g_out_ld, g_x, g_x_ld, g_x0, g_x0_ld, g_w, g_b, stream, nullptr, nullptr);
    }
    float* ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    unsigned grid = 0;
    const int rc = dcn_v1_bwd_impl(x, x_ld, x0, x0_ld, batch, dim, n_layers, w, b, g_out, g_out_ld, g_x, g_x_ld, g_x0, g_x0_ld, g_w, g_b, stream, ws, &grid);
    if (rc != NRX_OK) return rc;
    NRX_REQUIRE(grid >= 1 && grid <= (unsigned)DCN_V1_MAX_BLOCKS, "nrx_dcn_v1_bwd_ordered: unexpected launch shape");
    const int n_out = n_layers * dim;
    hipLaunchKernelGGL(dcn_v1_reduce_kernel, dim3((unsigned)((2 * n_out + 63) / 64)), dim3(1024), 0, st, (const float*)ws,
                       (const float*)(ws + (size_t)DCN_V1_MAX_BLOCKS * n_layers * dim), (int)grid, n_out, g_w, g_b);
    NRX_LAUNCH_CHECK("nrx_dcn_v1_bwd_ordered");
    return NRX_OK;
}

extern "C" int nrx_embed_dcn_v1_fwd(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t width,
                                    float* out, int64_t out_ld, int32_t n_layers, const float* w, const float* b,
                                    int32_t* status, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feats != nullptr && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES,
                "nrx_embed_dcn_v1_fwd: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(batch >= 0 && out != nullptr && width >= 4, "nrx_embed_dcn_v1_fwd: bad argument");
    NRX_REQUIRE(n_layers >= 1 && n_layers <= NRX_MAX_DCN_LAYERS && w && b, "nrx_embed_dcn_v1_fwd: n_layers must be in [1, %d]", NRX_MAX_DCN_LAYERS);
#define NRX_UNSUP(cond, ...) do { if (!(cond)) { nrx_set_error(__VA_ARGS__); return NRX_ERR_UNSUPPORTED; } } while (0)
    NRX_UNSUP((width & 3) == 0 && width <= 2048, "nrx_embed_dcn_v1_fwd: width must be a multiple of 4 and <= 2048");
    NRX_UNSUP((out_ld & 3) == 0 && out_ld >= 2 * (int64_t)width && nrx_aligned16(out), "nrx_embed_dcn_v1_fwd: out must be 16-byte aligned with out_ld %% 4 == 0 and >= 2*width");
    EmbedDcnArgs a;
    int prev_end = 0;
    for (int i = 0; i < n_feats; ++i) {
        const nrx_feature_t& s = feats[i];
        NRX_UNSUP(s.kind == NRX_SPARSE && s.wide_col < 0, "nrx_embed_dcn_v1_fwd: feature %d: only plain single-valued features are fused", i);
        NRX_UNSUP((s.dim & 3) == 0 && (s.out_col & 3) == 0 && nrx_aligned16(s.table), "nrx_embed_dcn_v1_fwd: feature %d: dim/out_col %% 4 and 16-byte aligned table required", i);
        NRX_REQUIRE(s.table && s.index && s.rows >= 1 && s.rows <= 0x7fffffffLL, "nrx_embed_dcn_v1_fwd: feature %d: bad table/index/rows", i);
        NRX_REQUIRE(s.index_bits == feats[0].index_bits && (s.index_bits == 32 || s.index_bits == 64), "nrx_embed_dcn_v1_fwd: mixed index widths");
        NRX_REQUIRE(s.out_col >= prev_end && s.out_col + s.dim <= width, "nrx_embed_dcn_v1_fwd: features must be ordered by out_col, non-overlapping, inside width");
        NRX_UNSUP(s.out_col == prev_end, "nrx_embed_dcn_v1_fwd: the concat must be gap-free");
        prev_end = s.out_col + s.dim;
        a.table[i] = s.table;
        a.index[i] = s.index;
        a.rows[i] = s.rows;
        a.col[i] = s.out_col;
    }
    NRX_UNSUP(prev_end == width, "nrx_embed_dcn_v1_fwd: features must fill the whole width");
#undef NRX_UNSUP
    a.col[n_feats] = width;
    if (batch == 0) return NRX_OK;
    a.batch = batch;
    a.out = out;
    a.out_ld = out_ld;
    a.w = w;
    a.b = b;
    a.status = status;
    a.n = n_feats;
    a.width = width;
    a.n_layers = n_layers;
    a.idx64 = feats[0].index_bits == 64;
    const size_t smem = (size_t)2 * n_layers * width * sizeof(float);
    NRX_REQUIRE(smem <= 64 * 1024, "nrx_embed_dcn_v1_fwd: n_layers*width too large for the LDS stage");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {   // grouped variant: every feature the same 128- or 256-byte row, at most 8 of them
        const int D0 = feats[0].dim;
        bool uni = (D0 == 32 || D0 == 64) && n_feats >= 2 && n_feats <= 8 && nrx_aligned16(w) && nrx_aligned16(b);
        for (int i = 0; i < n_feats && uni; ++i) uni = feats[i].dim == D0;
        if (uni) {
            const int tb = NRX_BLOCK / (D0 / 4);
            const dim3 ggrid((unsigned)((batch + tb - 1) / tb));
            int64_t table_bytes = 0;
            for (int i = 0; i < n_feats; ++i) table_bytes += feats[i].rows * (int64_t)D0 * 4;
            const bool nt = table_bytes > (256ll << 20);     // non-temporal row loads once the tables exceed the Infinity Cache
            // NRX_EDCN_VARIANT (measurement knob): 0 = one tile per block (round 1-3), 1 = persistent blocks, 2 = + non-temporal stores (default);
            // NRX_EDCN_BPC = resident blocks per CU of the persistent forms
            static const int variant = getenv("NRX_EDCN_VARIANT") ? atoi(getenv("NRX_EDCN_VARIANT")) : 2;
            static const int bpc = getenv("NRX_EDCN_BPC") ? atoi(getenv("NRX_EDCN_BPC")) : 16;
            const unsigned pg = ggrid.x < (unsigned)(256 * bpc) ? ggrid.x : (unsigned)(256 * bpc);
            const dim3 pgrid(pg);
#define NRX_EG3(QL_, N_, I_) do { \
        if (variant == 4 && nt) hipLaunchKernelGGL((embed_dcn_v1_group_persist_kernel<QL_, N_, I_, true, true, true>), pgrid, dim3(NRX_BLOCK), smem, st, a); \
        else if (variant == 4) hipLaunchKernelGGL((embed_dcn_v1_group_persist_kernel<QL_, N_, I_, false, true, true>), pgrid, dim3(NRX_BLOCK), smem, st, a); \
        else if (variant == 5 && nt) hipLaunchKernelGGL((embed_dcn_v1_group_persist_kernel<QL_, N_, I_, true, false, true>), pgrid, dim3(NRX_BLOCK), smem, st, a); \
        else if (variant == 5) hipLaunchKernelGGL((embed_dcn_v1_group_persist_kernel<QL_, N_, I_, false, false, true>), pgrid, dim3(NRX_BLOCK), smem, st, a); \
        else if (variant >= 2 && nt) hipLaunchKernelGGL((embed_dcn_v1_group_persist_kernel<QL_, N_, I_, true, true>), pgrid, dim3(NRX_BLOCK), smem, st, a); \
        else if (variant >= 2) hipLaunchKernelGGL((embed_dcn_v1_group_persist_kernel<QL_, N_, I_, false, true>), pgrid, dim3(NRX_BLOCK), smem, st, a); \
        else if (variant == 1 && nt) hipLaunchKernelGGL((embed_dcn_v1_group_persist_kernel<QL_, N_, I_, true, false>), pgrid, dim3(NRX_BLOCK), smem, st, a); \
        else if (variant == 1) hipLaunchKernelGGL((embed_dcn_v1_group_persist_kernel<QL_, N_, I_, false, false>), pgrid, dim3(NRX_BLOCK), smem, st, a); \
        else if (nt) hipLaunchKernelGGL((embed_dcn_v1_group_kernel<QL_, N_, I_, true>), ggrid, dim3(NRX_BLOCK), smem, st, a); \
        else hipLaunchKernelGGL((embed_dcn_v1_group_kernel<QL_, N_, I_, false>), ggrid, dim3(NRX_BLOCK), smem, st, a); } while (0)
#define NRX_EG2(QL_, N_) do { if (a.idx64) NRX_EG3(QL_, N_, true); else NRX_EG3(QL_, N_, false); } while (0)
#define NRX_EG(QL_) switch (n_feats) { case 2: NRX_EG2(QL_, 2); break; case 3: NRX_EG2(QL_, 3); break; case 4: NRX_EG2(QL_, 4); break; \
                                       case 5: NRX_EG2(QL_, 5); break; case 6: NRX_EG2(QL_, 6); break; case 7: NRX_EG2(QL_, 7); break; \
                                       default: NRX_EG2(QL_, 8); break; }
            if (D0 == 32) NRX_EG(3) else NRX_EG(4)
#undef NRX_EG
#undef NRX_EG2
#undef NRX_EG3
            NRX_LAUNCH_CHECK("nrx_embed_dcn_v1_fwd(grouped)");
            return NRX_OK;
        }
    }
    const int R = (width + 255) / 256;
    const unsigned grid = stream_grid((batch + 7) / 8, NRX_BLOCK / 64);
#define NRX_ED(R_, S_) hipLaunchKernelGGL((embed_dcn_v1_kernel<R_, S_>), dim3(grid), dim3(NRX_BLOCK), smem, st, a)
    if (R <= 1) NRX_ED(1, 8); else if (R <= 2) NRX_ED(2, 8); else if (R <= 4) NRX_ED(4, 4); else NRX_ED(8, 2);
#undef NRX_ED
    NRX_LAUNCH_CHECK("nrx_embed_dcn_v1_fwd");
    return NRX_OK;
}
