// DCN-v2 cross layer on the CDNA4 matrix cores (fp32 in / fp32 accumulate, exact f32 fma chain):
//     out = act( x0 * (x_l W^T + bias) + x_l )
// Reference: DCNv2Layer.forward + the ReLU DCNv2Net inserts after every layer
// (src/model/sort/dcn/dcn_arch.py:33-50, 73-91).  This is the only dense contraction on the
// path (2*D^2 flop per sample per layer) -> v_mfma_f32_32x32x2_f32; everything else is HBM-bound.
//
// Tiling: 256-thread block = 4 wavefronts computes a 256 x 64 output tile; wave (wm, wn) owns a
// 128 x 32 sub-tile = four 32x32 MFMA accumulators sharing one B fragment (64 MFMAs = 4096 matrix-
// pipe cycles per 32-deep K slab between barriers; 1.25 LDS dwords per MFMA).  K is walked in steps
// of 32 through LDS.  Both operands are K-contiguous in memory (x rows; nn.Linear weight rows), so the
// global->LDS stage reads 128 B per row segment (8 lanes x float4) and stores TRANSPOSED,
// As[k][m] / Ws[k][n] with leading dimensions 129 / 65 (odd): the fragment reads
// (lane -> consecutive m or n at fixed k) and the transposing ds_write_b32 are both conflict-free.
// The Hadamard / bias / residual / ReLU epilogue is fused on the accumulator registers
// (C layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
#include "nrx_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TM = 2;                       // 32-row MFMA tiles per wave along M
constexpr int BM = 2 * TM * 32, BN = 64, BK = 32;
constexpr int LDA = BM + 1, LDW = BN + 1;

__device__ __forceinline__ float4 guarded_load4(const float* base, int64_t row, int64_t nrows, int64_t ld, int k, int K, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows && k < K) {
        const float* p = base + row * ld + k;
        if (vec && k + 4 <= K) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
            if (k + 3 < K) v.w = p[3];
        }
    }
    return v;
}

// Branch-free 16-byte load for the aligned fast path (ld % 4 == 0, K % 4 == 0): out-of-range rows /
// k are clamped to a valid address and the result is zeroed with selects, so the six prefetch loads of
// a slab are issued back-to-back with no exec-mask control flow.
__device__ __forceinline__ float4 clamped_load4(const float* base, int64_t row, int64_t nrows, int64_t ld, int k, int K) {
    const bool ok = row < nrows && k < K;
    const int64_t r = row < nrows ? row : nrows - 1;
    const int kc = k < K ? k : 0;
    float4 v = nrx_ldg4(base + r * ld + kc, 0);
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}

template <bool RELU, bool VEC>
__global__ __launch_bounds__(256, 5) void dcn_v2_layer_kernel(const float* __restrict__ x0, const float* __restrict__ xl, int64_t ld,
                                                           int64_t M, int N, const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ out, int64_t out_ld, unsigned nx) {
    __shared__ float As[BK * LDA];
    __shared__ float Ws[BK * LDW];
    const int K = N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware tile order (guide T1, bijective form): hardware places block b on XCD b % 8; remap so
    // that the nx column tiles of one 128-row panel of x_l are consecutive on ONE XCD and share its L2
    // (without it the panel was fetched from DRAM once per XCD: 516 MB read vs ~250 MB, measured).
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    const unsigned xcd = bid & 7u, qd = nb >> 3, rm = nb & 7u;
    const unsigned logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int64_t m0 = (int64_t)(logical / nx) * BM;
    const int n0 = (int)(logical % nx) * BN;

    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int srow = tid >> 3;        // 0..31
    const int skq = (tid & 7) * 4;    // k offset inside the BK slab

    // Register-prefetch pipeline: the global loads of slab k+1 are in flight while slab k's 32 MFMAs
    // per wave (2048 cycles) run out of LDS; one LDS buffer, two barriers per slab.
    constexpr int AP = BM / 32;       // float4 loads of the A slab per thread
    float4 a[AP], w[2];
#pragma unroll
    for (int p = 0; p < AP; ++p) a[p] = VEC ? clamped_load4(xl, m0 + srow + 32 * p, M, ld, skq, K) : guarded_load4(xl, m0 + srow + 32 * p, M, ld, skq, K, false);
#pragma unroll
    for (int p = 0; p < 2; ++p) w[p] = VEC ? clamped_load4(W, n0 + srow + 32 * p, N, K, skq, K) : guarded_load4(W, n0 + srow + 32 * p, N, K, skq, K, false);

    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int m = srow + 32 * p;
            As[(skq + 0) * LDA + m] = a[p].x;
            As[(skq + 1) * LDA + m] = a[p].y;
            As[(skq + 2) * LDA + m] = a[p].z;
            As[(skq + 3) * LDA + m] = a[p].w;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int n = srow + 32 * p;
            Ws[(skq + 0) * LDW + n] = w[p].x;
            Ws[(skq + 1) * LDW + n] = w[p].y;
            Ws[(skq + 2) * LDW + n] = w[p].z;
            Ws[(skq + 3) * LDW + n] = w[p].w;
        }
        __syncthreads();
        if (k0 + BK < K) {           // next slab: loads stay in flight across the MFMA block below
#pragma unroll
            for (int p = 0; p < AP; ++p) a[p] = VEC ? clamped_load4(xl, m0 + srow + 32 * p, M, ld, k0 + BK + skq, K) : guarded_load4(xl, m0 + srow + 32 * p, M, ld, k0 + BK + skq, K, false);
#pragma unroll
            for (int p = 0; p < 2; ++p) w[p] = VEC ? clamped_load4(W, n0 + srow + 32 * p, N, K, k0 + BK + skq, K) : guarded_load4(W, n0 + srow + 32 * p, N, K, k0 + BK + skq, K, false);
        }
        // fragments of half a slab (8 k-pairs: 8 B + 32 A dwords) are read ahead of a dense block of
        // 32 MFMAs, so the LDS latency is paid twice per slab instead of once per MFMA group
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float fb[BK / 4], fa[TM][BK / 4];
#pragma unroll
            for (int i = 0; i < BK / 4; ++i) {
                const int kr = 2 * (h * (BK / 4) + i) + hi;
                fb[i] = Ws[kr * LDW + wn * 32 + l31];
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[t][i] = As[kr * LDA + wm * (32 * TM) + 32 * t + l31];
            }
#pragma unroll
            for (int i = 0; i < BK / 4; ++i)
#pragma unroll
                for (int t = 0; t < TM; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t][i], fb[i], acc[t], 0, 0, 0);
        }
        __syncthreads();             // slab fully consumed before the next LDS write
    }

    const int col = n0 + wn * 32 + l31;
    const bool same_x = (x0 == xl);
    if (col < N) {
        const float bc = bias[col];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * (32 * TM) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < M) {
                    const float lin = acc[t][r] + bc;
                    const float xv = xl[row * ld + col];
                    const float x0v = same_x ? xv : x0[row * ld + col];     // layer 0: x0 is x_l, one load
                    float v = x0v * lin + xv;
                    if (RELU) v = v > 0.f ? v : 0.f;
                    out[row * out_ld + col] = v;
                }
            }
        }
    }
}

}  // namespace

extern "C" int nrx_dcn_v2_layer_fwd(const float* x0, const float* xl, int64_t ld, int64_t batch, int32_t dim,
                                    const float* W, const float* bias, int32_t relu, float* out,
                                    int64_t out_ld, void* stream) {
    NRX_REQUIRE(x0 && xl && W && bias && out && batch >= 0 && dim >= 1 && ld >= dim && out_ld >= dim,
                "nrx_dcn_v2_layer_fwd: bad argument");
    NRX_REQUIRE(out != xl && out != x0, "nrx_dcn_v2_layer_fwd: out must not alias the inputs");
    if (batch == 0) return NRX_OK;
    const bool vec = (ld & 3) == 0 && (dim & 3) == 0 && nrx_aligned16(xl) && nrx_aligned16(W);
    const unsigned nx = (unsigned)((dim + BN - 1) / BN);
    const int64_t nblocks = (int64_t)nx * ((batch + BM - 1) / BM);
    NRX_REQUIRE(nblocks <= 0x7fffffffLL, "nrx_dcn_v2_layer_fwd: batch too large for one launch");
    dim3 grid((unsigned)nblocks);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define NRX_DCN2(R_, V_) hipLaunchKernelGGL((dcn_v2_layer_kernel<R_, V_>), grid, dim3(256), 0, st, x0, xl, ld, batch, dim, W, bias, out, out_ld, nx)
    if (relu) { if (vec) NRX_DCN2(true, true); else NRX_DCN2(true, false); }
    else      { if (vec) NRX_DCN2(false, true); else NRX_DCN2(false, false); }
#undef NRX_DCN2
    NRX_LAUNCH_CHECK("nrx_dcn_v2_layer_fwd");
    return NRX_OK;
}
