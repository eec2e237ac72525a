// DCN-v2 cross layer on the CDNA4 matrix cores (fp32 in / fp32 accumulate, exact f32 fma chain):
//     out = act( x0 * (x_l W^T + bias) + x_l )
// Reference: DCNv2Layer.forward + the ReLU DCNv2Net inserts after every layer
// (src/model/sort/dcn/dcn_arch.py:33-50, 73-91).  This is the only dense contraction on the
// path (2*D^2 flop per sample per layer) -> v_mfma_f32_32x32x2_f32; everything else is HBM-bound.
//
// Tiling: 256-thread block = 4 wavefronts computes a 128 x 64 output tile; wave (wm, wn) owns a
// 64 x 32 sub-tile = two 32x32 MFMA accumulators sharing one B fragment.  K is walked in steps of
// 32 through LDS.  Both operands are K-contiguous in memory (x rows; nn.Linear weight rows), so the
// global->LDS stage reads 128 B per row segment (8 lanes x float4) and stores TRANSPOSED,
// As[k][m] / Ws[k][n] with leading dimensions 129 / 65 (odd): the fragment reads
// (lane -> consecutive m or n at fixed k) and the transposing ds_write_b32 are both conflict-free.
// The Hadamard / bias / residual / ReLU epilogue is fused on the accumulator registers
// (C layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
#include "nrx_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 64, BK = 32;
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

template <bool RELU>
__global__ __launch_bounds__(256) void dcn_v2_layer_kernel(const float* __restrict__ x0, const float* __restrict__ xl, int64_t ld,
                                                           int64_t M, int N, const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ out, int64_t out_ld, bool vec) {
    __shared__ float As[BK * LDA];
    __shared__ float Ws[BK * LDW];
    const int K = N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int srow = tid >> 3;        // 0..31
    const int skq = (tid & 7) * 4;    // k offset inside the BK slab

    for (int k0 = 0; k0 < K; k0 += BK) {
        float4 a[4], w[2];
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p] = guarded_load4(xl, m0 + srow + 32 * p, M, ld, k0 + skq, K, vec);
#pragma unroll
        for (int p = 0; p < 2; ++p) w[p] = guarded_load4(W, n0 + srow + 32 * p, N, K, k0 + skq, K, vec && ((K & 3) == 0));
        __syncthreads();   // previous slab fully consumed
#pragma unroll
        for (int p = 0; p < 4; ++p) {
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
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float bf = Ws[(kk + hi) * LDW + wn * 32 + l31];
            const float a0 = As[(kk + hi) * LDA + wm * 64 + l31];
            const float a1 = As[(kk + hi) * LDA + wm * 64 + 32 + l31];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf, acc1, 0, 0, 0);
        }
    }

    const int col = n0 + wn * 32 + l31;
    if (col < N) {
        const float bc = bias[col];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < M) {
                    const float lin = (t == 0 ? acc0[r] : acc1[r]) + bc;
                    float v = x0[row * ld + col] * lin + xl[row * ld + col];
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
    const bool vec = (ld & 3) == 0 && nrx_aligned16(xl) && nrx_aligned16(W);
    dim3 grid((dim + BN - 1) / BN, (unsigned)((batch + BM - 1) / BM));
    NRX_REQUIRE(grid.y <= 65535u * 16u, "nrx_dcn_v2_layer_fwd: batch too large for one launch");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (relu)
        hipLaunchKernelGGL(dcn_v2_layer_kernel<true>, grid, dim3(256), 0, st, x0, xl, ld, batch, dim, W, bias, out, out_ld, vec);
    else
        hipLaunchKernelGGL(dcn_v2_layer_kernel<false>, grid, dim3(256), 0, st, x0, xl, ld, batch, dim, W, bias, out, out_ld, vec);
    NRX_LAUNCH_CHECK("nrx_dcn_v2_layer_fwd");
    return NRX_OK;
}
