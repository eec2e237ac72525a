// Backward of the DCN-v2 cross layer on the CDNA4 matrix cores (fp32 in / fp32 accumulate).
// Reference arithmetic: autograd of DCNv2Layer.forward + the ReLU DCNv2Net puts after it
// (src/model/sort/dcn/dcn_arch.py:33-50, 73-91):   out = act(x0 * lin + xl),  lin = xl W^T + b.
// Given g = dL/dout (gm = g where out > 0 with ReLU, else g):
//     glin = gm * x0                 g_x0 += gm * lin                 g_b = sum_rows glin
//     g_xl = gm + glin W             (dgrad:  [B, D] x [D, D])
//     g_W  = glin^T xl               (wgrad:  [D, B] x [B, D], the batch is the contraction)
// Three launches per layer:
//   dcn_v2_bwd_prep_kernel   elementwise: gm, glin, g_x0 accumulation, column sums of glin (HBM-bound)
//   dcn2_gemm_kernel<DGRAD>  g_xl = gm + glin W: the forward kernel's tiling (128 x 64 block tile, 4 waves x two 32x32
//                            accumulators sharing a B fragment, K in slabs of 32 through LDS, register prefetch); the B
//                            operand W is read K-major (its rows are the contraction index), so its slab lands in LDS
//                            without a transpose
//   dcn2_gemm_kernel<WGRAD>  g_W += glin^T xl: both operands K-major (a batch row is contiguous over the output index),
//                            split over the batch: every block owns one 128 x 64 tile of g_W and a slice of the batch and
//                            adds its partial tile with fp32 atomics (g_W pre-zeroed; D*D outputs x ~40 slices)
// The forward saves lin when asked (nrx_dcn_v2_layer_fwd's lin_out): one extra [B, D] write instead of a third GEMM here.
#include "nrx_common.h"
#include <stdlib.h>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TM = 2;
constexpr int BM = 2 * TM * 32, BN = 64, BK = 32;      // 128 x 64 block tile, K slabs of 32
constexpr int LDW = BN + 1;

enum { DGRAD = 0, WGRAD = 1 };

// ---- elementwise preparation ------------------------------------------------------------------------------------
// glin = gm * x0 (the GEMMs' operand), g_x0 (+)= gm * lin, g_b = column sums of glin; gm itself is not materialised: the
// ReLU mask [out > 0] leaves this kernel as BITS, transposed for the dgrad epilogue -- maskT[row / 32][col] holds the 32 rows
// of one column, so a lane of the GEMM (one column, 16 rows of a 32-row MFMA tile) reads ONE dword per tile instead of 16
// values of `out` (re-reading `out` there cost 30 us of the 165 us launch at D = 320).  TPR = 2^TPRLOG2 threads cover one
// row (4 columns each); a thread owns its 4 columns over one GROUP of 32 consecutive rows (8 passes of U = 4 rows in flight,
// HBM-bound streaming: 4-5 reads + 2 writes of [B, D]) and collects the group's mask bits and its column sums in registers;
// a block covers 256 / TPR groups per sweep; then one atomic per column per block.
template <int TPRLOG2, bool VEC>
__global__ __launch_bounds__(NRX_BLOCK) void dcn_v2_bwd_prep_kernel(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ out,
                                                                  const float* __restrict__ x0, const float* __restrict__ lin,
                                                                  int64_t ld, int64_t M, int D, int relu,
                                                                  float* __restrict__ glin, int64_t w_ld, float* __restrict__ g_x0,
                                                                  int64_t gx0_ld, int accumulate_x0, float* __restrict__ g_b,
                                                                  uint32_t* __restrict__ maskT) {
    constexpr int TPR = 1 << TPRLOG2, RPB = NRX_BLOCK / TPR, U = 4;
    const int cc = (threadIdx.x & (TPR - 1)) * 4;          // D <= 4 TPR (the launch picks TPR; dims up to 1024)
    const int rsub = threadIdx.x >> TPRLOG2;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = VEC && cc + 4 <= D;
    const int64_t ngroups = (M + 31) >> 5;
    if (cc < D)
    for (int64_t gi = (int64_t)blockIdx.x * RPB + rsub; gi < ngroups; gi += (int64_t)gridDim.x * RPB) {
        uint32_t bits[4] = {0u, 0u, 0u, 0u};
        for (int it = 0; it < 32 / U; ++it) {
            const int64_t r0 = gi * 32 + it * U;
            if (r0 >= M) break;
            float4 gv[U], xv[U], lv[U], ov[U], ax[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + u;
                const int64_t rc = r < M ? r : M - 1;
                if (full) {
                    gv[u] = *reinterpret_cast<const float4*>(g + rc * g_ld + cc);
                    xv[u] = *reinterpret_cast<const float4*>(x0 + rc * ld + cc);
                    lv[u] = *reinterpret_cast<const float4*>(lin + rc * ld + cc);
                    ov[u] = relu ? *reinterpret_cast<const float4*>(out + rc * ld + cc) : make_float4(1.f, 1.f, 1.f, 1.f);
                    ax[u] = accumulate_x0 ? *reinterpret_cast<const float4*>(g_x0 + rc * gx0_ld + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    float t[5][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool in = cc + j < D;
                        t[0][j] = in ? g[rc * g_ld + cc + j] : 0.f;
                        t[1][j] = in ? x0[rc * ld + cc + j] : 0.f;
                        t[2][j] = in ? lin[rc * ld + cc + j] : 0.f;
                        t[3][j] = (in && relu) ? out[rc * ld + cc + j] : 1.f;
                        t[4][j] = (in && accumulate_x0) ? g_x0[rc * gx0_ld + cc + j] : 0.f;
                    }
                    gv[u] = make_float4(t[0][0], t[0][1], t[0][2], t[0][3]); xv[u] = make_float4(t[1][0], t[1][1], t[1][2], t[1][3]);
                    lv[u] = make_float4(t[2][0], t[2][1], t[2][2], t[2][3]); ov[u] = make_float4(t[3][0], t[3][1], t[3][2], t[3][3]);
                    ax[u] = make_float4(t[4][0], t[4][1], t[4][2], t[4][3]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + u;
                if (r >= M) continue;
                const bool k0 = !relu || ov[u].x > 0.f, k1 = !relu || ov[u].y > 0.f, k2 = !relu || ov[u].z > 0.f, k3 = !relu || ov[u].w > 0.f;
                const int sh = it * U + u;
                bits[0] |= (uint32_t)k0 << sh; bits[1] |= (uint32_t)k1 << sh; bits[2] |= (uint32_t)k2 << sh; bits[3] |= (uint32_t)k3 << sh;
                float4 m_;
                m_.x = k0 ? gv[u].x : 0.f; m_.y = k1 ? gv[u].y : 0.f; m_.z = k2 ? gv[u].z : 0.f; m_.w = k3 ? gv[u].w : 0.f;
                const float4 gl = make_float4(m_.x * xv[u].x, m_.y * xv[u].y, m_.z * xv[u].z, m_.w * xv[u].w);
                const float4 gx = make_float4(fmaf(m_.x, lv[u].x, ax[u].x), fmaf(m_.y, lv[u].y, ax[u].y), fmaf(m_.z, lv[u].z, ax[u].z),
                                              fmaf(m_.w, lv[u].w, ax[u].w));
                bs[0] += gl.x; bs[1] += gl.y; bs[2] += gl.z; bs[3] += gl.w;
                if (full) {
                    *reinterpret_cast<float4*>(glin + r * w_ld + cc) = gl;
                    *reinterpret_cast<float4*>(g_x0 + r * gx0_ld + cc) = gx;
                } else {
                    const float gla[4] = {gl.x, gl.y, gl.z, gl.w}, gxa[4] = {gx.x, gx.y, gx.z, gx.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cc + j < D) { glin[r * w_ld + cc + j] = gla[j]; g_x0[r * gx0_ld + cc + j] = gxa[j]; }
                }
            }
        }
        if (relu)                                       // w_ld is a multiple of 4 and cc < D <= w_ld: the 16-byte store stays in the row
            *reinterpret_cast<uint4*>(maskT + gi * w_ld + cc) = make_uint4(bits[0], bits[1], bits[2], bits[3]);
    }
    // column sums: the RPB row-lanes of a column meet in LDS, then ONE device atomic per column per block (same-address
    // device atomics serialise: one per thread made this kernel 8x slower)
    __shared__ float s_b[NRX_BLOCK * 4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s_b[(rsub * TPR + (threadIdx.x & (TPR - 1))) * 4 + j] = bs[j];
    __syncthreads();
    for (int i = threadIdx.x; i < TPR * 4; i += NRX_BLOCK) {
        float t = 0.f;
        for (int r = 0; r < RPB; ++r) t += s_b[r * TPR * 4 + i];
        if (i < D && g_b != nullptr) unsafeAtomicAdd(g_b + i, t);      // (null: the column sums come from the wgrad launch -- the ordered mode)
    }
}

// Epilogue of a block tile, shared by the fp32 and the split-bf16 GEMM kernels (same accumulator layout: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
template <int MODE, int TM>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[TM], int64_t m0, int n0, int wm, int wn, int l31, int hi, int64_t M, int N,
                                              const float* __restrict__ addend, int64_t add_ld, const uint32_t* __restrict__ maskT,
                                              int64_t mask_ld, float* __restrict__ out, int64_t out_ld, const float* __restrict__ add2,
                                              int64_t add2_ld) {
    const int col = n0 + wn * 32 + l31;
    if (col >= N) return;
    const int64_t r0 = m0 + wm * (32 * TM);
    if (MODE == DGRAD && r0 + 32 * TM <= M) {
        // full tile: a wave-uniform row pointer (scalar unit) + one fixed per-lane byte offset per array, the loads of a
        // 32-row sub-tile batched ahead of the arithmetic (as the forward's epilogue; per-element 64-bit address arithmetic and
        // load -> use chains made this epilogue the difference between 186 us here and 155 us for the forward)
        const uint32_t lo_a = (uint32_t)(((int64_t)(4 * hi) * add_ld + col) * 4);
        const uint32_t lo_2 = (uint32_t)(((int64_t)(4 * hi) * add2_ld + col) * 4);
        const uint32_t lo_o = (uint32_t)(((int64_t)(4 * hi) * out_ld + col) * 4);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            float gv[16], av[16];
            // the ReLU mask of this lane's column over the tile's 32 rows: bit (r & 3) + 8 (r >> 2) + 4 hi is row r's
            const uint32_t mw = (maskT != nullptr ? maskT[((r0 + t * 32) >> 5) * mask_ld + col] : 0xffffffffu) >> (4 * hi);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = r0 + t * 32 + (r & 3) + 8 * (r >> 2);          // + 4 * hi, folded into the lane offsets
                gv[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(addend + row * add_ld) + lo_a);
                av[r] = add2 != nullptr ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(add2 + row * add2_ld) + lo_2) : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = r0 + t * 32 + (r & 3) + 8 * (r >> 2);
                float v = ((mw >> ((r & 3) + 8 * (r >> 2))) & 1u ? gv[r] : 0.f) + acc[t][r];
                if (add2 != nullptr) v += av[r];
                *reinterpret_cast<float*>(reinterpret_cast<char*>(out + row * out_ld) + lo_o) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = r0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < M) {
                if (MODE == DGRAD) {        // g_xl = gm + glin W, gm = g (x) [forward output > 0] rebuilt here
                    float gmv = addend[row * add_ld + col];
                    if (maskT != nullptr && !((maskT[(row >> 5) * mask_ld + col] >> (row & 31)) & 1u)) gmv = 0.f;
                    float v = gmv + acc[t][r];
                    if (add2 != nullptr) v += add2[row * add2_ld + col];      // layer 0 (x0 is x_l): dL/dx = g_xl + g_x0 in one pass
                    out[row * out_ld + col] = v;
                } else if (add2 != nullptr) {      // ordered mode: this batch slice's partial tile, summed in slice order by wgrad_reduce_kernel
                    const_cast<float*>(add2)[row * (int64_t)N + col] = acc[t][r];
                } else {
                    unsafeAtomicAdd(out + row * out_ld + col, acc[t][r]);
                }
            }
        }
    }
}

// ---- the GEMM ----------------------------------------------------------------------------------------------------
// C[m, n] = sum_k A(m, k) B(k, n) over k in [kbeg, kend).
//   DGRAD: A(m, k) = glin[m * lda + k]   (M-major rows, transposed into LDS like the forward's x slab)
//          B(k, n) = W[k * ldb + n]      (K-major: copied straight)           epilogue: out[m, n] = gm[m, n] + C
//   WGRAD: A(m, k) = glin[k * lda + m]   (K-major)     B(k, n) = xl[k * ldb + n]  (K-major)   epilogue: atomicAdd(out[m, n], C)
// M, N, K are the GEMM's own dims (DGRAD: batch, D, D;  WGRAD: D, D, batch).  Rows / columns past M / N are clamped on
// load (their products land in accumulator rows / columns that are never stored); a partial last K slab is zero-filled.
// Block -> (tile, batch slice).  Hardware places block b on XCD b % 8, each XCD has its own L2.
//   DGRAD (one slice): the nx column tiles of one row panel are consecutive on ONE XCD and share the panel in its L2 (the forward's
//   bijective remap).
//   WGRAD: every tile of the output reads a column block of A and one of B over the SAME batch slice, so ALL tiles of one slice go to one
//   XCD, consecutively (slice = 8 * round + xcd): the slice's rows are fetched from DRAM once and shared through that L2 by the tiles
//   running side by side.  In block order tile-fastest over all XCDs (round 2) the 25 tiles of a slice at D = 320 were spread over the
//   8 L2s: 616 MB fetched per launch for 168 MB of operands, 7.0 TB/s -- the launch was bound by exactly that
//   (profiles/r03_dcn_v2_bf16x3.txt).
// (With the round-2 slice count the fp32 launch got SLOWER under this order -- D = 320: 133 -> 155 us, step 485 -> 510 us -- until the count
// was balanced over the XCDs' block slots: launch_wgrad.)
template <int MODE, bool SLICE_PER_XCD>
__device__ __forceinline__ void gemm_block_to_tile(unsigned bid, unsigned ntiles, unsigned& tile, unsigned& ks) {
    if (MODE == WGRAD && SLICE_PER_XCD) {
        const unsigned xcd = bid & 7u, seq = bid >> 3;
        ks = (seq / ntiles) * 8u + xcd;
        tile = seq % ntiles;
    } else {
        const unsigned tile_lin = bid % ntiles;
        ks = bid / ntiles;
        const unsigned xcd = tile_lin & 7u, qd = ntiles >> 3, rm = ntiles & 7u;
        tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (tile_lin >> 3);
    }
}

template <int MODE, bool VEC, int TMv = 2>
__global__ __launch_bounds__(256, 4) void dcn2_gemm_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                         int64_t M, int N, int64_t K, int64_t kslice, const float* __restrict__ addend,
                                                         int64_t add_ld, const uint32_t* __restrict__ maskT, int64_t mask_ld,
                                                         float* __restrict__ out, int64_t out_ld, unsigned nx, unsigned ntiles,
                                                         const float* __restrict__ add2, int64_t add2_ld, float* __restrict__ colsum,
                                                         int slice_per_xcd = 0) {
    constexpr int TM = TMv, BM = 2 * TM * 32, LDA = BM + 1;      // TMv = 1: 64 x 64 block tiles (wgrad of narrow layers: twice the tiles)
    __shared__ float As[BK * LDA];
    __shared__ float Ws[BK * LDW];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    // tile id (XCD-aware bijective remap as in the forward: the nx column tiles of one row panel share an L2) and K slice
    unsigned logical, ks;
    if (MODE == WGRAD && slice_per_xcd) gemm_block_to_tile<MODE, true>(blockIdx.x, ntiles, logical, ks);
    else gemm_block_to_tile<MODE, false>(blockIdx.x, ntiles, logical, ks);
    if ((int64_t)ks * kslice >= K) return;           // WGRAD: the slice count is rounded up to whole XCD rounds (block-uniform exit)
    const int64_t m0 = (int64_t)(logical / nx) * BM;
    const int n0 = (int)(logical % nx) * BN;
    const int64_t kbeg = (int64_t)ks * kslice;
    const int64_t kend = kbeg + kslice < K ? kbeg + kslice : K;

    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    constexpr int AP = BM / 32;           // float4 per thread of the A slab (4), B slab: 2
    float4 a[AP], w[2];
    auto zero4 = [] { return make_float4(0.f, 0.f, 0.f, 0.f); };
    // Aligned (VEC) path, written for few VALU instructions per slab like the forward kernel (VALU time is not hidden under
    // the fp32 MFMA): wave-uniform tile bases advanced on the scalar unit + fixed 32-bit per-thread byte offsets; rows /
    // columns past the edge are CLAMPED (their products land in accumulator rows / columns that are never stored), and only
    // a partial last K slab pays for zero selects.  M-major operand: thread (row = tid / 8 (+32 p), 4 floats at k = 4 (tid % 8));
    // K-major operand: thread (k = tid / 8, 4 floats at column 4 (tid % 8 + 8 p)).
    const int srow = tid >> 3, skq = (tid & 7) * 4;
    const int kk = tid >> 3, c4 = tid & 7;
    uint32_t oa[AP], ow[2];
    const char* abase;
    const char* bbase = reinterpret_cast<const char*>(B + n0);
    if (MODE == DGRAD) {
        abase = reinterpret_cast<const char*>(A + m0 * lda);
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int64_t r = m0 + srow + 32 * p < M ? srow + 32 * p : M - 1 - m0;
            oa[p] = (uint32_t)((r * lda + skq) * 4);
        }
    } else {
        abase = reinterpret_cast<const char*>(A + m0);
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            int64_t m = 4 * (c4 + 8 * p);
            if (m0 + m + 4 > M) m = M - 4 - m0 > 0 ? M - 4 - m0 : 0;
            oa[p] = (uint32_t)(((int64_t)kk * lda + m) * 4);
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        int n = 4 * (c4 + 8 * p);
        if (n0 + n + 4 > N) n = N - 4 - n0 > 0 ? N - 4 - n0 : 0;
        ow[p] = (uint32_t)(((int64_t)kk * ldb + n) * 4);
    }
    auto load_slab = [&](int64_t k0, float4 (&a)[AP], float4 (&w)[2]) {
        if (VEC) {
            // K-major operands advance by whole rows of the source, the M-major one by 32 floats
            const char* ak = abase + (MODE == DGRAD ? (size_t)k0 * 4 : (size_t)k0 * (size_t)lda * 4);
            const char* bk = bbase + (size_t)k0 * (size_t)ldb * 4;
            if (k0 + BK <= kend) {
#pragma unroll
                for (int p = 0; p < AP; ++p) a[p] = *reinterpret_cast<const float4*>(ak + oa[p]);
#pragma unroll
                for (int p = 0; p < 2; ++p) w[p] = *reinterpret_cast<const float4*>(bk + ow[p]);
            } else {                    // partial last slab: out-of-range k re-reads slab kbeg's address and is zeroed
                const bool oka = MODE == DGRAD ? k0 + skq + 4 <= kend : k0 + kk < kend;
                const bool okb = k0 + kk < kend;
                // out-of-range lanes read the slice's FIRST k (column kbeg of their row / row kbeg of the K-major source): always inside the
                // operand.  (They used to read "their" position of slab kbeg -- k = kbeg + skq, or source row kbeg + kk -- which lies past the
                // row / past the batch when the slice is shorter than a slab: D = 16, or a batch of 11 rows read 20 rows past the end of the
                // tensor.  The values were zeroed, so the results were right; the read faulted when the tensor ended at the end of a mapping.)
                const size_t backa = oka ? 0 : (MODE == DGRAD ? (size_t)(k0 - kbeg + skq) * 4 : (size_t)(k0 - kbeg + kk) * (size_t)lda * 4);
                const size_t backb = okb ? 0 : (size_t)(k0 - kbeg + kk) * (size_t)ldb * 4;
#pragma unroll
                for (int p = 0; p < AP; ++p) {
                    const float4 t = *reinterpret_cast<const float4*>(ak + oa[p] - backa);
                    a[p] = oka ? t : zero4();
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const float4 t = *reinterpret_cast<const float4*>(bk + ow[p] - backb);
                    w[p] = okb ? t : zero4();
                }
            }
            return;
        }
        if (MODE == DGRAD) {
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                int64_t r = m0 + srow + 32 * p;
                r = r < M ? r : M - 1;
                const int64_t k = k0 + skq;
                a[p].x = k < kend ? A[r * lda + k] : 0.f; a[p].y = k + 1 < kend ? A[r * lda + k + 1] : 0.f;
                a[p].z = k + 2 < kend ? A[r * lda + k + 2] : 0.f; a[p].w = k + 3 < kend ? A[r * lda + k + 3] : 0.f;
            }
        } else {
            const int64_t k = k0 + kk;
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                const int64_t m = m0 + 4 * (c4 + 8 * p);
                if (k < kend) {
                    a[p].x = m < M ? A[k * lda + m] : 0.f; a[p].y = m + 1 < M ? A[k * lda + m + 1] : 0.f;
                    a[p].z = m + 2 < M ? A[k * lda + m + 2] : 0.f; a[p].w = m + 3 < M ? A[k * lda + m + 3] : 0.f;
                } else a[p] = zero4();
            }
        }
        {
            const int64_t k = k0 + kk;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int n = n0 + 4 * (c4 + 8 * p);
                if (k < kend) {
                    w[p].x = n < N ? B[k * ldb + n] : 0.f; w[p].y = n + 1 < N ? B[k * ldb + n + 1] : 0.f;
                    w[p].z = n + 2 < N ? B[k * ldb + n + 2] : 0.f; w[p].w = n + 3 < N ? B[k * ldb + n + 3] : 0.f;
                } else w[p] = zero4();
            }
        }
    };
    // WGRAD, optional: column sums of A over this block's batch slice (a dense layer's bias gradient), taken from the staging
    // registers by the blocks of the first column tile; block-uniform
    const bool do_cs = MODE == WGRAD && colsum != nullptr && n0 == 0;
    float4 cs[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) cs[p] = zero4();
    const bool wave_live = m0 + wm * (32 * TM) < M && n0 + wn * 32 < N;      // wave-uniform
    auto stage = [&](const float4 (&a)[AP], const float4 (&w)[2]) {
        if (MODE == DGRAD) {
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                const int m = srow + 32 * p;
                As[(skq + 0) * LDA + m] = a[p].x;
                As[(skq + 1) * LDA + m] = a[p].y;
                As[(skq + 2) * LDA + m] = a[p].z;
                As[(skq + 3) * LDA + m] = a[p].w;
            }
        } else {
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                const int m = 4 * ((tid & 7) + 8 * p);
                As[kk * LDA + m + 0] = a[p].x;
                As[kk * LDA + m + 1] = a[p].y;
                As[kk * LDA + m + 2] = a[p].z;
                As[kk * LDA + m + 3] = a[p].w;
                if (do_cs) { cs[p].x += a[p].x; cs[p].y += a[p].y; cs[p].z += a[p].z; cs[p].w += a[p].w; }
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int n = 4 * ((tid & 7) + 8 * p);
            Ws[kk * LDW + n + 0] = w[p].x;
            Ws[kk * LDW + n + 1] = w[p].y;
            Ws[kk * LDW + n + 2] = w[p].z;
            Ws[kk * LDW + n + 3] = w[p].w;
        }
    };
    auto mma = [&](int64_t k0) {
        if (!wave_live) return;                        // this wave's 64 x 32 sub-tile lies wholly past M or N (D = 320: the third 128-row tile of g_W
                                                       // has rows 320..383) -- it only helps staging
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && k0 + BK / 2 >= kend) break;   // upper half of the last slab all padding: skip its MFMAs
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
    };
    if (kbeg < kend) load_slab(kbeg, a, w);
    // (two slabs in flight instead of one: measured on the wgrad of narrow layers -- 31.6 vs 31.5 us at dim = 112, 135.0 vs 132.3 at 320 -- not kept)
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        stage(a, w);
        __syncthreads();
        if (k0 + BK < kend) load_slab(k0 + BK, a, w);
        mma(k0);
        __syncthreads();
    }

    if (do_cs) {                         // (the K loop ended on a barrier: the A slab area is free)
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int m = 4 * ((tid & 7) + 8 * p);
            const bool own = !VEC || m0 + m + 4 <= M;          // aligned path: columns past M were CLAMPED loads (copies of other columns)
            As[kk * LDA + m + 0] = own ? cs[p].x : 0.f;
            As[kk * LDA + m + 1] = own ? cs[p].y : 0.f;
            As[kk * LDA + m + 2] = own ? cs[p].z : 0.f;
            As[kk * LDA + m + 3] = own ? cs[p].w : 0.f;
        }
        __syncthreads();
        if (tid < BM && m0 + tid < M) {
            float t = 0.f;
#pragma unroll 8
            for (int k = 0; k < BK; ++k) t += As[k * LDA + tid];
            if (MODE == WGRAD && add2 != nullptr) colsum[(int64_t)ks * M + m0 + tid] = t;      // (ordered mode: per slice)
            else unsafeAtomicAdd(colsum + m0 + tid, t);
        }
    }
    gemm_epilogue<MODE, TM>(acc, m0, n0, wm, wn, l31, hi, M, N, addend, add_ld, maskT, mask_ld, out, out_ld,
                            MODE == WGRAD && add2 != nullptr ? add2 + (int64_t)ks * add2_ld : add2, add2_ld);
}

// ---- narrow layers (dim <= 112, the reference's own width): preparation + dgrad in ONE launch --------------------------------------------------
// At these widths every launch of the layer is a streaming pass -- K = dim is three or four slabs, the matrix work is nothing -- and the preparation
// kernel alone moved 197 MB (reads g, x0, lin, out, g_x0; writes glin, g_x0, the mask bits) before the dgrad read glin and g again: 43.7 + 31.0 us of
// the layer's 103.5 (profiles/r04_dcn_fp32_d112_rocprof_summary.txt).  Here a block owns a PANEL of 64 rows over ALL columns:
//   * W -- all of it, 50 KB, from L2 -- and the panel's elementwise inputs are requested up front (W first: loads return in order, so W is stored to
//     LDS while the panel's rows are still in flight);
//   * g_x0 and glin (the wgrad's operand) leave for memory, glin also stays in LDS K-major (the dgrad's A operand, the whole K), gm = g (x) [out > 0]
//     stays in registers;
//   * the dgrad of the panel runs from LDS with no barrier inside (one 32 x 32 accumulator per wavefront and 64-column tile of the output; same k
//     order as dcn2_gemm_kernel<DGRAD>: value for value its sums);
//   * gm goes through LDS (the dead A panel's place) into the accumulator layout: g_xl = gm + glin W (+ g_x0 for the stack's first layer);
//   * the first blocks clear g_W / g_b on the way (the wgrad launch behind adds to them: no fill launch); g_b comes from the wgrad's staging
//     registers (launch_wgrad's colsum).
// 80 KB of LDS: two blocks per compute unit.  232 MB of fabric traffic per launch (146 read + 89 written) against 294 for the two launches it
// replaces: 54.6 us against 74.7 (+ 4.7 of the fill launch); a 3-layer forward + backward step at B = 65 536 396 -> 343 us
// (profiles/r05_dcn_fp32_d112_rocprof_summary.txt).  Forms that lost on the way: W in K slabs inside the GEMM loop (a block's life was eight
// dependent L2 round trips and barriers: 81.7 us); g and out re-read in the epilogue (they came from memory, not from L2: 206 MB read, 56.7 us).
template <bool RELU>
__global__ __launch_bounds__(256, 2) void dcn2_bwd_panel_kernel(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ out,
                                                                const float* __restrict__ x0, const float* __restrict__ lin, int64_t ld,
                                                                int64_t M, int D, const float* __restrict__ W, float* __restrict__ glin, int64_t w_ld,
                                                                float* __restrict__ g_x0, int64_t gx0_ld, int acc_x0, int fold,
                                                                float* __restrict__ g_xl, int64_t gxl_ld, float* __restrict__ zero_a, int zero_na,
                                                                float* __restrict__ zero_b, int zero_nb) {
    constexpr int PM = 64, LDP = PM + 1, U = 4;
    // g_W and g_b start from zero (the wgrad launch behind this one adds to them): cleared here instead of by a launch of their own
    for (int i = blockIdx.x * 256 + threadIdx.x; i < zero_na + zero_nb; i += gridDim.x * 256) {
        if (i < zero_na) zero_a[i] = 0.f;
        else zero_b[i - zero_na] = 0.f;
    }
    extern __shared__ __attribute__((aligned(16))) float panel_smem[];
    const int Kp = (D + 15) & ~15;                    // the MFMA loop consumes K in halves of 16
    const int LDF = D + 4;                            // (a multiple of 4: rows of W and of gm are stored as 16-byte pieces)
    float* As = panel_smem;                           // [Kp][LDP]: glin of the panel, K-major
    const int a_floats = Kp * LDP > PM * LDF ? Kp * LDP : PM * LDF;      // (the panel's place later holds gm as [64][LDF])
    float* Wl = As + a_floats;                        // [Kp][LDF] (+ 64 floats of slack): ALL of W
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, hi = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * PM;
    for (int i = tid; i < (Kp - D) * LDP; i += 256) As[D * LDP + i] = 0.f;
    for (int i = tid; i < (Kp - D) * LDF + 64; i += 256) Wl[D * LDF + i] = 0.f;
    const int c4n = D >> 2;
    // W first (L2), the panel's rows behind it (memory): the loads return in order, so W is stored to LDS while the panel's are still in flight
    constexpr int NW = 13;                            // float4 of W per thread: dim <= 112 -> at most 3136 / 256
    float4 wv[NW];
    {
        const int n4 = D * c4n;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = u * 256 + tid;
            wv[u] = i < n4 ? reinterpret_cast<const float4*>(W)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // ---- the panel's elementwise part: thread = (row lane, 4 columns); U rows in flight, at most two rounds (a round covers >= 32 rows: dim <= 128);
    // gm = g (x) [out > 0] stays in registers for the epilogue
    constexpr int IT = 2;
    float4 gmk[IT][U];
    const int rpp = 256 / c4n;
    const int rsub = tid / c4n, c4 = tid - rsub * c4n, cc = 4 * c4;
    float4 gv[IT][U], xv[IT][U], lv[IT][U], ov[IT][U], ax[IT][U];
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rb = rsub + it * rpp * U;
            const int64_t r = m0 + rb + u * rpp;
            const int64_t rc = r < M ? r : M - 1;          // (rows past the batch: clamped loads, nothing stored)
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            gv[it][u] = xv[it][u] = lv[it][u] = ax[it][u] = z;
            ov[it][u] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (rsub < rpp && rb + u * rpp < PM) {         // (rows past the panel are another block's)
                gv[it][u] = *reinterpret_cast<const float4*>(g + rc * g_ld + cc);
                xv[it][u] = *reinterpret_cast<const float4*>(x0 + rc * ld + cc);
                lv[it][u] = *reinterpret_cast<const float4*>(lin + rc * ld + cc);
                if (RELU) ov[it][u] = *reinterpret_cast<const float4*>(out + rc * ld + cc);
                if (acc_x0) ax[it][u] = *reinterpret_cast<const float4*>(g_x0 + rc * gx0_ld + cc);
            }
        }
    {
        const int n4 = D * c4n;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = u * 256 + tid;
            if (i < n4) {
                const int k = i / c4n, n = 4 * (i - k * c4n);
                *reinterpret_cast<float4*>(Wl + k * LDF + n) = wv[u];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = rsub + it * rpp * U + u * rpp;
            float4 m_;
            m_.x = !RELU || ov[it][u].x > 0.f ? gv[it][u].x : 0.f; m_.y = !RELU || ov[it][u].y > 0.f ? gv[it][u].y : 0.f;
            m_.z = !RELU || ov[it][u].z > 0.f ? gv[it][u].z : 0.f; m_.w = !RELU || ov[it][u].w > 0.f ? gv[it][u].w : 0.f;
            gmk[it][u] = m_;
            if (rsub >= rpp || row >= PM) continue;
            const float4 gl = make_float4(m_.x * xv[it][u].x, m_.y * xv[it][u].y, m_.z * xv[it][u].z, m_.w * xv[it][u].w);
            const float4 gx = make_float4(fmaf(m_.x, lv[it][u].x, ax[it][u].x), fmaf(m_.y, lv[it][u].y, ax[it][u].y),
                                          fmaf(m_.z, lv[it][u].z, ax[it][u].z), fmaf(m_.w, lv[it][u].w, ax[it][u].w));
            As[(cc + 0) * LDP + row] = gl.x; As[(cc + 1) * LDP + row] = gl.y;
            As[(cc + 2) * LDP + row] = gl.z; As[(cc + 3) * LDP + row] = gl.w;
            if (m0 + row < M) {
                *reinterpret_cast<float4*>(glin + (m0 + row) * w_ld + cc) = gl;
                *reinterpret_cast<float4*>(g_x0 + (m0 + row) * gx0_ld + cc) = gx;
            }
        }
    __syncthreads();          // the panel is in LDS; this block's g_x0 rows are written (the fold below reads them back)
    // ---- dgrad of the panel, the one or two 64-column tiles of g_xl: both operands in LDS, no barrier inside
    const int nx = (D + BN - 1) / BN;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
        if (nt >= nx) continue;
        const float* wcol = Wl + nt * BN + wn * 32 + l31;      // (columns past dim read into the next row / the slack: their products are never stored)
        const float* arow = As + wm * 32 + l31;
        for (int k0 = 0; k0 < Kp; k0 += BK / 2) {
            float fb[BK / 4], fa[BK / 4];
#pragma unroll
            for (int i = 0; i < BK / 4; ++i) {
                const int kr = k0 + 2 * i + hi;
                fb[i] = wcol[kr * LDF];
                fa[i] = arow[kr * LDP];
            }
#pragma unroll
            for (int i = 0; i < BK / 4; ++i) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc[nt], 0, 0, 0);
        }
    }
    // ---- gm through LDS into the accumulator layout (the A panel is dead: its place, as [row][LDF])
    __syncthreads();
    float* Gs = As;
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = rsub + it * rpp * U + u * rpp;
            if (rsub < rpp && row < PM) *reinterpret_cast<float4*>(Gs + row * LDF + cc) = gmk[it][u];
        }
    __syncthreads();
    // epilogue (accumulator layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): g_xl = gm + acc (+ g_x0)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = nt * BN + wn * 32 + l31;
        if (nt >= nx || col >= D) continue;
        const int rl0 = wm * 32 + 4 * hi;
        float av[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + rl0 + (r & 3) + 8 * (r >> 2);
            av[r] = fold && row < M ? g_x0[row * gx0_ld + col] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = rl0 + (r & 3) + 8 * (r >> 2);
            const int64_t row = m0 + rl;
            if (row >= M) continue;
            float v = Gs[rl * LDF + col] + acc[nt][r];
            if (fold) v += av[r];
            g_xl[row * gxl_ld + col] = v;
        }
    }
}

// ---- the GEMM in split-bf16 math (flags bit 1; see dcn_v2_layer_bf16x3_kernel in nrx_dcn2.hip) ----------------------------------
// Same tiles, same epilogue; operands as bf16 hi / lo parts, [row][k] in LDS with 80-byte rows, three v_mfma_f32_32x32x16_bf16 per
// fragment pair.  A bf16 fragment is 8 CONSECUTIVE k of one row.  The M-major operand (dgrad's glin) has them contiguous in memory
// (float4 loads, as the forward's x).  A K-major operand (W in dgrad; glin and x_l in wgrad: a memory row is one k) does not: there a
// lane owns ONE output row / column and fetches its 8 k as 8 dword loads -- 64 lanes = 64 consecutive floats of each memory row,
// still coalesced -- splits them and writes its fragment with one ds_write_b128 per part (a transposing LDS write of 2-byte
// elements would be 8 conflicting ds_write_b16 per float4).
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __bf16 nrx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float nrx_f32x2 __attribute__((ext_vector_type(2)));
constexpr int LDH = 40;      // halfs per LDS row (32 k + 16 bytes of padding: conflict-free ds_read_b128 / ds_write_b128)

__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const nrx_f32x2 v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, nrx_bf16x2));
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    const nrx_f32x2 r = {a - ha, b - hb};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, nrx_bf16x2));
}

// the K-major operand's staging: `pairs` (row, k-octet) fragments per thread; fragment i of thread tid: row = (tid + 256 i) % ROWS,
// octet = (tid + 256 i) / ROWS.  src(k, row) = base[k * ld + row0 + row]; rows past `nrows` are clamped, k past kend zeroed.
template <int ROWS>
struct KMajorStage {
    static constexpr int PAIRS = ROWS * 4 / 256;          // 32 k = 4 octets per row
    float v[PAIRS][8];
    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, int64_t kend, int tid) {
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int f = tid + 256 * i, row = f % ROWS, oct = f / ROWS;
            int64_t r = row0 + row;
            r = r < nrows ? r : nrows - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t k = k0 + 8 * oct + j;
                const float t = base[(k < kend ? k : kend - 1) * ld + r];
                v[i][j] = k < kend ? t : 0.f;
            }
        }
    }
    // Full slab (k0 + 32 <= kend): no selects and no per-lane 64-bit address arithmetic -- `slab` = &base[k0 * ld + row0] is wave-uniform
    // (advanced on the scalar unit, + j rows of the source per load), `off[i]` a fixed 32-bit byte offset per lane (prepare()).  The
    // guarded form above spent ~12 vector instructions per load: 34 per MFMA in the wgrad launch, which was bound by them
    // (profiles/r03_dcn_v2_bf16x3.txt).
    __device__ __forceinline__ void prepare(uint32_t (&off)[PAIRS], int64_t ld, int64_t row0, int64_t nrows, int tid) const {
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int f = tid + 256 * i, row = f % ROWS, oct = f / ROWS;
            const int64_t r = row0 + row < nrows ? row : nrows - 1 - row0;            // clamped like load(): the product lands in rows never stored
            off[i] = (uint32_t)(((int64_t)(8 * oct) * ld + r) * 4);
        }
    }
    __device__ __forceinline__ void load_full(const char* __restrict__ slab, size_t ld_bytes, const uint32_t (&off)[PAIRS]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const char* rowj = slab + j * ld_bytes;
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) v[i][j] = *reinterpret_cast<const float*>(rowj + off[i]);
        }
    }
    __device__ __forceinline__ void store(unsigned short* __restrict__ sh, unsigned short* __restrict__ sl, int tid) const {
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int f = tid + 256 * i, row = f % ROWS, oct = f / ROWS;
            uint4 h, l;
            split2(v[i][0], v[i][1], h.x, l.x);
            split2(v[i][2], v[i][3], h.y, l.y);
            split2(v[i][4], v[i][5], h.z, l.z);
            split2(v[i][6], v[i][7], h.w, l.w);
            *reinterpret_cast<uint4*>(&sh[row * LDH + 8 * oct]) = h;
            *reinterpret_cast<uint4*>(&sl[row * LDH + 8 * oct]) = l;
        }
    }
};

template <int MODE, int TMv = 2>
__global__ __launch_bounds__(256, 4) void dcn2_gemm_split_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                               int64_t M, int N, int64_t K, int64_t kslice, const float* __restrict__ addend,
                                                               int64_t add_ld, const uint32_t* __restrict__ maskT, int64_t mask_ld,
                                                               float* __restrict__ out, int64_t out_ld, unsigned nx, unsigned ntiles,
                                                               const float* __restrict__ add2, int64_t add2_ld) {
    constexpr int TM = TMv, BM = 2 * TM * 32;
    __shared__ __attribute__((aligned(16))) unsigned short Ah[BM * LDH], Al[BM * LDH], Wh[BN * LDH], Wl[BN * LDH];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    unsigned logical, ks;
    gemm_block_to_tile<MODE, true>(blockIdx.x, ntiles, logical, ks);
    if ((int64_t)ks * kslice >= K) return;           // WGRAD: the slice count is rounded up to whole XCD rounds (block-uniform exit)
    const int64_t m0 = (int64_t)(logical / nx) * BM;
    const int n0 = (int)(logical % nx) * BN;
    const int64_t kbeg = (int64_t)ks * kslice;
    const int64_t kend = kbeg + kslice < K ? kbeg + kslice : K;

    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // staging registers: DGRAD A is M-major (float4 along k, as the forward's x); everything else K-major
    constexpr int AP = BM / 32;
    float4 a4[MODE == DGRAD ? AP : 1];
    KMajorStage<BM> ak;
    KMajorStage<BN> bk;
    const int srow = tid >> 3, skq = (tid & 7) * 4;
    uint32_t oa4[MODE == DGRAD ? AP : 1], oak[KMajorStage<BM>::PAIRS], obk[KMajorStage<BN>::PAIRS];
    if (MODE == DGRAD) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int64_t r = m0 + srow + 32 * p < M ? srow + 32 * p : M - 1 - m0;
            oa4[p] = (uint32_t)((r * lda + skq) * 4);
        }
    } else {
        ak.prepare(oak, lda, m0, M, tid);
    }
    bk.prepare(obk, ldb, n0, N, tid);
    const char* const a_tile = reinterpret_cast<const char*>(MODE == DGRAD ? A + m0 * lda : A + m0);
    const char* const b_tile = reinterpret_cast<const char*>(B + n0);
    auto load_slab = [&](int64_t k0) {
        if (k0 + BK <= kend) {            // full slab (block-uniform): wave-uniform bases + fixed per-lane offsets
            if (MODE == DGRAD) {
                const char* ak0 = a_tile + (size_t)k0 * 4;
#pragma unroll
                for (int p = 0; p < AP; ++p) a4[p] = *reinterpret_cast<const float4*>(ak0 + oa4[p]);
            } else {
                ak.load_full(a_tile + (size_t)k0 * (size_t)lda * 4, (size_t)lda * 4, oak);
            }
            bk.load_full(b_tile + (size_t)k0 * (size_t)ldb * 4, (size_t)ldb * 4, obk);
            return;
        }
        if (MODE == DGRAD) {
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                int64_t r = m0 + srow + 32 * p;
                r = r < M ? r : M - 1;
                const int64_t k = k0 + skq;                      // K % 4 == 0 (aligned shapes only): a float4 is all-in or all-out
                const float4 t = *reinterpret_cast<const float4*>(A + r * lda + (k + 4 <= kend ? k : kbeg));
                a4[p] = k + 4 <= kend ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            ak.load(A, lda, m0, M, k0, kend, tid);
        }
        bk.load(B, ldb, n0, N, k0, kend, tid);
    };
    if (kbeg < kend) load_slab(kbeg);
    const bool wave_live = m0 + wm * (32 * TM) < M && n0 + wn * 32 < N;

    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        if (MODE == DGRAD) {
#pragma unroll
            for (int p = 0; p < AP; ++p) {
                uint2 h, l;
                split2(a4[p].x, a4[p].y, h.x, l.x);
                split2(a4[p].z, a4[p].w, h.y, l.y);
                *reinterpret_cast<uint2*>(&Ah[(srow + 32 * p) * LDH + skq]) = h;
                *reinterpret_cast<uint2*>(&Al[(srow + 32 * p) * LDH + skq]) = l;
            }
        } else {
            ak.store(Ah, Al, tid);
        }
        bk.store(Wh, Wl, tid);
        __syncthreads();
        if (k0 + BK < kend) load_slab(k0 + BK);
        if (wave_live) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (s2 == 1 && k0 + 16 >= kend) break;
                const int ko = s2 * 16 + 8 * hi;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Wh[(wn * 32 + l31) * LDH + ko]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Wl[(wn * 32 + l31) * LDH + ko]);
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    const int row = wm * (32 * TM) + 32 * t + l31;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ah[row * LDH + ko]);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Al[row * LDH + ko]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    gemm_epilogue<MODE, TM>(acc, m0, n0, wm, wn, l31, hi, M, N, addend, add_ld, maskT, mask_ld, out, out_ld, add2, add2_ld);
}

// Ordered mode of the wgrad (bit-reproducible g_W / g_b): the blocks store their batch slice's partial tile (and column sums) instead of adding them with
// float atomics; this kernel sums the slices IN SLICE ORDER.  64 outputs x 16 slice groups per block: a thread sums every 16th slice of its output (a
// fixed association: the same bits run to run), the groups are added in order through LDS.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, int slices, int64_t mn, float* __restrict__ out,
                                                             const float* __restrict__ cs_partial, int m, float* __restrict__ colsum) {
    __shared__ float s_p[16][64];
    const int li = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + li, tot = mn + (cs_partial != nullptr ? m : 0);
    float a0 = 0.f, a1 = 0.f;
    if (i < tot) {
        const float* p = i < mn ? partial + i : cs_partial + (i - mn);
        const int64_t stride = i < mn ? mn : m;
        int s = c;
        for (; s + 16 < slices; s += 32) { a0 += p[(int64_t)s * stride]; a1 += p[(int64_t)(s + 16) * stride]; }
        if (s < slices) a0 += p[(int64_t)s * stride];
    }
    s_p[c][li] = a0 + a1;
    __syncthreads();
    if (c == 0 && i < tot) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += s_p[k][li];
        if (i < mn) out[i] = v;
        else colsum[i - mn] = v;
    }
}

// The launch shape of the wgrad: tile height, tiles, batch slices, rows per slice (see launch_wgrad)
struct WgradShape { bool small; unsigned nx, nt; int64_t splits, kslice; bool xcd; };
WgradShape wgrad_shape(int M, int N, int64_t batch, bool use_split) {
    static const int env_tile = getenv("NRX_WGRAD_TILE") ? atoi(getenv("NRX_WGRAD_TILE")) : 0;            // tuning: 64 | 128
    static const int env_blocks = getenv("NRX_WGRAD_BLOCKS") ? atoi(getenv("NRX_WGRAD_BLOCKS")) : 0;      // tuning: target block count
    static const int env_rows = getenv("NRX_WGRAD_MIN_ROWS") ? atoi(getenv("NRX_WGRAD_MIN_ROWS")) : 0;    // tuning: shortest batch slice
    WgradShape w;
    w.small = env_tile ? env_tile == 64 : M <= 384;
    const int bm = w.small ? 64 : BM;
    w.nx = (unsigned)((N + BN - 1) / BN);
    w.nt = w.nx * (unsigned)((M + bm - 1) / bm);
    const unsigned nt = w.nt;
    int64_t splits = ((env_blocks ? env_blocks : w.small ? 1536 : 1024) + nt - 1) / nt;
    int64_t kslice = ((batch + splits - 1) / splits + BK - 1) / BK * BK;
    const int64_t kmin = env_rows ? (env_rows + BK - 1) / BK * BK : (w.small && nt > 1 ? 16 : 8) * BK;
    if (kslice < kmin) kslice = kmin;
    splits = (batch + kslice - 1) / kslice;
    // Slices per XCD (gemm_block_to_tile) with the slice count chosen so that every XCD's 32 x 4 block slots hold WHOLE slices in whole rounds:
    // D = 320 (25 tiles of 64 x 64): 5 slices per XCD and round, 2 rounds -> 80 slices of 832 rows, 2000 blocks.  With the round-2 slice
    // count (61) the same order left the XCDs unevenly loaded and the fp32 launch got slower (133 -> 155 us); balanced, it is faster for
    // both kernels (same box, forward + backward per step: fp32 495.6 -> 491.5 us, split-bf16 343.6 -> 335.4, D = 512 1061 -> 1055).
    // NRX_WGRAD_XCD=0 restores the round-2 order for the fp32 kernel and the unbalanced count.
    static const int xcd_env = getenv("NRX_WGRAD_XCD") ? atoi(getenv("NRX_WGRAD_XCD")) : 1;
    w.xcd = xcd_env == 1 && !use_split;
    if (xcd_env == 1) {
        const int64_t per_round = 128 / nt > 0 ? 128 / nt : 1;                 // slices an XCD's 32 x 4 block slots hold side by side
        const int64_t want = env_blocks ? env_blocks : w.small ? 1536 : 1024;
        int64_t rounds = (want + 4 * per_round * nt) / (8 * per_round * nt);
        if (rounds < 1) rounds = 1;
        splits = 8 * per_round * rounds;
        kslice = ((batch + splits - 1) / splits + BK - 1) / BK * BK;
        if (kslice < kmin) kslice = kmin;
        splits = (batch + kslice - 1) / kslice;
    }
    w.splits = splits;
    w.kslice = kslice;
    return w;
}
// scratch of the ordered mode: one [M, N] tile set and M column sums per batch slice
size_t wgrad_ordered_bytes(int M, int N, int64_t batch) {
    if (batch <= 0) return 0;
    const WgradShape w = wgrad_shape(M, N, batch, false);
    return (size_t)w.splits * ((size_t)M * N + (size_t)M) * sizeof(float) + 256;
}

// out[M, N] += A^T B over the batch (A [batch, M], B [batch, N], both K-major), split into batch slices; out pre-zeroed.
// Tile and slice choice (tools/dcn2_dgrad_probe.hip, B = 65 536): up to M = 384 the 64 x 64 block tile wins (M = 320 is 5 whole
// tiles instead of 2.5; narrow layers get twice the tiles to spread over the CUs: D = 112 38 -> 30 us, D = 320 135 -> 123 us), above
// it the 128 x 64 tile (less LDS traffic per MFMA: D = 512 280 vs 298 us).  ~1 536 (64 x 64) / ~1 024 (128 x 64) blocks, but never
// slices shorter than 512 / 256 batch rows (256 for a single tile): with few tiles the atomics of a short slice cost more than the
// blocks it adds.  NRX_WGRAD_TILE / NRX_WGRAD_BLOCKS / NRX_WGRAD_MIN_ROWS override the choice (tools/run_wgrad_sweep.sh).
// ordered (a scratch of wgrad_ordered_bytes): no atomics -- per-slice partials + wgrad_reduce_kernel; out / colsum need no zero fill then.
void launch_wgrad(const float* A, int64_t lda, const float* B, int64_t ldb, int M, int N, int64_t batch, float* out, float* colsum, bool vec, hipStream_t st,
                  bool split = false, float* ordered = nullptr) {
    const bool use_split = split && vec && colsum == nullptr && ordered == nullptr;
    const WgradShape w = wgrad_shape(M, N, batch, use_split);
    const bool small = w.small;
    const unsigned nx = w.nx, nt = w.nt;
    const int64_t splits = w.splits, kslice = w.kslice;
    const bool xcd_fp32 = w.xcd;
    const float* part = ordered;
    float* cs_part = ordered != nullptr && colsum != nullptr ? ordered + (size_t)splits * M * N : colsum;
    const dim3 grid((unsigned)(nt * ((use_split || xcd_fp32) ? (splits + 7) / 8 * 8 : splits)));     // slices per XCD: whole XCD rounds of slices (gemm_block_to_tile); surplus blocks leave at once
#define NRX_WGRAD(VEC_, TM_)                                                                                                          \
    hipLaunchKernelGGL((dcn2_gemm_kernel<WGRAD, VEC_, TM_>), grid, dim3(256), 0, st, A, lda, B, ldb, (int64_t)M, N, batch, kslice,     \
                       (const float*)nullptr, (int64_t)0, (const uint32_t*)nullptr, (int64_t)0, out, (int64_t)N, nx, nt,              \
                       part, (int64_t)M * N, cs_part, xcd_fp32 ? 1 : 0)
    if (use_split) {
        if (small) hipLaunchKernelGGL((dcn2_gemm_split_kernel<WGRAD, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, (int64_t)M, N, batch, kslice,
                                      (const float*)nullptr, (int64_t)0, (const uint32_t*)nullptr, (int64_t)0, out, (int64_t)N, nx, nt, (const float*)nullptr, (int64_t)0);
        else hipLaunchKernelGGL((dcn2_gemm_split_kernel<WGRAD, 2>), grid, dim3(256), 0, st, A, lda, B, ldb, (int64_t)M, N, batch, kslice,
                                (const float*)nullptr, (int64_t)0, (const uint32_t*)nullptr, (int64_t)0, out, (int64_t)N, nx, nt, (const float*)nullptr, (int64_t)0);
    } else
    if (small) { if (vec) NRX_WGRAD(true, 1); else NRX_WGRAD(false, 1); }
    else       { if (vec) NRX_WGRAD(true, 2); else NRX_WGRAD(false, 2); }
#undef NRX_WGRAD
    if (ordered != nullptr) {
        const int64_t tot = (int64_t)M * N + (colsum != nullptr ? M : 0);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(1024), 0, st, part, (int)splits, (int64_t)M * N, out,
                           colsum != nullptr ? (const float*)cs_part : (const float*)nullptr, M, colsum);
    }
}

}  // namespace

extern "C" int64_t nrx_dcn_v2_layer_bwd_workspace(int64_t batch, int32_t dim) {
    if (batch < 0 || dim < 1) return -1;
    const int64_t ld = (dim + 3) & ~3;
    return (batch + (batch + 31) / 32) * ld * (int64_t)sizeof(float) + 512 +         // glin [batch, ld] + the ReLU mask bits [ceil(batch / 32), ld]
           (int64_t)wgrad_ordered_bytes(dim, dim, batch) + 256;                      // + the per-slice partial tiles of the ordered wgrad (flags bit 2)
}

extern "C" int nrx_dcn_v2_layer_bwd(const float* x0, const float* xl, int64_t ld, const float* lin, const float* out, int32_t relu,
                                    int64_t batch, int32_t dim, const float* W, const float* g_out, int64_t g_ld,
                                    float* g_xl, int64_t gxl_ld, float* g_x0, int64_t gx0_ld, int32_t accumulate_x0,
                                    float* g_W, float* g_b, void* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(x0 && xl && lin && W && g_out && g_xl && g_x0 && g_W && g_b && workspace && batch >= 0 && dim >= 1 && ld >= dim,
                "nrx_dcn_v2_layer_bwd: bad argument");
    const bool split = (relu & 2) != 0;        // flags as in nrx_dcn_v2_layer_fwd: bit 0 = ReLU, bit 1 = split-bf16 matrix math
    // bit 2 (backward only): g_W / g_b summed over the batch slices in a fixed order (no float atomics).  Layers up to 128 wide take that mode
    // whatever the bit says: there it costs nothing (dim = 64: GEMM 19.8 -> 12.7 us + 4.7 for the reduction launch; 112: 31.8 -> 28.1 + 5.0; at
    // dim = 320 the 33 MB of partial tiles make it 12 us slower per layer) and the layer's whole backward is bit-reproducible.
    // NRX_DCN2_NARROW_ORDERED=0: atomics unless asked.
    static const bool narrow_ordered = !(getenv("NRX_DCN2_NARROW_ORDERED") && atoi(getenv("NRX_DCN2_NARROW_ORDERED")) == 0);
    const bool ordered = (relu & 4) != 0 || (narrow_ordered && dim <= 128);
    relu &= 1;
    NRX_REQUIRE(!relu || out != nullptr, "nrx_dcn_v2_layer_bwd: the ReLU mask needs the layer's forward output");
    NRX_REQUIRE(g_ld >= dim && gxl_ld >= dim && gx0_ld >= dim, "nrx_dcn_v2_layer_bwd: bad leading dimension");
    NRX_REQUIRE(g_xl != g_out, "nrx_dcn_v2_layer_bwd: g_xl must not alias g_out");
    NRX_REQUIRE(dim <= 1024, "nrx_dcn_v2_layer_bwd: dim %d > 1024 unsupported", dim);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int acc_x0 = accumulate_x0 & 1;                          // bit 0: add to g_x0 instead of overwriting it
    const float* fold = (accumulate_x0 & 2) ? g_x0 : nullptr;      // bit 1: fold the (written or accumulated) g_x0 into g_xl (the stack's first layer)
    const int64_t wld = (dim + 3) & ~3;
    const bool vec = (dim & 3) == 0 && (ld & 3) == 0 && (g_ld & 3) == 0 && (gx0_ld & 3) == 0 && (gxl_ld & 3) == 0 && nrx_aligned16(x0) &&
                     nrx_aligned16(xl) && nrx_aligned16(lin) && nrx_aligned16(g_out) && nrx_aligned16(g_x0) && nrx_aligned16(g_xl) &&
                     nrx_aligned16(W) && (out == nullptr || nrx_aligned16(out));
    // narrow layers, fp32 math, aligned operands: preparation + dgrad as one launch (dcn2_bwd_panel_kernel), which also clears g_W / g_b;
    // NRX_DCN2_PANEL=0: the three-launch path
    // (two blocks per compute unit need <= 80 KB of LDS each: dim <= 112; at dim = 128 -- 100 KB, one block -- the three launches are faster: 415 vs 426 us
    // per 3-layer step)
    static const bool panel_on = !(getenv("NRX_DCN2_PANEL") && atoi(getenv("NRX_DCN2_PANEL")) == 0);
    const size_t kp_ = (size_t)((dim + 15) & ~15);
    const size_t lds = ((kp_ * 65 > (size_t)64 * (dim + 4) ? kp_ * 65 : (size_t)64 * (dim + 4)) + kp_ * (dim + 4) + 64) * sizeof(float);
    const bool panel = panel_on && vec && !split && dim >= 8 && dim <= 112 && lds <= 80 * 1024 && batch > 0;
    if (!panel && (!ordered || batch == 0)) {          // (the ordered wgrad writes every element of g_W / g_b: no fill)
        if (g_b == g_W + (size_t)dim * dim) {          // g_b right behind g_W (what the Python layer allocates): one fill launch
            if (nrx_zero_async(g_W, sizeof(float) * ((size_t)dim * dim + dim), st) != NRX_OK) return NRX_ERR_LAUNCH;
        } else if (nrx_zero_async(g_W, sizeof(float) * (size_t)dim * dim, st) != NRX_OK || nrx_zero_async(g_b, sizeof(float) * (size_t)dim, st) != NRX_OK)
            return NRX_ERR_LAUNCH;
    }
    if (batch == 0) return NRX_OK;
    float* glin = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    uint32_t* maskT = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(glin + batch * wld) + 255) & ~(uintptr_t)255);
    float* part = ordered ? reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(maskT + ((batch + 31) / 32) * wld) + 255) & ~(uintptr_t)255) : nullptr;
    if (panel) {
        static const bool attr = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn2_bwd_panel_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn2_bwd_panel_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) == hipSuccess;
        }();
        (void)attr;
        const unsigned blocks = (unsigned)((batch + 63) / 64);
        if (relu) hipLaunchKernelGGL((dcn2_bwd_panel_kernel<true>), dim3(blocks), dim3(256), lds, st, g_out, g_ld, out, x0, lin, ld, batch, (int)dim, W,
                                     glin, wld, g_x0, gx0_ld, acc_x0, fold != nullptr ? 1 : 0, g_xl, gxl_ld, g_W, (int)(dim * dim), g_b, (int)dim);
        else hipLaunchKernelGGL((dcn2_bwd_panel_kernel<false>), dim3(blocks), dim3(256), lds, st, g_out, g_ld, out, x0, lin, ld, batch, (int)dim, W,
                                glin, wld, g_x0, gx0_ld, acc_x0, fold != nullptr ? 1 : 0, g_xl, gxl_ld, g_W, (int)(dim * dim), g_b, (int)dim);
        launch_wgrad(glin, wld, xl, ld, dim, dim, batch, g_W, g_b, vec, st, false, part);       // (g_b: column sums of glin from the wgrad's staging registers)
        NRX_LAUNCH_CHECK("nrx_dcn_v2_layer_bwd");
        return NRX_OK;
    }
    {
        int tl = 2;                                   // threads per row = 2^tl >= dim / 4 (<= 256)
        while ((4 << tl) < dim && tl < 8) ++tl;
        const int rpb = NRX_BLOCK >> tl;                 // 32-row groups per block and sweep
        int64_t grid = ((batch + 31) / 32 + rpb - 1) / rpb;
        if (grid > 512) grid = 512;
#define NRX_PREP(TL_)                                                                                                              \
    case TL_:                                                                                                                      \
        if (vec) hipLaunchKernelGGL((dcn_v2_bwd_prep_kernel<TL_, true>), dim3((unsigned)grid), dim3(NRX_BLOCK), 0, st, g_out, g_ld, out,  \
                                    x0, lin, ld, batch, dim, relu, glin, wld, g_x0, gx0_ld, acc_x0, ordered ? (float*)nullptr : g_b, maskT);           \
        else hipLaunchKernelGGL((dcn_v2_bwd_prep_kernel<TL_, false>), dim3((unsigned)grid), dim3(NRX_BLOCK), 0, st, g_out, g_ld, out,    \
                                x0, lin, ld, batch, dim, relu, glin, wld, g_x0, gx0_ld, acc_x0, ordered ? (float*)nullptr : g_b, maskT);               \
        break;
        switch (tl) { NRX_PREP(2) NRX_PREP(3) NRX_PREP(4) NRX_PREP(5) NRX_PREP(6) NRX_PREP(7) default: NRX_PREP(8) }
#undef NRX_PREP
    }
    const uint32_t* mask = relu ? maskT : nullptr;
    {   // dgrad: g_xl = gm + glin W        (M = batch, N = K = dim; B operand = W rows, K-major)
        const unsigned nx = (unsigned)((dim + BN - 1) / BN);
        const int64_t nt = (int64_t)nx * ((batch + BM - 1) / BM);
        NRX_REQUIRE(nt <= 0x7fffffffLL, "nrx_dcn_v2_layer_bwd: batch too large for one launch");
        if (vec && split && batch >= 8)
            hipLaunchKernelGGL((dcn2_gemm_split_kernel<DGRAD>), dim3((unsigned)nt), dim3(256), 0, st, glin, wld, W, (int64_t)dim, batch,
                               dim, (int64_t)dim, (int64_t)dim, g_out, g_ld, mask, wld, g_xl, gxl_ld, nx, (unsigned)nt, fold, gx0_ld);
        else if (vec) hipLaunchKernelGGL((dcn2_gemm_kernel<DGRAD, true>), dim3((unsigned)nt), dim3(256), 0, st, glin, wld, W, (int64_t)dim, batch,
                                    dim, (int64_t)dim, (int64_t)dim, g_out, g_ld, mask, wld, g_xl, gxl_ld, nx, (unsigned)nt, fold, gx0_ld, (float*)nullptr);
        else hipLaunchKernelGGL((dcn2_gemm_kernel<DGRAD, false>), dim3((unsigned)nt), dim3(256), 0, st, glin, wld, W, (int64_t)dim, batch,
                                dim, (int64_t)dim, (int64_t)dim, g_out, g_ld, mask, wld, g_xl, gxl_ld, nx, (unsigned)nt, fold, gx0_ld, (float*)nullptr);
    }
    // wgrad: g_W[i, j] += sum_b glin[b, i] xl[b, j]   (M = N = dim, K = batch, split over the batch)
    if (ordered) launch_wgrad(glin, wld, xl, ld, dim, dim, batch, g_W, g_b, vec, st, false, part);      // (fp32 matrix math whatever bit 1 says; g_b from its staging registers)
    else launch_wgrad(glin, wld, xl, ld, dim, dim, batch, g_W, nullptr, vec, st, split && batch >= 8);
    NRX_LAUNCH_CHECK("nrx_dcn_v2_layer_bwd");
    return NRX_OK;
}

extern "C" int64_t nrx_linear_wgrad_ordered_workspace(int64_t batch, int32_t out_features, int32_t in_features) {
    if (batch < 0 || out_features < 1 || in_features < 1) return -1;
    return (int64_t)wgrad_ordered_bytes(out_features, in_features, batch) + 256;
}

extern "C" int nrx_linear_wgrad_ordered(const float* g, int64_t g_ld, const float* a, int64_t a_ld, int64_t batch, int32_t out_features,
                                        int32_t in_features, float* g_W, float* g_b, void* workspace, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(g && a && g_W && workspace && batch >= 0 && out_features >= 1 && in_features >= 1 && g_ld >= out_features && a_ld >= in_features,
                "nrx_linear_wgrad_ordered: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0) {
        if (nrx_zero_async(g_W, sizeof(float) * (size_t)out_features * in_features, st) != NRX_OK) return NRX_ERR_LAUNCH;
        if (g_b != nullptr && nrx_zero_async(g_b, sizeof(float) * (size_t)out_features, st) != NRX_OK) return NRX_ERR_LAUNCH;
        return NRX_OK;
    }
    const bool vec = (g_ld & 3) == 0 && (a_ld & 3) == 0 && (out_features & 3) == 0 && (in_features & 3) == 0 && nrx_aligned16(g) && nrx_aligned16(a);
    float* part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    launch_wgrad(g, g_ld, a, a_ld, out_features, in_features, batch, g_W, g_b, vec, st, false, part);
    NRX_LAUNCH_CHECK("nrx_linear_wgrad_ordered");
    return NRX_OK;
}

extern "C" int nrx_linear_wgrad(const float* g, int64_t g_ld, const float* a, int64_t a_ld, int64_t batch, int32_t out_features,
                                int32_t in_features, float* g_W, float* g_b, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(g && a && g_W && batch >= 0 && out_features >= 1 && in_features >= 1 && g_ld >= out_features && a_ld >= in_features,
                "nrx_linear_wgrad: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (nrx_zero_async(g_W, sizeof(float) * (size_t)out_features * in_features, st) != NRX_OK) return NRX_ERR_LAUNCH;
    if (g_b != nullptr && nrx_zero_async(g_b, sizeof(float) * (size_t)out_features, st) != NRX_OK) return NRX_ERR_LAUNCH;
    if (batch == 0) return NRX_OK;
    const bool vec = (g_ld & 3) == 0 && (a_ld & 3) == 0 && (out_features & 3) == 0 && (in_features & 3) == 0 && nrx_aligned16(g) && nrx_aligned16(a);
    launch_wgrad(g, g_ld, a, a_ld, out_features, in_features, batch, g_W, g_b, vec, st);
    NRX_LAUNCH_CHECK("nrx_linear_wgrad");
    return NRX_OK;
}
