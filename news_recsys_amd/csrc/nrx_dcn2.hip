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
#include <type_traits>
#include <cstdlib>
#include <cstring>

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

template <bool RELU, bool VEC, bool KEEPX = false>
__global__ __launch_bounds__(256, KEEPX ? 4 : 5) void dcn_v2_layer_kernel(const float* __restrict__ x0, const float* __restrict__ xl, int64_t ld,
                                                           int64_t M, int N, const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ out, int64_t out_ld, unsigned nx,
                                                           float* __restrict__ lin_out) {
    constexpr int XLD = 68;                              // KEEPX: floats per row of the x_l tile handed to the epilogue through LDS
    __shared__ __attribute__((aligned(16))) float s_all[KEEPX ? BM * XLD : BK * LDA + BK * LDW];
    static_assert(BM * XLD >= BK * LDA + BK * LDW, "operand slabs");
    float* const As = s_all;
    float* const Ws = s_all + BK * LDA;
    const int K = N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform by construction; tell the compiler
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
    // VALU instructions do not overlap the fp32 MFMA on gfx950 (profiles/r01_mfma_f32_valu_overlap_probe.txt:
    // each one costs its 4 cycles of matrix time), so the aligned path keeps them out of the k loop: the
    // tile base is a wave-uniform pointer advanced on the scalar unit, each thread adds a fixed 32-bit byte
    // offset (global_load saddr + voffset form), rows past M / N are clamped instead of zeroed (their
    // outputs are never stored) and only a partial last slab pays for zero selects.
    constexpr int AP = BM / 32;       // float4 loads of the A slab per thread
    float4 a[AP], w[2];
    uint32_t oa[AP], ow[2];
    const char* const xtile = reinterpret_cast<const char*>(xl + m0 * ld);
    const char* const wtile = reinterpret_cast<const char*>(W + (int64_t)n0 * K);
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int64_t r = m0 + srow + 32 * p < M ? srow + 32 * p : M - 1 - m0;
        oa[p] = (uint32_t)((r * ld + skq) * 4);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = n0 + srow + 32 * p < N ? srow + 32 * p : N - 1 - n0;
        ow[p] = (uint32_t)((r * K + skq) * 4);
    }
    auto load_slab = [&](int k0) {
        if (VEC) {
            const char* xk = xtile + (size_t)k0 * 4;
            const char* wk = wtile + (size_t)k0 * 4;
            if (k0 + BK <= K) {
#pragma unroll
                for (int p = 0; p < AP; ++p) a[p] = *reinterpret_cast<const float4*>(xk + oa[p]);
#pragma unroll
                for (int p = 0; p < 2; ++p) w[p] = *reinterpret_cast<const float4*>(wk + ow[p]);
            } else {                    // partial last slab: K % 4 == 0, so a float4 is all-in or all-out;
                const bool ok = k0 + skq < K;                  // out-of-range lanes read column 0 of their row and zero it (k0 + skq of
                const size_t back = ok ? 0 : (size_t)(k0 + skq) * 4;          // slab 0 lies past the row when K < 32: an out-of-bounds read at the tensor's end)
#pragma unroll
                for (int p = 0; p < AP; ++p) {
                    const float4 t = *reinterpret_cast<const float4*>(xk + oa[p] - back);
                    a[p] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const float4 t = *reinterpret_cast<const float4*>(wk + ow[p] - back);
                    w[p] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < AP; ++p) a[p] = guarded_load4(xl, m0 + srow + 32 * p, M, ld, k0 + skq, K, false);
#pragma unroll
            for (int p = 0; p < 2; ++p) w[p] = guarded_load4(W, n0 + srow + 32 * p, N, K, k0 + skq, K, false);
        }
    };
    load_slab(0);
    // KEEPX: the epilogue's x_l tile is two of the A slabs this block stages (k0 = n0, n0 + 32); their staging registers are kept and
    // handed over through LDS instead of a second fetch ten slabs later (see dcn_v2_layer_bf16x3_kernel)
    float4 xkeep[KEEPX ? 2 : 1][AP];
    if (KEEPX) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int p = 0; p < AP; ++p) xkeep[s2][p] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    for (int k0 = 0; k0 < K; k0 += BK) {
        if (KEEPX && k0 == n0) {
#pragma unroll
            for (int p = 0; p < AP; ++p) { xkeep[0][p].x = a[p].x; xkeep[0][p].y = a[p].y; xkeep[0][p].z = a[p].z; xkeep[0][p].w = a[p].w; }
        }
        if (KEEPX && k0 == n0 + BK) {
#pragma unroll
            for (int p = 0; p < AP; ++p) { xkeep[KEEPX ? 1 : 0][p].x = a[p].x; xkeep[KEEPX ? 1 : 0][p].y = a[p].y; xkeep[KEEPX ? 1 : 0][p].z = a[p].z; xkeep[KEEPX ? 1 : 0][p].w = a[p].w; }
        }
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
        if (k0 + BK < K) load_slab(k0 + BK);      // next slab: loads stay in flight across the MFMA block below
        // fragments of half a slab (8 k-pairs: 8 B + 32 A dwords) are read ahead of a dense block of
        // 32 MFMAs, so the LDS latency is paid twice per slab instead of once per MFMA group
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && k0 + BK / 2 >= K) break;      // the last slab's upper half is all padding (K % 32 <= 16): skip its MFMAs
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

    // Epilogue, also written for few VALU instructions: the address of register r's element is a wave-uniform
    // row pointer (tile base + constant * ld, scalar unit) plus one fixed per-lane byte offset, so a full
    // tile costs add-bias, fma, max per element; only a tile that crosses M takes the guarded path.
    if (KEEPX) {                     // (the K loop ended on a barrier: the slabs are free)
#pragma unroll
        for (int p = 0; p < AP; ++p)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) *reinterpret_cast<float4*>(&s_all[(srow + 32 * p) * XLD + 32 * s2 + skq]) = xkeep[KEEPX ? s2 : 0][p];
        __syncthreads();
    }
    const int col = n0 + wn * 32 + l31;
    const bool same_x = (x0 == xl);
    if (col < N) {
        const float bc = bias[col];
        const int64_t r0 = m0 + wm * (32 * TM);                       // wave-uniform first row
        const uint32_t lo = (uint32_t)(((int64_t)(4 * hi) * ld + col) * 4);         // per-lane byte offset (inputs)
        const uint32_t lo_o = (uint32_t)(((int64_t)(4 * hi) * out_ld + col) * 4);   // per-lane byte offset (output)
        if (r0 + 32 * TM <= M) {
            // 16 independent loads in flight per batch, then the arithmetic and the stores (no branch inside)
            auto tile_out = [&](auto same) {
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    float xv[16], x0v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = r0 + t * 32 + (r & 3) + 8 * (r >> 2);      // + 4 * hi, folded into lo
                        if (KEEPX) xv[r] = s_all[(wm * (32 * TM) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * XLD + wn * 32 + l31];
                        else xv[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xl + row * ld) + lo);
                        x0v[r] = decltype(same)::value ? xv[r] : *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x0 + row * ld) + lo);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = r0 + t * 32 + (r & 3) + 8 * (r >> 2);
                        const float lin = acc[t][r] + bc;
                        if (lin_out != nullptr)      // training: the backward needs x_l W^T + b (wave-uniform branch)
                            *reinterpret_cast<float*>(reinterpret_cast<char*>(lin_out + row * ld) + lo) = lin;
                        float v = fmaf(x0v[r], lin, xv[r]);
                        if (RELU) v = fmaxf(v, 0.f);
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(out + row * out_ld) + lo_o) = v;
                    }
                }
            };
            if (same_x) tile_out(std::true_type{}); else tile_out(std::false_type{});
        } else {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = r0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < M) {
                        const float xv = KEEPX ? s_all[(wm * (32 * TM) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * XLD + wn * 32 + l31] : xl[row * ld + col];
                        const float x0v = same_x ? xv : x0[row * ld + col];     // layer 0: x0 is x_l, one load
                        const float lin = acc[t][r] + bc;
                        if (lin_out != nullptr) lin_out[row * ld + col] = lin;
                        float v = fmaf(x0v, lin, xv);
                        if (RELU) v = fmaxf(v, 0.f);
                        out[row * out_ld + col] = v;
                    }
                }
            }
        }
    }
}

// ---- split-bf16 form (opt-in: dcn_cfg.math = bf16x3; flags bit 1 of nrx_dcn_v2_layer_fwd) --------------------------------------
// fp32 MFMA runs at the fp32 VECTOR rate (1/16 of the bf16 matrix rate): at D = 320 the layer is 85 us of matrix time next to
// ~50 us of memory time.  Each operand is split ONCE, in the load stage, into two bfloat16 parts, x = xh + xl (xh = bf16(x),
// xl = bf16(x - xh)): 16 significant bits instead of 24.  x W^T ~= xl wh + xh wl + xh wh on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation (one accumulator: a second one for the small terms bought no measurable accuracy and cost a resident block per
// CU); xl wl (2^-18 relative) is dropped.  Three
// bf16 MFMAs of 16 k per 32 cycles instead of eight fp32 MFMAs of 2 k per 64: 5.3x less matrix time, the layer becomes
// memory-bound.  Result: NOT the fp32 fma chain of the default kernel (which stays value-exact against the C oracle); error
// vs float64 measured in tests/test_dcn2_bf16x3.py (max |err| ~5e-6 of max |x W^T| at D = 320, ~10x the fp32 kernel's).
// LDS: operands as bf16, [row][k] with 80-byte rows (32 k + 16 bytes of padding): a lane's fragment (8 consecutive k of one row)
// is one ds_read_b128, conflict-free across the 16 lanes an LDS cycle serves.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int LDH = 40;      // halfs per LDS row

// x = hi + lo in bfloat16 (round-to-nearest-even both times), two elements per v_cvt_pk_bf16_f32: 10 vector instructions per
// float4.  (Written element by element through __bf16 casts the compiler spent ~50: the first version of this kernel issued 25
// vector instructions per MFMA and was bound by them -- profiles/r03_dcn_v2_bf16x3.txt.)
typedef __bf16 nrx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float nrx_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const nrx_f32x2 v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, nrx_bf16x2));
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    const nrx_f32x2 r = {a - ha, b - hb};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, nrx_bf16x2));
}
__device__ __forceinline__ void split_bf16x4(const float4& v, uint2& hi, uint2& lo) {
    split_bf16x2(v.x, v.y, hi.x, lo.x);
    split_bf16x2(v.z, v.w, hi.y, lo.y);
}

template <bool RELU>
__global__ __launch_bounds__(256, 4) void dcn_v2_layer_bf16x3_kernel(const float* __restrict__ x0, const float* __restrict__ xl, int64_t ld,
                                                                  int64_t M, int N, const float* __restrict__ W, const float* __restrict__ bias,
                                                                  float* __restrict__ out, int64_t out_ld, unsigned nx,
                                                                  float* __restrict__ lin_out) {
    __shared__ __attribute__((aligned(16))) unsigned short s_all[16384];      // one array (operand images 30 720 B; the epilogue reuses it: 32 768 B, five blocks per CU)
    static_assert(2 * BM * LDH + 2 * BN * LDH <= 16384, "operand images");
    unsigned short* const Ah = s_all;
    unsigned short* const Al = Ah + BM * LDH;
    unsigned short* const Wh = Al + BM * LDH;
    unsigned short* const Wl = Wh + BN * LDH;
    const int K = N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned nb = gridDim.x, bid = blockIdx.x;          // XCD-aware tile order, as in the fp32 kernel
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
    constexpr int AP = BM / 32;
    float4 a[AP], w[2];
    uint32_t oa[AP], ow[2];
    const char* const xtile = reinterpret_cast<const char*>(xl + m0 * ld);
    const char* const wtile = reinterpret_cast<const char*>(W + (int64_t)n0 * K);
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int64_t r = m0 + srow + 32 * p < M ? srow + 32 * p : M - 1 - m0;
        oa[p] = (uint32_t)((r * ld + skq) * 4);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = n0 + srow + 32 * p < N ? srow + 32 * p : N - 1 - n0;
        ow[p] = (uint32_t)((r * K + skq) * 4);
    }
    auto load_slab = [&](int k0) {
        const char* xk = xtile + (size_t)k0 * 4;
        const char* wk = wtile + (size_t)k0 * 4;
        if (k0 + BK <= K) {        // full slab (block-uniform): plain loads.  Selects on the loaded values make the compiler wait for the loads
                                   // where they are issued -- with them on every slab the "prefetch" was none (the loads were drained before the
                                   // MFMAs they should have run under; seen in the ISA: s_waitcnt vmcnt(4) .. (0) right behind the loads)
#pragma unroll
            for (int p = 0; p < AP; ++p) a[p] = *reinterpret_cast<const float4*>(xk + oa[p]);
#pragma unroll
            for (int p = 0; p < 2; ++p) w[p] = *reinterpret_cast<const float4*>(wk + ow[p]);
            return;
        }
        const bool ok = k0 + skq < K;                  // K % 4 == 0: a float4 is all-in or all-out; lanes past K read column 0 of their row and zero it
        const size_t back = ok ? 0 : (size_t)(k0 + skq) * 4;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const float4 t = *reinterpret_cast<const float4*>(xk + oa[p] - back);
            a[p] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float4 t = *reinterpret_cast<const float4*>(wk + ow[p] - back);
            w[p] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
        }
    };
    load_slab(0);
    // The epilogue needs x_l[m0 .. m0+127, n0 .. n0+63] -- which IS two of the A slabs this block stages (k0 = n0 and n0 + 32): the staging
    // registers of those two iterations are kept (32 VGPRs) and handed to the epilogue's lanes through LDS, instead of reading the tile from
    // memory again ten slabs later, when it has left the L2 (the launch fetched x twice: 158 MB for 84 MB -- profiles/r03_dcn_v2_bf16x3.txt).
    float4 xkeep[2][AP];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int p = 0; p < AP; ++p) xkeep[s2][p] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (k0 == n0) {                                                    // (component by component: a struct copy under a condition left the arrays in scratch memory)
#pragma unroll
            for (int p = 0; p < AP; ++p) { xkeep[0][p].x = a[p].x; xkeep[0][p].y = a[p].y; xkeep[0][p].z = a[p].z; xkeep[0][p].w = a[p].w; }
        }
        if (k0 == n0 + BK) {
#pragma unroll
            for (int p = 0; p < AP; ++p) { xkeep[1][p].x = a[p].x; xkeep[1][p].y = a[p].y; xkeep[1][p].z = a[p].z; xkeep[1][p].w = a[p].w; }
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            uint2 h, l;
            split_bf16x4(a[p], h, l);
            *reinterpret_cast<uint2*>(&Ah[(srow + 32 * p) * LDH + skq]) = h;
            *reinterpret_cast<uint2*>(&Al[(srow + 32 * p) * LDH + skq]) = l;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint2 h, l;
            split_bf16x4(w[p], h, l);
            *reinterpret_cast<uint2*>(&Wh[(srow + 32 * p) * LDH + skq]) = h;
            *reinterpret_cast<uint2*>(&Wl[(srow + 32 * p) * LDH + skq]) = l;
        }
        __syncthreads();
        if (k0 + BK < K) load_slab(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 1 && k0 + 16 >= K) break;         // the last slab's upper half is all padding
            const int ko = ks * 16 + 8 * hi;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Wh[(wn * 32 + l31) * LDH + ko]);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Wl[(wn * 32 + l31) * LDH + ko]);
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int row = wm * (32 * TM) + 32 * t + l31;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ah[row * LDH + ko]);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Al[row * LDH + ko]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);       // the two small terms, then the leading one
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // Epilogue through LDS: an accumulator tile is "one column per lane" (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)),
    // so applied straight from the registers every x_l / x_0 / lin / out access is a 4-byte one -- 16 instructions per array and tile.  With
    // the matrix time gone those were the kernel: 89 us inference, 147 us in training form (two more arrays) at D = 320.  Each wavefront
    // transposes its tile through its own 4.6 KB of LDS (the operand buffers are free now) and a lane owns 4 CONSECUTIVE columns of a
    // row: 16-byte accesses, 8 rows x 128 bytes per instruction, 4 instructions per array and tile.
    const int er = lane >> 3, ec = (lane & 7) * 4;                         // this lane's row (of 8 per pass) and first column in the tile
    // x_l tile: staging layout -> epilogue layout.  The staging thread (wave j, lane) holds rows 8 j + (lane >> 3) + 32 p, columns
    // 4 (lane & 7) + 32 s of the tile; the epilogue's (wave (wm, wn), lane) wants rows 64 wm + 32 t + 8 j + (lane >> 3), columns
    // 32 wn + 4 (lane & 7): same lane, p = 2 wm + t, s = wn -- one conflict-free 16-byte LDS write / read per piece.  (The K loop
    // ended on a barrier: the operand images are free.)
    float4 xs[TM][4];
    {
        float4* s_x = reinterpret_cast<float4*>(s_all);                    // [wave 4][p 4][s 2][lane 64]: 32 768 bytes
#pragma unroll
        for (int p = 0; p < AP; ++p)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) s_x[((wid * AP + p) * 2 + s2) * 64 + lane] = xkeep[s2][p];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) xs[t][j] = s_x[((j * AP + 2 * wm + t) * 2 + wn) * 64 + lane];
        __syncthreads();
    }
    float* s_t = reinterpret_cast<float*>(s_all) + wid * (32 * 36);       // [32 rows][36 floats] per wavefront: 4 x 4608 bytes
    const int colv = n0 + wn * 32 + ec;
    const bool same_x = (x0 == xl);
    const bool col_ok = colv < N;                                          // N % 4 == 0: the four columns are in or out together
    const int colc = col_ok ? colv : 0;                                    // clamped: loads never branch, values past N are not stored
    const float4 bc = *reinterpret_cast<const float4*>(bias + colc);
    const int64_t r0 = m0 + wm * (32 * TM);
    auto tile_out = [&](auto same) {                 // same: x0 is x_l (a stack's first layer) -- nothing but the bias is read from memory
#pragma unroll
        for (int t = 0; t < TM; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_t[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + l31] = acc[t][r];
            // (the wavefront reads back what it wrote: no block barrier -- the LDS operations of one wavefront complete in order; the
            // waits keep the compiler from moving accesses across them)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float4 lin4[4], x0v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lin4[j] = *reinterpret_cast<const float4*>(&s_t[(er + 8 * j) * 36 + ec]);
                if (!decltype(same)::value) {
                    int64_t row = r0 + t * 32 + er + 8 * j;
                    row = row < M ? row : M - 1;                            // clamp: the value is never stored
                    x0v[j] = *reinterpret_cast<const float4*>(x0 + row * ld + colc);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t row = r0 + t * 32 + er + 8 * j;
                if (row >= M || !col_ok) continue;
                const float xx = xs[t][j].x, xy = xs[t][j].y, xz = xs[t][j].z, xw = xs[t][j].w;
                const float ox = decltype(same)::value ? xx : x0v[j].x, oy = decltype(same)::value ? xy : x0v[j].y;
                const float oz = decltype(same)::value ? xz : x0v[j].z, ow4 = decltype(same)::value ? xw : x0v[j].w;
                const float4 lin = make_float4(lin4[j].x + bc.x, lin4[j].y + bc.y, lin4[j].z + bc.z, lin4[j].w + bc.w);
                if (lin_out != nullptr) *reinterpret_cast<float4*>(lin_out + row * ld + colv) = lin;
                float4 v = make_float4(fmaf(ox, lin.x, xx), fmaf(oy, lin.y, xy), fmaf(oz, lin.z, xz), fmaf(ow4, lin.w, xw));
                if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                *reinterpret_cast<float4*>(out + row * out_ld + colv) = v;
            }
        }
    };
    if (same_x) tile_out(std::true_type{}); else tile_out(std::false_type{});
}

}  // namespace

extern "C" int nrx_dcn_v2_layer_fwd(const float* x0, const float* xl, int64_t ld, int64_t batch, int32_t dim,
                                    const float* W, const float* bias, int32_t relu, float* out,
                                    int64_t out_ld, float* lin_out, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(x0 && xl && W && bias && out && batch >= 0 && dim >= 1 && ld >= dim && out_ld >= dim,
                "nrx_dcn_v2_layer_fwd: bad argument");
    NRX_REQUIRE(out != xl && out != x0, "nrx_dcn_v2_layer_fwd: out must not alias the inputs");
    if (batch == 0) return NRX_OK;
    const bool split = (relu & 2) != 0;       // flags: bit 0 = ReLU, bit 1 = split-bf16 math (see dcn_v2_layer_bf16x3_kernel)
    relu &= 1;
    const bool vec = (ld & 3) == 0 && (dim & 3) == 0 && nrx_aligned16(xl) && nrx_aligned16(W);
    const unsigned nx = (unsigned)((dim + BN - 1) / BN);
    const int64_t nblocks = (int64_t)nx * ((batch + BM - 1) / BM);
    NRX_REQUIRE(nblocks <= 0x7fffffffLL, "nrx_dcn_v2_layer_fwd: batch too large for one launch");
    dim3 grid((unsigned)nblocks);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define NRX_DCN2(R_, V_) hipLaunchKernelGGL((dcn_v2_layer_kernel<R_, V_>), grid, dim3(256), 0, st, x0, xl, ld, batch, dim, W, bias, out, out_ld, nx, lin_out)
    static const bool keepx_on = !(getenv("NRX_DCN2_KEEPX") && atoi(getenv("NRX_DCN2_KEEPX")) == 0);   // measurement knob: "0" = the epilogue fetches x_l again
    const bool vec_epi = nrx_aligned16(x0) && nrx_aligned16(out) && (out_ld & 3) == 0 && nrx_aligned16(bias) &&
                         (lin_out == nullptr || nrx_aligned16(lin_out));      // the split kernel's epilogue moves 16-byte pieces
    if (split && vec && vec_epi && batch >= 8) {        // (unaligned shapes and tiny batches take the fp32 kernel: more exact, never wrong)
        if (relu) hipLaunchKernelGGL((dcn_v2_layer_bf16x3_kernel<true>), grid, dim3(256), 0, st, x0, xl, ld, batch, dim, W, bias, out, out_ld, nx, lin_out);
        else hipLaunchKernelGGL((dcn_v2_layer_bf16x3_kernel<false>), grid, dim3(256), 0, st, x0, xl, ld, batch, dim, W, bias, out, out_ld, nx, lin_out);
    } else
    if (vec && keepx_on) {
        if (relu) hipLaunchKernelGGL((dcn_v2_layer_kernel<true, true, true>), grid, dim3(256), 0, st, x0, xl, ld, batch, dim, W, bias, out, out_ld, nx, lin_out);
        else hipLaunchKernelGGL((dcn_v2_layer_kernel<false, true, true>), grid, dim3(256), 0, st, x0, xl, ld, batch, dim, W, bias, out, out_ld, nx, lin_out);
    } else
    if (relu) { if (vec) NRX_DCN2(true, true); else NRX_DCN2(true, false); }
    else      { if (vec) NRX_DCN2(false, true); else NRX_DCN2(false, false); }
#undef NRX_DCN2
    NRX_LAUNCH_CHECK("nrx_dcn_v2_layer_fwd");
    return NRX_OK;
}
