
// Dev probe: where do the non-matrix microseconds of dcn_v2_layer_kernel go?  Variants of the product kernel
// (body taken from news_recsys_amd/csrc/nrx_dcn2.hip by tools/make_dcn2_phase_probe.py) with the main-loop global loads
// and / or the epilogue's memory traffic removed.  Build: hipcc --offload-arch=gfx950 -O3 -I include -I news_recsys_amd/csrc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "nrx_common.h"
#include <type_traits>
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TM = PROBE_TM;                       // 32-row MFMA tiles per wave along M
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

template <bool RELU, bool VEC, bool MAINLOAD, int EPI, int PRIO>
__global__ __launch_bounds__(256, PROBE_OCC) void dcn_v2_layer_kernel(const float* __restrict__ x0, const float* __restrict__ xl, int64_t ld,
                                                           int64_t M, int N, const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ out, int64_t out_ld, unsigned nx,
                                                           float* __restrict__ lin_out) {
    __shared__ float As[BK * LDA];
    __shared__ float Ws[BK * LDW];
    const int K = N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform by construction; tell the compiler
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, hi = lane >> 5;
#ifdef PROBE_ASM_KLOOP
    static_assert(TM == 2 && LDA == 129 && LDW == 65, "the asm K loop's offsets are written for the 128x64 tile");
    const uint32_t lds_bw = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) float*)Ws + hi * LDW + wn * 32 + l31);
    const uint32_t lds_ba = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) float*)As + hi * LDA + wm * (32 * TM) + l31);
#endif
    // XCD-aware tile order (guide T1, bijective form): hardware places block b on XCD b % 8; remap so
    // that the nx column tiles of one 128-row panel of x_l are consecutive on ONE XCD and share its L2
    // (without it the panel was fetched from DRAM once per XCD: 516 MB read vs ~250 MB, measured).
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    const unsigned xcd = bid & 7u, qd = nb >> 3, rm = nb & 7u;
    const unsigned logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int64_t m0 = (int64_t)(logical / nx) * BM;
    const int n0 = (int)(logical % nx) * BN;

    if (PRIO >= 10) {
        // stagger: a first-round block sleeps (its wave slot) x (PRIO - 10) x ~1.1 us before it starts (s_sleep 40 = 2560 cycles)
        if (blockIdx.x < 1280u) {
            __shared__ unsigned s_slot2;
            if (tid == 0) s_slot2 = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 15u;
            __syncthreads();
            const unsigned n = __builtin_amdgcn_readfirstlane(s_slot2) * (unsigned)(PRIO - 10);
            for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(40);
        }
    } else if (PRIO) {
        // distinct issue priority per co-resident block: PRIO 1 = this wave's slot on its SIMD, PRIO 2 = wave 0's slot for the whole block
        unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 15u;     // HW_REG_HW_ID.wave_id
        if (PRIO == 2) {
            __shared__ unsigned s_slot;
            if (tid == 0) s_slot = slot;
            __syncthreads();
            slot = __builtin_amdgcn_readfirstlane(s_slot);
        }
        switch (slot & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // EPI == 2: the epilogue's x0 operands are fetched now, into registers, and arrive under the K loop
    float x0pre[TM][16];
    if (EPI == 2 && n0 + wn * 32 + l31 < N) {
        const int colp = n0 + wn * 32 + l31;
        const int64_t r0p = m0 + wm * (32 * TM);
        const uint32_t lop = (uint32_t)(((int64_t)(4 * hi) * ld + colp) * 4);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = r0p + t * 32 + (r & 3) + 8 * (r >> 2);
                x0pre[t][r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x0 + row * ld) + lop);
            }
    }

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
                const bool ok = k0 + skq < K;                  // out-of-range lanes re-read slab 0 and zero it
                const size_t back = ok ? 0 : (size_t)k0 * 4;
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
        if (MAINLOAD && k0 + BK < K) load_slab(k0 + BK);      // next slab: loads stay in flight across the MFMA block below
        // fragments of half a slab (8 k-pairs: 8 B + 32 A dwords) are read ahead of a dense block of
        // 32 MFMAs, so the LDS latency is paid twice per slab instead of once per MFMA group
#ifdef PROBE_ASM_KLOOP
        // hand-scheduled half slab: 12 fragment reads (ds_read_b32 with immediate offsets: no address VALU) in flight, then every
        // MFMA pair waits only for its own three fragments while the reads of the pair four steps ahead are issued behind it
        // (lgkmcnt is a 4-bit counter: never more than 12 LDS reads outstanding)
#define NRX_RDW(D_, H_, I_) "ds_read_b32 %[" #D_ "], %[bw] offset:(16*" #H_ "+2*" #I_ ")*260\n"
#define NRX_RDA(D_, H_, I_, T_) "ds_read_b32 %[" #D_ "], %[ba] offset:(16*" #H_ "+2*" #I_ ")*516+128*" #T_ "\n"
#define NRX_RD3(H_, I_) NRX_RDW(b##I_, H_, I_) NRX_RDA(a0##I_, H_, I_, 0) NRX_RDA(a1##I_, H_, I_, 1)
#define NRX_MM(I_) "v_mfma_f32_32x32x2_f32 %[c0], %[a0" #I_ "], %[b" #I_ "], %[c0]\n" "v_mfma_f32_32x32x2_f32 %[c1], %[a1" #I_ "], %[b" #I_ "], %[c1]\n"
#define NRX_HALF(H_)                                                                                                        \
        {                                                                                                                    \
            float b0, b1, b2, b3, b4, b5, b6, b7, a00, a01, a02, a03, a04, a05, a06, a07, a10, a11, a12, a13, a14, a15, a16, a17;   \
            asm volatile(NRX_RD3(H_, 0) NRX_RD3(H_, 1) NRX_RD3(H_, 2) NRX_RD3(H_, 3)                                         \
                         "s_waitcnt lgkmcnt(9)\n" NRX_MM(0) NRX_RD3(H_, 4)                                                   \
                         "s_waitcnt lgkmcnt(9)\n" NRX_MM(1) NRX_RD3(H_, 5)                                                   \
                         "s_waitcnt lgkmcnt(9)\n" NRX_MM(2) NRX_RD3(H_, 6)                                                   \
                         "s_waitcnt lgkmcnt(9)\n" NRX_MM(3) NRX_RD3(H_, 7)                                                   \
                         "s_waitcnt lgkmcnt(9)\n" NRX_MM(4)                                                                  \
                         "s_waitcnt lgkmcnt(6)\n" NRX_MM(5)                                                                  \
                         "s_waitcnt lgkmcnt(3)\n" NRX_MM(6)                                                                  \
                         "s_waitcnt lgkmcnt(0)\n" NRX_MM(7)                                                                  \
                         : [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [b4] "=&v"(b4), [b5] "=&v"(b5), [b6] "=&v"(b6), [b7] "=&v"(b7), \
                           [a00] "=&v"(a00), [a01] "=&v"(a01), [a02] "=&v"(a02), [a03] "=&v"(a03), [a04] "=&v"(a04), [a05] "=&v"(a05), [a06] "=&v"(a06), [a07] "=&v"(a07), \
                           [a10] "=&v"(a10), [a11] "=&v"(a11), [a12] "=&v"(a12), [a13] "=&v"(a13), [a14] "=&v"(a14), [a15] "=&v"(a15), [a16] "=&v"(a16), [a17] "=&v"(a17), \
                           [c0] "+v"(acc[0]), [c1] "+v"(acc[1])                                                              \
                         : [bw] "v"(lds_bw), [ba] "v"(lds_ba)                                                                \
                         : "memory");                                                                                        \
        }
        NRX_HALF(0)
        if (k0 + BK / 2 < K) NRX_HALF(1)
#else
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && k0 + BK / 2 >= K) break;      // the last slab's upper half is all padding (K % 32 <= 16): skip its MFMAs
            float fb[BK / 4], fa[TM][BK / 4];
#pragma unroll
            for (int i = 0; i < BK / 4; ++i) {
                const int kr = 2 * (h * (BK / 4) + i) + hi;
#ifdef PROBE_VOLATILE_LDS
                fb[i] = ((const volatile __attribute__((address_space(3))) float*)Ws)[kr * LDW + wn * 32 + l31];
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[t][i] = ((const volatile __attribute__((address_space(3))) float*)As)[kr * LDA + wm * (32 * TM) + 32 * t + l31];
#else
                fb[i] = Ws[kr * LDW + wn * 32 + l31];
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[t][i] = As[kr * LDA + wm * (32 * TM) + 32 * t + l31];
#endif
            }
#pragma unroll
            for (int i = 0; i < BK / 4; ++i)
#pragma unroll
                for (int t = 0; t < TM; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t][i], fb[i], acc[t], 0, 0, 0);
        }
#endif
        __syncthreads();             // slab fully consumed before the next LDS write
    }

    // Epilogue, also written for few VALU instructions: the address of register r's element is a wave-uniform
    // row pointer (tile base + constant * ld, scalar unit) plus one fixed per-lane byte offset, so a full
    // tile costs add-bias, fma, max per element; only a tile that crosses M takes the guarded path.
    if (EPI == 1) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[t][r];
        if (s == 1.2345e38f) out[tid] = s;
        return;
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
                        xv[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xl + row * ld) + lo);
                        x0v[r] = decltype(same)::value ? xv[r] : (EPI == 2 ? x0pre[t][r] : *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x0 + row * ld) + lo));
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
                        const float xv = xl[row * ld + col];
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

}  // namespace
template <bool MAINLOAD, int EPI, int PRIO = 0>
static void run(const char* name, const float* x0, const float* x, const float* W, const float* b, float* out, int64_t B, int D, int iters) {
    const unsigned nx = (unsigned)((D + BN - 1) / BN);
    dim3 grid((unsigned)(nx * ((B + BM - 1) / BM)));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int warm = getenv("WARM") ? atoi(getenv("WARM")) : 300;      // ~40 ms of the same launch: the clocks settle (5 launches measure the ramp)
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL((dcn_v2_layer_kernel<true, true, MAINLOAD, EPI, PRIO>), grid, dim3(256), 0, 0, x0, x, (int64_t)D, B, D, W, b, out, (int64_t)D, nx, (float*)nullptr);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((dcn_v2_layer_kernel<true, true, MAINLOAD, EPI, PRIO>), grid, dim3(256), 0, 0, x0, x, (int64_t)D, B, D, W, b, out, (int64_t)D, nx, (float*)nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, tf = (2.0 * D * D + 3.0 * D) * B / us * 1e-6;
    unsigned long long hsh = 0;
    if (EPI == 0 && MAINLOAD) {             // FNV-1a over the output bits: equal across builds of this probe <=> same values
        std::vector<unsigned> h((size_t)B * D);
        hipMemcpy(h.data(), out, (size_t)B * D * 4, hipMemcpyDeviceToHost);
        hsh = 1469598103934665603ull;
        for (unsigned v : h) { hsh ^= v; hsh *= 1099511628211ull; }
    }
    printf("D=%d %-52s %8.1f us  %6.1f TF  %5.1f %% of 157.3   out hash %016llx\n", D, name, us, tf, tf / 157.3 * 100, hsh);
}
__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        if (seed == 0u) { p[i] = 0.f; continue; }
        if (seed >= 100u) { p[i] = ((h & 0xffff) / 65536.f - 0.5f) * 0.2f; continue; }      // LOWENT=1: 16 random mantissa bits
        unsigned g = h * 747796405u + 2891336453u; g ^= g >> 16;
        p[i] = ((float)(g >> 8) * (1.f / 16777216.f) - 0.5f) * 0.2f;                          // 24 random mantissa bits
    }
}
int main(int argc, char** argv) {
    const int64_t B = 65536; const int iters = 100;
    const unsigned z = getenv("ZERO") ? 0u : getenv("LOWENT") ? 100u : 1u;     // ZERO=1: all operands zero-filled (the DVFS check: same instructions, less switching power)
    printf(z == 0u ? "== zero-filled operands\n" : z == 100u ? "== operands with 16 random mantissa bits\n" : "== operands with 24 random mantissa bits\n");
    for (int D : {320, 112}) {
        float *x, *x0, *W, *b, *out;
        hipMalloc(&x, B * D * 4); hipMalloc(&x0, B * D * 4); hipMalloc(&out, B * D * 4); hipMalloc(&W, (size_t)D * D * 4); hipMalloc(&b, D * 4);
        fill<<<1024, 256>>>(x, B * D, 1 * z); fill<<<1024, 256>>>(x0, B * D, 2 * z); fill<<<64, 256>>>(W, (size_t)D * D, 3 * z); fill<<<1, 256>>>(b, D, 4 * z);
        hipDeviceSynchronize();
        run<true, 0>("product form (x0 != x_l, inference)", x0, x, W, b, out, B, D, iters);
        run<true, 0>("product form, x0 == x_l", x, x, W, b, out, B, D, iters);
        run<true, 0, 12>("product, first-round blocks staggered 2.2 us per wave slot", x0, x, W, b, out, B, D, iters);
        run<true, 0, 15>("product, first-round blocks staggered 5.5 us per wave slot", x0, x, W, b, out, B, D, iters);
        run<true, 0, 20>("product, first-round blocks staggered 11 us per wave slot", x0, x, W, b, out, B, D, iters);
        run<true, 0, 1>("product + per-slot issue priority (wave's own slot)", x0, x, W, b, out, B, D, iters);
        run<true, 0, 2>("product + per-slot issue priority (block-uniform)", x0, x, W, b, out, B, D, iters);
        run<false, 1, 2>("LDS + MFMA loop only + priority (block-uniform)", x0, x, W, b, out, B, D, iters);
        run<true, 2>("x0 tile fetched before the K loop (registers)", x0, x, W, b, out, B, D, iters);
        run<true, 1>("no epilogue memory traffic", x0, x, W, b, out, B, D, iters);
        run<false, 0>("no main-loop global loads (slab 0 reused)", x0, x, W, b, out, B, D, iters);
        run<false, 1>("neither: LDS + MFMA loop only", x0, x, W, b, out, B, D, iters);
        hipFree(x); hipFree(x0); hipFree(W); hipFree(b); hipFree(out);
    }
    return 0;
}
