// Dev probe: achievable v_mfma_f32_32x32x2_f32 / 16x16x4_f32 rate vs waves per SIMD and accumulator interleave,
// with and without independent VALU work mixed in.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC, int VALU>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < VALU; ++u) v[(s + u) & 7] = fmaf(v[(s + u) & 7], 1.0001f, 0.5f);
        }
    }
    float r = 0.f;
    for (int t = 0; t < NACC; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// MFMA + LDS traffic shaped like a GEMM inner loop: per MFMA, LDSR ds_read_b32 feeding the NEXT MFMA's operands
// and LDSW ds_write_b32, no VALU.
template <int LDSR, int LDSW, int BAR>
__global__ __launch_bounds__(256) void k32_lds(float* out, int iters, float a, float b) {
    __shared__ float sm[8192];
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = i * 1e-6f;
    __syncthreads();
    float fa = a, fb = b;
    const int base = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float na = fa, nb = fb;
            if (LDSR >= 1) na = sm[base + 256 * s];
            if (LDSR >= 2) nb = sm[base + 256 * s + 2048];
            if (LDSR >= 3) na += 0.f * sm[base + 256 * s + 4096];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc[1], 0, 0, 0);
            if (LDSW >= 1) sm[base + 256 * s + 6144 - 2048 * (s & 1)] = fa;
            fa = na; fb = nb;
        }
        if (BAR) __syncthreads();
    }
    float r = 0.f;
    for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    float r = 0.f;
    for (int t = 0; t < NACC; ++t) for (int i = 0; i < 4; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename F>
static void run(const char* name, F launch, int blocks, double flop_per_block_iter, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    launch(blocks, 10);
    hipDeviceSynchronize();
    hipEventRecord(s);
    launch(blocks, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("%-46s blocks/CU=%d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks / 256, ms, flop_per_block_iter * blocks * iters / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 16 * 4);
    const int iters = 20000;
    for (int bpc : {1, 2, 4}) {
        const int blocks = 256 * bpc;
        run("32x32x2 1 acc (dependent chain)", [&](int b, int n) { hipLaunchKernelGGL((k32<1, 0>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 1 * 4096, iters);
        run("32x32x2 2 acc interleaved", [&](int b, int n) { hipLaunchKernelGGL((k32<2, 0>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 2 * 4096, iters);
        run("32x32x2 4 acc interleaved", [&](int b, int n) { hipLaunchKernelGGL((k32<4, 0>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 4 * 4096, iters);
        run("32x32x2 2 acc + 4 indep. VALU per MFMA pair", [&](int b, int n) { hipLaunchKernelGGL((k32<2, 4>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 2 * 4096, iters);
        run("32x32x2 2 acc + 12 indep. VALU per MFMA pair", [&](int b, int n) { hipLaunchKernelGGL((k32<2, 12>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 2 * 4096, iters);
        run("32x32x2 1 acc + 8 indep. VALU per MFMA", [&](int b, int n) { hipLaunchKernelGGL((k32<1, 8>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 1 * 4096, iters);
        run("32x32x2 2 acc + 1 ds_read per 2 MFMA", [&](int b, int n) { hipLaunchKernelGGL((k32_lds<1, 0, 0>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 2 * 4096, iters);
        run("32x32x2 2 acc + 2 ds_read per 2 MFMA", [&](int b, int n) { hipLaunchKernelGGL((k32_lds<2, 0, 0>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 2 * 4096, iters);
        run("32x32x2 2 acc + 2 ds_read + 1 ds_write per 2 MFMA", [&](int b, int n) { hipLaunchKernelGGL((k32_lds<2, 1, 0>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 2 * 4096, iters);
        run("  same + s_barrier per 16 MFMA", [&](int b, int n) { hipLaunchKernelGGL((k32_lds<2, 1, 1>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 2 * 4096, iters);
        run("16x16x4 4 acc interleaved", [&](int b, int n) { hipLaunchKernelGGL((k16<4>), dim3(b), dim3(256), 0, 0, out, n, 1.f, 2.f); }, blocks, 4.0 * 8 * 4 * 2048, iters);
    }
    return 0;
}
