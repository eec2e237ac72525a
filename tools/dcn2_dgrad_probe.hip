// Dev probe: what do the epilogue operands of dcn2_gemm_kernel<DGRAD> cost?  The product kernel (included as is) is launched with
// (a) g + mask bits + fold (layer 0 of a stack), (b) g + mask bits (other layers), (c) g only (no ReLU),
// (with the mask read as the forward output's values, before the bit form: 165.3 / 156.0 / 125.7 us at D = 320)
// next to dcn2_gemm_kernel<WGRAD> at its product split.  Settled clocks: WARM launches (~50 ms) before the timed ones.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include -I news_recsys_amd/csrc tools/dcn2_dgrad_probe.hip news_recsys_amd/csrc/nrx_lib.hip -o /tmp/dgrad_probe
#include "../news_recsys_amd/csrc/nrx_dcn2_bwd.hip"
#include <stdio.h>
#include <stdlib.h>

template <class F> static float timeit(F f, int warm, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < warm; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 320;
    const int64_t B = argc > 2 ? atoll(argv[2]) : 65536;
    const int warm = 400, iters = 200;
    const size_t n = (size_t)B * D;
    float *glin, *W, *g, *o, *gx0, *gxl, *xl, *gW;
    uint32_t* mbits;
    hipMalloc(&glin, n * 4); hipMalloc(&g, n * 4); hipMalloc(&o, n * 4); hipMalloc(&gx0, n * 4); hipMalloc(&gxl, n * 4); hipMalloc(&xl, n * 4);
    hipMalloc(&mbits, (size_t)((B + 31) / 32) * D * 4); hipMemset(mbits, 0x5a, (size_t)((B + 31) / 32) * D * 4);
    hipMalloc(&W, (size_t)D * D * 4); hipMalloc(&gW, (size_t)D * D * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    for (float* p : {glin, g, o, gx0, xl}) hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)D * D * 4, hipMemcpyHostToDevice);
    const unsigned nx = (unsigned)((D + BN - 1) / BN);
    const unsigned nt = nx * (unsigned)((B + BM - 1) / BM);
    auto dgrad = [&](const uint32_t* mask, const float* fold) {
        hipLaunchKernelGGL((dcn2_gemm_kernel<DGRAD, true>), dim3(nt), dim3(256), 0, 0, glin, (int64_t)D, W, (int64_t)D, B, D, (int64_t)D, (int64_t)D,
                           g, (int64_t)D, mask, (int64_t)D, gxl, (int64_t)D, nx, nt, fold, (int64_t)D, (float*)nullptr);
    };
    const double gf = 2.0 * B * D * D * 1e-9;
    float t;
    t = timeit([&] { dgrad(mbits, gx0); }, warm, iters);     printf("D=%d dgrad g+bits+fold            %7.1f us  %5.1f TF\n", D, t, gf / t * 1e3);
    t = timeit([&] { dgrad(mbits, nullptr); }, warm, iters); printf("D=%d dgrad g+bits                 %7.1f us  %5.1f TF\n", D, t, gf / t * 1e3);
    t = timeit([&] { dgrad(nullptr, nullptr); }, warm, iters); printf("D=%d dgrad g                      %7.1f us  %5.1f TF\n", D, t, gf / t * 1e3);
    for (int target : {512, 1024, 1536, 2048}) {        // 64 x 64 tiles
        const unsigned wnt = nx * (unsigned)((D + 63) / 64);
        int64_t splits = (target + wnt - 1) / wnt;
        int64_t kslice = ((B + splits - 1) / splits + BK - 1) / BK * BK;
        if (kslice < (getenv("KMIN") ? atoi(getenv("KMIN")) : 8) * BK) kslice = (getenv("KMIN") ? atoi(getenv("KMIN")) : 8) * BK;
        splits = (B + kslice - 1) / kslice;
        t = timeit([&] {
            hipLaunchKernelGGL((dcn2_gemm_kernel<WGRAD, true, 1>), dim3((unsigned)(wnt * splits)), dim3(256), 0, 0, glin, (int64_t)D, xl, (int64_t)D, (int64_t)D, D, B,
                               kslice, (const float*)nullptr, (int64_t)0, (const uint32_t*)nullptr, (int64_t)0, gW, (int64_t)D, nx, wnt, (const float*)nullptr, (int64_t)0, (float*)nullptr);
        }, warm, iters);
        printf("D=%d wgrad 64x64 (%u tiles x %lld slices of %lld rows)   %7.1f us  %5.1f TF\n", D, wnt, (long long)splits, (long long)kslice, t, gf / t * 1e3);
    }
    for (int target : {512, 1024}) {        // the product launch aims at ~1024 blocks
        const unsigned wnt = nx * (unsigned)((D + BM - 1) / BM);
        int64_t splits = (target + wnt - 1) / wnt;
        int64_t kslice = ((B + splits - 1) / splits + BK - 1) / BK * BK;
        splits = (B + kslice - 1) / kslice;
        t = timeit([&] {
            hipLaunchKernelGGL((dcn2_gemm_kernel<WGRAD, true>), dim3((unsigned)(wnt * splits)), dim3(256), 0, 0, glin, (int64_t)D, xl, (int64_t)D, (int64_t)D, D, B,
                               kslice, (const float*)nullptr, (int64_t)0, (const uint32_t*)nullptr, (int64_t)0, gW, (int64_t)D, nx, wnt, (const float*)nullptr, (int64_t)0, (float*)nullptr);
        }, warm, iters);
        printf("D=%d wgrad (%u tiles x %lld slices of %lld rows)   %7.1f us  %5.1f TF\n", D, wnt, (long long)splits, (long long)kslice, t, gf / t * 1e3);
    }
    {   // dgrad and wgrad of one layer: back to back on one stream vs side by side on two (both read glin; each alone sits at the sum of its
        // matrix and memory bounds -- do the phases of two different kernels overlap?)
        const unsigned wnt = nx * (unsigned)((D + 63) / 64);
        int64_t splits = (1536 + wnt - 1) / wnt;
        int64_t kslice = ((B + splits - 1) / splits + BK - 1) / BK * BK;
        if (kslice < 16 * BK) kslice = 16 * BK;
        splits = (B + kslice - 1) / kslice;
        hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipEvent_t ef, ej; hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
        auto pair = [&](bool two) {
            hipStream_t sw = two ? s2 : s1;
            if (two) { hipEventRecord(ef, s1); hipStreamWaitEvent(s2, ef, 0); }
            hipLaunchKernelGGL((dcn2_gemm_kernel<DGRAD, true>), dim3(nt), dim3(256), 0, s1, glin, (int64_t)D, W, (int64_t)D, B, D, (int64_t)D, (int64_t)D,
                               g, (int64_t)D, mbits, (int64_t)D, gxl, (int64_t)D, nx, nt, (const float*)nullptr, (int64_t)D, (float*)nullptr);
            hipLaunchKernelGGL((dcn2_gemm_kernel<WGRAD, true, 1>), dim3((unsigned)(wnt * splits)), dim3(256), 0, sw, glin, (int64_t)D, xl, (int64_t)D, (int64_t)D, D, B,
                               kslice, (const float*)nullptr, (int64_t)0, (const uint32_t*)nullptr, (int64_t)0, gW, (int64_t)D, nx, wnt, (const float*)nullptr, (int64_t)0, (float*)nullptr);
            if (two) { hipEventRecord(ej, s2); hipStreamWaitEvent(s1, ej, 0); }
        };
        for (int two = 0; two < 2; ++two) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < warm; ++i) pair(two);
            hipEventRecord(e0, s1);
            for (int i = 0; i < iters; ++i) pair(two);
            hipEventRecord(e1, s1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("D=%d dgrad + wgrad %s: %7.1f us per pair\n", D, two ? "on two streams (fork / join by events)" : "back to back on one stream", ms * 1e3f / iters);
        }
    }
    return 0;
}
