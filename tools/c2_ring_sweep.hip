// Dev harness (not part of the product): sweeps the ring form of the uniform gather kernel (csrc/nrx_embed_ring.h)
// over ring depth R / load policy / addressing on the C2 shape (26 tables x 1M x 16 fp32, B = 65536, uniform ids),
// next to the library's current nrx_embed_fwd, checking every variant bit-for-bit against it.
//   make -C news_recsys_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inews_recsys_amd/csrc \
//       tools/c2_ring_sweep.hip -Lnews_recsys_amd/lib -lnrx_hip -Wl,-rpath,'$ORIGIN/../../news_recsys_amd/lib' -o tools/bin/c2_ring_sweep
// usage: c2_ring_sweep [F=26] [rows=1000000] [D=16] [distinct_out=0] [zipf=0]   (zipf=1: Zipf(1.05) ids as bench.py --ids zipf)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "nrx_embed_ring.h"
#include "legacy/embed_fwd_uniform_r01.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
void nrx_set_error(const char*, ...) {}

static uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

__global__ void fill_random(float* p, int64_t n, uint64_t seed) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        uint64_t x = (uint64_t)i + seed;
        x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        p[i] = (float)((int64_t)(x & 0xffffff) - 0x800000) * (1.0f / 0x400000);
    }
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 26;
    const int64_t rows = argc > 2 ? atoll(argv[2]) : 1000000;
    const int D = argc > 3 ? atoi(argv[3]) : 16;
    const int distinct = argc > 4 ? atoi(argv[4]) : 0;
    const int zipf = argc > 5 ? atoi(argv[5]) : 0;
    const int64_t B = 65536;
    const int POOL = 8, STEPS = 60, REPS = 3;
    const int Q = D / 4;
    printf("# F=%d rows=%lld D=%d B=%lld pool=%d steps=%d distinct_out=%d ids=%s\n", F, (long long)rows, D, (long long)B, POOL, STEPS, distinct,
           zipf ? "zipf(1.05)" : "uniform");

    std::vector<float*> tables(F);
    for (int f = 0; f < F; ++f) {
        CK(hipMalloc(&tables[f], rows * D * 4));
        const int64_t n = rows * D;
        hipLaunchKernelGGL(fill_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tables[f], n, (uint64_t)f << 40);
    }
    std::vector<std::vector<int64_t*>> ids(POOL, std::vector<int64_t*>(F));
    std::vector<int64_t> h(B);
    uint64_t s = 1234567;
    for (int pl = 0; pl < POOL; ++pl)
        for (int f = 0; f < F; ++f) {
            for (int64_t i = 0; i < B; ++i) {
                s = mix64(s);
                if (zipf) {      // continuous power law on [1, n], inverse CDF (bench.py draw_ids)
                    const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0), a = 1.05, n = (double)(rows - 1);
                    const double r = pow(1.0 + u * (pow(n, 1.0 - a) - 1.0), 1.0 / (1.0 - a));
                    int64_t v = (int64_t)r;
                    h[i] = v < 1 ? 1 : (v > rows - 1 ? rows - 1 : v);
                } else {
                    h[i] = 1 + (int64_t)(s % (uint64_t)(rows - 1));
                }
            }
            CK(hipMalloc(&ids[pl][f], B * 8));
            CK(hipMemcpy(ids[pl][f], h.data(), B * 8, hipMemcpyHostToDevice));
        }
    const int NOUT = distinct ? POOL : 1;
    std::vector<float*> outs(NOUT);
    for (int i = 0; i < NOUT; ++i) CK(hipMalloc(&outs[i], B * F * D * 4));
    float *out_ref, *fm, *fm_ref;
    CK(hipMalloc(&out_ref, B * F * D * 4));
    CK(hipMalloc(&fm, B * 4));
    CK(hipMalloc(&fm_ref, B * 4));
    int32_t* status;
    CK(hipMalloc(&status, 16));
    CK(hipMemset(status, 0, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());

    std::vector<float> h_ref((size_t)B * F * D), h_out((size_t)B * F * D), h_fm_ref(B), h_fm(B);
    const double alg = (double)B * (F * (8.0 + 8.0 * D) + 4);

    auto feats_for = [&](int pl, std::vector<nrx_feature_t>& fe) {
        fe.resize(F);
        for (int f = 0; f < F; ++f) {
            fe[f].table = tables[f]; fe[f].index = ids[pl][f]; fe[f].weight = nullptr; fe[f].rows = rows; fe[f].dim = D; fe[f].bag_len = 0;
            fe[f].kind = NRX_SPARSE; fe[f].index_bits = 64; fe[f].out_col = D * f; fe[f].wide_col = -1; fe[f].fm_field = 1; fe[f].flags = 0;
        }
    };
    auto ua_for = [&](int pl, float* out, float* fmo) {
        UniformArgs ua;
        ua.unal = 0;
        for (int i = 0; i < NRX_MAX_FEATURES; ++i) ua.feat_id[i] = (uint8_t)i;
        for (int f = 0; f < F; ++f) { ua.table[f] = tables[f]; ua.index[f] = ids[pl][f]; ua.rows[f] = rows; ua.col4[f] = f * Q; }
        ua.batch = B; ua.out = (float4*)out; ua.ld4 = (int64_t)F * Q; ua.fm_out = fmo; ua.fm_sums = nullptr; ua.sums_ld = 0; ua.status = status; ua.n = F; ua.idx64 = 1;
        return ua;
    };

    // reference result (pool 0) from the library
    {
        std::vector<nrx_feature_t> fe;
        feats_for(0, fe);
        int rc = nrx_embed_fwd(fe.data(), F, B, out_ref, (int64_t)F * D, nullptr, 0, fm_ref, status, nullptr);
        if (rc) { printf("nrx_embed_fwd failed: %d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_ref.data(), out_ref, h_ref.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h_fm_ref.data(), fm_ref, B * 4, hipMemcpyDeviceToHost));
    }

    auto run = [&](const char* name, auto launch, bool verify, bool has_out, bool has_fm) {
        if (verify) {
            CK(hipMemset(outs[0], 0xff, B * F * D * 4));
            CK(hipMemset(fm, 0xff, B * 4));
            launch(0, outs[0]);
            CK(hipDeviceSynchronize());
            bool ok = true;
            if (has_out) {
                CK(hipMemcpy(h_out.data(), outs[0], h_out.size() * 4, hipMemcpyDeviceToHost));
                ok &= memcmp(h_out.data(), h_ref.data(), h_out.size() * 4) == 0;
            }
            if (has_fm) {
                CK(hipMemcpy(h_fm.data(), fm, B * 4, hipMemcpyDeviceToHost));
                // the concat must be bit-exact; the FM logit is a float reduction whose contraction (fma or mul + add) the
                // compiler chooses per instantiation: compared within 1e-5 of the largest logit
                float scale = 1.f;
                for (int64_t i = 0; i < B; ++i) scale = fabsf(h_fm_ref[i]) > scale ? fabsf(h_fm_ref[i]) : scale;
                for (int64_t i = 0; i < B && ok; ++i) ok &= fabsf(h_fm[i] - h_fm_ref[i]) <= 1e-5f * scale;
            }
            if (!ok) { printf("%-44s MISMATCH vs library result\n", name); fflush(stdout); return; }
        }
        double best = 1e30, sum = 0;
        for (int r = 0; r < REPS; ++r) {
            for (int i = 0; i < 5; ++i) launch(i % POOL, outs[i % NOUT]);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < STEPS; ++i) launch(i % POOL, outs[i % NOUT]);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / STEPS;
            best = std::min(best, us); sum += us;
        }
        printf("%-44s best %7.2f us  mean %7.2f us   alg %7.1f GB/s  frac(8TB/s) %.3f\n", name, best, sum / REPS, alg / best / 1e3, alg / best / 1e3 / 8000.0);
        fflush(stdout);
    };

#define LEGACY(QL, U, FM_, ST, NT)                                                                                               \
    if (Q == (1 << QL) && F >= U) run("round-1 burst kernel Q=" #QL " U=" #U " FM=" #FM_ " ST=" #ST " NT=" #NT, [&](int pl, float* out) { \
        constexpr int TB = 256 >> QL;                                                                                           \
        hipLaunchKernelGGL((embed_fwd_uniform<QL, U, true, FM_, ST, NT>), dim3((unsigned)((B + TB - 1) / TB)), dim3(256), 0, 0, ua_for(pl, out, fm)); \
    }, true, ST, FM_)
    LEGACY(2, 13, true, true, true);
    LEGACY(2, 8, true, true, true);
    LEGACY(3, 8, false, true, true);
    LEGACY(4, 5, false, true, true);

    run("library nrx_embed_fwd (current)", [&](int pl, float* out) {
        std::vector<nrx_feature_t> fe; feats_for(pl, fe);
        nrx_embed_fwd(fe.data(), F, B, out, (int64_t)F * D, nullptr, 0, fm, status, nullptr);
    }, true, true, true);

#define RING(QL, R, FM_, ST, NT, W)                                                                                          \
    if (Q == (1 << QL) && F >= R) run("ring Q=" #QL " R=" #R " FM=" #FM_ " ST=" #ST " NT=" #NT " W=" #W, [&](int pl, float* out) { \
        constexpr int TB = 256 >> QL;                                                                                           \
        hipLaunchKernelGGL((embed_fwd_ring<QL, R, FM_, ST, NT, W>), dim3((unsigned)((B + TB - 1) / TB)), dim3(256), (size_t)F * TB * 4, 0, \
                           ua_for(pl, out, fm));                                                                                \
    }, true, ST, FM_)

    // ring depth x policy (C2: Q = 4)
    RING(2, 1, true, true, true, 4);
    RING(2, 2, true, true, true, 4);
    RING(2, 4, true, true, true, 4);
    RING(2, 6, true, true, true, 4);
    RING(2, 8, true, true, true, 4);
    RING(2, 10, true, true, true, 4);
    RING(2, 13, true, true, true, 2);
    RING(2, 20, true, true, true, 2);
    RING(2, 8, true, true, false, 4);
    RING(2, 4, true, true, false, 4);
    RING(2, 8, false, true, true, 4);      // no FM epilogue
    RING(2, 8, true, false, true, 4);      // gather only (FM logit only)
    // wider rows
    RING(3, 4, false, true, true, 4);
    RING(3, 8, false, true, true, 4);
    RING(3, 8, false, true, false, 4);
    RING(3, 10, false, true, true, 4);
    RING(4, 4, false, true, true, 4);
    RING(4, 5, false, true, true, 4);
    return 0;
}
