// Dev probe (not part of the product): variants of the C2-shaped gather (F x [rows,16] fp32 tables,
// B samples, out [B, F*16]) to find which load/store cache policy and lane mapping gives 64-byte
// (instead of 128-byte) DRAM fetches for 64-byte rows and full-line stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_probe.hip -o tools/gather_probe && tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define GLOBAL __attribute__((address_space(1)))
#define CONSTAS __attribute__((address_space(4)))
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u64x2 = __attribute__((ext_vector_type(2))) unsigned long long;

constexpr int MAXF = 64;
struct Args {
    const float* table[MAXF];
    const int64_t* index[MAXF];
    float* out;
    int64_t batch;
    int32_t n;
    int32_t ld4;
};

enum { LD_PLAIN = 0, LD_NT = 1, LD_SC1 = 2, LD_SC01 = 3, LD_ATOM8 = 4 };
enum { ST_PLAIN = 0, ST_NT = 1 };

template <int LD>
__device__ __forceinline__ void issue_row_load(f32x4& v, const float* table, int64_t off16) {
    const GLOBAL f32x4* p = (const GLOBAL f32x4*)(table) + off16;
    if (LD == LD_PLAIN) v = *p;
    else if (LD == LD_NT) v = __builtin_nontemporal_load(p);
    else if (LD == LD_SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if (LD == LD_SC01) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else {
        const GLOBAL unsigned long long* q = (const GLOBAL unsigned long long*)p;
        u64x2 t;
        t.x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = __builtin_bit_cast(f32x4, t);
    }
}

template <int ST>
__device__ __forceinline__ void row_store(float* out, int64_t off16, f32x4 v) {
    GLOBAL f32x4* p = (GLOBAL f32x4*)(out) + off16;
    if (ST == ST_NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// mapping B: lane = (sample, quarter); U features in flight
template <int U, int LD, int ST, bool STORE>
__global__ __launch_bounds__(256) void gather_b(const Args a_) {
    const CONSTAS Args* a = (const CONSTAS Args*)__builtin_amdgcn_kernarg_segment_ptr();
    const int q = threadIdx.x & 3;
    const int64_t b = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    if (b >= a->batch) return;
    float acc = 0.f;
    for (int f0 = 0; f0 < a->n; f0 += U) {
        int64_t id[U];
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) id[u] = ((const GLOBAL int64_t*)a->index[f0 + u])[b];
#pragma unroll
        for (int u = 0; u < U; ++u) issue_row_load<LD>(v[u], a->table[f0 + u], id[u] * 4 + q);
        if (LD == LD_SC1 || LD == LD_SC01) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (LD == LD_SC1 || LD == LD_SC01) asm volatile("" : "+v"(v[u]));
            if (STORE) row_store<ST>(a->out, b * a->ld4 + (f0 + u) * 4 + q, v[u]);
            else acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (!STORE) if (acc == 12345.678f) a->out[b] = acc;
}

// mapping C: lane = (sample s = lane>>3 within wave, feature parity p, quarter q): each 8-lane group
// stores one full aligned 128 B line (two adjacent 64 B fields of one sample).  U pairs in flight.
template <int U, int LD, int ST>
__global__ __launch_bounds__(256) void gather_c(const Args a_) {
    const CONSTAS Args* a = (const CONSTAS Args*)__builtin_amdgcn_kernarg_segment_ptr();
    const int q = threadIdx.x & 3;
    const int p = (threadIdx.x >> 2) & 1;
    const int64_t b = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (b >= a->batch) return;
    const int npair = a->n / 2;     // n even
    for (int j0 = 0; j0 < npair; j0 += U) {
        int64_t id[U];
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // per-lane select between the two features of the pair
            const int64_t i0 = ((const GLOBAL int64_t*)a->index[2 * (j0 + u)])[b];
            const int64_t i1 = ((const GLOBAL int64_t*)a->index[2 * (j0 + u) + 1])[b];
            id[u] = p ? i1 : i0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* t = p ? a->table[2 * (j0 + u) + 1] : a->table[2 * (j0 + u)];
            issue_row_load<LD>(v[u], t, id[u] * 4 + q);
        }
        if (LD == LD_SC1 || LD == LD_SC01) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (LD == LD_SC1 || LD == LD_SC01) asm volatile("" : "+v"(v[u]));
            row_store<ST>(a->out, b * a->ld4 + (2 * (j0 + u) + p) * 4 + q, v[u]);
        }
    }
}

// pure streaming copy of the same output size (reference point for the store side)
__global__ __launch_bounds__(256) void stream_write(float* out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { f32x4 v = {1.f, 2.f, 3.f, (float)i}; ((GLOBAL f32x4*)out)[i] = v; }
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 26;
    const int64_t rows = argc > 2 ? atoll(argv[2]) : 1000000;
    const int64_t B = 65536;
    const int POOL = 4, STEPS = 30;
    std::vector<float*> tables(F);
    for (int f = 0; f < F; ++f) { CK(hipMalloc(&tables[f], rows * 64)); CK(hipMemset(tables[f], 0x3c, rows * 64)); }
    std::vector<std::vector<int64_t*>> ids(POOL, std::vector<int64_t*>(F));
    std::vector<int64_t> h(B);
    uint64_t s = 88172645463325252ull;
    for (int pl = 0; pl < POOL; ++pl)
        for (int f = 0; f < F; ++f) {
            for (int64_t i = 0; i < B; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = 1 + (int64_t)(s % (uint64_t)(rows - 1)); }
            CK(hipMalloc(&ids[pl][f], B * 8));
            CK(hipMemcpy(ids[pl][f], h.data(), B * 8, hipMemcpyHostToDevice));
        }
    float* out;
    CK(hipMalloc(&out, B * F * 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch(i % POOL);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < STEPS; ++i) launch(i % POOL);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / STEPS;
        const double alg = (double)B * F * (8 + 64 + 64);
        printf("%-34s %8.1f us   alg %7.1f GB/s\n", name, us, alg / us / 1e3);
        fflush(stdout);
    };
    auto mk = [&](int pl) { Args a; for (int f = 0; f < F; ++f) { a.table[f] = tables[f]; a.index[f] = ids[pl][f]; } a.out = out; a.batch = B; a.n = F; a.ld4 = F * 4; return a; };

#define RUN_B(U, LD, ST, STORE) run("B U=" #U " " #LD " " #ST " store=" #STORE, [&](int pl) { hipLaunchKernelGGL((gather_b<U, LD, ST, STORE>), dim3(B / 64), dim3(256), 0, 0, mk(pl)); })
#define RUN_C(U, LD, ST) run("C U=" #U " " #LD " " #ST, [&](int pl) { hipLaunchKernelGGL((gather_c<U, LD, ST>), dim3(B / 32), dim3(256), 0, 0, mk(pl)); })
    run("stream_write (same bytes out)", [&](int) { hipLaunchKernelGGL(stream_write, dim3((unsigned)(B * F * 4 / 256)), dim3(256), 0, 0, out, B * F * 4); });
    if (F % 13 == 0) {
        RUN_B(13, LD_PLAIN, ST_PLAIN, true);
        RUN_B(13, LD_NT, ST_PLAIN, true);
        RUN_B(13, LD_SC1, ST_PLAIN, true);
        RUN_B(13, LD_SC01, ST_PLAIN, true);
        RUN_B(13, LD_ATOM8, ST_PLAIN, true);
        RUN_B(13, LD_PLAIN, ST_NT, true);
        RUN_B(13, LD_NT, ST_NT, true);
        RUN_B(13, LD_PLAIN, ST_PLAIN, false);
        RUN_B(13, LD_NT, ST_PLAIN, false);
        RUN_B(13, LD_SC1, ST_PLAIN, false);
        RUN_C(13, LD_PLAIN, ST_PLAIN);
        RUN_C(13, LD_NT, ST_PLAIN);
        RUN_C(13, LD_NT, ST_NT);
        RUN_C(13, LD_SC1, ST_NT);
    }
    return 0;
}
