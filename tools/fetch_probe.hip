// Dev probe: does ANY (allocation attribute x load policy) pair make a 64-byte embedding row cost less than a 128-byte EA request?
// Gather-only kernel over 26 x 1M x 16 fp32 rows, uniform ids (the C2 read side: 123 MB useful = 109 MB of rows + 13.6 MB of ids);
// table memory from hipMalloc / hipExtMallocWithFlags(Uncached | Finegrained); loads plain, nontemporal, sc1, sc0 sc1.
// Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum (128-byte reads) to see the request sizes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_probe.hip -o tools/bin/fetch_probe;  usage: fetch_probe [default|uncached|finegrained]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ f4 ld(const f4* p) {
    f4 v;
    if (POLICY == 0) v = *p;
    else if (POLICY == 1) v = __builtin_nontemporal_load(p);
    else if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// 4 lanes per row (16 B each), 16 rows per wave instruction, every lane group walks the F features of its sample
template <int POLICY>
__global__ __launch_bounds__(256) void gather_only(const float* __restrict__ tab, const int64_t* __restrict__ ids, int B, int F, int64_t rows,
                                                   float* __restrict__ sink) {
    const int q = threadIdx.x & 3;
    const int64_t b = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
    if (b >= B) return;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (POLICY <= 1) {
#pragma unroll 13
        for (int f = 0; f < F; ++f) {
            const int64_t id = ids[(int64_t)f * B + b];
            acc += ld<POLICY>(reinterpret_cast<const f4*>(tab) + ((int64_t)f * rows + id) * 4 + q);
        }
    } else {
        for (int f = 0; f < F; ++f) {
            const int64_t id = ids[(int64_t)f * B + b];
            acc += ld<POLICY>(reinterpret_cast<const f4*>(tab) + ((int64_t)f * rows + id) * 4 + q);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) sink[b] = acc.x;
}

static uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void fill_random(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h >> 8) * (1.f / 16777216.f) - 0.5f;
    }
}

template <int POLICY>
static void run(const char* mode, const char* name, const float* tab, const int64_t* ids, int B, int F, int64_t rows, float* sink) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    dim3 grid((unsigned)(((int64_t)B * 4 + 255) / 256));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gather_only<POLICY>, grid, dim3(256), 0, 0, tab, ids, B, F, rows, sink);
    hipEventRecord(s);
    const int iters = 30;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gather_only<POLICY>, grid, dim3(256), 0, 0, tab, ids, B, F, rows, sink);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double us = ms / iters * 1e3;
    printf("%-12s %-28s %8.1f us   %7.1f GB/s of rows + ids\n", mode, name, us, ((double)B * F * 72) / (us * 1e-6) / 1e9);
}

int main(int argc, char** argv) {
    const int B = 65536, F = 26; const int64_t rows = 1048576;
    const char* mode = argc > 1 ? argv[1] : "default";
    float* tab; float* sink; int64_t* ids;
    const size_t tb = (size_t)F * rows * 64;
    hipError_t e = hipSuccess;
    if (!strcmp(mode, "uncached")) e = hipExtMallocWithFlags((void**)&tab, tb, hipDeviceMallocUncached);
    else if (!strcmp(mode, "finegrained")) e = hipExtMallocWithFlags((void**)&tab, tb, hipDeviceMallocFinegrained);
    else e = hipMalloc(&tab, tb);
    if (e != hipSuccess) { printf("allocation (%s) failed: %s\n", mode, hipGetErrorString(e)); return 1; }
    hipMalloc(&sink, (size_t)B * 4); hipMalloc(&ids, (size_t)B * F * 8);
    int64_t* h = (int64_t*)malloc((size_t)B * F * 8);
    for (int f = 0; f < F; ++f) for (int b = 0; b < B; ++b) h[(size_t)f * B + b] = (int64_t)(mix64((uint64_t)b * F + f) % (uint64_t)rows);
    hipMemcpy(ids, h, (size_t)B * F * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, tab, (size_t)F * rows * 16); hipDeviceSynchronize();
    run<0>(mode, "plain loads", tab, ids, B, F, rows, sink);
    run<1>(mode, "nontemporal loads", tab, ids, B, F, rows, sink);
    run<2>(mode, "sc1 loads (one in flight)", tab, ids, B, F, rows, sink);
    run<3>(mode, "sc0 sc1 loads (one in flight)", tab, ids, B, F, rows, sink);
    return 0;
}
