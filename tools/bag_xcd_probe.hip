// Dev probe: masked-mean pooling of a [B, L] history over a table that does NOT fit one XCD's L2 (C4: 200 000 x 16 fp32 = 12.8 MB, L2 = 4 MB per XCD),
// (a) the product's mapping: a block pools its 64 samples over the whole table;
// (b) XCD-partitioned: the blocks of XCD group p (hardware places block b on XCD b % 8; p = xcd % P) pool only the rows of the p-th P-th of the table --
//     every sample is visited by P blocks, each block's staging wave compacts the sample's entries to its partition (ballot + popcount, order kept),
//     the partial sums are combined with float atomics on a pre-zeroed output (a product version would use a scratch + a second pass).
// The table quarter (3.2 MB) then stays L2-resident on its XCDs: ~100 % hits for +(P - 1) x 12 B of id / mask re-reads per lookup.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/bag_xcd_probe.hip -o tools/bin/bag_xcd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
struct Pair { int32_t id; float w; };
constexpr int L = 50, D = 16, TB = 64;      // 64 samples per block, 4 lanes per sample

template <int P>
__global__ __launch_bounds__(256) void pool_kernel(const int64_t* __restrict__ ids, const float* __restrict__ mask, const float* __restrict__ table,
                                                   int64_t rows, int64_t B, float* __restrict__ out) {
    __shared__ Pair s_bag[TB][L | 1];
    __shared__ int s_cnt[TB];
    __shared__ float s_den[TB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned bid = blockIdx.x, xcd = bid & 7u;
    const int p = P > 1 ? (int)(xcd % P) : 0;
    // sample tile: with P partitions the 8 XCDs form 8 / P groups of P; group g of "super tile" t handles tile t * (8 / P) + g
    const int64_t tile = P > 1 ? (int64_t)(bid >> 3) * (8 / P) + (xcd / P) : bid;
    const int64_t b0 = tile * TB;
    if (b0 >= B) return;
    const int64_t lo = rows * p / P, hi = rows * (p + 1) / P;
    for (int s = wid; s < TB; s += 4) {                       // one wavefront stages one sample at a time: lanes = bag positions
        const int64_t b = b0 + s;
        int64_t id = 0; float w = 0.f;
        if (lane < L && b < B) { id = ids[b * L + lane]; w = mask[b * L + lane]; }
        float den = w;
        for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o, 64);
        const bool keep = w != 0.f && id >= lo && id < hi;
        const unsigned long long m = __ballot(keep);
        if (keep) { Pair pr; pr.id = (int32_t)id; pr.w = w; s_bag[s][__popcll(m & ((1ull << lane) - 1))] = pr; }
        if (lane == 0) { s_cnt[s] = __popcll(m); s_den[s] = den + 1e-8f; }
    }
    __syncthreads();
    const int q = tid & 3, sb = tid >> 2;
    const int64_t b = b0 + sb;
    if (b >= B) return;
    const int n = s_cnt[sb];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    int l = 0;
    for (; l + 4 <= n; l += 4) {
        Pair pr[4]; f4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] = s_bag[sb][l + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const f4*>(table + (int64_t)pr[u].id * D + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += r[u] * pr[u].w;
    }
    for (; l < n; ++l) { const Pair pr = s_bag[sb][l]; acc += *reinterpret_cast<const f4*>(table + (int64_t)pr.id * D + 4 * q) * pr.w; }
    const float d = s_den[sb];
    float* o = out + b * D + 4 * q;
    if (P > 1) { unsafeAtomicAdd(o, acc.x / d); unsafeAtomicAdd(o + 1, acc.y / d); unsafeAtomicAdd(o + 2, acc.z / d); unsafeAtomicAdd(o + 3, acc.w / d); }
    else *reinterpret_cast<f4*>(o) = acc / d;
}

static uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

template <int P>
static void run(const char* name, const int64_t* const* ids, const float* const* mask, const float* table, int64_t rows, int64_t B, float* out, std::vector<float>* keep) {
    const int64_t tiles = (B + TB - 1) / TB;
    const unsigned grid = (unsigned)(P > 1 ? (tiles + 8 / P - 1) / (8 / P) * 8 : tiles);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    auto go = [&](int i) {
        if (P > 1) hipMemsetAsync(out, 0, (size_t)B * D * 4, 0);
        hipLaunchKernelGGL(pool_kernel<P>, dim3(grid), dim3(256), 0, 0, ids[i % 8], mask[i % 8], table, rows, B, out);
    };
    for (int i = 0; i < 20; ++i) go(i);
    hipEventRecord(s, 0);
    const int iters = 100;
    for (int i = 0; i < iters; ++i) go(i);
    hipEventRecord(e, 0); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    go(0); hipDeviceSynchronize();
    std::vector<float> h((size_t)B * D);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    double maxrel = 0;
    if (keep->empty()) *keep = h; else for (size_t i = 0; i < h.size(); ++i) { const double d = fabs((double)h[i] - (*keep)[i]) / fmax(1.0, fabs((double)(*keep)[i])); if (d > maxrel) maxrel = d; }
    printf("%-44s %7.1f us per launch%s   max rel diff vs the unpartitioned result %.1e\n", name, ms / iters * 1e3, P > 1 ? " (incl. the output memset)" : "", maxrel);
}

int main() {
    const int64_t B = 65536, rows = 200000;
    float* table; hipMalloc(&table, rows * D * 4);
    std::vector<float> ht((size_t)rows * D);
    for (size_t i = 0; i < ht.size(); ++i) ht[i] = (float)((mix64(i) >> 40) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(table, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    int64_t* ids[8]; float* mask[8];
    std::vector<int64_t> hi((size_t)B * L); std::vector<float> hm((size_t)B * L);
    for (int k = 0; k < 8; ++k) {                               // a pool of 8 id sets, as bench.py cycles them
        for (size_t i = 0; i < hi.size(); ++i) { hi[i] = 1 + (int64_t)(mix64(i * 8 + k) % (uint64_t)(rows - 1)); hm[i] = (i % L) < (mix64((i / L) * 8 + k + 77) % (L + 1)) ? 1.f : 0.f; }
        hipMalloc(&ids[k], hi.size() * 8); hipMalloc(&mask[k], hm.size() * 4);
        hipMemcpy(ids[k], hi.data(), hi.size() * 8, hipMemcpyHostToDevice); hipMemcpy(mask[k], hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    }
    float* out; hipMalloc(&out, B * D * 4);
    std::vector<float> keep;
    run<1>("one block per 64 samples, whole table", ids, mask, table, rows, B, out, &keep);
    run<2>("XCD-partitioned, 2 row ranges", ids, mask, table, rows, B, out, &keep);
    run<4>("XCD-partitioned, 4 row ranges", ids, mask, table, rows, B, out, &keep);
    run<8>("XCD-partitioned, 8 row ranges", ids, mask, table, rows, B, out, &keep);
    return 0;
}
