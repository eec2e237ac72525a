// Dev probe: cost of writing the [B, 416] fp32 concat as 64-byte pieces (one 16-float feature row per 4-lane
// group, what the fused gather does) vs 128- / 256-byte contiguous pieces per lane group, and of the same
// store patterns behind a cached gather.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "nrx_embed.h"
typedef float f4 __attribute__((ext_vector_type(4)));
static uint64_t mix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

// G = lanes per contiguous piece (4 -> 64 B, 8 -> 128 B, 16 -> 256 B).  Each wave instruction writes 64/G
// samples; a lane group walks the row in steps of G*16 bytes.
template <int G, bool GATHER>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, const float* __restrict__ tab, const int* __restrict__ ids, int B, int F, int64_t rows) {
    const int lane_in_g = threadIdx.x % G;
    const int64_t sample = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    if (sample >= B) return;
    const int row_f4 = F * 4;                       // float4 per output row
    f4* o = reinterpret_cast<f4*>(out) + sample * row_f4;
    for (int p = lane_in_g; p < row_f4; p += G) {
        f4 v = {1.f, 2.f, 3.f, 4.f};
        if (GATHER) {
            const int f = p >> 2;
            const int id = ids[sample * F + f];
            v = reinterpret_cast<const f4*>(tab)[((int64_t)f * rows + id) * 4 + (p & 3)];
        }
        o[p] = v;
    }
}

// production-like variants of the 16-lane mapping: IDMODE 0 = int32 [B,F]; 1 = int64 [B,F]; 2 = int64 feature-major [F][B]
// PTRS: table base through a per-feature pointer array in LDS; FM: field-sum epilogue + 4-byte output
template <int IDMODE, bool PTRS, bool FM>
__global__ __launch_bounds__(256) void k16(float* __restrict__ out, const float* __restrict__ tab, const void* __restrict__ ids, int B, int F, int64_t rows,
                                           float* __restrict__ fm_out) {
    __shared__ const float* s_tab[64];
    if (PTRS) {
        if (threadIdx.x < F) s_tab[threadIdx.x] = tab + (int64_t)threadIdx.x * rows * 16;
        __syncthreads();
    }
    const int ls = threadIdx.x & 15;
    const int64_t sample = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (sample >= B) return;
    const int row_f4 = F * 4;
    f4* o = reinterpret_cast<f4*>(out) + sample * row_f4;
    f4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    for (int p = ls; p < row_f4; p += 16) {
        const int f = p >> 2;
        int64_t id;
        if (IDMODE == 0) id = reinterpret_cast<const int*>(ids)[sample * F + f];
        else if (IDMODE == 1) id = reinterpret_cast<const int64_t*>(ids)[sample * F + f];
        else id = reinterpret_cast<const int64_t*>(ids)[(int64_t)f * B + sample];
        const f4* base = PTRS ? reinterpret_cast<const f4*>(s_tab[f]) : reinterpret_cast<const f4*>(tab) + (int64_t)f * rows * 4;
        const f4 v = base[id * 4 + (p & 3)];
        o[p] = v;
        if (FM) { s += v; q += v * v; }
    }
    if (FM) {
        for (int off = 4; off < 16; off <<= 1) {
            s.x += __shfl_xor(s.x, off, 64); s.y += __shfl_xor(s.y, off, 64); s.z += __shfl_xor(s.z, off, 64); s.w += __shfl_xor(s.w, off, 64);
        }
        float part = -0.5f * (q.x + q.y + q.z + q.w);
        if ((ls >> 2) == 0) part += 0.5f * (s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w);
        for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (ls == 0) fm_out[sample] = part;
    }
}

// as k16<2,true,true> but pointers / sizes arrive in a 1.5 KB by-value argument block (like the product kernel)
struct BigArgs { const float* table[64]; const int64_t* index[64]; int64_t rows[64]; int n; int B; float* out; float* fm; };
template <bool CHECK>
__global__ __launch_bounds__(256) void k16_args(const BigArgs a) {
    __shared__ const float* s_tab[64];
    __shared__ const int64_t* s_idx[64];
    __shared__ int64_t s_rows[64];
    if (threadIdx.x < a.n) { s_tab[threadIdx.x] = a.table[threadIdx.x]; s_idx[threadIdx.x] = a.index[threadIdx.x]; s_rows[threadIdx.x] = a.rows[threadIdx.x]; }
    __syncthreads();
    const int ls = threadIdx.x & 15;
    const int64_t sample = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (sample >= a.B) return;
    const int row_f4 = a.n * 4;
    f4* o = reinterpret_cast<f4*>(a.out) + sample * row_f4;
    f4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    int bad = 0;
    for (int p = ls; p < row_f4; p += 16) {
        const int f = p >> 2;
        int64_t id = s_idx[f][sample];
        if (CHECK) { const bool oob = (uint64_t)id >= (uint64_t)s_rows[f]; bad |= oob; id = oob ? 0 : id; }
        const f4 v = reinterpret_cast<const f4*>(s_tab[f])[id * 4 + (p & 3)];
        o[p] = v;
        s += v; q += v * v;
    }
    for (int off = 4; off < 16; off <<= 1) {
        s.x += __shfl_xor(s.x, off, 64); s.y += __shfl_xor(s.y, off, 64); s.z += __shfl_xor(s.z, off, 64); s.w += __shfl_xor(s.w, off, 64);
    }
    float part = -0.5f * (q.x + q.y + q.z + q.w) + (float)bad;
    if ((ls >> 2) == 0) part += 0.5f * (s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w);
    for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (ls == 0) a.fm[sample] = part;
}

__global__ void fill_random(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(int)(x & 0xffff) * (1.0f / 65536.0f) - 0.5f;
    }
}

template <typename L>
static void run(const char* name, L launch, double bytes) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(s);
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 20;
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("%-58s %8.1f us  %7.1f GB/s written\n", name, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e9);
}

int main() {
    const int B = 65536, F = 26;
    float *out, *tab; int* ids;
    hipMalloc(&out, (size_t)B * F * 64);
    if (const char* m = getenv("TABLE_ALLOC")) {
        const unsigned flag = !strcmp(m, "uncached") ? hipDeviceMallocUncached : (!strcmp(m, "finegrained") ? hipDeviceMallocFinegrained : hipDeviceMallocDefault);
        hipError_t e = hipExtMallocWithFlags((void**)&tab, (size_t)F * 1048576 * 64, flag);
        printf("table allocation: %s (flag %u) -> %s\n", m, flag, hipGetErrorString(e));
    } else
    hipMalloc(&tab, (size_t)F * 1048576 * 64);
    hipMalloc(&ids, (size_t)B * F * 4);
    int* h = (int*)malloc((size_t)B * F * 4);
    for (int i = 0; i < B * F; ++i) h[i] = (int)((i * 2654435761u) >> 20) & 4095;
    hipMemcpy(ids, h, (size_t)B * F * 4, hipMemcpyHostToDevice);
    if (getenv("RANDOM_TABLE")) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, tab, (size_t)F * 1048576 * 16); hipDeviceSynchronize(); printf("tables filled with random values\n"); }
    else hipMemset(tab, 0, (size_t)F * 1048576 * 64);
    int* idsL; hipMalloc(&idsL, (size_t)B * F * 4);
    for (int i = 0; i < B * F; ++i) h[i] = (int)(((uint64_t)i * 0x9E3779B97F4A7C15ull) >> 44) & 1048575;
    hipMemcpy(idsL, h, (size_t)B * F * 4, hipMemcpyHostToDevice);
    const double bytes = (double)B * F * 64;
#define RUN(G_, GA_, name) run(name, [&]() { hipLaunchKernelGGL((k<G_, GA_>), dim3((unsigned)(((int64_t)B * G_ + 255) / 256)), dim3(256), 0, 0, out, tab, IDS, B, F, ROWS); }, bytes)
    const int* IDS = ids; int64_t ROWS = 4096;
    RUN(4, false, "store only, 64-byte pieces (4 lanes per piece)");
    RUN(8, false, "store only, 128-byte pieces (8 lanes per piece)");
    RUN(16, false, "store only, 256-byte pieces (16 lanes per piece)");
    RUN(64, false, "store only, 1-KiB pieces (whole wave contiguous)");
    RUN(4, true, "cached gather (4k-row tables) + 64-byte pieces");
    RUN(8, true, "cached gather (4k-row tables) + 128-byte pieces");
    RUN(16, true, "cached gather (4k-row tables) + 256-byte pieces");
    {
        int64_t *i64bf, *i64fb; float* fmo;
        hipMalloc(&i64bf, (size_t)B * F * 8); hipMalloc(&i64fb, (size_t)B * F * 8); hipMalloc(&fmo, (size_t)B * 4);
        int64_t* h8 = (int64_t*)malloc((size_t)B * F * 8);
        for (int pass = 0; pass < 2; ++pass) {
            const int64_t R = pass == 0 ? 4096 : 1048576;
            for (int b = 0; b < B; ++b) for (int f = 0; f < F; ++f) h8[(size_t)b * F + f] = (int64_t)(mix64((uint64_t)b * F + f) % (uint64_t)R);
            hipMemcpy(i64bf, h8, (size_t)B * F * 8, hipMemcpyHostToDevice);
            for (int b = 0; b < B; ++b) for (int f = 0; f < F; ++f) h8[(size_t)f * B + b] = (int64_t)(mix64((uint64_t)b * F + f) % (uint64_t)R);
            hipMemcpy(i64fb, h8, (size_t)B * F * 8, hipMemcpyHostToDevice);
            printf("--- 16-lane mapping, production-like variants, %lld-row tables\n", (long long)R);
#define RUN16(M_, P_, FM_, IDP, name) run(name, [&]() { hipLaunchKernelGGL((k16<M_, P_, FM_>), dim3((unsigned)(((int64_t)B * 16 + 255) / 256)), dim3(256), 0, 0, out, tab, (const void*)(IDP), B, F, R, fmo); }, bytes)
            RUN16(1, false, false, i64bf, "int64 ids [B,F]");
            RUN16(2, false, false, i64fb, "int64 ids feature-major [F][B]");
            RUN16(2, true, false, i64fb, "  + table pointers via LDS");
            RUN16(2, true, true, i64fb, "  + FM epilogue");
            BigArgs ba; ba.n = F; ba.B = B; ba.out = out; ba.fm = fmo;
            static float* sep[64]; static int64_t* sepi[64];
            const bool split = getenv("TABLE_SPLIT") != nullptr;
            for (int f = 0; f < F; ++f) {
                ba.table[f] = tab + (int64_t)f * R * 16; ba.index[f] = i64fb + (int64_t)f * B; ba.rows[f] = R;
                if (split) {            // one allocation per table / id array, like 26 torch tensors
                    if (sep[f]) { hipFree(sep[f]); hipFree(sepi[f]); }
                    hipMalloc(&sep[f], (size_t)R * 64); hipMalloc(&sepi[f], (size_t)B * 8);
                    hipMemcpy(sep[f], ba.table[f], (size_t)R * 64, hipMemcpyDeviceToDevice);
                    hipMemcpy(sepi[f], ba.index[f], (size_t)B * 8, hipMemcpyDeviceToDevice);
                    ba.table[f] = sep[f]; ba.index[f] = sepi[f];
                }
            }
            if (split) printf("(one hipMalloc per table and per id array)\n");
            run("  + pointers/sizes in a by-value argument block", [&]() { hipLaunchKernelGGL((k16_args<false>), dim3((unsigned)(((int64_t)B * 16 + 255) / 256)), dim3(256), 0, 0, ba); }, bytes);
            {
                nrx_feature_t fe[64];
                int32_t* status; hipMalloc(&status, 16); hipMemset(status, 0, 16);
                for (int f = 0; f < F; ++f) {
                    fe[f].table = ba.table[f]; fe[f].index = ba.index[f]; fe[f].weight = nullptr; fe[f].rows = R;
                    fe[f].dim = 16; fe[f].bag_len = 0; fe[f].kind = NRX_SPARSE; fe[f].index_bits = 64; fe[f].out_col = 16 * f; fe[f].wide_col = -1;
                    fe[f].fm_field = 1; fe[f].flags = 0;
                }
                run("PRODUCT nrx_embed_fwd (same buffers, through the C-ABI)", [&]() {
                    if (nrx_embed_fwd(fe, F, B, out, 16 * F, nullptr, 0, fmo, status, nullptr) != 0) { printf("nrx error: %s\n", nrx_last_error()); exit(1); } }, bytes);
            }
            run("  + out-of-range check", [&]() { hipLaunchKernelGGL((k16_args<true>), dim3((unsigned)(((int64_t)B * 16 + 255) / 256)), dim3(256), 0, 0, ba); }, bytes);
        }
    }
    IDS = idsL; ROWS = 1048576;
    RUN(4, true, "DRAM gather (1M-row tables) + 64-byte pieces");
    RUN(8, true, "DRAM gather (1M-row tables) + 128-byte pieces");
    RUN(16, true, "DRAM gather (1M-row tables) + 256-byte pieces");
    return 0;
}
