// Dev probe: what does v_mfma_f32_32x32x2_f32 compute, bit for bit?  D = C + A[:,0]*B[0,:] + A[:,1]*B[1,:].
// Compares the device result with candidate host evaluations (sequential fmaf chains in either order,
// exactly-rounded sum, unfused products).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_f32_probe.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void probe(const float* a, const float* b, const float* c, float* d, int chain) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        acc[r] = c[row * 32 + (lane & 31)];
    }
    for (int s = 0; s < chain; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s * 64) + lane], b[(s * 64) + lane], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        d[row * 32 + (lane & 31)] = acc[r];
    }
}

static float rnd(int spread) {
    const float m = (float)rand() / RAND_MAX * 2.f - 1.f;
    return ldexpf(m, rand() % (2 * spread + 1) - spread);
}

int main() {
    const int chain = 8;
    float *a, *b, *c, *d;
    hipMallocManaged(&a, chain * 64 * 4); hipMallocManaged(&b, chain * 64 * 4);
    hipMallocManaged(&c, 1024 * 4); hipMallocManaged(&d, 1024 * 4);
    for (int spread : {0, 3, 12}) {
        long bad[5] = {0, 0, 0, 0, 0}, total = 0;
        for (int trial = 0; trial < 200; ++trial) {
            srand(trial * 7 + spread);
            for (int i = 0; i < chain * 64; ++i) { a[i] = rnd(spread); b[i] = rnd(spread); }
            for (int i = 0; i < 1024; ++i) c[i] = trial % 2 ? rnd(spread) : 0.f;
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, c, d, chain);
            hipDeviceSynchronize();
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                float h0 = c[i * 32 + j], h1 = h0, h3 = h0, h4 = h0;
                double h2 = h0;
                for (int s = 0; s < chain; ++s) {
                    const float a0 = a[s * 64 + i], a1 = a[s * 64 + 32 + i], b0 = b[s * 64 + j], b1 = b[s * 64 + 32 + j];
                    h0 = fmaf(a1, b1, fmaf(a0, b0, h0));                       // k = 0 then k = 1, fused
                    h1 = fmaf(a0, b0, fmaf(a1, b1, h1));                       // k = 1 then k = 0, fused
                    h2 = (double)(float)(h2 + (double)a0 * b0 + (double)a1 * b1);   // exact 3-term sum, one rounding
                    h3 = h3 + (float)((double)a0 * b0 + (double)a1 * b1);      // pair exact, then add
                    h4 = (h4 + a0 * b0) + a1 * b1;                             // unfused
                }
                const float got = d[i * 32 + j];
                const float cand[5] = {h0, h1, (float)h2, h3, h4};
                for (int h = 0; h < 5; ++h) bad[h] += memcmp(&got, &cand[h], 4) != 0;
                ++total;
            }
        }
        printf("exponent spread +-%2d: mismatches of %ld  fma(k0,k1)=%ld  fma(k1,k0)=%ld  exact3=%ld  pair-exact=%ld  unfused=%ld\n",
               spread, total, bad[0], bad[1], bad[2], bad[3], bad[4]);
    }
    return 0;
}
