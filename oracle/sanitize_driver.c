/* ASan + UBSan run of the C oracle (TEST INFRASTRUCTURE): every function of oracle/nrx_oracle.c on exact-size heap buffers, incl.
 * the edge cases the reference's path has -- empty batch, all-masked bags, ids at the table's last row, out-of-range ids
 * (counted, read as row 0), k larger than the item count, exclusion lists, a K that is no multiple of the unroll.
 * Built and run by tests/test_sanitizers.py (`make -C oracle sanitize`): any out-of-bounds access, signed overflow or misaligned
 * access aborts the run. */
#include <stdio.h>
#include <stdlib.h>

#include "nrx_oracle.c"

static float* fbuf(size_t n, unsigned seed) {
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));
    for (size_t i = 0; i < n; ++i) { seed = seed * 1664525u + 1013904223u; p[i] = (float)((int)(seed >> 9) % 2001 - 1000) / 500.0f; }
    return p;
}

int main(void) {
    enum { B = 37, D = 16, L = 5, ROWS = 11 };
    /* embed_concat: sparse + dense + masked-mean bag (one sample all-masked) + mean bag; one out-of-range id */
    float* t1 = fbuf((size_t)ROWS * D, 1);
    float* t2 = fbuf((size_t)ROWS * 8, 2);
    int64_t* ids = (int64_t*)malloc(B * sizeof(int64_t));
    int64_t* bag = (int64_t*)malloc((size_t)B * L * sizeof(int64_t));
    float* mask = (float*)malloc((size_t)B * L * sizeof(float));
    double* dense = (double*)malloc(B * sizeof(double));
    for (int b = 0; b < B; ++b) {
        ids[b] = b % ROWS;
        dense[b] = 0.25 * b;
        for (int l = 0; l < L; ++l) { bag[b * L + l] = (b + l) % ROWS; mask[b * L + l] = (b == 3) ? 0.f : (float)((b + l) % 2); }
    }
    ids[B - 1] = ROWS - 1;
    ids[5] = ROWS + 100;       /* out of range: counted, row 0 */
    ids[6] = -4;
    oracle_feature_t f[4] = {{t1, ids, NULL, ROWS, D, 0, O_SPARSE, 0},
                             {NULL, (const int64_t*)dense, NULL, 0, 1, 0, O_DENSE, D},
                             {t2, bag, mask, ROWS, 8, L, O_BAG_MASKED_MEAN, D + 1},
                             {t2, bag, NULL, ROWS, 8, L, O_BAG_MEAN, D + 9}};
    const int W = D + 1 + 8 + 8;
    float* out = (float*)malloc((size_t)B * W * sizeof(float));
    const int64_t bad = oracle_embed_concat(f, 4, B, out, W);
    if (bad != 2) { fprintf(stderr, "embed_concat: %lld bad ids, expected 2\n", (long long)bad); return 1; }
    for (int k = 0; k < 8; ++k)
        if (out[3 * W + D + 1 + k] != 0.f) { fprintf(stderr, "all-masked bag must pool to exact zero\n"); return 1; }
    if (oracle_embed_concat(f, 4, 0, out, W) != 0) return 1;                                  /* empty batch */
    /* FM on a [B, F * D] concat */
    float* feat = fbuf((size_t)B * 3 * D, 3);
    float* logit = (float*)malloc(B * sizeof(float));
    oracle_fm_logit(feat, 3 * D, 3, D, B, logit);
    oracle_fm_logit(feat, 3 * D, 3, D, 0, logit);
    /* DCN v1 (3 layers) and v2 (K = 20: not a multiple of 8) */
    float* w = fbuf(3 * 20, 4);
    float* bb = fbuf(3 * 20, 5);
    float* x = fbuf((size_t)B * 20, 6);
    float* y = (float*)malloc((size_t)B * 20 * sizeof(float));
    oracle_dcn_v1(x, 20, B, 20, 3, w, bb, y, 20);
    float* Wm = fbuf(20 * 20, 7);
    oracle_dcn_v2_layer(x, x, 20, B, 20, Wm, bb, 1, y, 20);
    oracle_dcn_v2_layer(x, y, 20, B, 20, Wm, bb, 0, y, 20 == 20 ? 20 : 0);
    /* top-k: k > n_items, exclusions, zero queries */
    enum { NI = 7, NQ = 5, K = 10 };
    float* items = fbuf(NI * 8, 8);
    float* qs = fbuf(NQ * 8, 9);
    int64_t excl_off[NQ + 1] = {0, 0, 2, 2, 5, 7};
    int64_t excl[7] = {1, 3, 0, 2, 6, 4, 5};
    int64_t* oi = (int64_t*)malloc(NQ * K * sizeof(int64_t));
    float* os = (float*)malloc(NQ * K * sizeof(float));
    oracle_topk_ip(items, NI, 8, qs, NQ, K, excl_off, excl, oi, os);
    oracle_topk_ip(items, NI, 8, qs, NQ, 3, NULL, NULL, oi, os);
    oracle_topk_ip(items, 0, 8, qs, NQ, 3, NULL, NULL, oi, os);
    oracle_topk_ip(items, NI, 8, qs, 0, 3, NULL, NULL, oi, os);
    free(t1); free(t2); free(ids); free(bag); free(mask); free(dense); free(out); free(feat); free(logit); free(w); free(bb); free(x); free(y);
    free(Wm); free(items); free(qs); free(oi); free(os);
    puts("oracle sanitize driver: OK");
    return 0;
}
