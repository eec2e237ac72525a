/* CPU ORACLE (plain C + OpenMP) -- TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH.
 *
 * Restatement of the reference's hot path for the timed CPU baseline of bench.py and for parity
 * checks at sizes where the numpy oracle is slow.  Same functions as oracle/ref_np.py (which is
 * pinned against golden vectors captured from the reference); tests/test_oracle_golden.py checks this
 * file against the same goldens.  Reference (paths relative to /root/reference):
 *   get_embeddings_from_batch / get_feature_embedding / array_feature_pooling
 *                                   src/model/BaseModel/base_model.py:262-308
 *   FMModel.forward (pre-sigmoid)   src/model/sort/fm/model.py:18-26, 48-59
 *   DCNLayer / DCNNet               src/model/sort/dcn/dcn_arch.py:14-30, 63-70  (algebraic form)
 * Parallelisation: over samples (what ATen's CPU kernels do for the reference's gather / cat).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>

enum { O_SPARSE = 0, O_DENSE = 1, O_BAG_MASKED_MEAN = 2, O_BAG_MEAN = 3 };

typedef struct {
    const float* table;   /* [rows, dim] */
    const int64_t* index; /* [B] or [B, bag_len]; O_DENSE: const double* values */
    const float* weight;  /* [B, bag_len] or NULL */
    int64_t rows;
    int32_t dim, bag_len, kind, out_col;
} oracle_feature_t;

int oracle_threads(void) { return omp_get_max_threads(); }
void oracle_set_threads(int n) { omp_set_num_threads(n); }

/* returns the number of out-of-range ids (reference: IndexError), 0 on success */
int64_t oracle_embed_concat(const oracle_feature_t* f, int32_t n, int64_t B, float* out, int64_t ld) {
    int64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int64_t b = 0; b < B; ++b) {
        float* o = out + b * ld;
        for (int32_t i = 0; i < n; ++i) {
            const oracle_feature_t* s = &f[i];
            float* dst = o + s->out_col;
            const int D = s->dim;
            if (s->kind == O_DENSE) {
                dst[0] = (float)((const double*)s->index)[b];
            } else if (s->kind == O_SPARSE) {
                int64_t id = s->index[b];
                if (id < 0 || id >= s->rows) { ++bad; id = 0; }
                memcpy(dst, s->table + id * D, sizeof(float) * D);
            } else {
                const int L = s->bag_len;
                float den = 0.f;
                for (int k = 0; k < D; ++k) dst[k] = 0.f;
                for (int l = 0; l < L; ++l) {
                    int64_t id = s->index[b * L + l];
                    const float w = s->weight ? s->weight[b * L + l] : 1.0f;
                    if (id < 0 || id >= s->rows) { ++bad; id = 0; }
                    const float* row = s->table + id * D;
                    den += w;
                    for (int k = 0; k < D; ++k) dst[k] += row[k] * w;
                }
                const float d = (s->kind == O_BAG_MASKED_MEAN) ? den + 1e-8f : (float)L;
                for (int k = 0; k < D; ++k) dst[k] /= d;
            }
        }
    }
    return bad;
}

/* feat [B, ld] holding n_fields fields of `dim` columns: col 0 = w, cols 1.. = v */
void oracle_fm_logit(const float* feat, int64_t ld, int32_t n_fields, int32_t dim, int64_t B, float* logit) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* x = feat + b * ld;
        float first = 0.f, second = 0.f;
        for (int f = 0; f < n_fields; ++f) first += x[f * dim];
        for (int k = 1; k < dim; ++k) {
            float s = 0.f, sq = 0.f;
            for (int f = 0; f < n_fields; ++f) {
                const float v = x[f * dim + k];
                s += v;
                sq += v * v;
            }
            second += s * s - sq;
        }
        logit[b] = first + 0.5f * second;
    }
}

/* x_{l+1} = x0 * (x_l . w_l) + b_l + x_l ; w, b: [n_layers, dim]; out may not alias x */
void oracle_dcn_v1(const float* x, int64_t x_ld, int64_t B, int32_t dim, int32_t n_layers,
                   const float* w, const float* bias, float* out, int64_t out_ld) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < B; ++r) {
        const float* x0 = x + r * x_ld;
        float* xl = out + r * out_ld;
        memcpy(xl, x0, sizeof(float) * dim);
        for (int l = 0; l < n_layers; ++l) {
            const float* wl = w + (int64_t)l * dim;
            const float* bl = bias + (int64_t)l * dim;
            float s = 0.f;
            for (int k = 0; k < dim; ++k) s += xl[k] * wl[k];
            for (int k = 0; k < dim; ++k) xl[k] = x0[k] * s + bl[k] + xl[k];
        }
    }
}

/* Exact inner-product top-k: what faiss.IndexFlatIP.search returns as used by
 * src/model/model_utils/TopKSearcher.py:50-84 and DSSM.hit_rate (recall/DSSM/model.py:182-228).  faiss
 * (faiss-cpu, unpinned in the reference's requirements; absent from this image) documents IndexFlatIP as
 * exhaustive search by inner product with results sorted by decreasing score, label -1 / score -FLT_MAX
 * where fewer than k vectors exist; its tie order and fp32 summation order are unspecified.  This
 * restatement fixes both: score = the fp32 fma chain in the kernel's matrix-core order (below), ties toward
 * the lower index.  excl (optional CSR, lists ascending): items a query must not return -- the reference's
 * "search k + len(history), drop history, keep k" (model.py:209-221) yields the same list.          */
void oracle_topk_ip(const float* items, int64_t n_items, int32_t dim, const float* queries, int64_t n_queries,
                    int32_t k, const int64_t* excl_off, const int64_t* excl_items, int64_t* out_idx, float* out_score) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t q = 0; q < n_queries; ++q) {
        int64_t* oi = out_idx + q * k;
        float* os = out_score + q * k;
        for (int j = 0; j < k; ++j) { oi[j] = -1; os[j] = -FLT_MAX; }
        const int64_t e0 = excl_off ? excl_off[q] : 0, e1 = excl_off ? excl_off[q + 1] : 0;
        int64_t ep = e0;
        const float* qv = queries + q * dim;
        for (int64_t i = 0; i < n_items; ++i) {
            while (ep < e1 && excl_items[ep] < i) ++ep;
            if (ep < e1 && excl_items[ep] == i) continue;
            const float* v = items + i * dim;
            /* the kernel's matrix-core order: P = dim padded to 8, 16, 32, 64 or 128, H = P/2; per step j the
             * MFMA fuses element j then element H+j (pads are zeros) */
            int H = 4;                       /* half of the padded width: 4, 8, 16, 32 or 64 (power of two >= dim/2) */
            while (2 * H < dim) H *= 2;
            float a = 0.f;
            for (int j = 0; j < H; ++j) {
                a = fmaf(j < dim ? v[j] : 0.f, j < dim ? qv[j] : 0.f, a);
                a = fmaf(H + j < dim ? v[H + j] : 0.f, H + j < dim ? qv[H + j] : 0.f, a);
            }
            if (!(a > os[k - 1])) continue;
            int j = k - 1;
            while (j > 0 && a > os[j - 1]) { os[j] = os[j - 1]; oi[j] = oi[j - 1]; --j; }
            os[j] = a; oi[j] = i;
        }
    }
}

/* One DCN-v2 cross layer, out = act(x0 * (x_l W^T + b) + x_l)  (DCNv2Layer.forward + ReLU,
 * src/model/sort/dcn/dcn_arch.py:39-50,78-81) in the exact fp32 order of the matrix-core kernel: the dot
 * product is one fused-multiply-add chain over k ascending (v_mfma_f32_32x32x2_f32 is bit-for-bit
 * fma(a1,b1, fma(a0,b0,c)): profiles/r01_mfma_f32_semantics.txt), then fma(x0, lin + b, x_l).  The numpy
 * restatement (ref_np.dcn_v2, BLAS order) stays the tolerance reference; this one pins the kernel bitwise. */
void oracle_dcn_v2_layer(const float* x0, const float* xl, int64_t ld, int64_t B, int32_t D, const float* W,
                         const float* bias, int32_t relu, float* out, int64_t out_ld) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* xr = xl + b * ld;
        for (int n = 0; n < D; ++n) {
            const float* wr = W + (int64_t)n * D;
            float acc = 0.f;
            for (int k = 0; k < D; ++k) acc = fmaf(xr[k], wr[k], acc);
            float v = fmaf(x0[b * ld + n], acc + bias[n], xr[n]);
            if (relu && !(v > 0.f)) v = 0.f;
            out[b * out_ld + n] = v;
        }
    }
}
