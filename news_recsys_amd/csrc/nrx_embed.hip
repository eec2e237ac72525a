// Fused multi-table embedding gather (+ bag pooling) -> concat, with optional Wide&Deep column
// routing and FM epilogue, and its dense-grad backward.  gfx950 / wave64.
//
// Replaces, per launch, the reference's Python loop over features
//   BaseModel.get_embeddings_from_batch   src/model/BaseModel/base_model.py:284-308
//     get_feature_embedding               :262-271   (one ATen gather per feature)
//     array_feature_pooling               :273-282   (mul, sum, sum, add, div over [B,L,D])
//     torch.cat                           :308
//   WideDeep.get_inp_embedding            src/model/sort/widedeep/model.py:53-69
//   FM.get_inp_embedding + FMModel.forward (pre-sigmoid) src/model/sort/fm/model.py:18-26,48-59
//
// Work decomposition ("row-group" mapping): a sample is owned by Q = 2^QLOG2 adjacent lanes, lane q
// holding columns [4q, 4q+4) of whichever feature is being processed; a 256-thread block owns
// TB = 256/Q consecutive samples and walks the features.  Consequences:
//   * a D-float row is fetched by D/4 adjacent lanes with one 16 B load each (coalesced 64..256 B
//     segments of HBM), and lands in its final place in the [B, sum D] concat -- no torch.cat pass;
//   * single-valued ids are read coalesced over the block's samples (8 B x TB contiguous per feature);
//   * bag ids/weights ([B, L] padded, reference layout) are staged per block through LDS as packed
//     {int32 id, f32 weight} pairs, read back as broadcast ds_read_b64;
//   * the FM sums over fields stay in registers across the feature walk; the final reduction over
//     the Q lanes of a sample is a wavefront shuffle (Q <= 64 lanes never straddle a wave).
// The uniform kernel (nrx_embed_ring.h) is the same mapping specialised for "all features single-valued, same D = 4Q":
// ids staged once per block in LDS, then a ring of R row loads per lane kept in flight across the feature walk.
#include "nrx_common.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "nrx_embed_ring.h"   // UniformArgs, fm_accumulate, group_sum, embed_fwd_ring

bool nrx_launch_uniform_wide(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, float* out, int64_t out_ld,
                             float* wide_out, int64_t wide_ld, int32_t* status, hipStream_t st);   // nrx_embed_wide.hip

namespace {

struct EmbedArgs {
    FeatDev f[NRX_MAX_FEATURES];
    int64_t batch;
    float* out;          // fwd: concat out (may be null); bwd: g_out (read only)
    int64_t out_ld;
    float* wide;         // fwd: wide_out; bwd: g_wide (read only)
    int64_t wide_ld;
    float* fm_out;
    float* fm_sums;      // fwd (optional): [batch, sums_ld] field sums of the FM epilogue; bwd: the same tensor (read only)
    int64_t sums_ld;
    const float* g_fm;   // bwd (optional): dL/d fm_out [batch]
    const float* feat;   // bwd (with g_fm): the forward concat [batch, feat_ld]
    int64_t feat_ld;
    int32_t* status;
    int32_t n;
    int32_t lds_chunk;   // bag entries staged per pass
    uint8_t feat_id[NRX_MAX_FEATURES];   // fwd: the feature's index in the caller's list (what an out-of-range report names)
};
static_assert(sizeof(EmbedArgs) <= 3584, "kernarg budget");

struct BagPair {
    int32_t id;
    float w;
};

__device__ __forceinline__ float4 load_row4(const float* table, int64_t id, int D, int k0, bool vec) {
    const float* p = table + id * (int64_t)D + k0;
    float4 v;
    if (vec) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        v.x = p[0];
        v.y = (k0 + 1 < D) ? p[1] : 0.f;
        v.z = (k0 + 2 < D) ? p[2] : 0.f;
        v.w = (k0 + 3 < D) ? p[3] : 0.f;
    }
    return v;
}

// Entry `pos` of sample b's bag as the raw (id, weight) the pooling uses.  Padded form: ids [B, L] + optional weights.
// CSR form (NRX_FEAT_BAG_CSR): f.weight holds int64 offsets [B + 1]; real entries weigh 1, the positions past the bag's
// end are what DataReader pads with -- id 0, mask 0 (weight 1 for NRX_BAG_MEAN, whose padded form has no mask).
// s_off (CSR only): the offsets of the block's samples, staged in LDS by the caller (s_off[s] = offsets[b0 + s], s <= nb) --
// read from memory per entry they put a second dependent load in front of every id load (CSR 54 us vs padded 51 us at
// the C4 shape; with the LDS copy the CSR form is the faster one).
__device__ __forceinline__ void bag_entry(const FeatDev& f, int64_t b, int s, const int64_t* s_off, int pos, int L, int64_t& id, float& w) {
    if (f.flags & NRX_FEAT_BAG_CSR) {
        const int64_t o0 = s_off[s];
        if ((int64_t)pos < s_off[s + 1] - o0) {
            id = nrx_load_id(f.index, o0 + pos, f.idx64);
            w = 1.0f;
        } else {
            id = 0;
            w = f.kind == NRX_BAG_MEAN ? 1.0f : 0.f;
        }
    } else {
        const int64_t gi = b * (int64_t)L + pos;
        id = nrx_load_id(f.index, gi, f.idx64);
        w = f.weight ? f.weight[gi] : 1.0f;
    }
}

// --------------------------------------------------------------------------------------------
// Generic forward: any mix of sparse / dense / bag features, any dims, wide routing, FM.
// --------------------------------------------------------------------------------------------
template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void embed_fwd_generic(const EmbedArgs a) {
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BagPair* s_bag = reinterpret_cast<BagPair*>(smem);

    const int tid = threadIdx.x;
    const int q = tid & (Q - 1);
    const int sb = tid >> QLOG2;
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int64_t b = b0 + sb;
    const bool live = b < a.batch;
    const int nb = (int)((a.batch - b0) < (int64_t)TB ? (a.batch - b0) : (int64_t)TB);
    const bool out_vec = a.out != nullptr && ((a.out_ld & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(a.out) & 15u) == 0);

    float fm_first = 0.f;
    float4 fm_s = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 fm_q = make_float4(0.f, 0.f, 0.f, 0.f);

    // The first few single-valued features of one chunk (dim <= 4Q, 16-byte-aligned table) are fetched UP FRONT -- id, then row -- so that their
    // two dependent round trips run under the first bag feature's staging instead of in front of / behind it (a DSSM tower: user id + history
    // bag + item id).  Which features those are is wave-uniform (scalar loads of the descriptors); the values wait in registers.
    constexpr int NPRE = 4;
    int pre_feat[NPRE];
    float4 pre_row[NPRE];
    {
        int ns = 0;
#pragma unroll
        for (int s2 = 0; s2 < NPRE; ++s2) pre_feat[s2] = -1;
        for (int fi = 0; fi < a.n && ns < NPRE; ++fi) {
            const FeatDev& f = a.f[fi];
            if (f.kind == NRX_SPARSE && f.dim <= 4 * Q && (f.dim & 3) == 0 && (reinterpret_cast<uintptr_t>(f.table) & 15u) == 0) {
#pragma unroll
                for (int s2 = 0; s2 < NPRE; ++s2)
                    if (s2 == ns) pre_feat[s2] = fi;
                ++ns;
            }
        }
        int64_t pid[NPRE];
#pragma unroll
        for (int s2 = 0; s2 < NPRE; ++s2) {
            const int fi = pre_feat[s2] >= 0 ? pre_feat[s2] : 0;
            pid[s2] = (pre_feat[s2] >= 0 && live) ? nrx_load_id(a.f[fi].index, b, a.f[fi].idx64) : 0;
        }
#pragma unroll
        for (int s2 = 0; s2 < NPRE; ++s2) {
            pre_row[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pre_feat[s2] >= 0) {
                const FeatDev& f = a.f[pre_feat[s2]];
                if ((uint64_t)pid[s2] >= (uint64_t)f.rows) {
                    if (q == 0 && live) nrx_report_oob(a.status, a.feat_id[pre_feat[s2]], b, pid[s2]);
                    pid[s2] = 0;
                }
                if (live && q * 4 < f.dim) pre_row[s2] = *reinterpret_cast<const float4*>(f.table + pid[s2] * (int64_t)f.dim + q * 4);
            }
        }
    }

    for (int fi = 0; fi < a.n; ++fi) {
        const FeatDev& f = a.f[fi];
        const int D = f.dim;
        const bool vec_load = ((D & 3) == 0) && ((reinterpret_cast<uintptr_t>(f.table) & 15u) == 0);
        for (int kc = 0; kc < D; kc += 4 * Q) {
            const int k0 = kc + q * 4;
            const bool active = live && k0 < D;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool pre = false;
#pragma unroll
            for (int s2 = 0; s2 < NPRE; ++s2)
                if (fi == pre_feat[s2]) { v = pre_row[s2]; pre = true; }          // wave-uniform
            if (pre) {
                // fetched up front
            } else if (f.kind == NRX_SPARSE) {
                if (active) {
                    int64_t id = nrx_load_id(f.index, b, f.idx64);
                    if ((uint64_t)id >= (uint64_t)f.rows) {
                        if (q == 0) nrx_report_oob(a.status, a.feat_id[fi], b, id);
                        id = 0;
                    }
                    v = load_row4(f.table, id, D, k0, vec_load);
                }
            } else if (f.kind == NRX_DENSE) {
                if (active && k0 == 0)
                    v.x = f.idx64 ? (float)reinterpret_cast<const double*>(f.index)[b]
                                  : reinterpret_cast<const float*>(f.index)[b];
            } else {
                // ---- bag: stage {id, weight} of the block's samples through LDS, chunk by chunk
                const int L = f.bag_len;
                const int lc = a.lds_chunk;
                const int stride = lc | 1;   // odd pair-stride: conflict-free broadcast ds_read_b64
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                float den = 0.f;
                int64_t* s_off = reinterpret_cast<int64_t*>(s_bag + TB * stride);
                if (f.flags & NRX_FEAT_BAG_CSR) {
                    __syncthreads();            // (the previous feature's readers are done with the LDS area)
                    for (int i = tid; i <= nb; i += NRX_BLOCK) s_off[i] = reinterpret_cast<const int64_t*>(f.weight)[b0 + i];   // nb + 1 entries: nb can equal NRX_BLOCK
                }
                for (int l0 = 0; l0 < L; l0 += lc) {
                    const int cur = (L - l0) < lc ? (L - l0) : lc;
                    __syncthreads();
                    int e_first = tid;
                    if (!(f.flags & NRX_FEAT_BAG_CSR)) {
                        // padded form: four entries per thread and round, their id / weight loads issued together (one entry per round put
                        // ~12 dependent round trips in front of the block's first row load at L = 50)
                        constexpr int SU = 4;
                        const int n_e = nb * cur;
                        const bool has_w = f.weight != nullptr;
                        for (; e_first < n_e; e_first += SU * NRX_BLOCK) {
                            int64_t id4[SU];
                            float w4[SU];
                            int pos4[SU];
#pragma unroll
                            for (int u = 0; u < SU; ++u) {
                                const int e = e_first + u * NRX_BLOCK;
                                const int ec = e < n_e ? e : n_e - 1;
                                const int s = ec / cur;
                                const int l = ec - s * cur;
                                const int64_t gi = (b0 + s) * (int64_t)L + l0 + l;
                                pos4[u] = s * stride + l;
                                id4[u] = nrx_load_id(f.index, gi, f.idx64);
                                w4[u] = has_w ? f.weight[gi] : 1.0f;
                            }
#pragma unroll
                            for (int u = 0; u < SU; ++u) {
                                const int e = e_first + u * NRX_BLOCK;
                                if (e < n_e) {
                                    int64_t id = id4[u];
                                    if ((uint64_t)id >= (uint64_t)f.rows) {
                                        nrx_report_oob(a.status, a.feat_id[fi], b0 + e / cur, id);
                                        id = 0;
                                    }
                                    BagPair p;
                                    p.id = (int32_t)id;
                                    p.w = w4[u];
                                    s_bag[pos4[u]] = p;
                                }
                            }
                        }
                        e_first = nb * cur;          // (nothing left for the per-entry loop below)
                    } else if (s_off[nb] > s_off[0]) {
                        // CSR form, same batching: positions past a bag's end read the block's first value (there is one) and become {0, pad weight}
                        constexpr int SU = 4;
                        const int n_e = nb * cur;
                        const int64_t blk_first = s_off[0];
                        const float pad_w = f.kind == NRX_BAG_MEAN ? 1.0f : 0.f;
                        for (; e_first < n_e; e_first += SU * NRX_BLOCK) {
                            int64_t id4[SU];
                            bool in4[SU];
                            int pos4[SU];
#pragma unroll
                            for (int u = 0; u < SU; ++u) {
                                const int e = e_first + u * NRX_BLOCK;
                                const int ec = e < n_e ? e : n_e - 1;
                                const int s = ec / cur;
                                const int l = ec - s * cur;
                                const int64_t o0 = s_off[s];
                                in4[u] = (int64_t)(l0 + l) < s_off[s + 1] - o0;
                                pos4[u] = s * stride + l;
                                id4[u] = nrx_load_id(f.index, in4[u] ? o0 + l0 + l : blk_first, f.idx64);
                            }
#pragma unroll
                            for (int u = 0; u < SU; ++u) {
                                const int e = e_first + u * NRX_BLOCK;
                                if (e < n_e) {
                                    int64_t id = in4[u] ? id4[u] : 0;
                                    if ((uint64_t)id >= (uint64_t)f.rows) {
                                        nrx_report_oob(a.status, a.feat_id[fi], b0 + e / cur, id);
                                        id = 0;
                                    }
                                    BagPair p;
                                    p.id = (int32_t)id;
                                    p.w = in4[u] ? 1.0f : pad_w;
                                    s_bag[pos4[u]] = p;
                                }
                            }
                        }
                        e_first = nb * cur;
                    }
                    for (int e = e_first; e < nb * cur; e += NRX_BLOCK) {
                        const int s = e / cur;
                        const int l = e - s * cur;
                        int64_t id;
                        float w;
                        bag_entry(f, b0 + s, s, s_off, l0 + l, L, id, w);
                        if ((uint64_t)id >= (uint64_t)f.rows) {
                            nrx_report_oob(a.status, a.feat_id[fi], b0 + s, id);
                            id = 0;
                        }
                        BagPair p;
                        p.id = (int32_t)id;
                        p.w = w;
                        s_bag[s * stride + l] = p;
                    }
                    __syncthreads();
                    if (active) {
                        const BagPair* row = s_bag + sb * stride;
                        constexpr int U = 8;
                        int l = 0;
                        // 16-byte-aligned tables (every reference shape): the U row loads of a pass are issued unconditionally -- an entry with
                        // weight 0 reads row 0 and its value is replaced by zeros where it is USED.  A load behind `if (w != 0)` is a branch around
                        // a load: the compiler then waits vmcnt(0) in front of every one of them and the eight loads go out one behind the other
                        // (seen in the ISA); so does the aligned / unaligned choice inside load_row4 when it is made per load.
                        if (vec_load) {
                            for (; l + U <= cur; l += U) {
                                BagPair p[U];
                                float4 r[U];
#pragma unroll
                                for (int u = 0; u < U; ++u) p[u] = row[l + u];
#pragma unroll
                                for (int u = 0; u < U; ++u)
                                    r[u] = *reinterpret_cast<const float4*>(f.table + (int64_t)(p[u].w != 0.f ? p[u].id : 0) * (int64_t)D + k0);
#pragma unroll
                                for (int u = 0; u < U; ++u) {
#pragma clang fp contract(off)
                                    const bool on = p[u].w != 0.f;
                                    const float rx = on ? r[u].x : 0.f, ry = on ? r[u].y : 0.f, rz = on ? r[u].z : 0.f, rw = on ? r[u].w : 0.f;
                                    den += p[u].w;
                                    acc.x += rx * p[u].w;
                                    acc.y += ry * p[u].w;
                                    acc.z += rz * p[u].w;
                                    acc.w += rw * p[u].w;
                                }
                            }
                        }
                        for (; l + U <= cur; l += U) {
                            BagPair p[U];
                            float4 r[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) p[u] = row[l + u];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (p[u].w != 0.f) r[u] = load_row4(f.table, p[u].id, D, k0, vec_load);
                            }
#pragma unroll
                            for (int u = 0; u < U; ++u) {
#pragma clang fp contract(off)
                                den += p[u].w;
                                acc.x += r[u].x * p[u].w;
                                acc.y += r[u].y * p[u].w;
                                acc.z += r[u].z * p[u].w;
                                acc.w += r[u].w * p[u].w;
                            }
                        }
                        for (; l < cur; ++l) {
#pragma clang fp contract(off)
                            const BagPair p = row[l];
                            den += p.w;
                            if (p.w != 0.f) {
                                const float4 r = load_row4(f.table, p.id, D, k0, vec_load);
                                acc.x += r.x * p.w;
                                acc.y += r.y * p.w;
                                acc.z += r.z * p.w;
                                acc.w += r.w * p.w;
                            }
                        }
                    }
                }
                if (f.kind == NRX_BAG_MASKED_MEAN) {
                    const float d = den + 1e-8f;
                    v = make_float4(acc.x / d, acc.y / d, acc.z / d, acc.w / d);
                } else if (f.kind == NRX_BAG_MEAN) {
                    const float d = (float)L;
                    v = make_float4(acc.x / d, acc.y / d, acc.z / d, acc.w / d);
                } else {
                    v = acc;
                }
            }

            if (active) {
                if (f.wide_col >= 0) {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k0 + j;
                        if (k < D) {
                            if (k == 0) {
                                if (a.wide) a.wide[b * a.wide_ld + f.wide_col] = vv[j];
                            } else if (a.out) {
                                a.out[b * a.out_ld + f.out_col + k - 1] = vv[j];
                            }
                        }
                    }
                } else if (a.out) {
                    float* p = a.out + b * a.out_ld + f.out_col + k0;
                    if (out_vec && ((f.out_col & 3) == 0) && (k0 + 4 <= D)) {
                        *reinterpret_cast<float4*>(p) = v;
                    } else {
                        p[0] = v.x;
                        if (k0 + 1 < D) p[1] = v.y;
                        if (k0 + 2 < D) p[2] = v.z;
                        if (k0 + 3 < D) p[3] = v.w;
                    }
                }
                if (f.fm) fm_accumulate(v, k0, D, fm_first, fm_s, fm_q);
            }
        }
    }

    if (a.fm_out != nullptr) {
        // 0.5 * sum_k [(sum_f v)^2 - sum_f v^2] over this lane's 4 columns, then over the Q lanes
        float part = fm_lane_part(fm_s, fm_q, fm_first);
        if (a.fm_sums != nullptr && live) {     // field sums for the FM backward (one chunk: dims <= 4Q here)
            float* sp = a.fm_sums + b * a.sums_ld + 4 * q;
            const float sv[4] = {q == 0 ? fm_first : fm_s.x, fm_s.y, fm_s.z, fm_s.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < a.sums_ld) sp[j] = sv[j];
        }
        part = group_sum<Q>(part);
        if (live && q == 0) a.fm_out[b] = part;
    }
}

// --------------------------------------------------------------------------------------------
// Small batches: ONE BLOCK PER SAMPLE.  The kernels above give a sample to Q lanes and let a block of 256 / Q samples walk
// the features one after the other -- right for B = 65536, but the reference trains with B = 512 (sort/deep/train_cf_deep.yaml:48),
// where that is 8 blocks on a 256-CU chip, each running 26 (or 50 bag entries') dependent load rounds: 12.6 us for the C2
// plan, 21.9 us for the DSSM user tower (profiles/r02_small_batch_kernel_times.txt).  Here every (feature, bag entry, 16-byte
// chunk) of the sample is one work item with its own thread: all of the sample's row fetches are in flight at once and land
// in LDS; a second step forms the outputs from LDS IN THE ORDER the big kernels use -- bag entries summed sequentially in l,
// FM fields accumulated sequentially in plan order by the first Q lanes -- so the result is bit-identical to theirs.
// Eligible: sparse / dense / padded bag features, table dims % 4 == 0 (16-byte aligned tables), no wide routing, no CSR bags.
// --------------------------------------------------------------------------------------------
template <int Q>
__device__ __forceinline__ float group_sum_rt(float v) { return group_sum<Q>(v); }

// static LDS of embed_fwd_small_kernel (its copy of the descriptors), counted by the launcher's 64 KB eligibility test
#define NRX_SMALL_STATIC_LDS (NRX_MAX_FEATURES * sizeof(FeatDev))

__global__ __launch_bounds__(NRX_BLOCK) void embed_fwd_small_kernel(const EmbedArgs args_in_kernarg) {
    const NRX_CONST EmbedArgs* a = nrx_kernarg<EmbedArgs>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = a->n;
    int* s_ibase = reinterpret_cast<int*>(smem);                 // [n + 1] first work item of feature fi
    int* s_obase = s_ibase + (NRX_MAX_FEATURES + 1);             // [n + 1] first output chunk of feature fi
    float4* s_rows = reinterpret_cast<float4*>(s_obase + (NRX_MAX_FEATURES + 1) + 2);     // [items] fetched row chunks (16-byte aligned: 2 x 65 + 2 ints)
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    // the descriptors go to LDS first (one coalesced pass over the kernarg block): a work item picks its feature by a per-thread
    // index, and a per-thread read of the argument block is a global load -- three dependent ones (kind / dim, ids, table) in front of
    // every row fetch
    __shared__ FeatDev s_f[NRX_MAX_FEATURES];
    {
        const NRX_CONST uint32_t* src = reinterpret_cast<const NRX_CONST uint32_t*>(a->f);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_f);
        for (int i = tid; i < n * (int)(sizeof(FeatDev) / 4); i += NRX_BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    // ---- item / output-chunk bases: one wavefront scans the <= 64 features
    if (tid < 64) {
        int items = 0, chunks = 0;
        if (tid < n) {
            const int L = s_f[tid].kind >= NRX_BAG_MASKED_MEAN ? s_f[tid].bag_len : 1;
            chunks = s_f[tid].kind == NRX_DENSE ? 1 : s_f[tid].dim / 4;
            items = L * chunks;
        }
        int ii = items, cc = chunks;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t1 = __shfl_up(ii, off, 64), t2 = __shfl_up(cc, off, 64);
            if (tid >= off) { ii += t1; cc += t2; }
        }
        if (tid < n) { s_ibase[tid + 1] = ii; s_obase[tid + 1] = cc; }
        if (tid == 0) { s_ibase[0] = 0; s_obase[0] = 0; }
        // the FM fields' lane count (they share one dim: the last field's, as the per-feature loop this replaces left it), found here by
        // one ballot instead of by every thread walking the descriptors after the outputs are done
        const unsigned long long fm_mask = __ballot(tid < n && s_f[tid < n ? tid : 0].fm != 0);
        if (tid == 0) s_obase[NRX_MAX_FEATURES + 1] = fm_mask ? s_f[63 - __builtin_clzll(fm_mask)].dim / 4 : 1;
    }
    __syncthreads();
    const int n_items = s_ibase[n], n_out = s_obase[n];
    float* s_wt = reinterpret_cast<float*>(s_rows + n_items);     // [items] the entry's weight (same for the chunks of an entry)
    float4* s_val = reinterpret_cast<float4*>(s_wt + ((n_items + 3) & ~3));      // [n_out] finished output chunks (for the FM epilogue)
    auto feat_of = [&](const int* base, int t) {                  // feature owning index t of a prefix array
        int lo = 0, hi = n;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (base[mid] <= t) lo = mid; else hi = mid;
        }
        return lo;
    };
    // ---- step 1: every work item fetches its row chunk
    for (int t = tid; t < n_items; t += NRX_BLOCK) {
        const int fi = feat_of(s_ibase, t);
        const FeatDev& f = s_f[fi];
        const int r = t - s_ibase[fi];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float w = 1.0f;
        if (f.kind == NRX_DENSE) {
            v.x = f.idx64 ? (float)reinterpret_cast<const double*>(f.index)[b] : reinterpret_cast<const float*>(f.index)[b];
        } else {
            const int C = f.dim / 4;
            const int l = r / C, c = r - l * C;
            const bool bag = f.kind >= NRX_BAG_MASKED_MEAN;
            const int64_t gi = bag ? b * (int64_t)f.bag_len + l : b;
            int64_t id = nrx_load_id(f.index, gi, f.idx64);
            if (bag) w = f.weight ? f.weight[gi] : 1.0f;
            if ((uint64_t)id >= (uint64_t)f.rows) {
                if (c == 0) nrx_report_oob(a->status, a->feat_id[fi], b, id);
                id = 0;
            }
            if (!bag || w != 0.f) v = *reinterpret_cast<const float4*>(f.table + id * (int64_t)f.dim + 4 * c);
        }
        s_rows[t] = v;
        s_wt[t] = w;
    }
    __syncthreads();
    // ---- step 2: one thread per output chunk; bags summed in entry order with the generic kernel's arithmetic
    for (int t = tid; t < n_out; t += NRX_BLOCK) {
        const int fi = feat_of(s_obase, t);
        const FeatDev& f = s_f[fi];
        const int c = t - s_obase[fi];
        const int ib = s_ibase[fi];
        float4 v;
        if (f.kind < NRX_BAG_MASKED_MEAN) {
            v = s_rows[ib + c];
        } else {
            const int C = f.dim / 4, L = f.bag_len;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            float den = 0.f;
            constexpr int UB = 10;                        // LDS reads of UB entries issued ahead of their dependent adds
            int l = 0;
            for (; l + UB <= L; l += UB) {
                float w[UB];
                float4 r[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) { w[u] = s_wt[ib + (l + u) * C + c]; r[u] = s_rows[ib + (l + u) * C + c]; }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
#pragma clang fp contract(off)
                    den += w[u];
                    acc.x += r[u].x * w[u]; acc.y += r[u].y * w[u]; acc.z += r[u].z * w[u]; acc.w += r[u].w * w[u];
                }
            }
            for (; l < L; ++l) {
#pragma clang fp contract(off)
                const float w = s_wt[ib + l * C + c];
                const float4 r = s_rows[ib + l * C + c];
                den += w;
                acc.x += r.x * w; acc.y += r.y * w; acc.z += r.z * w; acc.w += r.w * w;
            }
            if (f.kind == NRX_BAG_MASKED_MEAN) {
                const float d = den + 1e-8f;
                v = make_float4(acc.x / d, acc.y / d, acc.z / d, acc.w / d);
            } else if (f.kind == NRX_BAG_MEAN) {
                const float d = (float)L;
                v = make_float4(acc.x / d, acc.y / d, acc.z / d, acc.w / d);
            } else {
                v = acc;
            }
        }
        s_val[t] = v;
        if (a->out != nullptr) {
            float* p = a->out + b * a->out_ld + f.out_col + 4 * c;
            if (f.kind == NRX_DENSE) p[0] = v.x;
            else if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) *reinterpret_cast<float4*>(p) = v;
            else { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
        }
    }
    if (a->fm_out == nullptr) return;
    __syncthreads();
    // ---- FM epilogue: the first Q lanes walk the fields in plan order, as a sample's Q lanes do in the big kernels
    const int Q = s_obase[NRX_MAX_FEATURES + 1];       // FM fields share one dim (fm/model.py:48-59 stacks them)
    if (tid >= 64) return;
    const int q = tid;
    float fm_first = 0.f;
    float4 fm_s = make_float4(0.f, 0.f, 0.f, 0.f), fm_q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < Q) {
        // eight fields per round: their flags and chunk bases, then their values, are read from LDS together and accumulated in plan order
        // (one field per round was three dependent LDS round trips per field: ~7 us of a 12 us launch at 26 fields)
        constexpr int FU = 8;
        for (int f0 = 0; f0 < n; f0 += FU) {
            bool is_fm[FU];
            int ob[FU], dm[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int fi = f0 + u < n ? f0 + u : n - 1;
                is_fm[u] = f0 + u < n && s_f[fi].fm != 0;
                dm[u] = s_f[fi].dim;
                ob[u] = s_obase[fi];
            }
            float4 v[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) v[u] = s_val[is_fm[u] ? ob[u] + q : 0];
#pragma unroll
            for (int u = 0; u < FU; ++u)
                if (is_fm[u]) fm_accumulate(v[u], 4 * q, dm[u], fm_first, fm_s, fm_q);
        }
    }
    float part = fm_lane_part(fm_s, fm_q, fm_first);
    if (a->fm_sums != nullptr && q < Q) {
        float* sp = a->fm_sums + b * a->sums_ld + 4 * q;
        const float sv[4] = {q == 0 ? fm_first : fm_s.x, fm_s.y, fm_s.z, fm_s.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * q + j < a->sums_ld) sp[j] = sv[j];
    }
    switch (Q) {                                         // the same butterfly as group_sum<Q> of the big kernels
        case 1: break;
        case 2: part = group_sum_rt<2>(part); break;
        case 4: part = group_sum_rt<4>(part); break;
        case 8: part = group_sum_rt<8>(part); break;
        case 16: part = group_sum_rt<16>(part); break;
        case 32: part = group_sum_rt<32>(part); break;
        default: part = group_sum_rt<64>(part); break;
    }
    if (q == 0) a->fm_out[b] = part;
}

// --------------------------------------------------------------------------------------------
// Backward: dense-grad scatter-add (what autograd gives nn.Embedding(sparse=False)).
// f.table is the grad table; a.out = g_out, a.wide = g_wide (both read only here).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_row4(float* gtable, int64_t id, int D, int k0, float4 g) {
    float* p = gtable + id * (int64_t)D + k0;
    unsafeAtomicAdd(p, g.x);
    if (k0 + 1 < D) unsafeAtomicAdd(p + 1, g.y);
    if (k0 + 2 < D) unsafeAtomicAdd(p + 2, g.z);
    if (k0 + 3 < D) unsafeAtomicAdd(p + 3, g.w);
}

template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void embed_bwd_generic(const EmbedArgs a) {
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BagPair* s_bag = reinterpret_cast<BagPair*>(smem);

    const int tid = threadIdx.x;
    const int q = tid & (Q - 1);
    const int sb = tid >> QLOG2;
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int64_t b = b0 + sb;
    const bool live = b < a.batch;
    const int nb = (int)((a.batch - b0) < (int64_t)TB ? (a.batch - b0) : (int64_t)TB);

    // blockIdx.y strides over the features: a small batch is a handful of blocks, and one block walking 26 features is 26 dependent
    // (upstream row, id) -> atomics round trips in a row; the host spreads them over gridDim.y blocks when the batch alone does not fill the chip
    for (int fi = blockIdx.y; fi < a.n; fi += gridDim.y) {
        const FeatDev& f = a.f[fi];
        if (f.kind == NRX_DENSE) continue;
        const int D = f.dim;
        float* gtable = const_cast<float*>(f.table);
        for (int kc = 0; kc < D; kc += 4 * Q) {
            const int k0 = kc + q * 4;
            const bool active = live && k0 < D;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (active) {
                float gg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    if (k < D) {
                        if (f.wide_col >= 0) {
                            if (k == 0) gg[j] = a.wide ? a.wide[b * a.wide_ld + f.wide_col] : 0.f;
                            else gg[j] = a.out ? a.out[b * a.out_ld + f.out_col + k - 1] : 0.f;
                        } else {
                            gg[j] = a.out ? a.out[b * a.out_ld + f.out_col + k] : 0.f;
                            if (f.fm && a.g_fm != nullptr)      // d fm / d field: column 0 -> 1, column k -> S_k - v_k
                                gg[j] += (k == 0) ? a.g_fm[b]
                                                  : a.g_fm[b] * (a.fm_sums[b * a.sums_ld + k] - a.feat[b * a.feat_ld + f.out_col + k]);
                        }
                    }
                }
                g = make_float4(gg[0], gg[1], gg[2], gg[3]);
            }
            if (f.kind == NRX_SPARSE) {
                if (active && blockIdx.z == 0) {
                    const int64_t id = nrx_load_id(f.index, b, f.idx64);
                    if (id >= ((f.flags & NRX_FEAT_ROW0_IS_DATA) ? 0 : 1) && id < f.rows) atomic_add_row4(gtable, id, D, k0, g);
                }
                continue;
            }
            // ---- bags
            const int L = f.bag_len;
            float den = 1.0f;
            if (f.kind == NRX_BAG_MASKED_MEAN) {
                den = 0.f;
                if (active) {
                    if (f.flags & NRX_FEAT_BAG_CSR) {
                        const int64_t* offs = reinterpret_cast<const int64_t*>(f.weight);
                        const int64_t n = offs[b + 1] - offs[b];
                        den = (float)(n < (int64_t)L ? n : (int64_t)L);      // the forward's sum of n ones
                    } else {
                        // eight weights in flight, added in position order (the loop as written waited for every load: L round trips per sample)
                        const float* wp = f.weight + b * (int64_t)L;
                        int l = 0;
                        for (; l + 8 <= L; l += 8) {
                            float w8[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) w8[u] = wp[l + u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) den += w8[u];
                        }
                        for (; l < L; ++l) den += wp[l];
                    }
                }
                den += 1e-8f;
            } else if (f.kind == NRX_BAG_MEAN) {
                den = (float)L;
            }
            const float4 gs = make_float4(g.x / den, g.y / den, g.z / den, g.w / den);
            const int lc = a.lds_chunk;
            const int stride = lc | 1;
            int64_t* s_off = reinterpret_cast<int64_t*>(s_bag + TB * stride);
            if (f.flags & NRX_FEAT_BAG_CSR) {
                __syncthreads();
                for (int i = tid; i <= nb; i += NRX_BLOCK) s_off[i] = reinterpret_cast<const int64_t*>(f.weight)[b0 + i];   // nb + 1 entries: nb can equal NRX_BLOCK
            }
            // blockIdx.z strides over the bag positions (small batches: the host splits a bag's L entries over gridDim.z blocks, each with its own slice
            // [zlo, zhi) of every sample's bag; the denominator above is formed in full by each)
            const int zper = (L + (int)gridDim.z - 1) / (int)gridDim.z;
            const int zlo = (int)blockIdx.z * zper;
            const int zhi = (zlo + zper) < L ? (zlo + zper) : L;
            for (int l0 = zlo; l0 < zhi; l0 += lc) {
                const int cur = (zhi - l0) < lc ? (zhi - l0) : lc;
                __syncthreads();
                int e_first = tid;
                if (!(f.flags & NRX_FEAT_BAG_CSR)) {
                    // padded form: four entries per thread and round, their id / weight loads issued together (as in the forward's staging loop)
                    constexpr int SU = 4;
                    const int n_e = nb * cur;
                    const bool has_w = f.weight != nullptr;
                    for (; e_first < n_e; e_first += SU * NRX_BLOCK) {
                        int64_t id4[SU];
                        float w4[SU];
                        int pos4[SU];
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            const int e = e_first + u * NRX_BLOCK;
                            const int ec = e < n_e ? e : n_e - 1;
                            const int s = ec / cur;
                            const int l = ec - s * cur;
                            const int64_t gi = (b0 + s) * (int64_t)L + l0 + l;
                            pos4[u] = s * stride + l;
                            id4[u] = nrx_load_id(f.index, gi, f.idx64);
                            w4[u] = has_w ? f.weight[gi] : 1.0f;
                        }
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            if (e_first + u * NRX_BLOCK < n_e) {
                                const bool oob = (uint64_t)id4[u] >= (uint64_t)f.rows;
                                BagPair p;
                                p.id = oob ? 0 : (int32_t)id4[u];
                                p.w = oob ? 0.f : w4[u];
                                s_bag[pos4[u]] = p;
                            }
                        }
                    }
                    e_first = n_e;
                }
                for (int e = e_first; e < nb * cur; e += NRX_BLOCK) {
                    const int s = e / cur;
                    const int l = e - s * cur;
                    int64_t id;
                    float w;
                    bag_entry(f, b0 + s, s, s_off, l0 + l, L, id, w);
                    const bool oob = (uint64_t)id >= (uint64_t)f.rows;
                    BagPair p;
                    p.id = oob ? 0 : (int32_t)id;
                    p.w = oob ? 0.f : w;
                    s_bag[s * stride + l] = p;
                }
                __syncthreads();
                if (active) {
                    const BagPair* row = s_bag + sb * stride;
                    for (int l = 0; l < cur; ++l) {
                        const BagPair p = row[l];
                        if ((p.id != 0 || (f.flags & NRX_FEAT_ROW0_IS_DATA)) && p.w != 0.f)
                            atomic_add_row4(gtable, p.id, D, k0,
                                            make_float4(gs.x * p.w, gs.y * p.w, gs.z * p.w, gs.w * p.w));
                    }
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// Deterministic row-sparse backward of one table: segmented reduction over id-sorted lookups.
// One Q-lane group per unique row; entries of a row are summed in sorted (stable) order.
// --------------------------------------------------------------------------------------------
struct SortedBwdArgs {
    FeatDev f[NRX_MAX_FEATURES];      // the features reading this table (index pointer unused)
    int64_t off[NRX_MAX_FEATURES + 1];  // flat lookup offset of each feature
    int64_t batch;
    const float* g_out;
    int64_t out_ld;
    const float* g_wide;
    int64_t wide_ld;
    const int64_t* order;
    const int64_t* seg_start;
    const int64_t* uniq_keys;   // optional: (table << 40 | row) of each unique entry; row 0 -> zero grad
    int64_t n_unique;
    const int64_t* n_unique_dev;   // optional: actual count on the device (n_unique is then an upper bound)
    int64_t uniform_len;           // > 0: every feature has this many flat lookups (feature = p / uniform_len)
    uint64_t uniform_magic;        // floor(2^64 / uniform_len) + 1: p / uniform_len == mulhi64(p, magic) for p < 2^32
    const float* g_fm;             // optional FM gradient inputs (see nrx_fm_grad_t)
    const float* fm_sums;
    int64_t sums_ld;
    const float* feat;
    int64_t feat_ld;
    int32_t* long_ws;              // optional workspace of the long-segment path (see sorted_long_kernel); null = none
    int64_t long_items_cap;        // capacity of the item list
    int64_t long_slots_cap;        // capacity of the partial-sum slots
    const float* scale;            // fast form with bag features: per flat lookup, the factor of its upstream row (bag_scale_kernel)
    const float* bag_inv;          // ... and its compact form when every weight is 0 or 1 (long_ws[3] == 0): per (feature, sample)
    const uint32_t* bag_bits;      //     the factor of a weight-1 lookup, and one bit per lookup (weight != 0); both stay in L2
    float* values;
    int32_t n;
    int32_t dim;
    int32_t long_t;
    int32_t dense;                 // 0: row sums -> values[u];  1 / 2: straight into the dense gradient tables, f[t].index = base of table t's
                                   //    [rows, dim] gradient (the slot's id pointer is unused here), at the row the key names; 2 adds to what is there
    const float* gs_all;           // bag launches with 0/1 weights: ONE staging array of every feature's (pre-scaled) upstream rows
    const int32_t* walk;           // placement mode (nrx_embed_bwd_placed): the unique rows this launch reduces, and how many;
    const int64_t* n_walk_dev;     //   null = every unique row
    int32_t regular;               // 1: every feature single-valued, no wide routing, uniform_len > 0, out_col = col0 + i * col_stride,
    int32_t col0, col_stride;      //    one FM flag for all -- the per-feature fields are then arithmetic on the feature index
    int32_t all_fm;
};
static_assert(sizeof(SortedBwdArgs) <= 3840, "kernarg budget");

// Where the gradient row of unique entry u (key = table << 40 | row) goes: values[u], or -- dense mode (nrx_embed_bwd_placed_dense) -- its
// place in the table's dense gradient: what nrx_rows_to_dense did in a pass of its own (a read and a write of every unique row).
__device__ __forceinline__ NRX_GLOBAL float* sorted_row_dst(const NRX_CONST SortedBwdArgs* a, int64_t u, int64_t key) {
    if (a->dense == 0) return nrx_gmut<float>(a->values) + u * (int64_t)a->dim;
    return nrx_gmut<float>(reinterpret_cast<float*>(const_cast<void*>(a->f[key >> 40].index))) + (key & ((1ll << 40) - 1)) * (int64_t)a->dim;
}
template <int Q>
__device__ __forceinline__ void sorted_store4(const NRX_CONST SortedBwdArgs* a, int64_t u, int64_t key, int q, float4 acc) {
    NRX_GLOBAL nrx_f32x4* dst = reinterpret_cast<NRX_GLOBAL nrx_f32x4*>(sorted_row_dst(a, u, key)) + q;       // dim == 4 Q in the fast kernels
    if (a->dense == 2) {
        const nrx_f32x4 o = *dst;
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    nrx_f32x4 t;
    t.x = acc.x; t.y = acc.y; t.z = acc.z; t.w = acc.w;
    *dst = t;
}

// feature of flat lookup p (<= 64 features): direct when every feature contributes the same number of lookups
__device__ __forceinline__ int sorted_feat_of(const NRX_CONST SortedBwdArgs* a, int64_t p) {
    if (a->uniform_len > 0) return (int)__umul64hi((uint64_t)p, a->uniform_magic);    // exact for p < 2^32 (host-checked)
    if (a->n <= 4) {                  // few features (a shared table: history + item id): compare against scalars, no loads
        const int64_t big = 0x7fffffffffffffffLL;
        const int64_t o1 = a->off[1], o2 = a->n > 2 ? a->off[2] : big, o3 = a->n > 3 ? a->off[3] : big;
        return (int)(p >= o1) + (int)(p >= o2) + (int)(p >= o3);
    }
    int l0 = 0, h0 = a->n;            // a chain of dependent loads from the argument block
    while (h0 - l0 > 1) {
        const int mid = (l0 + h0) >> 1;
        if (a->off[mid] <= p) l0 = mid; else h0 = mid;
    }
    return l0;
}

// General form: any feature kinds (bags, wide routing), any dim.  One Q-lane group per unique row.
template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void embed_bwd_sorted_kernel(const SortedBwdArgs args_in_kernarg) {
    const NRX_CONST SortedBwdArgs* a = nrx_kernarg<SortedBwdArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    const int q = threadIdx.x & (Q - 1);
    const int64_t u = (int64_t)blockIdx.x * TB + (threadIdx.x >> QLOG2);
    if (u >= a->n_unique) return;
    if (a->n_unique_dev != nullptr && u >= nrx_gconst<int64_t>(a->n_unique_dev)[0]) return;
    const int D = a->dim;
    const int64_t lo = nrx_gconst<int64_t>(a->seg_start)[u];
    int64_t hi = nrx_gconst<int64_t>(a->seg_start)[u + 1];
    // padding row (id 0) never trains (nn.Embedding(padding_idx=0)): its segment is skipped, zeros are written
    const int64_t ukey = a->uniq_keys != nullptr ? nrx_gconst<int64_t>(a->uniq_keys)[u] : 1;
    if ((ukey & ((1ll << 40) - 1)) == 0) hi = lo;
    const bool add_to = a->dense == 2;
    constexpr int MAXC = 4;            // up to 4 column chunks per lane (D <= 16 Q), else the slow loop below
    float acc[MAXC][4];
    const int nchunk = (D + 4 * Q - 1) / (4 * Q);
    if (nchunk <= MAXC) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
        for (int64_t e = lo; e < hi; ++e) {
            const int64_t p = nrx_gconst<int64_t>(a->order)[e];
            const int fi = sorted_feat_of(a, p);
            const int64_t r = p - a->off[fi];
            const int kind = a->f[fi].kind;
            const int L = kind >= NRX_BAG_MASKED_MEAN ? a->f[fi].bag_len : 1;
            const int64_t b = r / L;
            float scale = 1.0f;
            if (kind == NRX_BAG_MASKED_MEAN) {      // the Q lanes share the row of L weights, then a group reduction
                const NRX_GLOBAL float* w = nrx_gconst<float>(a->f[fi].weight) + b * L;
                float den = 0.f;
                for (int l = q; l < L; l += Q) den += w[l];
                den = group_sum<Q>(den);
                scale = w[r - b * L] / (den + 1e-8f);
            } else if (kind == NRX_BAG_MEAN) {
                scale = 1.0f / (float)L;
            } else if (kind == NRX_BAG_SUM && a->f[fi].weight != nullptr) {
                scale = nrx_gconst<float>(a->f[fi].weight)[r];
            }
            const int wide_col = a->f[fi].wide_col, out_col = a->f[fi].out_col;
            const bool fm = a->g_fm != nullptr && a->f[fi].fm;
            const float gf = fm ? nrx_gconst<float>(a->g_fm)[b] : 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int k0 = (c * Q + q) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    if (c < nchunk && k < D) {
                        float g;
                        if (wide_col >= 0) {
                            g = (k == 0) ? (a->g_wide ? nrx_gconst<float>(a->g_wide)[b * a->wide_ld + wide_col] : 0.f)
                                         : (a->g_out ? nrx_gconst<float>(a->g_out)[b * a->out_ld + out_col + k - 1] : 0.f);
                        } else {
                            g = a->g_out ? nrx_gconst<float>(a->g_out)[b * a->out_ld + out_col + k] : 0.f;
                            if (fm)
                                g += (k == 0) ? gf
                                              : gf * (nrx_gconst<float>(a->fm_sums)[b * a->sums_ld + k] -
                                                      nrx_gconst<float>(a->feat)[b * a->feat_ld + out_col + k]);
                        }
                        acc[c][j] += g * scale;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k0 = (c * Q + q) * 4;
            NRX_GLOBAL float* dst = sorted_row_dst(a, u, ukey) + k0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c < nchunk && k0 + j < D) dst[j] = add_to ? acc[c][j] + dst[j] : acc[c][j];
        }
        return;
    }
    // very wide rows (D > 16 Q = 1024 at Q = 64): column chunk outermost, the entries are re-walked per chunk
    for (int k0 = q * 4; k0 < D; k0 += 4 * Q) {
        float ac[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t e = lo; e < hi; ++e) {
            const int64_t p = nrx_gconst<int64_t>(a->order)[e];
            const int fi = sorted_feat_of(a, p);
            const int64_t r = p - a->off[fi];
            const int kind = a->f[fi].kind;
            const int L = kind >= NRX_BAG_MASKED_MEAN ? a->f[fi].bag_len : 1;
            const int64_t b = r / L;
            float scale = 1.0f;
            if (kind == NRX_BAG_MASKED_MEAN) {
                const NRX_GLOBAL float* w = nrx_gconst<float>(a->f[fi].weight) + b * L;
                float den = 0.f;
                for (int l = 0; l < L; ++l) den += w[l];
                scale = w[r - b * L] / (den + 1e-8f);
            } else if (kind == NRX_BAG_MEAN) {
                scale = 1.0f / (float)L;
            } else if (kind == NRX_BAG_SUM && a->f[fi].weight != nullptr) {
                scale = nrx_gconst<float>(a->f[fi].weight)[r];
            }
            const int wide_col = a->f[fi].wide_col, out_col = a->f[fi].out_col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j;
                if (k < D) {
                    float g;
                    if (wide_col >= 0)
                        g = (k == 0) ? (a->g_wide ? nrx_gconst<float>(a->g_wide)[b * a->wide_ld + wide_col] : 0.f)
                                     : (a->g_out ? nrx_gconst<float>(a->g_out)[b * a->out_ld + out_col + k - 1] : 0.f);
                    else
                        g = a->g_out ? nrx_gconst<float>(a->g_out)[b * a->out_ld + out_col + k] : 0.f;
                    ac[j] += g * scale;
                }
            }
        }
        NRX_GLOBAL float* dst = sorted_row_dst(a, u, ukey) + k0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < D) dst[j] = add_to ? ac[j] + dst[j] : ac[j];
    }
}

// ---- work lists of the long-segment path (workspace layout: 4 counters | items | multi-chunk rows | partial sums)
constexpr int SORTED_LONG_T = 16;            // segments longer than this leave the lane-group kernel (32 for bag launches)
#ifndef NRX_LONG_CHUNK
#define NRX_LONG_CHUNK 256
#endif
constexpr int SORTED_LONG_CHUNK = NRX_LONG_CHUNK;       // entries per work item (small: items are the unit of load balance)
struct LongItem { int32_t u; int32_t dest; int64_t e_begin; int32_t len; int32_t m; };      // dest < 0: straight to values[u]; m: its row's LongMulti (several items)
struct LongMulti { int32_t u; int32_t slot0; int32_t nchunks; int32_t done; };              // done: items (or item GROUPS) of the row finished so far (the last one adds the partials)
// Rows of more than SORTED_LONG_GROUP items (> 8192 lookups: the hottest ids of a Zipf law) count their items in groups of that many: a group's
// last finisher adds the group's partials into a second-level partial, the last GROUP's finisher adds those.  One counter for all items of
// such a row is ~1000 device-scope atomics on ONE address, which the memory side serialises at ~0.1 us each (C4 with Zipf ids: +90 us).
// Slot layout of a multi-item row: [slot0, +nchunks) item partials | [+ngroups) group partials | [+ (ngroups + 3) / 4) slots of int32 group counters.
constexpr int SORTED_LONG_GROUP = 32;
__host__ __device__ __forceinline__ int sorted_long_groups(int nchunks) { return nchunks > SORTED_LONG_GROUP ? (nchunks + SORTED_LONG_GROUP - 1) / SORTED_LONG_GROUP : 0; }
__host__ __device__ __forceinline__ int sorted_long_slots(int nchunks) {
    const int ng = sorted_long_groups(nchunks);
    return nchunks + (ng ? ng + (ng + 3) / 4 : 0);
}
// (sum of sorted_long_slots over the multi-item rows of n lookups: <= n / 256 + n / 257 item slots, + <= n / 4096 group slots and counter slots)
__host__ __device__ __forceinline__ int64_t sorted_long_slots_cap(int64_t n) { return 2 * n / SORTED_LONG_CHUNK + n / 2048 + 8; }

__device__ __forceinline__ LongItem* sorted_long_items(const NRX_CONST SortedBwdArgs* a) {
    return reinterpret_cast<LongItem*>(a->long_ws + 4);
}
__device__ __forceinline__ LongMulti* sorted_long_multi(const NRX_CONST SortedBwdArgs* a) {
    return reinterpret_cast<LongMulti*>(reinterpret_cast<char*>(a->long_ws + 4) + a->long_items_cap * sizeof(LongItem));
}
__device__ __forceinline__ float* sorted_long_partials(const NRX_CONST SortedBwdArgs* a) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(sorted_long_multi(a)) + a->long_slots_cap * sizeof(LongMulti));
}

__device__ __forceinline__ void sorted_long_write_items(const NRX_CONST SortedBwdArgs* a, int32_t u, int64_t lo, int64_t hi, int nchunks,
                                                        int slot0, int base, int m) {
    for (int c = 0; c < nchunks; ++c) {
        if (base + c >= a->long_items_cap) break;
        LongItem w;
        w.u = u;
        w.dest = nchunks > 1 ? slot0 + c : -1;
        w.e_begin = lo + (int64_t)c * SORTED_LONG_CHUNK;
        w.len = (int32_t)((w.e_begin + SORTED_LONG_CHUNK < hi ? w.e_begin + SORTED_LONG_CHUNK : hi) - w.e_begin);
        w.m = m;
        sorted_long_items(a)[base + c] = w;
    }
}

// Fast form for the common launch: every feature single-valued without wide routing, D = 4 Q, everything 16-byte
// aligned.  A Q-lane group owns R consecutive unique rows: their segment bounds, then their first entries' lookup
// indices, then all their upstream rows (g_out, and for FM fields the forward value and the field sums) are fetched as
// R independent requests per lane -- the reduction is a chain of three dependent random reads per row, so what bounds it
// is how many chains a lane keeps in flight.  Segments longer than one entry (duplicate ids) continue in a loop that
// adds the remaining entries in sorted order: the summation order is the sorted order, as in the general form.
// UNAL: the launch has Wide&Deep column routing (widedeep/model.py:53-69) -- after the first wide feature nothing in g_out is
// 16-byte aligned (deep blocks of D - 1 floats), and column 0 of a wide feature's gradient comes from g_wide.  Global memory
// only needs dword alignment for multi-dword accesses: the lane's 4 floats are one dword-aligned 16-byte load at the shifted
// position; lane 0 of a wide feature takes 3 floats from g_out and its first one from g_wide.
typedef float nrx_f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float nrx_f32x3u __attribute__((ext_vector_type(3), aligned(4)));

// What the reduction needs to know about the feature of a lookup.  The feature index differs from lane group to lane group,
// so reading the argument block by it is a VECTOR load per field -- five per lookup, each a full pass of the CU's address
// unit for a handful of bytes; the C4 launch (history + item id on the news table) issued 3 M load instructions for 3.3 M
// lookups and was bound by exactly that.  Launches of <= 4 features read all candidates with scalar loads and select.
struct FeatLite {
    int64_t off;        // flat lookup offset of the feature
    uint64_t magic;     // 2^64 reciprocal of bag_len (bags)
    int32_t out_col, wide_col;
    int32_t bag_len;
    int32_t kind;
    bool fm;
    const float* gs;    // bag feature with 0/1 weights: its upstream rows ALREADY scaled, [B, dim] (bag_scale_kernel), or null
};
__device__ __forceinline__ FeatLite sorted_feat(const NRX_CONST SortedBwdArgs* a, int fi) {
    FeatLite f;
    if (a->regular) {                  // (the C2 / C5 shape: 26 / 40 single-valued features laid out back to back)
        f.off = (int64_t)fi * a->uniform_len; f.magic = 0; f.out_col = a->col0 + fi * a->col_stride; f.wide_col = -1;
        f.bag_len = 0; f.kind = NRX_SPARSE; f.fm = a->all_fm != 0; f.gs = nullptr;
        return f;
    }
    if (a->n <= 4) {
        f.off = a->off[0]; f.magic = (uint64_t)a->f[0].rows; f.out_col = a->f[0].out_col; f.wide_col = a->f[0].wide_col;
        f.bag_len = a->f[0].bag_len; f.kind = a->f[0].kind; f.fm = a->f[0].fm != 0; f.gs = a->f[0].table;
#pragma unroll
        for (int i = 1; i < 4; ++i) {           // constant indices: scalar loads (entries past n are inside the block and unused)
            const bool hit = fi == i;
            f.off = hit ? a->off[i] : f.off;
            f.magic = hit ? (uint64_t)a->f[i].rows : f.magic;
            f.out_col = hit ? a->f[i].out_col : f.out_col;
            f.wide_col = hit ? a->f[i].wide_col : f.wide_col;
            f.bag_len = hit ? (int32_t)a->f[i].bag_len : f.bag_len;
            f.kind = hit ? (int32_t)a->f[i].kind : f.kind;
            f.fm = hit ? a->f[i].fm != 0 : f.fm;
            f.gs = hit ? a->f[i].table : f.gs;
        }
        return f;
    }
    f.off = a->off[fi]; f.magic = (uint64_t)a->f[fi].rows; f.out_col = a->f[fi].out_col; f.wide_col = a->f[fi].wide_col;
    f.bag_len = a->f[fi].bag_len; f.kind = a->f[fi].kind; f.fm = a->f[fi].fm != 0; f.gs = a->f[fi].table;
    return f;
}
// d fm / d field folded into an upstream row chunk (columns 4q .. 4q+3 of the field): column 0 -> g_fm, column k -> g_fm (S_k - v_k).
// One definition (explicit fused multiply-adds) for the walk, the work-list and the placement kernels: their results must agree
// bit for bit.
__device__ __forceinline__ void fm_fold4(float4& t, float gf, const float4& s, const float4& v, int q) {
    t.x = q == 0 ? t.x + gf : __builtin_fmaf(gf, s.x - v.x, t.x);
    t.y = __builtin_fmaf(gf, s.y - v.y, t.y);
    t.z = __builtin_fmaf(gf, s.z - v.z, t.z);
    t.w = __builtin_fmaf(gf, s.w - v.w, t.w);
}

// Columns 4q .. 4q+3 of a feature's upstream row.  UNAL: see above (shifted deep blocks, column 0 from the wide gradient).
template <bool UNAL>
__device__ __forceinline__ float4 upstream_chunk(const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld, int out_col,
                                                 int wide_col, int64_t b, int q) {
    if (!UNAL) return g_out ? nrx_ldg4(g_out, (b * out_ld + out_col) / 4 + q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int wc = wide_col;                                   // the lanes of a group share the feature
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    // one load shape for every lane (a second path for "lane 0 of a wide feature" would run serially in every wavefront that
    // mixes wide and plain rows): 16 bytes at the shifted position; lane 0 of a wide feature thereby reads one float of the
    // previous feature's block, which the select below replaces by the wide gradient -- except at the very first element of
    // g_out (sample 0, column 0), where the load starts one float later and is shifted back in registers
    int64_t eo = b * out_ld + out_col + 4 * q - (wc >= 0 ? 1 : 0);
    const bool edge = eo < 0;
    if (g_out) {
        const nrx_f32x4u t = *reinterpret_cast<const nrx_f32x4u*>(g_out + (edge ? 0 : eo));
        g = edge ? make_float4(0.f, t.x, t.y, t.z) : make_float4(t.x, t.y, t.z, t.w);
    }
    if (wc >= 0 && q == 0) g.x = g_wide ? nrx_gconst<float>(g_wide)[b * wide_ld + wc] : 0.f;
    return g;
}
template <bool UNAL>
__device__ __forceinline__ float4 sorted_upstream(const NRX_CONST SortedBwdArgs* a, const FeatLite& f, int64_t b, int q) {
    return upstream_chunk<UNAL>(a->g_out, a->out_ld, a->g_wide, a->wide_ld, f.out_col, f.wide_col, b, q);
}

// ---- Placement pass (nrx_embed_bwd_placed).  A unique row looked up ONCE in the launch needs no reduction: its gradient is
// that lookup's upstream row.  The plan (nrx_sparse_plan_place) says where it goes: dest[flat lookup] = unique index, or -1.
// This kernel is the forward ring kernel run backwards: a sample is owned by Q lanes, the block's dest words go through LDS
// once (coalesced), the upstream rows -- g_out, and for FM fields the forward concat and the field sums -- are read where
// they lie, sample-major and fully coalesced, and each placed row leaves as one 4-D-byte store.  The sorted walk reads the
// same rows through the sort permutation instead: one 128-byte fabric request per 64-byte row and array (C2: 678 MB of
// requests for 230 MB of operands -- profiles/r03_bwd_c2_counters_before.txt).
struct PlaceArgs {
    int64_t off[NRX_MAX_FEATURES];       // flat lookup offset of the i-th PLACEABLE (single-valued) feature
    int32_t out_col[NRX_MAX_FEATURES];
    int32_t wide_col[NRX_MAX_FEATURES];
    uint8_t fm[NRX_MAX_FEATURES];
    int64_t batch;
    const float* g_out;
    int64_t out_ld;
    const float* g_wide;
    int64_t wide_ld;
    const float* g_fm;
    const float* fm_sums;
    int64_t sums_ld;
    const float* feat;
    int64_t feat_ld;
    const int32_t* dest;
    float* values;
    int32_t* long_ws;                    // the walk's four work-list counters: cleared here (the walk is the next launch)
    int32_t n;
    int32_t nt;                          // non-temporal upstream loads (default; NRX_PLACE_NT=0 turns them off): the rows are read once
    // DENSE (nrx_embed_bwd_placed_dense): a placed row goes straight to its place in the table's dense gradient -- the lookup's own id
    // names the row (the ids are read where they lie, sample-major), dest >= 0 only says "placed"
    const void* ids[NRX_MAX_FEATURES];   // per placeable feature
    float* grad[NRX_MAX_FEATURES];       // per placeable feature: base of its table's [rows, 4 Q] gradient
    int32_t idx64;
    int32_t add_to;                      // 1: add to what the table holds (a table fed by a second launch group)
    int32_t stnt;                        // 1: placed rows leave with non-temporal stores (measurement knob NRX_PLACE_STNT)
    uint64_t fm_mask;                    // bit f = fm[f] (the full-line form reads flags of two features per step from here)
    int32_t multi_shift;                 // MULTI (nrx_embed_bwd_scatter_multi): dest = (base number << multi_shift) | row; the bases live in grad[]
};
static_assert(sizeof(PlaceArgs) <= 3584, "kernarg budget");

template <int QLOG2, int U, bool FM, bool UNAL, bool DENSE = false, bool MULTI = false>
__global__ __launch_bounds__(NRX_BLOCK) void embed_bwd_place_kernel(const PlaceArgs args_in_kernarg) {
    const NRX_CONST PlaceArgs* a = nrx_kernarg<PlaceArgs>();
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;                 // samples per block (<= 64)
    constexpr int FPW = 64 / TB;                      // features staged per wavefront instruction
    extern __shared__ __attribute__((aligned(16))) int32_t s_dest[];      // [n][TB]
    const int tid = threadIdx.x;
    const int n = a->n;
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int nb = (int)((a->batch - b0) < (int64_t)TB ? (a->batch - b0) : (int64_t)TB);
    if (blockIdx.x == 0 && tid < 4 && a->long_ws != nullptr) a->long_ws[tid] = 0;
    {   // ---- stage the block's dest words: a wavefront instruction covers FPW features x TB samples (TB consecutive words each)
        const int lane = tid & 63, wave = tid >> 6;
        const int s = lane & (TB - 1), fl = lane / TB;
        constexpr int PASS = 4;
        for (int k0 = 0; k0 * 4 * FPW < n; k0 += PASS) {
            int32_t d[PASS];
#pragma unroll
            for (int u = 0; u < PASS; ++u) {
                const int f = ((k0 + u) * 4 + wave) * FPW + fl;
                const int fc = f < n ? f : n - 1;
                d[u] = nrx_gconst<int32_t>(a->dest)[a->off[fc] + b0 + (s < nb ? s : nb - 1)];
            }
#pragma unroll
            for (int u = 0; u < PASS; ++u) {
                const int f = ((k0 + u) * 4 + wave) * FPW + fl;
                if (f < n) s_dest[f * TB + s] = s < nb ? d[u] : -1;
            }
        }
    }
    __syncthreads();
    const int q = tid & (Q - 1);
    const int sb = tid >> QLOG2;
    const int64_t b = b0 + sb;
    if (b >= a->batch) return;
    float gf = 0.f;
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FM) {
        gf = nrx_gconst<float>(a->g_fm)[b];
        S = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
    }
    // Measured on C2 (26 x D=16, FM; profiles/r03_place_variants.txt): the pass alone 78 us = loads only 38 us + stores only 28 us and
    // then some -- sequential reads and random 64-byte row writes do not overlap well; non-temporal loads -7 us (C5 set: -16 us),
    // 4 instead of 8 fetches in flight -2 us, 13 in flight +36 us (3 waves per SIMD).
    const bool nt = a->nt != 0;
    auto fetch = [&](int f, int32_t& d, float4& g, float4& v) {          // f wave-uniform: column numbers come from scalar loads
        if (d >= 0) {
            if (DENSE) {        // the row number replaces the unique index (a placed lookup's id is in range and not 0: the plan saw it)
                const void* ip = a->ids[f];
                d = a->idx64 ? (int32_t)nrx_gconst<int64_t>(ip)[b] : nrx_gconst<int32_t>(ip)[b];
            }
            if (!UNAL && nt) {
                g = a->g_out != nullptr ? nrx_ldg4_nt(a->g_out, (b * a->out_ld + a->out_col[f]) / 4 + q) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (FM) v = nrx_ldg4_nt(a->feat, (b * a->feat_ld + a->out_col[f]) / 4 + q);
                return;
            }
            g = upstream_chunk<UNAL>(a->g_out, a->out_ld, a->g_wide, a->wide_ld, a->out_col[f], a->wide_col[f], b, q);
            if (FM) v = nrx_ldg4(a->feat, (b * a->feat_ld + a->out_col[f]) / 4 + q);
        }
    };
    auto place = [&](int f, int32_t d, const float4& g, const float4& v) {
        if (d >= 0) {
            float4 t = g;
            if (FM) fm_fold4(t, a->fm[f] ? gf : 0.f, S, v, q);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);         // 0 + t, as the walk forms it (a -0 becomes +0 there too)
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            if (DENSE) {
                float* base = a->grad[f];
                if (a->add_to) {
                    const float4 o = nrx_ldg4(base, (int64_t)d * Q + q);
                    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                }
                nrx_stg4(base, (int64_t)d * Q + q, acc);
            } else if (MULTI) {     // the row goes to one of several buffers (the owners' arenas, as this process maps them): base number in the high bits
                float* base = a->grad[(uint32_t)d >> a->multi_shift];
                nrx_stg4(base, (int64_t)(d & ((1 << a->multi_shift) - 1)) * Q + q, acc);
            } else if (a->stnt) {
                nrx_f32x4 tv;
                tv.x = acc.x; tv.y = acc.y; tv.z = acc.z; tv.w = acc.w;
                __builtin_nontemporal_store(tv, (NRX_GLOBAL nrx_f32x4*)(a->values) + (int64_t)d * Q + q);
            } else {
                nrx_stg4(a->values, (int64_t)d * Q + q, acc);
            }
        }
    };
    if (n >= U) {
        // ring, as in the forward (embed_fwd_ring): `place feature f; fetch feature f + U` -- U row fetches in flight per lane from
        // the first feature to the last, loads and stores interleaved at row granularity
        const int32_t* s_my = s_dest + sb;
        int32_t d[U];
        float4 g[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = s_my[u * TB];
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(u, d[u], g[u], v[u]);
        int f0 = 0;
        for (; f0 + 2 * U <= n; f0 += U) {
            int32_t dn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dn[u] = s_my[(f0 + U + u) * TB];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                place(f0 + u, d[u], g[u], v[u]);
                d[u] = dn[u];
                fetch(f0 + U + u, d[u], g[u], v[u]);
            }
        }
        // f0 + U <= n < f0 + 2U: drain; the n - f0 - U fetches still to be issued sit behind wave-uniform branches
        int32_t dn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = f0 + U + u;
            dn[u] = f < n ? s_my[f * TB] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            place(f0 + u, d[u], g[u], v[u]);
            d[u] = dn[u];
            if (f0 + U + u < n) fetch(f0 + U + u, d[u], g[u], v[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (f0 + U + u < n) place(f0 + U + u, d[u], g[u], v[u]);
        return;
    }
    for (int f0 = 0; f0 < n; f0 += U) {
        int32_t d[U];
        float4 g[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = f0 + u < n ? s_dest[(f0 + u) * TB + sb] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(f0 + u < n ? f0 + u : n - 1, d[u], g[u], v[u]);      // wave-uniform feature: column numbers come from scalar loads
#pragma unroll
        for (int u = 0; u < U; ++u) place(f0 + u < n ? f0 + u : n - 1, d[u], g[u], v[u]);
    }
}

// Full-line form of the placement pass for 64-byte rows (D = 16): a sample is owned by 2 Q = 8 lanes that take features 2j and 2j + 1 TOGETHER,
// so one load instruction covers whole 128-byte lines of the upstream rows (and of the forward concat for FM) -- in the form above the two
// halves of a line are requested by different instructions a ring step apart: 3.4 M 64-byte read requests per C2 launch and, with the
// non-temporal hint, 318 MB fetched for 229 MB of distinct lines.  Same arithmetic per (sample, feature): bit-identical results.  The host
// picks it when every pair (2j, 2j + 1) is one aligned line (place_lines_ok).
#ifndef NRX_LINES_U
#define NRX_LINES_U 2                   // feature PAIRS in flight per lane (build-time knob for tools/build_variant.sh): C2 launch on one box, rotated
                                        // builds: 8 -> 71.1 us, 4 -> 60.6 / 65.2 (two boxes), 3 -> 64.0, 2 -> 59.6 / 63.0, 1 -> 62.6
#endif
constexpr int PLACE_LINES_TBP = 40;
// HASG = false: no upstream gradient of the concat (an FM model whose loss reads the logit only, fm/model.py:44-59: the rows' gradient is the FM term
// alone) -- the pass then reads the forward concat and the field sums, not 109 MB of zeros.
template <int U, bool FM, bool DENSE, bool HASG = true>
__global__ __launch_bounds__(NRX_BLOCK) void embed_bwd_place_lines_kernel(const PlaceArgs args_in_kernarg) {
    const NRX_CONST PlaceArgs* a = nrx_kernarg<PlaceArgs>();
    constexpr int Q = 4;
    constexpr int TB = NRX_BLOCK / (2 * Q);           // samples per block
    // a feature's TB dest words sit TBP apart: with stride TB = 32 the words of features 2j and 2j + 1 of a sample share a bank, and a wavefront
    // (8 samples x the two features of a pair) read them as a two-way conflict: 44 % of the kernel's LDS cycles (profiles/r04_fwd_bwd_c2_rocprof_summary.txt).
    // 40 = 8 mod 32: the second feature's eight words land eight banks on.
    constexpr int TBP = PLACE_LINES_TBP;
    extern __shared__ __attribute__((aligned(16))) int32_t s_dest[];      // [n][TBP]
    const int tid = threadIdx.x;
    const int n = a->n;
    const int64_t b0 = (int64_t)blockIdx.x * TB;
    const int nb = (int)((a->batch - b0) < (int64_t)TB ? (a->batch - b0) : (int64_t)TB);
    if (blockIdx.x == 0 && tid < 4 && a->long_ws != nullptr) a->long_ws[tid] = 0;
    for (int i = tid; i < n * TB; i += NRX_BLOCK) {   // TB consecutive dest words per feature
        const int f = i / TB, s = i - f * TB;
        s_dest[f * TBP + s] = s < nb ? nrx_gconst<int32_t>(a->dest)[a->off[f] + b0 + s] : -1;
    }
    __syncthreads();
    const int q = tid & (Q - 1);
    const int par = (tid >> 2) & 1;
    const int sb = tid >> 3;
    const int64_t b = b0 + sb;
    if (b >= a->batch) return;
    float gf = 0.f;
    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FM) {
        gf = nrx_gconst<float>(a->g_fm)[b];
        S = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
    }
    const int np = (n + 1) >> 1;                      // feature pairs (an odd last feature: the upper half of the lanes idles)
    const int32_t* s_my = s_dest + sb;
    auto dest_of = [&](int j) -> int32_t {
        const int f = 2 * j + par;
        return f < n ? s_my[f * TBP] : -1;
    };
    auto fetch = [&](int j, int32_t& d, float4& g, float4& v) {           // j wave-uniform: the two features' columns come from scalar loads
        if (d >= 0) {
            const int f0 = 2 * j, f1 = 2 * j + 1 < n ? 2 * j + 1 : 2 * j;
            const int col = par ? a->out_col[f1] : a->out_col[f0];
            if (DENSE) {
                const void* ip = par ? a->ids[f1] : a->ids[f0];
                d = a->idx64 ? (int32_t)nrx_gconst<int64_t>(ip)[b] : nrx_gconst<int32_t>(ip)[b];
            }
            g = HASG ? nrx_ldg4_nt(a->g_out, (b * a->out_ld + col) / 4 + q) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (FM) v = nrx_ldg4_nt(a->feat, (b * a->feat_ld + col) / 4 + q);
        }
    };
    auto place = [&](int j, int32_t d, const float4& g, const float4& v) {
        if (d >= 0) {
            const int f0 = 2 * j, f1 = 2 * j + 1 < n ? 2 * j + 1 : 2 * j;
            float4 t = g;
            if (FM) fm_fold4(t, ((a->fm_mask >> (par ? f1 : f0)) & 1ull) ? gf : 0.f, S, v, q);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);         // 0 + t, as the walk forms it (a -0 becomes +0 there too)
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            if (DENSE) {
                float* base = par ? a->grad[f1] : a->grad[f0];
                if (a->add_to) {
                    const float4 o = nrx_ldg4(base, (int64_t)d * Q + q);
                    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                }
                nrx_stg4(base, (int64_t)d * Q + q, acc);
            } else {
                nrx_stg4(a->values, (int64_t)d * Q + q, acc);
            }
        }
    };
    if (np >= U) {
        int32_t d[U];
        float4 g[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = dest_of(u);
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(u, d[u], g[u], v[u]);
        int j0 = 0;
        for (; j0 + 2 * U <= np; j0 += U) {
            int32_t dn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dn[u] = dest_of(j0 + U + u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                place(j0 + u, d[u], g[u], v[u]);
                d[u] = dn[u];
                fetch(j0 + U + u, d[u], g[u], v[u]);
            }
        }
        int32_t dn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dn[u] = j0 + U + u < np ? dest_of(j0 + U + u) : -1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            place(j0 + u, d[u], g[u], v[u]);
            d[u] = dn[u];
            if (j0 + U + u < np) fetch(j0 + U + u, d[u], g[u], v[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j0 + U + u < np) place(j0 + U + u, d[u], g[u], v[u]);
        return;
    }
    for (int j0 = 0; j0 < np; j0 += U) {
        int32_t d[U];
        float4 g[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = j0 + u < np ? dest_of(j0 + u) : -1;
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(j0 + u < np ? j0 + u : np - 1, d[u], g[u], v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) place(j0 + u < np ? j0 + u : np - 1, d[u], g[u], v[u]);
    }
}

// Upstream row chunk + factor of ONE sorted entry (flat lookup p of feature f) for launches with bag features.  Three forms, all
// giving the same product bit for bit:  general weights: g_out row x scale[p];  0/1 weights: g_out row x (bit ? inv[sample] : 0);
// 0/1 weights with the rows pre-scaled by bag_scale_kernel (f.gs): gs row x (bit ? 1 : 0) -- one L2-resident row request per
// lookup instead of a row, a factor and a bit word (the C4 walk issued 13 M L1->L2 requests for 3.4 M lookups and ran at their
// rate: profiles/r03_bwd_c4_counters_before.txt).  The bit word is skipped as well when no live lookup has weight 0 (DataReader's
// masks are zero exactly at the padding ids, whose row is never walked).
template <bool UNAL>
__device__ __forceinline__ void sorted_bag_entry(const NRX_CONST SortedBwdArgs* a, const FeatLite& f, int64_t p, int q, int Q,
                                                 bool bag_binary, bool need_bits, int64_t& b, float4& g, float& sc) {
    b = p - f.off;
    sc = 1.0f;
    if (f.kind >= NRX_BAG_MASKED_MEAN) {
        if (f.bag_len > 1) b = (int64_t)__umul64hi((uint64_t)b, f.magic);   // b / bag_len (b < 2^32)
        if (bag_binary) {
            bool bit = true;
            if (need_bits || f.gs == nullptr) bit = (nrx_gconst<uint32_t>(a->bag_bits)[p >> 5] >> (p & 31)) & 1u;
            if (!UNAL && f.gs != nullptr) {
                sc = bit ? 1.0f : 0.f;
                g = nrx_ldg4(f.gs, b * Q + q);
                return;
            }
            sc = bit ? nrx_gconst<float>(a->bag_inv)[f.off + b] : 0.f;
        } else {
            sc = nrx_gconst<float>(a->scale)[p];
        }
    }
    g = sorted_upstream<UNAL>(a, f, b, q);
}

// BAG: some features are bags -- a lookup's sample is (flat index) / L and its upstream row is scaled by the per-lookup
// factor bag_scale_kernel left in a->scale (mask / (sum mask + 1e-8), 1 / L, or the weight).
// The general decode (more than 4 features, not regular) through an LDS copy of the per-feature fields: the same fields read from the
// argument block with per-lane indices are VECTOR loads (and a binary search over a->off[] a chain of them), counted in the same in-order
// queue as the row loads -- every row load then waited for the decode loads of the entry before it.  LDS reads are counted separately.
struct SortedFeatLds {
    int64_t off[NRX_MAX_FEATURES + 1];
    FeatLite f[NRX_MAX_FEATURES];
};
__device__ __forceinline__ void sorted_feat_stage(const NRX_CONST SortedBwdArgs* a, SortedFeatLds* t) {      // block-uniform call; ends on a barrier
    for (int i = threadIdx.x; i <= a->n; i += NRX_BLOCK) t->off[i] = a->off[i];
    for (int i = threadIdx.x; i < a->n; i += NRX_BLOCK) {
        FeatLite f;
        f.off = a->off[i]; f.magic = (uint64_t)a->f[i].rows; f.out_col = a->f[i].out_col; f.wide_col = a->f[i].wide_col;
        f.bag_len = a->f[i].bag_len; f.kind = a->f[i].kind; f.fm = a->f[i].fm != 0; f.gs = a->f[i].table;
        t->f[i] = f;
    }
    __syncthreads();
}
// DEC (a template argument of the walk / work-list kernels: ONE decode form per instruction stream -- with several behind run-time branches the
// compiler's wait counts at every row load are those of the worst form):  1 = regular (arithmetic, see REG below);  2 = at most 4 features
// (scalar compares and selects on argument words, no loads);  0 = general: the LDS copy.
template <int DEC>
__device__ __forceinline__ FeatLite sorted_decode(const NRX_CONST SortedBwdArgs* a, const SortedFeatLds* t, int64_t p) {
    if (DEC == 2) {
        const int64_t big = 0x7fffffffffffffffLL;
        const int64_t o1 = a->n > 1 ? a->off[1] : big, o2 = a->n > 2 ? a->off[2] : big, o3 = a->n > 3 ? a->off[3] : big;
        const int fi = (int)(p >= o1) + (int)(p >= o2) + (int)(p >= o3);
        FeatLite f;
        f.off = a->off[0]; f.magic = (uint64_t)a->f[0].rows; f.out_col = a->f[0].out_col; f.wide_col = a->f[0].wide_col;
        f.bag_len = a->f[0].bag_len; f.kind = a->f[0].kind; f.fm = a->f[0].fm != 0; f.gs = a->f[0].table;
#pragma unroll
        for (int i = 1; i < 4; ++i) {           // constant indices: scalar loads (entries past n are inside the block and unused)
            const bool hit = fi == i;
            f.off = hit ? a->off[i] : f.off;
            f.magic = hit ? (uint64_t)a->f[i].rows : f.magic;
            f.out_col = hit ? a->f[i].out_col : f.out_col;
            f.wide_col = hit ? a->f[i].wide_col : f.wide_col;
            f.bag_len = hit ? (int32_t)a->f[i].bag_len : f.bag_len;
            f.kind = hit ? (int32_t)a->f[i].kind : f.kind;
            f.fm = hit ? a->f[i].fm != 0 : f.fm;
            f.gs = hit ? a->f[i].table : f.gs;
        }
        return f;
    }
    int fi;
    if (a->uniform_len > 0) {
        fi = (int)__umul64hi((uint64_t)p, a->uniform_magic);
    } else {
        int l0 = 0, h0 = a->n;
        while (h0 - l0 > 1) {
            const int mid = (l0 + h0) >> 1;
            if (t->off[mid] <= p) l0 = mid; else h0 = mid;
        }
        fi = l0;
    }
    return t->f[fi];
}

// ---- Rows looked up exactly TWICE (placement plans of nrx_sparse_plan_lds).  The plan leaves one record {unique index, first lookup, second
// lookup} per such row; a lane group takes a record, fetches the two upstream rows and stores 0 + first + second -- the sum the sorted walk forms
// for a two-entry segment, bit for bit.  No order words, no segment bounds: one dependent round trip behind the record.  On uniform ids these rows
// were nearly all of the walk's work (C2: 52 K of 53 K walked rows).  The records ride in the argument block's bag fields (a pair plan has no bag
// feature): a->scale = the records, a->bag_inv = their number (device int64), a->bag_bits = the number of BLOCKS of the launch that take records
// (the walk kernel's first blocks: the pair rows and the walked rows are independent chains of round trips -- in one launch they overlap).
__device__ __forceinline__ const void* pairs_recs(const NRX_CONST SortedBwdArgs* a) { return a->scale; }
__device__ __forceinline__ int pairs_blocks(const NRX_CONST SortedBwdArgs* a) { return (int)reinterpret_cast<intptr_t>(a->bag_bits); }
template <int QLOG2, bool FM, bool UNAL, int DEC>
__device__ __forceinline__ void pairs_body(const NRX_CONST SortedBwdArgs* a, const SortedFeatLds* s_ft, int bx, int gx) {
    constexpr int Q = 1 << QLOG2, G = NRX_BLOCK / Q;
    constexpr bool REG = DEC == 1;
    const uint64_t reg_magic = a->uniform_magic;
    const int64_t reg_len = a->uniform_len;
    const int reg_col0 = a->col0, reg_stride = a->col_stride;
    const bool reg_fm = a->all_fm != 0;
    const int tid = threadIdx.x, q = tid & (Q - 1), grp = tid >> QLOG2;
    using nrx_i32x4 = __attribute__((ext_vector_type(4))) int;
    const NRX_GLOBAL nrx_i32x4* recs = nrx_gconst<nrx_i32x4>(pairs_recs(a));
    const int64_t cap = a->off[a->n] / 2 + 1;
    int64_t i = (int64_t)bx * G + grp;
    nrx_i32x4 rec = {0, 0, 0, 0};
    if (i < cap) rec = recs[i];                         // requested next to the count: a record past the count is read (inside the buffer) and dropped
    int64_t n = nrx_gconst<int64_t>(a->bag_inv)[0];
    n = n < cap ? n : cap;
    // (FM: the row's forward value is fetched with its FIRST lookup and reused for the second -- the same bits for every lookup of a row, and what
    //  the walk does for the rows it reduces: both planners' plans give the same gradient even over a concat that does not belong to these ids)
    float4 v_first = make_float4(0.f, 0.f, 0.f, 0.f);
    auto row_of = [&](int64_t p, bool first) -> float4 {          // the gradient row chunk of lookup p, as the walk forms it
        float4 g, v = v_first, s = make_float4(0.f, 0.f, 0.f, 0.f);
        float gf = 0.f;
        if (REG && !UNAL) {
            const int fi = (int)__umul64hi((uint64_t)p, reg_magic);
            const int64_t b = p - (int64_t)fi * reg_len;
            const int col = reg_col0 + fi * reg_stride;
            g = a->g_out != nullptr ? nrx_ldg4(a->g_out, (b * a->out_ld + col) / 4 + q) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (FM) {
                gf = reg_fm ? nrx_gconst<float>(a->g_fm)[b] : 0.f;
                if (first) v = v_first = nrx_ldg4(a->feat, (b * a->feat_ld + col) / 4 + q);
                s = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
            }
        } else {
            const FeatLite f = sorted_decode<DEC>(a, s_ft, p);
            const int64_t b = p - f.off;
            g = sorted_upstream<UNAL>(a, f, b, q);
            if (FM) {
                gf = f.fm ? nrx_gconst<float>(a->g_fm)[b] : 0.f;
                if (first) v = v_first = nrx_ldg4(a->feat, (b * a->feat_ld + f.out_col) / 4 + q);
                s = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
            }
        }
        if (FM) fm_fold4(g, gf, s, v, q);
        return g;
    };
    for (; i < n; i += (int64_t)gx * G) {
        const int64_t u = rec.x, p1 = (int64_t)(uint32_t)rec.y, p2 = (int64_t)(uint32_t)rec.z;
        const int64_t inext = i + (int64_t)gx * G;
        if (inext < n) rec = recs[inext];
        const int64_t key = a->uniq_keys != nullptr ? nrx_gconst<int64_t>(a->uniq_keys)[u] : 1;
        const float4 t_lo = row_of(p1, true), t_hi = row_of(p2, false);         // (the plan lists the lookups of a row in ascending order)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        acc.x += t_lo.x; acc.y += t_lo.y; acc.z += t_lo.z; acc.w += t_lo.w;
        acc.x += t_hi.x; acc.y += t_hi.y; acc.z += t_hi.z; acc.w += t_hi.w;
        sorted_store4<Q>(a, u, key, q, acc);
    }
}
// (launches with Wide&Deep column routing: the records in a launch of their own)
template <int QLOG2, bool FM, bool UNAL, int DEC>
__global__ __launch_bounds__(NRX_BLOCK) void embed_bwd_pairs_kernel(const SortedBwdArgs args_in_kernarg) {
    const NRX_CONST SortedBwdArgs* a = nrx_kernarg<SortedBwdArgs>();
    __shared__ SortedFeatLds s_ft;
    if (DEC == 0) sorted_feat_stage(a, &s_ft);
    pairs_body<QLOG2, FM, UNAL, DEC>(a, &s_ft, (int)blockIdx.x, (int)gridDim.x);
}

// REG: the launch's features are `regular` (SortedBwdArgs::regular: single-valued, equally long, equally spaced columns, one FM flag) --
// feature, sample and column of a sorted entry are then ARITHMETIC on its lookup number.  As a run-time branch inside the general decode
// (round 2) it cost what the general decode costs: that one reads the per-feature fields from the argument block with vector loads, and with
// both forms in one instruction stream the compiler waits vmcnt(0) in front of every row load -- the 8 row loads of a pass went out ONE AT A
// TIME, each behind the previous one's arrival (seen in the ISA; the C5 walk ran at 1.65 TB/s with 86 % of its wave time waiting).
template <int QLOG2, int R, bool FM, bool BAG, bool UNAL, int UP = 1, int DEC = 0, bool PAIRS = false>
__global__ __launch_bounds__(NRX_BLOCK) void embed_bwd_sorted_fast_kernel(const SortedBwdArgs args_in_kernarg) {
    const NRX_CONST SortedBwdArgs* a = nrx_kernarg<SortedBwdArgs>();
    // (REG) the launch's scalars, read once
    const uint64_t reg_magic = a->uniform_magic;
    const int64_t reg_len = a->uniform_len;
    const int reg_col0 = a->col0, reg_stride = a->col_stride;
    const bool reg_fm = a->all_fm != 0;
    const float* const up_g = a->g_out;
    const int64_t up_ld = a->out_ld;
    __shared__ SortedFeatLds s_ft;
    constexpr bool REG = DEC == 1;
    if (DEC == 0) sorted_feat_stage(a, &s_ft);
    int64_t bx = blockIdx.x, gx = gridDim.x;
    if (PAIRS) {               // the launch's first blocks take the pair records (pairs_body), the others walk
        const int pb = pairs_blocks(a);
        if (bx < pb) {
            pairs_body<QLOG2, FM, UNAL, DEC>(a, &s_ft, (int)bx, pb);
            return;
        }
        bx -= pb;
        gx -= pb;
    }
    const bool bag_binary = BAG && a->bag_bits != nullptr && (a->long_ws[3] & 1) == 0;      // every bag weight is 0 or 1 (bag_scale_kernel)
    const bool need_bits = BAG && a->bag_bits != nullptr && (a->long_ws[3] & 2) != 0;       // some live lookup has weight 0
    constexpr int Q = 1 << QLOG2;
    constexpr int TB = NRX_BLOCK / Q;
    const int q = threadIdx.x & (Q - 1);
    // n = rows of this launch: all unique rows, or (placement mode) the `walk` list -- then u0 + r indexes the list
    int64_t n = a->n_unique;
    const bool listed = a->walk != nullptr;
    {
        const int64_t* ndp = listed ? a->n_walk_dev : a->n_unique_dev;
        if (ndp != nullptr) {
            const int64_t nd = nrx_gconst<int64_t>(ndp)[0];
            n = nd < n ? nd : n;
        }
    }
    // The grid is sized by a host-side BOUND on the row count (the count itself lives on the device): a block takes row groups
    // blockIdx.x, blockIdx.x + gridDim.x, ... -- with one group per block, a C5 launch started ~15 000 blocks that each waited for
    // the count to arrive from memory only to leave.
    auto body = [&](int64_t blk) {
    const int64_t u0 = (blk * TB + (threadIdx.x >> QLOG2)) * R;
    if (__ballot(u0 < n) == 0ull) return;      // whole wavefronts past the last row leave; inside the last live wavefront the
                                               // lane groups past it stay (the work-list append below is a wavefront scan)
    const NRX_GLOBAL int64_t* seg = nrx_gconst<int64_t>(a->seg_start);
    int64_t lo[R], hi[R], key[R], urow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t li = u0 + r < n ? u0 + r : n - 1;
        urow[r] = listed ? (int64_t)nrx_gconst<int32_t>(a->walk)[li] : li;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t u = urow[r];
        lo[r] = seg[u];
        hi[r] = seg[u + 1];
        key[r] = a->uniq_keys != nullptr ? nrx_gconst<int64_t>(a->uniq_keys)[u] : 1;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if ((key[r] & ((1ll << 40) - 1)) == 0 || u0 + r >= n) hi[r] = lo[r];      // padding row: zeros
    // rows looked up many times (hot ids of a skewed distribution, tiny tables) would serialise this lane group for
    // their whole segment: they go to a work list and are reduced by whole wavefronts (sorted_long_kernel)
    // The list positions of a whole wavefront come from ONE atomic: the lanes' item counts are scanned, the last lane adds
    // the total to the item counter and every lane places its rows' items after the lanes below it.  (One atomic per row:
    // every second row of the C4 news table is long -- 100 k round trips to the same counter.)  Rows of several chunks
    // (rare: > 256 entries) still take their partial-sum slots individually.
    bool lng[R];
    int need = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        lng[r] = a->long_ws != nullptr && hi[r] - lo[r] > a->long_t;
        if (lng[r] && q == 0) need += (int)((hi[r] - lo[r] + SORTED_LONG_CHUNK - 1) / SORTED_LONG_CHUNK);
    }
    if (__ballot(need != 0) != 0ull) {                       // wave-uniform
        // Round 6: ONE request per wavefront to the item / multi-row counters (a 64-bit add on the adjacent words long_ws[0] | long_ws[1]) and one more
        // to the slot counter only when the wavefront holds a multi-item row -- the memory side serialises requests to one line at ~40 ns each, and a
        // Zipf batch sent three per wavefront with long rows (C2 Zipf: ~570 requests = 23 of the walk's 44 us).  And the items of a row of many
        // chunks are written by the WHOLE wavefront: the hottest row of C4's Zipf batch is ~960 items, which one lane wrote one after the other.
        const int lane = threadIdx.x & 63;
        int n_multi = 0, n_slots = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (lng[r] && q == 0) {
                const int nchunks = (int)((hi[r] - lo[r] + SORTED_LONG_CHUNK - 1) / SORTED_LONG_CHUNK);
                if (nchunks > 1) { ++n_multi; n_slots += sorted_long_slots(nchunks); }
            }
        }
        unsigned long long packed = (unsigned long long)(uint32_t)need | ((unsigned long long)(uint32_t)n_multi << 32);
        unsigned long long incl = packed;
        int incl_s = n_slots;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long t = __shfl_up(incl, off, 64);
            const int ts = __shfl_up(incl_s, off, 64);
            if (lane >= off) { incl += t; incl_s += ts; }
        }
        unsigned long long base2 = 0;
        int base_s = 0;
        const int tot_s = __shfl(incl_s, 63, 64);
        if (lane == 63) {
            base2 = atomicAdd(reinterpret_cast<unsigned long long*>(a->long_ws), incl);      // (long_ws is 16-byte aligned: [0] items, [1] multi-item rows)
            if (tot_s != 0) base_s = atomicAdd(&a->long_ws[2], tot_s);
        }
        base2 = __shfl(base2, 63, 64) + incl - packed;
        int base = (int)(uint32_t)base2, m_next = (int)(uint32_t)(base2 >> 32);
        int slot_next = __shfl(base_s, 63, 64) + incl_s - n_slots;
        // per row: its first item, its slots, its multi-row record; rows of more than a few chunks are handed to the whole wavefront below
        constexpr int COOP = 4;
        int c_n[R], c_slot0[R], c_base[R], c_m[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            c_n[r] = 0; c_slot0[r] = -1; c_base[r] = 0; c_m[r] = -1;
            if (lng[r] && q == 0) {
                const int nchunks = (int)((hi[r] - lo[r] + SORTED_LONG_CHUNK - 1) / SORTED_LONG_CHUNK);
                int slot0 = -1, m = -1;
                if (nchunks > 1) {
                    slot0 = slot_next;
                    slot_next += sorted_long_slots(nchunks);
                    m = m_next++;
                    if (m < a->long_slots_cap) {
                        LongMulti w;
                        w.u = (int32_t)urow[r]; w.slot0 = slot0; w.nchunks = nchunks; w.done = 0;
                        sorted_long_multi(a)[m] = w;
                    }
                }
                if (nchunks <= COOP) {
                    const int ng = sorted_long_groups(nchunks);      // (0 for so few chunks)
                    (void)ng;
                    sorted_long_write_items(a, (int32_t)urow[r], lo[r], hi[r], nchunks, slot0, base, m);
                } else {
                    c_n[r] = nchunks; c_slot0[r] = slot0; c_base[r] = base; c_m[r] = m;
                }
                base += nchunks;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            unsigned long long todo = __ballot(c_n[r] != 0);
            while (todo != 0ull) {                               // wave-uniform: one row of many chunks at a time, every lane writes items
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int nchunks = __shfl(c_n[r], src, 64), slot0 = __shfl(c_slot0[r], src, 64), b0 = __shfl(c_base[r], src, 64), m = __shfl(c_m[r], src, 64);
                const int32_t u = (int32_t)__shfl(urow[r], src, 64);
                const int64_t rlo = __shfl(lo[r], src, 64), rhi = __shfl(hi[r], src, 64);
                for (int c = lane; c < nchunks; c += 64) {
                    if (b0 + c >= a->long_items_cap) break;
                    LongItem w;
                    w.u = u;
                    w.dest = slot0 + c;
                    w.e_begin = rlo + (int64_t)c * SORTED_LONG_CHUNK;
                    w.len = (int32_t)((w.e_begin + SORTED_LONG_CHUNK < rhi ? w.e_begin + SORTED_LONG_CHUNK : rhi) - w.e_begin);
                    w.m = m;
                    sorted_long_items(a)[b0 + c] = w;
                }
                const int ng = sorted_long_groups(nchunks);
                if (ng != 0 && (int64_t)slot0 + sorted_long_slots(nchunks) <= a->long_slots_cap) {      // the group counters start at zero
                    int32_t* gc = reinterpret_cast<int32_t*>(sorted_long_partials(a) + (int64_t)(slot0 + nchunks + ng) * (4 * Q));
                    for (int j = lane; j < ng; j += 64) gc[j] = 0;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (lng[r]) hi[r] = lo[r];
    float4 acc[R];
    int64_t e[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        e[r] = lo[r];
    }
    // One pass takes the next sorted entry of each of the R rows: R independent chains (lookup index -> upstream row) per
    // lane, added in sorted order.  The lookup indices of the NEXT pass are requested before this pass's rows, so a pass
    // costs one memory round trip, not two.
    // UP sorted entries of each of the R rows per pass (UP = 1: uniform-id launches, where most rows have one entry; UP = 4: bag
    // launches, where a row of the pooled table is looked up ~L times and a one-entry pass made the walk latency-bound:
    // all rows of the launch are in flight at once, so the only parallelism left is INSIDE a row).  The lookup indices of the
    // next pass are requested before this pass's rows: one memory round trip per pass, not two.
    // The UP consecutive order words of a row are ONE request per lane group: lane q fetches word (q mod UP), the group shares them
    // through the cross-lane unit (every lane loading all UP words made UP requests of 8 bytes each).
    constexpr bool WIDE = UP > 1 && UP <= Q;
    const int gbase = (threadIdx.x & 63) & ~(Q - 1);
    int64_t pn[R][WIDE ? 1 : UP];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (WIDE) {
            const int64_t ei = e[r] + (q & (UP - 1));
            pn[r][0] = nrx_gconst<int64_t>(a->order)[ei < hi[r] ? ei : lo[0]];
        } else {
#pragma unroll
            for (int j = 0; j < (WIDE ? 1 : UP); ++j) pn[r][j] = nrx_gconst<int64_t>(a->order)[e[r] + j < hi[r] ? e[r] + j : lo[0]];
        }
    }
    // FM fields: the forward value v of a lookup is its table row -- the SAME bits for every lookup of one unique row (the forward copies rows
    // verbatim) -- so a row's first entry fetches it and the others reuse it: one 128-byte line per ROW instead of per lookup, same fold, same
    // bits (Zipf ids: half of the lookups sit on rows looked up more than 16 times; the walk and the work lists read a third less).
    float4 vrow[R];
    bool have_v = false;
    auto pass = [&]() -> bool {
        int64_t p[R][UP];
        bool on[R][UP];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < UP; ++j) {
                on[r][j] = e[r] + j < hi[r];
                p[r][j] = WIDE ? __shfl(pn[r][0], gbase + j, 64) : pn[r][WIDE ? 0 : j];
            }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (WIDE) {
                const int64_t ei = e[r] + UP + (q & (UP - 1));
                if (ei < hi[r]) pn[r][0] = nrx_gconst<int64_t>(a->order)[ei];
            } else {
#pragma unroll
                for (int j = 0; j < (WIDE ? 1 : UP); ++j)
                    if (e[r] + UP + j < hi[r]) pn[r][j] = nrx_gconst<int64_t>(a->order)[e[r] + UP + j];      // (one-entry rows request nothing)
            }
        }
        float4 g[R][UP], v[R][UP], s[R][UP];
        float gf[R][UP], sc[R][UP];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < UP; ++j) {
                if (REG && !BAG && !UNAL) {
                    const int fi = (int)__umul64hi((uint64_t)p[r][j], reg_magic);
                    const int64_t b = p[r][j] - (int64_t)fi * reg_len;
                    const int col = reg_col0 + fi * reg_stride;
                    sc[r][j] = 1.0f;
                    g[r][j] = up_g != nullptr ? nrx_ldg4(up_g, (b * up_ld + col) / 4 + q) : make_float4(0.f, 0.f, 0.f, 0.f);      // (no g_out: an FM model whose loss reads the logit only)
                    if (FM) {
                        gf[r][j] = reg_fm ? nrx_gconst<float>(a->g_fm)[b] : 0.f;
                        if (j == 0 && !have_v) v[r][0] = nrx_ldg4(a->feat, (b * a->feat_ld + col) / 4 + q);
                        s[r][j] = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
                    }
                    continue;
                }
                const FeatLite f = sorted_decode<DEC>(a, &s_ft, p[r][j]);
                int64_t b;
                if (BAG) {
                    sorted_bag_entry<UNAL>(a, f, p[r][j], q, Q, bag_binary, need_bits, b, g[r][j], sc[r][j]);
                } else {
                    b = p[r][j] - f.off;
                    sc[r][j] = 1.0f;
                    g[r][j] = sorted_upstream<UNAL>(a, f, b, q);
                }
                if (FM) {
                    gf[r][j] = f.fm ? nrx_gconst<float>(a->g_fm)[b] : 0.f;
                    if (j == 0 && !have_v) v[r][0] = nrx_ldg4(a->feat, (b * a->feat_ld + f.out_col) / 4 + q);
                    s[r][j] = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
                }
            }
        if (FM && !have_v) {
#pragma unroll
            for (int r = 0; r < R; ++r) vrow[r] = v[r][0];
            have_v = true;
        }
        bool more = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < UP; ++j) {
                float4 t = g[r][j];
                if (FM) fm_fold4(t, gf[r][j], s[r][j], vrow[r], q);      // d fm / d field: column 0 -> 1, column k -> S_k - v_k
                if (on[r][j]) {               // added in sorted order: j ascending inside the pass
                    if (BAG) {
#pragma clang fp contract(off)
                        acc[r].x += t.x * sc[r][j]; acc[r].y += t.y * sc[r][j]; acc[r].z += t.z * sc[r][j]; acc[r].w += t.w * sc[r][j];
                    } else {
                        acc[r].x += t.x; acc[r].y += t.y; acc[r].z += t.z; acc[r].w += t.w;
                    }
                }
            }
            e[r] += UP;
            more |= e[r] < hi[r];
        }
        return more;
    };
    // Staged form (a->gs_all: every feature's upstream rows, already scaled, in one array): the lane that fetched an order word
    // also DECODES it -- feature, sample, row of the staging array, weight bit -- and the group exchanges finished 32-bit row
    // numbers.  In the general pass every lane decodes all R x UP entries of its group (the same work in each of the Q lanes):
    // ~100 vector instructions per entry; the C4 walk was bound by them once its memory requests had been cut.
    auto pass_staged = [&]() -> bool {
        int32_t mine[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t ei = e[r] + (q & (UP - 1));
            const int64_t pw = pn[r][0];
            int32_t id = -1;
            if (ei < hi[r]) {
                const FeatLite f = sorted_decode<DEC>(a, &s_ft, pw);
                uint64_t b = (uint64_t)(pw - f.off);
                if (f.bag_len > 1) b = __umul64hi(b, f.magic);
                id = (int32_t)((f.gs - a->gs_all) >> (QLOG2 + 2)) + (int32_t)b;
                if (need_bits && f.kind >= NRX_BAG_MASKED_MEAN &&
                    ((nrx_gconst<uint32_t>(a->bag_bits)[pw >> 5] >> (pw & 31)) & 1u) == 0) id = -1;
            }
            mine[r] = id;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t ei = e[r] + UP + (q & (UP - 1));
            if (ei < hi[r]) pn[r][0] = nrx_gconst<int64_t>(a->order)[ei];
        }
        int32_t id[R][UP];
        float4 g[R][UP];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < UP; ++j) id[r][j] = __shfl(mine[r], gbase + j, 64);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < UP; ++j)
                if (id[r][j] >= 0) g[r][j] = nrx_ldg4(a->gs_all, (int64_t)id[r][j] * Q + q);
        bool more = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < UP; ++j)
                if (id[r][j] >= 0) {          // added in sorted order: j ascending inside the pass
                    acc[r].x += g[r][j].x; acc[r].y += g[r][j].y; acc[r].z += g[r][j].z; acc[r].w += g[r][j].w;
                }
            e[r] += UP;
            more |= e[r] < hi[r];
        }
        return more;
    };
    bool more = true;
    if (BAG && WIDE && bag_binary && a->gs_all != nullptr) {
        while (more) more = pass_staged();
    } else {
        while (more) more = pass();
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (u0 + r < n && !lng[r]) sorted_store4<Q>(a, urow[r], key[r], q, acc[r]);
    };
    for (int64_t blk = bx; blk * (TB * R) < n; blk += gx) body(blk);
}

// Partial sums of multi-item rows cross XCDs inside ONE launch (the item that finishes a row last may run on another XCD than the items that
// left the partials): agent-scope accesses -- the store is written through the XCD's L2, the load does not hit in it.  16 bytes as four dwords
// (the atomic builtins take scalars; a few hundred partials per launch).
__device__ __forceinline__ void long_partial_publish(float* base, int64_t i, float4 v) {
    float* p = base + i * 4;
    __hip_atomic_store(p + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 long_partial_fetch(const float* base, int64_t i) {
    const float* p = base + i * 4;
    float4 v;
    v.x = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// Long segments.  An item = up to SORTED_LONG_CHUNK consecutive sorted entries of ONE unique row; a wavefront reduces an
// item: its 64 / Q lane groups stride over the entries (two in flight each), then a fixed xor-shuffle tree adds the
// groups -- the same entries always meet in the same order, so the result is reproducible.  A row of one item is written
// straight to values; a row of several items leaves one partial per item, which the wavefront finishing the row's last item adds in item order.
template <int QLOG2, bool FM, bool BAG, bool UNAL, int DEC = 0>      // DEC: as in embed_bwd_sorted_fast_kernel
__global__ __launch_bounds__(NRX_BLOCK) void sorted_long_kernel(const SortedBwdArgs args_in_kernarg) {
    const NRX_CONST SortedBwdArgs* a = nrx_kernarg<SortedBwdArgs>();
    const uint64_t reg_magic = a->uniform_magic;
    const int64_t reg_len = a->uniform_len;
    const int reg_col0 = a->col0, reg_stride = a->col_stride;
    const bool reg_fm = a->all_fm != 0;
    const float* const up_g = a->g_out;
    const int64_t up_ld = a->out_ld;
    __shared__ SortedFeatLds s_ft;
    constexpr bool REG = DEC == 1;
    if (DEC == 0) sorted_feat_stage(a, &s_ft);
    const bool bag_binary = BAG && a->bag_bits != nullptr && (a->long_ws[3] & 1) == 0;      // every bag weight is 0 or 1 (bag_scale_kernel)
    const bool need_bits = BAG && a->bag_bits != nullptr && (a->long_ws[3] & 2) != 0;
    constexpr int Q = 1 << QLOG2, G = 64 / Q;
    const int lane = threadIdx.x & 63, q = lane & (Q - 1), g = lane >> QLOG2;
    const int nitems = a->long_ws[0] < a->long_items_cap ? a->long_ws[0] : (int)a->long_items_cap;
    const LongItem* items = sorted_long_items(a);
    float* partial = sorted_long_partials(a);
    const int nwaves = gridDim.x * (NRX_BLOCK / 64);
    for (int it = blockIdx.x * (NRX_BLOCK / 64) + (threadIdx.x >> 6); it < nitems; it += nwaves) {
        const LongItem w = items[it];
        const int64_t w_end = w.e_begin + w.len;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // The item's sorted entries in chunks of 64: ONE order word per lane and chunk (the next chunk's requested before this chunk's rows),
        // handed to the lane groups through the cross-lane unit; group g takes entries g, g + G, g + 2G, ... of the item, as before (same
        // accumulation order per group, same tree over the groups: same bits).  Round 2's loop fetched the order words of 4 entries per
        // group on demand: two dependent round trips per 4 G entries.
        if (BAG) {               // bag launches keep round 2's loop (order words fetched per lane group, 4 entries in flight): the chunked form measured
                                 // 5-10 us slower on the C4 Zipf work list (50.6 -> 61.5 us for the launch)
            constexpr int UB = 4;
            for (int64_t e0 = w.e_begin + g; e0 < w_end; e0 += UB * G) {
                int64_t p[UB];
                bool on[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    on[k] = e0 + k * G < w_end;
                    p[k] = nrx_gconst<int64_t>(a->order)[on[k] ? e0 + k * G : w.e_begin];
                }
                float4 gr[UB];
                float sc[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    const FeatLite f = sorted_decode<DEC>(a, &s_ft, p[k]);
                    int64_t b;
                    sorted_bag_entry<UNAL>(a, f, p[k], q, Q, bag_binary, need_bits, b, gr[k], sc[k]);
                }
#pragma unroll
                for (int k = 0; k < UB; ++k)
                    if (on[k]) {
#pragma clang fp contract(off)
                        acc.x += gr[k].x * sc[k]; acc.y += gr[k].y * sc[k]; acc.z += gr[k].z * sc[k]; acc.w += gr[k].w * sc[k];
                    }
            }
        } else {
        constexpr int UL = Q < 8 ? Q : 8;                       // rows in flight per lane group (a chunk gives every group Q entries)
        float4 vrow = make_float4(0.f, 0.f, 0.f, 0.f);
        bool have_v = false;
        int64_t pw_next = nrx_gconst<int64_t>(a->order)[w.e_begin + lane < w_end ? w.e_begin + lane : w.e_begin];
        for (int64_t c0 = w.e_begin; c0 < w_end; c0 += 64) {
            const int64_t pw = pw_next;
            if (c0 + 64 < w_end) pw_next = nrx_gconst<int64_t>(a->order)[c0 + 64 + lane < w_end ? c0 + 64 + lane : w.e_begin];
#pragma unroll
            for (int kb = 0; kb < Q; kb += UL) {
                if (c0 + (int64_t)kb * G >= w_end) break;               // wave-uniform (the sub-batch's first entry): every lane stays for the shuffles below
                int64_t p[UL];
                bool on[UL];
#pragma unroll
                for (int k = 0; k < UL; ++k) {
                    const int idx = g + (kb + k) * G;
                    on[k] = c0 + idx < w_end;
                    p[k] = __shfl(pw, idx, 64);
                }
                float4 gr[UL], s_[UL];
                float gf[UL], sc[UL];
#pragma unroll
                for (int k = 0; k < UL; ++k) {
                    if (REG && !BAG && !UNAL) {
                        const int fi = (int)__umul64hi((uint64_t)p[k], reg_magic);
                        const int64_t b = p[k] - (int64_t)fi * reg_len;
                        const int col = reg_col0 + fi * reg_stride;
                        sc[k] = 1.0f;
                        gr[k] = up_g != nullptr ? nrx_ldg4(up_g, (b * up_ld + col) / 4 + q) : make_float4(0.f, 0.f, 0.f, 0.f);
                        if (FM) {
                            gf[k] = reg_fm ? nrx_gconst<float>(a->g_fm)[b] : 0.f;
                            if (k == 0 && !have_v) vrow = nrx_ldg4(a->feat, (b * a->feat_ld + col) / 4 + q);      // (the item's row: one forward value for all its entries)
                            s_[k] = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
                        }
                        continue;
                    }
                    const FeatLite f = sorted_decode<DEC>(a, &s_ft, p[k]);
                    int64_t b;
                    if (BAG) {
                        sorted_bag_entry<UNAL>(a, f, p[k], q, Q, bag_binary, need_bits, b, gr[k], sc[k]);
                    } else {
                        b = p[k] - f.off;
                        sc[k] = 1.0f;
                        gr[k] = sorted_upstream<UNAL>(a, f, b, q);
                    }
                    if (FM) {
                        gf[k] = f.fm ? nrx_gconst<float>(a->g_fm)[b] : 0.f;
                        if (k == 0 && !have_v) vrow = nrx_ldg4(a->feat, (b * a->feat_ld + f.out_col) / 4 + q);
                        s_[k] = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
                    }
                }
                have_v = true;
#pragma unroll
                for (int k = 0; k < UL; ++k) {
                    float4 t = gr[k];
                    if (FM) fm_fold4(t, gf[k], s_[k], vrow, q);
                    if (on[k]) {
                        if (BAG) {
#pragma clang fp contract(off)
                            acc.x += t.x * sc[k]; acc.y += t.y * sc[k]; acc.z += t.z * sc[k]; acc.w += t.w * sc[k];
                        } else {
                            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                        }
                    }
                }
            }
        }
        }
#pragma unroll
        for (int off = Q; off < 64; off <<= 1) {             // add the G lane groups: fixed tree
            acc.x += __shfl_xor(acc.x, off, 64);
            acc.y += __shfl_xor(acc.y, off, 64);
            acc.z += __shfl_xor(acc.z, off, 64);
            acc.w += __shfl_xor(acc.w, off, 64);
        }
        if (g == 0) {
            if (w.dest < 0) sorted_store4<Q>(a, w.u, a->dense ? nrx_gconst<int64_t>(a->uniq_keys)[w.u] : 0, q, acc);
#ifdef NRX_COMBINE_SEPARATE
            else if (w.dest < a->long_slots_cap) nrx_stg4(partial, (int64_t)w.dest * Q + q, acc);
#else
            else if (w.dest < a->long_slots_cap) long_partial_publish(partial, (int64_t)w.dest * Q + q, acc);
#endif
        }
        // A row of several items: the wavefront that finishes the row's LAST item adds the partials -- its lane groups stride over them in item
        // order, then the same fixed xor-shuffle tree as above: whichever wavefront comes last forms the same sum.  (Was a launch of its own,
        // sorted_combine_kernel: 4.6 us on every step to find, on uniform ids, an empty list.)
        // The hand-over costs NO fence.  Round 5 had `__threadfence()` on both sides of the count: at agent scope that is buffer_wbl2 + buffer_inv
        // -- a write-back of every dirty line of the XCD's L2 (this launch's own gradient rows) and an invalidate, per ITEM: Zipf ids have
        // thousands of multi-item rows and the launch went 54 -> 187 us (C2) / 43 -> 331 us (C4; profiles/r06_zipf_regression.txt).  Instead the
        // partials alone are written THROUGH the L2 (agent-scope stores: sc1), the wavefront waits for their acknowledgement (vmcnt(0)) before
        // its lane 0 counts the item (a device-scope atomic: performed at the memory side), and the last finisher -- whose loads are issued
        // behind the returned count -- reads them with agent-scope loads, which do not hit in its own XCD's L2 (the slots are reused every step).
#ifndef NRX_COMBINE_SEPARATE
        if (w.dest >= 0 && w.m >= 0 && w.m < a->long_slots_cap) {            // wave-uniform
            LongMulti* mrow = sorted_long_multi(a) + w.m;
            const int nch = mrow->nchunks, slot0 = mrow->slot0;
            // `count` partials from slot `first` on, added by this wavefront: its lane groups stride over them (four in flight each), then the fixed tree
            auto add_partials = [&](int first, int count) -> float4 {
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((int64_t)first + count > a->long_slots_cap) count = (int)a->long_slots_cap - first;
                for (int c = g; c < count; c += 4 * G) {
                    float4 tt[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        tt[k] = long_partial_fetch(partial, (int64_t)(first + (c + k * G < count ? c + k * G : c)) * Q + q);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c + k * G < count) { sum.x += tt[k].x; sum.y += tt[k].y; sum.z += tt[k].z; sum.w += tt[k].w; }
                }
#pragma unroll
                for (int off = Q; off < 64; off <<= 1) {
                    sum.x += __shfl_xor(sum.x, off, 64);
                    sum.y += __shfl_xor(sum.y, off, 64);
                    sum.z += __shfl_xor(sum.z, off, 64);
                    sum.w += __shfl_xor(sum.w, off, 64);
                }
                return sum;
            };
            auto count_one = [&](int32_t* counter) -> int {                  // this wavefront's writes have reached memory before the count says so
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return __shfl(old, 0, 64);
            };
            const int ng = sorted_long_groups(nch);
            bool last = false;
            int first = slot0, count = nch;
            if (ng == 0) {
                last = count_one(&mrow->done) == nch - 1;
            } else if ((int64_t)slot0 + sorted_long_slots(nch) <= a->long_slots_cap) {
                int32_t* gc = reinterpret_cast<int32_t*>(partial + (int64_t)(slot0 + nch + ng) * (4 * Q));
                const int grp = (w.dest - slot0) / SORTED_LONG_GROUP;
                const int gsz = nch - grp * SORTED_LONG_GROUP < SORTED_LONG_GROUP ? nch - grp * SORTED_LONG_GROUP : SORTED_LONG_GROUP;
                if (count_one(gc + grp) == gsz - 1) {                        // this group's last item: its partials -> the group's partial
                    const float4 gsum = add_partials(slot0 + grp * SORTED_LONG_GROUP, gsz);
                    if (g == 0) long_partial_publish(partial, (int64_t)(slot0 + nch + grp) * Q + q, gsum);
                    last = count_one(&mrow->done) == ng - 1;
                    first = slot0 + nch;
                    count = ng;
                }
            }
            if (last) {
                const float4 sum = add_partials(first, count);
                if (g == 0) sorted_store4<Q>(a, w.u, a->dense ? nrx_gconst<int64_t>(a->uniq_keys)[w.u] : 0, q, sum);
            }
        }
#endif
    }
}

#ifdef NRX_COMBINE_SEPARATE
// A row of several items: a wavefront adds its partials -- the 64 / Q lane groups stride over them in item order, then the
// same fixed xor-shuffle tree as above.
template <int QLOG2>
__global__ __launch_bounds__(NRX_BLOCK) void sorted_combine_kernel(const SortedBwdArgs args_in_kernarg) {
    const NRX_CONST SortedBwdArgs* a = nrx_kernarg<SortedBwdArgs>();
    constexpr int Q = 1 << QLOG2, G = 64 / Q;
    const int lane = threadIdx.x & 63, q = lane & (Q - 1), g = lane >> QLOG2;
    const int nmulti = a->long_ws[1] < a->long_slots_cap ? a->long_ws[1] : (int)a->long_slots_cap;
    const LongMulti* multi = sorted_long_multi(a);
    const float* partial = sorted_long_partials(a);
    const int nwaves = gridDim.x * (NRX_BLOCK / 64);
    for (int m = blockIdx.x * (NRX_BLOCK / 64) + (threadIdx.x >> 6); m < nmulti; m += nwaves) {
        const LongMulti w = multi[m];
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = g; c < w.nchunks; c += G) {
            if (w.slot0 + c >= a->long_slots_cap) break;
            const float4 tt = nrx_ldg4(partial, (int64_t)(w.slot0 + c) * Q + q);
            sum.x += tt.x; sum.y += tt.y; sum.z += tt.z; sum.w += tt.w;
        }
#pragma unroll
        for (int off = Q; off < 64; off <<= 1) {
            sum.x += __shfl_xor(sum.x, off, 64);
            sum.y += __shfl_xor(sum.y, off, 64);
            sum.z += __shfl_xor(sum.z, off, 64);
            sum.w += __shfl_xor(sum.w, off, 64);
        }
        if (g == 0) sorted_store4<Q>(a, w.u, a->dense ? nrx_gconst<int64_t>(a->uniq_keys)[w.u] : 0, q, sum);
    }
}
#endif

// Per-lookup factor of a bag feature's upstream row (what the general kernel recomputes per lookup from the whole row of
// weights): w / (sum_l w + 1e-8) (masked mean, base_model.py:278-282), 1 / L (mean) or w (sum).  16 lanes per sample.
__global__ __launch_bounds__(NRX_BLOCK) void bag_scale_kernel(const float* __restrict__ w, int kind, int64_t batch, int L,
                                                            float* __restrict__ scale, float* __restrict__ inv, uint32_t* __restrict__ bits,
                                                            int64_t base, int32_t* __restrict__ nonbinary, const void* __restrict__ ids,
                                                            int idx64, const float* __restrict__ g_out, int64_t out_ld, int out_col, int D,
                                                            float* __restrict__ gs) {
    // flags in *nonbinary: bit 0 = some weight is neither 0 nor 1; bit 1 = some LIVE lookup (id != 0; any lookup when ids is null)
    // has weight 0 -- only then does the reduction need the bit words next to the pre-scaled rows.
    // gs (optional, [batch, D]): the sample's upstream row times `one`, for the reduction's one-request-per-lookup form.
    // also leaves the compact form: inv[b] = factor of a weight-1 lookup of sample b, bits = one bit per lookup (weight != 0;
    // `bits` pre-zeroed, indexed by the flat lookup base + b L + l), and raises *nonbinary when some weight is neither 0 nor 1.
    // DataReader's masks are 0/1 (data_reader.py:96-109), and then the reduction reads 0.7 MB of L2-resident words instead of
    // one random 4-byte scale (= a 128-byte line) per lookup.
    const int q = threadIdx.x & 15;
    const int64_t b = ((int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x) >> 4;
    if (b >= batch) return;                                   // whole 16-lane groups leave together
    float den = 0.f;
    // L <= 64 (every reference shape): a lane's up to four weights are loaded TOGETHER and kept for the second loop -- the two loops each
    // put one dependent round trip per 16 entries in front of the next
    constexpr int WK = 4;
    float wk[WK];
    const bool keep = L <= 16 * WK && w != nullptr && kind != NRX_BAG_MEAN;
    if (keep) {
#pragma unroll
        for (int i = 0; i < WK; ++i) {
            const int l = q + 16 * i;
            wk[i] = w[b * L + (l < L ? l : L - 1)];
        }
    }
    if (kind == NRX_BAG_MASKED_MEAN) {
        if (keep) {
#pragma unroll
            for (int i = 0; i < WK; ++i) den += q + 16 * i < L ? wk[i] : 0.f;
        } else {
            for (int l = q; l < L; l += 16) den += w[b * L + l];
        }
        den = group_sum<16>(den) + 1e-8f;
    }
    const float one = kind == NRX_BAG_MASKED_MEAN ? 1.0f / den : (kind == NRX_BAG_MEAN ? 1.0f / (float)L : 1.0f);
    if (q == 0) inv[b] = one;
    if (gs != nullptr) {
        for (int k = q; k < D / 4; k += 16) {
            const float4 t = nrx_ldg4(g_out, (b * out_ld + out_col) / 4 + k);
            nrx_stg4(gs, b * (D / 4) + k, make_float4(t.x * one, t.y * one, t.z * one, t.w * one));
        }
    }
    const int gsh = (threadIdx.x & 63) & ~15;                 // this group's first lane inside the wavefront
    bool odd = false, zero_live = false;
    for (int l0 = 0; l0 < L; l0 += 16) {
        const int l = l0 + q;
        const bool in = l < L;
        float wv = 1.0f;
        if (keep) {
            const int i = l0 >> 4;
            wv = in ? (i == 0 ? wk[0] : i == 1 ? wk[1] : i == 2 ? wk[2] : wk[3]) : 1.0f;
        } else if (in && w != nullptr && kind != NRX_BAG_MEAN) {
            wv = w[b * L + l];
        }
        if (in) {
            float v;
            if (kind == NRX_BAG_MASKED_MEAN) v = wv / den;
            else if (kind == NRX_BAG_MEAN) v = 1.0f / (float)L;
            else v = wv;
            scale[b * L + l] = v;
        }
        odd |= in && wv != 0.0f && wv != 1.0f;
        if (in && wv == 0.0f) zero_live |= ids == nullptr || nrx_load_id(ids, b * L + l, idx64 != 0) != 0;
        const uint32_t m = (uint32_t)((__ballot(in && wv != 0.0f) >> gsh) & 0xffffull);     // the group's 16 lookups
        if (q == 0 && m != 0u) {
            const int64_t p0 = base + b * L + l0;
            const int sh = (int)(p0 & 31);
            atomicOr(&bits[p0 >> 5], m << sh);
            if (sh > 16) atomicOr(&bits[(p0 >> 5) + 1], m >> (32 - sh));
        }
    }
    if (__ballot(odd) != 0ull && (threadIdx.x & 63) == 0) atomicOr(nonbinary, 1);
    if (__ballot(zero_live) != 0ull && (threadIdx.x & 63) == 0) atomicOr(nonbinary, 2);
}

// Staging array of a bag launch (SortedBwdArgs::gs_all): the rows of its SINGLE-VALUED features are plain copies of their g_out
// columns (block y of the grid = the y-th such feature).
struct StageArgs {
    int32_t out_col[NRX_MAX_FEATURES];
    int32_t block[NRX_MAX_FEATURES];       // the feature's block number in the staging array (its index in the launch)
};
__global__ __launch_bounds__(NRX_BLOCK) void stage_rows_kernel(const StageArgs args_in_kernarg, const float* __restrict__ g_out, int64_t out_ld,
                                                             int64_t batch, int d4, float* __restrict__ gs) {
    const NRX_CONST StageArgs* a = nrx_kernarg<StageArgs>();
    const int y = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * NRX_BLOCK + threadIdx.x;      // float4 index inside the feature's block
    if (i >= batch * d4) return;
    const int64_t b = i / d4;
    const int k = (int)(i - b * d4);
    nrx_stg4(gs, ((int64_t)a->block[y] * batch + b) * d4 + k, nrx_ldg4(g_out, (b * out_ld + a->out_col[y]) / 4 + k));
}

// ----------------------------------------------------------------------------------- host side
int ceil_log2(int x) {
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

int pack_features(const nrx_feature_t* feats, int32_t n, EmbedArgs& a, int& max_dim, int& max_bag, const char* who) {
    max_dim = 1;
    max_bag = 0;
    for (int i = 0; i < n; ++i) {
        const nrx_feature_t& s = feats[i];
        NRX_REQUIRE(s.kind >= NRX_SPARSE && s.kind <= NRX_BAG_SUM, "%s: feature %d: bad kind %d", who, i, s.kind);
        NRX_REQUIRE(s.index != nullptr, "%s: feature %d: null index pointer", who, i);
        NRX_REQUIRE(s.index_bits == 32 || s.index_bits == 64, "%s: feature %d: index_bits must be 32 or 64", who, i);
        NRX_REQUIRE(s.dim >= 1 && s.dim <= 32767, "%s: feature %d: dim %d out of range", who, i, s.dim);
        if (s.kind == NRX_DENSE) {
            NRX_REQUIRE(s.dim == 1, "%s: feature %d: dense features have dim 1", who, i);
        } else {
            NRX_REQUIRE(s.table != nullptr, "%s: feature %d: null table pointer", who, i);
            NRX_REQUIRE(s.rows >= 1 && s.rows <= 0x7fffffffLL, "%s: feature %d: rows %lld out of range", who, i, (long long)s.rows);
        }
        if (s.kind >= NRX_BAG_MASKED_MEAN) {
            NRX_REQUIRE(s.bag_len >= 1 && s.bag_len <= 32767, "%s: feature %d: bag_len %d out of range", who, i, s.bag_len);
            NRX_REQUIRE(s.kind != NRX_BAG_MASKED_MEAN || s.weight != nullptr, "%s: feature %d: masked mean needs weights", who, i);
            NRX_REQUIRE(!(s.flags & NRX_FEAT_BAG_CSR) || s.weight != nullptr, "%s: feature %d: CSR bag without offsets", who, i);
        } else {
            NRX_REQUIRE(!(s.flags & NRX_FEAT_BAG_CSR), "%s: feature %d: NRX_FEAT_BAG_CSR on a non-bag feature", who, i);
        }
        NRX_REQUIRE(s.out_col >= 0, "%s: feature %d: negative out_col", who, i);
        FeatDev& d = a.f[i];
        d.table = s.table;
        d.index = s.index;
        d.weight = (s.kind == NRX_BAG_MEAN && !(s.flags & NRX_FEAT_BAG_CSR)) ? nullptr : s.weight;
        d.rows = s.rows;
        d.out_col = s.out_col;
        d.wide_col = s.wide_col;
        d.dim = (int16_t)s.dim;
        d.bag_len = (int16_t)s.bag_len;
        d.kind = (uint8_t)s.kind;
        d.idx64 = s.index_bits == 64;
        d.fm = s.fm_field != 0;
        d.flags = (uint8_t)(s.flags & (NRX_FEAT_ROW0_IS_DATA | NRX_FEAT_BAG_CSR));
        if (s.dim > max_dim) max_dim = s.dim;
        if (s.kind >= NRX_BAG_MASKED_MEAN && s.bag_len > max_bag) max_bag = s.bag_len;
    }
    return NRX_OK;
}

#define NRX_QSWITCH(qlog2, ...)               \
    switch (qlog2) {                           \
        case 0: { constexpr int QL = 0; __VA_ARGS__; } break; \
        case 1: { constexpr int QL = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int QL = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int QL = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int QL = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int QL = 5; __VA_ARGS__; } break; \
        default: { constexpr int QL = 6; __VA_ARGS__; } break; \
    }

// pick Q (lanes per sample) and the LDS bag chunk for the generic kernels
void plan_generic(int max_dim, int max_bag, int& qlog2, int& lds_chunk, size_t& smem) {
    qlog2 = ceil_log2((max_dim + 3) / 4);
    if (qlog2 > 6) qlog2 = 6;
    const int tb = NRX_BLOCK >> qlog2;
    lds_chunk = 0;
    smem = 0;
    if (max_bag > 0) {
        // keep the staging tile <= 32 KiB so >= 4 blocks stay resident per CU
        int cap = (32 * 1024) / (tb * (int)sizeof(BagPair));
        if (cap < 8) cap = 8;
        lds_chunk = max_bag < cap ? max_bag : cap;
        smem = (size_t)tb * (lds_chunk | 1) * sizeof(BagPair) + (size_t)(tb + 1) * sizeof(int64_t);     // + the CSR offsets of the block's samples
    }
}

// Uniform features (all single-valued, one D = 4Q): the ring kernel of nrx_embed_ring.h.  R = rows in flight per lane.
template <int QLOG2, int R, bool NT>
void launch_ring_r(const UniformArgs& ua, int64_t batch, bool fm, bool store, hipStream_t st) {
    constexpr int TB = NRX_BLOCK >> QLOG2;
    const dim3 grid((unsigned)((batch + TB - 1) / TB)), block(NRX_BLOCK);
    const size_t lds = (size_t)ua.n * TB * sizeof(int32_t);
    if (fm && store) hipLaunchKernelGGL((embed_fwd_ring<QLOG2, R, true, true, NT>), grid, block, lds, st, ua);
    else if (fm) hipLaunchKernelGGL((embed_fwd_ring<QLOG2, R, true, false, NT>), grid, block, lds, st, ua);
    else hipLaunchKernelGGL((embed_fwd_ring<QLOG2, R, false, true, NT>), grid, block, lds, st, ua);
}

template <int QLOG2>
void launch_uniform(const UniformArgs& ua, int64_t batch, bool fm, bool store, hipStream_t st) {
    const int n = ua.n;
    // non-temporal row loads once the launch's tables exceed the 256 MiB Infinity Cache (cache-resident tables lose
    // 15-30 % with them, DRAM-resident ones gain 5-8 %)
    int64_t table_bytes = 0;
    for (int i = 0; i < n; ++i) table_bytes += ua.rows[i] * (int64_t)(16 << QLOG2);
    const bool nt = table_bytes > (256ll << 20);
#define NRX_LR(R_) (nt ? launch_ring_r<QLOG2, R_, true>(ua, batch, fm, store, st) : launch_ring_r<QLOG2, R_, false>(ua, batch, fm, store, st))
    // ring depth: 8 rows in flight per lane, except 64-byte rows streamed from DRAM (C2), where 4 measured 2-3 % faster
    // with a recycled output buffer (57.7 vs 59.1 us; profiles/r02_c2_ring_sweep.md)
    if (n >= 8 && !(QLOG2 == 2 && nt)) NRX_LR(8);
    else if (n >= 4) NRX_LR(4);
    else NRX_LR(1);
#undef NRX_LR
}

}  // namespace

// Largest batch that takes the one-block-per-sample kernel.  Process-wide, settable at run time (nrx_set_small_batch_max) so that the
// parity tests can put BOTH kernel families in front of the oracle on the same shapes; first read comes from NRX_SMALL_BATCH.
static std::atomic<int64_t> g_small_batch_max{-1};
static int64_t small_batch_max() {
    int64_t v = g_small_batch_max.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("NRX_SMALL_BATCH");
        v = e ? atoll(e) : 2048;
        if (v < 0) v = 0;
        g_small_batch_max.store(v, std::memory_order_relaxed);
    }
    return v;
}

extern "C" int64_t nrx_set_small_batch_max(int64_t max_batch) {
    const int64_t prev = small_batch_max();
    if (max_batch >= 0) g_small_batch_max.store(max_batch, std::memory_order_relaxed);
    return prev;
}

extern "C" int nrx_embed_fwd(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                             float* out, int64_t out_ld, float* wide_out, int64_t wide_ld,
                             float* fm_out, int32_t* status, void* stream) {
    NRX_TRACE();
    return nrx_embed_fwd_train(feats, n_feats, batch, out, out_ld, wide_out, wide_ld, fm_out, nullptr, 0, status, stream);
}

extern "C" int nrx_embed_fwd_train(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                                   float* out, int64_t out_ld, float* wide_out, int64_t wide_ld,
                                   float* fm_out, float* fm_sums, int64_t sums_ld, int32_t* status, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feats != nullptr && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES,
                "nrx_embed_fwd: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(batch >= 0, "nrx_embed_fwd: negative batch");
    NRX_REQUIRE(out != nullptr || fm_out != nullptr || wide_out != nullptr, "nrx_embed_fwd: no output requested");
    if (batch == 0) return NRX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    for (int i = 0; i < n_feats; ++i)
        NRX_REQUIRE(!(feats[i].flags & NRX_FEAT_BAG_CSR) || feats[i].kind >= NRX_BAG_MASKED_MEAN,
                    "nrx_embed_fwd: feature %d: NRX_FEAT_BAG_CSR on a non-bag feature", i);
    // ---- small batches: one block per sample (embed_fwd_small_kernel).  NRX_SMALL_BATCH = largest batch that takes it (0: never)
    {
        bool ok = batch <= small_batch_max();
        int64_t items = 0, outs = 0;
        int fm_fields = 0, fm_dim = 0;
        for (int i = 0; i < n_feats && ok; ++i) {
            const nrx_feature_t& s = feats[i];
            ok = s.wide_col < 0 && !(s.flags & NRX_FEAT_BAG_CSR) && s.kind >= NRX_SPARSE && s.kind <= NRX_BAG_SUM;
            if (s.kind != NRX_DENSE) ok = ok && s.table != nullptr && (s.dim & 3) == 0 && nrx_aligned16(s.table) && s.dim >= 4;
            if (s.kind == NRX_BAG_MASKED_MEAN) ok = ok && s.weight != nullptr;
            const int64_t L = s.kind >= NRX_BAG_MASKED_MEAN ? s.bag_len : 1, c = s.kind == NRX_DENSE ? 1 : s.dim / 4;
            items += L * c;
            outs += c;
            if (s.fm_field) {
                ok = ok && s.kind != NRX_DENSE && (fm_fields == 0 || fm_dim == s.dim) && s.dim <= 256;
                fm_dim = s.dim;
                ++fm_fields;
            }
        }
        ok = ok && items <= 2048 && (fm_fields == 0 || fm_out != nullptr) && !(out == nullptr && fm_fields == 0);
        // the block's LDS image (dynamic part below + the kernel's static descriptor copy) must fit a block's 64 KB: larger plans are
        // the big kernels' -- a failed launch here would be an error where they would have served the plan
        const size_t smem_small = (2 * (NRX_MAX_FEATURES + 1) + 2) * sizeof(int) + (size_t)items * 16 +
                                  (size_t)((items + 3) & ~(int64_t)3) * 4 + (size_t)outs * 16 + 16;
        ok = ok && smem_small + NRX_SMALL_STATIC_LDS <= 64 * 1024;
        if (ok) {
            EmbedArgs a;
            int max_dim, max_bag;
            int rc = pack_features(feats, n_feats, a, max_dim, max_bag, "nrx_embed_fwd");
            if (rc != NRX_OK) return rc;
            for (int i = 0; i < NRX_MAX_FEATURES; ++i) a.feat_id[i] = (uint8_t)i;
            a.batch = batch; a.out = out; a.out_ld = out_ld; a.wide = nullptr; a.wide_ld = 0;
            a.fm_out = fm_fields > 0 ? fm_out : nullptr;
            a.fm_sums = fm_fields > 0 ? fm_sums : nullptr;
            a.sums_ld = sums_ld;
            a.g_fm = nullptr; a.feat = nullptr; a.feat_ld = 0;
            a.status = status;
            a.n = n_feats;
            a.lds_chunk = 0;
            hipLaunchKernelGGL(embed_fwd_small_kernel, dim3((unsigned)batch), dim3(NRX_BLOCK), smem_small, st, a);
            NRX_LAUNCH_CHECK("nrx_embed_fwd(small batch)");
            return NRX_OK;
        }
    }
    // ---- uniform fast path?  One launch when every feature is single-valued with the same width D = 4Q.  Otherwise, with no FM
    // epilogue and no wide split in play, and enough lookups to amortise the launches, the features that DO qualify are served by one uniform launch
    // per width, each writing its own columns of the same concat, and only the rest -- bags, dense values, odd widths -- goes
    // through the generic kernel, which walks its features one at a time (a 24-feature mix of widths 16 / 32 / 64: 186 us
    // generic, 95 us as three uniform launches).
    auto eligible = [&](const nrx_feature_t& s) {
        const int q = s.dim / 4;
        return s.kind == NRX_SPARSE && s.wide_col < 0 && s.index_bits == feats[0].index_bits && s.table != nullptr &&
               nrx_aligned16(s.table) && s.rows >= 1 && (s.dim & 3) == 0 && q >= 4 && q <= 64 && (q & (q - 1)) == 0;
    };
    // (first columns that are no multiple of 4 floats -- a dense value earlier in the sorted order -- and an unaligned `out` take the
    // ring kernel's dword-aligned store form: UniformArgs::unal; they used to drop to the generic kernel, 0.33 vs 0.6+ of peak)
    const bool out_ok = true;
    const bool out_al = (out == nullptr) || (nrx_aligned16(out) && (out_ld & 3) == 0);
    int n_fm = 0, n_dims = 0, n_el = 0;
    int dims_seen[8];
    bool el[NRX_MAX_FEATURES];
    for (int i = 0; i < n_feats; ++i) {
        n_fm += feats[i].fm_field != 0;
        el[i] = out_ok && eligible(feats[i]);
        if (!el[i]) continue;
        int j = 0;
        while (j < n_dims && dims_seen[j] != feats[i].dim) ++j;
        if (j == n_dims) {
            if (n_dims == 8) { el[i] = false; continue; }
            dims_seen[n_dims++] = feats[i].dim;
        }
        ++n_el;
    }
    const bool single = n_el == n_feats && n_dims == 1 && wide_out == nullptr && (n_fm == 0 || n_fm == n_feats) &&
                        ((out != nullptr) || (fm_out != nullptr && n_fm > 0));
    // split only when there is work to amortise the extra launches: at the reference's own feature set (5 features, widths 16 / 32)
    // one generic launch costs 6.5 us at B <= 16 384 against 9.5 us for two uniform ones, and 15.0 against 12.0 us at B = 65 536
    // (tools/probe_c1_split.py).  NRX_SPLIT_MIN_LOOKUPS overrides the threshold (tests force the split on small batches).
    int64_t min_lookups = 262144;
    if (const char* e = getenv("NRX_SPLIT_MIN_LOOKUPS")) min_lookups = atoll(e);
    const bool per_width = !single && n_el >= 2 && batch * (int64_t)n_el >= min_lookups && n_fm == 0 && fm_out == nullptr &&
                           wide_out == nullptr && out != nullptr;
    nrx_feature_t rest[NRX_MAX_FEATURES];
    uint8_t rest_id[NRX_MAX_FEATURES];
    for (int i = 0; i < NRX_MAX_FEATURES; ++i) rest_id[i] = (uint8_t)i;
    if (single || per_width) {
        for (int g = 0; g < n_dims; ++g) {
            const int D0 = dims_seen[g], Q0 = D0 / 4;
            UniformArgs ua;
            int n = 0;
            bool unal = !out_al;
            for (int i = 0; i < n_feats; ++i)
                if (el[i] && feats[i].dim == D0) unal |= (feats[i].out_col & 3) != 0;
            for (int i = 0; i < n_feats; ++i) {
                if (!el[i] || feats[i].dim != D0) continue;
                NRX_REQUIRE(feats[i].index != nullptr, "nrx_embed_fwd: feature %d: null index pointer", i);
                NRX_REQUIRE(feats[i].rows <= 0x7fffffffLL, "nrx_embed_fwd: feature %d: rows out of range", i);
                ua.table[n] = feats[i].table;
                ua.index[n] = feats[i].index;
                ua.rows[n] = feats[i].rows;
                ua.col4[n] = unal ? feats[i].out_col : feats[i].out_col / 4;
                ua.feat_id[n] = (uint8_t)i;
                ++n;
            }
            ua.batch = batch;
            ua.out = reinterpret_cast<float4*>(out);
            ua.ld4 = unal ? out_ld : out_ld / 4;
            ua.unal = unal ? 1 : 0;
            {   // non-temporal stores for the concat when every piece a lane group writes is one or more WHOLE 128-byte lines (rows of >= 32
                // floats, line-aligned columns and row stride): C5 plain concat 125.5 -> 124.3 us, 132.0 -> 128.5 with distinct output buffers.
                // Half-line pieces (D = 16) must merge in L2 first: with non-temporal stores the C2 launch takes 77.8 us instead of 58.5
                // (tools/ab_stnt.py; NRX_FWD_STNT=0|1 overrides)
                bool lines = D0 >= 32 && !unal && out != nullptr && (reinterpret_cast<uintptr_t>(out) & 127u) == 0 && (out_ld & 31) == 0;
                for (int i = 0; i < n_feats && lines; ++i)
                    if (el[i] && feats[i].dim == D0) lines = (feats[i].out_col & 31) == 0;
                const char* e = getenv("NRX_FWD_STNT");
                ua.stnt = e ? atoi(e) : (lines ? 1 : 0);
            }
            ua.fm_out = fm_out;
            ua.fm_sums = (fm_out != nullptr && n_fm > 0) ? fm_sums : nullptr;
            ua.sums_ld = sums_ld;
            ua.status = status;
            ua.n = n;
            ua.idx64 = feats[0].index_bits == 64;
            const bool fm = fm_out != nullptr && n_fm > 0;
            const bool store = out != nullptr;
            switch (Q0) {
                case 4: launch_uniform<2>(ua, batch, fm, store, st); break;
                case 8: launch_uniform<3>(ua, batch, fm, store, st); break;
                case 16: launch_uniform<4>(ua, batch, fm, store, st); break;
                case 32: launch_uniform<5>(ua, batch, fm, store, st); break;
                default: launch_uniform<6>(ua, batch, fm, store, st); break;
            }
        }
        NRX_LAUNCH_CHECK("nrx_embed_fwd(uniform)");
        if (n_el == n_feats) return NRX_OK;
        int n_rest = 0;                                  // the features the uniform launches did not cover: generic kernel below
        for (int i = 0; i < n_feats; ++i)
            if (!el[i]) { rest_id[n_rest] = (uint8_t)i; rest[n_rest++] = feats[i]; }
        feats = rest;
        n_feats = n_rest;
    }

    // ---- uniform features with the Wide&Deep column split (nrx_embed_wide.hip)
    if (fm_out == nullptr && nrx_launch_uniform_wide(feats, n_feats, batch, out, out_ld, wide_out, wide_ld, status, st)) {
        NRX_LAUNCH_CHECK("nrx_embed_fwd(uniform+wide)");
        return NRX_OK;
    }

    // ---- generic path
    EmbedArgs a;
    int max_dim, max_bag;
    int rc = pack_features(feats, n_feats, a, max_dim, max_bag, "nrx_embed_fwd");
    if (rc != NRX_OK) return rc;
    for (int i = 0; i < NRX_MAX_FEATURES; ++i) a.feat_id[i] = rest_id[i];
    a.batch = batch;
    a.out = out;
    a.out_ld = out_ld;
    a.wide = wide_out;
    a.wide_ld = wide_ld;
    a.fm_out = (n_fm > 0) ? fm_out : nullptr;
    a.fm_sums = (n_fm > 0) ? fm_sums : nullptr;
    a.sums_ld = sums_ld;
    a.g_fm = nullptr;
    a.feat = nullptr;
    a.feat_ld = 0;
    a.status = status;
    a.n = n_feats;
    NRX_REQUIRE(n_fm == 0 || fm_out != nullptr, "nrx_embed_fwd: fm_field set but fm_out is null");
    NRX_REQUIRE(n_fm == 0 || max_dim <= 256, "nrx_embed_fwd: fused FM epilogue supports dim <= 256");
    int qlog2;
    size_t smem;
    plan_generic(max_dim, max_bag, qlog2, a.lds_chunk, smem);
    const int tb = NRX_BLOCK >> qlog2;
    const unsigned grid = (unsigned)((batch + tb - 1) / tb);
    NRX_QSWITCH(qlog2, { hipLaunchKernelGGL((embed_fwd_generic<QL>), dim3(grid), dim3(NRX_BLOCK), smem, st, a); });
    NRX_LAUNCH_CHECK("nrx_embed_fwd(generic)");
    return NRX_OK;
}

// FM gradient folded into the embedding backward (see nrx_fm_grad_t): validated once for both backward forms
// ---- small deterministic dense backward (round 4): ONE launch, block per table, for the reference's own batch sizes.
// A block gathers its table's lookups as (row << 32 | feature << 12 | position) keys in LDS, sorts them there (bitonic: <= 4096 keys),
// finds the runs of equal rows and reduces each run IN SORTED ORDER straight into the table's dense gradient: no plan arrays, no workspace,
// no atomics, so the result is the same bit pattern run to run -- what the float-atomic scatter (nrx_embed_bwd), one launch too, cannot say.
// Rows with <= SD_LONG lookups are summed by one lane group in order; longer rows by a whole wavefront (lane groups stride the run, then a
// fixed xor tree) -- both orders depend on the sorted keys only.  Eligibility is the host's business (nrx_embed_bwd_small).
constexpr int SD_MAX = 4096;          // lookups per table (12 position bits in the key; 32 KB of keys)
constexpr int SD_LONG = 32;
constexpr int SD_THREADS = 1024;
struct SmallDetArgs {
    const void* ids[NRX_MAX_FEATURES];
    const float* weight[NRX_MAX_FEATURES];
    float* grad[NRX_MAX_FEATURES];          // the dense gradient of the feature's table
    int64_t rows[NRX_MAX_FEATURES];
    int32_t out_col[NRX_MAX_FEATURES];
    int32_t wide_col[NRX_MAX_FEATURES];
    // bag_len (0: single-valued) | kind << 16 | flags << 24 (1: FM field, 2: row 0 is data).  ONE 4-byte word per feature, no byte arrays
    // indexed by the feature: with them the compiler formed `kernarg + f` once and addressed the 8-byte arrays as
    // s_load_dwordx2 [kernarg + f], soffset 7 f -- and a scalar load ignores the two low bits of its base (wrong pointer for f % 4 != 0)
    int32_t meta[NRX_MAX_FEATURES];
    uint8_t seg_feat[NRX_MAX_FEATURES];     // feature indices grouped by table, ascending inside a table
    uint8_t seg_ptr[NRX_MAX_FEATURES + 1];  // block s owns seg_feat[seg_ptr[s] .. seg_ptr[s + 1])
    uint8_t seg_ql[NRX_MAX_FEATURES];       // log2 of the lanes per row of block s's table (dim / 4, rounded up to a power of two)
    int32_t seg_dim[NRX_MAX_FEATURES];      // its row width
    // SINK (nrx_embed_bwd_small_sparse): block s leaves its unique rows in slots seg_off[s] .. of (uniq, values) -- as many slots as the
    // table has lookups in the launch, the unused ones keyed -1 -- instead of storing into a dense gradient
    int32_t seg_off[NRX_MAX_FEATURES];
    uint8_t seg_tid[NRX_MAX_FEATURES];      // the table index the keys carry (key = table << 40 | row)
    int64_t* uniq;
    float* values;
    const float* g_out; int64_t out_ld;
    const float* g_wide; int64_t wide_ld;
    const float* g_fm; const float* fm_sums; int64_t sums_ld; const float* feat; int64_t feat_ld;
    int32_t batch, idx64, n_pow2, add_to;
};
static_assert(sizeof(SmallDetArgs) <= 3840, "SmallDetArgs must fit the kernel-argument segment");

struct SdFeat { const float* weight; const void* ids; int64_t rows; int32_t out_col, wide_col, L, kind, flags, den_base; };

// Bitonic sort of s_key[0 .. N) by the first NT threads of the block (N / NT = E keys each, in registers; blocked: thread t ends up with
// sorted positions t E .. t E + E - 1).  Partners inside the thread are exchanged in registers, inside the wavefront by shuffles; only the
// stages that cross wavefronts go through LDS with barriers (6 of 45 stages at N = 512, 10 of 78 at N = 4096 -- all of them through LDS were
// 10 us of a 22 us launch).  Every thread of the block calls this (the barriers); NT is a multiple of 64, so a wavefront is in or out whole.
template <int E, bool WAVE = false>
__device__ __forceinline__ void sd_sort(uint64_t* s_key, int N, int NT, int tid) {
    const bool active = tid < NT;         // WAVE: called by ONE wavefront (NT = 64, tid = lane): no stage crosses wavefronts, no barrier
    uint64_t k[E];
#pragma unroll
    for (int e = 0; e < E; ++e) k[e] = active ? s_key[tid * E + e] : ~0ull;
    for (int k2 = 2; k2 <= N; k2 <<= 1) {
        for (int j = k2 >> 1; j >= E; j >>= 1) {      // partner in another thread: thread tid ^ m, same slot e
            const int m = j / E;
            const bool lower = (tid & m) == 0;
            uint64_t o[E];
            if (m < 64) {
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = (uint64_t)__shfl_xor((unsigned long long)k[e], m, 64);
            } else {
                __syncthreads();
                if (active) {
#pragma unroll
                    for (int e = 0; e < E; ++e) s_key[e * NT + tid] = k[e];
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = active ? s_key[e * NT + (tid ^ m)] : ~0ull;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool up = ((tid * E + e) & k2) == 0;
                const bool take_min = lower == up;
                const uint64_t lo = k[e] < o[e] ? k[e] : o[e], hi = k[e] < o[e] ? o[e] : k[e];
                k[e] = take_min ? lo : hi;
            }
        }
        // both keys in this thread: the distances are compile-time constants (a run-time j here indexed k[] dynamically -- the array went to
        // scratch memory and the E > 1 sorts were no faster than the all-LDS form)
#pragma unroll
        for (int jj = E >> 1; jj > 0; jj >>= 1) {
            if (jj <= (k2 >> 1)) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & jj) == 0) {
                        const bool up = ((tid * E + e) & k2) == 0;
                        const uint64_t x = k[e], y = k[e | jj];
                        const bool sw = (x > y) == up;
                        k[e] = sw ? y : x;
                        k[e | jj] = sw ? x : y;
                    }
                }
            }
        }
    }
    if (!WAVE) __syncthreads();
    if (active) {
#pragma unroll
        for (int e = 0; e < E; ++e) s_key[tid * E + e] = k[e];
    }
}

// position in a list that the whole wavefront appends to: one LDS atomic per wavefront
__device__ __forceinline__ int sd_append(int* counter, bool pred, int lane) {
    const uint64_t m = __ballot(pred);
    int base = 0;
    if (lane == 0 && m != 0) base = atomicAdd(counter, (int)__popcll(m));
    base = __shfl(base, 0, 64);
    return base + (int)__popcll(m & ((1ull << lane) - 1ull));
}

template <bool FM, bool GEN, bool SINK = false>
__global__ __launch_bounds__(SD_THREADS) void embed_bwd_small_det_kernel(const SmallDetArgs args_in_kernarg) {
    const NRX_CONST SmallDetArgs* a = nrx_kernarg<SmallDetArgs>();
    extern __shared__ __attribute__((aligned(16))) unsigned char sd_smem[];
    __shared__ SdFeat s_f[NRX_MAX_FEATURES];
    __shared__ int s_wsum[SD_THREADS / 64];
    __shared__ int s_nuniq, s_nlong, s_nsingle, s_nmulti;
    __shared__ uint16_t s_long[SD_MAX / SD_LONG];      // rows with more than SD_LONG lookups (any order: a row's sum does not depend on who forms it)
    const int N = a->n_pow2, H = 2 * N;
    // dynamic LDS (sd_smem_bytes): keys | hash table (row, count), later the keys of the rows looked up more than once | run starts |
    // positions of the once-only lookups | positions of the others | masked-mean denominators
    uint64_t* s_key = reinterpret_cast<uint64_t*>(sd_smem);
    uint32_t* s_hrow = reinterpret_cast<uint32_t*>(s_key + N);
    int* s_hcnt = reinterpret_cast<int*>(s_hrow + H);
    uint64_t* s_mkey = reinterpret_cast<uint64_t*>(s_hrow);                  // (the hash table is dead by then)
    uint16_t* s_us = reinterpret_cast<uint16_t*>(s_hcnt + H);                // [N + 8]: start of every run of equal rows, then the end
    uint16_t* s_slot = s_us;                                                 // hash slot of every position, until the lists are made (H <= 8192: a slot fits, 0xffff = none)
    uint16_t* s_single = s_us + N + 8;
    uint16_t* s_mpos = s_single + N;
    float* s_den = reinterpret_cast<float*>(s_mpos + N);
    const int tid = threadIdx.x, s = blockIdx.x, NT = blockDim.x;        // NT: a power of two, 256 .. SD_THREADS, <= N
    const int lane = tid & 63, wv = tid >> 6;
    const int f0 = a->seg_ptr[s], f1 = a->seg_ptr[s + 1];
    const int B = a->batch;
    const int hshift = 32 - (31 - __clz(H));
    // ---- per-feature fields of this table; denominators of its masked-mean bags
    if (tid == 0) {                       // (uniform indices into the argument block: scalar loads)
        int db = 0;
        for (int fi = f0; fi < f1; ++fi) {
            const int f = a->seg_feat[fi];
            SdFeat t;
            const int m = a->meta[f];
            t.weight = a->weight[f]; t.ids = a->ids[f]; t.rows = a->rows[f]; t.out_col = a->out_col[f]; t.wide_col = a->wide_col[f];
            t.L = m & 0xffff; t.kind = (m >> 16) & 0xff; t.flags = (m >> 24) & 0xff; t.den_base = db;
            s_f[f] = t;
            db += t.kind == NRX_BAG_MASKED_MEAN ? B : 0;
        }
        s_nuniq = 0; s_nlong = 0; s_nsingle = 0; s_nmulti = 0;
    }
    for (int i = tid; i < H; i += NT) { s_hrow[i] = 0xffffffffu; s_hcnt[i] = 0; }
    __syncthreads();
    int base = 0;
    for (int fi = f0; fi < f1; ++fi) {
        const int f = a->seg_feat[fi];
        const SdFeat ft = s_f[f];
        const int L = ft.L, kind = ft.kind;
        const int len = B * (L > 0 ? L : 1);
        const bool keep0 = (ft.flags & 2) != 0;
        const uint64_t rows = (uint64_t)ft.rows;
        const NRX_GLOBAL float* w = nrx_gconst<float>(ft.weight);
        const bool has_w = ft.weight != nullptr && kind != NRX_BAG_MEAN && kind != NRX_SPARSE;
        for (int i = tid; i < len; i += NT) {
            const int64_t id = a->idx64 ? nrx_gconst<int64_t>(ft.ids)[i] : (int64_t)nrx_gconst<int32_t>(ft.ids)[i];
            bool ok = (uint64_t)id < rows && (id != 0 || keep0);
            if (ok && has_w) ok = w[i] != 0.f;                       // a masked position adds nothing
            s_key[base + i] = ok ? ((uint64_t)id << 32 | (uint32_t)f << 12 | (uint32_t)i) : ~0ull;
            // count the row's lookups: open addressing at load factor <= 1/2; which slot a row lands in depends on the race, the counts do not
            int sl = 0xffff;
            if (ok) {
                const uint32_t row = (uint32_t)id;
                uint32_t h = (row * 2654435761u) >> hshift;
                while (true) {
                    const uint32_t old = atomicCAS(&s_hrow[h], 0xffffffffu, row);
                    if (old == 0xffffffffu || old == row) break;
                    h = (h + 1) & (uint32_t)(H - 1);
                }
                atomicAdd(&s_hcnt[h], 1);
                sl = (int)h;
            }
            s_slot[base + i] = (uint16_t)sl;
        }
        if (kind == NRX_BAG_MASKED_MEAN) {
            float* den = s_den + ft.den_base;
            for (int b = tid; b < B; b += NT) {
                float d = 0.f;
                for (int l = 0; l < L; ++l) d += w[(int64_t)b * L + l];
                den[b] = d + 1e-8f;
            }
        }
        base += len;
    }
    for (int i = base + tid; i < N; i += NT) { s_key[i] = ~0ull; s_slot[i] = 0xffff; }
    if (SINK) {                           // every slot of the region starts as filler; the rows found below overwrite theirs (after barriers)
        NRX_GLOBAL int64_t* uq = nrx_gmut<int64_t>(a->uniq) + a->seg_off[s];
        for (int i = tid; i < base; i += NT) uq[i] = -1;
    }
    __syncthreads();
    // ---- how often is each row looked up?  A row looked up ONCE (nearly all of them at these batch sizes over the reference's tables) needs no
    // ordering: its gradient is that one lookup's contribution.  Only the others are sorted -- the full sort was half of the launch.
    const int CNT = N / NT;               // 1, 2 or 4 positions per thread
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < CNT) {                    // (uniform: every lane of the block takes part in the appends)
            const int sl = s_slot[c * NT + tid];
            const int n = sl != 0xffff ? s_hcnt[sl] : 0;
            const int ps = sd_append(&s_nsingle, n == 1, lane);
            if (n == 1) s_single[ps] = (uint16_t)(c * NT + tid);
            const int pm = sd_append(&s_nmulti, n > 1, lane);
            if (n > 1) s_mpos[pm] = (uint16_t)(c * NT + tid);
        }
    }
    __syncthreads();
    const int n_single = s_nsingle, n_multi = s_nmulti;
    // ---- per-lookup contribution (columns 4q .. 4q + 3) and the store of a finished row
    const int ql = a->seg_ql[s];
    const int Q = 1 << ql, q = tid & (Q - 1);
    const int64_t D = a->seg_dim[s];      // GEN: any width (the reference's wide features are 4 k + 1 wide), any alignment, element by element
    NRX_GLOBAL float* gtab = nrx_gmut<float>(a->grad[a->seg_feat[f0]]);
    auto contrib = [&](uint64_t key) -> float4 {
        const uint32_t p = (uint32_t)key;
        const int f = (int)(p >> 12), i = (int)(p & 4095u);
        const SdFeat ft = s_f[f];
        int64_t b = i;
        float sc = 1.0f;
        if (ft.L > 0) {
            b = i / ft.L;
            if (ft.kind == NRX_BAG_MASKED_MEAN) sc = nrx_gconst<float>(ft.weight)[i] / s_den[ft.den_base + b];
            else if (ft.kind == NRX_BAG_MEAN) sc = 1.0f / (float)ft.L;
            else if (ft.weight != nullptr) sc = nrx_gconst<float>(ft.weight)[i];
        }
        float4 t;
        if (GEN) {
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            const NRX_GLOBAL float* go = nrx_gconst<float>(a->g_out);
            const int shift = ft.wide_col >= 0 ? 1 : 0;          // a wide feature: column 0 comes from the wide gradient, the deep block is one narrower
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * q + j;
                if (k < D) {
                    if (shift && k == 0) e[j] = a->g_wide ? nrx_gconst<float>(a->g_wide)[b * a->wide_ld + ft.wide_col] : 0.f;
                    else e[j] = go ? go[b * a->out_ld + ft.out_col + k - shift] : 0.f;
                    if (FM) {
                        if (ft.flags & 1) {           // d fm / d field: column 0 -> 1, column k -> S_k - v_k (fm_fold4's arithmetic)
                            const float gf = nrx_gconst<float>(a->g_fm)[b];
                            if (k == 0) e[j] += gf;
                            else e[j] = __builtin_fmaf(gf, nrx_gconst<float>(a->fm_sums)[b * a->sums_ld + k] -
                                                               nrx_gconst<float>(a->feat)[b * a->feat_ld + ft.out_col + k], e[j]);
                        }
                    }
                }
            }
            t = make_float4(e[0], e[1], e[2], e[3]);
        } else {
            t = upstream_chunk<false>(a->g_out, a->out_ld, a->g_wide, a->wide_ld, ft.out_col, ft.wide_col, b, q);
        }
        if (FM && !GEN) {
            if (ft.flags & 1) {
                const float gf = nrx_gconst<float>(a->g_fm)[b];
                const float4 S = nrx_ldg4(a->fm_sums, (b * a->sums_ld) / 4 + q);
                const float4 v = nrx_ldg4(a->feat, (b * a->feat_ld + ft.out_col) / 4 + q);
                fm_fold4(t, gf, S, v, q);
            }
        }
        return make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc);      // (products then sums: nothing here contracts into an fma across the call)
    };
    const bool add_to = a->add_to != 0;
    const int64_t key_hi = SINK ? (int64_t)a->seg_tid[s] << 40 : 0;
    const int64_t slot0 = SINK ? a->seg_off[s] : 0;
    auto store = [&](int slot, uint32_t row, float4 acc) {
        if (SINK) {                         // slot: the row's place in the block's region (once-only rows first, then the sorted runs)
            if (q == 0) nrx_gmut<int64_t>(a->uniq)[slot0 + slot] = key_hi | (int64_t)row;
            NRX_GLOBAL float* d1 = nrx_gmut<float>(a->values) + (slot0 + slot) * D + 4 * q;
            if (GEN) {
                const float e[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * q + j < D) d1[j] = e[j];
            } else {
                nrx_f32x4 r;
                r.x = acc.x; r.y = acc.y; r.z = acc.z; r.w = acc.w;
                *reinterpret_cast<NRX_GLOBAL nrx_f32x4*>(d1) = r;
            }
            return;
        }
        if (GEN) {
            NRX_GLOBAL float* d1 = gtab + (int64_t)row * D + 4 * q;
            const float e[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < D) d1[j] = add_to ? d1[j] + e[j] : e[j];
            return;
        }
        NRX_GLOBAL nrx_f32x4* dst = reinterpret_cast<NRX_GLOBAL nrx_f32x4*>(gtab + (int64_t)row * D) + q;
        nrx_f32x4 r;
        r.x = acc.x; r.y = acc.y; r.z = acc.z; r.w = acc.w;
        if (add_to) {                       // a table fed by an earlier call too
            const nrx_f32x4 o = *dst;
            r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
        }
        *dst = r;
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- rows looked up once: RS per lane group in flight (a row is a chain position -> key -> upstream row -> store, ~1 us of latency;
    // one after the other that was 16 chains per group at B = 4096).  w0: the first wavefront that takes part.
    auto singles = [&](int w0) {
        constexpr int RS = 8;
        const int Gs = (NT - 64 * w0) >> ql, gs = (tid - 64 * w0) >> ql;
        for (int x0 = 0; x0 < n_single; x0 += Gs * RS) {
            uint64_t key[RS];
            float4 t[RS];
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const int x = x0 + r * Gs + gs;
                key[r] = x < n_single ? s_key[s_single[x]] : ~0ull;
            }
#pragma unroll
            for (int r = 0; r < RS; ++r) t[r] = key[r] != ~0ull ? contrib(key[r]) : zero4;
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                if (key[r] == ~0ull) continue;
                float4 acc = zero4;       // 0 + t, as the sorted form adds its first term
                acc.x += t[r].x; acc.y += t[r].y; acc.z += t[r].z; acc.w += t[r].w;
                store(x0 + r * Gs + gs, (uint32_t)(key[r] >> 32), acc);
            }
        }
    };
    const int gg = lane >> ql, G64 = 64 >> ql;
    // a run of more than SD_LONG lookups of one row, by a whole wavefront: its lane groups stride the run, then a fixed xor tree
    auto long_run = [&](const uint64_t* keys, int st, int en, int slot) {
        float4 acc = zero4;
        for (int e = st + gg; e < en; e += 2 * G64) {
            const float4 t0 = contrib(keys[e]);
            const float4 t1 = e + G64 < en ? contrib(keys[e + G64]) : zero4;
            acc.x += t0.x; acc.y += t0.y; acc.z += t0.z; acc.w += t0.w;
            if (e + G64 < en) { acc.x += t1.x; acc.y += t1.y; acc.z += t1.z; acc.w += t1.w; }
        }
        for (int o = Q; o < 64; o <<= 1) {
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
            acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        if (gg == 0) store(slot, (uint32_t)(keys[st] >> 32), acc);
    };
    // the runs of <= SD_LONG lookups: lane group g_ of G_ takes runs g_, g_ + G_, ..., R at a time (their first two terms in flight together),
    // each run added in order
    auto runs = [&](const uint64_t* keys, int U_, int G_, int g_) {
        constexpr int R = 4;
        for (int u0 = 0; u0 < U_; u0 += G_ * R) {
            int st[R], cnt[R];
            float4 t0[R], t1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int u = u0 + r * G_ + g_;
                st[r] = 0; cnt[r] = 0;
                if (u < U_) {
                    st[r] = s_us[u];
                    cnt[r] = s_us[u + 1] - st[r];
                    if (cnt[r] > SD_LONG) cnt[r] = 0;       // long_run's
                }
                t0[r] = cnt[r] > 0 ? contrib(keys[st[r]]) : zero4;
                t1[r] = cnt[r] > 1 ? contrib(keys[st[r] + 1]) : zero4;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (cnt[r] == 0) continue;
                float4 acc = zero4;
                acc.x += t0[r].x; acc.y += t0[r].y; acc.z += t0[r].z; acc.w += t0[r].w;
                if (cnt[r] > 1) { acc.x += t1[r].x; acc.y += t1[r].y; acc.z += t1[r].z; acc.w += t1[r].w; }
                const int en = st[r] + cnt[r];
                for (int e = st[r] + 2; e < en; e += 4) {
                    float4 t[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = e + j < en ? contrib(keys[e + j]) : zero4;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (e + j < en) { acc.x += t[j].x; acc.y += t[j].y; acc.z += t[j].z; acc.w += t[j].w; }
                }
                store(n_single + u0 + r * G_ + g_, (uint32_t)(keys[st[r]] >> 32), acc);
            }
        }
    };
    if (n_multi <= 64) {
        // the usual case at these batch sizes: a handful of lookups share their row with another.  Wavefront 0 sorts them with shuffles
        // alone -- no barrier -- and adds their runs while the others store the once-only rows; the two parts used to run one after the
        // other.  (Up to 256 keys, 4 per lane, through the general code was tried: slower -- one wavefront is then the tail of the launch)
        if (wv != 0) { singles(1); return; }
        if (n_multi == 0) return;
        uint64_t k = lane < n_multi ? s_key[s_mpos[lane]] : ~0ull;
        for (int k2 = 2; k2 <= 64; k2 <<= 1) {
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)k, j, 64);
                const bool take_min = ((lane & j) == 0) == ((lane & k2) == 0);
                const uint64_t lo = k < o ? k : o, hi = k < o ? o : k;
                k = take_min ? lo : hi;
            }
        }
        const uint32_t row = (uint32_t)(k >> 32);
        const uint32_t prow = (uint32_t)__shfl_up((int)row, 1, 64);
        const bool head = k != ~0ull && (lane == 0 || prow != row);
        const uint64_t hm = __ballot(head);
        const int n_rows = (int)__popcll(hm);
        s_mkey[lane] = k;                 // (one wavefront: its LDS accesses complete in program order)
        if (head) s_us[__popcll(hm & ((1ull << lane) - 1ull))] = (uint16_t)lane;
        if (lane == 0) s_us[n_rows] = (uint16_t)n_multi;
        for (int r = gg; r < n_rows; r += G64) {
            const int st = s_us[r], en = s_us[r + 1];
            if (en - st > SD_LONG) continue;
            float4 acc = zero4;
            for (int e = st; e < en; e += 4) {
                float4 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = e + j < en ? contrib(s_mkey[e + j]) : zero4;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (e + j < en) { acc.x += t[j].x; acc.y += t[j].y; acc.z += t[j].z; acc.w += t[j].w; }
            }
            store(n_single + r, (uint32_t)(s_mkey[st] >> 32), acc);
        }
        for (int r = 0; r < n_rows; ++r) {
            const int st = s_us[r], en = s_us[r + 1];
            if (en - st > SD_LONG) long_run(s_mkey, st, en, n_single + r);
        }
        return;
    }
    singles(0);
    // ---- the other rows: sort their lookups by (row, feature, position), find the runs of equal rows, add each run in that order
    int M = 64;
    while (M < n_multi) M <<= 1;
    for (int i = tid; i < M; i += NT) s_mkey[i] = i < n_multi ? s_key[s_mpos[i]] : ~0ull;
    __syncthreads();
    const int NTs = NT < M ? NT : M;
    if (M == NTs) sd_sort<1>(s_mkey, M, NTs, tid);
    else if (M == 2 * NTs) sd_sort<2>(s_mkey, M, NTs, tid);
    else sd_sort<4>(s_mkey, M, NTs, tid);
    __syncthreads();
    const int C = M / NTs;                // thread t < NTs owns sorted keys [t C, (t + 1) C)
    int heads = 0;
    if (tid < NTs) {
        for (int i = tid * C; i < (tid + 1) * C; ++i) {
            const uint64_t kx = s_mkey[i];
            heads += kx != ~0ull && (i == 0 || (uint32_t)(s_mkey[i - 1] >> 32) != (uint32_t)(kx >> 32));
        }
    }
    int incl = heads;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int rank = incl - heads;
    for (int j = 0; j < wv; ++j) rank += s_wsum[j];
    if (tid == NT - 1) s_nuniq = rank + heads;
    if (tid < NTs) {
        for (int i = tid * C; i < (tid + 1) * C; ++i) {
            const uint64_t kx = s_mkey[i];
            if (kx != ~0ull && (i == 0 || (uint32_t)(s_mkey[i - 1] >> 32) != (uint32_t)(kx >> 32))) s_us[rank++] = (uint16_t)i;
        }
    }
    __syncthreads();
    const int U = s_nuniq;
    if (tid == 0) s_us[U] = (uint16_t)n_multi;
    __syncthreads();
    for (int u = tid; u < U; u += NT)
        if ((int)s_us[u + 1] - (int)s_us[u] > SD_LONG) s_long[atomicAdd(&s_nlong, 1)] = (uint16_t)u;
    runs(s_mkey, U, NT >> ql, tid >> ql);
    __syncthreads();
    const int n_long = s_nlong;
    for (int li = wv; li < n_long; li += NT / 64) {
        const int u = s_long[li];
        long_run(s_mkey, s_us[u], s_us[u + 1], n_single + u);
    }
}

// dynamic LDS of embed_bwd_small_det_kernel for N keys and n_den denominators (the layout at the top of the kernel)
static size_t sd_smem_bytes(size_t N, size_t n_den) { return N * 8 + 2 * N * 8 + (N + 8) * 2 + N * 2 + N * 2 + n_den * 4 + 16; }

static int check_fm_grad(const nrx_fm_grad_t* fm, const nrx_feature_t* feats, int32_t n_feats, const char* who) {
    if (fm == nullptr || fm->g_fm == nullptr) return NRX_OK;
    NRX_REQUIRE(fm->fm_sums != nullptr && fm->feat != nullptr, "%s: FM gradient needs fm_sums and the forward concat", who);
    for (int i = 0; i < n_feats; ++i)
        if (feats[i].fm_field) {
            NRX_REQUIRE(feats[i].wide_col < 0, "%s: feature %d: an FM field cannot also be a wide feature", who, i);
            NRX_REQUIRE(feats[i].dim <= fm->sums_ld, "%s: feature %d: dim %d > sums_ld", who, i, feats[i].dim);
        }
    return NRX_OK;
}

extern "C" int nrx_embed_bwd(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                             const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                             const nrx_fm_grad_t* fm, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feats != nullptr && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES,
                "nrx_embed_bwd: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(batch >= 0, "nrx_embed_bwd: negative batch");
    const bool has_fm = fm != nullptr && fm->g_fm != nullptr;
    NRX_REQUIRE(g_out != nullptr || g_wide != nullptr || has_fm, "nrx_embed_bwd: no upstream gradient");
    if (batch == 0) return NRX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    EmbedArgs a;
    int max_dim, max_bag;
    int rc = pack_features(feats, n_feats, a, max_dim, max_bag, "nrx_embed_bwd");
    if (rc != NRX_OK) return rc;
    rc = check_fm_grad(fm, feats, n_feats, "nrx_embed_bwd");
    if (rc != NRX_OK) return rc;
    a.g_fm = has_fm ? fm->g_fm : nullptr;
    a.fm_sums = has_fm ? const_cast<float*>(fm->fm_sums) : nullptr;
    a.sums_ld = has_fm ? fm->sums_ld : 0;
    a.feat = has_fm ? fm->feat : nullptr;
    a.feat_ld = has_fm ? fm->feat_ld : 0;
    a.batch = batch;
    a.out = const_cast<float*>(g_out);
    a.out_ld = out_ld;
    a.wide = const_cast<float*>(g_wide);
    a.wide_ld = wide_ld;
    a.fm_out = nullptr;
    a.status = nullptr;
    a.n = n_feats;
    int qlog2;
    size_t smem;
    plan_generic(max_dim, max_bag, qlog2, a.lds_chunk, smem);
    const int tb = NRX_BLOCK >> qlog2;
    const unsigned grid = (unsigned)((batch + tb - 1) / tb);
    static const int gy_env = getenv("NRX_BWD_GY") ? atoi(getenv("NRX_BWD_GY")) : 0;      // measurement knob: 0 = choose, n = that many feature slices
    unsigned gy = 1;
    if (gy_env > 0) gy = (unsigned)gy_env;
    else while (gy < (unsigned)n_feats && grid * gy < 2048) gy *= 2;                         // until ~8 blocks per CU are in the launch
    if (gy > (unsigned)n_feats) gy = (unsigned)n_feats;
    static const int gz_env = getenv("NRX_BWD_GZ") ? atoi(getenv("NRX_BWD_GZ")) : 0;      // measurement knob: slices of a bag's positions
    unsigned gz = 1;
    if (gz_env > 0) gz = (unsigned)gz_env;
    else while (gz * 8 <= (unsigned)max_bag && grid * gy * gz < 2048) gz *= 2;               // slices of >= 4 positions
    if (max_bag > 0 && gz > (unsigned)max_bag) gz = (unsigned)max_bag;
    if (max_bag <= 0) gz = 1;
    NRX_QSWITCH(qlog2, { hipLaunchKernelGGL((embed_bwd_generic<QL>), dim3(grid, gy, gz), dim3(NRX_BLOCK), smem, st, a); });
    NRX_LAUNCH_CHECK("nrx_embed_bwd");
    return NRX_OK;
}

// The deterministic form of nrx_embed_bwd for small launches (embed_bwd_small_det_kernel): same arguments + accumulate,
// NRX_ERR_UNSUPPORTED (nothing enqueued, nrx_last_error untouched) when the launch is outside its shapes -- the caller then takes
// nrx_embed_bwd.  Shapes: every table fed by <= 4096 lookups of this call, dim <= 256, padded (not CSR) bags, rows < 2^32, ids of one width.
// Widths 4 * 2^k with 16-byte-aligned rows and upstream columns take the 16-byte form of the kernel, everything else (the reference's
// 16 + 1-column wide features, LR's dim-1 tables) the element-by-element form.
static int small_det_launch(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                            const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                            const nrx_fm_grad_t* fm, int32_t accumulate, void* stream,
                            const int32_t* table_of /* sink form: the table index of every feature */, int64_t* sink_uniq, float* sink_values,
                            int64_t sink_cap) {
    const bool sink = sink_uniq != nullptr;
    NRX_REQUIRE(feats != nullptr && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES,
                "nrx_embed_bwd_small: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(batch >= 0, "nrx_embed_bwd_small: negative batch");
    const bool has_fm = fm != nullptr && fm->g_fm != nullptr;
    NRX_REQUIRE(g_out != nullptr || g_wide != nullptr || has_fm, "nrx_embed_bwd_small: no upstream gradient");
    if (batch == 0) return NRX_OK;
    if (batch > SD_MAX) return NRX_ERR_UNSUPPORTED;
    int rc = check_fm_grad(fm, feats, n_feats, "nrx_embed_bwd_small");
    if (rc != NRX_OK) return rc;
    SmallDetArgs a = {};
    bool unal = (reinterpret_cast<uintptr_t>(g_out) & 15) != 0 || (out_ld & 3) != 0;
    bool any_fm = false;
    int bits = 0;
    // segments: features that share a gradient table, in order of first appearance
    int64_t seg_tab[NRX_MAX_FEATURES];
    int64_t seg_n[NRX_MAX_FEATURES], seg_den[NRX_MAX_FEATURES];
    int seg_dim[NRX_MAX_FEATURES], seg_of[NRX_MAX_FEATURES], n_seg = 0;
    for (int i = 0; i < n_feats; ++i) {
        const nrx_feature_t& f = feats[i];
        seg_of[i] = -1;
        if (f.kind == NRX_DENSE) continue;
        NRX_REQUIRE(f.kind >= NRX_SPARSE && f.kind <= NRX_BAG_SUM, "nrx_embed_bwd_small: feature %d: bad kind %d", i, f.kind);
        NRX_REQUIRE((sink || f.table != nullptr) && f.index != nullptr && f.rows >= 1 && f.dim >= 1, "nrx_embed_bwd_small: feature %d: null table / ids", i);
        NRX_REQUIRE(!sink || (table_of[i] >= 0 && table_of[i] < 256), "nrx_embed_bwd_small_sparse: feature %d: table index outside [0, 256)", i);
        const bool bag = f.kind >= NRX_BAG_MASKED_MEAN;
        NRX_REQUIRE(!bag || f.bag_len >= 1, "nrx_embed_bwd_small: feature %d: bag_len < 1", i);
        NRX_REQUIRE(f.kind != NRX_BAG_MASKED_MEAN || f.weight != nullptr, "nrx_embed_bwd_small: feature %d: masked mean needs weights", i);
        if ((f.flags & NRX_FEAT_BAG_CSR) || f.rows >= (1ll << 32) || (f.index_bits != 32 && f.index_bits != 64)) return NRX_ERR_UNSUPPORTED;
        if (bits == 0) bits = f.index_bits;
        if (bits != f.index_bits) return NRX_ERR_UNSUPPORTED;
        if (f.dim > 256) return NRX_ERR_UNSUPPORTED;
        bool pow2 = false;
        for (int k = 0; k <= 6; ++k) pow2 |= f.dim == (4 << k);
        if (!pow2 || (!sink && (reinterpret_cast<uintptr_t>(f.table) & 15) != 0)) unal = true;      // element-by-element form
        const int64_t len = batch * (bag ? f.bag_len : 1);
        if (len > SD_MAX || (bag && f.bag_len > SD_MAX)) return NRX_ERR_UNSUPPORTED;
        int sgi = -1;
        const int64_t tab_id = sink ? (int64_t)table_of[i] : (int64_t)reinterpret_cast<uintptr_t>(f.table);
        for (int k = 0; k < n_seg; ++k) if (seg_tab[k] == tab_id) sgi = k;
        if (sgi < 0) { sgi = n_seg++; seg_tab[sgi] = tab_id; seg_n[sgi] = 0; seg_den[sgi] = 0; seg_dim[sgi] = f.dim; }
        if (sink && f.dim != feats[0].dim) return NRX_ERR_UNSUPPORTED;          // one [cap, dim] values array
        if (seg_dim[sgi] != f.dim) return NRX_ERR_UNSUPPORTED;
        seg_n[sgi] += len;
        if (f.kind == NRX_BAG_MASKED_MEAN) seg_den[sgi] += batch;
        if (seg_n[sgi] > SD_MAX) return NRX_ERR_UNSUPPORTED;
        seg_of[i] = sgi;
        if (f.wide_col >= 0 || (f.out_col & 3) != 0) unal = true;
        const bool isfm = has_fm && f.fm_field != 0;
        any_fm |= isfm;
        a.ids[i] = f.index; a.weight[i] = f.weight; a.grad[i] = const_cast<float*>(f.table); a.rows[i] = f.rows;
        a.out_col[i] = f.out_col; a.wide_col[i] = f.wide_col; a.meta[i] = (bag ? f.bag_len : 0) | f.kind << 16 | ((isfm ? 1 : 0) | ((f.flags & NRX_FEAT_ROW0_IS_DATA) ? 2 : 0)) << 24;
    }
    if (n_seg == 0) return NRX_OK;
    if (sink) {
        if ((reinterpret_cast<uintptr_t>(sink_values) & 15) != 0) unal = true;
        int64_t off = 0;
        for (int sgi = 0; sgi < n_seg; ++sgi) { a.seg_off[sgi] = (int32_t)off; a.seg_tid[sgi] = (uint8_t)seg_tab[sgi]; off += seg_n[sgi]; }
        NRX_REQUIRE(sink_cap >= off, "nrx_embed_bwd_small_sparse: capacity %lld < the launch's %lld lookups", (long long)sink_cap, (long long)off);
        a.uniq = sink_uniq; a.values = sink_values;
    }
    if (any_fm && ((reinterpret_cast<uintptr_t>(fm->fm_sums) & 15) != 0 || (reinterpret_cast<uintptr_t>(fm->feat) & 15) != 0 ||
                   (fm->sums_ld & 3) != 0 || (fm->feat_ld & 3) != 0))
        unal = true;                           // the element-by-element form folds FM terms at any alignment
    int64_t max_n = 0, max_den = 0;
    int k = 0;
    for (int sgi = 0; sgi < n_seg; ++sgi) {
        a.seg_ptr[sgi] = (uint8_t)k;
        for (int i = 0; i < n_feats; ++i) if (seg_of[i] == sgi) a.seg_feat[k++] = (uint8_t)i;
        int ql = 0;
        while ((4 << ql) < seg_dim[sgi]) ++ql;
        a.seg_ql[sgi] = (uint8_t)ql;
        a.seg_dim[sgi] = seg_dim[sgi];
        if (seg_n[sgi] > max_n) max_n = seg_n[sgi];
        if (seg_den[sgi] > max_den) max_den = seg_den[sgi];
    }
    a.seg_ptr[n_seg] = (uint8_t)k;
    int N = 256;
    while (N < max_n) N <<= 1;
    a.n_pow2 = N;
    a.add_to = accumulate != 0;
    const int nt = N < SD_THREADS ? N : SD_THREADS;
    a.batch = (int32_t)batch;
    a.idx64 = bits == 64;
    a.g_out = g_out; a.out_ld = out_ld; a.g_wide = g_wide; a.wide_ld = wide_ld;
    a.g_fm = any_fm ? fm->g_fm : nullptr; a.fm_sums = any_fm ? fm->fm_sums : nullptr; a.sums_ld = any_fm ? fm->sums_ld : 0;
    a.feat = any_fm ? fm->feat : nullptr; a.feat_ld = any_fm ? fm->feat_ld : 0;
    const size_t smem = sd_smem_bytes((size_t)N, (size_t)max_den);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const bool lds_ok = [] {          // N = 4096 takes ~140 KB of the CU's 160 KB: above the 64 KB a kernel gets without asking
        const int most = (int)sd_smem_bytes(SD_MAX, SD_MAX);
        bool ok = true;
#define NRX_SD_ATTR(FM_, GEN_, SINK_) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(embed_bwd_small_det_kernel<FM_, GEN_, SINK_>), \
                                                                      hipFuncAttributeMaxDynamicSharedMemorySize, most) == hipSuccess
        NRX_SD_ATTR(true, false, false); NRX_SD_ATTR(true, true, false); NRX_SD_ATTR(false, true, false); NRX_SD_ATTR(false, false, false);
        NRX_SD_ATTR(true, false, true); NRX_SD_ATTR(true, true, true); NRX_SD_ATTR(false, true, true); NRX_SD_ATTR(false, false, true);
#undef NRX_SD_ATTR
        return ok;
    }();
    if (!lds_ok && smem > 60 * 1024) return NRX_ERR_UNSUPPORTED;
#define NRX_SD_GO(FM_, GEN_) do { if (sink) hipLaunchKernelGGL((embed_bwd_small_det_kernel<FM_, GEN_, true>), dim3(n_seg), dim3(nt), smem, st, a); \
                                  else hipLaunchKernelGGL((embed_bwd_small_det_kernel<FM_, GEN_, false>), dim3(n_seg), dim3(nt), smem, st, a); } while (0)
    if (any_fm && unal) NRX_SD_GO(true, true);
    else if (any_fm) NRX_SD_GO(true, false);
    else if (unal) NRX_SD_GO(false, true);
    else NRX_SD_GO(false, false);
#undef NRX_SD_GO
    NRX_LAUNCH_CHECK("nrx_embed_bwd_small");
    return NRX_OK;
}

extern "C" int nrx_embed_bwd_small(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                                   const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                   const nrx_fm_grad_t* fm, int32_t accumulate, void* stream) {
    NRX_TRACE();
    return small_det_launch(feats, n_feats, batch, g_out, out_ld, g_wide, wide_ld, fm, accumulate, stream, nullptr, nullptr, nullptr, 0);
}

// The row-sparse form of the same launch: instead of storing into dense gradient tables, block s leaves (key = table_of << 40 | row, summed row)
// pairs in its own region of (uniq_keys, values): as many slots as its table has lookups in the launch, regions in order of the tables' first
// appearance among the features, unused slots keyed -1 (nrx_sparse_adam_step skips them).  What FusedSparseAdam's sink takes at the reference's
// batch sizes in ONE launch (the planned form: ~12).  The pairs of a region are in no particular order; every row appears once.
extern "C" int nrx_embed_bwd_small_sparse(const nrx_feature_t* feats, const int32_t* table_of, int32_t n_feats, int64_t batch,
                                          const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                          const nrx_fm_grad_t* fm, int64_t* uniq_keys, float* values, int64_t capacity, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(table_of != nullptr && uniq_keys != nullptr && values != nullptr, "nrx_embed_bwd_small_sparse: null table_of / uniq_keys / values");
    return small_det_launch(feats, n_feats, batch, g_out, out_ld, g_wide, wide_ld, fm, 0, stream, table_of, uniq_keys, values, capacity);
}

extern "C" int64_t nrx_embed_bwd_sorted_workspace(int64_t n_lookups, int32_t dim) {
    if (n_lookups < 0 || dim < 1) return -1;
    const int64_t items = n_lookups / SORTED_LONG_T + 8, slots = sorted_long_slots_cap(n_lookups);
    return 32 + items * (int64_t)sizeof(LongItem) + slots * (int64_t)sizeof(LongMulti) + slots * (int64_t)dim * 4 + 64 +
           2 * (n_lookups * 4 + 64) + n_lookups / 8 + 128;       // + bag features: per-lookup scale, per-sample factor, weight bits
}

static int embed_bwd_sorted_impl(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                 const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                 const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                                 int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                                 uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                                 void* workspace, int64_t ws_bytes /* 0: the size nrx_embed_bwd_sorted_workspace promises */, void* stream,
                                 float* const* grad_tables = nullptr, int32_t n_tables = 0, int32_t add_to = 0,
                                 bool pairs = false /* the plan is nrx_sparse_plan_lds's: rows looked up twice are finished by embed_bwd_pairs_kernel */,
                                 void* aux_stream = nullptr /* pairs: the pair pass, the walk and the work lists run THERE, next to the placement pass */,
                                 const int32_t* pair_recs = nullptr, const int64_t* n_pairs = nullptr,
                                 bool place_only = false /* nrx_embed_bwd_scatter: the placement pass alone (dest is the caller's) */,
                                 float* const* multi_bases = nullptr, int multi_n = 0, int multi_shift = 0 /* place_only into several buffers */,
                                 bool skip_place = false /* nrx_embed_bwd_walk: the placed rows are in values already (written by the requesters) */) {
    NRX_REQUIRE(feats != nullptr && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES,
                "nrx_embed_bwd_sorted: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    NRX_REQUIRE(batch >= 0 && dim >= 1 && n_unique >= 0, "nrx_embed_bwd_sorted: bad argument");
    const bool has_fm = fm != nullptr && fm->g_fm != nullptr;
    NRX_REQUIRE(g_out != nullptr || g_wide != nullptr || has_fm, "nrx_embed_bwd_sorted: no upstream gradient");
    if (n_unique == 0 || batch == 0) return NRX_OK;
    const bool dense = grad_tables != nullptr;
    NRX_REQUIRE(order && seg_start && (values || dense), "nrx_embed_bwd_sorted: null buffer");
    NRX_REQUIRE(!dense || (uniq_keys != nullptr && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES),
                "nrx_embed_bwd_placed_dense: needs uniq_keys and 1 .. %d tables", NRX_MAX_FEATURES);
    {
        int rc = check_fm_grad(fm, feats, n_feats, "nrx_embed_bwd_sorted");
        if (rc != NRX_OK) return rc;
    }
    SortedBwdArgs a;
    a.g_fm = has_fm ? fm->g_fm : nullptr;
    a.fm_sums = has_fm ? fm->fm_sums : nullptr;
    a.sums_ld = has_fm ? fm->sums_ld : 0;
    a.feat = has_fm ? fm->feat : nullptr;
    a.feat_ld = has_fm ? fm->feat_ld : 0;
    a.long_ws = nullptr;
    a.long_items_cap = a.long_slots_cap = 0;
    a.scale = nullptr;
    a.bag_inv = nullptr;
    a.bag_bits = nullptr;
    a.walk = nullptr;
    a.n_walk_dev = nullptr;
    a.gs_all = nullptr;
    a.long_t = SORTED_LONG_T;
    a.dense = dense ? (add_to ? 2 : 1) : 0;
    for (int i = 0; i < NRX_MAX_FEATURES; ++i) a.f[i].index = nullptr;
    int64_t off = 0;
    for (int i = 0; i < n_feats; ++i) {
        const nrx_feature_t& s = feats[i];
        NRX_REQUIRE(s.kind == NRX_SPARSE || (s.kind >= NRX_BAG_MASKED_MEAN && s.kind <= NRX_BAG_SUM),
                    "nrx_embed_bwd_sorted: feature %d: kind %d has no table gradient", i, s.kind);
        NRX_REQUIRE(s.dim == dim, "nrx_embed_bwd_sorted: feature %d: dim %d != table dim %d", i, s.dim, dim);
        NRX_REQUIRE(s.kind != NRX_BAG_MASKED_MEAN || s.weight != nullptr, "nrx_embed_bwd_sorted: feature %d: masked mean needs weights", i);
        if (s.flags & NRX_FEAT_BAG_CSR) {
            nrx_set_error("nrx_embed_bwd_sorted: feature %d: CSR bags are not planned; expand with nrx_csr_to_padded", i);
            return NRX_ERR_UNSUPPORTED;
        }
        FeatDev& d = a.f[i];
        d.table = nullptr;
        d.weight = (s.kind == NRX_BAG_MEAN) ? nullptr : s.weight;
        // fast form: `rows` carries the 2^64 reciprocal of bag_len (sample = lookup / bag_len as one multiply-high, exact for
        // lookups < 2^32; (2^64 - 1) / d + 1 is floor(2^64 / d) + 1, or 2^64 / d itself for a power of two -- both exact)
        d.rows = s.bag_len > 1 ? (int64_t)(~0ull / (uint64_t)s.bag_len + 1) : 0;
        d.out_col = s.out_col;
        d.wide_col = s.wide_col;
        d.dim = (int16_t)s.dim;
        d.bag_len = (int16_t)s.bag_len;
        d.kind = (uint8_t)s.kind;
        d.idx64 = 1;
        d.fm = has_fm && s.fm_field != 0;
        d.flags = 0;
        a.off[i] = off;
        off += batch * (s.kind == NRX_SPARSE ? 1 : s.bag_len);
    }
    a.off[n_feats] = off;
    bool grads_al = true;
    if (dense) {                       // slot t of the descriptor array carries table t's gradient base (see SortedBwdArgs::dense)
        for (int t = 0; t < n_tables; ++t) {
            NRX_REQUIRE(grad_tables[t] != nullptr, "nrx_embed_bwd_placed_dense: table %d: null gradient pointer", t);
            a.f[t].index = grad_tables[t];
            grads_al = grads_al && nrx_aligned16(grad_tables[t]);
        }
        for (int i = 0; i < n_feats; ++i)
            NRX_REQUIRE(feats[i].table != nullptr, "nrx_embed_bwd_placed_dense: feature %d: feats[i].table must be its table's gradient", i);
    }
    a.uniform_len = a.off[1] - a.off[0];
    for (int i = 0; i < n_feats && a.uniform_len > 0; ++i)
        if (a.off[i + 1] - a.off[i] != a.uniform_len) a.uniform_len = 0;
    if (a.uniform_len < 2 || off >= 0xffffffffLL) a.uniform_len = 0;      // the reciprocal form needs a divisor >= 2 and lookups < 2^32
    a.uniform_magic = a.uniform_len > 0 ? ~0ull / (uint64_t)a.uniform_len + 1 : 0;
    a.col0 = feats[0].out_col;
    a.col_stride = n_feats > 1 ? feats[1].out_col - feats[0].out_col : dim;
    a.all_fm = a.f[0].fm;
    a.regular = a.uniform_len > 0 && n_feats > 4;
    for (int i = 0; i < n_feats && a.regular; ++i)
        a.regular = feats[i].kind == NRX_SPARSE && feats[i].wide_col < 0 && feats[i].out_col == a.col0 + i * a.col_stride &&
                    a.f[i].fm == a.f[0].fm;
    a.batch = batch;
    a.g_out = g_out;
    a.out_ld = out_ld;
    a.g_wide = g_wide;
    a.wide_ld = wide_ld;
    a.order = order;
    a.seg_start = seg_start;
    a.uniq_keys = uniq_keys;
    a.n_unique = n_unique;
    a.n_unique_dev = n_unique_dev;
    a.values = values;
    a.n = n_feats;
    a.dim = dim;
    int ql = ceil_log2((dim + 3) / 4);
    if (ql > 6) ql = 6;
    const int tb = NRX_BLOCK >> ql;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // fast form: plain single-valued features, D = 4 Q exactly, float4-addressable everywhere
    bool fast = dim == (4 << ql) && ql >= 2 && ql <= 4 && (dense ? grads_al : nrx_aligned16(values)) &&
                (!has_fm || (nrx_aligned16(fm->feat) && (fm->feat_ld & 3) == 0 && nrx_aligned16(fm->fm_sums) &&
                             (fm->sums_ld & 3) == 0 && fm->sums_ld >= dim));
    bool has_bag = false;
    bool unal = !(g_out == nullptr || (nrx_aligned16(g_out) && (out_ld & 3) == 0));      // wide routing / shifted columns / odd strides
    for (int i = 0; i < n_feats && fast; ++i) {
        unal |= feats[i].wide_col >= 0 || (feats[i].out_col & 3) != 0;
        if (feats[i].wide_col >= 0) fast = feats[i].kind == NRX_SPARSE;                 // (the split is defined for single-valued features)
        has_bag |= feats[i].kind != NRX_SPARSE;
    }
    if (unal) fast = fast && !has_fm && (reinterpret_cast<uintptr_t>(g_out) & 3u) == 0;
    // bag features ride the fast form through a per-lookup scale array that lives in the workspace; FM fields are
    // single-valued by construction (fm/model.py:48-59 stacks [B, D] tensors)
    if (has_bag) fast = fast && workspace != nullptr && !has_fm && off < 0xffffffffLL;
    // threshold sweep (T = 16 / 24 / 32 / 48 / 64, fwd+bwd us): C4 521 / 505 / 493 / 492 / 492, C4 Zipf 581 / 575 / 564 / 598 / 674,
    // C2 Zipf 501 / 545 / 588 / 678 / 758, C5 478 / 491 / 527 / 578 / 618: bag launches (a few rows, each looked up ~L times) take 32
    if (has_bag) a.long_t = 2 * SORTED_LONG_T;
    // NRX_FEAT_MANY_PER_ROW (the sharded step's pooled channel: a bag feature's lookups arrive at the owner as ONE single-valued pseudo-feature, ~16
    // lookups per row of the pooled table): the row statistics of a bag launch, hence its threshold (16 -> 32: the walk 110 -> 68 us, the work lists
    // 103 -> 5 us on C4's pseudo-batch).  A flag of the CALLER, not a guess from the table sizes: the threshold decides which rows leave the
    // in-order walk for the tree-summed work lists, so two call paths of one launch must agree on it to produce the same bits (a heuristic on
    // feats[].rows -- which the row-sparse and the dense entry points fill differently -- broke exactly that: tests/stress_embed_bwd.py).
    for (int i = 0; i < n_feats; ++i)
        if (!has_bag && (feats[i].flags & NRX_FEAT_MANY_PER_ROW)) a.long_t = 2 * SORTED_LONG_T;
    if (const char* e = getenv("NRX_LONG_T")) { const int v = atoi(e); if (v >= 2 && v <= 256) a.long_t = v; }      // measurement knob
    // placement mode: single-lookup rows are stored by the placement pass, the walk reduces the listed rows only
    const bool placed = fast && dest != nullptr;
    if (pairs && !(placed && !has_bag)) {
        nrx_set_error("nrx_embed_bwd_placed_pairs: the launch is outside the placement pass's shapes (dim 16 / 32 / 64, aligned operands, single-valued features)");
        return NRX_ERR_UNSUPPORTED;
    }
    if (skip_place && !(placed && !has_bag)) {
        nrx_set_error("nrx_embed_bwd_walk: the launch is outside the placement plan's shapes (dim 16 / 32 / 64, aligned operands, single-valued features)");
        return NRX_ERR_UNSUPPORTED;
    }
    if (place_only && multi_bases != nullptr && unal) {
        nrx_set_error("nrx_embed_bwd_scatter_multi: aligned upstream rows only");
        return NRX_ERR_UNSUPPORTED;
    }
    if (place_only && !(placed && !has_bag)) {
        nrx_set_error("nrx_embed_bwd_scatter: the launch is outside the placement pass's shapes (dim 16 / 32 / 64, aligned operands, single-valued features)");
        return NRX_ERR_UNSUPPORTED;
    }
    if (fast) {
        constexpr int R = 4;
        constexpr int RB = 2;                   // bag launches: 2 rows x 4 entries per pass (see the kernel)
        const bool bag_shape = has_bag && !unal && !has_fm;
        const bool wide_pass = bag_shape || placed;       // placement mode: every walked row has >= 2 entries -> 2 rows x 4 entries too
        int64_t n_rows = n_unique;                        // rows of the walk launch (upper bound when the count lives on the device)
        int n_place = 0;
        PlaceArgs pa;
        if (placed) {
            int64_t placeable = 0;
            for (int i = 0; i < n_feats; ++i) {
                if (!((place_feats >> i) & 1ull)) continue;
                NRX_REQUIRE(feats[i].kind == NRX_SPARSE, "nrx_embed_bwd_placed: feature %d is not single-valued: it cannot be placed", i);
                pa.off[n_place] = a.off[i];
                pa.ids[n_place] = feats[i].index;
                pa.grad[n_place] = const_cast<float*>(feats[i].table);
                NRX_REQUIRE(!dense || (feats[i].index != nullptr && (feats[i].index_bits == 32 || feats[i].index_bits == 64) &&
                                       feats[i].index_bits == feats[0].index_bits),
                            "nrx_embed_bwd_placed_dense: feature %d: the placement pass reads the ids (one width per launch)", i);
                pa.out_col[n_place] = feats[i].out_col;
                pa.wide_col[n_place] = feats[i].wide_col;
                pa.fm[n_place] = a.f[i].fm;
                placeable += batch;
                ++n_place;
            }
            // walked rows: >= 2 lookups each, or one lookup of a feature outside the mask, or a table's padding row
            const int64_t bound = placeable / 2 + (off - placeable) + n_feats + 1;
            if (bound < n_rows) n_rows = bound;
            a.walk = walk;
            a.n_walk_dev = n_walk;
            a.n_unique = n_rows;
        }
        const int64_t groups = (n_rows + (wide_pass ? RB : R) - 1) / (wide_pass ? RB : R);
        unsigned grid = (unsigned)((groups + tb - 1) / tb);
        {   // the walk's blocks stride over the row groups: when the row count is a device-side number (n_rows is only its bound) a few rounds of resident blocks are enough
            static const int cap = getenv("NRX_WALK_GRID") ? atoi(getenv("NRX_WALK_GRID")) : 4096;
            if ((n_unique_dev != nullptr || placed) && cap > 0 && grid > (unsigned)cap) grid = (unsigned)cap;
            if (pairs && grid > 512u) grid = 512u;      // (only the rows looked up 3+ times are walked: a few per thousand lookups on near-unique ids)
        }
        // (4 rows x 4 entries per lane group instead of 2 x 4: C5 446.9 -> 458.1 us, C3 163.2 -> 169.6 -- measured, not kept)
        if (workspace != nullptr) {        // long segments (hot rows) go through the wavefront-per-item path
            a.long_ws = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15);
            a.long_items_cap = off / SORTED_LONG_T + 8;
            a.long_slots_cap = sorted_long_slots_cap(off);
            // the four work-list counters are cleared by a kernel (nrx_zero_async), not hipMemsetAsync: inside a captured HIP graph
            // the 16-byte memset node did not take effect on replay (counters kept growing, the list was read past what was written)
            // (placement mode: the placement pass clears them -- one launch less)
            // (launches with bag features: cleared together with the weight-bit words below -- one launch, not two)
            if ((!(placed && n_place > 0) || skip_place) && !has_bag && nrx_zero_async(a.long_ws, 16, st) != NRX_OK) return NRX_ERR_LAUNCH;
            const bool side_zero = pairs && placed && n_place > 0 && aux_stream != nullptr && aux_stream != stream;      // (side-stream mode: the placement pass
            if (side_zero && nrx_zero_async(a.long_ws, 16, reinterpret_cast<hipStream_t>(aux_stream)) != NRX_OK) return NRX_ERR_LAUNCH;      //  runs elsewhere: the counters are cleared on the walk's stream)
        }
        const bool side_mode = pairs && aux_stream != nullptr && aux_stream != stream;
        auto launch_place = [&]() {
        if (placed && n_place > 0 && !skip_place) {
            pa.batch = batch;
            pa.g_out = g_out; pa.out_ld = out_ld; pa.g_wide = g_wide; pa.wide_ld = wide_ld;
            pa.g_fm = a.g_fm; pa.fm_sums = a.fm_sums; pa.sums_ld = a.sums_ld; pa.feat = a.feat; pa.feat_ld = a.feat_ld;
            pa.dest = dest;
            pa.values = values;
            pa.idx64 = feats[0].index_bits == 64;
            pa.add_to = add_to ? 1 : 0;
            { const char* e = getenv("NRX_PLACE_STNT"); pa.stnt = e ? atoi(e) : 0; }
            pa.long_ws = side_mode ? nullptr : a.long_ws;
            pa.n = n_place;
            { const char* e = getenv("NRX_PLACE_NT"); pa.nt = e ? atoi(e) : 1; }
            const int uvar = getenv("NRX_PLACE_U") ? atoi(getenv("NRX_PLACE_U")) : 4;       // fetches in flight per lane (4 | 8)
            // full-line form (embed_bwd_place_lines_kernel): 64-byte rows whose feature pairs (2j, 2j + 1) are one aligned 128-byte line of the
            // upstream rows (and of the forward concat); NRX_PLACE_LINES=0 keeps the one-feature-per-lane-group form
            bool lines = ql == 2 && !unal && pa.nt != 0 && pa.stnt == 0 && (g_out != nullptr || has_fm) && (reinterpret_cast<uintptr_t>(g_out) & 127) == 0 &&
                         (g_out == nullptr || (out_ld & 31) == 0) && n_place <= 64;
            if (lines && has_fm) lines = (reinterpret_cast<uintptr_t>(pa.feat) & 127) == 0 && (pa.feat_ld & 31) == 0;
            pa.fm_mask = 0;
            for (int i = 0; i < n_place; ++i) {
                if (pa.fm[i]) pa.fm_mask |= 1ull << i;
                if (pa.wide_col[i] >= 0) lines = false;
                if ((i & 1) == 0 ? (pa.out_col[i] & 31) != 0 : pa.out_col[i] != pa.out_col[i - 1] + 16) lines = false;
            }
            { const char* e = getenv("NRX_PLACE_LINES"); if (e && atoi(e) == 0) lines = false; }
            pa.multi_shift = 0;
            if (multi_bases != nullptr) {      // several destination buffers: the one-feature-per-lane-group form carries the base table (in grad[])
                for (int i = 0; i < multi_n; ++i) pa.grad[i] = multi_bases[i];
                pa.multi_shift = multi_shift;
                const unsigned pgrid = (unsigned)((batch + tb - 1) / tb);
                const size_t plds = (size_t)n_place * tb * 4;
                if (ql == 2) { if (has_fm) hipLaunchKernelGGL((embed_bwd_place_kernel<2, 4, true, false, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);
                               else hipLaunchKernelGGL((embed_bwd_place_kernel<2, 4, false, false, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa); }
                else if (ql == 3) { if (has_fm) hipLaunchKernelGGL((embed_bwd_place_kernel<3, 4, true, false, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);
                                    else hipLaunchKernelGGL((embed_bwd_place_kernel<3, 4, false, false, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa); }
                else { if (has_fm) hipLaunchKernelGGL((embed_bwd_place_kernel<4, 4, true, false, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);
                       else hipLaunchKernelGGL((embed_bwd_place_kernel<4, 4, false, false, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa); }
                return;
            }
            if (lines) {
                const unsigned lgrid = (unsigned)((batch + 31) / 32);
                const size_t llds = (size_t)n_place * PLACE_LINES_TBP * 4;
                if (g_out == nullptr && dense) hipLaunchKernelGGL((embed_bwd_place_lines_kernel<NRX_LINES_U, true, true, false>), dim3(lgrid), dim3(NRX_BLOCK), llds, st, pa);
                else if (g_out == nullptr) hipLaunchKernelGGL((embed_bwd_place_lines_kernel<NRX_LINES_U, true, false, false>), dim3(lgrid), dim3(NRX_BLOCK), llds, st, pa);
                else if (dense && has_fm) hipLaunchKernelGGL((embed_bwd_place_lines_kernel<NRX_LINES_U, true, true>), dim3(lgrid), dim3(NRX_BLOCK), llds, st, pa);
                else if (dense) hipLaunchKernelGGL((embed_bwd_place_lines_kernel<NRX_LINES_U, false, true>), dim3(lgrid), dim3(NRX_BLOCK), llds, st, pa);
                else if (has_fm) hipLaunchKernelGGL((embed_bwd_place_lines_kernel<NRX_LINES_U, true, false>), dim3(lgrid), dim3(NRX_BLOCK), llds, st, pa);
                else hipLaunchKernelGGL((embed_bwd_place_lines_kernel<NRX_LINES_U, false, false>), dim3(lgrid), dim3(NRX_BLOCK), llds, st, pa);
            } else {
            const unsigned pgrid = (unsigned)((batch + tb - 1) / tb);
            const size_t plds = (size_t)n_place * tb * 4;
            constexpr int U = 8;
#ifndef NRX_PLACE_UDEF
#define NRX_PLACE_UDEF 4                // fetches in flight per lane of the one-feature-per-lane-group form (build-time knob)
#endif
#define NRX_PL(QL_)                                                                                                        \
    {                                                                                                                      \
        if (dense && has_fm) hipLaunchKernelGGL((embed_bwd_place_kernel<QL_, NRX_PLACE_UDEF, true, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);   \
        else if (dense && unal) hipLaunchKernelGGL((embed_bwd_place_kernel<QL_, U, false, true, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa); \
        else if (dense) hipLaunchKernelGGL((embed_bwd_place_kernel<QL_, NRX_PLACE_UDEF, false, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);        \
        else if (has_fm && uvar == 4) hipLaunchKernelGGL((embed_bwd_place_kernel<QL_, NRX_PLACE_UDEF, true, false>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);   \
        else if (has_fm) hipLaunchKernelGGL((embed_bwd_place_kernel<QL_, U, true, false>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);   \
        else if (unal) hipLaunchKernelGGL((embed_bwd_place_kernel<QL_, U, false, true>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa); \
        else if (uvar == 4) hipLaunchKernelGGL((embed_bwd_place_kernel<QL_, NRX_PLACE_UDEF, false, false>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);        \
        else hipLaunchKernelGGL((embed_bwd_place_kernel<QL_, U, false, false>), dim3(pgrid), dim3(NRX_BLOCK), plds, st, pa);        \
    }
            if (ql == 2) NRX_PL(2) else if (ql == 3) NRX_PL(3) else NRX_PL(4)
#undef NRX_PL
            }
        }
        };
        // Pair plans with an auxiliary stream: the pair pass, the walk and the work lists are short chains of dependent round trips (18 + 10 + 9 us
        // on C2, nearly all of it latency); the placement pass is 60 us of streaming.  They touch disjoint rows: the small launches go to the
        // auxiliary stream FIRST (they get their wavefront slots before the placement pass fills every compute unit), the placement pass follows
        // on the caller's stream, which then waits for the auxiliary one.
        const bool side = pairs && aux_stream != nullptr && aux_stream != stream;
        hipStream_t sw = side ? reinterpret_cast<hipStream_t>(aux_stream) : st;
        static thread_local hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        if (side) {
            if (ev_fork == nullptr) {
                if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) {
                    nrx_set_error("nrx_embed_bwd_placed_pairs: hipEventCreate failed");
                    return NRX_ERR_LAUNCH;
                }
            }
            if (hipEventRecord(ev_fork, st) != hipSuccess || hipStreamWaitEvent(sw, ev_fork, 0) != hipSuccess) {
                nrx_set_error("nrx_embed_bwd_placed_pairs: fork onto the auxiliary stream failed");
                return NRX_ERR_LAUNCH;
            }
        }
        if (!side) launch_place();
        if (place_only) {
            NRX_LAUNCH_CHECK("nrx_embed_bwd_scatter");
            return NRX_OK;
        }
        unsigned pair_blocks = 0;
        if (pairs) {
            // (a plan with pair rows is a placement plan over single-valued features: the launch is `placed`, has no bags)
            const int64_t pgroups = (off / 2 + 1 + (NRX_BLOCK >> ql) - 1) / (NRX_BLOCK >> ql);      // (at most every second lookup starts a pair)
            pair_blocks = (unsigned)(pgroups < 512 ? pgroups : 512);
            a.scale = reinterpret_cast<const float*>(pair_recs);
            a.bag_inv = reinterpret_cast<const float*>(n_pairs);
            a.bag_bits = reinterpret_cast<const uint32_t*>(static_cast<intptr_t>(pair_blocks));
            if (unal) {        // column routing: the records in a launch of their own (the fused walk form exists for the aligned shapes)
                if (ql == 2) hipLaunchKernelGGL((embed_bwd_pairs_kernel<2, false, true, 0>), dim3(pair_blocks), dim3(NRX_BLOCK), 0, sw, a);
                else if (ql == 3) hipLaunchKernelGGL((embed_bwd_pairs_kernel<3, false, true, 0>), dim3(pair_blocks), dim3(NRX_BLOCK), 0, sw, a);
                else hipLaunchKernelGGL((embed_bwd_pairs_kernel<4, false, true, 0>), dim3(pair_blocks), dim3(NRX_BLOCK), 0, sw, a);
                pair_blocks = 0;
            }
        }
        if (has_bag) {
            char* end = reinterpret_cast<char*>(a.long_ws + 4) + a.long_items_cap * sizeof(LongItem) +
                        a.long_slots_cap * sizeof(LongMulti) + a.long_slots_cap * (size_t)dim * 4;
            float* scale = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(end) + 15) & ~(uintptr_t)15);
            float* inv = scale + ((off + 15) & ~(int64_t)15);
            uint32_t* bits = reinterpret_cast<uint32_t*>(inv + ((off + 15) & ~(int64_t)15));
            a.scale = scale;
            a.bag_inv = inv;
            a.bag_bits = bits;
            if (!(placed && n_place > 0) && workspace != nullptr) {
                if (nrx_zero2_async(a.long_ws, 16, bits, (size_t)(off / 32 + 2) * 4, st) != NRX_OK) return NRX_ERR_LAUNCH;
            } else if (nrx_zero_async(bits, (size_t)(off / 32 + 2) * 4, st) != NRX_OK) return NRX_ERR_LAUNCH;
            // pre-scaled upstream rows of the 0/1-weight bag features ([batch, dim] each) live behind the bit words -- when the
            // caller's workspace is known to hold them (nrx_embed_bwd_workspace_for) and the launch reads g_out aligned
            float* gs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(bits + off / 32 + 2) + 255) & ~(uintptr_t)255);
            // (every feature gets a block of the staging array: a single-valued feature's rows are copied, as "bags of one with
            // weight 1" -- then one array serves every lookup of the launch and a row is named by a 32-bit number: pass_staged)
            bool all_01 = true;                  // kinds whose factor is per SAMPLE when the weights are 0/1
            for (int i = 0; i < n_feats; ++i) all_01 &= feats[i].kind != NRX_BAG_SUM;
            const int64_t gs_end = (reinterpret_cast<char*>(gs) - reinterpret_cast<char*>(workspace)) + (int64_t)n_feats * batch * dim * 4;
            const char* gs_env = getenv("NRX_BAG_PRESCALE");             // "0": the per-lookup factor form (tests compare the two)
            const bool use_gs = bag_shape && g_out != nullptr && ws_bytes >= gs_end && !(gs_env && gs_env[0] == '0') &&
                                (int64_t)n_feats * batch < 0x7fffffffLL;
            if (use_gs && all_01) a.gs_all = gs;
            StageArgs sg;
            int n_sg = 0;
            for (int i = 0; i < n_feats; ++i) {
                if (batch == 0) continue;
                float* gsi = (use_gs && feats[i].kind != NRX_BAG_SUM) ? gs + (int64_t)i * batch * dim : nullptr;
                if (feats[i].kind == NRX_SPARSE) {
                    if (a.gs_all != nullptr) {
                        a.f[i].table = gsi;
                        sg.out_col[n_sg] = feats[i].out_col;
                        sg.block[n_sg] = i;
                        ++n_sg;
                    }
                    continue;
                }
                a.f[i].table = gsi;
                const int64_t groups16 = batch;                                          // 16 lanes per sample
                hipLaunchKernelGGL(bag_scale_kernel, dim3((unsigned)((groups16 * 16 + NRX_BLOCK - 1) / NRX_BLOCK)), dim3(NRX_BLOCK), 0, st,
                                   a.f[i].weight, (int)feats[i].kind, batch, (int)feats[i].bag_len, scale + a.off[i], inv + a.off[i], bits,
                                   a.off[i], a.long_ws + 3, feats[i].index, (int)(feats[i].index_bits == 64), g_out, out_ld,
                                   (int)feats[i].out_col, (int)dim, gsi);
            }
            if (n_sg > 0)
                hipLaunchKernelGGL(stage_rows_kernel, dim3((unsigned)((batch * (dim / 4) + NRX_BLOCK - 1) / NRX_BLOCK), (unsigned)n_sg),
                                   dim3(NRX_BLOCK), 0, st, sg, g_out, out_ld, batch, (int)(dim / 4), gs);
        }
        const bool reg = a.regular && (g_out != nullptr || has_fm) && !unal;        // arithmetic decode (embed_bwd_sorted_fast_kernel<.., DEC = 1>)
        const bool few = n_feats <= 4;                                  // scalar decode (DEC = 2); else the LDS table (DEC = 0)
#define NRX_SF(QL_)                                                                                                        \
    {                                                                                                                      \
        if (pair_blocks != 0 && has_fm && reg) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, true, false, false, 4, 1, true>), dim3(grid + pair_blocks), dim3(NRX_BLOCK), 0, sw, a); \
        else if (pair_blocks != 0 && has_fm) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, true, false, false, 4, 0, true>), dim3(grid + pair_blocks), dim3(NRX_BLOCK), 0, sw, a); \
        else if (pair_blocks != 0 && reg) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, false, false, 4, 1, true>), dim3(grid + pair_blocks), dim3(NRX_BLOCK), 0, sw, a); \
        else if (pair_blocks != 0) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, false, false, 4, 0, true>), dim3(grid + pair_blocks), dim3(NRX_BLOCK), 0, sw, a); \
        else if (placed && has_fm && reg) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, true, false, false, 4, 1>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); \
        else if (placed && !has_bag && !unal && !has_fm && reg) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, false, false, 4, 1>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); \
        else if (!placed && has_fm && reg) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, R, true, false, false, 1, 1>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); \
        else if (!placed && !has_fm && !unal && !has_bag && reg) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, R, false, false, false, 1, 1>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); \
        else if (placed && has_fm) { if (few) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, true, false, false, 4, 2>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, true, false, false, 4, 0>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else if (placed && unal) { if (few) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, true, true, 4, 2>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, true, true, 4, 0>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else if (placed && !has_bag) { if (few) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, false, false, 4, 2>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, false, false, 4, 0>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else if (has_fm) { if (few) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, R, true, false, false, 1, 2>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, R, true, false, false, 1, 0>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else if (unal) { if (few) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, R, false, true, true, 1, 2>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, R, false, true, true, 1, 0>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else if (has_bag) { if (few) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, true, false, 4, 2>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, RB, false, true, false, 4, 0>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else { if (few) hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, R, false, false, false, 1, 2>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((embed_bwd_sorted_fast_kernel<QL_, R, false, false, false, 1, 0>), dim3(grid), dim3(NRX_BLOCK), 0, sw, a); } \
    }
        if (ql == 2) NRX_SF(2) else if (ql == 3) NRX_SF(3) else NRX_SF(4)
#undef NRX_SF
        if (workspace != nullptr) {
            // (grid: 2048 blocks = 8192 wavefronts striding over the items; 1024 / 4096 measured, no difference on Zipf or uniform ids)
            const unsigned long_grid = 2048u;
#define NRX_SL(QL_)                                                                                                        \
    {                                                                                                                      \
        if (has_fm && reg) hipLaunchKernelGGL((sorted_long_kernel<QL_, true, false, false, 1>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a);      \
        else if (!has_fm && !unal && !has_bag && reg) hipLaunchKernelGGL((sorted_long_kernel<QL_, false, false, false, 1>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); \
        else if (has_fm) { if (few) hipLaunchKernelGGL((sorted_long_kernel<QL_, true, false, false, 2>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((sorted_long_kernel<QL_, true, false, false, 0>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else if (unal) { if (few) hipLaunchKernelGGL((sorted_long_kernel<QL_, false, true, true, 2>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((sorted_long_kernel<QL_, false, true, true, 0>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else if (has_bag) { if (few) hipLaunchKernelGGL((sorted_long_kernel<QL_, false, true, false, 2>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((sorted_long_kernel<QL_, false, true, false, 0>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); } \
        else { if (few) hipLaunchKernelGGL((sorted_long_kernel<QL_, false, false, false, 2>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); else hipLaunchKernelGGL((sorted_long_kernel<QL_, false, false, false, 0>), dim3(long_grid), dim3(NRX_BLOCK), 0, sw, a); } \
    }
            if (ql == 2) NRX_SL(2) else if (ql == 3) NRX_SL(3) else NRX_SL(4)
#undef NRX_SL
#ifdef NRX_COMBINE_SEPARATE
            if (ql == 2) hipLaunchKernelGGL((sorted_combine_kernel<2>), dim3(64), dim3(NRX_BLOCK), 0, sw, a);
            else if (ql == 3) hipLaunchKernelGGL((sorted_combine_kernel<3>), dim3(64), dim3(NRX_BLOCK), 0, sw, a);
            else hipLaunchKernelGGL((sorted_combine_kernel<4>), dim3(64), dim3(NRX_BLOCK), 0, sw, a);
#endif
        }
        if (side) {
            if (hipEventRecord(ev_join, sw) != hipSuccess) {
                nrx_set_error("nrx_embed_bwd_placed_pairs: join of the auxiliary stream failed");
                return NRX_ERR_LAUNCH;
            }
            launch_place();
            if (hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) {
                nrx_set_error("nrx_embed_bwd_placed_pairs: join of the auxiliary stream failed");
                return NRX_ERR_LAUNCH;
            }
        }
        NRX_LAUNCH_CHECK("nrx_embed_bwd_sorted(fast)");
        return NRX_OK;
    }
    const unsigned grid = (unsigned)((n_unique + tb - 1) / tb);
    NRX_QSWITCH(ql, { hipLaunchKernelGGL((embed_bwd_sorted_kernel<QL>), dim3(grid), dim3(NRX_BLOCK), 0, st, a); });
    NRX_LAUNCH_CHECK("nrx_embed_bwd_sorted");
    return NRX_OK;
}

extern "C" int nrx_embed_bwd_sorted(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                    const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                    const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                                    int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                                    void* workspace, void* stream) {
    NRX_TRACE();
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg_start, uniq_keys, n_unique,
                                 n_unique_dev, fm, values, 0, nullptr, nullptr, nullptr, workspace, 0, stream);
}

extern "C" int nrx_embed_bwd_placed_dense(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                          const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                          const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                                          int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm,
                                          float* const* grad_tables, int32_t n_tables, int32_t accumulate,
                                          uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                                          void* workspace, int64_t workspace_bytes, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(grad_tables != nullptr, "nrx_embed_bwd_placed_dense: null grad_tables");
    NRX_REQUIRE((dest != nullptr) == (walk != nullptr) && (dest != nullptr) == (n_walk != nullptr),
                "nrx_embed_bwd_placed_dense: dest, walk and n_walk come together (all null: no placement)");
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg_start, uniq_keys, n_unique,
                                 n_unique_dev, fm, nullptr, place_feats, dest, walk, n_walk, workspace, workspace_bytes, stream,
                                 grad_tables, n_tables, accumulate);
}

// ---- the whole deterministic dense-gradient backward of one launch group in ONE call: plan (nrx_sparse_plan_place) + reduction into the dense
// gradient tables (nrx_embed_bwd_placed_dense), every intermediate buffer carved out of one caller-provided workspace.  What the Python
// wrapper did with ten allocations and two library calls per group; at the reference's batch sizes (a few hundred to a few thousand samples)
// that host work was most of the step.
static inline size_t nrx_al256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int64_t nrx_embed_bwd_dense_sorted_workspace(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim, int32_t n_tables) {
    if (feats == nullptr || n_feats < 1 || n_feats > NRX_MAX_FEATURES || batch < 0 || dim < 1 || n_tables < 1) return -1;
    int64_t n = 0;
    for (int i = 0; i < n_feats; ++i) n += batch * (feats[i].kind == NRX_SPARSE ? 1 : feats[i].bag_len);
    const int64_t pw = nrx_sparse_plan_workspace(n), bw = nrx_embed_bwd_workspace_for(feats, n_feats, batch, dim);
    if (pw < 0 || bw < 0) return -1;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return (int64_t)(nrx_al256(nn * 8) * 2 + nrx_al256((nn + 1) * 8) + nrx_al256((size_t)(n_tables + 2) * 8) + nrx_al256(nn * 4) * 2 + 256 +
                     nrx_al256((size_t)pw) + nrx_al256((size_t)bw) + 512);
}

extern "C" int nrx_embed_bwd_dense_sorted(const nrx_feature_t* feats, const int32_t* table_of, int32_t n_feats, int32_t n_tables, int64_t batch,
                                          int32_t dim, const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                          const nrx_fm_grad_t* fm, float* const* grad_tables, int32_t accumulate, int32_t place,
                                          void* workspace, int64_t workspace_bytes, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feats && table_of && grad_tables && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES,
                "nrx_embed_bwd_dense_sorted: bad feature / table count");
    NRX_REQUIRE(workspace != nullptr && workspace_bytes >= nrx_embed_bwd_dense_sorted_workspace(feats, n_feats, batch, dim, n_tables),
                "nrx_embed_bwd_dense_sorted: workspace too small (nrx_embed_bwd_dense_sorted_workspace)");
    if (batch == 0) return NRX_OK;
    const void* ids[NRX_MAX_FEATURES];
    int64_t lens[NRX_MAX_FEATURES], rows[NRX_MAX_FEATURES];
    int64_t n = 0;
    uint64_t pmask = 0;
    int64_t n_sparse = 0;
    for (int i = 0; i < n_feats; ++i) {
        const nrx_feature_t& s = feats[i];
        NRX_REQUIRE(s.kind == NRX_SPARSE || (s.kind >= NRX_BAG_MASKED_MEAN && s.kind <= NRX_BAG_SUM), "nrx_embed_bwd_dense_sorted: feature %d: kind %d has no table gradient", i, s.kind);
        NRX_REQUIRE(s.index != nullptr && s.index_bits == feats[0].index_bits && (s.index_bits == 32 || s.index_bits == 64),
                    "nrx_embed_bwd_dense_sorted: feature %d: ids of one width (32 or 64 bits) are needed", i);
        NRX_REQUIRE(!(s.flags & NRX_FEAT_BAG_CSR), "nrx_embed_bwd_dense_sorted: feature %d: CSR bags are not planned; expand with nrx_csr_to_padded", i);
        NRX_REQUIRE(table_of[i] >= 0 && table_of[i] < n_tables && s.rows >= 1, "nrx_embed_bwd_dense_sorted: feature %d: bad table / rows", i);
        ids[i] = s.index;
        lens[i] = batch * (s.kind == NRX_SPARSE ? 1 : s.bag_len);
        rows[i] = s.rows;
        n += lens[i];
        if (s.kind == NRX_SPARSE) { pmask |= 1ull << i; n_sparse += batch; }
    }
    // placement pays when the single-valued features are a fair share of the lookups (ops.place_mask's rule)
    const bool placed = place != 0 && pmask != 0 && n_sparse * 4 >= n;
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const size_t nn = (size_t)(n > 0 ? n : 1);
    int64_t* order = (int64_t*)w;     w += nrx_al256(nn * 8);
    int64_t* uniq = (int64_t*)w;      w += nrx_al256(nn * 8);
    int64_t* seg = (int64_t*)w;       w += nrx_al256((nn + 1) * 8);
    int64_t* counts = (int64_t*)w;    w += nrx_al256((size_t)(n_tables + 2) * 8);
    int32_t* dest = (int32_t*)w;      w += nrx_al256(nn * 4);
    int32_t* walk = (int32_t*)w;      w += nrx_al256(nn * 4);
    int64_t* n_walk = (int64_t*)w;    w += 256;
    void* plan_ws = w;                w += nrx_al256((size_t)nrx_sparse_plan_workspace(n));
    void* bwd_ws = w;
    const int64_t bwd_bytes = nrx_embed_bwd_workspace_for(feats, n_feats, batch, dim);
    int rc;
    if (placed) rc = nrx_sparse_plan_place(ids, lens, table_of, rows, n_feats, feats[0].index_bits, n_tables, pmask, order, uniq, seg, counts,
                                           dest, walk, n_walk, plan_ws, stream);
    else rc = nrx_sparse_plan(ids, lens, table_of, rows, n_feats, feats[0].index_bits, n_tables, order, uniq, seg, counts, plan_ws, stream);
    if (rc != NRX_OK) return rc;
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg, uniq, n, counts, fm, nullptr,
                                 placed ? pmask : 0, placed ? dest : nullptr, placed ? walk : nullptr, placed ? n_walk : nullptr, bwd_ws,
                                 bwd_bytes, stream, grad_tables, n_tables, accumulate);
}

// nrx_embed_bwd_dense_sorted with the planner as an argument: planner == 1 takes the one-kernel planner (nrx_sparse_plan_lds; `state` = its
// control block) when the launch qualifies, and falls back to the sorted planner when it does not; stats (optional, may be mapped host memory):
// the plan's duplicate statistics in nrx_sparse_plan_lds's format, from either planner -- what the caller's choice for the NEXT batch needs.
extern "C" int64_t nrx_embed_bwd_dense_planned_workspace(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim, int32_t n_tables) {
    const int64_t base = nrx_embed_bwd_dense_sorted_workspace(feats, n_feats, batch, dim, n_tables);
    if (base < 0) return -1;
    int64_t n = 0;
    for (int i = 0; i < n_feats; ++i) n += batch * (feats[i].kind == NRX_SPARSE ? 1 : feats[i].bag_len);
    return base + (int64_t)nrx_al256((size_t)(n / 2 + 1) * 16) + (int64_t)nrx_al256((size_t)nrx_sparse_plan_lds_workspace(n)) + 512;
}

extern "C" int nrx_embed_bwd_dense_planned(const nrx_feature_t* feats, const int32_t* table_of, int32_t n_feats, int32_t n_tables, int64_t batch,
                                           int32_t dim, const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                           const nrx_fm_grad_t* fm, float* const* grad_tables, int32_t accumulate, int32_t planner,
                                           void* state, int64_t* stats, void* workspace, int64_t workspace_bytes, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feats && table_of && grad_tables && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES,
                "nrx_embed_bwd_dense_planned: bad feature / table count");
    NRX_REQUIRE(workspace != nullptr && workspace_bytes >= nrx_embed_bwd_dense_planned_workspace(feats, n_feats, batch, dim, n_tables),
                "nrx_embed_bwd_dense_planned: workspace too small (nrx_embed_bwd_dense_planned_workspace)");
    if (batch == 0) return NRX_OK;
    const void* ids[NRX_MAX_FEATURES];
    int64_t lens[NRX_MAX_FEATURES], rows[NRX_MAX_FEATURES];
    int64_t n = 0, n_sparse = 0;
    uint64_t pmask = 0;
    for (int i = 0; i < n_feats; ++i) {
        const nrx_feature_t& s = feats[i];
        NRX_REQUIRE(s.kind == NRX_SPARSE || (s.kind >= NRX_BAG_MASKED_MEAN && s.kind <= NRX_BAG_SUM), "nrx_embed_bwd_dense_planned: feature %d: kind %d has no table gradient", i, s.kind);
        NRX_REQUIRE(s.index != nullptr && s.index_bits == feats[0].index_bits && (s.index_bits == 32 || s.index_bits == 64),
                    "nrx_embed_bwd_dense_planned: feature %d: ids of one width (32 or 64 bits) are needed", i);
        NRX_REQUIRE(!(s.flags & NRX_FEAT_BAG_CSR), "nrx_embed_bwd_dense_planned: feature %d: CSR bags are not planned; expand with nrx_csr_to_padded", i);
        NRX_REQUIRE(table_of[i] >= 0 && table_of[i] < n_tables && s.rows >= 1, "nrx_embed_bwd_dense_planned: feature %d: bad table / rows", i);
        ids[i] = s.index;
        lens[i] = batch * (s.kind == NRX_SPARSE ? 1 : s.bag_len);
        rows[i] = s.rows;
        n += lens[i];
        if (s.kind == NRX_SPARSE) { pmask |= 1ull << i; n_sparse += batch; }
    }
    const bool placed = pmask != 0 && n_sparse * 4 >= n;
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const size_t nn = (size_t)(n > 0 ? n : 1);
    int64_t* order = (int64_t*)w;     w += nrx_al256(nn * 8);
    int64_t* uniq = (int64_t*)w;      w += nrx_al256(nn * 8);
    int64_t* seg = (int64_t*)w;       w += nrx_al256((nn + 1) * 8);
    int64_t* counts = (int64_t*)w;    w += nrx_al256((size_t)(n_tables + 2) * 8);
    int32_t* dest = (int32_t*)w;      w += nrx_al256(nn * 4);
    int32_t* walk = (int32_t*)w;      w += nrx_al256(nn * 4);
    int64_t* n_walk = (int64_t*)w;    w += 256;                     // [0] walk rows  [1] pair records
    int32_t* pairs = (int32_t*)w;     w += nrx_al256((size_t)(n / 2 + 1) * 16);
    void* plan_ws = w;
    {
        const size_t a_ = nrx_al256((size_t)nrx_sparse_plan_workspace(n)), b_ = nrx_al256((size_t)nrx_sparse_plan_lds_workspace(n));
        w += a_ > b_ ? a_ : b_;
    }
    void* bwd_ws = w;
    const int64_t bwd_bytes = nrx_embed_bwd_workspace_for(feats, n_feats, batch, dim);
    const int ql = ceil_log2((dim + 3) / 4);
    int rc;
    // the one-kernel planner: every feature single-valued, a width the placement pass takes, the launch inside the planner's shapes
    if (planner == 1 && state != nullptr && placed && n_sparse == n && dim == (4 << ql) && ql >= 2 && ql <= 4 &&
        nrx_sparse_plan_lds_ok(lens, table_of, rows, n_feats, n_tables)) {
        rc = nrx_sparse_plan_lds(ids, lens, table_of, rows, n_feats, feats[0].index_bits, n_tables, order, uniq, seg, counts, dest, walk, n_walk,
                                 pairs, n_walk + 1, stats, state, plan_ws, stream);
        if (rc != NRX_OK) return rc;
        rc = embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg, uniq, n, counts, fm, nullptr, pmask, dest,
                                   walk, n_walk, bwd_ws, bwd_bytes, stream, grad_tables, n_tables, accumulate, true, nullptr, pairs, n_walk + 1);
        if (rc != NRX_ERR_UNSUPPORTED) return rc;           // (outside the pair pass's shapes: nothing was enqueued -- the sorted planner's plan below)
    }
    if (placed) rc = nrx_sparse_plan_place(ids, lens, table_of, rows, n_feats, feats[0].index_bits, n_tables, pmask, order, uniq, seg, counts,
                                           dest, walk, n_walk, plan_ws, stream);
    else rc = nrx_sparse_plan(ids, lens, table_of, rows, n_feats, feats[0].index_bits, n_tables, order, uniq, seg, counts, plan_ws, stream);
    if (rc != NRX_OK) return rc;
    if (stats != nullptr && placed) {
        rc = nrx_sparse_plan_stats(counts, n_walk, n, stats, stream);
        if (rc != NRX_OK) return rc;
    }
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg, uniq, n, counts, fm, nullptr,
                                 placed ? pmask : 0, placed ? dest : nullptr, placed ? walk : nullptr, placed ? n_walk : nullptr, bwd_ws,
                                 bwd_bytes, stream, grad_tables, n_tables, accumulate);
}

// The row-sparse counterpart of nrx_embed_bwd_dense_planned: plan (either planner) + reduction in ONE call, the unique keys, their summed rows and the
// per-table counts left in caller-owned arrays (what the fused optimizer's sink holds), every intermediate in one workspace.
extern "C" int64_t nrx_embed_bwd_sparse_planned_workspace(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim, int32_t n_tables) {
    return nrx_embed_bwd_dense_planned_workspace(feats, n_feats, batch, dim, n_tables);
}

extern "C" int nrx_embed_bwd_sparse_planned(const nrx_feature_t* feats, const int32_t* table_of, int32_t n_feats, int32_t n_tables, int64_t batch,
                                            int32_t dim, const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                            const nrx_fm_grad_t* fm, int64_t* uniq_keys, float* values, int64_t* counts, int32_t planner,
                                            void* state, int64_t* stats, void* workspace, int64_t workspace_bytes, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(feats && table_of && uniq_keys && values && counts && n_feats >= 1 && n_feats <= NRX_MAX_FEATURES && n_tables >= 1 && n_tables <= NRX_MAX_FEATURES,
                "nrx_embed_bwd_sparse_planned: bad argument");
    NRX_REQUIRE(workspace != nullptr && workspace_bytes >= nrx_embed_bwd_sparse_planned_workspace(feats, n_feats, batch, dim, n_tables),
                "nrx_embed_bwd_sparse_planned: workspace too small (nrx_embed_bwd_sparse_planned_workspace)");
    if (batch == 0) return nrx_zero_async(counts, sizeof(int64_t) * (size_t)(n_tables + 2), reinterpret_cast<hipStream_t>(stream)) == NRX_OK ? NRX_OK : NRX_ERR_LAUNCH;
    const void* ids[NRX_MAX_FEATURES];
    int64_t lens[NRX_MAX_FEATURES], rows[NRX_MAX_FEATURES];
    int64_t n = 0, n_sparse = 0;
    uint64_t pmask = 0;
    for (int i = 0; i < n_feats; ++i) {
        const nrx_feature_t& s = feats[i];
        NRX_REQUIRE(s.kind == NRX_SPARSE || (s.kind >= NRX_BAG_MASKED_MEAN && s.kind <= NRX_BAG_SUM), "nrx_embed_bwd_sparse_planned: feature %d: kind %d has no table gradient", i, s.kind);
        NRX_REQUIRE(s.index != nullptr && s.index_bits == feats[0].index_bits && (s.index_bits == 32 || s.index_bits == 64),
                    "nrx_embed_bwd_sparse_planned: feature %d: ids of one width (32 or 64 bits) are needed", i);
        NRX_REQUIRE(!(s.flags & NRX_FEAT_BAG_CSR), "nrx_embed_bwd_sparse_planned: feature %d: CSR bags are not planned; expand with nrx_csr_to_padded", i);
        NRX_REQUIRE(table_of[i] >= 0 && table_of[i] < n_tables && s.rows >= 1, "nrx_embed_bwd_sparse_planned: feature %d: bad table / rows", i);
        ids[i] = s.index;
        lens[i] = batch * (s.kind == NRX_SPARSE ? 1 : s.bag_len);
        rows[i] = s.rows;
        n += lens[i];
        if (s.kind == NRX_SPARSE) { pmask |= 1ull << i; n_sparse += batch; }
    }
    const bool placed = pmask != 0 && n_sparse * 4 >= n;
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const size_t nn = (size_t)(n > 0 ? n : 1);
    int64_t* order = (int64_t*)w;     w += nrx_al256(nn * 8);
    int64_t* seg = (int64_t*)w;       w += nrx_al256((nn + 1) * 8);
    int32_t* dest = (int32_t*)w;      w += nrx_al256(nn * 4);
    int32_t* walk = (int32_t*)w;      w += nrx_al256(nn * 4);
    int64_t* n_walk = (int64_t*)w;    w += 256;                     // [0] walk rows  [1] pair records
    int32_t* pairs = (int32_t*)w;     w += nrx_al256((size_t)(n / 2 + 1) * 16);
    void* plan_ws = w;
    {
        const size_t a_ = nrx_al256((size_t)nrx_sparse_plan_workspace(n)), b_ = nrx_al256((size_t)nrx_sparse_plan_lds_workspace(n));
        w += a_ > b_ ? a_ : b_;
    }
    void* bwd_ws = w;
    const int64_t bwd_bytes = nrx_embed_bwd_workspace_for(feats, n_feats, batch, dim);
    const int ql = ceil_log2((dim + 3) / 4);
    int rc;
    if (planner == 1 && state != nullptr && placed && n_sparse == n && dim == (4 << ql) && ql >= 2 && ql <= 4 &&
        nrx_sparse_plan_lds_ok(lens, table_of, rows, n_feats, n_tables)) {
        rc = nrx_sparse_plan_lds(ids, lens, table_of, rows, n_feats, feats[0].index_bits, n_tables, order, uniq_keys, seg, counts, dest, walk, n_walk,
                                 pairs, n_walk + 1, stats, state, plan_ws, stream);
        if (rc != NRX_OK) return rc;
        rc = embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg, uniq_keys, n, counts, fm, values, pmask, dest,
                                   walk, n_walk, bwd_ws, bwd_bytes, stream, nullptr, 0, 0, true, nullptr, pairs, n_walk + 1);
        if (rc != NRX_ERR_UNSUPPORTED) return rc;           // (outside the pair pass's shapes: nothing was enqueued -- the sorted planner's plan below)
    }
    if (placed) rc = nrx_sparse_plan_place(ids, lens, table_of, rows, n_feats, feats[0].index_bits, n_tables, pmask, order, uniq_keys, seg, counts,
                                           dest, walk, n_walk, plan_ws, stream);
    else rc = nrx_sparse_plan(ids, lens, table_of, rows, n_feats, feats[0].index_bits, n_tables, order, uniq_keys, seg, counts, plan_ws, stream);
    if (rc != NRX_OK) return rc;
    if (stats != nullptr && placed) {
        rc = nrx_sparse_plan_stats(counts, n_walk, n, stats, stream);
        if (rc != NRX_OK) return rc;
    }
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg, uniq_keys, n, counts, fm, values,
                                 placed ? pmask : 0, placed ? dest : nullptr, placed ? walk : nullptr, placed ? n_walk : nullptr, bwd_ws,
                                 bwd_bytes, stream);
}

// workspace size that also holds the pre-scaled upstream rows of the launch's 0/1-weight bag features
extern "C" int64_t nrx_embed_bwd_workspace_for(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim) {
    if (feats == nullptr || n_feats < 0 || batch < 0 || dim < 1) return -1;
    int64_t n_lookups = 0, n_gs = 0;
    for (int i = 0; i < n_feats; ++i) {
        n_lookups += batch * (feats[i].kind == NRX_SPARSE ? 1 : feats[i].bag_len);
        n_gs |= feats[i].kind == NRX_BAG_MASKED_MEAN || feats[i].kind == NRX_BAG_MEAN;
    }
    return nrx_embed_bwd_sorted_workspace(n_lookups, dim) + 512 + (n_gs ? (int64_t)n_feats * batch * (int64_t)dim * 4 : 0);
}

extern "C" int nrx_embed_bwd_placed(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                    const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                    const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                                    int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                                    uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE((dest != nullptr) == (walk != nullptr) && (dest != nullptr) == (n_walk != nullptr),
                "nrx_embed_bwd_placed: dest, walk and n_walk come together (all null: no placement)");
    NRX_REQUIRE(workspace == nullptr || workspace_bytes == 0 || workspace_bytes >= nrx_embed_bwd_sorted_workspace(0, dim),
                "nrx_embed_bwd_placed: workspace_bytes too small");
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg_start, uniq_keys, n_unique,
                                 n_unique_dev, fm, values, place_feats, dest, walk, n_walk, workspace, workspace_bytes, stream);
}

// The placement pass alone, with the caller's destinations: values[dest[p]] = the upstream row of lookup p (flat, feature-major), FM term folded
// in.  The requester's half of the sharded backward (every lookup's gradient row goes to its slot of the send buffer: a permutation).
extern "C" int nrx_embed_bwd_scatter(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                     const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld, const nrx_fm_grad_t* fm,
                                     const int32_t* dest, float* values, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(dest != nullptr && values != nullptr, "nrx_embed_bwd_scatter: null dest / values");
    NRX_REQUIRE(n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "nrx_embed_bwd_scatter: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    for (int i = 0; i < n_feats; ++i)
        NRX_REQUIRE(feats[i].kind == NRX_SPARSE, "nrx_embed_bwd_scatter: feature %d is not single-valued", i);
    const uint64_t mask = n_feats == 64 ? ~0ull : ((1ull << n_feats) - 1ull);
    // (order / seg_start are the walk's inputs: never read by the placement pass; any non-null address passes the argument checks)
    const int64_t* dummy = reinterpret_cast<const int64_t*>(dest);
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, dummy, dummy, nullptr, /*n_unique=*/1, nullptr, fm,
                                 values, mask, dest, reinterpret_cast<const int32_t*>(dest), dummy, nullptr, 0, stream, nullptr, 0, 0, false, nullptr,
                                 nullptr, nullptr, /*place_only=*/true);
}

// nrx_embed_bwd_scatter into SEVERAL buffers: dest[p] = (base number << shift) | row -- the owners' gradient arenas as this process maps them.
extern "C" int nrx_embed_bwd_scatter_multi(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                           const float* g_out, int64_t out_ld, const nrx_fm_grad_t* fm, const int32_t* dest,
                                           float* const* bases, int32_t n_bases, int32_t shift, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(dest != nullptr && bases != nullptr && n_bases >= 1 && n_bases <= NRX_MAX_FEATURES && shift >= 1 && shift <= 30 &&
                    (int64_t)n_bases <= (1ll << (31 - shift)),
                "nrx_embed_bwd_scatter_multi: needs dest, 1..%d bases and a shift that leaves the base number below bit 31", NRX_MAX_FEATURES);
    NRX_REQUIRE(n_feats >= 1 && n_feats <= NRX_MAX_FEATURES, "nrx_embed_bwd_scatter_multi: n_feats must be in [1, %d]", NRX_MAX_FEATURES);
    for (int i = 0; i < n_feats; ++i)
        NRX_REQUIRE(feats[i].kind == NRX_SPARSE && feats[i].wide_col < 0, "nrx_embed_bwd_scatter_multi: feature %d is not a plain single-valued feature", i);
    for (int i = 0; i < n_bases; ++i)
        NRX_REQUIRE(bases[i] != nullptr && nrx_aligned16(bases[i]), "nrx_embed_bwd_scatter_multi: base %d: null / unaligned", i);
    const uint64_t mask = n_feats == 64 ? ~0ull : ((1ull << n_feats) - 1ull);
    const int64_t* dummy = reinterpret_cast<const int64_t*>(dest);
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, nullptr, 0, dummy, dummy, nullptr, /*n_unique=*/1, nullptr, fm,
                                 bases[0], mask, dest, reinterpret_cast<const int32_t*>(dest), dummy, nullptr, 0, stream, nullptr, 0, 0, false, nullptr,
                                 nullptr, nullptr, /*place_only=*/true, bases, n_bases, shift);
}

// The reduction WITHOUT its placement pass: the rows the plan places (dest >= 0) are in values[] already -- written there by the requesters
// (nrx_embed_bwd_scatter_multi with dest = the plan's unique index) -- and only the listed rows (and the pair records) are reduced from g_out.
extern "C" int nrx_embed_bwd_walk(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                  const float* g_out, int64_t out_ld, const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                                  int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                                  uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                                  const int32_t* pairs, const int64_t* n_pairs, void* workspace, int64_t workspace_bytes, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(dest != nullptr && walk != nullptr && n_walk != nullptr && uniq_keys != nullptr && values != nullptr && workspace != nullptr,
                "nrx_embed_bwd_walk: needs the placement plan (dest, walk, n_walk, uniq_keys), values and the work-list workspace");
    NRX_REQUIRE((pairs == nullptr) == (n_pairs == nullptr) && (pairs == nullptr || nrx_aligned16(pairs)), "nrx_embed_bwd_walk: pairs and n_pairs go together");
    for (int i = 0; i < n_feats; ++i)
        NRX_REQUIRE(feats != nullptr && feats[i].kind == NRX_SPARSE, "nrx_embed_bwd_walk: feature %d: single-valued features only", i);
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, nullptr, 0, order, seg_start, uniq_keys, n_unique, n_unique_dev, fm, values,
                                 place_feats, dest, walk, n_walk, workspace, workspace_bytes, stream, nullptr, 0, 0, pairs != nullptr, nullptr, pairs, n_pairs,
                                 false, nullptr, 0, 0, /*skip_place=*/true);
}

// nrx_embed_bwd_placed / nrx_embed_bwd_placed_dense for the placement plans of nrx_sparse_plan_lds (pair records for the rows looked up twice).
// values != NULL: row-sparse destination; grad_tables != NULL: the dense gradient tables.
extern "C" int nrx_embed_bwd_placed_pairs(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                          const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                          const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                                          int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                                          float* const* grad_tables, int32_t n_tables, int32_t accumulate,
                                          uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                                          const int32_t* pairs, const int64_t* n_pairs,
                                          void* workspace, int64_t workspace_bytes, void* aux_stream, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(dest != nullptr && walk != nullptr && n_walk != nullptr && uniq_keys != nullptr && pairs != nullptr && n_pairs != nullptr,
                "nrx_embed_bwd_placed_pairs: needs the whole plan (dest, walk, n_walk, pairs, n_pairs, uniq_keys)");
    NRX_REQUIRE(nrx_aligned16(pairs), "nrx_embed_bwd_placed_pairs: pairs must be 16-byte aligned");
    NRX_REQUIRE((values != nullptr) != (grad_tables != nullptr), "nrx_embed_bwd_placed_pairs: exactly one of values / grad_tables");
    NRX_REQUIRE(workspace != nullptr, "nrx_embed_bwd_placed_pairs: needs the work-list workspace (nrx_embed_bwd_workspace_for)");
    for (int i = 0; i < n_feats; ++i)
        NRX_REQUIRE(feats != nullptr && feats[i].kind == NRX_SPARSE && ((place_feats >> i) & 1ull), "nrx_embed_bwd_placed_pairs: feature %d: every feature must be single-valued and placeable", i);
    return embed_bwd_sorted_impl(feats, n_feats, batch, dim, g_out, out_ld, g_wide, wide_ld, order, seg_start, uniq_keys, n_unique,
                                 n_unique_dev, fm, values, place_feats, dest, walk, n_walk, workspace, workspace_bytes, stream,
                                 grad_tables, n_tables, accumulate, true, aux_stream, pairs, n_pairs);
}
