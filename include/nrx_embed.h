/*
 * nrx_embed.h -- C-ABI of libnrx_hip.so: the MI355X (gfx950) embedding / pooling /
 * feature-interaction hot path behind News_Recsys' BaseModel surface.
 *
 * The reference (ZhangHaoyang493/News_Recsys) has NO native / FFI layer: its "operator API"
 * for this path is the Python surface of BaseModel and its subclasses.  Each entry point below
 * therefore names the reference Python function(s) whose arithmetic it replaces (paths relative
 * to the reference root).  The Python host side (news_recsys_amd/_lib.py, ops.py) binds these
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes only; no torch / ATen types.
 *   - every data pointer is a CALLER-OWNED DEVICE pointer (tensor.data_ptr()); descriptor
 *     arrays (`nrx_feature_t*`, `float* const*` lists) are HOST arrays read during the call.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     calls only enqueue work: they never synchronise, allocate or free device memory.
 *   - return value: NRX_OK (0) or a negative NRX_ERR_* code; nrx_last_error() gives text.
 *   - tables are fp32 row-major [rows, dim]; ids are int64 or int32 (the reference casts
 *     with .long(), base_model.py:271); id 0 is the padding row (feature_extractor_base.py:162).
 *   - out-of-range ids never fault: the lookup is treated as row 0, and `status` (device
 *     int32[4], caller-zeroed: {count, feature, sample, id_lo}) records the offence so the
 *     host wrapper can raise IndexError like torch does on CPU.
 *   - re-entrant: no global mutable state except the thread-local last-error string.
 */
#ifndef NRX_EMBED_H
#define NRX_EMBED_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define NRX_API __attribute__((visibility("default")))
#else
#define NRX_API
#endif

#define NRX_ABI_VERSION 3 /* 3 (round 6): nrx_gather_inbox_place gained `out_rows`, nrx_dcn_v2_layer_bwd flags bit 2 (round 5, unversioned then); + the per-feature routing and the sharded backward entry points */
#define NRX_MAX_FEATURES 64   /* per launch; the host splits wider feature sets */
#define NRX_MAX_DCN_LAYERS 8

#define NRX_OK 0
#define NRX_ERR_BAD_ARG (-1)
#define NRX_ERR_LAUNCH (-2)
#define NRX_ERR_UNSUPPORTED (-3)

/* How one input feature becomes columns of the concat (base_model.py:284-308). */
enum nrx_feature_kind {
    NRX_SPARSE = 0,          /* ids [B]      -> table row                      (base_model.py:267-271) */
    NRX_DENSE = 1,           /* value [B]    -> one column, value.float()      (base_model.py:264-265) */
    NRX_BAG_MASKED_MEAN = 2, /* ids [B,L] + weight [B,L] -> sum(w*e)/(sum(w)+1e-8)  (base_model.py:278-282) */
    NRX_BAG_MEAN = 3,        /* ids [B,L], no mask -> mean over L incl. padding      (base_model.py:275-276) */
    NRX_BAG_SUM = 4          /* ids [B,L] (+ optional weight) -> sum(w*e); owner-side partial pooling */
};

typedef struct nrx_feature {
    const float* table;  /* device [rows, dim]; NULL for NRX_DENSE. In *_bwd calls: float* grad table */
    const void* index;   /* device ids [B] / [B, bag_len]; for NRX_DENSE the values (f32 or f64)   */
    const float* weight; /* device [B, bag_len] mask / weights, or NULL                           */
    int64_t rows;        /* table rows (ids must lie in [0, rows))                                 */
    int32_t dim;         /* embedding dim (1 for NRX_DENSE)                                        */
    int32_t bag_len;     /* L for bag kinds, 0 otherwise                                           */
    int32_t kind;        /* enum nrx_feature_kind                                                  */
    int32_t index_bits;  /* 32 or 64: width of ids (or of the dense value type)                    */
    int32_t out_col;     /* first column of this feature in `out`                                  */
    int32_t wide_col;    /* -1, or: column 0 goes to wide_out[:, wide_col] and columns 1..dim-1 to
                            out[:, out_col .. out_col+dim-2]          (widedeep/model.py:58-66)   */
    int32_t fm_field;    /* 1: field of the FM epilogue (col 0 = w, cols 1.. = v; fm/model.py:48-59) */
    int32_t flags;       /* NRX_FEAT_* bits                                                        */
} nrx_feature_t;

/* nrx_feature.flags: row 0 of `table` is an ordinary row (the "table" is a buffer of routed rows
 * addressed by slot, sharding step 4): the backward must not treat index 0 as the padding row. */
#define NRX_FEAT_ROW0_IS_DATA 1
/* nrx_feature.flags, bag kinds only: the bag arrives in CSR form instead of the reference's padded [B, L] + mask
 * (data_reader.py:96-109 builds the padded form per sample on the host): `index` = the concatenated ids of all bags
 * (int32 / int64 per index_bits), `weight` is reinterpreted as `const int64_t* offsets` (device, [B + 1], ascending),
 * sample b's bag = index[offsets[b] .. offsets[b + 1]) truncated to its first bag_len entries -- DataReader's
 * truncation.  Every entry counts with weight 1, i.e. the result is bit-identical to the padded form with DataReader's
 * mask (NRX_BAG_MASKED_MEAN: sum / (n + 1e-8); NRX_BAG_MEAN: the bag_len - n missing positions read row 0 like padding
 * ids do; NRX_BAG_SUM: sum) while a sample costs 4-8 B per REAL entry instead of 12 B x bag_len.  Accepted by
 * nrx_embed_fwd / nrx_embed_fwd_train / nrx_embed_bwd; the sorted backward wants the padded form (nrx_csr_to_padded).
 * (NRX_FEAT_ROW0_IS_DATA is likewise honoured by nrx_embed_bwd only: for the sorted backward row 0 of every table is the
 * padding row and gets a zero gradient.) */
#define NRX_FEAT_BAG_CSR 2
/* nrx_feature.flags, single-valued features of the sorted backward (nrx_embed_bwd_sorted / _placed / _walk): the caller knows that the feature's
 * lookups outnumber the rows they name many times over (a bag feature flattened into one pseudo-feature: the sharded step's pooled channel) -- rows
 * leave the in-order walk for the wavefront-per-chunk work lists from 32 lookups on, as in a launch with bag features, instead of 16. */
#define NRX_FEAT_MANY_PER_ROW 4

/* ---- library ---------------------------------------------------------------------------- */
NRX_API int nrx_abi_version(void);
NRX_API const char* nrx_last_error(void);
/* device facts used by the bench harness:
 * {CUs, wavefront size, core clock kHz, global memory bytes, memory clock kHz, memory bus bits} */
NRX_API int nrx_device_info(int device, int64_t info[6]);
/* measurement utility of the bench harness: dst[0..bytes) = src[0..bytes), 16 bytes per lane, non-temporal -- the plain-copy rate
 * of the box the roofline fractions are read against.  16-byte aligned device pointers, bytes % 16 == 0. */
NRX_API int nrx_stream_copy(void* dst, const void* src, int64_t bytes, void* stream);

/* ---- fused multi-table gather (+pool) -> concat, optional wide split and FM epilogue --------
 * Replaces BaseModel.get_embeddings_from_batch (base_model.py:284-308) = per-feature
 * get_feature_embedding (262-271) + array_feature_pooling (273-282) + torch.cat, and, when
 * requested, WideDeep.get_inp_embedding's column routing (widedeep/model.py:53-69) and
 * FM.get_inp_embedding + the pre-sigmoid part of FMModel.forward (fm/model.py:18-26,48-59):
 *   fm_out[b] = sum_f w_f + 0.5 * sum_k[(sum_f v_fk)^2 - sum_f v_fk^2]      (no bias, no sigmoid)
 * out      : [B, out_ld] fp32 (may be NULL only if every feature routes nowhere else -- i.e. never)
 * wide_out : [B, wide_ld] or NULL;  fm_out : [B] or NULL;  status: device int32[4] or NULL.      */
NRX_API int nrx_embed_fwd(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                  float* out, int64_t out_ld, float* wide_out, int64_t wide_ld,
                  float* fm_out, int32_t* status, void* stream);

/* Largest batch nrx_embed_fwd / nrx_embed_fwd_train serve with the one-block-per-sample kernel (default 2048, first read from the
 * environment variable NRX_SMALL_BATCH; 0 = never).  Process-wide; returns the previous limit; a negative argument only queries.  Both
 * kernel families produce the same bits -- the knob exists so that tests and A/B runs can pick the family on any batch size.  */
NRX_API int64_t nrx_set_small_batch_max(int64_t max_batch);

/* Training form of nrx_embed_fwd: additionally leaves the FM epilogue's per-sample field sums in
 * fm_sums [B, sums_ld] (column k >= 1: sum_f v_fk, column 0: sum_f w_f; sums_ld >= the FM field dim) -- 4 B x dim per
 * sample, what the backward needs to form d fm / d field without a pass of its own (nrx_fm_grad_t below).
 * fm_sums == NULL: identical to nrx_embed_fwd.                                                     */
NRX_API int nrx_embed_fwd_train(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                        float* out, int64_t out_ld, float* wide_out, int64_t wide_ld,
                        float* fm_out, float* fm_sums, int64_t sums_ld, int32_t* status, void* stream);

/* Gradient of the FM epilogue, folded into the embedding backward (autograd of FMModel.forward, fm/model.py:18-26):
 * for a lookup of an FM field (fm_field = 1) the upstream row becomes
 *   g_out[b, col + k] + g_fm[b] * (k == 0 ? 1 : fm_sums[b, k] - feat[b, col + k])
 * where feat is the forward concat (the field's value v_fk).  All device pointers; NULL struct or NULL g_fm: no FM term. */
typedef struct nrx_fm_grad {
    const float* g_fm;     /* [B]            dL / d fm_out                                   */
    const float* fm_sums;  /* [B, sums_ld]   from nrx_embed_fwd_train                        */
    int64_t sums_ld;
    const float* feat;     /* [B, feat_ld]   the forward's `out`                             */
    int64_t feat_ld;
} nrx_fm_grad_t;

/* Backward of nrx_embed_fwd's gather/pool/concat/wide-split: scatter-adds into DENSE grad tables
 * (feats[i].table is the float* grad table [rows, dim], pre-zeroed by the caller), i.e. what
 * autograd produces for nn.Embedding(sparse=False) (base_model.py:164); the padding row 0
 * receives no gradient.  g_out [B, out_ld] (or NULL), g_wide [B, wide_ld] (or NULL).
 * The FM epilogue's gradient rides along through `fm` (nrx_fm_grad_t; NULL = none): no separate FM backward pass
 * and no [B, sum D] temporary.  NRX_DENSE features are inputs and receive no gradient.          */
NRX_API int nrx_embed_bwd(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                  const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                  const nrx_fm_grad_t* fm, void* stream);

/* nrx_embed_bwd for the reference's own batch sizes (a few hundred to a few thousand samples), DETERMINISTIC: one launch, one block
 * per gradient table; a block sorts its table's lookups in LDS and adds every row's contributions in sorted order -- no atomics, the
 * same bits run to run, where nrx_embed_bwd's float atomics differ in the last place.  Arguments as nrx_embed_bwd, plus `accumulate`:
 * 0 = every gradient row the launch touches is STORED (pre-zeroed tables, this call their only writer -- no read of the old row),
 * 1 = added to what is there (a table fed by an earlier call too).  Returns NRX_ERR_UNSUPPORTED with nothing enqueued when the launch is outside its shapes (the caller
 * then takes nrx_embed_bwd or the planned reduction): every table fed by <= 4096 lookups of this call, dim <= 256, padded (not CSR)
 * bags, rows < 2^32, ids of one width.  (Widths 4 * 2^k on 16-byte-aligned rows and columns move 16 bytes per lane; any other width or
 * alignment -- the reference's 16 + 1-column wide features, LR's dim-1 tables -- goes element by element.)  Replaces, for those shapes, autograd's index_add_ of base_model.py:164's embeddings on the CPU
 * (deterministic there too).                                                                                                      */
NRX_API int nrx_embed_bwd_small(const nrx_feature_t* feats, int32_t n_feats, int64_t batch,
                  const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                  const nrx_fm_grad_t* fm, int32_t accumulate, void* stream);

/* The ROW-SPARSE form of nrx_embed_bwd_small (what the fused row-sparse optimizer's sink takes at the reference's batch sizes -- one launch where
 * nrx_sparse_plan + nrx_embed_bwd_sorted are ~12): the block of a table leaves (key = table_of << 40 | row, the row's summed gradient) pairs in
 * its own region of uniq_keys [capacity] / values [capacity, dim] -- as many slots as the table has lookups in the launch, regions in order of
 * the tables' first appearance among the features, unused slots keyed -1 (nrx_sparse_adam_step skips negative keys).  Every row appears once; a
 * region's pairs are in no particular order (the sums are order-fixed: same bits run to run).  feats[i].table is not read; table_of (HOST,
 * n_feats, values in [0, 256)); one dim for all features; capacity >= the launch's lookups.  NRX_ERR_UNSUPPORTED as nrx_embed_bwd_small. */
NRX_API int nrx_embed_bwd_small_sparse(const nrx_feature_t* feats, const int32_t* table_of, int32_t n_feats, int64_t batch,
                  const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                  const nrx_fm_grad_t* fm, int64_t* uniq_keys, float* values, int64_t capacity, void* stream);

/* Deterministic row-sparse backward for ONE table (alternative to nrx_embed_bwd's dense atomics).
 * The caller has sorted the table's lookups by row id (stable): `order[e]` is the flat lookup index of
 * the e-th sorted entry, flat lookups being feature-major over the n_feats features that read the
 * table (feats[i]: kind, index (unused), weight, bag_len, out_col, wide_col as in the forward; a
 * sparse feature contributes B lookups, a bag feature B*bag_len), and unique row u owns the sorted
 * entries seg_start[u] .. seg_start[u+1).  For every unique row the kernel sums, in sorted order,
 *   g_out[b, cols of the feature] * (1 | w/(sum w + 1e-8) | 1/L | w)
 * into values[u, :dim] -- no atomics, bit-reproducible.  uniq_keys (optional, device int64
 * [n_unique]): the sort key of each unique entry; entries whose low 40 bits (the row id) are 0 -- the
 * padding row, which never trains -- get zeros.  order / seg_start: device int64; values [n_unique, dim].
 * n_unique_dev (optional, device int64[1]): the actual number of unique entries when the host does not
 * know it yet (nrx_sparse_plan's counts[0]); n_unique is then an upper bound that sizes the launch.
 * fm (optional): the FM epilogue's gradient, as in nrx_embed_bwd.
 * workspace (optional, device bytes >= nrx_embed_bwd_sorted_workspace(n_lookups, dim)): rows looked up more than 16 times
 * in the batch (hot ids of a skewed click log, tiny tables) are then reduced by whole wavefronts in chunks of 256 lookups
 * instead of serially by one lane group -- same fixed summation order from run to run.  NULL: every row is walked by its
 * lane group (fine for near-unique ids). */
NRX_API int64_t nrx_embed_bwd_sorted_workspace(int64_t n_lookups, int32_t dim);
NRX_API int nrx_embed_bwd_sorted(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                         const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                         const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                         int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                         void* workspace, void* stream);

/* Whole planning step of the row-sparse backward in one call (replaces nrx_make_table_keys + an external
 * 64-bit sort + unique + scan): for the flat, feature-major lookup list of n_feats features (ids[f]: lens[f]
 * ids of width index_bits reading table table_of[f] with rows[f] rows; HOST arrays of device pointers /
 * sizes) it leaves on the device
 *   order     [n]      flat lookup index of the e-th entry in (table, row)-sorted, stable order
 *   uniq_keys [<= n]   (table << 40) | row of every distinct (table, row), ascending
 *   seg_start [<= n+1] unique entry u owns sorted entries seg_start[u] .. seg_start[u+1)
 *   counts    [n_tables + 2]   counts[0] = n_unique; table t owns unique entries counts[1+t] .. counts[2+t)
 * -- exactly the inputs of nrx_embed_bwd_sorted.  Ids < 0 or >= rows[f] fall on row 0 (the padding row, which
 * never trains).  Keys are sorted on table_bits + row_bits bits only (32-bit keys when they fit).  All
 * outputs device int64; workspace >= nrx_sparse_plan_workspace(n) device bytes; n < 2^32 - 1.          */
NRX_API int64_t nrx_sparse_plan_workspace(int64_t n_lookups);
NRX_API int nrx_sparse_plan(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                    int32_t n_feats, int32_t index_bits, int32_t n_tables, int64_t* order, int64_t* uniq_keys,
                    int64_t* seg_start, int64_t* counts, void* workspace, void* stream);

/* nrx_sparse_plan plus the PLACEMENT of the rows that need no reduction (autograd of nn.Embedding,
 * src/model/BaseModel/base_model.py:262-308, gives a row looked up once exactly that lookup's upstream row).
 * A unique (table, row) that is looked up exactly once in the launch, is not the padding row, and whose lookup belongs to a
 * feature f with bit f of place_feats set (pass the single-valued features: a bag lookup is scaled, not copied), is "placed":
 *   dest   [n]        int32: dest[p] = the unique index u of lookup p's row if that row is placed, else -1 (written for the
 *                            lookups of the features in place_feats only; the other words are left untouched)
 *   walk   [<= n]     int32: the unique indices that are NOT placed (several lookups, a non-placeable lookup, row 0), ascending
 *   n_walk [1]        int64: how many
 * nrx_embed_bwd_placed streams the upstream rows sample-major and stores the placed ones (coalesced reads, one 4-D-byte
 * write per row; the walk of every sorted entry re-fetches upstream rows at random: 128-byte fabric requests for 64-byte rows)
 * and reduces only the `walk` rows in sorted order.  Everything else as nrx_sparse_plan (same workspace size). */
NRX_API int nrx_sparse_plan_place(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                          int32_t n_feats, int32_t index_bits, int32_t n_tables, uint64_t place_feats, int64_t* order,
                          int64_t* uniq_keys, int64_t* seg_start, int64_t* counts, int32_t* dest, int32_t* walk,
                          int64_t* n_walk, void* workspace, void* stream);

/* ONE-KERNEL planner for near-unique id batches over mid-size tables (the C2 shape: 26 tables x 1 M rows, 65 536 ids each): the role of
 * nrx_sparse_plan_place by another method -- row bitmaps in LDS instead of a sort (csrc/nrx_plan_lds.hip).  Every feature must be
 * single-valued and placeable (lens all equal), every feature of a table must state the same rows; nrx_sparse_plan_lds_ok says whether a launch
 * qualifies (<= 4096 row ranges of 131 072 rows, scanning redundancy <= 16) -- else NRX_ERR_UNSUPPORTED and nothing is enqueued.  Outputs:
 *   uniq_keys / counts   as nrx_sparse_plan (complete)
 *   dest   [n] int32     dest[p] >= 0: lookup p's row is looked up once: its unique index (as nrx_sparse_plan_place); else -1
 *   pairs  [n / 2 + 1][4] int32 (16-byte aligned), n_pairs [1] int64: one record {unique index u, first lookup, second lookup, 0} per row looked
 *                        up exactly TWICE (not the padding row), ascending by u: the sum of two rows needs no sorted walk
 *   walk / n_walk        the unique rows looked up three times or more, and the padding rows: ascending
 *   order / seg_start    defined for the walk rows ONLY: row u owns order[seg_start[u] .. seg_start[u + 1]), its lookups ascending
 *   stats  [4] int64     (optional) unique rows, walk rows, lookups of the walk rows, n -- what a caller needs to choose the planner for the
 *                        NEXT batch: a batch with many rows looked up 3+ times is planned correctly but slowly here (one block sorts a range's list)
 * state: nrx_sparse_plan_lds_state_bytes() device bytes, ZERO before the first call, then owned by the planner (one stream at a time);
 * workspace: nrx_sparse_plan_lds_workspace(n) bytes of scratch.  Consumed by nrx_embed_bwd_placed_pairs.  Replaces, for these shapes, autograd's
 * index bookkeeping of nn.Embedding (src/model/BaseModel/base_model.py:262-308). */
NRX_API int64_t nrx_sparse_plan_lds_state_bytes(void);
NRX_API int64_t nrx_sparse_plan_lds_workspace(int64_t n_lookups);
NRX_API int nrx_sparse_plan_lds_ok(const int64_t* lens, const int32_t* table_of, const int64_t* rows, int32_t n_feats, int32_t n_tables);
NRX_API int nrx_sparse_plan_lds(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                        int32_t n_feats, int32_t index_bits, int32_t n_tables, int64_t* order, int64_t* uniq_keys,
                        int64_t* seg_start, int64_t* counts, int32_t* dest, int32_t* walk, int64_t* n_walk, int32_t* pairs,
                        int64_t* n_pairs, int64_t* stats, void* state, void* workspace, void* stream);
/* nrx_sparse_plan / nrx_sparse_plan_place with options and statistics.  dest / walk / n_walk all null: the plan without placement (place_feats
 * unused).  flags: NRX_PLAN_SPLIT_PADDING -- the lookups of the padding row (id 0, out-of-range ids) are set aside before the sort (they are
 * written straight to the front of their table's run, in lookup order) and every pass sorts the live lookups only: same plan, bit for bit;
 * pays when a large share of the lookups is padding (padded histories, reference: src/dataset/DataReader/data_reader.py pads every multi-valued
 * feature to max_len with id 0): 142 -> 114 us for 3.4 M lookups half of which are padding, ~17 us LOST on a launch without padding -- which
 * is why it is the caller's choice.  stats (null, or 5 int64 -- mapped host memory is fine): {unique rows, walk rows or -1, -1, n, lookups of
 * the padding rows} of THIS plan, written on the stream: what the caller decides the next batch's flag by (ops.PadPolicy).  Workspace as
 * nrx_sparse_plan.
 * NRX_PLAN_PAIRS (placement form, every feature in place_feats): rows looked up exactly TWICE leave the walk list as records
 * pairs[k] = {unique index, first lookup, second lookup, 0} (ascending; n_pairs[0] of them; capacity n / 2 + 1 records of 16 bytes) -- the plan
 * nrx_sparse_plan_lds makes, from the sort: dest, walk, pairs as oracle/ref_np.py sparse_plan_pairs defines them (order / seg_start stay those
 * of nrx_sparse_plan: complete) -- for nrx_embed_bwd_placed_pairs.  pairs / n_pairs are ignored without the flag (may be null).  (Measured on the
 * bench shapes it does not pay behind the sort -- the longer emit kernel and the pair pass cost what the shorter walk saves: C5 394.6 -> 398.0 us per
 * forward + backward -- so the Python layer leaves it off; it is the one-kernel planner's plan form made available from the sorted one.) */
#define NRX_PLAN_SPLIT_PADDING 1u
#define NRX_PLAN_PAIRS 2u
/* NRX_PLAN_PAYLOAD (alone; place_feats = 0 or no placement outputs): `pairs` is read as `const uint32_t* payload` [n] and order[] lists
 * payload[p] where it would list the lookup p -- same unique rows, segments and counts.  For callers whose reduction fetches a lookup's upstream
 * row somewhere else than at its own position (the sharded step's pooled channel: an owner's inbox entry names the row of its (source, sample)
 * in a block of sample gradients).  Served by the table-segmented sort (<= 64 tables); else NRX_ERR_UNSUPPORTED. */
#define NRX_PLAN_PAYLOAD 4u
NRX_API int nrx_sparse_plan_ex(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* rows,
                               int32_t n_feats, int32_t index_bits, int32_t n_tables, uint64_t place_feats, uint32_t flags, int64_t* order,
                               int64_t* uniq_keys, int64_t* seg_start, int64_t* counts, int32_t* dest, int32_t* walk, int64_t* n_walk,
                               int32_t* pairs, int64_t* n_pairs, int64_t* stats, void* workspace, void* stream);

/* stats of a nrx_sparse_plan_place plan in the same format ({unique rows, walk rows, -1 = not counted, n}): a caller that picks the planner of the
 * next batch from the previous batch's statistics has them from either planner.  stats may be mapped (pinned) host memory. */
NRX_API int nrx_sparse_plan_stats(const int64_t* counts, const int64_t* n_walk, int64_t n_lookups, int64_t* stats, void* stream);

/* The reduction behind nrx_sparse_plan_lds: nrx_embed_bwd_placed (values != NULL: row-sparse destination) or nrx_embed_bwd_placed_dense
 * (grad_tables != NULL; n_tables, accumulate as there) -- exactly one of the two -- plus a pass over the pair records: a lane group fetches the
 * two upstream rows of a record and stores 0 + first + second, the sum the sorted walk forms for a two-entry segment, bit for bit.
 * Every feature must be NRX_SPARSE and named in place_feats; workspace as nrx_embed_bwd_placed (required).
 * NRX_ERR_UNSUPPORTED (nothing enqueued) outside the placement pass's shapes (dim 16 / 32 / 64, 16-byte-aligned operands).
 * aux_stream (optional, another stream of the same device; a measurement knob -- on C2 the overlap LOSES 11 us, profiles/r05_pairs_aux.txt): the
 * pair pass, the walk and the work lists are enqueued THERE, behind what `stream` holds at the call, and the placement pass on `stream`, which
 * then waits for aux_stream.  NULL: everything on `stream`, one launch after the other. */
NRX_API int nrx_embed_bwd_placed_pairs(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                               const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                               const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                               int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                               float* const* grad_tables, int32_t n_tables, int32_t accumulate,
                               uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                               const int32_t* pairs, const int64_t* n_pairs,
                               void* workspace, int64_t workspace_bytes, void* aux_stream, void* stream);

/* nrx_embed_bwd_sorted driven by a placement plan (nrx_sparse_plan_place): same arguments and the same values[] -- bit for
 * bit -- plus dest / walk / n_walk (all three NULL: no placement, every row is walked).  place_feats must be the mask the plan
 * was made with; it may only name NRX_SPARSE features.  Launch shapes outside the fast form (odd dims, unaligned FM inputs)
 * ignore the placement and walk every row.
 * workspace_bytes: the size of `workspace` (0 = unknown: taken to be nrx_embed_bwd_sorted_workspace's).  With at least
 * nrx_embed_bwd_workspace_for(feats, ...) bytes the bag features with 0/1 weights (masked mean over DataReader's masks, mean) are
 * reduced from PRE-SCALED upstream rows kept in the workspace -- one row request per lookup instead of a row, a factor and a
 * weight bit -- and feats[i].index (the ids, optional here) lets the launch skip the weight bits when every zero weight sits on a
 * padding id. */
NRX_API int64_t nrx_embed_bwd_workspace_for(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim);
NRX_API int nrx_embed_bwd_placed(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                         const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                         const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                         int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                         uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* Fused row-sparse Adam(W) over the unique rows of nrx_sparse_plan / nrx_embed_bwd_sorted: for every unique entry u
 * (key = (table << 40) | row, gradient grads[u, :dim]) of up to NRX_MAX_FEATURES tables sharing `dim`,
 *   m += (g - m)(1 - beta1);  v += (g*g - v)(1 - beta2);  w -= w * lr_times_weight_decay;
 *   w -= step_size * m / (sqrt(v) + eps)        with step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
 * -- torch.optim.SparseAdam's update (plus optional decoupled decay of the touched rows); rows that were not
 * looked up do not move, row 0 (padding) never moves.  This replaces, for the embedding tables, the reference's
 * dense AdamW over every row (src/model/sort/deep/model.py:54-65) -- a documented deviation, see DESIGN.md.
 * n_unique_dev (optional, device int64[1]): actual count; n_unique is then an upper bound sizing the launch.
 * step_size_dev (optional, device float[1]): overrides step_size with a value read on the device, so a training
 * loop captured in a hipGraph can advance the bias correction between replays.
 * Moment layout: exp_avg[t] / exp_avg_sq[t] are [rows, dim] arrays; a table with exp_avg_sq[t] == exp_avg[t] + dim is taken
 * as ONE [rows, 2, dim] array (a row's two moments adjacent: one 128-byte line per row at dim 16). */
NRX_API int nrx_sparse_adam_step(float* const* tables, float* const* exp_avg, float* const* exp_avg_sq, int32_t n_tables,
                         int32_t dim, const int64_t* uniq_keys, const float* grads, int64_t n_unique,
                         const int64_t* n_unique_dev, float step_size, const float* step_size_dev, float beta1,
                         float beta2, float eps, float lr_times_weight_decay, void* stream);

/* Exact dense AdamW from row-sparse gradients ("exact-dense mode").  The reference trains every embedding table with one dense
 * torch.optim.AdamW over model.parameters() (src/model/sort/deep/model.py:54-65): every row moves every step.  nrx_rows_mark writes, for every
 * key i = (t << 40 | row) of a unique-key list (negative keys / tables >= n_tables: fillers; row 0: the padding row), slot_maps[t][row] = i --
 * the maps (int32 [rows[t]], all -1 before); nrx_dense_adamw_rows then does ONE AdamW step (torch's arithmetic: decoupled weight decay, bias
 * corrections of step `step` >= 1) over EVERY row of the n_tables [rows[t], dim] tables, the gradient of a row being grads[slot] where its slot
 * is >= 0 and zero elsewhere, and resets the slots it consumed to -1.  exp_avg / exp_avg_sq: [rows[t], dim] (torch's state layout).  hyper_dev
 * (optional, device): {lr / bias_correction1, 1 / sqrt(bias_correction2)} read instead of the values derived from `step` (captured loops).  No dense
 * gradient tensor is formed, zero-filled or read: 6 table-sized transfers per step instead of torch's 7 + the zero fill. */
NRX_API int nrx_rows_mark(const int64_t* uniq_keys, int64_t n, const int64_t* n_dev, int32_t* const* slot_maps, const int64_t* rows,
                  int32_t n_tables, int32_t unmark /* != 0: write -1 instead: undo a marking */, void* stream);
/* Two unique-key lists that may name the same row (DSSM's two towers share the news table, recall/DSSM/model.py:148-180: two backward launches):
 * with list A marked (nrx_rows_mark), every (key, row) pair of list B whose row A also holds is added to A's row (values_a[slot]) and its key
 * becomes the filler -1; the other pairs of B stay.  One addition per shared row: deterministic.  Afterwards A and B are disjoint. */
NRX_API int nrx_rows_merge(int64_t* keys_b, const float* values_b, int64_t n, const int64_t* n_dev, float* values_a, int32_t* const* slot_maps,
                  const int64_t* rows, int32_t n_tables, int32_t dim, void* stream);
NRX_API int nrx_dense_adamw_rows(float* const* tables, float* const* exp_avg, float* const* exp_avg_sq, int32_t* const* slot_maps,
                  const int64_t* rows, int32_t n_tables, int32_t dim, const float* grads, int64_t step, float lr, float beta1,
                  float beta2, float eps, float weight_decay, const float* hyper_dev, void* stream);

/* nrx_embed_bwd_placed with the DENSE gradient tables as its destination: every unique row's sum is stored (accumulate == 0) or added
 * (accumulate != 0: a table fed by a second launch group) at grad_tables[table][row, :dim] -- the row the key names -- instead of
 * values[u]: with zero-filled tables, the dense [rows, dim] .grad autograd gives the reference's nn.Embedding tables
 * (src/model/BaseModel/base_model.py:164), formed by the deterministic sorted reduction in ONE pass (nrx_embed_bwd_placed +
 * nrx_rows_to_dense write and re-read every unique row in between; same bits).  grad_tables: HOST array of n_tables
 * (<= NRX_MAX_FEATURES) device pointers indexed by the plan's table ids, all non-null; feats[i].table must point at the gradient of
 * feature i's table (= grad_tables[table_of[i]]); with a placement, feats[i].index of the placed features must be the ids the plan
 * was made from (the placement pass reads the row number there).  Everything else as nrx_embed_bwd_placed. */
NRX_API int nrx_embed_bwd_placed_dense(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                               const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                               const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                               int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm,
                               float* const* grad_tables, int32_t n_tables, int32_t accumulate,
                               uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* The whole deterministic dense-gradient backward of one launch group in ONE call: nrx_sparse_plan_place (or nrx_sparse_plan when place == 0
 * or the single-valued features are under a quarter of the lookups) + nrx_embed_bwd_placed_dense, all intermediates inside `workspace`
 * (nrx_embed_bwd_dense_sorted_workspace bytes).  feats[i].index = the ids, feats[i].rows the table rows, feats[i].table the GRADIENT of feature
 * i's table; table_of (HOST, n_feats): the table id of each feature; grad_tables (HOST, n_tables) as in nrx_embed_bwd_placed_dense.  Autograd of
 * nn.Embedding over every lookup feature (src/model/BaseModel/base_model.py:262-308) with the reference's dense [rows, dim] .grad, bit-reproducible. */
NRX_API int64_t nrx_embed_bwd_dense_sorted_workspace(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim, int32_t n_tables);
NRX_API int nrx_embed_bwd_dense_sorted(const nrx_feature_t* feats, const int32_t* table_of, int32_t n_feats, int32_t n_tables, int64_t batch,
                               int32_t dim, const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                               const nrx_fm_grad_t* fm, float* const* grad_tables, int32_t accumulate, int32_t place,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* nrx_embed_bwd_dense_sorted with the planner as an argument.  planner == 1: the one-kernel planner (nrx_sparse_plan_lds; `state` = its control
 * block, see there) when the launch qualifies -- every feature single-valued, dim 16 / 32 / 64, inside the planner's shapes -- else, and with
 * planner == 0, the sorted planner.  stats (optional, int64 [4], may be mapped host memory): the plan's duplicate statistics in nrx_sparse_plan_lds's
 * format, from either planner: what the caller's choice of planner for the NEXT batch needs.  Same gradients from both, bit for bit. */
NRX_API int64_t nrx_embed_bwd_dense_planned_workspace(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim, int32_t n_tables);
NRX_API int nrx_embed_bwd_dense_planned(const nrx_feature_t* feats, const int32_t* table_of, int32_t n_feats, int32_t n_tables, int64_t batch,
                                int32_t dim, const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                const nrx_fm_grad_t* fm, float* const* grad_tables, int32_t accumulate, int32_t planner,
                                void* state, int64_t* stats, void* workspace, int64_t workspace_bytes, void* stream);

/* The ROW-SPARSE counterpart of nrx_embed_bwd_dense_planned (what the fused optimizer's sink takes): plan + reduction of one launch group in ONE
 * call.  uniq_keys [n] / values [n, dim] / counts [n_tables + 2] (n = the launch's lookups: the worst case) are the caller's -- they outlive the call;
 * every intermediate lives in `workspace` (nrx_embed_bwd_sparse_planned_workspace bytes, reusable by the next call on the same stream).  planner /
 * state / stats as nrx_embed_bwd_dense_planned.  Same keys and rows as nrx_sparse_plan_place + nrx_embed_bwd_placed, bit for bit. */
NRX_API int64_t nrx_embed_bwd_sparse_planned_workspace(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim, int32_t n_tables);
NRX_API int nrx_embed_bwd_sparse_planned(const nrx_feature_t* feats, const int32_t* table_of, int32_t n_feats, int32_t n_tables, int64_t batch,
                                 int32_t dim, const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld,
                                 const nrx_fm_grad_t* fm, int64_t* uniq_keys, float* values, int64_t* counts, int32_t planner,
                                 void* state, int64_t* stats, void* workspace, int64_t workspace_bytes, void* stream);

/* Unique-row gradients -> dense gradient tables: for every unique entry u of nrx_sparse_plan / nrx_embed_bwd_sorted
 * (key = (table << 40) | row, gradient rows[u, :dim]), tables[table][row, :dim] = rows[u] (accumulate == 0) or += rows[u]
 * (accumulate != 0: a table fed by more than one reduction).  With zero-filled tables this forms what autograd gives the
 * reference's nn.Embedding(size, dim, padding_idx=0) tables (src/model/BaseModel/base_model.py:164) -- a dense [rows, dim] .grad --
 * from the deterministic sorted reduction, instead of the float atomics of nrx_embed_bwd.  tables: HOST array of n_tables
 * (<= NRX_MAX_FEATURES) device pointers, all non-null.  n_unique_dev (optional, device int64[1]): actual count; n_unique is
 * then an upper bound sizing the launch. */
NRX_API int nrx_rows_to_dense(float* const* tables, int32_t n_tables, int32_t dim, const int64_t* uniq_keys, const float* rows,
                      int64_t n_unique, const int64_t* n_unique_dev, int32_t accumulate, void* stream);

/* Composite sort keys for nrx_embed_bwd_sorted over SEVERAL tables at once: for the flat,
 * feature-major lookup list of n_feats features (ids[f]: lens[f] elements; HOST pointer arrays),
 * keys[p] = (table_of[f] << 40) | id, so that one stable sort groups the lookups by (table, row).
 * Negative ids map to row 0 of their table (the padding row, which never trains).               */
NRX_API int nrx_make_table_keys(const void* const* ids, const int64_t* lens, const int32_t* table_of,
                        int32_t n_feats, int32_t index_bits, int64_t* keys, void* stream);

/* ---- standalone pooling on materialised embeddings -------------------------------------------
 * BaseModel.array_feature_pooling(emb[B,L,D], mask[B,L] | None) (base_model.py:273-282).       */
NRX_API int nrx_bag_pool_fwd(const float* emb, const float* mask, int64_t batch, int32_t bag_len,
                     int32_t dim, float* out, void* stream);
NRX_API int nrx_bag_pool_bwd(const float* g_out, const float* mask, int64_t batch, int32_t bag_len,
                     int32_t dim, float* g_emb, void* stream);

/* ---- standalone FM on a concat [B, n_fields*dim] (all fields share dim) ---------------------
 * FM.get_inp_embedding + FMModel.forward without bias/sigmoid (fm/model.py:18-26,48-59).       */
NRX_API int nrx_fm_fwd(const float* feat, int64_t ld, int32_t n_fields, int32_t dim, int64_t batch,
               float* fm_out, void* stream);
/* Training form of nrx_fm_fwd: also leaves the field sums nrx_embed_fwd_train's epilogue leaves (fm_sums [B, sums_ld]: column k >= 1 sum_f v_fk,
 * column 0 sum_f w_f; sums_ld >= dim) -- for a concat that was finished by someone else (the owners of a one-sided sharded forward). */
NRX_API int nrx_fm_fwd_train(const float* feat, int64_t ld, int32_t n_fields, int32_t dim, int64_t batch,
                     float* fm_out, float* fm_sums, int64_t sums_ld, void* stream);
/* g_feat[b, :] = (g_in ? g_in[b, :] : 0) + g_fm[b] * d fm / d feat[b, :].  g_in may be g_feat (in place) or another
 * buffer (e.g. the upstream gradient of the concat, left untouched: no copy needed) or NULL.                 */
NRX_API int nrx_fm_bwd(const float* feat, int64_t ld, int32_t n_fields, int32_t dim, int64_t batch,
               const float* g_fm, const float* g_in, int64_t g_in_ld, float* g_feat, int64_t g_ld, void* stream);

/* FM head: out[b] = sigmoid(bias[0] + logit[b]) -- the last line of FMModel.forward (sort/fm/model.py:25-26; bias: a device scalar, NULL = 0) -- and
 * its autograd in one launch: g_logit[b] = g_out[b * g_stride] * out[b] (1 - out[b]) (g_stride 0: one upstream value for every sample, what the
 * gradient of a sum is) and g_bias[0] = sum_b g_logit[b] (NULL: not wanted), summed in a fixed order (bit-reproducible).  state: nrx_fm_head_state_bytes()
 * device bytes, zero before the first use, then owned by the call (one stream at a time). */
NRX_API int nrx_fm_head_fwd(const float* logit, const float* bias, float* out, int64_t batch, void* stream);
NRX_API int64_t nrx_fm_head_state_bytes(void);
NRX_API int nrx_fm_head_bwd(const float* g_out, int64_t g_stride, const float* out, float* g_logit, float* g_bias, void* state, int64_t batch,
                    void* stream);

/* ---- DCN v1 cross network: x_{l+1} = x0 * (x_l . w_l) + b_l + x_l ------------------------------
 * DCNLayer.forward / DCNNet.forward (dcn/dcn_arch.py:14-30, 63-70) in the algebraic O(B*D)
 * form (the reference materialises a [B,D,D] outer product).  w, b: device [n_layers, dim].
 * All layers run in one launch with the row held in registers.  x is the input of the FIRST of the n_layers
 * layers; x0 is the layer-0 input of the whole cross network (DCNLayer.forward(x_l, x_0) takes both,
 * dcn_arch.py:14): NULL (or x itself) when the stack starts at layer 0 -- DCNNet.forward's case.
 * out may alias a different column block of the same buffer as x (e.g. out = x + dim with ld = 2*dim
 * gives cat[x, cross], dcn/model.py:29).                                                          */
NRX_API int nrx_dcn_v1_fwd(const float* x, int64_t x_ld, const float* x0, int64_t x0_ld, int64_t batch, int32_t dim,
                   int32_t n_layers, const float* w, const float* b, float* out, int64_t out_ld, void* stream);
/* g_w, g_b: device [n_layers, dim], pre-zeroed, accumulated with atomics.  With x0 == NULL, g_x0 must be NULL
 * and g_x receives the whole input gradient; with a separate x0, g_x = dL/dx (the stack's first input) and
 * g_x0 = dL/dx0. */
NRX_API int nrx_dcn_v1_bwd(const float* x, int64_t x_ld, const float* x0, int64_t x0_ld, int64_t batch, int32_t dim,
                   int32_t n_layers, const float* w, const float* b, const float* g_out, int64_t g_out_ld,
                   float* g_x, int64_t g_x_ld, float* g_x0, int64_t g_x0_ld, float* g_w, float* g_b, void* stream);
/* nrx_dcn_v1_bwd with g_w / g_b (sums over the batch) added block by block in BLOCK ORDER by a second launch instead of with float atomics: the same
 * bits run to run (g_x / g_x0 never depended on atomics).  g_w / g_b are overwritten (no zero fill needed).  workspace:
 * nrx_dcn_v1_bwd_ordered_workspace(dim, n_layers) device bytes.  NRX_ERR_UNSUPPORTED (nothing enqueued) when n_layers x dim is beyond the LDS slabs
 * of the fixed-order block sum (the per-row LDS-atomic body: e.g. 8 layers x 2048). */
NRX_API int64_t nrx_dcn_v1_bwd_ordered_workspace(int32_t dim, int32_t n_layers);
NRX_API int nrx_dcn_v1_bwd_ordered(const float* x, int64_t x_ld, const float* x0, int64_t x0_ld, int64_t batch, int32_t dim,
                                   int32_t n_layers, const float* w, const float* b, const float* g_out, int64_t g_out_ld,
                                   float* g_x, int64_t g_x_ld, float* g_x0, int64_t g_x0_ld, float* g_w, float* g_b, void* workspace,
                                   void* stream);

/* Fused gather -> concat -> DCN-v1 cross for the DCN ranker (dcn/model.py:25-29 on top of
 * base_model.py:284-308): out[:, 0:width] = concat of the looked-up rows (x), out[:, width:2*width] =
 * cross(x); the rows stay in registers through all cross layers, x is never re-read.  Restrictions
 * (else NRX_ERR_UNSUPPORTED; use nrx_embed_fwd + nrx_dcn_v1_fwd): every feature NRX_SPARSE with
 * dim % 4 == 0, out_col % 4 == 0, width % 4 == 0, width <= 2048, 16-byte aligned tables / out,
 * out_ld % 4 == 0 and out_ld >= 2*width.  Bit-identical to the two-launch path.                */
NRX_API int nrx_embed_dcn_v1_fwd(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t width,
                         float* out, int64_t out_ld, int32_t n_layers, const float* w, const float* b,
                         int32_t* status, void* stream);

/* ---- DCN v2 cross layer on the matrix cores: out = act(x0 * (x_l W^T + bias) + x_l) -----------
 * DCNv2Layer.forward + the ReLU DCNv2Net puts after every layer (dcn_arch.py:33-50, 73-91).
 * W: device [dim, dim] (nn.Linear weight: out x in), bias [dim].  fp32 in / fp32 accumulate on
 * v_mfma_f32_32x32x2_f32 (exact f32 fma chain).  relu: FLAGS -- bit 0: apply ReLU (reference) / none; bit 1 (opt-in,
 * dcn_cfg.math = bf16x3): split-bf16 matrix math -- every operand as two bfloat16 parts, x W^T ~= xh wh + xh wl + xl wh on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation (16 significant operand bits; |err| ~5e-6 max|x W^T| at dim 320, stated in
 * tests/test_dcn2_bf16x3.py); aligned shapes only, others take the fp32 form.
 * lin_out (optional, [batch, ld]): x_l W^T + bias before the Hadamard -- what the backward needs (training).      */
NRX_API int nrx_dcn_v2_layer_fwd(const float* x0, const float* xl, int64_t ld, int64_t batch, int32_t dim,
                         const float* W, const float* bias, int32_t relu, float* out,
                         int64_t out_ld, float* lin_out, void* stream);
/* Backward of one DCN-v2 layer on the matrix cores (autograd of dcn_arch.py:39-50 + the ReLU of :80):
 *   gm = g_out * (out > 0) [relu] ;  glin = gm * x0 ;  g_x0 (+)= gm * lin ;  g_b = sum_rows glin ;
 *   g_xl = gm + glin W   (MFMA dgrad) ;  g_W = glin^T x_l   (MFMA wgrad, split over the batch, fp32 atomics).
 * lin: the forward's lin_out; out: the forward's output (ReLU mask; may be NULL when relu == 0).  g_x0 is overwritten,
 * or accumulated into when bit 0 of accumulate_x0 is set (x0 feeds every layer); bit 1 additionally adds the resulting g_x0
 * into g_xl in the dgrad epilogue -- the stack's FIRST layer, whose x_l is x0, so that g_xl is dL/dx of the whole stack.  g_W [dim, dim] and g_b [dim] are overwritten.
 * relu: bit 0 = ReLU, bit 1 = split-bf16 matrix math (as nrx_dcn_v2_layer_fwd), bit 2 = ORDERED wgrad: g_W / g_b summed over the batch slices in a fixed
 * order instead of with float atomics (bit-reproducible; the wgrad then runs in fp32 math whatever bit 1 says); layers up to 128 wide take the
 * ordered mode by default (it costs nothing there).
 * Launches: elementwise preparation + dgrad + wgrad (+ a fill); for dim <= 112 with aligned operands and fp32 math the preparation and the
 * dgrad are ONE launch over 64-row panels (csrc/nrx_dcn2_bwd.hip, dcn2_bwd_panel_kernel: same g_xl / g_x0 value for value).
 * workspace: nrx_dcn_v2_layer_bwd_workspace(batch, dim) device bytes.                                           */
NRX_API int64_t nrx_dcn_v2_layer_bwd_workspace(int64_t batch, int32_t dim);
NRX_API int nrx_dcn_v2_layer_bwd(const float* x0, const float* xl, int64_t ld, const float* lin, const float* out, int32_t relu,
                         int64_t batch, int32_t dim, const float* W, const float* g_out, int64_t g_ld,
                         float* g_xl, int64_t gxl_ld, float* g_x0, int64_t gx0_ld, int32_t accumulate_x0,
                         float* g_W, float* g_b, void* workspace, void* stream);

/* Weight gradient of a dense layer fed by the path's concat (the MLP heads of the rankers, src/model/model_utils/utils.py:6-17:
 * y = a W^T + b  =>  g_W[o, i] = sum_b g[b, o] a[b, i]): the batch is the contraction, which the vendor GEMMs serve badly at
 * B = 65 536 (0.2-0.3 ms per layer); this is the split-over-the-batch matrix-core kernel of nrx_dcn_v2_layer_bwd's wgrad with
 * M = out_features, N = in_features.  g_W [out_features, in_features] (contiguous) is overwritten; g_b [out_features] (optional, may be
 * NULL) receives the bias gradient sum_b g[b, o], taken from the same pass over g; fp32 atomics across the batch slices (summation
 * order not fixed).                                                                                                 */
NRX_API int nrx_linear_wgrad(const float* g, int64_t g_ld, const float* a, int64_t a_ld, int64_t batch, int32_t out_features,
                     int32_t in_features, float* g_W, float* g_b, void* stream);
/* nrx_linear_wgrad with g_W / g_b summed over the batch in a FIXED order -- the blocks store their batch slice's partial tile, a second launch adds the
 * slices in slice order: no float atomics, the same bits run to run (what GraphedStep(deterministic=True) and NRX_WGRAD=ordered use).  workspace:
 * nrx_linear_wgrad_ordered_workspace(batch, out_features, in_features) device bytes.  nrx_dcn_v2_layer_bwd takes the same mode as bit 2 of its flags. */
NRX_API int64_t nrx_linear_wgrad_ordered_workspace(int64_t batch, int32_t out_features, int32_t in_features);
NRX_API int nrx_linear_wgrad_ordered(const float* g, int64_t g_ld, const float* a, int64_t a_ld, int64_t batch, int32_t out_features,
                                     int32_t in_features, float* g_W, float* g_b, void* workspace, void* stream);

/* ---- integer utilities of the row-sharded path (bit-exact vs the CPU definitions) -------------
 * Row r of a table lives on rank r % world at local row r / world.                              */
/* Stable bucketing of a flat id list by owner rank (three deterministic passes: per-wavefront
 * histograms, scan, ballot-ranked placement; no order-dependent atomics).  Outputs (device int64):
 *   counts[world]  ids per owner;
 *   local_rows[n]  the send buffer: (id / world) grouped by owner, source order kept inside a bucket;
 *   slot[n]        slot[i] = position of source element i in that send buffer (the un-permute map).
 * workspace: device int64[nrx_bucketize_workspace(n, world)], contents undefined on entry.        */
NRX_API int64_t nrx_bucketize_workspace(int64_t n, int32_t world);
NRX_API int nrx_bucketize_by_owner(const void* ids, int32_t index_bits, int64_t n, int32_t world,
                           int64_t* counts, int64_t* local_rows, int64_t* slot,
                           int64_t* workspace, void* stream);
/* Owner-side gather of a flat id list that is segmented by table: segment s covers
 * local_rows[seg_start[s] .. seg_start[s+1]) and reads tables[seg_table[s]]; every table is `dim`
 * wide.  tables / table_rows: HOST arrays (n_tables device pointers / row counts); seg_start
 * (n_seg+1, int64) and seg_table (n_seg, int32): DEVICE arrays; n_rows = seg_start[n_seg].
 * out_rows: device [n_rows, dim].  Out-of-range rows read row 0 and are recorded in `status`.     */
NRX_API int nrx_gather_rows_segmented(const float* const* tables, const int64_t* table_rows, int32_t n_tables,
                              const int64_t* seg_start, const int32_t* seg_table, int32_t n_seg,
                              int64_t n_rows, int32_t dim, const int64_t* local_rows,
                              float* out_rows, int32_t* status, void* stream);
/* Backward of nrx_gather_rows_segmented on the owner: grad_tables[seg_table[s]][local_rows[p]] +=
 * g_rows[p] for p in segment s (fp32 atomics).  skip_row0 != 0: local row 0 is the global padding
 * row (true on rank 0 only) and receives no gradient (nn.Embedding(padding_idx=0), base_model.py:164). */
NRX_API int nrx_scatter_add_rows_segmented(float* const* grad_tables, const int64_t* table_rows, int32_t n_tables,
                                   const int64_t* seg_start, const int32_t* seg_table, int32_t n_seg,
                                   int64_t n_rows, int32_t dim, const int64_t* local_rows,
                                   const float* g_rows, int32_t skip_row0, void* stream);
/* ---- fixed-capacity routing: the sync-free form of the exchange ------------------------------
 * The ids of one exchange come as n_feats separate device arrays (ids[f]: lens[f] elements, all
 * int64 or all int32; ids / lens are HOST arrays, n_feats <= NRX_MAX_FEATURES); flat position p
 * enumerates them feature-major.  Every (source, owner) pair gets a send block of `cap` slots, so
 * the all-to-alls use equal splits and nobody needs the counts on the host:
 *   send_rows[o*cap + k]      int32 local row (id / world) of the k-th id owned by o, in source order
 *                             (slots past the block's count are left untouched: the owner learns the
 *                             count from counts2d and never reads them); ids outside [0, 2^31) go to rank 0
 *                             as -1 / INT32_MAX and are reported there as out of range;
 *   slot[p]                   int32 o*cap + k: where source position p's row will sit in the returned-row
 *                             buffer [world*cap, dim]; -1 if block o overflowed (k >= cap);   world*cap < 2^31
 *   counts2d[o*n_feats + f]   ids of feature f owned by o (the owner's inbox segmentation);
 *   overflow[0]               RUNNING MAXIMUM, since the caller last zeroed it, of the largest block count of a call:
 *                             > cap means a capacity was exceeded and the caller must redo that step exactly (a bound
 *                             launch that is replayed many times is checked once, not after every replay).
 * workspace: device int64[nrx_route_workspace(n_total, world)].  Deterministic.               */
NRX_API int64_t nrx_route_workspace(int64_t n_total, int32_t world);
NRX_API int nrx_route_ids(const void* const* ids, const int64_t* lens, int32_t n_feats, int32_t index_bits,
                  int32_t world, int64_t cap, int32_t* send_rows, int32_t* slot, int64_t* counts2d,
                  int64_t* overflow, int64_t* workspace, void* stream);
/* De-duplicated form of nrx_route_ids (per-destination dedup, SURVEY 7 hard part 1b): every DISTINCT (owner, table, local
 * row) of the exchange is sent once.  table_of[f] (HOST, n_feats): table index of feature f; table_local_rows[t] (HOST,
 * n_tables): rows of the largest local shard of table t (ids that cannot be rows are sent as that value and reported by the
 * owner).  Layout as nrx_route_ids with TABLES in the place of features: inside owner o's block the unique rows are ordered
 * by table then row, counts2d is [world, n_tables] (the owner gathers with feat_table = identity), slot[p] -- for every
 * lookup, duplicates included -- is the position of its unique entry, overflow[0] the largest block's unique count.
 * workspace: nrx_route_dedup_workspace(n_total, world) device bytes.  Deterministic (stable bit-limited radix sort).    */
NRX_API int64_t nrx_route_dedup_workspace(int64_t n_total, int32_t world);
NRX_API int nrx_route_ids_dedup(const void* const* ids, const int64_t* lens, const int32_t* table_of, const int64_t* table_local_rows,
                        int32_t n_feats, int32_t n_tables, int32_t index_bits, int32_t world, int64_t cap,
                        int32_t* send_rows, int32_t* slot, int64_t* counts2d, int64_t* overflow, void* workspace, void* stream);
/* np.unique(ids, return_inverse=True) on the device: unique_out [<= n] ascending distinct values, inverse_out [n] with
 * unique_out[inverse_out[i]] == ids[i], n_unique (device int64[1]).  workspace: nrx_unique_inverse_workspace(n) bytes.   */
NRX_API int64_t nrx_unique_inverse_workspace(int64_t n);
NRX_API int nrx_unique_inverse(const void* ids, int32_t index_bits, int64_t n, int64_t* unique_out, int64_t* inverse_out,
                       int64_t* n_unique, void* workspace, void* stream);
/* Owner side of the fixed-capacity exchange.  inbox_rows [world*cap]: block s came from rank s, its
 * valid prefix has sum_f recv2d[s*n_feats+f] entries, feature-major.  feat_table (HOST, n_feats):
 * table index of each feature.  Gathers into out_rows [world*cap, dim] (slots past a block's count
 * are not written).  recv2d is a DEVICE array: nothing is read back to the host.               */
NRX_API int nrx_gather_inbox(const float* const* tables, const int64_t* table_rows, int32_t n_tables,
                     const int32_t* feat_table, int32_t n_feats, int32_t world, int64_t cap,
                     const int64_t* recv2d, const int32_t* inbox_rows, int32_t dim,
                     float* out_rows, int32_t* status, void* stream);
/* One-sided placement (round 4): the exchange without the row buffer.  nrx_route_ids_pos = nrx_route_ids that also records, per sent
 * id, its position inside its feature (send_pos [world*cap] int32: the sample, for [B] features) -- it travels with the local rows.
 * nrx_gather_inbox_place = nrx_gather_inbox whose rows do not go to out_rows: slot j of source s's block (feature f by the counts) is
 * written to  peer_out[s] + inbox_pos[s*cap + j] * out_ld + feat_col[f]  -- its final place in the REQUESTER's concat buffer.
 * peer_out (HOST, world): rank s's [B, out_ld] buffer as THIS process can address it (hipIpcOpenMemHandle / a peer mapping over
 * xGMI; this rank's own pointer at s == rank); feat_col (HOST, n_feats): first column of every feature, a multiple of 4; dim in
 * out_rows: rows of a requester's buffer -- a position outside [0, out_rows) (it came from a peer) is dropped and reported through `status`;
 * 16..256, % 4 == 0, out_ld % 4 == 0 (else NRX_ERR_UNSUPPORTED).  The requester may read its buffer once every owner's launch
 * has completed (a collective after the launch on each rank's stream orders that).  Removes the row all-to-all AND the second pass
 * over the rows (the final launch that re-reads them): bytes per looked-up row 4D read + 4D written, once.                  */
NRX_API int nrx_route_ids_pos(const void* const* ids, const int64_t* lens, int32_t n_feats, int32_t index_bits,
                      int32_t world, int64_t cap, int32_t* send_rows, int32_t* send_pos, int32_t* slot, int64_t* counts2d,
                      int64_t* overflow, int64_t* workspace, void* stream);
NRX_API int nrx_gather_inbox_place(const float* const* tables, const int64_t* table_rows, int32_t n_tables,
                           const int32_t* feat_table, int32_t n_feats, int32_t world, int64_t cap,
                           const int64_t* recv2d, const int32_t* inbox_rows, const int32_t* inbox_pos, int32_t dim,
                           float* const* peer_out, int64_t out_ld, int64_t out_rows, const int32_t* feat_col, int32_t* status, void* stream);
/* ---- per-feature fixed-capacity routing (round 6): the exchange whose OWNER side is the single-GPU engine --------------------------------
 * nrx_route_ids packs the features of a (source, owner) block behind each other at device-resident offsets: nothing on the owner is then a
 * plain id array per feature.  Here every (source, owner, FEATURE) triple owns capf slots (n_feats single-valued features of `batch` ids each;
 * ids: HOST array of device pointers, all int32 or all int64):
 *   send_ids[(o * n_feats + f) * capf + k]  int32 OWNER ID of the k-th id of feature f owned by rank o (= id % world), in sample order:
 *                                           0 = nothing (an empty slot k >= count -- the call zero-fills the tails -- or the global padding id 0);
 *                                           v >= 1 = local row v - 1 of o's shard (id / world + 1).  Ids that cannot be rows (< 0, >= 2^31 - 1) go
 *                                           to rank 0 as -1 / INT32_MAX, where the owner's forward reports them out of range.
 *   send_pos (optional, same layout)        the sample b of that id (one-sided placement); -1 in the tails
 *   slot[f * batch + b]                     int32 (o * capf + k) * n_feats + f: the lookup's row in the returned [world * capf, n_feats, dim]
 *                                           buffer viewed as rows of `dim` floats -- and in the gradient send buffer; -1 when k >= capf
 *   counts[o * n_feats + f]                 int64 ids of feature f owned by o (may exceed capf)
 *   overflow[0]                             running maximum of those counts since the caller zeroed it (> capf: redo the step with a larger capf)
 * After an equal-split all-to-all of send_ids (n_feats * capf words per peer) and nrx_inbox_transpose, the owner holds per feature ONE array of
 * world * capf owner ids, [f][s][k]: a batch of world * capf pseudo-samples for nrx_embed_fwd, nrx_sparse_plan*, nrx_embed_bwd_* over shard
 * tables that carry a LEADING DUMMY ROW (table pointer = the dummy row, rows = local rows + 1): owner id 0 reads zeros and never trains, exactly
 * the padding row of a single-GPU table.  Rows / gradient rows travel as [s][k][f][dim] = a [world * capf, n_feats * dim] concat (out_col f * dim).
 * ONE launch; deterministic (ballot ranks + a published-totals chain over the tiles of a feature; definition: oracle/ref_np.py route_feat).
 * state: nrx_route_feat_state_bytes() device bytes, ZERO before the first call, then owned by the call (one stream at a time).
 * No reference counterpart (the reference is single-device: src/model/sort/deep/train.py:38-44).                                              */
NRX_API int64_t nrx_route_feat_state_bytes(int32_t n_feats, int64_t batch, int32_t world);
NRX_API int nrx_route_feat(const void* const* ids, int32_t n_feats, int64_t batch, int32_t index_bits, int32_t world, int64_t capf,
                   int32_t* send_ids, int32_t* send_pos, int32_t* slot, int64_t* counts, int64_t* overflow, void* state, void* stream);
/* [world][n_feats][capf] int32 (as the all-to-all leaves it) -> [n_feats][world][capf] (per-feature arrays); optionally a second array (the
 * positions) in the same launch.  capf % 4 == 0, 16-byte aligned buffers. */
NRX_API int nrx_inbox_transpose(const int32_t* inbox_a, int32_t* out_a, const int32_t* inbox_b, int32_t* out_b, int32_t world, int32_t n_feats,
                        int64_t capf, void* stream);
/* One-sided placement in the per-feature layout (the counterpart of nrx_gather_inbox_place): for every feature f and pseudo-sample b' = s * capf + k
 * with owner_pos[f][b'] >= 0, the row tables[f][owner_ids[f][b']] (tables[f] = the ARENA base: row 0 the dummy row; owner id 0 writes zeros -- a
 * lookup of the padding id) goes to peer_out[s] + owner_pos[f][b'] * out_ld + feat_col[f]: its final place in the REQUESTER's [out_rows, out_ld]
 * concat, mapped by this process (hipIpc / a peer mapping over xGMI; its own buffer at s == rank).  owner_ids / owner_pos: [n_feats][world * capf]
 * (nrx_inbox_transpose's output; at world 1 nrx_route_feat's own send arrays).  Positions >= out_rows and ids outside [0, table_rows[f]) are
 * dropped / read as zeros and reported through `status` (device int32[4] or NULL).  dim 16 .. 256 (a power of two), feat_col and out_ld
 * multiples of 4 (else NRX_ERR_UNSUPPORTED).  The requester may read its buffer once every owner's launch has completed (a collective behind the
 * launches orders that).  Removes the row all-to-all AND the final un-permuting launch of the buffer path. */
NRX_API int nrx_gather_place_feat(const float* const* tables, const int64_t* table_rows, const int32_t* feat_col, int32_t n_feats, int32_t world,
                          int64_t capf, const int32_t* owner_ids, const int32_t* owner_pos, int32_t dim, float* const* peer_out,
                          int64_t out_ld, int64_t out_rows, int32_t* status, void* stream);
/* The requester's half of the sharded backward (and any other permutation of upstream rows): values[dest[p], :dim] = the upstream row of lookup p
 * -- g_out[b, cols of feature f] with the FM term folded in as in nrx_embed_bwd (fm may be NULL) -- for the flat, feature-major lookups p = f *
 * batch + b of n_feats NRX_SPARSE features; dest[p] < 0: skipped.  Every written row is written once (dest is a partial permutation: the
 * caller's promise).  This is nrx_embed_bwd_placed's placement pass with the caller's destinations: upstream rows read where they lie
 * (sample-major, coalesced, whole 128-byte lines for 64-byte rows), one store per row.  NRX_ERR_UNSUPPORTED (nothing enqueued) outside its
 * shapes (dim 16 / 32 / 64, 16-byte aligned operands, no wide routing).  Autograd of the row copies of
 * src/model/BaseModel/base_model.py:262-271 with respect to the looked-up rows. */
NRX_API int nrx_embed_bwd_scatter(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                          const float* g_out, int64_t out_ld, const float* g_wide, int64_t wide_ld, const nrx_fm_grad_t* fm,
                          const int32_t* dest, float* values, void* stream);
/* ---- the sharded backward WITHOUT the owner's placement pass (round 6) ------------------------------------------------------------------
 * A gradient row whose table row is looked up once in the whole exchange needs no reduction: the requester can write it straight into the owner's
 * values[u].  The owner plans first (its plan depends on the owner ids only) and the plan's dest[] travels back to the requesters in send_ids'
 * layout; nrx_shard_dest_combine turns (slot, dest_req) into ONE destination per lookup -- dest_out[p] = (owner << shift) | row of the owner's
 * gradient ARENA (values rows first; from row recv_row0 on the receive buffer [source][k][f], whose row for this lookup is (rank * capf + k) *
 * n_feats + f), -1 for a dropped lookup; nrx_embed_bwd_scatter_multi (nrx_embed_bwd_scatter with `bases[owner]` = the owner's arena as this process
 * maps it: hipIpc / a peer mapping; its own at owner == rank) stores every lookup's upstream row there -- the requester's pack IS the owner's
 * placement pass -- and nrx_embed_bwd_walk reduces what is left on the owner: the listed rows (and the pair records of an nrx_sparse_plan_lds
 * plan; pairs / n_pairs NULL for a sorted plan) from the receive buffer, nrx_embed_bwd_placed(_pairs) minus the placement pass: same values bit
 * for bit.  A collective between the scatter and the walk is the completion fence.  At world 1 the backward is the direct path's: plan, one
 * placement pass, walk.  NRX_ERR_UNSUPPORTED outside the placement pass's shapes (dim 16 / 32 / 64, aligned operands). */
NRX_API int nrx_shard_dest_combine(const int32_t* slot, const int32_t* dest_req, int32_t n_feats, int64_t batch, int64_t capf, int64_t recv_row0,
                           int32_t rank, int32_t world, int32_t shift, int32_t* dest_out, void* stream);
NRX_API int nrx_embed_bwd_scatter_multi(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                                const float* g_out, int64_t out_ld, const nrx_fm_grad_t* fm, const int32_t* dest,
                                float* const* bases, int32_t n_bases, int32_t shift, void* stream);
NRX_API int nrx_embed_bwd_walk(const nrx_feature_t* feats, int32_t n_feats, int64_t batch, int32_t dim,
                       const float* g_out, int64_t out_ld, const int64_t* order, const int64_t* seg_start, const int64_t* uniq_keys,
                       int64_t n_unique, const int64_t* n_unique_dev, const nrx_fm_grad_t* fm, float* values,
                       uint64_t place_feats, const int32_t* dest, const int32_t* walk, const int64_t* n_walk,
                       const int32_t* pairs, const int64_t* n_pairs, void* workspace, int64_t workspace_bytes, void* stream);
/* Backward of nrx_gather_inbox: grad_tables[..][row] += g_rows[p] over the valid prefixes.        */
NRX_API int nrx_scatter_add_inbox(float* const* grad_tables, const int64_t* table_rows, int32_t n_tables,
                          const int32_t* feat_table, int32_t n_feats, int32_t world, int64_t cap,
                          const int64_t* recv2d, const int32_t* inbox_rows, int32_t dim,
                          const float* g_rows, int32_t skip_row0, void* stream);
/* ---- owner-side partial pooling of row-sharded bag features -------------------
 * A bag feature whose table is row-sharded does not fetch its L rows per sample across the fabric: every owner pools
 * the rows it holds and returns ONE partial vector per (sample, owner); the source adds the `world` partials.
 *   nrx_bag_norm_weights   per-lookup weight with the pooling's normalisation folded in:
 *                          masked mean  w / (sum_l w + 1e-8)   (array_feature_pooling, base_model.py:278-282)
 *                          mean         1 / L                  (:275-276);     sum: w (or 1)
 *   nrx_route_bags         like nrx_route_ids for n_feats bag features ([batch, bag_len] ids each): lookups with weight
 *                          != 0 are placed, in source order, in their owner's send block as a triple
 *                            send_rows[d] local row, send_tag[d] = f * batch + sample, send_w[d] weight
 *                          (no slot map: nothing comes back per lookup); counts2d / overflow as in nrx_route_ids
 *   nrx_pool_inbox_fwd     owner: partial[s][tag][:] = sum over source s's entries with that tag of w * table row,
 *                          in source order (deterministic, no atomics); every element of partial
 *                          [world, n_feats*batch, dim] is written; workspace: nrx_pool_inbox_workspace(...) device bytes
 *   nrx_pool_inbox_bwd     owner: grad_table[row] += w * g_partial[s][tag][:] per entry (fp32 atomics; skip_row0 as in
 *                          nrx_scatter_add_inbox)
 * The source finishes with an NRX_BAG_SUM of bag_len = world over the returned slabs (rows o*n_feats*batch + tag).  */
NRX_API int nrx_bag_norm_weights(const float* mask, int64_t batch, int32_t bag_len, int32_t kind, float* out_w, void* stream);
/* nrx_bag_norm_weights that also leaves out_inv[b]: the weight every live entry of sample b carries when the mask is 0 / 1 (1 / (sum w + 1e-8), 0
 * for an empty bag; 1 / L for the plain mean) -- what the bound sharded step's pooled backward pre-multiplies the sample's upstream row by. */
NRX_API int nrx_bag_norm_weights_inv(const float* mask, int64_t batch, int32_t bag_len, int32_t kind, float* out_w, float* out_inv, void* stream);
/* The pooled channel's backward, requester side: dst[c * copy_stride + b * dim + k] = g_out[b * ld + col + k] * (scale ? scale[b] : 1) for
 * c < copies -- the [batch, dim] block of sample gradients of one bag feature, once per owner (the bound sharded step's g_send; scale = the 1 / den
 * of nrx_bag_norm_weights_inv when the masks are 0 / 1).  Plain fp32 multiply: the same values torch's broadcast multiply gives. */
NRX_API int nrx_bag_upstream_rows(const float* g_out, int64_t ld, int32_t col, int32_t dim, int64_t batch, const float* scale, int32_t copies,
                          int64_t copy_stride, float* dst, void* stream);
NRX_API int nrx_route_bags(const void* const* ids, const float* const* weights, const int32_t* bag_lens, int32_t n_feats,
                   int32_t index_bits, int64_t batch, int32_t world, int64_t cap, int32_t* send_rows, int32_t* send_tag,
                   float* send_w, int64_t* counts2d, int64_t* overflow, int64_t* workspace, void* stream);
/* nrx_route_bags in ONE launch (round 6): same arguments and the same results bit for bit (definition: oracle/ref_np.py route_bags), with a state
 * block in the place of the workspace -- nrx_route_bags_one_state_bytes() device bytes, ZERO before the first call, then owned by the call (one
 * stream at a time).  nrx_route_feat's construction (ballot ranks + published per-tile totals, an epoch mark, relaxed agent-scope polling) with the
 * chain over all tiles of all features.  Slots past a block's count are left untouched, as nrx_route_bags leaves them. */
NRX_API int64_t nrx_route_bags_one_state_bytes(const int32_t* bag_lens, int32_t n_feats, int64_t batch, int32_t world);
NRX_API int nrx_route_bags_one(const void* const* ids, const float* const* weights, const int32_t* bag_lens, int32_t n_feats,
                       int32_t index_bits, int64_t batch, int32_t world, int64_t cap, int32_t* send_rows, int32_t* send_tag,
                       float* send_w, int64_t* counts2d, int64_t* overflow, void* state, void* stream);
/* The routing launch with the launches either side of it folded in (round 6; replaces nrx_bag_norm_weights + nrx_route_bags_one at the source and
 * the memset + marking pass inside nrx_pool_inbox_fwd at the owner).  masks[f] = the RAW mask / weights of feature f (null: all ones), kinds[f] =
 * NRX_BAG_MASKED_MEAN | NRX_BAG_MEAN | NRX_BAG_SUM: the weight that travels is mask / den, formed in the launch with nrx_bag_norm_weights' own
 * arithmetic and summation order (bit-identical; inv_out[f], optional: nrx_bag_norm_weights_inv's out_inv).  Tiles hold WHOLE samples, so the
 * entries of one (sample, owner) are contiguous in the owner's block and the tile writes their bounds: send_run [world][n_feats * batch][2] int32 =
 * {first slot, one past the last} of tag's run inside block o ({0, 0}: no entry; clamped to cap) -- send_tag is not written at all.  send_rows /
 * send_w / counts2d / overflow: as nrx_route_bags_one (definition: oracle/ref_np.py route_bags + bag_norm_weights; the runs: route_bags_runs).
 * nrx_route_bags_runs_state_bytes returns 0 for shapes outside this form (a bag longer than 4096 entries, or 4096 / bag_len * world > 4096):
 * use nrx_route_bags_one there.  The state block is ZERO before the first call, then owned by the call. */
NRX_API int64_t nrx_route_bags_runs_state_bytes(const int32_t* bag_lens, int32_t n_feats, int64_t batch, int32_t world);
NRX_API int nrx_route_bags_runs(const void* const* ids, const float* const* masks, const int32_t* kinds, const int32_t* bag_lens, int32_t n_feats,
                        int32_t index_bits, int64_t batch, int32_t world, int64_t cap, int32_t* send_rows, float* send_w, int32_t* send_run,
                        float* const* inv_out, int64_t* counts2d, int64_t* overflow, void* state, void* stream);
/* nrx_pool_inbox_fwd over run bounds that arrived with the exchange (block s of `run` = source s's send_run block for this owner): the pooling
 * launch alone; same partial sums bit for bit. */
NRX_API int nrx_pool_inbox_fwd_runs(const float* const* tables, const int64_t* table_rows, int32_t n_tables, const int32_t* feat_table,
                            int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                            const int32_t* inbox_rows, const float* inbox_w, const int32_t* run, int32_t dim, float* partial,
                            int32_t* status, void* stream);
/* The training step's per-entry words of the runs form, from the run bounds (the tags stayed at the source): tag_out [world * cap] (what
 * nrx_pool_inbox_expand reads as inbox_tag; only the entries inside runs are written), and / or oid_out (+ payload_out) [world * cap] =
 * nrx_pool_inbox_owner_ids' words exactly (table_rows / skip_row0 as there; every slot written, owner id 0 past a block's count). */
NRX_API int nrx_pool_inbox_runs_words(int64_t table_rows, int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                              const int32_t* inbox_rows, const int32_t* run, int32_t skip_row0, int32_t* tag_out, int32_t* oid_out,
                              uint32_t* payload_out, void* stream);
NRX_API int64_t nrx_pool_inbox_workspace(int32_t n_feats, int64_t batch, int32_t world);
NRX_API int nrx_pool_inbox_fwd(const float* const* tables, const int64_t* table_rows, int32_t n_tables, const int32_t* feat_table,
                       int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                       const int32_t* inbox_rows, const int32_t* inbox_tag, const float* inbox_w, int32_t dim,
                       float* partial, void* workspace, int32_t* status, void* stream);
NRX_API int nrx_pool_inbox_bwd(float* const* grad_tables, const int64_t* table_rows, int32_t n_tables, const int32_t* feat_table,
                       int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                       const int32_t* inbox_rows, const int32_t* inbox_tag, const float* inbox_w, int32_t dim,
                       const float* g_partial, int32_t skip_row0, void* stream);
/* The pooled channel's backward in the single-GPU engine's terms (round 6; the bags of ONE table): every inbox entry e = s * cap + j of
 * nrx_route_bags becomes one pseudo-lookup -- owner_ids[e] (int32 [world * cap]): 0 = nothing (past the block's count, the global padding row
 * when skip_row0, a row outside [0, table_rows)), else local row + 1 = the row of the shard's ARENA (leading dummy row); g_rows[e, :dim] =
 * inbox_w[e] * g_partial[s][inbox_tag[e]] (written for the live entries only).  A single-valued feature of world * cap pseudo-samples over the
 * arena, upstream rows g_rows: nrx_sparse_plan* + nrx_embed_bwd_placed reduce it to deterministic row-sparse (keys, values) -- what
 * nrx_pool_inbox_bwd does with float atomics into a dense shard gradient.  Autograd of the owner-side partial pooling
 * (array_feature_pooling, src/model/BaseModel/base_model.py:273-282, split over the owners). */
NRX_API int nrx_pool_inbox_expand(int64_t table_rows, int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                          const int32_t* inbox_rows, const int32_t* inbox_tag, const float* inbox_w, int32_t dim,
                          const float* g_partial, int32_t skip_row0, int32_t* owner_ids, float* g_rows, void* stream);
/* g_rows == NULL: the owner ids alone (g_partial is then not read).  With them and nrx_pool_order_remap the expansion is never materialised: a
 * plan of the owner ids (nrx_sparse_plan_place with place_feats = 0: every row is walked) lists inbox entries in order[]; the remap replaces
 * every entry e = s * cap + j by s * n_tags + inbox_tag[e] -- the row of the [world * n_tags, dim] block of SAMPLE gradients the entry's upstream
 * row is (n_tags = n_feats * batch) -- and nrx_embed_bwd_placed then reduces ONE single-valued feature whose g_out is that block (4 MB per
 * source instead of a [world * cap, dim] array of expanded rows: the walk reads an L2-resident array, as the single-GPU bag backward does).
 * The per-entry weight must already be IN the block: exact when all non-zero weights of a sample are equal (DataReader's 0/1 masks,
 * src/dataset/DataReader/data_reader.py:96-109; mean pooling) -- the requester then sends g * w_sample; else use the expansion. */
/* The owner ids of nrx_pool_inbox_expand alone (one thread per entry) and, optionally, every entry's payload for
 * nrx_sparse_plan_ex(NRX_PLAN_PAYLOAD): payload[e] = s * n_feats * batch + inbox_tag[e] (0 for the entries that are nothing) -- the plan's order[]
 * then names the rows of the block of sample gradients directly and nrx_pool_order_remap is not needed. */
NRX_API int nrx_pool_inbox_owner_ids(int64_t table_rows, int32_t n_feats, int64_t batch, int32_t world, int64_t cap, const int64_t* recv2d,
                             const int32_t* inbox_rows, const int32_t* inbox_tag, int32_t skip_row0, int32_t* owner_ids, uint32_t* payload,
                             void* stream);
NRX_API int nrx_pool_order_remap(int64_t* order, int64_t n_entries, const int32_t* inbox_tag, int64_t cap, int64_t n_tags, int32_t world, void* stream);
/* Expands a CSR batch of an array feature (values[offsets[b] .. offsets[b+1]), offsets relative to the
 * batch, device int64[batch+1]) into the reference's padded form: ids [batch, bag_len] (0-padded,
 * same integer width as `values`) and mask float32 [batch, bag_len] (1 = real, 0 = padding) --
 * what DataReader builds per sample on the host (src/dataset/DataReader/data_reader.py:96-109;
 * longer arrays are truncated to bag_len like :104-106).  rows (optional, device int64[batch]): batch row b
 * is row rows[b] of a larger CSR (a dataset kept resident in HBM; offsets then index the whole dataset). */
NRX_API int nrx_csr_to_padded(const void* values, int32_t value_bits, const int64_t* offsets, const int64_t* rows, int64_t batch,
                      int32_t bag_len, void* ids_out, float* mask_out, void* stream);
/* Per-user ranking metrics of the reference's validation loop (base_model.py:333-435) on the device.
 * Inputs are the validation samples sorted by (user, score descending, arrival order) -- i.e. every
 * user's samples are contiguous, seg_start[u] .. seg_start[u+1), and inside a segment they are in the
 * order of Python's stable sorted(items, key=score, reverse=True).  scores fp32, labels fp32 (1 = positive).
 * Outputs, fp64 [n_users] each: auc (NaN when the user has a single class -- the reference skips those),
 * ndcg@k, hr@k, mrr@k (0 for users without positives, as the reference records them).            */
NRX_API int nrx_user_rank_metrics(const float* scores, const float* labels, const int64_t* seg_start, int64_t n_users,
                          int32_t k, double* auc, double* ndcg, double* hr, double* mrr, void* stream);
/* Exact inner-product top-k retrieval (replaces faiss.IndexFlatIP.search as wrapped by
 * src/model/model_utils/TopKSearcher.py:50-84 and used by DSSM.hit_rate, recall/DSSM/model.py:182-228).
 * items [n_items, dim], queries [n_queries, dim] fp32 row-major (dim % 4 == 0, dim <= 128, k <= 32 per call;
 * a caller that wants more runs ceil(k/32) calls, each excluding what the earlier ones returned).
 * Optional per-query exclusion lists in CSR form (excl_offsets [n_queries+1], excl_items sorted ascending
 * inside each list; both device int64): excluded items never enter the result -- the reference instead
 * over-fetches k + len(history) and filters on the host.  Output: out_idx / out_score [n_queries, k], scores
 * descending, ties broken toward the lower item index; unused slots (fewer than k candidates) = -1 / -FLT_MAX
 * (what faiss's heap leaves there).  score = fp32 fma chain, element order j, H+j (H = half of dim padded to 8, 16, 32, 64 or 128).
 * workspace: device bytes >= nrx_topk_workspace(n_items, n_queries, k).                          */
NRX_API int64_t nrx_topk_workspace(int64_t n_items, int64_t n_queries, int32_t k);
NRX_API int nrx_topk_ip(const float* items, int64_t n_items, int32_t dim, const float* queries, int64_t n_queries,
                int32_t k, const int64_t* excl_offsets, const int64_t* excl_items,
                int64_t* out_idx, float* out_score, void* workspace, void* stream);
/* lens[b] = #(mask[b,:] != 0); used to build CSR offsets from the reference's padded masks. */
NRX_API int nrx_mask_lengths(const float* mask, int64_t batch, int32_t bag_len, int64_t* lens, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NRX_EMBED_H */
