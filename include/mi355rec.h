/*
 * mi355rec.h - C ABI of libmi355rec.so (gfx950 / MI355X).
 *
 * The reference (wuchaochen/recommendation) has no native boundary: its hot path is the
 * Python class API of recommendation/code/r_model.py, whose arithmetic executes inside
 * TensorFlow.  This header is the boundary that replaces those TensorFlow op calls.  Each
 * entry point cites the reference lines whose computation it takes over.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Every pointer is DEVICE memory owned by the
 *     caller unless the comment says "host".
 *   - Every function returns 0 on success or a negative MI_ERR_* code; mi_last_error()
 *     gives a host string for the calling thread's last failure.
 *   - Work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises
 *     the host.  No entry point allocates device memory: scratch comes in through a
 *     caller-provided workspace whose size a *_workspace_bytes() call reports.
 *   - Bad ids are REPORTED, never clamped: kernels OR bits into the caller's device
 *     error word `err` (MI_IDERR_*), the entry skipped; the caller reads it back when it
 *     wants the check (the reference raises InvalidArgumentError at session.run).
 *   - Re-entrant across distinct streams and workspaces.  The library's only state of its own: the
 *     per-thread error string; the RCCL communicators behind mi_comm_* handles (one table per
 *     process); and the built-in scratch array of mi_embag_bwd_prepare_noalloc (one copy per
 *     device, shared by every call on that device - see its comment).  Everything else lives in
 *     the caller's buffers.
 */
#ifndef MI355REC_H
#define MI355REC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_ERR_INVALID_ARG (-1) /* shape / alignment / enum check failed on the host   */
#define MI_ERR_HIP (-2)         /* a HIP runtime call or kernel launch failed           */
#define MI_ERR_WORKSPACE (-3)   /* workspace too small                                  */
#define MI_ERR_UNSUPPORTED (-4) /* valid request outside what the kernels are built for */

/* bits a kernel may OR into the device error word */
#define MI_IDERR_RANGE 1    /* id >= rows of its table (or negative where not allowed)    */
#define MI_IDERR_ORDER 2    /* ids of one bag not strictly increasing (repeat / disorder) */
#define MI_IDERR_INTERNAL 4 /* reserved: no kernel of this library waits on another workgroup, so none raises it; the
                              host reports any bit it does not know as an internal device error */
#define MI_IDERR_CAPACITY 8 /* more rows for one rank pair than the exchange block holds  */

/* flags for id validation */
#define MI_CHECK_RANGE 1     /* id must be < rows                                         */
#define MI_CHECK_INCREASING 2 /* ids of a bag strictly increasing (tf.sparse_to_dense)    */
#define MI_CHECK_NONNEG 4    /* a negative id is an error instead of a padding entry      */

#define MI_POOL_SUM 0
#define MI_POOL_MEAN 1

#define MI_DT_F32 0
#define MI_DT_BF16 1

const char* mi_version(void);
const char* mi_last_error(void);

/* ------------------------------------------------------------------------------------
 * Bags.  A "bag" is the list of ids one sample feeds to one table.  n_bags = B * T and
 * bag i belongs to sample i / T and table i % T.  Ids are table-local, int32 or int64
 * (idx_bits).  Layout: either fixed width (offsets == NULL, bag i = indices[i*bag_len ..
 * (i+1)*bag_len)) or CSR (offsets[n_bags+1], int64).  A negative id is a padding entry.
 * All T tables share `dim` and live in ONE row-major fp32 array `weights[total_rows][dim]`;
 * table t owns rows [table_offsets[t], table_offsets[t+1]) (int64, device, T+1 entries).
 * ---------------------------------------------------------------------------------- */

/* Embedding gather + sum/mean pooling, all tables in one launch.
 * out[(i/T)*out_ld + (i%T)*dim + d] = pool_j weights[table_offsets[i%T] + id_j][d].
 * Replaces one-hot/n-hot materialisation followed by the first tf.layers.dense MatMul:
 * r_model.py:60-66 (one-hot), :68-80 (n-hot == sum pooling), :82-96 (padding id),
 * :53-54 (the MatMul and BiasAdd the dense matrix feeds).  `bias` (T*dim floats, added to
 * every pooled row after pooling) may be NULL. */
int mi_embag_fwd(const float* weights, const int64_t* table_offsets, int T, int dim,
                 const void* indices, int idx_bits, const int64_t* offsets, int bag_len,
                 int64_t n_bags, int pool_mode, int check_flags, const float* bias,
                 void* out, int out_dtype, int64_t out_ld, int32_t* err, void* stream);

/* Dense one-hot / n-hot matrix exactly as the reference materialises it (kept for callers
 * that consume the dense matrix itself, e.g. the label tensor of the loss):
 * out[b][c] = 1.0 if c is among the non-negative ids of bag b else 0.0; ids validated with
 * check_flags against `size`.  r_model.py:60-96. */
int mi_onehot_dense(const void* indices, int idx_bits, int bag_len, int64_t n_bags, int size,
                    int check_flags, float* out, int32_t* err, void* stream);

/* Validation only (tf.sparse_to_dense validate_indices=True semantics, r_model.py:65,76,90);
 * table t has table_offsets[t+1]-table_offsets[t] rows. */
int mi_validate_ids(const int64_t* table_offsets, int T, const void* indices, int idx_bits,
                    const int64_t* offsets, int bag_len, int64_t n_bags, int check_flags,
                    int32_t* err, void* stream);

/* Several id columns at once.  The reference encodes six features and the label with one
 * sparse_to_dense each (r_model.py:42-51 -> :60-96) and joins the encodings of a recommended list and its
 * clicked colour with tf.concat (r_model.py:34-37); on the device each of those is a launch of a few hundred
 * threads, so the checks of up to MI_MAX_ID_COLUMNS columns share one launch and the concatenation's id matrix
 * is written by one.  A column holds n_bags fixed-length bags of ids of ONE table of `rows` rows. */
#define MI_MAX_ID_COLUMNS 8
typedef struct mi_id_column {
  const void* ids;     /* [n_bags][bag_len], int32 or int64 (the call's idx_bits) */
  int64_t n_bags;
  int64_t rows;        /* size of the column's table (the one-hot width) */
  int32_t bag_len;
  int32_t check_flags; /* MI_CHECK_*; unused by the concatenation */
} mi_id_column;
/* mi_validate_ids for every column (same rules, same error word). */
int mi_validate_id_columns(const mi_id_column* cols, int n_cols, int idx_bits, int32_t* err, void* stream);
/* out[b] = [ids of column 0 | ids of column 1 + rows of column 0 | ...]: the ids of tf.concat(axis=1) of the
 * columns' one-hot / n-hot matrices (ids < 0, "no entry", stay as they are); every column has B bags; out is
 * [B][sum of bag_len] of the same integer type. */
int mi_concat_id_columns(const mi_id_column* cols, int n_cols, int idx_bits, int64_t B, void* out, void* stream);

/* Batch staging: dst[0..bytes) (device) = src[0..bytes), where src is host memory that the device can read
 * (pinned / registered: hipHostMalloc, hipHostRegister, torch pin_memory), copied by a KERNEL on `stream`.
 * The reference's loop reads a batch from its dataset iterator every step (local_train_utils.py:34-41,60-63 -
 * tf.data feeds the session); here a batch crosses PCIe once per step, and this is the copy for callers that
 * overlap it with the previous step on another stream: a copy kernel orders with compute kernels through device-side
 * stream dependencies (microseconds), whereas a copy-engine transfer (hipMemcpyAsync) and a kernel that wait on
 * each other go through the host runtime's event thread (~100 us per hand-over, measured).  Both pointers and
 * bytes must be multiples of 16. */
int mi_stage_from_host(void* dst, const void* src_host, size_t bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Sparse backward, step 1: deterministic sort by (table,row).
 * Entry p (position in `indices`) gets key = table_offsets[t] + id (a global row, uint32;
 * total_rows must be < 2^32-1); padding and invalid entries sort to the end and are not
 * counted.  Stable LSD radix sort, so equal rows keep ascending p.  Outputs:
 *   sorted_rows[n]  uint32  global row of each sorted entry (first n_valid meaningful)
 *   perm[n]         int32   original position p of each sorted entry
 *   unique_rows[n]  uint32  distinct rows, ascending (first n_unique meaningful)
 *   seg_offsets[n+1] int32  start of each distinct row's run in sorted_rows; entry
 *                           [n_unique] = n_valid
 *   counts[2]       int32   {n_unique, n_valid}
 * These are the bit-exact artefacts (== numpy stable argsort / unique).
 * unique_rows, seg_offsets and counts may ALL be NULL: run detection (three launches) is
 * then skipped and the apply calls find the runs in sorted_rows themselves.  In that form,
 * with fixed-width bags of at most 8192 entries per table, the whole sort is ONE launch
 * (each table sorted in the LDS of one CU) and padding / invalid entries end their TABLE's
 * block of sorted_rows (as the value total_rows) instead of the whole array.
 * No reference counterpart: TensorFlow's MatMul gradient builds the dense [V,D] gradient
 * (model_trainer.py:157-158); this is its sparse equivalent.
 * ---------------------------------------------------------------------------------- */
size_t mi_embag_bwd_workspace_bytes(int64_t n_entries);
int mi_embag_bwd_prepare(const int64_t* table_offsets, int T, int64_t total_rows,
                         const void* indices, int idx_bits, const int64_t* offsets, int bag_len,
                         int64_t n_bags, int64_t n_entries,
                         uint32_t* sorted_rows, int32_t* perm, uint32_t* unique_rows,
                         int32_t* seg_offsets, int32_t* counts,
                         void* workspace, size_t workspace_bytes, void* stream);

/* mi_embag_bwd_prepare for callers that cannot allocate device memory or bring a workspace (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  Same arguments and results as mi_embag_bwd_prepare, minus the
 * workspace: the sort scratch is a fixed array inside the library's device image (about 18.9 MB
 * per GPU, held until the process ends), so n_entries is limited to MI_NOALLOC_MAX_ENTRIES;
 * more returns MI_ERR_WORKSPACE.  The call itself allocates nothing and frees nothing.  All calls on
 * one GPU share that array: two calls that overlap in time on the same GPU (different streams or
 * threads) race; calls on one stream, or on different GPUs, do not.  The array keeps the previous
 * call's contents; every part of it is rewritten before it is read, so nothing needs clearing.
 * The FIRST call on each device looks the array's device address up (hipGetSymbolAddress), which a
 * capturing stream cannot record: make that call outside stream capture - one with n_entries = 0
 * (its buffers may be NULL) does the lookup and launches nothing; under capture an unresolved address returns
 * MI_ERR_UNSUPPORTED.  Later calls on that device may be captured and replayed. */
#define MI_NOALLOC_MAX_ENTRIES (1 << 21)
int mi_embag_bwd_prepare_noalloc(const int64_t* table_offsets, int T, int64_t total_rows,
                                 const void* indices, int idx_bits, const int64_t* offsets, int bag_len,
                                 int64_t n_bags, int64_t n_entries,
                                 uint32_t* sorted_rows, int32_t* perm, uint32_t* unique_rows,
                                 int32_t* seg_offsets, int32_t* counts, void* stream);

/* Generic stable key/value radix sort used by the above (exported for tests).
 * keys/vals are overwritten with the sorted result; key_bits = number of low bits that
 * differ (passes = ceil(key_bits/9), 8- or 9-bit digits). */
int mi_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int key_bits,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Sparse backward, step 2: segmented reduction of the bag gradients + optimizer.
 * grad_out[(i/T)*grad_ld + (i%T)*dim + d] is dL/d(pooled output of bag i); an entry of a
 * mean-pooled bag receives it divided by the bag's count of non-padding ids.
 * The gradient of distinct row r is the sum over its run, taken in a fixed order
 * (deterministic, no atomics).  `long_list` is int32 scratch of n_entries+2 ints whose
 * first two ints the caller zeroes ONCE after allocating it; every call leaves them zero
 * again (two calls sharing one long_list must not overlap in time).
 * ---------------------------------------------------------------------------------- */
typedef struct mi_bwd_plan {
  const uint32_t* sorted_rows;
  const int32_t* perm;
  const uint32_t* unique_rows; /* these three: all set, or all NULL (see mi_embag_bwd_prepare) */
  const int32_t* seg_offsets;
  const int32_t* counts;  /* {n_unique, n_valid} on device */
  int64_t n_entries;      /* capacity (upper bound of n_valid) */
  int64_t n_bags;
  int64_t total_rows;     /* as given to mi_embag_bwd_prepare */
  const int64_t* offsets; /* CSR offsets or NULL */
  int bag_len;
  int T;
  int pool_mode;
  int32_t* long_list; /* scratch, n_entries + 2 ints, zeroed once by the caller (the calls leave it zero) */
  /* needed for MI_POOL_MEAN only (the bag's count of valid ids); may be NULL for sum */
  const void* indices;
  int idx_bits;
  const int64_t* table_offsets;
  /* optional, filled by mi_embag_bag_info (either may be NULL):
   * bag_scale[n_bags]: 1 / (count of valid ids of the bag), 1 for a bag with fewer than two - what an entry of a
   *   mean-pooled bag multiplies its bag's gradient by; without it every entry recounts its bag's ids;
   * entry_bag[n_entries]: the bag of each position of `indices` for CSR bags; without it every entry finds its
   *   bag by binary search in `offsets`. */
  const float* bag_scale;
  const int32_t* entry_bag;
} mi_bwd_plan;

/* Per-bag facts the apply calls would otherwise recompute per ENTRY (see mi_bwd_plan): one wave per bag counts the
 * bag's valid ids (0 <= id < rows of its table) into bag_scale and, for CSR bags, writes entry_bag.  Either output
 * may be NULL.  No reference counterpart (the reference's bags are fixed-width and sum-pooled, r_model.py:68-80). */
int mi_embag_bag_info(const int64_t* table_offsets, int T, const void* indices, int idx_bits,
                      const int64_t* offsets, int bag_len, int64_t n_bags, float* bag_scale,
                      int32_t* entry_bag, void* stream);

/* Dense gradient dW[total_rows][dim] (zero on untouched rows): what TensorFlow's MatMul
 * gradient X^T dY yields for a one-hot X (model_trainer.py:157-158). */
int mi_embag_bwd_dense_grad(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld,
                            int dim, int64_t total_rows, float* dW, void* stream);

/* Fused sparse updates: only rows that appear in the batch are read and written. */
int mi_embag_bwd_apply_sgd(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld,
                           int dim, float* weights, float lr, void* stream);
/* state[total_rows] fp32, one accumulator per row: s += mean_d(g^2); w -= lr*g/(sqrt(s)+eps) */
int mi_embag_bwd_apply_adagrad_rowwise(const mi_bwd_plan* plan, const float* grad_out,
                                       int64_t grad_ld, int dim, float* weights, float* state,
                                       float lr, float eps, void* stream);
/* Lazy (sparse) Adam in TensorFlow's form; lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the
 * caller: m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2; w -= lr_t*m/(sqrt(v)+eps)  (touched rows only). */
int mi_embag_bwd_apply_adam_tf(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld,
                               int dim, float* weights, float* m, float* v, float lr_t,
                               float beta1, float beta2, float eps, void* stream);
/* Dense Adam in TensorFlow's form over ALL total_rows rows, gradient zero on untouched rows:
 * reproduces tf.train.AdamOptimizer on a first-layer kernel (model_trainer.py:157-158)
 * without materialising dW. */
int mi_embag_bwd_apply_adam_dense_tf(const mi_bwd_plan* plan, const float* grad_out,
                                     int64_t grad_ld, int dim, int64_t total_rows, float* weights,
                                     float* m, float* v, float lr_t, float beta1, float beta2,
                                     float eps, void* stream);

/* ------------------------------------------------------------------------------------
 * Small dense fp32 layers of the colour model (tf.layers.dense, activation=None):
 * y = x W + b, r_model.py:52-58.  x[B][K] (row stride x_ld), W[K][N], b[N], y[B][N] (y_ld).
 * ---------------------------------------------------------------------------------- */
int mi_linear_fwd(const float* x, int64_t x_ld, const float* W, const float* b, float* y,
                  int64_t y_ld, int64_t B, int K, int N, void* stream);
/* dx = dy W^T (may be NULL), dW = x^T dy, db = sum_b dy; deterministic two-stage sums.
 * workspace: mi_linear_bwd_workspace_bytes(B,K,N). */
size_t mi_linear_bwd_workspace_bytes(int64_t B, int K, int N);
int mi_linear_bwd(const float* x, int64_t x_ld, const float* W, const float* dy, int64_t dy_ld,
                  float* dx, int64_t dx_ld, float* dW, float* db, int64_t B, int K, int N,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Heads of the colour model in one launch per row block:
 *   probs = softmax(logits)                               r_model.py:99,104,109
 *   top_idx/top_val = top_k(probs, k): descending, ties -> lower index   r_model.py:100,105
 *   loss_rows[b] = -sum_c labels[b][c]*log_softmax(logits)[b][c]         r_model.py:116-118
 *   dlogits = (probs - labels) * grad_scale   (TensorFlow's fused xent gradient; an
 *             all-zero label row still gets probs*grad_scale)
 *   hit_rows[b] = #{j : label_ids[b] == top_idx[b][j]}                   r_model.py:132-138
 * Any output pointer may be NULL; labels (dense [B][N]) may be NULL when loss_rows and
 * dlogits are NULL; label_ids (int32 [B]) may be NULL when hit_rows is NULL. N <= 4096. */
int mi_softmax_heads(const float* logits, int64_t ld, int64_t B, int N, int k,
                     const float* labels, const int32_t* label_ids, float grad_scale,
                     float* probs, int32_t* top_idx, float* top_val, float* loss_rows,
                     float* dlogits, float* hit_rows, void* stream);

/* hit_rows[b] = #{j < k : labels[b] == preds[b][j]} as float (r_model.py:132-138, the
 * tile/reshape/transpose/equal/matmul-with-ones construction). */
int mi_hit_rows(const int32_t* labels, const int32_t* preds, int64_t B, int k, float* hit_rows,
                void* stream);

/* out[n] = sum_b x[b][n] for n < N (fixed order): the BiasAdd gradient. */
int mi_colsum(const float* x, int64_t ld, int64_t B, int N, float* out, void* stream);

/* out[0] = scale * sum_i x[i], summed in a fixed order (reduce_mean r_model.py:117,
 * reduce_sum / batch_size r_model.py:138-139). */
int mi_reduce_sum(const float* x, int64_t n, float scale, float* out, void* stream);

/* tf.train.AdamOptimizer applied densely to a flat fp32 buffer (model_trainer.py:157-158):
 * m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2; p -= lr_t*m/(sqrt(v)+eps).  lr_t = alpha_t of the step;
 * when lr_dev != NULL it is read from lr_dev[0] on the device (mi_adam_schedule_step). */
int mi_adam_tf_dense(float* p, float* m, float* v, const float* g, int64_t n, float lr_t,
                     const float* lr_dev, float beta1, float beta2, float eps, void* stream);

/* ------------------------------------------------------------------------------------
 * FM (not in the reference; extension named by the benchmark definition).
 * logit[b] = bias + sum_f w1[row(b,f)] + sum_j wd[j]*dense[b][j]
 *            + 0.5 * sum_k ( (sum_f V[row(b,f)][k])^2 - sum_f V[row(b,f)][k]^2 )
 * one id per field: ids[B][T].
 *
 * HBM layout: one RECORD per global row holding everything a step touches for that row -
 * the embedding V[dim], the first-order weight w1 and the optimizer state of both - so the
 * forward reads one 128-byte line per (sample, field) and the update reads and writes one
 * or two lines per distinct row, instead of six scattered lines across separate arrays.
 *   rec[row*stride + 0 .. dim)        V
 *   rec[row*stride + off_w1]          w1
 *   rec[row*stride + off_vs0 ..]      V state 0: Adagrad row accumulator (1 float) | Adam m[dim]
 *   rec[row*stride + off_vs1 ..]      V state 1: Adam v[dim]
 *   rec[row*stride + off_ws0/off_ws1] w1 state (1 float each)
 * stride is a multiple of 4 floats and rec is 16-byte aligned (128-byte records recommended);
 * vector offsets (off_vs0/off_vs1 under Adam) are multiples of 4.  Unused offsets are ignored.
 * ---------------------------------------------------------------------------------- */
typedef struct mi_fm_table {
  float* rec;
  int64_t stride;
  int dim;
  int off_w1;
  int off_vs0, off_vs1;
  int off_ws0, off_ws1;
} mi_fm_table;

#define MI_OPT_SGD 0
#define MI_OPT_ADAGRAD_ROWWISE 1
#define MI_OPT_ADAM_TF 2

/* Forward.  Also emits S[b][k] = sum_f V[row][k] (kept for backward) and, when labels !=
 * NULL, BCE-with-logits loss rows and g[b] = (sigmoid(logit)-y)*gscale. */
int mi_fm_fwd(const mi_fm_table* tab, const int64_t* table_offsets, int T, const void* ids,
              int idx_bits, int64_t B, const float* dense, int n_dense, const float* wd,
              const float* bias, const float* labels, float gscale, float* logit, float* S,
              float* loss_rows, float* g, int32_t* err, void* stream);

/* FM backward fused with the sparse update of V and w1.  For distinct row r with run
 * {b_i}: dV[r] = sum_i g[b_i]*S[b_i] - V[r]*sum_i g[b_i];  dw1[r] = sum_i g[b_i].
 * plan comes from mi_embag_bwd_prepare on ids (bag_len 1); its run arrays may be NULL.
 * The sample of entry p is p / T; entry_sample must be NULL (kept for ABI stability: a caller with
 * another entry -> sample map sorts with T = 1 and the sample index as the sort's payload, as
 * mi_shard_owner_bwd_dense does).
 * lr is alpha_t for MI_OPT_ADAM_TF (see mi_adam_tf_dense); when lr_dev != NULL the step
 * size is read from lr_dev[0] on the device instead (see mi_adam_schedule_step), which lets a
 * captured HIP graph be replayed step after step. */
int mi_fm_bwd_apply(const mi_bwd_plan* plan, const mi_fm_table* tab, const float* g,
                    const float* S, const int32_t* entry_sample, int opt, float lr,
                    const float* lr_dev, float beta1, float beta2, float eps, void* stream);

/* Dense-side FM gradients: dwd[j] = sum_b g[b]*dense[b][j], dbias = sum_b g[b]
 * (fixed-order sums).  out = dwd[n_dense] followed by dbias. */
int mi_fm_bwd_dense(const float* g, const float* dense, int n_dense, int64_t B, float* out,
                    void* stream);

/* The whole dense side of an FM step in ONE launch: the gradients above, TensorFlow-form
 * Adam on dp = {wd[n_dense], bias} with state dm/dv, and loss_out[0] = mean(loss_rows). */
int mi_fm_dense_step(const float* g, const float* dense, int n_dense, int64_t B,
                     const float* loss_rows, float* dp, float* dm, float* dv, float lr_t,
                     const float* lr_dev, float beta1, float beta2, float eps, float* loss_out,
                     void* stream);

/* mi_fm_fwd, the per-table sort of mi_embag_bwd_prepare (the form without run arrays) and,
 * when sched != NULL, mi_adam_schedule_step in ONE launch: the three need only the step's inputs,
 * so the launch lasts as long as the longest of them.  B <= mi_fm_lds_sort_single_max() (8192: a
 * table's B entries are sorted in the LDS of one to eight CUs) for ids in either layout; up to
 * mi_fm_fwd_sort_max_batch() (131072) with table-major ids only - a larger table is shared by
 * ceil(B / 4096) workgroups, each of which walks all the table's ids and sorts the entries between two
 * cuts taken from 1024 samples.  sorted_rows[B*T], perm[B*T] as mi_embag_bwd_prepare writes them.  Results are bit-identical to the separate calls.  ids_table_major != 0: ids is
 * [T][B], one contiguous array per field - how the reference feeds its features, one tensor per
 * field (r_model.py:42-51) - instead of [B][T]; a table's ids are then read as whole lines by its
 * sort workgroups (one line per id otherwise); perm keeps the positions b*T + t either way.
 * sorted_rows == perm == NULL: no sort in this launch (any B) - forward + schedule only, for a caller
 * that sorted this batch's ids earlier, beside the previous step's update (mi_embag_bwd_prepare on a
 * second stream: the sort needs nothing but the ids). */
int mi_fm_fwd_sort_max_batch(void);
int mi_fm_lds_sort_single_max(void);
int mi_fm_fwd_sort(const mi_fm_table* tab, const int64_t* table_offsets, int T, const void* ids,
                   int idx_bits, int64_t B, const float* dense, int n_dense, const float* wd,
                   const float* bias, const float* labels, float gscale, float* logit, float* S,
                   float* loss_rows, float* g, int32_t* err, int64_t total_rows,
                   uint32_t* sorted_rows, int32_t* perm, float* sched, float lr, float beta1,
                   float beta2, int ids_table_major, void* stream);

/* The sort of mi_fm_fwd_sort on its own (one launch, every table sorted in the LDS of one to four
 * CUs), for ids [B][T] or, with ids_table_major != 0, [T][B]; the same limits on B.  Same
 * sorted_rows / perm, bit for bit.  The sort needs nothing but the ids, so a training loop that knows
 * its next batch issues this on a second stream beside the CURRENT step's update and leaves the step
 * itself with forward (mi_fm_fwd_sort without sort outputs) + update. */
int mi_fm_sort_ids(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int64_t B,
                   int ids_table_major, int64_t total_rows, uint32_t* sorted_rows, int32_t* perm,
                   void* stream);

/* mi_fm_bwd_apply and mi_fm_dense_step in ONE launch (both start from g; neither needs the
 * other).  lr / lr_dev belong to the sparse side (optimizer `opt`), dense_lr_t / dense_lr_dev
 * to the dense side's Adam.  Results are bit-identical to the separate calls. */
int mi_fm_bwd_apply_dense(const mi_bwd_plan* plan, const mi_fm_table* tab, const float* g,
                          const float* S, int opt, float lr, const float* lr_dev, float beta1,
                          float beta2, float eps, const float* dense, int n_dense,
                          const float* loss_rows, float* dp, float* dm, float* dv, float dense_lr_t,
                          const float* dense_lr_dev, float* loss_out, void* stream);

/* mi_fm_bwd_apply_dense whose launch also brings in the NEXT batch: beside the update and the dense
 * side, up to 64 further workgroups copy stage_bytes (a multiple of 16) from pinned host memory
 * (stage_src_host, a host pointer the device can read: hipHostMalloc / a pinned torch tensor) to
 * stage_dst, the device block the next step reads its ids / dense features / labels from.  The
 * reference's loop reads its batch every step (recommendation/local_train_utils.py:34-41,60-63);
 * this is that read, inside the step's own launch instead of on a second stream behind events.
 * The update's results are the bits of mi_fm_bwd_apply_dense; stage_dst must not be a buffer this
 * step still reads (keep two blocks and alternate).  Needs the plan of the one-launch update. */
int mi_fm_bwd_apply_dense_stage(const mi_bwd_plan* plan, const mi_fm_table* tab, const float* g,
                                const float* S, int opt, float lr, const float* lr_dev, float beta1,
                                float beta2, float eps, const float* dense, int n_dense,
                                const float* loss_rows, float* dp, float* dm, float* dv,
                                float dense_lr_t, const float* dense_lr_dev, float* loss_out,
                                void* stage_dst, const void* stage_src_host, size_t stage_bytes,
                                void* stream);

/* tf.train.AdamOptimizer's scalar state kept on the device, as TensorFlow keeps it
 * (model_trainer.py:157): sched = {alpha_t, beta1_power, beta2_power}, initialised by the
 * caller to {0, 1, 1}.  One call = one optimizer step: both powers are multiplied by their
 * beta and alpha_t = lr*sqrt(1-beta2_power)/(1-beta1_power), all in fp32. */
int mi_adam_schedule_step(float* sched, float lr, float beta1, float beta2, void* stream);

/* ------------------------------------------------------------------------------------
 * Collectives over RCCL (xGMI inside a node), one communicator per process / GPU.
 * They replace the reference's parameter-server traffic (model_trainer.py:124-136: every variable
 * read and optimizer apply is a gRPC hop to the ps task; train_job.py:41-42 starts the workers)
 * for hosts that do not go through torch.distributed: the exchanges of the row-wise sharded tables
 * and the sum of replicated dense gradients.  Every call enqueues on `stream` and returns; counts
 * are fixed per rank pair (no size negotiation, nothing on the host waits for the device).
 * RCCL is resolved with dlopen at the first call (a copy the process already holds is reused);
 * without it the calls return MI_ERR_UNSUPPORTED.
 *   mi_comm_unique_id   rank 0 fills 128 bytes that the host hands to every rank (file, socket, MPI ...)
 *   mi_comm_init        collective over all ranks: ncclCommInitRank on the current device
 *   mi_a2a_indices      recv[s][i] = rank s's send[me][i], int32 (slot blocks / id lists)
 *   mi_a2a_pooled       the same for fp32 rows (partial pools, pooled rows, gradient rows)
 *   mi_allgather_bytes  recv[s] = rank s's send, bytes_per_rank each
 *   mi_allreduce_dense  buf <- sum over ranks, in place, fp32 (dense gradients of replicated weights)
 * ---------------------------------------------------------------------------------- */
#define MI_COMM_ID_BYTES 128
int mi_comm_unique_id(void* id_out);
int mi_comm_init(const void* unique_id, int n_ranks, int rank, void** comm_out);
int mi_comm_destroy(void* comm);
int mi_comm_size(const void* comm, int* n_ranks, int* rank);
int mi_a2a_indices(void* comm, const int32_t* send, int32_t* recv, int64_t ints_per_pair, void* stream);
int mi_a2a_pooled(void* comm, const float* send, float* recv, int64_t floats_per_pair, void* stream);
int mi_allgather_bytes(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
int mi_allreduce_dense(void* comm, float* buf, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * Row-wise sharded tables (N ranks, one process per GPU).  Global row r = table_offsets[t] + id
 * lives on rank r % N as local row r / N.  Replaces the parameter-server variable traffic of
 * model_trainer.py:124-136 (every variable read and every Adam apply a gRPC hop to the ps task).
 * The three collectives of a step are issued by the host between these calls, all with
 * fixed sizes (nothing on the host depends on the ids):
 *   all-to-all  send[N][blk] int32            requester -> owner   blk = mi_shard_block_ints(B,T)
 *   all-to-all  partial[N][B][dim+2] fp32      owner -> requester
 *   all-gather  X[B][ld] fp32 -> X_all[N*B][ld]
 * and a step is four launches: route, owner forward + sort, combine, owner backward + dense side.
 * ---------------------------------------------------------------------------------- */

/* Ints per rank pair in the row exchange: ceil(B/64)*T chunk counts, then one slot per
 * (table, sample). */
int64_t mi_shard_block_ints(int64_t B, int T);

/* Requester: route the ids of a batch [B][T].  Block o of send (for owner o) =
 *   cnt[c][t]   how many slots of table t in samples [64c, 64c+64) are owner o's
 *   slot[t][b]  the local row of (b, t) when rank o owns it, -1 otherwise (also for an id outside
 *               its table, which sets MI_IDERR_RANGE in *err). */
int mi_shard_route(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int64_t B,
                   int n_ranks, int32_t* send, int32_t* err, void* stream);

/* Owner, ONE launch: recv[N][blk] = the blocks received from every rank.  Writes
 *   partial[N][B][dim+2]   { sum v (dim), sum |v|^2, sum w1 } per (source rank, sample)
 *   sorted_rows[n]         the local rows of all valid slots, ascending (tables lie in row order)
 *   perm[n]                s*row_stride + b of every sorted entry (the row of X_all the backward
 *                          reads; row_stride >= B); stable: equal rows keep (s, b) order
 *   n_total[0]             n <= N*B*T (size sorted_rows / perm for that)
 * and, when sched != NULL, advances the Adam schedule (see mi_adam_schedule_step).  sorted_rows,
 * perm and n_total may all be NULL (forward only).  Any id skew is sorted correctly: a table with
 * more than 8192 entries here is split on further key bits inside the workgroup, only slower.
 * n_ranks * B < 2^24, T <= 64. */
int mi_shard_owner_fwd_sort(const mi_fm_table* tab, int64_t local_rows, const int64_t* table_offsets,
                            int T, int rank, int n_ranks, const int32_t* recv, int64_t B,
                            int64_t row_stride, float* partial, uint32_t* sorted_rows, int32_t* perm, int32_t* n_total,
                            float* sched, float lr, float beta1, float beta2, int32_t* err,
                            void* stream);

/* Requester: partials[N][B][dim+2] (one slab per owner, summed in rank order) + dense part +
 * bias -> one row per sample of what the other ranks need,
 *   X[b] = [ S (dim) | g = (sigmoid(logit)-y)*gscale | BCE loss row | 0.. ]
 * with ld floats per row (ld % 4 == 0, ld >= dim+2; when ld >= dim+2+n_dense the dense features
 * are stored behind the loss as well).  logit may be NULL.  part_out (may be
 * NULL): mi_shard_dense_part_floats(B, dim, n_dense) floats, one partial per workgroup of this
 * launch of the dense-side sums over its samples (sum g*dense[.][j], sum g, sum loss); placed right
 * behind the B rows of X it travels with them in the all-gather. */
int64_t mi_shard_dense_part_floats(int64_t B, int dim, int n_dense);
int mi_fm_combine(const float* partials, int n_ranks, int64_t B, int dim, const float* dense,
                  int n_dense, const float* wd, const float* bias, const float* labels,
                  float gscale, float* logit, float* X, int ld, float* part_out, void* stream);

/* Owner + replica, ONE launch after the all-gather: the sparse backward with the fused optimizer
 * over sorted_rows / perm / n_total as mi_shard_owner_fwd_sort wrote them (n_slots = their capacity,
 * which sizes the launch), reading g and S from the gathered matrix X_all (rank s owns rows
 * [s*row_stride, (s+1)*row_stride): B sample rows, then the partials of its combine launch); and
 * the dense side: one workgroup adds the partials of all ranks in a fixed order - the same bits on
 * every replica, no all-reduce - applies TensorFlow-form Adam to dp[n_dense+1] = {wd, bias} and
 * writes loss_out[0] = mean loss over the n_ranks*B samples.  scratch:
 * mi_shard_bwd_scratch_floats(n_slots) floats, zeroed once by the caller. */
int64_t mi_shard_bwd_scratch_floats(int64_t n_slots);
int mi_shard_owner_bwd_dense(const mi_fm_table* tab, int64_t local_rows, const uint32_t* sorted_rows,
                             const int32_t* perm, const int32_t* n_total, int64_t n_slots,
                             const float* X_all, int ld, int n_ranks, int64_t B, int64_t row_stride,
                             int opt, float lr, const float* lr_dev, float beta1, float beta2,
                             float eps, int n_dense, float* dp, float* dm, float* dv,
                             float dense_lr_t, const float* dense_lr_dev, float* scratch,
                             float* loss_out, void* stream);

/* mi_fm_dense_step over n rows laid out as mi_fm_combine writes them: every replica computes
 * the dense-side gradients of ALL N*B samples itself (same order on every rank, so replicas
 * stay bit-identical without an all-reduce); loss_out[0] = mean of the n loss rows. */
int mi_fm_dense_step_rows(const float* X, int ld, int dim, int n_dense, int64_t n, float* dp,
                          float* dm, float* dv, float lr_t, const float* lr_dev, float beta1,
                          float beta2, float eps, float* loss_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Row-wise sharded embedding BAGS (any dim that is a multiple of 4, any bag length): the generic
 * form of the above for models that consume the pooled rows themselves (DCN-v2, DIN, plain
 * embedding bags).  Same placement - global row r on rank r % N as local row r / N - and the same
 * reference counterpart (the parameter-server variable traffic, model_trainer.py:124-136).  A step:
 *   all-gather of the ids                              [N][B][T][L]
 *   mi_shard_bag_slots (owner view)    which bags of which rank this rank holds ids of, their rows
 *                                      in the per-destination blocks, the owner's local ids
 *   mi_shard_bag_gather                partial pool of every such bag -> send blocks [N][cap][dim]
 *   all-to-all of the blocks           (fixed capacity `cap` rows per rank pair)
 *   mi_shard_bag_slots (requester view) which owners contribute to each of MY bags, and where
 *   mi_shard_bag_combine               pooled[b][t] = sum of the owners' rows, rank order
 *   ... the model's interaction and loss ...
 *   mi_shard_bag_pack                  gradient rows of my bags -> blocks in the same compact order
 *   all-to-all of the blocks
 *   mi_embag_bwd_prepare on the local ids + mi_embag_bwd_apply_rows: the ordinary sort-by-row
 *                                      update of the owner's rows, gradient rows read from the blocks.
 * With one id per bag a rank receives B*T rows per exchange in all, not N*B*T.  cap is the
 * caller's choice (mean B*T/N rows for one-id bags; twice that covers Zipf(1.05) ids); a pair that
 * needs more raises MI_IDERR_CAPACITY in *err and its surplus bags contribute nothing.
 * ---------------------------------------------------------------------------------- */

/* Slots.  ids: bags of bag_len ids, n_bags = B*T bags per part, bag i of a part belongs to table
 * i % T.  owner >= 0 (owner view): ids holds n_parts parts (one per source rank, as all-gathered)
 * and rank `owner` is asked about; owner < 0 (requester view): ids holds ONE part (this rank's ids)
 * and part p asks which of its bags rank p contributes to (n_parts == n_ranks).  Outputs:
 *   local_ids[n_parts][n_bags][bag_len]  (owner view; may be NULL) the owner's local id, local to its
 *                                 share of the table (see mi_shard_bag_gather), -1 for ids it does not hold
 *   row_of_bag[n_parts][n_bags]   p*cap + rank of the bag among part p's contributing bags, or -1
 *   bag_of_row[n_parts*cap]       the inverse (rows beyond cnt[p] are not meaningful)
 *   cnt[n_parts]                  rows of each block
 * workspace: mi_shard_bag_slots_workspace_bytes(n_bags, n_parts).  Two launches. */
size_t mi_shard_bag_slots_workspace_bytes(int64_t n_bags, int n_parts);
int mi_shard_bag_slots(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int bag_len,
                       int64_t n_bags, int n_parts, int owner, int n_ranks, int64_t cap,
                       int32_t* local_ids, int32_t* row_of_bag, int32_t* bag_of_row, int32_t* cnt,
                       void* workspace, size_t workspace_bytes, int32_t* err, void* stream);

/* Bucketed id exchange (instead of the all-gather of ids): block p of this rank's send buffers is the CSR form of
 * "my bags that rank p contributes to" - send_ids[p][cap_ids] int32, rank p's ids already as ITS local rows
 * (global row / N) packed in bag order, -1 behind the last; send_off[p][cap] int64, the first id of each slot's
 * bag, absolute in the OWNER's receive buffer (this rank's block starts there at rank*cap_ids).  After the two
 * all-to-alls the owner holds ids[N*cap_ids] and off[N*cap] (+ one closing entry N*cap_ids it sets once) and runs
 * the ordinary CSR embedding bag over its local rows: mi_embag_fwd(T = 1, offsets = off, n_bags = N*cap) writes
 * the send blocks of the row exchange, mi_embag_bwd_prepare + mi_embag_bwd_apply_* on the same ids / offsets with
 * the received gradient rows update its rows.  row_of_bag / cnt: the requester view of mi_shard_bag_slots for the
 * same ids.  Bytes on the wire: 4 per id + 8 per (bag, owner) against 8*N per id.  More than cap_ids ids for one
 * owner: MI_IDERR_CAPACITY in *err.  Two launches + one fill. */
size_t mi_shard_bag_bucket_workspace_bytes(int64_t n_bags, int n_ranks);
int mi_shard_bag_bucket(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int bag_len,
                        int64_t n_bags, int rank, int n_ranks, int64_t cap, int64_t cap_ids,
                        const int32_t* row_of_bag, const int32_t* cnt, int32_t* send_ids, int64_t* send_off,
                        void* workspace, size_t workspace_bytes, int32_t* err, void* stream);


/* Owner: out_rows[p*cap + slot][dim] = sum over the local ids >= 0 of bag bag_of_row[..] of part p of
 * weights[local_table_offsets[t] + local id].  weights is this rank's share [local rows][dim];
 * local_table_offsets[T+1] (int64, device): local rows that belong to tables before t, i.e.
 * ceil((table_offsets[t] - rank) / N). */
int mi_shard_bag_gather(const float* weights, const int64_t* local_table_offsets, int T, int dim,
                        const int32_t* local_ids, int bag_len, int64_t n_bags, int n_parts, int64_t cap,
                        const int32_t* bag_of_row, const int32_t* cnt, float* out_rows, void* stream);

/* Requester: out[b][t*dim ..] = sum over parts p (ascending) with row_of_bag[p][b*T+t] >= 0 of
 * rows[that row]; a bag no owner contributes to (all ids padding) gets zeros. */
int mi_shard_bag_combine(const float* rows, const int32_t* row_of_bag, int n_parts, int64_t n_bags, int T,
                         int dim, float* out, int64_t out_ld, void* stream);

/* Requester, backward: rows_out[p*cap + slot] = grad[b][t*dim ..] of bag bag_of_row[p*cap + slot]. */
int mi_shard_bag_pack(const float* grad, int64_t grad_ld, int T, int dim, const int32_t* bag_of_row,
                      const int32_t* cnt, int n_parts, int64_t cap, float* rows_out, void* stream);

/* mi_embag_bwd_apply_sgd / _adagrad_rowwise (opt = MI_OPT_SGD / MI_OPT_ADAGRAD_ROWWISE) with the bag
 * gradients kept as compact rows: the gradient of bag i is grad_rows[row_of_bag[i]] (zero when
 * row_of_bag[i] < 0); sum pooling only. */
int mi_embag_bwd_apply_rows(const mi_bwd_plan* plan, const float* grad_rows, const int32_t* row_of_bag,
                            int dim, int opt, float* weights, float* state, float lr, float eps,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * bf16 MFMA contractions (DCN-v2 cross layers and MLP, DIN activation unit).  Not in the
 * reference, whose dense layers are fp32 tf.layers.dense on a 5,910-parameter network
 * (r_model.py:53-57); these are the dense contractions of the interaction models the benchmark
 * definition adds, and the only work put on the matrix cores.  bf16 operands, fp32 accumulate;
 * agreement with an fp32 evaluation is within 1e-2.
 * ---------------------------------------------------------------------------------- */
#define MI_EPI_SLAB 0      /* out: fp32 [k_splits][M][ld_out] partial products (summed by mi_slab_adam) */
#define MI_EPI_BIAS_ACT 1  /* out (bf16) = act(C + bias[n]); flags & MI_EPI_RELU.  With bias_rows > 0 the
                              bias is a matrix, one row per bias_rows consecutive rows of C:
                              bias[(m / bias_rows) * bias_ld + n] (a term that only depends on the
                              sample when C has one row per (sample, position)) */
#define MI_EPI_CROSS_FWD 2 /* u = C + bias -> out1 (bf16, optional); out (bf16) = in0 * u + in1 */
#define MI_EPI_MASK 3      /* out (bf16) = C where in0 > 0 else 0 (in0 NULL: plain store) */
#define MI_EPI_CROSS_BWD 4 /* v = C + in1 (bf16, optional) -> out (bf16); with out1: out1 (bf16) = v * in0 */
#define MI_EPI_RELU 1

typedef struct mi_gemm_epilogue {
  int mode, flags;
  void* out;
  int64_t ld_out;
  const float* bias;
  const void* in0;
  int64_t ld_in0;
  const void* in1;
  int64_t ld_in1;
  void* out1;
  int64_t ld_out1;
  int64_t bias_ld; /* MI_EPI_BIAS_ACT with a bias matrix: floats between its rows, a multiple of 4 */
  int bias_rows;   /* 0: bias is a vector */
} mi_gemm_epilogue;

/* C[m][n] = sum_k A(m,k)*B(n,k), followed by the epilogue.  An operand is stored either
 * reduction-contiguous (A[m*lda + k], B[n*ldb + k]) or, with its *_reduction_major flag set,
 * reduction-major (A[k*lda + m], B[k*ldb + n]) - the backward contractions need the latter and
 * get it without any transposed copy in HBM.  M, N multiples of 8, K a multiple of 64;
 * k_splits > 1 (MI_EPI_SLAB only) cuts K into that many slices, each a whole number of 64-deep
 * tiles. */
int mi_gemm_bf16(int a_reduction_major, int b_reduction_major, int64_t M, int64_t N, int64_t K,
                 const void* A, int64_t lda, const void* B, int64_t ldb, int k_splits,
                 const mi_gemm_epilogue* epi, void* stream);

/* Memory-bound pieces around the contractions (all sums in a fixed order, no float atomics). */

/* x[b][col0+j] = bf16(dense[b][j]) for j < n_dense, zero up to column d_pad: the dense-feature
 * tail of the interaction input row whose head mi_embag_fwd (out_dtype MI_DT_BF16) writes. */
int mi_pack_dense_bf16(const float* dense, int n_dense, int64_t B, void* x, int64_t ld, int col0,
                       int d_pad, void* stream);

/* partial[s][n] = sum of X[m][n] (bf16) over the rows of slice s (slices of ceil(M/slices)
 * rows): bias gradients, finished by mi_slab_adam. */
int mi_colsum_bf16(const void* X, int64_t ld, int64_t M, int N, int slices, float* partial,
                   int64_t ld_partial, void* stream);

/* g[i] = sum_s slab[s*slab_stride + i] (s ascending), then tf.train.AdamOptimizer's update
 * (model_trainer.py:157; the form of mi_adam_tf_dense) on the fp32 master w with state m, v,
 * and w_bf16[i] = bf16(w[i]) for the next contraction.  grad_out (optional) receives g; with
 * w == NULL only grad_out is written. */
int mi_slab_adam(const float* slab, int n_slabs, int64_t n, int64_t slab_stride, float* w, float* m,
                 float* v, float lr_t, const float* lr_dev, float beta1, float beta2, float eps,
                 void* w_bf16, float* grad_out, void* stream);

/* 1-wide output layer on bf16 activations h[B][K]: logit = h.w + bias, BCE-with-logits loss
 * rows and g = (sigmoid(logit) - y) * gscale. */
int mi_head_fwd(const void* h, int64_t ld, int K, const float* w, const float* bias,
                const float* labels, float gscale, int64_t B, float* logit, float* loss_rows,
                float* g, void* stream);
/* Its backward: dh[b][k] = g[b]*w[k] where h[b][k] > 0 else 0 (h is a ReLU output), and
 * partial[s][0..K) = sum_b g[b]*h[b][k], partial[s][K] = sum_b g[b] over slice s. */
int mi_head_bwd(const float* g, const float* w, const void* h, int64_t ld_h, int K, int64_t B,
                void* dh, int64_t ld_dh, int slices, float* partial, int64_t ld_partial,
                void* stream);

/* out (bf16) = in (fp32). */
int mi_cast_bf16(const float* in, int64_t n, void* out, void* stream);

/* dst[c][r] = src[r][c], bf16, R x C.  A weight kept in both orientations lets the data-gradient
 * contraction dX = dY*W read W row-stored like the forward does (the reduction-major operand path
 * of mi_gemm_bf16 is the slower one); 2 MB per 1024 x 1024 layer and step. */
int mi_transpose_bf16(const void* src, int64_t ld_src, int R, int C, void* dst, int64_t ld_dst,
                      void* stream);

/* Gradient of the cross network's input x0 (x_{l+1} = x0*u_l + x_l, L layers, L <= 8):
 *   out[m][n] (fp32) = dx[0][m][n] + sum_{l < L} dx[l+1][m][n] * u[l][m][n]
 * dx[0..L] and u[0..L-1] are host arrays of device pointers to bf16 [M][ld] tensors (what the
 * MI_EPI_CROSS_BWD / MI_EPI_CROSS_FWD epilogues stored); n < N, N % 4 == 0. */
int mi_cross_input_grad(const void* const* dx, const void* const* u, int L, int64_t M, int N,
                        int64_t ld, float* out, int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------------------
 * DIN local activation unit (not in the reference; its nearest analogue is the 6-id bag with
 * uniform weights, r_model.py:68-80).  A sample has a history of up to L item ids hist[B][L]
 * (len[b] of them live), and a candidate item qid[b]; all ids index ONE table E[rows][dim],
 * dim = 64.  Each live history item k_j gets the weight
 *   a_j = MLP([q, k_j, q-k_j, q*k_j])     4*dim -> h1 -> h2 -> 1, ReLU on the hidden layers
 * and pooled[b] = sum_j a_j k_j (no softmax).  The MLP is run by the caller over all B*L rows
 * of A with mi_gemm_bf16 (two wide layers) and mi_head_fwd (the 1-wide one); its backward with
 * mi_head_bwd / mi_gemm_bf16; these entry points do the memory-bound rest.
 * ---------------------------------------------------------------------------------- */

/* Gather + build: A[B*L][a_cols] bf16 rows, a_cols = 4*dim: [q | k | q-k | q*k], or a_cols = 2*dim:
 * [k | q*k] (the factored unit below) - zero rows for positions j >= len[b], negative ids, and
 * ids >= rows (the latter also set MI_IDERR_RANGE) - and ids_all[B][L+1] int32 = the live history
 * ids (else -1) followed by the candidate id: the index list of the sparse backward.  q_out
 * (optional) receives the fp32 candidate rows. */
int mi_din_lau_build(const float* E, int64_t rows, int dim, const void* hist, const void* qid,
                     int idx_bits, const int32_t* len, int64_t B, int L, void* A, int32_t* ids_all,
                     float* q_out, int64_t ld_q, int a_cols, int32_t* err, void* stream);

/* pooled[b][0..dim) = sum over live j (ascending) of a[b*L+j] * k_j, k_j taken from A. */
int mi_din_lau_pool(const void* A, int a_cols, const float* a, const int32_t* ids_all, int64_t B,
                    int L, int dim, float* pooled, int64_t ld, void* stream);

/* da[b*L+j] = <dpooled[b], k_j> on live positions, 0 elsewhere. */
int mi_din_lau_pool_bwd(const void* A, int a_cols, const float* dpooled, int64_t ld,
                        const int32_t* ids_all, int64_t B, int L, int dim, float* da, void* stream);

/* dA[B*L][4*dim] (bf16) = gradient of the MLP's input rows.  Writes dE[B][L+1][dim] fp32, laid
 * out like ids_all:  d k_j = a_j*dpooled + dA_k - dA_d + q*dA_p;  d q = dq_extra (optional, what
 * the layers above send to q directly) + sum_j (dA_q + dA_d + k_j*dA_p), j ascending. */
int mi_din_lau_scatter(const void* A, const void* dA, const float* a, const float* dpooled,
                       int64_t ld_dp, const float* dq_extra, int64_t ld_dq, const int32_t* ids_all,
                       int64_t B, int L, int dim, float* dE, void* stream);

/* The factored unit.  The first layer is linear in its input [q | k | q-k | q*k], so
 *   A W1^T = q (Wa+Wc)^T + k (Wb-Wc)^T + (q*k) Wd^T          W1 = [Wa | Wb | Wc | Wd]
 * and only [k | q*k] needs a row per (sample, position); the q term is one bias row per SAMPLE
 * (mi_gemm_epilogue.bias_rows = L).  Same mathematics, half the bytes of the largest tensors.
 *
 * mi_din_lau_scatter_fact: A, dA are [B*L][2*dim] = [k | q*k], [gk | gp]; q fp32 [B][ld_q]:
 *   d k_j = a_j*dpooled + gk + q*gp;   d q = dq_extra + sum_j k_j*gp, j ascending.
 * mi_din_qterm_fwd: rowbias[B][H] fp32 = b1 + q Wq^T (Wq [H][dim] = Wa+Wc), the per-sample term.
 * mi_din_qterm_bwd: from S[B][H] = per-sample sums of dZ (mi_colsum_bf16 with one slice per
 *   sample): dq[b] = dq_in[b] (optional) + S[b] Wq, and per 32 samples a partial
 *   part[chunk] = { (q^T S)[dim][H], sum_b S[b] (H) }  (mi_din_qterm_part_floats floats in all):
 *   the gradient of Wa (transposed) and of b1, to be added in chunk order.
 * mi_din_w1_step: TensorFlow-form Adam on the master W1 [H][4*dim] (m, v alike) from
 *   slab[n_slabs][H][2*dim]  partial products of dZ^T [K | Q*K] (added in slab order) -> g_b, g_d
 *   dWaT[n_a][dim][H] (rows a_stride floats apart) partials of q^T S, added in order  -> g_a
 *   g_c = g_a - g_b
 * and the tensors the next step reads: Wf [H][2*dim] bf16 = [Wb-Wc | Wd], WfT [2*dim][H] its
 * transpose, Wq [H][dim] fp32 = Wa+Wc, WqT [dim][H].  grad_out (optional) [H][4*dim]. */
int mi_din_lau_scatter_fact(const void* A, const void* dA, const float* a, const float* dpooled,
                            int64_t ld_dp, const float* dq_extra, int64_t ld_dq, const float* q,
                            int64_t ld_q, const int32_t* ids_all, int64_t B, int L, int dim,
                            float* dE, void* stream);
int64_t mi_din_qterm_part_floats(int64_t B, int H, int dim);
int mi_din_qterm_fwd(const float* q, int64_t ld_q, const float* Wq, const float* b1, int64_t B, int H,
                     int dim, float* rowbias, void* stream);
int mi_din_qterm_bwd(const float* q, int64_t ld_q, const float* S, const float* Wq, int64_t B, int H,
                     int dim, const float* dq_in, int64_t ld_in, float* dq, int64_t ld_dq,
                     float* part, void* stream);
int mi_din_w1_step(const float* slab, int n_slabs, int64_t slab_stride, const float* dWaT, int n_a,
                   int64_t a_stride, int H, int dim, float* W, float* m, float* v, float lr_t,
                   const float* lr_dev,
                   float beta1, float beta2, float eps, void* Wf, void* WfT, float* Wq, float* WqT,
                   float* grad_out, void* stream);

/* out[m][n] = a[m][n] * b[m][n], bf16, n < N (N % 8 == 0): the product a CROSS_BWD epilogue would have stored
 * when the producer of `a` is not one of this library's contractions. */
int mi_mul_bf16(const void* a, int64_t ld_a, const void* b, int64_t ld_b, int64_t M, int N, void* out,
                int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Whole-network entry points: the launch sequences of the two interaction models behind one call each
 * way, for hosts that are not Python.  Not in the reference (its network is r_model.py:52-58); they stand
 * for what a maintainer would register as ONE operator per direction.  Every step inside is one of the
 * entry points above, in the order recommendation_amd/dcn.py and din.py issue them: same bits.
 *
 * A dense parameter carries tf.train.AdamOptimizer's state (model_trainer.py:157): fp32 master w with
 * m, v; matrices also their bf16 working copy [rows][cols] and its transpose [cols][rows], both
 * refreshed by the backward call.  grad_out (optional) receives the gradient.
 * ---------------------------------------------------------------------------------- */
typedef struct mi_adam_param {
  float* w;
  float* m;
  float* v;
  void* w_bf16;   /* matrices only */
  void* w_bf16_t; /* matrices only */
  float* grad_out;
} mi_adam_param;

#define MI_CROSS_MAX_LAYERS 8
/* DCN-v2 cross network: x_{l+1} = x0 * (x_l W_l^T + b_l) + x_l, rows of `width` bf16 values (a multiple of
 * 64, zero padded), W_l [width][width], b_l [width]. */
typedef struct mi_cross_net {
  int n_layers, width;
  mi_adam_param W[MI_CROSS_MAX_LAYERS];
  mi_adam_param b[MI_CROSS_MAX_LAYERS];
} mi_cross_net;

/* Workspace (256-byte aligned) shared by the two calls of a step: x_1..x_L, u_0..u_{L-1} (written by the
 * forward, read by the backward), dx_0..dx_L, two du buffers, fp32 reduction slabs.  mi_cross_workspace_part
 * gives offset and size of one of them, for a caller that feeds or reads them in place:
 * MI_CROSS_PART_X index 1..L, _U 0..L-1, _DX 0..L, _DU 0..1, _SLAB 0. */
#define MI_CROSS_PART_X 0
#define MI_CROSS_PART_U 1
#define MI_CROSS_PART_DX 2
#define MI_CROSS_PART_DU 3
#define MI_CROSS_PART_SLAB 4
size_t mi_cross_workspace_bytes(int64_t B, int width, int n_layers);
int mi_cross_workspace_part(int64_t B, int width, int n_layers, int what, int index, size_t* offset,
                            size_t* bytes);

/* Forward over x0 [B][ld] bf16 (B a multiple of 64): L launches; *x_last = x_L inside the workspace. */
int mi_cross_fwd(const mi_cross_net* net, const void* x0, int64_t ld, int64_t B, void* workspace,
                 size_t workspace_bytes, const void** x_last, void* stream);

/* Backward + Adam step of every W_l, b_l.  The caller has put dL/dx_L (bf16) into workspace part DX[L] -
 * and, with du_last_ready != 0, dL/dx_L * x0 into DU[L & 1] (an MI_EPI_CROSS_BWD epilogue with out1 stores
 * both at once); otherwise the product is formed here.  Data gradients use the weights the forward saw.
 * dx0 [B][ld_dx0] fp32 = dL/dx0, the gradient of the embedding rows and dense features.  lr_t / lr_dev as in
 * mi_slab_adam.  apply_adam == 0: gradients only - every W_l, b_l gradient goes to its grad_out (required) and
 * no parameter changes: the form a data-parallel step needs, which adds the ranks' gradients
 * (mi_allreduce_dense) before it applies them (model_trainer.py:135-158 runs the same graph on every worker). */
int mi_cross_bwd(const mi_cross_net* net, const void* x0, int64_t ld, int64_t B, void* workspace,
                 size_t workspace_bytes, int du_last_ready, int apply_adam, float lr_t, const float* lr_dev,
                 float beta1, float beta2, float eps, float* dx0, int64_t ld_dx0, void* stream);

/* K-slices mi_cross_bwd / mi_din_lau_bwd use for a weight-gradient contraction [n_out][k_in] over m rows
 * (mi_gemm_bf16 with MI_EPI_SLAB): for callers that size their own slabs the same way. */
int mi_gemm_wgrad_splits(int64_t n_out, int64_t k_in, int64_t m);

/* DIN local activation unit (see above): dim = 64; h1, h2 the PADDED hidden widths (multiples of 64), L the
 * history length.  w3 holds w3[h2] followed by b3 (h2 + 4 floats).  factored != 0: the first layer in its
 * factored form (h1 <= 128), whose derived tensors W1f [h1][2*dim] bf16, W1fT, Wq [h1][dim] fp32, WqT the
 * backward call keeps current (mi_din_w1_step); W1 then needs only w, m, v.  Otherwise W1 carries its bf16
 * copies like W2. */
typedef struct mi_din_lau {
  int dim, h1, h2, L, factored;
  mi_adam_param W1, b1, W2, b2, w3;
  void* W1f;
  void* W1fT;
  float* Wq;
  float* WqT;
} mi_din_lau;

#define MI_DIN_PART_A 0       /* [B*L][a_cols] bf16 rows of the unit's input (a_cols = 2*dim factored, else 4*dim) */
#define MI_DIN_PART_DA 1
#define MI_DIN_PART_H1 2
#define MI_DIN_PART_DZ1 3
#define MI_DIN_PART_H2 4
#define MI_DIN_PART_DH2 5
#define MI_DIN_PART_WEIGHTS 6 /* a[B*L] fp32: the activation weights */
size_t mi_din_lau_workspace_bytes(int64_t B, const mi_din_lau* net);
int mi_din_lau_workspace_part(int64_t B, const mi_din_lau* net, int what, size_t* offset, size_t* bytes);

/* Forward: gather + build, the unit's three layers over all B*L rows, weighted pooling.
 * feat [B][ld_feat] fp32 = [pooled | q] (2*dim columns); ids_all [B][L+1] int32 as mi_din_lau_build writes
 * it (the index list of the sparse backward).  B*L a multiple of 64. */
int mi_din_lau_fwd(const mi_din_lau* net, const float* E, int64_t rows, const void* hist, const void* qid,
                   int idx_bits, const int32_t* len, int64_t B, void* workspace, size_t workspace_bytes,
                   float* feat, int64_t ld_feat, int32_t* ids_all, int32_t* err, void* stream);

/* The factored unit's forward as one launch (+ two small ones for the candidates' rows and the first layer's per-sample
 * term; persistent workgroups, one sample at a time; rows, both hidden
 * layers and the head stay in LDS between the layers): what mi_din_lau_fwd runs when mi_din_lau_fwd_fused_ok() - dim 64, hidden widths
 * 128 / 64, L <= 112, factored first layer.  Writes the same A, H1, H2, a, ids_all and feat = [pooled | q] as
 * mi_din_lau_build + mi_din_qterm_fwd + two mi_gemm_bf16 + mi_head_fwd + mi_din_lau_pool, bit for bit.
 * No counterpart in the reference (r_model.py:68-80 is the nearest: a 6-id bag with uniform weights).
 * W1f [128][128] bf16 (columns for k, then q*k), Wq [128][64], b1 [128], rowbias [B][128] (scratch: b1 + Wq q), W2 [64][128] bf16, b2 [64], w3 [65]. */
int mi_din_lau_fwd_fused_ok(int dim, int h1, int h2, int L, int factored);
int mi_din_lau_fwd_fused(const float* E, int64_t rows, const void* hist, const void* qid, int idx_bits,
                         const int32_t* len, int64_t B, int L, const void* W1f, const float* Wq, const float* b1,
                         const void* W2, const float* b2, const float* w3, void* A, void* H1, void* H2, float* a,
                         int32_t* ids_all, float* feat, int64_t ld_feat, float* rowbias, int32_t* err, void* stream);

/* The data path of the factored unit's backward as one launch per sample (same conditions as the fused forward):
 * from dfeat [B][ld_dfeat] = dL/d[pooled | q] and what the forward left (A, H1, H2, a, ids_all, feat) to
 *   dZ2 [B*L][64], dZ1 [B*L][128] bf16 (what the weight-gradient contractions read),
 *   dE [B*(L+1)][64] fp32: rows of the history positions complete; the candidate's row = dq_above + sum_j k_j * gp,
 *     to which the caller adds S Wq (mi_din_qterm_bwd with dq_in = dq = that row),
 *   S [B][128] = per-sample column sums of dZ1,
 *   part [B][mi_din_lau_bwd_part_floats()] scratch and part_sums [slices][same]: partial sums over runs of samples of
 *     [dL/dw3 (64) | dL/d(bias3) | 0 0 0 | dL/db2 (64)] (slices must cut B into equal runs: ceil(B / ceil(B / slices)) == slices).
 * dZ2, dZ1 and the history rows of dE are bit-identical to mi_din_lau_pool_bwd + mi_head_bwd + the two masked
 * mi_gemm_bf16 + mi_din_lau_scatter_fact; the sums are added in another order.  No counterpart in the reference. */
int mi_din_lau_bwd_part_floats(void);
int mi_din_lau_bwd_fused(int64_t B, int L, const void* A, const void* H1, const void* H2, const float* a,
                         const int32_t* ids_all, const float* feat, int64_t ld_feat, const float* dfeat, int64_t ld_dfeat,
                         const float* w3, const void* W2T, const void* W1fT, void* dZ2, void* dZ1, float* dE, float* S,
                         float* part, int slices, float* part_sums, void* stream);

/* Backward + Adam step of the unit's parameters from dfeat [B][ld_dfeat] = dL/d[pooled | q]; the workspace
 * and feat / ids_all as the forward of the same batch left them.  dE [B*(L+1)][dim] fp32: gradient of the
 * gathered rows, laid out like ids_all (input of the sparse update). */
int mi_din_lau_bwd(const mi_din_lau* net, int64_t B, void* workspace, size_t workspace_bytes,
                   const float* feat, int64_t ld_feat, const float* dfeat, int64_t ld_dfeat,
                   const int32_t* ids_all, float lr_t, const float* lr_dev, float beta1, float beta2,
                   float eps, float* dE, void* stream);

/* BCE-with-logits on fp32 logits: loss rows and g = (sigmoid(z) - y) * gscale (either may be NULL). */
int mi_bce_logits(const float* logits, const float* labels, int64_t B, float gscale,
                  float* loss_rows, float* g, void* stream);

/* ------------------------------------------------------------------------------------
 * Host-side tokenizer of the record wire format (no GPU work; all pointers are HOST memory).
 * A record is one line "user country l1 click1 l2 click2 [label]" with space-separated fields,
 * l1 / l2 comma-separated id lists of bag_len entries: r_model.py:143-177 (Sample.read_csv /
 * read_label_data: tf.decode_csv(field_delim=' ') with record_defaults [0, 0, '0,0,0,0,0,0', 0,
 * '0,0,0,0,0,0', 0(, 0)], tf.string_split(sep=',') + to_number on the two list fields).
 * Parses at most max_rows records from text[0..len):
 *   scalars[row][n_fields-2]  the int32 scalar fields in record order
 *   bags[row][2][bag_len]     the two lists
 * n_rows = records parsed, consumed = bytes of text they (and skipped blank lines) took.  Any
 * malformed record (field count, non-int32 value, list length != bag_len) is an error, as it is
 * for decode_csv; mi_last_error() names the line. */
int mi_parse_records(const char* text, int64_t len, int n_fields, int bag_len, int64_t max_rows,
                     int32_t* scalars, int32_t* bags, int64_t* n_rows, int64_t* consumed);

#ifdef __cplusplus
}
#endif
#endif /* MI355REC_H */
