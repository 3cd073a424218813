// Factorisation-machine forward fused with the embedding gather, and the dense side of the
// step.  The reference has no FM (its model is the linear colour network of
// recommendation/code/r_model.py:31-40); this is the extension the benchmark definition
// names, built on the same row gather as the colour model's first layers.
//
//   logit[b] = bias + sum_f w1[row(b,f)] + sum_j wd[j]*dense[b][j]
//              + 0.5 * sum_k ( S[b][k]^2 - sum_f V[row(b,f)][k]^2 ),  S[b][k] = sum_f V[row(b,f)][k]
//
// The pairwise term is formed in registers while the rows stream in: every lane keeps a
// running sum and a running sum of squares of the 4 columns it owns, so a row is read once
// and nothing but S[b] (needed by backward) and the logit is written.
//
// HBM layout: one record per row (mi_fm_table): V[dim] then w1 in the same 128-byte line, so
// a (sample, field) pair costs one line, not one line for V plus one for a 4-byte w1.
//
// Lane mapping (G lanes per sample, G = 16 for dim <= 64): lane l owns column chunk
// c = l % (dim/4) and walks fields q, q+FP, ... with q = l / (dim/4), FP = G / (dim/4).
// For dim 16 that is 4 lanes x 16 B = one 64 B row per field, 4 fields in flight per
// sample per step, 6-7 independent row loads per lane.
#include "common.h"

namespace {

}  // namespace
#include "fm_body.h"
#include "table_sort.h"
namespace {

template <typename IDX, int DIM>
__global__ __launch_bounds__(256) void fm_fwd_kernel(FmFwdArgs a, float* __restrict__ sched, float lr, float b1,
                                                    float b2) {
  if (sched && blockIdx.x == 0 && threadIdx.x == 0) adam_schedule_body(sched, lr, b1, b2);
  fm_fwd_body<IDX, DIM>(a, blockIdx.x, 256);
}

// One launch for everything of a step that needs only its inputs: fwd_blocks workgroups
// run the forward, the others sort the tables' (row, position) pairs in LDS (up to `parts`
// per table), and one thread advances the optimizer schedule (its consumers run in later launches).
// The two halves do not depend on each other, so the launch lasts as long as the longer one instead
// of their sum.
template <typename IDX, int DIM, bool SORT_FIRST>
__global__ __launch_bounds__(TS_THREADS) void fm_fwd_sort_kernel(FmFwdArgs a, int fwd_blocks, int parts,
                                                                uint32_t sentinel, uint32_t* __restrict__ sorted_rows,
                                                                uint32_t* __restrict__ perm, float* __restrict__ sched,
                                                                float lr, float b1, float b2) {
  // Beyond 8192 samples the launch holds more workgroups than the chip takes at once (a thousand forward workgroups
  // of ~8 us and four hundred sort workgroups of ~30 us) and the sort workgroups come first in the grid, so that the
  // long ones are not the last to start (batch 65536: the launch 100 -> 99 us with uniform ids, 98 -> 91 us with
  // Zipf ids).  Up to 8192 samples every workgroup of the launch is resident at once and the forward comes first.
  // (The order is fixed at compile time: as a run-time choice it cost the 8192-sample launch 0.4 us.)
  if constexpr (SORT_FIRST) {
    const int sort_blocks = (int) gridDim.x - fwd_blocks;
    if ((int) blockIdx.x < sort_blocks) {
      int t, part, np;
      if (!ts_assign(a.toffs, a.T, a.B, parts, (int) blockIdx.x, t, part, np)) return;
      table_sort_body<IDX>(t, part, np, a.toffs, a.T, a.ids, 1, a.B, sentinel, sorted_rows, perm, a.ids_tm != 0);
      return;
    }
    const int fb = (int) blockIdx.x - sort_blocks;
    if (sched && fb == 0 && threadIdx.x == 0) adam_schedule_body(sched, lr, b1, b2);
    fm_fwd_body<IDX, DIM>(a, fb, TS_THREADS);
  } else {
    if ((int) blockIdx.x >= fwd_blocks) {
      int t, part, np;
      if (!ts_assign(a.toffs, a.T, a.B, parts, (int) blockIdx.x - fwd_blocks, t, part, np)) return;
      table_sort_body<IDX>(t, part, np, a.toffs, a.T, a.ids, 1, a.B, sentinel, sorted_rows, perm, a.ids_tm != 0);
      return;
    }
    if (sched && blockIdx.x == 0 && threadIdx.x == 0) adam_schedule_body(sched, lr, b1, b2);
    fm_fwd_body<IDX, DIM>(a, blockIdx.x, TS_THREADS);
  }
}

// The per-table sort on its own, for ids in either layout.
template <typename IDX>
__global__ __launch_bounds__(TS_THREADS) void fm_sort_kernel(const int64_t* __restrict__ toffs, int T,
                                                            const void* __restrict__ ids, int B, int parts,
                                                            uint32_t sentinel, uint32_t* __restrict__ sorted_rows,
                                                            uint32_t* __restrict__ perm, int ids_tm) {
  int t, part, np;
  if (!ts_assign(toffs, T, B, parts, (int) blockIdx.x, t, part, np)) return;
  table_sort_body<IDX>(t, part, np, toffs, T, ids, 1, B, sentinel, sorted_rows, perm, ids_tm != 0);
}

__global__ __launch_bounds__(FM_DENSE_THREADS) void fm_dense_kernel(FmDenseArgs a) {
  __shared__ float sh[FM_DENSE_THREADS];
  fm_dense_body(a, blockIdx.x, sh);
}

__global__ void adam_schedule_kernel(float* __restrict__ sched, float lr, float b1, float b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  adam_schedule_body(sched, lr, b1, b2);
}

// sorted_rows == NULL: the forward alone (256-thread workgroups; + optimizer schedule when sched != NULL).
// Otherwise the fused forward + per-table sort (+ optimizer schedule when sched != NULL) launch.
template <typename IDX>
int launch_fm(const FmFwdArgs& a, uint32_t sentinel, uint32_t* sorted_rows, uint32_t* perm, float* sched, float lr,
              float b1, float b2, hipStream_t s) {
#define FM_CASE(D)                                                                                                  \
  case D: {                                                                                                         \
    constexpr int G_ = (D / 4) > 16 ? (D / 4) : 16;                                                                 \
    if (!sorted_rows)                                                                                               \
      hipLaunchKernelGGL((fm_fwd_kernel<IDX, D>), dim3((unsigned) mi_cdiv(a.B, 256 / G_)), dim3(256), 0, s, a,      \
                         sched, lr, b1, b2);                                                                        \
    else {                                                                                                          \
      const int fwd_blocks_ = (int) mi_cdiv(a.B, TS_THREADS / G_);                                                  \
      const int parts_ = mi_table_sort_parts(a.T, a.B, fwd_blocks_);                                                \
      const dim3 grid_((unsigned) (mi_table_sort_blocks(a.T, a.B, parts_) + fwd_blocks_));                          \
      if (a.B > TS_MAX)                                                                                             \
        hipLaunchKernelGGL((fm_fwd_sort_kernel<IDX, D, true>), grid_, dim3(TS_THREADS), 0, s, a, fwd_blocks_,       \
                           parts_, sentinel, sorted_rows, perm, sched, lr, b1, b2);                                 \
      else                                                                                                          \
        hipLaunchKernelGGL((fm_fwd_sort_kernel<IDX, D, false>), grid_, dim3(TS_THREADS), 0, s, a, fwd_blocks_,      \
                           parts_, sentinel, sorted_rows, perm, sched, lr, b1, b2);                                 \
    }                                                                                                               \
    break;                                                                                                          \
  }
  switch (a.tab.dim) {
    FM_CASE(4)
    FM_CASE(8)
    FM_CASE(16)
    FM_CASE(32)
    FM_CASE(64)
    FM_CASE(128)
    default:
      mi_set_error("mi_fm_fwd: dim %d not built (4, 8, 16, 32, 64, 128)", a.tab.dim);
      return MI_ERR_UNSUPPORTED;
  }
#undef FM_CASE
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int check_fwd(const char* who, const mi_fm_table* tab, const int64_t* table_offsets, int T, const void* ids,
              int idx_bits, int64_t B, const float* dense, int n_dense, const float* wd, const float* S) {
  const int rc = mi_fm_check_table(tab, MI_OPT_SGD, who);
  if (rc) return rc;
  MI_REQUIRE(table_offsets && ids, "%s: null pointer", who);
  MI_REQUIRE(T >= 1 && B >= 0 && B < ((int64_t) 1 << 31) / T, "%s: bad T/B", who);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "%s: idx_bits must be 32 or 64", who);
  MI_REQUIRE(n_dense == 0 || (dense && wd), "%s: dense features need dense and wd", who);
  MI_REQUIRE(((uintptr_t) S & 15) == 0, "%s: S must be 16-byte aligned", who);
  return MI_OK;
}

FmDenseArgs dense_args(const float* g, int64_t g_ld, const float* dense, int64_t dense_ld, int n_dense, int64_t B,
                       float* grad_out, const float* loss_rows, int64_t loss_ld, float* dp, float* dm, float* dv,
                       float lr_t, const float* lr_dev, float b1, float b2, float eps, float* loss_out) {
  FmDenseArgs a;
  a.g = g; a.g_ld = g_ld; a.dense = dense; a.dense_ld = dense_ld; a.n_dense = n_dense; a.B = B;
  a.grad_out = grad_out; a.loss_rows = loss_rows; a.loss_ld = loss_ld; a.dp = dp; a.dm = dm; a.dv = dv;
  a.lr_t = lr_t; a.lr_dev = lr_dev; a.b1 = b1; a.b2 = b2; a.eps = eps; a.loss_out = loss_out;
  return a;
}

}  // namespace

int mi_fm_check_table(const mi_fm_table* t, int opt, const char* who) {
  MI_REQUIRE(t && t->rec, "%s: null table", who);
  MI_REQUIRE(((uintptr_t) t->rec & 15) == 0 && t->stride % 4 == 0, "%s: records must be 16-byte aligned, stride %% 4 == 0",
             who);
  MI_REQUIRE(t->dim >= 4 && t->dim % 4 == 0, "%s: dim %d must be a multiple of 4", who, t->dim);
  MI_REQUIRE(t->off_w1 >= t->dim && t->off_w1 < t->stride, "%s: off_w1 %d outside [dim, stride)", who, t->off_w1);
  auto scalar_ok = [&](int off) { return off >= t->dim && off < t->stride && off != t->off_w1; };
  auto vector_ok = [&](int off) { return off >= t->dim && off % 4 == 0 && off + t->dim <= t->stride &&
                                         !(t->off_w1 >= off && t->off_w1 < off + t->dim); };
  if (opt == MI_OPT_ADAGRAD_ROWWISE) {
    MI_REQUIRE(scalar_ok(t->off_vs0) && scalar_ok(t->off_ws0) && t->off_vs0 != t->off_ws0,
               "%s: Adagrad accumulators (off_vs0=%d, off_ws0=%d) must be distinct floats in [dim, stride)", who,
               t->off_vs0, t->off_ws0);
  } else if (opt == MI_OPT_ADAM_TF) {
    MI_REQUIRE(vector_ok(t->off_vs0) && vector_ok(t->off_vs1) &&
                   (t->off_vs0 + t->dim <= t->off_vs1 || t->off_vs1 + t->dim <= t->off_vs0),
               "%s: Adam m/v blocks (off_vs0=%d, off_vs1=%d) must be disjoint, 4-aligned, inside the record", who,
               t->off_vs0, t->off_vs1);
    auto outside = [&](int off) { return scalar_ok(off) && !(off >= t->off_vs0 && off < t->off_vs0 + t->dim) &&
                                         !(off >= t->off_vs1 && off < t->off_vs1 + t->dim); };
    MI_REQUIRE(outside(t->off_ws0) && outside(t->off_ws1) && t->off_ws0 != t->off_ws1,
               "%s: w1 and its Adam state must be distinct floats outside V, m and v", who);
  } else {
    MI_REQUIRE(opt == MI_OPT_SGD, "%s: unknown optimizer %d", who, opt);
  }
  return MI_OK;
}

extern "C" int mi_fm_fwd(const mi_fm_table* tab, const int64_t* table_offsets, int T, const void* ids, int idx_bits,
                         int64_t B, const float* dense, int n_dense, const float* wd, const float* bias,
                         const float* labels, float gscale, float* logit, float* S, float* loss_rows, float* g,
                         int32_t* err, void* stream) {
  if (B == 0) return MI_OK;
  const int rc = check_fwd("mi_fm_fwd", tab, table_offsets, T, ids, idx_bits, B, dense, n_dense, wd, S);
  if (rc) return rc;
  const FmFwdArgs a{*tab, table_offsets, T, ids, (int) B, dense, n_dense, wd, bias, labels, gscale, logit, S, loss_rows,
                    g, err, 0};
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32) return launch_fm<int32_t>(a, 0u, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, s);
  return launch_fm<int64_t>(a, 0u, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, s);
}

extern "C" int mi_fm_fwd_sort_max_batch(void) { return TS_BIG_MAX; }
extern "C" int mi_fm_lds_sort_single_max(void) { return TS_MAX; }

extern "C" int mi_fm_fwd_sort(const mi_fm_table* tab, const int64_t* table_offsets, int T, const void* ids,
                              int idx_bits, int64_t B, const float* dense, int n_dense, const float* wd,
                              const float* bias, const float* labels, float gscale, float* logit, float* S,
                              float* loss_rows, float* g, int32_t* err, int64_t total_rows, uint32_t* sorted_rows,
                              int32_t* perm, float* sched, float lr, float beta1, float beta2, int ids_table_major,
                              void* stream) {
  const char* who = "mi_fm_fwd_sort";
  const int rc = check_fwd(who, tab, table_offsets, T, ids, idx_bits, B, dense, n_dense, wd, S);
  if (rc) return rc;
  MI_REQUIRE((sorted_rows != nullptr) == (perm != nullptr), "%s: sorted_rows and perm go together", who);
  if (sorted_rows && (B < 1 || B > TS_BIG_MAX || (B > TS_MAX && !ids_table_major))) {
    mi_set_error("%s: batch %lld outside 1..%d (%d unless the ids are table-major: a large table's workgroups each walk "
                 "all its ids); use mi_fm_fwd + mi_embag_bwd_prepare", who, (long long) B, TS_BIG_MAX, TS_MAX);
    return MI_ERR_UNSUPPORTED;
  }
  if (B == 0) return MI_OK;
  if (sorted_rows && (total_rows < 1 || total_rows >= (int64_t) 0xFFFFFFFFll)) {
    mi_set_error("%s: total_rows %lld does not fit the uint32 row key", who, (long long) total_rows);
    return MI_ERR_UNSUPPORTED;
  }
  const FmFwdArgs a{*tab, table_offsets, T, ids, (int) B, dense, n_dense, wd, bias, labels, gscale, logit, S, loss_rows,
                    g, err, ids_table_major ? 1 : 0};
  hipStream_t s = (hipStream_t) stream;
  uint32_t* pv = reinterpret_cast<uint32_t*>(perm);
  if (idx_bits == 32) return launch_fm<int32_t>(a, (uint32_t) total_rows, sorted_rows, pv, sched, lr, beta1, beta2, s);
  return launch_fm<int64_t>(a, (uint32_t) total_rows, sorted_rows, pv, sched, lr, beta1, beta2, s);
}

extern "C" int mi_fm_sort_ids(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int64_t B,
                              int ids_table_major, int64_t total_rows, uint32_t* sorted_rows, int32_t* perm,
                              void* stream) {
  const char* who = "mi_fm_sort_ids";
  MI_REQUIRE(table_offsets && ids && sorted_rows && perm, "%s: null pointer", who);
  MI_REQUIRE(T >= 1 && (idx_bits == 32 || idx_bits == 64), "%s: bad T / idx_bits", who);
  if (B < 1 || B > TS_BIG_MAX || (B > TS_MAX && !ids_table_major)) {
    mi_set_error("%s: batch %lld outside 1..%d (%d unless the ids are table-major: a large table's workgroups each walk "
                 "all its ids); use mi_embag_bwd_prepare", who, (long long) B, TS_BIG_MAX, TS_MAX);
    return MI_ERR_UNSUPPORTED;
  }
  if (total_rows < 1 || total_rows >= (int64_t) 0xFFFFFFFFll) {
    mi_set_error("%s: total_rows %lld does not fit the uint32 row key", who, (long long) total_rows);
    return MI_ERR_UNSUPPORTED;
  }
  const int parts = mi_table_sort_parts(T, (int) B, 0);
  const unsigned grid = (unsigned) mi_table_sort_blocks(T, (int) B, parts);
  uint32_t* pv = reinterpret_cast<uint32_t*>(perm);
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    hipLaunchKernelGGL(fm_sort_kernel<int32_t>, dim3(grid), dim3(TS_THREADS), 0, s, table_offsets, T, ids, (int) B,
                       parts, (uint32_t) total_rows, sorted_rows, pv, ids_table_major ? 1 : 0);
  else
    hipLaunchKernelGGL(fm_sort_kernel<int64_t>, dim3(grid), dim3(TS_THREADS), 0, s, table_offsets, T, ids, (int) B,
                       parts, (uint32_t) total_rows, sorted_rows, pv, ids_table_major ? 1 : 0);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_adam_schedule_step(float* sched, float lr, float beta1, float beta2, void* stream) {
  MI_REQUIRE(sched, "mi_adam_schedule_step: null sched");
  hipLaunchKernelGGL(adam_schedule_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, sched, lr, beta1, beta2);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fm_bwd_dense(const float* g, const float* dense, int n_dense, int64_t B, float* out,
                               void* stream) {
  MI_REQUIRE(g && out && n_dense >= 0 && B >= 0 && (n_dense == 0 || dense), "mi_fm_bwd_dense: bad arguments");
  hipLaunchKernelGGL(fm_dense_kernel, dim3(n_dense + 1), dim3(FM_DENSE_THREADS), 0, (hipStream_t) stream,
                     dense_args(g, 1, dense, n_dense, n_dense, B, out, nullptr, 1, nullptr, nullptr, nullptr, 0.f,
                                nullptr, 0.f, 0.f, 0.f, nullptr));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fm_dense_step(const float* g, const float* dense, int n_dense, int64_t B, const float* loss_rows,
                                float* dp, float* dm, float* dv, float lr_t, const float* lr_dev, float beta1,
                                float beta2, float eps, float* loss_out, void* stream) {
  MI_REQUIRE(g && dp && dm && dv && n_dense >= 0 && B >= 1 && (n_dense == 0 || dense), "mi_fm_dense_step: bad arguments");
  MI_REQUIRE((loss_rows == nullptr) == (loss_out == nullptr), "mi_fm_dense_step: loss_rows and loss_out go together");
  hipLaunchKernelGGL(fm_dense_kernel, dim3(n_dense + 1 + (loss_rows ? 1 : 0)), dim3(FM_DENSE_THREADS), 0,
                     (hipStream_t) stream,
                     dense_args(g, 1, dense, n_dense, n_dense, B, nullptr, loss_rows, 1, dp, dm, dv, lr_t, lr_dev, beta1,
                                beta2, eps, loss_out));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fm_dense_step_rows(const float* X, int ld, int dim, int n_dense, int64_t n, float* dp, float* dm,
                                     float* dv, float lr_t, const float* lr_dev, float beta1, float beta2, float eps,
                                     float* loss_out, void* stream) {
  MI_REQUIRE(X && dp && dm && dv && loss_out && n_dense >= 0 && n >= 1 && dim >= 1 && ld >= dim + 2 + n_dense,
             "mi_fm_dense_step_rows: bad arguments");
  hipLaunchKernelGGL(fm_dense_kernel, dim3(n_dense + 2), dim3(FM_DENSE_THREADS), 0, (hipStream_t) stream,
                     dense_args(X + dim, ld, X + dim + 2, ld, n_dense, n, nullptr, X + dim + 1, ld, dp, dm, dv, lr_t,
                                lr_dev, beta1, beta2, eps, loss_out));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

#ifdef TS_TRACE
extern "C" int mi_debug_ts_trace(unsigned long long* host_out, int n) {
  return (int) hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ts_trace_buf), (size_t) n * 8);
}
#endif
