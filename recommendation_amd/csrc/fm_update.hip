// Sparse backward, step 2, for FM row records (single GPU and the owner side of row-wise sharded tables): the
// entry points over the kernels of sparse_update.h, in a translation unit of their own so that the two
// families of template instantiations compile side by side.
#include <stdlib.h>

#include "sparse_update.h"

#ifdef SEG_TRACE
extern "C" int mi_debug_seg_trace(unsigned long long* host_out, int n) {
  return (int) hipMemcpyFromSymbol(host_out, HIP_SYMBOL(seg_trace_buf), (size_t) n * 8);
}
extern "C" int mi_debug_seg_phases(unsigned long long* host_out, int n) {
  return (int) hipMemcpyFromSymbol(host_out, HIP_SYMBOL(seg_phase_buf), (size_t) n * 8);
}
#endif

template <class Extra>
static int fm_bwd_apply_impl(const char* who, const mi_bwd_plan* plan, const mi_fm_table* tab, const float* g,
                             int g_ld, const float* S, int S_ld, const int32_t* entry_sample, int opt, float lr,
                             const float* lr_dev, float beta1, float beta2, float eps, void* stream,
                             const Extra& extra, const int32_t* n_dev = nullptr) {
  PlanDev pl;
  int rc = check_plan(plan, &pl, false, who);
  if (rc) return rc;
  pl.n_dev = n_dev;
  MI_REQUIRE(g && S, "%s: null pointer", who);
  rc = mi_fm_check_table(tab, opt, who);
  if (rc) return rc;
  MI_REQUIRE(!plan->offsets && plan->bag_len == 1 && plan->pool_mode == MI_POOL_SUM,
             "%s: FM takes one id per field (bag_len 1, sum)", who);
  MI_REQUIRE(al16(S) && S_ld % 4 == 0 && S_ld >= tab->dim && g_ld >= 1,
             "%s: S rows must be 16-byte aligned and at least dim floats apart", who);
  Geometry geo;
  rc = geometry(tab->dim, true, &geo);
  if (rc) return rc;
  MI_REQUIRE(!entry_sample, "%s: entry_sample is no longer taken; sort with T = 1 and the sample as payload", who);
  FmSrc src{g, S, g_ld, S_ld};
  OptArgs a{lr, beta1, beta2, eps, lr_dev};
  hipStream_t s = (hipStream_t) stream;
  // records in the standard dim-16 layout: the updater that reads and writes whole 64-byte halves of lines
  static const bool pack_off = getenv("MI355REC_FM_NO_PACK") != nullptr;
  const bool pack = !pack_off && geo.vec == 4 && geo.lanes == 4 && fm_pack_layout_ok(*tab, opt);
  switch (opt) {
    case MI_OPT_SGD: {
      if (pack) return run_segments(plan, pl, src, FmPackUpd<MI_OPT_SGD>{*tab, a}, geo, s, extra);
      FmUpd<MI_OPT_SGD> upd{*tab, a};
      return run_segments(plan, pl, src, upd, geo, s, extra);
    }
    case MI_OPT_ADAGRAD_ROWWISE: {
      if (!(pow2(geo.lanes) && geo.lanes <= 64)) {
        mi_set_error("%s: row-wise Adagrad needs dim/4 to be a power of two <= 64", who);
        return MI_ERR_UNSUPPORTED;
      }
      if (pack) return run_segments(plan, pl, src, FmPackUpd<MI_OPT_ADAGRAD_ROWWISE>{*tab, a}, geo, s, extra);
      FmUpd<MI_OPT_ADAGRAD_ROWWISE> upd{*tab, a};
      return run_segments(plan, pl, src, upd, geo, s, extra);
    }
    case MI_OPT_ADAM_TF: {
      if (pack) return run_segments(plan, pl, src, FmPackUpd<MI_OPT_ADAM_TF>{*tab, a}, geo, s, extra);
      FmUpd<MI_OPT_ADAM_TF> upd{*tab, a};
      return run_segments(plan, pl, src, upd, geo, s, extra);
    }
    default:
      mi_set_error("%s: unknown optimizer %d", who, opt);
      return MI_ERR_INVALID_ARG;
  }
}

extern "C" int mi_fm_bwd_apply(const mi_bwd_plan* plan, const mi_fm_table* tab, const float* g, const float* S,
                               const int32_t* entry_sample, int opt, float lr, const float* lr_dev, float beta1,
                               float beta2, float eps, void* stream) {
  MI_REQUIRE(tab, "mi_fm_bwd_apply: null table");
  return fm_bwd_apply_impl("mi_fm_bwd_apply", plan, tab, g, 1, S, tab->dim, entry_sample, opt, lr, lr_dev, beta1,
                           beta2, eps, stream, NoExtra{0});
}

extern "C" int64_t mi_shard_bwd_scratch_floats(int64_t n_slots) { return n_slots + 2; }

extern "C" int mi_shard_owner_bwd_dense(const mi_fm_table* tab, int64_t local_rows, const uint32_t* sorted_rows,
                                        const int32_t* perm, const int32_t* n_total, int64_t n_slots, const float* X,
                                        int ld, int n_ranks, int64_t B, int64_t row_stride, int opt, float lr,
                                        const float* lr_dev, float beta1, float beta2, float eps, int n_dense,
                                        float* dp, float* dm, float* dv, float dense_lr_t, const float* dense_lr_dev,
                                        float* scratch, float* loss_out, void* stream) {
  const char* who = "mi_shard_owner_bwd_dense";
  MI_REQUIRE(tab && sorted_rows && perm && n_total && X && dp && dm && dv && scratch && loss_out, "%s: null pointer",
             who);
  MI_REQUIRE(n_slots >= 0 && n_slots < ((int64_t) 1 << 31) && n_ranks >= 1 && B >= 1 && n_dense >= 0 &&
                 n_dense + 2 <= FM_DENSE_THREADS && ld >= tab->dim + 2,
             "%s: bad n_slots/n_ranks/B/n_dense/ld", who);
  MI_REQUIRE(local_rows >= 1 && local_rows < (int64_t) 0x7FFFFFFFll, "%s: local_rows out of range", who);
  const int64_t part = mi_shard_dense_part_floats(B, tab->dim, n_dense);
  MI_REQUIRE((row_stride - B) * ld >= part, "%s: row_stride leaves no room for the combine launch's partials", who);
  mi_bwd_plan plan = {};
  plan.sorted_rows = sorted_rows;
  plan.perm = perm;
  plan.long_list = reinterpret_cast<int32_t*>(scratch);
  plan.T = 1;  // entry p's sample is p itself: the sort's payload is the row of X
  plan.n_entries = n_slots;
  plan.n_bags = n_slots;
  plan.bag_len = 1;
  plan.total_rows = local_rows;
  plan.pool_mode = MI_POOL_SUM;
  FmDenseFinalExtra extra;
  extra.blocks = 1;
  FmDenseFinalArgs& a = extra.a;
  a.ct = 1;
  while (a.ct < n_dense + 2) a.ct <<= 1;
  a.n_blocks = (int) (part / a.ct);
  a.X = X; a.ld = ld; a.row_stride = row_stride; a.B = B; a.n_ranks = n_ranks; a.n_dense = n_dense;
  a.dp = dp; a.dm = dm; a.dv = dv; a.lr_t = dense_lr_t; a.lr_dev = dense_lr_dev;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.loss_out = loss_out;
  return fm_bwd_apply_impl(who, &plan, tab, X + tab->dim, ld, X, ld, nullptr, opt, lr, lr_dev, beta1, beta2, eps,
                           stream, extra, n_total);
}

extern "C" int mi_fm_bwd_apply_dense(const mi_bwd_plan* plan, const mi_fm_table* tab, const float* g, const float* S,
                                     int opt, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                                     const float* dense, int n_dense, const float* loss_rows, float* dp, float* dm,
                                     float* dv, float dense_lr_t, const float* dense_lr_dev, float* loss_out,
                                     void* stream) {
  const char* who = "mi_fm_bwd_apply_dense";
  MI_REQUIRE(plan && tab, "%s: null pointer", who);
  MI_REQUIRE(g && dp && dm && dv && n_dense >= 0 && (n_dense == 0 || dense), "%s: bad dense-side arguments", who);
  MI_REQUIRE((loss_rows == nullptr) == (loss_out == nullptr), "%s: loss_rows and loss_out go together", who);
  const int64_t B = plan->n_bags / (plan->T > 0 ? plan->T : 1);
  MI_REQUIRE(B >= 1, "%s: empty batch", who);
  if (plan->unique_rows) {  // plans with run arrays take the two-launch path
    const int rc = mi_fm_bwd_apply(plan, tab, g, S, nullptr, opt, lr, lr_dev, beta1, beta2, eps, stream);
    if (rc) return rc;
    return mi_fm_dense_step(g, dense, n_dense, B, loss_rows, dp, dm, dv, dense_lr_t, dense_lr_dev, beta1, beta2, eps,
                            loss_out, stream);
  }
  FmDenseExtra extra;
  extra.blocks = n_dense + 1 + (loss_rows ? 1 : 0);
  extra.a.g = g; extra.a.g_ld = 1; extra.a.dense = dense; extra.a.dense_ld = n_dense; extra.a.n_dense = n_dense;
  extra.a.B = B; extra.a.grad_out = nullptr; extra.a.loss_rows = loss_rows; extra.a.loss_ld = 1;
  extra.a.dp = dp; extra.a.dm = dm; extra.a.dv = dv; extra.a.lr_t = dense_lr_t; extra.a.lr_dev = dense_lr_dev;
  extra.a.b1 = beta1; extra.a.b2 = beta2; extra.a.eps = eps; extra.a.loss_out = loss_out;
  return fm_bwd_apply_impl(who, plan, tab, g, 1, S, tab->dim, nullptr, opt, lr, lr_dev, beta1, beta2, eps, stream,
                           extra);
}

extern "C" int mi_fm_bwd_apply_dense_stage(const mi_bwd_plan* plan, const mi_fm_table* tab, const float* g,
                                           const float* S, int opt, float lr, const float* lr_dev, float beta1,
                                           float beta2, float eps, const float* dense, int n_dense,
                                           const float* loss_rows, float* dp, float* dm, float* dv, float dense_lr_t,
                                           const float* dense_lr_dev, float* loss_out, void* stage_dst,
                                           const void* stage_src_host, size_t stage_bytes, void* stream) {
  const char* who = "mi_fm_bwd_apply_dense_stage";
  MI_REQUIRE(plan && tab, "%s: null pointer", who);
  MI_REQUIRE(g && dp && dm && dv && n_dense >= 0 && (n_dense == 0 || dense), "%s: bad dense-side arguments", who);
  MI_REQUIRE((loss_rows == nullptr) == (loss_out == nullptr), "%s: loss_rows and loss_out go together", who);
  MI_REQUIRE(stage_dst && stage_src_host && stage_bytes > 0 && stage_bytes % 16 == 0 &&
                 ((uintptr_t) stage_dst & 15) == 0 && ((uintptr_t) stage_src_host & 15) == 0,
             "%s: the staged block needs both pointers and a size that are multiples of 16", who);
  const int64_t B = plan->n_bags / (plan->T > 0 ? plan->T : 1);
  MI_REQUIRE(B >= 1, "%s: empty batch", who);
  if (plan->unique_rows) {
    mi_set_error("%s: takes the plan of the one-launch update (no run arrays)", who);
    return MI_ERR_UNSUPPORTED;
  }
  FmDenseStageExtra extra;
  extra.dense_blocks = n_dense + 1 + (loss_rows ? 1 : 0);
  extra.n = stage_bytes / 16;
  const size_t want = (extra.n + (size_t) UBLOCK * 8 - 1) / ((size_t) UBLOCK * 8);
  extra.blocks = extra.dense_blocks + (int) (want < 64 ? want : 64);
  extra.dst = (float4*) stage_dst;
  extra.src = (const float4*) stage_src_host;
  extra.a.g = g; extra.a.g_ld = 1; extra.a.dense = dense; extra.a.dense_ld = n_dense; extra.a.n_dense = n_dense;
  extra.a.B = B; extra.a.grad_out = nullptr; extra.a.loss_rows = loss_rows; extra.a.loss_ld = 1;
  extra.a.dp = dp; extra.a.dm = dm; extra.a.dv = dv; extra.a.lr_t = dense_lr_t; extra.a.lr_dev = dense_lr_dev;
  extra.a.b1 = beta1; extra.a.b2 = beta2; extra.a.eps = eps; extra.a.loss_out = loss_out;
  return fm_bwd_apply_impl(who, plan, tab, g, 1, S, tab->dim, nullptr, opt, lr, lr_dev, beta1, beta2, eps, stream,
                           extra);
}

