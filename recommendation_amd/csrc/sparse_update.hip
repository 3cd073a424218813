// Sparse backward, step 2, for embedding bags: the entry points that sum bag gradients per distinct row and apply
// SGD / row-wise Adagrad / TensorFlow-form Adam (kernels and their description: sparse_update.h).
#include "sparse_update.h"

namespace {

template <int OPT>
int bag_apply(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld, int dim, float* w, float* s0,
              float* s1, OptArgs a, void* stream, const char* who) {
  PlanDev pl;
  int rc = check_plan(plan, &pl, false, who);
  if (rc) return rc;
  MI_REQUIRE(grad_out && w, "%s: null pointer", who);
  MI_REQUIRE(dim >= 1 && grad_ld >= (int64_t) plan->T * dim, "%s: grad_ld %lld < T*dim", who, (long long) grad_ld);
  const bool a16 = al16(grad_out) && al16(w) && (grad_ld % 4 == 0) && (!s1 || (al16(s0) && al16(s1)));
  Geometry g;
  rc = geometry(dim, a16, &g);
  if (rc) return rc;
  if (OPT == MI_OPT_ADAGRAD_ROWWISE && !(pow2(g.lanes) && g.lanes <= 64)) {
    mi_set_error("%s: row-wise Adagrad needs dim/%d to be a power of two <= 64 (dim=%d)", who, g.vec, dim);
    return MI_ERR_UNSUPPORTED;
  }
  BagGradSrc src{grad_out, grad_ld, dim};
  BagUpd<OPT> upd{w, s0, s1, dim, a};
  return run_segments(plan, pl, src, upd, g, (hipStream_t) stream);
}


}  // namespace

extern "C" int mi_embag_bwd_dense_grad(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld, int dim,
                                       int64_t total_rows, float* dW, void* stream) {
  PlanDev pl;
  int rc = check_plan(plan, &pl, false, "mi_embag_bwd_dense_grad");
  if (rc) return rc;
  MI_REQUIRE(grad_out && dW, "mi_embag_bwd_dense_grad: null pointer");
  MI_REQUIRE(dim >= 1 && total_rows >= 1 && grad_ld >= (int64_t) plan->T * dim, "mi_embag_bwd_dense_grad: bad shape");
  Geometry g;
  rc = geometry(dim, al16(grad_out) && al16(dW) && grad_ld % 4 == 0, &g);
  if (rc) return rc;
  hipStream_t s = (hipStream_t) stream;
  MI_HIP(hipMemsetAsync(dW, 0, (size_t) total_rows * dim * sizeof(float), s));
  BagGradSrc src{grad_out, grad_ld, dim};
  DenseGradUpd upd{dW, dim};
  return run_segments(plan, pl, src, upd, g, s);
}

extern "C" int mi_embag_bwd_apply_sgd(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld, int dim,
                                      float* weights, float lr, void* stream) {
  return bag_apply<MI_OPT_SGD>(plan, grad_out, grad_ld, dim, weights, nullptr, nullptr, OptArgs{lr, 0, 0, 0, nullptr}, stream,
                               "mi_embag_bwd_apply_sgd");
}

extern "C" int mi_embag_bwd_apply_adagrad_rowwise(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld,
                                                  int dim, float* weights, float* state, float lr, float eps,
                                                  void* stream) {
  MI_REQUIRE(state, "mi_embag_bwd_apply_adagrad_rowwise: null state");
  return bag_apply<MI_OPT_ADAGRAD_ROWWISE>(plan, grad_out, grad_ld, dim, weights, state, nullptr,
                                           OptArgs{lr, 0, 0, eps, nullptr}, stream, "mi_embag_bwd_apply_adagrad_rowwise");
}

extern "C" int mi_embag_bwd_apply_rows(const mi_bwd_plan* plan, const float* grad_rows, const int32_t* row_of_bag,
                                       int dim, int opt, float* weights, float* state, float lr, float eps,
                                       void* stream) {
  const char* who = "mi_embag_bwd_apply_rows";
  PlanDev pl;
  int rc = check_plan(plan, &pl, false, who);
  if (rc) return rc;
  MI_REQUIRE(grad_rows && row_of_bag && weights, "%s: null pointer", who);
  MI_REQUIRE(plan->pool_mode == MI_POOL_SUM, "%s: sum pooling only", who);
  MI_REQUIRE(opt == MI_OPT_SGD || (opt == MI_OPT_ADAGRAD_ROWWISE && state), "%s: optimizer must be SGD or row-wise Adagrad (with state)",
             who);
  Geometry g;
  rc = geometry(dim, al16(grad_rows) && al16(weights), &g);
  if (rc) return rc;
  RowsGradSrc src{grad_rows, row_of_bag, dim};
  hipStream_t s = (hipStream_t) stream;
  if (opt == MI_OPT_SGD) {
    BagUpd<MI_OPT_SGD> upd{weights, nullptr, nullptr, dim, OptArgs{lr, 0, 0, 0, nullptr}};
    return run_segments(plan, pl, src, upd, g, s);
  }
  if (!(pow2(g.lanes) && g.lanes <= 64)) {
    mi_set_error("%s: row-wise Adagrad needs dim/%d to be a power of two <= 64 (dim=%d)", who, g.vec, dim);
    return MI_ERR_UNSUPPORTED;
  }
  BagUpd<MI_OPT_ADAGRAD_ROWWISE> upd{weights, state, nullptr, dim, OptArgs{lr, 0, 0, eps, nullptr}};
  return run_segments(plan, pl, src, upd, g, s);
}

extern "C" int mi_embag_bwd_apply_adam_tf(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld, int dim,
                                          float* weights, float* m, float* v, float lr_t, float beta1, float beta2,
                                          float eps, void* stream) {
  MI_REQUIRE(m && v, "mi_embag_bwd_apply_adam_tf: null state");
  return bag_apply<MI_OPT_ADAM_TF>(plan, grad_out, grad_ld, dim, weights, m, v, OptArgs{lr_t, beta1, beta2, eps, nullptr},
                                   stream, "mi_embag_bwd_apply_adam_tf");
}

extern "C" int mi_embag_bwd_apply_adam_dense_tf(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld,
                                                int dim, int64_t total_rows, float* weights, float* m, float* v,
                                                float lr_t, float beta1, float beta2, float eps, void* stream) {
  const char* who = "mi_embag_bwd_apply_adam_dense_tf";
  PlanDev pl;
  int rc = check_plan(plan, &pl, true, who);
  if (rc) return rc;
  MI_REQUIRE(grad_out && weights && m && v, "%s: null pointer", who);
  MI_REQUIRE(dim >= 1 && total_rows >= 1 && grad_ld >= (int64_t) plan->T * dim, "%s: bad shape", who);
  Geometry g;
  rc = geometry(dim, al16(grad_out) && al16(weights) && al16(m) && al16(v) && grad_ld % 4 == 0, &g);
  if (rc) return rc;
  BagGradSrc src{grad_out, grad_ld, dim};
  BagUpd<MI_OPT_ADAM_TF> upd{weights, m, v, dim, OptArgs{lr_t, beta1, beta2, eps, nullptr}};
  const unsigned grid = (unsigned) mi_cdiv(total_rows, g.gpb);
  hipStream_t s = (hipStream_t) stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(adam_dense_kernel<4>, dim3(grid), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes, g.gpb, total_rows);
  else
    hipLaunchKernelGGL(adam_dense_kernel<1>, dim3(grid), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes, g.gpb, total_rows);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

