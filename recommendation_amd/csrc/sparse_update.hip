// Sparse backward, step 2: per distinct row, sum the gradients of the entries that touched
// it (fixed order, no atomics) and apply the optimizer in the same kernel.
//
// The reference's gradient of a first-layer kernel is TensorFlow's dense X^T dY with
// one-hot X followed by tf.train.AdamOptimizer on every row
// (recommendation/code/model_trainer.py:157-158); `dense_grad` and `adam_dense_tf`
// reproduce exactly that, the other updaters touch only the rows present in the batch.
//
// Work split: a run (entries of one distinct row, contiguous after the sort) is served by
// dim/VEC adjacent lanes, each owning VEC columns.  Runs longer than LONG_RUN are queued
// to `long_list` and handled by a second launch in which a whole workgroup strides over
// the run and combines its partial sums through LDS in a fixed tree, so hot rows (a
// table with three rows gets a third of the batch each) do not serialise on four lanes.
#include "common.h"

namespace {

constexpr int LONG_RUN = 96;
constexpr int UBLOCK = 256;
constexpr int LONG_GRID = 1024;

struct PlanDev {
  const uint32_t* unique_rows;
  const int32_t* perm;
  const int32_t* seg_offsets;
  const int32_t* counts;
  int32_t* long_list;
  const int64_t* offsets;
  const void* indices;
  const int64_t* toffs;
  int n_bags, bag_len, T, pool_mode, idx_bits;

  __device__ __forceinline__ int bag_of(int p) const {
    if (!offsets) return p / bag_len;
    int lo = 0, hi = n_bags;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
  }
  __device__ __forceinline__ float mean_scale(int bag) const {
    if (pool_mode != MI_POOL_MEAN) return 1.0f;
    const int t = bag % T;
    const int64_t rows = toffs[t + 1] - toffs[t];
    int64_t beg, end;
    if (offsets) { beg = offsets[bag]; end = offsets[bag + 1]; }
    else { beg = (int64_t) bag * bag_len; end = beg + bag_len; }
    int cnt = 0;
    for (int64_t j = beg; j < end; ++j) {
      const int64_t id = (idx_bits == 32) ? mi_ld_id<int32_t>(indices, j) : mi_ld_id<int64_t>(indices, j);
      cnt += (id >= 0 && id < rows) ? 1 : 0;
    }
    return cnt > 1 ? 1.0f / (float) cnt : 1.0f;
  }
};

template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float* o) {
  if constexpr (VEC == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
  } else {
    o[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float* o) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    *p = o[0];
  }
}

// ---- per-entry sources ------------------------------------------------------------------
// gradient of the pooled bag output
struct BagGradSrc {
  const float* grad_out;
  int64_t grad_ld;
  int dim;
  template <int VEC>
  __device__ __forceinline__ void load(const PlanDev& pl, int p, int c, float* o, float& sc) const {
    const int bag = pl.bag_of(p);
    const int b = bag / pl.T;
    const int t = bag - b * pl.T;
    ldv<VEC>(grad_out + (int64_t) b * grad_ld + (int64_t) t * dim + c * VEC, o);
    if (pl.pool_mode == MI_POOL_MEAN) {
      const float s = pl.mean_scale(bag);
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] *= s;
    }
    sc = 0.f;
  }
};

// FM: entry of sample b contributes g[b]*S[b] (vector) and g[b] (scalar); bag_len == 1
struct FmSrc {
  const float* g;
  const float* S;
  int dim;
  template <int VEC>
  __device__ __forceinline__ void load(const PlanDev& pl, int p, int c, float* o, float& sc) const {
    const int b = p / pl.T;
    const float gb = g[b];
    ldv<VEC>(S + (int64_t) b * dim + c * VEC, o);
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] *= gb;
    sc = gb;
  }
};

// ---- optimizers on one row (lanes cooperate; each owns VEC columns) -------------------------
__device__ __forceinline__ float group_sum(float v, int lanes) {
  for (int o = lanes >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct OptArgs {
  float lr, beta1, beta2, eps;
};

template <int OPT, int VEC>
__device__ __forceinline__ void opt_row(float* w, float* s0, float* s1, int64_t row, int dim, int c, int lanes,
                                        const float* g, const OptArgs& a) {
  const int64_t off = row * dim + c * VEC;
  float wv[VEC];
  ldv<VEC>(w + off, wv);
  if constexpr (OPT == MI_OPT_SGD) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) wv[k] -= a.lr * g[k];
  } else if constexpr (OPT == MI_OPT_ADAGRAD_ROWWISE) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) ss += g[k] * g[k];
    ss = group_sum(ss, lanes);
    const float st = s0[row] + ss / (float) dim;
    const float inv = a.lr / (sqrtf(st) + a.eps);
#pragma unroll
    for (int k = 0; k < VEC; ++k) wv[k] -= inv * g[k];
    if (c == 0) s0[row] = st;
  } else {  // MI_OPT_ADAM_TF
    float mv[VEC], vv[VEC];
    ldv<VEC>(s0 + off, mv);
    ldv<VEC>(s1 + off, vv);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      // the three statements of TensorFlow's ApplyAdam kernel, (1 - beta) formed in fp32 as there
      mv[k] += (g[k] - mv[k]) * (1.0f - a.beta1);
      vv[k] += (g[k] * g[k] - vv[k]) * (1.0f - a.beta2);
      wv[k] -= (mv[k] * a.lr) / (sqrtf(vv[k]) + a.eps);
    }
    stv<VEC>(s0 + off, mv);
    stv<VEC>(s1 + off, vv);
  }
  stv<VEC>(w + off, wv);
}

// ---- row updaters -------------------------------------------------------------------------
template <int OPT>
struct BagUpd {
  float* w; float* s0; float* s1; int dim; OptArgs a;
  template <int VEC>
  __device__ __forceinline__ void apply(int64_t row, int c, int lanes, const float* g, float) const {
    opt_row<OPT, VEC>(w, s0, s1, row, dim, c, lanes, g, a);
  }
};

struct DenseGradUpd {
  float* dW; int dim;
  template <int VEC>
  __device__ __forceinline__ void apply(int64_t row, int c, int, const float* g, float) const {
    stv<VEC>(dW + row * dim + c * VEC, g);
  }
};

template <int OPT>
struct FmUpd {
  float* V; float* w1; float* st0; float* st1; float* st2; float* st3; int dim; OptArgs a;
  template <int VEC>
  __device__ __forceinline__ void apply(int64_t row, int c, int lanes, const float* acc, float gsum) const {
    float vr[VEC], gr[VEC];
    ldv<VEC>(V + row * dim + c * VEC, vr);
#pragma unroll
    for (int k = 0; k < VEC; ++k) gr[k] = acc[k] - vr[k] * gsum;
    if constexpr (OPT == MI_OPT_ADAGRAD_ROWWISE) {
      opt_row<OPT, VEC>(V, st0, nullptr, row, dim, c, lanes, gr, a);
      if (c == 0) opt_row<OPT, 1>(w1, st1, nullptr, row, 1, 0, 1, &gsum, a);
    } else {
      opt_row<OPT, VEC>(V, st0, st1, row, dim, c, lanes, gr, a);
      if (c == 0) opt_row<OPT, 1>(w1, st2, st3, row, 1, 0, 1, &gsum, a);
    }
  }
};

// ---- kernels ------------------------------------------------------------------------------
template <int VEC, class Src, class Upd>
__global__ __launch_bounds__(UBLOCK) void seg_short_kernel(PlanDev pl, Src src, Upd upd, int lanes, int gpb) {
  const int sub = threadIdx.x / lanes;
  const int c = threadIdx.x - sub * lanes;
  if (sub >= gpb) return;
  const int u = blockIdx.x * gpb + sub;
  if (u >= pl.counts[0]) return;
  const int beg = pl.seg_offsets[u], end = pl.seg_offsets[u + 1];
  if (end - beg > LONG_RUN) {
    if (c == 0) {
      const int slot = atomicAdd(&pl.long_list[0], 1);
      pl.long_list[1 + slot] = u;
    }
    return;
  }
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  float sc = 0.f;
  int e = beg;
  for (; e + 4 <= end; e += 4) {
    int p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = pl.perm[e + i];
    float v[4][VEC], s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) src.template load<VEC>(pl, p[i], c, v[i], s[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += v[i][k];
      sc += s[i];
    }
  }
  for (; e < end; ++e) {
    float v[VEC], s;
    src.template load<VEC>(pl, pl.perm[e], c, v, s);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += v[k];
    sc += s;
  }
  upd.template apply<VEC>((int64_t) pl.unique_rows[u], c, lanes, acc, sc);
}

template <int VEC, class Src, class Upd>
__global__ __launch_bounds__(UBLOCK) void seg_long_kernel(PlanDev pl, Src src, Upd upd, int lanes, int groups) {
  __shared__ float red[UBLOCK * VEC];
  __shared__ float reds[UBLOCK];
  const int sub = threadIdx.x / lanes;
  const int c = threadIdx.x - sub * lanes;
  const bool active = sub < groups;
  const int n_long = pl.long_list[0];
  int top = 1;
  while (top < groups) top <<= 1;
  for (int i = blockIdx.x; i < n_long; i += gridDim.x) {
    const int u = pl.long_list[1 + i];
    const int beg = pl.seg_offsets[u], end = pl.seg_offsets[u + 1];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    float sc = 0.f;
    if (active) {
      int e = beg + sub;
      for (; e + 3 * groups < end; e += 4 * groups) {
        int p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = pl.perm[e + q * groups];
        float v[4][VEC], s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) src.template load<VEC>(pl, p[q], c, v[q], s[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] += v[q][k];
          sc += s[q];
        }
      }
      for (; e < end; e += groups) {
        float v[VEC], s;
        src.template load<VEC>(pl, pl.perm[e], c, v, s);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += v[k];
        sc += s;
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) red[(sub * lanes + c) * VEC + k] = acc[k];
      reds[sub * lanes + c] = sc;
    }
    __syncthreads();
    for (int stride = top >> 1; stride >= 1; stride >>= 1) {
      if (active && sub < stride && sub + stride < groups) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[(sub * lanes + c) * VEC + k] += red[((sub + stride) * lanes + c) * VEC + k];
        reds[sub * lanes + c] += reds[(sub + stride) * lanes + c];
      }
      __syncthreads();
    }
    if (sub == 0) {
      float g[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) g[k] = red[c * VEC + k];
      upd.template apply<VEC>((int64_t) pl.unique_rows[u], c, lanes, g, reds[c]);
    }
    __syncthreads();
  }
}

// Dense TensorFlow-form Adam over every row of the table; rows absent from the batch get g = 0.
template <int VEC>
__global__ __launch_bounds__(UBLOCK) void adam_dense_kernel(PlanDev pl, BagGradSrc src, BagUpd<MI_OPT_ADAM_TF> upd,
                                                           int lanes, int gpb, int64_t total_rows) {
  const int sub = threadIdx.x / lanes;
  const int c = threadIdx.x - sub * lanes;
  if (sub >= gpb) return;
  const int64_t row = (int64_t) blockIdx.x * gpb + sub;
  if (row >= total_rows) return;
  const int n_unique = pl.counts[0];
  int lo = 0, hi = n_unique;  // first u with unique_rows[u] >= row
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t) pl.unique_rows[mid] < row) lo = mid + 1; else hi = mid;
  }
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  if (lo < n_unique && (int64_t) pl.unique_rows[lo] == row) {
    const int beg = pl.seg_offsets[lo], end = pl.seg_offsets[lo + 1];
    for (int e = beg; e < end; ++e) {
      float v[VEC], s;
      src.template load<VEC>(pl, pl.perm[e], c, v, s);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += v[k];
    }
  }
  upd.template apply<VEC>(row, c, lanes, acc, 0.f);
}

// ---- host side --------------------------------------------------------------------------
struct Geometry {
  int vec, lanes, gpb;
};

int geometry(int dim, bool aligned16, Geometry* g) {
  g->vec = (dim % 4 == 0 && aligned16) ? 4 : 1;
  g->lanes = dim / g->vec;
  if (g->lanes < 1 || g->lanes > UBLOCK) {
    mi_set_error("sparse update: dim %d needs %d lanes per row (max %d)", dim, g->lanes, UBLOCK);
    return MI_ERR_UNSUPPORTED;
  }
  g->gpb = UBLOCK / g->lanes;
  return MI_OK;
}

int check_plan(const mi_bwd_plan* plan, PlanDev* pl, const char* who) {
  MI_REQUIRE(plan, "%s: null plan", who);
  MI_REQUIRE(plan->sorted_rows && plan->perm && plan->unique_rows && plan->seg_offsets && plan->counts &&
                 plan->long_list,
             "%s: plan has a null pointer", who);
  MI_REQUIRE(plan->T >= 1 && plan->n_entries >= 0 && plan->n_entries < ((int64_t) 1 << 31),
             "%s: plan T / n_entries out of range", who);
  MI_REQUIRE(plan->offsets || plan->bag_len >= 1, "%s: plan needs offsets or bag_len >= 1", who);
  MI_REQUIRE(plan->pool_mode == MI_POOL_SUM ||
                 (plan->pool_mode == MI_POOL_MEAN && plan->indices && plan->table_offsets &&
                  (plan->idx_bits == 32 || plan->idx_bits == 64)),
             "%s: mean pooling needs indices, idx_bits and table_offsets in the plan", who);
  pl->unique_rows = plan->unique_rows;
  pl->perm = plan->perm;
  pl->seg_offsets = plan->seg_offsets;
  pl->counts = plan->counts;
  pl->long_list = plan->long_list;
  pl->offsets = plan->offsets;
  pl->indices = plan->indices;
  pl->toffs = plan->table_offsets;
  pl->n_bags = (int) plan->n_bags;
  pl->bag_len = plan->bag_len;
  pl->T = plan->T;
  pl->pool_mode = plan->pool_mode;
  pl->idx_bits = plan->idx_bits;
  return MI_OK;
}

template <class Src, class Upd>
int run_segments(const mi_bwd_plan* plan, const PlanDev& pl, const Src& src, const Upd& upd, const Geometry& g,
                 hipStream_t s) {
  if (plan->n_entries == 0) return MI_OK;
  MI_HIP(hipMemsetAsync(pl.long_list, 0, sizeof(int32_t), s));
  const unsigned grid = (unsigned) mi_cdiv(plan->n_entries, g.gpb);
  if (g.vec == 4) {
    hipLaunchKernelGGL((seg_short_kernel<4, Src, Upd>), dim3(grid), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes, g.gpb);
    hipLaunchKernelGGL((seg_long_kernel<4, Src, Upd>), dim3(LONG_GRID), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes,
                       g.gpb);
  } else {
    hipLaunchKernelGGL((seg_short_kernel<1, Src, Upd>), dim3(grid), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes, g.gpb);
    hipLaunchKernelGGL((seg_long_kernel<1, Src, Upd>), dim3(LONG_GRID), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes,
                       g.gpb);
  }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
bool al16(const void* p) { return ((uintptr_t) p & 15) == 0; }

template <int OPT>
int bag_apply(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld, int dim, float* w, float* s0,
              float* s1, OptArgs a, void* stream, const char* who) {
  PlanDev pl;
  int rc = check_plan(plan, &pl, who);
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
  int rc = check_plan(plan, &pl, "mi_embag_bwd_dense_grad");
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
  return bag_apply<MI_OPT_SGD>(plan, grad_out, grad_ld, dim, weights, nullptr, nullptr, OptArgs{lr, 0, 0, 0}, stream,
                               "mi_embag_bwd_apply_sgd");
}

extern "C" int mi_embag_bwd_apply_adagrad_rowwise(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld,
                                                  int dim, float* weights, float* state, float lr, float eps,
                                                  void* stream) {
  MI_REQUIRE(state, "mi_embag_bwd_apply_adagrad_rowwise: null state");
  return bag_apply<MI_OPT_ADAGRAD_ROWWISE>(plan, grad_out, grad_ld, dim, weights, state, nullptr,
                                           OptArgs{lr, 0, 0, eps}, stream, "mi_embag_bwd_apply_adagrad_rowwise");
}

extern "C" int mi_embag_bwd_apply_adam_tf(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld, int dim,
                                          float* weights, float* m, float* v, float lr_t, float beta1, float beta2,
                                          float eps, void* stream) {
  MI_REQUIRE(m && v, "mi_embag_bwd_apply_adam_tf: null state");
  return bag_apply<MI_OPT_ADAM_TF>(plan, grad_out, grad_ld, dim, weights, m, v, OptArgs{lr_t, beta1, beta2, eps},
                                   stream, "mi_embag_bwd_apply_adam_tf");
}

extern "C" int mi_embag_bwd_apply_adam_dense_tf(const mi_bwd_plan* plan, const float* grad_out, int64_t grad_ld,
                                                int dim, int64_t total_rows, float* weights, float* m, float* v,
                                                float lr_t, float beta1, float beta2, float eps, void* stream) {
  const char* who = "mi_embag_bwd_apply_adam_dense_tf";
  PlanDev pl;
  int rc = check_plan(plan, &pl, who);
  if (rc) return rc;
  MI_REQUIRE(grad_out && weights && m && v, "%s: null pointer", who);
  MI_REQUIRE(dim >= 1 && total_rows >= 1 && grad_ld >= (int64_t) plan->T * dim, "%s: bad shape", who);
  Geometry g;
  rc = geometry(dim, al16(grad_out) && al16(weights) && al16(m) && al16(v) && grad_ld % 4 == 0, &g);
  if (rc) return rc;
  BagGradSrc src{grad_out, grad_ld, dim};
  BagUpd<MI_OPT_ADAM_TF> upd{weights, m, v, dim, OptArgs{lr_t, beta1, beta2, eps}};
  const unsigned grid = (unsigned) mi_cdiv(total_rows, g.gpb);
  hipStream_t s = (hipStream_t) stream;
  if (g.vec == 4)
    hipLaunchKernelGGL(adam_dense_kernel<4>, dim3(grid), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes, g.gpb, total_rows);
  else
    hipLaunchKernelGGL(adam_dense_kernel<1>, dim3(grid), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes, g.gpb, total_rows);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fm_bwd_apply(const mi_bwd_plan* plan, const float* g, const float* S, int dim, float* V, float* w1,
                               int opt, float* st0, float* st1, float* st2, float* st3, float lr, float beta1,
                               float beta2, float eps, void* stream) {
  const char* who = "mi_fm_bwd_apply";
  PlanDev pl;
  int rc = check_plan(plan, &pl, who);
  if (rc) return rc;
  MI_REQUIRE(g && S && V && w1, "%s: null pointer", who);
  MI_REQUIRE(!plan->offsets && plan->bag_len == 1 && plan->pool_mode == MI_POOL_SUM,
             "%s: FM takes one id per field (bag_len 1, sum)", who);
  Geometry geo;
  rc = geometry(dim, al16(S) && al16(V) && (!st0 || al16(st0)) && (!st1 || al16(st1)), &geo);
  if (rc) return rc;
  FmSrc src{g, S, dim};
  OptArgs a{lr, beta1, beta2, eps};
  hipStream_t s = (hipStream_t) stream;
  switch (opt) {
    case MI_OPT_SGD: {
      FmUpd<MI_OPT_SGD> upd{V, w1, nullptr, nullptr, nullptr, nullptr, dim, a};
      return run_segments(plan, pl, src, upd, geo, s);
    }
    case MI_OPT_ADAGRAD_ROWWISE: {
      MI_REQUIRE(st0 && st1, "%s: row-wise Adagrad needs st0 (V) and st1 (w1) accumulators", who);
      if (!(pow2(geo.lanes) && geo.lanes <= 64)) {
        mi_set_error("%s: row-wise Adagrad needs dim/%d to be a power of two <= 64", who, geo.vec);
        return MI_ERR_UNSUPPORTED;
      }
      FmUpd<MI_OPT_ADAGRAD_ROWWISE> upd{V, w1, st0, st1, nullptr, nullptr, dim, a};
      return run_segments(plan, pl, src, upd, geo, s);
    }
    case MI_OPT_ADAM_TF: {
      MI_REQUIRE(st0 && st1 && st2 && st3, "%s: Adam needs st0..st3", who);
      FmUpd<MI_OPT_ADAM_TF> upd{V, w1, st0, st1, st2, st3, dim, a};
      return run_segments(plan, pl, src, upd, geo, s);
    }
    default:
      mi_set_error("%s: unknown optimizer %d", who, opt);
      return MI_ERR_INVALID_ARG;
  }
}
