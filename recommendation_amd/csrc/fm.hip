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

template <typename IDX, int DIM>
__global__ __launch_bounds__(256) void fm_fwd_kernel(mi_fm_table tab, const int64_t* __restrict__ toffs, int T,
                                                    const void* __restrict__ ids, int B,
                                                    const float* __restrict__ dense, int n_dense,
                                                    const float* __restrict__ wd, const float* __restrict__ bias,
                                                    const float* __restrict__ labels, float gscale,
                                                    float* __restrict__ logit, float* __restrict__ S,
                                                    float* __restrict__ loss_rows, float* __restrict__ g,
                                                    int32_t* __restrict__ err) {
  constexpr int CH = DIM / 4;                // lanes across one row
  constexpr int G = CH > 16 ? CH : 16;       // lanes per sample
  constexpr int FP = G / CH;                 // fields in flight per sample
  const int gid = blockIdx.x * (256 / G) + threadIdx.x / G;
  if (gid >= B) return;
  const int l = threadIdx.x % G;
  const int c = l % CH;
  const int q = l / CH;
  const float* __restrict__ rec = tab.rec;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
  float lin = 0.f;
  int bad = 0;
  // Fields in batches of UF per lane: first every id of the batch, then every row, so the
  // loads of a batch are all in flight together (two dependent round trips per batch, not
  // two per field).  A slot past T or with a bad id reads row 0 and is masked out.
  constexpr int UF = 8;
  for (int f0 = q; f0 < T; f0 += FP * UF) {
    int64_t row[UF];
    bool ok[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int f = f0 + u * FP;
      const bool present = f < T;
      const int fs = present ? f : 0;
      const int64_t id = mi_ld_id<IDX>(ids, (int64_t) gid * T + fs);
      const int64_t base = toffs[fs];
      const int64_t rows = toffs[fs + 1] - base;
      ok[u] = present && id >= 0 && id < rows;  // FM has no padding id: every field carries one id
      if (present && !ok[u]) bad |= MI_IDERR_RANGE;
      row[u] = ok[u] ? base + id : 0;
    }
    float4 v[UF];
    float w[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const float* r = rec + row[u] * tab.stride;
      v[u] = *reinterpret_cast<const float4*>(r + c * 4);
      w[u] = (c == 0) ? r[tab.off_w1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const float4 x = make_float4(ok[u] ? v[u].x : 0.f, ok[u] ? v[u].y : 0.f, ok[u] ? v[u].z : 0.f,
                                   ok[u] ? v[u].w : 0.f);
      lin += ok[u] ? w[u] : 0.f;
      s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
      sq.x = fmaf(x.x, x.x, sq.x); sq.y = fmaf(x.y, x.y, sq.y);
      sq.z = fmaf(x.z, x.z, sq.z); sq.w = fmaf(x.w, x.w, sq.w);
    }
  }
  // combine the field phases (lanes that own the same columns)
#pragma unroll
  for (int o = CH; o < G; o <<= 1) {
    s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
    s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
    sq.x += __shfl_xor(sq.x, o, 64); sq.y += __shfl_xor(sq.y, o, 64);
    sq.z += __shfl_xor(sq.z, o, 64); sq.w += __shfl_xor(sq.w, o, 64);
  }
  if (q == 0 && S) *reinterpret_cast<float4*>(S + (int64_t) gid * DIM + c * 4) = s;
  float pair = (q == 0) ? 0.5f * ((s.x * s.x - sq.x) + (s.y * s.y - sq.y) + (s.z * s.z - sq.z) + (s.w * s.w - sq.w)) : 0.f;
  for (int j = l; j < n_dense; j += G) lin = fmaf(wd[j], dense[(int64_t) gid * n_dense + j], lin);
  float tot = pair + lin;
#pragma unroll
  for (int o = 1; o < G; o <<= 1) tot += __shfl_xor(tot, o, 64);
#pragma unroll
  for (int o = 1; o < G; o <<= 1) bad |= __shfl_xor(bad, o, 64);
  if (l == 0) {
    const float z = tot + (bias ? bias[0] : 0.f);
    if (logit) logit[gid] = z;
    if (labels) {
      const float y = labels[gid];
      const float az = fabsf(z);
      const float e = expf(-az);
      if (loss_rows) loss_rows[gid] = fmaxf(z, 0.f) - z * y + log1pf(e);
      const float sig = z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      if (g) g[gid] = (sig - y) * gscale;
    }
    if (bad && err) atomicOr(err, bad);
  }
}

__device__ __forceinline__ float block_sum_1024(float acc, float* sh) {
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int) threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  return sh[0];
}

// Workgroup j < n_dense: dwd[j] = sum_b g[b]*dense[b][j]; j == n_dense: dbias = sum_b g[b];
// j == n_dense+1 (only when loss_rows != NULL): mean of the loss rows.  Fixed summation
// order.  When dp != NULL the gradient is consumed in place by TensorFlow-form Adam on dp[j].
// g, dense rows and loss rows are g_ld, dense_ld and loss_ld floats apart (1, n_dense, 1 for plain arrays).
__global__ __launch_bounds__(1024) void fm_dense_kernel(const float* __restrict__ g, int64_t g_ld,
                                                       const float* __restrict__ dense, int64_t dense_ld,
                                                       int n_dense, int64_t B, float* __restrict__ grad_out,
                                                       const float* __restrict__ loss_rows, int64_t loss_ld,
                                                       float* __restrict__ dp,
                                                       float* __restrict__ dm, float* __restrict__ dv, float lr_t,
                                                       const float* __restrict__ lr_dev, float b1, float b2,
                                                       float eps, float* __restrict__ loss_out) {
  __shared__ float sh[1024];
  const int j = blockIdx.x;
  float acc = 0.f;
  if (j < n_dense) {
    for (int64_t b = threadIdx.x; b < B; b += 1024) acc = fmaf(g[b * g_ld], dense[b * dense_ld + j], acc);
  } else if (j == n_dense) {
    for (int64_t b = threadIdx.x; b < B; b += 1024) acc += g[b * g_ld];
  } else {
    for (int64_t b = threadIdx.x; b < B; b += 1024) acc += loss_rows[b * loss_ld];
  }
  const float tot = block_sum_1024(acc, sh);
  if (threadIdx.x != 0) return;
  if (j > n_dense) {
    loss_out[0] = tot / (float) B;
    return;
  }
  if (grad_out) grad_out[j] = tot;
  if (dp) {
    const float mi = dm[j] + (tot - dm[j]) * (1.0f - b1);
    const float vi = dv[j] + (tot * tot - dv[j]) * (1.0f - b2);
    dm[j] = mi;
    dv[j] = vi;
    dp[j] -= (mi * (lr_dev ? lr_dev[0] : lr_t)) / (sqrtf(vi) + eps);
  }
}

// sched = {alpha_t, beta1_power, beta2_power}: TensorFlow keeps the two powers as fp32 variables
// on the device and multiplies them by beta once per step; alpha_t is formed from them in fp32.
__global__ void adam_schedule_kernel(float* __restrict__ sched, float lr, float b1, float b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float b1p = sched[1] * b1;
  const float b2p = sched[2] * b2;
  sched[1] = b1p;
  sched[2] = b2p;
  const float num = lr * sqrtf(1.0f - b2p);
  sched[0] = num / (1.0f - b1p);
}

template <typename IDX>
int launch_fm(const mi_fm_table& tab, const int64_t* toffs, int T, const void* ids, int B, const float* dense,
              int n_dense, const float* wd, const float* bias, const float* labels, float gscale, float* logit,
              float* S, float* loss_rows, float* g, int32_t* err, hipStream_t s) {
#define FM_CASE(D)                                                                                               \
  case D: {                                                                                                      \
    constexpr int G_ = (D / 4) > 16 ? (D / 4) : 16;                                                              \
    hipLaunchKernelGGL((fm_fwd_kernel<IDX, D>), dim3((unsigned) mi_cdiv(B, 256 / G_)), dim3(256), 0, s, tab, toffs, \
                       T, ids, B, dense, n_dense, wd, bias, labels, gscale, logit, S, loss_rows, g, err);         \
    break;                                                                                                       \
  }
  switch (tab.dim) {
    FM_CASE(4)
    FM_CASE(8)
    FM_CASE(16)
    FM_CASE(32)
    FM_CASE(64)
    FM_CASE(128)
    default:
      mi_set_error("mi_fm_fwd: dim %d not built (4, 8, 16, 32, 64, 128)", tab.dim);
      return MI_ERR_UNSUPPORTED;
  }
#undef FM_CASE
  MI_LAUNCH_CHECK();
  return MI_OK;
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
  const int rc = mi_fm_check_table(tab, MI_OPT_SGD, "mi_fm_fwd");
  if (rc) return rc;
  MI_REQUIRE(table_offsets && ids, "mi_fm_fwd: null pointer");
  MI_REQUIRE(T >= 1 && B >= 0 && B < ((int64_t) 1 << 31) / T, "mi_fm_fwd: bad T/B");
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_fm_fwd: idx_bits must be 32 or 64");
  MI_REQUIRE(n_dense == 0 || (dense && wd), "mi_fm_fwd: dense features need dense and wd");
  MI_REQUIRE(((uintptr_t) S & 15) == 0, "mi_fm_fwd: S must be 16-byte aligned");
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    return launch_fm<int32_t>(*tab, table_offsets, T, ids, (int) B, dense, n_dense, wd, bias, labels, gscale, logit, S,
                              loss_rows, g, err, s);
  return launch_fm<int64_t>(*tab, table_offsets, T, ids, (int) B, dense, n_dense, wd, bias, labels, gscale, logit, S,
                            loss_rows, g, err, s);
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
  hipLaunchKernelGGL(fm_dense_kernel, dim3(n_dense + 1), dim3(1024), 0, (hipStream_t) stream, g, (int64_t) 1, dense,
                     (int64_t) n_dense, n_dense, B, out, (const float*) nullptr, (int64_t) 1, (float*) nullptr, (float*) nullptr, (float*) nullptr, 0.f,
                     (const float*) nullptr, 0.f, 0.f, 0.f, (float*) nullptr);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fm_dense_step(const float* g, const float* dense, int n_dense, int64_t B, const float* loss_rows,
                                float* dp, float* dm, float* dv, float lr_t, const float* lr_dev, float beta1,
                                float beta2, float eps, float* loss_out, void* stream) {
  MI_REQUIRE(g && dp && dm && dv && n_dense >= 0 && B >= 1 && (n_dense == 0 || dense), "mi_fm_dense_step: bad arguments");
  MI_REQUIRE((loss_rows == nullptr) == (loss_out == nullptr), "mi_fm_dense_step: loss_rows and loss_out go together");
  hipLaunchKernelGGL(fm_dense_kernel, dim3(n_dense + 1 + (loss_rows ? 1 : 0)), dim3(1024), 0, (hipStream_t) stream, g,
                     (int64_t) 1, dense, (int64_t) n_dense, n_dense, B, (float*) nullptr, loss_rows, (int64_t) 1, dp, dm, dv,
                     lr_t, lr_dev, beta1, beta2, eps, loss_out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fm_dense_step_rows(const float* X, int ld, int dim, int n_dense, int64_t n, float* dp, float* dm,
                                     float* dv, float lr_t, const float* lr_dev, float beta1, float beta2, float eps,
                                     float* loss_out, void* stream) {
  MI_REQUIRE(X && dp && dm && dv && loss_out && n_dense >= 0 && n >= 1 && dim >= 1 && ld >= dim + 2 + n_dense,
             "mi_fm_dense_step_rows: bad arguments");
  hipLaunchKernelGGL(fm_dense_kernel, dim3(n_dense + 2), dim3(1024), 0, (hipStream_t) stream, X + dim, (int64_t) ld,
                     X + dim + 2, (int64_t) ld, n_dense, n, (float*) nullptr, X + dim + 1, (int64_t) ld, dp, dm, dv, lr_t,
                     lr_dev, beta1, beta2, eps, loss_out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
