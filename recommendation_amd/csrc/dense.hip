// Small dense fp32 pieces of the reference colour model: tf.layers.dense without
// activation (recommendation/code/r_model.py:52-58), the softmax / top-k / cross-entropy /
// accuracy heads (r_model.py:98-119, 132-140) and TensorFlow-form dense Adam
// (recommendation/code/model_trainer.py:157-158).  Layer widths here are 3..128 columns:
// these kernels are bandwidth- and launch-bound plumbing, not MFMA work.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, int64_t x_ld,
                                                        const float* __restrict__ W, const float* __restrict__ b,
                                                        float* __restrict__ y, int64_t y_ld, int64_t B, int K, int N) {
  const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * N) return;
  const int64_t row = g / N;
  const int n = (int) (g - row * N);
  float acc = 0.f;
  const float* xr = x + row * x_ld;
  const float* wc = W + n;
  int k = 0;
  for (; k + 8 <= K; k += 8) {  // sixteen loads in flight, added in row order
    float a[8], w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = xr[k + j]; w[j] = wc[(int64_t) (k + j) * N]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(a[j], w[j], acc);
  }
  for (; k < K; ++k) acc = fmaf(xr[k], wc[(int64_t) k * N], acc);
  y[row * y_ld + n] = acc + (b ? b[n] : 0.f);
}

constexpr int ROWS_PER_SPLIT = 16;

// Backward of y = x W + b in one launch of two kinds of workgroup (they share nothing):
//   [0, splits)        partial[s][k*N+n] = sum over the split's rows of x[r][k]*dy[r][n]; partial[s][K*N+n] = sum dy[r][n]
//   [splits, gridDim)  dx[r][k] = sum_n dy[r][n] W[k][n], one thread per element (only when dx is wanted)
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float* __restrict__ x, int64_t x_ld,
                                                        const float* __restrict__ W, const float* __restrict__ dy,
                                                        int64_t dy_ld, float* __restrict__ dx, int64_t dx_ld,
                                                        float* __restrict__ partial, int64_t B, int K, int N,
                                                        int splits, int rows_per_split) {
  if ((int) blockIdx.x >= splits) {
    const int64_t g = (int64_t) (blockIdx.x - splits) * blockDim.x + threadIdx.x;
    if (g >= B * K) return;
    const int64_t row = g / K;
    const int k = (int) (g - row * K);
    float acc = 0.f;
    const float* dr = dy + row * dy_ld;
    const float* wr = W + (int64_t) k * N;
    int n = 0;
    for (; n + 8 <= N; n += 8) {  // sixteen loads in flight, added in column order
      float a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = dr[n + j]; b[j] = wr[n + j]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(a[j], b[j], acc);
    }
    for (; n < N; ++n) acc = fmaf(dr[n], wr[n], acc);
    dx[row * dx_ld + k] = acc;
    return;
  }
  const int64_t r0 = (int64_t) blockIdx.x * rows_per_split;
  const int64_t r1 = min(r0 + (int64_t) rows_per_split, B);
  const int total = K * N + N;
  float* out = partial + (int64_t) blockIdx.x * total;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    float acc = 0.f;
    // rows in batches of eight: the loads of a batch are requested together (from a clamped row, masked when
    // added) and added in row order, so a split of sixteen rows costs two round trips
    const bool is_w = e < K * N;
    const int k = is_w ? e / N : 0, n = is_w ? e - k * N : e - K * N;
    for (int64_t r = r0; r < r1; r += 8) {
      float a[8], d[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t rr = min(r + j, r1 - 1);
        a[j] = is_w ? x[rr * x_ld + k] : 1.0f;
        d[j] = dy[rr * dy_ld + n];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (r + j < r1) acc = is_w ? fmaf(a[j], d[j], acc) : acc + d[j];
    }
    out[e] = acc;
  }
}

__global__ __launch_bounds__(256) void linear_dw_final_kernel(const float* __restrict__ partial, int splits, int K,
                                                             int N, float* __restrict__ dW, float* __restrict__ db) {
  const int total = K * N + N;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  float acc = 0.f;
  int s = 0;
  for (; s + 8 <= splits; s += 8) {  // eight loads in flight, added in slice order
    float p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = partial[(int64_t) (s + j) * total + e];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += p[j];
  }
  for (; s < splits; ++s) acc += partial[(int64_t) s * total + e];
  if (e < K * N) dW[e] = acc;
  else if (db) db[e - K * N] = acc;
}

// The same under many slices (a batch of thousands of rows): 16 lanes share an element's slices - lane l adds
// slices l, l + 16, ... in order, eight loads in flight - and the 16 partial sums are added in lane order.
__global__ __launch_bounds__(256) void linear_dw_final_wide_kernel(const float* __restrict__ partial, int splits, int K,
                                                                  int N, float* __restrict__ dW, float* __restrict__ db) {
  const int total = K * N + N;
  const int lane = threadIdx.x & 15;
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = e < total;
  float acc = 0.f;
  if (live) {
    for (int s0 = lane; s0 < splits; s0 += 16 * 8) {
      float p[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = s0 + 16 * j;
        p[j] = s < splits ? partial[(int64_t) s * total + e] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += p[j];
    }
  }
  // lanes 0..15 of a group in lane order: ((l0 + l1) + l2) + ...
  float sum = __shfl(acc, (threadIdx.x & 48) | 0, 64);
#pragma unroll
  for (int l = 1; l < 16; ++l) sum += __shfl(acc, (threadIdx.x & 48) | l, 64);
  if (!live || lane != 0) return;
  if (e < K * N) dW[e] = sum;
  else if (db) db[e - K * N] = sum;
}

// one wave per row
__global__ __launch_bounds__(256) void softmax_heads_kernel(const float* __restrict__ logits, int64_t ld, int64_t B,
                                                           int N, int k, const float* __restrict__ labels,
                                                           const int32_t* __restrict__ label_ids, float grad_scale,
                                                           float* __restrict__ probs, int32_t* __restrict__ top_idx,
                                                           float* __restrict__ top_val, float* __restrict__ loss_rows,
                                                           float* __restrict__ dlogits, float* __restrict__ hit_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* z = logits + row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < N; c += 64) mx = fmaxf(mx, z[c]);
  mx = mi_wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < N; c += 64) sum += expf(z[c] - mx);
  sum = mi_wave_sum(sum);
  const float inv = 1.0f / sum;
  const float lse = logf(sum);
  float lrow = 0.f;
  for (int c = lane; c < N; c += 64) {
    const float zc = z[c] - mx;
    const float p = expf(zc) * inv;
    if (probs) probs[row * N + c] = p;
    if (labels) {
      const float y = labels[row * N + c];
      lrow -= y * (zc - lse);
      if (dlogits) dlogits[row * N + c] = (p - y) * grad_scale;
    }
  }
  if (loss_rows) {
    lrow = mi_wave_sum(lrow);
    if (lane == 0) loss_rows[row] = lrow;
  }
  if (top_idx || top_val || hit_rows) {
    // k rounds of arg-max over the not-yet-taken classes; ties go to the lower class index
    float last_v = INFINITY;
    int last_i = -1;
    float hits = 0.f;
    const int lab = label_ids ? label_ids[row] : -1;
    for (int j = 0; j < k; ++j) {
      float bv = -INFINITY;
      int bi = 0x7FFFFFFF;
      for (int c = lane; c < N; c += 64) {
        const float p = expf(z[c] - mx) * inv;
        const bool open = (p < last_v) || (p == last_v && c > last_i);
        if (open && (p > bv || (p == bv && c < bi))) { bv = p; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      last_v = bv;
      last_i = bi;
      if (lane == 0) {
        if (top_idx) top_idx[row * k + j] = bi;
        if (top_val) top_val[row * k + j] = bv;
      }
      hits += (bi == lab) ? 1.f : 0.f;
    }
    if (hit_rows && lane == 0) hit_rows[row] = hits;
  }
}

__global__ __launch_bounds__(256) void hit_rows_kernel(const int32_t* __restrict__ labels,
                                                      const int32_t* __restrict__ preds, int64_t B, int k,
                                                      float* __restrict__ hit_rows) {
  const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int32_t lab = labels[b];
  float h = 0.f;
  for (int j = 0; j < k; ++j) h += (preds[b * k + j] == lab) ? 1.f : 0.f;
  hit_rows[b] = h;
}

// One workgroup per block of 64 columns.  The 256 threads are cw columns x 256/cw row phases, cw = the block's
// column count rounded up to a power of two (a narrow matrix keeps every lane busy: 8 columns -> 32 row phases);
// a thread adds its rows eight loads at a time, then the row phases are added by a fixed LDS tree.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int64_t B, int N,
                                                    float* __restrict__ out) {
  __shared__ float sh[256];
  const int n0 = blockIdx.x * 64;
  const int cols = min(64, N - n0);
  int cw = 1;
  while (cw < cols) cw <<= 1;
  const int phases = 256 / cw;
  const int c = threadIdx.x & (cw - 1), r = threadIdx.x / cw;
  const bool live = c < cols;
  const float* col = x + n0 + (live ? c : 0);
  float acc = 0.f;
  for (int64_t b = r; b < B; b += (int64_t) phases * 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t bb = b + (int64_t) j * phases;
      v[j] = col[(bb < B ? bb : b) * ld];  // no branch around the load; masked below
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (b + (int64_t) j * phases < B) acc += v[j];
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = phases >> 1; s > 0; s >>= 1) {  // phases is a power of two
    if (r < s) sh[threadIdx.x] += sh[threadIdx.x + s * cw];
    __syncthreads();
  }
  if (r == 0 && live) out[n0 + c] = sh[c];
}

constexpr int RED_BLOCK = 1024;
__global__ __launch_bounds__(RED_BLOCK) void reduce_sum_kernel(const float* __restrict__ x, int64_t n, float scale,
                                                              float* __restrict__ out) {
  __shared__ float sh[RED_BLOCK];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += RED_BLOCK) acc += x[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = RED_BLOCK / 2; s > 0; s >>= 1) {
    if ((int) threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0] * scale;
}

__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                     const float* __restrict__ g, int64_t n, float lr_t,
                                                     const float* __restrict__ lr_dev, float b1, float b2, float eps) {
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  // TensorFlow's ApplyAdam kernel, statement for statement
  const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
  const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
  m[i] = mi;
  v[i] = vi;
  p[i] -= (mi * (lr_dev ? lr_dev[0] : lr_t)) / (sqrtf(vi) + eps);
}

int splits_for(int64_t B) { return (int) mi_cdiv(B > 0 ? B : 1, ROWS_PER_SPLIT); }

}  // namespace

extern "C" int mi_linear_fwd(const float* x, int64_t x_ld, const float* W, const float* b, float* y, int64_t y_ld,
                             int64_t B, int K, int N, void* stream) {
  MI_REQUIRE(x && W && y, "mi_linear_fwd: null pointer");
  MI_REQUIRE(B >= 0 && K >= 1 && N >= 1 && x_ld >= K && y_ld >= N, "mi_linear_fwd: bad shape");
  if (B == 0) return MI_OK;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned) mi_cdiv(B * N, 256)), dim3(256), 0, (hipStream_t) stream, x,
                     x_ld, W, b, y, y_ld, B, K, N);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" size_t mi_linear_bwd_workspace_bytes(int64_t B, int K, int N) {
  return (size_t) splits_for(B) * ((size_t) K * N + N) * sizeof(float);
}

extern "C" int mi_linear_bwd(const float* x, int64_t x_ld, const float* W, const float* dy, int64_t dy_ld, float* dx,
                             int64_t dx_ld, float* dW, float* db, int64_t B, int K, int N, void* workspace,
                             size_t workspace_bytes, void* stream) {
  MI_REQUIRE(x && W && dy && dW, "mi_linear_bwd: null pointer");
  MI_REQUIRE(B >= 0 && K >= 1 && N >= 1 && x_ld >= K && dy_ld >= N && (!dx || dx_ld >= K), "mi_linear_bwd: bad shape");
  hipStream_t s = (hipStream_t) stream;
  if (B == 0) {
    MI_HIP(hipMemsetAsync(dW, 0, (size_t) K * N * sizeof(float), s));
    if (db) MI_HIP(hipMemsetAsync(db, 0, (size_t) N * sizeof(float), s));
    return MI_OK;
  }
  if (!workspace || workspace_bytes < mi_linear_bwd_workspace_bytes(B, K, N)) {
    mi_set_error("mi_linear_bwd: workspace %zu < %zu", workspace_bytes, mi_linear_bwd_workspace_bytes(B, K, N));
    return MI_ERR_WORKSPACE;
  }
  const int splits = splits_for(B);
  const unsigned dx_blocks = dx ? (unsigned) mi_cdiv(B * K, 256) : 0u;
  hipLaunchKernelGGL(linear_bwd_kernel, dim3((unsigned) splits + dx_blocks), dim3(256), 0, s, x, x_ld, W, dy, dy_ld, dx,
                     dx_ld, (float*) workspace, B, K, N, splits, ROWS_PER_SPLIT);
  if (splits >= 64)  // a walk over that many slices by one thread is a chain of splits / 8 round trips
    hipLaunchKernelGGL(linear_dw_final_wide_kernel, dim3((unsigned) mi_cdiv(K * N + N, 16)), dim3(256), 0, s,
                       (const float*) workspace, splits, K, N, dW, db);
  else
    hipLaunchKernelGGL(linear_dw_final_kernel, dim3((unsigned) mi_cdiv(K * N + N, 256)), dim3(256), 0, s,
                       (const float*) workspace, splits, K, N, dW, db);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_softmax_heads(const float* logits, int64_t ld, int64_t B, int N, int k, const float* labels,
                                const int32_t* label_ids, float grad_scale, float* probs, int32_t* top_idx,
                                float* top_val, float* loss_rows, float* dlogits, float* hit_rows, void* stream) {
  MI_REQUIRE(logits, "mi_softmax_heads: null logits");
  MI_REQUIRE(B >= 0 && N >= 1 && N <= 4096 && ld >= N, "mi_softmax_heads: bad shape (N=%d)", N);
  MI_REQUIRE(k >= 0 && k <= N, "mi_softmax_heads: k=%d outside 0..N", k);
  MI_REQUIRE(labels || (!loss_rows && !dlogits), "mi_softmax_heads: loss/dlogits need labels");
  MI_REQUIRE(label_ids || !hit_rows, "mi_softmax_heads: hit_rows needs label_ids");
  MI_REQUIRE(k >= 1 || (!top_idx && !top_val && !hit_rows), "mi_softmax_heads: top-k outputs need k >= 1");
  if (B == 0) return MI_OK;
  hipLaunchKernelGGL(softmax_heads_kernel, dim3((unsigned) mi_cdiv(B, 4)), dim3(256), 0, (hipStream_t) stream, logits,
                     ld, B, N, k, labels, label_ids, grad_scale, probs, top_idx, top_val, loss_rows, dlogits, hit_rows);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_hit_rows(const int32_t* labels, const int32_t* preds, int64_t B, int k, float* hit_rows,
                           void* stream) {
  MI_REQUIRE(labels && preds && hit_rows && B >= 0 && k >= 1, "mi_hit_rows: bad arguments");
  if (B == 0) return MI_OK;
  hipLaunchKernelGGL(hit_rows_kernel, dim3((unsigned) mi_cdiv(B, 256)), dim3(256), 0, (hipStream_t) stream, labels,
                     preds, B, k, hit_rows);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_colsum(const float* x, int64_t ld, int64_t B, int N, float* out, void* stream) {
  MI_REQUIRE(x && out && B >= 0 && N >= 1 && ld >= N, "mi_colsum: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned) mi_cdiv(N, 64)), dim3(256), 0, (hipStream_t) stream, x, ld, B, N,
                     out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_reduce_sum(const float* x, int64_t n, float scale, float* out, void* stream) {
  MI_REQUIRE(out && (x || n == 0) && n >= 0, "mi_reduce_sum: bad arguments");
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(RED_BLOCK), 0, (hipStream_t) stream, x, n, scale, out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_adam_tf_dense(float* p, float* m, float* v, const float* g, int64_t n, float lr_t,
                                const float* lr_dev, float beta1, float beta2, float eps, void* stream) {
  MI_REQUIRE(p && m && v && g && n >= 0, "mi_adam_tf_dense: bad arguments");
  if (n == 0) return MI_OK;
  hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned) mi_cdiv(n, 256)), dim3(256), 0, (hipStream_t) stream, p, m, v, g, n,
                     lr_t, lr_dev, beta1, beta2, eps);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
