// Factorisation-machine forward fused with the embedding gather, and the dense-side
// gradient sums.  The reference has no FM (its model is the linear colour network of
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
// Lane mapping (G lanes per sample, G = 16 for dim <= 64): lane l owns column chunk
// c = l % (dim/4) and walks fields q, q+FP, ... with q = l / (dim/4), FP = G / (dim/4).
// For dim 16 that is 4 lanes x 16 B = one 64 B row per field, 4 fields in flight per
// sample per step, 6-7 independent row loads per lane.
#include "common.h"

namespace {

template <typename IDX, int DIM>
__global__ __launch_bounds__(256) void fm_fwd_kernel(const float* __restrict__ V, const float* __restrict__ w1,
                                                    const int64_t* __restrict__ toffs, int T,
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
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
  float lin = 0.f;
  int bad = 0;
  for (int f = q; f < T; f += FP) {
    const int64_t id = mi_ld_id<IDX>(ids, (int64_t) gid * T + f);
    const int64_t base = toffs[f];
    const int64_t rows = toffs[f + 1] - base;
    if (id < 0 || id >= rows) {  // FM has no padding id: every field carries one id
      bad |= MI_IDERR_RANGE;
      continue;
    }
    const int64_t row = base + id;
    const float4 v = *reinterpret_cast<const float4*>(V + row * DIM + c * 4);
    if (c == 0) lin += w1[row];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    sq.x = fmaf(v.x, v.x, sq.x); sq.y = fmaf(v.y, v.y, sq.y);
    sq.z = fmaf(v.z, v.z, sq.z); sq.w = fmaf(v.w, v.w, sq.w);
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

// out[j] = sum_b g[b]*dense[b][j] for j < n_dense, out[n_dense] = sum_b g[b]; one workgroup
// per output, fixed summation order.
__global__ __launch_bounds__(1024) void fm_bwd_dense_kernel(const float* __restrict__ g, const float* __restrict__ dense,
                                                           int n_dense, int64_t B, float* __restrict__ out) {
  __shared__ float sh[1024];
  const int j = blockIdx.x;
  float acc = 0.f;
  if (j < n_dense) {
    for (int64_t b = threadIdx.x; b < B; b += 1024) acc = fmaf(g[b], dense[b * n_dense + j], acc);
  } else {
    for (int64_t b = threadIdx.x; b < B; b += 1024) acc += g[b];
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int) threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[j] = sh[0];
}

template <typename IDX>
int launch_fm(const float* V, const float* w1, const int64_t* toffs, int T, int dim, const void* ids, int B,
              const float* dense, int n_dense, const float* wd, const float* bias, const float* labels, float gscale,
              float* logit, float* S, float* loss_rows, float* g, int32_t* err, hipStream_t s) {
#define FM_CASE(D)                                                                                              \
  case D: {                                                                                                     \
    constexpr int G_ = (D / 4) > 16 ? (D / 4) : 16;                                                             \
    hipLaunchKernelGGL((fm_fwd_kernel<IDX, D>), dim3((unsigned) mi_cdiv(B, 256 / G_)), dim3(256), 0, s, V, w1,   \
                       toffs, T, ids, B, dense, n_dense, wd, bias, labels, gscale, logit, S, loss_rows, g, err); \
    break;                                                                                                      \
  }
  switch (dim) {
    FM_CASE(4)
    FM_CASE(8)
    FM_CASE(16)
    FM_CASE(32)
    FM_CASE(64)
    FM_CASE(128)
    default:
      mi_set_error("mi_fm_fwd: dim %d not built (4, 8, 16, 32, 64, 128)", dim);
      return MI_ERR_UNSUPPORTED;
  }
#undef FM_CASE
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace

extern "C" int mi_fm_fwd(const float* V, const float* w1, const int64_t* table_offsets, int T, int dim,
                         const void* ids, int idx_bits, int64_t B, const float* dense, int n_dense, const float* wd,
                         const float* bias, const float* labels, float gscale, float* logit, float* S,
                         float* loss_rows, float* g, int32_t* err, void* stream) {
  MI_REQUIRE(V && w1 && table_offsets && ids, "mi_fm_fwd: null pointer");
  MI_REQUIRE(T >= 1 && B >= 0 && B < ((int64_t) 1 << 31) / T, "mi_fm_fwd: bad T/B");
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_fm_fwd: idx_bits must be 32 or 64");
  MI_REQUIRE(n_dense == 0 || (dense && wd), "mi_fm_fwd: dense features need dense and wd");
  MI_REQUIRE((((uintptr_t) V | (uintptr_t) S) & 15) == 0, "mi_fm_fwd: V and S must be 16-byte aligned");
  if (B == 0) return MI_OK;
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    return launch_fm<int32_t>(V, w1, table_offsets, T, dim, ids, (int) B, dense, n_dense, wd, bias, labels, gscale,
                              logit, S, loss_rows, g, err, s);
  return launch_fm<int64_t>(V, w1, table_offsets, T, dim, ids, (int) B, dense, n_dense, wd, bias, labels, gscale,
                            logit, S, loss_rows, g, err, s);
}

extern "C" int mi_fm_bwd_dense(const float* g, const float* dense, int n_dense, int64_t B, float* out,
                               void* stream) {
  MI_REQUIRE(g && out && n_dense >= 0 && B >= 0 && (n_dense == 0 || dense), "mi_fm_bwd_dense: bad arguments");
  hipLaunchKernelGGL(fm_bwd_dense_kernel, dim3(n_dense + 1), dim3(1024), 0, (hipStream_t) stream, g, dense, n_dense, B,
                     out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
