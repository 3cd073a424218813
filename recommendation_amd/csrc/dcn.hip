// Memory-bound pieces around the bf16 contractions of the DCN-v2 / DIN dense side: packing the
// dense features into the bf16 input row, column sums (bias gradients), the split-K slab
// reduction fused with TensorFlow-form Adam (recommendation/code/model_trainer.py:157) and the
// bf16 re-cast of the weights, and the 1-wide output layer with its BCE loss.  None of this is
// in the reference (its network is 5,910 fp32 parameters, recommendation/code/r_model.py:24-58).
// Every sum has a fixed order; nothing uses float atomics.
#include "common.h"

namespace {

// x[b][col0 + j] = bf16(dense[b][j]) for j < nd, zero for col0+nd <= col < d_pad
__global__ __launch_bounds__(256) void pack_dense_kernel(const float* __restrict__ dense, int nd, int64_t B,
                                                        unsigned short* __restrict__ x, int64_t ld, int col0,
                                                        int d_pad) {
  const int w = d_pad - col0;
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= B * w) return;
  const int64_t b = i / w;
  const int j = (int) (i - b * w);
  x[b * ld + col0 + j] = j < nd ? mi_f2bf(dense[b * nd + j]) : (unsigned short) 0;
}

// partial[s][n] = sum over rows of slice s of X[m][n].  Block = CT column threads (8 columns each)
// x RL = 256/CT row lanes (CT = 64, or 16 for matrices of up to 128 columns, which would leave most
// of 64 column threads idle); a row lane walks rows lo+ty, lo+ty+RL, ... with four loads in flight,
// the lanes are then added in lane order: a fixed order for a given (M, N, slices).
template <int CT>
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const unsigned short* __restrict__ X, int64_t ld, int64_t M,
                                                         int N, int rows_per_slice, float* __restrict__ partial,
                                                         int64_t ld_partial) {
  constexpr int RL = 256 / CT;
  __shared__ float sh[RL - 1][CT][8];
  const int tx = threadIdx.x % CT, ty = threadIdx.x / CT;
  const int n = (blockIdx.x * CT + tx) * 8;
  const int s = blockIdx.y;
  const int64_t lo = (int64_t) s * rows_per_slice;
  const int64_t hi = lo + rows_per_slice < M ? lo + rows_per_slice : M;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  if (n < N) {
    const bool wide = n + 8 <= N;  // N % 4 == 0: the last thread of a row may own only 4 columns
    int64_t m = lo + ty;
    for (; m + 3 * RL < hi; m += 4 * RL) {
      uint4 u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned short* p = X + (m + RL * j) * ld + n;
        if (wide) u[j] = *reinterpret_cast<const uint4*>(p);
        else { const uint2 h = *reinterpret_cast<const uint2*>(p); u[j] = make_uint4(h.x, h.y, 0u, 0u); }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] += __uint_as_float(u[j].x << 16); acc[1] += __uint_as_float(u[j].x & 0xFFFF0000u);
        acc[2] += __uint_as_float(u[j].y << 16); acc[3] += __uint_as_float(u[j].y & 0xFFFF0000u);
        acc[4] += __uint_as_float(u[j].z << 16); acc[5] += __uint_as_float(u[j].z & 0xFFFF0000u);
        acc[6] += __uint_as_float(u[j].w << 16); acc[7] += __uint_as_float(u[j].w & 0xFFFF0000u);
      }
    }
    for (; m < hi; m += RL) {
      const unsigned short* p = X + m * ld + n;
      uint4 u;
      if (wide) u = *reinterpret_cast<const uint4*>(p);
      else { const uint2 h = *reinterpret_cast<const uint2*>(p); u = make_uint4(h.x, h.y, 0u, 0u); }
      acc[0] += __uint_as_float(u.x << 16); acc[1] += __uint_as_float(u.x & 0xFFFF0000u);
      acc[2] += __uint_as_float(u.y << 16); acc[3] += __uint_as_float(u.y & 0xFFFF0000u);
      acc[4] += __uint_as_float(u.z << 16); acc[5] += __uint_as_float(u.z & 0xFFFF0000u);
      acc[6] += __uint_as_float(u.w << 16); acc[7] += __uint_as_float(u.w & 0xFFFF0000u);
    }
  }
  if (ty > 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[ty - 1][tx][k] = acc[k];
  }
  __syncthreads();
  if (ty == 0 && n < N) {
#pragma unroll
    for (int t = 0; t < RL - 1; ++t)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += sh[t][tx][k];
    float* out = partial + (int64_t) s * ld_partial + n;
    *reinterpret_cast<float4*>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (n + 8 <= N) *reinterpret_cast<float4*>(out + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

// g = sum_s slab[s][i] (s ascending) -> TensorFlow-form Adam on the fp32 master -> bf16 copy
__global__ __launch_bounds__(256) void slab_adam_kernel(const float* __restrict__ slab, int S, int64_t n,
                                                       int64_t slab_stride, float* __restrict__ w,
                                                       float* __restrict__ m, float* __restrict__ v, float lr_t,
                                                       const float* __restrict__ lr_dev, float b1, float b2, float eps,
                                                       unsigned short* __restrict__ w_bf16,
                                                       float* __restrict__ grad_out) {
  const int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 8 <= S; s += 8) {  // eight loads in flight, added in slab order
    float4 p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = *reinterpret_cast<const float4*>(slab + (int64_t) (s + j) * slab_stride + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) { g.x += p[j].x; g.y += p[j].y; g.z += p[j].z; g.w += p[j].w; }
  }
  for (; s < S; ++s) {
    const float4 p = *reinterpret_cast<const float4*>(slab + (int64_t) s * slab_stride + i);
    g.x += p.x; g.y += p.y; g.z += p.z; g.w += p.w;
  }
  if (grad_out) *reinterpret_cast<float4*>(grad_out + i) = g;
  if (!w) return;
  const float alpha = lr_dev ? lr_dev[0] : lr_t;
  float4 wi = *reinterpret_cast<const float4*>(w + i);
  float4 mi = *reinterpret_cast<const float4*>(m + i);
  float4 vi = *reinterpret_cast<const float4*>(v + i);
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#define ADAM1(c)                                   \
  mi.c += (g.c - mi.c) * omb1;                     \
  vi.c += (g.c * g.c - vi.c) * omb2;               \
  wi.c -= (mi.c * alpha) / (sqrtf(vi.c) + eps);
  ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
  *reinterpret_cast<float4*>(w + i) = wi;
  *reinterpret_cast<float4*>(m + i) = mi;
  *reinterpret_cast<float4*>(v + i) = vi;
  if (w_bf16) {
    ushort4 u;
    u.x = mi_f2bf(wi.x); u.y = mi_f2bf(wi.y); u.z = mi_f2bf(wi.z); u.w = mi_f2bf(wi.w);
    *reinterpret_cast<ushort4*>(w_bf16 + i) = u;
  }
}

// The same for a short vector under many slabs (a bias under 256 row slices): one thread per four
// elements would walk all the slabs alone, so 16 lanes share them - lane l adds slabs l, l+16, ...
// in order, then the 16 partial sums are added in lane order.  A fixed order for a given S.
__global__ __launch_bounds__(256) void slab_adam_small_kernel(const float* __restrict__ slab, int S, int64_t n,
                                                             int64_t slab_stride, float* __restrict__ w,
                                                             float* __restrict__ m, float* __restrict__ v, float lr_t,
                                                             const float* __restrict__ lr_dev, float b1, float b2,
                                                             float eps, unsigned short* __restrict__ w_bf16,
                                                             float* __restrict__ grad_out) {
  __shared__ float4 sh[16][16];
  const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int64_t i = ((int64_t) blockIdx.x * 16 + col) * 4;
  const bool live = i < n;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    for (int s0 = lane; s0 < S; s0 += 16 * 8) {
      float4 p[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = s0 + 16 * j;
        p[j] = s < S ? *reinterpret_cast<const float4*>(slab + (int64_t) s * slab_stride + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { g.x += p[j].x; g.y += p[j].y; g.z += p[j].z; g.w += p[j].w; }
    }
  }
  sh[lane][col] = g;
  __syncthreads();
  if (lane != 0 || !live) return;
  for (int l = 1; l < 16; ++l) { const float4 p = sh[l][col]; g.x += p.x; g.y += p.y; g.z += p.z; g.w += p.w; }
  if (grad_out) *reinterpret_cast<float4*>(grad_out + i) = g;
  if (!w) return;
  const float alpha = lr_dev ? lr_dev[0] : lr_t;
  float4 wi = *reinterpret_cast<const float4*>(w + i);
  float4 mi = *reinterpret_cast<const float4*>(m + i);
  float4 vi = *reinterpret_cast<const float4*>(v + i);
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#define ADAM1(c)                                   \
  mi.c += (g.c - mi.c) * omb1;                     \
  vi.c += (g.c * g.c - vi.c) * omb2;               \
  wi.c -= (mi.c * alpha) / (sqrtf(vi.c) + eps);
  ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
  *reinterpret_cast<float4*>(w + i) = wi;
  *reinterpret_cast<float4*>(m + i) = mi;
  *reinterpret_cast<float4*>(v + i) = vi;
  if (w_bf16) {
    ushort4 u;
    u.x = mi_f2bf(wi.x); u.y = mi_f2bf(wi.y); u.z = mi_f2bf(wi.z); u.w = mi_f2bf(wi.w);
    *reinterpret_cast<ushort4*>(w_bf16 + i) = u;
  }
}

// Output layer: logit = h . w + bias; BCE-with-logits loss row; g = dL/dlogit.  A group of `lpr`
// lanes (a power of two <= 64, 4 columns per lane and pass) per sample.
__global__ __launch_bounds__(256) void head_fwd_kernel(const unsigned short* __restrict__ h, int64_t ld, int K,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ labels, float gscale, int64_t B,
                                                      int lpr, float* __restrict__ logit,
                                                      float* __restrict__ loss_rows, float* __restrict__ g) {
  const int per_block = 256 / lpr;
  const int64_t b = (int64_t) blockIdx.x * per_block + threadIdx.x / lpr;
  const int lane = threadIdx.x % lpr;
  float acc = 0.f;
  if (b < B) {
    for (int k = lane * 4; k < K; k += lpr * 4) {
      const ushort4 u = *reinterpret_cast<const ushort4*>(h + b * ld + k);
      const float4 ww = *reinterpret_cast<const float4*>(w + k);
      acc = fmaf(mi_bf2f(u.x), ww.x, acc); acc = fmaf(mi_bf2f(u.y), ww.y, acc);
      acc = fmaf(mi_bf2f(u.z), ww.z, acc); acc = fmaf(mi_bf2f(u.w), ww.w, acc);
    }
  }
  for (int o = lpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane != 0 || b >= B) return;
  const float z = acc + (bias ? bias[0] : 0.f);
  if (logit) logit[b] = z;
  if (labels) {
    const float y = labels[b];
    const float e = expf(-fabsf(z));
    if (loss_rows) loss_rows[b] = fmaxf(z, 0.f) - z * y + log1pf(e);
    const float sig = z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
    if (g) g[b] = (sig - y) * gscale;
  }
}

// partial[s][k] = sum over samples of slice s of g[b]*h[b][k] (k < K);  partial[s][K] = sum g[b].
// Block = ct column threads (4 columns each; a power of two <= 64; the thread at k == K owns the
// bias column) x 256/ct row lanes; a row lane walks rows lo+ty, lo+ty+rl, ...; lanes are then
// added in lane order.  With w and dh given the same pass also writes the data gradient
// dh[b][k] = g[b]*w[k] where h[b][k] > 0, else 0 (h is read once for both).
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ g,
                                                        const unsigned short* __restrict__ h, int64_t ld, int K,
                                                        int64_t B, int rows_per_slice, int ct,
                                                        float* __restrict__ partial, int64_t ld_partial,
                                                        const float* __restrict__ w, unsigned short* __restrict__ dh,
                                                        int64_t ld_dh) {
  __shared__ float sh[256][4];
  const int rl = 256 / ct;
  const int tx = threadIdx.x % ct, ty = threadIdx.x / ct;
  const int k = (blockIdx.x * ct + tx) * 4;
  const int s = blockIdx.y;
  const int64_t lo = (int64_t) s * rows_per_slice;
  const int64_t hi = lo + rows_per_slice < B ? lo + rows_per_slice : B;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) {
    const float4 ww = dh ? *reinterpret_cast<const float4*>(w + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    auto put = [&](int64_t row, float gb, ushort4 u) {
      ushort4 o;
      o.x = mi_bf2f(u.x) > 0.f ? mi_f2bf(gb * ww.x) : (unsigned short) 0;
      o.y = mi_bf2f(u.y) > 0.f ? mi_f2bf(gb * ww.y) : (unsigned short) 0;
      o.z = mi_bf2f(u.z) > 0.f ? mi_f2bf(gb * ww.z) : (unsigned short) 0;
      o.w = mi_bf2f(u.w) > 0.f ? mi_f2bf(gb * ww.w) : (unsigned short) 0;
      *reinterpret_cast<ushort4*>(dh + row * ld_dh + k) = o;
    };
    int64_t b = lo + ty;
    for (; b + 3 * (int64_t) rl < hi; b += 4 * (int64_t) rl) {
      ushort4 u[4];
      float gb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u[j] = *reinterpret_cast<const ushort4*>(h + (b + (int64_t) j * rl) * ld + k);
        gb[j] = g[b + (int64_t) j * rl];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc.x = fmaf(gb[j], mi_bf2f(u[j].x), acc.x); acc.y = fmaf(gb[j], mi_bf2f(u[j].y), acc.y);
        acc.z = fmaf(gb[j], mi_bf2f(u[j].z), acc.z); acc.w = fmaf(gb[j], mi_bf2f(u[j].w), acc.w);
      }
      if (dh) {
#pragma unroll
        for (int j = 0; j < 4; ++j) put(b + (int64_t) j * rl, gb[j], u[j]);
      }
    }
    for (; b < hi; b += rl) {
      const float gb = g[b];
      const ushort4 u = *reinterpret_cast<const ushort4*>(h + b * ld + k);
      acc.x = fmaf(gb, mi_bf2f(u.x), acc.x); acc.y = fmaf(gb, mi_bf2f(u.y), acc.y);
      acc.z = fmaf(gb, mi_bf2f(u.z), acc.z); acc.w = fmaf(gb, mi_bf2f(u.w), acc.w);
      if (dh) put(b, gb, u);
    }
  } else if (k == K) {
    for (int64_t b = lo + ty; b < hi; b += rl) acc.x += g[b];
  }
  sh[threadIdx.x][0] = acc.x; sh[threadIdx.x][1] = acc.y; sh[threadIdx.x][2] = acc.z; sh[threadIdx.x][3] = acc.w;
  __syncthreads();
  if (ty != 0 || k > K) return;
  for (int t = 1; t < rl; ++t) {
    acc.x += sh[t * ct + tx][0]; acc.y += sh[t * ct + tx][1]; acc.z += sh[t * ct + tx][2]; acc.w += sh[t * ct + tx][3];
  }
  float* out = partial + (int64_t) s * ld_partial;
  if (k == K) out[K] = acc.x;
  else *reinterpret_cast<float4*>(out + k) = acc;
}

// out (bf16) = in (fp32), n multiple of 4
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, int64_t n,
                                                       unsigned short* __restrict__ out) {
  const int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float4 f = *reinterpret_cast<const float4*>(in + i);
  ushort4 u;
  u.x = mi_f2bf(f.x); u.y = mi_f2bf(f.y); u.z = mi_f2bf(f.z); u.w = mi_f2bf(f.w);
  *reinterpret_cast<ushort4*>(out + i) = u;
}

// dst[c][r] = src[r][c] (bf16), 64 x 64 tiles through LDS so that both sides move whole rows
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const unsigned short* __restrict__ src, int64_t ld_src,
                                                            int R, int C, unsigned short* __restrict__ dst,
                                                            int64_t ld_dst) {
  __shared__ unsigned short tile[64][66];  // 66: the column reads below fall on different banks
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int q = (threadIdx.x & 15) * 4, rr = threadIdx.x >> 4;  // four elements per thread and pass, 16 rows per pass
  const bool vec = ((ld_src | ld_dst) & 3) == 0 && (((uintptr_t) src | (uintptr_t) dst) & 7) == 0 && r0 + 64 <= R && c0 + 64 <= C;
  if (vec) {
    ushort4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const ushort4*>(src + (int64_t) (r0 + rr + 16 * p) * ld_src + c0 + q);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned short* t = &tile[rr + 16 * p][q];
      t[0] = v[p].x; t[1] = v[p].y; t[2] = v[p].z; t[3] = v[p].w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int cr = rr + 16 * p;  // row of dst = column of src
      ushort4 o;
      o.x = tile[q][cr]; o.y = tile[q + 1][cr]; o.z = tile[q + 2][cr]; o.w = tile[q + 3][cr];
      *reinterpret_cast<ushort4*>(dst + (int64_t) (c0 + cr) * ld_dst + r0 + q) = o;
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(int64_t) (r0 + i) * ld_src + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < C && r0 + tx < R) dst[(int64_t) (c0 + i) * ld_dst + r0 + tx] = tile[tx][i];
}

struct CrossPtrs {
  const unsigned short* dx[9];
  const unsigned short* u[8];
};

// out = a * b elementwise over [M][N] bf16 (8 values per thread)
__global__ __launch_bounds__(256) void mul_bf16_kernel(const uint16_t* __restrict__ a, int64_t ld_a,
                                                      const uint16_t* __restrict__ b, int64_t ld_b, int64_t M, int N,
                                                      uint16_t* __restrict__ out, int64_t ld_out) {
  const int per_row = N / 8;
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= M * per_row) return;
  const int64_t m = i / per_row;
  const int n = (int) (i - m * per_row) * 8;
  const uint4 va = *reinterpret_cast<const uint4*>(a + m * ld_a + n), vb = *reinterpret_cast<const uint4*>(b + m * ld_b + n);
  const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
  uint32_t wo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = __uint_as_float(wa[j] << 16) * __uint_as_float(wb[j] << 16);
    const float hi = __uint_as_float(wa[j] & 0xFFFF0000u) * __uint_as_float(wb[j] & 0xFFFF0000u);
    wo[j] = (uint32_t) mi_f2bf(lo) | ((uint32_t) mi_f2bf(hi) << 16);
  }
  *reinterpret_cast<uint4*>(out + m * ld_out + n) = make_uint4(wo[0], wo[1], wo[2], wo[3]);
}


// out = dx[0] + sum_l dx[l+1] * u[l], thread per 4 columns, l ascending
__global__ __launch_bounds__(256) void cross_input_grad_kernel(CrossPtrs p, int L, int64_t M, int N, int64_t ld,
                                                              float* __restrict__ out, int64_t ld_out) {
  const int nq = N / 4;
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= M * nq) return;
  const int64_t m = i / nq;
  const int n = (int) (i - m * nq) * 4;
  const int64_t off = m * ld + n;
  auto ld4 = [&](const unsigned short* q) {
    const ushort4 v = *reinterpret_cast<const ushort4*>(q + off);
    return make_float4(mi_bf2f(v.x), mi_bf2f(v.y), mi_bf2f(v.z), mi_bf2f(v.w));
  };
  float4 acc = ld4(p.dx[0]);
  for (int l = 0; l < L; ++l) {
    const float4 d = ld4(p.dx[l + 1]), uu = ld4(p.u[l]);
    acc.x = fmaf(d.x, uu.x, acc.x); acc.y = fmaf(d.y, uu.y, acc.y);
    acc.z = fmaf(d.z, uu.z, acc.z); acc.w = fmaf(d.w, uu.w, acc.w);
  }
  *reinterpret_cast<float4*>(out + m * ld_out + n) = acc;
}

bool a16(const void* p) { return ((uintptr_t) p & 15) == 0; }
bool a8(const void* p) { return ((uintptr_t) p & 7) == 0; }

}  // namespace

extern "C" int mi_pack_dense_bf16(const float* dense, int n_dense, int64_t B, void* x, int64_t ld, int col0,
                                  int d_pad, void* stream) {
  MI_REQUIRE(x && B >= 0 && n_dense >= 0 && col0 >= 0 && col0 + n_dense <= d_pad && d_pad <= ld &&
                 (n_dense == 0 || dense),
             "mi_pack_dense_bf16: bad arguments");
  const int64_t n = B * (d_pad - col0);
  if (n == 0) return MI_OK;
  hipLaunchKernelGGL(pack_dense_kernel, dim3((unsigned) mi_cdiv(n, 256)), dim3(256), 0, (hipStream_t) stream, dense,
                     n_dense, B, (unsigned short*) x, ld, col0, d_pad);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_colsum_bf16(const void* X, int64_t ld, int64_t M, int N, int slices, float* partial,
                              int64_t ld_partial, void* stream) {
  MI_REQUIRE(X && partial && M >= 1 && N >= 4 && N % 4 == 0 && ld % 4 == 0 && a8(X) && a16(partial) &&
                 ld_partial % 4 == 0 && ld_partial >= N && slices >= 1 && slices <= 65535,
             "mi_colsum_bf16: bad arguments");
  MI_REQUIRE(ld % 8 == 0 && ((uintptr_t) X & 15) == 0, "mi_colsum_bf16: rows must be 16-byte aligned");
  const int rps = (int) mi_cdiv(M, slices);
  if (N <= 128)
    hipLaunchKernelGGL(colsum_bf16_kernel<16>, dim3(1, (unsigned) slices), dim3(256), 0, (hipStream_t) stream,
                       (const unsigned short*) X, ld, M, N, rps, partial, ld_partial);
  else
    hipLaunchKernelGGL(colsum_bf16_kernel<64>, dim3((unsigned) mi_cdiv(N, 512), (unsigned) slices), dim3(256), 0,
                       (hipStream_t) stream, (const unsigned short*) X, ld, M, N, rps, partial, ld_partial);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_slab_adam(const float* slab, int n_slabs, int64_t n, int64_t slab_stride, float* w, float* m,
                            float* v, float lr_t, const float* lr_dev, float beta1, float beta2, float eps,
                            void* w_bf16, float* grad_out, void* stream) {
  MI_REQUIRE(slab && n_slabs >= 1 && n >= 4 && n % 4 == 0 && slab_stride % 4 == 0 && slab_stride >= n && a16(slab),
             "mi_slab_adam: slabs must be 16-byte aligned, n a multiple of 4");
  MI_REQUIRE((w && m && v && a16(w) && a16(m) && a16(v)) || (!w && grad_out), "mi_slab_adam: need w, m, v or grad_out");
  MI_REQUIRE((!w_bf16 || a8(w_bf16)) && (!grad_out || a16(grad_out)), "mi_slab_adam: misaligned output");
  if (n <= 16384 && n_slabs >= 32)  // a few workgroups' worth of elements under many slabs
    hipLaunchKernelGGL(slab_adam_small_kernel, dim3((unsigned) mi_cdiv(n / 4, 16)), dim3(256), 0, (hipStream_t) stream,
                       slab, n_slabs, n, slab_stride, w, m, v, lr_t, lr_dev, beta1, beta2, eps,
                       (unsigned short*) w_bf16, grad_out);
  else
    hipLaunchKernelGGL(slab_adam_kernel, dim3((unsigned) mi_cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t) stream, slab,
                       n_slabs, n, slab_stride, w, m, v, lr_t, lr_dev, beta1, beta2, eps, (unsigned short*) w_bf16,
                       grad_out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_head_fwd(const void* h, int64_t ld, int K, const float* w, const float* bias, const float* labels,
                           float gscale, int64_t B, float* logit, float* loss_rows, float* g, void* stream) {
  MI_REQUIRE(h && w && B >= 1 && K >= 4 && K % 4 == 0 && ld % 4 == 0 && a8(h) && a16(w), "mi_head_fwd: bad arguments");
  int lpr = 1;
  while (lpr < 64 && lpr * 4 < K) lpr <<= 1;
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned) mi_cdiv(B, 256 / lpr)), dim3(256), 0, (hipStream_t) stream,
                     (const unsigned short*) h, ld, K, w, bias, labels, gscale, B, lpr, logit, loss_rows, g);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_head_bwd(const float* g, const float* w, const void* h, int64_t ld_h, int K, int64_t B, void* dh,
                           int64_t ld_dh, int slices, float* partial, int64_t ld_partial, void* stream) {
  MI_REQUIRE(g && w && h && dh && partial && B >= 1 && K >= 4 && K % 4 == 0 && ld_h % 4 == 0 && ld_dh % 4 == 0 &&
                 a8(h) && a8(dh) && a16(w) && a16(partial) && ld_partial % 4 == 0 && ld_partial >= K + 4 &&
                 slices >= 1 && slices <= 65535,
             "mi_head_bwd: bad arguments");
  hipStream_t s = (hipStream_t) stream;
  const int rps = (int) mi_cdiv(B, slices);
  int ct = 1;
  while (ct < 64 && ct < K / 4 + 1) ct <<= 1;
  hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned) mi_cdiv(K / 4 + 1, ct), (unsigned) slices), dim3(256), 0, s, g,
                     (const unsigned short*) h, ld_h, K, B, rps, ct, partial, ld_partial, w, (unsigned short*) dh, ld_dh);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_cast_bf16(const float* in, int64_t n, void* out, void* stream) {
  MI_REQUIRE(in && out && n >= 4 && n % 4 == 0 && a16(in) && a8(out), "mi_cast_bf16: bad arguments");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned) mi_cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t) stream, in, n,
                     (unsigned short*) out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_transpose_bf16(const void* src, int64_t ld_src, int R, int C, void* dst, int64_t ld_dst,
                                 void* stream) {
  MI_REQUIRE(src && dst && R >= 1 && C >= 1 && ld_src >= C && ld_dst >= R, "mi_transpose_bf16: bad arguments");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned) mi_cdiv(C, 64), (unsigned) mi_cdiv(R, 64)), dim3(256), 0,
                     (hipStream_t) stream, (const unsigned short*) src, ld_src, R, C, (unsigned short*) dst, ld_dst);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_mul_bf16(const void* a, int64_t ld_a, const void* b, int64_t ld_b, int64_t M, int N, void* out,
                           int64_t ld_out, void* stream) {
  MI_REQUIRE(a && b && out && M >= 0 && N >= 8 && N % 8 == 0 && ld_a >= N && ld_b >= N && ld_out >= N && ld_a % 8 == 0 &&
                 ld_b % 8 == 0 && ld_out % 8 == 0 && (((uintptr_t) a | (uintptr_t) b | (uintptr_t) out) & 15) == 0,
             "mi_mul_bf16: N and the leading dimensions must be multiples of 8, pointers 16-byte aligned");
  if (M == 0) return MI_OK;
  hipLaunchKernelGGL(mul_bf16_kernel, dim3((unsigned) mi_cdiv(M * (N / 8), 256)), dim3(256), 0, (hipStream_t) stream,
                     (const uint16_t*) a, ld_a, (const uint16_t*) b, ld_b, M, N, (uint16_t*) out, ld_out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_cross_input_grad(const void* const* dx, const void* const* u, int L, int64_t M, int N, int64_t ld,
                                   float* out, int64_t ld_out, void* stream) {
  MI_REQUIRE(dx && out && L >= 0 && L <= 8 && (L == 0 || u) && M >= 1 && N >= 4 && N % 4 == 0 && ld % 4 == 0 &&
                 ld_out % 4 == 0 && ld >= N && ld_out >= N && a16(out),
             "mi_cross_input_grad: bad arguments");
  CrossPtrs p;
  for (int l = 0; l <= L; ++l) {
    MI_REQUIRE(dx[l] && a8(dx[l]), "mi_cross_input_grad: dx[%d] null or misaligned", l);
    p.dx[l] = (const unsigned short*) dx[l];
  }
  for (int l = 0; l < L; ++l) {
    MI_REQUIRE(u[l] && a8(u[l]), "mi_cross_input_grad: u[%d] null or misaligned", l);
    p.u[l] = (const unsigned short*) u[l];
  }
  hipLaunchKernelGGL(cross_input_grad_kernel, dim3((unsigned) mi_cdiv(M * (N / 4), 256)), dim3(256), 0,
                     (hipStream_t) stream, p, L, M, N, ld, out, ld_out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
