// DIN local activation unit: the memory-bound kernels either side of its bf16 contractions.
//
// Not in the reference: its nearest analogue is the 6-id bag with uniform weights
// (recommendation/code/r_model.py:68-80, n-hot == sum pooling).  Here a sample has a behaviour
// history of up to L item ids and a candidate item q; every history item k_j gets a weight
//   a_j = MLP([q, k_j, q - k_j, q * k_j])            (4D -> h1 -> h2 -> 1, ReLU hidden layers)
// and the pooled interest vector is  sum_{j < len} a_j k_j  (no softmax, as in the DIN paper).
// The MLP runs as three contractions over all B*L (sample, position) rows (mi_gemm_bf16 for the
// two wide layers, mi_head_fwd for the 1-wide one); the kernels here build its input rows from
// the gathered embeddings, pool, and turn the contraction's input gradient back into gradients
// of the embedding rows.  Positions j >= len carry id -1 in ids_all and contribute nothing.
#include "common.h"

namespace {

// One 16-lane group per (sample, position): gather k_j and q (fp32 rows of D = 64 floats),
// write the bf16 row [q | k | q-k | q*k] of A, and the position's id into ids_all[b][L+1]
// (history slots 0..L-1, the candidate in slot L; masked or invalid slots get -1).
template <typename IDX>
__global__ __launch_bounds__(256) void din_build_kernel(const float* __restrict__ E, int64_t rows,
                                                       const void* __restrict__ hist, const void* __restrict__ qid,
                                                       const int32_t* __restrict__ len, int64_t B, int L,
                                                       unsigned short* __restrict__ A, int32_t* __restrict__ ids_all,
                                                       float* __restrict__ qout, int64_t ld_q,
                                                       int32_t* __restrict__ err) {
  constexpr int D = 64;
  const int64_t item = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
  if (item >= B * L) return;
  const int c = threadIdx.x & 15;
  const int64_t b = item / L;
  const int j = (int) (item - b * L);
  const int64_t q_id = mi_ld_id<IDX>(qid, b);
  const int64_t k_id = mi_ld_id<IDX>(hist, item);
  int n = len[b];
  n = n < 0 ? 0 : (n > L ? L : n);
  const bool q_ok = q_id >= 0 && q_id < rows;
  const bool live = j < n && k_id >= 0;
  const bool k_ok = live && k_id < rows;
  if (err && ((!q_ok && j == 0) || (live && !k_ok)) && c == 0) atomicOr(err, MI_IDERR_RANGE);
  const float4 q = q_ok ? *reinterpret_cast<const float4*>(E + q_id * D + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 k = k_ok ? *reinterpret_cast<const float4*>(E + k_id * D + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned short* row = A + item * (4 * D) + c * 4;
  auto put = [&](int block, float x, float y, float z, float w) {
    ushort4 u;
    u.x = mi_f2bf(x); u.y = mi_f2bf(y); u.z = mi_f2bf(z); u.w = mi_f2bf(w);
    *reinterpret_cast<ushort4*>(row + block * D) = u;
  };
  if (k_ok) {
    put(0, q.x, q.y, q.z, q.w);
    put(1, k.x, k.y, k.z, k.w);
    put(2, q.x - k.x, q.y - k.y, q.z - k.z, q.w - k.w);
    put(3, q.x * k.x, q.y * k.y, q.z * k.z, q.w * k.w);
  } else {
    put(0, 0.f, 0.f, 0.f, 0.f); put(1, 0.f, 0.f, 0.f, 0.f); put(2, 0.f, 0.f, 0.f, 0.f); put(3, 0.f, 0.f, 0.f, 0.f);
  }
  if (c == 0) ids_all[b * (L + 1) + j] = k_ok ? (int32_t) k_id : -1;
  if (j == 0) {
    if (c == 0) ids_all[b * (L + 1) + L] = q_ok ? (int32_t) q_id : -1;
    if (qout) *reinterpret_cast<float4*>(qout + b * ld_q + c * 4) = q;
  }
}

// pooled[b] = sum_{j} a[b][j] * k_j over live positions, k_j read back from A's k block (bf16),
// positions in ascending order.  One 16-lane group per sample.
__global__ __launch_bounds__(256) void din_pool_kernel(const unsigned short* __restrict__ A, const float* __restrict__ a,
                                                      const int32_t* __restrict__ ids_all, int64_t B, int L,
                                                      float* __restrict__ pooled, int64_t ld) {
  constexpr int D = 64;
  const int64_t b = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
  if (b >= B) return;
  const int c = threadIdx.x & 15;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // eight positions' loads in flight at a time (a walk position by position is a chain of L round
  // trips); the terms are still added in ascending position order
  for (int j0 = 0; j0 < L; j0 += 8) {
    float w[8];
    ushort4 u[8];
    bool on[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = j0 + t;
      on[t] = j < L && ids_all[b * (L + 1) + (j < L ? j : 0)] >= 0;
      const int64_t at = b * L + (j < L ? j : 0);
      w[t] = a[at];
      u[t] = *reinterpret_cast<const ushort4*>(A + at * (4 * D) + D + c * 4);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (on[t]) {
        acc.x = fmaf(w[t], mi_bf2f(u[t].x), acc.x); acc.y = fmaf(w[t], mi_bf2f(u[t].y), acc.y);
        acc.z = fmaf(w[t], mi_bf2f(u[t].z), acc.z); acc.w = fmaf(w[t], mi_bf2f(u[t].w), acc.w);
      }
    }
  }
  *reinterpret_cast<float4*>(pooled + b * ld + c * 4) = acc;
}

// da[b][j] = <dpooled[b], k_j> on live positions, 0 elsewhere.  16-lane group per (sample, position).
__global__ __launch_bounds__(256) void din_pool_bwd_kernel(const unsigned short* __restrict__ A,
                                                          const float* __restrict__ dpooled, int64_t ld,
                                                          const int32_t* __restrict__ ids_all, int64_t B, int L,
                                                          float* __restrict__ da) {
  constexpr int D = 64;
  const int64_t item = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
  if (item >= B * L) return;
  const int c = threadIdx.x & 15;
  const int64_t b = item / L;
  const int j = (int) (item - b * L);
  float s = 0.f;
  if (ids_all[b * (L + 1) + j] >= 0) {
    const float4 d = *reinterpret_cast<const float4*>(dpooled + b * ld + c * 4);
    const ushort4 u = *reinterpret_cast<const ushort4*>(A + item * (4 * D) + D + c * 4);
    s = d.x * mi_bf2f(u.x) + d.y * mi_bf2f(u.y) + d.z * mi_bf2f(u.z) + d.w * mi_bf2f(u.w);
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
  if (c == 0) da[item] = s;
}

// From dA = dL/d[q | k | q-k | q*k] (bf16, [B*L][4D]) to gradients of the embedding rows, laid out
// like ids_all: dE[b][j] for history slot j, dE[b][L] for the candidate.
//   d k_j = a_j * dpooled + dA_k - dA_d + q * dA_p
//   d q   = sum_j (dA_q + dA_d + k_j * dA_p)  + dq_extra (what the layers above the unit send to q)
// One 16-lane group per sample walks its positions in order (fixed summation order for d q).
__global__ __launch_bounds__(256) void din_scatter_kernel(const unsigned short* __restrict__ A,
                                                         const unsigned short* __restrict__ dA,
                                                         const float* __restrict__ a,
                                                         const float* __restrict__ dpooled, int64_t ld_dp,
                                                         const float* __restrict__ dq_extra, int64_t ld_dq,
                                                         const int32_t* __restrict__ ids_all, int64_t B, int L,
                                                         float* __restrict__ dE) {
  constexpr int D = 64;
  const int64_t b = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
  if (b >= B) return;
  const int c = threadIdx.x & 15;
  const float4 dp = *reinterpret_cast<const float4*>(dpooled + b * ld_dp + c * 4);
  float4 dq = dq_extra ? *reinterpret_cast<const float4*>(dq_extra + b * ld_dq + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  auto ld4 = [&](const unsigned short* p) {
    const ushort4 u = *reinterpret_cast<const ushort4*>(p);
    return make_float4(mi_bf2f(u.x), mi_bf2f(u.y), mi_bf2f(u.z), mi_bf2f(u.w));
  };
  for (int j = 0; j < L; ++j) {
    float* out = dE + (b * (L + 1) + j) * D + c * 4;
    if (ids_all[b * (L + 1) + j] < 0) {
      *reinterpret_cast<float4*>(out) = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const int64_t item = b * L + j;
    const unsigned short* ar = A + item * (4 * D) + c * 4;
    const unsigned short* gr = dA + item * (4 * D) + c * 4;
    const float4 q = ld4(ar), k = ld4(ar + D);
    const float4 gq = ld4(gr), gk = ld4(gr + D), gd = ld4(gr + 2 * D), gp = ld4(gr + 3 * D);
    const float w = a[item];
    float4 dk;
    dk.x = fmaf(w, dp.x, gk.x - gd.x + q.x * gp.x);
    dk.y = fmaf(w, dp.y, gk.y - gd.y + q.y * gp.y);
    dk.z = fmaf(w, dp.z, gk.z - gd.z + q.z * gp.z);
    dk.w = fmaf(w, dp.w, gk.w - gd.w + q.w * gp.w);
    *reinterpret_cast<float4*>(out) = dk;
    dq.x += gq.x + gd.x + k.x * gp.x;
    dq.y += gq.y + gd.y + k.y * gp.y;
    dq.z += gq.z + gd.z + k.z * gp.z;
    dq.w += gq.w + gd.w + k.w * gp.w;
  }
  *reinterpret_cast<float4*>(dE + (b * (L + 1) + L) * D + c * 4) = dq;
}

// BCE-with-logits on fp32 logits: loss rows and g = (sigmoid(z) - y) * gscale
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ z, const float* __restrict__ y, int64_t B,
                                                 float gscale, float* __restrict__ loss_rows, float* __restrict__ g) {
  const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float zz = z[b], yy = y[b];
  const float e = expf(-fabsf(zz));
  if (loss_rows) loss_rows[b] = fmaxf(zz, 0.f) - zz * yy + log1pf(e);
  const float sig = zz >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
  if (g) g[b] = (sig - yy) * gscale;
}

}  // namespace

#define DIN_D 64

extern "C" int mi_din_lau_build(const float* E, int64_t rows, int dim, const void* hist, const void* qid, int idx_bits,
                                const int32_t* len, int64_t B, int L, void* A, int32_t* ids_all, float* q_out,
                                int64_t ld_q, int32_t* err, void* stream) {
  MI_REQUIRE(E && hist && qid && len && A && ids_all, "mi_din_lau_build: null pointer");
  MI_REQUIRE(dim == DIN_D, "mi_din_lau_build: dim %d not built (64)", dim);
  MI_REQUIRE(B >= 1 && L >= 1 && B * L < ((int64_t) 1 << 31) && rows >= 1 && rows < ((int64_t) 1 << 31),
             "mi_din_lau_build: bad B/L/rows");
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_din_lau_build: idx_bits must be 32 or 64");
  MI_REQUIRE(((uintptr_t) E & 15) == 0 && ((uintptr_t) A & 7) == 0 && (!q_out || (((uintptr_t) q_out & 15) == 0 && ld_q % 4 == 0)),
             "mi_din_lau_build: misaligned buffer");
  const unsigned grid = (unsigned) mi_cdiv(B * L, 16);
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    hipLaunchKernelGGL(din_build_kernel<int32_t>, dim3(grid), dim3(256), 0, s, E, rows, hist, qid, len, B, L,
                       (unsigned short*) A, ids_all, q_out, ld_q, err);
  else
    hipLaunchKernelGGL(din_build_kernel<int64_t>, dim3(grid), dim3(256), 0, s, E, rows, hist, qid, len, B, L,
                       (unsigned short*) A, ids_all, q_out, ld_q, err);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_din_lau_pool(const void* A, const float* a, const int32_t* ids_all, int64_t B, int L, int dim,
                               float* pooled, int64_t ld, void* stream) {
  MI_REQUIRE(A && a && ids_all && pooled && B >= 1 && L >= 1 && dim == DIN_D && ld % 4 == 0 &&
                 ((uintptr_t) pooled & 15) == 0,
             "mi_din_lau_pool: bad arguments");
  hipLaunchKernelGGL(din_pool_kernel, dim3((unsigned) mi_cdiv(B, 16)), dim3(256), 0, (hipStream_t) stream,
                     (const unsigned short*) A, a, ids_all, B, L, pooled, ld);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_din_lau_pool_bwd(const void* A, const float* dpooled, int64_t ld, const int32_t* ids_all, int64_t B,
                                   int L, int dim, float* da, void* stream) {
  MI_REQUIRE(A && dpooled && ids_all && da && B >= 1 && L >= 1 && dim == DIN_D && ld % 4 == 0 &&
                 ((uintptr_t) dpooled & 15) == 0,
             "mi_din_lau_pool_bwd: bad arguments");
  hipLaunchKernelGGL(din_pool_bwd_kernel, dim3((unsigned) mi_cdiv(B * L, 16)), dim3(256), 0, (hipStream_t) stream,
                     (const unsigned short*) A, dpooled, ld, ids_all, B, L, da);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_din_lau_scatter(const void* A, const void* dA, const float* a, const float* dpooled, int64_t ld_dp,
                                  const float* dq_extra, int64_t ld_dq, const int32_t* ids_all, int64_t B, int L,
                                  int dim, float* dE, void* stream) {
  MI_REQUIRE(A && dA && a && dpooled && ids_all && dE && B >= 1 && L >= 1 && dim == DIN_D && ld_dp % 4 == 0 &&
                 ((uintptr_t) dpooled & 15) == 0 && ((uintptr_t) dE & 15) == 0 &&
                 (!dq_extra || (ld_dq % 4 == 0 && ((uintptr_t) dq_extra & 15) == 0)),
             "mi_din_lau_scatter: bad arguments");
  hipLaunchKernelGGL(din_scatter_kernel, dim3((unsigned) mi_cdiv(B, 16)), dim3(256), 0, (hipStream_t) stream,
                     (const unsigned short*) A, (const unsigned short*) dA, a, dpooled, ld_dp, dq_extra, ld_dq, ids_all,
                     B, L, dE);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_bce_logits(const float* logits, const float* labels, int64_t B, float gscale, float* loss_rows,
                             float* g, void* stream) {
  MI_REQUIRE(logits && labels && B >= 1, "mi_bce_logits: bad arguments");
  hipLaunchKernelGGL(bce_kernel, dim3((unsigned) mi_cdiv(B, 256)), dim3(256), 0, (hipStream_t) stream, logits, labels,
                     B, gscale, loss_rows, g);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
