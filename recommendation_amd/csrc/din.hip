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
// FACT: the row is [k | q*k] only (2D columns).  The unit's first layer is linear in its input, so
// the q and q-k blocks can be folded into the weights: A W^T = q (Wa+Wc)^T + k (Wb-Wc)^T + (q*k) Wd^T,
// and the q term is one small product per SAMPLE instead of two more column blocks per position.
template <typename IDX, bool FACT>
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
  unsigned short* row = A + item * ((FACT ? 2 : 4) * D) + c * 4;
  auto put = [&](int block, float x, float y, float z, float w) {
    ushort4 u;
    u.x = mi_f2bf(x); u.y = mi_f2bf(y); u.z = mi_f2bf(z); u.w = mi_f2bf(w);
    *reinterpret_cast<ushort4*>(row + block * D) = u;
  };
  if constexpr (FACT) {
    if (k_ok) {
      put(0, k.x, k.y, k.z, k.w);
      put(1, q.x * k.x, q.y * k.y, q.z * k.z, q.w * k.w);
    } else {
      put(0, 0.f, 0.f, 0.f, 0.f); put(1, 0.f, 0.f, 0.f, 0.f);
    }
  } else if (k_ok) {
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
__global__ __launch_bounds__(256) void din_pool_kernel(const unsigned short* __restrict__ A, int a_cols, int k_off,
                                                      const float* __restrict__ a,
                                                      const int32_t* __restrict__ ids_all, int64_t B, int L,
                                                      float* __restrict__ pooled, int64_t ld) {
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
      u[t] = *reinterpret_cast<const ushort4*>(A + at * a_cols + k_off + c * 4);
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
__global__ __launch_bounds__(256) void din_pool_bwd_kernel(const unsigned short* __restrict__ A, int a_cols, int k_off,
                                                          const float* __restrict__ dpooled, int64_t ld,
                                                          const int32_t* __restrict__ ids_all, int64_t B, int L,
                                                          float* __restrict__ da) {
  const int64_t item = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
  if (item >= B * L) return;
  const int c = threadIdx.x & 15;
  const int64_t b = item / L;
  const int j = (int) (item - b * L);
  float s = 0.f;
  if (ids_all[b * (L + 1) + j] >= 0) {
    const float4 d = *reinterpret_cast<const float4*>(dpooled + b * ld + c * 4);
    const ushort4 u = *reinterpret_cast<const ushort4*>(A + item * a_cols + k_off + c * 4);
    // explicit roundings (the fused backward computes the same value and is held to these bits)
    s = __fmul_rn(d.x, mi_bf2f(u.x));
    s = fmaf(d.y, mi_bf2f(u.y), s);
    s = fmaf(d.z, mi_bf2f(u.z), s);
    s = fmaf(d.w, mi_bf2f(u.w), s);
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

// The same from the factored rows A = [k | q*k], dA = dL/dA = [gk | gp] (bf16, [B*L][2D]), q in fp32:
//   d k_j = a_j * dpooled + gk + q * gp
//   d q   = sum_j k_j * gp  + dq_extra (the layers above, and the per-sample q term of the first layer)
// Eight positions' loads in flight; d q still adds the positions in ascending order.
__global__ __launch_bounds__(256) void din_scatter_fact_kernel(const unsigned short* __restrict__ A,
                                                              const unsigned short* __restrict__ dA,
                                                              const float* __restrict__ a,
                                                              const float* __restrict__ dpooled, int64_t ld_dp,
                                                              const float* __restrict__ dq_extra, int64_t ld_dq,
                                                              const float* __restrict__ qv, int64_t ld_q,
                                                              const int32_t* __restrict__ ids_all, int64_t B, int L,
                                                              float* __restrict__ dE) {
  constexpr int D = 64;
  const int64_t b = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
  if (b >= B) return;
  const int c = threadIdx.x & 15;
  const float4 dp = *reinterpret_cast<const float4*>(dpooled + b * ld_dp + c * 4);
  const float4 q = *reinterpret_cast<const float4*>(qv + b * ld_q + c * 4);
  float4 dq = dq_extra ? *reinterpret_cast<const float4*>(dq_extra + b * ld_dq + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  auto cv = [](ushort4 u) { return make_float4(mi_bf2f(u.x), mi_bf2f(u.y), mi_bf2f(u.z), mi_bf2f(u.w)); };
  for (int j0 = 0; j0 < L; j0 += 8) {
    ushort4 uk[8], ugk[8], ugp[8];
    float w[8];
    bool on[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = j0 + t < L ? j0 + t : 0;
      on[t] = j0 + t < L && ids_all[b * (L + 1) + j] >= 0;
      const int64_t item = b * L + j;
      uk[t] = *reinterpret_cast<const ushort4*>(A + item * (2 * D) + c * 4);
      ugk[t] = *reinterpret_cast<const ushort4*>(dA + item * (2 * D) + c * 4);
      ugp[t] = *reinterpret_cast<const ushort4*>(dA + item * (2 * D) + D + c * 4);
      w[t] = a[item];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (j0 + t >= L) break;
      float4 dk = make_float4(0.f, 0.f, 0.f, 0.f);
      if (on[t]) {
        const float4 k = cv(uk[t]), gk = cv(ugk[t]), gp = cv(ugp[t]);
        dk.x = fmaf(w[t], dp.x, fmaf(q.x, gp.x, gk.x));
        dk.y = fmaf(w[t], dp.y, fmaf(q.y, gp.y, gk.y));
        dk.z = fmaf(w[t], dp.z, fmaf(q.z, gp.z, gk.z));
        dk.w = fmaf(w[t], dp.w, fmaf(q.w, gp.w, gk.w));
        dq.x = fmaf(k.x, gp.x, dq.x); dq.y = fmaf(k.y, gp.y, dq.y);
        dq.z = fmaf(k.z, gp.z, dq.z); dq.w = fmaf(k.w, gp.w, dq.w);
      }
      *reinterpret_cast<float4*>(dE + (b * (L + 1) + j0 + t) * D + c * 4) = dk;
    }
  }
  *reinterpret_cast<float4*>(dE + (b * (L + 1) + L) * D + c * 4) = dq;
}

// The per-sample q term of the factored first layer, forward: rowbias[b][n] = b1[n] + sum_j q[b][j] Wq[n][j]
// (fp32; Wq [H][D] staged in LDS, H <= 128).  A thread owns four consecutive n of one sample.
__global__ __launch_bounds__(256) void din_qterm_fwd_kernel(const float* __restrict__ q, int64_t ld_q,
                                                           const float* __restrict__ Wq, const float* __restrict__ b1,
                                                           int64_t B, int H, float* __restrict__ out) {
  constexpr int D = 64;
  __shared__ float w[128][D + 1];
  // (eight loads requested before the first is stored: a plain loop is a chain of H*D/256 memory round trips)
#pragma unroll 8
  for (int i = threadIdx.x; i < H * D; i += 256) w[i / D][i % D] = Wq[i];
  __syncthreads();
  const int per = H / 4;  // threads per sample
  const int64_t b = (int64_t) blockIdx.x * (256 / per) + threadIdx.x / per;
  if (b >= B) return;
  const int n0 = (threadIdx.x % per) * 4;
  float acc[4] = {b1[n0], b1[n0 + 1], b1[n0 + 2], b1[n0 + 3]};
  const float* qr = q + b * ld_q;
#pragma unroll 4
  for (int j = 0; j < D; j += 4) {
    const float4 x = *reinterpret_cast<const float4*>(qr + j);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = fmaf(x.x, w[n0 + t][j], acc[t]);
      acc[t] = fmaf(x.y, w[n0 + t][j + 1], acc[t]);
      acc[t] = fmaf(x.z, w[n0 + t][j + 2], acc[t]);
      acc[t] = fmaf(x.w, w[n0 + t][j + 3], acc[t]);
    }
  }
  *reinterpret_cast<float4*>(out + b * H + n0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ... and backward, from S[b][n] = the sample's sum of dZ over its positions.  A workgroup takes
// QB_CHUNK samples:  dq[b][j] = sum_n S[b][n] Wq[n][j]  for its samples, and its share of
//   dWaT[j][n] = sum_b q[b][j] S[b][n]   and   db1[n] = sum_b S[b][n]
// as partials part[chunk][D*H + H] (samples in ascending order; the chunks are added in order by
// the weight step / mi_slab_adam).
constexpr int QB_CHUNK = 32;
__global__ __launch_bounds__(256) void din_qterm_bwd_kernel(const float* __restrict__ q, int64_t ld_q,
                                                           const float* __restrict__ S, const float* __restrict__ Wq,
                                                           int64_t B, int H, float* __restrict__ dq, int64_t ld_dq,
                                                           const float* __restrict__ dq_in, int64_t ld_in,
                                                           float* __restrict__ part) {
  constexpr int D = 64;
  __shared__ float sq[QB_CHUNK][D];
  __shared__ float sS[QB_CHUNK][128];
  __shared__ float sW[128][D];
  // (staging loops unrolled so that eight loads are requested before the first is stored: as plain loops they are
  // chains of 32 + 8 + 16 memory round trips; rows past the batch re-read row b0 and are zeroed)
#pragma unroll 8
  for (int i = threadIdx.x; i < H * D; i += 256) sW[i / D][i % D] = Wq[i];
  const int64_t b0 = (int64_t) blockIdx.x * QB_CHUNK;
  const int nb = (int) (B - b0 < QB_CHUNK ? B - b0 : QB_CHUNK);
#pragma unroll 8
  for (int i = threadIdx.x; i < QB_CHUNK * D; i += 256) {
    const int r = i / D, c = i % D;
    const float v = q[(b0 + (r < nb ? r : 0)) * ld_q + c];
    sq[r][c] = r < nb ? v : 0.f;
  }
#pragma unroll 8
  for (int i = threadIdx.x; i < QB_CHUNK * H; i += 256) {
    const int r = i / H, c = i % H;
    const float v = S[(b0 + (r < nb ? r : 0)) * H + c];
    sS[r][c] = r < nb ? v : 0.f;
  }
  __syncthreads();
  // dq: TPS threads per sample, CPT columns j each
  {
    constexpr int TPS = 256 / QB_CHUNK, CPT = D / TPS;
    const int r = threadIdx.x / TPS, j0 = (threadIdx.x % TPS) * CPT;
    if (r < nb) {
      float acc[CPT];
#pragma unroll
      for (int t = 0; t < CPT; ++t) acc[t] = dq_in ? dq_in[(b0 + r) * ld_in + j0 + t] : 0.f;
      for (int n = 0; n < H; ++n) {
        const float sv = sS[r][n];
#pragma unroll
        for (int t4 = 0; t4 < CPT / 4; ++t4) {
          const float4 wv = *reinterpret_cast<const float4*>(&sW[n][j0 + 4 * t4]);
          acc[4 * t4] = fmaf(sv, wv.x, acc[4 * t4]);
          acc[4 * t4 + 1] = fmaf(sv, wv.y, acc[4 * t4 + 1]);
          acc[4 * t4 + 2] = fmaf(sv, wv.z, acc[4 * t4 + 2]);
          acc[4 * t4 + 3] = fmaf(sv, wv.w, acc[4 * t4 + 3]);
        }
      }
#pragma unroll
      for (int t = 0; t < CPT; ++t) dq[(b0 + r) * ld_dq + j0 + t] = acc[t];
    }
  }
  // partial dWaT: thread owns (j = tid / 4, 32 columns n)
  {
    const int j = threadIdx.x >> 2, n0 = (threadIdx.x & 3) * (H / 4);
    float* out = part + (int64_t) blockIdx.x * (D * H + H);
    for (int nn = 0; nn < H / 4; nn += 8) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < QB_CHUNK; ++r) {
        const float qv = sq[r][j];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = fmaf(qv, sS[r][n0 + nn + t], acc[t]);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) out[(int64_t) j * H + n0 + nn + t] = acc[t];
    }
    if (threadIdx.x < H) {
      float a1 = 0.f;
      for (int r = 0; r < QB_CHUNK; ++r) a1 += sS[r][threadIdx.x];
      out[D * H + threadIdx.x] = a1;
    }
  }
}

// Weight step of the factored first layer.  The master W1 [H][4D] = [Wa | Wb | Wc | Wd] (columns for
// q, k, q-k, q*k) keeps TensorFlow-form Adam; its gradient is assembled from
//   slab[s][n][0..D)   partials of dZ^T K        -> g_b          dWaT[j][n] = (q^T S)[j][n] -> g_a
//   slab[s][n][D..2D)  partials of dZ^T (Q*K)    -> g_d          g_c = g_a - g_b   (d/d(q-k) = d/dq - d/dk)
// (S = per-sample sums of dZ), slabs added in slab order.  Then the tensors the next step reads:
//   Wf [H][2D] bf16 = [Wb - Wc | Wd], WfT its transpose, Wq [H][D] fp32 = Wa + Wc, WqT its transpose.
// One thread per (n, j).
__global__ __launch_bounds__(256) void din_w1_step_kernel(const float* __restrict__ slab, int S, int64_t slab_stride,
                                                         const float* __restrict__ dWaT, int n_a, int64_t a_stride,
                                                         int H,
                                                         float* __restrict__ W, float* __restrict__ Mo,
                                                         float* __restrict__ Vo, float lr_t,
                                                         const float* __restrict__ lr_dev, float b1, float b2, float eps,
                                                         unsigned short* __restrict__ Wf, unsigned short* __restrict__ WfT,
                                                         float* __restrict__ Wq, float* __restrict__ WqT,
                                                         float* __restrict__ grad_out) {
  constexpr int D = 64;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= H * D) return;
  const int n = e / D, j = e - n * D;
  float gb = 0.f, gd = 0.f;
  for (int s0 = 0; s0 < S; s0 += 8) {
    float pb[8], pd[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bool in = s0 + t < S;
      const float* p = slab + (int64_t) (in ? s0 + t : 0) * slab_stride + (int64_t) n * (2 * D) + j;
      pb[t] = in ? p[0] : 0.f;
      pd[t] = in ? p[D] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) { gb += pb[t]; gd += pd[t]; }
  }
  float ga = 0.f;
  for (int s0 = 0; s0 < n_a; s0 += 8) {
    float pa[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) pa[t] = s0 + t < n_a ? dWaT[(int64_t) (s0 + t) * a_stride + (int64_t) j * H + n] : 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) ga += pa[t];
  }
  const float g[4] = {ga, gb, ga - gb, gd};
  const float alpha = lr_dev ? lr_dev[0] : lr_t;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  float w[4];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    const int64_t at = (int64_t) n * (4 * D) + blk * D + j;
    if (grad_out) grad_out[at] = g[blk];
    float mi = Mo[at], vi = Vo[at];
    mi += (g[blk] - mi) * omb1;
    vi += (g[blk] * g[blk] - vi) * omb2;
    w[blk] = W[at] - (mi * alpha) / (sqrtf(vi) + eps);
    Mo[at] = mi; Vo[at] = vi; W[at] = w[blk];
  }
  const unsigned short u = mi_f2bf(w[1] - w[2]), d = mi_f2bf(w[3]);
  Wf[(int64_t) n * (2 * D) + j] = u;
  Wf[(int64_t) n * (2 * D) + D + j] = d;
  WfT[(int64_t) j * H + n] = u;
  WfT[(int64_t) (D + j) * H + n] = d;
  const float wq = w[0] + w[2];
  Wq[(int64_t) n * D + j] = wq;
  WqT[(int64_t) j * H + n] = wq;
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
                                int64_t ld_q, int a_cols, int32_t* err, void* stream) {
  MI_REQUIRE(E && hist && qid && len && A && ids_all, "mi_din_lau_build: null pointer");
  MI_REQUIRE(a_cols == 4 * DIN_D || a_cols == 2 * DIN_D, "mi_din_lau_build: a_cols must be 4*dim (full rows) or 2*dim ([k | q*k])");
  MI_REQUIRE(dim == DIN_D, "mi_din_lau_build: dim %d not built (64)", dim);
  MI_REQUIRE(B >= 1 && L >= 1 && B * L < ((int64_t) 1 << 31) && rows >= 1 && rows < ((int64_t) 1 << 31),
             "mi_din_lau_build: bad B/L/rows");
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_din_lau_build: idx_bits must be 32 or 64");
  MI_REQUIRE(((uintptr_t) E & 15) == 0 && ((uintptr_t) A & 7) == 0 && (!q_out || (((uintptr_t) q_out & 15) == 0 && ld_q % 4 == 0)),
             "mi_din_lau_build: misaligned buffer");
  const unsigned grid = (unsigned) mi_cdiv(B * L, 16);
  hipStream_t s = (hipStream_t) stream;
#define DIN_BUILD(IDX, FACT)                                                                                      \
  hipLaunchKernelGGL((din_build_kernel<IDX, FACT>), dim3(grid), dim3(256), 0, s, E, rows, hist, qid, len, B, L,  \
                     (unsigned short*) A, ids_all, q_out, ld_q, err)
  if (a_cols == 2 * DIN_D) { if (idx_bits == 32) DIN_BUILD(int32_t, true); else DIN_BUILD(int64_t, true); }
  else { if (idx_bits == 32) DIN_BUILD(int32_t, false); else DIN_BUILD(int64_t, false); }
#undef DIN_BUILD
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_din_lau_pool(const void* A, int a_cols, const float* a, const int32_t* ids_all, int64_t B, int L,
                               int dim, float* pooled, int64_t ld, void* stream) {
  MI_REQUIRE(A && a && ids_all && pooled && B >= 1 && L >= 1 && dim == DIN_D && ld % 4 == 0 &&
                 ((uintptr_t) pooled & 15) == 0 && (a_cols == 4 * DIN_D || a_cols == 2 * DIN_D),
             "mi_din_lau_pool: bad arguments");
  hipLaunchKernelGGL(din_pool_kernel, dim3((unsigned) mi_cdiv(B, 16)), dim3(256), 0, (hipStream_t) stream,
                     (const unsigned short*) A, a_cols, a_cols == 2 * DIN_D ? 0 : DIN_D, a, ids_all, B, L, pooled, ld);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_din_lau_pool_bwd(const void* A, int a_cols, const float* dpooled, int64_t ld, const int32_t* ids_all,
                                   int64_t B, int L, int dim, float* da, void* stream) {
  MI_REQUIRE(A && dpooled && ids_all && da && B >= 1 && L >= 1 && dim == DIN_D && ld % 4 == 0 &&
                 ((uintptr_t) dpooled & 15) == 0 && (a_cols == 4 * DIN_D || a_cols == 2 * DIN_D),
             "mi_din_lau_pool_bwd: bad arguments");
  hipLaunchKernelGGL(din_pool_bwd_kernel, dim3((unsigned) mi_cdiv(B * L, 16)), dim3(256), 0, (hipStream_t) stream,
                     (const unsigned short*) A, a_cols, a_cols == 2 * DIN_D ? 0 : DIN_D, dpooled, ld, ids_all, B, L, da);
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

extern "C" int mi_din_lau_scatter_fact(const void* A, const void* dA, const float* a, const float* dpooled, int64_t ld_dp,
                                       const float* dq_extra, int64_t ld_dq, const float* q, int64_t ld_q,
                                       const int32_t* ids_all, int64_t B, int L, int dim, float* dE, void* stream) {
  MI_REQUIRE(A && dA && a && dpooled && q && ids_all && dE && B >= 1 && L >= 1 && dim == DIN_D && ld_dp % 4 == 0 &&
                 ld_q % 4 == 0 && ((uintptr_t) dpooled & 15) == 0 && ((uintptr_t) q & 15) == 0 && ((uintptr_t) dE & 15) == 0 &&
                 (!dq_extra || (ld_dq % 4 == 0 && ((uintptr_t) dq_extra & 15) == 0)),
             "mi_din_lau_scatter_fact: bad arguments");
  hipLaunchKernelGGL(din_scatter_fact_kernel, dim3((unsigned) mi_cdiv(B, 16)), dim3(256), 0, (hipStream_t) stream,
                     (const unsigned short*) A, (const unsigned short*) dA, a, dpooled, ld_dp, dq_extra, ld_dq, q, ld_q,
                     ids_all, B, L, dE);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int64_t mi_din_qterm_part_floats(int64_t B, int H, int dim) {
  return mi_cdiv(B, QB_CHUNK) * ((int64_t) dim * H + H);
}

extern "C" int mi_din_qterm_fwd(const float* q, int64_t ld_q, const float* Wq, const float* b1, int64_t B, int H,
                                int dim, float* rowbias, void* stream) {
  MI_REQUIRE(q && Wq && b1 && rowbias && B >= 1 && dim == DIN_D && H >= 4 && H <= 128 && H % 4 == 0 && 256 % (H / 4) == 0 &&
                 ld_q % 4 == 0 && ((uintptr_t) q & 15) == 0 && ((uintptr_t) rowbias & 15) == 0,
             "mi_din_qterm_fwd: bad arguments (H <= 128, a power of two times 4)");
  const int per = H / 4;
  hipLaunchKernelGGL(din_qterm_fwd_kernel, dim3((unsigned) mi_cdiv(B, 256 / per)), dim3(256), 0, (hipStream_t) stream, q,
                     ld_q, Wq, b1, B, H, rowbias);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_din_qterm_bwd(const float* q, int64_t ld_q, const float* S, const float* Wq, int64_t B, int H, int dim,
                                const float* dq_in, int64_t ld_in, float* dq, int64_t ld_dq, float* part,
                                void* stream) {
  MI_REQUIRE(q && S && Wq && dq && part && B >= 1 && dim == DIN_D && H >= 32 && H <= 128 && H % 32 == 0,
             "mi_din_qterm_bwd: bad arguments (H a multiple of 32, <= 128)");
  hipLaunchKernelGGL(din_qterm_bwd_kernel, dim3((unsigned) mi_cdiv(B, QB_CHUNK)), dim3(256), 0, (hipStream_t) stream, q,
                     ld_q, S, Wq, B, H, dq, ld_dq, dq_in, ld_in, part);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_din_w1_step(const float* slab, int n_slabs, int64_t slab_stride, const float* dWaT, int n_a,
                              int64_t a_stride, int H, int dim,
                              float* W, float* m, float* v, float lr_t, const float* lr_dev, float beta1, float beta2,
                              float eps, void* Wf, void* WfT, float* Wq, float* WqT, float* grad_out, void* stream) {
  MI_REQUIRE(slab && dWaT && W && m && v && Wf && WfT && Wq && WqT && n_slabs >= 1 && H >= 1 && dim == DIN_D &&
                 slab_stride >= (int64_t) H * 2 * DIN_D && n_a >= 1 && a_stride >= (int64_t) H * DIN_D,
             "mi_din_w1_step: bad arguments");
  hipLaunchKernelGGL(din_w1_step_kernel, dim3((unsigned) mi_cdiv((int64_t) H * DIN_D, 256)), dim3(256), 0,
                     (hipStream_t) stream, slab, n_slabs, slab_stride, dWaT, n_a, a_stride, H, W, m, v, lr_t, lr_dev, beta1,
                     beta2, eps,
                     (unsigned short*) Wf, (unsigned short*) WfT, Wq, WqT, grad_out);
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
