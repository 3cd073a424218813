// Device-side bodies of the FM forward, the dense side of the step and the optimizer schedule,
// shared by the stand-alone launches and by the fused launches of the FM step.
#pragma once
#include "common.h"

struct FmFwdArgs {
  mi_fm_table tab;
  const int64_t* toffs;
  int T;
  const void* ids;
  int B;
  const float* dense;
  int n_dense;
  const float* wd;
  const float* bias;
  const float* labels;
  float gscale;
  float* logit;
  float* S;
  float* loss_rows;
  float* g;
  int32_t* err;
  int ids_tm;  // ids are stored table-major [T][B] (one array per field, as the reference feeds its features) instead of [B][T]
};

// Forward of the samples of virtual workgroup `vblock` of `nthreads` threads.
template <typename IDX, int DIM>
__device__ __forceinline__ void fm_fwd_body(const FmFwdArgs& a, int vblock, int nthreads) {
  const mi_fm_table& tab = a.tab;
  const int64_t* __restrict__ toffs = a.toffs;
  const void* __restrict__ ids = a.ids;
  const float* __restrict__ dense = a.dense;
  const float* __restrict__ wd = a.wd;
  const float* __restrict__ bias = a.bias;
  const float* __restrict__ labels = a.labels;
  float* __restrict__ logit = a.logit;
  float* __restrict__ S = a.S;
  float* __restrict__ loss_rows = a.loss_rows;
  float* __restrict__ g = a.g;
  int32_t* __restrict__ err = a.err;
  const int T = a.T, B = a.B, n_dense = a.n_dense;
  const float gscale = a.gscale;
  constexpr int CH = DIM / 4;                // lanes across one row
  constexpr int G = CH > 16 ? CH : 16;       // lanes per sample
  constexpr int FP = G / CH;                 // fields in flight per sample
  const int gid = vblock * (nthreads / G) + threadIdx.x / G;
  if (gid >= B) return;
  const int l = threadIdx.x % G;
  const int c = l % CH;
  const int q = l / CH;
  const float* __restrict__ rec = tab.rec;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
  float lin = 0.f;
  int bad = 0;
  // What the sample needs LAST is requested first: its dense features and their weights (lane l takes feature l),
  // its label and the bias.  Loaded where they are used, each of the two costs the sample a round trip of its own
  // behind the rows' (the label's alone 1.2 us of a 10 us forward).
  float d0 = 0.f, wd0 = 0.f, y0 = 0.f, bias0 = 0.f;
  if (n_dense > 0) {  // uniform
    const int j0 = l < n_dense ? l : 0;
    d0 = dense[(int64_t) gid * n_dense + j0];
    wd0 = wd[j0];
  }
  if (labels) y0 = labels[gid];  // uniform
  if (bias) bias0 = bias[0];     // uniform
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
      const int64_t id = mi_ld_id<IDX>(ids, a.ids_tm ? (int64_t) fs * B + gid : (int64_t) gid * T + fs);
      const int64_t base = toffs[fs];
      const int64_t next = toffs[fs + 1];
      // (bitwise &: with && the load of toffs[fs + 1] is only made when the id is present and non-negative,
      // i.e. behind a branch, and the compiler waits for all outstanding loads where that branch joins -
      // the eight fields of a lane became eight dependent round trips)
      ok[u] = present & (id >= 0) & (id < next - base);  // FM has no padding id: every field carries one id
      if (present && !ok[u]) bad |= MI_IDERR_RANGE;
      row[u] = ok[u] ? base + id : 0;
    }
    float4 v[UF];
    float w[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const float* r = rec + row[u] * tab.stride;
      v[u] = *reinterpret_cast<const float4*>(r + c * 4);
      w[u] = r[tab.off_w1];  // every lane of the row (same word: no extra traffic); only lane c == 0 adds it below
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const float4 x = make_float4(ok[u] ? v[u].x : 0.f, ok[u] ? v[u].y : 0.f, ok[u] ? v[u].z : 0.f,
                                   ok[u] ? v[u].w : 0.f);
      lin += (ok[u] && c == 0) ? w[u] : 0.f;
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
  if (l < n_dense) lin = fmaf(wd0, d0, lin);
  for (int j = l + G; j < n_dense; j += G) lin = fmaf(wd[j], dense[(int64_t) gid * n_dense + j], lin);
  float tot = pair + lin;
#pragma unroll
  for (int o = 1; o < G; o <<= 1) tot += __shfl_xor(tot, o, 64);
#pragma unroll
  for (int o = 1; o < G; o <<= 1) bad |= __shfl_xor(bad, o, 64);
  if (l == 0) {
    const float z = tot + bias0;
    if (logit) logit[gid] = z;
    if (labels) {
      const float y = y0;
      const float az = fabsf(z);
      const float e = expf(-az);
      if (loss_rows) loss_rows[gid] = fmaxf(z, 0.f) - z * y + log1pf(e);
      const float sig = z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      if (g) g[gid] = (sig - y) * gscale;
    }
    if (bad && err) atomicOr(err, bad);
  }
}

struct FmDenseArgs {
  const float* g;
  int64_t g_ld;
  const float* dense;
  int64_t dense_ld;
  int n_dense;
  int64_t B;
  float* grad_out;
  const float* loss_rows;
  int64_t loss_ld;
  float* dp;
  float* dm;
  float* dv;
  float lr_t;
  const float* lr_dev;
  float b1, b2, eps;
  float* loss_out;
};
constexpr int FM_DENSE_THREADS = 256;

// Virtual workgroup j < n_dense: dwd[j] = sum_b g[b]*dense[b][j]; j == n_dense: dbias = sum_b g[b];
// j == n_dense+1 (only when loss_rows != NULL): mean of the loss rows.  Fixed summation order
// (thread t takes b = t, t+256, ...; then a halving tree).  When dp != NULL the gradient is consumed
// in place by TensorFlow-form Adam on dp[j].  g, dense rows and loss rows are g_ld, dense_ld and
// loss_ld floats apart (1, n_dense, 1 for plain arrays).  All FM_DENSE_THREADS threads take part.
__device__ __forceinline__ void fm_dense_body(const FmDenseArgs& a, int j, float* sh /* FM_DENSE_THREADS floats */) {
  constexpr int NT = FM_DENSE_THREADS;
  // 16 samples per thread per round, all their loads requested before the first is used (a plain
  // loop is a chain of dependent round trips, 32 of them at batch 8192; this is two - more in flight
  // would cost the fused update launch, which this body shares registers with, a wave per SIMD); the
  // additions keep the order b = t, t+256, ... of that loop, so the sums are the same bits.
  constexpr int UR = 16;
  // what the closing optimizer step reads, requested now so that it does not cost a round trip of its own
  float pm = 0.f, pv = 0.f, pp = 0.f, plr = a.lr_t;
  if (threadIdx.x == 0 && j <= a.n_dense && a.dp) {
    pm = a.dm[j];
    pv = a.dv[j];
    pp = a.dp[j];
    if (a.lr_dev) plr = a.lr_dev[0];
  }
  // The mode is decided ONCE per workgroup and each loop body is free of branches around its loads: where
  // control flow joins behind a load the compiler waits for every outstanding load, which turned the 16
  // requests of a round into 16 dependent round trips.
  float acc = 0.f;
  if (j < a.n_dense) {
    const float* __restrict__ gp = a.g;
    const float* __restrict__ dp_ = a.dense + j;
    for (int64_t b0 = threadIdx.x; b0 < a.B; b0 += (int64_t) UR * NT) {
      float x[UR], y[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int64_t b = b0 + (int64_t) u * NT;
        const int64_t bs = b < a.B ? b : b0;
        x[u] = gp[bs * a.g_ld];
        y[u] = dp_[bs * a.dense_ld];
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) acc = (b0 + (int64_t) u * NT < a.B) ? fmaf(x[u], y[u], acc) : acc;
    }
  } else {
    const float* __restrict__ src = j == a.n_dense ? a.g : a.loss_rows;
    const int64_t ld = j == a.n_dense ? a.g_ld : a.loss_ld;
    for (int64_t b0 = threadIdx.x; b0 < a.B; b0 += (int64_t) UR * NT) {
      float x[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int64_t b = b0 + (int64_t) u * NT;
        x[u] = src[(b < a.B ? b : b0) * ld];
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) acc = (b0 + (int64_t) u * NT < a.B) ? acc + x[u] : acc;
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int) threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const float tot = sh[0];
  if (threadIdx.x != 0) return;
  if (j > a.n_dense) {
    a.loss_out[0] = tot / (float) a.B;
    return;
  }
  if (a.grad_out) a.grad_out[j] = tot;
  if (a.dp) {
    const float mi = pm + (tot - pm) * (1.0f - a.b1);
    const float vi = pv + (tot * tot - pv) * (1.0f - a.b2);
    a.dm[j] = mi;
    a.dv[j] = vi;
    a.dp[j] = pp - (mi * plr) / (sqrtf(vi) + a.eps);
  }
}

// The dense side of a sharded step.  Every rank's combine launch left, behind its B sample rows of
// the gathered matrix, one partial per workgroup: part[k][ct] = that workgroup's sums of
// g*dense[.][j] (j < n_dense), g and loss.  ONE workgroup adds the n_ranks*n_blocks partials in a
// fixed order (thread (r, j) takes partials r, r+rl, ...; then a tree over r) and applies
// TensorFlow-form Adam to dp = {wd, bias} / writes the mean loss: the same bits on every replica.
struct FmDenseFinalArgs {
  const float* X;    // gathered matrix: rank s owns rows [s*row_stride, (s+1)*row_stride)
  int64_t ld, row_stride, B;
  int n_ranks, n_blocks, ct, n_dense;
  float* dp;
  float* dm;
  float* dv;
  float lr_t;
  const float* lr_dev;
  float b1, b2, eps;
  float* loss_out;
};

__device__ __forceinline__ void fm_dense_final_body(const FmDenseFinalArgs& a, float* sh) {
  constexpr int NT = FM_DENSE_THREADS;
  const int ct = a.ct, rl = NT / ct;
  const int j = threadIdx.x % ct, rr = threadIdx.x / ct;
  const int total = a.n_ranks * a.n_blocks;
  float tot = 0.f;
  for (int i0 = rr; i0 < total; i0 += 8 * rl) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * rl;
      v[u] = 0.f;
      if (i < total) {
        const int s = i / a.n_blocks, k = i - s * a.n_blocks;
        v[u] = a.X[(s * a.row_stride + a.B) * a.ld + (int64_t) k * ct + j];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) tot += v[u];
  }
  sh[threadIdx.x] = tot;
  __syncthreads();
  for (int s = rl / 2; s > 0; s >>= 1) {
    if (rr < s) sh[threadIdx.x] += sh[threadIdx.x + s * ct];
    __syncthreads();
  }
  if ((int) threadIdx.x >= a.n_dense + 2) return;
  const int k = threadIdx.x;
  tot = sh[k];
  if (k == a.n_dense + 1) {
    a.loss_out[0] = tot / (float) (a.B * a.n_ranks);
    return;
  }
  const float mi = a.dm[k] + (tot - a.dm[k]) * (1.0f - a.b1);
  const float vi = a.dv[k] + (tot * tot - a.dv[k]) * (1.0f - a.b2);
  a.dm[k] = mi;
  a.dv[k] = vi;
  a.dp[k] -= (mi * (a.lr_dev ? a.lr_dev[0] : a.lr_t)) / (sqrtf(vi) + a.eps);
}

// sched = {alpha_t, beta1_power, beta2_power}: TensorFlow keeps the two powers as fp32 variables
// on the device and multiplies them by beta once per step; alpha_t is formed from them in fp32.
// One thread calls this.
__device__ __forceinline__ void adam_schedule_body(float* __restrict__ sched, float lr, float b1, float b2) {
  const float b1p = sched[1] * b1;
  const float b2p = sched[2] * b2;
  sched[1] = b1p;
  sched[2] = b2p;
  const float num = lr * sqrtf(1.0f - b2p);
  sched[0] = num / (1.0f - b1p);
}

