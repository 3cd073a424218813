// Sparse backward, step 2: per distinct row, sum the gradients of the entries that touched
// it (fixed order, no atomics) and apply the optimizer in the same kernel.
//
// The reference's gradient of a first-layer kernel is TensorFlow's dense X^T dY with
// one-hot X followed by tf.train.AdamOptimizer on every row
// (recommendation/code/model_trainer.py:157-158); `dense_grad` and `adam_dense_tf`
// reproduce exactly that, the other updaters touch only the rows present in the batch.
//
// Work split: a run (entries of one distinct row, contiguous after the sort) is served by
// dim/VEC adjacent lanes, each owning VEC columns.  A serial loop over a run is a chain of
// dependent loads (permutation entry -> gradient source), so a lane group sums only runs of up
// to GROUP_MAX entries itself (all loads of a run issued together); a longer run of up to
// WAVE_MAX entries is summed by the whole wave that met it, and longer ones by a workgroup: with
// run arrays in the plan they are queued to `long_list` for a second launch; straight from
// sorted_rows the workgroups of the same launch find them themselves, and a run of more than
// SPLIT_MIN entries is summed piecewise by the workgroups of the chunks it spans (partials and
// tickets in `long_list`).  Partial sums are combined in fixed trees and fixed position order, so
// hot rows (a table with three rows gets a third of the batch each) neither serialise on four
// lanes nor change bits from run to run.
//
// Runs come either from the compacted arrays of mi_embag_bwd_prepare (unique_rows /
// seg_offsets) or, when the plan carries none, straight from sorted_rows: one wave per 64
// sorted entries finds the heads of the runs that start there with a ballot.  The second form
// saves the three run-detection launches per step.
#pragma once
#include <type_traits>

#include "common.h"
#include "fm_body.h"

namespace {

#ifndef MI_WAVE_MAX
#define MI_WAVE_MAX 256
#endif
#ifndef MI_SPLIT_GATE
#define MI_SPLIT_GATE 2
#endif
#ifndef MI_SPLIT_MIN
#define MI_SPLIT_MIN (4 * 1024)
#endif
constexpr int WAVE_MAX = MI_WAVE_MAX;   // one wave sums runs up to this; longer runs are queued for a workgroup
constexpr int UBLOCK = 256;
constexpr int LONG_GRID = 512;

struct PlanDev {
  const uint32_t* sorted_rows;
  const uint32_t* unique_rows;
  const int32_t* perm;
  const int32_t* seg_offsets;
  const int32_t* counts;
  int32_t* long_list;  // [0] count, [1] finished-workgroup ticket, [2..] queued runs
  const int32_t* n_dev;  // when set, the number of entries is read from the device (n is then its upper bound)
  int split;             // long runs are summed piecewise by several workgroups (long_list has room for the partials)
  const int64_t* offsets;
  const void* indices;
  const int64_t* toffs;
  const float* bag_scale;     // optional: 1 / count of valid ids per bag (mean pooling)
  const int32_t* entry_bag;   // optional: bag of each position (CSR bags)
  uint32_t sentinel;
  int n, n_bags, bag_len, T, pool_mode, idx_bits;

  __device__ __forceinline__ int bag_of(int p) const {
    if (!offsets) return p / bag_len;
    if (entry_bag) {
      // The value is consumed (an empty asm that names it) before the paths join: a load still in flight where
      // they join would make the fixed-length path above, which loads nothing, wait there for every load the
      // caller has outstanding - and the callers keep several heads' gradient rows in flight.
      int e = entry_bag[p];
      asm volatile("" : "+v"(e));
      return e;
    }
    int lo = 0, hi = n_bags;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
  }
  __device__ __forceinline__ float mean_scale(int bag) const {
    if (pool_mode != MI_POOL_MEAN) return 1.0f;
    if (bag_scale) return bag_scale[bag];
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
  static constexpr bool kBatchSingles = true;  // light per-entry state: four single-entry runs per lane group at once
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

// gradient rows kept compactly, one per bag that has any: row_of_bag[bag] names the row (or -1: no
// gradient for this bag).  The form in which an owner of row-wise sharded tables receives the
// gradients of the bags it contributed to (see shard_bag.hip).
struct RowsGradSrc {
  static constexpr bool kBatchSingles = true;
  const float* rows;
  const int32_t* row_of_bag;
  int dim;
  template <int VEC>
  __device__ __forceinline__ void load(const PlanDev& pl, int p, int c, float* o, float& sc) const {
    const int r = row_of_bag[pl.bag_of(p)];
    ldv<VEC>(rows + (int64_t) (r >= 0 ? r : 0) * dim + c * VEC, o);
    if (r < 0) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) o[k] = 0.f;
    }
    sc = 0.f;
  }
};

// FM: entry of sample b contributes g[b]*S[b] (vector) and g[b] (scalar); bag_len == 1.
// The sample of entry p is p / T (a caller with another entry -> sample map sorts with T = 1 and makes the
// sample the sort's payload, as the sharded owner launch does: the entries an
// owner received from all ranks, each tagged with its requester's sample).
struct FmSrc {
  // (measured with two rows per round on the distinct-row waves of the FM step: no change, 24.4 vs 24.1 us)
  static constexpr bool kBatchSingles = false;
  const float* g;
  const float* S;
  int g_ld, S_ld;  // floats between consecutive samples (1 and dim when g and S are separate arrays)
  template <int VEC>
  __device__ __forceinline__ void load(const PlanDev& pl, int p, int c, float* o, float& sc) const {
    const int b = p / pl.T;
    const float gb = g[(int64_t) b * g_ld];
    ldv<VEC>(S + (int64_t) b * S_ld + c * VEC, o);
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
  const float* lr_dev;  // when set, the step size is read from device memory (a schedule kept on the device)
  __device__ __forceinline__ float step() const { return lr_dev ? *lr_dev : lr; }
};

// What an optimizer step reads from the row, loaded EARLY (as soon as the row is known) so
// that these loads fly together with the gradient gather instead of after it.
template <int OPT, int VEC>
struct OptRegs {
  float w[VEC], s0[VEC], s1[VEC];
};

// w, s0, s1 point at THIS row's data: w[dim]; s0 = Adagrad accumulator (1 float) or Adam m[dim];
// s1 = Adam v[dim].
template <int OPT, int VEC>
__device__ __forceinline__ void opt_load(const float* w, const float* s0, const float* s1, int c,
                                         OptRegs<OPT, VEC>& r) {
  ldv<VEC>(w + c * VEC, r.w);
  if constexpr (OPT == MI_OPT_ADAGRAD_ROWWISE) {
    r.s0[0] = s0[0];
  } else if constexpr (OPT == MI_OPT_ADAM_TF) {
    ldv<VEC>(s0 + c * VEC, r.s0);
    ldv<VEC>(s1 + c * VEC, r.s1);
  }
}

// a / b and sqrt(x) through the hardware's 1-ulp reciprocal and square root (v_rcp_f32, v_sqrt_f32): the
// correctly rounded forms the compiler emits otherwise are ~10 instructions each, and with four columns
// per lane and a row per lane group they were most of the kernel's vector instructions.  1 ulp is 6e-8
// relative on the step, far inside the 1e-5 bar; every path uses the same forms, so results stay
// bitwise reproducible and equal between launch variants.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

template <int OPT, int VEC>
__device__ __forceinline__ void opt_finish(float* w, float* s0, float* s1, int dim, int c, int lanes, const float* g,
                                           const OptArgs& a, OptRegs<OPT, VEC>& r) {
  const int off = c * VEC;
  if constexpr (OPT == MI_OPT_SGD) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) r.w[k] -= a.step() * g[k];
  } else if constexpr (OPT == MI_OPT_ADAGRAD_ROWWISE) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) ss += g[k] * g[k];
    ss = group_sum(ss, lanes);
    const float st = r.s0[0] + ss / (float) dim;
    const float inv = fast_div(a.step(), fast_sqrt(st) + a.eps);
#pragma unroll
    for (int k = 0; k < VEC; ++k) r.w[k] -= inv * g[k];
    if (c == 0) s0[0] = st;
  } else {  // MI_OPT_ADAM_TF
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      // the three statements of TensorFlow's ApplyAdam kernel, (1 - beta) formed in fp32 as there
      r.s0[k] += (g[k] - r.s0[k]) * (1.0f - a.beta1);
      r.s1[k] += (g[k] * g[k] - r.s1[k]) * (1.0f - a.beta2);
      r.w[k] -= fast_div(r.s0[k] * a.step(), fast_sqrt(r.s1[k]) + a.eps);
    }
    stv<VEC>(s0 + off, r.s0);
    stv<VEC>(s1 + off, r.s1);
  }
  stv<VEC>(w + off, r.w);
}

// ---- row updaters: load(row) early, finish(row, summed gradient) late -----------------------
template <int OPT>
struct BagUpd {
  // rows a lane group takes per round in waves of distinct rows (see seg_sorted_kernel): four pay for
  // SGD and Adagrad; Adam's two moment rows per row make the registers of four rows cost more
  // occupancy than the batching gains (measured), so it stays on the one-row path (0)
  static constexpr int kBatchSingles = OPT != MI_OPT_ADAM_TF ? 4 : 0;
  float* w; float* s0; float* s1; int dim; OptArgs a;
  template <int VEC> using Regs = OptRegs<OPT, VEC>;
  __device__ __forceinline__ float* p0(int64_t row) const {
    return s0 ? (OPT == MI_OPT_ADAGRAD_ROWWISE ? s0 + row : s0 + row * dim) : nullptr;
  }
  __device__ __forceinline__ float* p1(int64_t row) const { return s1 ? s1 + row * dim : nullptr; }
  template <int VEC>
  __device__ __forceinline__ void load(int64_t row, int c, Regs<VEC>& r) const {
    opt_load<OPT, VEC>(w + row * dim, p0(row), p1(row), c, r);
  }
  template <int VEC>
  __device__ __forceinline__ void finish(int64_t row, int c, int lanes, Regs<VEC>& r, const float* g, float) const {
    opt_finish<OPT, VEC>(w + row * dim, p0(row), p1(row), dim, c, lanes, g, a, r);
  }
};

template <class U, class = void>
struct rows16_only : std::false_type {};
template <class U>
struct rows16_only<U, std::void_t<decltype(U::kRows16)>> : std::true_type {};

struct NoRegs {};
struct DenseGradUpd {
  static constexpr int kBatchSingles = 4;
  float* dW; int dim;
  template <int VEC> using Regs = NoRegs;
  template <int VEC>
  __device__ __forceinline__ void load(int64_t, int, NoRegs&) const {}
  template <int VEC>
  __device__ __forceinline__ void finish(int64_t row, int c, int, NoRegs&, const float* g, float) const {
    stv<VEC>(dW + row * dim + c * VEC, g);
  }
};

// FM row record: V, w1 and their optimizer state side by side (see mi_fm_table)
template <int OPT, int VEC>
struct FmRegs {
  OptRegs<OPT, VEC> v;
  OptRegs<OPT, 1> w;
};
template <int OPT>
struct FmUpd {
  static constexpr int kBatchSingles = 0;
  mi_fm_table t; OptArgs a;
  template <int VEC> using Regs = FmRegs<OPT, VEC>;
  template <int VEC>
  __device__ __forceinline__ void load(int64_t row, int c, Regs<VEC>& r) const {
    const float* rec = t.rec + row * t.stride;
    opt_load<OPT, VEC>(rec, rec + t.off_vs0, rec + t.off_vs1, c, r.v);
    // every lane loads the first-order weight and its state (the lanes of a row read the same words: no extra
    // traffic); a branch for lane 0 alone would make the compiler wait for all outstanding loads behind it
    opt_load<OPT, 1>(rec + t.off_w1, rec + t.off_ws0, rec + t.off_ws1, 0, r.w);
  }
  template <int VEC>
  __device__ __forceinline__ void finish(int64_t row, int c, int lanes, Regs<VEC>& r, const float* acc,
                                         float gsum) const {
    float* rec = t.rec + row * t.stride;
    float gr[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) gr[k] = acc[k] - r.v.w[k] * gsum;
    opt_finish<OPT, VEC>(rec, rec + t.off_vs0, rec + t.off_vs1, t.dim, c, lanes, gr, a, r.v);
    if (c == 0) opt_finish<OPT, 1>(rec + t.off_w1, rec + t.off_ws0, rec + t.off_ws1, 1, 0, 1, &gsum, a, r.w);
  }
};

// The same for records in the standard dim-16 layout (V[16] | w1, its state, padding to 128 bytes | Adam: m[16] |
// v[16]): every access is a 16-byte access of each of the row's four lanes to one 64-byte half of a line - lane c
// takes piece c of every half, so the half that holds the three scalars is read and written whole, padding
// included, by ONE instruction instead of three 4-byte loads and three 4-byte stores of lane 0.  Measured on
// 94 k sorted distinct records (a batch of 8192): read-modify-write with the scalars as single words 14.8-15.7 us,
// without them 13.0-13.2 us, with every half written whole 13.3 us, with the scalars' half written in part 14.7 us.
// The scalars live in lane 0 (piece 4); the formulas are those of opt_finish, statement for statement.
template <int OPT>
struct FmPackRegs {
  float V[4], sc[4], m[4], v[4];
};
__device__ __forceinline__ float quad_lane0(float x) {  // lane 0's value in every lane of a group of four
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x00, 0xf, 0xf, true));  // quad_perm:[0,0,0,0]
}
template <int OPT>
struct FmPackUpd {
  static constexpr int kBatchSingles = 0;
  static constexpr bool kRows16 = true;  // only the kernels for rows of 16 floats (four lanes of four) are built
  mi_fm_table t; OptArgs a;
  template <int VEC> using Regs = FmPackRegs<OPT>;
  template <int VEC>
  __device__ __forceinline__ void load(int64_t row, int c, Regs<VEC>& r) const {
    static_assert(VEC == 4, "packed FM records are rows of 16 floats");
    const float* __restrict__ rec = t.rec + row * t.stride + c * 4;
    ldv<4>(rec, r.V);
    ldv<4>(rec + 16, r.sc);
    if constexpr (OPT == MI_OPT_ADAM_TF) {
      ldv<4>(rec + 32, r.m);
      ldv<4>(rec + 48, r.v);
    }
  }
  template <int VEC>
  __device__ __forceinline__ void finish(int64_t row, int c, int, Regs<VEC>& r, const float* acc, float gsum) const {
    float* __restrict__ rec = t.rec + row * t.stride + c * 4;
    const float st = a.step();
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = acc[k] - r.V[k] * gsum;
    float s0 = r.sc[0], s1 = r.sc[1], s2 = r.sc[2];  // lane 0: w1 and its state
    if constexpr (OPT == MI_OPT_SGD) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r.V[k] -= st * g[k];
      s0 -= st * gsum;
    } else if constexpr (OPT == MI_OPT_ADAGRAD_ROWWISE) {
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) ss += g[k] * g[k];
      ss = group_sum(ss, 4);
      const float sv = quad_lane0(r.sc[1]) + ss / 16.0f;  // the row's accumulator: lane 0's second scalar
      const float inv = fast_div(st, fast_sqrt(sv) + a.eps);
#pragma unroll
      for (int k = 0; k < 4; ++k) r.V[k] -= inv * g[k];
      const float sw = s2 + gsum * gsum / 1.0f;
      const float invw = fast_div(st, fast_sqrt(sw) + a.eps);
      s0 -= invw * gsum;
      s1 = sv;
      s2 = sw;
    } else {
      const float c1 = 1.0f - a.beta1, c2 = 1.0f - a.beta2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r.m[k] += (g[k] - r.m[k]) * c1;
        r.v[k] += (g[k] * g[k] - r.v[k]) * c2;
        r.V[k] -= fast_div(r.m[k] * st, fast_sqrt(r.v[k]) + a.eps);
      }
      s1 += (gsum - s1) * c1;
      s2 += (gsum * gsum - s2) * c2;
      s0 -= fast_div(s1 * st, fast_sqrt(s2) + a.eps);
      stv<4>(rec + 32, r.m);
      stv<4>(rec + 48, r.v);
    }
    stv<4>(rec, r.V);
    // lanes 1-3 hold padding: written back as read, so the half goes out whole
    const float o[4] = {c == 0 ? s0 : r.sc[0], c == 0 ? s1 : r.sc[1], c == 0 ? s2 : r.sc[2], r.sc[3]};
    stv<4>(rec + 16, o);
  }
};

// the layouts FmPackUpd is built for (ops.FmRecords / mi_fm_table with dim 16)
inline bool fm_pack_layout_ok(const mi_fm_table& t, int opt) {
  if (t.dim != 16 || t.off_w1 != 16 || ((uintptr_t) t.rec & 63) != 0) return false;
  if (opt == MI_OPT_ADAM_TF)
    return t.stride == 64 && t.off_ws0 == 17 && t.off_ws1 == 18 && t.off_vs0 == 32 && t.off_vs1 == 48;
  if (opt == MI_OPT_ADAGRAD_ROWWISE) return t.stride == 32 && t.off_vs0 == 17 && t.off_ws0 == 18;
  return opt == MI_OPT_SGD && t.stride == 32;
}

// ---- kernels ------------------------------------------------------------------------------
// Sum of the gradient sources of entries beg, beg+step, ... < end (fixed order).  Rounds of
// eight: the permutation entries of round r+1 are requested before the sources of round r are
// consumed, so a round costs one dependent memory round trip, not two.
#ifndef MI_SUM_UR
#define MI_SUM_UR 8
#endif
#ifndef MI_LONG_UR
#define MI_LONG_UR 8
#endif
template <int VEC, class Src, int UR = MI_SUM_UR>
__device__ __forceinline__ void sum_run(const PlanDev& pl, const Src& src, int c, int beg, int end, int step,
                                        float* acc, float& sc) {
  if (beg >= end) return;
  int p[UR];
  bool in[UR];
#pragma unroll
  for (int i = 0; i < UR; ++i) {
    in[i] = beg + i * step < end;
    p[i] = pl.perm[in[i] ? beg + i * step : beg];
  }
  for (int e = beg; e < end; e += UR * step) {
    float v[UR][VEC], s[UR];
#pragma unroll
    for (int i = 0; i < UR; ++i) src.template load<VEC>(pl, p[i], c, v[i], s[i]);
    bool cur[UR];
#pragma unroll
    for (int i = 0; i < UR; ++i) cur[i] = in[i];
    const int nx = e + UR * step;
    if (nx < end) {
#pragma unroll
      for (int i = 0; i < UR; ++i) {
        in[i] = nx + i * step < end;
        p[i] = pl.perm[in[i] ? nx + i * step : nx];
      }
    }
#pragma unroll
    for (int i = 0; i < UR; ++i) {
      if (cur[i]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += v[i][k];
        sc += s[i];
      }
    }
  }
}

__device__ __forceinline__ void queue_long(const PlanDev& pl, int token) {
  const int slot = atomicAdd(&pl.long_list[0], 1);
  pl.long_list[2 + slot] = token;
}

// End of the run that starts at `beg` (first index whose row is greater; rows ascend and the
// sentinel is the largest), found by one wave: first a 64 x 16 window (runs up to 1024 take two
// dependent loads), then a 64-way search of whatever is left (3 more for 2^18 entries).
__device__ __forceinline__ int run_end_wave(const uint32_t* keys, int n, int beg, uint32_t row) {
  const int lane = threadIdx.x & 63;
  int lo = beg + 1, hi = n;  // answer in [lo, hi]
  bool first = true;
#ifdef MI_WIDE_SEARCH
  if (hi - lo > 1024) {
    // first look 4096 entries ahead in ONE round trip: lane l probes entries lo + (l + 64 j) * 16, j = 0..3
    uint32_t k[4];
    bool gt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = lo + (lane + 64 * j) * 16;
      k[j] = keys[idx < hi ? idx : lo];
      gt[j] = idx >= hi;
    }
    int P = 256;  // first probe (in order j-major) beyond the run; uniform
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const unsigned long long m = __ballot(gt[j] || k[j] > row);
      if (m) P = 64 * j + __ffsll((long long) m) - 1;
    }
    const int new_hi = P < 256 ? min(hi, lo + P * 16) : hi;
    const int new_lo = P > 0 ? lo + (P - 1) * 16 + 1 : lo;
    lo = new_lo;
    hi = new_hi;
    first = false;
  }
#endif
  while (lo < hi) {
    const int span = hi - lo;
    int step = (span + 63) / 64;
    if (first && step > 16) step = 16;  // first look only 1024 entries ahead
    first = false;
    const int idx = lo + lane * step;
    const bool gt = idx >= hi || keys[idx] > row;
    const unsigned long long m = __ballot(gt);   // monotone: 0..0 1..1
    const int f = m ? __ffsll((long long) m) - 1 : 64;
    const int new_hi = f < 64 ? min(hi, lo + f * step) : hi;
    const int new_lo = f > 0 ? lo + (f - 1) * step + 1 : lo;
    lo = new_lo;
    hi = new_hi;
  }
  return lo;
}

// Start of the run of `row` that contains entry `at` (first index of the contiguous stretch of
// entries equal to row that ends at `at`), found by one wave the same way, looking backwards.
__device__ __forceinline__ int run_begin_wave(const uint32_t* keys, int at, uint32_t row) {
  const int lane = threadIdx.x & 63;
  int lo = 0, hi = at;  // answer in [lo, hi]: keys[hi] == row
  bool first = true;
  while (lo < hi) {
    const int span = hi - lo;
    int step = (span + 63) / 64;
    if (first && step > 16) step = 16;
    first = false;
    const int idx = hi - (lane + 1) * step;  // lane 0 nearest to hi
    const bool ne = idx < lo || keys[idx] != row;
    const unsigned long long m = __ballot(ne);   // monotone in the lane: 0..0 1..1
    const int f = m ? __ffsll((long long) m) - 1 : 64;
    // lanes < f are still inside the run; lane f (if any) is outside
    const int new_hi = f > 0 ? hi - f * step : hi;
    const int new_lo = f < 64 ? max(lo, hi - (f + 1) * step + 1) : lo;
    lo = new_lo;
    hi = new_hi;
  }
  return lo;
}

// One wave sums the run [beg, end) of `row`: its 64/lanes lane groups stride over the entries,
// then a fixed shuffle tree combines them and group 0 applies the update.  Every lane of the
// wave must call this together.  Needs lanes to be a power of two <= 64.
template <int VEC, class Src, class Upd>
__device__ __forceinline__ void wave_run(const PlanDev& pl, const Src& src, const Upd& upd, int lanes, int beg,
                                         int end, uint32_t row) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / lanes, c = lane - sub * lanes, wgroups = 64 / lanes;
  typename Upd::template Regs<VEC> regs;
  if (sub == 0) upd.template load<VEC>((int64_t) row, c, regs);
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  float sc = 0.f;
  sum_run<VEC, Src, MI_LONG_UR>(pl, src, c, beg + sub, end, wgroups, acc, sc);
  for (int o = lanes; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
    sc += __shfl_xor(sc, o, 64);
  }
  if (sub == 0) upd.template finish<VEC>((int64_t) row, c, lanes, regs, acc, sc);
}

// A lane group sums a run of up to 4 / up to 16 entries: all permutation entries in one round
// trip, all gradient sources in the next, added in entry order.
template <int VEC, class Src>
__device__ __forceinline__ void group_sum4(const PlanDev& pl, const Src& src, int c, int beg, int len, float* acc,
                                           float& sc) {
  int p[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = pl.perm[q < len ? beg + q : beg];
  float v[4][VEC], s[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) src.template load<VEC>(pl, p[q], c, v[q], s[q]);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < len) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += v[q][k];
      sc += s[q];
    }
  }
}

template <int VEC, class Src>
__device__ __forceinline__ void group_sum16(const PlanDev& pl, const Src& src, int c, int beg, int len, float* acc,
                                            float& sc) {
  int p[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) p[q] = pl.perm[q < len ? beg + q : beg];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float v[8][VEC], s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) src.template load<VEC>(pl, p[h * 8 + q], c, v[q], s[q]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (h * 8 + q < len) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += v[q][k];
        sc += s[q];
      }
    }
  }
}

// A lane group owns the run [beg, beg+len) of `row`, len <= 16: sum and apply.
template <int VEC, class Src, class Upd>
__device__ __forceinline__ void group_run(const PlanDev& pl, const Src& src, const Upd& upd, int lanes, int c,
                                          int beg, int len, uint32_t row) {
  typename Upd::template Regs<VEC> regs;
  upd.template load<VEC>((int64_t) row, c, regs);
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  float sc = 0.f;
  if (len <= 4) group_sum4<VEC>(pl, src, c, beg, len, acc, sc);
  else group_sum16<VEC>(pl, src, c, beg, len, acc, sc);
  upd.template finish<VEC>((int64_t) row, c, lanes, regs, acc, sc);
}

// The same for a run of up to 4 entries whose permutation entries the caller already holds (p[q] for
// q < len; the others may be anything valid): the row's state and the gradient sources are requested
// together, one memory round trip instead of two.
template <int VEC, class Src, class Upd>
__device__ __forceinline__ void group_run4_pre(const PlanDev& pl, const Src& src, const Upd& upd, int lanes, int c,
                                               const int* p, int len, uint32_t row) {
  typename Upd::template Regs<VEC> regs;
  upd.template load<VEC>((int64_t) row, c, regs);
  float v[4][VEC], s[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) src.template load<VEC>(pl, p[q], c, v[q], s[q]);
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  float sc = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < len) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += v[q][k];
      sc += s[q];
    }
  }
  upd.template finish<VEC>((int64_t) row, c, lanes, regs, acc, sc);
}

constexpr int GROUP_MAX = 16;  // a lane group sums runs up to this itself

// The runs some lane groups of a wave could not take (more than GROUP_MAX entries, or end not
// known yet: end < 0) are summed by the whole wave, one after the other; only runs beyond
// WAVE_MAX go to the queue.  Every lane of the wave must call this together.
template <int VEC, class Src, class Upd>
__device__ __forceinline__ void wave_pending(const PlanDev& pl, const Src& src, const Upd& upd, int lanes,
                                             bool pending, int beg, int end, uint32_t row, int token,
                                             bool use_queue = true) {
  unsigned long long todo = __ballot(pending);
  while (todo) {  // uniform over the wave
    const int from = __ffsll((long long) todo) - 1;
    todo &= todo - 1;
    const int rb = __shfl(beg, from, 64);
    int re = __shfl(end, from, 64);
    const uint32_t rr = (uint32_t) __shfl((int) row, from, 64);
    const int tk = __shfl(token, from, 64);
    if (re < 0) re = run_end_wave(pl.sorted_rows, pl.n, rb, rr);
    if (re - rb > WAVE_MAX) {  // a workgroup's job: queued, or found by the long-run workgroups themselves
      if (use_queue && (threadIdx.x & 63) == 0) queue_long(pl, tk);
      continue;
    }
    wave_run<VEC>(pl, src, upd, lanes, rb, re, rr);
  }
}

// Runs from the compacted arrays of the plan: one lane group per distinct row.
template <int VEC, class Src, class Upd>
__global__ __launch_bounds__(UBLOCK) void seg_runs_kernel(PlanDev pl, Src src, Upd upd, int lanes, int gpb,
                                                         int wave_ok) {
  const int sub = threadIdx.x / lanes;
  const int c = threadIdx.x - sub * lanes;
  const int item = blockIdx.x * gpb + sub;
  int beg = 0, end = 0;
  uint32_t row = 0;
  bool pending = false;
  if (sub < gpb && item < pl.counts[0]) {
    beg = pl.seg_offsets[item];
    end = pl.seg_offsets[item + 1];
    row = pl.unique_rows[item];
    if (end - beg <= GROUP_MAX) group_run<VEC>(pl, src, upd, lanes, c, beg, end - beg, row);
    else pending = (c == 0);
  }
  if (!wave_ok) {  // lane groups straddle waves: longer runs all go to the queue
    if (pending) queue_long(pl, item);
    return;
  }
  wave_pending<VEC>(pl, src, upd, lanes, pending, beg, end, row, item);
}

// position of the k-th (0-based) set bit of m; k < popcount(m)
__device__ __forceinline__ int kth_set_bit(unsigned long long m, int k) {
  int pos = 0;
#pragma unroll
  for (int b = 32; b > 0; b >>= 1) {
    const unsigned long long prefix = m & ((1ull << (pos + b)) - 1ull);  // bits [0, pos+b), pos+b <= 63
    if (__popcll(prefix) <= k) pos += b;
  }
  return pos;
}

// One workgroup sums the entries [beg, end): its lane groups stride over them and combine partial
// sums through a fixed LDS tree, leaving the totals in red[c*VEC + k] / reds[c] (c < lanes).  The
// order of additions depends only on the number of entries: reproducible bit for bit.  All threads
// call this; the totals are visible to all on return.
template <int VEC, class Src>
__device__ __forceinline__ void block_sum(const PlanDev& pl, const Src& src, int lanes, int groups, int beg, int end,
                                          float* red, float* reds) {
  const int sub = threadIdx.x / lanes, c = threadIdx.x - sub * lanes;
  const bool active = sub < groups;
  int top = 1;
  while (top < groups) top <<= 1;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  float sc = 0.f;
  if (active) {
    sum_run<VEC, Src, MI_LONG_UR>(pl, src, c, beg + sub, end, groups, acc, sc);
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
}

// One workgroup sums the run of `row` that starts at `beg` (end looked up unless given) and applies
// the update.  All threads call this.
template <int VEC, class Src, class Upd>
__device__ __forceinline__ void block_run(const PlanDev& pl, const Src& src, const Upd& upd, int lanes, int groups,
                                          int beg, int end, uint32_t row, float* red, float* reds, int* sh_end) {
  const int sub = threadIdx.x / lanes, c = threadIdx.x - sub * lanes;
  if (end < 0) {
    if (threadIdx.x < 64) {
      const int e = run_end_wave(pl.sorted_rows, pl.n, beg, row);
      if (threadIdx.x == 0) *sh_end = e;
    }
    __syncthreads();
    end = *sh_end;
  }
  typename Upd::template Regs<VEC> regs;
  if (sub == 0) upd.template load<VEC>((int64_t) row, c, regs);
  block_sum<VEC>(pl, src, lanes, groups, beg, end, red, reds);
  if (sub == 0) {
    float g[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) g[k] = red[c * VEC + k];
    upd.template finish<VEC>((int64_t) row, c, lanes, regs, g, reds[c]);
  }
  __syncthreads();
}

// Runs straight from sorted_rows (no run arrays in the plan), ONE launch for every run length.
//
// Workgroups [long_blocks, ...): one WAVE per 64 consecutive sorted entries.  Lane l holds entry
// l's row; a ballot of "row differs from the predecessor's" gives the heads of the runs that
// start in this wave and, from the distance to the next head, their lengths - one coalesced
// load instead of per-run probing.  The wave's 64/lanes lane groups then take the heads 64/lanes
// at a time.  A run that continues past the wave's last entry, or is longer than GROUP_MAX, is
// left to the whole wave (wave_pending); one longer than WAVE_MAX is skipped here.
//
// Workgroups [0, long_blocks): each scans LONG_CHUNK entries for heads of runs longer than
// WAVE_MAX (entry i heads one iff its row differs from entry i-1's and equals entry
// i+WAVE_MAX's) and sums those with all its threads.  They are dispatched first and work beside
// the short runs instead of after them; nothing is queued, no atomics on global memory.
// Narrow rows (LANES = dim / VEC <= 4 lanes per row): one wave's 64 sorted entries with EVERYTHING
// requested up front.  After the wave's rows and permutation entries (one coalesced round trip) the
// heads are ranked by a ballot, every lane group learns the (up to LANES) runs it owns from a
// wave-private LDS table, and requests in one go the optimizer state of those rows and the gradient
// sources of its LANES entries (entry e is fetched by group e % G whoever owns its run).  The
// sources pass through wave-private LDS, where the owner of a run adds its entries in sorted order -
// the order group_sum4 / group_sum16 use, so runs of up to 16 entries keep their bits - and applies
// the update.  Two dependent memory round trips per wave however the 64 entries split into runs,
// and no per-round search for the k-th head.  The one run that may continue past the wave's last
// entry is left to wave_pending as before.
// cs: (64 + 64/LANES) * (LANES*VEC + 4) floats, ss: 64 + 64/LANES floats, hp: 65 ints - this wave's own.
template <int VEC, int LANES, class Src, class Upd>
__device__ __forceinline__ void short_wave_lds(const PlanDev& pl, const Src& src, const Upd& upd, int base,
                                               uint32_t key, int perm_reg, uint32_t key2, int perm2,
                                               unsigned long long hm, uint32_t sentinel, float* __restrict__ cs,
                                               float* __restrict__ ss, int* __restrict__ hp) {
  constexpr int G = 64 / LANES, R = LANES, STR = LANES * VEC + 4;
  const int lane = threadIdx.x & 63, g = lane / LANES, c = lane - g * LANES;
  const int nheads = __popcll(hm);
  if (nheads == 0) return;  // all 64 entries continue a run that started earlier: its opener's job (uniform)
  // The wave's last run may go on in the following entries (key2 / perm2: the next 64 rows and permutation
  // entries, requested with the wave's own; a row past the end of the array reads as the sentinel).  Up to G
  // more entries are taken along - one more source per lane group, in the same round trip as the others -
  // so the run costs its owner nothing extra; only a longer continuation is left to wave_pending, whose
  // search for the end and second round of loads start when everything else is done.
  const uint32_t last_row = (uint32_t) __shfl((int) key, 63, 64);
  const unsigned long long differs = __ballot(key2 != last_row);
  const int ext_all = last_row == sentinel ? 0 : (differs ? __ffsll((long long) differs) - 1 : 64);  // uniform
  const int ext = ext_all <= G ? ext_all : 0;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  if ((hm >> lane) & 1ull) hp[__popcll(hm & lt)] = lane;
  if (lane == 0) hp[nheads] = 64;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int pos[R], len[R];
  uint32_t row[R];
  bool have[R];
  bool pending = false;
  int pend_beg = 0;
  uint32_t pend_row = 0;
  typename Upd::template Regs<VEC> regs[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int k = q * G + g;
    const bool in = k < nheads;
    pos[q] = in ? hp[k] : 0;
    len[q] = in ? hp[k + 1] - pos[q] : 0;
    row[q] = (uint32_t) __shfl((int) key, pos[q], 64);
    have[q] = in && row[q] != sentinel;
    if (have[q] && k == nheads - 1) {  // the wave's last run
      if (ext_all > G) {               // goes on for more entries than are taken along
        have[q] = false;
        pending = (c == 0);
        pend_beg = base + pos[q];
        pend_row = row[q];
      } else {
        len[q] += ext;  // its continuation is staged behind the wave's own 64 entries
      }
    }
    // unconditional (a branch around a load makes the compiler wait for all outstanding loads where the
    // paths join): a slot without a run reads row 0, which exists
    upd.template load<VEC>(have[q] ? (int64_t) row[q] : 0, c, regs[q]);
  }
  float v[R][VEC], sv[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int p = __shfl(perm_reg, r * G + g, 64);
    src.template load<VEC>(pl, p, c, v[r], sv[r]);
  }
  if (ext > 0) {  // uniform; joins where every load above is awaited anyway
    float v2[VEC], s2;
    src.template load<VEC>(pl, __shfl(perm2, g < ext ? g : 0, 64), c, v2, s2);
    if (g < ext) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) cs[(64 + g) * STR + c * VEC + k] = v2[k];
      if (c == 0) ss[64 + g] = s2;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = r * G + g;
#pragma unroll
    for (int k = 0; k < VEC; ++k) cs[e * STR + c * VEC + k] = v[r][k];
    if (c == 0) ss[e] = sv[r];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if (have[q]) {
      float acc[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
      float sc = 0.f;
      for (int j = 0; j < len[q]; ++j) {
        const int e = pos[q] + j;
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += cs[e * STR + c * VEC + k];
        sc += ss[e];
      }
      upd.template finish<VEC>((int64_t) row[q], c, LANES, regs[q], acc, sc);
    }
  }
  wave_pending<VEC>(pl, src, upd, LANES, pending, pend_beg, -1, pend_row, pend_beg, false);
}

#ifndef MI_LONG_CHUNK
#define MI_LONG_CHUNK 1024
#endif
constexpr int LONG_CHUNK = MI_LONG_CHUNK;
constexpr int SPLIT_MIN = MI_SPLIT_MIN;  // runs longer than this are summed piecewise by several workgroups
#ifndef MI_WAVE_RUN_MAX
#define MI_WAVE_RUN_MAX 1024
#endif
constexpr int WAVE_RUN_MAX = MI_WAVE_RUN_MAX;  // a long-run workgroup leaves runs up to this to single waves
constexpr int LONG_CAP = LONG_CHUNK / (WAVE_MAX + 1) + 2;  // heads of long runs are more than WAVE_MAX apart

//
// Workgroups before those (`extra.blocks` of them, possibly none) do unrelated work that is ready
// at the same time - the dense side of an FM step - so that it shares the launch instead of
// waiting for its own.
struct NoExtra {
  int blocks;
  __device__ __forceinline__ void run(int, float*) const {}
};
struct FmDenseExtra {
  int blocks;
  FmDenseArgs a;
  __device__ __forceinline__ void run(int j, float* sh) const { fm_dense_body(a, j, sh); }
};
// The dense side of an FM step (workgroups [0, dense_blocks)) and, behind it, workgroups that copy the NEXT batch's
// block from pinned host memory into the device buffers the next step reads (n 16-byte words; the reads cross PCIe, a
// round trip of microseconds: eight per thread in flight).  The batch then arrives through the step's own launch -
// no second stream, no event - while the update keeps the memory system busy.
struct FmDenseStageExtra {
  int blocks;
  FmDenseArgs a;
  int dense_blocks;
  float4* dst;
  const float4* src;
  size_t n;
  __device__ __forceinline__ void run(int j, float* sh) const {
    if (j < dense_blocks) {  // uniform
      fm_dense_body(a, j, sh);
      return;
    }
    const size_t stride = (size_t) (blocks - dense_blocks) * UBLOCK;
    for (size_t i = (size_t) (j - dense_blocks) * UBLOCK + threadIdx.x; i < n; i += 8 * stride) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t k = i + u * stride;
        v[u] = src[k < n ? k : i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t k = i + u * stride;
        if (k < n) dst[k] = v[u];
      }
    }
  }
};
struct FmDenseFinalExtra {
  int blocks;  // 1
  FmDenseFinalArgs a;
  __device__ __forceinline__ void run(int, float* sh) const { fm_dense_final_body(a, sh); }
};
static_assert(FM_DENSE_THREADS == UBLOCK, "the dense side runs in workgroups of the sparse update's size");

#ifdef SEG_TRACE
static __device__ unsigned long long seg_trace_buf[4096 * 2];  // one copy per translation unit
static __device__ unsigned long long seg_phase_buf[4096 * 8];  // phase stamps of the long-run workgroups
#define SEG_PHASE(k)                                                                                  \
  do {                                                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                       \
    if (threadIdx.x == 0 && blockIdx.x < 4096) seg_phase_buf[blockIdx.x * 8 + (k)] = wall_clock64();  \
  } while (0)
#else
#define SEG_PHASE(k)
#endif

// LP: lanes per row when that is 1, 2 or 4 (rows of up to 16 floats: only the LDS path for the short runs
// is compiled, which keeps the kernel under 128 vector registers - four waves per SIMD instead of
// three), 0 for any other width (lanes is then read from the argument).
template <int VEC, bool SPLIT, int LP, class Src, class Upd, class Extra>
__device__ __forceinline__ void seg_sorted_body(PlanDev pl, Src src, Upd upd, int lanes_arg, int groups,
                                                int long_blocks, Extra extra, int epw_arg) {
  const int lanes = LP > 0 ? LP : lanes_arg;
  // entries per wave: 64, or fewer (wide rows, few entries) so that a small batch still spreads over
  // the chip - a wave of 64 entries with two lane groups is 8 or more rounds of dependent loads
  const int epw = LP > 0 ? 64 : epw_arg;
  if (pl.n_dev) pl.n = min(pl.n, pl.n_dev[0]);
  const uint32_t* keys = pl.sorted_rows;
  const uint32_t sentinel = pl.sentinel;
#ifdef MI_EXTRA_LAST
  const int extra_first = (int) gridDim.x - extra.blocks;
  if ((int) blockIdx.x >= extra_first) {
    __shared__ float sh[UBLOCK];
    extra.run(blockIdx.x - extra_first, sh);
    return;
  }
  const int bid = blockIdx.x;
#else
  if ((int) blockIdx.x < extra.blocks) {
    __shared__ float sh[UBLOCK];
    extra.run(blockIdx.x, sh);
    return;
  }
  const int bid = blockIdx.x - extra.blocks;
#endif
  if (bid < long_blocks) {
    __shared__ float red[UBLOCK * VEC];
    __shared__ float reds[UBLOCK];
    __shared__ int sh_end, sh_n, sh_heads[LONG_CAP], sh_span[2], sh_last;
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const int lo = bid * LONG_CHUNK;
    if (lo >= pl.n) return;  // uniform
    const int hi = lo + LONG_CHUNK < pl.n ? lo + LONG_CHUNK : pl.n;
    // what decides whether a run of more than SPLIT_MIN entries enters from the left (it then covers
    // lo - SPLIT_MIN/2 or lo + SPLIT_MIN/2), requested now so that the answer is there when needed
    uint32_t k_lo = 0, k_prev = 1, k_far_l = 1, k_far_r = 1;
    if (SPLIT) {
      k_lo = keys[lo];
      k_prev = lo > 0 ? keys[lo - 1] : ~k_lo;
      k_far_l = lo >= SPLIT_MIN / 2 ? keys[lo - SPLIT_MIN / 2] : ~k_lo;
      k_far_r = lo + SPLIT_MIN / 2 < pl.n ? keys[lo + SPLIT_MIN / 2] : ~k_lo;
    }
    {
      // every key this scan looks at is requested before any is examined (one memory round trip for the
      // chunk instead of up to three dependent ones per 256 entries)
      constexpr int IT = LONG_CHUNK / UBLOCK;
      uint32_t k0[IT], k1[IT], km[IT], kb[IT];
      bool ok[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int i = lo + it * UBLOCK + (int) threadIdx.x;
        ok[it] = i < hi && i + WAVE_MAX < pl.n;
        k0[it] = keys[ok[it] ? i : lo];
        k1[it] = keys[ok[it] ? i + WAVE_MAX : lo];
        km[it] = keys[ok[it] && i > 0 ? i - 1 : lo];
        kb[it] = SPLIT ? keys[ok[it] && i + SPLIT_MIN < pl.n ? i + SPLIT_MIN : lo] : 0u;
      }
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int i = lo + it * UBLOCK + (int) threadIdx.x;
        if (!ok[it] || k0[it] == sentinel || k1[it] != k0[it]) continue;
        if (i > 0 && km[it] == k0[it]) continue;
        // bit 30 marks a run of more than SPLIT_MIN entries (positions are below 2^30: see the host check)
        const bool big = SPLIT && i + SPLIT_MIN < pl.n && kb[it] == k0[it];
        sh_heads[atomicAdd(&sh_n, 1)] = i | (big ? (1 << 30) : 0);
      }
    }
    __syncthreads();
    const int nh = sh_n;
    SEG_PHASE(0);  // scan done
    if (threadIdx.x == 0) {  // ascending position: the order a workgroup takes its runs in must not depend on timing
      for (int x = 1; x < nh; ++x) {
        const int v = sh_heads[x];
        int y = x - 1;
        for (; y >= 0 && (sh_heads[y] & ~(1 << 30)) > (v & ~(1 << 30)); --y) sh_heads[y + 1] = sh_heads[y];
        sh_heads[y + 1] = v;
      }
    }
    __syncthreads();
    // Heads of runs of up to WAVE_RUN_MAX entries are dealt to the workgroup's waves round robin, each
    // summed by its wave alone: a chunk that holds several such heads (a table of 27 rows at batch 8192:
    // runs of ~300, three or four heads per chunk) works on them side by side instead of one after
    // the other with all its threads and a dozen barriers each.
    __shared__ int sh_done[LONG_CAP], sh_ends[LONG_CAP];
    SEG_PHASE(1);  // heads in order
    for (int h = threadIdx.x >> 6; h < nh; h += UBLOCK / 64) {  // uniform over the wave
      const int hv = sh_heads[h];
      int done = 0, hend_out = -1;
      if (!(hv >> 30)) {
        const uint32_t hrow = keys[hv];
        const int hend = run_end_wave(keys, pl.n, hv, hrow);
        hend_out = hend;  // kept for the workgroup phase below: no second search
        if (hend - hv <= WAVE_RUN_MAX) {
          wave_run<VEC>(pl, src, upd, lanes, hv, hend, hrow);
          done = 1;
        }
      }
      if ((threadIdx.x & 63) == 0) {
        sh_done[h] = done;
        sh_ends[h] = hend_out;
      }
    }
    __syncthreads();
    SEG_PHASE(2);  // wave-dealt heads done (searches, runs of up to WAVE_RUN_MAX)
    // A long run is summed in pieces, one per LONG_CHUNK it touches, each by the workgroup of that
    // chunk, so that a row hit by a large part of the batch does not keep ONE workgroup busy while
    // the rest of the chip is done.  Piece h = -1: the run that enters this chunk from the left.
    const int sub = threadIdx.x / lanes, c = threadIdx.x - sub * lanes;
    const int dimv = lanes * VEC + 1;  // floats of a partial: the vector sum and the scalar
    int32_t* ticket = pl.long_list + 2;
    float* parts = reinterpret_cast<float*>(pl.long_list + 2 + long_blocks);
    for (int h = SPLIT ? -1 : 0; h < nh; ++h) {  // uniform
      int beg, piece_lo;
      uint32_t row;
      if (h < 0) {
        row = k_lo;
        if (row == sentinel || k_prev != row) continue;  // nothing enters from the left
        if (k_far_l != row && k_far_r != row) continue;  // nothing long enough to be shared
        if (threadIdx.x < 64) {
          const int b = run_begin_wave(keys, lo, row);
          if (threadIdx.x == 0) sh_span[0] = b;
        }
        piece_lo = lo;
      } else {
        if (sh_done[h]) continue;  // summed by one of the waves above
        piece_lo = sh_heads[h] & ~(1 << 30);
        row = keys[piece_lo];
        if (!SPLIT || !(sh_heads[h] >> 30)) {  // the whole run is this workgroup's
          block_run<VEC>(pl, src, upd, lanes, groups, piece_lo, sh_ends[h], row, red, reds, &sh_end);
          continue;
        }
        if (threadIdx.x == 0) sh_span[0] = piece_lo;
      }
      if (threadIdx.x < 64) {
        const int e = run_end_wave(keys, pl.n, piece_lo, row);
        if (threadIdx.x == 0) sh_span[1] = e;
      }
      __syncthreads();
      beg = sh_span[0];
      const int end = sh_span[1];
      __syncthreads();  // sh_span is rewritten by the next piece
      if (end - beg <= WAVE_MAX) continue;  // a short run that straddles the chunk border: the waves' job
      const int first_chunk = beg / LONG_CHUNK, last_chunk = (end - 1) / LONG_CHUNK;
      // Every workgroup that meets the run takes the same decision from (beg, end).  Pieces cost a
      // device-wide fence each, so only runs of several chunks are split; the others belong to the
      // workgroup of the chunk they start in.
      if (end - beg <= SPLIT_MIN) continue;  // (a run entering from the left that its head's workgroup sums alone)
      const int piece_hi = end < lo + LONG_CHUNK ? end : lo + LONG_CHUNK;
      block_sum<VEC>(pl, src, lanes, groups, piece_lo, piece_hi, red, reds);
      float* mine = parts + ((size_t) bid * 2 + (h < 0 ? 0 : 1)) * dimv;
      // The partial goes out with device-scope stores (written through to where every XCD sees
      // it) and is acknowledged before the ticket is taken.  A device-wide fence would do the same
      // but also writes back this XCD's whole L2, which the row updates around keep full of dirty
      // lines: that cost more than the piecewise sum saved.
      if (sub == 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          __hip_atomic_store(mine + c * VEC + k, red[c * VEC + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c == 0) __hip_atomic_store(mine + dimv - 1, reds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)
        sh_last = __hip_atomic_fetch_add(&ticket[first_chunk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                          last_chunk - first_chunk
                      ? 1 : 0;
      __syncthreads();
      if (!sh_last) continue;
      // the last piece to arrive adds the partials in position order and applies the update
      if (sub == 0) {
        typename Upd::template Regs<VEC> regs;
        upd.template load<VEC>((int64_t) row, c, regs);
        float g[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) g[k] = 0.f;
        float gs = 0.f;
        for (int ch = first_chunk; ch <= last_chunk; ++ch) {
          const float* p = parts + ((size_t) ch * 2 + (ch == first_chunk ? 1 : 0)) * dimv;
#pragma unroll
          for (int k = 0; k < VEC; ++k) g[k] += __hip_atomic_load(p + c * VEC + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gs += __hip_atomic_load(p + dimv - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        upd.template finish<VEC>((int64_t) row, c, lanes, regs, g, gs);
      }
      if (threadIdx.x == 0) __hip_atomic_store(&ticket[first_chunk], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int base = ((bid - long_blocks) * (UBLOCK / 64) + (threadIdx.x >> 6)) * epw;
  if (base >= pl.n) return;  // uniform over the wave
  const int i = base + lane;
  const bool mine = lane < epw && i < pl.n;
  const uint32_t key = mine ? keys[i] : sentinel;
  // the wave's permutation entries, requested with its rows (one coalesced load): a run that closes
  // inside the wave gets its entries by shuffle instead of a dependent load per run
  const int perm_reg = pl.perm[mine ? i : base];
  // narrow rows: also the next 64 rows and permutation entries, for a last run that goes on past the wave
  const bool mine2 = LP > 0 && i + 64 < pl.n;
  const uint32_t key2r = LP > 0 ? keys[mine2 ? i + 64 : base] : 0u;
  const int perm2 = LP > 0 ? pl.perm[mine2 ? i + 64 : base] : 0;
  const uint32_t key2 = mine2 ? key2r : sentinel;
  uint32_t edge = sentinel;  // lane 0: the row before the wave's first; lane epw-1: the row after its last
  if (lane == 0 && base > 0) edge = keys[base - 1];
  if (lane == epw - 1 && base + epw < pl.n) edge = keys[base + epw];
  uint32_t prev = (uint32_t) __shfl_up((int) key, 1, 64);
  const uint32_t before = (uint32_t) __shfl((int) edge, 0, 64);
  const uint32_t after_wave = (uint32_t) __shfl((int) edge, epw - 1, 64);
  if (lane == 0) prev = base > 0 ? before : ~key;
  // Every change of row starts a run, padding (sentinel) runs included - those are skipped below.
  // (After a per-table sort the sentinels close each table's block instead of the whole array.)
  const unsigned long long wmask = epw >= 64 ? ~0ull : ((1ull << epw) - 1ull);  // the lanes that hold entries
  const bool head = key != prev;
  const unsigned long long hm = __ballot(head) & wmask;
  const int nheads = __popcll(hm);
  if constexpr (LP > 0) {
    __shared__ float lds_cs[UBLOCK / 64][(64 + 64 / LP) * (LP * VEC + 4)];
    __shared__ float lds_ss[UBLOCK / 64][64 + 64 / LP];
    __shared__ int lds_hp[UBLOCK / 64][65];
    const int w = threadIdx.x >> 6;
    short_wave_lds<VEC, LP>(pl, src, upd, base, key, perm_reg, key2, perm2, hm, sentinel, lds_cs[w], lds_ss[w],
                            lds_hp[w]);
    return;
  }
  const int G = 64 / lanes, g = lane / lanes, c = lane - g * lanes;
  if constexpr (Src::kBatchSingles && Upd::kBatchSingles > 0) {
    // A full wave whose 64 entries are 64 different rows, none of them padding and the last one not
    // continued in the next wave (the usual case once the tables are much larger than the batch):
    // every run is one entry, so a lane group takes four of them per round - four permutation entries
    // and row states requested together, then the four gradient rows - instead of one run per round.
    // Nothing in the round is conditional (a branch around a load makes the compiler wait for all
    // outstanding loads where the paths join, which would serialise the four again).
    const uint32_t last_key = (uint32_t) __shfl((int) key, epw - 1, 64);
    constexpr int BT = Upd::kBatchSingles;  // rows per lane group per round
    // (a per-table sort leaves padding entries at the end of each TABLE's block, so a lone one can sit
    // anywhere in the wave and still differ from both neighbours: every lane is checked, not the last)
    const bool no_padding = (__ballot(key == sentinel) & wmask) == 0ull;
    if (hm == wmask && lanes >= BT && epw >= BT * G && base + epw <= pl.n && no_padding &&
        !(base + epw < pl.n && after_wave == last_key)) {  // uniform
      for (int r0 = 0; r0 * G < epw; r0 += BT) {  // epw / G rounds of one row per group, BT of them at a time
        uint32_t row[BT];
        int p[BT];
        typename Upd::template Regs<VEC> regs[BT];
#pragma unroll
        for (int u = 0; u < BT; ++u) {
          const int pos = (r0 + u) * G + g;
          row[u] = (uint32_t) __shfl((int) key, pos, 64);
          p[u] = pl.perm[base + pos];
          upd.template load<VEC>((int64_t) row[u], c, regs[u]);
        }
        float v[BT][VEC], sc[BT];
#pragma unroll
        for (int u = 0; u < BT; ++u) src.template load<VEC>(pl, p[u], c, v[u], sc[u]);
#pragma unroll
        for (int u = 0; u < BT; ++u) {
          float acc[VEC];
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] = 0.f + v[u][k];  // the sum of a one-entry run, as group_run forms it
          upd.template finish<VEC>((int64_t) row[u], c, lanes, regs[u], acc, 0.f + sc[u]);
        }
      }
      return;
    }
  }
  // One round of the general path: lane group g takes head k (if there is one) whatever its run looks like.
  auto one_head = [&](int k) {
    const int pos = k < nheads ? kth_set_bit(hm, k) : 0;
    const unsigned long long later = (pos < 63) ? (hm & ~((2ull << pos) - 1ull)) : 0ull;  // heads after pos
    const int endpos = later ? __ffsll((long long) later) - 1 : epw;
    const uint32_t row = (uint32_t) __shfl((int) key, pos, 64);
    const bool have = k < nheads && row != sentinel;
    const bool open = have && !later && base + epw < pl.n && after_wave == row;  // continues in the next wave
    const int len = endpos - pos;
    int p4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) p4[q] = __shfl(perm_reg, pos + q < epw ? pos + q : pos, 64);  // every lane takes part
    bool pending = false;
    if (have) {
      if (!open && len <= 4) group_run4_pre<VEC>(pl, src, upd, lanes, c, p4, len, row);
      else if (!open && len <= GROUP_MAX) group_run<VEC>(pl, src, upd, lanes, c, base + pos, len, row);
      else pending = (c == 0);
    }
    wave_pending<VEC>(pl, src, upd, lanes, pending, base + pos, open ? -1 : base + endpos, row, base + pos, false);
  };
  if constexpr (Src::kBatchSingles && Upd::kBatchSingles > 0) {
    // Waves that mix one-entry runs with a few longer ones (a third of the waves when the tables are ~70 x the
    // batch): a lane group still takes BT heads per round - row states and first gradient sources of all BT
    // requested together, the second sources only if some run of the round has two entries - as long as every
    // run of the round closes inside the wave with at most two entries and none is padding.  A round that holds
    // anything else goes head by head as before.  Sums are formed as group_run4_pre forms them: same bits.
    constexpr int BT = Upd::kBatchSingles;
    if (lanes >= BT) {
      for (int r0 = 0; r0 * G < nheads; r0 += BT) {  // uniform over the wave
        int pos[BT], len[BT];
        uint32_t row[BT];
        bool have[BT];
        bool simple = true, two = false;
#pragma unroll
        for (int u = 0; u < BT; ++u) {
          const int k = (r0 + u) * G + g;
          const bool in = k < nheads;
          pos[u] = in ? kth_set_bit(hm, k) : 0;
          const unsigned long long later = (pos[u] < 63) ? (hm & ~((2ull << pos[u]) - 1ull)) : 0ull;
          const int endpos = later ? __ffsll((long long) later) - 1 : epw;
          row[u] = (uint32_t) __shfl((int) key, pos[u], 64);
          have[u] = in && row[u] != sentinel;
          const bool open = have[u] && !later && base + epw < pl.n && after_wave == row[u];
          len[u] = endpos - pos[u];
          simple = simple && (!in || (have[u] && !open && len[u] <= 2));
          two = two || (have[u] && len[u] == 2);
        }
        if (__ballot(!simple) == 0ull) {  // uniform
          typename Upd::template Regs<VEC> regs[BT];
          int p0[BT], p1[BT];
#pragma unroll
          for (int u = 0; u < BT; ++u) {
            p0[u] = __shfl(perm_reg, pos[u], 64);
            p1[u] = __shfl(perm_reg, pos[u] + 1 < epw ? pos[u] + 1 : pos[u], 64);
            upd.template load<VEC>(have[u] ? (int64_t) row[u] : 0, c, regs[u]);  // no branch around the load
          }
          float v[BT][VEC], sc[BT];
#pragma unroll
          for (int u = 0; u < BT; ++u) src.template load<VEC>(pl, p0[u], c, v[u], sc[u]);
#pragma unroll
          for (int u = 0; u < BT; ++u) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[u][k] = 0.f + v[u][k];
            sc[u] = 0.f + sc[u];
          }
          if (__ballot(two) != 0ull) {  // uniform; everything requested so far is awaited below anyway
            float v2[BT][VEC], s2[BT];
#pragma unroll
            for (int u = 0; u < BT; ++u) src.template load<VEC>(pl, p1[u], c, v2[u], s2[u]);
#pragma unroll
            for (int u = 0; u < BT; ++u) {
              if (len[u] == 2) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[u][k] += v2[u][k];
                sc[u] += s2[u];
              }
            }
          }
#pragma unroll
          for (int u = 0; u < BT; ++u) {
            if (have[u]) upd.template finish<VEC>((int64_t) row[u], c, lanes, regs[u], v[u], sc[u]);
          }
        } else {
          for (int u = 0; u < BT && (r0 + u) * G < nheads; ++u) one_head((r0 + u) * G + g);  // uniform bounds
        }
      }
      return;
    }
  }
  for (int r = 0; r * G < nheads; ++r) one_head(r * G + g);  // uniform over the wave
}



template <int VEC, bool SPLIT, int LP, class Src, class Upd, class Extra>
__global__ __launch_bounds__(UBLOCK) void seg_sorted_kernel(PlanDev pl, Src src, Upd upd, int lanes_arg, int groups,
                                                           int long_blocks, Extra extra, int epw_arg) {
#ifdef SEG_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    seg_trace_buf[blockIdx.x * 2] = wall_clock64();
    seg_trace_buf[blockIdx.x * 2 + 1] = 0;
  }
#endif
  seg_sorted_body<VEC, SPLIT, LP>(pl, src, upd, lanes_arg, groups, long_blocks, extra, epw_arg);
#ifdef SEG_TRACE
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) atomicMax(&seg_trace_buf[blockIdx.x * 2 + 1], (unsigned long long) wall_clock64());
#endif
}

// Queued runs (longer than WAVE_MAX, or longer than 4 when lane groups straddle waves): one
// workgroup per run.
template <int VEC, class Src, class Upd>
__global__ __launch_bounds__(UBLOCK) void seg_queue_kernel(PlanDev pl, Src src, Upd upd, int lanes, int groups) {
  __shared__ float red[UBLOCK * VEC];
  __shared__ float reds[UBLOCK];
  __shared__ int sh_end;
  const int n_q = pl.long_list[0];
  for (int i = blockIdx.x; i < n_q; i += gridDim.x) {
    const int token = pl.long_list[2 + i];
    if (pl.seg_offsets)
      block_run<VEC>(pl, src, upd, lanes, groups, pl.seg_offsets[token], pl.seg_offsets[token + 1],
                     pl.unique_rows[token], red, reds, &sh_end);
    else
      block_run<VEC>(pl, src, upd, lanes, groups, token, -1, pl.sorted_rows[token], red, reds, &sh_end);
  }
  // The workgroup that finishes last leaves the queue empty for the next call: every
  // workgroup read the count before taking its ticket, so nobody still needs it.
  if (threadIdx.x == 0) {
    const int ticket = atomicAdd(&pl.long_list[1], 1);
    if (ticket == (int) gridDim.x - 1) {
      atomicExch(&pl.long_list[0], 0);
      atomicExch(&pl.long_list[1], 0);
    }
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
  OptRegs<MI_OPT_ADAM_TF, VEC> regs;
  upd.template load<VEC>(row, c, regs);
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  float sc = 0.f;
  if (lo < n_unique && (int64_t) pl.unique_rows[lo] == row)
    sum_run<VEC>(pl, src, c, pl.seg_offsets[lo], pl.seg_offsets[lo + 1], 1, acc, sc);
  upd.template finish<VEC>(row, c, lanes, regs, acc, 0.f);
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

int check_plan(const mi_bwd_plan* plan, PlanDev* pl, bool need_runs, const char* who) {
  MI_REQUIRE(plan, "%s: null plan", who);
  MI_REQUIRE(plan->sorted_rows && plan->perm && plan->long_list, "%s: plan has a null pointer", who);
  const bool has_runs = plan->unique_rows && plan->seg_offsets && plan->counts;
  MI_REQUIRE(has_runs || (!plan->unique_rows && !plan->seg_offsets && !plan->counts),
             "%s: unique_rows, seg_offsets and counts must be all set or all NULL", who);
  MI_REQUIRE(has_runs || !need_runs, "%s: this entry point needs the compacted runs in the plan", who);
  MI_REQUIRE(plan->T >= 1 && plan->n_entries >= 0 && plan->n_entries < ((int64_t) 1 << 31),
             "%s: plan T / n_entries out of range", who);
  MI_REQUIRE(plan->total_rows >= 1 && plan->total_rows < (int64_t) 0xFFFFFFFFll, "%s: plan total_rows out of range",
             who);
  MI_REQUIRE(plan->offsets || plan->bag_len >= 1, "%s: plan needs offsets or bag_len >= 1", who);
  MI_REQUIRE(plan->pool_mode == MI_POOL_SUM ||
                 (plan->pool_mode == MI_POOL_MEAN &&
                  (plan->bag_scale || (plan->indices && plan->table_offsets &&
                                       (plan->idx_bits == 32 || plan->idx_bits == 64)))),
             "%s: mean pooling needs bag_scale, or indices, idx_bits and table_offsets, in the plan", who);
  pl->sorted_rows = plan->sorted_rows;
  pl->unique_rows = plan->unique_rows;
  pl->perm = plan->perm;
  pl->seg_offsets = plan->seg_offsets;
  pl->counts = plan->counts;
  pl->long_list = plan->long_list;
  pl->n_dev = nullptr;
  pl->split = 0;
  pl->offsets = plan->offsets;
  pl->indices = plan->indices;
  pl->toffs = plan->table_offsets;
  pl->bag_scale = plan->bag_scale;
  pl->entry_bag = plan->offsets ? plan->entry_bag : nullptr;
  pl->sentinel = (uint32_t) plan->total_rows;
  pl->n = (int) plan->n_entries;
  pl->n_bags = (int) plan->n_bags;
  pl->bag_len = plan->bag_len;
  pl->T = plan->T;
  pl->pool_mode = plan->pool_mode;
  pl->idx_bits = plan->idx_bits;
  return MI_OK;
}

// `extra` (see seg_sorted_kernel) is only taken by plans without run arrays; the caller checks.
template <class Src, class Upd, class Extra = NoExtra>
int run_segments(const mi_bwd_plan* plan, const PlanDev& pl, const Src& src, const Upd& upd, const Geometry& g,
                 hipStream_t s, const Extra& extra = Extra{0}) {
  if (plan->n_entries == 0 && extra.blocks == 0) return MI_OK;
  const int wave_ok = (g.lanes <= 64 && (g.lanes & (g.lanes - 1)) == 0) ? 1 : 0;  // shuffle trees need 2^k lanes
  if constexpr (rows16_only<Upd>::value) {
    if (g.vec != 4 || g.lanes != 4) {
      mi_set_error("sparse update: this updater is built for rows of 16 floats");
      return MI_ERR_UNSUPPORTED;
    }
  }
  if (pl.seg_offsets) {
    const unsigned grid = (unsigned) mi_cdiv(plan->n_entries, g.gpb);
    if (rows16_only<Upd>::value || g.vec == 4)
      hipLaunchKernelGGL((seg_runs_kernel<4, Src, Upd>), dim3(grid), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes, g.gpb,
                         wave_ok);
    else if constexpr (!rows16_only<Upd>::value)
      hipLaunchKernelGGL((seg_runs_kernel<1, Src, Upd>), dim3(grid), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes, g.gpb,
                         wave_ok);
  } else {
    if (!wave_ok) {
      mi_set_error("sparse update: a plan without run arrays needs dim/%d = %d lanes per row to be a power of two <= 64",
                   g.vec, g.lanes);
      return MI_ERR_UNSUPPORTED;
    }
    const int long_blocks = (int) mi_cdiv(plan->n_entries, LONG_CHUNK);
    // piecewise long runs need tickets and two partials per chunk behind the two header words of long_list
    PlanDev pls = pl;
    // ... and are only worth their extra work when one row can collect several times SPLIT_MIN
    // entries: with fixed-width bags a row gets at most the n_entries / T entries of its table
    const int64_t per_table = plan->offsets ? plan->n_entries : plan->n_entries / plan->T;
    pls.split = (int64_t) long_blocks * (1 + 2 * (g.lanes * g.vec + 1)) <= plan->n_entries &&
                        plan->n_entries < ((int64_t) 1 << 30) && per_table > MI_SPLIT_GATE * SPLIT_MIN
                    ? 1 : 0;
    // entries per wave: 64, halved (down to 16, and never below four rows per lane group) while that leaves
    // fewer than 8192 waves - rows of 32 lanes give a wave of 64 entries 8 to 32 dependent rounds, and a
    // batch of 65536 such entries would occupy one wave per SIMD
    int epw = 64;
    if (g.lanes > 4) {
      const int floor_epw = 4 * (64 / g.lanes) > 16 ? 4 * (64 / g.lanes) : 16;
      while (epw > floor_epw && plan->n_entries / epw < 8192) epw >>= 1;
    }
    const unsigned grid = (unsigned) (extra.blocks + long_blocks + mi_cdiv(plan->n_entries, (int64_t) epw * (UBLOCK / 64)));
#define SEG_LAUNCH(V, SP, LP)                                                                                          \
  hipLaunchKernelGGL((seg_sorted_kernel<V, SP, LP, Src, Upd, Extra>), dim3(grid), dim3(UBLOCK), 0, s, pls, src, upd,   \
                     g.lanes, g.gpb, long_blocks, extra, epw)
#define SEG_LANES(V, SP)                                                                                               \
  switch (g.lanes) {                                                                                                   \
    case 1: SEG_LAUNCH(V, SP, 1); break;                                                                               \
    case 2: SEG_LAUNCH(V, SP, 2); break;                                                                               \
    case 4: SEG_LAUNCH(V, SP, 4); break;                                                                               \
    default: SEG_LAUNCH(V, SP, 0); break;                                                                              \
  }
    if constexpr (rows16_only<Upd>::value) {
      if (pls.split) { SEG_LAUNCH(4, true, 4); } else { SEG_LAUNCH(4, false, 4); }
    } else if (g.vec == 4) {
      if (pls.split) { SEG_LANES(4, true) } else { SEG_LANES(4, false) }
    } else {
      if (pls.split) { SEG_LANES(1, true) } else { SEG_LANES(1, false) }
    }
#undef SEG_LANES
#undef SEG_LAUNCH
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
  if (rows16_only<Upd>::value || g.vec == 4)
    hipLaunchKernelGGL((seg_queue_kernel<4, Src, Upd>), dim3(LONG_GRID), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes,
                       g.gpb);
  else if constexpr (!rows16_only<Upd>::value)
    hipLaunchKernelGGL((seg_queue_kernel<1, Src, Upd>), dim3(LONG_GRID), dim3(UBLOCK), 0, s, pl, src, upd, g.lanes,
                       g.gpb);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
bool al16(const void* p) { return ((uintptr_t) p & 15) == 0; }

}  // namespace
