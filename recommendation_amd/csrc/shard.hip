// Row-wise sharded tables: the kernels either side of the RCCL exchanges.
//
// The reference distributes with TensorFlow's asynchronous parameter server
// (recommendation/code/model_trainer.py:124-136: every variable read and every Adam apply is a
// gRPC hop to the ps task).  Here the rows of all tables are dealt round-robin over the N ranks
// (global row r lives on rank r % N as local row r / N) and a step exchanges, per rank pair,
//   requester -> owner : the local rows it needs, grouped by sample (route_* kernels)
//   owner -> requester : per SAMPLE partial pools  sum v, sum |v|^2, sum w1  (fm_partial_fwd)
// so the payload coming back is D+2 floats per sample and owner, not D+1 per (sample, field).
// Nothing here uses float atomics: every sum has a fixed order (fields in field order, owners
// in rank order), so a sharded run reproduces itself bit for bit.
#include "common.h"

namespace {

constexpr int MAX_RANKS = 16;

template <typename IDX>
__device__ __forceinline__ bool route_one(const int64_t* toffs, const void* ids, int T, int b, int f, int N,
                                          int& dest, int& local) {
  const int64_t id = mi_ld_id<IDX>(ids, (int64_t) b * T + f);
  const int64_t base = toffs[f];
  const int64_t rows = toffs[f + 1] - base;
  if (id < 0 || id >= rows) return false;
  const int64_t g = base + id;
  dest = (int) (g % N);
  local = (int) (g / N);
  return true;
}

// The exchange buffer has one block per rank, all the same size so that the all-to-all needs no
// size negotiation (and no host synchronisation):
//   block d = [ bag_off[0..B] | local rows, grouped by sample, fields in field order | unused ]
// with stride B+1+B*T ints; bag_off[B] is the number of rows in the block.

// thread per sample: how many of its T entries go to each rank -> block[dest][b]
template <typename IDX>
__global__ __launch_bounds__(256) void route_count_kernel(const int64_t* __restrict__ toffs, int T,
                                                         const void* __restrict__ ids, int B, int N, int stride,
                                                         int32_t* __restrict__ buf, int32_t* __restrict__ err) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int cnt[MAX_RANKS];
#pragma unroll
  for (int r = 0; r < MAX_RANKS; ++r) cnt[r] = 0;
  bool bad = false;
  for (int f = 0; f < T; ++f) {
    int dest, local;
    if (route_one<IDX>(toffs, ids, T, b, f, N, dest, local)) {
#pragma unroll
      for (int r = 0; r < MAX_RANKS; ++r) cnt[r] += (r == dest) ? 1 : 0;
    } else {
      bad = true;
    }
  }
#pragma unroll
  for (int r = 0; r < MAX_RANKS; ++r)
    if (r < N) buf[(size_t) r * stride + b] = cnt[r];
  if (bad && err) atomicOr(err, MI_IDERR_RANGE);
}

// workgroup per destination rank: exclusive scan of block[dest][0..B) in place (entry [B] = total)
__global__ __launch_bounds__(256) void route_scan_kernel(int32_t* __restrict__ buf, int B, int stride) {
  __shared__ int32_t wsum[4];
  int32_t* row = buf + (size_t) blockIdx.x * stride;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int32_t carry = 0;
  for (int base = 0; base < B; base += 256) {
    const int i = base + threadIdx.x;
    const int32_t v = i < B ? row[i] : 0;
    int32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int32_t wb = 0;
    for (int k = 0; k < w; ++k) wb += wsum[k];
    if (i < B) row[i] = carry + wb + inc - v;
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  if (threadIdx.x == 0) row[B] = carry;
}

// thread per sample: write its local rows into the blocks
template <typename IDX>
__global__ __launch_bounds__(256) void route_fill_kernel(const int64_t* __restrict__ toffs, int T,
                                                        const void* __restrict__ ids, int B, int N, int stride,
                                                        int32_t* __restrict__ buf) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int slot[MAX_RANKS];
#pragma unroll
  for (int r = 0; r < MAX_RANKS; ++r) slot[r] = r < N ? r * stride + (B + 1) + buf[(size_t) r * stride + b] : 0;
  for (int f = 0; f < T; ++f) {
    int dest, local;
    if (!route_one<IDX>(toffs, ids, T, b, f, N, dest, local)) continue;
    int at = 0;
#pragma unroll
    for (int r = 0; r < MAX_RANKS; ++r) {
      if (r == dest) { at = slot[r]; slot[r] += 1; }
    }
    buf[at] = local;
  }
}

// Owner side.  Bag (src, b) = rows [bag_off[b], bag_off[b+1]) of received block src.
// One lane group (DIM/4 lanes) per bag: partial[src][b] = { sum v (DIM), sum |v|^2, sum w1 }.
// The rows are also written out back to back over all blocks (rows_out, what the backward's sort
// takes) with entry_sample[entry] = src*B + b, and n_out[0] = their number.  Empty bag -> zeros.
template <int DIM>
__global__ __launch_bounds__(256) void fm_partial_fwd_kernel(mi_fm_table tab, const int32_t* __restrict__ recv,
                                                            int stride, int N, int B, int64_t local_rows,
                                                            float* __restrict__ partial,
                                                            int32_t* __restrict__ rows_out,
                                                            int32_t* __restrict__ entry_sample,
                                                            int32_t* __restrict__ n_out, int32_t* __restrict__ err) {
  constexpr int CH = DIM / 4;
  const int bag = blockIdx.x * (256 / CH) + threadIdx.x / CH;
  if (bag >= N * B) return;
  const int c = threadIdx.x % CH;
  const int src = bag / B, b = bag - src * B;
  const int32_t* blk = recv + (size_t) src * stride;
  int cbase = 0, total = 0;
  for (int r = 0; r < N; ++r) {
    const int cnt = recv[(size_t) r * stride + B];
    if (r < src) cbase += cnt;
    total += cnt;
  }
  if (bag == 0 && c == 0) n_out[0] = total;
  const int beg = blk[b], end = blk[b + 1];
  const int32_t* rows = blk + (B + 1);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float q = 0.f, lin = 0.f;
  bool bad = beg < 0 || end < beg || end > stride - (B + 1);
  constexpr int UF = 4;
  for (int e0 = beg; !bad && e0 < end; e0 += UF) {
    int64_t row[UF];
    bool ok[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const bool present = e0 + u < end;
      const int64_t r = rows[present ? e0 + u : beg];
      ok[u] = present && r >= 0 && r < local_rows;
      if (present && !ok[u]) bad = true;
      row[u] = ok[u] ? r : 0;
      if (present && c == 0) {
        rows_out[cbase + e0 + u] = (int32_t) row[u];
        entry_sample[cbase + e0 + u] = bag;
      }
    }
    float4 v[UF];
    float w[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const float* r = tab.rec + row[u] * tab.stride;
      v[u] = *reinterpret_cast<const float4*>(r + c * 4);
      w[u] = (c == 0) ? r[tab.off_w1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      if (ok[u]) {
        s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
        q = fmaf(v[u].x, v[u].x, q); q = fmaf(v[u].y, v[u].y, q);
        q = fmaf(v[u].z, v[u].z, q); q = fmaf(v[u].w, v[u].w, q);
        lin += w[u];
      }
    }
  }
#pragma unroll
  for (int o = 1; o < CH; o <<= 1) q += __shfl_xor(q, o, 64);
  float* out = partial + (size_t) bag * (DIM + 2);
  out[c * 4 + 0] = s.x; out[c * 4 + 1] = s.y; out[c * 4 + 2] = s.z; out[c * 4 + 3] = s.w;
  if (c == 0) { out[DIM] = q; out[DIM + 1] = lin; }
  if (bad && err) atomicOr(err, MI_IDERR_RANGE);
}

// Requester side: sum the owners' partials in rank order, add the dense part and the bias, and
// write one row per sample of what the other ranks need:
//   X[b] = [ S (DIM) | dL/dlogit | BCE loss row | dense features (n_dense) | padding ]   (ld floats)
// so that a single all-gather of X feeds every owner's backward and every replica's dense step.
template <int DIM>
__global__ __launch_bounds__(256) void fm_combine_kernel(const float* __restrict__ partials, int N, int B,
                                                        const float* __restrict__ dense, int n_dense,
                                                        const float* __restrict__ wd, const float* __restrict__ bias,
                                                        const float* __restrict__ labels, float gscale,
                                                        float* __restrict__ logit, float* __restrict__ X, int ld) {
  constexpr int CH = DIM / 4;
  const int b = blockIdx.x * (256 / CH) + threadIdx.x / CH;
  if (b >= B) return;
  const int c = threadIdx.x % CH;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float q = 0.f, lin = 0.f;
  for (int r = 0; r < N; ++r) {
    const float* p = partials + ((size_t) r * B + b) * (DIM + 2);
    s.x += p[c * 4 + 0]; s.y += p[c * 4 + 1]; s.z += p[c * 4 + 2]; s.w += p[c * 4 + 3];
    if (c == 0) { q += p[DIM]; lin += p[DIM + 1]; }
  }
  float* x = X + (size_t) b * ld;
  *reinterpret_cast<float4*>(x + c * 4) = s;
  float ss = s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
  for (int j = c; j < n_dense; j += CH) {
    const float d = dense[(size_t) b * n_dense + j];
    x[DIM + 2 + j] = d;
    lin = fmaf(wd[j], d, lin);
  }
  for (int j = DIM + 2 + n_dense + c; j < ld; j += CH) x[j] = 0.f;
#pragma unroll
  for (int o = 1; o < CH; o <<= 1) {
    ss += __shfl_xor(ss, o, 64);
    lin += __shfl_xor(lin, o, 64);
  }
  if (c == 0) {
    const float z = 0.5f * (ss - q) + lin + (bias ? bias[0] : 0.f);
    if (logit) logit[b] = z;
    float loss = 0.f, gb = 0.f;
    if (labels) {
      const float y = labels[b];
      const float e = expf(-fabsf(z));
      loss = fmaxf(z, 0.f) - z * y + log1pf(e);
      const float sig = z >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      gb = (sig - y) * gscale;
    }
    x[DIM] = gb;
    x[DIM + 1] = loss;
  }
}

}  // namespace

extern "C" int64_t mi_shard_block_ints(int64_t B, int T) { return B + 1 + B * T; }

extern "C" int mi_shard_route(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int64_t B,
                              int n_ranks, int32_t* send, int32_t* err, void* stream) {
  MI_REQUIRE(table_offsets && ids && send, "mi_shard_route: null pointer");
  MI_REQUIRE(T >= 1 && B >= 1 && B < ((int64_t) 1 << 31) / T / 17, "mi_shard_route: bad T/B");
  MI_REQUIRE(n_ranks >= 1 && n_ranks <= MAX_RANKS, "mi_shard_route: n_ranks must be 1..%d", MAX_RANKS);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_shard_route: idx_bits must be 32 or 64");
  hipStream_t s = (hipStream_t) stream;
  const unsigned grid = (unsigned) mi_cdiv(B, 256);
  const int stride = (int) mi_shard_block_ints(B, T);
  if (idx_bits == 32)
    hipLaunchKernelGGL(route_count_kernel<int32_t>, dim3(grid), dim3(256), 0, s, table_offsets, T, ids, (int) B,
                       n_ranks, stride, send, err);
  else
    hipLaunchKernelGGL(route_count_kernel<int64_t>, dim3(grid), dim3(256), 0, s, table_offsets, T, ids, (int) B,
                       n_ranks, stride, send, err);
  hipLaunchKernelGGL(route_scan_kernel, dim3(n_ranks), dim3(256), 0, s, send, (int) B, stride);
  if (idx_bits == 32)
    hipLaunchKernelGGL(route_fill_kernel<int32_t>, dim3(grid), dim3(256), 0, s, table_offsets, T, ids, (int) B,
                       n_ranks, stride, send);
  else
    hipLaunchKernelGGL(route_fill_kernel<int64_t>, dim3(grid), dim3(256), 0, s, table_offsets, T, ids, (int) B,
                       n_ranks, stride, send);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fm_partial_fwd(const mi_fm_table* tab, int64_t local_rows, const int32_t* recv, int n_ranks,
                                 int64_t B, int T, float* partial, int32_t* rows_out, int32_t* entry_sample,
                                 int32_t* n_out, int32_t* err, void* stream) {
  const int rc = mi_fm_check_table(tab, MI_OPT_SGD, "mi_fm_partial_fwd");
  if (rc) return rc;
  MI_REQUIRE(recv && partial && rows_out && entry_sample && n_out, "mi_fm_partial_fwd: null pointer");
  MI_REQUIRE(n_ranks >= 1 && n_ranks <= MAX_RANKS && T >= 1 && B >= 1 && B < ((int64_t) 1 << 31) / T / 17,
             "mi_fm_partial_fwd: bad n_ranks/B/T");
  hipStream_t s = (hipStream_t) stream;
  const int64_t bags = (int64_t) n_ranks * B;
  const int stride = (int) mi_shard_block_ints(B, T);
#define PF_CASE(D)                                                                                               \
  case D:                                                                                                        \
    hipLaunchKernelGGL(fm_partial_fwd_kernel<D>, dim3((unsigned) mi_cdiv(bags, 256 / (D / 4))), dim3(256), 0, s, \
                       *tab, recv, stride, n_ranks, (int) B, local_rows, partial, rows_out, entry_sample, n_out, \
                       err);                                                                                     \
    break;
  switch (tab->dim) {
    PF_CASE(4) PF_CASE(8) PF_CASE(16) PF_CASE(32) PF_CASE(64) PF_CASE(128)
    default:
      mi_set_error("mi_fm_partial_fwd: dim %d not built (4, 8, 16, 32, 64, 128)", tab->dim);
      return MI_ERR_UNSUPPORTED;
  }
#undef PF_CASE
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fm_combine(const float* partials, int n_ranks, int64_t B, int dim, const float* dense,
                             int n_dense, const float* wd, const float* bias, const float* labels, float gscale,
                             float* logit, float* X, int ld, void* stream) {
  MI_REQUIRE(partials && X && n_ranks >= 1 && B >= 1, "mi_fm_combine: bad arguments");
  MI_REQUIRE(n_dense == 0 || (dense && wd), "mi_fm_combine: dense features need dense and wd");
  MI_REQUIRE(ld >= dim + 2 + n_dense && ld % 4 == 0 && ((uintptr_t) X & 15) == 0,
             "mi_fm_combine: X rows need ld >= dim+2+n_dense, ld %% 4 == 0, 16-byte alignment");
  hipStream_t s = (hipStream_t) stream;
#define CB_CASE(D)                                                                                                 \
  case D:                                                                                                          \
    hipLaunchKernelGGL(fm_combine_kernel<D>, dim3((unsigned) mi_cdiv(B, 256 / (D / 4))), dim3(256), 0, s, partials, \
                       n_ranks, (int) B, dense, n_dense, wd, bias, labels, gscale, logit, X, ld);                   \
    break;
  switch (dim) {
    CB_CASE(4) CB_CASE(8) CB_CASE(16) CB_CASE(32) CB_CASE(64) CB_CASE(128)
    default:
      mi_set_error("mi_fm_combine: dim %d not built (4, 8, 16, 32, 64, 128)", dim);
      return MI_ERR_UNSUPPORTED;
  }
#undef CB_CASE
  MI_LAUNCH_CHECK();
  return MI_OK;
}
