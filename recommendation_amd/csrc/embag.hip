// Embedding gather + sum/mean pooling, dense one-hot materialisation, id validation.
//
// The reference turns categorical ids into dense one-hot / n-hot matrices
// (recommendation/code/r_model.py:60-96) and multiplies them into the first
// tf.layers.dense kernel of each tower (r_model.py:53-54).  one_hot(x) @ W == W[x] and
// n_hot(x) @ W == sum_j W[x_j], so the same numbers come from a row gather with sum
// pooling; that is what embag_fwd_kernel does, for all tables of a model in one launch.
//
// Thread mapping: a bag (one sample's ids for one table) is served by dim/VEC adjacent
// lanes, each owning VEC consecutive columns of the pooled row; no cross-lane reduction
// is needed and the pooled output of a block is one contiguous, fully coalesced store.
// Row reads are 16 B per lane (VEC = 4), i.e. a 64 B row is one 4-lane segment.
#include "common.h"

namespace {

template <int VEC>
struct Acc {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ void ld_row(const float* p, Acc<VEC>& a);
template <>
__device__ __forceinline__ void ld_row<4>(const float* p, Acc<4>& a) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  a.v[0] = q.x; a.v[1] = q.y; a.v[2] = q.z; a.v[3] = q.w;
}
template <>
__device__ __forceinline__ void ld_row<1>(const float* p, Acc<1>& a) {
  a.v[0] = *p;
}

template <int VEC, bool BF16>
__device__ __forceinline__ void st_out(void* out, int64_t off, const Acc<VEC>& a) {
  if constexpr (BF16) {
    unsigned short* o = reinterpret_cast<unsigned short*>(out) + off;
    if constexpr (VEC == 4) {
      ushort4 q;
      q.x = mi_f2bf(a.v[0]); q.y = mi_f2bf(a.v[1]); q.z = mi_f2bf(a.v[2]); q.w = mi_f2bf(a.v[3]);
      *reinterpret_cast<ushort4*>(o) = q;
    } else {
      o[0] = mi_f2bf(a.v[0]);
    }
  } else {
    float* o = reinterpret_cast<float*>(out) + off;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(o) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
    } else {
      o[0] = a.v[0];
    }
  }
}

constexpr int EMBAG_BLOCK = 256;
constexpr int UNROLL = 4;
#ifndef MI_EMBAG_WAVE_MIN_LEN
#define MI_EMBAG_WAVE_MIN_LEN 16
#endif
#ifndef MI_EMBAG_WAVE_MAX_THREADS
#define MI_EMBAG_WAVE_MAX_THREADS (8 * 64 * 4 * 256)  // 8 waves on each of the chip's 1,024 SIMDs
#endif

template <typename IDX, int VEC, bool BF16>
__global__ __launch_bounds__(EMBAG_BLOCK) void embag_fwd_kernel(
    const float* __restrict__ weights, const int64_t* __restrict__ toffs, int T, int dim, int lanes,
    int bags_per_block, const void* __restrict__ indices, const int64_t* __restrict__ offsets,
    int bag_len, int n_bags, int pool_mode, int flags, const float* __restrict__ bias,
    void* __restrict__ out, int64_t out_ld, int32_t* __restrict__ err) {
  const int sub = threadIdx.x / lanes;
  const int c = threadIdx.x - sub * lanes;
  if (sub >= bags_per_block) return;
  const int bag = blockIdx.x * bags_per_block + sub;
  if (bag >= n_bags) return;
  const int b = bag / T;
  const int t = bag - b * T;
  const int64_t base = toffs[t];
  const int64_t rows = toffs[t + 1] - base;
  int64_t beg, end;
  if (offsets) {
    beg = offsets[bag];
    end = offsets[bag + 1];
  } else {
    beg = (int64_t) bag * bag_len;
    end = beg + bag_len;
  }
  Acc<VEC> acc;
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc.v[k] = 0.f;
  int count = 0;
  int bad = 0;
  int64_t prev = -1;
  const float* wcol = weights + (int64_t) c * VEC;
  for (int64_t j0 = beg; j0 < end; j0 += UNROLL) {
    int64_t id[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t j = j0 + u;
      const bool present = (j < end);
      id[u] = mi_ld_id<IDX>(indices, present ? j : j0);
      if (!present) id[u] = -1;
      if (present) {
        if (id[u] < 0) {
          if (flags & MI_CHECK_NONNEG) bad |= MI_IDERR_RANGE;
        } else {
          if (id[u] >= rows && (flags & MI_CHECK_RANGE)) bad |= MI_IDERR_RANGE;
          if (flags & MI_CHECK_INCREASING) {
            if (id[u] <= prev) bad |= MI_IDERR_ORDER;
            prev = id[u];
          }
        }
      }
      ok[u] = present && id[u] >= 0 && id[u] < rows;  // an id outside the table is never dereferenced
    }
    Acc<VEC> r[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      // no branch around the load (the compiler waits for all outstanding loads where such a branch joins):
      // an entry without a valid id reads row 0 of the array and is zeroed afterwards
      const int64_t row = ok[u] ? (base + id[u]) : 0;
      ld_row<VEC>(wcol + row * dim, r[u]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (!ok[u]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[u].v[k] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc.v[k] += r[u].v[k];
      count += ok[u] ? 1 : 0;
    }
  }
  if (pool_mode == MI_POOL_MEAN && count > 1) {
    const float inv = 1.0f / (float) count;
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc.v[k] *= inv;
  }
  if (bias) {
    Acc<VEC> bv;
    ld_row<VEC>(bias + (int64_t) t * dim + (int64_t) c * VEC, bv);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc.v[k] += bv.v[k];
  }
  st_out<VEC, BF16>(out, (int64_t) b * out_ld + (int64_t) t * dim + (int64_t) c * VEC, acc);
  if (bad && c == 0 && err) atomicOr(err, bad);
}

// One id per bag (the one-hot encoders, Criteo-style categorical fields): a lane group serves
// L1_ROWS bags and issues their id loads, then their row loads, together, so that a thread has
// L1_ROWS 16-byte row reads in flight instead of one; the kernel is bound by how many bytes the
// chip keeps in flight against HBM latency, not by arithmetic.  Same results as the general kernel.
constexpr int L1_ROWS = 4;
template <typename IDX, bool BF16, bool NT>
__global__ __launch_bounds__(EMBAG_BLOCK) void embag_fwd_l1_kernel(
    const float* __restrict__ weights, const int64_t* __restrict__ toffs, int T, int dim, int lanes,
    int groups, const void* __restrict__ indices, int n_bags, int flags, const float* __restrict__ bias,
    void* __restrict__ out, int64_t out_ld, int32_t* __restrict__ err) {
  const int sub = threadIdx.x / lanes;
  const int c = threadIdx.x - sub * lanes;
  if (sub >= groups) return;
  // bags of a block: [blockIdx.x * groups * L1_ROWS, ...); row u of group `sub` is bag0 + u*groups + sub, so
  // that the stores of one u are contiguous over the block
  const int bag0 = blockIdx.x * groups * L1_ROWS;
  int64_t id[L1_ROWS];
  int bag[L1_ROWS], tt[L1_ROWS];
  bool okv[L1_ROWS];
#pragma unroll
  for (int u = 0; u < L1_ROWS; ++u) {
    bag[u] = bag0 + u * groups + sub;
    id[u] = mi_ld_id<IDX>(indices, bag[u] < n_bags ? bag[u] : 0);   // no branch around the load; masked below
  }
  int bad = 0;
  Acc<4> r[L1_ROWS];
#pragma unroll
  for (int u = 0; u < L1_ROWS; ++u) {
    const int b = bag[u] / T;
    tt[u] = bag[u] - b * T;
    const bool present = bag[u] < n_bags;
    const int64_t base = toffs[tt[u]];          // tt < T whatever the bag number
    const int64_t rows = toffs[tt[u] + 1] - base;
    if (present) {
      if (id[u] < 0) {
        if (flags & MI_CHECK_NONNEG) bad |= MI_IDERR_RANGE;
      } else if (id[u] >= rows && (flags & MI_CHECK_RANGE)) {
        bad |= MI_IDERR_RANGE;
      }
    }
    okv[u] = present && id[u] >= 0 && id[u] < rows;
    // the row load carries no branch: a bag without a valid id reads row 0 of the array and is zeroed below
    ld_row<4>(weights + (okv[u] ? base + id[u] : 0) * dim + (int64_t) c * 4, r[u]);
  }
#pragma unroll
  for (int u = 0; u < L1_ROWS; ++u) {
    if (!okv[u]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r[u].v[k] = 0.f;
    }
  }
#pragma unroll
  for (int u = 0; u < L1_ROWS; ++u) {
    if (bag[u] >= n_bags) continue;
    if (bias) {
      Acc<4> bv;
      ld_row<4>(bias + (int64_t) tt[u] * dim + (int64_t) c * 4, bv);
#pragma unroll
      for (int k = 0; k < 4; ++k) r[u].v[k] += bv.v[k];
    }
    const int b = bag[u] / T;
    if constexpr (NT && !BF16) {  // four adjacent dword streaming stores: merged into one 16-byte store
      float* o = reinterpret_cast<float*>(out) + (int64_t) b * out_ld + (int64_t) tt[u] * dim + (int64_t) c * 4;
      __builtin_nontemporal_store(r[u].v[0], o);
      __builtin_nontemporal_store(r[u].v[1], o + 1);
      __builtin_nontemporal_store(r[u].v[2], o + 2);
      __builtin_nontemporal_store(r[u].v[3], o + 3);
    } else
    st_out<4, BF16>(out, (int64_t) b * out_ld + (int64_t) tt[u] * dim + (int64_t) c * 4, r[u]);
  }
  if (bad && c == 0 && err) atomicOr(err, bad);
}

// Multi-hot bags of sixteen ids or more when the batch alone does not fill the chip: ONE WAVE per bag.  The
// lane-group kernel above walks a bag four ids per round, so a 100-id bag is 25 dependent pairs of round trips
// and only many bags in flight hide them.  Here the wave loads 64 of the bag's ids with one coalesced
// instruction, checks them (range against the table, strictly-increasing order against the neighbour lane),
// stages the rows they name in wave-private LDS, and its 64 / lanes lane groups then take the rows WAVE_ROWS at
// a time, all requested before the first is added.  A shuffle tree over the lane groups (a wave64 segment
// reduction: the segments are the columns) leaves the pooled row in group 0.  The order of additions depends
// only on the bag's length.  Needs lanes = dim / 4 to be a power of two <= 64.
constexpr int WAVE_ROWS = 8;
template <typename IDX, bool BF16>
__global__ __launch_bounds__(EMBAG_BLOCK) void embag_fwd_wave_kernel(
    const float* __restrict__ weights, const int64_t* __restrict__ toffs, int T, int dim, int lanes,
    const void* __restrict__ indices, int bag_len, int n_bags, int pool_mode, int flags,
    const float* __restrict__ bias, void* __restrict__ out, int64_t out_ld, int32_t* __restrict__ err) {
  __shared__ int64_t rows_lds[EMBAG_BLOCK / 64][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bag = blockIdx.x * (EMBAG_BLOCK / 64) + w;
  if (bag >= n_bags) return;  // uniform over the wave
  const int G = 64 / lanes, g = lane / lanes, c = lane - g * lanes;
  const int b = bag / T, t = bag - b * T;
  const int64_t base = toffs[t];
  const int64_t rows = toffs[t + 1] - base;
  const int64_t beg = (int64_t) bag * bag_len;
  int64_t* __restrict__ rl = rows_lds[w];
  Acc<4> acc;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc.v[k] = 0.f;
  int count = 0, bad = 0;
  int64_t carry = -1;  // the last non-negative id of the chunks before (order check)
  const float* wcol = weights + (int64_t) c * 4;
  for (int j0 = 0; j0 < bag_len; j0 += 64) {  // uniform
    const int nj = bag_len - j0 < 64 ? bag_len - j0 : 64;
    const bool present = lane < nj;
    const int64_t id = mi_ld_id<IDX>(indices, beg + j0 + (present ? lane : 0));
    const bool nonneg = present && id >= 0;
    if (present && id < 0 && (flags & MI_CHECK_NONNEG)) bad |= MI_IDERR_RANGE;
    if (nonneg && id >= rows && (flags & MI_CHECK_RANGE)) bad |= MI_IDERR_RANGE;
    if (flags & MI_CHECK_INCREASING) {
      // the id before mine that is not padding: the nearest non-negative one in lower lanes, else the carry
      const unsigned long long nn = __ballot(nonneg);
      const unsigned long long below = lane ? (nn & (~0ull >> (64 - lane))) : 0ull;
      const int src = below ? 63 - __clzll((long long) below) : 0;
      const int64_t lo32 = (int64_t) (uint32_t) __shfl((int) (uint32_t) id, src, 64);
      const int64_t hi32 = (int64_t) __shfl((int) (id >> 32), src, 64);
      const int64_t prev = below ? ((hi32 << 32) | lo32) : carry;
      if (nonneg && id <= prev) bad |= MI_IDERR_ORDER;
      const int last = nn ? 63 - __clzll((long long) nn) : 0;
      const int64_t l_lo = (int64_t) (uint32_t) __shfl((int) (uint32_t) id, last, 64);
      const int64_t l_hi = (int64_t) __shfl((int) (id >> 32), last, 64);
      if (nn) carry = (l_hi << 32) | l_lo;
    }
    const bool ok = nonneg && id < rows;  // an id outside the table is never dereferenced
    count += __popcll(__ballot(ok));
    rl[lane] = ok ? base + id : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e0 = 0; e0 < nj; e0 += WAVE_ROWS * G) {  // uniform
      int64_t row[WAVE_ROWS];
      Acc<4> r[WAVE_ROWS];
#pragma unroll
      for (int u = 0; u < WAVE_ROWS; ++u) {
        const int e = e0 + u * G + g;
        row[u] = e < nj ? rl[e] : -1;
        ld_row<4>(wcol + (row[u] >= 0 ? row[u] : 0) * dim, r[u]);  // no branch around the load; masked below
      }
#pragma unroll
      for (int u = 0; u < WAVE_ROWS; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc.v[k] += row[u] >= 0 ? r[u].v[k] : 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the rows are consumed before the next chunk overwrites them
  }
  for (int o = lanes; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc.v[k] += __shfl_xor(acc.v[k], o, 64);
  }
  if (pool_mode == MI_POOL_MEAN && count > 1) {
    const float inv = 1.0f / (float) count;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc.v[k] *= inv;
  }
  if (g == 0) {
    if (bias) {
      Acc<4> bv;
      ld_row<4>(bias + (int64_t) t * dim + (int64_t) c * 4, bv);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc.v[k] += bv.v[k];
    }
    st_out<4, BF16>(out, (int64_t) b * out_ld + (int64_t) t * dim + (int64_t) c * 4, acc);
  }
  bad |= __shfl_xor(bad, 32, 64);
  for (int o = 16; o > 0; o >>= 1) bad |= __shfl_xor(bad, o, 64);
  if (bad && lane == 0 && err) atomicOr(err, bad);
}

template <typename IDX>
__global__ __launch_bounds__(256) void onehot_dense_kernel(const void* __restrict__ indices, int bag_len,
                                                          int n_bags, int size, int flags,
                                                          float* __restrict__ out,
                                                          int32_t* __restrict__ err) {
  // one thread per (bag, column); the ids of a bag are few, so each thread scans them.
  const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t) n_bags * size;
  if (g >= total) return;
  const int bag = (int) (g / size);
  const int col = (int) (g - (int64_t) bag * size);
  float v = 0.f;
  int bad = 0;
  int64_t prev = -1;
  for (int j = 0; j < bag_len; ++j) {
    const int64_t id = mi_ld_id<IDX>(indices, (int64_t) bag * bag_len + j);
    if (id < 0) {
      if (flags & MI_CHECK_NONNEG) bad |= MI_IDERR_RANGE;
      continue;
    }
    if (id >= size && (flags & MI_CHECK_RANGE)) bad |= MI_IDERR_RANGE;
    if (flags & MI_CHECK_INCREASING) {
      if (id <= prev) bad |= MI_IDERR_ORDER;
      prev = id;
    }
    if (id == col) v = 1.0f;
  }
  out[g] = v;
  if (bad && col == 0 && err) atomicOr(err, bad);
}

template <typename IDX>
__global__ __launch_bounds__(256) void validate_kernel(const int64_t* __restrict__ toffs, int T,
                                                      const void* __restrict__ indices,
                                                      const int64_t* __restrict__ offsets, int bag_len,
                                                      int n_bags, int flags, int32_t* __restrict__ err) {
  const int bag = blockIdx.x * blockDim.x + threadIdx.x;
  if (bag >= n_bags) return;
  const int t = bag % T;
  const int64_t rows = toffs[t + 1] - toffs[t];
  int64_t beg, end;
  if (offsets) {
    beg = offsets[bag];
    end = offsets[bag + 1];
  } else {
    beg = (int64_t) bag * bag_len;
    end = beg + bag_len;
  }
  int bad = 0;
  int64_t prev = -1;
  for (int64_t j = beg; j < end; ++j) {
    const int64_t id = mi_ld_id<IDX>(indices, j);
    if (id < 0) {
      if (flags & MI_CHECK_NONNEG) bad |= MI_IDERR_RANGE;
      continue;
    }
    if (id >= rows && (flags & MI_CHECK_RANGE)) bad |= MI_IDERR_RANGE;
    if (flags & MI_CHECK_INCREASING) {
      if (id <= prev) bad |= MI_IDERR_ORDER;
      prev = id;
    }
  }
  if (bad) atomicOr(err, bad);
}

// Several id columns in one launch (blockIdx.y = column): the checks of validate_kernel for columns that each
// hold fixed-length bags of ids of one table, and the id matrix of their concatenation.
struct IdColumns {
  mi_id_column c[MI_MAX_ID_COLUMNS];
  int n;
};

template <typename IDX>
__global__ __launch_bounds__(256) void validate_columns_kernel(IdColumns cols, int32_t* __restrict__ err) {
  const mi_id_column col = cols.c[blockIdx.y];
  const int64_t bag = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (bag >= col.n_bags) return;
  int bad = 0;
  int64_t prev = -1;
  for (int j = 0; j < col.bag_len; ++j) {
    const int64_t id = mi_ld_id<IDX>(col.ids, bag * col.bag_len + j);
    if (id < 0) {
      if (col.check_flags & MI_CHECK_NONNEG) bad |= MI_IDERR_RANGE;
      continue;
    }
    if (id >= col.rows && (col.check_flags & MI_CHECK_RANGE)) bad |= MI_IDERR_RANGE;
    if (col.check_flags & MI_CHECK_INCREASING) {
      if (id <= prev) bad |= MI_IDERR_ORDER;
      prev = id;
    }
  }
  if (bad) atomicOr(err, bad);
}

// out[b] = [ ids of column 0 | ids of column 1 + rows of column 0 | ... ], negative ids (no entry) kept as they are
template <typename IDX>
__global__ __launch_bounds__(256) void concat_columns_kernel(IdColumns cols, int64_t B, int width,
                                                            IDX* __restrict__ out) {
  const int64_t g = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * width) return;
  const int64_t b = g / width;
  int j = (int) (g - b * width);
  int64_t base = 0;
  int i = 0;
  while (i + 1 < cols.n && j >= cols.c[i].bag_len) {  // at most MI_MAX_ID_COLUMNS steps
    j -= cols.c[i].bag_len;
    base += cols.c[i].rows;
    ++i;
  }
  const int64_t id = mi_ld_id<IDX>(cols.c[i].ids, b * cols.c[i].bag_len + j);
  out[g] = (IDX) (id >= 0 ? id + base : id);
}

// One wave per bag: lanes stride over the bag's ids (coalesced), count the valid ones with a wave reduction and
// note the bag against each position.
template <typename IDX>
__global__ __launch_bounds__(256) void bag_info_kernel(const int64_t* __restrict__ toffs, int T,
                                                      const void* __restrict__ indices,
                                                      const int64_t* __restrict__ offsets, int bag_len, int n_bags,
                                                      float* __restrict__ bag_scale, int32_t* __restrict__ entry_bag) {
  const int lane = threadIdx.x & 63;
  const int bag = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bag >= n_bags) return;  // uniform over the wave
  const int t = bag % T;
  const int64_t rows = toffs[t + 1] - toffs[t];
  int64_t beg, end;
  if (offsets) {
    beg = offsets[bag];
    end = offsets[bag + 1];
  } else {
    beg = (int64_t) bag * bag_len;
    end = beg + bag_len;
  }
  int cnt = 0;
  for (int64_t j = beg + lane; j < end; j += 64) {
    const int64_t id = mi_ld_id<IDX>(indices, j);
    cnt += (id >= 0 && id < rows) ? 1 : 0;
    if (entry_bag) entry_bag[j] = bag;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0 && bag_scale) bag_scale[bag] = cnt > 1 ? 1.0f / (float) cnt : 1.0f;
}

template <typename IDX>
int launch_fwd(const float* weights, const int64_t* toffs, int T, int dim, const void* indices,
               const int64_t* offsets, int bag_len, int n_bags, int pool_mode, int flags, const float* bias,
               void* out, int out_dtype, int64_t out_ld, int32_t* err, hipStream_t s) {
  const bool vec4 = (dim % 4 == 0) && (((uintptr_t) weights & 15) == 0) && (out_ld % 4 == 0) &&
                    (((uintptr_t) bias & 15) == 0) &&
                    (((uintptr_t) out & (out_dtype == MI_DT_BF16 ? 7 : 15)) == 0);
  const int vec = vec4 ? 4 : 1;
  const int lanes = dim / vec;
  MI_REQUIRE(lanes >= 1 && lanes <= EMBAG_BLOCK, "mi_embag_fwd: dim %d needs %d lanes per bag (max %d)", dim,
             lanes, EMBAG_BLOCK);
  const int bpb = EMBAG_BLOCK / lanes;
  if (vec4 && !offsets && bag_len == 1 && n_bags >= 65536 && n_bags < (1 << 30)) {
    // one id per bag and enough bags to fill the chip several times over: the batched kernel (its bag
    // numbers run up to a workgroup's worth past n_bags, kept well inside an int)
    // an fp32 output larger than the 256 MB last-level cache is written with streaming stores: its
    // reader cannot find it on chip anyway, and the gathered rows keep the cache
    const bool nt = out_dtype != MI_DT_BF16 && (int64_t) n_bags * dim * 4 > ((int64_t) 256 << 20);
#define L1(BF, NT_)                                                                                                 \
  hipLaunchKernelGGL((embag_fwd_l1_kernel<IDX, BF, NT_>), dim3((unsigned) mi_cdiv(n_bags, bpb * L1_ROWS)),          \
                     dim3(EMBAG_BLOCK), 0, s, weights, toffs, T, dim, lanes, bpb, indices, n_bags, flags, bias, out, \
                     out_ld, err)
    if (out_dtype == MI_DT_BF16) L1(true, false);
    else if (nt) L1(false, true);
    else L1(false, false);
#undef L1
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
  // bags of 16 ids or more: one wave per bag when the launch is too small for the lane-group kernel's serial
  // rounds to hide behind one another (fewer than ~8 waves per SIMD of lane groups), and at any size once a bag
  // gives every lane group six rows or more (measured level or ahead there: +17 % at 100 ids of dim 16, +1-5 %
  // at dim 64 / 128; behind by 3-8 % at 20 ids of dim 64, which this rule leaves to the lane-group kernel)
  // (and at least four rounds of rows per lane group: with 64 / lanes groups a short bag leaves most of a round's
  // loads masked - measured slower than the lane-group kernel at 20 ids of dim 16)
  if (vec4 && !offsets && bag_len >= MI_EMBAG_WAVE_MIN_LEN && (lanes & (lanes - 1)) == 0 && lanes <= 64 &&
      bag_len >= 4 * (64 / lanes) &&
      ((int64_t) n_bags * lanes < (int64_t) MI_EMBAG_WAVE_MAX_THREADS || bag_len >= 6 * (64 / lanes))) {
    const unsigned wgrid = (unsigned) mi_cdiv(n_bags, EMBAG_BLOCK / 64);
#define WAVE(BF)                                                                                                    \
  hipLaunchKernelGGL((embag_fwd_wave_kernel<IDX, BF>), dim3(wgrid), dim3(EMBAG_BLOCK), 0, s, weights, toffs, T, dim, \
                     lanes, indices, bag_len, n_bags, pool_mode, flags, bias, out, out_ld, err)
    if (out_dtype == MI_DT_BF16) WAVE(true); else WAVE(false);
#undef WAVE
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
  const unsigned grid = (unsigned) mi_cdiv(n_bags, bpb);
#define LAUNCH(V, BF)                                                                                   \
  hipLaunchKernelGGL((embag_fwd_kernel<IDX, V, BF>), dim3(grid), dim3(EMBAG_BLOCK), 0, s, weights, toffs, T, \
                     dim, lanes, bpb, indices, offsets, bag_len, n_bags, pool_mode, flags, bias, out, out_ld, err)
  if (out_dtype == MI_DT_BF16) {
    if (vec4) LAUNCH(4, true); else LAUNCH(1, true);
  } else {
    if (vec4) LAUNCH(4, false); else LAUNCH(1, false);
  }
#undef LAUNCH
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace

extern "C" int mi_embag_fwd(const float* weights, const int64_t* table_offsets, int T, int dim,
                            const void* indices, int idx_bits, const int64_t* offsets, int bag_len,
                            int64_t n_bags, int pool_mode, int check_flags, const float* bias, void* out,
                            int out_dtype, int64_t out_ld, int32_t* err, void* stream) {
  if (n_bags == 0) return MI_OK;  // empty batch: nothing to read or write
  MI_REQUIRE(weights && table_offsets && indices && out, "mi_embag_fwd: null pointer");
  MI_REQUIRE(T >= 1 && dim >= 1, "mi_embag_fwd: T=%d dim=%d", T, dim);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_embag_fwd: idx_bits must be 32 or 64");
  MI_REQUIRE(n_bags >= 0 && n_bags < (int64_t) 1 << 31, "mi_embag_fwd: n_bags out of range");
  MI_REQUIRE(n_bags % T == 0, "mi_embag_fwd: n_bags (%lld) must be a multiple of T (%d)", (long long) n_bags, T);
  MI_REQUIRE(offsets || bag_len >= 0, "mi_embag_fwd: bag_len < 0 without offsets");
  MI_REQUIRE(pool_mode == MI_POOL_SUM || pool_mode == MI_POOL_MEAN, "mi_embag_fwd: bad pool_mode");
  MI_REQUIRE(out_dtype == MI_DT_F32 || out_dtype == MI_DT_BF16, "mi_embag_fwd: bad out_dtype");
  MI_REQUIRE(out_ld >= (int64_t) T * dim, "mi_embag_fwd: out_ld %lld < T*dim %lld", (long long) out_ld,
             (long long) T * dim);
  if (n_bags == 0) return MI_OK;
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    return launch_fwd<int32_t>(weights, table_offsets, T, dim, indices, offsets, bag_len, (int) n_bags,
                               pool_mode, check_flags, bias, out, out_dtype, out_ld, err, s);
  return launch_fwd<int64_t>(weights, table_offsets, T, dim, indices, offsets, bag_len, (int) n_bags, pool_mode,
                             check_flags, bias, out, out_dtype, out_ld, err, s);
}

extern "C" int mi_embag_bag_info(const int64_t* table_offsets, int T, const void* indices, int idx_bits,
                                 const int64_t* offsets, int bag_len, int64_t n_bags, float* bag_scale,
                                 int32_t* entry_bag, void* stream) {
  if (n_bags == 0 || (!bag_scale && !entry_bag)) return MI_OK;
  MI_REQUIRE(table_offsets && indices, "mi_embag_bag_info: null pointer");
  MI_REQUIRE(T >= 1, "mi_embag_bag_info: T=%d", T);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_embag_bag_info: idx_bits must be 32 or 64");
  MI_REQUIRE(n_bags > 0 && n_bags < (int64_t) 1 << 31, "mi_embag_bag_info: n_bags out of range");
  MI_REQUIRE(offsets || bag_len >= 0, "mi_embag_bag_info: bag_len < 0 without offsets");
  const unsigned grid = (unsigned) mi_cdiv(n_bags, 4);
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    hipLaunchKernelGGL(bag_info_kernel<int32_t>, dim3(grid), dim3(256), 0, s, table_offsets, T, indices, offsets,
                       bag_len, (int) n_bags, bag_scale, entry_bag);
  else
    hipLaunchKernelGGL(bag_info_kernel<int64_t>, dim3(grid), dim3(256), 0, s, table_offsets, T, indices, offsets,
                       bag_len, (int) n_bags, bag_scale, entry_bag);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_onehot_dense(const void* indices, int idx_bits, int bag_len, int64_t n_bags, int size,
                               int check_flags, float* out, int32_t* err, void* stream) {
  if (n_bags == 0) return MI_OK;
  MI_REQUIRE(indices && out, "mi_onehot_dense: null pointer");
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_onehot_dense: idx_bits must be 32 or 64");
  MI_REQUIRE(bag_len >= 1 && size >= 1, "mi_onehot_dense: bag_len=%d size=%d", bag_len, size);
  MI_REQUIRE(n_bags >= 0 && n_bags < (int64_t) 1 << 31, "mi_onehot_dense: n_bags out of range");
  if (n_bags == 0) return MI_OK;
  const int64_t total = n_bags * size;
  const unsigned grid = (unsigned) mi_cdiv(total, 256);
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    hipLaunchKernelGGL(onehot_dense_kernel<int32_t>, dim3(grid), dim3(256), 0, s, indices, bag_len, (int) n_bags,
                       size, check_flags, out, err);
  else
    hipLaunchKernelGGL(onehot_dense_kernel<int64_t>, dim3(grid), dim3(256), 0, s, indices, bag_len, (int) n_bags,
                       size, check_flags, out, err);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_validate_ids(const int64_t* table_offsets, int T, const void* indices, int idx_bits,
                               const int64_t* offsets, int bag_len, int64_t n_bags, int check_flags,
                               int32_t* err, void* stream) {
  if (n_bags == 0) return MI_OK;
  MI_REQUIRE(table_offsets && indices && err, "mi_validate_ids: null pointer");
  MI_REQUIRE(T >= 1, "mi_validate_ids: T=%d", T);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_validate_ids: idx_bits must be 32 or 64");
  MI_REQUIRE(n_bags >= 0 && n_bags < (int64_t) 1 << 31, "mi_validate_ids: n_bags out of range");
  if (n_bags == 0) return MI_OK;
  const unsigned grid = (unsigned) mi_cdiv(n_bags, 256);
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    hipLaunchKernelGGL(validate_kernel<int32_t>, dim3(grid), dim3(256), 0, s, table_offsets, T, indices, offsets,
                       bag_len, (int) n_bags, check_flags, err);
  else
    hipLaunchKernelGGL(validate_kernel<int64_t>, dim3(grid), dim3(256), 0, s, table_offsets, T, indices, offsets,
                       bag_len, (int) n_bags, check_flags, err);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

namespace {
int pack_columns(const mi_id_column* cols, int n_cols, int idx_bits, const char* who, IdColumns* out) {
  MI_REQUIRE(cols && n_cols >= 1 && n_cols <= MI_MAX_ID_COLUMNS, "%s: 1..%d columns, got %d", who, MI_MAX_ID_COLUMNS,
             n_cols);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "%s: idx_bits must be 32 or 64", who);
  out->n = n_cols;
  for (int i = 0; i < n_cols; ++i) {
    MI_REQUIRE(cols[i].ids || cols[i].n_bags == 0, "%s: column %d has no ids", who, i);
    MI_REQUIRE(cols[i].n_bags >= 0 && cols[i].n_bags < (int64_t) 1 << 31 && cols[i].bag_len >= 1 && cols[i].rows >= 1,
               "%s: column %d: n_bags / bag_len / rows out of range", who, i);
    out->c[i] = cols[i];
  }
  return MI_OK;
}
}  // namespace

extern "C" int mi_validate_id_columns(const mi_id_column* cols, int n_cols, int idx_bits, int32_t* err, void* stream) {
  const char* who = "mi_validate_id_columns";
  IdColumns k;
  int rc = pack_columns(cols, n_cols, idx_bits, who, &k);
  if (rc) return rc;
  MI_REQUIRE(err, "%s: null error word", who);
  int64_t most = 0;
  for (int i = 0; i < n_cols; ++i) most = cols[i].n_bags > most ? cols[i].n_bags : most;
  if (most == 0) return MI_OK;
  const dim3 grid((unsigned) mi_cdiv(most, 256), (unsigned) n_cols);
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    hipLaunchKernelGGL(validate_columns_kernel<int32_t>, grid, dim3(256), 0, s, k, err);
  else
    hipLaunchKernelGGL(validate_columns_kernel<int64_t>, grid, dim3(256), 0, s, k, err);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_concat_id_columns(const mi_id_column* cols, int n_cols, int idx_bits, int64_t B, void* out,
                                    void* stream) {
  const char* who = "mi_concat_id_columns";
  IdColumns k;
  int rc = pack_columns(cols, n_cols, idx_bits, who, &k);
  if (rc) return rc;
  MI_REQUIRE(B >= 0 && B < (int64_t) 1 << 31, "%s: B out of range", who);
  int64_t width = 0;
  for (int i = 0; i < n_cols; ++i) {
    MI_REQUIRE(cols[i].n_bags == B, "%s: column %d has %lld bags, not B = %lld", who, i, (long long) cols[i].n_bags,
               (long long) B);
    width += cols[i].bag_len;
  }
  if (B == 0) return MI_OK;
  MI_REQUIRE(out && width < (int64_t) 1 << 20, "%s: null output / width out of range", who);
  const unsigned grid = (unsigned) mi_cdiv(B * width, 256);
  hipStream_t s = (hipStream_t) stream;
  if (idx_bits == 32)
    hipLaunchKernelGGL(concat_columns_kernel<int32_t>, dim3(grid), dim3(256), 0, s, k, B, (int) width, (int32_t*) out);
  else
    hipLaunchKernelGGL(concat_columns_kernel<int64_t>, dim3(grid), dim3(256), 0, s, k, B, (int) width, (int64_t*) out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

// ---- batch staging -------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void stage_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src,
                                                         size_t n) {
  // eight 16-byte reads per thread in flight: the reads cross PCIe, a round trip of microseconds
  const size_t base = ((size_t) blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t) gridDim.x * 256;
  for (size_t i = base; i < n; i += 8 * stride) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t j = i + u * stride;
      v[u] = src[j < n ? j : i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t j = i + u * stride;
      if (j < n) dst[j] = v[u];
    }
  }
}
}  // namespace

extern "C" int mi_stage_from_host(void* dst, const void* src_host, size_t bytes, void* stream) {
  const char* who = "mi_stage_from_host";
  MI_REQUIRE(dst && src_host, "%s: null pointer", who);
  MI_REQUIRE(bytes % 16 == 0 && ((uintptr_t) dst & 15) == 0 && ((uintptr_t) src_host & 15) == 0,
             "%s: pointers and size must be multiples of 16", who);
  if (bytes == 0) return MI_OK;
  const size_t n = bytes / 16;
  const size_t want = (n + 256 * 8 - 1) / (256 * 8);
  const unsigned grid = (unsigned) (want < 1024 ? want : 1024);
  hipLaunchKernelGGL(stage_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t) stream, (float4*) dst,
                     (const float4*) src_host, n);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
