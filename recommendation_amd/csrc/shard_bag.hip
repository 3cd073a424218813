// Row-wise sharded embedding bags: the kernels around the two all-to-all exchanges of pooled rows.
//
// The reference's only distribution is TensorFlow's parameter server: every variable read and
// every optimizer apply of a worker is a gRPC hop to the ps task
// (recommendation/code/model_trainer.py:124-136, recommendation/train_job.py:41-42).  Here the
// rows of all tables are dealt round robin over the N ranks - global row r = table_offsets[t] + id
// lives on rank r % N as local row r / N - and a step exchanges POOLED rows, not table rows:
//
//   every rank all-gathers the ids (8 bytes an id; the rows are 512 bytes)
//   owner:     for every bag of every rank it holds an id of, the partial pool over ITS ids
//              -> compact blocks of rows, one block per destination rank      (mi_shard_bag_gather)
//   all-to-all of the blocks (fixed capacity per rank pair)
//   requester: pooled[b][t] = sum over owners of the partial rows, rank order  (mi_shard_bag_combine)
//   backward:  the gradient row of every bag goes back to each owner that contributed to it, in the
//              same compact order (mi_shard_bag_pack, all-to-all), and the owner runs the ordinary
//              sort-by-row update on its local rows with the received rows as gradient source.
//
// With one id per bag exactly one owner contributes, so a rank receives B*T rows in all instead
// of N*B*T.  Which owners contribute to a bag, and where a bag's row sits in a block, are pure
// functions of the ids: requester and owner each derive them from the ids they hold
// (mi_shard_bag_slots), nothing but rows crosses the fabric after the id gather.
#include "common.h"

namespace {

constexpr int SB_BLOCK = 256;

__device__ __forceinline__ int64_t local_table_start(int64_t toff, int owner, int N) {
  // rows r < toff with r % N == owner
  return (toff + N - 1 - owner) / N;
}

// Pass 1 over the bags of every part: ownership flags, the owner's local ids, per-chunk counts.
// A "part" is a source rank when owner_fixed >= 0 (owner view: ids holds the ids of all ranks, one
// part after the other) or an owner when owner_fixed < 0 (requester view: ids holds this rank's ids
// only and part p asks which of them rank p owns).
template <typename IDX>
__global__ __launch_bounds__(SB_BLOCK) void bag_flags_kernel(const int64_t* __restrict__ toffs, int T,
                                                            const void* __restrict__ ids, int L, int n_bags,
                                                            int owner_fixed, int N, int32_t* __restrict__ local_ids,
                                                            unsigned char* __restrict__ flags,
                                                            int32_t* __restrict__ chunk_cnt, int n_chunks,
                                                            int32_t* __restrict__ err) {
  __shared__ int wcnt[SB_BLOCK / 64];
  const int p = blockIdx.y;
  const int owner = owner_fixed >= 0 ? owner_fixed : p;
  const int64_t part_off = owner_fixed >= 0 ? (int64_t) p * n_bags : 0;
  const int bag = blockIdx.x * SB_BLOCK + threadIdx.x;
  bool any = false;
  int bad = 0;
  if (bag < n_bags) {
    const int t = bag % T;
    const int64_t base = toffs[t];
    const int64_t rows = toffs[t + 1] - base;
    const int64_t lstart = local_table_start(base, owner, N);
    for (int l = 0; l < L; ++l) {
      const int64_t id = mi_ld_id<IDX>(ids, (part_off + bag) * L + l);
      const bool valid = id >= 0 && id < rows;
      if (id >= rows) bad |= MI_IDERR_RANGE;  // a negative id is a padding entry
      const int64_t row = base + (valid ? id : 0);
      const bool own = valid && (int) (row % N) == owner;
      if (local_ids) local_ids[((int64_t) p * n_bags + bag) * L + l] = own ? (int32_t) (row / N - lstart) : -1;
      any |= own;
    }
    flags[(int64_t) p * n_bags + bag] = any ? 1 : 0;
  }
  const unsigned long long m = __ballot(any);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) chunk_cnt[(int64_t) p * n_chunks + blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
  if (bad && err) atomicOr(err, bad);
}

// Pass 2: a contributing bag's row in its part's block = its rank among the part's contributing
// bags (bag order), found from the chunk counts and a ballot; row_of_bag[p][bag] = p*cap + slot or
// -1, bag_of_row[p*cap + slot] = bag, cnt[p] = rows of the block.  More than cap contributing bags
// in a part: the surplus gets no row and MI_IDERR_CAPACITY is raised.
__global__ __launch_bounds__(SB_BLOCK) void bag_slots_kernel(const unsigned char* __restrict__ flags,
                                                            const int32_t* __restrict__ chunk_cnt, int n_bags,
                                                            int n_chunks, int cap, int32_t* __restrict__ row_of_bag,
                                                            int32_t* __restrict__ bag_of_row,
                                                            int32_t* __restrict__ cnt, int32_t* __restrict__ err) {
  __shared__ int red[SB_BLOCK];
  __shared__ int wbase[SB_BLOCK / 64];
  const int p = blockIdx.y, chunk = blockIdx.x;
  int s = 0;
  for (int i = threadIdx.x; i < chunk; i += SB_BLOCK) s += chunk_cnt[(int64_t) p * n_chunks + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = SB_BLOCK / 2; o > 0; o >>= 1) {
    if ((int) threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const int base = red[0];
  const int bag = chunk * SB_BLOCK + threadIdx.x;
  const bool f = bag < n_bags && flags[(int64_t) p * n_bags + bag] != 0;
  const unsigned long long m = __ballot(f);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wbase[w] = __popcll(m);
  __syncthreads();
  int before = 0;
  for (int i = 0; i < w; ++i) before += wbase[i];
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  const int slot = base + before + __popcll(m & lt);
  if (bag < n_bags) {
    int r = -1;
    if (f) {
      if (slot < cap) {
        r = p * cap + slot;
        bag_of_row[r] = bag;
      } else if (err) {
        atomicOr(err, MI_IDERR_CAPACITY);
      }
    }
    row_of_bag[(int64_t) p * n_bags + bag] = r;
  }
  if (chunk == n_chunks - 1 && threadIdx.x == 0) {
    const int total = base + wbase[0] + wbase[1] + wbase[2] + wbase[3];
    cnt[p] = total < cap ? total : cap;
  }
}

// Owner: one lane group per row of the blocks, GR rows per group in flight.  Row r = p*cap + slot
// is the partial pool of bag bag_of_row[r] of part p over the owner's local ids of that bag.
constexpr int GR = 4;
__global__ __launch_bounds__(SB_BLOCK) void bag_gather_kernel(const float* __restrict__ W,
                                                             const int64_t* __restrict__ ltoffs, int T, int dim,
                                                             int lanes, int groups,
                                                             const int32_t* __restrict__ local_ids, int L, int n_bags,
                                                             int n_parts, int cap,
                                                             const int32_t* __restrict__ bag_of_row,
                                                             const int32_t* __restrict__ cnt,
                                                             float* __restrict__ out_rows) {
  const int sub = threadIdx.x / lanes, c = threadIdx.x - sub * lanes;
  if (sub >= groups) return;
  const int p = blockIdx.y;
  const int n = cnt[p];
  const int slot0 = blockIdx.x * groups * GR;
  if (slot0 >= n) return;  // uniform
  int slot[GR], bag[GR];
  bool live[GR];
#pragma unroll
  for (int u = 0; u < GR; ++u) {
    slot[u] = slot0 + u * groups + sub;
    live[u] = slot[u] < n;
    bag[u] = bag_of_row[(int64_t) p * cap + (live[u] ? slot[u] : 0)];
  }
  float4 acc[GR];
#pragma unroll
  for (int u = 0; u < GR; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < L; ++l) {
    int32_t lid[GR];
#pragma unroll
    for (int u = 0; u < GR; ++u) lid[u] = live[u] ? local_ids[((int64_t) p * n_bags + bag[u]) * L + l] : -1;
    float4 v[GR];
#pragma unroll
    for (int u = 0; u < GR; ++u) {
      const int t = bag[u] % T;
      const int64_t row = ltoffs[t] + (lid[u] >= 0 ? lid[u] : 0);
      v[u] = *reinterpret_cast<const float4*>(W + row * dim + (int64_t) c * 4);
    }
#pragma unroll
    for (int u = 0; u < GR; ++u) {
      if (lid[u] >= 0) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
    }
  }
#pragma unroll
  for (int u = 0; u < GR; ++u) {
    if (live[u]) *reinterpret_cast<float4*>(out_rows + ((int64_t) p * cap + slot[u]) * dim + (int64_t) c * 4) = acc[u];
  }
}

// Requester: pooled row of bag (b, t) = the owners' partial rows added in rank order.
__global__ __launch_bounds__(SB_BLOCK) void bag_combine_kernel(const float* __restrict__ rows,
                                                              const int32_t* __restrict__ row_of_bag, int n_parts,
                                                              int n_bags, int T, int dim, int lanes, int groups,
                                                              float* __restrict__ out, int64_t out_ld) {
  const int sub = threadIdx.x / lanes, c = threadIdx.x - sub * lanes;
  if (sub >= groups) return;
  const int bag = blockIdx.x * groups + sub;
  if (bag >= n_bags) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p0 = 0; p0 < n_parts; p0 += 8) {
    int r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = p0 + u < n_parts ? row_of_bag[(int64_t) (p0 + u) * n_bags + bag] : -1;
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = *reinterpret_cast<const float4*>(rows + (int64_t) (r[u] >= 0 ? r[u] : 0) * dim + (int64_t) c * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (r[u] >= 0) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  const int b = bag / T, t = bag - b * T;
  *reinterpret_cast<float4*>(out + (int64_t) b * out_ld + (int64_t) t * dim + (int64_t) c * 4) = acc;
}

// Requester, backward: row r = p*cap + slot of the blocks that go back to the owners = the gradient
// row of bag bag_of_row[r].
__global__ __launch_bounds__(SB_BLOCK) void bag_pack_kernel(const float* __restrict__ grad, int64_t grad_ld, int T,
                                                           int dim, int lanes, int groups,
                                                           const int32_t* __restrict__ bag_of_row,
                                                           const int32_t* __restrict__ cnt, int cap,
                                                           float* __restrict__ rows_out) {
  const int sub = threadIdx.x / lanes, c = threadIdx.x - sub * lanes;
  if (sub >= groups) return;
  const int p = blockIdx.y;
  const int n = cnt[p];
  const int slot0 = blockIdx.x * groups * GR;
  if (slot0 >= n) return;
  int bag[GR];
  bool live[GR];
#pragma unroll
  for (int u = 0; u < GR; ++u) {
    const int slot = slot0 + u * groups + sub;
    live[u] = slot < n;
    bag[u] = bag_of_row[(int64_t) p * cap + (live[u] ? slot : 0)];
  }
  float4 v[GR];
#pragma unroll
  for (int u = 0; u < GR; ++u) {
    const int b = bag[u] / T, t = bag[u] - b * T;
    v[u] = *reinterpret_cast<const float4*>(grad + (int64_t) b * grad_ld + (int64_t) t * dim + (int64_t) c * 4);
  }
#pragma unroll
  for (int u = 0; u < GR; ++u) {
    const int slot = slot0 + u * groups + sub;
    if (live[u]) *reinterpret_cast<float4*>(rows_out + ((int64_t) p * cap + slot) * dim + (int64_t) c * 4) = v[u];
  }
}

int row_geometry(const char* who, int dim, int* lanes, int* groups) {
  MI_REQUIRE(dim >= 4 && dim % 4 == 0 && dim / 4 <= SB_BLOCK, "%s: dim %d must be a multiple of 4, at most %d", who, dim,
             4 * SB_BLOCK);
  *lanes = dim / 4;
  *groups = SB_BLOCK / *lanes;
  return MI_OK;
}


// ---- bucketed id exchange -----------------------------------------------------------------------------
// Instead of every rank seeing every rank's ids (all-gather: N times the ids at 8 bytes each), a requester sends
// each owner ONLY the ids that owner holds, already as the owner's local rows (global row / N, int32), packed in
// bag order behind per-slot offsets: block p = the CSR form of "my bags that rank p contributes to".  The owner's
// side is then the ordinary CSR embedding bag over its local rows (mi_embag_fwd / mi_embag_bwd_prepare with
// offsets), N*cap bags in all.
//
// Pass 1: ids of each 256-bag chunk that rank p holds, counted over the bags that have a slot in p's block.
template <typename IDX>
__global__ __launch_bounds__(SB_BLOCK) void bag_idcount_kernel(const int64_t* __restrict__ toffs, int T,
                                                              const void* __restrict__ ids, int L, int n_bags, int N,
                                                              const int32_t* __restrict__ row_of_bag,
                                                              int32_t* __restrict__ chunk_ids, int n_chunks) {
  __shared__ int wcnt[SB_BLOCK / 64];
  const int p = blockIdx.y;
  const int bag = blockIdx.x * SB_BLOCK + threadIdx.x;
  int c = 0;
  if (bag < n_bags && row_of_bag[(int64_t) p * n_bags + bag] >= 0) {
    const int t = bag % T;
    const int64_t base = toffs[t], rows = toffs[t + 1] - base;
    for (int l = 0; l < L; ++l) {
      const int64_t id = mi_ld_id<IDX>(ids, (int64_t) bag * L + l);
      c += (id >= 0 && id < rows && (int) ((base + id) % N) == p) ? 1 : 0;
    }
  }
  int w = c;
  for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) chunk_ids[(int64_t) p * n_chunks + blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// Pass 2: every bag with a slot writes its ids of rank p, as local rows, at its place in block p (chunks before it +
// bags before it in the chunk) and its slot's offset - absolute in the owner's receive buffer, where this rank's
// block starts at rank * cap_ids.  Slots beyond the block's last bag get the end of the used part, so that the
// block's unused tail (left at -1 by the caller's fill: padding ids) belongs to no bag but the last, harmlessly.
template <typename IDX>
__global__ __launch_bounds__(SB_BLOCK) void bag_bucket_kernel(const int64_t* __restrict__ toffs, int T,
                                                             const void* __restrict__ ids, int L, int n_bags, int N,
                                                             int rank, int cap, int64_t cap_ids,
                                                             const int32_t* __restrict__ row_of_bag,
                                                             const int32_t* __restrict__ cnt,
                                                             const int32_t* __restrict__ chunk_ids, int n_chunks,
                                                             int32_t* __restrict__ send_ids,
                                                             int64_t* __restrict__ send_off, int32_t* __restrict__ err) {
  __shared__ int red[SB_BLOCK];
  __shared__ int wsum[SB_BLOCK / 64];
  const int p = blockIdx.y, chunk = blockIdx.x;
  int s = 0;
  for (int i = threadIdx.x; i < chunk; i += SB_BLOCK) s += chunk_ids[(int64_t) p * n_chunks + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = SB_BLOCK / 2; o > 0; o >>= 1) {
    if ((int) threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const int base = red[0];
  const int bag = chunk * SB_BLOCK + threadIdx.x;
  const int r = bag < n_bags ? row_of_bag[(int64_t) p * n_bags + bag] : -1;
  int c = 0;
  int64_t tbase = 0, rows = 0;
  if (r >= 0) {
    const int t = bag % T;
    tbase = toffs[t];
    rows = toffs[t + 1] - tbase;
    for (int l = 0; l < L; ++l) {
      const int64_t id = mi_ld_id<IDX>(ids, (int64_t) bag * L + l);
      c += (id >= 0 && id < rows && (int) ((tbase + id) % N) == p) ? 1 : 0;
    }
  }
  // exclusive prefix of c over the chunk's 256 bags
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = c;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int before = 0;
  for (int i = 0; i < w; ++i) before += wsum[i];
  const int64_t pos = (int64_t) base + before + incl - c;
  const int64_t block0 = (int64_t) rank * cap_ids;  // where this rank's block lies in the owner's receive buffer
  if (r >= 0) {
    send_off[(int64_t) p * cap + (r - p * cap)] = block0 + (pos < cap_ids ? pos : cap_ids);
    if (pos + c <= cap_ids) {
      int32_t* dst = send_ids + (int64_t) p * cap_ids + pos;
      int k = 0;
      for (int l = 0; l < L; ++l) {
        const int64_t id = mi_ld_id<IDX>(ids, (int64_t) bag * L + l);
        if (id >= 0 && id < rows && (int) ((tbase + id) % N) == p) dst[k++] = (int32_t) ((tbase + id) / N);
      }
    } else if (err) {
      atomicOr(err, MI_IDERR_CAPACITY);
    }
  }
  if (chunk == n_chunks - 1) {
    const int total = base + wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const int64_t end = block0 + (total < cap_ids ? total : cap_ids);
    for (int slot = cnt[p] + (int) threadIdx.x; slot < cap; slot += SB_BLOCK) send_off[(int64_t) p * cap + slot] = end;
  }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t) 255; }

}  // namespace

extern "C" size_t mi_shard_bag_slots_workspace_bytes(int64_t n_bags, int n_parts) {
  const int64_t n_chunks = mi_cdiv(n_bags > 0 ? n_bags : 1, SB_BLOCK);
  return align256((size_t) n_bags * n_parts) + align256((size_t) n_chunks * n_parts * 4);
}

extern "C" int mi_shard_bag_slots(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int bag_len,
                                  int64_t n_bags, int n_parts, int owner, int n_ranks, int64_t cap,
                                  int32_t* local_ids, int32_t* row_of_bag, int32_t* bag_of_row, int32_t* cnt,
                                  void* workspace, size_t workspace_bytes, int32_t* err, void* stream) {
  const char* who = "mi_shard_bag_slots";
  MI_REQUIRE(table_offsets && ids && row_of_bag && bag_of_row && cnt && workspace, "%s: null pointer", who);
  MI_REQUIRE(T >= 1 && bag_len >= 1 && n_bags >= 1 && n_bags % T == 0 && n_bags < ((int64_t) 1 << 30),
             "%s: bad T / bag_len / n_bags", who);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "%s: idx_bits must be 32 or 64", who);
  MI_REQUIRE(n_ranks >= 1 && n_parts >= 1 && n_parts <= 65535 && owner < n_ranks, "%s: bad n_ranks / n_parts / owner", who);
  MI_REQUIRE(owner >= 0 || n_parts == n_ranks, "%s: the requester view has one part per rank", who);
  MI_REQUIRE(cap >= 1 && cap * n_parts < ((int64_t) 1 << 31), "%s: capacity out of range", who);
  if (workspace_bytes < mi_shard_bag_slots_workspace_bytes(n_bags, n_parts)) {
    mi_set_error("%s: workspace %zu < %zu", who, workspace_bytes, mi_shard_bag_slots_workspace_bytes(n_bags, n_parts));
    return MI_ERR_WORKSPACE;
  }
  const int n_chunks = (int) mi_cdiv(n_bags, SB_BLOCK);
  unsigned char* flags = (unsigned char*) workspace;
  int32_t* chunk_cnt = (int32_t*) ((char*) workspace + align256((size_t) n_bags * n_parts));
  hipStream_t s = (hipStream_t) stream;
  const dim3 grid(n_chunks, n_parts);
  if (idx_bits == 32)
    hipLaunchKernelGGL(bag_flags_kernel<int32_t>, grid, dim3(SB_BLOCK), 0, s, table_offsets, T, ids, bag_len, (int) n_bags,
                       owner, n_ranks, local_ids, flags, chunk_cnt, n_chunks, err);
  else
    hipLaunchKernelGGL(bag_flags_kernel<int64_t>, grid, dim3(SB_BLOCK), 0, s, table_offsets, T, ids, bag_len, (int) n_bags,
                       owner, n_ranks, local_ids, flags, chunk_cnt, n_chunks, err);
  hipLaunchKernelGGL(bag_slots_kernel, grid, dim3(SB_BLOCK), 0, s, flags, chunk_cnt, (int) n_bags, n_chunks, (int) cap,
                     row_of_bag, bag_of_row, cnt, err);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" size_t mi_shard_bag_bucket_workspace_bytes(int64_t n_bags, int n_ranks) {
  return align256((size_t) mi_cdiv(n_bags > 0 ? n_bags : 1, SB_BLOCK) * n_ranks * 4);
}

extern "C" int mi_shard_bag_bucket(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int bag_len,
                                   int64_t n_bags, int rank, int n_ranks, int64_t cap, int64_t cap_ids,
                                   const int32_t* row_of_bag, const int32_t* cnt, int32_t* send_ids, int64_t* send_off,
                                   void* workspace, size_t workspace_bytes, int32_t* err, void* stream) {
  const char* who = "mi_shard_bag_bucket";
  MI_REQUIRE(table_offsets && ids && row_of_bag && cnt && send_ids && send_off && workspace, "%s: null pointer", who);
  MI_REQUIRE(T >= 1 && bag_len >= 1 && n_bags >= 1 && n_bags % T == 0 && n_bags < ((int64_t) 1 << 30),
             "%s: bad T / bag_len / n_bags", who);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "%s: idx_bits must be 32 or 64", who);
  MI_REQUIRE(n_ranks >= 1 && n_ranks <= 65535 && rank >= 0 && rank < n_ranks, "%s: bad rank / n_ranks", who);
  MI_REQUIRE(cap >= 1 && cap * n_ranks < ((int64_t) 1 << 31) && cap_ids >= 1 && cap_ids * n_ranks < ((int64_t) 1 << 31),
             "%s: capacity out of range", who);
  if (workspace_bytes < mi_shard_bag_bucket_workspace_bytes(n_bags, n_ranks)) {
    mi_set_error("%s: workspace %zu < %zu", who, workspace_bytes, mi_shard_bag_bucket_workspace_bytes(n_bags, n_ranks));
    return MI_ERR_WORKSPACE;
  }
  const int n_chunks = (int) mi_cdiv(n_bags, SB_BLOCK);
  int32_t* chunk_ids = (int32_t*) workspace;
  hipStream_t s = (hipStream_t) stream;
  const dim3 grid(n_chunks, n_ranks);
  // every slot of the id blocks starts as a padding id; the kernels below write the live ones
  MI_HIP(hipMemsetAsync(send_ids, 0xFF, (size_t) n_ranks * cap_ids * 4, s));
  if (idx_bits == 32) {
    hipLaunchKernelGGL(bag_idcount_kernel<int32_t>, grid, dim3(SB_BLOCK), 0, s, table_offsets, T, ids, bag_len, (int) n_bags,
                       n_ranks, row_of_bag, chunk_ids, n_chunks);
    hipLaunchKernelGGL(bag_bucket_kernel<int32_t>, grid, dim3(SB_BLOCK), 0, s, table_offsets, T, ids, bag_len, (int) n_bags,
                       n_ranks, rank, (int) cap, cap_ids, row_of_bag, cnt, chunk_ids, n_chunks, send_ids, send_off, err);
  } else {
    hipLaunchKernelGGL(bag_idcount_kernel<int64_t>, grid, dim3(SB_BLOCK), 0, s, table_offsets, T, ids, bag_len, (int) n_bags,
                       n_ranks, row_of_bag, chunk_ids, n_chunks);
    hipLaunchKernelGGL(bag_bucket_kernel<int64_t>, grid, dim3(SB_BLOCK), 0, s, table_offsets, T, ids, bag_len, (int) n_bags,
                       n_ranks, rank, (int) cap, cap_ids, row_of_bag, cnt, chunk_ids, n_chunks, send_ids, send_off, err);
  }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_shard_bag_gather(const float* weights, const int64_t* local_table_offsets, int T, int dim,
                                   const int32_t* local_ids, int bag_len, int64_t n_bags, int n_parts, int64_t cap,
                                   const int32_t* bag_of_row, const int32_t* cnt, float* out_rows, void* stream) {
  const char* who = "mi_shard_bag_gather";
  MI_REQUIRE(weights && local_table_offsets && local_ids && bag_of_row && cnt && out_rows, "%s: null pointer", who);
  MI_REQUIRE(T >= 1 && bag_len >= 1 && n_bags >= 1 && n_parts >= 1 && n_parts <= 65535 && cap >= 1, "%s: bad shape", who);
  MI_REQUIRE((((uintptr_t) weights | (uintptr_t) out_rows) & 15) == 0, "%s: rows must be 16-byte aligned", who);
  int lanes, groups;
  const int rc = row_geometry(who, dim, &lanes, &groups);
  if (rc) return rc;
  const dim3 grid((unsigned) mi_cdiv(cap, (int64_t) groups * GR), n_parts);
  hipLaunchKernelGGL(bag_gather_kernel, grid, dim3(SB_BLOCK), 0, (hipStream_t) stream, weights, local_table_offsets, T, dim,
                     lanes, groups, local_ids, bag_len, (int) n_bags, n_parts, (int) cap, bag_of_row, cnt, out_rows);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_shard_bag_combine(const float* rows, const int32_t* row_of_bag, int n_parts, int64_t n_bags, int T,
                                    int dim, float* out, int64_t out_ld, void* stream) {
  const char* who = "mi_shard_bag_combine";
  MI_REQUIRE(rows && row_of_bag && out, "%s: null pointer", who);
  MI_REQUIRE(T >= 1 && n_bags >= 1 && n_bags % T == 0 && n_parts >= 1 && out_ld >= (int64_t) T * dim && out_ld % 4 == 0,
             "%s: bad shape", who);
  MI_REQUIRE((((uintptr_t) rows | (uintptr_t) out) & 15) == 0, "%s: rows must be 16-byte aligned", who);
  int lanes, groups;
  const int rc = row_geometry(who, dim, &lanes, &groups);
  if (rc) return rc;
  hipLaunchKernelGGL(bag_combine_kernel, dim3((unsigned) mi_cdiv(n_bags, groups)), dim3(SB_BLOCK), 0, (hipStream_t) stream,
                     rows, row_of_bag, n_parts, (int) n_bags, T, dim, lanes, groups, out, out_ld);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_shard_bag_pack(const float* grad, int64_t grad_ld, int T, int dim, const int32_t* bag_of_row,
                                 const int32_t* cnt, int n_parts, int64_t cap, float* rows_out, void* stream) {
  const char* who = "mi_shard_bag_pack";
  MI_REQUIRE(grad && bag_of_row && cnt && rows_out, "%s: null pointer", who);
  MI_REQUIRE(T >= 1 && n_parts >= 1 && n_parts <= 65535 && cap >= 1 && grad_ld >= (int64_t) T * dim && grad_ld % 4 == 0,
             "%s: bad shape", who);
  MI_REQUIRE((((uintptr_t) grad | (uintptr_t) rows_out) & 15) == 0, "%s: rows must be 16-byte aligned", who);
  int lanes, groups;
  const int rc = row_geometry(who, dim, &lanes, &groups);
  if (rc) return rc;
  const dim3 grid((unsigned) mi_cdiv(cap, (int64_t) groups * GR), n_parts);
  hipLaunchKernelGGL(bag_pack_kernel, grid, dim3(SB_BLOCK), 0, (hipStream_t) stream, grad, grad_ld, T, dim, lanes, groups,
                     bag_of_row, cnt, (int) cap, rows_out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
