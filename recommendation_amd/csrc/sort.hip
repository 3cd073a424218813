// Deterministic sort-by-row for the sparse backward: stable LSD radix sort of
// (global row, entry position) pairs, then run detection.
//
// TensorFlow builds the gradient of a first-layer kernel densely as X^T dY with a
// one-hot X (recommendation/code/model_trainer.py:157-158).  The sparse equivalent needs,
// for every distinct row, the list of entries that touched it, in a reproducible order:
// that list is what this file produces (== numpy's stable argsort + unique).
//
// Radix sort, 8 bits per pass, three launches per pass:
//   hist    : per-tile digit histogram (LDS integer atomics: order-independent counts)
//   scan    : exclusive scan of the digit-major [256][tiles] table (one workgroup)
//   scatter : per tile, stable rank of every key among equal digits (wave-level digit
//             matching by ballots, wave64), plus the scanned base -> final position.
// Keys are uint32; a pass costs 4+4 B read and 4+4 B written per pair for scatter and
// 4 B read for hist: 20 B per pair per pass.
#include "common.h"

namespace {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_BLOCK = 256;
constexpr int SORT_WAVES = SORT_BLOCK / MI_WAVE;
constexpr int ITEMS = 8;
constexpr int TILE = SORT_BLOCK * ITEMS;  // 2048 pairs per workgroup
constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(SORT_BLOCK) void hist_kernel(const uint32_t* __restrict__ keys, int n, int shift,
                                                         uint32_t* __restrict__ hist, int tiles) {
  __shared__ uint32_t h[RADIX];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * TILE;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int idx = base + i * SORT_BLOCK + threadIdx.x;
    if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & (RADIX - 1)], 1u);
  }
  __syncthreads();
  hist[(size_t) threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];
}

// Exclusive scan of `total` uint32 in place by ONE workgroup; optionally also stores the
// grand total.
__global__ __launch_bounds__(SCAN_BLOCK) void scan_kernel(uint32_t* __restrict__ data, int total,
                                                         uint32_t* __restrict__ grand_total) {
  __shared__ uint32_t wsum[SCAN_BLOCK / MI_WAVE];
  const int tid = threadIdx.x;
  const int chunk = (total + SCAN_BLOCK - 1) / SCAN_BLOCK;
  const int beg = min(tid * chunk, total);
  const int end = min(beg + chunk, total);
  uint32_t s = 0;
  for (int i = beg; i < end; ++i) s += data[i];
  // inclusive scan across the wave
  const int lane = tid & 63, w = tid >> 6;
  uint32_t inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  if (w == 0) {
    uint32_t v = (lane < SCAN_BLOCK / MI_WAVE) ? wsum[lane] : 0;
    uint32_t vi = v;
#pragma unroll
    for (int o = 1; o < SCAN_BLOCK / MI_WAVE; o <<= 1) {
      const uint32_t t = __shfl_up(vi, o, 64);
      if (lane >= o) vi += t;
    }
    if (lane < SCAN_BLOCK / MI_WAVE) wsum[lane] = vi - v;  // exclusive wave base
    if (lane == SCAN_BLOCK / MI_WAVE - 1 && grand_total) *grand_total = vi;
  }
  __syncthreads();
  uint32_t run = wsum[w] + inc - s;
  for (int i = beg; i < end; ++i) {
    const uint32_t t = data[i];
    data[i] = run;
    run += t;
  }
}

__global__ __launch_bounds__(SORT_BLOCK) void scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out, int n, int shift,
                                                            const uint32_t* __restrict__ hist, int tiles) {
  __shared__ uint32_t cnt[SORT_WAVES][RADIX];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int i = 0; i < SORT_WAVES; ++i) cnt[i][tid] = 0;
  __syncthreads();
  const int wbase = blockIdx.x * TILE + w * (MI_WAVE * ITEMS);
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t key[ITEMS], loc[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int idx = wbase + j * MI_WAVE + lane;
    const bool valid = idx < n;
    key[j] = valid ? keys_in[idx] : 0xFFFFFFFFu;
    const uint32_t d = (key[j] >> shift) & (RADIX - 1);
    unsigned long long mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      mask &= bit ? bal : ~bal;
    }
    const uint32_t before = (uint32_t) __popcll(mask & lt);
    const uint32_t old = cnt[w][d];
    if (valid && before == 0) cnt[w][d] = old + (uint32_t) __popcll(mask);
    loc[j] = old + before;
  }
  __syncthreads();
  {
    // digit `tid`: global base of this tile, then running base per wave
    uint32_t run = hist[(size_t) tid * tiles + blockIdx.x];
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) {
      const uint32_t t = cnt[i][tid];
      cnt[i][tid] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int idx = wbase + j * MI_WAVE + lane;
    if (idx < n) {
      const uint32_t d = (key[j] >> shift) & (RADIX - 1);
      const uint32_t pos = cnt[w][d] + loc[j];
      keys_out[pos] = key[j];
      vals_out[pos] = vals_in[idx];
    }
  }
}

struct SortWs {
  uint32_t* keys_alt;
  uint32_t* vals_alt;
  uint32_t* hist;
  uint32_t* seg_blocks;  // per-block head counts, then valid counts
  int tiles;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t) 255; }

size_t sort_ws_bytes(int64_t n) {
  const int64_t tiles = mi_cdiv(n > 0 ? n : 1, TILE);
  return align256((size_t) n * 4) * 2 + align256((size_t) RADIX * tiles * 4) + align256((size_t) (2 * tiles + 2) * 4);
}

SortWs carve(void* ws, int64_t n) {
  SortWs s;
  char* p = (char*) ws;
  s.tiles = (int) mi_cdiv(n > 0 ? n : 1, TILE);
  s.keys_alt = (uint32_t*) p;
  p += align256((size_t) n * 4);
  s.vals_alt = (uint32_t*) p;
  p += align256((size_t) n * 4);
  s.hist = (uint32_t*) p;
  p += align256((size_t) RADIX * s.tiles * 4);
  s.seg_blocks = (uint32_t*) p;
  return s;
}

// Sorts (keys, vals); result ends in (keys, vals).
int radix_sort(uint32_t* keys, uint32_t* vals, int n, int key_bits, const SortWs& ws, hipStream_t s) {
  if (n <= 1) return MI_OK;
  const int passes = (int) mi_cdiv(key_bits > 0 ? key_bits : 1, RADIX_BITS);
  uint32_t *ki = keys, *vi = vals, *ko = ws.keys_alt, *vo = ws.vals_alt;
  for (int p = 0; p < passes; ++p) {
    const int shift = p * RADIX_BITS;
    hipLaunchKernelGGL(hist_kernel, dim3(ws.tiles), dim3(SORT_BLOCK), 0, s, ki, n, shift, ws.hist, ws.tiles);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws.hist, RADIX * ws.tiles, (uint32_t*) nullptr);
    hipLaunchKernelGGL(scatter_kernel, dim3(ws.tiles), dim3(SORT_BLOCK), 0, s, ki, vi, ko, vo, n, shift, ws.hist,
                       ws.tiles);
    uint32_t* t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
  }
  MI_LAUNCH_CHECK();
  if (ki != keys) {
    MI_HIP(hipMemcpyAsync(keys, ki, (size_t) n * 4, hipMemcpyDeviceToDevice, s));
    MI_HIP(hipMemcpyAsync(vals, vi, (size_t) n * 4, hipMemcpyDeviceToDevice, s));
  }
  return MI_OK;
}

// ---- key generation -------------------------------------------------------------------
template <typename IDX>
__global__ __launch_bounds__(256) void keygen_kernel(const int64_t* __restrict__ toffs, int T,
                                                    const void* __restrict__ indices,
                                                    const int64_t* __restrict__ offsets, int bag_len, int n_bags,
                                                    int n, uint32_t sentinel, uint32_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int bag;
  if (offsets) {
    // largest bag with offsets[bag] <= p
    int lo = 0, hi = n_bags;  // offsets[0] = 0 <= p < offsets[n_bags]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= p) lo = mid; else hi = mid;
    }
    bag = lo;
  } else {
    bag = p / bag_len;
  }
  const int t = bag % T;
  const int64_t base = toffs[t];
  const int64_t rows = toffs[t + 1] - base;
  const int64_t id = mi_ld_id<IDX>(indices, p);
  keys[p] = (id >= 0 && id < rows) ? (uint32_t) (base + id) : sentinel;
  vals[p] = (uint32_t) p;
}

// ---- run detection --------------------------------------------------------------------
// A "head" is a sorted entry whose row differs from its predecessor's (sentinels excluded).
constexpr int SEG_ITEMS = 8;
constexpr int SEG_TILE = 256 * SEG_ITEMS;

__device__ __forceinline__ bool is_head(const uint32_t* keys, int i, uint32_t sentinel) {
  const uint32_t k = keys[i];
  if (k == sentinel) return false;
  return i == 0 || keys[i - 1] != k;
}

__global__ __launch_bounds__(256) void seg_count_kernel(const uint32_t* __restrict__ keys, int n,
                                                       uint32_t sentinel, uint32_t* __restrict__ blocks,
                                                       int nblk) {
  __shared__ uint32_t sh[2][4];
  const int base = blockIdx.x * SEG_TILE + threadIdx.x * SEG_ITEMS;
  uint32_t heads = 0, valid = 0;
#pragma unroll
  for (int i = 0; i < SEG_ITEMS; ++i) {
    const int idx = base + i;
    if (idx < n) {
      heads += is_head(keys, idx, sentinel) ? 1u : 0u;
      valid += keys[idx] != sentinel ? 1u : 0u;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    heads += __shfl_xor(heads, o, 64);
    valid += __shfl_xor(valid, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = heads; sh[1][w] = valid; }
  __syncthreads();
  if (threadIdx.x == 0) {
    blocks[blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    blocks[nblk + blockIdx.x] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

// one workgroup: exclusive scan of head counts, totals -> counts[], closing seg offset
__global__ __launch_bounds__(SCAN_BLOCK) void seg_scan_kernel(uint32_t* __restrict__ blocks, int nblk,
                                                             int32_t* __restrict__ counts,
                                                             int32_t* __restrict__ seg_offsets) {
  __shared__ uint32_t wsum[2][SCAN_BLOCK / MI_WAVE];
  const int tid = threadIdx.x;
  const int chunk = (nblk + SCAN_BLOCK - 1) / SCAN_BLOCK;
  const int beg = min(tid * chunk, nblk);
  const int end = min(beg + chunk, nblk);
  uint32_t s = 0, sv = 0;
  for (int i = beg; i < end; ++i) { s += blocks[i]; sv += blocks[nblk + i]; }
  const int lane = tid & 63, w = tid >> 6;
  uint32_t inc = s, incv = sv;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    const uint32_t tv = __shfl_up(incv, o, 64);
    if (lane >= o) { inc += t; incv += tv; }
  }
  if (lane == 63) { wsum[0][w] = inc; wsum[1][w] = incv; }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0, runv = 0;
    for (int i = 0; i < SCAN_BLOCK / MI_WAVE; ++i) {
      const uint32_t t = wsum[0][i]; wsum[0][i] = run; run += t;
      runv += wsum[1][i];
    }
    counts[0] = (int32_t) run;
    counts[1] = (int32_t) runv;
    seg_offsets[run] = (int32_t) runv;
  }
  __syncthreads();
  uint32_t run = wsum[0][w] + inc - s;
  for (int i = beg; i < end; ++i) {
    const uint32_t t = blocks[i];
    blocks[i] = run;
    run += t;
  }
}

__global__ __launch_bounds__(256) void seg_write_kernel(const uint32_t* __restrict__ keys, int n,
                                                       uint32_t sentinel, const uint32_t* __restrict__ blocks,
                                                       uint32_t* __restrict__ unique_rows,
                                                       int32_t* __restrict__ seg_offsets) {
  __shared__ uint32_t wsum[4];
  const int base = blockIdx.x * SEG_TILE + threadIdx.x * SEG_ITEMS;
  bool head[SEG_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SEG_ITEMS; ++i) {
    const int idx = base + i;
    head[i] = idx < n && is_head(keys, idx, sentinel);
    s += head[i] ? 1u : 0u;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t wbase = 0;
  for (int i = 0; i < w; ++i) wbase += wsum[i];
  uint32_t u = blocks[blockIdx.x] + wbase + inc - s;
#pragma unroll
  for (int i = 0; i < SEG_ITEMS; ++i) {
    if (head[i]) {
      unique_rows[u] = keys[base + i];
      seg_offsets[u] = base + i;
      ++u;
    }
  }
}

int bit_width_u64(uint64_t v) {
  int b = 0;
  while (v) { ++b; v >>= 1; }
  return b;
}

}  // namespace

extern "C" size_t mi_embag_bwd_workspace_bytes(int64_t n_entries) { return sort_ws_bytes(n_entries); }

extern "C" int mi_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int key_bits, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  MI_REQUIRE(n >= 0 && n < ((int64_t) 1 << 31) - TILE, "mi_sort_pairs_u32: n out of range");
  MI_REQUIRE(key_bits >= 0 && key_bits <= 32, "mi_sort_pairs_u32: key_bits must be 0..32");
  if (n <= 1) return MI_OK;
  MI_REQUIRE(keys && vals && workspace, "mi_sort_pairs_u32: null pointer");
  if (workspace_bytes < sort_ws_bytes(n)) {
    mi_set_error("mi_sort_pairs_u32: workspace %zu < %zu", workspace_bytes, sort_ws_bytes(n));
    return MI_ERR_WORKSPACE;
  }
  return radix_sort(keys, vals, (int) n, key_bits, carve(workspace, n), (hipStream_t) stream);
}

extern "C" int mi_embag_bwd_prepare(const int64_t* table_offsets, int T, int64_t total_rows, const void* indices,
                                    int idx_bits, const int64_t* offsets, int bag_len, int64_t n_bags,
                                    int64_t n_entries, uint32_t* sorted_rows, int32_t* perm,
                                    uint32_t* unique_rows, int32_t* seg_offsets, int32_t* counts, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  MI_REQUIRE(table_offsets && indices && sorted_rows && perm && unique_rows && seg_offsets && counts,
             "mi_embag_bwd_prepare: null pointer");
  MI_REQUIRE(T >= 1 && n_bags >= 0 && n_bags % T == 0, "mi_embag_bwd_prepare: n_bags must be a multiple of T");
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_embag_bwd_prepare: idx_bits must be 32 or 64");
  MI_REQUIRE(n_entries >= 0 && n_entries < ((int64_t) 1 << 31) - TILE, "mi_embag_bwd_prepare: n_entries out of range");
  MI_REQUIRE(offsets || (bag_len >= 1 && n_entries == n_bags * bag_len),
             "mi_embag_bwd_prepare: n_entries must equal n_bags*bag_len for fixed-width bags");
  if (total_rows < 1 || total_rows >= (int64_t) 0xFFFFFFFFll) {
    mi_set_error("mi_embag_bwd_prepare: total_rows %lld does not fit the uint32 row key", (long long) total_rows);
    return MI_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t) stream;
  if (n_entries == 0) {
    MI_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s));
    MI_HIP(hipMemsetAsync(seg_offsets, 0, sizeof(int32_t), s));
    return MI_OK;
  }
  MI_REQUIRE(workspace, "mi_embag_bwd_prepare: null workspace");
  if (workspace_bytes < sort_ws_bytes(n_entries)) {
    mi_set_error("mi_embag_bwd_prepare: workspace %zu < %zu", workspace_bytes, sort_ws_bytes(n_entries));
    return MI_ERR_WORKSPACE;
  }
  const int n = (int) n_entries;
  const SortWs ws = carve(workspace, n_entries);
  const uint32_t sentinel = (uint32_t) total_rows;  // greater than every valid row
  const unsigned kgrid = (unsigned) mi_cdiv(n, 256);
  uint32_t* vals = reinterpret_cast<uint32_t*>(perm);
  if (idx_bits == 32)
    hipLaunchKernelGGL(keygen_kernel<int32_t>, dim3(kgrid), dim3(256), 0, s, table_offsets, T, indices, offsets,
                       bag_len, (int) n_bags, n, sentinel, sorted_rows, vals);
  else
    hipLaunchKernelGGL(keygen_kernel<int64_t>, dim3(kgrid), dim3(256), 0, s, table_offsets, T, indices, offsets,
                       bag_len, (int) n_bags, n, sentinel, sorted_rows, vals);
  MI_LAUNCH_CHECK();
  const int rc = radix_sort(sorted_rows, vals, n, bit_width_u64((uint64_t) total_rows), ws, s);
  if (rc != MI_OK) return rc;
  const int nblk = (int) mi_cdiv(n, SEG_TILE);  // == ws.tiles (same tile size)
  hipLaunchKernelGGL(seg_count_kernel, dim3(nblk), dim3(256), 0, s, sorted_rows, n, sentinel, ws.seg_blocks, nblk);
  hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws.seg_blocks, nblk, counts, seg_offsets);
  hipLaunchKernelGGL(seg_write_kernel, dim3(nblk), dim3(256), 0, s, sorted_rows, n, sentinel, ws.seg_blocks,
                     unique_rows, seg_offsets);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
