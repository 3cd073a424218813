// Deterministic sort-by-row for the sparse backward: stable LSD radix sort of
// (global row, entry position) pairs, then run detection.
//
// TensorFlow builds the gradient of a first-layer kernel densely as X^T dY with a
// one-hot X (recommendation/code/model_trainer.py:157-158).  The sparse equivalent needs,
// for every distinct row, the list of entries that touched it, in a reproducible order:
// that list is what this file produces (== numpy's stable argsort + unique).
//
// Radix sort, 8- or 9-bit digits, three launches per pass:
//   hist       : per-tile digit histogram (LDS integer atomics: order-independent counts)
//   digit_scan : one workgroup per digit scans that digit's per-tile counts (a contiguous
//                row of the digit-major [radix][tiles] table) and stores the digit total
//   scatter    : per tile, stable rank of every key among equal digits (wave-level digit
//                matching by ballots, wave64) + digit base + tile offset -> final position;
//                the pairs pass through LDS in the tile's digit-sorted order so that the stores
//                of a wave are contiguous runs.
// Keys are uint32; a pass costs 4+4 B read and 4+4 B written per pair for scatter and
// 4 B read for hist: 20 B per pair per pass.
#include "common.h"
#include <mutex>

namespace {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_WAVES = SORT_BLOCK / MI_WAVE;
constexpr int SMALL_N = 1 << 19;  // up to here: 4 pairs per thread (1024 per workgroup), else 8 (2048)
constexpr int MAX_RADIX = 512;
constexpr int SCAN_BLOCK = 1024;

// inclusive scan over the 256 threads of a workgroup; `total` gets the workgroup sum
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* wsum /*[4]*/, uint32_t& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // wsum may still be read from a previous call
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int i = 0; i < SORT_WAVES; ++i) {
    if (i < w) base += wsum[i];
  }
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + inc;
}

template <int RB, int ITEMS>
__global__ __launch_bounds__(SORT_BLOCK) void hist_kernel(const uint32_t* __restrict__ keys, int n, int shift,
                                                         uint32_t* __restrict__ hist, int tiles) {
  constexpr int RADIX = 1 << RB;
  constexpr int TILE = SORT_BLOCK * ITEMS;
  __shared__ uint32_t h[RADIX];
  for (int d = threadIdx.x; d < RADIX; d += SORT_BLOCK) h[d] = 0;
  __syncthreads();
  const int base = blockIdx.x * TILE;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int idx = base + i * SORT_BLOCK + threadIdx.x;
    if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & (RADIX - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RADIX; d += SORT_BLOCK) hist[(size_t) d * tiles + blockIdx.x] = h[d];
}

// One workgroup per digit: exclusive scan of that digit's per-tile counts (a contiguous,
// coalesced row of the digit-major table), and the digit's total.
__global__ __launch_bounds__(SORT_BLOCK) void digit_scan_kernel(uint32_t* __restrict__ hist, int tiles,
                                                               uint32_t* __restrict__ totals) {
  __shared__ uint32_t wsum[SORT_WAVES];
  uint32_t* row = hist + (size_t) blockIdx.x * tiles;
  uint32_t carry = 0;
  for (int base = 0; base < tiles; base += SORT_BLOCK) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < tiles ? row[i] : 0u;
    uint32_t total;
    const uint32_t inc = block_inclusive_scan(v, wsum, total);
    if (i < tiles) row[i] = carry + inc - v;
    carry += total;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <int RB, int ITEMS>
__global__ __launch_bounds__(SORT_BLOCK) void scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out, int n, int shift,
                                                            const uint32_t* __restrict__ hist, int tiles,
                                                            const uint32_t* __restrict__ totals) {
  constexpr int RADIX = 1 << RB;
  constexpr int PER = RADIX / SORT_BLOCK;  // digits owned by one thread (consecutive)
  constexpr int TILE = SORT_BLOCK * ITEMS;
  __shared__ uint32_t cnt[SORT_WAVES][RADIX];
  __shared__ uint32_t wsum[SORT_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int d = tid; d < RADIX; d += SORT_BLOCK) {
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) cnt[i][d] = 0;
  }
  // every global load this workgroup needs is issued up front: its pairs, the digit totals and
  // this tile's offsets (one memory round trip, not three)
  const int wbase = blockIdx.x * TILE + w * (MI_WAVE * ITEMS);
  uint32_t key[ITEMS], val[ITEMS], loc[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int idx = wbase + j * MI_WAVE + lane;
    key[j] = idx < n ? keys_in[idx] : 0xFFFFFFFFu;
    val[j] = idx < n ? vals_in[idx] : 0u;
  }
  uint32_t tot[PER], toff[PER], tsum = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    tot[q] = totals[tid * PER + q];
    toff[q] = hist[(size_t) (tid * PER + q) * tiles + blockIdx.x];
    tsum += tot[q];
  }
  // digit bases = exclusive scan of the digit totals (every workgroup redoes this small scan)
  uint32_t grand;
  uint32_t dbase = block_inclusive_scan(tsum, wsum, grand) - tsum;  // also orders the cnt[] zeroing
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int idx = wbase + j * MI_WAVE + lane;
    const bool valid = idx < n;
    const uint32_t d = (key[j] >> shift) & (RADIX - 1);
    // lanes holding the same digit (see table_sort_kernel for the bit trick)
    const unsigned long long vb = __ballot(valid);
    uint32_t mlo = (uint32_t) vb, mhi = (uint32_t) (vb >> 32);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int x = ((int) (d << (31 - b))) >> 31;
      const unsigned long long bal = __ballot(x != 0);
      mlo &= ~((uint32_t) bal ^ (uint32_t) x);
      mhi &= ~((uint32_t) (bal >> 32) ^ (uint32_t) x);
    }
    const uint32_t before = (uint32_t) (__popc(mlo & (uint32_t) lt) + __popc(mhi & (uint32_t) (lt >> 32)));
    const uint32_t old = cnt[w][d];
    if (valid && before == 0) cnt[w][d] = old + (uint32_t) (__popc(mlo) + __popc(mhi));
    loc[j] = old + before;
  }
  __syncthreads();
  // Per digit: where the tile's entries of that digit start in the OUTPUT (digit base + this tile's offset inside
  // the digit) and where they start in the tile's own digit-sorted order.  The entries pass through LDS in that
  // local order and are written out by consecutive threads: entries of one digit are then neighbours both in LDS
  // and in the output, so a wave's stores are a few contiguous runs instead of 64 scattered words.
  __shared__ uint32_t delta[RADIX];          // output position minus local position, per digit
  __shared__ uint32_t skey[TILE], sval[TILE];
  uint32_t wc[PER][SORT_WAVES], lsum = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) {
      wc[q][i] = cnt[i][tid * PER + q];
      lsum += wc[q][i];
    }
  }
  uint32_t lgrand;
  uint32_t lrun = block_inclusive_scan(lsum, wsum, lgrand) - lsum;  // local exclusive prefix over the digits
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int d = tid * PER + q;
    delta[d] = dbase + toff[q] - lrun;
    dbase += tot[q];
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) {  // a running LOCAL base per wave
      cnt[i][d] = lrun;
      lrun += wc[q][i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int idx = wbase + j * MI_WAVE + lane;
    if (idx < n) {
      const uint32_t d = (key[j] >> shift) & (RADIX - 1);
      const uint32_t lp = cnt[w][d] + loc[j];
      skey[lp] = key[j];
      sval[lp] = val[j];
    }
  }
  __syncthreads();
  const int nv = min(TILE, n - (int) blockIdx.x * TILE);  // valid entries of this tile
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int i = j * SORT_BLOCK + tid;
    if (i < nv) {
      const uint32_t k = skey[i];
      const uint32_t pos = delta[(k >> shift) & (RADIX - 1)] + (uint32_t) i;
      keys_out[pos] = k;
      vals_out[pos] = sval[i];
    }
  }
}

struct SortWs {
  uint32_t* keys_alt;
  uint32_t* vals_alt;
  uint32_t* hist;
  uint32_t* totals;
  uint32_t* seg_blocks;  // per-block head counts, then valid counts
  int tiles;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t) 255; }

int items_for(int64_t n) { return n <= SMALL_N ? 4 : 8; }
int64_t tiles_for(int64_t n) { return mi_cdiv(n > 0 ? n : 1, (int64_t) SORT_BLOCK * items_for(n)); }

size_t sort_ws_bytes(int64_t n) {
  const int64_t tiles = tiles_for(n);
  return align256((size_t) n * 4) * 2 + align256((size_t) MAX_RADIX * tiles * 4) + align256(MAX_RADIX * 4) +
         align256((size_t) (2 * tiles + 2) * 4);
}

SortWs carve(void* ws, int64_t n) {
  SortWs s;
  char* p = (char*) ws;
  s.tiles = (int) tiles_for(n);
  s.keys_alt = (uint32_t*) p;
  p += align256((size_t) n * 4);
  s.vals_alt = (uint32_t*) p;
  p += align256((size_t) n * 4);
  s.hist = (uint32_t*) p;
  p += align256((size_t) MAX_RADIX * s.tiles * 4);
  s.totals = (uint32_t*) p;
  p += align256(MAX_RADIX * 4);
  s.seg_blocks = (uint32_t*) p;
  return s;
}

// digits of 8 bits, or 9 where that saves a whole pass (e.g. 26-bit rows: 3 passes, not 4)
void pass_plan(int key_bits, int* passes, int* rb) {
  if (key_bits < 1) key_bits = 1;
  *passes = (int) mi_cdiv(key_bits, 9);
  *rb = mi_cdiv(key_bits, *passes) <= 8 ? 8 : 9;
}

template <int RB, int ITEMS>
void launch_pass(const uint32_t* ki, const uint32_t* vi, uint32_t* ko, uint32_t* vo, int n, int shift,
                 const SortWs& ws, hipStream_t s) {
  hipLaunchKernelGGL((hist_kernel<RB, ITEMS>), dim3(ws.tiles), dim3(SORT_BLOCK), 0, s, ki, n, shift, ws.hist,
                     ws.tiles);
  hipLaunchKernelGGL(digit_scan_kernel, dim3(1 << RB), dim3(SORT_BLOCK), 0, s, ws.hist, ws.tiles, ws.totals);
  hipLaunchKernelGGL((scatter_kernel<RB, ITEMS>), dim3(ws.tiles), dim3(SORT_BLOCK), 0, s, ki, vi, ko, vo, n, shift,
                     ws.hist, ws.tiles, ws.totals);
}

// Sorts (keys, vals); result ends in (keys, vals).  With `input_in_alt` the caller has put the
// unsorted pairs in the workspace's alternate buffers instead (it does so when the number of
// passes is odd, which makes the last pass land in (keys, vals) without a copy).
int radix_sort(uint32_t* keys, uint32_t* vals, int n, int key_bits, const SortWs& ws, hipStream_t s,
               bool input_in_alt = false) {
  if (n <= 1 && !input_in_alt) return MI_OK;
  int passes, rb;
  pass_plan(key_bits, &passes, &rb);
  uint32_t *ki = keys, *vi = vals, *ko = ws.keys_alt, *vo = ws.vals_alt;
  if (input_in_alt) { ki = ws.keys_alt; vi = ws.vals_alt; ko = keys; vo = vals; }
  for (int p = 0; p < passes; ++p) {
    const bool small = items_for(n) == 4;
    if (rb == 8) {
      if (small) launch_pass<8, 4>(ki, vi, ko, vo, n, p * rb, ws, s);
      else launch_pass<8, 8>(ki, vi, ko, vo, n, p * rb, ws, s);
    } else {
      if (small) launch_pass<9, 4>(ki, vi, ko, vo, n, p * rb, ws, s);
      else launch_pass<9, 8>(ki, vi, ko, vo, n, p * rb, ws, s);
    }
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

}  // namespace
#include "table_sort.h"
namespace {

template <typename IDX>
__global__ __launch_bounds__(TS_THREADS) void table_sort_kernel(const int64_t* __restrict__ toffs, int T,
                                                               const void* __restrict__ indices, int bag_len,
                                                               int n, int parts, uint32_t sentinel,
                                                               uint32_t* __restrict__ sorted_rows,
                                                               uint32_t* __restrict__ perm, int ids_tm) {
  int t, part, np;
  if (!ts_assign(toffs, T, n, parts, (int) blockIdx.x, t, part, np)) return;
  table_sort_body<IDX>(t, part, np, toffs, T, indices, bag_len, n, sentinel,
                       sorted_rows, perm, ids_tm != 0);
}

// Fixed-width bags, sample-major ids [n_bags / T][T][bag_len] -> int32 ids table-major [T][per_table] (entry e of a
// table: its e-th (sample, position-in-bag) pair).  Ids outside int32 become -1: outside every table.
template <typename IDX>
__global__ __launch_bounds__(256) void ids_table_major_kernel(const void* __restrict__ indices, int T, int bag_len,
                                                             int per_table, int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int per_sample = T * bag_len;
  const int b = i / per_sample, rem = i - b * per_sample;
  const int t = rem / bag_len, j = rem - t * bag_len;
  const int64_t id = mi_ld_id<IDX>(indices, (int64_t) i);
  out[(size_t) t * per_table + (size_t) b * bag_len + j] = (id >= 0 && id <= 0x7FFFFFFFll) ? (int32_t) id : -1;
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
  MI_REQUIRE(n >= 0 && n < ((int64_t) 1 << 31) - 4096, "mi_sort_pairs_u32: n out of range");
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
  MI_REQUIRE(table_offsets && indices && sorted_rows && perm, "mi_embag_bwd_prepare: null pointer");
  const bool want_runs = unique_rows && seg_offsets && counts;
  MI_REQUIRE(want_runs || (!unique_rows && !seg_offsets && !counts),
             "mi_embag_bwd_prepare: unique_rows, seg_offsets and counts must be all set or all NULL");
  MI_REQUIRE(T >= 1 && n_bags >= 0 && n_bags % T == 0, "mi_embag_bwd_prepare: n_bags must be a multiple of T");
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_embag_bwd_prepare: idx_bits must be 32 or 64");
  MI_REQUIRE(n_entries >= 0 && n_entries < ((int64_t) 1 << 31) - 4096, "mi_embag_bwd_prepare: n_entries out of range");
  MI_REQUIRE(offsets || (bag_len >= 1 && n_entries == n_bags * bag_len),
             "mi_embag_bwd_prepare: n_entries must equal n_bags*bag_len for fixed-width bags");
  if (total_rows < 1 || total_rows >= (int64_t) 0xFFFFFFFFll) {
    mi_set_error("mi_embag_bwd_prepare: total_rows %lld does not fit the uint32 row key", (long long) total_rows);
    return MI_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t) stream;
  if (n_entries == 0) {
    if (want_runs) {
      MI_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), s));
      MI_HIP(hipMemsetAsync(seg_offsets, 0, sizeof(int32_t), s));
    }
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
  if (!want_runs && !offsets && n / T <= TS_BIG_MAX) {
    // fixed-width bags, few enough entries per table: every table sorted in LDS by one launch.
    // (Only without run arrays: padding entries then end each table's block, not the array.)
    const int per_table = n / T;
    const int parts = mi_table_sort_parts(T, per_table, 0);
    const unsigned grid = (unsigned) mi_table_sort_blocks(T, per_table, parts);
    uint32_t* pv = reinterpret_cast<uint32_t*>(perm);
    if (per_table > TS_MAX && T > 1) {
      // a table's parts each walk all its ids, twice: a launch before lays them out table by table (int32) so that
      // the walks read whole lines instead of one line per id
      int32_t* tm = reinterpret_cast<int32_t*>(ws.keys_alt);
      if (idx_bits == 32)
        hipLaunchKernelGGL(ids_table_major_kernel<int32_t>, dim3((unsigned) mi_cdiv(n, 256)), dim3(256), 0, s, indices,
                           T, bag_len, per_table, n, tm);
      else
        hipLaunchKernelGGL(ids_table_major_kernel<int64_t>, dim3((unsigned) mi_cdiv(n, 256)), dim3(256), 0, s, indices,
                           T, bag_len, per_table, n, tm);
      hipLaunchKernelGGL(table_sort_kernel<int32_t>, dim3(grid), dim3(TS_THREADS), 0, s, table_offsets, T, tm, bag_len,
                         per_table, parts, sentinel, sorted_rows, pv, 1);
    } else if (idx_bits == 32) {
      hipLaunchKernelGGL(table_sort_kernel<int32_t>, dim3(grid), dim3(TS_THREADS), 0, s, table_offsets, T, indices,
                         bag_len, per_table, parts, sentinel, sorted_rows, pv, T == 1 ? 1 : 0);  // (one table: its ids are one run already)
    } else {
      hipLaunchKernelGGL(table_sort_kernel<int64_t>, dim3(grid), dim3(TS_THREADS), 0, s, table_offsets, T, indices,
                         bag_len, per_table, parts, sentinel, sorted_rows, pv, T == 1 ? 1 : 0);  // (one table: its ids are one run already)
    }
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
  const unsigned kgrid = (unsigned) mi_cdiv(n, 256);
  uint32_t* vals = reinterpret_cast<uint32_t*>(perm);
  const int key_bits = bit_width_u64((uint64_t) total_rows);
  int passes, rb;
  pass_plan(key_bits, &passes, &rb);
  const bool odd = (passes & 1) != 0;   // odd pass count: start in the alternate buffers, end in place
  uint32_t* k0 = odd ? ws.keys_alt : sorted_rows;
  uint32_t* v0 = odd ? ws.vals_alt : vals;
  if (idx_bits == 32)
    hipLaunchKernelGGL(keygen_kernel<int32_t>, dim3(kgrid), dim3(256), 0, s, table_offsets, T, indices, offsets,
                       bag_len, (int) n_bags, n, sentinel, k0, v0);
  else
    hipLaunchKernelGGL(keygen_kernel<int64_t>, dim3(kgrid), dim3(256), 0, s, table_offsets, T, indices, offsets,
                       bag_len, (int) n_bags, n, sentinel, k0, v0);
  MI_LAUNCH_CHECK();
  const int rc = radix_sort(sorted_rows, vals, n, key_bits, ws, s, odd);
  if (rc != MI_OK) return rc;
  if (!want_runs) return MI_OK;
  const int nblk = (int) mi_cdiv(n, SEG_TILE);  // <= ws.tiles
  hipLaunchKernelGGL(seg_count_kernel, dim3(nblk), dim3(256), 0, s, sorted_rows, n, sentinel, ws.seg_blocks, nblk);
  hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws.seg_blocks, nblk, counts, seg_offsets);
  hipLaunchKernelGGL(seg_write_kernel, dim3(nblk), dim3(256), 0, s, sorted_rows, n, sentinel, ws.seg_blocks,
                     unique_rows, seg_offsets);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

// ---- the entry point that allocates nothing ------------------------------------------------
// Sort scratch for callers that cannot bring a workspace: a fixed array in the library's own
// device image (every GPU the library runs on holds its own copy until the process ends).
// `noalloc_scratch` is only a host-side stand-in for the symbol: the device address is looked up
// once per device with hipGetSymbolAddress and only that pointer is ever handed to kernels.
constexpr int64_t NOALLOC_MAX_ENTRIES = MI_NOALLOC_MAX_ENTRIES;
constexpr size_t NOALLOC_SCRATCH_BYTES = 18884864;  // sort_ws_bytes(NOALLOC_MAX_ENTRIES), checked at first use
static __device__ __align__(256) unsigned char noalloc_scratch[NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
void* g_noalloc_ptr[NOALLOC_MAX_DEVICES];
std::mutex g_noalloc_mutex;

// Device address of this device's copy of noalloc_scratch (looked up on the first call per device).
int noalloc_scratch_ptr(void** out, hipStream_t stream) {
  int dev = -1;
  MI_HIP(hipGetDevice(&dev));
  MI_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "mi_embag_bwd_prepare_noalloc: device %d outside 0..%d", dev,
             NOALLOC_MAX_DEVICES - 1);
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  if (!g_noalloc_ptr[dev]) {
    // the lookup resolves a symbol of the code object: not an operation a capturing stream can record
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      mi_set_error("mi_embag_bwd_prepare_noalloc: the first call on a device must be made outside stream capture "
                   "(a call with n_entries = 0 resolves the scratch address and launches nothing)");
      return MI_ERR_UNSUPPORTED;
    }
    void* p = nullptr;
    MI_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(noalloc_scratch)));
    MI_REQUIRE(p && ((uintptr_t) p & 255) == 0, "mi_embag_bwd_prepare_noalloc: scratch symbol not resolved");
    g_noalloc_ptr[dev] = p;
  }
  *out = g_noalloc_ptr[dev];
  return MI_OK;
}
}  // namespace

extern "C" int mi_embag_bwd_prepare_noalloc(const int64_t* table_offsets, int T, int64_t total_rows,
                                            const void* indices, int idx_bits, const int64_t* offsets, int bag_len,
                                            int64_t n_bags, int64_t n_entries, uint32_t* sorted_rows, int32_t* perm,
                                            uint32_t* unique_rows, int32_t* seg_offsets, int32_t* counts,
                                            void* stream) {
  static_assert(NOALLOC_SCRATCH_BYTES % 256 == 0, "scratch is carved in 256-byte units");
  if (n_entries < 0 || n_entries > NOALLOC_MAX_ENTRIES || sort_ws_bytes(n_entries) > NOALLOC_SCRATCH_BYTES) {
    mi_set_error("mi_embag_bwd_prepare_noalloc: n_entries %lld beyond the built-in scratch (at most %lld entries)",
                 (long long) n_entries, (long long) NOALLOC_MAX_ENTRIES);
    return MI_ERR_WORKSPACE;
  }
  void* ws = nullptr;
  const int rc = noalloc_scratch_ptr(&ws, (hipStream_t) stream);
  if (rc != MI_OK) return rc;
  if (n_entries == 0 && (!indices || !sorted_rows || !perm)) return MI_OK;  // a warm-up call: the address is resolved
  // every buffer carved from the scratch is fully written before it is read (alternate key / value
  // buffers by key generation or a pass, histograms and block counts by their own kernels), so
  // nothing of the previous call's contents needs clearing
  return mi_embag_bwd_prepare(table_offsets, T, total_rows, indices, idx_bits, offsets, bag_len, n_bags, n_entries,
                              sorted_rows, perm, unique_rows, seg_offsets, counts, ws, NOALLOC_SCRATCH_BYTES, stream);
}
