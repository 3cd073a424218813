// Per-table stable radix sort in LDS, shared by the stand-alone sort launch and by the fused
// forward+sort launch of the FM step.
#pragma once
#include "common.h"

// ---- per-table sort in LDS -------------------------------------------------------------------
// With fixed-width bags every table receives the same number of entries, B*bag_len.  When that
// is at most TS_MAX the whole sort of a table fits the LDS of one CU (two key/value images of
// 64 KB): ONE launch, all radix passes on chip, and each table only makes the passes its own
// row count needs (a table of 3 rows: one pass; 10 M rows: three).  Tables are laid out back to
// back in row order, so concatenating the per-table results IS the global sort by
// (row, position); padding / invalid entries end each table's block as sentinels.
//
// A table may be shared by `parts` workgroups without any communication between them: every one
// reads all the table's ids, histograms their top 8 key bits, cuts the 256 bins into `parts`
// contiguous ranges of about equal population (the same cut in every workgroup, it only depends
// on the data), keeps the entries of ITS range in their original order, sorts those, and writes
// them where they belong: behind the entries of the lower ranges, whose number it knows from the
// histogram.  The ranking work, which dominates, drops by the number of parts for evenly spread
// keys; a single over-full bin cannot be cut, so a heavily skewed table gains less.
constexpr int TS_THREADS = 1024;
constexpr int TS_WAVES = TS_THREADS / MI_WAVE;
constexpr int TS_ITEMS = 8;
constexpr int TS_MAX = TS_THREADS * TS_ITEMS;  // 8192 entries per table
constexpr int TS_RADIX = 256;
constexpr int TS_PARTS_MAX = 8;

// How many workgroups share a table: up to 4 once a table has enough entries to make the split
// worth its extra pass over the ids, as long as all of them (plus `other_blocks` workgroups of the
// same launch) still fit the chip's 256 CUs at one workgroup per CU.
static inline int mi_table_sort_parts(int T, int per_table, int other_blocks) {
  int parts = per_table >= 4096 ? 4 : (per_table >= 2048 ? 2 : 1);
  while (parts > 1 && T * parts + other_blocks > 248) parts >>= 1;
  return parts;
}

// The LDS arrays of a sorting workgroup (152 KB: one workgroup per CU).
struct TsLds {
  uint32_t kbuf[2][TS_MAX];
  uint32_t vbuf[2][TS_MAX];
  uint32_t cnt[TS_WAVES][TS_RADIX];
  uint32_t dbase[TS_RADIX];
  uint32_t qsum[4][TS_RADIX];
  uint32_t wsum[TS_WAVES];
  int cut[2];
};

// Stable LSD radix sort of the m <= TS_MAX (key, value) pairs in L.kbuf[cur] / L.vbuf[cur] by the low
// 8*passes key bits, entirely in LDS; returns the image that holds the result.  All TS_THREADS
// threads call this; the caller's fill needs no barrier of its own (the first pass has one) and
// the result is visible to every thread on return.
__device__ __forceinline__ int ts_lds_sort(TsLds& L, int m, int passes, int cur) {
  auto& kbuf = L.kbuf;
  auto& vbuf = L.vbuf;
  auto& cnt = L.cnt;
  auto& dbase = L.dbase;
  auto& qsum = L.qsum;
  auto& wsum = L.wsum;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // the m entries are spread over all waves: `items` per thread, wave w owns [w*64*items, (w+1)*64*items)
  const int items = (m + TS_THREADS - 1) / TS_THREADS;
  const int wbase = w * MI_WAVE * items;
  for (int pass = 0; pass < passes && m > 0; ++pass) {
    const int shift = pass * 8;
    for (int i = tid; i < TS_WAVES * TS_RADIX; i += TS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();  // also covers the initial fill / the previous pass's scatter
    // rank inside the wave's slice (entry j of lane l: slice + j*64 + l)
    uint32_t key[TS_ITEMS], loc[TS_ITEMS];
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      if (j >= items) break;  // uniform over the workgroup
      const int e = wbase + j * MI_WAVE + lane;
      const bool valid = e < m;
      key[j] = valid ? kbuf[cur][e] : 0xFFFFFFFFu;
      const uint32_t d = (key[j] >> shift) & (TS_RADIX - 1);
      // lanes holding the same digit: AND over the 8 digit bits of "lanes whose bit equals mine".
      // x = 0 / ~0 (bit clear / set), so ~(ballot ^ x) is that set; the two 32-bit halves are kept
      // apart (sign-extending bit extract, xnor, and: 6 vector instructions per bit).
      const unsigned long long vb = __ballot(valid);
      uint32_t mlo = (uint32_t) vb, mhi = (uint32_t) (vb >> 32);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int x = ((int) (d << (31 - b))) >> 31;
        const unsigned long long bl = __ballot(x != 0);
        mlo &= ~((uint32_t) bl ^ (uint32_t) x);
        mhi &= ~((uint32_t) (bl >> 32) ^ (uint32_t) x);
      }
      const uint32_t before = (uint32_t) (__popc(mlo & (uint32_t) lt) + __popc(mhi & (uint32_t) (lt >> 32)));
      const uint32_t old = cnt[w][d];
      if (valid && before == 0) cnt[w][d] = old + (uint32_t) (__popc(mlo) + __popc(mhi));
      loc[j] = old + before;
    }
    __syncthreads();
    // digit totals -> per-wave running bases, then an exclusive scan of the 256 totals
    // thread (q, d) = (tid >> 8, tid & 255) scans waves 4q..4q+3 of digit d; the four partial
    // totals of a digit are then chained through LDS
    uint32_t partial = 0;
    {
      const int dq = tid & (TS_RADIX - 1), q4 = (tid >> 8) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t c = cnt[q4 + i][dq];
        cnt[q4 + i][dq] = partial;
        partial += c;
      }
      qsum[tid >> 8][dq] = partial;
    }
    __syncthreads();
    uint32_t total = 0;
    {
      const int dq = tid & (TS_RADIX - 1), q = tid >> 8;
      uint32_t before_q = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t c = qsum[i][dq];
        if (i < q) before_q += c;
        total += c;
      }
      const int q4 = q * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) cnt[q4 + i][dq] += before_q;
      if (tid >= TS_RADIX) total = 0;
    }
    uint32_t inc = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t x = __shfl_up(inc, o, 64);
      if (lane >= o) inc += x;
    }
    if (tid < TS_RADIX && lane == 63) wsum[w] = inc;
    __syncthreads();
    if (tid < TS_RADIX) {
      uint32_t wb = 0;
      for (int i = 0; i < w; ++i) wb += wsum[i];
      dbase[tid] = wb + inc - total;
    }
    __syncthreads();
    uint32_t val[TS_ITEMS];
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      if (j >= items) break;
      const int e = wbase + j * MI_WAVE + lane;
      val[j] = e < m ? vbuf[cur][e] : 0u;
    }
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      if (j >= items) break;
      const int e = wbase + j * MI_WAVE + lane;
      if (e < m) {
        const uint32_t d = (key[j] >> shift) & (TS_RADIX - 1);
        const uint32_t pos = dbase[d] + cnt[w][d] + loc[j];
        kbuf[cur ^ 1][pos] = key[j];
        vbuf[cur ^ 1][pos] = val[j];
      }
    }
    cur ^= 1;
    __syncthreads();  // the scatter is complete (and cnt / dbase free) before the next pass
  }
  return cur;
}

// The n <= TS_MAX (key, value) pairs of a table lie in image `parts > 1 ? 1 : 0` of L, keys below
// 2^bits.  Sorts part `part` of `parts` of them (see above) into the returned image, entries
// [0, m); out_base = how many entries of the table sort before this part's.  All threads call this.
__device__ __forceinline__ int ts_sort_part(TsLds& L, int n, int part, int parts, int bits, int& m_out,
                                            int& out_base_out) {
  auto& kbuf = L.kbuf;
  auto& vbuf = L.vbuf;
  auto& dbase = L.dbase;
  auto& qsum = L.qsum;
  auto& wsum = L.wsum;
  auto& cut = L.cut;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int passes = (bits + 7) / 8;
  int m = n, out_base = 0;  // entries this workgroup sorts, and where its output starts in the table's block
  if (parts > 1) {
    const int top = bits > 8 ? bits - 8 : 0;  // bin = key >> top, 0..255
    if (tid < TS_RADIX) dbase[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TS_ITEMS; ++k) {
      const int e = k * TS_THREADS + tid;
      if (e < n) atomicAdd(&dbase[kbuf[1][e] >> top], 1u);  // integer counts: the order of the adds does not matter
    }
    __syncthreads();
    // exclusive prefix of the 256 bin counts -> qsum[0][bin]
    uint32_t c = tid < TS_RADIX ? dbase[tid] : 0u, inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t x = __shfl_up(inc, o, 64);
      if (lane >= o) inc += x;
    }
    if (tid < TS_RADIX && lane == 63) wsum[w] = inc;
    __syncthreads();
    if (tid < TS_RADIX) {
      uint32_t wb = 0;
      for (int i = 0; i < w; ++i) wb += wsum[i];
      qsum[0][tid] = wb + inc - c;
    }
    __syncthreads();
    // range q starts at the last bin whose prefix does not exceed q*n/parts (bin 0 for q = 0)
    if (tid < TS_RADIX) {
      const uint32_t mine = qsum[0][tid], next = tid + 1 < TS_RADIX ? qsum[0][tid + 1] : 0xFFFFFFFFu;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int q = part + s;
        if (q > 0 && q < parts) {
          const uint32_t target = (uint32_t) (((int64_t) q * n) / parts);
          if (mine <= target && next > target) cut[s] = tid;
        }
      }
    }
    if (tid == 0) {
      if (part == 0) cut[0] = 0;
      if (part == parts - 1) cut[1] = TS_RADIX;
    }
    __syncthreads();
    const int b_lo = cut[0], b_hi = cut[1];
    out_base = (int) qsum[0][b_lo];
    m = (b_hi == TS_RADIX ? n : (int) qsum[0][b_hi]) - out_base;
    // stable selection: entry e of the table keeps its place among the selected ones
    unsigned long long bal[TS_ITEMS];
    uint32_t key[TS_ITEMS];
    const int wbase = w * (MI_WAVE * TS_ITEMS);
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      const int e = wbase + j * MI_WAVE + lane;
      key[j] = e < n ? kbuf[1][e] : 0u;
      const int bin = (int) (key[j] >> top);
      bal[j] = __ballot(e < n && bin >= b_lo && bin < b_hi);
      if (lane == 0) qsum[1][w * TS_ITEMS + j] = (uint32_t) __popcll(bal[j]);
    }
    __syncthreads();
    if (tid < TS_WAVES * TS_ITEMS) {  // 128 group counts -> exclusive prefix (two waves)
      const uint32_t g = qsum[1][tid];
      uint32_t gi = g;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(gi, o, 64);
        if (lane >= o) gi += x;
      }
      if (lane == 63) wsum[w] = gi;
      qsum[2][tid] = gi - g;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      const int e = wbase + j * MI_WAVE + lane;
      if ((bal[j] >> lane) & 1ull) {
        const int g = w * TS_ITEMS + j;
        const uint32_t dest = qsum[2][g] + (g >= 64 ? wsum[0] : 0u) + (uint32_t) __popcll(bal[j] & lt);
        kbuf[0][dest] = key[j];
        vbuf[0][dest] = vbuf[1][e];
      }
    }
    __syncthreads();
  }
  const int cur = ts_lds_sort(L, m, passes, 0);
  if (passes == 0) __syncthreads();  // keys of zero bits: the fill must still be visible
  m_out = m;
  out_base_out = out_base;
  return cur;
}

// Part `part` of `parts` of table `t` is sorted by the calling workgroup (all TS_THREADS threads
// take part).
template <typename IDX>
__device__ __forceinline__ void table_sort_body(int t, int part, int parts, const int64_t* __restrict__ toffs, int T,
                                                const void* __restrict__ indices, int bag_len, int n,
                                                uint32_t sentinel, uint32_t* __restrict__ sorted_rows,
                                                uint32_t* __restrict__ perm, bool ids_tm = false) {
  __shared__ TsLds L;
  auto& kbuf = L.kbuf;
  auto& vbuf = L.vbuf;
  const int tid = threadIdx.x;
  const int64_t base_row = toffs[t];
  const uint32_t rows = (uint32_t) (toffs[t + 1] - base_row);
  int bits = 0;
  for (uint32_t v = rows; v; v >>= 1) ++bits;  // values 0..rows
  // local key: the id, or `rows` (one past the largest) for a padding / out-of-range entry;
  // image 0 when the workgroup sorts the whole table, image 1 (scratch) when it first selects
  const int first = parts > 1 ? 1 : 0;
  // all of a thread's ids are requested before the first is used: the loads carry no branch around them
  // (entries past n re-read entry 0; where control flow joins behind a load the compiler waits for every
  // outstanding load, which made these eight loads eight dependent round trips)
  int64_t idv[TS_ITEMS];
  int pv[TS_ITEMS];
#pragma unroll
  for (int k = 0; k < TS_ITEMS; ++k) {
    const int e = k * TS_THREADS + tid;
    const int es = e < n ? e : 0;
    const int b = es / bag_len;
    pv[k] = (b * T + t) * bag_len + (es - b * bag_len);
    // table-major ids (bag_len 1): the table's ids are one contiguous run, a full-line load per wave
    // instead of one line per id
    idv[k] = mi_ld_id<IDX>(indices, ids_tm ? (int64_t) t * n + es : (int64_t) pv[k]);
  }
#pragma unroll
  for (int k = 0; k < TS_ITEMS; ++k) {
    const int e = k * TS_THREADS + tid;
    if (e < n) {
      kbuf[first][e] = (idv[k] >= 0 && idv[k] < (int64_t) rows) ? (uint32_t) idv[k] : rows;
      vbuf[first][e] = (uint32_t) pv[k];
    }
  }
  int m, out_base;  // entries this workgroup sorts, and where its output starts in the table's block
  const int cur = ts_sort_part(L, n, part, parts, bits, m, out_base);
  const size_t out = (size_t) t * n + out_base;
#pragma unroll
  for (int k = 0; k < TS_ITEMS; ++k) {
    const int e = k * TS_THREADS + tid;
    if (e < m) {
      const uint32_t lk = kbuf[cur][e];
      sorted_rows[out + e] = lk < rows ? (uint32_t) (base_row + lk) : sentinel;
      perm[out + e] = vbuf[cur][e];
    }
  }
}
