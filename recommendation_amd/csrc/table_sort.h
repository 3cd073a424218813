// Per-table stable radix sort in LDS, shared by the stand-alone sort launch and by the fused
// forward+sort launch of the FM step.
#pragma once
#include "common.h"

// ---- per-table sort in LDS -------------------------------------------------------------------
// With fixed-width bags every table receives the same number of entries, B*bag_len.  When that
// is at most TS_MAX the whole sort of a table fits the LDS of one CU (two key/value images of
// 64 KB): ONE launch, one workgroup per table, all radix passes on chip, and each table only
// makes the passes its own row count needs (a table of 3 rows: one pass; 10 M rows: three).
// Tables are laid out back to back in row order, so concatenating the per-table results IS the
// global sort by (row, position); padding / invalid entries end each table's block as sentinels.
constexpr int TS_THREADS = 1024;
constexpr int TS_WAVES = TS_THREADS / MI_WAVE;
constexpr int TS_ITEMS = 8;
constexpr int TS_MAX = TS_THREADS * TS_ITEMS;  // 8192 entries per table
constexpr int TS_RADIX = 256;

// Table `t` is sorted by the calling workgroup (all TS_THREADS threads take part).
template <typename IDX>
__device__ __forceinline__ void table_sort_body(int t, const int64_t* __restrict__ toffs, int T,
                                                const void* __restrict__ indices, int bag_len, int n,
                                                uint32_t sentinel, uint32_t* __restrict__ sorted_rows,
                                                uint32_t* __restrict__ perm) {
  __shared__ uint32_t kbuf[2][TS_MAX];
  __shared__ uint32_t vbuf[2][TS_MAX];
  __shared__ uint32_t cnt[TS_WAVES][TS_RADIX];
  __shared__ uint32_t dbase[TS_RADIX];
  __shared__ uint32_t qsum[4][TS_RADIX];
  __shared__ uint32_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t base_row = toffs[t];
  const uint32_t rows = (uint32_t) (toffs[t + 1] - base_row);
  // local key: the id, or `rows` (one past the largest) for a padding / out-of-range entry
#pragma unroll
  for (int k = 0; k < TS_ITEMS; ++k) {
    const int e = k * TS_THREADS + tid;
    if (e < n) {
      const int b = e / bag_len;
      const int p = (b * T + t) * bag_len + (e - b * bag_len);
      const int64_t id = mi_ld_id<IDX>(indices, p);
      kbuf[0][e] = (id >= 0 && id < (int64_t) rows) ? (uint32_t) id : rows;
      vbuf[0][e] = (uint32_t) p;
    }
  }
  int bits = 0;
  for (uint32_t v = rows; v; v >>= 1) ++bits;  // values 0..rows
  const int passes = (bits + 7) / 8;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int cur = 0;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = pass * 8;
    for (int i = tid; i < TS_WAVES * TS_RADIX; i += TS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();  // also covers the initial fill / the previous pass's scatter
    // rank inside the wave's 512-entry slice (entry j of lane l: slice + j*64 + l)
    uint32_t key[TS_ITEMS], loc[TS_ITEMS];
    const int wbase = w * (MI_WAVE * TS_ITEMS);
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      const int e = wbase + j * MI_WAVE + lane;
      const bool valid = e < n;
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
    // digit totals -> per-wave running bases, then an exclusive scan of the 256 totals
    // thread (q, d) = (tid >> 8, tid & 255) scans waves 4q..4q+3 of digit d; the four partial
    // totals of a digit are then chained through LDS
    uint32_t part = 0;
    {
      const int dq = tid & (TS_RADIX - 1), q4 = (tid >> 8) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t c = cnt[q4 + i][dq];
        cnt[q4 + i][dq] = part;
        part += c;
      }
      qsum[tid >> 8][dq] = part;
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
      const int e = wbase + j * MI_WAVE + lane;
      val[j] = e < n ? vbuf[cur][e] : 0u;
    }
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      const int e = wbase + j * MI_WAVE + lane;
      if (e < n) {
        const uint32_t d = (key[j] >> shift) & (TS_RADIX - 1);
        const uint32_t pos = dbase[d] + cnt[w][d] + loc[j];
        kbuf[cur ^ 1][pos] = key[j];
        vbuf[cur ^ 1][pos] = val[j];
      }
    }
    cur ^= 1;
    __syncthreads();  // the scatter is complete (and cnt / dbase free) before the next pass
  }
  const size_t out = (size_t) t * n;
#pragma unroll
  for (int k = 0; k < TS_ITEMS; ++k) {
    const int e = k * TS_THREADS + tid;
    if (e < n) {
      const uint32_t lk = kbuf[cur][e];
      sorted_rows[out + e] = lk < rows ? (uint32_t) (base_row + lk) : sentinel;
      perm[out + e] = vbuf[cur][e];
    }
  }
}

