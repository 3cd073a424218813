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
// A table may be shared by `parts` workgroups without any communication between them.  Every one
// reads all the table's ids and 64 evenly spaced samples of them, ranks the samples (the same
// ranking in every workgroup, it only depends on the data) and takes the samples of rank
// 64*q/parts as cuts of the order (key, position).  Positions are distinct, so whatever the keys
// are - even all equal - the cuts divide the table into about equal parts.  The workgroup keeps
// the entries between ITS two cuts in their original order, sorts those by key, and writes them
// behind the entries below its lower cut, which it counted while selecting.  The ranking work,
// which dominates, drops by the number of parts.
#ifdef TS_TRACE
static __device__ unsigned long long ts_trace_buf[1024 * 16];
#define TS_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) ts_trace_buf[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define TS_MARK(i) do {} while (0)
#endif
constexpr int TS_THREADS = 1024;
constexpr int TS_WAVES = TS_THREADS / MI_WAVE;
constexpr int TS_ITEMS = 8;
constexpr int TS_MAX = TS_THREADS * TS_ITEMS;  // 8192 entries per table
constexpr int TS_RADIX = 256;
constexpr int TS_PARTS_MAX = 8;
// Tables of more than TS_MAX entries are shared by ceil(n / TS_BIG_PART) workgroups, each of which walks all the
// table's ids twice (count, then fill) and sorts its share of about TS_BIG_PART entries: see table_sort_big_body.
constexpr int TS_BIG_MAX = 128 * 1024;  // entries per table (positions take 17 bits of the 64-bit order word)
constexpr int TS_BIG_PART = 4096;
constexpr int TS_XCDS = 8;

// How many workgroups are launched per table: up to TS_PMAX once a table has enough entries to make
// sharing worth its extra pass over the ids.  Only tables of more than 65535 rows use them all (see
// table_sort_body), the others' spare workgroups leave at once; about half are taken to stay, and
// those (plus `other_blocks` workgroups of the same launch) should fit the chip's 256 CUs at one
// workgroup per CU.
#ifndef TS_PMAX
#define TS_PMAX 8
#endif
static inline int mi_table_sort_parts(int T, int per_table, int other_blocks) {
  if (per_table > TS_MAX) return (per_table + TS_BIG_PART - 1) / TS_BIG_PART;  // at most 32
  int parts = per_table >= 4096 ? TS_PMAX : (per_table >= 2048 ? 2 : 1);
  while (parts > 1 && T * parts / 2 + other_blocks > 248) parts >>= 1;
  return parts;
}

// Sort workgroups a launch needs (see ts_assign for how they are dealt to tables).
static inline int mi_table_sort_blocks(int T, int per_table, int parts) {
  return per_table > TS_MAX ? (T + TS_XCDS - 1) / TS_XCDS * TS_XCDS * parts : T * parts;
}

// The LDS arrays of a sorting workgroup (152 KB: one workgroup per CU).
struct alignas(16) TsLds {
  uint32_t kbuf[2][TS_MAX];
  uint32_t vbuf[2][TS_MAX];
  uint32_t cnt[TS_WAVES][TS_RADIX];
  uint32_t dbase[TS_RADIX];
  uint32_t qsum[4][TS_RADIX];
  uint32_t wsum[TS_WAVES];
  int cut[2];
  unsigned long long cut64[2];
  uint32_t wsel[TS_WAVES][2];
};

// Inclusive prefix sum over the 64 lanes of a wave, all lanes active, in six data-parallel-primitive
// adds (shifts inside rows of 16 lanes, then the last lane of a row broadcast into the following
// rows) instead of six round trips through the LDS permute path.
__device__ __forceinline__ uint32_t ts_wave_scan(uint32_t x) {
  int v = (int) x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return (uint32_t) v;
}

// Stable LSD radix sort of the m <= TS_MAX (key, value) pairs in L.kbuf[cur] / L.vbuf[cur] by the low
// `bits` key bits (8 per pass), entirely in LDS; returns the image that holds the result.  All
// TS_THREADS threads call this; the caller's fill needs no barrier of its own and the result is
// visible to every thread on return.
//
// A pass has three phases and three barriers.  (1) Every wave ranks its own slice: lanes with the
// same digit find each other by ballots, the counts go to the wave's own row of cnt, which the wave
// itself cleared (no barrier between the clear and the ranking).  (2) ONE wave turns the 16 x 256
// counts into scatter bases: lane l owns digits 4l..4l+3, reads them for every wave with 128-bit
// loads, and needs nothing but a wave scan to know every base.  (3) Every wave scatters its
// entries.  A chain of small cooperative scans here used to cost three more barriers a pass, and
// at two entries per thread the barriers and the LDS round trips between them were the whole cost.
__device__ __forceinline__ int ts_lds_sort_bits(TsLds& L, int m, int bits, int cur) {
  auto& kbuf = L.kbuf;
  auto& vbuf = L.vbuf;
  auto& cnt = L.cnt;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // the m entries are spread over the waves in slices of whole 64-entry groups: every wave takes m / TS_THREADS
  // groups and the last waves one more, so no SIMD ranks a group more than it must
  // (the LAST waves: wave 0 has the scan below to itself)
  const int items_lo = m / TS_THREADS, extra_waves = (m - items_lo * TS_THREADS + MI_WAVE - 1) / MI_WAVE;
  const int first_extra = TS_WAVES - extra_waves;
  const int items = items_lo + (w >= first_extra ? 1 : 0);  // uniform over the wave
  const int wbase = MI_WAVE * (w * items_lo + (w >= first_extra ? w - first_extra : 0));
  __syncthreads();  // the caller's fill
  for (int shift = 0; shift < bits && m > 0; shift += 8) {
    const int nb = bits - shift < 8 ? bits - shift : 8;  // digit bits that can be set in this pass
    *reinterpret_cast<uint4*>(&cnt[w][4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
    // rank inside the wave's slice (entry j of lane l: slice + j*64 + l)
    uint32_t key[TS_ITEMS], val[TS_ITEMS], loc[TS_ITEMS];
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      if (j >= items) break;
      const int e = wbase + j * MI_WAVE + lane;
      const bool valid = e < m;
      key[j] = valid ? kbuf[cur][e] : 0xFFFFFFFFu;
      val[j] = valid ? vbuf[cur][e] : 0u;
    }
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      if (j >= items) break;
      const bool valid = wbase + j * MI_WAVE + lane < m;
      const uint32_t d = (key[j] >> shift) & (TS_RADIX - 1);
      // lanes holding the same digit: AND over the digit bits of "lanes whose bit equals mine".
      // x = 0 / ~0 (bit clear / set), so ~(ballot ^ x) is that set; the two 32-bit halves are kept
      // apart (sign-extending bit extract, xnor, and: 6 vector instructions per bit).
      const unsigned long long vb = __ballot(valid);
      uint32_t mlo = (uint32_t) vb, mhi = (uint32_t) (vb >> 32);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b >= nb) break;  // uniform
        int x;  // (written out: the compiler's own form is a shift, a compare and a second shift)
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(x) : "v"(d), "n"(b));
        const unsigned long long bl = __builtin_amdgcn_ballot_w64(x != 0);
        mlo &= ~((uint32_t) bl ^ (uint32_t) x);
        mhi &= ~((uint32_t) (bl >> 32) ^ (uint32_t) x);
      }
      const uint32_t before = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));  // same-digit lanes below mine
      const uint32_t old = cnt[w][d];
      if (valid && before == 0) cnt[w][d] = old + (uint32_t) (__popc(mlo) + __popc(mhi));
      loc[j] = old + before;
    }
    __syncthreads();
    if (w == 0) {
      // cnt[ww][d] <- entries that precede wave ww's first entry of digit d: all entries of lower
      // digits, and those of digit d in lower waves
      uint4 c[TS_WAVES];
#pragma unroll
      for (int ww = 0; ww < TS_WAVES; ++ww) c[ww] = *reinterpret_cast<const uint4*>(&cnt[ww][4 * lane]);
      uint4 run = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int ww = 0; ww < TS_WAVES; ++ww) {
        const uint4 x = c[ww];
        c[ww] = run;
        run.x += x.x; run.y += x.y; run.z += x.z; run.w += x.w;
      }
      const uint32_t mine = run.x + run.y + run.z + run.w;
      const uint32_t inc = ts_wave_scan(mine);
      const uint32_t b0 = inc - mine, b1 = b0 + run.x, b2 = b1 + run.y, b3 = b2 + run.z;
#pragma unroll
      for (int ww = 0; ww < TS_WAVES; ++ww)
        *reinterpret_cast<uint4*>(&cnt[ww][4 * lane]) = make_uint4(c[ww].x + b0, c[ww].y + b1, c[ww].z + b2, c[ww].w + b3);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      if (j >= items) break;
      const int e = wbase + j * MI_WAVE + lane;
      if (e < m) {
        const uint32_t d = (key[j] >> shift) & (TS_RADIX - 1);
        const uint32_t pos = cnt[w][d] + loc[j];
        kbuf[cur ^ 1][pos] = key[j];
        vbuf[cur ^ 1][pos] = val[j];
      }
    }
    cur ^= 1;
    __syncthreads();  // the scatter is complete (and cnt free) before the next pass
    TS_MARK(4 + (shift >> 3));
  }
  return cur;
}

__device__ __forceinline__ int ts_lds_sort(TsLds& L, int m, int passes, int cur) {
  return ts_lds_sort_bits(L, m, passes * 8, cur);
}

// Which table, and which part of it, sort slot `slot` of a launch with parts_max slots per table works
// on.  A table whose keys need fewer passes is shared by fewer workgroups: the passes are what
// sharing shortens, reading and selecting the ids costs the same in every part.  The slots in use
// come first, so the workgroups of the unused ones (for which this returns false) are the last of
// the launch and leave at once.  Every wave works it out for itself, lane l looking at table l: one
// round trip for the offsets and a wave scan, instead of a loop of T dependent scalar loads.
__device__ __forceinline__ bool ts_assign(const int64_t* __restrict__ toffs, int T, int n, int parts_max, int slot,
                                          int& t_out, int& part_out, int& parts_out) {
  if (n > TS_MAX) {
    // Every part of a large table walks all the table's ids: the parts of one table go to workgroups that share an
    // XCD (workgroups are dealt to the XCDs round robin, so slots of equal index mod 8 sit on one), whose L2 then
    // serves all but the first walk.  Table t belongs to XCD t mod 8; the launch has ceil(T / 8) * 8 * parts slots.
#ifdef TS_BIG_MAP_LINEAR
    t_out = slot / parts_max;
    part_out = slot % parts_max;
#else
    const int r = slot % TS_XCDS, idx = slot / TS_XCDS;
    t_out = (idx / parts_max) * TS_XCDS + r;
    part_out = idx % parts_max;
#endif
    parts_out = parts_max;
    return t_out < T;
  }
  const int lane = threadIdx.x & (MI_WAVE - 1);
  int base = 0;
  for (int t0 = 0; t0 < T; t0 += MI_WAVE) {  // lane l looks at table t0 + l
    const int tt = t0 + lane, tc = tt < T ? tt : T - 1;
    const uint32_t rows = (uint32_t) (toffs[tc + 1] - toffs[tc]);
    int want = (n < MI_WAVE || parts_max == 1) ? 1
               : (rows > 65535u ? parts_max : (rows > 255u ? (parts_max + 1) / 2 : (parts_max + 3) / 4));
    if (tt >= T) want = 0;
    const int inc = (int) ts_wave_scan((uint32_t) want);
    const int total = __builtin_amdgcn_readlane(inc, MI_WAVE - 1);
    if (slot < base + total) {
      const int l = __popcll(__ballot(base + inc <= slot));  // tables that end at or before the slot
      t_out = t0 + l;
      parts_out = __builtin_amdgcn_readlane(want, l);
      part_out = slot - base - __builtin_amdgcn_readlane(inc - want, l);
      return true;
    }
    base += total;
  }
  return false;
}

// Part `part` of `parts` of a table of TS_MAX < n <= TS_BIG_MAX entries whose ids lie TABLE-MAJOR (the table's n
// ids one contiguous run: entry e at indices[t * n + e]).  The scheme of table_sort_body with three changes.
// (1) 1024 samples, one per thread, at evenly spaced positions each shifted by a fixed pseudo-random amount (ids
// that repeat with the period of the spacing would otherwise all show the same key); they are ranked by sorting
// them with the LDS sort itself.  (2) The ids do not fit the registers, so they are walked twice - once to count
// the entries between the part's two cuts and those below, once to move the selected ones into LDS in table
// order; wave w walks the w-th sixteenth of the table.  (3) Whatever the data, a part holds more than TS_MAX
// entries only if the samples misjudge it badly; it then halves its range of the order (key, position) until the
// count fits, sorts that, and goes on with the rest of its range - slower, never wrong.
//
// A walk of 65536 ids takes one workgroup about 18 us whatever else runs on the chip (a single CU streams some
// 30 GB/s this way), which is most of the part's time; an entry costs as little else as possible: one load at a constant offset from the lane's pointer, one range check, and two compares
// against thresholds that are the same for the whole wave - the order (key, position) against a cut (K, P) is
// key >= K + 1 for the 64 entries of a round that all lie before P and key >= K for a round that lies behind it;
// only the one round that contains P compares per lane.
template <typename IDX>
__device__ __forceinline__ void table_sort_big_body(TsLds& L, int t, int part, int parts,
                                                    const int64_t* __restrict__ toffs, int T,
                                                    const void* __restrict__ indices, int bag_len, int n,
                                                    uint32_t sentinel, uint32_t* __restrict__ sorted_rows,
                                                    uint32_t* __restrict__ perm) {
  auto& kbuf = L.kbuf;
  auto& vbuf = L.vbuf;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  TS_MARK(0);
  const int64_t base_row = toffs[t];
  const uint32_t rows = (uint32_t) (toffs[t + 1] - base_row);
  const IDX* __restrict__ tab_ids = reinterpret_cast<const IDX*>(indices) + (int64_t) t * n;
  auto key_of_id = [&](IDX id) {  // an id outside [0, rows) sorts behind every row
    if constexpr (sizeof(IDX) == 8) return (unsigned long long) id < (unsigned long long) rows ? (uint32_t) id : rows;
    else return (uint32_t) id < rows ? (uint32_t) id : rows;
  };
  const int wspan = (n + TS_WAVES * MI_WAVE - 1) / (TS_WAVES * MI_WAVE) * MI_WAVE;  // entries of a wave, whole 64s
  const int w_lo = w * wspan, w_hi = w_lo + wspan < n ? w_lo + wspan : n;
  const int stride = n / TS_THREADS;  // >= 8
  auto sample_pos = [&](int i) {
    uint32_t h = (uint32_t) i * 2654435761u;
    h ^= h >> 15;
    return i * stride + (int) (((h >> 8) * (uint32_t) stride) >> 24);  // stride <= 128: the product fits 32 bits
  };
  int all_bits = 0;
  for (uint32_t v = rows; v; v >>= 1) ++all_bits;  // keys 0..rows
  kbuf[0][tid] = key_of_id(tab_ids[sample_pos(tid)]);
  vbuf[0][tid] = (uint32_t) tid;
  {
    // ascending sample index is ascending position: the stable sort by key is the order (key, position)
    const int sc = ts_lds_sort_bits(L, TS_THREADS, all_bits, 0);
    const unsigned long long top = (unsigned long long) (rows + 1u) << 17;  // above every entry's order word
    if (tid < 2) {
      const int q = part + tid;
      unsigned long long c = tid == 0 ? 0ull : top;
      if (q > 0 && q < parts) {
        const int at = q * TS_THREADS / parts;
        c = ((unsigned long long) kbuf[sc][at] << 17) | (unsigned long long) sample_pos((int) vbuf[sc][at]);
      }
      L.cut64[tid] = c;
    }
  }
  __syncthreads();
  TS_MARK(1);
  const unsigned long long hi = L.cut64[1];
  unsigned long long cur_lo = L.cut64[0];
#ifndef TS_BIG_WALK
#define TS_BIG_WALK 16
#endif
  constexpr int WALK = TS_BIG_WALK;  // entries per lane and round
  // a lane's slot beyond the wave's share: a key above every threshold (rows + 1 <= 2^32 - 1 is the highest), so
  // it is neither selected nor counted as below
  constexpr uint32_t TS_NO_ENTRY = 0xFFFFFFFFu;
  // A round is WALK * 64 consecutive entries from r0, read as WALK / PER groups of 64 * PER entries: in a group
  // lane l holds the PER consecutive entries from r0 + (group * 64 + l) * PER - one 16-byte load where the table's
  // ids are aligned to that and the round lies inside the wave's share (four times fewer memory instructions than
  // one id per load; the walk is bound by those), single loads otherwise.  (Slots past the share re-read the
  // share's first entry: no branch between the loads.)
  constexpr int PER = 16 / (int) sizeof(IDX), GROUPS = WALK / PER;
  const bool wide_ok = ((uintptr_t) tab_ids & 15) == 0;  // (w_lo and the rounds' starts are multiples of 64 entries)
  auto load_round = [&](int r0, uint32_t* key) {
    if (wide_ok && r0 + WALK * MI_WAVE <= w_hi) {  // uniform
      const uint4* __restrict__ p = reinterpret_cast<const uint4*>(tab_ids + r0) + lane;
      uint4 raw[GROUPS];
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) raw[g] = p[g * MI_WAVE];
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) {
        if constexpr (sizeof(IDX) == 4) {
          key[g * PER + 0] = key_of_id((IDX) raw[g].x);
          key[g * PER + 1] = key_of_id((IDX) raw[g].y);
          key[g * PER + 2] = key_of_id((IDX) raw[g].z);
          key[g * PER + 3] = key_of_id((IDX) raw[g].w);
        } else {
          key[g * PER + 0] = key_of_id((IDX) (((unsigned long long) raw[g].y << 32) | raw[g].x));
          key[g * PER + 1] = key_of_id((IDX) (((unsigned long long) raw[g].w << 32) | raw[g].z));
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < WALK; ++k) {
        const int e = r0 + ((k / PER) * MI_WAVE + lane) * PER + (k % PER);
        const uint32_t kk = key_of_id(tab_ids[e < w_hi ? e : w_lo]);
        key[k] = e < w_hi ? kk : TS_NO_ENTRY;
      }
    }
  };
  // The first walk of a part does both jobs at once where it can: beside counting it already moves the entries it
  // selects, in entry order, into the wave's own sixteenth of the SECOND key / value image (which lies idle until the
  // radix passes); when the counts of all waves are known a wave copies its entries from there to their place in the
  // first image, and the second walk over the table's ids (a quarter of a part's time) is not made.  A wave that
  // selects more than its sixteenth holds (ids sorted or clustered by position) says so, and the part falls back to
  // the second walk; so does a part that had to be narrowed.
  constexpr uint32_t STAGE_CAP = TS_MAX / TS_WAVES;
  bool first_turn = true;
  for (;;) {  // uniform: one turn unless the part had to be narrowed
    unsigned long long cur_hi = hi;
    uint32_t wave_first, mine = 0;
    int m, below;
    bool staged = false;
    for (;;) {  // uniform
      uint32_t nsel = 0, nbelow = 0;
      const uint32_t klo = (uint32_t) (cur_lo >> 17), khi = (uint32_t) (cur_hi >> 17);
      const int plo = (int) (cur_lo & 0x1FFFFull), phi = (int) (cur_hi & 0x1FFFFull);
      bool ovf = false;
      if (first_turn) {  // uniform
        uint32_t* const ks = &kbuf[1][w * STAGE_CAP];
        uint32_t* const vs = &vbuf[1][w * STAGE_CAP];
        uint32_t srun = 0;
        for (int r0 = w_lo; r0 < w_hi; r0 += WALK * MI_WAVE) {
          uint32_t key[WALK];
          load_round(r0, key);
#pragma unroll
          for (int g = 0; g < GROUPS; ++g) {
            const int g0 = r0 + g * MI_WAVE * PER, g1 = g0 + MI_WAVE * PER;
            const int e_first = g0 + lane * PER;
            const bool plain = (g0 >= plo || g1 <= plo) && (g0 >= phi || g1 <= phi);  // uniform: no cut inside the group
            bool sel[PER];
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
              const uint32_t kk = key[g * PER + j];
              const uint32_t tlo = klo + ((plain ? g0 : e_first + j) < plo ? 1u : 0u);
              const uint32_t thi = khi + ((plain ? g0 : e_first + j) < phi ? 1u : 0u);
              sel[j] = kk - tlo < thi - tlo;  // tlo <= kk < thi (tlo <= thi; never for TS_NO_ENTRY)
              c += sel[j] ? 1u : 0u;
              nbelow += kk < tlo ? 1u : 0u;
            }
            if (__builtin_amdgcn_ballot_w64(c != 0)) {  // uniform
              const uint32_t incl = ts_wave_scan(c);
              const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) incl, MI_WAVE - 1);
              if (srun + tot <= STAGE_CAP) {  // uniform
                uint32_t d = srun + incl - c;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                  if (sel[j]) {
                    ks[d] = key[g * PER + j] - klo;
                    vs[d] = (uint32_t) (e_first + j);
                    ++d;
                  }
                }
              } else {
                ovf = true;
              }
              srun += tot;
            }
          }
        }
        nsel = srun;
      } else {
      for (int r0 = w_lo; r0 < w_hi; r0 += WALK * MI_WAVE) {
        uint32_t key[WALK];
        load_round(r0, key);
#pragma unroll
        for (int k = 0; k < WALK; ++k) {
          const int g0 = r0 + (k / PER) * MI_WAVE * PER, g1 = g0 + MI_WAVE * PER;  // the group's entries (uniform)
          // (per-lane counters, added up once behind the walk: masks counted on the scalar unit made every entry
          // wait for a compare to cross over to it)
          uint32_t tlo, thi;
          if ((g0 >= plo || g1 <= plo) && (g0 >= phi || g1 <= phi)) {  // uniform: no cut inside the group
            tlo = klo + (g0 < plo ? 1u : 0u);
            thi = khi + (g0 < phi ? 1u : 0u);
          } else {
            const int e = g0 + lane * PER + (k % PER);
            tlo = klo + (e < plo ? 1u : 0u);
            thi = khi + (e < phi ? 1u : 0u);
          }
          // tlo <= thi, so tlo <= key < thi is one unsigned compare (and never holds for TS_NO_ENTRY)
          nsel += (key[k] - tlo < thi - tlo) ? 1u : 0u;
          nbelow += key[k] < tlo ? 1u : 0u;
        }
      }
      nsel = __builtin_amdgcn_readlane((int) ts_wave_scan(nsel), MI_WAVE - 1);
      }
      nbelow = __builtin_amdgcn_readlane((int) ts_wave_scan(nbelow), MI_WAVE - 1);
      __syncthreads();  // (the previous turn's readers of wsel and of the key images are done)
      if (lane == 0) {
        L.wsel[w][0] = nsel | (ovf ? 0x80000000u : 0u);  // (a wave's count is at most n / 16 <= 8192)
        L.wsel[w][1] = nbelow;
      }
      __syncthreads();
      uint32_t a_sel = lane < TS_WAVES ? L.wsel[lane][0] : 0u;
      const uint32_t a_below = lane < TS_WAVES ? L.wsel[lane][1] : 0u;
      const bool any_ovf = __builtin_amdgcn_ballot_w64((a_sel >> 31) != 0u) != 0ull;
      a_sel &= 0x7FFFFFFFu;
      const uint32_t i_sel = ts_wave_scan(a_sel), i_below = ts_wave_scan(a_below);
      wave_first = (uint32_t) __builtin_amdgcn_readlane((int) (i_sel - a_sel), w);
      mine = (uint32_t) __builtin_amdgcn_readlane((int) a_sel, w);
      m = __builtin_amdgcn_readlane((int) i_sel, MI_WAVE - 1);
      below = __builtin_amdgcn_readlane((int) i_below, MI_WAVE - 1);
      TS_MARK(2);
      const bool was_first = first_turn;
      first_turn = false;
      if (m <= TS_MAX) {
        staged = was_first && !any_ovf;
        break;
      }
      cur_hi = cur_lo + (cur_hi - cur_lo) / 2;  // order words are distinct: a range of one word holds one entry
    }
    const uint32_t key_lo = (uint32_t) (cur_lo >> 17);
    const uint32_t key_hi = (uint32_t) (cur_hi >> 17);
    const uint32_t span = (key_hi < rows ? key_hi : rows) - key_lo;
    int bits = 0;
    for (uint32_t v = span; v; v >>= 1) ++bits;
    uint32_t run = wave_first;
    const int plo = (int) (cur_lo & 0x1FFFFull), phi = (int) (cur_hi & 0x1FFFFull);
    if (staged) {  // uniform: the first walk left this wave's entries in its sixteenth of the second image
      for (uint32_t i = (uint32_t) lane; i < mine; i += MI_WAVE) {
        kbuf[0][wave_first + i] = kbuf[1][w * STAGE_CAP + i];
        vbuf[0][wave_first + i] = vbuf[1][w * STAGE_CAP + i];
      }
    }
    for (int r0 = w_lo; r0 < (staged ? w_lo : w_hi); r0 += WALK * MI_WAVE) {
      uint32_t key[WALK];
      load_round(r0, key);
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) {
        // the group's selected entries go to LDS in entry order: lanes in order, a lane's PER entries in order
        const int g0 = r0 + g * MI_WAVE * PER, g1 = g0 + MI_WAVE * PER;
        const int e_first = g0 + lane * PER;
        const bool plain = (g0 >= plo || g1 <= plo) && (g0 >= phi || g1 <= phi);  // uniform: no cut inside the group
        bool sel[PER];
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const uint32_t kk = key[g * PER + j];
          const uint32_t tlo = key_lo + ((plain ? g0 : e_first + j) < plo ? 1u : 0u);
          const uint32_t thi = key_hi + ((plain ? g0 : e_first + j) < phi ? 1u : 0u);
          sel[j] = kk - tlo < thi - tlo;  // tlo <= kk < thi (tlo <= thi)
          c += sel[j] ? 1u : 0u;
        }
        if (__builtin_amdgcn_ballot_w64(c != 0)) {  // uniform
          const uint32_t incl = ts_wave_scan(c);
          uint32_t d = run + incl - c;
#pragma unroll
          for (int j = 0; j < PER; ++j) {
            if (sel[j]) {
              kbuf[0][d] = key[g * PER + j] - key_lo;
              vbuf[0][d] = (uint32_t) (e_first + j);  // the entry's number; its place among the ids is worked out at the end
              ++d;
            }
          }
          run += (uint32_t) __builtin_amdgcn_readlane((int) incl, MI_WAVE - 1);
        }
      }
    }
    TS_MARK(3);
    const int cur = ts_lds_sort_bits(L, m, bits, 0);  // (its first barrier makes the fill visible)
    const size_t out = (size_t) t * n + below;
    const float inv_len = __builtin_amdgcn_rcpf((float) bag_len);
#pragma unroll
    for (int k = 0; k < TS_ITEMS; ++k) {
      const int i = k * TS_THREADS + tid;
      if (i < m) {
        const uint32_t lk = kbuf[cur][i] + key_lo;
        sorted_rows[out + i] = lk < rows ? (uint32_t) (base_row + lk) : sentinel;
        // where entry e's id lies in a SAMPLE-major id array (what perm holds whichever layout was read; the
        // division as in table_sort_body, whose bound on the rounding holds for e < 2^17 as well)
        const int e = (int) vbuf[cur][i];
        const int b = (int) (((float) e + 0.5f) * inv_len);
        perm[out + i] = (uint32_t) ((b * T + t) * bag_len + (e - b * bag_len));
      }
    }
    TS_MARK(8);
#ifdef TS_TRACE
    if (threadIdx.x == 0) { ts_trace_buf[blockIdx.x * 16 + 9] = (unsigned long long) bits; ts_trace_buf[blockIdx.x * 16 + 10] = (unsigned long long) m; }
#endif
    if (cur_hi >= hi) break;
    cur_lo = cur_hi;
  }
}

// Part `part` of `parts` of table `t` is sorted by the calling workgroup (all TS_THREADS threads
// take part).
template <typename IDX>
__device__ __forceinline__ void table_sort_body(int t, int part, int parts, const int64_t* __restrict__ toffs, int T,
                                                const void* __restrict__ indices, int bag_len, int n,
                                                uint32_t sentinel, uint32_t* __restrict__ sorted_rows,
                                                uint32_t* __restrict__ perm, bool ids_tm = false) {
  __shared__ TsLds L;
  if (n > TS_MAX) {  // uniform
    table_sort_big_body<IDX>(L, t, part, parts, toffs, T, indices, bag_len, n, sentinel, sorted_rows, perm);  // (table-major ids: the host checks)
    return;
  }
  auto& kbuf = L.kbuf;
  auto& vbuf = L.vbuf;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  TS_MARK(0);
  const int64_t base_row = toffs[t];
  const uint32_t rows = (uint32_t) (toffs[t + 1] - base_row);
  int bits = 0;
  for (uint32_t v = rows; v; v >>= 1) ++bits;  // values 0..rows
  if (parts > 1 && n < MI_WAVE) {  // too few entries to sample: one workgroup sorts them all
    if (part > 0) return;
    parts = 1;
  }
  // entry e of the table is its e-th (sample, position-in-bag) pair; equal rows keep that order.  Wave w holds
  // e = w*512 + k*64 + lane.  Local key: the id, or `rows` (one past the largest) for a padding / out-of-range
  // entry.  All of a thread's ids are requested before the first is used: the loads carry no branch around
  // them (entries past n re-read entry 0; where control flow joins behind a load the compiler waits for every
  // outstanding load, which made these eight loads eight dependent round trips)
  // e / bag_len without the integer division (some 25 instructions, nine times per thread): e and bag_len are
  // at most TS_MAX, so (e + 0.5) / bag_len in single precision is off by far less than the 0.5 / bag_len that
  // separates it from the next integer
  const float inv_len = __builtin_amdgcn_rcpf((float) bag_len);
  auto place = [&](int e) {  // where entry e's id lies in a sample-major id array
    if (bag_len == 1) return e * T + t;
    const int b = (int) (((float) e + 0.5f) * inv_len);
    return (b * T + t) * bag_len + (e - b * bag_len);
  };
  int64_t idv[TS_ITEMS], sid;
  int pv[TS_ITEMS], es[TS_ITEMS];
  const int se = lane * (n >> 6);  // the 64 samples (parts > 1: n >= 64)
#pragma unroll
  for (int k = 0; k < TS_ITEMS; ++k) {
    const int e = (w * TS_ITEMS + k) * MI_WAVE + lane;
    es[k] = e < n ? e : 0;
    pv[k] = place(es[k]);
  }
  if (ids_tm) {
    // table-major ids (bag_len 1): the table's ids are one contiguous run, a full-line load per wave
    // instead of one line per id
    const int64_t first = (int64_t) t * n;
#pragma unroll
    for (int k = 0; k < TS_ITEMS; ++k) idv[k] = mi_ld_id<IDX>(indices, first + es[k]);
    sid = mi_ld_id<IDX>(indices, first + se);
  } else {
#pragma unroll
    for (int k = 0; k < TS_ITEMS; ++k) idv[k] = mi_ld_id<IDX>(indices, (int64_t) pv[k]);
    sid = mi_ld_id<IDX>(indices, (int64_t) place(se));
  }
  uint32_t key[TS_ITEMS];
#pragma unroll
  for (int k = 0; k < TS_ITEMS; ++k) key[k] = (idv[k] >= 0 && idv[k] < (int64_t) rows) ? (uint32_t) idv[k] : rows;
  int m = n, out_base = 0;  // entries this workgroup sorts, and where its output starts in the table's block
  uint32_t key_lo = 0;      // subtracted from the keys it sorts
  TS_MARK(1);
  if (parts == 1) {
#pragma unroll
    for (int k = 0; k < TS_ITEMS; ++k) {
      const int e = (w * TS_ITEMS + k) * MI_WAVE + lane;
      if (e < n) {
        kbuf[0][e] = key[k];
        vbuf[0][e] = (uint32_t) pv[k];
      }
    }
  } else {
    // rank the samples by (key, lane) - the FULL key: the cuts below are the composites (key, position) of the
    // samples of certain ranks, position grows with the lane, so this ranking is exactly the order of the
    // composites and the cuts ascend with the rank.  (A ranking on truncated keys lets two samples whose kept
    // bits agree swap places against the composite order; the cuts are then not monotone, some parts are empty
    // and their neighbours overlap.  Every part still writes each of its entries at that entry's global place -
    // entries below its lower cut + entries selected before it - so the output stays right, but the overlapping
    // parts do the same work twice, with a key span that wraps to 32 bits.)  Every wave holds all 64 samples;
    // wave w counts how many of samples 4w..4w+3 precede each, and after one barrier every wave adds up the
    // sixteen partial ranks for itself.
    const uint32_t skey = (sid >= 0 && sid < (int64_t) rows) ? (uint32_t) sid : rows;
    {
      uint32_t pr = 0;
#pragma unroll
      for (int j = 0; j < MI_WAVE / TS_WAVES; ++j) {
        const int other = w * (MI_WAVE / TS_WAVES) + j;  // uniform over the wave
        const uint32_t ok = (uint32_t) __builtin_amdgcn_readlane((int) skey, other);
        pr += (ok < skey || (ok == skey && other < lane)) ? 1u : 0u;
      }
      L.cnt[w][lane] = pr;
    }
    TS_MARK(11);
    __syncthreads();
    TS_MARK(12);
    uint32_t rank = 0;
#pragma unroll
    for (int i = 0; i < TS_WAVES; ++i) rank += L.cnt[i][lane];
    unsigned long long lo = 0ull, hi = ~0ull;
    {
      const unsigned long long sc = ((unsigned long long) skey << 13) | (unsigned long long) se;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int q = part + s;
        if (q > 0 && q < parts) {
          const unsigned long long hit = __ballot(rank == (uint32_t) (q * MI_WAVE / parts));  // one lane: ranks are a permutation
          const int src = __ffsll((long long) hit) - 1;
          const unsigned long long c =
              ((unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (sc >> 32), src) << 32) |
              (unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) sc, src);
          if (s == 0) lo = c; else hi = c;
        }
      }
    }
    TS_MARK(2);
    // stable selection: entry e of the table keeps its place among the selected ones.  The masks are kept as
    // the 64-bit scalars the compares produce, so combining and counting them costs no vector instruction.
    unsigned long long bal[TS_ITEMS];
    uint32_t sel_before[TS_ITEMS];
    uint32_t nsel = 0, nbelow = 0;
#pragma unroll
    for (int k = 0; k < TS_ITEMS; ++k) {
      const int e = (w * TS_ITEMS + k) * MI_WAVE + lane;
      const unsigned long long c = ((unsigned long long) key[k] << 13) | (unsigned long long) e;
      const unsigned long long valid = __builtin_amdgcn_ballot_w64(e < n), ge = __builtin_amdgcn_ballot_w64(c >= lo),
                               lt = __builtin_amdgcn_ballot_w64(c < hi);
      bal[k] = valid & ge & lt;
      sel_before[k] = nsel;
      nsel += (uint32_t) __popcll(bal[k]);
      nbelow += (uint32_t) __popcll(valid & ~ge);
    }
    if (lane == 0) {
      L.wsel[w][0] = nsel;
      L.wsel[w][1] = nbelow;
    }
    __syncthreads();
    // lane i < 16 takes wave i's two counts; a wave scan gives every wave its first place, and the totals
    const uint32_t a_sel = lane < TS_WAVES ? L.wsel[lane][0] : 0u, a_below = lane < TS_WAVES ? L.wsel[lane][1] : 0u;
    const uint32_t i_sel = ts_wave_scan(a_sel), i_below = ts_wave_scan(a_below);
    const uint32_t wave_first = (uint32_t) __builtin_amdgcn_readlane((int) (i_sel - a_sel), w);
    m = __builtin_amdgcn_readlane((int) i_sel, MI_WAVE - 1);
    out_base = __builtin_amdgcn_readlane((int) i_below, MI_WAVE - 1);
    // the part's keys lie between the keys of its two cuts: sorting key - key_lo needs fewer bits (none at all
    // where the part is a run of one key)
    key_lo = (uint32_t) (lo >> 13);
    const uint32_t span = (part == parts - 1 ? rows : (uint32_t) (hi >> 13)) - key_lo;
    bits = 0;
    for (uint32_t v = span; v; v >>= 1) ++bits;
#pragma unroll
    for (int k = 0; k < TS_ITEMS; ++k) {
      if (__builtin_amdgcn_inverse_ballot_w64(bal[k])) {
        const uint32_t dest = __builtin_amdgcn_mbcnt_hi((uint32_t) (bal[k] >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t) bal[k], wave_first + sel_before[k]));
        kbuf[0][dest] = key[k] - key_lo;
        vbuf[0][dest] = (uint32_t) pv[k];
      }
    }
  }
  TS_MARK(3);
  const int cur = ts_lds_sort_bits(L, m, bits, 0);  // (its first barrier makes the fill visible, also for keys of zero bits)
  const size_t out = (size_t) t * n + out_base;
#pragma unroll
  for (int k = 0; k < TS_ITEMS; ++k) {
    const int e = k * TS_THREADS + tid;
    if (e < m) {
      const uint32_t lk = kbuf[cur][e] + key_lo;
      sorted_rows[out + e] = lk < rows ? (uint32_t) (base_row + lk) : sentinel;
      perm[out + e] = vbuf[cur][e];
    }
  }
  TS_MARK(8);
#ifdef TS_TRACE
  if (threadIdx.x == 0) { ts_trace_buf[blockIdx.x * 16 + 9] = (unsigned long long) bits; ts_trace_buf[blockIdx.x * 16 + 10] = (unsigned long long) m; }
#endif
}
