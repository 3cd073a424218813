// Row-wise sharded tables: the kernels either side of the RCCL exchanges.
//
// The reference distributes with TensorFlow's asynchronous parameter server
// (recommendation/code/model_trainer.py:124-136: every variable read and every Adam apply is a
// gRPC hop to the ps task).  Here the rows of all tables are dealt round-robin over the N ranks
// (global row r lives on rank r % N as local row r / N) and a step exchanges, per rank pair,
//   requester -> owner : its [T][B] matrix of local rows, -1 where the row lives elsewhere
//   owner -> requester : per SAMPLE partial pools  sum v, sum |v|^2, sum w1
// so the payload coming back is D+2 floats per sample and owner, not D+1 per (sample, field).
// Every buffer has a fixed size and nothing on the host ever waits for a count: the owner finds
// the entries addressed to it by looking at the slots.  Nothing here uses float atomics: every
// sum has a fixed order (fields in field order, owners in rank order), so a sharded run
// reproduces itself bit for bit.
#include "common.h"
#include "fm_body.h"
#include "table_sort.h"

namespace {

constexpr int MAX_RANKS = 16;
constexpr int ROUTE_SAMPLES = 64;  // samples per workgroup of the route kernel

// Requester: ids [B][T] -> one block per owner o,
//   block o = [ cnt[c][t] : chunk c = 64 consecutive samples, how many of its slots of table t are o's ]
//             [ slot[t][b] : the local row of (b, t) when rank o owns it, else -1 ]
// A workgroup takes one chunk: the ids are read as they lie (sample-major), turned around in LDS and
// written table-major, so both sides of the transpose are full-line accesses.  The counts let the
// owner place every chunk's entries without a counting pass of its own.
constexpr int ROUTE_THREADS = 1024;
template <typename IDX>
__global__ __launch_bounds__(ROUTE_THREADS) void route_kernel(const int64_t* __restrict__ toffs, int T,
                                                   const void* __restrict__ ids, int B, int N, int64_t stride,
                                                   int32_t* __restrict__ send, int32_t* __restrict__ err) {
  extern __shared__ int32_t tile[];  // [T][ROUTE_SAMPLES] local row, and the owner behind it
  int32_t* loc = tile;
  int32_t* own = tile + T * ROUTE_SAMPLES;
  const int chunk = blockIdx.x, n_chunks = gridDim.x;
  const int b0 = chunk * ROUTE_SAMPLES;
  const int nb = min(ROUTE_SAMPLES, B - b0);
  bool bad = false;
  for (int i = threadIdx.x; i < nb * T; i += ROUTE_THREADS) {
    const int bl = i / T, t = i - bl * T;
    const int64_t id = mi_ld_id<IDX>(ids, (int64_t) b0 * T + i);
    const int64_t base = toffs[t];
    const int64_t rows = toffs[t + 1] - base;
    int o = -1, l = -1;
    if (id >= 0 && id < rows) {
      const int64_t g = base + id;
      if (g < ((int64_t) 1 << 32)) {  // the usual case: a 32-bit division is a fraction of a 64-bit one
        const uint32_t g32 = (uint32_t) g;
        l = (int) (g32 / (uint32_t) N);
        o = (int) (g32 - (uint32_t) l * (uint32_t) N);
      } else {
        o = (int) (g % N);
        l = (int) (g / N);
      }
    } else {
      bad = true;
    }
    loc[t * ROUTE_SAMPLES + bl] = l;
    own[t * ROUTE_SAMPLES + bl] = o;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // a wave per (owner, table) row of 64 slots; the owners are dealt over the gridDim.y workgroups of a chunk
  const int n_mine = (N - (int) blockIdx.y + (int) gridDim.y - 1) / (int) gridDim.y;
  for (int ot = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); ot < n_mine * T; ot += ROUTE_THREADS / 64) {
    const int oi = ot / T, t = ot - oi * T;
    const int o = (int) blockIdx.y + oi * (int) gridDim.y;
    const bool mine = lane < nb && own[t * ROUTE_SAMPLES + lane] == o;
    int32_t* blk = send + (size_t) o * stride;
    if (lane < nb) blk[(size_t) n_chunks * T + (size_t) t * B + b0 + lane] = mine ? loc[t * ROUTE_SAMPLES + lane] : -1;
    const int cnt = __popcll(__ballot(mine));
    if (lane == 0) blk[chunk * T + t] = cnt;
  }
  if (bad && err && blockIdx.y == 0) atomicOr(err, MI_IDERR_RANGE);
}

// ---- owner side ------------------------------------------------------------------------------
// Received: recv[s][t][b], the slot of sample b of rank s for table t.  Table t's rows on rank r
// are the local rows [first_local(t), first_local(t+1)).
__device__ __forceinline__ int64_t first_local(const int64_t* toffs, int t, int rank, int N) {
  return (toffs[t] - rank + N - 1) / N;  // toffs[t] >= 0, rank < N: the numerator is never negative
}

struct OwnerArgs {
  mi_fm_table tab;
  const int64_t* toffs;
  const int32_t* recv;  // [N] blocks of `stride` ints: cnt[n_chunks][T], then slot[T][B]
  int64_t stride;
  int T, B, N, rank, n_chunks;
  int row_stride;  // rows of the gathered sample matrix per source rank (>= B): entry (s, b) is row s*row_stride + b
  __device__ __forceinline__ const int32_t* cnt(int s) const { return recv + (size_t) s * stride; }
  __device__ __forceinline__ const int32_t* slots(int s) const { return recv + (size_t) s * stride + (size_t) n_chunks * T; }
  float* partial;
  int32_t* err;
};

// Forward of the bags of virtual workgroup `vblock`: bag (s, b) = the valid slots recv[s][*][b];
// partial[s][b] = { sum v (DIM), sum |v|^2, sum w1 }, zeros for a bag without rows here.  Same lane
// mapping as the single-GPU forward: G lanes per bag, FP fields in flight.
template <int DIM>
__device__ __forceinline__ void owner_fwd_body(const OwnerArgs& a, int vblock, int nthreads, const int32_t* lo_tab) {
  constexpr int CH = DIM / 4;
  constexpr int G = CH > 16 ? CH : 16;
  constexpr int FP = G / CH;
  const int bag = vblock * (nthreads / G) + threadIdx.x / G;
  if (bag >= a.N * a.B) return;
  const int l = threadIdx.x % G, c = l % CH, q = l / CH;
  const int s = bag / a.B, b = bag - s * a.B;
  const int32_t* __restrict__ col = a.slots(s) + b;
  const float* __restrict__ rec = a.tab.rec;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), sq = sum;
  float lin = 0.f;
  bool bad = false;
  constexpr int UF = 8;
  for (int f0 = q; f0 < a.T; f0 += FP * UF) {
    int32_t slot[UF];
    bool ok[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int f = f0 + u * FP;
      slot[u] = f < a.T ? col[(size_t) f * a.B] : -1;
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int f = f0 + u * FP;
      const int fs = f < a.T ? f : 0;
      ok[u] = slot[u] >= lo_tab[fs] && slot[u] < lo_tab[fs + 1];
      if (slot[u] >= 0 && !ok[u]) bad = true;   // a row of another table, or past the shard
    }
    float4 v[UF];
    float w[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      w[u] = 0.f;
      if (ok[u]) {
        const float* r = rec + (int64_t) slot[u] * a.tab.stride;
        v[u] = *reinterpret_cast<const float4*>(r + c * 4);
        if (c == 0) w[u] = r[a.tab.off_w1];
      }
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      lin += w[u];
      sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w;
      sq.x = fmaf(v[u].x, v[u].x, sq.x); sq.y = fmaf(v[u].y, v[u].y, sq.y);
      sq.z = fmaf(v[u].z, v[u].z, sq.z); sq.w = fmaf(v[u].w, v[u].w, sq.w);
    }
  }
#pragma unroll
  for (int o = CH; o < G; o <<= 1) {  // the field phases of a column chunk
    sum.x += __shfl_xor(sum.x, o, 64); sum.y += __shfl_xor(sum.y, o, 64);
    sum.z += __shfl_xor(sum.z, o, 64); sum.w += __shfl_xor(sum.w, o, 64);
  }
  float qq = (sq.x + sq.y) + (sq.z + sq.w);
#pragma unroll
  for (int o = 1; o < G; o <<= 1) {
    qq += __shfl_xor(qq, o, 64);
    lin += __shfl_xor(lin, o, 64);
  }
  int anybad = bad ? 1 : 0;
#pragma unroll
  for (int o = 1; o < G; o <<= 1) anybad |= __shfl_xor(anybad, o, 64);
  float* out = a.partial + (size_t) bag * (DIM + 2);
  if (q == 0) {  // rows of DIM+2 floats are only 8-byte aligned
    *reinterpret_cast<float2*>(out + c * 4) = make_float2(sum.x, sum.y);
    *reinterpret_cast<float2*>(out + c * 4 + 2) = make_float2(sum.z, sum.w);
  }
  if (l == 0) { out[DIM] = qq; out[DIM + 1] = lin; }
  if (l == 0 && anybad && a.err) atomicOr(a.err, MI_IDERR_RANGE);
}

// ---- owner-side sort --------------------------------------------------------------------------
// Table t's entries at this owner are the valid slots among the N*B of recv[*][t][*]: B on
// average, N*B at worst, in stream order e = s*B + b - which is also the row of the gathered
// sample matrix the backward reads, so the sort's payload is simply e.  `parts` workgroups share
// a table the way the single-GPU sort does (histogram of the top 8 key bits, equal-count cut, no
// communication).  A range of more than TS_MAX entries is taken apart further: bins are sorted in
// groups that fit, an over-full bin is refined on its next 8 key bits (at most four levels for a
// 32-bit key), and an over-full single key needs no sorting at all - its entries are copied in
// stream order.  So any skew is sorted correctly; only its speed degrades.
constexpr uint32_t OS_INVALID = 0xFFFFFFFFu;
constexpr int OS_LEVELS = 4;

struct OwnerStream {
  const int32_t* slots;  // table t's slots of source 0
  int64_t stride;        // ints between consecutive sources
  int B, NC, n_pairs;    // samples per source, 64-slot chunks per source, (source, chunk) pairs = N*NC
  int RS;                // payload of entry (s, b) = s*RS + b
  int64_t lo;
  uint32_t rows;
  __device__ __forceinline__ uint32_t to_key(int32_t v) const {
    const int64_t k = (int64_t) v - lo;
    return (v >= 0 && k >= 0 && k < (int64_t) rows) ? (uint32_t) k : OS_INVALID;
  }
};

// (source, chunk) of pair p, advanced without divisions; wave-uniform (kept in scalar registers)
struct PairIter {
  int s, ch;
  __device__ __forceinline__ PairIter(int p, int NC) : s(p / NC), ch(p - (p / NC) * NC) {}
  __device__ __forceinline__ void advance(int k, int NC) {
    ch += k;
    while (ch >= NC) { ch -= NC; ++s; }
  }
};
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); }

// entries of the frame: valid keys with key >> psh == pfx; bin = (key >> sh) & mask
struct OsFrame {
  int psh;
  uint32_t pfx;
  int sh;
  uint32_t mask;
  __device__ __forceinline__ bool in(uint32_t key) const {
    return key != OS_INVALID && (uint32_t) ((unsigned long long) key >> psh) == pfx;
  }
  __device__ __forceinline__ int bin(uint32_t key) const { return (int) ((key >> sh) & mask); }
};

// Wave w walks its contiguous run of (source, chunk) pairs, OS_FLIGHT chunks in flight (what these
// passes cost is dependent round trips to memory, so few, wide rounds); f(key, e) is called by all
// lanes of the wave together for every chunk (key = OS_INVALID past a source's last sample),
// e = s*RS + b being the entry's payload (ascending in stream order).
constexpr int OS_FLIGHT = 32;
template <class F>
__device__ __forceinline__ void os_stream(const OwnerStream& S, F&& f) {
  const int lane = threadIdx.x & 63, w = wave_id();
  const int ppw = (S.n_pairs + TS_WAVES - 1) / TS_WAVES;
  const int beg = w * ppw, end = min(beg + ppw, S.n_pairs);
  if (beg >= end) return;
  PairIter it(beg, S.NC);
  for (int p0 = beg; p0 < end; p0 += OS_FLIGHT) {
    int32_t v[OS_FLIGHT];
    PairIter jt = it;
#pragma unroll
    for (int u = 0; u < OS_FLIGHT; ++u) {
      const int b = jt.ch * 64 + lane;
      v[u] = (p0 + u < end && b < S.B) ? S.slots[(size_t) jt.s * S.stride + b] : -1;
      jt.advance(1, S.NC);
    }
#pragma unroll
    for (int u = 0; u < OS_FLIGHT; ++u) {
      if (p0 + u < end) f(S.to_key(v[u]), it.s * S.RS + it.ch * 64 + lane);
      it.advance(1, S.NC);
    }
  }
}

__device__ __forceinline__ void os_hist(const OwnerStream& S, const OsFrame& fr, uint32_t* hist) {
  for (int i = threadIdx.x; i < TS_RADIX; i += TS_THREADS) hist[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  os_stream(S, [&](uint32_t key, int) {
    // integer counts: any order gives the same result.  A skewed table sends most of a wave to one
    // bin, where 64 atomics on one word would serialise: the lanes that share the first active
    // lane's bin are counted with a ballot and added once, only the others add for themselves.
    const bool in = fr.in(key);
    const int b = in ? fr.bin(key) : -1;
    const unsigned long long act = __ballot(in);
    if (act == 0) return;  // uniform over the wave
    const int lead = __ffsll((long long) act) - 1;
    const int b0 = __shfl(b, lead, 64);
    const unsigned long long same = __ballot(b == b0);
    if (lane == lead) atomicAdd(&hist[b0], (uint32_t) __popcll(same));
    if (in && b != b0) atomicAdd(&hist[b], 1u);
  });
  __syncthreads();
}

// Stable selection of the frame's entries with bin in [b0, b1): into LDS image 0 (to_lds) or, for
// a single over-full key, straight to the output at sorted_rows/perm + out.  Returns nothing; the
// count is known from the histogram.
__device__ __forceinline__ void os_select(TsLds& L, const OwnerStream& S, const OsFrame& fr, int b0, int b1,
                                          bool to_lds, uint32_t* __restrict__ sorted_rows,
                                          uint32_t* __restrict__ perm, size_t out) {
  const int lane = threadIdx.x & 63, w = wave_id();
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int mine = 0;
  os_stream(S, [&](uint32_t key, int) {
    const int b = fr.bin(key);
    mine += __popcll(__ballot(fr.in(key) && b >= b0 && b < b1));
  });
  __syncthreads();  // L.wsum may still be read by the previous user
  if (lane == 0) L.wsum[w] = (uint32_t) mine;
  __syncthreads();
  uint32_t at = 0;
  for (int i = 0; i < w; ++i) at += L.wsum[i];
  os_stream(S, [&](uint32_t key, int e) {
    const int b = fr.bin(key);
    const bool sel = fr.in(key) && b >= b0 && b < b1;
    const unsigned long long bal = __ballot(sel);
    if (sel) {
      const uint32_t d = at + (uint32_t) __popcll(bal & lt);
      if (to_lds) {
        L.kbuf[0][d] = key;
        L.vbuf[0][d] = (uint32_t) e;
      } else {
        sorted_rows[out + d] = (uint32_t) (S.lo + key);
        perm[out + d] = (uint32_t) e;
      }
    }
    at += (uint32_t) __popcll(bal);
  });
  __syncthreads();
}

// Emits, in key order, the frame's entries whose bin lies in [i, end), to the output position
// `out` (advanced).  hist = L-resident bin counts of this frame.
template <int LEVEL>
__device__ __forceinline__ void os_emit(TsLds& L, uint32_t (*lvl)[TS_RADIX], const OwnerStream& S, const OsFrame& fr,
                                        int i, int end, int bits, uint32_t* __restrict__ sorted_rows,
                                        uint32_t* __restrict__ perm, size_t& out) {
  const uint32_t* hist = lvl[LEVEL];
  while (i < end) {  // uniform over the workgroup
    const uint32_t c = hist[i];
    if (c == 0) { ++i; continue; }
    if (c > (uint32_t) TS_MAX) {
      if (fr.sh == 0) {  // one key: no order to establish
        os_select(L, S, fr, i, i + 1, false, sorted_rows, perm, out);
        out += c;
      } else if constexpr (LEVEL + 1 < OS_LEVELS) {
        OsFrame ch;
        ch.psh = fr.sh;
        ch.pfx = (uint32_t) ((((unsigned long long) fr.pfx << (fr.psh - fr.sh)) & 0xFFFFFFFFull) | (uint32_t) i);
        ch.sh = fr.sh > 8 ? fr.sh - 8 : 0;
        ch.mask = (1u << (fr.sh - ch.sh)) - 1u;
        os_hist(S, ch, lvl[LEVEL + 1]);
        os_emit<LEVEL + 1>(L, lvl, S, ch, 0, (int) ch.mask + 1, bits, sorted_rows, perm, out);
      }
      ++i;
      continue;
    }
    int j = i;
    uint32_t m = 0;
    while (j < end && m + hist[j] <= (uint32_t) TS_MAX) m += hist[j++];
    os_select(L, S, fr, i, j, true, nullptr, nullptr, 0);
    const int nbits = min(bits, fr.sh + 8);
    const int cur = ts_lds_sort(L, (int) m, (nbits + 7) / 8, 0);
    for (int e = threadIdx.x; e < (int) m; e += TS_THREADS) {
      sorted_rows[out + e] = (uint32_t) (S.lo + L.kbuf[cur][e]);
      perm[out + e] = L.vbuf[cur][e];
    }
    __syncthreads();  // the images are free again
    out += m;
    i = j;
  }
}

// The n <= 2*TS_MAX valid (key, value) pairs of a table lie in the two key images / two value images
// of L taken as one array each.  Cuts the key range as ts_sort_part does; when this part's share
// fits one image (m <= TS_MAX) it is compacted in place to the front (stable), sorted, and the
// image holding it is returned.  A larger share that is one key is written straight to
// sorted_rows / perm [base + ..) (returns -2); any other larger share returns -1, nothing moved.
// All threads call this.
constexpr int OS_FLAT = 2 * TS_MAX;
constexpr int OS_ITEMS = OS_FLAT / TS_THREADS;  // 16
__device__ __forceinline__ int os_sort_flat(TsLds& L, int n, int part, int parts, int bits, int& m_out,
                                            int& out_base_out, uint32_t lo, uint32_t* __restrict__ sorted_rows,
                                            uint32_t* __restrict__ perm, size_t base) {
  uint32_t* fk = &L.kbuf[0][0];
  uint32_t* fv = &L.vbuf[0][0];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int top = bits > 8 ? bits - 8 : 0;
  if (tid < TS_RADIX) L.dbase[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < OS_ITEMS; ++k) {
    const int e = k * TS_THREADS + tid;
    if (e < n) atomicAdd(&L.dbase[fk[e] >> top], 1u);  // integer counts: any order gives the same result
  }
  __syncthreads();
  uint32_t c = tid < TS_RADIX ? L.dbase[tid] : 0u, inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t x = __shfl_up(inc, o, 64);
    if (lane >= o) inc += x;
  }
  if (tid < TS_RADIX && lane == 63) L.wsum[w] = inc;
  __syncthreads();
  if (tid < TS_RADIX) {
    uint32_t wb = 0;
    for (int i = 0; i < w; ++i) wb += L.wsum[i];
    L.qsum[0][tid] = wb + inc - c;
  }
  __syncthreads();
  if (tid < TS_RADIX) {
    const uint32_t mine = L.qsum[0][tid], next = tid + 1 < TS_RADIX ? L.qsum[0][tid + 1] : 0xFFFFFFFFu;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = part + s;
      if (q > 0 && q < parts) {
        const uint32_t target = (uint32_t) (((int64_t) q * n) / parts);
        if (mine <= target && next > target) L.cut[s] = tid;
      }
    }
  }
  if (tid == 0) {
    if (part == 0) L.cut[0] = 0;
    if (part == parts - 1) L.cut[1] = TS_RADIX;
  }
  __syncthreads();
  const int b_lo = L.cut[0], b_hi = L.cut[1];
  const int out_base = (int) L.qsum[0][b_lo];
  const int m = (b_hi == TS_RADIX ? n : (int) L.qsum[0][b_hi]) - out_base;
  m_out = m;
  out_base_out = out_base;
  // a share that does not fit one image: hopeless here unless it is a single key (then its entries
  // only have to be copied out in order)
  const bool direct = m > TS_MAX && top == 0 && b_hi == b_lo + 1;
  if (m > TS_MAX && !direct) return -1;
  // stable selection, in place: everything is read into registers before anything is written
  unsigned long long bal[OS_ITEMS];
  uint32_t key[OS_ITEMS], val[OS_ITEMS];
  const int wbase = w * (MI_WAVE * OS_ITEMS);
#pragma unroll
  for (int j = 0; j < OS_ITEMS; ++j) {
    const int e = wbase + j * MI_WAVE + lane;
    key[j] = e < n ? fk[e] : 0u;
    val[j] = e < n ? fv[e] : 0u;
    const int bin = (int) (key[j] >> top);
    bal[j] = __ballot(e < n && bin >= b_lo && bin < b_hi);
    if (lane == 0) L.qsum[1][w * OS_ITEMS + j] = (uint32_t) __popcll(bal[j]);
  }
  __syncthreads();
  if (tid < TS_WAVES * OS_ITEMS) {  // 256 group counts -> exclusive prefix (four waves)
    const uint32_t g = L.qsum[1][tid];
    uint32_t gi = g;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t x = __shfl_up(gi, o, 64);
      if (lane >= o) gi += x;
    }
    if (lane == 63) L.wsum[w] = gi;
    L.qsum[2][tid] = gi - g;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < OS_ITEMS; ++j) {
    if ((bal[j] >> lane) & 1ull) {
      const int g = w * OS_ITEMS + j;
      uint32_t before = 0;
      for (int i = 0; i < (g >> 6); ++i) before += L.wsum[i];
      const uint32_t dest = L.qsum[2][g] + before + (uint32_t) __popcll(bal[j] & lt);
      if (direct) {
        sorted_rows[base + out_base + dest] = lo + key[j];
        perm[base + out_base + dest] = val[j];
      } else {
        fk[dest] = key[j];
        fv[dest] = val[j];
      }
    }
  }
  if (direct) return -2;
  __syncthreads();
  const int passes = (bits + 7) / 8;
  const int cur = ts_lds_sort(L, m, passes, 0);
  if (passes == 0) __syncthreads();
  return cur;
}

// Part `part` of `parts` of table t.  The output is compact: table t's entries follow those of the
// tables before it (their numbers come from the chunk counts the requesters sent along), and
// n_total[0] = the number of entries of all tables.
//
// Usual case, at most 2*TS_MAX entries for the table (the average is B) and a share of at most
// TS_MAX for this part: the chunk counts give every 64-slot chunk its place, so ONE pass over the
// slots compacts the valid ones into LDS in stream order, and the rest is the single-GPU per-table
// sort.  Otherwise the streaming multi-level path above (same cut, so the parts of a table may
// take different paths).
__device__ __forceinline__ void owner_sort_body(TsLds& L, uint32_t (*lvl)[TS_RADIX], const OwnerArgs& a, int t, int part,
                                                int parts, uint32_t* __restrict__ sorted_rows,
                                                uint32_t* __restrict__ perm, int32_t* __restrict__ n_total) {
  const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int T = a.T, N = a.N, NC = a.n_chunks;
  OwnerStream S;
  S.slots = a.slots(0) + (size_t) t * a.B;
  S.stride = a.stride;
  S.B = a.B; S.NC = NC; S.n_pairs = N * NC; S.RS = a.row_stride;
  S.lo = first_local(a.toffs, t, a.rank, N);
  S.rows = (uint32_t) (first_local(a.toffs, t + 1, a.rank, N) - S.lo);
  int bits = 0;
  for (uint32_t v = S.rows; v; v >>= 1) ++bits;
  // entries per table from the chunk counts: thread (g, t') sums cnt[s][c][t'] over the (s, c) pairs
  // g, g+G, ...; then the G partials of a table are added in order
  uint32_t* tsum = &lvl[0][0];  // [T <= 64] totals, then [T+1] exclusive prefix behind them
  const bool vec = (a.B % 256 == 0) && (((uintptr_t) a.recv & 15) == 0);  // then every block, count row and slot row is 16-byte aligned
  if (vec) {
    // integer sums: LDS atomics, any order gives the same totals
    if (tid < 64) tsum[tid] = 0;
    __syncthreads();
    const int per_src = NC * T / 4;  // int4 per source (NC % 4 == 0)
    for (int s = 0; s < N; ++s) {
      const int4* c4 = reinterpret_cast<const int4*>(a.cnt(s));
      for (int i = tid; i < per_src; i += TS_THREADS) {
        const int4 v = c4[i];
        int tp = (4 * i) % T;
        atomicAdd(&tsum[tp], (uint32_t) v.x); tp = tp + 1 == T ? 0 : tp + 1;
        atomicAdd(&tsum[tp], (uint32_t) v.y); tp = tp + 1 == T ? 0 : tp + 1;
        atomicAdd(&tsum[tp], (uint32_t) v.z); tp = tp + 1 == T ? 0 : tp + 1;
        atomicAdd(&tsum[tp], (uint32_t) v.w);
      }
    }
    __syncthreads();
  } else {
    const int tp = lane, g = w;  // G = TS_WAVES groups
    uint32_t acc = 0;
    PairIter it(g, NC);
    for (int sc0 = g; sc0 < N * NC; sc0 += TS_WAVES * 8) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = (sc0 + u * TS_WAVES < N * NC && tp < T) ? (uint32_t) a.cnt(it.s)[it.ch * T + tp] : 0u;
        it.advance(TS_WAVES, NC);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    L.cnt[g][tp] = acc;
    __syncthreads();
    if (tid < T) {
      uint32_t tot = 0;
      for (int i = 0; i < TS_WAVES; ++i) tot += L.cnt[i][tid];
      tsum[tid] = tot;
    }
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t run = 0;
    for (int i = 0; i < T; ++i) { tsum[64 + i] = run; run += tsum[i]; }
    tsum[64 + T] = run;
  }
  __syncthreads();
  const uint32_t n_valid = tsum[t];
  const size_t base = tsum[64 + t];
  if (t == 0 && part == 0 && tid == 0) n_total[0] = (int32_t) tsum[64 + T];
  __syncthreads();  // lvl[0] is reused below
  // Units of the slot pass: 64-slot chunks, or groups of four chunks read as one int4 per lane when
  // everything is 16-byte aligned (B % 256 == 0).  Unit u = (source, unit of that source).
  const int UPS = vec ? NC / 4 : NC;  // units per source
  const int n_units = N * UPS;
  const bool one_key = S.rows == 1;   // a single local row: nothing to sort, any number of entries
  if (n_units <= TS_WAVES * TS_RADIX &&
      (one_key || (n_valid <= (uint32_t) OS_FLAT && (parts > 1 || n_valid <= (uint32_t) TS_MAX)))) {
    uint32_t* fk = &L.kbuf[0][0];
    uint32_t* fv = &L.vbuf[0][0];
    // place of every unit: exclusive prefix of this table's chunk counts, kept in L.cnt
    uint32_t* cbase = &L.cnt[0][0];
    uint32_t carry = 0;
    for (int p0 = 0; p0 < n_units; p0 += TS_THREADS) {  // uniform
      const int p = p0 + tid;
      uint32_t c = 0;
      if (p < n_units) {
        const int s = p / UPS, j = p - s * UPS;
        const int32_t* cs = a.cnt(s);
        if (vec) {
#pragma unroll
          for (int i = 0; i < 4; ++i) c += (uint32_t) cs[(4 * j + i) * T + t];
        } else {
          c = (uint32_t) cs[j * T + t];
        }
      }
      uint32_t inc = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(inc, o, 64);
        if (lane >= o) inc += x;
      }
      if (lane == 63) L.wsum[w] = inc;
      __syncthreads();
      uint32_t wb = 0, all = 0;
      for (int i = 0; i < TS_WAVES; ++i) {
        if (i < w) wb += L.wsum[i];
        all += L.wsum[i];
      }
      if (p < n_units) cbase[p] = carry + wb + inc - c;
      carry += all;
      __syncthreads();
    }
    // One pass over the slots.  one_key: the parts share the units and write the entries straight to
    // their places; otherwise every part stages all of the table's entries in LDS (wave w takes
    // units w, w+16, ...).
    const int u_lo = one_key ? (int) ((int64_t) n_units * part / parts) : 0;
    const int u_hi = one_key ? (int) ((int64_t) n_units * (part + 1) / parts) : n_units;
    auto put = [&](uint32_t d, uint32_t key, uint32_t e) {
      if (one_key) {
        sorted_rows[base + d] = (uint32_t) S.lo;
        perm[base + d] = e;
      } else if (d < (uint32_t) OS_FLAT) {  // counts and slots disagree only if the exchange was corrupted
        fk[d] = key;
        fv[d] = e;
      }
    };
    const uint32_t kmax = S.rows ? S.rows - 1 : 0;  // a slot of another table (reported by the forward) must not leave the key range
    if (vec) {
      constexpr int FL = 8;
      PairIter it(u_lo + w, UPS);
      for (int p0 = u_lo + w; p0 < u_hi; p0 += TS_WAVES * FL) {
        int4 v[FL];
        PairIter jt = it;
#pragma unroll
        for (int u = 0; u < FL; ++u) {
          v[u] = make_int4(-1, -1, -1, -1);
          if (p0 + u * TS_WAVES < u_hi)
            v[u] = *reinterpret_cast<const int4*>(S.slots + (size_t) jt.s * S.stride + jt.ch * 256 + lane * 4);
          jt.advance(TS_WAVES, UPS);
        }
#pragma unroll
        for (int u = 0; u < FL; ++u) {
          const int p = p0 + u * TS_WAVES;
          if (p < u_hi) {  // uniform over the wave
            const int32_t x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            const unsigned long long any = __ballot((x[0] & x[1] & x[2] & x[3]) >= 0);
            if (any) {
              uint32_t before = 0;
#pragma unroll
              for (int k = 0; k < 4; ++k) before += (uint32_t) __popcll(__ballot(x[k] >= 0) & lt);
              uint32_t d = cbase[p] + before;
              const uint32_t e0 = (uint32_t) (it.s * a.row_stride + it.ch * 256 + lane * 4);
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (x[k] >= 0) put(d++, min((uint32_t) ((int64_t) x[k] - S.lo), kmax), e0 + k);
            }
          }
          it.advance(TS_WAVES, UPS);
        }
      }
    } else {
      constexpr int FL = 8;
      PairIter it(u_lo + w, UPS);
      for (int p0 = u_lo + w; p0 < u_hi; p0 += TS_WAVES * FL) {
        int32_t v[FL];
        PairIter jt = it;
#pragma unroll
        for (int u = 0; u < FL; ++u) {
          const int b = jt.ch * 64 + lane;
          v[u] = (p0 + u * TS_WAVES < u_hi && b < a.B) ? S.slots[(size_t) jt.s * S.stride + b] : -1;
          jt.advance(TS_WAVES, UPS);
        }
#pragma unroll
        for (int u = 0; u < FL; ++u) {
          const int p = p0 + u * TS_WAVES;
          if (p < u_hi) {  // uniform over the wave
            const bool ok = v[u] >= 0;
            const unsigned long long bal = __ballot(ok);
            if (ok)
              put(cbase[p] + (uint32_t) __popcll(bal & lt), min((uint32_t) ((int64_t) v[u] - S.lo), kmax),
                  (uint32_t) (it.s * a.row_stride + it.ch * 64 + lane));
          }
          it.advance(TS_WAVES, UPS);
        }
      }
    }
    if (one_key) return;
    __syncthreads();
    int m, out_base, cur;
    if (parts > 1) {
      cur = os_sort_flat(L, (int) n_valid, part, parts, bits, m, out_base, (uint32_t) S.lo, sorted_rows, perm, base);
    } else {
      m = (int) n_valid;
      out_base = 0;
      cur = ts_lds_sort(L, m, (bits + 7) / 8, 0);
      if (bits == 0) __syncthreads();
    }
    if (cur == -2) return;  // written by os_sort_flat
    if (cur >= 0) {
      const size_t out = base + out_base;
      for (int e = tid; e < m; e += TS_THREADS) {
        sorted_rows[out + e] = (uint32_t) (S.lo + L.kbuf[cur][e]);
        perm[out + e] = L.vbuf[cur][e];
      }
      return;
    }
    __syncthreads();  // this part's share does not fit: stream it
  }
  // streaming path
  OsFrame top;
  top.psh = 32; top.pfx = 0;
  top.sh = bits > 8 ? bits - 8 : 0;
  top.mask = TS_RADIX - 1;
  os_hist(S, top, lvl[0]);
  uint32_t c = tid < TS_RADIX ? lvl[0][tid] : 0u, inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t x = __shfl_up(inc, o, 64);
    if (lane >= o) inc += x;
  }
  if (tid < TS_RADIX && lane == 63) L.wsum[w] = inc;
  __syncthreads();
  const uint32_t n_seen = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
  if (tid < TS_RADIX) {
    uint32_t wb = 0;
    for (int i = 0; i < w; ++i) wb += L.wsum[i];
    L.dbase[tid] = wb + inc - c;
  }
  __syncthreads();
  if (tid < TS_RADIX) {
    const uint32_t mine = L.dbase[tid], next = tid + 1 < TS_RADIX ? L.dbase[tid + 1] : 0xFFFFFFFFu;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = part + s;
      if (q > 0 && q < parts) {
        const uint32_t target = (uint32_t) (((int64_t) q * n_seen) / parts);
        if (mine <= target && next > target) L.cut[s] = tid;
      }
    }
  }
  if (tid == 0) {
    if (part == 0) L.cut[0] = 0;
    if (part == parts - 1) L.cut[1] = TS_RADIX;
  }
  __syncthreads();
  const int b_lo = L.cut[0], b_hi = L.cut[1];
  size_t out = base + L.dbase[b_lo];
  __syncthreads();  // L.dbase and L.wsum are reused by the sort
  os_emit<0>(L, lvl, S, top, b_lo, b_hi, bits, sorted_rows, perm, out);
}

// One launch for everything an owner can do once the slots have arrived: workgroups [0, T*parts)
// sort, the others run the partial forward, one thread advances the optimizer schedule.
template <int DIM>
__global__ __launch_bounds__(TS_THREADS) void owner_fwd_sort_kernel(OwnerArgs a, int parts,
                                                                   uint32_t* __restrict__ sorted_rows,
                                                                   uint32_t* __restrict__ perm,
                                                                   int32_t* __restrict__ n_total,
                                                                   float* __restrict__ sched, float lr, float b1,
                                                                   float b2) {
  const int sort_blocks = sorted_rows ? a.T * parts : 0;
  if ((int) blockIdx.x < sort_blocks) {
    __shared__ TsLds L;
    __shared__ uint32_t lvl[OS_LEVELS][TS_RADIX];
    owner_sort_body(L, lvl, a, blockIdx.x / parts, blockIdx.x % parts, parts, sorted_rows, perm, n_total);
    return;
  }
  extern __shared__ int32_t lo_tab[];  // first local row of every table, T+1 entries
  for (int t = threadIdx.x; t <= a.T; t += TS_THREADS) lo_tab[t] = (int32_t) first_local(a.toffs, t, a.rank, a.N);
  __syncthreads();
  if (sched && (int) blockIdx.x == sort_blocks && threadIdx.x == 0) adam_schedule_body(sched, lr, b1, b2);
  owner_fwd_body<DIM>(a, blockIdx.x - sort_blocks, TS_THREADS, lo_tab);
}

// Requester side: sum the owners' partials in rank order, add the dense part and the bias, and
// write one row per sample of what the other ranks need:
//   X[b] = [ S (DIM) | dL/dlogit | BCE loss row | padding ]   (ld floats; with ld >= DIM+2+n_dense the
//          dense features follow the loss, for callers that want the whole sample in one row)
// so that a single all-gather of X feeds every owner's backward and every replica's dense step.
template <int DIM>
__global__ __launch_bounds__(256) void fm_combine_kernel(const float* __restrict__ partials, int N, int B,
                                                        const float* __restrict__ dense, int n_dense,
                                                        const float* __restrict__ wd, const float* __restrict__ bias,
                                                        const float* __restrict__ labels, float gscale,
                                                        float* __restrict__ logit, float* __restrict__ X, int ld,
                                                        float* __restrict__ part_out, int ct) {
  constexpr int CH = DIM / 4;
  constexpr int SPB = 256 / CH;  // samples per workgroup
  __shared__ float sh_g[SPB], sh_l[SPB], sh_red[256];
  const int bl = threadIdx.x / CH;
  const int b = blockIdx.x * SPB + bl;
  const int c = threadIdx.x % CH;
  const bool live = b < B;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float q = 0.f, lin = 0.f;
  if (live) {
  for (int r = 0; r < N; ++r) {
    const float* p = partials + ((size_t) r * B + b) * (DIM + 2);
    s.x += p[c * 4 + 0]; s.y += p[c * 4 + 1]; s.z += p[c * 4 + 2]; s.w += p[c * 4 + 3];
    if (c == 0) { q += p[DIM]; lin += p[DIM + 1]; }
  }
  float* x = X + (size_t) b * ld;
  *reinterpret_cast<float4*>(x + c * 4) = s;
  float ss = s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
  const bool keep_dense = ld >= DIM + 2 + n_dense;
  for (int j = c; j < n_dense; j += CH) {
    const float d = dense[(size_t) b * n_dense + j];
    if (keep_dense) x[DIM + 2 + j] = d;
    lin = fmaf(wd[j], d, lin);
  }
  for (int j = DIM + 2 + (keep_dense ? n_dense : 0) + c; j < ld; j += CH) x[j] = 0.f;
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
    sh_g[bl] = gb;
    sh_l[bl] = loss;
  }
  } else if (c == 0) {
    sh_g[bl] = 0.f;
    sh_l[bl] = 0.f;
  }
  if (!part_out) return;  // uniform
  // This workgroup's share of the dense-side sums, columns j < n_dense: sum g*dense[.][j]; then
  // sum g; then sum loss.  Thread (r, j) adds samples r, r+rl, ... in order, then a tree over r.
  __syncthreads();
  const int rl = 256 / ct, j = threadIdx.x % ct, r0 = threadIdx.x / ct;
  const int b0 = blockIdx.x * SPB;
  float acc = 0.f;
  if (j < n_dense + 2)
    for (int i = r0; i < SPB && b0 + i < B; i += rl)
      acc = j < n_dense ? fmaf(sh_g[i], dense[(size_t) (b0 + i) * n_dense + j], acc) : acc + (j == n_dense ? sh_g[i] : sh_l[i]);
  sh_red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = rl / 2; st > 0; st >>= 1) {
    if (r0 < st) sh_red[threadIdx.x] += sh_red[threadIdx.x + st * ct];
    __syncthreads();
  }
  if (r0 == 0) part_out[(size_t) blockIdx.x * ct + j] = sh_red[j];
}

}  // namespace

extern "C" int64_t mi_shard_block_ints(int64_t B, int T) { return mi_cdiv(B, ROUTE_SAMPLES) * T + B * T; }

extern "C" int mi_shard_route(const int64_t* table_offsets, int T, const void* ids, int idx_bits, int64_t B,
                              int n_ranks, int32_t* send, int32_t* err, void* stream) {
  MI_REQUIRE(table_offsets && ids && send, "mi_shard_route: null pointer");
  MI_REQUIRE(T >= 1 && T <= 64 && B >= 1 && B < ((int64_t) 1 << 31) / T / 17, "mi_shard_route: bad T/B (T <= 64)");
  MI_REQUIRE(n_ranks >= 1 && n_ranks <= MAX_RANKS, "mi_shard_route: n_ranks must be 1..%d", MAX_RANKS);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "mi_shard_route: idx_bits must be 32 or 64");
  hipStream_t s = (hipStream_t) stream;
  const unsigned grid = (unsigned) mi_cdiv(B, ROUTE_SAMPLES);
  const size_t lds = (size_t) 2 * T * ROUTE_SAMPLES * sizeof(int32_t);
  // enough workgroups for the chip: the owners of a chunk are shared by up to four of them
  unsigned osplit = 1;
  while (osplit < 4 && osplit * 2 <= (unsigned) n_ranks && grid * osplit < 256) osplit *= 2;
  if (idx_bits == 32)
    hipLaunchKernelGGL(route_kernel<int32_t>, dim3(grid, osplit), dim3(ROUTE_THREADS), lds, s, table_offsets, T, ids, (int) B, n_ranks,
                       mi_shard_block_ints(B, T), send, err);
  else
    hipLaunchKernelGGL(route_kernel<int64_t>, dim3(grid, osplit), dim3(ROUTE_THREADS), lds, s, table_offsets, T, ids, (int) B, n_ranks,
                       mi_shard_block_ints(B, T), send, err);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_shard_owner_fwd_sort(const mi_fm_table* tab, int64_t local_rows, const int64_t* table_offsets, int T,
                                       int rank, int n_ranks, const int32_t* recv, int64_t B, int64_t row_stride,
                                       float* partial,
                                       uint32_t* sorted_rows, int32_t* perm, int32_t* n_total, float* sched,
                                       float lr, float beta1, float beta2, int32_t* err, void* stream) {
  const char* who = "mi_shard_owner_fwd_sort";
  const int rc = mi_fm_check_table(tab, MI_OPT_SGD, who);
  if (rc) return rc;
  MI_REQUIRE(table_offsets && recv && partial, "%s: null pointer", who);
  MI_REQUIRE((sorted_rows == nullptr) == (perm == nullptr) && (sorted_rows == nullptr) == (n_total == nullptr),
             "%s: sorted_rows, perm and n_total go together", who);
  MI_REQUIRE(T <= 64, "%s: at most 64 tables", who);
  MI_REQUIRE(local_rows >= 1 && local_rows < (int64_t) 0x7FFFFFFFll, "%s: local_rows must fit an int32 slot", who);
  MI_REQUIRE(n_ranks >= 1 && n_ranks <= MAX_RANKS && rank >= 0 && rank < n_ranks && T >= 1 && B >= 1 &&
                 B < ((int64_t) 1 << 24) / n_ranks && (int64_t) n_ranks * B * T < ((int64_t) 1 << 31),
             "%s: bad rank/n_ranks/B/T (n_ranks*B < 2^24)", who);
  MI_REQUIRE(row_stride >= B && (int64_t) n_ranks * row_stride < ((int64_t) 1 << 31), "%s: row_stride must be >= B", who);
  OwnerArgs a{*tab, table_offsets, recv, mi_shard_block_ints(B, T), T, (int) B, n_ranks, rank,
              (int) mi_cdiv(B, ROUTE_SAMPLES), (int) row_stride, partial, err};
  hipStream_t s = (hipStream_t) stream;
  const int64_t bags = (int64_t) n_ranks * B;
  const size_t lds = (size_t) (T + 1) * sizeof(int32_t);
#define OW_CASE(D)                                                                                                  \
  case D: {                                                                                                         \
    constexpr int G_ = (D / 4) > 16 ? (D / 4) : 16;                                                                 \
    const int fwd_blocks_ = (int) mi_cdiv(bags, TS_THREADS / G_);                                                   \
    const int parts_ = sorted_rows ? (mi_table_sort_parts(T, (int) B, 0) < 4 ? mi_table_sort_parts(T, (int) B, 0) : 4) : 0;                                        \
    hipLaunchKernelGGL((owner_fwd_sort_kernel<D>), dim3((unsigned) (T * parts_ + fwd_blocks_)), dim3(TS_THREADS),   \
                       lds, s, a, parts_, sorted_rows, reinterpret_cast<uint32_t*>(perm), n_total, sched, lr, beta1, \
                       beta2);                                                                                      \
    break;                                                                                                          \
  }
  switch (tab->dim) {
    OW_CASE(4) OW_CASE(8) OW_CASE(16) OW_CASE(32) OW_CASE(64) OW_CASE(128)
    default:
      mi_set_error("%s: dim %d not built (4, 8, 16, 32, 64, 128)", who, tab->dim);
      return MI_ERR_UNSUPPORTED;
  }
#undef OW_CASE
  MI_LAUNCH_CHECK();
  return MI_OK;
}

static int dense_ct(int n_dense) {
  int ct = 1;
  while (ct < n_dense + 2) ct <<= 1;
  return ct;
}

extern "C" int64_t mi_shard_dense_part_floats(int64_t B, int dim, int n_dense) {
  return dim >= 4 ? mi_cdiv(B, 256 / (dim / 4)) * dense_ct(n_dense) : 0;
}

extern "C" int mi_fm_combine(const float* partials, int n_ranks, int64_t B, int dim, const float* dense,
                             int n_dense, const float* wd, const float* bias, const float* labels, float gscale,
                             float* logit, float* X, int ld, float* part_out, void* stream) {
  MI_REQUIRE(n_dense + 2 <= 256, "mi_fm_combine: at most 254 dense features");
  MI_REQUIRE(partials && X && n_ranks >= 1 && B >= 1, "mi_fm_combine: bad arguments");
  MI_REQUIRE(n_dense == 0 || (dense && wd), "mi_fm_combine: dense features need dense and wd");
  MI_REQUIRE(ld >= dim + 2 && ld % 4 == 0 && ((uintptr_t) X & 15) == 0,
             "mi_fm_combine: X rows need ld >= dim+2, ld %% 4 == 0, 16-byte alignment");
  hipStream_t s = (hipStream_t) stream;
#define CB_CASE(D)                                                                                                 \
  case D:                                                                                                          \
    hipLaunchKernelGGL(fm_combine_kernel<D>, dim3((unsigned) mi_cdiv(B, 256 / (D / 4))), dim3(256), 0, s, partials, \
                       n_ranks, (int) B, dense, n_dense, wd, bias, labels, gscale, logit, X, ld, part_out,          \
                       dense_ct(n_dense));                                                                          \
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
