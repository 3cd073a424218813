// DIN local activation unit, forward, ONE launch with one workgroup per sample: the sample's history rows
// are gathered once, its [k | q*k] rows and both hidden layers live in LDS, and the three layers run on the
// matrix cores out of LDS - the [B*L][128] matrices are still written (the backward reads them) but no
// longer read back by a second, third and fourth launch.  Same arithmetic in the same order as the
// separate launches (mi_din_lau_build, mi_din_qterm_fwd, mi_gemm_bf16 x 2, mi_head_fwd, mi_din_lau_pool):
// every output is bit-identical to theirs.
//
// Not in the reference: its nearest analogue is the 6-id bag with uniform weights
// (recommendation/code/r_model.py:68-80); see din.hip for the unit itself.
#include "common.h"

namespace {

typedef short s8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int FD = 64;           // embedding width
constexpr int FH1 = 128;         // first hidden layer (padded)
constexpr int FH2 = 64;          // second hidden layer (padded)
constexpr int FROWS = 112;       // rows of a sample's tile: 7 fragments of 16, L <= FROWS
constexpr int FFRAGS = FROWS / 16;
constexpr int IMG = FROWS * 128;  // bytes of the LDS image of [FROWS][64] bf16

// [rows][64 k] bf16 as 128-byte rows of eight 16-byte chunks, chunk position XORed with the row: the
// 16 rows of an MFMA operand fragment spread over all banks (the layout mi_gemm_bf16 stages its tiles in)
__device__ __forceinline__ int img_off(int r, int ch) { return r * 128 + ((ch ^ (r & 7)) << 4); }

struct FusedArgs {
  const float* E;
  int64_t rows;
  const void* hist;
  const void* qid;
  const int32_t* len;
  int64_t B;
  int L;
  const unsigned short* W1f;  // [FH1][2*FD] bf16: columns for k, then for q*k
  const float* rowbias;       // [B][FH1]: b1 + Wq q, the first layer's per-sample term
  const unsigned short* W2;   // [FH2][FH1] bf16
  const float* b2;            // [FH2]
  const float* w3;            // [FH2] weights, bias behind them
  unsigned short* A;          // [B*L][2*FD]
  unsigned short* H1;         // [B*L][FH1]
  unsigned short* H2;         // [B*L][FH2]
  float* a;                   // [B*L]
  int32_t* ids_all;           // [B][L+1]
  float* feat;                // [B][ld_feat] = [pooled | q]
  int64_t ld_feat;
  int32_t* err;
};

struct FusedLds {
  char a0[IMG];    // k block of the input rows (kept to the end: the pooling reads it)
  char a1[IMG];    // q*k block; then columns 0-63 of the first hidden layer; then the head's partial sums
  char hb[IMG];    // columns 64-127 of the first hidden layer; then the second hidden layer
  float rowbias[FH1];
  float w[FROWS];
};

#ifdef DIN_TRACE
static __device__ unsigned long long din_trace_buf[64 * 16];
#define DIN_MARK(k) do { if (t == 0 && it < 4 && (blockIdx.x & ~15u) == 4096u) din_trace_buf[(blockIdx.x & 15) * 64 + it * 16 + (k)] = wall_clock64(); } while (0)
#else
#define DIN_MARK(k) do { } while (0)
#endif

// Barrier between phases that hand data over through LDS only.  __syncthreads() also drains every outstanding
// GLOBAL operation of the wave - here the copies of the tiles to HBM and the next sample's prefetched rows, i.e.
// exactly the traffic that is meant to stay in flight across the phases.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint2 pack4(float x, float y, float z, float w) {
  return make_uint2((uint32_t) mi_f2bf(x) | ((uint32_t) mi_f2bf(y) << 16), (uint32_t) mi_f2bf(z) | ((uint32_t) mi_f2bf(w) << 16));
}

// rows [0, L) of a two-image tile (128 columns) or a one-image tile (64 columns) to global memory, 16 bytes a lane,
// whole rows contiguous
template <int IMAGES>
__device__ __forceinline__ void copy_out(const char* img0, const char* img1, unsigned short* dst, int L) {
  constexpr int CPR = IMAGES * 8;  // 16-byte chunks per row
#pragma unroll
  for (int s = 0; s < (FROWS * CPR + 255) / 256; ++s) {
    const int idx = threadIdx.x + 256 * s;
    const int r = idx / CPR, c = idx % CPR;
    if (r < L) {
      const char* src = (c < 8 ? img0 : img1) + img_off(r, c & 7);
      *reinterpret_cast<uint4*>(dst + (int64_t) r * (CPR * 8) + c * 8) = *reinterpret_cast<const uint4*>(src);
    }
  }
}

// what a thread has requested for a sample: its 16-byte quarter of up to FFRAGS history rows and of the
// candidate's row, and one element of the sample's bias row
struct Gathered {
  float4 k[FFRAGS];
  float4 q;
  float rowbias;
  int32_t id[FFRAGS];  // the row's id; -1 masked, -2 out of range
};
// the raw ids of a sample as loaded (nothing is computed from them until the rows are requested: a use right
// behind the loads would make the wave wait for them, and for every older load with them)
template <typename IDX>
struct Ids {
  IDX kid[FFRAGS];
  int n;
};

template <typename IDX>
__device__ __forceinline__ Ids<IDX> load_ids(const FusedArgs& p, int64_t b, int g) {
  Ids<IDX> r;
  r.n = p.len[b];
#pragma unroll
  for (int s = 0; s < FFRAGS; ++s) {
    const int j = g + 16 * s;
    r.kid[s] = reinterpret_cast<const IDX*>(p.hist)[b * p.L + (j < p.L ? j : 0)];  // read unconditionally, masked later
  }
  return r;
}

template <typename IDX>
__device__ __forceinline__ Gathered request_rows(const FusedArgs& p, const Ids<IDX>& ids, int64_t b, int g, int c, int t) {
  Gathered r;
  r.q = *reinterpret_cast<const float4*>(p.feat + b * p.ld_feat + FD + c * 4);
  r.rowbias = p.rowbias[b * FH1 + (t & (FH1 - 1))];
  const int n = ids.n < 0 ? 0 : (ids.n > p.L ? p.L : ids.n);
#pragma unroll
  for (int s = 0; s < FFRAGS; ++s) {
    const int j = g + 16 * s;
    const int64_t id = (int64_t) ids.kid[s];
    r.id[s] = (j >= n || id < 0) ? -1 : (id >= p.rows ? -2 : (int32_t) id);
    r.k[s] = *reinterpret_cast<const float4*>(p.E + (int64_t) (r.id[s] >= 0 ? r.id[s] : 0) * FD + c * 4);
  }
  return r;
}

// The candidate's row of every sample: feat[b][FD..2FD) = E[qid[b]] (zeros when out of range, which is reported),
// ids_all[b][L] = the id or -1.  16 lanes per sample.
template <typename IDX>
__global__ __launch_bounds__(256) void din_q_rows_kernel(const float* __restrict__ E, int64_t rows,
                                                        const void* __restrict__ qid, int64_t B, int L,
                                                        float* __restrict__ feat, int64_t ld_feat,
                                                        int32_t* __restrict__ ids_all, int32_t* __restrict__ err) {
  const int64_t b = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
  if (b >= B) return;
  const int c = threadIdx.x & 15;
  const int64_t id = mi_ld_id<IDX>(qid, b);
  const bool ok = id >= 0 && id < rows;
  const float4 q = ok ? *reinterpret_cast<const float4*>(E + id * FD + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(feat + b * ld_feat + FD + c * 4) = q;
  if (c == 0) {
    ids_all[b * (L + 1) + L] = ok ? (int32_t) id : -1;
    if (!ok && err) atomicOr(err, MI_IDERR_RANGE);
  }
}

// One workgroup per sample, three resident per CU (LDS and registers both allow three): while one waits for its
// rows the others run their layers.
template <typename IDX>
__global__ __launch_bounds__(256, 3) void din_lau_fwd_fused_kernel(FusedArgs p) {
  __shared__ __attribute__((aligned(16))) FusedLds S;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int L = p.L;
  // --- the weights' MFMA fragments, straight from global memory into registers, once per workgroup ---
  // fragment (16 outputs n, 32 inputs k): lane l holds row l & 15, k = 8 * (l >> 4) .. + 7
  s8v wb1[2][4], wb2[4];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      wb1[jj][kk] = *reinterpret_cast<const s8v*>(p.W1f + ((2 * w + jj) * 16 + (lane & 15)) * (2 * FD) + kk * 32 + 8 * (lane >> 4));
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    wb2[kk] = *reinterpret_cast<const s8v*>(p.W2 + (w * 16 + (lane & 15)) * FH1 + kk * 32 + 8 * (lane >> 4));
  const int n02 = w * 16 + (lane >> 4) * 4;  // the lane's four outputs of layer 2
  const float4 bb2 = *reinterpret_cast<const float4*>(p.b2 + n02);
  const float4 ww3 = *reinterpret_cast<const float4*>(p.w3 + n02);
  const float b3 = p.w3[FH2];

  // 16 lanes per row (4 floats each), rows g, g + 16, ...
  const int c = t & 15, g = t >> 4;
  const int64_t b = blockIdx.x;
  const int it = 0;
  (void) it;
  const Ids<IDX> ids = load_ids<IDX>(p, b, g);
  const Gathered cur = request_rows(p, ids, b, g, c, t);
  {
    char* const a0 = S.a0;
    DIN_MARK(0);
    // --- the sample's rows [k | q*k] into LDS (and the bookkeeping the backward needs) ---
    const float4 qq = cur.q;
    bool bad = false;
    if (t < FH1) S.rowbias[t] = cur.rowbias;
#pragma unroll
    for (int s = 0; s < FFRAGS; ++s) {
      const int j = g + 16 * s;
      const bool ok = cur.id[s] >= 0;
      bad |= cur.id[s] == -2;
      const float4 kv = ok ? cur.k[s] : make_float4(0.f, 0.f, 0.f, 0.f);
      const int at = img_off(j, c >> 1) + (c & 1) * 8;
      *reinterpret_cast<uint2*>(a0 + at) = pack4(kv.x, kv.y, kv.z, kv.w);
      *reinterpret_cast<uint2*>(S.a1 + at) = pack4(qq.x * kv.x, qq.y * kv.y, qq.z * kv.z, qq.w * kv.w);
      if (c == 0 && j < L) p.ids_all[b * (L + 1) + j] = ok ? cur.id[s] : -1;
    }
    if (p.err && bad && c == 0) atomicOr(p.err, MI_IDERR_RANGE);
    DIN_MARK(1);
    lds_barrier();
    DIN_MARK(2);
    copy_out<2>(a0, S.a1, p.A + b * L * (2 * FD), L);
    DIN_MARK(3);
    // --- layer 1: wave w owns outputs [32 w, 32 w + 32) of all rows ---
    f4v acc1[FFRAGS][2];
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i) {
      acc1[i][0] = f4v{0.f, 0.f, 0.f, 0.f};
      acc1[i][1] = f4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const char* img = kk < 2 ? a0 : S.a1;
        const s8v fa = *reinterpret_cast<const s8v*>(img + img_off(i * 16 + (lane & 15), (kk & 1) * 4 + (lane >> 4)));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc1[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, wb1[jj][kk]),
                                                                __builtin_bit_cast(bf8v, fa), acc1[i][jj], 0, 0, 0);
      }
    DIN_MARK(4);
    lds_barrier();  // a1 has been read by everybody (it is overwritten now); rowbias is there
    // a lane holds outputs n0 .. n0 + 3 of row 16 i + (lane & 15)
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int n0 = (2 * w + jj) * 16 + (lane >> 4) * 4, m = i * 16 + (lane & 15);
        const float4 rb = *reinterpret_cast<const float4*>(&S.rowbias[n0]);
        const f4v v = acc1[i][jj];
        char* img = n0 < 64 ? S.a1 : S.hb;
        *reinterpret_cast<uint2*>(img + img_off(m, (n0 & 63) >> 3) + (n0 & 7) * 2) =
            pack4(fmaxf(v.x + rb.x, 0.f), fmaxf(v.y + rb.y, 0.f), fmaxf(v.z + rb.z, 0.f), fmaxf(v.w + rb.w, 0.f));
      }
    lds_barrier();
    DIN_MARK(5);
    copy_out<2>(S.a1, S.hb, p.H1 + b * L * FH1, L);
    DIN_MARK(6);
    // --- layer 2: wave w owns outputs [16 w, 16 w + 16) ---
    f4v acc2[FFRAGS];
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i) acc2[i] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const char* img = kk < 2 ? S.a1 : S.hb;
        const s8v fa = *reinterpret_cast<const s8v*>(img + img_off(i * 16 + (lane & 15), (kk & 1) * 4 + (lane >> 4)));
        acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, wb2[kk]), __builtin_bit_cast(bf8v, fa),
                                                          acc2[i], 0, 0, 0);
      }
    DIN_MARK(7);
    lds_barrier();  // the first hidden layer has been read (and copied out) by everybody
    // second hidden layer into hb; the head's partial sums - the four columns of a lane, as one 16-lane
    // group's lane of the separate head kernel adds them - into a1 as part[column group 0..15][row]
    {
      float* part = reinterpret_cast<float*>(S.a1);
#pragma unroll
      for (int i = 0; i < FFRAGS; ++i) {
        const int m = i * 16 + (lane & 15);
        const f4v v = acc2[i];
        const uint2 u = pack4(fmaxf(v.x + bb2.x, 0.f), fmaxf(v.y + bb2.y, 0.f), fmaxf(v.z + bb2.z, 0.f), fmaxf(v.w + bb2.w, 0.f));
        *reinterpret_cast<uint2*>(S.hb + img_off(m, n02 >> 3) + (n02 & 7) * 2) = u;
        float s = 0.f;
        s = fmaf(__uint_as_float(u.x << 16), ww3.x, s);
        s = fmaf(__uint_as_float(u.x & 0xFFFF0000u), ww3.y, s);
        s = fmaf(__uint_as_float(u.y << 16), ww3.z, s);
        s = fmaf(__uint_as_float(u.y & 0xFFFF0000u), ww3.w, s);
        part[(n02 >> 2) * FROWS + m] = s;
      }
    }
    lds_barrier();
    DIN_MARK(8);
    copy_out<1>(S.hb, S.hb, p.H2 + b * L * FH2, L);
    // the 16 partial sums of a row added as the separate head kernel's shuffle tree adds them (xor 8, 4, 2, 1)
    if (t < FROWS) {
      const float* part = reinterpret_cast<const float*>(S.a1);
      float s[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = part[i * FROWS + t];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < o; ++i) s[i] += s[i + o];
      const float z = s[0] + b3;
      S.w[t] = z;
      if (t < L) p.a[b * L + t] = z;
    }
    lds_barrier();
    DIN_MARK(9);
    // --- pooling: pooled = sum_j a_j k_j, positions in ascending order (the separate kernel's order), one
    // column a lane.  Masked positions hold k = 0 and a finite a_j, so their term leaves the sum as it is: no
    // test per position (a branch per position makes every LDS read wait for the one before).  The loop is bound
    // by instruction issue of the one wave that runs it: one LDS read, one shift and one fma per position ---
    if (w == 0) {
      float acc = 0.f;
      // byte offset of column `lane` in row r: r * 128 + (((lane >> 3) ^ (r & 7)) << 4) + (lane & 7) * 2
      int base[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) base[r] = (((lane >> 3) ^ r) << 4) + (lane & 7) * 2;
#pragma unroll
      for (int j0 = 0; j0 < FROWS; j0 += 16) {
        unsigned short u[16];
        float ww[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          ww[s] = S.w[j0 + s];
          u[s] = *reinterpret_cast<const unsigned short*>(a0 + (j0 + s) * 128 + base[s & 7]);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
          acc = fmaf(j0 + s < L ? ww[s] : 0.f, __uint_as_float((uint32_t) u[s] << 16), acc);
      }
      p.feat[b * p.ld_feat + lane] = acc;
    }
    DIN_MARK(10);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward of the unit's DATA path, one workgroup per sample: from dL/d[pooled | q] to the gradient of every
// gathered row, through both hidden layers on the matrix cores out of LDS.  What it leaves in HBM is what the
// weight-gradient contractions read (dZ2, dZ1 beside the forward's A and H1) and the per-sample sums the small
// parameters need; dA and the per-position weights' gradient never reach HBM.
//   da_j   = <dpooled, k_j>                                  (0 on masked positions)
//   dZ2    = da_j w3 where H2 > 0                             [L][FH2]
//   dZ1    = (dZ2 W2) where H1 > 0                            [L][FH1]
//   [gk | gp] = dZ1 W1f  (rounded to bf16 as the separate launches store it)
//   d k_j  = a_j dpooled + gk + q * gp;   d q = dq_above + sum_j k_j * gp   (+ S Wq, added by mi_din_qterm_bwd)
//   S = column sums of dZ1,  part = [sum_j da_j H2[j] | sum_j da_j | 0 0 0 | column sums of dZ2]
constexpr int FPART = FH2 + 4 + FH2;
struct FusedBwdArgs {
  int64_t B;
  int L;
  const unsigned short* A;    // [B*L][2*FD]: only the k block is read
  const unsigned short* H1;   // [B*L][FH1]
  const unsigned short* H2;   // [B*L][FH2]
  const float* a;             // [B*L]
  const int32_t* ids_all;     // [B][L+1]
  const float* feat;          // [B][ld_feat] = [pooled | q]
  int64_t ld_feat;
  const float* dfeat;         // [B][ld_dfeat] = dL/d[pooled | q]
  int64_t ld_dfeat;
  const float* w3;            // [FH2] (+ bias)
  const unsigned short* W2T;  // [FH1][FH2] bf16
  const unsigned short* W1fT; // [2*FD][FH1] bf16
  unsigned short* dZ2;        // [B*L][FH2]
  unsigned short* dZ1;        // [B*L][FH1]
  float* dE;                  // [B*(L+1)][FD]
  float* S;                   // [B][FH1]
  float* part;                // [B][FPART]: [sum_j da_j H2[j] | sum_j da_j | 0 0 0 | column sums of dZ2]
};

struct FusedBwdLds {
  char h1[2][IMG];   // H1, then dZ1 in place (two 64-column images); then the d k rows (fp32, [FROWS][FD])
  char z2[IMG];      // dZ2; then gp (bf16, [FROWS][FD])
  float da[FROWS];
  float a[FROWS];
  int live[FROWS];
  float red[4][FH1];  // cross-wave partial sums
  float misc[4];
};

__device__ __forceinline__ float bfl(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfh(uint32_t u) { return __uint_as_float(u & 0xFFFF0000u); }

template <int DUMMY>
__global__ __launch_bounds__(256, 3) void din_lau_bwd_fused_kernel(FusedBwdArgs p) {
  __shared__ __attribute__((aligned(16))) FusedBwdLds S;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int L = p.L;
  const int64_t b = blockIdx.x;
  const int64_t row0 = b * L;
  const int it = 1;  // (trace slot of the backward)
  (void) it;
  DIN_MARK(0);
  // --- everything the sample needs from HBM, requested at once: 16-byte chunks, whole rows contiguous ---
  // k block and H2: chunk c8 = t & 7 (8 columns) of rows (t >> 3) + 32 s;  H1: chunk c16 = t & 15 of rows (t >> 4) + 16 s
  const int c8 = t & 7, r8 = t >> 3, c16 = t & 15, r16 = t >> 4;
  uint4 kreg[4], h2reg[4], h1reg[FFRAGS];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int r = r8 + 32 * s;
    kreg[s] = h2reg[s] = make_uint4(0u, 0u, 0u, 0u);
    if (r < L) {
      kreg[s] = *reinterpret_cast<const uint4*>(p.A + (row0 + r) * (2 * FD) + c8 * 8);
      h2reg[s] = *reinterpret_cast<const uint4*>(p.H2 + (row0 + r) * FH2 + c8 * 8);
    }
  }
#pragma unroll
  for (int s = 0; s < FFRAGS; ++s) {
    const int r = r16 + 16 * s;
    h1reg[s] = r < L ? *reinterpret_cast<const uint4*>(p.H1 + (row0 + r) * FH1 + c16 * 8) : make_uint4(0u, 0u, 0u, 0u);
  }
  if (t < FROWS) {
    S.a[t] = t < L ? p.a[row0 + t] : 0.f;
    S.live[t] = t < L ? (p.ids_all[b * (L + 1) + t] >= 0 ? 1 : 0) : 0;
  }
  // dpooled and w3 for the thread's 8 columns; weights' fragments (16 outputs, 32 inputs: lane l holds row l & 15,
  // inputs 8 (l >> 4) .. + 7).  Layer 2 back: wave w owns outputs [32 w, 32 w + 32) of dH1; layer 1 back: outputs
  // [16 w, 16 w + 16) of gk AND of gp, so that one lane holds both halves of its columns.
  const float4 dp0 = *reinterpret_cast<const float4*>(p.dfeat + b * p.ld_dfeat + c8 * 8);
  const float4 dp1 = *reinterpret_cast<const float4*>(p.dfeat + b * p.ld_dfeat + c8 * 8 + 4);
  const float4 w30 = *reinterpret_cast<const float4*>(p.w3 + c8 * 8);
  const float4 w31 = *reinterpret_cast<const float4*>(p.w3 + c8 * 8 + 4);
  s8v wb2[2][2], wb1[2][4];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      wb2[jj][kk] = *reinterpret_cast<const s8v*>(p.W2T + ((2 * w + jj) * 16 + (lane & 15)) * FH2 + kk * 32 + 8 * (lane >> 4));
  const int d0 = w * 16 + (lane >> 4) * 4;  // the lane's four columns of d k

  DIN_MARK(1);
  // --- da_j = <dpooled, k_j>: four columns at a time, then the tree of the separate kernel (its 16 lanes of 4
  // columns are this kernel's 8 lanes of 2 x 4) ---
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 u = kreg[s];
    float lo = __fmul_rn(dp0.x, bfl(u.x));
    lo = fmaf(dp0.y, bfh(u.x), lo);
    lo = fmaf(dp0.z, bfl(u.y), lo);
    lo = fmaf(dp0.w, bfh(u.y), lo);
    float hi = __fmul_rn(dp1.x, bfl(u.z));
    hi = fmaf(dp1.y, bfh(u.z), hi);
    hi = fmaf(dp1.z, bfl(u.w), hi);
    hi = fmaf(dp1.w, bfh(u.w), hi);
    lo = __fadd_rn(lo, hi);
    lo += __shfl_xor(lo, 1, 64);
    lo += __shfl_xor(lo, 2, 64);
    lo += __shfl_xor(lo, 4, 64);
    const int r = r8 + 32 * s;
    if (c8 == 0 && r < FROWS) S.da[r] = lo;  // masked positions and the rows behind the history hold k = 0: da = 0
  }
  // H1 into LDS as the operand images
#pragma unroll
  for (int s = 0; s < FFRAGS; ++s)
    *reinterpret_cast<uint4*>(S.h1[c16 >> 3] + img_off(r16 + 16 * s, c16 & 7)) = h1reg[s];
  lds_barrier();
  DIN_MARK(2);
  // --- dZ2 = da_j w3 where H2 > 0: by the threads that hold the H2 chunks, straight to HBM and into the operand
  // image; the same pass leaves the thread's share of the sums over the sample's rows ---
  float sw3[8], sb2[8], sda = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sw3[i] = sb2[i] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int r = r8 + 32 * s;
    if (r < FROWS) {  // uniform per wave-instruction group of rows; rows >= L hold zeros
      const float g = S.da[r];
      const uint4 h = h2reg[s];
      const uint32_t hu[4] = {h.x, h.y, h.z, h.w};
      const float wv[8] = {w30.x, w30.y, w30.z, w30.w, w31.x, w31.y, w31.z, w31.w};
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float h0 = bfl(hu[i]), h1v = bfh(hu[i]);
        const unsigned short z0 = h0 > 0.f ? mi_f2bf(g * wv[2 * i]) : (unsigned short) 0;
        const unsigned short z1 = h1v > 0.f ? mi_f2bf(g * wv[2 * i + 1]) : (unsigned short) 0;
        o[i] = (uint32_t) z0 | ((uint32_t) z1 << 16);
        sw3[2 * i] = fmaf(g, h0, sw3[2 * i]);
        sw3[2 * i + 1] = fmaf(g, h1v, sw3[2 * i + 1]);
        sb2[2 * i] += bfl(o[i]);
        sb2[2 * i + 1] += bfh(o[i]);
      }
      if (c8 == 0) sda += g;
      const uint4 z = make_uint4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<uint4*>(S.z2 + img_off(r, c8)) = z;
      if (r < L) *reinterpret_cast<uint4*>(p.dZ2 + (row0 + r) * FH2 + c8 * 8) = z;
    }
  }
  // the sums over the rows: lanes with the same chunk (lane bits 3-5), then the four waves
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      sw3[i] += __shfl_xor(sw3[i], o, 64);
      sb2[i] += __shfl_xor(sb2[i], o, 64);
    }
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) sda += __shfl_xor(sda, o, 64);
  if (lane < 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      S.red[w][lane * 8 + i] = sw3[i];
      S.red[w][64 + lane * 8 + i] = sb2[i];
    }
    if (lane == 0) S.misc[w] = sda;
  }
  DIN_MARK(3);
  lds_barrier();  // dZ2 image and the partial sums are there
  // (the last layer's fragments, requested now that the registers of the H1 / H2 chunks and of the sums are free;
  // needed two phases further on)
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      wb1[jj][kk] = *reinterpret_cast<const s8v*>(p.W1fT + ((w + 4 * jj) * 16 + (lane & 15)) * FH1 + kk * 32 + 8 * (lane >> 4));
  if (t < 128) {
    const float v = (S.red[0][t] + S.red[1][t]) + (S.red[2][t] + S.red[3][t]);
    p.part[b * FPART + (t < 64 ? t : t + 4)] = v;
  } else if (t < 132) {  // the head's bias gradient (sum of da) and the three padding entries behind it
    p.part[b * FPART + FH2 + (t - 128)] = t == 128 ? (S.misc[0] + S.misc[1]) + (S.misc[2] + S.misc[3]) : 0.f;
  }
  DIN_MARK(4);
  // --- dH1 = dZ2 W2, dZ1 = dH1 where H1 > 0, in place over H1 ---
  {
    f4v acc[FFRAGS][2];
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i) acc[i][0] = acc[i][1] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const s8v fa = *reinterpret_cast<const s8v*>(S.z2 + img_off(i * 16 + (lane & 15), kk * 4 + (lane >> 4)));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, wb2[jj][kk]),
                                                               __builtin_bit_cast(bf8v, fa), acc[i][jj], 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int n0 = (2 * w + jj) * 16 + (lane >> 4) * 4, m = i * 16 + (lane & 15);
        char* at = S.h1[n0 >> 6] + img_off(m, (n0 & 63) >> 3) + (n0 & 7) * 2;
        const uint2 h = *reinterpret_cast<const uint2*>(at);
        const f4v v = acc[i][jj];
        uint2 z;
        z.x = (uint32_t) (bfl(h.x) > 0.f ? mi_f2bf(v.x) : (unsigned short) 0) |
              ((uint32_t) (bfh(h.x) > 0.f ? mi_f2bf(v.y) : (unsigned short) 0) << 16);
        z.y = (uint32_t) (bfl(h.y) > 0.f ? mi_f2bf(v.z) : (unsigned short) 0) |
              ((uint32_t) (bfh(h.y) > 0.f ? mi_f2bf(v.w) : (unsigned short) 0) << 16);
        *reinterpret_cast<uint2*>(at) = z;
      }
  }
  DIN_MARK(5);
  lds_barrier();  // dZ1 is complete
  // --- dZ1 to HBM (whole rows contiguous) and its column sums over the sample's rows ---
  {
    float cs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cs[i] = 0.f;
#pragma unroll
    for (int s = 0; s < FFRAGS; ++s) {
      const int r = r16 + 16 * s;
      const uint4 z = *reinterpret_cast<const uint4*>(S.h1[c16 >> 3] + img_off(r, c16 & 7));
      if (r < L) {
        *reinterpret_cast<uint4*>(p.dZ1 + (row0 + r) * FH1 + c16 * 8) = z;
        cs[0] += bfl(z.x); cs[1] += bfh(z.x); cs[2] += bfl(z.y); cs[3] += bfh(z.y);
        cs[4] += bfl(z.z); cs[5] += bfh(z.z); cs[6] += bfl(z.w); cs[7] += bfh(z.w);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      cs[i] += __shfl_xor(cs[i], 16, 64);
      cs[i] += __shfl_xor(cs[i], 32, 64);
    }
    // S.red was last read behind the barrier before last: free again
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) S.red[w][lane * 8 + i] = cs[i];
    }
  }
  DIN_MARK(6);
  // --- [gk | gp] = dZ1 W1f: wave w owns columns [16 w, 16 w + 16) of both halves ---
  // (dpooled and q of the lane's columns are requested here: used behind the products, and eight registers fewer
  // through the phases above)
  const float4 dpd = *reinterpret_cast<const float4*>(p.dfeat + b * p.ld_dfeat + d0);
  const float4 qd = *reinterpret_cast<const float4*>(p.feat + b * p.ld_feat + FD + d0);
  f4v acc[FFRAGS][2];
#pragma unroll
  for (int i = 0; i < FFRAGS; ++i) acc[i][0] = acc[i][1] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < FFRAGS; ++i)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const s8v fa = *reinterpret_cast<const s8v*>(S.h1[kk >> 1] + img_off(i * 16 + (lane & 15), (kk & 1) * 4 + (lane >> 4)));
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, wb1[jj][kk]),
                                                             __builtin_bit_cast(bf8v, fa), acc[i][jj], 0, 0, 0);
    }
  DIN_MARK(7);
  lds_barrier();  // dZ1 (and dZ2) have been read by everybody: their space takes d k and gp now
  if (t < FH1) p.S[b * FH1 + t] = (S.red[0][t] + S.red[1][t]) + (S.red[2][t] + S.red[3][t]);
  // d k = a_j dpooled + gk + q * gp on live positions (gk, gp rounded to bf16 as the separate launches store them):
  // fp32 rows of FD floats, their 16-byte chunks XORed with the row so that a fragment's 16 rows spread over banks
  {
    float* dk = reinterpret_cast<float*>(S.h1[0]);
#pragma unroll
    for (int i = 0; i < FFRAGS; ++i) {
      const int m = i * 16 + (lane & 15);
      const f4v vk = acc[i][0], vp = acc[i][1];
      const uint2 gpu = pack4(vp.x, vp.y, vp.z, vp.w);
      const uint2 gku = pack4(vk.x, vk.y, vk.z, vk.w);
      const float aw = S.a[m];
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (S.live[m]) {
        o.x = fmaf(aw, dpd.x, fmaf(qd.x, bfl(gpu.x), bfl(gku.x)));
        o.y = fmaf(aw, dpd.y, fmaf(qd.y, bfh(gpu.x), bfh(gku.x)));
        o.z = fmaf(aw, dpd.z, fmaf(qd.z, bfl(gpu.y), bfl(gku.y)));
        o.w = fmaf(aw, dpd.w, fmaf(qd.w, bfh(gpu.y), bfh(gku.y)));
      }
      *reinterpret_cast<float4*>(dk + m * FD + (((d0 >> 2) ^ (m & 15)) << 2)) = o;
      *reinterpret_cast<uint2*>(S.z2 + img_off(m, d0 >> 3) + (d0 & 7) * 2) = gpu;
    }
  }
  lds_barrier();
  DIN_MARK(8);
  // --- d k rows to HBM; d q = dq_above + sum_j k_j * gp over the threads that still hold the k chunks ---
#pragma unroll
  for (int s = 0; s < FFRAGS; ++s) {
    const int r = r16 + 16 * s;
    if (r < L) {
      const float* dk = reinterpret_cast<const float*>(S.h1[0]);
      *reinterpret_cast<float4*>(p.dE + (b * (L + 1) + r) * FD + c16 * 4) =
          *reinterpret_cast<const float4*>(dk + r * FD + ((c16 ^ (r & 15)) << 2));
    }
  }
  {
    float sq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sq[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = r8 + 32 * s;
      if (r < FROWS) {
        const uint4 g = *reinterpret_cast<const uint4*>(S.z2 + img_off(r, c8));
        const uint4 k = kreg[s];  // zero on masked positions and behind the history
        sq[0] = fmaf(bfl(k.x), bfl(g.x), sq[0]); sq[1] = fmaf(bfh(k.x), bfh(g.x), sq[1]);
        sq[2] = fmaf(bfl(k.y), bfl(g.y), sq[2]); sq[3] = fmaf(bfh(k.y), bfh(g.y), sq[3]);
        sq[4] = fmaf(bfl(k.z), bfl(g.z), sq[4]); sq[5] = fmaf(bfh(k.z), bfh(g.z), sq[5]);
        sq[6] = fmaf(bfl(k.w), bfl(g.w), sq[6]); sq[7] = fmaf(bfh(k.w), bfh(g.w), sq[7]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) sq[i] += __shfl_xor(sq[i], o, 64);
    if (lane < 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) S.red[w][lane * 8 + i] = sq[i];
    }
  }
  lds_barrier();
  DIN_MARK(9);
  if (t < FD)
    p.dE[(b * (L + 1) + L) * FD + t] =
        p.dfeat[b * p.ld_dfeat + FD + t] + ((S.red[0][t] + S.red[1][t]) + (S.red[2][t] + S.red[3][t]));
}

// out[s][c] = sum of in[r][c] over the rows of slice s (rows in ascending order): one thread per column
__global__ __launch_bounds__(256) void rows_sum_f32_kernel(const float* __restrict__ in, int64_t R, int n, int rps,
                                                          float* __restrict__ out) {
  const int c = threadIdx.x;
  if (c >= n) return;
  const int64_t lo = (int64_t) blockIdx.x * rps;
  const int64_t hi = lo + rps < R ? lo + rps : R;
  float acc = 0.f;
  int64_t r = lo;
  for (; r + 8 <= hi; r += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = in[(r + j) * n + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; r < hi; ++r) acc += in[r * n + c];
  out[(int64_t) blockIdx.x * n + c] = acc;
}

}  // namespace

extern "C" int mi_din_lau_fwd_fused_ok(int dim, int h1, int h2, int L, int factored) {
  return factored && dim == FD && h1 == FH1 && h2 == FH2 && L >= 1 && L <= FROWS;
}

extern "C" int mi_din_lau_fwd_fused(const float* E, int64_t rows, const void* hist, const void* qid, int idx_bits,
                                    const int32_t* len, int64_t B, int L, const void* W1f, const float* Wq,
                                    const float* b1, const void* W2, const float* b2, const float* w3, void* A, void* H1,
                                    void* H2, float* a, int32_t* ids_all, float* feat, int64_t ld_feat, float* rowbias,
                                    int32_t* err, void* stream) {
  const char* who = "mi_din_lau_fwd_fused";
  MI_REQUIRE(E && hist && qid && len && W1f && Wq && b1 && W2 && b2 && w3 && A && H1 && H2 && a && ids_all && feat && rowbias,
             "%s: null pointer", who);
  MI_REQUIRE(B >= 1 && B < ((int64_t) 1 << 31) && L >= 1 && L <= FROWS && rows >= 1 && rows < ((int64_t) 1 << 31),
             "%s: bad B / L / rows (L <= %d)", who, FROWS);
  MI_REQUIRE(idx_bits == 32 || idx_bits == 64, "%s: idx_bits must be 32 or 64", who);
  MI_REQUIRE(ld_feat >= 2 * FD && ld_feat % 4 == 0, "%s: ld_feat", who);
  auto a16 = [](const void* p) { return ((uintptr_t) p & 15) == 0; };
  MI_REQUIRE(a16(E) && a16(W1f) && a16(W2) && a16(b2) && a16(w3) && a16(A) && a16(H1) && a16(H2) && a16(feat) && a16(rowbias),
             "%s: misaligned buffer", who);
  hipStream_t s = (hipStream_t) stream;
  // the candidates' rows, then the first layer's term in them (one bias row per sample), then the unit itself
  const unsigned qgrid = (unsigned) mi_cdiv(B, 16);
  if (idx_bits == 32)
    hipLaunchKernelGGL(din_q_rows_kernel<int32_t>, dim3(qgrid), dim3(256), 0, s, E, rows, qid, B, L, feat, ld_feat, ids_all, err);
  else
    hipLaunchKernelGGL(din_q_rows_kernel<int64_t>, dim3(qgrid), dim3(256), 0, s, E, rows, qid, B, L, feat, ld_feat, ids_all, err);
  MI_LAUNCH_CHECK();
  {
    const int rc = mi_din_qterm_fwd(feat + FD, ld_feat, Wq, b1, B, FH1, FD, rowbias, stream);
    if (rc != MI_OK) return rc;
  }
  FusedArgs p{E, rows, hist, qid, len, B, L, (const unsigned short*) W1f, rowbias, (const unsigned short*) W2, b2, w3,
              (unsigned short*) A, (unsigned short*) H1, (unsigned short*) H2, a, ids_all, feat, ld_feat, err};
  const unsigned grid = (unsigned) B;
  if (idx_bits == 32) hipLaunchKernelGGL(din_lau_fwd_fused_kernel<int32_t>, dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(din_lau_fwd_fused_kernel<int64_t>, dim3(grid), dim3(256), 0, s, p);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_din_lau_bwd_part_floats(void) { return FPART; }

extern "C" int mi_din_lau_bwd_fused(int64_t B, int L, const void* A, const void* H1, const void* H2, const float* a,
                                    const int32_t* ids_all, const float* feat, int64_t ld_feat, const float* dfeat,
                                    int64_t ld_dfeat, const float* w3, const void* W2T, const void* W1fT, void* dZ2,
                                    void* dZ1, float* dE, float* S, float* part, int slices, float* part_sums,
                                    void* stream) {
  const char* who = "mi_din_lau_bwd_fused";
  MI_REQUIRE(A && H1 && H2 && a && ids_all && feat && dfeat && w3 && W2T && W1fT && dZ2 && dZ1 && dE && S && part && part_sums,
             "%s: null pointer", who);
  MI_REQUIRE(B >= 1 && B < ((int64_t) 1 << 31) && L >= 1 && L <= FROWS && slices >= 1 && slices <= B,
             "%s: bad B / L / slices (L <= %d)", who, FROWS);
  MI_REQUIRE(ld_feat >= 2 * FD && ld_feat % 4 == 0 && ld_dfeat >= 2 * FD && ld_dfeat % 4 == 0, "%s: leading dimensions", who);
  auto a16 = [](const void* p) { return ((uintptr_t) p & 15) == 0; };
  MI_REQUIRE(a16(A) && a16(H1) && a16(H2) && a16(feat) && a16(dfeat) && a16(w3) && a16(W2T) && a16(W1fT) && a16(dZ2) &&
                 a16(dZ1) && a16(dE),
             "%s: misaligned buffer", who);
  FusedBwdArgs p{B, L, (const unsigned short*) A, (const unsigned short*) H1, (const unsigned short*) H2, a, ids_all, feat,
                 ld_feat, dfeat, ld_dfeat, w3, (const unsigned short*) W2T, (const unsigned short*) W1fT,
                 (unsigned short*) dZ2, (unsigned short*) dZ1, dE, S, part};
  hipStream_t s = (hipStream_t) stream;
  hipLaunchKernelGGL(din_lau_bwd_fused_kernel<0>, dim3((unsigned) B), dim3(256), 0, s, p);
  MI_LAUNCH_CHECK();
  // the per-sample sums of the small parameters' gradients down to `slices` rows (samples in ascending order)
  const int rps = (int) mi_cdiv(B, slices);
  MI_REQUIRE(mi_cdiv(B, rps) == slices, "%s: %d slices do not divide %lld samples into equal runs", who, slices, (long long) B);
  hipLaunchKernelGGL(rows_sum_f32_kernel, dim3((unsigned) slices), dim3(256), 0, s, part, B, FPART, rps, part_sums);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

#ifdef DIN_TRACE
extern "C" int mi_debug_din_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(din_trace_buf), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? 0 : -1;
}
#endif
