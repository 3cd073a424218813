// bf16 MFMA contraction for the dense parts of the feature-interaction models (DCN-v2 cross
// layers and MLP, DIN activation unit).  The reference's dense layers are tf.layers.dense in
// fp32 on a 5,910-parameter network (recommendation/code/r_model.py:53-57); the contractions
// here are the ones the larger interaction models add, and the only work in this library that
// is put on the matrix cores.
//
//   C[m][n] = sum_k A(m,k) * B(n,k)        fp32 accumulate, bf16 operands
//
// Either operand may be stored "reduction-contiguous" (row m holds its k values back to back:
// activations X[M][K], weights W[N][K]) or "reduction-major" (row k holds all m: what the
// backward contractions need - dX = dY*W sums over W's ROW index, dW = dY^T*X sums over the
// batch).  Reduction-major tiles are staged as they lie in memory and read from LDS with the
// transposing read (ds_read_b64_tr_b16), so no transposed copy of any activation or weight is
// ever written to HBM.
//
// A wave owns FM x FN fragments of mfma_f32_16x16x32_bf16 (128x64 or 64x64 results); a workgroup
// is WM x WN waves (256x256x64 tiles with 8 waves, or 128x128x64 with 4).  Tiles are staged
// global -> LDS by DMA one tile ahead, retired with a counted vmcnt wait and ONE raw barrier per
// K tile; inside a tile the fragment reads run three MFMA rows ahead of the MFMAs that use them.
#include "common.h"
#include <type_traits>

namespace {

typedef short s8v __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4v* lds_s4v_ptr;

constexpr int BK = 64;
constexpr int SUB_BYTES = 128 * BK * 2;  // an operand tile is made of 128-row sub-tiles of 16 KiB

// LDS image of a reduction-contiguous tile: [128 rows][64 k] = 128-byte rows of eight 16-byte
// chunks; chunk position XORed with the row so that the 16 rows of a fragment spread over banks.
__device__ __forceinline__ int row_img(int r, int ch) { return r * 128 + ((ch ^ (r & 7)) << 4); }
// LDS image of a reduction-major tile: [64 k][128 rows] = 256-byte rows of sixteen chunks, with
// the XOR under which both the transposing reads of a 16x16x32 operand are conflict-free.
__device__ __forceinline__ int tr_img(int kr, int ch) {
  return kr * 256 + ((ch ^ (((kr & 3) << 2) | ((kr >> 2) & 3))) << 4);
}

// Staging: global -> LDS directly (LDS-DMA, global_load_lds_dwordx4), no registers and no
// ds_write in between.  One wave-instruction deposits 64 x 16 bytes at a wave-uniform LDS base
// + lane*16, i.e. it fills LDS linearly; the XOR of the images above is therefore applied to
// the SOURCE address (lane -> which chunk of which row it fetches).  A 16 KiB sub-tile is 16
// such 1 KiB pieces; the tile's pieces are dealt round-robin over the workgroup's waves.
// rows [row0, row0 + 128*NSUB) x k [k0, k0+64) of an operand with R rows in total
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// The per-lane source pointers of a wave's pieces at k = 0 are computed once (src_init); staging a
// K tile then costs one 64-bit add, one M0 write and one DMA instruction per piece.
template <bool TR, int NSUB, int NWAVES>
struct Stager {
  static constexpr int PER_WAVE = NSUB * 16 / NWAVES;
  const unsigned short* src[PER_WAVE];
  int64_t kstride;  // elements per unit of k
  __device__ __forceinline__ void init(const unsigned short* __restrict__ P, int64_t ld, int64_t row0, int64_t R,
                                       int wave, int lane) {
    kstride = TR ? ld : 1;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int gp = wave + NWAVES * i;  // piece of the whole tile
      const int sub = gp >> 4, piece = gp & 15;
      if constexpr (!TR) {
        const int r = piece * 8 + (lane >> 3), ch = (lane & 7) ^ (r & 7);
        int64_t rr = row0 + sub * 128 + r;
        rr = rr < R ? rr : R - 1;  // out-of-range rows read a valid row; their results are never stored
        src[i] = P + rr * ld + ch * 8;
      } else {
        const int kr = piece * 4 + (lane >> 4), ch = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
        int64_t m = row0 + sub * 128 + ch * 8;
        m = m <= R - 8 ? m : R - 8;
        src[i] = P + kr * ld + m;
      }
    }
  }
  __device__ __forceinline__ void stage_one(int i, int64_t k0, char* lds, int wave) const {
    const int gp = __builtin_amdgcn_readfirstlane(wave + NWAVES * i);
    __builtin_amdgcn_global_load_lds((gptr_t) (src[i] + k0 * kstride), (lptr_t) (lds + gp * 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void stage(int64_t k0, char* lds, int wave) const {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int gp = __builtin_amdgcn_readfirstlane(wave + NWAVES * i);
      __builtin_amdgcn_global_load_lds((gptr_t) (src[i] + k0 * kstride), (lptr_t) (lds + gp * 1024), 16, 0, 0);
    }
  }
};

// fragment f (16 rows of the tile), k-step kk (32 of the tile's 64 k): lane l gets row l&15,
// k = 8*(l>>4) + 0..7
template <bool TR>
__device__ __forceinline__ s8v read_frag(const char* tile, int frag, int kk, int lane) {
  const char* lds = tile + (frag >> 3) * SUB_BYTES;
  const int f = frag & 7;
  if constexpr (!TR) {
    const int r = f * 16 + (lane & 15), ch = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const s8v*>(lds + row_img(r, ch));
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int c0 = f * 2 + (p >> 1), r0 = kk * 32 + 8 * g + q;
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr) (lds + tr_img(r0, c0) + 8 * (p & 1)));
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v_ptr) (lds + tr_img(r0 + 4, c0) + 8 * (p & 1)));
    return s8v{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  }
}

struct GemmArgs {
  const unsigned short* A;
  const unsigned short* B;
  int64_t lda, ldb, M, N, K, k_per_split;
  int tiles_n;
  mi_gemm_epilogue e;
};

__device__ __forceinline__ float4 ld_bf4(const void* p, int64_t ld, int64_t m, int64_t n) {
  const ushort4 u = *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(p) + m * ld + n);
  return make_float4(mi_bf2f(u.x), mi_bf2f(u.y), mi_bf2f(u.z), mi_bf2f(u.w));
}
__device__ __forceinline__ void st_bf4(void* p, int64_t ld, int64_t m, int64_t n, float4 v) {
  ushort4 u;
  u.x = mi_f2bf(v.x); u.y = mi_f2bf(v.y); u.z = mi_f2bf(v.z); u.w = mi_f2bf(v.w);
  *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(p) + m * ld + n) = u;
}

// The epilogue of C[m][n..n+3] (n % 4 == 0, all four in range) in two steps, so that the global
// loads it needs can be issued for many positions at once, ahead of the arithmetic:
// epi_load fetches the bf16 inputs of a position, epi_apply consumes them with the accumulator.
struct EpiIn {
  ushort4 p0, p1;
};
__device__ __forceinline__ float4 bf4(ushort4 u) {
  return make_float4(mi_bf2f(u.x), mi_bf2f(u.y), mi_bf2f(u.z), mi_bf2f(u.w));
}
__device__ __forceinline__ ushort4 ld_u4(const void* p, int64_t ld, int64_t m, int64_t n) {
  return *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(p) + m * ld + n);
}

__device__ __forceinline__ EpiIn epi_load(const mi_gemm_epilogue& e, int64_t m, int64_t n) {
  EpiIn r;
  r.p0 = r.p1 = make_ushort4(0, 0, 0, 0);
  switch (e.mode) {
    case MI_EPI_CROSS_FWD:
      r.p0 = ld_u4(e.in0, e.ld_in0, m, n);
      r.p1 = ld_u4(e.in1, e.ld_in1, m, n);
      break;
    case MI_EPI_MASK:
      if (e.in0) r.p0 = ld_u4(e.in0, e.ld_in0, m, n);
      break;
    case MI_EPI_CROSS_BWD:
      if (e.out1) r.p0 = ld_u4(e.in0, e.ld_in0, m, n);
      if (e.in1) r.p1 = ld_u4(e.in1, e.ld_in1, m, n);
      break;
    default:
      break;
  }
  return r;
}

__device__ __forceinline__ void epi_apply(const mi_gemm_epilogue& e, int64_t M, int z, int64_t m, int64_t n, float4 v,
                                          const EpiIn& in, float4 bias) {
  switch (e.mode) {
    case MI_EPI_SLAB: {  // fp32 partial of K-slice z
      float* o = reinterpret_cast<float*>(e.out) + ((int64_t) z * M + m) * e.ld_out + n;
      *reinterpret_cast<float4*>(o) = v;
      break;
    }
    case MI_EPI_BIAS_ACT: {  // out = act(acc + bias)
      if (e.bias_rows > 0) bias = *reinterpret_cast<const float4*>(e.bias + (m / e.bias_rows) * e.bias_ld + n);
      v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
      if (e.flags & MI_EPI_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      st_bf4(e.out, e.ld_out, m, n, v);
      break;
    }
    case MI_EPI_CROSS_FWD: {  // u = acc + bias -> out1;  out = in0 (x0) * u + in1 (x_l)
      v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
      if (e.out1) st_bf4(e.out1, e.ld_out1, m, n, v);
      const float4 x0 = bf4(in.p0), xl = bf4(in.p1);
      st_bf4(e.out, e.ld_out, m, n,
             make_float4(fmaf(x0.x, v.x, xl.x), fmaf(x0.y, v.y, xl.y), fmaf(x0.z, v.z, xl.z), fmaf(x0.w, v.w, xl.w)));
      break;
    }
    case MI_EPI_MASK: {  // out = acc where in0 > 0 else 0 (ReLU backward of the layer that produced in0)
      if (e.in0) {
        const float4 h = bf4(in.p0);
        v.x = h.x > 0.f ? v.x : 0.f; v.y = h.y > 0.f ? v.y : 0.f;
        v.z = h.z > 0.f ? v.z : 0.f; v.w = h.w > 0.f ? v.w : 0.f;
      }
      st_bf4(e.out, e.ld_out, m, n, v);
      break;
    }
    case MI_EPI_CROSS_BWD: {  // v = acc + in1 (dx_{l+1}, optional) is dx_l -> out;  out1 = v * in0 (x0) is du_{l-1}
      if (e.in1) {
        const float4 r = bf4(in.p1);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      st_bf4(e.out, e.ld_out, m, n, v);
      if (e.out1) {
        const float4 x0 = bf4(in.p0);
        st_bf4(e.out1, e.ld_out1, m, n, make_float4(v.x * x0.x, v.y * x0.y, v.z * x0.z, v.w * x0.w));
      }
      break;
    }
    default:
      break;
  }
}

// The same with the mode and its optional tensors known at compile time (F0, F1 per mode:
// BIAS_ACT: F0 = ReLU; CROSS_FWD: F0 = out1 wanted; MASK: F0 = in0 given; CROSS_BWD: F0 = out1 (and
// in0), F1 = in1 given).  The run-time switch above puts a branch around every element, and at
// every join the compiler has to wait for ALL outstanding memory operations, the previous rows'
// stores included: measured, an input tile cost 17 us per 256 x 256 tile that way.  Without
// branches the waits are counted and loads run ahead of stores.
template <int MODE, bool F0, bool F1>
struct EpiInT {
  ushort4 p0, p1;
};
template <int MODE, bool F0, bool F1>
__device__ __forceinline__ void epi_load_t(const mi_gemm_epilogue& e, int64_t m, int64_t n, ushort4& p0, ushort4& p1,
                                           float4& pb) {
  if constexpr (MODE == MI_EPI_BIAS_ACT) {
    // F1: the bias is a matrix with one row per bias_rows rows of C (m < 2^32 on this path: see the host check)
    if constexpr (F1) pb = *reinterpret_cast<const float4*>(e.bias + (int64_t) ((uint32_t) m / (uint32_t) e.bias_rows) * e.bias_ld + n);
  } else if constexpr (MODE == MI_EPI_CROSS_FWD) {
    p0 = ld_u4(e.in0, e.ld_in0, m, n);
    p1 = ld_u4(e.in1, e.ld_in1, m, n);
  } else if constexpr (MODE == MI_EPI_MASK) {
    if constexpr (F0) p0 = ld_u4(e.in0, e.ld_in0, m, n);
  } else if constexpr (MODE == MI_EPI_CROSS_BWD) {
    if constexpr (F0) p0 = ld_u4(e.in0, e.ld_in0, m, n);
    if constexpr (F1) p1 = ld_u4(e.in1, e.ld_in1, m, n);
  }
}
template <int MODE, bool F0, bool F1>
__device__ __forceinline__ void epi_apply_t(const mi_gemm_epilogue& e, int64_t M, int z, int64_t m, int64_t n, float4 v,
                                            ushort4 p0, ushort4 p1, float4 pb, float4 bias) {
  if constexpr (MODE == MI_EPI_SLAB) {
    float* o = reinterpret_cast<float*>(e.out) + ((int64_t) z * M + m) * e.ld_out + n;
    *reinterpret_cast<float4*>(o) = v;
  } else if constexpr (MODE == MI_EPI_BIAS_ACT) {
    if constexpr (F1) bias = pb;
    v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
    if constexpr (F0) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    st_bf4(e.out, e.ld_out, m, n, v);
  } else if constexpr (MODE == MI_EPI_CROSS_FWD) {
    v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
    if constexpr (F0) st_bf4(e.out1, e.ld_out1, m, n, v);
    const float4 x0 = bf4(p0), xl = bf4(p1);
    st_bf4(e.out, e.ld_out, m, n,
           make_float4(fmaf(x0.x, v.x, xl.x), fmaf(x0.y, v.y, xl.y), fmaf(x0.z, v.z, xl.z), fmaf(x0.w, v.w, xl.w)));
  } else if constexpr (MODE == MI_EPI_MASK) {
    if constexpr (F0) {
      const float4 h = bf4(p0);
      v.x = h.x > 0.f ? v.x : 0.f; v.y = h.y > 0.f ? v.y : 0.f;
      v.z = h.z > 0.f ? v.z : 0.f; v.w = h.w > 0.f ? v.w : 0.f;
    }
    st_bf4(e.out, e.ld_out, m, n, v);
  } else {  // MI_EPI_CROSS_BWD
    if constexpr (F1) {
      const float4 r = bf4(p1);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    st_bf4(e.out, e.ld_out, m, n, v);
    if constexpr (F0) {
      const float4 x0 = bf4(p0);
      st_bf4(e.out1, e.ld_out1, m, n, make_float4(v.x * x0.x, v.y * x0.y, v.z * x0.z, v.w * x0.w));
    }
  }
}

// Epilogue of a wave's FM x FN fragments for a tile that lies wholly inside the matrix (no bounds
// checks), mode fixed at compile time.  Fragment rows go two at a time and the input loads of the
// next pair are issued before the stores of this one.
template <int MODE, bool F0, bool F1, bool NGUARD, int FM, int FN, class Acc>
__device__ __forceinline__ void epilogue_interior(const GemmArgs& a, Acc& acc, char* scratch, int ep_row, int64_t m_base,
                                                  int64_t n, bool n_ok, int z, int lane, float4 bias) {
  // two input tensors: one fragment row at a time, or the accumulators and two pairs' inputs do not fit the registers
  constexpr bool TWO_IN = MODE == MI_EPI_CROSS_FWD || (MODE == MI_EPI_CROSS_BWD && F0 && F1);
  constexpr int CH = TWO_IN ? 1 : 2, NCH = FM / CH;
  ushort4 p0[NCH][CH][4], p1[NCH][CH][4];
  float4 pb[NCH][CH][4];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int ii = 0; ii < CH; ++ii)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (!NGUARD || n_ok)  // NGUARD: the tile hangs over the last column; lanes beyond it only help with the transposes
          epi_load_t<MODE, F0, F1>(a.e, m_base + (c * CH + ii) * 16 + q * 4 + (lane >> 4), n, p0[c][ii][q], p1[c][ii][q],
                                   pb[c][ii][q]);
  };
  load_chunk(0);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) load_chunk(c + 1);
#pragma unroll
    for (int ii = 0; ii < CH; ++ii) {
      const int i = c * CH + ii;
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<f4v*>(scratch + (lane & 15) * ep_row + (j * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
      asm volatile("" ::: "memory");  // ... and this row's reads behind its writes
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = q * 4 + (lane >> 4);
        const f4v v = *reinterpret_cast<const f4v*>(scratch + row * ep_row + (lane & 15) * 16);
        if (!NGUARD || n_ok)
          epi_apply_t<MODE, F0, F1>(a.e, a.M, z, m_base + i * 16 + row, n, make_float4(v.x, v.y, v.z, v.w), p0[c][ii][q],
                                    p1[c][ii][q], pb[c][ii][q], bias);
      }
      asm volatile("" ::: "memory");  // the next fragment row's LDS writes stay behind this row's reads
    }
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool A_TR, bool B_TR, int WM, int WN, int FM, int FN, int STAGES>
__global__ __launch_bounds__(WM* WN * 64, WM * WN == 4 ? 2 : 1) void gemm_bf16_kernel(GemmArgs a) {
  // a wave owns FM x FN fragments of 16x16; the workgroup is WM x WN waves
  constexpr int NWAVES = WM * WN, BM = 16 * FM * WM, BN = 16 * FN * WN;
  constexpr int A_SUB = BM / 128, B_SUB = BN / 128;
  constexpr int A_BYTES = A_SUB * SUB_BYTES, B_BYTES = B_SUB * SUB_BYTES;
  constexpr int PER_TILE = (A_SUB + B_SUB) * 16 / NWAVES;  // DMA instructions per wave and K tile
  constexpr int D = STAGES - 1;                             // tiles in flight ahead of the one being multiplied
  constexpr bool INTERLEAVE = NWAVES == 8;                  // spread the DMA over the MFMA rows (one workgroup per CU)
  static_assert(BM % 128 == 0 && BN % 128 == 0 && (A_SUB * 16) % NWAVES == 0 && (B_SUB * 16) % NWAVES == 0, "tile");
  static_assert(D >= 1 && D <= 3, "stages");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const As = smem;                     // STAGES buffers of A_BYTES
  char* const Bs = smem + STAGES * A_BYTES;  // STAGES buffers of B_BYTES
  // consecutive workgroup ids go round-robin over the 8 XCDs; renumber so that each XCD (its own
  // L2) works on consecutive tiles, i.e. the n-tiles that share one A row block
  const int nb = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nb >> 3, r = nb & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  const int tn = lid % a.tiles_n, tm = lid / a.tiles_n;
  const int64_t m0 = (int64_t) tm * BM, n0 = (int64_t) tn * BN;
  const int z = blockIdx.y;
  const int64_t kbeg = (int64_t) z * a.k_per_split;
  const int64_t kend = kbeg + a.k_per_split < a.K ? kbeg + a.k_per_split : a.K;
  const int ktiles = (int) ((kend - kbeg) / BK);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w / WN, wc = w % WN;
  // the epilogue's bias is requested now: a load issued at the end would be one more round trip per tile
  const int64_t n = n0 + wc * FN * 16 + (lane & 15) * 4;
  const bool n_ok = n < a.N;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n_ok && a.e.bias && ((a.e.mode == MI_EPI_BIAS_ACT && a.e.bias_rows <= 0) || a.e.mode == MI_EPI_CROSS_FWD))
    bias = *reinterpret_cast<const float4*>(a.e.bias + n);

  f4v acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

  Stager<A_TR, A_SUB, NWAVES> sta;
  Stager<B_TR, B_SUB, NWAVES> stb;
  sta.init(a.A, a.lda, m0, a.M, w, lane);
  stb.init(a.B, a.ldb, n0, a.N, w, lane);
#define STAGE(kt_, buf_)                                             \
  do {                                                               \
    const int64_t k0_ = kbeg + (int64_t) (kt_) * BK;                 \
    sta.stage(k0_, As + (buf_) * A_BYTES, w);                        \
    stb.stage(k0_, Bs + (buf_) * B_BYTES, w);                        \
  } while (0)

  if constexpr (NWAVES == 8) {  // one workgroup per CU: nothing else covers a bubble at the tile boundary
  // The K loop as ONE stream of MFMA rows (a row = one A fragment against the FN resident B
  // fragments) that runs across tile boundaries.  An A fragment is read RING rows before the MFMAs
  // that use it; the barrier that hands over the next tile sits RING rows before the end of a tile,
  // right after the last LDS read of the tile has been issued, so that the rows left over
  // (their operands are in registers) cover the first reads of the next tile: no MFMA ever waits for
  // a read that was issued after a barrier.  The buffer the barrier frees is restaged (tile kt+2) one
  // DMA piece per row in the MFMAs' shadow, over the rest of this tile and the start of the next.
  static_assert(STAGES == 2, "two staging buffers");
  constexpr int A_PER = A_SUB * 16 / NWAVES, B_PER = B_SUB * 16 / NWAVES;
  constexpr int ROWS = 2 * FM;             // MFMA rows of a K tile: k-step 0, then k-step 1
  constexpr int RING = 4;                  // A fragments in flight ahead of the MFMAs
  constexpr int RB = ROWS - RING;          // the row whose end holds the barrier
  constexpr int TAIL = ROWS - 1 - RB;      // rows of a tile behind its barrier
  constexpr int PPR = (PER_TILE + ROWS / 2 - 1) / (ROWS / 2);  // DMA pieces per row
  constexpr int TAILP = TAIL * PPR < PER_TILE ? TAIL * PPR : PER_TILE;  // pieces of a tile issued in the tail rows
  constexpr int HEAD_ROWS = (PER_TILE - TAILP + PPR - 1) / PPR;  // rows at the start of a tile that finish the staging of the next
  static_assert(FM >= FN && ROWS % RING == 0 && HEAD_ROWS <= RB, "wave tile");
  auto stage_piece = [&](int p, int64_t k0, int buf) {
    if (p < A_PER) sta.stage_one(p, k0, As + buf * A_BYTES, w);
    else stb.stage_one(p - A_PER, k0, Bs + buf * B_BYTES, w);
  };
  s8v fb[2][FN], fa[RING];
  if (ktiles > 0) {
    STAGE(0, 0);
    if (ktiles > 1) {
#pragma unroll
      for (int p = 0; p < TAILP; ++p) stage_piece(p, kbeg + BK, 1);
      wait_vmcnt<TAILP>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[0][j] = read_frag<B_TR>(Bs, wc * FN + j, 0, lane);
#pragma unroll
    for (int q = 0; q < RING; ++q) fa[q] = read_frag<A_TR>(As, wr * FM + q % FM, q / FM, lane);
  }
  int cur = 0;  // buffer of tile kt
  // One K tile.  MORE1 / MORE2: tiles kt+1 / kt+2 exist - compile-time, so that the body is straight-line
  // code (with a branch around each DMA piece the compiler drains ALL outstanding LDS reads at every
  // join); the loop runs the steady state and the last two tiles are peeled.
  auto tile = [&](int kt, auto more1_c, auto more2_c) {
    constexpr bool MORE1 = decltype(more1_c)::value, MORE2 = decltype(more2_c)::value;
    const int64_t k1 = kbeg + (int64_t) (kt + 1) * BK, k2 = k1 + BK;
    const char* Ac = As + cur * A_BYTES;
    const char* Bc = Bs + cur * B_BYTES;
    const char* An = As + (cur ^ 1) * A_BYTES;
    const char* Bn = Bs + (cur ^ 1) * B_BYTES;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int kk = r / FM, i = r % FM;
      // the rest of tile kt+1 (its first pieces went out behind the previous barrier) ...
      if (MORE1 && r <= RB) {
#pragma unroll
        for (int q = 0; q < PPR; ++q)
          if (TAILP + r * PPR + q < PER_TILE) stage_piece(TAILP + r * PPR + q, k1, cur ^ 1);
      }
      // ... and the first pieces of tile kt+2 into the buffer this tile's barrier has freed
      if (MORE2 && r > RB) {
#pragma unroll
        for (int q = 0; q < PPR; ++q)
          if ((r - RB - 1) * PPR + q < TAILP) stage_piece((r - RB - 1) * PPR + q, k2, cur);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j)
        // first operand indexes the result's rows (kept in registers): n; second its columns (lanes): m
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, fb[kk][j]),
                                                            __builtin_bit_cast(bf8v, fa[r % RING]), acc[i][j], 0, 0, 0);
      if (MORE1 && r == RB) {
        // every LDS read of this tile has been issued: once they are back the buffer may be
        // overwritten; the wave's own DMA of tile kt+1 has landed at vmcnt(0), everybody's at the barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[0][j] = read_frag<B_TR>(Bn, wc * FN + j, 0, lane);
      }
      if (r + RING < ROWS) fa[r % RING] = read_frag<A_TR>(Ac, wr * FM + (r + RING) % FM, (r + RING) / FM, lane);
      else if (MORE1) fa[r % RING] = read_frag<A_TR>(An, wr * FM + (r + RING - ROWS) % FM, (r + RING - ROWS) / FM, lane);
      // the B fragments of k-step 1: their registers are free once the previous tile's last row is done
      if (r < FN) fb[1][r] = read_frag<B_TR>(Bc, wc * FN + r, 1, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    cur ^= 1;
  };
  {
    int kt = 0;
    for (; kt + 2 < ktiles; ++kt) tile(kt, std::true_type{}, std::true_type{});
    if (kt + 1 < ktiles) tile(kt++, std::true_type{}, std::false_type{});
    if (kt < ktiles) tile(kt, std::false_type{}, std::false_type{});
  }
  } else {  // two workgroups per CU cover each other; the tile-by-tile loop needs fewer registers
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < ktiles) STAGE(s, s);
  int cur = 0;  // buffer of tile kt
  for (int kt = 0; kt < ktiles; ++kt) {
    // Tile kt must have landed; the younger tiles already issued (at most D-1 of them) may stay in
    // flight.  Each wave waits for its own DMA, the barrier then covers everybody's - and also
    // tells that every wave is done reading the buffer that is restaged right below.
    const int younger = (kt + D - 1 < ktiles ? kt + D - 1 : ktiles - 1) - kt;
    if (D >= 3 && younger == 2) wait_vmcnt<2 * PER_TILE>();
    else if (D >= 2 && younger == 1) wait_vmcnt<PER_TILE>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool more = kt + D < ktiles;
    int nxt = cur + D;
    nxt = nxt >= STAGES ? nxt - STAGES : nxt;
    const int64_t knext = kbeg + (int64_t) (kt + D) * BK;
    constexpr int A_PER = A_SUB * 16 / NWAVES, B_PER = B_SUB * 16 / NWAVES;
    constexpr int ROWS = 2 * FM;                                   // MFMA rows of a K tile: k-step 0, then k-step 1
    constexpr int PPR = (A_PER + B_PER + ROWS - 1) / ROWS;         // DMA pieces per MFMA row
    constexpr int RING = 3;                                        // A fragments in flight ahead of the MFMAs
    static_assert(FM >= FN && FM >= RING, "wave tile");
    const char* Ac = As + cur * A_BYTES;
    const char* Bc = Bs + cur * B_BYTES;
    // Software pipeline over the 2*FM MFMA rows of the tile.  Only the B fragments of a k-step
    // (FN of them) stay resident; A fragments stream through a ring of three, each read three rows
    // before the MFMAs that use it, and the B fragments of k-step 1 arrive during the last rows of
    // k-step 0.  With interleaved staging one DMA piece of the next tile goes out per row, in the
    // MFMAs' shadow, instead of a burst after the barrier when every wave would issue it at once.
    s8v fb[2][FN], fa[RING];
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[0][j] = read_frag<B_TR>(Bc, wc * FN + j, 0, lane);
#pragma unroll
    for (int q = 0; q < RING; ++q) fa[q] = read_frag<A_TR>(Ac, wr * FM + q, 0, lane);
    if (!INTERLEAVE && more) STAGE(kt + D, nxt);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      constexpr int dummy = 0;
      (void) dummy;
      const int kk = r / FM, i = r % FM;
#pragma unroll
      for (int j = 0; j < FN; ++j)
        // first operand indexes the result's rows (kept in registers): n; second its columns (lanes): m
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, fb[kk][j]),
                                                            __builtin_bit_cast(bf8v, fa[r % RING]), acc[i][j], 0, 0, 0);
      if (INTERLEAVE && more) {
#pragma unroll
        for (int q = 0; q < PPR; ++q) {
          const int pc = r * PPR + q;
          if (pc < A_PER) sta.stage_one(pc, knext, As + nxt * A_BYTES, w);
          else if (pc < A_PER + B_PER) stb.stage_one(pc - A_PER, knext, Bs + nxt * B_BYTES, w);
        }
      }
      if (r + RING < ROWS) fa[r % RING] = read_frag<A_TR>(Ac, wr * FM + (r + RING) % FM, (r + RING) / FM, lane);
      if (kk == 0 && i >= FM - FN) fb[1][i - (FM - FN)] = read_frag<B_TR>(Bc, wc * FN + i - (FM - FN), 1, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  }
  }
#undef STAGE
  // Epilogue.  In the accumulators a lane holds 4 consecutive n of ONE row m, and the 16 lanes of a
  // group hold 16 different rows: stored like that, every instruction touches 16 rows with 32 bytes
  // each.  Each fragment row (16 x 16*FN results) is therefore passed through a wave-private
  // piece of the (now idle) staging LDS and read back with 16 lanes along a row, so that a global
  // instruction covers 4 rows of 64 consecutive results (128 or 256 contiguous bytes per row) instead.
  static_assert(FN == 4, "the epilogue transposes 16 x 64 blocks");
  constexpr int EP_ROW = 64 * 4 + 16;  // bytes of one row of 64 floats, padded against bank conflicts
  __builtin_amdgcn_s_barrier();        // every wave is done with the staging buffers
  char* const scratch = smem + w * (16 * EP_ROW);
  if (m0 + BM <= a.M) {  // uniform over the workgroup: all rows inside
    const int64_t mb = m0 + (int64_t) wr * FM * 16;
    const mi_gemm_epilogue& e = a.e;
    const bool n_edge = n0 + BN > a.N;
#define EPI(MODE, F0, F1)                                                                                          \
  do {                                                                                                             \
    if (n_edge) epilogue_interior<MODE, F0, F1, true, FM, FN>(a, acc, scratch, EP_ROW, mb, n, n_ok, z, lane, bias); \
    else epilogue_interior<MODE, F0, F1, false, FM, FN>(a, acc, scratch, EP_ROW, mb, n, n_ok, z, lane, bias);      \
  } while (0)
    switch (e.mode) {
      case MI_EPI_SLAB: EPI(MI_EPI_SLAB, false, false); break;
      case MI_EPI_BIAS_ACT:
        if (e.bias_rows > 0) { if (e.flags & MI_EPI_RELU) EPI(MI_EPI_BIAS_ACT, true, true); else EPI(MI_EPI_BIAS_ACT, false, true); }
        else { if (e.flags & MI_EPI_RELU) EPI(MI_EPI_BIAS_ACT, true, false); else EPI(MI_EPI_BIAS_ACT, false, false); }
        break;
      case MI_EPI_CROSS_FWD:
        if (e.out1) EPI(MI_EPI_CROSS_FWD, true, false); else EPI(MI_EPI_CROSS_FWD, false, false);
        break;
      case MI_EPI_MASK:
        if (e.in0) EPI(MI_EPI_MASK, true, false); else EPI(MI_EPI_MASK, false, false);
        break;
      default:  // MI_EPI_CROSS_BWD
        if (e.out1) { if (e.in1) EPI(MI_EPI_CROSS_BWD, true, true); else EPI(MI_EPI_CROSS_BWD, true, false); }
        else { if (e.in1) EPI(MI_EPI_CROSS_BWD, false, true); else EPI(MI_EPI_CROSS_BWD, false, false); }
        break;
    }
#undef EPI
    return;
  }
  // a tile hanging over the last row: every position checked, mode looked up per element
  constexpr int CH = 4;
  static_assert(FM % CH == 0, "wave tile");
#pragma unroll
  for (int i0 = 0; i0 < FM; i0 += CH) {
    EpiIn in[CH][4];
#pragma unroll
    for (int ii = 0; ii < CH; ++ii)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t m = m0 + (wr * FM + i0 + ii) * 16 + q * 4 + (lane >> 4);
        if (m < a.M && n_ok) in[ii][q] = epi_load(a.e, m, n);
      }
#pragma unroll
    for (int ii = 0; ii < CH; ++ii) {
      const int i = i0 + ii;
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<f4v*>(scratch + (lane & 15) * EP_ROW + (j * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = q * 4 + (lane >> 4);
        const f4v v = *reinterpret_cast<const f4v*>(scratch + row * EP_ROW + (lane & 15) * 16);
        const int64_t m = m0 + (wr * FM + i) * 16 + row;
        if (m < a.M && n_ok) epi_apply(a.e, a.M, z, m, n, make_float4(v.x, v.y, v.z, v.w), in[ii][q], bias);
      }
    }
  }
}

template <bool A_TR, bool B_TR, int WM, int WN, int FM, int FN, int STAGES>
int launch_gemm(GemmArgs& g, int splits, hipStream_t s) {
  constexpr int BM = 16 * FM * WM, BN = 16 * FN * WN;
  constexpr size_t lds = (size_t) STAGES * (BM + BN) * BK * 2;
  static bool configured = false;  // per instantiation; racing first calls set the same value
  if (lds > 64 * 1024 && !configured) {
    MI_HIP(hipFuncSetAttribute((const void*) gemm_bf16_kernel<A_TR, B_TR, WM, WN, FM, FN, STAGES>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    configured = true;
  }
  g.tiles_n = (int) mi_cdiv(g.N, BN);
  const int64_t tiles = mi_cdiv(g.M, BM) * g.tiles_n;
  MI_REQUIRE(tiles < ((int64_t) 1 << 31), "mi_gemm_bf16: too many tiles");
  hipLaunchKernelGGL((gemm_bf16_kernel<A_TR, B_TR, WM, WN, FM, FN, STAGES>), dim3((unsigned) tiles, (unsigned) splits),
                     dim3(WM * WN * 64), lds, s, g);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <bool A_TR, bool B_TR>
int dispatch_gemm(GemmArgs& g, int splits, hipStream_t s) {
  // Two workgroup shapes (measured on the 65536-row shapes of a DCN-v2 step, TFLOP/s):
  //   256x256, 8 waves x (128x64), 128 KiB LDS, 1 workgroup/CU:  forward 1040-1100, masked dX 640-840, dW 241-429
  //   128x128, 4 waves x (64x64),   64 KiB LDS, 2 workgroups/CU: forward and dX about 10 % below that, dW 650-730
  // The big tile halves the LDS and L2 bytes per flop and wins whenever the operands are read the
  // way they lie (ds_read_b128); with both operands reduction-major (weight gradients) its
  // transposing reads and 241 VGPRs lose, and narrow outputs do not fill its 256 columns.
  const int64_t big = mi_cdiv(g.M, 256) * mi_cdiv(g.N, 256) * splits;
  const bool fills = mi_cdiv(g.N, 256) * 256 * 100 <= g.N * 115;  // at most 15 % of the columns are padding
  // ... and the cross layer's forward epilogue (two input and two output tensors per tile, five 58.7 MB tensors a
  // launch at width 448) runs better under two workgroups per CU, whose epilogues overlap the other's K loop:
  // 54 / 70 / 65 us with the big tile, 48 / 57 / 57 us with the small one for the three layers at batch 65536
  const bool heavy_epilogue = g.e.mode == MI_EPI_CROSS_FWD;
  if (!heavy_epilogue && !A_TR && big >= 128 && fills && (!B_TR || g.N >= 512)) return launch_gemm<A_TR, B_TR, 2, 4, 8, 4, 2>(g, splits, s);
  return launch_gemm<A_TR, B_TR, 2, 2, 4, 4, 2>(g, splits, s);
}

bool al16p(const void* p) { return ((uintptr_t) p & 15) == 0; }
bool al8p(const void* p) { return ((uintptr_t) p & 7) == 0; }

}  // namespace

extern "C" int mi_gemm_bf16(int a_reduction_major, int b_reduction_major, int64_t M, int64_t N, int64_t K,
                            const void* A, int64_t lda, const void* B, int64_t ldb, int k_splits,
                            const mi_gemm_epilogue* epi, void* stream) {
  const char* who = "mi_gemm_bf16";
  MI_REQUIRE(A && B && epi, "%s: null pointer", who);
  MI_REQUIRE(M >= 8 && N >= 8 && K >= BK && M % 8 == 0 && N % 8 == 0 && K % BK == 0,
             "%s: M, N must be multiples of 8 and K of %d (got %lld, %lld, %lld)", who, BK, (long long) M, (long long) N,
             (long long) K);
  MI_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && al16p(A) && al16p(B), "%s: operands need 16-byte aligned rows", who);
  MI_REQUIRE(lda >= (a_reduction_major ? M : K) && ldb >= (b_reduction_major ? N : K), "%s: leading dimension too small", who);
  MI_REQUIRE(k_splits >= 1 && k_splits <= 4096, "%s: k_splits %d", who, k_splits);
  MI_REQUIRE(epi->mode == MI_EPI_SLAB || k_splits == 1, "%s: only MI_EPI_SLAB takes a split reduction", who);
  const mi_gemm_epilogue& e = *epi;
  auto bf_ok = [](const void* p, int64_t ld) { return !p || (al8p(p) && ld % 4 == 0); };
  switch (e.mode) {
    case MI_EPI_SLAB:
      MI_REQUIRE(e.out && al16p(e.out) && e.ld_out % 4 == 0 && e.ld_out >= N, "%s: slab output", who);
      break;
    case MI_EPI_BIAS_ACT:
      MI_REQUIRE(e.out && bf_ok(e.out, e.ld_out) && (!e.bias || al16p(e.bias)), "%s: bias/activation epilogue", who);
      MI_REQUIRE(e.bias_rows == 0 || (e.bias && e.bias_rows > 0 && e.bias_ld % 4 == 0 && e.bias_ld >= N && M < ((int64_t) 1 << 32)),
                 "%s: a bias matrix needs bias, bias_rows > 0, bias_ld %% 4 == 0, bias_ld >= N", who);
      break;
    case MI_EPI_CROSS_FWD:
      MI_REQUIRE(e.out && e.bias && e.in0 && e.in1 && al16p(e.bias) && bf_ok(e.out, e.ld_out) && bf_ok(e.out1, e.ld_out1) &&
                     bf_ok(e.in0, e.ld_in0) && bf_ok(e.in1, e.ld_in1),
                 "%s: cross-forward epilogue", who);
      break;
    case MI_EPI_MASK:
      MI_REQUIRE(e.out && bf_ok(e.out, e.ld_out) && bf_ok(e.in0, e.ld_in0), "%s: mask epilogue", who);
      break;
    case MI_EPI_CROSS_BWD:
      MI_REQUIRE(e.out && bf_ok(e.out, e.ld_out) && bf_ok(e.in1, e.ld_in1) &&
                     (!e.out1 || (e.in0 && bf_ok(e.out1, e.ld_out1) && bf_ok(e.in0, e.ld_in0))),
                 "%s: cross-backward epilogue", who);
      break;
    default:
      mi_set_error("%s: unknown epilogue mode %d", who, e.mode);
      return MI_ERR_INVALID_ARG;
  }
  int64_t k_per = mi_cdiv(mi_cdiv(K, k_splits), BK) * BK;
  const int splits = (int) mi_cdiv(K, k_per);
  MI_REQUIRE(splits == k_splits, "%s: K=%lld does not divide into %d slices of whole %d-deep tiles (would be %d)", who,
             (long long) K, k_splits, BK, splits);
  GemmArgs g;
  g.A = (const unsigned short*) A;
  g.B = (const unsigned short*) B;
  g.lda = lda; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.k_per_split = k_per;
  g.tiles_n = 0;
  g.e = e;
  hipStream_t s = (hipStream_t) stream;
  if (!a_reduction_major && !b_reduction_major) return dispatch_gemm<false, false>(g, splits, s);
  if (!a_reduction_major && b_reduction_major) return dispatch_gemm<false, true>(g, splits, s);
  if (a_reduction_major && b_reduction_major) return dispatch_gemm<true, true>(g, splits, s);
  return dispatch_gemm<true, false>(g, splits, s);
}
