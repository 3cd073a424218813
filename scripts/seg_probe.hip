// Probe: achievable HBM rate for random contiguous segments of S bytes (S = 64 ... 8192) of a large
// table - read, read-modify-write, or write: the ceiling of a row gather / row update at that row
// width.  Segments of one launch are distinct (a permutation prefix), so no two groups update the same one.
// Not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int R, int MODE>  // MODE 0: read; 1: read-modify-write of the segment; 2: write only
__global__ __launch_bounds__(256) void seg_read_kernel(float4* __restrict__ table, const int32_t* __restrict__ idx,
                                                       int64_t n_seg, int seg_vec /* float4 per segment */,
                                                       int lanes /* lanes per segment, <= 64 */, float* __restrict__ out) {
  const int groups = 256 / lanes;
  const int sub = threadIdx.x / lanes, c = threadIdx.x - sub * lanes;
  const int pieces = seg_vec / lanes;
  const int64_t first = ((int64_t) blockIdx.x * groups + sub) * R;
  float acc = 0.f;
  int32_t id[R];
#pragma unroll
  for (int u = 0; u < R; ++u) id[u] = first + u < n_seg ? idx[first + u] : -1;
  for (int p = 0; p < pieces; ++p) {
    float4 v[R];
#pragma unroll
    for (int u = 0; u < R; ++u)
      v[u] = (MODE != 2 && id[u] >= 0) ? table[(int64_t) id[u] * seg_vec + p * lanes + c]
                                       : make_float4(0.f, (float) c, 0.f, 0.f);
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < R; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    } else {
#pragma unroll
      for (int u = 0; u < R; ++u)
        if (id[u] >= 0) {
          v[u].x += 1.0f;
          table[(int64_t) id[u] * seg_vec + p * lanes + c] = v[u];
        }
    }
  }
  if (acc == 1234.56789f) out[0] = acc;  // never true for the probe's data; keeps the loads alive
}

extern "C" int seg_probe(void* table, const int32_t* idx, int64_t n_seg, int seg_bytes, int rows_in_flight, int mode,
                         float* out, void* stream) {
  const int seg_vec = seg_bytes / 16;
  const int lanes = seg_vec < 64 ? seg_vec : 64;
  const int groups = 256 / lanes;
  hipStream_t s = (hipStream_t) stream;
#define GO1(R, M)                                                                                                 \
  hipLaunchKernelGGL((seg_read_kernel<R, M>), dim3((unsigned) ((n_seg + (int64_t) groups * R - 1) / ((int64_t) groups * R))), \
                     dim3(256), 0, s, (float4*) table, idx, n_seg, seg_vec, lanes, out)
#define GO(R) do { if (mode == 0) GO1(R, 0); else if (mode == 1) GO1(R, 1); else GO1(R, 2); } while (0)
  switch (rows_in_flight) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 8: GO(8); break;
    default: return 1;
  }
#undef GO
#undef GO1
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
