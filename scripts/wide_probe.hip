// Probe: read-modify-write of sorted distinct 512-byte rows (dim 128 fp32) with one fp32 state word per row, by where
// the state word lives.  Not part of the library.  32 lanes per row, one float4 each.
//   mode 0: rows only (stride 128 floats), no state
//   mode 1: rows (stride 128) + the state word in a separate array state[row]
//   mode 2: the state word behind the row in the same record, `stride` floats per record (132: packed, 160: records
//           start on 128-byte lines)
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void wide_rmw_kernel(float* __restrict__ w, float* __restrict__ state,
                                                      const uint32_t* __restrict__ rows, int n, int stride) {
  const int grp = (blockIdx.x * 256 + threadIdx.x) / 32, c = threadIdx.x & 31;
  if (grp >= n) return;
  const size_t row = rows[grp];
  float* p = w + row * (size_t) stride;
  float4 x = *reinterpret_cast<const float4*>(p + c * 4);
  float s = 1.f;
  if (MODE == 1 && c == 0) s = state[row];
  if (MODE == 2 && c == 0) s = p[128];
  s = __shfl(s, threadIdx.x & 32, 64) + 1e-3f;
  const float k = 1e-3f * __frsqrt_rn(s);
  x.x -= k * x.x; x.y -= k * x.y; x.z -= k * x.z; x.w -= k * x.w;
  *reinterpret_cast<float4*>(p + c * 4) = x;
  if (MODE == 1 && c == 0) state[row] = s;
  if (MODE == 2 && c == 0) p[128] = s;
}

extern "C" int wide_probe(float* w, float* state, const uint32_t* rows, int n, int stride, int mode, void* stream) {
  const dim3 grid((unsigned) (((size_t) n * 32 + 255) / 256));
  hipStream_t s = (hipStream_t) stream;
  if (mode == 0) hipLaunchKernelGGL(wide_rmw_kernel<0>, grid, dim3(256), 0, s, w, state, rows, n, stride);
  else if (mode == 1) hipLaunchKernelGGL(wide_rmw_kernel<1>, grid, dim3(256), 0, s, w, state, rows, n, stride);
  else hipLaunchKernelGGL(wide_rmw_kernel<2>, grid, dim3(256), 0, s, w, state, rows, n, stride);
  return (int) hipGetLastError();
}
