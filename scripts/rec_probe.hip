// Probe: rate of a read-modify-write of sorted distinct 256-byte records (the FM Adam row record: V[16] | w1 m_w1
// v_w1 pad[13] | m[16] | v[16]) by access shape.  Not part of the library.
//   shape 0: four lanes per record: three float4 loads (V, m, v) and three dword loads (the first-order weight and
//            its two moments, in V's line) per lane, stored back the same way
//   shape 1: the same without the dword accesses
//   shape 2: sixteen lanes per record, ONE float4 load and store per lane (every line touched by one instruction)
//   shape 3: as 2, stores only from the lanes that hold data (0-4, 8-15)
// R records in flight per lane group.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int R, int SHAPE>
__global__ __launch_bounds__(256) void rec_rmw_kernel(float* __restrict__ rec, const uint32_t* __restrict__ rows, int n) {
  if constexpr (SHAPE <= 1) {
    const int grp = (blockIdx.x * 256 + threadIdx.x) / 4, c = threadIdx.x & 3;
    const int first = grp * R;
    if (first >= n) return;
    uint32_t row[R];
#pragma unroll
    for (int u = 0; u < R; ++u) row[u] = rows[first + u < n ? first + u : first];
    float4 V[R], m[R], v[R];
    float w[R], mw[R], vw[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const float* p = rec + (size_t) row[u] * 64;
      V[u] = *reinterpret_cast<const float4*>(p + c * 4);
      m[u] = *reinterpret_cast<const float4*>(p + 32 + c * 4);
      v[u] = *reinterpret_cast<const float4*>(p + 48 + c * 4);
      if (SHAPE == 0) { w[u] = p[16]; mw[u] = p[17]; vw[u] = p[18]; }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (first + u >= n) continue;
      float* p = rec + (size_t) row[u] * 64;
      m[u].x += 0.1f * (V[u].x - m[u].x); m[u].y += 0.1f * (V[u].y - m[u].y);
      m[u].z += 0.1f * (V[u].z - m[u].z); m[u].w += 0.1f * (V[u].w - m[u].w);
      v[u].x += 0.001f * (V[u].x * V[u].x - v[u].x); v[u].y += 0.001f * (V[u].y * V[u].y - v[u].y);
      v[u].z += 0.001f * (V[u].z * V[u].z - v[u].z); v[u].w += 0.001f * (V[u].w * V[u].w - v[u].w);
      V[u].x -= 1e-3f * m[u].x; V[u].y -= 1e-3f * m[u].y; V[u].z -= 1e-3f * m[u].z; V[u].w -= 1e-3f * m[u].w;
      *reinterpret_cast<float4*>(p + c * 4) = V[u];
      *reinterpret_cast<float4*>(p + 32 + c * 4) = m[u];
      *reinterpret_cast<float4*>(p + 48 + c * 4) = v[u];
      if (SHAPE == 0 && c == 0) { p[16] = w[u] - 1e-3f * mw[u]; p[17] = mw[u] * 0.9f; p[18] = vw[u] * 0.999f; }
    }
  } else {
    const int grp = (blockIdx.x * 256 + threadIdx.x) / 16, j = threadIdx.x & 15;
    const int first = grp * R;
    if (first >= n) return;
    uint32_t row[R];
#pragma unroll
    for (int u = 0; u < R; ++u) row[u] = rows[first + u < n ? first + u : first];
    float4 x[R];
#pragma unroll
    for (int u = 0; u < R; ++u) x[u] = *reinterpret_cast<const float4*>(rec + (size_t) row[u] * 64 + j * 4);
#pragma unroll
    for (int u = 0; u < R; ++u) {
      if (first + u >= n) continue;
      x[u].x = x[u].x * 0.999f + 1e-3f; x[u].y = x[u].y * 0.999f + 1e-3f;
      x[u].z = x[u].z * 0.999f + 1e-3f; x[u].w = x[u].w * 0.999f + 1e-3f;
      if (SHAPE == 2 || j <= 4 || j >= 8) *reinterpret_cast<float4*>(rec + (size_t) row[u] * 64 + j * 4) = x[u];
    }
  }
}

// Forward-shaped gather: random (unsorted) records, V (64 B) and the first-order weight of each.
//   shape 0: four lanes per record, a float4 of V and the dword w1 per lane (two instructions on one line)
//   shape 1: V only;  shape 2: five of eight lanes per record load a float4 (lane 4: the scalars' 16 bytes)
template <int R, int SHAPE>
__global__ __launch_bounds__(256) void rec_gather_kernel(const float* __restrict__ rec, const uint32_t* __restrict__ rows,
                                                         int n, float* __restrict__ out) {
  constexpr int L = SHAPE == 2 ? 8 : 4;
  const int grp = (blockIdx.x * 256 + threadIdx.x) / L, c = threadIdx.x & (L - 1);
  const int first = grp * R;
  if (first >= n) return;
  uint32_t row[R];
#pragma unroll
  for (int u = 0; u < R; ++u) row[u] = rows[first + u < n ? first + u : first];
  float4 V[R];
  float w[R];
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const float* p = rec + (size_t) row[u] * 64;
    V[u] = *reinterpret_cast<const float4*>(p + (c < 5 ? c : 0) * 4);
    w[u] = SHAPE == 0 ? p[16] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < R; ++u) acc += V[u].x + V[u].y + V[u].z + V[u].w + w[u];
  if (acc == 1234.56789f) out[0] = acc;
}

extern "C" int rec_probe(void* rec, const uint32_t* rows, int n, int shape, int R, int gather, float* out, void* stream) {
  hipStream_t s = (hipStream_t) stream;
#define RMW(Rv, SH) do { const int per = (SH <= 1 ? 64 : 16) * Rv; \
    hipLaunchKernelGGL((rec_rmw_kernel<Rv, SH>), dim3((n + per - 1) / per), dim3(256), 0, s, (float*) rec, rows, n); } while (0)
#define GAT(Rv, SH) do { const int per = (SH == 2 ? 32 : 64) * Rv; \
    hipLaunchKernelGGL((rec_gather_kernel<Rv, SH>), dim3((n + per - 1) / per), dim3(256), 0, s, (const float*) rec, rows, n, out); } while (0)
#define BYR(M, SH) switch (R) { case 1: M(1, SH); break; case 2: M(2, SH); break; case 4: M(4, SH); break; case 8: M(8, SH); break; case 16: M(16, SH); break; default: return 1; }
  if (!gather) {
    switch (shape) { case 0: BYR(RMW, 0) break; case 1: BYR(RMW, 1) break; case 2: BYR(RMW, 2) break; case 3: BYR(RMW, 3) break; default: return 1; }
  } else {
    switch (shape) { case 0: BYR(GAT, 0) break; case 1: BYR(GAT, 1) break; case 2: BYR(GAT, 2) break; default: return 1; }
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
