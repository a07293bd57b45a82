// lanesread.hip — development micro-benchmark (not part of the product): the read pattern of the row-per-lane kernels
// (hist_lanes: a (C, M) float32 array reduced over its LEADING axis, lane <-> row, loop <-> leading index) with one row per
// lane (4-byte loads, 256 threads per 256 rows) against four rows per lane (16-byte loads, 64 threads per 256 rows), same
// bytes in flight per lane group; bare loads summed into a register.  Build: hipcc --offload-arch=gfx950 -O3 -o lanesread lanesread.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void __launch_bounds__(256) rows1(const float* a, long M, int C, float* out) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  float acc = 0;
  for (int c = 0; c + UNROLL <= C; c += UNROLL) {
    float v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(a + (long)(c + u) * M + r);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc == 1234.5f) out[0] = acc;
}
template <int UNROLL, int TPB>
__global__ void __launch_bounds__(TPB) rows4(const float* a, long M, int C, float* out) {
  const long r = ((long)blockIdx.x * TPB + threadIdx.x) * 4;
  f4 acc = {0, 0, 0, 0};
  for (int c = 0; c + UNROLL <= C; c += UNROLL) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a + (long)(c + u) * M + r));
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = acc[0];
}
template <typename F>
static void timeit(const char* name, F launch, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  printf("{\"variant\": \"%s\", \"ms\": %.4f, \"TBs\": %.2f}\n", name, best, bytes / (best * 1e-3) / 1e12);
}
int main() {
  const long M = 720L * 1440L; const int C = 365;
  float *a, *out;
  CK(hipMalloc(&a, M * C * 4)); CK(hipMalloc(&out, 64)); CK(hipMemset(a, 0, M * C * 4));
  const double bytes = (double)M * C * 4;
  timeit("1 row per lane, 4-byte loads, 8 in flight, 256 threads", [&] { hipLaunchKernelGGL(rows1<8>, dim3(M / 256), dim3(256), 0, 0, a, M, C, out); }, bytes);
  timeit("1 row per lane, 4-byte loads, 16 in flight, 256 threads", [&] { hipLaunchKernelGGL(rows1<16>, dim3(M / 256), dim3(256), 0, 0, a, M, C, out); }, bytes);
  timeit("4 rows per lane, 16-byte loads, 8 in flight, 64 threads", [&] { hipLaunchKernelGGL((rows4<8, 64>), dim3(M / 256), dim3(64), 0, 0, a, M, C, out); }, bytes);
  timeit("4 rows per lane, 16-byte loads, 4 in flight, 64 threads", [&] { hipLaunchKernelGGL((rows4<4, 64>), dim3(M / 256), dim3(64), 0, 0, a, M, C, out); }, bytes);
  timeit("4 rows per lane, 16-byte loads, 8 in flight, 256 threads", [&] { hipLaunchKernelGGL((rows4<8, 256>), dim3(M / 1024), dim3(256), 0, 0, a, M, C, out); }, bytes);
  timeit("4 rows per lane, 16-byte loads, 4 in flight, 256 threads", [&] { hipLaunchKernelGGL((rows4<4, 256>), dim3(M / 1024), dim3(256), 0, 0, a, M, C, out); }, bytes);
  return 0;
}
