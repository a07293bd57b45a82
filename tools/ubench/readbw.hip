// readbw.hip — development micro-benchmark: ceiling of a read-only HBM stream on MI355X with the
// same load shape as hist_fast (16-byte non-temporal loads, tiles interleaved over the grid).
// Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -o readbw readbw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int UNROLL, bool NT, int NARR>
__global__ void __launch_bounds__(1024) rd(const double* a, const double* b, long n, double* out) {
  const long tile = (long)blockDim.x * 2 * UNROLL;
  const long ntiles = n / tile;
  double acc = 0;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long base = t * tile;
    d2 v[NARR][UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long i = base + ((long)u * blockDim.x + threadIdx.x) * 2;
      if (NT) { v[0][u] = __builtin_nontemporal_load((const d2*)(a + i)); if (NARR > 1) v[NARR - 1][u] = __builtin_nontemporal_load((const d2*)(b + i)); }
      else { v[0][u] = *(const d2*)(a + i); if (NARR > 1) v[NARR - 1][u] = *(const d2*)(b + i); }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int k = 0; k < NARR; ++k) acc += v[k][u][0] + v[k][u][1];
  }
  if (acc == 123.456) out[0] = acc;
}

template <int UNROLL, bool NT, int NARR>
void run(const char* name, const double* a, const double* b, long n, double* out, int block, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < 7; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rd<UNROLL, NT, NARR>), dim3(grid), dim3(block), 0, 0, a, b, n, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  printf("{\"case\": \"readbw\", \"variant\": \"%s\", \"arrays\": %d, \"unroll\": %d, \"nt\": %d, \"block\": %d, \"grid\": %d, \"ms\": %.4f, \"gbs\": %.1f}\n",
         name, NARR, UNROLL, (int)NT, block, grid, ms[ms.size() / 2], 8.0 * NARR * n / ms[ms.size() / 2] / 1e6);
}

// `readbw quick`: the handful of geometries the evidence sets carry (tools/evidence_set.sh -> <tag>_bare_read_ceiling.txt):
// what a kernel that ONLY reads one / two 8 GB streams reaches on the box — and in the very gpurun call — the bench lines
// of the set were measured on, so that `roofline.frac` can be read against the chip's own read ceiling, not a quoted one.
int main(int argc, char** argv) {
  const bool quick = argc > 1 && argv[1][0] == 'q';
  const long n = 1000000000L;
  double *a, *b, *out;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0x11, n * 8)); CK(hipMemset(b, 0x22, n * 8));
  if (quick) {
    for (int rep = 0; rep < 2; ++rep)
      for (int block : {256, 512})
        for (int grid : {1024, 2048}) {
          run<8, true, 1>("one_array", a, b, n, out, block, grid);
          run<4, true, 2>("two_arrays", a, b, n, out, block, grid);
        }
    return 0;
  }
  for (int block : {256, 512, 1024})
    for (int grid : {256, 512, 1024, 2048, 4096}) {
      run<4, true, 2>("two_arrays", a, b, n, out, block, grid);
      run<4, false, 2>("two_arrays", a, b, n, out, block, grid);
      run<4, true, 1>("one_array", a, b, n, out, block, grid);
      run<8, true, 1>("one_array", a, b, n, out, block, grid);
      run<8, true, 2>("two_arrays", a, b, n, out, block, grid);
      run<2, true, 2>("two_arrays", a, b, n, out, block, grid);
    }
  return 0;
}
