// l2atomic.hip — development micro-benchmark: rate of global atomic adds to random addresses of a
// region of a given size, by memory scope.  Question: do workgroup-scope atomics (executed in the
// issuing XCD's L2) run fast enough to hold mid-size histograms (beyond LDS, within L2) in one
// private copy per XCD?  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2atomic l2atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// every lane does `per_lane` atomic adds at pseudo-random slots of copy (XCD id) of the table
template <typename T, int SCOPE, bool PER_XCD>
__global__ void __launch_bounds__(256) hammer(T* table, uint32_t slots_mask, uint32_t copy_stride, int per_lane) {
  T* my = table + (PER_XCD ? (size_t)xcc_id() * copy_stride : 0);
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < per_lane; ++i) {
    s = s * 1664525u + 1013904223u;
    const uint32_t slot = (s >> 7) & slots_mask;
    __hip_atomic_fetch_add(my + slot, (T)1, __ATOMIC_RELAXED, SCOPE);
  }
}

template <typename T, int SCOPE, bool PER_XCD>
void run(const char* tname, const char* sname, T* table, size_t region_bytes, int grid) {
  const uint32_t slots = (uint32_t)(region_bytes / sizeof(T));
  const int per_lane = 512;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((hammer<T, SCOPE, PER_XCD>), dim3(grid), dim3(256), 0, 0, table, slots - 1, slots, per_lane);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r >= 1) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double n = (double)grid * 256 * per_lane;
  printf("{\"case\": \"l2atomic\", \"type\": \"%s\", \"scope\": \"%s\", \"per_xcd_copy\": %d, \"region_kib\": %zu, \"grid\": %d, \"ms\": %.4f, \"atomics_per_s\": %.3e}\n",
         tname, sname, (int)PER_XCD, region_bytes / 1024, grid, ms[ms.size() / 2], n / ms[ms.size() / 2] * 1e3);
  fflush(stdout);
}

int main() {
  void* table;
  CK(hipMalloc(&table, (size_t)8 * 16 * 1024 * 1024));
  CK(hipMemset(table, 0, (size_t)8 * 16 * 1024 * 1024));
  for (size_t kib : {64, 512, 2048, 8192}) {
    const size_t bytes = kib * 1024;
    for (int grid : {2048, 8192}) {
      run<double, __HIP_MEMORY_SCOPE_AGENT, false>("f64", "agent", (double*)table, bytes, grid);
      run<double, __HIP_MEMORY_SCOPE_WORKGROUP, true>("f64", "workgroup", (double*)table, bytes, grid);
      run<double, __HIP_MEMORY_SCOPE_WAVEFRONT, true>("f64", "wavefront", (double*)table, bytes, grid);
      run<unsigned int, __HIP_MEMORY_SCOPE_AGENT, false>("u32", "agent", (unsigned int*)table, bytes, grid);
      run<unsigned int, __HIP_MEMORY_SCOPE_WORKGROUP, true>("u32", "workgroup", (unsigned int*)table, bytes, grid);
    }
  }
  // sanity: per-XCD copies of a u32 table really receive everything (sum over copies == issued)
  CK(hipMemset(table, 0, (size_t)8 * 16 * 1024 * 1024));
  const uint32_t slots = 1024;
  hipLaunchKernelGGL((hammer<unsigned int, __HIP_MEMORY_SCOPE_WORKGROUP, true>), dim3(1024), dim3(256), 0, 0, (unsigned int*)table,
                     slots - 1, slots, 64);
  CK(hipDeviceSynchronize());
  std::vector<unsigned int> h(8 * slots);
  CK(hipMemcpy(h.data(), table, h.size() * 4, hipMemcpyDeviceToHost));
  unsigned long long tot = 0, per[8] = {0};
  for (int c = 0; c < 8; ++c) for (uint32_t i = 0; i < slots; ++i) { tot += h[c * slots + i]; per[c] += h[c * slots + i]; }
  printf("{\"case\": \"l2atomic_sanity\", \"issued\": %llu, \"counted\": %llu, \"per_xcd\": [%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu]}\n",
         1024ull * 256 * 64, tot, per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7]);
  return 0;
}
