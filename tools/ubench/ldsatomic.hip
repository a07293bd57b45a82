// ldsatomic.hip — development micro-benchmark (not part of the product): what a CU's LDS gives for the atomics the
// histogram kernels issue, with no memory traffic at all.  Every lane performs `iters` x 8 atomic adds on a 16 Ki-element
// LDS histogram (the partition size of the adding-up pass; 128 KB of float64) at addresses drawn by a per-lane LCG, for
//   f64 / u32 / returning u32 adds,   1024 / 768 / 512 / 256 threads per CU (one workgroup per CU),   and the patterns
//   random        uniform over the 16 Ki bins (C5's adding-up pass with uniform samples)
//   normal        a sum of four uniforms, sigma ~ 1/14 of the range (bins of a Gaussian sample, C3 / C5 as benched)
//   lane_private  bin % 32 == lane % 32: conflict-free for 8-byte slots (what the lane-bank copies of hist_fast give C2)
//   one           every lane the same bin;  few4 / few16 / few64: that many bins (the rank counters of a routing pass with as
//                 many partitions);  random_but_1pct_on_one_address: the adding-up pass with a trash slot
// Output: one JSON line per variant with atomics per clock and CU (clock from s_memtime) and the wall rate in G/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o ldsatomic ldsatomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int kBins = 16384;
extern __shared__ unsigned char smem[];

enum { RANDOM = 0, NORMAL = 1, LANE_PRIVATE = 2, ONE = 3, FEW4 = 4, FEW16 = 5, FEW64 = 6, MOSTLY_RANDOM = 7 };

__device__ __forceinline__ uint32_t lcg(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return s;
}

template <int KIND /* 0 f64, 1 u32, 2 u32 returning */, int PATTERN>
__global__ void __launch_bounds__(1024) lds_adds(int iters, double* sink, unsigned long long* stamps) {
  double* hd = reinterpret_cast<double*>(smem);
  uint32_t* hu = reinterpret_cast<uint32_t*>(smem);
  for (int i = threadIdx.x; i < kBins; i += blockDim.x) {
    if (KIND == 0) hd[i] = 0.0;
    else hu[i] = 0u;
  }
  __syncthreads();
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  const uint32_t lane = threadIdx.x & 31u;
  unsigned long long c0 = 0;
  if (threadIdx.x == 0) c0 = __builtin_readcyclecounter();
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t bin[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t r = lcg(s);
      if (PATTERN == RANDOM) bin[k] = r >> 18;
      else if (PATTERN == NORMAL) {
        const uint32_t r2 = lcg(s);
        bin[k] = (((r >> 20) + ((r >> 8) & 0xfffu) + (r2 >> 20) + ((r2 >> 8) & 0xfffu)) >> 4) + 6144u;  // 4 x U[0, 4096) / 16: sigma ~ 148 bins
      } else if (PATTERN == LANE_PRIVATE) bin[k] = ((r >> 18) & ~31u) | lane;
      else if (PATTERN == FEW4) bin[k] = 777u + (r >> 30) * 33u;         // 4 addresses (a routing pass with 4 partitions)
      else if (PATTERN == FEW16) bin[k] = 777u + (r >> 28) * 33u;        // 16 addresses
      else if (PATTERN == FEW64) bin[k] = 777u + (r >> 26) * 33u;        // 64 addresses
      else if (PATTERN == MOSTLY_RANDOM) bin[k] = (r & 0xffu) < 3u ? 16000u : (r >> 18);  // ~1 % of the adds on ONE address
      else bin[k] = 777u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (KIND == 0) unsafeAtomicAdd(hd + bin[k], 1.0);
      else if (KIND == 1) atomicAdd(hu + bin[k], 1u);
      else acc += atomicAdd(hu + bin[k], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) stamps[blockIdx.x] = __builtin_readcyclecounter() - c0;
  double t = 0;
  for (int i = threadIdx.x; i < kBins; i += blockDim.x) t += KIND == 0 ? hd[i] : (double)hu[i];
  if (t == -1.0 || acc == 0xdeadbeefu) sink[0] = t;
}

template <int KIND, int PATTERN>
static void run(const char* kind, const char* pattern, int block, int cus, double* sink, unsigned long long* stamps) {
  const int iters = 2000;
  const size_t lds = (size_t)kBins * (KIND == 0 ? 8 : 4);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_adds<KIND, PATTERN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((lds_adds<KIND, PATTERN>), dim3(cus), dim3(block), lds, 0, iters, sink, stamps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  double cyc = 0;
  for (int i = 0; i < cus; ++i) cyc += (double)stamps[i];
  cyc /= cus;
  const double per_cu = (double)iters * 8.0 * block;
  printf("{\"kind\": \"%s\", \"pattern\": \"%s\", \"threads_per_cu\": %d, \"atomics_per_clock_per_cu\": %.3f, \"G_atomics_per_s\": %.1f, \"ms\": %.4f}\n",
         kind, pattern, block, per_cu / cyc, per_cu * cus / (best * 1e-3) / 1e9, best);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* sink;
  unsigned long long* stamps;
  CK(hipMalloc(&sink, 64));
  CK(hipHostMalloc(&stamps, sizeof(unsigned long long) * cus));
  for (int block : {1024, 768, 512, 256}) {
    run<0, RANDOM>("f64", "random", block, cus, sink, stamps);
    run<0, NORMAL>("f64", "normal", block, cus, sink, stamps);
    run<0, LANE_PRIVATE>("f64", "lane_private", block, cus, sink, stamps);
    run<0, ONE>("f64", "one", block, cus, sink, stamps);
    run<1, RANDOM>("u32", "random", block, cus, sink, stamps);
    run<1, NORMAL>("u32", "normal", block, cus, sink, stamps);
    run<1, LANE_PRIVATE>("u32", "lane_private", block, cus, sink, stamps);
    run<2, RANDOM>("u32_returning", "random", block, cus, sink, stamps);
    run<2, NORMAL>("u32_returning", "normal", block, cus, sink, stamps);
    if (block == 1024) {
      run<1, ONE>("u32", "one", block, cus, sink, stamps);
      run<2, ONE>("u32_returning", "one", block, cus, sink, stamps);
      run<1, FEW4>("u32", "few4", block, cus, sink, stamps);
      run<2, FEW4>("u32_returning", "few4", block, cus, sink, stamps);
      run<2, FEW16>("u32_returning", "few16", block, cus, sink, stamps);
      run<2, FEW64>("u32_returning", "few64", block, cus, sink, stamps);
      run<1, MOSTLY_RANDOM>("u32", "random_but_1pct_on_one_address", block, cus, sink, stamps);
      run<0, MOSTLY_RANDOM>("f64", "random_but_1pct_on_one_address", block, cus, sink, stamps);
      run<0, FEW16>("f64", "few16", block, cus, sink, stamps);
    }
  }
  return 0;
}
