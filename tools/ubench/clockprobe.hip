// clockprobe.hip — development micro-benchmark (not part of the product): what the shader clock does over a burst of
// sub-millisecond streaming kernels that starts on an idle GPU.  VERDICT r2 "next" #4 asks for the trace behind DESIGN's
// "clock excursion" reading of the C4 shard's slow launches (launches 10-20 of a burst run 10-20 % slower than launches
// 1-9 and 30+).  Each launch streams 1.89 GB of float32 (the C4 shard: 456 x 1 036 800) with 16-byte non-temporal loads;
// lane 0 of workgroup 0 reads the shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) when it
// starts and when it ends, so   sclk = 100 MHz x d(s_memtime) / d(s_memrealtime)   is the clock averaged over that
// workgroup's life = the kernel.  Output: one JSON line per launch (index, start in ms since the burst began, duration by
// HIP events, sclk in MHz), for bursts after 0.5 s and after 0 s of idle.
// Build: hipcc --offload-arch=gfx950 -O3 -o clockprobe clockprobe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) stream_read(const f4* __restrict__ src, long n16, float* sink, unsigned long long* stamps) {
  unsigned long long c0 = 0, r0 = 0;
  const bool probe = blockIdx.x == 0 && threadIdx.x == 0;
  if (probe) {
    c0 = __builtin_readcyclecounter();      // s_memtime: shader cycles
    r0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(src + i);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
  if (probe) {
    // (workgroup 0 ends with the grid-stride loop like every other: its life is the kernel's, within a tail of a few us)
    stamps[0] = __builtin_readcyclecounter() - c0;
    stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

int main() {
  const long bytes = 456L * 1036800L * 4L;
  f4* src;
  float* sink;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(src, 0, bytes));
  const int launches = 80;
  unsigned long long* stamps;
  CK(hipHostMalloc((void**)&stamps, sizeof(unsigned long long) * 2 * launches, hipHostMallocDefault));
  std::vector<hipEvent_t> ev(launches + 1);
  for (auto& evt : ev) CK(hipEventCreate(&evt));
  hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, src, bytes / 16, sink, stamps);  // (module load)
  CK(hipDeviceSynchronize());
  for (int idle_ms : {500, 0, 2000}) {
    std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
    CK(hipEventRecord(ev[0]));
    for (int k = 0; k < launches; ++k) {
      hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, src, bytes / 16, sink, stamps + 2 * k);
      CK(hipEventRecord(ev[k + 1]));
    }
    CK(hipDeviceSynchronize());
    for (int k = 0; k < launches; ++k) {
      float t_end, dur;
      CK(hipEventElapsedTime(&t_end, ev[0], ev[k + 1]));
      CK(hipEventElapsedTime(&dur, ev[k], ev[k + 1]));
      const double sclk = stamps[2 * k + 1] ? 100.0 * (double)stamps[2 * k] / (double)stamps[2 * k + 1] : 0.0;
      printf("{\"idle_ms_before_burst\": %d, \"launch\": %d, \"end_ms\": %.4f, \"launch_ms\": %.4f, \"sclk_mhz\": %.0f, \"gbs\": %.0f}\n", idle_ms, k, t_end, dur,
             sclk, bytes / (dur * 1e-3) / 1e9);
    }
  }
  return 0;
}
