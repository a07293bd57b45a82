// mall.hip — development micro-benchmark (not part of the product): does the 256 MiB Infinity Cache
// keep freshly WRITTEN record streams on-die, so that a partition pass whose records are consumed
// chunk by chunk never pays HBM for them?  Emulates the C5 traffic shape: read 24 B/sample (nt),
// write 10 B/sample of records, read the records back.
//   A  write S bytes, then read them back: read-back rate against S
//   B  chunked pipeline over a 12 GB source: per chunk {produce: read 2.4 S nt + write S to the SAME
//      record buffer; consume: read S}, against the unchunked run (records of every chunk kept apart)
// Build: hipcc --offload-arch=gfx950 -O3 -o mall mall.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) wr(u4* dst, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    u4 v = {(unsigned)i, 1u, 2u, 3u};
    if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

template <bool NT>
__global__ void __launch_bounds__(256) rd(const u4* src, long n16, unsigned* out) {
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    u4 v = NT ? __builtin_nontemporal_load(src + i) : src[i];
    acc += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345u) out[0] = acc;
}

// produce: per lane and round 12 x 16 B read (nt) from src, 5 x 16 B written to rec
template <bool NTW>
__global__ void __launch_bounds__(256) produce(const u4* src, long rounds, u4* rec) {
  const long lanes = (long)gridDim.x * blockDim.x;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long r = 0; r < rounds; ++r) {
    u4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = __builtin_nontemporal_load(src + (r * 12 + k) * lanes + t);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      u4 o = v[2 * k] + v[2 * k + 1];
      if (k == 0) o += v[10] ^ v[11];
      if (NTW) __builtin_nontemporal_store(o, rec + (r * 5 + k) * lanes + t); else rec[(r * 5 + k) * lanes + t] = o;
    }
  }
}

static float median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  hipEvent_t e0, e1, e2, e3;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  unsigned* out; CK(hipMalloc(&out, 64));
  const long GB = 1L << 30, MB = 1L << 20;
  u4 *src, *rec;
  const long src_bytes = 12 * GB, rec_bytes = 5 * GB;
  CK(hipMalloc(&src, src_bytes)); CK(hipMalloc(&rec, rec_bytes));
  CK(hipMemset(src, 0x11, src_bytes)); CK(hipMemset(rec, 0, rec_bytes));
  const int grid = 2048, block = 256;

  // ---- A: write then read back ---------------------------------------------------------------
  for (int ntw = 0; ntw < 2; ++ntw)
    for (int ntr = 0; ntr < 2; ++ntr)
      for (long S : {16 * MB, 32 * MB, 64 * MB, 96 * MB, 128 * MB, 192 * MB, 256 * MB, 512 * MB, 2048 * MB}) {
        std::vector<float> tw, tr, trr;
        for (int it = 0; it < 7; ++it) {
          CK(hipEventRecord(e0));
          if (ntw) hipLaunchKernelGGL(wr<true>, dim3(grid), dim3(block), 0, 0, rec, S / 16);
          else hipLaunchKernelGGL(wr<false>, dim3(grid), dim3(block), 0, 0, rec, S / 16);
          CK(hipEventRecord(e1));
          if (ntr) hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(block), 0, 0, rec, S / 16, out);
          else hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(block), 0, 0, rec, S / 16, out);
          CK(hipEventRecord(e2));
          if (ntr) hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(block), 0, 0, rec, S / 16, out);
          else hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(block), 0, 0, rec, S / 16, out);
          CK(hipEventRecord(e3));
          CK(hipEventSynchronize(e3));
          float a, b, c;
          CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2)); CK(hipEventElapsedTime(&c, e2, e3));
          if (it >= 2) { tw.push_back(a); tr.push_back(b); trr.push_back(c); }
        }
        const float w = median(tw), r = median(tr), rr = median(trr);
        printf("{\"case\": \"A\", \"nt_store\": %d, \"nt_load\": %d, \"MB\": %ld, \"write_gbs\": %.0f, \"read_after_write_gbs\": %.0f, \"reread_gbs\": %.0f, \"read_us\": %.1f}\n",
               ntw, ntr, S / MB, S / w / 1e6, S / r / 1e6, S / rr / 1e6, r * 1e3);
        fflush(stdout);
      }

  // ---- B: chunked produce/consume pipeline -----------------------------------------------------
  // lanes = grid*block = 524288; one round = 12*16*lanes = 96 MiB read, 40 MiB written
  const long lanes = (long)grid * block;
  const long total_rounds = src_bytes / (12 * 16 * lanes);  // 128 rounds = 12 GB read, 5 GB written
  for (int ntw = 0; ntw < 2; ++ntw)
    for (int reuse = 0; reuse < 2; ++reuse)
      for (long rounds_per_chunk : {1L, 2L, 3L, 4L, 6L, 8L, 16L, 128L}) {
        if (reuse == 0 && rounds_per_chunk != 1 && rounds_per_chunk != 4 && rounds_per_chunk != 128) continue;
        std::vector<float> tt;
        for (int it = 0; it < 4; ++it) {
          CK(hipEventRecord(e0));
          for (long r0 = 0; r0 < total_rounds; r0 += rounds_per_chunk) {
            const long nr = std::min(rounds_per_chunk, total_rounds - r0);
            const u4* s = src + r0 * 12 * lanes;
            u4* d = reuse ? rec : rec + r0 * 5 * lanes;
            if (ntw) hipLaunchKernelGGL(produce<true>, dim3(grid), dim3(block), 0, 0, s, nr, d);
            else hipLaunchKernelGGL(produce<false>, dim3(grid), dim3(block), 0, 0, s, nr, d);
            hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(block), 0, 0, d, nr * 5 * lanes, out);
          }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float a; CK(hipEventElapsedTime(&a, e0, e1)); if (it >= 1) tt.push_back(a);
        }
        const float t = median(tt);
        printf("{\"case\": \"B\", \"nt_store\": %d, \"reuse_record_buffer\": %d, \"record_MB_per_chunk\": %ld, \"chunks\": %ld, \"ms\": %.3f, \"algorithmic_12GB_gbs\": %.0f, \"moved_22GB_gbs\": %.0f}\n",
               ntw, reuse, rounds_per_chunk * 40, (total_rounds + rounds_per_chunk - 1) / rounds_per_chunk, t, 12.0 * GB / t / 1e6, 22.0 * GB / t / 1e6);
        fflush(stdout);
      }
  // produce alone and consume alone, unchunked (the floor of each half)
  for (int ntw = 0; ntw < 2; ++ntw) {
    std::vector<float> tp, tc;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      if (ntw) hipLaunchKernelGGL(produce<true>, dim3(grid), dim3(block), 0, 0, src, total_rounds, rec);
      else hipLaunchKernelGGL(produce<false>, dim3(grid), dim3(block), 0, 0, src, total_rounds, rec);
      CK(hipEventRecord(e1));
      hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(block), 0, 0, rec, total_rounds * 5 * lanes, out);
      CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
      if (it >= 1) { tp.push_back(a); tc.push_back(b); }
    }
    printf("{\"case\": \"B-halves\", \"nt_store\": %d, \"produce_ms\": %.3f, \"produce_17GB_gbs\": %.0f, \"consume_ms\": %.3f, \"consume_gbs\": %.0f}\n",
           ntw, median(tp), 17.0 * GB / median(tp) / 1e6, median(tc), 5.0 * GB / median(tc) / 1e6);
  }
  return 0;
}
