// mixbw.hip — development micro-benchmark (not part of the product): the ceiling of a kernel that READS three 8-byte
// streams and WRITES one (24 B in, 8 B out per sample: the traffic mix of part_route with packed records), swept over
// launch geometry, loads in flight per lane, load / store cache policy and the size of the store bursts.  Round 2 took
// its "floor" for the C5 routing pass from ONE geometry of mall.hip (grid 2048 x 256, 12 loads + 5 stores per round:
// 5.0-5.1 TB/s); this sweep is what VERDICT r2 "next" #1a asks for.
//   shape R:W   16-byte loads and stores per lane and round (12:4 = the 24:8 mix; 8:8 copy; 12:0 read; 0:4 write)
//   layout      "front": every load instruction of the grid covers one contiguous span (the whole grid sweeps memory as one
//               front); "tile": a workgroup owns a contiguous tile per round (what part_route does: 4096 samples per tile)
//   split       0: every lane loads and stores;  1: one half of the workgroups only reads, the other only writes
//               (same bytes in total) — does mixing directions inside a wave cost anything?
// Build: hipcc --offload-arch=gfx950 -O3 -o mixbw mixbw.hip ; prints one JSON line per configuration.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u4 ld(const u4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u4* p, u4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// L loads per stream (3 streams) and W stores per lane and round.  TILE: workgroup-contiguous tiles.
template <int L, int W, bool NTL, bool NTS, bool TILE>
__global__ void mix(const u4* __restrict__ a, const u4* __restrict__ b, const u4* __restrict__ c, u4* __restrict__ o, long rounds) {
  const long lanes = (long)gridDim.x * blockDim.x;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  u4 acc = {0u, 0u, 0u, 0u};
  for (long r = 0; r < rounds; ++r) {
    u4 v[3][L > 0 ? L : 1];
#pragma unroll
    for (int k = 0; k < L; ++k) {
      long i;
      if (TILE) i = ((r * gridDim.x + blockIdx.x) * L + k) * blockDim.x + threadIdx.x;
      else i = (r * L + k) * lanes + t;
      v[0][k] = ld<NTL>(a + i);
      v[1][k] = ld<NTL>(b + i);
      v[2][k] = ld<NTL>(c + i);
    }
#pragma unroll
    for (int k = 0; k < L; ++k) acc += v[0][k] ^ v[1][k] ^ v[2][k];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      long i;
      if (TILE) i = ((r * gridDim.x + blockIdx.x) * W + k) * blockDim.x + threadIdx.x;
      else i = (r * W + k) * lanes + t;
      u4 ov = acc;
      ov[0] += (unsigned)k;
      st<NTS>(o + i, ov);
    }
  }
  if (W == 0 && acc[0] == 0x12345u && acc[1] == 7u) o[0] = acc;
}

// read only, NS input streams (1: the unweighted headline, 2: C2 / C3, 3: C5), L loads of 16 bytes per stream, lane and round
template <int NS, int L, bool NTL, bool TILE>
__global__ void rd_streams(const u4* __restrict__ a, const u4* __restrict__ b, const u4* __restrict__ c, u4* __restrict__ o, long rounds) {
  const long lanes = (long)gridDim.x * blockDim.x;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const u4* src[3] = {a, b, c};
  u4 acc = {0u, 0u, 0u, 0u};
  for (long r = 0; r < rounds; ++r) {
    u4 v[NS][L];
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const long i = TILE ? ((r * gridDim.x + blockIdx.x) * L + k) * blockDim.x + threadIdx.x : (r * L + k) * lanes + t;
#pragma unroll
      for (int q = 0; q < NS; ++q) v[q][k] = ld<NTL>(src[q] + i);
    }
#pragma unroll
    for (int k = 0; k < L; ++k)
#pragma unroll
      for (int q = 0; q < NS; ++q) acc += v[q][k];
  }
  if (acc[0] == 0x12345u && acc[1] == 7u) o[0] = acc;
}

// split: even workgroups read (twice their share), odd workgroups write (twice their share)
template <int L, int W, bool NTL, bool NTS>
__global__ void mix_split(const u4* __restrict__ a, const u4* __restrict__ b, const u4* __restrict__ c, u4* __restrict__ o, long rounds) {
  const long half = (long)(gridDim.x / 2) * blockDim.x;
  const long t = (long)(blockIdx.x / 2) * blockDim.x + threadIdx.x;
  u4 acc = {1u, 2u, 3u, 4u};
  if ((blockIdx.x & 1) == 0) {
    for (long r = 0; r < rounds; ++r) {
      u4 v[3][2 * L];
#pragma unroll
      for (int k = 0; k < 2 * L; ++k) {
        const long i = (r * 2 * L + k) * half + t;
        v[0][k] = ld<NTL>(a + i);
        v[1][k] = ld<NTL>(b + i);
        v[2][k] = ld<NTL>(c + i);
      }
#pragma unroll
      for (int k = 0; k < 2 * L; ++k) acc += v[0][k] ^ v[1][k] ^ v[2][k];
    }
    if (acc[0] == 0x12345u && acc[1] == 7u) o[0] = acc;
  } else {
    for (long r = 0; r < rounds; ++r) {
#pragma unroll
      for (int k = 0; k < 2 * W; ++k) {
        u4 ov = acc;
        ov[0] += (unsigned)(r + k);
        st<NTS>(o + (r * 2 * W + k) * half + t, ov);
      }
    }
  }
}

// part_route's own access shape.  Workgroup tiles of BLOCK * 4 elements of 8 bytes per stream: PAIR = 1: a lane reads
// 32 contiguous bytes per stream as two 16-byte loads (lane stride 32 B — what part_route does today), PAIR = 0: two
// dense 16-byte loads half a tile apart.  Writes: 8 bytes per element in all, as 16-byte stores (one per lane and two
// elements), scattered in PIECES of `piece` bytes over `streams` output cursors per workgroup (piece = 0: one dense run).
template <bool PAIR, bool NTS>
__global__ void route_shape(const u4* __restrict__ a, const u4* __restrict__ b, const u4* __restrict__ c, u4* __restrict__ o, long tiles_per_wg,
                            int piece16, int streams, long stream_stride16) {
  const int tid = threadIdx.x, B = blockDim.x;
  u4 acc = {0u, 0u, 0u, 0u};
  // this workgroup's output streams: stream s of workgroup g starts at (g * streams + s) * stream_stride16
  const long wg_out = (long)blockIdx.x * streams * stream_stride16;
  long cursor = 0;  // 16-byte units written so far to EVERY stream of this workgroup (kept equal across streams)
  for (long k = 0; k < tiles_per_wg; ++k) {
    const long tile = ((long)blockIdx.x + k * gridDim.x) * (2L * B);  // 16-byte units per stream and tile: 2 per lane
    u4 v[3][2];
    const long i0 = PAIR ? tile + 2L * tid : tile + tid;
    const long i1 = PAIR ? i0 + 1 : i0 + B;
    v[0][0] = __builtin_nontemporal_load(a + i0); v[0][1] = __builtin_nontemporal_load(a + i1);
    v[1][0] = __builtin_nontemporal_load(b + i0); v[1][1] = __builtin_nontemporal_load(b + i1);
    v[2][0] = __builtin_nontemporal_load(c + i0); v[2][1] = __builtin_nontemporal_load(c + i1);
    acc += v[0][0] ^ v[1][0] ^ v[2][0] ^ v[0][1] ^ v[1][1] ^ v[2][1];
    // 2 stores per lane and tile: 2 * B units per tile per workgroup
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long u = (long)j * B + tid;  // unit inside the tile's output
      long dst;
      if (piece16 == 0) {
        dst = ((long)blockIdx.x * tiles_per_wg + k) * (2L * B) + u;
      } else {
        const long pc = u / piece16, within = u % piece16;  // piece pc goes to stream pc % streams
        const long s = pc % streams, nth = pc / streams;
        const long per_tile = (2L * B) / piece16 / streams;  // pieces per stream and tile
        dst = wg_out + s * stream_stride16 + (k * per_tile + nth) * piece16 + within;
      }
      u4 ov = acc;
      ov[0] += (unsigned)j;
      st<NTS>(o + dst, ov);
    }
  }
  (void)cursor;
}

// route_phases: route_shape's traffic with part_route's PHASE structure added step by step, to see which step costs the
// 12 % between the bare traffic (5.3 TB/s) and the real pass (4.75):
//   MODE 0  load tile -> use -> store tile                       (= route_shape, PAIR loads, 64 streams of 512-byte pieces)
//   MODE 1  + __syncthreads() between load, use and store        (the workgroup moves in lockstep)
//   MODE 2  + the tile's records go through LDS (write, barrier, read back by another lane) before they are stored
//   MODE 3  + late prefetch: the NEXT tile's loads are issued before this tile's LDS round trip and waited for ahead of the stores
//   MODE 4  = MODE 3 + returning LDS atomics (4 per lane, 64 counters) ahead of the prefetch, a 64-lane scan phase between barriers
template <int MODE>
__global__ void route_phases(const u4* __restrict__ a, const u4* __restrict__ b, const u4* __restrict__ c, u4* __restrict__ o, long tiles_per_wg,
                             int piece16, int streams, long stream_stride16) {
  extern __shared__ u4 lds[];
  unsigned* cnt = reinterpret_cast<unsigned*>(lds + 2 * blockDim.x + 64);
  const int tid = threadIdx.x, B = blockDim.x;
  const long wg_out = (long)blockIdx.x * streams * stream_stride16;
  const long per_tile = (2L * B) / piece16 / streams;
  auto tile_at = [&](long k) { return ((long)blockIdx.x + k * gridDim.x) * (2L * B); };
  u4 v[3][2], n[3][2];
  auto load = [&](long k, u4 (&r)[3][2]) {
    const long i0 = tile_at(k) + 2L * tid, i1 = i0 + 1;
    r[0][0] = __builtin_nontemporal_load(a + i0); r[0][1] = __builtin_nontemporal_load(a + i1);
    r[1][0] = __builtin_nontemporal_load(b + i0); r[1][1] = __builtin_nontemporal_load(b + i1);
    r[2][0] = __builtin_nontemporal_load(c + i0); r[2][1] = __builtin_nontemporal_load(c + i1);
  };
  if (MODE >= 4) { if (tid < 64) cnt[tid] = 0; __syncthreads(); }
  load(0, v);
  for (long k = 0; k < tiles_per_wg; ++k) {
    if (MODE < 3 && k > 0) load(k, v);
    u4 rec0 = v[0][0] ^ v[1][0] ^ v[2][0], rec1 = v[0][1] ^ v[1][1] ^ v[2][1];
    unsigned rk = 0;
    if (MODE >= 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rk += atomicAdd(cnt + ((rec0[q] + tid * 7 + q * 13) & 63), 1u);
    }
    if (MODE >= 3) { const long kn = k + 1 < tiles_per_wg ? k + 1 : k; load(kn, n); }
    if (MODE >= 1) __syncthreads();
    if (MODE >= 4) {
      if (tid < 64) {  // a scan over the 64 counters by one wavefront, as the block layout of part_route
        unsigned x = cnt[tid];
        for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off, 64); x += (tid >= off) ? y : 0u; }
        cnt[tid] = x & 0u;  // (and cleared for the next tile)
      }
      __syncthreads();
    }
    if (MODE >= 2) {
      rec0[1] += rk;
      lds[tid] = rec0; lds[B + tid] = rec1;
      __syncthreads();
      const int src = (tid * 17 + 5) % B;  // another lane's records: the sort
      rec0 = lds[src]; rec1 = lds[B + src];
    }
    if (MODE >= 3) {
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) { asm volatile("" : "+v"(n[s3][0])); asm volatile("" : "+v"(n[s3][1])); v[s3][0] = n[s3][0]; v[s3][1] = n[s3][1]; }
    }
    if (MODE >= 1) __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long u = (long)j * B + tid;
      const long pc = u / piece16, within = u % piece16;
      const long sidx = pc % streams, nth = pc / streams;
      const long dst = wg_out + sidx * stream_stride16 + (k * per_tile + nth) * piece16 + within;
      __builtin_nontemporal_store(j ? rec1 : rec0, o + dst);
    }
  }
}

static float median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

typedef void (*kern_t)(const u4*, const u4*, const u4*, u4*, long);

struct Variant { const char* layout; int L, W, ntl, nts; kern_t k; int split; };

#define V(LAY, TILE_, L_, W_, NTL_, NTS_) {LAY, L_, W_, NTL_, NTS_, (kern_t)mix<L_, W_, NTL_, NTS_, TILE_>, 0}
#define VS(L_, W_, NTL_, NTS_) {"front", L_, W_, NTL_, NTS_, (kern_t)mix_split<L_, W_, NTL_, NTS_>, 1}

int main(int argc, char** argv) {
  const long GB = 1L << 30;
  const long stream_bytes = 4 * GB;  // per input stream: 12 GB read in all for L:W = 3:1
  u4 *a, *b, *c, *o;
  // MIXBW_SKEW=<bytes>: the four streams are carved out of ONE allocation, stream k starting k * (4 GiB + skew) into it —
  // separate hipMallocs of 4 GiB put the streams exactly 4 GiB apart, i.e. the same offset of every stream on the same
  // memory channel at the same time; the skew shows how much of a "ceiling" is that aliasing
  const char* skew_env = getenv("MIXBW_SKEW");
  const long skew = skew_env ? atol(skew_env) : -1;
  if (skew >= 0) {
    char* base;
    CK(hipMalloc(&base, 4 * (stream_bytes + skew) + (2L << 20)));
    a = (u4*)base; b = (u4*)(base + (stream_bytes + skew)); c = (u4*)(base + 2 * (stream_bytes + skew)); o = (u4*)(base + 3 * (stream_bytes + skew));
  } else {
  CK(hipMalloc(&a, stream_bytes)); CK(hipMalloc(&b, stream_bytes)); CK(hipMalloc(&c, stream_bytes)); CK(hipMalloc(&o, stream_bytes + (1L << 20)));
  }
  printf("{\"skew\": %ld, \"a\": \"%p\", \"b\": \"%p\", \"c\": \"%p\", \"o\": \"%p\"}\n", skew, (void*)a, (void*)b, (void*)c, (void*)o);
  CK(hipMemset(a, 0x11, stream_bytes)); CK(hipMemset(b, 0x22, stream_bytes)); CK(hipMemset(c, 0x33, stream_bytes)); CK(hipMemset(o, 0, stream_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<Variant> vars = {
      // the 24:8 mix, front layout: loads in flight per lane 3 x {1, 2, 4} x 16 B
      V("front", false, 1, 1, true, true), V("front", false, 1, 1, true, false), V("front", false, 1, 1, false, false), V("front", false, 1, 1, false, true),
      V("front", false, 2, 2, true, true), V("front", false, 2, 2, true, false), V("front", false, 2, 2, false, false), V("front", false, 2, 2, false, true),
      V("front", false, 4, 4, true, true), V("front", false, 4, 4, true, false), V("front", false, 4, 4, false, false), V("front", false, 4, 4, false, true),
      // tile layout (workgroup-contiguous)
      V("tile", true, 1, 1, true, true), V("tile", true, 1, 1, true, false),
      V("tile", true, 2, 2, true, true), V("tile", true, 2, 2, true, false),
      V("tile", true, 4, 4, true, true), V("tile", true, 4, 4, true, false),
      // bigger workgroup tiles: is it the 64 KiB contiguous per stream, workgroup and round that makes (tile, 4, 1024 threads) special?
      V("tile", true, 8, 8, true, true), V("tile", true, 16, 16, true, true), V("tile", true, 3, 3, true, true), V("tile", true, 6, 6, true, true),
      // direction split between workgroups
      VS(1, 1, true, true), VS(1, 1, true, false), VS(2, 2, true, true), VS(2, 2, true, false),
      // calibration: read only (3 streams), write only, 1:1 copy traffic (3 in, 3 out per lane)
      V("front", false, 2, 0, true, true), V("front", false, 4, 0, true, true), V("front", false, 4, 0, false, true),
      V("front", false, 0, 4, true, true), V("front", false, 0, 4, true, false),
      V("front", false, 1, 3, true, true), V("front", false, 1, 3, true, false), V("front", false, 2, 6, true, false),
  };
  const int grids[] = {256, 512, 1024, 2048, 4096, 8192};
  const int blocks[] = {256, 512, 1024};
  const bool only_route = argc > 1 && (argv[1][0] == 'r' || argv[1][0] == 's' || argv[1][0] == 'p');
  const bool quick_mix = argc > 1 && argv[1][0] == 'm';  // the 24:8 mix only, nt loads
  if (!only_route)
  for (const Variant& v : vars)
    for (int block : blocks)
      for (int grid : grids) {
        if ((long)grid * block > 4L * 1024 * 1024) continue;
        if (quick_mix && !(v.L == v.W && v.L > 0 && v.ntl && v.split == 0 && grid <= 2048)) continue;
        if (argc > 1 && argv[1][0] == 'm' && argv[1][1] == 'b' && !(v.L == v.W && v.L >= 3 && v.nts && v.layout[0] == 't')) continue;
        const long lanes = (long)grid * block;
        const int unit = std::max(v.L, v.W);
        const long rounds = stream_bytes / (16L * unit * lanes);
        if (rounds < 1) continue;
        const double rd = 3.0 * 16 * v.L * lanes * rounds, wr = 16.0 * v.W * lanes * rounds;
        std::vector<float> tt;
        for (int it = 0; it < 6; ++it) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(v.k, dim3(grid), dim3(block), 0, 0, a, b, c, o, rounds);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (it >= 2) tt.push_back(ms);
        }
        const float t = median(tt);
        printf("{\"layout\": \"%s\", \"split\": %d, \"loads_per_stream\": %d, \"stores\": %d, \"nt_load\": %d, \"nt_store\": %d, \"grid\": %d, \"block\": %d, "
               "\"rounds\": %ld, \"read_GB\": %.2f, \"write_GB\": %.2f, \"ms\": %.3f, \"moved_gbs\": %.0f}\n",
               v.layout, v.split, v.L, v.W, v.ntl, v.nts, grid, block, rounds, rd / 1e9, wr / 1e9, t, (rd + wr) / t / 1e6);
        fflush(stdout);
      }
  // ---- read-only ceilings by number of streams (argument "s") ---------------------------------------
  if (quick_mix) return 0;
  if (argc > 1 && argv[1][0] == 's') {
    struct RV { int ns, L, ntl, tile; kern_t k; };
#define R(NS_, L_, NTL_, T_) {NS_, L_, NTL_, T_, (kern_t)rd_streams<NS_, L_, NTL_, T_>}
    std::vector<RV> rv = {R(1, 1, true, false), R(1, 2, true, false), R(1, 4, true, false), R(1, 8, true, false), R(1, 4, true, true), R(1, 8, true, true), R(1, 4, false, false),
                          R(2, 1, true, false), R(2, 2, true, false), R(2, 4, true, false), R(2, 8, true, false), R(2, 4, true, true), R(2, 2, true, true), R(2, 4, false, false),
                          R(3, 1, true, false), R(3, 2, true, false), R(3, 4, true, false), R(3, 2, true, true)};
    for (const RV& v : rv)
      for (int block : {256, 512, 1024})
        for (int grid : {256, 512, 768, 1024, 2048, 4096}) {
          if ((long)grid * block > 2L * 1024 * 1024) continue;
          const long lanes = (long)grid * block;
          const long rounds = stream_bytes / (16L * v.L * lanes);
          if (rounds < 1) continue;
          std::vector<float> tt;
          for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.k, dim3(grid), dim3(block), 0, 0, a, b, c, o, rounds);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) tt.push_back(ms);
          }
          const float t = median(tt);
          const double rd = (double)v.ns * 16 * v.L * lanes * rounds;
          printf("{\"layout\": \"%s\", \"streams\": %d, \"loads_per_stream\": %d, \"nt_load\": %d, \"grid\": %d, \"block\": %d, \"KB_in_flight_per_cu\": %.0f, "
                 "\"read_GB\": %.2f, \"ms\": %.3f, \"read_gbs\": %.0f}\n",
                 v.tile ? "tile" : "front", v.ns, v.L, v.ntl, grid, block, (double)v.ns * 16 * v.L * lanes / 256 / 1024, rd / 1e9, t, rd / t / 1e6);
          fflush(stdout);
        }
    return 0;
  }
  // ---- part_route's phase structure, step by step (argument "p") ----------------------------------
  if (argc > 1 && argv[1][0] == 'p') {
    const long units_per_stream = stream_bytes / 16;
    for (int rep = 0; rep < 2; ++rep)
      for (int block : {512, 1024})
        for (int mode = 0; mode <= 4; ++mode) {
          const int per_cu = 1024 / block, grid = 256 * per_cu, streams = 64;
          const long tiles_per_wg = units_per_stream / (2L * block) / grid;
          const int piece16 = (2 * block) / streams;
          const long stream_stride16 = tiles_per_wg * piece16;
          const size_t lds_bytes = (size_t)(2 * block + 64) * 16 + 256;
          std::vector<float> tt;
          for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            switch (mode) {
              case 0: hipLaunchKernelGGL(route_phases<0>, dim3(grid), dim3(block), lds_bytes, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16); break;
              case 1: hipLaunchKernelGGL(route_phases<1>, dim3(grid), dim3(block), lds_bytes, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16); break;
              case 2: hipLaunchKernelGGL(route_phases<2>, dim3(grid), dim3(block), lds_bytes, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16); break;
              case 3: hipLaunchKernelGGL(route_phases<3>, dim3(grid), dim3(block), lds_bytes, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16); break;
              default: hipLaunchKernelGGL(route_phases<4>, dim3(grid), dim3(block), lds_bytes, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) tt.push_back(ms);
          }
          const float t = median(tt);
          const double rd = 3.0 * 16 * 2.0 * block * tiles_per_wg * grid, wr = 16.0 * 2.0 * block * tiles_per_wg * grid;
          printf("{\"layout\": \"route_phases\", \"mode\": %d, \"block\": %d, \"wg_per_cu\": %d, \"read_GB\": %.2f, \"write_GB\": %.2f, \"ms\": %.3f, \"moved_gbs\": %.0f}\n",
                 mode, block, per_cu, rd / 1e9, wr / 1e9, t, (rd + wr) / t / 1e6);
          fflush(stdout);
        }
    return 0;
  }
  // ---- part_route's access shape ----------------------------------------------------------------
  {
    const long units_per_stream = stream_bytes / 16;  // 16-byte units per input stream
    for (int block : {512, 1024})
      for (int per_cu : {1, 2})
        for (int pair = 0; pair < 2; ++pair)
          for (int nts = 0; nts < 2; ++nts)
            for (int streams : {0, 8, 16, 32, 64, 128, 256}) {  // 0: one dense output run per workgroup; else one piece per stream and tile
              if (argc > 1 && argv[1][1] == 'q' && !(block == 512 && per_cu == 1 && pair == 1 && nts == 1 && streams == 64)) continue;  // (rq: one shape, for counter runs)
              const int grid = 256 * per_cu * (1024 / block);
              const long tiles = units_per_stream / (2L * block);
              const long tiles_per_wg = tiles / grid;
              const int piece16 = streams ? (2 * block) / streams : 0;
              const int piece = piece16 * 16;
              const long stream_stride16 = piece16 ? tiles_per_wg * piece16 : 0;
              if (piece16 && (long)grid * streams * stream_stride16 * 16 > stream_bytes) continue;
              kern_t k = pair ? (nts ? (kern_t)nullptr : (kern_t)nullptr) : (kern_t)nullptr;
              (void)k;
              std::vector<float> tt;
              for (int it = 0; it < 6; ++it) {
                CK(hipEventRecord(e0));
                if (pair && nts) hipLaunchKernelGGL((route_shape<true, true>), dim3(grid), dim3(block), 0, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16);
                else if (pair) hipLaunchKernelGGL((route_shape<true, false>), dim3(grid), dim3(block), 0, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16);
                else if (nts) hipLaunchKernelGGL((route_shape<false, true>), dim3(grid), dim3(block), 0, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16);
                else hipLaunchKernelGGL((route_shape<false, false>), dim3(grid), dim3(block), 0, 0, a, b, c, o, tiles_per_wg, piece16, streams, stream_stride16);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) tt.push_back(ms);
              }
              const float t = median(tt);
              const double rd = 3.0 * 16 * 2.0 * block * tiles_per_wg * grid, wr = 16.0 * 2.0 * block * tiles_per_wg * grid;
              printf("{\"layout\": \"route_shape\", \"block\": %d, \"wg_per_cu\": %d, \"lane_pair_loads\": %d, \"nt_store\": %d, \"piece_bytes\": %d, "
                     "\"streams_per_wg\": %d, \"read_GB\": %.2f, \"write_GB\": %.2f, \"ms\": %.3f, \"moved_gbs\": %.0f}\n",
                     block, per_cu * (1024 / block), pair, nts, piece, streams, rd / 1e9, wr / 1e9, t, (rd + wr) / t / 1e6);
              fflush(stdout);
            }
  }
  return 0;
}
