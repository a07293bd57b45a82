// xhist_partition.hip.h — partitioned mode for histograms that do not fit the 160 KiB LDS of a CU
// (BASELINE C5: 1024 x 1024 float64 bins = 8 MiB).
//
// Why: device-scope atomics on MI355X execute at the memory side and serialise per cache line
// (~12 ns each; measured 2.4e10 atomics/s for C5's distribution), a hard ceiling ~15x below what
// the sample stream could feed.  LDS atomics have no such ceiling, so the bins are cut into
// partitions of 2^shift bins that DO fit LDS, and the samples are routed to their partition first:
//
//   pass 0  part_count          digitize; count samples per (workgroup, partition); spill   reads samples,
//                               the flat bin index of every sample (u32)                   writes 4 B/sample
//   prefix  part_prefix         exclusive scan -> offsets[p], base[workgroup][p]           tiny
//   pass A  part_scatter        read (flat, weight), sort each tile by partition in LDS,   reads 4 (+8) B/sample,
//                               write (code = bin mod 2^shift : u16, weight : f64) records  writes 2 (+8) B/sample
//                               into this workgroup's private slice of each partition stream
//   pass B  part_accumulate     stream the records (contiguous per partition, equal share    reads 2 (+8) B/sample
//                               per workgroup), ds_add into a 2^shift-bin LDS histogram,
//                               flush to the output at partition boundaries
//
// No global atomics on the sample path (pass A's slots come from pass 0's counts, so the record
// order is deterministic); HBM traffic for C5 = 16+4 | 4+8+10 | 10 = 52 B/sample instead of 24, which
// bounds this mode at ~0.46 of the streaming rate — against 0.07 for global atomics.
// Same tile->workgroup assignment in pass 0 and pass A (same grid, 8192-sample tiles) is what
// makes the counts valid slot reservations.
#pragma once

#include "xhist_kernels.hip.h"

namespace xhist {

// samples per workgroup tile, identical in the counting and the scatter pass (1024 threads x 8;
// within-box A/B against 512 x 8: 6.32 vs 6.65 ms for C5 — longer partition runs per store burst)
constexpr int kPartBlock = 1024, kPartTile = 8192;

// pass 0: digitize once; count kept samples per (workgroup, partition) and spill every sample's
// flat bin index (0xFFFFFFFF = dropped) so that pass A needs neither the samples nor the tables.
template <typename ST, int D, int VEC, int SCAN>
__global__ void __launch_bounds__(kPartBlock) part_count(const Params p, uint32_t* __restrict__ flat_out) {
  constexpr int UNROLL = kPartTile / kPartBlock / VEC;
  constexpr int CMP = __is_same(ST, float) ? 2 : 0;
  using CT = typename Dom<CMP>::T;
  using svec = typename VecOf<ST, VEC>::type;
  using fvec = typename VecOf<uint32_t, VEC>::type;

  const int tid = threadIdx.x;
  const int P = p.n_parts;
  const uint64_t* tab = stage_tables(p);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(xhist_smem + (size_t)p.table_words * 8);  // [(P+1) << 5]
  constexpr int kCl2 = 5;  // one copy per lane bank, as in the LDS histogram mode
  const uint32_t mycopy = (uint32_t)tid & 31u;
  for (int i = tid; i < ((P + 1) << kCl2); i += blockDim.x) cnt[i] = 0u;
  __syncthreads();

  const ST* sp[D];
#pragma unroll
  for (int d = 0; d < D; ++d) sp[d] = reinterpret_cast<const ST*>(p.s_ptr[d]);
  int max_steps = 1;
#pragma unroll
  for (int d = 0; d < D; ++d) max_steps = max(max_steps, p.dim[d].steps);

  const int64_t n_tiles = (p.n_cols + kPartTile - 1) / kPartTile;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t base = tile * kPartTile;
    svec xv[D][UNROLL];
    if (base + kPartTile <= p.n_cols) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + ((int64_t)u * kPartBlock + tid) * VEC;
#pragma unroll
        for (int d = 0; d < D; ++d) xv[d][u] = __builtin_nontemporal_load(reinterpret_cast<const svec*>(sp[d] + i));
      }
    } else {  // ragged last tile: positions past the end become NaN samples, which digitize drops
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int64_t i = base + ((int64_t)u * kPartBlock + tid) * VEC + v;
#pragma unroll
          for (int d = 0; d < D; ++d) xv[d][u][v] = (i < p.n_cols) ? sp[d][i] : (ST)__builtin_nanf("");
        }
    }

    uint32_t cntle[D][UNROLL][VEC];
    count_le_tile<CMP, SCAN, D, UNROLL, VEC>(xv, p, tab, max_steps, cntle);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      fvec fo;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        bool ok = true;
        uint32_t flat = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const int b = bin_from_count<CMP>((CT)xv[d][u][v], p.dim[d], cntle[d][u][v]);
          ok &= (b >= 0);
          flat = (d == 0) ? (uint32_t)b : flat * (uint32_t)p.dim[d].nb + (uint32_t)b;
        }
        const uint32_t part = ok ? (flat >> p.part_shift) : (uint32_t)P;  // P = trash counter
        atomicAdd(cnt + ((part << kCl2) + mycopy), 1u);
        fo[v] = ok ? flat : 0xffffffffu;
      }
      // the spill buffer is padded to whole tiles, so the ragged tile needs no store guard
      __builtin_nontemporal_store(fo, reinterpret_cast<fvec*>(flat_out + base + ((int64_t)u * kPartBlock + tid) * VEC));
    }
  }

  __syncthreads();
  for (int q = tid; q < P; q += blockDim.x) {
    uint32_t s = 0;
    for (int c = 0; c < 32; ++c) s += cnt[(q << kCl2) + ((c + tid) & 31)];
    p.part_counts[(size_t)blockIdx.x * P + q] = s;
  }
}

// pass A: read (flat bin, weight), sort each 8192-sample tile by partition in LDS, append every
// partition's run to this workgroup's private slice of that partition's record stream.
template <typename WT>
__global__ void __launch_bounds__(kPartBlock) part_scatter(const uint32_t* __restrict__ flat, const void* wv_, int64_t n,
                                                            const uint64_t* __restrict__ base_tbl, uint16_t* __restrict__ codes,
                                                            double* __restrict__ wrec, int shift, int P) {
  constexpr bool kWeighted = !__is_same(WT, NoWeight);
  using wscalar = typename std::conditional<kWeighted, WT, float>::type;
  static_assert(kPartTile == kPartBlock * 8, "8 samples per thread, as 2 groups of 4 consecutive");
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  // the caller's weights are only guaranteed element-aligned here (gfx950 vector loads need no more)
  typedef wscalar w4 __attribute__((ext_vector_type(4), aligned(sizeof(wscalar))));
  struct Rec { double w; uint32_t key; uint32_t pad; };  // key = part << 16 | code

  const int tid = threadIdx.x;
  uint32_t* cnt = reinterpret_cast<uint32_t*>(xhist_smem);                  // [P] rank counters
  uint64_t* gb = reinterpret_cast<uint64_t*>(xhist_smem + 1024);            // [P] next free slot per stream
  uint64_t* delta = gb + 256;                                               // [P] gb - exclusive scan
  uint32_t* scan = reinterpret_cast<uint32_t*>(delta + 256);               // [P]
  unsigned char* stage = xhist_smem + 1024 + 2048 + 2048 + 1024;           // 6 KiB of tables, then records
  Rec* srec = reinterpret_cast<Rec*>(stage);
  uint32_t* skey = reinterpret_cast<uint32_t*>(stage);                      // unweighted: 4-byte records
  const wscalar* wp = reinterpret_cast<const wscalar*>(wv_);
  const uint32_t code_mask = (1u << shift) - 1u;
  for (int i = tid; i < P; i += blockDim.x) {
    cnt[i] = 0u;
    gb[i] = base_tbl[(size_t)blockIdx.x * P + i];
  }
  __syncthreads();

  const int64_t n_tiles = (n + kPartTile - 1) / kPartTile;
  auto load_tile = [&](int64_t tile, u4 (&f)[2], w4 (&w)[2]) {
    const int64_t base = tile * kPartTile;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t i = base + ((int64_t)u * kPartBlock + tid) * 4;
      f[u] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(flat + i));  // padded: always in bounds
      if (kWeighted) {
        if (base + kPartTile <= n) {
          w[u] = __builtin_nontemporal_load(reinterpret_cast<const w4*>(wp + i));
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v) w[u][v] = (i + v < n) ? wp[i + v] : (wscalar)0;
        }
      }
    }
  };
  u4 f[2], fn[2];
  w4 w[2], wn[2];
  if ((int64_t)blockIdx.x < n_tiles) load_tile(blockIdx.x, f, w);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // the next tile's loads are in flight across this tile's barrier-separated phases
    if (tile + gridDim.x < n_tiles) load_tile(tile + gridDim.x, fn, wn);
    uint32_t rank[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) rank[u][v] = (f[u][v] != 0xffffffffu) ? atomicAdd(cnt + (f[u][v] >> shift), 1u) : 0u;
    __syncthreads();
    if (tid < P) {  // exclusive scan of the P counters (broadcast reads)
      uint32_t s = 0;
      for (int q = 0; q < tid; ++q) s += cnt[q];
      scan[tid] = s;
      delta[tid] = gb[tid] - s;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (f[u][v] != 0xffffffffu) {
          const uint32_t part = f[u][v] >> shift;
          const uint32_t slot = scan[part] + rank[u][v];
          const uint32_t key = (part << 16) | (f[u][v] & code_mask);
          if (kWeighted) {
            Rec r;
            r.w = (double)w[u][v];
            r.key = key;
            r.pad = 0;
            srec[slot] = r;
          } else {
            skey[slot] = key;
          }
        }
    __syncthreads();
    const uint32_t total = scan[P - 1] + cnt[P - 1];
    for (uint32_t j = tid; j < total; j += blockDim.x) {  // consecutive lanes -> consecutive slots of a run
      uint32_t key;
      double wj = 0.0;
      if (kWeighted) {
        const Rec r = srec[j];
        key = r.key;
        wj = r.w;
      } else {
        key = skey[j];
      }
      const uint64_t dst = delta[key >> 16] + j;
      __builtin_nontemporal_store((uint16_t)(key & 0xffffu), codes + dst);
      if (kWeighted) __builtin_nontemporal_store(wj, wrec + dst);
    }
    __syncthreads();
    if (tid < P) {
      gb[tid] += cnt[tid];
      cnt[tid] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f[u] = fn[u];
      w[u] = wn[u];
    }
  }
}

// counts[G][P] -> offsets[P+1] (start of each partition stream) and base[G][P] (start of each
// workgroup's slice inside it).  One workgroup of 1024 threads arranged as R row groups x P columns
// so that every global access is coalesced along P and each thread walks only G/R rows.
__global__ void __launch_bounds__(1024) part_prefix(const uint32_t* counts, int G, int P, uint64_t* offsets, uint64_t* base) {
  __shared__ uint64_t part[1024];  // [R][P] partial sums, then exclusive prefixes over r
  __shared__ uint64_t off[257];
  const int t = threadIdx.x;
  const int R = max(1, (int)blockDim.x / P);
  const int col = t % P, r = t / P;
  const bool active = r < R;
  const int g0 = active ? (int)((int64_t)G * r / R) : 0, g1 = active ? (int)((int64_t)G * (r + 1) / R) : 0;
  if (active) {
    uint64_t s = 0;
    for (int g = g0; g < g1; ++g) s += counts[(size_t)g * P + col];
    part[r * P + col] = s;
  }
  __syncthreads();
  if (t < P) {  // exclusive prefix over the row groups of column t; total of the column
    uint64_t run = 0;
    for (int k = 0; k < R; ++k) {
      const uint64_t v = part[k * P + t];
      part[k * P + t] = run;
      run += v;
    }
    off[t + 1] = run;  // column total, turned into offsets below
  }
  __syncthreads();
  if (t == 0) {
    uint64_t run = 0;
    for (int q = 0; q < P; ++q) {
      const uint64_t v = off[q + 1];
      off[q] = run;
      run += v;
    }
    off[P] = run;
  }
  __syncthreads();
  if (t <= P) offsets[t] = off[t];
  if (active) {
    uint64_t run = off[col] + part[r * P + col];
    for (int g = g0; g < g1; ++g) {
      base[(size_t)g * P + col] = run;
      run += counts[(size_t)g * P + col];
    }
  }
}

// pass B: every workgroup takes an equal share of the concatenated record streams (so the load is
// balanced whatever the distribution), accumulates in a 2^shift-bin LDS histogram and flushes it
// to the output whenever its range crosses into the next partition.
template <bool WEIGHTED>
__global__ void __launch_bounds__(1024) part_accumulate(const uint16_t* codes, const double* wrec, const uint64_t* offsets,
                                                         void* out_v, int64_t n_bins, int shift, int P) {
  using lds_t = typename std::conditional<WEIGHTED, double, uint32_t>::type;
  using out_t = typename std::conditional<WEIGHTED, double, unsigned long long>::type;
  lds_t* hist = reinterpret_cast<lds_t*>(xhist_smem);
  out_t* out = reinterpret_cast<out_t*>(out_v);
  const uint32_t bpp = 1u << shift;
  const int tid = threadIdx.x;
  for (uint32_t c = tid; c < bpp; c += blockDim.x) hist[c] = (lds_t)0;
  const uint64_t total = offsets[P];
  uint64_t lo = total / gridDim.x * blockIdx.x + min((uint64_t)blockIdx.x, total % gridDim.x);
  const uint64_t hi = lo + total / gridDim.x + (blockIdx.x < total % gridDim.x ? 1 : 0);
  int part = 0;
  while (part + 1 < P && offsets[part + 1] <= lo) ++part;
  __syncthreads();
  while (lo < hi) {
    while (part + 1 < P && offsets[part + 1] <= lo) ++part;
    const uint64_t pend = min(hi, offsets[part + 1]);
    // Records of one partition are contiguous: after a scalar head up to a 4-record boundary
    // every lane takes 4 consecutive records per load (8-byte code quad, 2 x 16-byte weight
    // pairs), 2 such groups in flight; a scalar tail finishes the range.
    uint64_t i = lo;
    const uint64_t head_end = min(pend, (lo + 3) & ~(uint64_t)3);
    for (uint64_t j = i + tid; j < head_end; j += blockDim.x) {
      const uint16_t c = codes[j];
      if (WEIGHTED) unsafeAtomicAdd(reinterpret_cast<double*>(hist) + c, wrec[j]);
      else atomicAdd(reinterpret_cast<uint32_t*>(hist) + c, 1u);
    }
    i = head_end;
    typedef uint16_t c4 __attribute__((ext_vector_type(4)));
    typedef double w2 __attribute__((ext_vector_type(2)));
    constexpr int kGroups = 2;
    const uint64_t step = (uint64_t)blockDim.x * 4 * kGroups;
    for (; i + step <= pend; i += step) {
      c4 cv[kGroups];
      w2 wa[kGroups], wb[kGroups];
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        const uint64_t j = i + ((uint64_t)g * blockDim.x + tid) * 4;
        cv[g] = __builtin_nontemporal_load(reinterpret_cast<const c4*>(codes + j));
        if (WEIGHTED) {
          wa[g] = __builtin_nontemporal_load(reinterpret_cast<const w2*>(wrec + j));
          wb[g] = __builtin_nontemporal_load(reinterpret_cast<const w2*>(wrec + j + 2));
        }
      }
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (WEIGHTED) unsafeAtomicAdd(reinterpret_cast<double*>(hist) + cv[g][k], k < 2 ? wa[g][k] : wb[g][k - 2]);
          else atomicAdd(reinterpret_cast<uint32_t*>(hist) + cv[g][k], 1u);
        }
    }
    for (uint64_t j = i + tid; j < pend; j += blockDim.x) {
      const uint16_t c = codes[j];
      if (WEIGHTED) unsafeAtomicAdd(reinterpret_cast<double*>(hist) + c, wrec[j]);
      else atomicAdd(reinterpret_cast<uint32_t*>(hist) + c, 1u);
    }
    __syncthreads();
    for (uint32_t c = tid; c < bpp; c += blockDim.x) {
      const lds_t v = hist[c];
      if (v != (lds_t)0) {
        const int64_t bin = ((int64_t)part << shift) + c;
        if (bin < n_bins) {
          if (WEIGHTED) unsafeAtomicAdd(reinterpret_cast<double*>(out) + bin, (double)v);
          else atomicAdd(reinterpret_cast<unsigned long long*>(out) + bin, (unsigned long long)v);
        }
        hist[c] = (lds_t)0;
      }
    }
    __syncthreads();
    lo = pend;
  }
}

}  // namespace xhist
