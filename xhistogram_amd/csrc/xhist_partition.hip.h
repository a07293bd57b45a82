// xhist_partition.hip.h — partitioned mode for histograms that do not fit the 160 KiB LDS of a CU
// (BASELINE C5: 1024 x 1024 float64 bins = 8 MiB).
//
// Why: global atomics on MI355X run at a flat ~2.4e10 (f64) / 2.7e10 (u32) per second for the whole
// chip — whatever the scope (agent, workgroup, wavefront), whether the table is 64 KiB or 8 MiB,
// one copy or one per XCD (tools/ubench/l2atomic.hip, profiles/r01_l_l2atomic_scopes.jsonl) — a
// hard ceiling ~15x below what the sample stream could feed.  LDS atomics have no such ceiling, so the bins are cut into
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
// bounds this mode at ~0.46 of the streaming rate — against 0.07 for global atomics.  Passes 0 and A
// mix reads and writes and run at 5.3 and 4.7 TB/s (a pure read stream reaches 7.0).
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
  constexpr int CMP = (__is_same(ST, float) && SCAN != kScanArith) ? 2 : 0;
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
//
// Stores are what this pass is made of, and on gfx950 a wavefront's global store costs about the
// same (~60 cycles of the CU's store path) whether each lane writes 2 or 16 bytes (measured here:
// 2-byte code stores 0.75 ms, 8-byte weight stores 0.8 ms, for 1 GB and 4 GB).  So every store is a
// 16-byte one: records leave in aligned GROUPS of GRP (8 codes = 16 B; 2 weights = 16 B), which
// needs every run to start on a group boundary of its stream:
//   * part_prefix rounds each (workgroup, partition) slice up to whole groups, so slices start aligned;
//   * inside a workgroup, the < GRP records of a partition that do not fill a group at the end of a
//     tile are CARRIED in LDS into that partition's run of the next tile;
//   * after its last tile the workgroup pads each partition's last group with neutral records
//     (weight 0 / the trash code 1 << shift), which is exactly the slack part_prefix reserved.
// LDS: counters and cursors, the carried records, and the sorted tile as two arrays (key u32,
// weight f64) of 8192 + 2 (GRP - 1) P slots — a partition's block starts on a group boundary and
// holds [carried | new] records.
constexpr int kScatterTableBytes = 12 * 1024;
// part_scatter may sort SUB-tiles of the counting pass's 8192-sample tiles (carried records make
// the store width independent of the run length).  Within-box A/B for C5: 4096-sample sub-tiles
// with two workgroups per CU 5.16-5.21 ms against 4.85 ms for whole tiles, one workgroup per CU.
constexpr int kScatterTile = 8192;
constexpr int kScatterLoads = kScatterTile / (kPartBlock * 4);  // 4-sample vectors per lane and sub-tile
static_assert(kPartTile % kScatterTile == 0 && kScatterLoads >= 1, "sub-tiles of whole lane quads");
__host__ __device__ constexpr int part_scatter_slots(int P, int grp) { return kScatterTile + 2 * (grp - 1) * P + grp; }
__host__ __device__ constexpr size_t part_scatter_lds(int P, int grp, bool weighted) {
  return (size_t)kScatterTableBytes + (size_t)P * grp * (weighted ? 12 : 4) + (size_t)part_scatter_slots(P, grp) * (weighted ? 12 : 4) + 64;
}

template <typename WT, int GRP>
__global__ void __launch_bounds__(kPartBlock) part_scatter(const uint32_t* __restrict__ flat, const void* wv_, int64_t n,
                                                            const uint64_t* __restrict__ base_tbl, uint16_t* __restrict__ codes,
                                                            void* __restrict__ wrec_, int shift, int P) {
  constexpr bool kWeighted = !__is_same(WT, NoWeight);
  using wscalar = typename std::conditional<kWeighted, WT, float>::type;
  // record weights keep the caller's precision: float32 weights travel as 4 bytes (converted to
  // float64 when they are added up in part_accumulate, exactly as the single-pass kernels do)
  using RT = typename std::conditional<__is_same(WT, float), float, double>::type;
  constexpr int RV = 16 / (int)sizeof(RT);  // record weights per 16-byte store
  typedef RT rvec __attribute__((ext_vector_type(RV)));
  RT* __restrict__ wrec = static_cast<RT*>(wrec_);
  constexpr int U = kScatterLoads, H = kPartTile / kScatterTile;
  static_assert(GRP == 8 || GRP == 4, "a group is one 16-byte or 8-byte code store");
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  // the caller's weights are only guaranteed element-aligned here (gfx950 vector loads need no more)
  typedef wscalar w4 __attribute__((ext_vector_type(4), aligned(sizeof(wscalar))));
  constexpr uint32_t kGm = GRP - 1;

  const int tid = threadIdx.x;
  uint32_t* cnt2 = reinterpret_cast<uint32_t*>(xhist_smem);                 // [2][256] rank counters, alternating per tile
  uint32_t* cin2 = cnt2 + 512;                                              // [2][256] carried records per partition
  uint64_t* gb = reinterpret_cast<uint64_t*>(xhist_smem + 4096);            // [P] next free (group-aligned) slot per stream
  uint64_t* delta = gb + 256;                                               // [P] stream slot of LDS slot 0 of the block
  uint32_t* first = reinterpret_cast<uint32_t*>(delta + 256);              // [P] LDS slot of the first NEW record
  uint32_t* endw = first + 256;                                             // [P] end of the whole groups of the block
  uint32_t* enda = endw + 256;                                              // [P] end of the records of the block
  uint32_t* total_p = enda + 256;                                           // LDS slots in use this tile
  unsigned char* dyn = xhist_smem + kScatterTableBytes;
  uint32_t* carry_key = reinterpret_cast<uint32_t*>(dyn);                  // [P][GRP]
  dyn += (size_t)P * GRP * 4;
  RT* carry_w = reinterpret_cast<RT*>(dyn);                                 // [P][GRP]
  if (kWeighted) dyn += (size_t)P * GRP * 8;
  const int S = part_scatter_slots(P, GRP);
  RT* sw = reinterpret_cast<RT*>(dyn);                                      // [S] weights of the sorted tile
  if (kWeighted) dyn += (size_t)S * 8;
  uint32_t* skey = reinterpret_cast<uint32_t*>(dyn);                        // [S] keys: part << 16 | code
  const wscalar* wp = reinterpret_cast<const wscalar*>(wv_);
  const uint32_t code_mask = (1u << shift) - 1u;
  for (int i = tid; i < 256; i += blockDim.x) {
    cnt2[i] = 0u;
    cnt2[256 + i] = 0u;
    cin2[i] = 0u;
    cin2[256 + i] = 0u;
    if (i < P) gb[i] = base_tbl[(size_t)blockIdx.x * P + i];
  }
  __syncthreads();

  const int64_t n_tiles = (n + kPartTile - 1) / kPartTile;
  // this workgroup's sub-tiles, in order: sub-tile k = part (k % H) of counting tile blockIdx + (k / H) grid
  const int64_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const int64_t n_sub = my_tiles * H;
  auto sub_base = [&](int64_t k) { return ((int64_t)blockIdx.x + (k / H) * gridDim.x) * kPartTile + (k % H) * kScatterTile; };
  // Loads and the register hand-over below are deliberately free of control flow: vmcnt is ONE
  // in-order counter for loads and stores, and every load the compiler sees under a condition makes
  // its bookkeeping conservative — it then waits (vmcnt(0)) for the record stores of the previous
  // tile before issuing the next tile's loads.  So: the sub-tile after the last one is the last one
  // again (one redundant load per workgroup), and in a ragged sub-tile a weight quad that would
  // cross the end of the array is read 4 elements back from the end and shifted into place when it
  // is USED (weights of positions past the end are never used: their spilled flat index is
  // 0xFFFFFFFF).  Requires n >= 4.
  auto load_tile = [&](int64_t base, u4 (&f)[U], w4 (&w)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + ((int64_t)u * kPartBlock + tid) * 4;
      f[u] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(flat + i));  // padded: always in bounds
      if (kWeighted) w[u] = __builtin_nontemporal_load(reinterpret_cast<const w4*>(wp + min(i, n - 4)));
    }
  };
  u4 f[U], fn[U];
  w4 w[U], wn[U];
  load_tile(sub_base(0), f, w);  // the grid never exceeds the number of tiles
  // the first sub-tile is waited for here, outside the loop
#pragma unroll
  for (int u = 0; u < U; ++u) {
    asm volatile("" : "+v"(f[u]));
    if (kWeighted) asm volatile("" : "+v"(w[u]));
  }
  uint32_t my_carry = 0;  // lane q < P: records of partition q carried into the next sub-tile
  int cur = 0;
  for (int64_t k = 0; k < n_sub; ++k, cur ^= 1) {
    const int64_t base = sub_base(k);
    uint32_t* cnt = cnt2 + (cur << 8);
    const uint32_t* cin = cin2 + (cur << 8);
    uint32_t rank[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) rank[u][v] = (f[u][v] != 0xffffffffu) ? atomicAdd(cnt + (f[u][v] >> shift), 1u) : 0u;
    // the next sub-tile's loads: issued once this one's registers are in use, waited for just BEFORE
    // this one's stores go out (below), so that wait never includes those stores
    load_tile(sub_base(k + 1 < n_sub ? k + 1 : k), fn, wn);
    __syncthreads();
    // Block layout by the first ceil(P/64) wavefronts, one partition per lane: block = carried + new
    // records rounded up to whole groups; exclusive scan of the block sizes with shuffles inside
    // the wavefront, the blocks of earlier wavefronts summed directly (no barrier between them).
    // The same lane owns gb[q]: it advances the stream cursor, keeps the new carry count, and
    // clears the OTHER counter set for the next tile, so a tile costs three barriers.
    if (tid < ((P + 63) & ~63)) {
      const int lane = tid & 63;
      const uint32_t c_in = my_carry;
      const uint32_t T = c_in + (tid < P ? cnt[tid] : 0u);
      const uint32_t block = (T + kGm) & ~kGm;
      uint32_t x = block;  // inclusive scan over the wavefront
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        x += lane >= off ? y : 0u;
      }
      uint32_t before = 0;  // blocks of the partitions handled by earlier wavefronts
      for (int q = lane; q < (tid & ~63); q += 64) before += (cin[q] + cnt[q] + kGm) & ~kGm;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
      const uint32_t B = before + x - block;
      if (tid < P) {
        const uint32_t whole = T & ~kGm;
        first[tid] = B + c_in;
        endw[tid] = B + whole;
        enda[tid] = B + T;
        const uint64_t g = gb[tid];
        delta[tid] = g - B;
        gb[tid] = g + whole;
        my_carry = T - whole;
        cin2[((cur ^ 1) << 8) + tid] = my_carry;
        if (tid == P - 1) *total_p = B + block;
      }
    }
    if (tid < 256) cnt2[((cur ^ 1) << 8) + tid] = 0u;
    __syncthreads();
    const bool ragged = base + kScatterTile > n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      w4 wu = w[u];
      if (kWeighted && ragged) {  // undo the 4-back read of a quad that crossed the end
        const int64_t i = base + ((int64_t)u * kPartBlock + tid) * 4;
        const int sh = (int)min(i - min(i, n - 4), (int64_t)4);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          wscalar x = w[u][v];
#pragma unroll
          for (int k2 = v + 1; k2 < 4; ++k2) x = (v + sh == k2) ? w[u][k2] : x;
          wu[v] = x;
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (f[u][v] != 0xffffffffu) {
          const uint32_t part = f[u][v] >> shift;
          const uint32_t slot = first[part] + rank[u][v];
          skey[slot] = (part << 16) | (f[u][v] & code_mask);
          if (kWeighted) sw[slot] = (RT)wu[v];
        }
    }
    for (int t = tid; t < P * GRP; t += blockDim.x) {  // the carried records go to the head of their block
      const int q = t / GRP, i = t % GRP;
      const uint32_t c = cin[q];
      if ((uint32_t)i < c) {
        const uint32_t slot = first[q] - c + (uint32_t)i;
        skey[slot] = carry_key[t];
        if (kWeighted) sw[slot] = carry_w[t];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f[u] = fn[u];
      w[u] = wn[u];
      asm volatile("" : "+v"(f[u]));  // the wait for the prefetch sits here, ahead of the stores
      if (kWeighted) asm volatile("" : "+v"(w[u]));
    }
    __syncthreads();
    const uint32_t total = *total_p;
    // codes: one lane per group — whole groups leave as one store, the partial last group of a
    // partition (keys and weights) becomes the carry.  A group's first slot is always in use.
    for (uint32_t g0 = (uint32_t)tid * GRP; g0 < total; g0 += kPartBlock * GRP) {
      uint32_t kk[GRP];
#pragma unroll
      for (int i = 0; i < GRP; i += 4) {
        const u4 q4 = *reinterpret_cast<const u4*>(skey + g0 + i);
        kk[i] = q4[0]; kk[i + 1] = q4[1]; kk[i + 2] = q4[2]; kk[i + 3] = q4[3];
      }
      const uint32_t q = kk[0] >> 16;
      if (g0 + GRP <= endw[q]) {
        const uint64_t dst = delta[q] + g0;
        if (GRP == 8) {
          u4 c4;
#pragma unroll
          for (int i = 0; i < 4; ++i) c4[i] = (kk[2 * i] & 0xffffu) | (kk[2 * i + 1] << 16);
          __builtin_nontemporal_store(c4, reinterpret_cast<u4*>(codes + dst));
        } else {
          u2 c2;
#pragma unroll
          for (int i = 0; i < 2; ++i) c2[i] = (kk[2 * i] & 0xffffu) | (kk[2 * i + 1] << 16);
          __builtin_nontemporal_store(c2, reinterpret_cast<u2*>(codes + dst));
        }
      } else {
        const uint32_t left = enda[q] - g0;  // 1 .. GRP-1 records
#pragma unroll
        for (int i = 0; i < GRP; ++i)
          if ((uint32_t)i < left) {
            carry_key[q * GRP + i] = kk[i];
            if (kWeighted) carry_w[q * GRP + i] = sw[g0 + i];
          }
      }
    }
    // weights: one lane per 16 bytes of records (2 float64 / 4 float32), so that a wavefront's store
    // is 1 KiB of one stream
    if (kWeighted) {
      static_assert(GRP % RV == 0, "16-byte weight stores never straddle a group");
      for (uint32_t t0 = (uint32_t)tid * RV; t0 < total; t0 += kPartBlock * RV) {
        const uint32_t g0 = t0 & ~kGm;
        const uint32_t q = skey[g0] >> 16;
        if (g0 + GRP <= endw[q]) {
          const rvec wq = *reinterpret_cast<const rvec*>(sw + t0);
          __builtin_nontemporal_store(wq, reinterpret_cast<rvec*>(wrec + delta[q] + t0));
        }
      }
    }
    // no barrier here: the next tile's ranking touches only the other counter set, and nobody
    // passes that tile's first barrier before every lane has finished this write-out
  }
  __syncthreads();
  // the last group of each stream slice: carried records padded with neutral ones
  if (tid < P && my_carry != 0u) {
    const uint64_t dst = gb[tid];
    for (int i = 0; i < GRP; ++i) {
      const bool real = (uint32_t)i < my_carry;
      codes[dst + i] = real ? (uint16_t)(carry_key[tid * GRP + i] & 0xffffu) : (uint16_t)(kWeighted ? 0u : (1u << shift));
      if (kWeighted) wrec[dst + i] = real ? carry_w[tid * GRP + i] : (RT)0;
    }
  }
}

// counts[G][P] -> offsets[P+1] (start of each partition stream) and base[G][P] (start of each
// workgroup's slice inside it); every slice is rounded up to whole groups of `grp` records, so all
// of them start group-aligned (part_scatter fills the slack with neutral records).  One workgroup
// of 1024 threads arranged as R row groups x P columns so that every global access is coalesced
// along P and each thread walks only G/R rows.
static __global__ void __launch_bounds__(1024) part_prefix(const uint32_t* counts, int G, int P, int grp, uint64_t* offsets, uint64_t* base) {
  __shared__ uint64_t part[1024];  // [R][P] partial sums, then exclusive prefixes over r
  __shared__ uint64_t off[257];
  const int t = threadIdx.x;
  const int R = max(1, (int)blockDim.x / P);
  const int col = t % P, r = t / P;
  const bool active = r < R;
  const int g0 = active ? (int)((int64_t)G * r / R) : 0, g1 = active ? (int)((int64_t)G * (r + 1) / R) : 0;
  const uint32_t gm = (uint32_t)grp - 1u;
  if (active) {
    uint64_t s = 0;
    for (int g = g0; g < g1; ++g) s += (counts[(size_t)g * P + col] + gm) & ~gm;
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
      run += (counts[(size_t)g * P + col] + gm) & ~gm;
    }
  }
}

// pass B: every workgroup takes an equal share of the concatenated record streams (so the load is
// balanced whatever the distribution), accumulates in a 2^shift-bin LDS histogram and flushes it
// to the output whenever its range crosses into the next partition.  The streams contain
// part_scatter's padding records: weight 0 (weighted) or the code 2^shift, one slot past the
// histogram, which is never flushed (unweighted).
// RT: type of the record weights (float32 weights travel as 4 bytes and are widened here)
template <bool WEIGHTED, typename RT = double>
__global__ void __launch_bounds__(1024) part_accumulate(const uint16_t* codes, const void* wrec_, const uint64_t* offsets,
                                                         void* out_v, int64_t n_bins, int shift, int P) {
  using lds_t = typename std::conditional<WEIGHTED, double, uint32_t>::type;
  using out_t = typename std::conditional<WEIGHTED, double, unsigned long long>::type;
  lds_t* hist = reinterpret_cast<lds_t*>(xhist_smem);
  out_t* out = reinterpret_cast<out_t*>(out_v);
  const RT* wrec = static_cast<const RT*>(wrec_);
  const uint32_t bpp = 1u << shift;
  const int tid = threadIdx.x;
  for (uint32_t c = tid; c <= bpp; c += blockDim.x) hist[c] = (lds_t)0;  // [bpp] = trash slot
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
      if (WEIGHTED) unsafeAtomicAdd(reinterpret_cast<double*>(hist) + c, (double)wrec[j]);
      else atomicAdd(reinterpret_cast<uint32_t*>(hist) + c, 1u);
    }
    i = head_end;
    typedef uint16_t c4 __attribute__((ext_vector_type(4)));
    typedef RT w4 __attribute__((ext_vector_type(4)));  // 4 record weights: 16 bytes (float32) or 2 x 16 (float64)
    constexpr int kGroups = 2;
    const uint64_t step = (uint64_t)blockDim.x * 4 * kGroups;
    for (; i + step <= pend; i += step) {
      c4 cv[kGroups];
      w4 wq[kGroups];
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        const uint64_t j = i + ((uint64_t)g * blockDim.x + tid) * 4;
        cv[g] = __builtin_nontemporal_load(reinterpret_cast<const c4*>(codes + j));
        if (WEIGHTED) wq[g] = __builtin_nontemporal_load(reinterpret_cast<const w4*>(wrec + j));
      }
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (WEIGHTED) unsafeAtomicAdd(reinterpret_cast<double*>(hist) + cv[g][k], (double)wq[g][k]);
          else atomicAdd(reinterpret_cast<uint32_t*>(hist) + cv[g][k], 1u);
        }
    }
    for (uint64_t j = i + tid; j < pend; j += blockDim.x) {
      const uint16_t c = codes[j];
      if (WEIGHTED) unsafeAtomicAdd(reinterpret_cast<double*>(hist) + c, (double)wrec[j]);
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
