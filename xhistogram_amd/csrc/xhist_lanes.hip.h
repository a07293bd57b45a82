// xhist_lanes.hip.h — "one row per lane" kernels for MANY SHORT ROWS and for reductions over LEADING
// axes (dim="time" of (time, lat, lon)), the shapes the reference's `block_size` loop exists for
// (core.py:86-134).
//
// hist_fast gives every row its own workgroup(s): perfect for long rows (C4: 10^6 columns), but a
// row of 20..4000 samples cannot amortise zeroing and flushing an LDS histogram, and when the rows
// are the contiguous direction (row stride 1, column stride M) its column-wise streaming is
// uncoalesced.  Here the mapping is transposed:
//   lane  <-> row r            (256 consecutive rows per workgroup: loads x[r + c * col_stride]
//                               are coalesced across lanes)
//   loop  <-> columns c        (UNROLL independent loads in flight per lane)
//   LDS   <-> hist[bin][lane]  (row pitch 257: lane-private, conflict-free, no cross-lane atomics)
// and the finished 256 x nbins tile is written out transposed through LDS with coalesced stores.
// If the rows alone fill the GPU, every output element is written exactly once by plain stores:
// no memset, no global atomics.  Few rows: the columns are split over blockIdx.y and the partial
// tiles are added with global atomics.
//
// Rows that are contiguous in memory ([M, C] row-major, C small) are first transposed into a
// [C, M] scratch by transpose_2d (tiled through LDS), then take the same kernel.
#pragma once

#include "xhist_kernels.hip.h"

namespace xhist {

constexpr int kLaneBlock = 256, kLanePitch = kLaneBlock + 1;

// PACKED16 (unweighted, at most 65535 columns per workgroup): uint16 counters, two rows per LDS
// word — half the LDS, twice the resident workgroups, twice the loads in flight.
template <typename ST, typename WT, int D, int SCAN, int UNROLL, bool PACKED16>
__global__ void __launch_bounds__(kLaneBlock) hist_lanes(const Params p, int32_t direct_store, int64_t cols_per_seg) {
  constexpr int CMP = __is_same(ST, float) ? 2 : 0;
  using CT = typename Dom<CMP>::T;
  constexpr bool kWeighted = !__is_same(WT, NoWeight);
  static_assert(!(PACKED16 && kWeighted), "packed counters count, they do not sum weights");
  using wscalar = typename std::conditional<kWeighted, WT, float>::type;
  using cnt_t = typename std::conditional<kWeighted, double, uint32_t>::type;
  using out_t = typename std::conditional<kWeighted, double, unsigned long long>::type;
  constexpr int kPitch = PACKED16 ? kLaneBlock / 2 + 1 : kLanePitch;  // LDS words per bin

  const int tid = threadIdx.x;
  // R rows per workgroup, G = 256 / R lane groups: with fewer than 256 rows in all, the idle lanes take
  // every G-th column instead of shadowing the last row (10^6 x 64 f64: G = 4; 10^7 x 2: G = 128)
  const int R = p.lane_rows ? p.lane_rows : kLaneBlock;  // power of two, 1 ... 256
  const int G = kLaneBlock / R;
  const int lane_row = tid & (R - 1), g = tid / R;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  const int rows_here = (int)min<int64_t>(R, p.n_rows - r0);
  const int64_t r = r0 + min(lane_row, rows_here - 1);  // lanes past the last row shadow it; never written out
  const uint64_t* tab = stage_tables(p);
  cnt_t* hist = reinterpret_cast<cnt_t*>(xhist_smem + (size_t)p.table_words * 8);
  const uint32_t nb = (uint32_t)p.n_bins;
  // counters: a column per lane (conflict-free, G copies of every row) — or, where nb x 257 words do not fit the LDS (joint
  // histograms over a leading axis: 20 x 20 bins are 411 KB), a column per ROW that the row's G lane groups share: every
  // add is an atomic anyway, and nb x (R + 1) words fit for R = 64 ... 16 rows per workgroup
  const bool shared = p.lane_pitch != 0;
  const uint32_t pitch = shared ? (uint32_t)p.lane_pitch : (uint32_t)kPitch;
  const uint32_t my_col = shared ? (uint32_t)lane_row : (uint32_t)tid;
  for (uint32_t i = tid; i < nb * pitch; i += kLaneBlock) hist[i] = (cnt_t)0;
  __syncthreads();

  const ST* sp[D];
  int64_t scs[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    sp[d] = reinterpret_cast<const ST*>(p.s_ptr[d]) + row_offset(p.row0 + r, p.s_rs[d], p.s_ir[d], p.s_os[d]);
    scs[d] = p.s_cs[d];
  }
  const wscalar* wp =
      kWeighted ? reinterpret_cast<const wscalar*>(p.w_ptr) + row_offset(p.row0 + r, p.w_rs, p.w_ir, p.w_os) : nullptr;
  int max_steps = 1;
#pragma unroll
  for (int d = 0; d < D; ++d) max_steps = max(max_steps, p.dim[d].steps);

  const int64_t c_lo = (int64_t)blockIdx.y * cols_per_seg;
  const int64_t c_hi = min(p.n_cols, c_lo + cols_per_seg);
  cnt_t* mine = hist + (PACKED16 ? my_col >> 1 : my_col);
  const uint32_t my_inc = PACKED16 ? 1u << ((my_col & 1u) << 4) : 1u;

  auto bin_and_add = [&](const ST (&x)[D], wscalar w) {
    bool ok = true;
    uint32_t flat = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      int b;
      if constexpr (scan_is_pack(SCAN)) {
        b = bin_of_sample_pack<CMP, SCAN>((CT)x[d], p.dim[d], tab);
      } else {
        uint32_t cnt;
        if constexpr (SCAN > 0) {
          cnt = count_le_scan<CMP, SCAN>((CT)x[d], p.dim[d], tab);
        } else {
          DigState s = digitize_begin<CMP>((CT)x[d], p.dim[d], tab);
          for (int k = 1; k < max_steps; ++k) upper_bound_step<CMP>((CT)x[d], p.dim[d], tab, s);
          cnt = s.lo;
        }
        b = bin_from_count<CMP>((CT)x[d], p.dim[d], cnt);
      }
      ok &= (b >= 0);
      flat = (d == 0) ? (uint32_t)b : __umul24(flat, (uint32_t)p.dim[d].nb) + (uint32_t)b;
    }
    cnt_t* slot = mine + (ok ? flat : 0u) * pitch;
    if (kWeighted) unsafeAtomicAdd(reinterpret_cast<double*>(slot), ok ? (double)w : 0.0);
    else atomicAdd(reinterpret_cast<uint32_t*>(slot), ok ? my_inc : 0u);
  };

  // columns are dealt to the groups one at a time (column base + u * G + g): with R rows being the contiguous
  // direction, a wavefront's load covers consecutive (column, row) elements whatever R is
  const int64_t chunk = (int64_t)G * UNROLL;
  const int64_t tail_lo = c_lo + (c_hi - c_lo) / chunk * chunk;
  for (int64_t c = c_lo + g; c < tail_lo; c += chunk) {  // (tail_lo - c_lo is a whole number of chunks)
    ST xv[UNROLL][D];
    wscalar wv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int d = 0; d < D; ++d) xv[u][d] = __builtin_nontemporal_load(sp[d] + (c + (int64_t)u * G) * scs[d]);
      if (kWeighted) wv[u] = __builtin_nontemporal_load(wp + (c + (int64_t)u * G) * p.w_cs);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) bin_and_add(xv[u], kWeighted ? wv[u] : (wscalar)0);
  }
  for (int64_t c = tail_lo + g; c < c_hi; c += G) {
    ST x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = sp[d][c * scs[d]];
    bin_and_add(x, kWeighted ? wp[c * p.w_cs] : (wscalar)0);
  }

  // transposed write-out: linear element j of the [rows_here, nb] output tile <- hist[b][row]
  __syncthreads();
  out_t* out = reinterpret_cast<out_t*>(p.out) + r0 * nb;
  const uint32_t total = (uint32_t)rows_here * nb;
  for (uint32_t j = tid; j < total; j += kLaneBlock) {
    const uint32_t row = j / nb, b = j - row * nb;
    cnt_t v = (cnt_t)0;
    for (int gg = 0; gg < (shared ? 1 : G); ++gg) {
      const uint32_t lane = (uint32_t)gg * (uint32_t)R + row;
      if constexpr (PACKED16) v += (hist[b * pitch + (lane >> 1)] >> ((lane & 1) << 4)) & 0xffffu;
      else v += hist[b * pitch + lane];
    }
    if (direct_store) {
      out[j] = (out_t)v;
    } else if (v != (cnt_t)0) {
      if (kWeighted) unsafeAtomicAdd(reinterpret_cast<double*>(out) + j, (double)v);
      else atomicAdd(reinterpret_cast<unsigned long long*>(out) + j, (unsigned long long)v);
    }
  }
}

// Fused form for the commonest short-row case (one input, unweighted, rows contiguous in memory,
// fewer than 65536 columns): the [256 rows x 128 bytes] chunk is loaded coalesced along the rows,
// turned around in LDS (pitch W+1: conflict-free both ways) and consumed one row per lane, so the
// data is read from HBM exactly once and no transposed copy exists.  Counters are uint16, two rows
// per LDS word (a row has < 65536 samples, so they cannot overflow): 50 bins cost 26 KB, not 51.
template <typename ST, int SCAN>
__global__ void __launch_bounds__(kLaneBlock) hist_lanes_rows1(const Params p, int32_t direct_store) {
  constexpr int CMP = __is_same(ST, float) ? 2 : 0;
  using CT = typename Dom<CMP>::T;
  constexpr int W = 128 / (int)sizeof(ST);   // columns per chunk: 32 f32 / 16 f64
  constexpr int VL = 16 / (int)sizeof(ST);   // elements per 16-byte load
  constexpr int TPR = W / VL;                // 8 lanes cover one row's chunk
  constexpr int RPP = kLaneBlock / TPR;      // 32 rows per pass
  constexpr int PASSES = kLaneBlock / RPP;   // 8 passes cover the 256 rows
  constexpr int PITCH = W + 1;
  constexpr int HP = kLaneBlock / 2 + 1;     // histogram row pitch in words (2 rows per word)
  using lvec = typename VecOf<ST, VL>::type;

  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * kLaneBlock;
  const int rows_here = (int)min<int64_t>(kLaneBlock, p.n_rows - r0);
  const uint64_t* tab = stage_tables(p);
  const uint32_t nb = (uint32_t)p.n_bins;
  uint32_t* hist = reinterpret_cast<uint32_t*>(xhist_smem + (size_t)p.table_words * 8);
  ST* tile = reinterpret_cast<ST*>(xhist_smem + (((size_t)p.table_words * 8 + (size_t)nb * HP * 4 + 15) & ~(size_t)15));
  for (uint32_t i = tid; i < nb * HP; i += kLaneBlock) hist[i] = 0u;

  const ST* base = reinterpret_cast<const ST*>(p.s_ptr[0]);
  const int lrow = tid / TPR, lcol = (tid % TPR) * VL;
  int64_t roff[PASSES];  // element offset of the PASSES rows this lane loads from
#pragma unroll
  for (int k = 0; k < PASSES; ++k)
    roff[k] = row_offset(p.row0 + r0 + min(k * RPP + lrow, rows_here - 1), p.s_rs[0], p.s_ir[0], p.s_os[0]);
  const DimTable& t = p.dim[0];
  uint32_t* myword = hist + (tid >> 1);
  const uint32_t myinc = 1u << ((tid & 1) << 4);

  auto load_chunk = [&](int64_t c0, lvec (&v)[PASSES]) {
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int row = k * RPP + lrow;
      const int64_t col = c0 + lcol;
      const ST* src = base + roff[k] + col;
      if (row < rows_here && col + VL <= p.n_cols) {
        v[k] = __builtin_nontemporal_load(reinterpret_cast<const lvec*>(src));
      } else {
#pragma unroll
        for (int j = 0; j < VL; ++j) v[k][j] = (row < rows_here && col + j < p.n_cols) ? src[j] : (ST)__builtin_nanf("");
      }
    }
  };

  lvec cur[PASSES], nxt[PASSES];
  load_chunk(0, cur);
  for (int64_t c0 = 0; c0 < p.n_cols; c0 += W) {
    if (c0 + W < p.n_cols) load_chunk(c0 + W, nxt);  // in flight across the LDS phases below
    __syncthreads();                                   // the previous chunk has been consumed
#pragma unroll
    for (int k = 0; k < PASSES; ++k)
#pragma unroll
      for (int j = 0; j < VL; ++j) tile[(k * RPP + lrow) * PITCH + lcol + j] = cur[k][j];
    __syncthreads();
    const ST* mine = tile + tid * PITCH;
#pragma unroll 8
    for (int c = 0; c < W; ++c) {
      const CT x = (CT)mine[c];
      int b;
      if constexpr (scan_is_pack(SCAN)) {
        b = bin_of_sample_pack<CMP, SCAN>(x, t, tab);
      } else {
        uint32_t cnt;
        if constexpr (SCAN > 0) {
          cnt = count_le_scan<CMP, SCAN>(x, t, tab);
        } else {
          DigState s = digitize_begin<CMP>(x, t, tab);
          for (int k = 1; k < t.steps; ++k) upper_bound_step<CMP>(x, t, tab, s);
          cnt = s.lo;
        }
        b = bin_from_count<CMP>(x, t, cnt);
      }
      atomicAdd(myword + (b >= 0 ? (uint32_t)b : 0u) * HP, b >= 0 ? myinc : 0u);
    }
#pragma unroll
    for (int k = 0; k < PASSES; ++k) cur[k] = nxt[k];
  }

  __syncthreads();
  unsigned long long* out = reinterpret_cast<unsigned long long*>(p.out) + r0 * nb;
  const uint32_t total = (uint32_t)rows_here * nb;
  for (uint32_t j = tid; j < total; j += kLaneBlock) {
    const uint32_t row = j / nb, b = j - row * nb;
    const unsigned long long v = (hist[b * HP + (row >> 1)] >> ((row & 1) << 4)) & 0xffffu;
    if (direct_store) out[j] = v;
    else if (v) atomicAdd(out + j, v);
  }
}

// Flat form for DENSE short rows (every input — and the weights — with row stride == columns, fewer than 65536 columns,
// 16-byte aligned): a workgroup takes R consecutive rows — one contiguous piece of memory per array — and streams it with
// aligned vector loads exactly like a long row, whatever the row length; a sample's row is its position divided by the row
// length (one multiply-high: position * ceil(2^40 / columns) >> 40, exact below 2^22 positions), and its counter lives in an
// LDS histogram [R rows][K copies][bins].  Counts are uint16, two per word (a row has < 65536 samples), weighted sums
// float64.  The lanes of a wavefront read neighbouring samples, i.e. one or two rows at a time, and meet on a row's few bins;
// K copies (lane l on copy l mod K) were built against that and measured useless (K = 1 is as fast or faster: see the host
// side), so K = 1.  The finished [R x bins] block is written once, with plain coalesced stores.  Against hist_lanes_rows1 (128-byte pieces of 256 rows turned around in LDS, two barriers per piece): 10^6 rows of
// 365 float32, 50 bins 0.79 -> 0.40 ms per call (1.86 GB in + out at 4.6 TB/s wall, ~5.3 by the kernel); against one
// 64-thread workgroup per row (weights, joint histograms) see DESIGN.  The host keeps it to rows of up to 800 samples.
template <typename ST, typename WT, int D, int SCAN>
__global__ void __launch_bounds__(kLaneBlock) hist_flat_rows(const Params p, int32_t direct_store, int32_t R, int32_t k_log2, uint64_t magic,
                                                             int64_t n_elems) {
  constexpr int CMP = __is_same(ST, float) ? 2 : 0;
  using CT = typename Dom<CMP>::T;
  constexpr bool kWeighted = !__is_same(WT, NoWeight);
  using wscalar = typename std::conditional<kWeighted, WT, float>::type;
  using cnt_t = typename std::conditional<kWeighted, double, uint32_t>::type;
  using out_t = typename std::conditional<kWeighted, double, unsigned long long>::type;
  constexpr int VEC = 16 / (int)sizeof(ST), UNROLL = (D == 1 && !kWeighted) ? 4 : 2;
  using svec = typename VecOf<ST, VEC>::type;
  using wvec = typename VecOf<wscalar, VEC>::type;
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  const int rows_here = (int)min<int64_t>(R, p.n_rows - r0);
  const uint64_t* tab = stage_tables(p);
  const uint32_t nb = (uint32_t)p.n_bins, nbp = kWeighted ? nb : ((nb + 1u) & ~1u);  // uint16 counters pair up: an even number per (row, copy)
  cnt_t* hist = reinterpret_cast<cnt_t*>(xhist_smem + (((size_t)p.table_words * 8 + 15) & ~(size_t)15));
  const uint32_t slots = ((uint32_t)R << k_log2) * (kWeighted ? nbp : (nbp >> 1));  // LDS elements of the histogram
  for (uint32_t i = tid; i < slots + 32u; i += kLaneBlock) hist[i] = (cnt_t)0;     // (+ 32 trash slots for dropped samples)
  __syncthreads();
  // dense rows: row r of the call starts at element r * n_cols of aligned arrays of n_elems elements
  const ST* base[D];
#pragma unroll
  for (int d = 0; d < D; ++d) base[d] = reinterpret_cast<const ST*>(p.s_ptr[d]);
  const wscalar* wbase = reinterpret_cast<const wscalar*>(p.w_ptr);
  const int64_t g0 = (p.row0 + r0) * p.n_cols, g1 = g0 + (int64_t)rows_here * p.n_cols;  // this workgroup's samples
  const int64_t v_lo = g0 / VEC, v_hi = (g1 + VEC - 1) / VEC;                            // ... and the aligned vectors that hold them
  const int64_t v_whole = n_elems / VEC;                                                 // vectors that lie wholly inside the arrays
  const uint32_t n_here = (uint32_t)(g1 - g0);
  const uint32_t copy_off = ((uint32_t)tid & ((1u << k_log2) - 1u)) * nbp;
  const uint32_t trash = slots + ((uint32_t)tid & 31u);
  int max_steps = 1;
#pragma unroll
  for (int d = 0; d < D; ++d) max_steps = max(max_steps, p.dim[d].steps);
  for (int64_t v0 = v_lo + tid; v0 < v_hi; v0 += kLaneBlock * UNROLL) {
    svec xv[D][UNROLL];
    wvec wv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t vv = min(v0 + (int64_t)u * kLaneBlock, v_hi - 1);  // (past the workgroup's last vector: that one again, masked below)
      if (vv < v_whole) {
#pragma unroll
        for (int d = 0; d < D; ++d) xv[d][u] = __builtin_nontemporal_load(reinterpret_cast<const svec*>(base[d]) + vv);
        if (kWeighted) wv[u] = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(wbase) + vv);
      } else {  // the arrays' ragged last vector (one lane of one workgroup): element by element
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const bool in = vv * VEC + v < n_elems;
#pragma unroll
          for (int d = 0; d < D; ++d) xv[d][u][v] = in ? base[d][vv * VEC + v] : (ST)__builtin_nanf("");
          if (kWeighted) wv[u][v] = in ? wbase[vv * VEC + v] : (wscalar)0;
        }
      }
    }
    uint32_t cnt[D][UNROLL][VEC];
    count_le_tile<CMP, SCAN, D, UNROLL, VEC>(xv, p, tab, max_steps, cnt);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t vv = v0 + (int64_t)u * kLaneBlock;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int64_t i64 = vv * VEC + v - g0;  // the sample's position among this workgroup's
        const uint32_t i = (uint32_t)i64;
        bool ok = (i64 >= 0) & (i64 < (int64_t)n_here);
        uint32_t flat = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const int b = bin_from_tile_count<CMP, SCAN>((CT)xv[d][u][v], p.dim[d], cnt[d][u][v]);
          ok &= (b >= 0);
          flat = (d == 0) ? (uint32_t)b : __umul24(flat, (uint32_t)p.dim[d].nb) + (uint32_t)b;
        }
        const uint32_t row = (uint32_t)(((uint64_t)i * magic) >> 40);
        const uint32_t idx = ((row << k_log2) * nbp) + copy_off + flat;
        if constexpr (kWeighted) unsafeAtomicAdd(reinterpret_cast<double*>(hist) + (ok ? idx : trash), ok ? (double)wv[u][v] : 0.0);
        else atomicAdd(reinterpret_cast<uint32_t*>(hist) + (ok ? (idx >> 1) : trash), ok ? (1u << ((idx & 1u) << 4)) : 0u);
      }
    }
  }
  __syncthreads();
  out_t* out = reinterpret_cast<out_t*>(p.out) + r0 * nb;
  const uint32_t total = (uint32_t)rows_here * nb;
  const uint32_t K = 1u << k_log2;
  // (row = j / nb by one multiply-high: j * nb < 2^32 here, so ceil(2^32 / nb) is exact — short rows write more than they
  // read, and a 32-bit division per output element showed: 18.25 x 10^6 rows of 20, 1.83 -> 1.77 ms)
  const uint32_t nb_magic = 0xffffffffu / nb + 1u;
  for (uint32_t j = tid; j < total; j += kLaneBlock) {
    const uint32_t row = nb == 1u ? j : __umulhi(j, nb_magic), b = j - row * nb;
    cnt_t v = (cnt_t)0;
    for (uint32_t k = 0; k < K; ++k) {
      const uint32_t idx = (row * K + k) * nbp + b;
      if constexpr (kWeighted) v += hist[idx];
      else v += (hist[idx >> 1] >> ((idx & 1u) << 4)) & 0xffffu;
    }
    if (direct_store) {
      out[j] = (out_t)v;
    } else if (v != (cnt_t)0) {
      if constexpr (kWeighted) unsafeAtomicAdd(reinterpret_cast<double*>(out) + j, (double)v);
      else atomicAdd(reinterpret_cast<unsigned long long*>(out) + j, (unsigned long long)v);
    }
  }
}

// [n_rows, n_cols] (row stride `rs` elements, unit column stride) -> dense [n_cols, n_rows].
// 64 x 64 tiles through LDS (pitch 65): coalesced 256-byte reads along rows, writes along columns.
template <typename T>
__global__ void __launch_bounds__(256) transpose_2d(const T* __restrict__ in, int64_t rs, int64_t n_rows, int64_t n_cols,
                                                     T* __restrict__ out) {
  __shared__ T tile[64][65];
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;  // rows on x: up to 2^31 tiles
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t rr = r0 + ty + 4 * k, cc = c0 + tx;
    if (rr < n_rows && cc < n_cols) tile[ty + 4 * k][tx] = in[rr * rs + cc];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t cc = c0 + ty + 4 * k, rr = r0 + tx;
    if (rr < n_rows && cc < n_cols) out[cc * n_rows + rr] = tile[tx][ty + 4 * k];
  }
}

// The other direction: a few rows whose CONTIGUOUS direction is the row index — a view [k rows, n cols] of an (n, K)
// table (element (r, c) at in[c * K + r]: the columns of a table histogrammed over its leading axis) -> dense [k, n].
// One table row per lane: the reads of a wavefront cover 64 consecutive table rows (16-byte loads when a row is
// whole 16-byte units), the writes of each table column are consecutive.
template <typename T>
__global__ void __launch_bounds__(256) gather_rows(const T* __restrict__ in, int64_t K, int k, int64_t n, T* __restrict__ out) {
  constexpr int kVec = 16 / (int)sizeof(T);
  typedef T tvec __attribute__((ext_vector_type(kVec), aligned(16)));
  const bool whole = K == k && (k % kVec) == 0 && ((uintptr_t)in % 16) == 0;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
    const T* row = in + c * K;
    if (whole) {
      for (int j = 0; j < k; j += 4 * kVec) {  // four 16-byte loads in flight per lane, then their stores
        tvec q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j + u * kVec < k) q[u] = __builtin_nontemporal_load(reinterpret_cast<const tvec*>(row + j + u * kVec));
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j + u * kVec < k) {
#pragma unroll
            for (int v = 0; v < kVec; ++v) __builtin_nontemporal_store(q[u][v], out + (int64_t)(j + u * kVec + v) * n + c);
          }
      }
    } else {
      for (int j = 0; j < k; ++j) out[(int64_t)j * n + c] = row[j];
    }
  }
}

// The same for a WHOLE table (K == k, rows of whole 16-byte units), tiled through LDS: 256 table rows are copied into
// LDS as they lie (flat 16-byte loads: every fetched line is used once, which the lane-per-row reads above leave to
// the 32 KiB vector cache — 2.7 TB/s for 16 float64 columns), then written out column by column, 256 consecutive
// elements at a time.  Row pitch k + 1 elements: the column reads are conflict-free.
template <typename T>
__global__ void __launch_bounds__(256) gather_rows_tiled(const T* __restrict__ in, int k, int64_t n, T* __restrict__ out) {
  constexpr int kVec = 16 / (int)sizeof(T), R = 256;
  typedef T tvec __attribute__((ext_vector_type(kVec), aligned(16)));
  T* tile = reinterpret_cast<T*>(xhist_smem);  // [R][k + 1]
  const int tid = threadIdx.x, pitch = k + 1;
  const int64_t n_tiles = (n + R - 1) / R;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t r0 = t * R;
    const int rows = (int)min<int64_t>(R, n - r0);
    const int n_vec = rows * k / kVec;
    const T* src = in + r0 * k;
    for (int i = tid; i < n_vec; i += R) {
      const tvec q = __builtin_nontemporal_load(reinterpret_cast<const tvec*>(src) + i);
      const int e = i * kVec, row = e / k, col = e - row * k;
#pragma unroll
      for (int v = 0; v < kVec; ++v) tile[row * pitch + col + v] = q[v];
    }
    __syncthreads();
    if (tid < rows)
      for (int j = 0; j < k; ++j) __builtin_nontemporal_store(tile[tid * pitch + j], out + (int64_t)j * n + r0 + tid);
    __syncthreads();
  }
}

}  // namespace xhist
