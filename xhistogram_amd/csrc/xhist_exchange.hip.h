// xhist_exchange.hip.h — the partitioned mode WITHOUT a record round trip through HBM: a window of the histogram is kept in
// the LDS of the compute units of an XCD, and records travel from the workgroup that read a sample to the workgroup that
// owns its bin through small rings in that XCD's memory path (round 5; BASELINE C5: 2 x float64 samples + float64 weights
// into 1024 x 1024 bins, /root/reference/xhistogram/core.py:73-83, :178-183).
//
// part_route + part_accumulate_chunks (xhist_route.hip.h) move 40 B per C5 sample — 24 read, 8 written as a record, 8 read
// back — and sit at the chip's rate for that mix (3.95-4.1 ms per 5*10^8 samples, 0.37-0.39 of 8 TB/s).  Here:
//
//   * 256 persistent workgroups of 1024 threads, ONE per compute unit, 32 per XCD (the workgroup reads HW_REG_XCC_ID and
//     takes a place among its XCD's 32 with an atomic).  The 32 workgroups of an XCD keep a WINDOW of histogram rows between
//     them: every 32 consecutive rows go to 32 different owners (rotated from block to block, exch_owner), so smooth
//     distributions — and the striped ones of three inputs — load them evenly; rows_per rows each, <= 15360 bins = 120 KB
//     of float64 in LDS.  Every XCD holds the whole window; the eight partial windows are added up by exchange_merge.  A "row" is the last dimension of the
//     histogram (256 consecutive bins of a one-dimensional one).
//   * Every workgroup is producer AND consumer.  Per tile of 4096 samples: digitize (bin_arith_fast + the exact redo, as
//     part_route does), rank the tile's records by owner with LDS counters, lay them out by owner in LDS, and write each
//     owner's run into the ring (this producer -> that owner): 32 x 32 rings of 512 records per XCD.  A record is ONE 8-byte
//     word {48-bit rounded weight (pack48), 14-bit bin inside the owner's rows, 2-bit lap tag}: the tag makes the word its
//     own ready flag — nothing is fenced, nothing ordered, no separate flag — and the owner takes the valid prefix of what it
//     finds (32 lanes per ring, 4 records each per tile) and adds it to its LDS rows; it publishes how far it has read
//     (credits), which the producer reads once per tile.  A ring without room makes its producer take from its own rings
//     until the owner has caught up (every workgroup is resident, so this always ends; a deadline turns a hang into the
//     fallback below).  Plain stores and agent-scope (L1-bypassing) loads are enough INSIDE one XCD: both sides
//     go through the same L2.  Across XCDs they are not (measured: stale for ever), and write-through stores cost
//     3.97 against 3.29 ms in the stand-alone kernel — which is why the window is per XCD and not 8 x larger.
//   * Samples outside the window go to a side copy of the output with memory-side atomics (2.4*10^10 per second: fine for
//     ~10 % of the samples, hopeless for half of them), so WHERE the window lies decides whether the mode pays:
//     exchange_probe histograms the rows (and the owners) of 2.6*10^5 samples spread over the input, exchange_pick takes the
//     best window and switches the mode on when it holds >= 88 % of them and no owner would get more than 1.25 x an even
//     share of the records — both on the GPU, every call, no host synchronisation; classic
//     kernels queued behind return at once when the mode is on, these return at once when it is off.
//   * Weights of both signs: the kernel reports them in the flags word of the packed routing pass, its merge does not run, and
//     the exact routing + adding-up passes queued behind redo the whole call from the (untouched) output — this kernel and
//     exchange_merge write only scratch until the merge.  A workgroup placement other than 32 per XCD, or a wait that outlives
//     its deadline: the mode is switched off for the call, and the classic packed passes queued behind take it.
//
// Traffic per C5 sample: 24 B read, 8 B written to the rings and 8 B read from them.  The counters say every ring read
// misses the L2 (the sample stream evicts the rings: FETCH 15.8 GB per 5*10^8 samples = 12.0 of samples + 3.74 of records;
// WRITE 5.07 GB = 3.76 of records + 1.3 of side atomics; with no stream beside them the reads do hit), so as many bytes cross
// the far side of the L2 as in the classic pair — but the 33 MB of rings are rewritten every few microseconds and need
// never reach HBM, the only bytes that must come from there are the samples', and there is no second kernel.  Measured
// (tools/ubench/xchg.hip is the kernel as a stand-alone program, profiles/r05_x_*): C5 shard 3.26-3.44 ms
// against 3.95-4.1; the producing half alone 2.5 ms.
#pragma once

#include "xhist_route.hip.h"

namespace xhist {

constexpr int kExchBlock = 1024, kExchTile = 4096;
constexpr int kExchXcds = 8, kExchRings = 32;            // XCDs of the chip; workgroups (= ring ends) per XCD
constexpr int kExchCapLog2 = 9, kExchCap = 1 << kExchCapLog2;  // records per ring (256: producers wait for credits half the time, 4.04 against 3.3 ms)
constexpr int kExchTake = 4;                             // ring records a lane looks at per tile (6: 3.34 against 3.15 ms — what it finds not yet written is traffic too)
constexpr int kExchMaxLocal = 15360;                     // bins a workgroup keeps: 120 KB of float64 next to the tile's 36 KB
constexpr int kExchMaxLocalExact = 14336;                // ... with EXACT records, whose tile takes 44 KB (the 16 low weight bits beside the packed word)
constexpr int kExchCtlBytes = 2048;
constexpr uint32_t kExchUnitRows = 32;                   // the probe counts rows in units of 32; a window starts on a unit
constexpr int kExchMinPpm = 880000;                      // window coverage from which the mode takes a call (exchange_pick; where 88 % comes from: DESIGN 4.2b)
constexpr uint32_t kExchArrivedMask = 0xffffu, kExchArrivePoison = 0x10000u;  // ExchCtl::arrived: a count and one "somebody gave up waiting" bit

struct ExchCtl {             // one per XCD
  uint32_t nreg;             // workgroups that took a place on this XCD
  uint32_t abort;            // [XCD 0 only] some workgroup gave up: everybody leaves
  uint32_t arrived;          // [XCD 0 only] workgroups that have started (low 16 bits) | kExchArrivePoison once one of them has stopped waiting for the rest
  uint32_t pad[29];
  uint32_t head[kExchRings][kExchRings];  // [producer][owner]: records of that ring the owner has taken
  uint32_t fin[kExchRings][kExchRings];   // [producer][owner]: 1 + final record count of the ring (0 while the producer works)
};

// what the digitize of one input needs (arithmetic edges: DimTable's e0_f, eL_f, step, inv_step, arith_h, nb)
struct ExchDim {
  double e0, eL, step, inv_step, arith_h;
  int32_t nb, pad;
};

// what only the rare paths of part_exchange need (the exact redo of a sample next to an edge, an abort, the epilogue), kept in
// device memory: as kernel arguments these values sat in scalar registers through the whole tile loop, and the loop spilled
// 88 of them into vector-register lanes — 140 v_readlane per tile.  exchange_pick copies them here from the arguments.
struct ExchCold {
  double eL[3], step[3];
  long long budget_ticks, arrive_ticks;
  uint32_t* flags;
  uint32_t* note;
  double* part;
};

struct ExchArgs {
  const double* s_ptr[3];    // the inputs (one row, unit stride) and their weights
  const double* w_ptr;
  int64_t n;                 // samples
  ExchDim dim[3];
  ExchCtl* ctl;              // [kExchXcds], zeroed per call
  uint64_t* rings;           // [xcd][owner][producer][kExchCap], zeroed per call (tag 0 = never written)
  uint32_t* rings_lo;        // EXACT records only: a second ring of the same shape, {the 16 weight bits the packed word drops | lap tag << 16}
  double* part;              // [xcd][owner][local_bins]: the XCD partials of the window
  double* side;              // [n_bins], zeroed per call: samples outside the window
  uint32_t* win;             // device words: [0] first row of the window, [1] mode on, [2] coverage in ppm (exchange_pick writes them)
  ExchCold* cold;            // device copy of the rarely needed arguments (exchange_pick writes it)
  uint32_t* counts;          // [n_units] rows of the probe's samples per unit of 32 rows, [32] per owner; zeroed per call
  uint32_t* flags;           // the packed pass's sign word: 1 negative, 2 positive weights seen; 3 also stands for "redo exactly"
  uint32_t* note;            // pinned host words (may be NULL): [2] += 1 per abort in flight, [3] coverage ppm of the last pick, [4] busiest owner, [5] += 1 per call whose workgroups were not all resident when it started
  int64_t row_len;           // L: bins per row (last dimension; 256 for one dimension)
  int64_t n_hist_rows;       // rows of the histogram
  int32_t rows_per;          // rows per owner: the window holds 32 * rows_per rows
  int32_t local_bins;        // rows_per * L
  int32_t n_units;           // ceil(n_hist_rows / 32)
  int32_t force;             // 1: mode on whatever the coverage (tests, "exchange" = 1)
  int32_t side_rot_mask;     // exch_side_index: (largest power of two <= L) - 1
  int32_t min_ppm;           // coverage that switches the mode on
  int32_t max_uneven_ppm;    // ... unless the busiest owner would get more than this many ppm of an even share of the records
  long long budget_ticks;    // deadline of a workgroup's waits once records travel, in ticks of the 100 MHz clock
  long long arrive_ticks;    // how long a workgroup waits for the other 255 to start before it hands the call to the classic passes
};

__host__ __device__ constexpr size_t exchange_lds(int local_bins, bool exact = false) {
  return (((size_t)local_bins * 8 + 15) & ~(size_t)15) + (size_t)kExchTile * (exact ? 10 : 8) + kExchTile + kExchCtlBytes;
}

__device__ __forceinline__ uint32_t exch_xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}
// loads that bypass the vector L1 (another compute unit's stores are never seen through it) and are served by the XCD's L2
__device__ __forceinline__ uint64_t exch_ld(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t exch_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void exch_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The owner of window row r: one of every 32 consecutive rows each, rotated by 13 from one block of 32 rows to the next.  Plain
// r mod 32 is as even for a smooth distribution, but rows = all inputs but the last, so a three-input histogram's heavy rows come
// in runs that repeat with the second input's bin count — 32 x 32 x 1024 bins of N(0,1) samples: eight owners took every record
// and the call 4.43 ms against 2.96 for the classic passes; rotated: see DESIGN 4.2b.  The row inside the owner's LDS copy stays r / 32.
__device__ __forceinline__ uint32_t exch_owner(uint32_t r) { return (r + 13u * (r >> 5)) & 31u; }

// Where bin (row, col) lives in the SIDE copy of the output (what lies outside the window, added with memory-side atomics): every
// row rotated by a row-dependent number of columns.  Those samples sit in the rows next to the window and — for a bell-shaped
// second input — in the same run of columns of each: chunks of a few KB, a whole row (8-16 KB) apart, which the memory channels
// interleave badly (512 x 2048 bins, 9 % outside: 2.79 ms against 2.39 for the classic passes; with the rotation: DESIGN 4.2b).
// rot_mask = the largest power of two <= L, minus one, so one conditional subtraction wraps.
__device__ __forceinline__ uint32_t exch_side_index(uint32_t row, uint32_t col, uint32_t L, uint32_t rot_mask) {
  uint32_t c = col + ((row * 40503u >> 4) & rot_mask);
  c -= c >= L ? L : 0u;
  return row * L + c;
}

// pack48 for finite weights below the top binade: round to nearest-even on bit 16 with one 64-bit add.  `special` comes back
// true for NaN / infinity / the top binade (where the carry could reach the infinity exponent): the caller redoes those with
// pack48 itself.  Same result as pack48 wherever `special` is false.
__device__ __forceinline__ uint64_t exch_pack_fast(double w, uint32_t code, bool& special) {
  const uint64_t b = (uint64_t)__double_as_longlong(w);
  const uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
  special = (hi & 0x7fe00000u) == 0x7fe00000u;
  const uint64_t r = b + 0x7fffull + ((lo >> 16) & 1u);
  return (r & ~0xffffull) | code;
}

// #{j : e_j <= x} for arithmetic edges, by exact compares against the recomputed edges (count_le_arith, xhist_kernels.hip.h)
__device__ __forceinline__ uint32_t exch_count_le(double x, const ExchDim& t) {
  double tt = (x - t.e0) * t.inv_step;
  tt = fmax(fmin(tt, (double)(t.nb - 1)), 0.0);
  const double gd = __builtin_floor(tt);
  double m0 = gd * t.step, m1 = (gd + 1.0) * t.step;
  asm volatile("" : "+v"(m0), "+v"(m1));
  const double e_g = m0 + t.e0;
  const uint32_t g = (uint32_t)(int)gd;
  const double e_g1 = (int)g + 1 == t.nb ? t.eL : m1 + t.e0;
  return g + (e_g <= x ? 1u : 0u) + (e_g1 <= x ? 1u : 0u);
}

// Bins of N samples per input, and whether the reference keeps the sample (core.py:163-174): bin_arith_fast's arithmetic
// (9 float64 operations per sample and input; "inside" stays a compare mask), and for a wavefront in which some lane met a
// sample on / next to an edge, a NaN or an infinity, that lane's samples again with the exact compares.
template <int D, int N>
__device__ __forceinline__ void exch_digitize(const double (&x)[D][N], const ExchArgs& xa, uint32_t (&g)[D][N], bool (&ins)[N], const volatile ExchCold* cold = nullptr) {
  bool near_any = false;
#pragma unroll
  for (int v = 0; v < N; ++v) ins[v] = true;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    double e0 = xa.dim[d].e0, inv = xa.dim[d].inv_step, h = xa.dim[d].arith_h;
    int nb = xa.dim[d].nb;
    asm volatile("" : "+s"(e0), "+s"(inv), "+s"(h), "+s"(nb));  // (scalar registers: left alone, the compiler copies them into vector ones)
    const double nbf = (double)nb;
#pragma unroll
    for (int v = 0; v < N; ++v) {
      double tt = (x[d][v] - e0) * inv;
      asm volatile("" : "+v"(tt));  // the ROUNDED product, as plan creation measured it (see bin_arith_fast)
      const double fl = __builtin_floor(tt);
      const double f = tt - fl;
      near_any |= !(__builtin_fabs(f - 0.5) < h);  // (NaN compares false: near)
      g[d][v] = (uint32_t)(uint64_t)__double_as_longlong(fl + 6755399441055744.0);  // fl mod 2^32 without a conversion
      ins[v] &= (fl >= 0.0) & (fl < nbf);
    }
  }
  if (__builtin_amdgcn_ballot_w64(near_any) != 0ull) {
    if (near_any) {
#pragma unroll
      for (int v = 0; v < N; ++v) ins[v] = true;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        ExchDim t = xa.dim[d];
        if (cold) {  // (the two constants only this path needs come from memory, not from registers held through the caller's loop)
          t.eL = cold->eL[d];
          t.step = cold->step[d];
        }
#pragma unroll
        for (int v = 0; v < N; ++v) {
          g[d][v] = (uint32_t)min((int)exch_count_le(x[d][v], t) - 1, t.nb - 1);  // (x == e_last counts E edges: last bin)
          ins[v] &= (x[d][v] >= t.e0) & (x[d][v] <= t.eL);                       // (false for NaN)
        }
      }
    }
  }
}

// row (all inputs but the last; 256-bin pieces of a one-dimensional histogram) and column of a sample
template <int D>
__device__ __forceinline__ void exch_row_col(const uint32_t (&g)[D], const ExchArgs& xa, uint32_t& row, uint32_t& col) {
  if constexpr (D == 1) {
    row = g[0] >> 8;
    col = g[0] & 255u;
  } else {
    row = g[0];
#pragma unroll
    for (int d = 1; d < D - 1; ++d) row = row * (uint32_t)xa.dim[d].nb + g[d];
    col = g[D - 1];
  }
}

// ---- where the window should lie: rows of a sample of the input, per unit of 32 rows ------------------------------
template <int D>
__global__ void __launch_bounds__(256) exchange_probe(const ExchArgs xa) {
  __shared__ uint32_t lc[4096 + kExchRings];  // rows per unit of 32, then records per owner
  const int tid = threadIdx.x;
  for (int i = tid; i < xa.n_units + kExchRings; i += 256) lc[i] = 0u;
  __syncthreads();
  const int64_t n = xa.n;
  // workgroup b looks at 1024 consecutive samples a (1 / grid)-th of the way further into the input
  const int64_t pos = (n / gridDim.x) * blockIdx.x + (int64_t)tid * 4;
  double x[D][4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int d = 0; d < D; ++d) x[d][v] = pos + v < n ? xa.s_ptr[d][pos + v] : (double)__builtin_nanf("");
  uint32_t g[D][4];
  bool ins[4];
  exch_digitize<D, 4>(x, xa, g, ins);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    uint32_t gg[D], row, col;
#pragma unroll
    for (int d = 0; d < D; ++d) gg[d] = g[d][v];
    exch_row_col<D>(gg, xa, row, col);
    if (ins[v]) {
      atomicAdd(lc + min(row / kExchUnitRows, (uint32_t)xa.n_units - 1u), 1u);
      // who would own it (a window starts on a multiple of 32 rows, which only rotates the owners: the LARGEST load is the same)
      atomicAdd(lc + xa.n_units + exch_owner(row), 1u);
    }
  }
  __syncthreads();
  for (int i = tid; i < xa.n_units + kExchRings; i += 256)
    if (lc[i]) atomicAdd(xa.counts + i, lc[i]);
}

// the run of `rows_per` units with the most samples; mode on when it holds enough of them (one workgroup of 64 lanes)
// (a template only so that the header can be included by several translation units)
template <int UNUSED = 0>
__global__ void __launch_bounds__(64) exchange_pick(const ExchArgs xa) {
  if (threadIdx.x != 0) return;
  const int nu = xa.n_units, wu = xa.rows_per;  // (a window of 32 * rows_per rows = rows_per units)
  uint64_t total = 0, run = 0, best = 0;
  int best_u = 0;
  for (int u = 0; u < nu; ++u) {
    const uint32_t c = xa.counts[u];
    total += c;
    run += c;
    if (u >= wu) run -= xa.counts[u - wu];
    if (u >= wu - 1 || u == nu - 1) {
      if (run > best) { best = run; best_u = max(u - wu + 1, 0); }
    }
  }
  if (wu >= nu) best_u = 0;  // the whole histogram fits the window
  const uint32_t ppm = total ? (uint32_t)(best * 1000000ull / total) : 1000000u;
  // the busiest owner's share of the records against an even one (1.0 = even): every workgroup takes what 32 producers send it
  // between its own tiles, so the busiest owner sets the pace — samples in a handful of rows (or striped rows that the rotation
  // does not spread) would make one workgroup take everything (measured: 5-60 ms against 2.5-3.9 for the classic passes)
  uint32_t busiest = 0;
  for (int k = 0; k < kExchRings; ++k) busiest = max(busiest, xa.counts[nu + k]);
  const uint32_t even_ppm = total ? (uint32_t)((uint64_t)busiest * kExchRings * 1000000ull / total) : 1000000u;
  xa.win[0] = (uint32_t)best_u * kExchUnitRows;
  xa.win[1] = (xa.force || (ppm >= (uint32_t)xa.min_ppm && even_ppm <= (uint32_t)xa.max_uneven_ppm)) ? 1u : 0u;
  xa.win[2] = ppm;
  xa.win[3] = even_ppm;
  if (xa.note) {
    xa.note[3] = ppm;
    xa.note[4] = even_ppm;
  }
  for (int d = 0; d < 3; ++d) {
    xa.cold->eL[d] = xa.dim[d].eL;
    xa.cold->step[d] = xa.dim[d].step;
  }
  xa.cold->budget_ticks = xa.budget_ticks;
  xa.cold->arrive_ticks = xa.arrive_ticks;
  xa.cold->flags = xa.flags;
  xa.cold->note = xa.note;
  xa.cold->part = xa.part;
}

// EXACT (round 6): the weight travels whole — the packed word carries its upper 48 bits (truncated, not rounded), a second ring of
// 4-byte words {the 16 bits below | the same lap tag} the rest; a record is valid when BOTH words carry the lap's tag (each is
// written and read as one aligned word).  The reference's arithmetic (float64 adds of unrounded weights, core.py:81), weights of
// any sign mixture, 12 bytes per record through the rings instead of 8; the tile's staging takes 8 KB more LDS, the window one row less.
template <int D, bool EXACT = false>
__global__ void __launch_bounds__(kExchBlock) part_exchange(const ExchArgs xa) {
  constexpr int BLOCK = kExchBlock, TILE = kExchTile, NS = kExchRings, CAP = kExchCap, CAPL = kExchCapLog2, CL = kExchTake;
  typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));  // (a view like x[1:] is 8-byte aligned only; the loads stay 16-byte ones)
  if (__builtin_nontemporal_load(xa.win + 1) == 0u) return;  // the window does not hold enough of this call's samples
  const int tid = threadIdx.x;
  const size_t hist_bytes = ((size_t)xa.local_bins * 8 + 15) & ~(size_t)15;
  double* hist = reinterpret_cast<double*>(xhist_smem);
  uint64_t* srec = reinterpret_cast<uint64_t*>(xhist_smem + hist_bytes);   // [TILE] the tile's ring records, by owner
  uint16_t* slo = reinterpret_cast<uint16_t*>(xhist_smem + hist_bytes + (size_t)TILE * 8);  // [TILE] EXACT: the 16 low weight bits of every staged record
  uint8_t* sd = xhist_smem + hist_bytes + (size_t)TILE * (EXACT ? 10 : 8);  // [TILE] owner of every staged record
  uint32_t* c = reinterpret_cast<uint32_t*>(sd + TILE);
  uint32_t* cnt2 = c;          // [2][64] records per owner of the tile, alternating per tile
  uint32_t* off = c + 128;     // [64] first staged slot of every owner; [32] = records staged
  uint32_t* tail = c + 192;    // [32] records this workgroup has sent to owner d (earlier tiles)
  uint32_t* wbase = c + 224;   // [32] ring position of this tile's first record for d
  uint32_t* credit = c + 256;  // [32] head[me][d] as last seen
  uint32_t* chead = c + 288;   // [32] records taken from ring (p -> me)
  uint32_t* lim = c + 320;     // [32] records of this tile's run for d that may be written so far
  uint32_t* sent = c + 352;    // [32] ... and that have been written
  uint32_t* wadj = c + 384;    // [32] wbase[d] - off[d]: ring position = wadj[d] + staged slot
  uint32_t* misc = c + 416;    // [0] XCD [1] place [3] runs not written out yet [4] abort [5] not everybody arrived [6, 7] start time
  const volatile ExchCold* cold = xa.cold;

  for (int i = tid; i < xa.local_bins; i += BLOCK) hist[i] = 0.0;
  if (tid < 128) cnt2[tid] = 0u;
  if (tid < NS) { tail[tid] = 0u; credit[tid] = 0u; chead[tid] = 0u; }
  if (tid == 0) {
    const uint32_t xc = exch_xcc_id() & 7u;
    misc[0] = xc;
    misc[1] = atomicAdd(&xa.ctl[xc].nreg, 1u);
    misc[3] = 0u;
    misc[4] = 0u;
    // Everybody must be resident before anybody produces a record (a workgroup waits for 255 others): say "here", then wait
    // for the rest — a few microseconds when the chip is free.  A compute unit held by somebody else's kernel (another stream,
    // another process, a collective) keeps one of us out: the wait ends after arrive_ticks, NOTHING has been consumed or
    // produced, and the classic passes queued behind take the call as if the probe had said no (note[5] counts those calls;
    // the plan stays on the mode).  The long deadline (budget_ticks) is for stalls once records travel.
    // (The word decides for everybody alike: the poison bit can only be set — by ONE compare-and-swap — while fewer than all have
    // arrived, and once it is set everybody who looks, or arrives later, leaves; all arrived without it: nobody can set it any more.)
    uint32_t v = atomicAdd(&xa.ctl[0].arrived, 1u) + 1u;
    const long long t0 = wall_clock64();
    const long long patience = cold->arrive_ticks;
    bool first_to_give_up = false;
    while ((v & kExchArrivedMask) < gridDim.x && !(v & kExchArrivePoison)) {
      if (wall_clock64() - t0 > patience) {
        const uint32_t seen = atomicCAS(&xa.ctl[0].arrived, v, v | kExchArrivePoison);
        first_to_give_up = seen == v;
        v = first_to_give_up ? (v | kExchArrivePoison) : seen;
        continue;
      }
      __builtin_amdgcn_s_sleep(4);
      v = exch_ld(&xa.ctl[0].arrived);
    }
    misc[5] = (v & kExchArrivePoison) ? (first_to_give_up ? 2u : 1u) : 0u;
    *reinterpret_cast<long long*>(misc + 6) = wall_clock64();  // from here on: waits give up budget_ticks later
  }
  __syncthreads();
  const uint32_t xcd = misc[0], me = misc[1];
  uint32_t* g_abort = &xa.ctl[0].abort;
  if (misc[5] || me >= (uint32_t)NS) {  // not everybody here, or not 32 workgroups on this XCD (no ring ends here): everybody leaves
    if (tid == 0) {
      exch_st(g_abort, 1u);
      exch_st(xa.win + 1, 0u);  // the mode is off for this call after all: the classic packed passes queued behind take it
      if (cold->note) {
        if (misc[5] == 2u) atomicAdd(cold->note + 5, 1u);        // once per call: by the workgroup that stopped the wait
        else if (misc[5] == 0u) atomicAdd(cold->note + 2, 1u);   // a placement other than 32 per XCD: an abort like those in flight
      }
    }
    return;
  }
  ExchCtl& C = xa.ctl[xcd];
  uint64_t* xring = xa.rings + (size_t)xcd * NS * NS * CAP;        // [owner][producer][CAP]
  const uint64_t* myring = xring + (size_t)me * NS * CAP;          // the rings that end here
  uint32_t* xring_lo = EXACT ? xa.rings_lo + (size_t)xcd * NS * NS * CAP : nullptr;
  const uint32_t* myring_lo = EXACT ? xring_lo + (size_t)me * NS * CAP : nullptr;
  const uint32_t r0 = __builtin_nontemporal_load(xa.win + 0);
  const uint32_t win_rows = (uint32_t)NS * (uint32_t)xa.rows_per;
  const uint32_t L = (uint32_t)xa.row_len;
  const int64_t n = xa.n;
  double* side = xa.side;
  const uint32_t rot_mask = (uint32_t)xa.side_rot_mask;

  const int64_t n_tiles = (n + TILE - 1) / TILE;
  const int64_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  auto tile_base = [&](int64_t k) { return ((int64_t)blockIdx.x + k * gridDim.x) * TILE; };
  // the lane's index, opaque to the optimiser: what is derived from it inside a phase is recomputed there (three
  // instructions) instead of being carried through the whole loop — as registers that spill, and a scratch reload between
  // the issue of a tile's loads and their use waits for the whole prefetch (the memory counter counts in order)
  auto lane_now = [&]() {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
  };
  // a lane's samples of a tile: two pairs, 2048 samples apart (16-byte lane loads, dense across the wavefront)
  d2 xv[D][2], wv[2];
  auto load_tile = [&](int64_t base) {
    if (base + TILE <= n) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t e = (lane_now() * 2u + (uint32_t)u * 2u * BLOCK) * 8u;
#pragma unroll
        for (int d = 0; d < D; ++d) xv[d][u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(reinterpret_cast<const char*>(xa.s_ptr[d] + base) + e));
        wv[u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(reinterpret_cast<const char*>(xa.w_ptr + base) + e));
      }
    } else {  // the tile that holds the end of the input (once per call): positions past the end are NaN samples, dropped
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int64_t i = base + (int64_t)tid * 2 + (int64_t)u * 2 * BLOCK + v;
#pragma unroll
          for (int d = 0; d < D; ++d) xv[d][u][v] = i < n ? xa.s_ptr[d][i] : (double)__builtin_nanf("");
          wv[u][v] = i < n ? xa.w_ptr[i] : 0.0;
        }
    }
  };
  // ---- the consumer side: 32 lanes per ring, CL records each ---------------------------------------------------
  const uint32_t psub = (uint32_t)tid >> 5, l = (uint32_t)tid & 31u;
  uint64_t rr[CL];
  uint32_t rb[EXACT ? CL : 1];
  uint32_t rh = 0;
  auto issue_ring_loads = [&]() {
    const uint32_t t = lane_now(), ps = t >> 5, ll = t & 31u;
    rh = chead[ps];
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      const uint32_t slot = (ps << CAPL) + ((rh + ll + 32u * j) & (uint32_t)(CAP - 1));
      rr[j] = exch_ld(reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(myring) + slot * 8u));
      if constexpr (EXACT) rb[j] = exch_ld(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(myring_lo) + slot * 4u));
    }
  };
  auto take_ring_records = [&]() {  // adds the valid prefix of what was loaded
    const uint32_t tid = lane_now(), psub = tid >> 5, l = tid & 31u;
    uint32_t pre = 0;
    bool cont = true;
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      const uint32_t pos = rh + l + 32u * j;
      const uint32_t expect = ((pos >> CAPL) + 1u) & 3u;
      bool valid = (((uint32_t)rr[j] >> 14) & 3u) == expect;
      if constexpr (EXACT) valid &= ((rb[j] >> 16) & 3u) == expect;
      const uint64_t bal = __builtin_amdgcn_ballot_w64(valid);
      const uint32_t m = (tid & 32) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
      if (cont) {
        if (m == 0xffffffffu) pre += 32u;
        else { pre += (uint32_t)__builtin_ctz(~m); cont = false; }
      }
    }
#pragma unroll
    for (int j = 0; j < CL; ++j)
      if (l + 32u * j < pre) {
        const uint64_t r = rr[j];
        const uint64_t wbits = EXACT ? ((r & ~0xffffull) | (uint64_t)(rb[j] & 0xffffu)) : (r & ~0xffffull);
        unsafeAtomicAdd(hist + ((uint32_t)r & 0x3fffu), __longlong_as_double((long long)wbits));
      }
    if (l == 0 && pre) {
      chead[psub] = rh + pre;
      exch_st(&C.head[psub][me], rh + pre);
    }
  };
  // the same, one record per lane at a time (the waits outside the tile loop's own rhythm: fewer registers, more latency)
  auto take_rolled = [&]() -> uint32_t {
    const uint32_t h = chead[psub];
    uint32_t pre = 0;
    bool cont = true;
#pragma unroll 1
    for (int j = 0; j < CL; ++j) {
      const uint32_t pos = h + l + 32u * j;
      const uint32_t slot = (psub << CAPL) + (pos & (uint32_t)(CAP - 1));
      const uint64_t r = exch_ld(reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(myring) + slot * 8u));
      uint32_t b = 0;
      if constexpr (EXACT) b = exch_ld(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(myring_lo) + slot * 4u));
      bool valid = (((uint32_t)r >> 14) & 3u) == (((pos >> CAPL) + 1u) & 3u);
      if constexpr (EXACT) valid &= ((b >> 16) & 3u) == (((pos >> CAPL) + 1u) & 3u);
      const uint64_t bal = __builtin_amdgcn_ballot_w64(valid);
      const uint32_t m = (tid & 32) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
      const uint32_t kk = !cont ? 0u : (m == 0xffffffffu ? 32u : (uint32_t)__builtin_ctz(~m));
      if (l < kk) unsafeAtomicAdd(hist + ((uint32_t)r & 0x3fffu), __longlong_as_double((long long)(EXACT ? ((r & ~0xffffull) | (uint64_t)(b & 0xffffu)) : (r & ~0xffffull))));
      pre += kk;
      cont = cont && kk == 32u;
    }
    if (l == 0 && pre) {
      chead[psub] = h + pre;
      exch_st(&C.head[psub][me], h + pre);
    }
    return pre;
  };
  auto deadline = [&]() { return wall_clock64() - *reinterpret_cast<const long long*>(misc + 6) > cold->budget_ticks; };

  uint64_t s_neg = 0, s_pos = 0;  // lanes that read a negative / positive weight (pack48 rounds: only one sign may travel packed) — scalar registers
  uint32_t cred_next = 0;
  if (my_tiles > 0) load_tile(tile_base(0));
  int buf = 0;
  bool aborted = false;
  for (int64_t k = 0; k < my_tiles; ++k, buf ^= 1) {
    uint32_t* cnt = cnt2 + buf * 64;
    // ---- digitize; owner and record of every sample; what lies outside the window goes straight to the side copy of the
    // output, with the weight as it was read -------------------------------------------------------------------------
    uint32_t dest[4];  // owner (32: outside the window, 33: dropped)
    uint64_t rec[4];
    uint32_t lo2[EXACT ? 2 : 1];  // EXACT: the 16 low weight bits of the four records, two to a register
    {
      double xs[D][4];
#pragma unroll
      for (int d = 0; d < D; ++d)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) xs[d][s4] = xv[d][s4 >> 1][s4 & 1];
      uint32_t g[D][4];
      bool ins[4];
      exch_digitize<D, 4>(xs, xa, g, ins, cold);
      bool special_any = false;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        uint32_t gg[D], row, col;
#pragma unroll
        for (int d = 0; d < D; ++d) gg[d] = g[d][s4];
        exch_row_col<D>(gg, xa, row, col);
        const uint32_t r = row - r0;
        const bool in_win = ins[s4] & (r < win_rows);
        const double wq = wv[s4 >> 1][s4 & 1];
        if constexpr (EXACT) {  // the weight's bits as they are: 48 with the code, 16 beside it — nothing rounded, no sign to watch
          const uint64_t b = (uint64_t)__double_as_longlong(wq);
          rec[s4] = (b & ~0xffffull) | ((r >> 5) * L + col);
          const uint32_t low = (uint32_t)b & 0xffffu;
          lo2[s4 >> 1] = (s4 & 1) ? (lo2[s4 >> 1] | (low << 16)) : low;
        } else {
          bool special;
          rec[s4] = exch_pack_fast(wq, (r >> 5) * L + col, special);  // (the code of a record that does not travel is never looked at)
          special_any |= special;
        }
        dest[s4] = in_win ? exch_owner(r) : (ins[s4] ? 32u : 33u);
        if (ins[s4] & !in_win) unsafeAtomicAdd(reinterpret_cast<double*>(reinterpret_cast<char*>(side) + exch_side_index(row, col, L, rot_mask) * 8u), wq);
        if constexpr (!EXACT) {
          s_neg |= __builtin_amdgcn_ballot_w64(wq < 0.0);
          s_pos |= __builtin_amdgcn_ballot_w64(wq > 0.0);
        }
      }
      if constexpr (!EXACT) {
        if (__builtin_amdgcn_ballot_w64(special_any) != 0ull) {  // NaN / infinite / top-binade weights: the careful rounding
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) rec[s4] = (uint64_t)__double_as_longlong(pack48(wv[s4 >> 1][s4 & 1], (uint32_t)rec[s4] & 0x3fffu));
        }
      }
    }
    // ---- ring loads first (older), then the next tile's samples (newer): waiting for the former leaves the latter in flight
    issue_ring_loads();
    if (tid < NS) cred_next = exch_ld(&C.head[me][tid]);
    load_tile(tile_base(k + 1 < my_tiles ? k + 1 : k));
    uint32_t rank[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) rank[s4] = dest[s4] < 32u ? atomicAdd(cnt + dest[s4], 1u) : 0u;
    __syncthreads();
    // ---- wavefront 0: block layout, credits, how much of every run may go out; everybody: take what has arrived ----
    if (tid < 64) {
      const uint32_t cn = tid < NS ? cnt[tid] : 0u;
      const uint32_t xs = wave_inclusive_scan_u32(cn);
      off[tid] = xs - cn;
      if (tid < NS) {
        credit[tid] = cred_next;
        const uint32_t t = tail[tid];
        wbase[tid] = t;
        wadj[tid] = t - (xs - cn);
        tail[tid] = t + cn;
        const uint32_t room = (uint32_t)CAP - (t - cred_next);  // (t - credit <= CAP always: nothing is written without room)
        lim[tid] = min(cn, room);
        sent[tid] = 0u;
        if (cn > room) misc[3] = 1u;
      }
      cnt2[(buf ^ 1) * 64 + tid] = 0u;
    }
    take_ring_records();
    __syncthreads();
    // ---- the tile's ring records, by owner, into LDS ---------------------------------------------------------------
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      if (dest[s4] < 32u) {
        const uint32_t i = off[dest[s4]] + rank[s4];
        srec[i] = rec[s4];
        if constexpr (EXACT) slo[i] = (uint16_t)(lo2[s4 >> 1] >> ((s4 & 1) * 16));
        sd[i] = (uint8_t)dest[s4];
      }
    __syncthreads();
    // ---- out: runs of consecutive records per ring; a run without room goes out in pieces as the owner catches up ----
    const uint32_t total = off[32];
    if (!misc[3]) {  // (uniform; the usual case) every run has room: all of the tile's records go out
      const uint32_t t_out = lane_now();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t i = t_out + q * BLOCK;
        if (i < total) {
          const uint32_t d = sd[i];
          const uint32_t pos = wadj[d] + i;
          const uint32_t tag = ((pos >> CAPL) + 1u) & 3u;
          const uint32_t slot = (((d << 5) + me) << CAPL) + (pos & (uint32_t)(CAP - 1));
          *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(xring) + slot * 8u) = (srec[i] & ~0xc000ull) | ((uint64_t)tag << 14);
          if constexpr (EXACT) *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(xring_lo) + slot * 4u) = (uint32_t)slo[i] | (tag << 16);
        }
      }
    } else {
      for (;;) {
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
          const uint32_t i = (uint32_t)tid + q * BLOCK;
          if (i < total) {
            const uint32_t d = sd[i];
            const uint32_t j = i - off[d];
            if (j >= sent[d] && j < lim[d]) {
              const uint32_t pos = wadj[d] + i;
              const uint32_t tag = ((pos >> CAPL) + 1u) & 3u;
              const uint32_t slot = (((d << 5) + me) << CAPL) + (pos & (uint32_t)(CAP - 1));
              *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(xring) + slot * 8u) = (srec[i] & ~0xc000ull) | ((uint64_t)tag << 14);
              if constexpr (EXACT) *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(xring_lo) + slot * 4u) = (uint32_t)slo[i] | (tag << 16);
            }
          }
        }
        if (!misc[3]) break;  // (uniform: changed only between the two barriers below)
        // some run is not out yet: keep taking from the rings that end here (nobody may wait for this workgroup), then look again
        __syncthreads();
        take_rolled();
        if (tid < NS) {
          sent[tid] = lim[tid];
          const uint32_t cr = exch_ld(&C.head[me][tid]);
          credit[tid] = cr;
          const uint32_t cn = tail[tid] - wbase[tid];
          const uint32_t room = (uint32_t)CAP - (wbase[tid] + lim[tid] - cr);
          lim[tid] = min(cn, lim[tid] + room);
        }
        __syncthreads();
        if (tid == 0) {
          bool all = true;
          for (int d = 0; d < NS; ++d) all &= (lim[d] == tail[d] - wbase[d]);
          misc[3] = all ? 0u : 1u;
          if (!all && (deadline() || exch_ld(g_abort))) { misc[4] = 1u; misc[3] = 0u; }
        }
        __syncthreads();
        if (misc[4]) break;
      }
    }
    if (misc[4]) { aborted = true; break; }
  }
  __syncthreads();
  if (!aborted) {
    // ---- no more records from here: say so, then take what is still on its way to this workgroup ----------------
    if (tid < NS) exch_st(&C.fin[me][tid], tail[tid] + 1u);
    for (;;) {
      const uint32_t pre = take_rolled();
      bool done = false;
      if (l == 0 && pre == 0) done = (exch_ld(&C.fin[psub][me]) == chead[psub] + 1u);
      done = __shfl(done ? 1 : 0, (tid & 32), 64) != 0;
      if (tid == 0 && (deadline() || exch_ld(g_abort))) misc[4] = 1u;
      if (__syncthreads_and(done ? 1 : 0)) break;
      if (misc[4]) { aborted = true; break; }
      __syncthreads();
    }
  }
  if (aborted) {
    if (tid == 0) {
      exch_st(g_abort, 1u);
      exch_st(xa.win + 1, 0u);  // the mode is off for this call after all: the classic packed passes queued behind take it
      if (cold->note) atomicAdd(cold->note + 2, 1u);
    }
    return;
  }
  __syncthreads();
  double* po = cold->part + ((size_t)xcd * NS + me) * (size_t)xa.local_bins;
  for (int i = tid; i < xa.local_bins; i += BLOCK) po[i] = hist[i];
  if constexpr (!EXACT) {  // (exact records: whatever the signs, this call's result stands)
    const uint32_t signs = (s_neg ? 1u : 0u) | (s_pos ? 2u : 0u);
    if ((tid & 63) == 0 && signs) atomicOr(cold->flags, signs);
  }
}

// out += side + (inside the window) the eight XCD partials.  Runs when the mode was on and the weights had one sign;
// otherwise the exact passes queued behind fill the output.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) exchange_merge(const ExchArgs xa, double* out, int64_t n_bins) {
  if (__builtin_nontemporal_load(xa.win + 1) == 0u) return;
  if ((__builtin_nontemporal_load(xa.flags) & 3u) == 3u) return;
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= n_bins) return;
  const uint32_t L = (uint32_t)xa.row_len;
  const uint32_t row = (uint32_t)(f / L), col = (uint32_t)(f - (int64_t)row * L);
  const uint32_t r = row - __builtin_nontemporal_load(xa.win + 0);
  double s = xa.side[exch_side_index(row, col, L, (uint32_t)xa.side_rot_mask)];
  if (r < (uint32_t)kExchRings * (uint32_t)xa.rows_per) {
    const uint32_t d = exch_owner(r), local = (r >> 5) * L + col;
#pragma unroll
    for (int xc = 0; xc < kExchXcds; ++xc) s += xa.part[((size_t)xc * kExchRings + d) * (size_t)xa.local_bins + local];
  }
  if (s != 0.0) out[f] += s;
}

}  // namespace xhist
