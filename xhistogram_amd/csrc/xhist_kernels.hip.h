// xhist_kernels.hip.h — CDNA4 (gfx950) kernels of the fused digitize -> joint index -> scatter-add
// hot path.  Written for MI355X only: 64-wide wavefronts, 160 KiB LDS/CU, 256 CUs in 8 XCDs.
//
// What one kernel launch replaces in the reference (/root/reference/xhistogram/core.py):
//   searchsorted(side="right") + right-edge fix-up   core.py:163-174
//   ravel_multi_index over the (E_d+1)-sized axes     core.py:178-183
//   row-offset trick + np.bincount (+ weights)        core.py:73-83
//   trimming the under/overflow (and NaN) bins        core.py:189-192
// The intermediate int64 index arrays of the reference (>= 40 B/sample of DRAM traffic) never
// exist: a sample is read once from HBM (8 B f64 / 4 B f32), binned in registers against edge
// tables staged in LDS, and added to a replicated sub-histogram in LDS with ds_add_u32 /
// ds_add_f64; each workgroup flushes its LDS partial into the output with global atomics.
//
// Exactness of digitize (the hard part of parity): the bin of x is defined by comparisons against
// the caller's exact edge values, never by (x - lo) * inv_width.  A uniform bucket grid over
// [e_0, e_last] is used ONLY as an accelerator: bucket(x) = clamp(int((x - e_0) * scale)) is a
// monotone non-decreasing function of x (IEEE subtraction, multiplication and truncation are all
// monotone), and the per-bucket table entry (start, cnt) is built by applying the SAME device
// function to the edges themselves.  Hence for every x in bucket b:
//     #{j : e_j <= x} = start[b] + #{ j in [start[b], start[b]+cnt[b]) : e_j <= x }
// exactly — edges in lower buckets are certainly <= x, edges in higher buckets certainly > x, and
// the (usually 0 or 1) edges sharing x's bucket are compared explicitly.  No assumption about
// uniformity or about rounding of the bucket arithmetic is needed for correctness.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xhist {

constexpr int kMaxDims = 8;

// dtype tags — must match xhist_dtype in include/xhist_amd.h
enum : int32_t {
  DT_F64 = 0, DT_F32 = 1, DT_F16 = 2, DT_I64 = 3, DT_I32 = 4, DT_I16 = 5, DT_I8 = 6,
  DT_U64 = 7, DT_U32 = 8, DT_U16 = 9, DT_U8 = 10, DT_BOOL = 11
};

// Per-dimension digitize parameters; lives in the kernel-argument segment (-> SGPRs).
struct DimTable {
  double e0_f, eL_f;    // first / last edge, float64 domain
  int64_t e0_i, eL_i;   // first / last edge, int64 domain
  double scale;         // lut_k / (e_last - e_0); 0 when the bucket grid is disabled (lut_k == 1)
  double bias;          // -e_0 * scale: float domains map x -> fma(x, scale, bias) in one instruction
  int32_t n_edges;      // E
  int32_t nb;           // E - 1 real bins
  int32_t lut_k;        // number of buckets K
  int32_t steps;        // binary-search steps inside one bucket: 2^steps > max edges per bucket
  int32_t edge_off;     // offset of this dimension's edges in the table blob, in 8-byte words
  int32_t lut_off;      // offset of this dimension's bucket table in the blob, in 4-byte words
  int64_t out_stride;   // stride of this dimension in the flat real-bin index (C order)
  // arithmetic edges (numpy.linspace: e_j = fl(fl(j * step) + e_0) for j < nb, e_nb given), verified
  // edge by edge at plan creation: digitize needs no table at all (count_le_scan<.., kScanArith>)
  double step, inv_step;
  int32_t arith;
  // 0.5 - delta, where delta bounds |fl(fl(e_j - e_0) * inv_step) - j| over every edge j (measured at plan creation with
  // the kernels' own arithmetic, doubled for margin): a sample whose position t = (x - e_0) * inv_step has a fractional
  // part with |frac - 0.5| < arith_h lies strictly inside bin floor(t) (bin_arith_fast).  0 = never decide by arithmetic.
  double arith_h;
  // float32 SAMPLES on arithmetic edges decided in float32 arithmetic (bin_arith32_fast, SCAN = kScanArith32):
  // t(x) = fmaf(x, a32_scale, a32_bias) is monotone in x, and plan creation has measured |t(B_j) - j| and |t(pred(B_j)) - j|
  // <= delta32 for every float32 bin boundary B_j (the smallest float32 >= e_j; > e_last for the last one: the right-edge rule
  // of core.py:170-173 lives in the boundary), with the same single-rounding fma.  a32_h = 0.5 - delta32; 0 = not offered.
  float a32_scale, a32_bias, a32_h;
  float a32_top;        // nb + 0.5: t is clamped to [-0.5, nb + 0.5] before its floor is taken (see bin_arith32_fast)
  // packed bucket entries (count_le_pack): bucket map of this dimension — 0: linear in (float)x (scale / bias above),
  // 1: the float32 BIT PATTERN of x (an order-preserving integer key, (key - key_lo) >> key_shift): uniform in log x —
  // geometric / logarithmic edges, which a linear grid piles into its first buckets
  int32_t map_kind;
  uint32_t key_lo;
  int32_t key_shift;
  // edges on BOTH sides of zero (a symmetric-log axis): magnitudes below key_floor = the smallest non-zero |edge| are lifted to
  // it before the key is taken (still monotone: everything in (-floor, +floor) holds no edge but possibly 0), and the ~250
  // empty binades between key(-floor) and key(+floor) are cut out of the key space: keys >= key_pos0 move down by key_gap.
  // key_floor = 0: one-sided edges, nothing of this
  float key_floor;
  uint32_t key_pos0, key_gap;
  int32_t is_i64;       // per-dimension domains (Dom<3>): this input compares in int64
  int64_t xor_bias;     // int64 domain of UNSIGNED values: 2^63, flipping the sign bit maps uint64 order onto int64 order
};

struct Params {
  // sample d, logical element (r, c) lives at s_ptr[d][row_offset(r) + c * s_cs[d]] (elements):
  //   row_offset(r) = r * s_rs[d]                                       when s_ir[d] == 0
  //                 = (r / s_ir[d]) * s_os[d] + (r % s_ir[d]) * s_rs[d]  otherwise (rows in groups of
  //                   s_ir: kept axes on both sides of the reduced ones, e.g. (time, LAT, lon) over lat)
  const void* s_ptr[kMaxDims];
  int64_t s_rs[kMaxDims];  // row stride, elements
  int64_t s_cs[kMaxDims];  // col stride, elements
  int64_t s_ir[kMaxDims];  // rows per group (0 = ungrouped)
  int64_t s_os[kMaxDims];  // stride between groups, elements
  int32_t s_dt[kMaxDims];
  const void* w_ptr;       // nullptr = unweighted
  int64_t w_rs, w_cs, w_ir, w_os;
  int32_t w_dt;
  // second weight array of the two-weight variant (same dtype, unit column stride) and its histogram
  const void* w2_ptr;
  int64_t w2_rs, w2_ir, w2_os;
  void* out2;
  int32_t direct_store;    // hist_fast, LDS histograms: this workgroup is the only one of its row — store, do not add
  // bin slice of the SLICED variant: this launch accumulates flat bins [slice_lo, slice_lo + slice_n) only
  int64_t slice_lo;
  int32_t slice_n;
  int32_t lane_rows;       // hist_lanes: rows per workgroup (64 / 128 / 256; 0 = 256) — fewer rows, more column streams
  int32_t lane_pitch;      // hist_lanes: 0 = one counter column per LANE (pitch 257 words per bin); else the lane groups of a row
                           //   share one column per ROW and this is the pitch (rows + 1, or rows / 2 + 1 for uint16 counters)
  int64_t row0;            // first logical row of this launch (inputs only; `out` is pre-advanced)
  int32_t n_dims;
  DimTable dim[kMaxDims];
  const uint64_t* tables;  // device blob: edges of every dimension, then bucket tables
  int32_t table_words;     // blob size in 8-byte words
  int32_t tables_in_lds;   // generic family only: stage the blob in LDS (1) or read it from L2 (0)
  int64_t n_rows, n_cols;
  int64_t n_bins;          // prod(nb_d)
  void* out;               // [n_rows, n_bins] uint64 counts or float64 sums
  int32_t copies_log2;     // LDS sub-histogram replication (lane-private banks)
  int32_t segs;            // workgroups cooperating on one row
  // partitioned mode (histograms too large for LDS), see xhist_partition.hip.h
  uint32_t* part_counts;        // [grid, n_parts]: samples of each workgroup per bin partition
  const uint64_t* part_base;    // [grid, n_parts]: first record slot of a workgroup in each partition stream
  uint16_t* part_codes;         // record stream: bin index inside its partition
  double* part_w;               // record stream: weight (weighted only)
  int32_t part_shift;           // bins per partition = 1 << part_shift
  int32_t parts_per_row;        // routing pass over several rows at once: n_parts = n_rows * parts_per_row (0: one row)
  int32_t n_parts;
};

__host__ __device__ __forceinline__ int64_t row_offset(int64_t r, int64_t rs, int64_t ir, int64_t os) {
  if (ir == 0) return r * rs;
  const int64_t g = r / ir;
  return g * os + (r - g * ir) * rs;
}

// ---------------------------------------------------------------------------------------------
// compare domains
// ---------------------------------------------------------------------------------------------
template <int CMP>
struct Dom;

template <>
struct Dom<0> {  // float64 compares (numpy promotes f32/int samples to f64 against f64 edges)
  using T = double;
  static __device__ __forceinline__ bool in_range(double x, const DimTable& t) {
    return (x >= t.e0_f) & (x <= t.eL_f);  // false for NaN
  }
  static __device__ __forceinline__ double offset(double x, const DimTable& t) { return x - t.e0_f; }
};

template <>
struct Dom<1> {  // exact int64 compares (integer / datetime64 samples against integer edges)
  using T = int64_t;
  static __device__ __forceinline__ bool in_range(int64_t x, const DimTable& t) {
    return (x >= t.e0_i) & (x <= t.eL_i);
  }
  static __device__ __forceinline__ double offset(int64_t x, const DimTable& t) {
    // x >= e0 for every x that matters; the unsigned difference is exact modulo 2^64 and its
    // conversion to double is monotone
    return (double)((uint64_t)x - (uint64_t)t.e0_i);
  }
};

template <>
struct Dom<2> {  // float32 samples against float64 edges, compared EXACTLY in float32:
  // the table holds thr_j = the smallest float32 >= e_j, so for every float32 x
  //   (double)x >= e_j  <=>  x >= thr_j ;   e0_f = thr_0,  eL_f = the largest float32 <= e_last
  // (both exactly representable, kept in the double fields of DimTable)
  using T = float;
  static __device__ __forceinline__ bool in_range(float x, const DimTable& t) {
    return (x >= (float)t.e0_f) & (x <= (float)t.eL_f);
  }
  static __device__ __forceinline__ float offset(float x, const DimTable& t) { return x - (float)t.e0_f; }
};

// per-dimension domains: the sample travels as 64 raw bits (an int64, or the bits of a float64)
template <>
struct Dom<3> {
  using T = int64_t;
};

// Monotone bucket index in [0, K-1] for ANY x (NaN -> 0).  Used by the table builder (on the
// edges) and by digitize (on the samples): the two must be the same code.
template <int CMP>
__device__ __forceinline__ int bucket_of(typename Dom<CMP>::T x, const DimTable& t) {
  // Any monotone non-decreasing map works (see the header comment); a fused multiply-add is
  // monotone in x for scale > 0 and is one instruction.  The table builder calls this very
  // function on the edges, so samples and edges always agree on the map.
  if (CMP == 2) {  // all-float32 arithmetic (full-rate VALU)
    float tt = __builtin_fmaf((float)x, (float)t.scale, (float)t.bias);
    tt = __builtin_amdgcn_fmed3f(tt, 0.0f, (float)(t.lut_k - 1));  // one-op clamp; NaN -> 0
    return (int)tt;
  }
  double tt = CMP == 0 ? __builtin_fma((double)x, t.scale, t.bias) : (double)Dom<CMP>::offset(x, t) * t.scale;
  tt = fmax(fmin(tt, (double)(t.lut_k - 1)), 0.0);  // fmin/fmax drop a NaN operand
  return (int)tt;
}

// digitize, split so that a tile's samples can run the common (one edge per bucket) part as one
// branch-free batch: begin = bucket-table read + first upper_bound step, more = further steps
// (only when some bucket holds more than one edge), end = right-edge fix-up + range check.
struct DigState {
  uint32_t lo;   // edges known to be <= x so far
  uint32_t len;  // edges of x's bucket still undecided
  bool ok;       // x inside [e_0, e_last] and not NaN
};

template <int CMP, typename TabPtr>
__device__ __forceinline__ void upper_bound_step(typename Dom<CMP>::T x, const DimTable& t, TabPtr tab, DigState& s) {
  if constexpr (CMP == 3) {  // per-dimension domain; the branch is uniform (t is a kernel argument)
    if (t.is_i64) upper_bound_step<1>(x, t, tab, s);
    else upper_bound_step<0>(__longlong_as_double(x), t, tab, s);
  } else {
    using T = typename Dom<CMP>::T;
    auto edges = reinterpret_cast<const T*>(tab + t.edge_off);
    const uint32_t half = s.len >> 1;
    const uint32_t mid = s.lo + half;
    const T e = edges[min((int)mid, t.n_edges - 1)];
    const bool le = (s.len != 0u) & (e <= x);
    s.lo = le ? mid + 1u : s.lo;
    s.len = le ? s.len - half - 1u : half;
  }
}

template <int CMP, typename TabPtr>
__device__ __forceinline__ DigState digitize_begin(typename Dom<CMP>::T x, const DimTable& t, TabPtr tab) {
  if constexpr (CMP == 3) {
    return t.is_i64 ? digitize_begin<1>(x, t, tab) : digitize_begin<0>(__longlong_as_double(x), t, tab);
  } else {
    auto lut = reinterpret_cast<const uint32_t*>(tab) + t.lut_off;
    DigState s;
    s.ok = Dom<CMP>::in_range(x, t);
    if (t.lut_k == 0) {  // more than 65535 edges: no bucket table, binary search over all of them
      s.lo = 0u;
      s.len = (uint32_t)t.n_edges;
    } else {
      const uint32_t ent = lut[bucket_of<CMP>(x, t)];
      s.lo = ent & 0xffffu;  // edges below x's bucket: certainly <= x
      s.len = ent >> 16;     // edges sharing the bucket: compared explicitly
    }
    upper_bound_step<CMP>(x, t, tab, s);
    return s;
  }
}

template <int CMP, typename TabPtr>
__device__ __forceinline__ void digitize_more(typename Dom<CMP>::T x, const DimTable& t, TabPtr tab, DigState& s) {
#pragma unroll 1
  for (int k = 1; k < t.steps; ++k) upper_bound_step<CMP>(x, t, tab, s);
}

// Linear form of the in-bucket count for tables whose buckets hold at most SCAN (<= 4) edges —
// every uniform-bin histogram has SCAN = 1.  These kernels use the plan's SECOND table set:
// uint16 `start` entries (no count is needed) on a grid twice as fine for the same LDS bytes.  Each dimension's edge array is followed by 4 NaN
// sentinels, and edges past the bucket's own ones lie in HIGHER buckets, hence are > x: so
//     #{e_j <= x} = start + sum_{k < SCAN} [ e[start + k] <= x ]
// with no count field, no clamping and no data-dependent control flow: SCAN independent LDS
// reads at immediate offsets, SCAN compares, SCAN add-with-carry.
constexpr int kScanArith = 5;

// #{e_j <= x} for arithmetic edges, float64 domain, no table.  g = floor((x - e_0) / step) clamped
// to [0, nb-1] is within one of the answer (plan creation admits only step >= 4 ulp of the largest
// edge magnitude, so an edge sits within 3/8 of a bin of e_0 + j step and the guess within 1e-8 of
// (x - e_0) / step); the two edges around the guess are RECOMPUTED with numpy's own two roundings
// (the empty asm keeps the product from being contracted into an fma) and compared exactly:
//     count = g + [e_g <= x] + [e_{g+1} <= x]
// NaN compares false twice and is dropped by the caller's range test.
__device__ __forceinline__ uint32_t count_le_arith(double x, const DimTable& t) {
  double tt = (x - t.e0_f) * t.inv_step;
  tt = fmax(fmin(tt, (double)(t.nb - 1)), 0.0);
  const double gd = __builtin_floor(tt);
  double m0 = gd * t.step, m1 = (gd + 1.0) * t.step;
  asm volatile("" : "+v"(m0), "+v"(m1));
  const double e_g = m0 + t.e0_f;
  const uint32_t g = (uint32_t)(int)gd;
  const double e_g1 = (int)g + 1 == t.nb ? t.eL_f : m1 + t.e0_f;
  return g + (e_g <= x ? 1u : 0u) + (e_g1 <= x ? 1u : 0u);
}

// Bin of x for arithmetic edges WITHOUT recomputing edges, when x is not within delta bins of one.
//   t(x) = fl(fl(x - e_0) * inv_step) is monotone non-decreasing in x (IEEE subtraction and multiplication by a positive
//   constant are), and plan creation has measured |t(e_j) - j| <= delta for EVERY edge with the same two operations.
//   Hence e_j <= x < e_{j+1} implies j - delta <= t(x) <= j + 1 + delta, and conversely a sample with g = floor(t(x)) and
//   delta < frac(t(x)) < 1 - delta cannot lie below e_g (t(x) <= g + delta would follow) nor at or above e_{g+1}
//   (t(x) >= g + 1 - delta): it is in bin g exactly when 0 <= g < nb, below e_0 when g < 0 (t(x) < 0 means x < e_0) and
//   above e_last when g >= nb.  Everything else — on or next to an edge, NaN, +-inf (frac is NaN) — sets `near` and is
//   decided by count_le_arith's exact compares.  9 full-rate float64 operations per sample and dimension instead of ~25
//   (C5's routing pass spent more than half of its instruction issue on the exact form).
// Returns the bin, or -1 for samples below e_0 / above e_last / NaN; valid only when `near` comes back false.
__device__ __forceinline__ int bin_arith_fast(double x, const DimTable& t, bool& near) {
  double tt = (x - t.e0_f) * t.inv_step;
  asm volatile("" : "+v"(tt));  // the ROUNDED product, as plan creation measured it: left alone, the compiler fuses it into `tt - fl` below
  const double fl = __builtin_floor(tt);
  const double f = tt - fl;
  near = !(__builtin_fabs(f - 0.5) < t.arith_h);  // (NaN compares false: near)
  // fl as an integer without a conversion instruction (quarter rate): the low word of fl + 1.5 * 2^52 is fl mod 2^32
  const double magic = fl + 6755399441055744.0;
  const int g = (int)(uint32_t)(uint64_t)__double_as_longlong(magic);
  const bool inside = (fl >= 0.0) & (fl < (double)t.nb);
  return inside ? g : -1;
}

// The same idea for float32 SAMPLES, in float32 arithmetic (BASELINE C4: (time, lat, lon) float32, 50 uniform bins).  The
// table digitize of such samples costs 12 vector-ALU instructions and three LDS operations per 4-byte sample (bucket, start
// table, threshold, compare, range test, clamp, slot) and keeps the SIMDs 60 % busy at 6.6 TB/s — too close to co-limited for
// a kernel that dask-chunk-sized calls run in the first milliseconds after an idle GPU, while the shader clock dips
// (0.83 of 8 TB/s in a tight loop, 0.71-0.75 cold).  Here:
//   t = fmaf(x, scale, bias)                 one rounding; monotone non-decreasing in x (scale > 0)
//   t = med3(t, -0.5, nb + 0.5)              NaN -> -0.5 (v_med3_f32 returns the minimum when an operand is NaN); +-inf, fill
//                                            values like 1e20 and everything else far outside land on -0.5 / nb + 0.5:
//                                            certainly dropped (t < -0.5 => x < B_0, t > nb + 0.5 => x >= B_nb, by the
//                                            inequalities below) and certainly not `near` — data full of NaNs (land points)
//                                            or sentinels never leave the fast path
//   g = floor(t), f = t - g                  |f - 0.5| < a32_h  =>  x lies strictly inside bin g's float32 boundaries
// Proof as for bin_arith_fast, with the float32 boundaries B_j in place of the edges: B_g <= x would fail only if
// x <= pred(B_g), and then t(x) <= t(pred(B_g)) <= g + delta, i.e. f <= delta; x < B_{g+1} would fail only if x >= B_{g+1},
// and then t(x) >= g + 1 - delta.  g < 0 / g >= nb are below B_0 / at or above B_nb by the same two inequalities.  Plan
// creation measures delta over every B_j AND its float32 predecessor with fmaf (correctly rounded on the host, v_fma_f32 on
// the device: the same number).  6.5 vector-ALU instructions (the fma and the `- 0.5` are packed two samples to an
// instruction) and ONE LDS operation per sample.
// Returns floor(t) in [-1 - pad_k, nb + pad_k] as an unsigned number: [0, nb) when the sample counts, outside when it does not.
// pad_k (0 ... 31, a per-lane constant of the PADDED histogram layout of hist_fast): the clamp is widened to
// [-0.5 - pad_k, nb + 0.5 + pad_k], so that what a lane drops lands in pad bin -1 - pad_k / nb + pad_k — dropped samples of a
// wavefront spread over 32 LDS slots whatever the number of histogram copies (NaN-heavy data, the NaN padding of a ragged
// tile: with ONE copy and one pad bin they met on one address between scattered adds, 3*10^6 samples 12 -> 19 us).
__device__ __forceinline__ uint32_t bin_arith32_fast(float x, const DimTable& t, bool& near, float pad_k = 0.0f) {
  float tt = __builtin_fmaf(x, t.a32_scale, t.a32_bias);
  tt = __builtin_amdgcn_fmed3f(tt, -0.5f - pad_k, t.a32_top + pad_k);
  const float fl = __builtin_floorf(tt);
  const float f = tt - fl;
  near = !(__builtin_fabsf(f - 0.5f) < t.a32_h);
  return (uint32_t)(int)fl;
}
// ... and the exact answer for a `near` sample, same convention: float64 compares against the recomputed edges
__device__ __forceinline__ uint32_t bin_arith32_exact(float x, const DimTable& t) {
  const double xd = (double)x;
  const uint32_t c = count_le_arith(xd, t);
  return Dom<0>::in_range(xd, t) ? min(c, (uint32_t)t.nb) - 1u : 0xffffffffu;
}

// ---------------------------------------------------------------------------------------------
// Packed bucket entries (SCAN = kScanPack2 / kScanPack3; float64 samples, NON-uniform edges — BASELINE C3).
// The table digitize above costs a dimension two DEPENDENT LDS reads (uint16 `start`, then 1-4 float64 edges at
// start) and the address arithmetic between them; C3's kernel keeps its LDS busy 74 % of the time and its VALU 57 %
// (profiles/r03_s_c3_sq_counters.txt).  Here a bucket is ONE aligned 16-byte entry
//     { thr_a, thr_b, thr_c : float32,  start : uint32 }        (one ds_read_b128 per sample and dimension)
// with thr = m(e_j) for the (at most three) edges of the bucket, m(x) = (float)x, NaN where the bucket has fewer.
// m is monotone non-decreasing, so with xf = m(x):   xf > m(e)  =>  x > e   and   xf < m(e)  =>  x < e;
// only xf == m(e) decides nothing (one sample in ~10^5 for C3's edges).  Those samples — which include every x == e_j,
// in particular the right edge x == e_last, and every x that is NaN-adjacent in float32 — set `near`; a wavefront in
// which any lane is near redoes that lane with an exact binary search over the float64 edges (count_le_exact).
// The bucket map runs on xf in float32 (bucket_of<2>), applied by the table builder to m(e_j) with the same code:
// the composition is monotone in x, hence edges of lower buckets are < x and edges of higher buckets > x, as ever.
// Without `near`:  count == 0  <=>  x < e_0,  count == E  <=>  x > e_last (x == e_last is near), NaN counts 0 —
// and the entries hold start - 1, so what comes out is the bin itself and the range test is `bin < nb` (unsigned):
// no float64 compare, no subtraction.
// ---------------------------------------------------------------------------------------------
constexpr int kScanPack2 = 6, kScanPack3 = 7;  // linear bucket map in every dimension, at most 2 / 3 edges per bucket
constexpr int kScanPackG = 8;                    // general: the map is chosen per dimension (DimTable::map_kind), 3 edges per bucket
constexpr bool scan_is_pack(int scan) { return scan == kScanPack2 || scan == kScanPack3 || scan == kScanPackG; }
constexpr int kScanArith32 = 9;  // float32 samples, arithmetic edges, float32 arithmetic (bin_arith32_fast); no tables
// digitize forms whose per-sample result is the BIN itself (>= nb as unsigned when the sample is dropped), not a count of edges
constexpr bool scan_gives_bin(int scan) { return scan_is_pack(scan) || scan == kScanArith32; }

// Order-preserving integer key of a float32: for any a, b (not NaN)  a < b  =>  key(a) < key(b), and -0.0 / +0.0 — equal as
// numbers — get the SAME key (the sample is canonicalised by adding +0.0 first: an edge at 0.0 and a sample -0.0 must
// meet in one bucket).  NaN lands at either end (sign bit), compares false against every threshold and is dropped.
__host__ __device__ __forceinline__ uint32_t float_order_key(float xf) {
  xf += 0.0f;  // -0.0 -> +0.0
  const uint32_t b = __builtin_bit_cast(uint32_t, xf);
  return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}

// bucket of the packed entries under the float-bits map: monotone in xf, hence in x
__host__ __device__ __forceinline__ int bucket_of_key(float xf, uint32_t key_lo, int key_shift, int lut_k, float key_floor = 0.0f,
                                                      uint32_t key_pos0 = 0u, uint32_t key_gap = 0u) {
  if (key_floor > 0.0f) {  // (uniform: a kernel argument) two-sided edges — see DimTable
    xf += 0.0f;            // -0.0 -> +0.0 BEFORE the sign is copied: both zeros go to +floor
    xf = __builtin_copysignf(__builtin_fmaxf(__builtin_fabsf(xf), key_floor), xf);
  }
  uint32_t k = float_order_key(xf);
  if (key_floor > 0.0f && k >= key_pos0) k -= key_gap;
  const uint32_t d = (k > key_lo ? k - key_lo : 0u) >> key_shift;
  return (int)(d < (uint32_t)(lut_k - 1) ? d : (uint32_t)(lut_k - 1));
}
__host__ __device__ __forceinline__ int bucket_of_key(float xf, const DimTable& t) {
  return bucket_of_key(xf, t.key_lo, t.key_shift, t.lut_k, t.key_floor, t.key_pos0, t.key_gap);
}
// the same with the two-sided form decided at compile time: the kernels pick it ONCE per dimension and tile — a branch per
// sample, even a uniform one, splits the register batch into basic blocks and serialises its LDS reads (measured: +17 %)
template <bool TWO_SIDED>
__device__ __forceinline__ int bucket_of_key_t(float xf, const DimTable& t) {
  if (TWO_SIDED) {
    xf += 0.0f;
    xf = __builtin_copysignf(__builtin_fmaxf(__builtin_fabsf(xf), t.key_floor), xf);
  }
  uint32_t k = float_order_key(xf);
  if (TWO_SIDED) k = k >= t.key_pos0 ? k - t.key_gap : k;
  const uint32_t d = (k > t.key_lo ? k - t.key_lo : 0u) >> t.key_shift;
  return (int)min(d, (uint32_t)(t.lut_k - 1));
}
typedef uint32_t pack_entry_t __attribute__((ext_vector_type(4), aligned(16)));

// returns the real-bin index, or a value >= nb (as unsigned) for a sample the reference drops — valid unless `near`
// KEYMAP: the bucket comes from the float32 bit pattern (DimTable::map_kind == 1) instead of the linear map
template <int NP, int KEYMAP, typename TabPtr>
__device__ __forceinline__ uint32_t count_le_pack(double x, const DimTable& t, TabPtr tab, bool& near) {
  float xf = (float)x;
  int b;
  if (KEYMAP) {  // 1: float-bit-pattern grid, 2: its two-sided form (edges on both sides of zero)
    // NaN has no place in the key order (its bit pattern sorts above +inf): it becomes FLT_MAX — beyond every threshold
    // (plan creation admits |e| <= 3e38 only), so it counts every edge and is dropped like any sample above e_last.
    // (the linear map sends NaN to bucket 0, where it counts no edge: dropped as well)
    xf = __builtin_fminf(xf, 3.402823466e+38f);
    b = bucket_of_key_t<KEYMAP == 2>(xf, t);
  } else {
    b = bucket_of<2>(xf, t);
  }
  const pack_entry_t e = reinterpret_cast<const pack_entry_t*>(tab)[t.lut_off + b];
  const float ta = __uint_as_float(e[0]), tb = __uint_as_float(e[1]), tc = __uint_as_float(e[2]);
  uint32_t c = e[3];  // start - 1: the sum below is the BIN, -1 (as 2^32 - 1) below e_0, nb above e_last
  c += (xf > ta) ? 1u : 0u;
  c += (xf > tb) ? 1u : 0u;
  near = (xf == ta) | (xf == tb);
  if (NP == 3) {
    c += (xf > tc) ? 1u : 0u;
    near |= (xf == tc);
  }
  return c;
}

// float32 SAMPLES need no pre-compare and no redo: the entry holds thr_j = the smallest float32 >= e_j, for which
// (double)x >= e_j  <=>  x >= thr_j  exactly (Dom<2>'s argument) — and, for the LAST edge, the smallest float32 > e_last, so
// that x == e_last counts E - 1 edges and lands in the last bin while anything above it counts E and is dropped: the
// right-edge rule of core.py:170-173 sits in the table, not in the kernel.  Buckets come from the same maps applied to thr.
template <int NP, int KEYMAP, typename TabPtr>
__device__ __forceinline__ uint32_t count_le_pack_f32(float x, const DimTable& t, TabPtr tab) {
  int b;
  if (KEYMAP) {
    x = __builtin_fminf(x, 3.402823466e+38f);  // NaN (and +inf) -> FLT_MAX: counts every threshold, dropped (see count_le_pack)
    b = bucket_of_key_t<KEYMAP == 2>(x, t);
  } else {
    b = bucket_of<2>(x, t);
  }
  const pack_entry_t e = reinterpret_cast<const pack_entry_t*>(tab)[t.lut_off + b];
  uint32_t c = e[3];
  c += (x >= __uint_as_float(e[0])) ? 1u : 0u;
  c += (x >= __uint_as_float(e[1])) ? 1u : 0u;
  if (NP == 3) c += (x >= __uint_as_float(e[2])) ? 1u : 0u;
  return c;
}

// exact bin over ALL float64 edges of a dimension, in the packed counts' convention: 2^32 - 1 for dropped samples
// (x < e_0, x > e_last, NaN), else min(#{e_j <= x}, nb) - 1 so that x == e_last lands in the last bin
template <typename TabPtr>
__device__ __forceinline__ uint32_t count_le_exact(double x, const DimTable& t, TabPtr tab) {
  const double* e = reinterpret_cast<const double*>(tab + t.edge_off);
  uint32_t lo = 0u, len = (uint32_t)t.n_edges;
  while (len) {
    const uint32_t half = len >> 1, mid = lo + half;
    const bool le = e[mid] <= x;
    lo = le ? mid + 1u : lo;
    len = le ? len - half - 1u : half;
  }
  return Dom<0>::in_range(x, t) ? min(lo, (uint32_t)t.nb) - 1u : 0xffffffffu;
}

template <int CMP, int SCAN, typename TabPtr>
__device__ __forceinline__ uint32_t count_le_scan(typename Dom<CMP>::T x, const DimTable& t, TabPtr tab) {
  using T = typename Dom<CMP>::T;
  if constexpr (SCAN == kScanArith) {
    static_assert(CMP == 0, "arithmetic edges are compared in float64");
    return count_le_arith(x, t);
  }
  // (two ways of sparing the lanes whose bucket holds no edge — 9 in 10 for uniform bins — their edge read were tried and
  // lost: a flag bit in the table entry with a branch around the reads, 1.18 -> 1.58 ms for the unweighted headline; the
  // same flag redirecting those lanes to conflict-free sentinels, branch-free, 1.18 -> 1.22 and C3 2.50 -> 2.55.  The C3
  // kernel keeps its LDS busy 74 % of the time, two thirds of it in bank-conflict cycles (profiles/r03_s_c3_sq_counters.txt),
  // and is still not bound by it.)
  auto lut = reinterpret_cast<const uint16_t*>(tab) + t.lut_off;  // start-only table, 2-byte entries
  const uint32_t start = lut[bucket_of<CMP>(x, t)];
  const T* e = reinterpret_cast<const T*>(tab + t.edge_off) + start;
  uint32_t lo = start;
#pragma unroll
  for (int k = 0; k < SCAN; ++k) lo += (e[k] <= x) ? 1u : 0u;
  return lo;
}

// Real-bin index in [0, nb) from lo = #{e_j <= x}, or -1 when the reference drops the sample
template <int CMP>
__device__ __forceinline__ int bin_from_count(typename Dom<CMP>::T x, const DimTable& t, uint32_t lo) {
  const int bin = min((int)lo - 1, t.nb - 1);  // x == e_last counts E edges -> last bin
  return Dom<CMP>::in_range(x, t) ? bin : -1;
}

// One sample, packed entries (count_le_pack / count_le_pack_f32): the real-bin index or -1.  For the kernels that digitize
// sample by sample (row-per-lane family) — the float64 redo is a per-lane branch there.
template <int CMP, int SCAN, typename TabPtr>
__device__ __forceinline__ int bin_of_sample_pack(typename Dom<CMP>::T x, const DimTable& t, TabPtr tab) {
  static_assert(scan_is_pack(SCAN) && (CMP == 0 || CMP == 2), "packed entries: float64 / float32 samples");
  constexpr int NP = SCAN == kScanPack2 ? 2 : 3;
  constexpr bool G = SCAN == kScanPackG;
  uint32_t c;
  const int km = !G || !t.map_kind ? 0 : (t.key_floor > 0.0f ? 2 : 1);  // (uniform: kernel arguments)
  if constexpr (CMP == 2) {
    c = km == 0 ? count_le_pack_f32<NP, 0>(x, t, tab) : (km == 1 ? count_le_pack_f32<NP, 1>(x, t, tab) : count_le_pack_f32<NP, 2>(x, t, tab));
  } else {
    bool near;
    c = km == 0 ? count_le_pack<NP, 0>(x, t, tab, near) : (km == 1 ? count_le_pack<NP, 1>(x, t, tab, near) : count_le_pack<NP, 2>(x, t, tab, near));
    if (near) c = count_le_exact(x, t, tab);
  }
  return c < (uint32_t)t.nb ? (int)c : -1;
}

// bin of a sample from what count_le_tile returned for it: a count of edges (table / arithmetic digitize) or, for the packed
// entries, the bin itself (>= nb as unsigned when dropped)
template <int CMP, int SCAN>
__device__ __forceinline__ int bin_from_tile_count(typename Dom<CMP>::T x, const DimTable& t, uint32_t cnt) {
  if constexpr (scan_gives_bin(SCAN)) return cnt < (uint32_t)t.nb ? (int)cnt : -1;
  else return bin_from_count<CMP>(x, t, cnt);
}

// Real-bin index in [0, nb), or -1 when the reference would drop the sample.
// s.lo == #{j : e_j <= x} (searchsorted side="right"); x == e_last gives E -> last bin.
__device__ __forceinline__ int digitize_end(const DimTable& t, const DigState& s) {
  const int bin = min((int)s.lo - 1, t.nb - 1);
  return s.ok ? bin : -1;
}

template <int CMP, typename TabPtr>
__device__ __forceinline__ int digitize(typename Dom<CMP>::T x, const DimTable& t, TabPtr tab) {
  DigState s = digitize_begin<CMP>(x, t, tab);
  digitize_more<CMP>(x, t, tab, s);
  return digitize_end(t, s);
}

// ---------------------------------------------------------------------------------------------
// element loads with conversion into the compare domain / to float64 weights
// ---------------------------------------------------------------------------------------------
template <typename OUT>
__device__ __forceinline__ OUT load_as(const void* p, int32_t dt, int64_t i) {
  switch (dt) {
    case DT_F64: return (OUT) reinterpret_cast<const double*>(p)[i];
    case DT_F32: return (OUT) reinterpret_cast<const float*>(p)[i];
    case DT_F16: return (OUT)(float) reinterpret_cast<const _Float16*>(p)[i];
    case DT_I64: return (OUT) reinterpret_cast<const int64_t*>(p)[i];
    case DT_I32: return (OUT) reinterpret_cast<const int32_t*>(p)[i];
    case DT_I16: return (OUT) reinterpret_cast<const int16_t*>(p)[i];
    case DT_I8: return (OUT) reinterpret_cast<const int8_t*>(p)[i];
    case DT_U64: return (OUT) reinterpret_cast<const uint64_t*>(p)[i];
    case DT_U32: return (OUT) reinterpret_cast<const uint32_t*>(p)[i];
    case DT_U16: return (OUT) reinterpret_cast<const uint16_t*>(p)[i];
    case DT_U8: return (OUT) reinterpret_cast<const uint8_t*>(p)[i];
    default: return (OUT)(reinterpret_cast<const uint8_t*>(p)[i] != 0);  // DT_BOOL
  }
}

// a sample in its compare domain; per-dimension domains (3) carry an int64 or the bits of a float64
template <int CMP>
__device__ __forceinline__ typename Dom<CMP>::T load_dom(const void* p, int32_t dt, int64_t i, const DimTable& t) {
  if constexpr (CMP == 3) return t.is_i64 ? (load_as<int64_t>(p, dt, i) ^ t.xor_bias) : __double_as_longlong(load_as<double>(p, dt, i));
  else if constexpr (CMP == 1) return load_as<int64_t>(p, dt, i) ^ t.xor_bias;
  else return load_as<typename Dom<CMP>::T>(p, dt, i);
}

// ---------------------------------------------------------------------------------------------
// accumulators: uint32 counts / float64 weight sums in LDS, uint64 / float64 in global memory
// ---------------------------------------------------------------------------------------------
struct NoWeight {};

template <typename WT>
struct Acc {  // weighted
  using lds_t = double;
  using out_t = double;
  static constexpr int kMaxCopiesLog2 = 4;  // 16 lanes x 8 B cover the 32 LDS write banks
  static __device__ __forceinline__ void lds_add(lds_t* h, uint32_t i, double w) { unsafeAtomicAdd(h + i, w); }
  static __device__ __forceinline__ void out_add(out_t* o, int64_t i, double w) { unsafeAtomicAdd(o + i, w); }
};

template <>
struct Acc<NoWeight> {  // unweighted
  using lds_t = uint32_t;
  using out_t = unsigned long long;
  static constexpr int kMaxCopiesLog2 = 5;  // 32 lanes x 4 B cover the 32 LDS write banks
  static __device__ __forceinline__ void lds_add(lds_t* h, uint32_t i, uint32_t) { atomicAdd(h + i, 1u); }
  static __device__ __forceinline__ void out_add(out_t* o, int64_t i, unsigned long long v) { atomicAdd(o + i, v); }
};

extern __shared__ __attribute__((aligned(16))) unsigned char xhist_smem[];

// stage the table blob (edges + bucket tables) in LDS; returns the LDS copy
__device__ __forceinline__ const uint64_t* stage_tables(const Params& p) {
  uint64_t* dst = reinterpret_cast<uint64_t*>(xhist_smem);
  for (int i = threadIdx.x; i < p.table_words; i += blockDim.x) dst[i] = p.tables[i];
  return dst;
}

// ---------------------------------------------------------------------------------------------
// FAST family: homogeneous float samples (f64 / f32), contiguous columns, 16-byte vector loads.
//   ST sample type, WT weight type (NoWeight / double / float), D inputs, VEC elements per load,
//   UNROLL loads in flight per input per lane, LDS_HIST: sub-histograms in LDS (else global atomics)
// Work split: workgroups (row, seg); the `segs` workgroups of a row walk its tiles interleaved
// (tile = seg, seg + segs, ...) so that the whole grid sweeps HBM as one front.
// ---------------------------------------------------------------------------------------------
// N-element vectors that may sit at any ELEMENT-aligned address: gfx950 global loads need only
// dword alignment, so rows whose stride is not a multiple of 16 bytes (C = 365, 3650, ...) still
// take 16-byte loads.
template <typename T, int N>
struct VecOf { typedef T type __attribute__((ext_vector_type(N), aligned(sizeof(T)))); };

// Four consecutive elements of ANY supported dtype as float64 — what numpy's promotion does to float32 / integer
// samples against float64 edges (core.py:170) and np.bincount to weights: the loads of the MIXED variant of hist_fast
// (inputs of different dtypes, integer weights).  Two steps, so that every load of a tile is in flight before the first
// conversion waits for one: load4_raw picks the load width from the element SIZE (one to two 16-byte loads, one 8-byte,
// one 4-byte; 1- and 2-byte rows arrive dword-aligned, the host side sees to that), raw4_as_double converts by dtype.
typedef double double4_t __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t raw_u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t raw_u2 __attribute__((ext_vector_type(2), aligned(4)));
struct Raw4 {
  raw_u4 a, b;
};
__device__ __forceinline__ Raw4 load4_raw(const unsigned char* base, int es, int64_t i) {
  Raw4 r;
  const unsigned char* p = base + i * es;
  if (es == 8) {
    r.a = __builtin_nontemporal_load(reinterpret_cast<const raw_u4*>(p));
    r.b = __builtin_nontemporal_load(reinterpret_cast<const raw_u4*>(p) + 1);
  } else if (es == 4) {
    r.a = __builtin_nontemporal_load(reinterpret_cast<const raw_u4*>(p));
  } else if (es == 2) {
    const raw_u2 q = __builtin_nontemporal_load(reinterpret_cast<const raw_u2*>(p));
    r.a[0] = q[0];
    r.a[1] = q[1];
  } else {
    r.a[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
  }
  return r;
}
// one dtype switch per FOUR elements
__device__ __forceinline__ double4_t raw4_as_double(const Raw4& r, int32_t dt) {
  double4_t o;
  auto w64 = [&](int v) { return v < 2 ? (((uint64_t)r.a[2 * v + 1] << 32) | r.a[2 * v]) : (((uint64_t)r.b[2 * v - 3] << 32) | r.b[2 * v - 4]); };
  auto w16 = [&](int v) { return (r.a[v >> 1] >> ((v & 1) * 16)) & 0xffffu; };
  auto w8 = [&](int v) { return (r.a[0] >> (v * 8)) & 0xffu; };
#define XH_CVT4(EXPR)            \
  _Pragma("unroll") for (int v = 0; v < 4; ++v) o[v] = (EXPR); \
  break;
  switch (dt) {
    case DT_F64: XH_CVT4(__longlong_as_double((long long)w64(v)))
    case DT_F32: XH_CVT4((double)__uint_as_float(r.a[v]))
    case DT_F16: XH_CVT4((double)(float)__builtin_bit_cast(_Float16, (uint16_t)w16(v)))
    case DT_I64: XH_CVT4((double)(int64_t)w64(v))
    case DT_I32: XH_CVT4((double)(int32_t)r.a[v])
    case DT_I16: XH_CVT4((double)(int16_t)(uint16_t)w16(v))
    case DT_I8: XH_CVT4((double)(int8_t)(uint8_t)w8(v))
    case DT_U64: XH_CVT4((double)w64(v))
    case DT_U32: XH_CVT4((double)r.a[v])
    case DT_U16: XH_CVT4((double)w16(v))
    case DT_U8: XH_CVT4((double)w8(v))
    default: XH_CVT4(w8(v) != 0u ? 1.0 : 0.0)  // DT_BOOL: any non-zero byte is True
  }
#undef XH_CVT4
  return o;
}
__device__ __forceinline__ int dt_size(int32_t dt) {
  return (dt == DT_F64 || dt == DT_I64 || dt == DT_U64) ? 8 : ((dt == DT_F32 || dt == DT_I32 || dt == DT_U32) ? 4 : ((dt == DT_F16 || dt == DT_I16 || dt == DT_U16) ? 2 : 1));
}

// #{edges <= x} for every sample of a register tile, as ONE branch-free batch: the table reads of
// all VEC x UNROLL x D samples are independent and can be in flight together.  Shared by every
// vector kernel (hist_fast, part_count).
template <int CMP, int SCAN, int D, int UNROLL, int VEC, typename XV, typename TabPtr>
__device__ __forceinline__ void count_le_tile(const XV (&xv)[D][UNROLL], const Params& p, TabPtr tab, int max_steps,
                                               uint32_t (&cnt)[D][UNROLL][VEC], float pad_k = 0.0f) {
  using CT = typename Dom<CMP>::T;
  if constexpr (SCAN == kScanArith) {
    // arithmetic edges: the bin by arithmetic alone for every sample that is not within delta bins of an edge
    // (bin_arith_fast: 9 float64 operations instead of ~25); a wavefront in which some lane met such a sample — or NaN,
    // +-inf — redoes that lane's batch with the exact compares.  Counts of dropped samples are 0 (callers test the range).
    bool near_any = false;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          bool near;
          cnt[d][u][v] = (uint32_t)(bin_arith_fast((double)xv[d][u][v], p.dim[d], near) + 1);
          near_any |= near;
        }
    if (__builtin_amdgcn_ballot_w64(near_any) != 0ull) {
      if (near_any) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
          for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) cnt[d][u][v] = count_le_arith((double)xv[d][u][v], p.dim[d]);
      }
    }
  } else if constexpr (SCAN == kScanArith32) {
    // float32 samples, arithmetic edges: the bin in float32 arithmetic (bin_arith32_fast); a wavefront in which some lane met
    // a sample next to a boundary (one in ~10^5), NaN or +-inf redoes that lane's batch in float64
    bool near_any = false;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      DimTable t = p.dim[d];
      asm volatile("" : "+s"(t.a32_scale), "+s"(t.a32_bias), "+s"(t.a32_h), "+s"(t.a32_top));  // (scalar registers: not copied into VGPRs ahead of the loop)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          bool near;
          cnt[d][u][v] = bin_arith32_fast((float)xv[d][u][v], t, near, pad_k);
          near_any |= near;
        }
    }
    // The redo is per SAMPLE SLOT, behind a wave-uniform branch each: with 1000 bins a sample is `near` with probability
    // 6e-4, so every second wavefront-batch of 1024-2048 samples holds one — redoing the whole batch of its lane in float64
    // (16-32 samples x ~40 instructions) doubled the time of launch-bound calls (3*10^6 samples: 14 -> 28 us); finding the
    // slot again costs 5 float32 instructions per slot, and the float64 compares run for the one or two slots that need them.
    if (__builtin_amdgcn_ballot_w64(near_any) != 0ull) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        DimTable t = p.dim[d];
        asm volatile("" : "+s"(t.a32_scale), "+s"(t.a32_bias), "+s"(t.a32_h), "+s"(t.a32_top));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            bool near;
            (void)bin_arith32_fast((float)xv[d][u][v], t, near, pad_k);
            if (__builtin_amdgcn_ballot_w64(near) != 0ull) {
              const uint32_t e = bin_arith32_exact((float)xv[d][u][v], p.dim[d]);
              cnt[d][u][v] = near ? e : cnt[d][u][v];
            }
          }
      }
    }
  } else if constexpr (scan_is_pack(SCAN) && CMP == 2) {  // float32 samples: exact in one compare per threshold
    constexpr int NP = SCAN == kScanPack2 ? 2 : 3;
    constexpr bool G = SCAN == kScanPackG;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int km = !G || !p.dim[d].map_kind ? 0 : (p.dim[d].key_floor > 0.0f ? 2 : 1);  // ONE uniform branch per dimension and tile
#define XH_PACK_F32_BATCH(KM)                                      \
  _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)               \
  _Pragma("unroll") for (int v = 0; v < VEC; ++v) cnt[d][u][v] = count_le_pack_f32<NP, KM>((float)xv[d][u][v], p.dim[d], tab);
      if (km == 0) { XH_PACK_F32_BATCH(0) }
      else if (km == 1) { XH_PACK_F32_BATCH(1) }
      else { XH_PACK_F32_BATCH(2) }
#undef XH_PACK_F32_BATCH
    }
  } else if constexpr (scan_is_pack(SCAN)) {
    static_assert(CMP == 0, "packed entries: float64 or float32 samples");
    constexpr int NP = SCAN == kScanPack2 ? 2 : 3;
    constexpr bool G = SCAN == kScanPackG;
    bool near_any = false;  // (a lane mask in SGPRs: OR-ing the samples' flags costs the vector ALU nothing)
    // (general variant: ONE uniform branch per dimension and tile picks that dimension's bucket map for the whole batch)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int km = !G || !p.dim[d].map_kind ? 0 : (p.dim[d].key_floor > 0.0f ? 2 : 1);
#define XH_PACK_F64_BATCH(KM)                                      \
  _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)               \
  _Pragma("unroll") for (int v = 0; v < VEC; ++v) {                \
    bool near;                                                     \
    cnt[d][u][v] = count_le_pack<NP, KM>((double)xv[d][u][v], p.dim[d], tab, near); \
    near_any |= near;                                              \
  }
      if (km == 0) { XH_PACK_F64_BATCH(0) }
      else if (km == 1) { XH_PACK_F64_BATCH(1) }
      else { XH_PACK_F64_BATCH(2) }
#undef XH_PACK_F64_BATCH
    }
    if (__builtin_amdgcn_ballot_w64(near_any) != 0ull) {  // rare: a sample whose float32 image equals an edge's
      if (near_any) {  // which of this lane's samples it was is found again here, not carried through the fast path
#pragma unroll
        for (int d = 0; d < D; ++d) {  // (unrolled: a run-time index into the register tile would send it to scratch)
          const int km = !G || !p.dim[d].map_kind ? 0 : (p.dim[d].key_floor > 0.0f ? 2 : 1);
#pragma unroll
          for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              bool near;
              if (km == 0) (void)count_le_pack<NP, 0>((double)xv[d][u][v], p.dim[d], tab, near);
              else if (km == 1) (void)count_le_pack<NP, 1>((double)xv[d][u][v], p.dim[d], tab, near);
              else (void)count_le_pack<NP, 2>((double)xv[d][u][v], p.dim[d], tab, near);
              if (near) cnt[d][u][v] = count_le_exact((double)xv[d][u][v], p.dim[d], tab);
            }
        }
      }
    }
  } else if constexpr (SCAN > 0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int d = 0; d < D; ++d) cnt[d][u][v] = count_le_scan<CMP, SCAN>((CT)xv[d][u][v], p.dim[d], tab);
  } else {  // crowded buckets (duplicate / very uneven edges): branch-free binary search
    DigState st[D][UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int d = 0; d < D; ++d) st[d][u][v] = digitize_begin<CMP>((CT)xv[d][u][v], p.dim[d], tab);
#pragma unroll 1
    for (int k = 1; k < max_steps; ++k) {  // a round is a no-op once a sample's bucket is decided
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v)
#pragma unroll
          for (int d = 0; d < D; ++d) upper_bound_step<CMP>((CT)xv[d][u][v], p.dim[d], tab, st[d][u][v]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int d = 0; d < D; ++d) cnt[d][u][v] = st[d][u][v].lo;
  }
}

// HIST: where a workgroup accumulates
//   kHistGlobal  device-scope atomics straight into the output (histogram too large for LDS)
//   kHistLds     replicated sub-histograms in LDS, one copy per lane bank (uint32 / float64)
//   kHistPacked  unweighted only: ONE sub-histogram of uint16 counters packed two per LDS word,
//                which lets a 256x256 joint histogram (128 KiB) live in the 160 KiB LDS of a CU.
//                Exactness under overflow: every increment is a RETURNING ds_add; the lane whose
//                own add wraps a 16-bit half (it sees the old half == 0xFFFF) books the lost 2^16
//                to the output with a global atomic, and when the wrapped half is the low one it
//                also books -1 to the neighbour bin whose half received the carry (and that
//                half's own wrap if the carry caused one).  Word values are sums of addends mod
//                2^32, so the result is independent of interleaving: no timing assumption.
constexpr int kHistGlobal = 0, kHistLds = 1, kHistPacked = 2;

// SCAN: 1..4 = linear in-bucket count over at most SCAN edges (count_le_scan), 0 = binary search
// W2: two weight arrays binned in ONE pass over the samples ("mean of A in the bins of x" =
//     sum(A w) / sum(w)): a second replicated LDS histogram, written to p.out2 — the samples are
//     read and digitized once instead of twice (reference TODO, xarray.py:106)
// SLICED: a histogram of up to a few times the LDS capacity is built in several launches, each
//     streaming ALL samples but keeping only the flat bins [slice_lo, slice_lo + slice_n) in LDS
//     (the rest go to the trash slot).  S passes cost S x the streaming time; the partitioned mode
//     moves ~3-4x the algorithmic bytes, so slices win up to S = 3-4 and work for any number of rows.
// I64DOM: int64 / datetime64 samples compared exactly in int64 against integer edges (Dom<1>)
// MIXED: the inputs (and the weights) may be of ANY dtype each (p.s_dt / p.w_dt); they are loaded four elements at a
//     time in their own type and consumed as float64 (ST = double, WT = double or NoWeight, VEC = 4) — float32 next to
//     float64 in a joint histogram, integer samples in a joint histogram, integer weights
template <typename ST, typename WT, int D, int VEC, int UNROLL, int HIST, int SCAN, bool W2 = false, bool SLICED = false, bool I64DOM = false,
          bool MIXED = false>
__global__ void __launch_bounds__(1024) hist_fast(const Params p) {
  constexpr bool LDS_HIST = HIST == kHistLds;
  static_assert(!MIXED || (__is_same(ST, double) && VEC == 4 && !W2 && !SLICED && !I64DOM && HIST == kHistLds), "mixed dtypes: float64 domain, LDS histograms");
  static_assert(!SLICED || HIST != kHistGlobal, "slices are for LDS-resident histograms");
  static_assert(!W2 || (HIST == kHistLds && !__is_same(WT, NoWeight)), "two weights: weighted, LDS histograms");
  // float32 samples: float32-threshold tables — except with arithmetic edges, which are float64
  constexpr int CMP = I64DOM ? 1 : ((__is_same(ST, float) && SCAN != kScanArith) ? 2 : 0);
  static_assert(!I64DOM || (__is_same(ST, int64_t) && SCAN == 0), "int64 domain: int64 samples, (start, cnt) tables");
  static_assert(!scan_is_pack(SCAN) || ((__is_same(ST, double) || __is_same(ST, float)) && !MIXED && !I64DOM), "packed bucket entries: float64 / float32 samples");
  static_assert(SCAN != kScanArith32 || (__is_same(ST, float) && !MIXED && !I64DOM), "float32 arithmetic digitize: float32 samples");
  using CT = typename Dom<CMP>::T;
  // float samples: positions past the end of a ragged tile become NaN (dropped by digitize);
  // integer samples: zero, masked by the past_end bit
  constexpr bool kFloatSamples = __is_same(ST, float) || __is_same(ST, double) || __is_same(ST, _Float16);
  const ST kPastEnd = kFloatSamples ? (ST)__builtin_nanf("") : (ST)0;
  using A = Acc<WT>;
  using lds_t = typename A::lds_t;
  using out_t = typename A::out_t;
  using svec = typename VecOf<ST, VEC>::type;
  constexpr bool kWeighted = !__is_same(WT, NoWeight);
  using wscalar = typename std::conditional<kWeighted, WT, float>::type;
  using wvec = typename VecOf<wscalar, VEC>::type;

  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x / p.segs;
  const int seg = blockIdx.x % p.segs;

  const uint64_t* tab = stage_tables(p);
  lds_t* hist = reinterpret_cast<lds_t*>(xhist_smem + (size_t)p.table_words * 8);
  const uint32_t cmask = (1u << p.copies_log2) - 1u;
  const uint32_t mycopy = (uint32_t)tid & cmask;
  const uint32_t hb = SLICED ? (uint32_t)p.slice_n : (uint32_t)p.n_bins;  // bins resident in LDS
  // samples the reference drops (out of range, NaN) still issue their LDS atomic, on one of 32 trash
  // slots picked by lane: with a single copy they would otherwise all meet on ONE address and
  // serialise (10^9 samples, 90 % out of range: 4.9 ms against 2.4)
  // PADDED (one float32 input digitized by bin_arith32_fast, whose result is floor(t) clamped to [-pad, nb + pad)): the
  // replicated histogram has `pad` = max(1, 32 / copies) bins in front and as many behind — 32 slots each way, the trash —
  // so the slot address is ONE shift-add of the result: no range compare, no select.  A lane's clamp is widened by
  // pad_k = (lane mod 32) / copies bins, which spreads what a wavefront drops over all 32 slots.
  constexpr bool PADDED = SCAN == kScanArith32 && D == 1 && LDS_HIST && !SLICED;
  const uint32_t pad_bins = PADDED ? max(1u, 32u >> p.copies_log2) : 0u;
  const float pad_k = PADDED ? (float)(((uint32_t)tid & 31u) >> p.copies_log2) : 0.0f;
  const uint32_t trash = PADDED ? mycopy : (hb << p.copies_log2) + ((uint32_t)tid & 31u);
  uint32_t* packed = reinterpret_cast<uint32_t*>(hist);
  const uint32_t hist_elems = PADDED ? ((hb + 2u * pad_bins) << p.copies_log2) : (hb << p.copies_log2) + 32u;  // one replicated histogram (+ trash slots)
  lds_t* hist2 = hist + hist_elems;                                          // W2: the second weight's
  if (LDS_HIST) {
    const uint32_t n = hist_elems * (W2 ? 2u : 1u);
    for (uint32_t i = tid; i < n; i += blockDim.x) hist[i] = (lds_t)0;
  }
  const uint32_t packed_words = (hb + 1u) >> 1;  // + 32 trash words
  if (HIST == kHistPacked) {
    for (uint32_t i = tid; i < packed_words + 32u; i += blockDim.x) packed[i] = 0u;
  }
  __syncthreads();

  const ST* sp[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    sp[d] = reinterpret_cast<const ST*>(p.s_ptr[d]) + row_offset(p.row0 + row, p.s_rs[d], p.s_ir[d], p.s_os[d]);
  const wscalar* wp = nullptr;
  if (kWeighted) wp = reinterpret_cast<const wscalar*>(p.w_ptr) + row_offset(p.row0 + row, p.w_rs, p.w_ir, p.w_os);
  const wscalar* wp2 = nullptr;
  if (W2) wp2 = reinterpret_cast<const wscalar*>(p.w2_ptr) + row_offset(p.row0 + row, p.w2_rs, p.w2_ir, p.w2_os);
  // MIXED: byte pointers to the row, element sizes from the dtype tags
  const unsigned char* mp[D];
  const unsigned char* mw = nullptr;
  if constexpr (MIXED) {
#pragma unroll
    for (int d = 0; d < D; ++d)
      mp[d] = reinterpret_cast<const unsigned char*>(p.s_ptr[d]) + row_offset(p.row0 + row, p.s_rs[d], p.s_ir[d], p.s_os[d]) * dt_size(p.s_dt[d]);
    if (kWeighted) mw = reinterpret_cast<const unsigned char*>(p.w_ptr) + row_offset(p.row0 + row, p.w_rs, p.w_ir, p.w_os) * dt_size(p.w_dt);
  }
  int msz[D + 1];
  if constexpr (MIXED) {
#pragma unroll
    for (int d = 0; d < D; ++d) msz[d] = dt_size(p.s_dt[d]);
    msz[D] = kWeighted ? dt_size(p.w_dt) : 0;
  }
  out_t* out = reinterpret_cast<out_t*>(p.out) + row * p.n_bins + (SLICED ? p.slice_lo : 0);  // bin 0 of the slice
  out_t* out2 = W2 ? reinterpret_cast<out_t*>(p.out2) + row * p.n_bins + (SLICED ? p.slice_lo : 0) : nullptr;

  // D == 1 fast scatter (see the tile loop): this lane's copy of bin -1, and its trash slot
  const uint32_t slot_shift = (uint32_t)p.copies_log2 + (sizeof(lds_t) == 8 ? 3u : 2u);
  unsigned char* slot_base0 = reinterpret_cast<unsigned char*>(hist + mycopy) + ((size_t)pad_bins << slot_shift);  // this lane's copy of bin 0
  unsigned char* slot_base = slot_base0 - ((size_t)1 << slot_shift);             // ... of "bin -1": addressed by edge counts
  lds_t* trash_slot = hist + trash;
  auto scatter = [&](bool ok, uint32_t flat, double w, double w2) {
    if (LDS_HIST) {
      if constexpr (SLICED) {
        // most samples belong to other slices: a shared trash slot would serialise their atomics
        // on one address, so they are predicated off instead
        if (ok) A::lds_add(hist, (flat << p.copies_log2) + mycopy, w);
      } else {
        const uint32_t idx = ok ? ((flat << p.copies_log2) + mycopy) : trash;
        A::lds_add(hist, idx, w);
        if constexpr (W2) A::lds_add(hist2, idx, w2);
      }
    } else if (ok) {
      if (kWeighted) A::out_add(out, (int64_t)flat, w);
      else A::out_add(out, (int64_t)flat, 1);
    }
  };
  // packed mode, step 1: returning add of 1 into the sample's 16-bit half (0 for a dropped sample)
  auto packed_add = [&](bool ok, uint32_t flat) -> uint32_t {
    if constexpr (SLICED) {
      if (!ok) return 0u;  // (see scatter: no shared dummy address for the samples of other slices)
    }
    const uint32_t idx = ok ? (flat >> 1) : packed_words + ((uint32_t)tid & 31u);
    const uint32_t inc = ok ? (1u << ((flat & 1u) << 4)) : 0u;
    return atomicAdd(packed + idx, inc);
  };
  // packed mode, step 2: book the wraps this lane's own add caused (rare)
  auto packed_fix = [&](bool ok, uint32_t flat, uint32_t old) {
    if (!ok) return;
    const bool hi = flat & 1u;
    const uint32_t half = hi ? (old >> 16) : (old & 0xffffu);
    if (half == 0xffffu) {
      atomicAdd(reinterpret_cast<unsigned long long*>(out) + flat, 65536ull);
      if (!hi && flat + 1u < hb) {  // the carry landed in the neighbour's half
        unsigned long long fix = ~0ull;            // -1
        if ((old >> 16) == 0xffffu) fix += 65536ull;  // ...and wrapped it too
        atomicAdd(reinterpret_cast<unsigned long long*>(out) + flat + 1, fix);
      }
    }
  };
  int max_steps = 1;
#pragma unroll
  for (int d = 0; d < D; ++d) max_steps = max(max_steps, p.dim[d].steps);

  const int64_t tile_elems = (int64_t)blockDim.x * VEC * UNROLL;
  const int64_t n_tiles = (p.n_cols + tile_elems - 1) / tile_elems;
  for (int64_t tile = seg; tile < n_tiles; tile += p.segs) {
    const int64_t base = tile * tile_elems;
    svec xv[D][UNROLL];
    wvec wv[UNROLL], wv2[UNROLL];
    uint32_t past_end = 0;  // bit (u * VEC + v): that sample lies beyond the row (ragged tile only)
    static_assert(UNROLL * VEC <= 32, "past_end is a 32-bit mask");
    if constexpr (MIXED) {
      Raw4 mraw[D + 1][UNROLL];
      uint32_t mfull = 0;  // bit u: the u-th vector of this lane was loaded whole and waits in mraw
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + ((int64_t)u * blockDim.x + tid) * VEC;
        if (i + VEC <= p.n_cols) {
#pragma unroll
          for (int d = 0; d < D; ++d) mraw[d][u] = load4_raw(mp[d], msz[d], i);
          if constexpr (kWeighted) mraw[D][u] = load4_raw(mw, msz[D], i);
          mfull |= 1u << u;
        } else {  // the ragged end of a row: element by element, NaN past it
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const bool in = i + v < p.n_cols;
#pragma unroll
            for (int d = 0; d < D; ++d) xv[d][u][v] = in ? (ST)load_as<double>(mp[d], p.s_dt[d], i + v) : (ST)__builtin_nanf("");
            if constexpr (kWeighted) wv[u][v] = in ? (wscalar)load_as<double>(mw, p.w_dt, i + v) : (wscalar)0;
          }
        }
      }
      // every load of the tile is in flight: now the conversions
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if ((mfull >> u) & 1u) {
#pragma unroll
          for (int d = 0; d < D; ++d) {
            const double4_t q = raw4_as_double(mraw[d][u], p.s_dt[d]);
#pragma unroll
            for (int v = 0; v < VEC; ++v) xv[d][u][v] = (ST)q[v];
          }
          if constexpr (kWeighted) {
            const double4_t q = raw4_as_double(mraw[D][u], p.w_dt);
#pragma unroll
            for (int v = 0; v < VEC; ++v) wv[u][v] = (wscalar)q[v];
          }
        }
    } else if (base + tile_elems <= p.n_cols) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + ((int64_t)u * blockDim.x + tid) * VEC;
#pragma unroll
        for (int d = 0; d < D; ++d)
          xv[d][u] = __builtin_nontemporal_load(reinterpret_cast<const svec*>(sp[d] + i));
        if (kWeighted) wv[u] = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(wp + i));
        if (W2) wv2[u] = __builtin_nontemporal_load(reinterpret_cast<const wvec*>(wp2 + i));
      }
    } else {
      // ragged last tile (for a short row: the whole row): vectors that fit are loaded whole, the
      // rest element-wise, and positions past the end become NaN samples, which digitize drops —
      // so the batch below runs unchanged
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + ((int64_t)u * blockDim.x + tid) * VEC;
        if (i + VEC <= p.n_cols) {
#pragma unroll
          for (int d = 0; d < D; ++d) xv[d][u] = *reinterpret_cast<const svec*>(sp[d] + i);
          if (kWeighted) wv[u] = *reinterpret_cast<const wvec*>(wp + i);
          if (W2) wv2[u] = *reinterpret_cast<const wvec*>(wp2 + i);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const bool in = i + v < p.n_cols;
#pragma unroll
            for (int d = 0; d < D; ++d) xv[d][u][v] = in ? sp[d][i + v] : kPastEnd;
            if (kWeighted) wv[u][v] = in ? wp[i + v] : (wscalar)0;
            if (W2) wv2[u][v] = in ? wp2[i + v] : (wscalar)0;
            if (!in) past_end |= 1u << (u * VEC + v);  // integer samples have no NaN: masked below
          }
        }
      }
    }
    {
      uint32_t cnt[D][UNROLL][VEC];  // #{edges <= x} per sample and dimension
      count_le_tile<CMP, SCAN, D, UNROLL, VEC>(xv, p, tab, max_steps, cnt, pad_k);
      bool okv[UNROLL][VEC];
      uint32_t flatv[UNROLL][VEC], oldv[UNROLL][VEC];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          if constexpr (PADDED) {
            // [-pad, nb + pad) -> the pad bins catch the dropped.  The offset is SIGNED: dropped samples carry bins -1 - k, and an
            // unsigned 32-bit offset would only land in the front pad bins as long as the address arithmetic itself is 32-bit
            lds_t* slot = reinterpret_cast<lds_t*>(slot_base0 + (ptrdiff_t)(int32_t)(cnt[0][u][v] << slot_shift));
            if (kWeighted) {
              unsafeAtomicAdd(reinterpret_cast<double*>(slot), (double)wv[u][v]);
              if constexpr (W2) unsafeAtomicAdd(reinterpret_cast<double*>(slot) + hist_elems, (double)wv2[u][v]);
            } else {
              atomicAdd(reinterpret_cast<uint32_t*>(slot), 1u);
            }
            continue;
          }
          if constexpr (D == 1 && LDS_HIST && !SLICED) {
            // one input, LDS histogram: the slot address comes straight from the edge count
            //   bin = min(cnt, nb) - 1  ->  byte offset (min(cnt, nb) << sh) from a base moved back
            //   by one bin; out-of-range / NaN / past-the-end samples go to the lane's trash slot
            // (packed entries: what count_le_tile returns is the bin, >= nb for dropped samples; slot_base sits one bin back)
            const bool ok1 = scan_gives_bin(SCAN) ? (cnt[0][u][v] < (uint32_t)p.dim[0].nb)
                                                : (Dom<CMP>::in_range((CT)xv[0][u][v], p.dim[0]) &
                                                   (kFloatSamples || !((past_end >> (u * VEC + v)) & 1u)));
            // (bins: the base of bin 0 itself — `+ 1` on the bin is an instruction per sample the variable shift does not fold)
            const uint32_t off = (scan_gives_bin(SCAN) ? cnt[0][u][v] : min(cnt[0][u][v], (uint32_t)p.dim[0].nb)) << slot_shift;
            lds_t* slot = ok1 ? reinterpret_cast<lds_t*>((scan_gives_bin(SCAN) ? slot_base0 : slot_base) + off) : trash_slot;
            if (kWeighted) {
              unsafeAtomicAdd(reinterpret_cast<double*>(slot), (double)wv[u][v]);
              if constexpr (W2) unsafeAtomicAdd(reinterpret_cast<double*>(slot) + hist_elems, (double)wv2[u][v]);
            } else {
              atomicAdd(reinterpret_cast<uint32_t*>(slot), 1u);
            }
            continue;
          }
          // a sample counts when EVERY dimension has it in range; `flat` of a dropped sample is garbage and never used
          // (the range predicates are lane masks: AND-ing them is scalar work, and no bin is ever patched to -1)
          bool ok = kFloatSamples || !((past_end >> (u * VEC + v)) & 1u);
          uint32_t flat = 0;
#pragma unroll
          for (int d = 0; d < D; ++d) {
            uint32_t b;
            if constexpr (scan_gives_bin(SCAN)) {
              b = cnt[d][u][v];  // the bin itself, >= nb (unsigned) when dropped
              ok &= b < (uint32_t)p.dim[d].nb;
            } else {
              b = min(cnt[d][u][v], (uint32_t)p.dim[d].nb) - 1u;  // x == e_last counts E edges -> last bin
              ok &= Dom<CMP>::in_range((CT)xv[d][u][v], p.dim[d]);
            }
            // the last dimension has stride 1; n_bins < 2^24 in every LDS mode and < 2^31 always
            if (d == 0) flat = b;
            else if (HIST != kHistGlobal) flat = __umul24(flat, (uint32_t)p.dim[d].nb) + b;  // full-rate mad_u24
            else flat = flat * (uint32_t)p.dim[d].nb + b;
          }
          if constexpr (SLICED) {  // keep only this launch's bins; from here on `flat` is relative to the slice
            flat -= (uint32_t)p.slice_lo;
            ok &= flat < hb;
          }
          if (HIST == kHistPacked) {
            okv[u][v] = ok;
            flatv[u][v] = flat;
            oldv[u][v] = packed_add(ok, flat);
          } else {
            scatter(ok, flat, kWeighted ? (double)wv[u][v] : 0.0, W2 ? (double)wv2[u][v] : 0.0);
          }
        }
      if (HIST == kHistPacked) {  // all returning adds are in flight before the first check
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) packed_fix(okv[u][v], flatv[u][v], oldv[u][v]);
      }
    }
  }

  if (HIST == kHistPacked) {
    __syncthreads();
    const uint32_t n = (hb + 1u) >> 1;
    for (uint32_t i = tid; i < n; i += blockDim.x) {
      const uint32_t word = packed[i];
      const uint32_t lo = word & 0xffffu, hi = word >> 16;
      if (lo) atomicAdd(reinterpret_cast<unsigned long long*>(out) + 2 * (int64_t)i, (unsigned long long)lo);
      if (hi && 2 * i + 1u < hb)
        atomicAdd(reinterpret_cast<unsigned long long*>(out) + 2 * (int64_t)i + 1, (unsigned long long)hi);
    }
  }

  if (LDS_HIST) {
    __syncthreads();
    const uint32_t copies = 1u << p.copies_log2;
    for (uint32_t b = tid; b < hb; b += blockDim.x) {
      typename std::conditional<kWeighted, double, unsigned long long>::type sum = 0;
      for (uint32_t c = 0; c < copies; ++c) sum += hist[((b + pad_bins) << p.copies_log2) + ((c + tid) & cmask)];
      if (p.direct_store) out[b] = (out_t)sum;  // the only workgroup of this row: plain store, zeros included
      else if (sum != 0) A::out_add(out, (int64_t)b, sum);
      if constexpr (W2) {
        double sum2 = 0;
        for (uint32_t c = 0; c < copies; ++c) sum2 += hist2[(b << p.copies_log2) + ((c + tid) & cmask)];
        if (sum2 != 0) A::out_add(out2, (int64_t)b, sum2);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GENERIC family: any dtype per input, any element strides (broadcast rows/cols), 1..8 inputs,
// float64 or int64 compare domain, tables in LDS or read through L2.  Scalar coalesced loads.
// ---------------------------------------------------------------------------------------------
// TLDS: the tables are staged in LDS (nearly always) — a template parameter so that the table reads compile to
// ds_read instead of flat loads through a pointer that could be either (2 x 10^8 f64 samples: 0.63 -> see DESIGN)
template <int CMP, bool WEIGHTED, bool LDS_HIST, bool TLDS>
__global__ void __launch_bounds__(1024) hist_generic(const Params p) {
  using WT = typename std::conditional<WEIGHTED, double, NoWeight>::type;
  using A = Acc<WT>;
  using lds_t = typename A::lds_t;
  using out_t = typename A::out_t;
  using CT = typename Dom<CMP>::T;

  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x / p.segs;
  const int seg = blockIdx.x % p.segs;

  const uint64_t* tab;
  size_t hist_off = 0;
  if constexpr (TLDS) {
    tab = stage_tables(p);
    hist_off = (size_t)p.table_words * 8;
  } else {
    tab = p.tables;  // the blob in global memory (tables beyond the LDS capacity)
  }
  lds_t* hist = reinterpret_cast<lds_t*>(xhist_smem + hist_off);
  const uint32_t cmask = (1u << p.copies_log2) - 1u;
  const uint32_t mycopy = (uint32_t)tid & cmask;
  const uint32_t trash = ((uint32_t)p.n_bins << p.copies_log2) + ((uint32_t)tid & 31u);  // 32 trash slots, see hist_fast
  if (LDS_HIST) {
    const uint32_t n = ((uint32_t)p.n_bins << p.copies_log2) + 32u;
    for (uint32_t i = tid; i < n; i += blockDim.x) hist[i] = (lds_t)0;
  }
  __syncthreads();

  out_t* out = reinterpret_cast<out_t*>(p.out) + row * p.n_bins;
  const int nd = p.n_dims;
  int64_t roff[kMaxDims];
#pragma unroll
  for (int d = 0; d < kMaxDims; ++d) roff[d] = d < nd ? row_offset(p.row0 + row, p.s_rs[d], p.s_ir[d], p.s_os[d]) : 0;
  const int64_t woff = WEIGHTED ? row_offset(p.row0 + row, p.w_rs, p.w_ir, p.w_os) : 0;

  auto scatter = [&](bool ok, int64_t flat, double w) {
    if (LDS_HIST) {
      const uint32_t idx = ok ? (((uint32_t)flat << p.copies_log2) + mycopy) : trash;
      A::lds_add(hist, idx, w);
    } else if (ok) {
      if (WEIGHTED) A::out_add(out, flat, w);
      else A::out_add(out, flat, 1);
    }
  };
  const int64_t stride = (int64_t)p.segs * blockDim.x;
  int64_t i = (int64_t)seg * blockDim.x + tid;
  if (nd <= 2) {
    // one or two inputs (nearly every call): 4 samples per lane and step as one batch, so the
    // dependent table reads of a sample overlap with those of the other three
    constexpr int B = 4;
    int max_steps = max(p.dim[0].steps, nd > 1 ? p.dim[1].steps : 1);
    for (; i + (B - 1) * stride < p.n_cols; i += B * stride) {
      CT x[B][2];
      double w[B];
      DigState st[B][2];
#pragma unroll
      for (int k = 0; k < B; ++k) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
          if (d < nd) x[k][d] = load_dom<CMP>(p.s_ptr[d], p.s_dt[d], roff[d] + (i + k * stride) * p.s_cs[d], p.dim[d]);
        w[k] = WEIGHTED ? load_as<double>(p.w_ptr, p.w_dt, woff + (i + k * stride) * p.w_cs) : 0.0;
      }
#pragma unroll
      for (int k = 0; k < B; ++k)
#pragma unroll
        for (int d = 0; d < 2; ++d)
          if (d < nd) st[k][d] = digitize_begin<CMP>(x[k][d], p.dim[d], tab);
#pragma unroll 1
      for (int r = 1; r < max_steps; ++r) {
#pragma unroll
        for (int k = 0; k < B; ++k)
#pragma unroll
          for (int d = 0; d < 2; ++d)
            if (d < nd) upper_bound_step<CMP>(x[k][d], p.dim[d], tab, st[k][d]);
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        bool ok = true;
        int64_t flat = 0;
#pragma unroll
        for (int d = 0; d < 2; ++d)
          if (d < nd) {
            const int b = digitize_end(p.dim[d], st[k][d]);
            ok &= (b >= 0);
            flat += (int64_t)b * p.dim[d].out_stride;
          }
        scatter(ok, flat, w[k]);
      }
    }
  }
  for (; i < p.n_cols; i += stride) {
    bool ok = true;
    int64_t flat = 0;
#pragma unroll
    for (int d = 0; d < kMaxDims; ++d) {
      if (d < nd) {
        const CT x = load_dom<CMP>(p.s_ptr[d], p.s_dt[d], roff[d] + i * p.s_cs[d], p.dim[d]);
        const int b = digitize<CMP>(x, p.dim[d], tab);
        ok &= (b >= 0);
        flat += (int64_t)b * p.dim[d].out_stride;
      }
    }
    double w = 0.0;
    if (WEIGHTED) w = load_as<double>(p.w_ptr, p.w_dt, woff + i * p.w_cs);
    scatter(ok, flat, w);
  }

  if (LDS_HIST) {
    __syncthreads();
    const uint32_t copies = 1u << p.copies_log2;
    for (uint32_t b = tid; b < (uint32_t)p.n_bins; b += blockDim.x) {
      typename std::conditional<WEIGHTED, double, unsigned long long>::type sum = 0;
      for (uint32_t c = 0; c < copies; ++c) sum += hist[(b << p.copies_log2) + ((c + tid) & cmask)];
      if (sum != 0) A::out_add(out, (int64_t)b, sum);
    }
  }
}

// Output zeroing as a kernel of this library rather than hipMemsetAsync: a memset node captured
// into a hipGraph was observed (ROCm 7.0 runtime under torch) to replay with a garbage fill
// value; a kernel node replays exactly.
static __global__ void __launch_bounds__(256) zero_words(unsigned long long* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0ull;
}

// ---------------------------------------------------------------------------------------------
// bucket-table builder: one workgroup per dimension applies bucket_of to that dimension's edges
// (scratch[j] = bucket of edge j) and turns the sorted bucket ids into (start | cnt << 16).
// ---------------------------------------------------------------------------------------------
template <int CMP, bool LUT16>
__global__ void __launch_bounds__(256) build_tables(const DimTable t, uint64_t* blob, int32_t* scratch) {
  using T = typename Dom<CMP>::T;
  const T* edges = reinterpret_cast<const T*>(blob + t.edge_off);
  uint32_t* lut = reinterpret_cast<uint32_t*>(blob) + t.lut_off;
  uint16_t* lut16 = reinterpret_cast<uint16_t*>(blob) + t.lut_off;
  for (int j = threadIdx.x; j < t.n_edges; j += blockDim.x) scratch[j] = bucket_of<CMP>(edges[j], t);
  __syncthreads();
  for (int b = threadIdx.x; b < t.lut_k; b += blockDim.x) {
    // lower_bound of b and of b+1 in the non-decreasing scratch[0..E)
    int lo = 0, hi = t.n_edges;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (scratch[m] < b) lo = m + 1; else hi = m; }
    const int start = lo;
    hi = t.n_edges;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (scratch[m] < b + 1) lo = m + 1; else hi = m; }
    if (LUT16) lut16[b] = (uint16_t)start;
    else lut[b] = (uint32_t)start | ((uint32_t)(lo - start) << 16);
  }
}

// packed-entry tables (count_le_pack): one workgroup per dimension.  thr_j = (float)e_j is computed HERE, with the
// conversion the kernels apply to the samples; entry b = { thr of the first three edges of bucket b (NaN beyond the
// bucket's own), start }.  The host reads the table back and offers it only if no bucket holds more than three edges.
// thr_given (float32-sample tables): the thresholds come from the host (smallest float32 >= e_j; > e_last for the last) and only
// the bucket map is applied here.
static __global__ void __launch_bounds__(256) build_pack_tables(const DimTable t, uint64_t* blob, int32_t* scratch, const float* thr_given) {
  const double* edges = reinterpret_cast<const double*>(blob + t.edge_off);
  pack_entry_t* ent = reinterpret_cast<pack_entry_t*>(blob) + t.lut_off;
  auto thr_of = [&](int j) { return thr_given ? thr_given[j] : (float)edges[j]; };
  for (int j = threadIdx.x; j < t.n_edges; j += blockDim.x)
    scratch[j] = t.map_kind ? bucket_of_key(thr_of(j), t) : bucket_of<2>(thr_of(j), t);
  __syncthreads();
  for (int b = threadIdx.x; b < t.lut_k; b += blockDim.x) {
    int lo = 0, hi = t.n_edges;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (scratch[m] < b) lo = m + 1; else hi = m; }
    const int start = lo;
    hi = t.n_edges;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (scratch[m] < b + 1) lo = m + 1; else hi = m; }
    const int cnt = lo - start;
    pack_entry_t e;
#pragma unroll
    for (int k = 0; k < 3; ++k) e[k] = k < cnt ? __float_as_uint(thr_of(start + k)) : 0x7fc00000u;
    e[3] = (uint32_t)start - 1u;  // (count_le_pack's sum is then the bin itself)
    ent[b] = e;
  }
}

// ---------------------------------------------------------------------------------------------
// min / max with numpy's NaN propagation (feeds np.histogram_bin_edges, core.py:383-388)
// partial[3*b + {0,1,2}] = {min, max, saw_nan} of workgroup b
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) minmax_kernel(const void* ptr, int32_t dt, int64_t rs, int64_t cs, int64_t ir, int64_t os,
                                                       int64_t n_rows, int64_t n_cols, double* partial) {
  double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
  int nan = 0;
  const int64_t total = n_rows * n_cols;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = k / n_cols, c = k - r * n_cols;
    const double x = load_as<double>(ptr, dt, row_offset(r, rs, ir, os) + c * cs);
    nan |= (x != x);
    mn = fmin(mn, x);
    mx = fmax(mx, x);
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = fmin(mn, __shfl_down(mn, off, 64));
    mx = fmax(mx, __shfl_down(mx, off, 64));
    nan |= __shfl_down(nan, off, 64);
  }
  __shared__ double s_mn[4], s_mx[4];
  __shared__ int s_nan[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_mn[wave] = mn; s_mx[wave] = mx; s_nan[wave] = nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { mn = fmin(mn, s_mn[w]); mx = fmax(mx, s_mx[w]); nan |= s_nan[w]; }
    partial[3 * blockIdx.x + 0] = mn;
    partial[3 * blockIdx.x + 1] = mx;
    partial[3 * blockIdx.x + 2] = (double)nan;
  }
}

// ---------------------------------------------------------------------------------------------
// count / min / max / sum — or the sum of squared deviations from a given mean — of the elements inside [lo, hi] (or of
// all elements): what numpy's bin-width estimators need of the data (np.histogram_bin_edges with bins = "sqrt", "sturges",
// "rice", "scott": core.py:383-388 hands the whole array to numpy; here the array stays where it is).
//   pass 0: partial[5*b + {0..4}] = {count, min, max, sum, saw NaN}      pass 1: partial[5*b] = sum (x - mean)^2
// Elements outside the range — and NaN when a range is given, as numpy's `keep` mask drops them — do not count.
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) moments_kernel(const void* ptr, int32_t dt, int64_t rs, int64_t cs, int64_t ir, int64_t os,
                                                        int64_t n_rows, int64_t n_cols, int use_range, double lo, double hi, int pass, double mean,
                                                        double* partial) {
  double cnt = 0.0, mn = __builtin_huge_val(), mx = -__builtin_huge_val(), sum = 0.0;
  int nan = 0;
  const int64_t total = n_rows * n_cols;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = k / n_cols, c = k - r * n_cols;
    const double x = load_as<double>(ptr, dt, row_offset(r, rs, ir, os) + c * cs);
    const bool keep = use_range ? ((x >= lo) & (x <= hi)) : true;
    if (!keep) continue;
    if (pass == 0) {
      nan |= (x != x);
      cnt += 1.0;
      mn = fmin(mn, x);
      mx = fmax(mx, x);
      sum += x;
    } else {
      const double dlt = x - mean;
      sum += dlt * dlt;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_down(cnt, off, 64);
    sum += __shfl_down(sum, off, 64);
    mn = fmin(mn, __shfl_down(mn, off, 64));
    mx = fmax(mx, __shfl_down(mx, off, 64));
    nan |= __shfl_down(nan, off, 64);
  }
  __shared__ double s_v[4][4];
  __shared__ int s_nan[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_v[wave][0] = cnt; s_v[wave][1] = mn; s_v[wave][2] = mx; s_v[wave][3] = sum; s_nan[wave] = nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      cnt += s_v[w][0]; mn = fmin(mn, s_v[w][1]); mx = fmax(mx, s_v[w][2]); sum += s_v[w][3]; nan |= s_nan[w];
    }
    double* o = partial + 5 * (int64_t)blockIdx.x;
    if (pass == 0) { o[0] = cnt; o[1] = mn; o[2] = mx; o[3] = sum; o[4] = (double)nan; }
    else o[0] = sum;
  }
}

// contiguous float64 / float32 data: 16-byte non-temporal loads, 4 in flight per lane, comparisons
// in the data's own type (exact); same partial[] layout as minmax_kernel
template <typename T>
__global__ void __launch_bounds__(256) minmax_flat(const T* __restrict__ x, int64_t n, double* partial) {
  constexpr int VEC = 16 / (int)sizeof(T), UNROLL = 4;
  using vec_t = typename VecOf<T, VEC>::type;
  T mn = (T)__builtin_huge_val(), mx = -(T)__builtin_huge_val();
  int nan = 0;
  const int64_t tile = (int64_t)blockDim.x * VEC * UNROLL;
  const int64_t n_full = n / tile;
  for (int64_t t = blockIdx.x; t < n_full; t += gridDim.x) {
    vec_t v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(x + t * tile + ((int64_t)u * blockDim.x + threadIdx.x) * VEC));
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const T e = v[u][k];
        nan |= (e != e);
        mn = e < mn ? e : mn;  // (NaN compares false: skipped here, reported through `nan`)
        mx = e > mx ? e : mx;
      }
  }
  for (int64_t i = n_full * tile + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T e = x[i];
    nan |= (e != e);
    mn = e < mn ? e : mn;
    mx = e > mx ? e : mx;
  }
  double dmn = (double)mn, dmx = (double)mx;
  for (int off = 32; off > 0; off >>= 1) {
    dmn = fmin(dmn, __shfl_down(dmn, off, 64));
    dmx = fmax(dmx, __shfl_down(dmx, off, 64));
    nan |= __shfl_down(nan, off, 64);
  }
  __shared__ double s_mn[4], s_mx[4];
  __shared__ int s_nan[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_mn[wave] = dmn; s_mx[wave] = dmx; s_nan[wave] = nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { dmn = fmin(dmn, s_mn[w]); dmx = fmax(dmx, s_mx[w]); nan |= s_nan[w]; }
    partial[3 * blockIdx.x + 0] = dmn;
    partial[3 * blockIdx.x + 1] = dmx;
    partial[3 * blockIdx.x + 2] = (double)nan;
  }
}

}  // namespace xhist
