// xhist_route.hip.h — the partitioned mode (histograms beyond LDS, BASELINE C5) in ONE routing pass.
//
// xhist_partition.hip.h routes samples to their bin partition in three passes — count (re-reads the
// samples only to size the record slots, spills a 4-byte flat index), prefix, scatter — and moves
// 52.5 B per C5 sample for 24 algorithmic.  Here the slots are not counted in advance: record space is
// handed out on demand in CHUNKS, so one kernel reads the samples, digitizes them, sorts each
// tile (4 samples per lane: 2048 or 4096) by partition in LDS and writes the records; a second one adds them up.
//
//   part_route              reads x (, y, z), w: 24 B/sample (C5); writes (code u16, weight) records:
//                           10 B/sample (8 when packed, see below), into chunks of 2^chunk_log2 records that belong to ONE
//                           (workgroup, partition) pair — no other workgroup writes there, so record
//                           addresses need no atomics; a lane that owns a partition takes a new chunk
//                           id from an LDS-resident stock (refilled from a global counter a tile ahead, so
//                           nobody waits for it) and files the id in the partition's chunk list; when the
//                           pool has no chunk left, records go straight to the output (kRouteDirect)
//   part_accumulate_chunks  every workgroup takes an equal share of the concatenated chunk lists
//                           (balanced for any distribution of the samples), streams the chunks
//                           (contiguous, 16-byte aligned), ds_adds into a 2^shift-bin LDS histogram
//                           and flushes it at partition boundaries: 10 (8) B/sample read
//
// HBM traffic 24 + 10 + 10 = 44 B per C5 sample, 40 with packed records.  What bounds it (round 3): the
// chip moves this read/write mix at 5.3-5.6 TB/s in the pass's access shape (32-byte lane loads, 64 scattered write
// streams per workgroup; tools/ubench/mixbw.hip, profiles/r03_a_*, r03_d_*), and a bare kernel with ALL of the pass's
// phases — barriers, LDS round trip, late prefetch, returning LDS atomics — still does (profiles/r03_ph_*): 3.05-3.25 ms
// for a 5*10^8-sample shard, against 3.26-3.42 measured; reading the records back takes 0.57 ms at best, 0.72-0.76 measured.
//
// Packed records.  float64 weights of ONE sign travel as 8-byte records — the weight's upper 48 bits with the bin
// code in the 16 that go (rounded to nearest: 2^-37 relative per weight, and no cancellation to amplify it): 24 + 8 + 8 = 40 B per sample,
// one record stream instead of two.  Weights of both signs keep full float64 records; which case a call is in is found
// out by the routing pass itself and acted on by the GPU (execute_partitioned_fused in xhist_exec_device.hip.h).
// Measured on C5 shards: the adding-up pass 0.83 -> 0.71 ms, the routing pass 3.5-3.66 ms where the two-stream form
// takes 3.57 in one process and 4.15 in the next on the same box (its five streams are sensitive to where the driver
// places the buffers; four streams are not: profiles/r02_u_records48.jsonl).
//
// The LDS sort, the aligned 16-byte record groups and the records carried from tile to tile are those
// of part_scatter (xhist_partition.hip.h), which documents them.
#pragma once

#include "xhist_partition.hip.h"

namespace xhist {

constexpr int kRouteGrp = 8;                  // records per aligned group: 8 codes = one 16-byte store
constexpr uint32_t kChunkFillMask = 0xfffffu;  // cmeta[id] = partition << 20 | records in the chunk
// delta2[q] of a block whose records beyond the current chunk found NO chunk left in the pool: they are added to the
// output one by one with memory-side atomics (slow, exact).  The pool is sized for the worst case, so this is the answer
// to a sizing bug or a shrunken pool ("route_pool_pct"), not a mode — but it keeps the library's "never aborts" promise
// where a trap would poison the HIP context of a long-lived worker (VERDICT r2 "weak" #5).
constexpr uint64_t kRouteDirect = ~(uint64_t)0;
// Workgroup size and samples per lane and tile are template parameters (1024 / 512 threads; 4 / 8 samples): the host picks
// them per dtype combination (route_geom_for in xhist_exec_device.hip.h holds the rule and the measurements).  In short:
// the pass's mixed read/write traffic wants long bursts per workgroup, so 1024 threads x 8 samples wherever the tile's
// samples and weights fit the registers next to the sort's state (counts; float32 weights; float32 samples), 1024 x 4
// for float64 samples with float64 weights (8 per lane spill there: what spills is reloaded inside the loop, and every
// reload waits for the prefetch in flight).  Round 2's 8-per-lane attempt (float32 pairs + weights 3.39 ms against 3.00)
// predates the split weight loads and the arithmetic digitize; with them the same shape runs 2.64 against 2.92.
// Two 512-thread workgroups per CU (VERDICT r2 "next" #1b) won 2-3 % while weights were loaded with the samples
// (profiles/r03_c5_blocks.txt: 3.43-3.48 -> 3.35-3.37 ms; three of 256 threads 3.51-3.55) and lose 2-4 % since.
constexpr int kRouteBlock = 1024;  // the largest workgroup (sizes the host-side worst cases)
__host__ __device__ constexpr int route_tile(int block, int spl = 4) { return spl * block; }  // spl: samples per lane and tile (4 or 8)
// chunks one workgroup can file in its LDS list (beyond: filed one by one, slowly)
// (= the workgroup size.  1024 for every size was tried — shorter lists force bigger chunks, and every workgroup ends with one
// to two batches of unused chunk ids, so 12 rows x 10^8 samples with 8192-record chunks size their pool at 19 GB for 7 GB of
// records — and lost: 4 KB more LDS per workgroup cost the second workgroup per CU at 128 partitions, 3.95 -> 4.8 ms for
// 8 x 6*10^7 float64 pairs into 512 x 512 bins, and 32 x 3*10^7 float32 pairs went 5.9 -> 7.5 ms with 1024-record chunks.)
__host__ __device__ constexpr int route_list_cap(int block) { return block; }
__host__ __device__ constexpr int route_ctl(int block) { return 18496 + 2 * 4 * route_list_cap(block); }  // control arrays of part_route (see the kernel)
constexpr int kAccBatch = 1024;                // chunks a workgroup of part_accumulate_chunks stages at a time
// most chunks one partition can need for ONE tile (every record of the tile, plus padding, minus what its chunk still holds)
__host__ __device__ constexpr int route_max_need(int chunk_log2, int tile) { return ((tile + 2 * kRouteGrp) >> chunk_log2) + 1; }
// chunk ids a workgroup takes from the pool at a time: enough for every partition to switch in one tile, and at least
// eight times the largest single request, so that the ids a range cannot serve any more (fewer than the largest request,
// dropped when the next range takes over) stay below an eighth of what it handed out
__host__ __device__ constexpr int route_batch(int P, int chunk_log2, int tile) {
  return 2 * P > 8 * route_max_need(chunk_log2, tile) ? 2 * P : 8 * route_max_need(chunk_log2, tile);
}

// float64 weights whose records carry 48 bits (sign, exponent, 36 mantissa bits) next to the 16-bit code: ONE 8-byte
// record per sample instead of 2 + 8 bytes in two streams (see "packed records" below)
struct Packed48 {};

// weight ROUNDED (to nearest, ties to even) to its upper 48 bits — sign, exponent, 36 mantissa bits: at most 2^-37 relative
// per weight and no bias (truncation, round 2's form, was 2^-36 and always towards zero: ADVICE r2) — with the bin code in
// the low 16 bits.  Non-finite weights are not rounded (a carry would walk through a NaN's payload into the sign); a NaN
// whose payload sits only in the bits that go would turn into an infinity and gets its quiet bit set first; a finite weight
// that rounding would carry into the infinity exponent (the top 2^-37 of the float64 range) is truncated instead.
__device__ __forceinline__ double pack48(double w, uint32_t code) {
  const uint64_t b = (uint64_t)__double_as_longlong(w);
  uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
  const bool finite = (hi & 0x7ff00000u) != 0x7ff00000u;
  hi = (w != w) ? (hi | 0x00080000u) : hi;
  const uint32_t lo_r = lo + 0x7fffu + ((lo >> 16) & 1u);
  const uint32_t hi_r = hi + (lo_r < lo ? 1u : 0u);
  const bool take = finite & ((hi_r & 0x7ff00000u) != 0x7ff00000u);
  lo = take ? lo_r : lo;
  hi = take ? hi_r : hi;
  lo = (lo & 0xffff0000u) | code;
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

struct RouteArgs {
  uint32_t* pool;       // [0]: next unused chunk id
  uint32_t* pcount;     // [P]: chunk ids filed per partition
  uint32_t* plist;      // [P][list_cap]: the chunk ids of each partition
  uint32_t* cmeta;      // [pool_chunks]: partition << 20 | fill, written once when a chunk is closed
  uint16_t* codes;      // [pool_chunks << chunk_log2]: bin index inside the partition
  void* wrec;           // [pool_chunks << chunk_log2]: weight (float64, or float32 for float32 weights)
  uint32_t list_cap;    // = chunks in the pool
  int32_t chunk_log2;
  // packed records: the routing pass ORs into *flags which signs the weights of the kept samples had (1 negative,
  // 2 positive); kernels with gate_mode != 0 return at once unless (*gate says both signs) == (gate_mode == 2);
  // a gate_mode-2 routing pass that does run leaves 1 in *hint (host memory: the plan's memory of mixed signs)
  uint32_t* flags;
  const uint32_t* gate;
  uint32_t* hint;
  int32_t gate_mode;
  uint32_t* dry;        // host memory (may be NULL): set to 1 when the chunk pool ran dry and records went straight to the output
  // the exchange mode (xhist_exchange.hip.h) took this call when *xgate != 0: the classic PACKED kernels queued behind it return
  // at once (the exact ones do not look here: they redo the call whenever the sign word says so)
  const uint32_t* xgate;
};

__device__ __forceinline__ bool route_gate_closed(const RouteArgs& ra) {
  if (ra.xgate && __builtin_nontemporal_load(ra.xgate) != 0u) return true;
  if (ra.gate_mode == 0) return false;
  const bool mixed = (__builtin_nontemporal_load(ra.gate) & 3u) == 3u;
  return (ra.gate_mode == 2) != mixed;
}

__host__ __device__ constexpr int part_route_slots(int P, int tile) { return tile + 2 * (kRouteGrp - 1) * P + kRouteGrp; }
__host__ __device__ constexpr size_t part_route_lds(size_t table_bytes, int P, bool weighted, int tile, int block) {
  return ((table_bytes + 15) & ~(size_t)15) + (size_t)route_ctl(block) + (size_t)P * kRouteGrp * (weighted ? 12 : 4) +
         (size_t)part_route_slots(P, tile) * (weighted ? 12 : 4) + 64;
}

// a 4-vector read `sh` elements before where it belongs (loads near the end of the array are pulled
// back in bounds): element v of the result is element v + sh of what was read, `fill` past the end
template <typename V, typename S>
__device__ __forceinline__ V pulled_back(V q, int sh, S fill) {
  V r;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    S x = fill;
#pragma unroll
    for (int k = v; k < 4; ++k) x = (v + sh == k) ? q[k] : x;
    r[v] = x;
  }
  return r;
}

// Inclusive add-scan over the 64 lanes of a wavefront with DPP moves (row shifts inside the rows of 16, then the two row
// broadcasts of GCN / CDNA): six dependent VALU operations.  The ds_bpermute form (__shfl_up x 6) costs an LDS round trip per
// step — ~700 cycles of pure latency in a phase of the routing pass in which ONE wavefront works and fifteen wait
// (profiles/r04_i_route_phase_cycles.txt: the block-layout scan took 2700 of a tile's 14700 cycles).
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t x) {
  // (old = 0, bound_ctrl off: lanes without a source — and rows masked out — contribute 0)
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8: inclusive inside every row of 16
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return x;
}

// `need` consecutive chunk ids from the workgroup's two LDS-resident id ranges a = {next0, end0, next1, end1};
// 0xFFFFFFFF when both are used up (the caller then goes to the global pool itself)
__device__ __forceinline__ uint32_t route_take_ids(uint32_t* a, uint32_t need) {
  uint32_t s = atomicAdd(a + 0, need);
  if (s + need <= a[1]) return s;
  s = atomicAdd(a + 2, need);
  if (s + need <= a[3]) return s;
  return 0xffffffffu;
}

// Nothing between the issue of a tile's loads and their first use may wait on the vector-memory counter: it
// counts loads, stores and returning atomics in order, so a wait for ANY of them — a register reloaded from
// scratch, a chunk id fetched from the global pool — is a wait for the whole prefetch, and the pass degenerates
// into load, wait, sort, store, one after the other (4.2 ms instead of 3.4 for a C5 shard).  Hence: everything
// the partition owners keep from tile to tile (record cursors, chunk lists) lives in LDS, and chunk ids come
// from an LDS-resident stock that one lane refills from the global pool a tile before it runs out.
template <typename ST, typename WT, int D, int SCAN, bool MULTI = false, int BLOCK = kRouteBlock, int SPL = 4>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) part_route(const Params p, const RouteArgs ra) {
  constexpr bool kWeighted = !__is_same(WT, NoWeight);
  constexpr bool PACK = __is_same(WT, Packed48);
  constexpr int CMP = (__is_same(ST, float) && SCAN != kScanArith) ? 2 : 0;
  using CT = typename Dom<CMP>::T;
  using wscalar = typename std::conditional<kWeighted, typename std::conditional<PACK, double, WT>::type, float>::type;
  using RT = typename std::conditional<__is_same(WT, float), float, double>::type;  // record weights keep the caller's precision
  constexpr int RV = 16 / (int)sizeof(RT);
  typedef RT rvec __attribute__((ext_vector_type(RV)));
  constexpr int kRouteTile = route_tile(BLOCK, SPL), kRouteListCap = route_list_cap(BLOCK), kRouteCtl = route_ctl(BLOCK);
  constexpr int GRP = kRouteGrp, U = kRouteTile / (BLOCK * 4);  // 4-sample vectors per lane and tile
  constexpr uint32_t kGm = GRP - 1;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef ST s4 __attribute__((ext_vector_type(4), aligned(sizeof(ST))));
  typedef wscalar w4 __attribute__((ext_vector_type(4), aligned(sizeof(wscalar))));

  if (route_gate_closed(ra)) return;
  const int tid = threadIdx.x;
  const int P = p.n_parts, shift = p.part_shift, lg = ra.chunk_log2;
  const uint32_t CH = 1u << lg;
  const int64_t n = p.n_cols;
  if (ra.gate_mode == 2 && blockIdx.x == 0 && tid == 0 && ra.hint) *ra.hint = 1u;
  double w_lo = 0.0, w_hi = 0.0;  // smallest / largest weight this lane read (packed records: which signs occur)
  const uint64_t* tab = stage_tables(p);
  unsigned char* ctl = xhist_smem + (((size_t)p.table_words * 8 + 15) & ~(size_t)15);
  uint32_t* cnt2 = reinterpret_cast<uint32_t*>(ctl);            // [2][256] rank counters, alternating per tile
  uint32_t* cin2 = cnt2 + 512;                                   // [2][256] carried records per partition
  uint64_t* delta = reinterpret_cast<uint64_t*>(ctl + 4096);     // [P] record slot of LDS slot 0 of the block (current chunk)
  uint64_t* delta2 = delta + 256;                                // [P] the same for the part of the block that went to a new chunk
  uint32_t* first = reinterpret_cast<uint32_t*>(ctl + 8192);     // [P] LDS slot of the first NEW record
  uint32_t* endw = first + 256;                                  // [P] end of the whole groups of the block
  uint32_t* enda = endw + 256;                                   // [P] end of the records of the block
  uint32_t* split = enda + 256;                                  // [P] LDS slot from which the block continues in the new chunk
  uint64_t* o_cur = reinterpret_cast<uint64_t*>(ctl + 12288);    // [P] next free record slot of the partition's chunk
  uint64_t* o_cend = o_cur + 256;                                // [P] end of that chunk (0 = no chunk yet)
  uint32_t* head = reinterpret_cast<uint32_t*>(ctl + 16384);     // [P] newest entry of the partition's chunk list
  uint32_t* ccnt = head + 256;                                   // [P] chunks this workgroup filed for the partition
  uint32_t* misc = ccnt + 256;                                   // [0] LDS slots in use this tile, [1] list entries, [4..7] id stock
  uint32_t* cl_id = misc + 16;                                   // [kRouteListCap] chunk ids filed by this workgroup ...
  uint32_t* cl_prev = cl_id + kRouteListCap;                     // ... chained per partition
  uint32_t* stock = misc + 4;
  unsigned char* dyn = ctl + kRouteCtl;
  uint32_t* carry_key = reinterpret_cast<uint32_t*>(dyn);       // [P][GRP]
  dyn += (size_t)P * GRP * 4;
  RT* carry_w = reinterpret_cast<RT*>(dyn);                      // [P][GRP]
  if (kWeighted) dyn += (size_t)P * GRP * 8;
  const int S = part_route_slots(P, kRouteTile);
  RT* sw = reinterpret_cast<RT*>(dyn);                           // [S] weights of the sorted tile
  if (kWeighted) dyn += (size_t)S * 8;
  uint32_t* skey = reinterpret_cast<uint32_t*>(dyn);             // [S] keys: part << 16 | code
  RT* __restrict__ wrec = static_cast<RT*>(ra.wrec);
  uint16_t* __restrict__ codes = ra.codes;
  const wscalar* wp = reinterpret_cast<const wscalar*>(p.w_ptr);
  const ST* sp[D];
#pragma unroll
  for (int d = 0; d < D; ++d) sp[d] = reinterpret_cast<const ST*>(p.s_ptr[d]);
  int max_steps = 1;
#pragma unroll
  for (int d = 0; d < D; ++d) max_steps = max(max_steps, p.dim[d].steps);
  const uint32_t code_mask = (1u << shift) - 1u;
  const uint32_t batch = (uint32_t)route_batch(P, lg, kRouteTile), max_need = (uint32_t)route_max_need(lg, kRouteTile);
  for (int i = tid; i < 256; i += blockDim.x) {
    cnt2[i] = 0u;
    cnt2[256 + i] = 0u;
    cin2[i] = 0u;
    cin2[256 + i] = 0u;
    o_cur[i] = 0;
    o_cend[i] = 0;
    head[i] = 0xffffffffu;
    ccnt[i] = 0u;
  }
  if (tid == 0) {
    const uint32_t b = atomicAdd(ra.pool, 2u * batch);
    stock[0] = b;
    stock[1] = b + batch;
    stock[2] = b + batch;
    stock[3] = b + 2u * batch;
    misc[1] = 0u;
  }
  bool refill_pending = false;  // lane 0: a batch of ids has been asked for, `refill_ids` arrives by the next tile
  uint32_t refill_ids = 0;
  __syncthreads();

  // a chunk becomes partition q's current one: chained into the workgroup's list (or, list full, filed directly)
  auto file_chunk = [&](int q, uint32_t id) {
    const uint32_t e = atomicAdd(misc + 1, 1u);
    if (e < (uint32_t)kRouteListCap) {
      cl_id[e] = id;
      cl_prev[e] = head[q];
      head[q] = e;
      ccnt[q] += 1u;
    } else {
      ra.plist[(size_t)q * ra.list_cap + atomicAdd(ra.pcount + q, 1u)] = id;
    }
  };

  // one record straight into the output (the pool-dry path): key = partition << 16 | code
  // (a packed-record pass does not add anything: it reports "both signs" instead, which makes the exact pass queued
  // behind it redo the whole call — adds of its own would be counted twice then)
  auto direct_add = [&](uint32_t key, RT wv) {
    if constexpr (PACK) return;
    const int part = (int)(key >> 16);
    const int row = part / p.parts_per_row;
    const int64_t bin = ((int64_t)(part - row * p.parts_per_row) << shift) + (key & 0xffffu);
    if (bin >= p.n_bins) return;
    if constexpr (kWeighted) {
      double v = (double)wv;
      unsafeAtomicAdd(reinterpret_cast<double*>(p.out) + (int64_t)row * p.n_bins + bin, v);
    } else {
      atomicAdd(reinterpret_cast<unsigned long long*>(p.out) + (int64_t)row * p.n_bins + bin, 1ull);
    }
  };

  // Several rows in one pass (a few time steps of a big joint histogram): tiles never straddle rows, a tile's partition
  // ids are offset by its row's share of the partitions, and everything downstream sees n_rows * parts_per_row partitions.
  const int64_t tiles_per_row = (n + kRouteTile - 1) / kRouteTile;
  const int64_t n_tiles = tiles_per_row * p.n_rows;
  const int64_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;  // the grid never exceeds the number of tiles
  constexpr bool multi = MULTI;  // (a compile-time split: the one-row pass — BASELINE C5 — pays nothing for the row arithmetic)
  auto tile_row = [&](int64_t k) -> int64_t { return multi ? ((int64_t)blockIdx.x + k * gridDim.x) / tiles_per_row : 0; };
  auto tile_base = [&](int64_t k) {  // first sample of the tile inside ITS row
    const int64_t t = (int64_t)blockIdx.x + k * gridDim.x;
    return (multi ? t - (t / tiles_per_row) * tiles_per_row : t) * kRouteTile;
  };
  // Loads are free of control flow (see part_scatter): a quad that would cross the end of the arrays is
  // read 4 elements back from the end and moved into place when it is used.  Requires n >= 4.
  // Addresses: a uniform 64-bit tile pointer (scalar registers) plus ONE 32-bit byte offset per lane and load —
  // twelve 64-bit lane addresses kept across the loop were a third of the register file.
  auto load_tile = [&](int64_t base, int64_t row, s4 (&x)[D][U], w4 (&w)[U], bool do_x = true, bool do_w = true) {
    const int64_t origin = min(base, n - 4);  // (a last tile of fewer than 4 samples reads the 4 before the end)
    const uint32_t last = (uint32_t)min(n - 4 - origin, (int64_t)kRouteTile);  // first element of the last whole quad, tile-relative
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t e = min((uint32_t)(u * BLOCK + tid) * 4u, last);
      if (do_x)
#pragma unroll
        for (int d = 0; d < D; ++d)
          x[d][u] = __builtin_nontemporal_load(reinterpret_cast<const s4*>(reinterpret_cast<const char*>(sp[d] + row * p.s_rs[d] + origin) + e * (uint32_t)sizeof(ST)));
      if (kWeighted && do_w)
        w[u] = __builtin_nontemporal_load(reinterpret_cast<const w4*>(reinterpret_cast<const char*>(wp + row * p.w_rs + origin) + e * (uint32_t)sizeof(wscalar)));
    }
  };
  // (Issuing the loads of tile k + 1 at the TOP of iteration k, into registers of their own, was tried and is slower — 3.35 ->
  // 3.46 ms for a C5 shard: this chip moves a read-write mix faster with FEWER bytes in flight, HISTORY 4.3.  The late prefetch
  // below has a tile's loads in flight during scan and sort only.)
  // WSPLIT: the weights of a tile are not needed before its records are sorted, so they are loaded at the top of the tile's
  // OWN iteration (and waited for ahead of the sort) instead of with the samples of the tile a whole iteration earlier:
  // the loads of a workgroup come in two smaller bursts, and the registers of the second weight buffer are free
  constexpr bool WSPLIT = kWeighted;
  s4 xv[D][U];
  w4 w[U], wn[U];
  load_tile(tile_base(0), tile_row(0), xv, w, true, !WSPLIT);
  if constexpr (WSPLIT) {  // (tile 0 has arrived before the loop is entered)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kWeighted && !WSPLIT) asm volatile("" : "+v"(w[u]));
#pragma unroll
      for (int d = 0; d < D; ++d) asm volatile("" : "+v"(xv[d][u]));
    }
  }
  int cur_set = 0;
  for (int64_t k = 0; k < my_tiles; ++k, cur_set ^= 1) {
    if constexpr (WSPLIT) load_tile(tile_base(k), tile_row(k), xv, w, false, true);
    const int64_t base = tile_base(k);
    const uint32_t row_off = (uint32_t)tile_row(k) * ((uint32_t)p.parts_per_row << shift);  // this row's first partition, as a flat index
    const bool ragged = base + kRouteTile > n;
    uint32_t* cnt = cnt2 + (cur_set << 8);
    const uint32_t* cin = cin2 + (cur_set << 8);
    if (ragged) {  // positions past the end become NaN samples, which digitize drops
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = base + ((int64_t)u * BLOCK + tid) * 4;
        const int sh = (int)min(i - min(i, n - 4), (int64_t)4);
#pragma unroll
        for (int d = 0; d < D; ++d) xv[d][u] = pulled_back(xv[d][u], sh, (ST)__builtin_nanf(""));
      }
    }
    // ---- digitize the tile: flat bin index per sample (0xFFFFFFFF = dropped) ----------------------
    uint32_t flat[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s4 xs[D][1];
#pragma unroll
      for (int d = 0; d < D; ++d) xs[d][0] = xv[d][u];
      int bins[D][4];
      if constexpr (SCAN == kScanArith) {
        // bins by arithmetic alone (bin_arith_fast) for every sample that is not within delta bins of an edge; a wavefront
        // in which some lane met such a sample (for C5's edges: one sample in 10^12) redoes that lane's samples with the
        // exact compares.  The edge constants stay in scalar registers: left alone, the compiler copies them (5 float64
        // per input) into vector registers ahead of the loop, and these workgroups have none to spare
        bool near_any = false;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          DimTable t = p.dim[d];
          asm volatile("" : "+s"(t.e0_f), "+s"(t.inv_step), "+s"(t.arith_h), "+s"(t.nb));
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            bool near;
            bins[d][v] = bin_arith_fast((double)xs[d][0][v], t, near);
            near_any |= near;
          }
        }
        if (__builtin_amdgcn_ballot_w64(near_any) != 0ull) {
          if (near_any) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
              DimTable t = p.dim[d];
              asm volatile("" : "+s"(t.e0_f), "+s"(t.eL_f), "+s"(t.step), "+s"(t.inv_step), "+s"(t.nb));
#pragma unroll
              for (int v = 0; v < 4; ++v)
                bins[d][v] = bin_from_count<CMP>((CT)xs[d][0][v], t, count_le_arith((double)xs[d][0][v], t));
            }
          }
        }
      } else {
        uint32_t cntle[D][1][4];
        count_le_tile<CMP, SCAN, D, 1, 4>(xs, p, tab, max_steps, cntle);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
          for (int v = 0; v < 4; ++v) bins[d][v] = bin_from_tile_count<CMP, SCAN>((CT)xs[d][0][v], p.dim[d], cntle[d][0][v]);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        bool ok = true;
        uint32_t fl = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const int b = bins[d][v];
          ok &= (b >= 0);
          fl = (d == 0) ? (uint32_t)b : fl * (uint32_t)p.dim[d].nb + (uint32_t)b;
        }
        flat[u][v] = ok ? fl + row_off : 0xffffffffu;
      }
    }
    uint32_t rank[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) rank[u][v] = (flat[u][v] != 0xffffffffu) ? atomicAdd(cnt + (flat[u][v] >> shift), 1u) : 0u;
    // the next tile's loads (the tile after the last one is the last one again: one redundant load per
    // workgroup instead of a branch around loads); waited for just before this tile's stores go out
    {
      const int64_t kn = k + 1 < my_tiles ? k + 1 : k;
      load_tile(tile_base(kn), tile_row(kn), xv, wn, true, !WSPLIT);
    }
    __syncthreads();
    // ---- block layout (as in part_scatter) + record space of every partition's block: LDS only -------
    if (tid < ((P + 63) & ~63)) {
      const int lane = tid & 63;
      const uint32_t c_in = tid < P ? cin[tid] : 0u;
      const uint32_t T = c_in + (tid < P ? cnt[tid] : 0u);
      const uint32_t block = (T + kGm) & ~kGm;
      const uint32_t x = wave_inclusive_scan_u32(block);  // inclusive scan over the wavefront
      uint32_t before = 0;  // blocks of the partitions handled by earlier wavefronts (none for up to 64 partitions: C5)
      if (__builtin_amdgcn_readfirstlane(tid) >= 64) {
        for (int q = lane; q < (tid & ~63); q += 64) before += (cin[q] + cnt[q] + kGm) & ~kGm;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
      }
      const uint32_t B = before + x - block;
      if (tid < P) {
        const uint32_t whole = T & ~kGm;
        first[tid] = B + c_in;
        endw[tid] = B + whole;
        enda[tid] = B + T;
        uint64_t cur = o_cur[tid];
        const uint64_t cend = o_cend[tid];
        delta[tid] = cur - B;
        const uint64_t room = cend - cur;
        if (whole <= room) {
          split[tid] = 0xffffffffu;
          o_cur[tid] = cur + whole;
        } else {  // the current chunk fills up inside this block: the rest goes to new chunk(s)
          const uint32_t n1 = (uint32_t)room, rest = whole - n1;
          const uint32_t need = (rest + CH - 1) >> lg;  // > 1 only when one tile sends more than a chunk to one partition
          if (cend != 0) ra.cmeta[(uint32_t)((cend - 1) >> lg)] = ((uint32_t)tid << 20) | CH;
          uint32_t id0 = route_take_ids(stock, need);
          if (id0 == 0xffffffffu) id0 = atomicAdd(ra.pool, need);  // stock used up (a burst of switches): costs a stall
          split[tid] = B + n1;
          if ((uint64_t)id0 + need <= ra.list_cap) {
            for (uint32_t i = 0; i < need; ++i) file_chunk(tid, id0 + i);  // consecutive ids: the rest of the block is one run
            for (uint32_t i = 0; i + 1 < need; ++i) ra.cmeta[id0 + i] = ((uint32_t)tid << 20) | CH;
            const uint64_t nb = (uint64_t)id0 << lg;
            delta2[tid] = nb - (B + n1);
            o_cur[tid] = nb + rest;
            o_cend[tid] = nb + ((uint64_t)need << lg);
          } else {  // the pool is dry: the rest of the block goes straight to the output; the partition keeps asking
            delta2[tid] = kRouteDirect;
            o_cur[tid] = cend;
            if (ra.dry) *ra.dry = 1u;
            if constexpr (PACK) atomicOr(ra.flags, 3u);
          }
        }
        cin2[((cur_set ^ 1) << 8) + tid] = T - whole;
        if (tid == P - 1) misc[0] = B + block;
      }
    }
    if (tid < 256) cnt2[((cur_set ^ 1) << 8) + tid] = 0u;
    __syncthreads();
    // ---- the tile, sorted by partition, into LDS -----------------------------------------------------
#pragma unroll
    for (int u = 0; u < U; ++u) {
      w4 wu = w[u];
      if (kWeighted && ragged) {
        const int64_t i = base + ((int64_t)u * BLOCK + tid) * 4;
        wu = pulled_back(w[u], (int)min(i - min(i, n - 4), (int64_t)4), (wscalar)0);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (flat[u][v] != 0xffffffffu) {
          const uint32_t part = flat[u][v] >> shift;
          const uint32_t slot = first[part] + rank[u][v];
          skey[slot] = (part << 16) | (flat[u][v] & code_mask);
          if constexpr (PACK) sw[slot] = pack48((double)wu[v], flat[u][v] & code_mask);
          else if constexpr (kWeighted) sw[slot] = (RT)wu[v];
        }
      if constexpr (PACK) {  // signs of every weight read, dropped samples included (zeros, NaNs and the ragged tail's fill are neutral)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          w_lo = fmin(w_lo, (double)wu[v]);
          w_hi = fmax(w_hi, (double)wu[v]);
        }
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
      if constexpr (!WSPLIT) {
        w[u] = wn[u];
        if (kWeighted) asm volatile("" : "+v"(w[u]));  // the wait for the prefetch sits here, ahead of the stores
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        asm volatile("" : "+v"(xv[d][u]));
      }
    }
    __syncthreads();
    const uint32_t total = misc[0];
    // ---- records out: one lane per group of 8 codes / per 16 bytes of weights -------------------------
    for (uint32_t g0 = (uint32_t)tid * GRP; g0 < total; g0 += BLOCK * GRP) {
      uint32_t kk[GRP];
#pragma unroll
      for (int i = 0; i < GRP; i += 4) {
        const u4 q4 = *reinterpret_cast<const u4*>(skey + g0 + i);
        kk[i] = q4[0]; kk[i + 1] = q4[1]; kk[i + 2] = q4[2]; kk[i + 3] = q4[3];
      }
      const uint32_t q = kk[0] >> 16;
      if (g0 + GRP <= endw[q]) {
        const uint64_t dlt = g0 < split[q] ? delta[q] : delta2[q];
        if (dlt == kRouteDirect) {
#pragma unroll
          for (int i = 0; i < GRP; ++i) direct_add(kk[i], kWeighted ? sw[g0 + i] : (RT)0);
          continue;
        }
        const uint64_t dst = dlt + g0;
        if constexpr (!PACK) {
          u4 c4;
#pragma unroll
          for (int i = 0; i < 4; ++i) c4[i] = (kk[2 * i] & 0xffffu) | (kk[2 * i + 1] << 16);
          __builtin_nontemporal_store(c4, reinterpret_cast<u4*>(codes + dst));
        }
      } else {
        const uint32_t left = enda[q] - g0;  // 1 .. GRP-1 records: the carry
#pragma unroll
        for (int i = 0; i < GRP; ++i)
          if ((uint32_t)i < left) {
            carry_key[q * GRP + i] = kk[i];
            if (kWeighted) carry_w[q * GRP + i] = sw[g0 + i];
          }
      }
    }
    if (kWeighted) {
      static_assert(GRP % RV == 0, "16-byte weight stores never straddle a group");
      for (uint32_t t0 = (uint32_t)tid * RV; t0 < total; t0 += BLOCK * RV) {
        const uint32_t g0 = t0 & ~kGm;
        const uint32_t q = skey[g0] >> 16;
        if (g0 + GRP <= endw[q]) {
          const uint64_t dlt = g0 < split[q] ? delta[q] : delta2[q];
          if (dlt == kRouteDirect) continue;  // (added to the output by the loop above)
          const rvec wq = *reinterpret_cast<const rvec*>(sw + t0);
          __builtin_nontemporal_store(wq, reinterpret_cast<rvec*>(wrec + dlt + t0));
        }
      }
    }
    // ---- the id stock: lane 0 moves the second range up when the first is used up, and asks the pool for a new
    // second range; the answer is picked up one tile later (nobody waits for it) -----------------------------------
    if (tid == 0) {
      if (refill_pending) {
        if (stock[2] >= stock[3]) {  // (always, unless the burst path refilled... it never does: the range is simply replaced)
          stock[2] = refill_ids;
          stock[3] = refill_ids + batch;
        }
        refill_pending = false;
      }
      if (stock[0] + max_need > stock[1] && stock[2] < stock[3]) {  // the first range can no longer serve the largest request
        stock[0] = stock[2];
        stock[1] = stock[3];
        stock[2] = stock[3] = 0u;
      }
      if (stock[2] >= stock[3]) {
        refill_ids = atomicAdd(ra.pool, batch);
        refill_pending = true;
      }
    }
    // no barrier here: the next tile's ranking touches only the other counter set, and nobody passes
    // that tile's first barrier before every lane has finished this write-out
  }
  __syncthreads();
  // the last group of each partition (carried records padded with neutral ones), the fill of the chunk in use,
  // and this workgroup's chunk lists handed over to the partitions' global lists
  if (tid < P) {
    const uint32_t my_carry = cin2[(cur_set << 8) + tid];
    uint64_t cur = o_cur[tid], cend = o_cend[tid];
    if (my_carry != 0u) {
      bool dry = false;
      if (cur == cend) {
        if (cend != 0) ra.cmeta[(uint32_t)((cend - 1) >> lg)] = ((uint32_t)tid << 20) | CH;
        uint32_t id0 = route_take_ids(stock, 1u);
        if (id0 == 0xffffffffu) id0 = atomicAdd(ra.pool, 1u);
        if ((uint64_t)id0 + 1u <= ra.list_cap) {
          file_chunk(tid, id0);
          cur = (uint64_t)id0 << lg;
          cend = cur + CH;
        } else {
          dry = true;
          if (ra.dry) *ra.dry = 1u;
          if constexpr (PACK) atomicOr(ra.flags, 3u);
        }
      }
      if (dry) {
        for (uint32_t i = 0; i < my_carry; ++i) direct_add(carry_key[tid * GRP + i], kWeighted ? carry_w[tid * GRP + i] : (RT)0);
      } else {
        for (int i = 0; i < GRP; ++i) {
          const bool real = (uint32_t)i < my_carry;
          if constexpr (!PACK) codes[cur + i] = real ? (uint16_t)(carry_key[tid * GRP + i] & 0xffffu) : (uint16_t)(kWeighted ? 0u : (1u << shift));
          if (kWeighted) wrec[cur + i] = real ? carry_w[tid * GRP + i] : (RT)0;  // (packed: +0.0 for bin 0)
        }
        cur += GRP;
      }
    }
    if (cend != 0) ra.cmeta[(uint32_t)((cend - 1) >> lg)] = ((uint32_t)tid << 20) | (uint32_t)(cur - (cend - CH));
    const uint32_t mine = ccnt[tid];
    if (mine != 0u) {
      const uint32_t pos0 = atomicAdd(ra.pcount + tid, mine);
      uint32_t e = head[tid];
      for (uint32_t j = 0; j < mine; ++j) {
        ra.plist[(size_t)tid * ra.list_cap + pos0 + j] = cl_id[e];
        e = cl_prev[e];
      }
    }
  }
  if constexpr (PACK) {
    const uint32_t signs = (__ballot(w_lo < 0.0) ? 1u : 0u) | (__ballot(w_hi > 0.0) ? 2u : 0u);
    if ((tid & 63) == 0 && signs) atomicOr(ra.flags, signs);
  }
}

// The adding-up pass over chunk lists.  Chunk k of the concatenation of all partitions' lists belongs to the
// partition whose offset range holds k; every workgroup takes an equal range of k.  Chunks hold whole
// groups of 8 records and start 2^chunk_log2-aligned, so every load is an aligned quad.
template <bool WEIGHTED, typename RT = double, bool PACK = false>
__global__ void __launch_bounds__(1024) part_accumulate_chunks(const RouteArgs ra, void* out_v, int64_t n_bins, int shift, int P, int parts_per_row) {
  if (route_gate_closed(ra)) return;
  using lds_t = typename std::conditional<WEIGHTED, double, uint32_t>::type;
  using out_t = typename std::conditional<WEIGHTED, double, unsigned long long>::type;
  lds_t* hist = reinterpret_cast<lds_t*>(xhist_smem);
  out_t* out = reinterpret_cast<out_t*>(out_v);
  const RT* wrec = static_cast<const RT*>(ra.wrec);
  const uint16_t* codes = ra.codes;
  const uint32_t bpp = 1u << shift;
  const int tid = threadIdx.x, lg = ra.chunk_log2;
  unsigned char* after = xhist_smem + (((size_t)(bpp + 1) * sizeof(lds_t) + 15) & ~(size_t)15);
  uint64_t* tbl = reinterpret_cast<uint64_t*>(after);          // [kAccBatch] chunk id | fill << 32
  uint32_t* offs = reinterpret_cast<uint32_t*>(tbl + kAccBatch);  // [P + 1] first chunk index of each partition
  for (uint32_t c = tid; c <= bpp; c += blockDim.x) hist[c] = (lds_t)0;  // [bpp] = trash slot of padding records
  if (tid == 0) {
    uint32_t run = 0;
    for (int q = 0; q < P; ++q) {
      offs[q] = run;
      run += ra.pcount[q];
    }
    offs[P] = run;
  }
  __syncthreads();
  const uint32_t total = offs[P];
  uint32_t lo = total / gridDim.x * blockIdx.x + min((uint32_t)blockIdx.x, total % gridDim.x);
  const uint32_t hi = lo + total / gridDim.x + (blockIdx.x < total % gridDim.x ? 1u : 0u);
  int part = 0;
  typedef uint16_t c4 __attribute__((ext_vector_type(4)));
  typedef RT w4 __attribute__((ext_vector_type(4)));
  constexpr int kGroups = 2;  // record quads per lane in flight (1 / 2 / 4: 717 / 714 / 716 us for C5, profiles/r04_*acc_groups*)
  const int qlg = lg - 2;  // quads per chunk, log2
  while (lo < hi) {
    while (part + 1 < P && offs[part + 1] <= lo) ++part;
    const uint32_t pend = min(hi, offs[part + 1]);
    for (uint32_t b0 = lo; b0 < pend; b0 += kAccBatch) {
      const uint32_t nb = min((uint32_t)kAccBatch, pend - b0);
      for (uint32_t j = tid; j < nb; j += blockDim.x) {
        const uint32_t id = ra.plist[(size_t)part * ra.list_cap + (b0 + j - offs[part])];
        tbl[j] = (uint64_t)id | ((uint64_t)(ra.cmeta[id] & kChunkFillMask) << 32);
      }
      __syncthreads();
      // Loads free of control flow, several per lane in flight; the adds of a load under `if (live)`.  A record is a 2-byte
      // code for counts, so a lane takes eight (one 16-byte load) and four such loads; weighted records come four to a lane
      // and load (quads: 32 bytes of float64 / packed records).  Pieces past a chunk's fill or past the batch re-read the
      // head of a chunk and are dropped.  (Round 2's form had the loads under the `if` as well — the compiler then waits for
      // one group's loads before it issues the next — and 8-byte loads of codes: 16 KB per CU in flight for counts, 5*10^8
      // float32 samples into 10^5 bins 0.31 ms -> 0.17.  Sending the dropped pieces' adds to a trash slot instead of
      // branching around them took 1.2 ms: lanes that meet on one LDS address inside otherwise scattered adds are slow.)
      // Fills are multiples of 8: records leave the routing pass in whole groups.
      if constexpr (!WEIGHTED) {
        typedef uint16_t c8 __attribute__((ext_vector_type(8)));
        constexpr int kOct = 4;
        const int olg = lg - 3;
        const uint32_t totalo = nb << olg;
        for (uint32_t O = tid; O < totalo; O += 1024 * kOct) {
          c8 cv[kOct];
          bool live[kOct];
#pragma unroll
          for (int g = 0; g < kOct; ++g) {
            const uint32_t Og = min(O + (uint32_t)g * 1024u, totalo - 1u);
            const uint64_t e = tbl[Og >> olg];
            const uint32_t within = (Og & ((1u << olg) - 1u)) << 3;
            live[g] = (O + (uint32_t)g * 1024u < totalo) & (within < (uint32_t)(e >> 32));
            const uint64_t at = ((uint64_t)(uint32_t)e << lg) + (live[g] ? within : 0u);
            cv[g] = __builtin_nontemporal_load(reinterpret_cast<const c8*>(codes + at));
          }
#pragma unroll
          for (int g = 0; g < kOct; ++g)
            if (live[g]) {
#pragma unroll
              for (int k = 0; k < 8; ++k) atomicAdd(reinterpret_cast<uint32_t*>(hist) + (uint32_t)cv[g][k], 1u);
            }
        }
      } else {
        const uint32_t totalq = nb << qlg;
        for (uint32_t Q = tid; Q < totalq; Q += 1024 * kGroups) {
          c4 cv[kGroups];
          w4 wq[kGroups];
          bool live[kGroups];
#pragma unroll
          for (int g = 0; g < kGroups; ++g) {
            const uint32_t Qg = min(Q + (uint32_t)g * 1024u, totalq - 1u);
            const uint64_t e = tbl[Qg >> qlg];
            const uint32_t within = (Qg & ((1u << qlg) - 1u)) << 2;
            live[g] = (Q + (uint32_t)g * 1024u < totalq) & (within < (uint32_t)(e >> 32));
            const uint64_t at = ((uint64_t)(uint32_t)e << lg) + (live[g] ? within : 0u);
            // (8-byte weights — 32 bytes a lane and load — are 3 % faster with their loads under the `if` after all: packed
            // C5 records 0.704-0.712 ms against 0.728-0.735; 4-byte weights the other way round, 0.489 against 0.502)
            if (sizeof(RT) < 8 || live[g]) {
              if constexpr (!PACK) cv[g] = __builtin_nontemporal_load(reinterpret_cast<const c4*>(codes + at));
              wq[g] = __builtin_nontemporal_load(reinterpret_cast<const w4*>(wrec + at));
            }
          }
#pragma unroll
          for (int g = 0; g < kGroups; ++g)
            if (live[g]) {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if constexpr (PACK) {
                  const uint64_t r = (uint64_t)__double_as_longlong((double)wq[g][k]);
                  unsafeAtomicAdd(reinterpret_cast<double*>(hist) + (uint32_t)(r & 0xffffull), __longlong_as_double((long long)(r & ~0xffffull)));
                } else {
                  unsafeAtomicAdd(reinterpret_cast<double*>(hist) + cv[g][k], (double)wq[g][k]);
                }
              }
            }
        }
      }
      __syncthreads();
    }
    for (uint32_t c = tid; c < bpp; c += blockDim.x) {
      const lds_t v = hist[c];
      if (v != (lds_t)0) {
        // (several rows in one pass: partition = row * parts_per_row + partition inside the row's histogram)
        const int row = parts_per_row ? part / parts_per_row : 0;
        const int64_t bin = ((int64_t)(part - row * parts_per_row) << shift) + c;
        if (bin < n_bins) {
          if (WEIGHTED) unsafeAtomicAdd(reinterpret_cast<double*>(out) + row * n_bins + bin, (double)v);
          else atomicAdd(reinterpret_cast<unsigned long long*>(out) + row * n_bins + bin, (unsigned long long)v);
        }
        hist[c] = (lds_t)0;
      }
    }
    if (tid == 0 && !WEIGHTED) hist[bpp] = (lds_t)0;
    __syncthreads();
    lo = pend;
  }
}

}  // namespace xhist
