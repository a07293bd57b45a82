// xchg.hip — development micro-benchmark (not part of the product): a beyond-LDS weighted 2-D histogram (BASELINE C5:
// 1024 x 1024 float64 bins = 8 MiB) WITHOUT a record round trip through HBM for the bulk of the samples.
//
// Idea.  One XCD has 32 CUs x 160 KB of LDS = 5 MB: a WINDOW of 480 histogram rows (480 x 1024 float64 = 3.75 MiB) fits
// the LDS of one XCD when every CU keeps 15 rows.  One persistent workgroup per CU is producer AND consumer: it reads a
// tile of samples, digitizes, sorts the tile's records by the CU that owns their row (row mod 32) in LDS and writes them
// into small rings — one per (producer, consumer) pair of the SAME XCD, so the consumer's reads are served by the XCD's
// L2 — then polls the 32 rings that end at it and adds what has arrived into its LDS rows.  A record is one 8-byte word
// {48-bit weight, 14-bit bin inside the owner's rows, 2-bit lap tag}: the tag makes the word its own "ready" flag (no
// fences, no separate flags, nothing ordered), the consumer publishes how far it has read (credits) once per step.
// Samples outside the window go to the output with memory-side atomics (fine for a few per cent of them).
// HBM traffic: 24 B read + 8 B written per sample (stores leave the L2 whatever one does), nothing read back.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o xchg xchg.hip
// Run:   ./xchg [dist 0=N(0,1) 1=uniform] [mode 0=exchange 1=everything to atomics 2=records written, never consumed] [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

#ifndef XSPL
#define XSPL 4
#endif
#ifndef XCAP_LOG2
#define XCAP_LOG2 8
#endif
#ifndef XCL
#define XCL 6
#endif
#ifndef XEARLY
#define XEARLY 0
#endif
#ifndef XTIME
#define XTIME 0
#endif
#ifndef XAUX
#define XAUX -1
#endif
#ifndef XSAME
#define XSAME 0
#endif
#ifndef XHINT
#define XHINT 0
#endif
#ifndef XCOLW
#define XCOLW 0
#endif
#ifndef XSTEAL
#define XSTEAL 0
#endif
#ifndef XEXACT
#define XEXACT 0
#endif
#ifndef XLD
#define XLD 0
#endif
#ifndef XNT
#define XNT 1
#endif
#ifndef XPRE
#define XPRE 0
#endif
#ifndef XDPP
#define XDPP 0
#endif
#ifndef XDIRECT
#define XDIRECT 0
#endif
constexpr int BLOCK = 1024, SPL = XSPL, TILE = BLOCK * SPL, U = SPL / 2;
constexpr int NX = 8, NS = 32;              // XCDs; workgroups (= ring ends) per XCD
constexpr int NBY = 1024, NBX = 1024;
constexpr int ROWS_PER_PART = XCOLW ? 20 : 15, COL0 = XCOLW ? 128 : 0, COLW = XCOLW ? 768 : 1024;
constexpr int ROWS_PER_PART_UNUSED = 15, PART_BINS = ROWS_PER_PART * COLW, WIN_ROWS = ROWS_PER_PART * NS;  // 480 rows in the window
constexpr int CAP_LOG2 = XCAP_LOG2, CAP = 1 << CAP_LOG2;  // records per ring
constexpr int CL = XCL;                       // ring records a lane looks at per step (32 lanes per ring: 192 records)
constexpr int PART_BYTES = PART_BINS * 8;
constexpr int LDS_BYTES = PART_BYTES + TILE * 8 + TILE + 2048;

struct Ctl {                 // one per XCD
  uint32_t nreg;             // workgroups that registered on this XCD
  uint32_t pad[31];
  uint32_t head[NS][NS];     // [producer p][consumer d]: records of ring (p -> d) the consumer has taken
  uint32_t fin[NS][NS];      // [producer p][consumer d]: final record count of the ring (0xFFFFFFFF while p produces)
  uint32_t tailpub[NS][NS];  // [producer p][consumer d]: XHINT — records p has written (or is writing) into the ring: a hint of how far to look
};

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ uint64_t ld_l2(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, XLD == 0 ? __HIP_MEMORY_SCOPE_AGENT : XLD == 1 ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ uint32_t ld_l2(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_l2(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef double d2 __attribute__((ext_vector_type(2)));

// status words: [0] error code, [1] stall iterations, [2] workgroups that finished, [3] records consumed (low 32)
template <int MODE>
__global__ void __launch_bounds__(BLOCK) xchg(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ w,
                                              int64_t n, Ctl* ctl, uint64_t* rings, double* part_out, double* out, int row0,
                                              uint32_t* status, long long budget_ticks) {
  extern __shared__ unsigned char smem[];
  double* hist = reinterpret_cast<double*>(smem);
  uint64_t* srec = reinterpret_cast<uint64_t*>(smem + PART_BYTES);
  uint8_t* sd = smem + PART_BYTES + TILE * 8;
  uint32_t* c = reinterpret_cast<uint32_t*>(sd + TILE);
  uint32_t* cnt2 = c;           // [2][64] records per destination of the tile (32 rings, 32 = atomics, 33 = dropped)
  uint32_t* off = c + 128;      // [64] first sorted slot of every destination
  uint32_t* tail = c + 192;     // [32] records this workgroup has sent to consumer d (all earlier tiles)
  uint32_t* wbase = c + 224;    // [32] ring position of this tile's first record for d
  uint32_t* credit = c + 256;   // [32] head[me][d] as last seen
  uint32_t* chead = c + 288;    // [32] records taken from ring (p -> me)
  uint32_t* misc = c + 320;     // [0] xcd [1] slot [3] stall [4] abort
  const int tid = threadIdx.x;
  const long long t_start = wall_clock64();

  for (int i = tid; i < PART_BINS; i += BLOCK) hist[i] = 0.0;
  if (tid < 128) cnt2[tid] = 0u;
  if (tid < 32) { tail[tid] = 0u; credit[tid] = 0u; chead[tid] = 0u; }
  if (tid == 0) {
    const uint32_t xc = xcc_id();
    misc[0] = xc;
    misc[1] = atomicAdd(&ctl[xc & 7].nreg, 1u);
    misc[3] = 0u;
    misc[4] = 0u;
    misc[8] = 0u;
    misc[9] = 0u;
  }
  __syncthreads();
  const uint32_t xcd = misc[0] & 7u, me = misc[1];
  if (me >= (uint32_t)NS) {  // more than 32 workgroups on one XCD: the protocol has no ring end for this one
    if (tid == 0) atomicExch(status + 0, 2u);
    return;
  }
  Ctl& C = ctl[xcd];
  uint64_t* xring = rings + (size_t)xcd * NS * NS * CAP;          // [consumer d][producer p][CAP]
  const uint64_t* myring = xring + (size_t)me * NS * CAP;         // rings that end here: [p][CAP]

  const int64_t n_tiles = n / TILE;  // (the benchmark's n is a multiple of the tile)
  const int64_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  auto tile_ptr = [&](int64_t k) { return XSAME ? (int64_t)tid * 2 : ((int64_t)blockIdx.x + k * gridDim.x) * TILE + (int64_t)tid * 2; };
  d2 xv[U], yv[U], wv[U];
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)0xffffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)0xffffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)0xffffffff, 0x00020000);
  auto load_tile = [&](int64_t k) {
    const int64_t b = tile_ptr(k);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (XAUX >= 0) {
        const int off = (int)((uint32_t)(b + u * 2 * BLOCK) * 8u);
        u4v a = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, XAUX < 0 ? 0 : XAUX);
        u4v bb = __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, XAUX < 0 ? 0 : XAUX);
        u4v cc = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, XAUX < 0 ? 0 : XAUX);
        xv[u] = __builtin_bit_cast(d2, a);
        yv[u] = __builtin_bit_cast(d2, bb);
        wv[u] = __builtin_bit_cast(d2, cc);
      } else {
        xv[u] = XNT ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(x + b + u * 2 * BLOCK)) : *reinterpret_cast<const d2*>(x + b + u * 2 * BLOCK);
        yv[u] = XNT ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(y + b + u * 2 * BLOCK)) : *reinterpret_cast<const d2*>(y + b + u * 2 * BLOCK);
        wv[u] = XNT ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(w + b + u * 2 * BLOCK)) : *reinterpret_cast<const d2*>(w + b + u * 2 * BLOCK);
      }
    }
  };
  // the consumer side: 32 lanes per ring (p = tid / 32), CL records each
  const uint32_t psub = (uint32_t)tid >> 5, l = (uint32_t)tid & 31u;
  uint64_t rr[CL];
  uint32_t rh = 0;  // head the loads in rr were issued at
  uint32_t hint = 0;  // XHINT: the ring's published tail as of the end of the tile before
  auto issue_ring_loads = [&]() {
    rh = chead[psub];
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      const uint32_t pos = rh + l + 32u * j;
      if (!XHINT || (int32_t)(hint - pos) > 0) rr[j] = ld_l2(myring + (size_t)psub * CAP + (pos & (uint32_t)(CAP - 1)));
      else rr[j] = ~0ull ^ ((uint64_t)((((pos >> CAP_LOG2) + 1u) & 3u)) << 14);  // (a tag that is certainly not the expected one)
    }
  };
  uint32_t taken_total = 0;
  auto take_ring_records = [&]() -> uint32_t {  // adds the valid prefix of what was loaded; returns its length (per ring)
    uint32_t pre = 0;
    bool cont = true;
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      const uint32_t pos = rh + l + 32u * j;
      const uint32_t expect = ((pos >> CAP_LOG2) + 1u) & 3u;
      const bool valid = (((uint32_t)rr[j] >> 14) & 3u) == expect;
      const uint64_t bal = __builtin_amdgcn_ballot_w64(valid);
      const uint32_t m = (tid & 32) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
      if (cont) {
        if (m == 0xffffffffu) pre += 32u;
        else { pre += (uint32_t)__builtin_ctz(~m); cont = false; }
      }
    }
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      if (l + 32u * j < pre) {
        const uint64_t r = rr[j];
        unsafeAtomicAdd(hist + ((uint32_t)r & 0x3fffu), __longlong_as_double((long long)(r & ~0xffffull)));
      }
    }
    if (l == 0 && pre) {
      chead[psub] = rh + pre;
      st_l2(&C.head[psub][me], rh + pre);
      taken_total += pre;
    }
    return pre;
  };


  // XSTEAL: whoever has issued its loads takes ring pairs by ticket until none is left (the wavefronts whose loads were
  // accepted first otherwise just wait at the barrier for the last one).  Measured SLOWER (3.49 against 3.25 ms): ring loads
  // issued behind a wavefront's own prefetch wait for that prefetch (the memory counter counts in order), and under load the
  // first wavefront's samples arrive no earlier than the last wavefront's loads are accepted.
  auto take_by_ticket = [&](uint32_t* ticket) {
    for (;;) {
      uint32_t t = 0;
      if ((tid & 63) == 0) t = atomicAdd(ticket, 1u);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      if (t >= 16u) break;
      const uint32_t ring = 2u * t + (((uint32_t)tid >> 5) & 1u);
      const uint32_t h = chead[ring];
      uint64_t q[CL];
#pragma unroll
      for (int j = 0; j < CL; ++j) q[j] = ld_l2(myring + (size_t)ring * CAP + ((h + l + 32u * j) & (uint32_t)(CAP - 1)));
      uint32_t pre = 0;
      bool cont = true;
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        const uint32_t pos = h + l + 32u * j;
        const bool valid = (((uint32_t)q[j] >> 14) & 3u) == (((pos >> CAP_LOG2) + 1u) & 3u);
        const uint64_t bal = __builtin_amdgcn_ballot_w64(valid);
        const uint32_t m = (tid & 32) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
        if (cont) {
          if (m == 0xffffffffu) pre += 32u;
          else { pre += (uint32_t)__builtin_ctz(~m); cont = false; }
        }
      }
#pragma unroll
      for (int j = 0; j < CL; ++j)
        if (l + 32u * j < pre) unsafeAtomicAdd(hist + ((uint32_t)q[j] & 0x3fffu), __longlong_as_double((long long)(q[j] & ~0xffffull)));
      if (l == 0 && pre) {
        chead[ring] = h + pre;
        st_l2(&C.head[ring][me], h + pre);
        taken_total += pre;
      }
    }
  };
  uint32_t cred_next = 0;
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;
#define PH(i) do { if (XTIME) { const long long t_ = clock64(); ph[i] += t_ - tc; tc = t_; } } while (0)
  if (XTIME) tc = clock64();
  if (my_tiles > 0) load_tile(0);
  int buf = 0;
  uint32_t stalls = 0;
  bool aborted = false;
  for (int64_t k = 0; k < my_tiles; ++k, buf ^= 1) {
    uint32_t* cnt = cnt2 + buf * 64;
    // ring loads first (older than everything below): waiting for them leaves the next tile's samples in flight
    if (XPRE && MODE == 0) {
      issue_ring_loads();
      if (tid < 32) cred_next = ld_l2(&C.head[me][tid]);
    }
    // ---- digitize (uniform bins, benchmark-grade arithmetic), pack ----------------------------------------
    uint32_t dest[SPL];
    uint64_t rec[SPL];
    uint32_t flat[SPL];
    bool near_any = false;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int s = u * 2 + v;
#if XEXACT
        // the product's arithmetic digitize (bin_arith_fast): 9 float64 operations per sample and dimension, plus pack48's rounding
        double ttx = (xv[u][v] - (-4.0)) * 128.0, tty = (yv[u][v] - (-4.0)) * 128.0;
        asm volatile("" : "+v"(ttx), "+v"(tty));
        const double flx = __builtin_floor(ttx), fly = __builtin_floor(tty);
        const double fx = ttx - flx, fy = tty - fly;
        const bool near = !(__builtin_fabs(fx - 0.5) < 0.4999) | !(__builtin_fabs(fy - 0.5) < 0.4999);
        near_any |= near;
        const int xb0 = (int)(uint32_t)(uint64_t)__double_as_longlong(flx + 6755399441055744.0), yb0 = (int)(uint32_t)(uint64_t)__double_as_longlong(fly + 6755399441055744.0);
        const bool ok = (flx >= 0.0) & (flx < 1024.0) & (fly >= 0.0) & (fly < 1024.0);
        const int xb = ok ? xb0 : 0, yb = ok ? yb0 : 0;
        {
          const uint64_t b_ = (uint64_t)__double_as_longlong(wv[u][v]);
          wv[u][v] = __longlong_as_double((long long)(b_ + 0x7fffull + ((b_ >> 16) & 1ull)));
        }
#else
        const double tx = (xv[u][v] + 4.0) * 128.0, ty = (yv[u][v] + 4.0) * 128.0;
        const bool ok = (tx >= 0.0) & (tx < 1024.0) & (ty >= 0.0) & (ty < 1024.0);
        const int xb = ok ? (int)tx : 0, yb = ok ? (int)ty : 0;
#endif
        const uint32_t r = (uint32_t)(xb - row0);
        const bool in_win = ok & (r < (uint32_t)WIN_ROWS) & ((uint32_t)(yb - COL0) < (uint32_t)COLW) & (MODE != 1);
        dest[s] = !ok ? 33u : (in_win ? (r & 31u) : 32u);
        const uint32_t local = (r >> 5) * (uint32_t)COLW + (uint32_t)(yb - COL0);
        rec[s] = ((uint64_t)__double_as_longlong(wv[u][v]) & ~0xffffull) | (in_win ? local : 0u);
        flat[s] = (uint32_t)xb * NBY + (uint32_t)yb;
      }
    if (XEXACT && __builtin_amdgcn_ballot_w64(near_any) != 0ull) { if (near_any) atomicAdd(status + 5, 1u); }
    PH(0);
    if (!XPRE && MODE == 0) {
      if (!XSTEAL) issue_ring_loads();
      if (tid < 32) cred_next = ld_l2(&C.head[me][tid]);
    }
    // samples outside the window: straight to the output (registers are free before the prefetch)
#pragma unroll
    for (int s = 0; s < SPL; ++s)
      if (dest[s] == 32u) unsafeAtomicAdd(out + flat[s], __longlong_as_double((long long)rec[s]));
    load_tile(k + 1 < my_tiles ? k + 1 : k);
    uint32_t rank[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) rank[s] = dest[s] < 32u ? atomicAdd(cnt + dest[s], 1u) : 0u;
    if (XSTEAL && MODE == 0) take_by_ticket(misc + 8 + buf);
    if (!XSTEAL && XPRE && MODE == 0) take_ring_records();
    PH(1);
    __syncthreads();
    PH(5);
    // ---- wavefront 0: block layout, credits; everybody: take what arrived ---------------------------------
    if (tid < 64) {
      const uint32_t cn = tid < 32 ? cnt[tid] : 0u;
      uint32_t xs = cn;
      if (XDPP) {
        xs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x111, 0xf, 0xf, false);
        xs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x112, 0xf, 0xf, false);
        xs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x114, 0xf, 0xf, false);
        xs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x118, 0xf, 0xf, false);
        xs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x142, 0xa, 0xf, false);
        xs += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)xs, 0x143, 0xc, 0xf, false);
      } else {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t t = __shfl_up(xs, o, 64);
          if (tid >= o) xs += t;
        }
      }
      off[tid] = xs - cn;
      if (tid < 32) {
        credit[tid] = cred_next;
        const uint32_t t = tail[tid];
        wbase[tid] = t;
        tail[tid] = t + cn;
        if (MODE == 0 && t + cn - credit[tid] > (uint32_t)CAP) misc[3] = 1u;
      }
      cnt2[(buf ^ 1) * 64 + tid] = 0u;
    }
    if (!XSTEAL && !XPRE && MODE == 0) take_ring_records();
    if (XSTEAL && tid == 0) misc[8 + (buf ^ 1)] = 0u;
    PH(6);
    __syncthreads();
    PH(7);
    // ---- a ring without room: keep taking (so that nobody waits for this workgroup) until the consumers caught up --
    while (MODE == 0 && misc[3]) {
      ++stalls;
      if (XHINT) hint = ld_l2(&C.tailpub[psub][me]);
      issue_ring_loads();
      take_ring_records();
      if (tid < 32) credit[tid] = ld_l2(&C.head[me][tid]);
      __syncthreads();
      if (tid == 0) {
        bool ok = true;
        for (int d = 0; d < NS; ++d) ok &= (tail[d] - credit[d] <= (uint32_t)CAP);
        if (ok) misc[3] = 0u;
        if (wall_clock64() - t_start > budget_ticks) { misc[4] = 1u; misc[3] = 0u; }
      }
      __syncthreads();
    }
    if (misc[4]) { aborted = true; break; }
    PH(2);
    if (XDIRECT) {
      // ---- records straight from the registers to their ring slots (no staging: a wavefront's stores are scattered) ----
#pragma unroll
      for (int s = 0; s < SPL; ++s)
        if (dest[s] < 32u) {
          const uint32_t d = dest[s];
          const uint32_t pos = wbase[d] + rank[s];
          const uint32_t tag = ((pos >> CAP_LOG2) + 1u) & 3u;
          xring[((size_t)d * NS + me) * CAP + (pos & (uint32_t)(CAP - 1))] = (rec[s] & ~0xc000ull) | ((uint64_t)tag << 14);
        }
      PH(4);
    } else {
      // ---- the tile's ring records, sorted by consumer, into LDS ------------------------------------------------
#pragma unroll
      for (int s = 0; s < SPL; ++s)
        if (dest[s] < 32u) {
          const uint32_t i = off[dest[s]] + rank[s];
          srec[i] = rec[s];
          sd[i] = (uint8_t)dest[s];
        }
      __syncthreads();
      PH(3);
      // ---- out: runs of consecutive records per ring -------------------------------------------------------------
      const uint32_t total = off[32];
#pragma unroll
      for (int q = 0; q < SPL; ++q) {
        const uint32_t i = (uint32_t)tid + q * BLOCK;
        if (i < total) {
          const uint32_t d = sd[i];
          const uint32_t pos = wbase[d] + (i - off[d]);
          const uint32_t tag = ((pos >> CAP_LOG2) + 1u) & 3u;
          xring[((size_t)d * NS + me) * CAP + (pos & (uint32_t)(CAP - 1))] = (srec[i] & ~0xc000ull) | ((uint64_t)tag << 14);
        }
      }
      PH(4);
    }
    if (XHINT && MODE == 0) {
      if (tid < 32) st_l2(&C.tailpub[me][tid], tail[tid]);
      hint = ld_l2(&C.tailpub[psub][me]);
    }
  }
  __syncthreads();
  if (aborted) {
    if (tid == 0) atomicExch(status + 0, 3u);
    return;
  }
  // ---- no more records from here: say so, then drain the rings that end here -----------------------------------
  if (tid < 32) st_l2(&C.fin[me][tid], tail[tid]);
  if (MODE == 0) {
    for (;;) {
      if (XHINT) hint = ld_l2(&C.tailpub[psub][me]);
      issue_ring_loads();
      const uint32_t pre = take_ring_records();
      bool done = false;
      if (l == 0 && pre == 0) done = (ld_l2(&C.fin[psub][me]) == chead[psub]);
      done = __shfl(done ? 1 : 0, (tid & 32), 64) != 0;
      if (tid == 0 && wall_clock64() - t_start > budget_ticks) misc[4] = 1u;
      if (__syncthreads_and(done ? 1 : 0)) break;
      if (misc[4]) { aborted = true; break; }
      __syncthreads();
    }
    if (aborted) {
      if (tid == 0) atomicExch(status + 0, 4u);
      return;
    }
  }
  __syncthreads();
  double* po = part_out + ((size_t)xcd * NS + me) * PART_BINS;
  for (int i = tid; i < PART_BINS; i += BLOCK) po[i] = hist[i];
  if (tid == 0) {
    atomicAdd(status + 1, stalls);
    atomicAdd(status + 2, 1u);
  }
  if (l == 0) atomicAdd(status + 3, taken_total);
  if (XTIME && (tid == 0 || tid == 64 || tid == 960) && blockIdx.x == 17)
    for (int i = 0; i < 8; ++i) status[16 + (tid == 0 ? 0 : tid == 64 ? 8 : 16) + i] = (uint32_t)(ph[i] / (my_tiles > 0 ? my_tiles : 1));
}

// the XCD partials of the window's rows, added up into the output (which holds what went there directly)
__global__ void merge(const double* part_out, double* out, int row0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over WIN_ROWS * NBY
  if (i >= WIN_ROWS * NBY) return;
  const int r = i / NBY, col = i % NBY;
  const int d = r & 31, local = (r >> 5) * COLW + (col - COL0);
  if ((unsigned)(col - COL0) >= (unsigned)COLW) return;
  double s = 0.0;
  for (int xc = 0; xc < NX; ++xc) s += part_out[((size_t)xc * NS + d) * PART_BINS + local];
  out[(size_t)(row0 + r) * NBY + col] += s;
}

__global__ void reference(const double* x, const double* y, const double* w, int64_t n, double* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double tx = (x[i] + 4.0) * 128.0, ty = (y[i] + 4.0) * 128.0;
    if ((tx >= 0.0) & (tx < 1024.0) & (ty >= 0.0) & (ty < 1024.0))
      unsafeAtomicAdd(out + (size_t)((int)tx) * NBY + (int)ty,
                      __longlong_as_double((long long)((uint64_t)__double_as_longlong(w[i]) & ~0xffffull)));
  }
}

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ void generate(double* x, double* y, double* w, int64_t n, int dist) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double u1 = ((mix(3 * i) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((mix(3 * i + 1) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u3 = (mix(3 * i + 2) >> 11) * (1.0 / 9007199254740992.0);
    if (dist == 0) {
      const double r = sqrt(-2.0 * log(u1));
      x[i] = r * cospi(2.0 * u2);
      y[i] = r * sinpi(2.0 * u2);
    } else {
      x[i] = -4.0 + 8.0 * u1;
      y[i] = -4.0 + 8.0 * u2;
    }
    w[i] = u3;
  }
}

template <int MODE>
static float launch(const double* x, const double* y, const double* w, int64_t n, Ctl* ctl, uint64_t* rings, double* part_out,
                    double* out, int row0, uint32_t* status, hipEvent_t e0, hipEvent_t e1) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xchg<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CK(hipMemsetAsync(ctl, 0, sizeof(Ctl) * NX));
  for (int xc = 0; xc < NX; ++xc) CK(hipMemsetAsync(reinterpret_cast<char*>(ctl + xc) + offsetof(Ctl, fin), 0xff, sizeof(uint32_t) * NS * NS));
  CK(hipMemsetAsync(rings, 0, (size_t)NX * NS * NS * CAP * 8));
  CK(hipMemsetAsync(out, 0, (size_t)NBX * NBY * 8));
  CK(hipMemsetAsync(status, 0, 256));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(xchg<MODE>, dim3(NX * NS), dim3(BLOCK), LDS_BYTES, 0, x, y, w, n, ctl, rings, part_out, out, row0, status,
                     (long long)200000000);  // 2 s of the 100 MHz clock
  if (MODE == 0) hipLaunchKernelGGL(merge, dim3((WIN_ROWS * NBY + 255) / 256), dim3(256), 0, 0, part_out, out, row0);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char** argv) {
  const int dist = argc > 1 ? atoi(argv[1]) : 0;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const int64_t n = (int64_t)TILE * (argc > 4 ? atol(argv[4]) : 122070);  // ~5e8
  const int row0 = (NBX - WIN_ROWS) / 2;
  double *x, *y, *w, *out, *ref, *part_out;
  uint64_t* rings;
  Ctl* ctl;
  uint32_t* status;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&w, n * 8));
  CK(hipMalloc(&out, (size_t)NBX * NBY * 8)); CK(hipMalloc(&ref, (size_t)NBX * NBY * 8));
  CK(hipMalloc(&part_out, (size_t)NX * NS * PART_BINS * 8));
  CK(hipMalloc(&rings, (size_t)NX * NS * NS * CAP * 8));
  CK(hipMalloc(&ctl, sizeof(Ctl) * NX));
  CK(hipMalloc(&status, 256));
  hipLaunchKernelGGL(generate, dim3(4096), dim3(256), 0, 0, x, y, w, n, dist);
  CK(hipMemset(ref, 0, (size_t)NBX * NBY * 8));
  hipLaunchKernelGGL(reference, dim3(4096), dim3(256), 0, 0, x, y, w, n, ref);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  uint32_t st[64];
  for (int r = 0; r < reps; ++r) {
    float t = mode == 0 ? launch<0>(x, y, w, n, ctl, rings, part_out, out, row0, status, e0, e1)
            : mode == 1 ? launch<1>(x, y, w, n, ctl, rings, part_out, out, row0, status, e0, e1)
                        : launch<2>(x, y, w, n, ctl, rings, part_out, out, row0, status, e0, e1);
    CK(hipMemcpy(st, status, 256, hipMemcpyDeviceToHost));
    printf("{\"case\": \"xchg\", \"dist\": %d, \"mode\": %d, \"rep\": %d, \"ms\": %.4f, \"status\": %u, \"stalls\": %u, \"wgs_done\": %u, \"taken\": %u, \"phases(digit,issue+rank,after_stall,scatter,writeout,B1,S2,B2)\": {\"w0\": [%u,%u,%u,%u,%u,%u,%u,%u], \"w1\": [%u,%u,%u,%u,%u,%u,%u,%u], \"w15\": [%u,%u,%u,%u,%u,%u,%u,%u]}}\n",
           dist, mode, r, t, st[0], st[1], st[2], st[3], st[16], st[17], st[18], st[19], st[20], st[21], st[22], st[23], st[24], st[25], st[26], st[27], st[28], st[29], st[30], st[31], st[32], st[33], st[34], st[35], st[36], st[37], st[38], st[39]);
    fflush(stdout);
    if (st[0] != 0) break;
    if (r > 0) ms.push_back(t);
  }
  std::vector<Ctl> hc(NX);
  CK(hipMemcpy(hc.data(), ctl, sizeof(Ctl) * NX, hipMemcpyDeviceToHost));
  printf("{\"case\": \"placement\", \"per_xcd\": [%u, %u, %u, %u, %u, %u, %u, %u]}\n", hc[0].nreg, hc[1].nreg, hc[2].nreg, hc[3].nreg,
         hc[4].nreg, hc[5].nreg, hc[6].nreg, hc[7].nreg);
  if (mode != 2 && st[0] == 0) {
    std::vector<double> ho((size_t)NBX * NBY), hr((size_t)NBX * NBY);
    CK(hipMemcpy(ho.data(), out, ho.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, hr.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0, so = 0.0, sr = 0.0;
    size_t bad = 0;
    for (size_t i = 0; i < ho.size(); ++i) {
      so += ho[i]; sr += hr[i];
      const double dlt = fabs(ho[i] - hr[i]), rel = dlt / (fabs(hr[i]) > 1e-300 ? fabs(hr[i]) : 1.0);
      if (rel > worst) worst = rel;
      if (rel > 1e-9) ++bad;
    }
    printf("{\"case\": \"check\", \"sum_out\": %.9e, \"sum_ref\": %.9e, \"worst_rel\": %.3e, \"bins_off\": %zu}\n", so, sr, worst, bad);
  }
  if (!ms.empty()) {
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    printf("{\"case\": \"summary\", \"dist\": %d, \"mode\": %d, \"n\": %lld, \"ms_median\": %.4f, \"ms_min\": %.4f, \"frac_of_8TBs_at_24B\": %.4f}\n", dist, mode,
           (long long)n, med, ms[0], (double)n * 24.0 / (med * 1e-3) / 8e12);
  }
  return 0;
}
