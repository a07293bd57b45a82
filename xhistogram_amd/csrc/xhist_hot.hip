// xhist_hot.hip — the kernels a FIRST call most likely needs, in a translation unit (= a code object) small enough to load in a
// fraction of a millisecond: output zeroing, the table builders of plan creation, and the vector kernels for one or two float32 /
// float64 inputs with the histogram in LDS on uniform-style edges (one edge per bucket, the arithmetic digitize, the float32
// arithmetic digitize) — BASELINE C1 / C2 / C4, `bins=int`, np.linspace.  See is_hot_kernel in xhist_pick.hip.h: the big
// translation units do not instantiate these, and nothing of theirs is loaded until a call needs it.
#include "xhist_pick.hip.h"

template <typename ST, typename WT, int D, int SCAN>
static kernel_fn hot_one() {
  constexpr bool unweighted = std::is_same<WT, NoWeight>::value;
  constexpr int wsz = unweighted ? 0 : (int)sizeof(typename std::conditional<unweighted, float, WT>::type);
  constexpr int VEC = 16 / (((int)sizeof(ST) > wsz) ? (int)sizeof(ST) : wsz);
  constexpr int U = unroll_for(D, VEC, SCAN);
  return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistLds, SCAN>;
}

template <typename ST, typename WT, int D>
static kernel_fn hot_scan(int scan) {
  if (scan == 1) return hot_one<ST, WT, D, 1>();
  if (scan == kScanArith) return hot_one<ST, WT, D, kScanArith>();
  if constexpr (std::is_same<ST, float>::value) {
    if (scan == kScanArith32) return hot_one<ST, WT, D, kScanArith32>();
  }
  return nullptr;
}

template <typename ST>
static kernel_fn hot_w(int wdt, int D, int scan) {
  if (D == 1) {
    if (wdt == -1) return hot_scan<ST, NoWeight, 1>(scan);
    if (wdt == XHIST_F64) return hot_scan<ST, double, 1>(scan);
    if (wdt == XHIST_F32) return hot_scan<ST, float, 1>(scan);
  } else if (D == 2) {
    if (wdt == -1) return hot_scan<ST, NoWeight, 2>(scan);
    if (wdt == XHIST_F64) return hot_scan<ST, double, 2>(scan);
    if (wdt == XHIST_F32) return hot_scan<ST, float, 2>(scan);
  }
  return nullptr;
}

kernel_fn xhist_pick_hot(int sdt, int wdt, int D, int scan, int hist) {
  if (!is_hot_kernel(D, scan, hist)) return nullptr;
  if (sdt == XHIST_F64) return hot_w<double>(wdt, D, scan);
  if (sdt == XHIST_F32) return hot_w<float>(wdt, D, scan);
  return nullptr;
}

// one unweighted float input, LDS histogram, tiles twice as long (xhist_pick_f64_long / _f32_long in the big units hand these out)
kernel_fn xhist_pick_hot_long(int sdt, int scan) {
  if (sdt == XHIST_F64 && scan == 1) return (kernel_fn)hist_fast<double, NoWeight, 1, 2, 8, kHistLds, 1>;
  if (sdt == XHIST_F32 && scan == 1) return (kernel_fn)hist_fast<float, NoWeight, 1, 4, 8, kHistLds, 1>;
  if (sdt == XHIST_F32 && scan == kScanArith32) return (kernel_fn)hist_fast<float, NoWeight, 1, 4, 8, kHistLds, kScanArith32>;
  return nullptr;
}

int xhist_hot_zero_words(unsigned long long* p, int64_t n, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(zero_words, dim3((unsigned)grid), dim3(256), 0, stream, p, n);
  return (int)hipGetLastError();
}

int xhist_hot_build_tables(int dom, bool lut16, const DimTable& t, uint64_t* blob, int32_t* scratch) {
  if (dom == 0 && !lut16) hipLaunchKernelGGL((build_tables<0, false>), dim3(1), dim3(256), 0, 0, t, blob, scratch);
  else if (dom == 0) hipLaunchKernelGGL((build_tables<0, true>), dim3(1), dim3(256), 0, 0, t, blob, scratch);
  else if (dom == 1) hipLaunchKernelGGL((build_tables<1, false>), dim3(1), dim3(256), 0, 0, t, blob, scratch);
  else if (!lut16) hipLaunchKernelGGL((build_tables<2, false>), dim3(1), dim3(256), 0, 0, t, blob, scratch);
  else hipLaunchKernelGGL((build_tables<2, true>), dim3(1), dim3(256), 0, 0, t, blob, scratch);
  return (int)hipGetLastError();
}

int xhist_hot_build_pack_tables(const DimTable& t, uint64_t* blob, int32_t* scratch, const float* thr) {
  hipLaunchKernelGGL(build_pack_tables, dim3(1), dim3(256), 0, 0, t, blob, scratch, thr);
  return (int)hipGetLastError();
}

int xhist_hot_minmax_flat(bool f64, const void* x, int64_t n, double* partial, int grid, hipStream_t stream) {
  if (f64) hipLaunchKernelGGL(minmax_flat<double>, dim3((unsigned)grid), dim3(256), 0, stream, static_cast<const double*>(x), n, partial);
  else hipLaunchKernelGGL(minmax_flat<float>, dim3((unsigned)grid), dim3(256), 0, stream, static_cast<const float*>(x), n, partial);
  return (int)hipGetLastError();
}
