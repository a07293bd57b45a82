/* oracle_c.c — plain-C restatement of the xhistogram hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Second, numpy-free statement of the contract the HIP kernels are checked against (the first
 * is oracle/oracle_np.py, which is pinned to golden vectors produced by the reference itself;
 * tests/test_oracle_c.py pins THIS file to the same vectors).  Only tests/, smoke() and the
 * cpu_baseline leg of bench.py may load it; nothing under xhistogram_amd/ does.
 *
 * Reference lines followed (/root/reference/xhistogram/core.py):
 *   upper_bound()      numpy searchsorted(edges, x, side="right")            core.py:170
 *   bin_of()           right-edge fix-up + the codes table of core.py:157-162  core.py:171-173
 *   xh_oracle_rows()   joint index (ravel_multi_index, C order, first input slowest) core.py:178-181,
 *                      per-row bincount with float64 weights core.py:73-83, trim of the
 *                      under/overflow/NaN bins core.py:189-192
 * Comparisons are in double (numpy promotes f32/int samples to f64 against f64 edges).
 */
#include <stddef.h>
#include <stdint.h>

/* number of edges <= x; NaN sorts after everything (numpy's searchsorted ordering) */
static int64_t upper_bound(const double* e, int64_t n, double x) {
  if (x != x) return n;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (e[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* real-bin index in [0, n-1) or -1 if the sample is dropped */
static int64_t bin_of(const double* e, int64_t n, double x) {
  int64_t code = upper_bound(e, n, x);
  if (x == e[n - 1]) code -= 1;          /* last bin is right-inclusive */
  if (code <= 0 || code >= n) return -1; /* underflow / overflow / NaN   */
  return code - 1;
}

/* samples: n_dims pointers to [rows, cols] row-major doubles; edges: n_dims pointers;
 * weights NULL or [rows, cols]; out_counts (unweighted) or out_sums (weighted): [rows, prod(nb)]
 * zero-initialised by the caller. */
int xh_oracle_rows(int n_dims, const double* const* samples, const double* const* edges, const int64_t* n_edges,
                   const double* weights, int64_t rows, int64_t cols, int64_t* out_counts, double* out_sums) {
  int64_t n_bins = 1;
  for (int d = 0; d < n_dims; ++d) n_bins *= (n_edges[d] - 1);
  for (int64_t r = 0; r < rows; ++r) {
    for (int64_t c = 0; c < cols; ++c) {
      int64_t flat = 0;
      int ok = 1;
      for (int d = 0; d < n_dims && ok; ++d) {
        int64_t b = bin_of(edges[d], n_edges[d], samples[d][r * cols + c]);
        if (b < 0) ok = 0;
        flat = flat * (n_edges[d] - 1) + b;
      }
      if (!ok) continue;
      if (weights) out_sums[r * n_bins + flat] += weights[r * cols + c];
      else out_counts[r * n_bins + flat] += 1;
    }
  }
  return 0;
}
