/* capi_client.c — a plain C client of libxhist_amd.so: the C ABI needs no Python and no torch.
 * Build:  gcc -O2 -I include tests/capi_client.c -o capi_client -L xhistogram_amd -lxhist_amd -Wl,-rpath,xhistogram_amd -lm
 * Exit code 0: the GPU result equals a scalar CPU loop written here (bin rule of core.py:157-174);
 * exit code 77: no GPU (the library refused loudly, which is the contract). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xhist_amd.h"

static int bin_of(const double* e, int n, double x) {
  if (x != x || x < e[0] || x > e[n - 1]) return -1;
  int k = 0;
  while (k + 1 < n && e[k + 1] <= x) ++k;
  return k < n - 1 ? k : n - 2;
}

int main(void) {
  enum { ROWS = 3, COLS = 100003, NB = 37 };
  int ndev = 0;
  if (xhist_abi_version() != XHIST_ABI_VERSION) return 2;
  xhist_device_count(&ndev);
  double edges[NB + 1];
  for (int i = 0; i <= NB; ++i) edges[i] = -3.0 + 6.0 * i / NB;
  const void* eptr[1] = {edges};
  int64_t elen[1] = {NB + 1};
  xhist_plan* plan = NULL;
  int rc = xhist_plan_create(0, 1, eptr, elen, XHIST_CMP_F64, &plan);
  if (ndev == 0) {
    if (rc != XHIST_ERR_NO_DEVICE) return 3;
    printf("no GPU: %s\n", xhist_last_error());
    return 77;
  }
  if (rc) { printf("plan: %s\n", xhist_last_error()); return 4; }

  float* x = malloc(sizeof(float) * ROWS * COLS);
  double* w = malloc(sizeof(double) * COLS);  /* one weight row, broadcast over the rows */
  uint64_t s = 88172645463325252ull;
  for (long i = 0; i < (long)ROWS * COLS; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    x[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 8.0 - 4.0);
  }
  x[5] = NAN; x[6] = 3.0f; x[7] = -3.0f;
  for (int c = 0; c < COLS; ++c) w[c] = 0.25 + (c % 7);

  xhist_array xa = {x, XHIST_F32, 0, COLS, 1, 0, 0};
  xhist_array wa = {w, XHIST_F64, 0, 0, 1, 0, 0}; /* row_stride 0 = broadcast */
  int64_t counts[ROWS * NB];
  double sums[ROWS * NB];
  rc = xhist_plan_execute(plan, &xa, NULL, ROWS, COLS, counts, XHIST_I64, XHIST_MEM_HOST, 0, NULL);
  if (rc) { printf("execute: %s\n", xhist_last_error()); return 5; }
  rc = xhist_bincount_rows(0, 1, &xa, &wa, ROWS, COLS, eptr, elen, XHIST_CMP_F64, sums, XHIST_F64, XHIST_MEM_HOST, 0, NULL);
  if (rc) { printf("one-shot: %s\n", xhist_last_error()); return 6; }

  int bad = 0;
  for (int r = 0; r < ROWS; ++r) {
    int64_t ref[NB] = {0};
    double refw[NB] = {0};
    for (int c = 0; c < COLS; ++c) {
      int b = bin_of(edges, NB + 1, (double)x[r * COLS + c]);
      if (b >= 0) { ref[b] += 1; refw[b] += w[c]; }
    }
    for (int b = 0; b < NB; ++b) {
      if (ref[b] != counts[r * NB + b]) ++bad;
      if (fabs(refw[b] - sums[r * NB + b]) > 1e-6 * fabs(refw[b])) ++bad;
    }
  }
  char desc[512];
  xhist_plan_describe(plan, desc, sizeof desc);
  double mm[2];
  rc = xhist_minmax(0, &xa, 1, COLS, mm, XHIST_MEM_HOST, NULL); /* row 0 holds the NaN */
  if (rc || mm[0] == mm[0]) ++bad;                              /* numpy: min of data with NaN is NaN */
  /* a block that lives on the GPU: uploaded once, described by strides, histogrammed where it lies (XHIST_MEM_DEVICE, the
   * result into a device buffer too); and the library's strided copy — the (ROWS, COLS) float32 block transposed and
   * converted to float64, as the block adapter does for layouts no three strides describe (core.py:218-226) */
  {
    void *dx = NULL, *dt = NULL, *dout = NULL;
    int dev = -1;
    if (xhist_buffer_alloc(0, sizeof(float) * ROWS * COLS, &dx) || xhist_buffer_alloc(0, sizeof(double) * ROWS * COLS, &dt) ||
        xhist_buffer_alloc(0, sizeof(int64_t) * ROWS * NB, &dout))
      ++bad;
    if (xhist_buffer_copy(0, dx, x, sizeof(float) * ROWS * COLS, 0, NULL)) ++bad;
    if (xhist_pointer_device(dx, &dev) || dev != 0) ++bad;
    xhist_array da = {dx, XHIST_F32, 0, COLS, 1, 0, 0};
    if (xhist_plan_execute(plan, &da, NULL, ROWS, COLS, dout, XHIST_I64, XHIST_MEM_DEVICE, 0, NULL)) ++bad;
    int64_t* counts_d = (int64_t*)malloc(sizeof(int64_t) * ROWS * NB);
    if (xhist_buffer_copy(0, counts_d, dout, sizeof(int64_t) * ROWS * NB, 1, NULL)) ++bad;
    for (int i = 0; i < ROWS * NB; ++i)
      if (counts_d[i] != counts[i]) ++bad;
    const int64_t shape[2] = {COLS, ROWS};
    const int64_t sstr[2] = {sizeof(float), sizeof(float) * COLS};    /* the transpose of the source */
    const int64_t dstr[2] = {sizeof(double) * ROWS, sizeof(double)};  /* C-contiguous (COLS, ROWS) */
    if (xhist_buffer_copy_nd(0, 2, shape, dx, XHIST_F32, sstr, dt, XHIST_F64, dstr, NULL)) ++bad;
    double* tr = (double*)malloc(sizeof(double) * ROWS * COLS);
    if (xhist_buffer_copy(0, tr, dt, sizeof(double) * ROWS * COLS, 1, NULL)) ++bad;
    for (int r = 0; r < ROWS; ++r)
      for (int c = 0; c < COLS; c += 97) {
        const double want = (double)x[r * COLS + c], got = tr[(size_t)c * ROWS + r];
        if (!(want == got || (want != want && got != got))) ++bad;
      }
    free(counts_d); free(tr);
    xhist_buffer_free(0, dx); xhist_buffer_free(0, dt); xhist_buffer_free(0, dout);
  }
  xhist_plan_destroy(plan);
  xhist_shutdown();
  printf("%s: %d mismatches; last launch: %s\n", bad ? "FAIL" : "OK", bad, desc);
  free(x); free(w);
  return bad ? 1 : 0;
}
