/* capi_comm_client.c — the multi-GPU exchange of the C ABI from plain C: one process per GPU, no Python,
 * no torch, no MPI.  The parent forks one child per visible GPU before anything touches HIP; rank 0
 * asks for the communicator id and publishes it through a file (any out-of-band channel will do),
 * every rank histograms ITS shard of the samples into device memory and one xhist_comm_allreduce
 * makes the full histogram appear on every GPU (the reference's bin_counts.sum(drop_axes), core.py:439).
 * Build:  gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/capi_comm_client.c -o capi_comm_client \
 *             -L xhistogram_amd -lxhist_amd -Wl,-rpath,xhistogram_amd -L /opt/rocm/lib -lamdhip64 -lm
 * Run:    ./capi_comm_client [world_size]      (default: the number of GPUs; exit 77 without a GPU)
 *         ./capi_comm_client lonely            rank 0 of a world of 2 whose peer never joins: xhist_comm_create must come
 *                                              back with XHIST_ERR_COMM inside the deadline (XHIST_AMD_COMM_TIMEOUT_S), not hang */
#include <time.h>
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include "xhist_amd.h"

#define NB 64
#define TOTAL 1000003L /* samples in the whole data set; rank r owns a contiguous share */

static double sample(long i) { /* deterministic, so every rank can recompute the whole-data answer */
  uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
  s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
  return (double)(s >> 11) / 9007199254740992.0 * 9.0 - 4.5; /* some land outside [-4, 4] */
}

static int rank_main(int rank, int world, const char* id_path) {
  char id[XHIST_COMM_ID_BYTES];
  char tmp[600];
  if (rank == 0) {
    if (xhist_comm_unique_id(id, sizeof id)) { printf("rank 0 id: %s\n", xhist_last_error()); return 3; }
    snprintf(tmp, sizeof tmp, "%s.tmp", id_path);
    FILE* f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) || rename(tmp, id_path)) return 3;
  } else {
    FILE* f = NULL;
    for (int tries = 0; tries < 6000 && !(f = fopen(id_path, "rb")); ++tries) usleep(10000);
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 3;
    fclose(f);
  }
  xhist_comm* comm = NULL;
  if (xhist_comm_create(rank, rank, world, id, sizeof id, &comm)) { printf("rank %d create: %s\n", rank, xhist_last_error()); return 4; }
  int r2 = -1, w2 = -1, dev = -1, ver = 0;
  if (xhist_comm_info(comm, &r2, &w2, &dev, &ver) || r2 != rank || w2 != world || dev != rank || ver <= 0) return 4;

  double edges[NB + 1];
  for (int i = 0; i <= NB; ++i) edges[i] = -4.0 + 8.0 * i / NB;
  const void* eptr[1] = {edges};
  int64_t elen[1] = {NB + 1};
  xhist_plan* plan = NULL;
  if (xhist_plan_create(rank, 1, eptr, elen, XHIST_CMP_F64, &plan)) { printf("plan: %s\n", xhist_last_error()); return 5; }

  const long base = TOTAL / world, extra = TOTAL % world;
  const long lo = rank * base + (rank < extra ? rank : extra), n = base + (rank < extra ? 1 : 0);
  double* x = (double*)malloc(sizeof(double) * (size_t)n);
  for (long i = 0; i < n; ++i) x[i] = sample(lo + i);
  if (hipSetDevice(rank) != hipSuccess) return 6;
  double* dx = NULL;
  int64_t* dcounts = NULL;
  double *dmm = NULL, *dall = NULL;
  if (hipMalloc((void**)&dx, sizeof(double) * (size_t)n) != hipSuccess || hipMalloc((void**)&dcounts, sizeof(int64_t) * NB) != hipSuccess ||
      hipMalloc((void**)&dmm, sizeof(double) * 2) != hipSuccess || hipMalloc((void**)&dall, sizeof(double) * 2 * (size_t)world) != hipSuccess)
    return 6;
  if (hipMemcpy(dx, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return 6;

  /* this rank's partial, then ONE in-place all-reduce: both asynchronous on the same (default) stream */
  xhist_array xa = {dx, XHIST_F64, 0, n, 1, 0, 0};
  if (xhist_plan_execute(plan, &xa, NULL, 1, n, dcounts, XHIST_I64, XHIST_MEM_DEVICE, 0, NULL)) { printf("execute: %s\n", xhist_last_error()); return 7; }
  if (xhist_comm_allreduce(comm, dcounts, NB, XHIST_I64, XHIST_REDUCE_SUM, NULL)) { printf("allreduce: %s\n", xhist_last_error()); return 8; }
  if (xhist_comm_wait(comm, NULL)) { printf("wait: %s\n", xhist_last_error()); return 8; }  /* completion with a deadline, not a bare sync */
  int64_t counts[NB];
  if (hipMemcpy(counts, dcounts, sizeof counts, hipMemcpyDeviceToHost) != hipSuccess) return 6;

  /* global minimum of the data (what bins=int needs), and an all-gather of every rank's (min, max) */
  double mm[2];
  if (xhist_minmax(rank, &xa, 1, n, mm, XHIST_MEM_DEVICE, NULL)) return 9;
  double neg[2] = {mm[0], -mm[1]};
  if (hipMemcpy(dmm, neg, sizeof neg, hipMemcpyHostToDevice) != hipSuccess) return 6;
  if (xhist_comm_allgather(comm, dmm, dall, 2, XHIST_F64, NULL)) { printf("allgather: %s\n", xhist_last_error()); return 8; }
  if (xhist_comm_allreduce(comm, dmm, 2, XHIST_F64, XHIST_REDUCE_MIN, NULL)) return 8;
  double gmm[2];
  double* all = (double*)malloc(sizeof(double) * 2 * (size_t)world);
  if (hipMemcpy(gmm, dmm, sizeof gmm, hipMemcpyDeviceToHost) != hipSuccess) return 6;
  if (hipMemcpy(all, dall, sizeof(double) * 2 * (size_t)world, hipMemcpyDeviceToHost) != hipSuccess) return 6;

  /* the whole-data answer, recomputed by a scalar loop on every rank */
  int bad = 0;
  int64_t ref[NB] = {0};
  double rmin = INFINITY, rmax = -INFINITY;
  for (long i = 0; i < TOTAL; ++i) {
    const double v = sample(i);
    rmin = fmin(rmin, v); rmax = fmax(rmax, v);
    if (v < edges[0] || v > edges[NB]) continue;
    int b = (int)((v - edges[0]) / (8.0 / NB));
    if (b > NB - 1) b = NB - 1;
    while (b > 0 && v < edges[b]) --b;                 /* settle on the exact edge values */
    while (b < NB - 1 && v >= edges[b + 1]) ++b;
    ++ref[b];
  }
  for (int b = 0; b < NB; ++b) bad += counts[b] != ref[b];
  bad += gmm[0] != rmin || -gmm[1] != rmax;
  bad += all[2 * rank] != mm[0] || all[2 * rank + 1] != -mm[1];
  double amin = INFINITY;
  for (int r = 0; r < world; ++r) amin = fmin(amin, all[2 * r]);
  bad += amin != rmin;

  xhist_plan_destroy(plan);
  if (xhist_comm_destroy(comm)) { printf("destroy: %s\n", xhist_last_error()); return 10; }
  hipFree(dx); hipFree(dcounts); hipFree(dmm); hipFree(dall);
  free(x); free(all);
  xhist_shutdown();
  printf("rank %d of %d: %s (%d mismatches, rccl %d)\n", rank, world, bad ? "FAIL" : "OK", bad, ver);
  return bad ? 1 : 0;
}

/* rank 0 of a world of two, alone: the rendezvous must end in a status code */
static int lonely_main(void) {
  char id[XHIST_COMM_ID_BYTES];
  if (xhist_comm_unique_id(id, sizeof id)) { printf("id: %s\n", xhist_last_error()); return 3; }
  xhist_comm* comm = NULL;
  const time_t t0 = time(NULL);
  const int rc = xhist_comm_create(0, 0, 2, id, sizeof id, &comm);
  const long took = (long)(time(NULL) - t0);
  printf("xhist_comm_create alone in a world of 2: rc %d after %ld s: %s\n", rc, took, rc ? xhist_last_error() : "(no error)");
  if (rc != XHIST_ERR_COMM || comm != NULL) return 1;
  xhist_shutdown();
  printf("OK: lonely rank got XHIST_ERR_COMM\n");
  return 0;
}

int main(int argc, char** argv) {
  if (xhist_abi_version() != XHIST_ABI_VERSION) return 2;
  if (argc > 1 && !strcmp(argv[1], "lonely")) {
    int n = 0;
    xhist_device_count(&n);
    if (n <= 0) { printf("no GPU: this library has no CPU path\n"); return 77; }
    return lonely_main();
  }
  /* the device count comes from a child: the parent must not initialise HIP before it forks */
  int fds[2];
  if (pipe(fds)) return 2;
  pid_t probe = fork();
  if (probe == 0) {
    int n = 0;
    xhist_device_count(&n);
    if (write(fds[1], &n, sizeof n) != (ssize_t)sizeof n) _exit(1);
    _exit(0);
  }
  int ndev = 0, st = 0;
  if (read(fds[0], &ndev, sizeof ndev) != (ssize_t)sizeof ndev) ndev = 0;
  waitpid(probe, &st, 0);
  if (ndev <= 0) { printf("no GPU: this library has no CPU path\n"); return 77; }
  int world = argc > 1 ? atoi(argv[1]) : ndev;
  if (world < 1 || world > ndev) { printf("world_size %d with %d GPUs\n", world, ndev); return 2; }

  char id_path[512];
  snprintf(id_path, sizeof id_path, "%s/xhist_comm_id_%ld", getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp", (long)getpid());
  remove(id_path);
  pid_t pids[64];
  for (int r = 0; r < world && r < 64; ++r) {
    pids[r] = fork();
    if (pids[r] == 0) {
      fflush(stdout);
      int rc = rank_main(r, world, id_path);
      fflush(stdout);
      _exit(rc);
    }
  }
  int worst = 0;
  for (int r = 0; r < world && r < 64; ++r) {
    int s = 0;
    waitpid(pids[r], &s, 0);
    const int rc = WIFEXITED(s) ? WEXITSTATUS(s) : 99;
    if (rc > worst) worst = rc;
  }
  remove(id_path);
  printf("%s: %d rank(s)\n", worst ? "FAIL" : "OK", world);
  return worst;
}
