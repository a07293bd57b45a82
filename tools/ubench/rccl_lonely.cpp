// rccl_lonely.cpp — development probe (not part of the product): what does RCCL do when rank 0 of a world of 2 is alone?
//   hipcc -O2 -o rccl_lonely rccl_lonely.cpp -lrccl -lpthread     ./rccl_lonely [seconds before abort] [mode]
//   mode 0: non-blocking init on the calling thread (RCCL 2.27.7 of ROCm 7.2: the call itself never returns)
//   mode 1: init on a helper thread, ncclCommAbort from the waiting thread at the deadline
//   mode 2: as 1 with a BLOCKING communicator
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const double limit = argc > 1 ? atof(argv[1]) : 4.0;
  const int mode = argc > 2 ? atoi(argv[2]) : 1;
  (void)hipSetDevice(0);
  ncclUniqueId id;
  printf("getUniqueId -> %d\n", (int)ncclGetUniqueId(&id)); fflush(stdout);
  static ncclComm_t comm = nullptr;
  static std::atomic<int> done{0};
  static ncclResult_t result = ncclSuccess;
  double t0 = now();
  std::thread th([&] {
    (void)hipSetDevice(0);
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.blocking = mode == 2 ? 1 : 0;
    result = ncclCommInitRankConfig(&comm, 2, id, 0, &cfg);
    done = 1;
  });
  if (mode == 0) th.join();
  while (!done && now() - t0 < limit) std::this_thread::sleep_for(std::chrono::milliseconds(10));
  printf("after %.3f s: done %d result %d comm %p\n", now() - t0, (int)done, (int)result, (void*)*(ncclComm_t volatile*)&comm); fflush(stdout);
  if (!done && *(ncclComm_t volatile*)&comm) {
    ncclResult_t st = ncclSuccess;
    ncclResult_t q = ncclCommGetAsyncError(comm, &st);
    printf("async error query -> %d, state %d (%s)\n", (int)q, (int)st, ncclGetErrorString(st)); fflush(stdout);
    double t1 = now();
    ncclResult_t r = ncclCommAbort(comm);
    printf("Abort -> %d (%s) after %.3f s\n", (int)r, ncclGetErrorString(r), now() - t1); fflush(stdout);
  }
  double t2 = now();
  while (!done && now() - t2 < 10) std::this_thread::sleep_for(std::chrono::milliseconds(10));
  printf("helper thread: done %d result %d (%s) %.3f s after the abort\n", (int)done, (int)result, ncclGetErrorString(result), now() - t2); fflush(stdout);
  if (done) th.join(); else th.detach();
  printf("exiting\n"); fflush(stdout);
  return 0;
}
