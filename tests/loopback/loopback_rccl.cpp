// loopback_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the five RCCL entry points libppcx.so binds with dlopen
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclGetErrorString), built over POSIX shared
// memory between the processes of ONE host. RCCL refuses two ranks on one device, so on a one-GPU box the gene-shard
// path over several ranks (ppcx_fit_nuts_comm: per-round all-reduce, rank-divergence guard) could never execute; with
// PPCX_RCCL_LIB pointing here it does. Semantics kept: the all-reduce is ordered on the given stream (it synchronises
// it), every rank receives the same bits (slots are reduced in rank order), sum and max over doubles. Nothing else of
// NCCL is implemented, and the product never loads this file by itself.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kSlotDoubles = 16384;          // the library reduces chains x 76 sums, and 5 for the guard
struct Shared {
  volatile long long arrived;                   // arrivals at barriers, ever: barrier k is complete when it reaches nranks (k + 1)
  volatile int attached;
  double slot[kMaxRanks][kSlotDoubles];
};
struct Comm { Shared* sh; int nranks, rank; long long phase; char name[64]; double* host; };
const double kTimeoutSeconds = 60.0;            // a peer that never arrives: report instead of hanging the box

double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
// all ranks reach the barrier. One counter that only grows: nothing is ever reset, so a rank that runs ahead into the next
// barrier cannot have its arrival wiped by a slower rank's reset (two alternating counters reset by the last arriver had
// exactly that race: once in a few hundred runs both ranks then waited for the timeout).
int barrier(Comm* c) {
  Shared* s = c->sh;
  const long long target = (long long)c->nranks * ++c->phase;
  __sync_add_and_fetch(&s->arrived, 1LL);
  const double t0 = now();
  while (s->arrived < target) {
    sched_yield();
    if (now() - t0 > kTimeoutSeconds) return 1;
  }
  return 0;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId_t;

__attribute__((visibility("default"))) int ncclGetUniqueId(ncclUniqueId_t* id) {
  memset(id->internal, 0, sizeof id->internal);
  snprintf(id->internal, sizeof id->internal, "/ppcx_loopback_%d_%ld", (int)getpid(), (long)(now() * 1e6));
  return 0;
}
__attribute__((visibility("default"))) int ncclCommInitRank(void** out, int nranks, ncclUniqueId_t id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;   // ncclInvalidArgument
  Comm* c = new Comm();
  c->nranks = nranks; c->rank = rank; c->phase = 0;
  strncpy(c->name, id.internal, sizeof c->name - 1);
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return 2; }            // ncclSystemError
  if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); delete c; return 2; }
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return 2; }
  c->sh = (Shared*)p;                            // a fresh segment is zero-filled
  c->host = nullptr;
  if (hipHostMalloc(&c->host, sizeof(double) * kSlotDoubles) != hipSuccess) { munmap(p, sizeof(Shared)); delete c; return 1; }
  __sync_add_and_fetch(&c->sh->attached, 1);
  const double t0 = now();
  while (c->sh->attached < nranks) { sched_yield(); if (now() - t0 > kTimeoutSeconds) return 2; }
  *out = c;
  return 0;
}
__attribute__((visibility("default"))) int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  if (__sync_sub_and_fetch(&c->sh->attached, 1) == 0) shm_unlink(c->name);
  munmap((void*)c->sh, sizeof(Shared));
  if (c->host) (void)hipHostFree(c->host);
  delete c;
  return 0;
}
// datatype 8 = ncclDouble; op 0 = ncclSum, 2 = ncclMax (the only ones the library uses)
__attribute__((visibility("default"))) int ncclAllReduce(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t st) {
  Comm* c = (Comm*)comm;
  if (!c || datatype != 8 || (op != 0 && op != 2) || count > kSlotDoubles) return 4;
  if (hipStreamSynchronize(st) != hipSuccess) return 1;                       // ncclUnhandledCudaError
  if (hipMemcpy(c->host, send, sizeof(double) * count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  memcpy((void*)c->sh->slot[c->rank], c->host, sizeof(double) * count);
  __sync_synchronize();
  if (barrier(c)) return 6;                                                    // ncclRemoteError: a peer never arrived
  for (size_t i = 0; i < count; ++i) {
    double r = c->sh->slot[0][i];
    for (int k = 1; k < c->nranks; ++k) { const double x = c->sh->slot[k][i]; r = op == 0 ? r + x : (x > r ? x : r); }
    c->host[i] = r;
  }
  if (barrier(c)) return 6;                                                    // nobody overwrites a slot a peer still reads
  if (hipMemcpy(recv, c->host, sizeof(double) * count, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}
__attribute__((visibility("default"))) const char* ncclGetErrorString(int e) {
  switch (e) { case 0: return "success"; case 1: return "HIP error in the loopback collective"; case 2: return "shared memory error";
               case 4: return "invalid argument"; case 6: return "a peer rank did not arrive (timeout)"; default: return "error"; }
}
}
