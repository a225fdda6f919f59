/* fake_rccl.c -- TEST INFRASTRUCTURE, never shipped and never loaded by the product on its own.
 *
 * A stand-in for librccl.so that lets N processes which SHARE ONE GPU run the collectives of
 * discorpy_amd/csrc/api_rccl.cpp (RCCL itself refuses two ranks on one device, and the test boxes have one GPU).
 * It is loaded only through the DCP_RCCL_PATH hook (api_rccl.cpp: rccl()) by tests/test_rccl_world.py and by
 * bench.py's native children when that variable is set.  It implements exactly the eight symbols the library binds:
 *
 *   ncclGetUniqueId  ncclCommInitRank  ncclCommDestroy  ncclAllGather  ncclBroadcast
 *   ncclGroupStart   ncclGroupEnd      ncclGetErrorString
 *
 * Rendezvous: a POSIX shared-memory object named by the unique id.  Payload: a host bounce slot per rank inside the
 * same object -- the sender copies device -> slot ON THE STREAM IT WAS GIVEN and waits for that stream, the ranks
 * meet at a barrier, the receivers copy slot -> device on their stream, second barrier, next chunk.  Stream order is
 * therefore honoured (a collective queued behind a kernel or an event wait sees that kernel's output, work queued
 * behind the collective sees its result); what is NOT reproduced is RCCL's asynchrony (every call here blocks the
 * host until its exchange is done) and, of course, xGMI.  A rank that never arrives makes the others fail with
 * ncclSystemError after FAKE_RCCL_TIMEOUT_S (default 60) instead of hanging the test box.
 *
 * Calls between ncclGroupStart and ncclGroupEnd are queued and run at ncclGroupEnd in call order, as RCCL's are.
 */
#define _GNU_SOURCE
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define FAKE_MAX_RANKS 16
#define FAKE_MAX_GROUP 64

enum { kSuccess = 0, kUnhandledCuda = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

typedef struct {
  char internal[128];
} ncclUniqueId;

typedef struct {
  atomic_int arrived;     /* barrier: ranks that reached the current generation */
  atomic_int generation;
  atomic_int attached;    /* ranks between CommInitRank and CommDestroy */
  atomic_int failed;      /* set by a rank that gave up: everybody else fails fast */
  atomic_llong calls[4];  /* statistics the tests read back: all-gathers, broadcasts, groups, bytes */
  size_t slot_bytes;
} Header;

typedef struct {
  Header* hdr;
  char* slots;
  size_t map_bytes, slot_bytes;
  int world, rank;
  double timeout_s;
  char name[64];
} Comm;

typedef struct {
  int kind; /* 0 all-gather, 1 broadcast */
  const void* send;
  void* recv;
  size_t bytes;
  int root;
  Comm* comm;
  hipStream_t stream;
} Op;

static __thread int g_depth = 0;
static __thread int g_nops = 0;
static __thread Op g_ops[FAKE_MAX_GROUP];

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static size_t dtype_bytes(int dt) {
  switch (dt) {
    case 0: case 1: return 1;           /* int8 / char, uint8 */
    case 2: case 3: case 7: return 4;   /* int32, uint32, float32 */
    case 4: case 5: case 8: return 8;   /* int64, uint64, float64 */
    case 6: case 9: return 2;           /* float16, bfloat16 */
    default: return 0;
  }
}

static int barrier(Comm* c) {
  Header* h = c->hdr;
  const int gen = atomic_load(&h->generation);
  if (atomic_fetch_add(&h->arrived, 1) + 1 == c->world) {
    atomic_store(&h->arrived, 0);
    atomic_fetch_add(&h->generation, 1);
    return kSuccess;
  }
  const double t_end = now_s() + c->timeout_s;
  unsigned spins = 0;
  while (atomic_load(&h->generation) == gen) {
    if (atomic_load(&h->failed)) return kSystemError;
    if ((++spins & 255u) == 0) {
      if (now_s() > t_end) {
        atomic_store(&h->failed, 1);
        fprintf(stderr, "fake_rccl: rank %d waited %.0f s at a barrier for its %d peers\n", c->rank, c->timeout_s, c->world - 1);
        return kSystemError;
      }
      struct timespec nap = {0, 50000};
      nanosleep(&nap, NULL);
    } else {
      sched_yield();
    }
  }
  return kSuccess;
}

static int run_op(const Op* op) {
  Comm* c = op->comm;
  const size_t slot = c->slot_bytes;
  atomic_fetch_add(&c->hdr->calls[op->kind], 1);
  atomic_fetch_add(&c->hdr->calls[3], (long long)op->bytes);
  for (size_t off = 0; off < op->bytes || (off == 0 && op->bytes == 0); off += slot) {
    const size_t n = op->bytes - off < slot ? op->bytes - off : slot;
    int rc;
    if (op->kind == 0) { /* all-gather: rank r's `bytes` land at recv + r * bytes on every rank */
      if (n && hipMemcpyAsync(c->slots + (size_t)c->rank * slot, (const char*)op->send + off, n, hipMemcpyDeviceToHost, op->stream) != hipSuccess)
        return kUnhandledCuda;
      if (hipStreamSynchronize(op->stream) != hipSuccess) return kUnhandledCuda;
      if ((rc = barrier(c)) != kSuccess) return rc;
      for (int r = 0; r < c->world && n; ++r) {
        char* dst = (char*)op->recv + (size_t)r * op->bytes + off;
        if (r == c->rank) {
          if ((const char*)op->send + off != dst &&
              hipMemcpyAsync(dst, (const char*)op->send + off, n, hipMemcpyDeviceToDevice, op->stream) != hipSuccess)
            return kUnhandledCuda;
        } else if (hipMemcpyAsync(dst, c->slots + (size_t)r * slot, n, hipMemcpyHostToDevice, op->stream) != hipSuccess) {
          return kUnhandledCuda;
        }
      }
    } else { /* broadcast from `root` */
      if (c->rank == op->root && n &&
          hipMemcpyAsync(c->slots + (size_t)op->root * slot, (const char*)op->send + off, n, hipMemcpyDeviceToHost, op->stream) != hipSuccess)
        return kUnhandledCuda;
      if (hipStreamSynchronize(op->stream) != hipSuccess) return kUnhandledCuda;
      if ((rc = barrier(c)) != kSuccess) return rc;
      if (n) {
        if (c->rank != op->root) {
          if (hipMemcpyAsync((char*)op->recv + off, c->slots + (size_t)op->root * slot, n, hipMemcpyHostToDevice, op->stream) != hipSuccess)
            return kUnhandledCuda;
        } else if ((const char*)op->send != (const char*)op->recv &&
                   hipMemcpyAsync((char*)op->recv + off, (const char*)op->send + off, n, hipMemcpyDeviceToDevice, op->stream) != hipSuccess) {
          return kUnhandledCuda;
        }
      }
    }
    if (hipStreamSynchronize(op->stream) != hipSuccess) return kUnhandledCuda;
    if ((rc = barrier(c)) != kSuccess) return rc; /* the slots may be overwritten from here on */
    if (op->bytes == 0) break;
  }
  return kSuccess;
}

static int submit(const Op* op) {
  if (g_depth > 0) {
    if (g_nops >= FAKE_MAX_GROUP) return kInvalidUsage;
    g_ops[g_nops++] = *op;
    return kSuccess;
  }
  return run_op(op);
}

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return kInvalidArgument;
  static atomic_int counter;
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/dcp_fake_rccl_%ld_%llx_%d", (long)getpid(), (unsigned long long)(now_s() * 1e6),
           atomic_fetch_add(&counter, 1));
  return kSuccess;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks) return kInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  if (strncmp(id.internal, "/dcp_fake_rccl_", 15) != 0) return kInvalidArgument;
  Comm* c = (Comm*)calloc(1, sizeof(Comm));
  if (!c) return kSystemError;
  const char* mb = getenv("FAKE_RCCL_SLOT_MB");
  const char* to = getenv("FAKE_RCCL_TIMEOUT_S");
  c->slot_bytes = (size_t)(mb && atoi(mb) > 0 ? atoi(mb) : 4) << 20;
  c->timeout_s = to && atof(to) > 0 ? atof(to) : 60.0;
  c->world = nranks;
  c->rank = rank;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->map_bytes = 4096 + (size_t)nranks * c->slot_bytes;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    fprintf(stderr, "fake_rccl: shm_open/ftruncate(%s): %s\n", c->name, strerror(errno));
    if (fd >= 0) close(fd);
    free(c);
    return kSystemError;
  }
  void* p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    free(c);
    return kSystemError;
  }
  c->hdr = (Header*)p; /* a fresh object is all zeros: a valid initial state of every field */
  c->slots = (char*)p + 4096;
  atomic_fetch_add(&c->hdr->attached, 1);
  const int rc = barrier(c); /* like ncclCommInitRank, returns when every rank has joined */
  if (rank == 0) shm_unlink(c->name); /* everybody has it mapped (or has given up): the name can go */
  if (rc != kSuccess) {
    munmap(p, c->map_bytes);
    free(c);
    return rc;
  }
  *comm = c;
  return kSuccess;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return kSuccess;
  atomic_fetch_sub(&c->hdr->attached, 1);
  munmap((void*)c->hdr, c->map_bytes);
  free(c);
  return kSuccess;
}

/* what a communicator says about itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), and a version no RCCL ever had */
int ncclCommCount(const void* comm, int* count) {
  if (!comm || !count) return kInvalidArgument;
  *count = ((const Comm*)comm)->world;
  return kSuccess;
}

int ncclCommUserRank(const void* comm, int* rank) {
  if (!comm || !rank) return kInvalidArgument;
  *rank = ((const Comm*)comm)->rank;
  return kSuccess;
}

int ncclCommCuDevice(const void* comm, int* device) {
  if (!comm || !device) return kInvalidArgument;
  return hipGetDevice(device) == hipSuccess ? kSuccess : kUnhandledCuda;
}

int ncclGetVersion(int* version) {
  if (!version) return kInvalidArgument;
  *version = 1;          /* "fake_rccl": real versions are >= 20000 */
  return kSuccess;
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream) {
  const size_t es = dtype_bytes(datatype);
  if (!comm || !es || (sendcount && (!sendbuff || !recvbuff))) return kInvalidArgument;
  Op op = {0, sendbuff, recvbuff, sendcount * es, 0, (Comm*)comm, stream};
  return submit(&op);
}

int ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, int datatype, int root, void* comm, hipStream_t stream) {
  const size_t es = dtype_bytes(datatype);
  Comm* c = (Comm*)comm;
  if (!c || !es || root < 0 || root >= c->world || (count && !recvbuff)) return kInvalidArgument;
  Op op = {1, sendbuff, recvbuff, count * es, root, c, stream};
  return submit(&op);
}

int ncclGroupStart(void) {
  ++g_depth;
  return kSuccess;
}

int ncclGroupEnd(void) {
  if (g_depth <= 0) return kInvalidUsage;
  if (--g_depth > 0) return kSuccess;
  int rc = kSuccess;
  const int n = g_nops;
  g_nops = 0;
  if (n) atomic_fetch_add(&g_ops[0].comm->hdr->calls[2], 1);
  for (int i = 0; i < n && rc == kSuccess; ++i) rc = run_op(&g_ops[i]);
  return rc;
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case kSuccess: return "no error";
    case kUnhandledCuda: return "unhandled HIP error (fake_rccl)";
    case kSystemError: return "system error: a peer did not arrive (fake_rccl)";
    case kInvalidArgument: return "invalid argument (fake_rccl)";
    case kInvalidUsage: return "invalid usage (fake_rccl)";
    default: return "internal error (fake_rccl)";
  }
}

/* statistics for the tests: what = 0 all-gathers, 1 broadcasts, 2 groups, 3 payload bytes -- summed over the ranks */
long long fake_rccl_stat(void* comm, int what) {
  Comm* c = (Comm*)comm;
  if (!c || what < 0 || what > 3) return -1;
  return atomic_load(&c->hdr->calls[what]);
}
