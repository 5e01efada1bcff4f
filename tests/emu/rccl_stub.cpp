// TEST INFRASTRUCTURE ONLY -- librccl_stub.so: the six RCCL entry points csrc/step_driver.hip looks up with dlsym, implemented
// between the processes of ONE host over POSIX shared memory.  `tzr_comm_create` takes the path of the RCCL library to use
// (include/tzrec_hip.h), so the CPU suite hands it this file and runs the unchanged native step driver -- its communicators, its
// program order of collectives -- at world size 2 / 4 next to the gloo process group (tests/test_sharded_gloo.py).
//
// A communicator = a control segment "/tzrs_<id>" (a sense-reversing barrier).  A collective = every rank publishes its send
// buffer as a segment of its own ("/tzrs_<id>_<rank>_<seq>"), barrier, every rank reads what it needs from the others'
// segments, barrier, the owner unlinks.  All-reduce adds the ranks' buffers in RANK ORDER (every rank computes the same
// sums: deterministic and identical on all ranks), `ncclAvg` multiplies the sum by 1 / world.  A rank that waits longer than
// TZR_STUB_TIMEOUT_S (default 120) for the others fails the call -- ranks issuing their collectives in different orders show
// up as an error, not as a hang.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rccl/rccl.h"

namespace {

struct Ctl {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> gen;
};

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

double timeout_s() {
  const char* e = getenv("TZR_STUB_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}

}  // namespace

struct ncclComm {
  int world = 0, rank = 0;
  std::string name;
  Ctl* ctl = nullptr;
  uint64_t seq = 0;
  bool barrier() {
    if (world == 1) return true;
    const uint32_t g = ctl->gen.load(std::memory_order_acquire);
    if (ctl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
      ctl->arrived.store(0, std::memory_order_relaxed);
      ctl->gen.fetch_add(1, std::memory_order_release);
      return true;
    }
    const double t0 = now_s(), lim = timeout_s();
    int spins = 0;
    while (ctl->gen.load(std::memory_order_acquire) == g) {
      if (++spins < 200) {
        sched_yield();
      } else {
        usleep(50);
        if (now_s() - t0 > lim) {
          fprintf(stderr, "[rccl_stub] rank %d of %s: waited %.0f s at collective %llu for the other ranks\n", rank, name.c_str(), lim,
                  (unsigned long long)seq);
          return false;
        }
      }
    }
    return true;
  }
  std::string seg(int r) const { return name + "_" + std::to_string(r) + "_" + std::to_string(seq); }
};

namespace {

void* map_shm(const std::string& name, size_t bytes, bool create) {
  const int fd = shm_open(name.c_str(), create ? (O_CREAT | O_RDWR) : O_RDONLY, 0600);
  if (fd < 0) return nullptr;
  if (create && ftruncate(fd, (off_t)bytes) != 0) {
    close(fd);
    return nullptr;
  }
  void* p = mmap(nullptr, bytes, create ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, fd, 0);
  close(fd);
  return p == MAP_FAILED ? nullptr : p;
}

// every rank publishes `bytes` of `send`; reader(r, peer's buffer) for every rank r in order; false on a timeout / system error
template <class F>
bool exchange(ncclComm* c, const void* send, size_t bytes, F reader) {
  if (c->world == 1) {
    reader(0, send);
    return true;
  }
  const std::string mine = c->seg(c->rank);
  void* pub = map_shm(mine, bytes, true);
  if (!pub) return false;
  std::memcpy(pub, send, bytes);
  bool ok = c->barrier();
  if (ok) {
    for (int r = 0; r < c->world && ok; ++r) {
      if (r == c->rank) {
        reader(r, pub);
        continue;
      }
      void* p = map_shm(c->seg(r), bytes, false);
      if (!p) {
        ok = false;
        break;
      }
      reader(r, p);
      munmap(p, bytes);
    }
    ok = c->barrier() && ok;
  }
  munmap(pub, bytes);
  shm_unlink(mine.c_str());
  c->seq++;
  return ok;
}

}  // namespace

extern "C" ncclResult_t ncclGetVersion(int* version) {
  if (!version) return ncclInvalidArgument;
  *version = 22606;  // (what the box's RCCL reports: 2.26.6)
  return ncclSuccess;
}

extern "C" ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  static std::atomic<uint32_t> counter{0};
  std::memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "/tzrs_%x_%llx_%x", (unsigned)getpid(), (unsigned long long)ts.tv_nsec + 1000000000ull * ts.tv_sec,
           counter.fetch_add(1));
  return ncclSuccess;
}

extern "C" ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks <= 0 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->world = nranks;
  c->rank = rank;
  c->name.assign(id.internal, strnlen(id.internal, NCCL_UNIQUE_ID_BYTES));
  if (nranks > 1) {
    c->ctl = static_cast<Ctl*>(map_shm(c->name, sizeof(Ctl), true));  // (zero pages: both counters start at 0 whoever comes first)
    if (!c->ctl || !c->barrier()) {  // everyone is attached before anyone unlinks
      delete c;
      return ncclSystemError;
    }
    if (!c->barrier()) {
      delete c;
      return ncclSystemError;
    }
    if (rank == 0) shm_unlink(c->name.c_str());  // (the mapping stays; the name does not outlive the job)
  }
  *comm = c;
  return ncclSuccess;
}

extern "C" ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclSuccess;
  if (comm->ctl) munmap(comm->ctl, sizeof(Ctl));
  delete comm;
  return ncclSuccess;
}

extern "C" ncclResult_t ncclAllToAll(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t) {
  if (!comm || !send || !recv || dt != ncclInt8) return ncclInvalidArgument;  // (the driver moves bytes)
  const size_t per = count, all = per * (size_t)comm->world;
  char* out = static_cast<char*>(recv);
  const int me = comm->rank;
  std::vector<char> tmp;  // (send == recv is not a case the driver produces; a copy keeps the stub safe anyway)
  if (send == recv) {
    tmp.assign(static_cast<const char*>(send), static_cast<const char*>(send) + all);
    send = tmp.data();
  }
  const bool ok = exchange(comm, send, all, [&](int r, const void* p) {
    std::memcpy(out + (size_t)r * per, static_cast<const char*>(p) + (size_t)me * per, per);
  });
  return ok ? ncclSuccess : ncclSystemError;
}

extern "C" ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                                      hipStream_t) {
  if (!comm || !send || !recv || dt != ncclFloat32 || (op != ncclSum && op != ncclAvg)) return ncclInvalidArgument;
  std::vector<float> acc(count, 0.0f);
  const bool ok = exchange(comm, send, count * sizeof(float), [&](int r, const void* p) {
    const float* x = static_cast<const float*>(p);
    if (r == 0)
      std::memcpy(acc.data(), x, count * sizeof(float));
    else
      for (size_t i = 0; i < count; ++i) acc[i] += x[i];
  });
  if (!ok) return ncclSystemError;
  float* out = static_cast<float*>(recv);
  if (op == ncclAvg) {
    const float inv = 1.0f / (float)comm->world;
    for (size_t i = 0; i < count; ++i) out[i] = acc[i] * inv;
  } else {
    std::memcpy(out, acc.data(), count * sizeof(float));
  }
  return ncclSuccess;
}
