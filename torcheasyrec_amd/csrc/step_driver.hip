// Native step driver: one host call issues a whole sharded train step -- its captured hipGraphs and the RCCL collectives
// between them -- instead of ~25 Python calls and 5 torch ProcessGroup calls.
//
// Replaces, for the steady state of a job, the host side of the reference's TrainPipelineSparseDist.progress
// (/root/reference/tzrec/utils/dist_util.py:221-303): there torchrec's pipeline issues the input dist, the forward /
// backward and the dists' collectives from Python through torch.distributed.  At a rank's share of the batch (8 192 samples
// per rank = `batch_size: 8192` of examples/dlrm_criteo.config) the step's kernels take ~0.24 ms and the HOST needs
// 0.35 ms to queue them that way (profiles/r04t): the step is as fast as Python.  Here:
//
//   * a COMMUNICATOR of this library's own (tzr_comm_*): RCCL reached directly (the librccl the process already holds,
//     through dlsym -- no second copy), created from a unique id that rank 0 makes and the job's launcher carries to the
//     others (torch.distributed's store / one broadcast at start-up).  No torch ProcessGroup in the step, hence no watchdog
//     thread polling events next to captures;
//   * a PROGRAM (tzr_step_*): a list of ops recorded once per pipeline slot -- launch graph / all-to-all / all-reduce
//     (issued on the driver's communication stream behind everything queued so far, completion event kept) / wait (the
//     compute stream waits for a collective's event) -- and tzr_step_run walks it: hipGraphLaunch + RCCL calls + event
//     waits, ~3-5 us each, no allocation, no host synchronisation.
//
// Collectives run on static buffers (the slot's message buffers, the replicas' accumulation buffer, the flat dense
// gradient): the same addresses every step, which is what makes the program a constant.
// Host-only code: this file holds no kernels.  RCCL's API is declared by <rccl/rccl.h>; nothing links against it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/tzrec_hip.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllToAll) AllToAll = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

template <class F>
bool sym(void* h, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(h, name));
  return *out != nullptr;
}

// `path`: the librccl.so the process already uses (torch's: <torch>/lib/librccl.so), or null for the default search
bool load_rccl(const char* path) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.ok) return true;
  void* h = nullptr;
  if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return false;
  Rccl r;
  r.handle = h;
  if (!sym(h, "ncclGetUniqueId", &r.GetUniqueId) || !sym(h, "ncclCommInitRank", &r.CommInitRank) ||
      !sym(h, "ncclCommDestroy", &r.CommDestroy) || !sym(h, "ncclAllReduce", &r.AllReduce) ||
      !sym(h, "ncclAllToAll", &r.AllToAll) || !sym(h, "ncclGetVersion", &r.GetVersion))
    return false;
  r.ok = true;
  g_rccl = r;
  return true;
}

struct Comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0;
};

enum OpKind { OP_GRAPH = 0, OP_A2A = 1, OP_ALLREDUCE = 2, OP_WAIT = 3 };

struct Op {
  int kind = 0;
  hipGraphExec_t graph = nullptr;
  Comm* comm = nullptr;
  const void* send = nullptr;
  void* recv = nullptr;
  size_t count = 0;  // all-to-all: BYTES per peer; all-reduce: floats
  int avg = 0;
  int sync = 0;       // collective: 1 = the compute stream waits for it right away (in stream order, like a kernel)
  int target = -1;    // wait: index of the collective op
  hipEvent_t issued = nullptr, done = nullptr;
};

struct Program {
  std::vector<Op> ops;
  hipStream_t comm_stream = nullptr;
};

int coll_issue(const Op& op, hipStream_t s) {
  ncclResult_t rc;
  if (op.kind == OP_A2A)
    rc = g_rccl.AllToAll(op.send, op.recv, op.count, ncclInt8, op.comm->comm, s);
  else
    rc = g_rccl.AllReduce(op.send, op.recv, op.count, ncclFloat32, op.avg ? ncclAvg : ncclSum, op.comm->comm, s);
  return rc == ncclSuccess ? TZR_OK : TZR_ERR_LAUNCH;
}

}  // namespace

extern "C" int tzr_comm_available(const char* librccl_path) { return load_rccl(librccl_path) ? 1 : 0; }

extern "C" int tzr_comm_version(const char* librccl_path) {
  if (!load_rccl(librccl_path)) return -1;
  int v = 0;
  return g_rccl.GetVersion(&v) == ncclSuccess ? v : -1;
}

extern "C" int tzr_comm_unique_id(const char* librccl_path, void* out, size_t out_bytes) {
  if (!out || out_bytes < NCCL_UNIQUE_ID_BYTES) return TZR_ERR_INVALID;
  if (!load_rccl(librccl_path)) return TZR_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return TZR_ERR_LAUNCH;
  std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return TZR_OK;
}

extern "C" int tzr_comm_create(const char* librccl_path, const void* unique_id, size_t id_bytes, int world, int rank,
                               void** out_comm) {
  if (!unique_id || id_bytes < NCCL_UNIQUE_ID_BYTES || world <= 0 || rank < 0 || rank >= world || !out_comm) return TZR_ERR_INVALID;
  if (!load_rccl(librccl_path)) return TZR_ERR_UNSUPPORTED;
  ncclUniqueId id;
  std::memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  Comm* c = new Comm();
  c->world = world;
  c->rank = rank;
  if (g_rccl.CommInitRank(&c->comm, world, id, rank) != ncclSuccess) {
    delete c;
    return TZR_ERR_LAUNCH;
  }
  *out_comm = c;
  return TZR_OK;
}

extern "C" int tzr_comm_destroy(void* comm) {
  if (!comm) return TZR_OK;
  Comm* c = static_cast<Comm*>(comm);
  if (c->comm && g_rccl.ok) g_rccl.CommDestroy(c->comm);
  delete c;
  return TZR_OK;
}

// In stream order on `stream`, like a kernel.  bytes_per_peer: what every rank sends to (and receives from) each rank.
extern "C" int tzr_comm_all_to_all(void* comm, const void* d_send, void* d_recv, int64_t bytes_per_peer, void* stream) {
  if (!comm || bytes_per_peer < 0 || !g_rccl.ok) return TZR_ERR_INVALID;
  if (bytes_per_peer == 0) return TZR_OK;
  if (!d_send || !d_recv) return TZR_ERR_INVALID;
  Op op;
  op.kind = OP_A2A;
  op.comm = static_cast<Comm*>(comm);
  op.send = d_send;
  op.recv = d_recv;
  op.count = (size_t)bytes_per_peer;
  return coll_issue(op, static_cast<hipStream_t>(stream));
}

extern "C" int tzr_comm_all_reduce(void* comm, float* d_buf, int64_t count, int average, void* stream) {
  if (!comm || count < 0 || !g_rccl.ok) return TZR_ERR_INVALID;
  if (count == 0) return TZR_OK;
  if (!d_buf) return TZR_ERR_INVALID;
  Op op;
  op.kind = OP_ALLREDUCE;
  op.comm = static_cast<Comm*>(comm);
  op.send = d_buf;
  op.recv = d_buf;
  op.count = (size_t)count;
  op.avg = average;
  return coll_issue(op, static_cast<hipStream_t>(stream));
}

extern "C" int tzr_step_create(void** out_program) {
  if (!out_program) return TZR_ERR_INVALID;
  Program* p = new Program();
  if (hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking) != hipSuccess) {
    delete p;
    return TZR_ERR_LAUNCH;
  }
  *out_program = p;
  return TZR_OK;
}

extern "C" int tzr_step_destroy(void* program) {
  if (!program) return TZR_OK;
  Program* p = static_cast<Program*>(program);
  for (Op& op : p->ops) {
    if (op.issued) (void)hipEventDestroy(op.issued);
    if (op.done) (void)hipEventDestroy(op.done);
  }
  if (p->comm_stream) (void)hipStreamDestroy(p->comm_stream);
  delete p;
  return TZR_OK;
}

// graph_exec: a hipGraphExec_t (torch.cuda.CUDAGraph.raw_cuda_graph_exec()).  Returns the op's index (>= 0) or an error.
extern "C" int tzr_step_add_graph(void* program, void* graph_exec) {
  if (!program || !graph_exec) return TZR_ERR_INVALID;
  Program* p = static_cast<Program*>(program);
  Op op;
  op.kind = OP_GRAPH;
  op.graph = static_cast<hipGraphExec_t>(graph_exec);
  p->ops.push_back(op);
  return (int)p->ops.size() - 1;
}

static int add_coll(Program* p, Op op, int sync) {
  op.sync = sync;
  if (hipEventCreateWithFlags(&op.issued, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&op.done, hipEventDisableTiming) != hipSuccess)
    return TZR_ERR_LAUNCH;
  p->ops.push_back(op);
  return (int)p->ops.size() - 1;
}

// sync != 0: the compute stream waits for the collective where it is issued; sync == 0: a later tzr_step_add_wait names it.
extern "C" int tzr_step_add_all_to_all(void* program, void* comm, const void* d_send, void* d_recv, int64_t bytes_per_peer,
                                       int sync) {
  if (!program || !comm || !d_send || !d_recv || bytes_per_peer <= 0) return TZR_ERR_INVALID;
  Op op;
  op.kind = OP_A2A;
  op.comm = static_cast<Comm*>(comm);
  op.send = d_send;
  op.recv = d_recv;
  op.count = (size_t)bytes_per_peer;
  return add_coll(static_cast<Program*>(program), op, sync);
}

extern "C" int tzr_step_add_all_reduce(void* program, void* comm, float* d_buf, int64_t count, int average, int sync) {
  if (!program || !comm || !d_buf || count <= 0) return TZR_ERR_INVALID;
  Op op;
  op.kind = OP_ALLREDUCE;
  op.comm = static_cast<Comm*>(comm);
  op.send = d_buf;
  op.recv = d_buf;
  op.count = (size_t)count;
  op.avg = average;
  return add_coll(static_cast<Program*>(program), op, sync);
}

extern "C" int tzr_step_add_wait(void* program, int collective_op) {
  if (!program) return TZR_ERR_INVALID;
  Program* p = static_cast<Program*>(program);
  if (collective_op < 0 || collective_op >= (int)p->ops.size()) return TZR_ERR_INVALID;
  const int k = p->ops[collective_op].kind;
  if (k != OP_A2A && k != OP_ALLREDUCE) return TZR_ERR_INVALID;
  Op op;
  op.kind = OP_WAIT;
  op.target = collective_op;
  p->ops.push_back(op);
  return (int)p->ops.size() - 1;
}

extern "C" int tzr_step_num_ops(void* program) { return program ? (int)static_cast<Program*>(program)->ops.size() : TZR_ERR_INVALID; }

// Queue the whole program: graphs on `stream`, collectives on the program's communication stream behind everything `stream`
// holds at that point.  Asynchronous; returns when everything is queued.
extern "C" int tzr_step_run(void* program, void* stream) {
  if (!program) return TZR_ERR_INVALID;
  Program* p = static_cast<Program*>(program);
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (Op& op : p->ops) {
    switch (op.kind) {
      case OP_GRAPH:
        if (hipGraphLaunch(op.graph, s) != hipSuccess) return TZR_ERR_LAUNCH;
        break;
      case OP_A2A:
      case OP_ALLREDUCE: {
        if (hipEventRecord(op.issued, s) != hipSuccess) return TZR_ERR_LAUNCH;
        if (hipStreamWaitEvent(p->comm_stream, op.issued, 0) != hipSuccess) return TZR_ERR_LAUNCH;
        const int rc = coll_issue(op, p->comm_stream);
        if (rc != TZR_OK) return rc;
        if (hipEventRecord(op.done, p->comm_stream) != hipSuccess) return TZR_ERR_LAUNCH;
        if (op.sync && hipStreamWaitEvent(s, op.done, 0) != hipSuccess) return TZR_ERR_LAUNCH;
        break;
      }
      case OP_WAIT:
        if (hipStreamWaitEvent(s, p->ops[op.target].done, 0) != hipSuccess) return TZR_ERR_LAUNCH;
        break;
      default:
        return TZR_ERR_INVALID;
    }
  }
  return TZR_OK;
}
