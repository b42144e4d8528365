// The one collective of the path, behind the C ABI: the all-reduce of {sum of log_prob, row count} over RCCL.
//
// Every bijector is row-wise over the batch, so a batch-sharded `Flow.log_prob` exchanges nothing
// but these 16 bytes per evaluation: one ncclAllReduce(sum) of two float64 on the compute stream, after the last
// layer.  At this size the collective is latency-bound (RCCL's one-shot / tree path over xGMI), not link-bound.
// The reference has no distributed code: there is nothing to translate.
//
// RCCL is resolved at run time (dlopen): the process that hosts this library normally has PyTorch's copy of
// librccl.so.1 loaded already, and a second, link-time copy of the runtime in one process is what we do not want.
// A box without RCCL still loads libflowcon_hip.so; only these entry points then return hipErrorNotSupported.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include "../../include/flowcon_hip.h"

namespace fc {

struct Rccl {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclCommAbort) comm_abort = nullptr;
  decltype(&ncclCommCount) comm_count = nullptr;
  bool ok = false;
};

static const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    // the copy already in the process first (PyTorch's, by soname), then the system one
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(h, "ncclAllReduce"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    r.comm_abort = reinterpret_cast<decltype(r.comm_abort)>(dlsym(h, "ncclCommAbort"));
    r.comm_count = reinterpret_cast<decltype(r.comm_count)>(dlsym(h, "ncclCommCount"));
    r.ok = r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy;
  });
  return r;
}

// ncclResult_t -> the ABI's int: 0 on success, otherwise 10000 + the RCCL code (outside hipError_t's range)
static int rc(ncclResult_t e) { return e == ncclSuccess ? 0 : 10000 + (int)e; }

}  // namespace fc

extern "C" int fc_comm_unique_id(void* id_out128) {
  const fc::Rccl& r = fc::rccl();
  if (!r.ok) return hipErrorNotSupported;
  if (!id_out128) return hipErrorInvalidValue;
  static_assert(sizeof(ncclUniqueId) == FC_COMM_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  const ncclResult_t e = r.get_unique_id(&id);
  if (e == ncclSuccess) memcpy(id_out128, &id, sizeof(id));
  return fc::rc(e);
}

extern "C" int fc_comm_init_rank(void** comm_out, int32_t nranks, const void* id128, int32_t rank) {
  const fc::Rccl& r = fc::rccl();
  if (!r.ok) return hipErrorNotSupported;
  if (!comm_out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return hipErrorInvalidValue;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t e = r.comm_init_rank(&comm, nranks, id, rank);   // binds the CURRENT device
  *comm_out = e == ncclSuccess ? static_cast<void*>(comm) : nullptr;
  return fc::rc(e);
}

extern "C" int fc_comm_init_rank_on_device(void** comm_out, int32_t nranks, const void* id128, int32_t rank, int32_t device) {
  // The current device is per host THREAD: a caller that sets up the communicator off its main thread (a deadline
  // around this blocking collective) would otherwise bind device 0 on every rank.
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess) return e;
  if (device < 0 || device >= count) return hipErrorInvalidDevice;
  e = hipSetDevice(device);
  if (e != hipSuccess) return e;
  return fc_comm_init_rank(comm_out, nranks, id128, rank);
}

extern "C" int fc_comm_abort(void* comm) {
  const fc::Rccl& r = fc::rccl();
  if (!r.ok || !r.comm_abort) return hipErrorNotSupported;
  if (!comm) return 0;
  return fc::rc(r.comm_abort(static_cast<ncclComm_t>(comm)));
}

extern "C" int fc_comm_destroy(void* comm) {
  const fc::Rccl& r = fc::rccl();
  if (!r.ok) return hipErrorNotSupported;
  if (!comm) return 0;
  return fc::rc(r.comm_destroy(static_cast<ncclComm_t>(comm)));
}

extern "C" int fc_allreduce_loglik(double* sum_count, void* comm, void* stream) {
  const fc::Rccl& r = fc::rccl();
  if (!r.ok) return hipErrorNotSupported;
  if (!sum_count || !comm) return hipErrorInvalidValue;
  return fc::rc(r.all_reduce(sum_count, sum_count, 2, ncclFloat64, ncclSum, static_cast<ncclComm_t>(comm),
                             static_cast<hipStream_t>(stream)));
}
