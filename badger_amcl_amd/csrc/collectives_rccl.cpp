// libbadger_pf_rccl.so -- the two collectives of the sharded path over RCCL (xGMI), for ranks that cannot use the
// mailbox exchange (bpf_shard_bootstrap falls back to this when a peer's memory cannot be mapped or the mailbox
// self-test fails).  A separate shared object, linked against librccl at build time, that the engine loads only when
// it needs it: librccl.so is 570 MB and the mailbox path never touches it.
//
//   totals   ncclAllGather of W per-shard weight totals (8 bytes each)
//   window   ncclAllReduce(sum, int64) of a [6][count] draw window in which every column has exactly one non-zero
//            writer, so the integer sum is bit-exact
//   counts   ncclAllReduce(sum, int32) of the beam-skip agreement counts
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

namespace
{
thread_local std::string g_error;
int fail(ncclResult_t r, const char* what)
{
  g_error = std::string(what) + ": " + ncclGetErrorString(r);
  return 1;
}
}  // namespace

extern "C" {

const char* bpfc_last_error() { return g_error.c_str(); }

int bpfc_unique_id_bytes() { return (int)sizeof(ncclUniqueId); }

int bpfc_unique_id(void* out)
{
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess)
    return fail(r, "ncclGetUniqueId");
  std::memcpy(out, &id, sizeof(id));
  return 0;
}

int bpfc_init(void** comm_out, int rank, int world, const void* unique_id)
{
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess)
    return fail(r, "ncclCommInitRank");
  *comm_out = comm;
  return 0;
}

int bpfc_destroy(void* comm)
{
  if (!comm)
    return 0;
  const ncclResult_t r = ncclCommDestroy(static_cast<ncclComm_t>(comm));
  return r == ncclSuccess ? 0 : fail(r, "ncclCommDestroy");
}

int bpfc_allgather_f64(void* comm, const double* send, double* recv, size_t count_per_rank, void* stream)
{
  const ncclResult_t r = ncclAllGather(send, recv, count_per_rank, ncclDouble, static_cast<ncclComm_t>(comm),
                                       static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? 0 : fail(r, "ncclAllGather");
}

int bpfc_allreduce_sum_i64(void* comm, long long* buf, size_t count, void* stream)
{
  const ncclResult_t r = ncclAllReduce(buf, buf, count, ncclInt64, ncclSum, static_cast<ncclComm_t>(comm),
                                       static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? 0 : fail(r, "ncclAllReduce(int64)");
}

int bpfc_allreduce_sum_i32(void* comm, int* buf, size_t count, void* stream)
{
  const ncclResult_t r = ncclAllReduce(buf, buf, count, ncclInt32, ncclSum, static_cast<ncclComm_t>(comm),
                                       static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? 0 : fail(r, "ncclAllReduce(int32)");
}

}  // extern "C"
