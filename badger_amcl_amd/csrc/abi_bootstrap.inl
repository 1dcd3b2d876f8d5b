// C-ABI: bring-up of a sharded filter without any Python or MPI around it.
//
//   bpf_shard_bootstrap(e, rank, world, "host:port", max_window, flags, &mode)
//
// Every rank of the node calls it with the same address.  Rank 0 listens there, the others connect (TCP, inside this
// library: no launcher, no torch.distributed); over that star the ranks gather their mailbox IPC handles, map each
// other's mailboxes, run the connect round and the four-window self-test, and agree (minimum over the ranks) on
// whether the mailbox exchange is usable.  If it is not -- a peer's memory cannot be mapped, a word or a window cell
// does not arrive -- every rank drops the mailbox, loads libbadger_pf_rccl.so (linked against librccl) and joins an
// RCCL communicator whose unique id rank 0 hands out over the same sockets: the totals then travel by ncclAllGather
// and the draw windows by an integer ncclAllReduce (abi_mailbox_step.inl).  The sockets are closed when the call
// returns; nothing of the data path ever goes through them.
//
//   bpf_shard_update_sensor_planar / bpf_shard_update_resample
//
// the sharded sensor update and resample as one call each, whichever exchange the bootstrap chose.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

namespace
{
struct Rendezvous
{
  int rank = 0, world = 1;
  int listen_fd = -1;
  std::vector<int> fds;  // rank 0: socket of every other rank (index = rank); others: fds[0] = rank 0
  std::string error;

  ~Rendezvous() { close_all(); }
  void close_all()
  {
    for (int fd : fds)
      if (fd >= 0)
        ::close(fd);
    fds.clear();
    if (listen_fd >= 0)
      ::close(listen_fd);
    listen_fd = -1;
  }

  static bool send_all(int fd, const void* p, size_t n)
  {
    const char* c = static_cast<const char*>(p);
    while (n > 0)
    {
      const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
      if (k <= 0)
        return false;
      c += k;
      n -= (size_t)k;
    }
    return true;
  }

  static bool recv_all(int fd, void* p, size_t n, int timeout_ms)
  {
    char* c = static_cast<char*>(p);
    while (n > 0)
    {
      pollfd pf{ fd, POLLIN, 0 };
      if (::poll(&pf, 1, timeout_ms) <= 0)
        return false;
      const ssize_t k = ::recv(fd, c, n, 0);
      if (k <= 0)
        return false;
      c += k;
      n -= (size_t)k;
    }
    return true;
  }

  bool open(int rank_, int world_, const std::string& host, int port, int timeout_ms)
  {
    rank = rank_;
    world = world_;
    if (world == 1)
      return true;
    const auto t0 = std::chrono::steady_clock::now();
    auto left = [&]() {
      return timeout_ms - (int)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    };
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (::getaddrinfo(host.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res)
    {
      error = "cannot resolve " + host;
      return false;
    }
    sockaddr_in addr = *reinterpret_cast<sockaddr_in*>(res->ai_addr);
    ::freeaddrinfo(res);
    const int one = 1;
    if (rank == 0)
    {
      listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
      ::setsockopt(listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
      if (::bind(listen_fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0 || ::listen(listen_fd, world) != 0)
      {
        error = "rank 0 cannot listen on " + host + ":" + std::to_string(port);
        return false;
      }
      fds.assign((size_t)world, -1);
      for (int got = 0; got < world - 1;)
      {
        pollfd pf{ listen_fd, POLLIN, 0 };
        if (left() <= 0 || ::poll(&pf, 1, left()) <= 0)
        {
          error = "rendez-vous: not every rank connected in time";
          return false;
        }
        const int fd = ::accept(listen_fd, nullptr, nullptr);
        if (fd < 0)
          continue;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        int peer = -1;
        if (!recv_all(fd, &peer, sizeof(peer), std::max(left(), 1)) || peer <= 0 || peer >= world || fds[(size_t)peer] >= 0)
        {
          ::close(fd);
          error = "rendez-vous: a peer announced an invalid or duplicate rank";
          return false;
        }
        fds[(size_t)peer] = fd;
        ++got;
      }
      return true;
    }
    // the server may not be up yet: retry until the time is out
    for (;;)
    {
      const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (::connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) == 0)
      {
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        if (!send_all(fd, &rank, sizeof(rank)))
        {
          ::close(fd);
          error = "rendez-vous: cannot announce the rank";
          return false;
        }
        fds.assign(1, fd);
        return true;
      }
      ::close(fd);
      if (left() <= 0)
      {
        error = "rendez-vous: cannot reach rank 0 at " + host + ":" + std::to_string(port);
        return false;
      }
      ::usleep(20000);
    }
  }

  // every rank contributes `bytes` bytes; every rank gets all of them in rank order
  bool all_gather(const void* mine, size_t bytes, std::vector<unsigned char>* all, int timeout_ms)
  {
    all->assign(bytes * (size_t)world, 0);
    std::memcpy(all->data() + bytes * (size_t)rank, mine, bytes);
    if (world == 1)
      return true;
    if (rank == 0)
    {
      for (int r = 1; r < world; ++r)
        if (!recv_all(fds[(size_t)r], all->data() + bytes * (size_t)r, bytes, timeout_ms))
        {
          error = "rendez-vous: rank " + std::to_string(r) + " went silent";
          return false;
        }
      for (int r = 1; r < world; ++r)
        if (!send_all(fds[(size_t)r], all->data(), all->size()))
        {
          error = "rendez-vous: cannot answer rank " + std::to_string(r);
          return false;
        }
      return true;
    }
    if (!send_all(fds[0], mine, bytes) || !recv_all(fds[0], all->data(), all->size(), timeout_ms))
    {
      error = "rendez-vous: rank 0 went silent";
      return false;
    }
    return true;
  }

  // minimum over the ranks of a small integer (agreement on success)
  bool all_min(int mine, int* out, int timeout_ms)
  {
    std::vector<unsigned char> all;
    if (!all_gather(&mine, sizeof(mine), &all, timeout_ms))
      return false;
    int m = mine;
    for (int r = 0; r < world; ++r)
    {
      int v;
      std::memcpy(&v, all.data() + sizeof(int) * (size_t)r, sizeof(int));
      m = std::min(m, v);
    }
    *out = m;
    return true;
  }
};

void collective_release(bpf_engine* e)
{
  bpf_engine::Collective& c = e->coll;
  if (c.comm && c.fn.destroy)
    (void)c.fn.destroy(c.comm);
  c.comm = nullptr;
  c.active = false;
  // the library stays loaded for the life of the process (RCCL does not like being unloaded)
}

// libbadger_pf_rccl.so sits next to this library
int collective_load(bpf_engine* e)
{
  bpf_engine::Collective& c = e->coll;
  if (c.lib)
    return BPF_OK;
  std::string dir;
  Dl_info info;
  if (::dladdr(reinterpret_cast<const void*>(&collective_release), &info) && info.dli_fname)
  {
    dir = info.dli_fname;
    const size_t slash = dir.find_last_of('/');
    dir = (slash == std::string::npos) ? std::string() : dir.substr(0, slash + 1);
  }
  const std::string path = dir + "libbadger_pf_rccl.so";
  c.lib = ::dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!c.lib)
    return e->fail(BPF_ERR_NOT_CONFIGURED, std::string("cannot load ") + path + ": " + ::dlerror());
  auto sym = [&](const char* name) { return ::dlsym(c.lib, name); };
  c.fn.last_error = reinterpret_cast<const char* (*)()>(sym("bpfc_last_error"));
  c.fn.unique_id_bytes = reinterpret_cast<int (*)()>(sym("bpfc_unique_id_bytes"));
  c.fn.unique_id = reinterpret_cast<int (*)(void*)>(sym("bpfc_unique_id"));
  c.fn.init = reinterpret_cast<int (*)(void**, int, int, const void*)>(sym("bpfc_init"));
  c.fn.destroy = reinterpret_cast<int (*)(void*)>(sym("bpfc_destroy"));
  c.fn.allgather_f64 = reinterpret_cast<int (*)(void*, const double*, double*, size_t, void*)>(sym("bpfc_allgather_f64"));
  c.fn.allreduce_sum_i64 = reinterpret_cast<int (*)(void*, long long*, size_t, void*)>(sym("bpfc_allreduce_sum_i64"));
  c.fn.allreduce_sum_i32 = reinterpret_cast<int (*)(void*, int*, size_t, void*)>(sym("bpfc_allreduce_sum_i32"));
  if (!c.fn.last_error || !c.fn.unique_id_bytes || !c.fn.unique_id || !c.fn.init || !c.fn.destroy ||
      !c.fn.allgather_f64 || !c.fn.allreduce_sum_i64 || !c.fn.allreduce_sum_i32)
    return e->fail(BPF_ERR_NOT_CONFIGURED, path + " does not export the collective entry points");
  return BPF_OK;
}
}  // namespace

int bpf_shard_bootstrap(bpf_engine* e, int rank, int world, const char* host_port, long long max_window, int flags,
                        int* mode_out)
{
  if (!e || !host_port || world < 1 || rank < 0 || rank >= world || max_window < 1)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bootstrap: 0 <= rank < world, host:port, max_window >= 1")
             : BPF_ERR_INVALID_ARGUMENT;
  const int timeout_ms = 30000;
  std::string hp(host_port);
  const size_t colon = hp.find_last_of(':');
  if (colon == std::string::npos)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "bootstrap: address must be host:port");
  const std::string host = hp.substr(0, colon);
  const int port = std::atoi(hp.c_str() + colon + 1);
  HIPCHK(e, hipSetDevice(e->device));
  collective_release(e);
  (void)bpf_shard_mailbox_destroy(e);
  e->shard_rank = rank;
  e->shard_world = world;
  Rendezvous rv;
  if (!rv.open(rank, world, host, port, timeout_ms))
    return e->fail(BPF_ERR_EXCHANGE, rv.error);
  // ---- the mailbox: create, gather the handles, connect, self-test; every step agreed by all ranks
  int ok = (flags & BPF_BOOTSTRAP_FORCE_COLLECTIVE) ? 0 : 1;
  if (world > kMailboxMaxWorld)
    ok = 0;
  int agreed = 0;
  if (!rv.all_min(ok, &agreed, timeout_ms))
    return e->fail(BPF_ERR_EXCHANGE, rv.error);
  if (agreed)
  {
    unsigned char mine[BPF_MAILBOX_HANDLE_BYTES] = { 0 };
    ok = bpf_shard_mailbox_create(e, rank, world, max_window, mine) == BPF_OK ? 1 : 0;
    std::vector<unsigned char> all;
    if (!rv.all_gather(mine, sizeof(mine), &all, timeout_ms) || !rv.all_min(ok, &agreed, timeout_ms))
      return e->fail(BPF_ERR_EXCHANGE, rv.error);
    if (agreed)
    {
      ok = bpf_shard_mailbox_connect(e, all.data()) == BPF_OK ? 1 : 0;
      if (!rv.all_min(ok, &agreed, timeout_ms))
        return e->fail(BPF_ERR_EXCHANGE, rv.error);
    }
    if (agreed)
    {
      // the words arrive; do the window cells?  (a peer's stores must be visible behind this GPU's caches)
      ok = bpf_shard_mailbox_selftest(e, 4) == BPF_OK ? 1 : 0;
      if (!rv.all_min(ok, &agreed, timeout_ms))
        return e->fail(BPF_ERR_EXCHANGE, rv.error);
    }
    if (!agreed)
      (void)bpf_shard_mailbox_destroy(e);
  }
  if (agreed)
  {
    if (mode_out)
      *mode_out = BPF_SHARD_EXCHANGE_MAILBOX;
    return BPF_OK;
  }
  if (flags & BPF_BOOTSTRAP_MAILBOX_ONLY)
    return e->fail(BPF_ERR_EXCHANGE, "bootstrap: the mailbox exchange could not be set up on every rank");
  // ---- RCCL: rank 0's unique id over the sockets, then the communicator
  ok = collective_load(e) == BPF_OK ? 1 : 0;
  if (!rv.all_min(ok, &agreed, timeout_ms))
    return e->fail(BPF_ERR_EXCHANGE, rv.error);
  if (!agreed)
    return ok ? e->fail(BPF_ERR_EXCHANGE, "bootstrap: a peer could not load the RCCL collectives") : e->last_status;
  const int id_bytes = e->coll.fn.unique_id_bytes();
  std::vector<unsigned char> id((size_t)id_bytes, 0), ids;
  if (rank == 0 && e->coll.fn.unique_id(id.data()) != 0)
    return e->fail(BPF_ERR_EXCHANGE, std::string("bootstrap: ") + e->coll.fn.last_error());
  if (!rv.all_gather(id.data(), id.size(), &ids, timeout_ms))
    return e->fail(BPF_ERR_EXCHANGE, rv.error);
  ok = e->coll.fn.init(&e->coll.comm, rank, world, ids.data()) == 0 ? 1 : 0;  // rank 0's id leads the gathered block
  const std::string init_error = ok ? std::string() : std::string(e->coll.fn.last_error());
  if (!rv.all_min(ok, &agreed, timeout_ms))
    return e->fail(BPF_ERR_EXCHANGE, rv.error);
  if (!agreed)
  {
    collective_release(e);
    return e->fail(BPF_ERR_EXCHANGE, "bootstrap: RCCL communicator: " + (ok ? std::string("a peer failed") : init_error));
  }
  e->coll.active = true;
  if (mode_out)
    *mode_out = BPF_SHARD_EXCHANGE_RCCL;
  return BPF_OK;
}

int bpf_shard_shutdown(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(e->device);
  collective_release(e);
  return bpf_shard_mailbox_destroy(e);
}

int bpf_shard_update_sensor_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                   double range_max, long long global_count)
{
  return bpf_shard_mailbox_update_sensor_planar(e, ranges, angles, range_count, range_max, global_count);
}

int bpf_shard_update_resample(bpf_engine* e, int* global_count_io, int* leaf_count_io, int* bin_count_out,
                              int* windows_out, int* window_hint_io, int* cdf_miss_out)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->d_shard_flags.p)
  {
    HIPCHK(e, e->d_shard_flags.reserve(4));
    HIPCHK(e, hipMemsetAsync(e->d_shard_flags.p, 0, 4 * sizeof(int), e->stream));
  }
  int rc = bpf_shard_mailbox_update_resample(e, e->d_shard_flags.p, global_count_io, leaf_count_io, bin_count_out,
                                             windows_out, window_hint_io);
  if (rc != BPF_OK)
    return rc;
  if (cdf_miss_out)
  {
    int flag = 0;
    HIPCHK(e, hipMemcpyAsync(&flag, e->d_shard_flags.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    *cdf_miss_out = flag;
  }
  return BPF_OK;
}
