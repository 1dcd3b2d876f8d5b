// C-ABI: mailbox exchange of the sharded path (kernels_mailbox.hpp): creation, IPC hand-shake, teardown.
namespace
{
void mailbox_release(bpf_engine* e)
{
  bpf_engine::Mailbox& m = e->mb;
  for (int r = 0; r < kMailboxMaxWorld; ++r)
  {
    if (m.opened[r] && m.peer[r])
      (void)hipIpcCloseMemHandle(m.peer[r]);
    m.opened[r] = false;
    m.peer[r] = nullptr;
  }
  if (m.own)
    (void)hipFree(m.own);
  m = bpf_engine::Mailbox{};
}
}  // namespace

int bpf_shard_mailbox_create(bpf_engine* e, int rank, int world, long long max_window, void* handle_out)
{
  if (!e || !handle_out || world < 1 || world > kMailboxMaxWorld || rank < 0 || rank >= world || max_window < 1)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "mailbox: 1 <= world <= 16, 0 <= rank < world, max_window >= 1")
             : BPF_ERR_INVALID_ARGUMENT;
  static_assert(sizeof(hipIpcMemHandle_t) == BPF_MAILBOX_HANDLE_BYTES, "IPC handle size");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  mailbox_release(e);
  int rc = ensure_scalars(e);
  if (rc != BPF_OK)
    return rc;
  bpf_engine::Mailbox& m = e->mb;
  m.bytes = kMailboxHeader + (size_t)2 * 6 * (size_t)max_window * sizeof(long long);
  // uncached: a peer's stores arrive over xGMI behind this GPU's L2, so nothing of the mailbox may live in it
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&m.own), m.bytes, hipDeviceMallocUncached) != hipSuccess)
  {
    (void)hipGetLastError();
    m.own = nullptr;
    HIPCHK(e, hipExtMallocWithFlags(reinterpret_cast<void**>(&m.own), m.bytes, hipDeviceMallocFinegrained));
  }
  HIPCHK(e, hipMemsetAsync(m.own, 0, m.bytes, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  hipIpcMemHandle_t h;
  HIPCHK(e, hipIpcGetMemHandle(&h, m.own));
  std::memcpy(handle_out, &h, sizeof(h));
  m.rank = rank;
  m.world = world;
  m.max_window = max_window;
  e->shard_rank = rank;
  e->shard_world = world;
  HIPCHK(e, e->h_mb_error.reserve(2));
  e->h_mb_error.p[0] = e->h_mb_error.p[1] = 0;
  HIPCHK(e, e->d_mb_error.reserve(2));
  HIPCHK(e, hipMemsetAsync(e->d_mb_error.p, 0, 2 * sizeof(unsigned), e->stream));
  HIPCHK(e, e->h_mb_result.reserve(1));
  HIPCHK(e, e->d_mb_counter.reserve(1));
  HIPCHK(e, hipMemsetAsync(e->d_mb_counter.p, 0, sizeof(unsigned), e->stream));
  return BPF_OK;
}

namespace
{
MailboxDev mailbox_dev(const bpf_engine* e)
{
  MailboxDev M{};
  if (!e->mb.active)
    return M;  // world == 0: kernels skip the exchange
  M.rank = e->mb.rank;
  M.world = e->mb.world;
  M.max_window = e->mb.max_window;
  for (int r = 0; r < e->mb.world; ++r)
    M.peer[r] = e->mb.peer[r];
  M.host_error = e->h_mb_error.p;
  M.dev_error = e->d_mb_error.p;
  M.timeout_ticks = (long long)e->mb_timeout_ms * 100000ll;
  return M;
}

// one full post-and-wait round over all peers (every rank has to be in it)
int mailbox_hello(bpf_engine* e)
{
  e->h_mb_result.p[0] = -1;
  const unsigned long long token = ++e->mb.hello;
  hipLaunchKernelGGL(k_mailbox_hello, dim3(1), dim3(64), 0, e->stream, mailbox_dev(e), token, e->h_mb_result.p);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (e->h_mb_result.p[0] != 1)
    return e->fail(BPF_ERR_EXCHANGE, "mailbox: a peer's word did not arrive in time");
  return BPF_OK;
}

bool mailbox_owns(const bpf_engine* e, const void* p)
{
  const char* c = static_cast<const char*>(p);
  return e->mb.active && c >= e->mb.own && c < e->mb.own + e->mb.bytes;
}

int mailbox_check(bpf_engine* e)
{
  if (e->mb.active && (__atomic_load_n(e->h_mb_error.p, __ATOMIC_ACQUIRE) != 0 ||
                       __atomic_load_n(e->h_mb_error.p + 1, __ATOMIC_ACQUIRE) != 0))
    return e->fail(BPF_ERR_EXCHANGE, "mailbox: a wait for a peer ran out of time; the shards are out of step "
                                     "(bpf_shard_mailbox_error_stage tells which exchange)");
  return BPF_OK;
}
}  // namespace

int bpf_shard_mailbox_set_timeout_ms(bpf_engine* e, int timeout_ms)
{
  if (!e || timeout_ms < 1 || timeout_ms > 600000)
    return BPF_ERR_INVALID_ARGUMENT;
  e->mb_timeout_ms = timeout_ms;
  return BPF_OK;
}

int bpf_shard_mailbox_error_stage(bpf_engine* e, int* totals_failed, int* window_failed)
{
  if (!e || !totals_failed || !window_failed)
    return BPF_ERR_INVALID_ARGUMENT;
  *totals_failed = *window_failed = 0;
  if (e->h_mb_error.p)
  {
    // the flags are raised by kernels: let the stream drain so that a wait still spinning has decided
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    *totals_failed = __atomic_load_n(e->h_mb_error.p, __ATOMIC_ACQUIRE) != 0;
    *window_failed = __atomic_load_n(e->h_mb_error.p + 1, __ATOMIC_ACQUIRE) != 0;
  }
  return BPF_OK;
}

int bpf_shard_mailbox_connect(bpf_engine* e, const void* handles)
{
  if (!e || !handles)
    return BPF_ERR_INVALID_ARGUMENT;
  bpf_engine::Mailbox& m = e->mb;
  if (!m.own)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_shard_mailbox_create first");
  HIPCHK(e, hipSetDevice(e->device));
  const char* hs = static_cast<const char*>(handles);
  for (int r = 0; r < m.world; ++r)
  {
    if (r == m.rank)
    {
      m.peer[r] = m.own;
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, hs + (size_t)r * BPF_MAILBOX_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    HIPCHK(e, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    m.peer[r] = static_cast<char*>(p);
    m.opened[r] = true;
  }
  m.active = true;
  int rc = mailbox_hello(e);
  if (rc != BPF_OK)
    m.active = false;
  return rc;
}

int bpf_shard_mailbox_selftest(bpf_engine* e, int rounds)
{
  if (!e || rounds < 1 || rounds > 64)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->mb.active)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "mailbox not connected");
  if (e->mb.win_wait)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "mailbox: an exchange is in flight");
  HIPCHK(e, hipSetDevice(e->device));
  const int n_cols = (int)std::min<long long>(e->mb.max_window, 4096);
  const int blocks = blocks_for(n_cols, 256);  // <= 16 blocks: the check may wait with all of them
  for (int r = 0; r < rounds; ++r)
  {
    const unsigned long long g = ++e->mb.win_gen;
    e->h_mb_result.p[0] = 0;
    hipLaunchKernelGGL(k_mailbox_selftest_write, dim3(blocks), dim3(256), 0, e->stream, mailbox_dev(e), n_cols,
                       (int)(g & 1), g, e->d_mb_counter.p);
    hipLaunchKernelGGL(k_mailbox_selftest_check, dim3(blocks), dim3(256), 0, e->stream, mailbox_dev(e), n_cols,
                       (int)(g & 1), g, e->h_mb_result.p);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (int rc = mailbox_check(e))
      return rc;
    if (e->h_mb_result.p[0] != 0)
      return e->fail(BPF_ERR_EXCHANGE, "mailbox self-test: window cells did not arrive as their owners wrote them");
  }
  return BPF_OK;
}

int bpf_shard_mailbox_destroy(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  mailbox_release(e);
  return BPF_OK;
}

int bpf_shard_mailbox_totals(bpf_engine* e, void** totals_dev)
{
  if (!e || !totals_dev)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->mb.active || e->mb.tot_gen == 0)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "mailbox: no totals posted yet");
  *totals_dev = e->mb.own + 256 + (size_t)(e->mb.tot_gen & 1) * kMailboxMaxWorld * sizeof(double);
  return BPF_OK;
}

int bpf_shard_mailbox_window(bpf_engine* e, void** window_dev, int* stride)
{
  if (!e || !window_dev || !stride)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->mb.active)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "mailbox not connected");
  if (e->mb.win_wait)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "mailbox: the previous window was never consumed");
  const unsigned long long g = ++e->mb.win_gen;
  *window_dev = e->mb.own + kMailboxHeader + (size_t)(g & 1) * 6 * (size_t)e->mb.max_window * sizeof(long long);
  *stride = (int)e->mb.max_window;
  return BPF_OK;
}
