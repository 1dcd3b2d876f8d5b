// C-ABI: the sharded sensor update and resample as single calls for the mailbox mode.  With the exchanges inside the
// kernels there is nothing left for a host-side collective layer to do between the stage functions, so the whole
// sequence of badger_amcl_amd/sharded.py runs here, stage function by stage function (same order, same arguments),
// and the host pays one call per update instead of a dozen.
//
// The same sequence also runs with RCCL collectives between the stages (bpf_shard_bootstrap falls back to that when the
// mailbox cannot be set up): the totals go through ncclAllGather after the scoring stage and every draw window through
// an integer ncclAllReduce(sum) before its first consumer -- ShardExchange hides which of the two it is.
namespace
{
struct ShardExchange
{
  bpf_engine* e;
  bool collective() const { return !e->mb.active; }

  // a fresh [6][stride] int64 window for `count` draws
  long long* next_window(int count, int* stride)
  {
    if (!collective())
    {
      void* w = nullptr;
      if (bpf_shard_mailbox_window(e, &w, stride) != BPF_OK)
        return nullptr;
      return static_cast<long long*>(w);
    }
    // two buffers in turn: the previous window stays readable while the next one is assembled
    DevBuf<long long>& buf = e->coll.window[e->coll.window_turn ^= 1];
    const size_t cols = (size_t)((count + 255) / 256) * 256;
    if (buf.reserve(6 * cols) != hipSuccess)
    {
      e->fail(BPF_ERR_HIP, "collective window allocation");
      return nullptr;
    }
    *stride = (int)cols;
    return buf.p;
  }

  // every shard's columns into every shard's copy of the window
  int assemble(long long* window, int stride)
  {
    if (!collective())
      return BPF_OK;  // the draw kernel stored them into all peers; the window's first consumer waits for them
    if (e->coll.fn.allreduce_sum_i64(e->coll.comm, window, (size_t)6 * stride, e->stream) != 0)
      return e->fail(BPF_ERR_EXCHANGE, std::string("RCCL window all-reduce: ") + e->coll.fn.last_error());
    return BPF_OK;
  }

  // the W weight totals of the scoring stage just issued
  int totals(void** out)
  {
    if (!collective())
      return bpf_shard_mailbox_totals(e, out);
    HIPCHK(e, e->coll.totals.reserve((size_t)e->shard_world));
    if (e->coll.fn.allgather_f64(e->coll.comm, &e->d_scalars.p->v[0], e->coll.totals.p, 1, e->stream) != 0)
      return e->fail(BPF_ERR_EXCHANGE, std::string("RCCL totals all-gather: ") + e->coll.fn.last_error());
    *out = e->coll.totals.p;
    return BPF_OK;
  }
};

int shard_step_ready(bpf_engine* e)
{
  if (!e->mb.active && !e->coll.active)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no shard exchange set up (bpf_shard_bootstrap / bpf_shard_mailbox_connect)");
  return BPF_OK;
}
}  // namespace

int bpf_shard_mailbox_update_sensor_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                           double range_max, long long global_count)
{
  if (!e || global_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  int rc = shard_step_ready(e);
  if (rc != BPF_OK)
    return rc;
  ShardExchange X{ e };
  e->mb_totals_valid = false;
  rc = bpf_shard_score_planar(e, ranges, angles, range_count, range_max);
  if (rc == BPF_SHARD_NEED_BEAM_COUNTS && X.collective())
  {
    // beam skipping: the per-beam agreement counts are summed over the shards between the two passes
    void* counts = nullptr;
    int n_counts = 0;
    rc = bpf_shard_beam_counts_dev(e, &counts, &n_counts);
    if (rc != BPF_OK)
      return rc;
    if (e->coll.fn.allreduce_sum_i32(e->coll.comm, static_cast<int*>(counts), (size_t)n_counts, e->stream) != 0)
      return e->fail(BPF_ERR_EXCHANGE, std::string("RCCL beam-count all-reduce: ") + e->coll.fn.last_error());
    rc = bpf_shard_score_planar_finish(e, ranges, angles, range_count, range_max, global_count);
  }
  if (rc != BPF_OK)
    return rc;  // includes BPF_SHARD_NEED_BEAM_COUNTS (mailbox mode): the caller sums the counts in between
  if (e->pm.max_beams < 2)
    return BPF_OK;
  void* totals = nullptr;
  rc = X.totals(&totals);
  if (rc != BPF_OK)
    return rc;
  rc = bpf_shard_normalize_dev(e, totals, e->shard_world, (int)global_count);
  if (rc != BPF_OK)
    return rc;
  e->mb_totals = totals;
  e->mb_totals_valid = true;
  return BPF_OK;
}

int bpf_shard_mailbox_update_resample(bpf_engine* e, void* flags_dev, int* global_count_io, int* leaf_count_io,
                                      int* bin_count_out, int* windows_out, int* window_hint_io)
{
  if (!e || !flags_dev || !global_count_io || !leaf_count_io || !bin_count_out || !windows_out || !window_hint_io)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  int rc = shard_step_ready(e);
  if (rc != BPF_OK)
    return rc;
  if (!e->mb_totals_valid)
    return e->fail(BPF_ERR_NOT_CONFIGURED,
                   "sharded resample: needs the totals of bpf_shard_mailbox_update_sensor_planar of this update");
  HIPCHK(e, hipSetDevice(e->device));
  ShardExchange X{ e };
  const int rank = e->shard_rank, W = e->shard_world;
  const int max_global = e->max_samples;  // engines of a sharded filter carry the GLOBAL bounds
  if (!X.collective() && max_global > e->mb.max_window)
    return e->fail(BPF_ERR_CAPACITY, "mailbox windows are smaller than max_samples");
  rc = bpf_shard_build_cdf(e, flags_dev);
  if (rc != BPF_OK)
    return rc;
  const uint64_t rng = e->rng;
  double w_diff = 0.0;
  int sys_count = 0;
  rc = bpf_shard_begin_resample(e, rng, *leaf_count_io, &w_diff, &sys_count);
  if (rc != BPF_OK)
    return rc;
  int stride = 0;
  int M = 0, leaf = 0, bins = 0;
  *windows_out = 0;
  auto rows = [&](long long* window, int k) { return reinterpret_cast<const double*>(window + (size_t)k * stride); };
  auto adopt_from = [&](const double* x, const double* y, const double* th) -> int {
    const int lo = (int)(((long long)M * rank) / W), hi = (int)(((long long)M * (rank + 1)) / W);
    if (M <= 8192)
      return bpf_shard_tail_small_dev(e, x, y, th, M, lo, hi, leaf, bins);
    int r2 = bpf_shard_adopt_dev(e, x + lo, y + lo, th + lo, hi - lo, M, leaf, bins);
    if (r2 != BPF_OK)
      return r2;
    return bpf_shard_converged_dev(e, x, y, M);
  };
  if (e->resample_model == BPF_RESAMPLE_SYSTEMATIC)
  {
    long long* window = X.next_window(sys_count, &stride);
    if (!window)
      return e->last_status;
    const bool tracking = e->fused_resample && sys_count <= kFusedWindow;
    const bool merged = tracking && !X.collective();  // mailbox: the draws and their consumer in ONE launch
    WindowArgs draw{};
    if (merged)
    {
      size_t lds_unused = 0;
      rc = systematic_window_args(e, rng, sys_count, e->mb_totals, 1, rank, W, window, stride, flags_dev, &draw,
                                  &lds_unused);
    }
    else
      rc = bpf_shard_systematic_window_dev(e, rng, sys_count, e->mb_totals, 1, rank, W, window, stride, flags_dev);
    if (rc != BPF_OK)
      return rc;
    rc = X.assemble(window, stride);
    if (rc != BPF_OK)
      return rc;
    *windows_out = 1;
    int fused_status = BPF_FUSED_TOO_MANY_BINS;
    if (tracking)
    {
      // tree of the new set (every sample, no stop rule), adoption and updateConverged in one launch
      rc = shard_stop_block(e, window, stride, sys_count, true, &fused_status, &M, &leaf, &bins,
                            merged ? &draw : nullptr);
      if (rc == BPF_OK && merged)
        rc = systematic_targets_in_use(e);
      if (rc != BPF_OK)
        return rc;
    }
    if (fused_status != BPF_FUSED_OK)
    {
      bpf_kld_reset(e);
      rc = bpf_kld_insert_dev(e, window, stride, sys_count);  // the host's tree
      if (rc != BPF_OK)
        return rc;
      bpf_kld_leaf_count(e, &leaf, &bins);
      M = sys_count;
      rc = adopt_from(rows(window, 0), rows(window, 1), rows(window, 2));
      if (rc != BPF_OK)
        return rc;
    }
  }
  else
  {
    bpf_kld_reset(e);
    int m0 = 0, stop = -1, need = 0;
    int win = std::max(1024, std::min(*window_hint_io, max_global));
    const int device_min = e->kld_device_min;
    bool device_declined = false;
    if (win > 4096 && max_global - 4096 >= device_min)
      win = 4096;  // keep the host's first window short when the device tree can take over after it
    long long* last_window = nullptr;
    int n_windows = 0;
    bool copied_any = false;
    bool have_counts = false;
    bool adopted = false;  // k_shard_stop_block has done the stop rule and the tail
    while (m0 < max_global && stop < 0)
    {
      if (m0 > 0 && !device_declined && max_global - m0 >= device_min && need >= device_min)
      {
        // no stop so far and a long stream ahead (a spread cloud): one window with every candidate, and the ordered
        // kd-tree replay runs on the device (every rank, redundantly)
        long long* whole = X.next_window(max_global, &stride);
        if (!whole)
          return e->last_status;
        rc = bpf_shard_draw_window_dev(e, rng, 0, max_global, e->mb_totals, 1, rank, W, whole, stride, flags_dev);
        if (rc != BPF_OK)
          return rc;
        rc = X.assemble(whole, stride);
        if (rc != BPF_OK)
          return rc;
        int handled = 0, dstop = -1, dleaf = 0, dbins = 0;
        rc = bpf_kld_stop_dev(e, whole, stride, max_global, &handled, &dstop, &dleaf, &dbins);
        if (rc != BPF_OK)
          return rc;
        ++*windows_out;
        if (handled)
        {
          stop = dstop;
          leaf = dleaf;
          bins = dbins;
          have_counts = true;
          last_window = whole;
          n_windows = 1;
          copied_any = false;
          m0 = 0;
          break;
        }
        device_declined = true;  // this stream is outside what the device tree takes: host replay
      }
      const int m1 = std::min(max_global, m0 + win), cnt = m1 - m0;
      long long* window = X.next_window(cnt, &stride);
      if (!window)
        return e->last_status;
      const bool tracking = m0 == 0 && e->fused_resample && cnt <= kFusedWindow && *window_hint_io <= kFusedWindow;
      WindowArgs draw{};
      const bool merged = tracking && !X.collective();  // mailbox: the draws and their consumer in ONE launch
      if (merged)
      {
        size_t lds_unused = 0;
        rc = draw_window_args(e, rng, m0, m1, e->mb_totals, 1, rank, W, window, stride, flags_dev, &draw, &lds_unused);
      }
      else
        rc = bpf_shard_draw_window_dev(e, rng, m0, m1, e->mb_totals, 1, rank, W, window, stride, flags_dev);
      if (rc != BPF_OK)
        return rc;
      rc = X.assemble(window, stride);
      if (rc != BPF_OK)
        return rc;
      if (tracking)
      {
        // the tracking regime: the stream is expected to stop inside this first window, and the stop rule, the
        // adoption of this rank's share and updateConverged run in one single-block launch on every rank
        int fused_status = BPF_FUSED_TOO_MANY_BINS;
        rc = shard_stop_block(e, window, stride, cnt, false, &fused_status, &M, &leaf, &bins, merged ? &draw : nullptr);
        if (rc != BPF_OK)
          return rc;
        if (fused_status == BPF_FUSED_OK)
        {
          ++*windows_out;
          adopted = true;
          break;
        }
        // no stop in the window, or keys / bins beyond what the kernel takes: the host replays this same window
      }
      rc = bpf_kld_feed_dev(e, window, stride, cnt, m0, &stop);  // the one host wait of the window
      if (rc != BPF_OK)
        return rc;
      ++*windows_out;
      ++n_windows;
      last_window = window;
      if (stop < 0 || n_windows > 1)
      {
        // a mailbox window is overwritten two exchanges later, and a set that spans windows is adopted from one
        // buffer: keep this window's poses
        HIPCHK(e, e->d_shard_out.reserve((size_t)3 * max_global));
        for (int k = 0; k < 3; ++k)
          HIPCHK(e, hipMemcpyAsync(e->d_shard_out.p + (size_t)k * max_global + m0, rows(window, k),
                                   (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        copied_any = true;
      }
      m0 = m1;
      if (stop < 0)
      {
        // next window: up to a quarter past the bound for the leaves seen so far (a lower estimate of the stop)
        int lc = 0, bc = 0;
        bpf_kld_leaf_count(e, &lc, &bc);
        int lim = 0;
        bpf_pf_resample_limit(e, lc, &lim);
        need = lim - m0;
        win = std::max(1024, (need + need / 4 + 1023) / 1024 * 1024);
      }
    }
    if (!adopted)
    {
      M = stop > 0 ? stop : max_global;
      if (!have_counts)
        bpf_kld_leaf_count(e, &leaf, &bins);
      if (copied_any)
        rc = adopt_from(e->d_shard_out.p, e->d_shard_out.p + max_global, e->d_shard_out.p + 2 * (size_t)max_global);
      else
        rc = adopt_from(rows(last_window, 0), rows(last_window, 1), rows(last_window, 2));
      if (rc != BPF_OK)
        return rc;
    }
    *window_hint_io = std::max(1024, ((M + M / 4) + 1023) / 1024 * 1024);
  }
  uint64_t rng_after = 0;
  rc = bpf_shard_end_resample(e, M, &rng_after);
  if (rc != BPF_OK)
    return rc;
  e->rng = rng_after;
  e->mb_totals_valid = false;  // the weights are 1/M now
  *global_count_io = M;
  *leaf_count_io = leaf;
  *bin_count_out = bins;
  return BPF_OK;
}
