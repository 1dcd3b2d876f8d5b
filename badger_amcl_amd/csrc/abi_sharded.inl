// C-ABI: sharded stages.
// ---------------------------------------------------------------------- sharded stages
namespace
{
// two-launch sum of the weights into scalars[0]; with a mailbox a third, tiny launch posts it to the peers
int shard_sum_and_post(bpf_engine* e, const double* w, int n)
{
  int rc = sum_into_slot(e, w, n, 0, 0, n);
  if (rc != BPF_OK || !e->mb.active)
    return rc;
  const unsigned long long gen = ++e->mb.tot_gen;
  hipLaunchKernelGGL(k_mailbox_post_total, dim3(1), dim3(64), 0, e->stream, &e->d_scalars.p->v[0], mailbox_dev(e),
                     (int)(gen & 1), gen);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

// local weight total into scalars[0] after a sharded scoring stage
int shard_local_total(bpf_engine* e)
{
  SampleSet& s = e->sets[e->cur];
  if (e->fused_partials > 0)
  {
    // the scoring kernel left per-block partials: one small launch folds them into the local total
    if (e->mb.active)
    {
      // mailbox: the fold and the post to the peers ride on the normalise launch that has to follow anyway
      // (bpf_shard_normalize_dev with bpf_shard_mailbox_totals); one launch less on the critical path
      ++e->mb.tot_gen;
      e->mb.fold_deferred = e->fused_partials;
      e->fused_partials = 0;
      return BPF_OK;
    }
    const unsigned long long gen = 0;
    ProfScope ps(e, BPF_K_REDUCE);
    hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(BPF_RED_BLOCK), 0, e->stream, e->d_block_partials.p,
                       e->fused_partials, e->d_scalars.p, 0, mailbox_dev(e), (int)(gen & 1), gen);
    HIPCHK(e, hipGetLastError());
    e->fused_partials = 0;
    return BPF_OK;
  }
  return shard_sum_and_post(e, s.w.p, e->sample_count);
}
}  // namespace

int bpf_shard_score_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                           double range_max)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (e->pm.max_beams < 2)
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  bool forced_zero = false;
  e->skip_pending = false;
  int rc = score_planar(e, s.dev(), e->sample_count, e->converged, ranges, angles, range_count, range_max,
                        &forced_zero, true, true);
  if (rc != BPF_OK)
    return rc;
  if (e->skip_pending)
    return BPF_SHARD_NEED_BEAM_COUNTS;  // sum bpf_shard_beam_counts_dev over the shards, then ..._finish
  return shard_local_total(e);
}

int bpf_shard_beam_counts_dev(bpf_engine* e, void** counts_dev, int* n_counts)
{
  if (!e || !counts_dev || !n_counts)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->skip_pending)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no beam-skip counting pass is pending");
  *counts_dev = e->d_obs_count.p;
  *n_counts = std::max(e->skip_fs.n_staged, 1);
  return BPF_OK;
}

int bpf_shard_score_planar_finish(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                  double range_max, long long global_count)
{
  if (!e || !ranges || !angles || global_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf || !e->skip_pending)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no beam-skip counting pass is pending");
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  bool forced_zero = false;
  int rc = score_planar_beamskip_finish(e, s.dev(), e->sample_count, global_count, ranges, angles, range_count,
                                        range_max, &forced_zero, true);
  if (rc != BPF_OK)
    return rc;
  return shard_local_total(e);
}

int bpf_shard_score_cloud(bpf_engine* e, const float* points_xyz, int n_points)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (e->cloud_max_beams < 2)
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  int rc = score_cloud(e, s.dev(), e->sample_count, points_xyz, n_points);
  if (rc != BPF_OK)
    return rc;
  return shard_sum_and_post(e, s.w.p, e->sample_count);
}

int bpf_shard_scalars_dev(bpf_engine* e, void** dev_ptr)
{
  if (!e || !dev_ptr)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  int rc = ensure_scalars(e);
  if (rc != BPF_OK)
    return rc;
  *dev_ptr = e->d_scalars.p;
  return BPF_OK;
}

int bpf_shard_normalize_dev(bpf_engine* e, const void* totals_dev, int world, int global_sample_count)
{
  if (!e || !e->have_pf || !totals_dev || world <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  const int n = e->sample_count;
  const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
  HIPCHK(e, e->d_tile_sums.reserve((size_t)nb));
  // totals that are this engine's mailbox slots: the kernel itself waits for the peers' posts of this update
  MailboxDev wait{};
  const double* fold = nullptr;
  int n_fold = 0;
  if (mailbox_owns(e, totals_dev))
  {
    if (world != e->mb.world)
      return e->fail(BPF_ERR_INVALID_ARGUMENT, "mailbox totals: world differs from the mailbox's");
    if (nb <= kMailboxFusedWaitBlocks)
    {
      wait = mailbox_dev(e);
      if (e->mb.fold_deferred > 0)
      {
        fold = e->d_block_partials.p;
        n_fold = e->mb.fold_deferred;
        e->mb.fold_deferred = 0;
      }
    }
    else
    {
      // a large normalise grid does not spin: fold + post and the wait get small launches of their own
      if (e->mb.fold_deferred > 0)
        hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(BPF_RED_BLOCK), 0, e->stream, e->d_block_partials.p,
                           e->mb.fold_deferred, e->d_scalars.p, 0, mailbox_dev(e), (int)(e->mb.tot_gen & 1),
                           e->mb.tot_gen);
      e->mb.fold_deferred = 0;
      hipLaunchKernelGGL(k_mailbox_wait, dim3(1), dim3(64), 0, e->stream, mailbox_dev(e), 0, (int)(e->mb.tot_gen & 1),
                         e->mb.tot_gen);
      HIPCHK(e, hipGetLastError());
    }
  }
  else if (e->mb.fold_deferred > 0)
  {
    // the caller normalises with totals of its own: the peers still expect this rank's post of the update
    hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(BPF_RED_BLOCK), 0, e->stream, e->d_block_partials.p,
                       e->mb.fold_deferred, e->d_scalars.p, 0, mailbox_dev(e), (int)(e->mb.tot_gen & 1), e->mb.tot_gen);
    HIPCHK(e, hipGetLastError());
    e->mb.fold_deferred = 0;
  }
  ProfScope ps(e, BPF_K_NORMALIZE);
  if (wait.world > 0 && e->fused_resample && !e->cdf_serial)
  {
    // mailbox mode with a grid small enough to wait in-kernel: the local CDF comes out of the same launch
    // (k_normalize_gathered_cdf), so the resample that follows starts with its draws
    int rcb = ensure_cdf_buffers(e, n);
    if (rcb != BPF_OK)
      return rcb;
    if (e->d_tile_slots.cap < (size_t)2 * BPF_RED_BLOCK)
    {
      HIPCHK(e, e->d_tile_slots.reserve((size_t)2 * BPF_RED_BLOCK));
      HIPCHK(e, hipMemsetAsync(e->d_tile_slots.p, 0xFF, 2 * BPF_RED_BLOCK * sizeof(unsigned long long), e->stream));
      e->tile_generation = 0;
    }
    HIPCHK(e, e->d_cdf_coarse.reserve((size_t)kFusedCoarse + 2));
    NormCdfGatherArgs G{};
    G.n.w = s.w.p;
    G.n.n = n;
    G.n.block_partials = fold;
    G.n.n_partials = n_fold;
    G.n.sc = e->d_scalars.p;
    G.n.alpha_slow = e->alpha_slow;
    G.n.alpha_fast = e->alpha_fast;
    G.n.tile_slots = e->d_tile_slots.p;
    G.n.generation = ++e->tile_generation;  // (only its parity matters)
    G.n.cdf = e->d_cdf.p;
    G.n.coarse = e->d_cdf_coarse.p;
    G.n.coarse_shift = fused_coarse_shift(n);
    G.n.guide = nullptr;  // k_draw_window bisects the CDF itself
    G.n.zero_word = e->d_flags.p;
    G.totals = static_cast<const double*>(totals_dev);
    G.world = world;
    G.global_n = global_sample_count;
    G.mb = wait;
    G.wait_parity = (int)(e->mb.tot_gen & 1);
    G.wait_gen = e->mb.tot_gen;
    G.zero_word2 = static_cast<int*>(e->shard_flags_last);
    G.sum_out = &e->d_scalars.p->v[7];
    hipLaunchKernelGGL(k_normalize_gathered_cdf, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, G);
    HIPCHK(e, hipGetLastError());
    e->tile_sums_n = -1;
    e->cdf_ready_n = n;
    e->cdf_coarse_n = n;
    e->cdf_guide_valid = false;
    e->shard_cdf_flags = e->shard_flags_last;
    e->shard_cdf_valid = true;
    return BPF_OK;
  }
  e->shard_cdf_valid = false;
  hipLaunchKernelGGL(k_normalize_gathered, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, s.w.p, n,
                     static_cast<const double*>(totals_dev), world, global_sample_count, e->d_scalars.p,
                     e->alpha_slow, e->alpha_fast, e->d_tile_sums.p, wait, (int)(e->mb.tot_gen & 1),
                     e->mb.tot_gen, fold, n_fold,
                     (mailbox_owns(e, totals_dev) && e->d_mb_error.p) ? (const unsigned*)e->d_mb_error.p : nullptr);
  HIPCHK(e, hipGetLastError());
  e->tile_sums_n = n;
  return BPF_OK;
}

int bpf_shard_build_cdf(bpf_engine* e, void* flags_dev)
{
  if (!e || !e->have_pf)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  if (e->shard_cdf_valid && e->cdf_ready_n == e->sample_count && !e->cdf_serial)
  {
    // k_normalize_gathered_cdf of this update left the CDF and its sum behind; it cleared the caller's miss flag too
    // if this is the word the previous resample used
    e->shard_cdf_valid = false;
    if (e->shard_cdf_flags != flags_dev && flags_dev)
      HIPCHK(e, hipMemsetAsync(flags_dev, 0, sizeof(int), e->stream));
    e->shard_flags_last = flags_dev;
    return BPF_OK;
  }
  e->shard_flags_last = flags_dev;
  // the scan clears the caller's miss flag and leaves the local CDF sum in scalars[7] itself
  return build_cdf(e, e->sets[e->cur].w.p, e->sample_count, static_cast<int*>(flags_dev), &e->d_scalars.p->v[7]);
}

namespace
{
// A window handed out by bpf_shard_mailbox_window: the draw kernel stores every column it owns into all peers'
// copies of that window and posts "done"; the window's first consumer kernel waits for every shard's word.
int shard_window_exchange(bpf_engine* e, const void* window_dev, int count, int world, WindowArgs* A)
{
  if (!mailbox_owns(e, window_dev))
    return BPF_OK;
  const unsigned long long g = e->mb.win_gen;
  const char* expect = e->mb.own + kMailboxHeader + (size_t)(g & 1) * 6 * (size_t)e->mb.max_window * sizeof(long long);
  if (window_dev != expect || e->mb.win_wait)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "mailbox window: take a fresh one from bpf_shard_mailbox_window per exchange");
  if (count > e->mb.max_window || world != e->mb.world)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "mailbox window: count beyond max_window, or another world size");
  A->mb = mailbox_dev(e);
  A->mb_parity = (int)(g & 1);
  A->mb_gen = g;
  A->mb_counter = e->d_mb_counter.p;
  e->mb.win_wait = true;
  return BPF_OK;
}

// wait arguments for the first kernel that reads an exchanged window (world 0: nothing to wait for); a consumer
// with a large grid gets a one-block wait kernel in front instead of spinning with all its blocks
MailboxDev shard_window_wait(bpf_engine* e, const void* window_dev, int consumer_blocks)
{
  if (mailbox_owns(e, window_dev) && e->mb.win_wait)
  {
    e->mb.win_wait = false;
    if (consumer_blocks <= kMailboxFusedWaitBlocks)
      return mailbox_dev(e);
    hipLaunchKernelGGL(k_mailbox_wait, dim3(1), dim3(64), 0, e->stream, mailbox_dev(e), 1, (int)(e->mb.win_gen & 1),
                       e->mb.win_gen);
  }
  return MailboxDev{};
}
}  // namespace

namespace
{
// the arguments of a multinomial draw window (k_draw_window, k_shard_resample_block); *lds_out = the dynamic LDS the
// CDF subsample takes in k_draw_window (0: none)
int draw_window_args(bpf_engine* e, uint64_t rng_state48, int m0, int m1, const void* sums_dev, int sums_are_totals,
                     int rank, int world, void* window_dev, int stride, void* flags_dev, WindowArgs* out, size_t* lds_out)
{
  if (!e || !e->have_pf || !sums_dev || !window_dev || !flags_dev || m1 <= m0 || stride < m1 - m0 || rank < 0 ||
      rank >= world)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad draw window arguments") : BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  WindowArgs& A = *out;
  A = WindowArgs{};
  A.src = e->sets[e->cur].dev();
  A.n_src = e->sample_count;
  A.cdf = e->d_cdf.p;
  A.sums = static_cast<const double*>(sums_dev);
  A.sums_are_totals = sums_are_totals;
  A.rank = rank;
  A.world = world;
  A.m0 = m0;
  A.m1 = m1;
  A.rng_state = rng_state48;
  A.jump = e->jump;
  A.window = static_cast<long long*>(window_dev);
  A.stride = stride;
  A.flags = static_cast<int*>(flags_dev);
  int rcm = shard_window_exchange(e, window_dev, m1 - m0, world, &A);
  if (rcm != BPF_OK)
    return rcm;
  if (e->shard_chain)
  {
    // w_diff > 0 (bpf_shard_begin_resample built the chain from this same stream state)
    if (rng_state48 != e->shard_rng0 || m1 > e->max_samples)
      return e->fail(BPF_ERR_INVALID_ARGUMENT, "draw window does not belong to the resample begun");
    A.chain = e->d_chain.p;
    A.write_random = rank == 0;
    int rcf = ensure_free_space(e, &A.free_space);
    if (rcf != BPF_OK)
      return rcf;
  }
  if (!e->shard_chain && m0 < kFusedWindow)
  {
    int rcj = ensure_fused_jump(e);
    if (rcj != BPF_OK)
      return rcj;
    A.jump_table = e->d_fused_jump.p;
    A.jump_table_n = kFusedWindow;
  }
  size_t lds = 0;
  if (e->cdf_coarse_n == A.n_src && A.n_src > 0 && m1 - m0 <= 2 * kFusedWindow)
  {
    // a short window after k_normalize_gathered_cdf: the draws bracket themselves in the CDF subsample first
    A.coarse = e->d_cdf_coarse.p;
    A.coarse_shift = fused_coarse_shift(A.n_src);
    lds = ((size_t)((A.n_src - 1) >> A.coarse_shift) + 2) * sizeof(double);
  }
  *lds_out = lds;
  return BPF_OK;
}
}  // namespace

int bpf_shard_draw_window_dev(bpf_engine* e, uint64_t rng_state48, int m0, int m1, const void* sums_dev,
                              int sums_are_totals, int rank, int world, void* window_dev, int stride, void* flags_dev)
{
  WindowArgs A{};
  size_t lds = 0;
  int rc = draw_window_args(e, rng_state48, m0, m1, sums_dev, sums_are_totals, rank, world, window_dev, stride, flags_dev,
                            &A, &lds);
  if (rc != BPF_OK)
    return rc;
  ProfScope ps(e, BPF_K_DRAW);
  hipLaunchKernelGGL(k_draw_window, dim3(blocks_for(m1 - m0, 256)), dim3(256), lds, e->stream, A);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

int bpf_shard_adopt_dev(bpf_engine* e, const void* x_dev, const void* y_dev, const void* theta_dev, int count,
                        int global_count, int leaf_count, int bin_count)
{
  if (!e || !e->have_pf || count < 0 || global_count <= 0 || (count > 0 && (!x_dev || !y_dev || !theta_dev)))
    return BPF_ERR_INVALID_ARGUMENT;
  if (count > e->max_samples)
    return e->fail(BPF_ERR_CAPACITY, "adopted shard larger than max_samples");
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& b = e->sets[e->cur ^ 1];
  if (count > 0)
  {
    ProfScope ps(e, BPF_K_FINALIZE);
    hipLaunchKernelGGL(k_adopt, dim3(blocks_for(count, 256)), dim3(256), 0, e->stream,
                       static_cast<const double*>(x_dev), static_cast<const double*>(y_dev),
                       static_cast<const double*>(theta_dev), b.dev(), count, 1.0 / (double)global_count);
    HIPCHK(e, hipGetLastError());
  }
  e->cur ^= 1;
  e->sample_count = count;
  e->tree_pending = false;
  e->leaf_count = leaf_count;
  e->bin_count = bin_count;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  return BPF_OK;
}

int bpf_shard_tail_small_dev(bpf_engine* e, const void* x_all_dev, const void* y_all_dev, const void* theta_all_dev,
                             int global_count, int lo, int hi, int leaf_count, int bin_count)
{
  if (!e || !e->have_pf || !x_all_dev || !y_all_dev || !theta_all_dev || global_count <= 0 || lo < 0 || hi < lo ||
      hi > global_count)
    return BPF_ERR_INVALID_ARGUMENT;
  if (hi - lo > e->max_samples)
    return e->fail(BPF_ERR_CAPACITY, "adopted shard larger than max_samples");
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& b = e->sets[e->cur ^ 1];
  {
    ProfScope ps(e, BPF_K_FINALIZE);
    hipLaunchKernelGGL(k_shard_tail_small, dim3(1), dim3(1024), 0, e->stream, static_cast<const double*>(x_all_dev),
                       static_cast<const double*>(y_all_dev), static_cast<const double*>(theta_all_dev), global_count,
                       lo, hi, b.dev(), e->dist_threshold, e->d_scalars.p, e->d_flags.p + 1);
  }
  HIPCHK(e, hipGetLastError());
  e->cur ^= 1;
  e->sample_count = hi - lo;
  e->tree_pending = false;
  e->leaf_count = leaf_count;
  e->bin_count = bin_count;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  e->converged_pending = true;
  e->conv_n = global_count;
  return BPF_OK;
}

namespace
{
// The sharded resample's stop rule, adoption and updateConverged in one single-block launch from an exchanged window
// (k_shard_stop_block): *status is BPF_FUSED_OK when it was handled (the engine then holds its share of the new set),
// another BPF_FUSED_* when the window is outside what the kernel takes -- nothing was changed and the window stays
// readable for the stage-by-stage path.
int shard_stop_block(bpf_engine* e, const long long* window, int stride, int count, bool systematic, int* status,
                     int* M_out, int* leaf_out, int* bins_out, const WindowArgs* draw = nullptr)
{
  *status = BPF_FUSED_TOO_MANY_BINS;
  if (count <= 0 || count > kFusedWindow || stride < count)
    return BPF_OK;
  if (!systematic)
  {
    int rc = ensure_limit_table(e, count);
    if (rc != BPF_OK)
      return rc;
  }
  HIPCHK(e, e->h_fused.reserve(32));
  if (!e->shard_stop_attr_set)
  {
    HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_shard_stop_block),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds));
    e->shard_stop_attr_set = true;
  }
  SampleSet& b = e->sets[e->cur ^ 1];
  ShardStopArgs A{};
  A.window = window;
  A.stride = stride;
  A.count = count;
  A.systematic = systematic ? 1 : 0;
  A.max_samples = e->max_samples;
  A.limit = e->d_kld_limit.p;
  A.rank = e->shard_rank;
  A.world = std::max(1, e->shard_world);
  A.dst = b.dev();
  A.thr = e->dist_threshold;
  A.sc = e->d_scalars.p;
  A.conv_count = e->d_flags.p + 1;
  A.mb = shard_window_wait(e, window, 1);
  A.wait_parity = (int)(e->mb.win_gen & 1);
  A.wait_gen = e->mb.win_gen;
  A.result_host = e->h_fused.p;
  e->fused_generation = (e->fused_generation % 0x3fffffff) + 1;
  A.generation = e->fused_generation;
  A.debug = getenv("BPF_DEBUG") != nullptr;
  if (draw != nullptr)
  {
    // mailbox mode: the draws of the window and their consumer in one launch (k_shard_resample_block)
    if (!e->shard_resample_attr_set)
    {
      HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_shard_resample_block),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds));
      e->shard_resample_attr_set = true;
    }
    ShardResampleArgs R{};
    R.W = *draw;
    R.S = A;
    ProfScope ps(e, BPF_K_DRAW);
    hipLaunchKernelGGL(k_shard_resample_block, dim3(blocks_for(count, kFusedDrawsPerBlock)), dim3(1024), kFusedLds,
                       e->stream, R);
  }
  else
  {
    ProfScope ps(e, BPF_K_FINALIZE);
    hipLaunchKernelGGL(k_shard_stop_block, dim3(1), dim3(1024), kFusedLds, e->stream, A);
  }
  HIPCHK(e, hipGetLastError());
  int r5[5] = { 0, 0, 0, 0, 0 };
  bool seen = false;
  int rcw = fused_result_wait(e, A.generation, 50, r5, &seen);
  if (rcw != BPF_OK)
    return rcw;
  if (!seen)
    return e->fail(BPF_ERR_HIP, "k_shard_stop_block did not publish its result");
  if (int rcx = mailbox_check(e))
    return rcx;
  const int* res = e->h_fused.p;  // (the debug stamps, [8 ..])
  *status = r5[3];
  if (A.debug)
    fprintf(stderr, "[shard stop block] count %d M %d leaf %d bins %d status %d levels %d (10 ns ticks): wait %d load %d "
            "dedup %d tree %d scan %d tail %d\n", count, r5[0], r5[1], r5[2], r5[3], r5[4], res[9] - res[8],
            res[11] - res[9], res[12] - res[11], res[13] - res[12], res[14] - res[13], res[15] - res[14]);
  if (r5[3] != BPF_FUSED_OK)
    return BPF_OK;
  const int M = r5[0], W = A.world;
  const int lo = (int)(((long long)M * A.rank) / W), hi = (int)(((long long)M * (A.rank + 1)) / W);
  if (hi - lo > e->max_samples)
    return e->fail(BPF_ERR_CAPACITY, "adopted shard larger than max_samples");
  *M_out = M;
  *leaf_out = r5[1];
  *bins_out = r5[2];
  e->cur ^= 1;
  e->sample_count = hi - lo;
  e->tree_pending = false;
  e->leaf_count = r5[1];
  e->bin_count = r5[2];
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  e->converged_pending = true;
  e->conv_n = M;
  e->hist_matches_set = false;  // no host histogram of this set
  return BPF_OK;
}
}  // namespace

int bpf_shard_converged_dev(bpf_engine* e, const void* x_all_dev, const void* y_all_dev, int global_count)
{
  if (!e || !e->have_pf || !x_all_dev || !y_all_dev || global_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  const double* x = static_cast<const double*>(x_all_dev);
  const double* y = static_cast<const double*>(y_all_dev);
  int rc = sum_into_slot(e, x, global_count, 3, 0, global_count);
  if (rc != BPF_OK)
    return rc;
  rc = sum_into_slot(e, y, global_count, 4, 0, global_count);
  if (rc != BPF_OK)
    return rc;
  HIPCHK(e, hipMemsetAsync(e->d_flags.p + 1, 0, sizeof(int), e->stream));
  const int grid = std::max(1, std::min(blocks_for(global_count, 256), 1024));
  hipLaunchKernelGGL(k_count_converged, dim3(grid), dim3(256), 0, e->stream, x, y, global_count, e->d_scalars.p,
                     e->dist_threshold, e->d_flags.p + 1);
  HIPCHK(e, hipGetLastError());
  e->converged_pending = true;
  e->conv_n = global_count;
  return BPF_OK;
}

uint64_t bpf_drand48_skip(uint64_t state48, uint64_t n)
{
  static LcgJump J;
  static bool init = false;
  if (!init)
  {
    lcg_tables(J);
    init = true;
  }
  return lcg_skip_host(state48 & ((1ull << 48) - 1), n, J);
}

int bpf_kld_reset(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->hist.clear();
  e->seen.reset((size_t)std::min(std::max(e->max_samples, 1024), 1 << 20));
  return BPF_OK;
}

int bpf_kld_feed(bpf_engine* e, const void* keys, int keys_are_int64, int stride, int n_keys, int first_draw_index,
                 int* stop_count_out)
{
  if (!e || !keys || !stop_count_out || stride < n_keys)
    return BPF_ERR_INVALID_ARGUMENT;
  *stop_count_out = -1;
  const long long* k64 = static_cast<const long long*>(keys);
  const int* k32 = static_cast<const int*>(keys);
  int cached_leaf = -1, cached_limit = 0;
  for (int q = 0; q < n_keys; ++q)
  {
    int k[3];
    for (int d = 0; d < 3; ++d)
      k[d] = keys_are_int64 ? (int)k64[(size_t)d * stride + q] : k32[(size_t)d * stride + q];
    if (e->seen.first_time(k[0], k[1], k[2]))
      e->hist.insert(k[0], k[1], k[2]);
    const int lc = e->hist.leaf_count();
    if (lc != cached_leaf)
    {
      cached_leaf = lc;
      cached_limit = resample_limit(lc, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
    }
    const int count = first_draw_index + q + 1;
    if (count > cached_limit)
    {
      *stop_count_out = count;
      break;
    }
  }
  return BPF_OK;
}

int bpf_kld_feed_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys, int first_draw_index,
                     int* stop_count_out)
{
  if (!e || !window_dev || !stop_count_out || stride < n_keys || n_keys <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->h_keys.reserve((size_t)n_keys * 3));
  const unsigned generation = ++e->done_generation;
  const MailboxDev wait = shard_window_wait(e, window_dev, blocks_for(n_keys, 256));
  hipLaunchKernelGGL(k_publish_window_keys, dim3(blocks_for(n_keys, 256)), dim3(256), 0, e->stream,
                     static_cast<const long long*>(window_dev), stride, n_keys, e->h_keys.p,
                     reinterpret_cast<unsigned*>(e->d_flags.p + 4), reinterpret_cast<volatile unsigned*>(e->h_done.p),
                     generation, wait, (int)(e->mb.win_gen & 1), e->mb.win_gen);
  HIPCHK(e, hipGetLastError());
  if (!wait_generation(e, generation))
    HIPCHK(e, hipStreamSynchronize(e->stream));
  if (int rcx = mailbox_check(e))
    return rcx;
  return bpf_kld_feed(e, e->h_keys.p, 0, n_keys, n_keys, first_draw_index, stop_count_out);
}

int bpf_shard_begin_resample(bpf_engine* e, uint64_t rng_state48, int leaf_count, double* w_diff_out,
                             int* systematic_count_out)
{
  if (!e || !w_diff_out || !systematic_count_out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  // w_diff = max(0, 1 - w_fast / w_slow) (particle_filter.cpp:438-440); the averages are the same on every shard
  double w_diff = 0.0;
  if (e->alpha_slow != 0.0 || e->alpha_fast != 0.0)
  {
    int rc = fetch_scalars(e);
    if (rc != BPF_OK)
      return rc;
    w_diff = 1.0 - e->h_scalars.p->v[2] / e->h_scalars.p->v[1];
    if (!(w_diff >= 0.0))
      w_diff = 0.0;
  }
  e->w_diff_last = w_diff;
  e->shard_w_diff = w_diff;
  e->shard_chain = false;
  e->shard_n_random = 0;
  e->shard_rng0 = rng_state48 & ((1ull << 48) - 1);
  int count = resample_limit(leaf_count, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
  if (w_diff > 0.0)
  {
    if (e->resample_model == BPF_RESAMPLE_SYSTEMATIC)
    {
      count *= (1.0 + w_diff);  // :295-306
      if (count > e->max_samples)
        count = e->max_samples;
      e->shard_n_random = (int)(w_diff * count);
    }
    else
    {
      FreeSpaceDev fs{};
      int rc = ensure_free_space(e, &fs);
      if (rc != BPF_OK)
        return rc;
      const uint64_t keep = e->rng;
      e->rng = e->shard_rng0;
      rc = build_draw_chain(e, w_diff, e->max_samples);
      e->rng = keep;
      if (rc != BPF_OK)
        return rc;
      e->shard_chain = true;
    }
  }
  *w_diff_out = w_diff;
  *systematic_count_out = count;
  return BPF_OK;
}

int bpf_shard_end_resample(bpf_engine* e, int sample_count, uint64_t* rng_state48_out)
{
  if (!e || !rng_state48_out || sample_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  uint64_t consumed;
  if (e->resample_model == BPF_RESAMPLE_SYSTEMATIC)
    consumed = 1ull + 2ull * (uint64_t)e->shard_n_random;
  else if (e->shard_chain)
  {
    if (sample_count > e->max_samples)
      return e->fail(BPF_ERR_INVALID_ARGUMENT, "sample_count beyond the chain");
    HIPCHK(e, e->h_chain_word.reserve(1));
    HIPCHK(e, hipMemcpyAsync(e->h_chain_word.p, e->d_chain.p + sample_count, sizeof(int), hipMemcpyDeviceToHost,
                             e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    consumed = (uint64_t)((unsigned)e->h_chain_word.p[0] & 0x7fffffffu) - 1ull;
  }
  else
    consumed = 2ull * (uint64_t)sample_count;
  *rng_state48_out = lcg_skip_host(e->shard_rng0, consumed, e->jump);
  if (e->shard_w_diff > 0.0)  // particle_filter.cpp:453-455
    HIPCHK(e, hipMemsetAsync(&e->d_scalars.p->v[1], 0, 2 * sizeof(double), e->stream));
  e->shard_chain = false;
  e->shard_n_random = 0;
  return BPF_OK;
}

int bpf_pf_resample_limit(bpf_engine* e, int leaf_count, int* count_out)
{
  if (!e || !count_out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  *count_out = resample_limit(leaf_count, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
  return BPF_OK;
}

namespace
{
// the arguments of a systematic draw window; the caller launches (and records targets_read behind the launch)
int systematic_window_args(bpf_engine* e, uint64_t rng_state48, int count, const void* sums_dev, int sums_are_totals,
                           int rank, int world, void* window_dev, int stride, void* flags_dev, WindowArgs* out,
                           size_t* lds_out)
{
  if (!e || !e->have_pf || !sums_dev || !window_dev || !flags_dev || count <= 0 || stride < count || rank < 0 ||
      rank >= world)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad systematic window arguments") : BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->h_targets.reserve((size_t)std::max(count, e->max_samples)));
  // the reference's serial chain (particle_filter.cpp:337-341): target += delta, -= 1 once it passes 1
  const uint64_t st = lcg_skip_host(rng_state48 & ((1ull << 48) - 1), 1, e->jump);
  double t = std::ldexp((double)st, -48);
  const int n_random = e->shard_n_random;
  if (n_random < 0 || n_random >= count)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "systematic window: random pose count out of range");
  const int n_systematic = count - n_random;
  const double delta = 1.0 / n_systematic;
  if (e->targets_read)  // a previous window kernel may still be reading the pinned targets
    HIPCHK(e, hipEventSynchronize(e->targets_read));
  for (int i = 0; i < n_systematic; ++i)
  {
    e->h_targets.p[i] = t;
    t += delta;
    if (t > 1.0)
      t -= 1.0;
  }
  WindowArgs& A = *out;
  A = WindowArgs{};
  A.src = e->sets[e->cur].dev();
  A.n_src = e->sample_count;
  A.cdf = e->d_cdf.p;
  A.sums = static_cast<const double*>(sums_dev);
  A.sums_are_totals = sums_are_totals;
  A.rank = rank;
  A.world = world;
  A.m0 = 0;
  A.m1 = count;
  A.rng_state = rng_state48 & ((1ull << 48) - 1);
  A.jump = e->jump;
  A.n_random = n_random;
  A.write_random = rank == 0;
  if (n_random > 0)
  {
    int rcf = ensure_free_space(e, &A.free_space);
    if (rcf != BPF_OK)
      return rcf;
  }
  A.window = static_cast<long long*>(window_dev);
  A.stride = stride;
  A.flags = static_cast<int*>(flags_dev);
  A.targets = e->h_targets.p;
  int rcm = shard_window_exchange(e, window_dev, count, world, &A);
  if (rcm != BPF_OK)
    return rcm;
  size_t lds = 0;
  if (e->cdf_coarse_n == A.n_src && A.n_src > 0 && count <= 2 * kFusedWindow)
  {
    A.coarse = e->d_cdf_coarse.p;
    A.coarse_shift = fused_coarse_shift(A.n_src);
    lds = ((size_t)((A.n_src - 1) >> A.coarse_shift) + 2) * sizeof(double);
  }
  *lds_out = lds;
  return BPF_OK;
}

// a window kernel that reads the pinned systematic targets has been launched: the next targets wait for it
int systematic_targets_in_use(bpf_engine* e)
{
  if (!e->targets_read)
    HIPCHK(e, hipEventCreateWithFlags(&e->targets_read, hipEventDisableTiming));
  HIPCHK(e, hipEventRecord(e->targets_read, e->stream));
  return BPF_OK;
}
}  // namespace

int bpf_shard_systematic_window_dev(bpf_engine* e, uint64_t rng_state48, int count, const void* sums_dev,
                                    int sums_are_totals, int rank, int world, void* window_dev, int stride,
                                    void* flags_dev)
{
  WindowArgs A{};
  size_t lds = 0;
  int rc = systematic_window_args(e, rng_state48, count, sums_dev, sums_are_totals, rank, world, window_dev, stride,
                                  flags_dev, &A, &lds);
  if (rc != BPF_OK)
    return rc;
  {
    ProfScope ps(e, BPF_K_DRAW);
    hipLaunchKernelGGL(k_draw_window, dim3(blocks_for(count, 256)), dim3(256), lds, e->stream, A);
  }
  HIPCHK(e, hipGetLastError());
  return systematic_targets_in_use(e);
}

int bpf_kld_insert(bpf_engine* e, const void* keys, int keys_are_int64, int stride, int n_keys)
{
  if (!e || !keys || stride < n_keys)
    return BPF_ERR_INVALID_ARGUMENT;
  const long long* k64 = static_cast<const long long*>(keys);
  const int* k32 = static_cast<const int*>(keys);
  for (int q = 0; q < n_keys; ++q)
  {
    int k[3];
    for (int d = 0; d < 3; ++d)
      k[d] = keys_are_int64 ? (int)k64[(size_t)d * stride + q] : k32[(size_t)d * stride + q];
    if (e->seen.first_time(k[0], k[1], k[2]))
      e->hist.insert(k[0], k[1], k[2]);
  }
  return BPF_OK;
}

int bpf_kld_insert_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys)
{
  if (!e || !window_dev || stride < n_keys || n_keys <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->h_keys.reserve((size_t)n_keys * 3));
  const unsigned generation = ++e->done_generation;
  const MailboxDev wait = shard_window_wait(e, window_dev, blocks_for(n_keys, 256));
  hipLaunchKernelGGL(k_publish_window_keys, dim3(blocks_for(n_keys, 256)), dim3(256), 0, e->stream,
                     static_cast<const long long*>(window_dev), stride, n_keys, e->h_keys.p,
                     reinterpret_cast<unsigned*>(e->d_flags.p + 4), reinterpret_cast<volatile unsigned*>(e->h_done.p),
                     generation, wait, (int)(e->mb.win_gen & 1), e->mb.win_gen);
  HIPCHK(e, hipGetLastError());
  if (!wait_generation(e, generation))
    HIPCHK(e, hipStreamSynchronize(e->stream));
  if (int rcx = mailbox_check(e))
    return rcx;
  return bpf_kld_insert(e, e->h_keys.p, 0, n_keys, n_keys);
}

int bpf_kld_stop_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys, int* handled_out,
                     int* stop_count_out, int* leaf_count_out, int* bin_count_out)
{
  if (!e || !window_dev || !handled_out || !stop_count_out || !leaf_count_out || !bin_count_out || stride < n_keys ||
      n_keys <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->d_keys.reserve((size_t)n_keys * 3));
  const MailboxDev wait = shard_window_wait(e, window_dev, blocks_for(n_keys, 256));
  hipLaunchKernelGGL(k_window_keys_to_aos, dim3(blocks_for(n_keys, 256)), dim3(256), 0, e->stream,
                     static_cast<const long long*>(window_dev), stride, n_keys, e->d_keys.p, wait,
                     (int)(e->mb.win_gen & 1), e->mb.win_gen);
  HIPCHK(e, hipGetLastError());
  bool handled = false;
  *stop_count_out = -1;
  *leaf_count_out = *bin_count_out = 0;
  int rc = kld_tree_on_device(e, n_keys, &handled, stop_count_out, leaf_count_out, bin_count_out);
  *handled_out = handled ? 1 : 0;
  if (rc == BPF_OK)
    rc = mailbox_check(e);
  return rc;
}

int bpf_kld_leaf_count(bpf_engine* e, int* leaf_count_out, int* bin_count_out)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (leaf_count_out)
    *leaf_count_out = e->hist.leaf_count();
  if (bin_count_out)
    *bin_count_out = e->hist.bin_count();
  return BPF_OK;
}

int bpf_set_option(bpf_engine* e, int option, int value)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (option == BPF_OPT_CDF_SERIAL)
    e->cdf_serial = value != 0;
  else if (option == BPF_OPT_WINDOW_PATH)
    e->window_enabled = value != 0;
  else if (option == BPF_OPT_COUNT_CELLS)
    e->count_cells = value != 0;
  else if (option == BPF_OPT_KLD_DEVICE_MIN)
    e->kld_device_min = value > 0 ? value : 0x7fffffff;
  else if (option == BPF_OPT_GRADED_SHARES)
    e->graded_shares = value != 0;
  else if (option == BPF_OPT_FUSED_RESAMPLE)
    e->fused_resample = value != 0;
  else if (option == BPF_OPT_CLOUD_DENSE)
    e->cloud_dense = value != 0;
  else if (option == BPF_OPT_LUT_HOST)
    e->lut_host = value != 0;
  else if (option == BPF_OPT_LUT_EXACT_EDT)
    e->lut_exact_edt = value != 0;
  else if (option == BPF_OPT_HOST_AUTO_REGISTER)
    e->host_auto_register = value != 0;
  else if (option == BPF_OPT_SEAM_CHUNKS)
    e->seam_chunks = value < 0 ? 0 : (value > kSeamMaxChunks ? kSeamMaxChunks : value);
  else if (option == BPF_OPT_KLD_PERSISTENT)
    e->kld_persistent = value != 0;
  else if (option == BPF_OPT_KLD_LOCAL)
    e->kld_local = value != 0;
  else if (option == BPF_OPT_TILE_SORT)
    e->tile_sort = value != 0;
  else if (option == BPF_OPT_HOST_DIRECT_PAGEABLE)
    e->host_direct = value != 0;
  else if (option == BPF_OPT_STATS_HOST)
  {
    e->stats_host = value != 0;
    e->stats_epoch = -1;
  }
  else
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "unknown option");
  return BPF_OK;
}

int bpf_get_cells_walked(bpf_engine* e, unsigned long long* out, int reset)
{
  if (!e || !out)
    return BPF_ERR_INVALID_ARGUMENT;
  *out = 0;
  if (!e->d_cells_walked.p)
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipMemcpyAsync(out, e->d_cells_walked.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (reset)
    HIPCHK(e, hipMemsetAsync(e->d_cells_walked.p, 0, sizeof(unsigned long long), e->stream));
  return BPF_OK;
}
