// Host half of normalisation and resampling: CDF, updateConverged, free-space list and draw chain (recovery),
// KLD stop rule (device tree and ordered host replay), multinomial and systematic resamplers.
int fetch_scalars(bpf_engine* e)
{
  HIPCHK(e, hipMemcpyAsync(e->h_scalars.p, e->d_scalars.p, sizeof(FilterScalars), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->h_flags.p, e->d_flags.p, 8 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (e->converged_pending)
  {
    // particle_filter.cpp:206-219, float arithmetic for the percentage
    const double pct = (float)e->h_flags.p[1] / (float)e->conv_n * 100;
    e->percent_converged = (float)pct;
    e->converged = pct >= e->conv_threshold;
    e->converged_pending = false;
  }
  return BPF_OK;
}

// zero_word2 / sum_out (optional): a second miss flag to clear and where to leave c[n], both written by the scan
// itself instead of by a memset and a copy behind it
// k_resample_block stages every (1 << shift)-th CDF value in LDS: the smallest shift that fits kFusedCoarse brackets
int fused_coarse_shift(int n)
{
  int shift = 0;
  while ((((n - 1) >> shift) + 1) > kFusedCoarse)
    shift++;
  return shift;
}

int ensure_cdf_buffers(bpf_engine* e, int n)
{
  HIPCHK(e, e->d_cdf.reserve((size_t)n + 1 + 64));  // slack: k_resample_block reads whole 32-value brackets
  if (!e->d_cdf_guide.p)
  {
    HIPCHK(e, e->d_cdf_guide.reserve(kCdfGuide + 2));
    HIPCHK(e, hipMemsetAsync(e->d_cdf_guide.p, 0xFF, (kCdfGuide + 2) * sizeof(int), e->stream));  // "no entry"
  }
  return BPF_OK;
}

// ParticleFilter::updateSensor's normalisation and the CDF of the normalised weights in one launch (k_normalize_cdf);
// needs the scoring kernel's per-block partials (e->fused_partials) and at most 256 tiles
int launch_normalize_cdf(bpf_engine* e, double* w, int n)
{
  const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
  int rc = ensure_cdf_buffers(e, n);
  if (rc != BPF_OK)
    return rc;
  if (e->d_tile_slots.cap < (size_t)2 * BPF_RED_BLOCK)
  {
    HIPCHK(e, e->d_tile_slots.reserve((size_t)2 * BPF_RED_BLOCK));
    HIPCHK(e, hipMemsetAsync(e->d_tile_slots.p, 0xFF, 2 * BPF_RED_BLOCK * sizeof(unsigned long long), e->stream));
    e->tile_generation = 0;
  }
  NormCdfArgs A{};
  A.w = w;
  A.n = n;
  A.block_partials = e->d_block_partials.p;
  A.n_partials = e->fused_partials;
  A.sc = e->d_scalars.p;
  A.alpha_slow = e->alpha_slow;
  A.alpha_fast = e->alpha_fast;
  A.tile_slots = e->d_tile_slots.p;
  A.generation = ++e->tile_generation;  // (only its parity matters)
  A.cdf = e->d_cdf.p;
  A.coarse_shift = fused_coarse_shift(n);
  HIPCHK(e, e->d_cdf_coarse.reserve((size_t)kFusedCoarse + 2));
  A.coarse = e->d_cdf_coarse.p;
  // k_resample_block searches through the subsample; the guide table is for the general path's draw kernel, worth its
  // writes only when a long draw stream is expected (the previous resample ran to the end: a spread cloud)
  const bool want_guide = e->window_hint >= e->max_samples;
  if (getenv("BPF_DEBUG"))
    fprintf(stderr, "[normalize_cdf] n %d window_hint %d max %d guide %d\n", n, e->window_hint, e->max_samples, (int)want_guide);
  A.guide = want_guide ? e->d_cdf_guide.p : nullptr;
  A.zero_word = e->d_flags.p;
  ProfScope ps(e, BPF_K_NORMALIZE);
  hipLaunchKernelGGL(k_normalize_cdf, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, A);
  HIPCHK(e, hipGetLastError());
  e->tile_sums_n = -1;
  e->cdf_ready_n = n;
  e->cdf_coarse_n = n;
  e->cdf_guide_valid = want_guide;
  return BPF_OK;
}

int build_cdf(bpf_engine* e, const double* w, int n, int* zero_word2 = nullptr, double* sum_out = nullptr)
{
  if (e->cdf_ready_n == n && w == e->sets[e->cur].w.p && !e->cdf_serial && zero_word2 == nullptr && sum_out == nullptr)
    return BPF_OK;  // k_normalize_cdf left it behind
  int rcb = ensure_cdf_buffers(e, n);
  if (rcb != BPF_OK)
    return rcb;
  e->cdf_coarse_n = -1;  // the scan kernels below leave no subsample
  e->cdf_guide_valid = !e->cdf_serial;
  ProfScope ps(e, BPF_K_CDF);
  if (e->cdf_serial)
  {
    hipLaunchKernelGGL(k_scan_serial, dim3(1), dim3(64), 0, e->stream, w, n, e->d_cdf.p);
    HIPCHK(e, hipMemsetAsync(e->d_flags.p, 0, sizeof(int), e->stream));
    if (zero_word2)
      HIPCHK(e, hipMemsetAsync(zero_word2, 0, sizeof(int), e->stream));
    if (sum_out)
      HIPCHK(e, hipMemcpyAsync(sum_out, e->d_cdf.p + n, sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  }
  else
  {
    const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
    double* tiles;
    if (e->tile_sums_n == n && w == e->sets[e->cur].w.p)
    {
      tiles = e->d_tile_sums.p;  // left behind by k_normalize_fused; consumed (scanned in place) here
      e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
    }
    else
    {
      HIPCHK(e, e->d_partials.reserve((size_t)nb));
      tiles = e->d_partials.p;
      hipLaunchKernelGGL(k_sum_partials, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, w, n, tiles);
    }
    if (nb <= 256)
    {
      // few tiles: every block of the final pass forms its own offset (one launch less)
      hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, w, n, tiles, 0, e->d_cdf.p,
                         e->d_flags.p, e->d_cdf_guide.p, zero_word2, sum_out);
    }
    else
    {
      hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(BPF_RED_BLOCK), 0, e->stream, tiles, nb, e->d_flags.p);
      hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, w, n, tiles, 1, e->d_cdf.p,
                         nullptr, e->d_cdf_guide.p, zero_word2, sum_out);
    }
  }
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

// updateConverged (particle_filter.cpp:170-220) on the current set, result fetched lazily
int launch_converged(bpf_engine* e)
{
  SampleSet& s = e->sets[e->cur];
  const int n = e->sample_count;
  const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
  HIPCHK(e, e->d_partials.reserve((size_t)2 * nb));
  const int grid = std::max(1, std::min(blocks_for(n, 256), 1024));
  ProfScope ps(e, BPF_K_FINALIZE);
  hipLaunchKernelGGL(k_sum_xy_partials, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, (const double*)s.x.p,
                     (const double*)s.y.p, n, e->d_partials.p, e->d_partials.p + nb, e->d_flags.p + 1);
  hipLaunchKernelGGL(k_converged_count, dim3(grid), dim3(BPF_RED_BLOCK), 0, e->stream, (const double*)s.x.p,
                     (const double*)s.y.p, n, (const double*)e->d_partials.p, (const double*)(e->d_partials.p + nb), nb,
                     e->d_scalars.p, e->dist_threshold, e->d_flags.p + 1);
  HIPCHK(e, hipGetLastError());
  e->converged_pending = true;
  e->conv_n = n;
  return BPF_OK;
}

// Node2D::updateFreeSpaceIndices (node_2d.cpp:317-337) for the current map and non_free_space_radius, cached
int ensure_free_space(bpf_engine* e, FreeSpaceDev* out)
{
  if (e->random_pose_mode != BPF_RANDOM_POSE_FREE_SPACE_2D)
    return e->fail(BPF_ERR_UNSUPPORTED,
                   "w_diff > 0: random pose injection needs a pose generator (bpf_pf_set_random_pose_generator); "
                   "the node's random_pose_fn_ callback (particle_filter.cpp:385-388) cannot be called from here");
  if (!e->have_map || !e->have_lut)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "random free-space poses need the 2-D map and its distance LUT");
  const double radius = e->pm.non_free_radius;
  if (e->free_map_version != e->map_version || e->free_radius != radius)
  {
    const int sx = e->map.size_x, sy = e->map.size_y;
    std::vector<int2> ij;
    ij.reserve((size_t)sx * sy / 2);
    for (int i = 0; i < sx; ++i)
      for (int j = 0; j < sy; ++j)
      {
        const size_t idx = i + (size_t)j * sx;
        if (e->h_cells8[idx] == -1 && (double)e->h_lut_f32[idx] > radius)
          ij.push_back(make_int2(i, j));
      }
    e->n_free = (int)ij.size();
    HIPCHK(e, e->d_free_ij.reserve(std::max<size_t>(ij.size(), 1)));
    if (!ij.empty())
      H2D_OR_RETURN(h2d_from_host_sync(e, e->d_free_ij.p, ij.data(), ij.size() * sizeof(int2)));
    e->free_map_version = e->map_version;
    e->free_radius = radius;
  }
  if (e->n_free <= 0)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "the map has no free cell to draw random poses from");
  out->ij = e->d_free_ij.p;
  out->n = e->n_free;
  out->size_x = e->map.size_x;
  out->size_y = e->map.size_y;
  out->origin_x = e->map.origin_x;
  out->origin_y = e->map.origin_y;
  out->resolution = e->map.resolution;
  return BPF_OK;
}

// Where every candidate draw 0 .. max_draws finds its stream elements when w_diff > 0 (kernels_recovery.hpp)
int build_draw_chain(bpf_engine* e, double w_diff, int max_draws)
{
  const long long positions = 3ll * ((long long)max_draws + 1) + 3;
  if (positions >= 0x7fffffffll)
    return e->fail(BPF_ERR_CAPACITY, "draw chain would pass 31-bit stream positions");
  const int n_seg = (int)((positions + kChainSeg - 1) / kChainSeg);
  HIPCHK(e, e->d_chain_bits.reserve((size_t)2 * n_seg));
  HIPCHK(e, e->d_chain_cnt.reserve((size_t)3 * n_seg));
  HIPCHK(e, e->d_chain_exit.reserve((size_t)3 * n_seg));
  HIPCHK(e, e->d_chain_entry.reserve((size_t)n_seg));
  HIPCHK(e, e->d_chain_base.reserve((size_t)n_seg));
  HIPCHK(e, e->d_chain.reserve((size_t)max_draws + 1));
  ChainArgs C{};
  C.rng_state = e->rng;
  C.w_diff = w_diff;
  C.n_seg = n_seg;
  C.max_draws = max_draws;
  C.seg_bits = e->d_chain_bits.p;
  C.seg_cnt = e->d_chain_cnt.p;
  C.seg_exit = e->d_chain_exit.p;
  C.seg_entry = e->d_chain_entry.p;
  C.seg_base = e->d_chain_base.p;
  C.chain = e->d_chain.p;
  C.jump = e->jump;
  ProfScope ps(e, BPF_K_DRAW);
  hipLaunchKernelGGL(k_chain_segments, dim3(blocks_for(n_seg, 256)), dim3(256), 0, e->stream, C);
  hipLaunchKernelGGL(k_chain_scan, dim3(1), dim3(1024), 0, e->stream, C);
  hipLaunchKernelGGL(k_chain_emit, dim3(blocks_for(n_seg, 256)), dim3(256), 0, e->stream, C);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

// resampleLimit per leaf count 0 .. upto on the device, cached per parameter set (host libm, as the reference
// evaluates it)
int ensure_limit_table(bpf_engine* e, int upto)
{
  if (e->kld_limit_key[0] != e->pop_err || e->kld_limit_key[1] != e->pop_z || e->kld_limit_key[2] != e->min_samples ||
      e->kld_limit_key[3] != e->max_samples || (int)e->kld_limit_host.size() < upto + 1)
  {
    const int n = std::max(upto, (int)e->kld_limit_host.size() - 1);
    e->kld_limit_host.resize((size_t)n + 1);
    for (int k = 0; k <= n; ++k)
      e->kld_limit_host[k] = resample_limit(k, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
    HIPCHK(e, e->d_kld_limit.reserve((size_t)n + 1));
    HIPCHK(e, hipMemcpyAsync(e->d_kld_limit.p, e->kld_limit_host.data(), ((size_t)n + 1) * sizeof(int),
                             hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->kld_limit_key[0] = e->pop_err;
    e->kld_limit_key[1] = e->pop_z;
    e->kld_limit_key[2] = e->min_samples;
    e->kld_limit_key[3] = e->max_samples;
  }
  return BPF_OK;
}

// The whole resample for a candidate stream of at most kFusedWindow draws as one single-block launch
// (k_resample_block): draws, histogram tree and KLD stop, weights 1/M, updateConverged.  *handled = false when the
// stop lies beyond the window, a key does not fit the packing or the tree is too deep: the caller runs the general path.
// draw m takes stream element 2 m + 2: the composed LCG step for that distance, per draw of a window kernel
int ensure_fused_jump(bpf_engine* e)
{
  if (e->d_fused_jump.p)
    return BPF_OK;
  std::vector<FusedJump> jt(kFusedWindow);
  const uint64_t mask = (1ull << 48) - 1;
  for (int m = 0; m < kFusedWindow; ++m)
  {
    uint64_t a = 1, c = 0, k = 2ull * (uint64_t)m + 2ull;
    for (int j = 0; k != 0 && j < 48; ++j, k >>= 1)
      if (k & 1)
      {
        a = (a * e->jump.A[j]) & mask;
        c = (c * e->jump.A[j] + e->jump.C[j]) & mask;
      }
    jt[m].a = a;
    jt[m].c = c;
  }
  HIPCHK(e, e->d_fused_jump.reserve(kFusedWindow));
  H2D_OR_RETURN(h2d_from_host_sync(e, e->d_fused_jump.p, jt.data(), jt.size() * sizeof(FusedJump)));
  HIPCHK(e, e->d_fused_keys.reserve(kFusedWindow));
  HIPCHK(e, e->d_fused_counter.reserve(1));
  // on the engine's stream: a plain hipMemset may still be in flight when the first launch counts its blocks
  HIPCHK(e, hipMemsetAsync(e->d_fused_counter.p, 0, sizeof(unsigned), e->stream));
  return BPF_OK;
}

// Waits for a window kernel's three result words (fused_publish in kernels_fused.hpp) of `generation` in e->h_fused;
// res[0..4] = M, leaf count, bin count, status, levels.  BPF_OK with *seen = false when they do not show up.
int fused_result_wait(bpf_engine* e, int generation, int spin_ms, int res[5], bool* seen)
{
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(e->h_fused.p);
  const auto t0 = std::chrono::steady_clock::now();
  auto there = [&]() {
    for (int k = 0; k < 3; ++k)
      if ((unsigned)(__atomic_load_n(&w[k], __ATOMIC_ACQUIRE) >> 32) != (unsigned)generation)
        return false;
    return true;
  };
  *seen = false;
  for (unsigned spins = 0; !*seen; ++spins)
  {
    *seen = there();
    if (!*seen && (spins & 1023) == 1023 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(spin_ms))
      break;
    if (!*seen)
      __builtin_ia32_pause();
  }
  if (!*seen)
  {
    HIPCHK(e, hipStreamSynchronize(e->stream));  // (a mailbox wait may be running up to its bound)
    *seen = there();
  }
  if (*seen)
  {
    const unsigned long long w0 = w[0], w1 = w[1], w2 = w[2];
    res[0] = (int)(unsigned)w0;
    res[1] = (int)((w1 >> 16) & 0xFFFF);
    res[2] = (int)(w1 & 0xFFFF);
    res[3] = (int)((w2 >> 8) & 0xFF);
    res[4] = (int)(w2 & 0xFF);
  }
  return BPF_OK;
}

int resample_block(bpf_engine* e, int window, bool systematic, const double* targets, bool* handled)
{
  *handled = false;
  SampleSet& a = e->sets[e->cur];
  SampleSet& b = e->sets[e->cur ^ 1];
  if (!systematic)
  {
    int rc = ensure_limit_table(e, window);
    if (rc != BPF_OK)
      return rc;
  }
  HIPCHK(e, e->h_fused.reserve(32));
  if (!e->fused_lds_attr_set)
  {
    HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_resample_block),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds));
    e->fused_lds_attr_set = true;
  }
  int rcj = ensure_fused_jump(e);
  if (rcj != BPF_OK)
    return rcj;
  ResampleBlockArgs A{};
  A.src = a.dev();
  A.n_src = e->sample_count;
  A.cdf = e->d_cdf.p;
  A.coarse_shift = fused_coarse_shift(A.n_src);
  A.coarse = (e->cdf_coarse_n == A.n_src) ? e->d_cdf_coarse.p : nullptr;
  A.dst = b.dev();
  A.window = window;
  A.max_samples = e->max_samples;
  A.systematic = systematic ? 1 : 0;
  A.targets = targets;
  A.rng_state = e->rng;
  A.jump = e->d_fused_jump.p;
  A.keys = e->d_fused_keys.p;
  A.counter = e->d_fused_counter.p;
  A.limit = e->d_kld_limit.p;
  A.miss_flag = e->d_flags.p;
  A.thr = e->dist_threshold;
  A.sc = e->d_scalars.p;
  A.conv_count = e->d_flags.p + 1;
  A.result_host = e->h_fused.p;
  e->fused_generation = (e->fused_generation % 0x3fffffff) + 1;
  A.generation = e->fused_generation;
  A.debug = getenv("BPF_DEBUG") != nullptr;
  {
    ProfScope ps(e, BPF_K_DRAW);
    hipLaunchKernelGGL(k_resample_block, dim3(blocks_for(window, kFusedDrawsPerBlock)), dim3(1024), kFusedLds,
                       e->stream, A);
  }
  HIPCHK(e, hipGetLastError());
  // the block publishes its result words in pinned memory (a ~25 us kernel)
  int r5[5] = { 0, 0, 0, 0, 0 };
  bool seen = false;
  int rcw = fused_result_wait(e, A.generation, 50, r5, &seen);
  if (rcw != BPF_OK)
    return rcw;
  if (!seen)
    return e->fail(BPF_ERR_HIP, "k_resample_block did not publish its result");
  const int* res = e->h_fused.p;  // (the debug stamps, [6] and [8 ..])
  if (getenv("BPF_DEBUG"))
  {
    fprintf(stderr, "[resample block] window %d M %d leaf %d bins %d status %d levels %d; last block (10 ns ticks): load %d "
            "dedup %d tree %d scan %d tail %d = %d shader clocks\n", window, r5[0], r5[1], r5[2], r5[3], r5[4],
            res[11] - res[9], res[12] - res[11], res[13] - res[12], res[14] - res[13], res[15] - res[14], res[6]);
    fprintf(stderr, "   block 0 draw phase (if it was not the last block): stage %d search %d gather+key %d fence %d\n",
            res[28] - res[20], res[29] - res[28], res[30] - res[29], res[31] - res[30]);
  }
  if (r5[3] != BPF_FUSED_OK)
    return BPF_OK;
  e->sample_count = r5[0];
  e->kld_leaf = r5[1];
  e->kld_bins = r5[2];
  e->kld_device_used = true;
  e->fused_used = 1;
  e->resample_windows = 1;
  *handled = true;
  return BPF_OK;
}

// The KLD stop rule for the whole candidate stream [0, maxs) on the device (kernels_kld.hpp), keys in
// e->d_keys (AoS): grows the histogram tree level by level and scans the leaf count.
// Returns BPF_OK with *stop_out = stop count (or -1: no stop), *leaf_out / *bins_out at the final count;
// *handled = false when a key does not fit the 64-bit packing or the tree is deeper than the level budget
// (the caller then replays on the host as before).
int kld_tree_on_device(bpf_engine* e, int maxs, bool* handled, int* stop_out, int* leaf_out, int* bins_out,
                       bool whole_stream = false, bool allow_local = true)
{
  *handled = false;
  const int n = maxs;
  if (n >= (1 << 30))
    return BPF_OK;  // element indices 2v + side are folded as 32-bit tags
  int rcl = ensure_limit_table(e, n);
  if (rcl != BPF_OK)
    return rcl;
  unsigned table = 1024;
  while (table < 2u * (unsigned)n)
    table <<= 1;
  constexpr int kMaxLevels = 256;
  const int tiles = blocks_for(n, kKldTile);
  HIPCHK(e, e->d_kld_hkey.reserve(table));
  HIPCHK(e, e->d_kld_htmin.reserve(table));
  HIPCHK(e, e->d_kld_slot.reserve((size_t)n));
  HIPCHK(e, e->d_kld_cur.reserve((size_t)n));
  HIPCHK(e, e->d_kld_first.reserve((size_t)n));
  HIPCHK(e, e->d_kld_child.reserve((size_t)2 * n));
  HIPCHK(e, e->d_kld_delta.reserve((size_t)n));
  HIPCHK(e, e->d_kld_counts.reserve((size_t)n));
  HIPCHK(e, e->d_kld_tiles.reserve((size_t)tiles));
  HIPCHK(e, e->d_kld_flags.reserve(4 + kMaxLevels));
  HIPCHK(e, e->h_kld.reserve(4 + kMaxLevels + 16));  // (+ the pieces form's eight result words)
  ProfScope ps(e, BPF_K_DRAW);
  KldArgs K{};
  K.keys = e->d_keys.p;
  K.n = n;
  K.h_key = e->d_kld_hkey.p;
  K.h_tmin = e->d_kld_htmin.p;
  K.h_mask = table - 1;
  K.slot = e->d_kld_slot.p;
  K.cur = e->d_kld_cur.p;
  K.first = e->d_kld_first.p;
  K.child = e->d_kld_child.p;
  K.delta = e->d_kld_delta.p;
  K.flags = e->d_kld_flags.p;
  K.limit = e->d_kld_limit.p;
  const dim3 grid(blocks_for(n, 256)), block(256);
  const bool local = e->kld_local && allow_local && !e->kld_persistent;
  // (pieces form: the previous build cleared the tables behind its result, while the host was turning around)
  const bool clean = local && e->kld_clean_table == table && e->kld_clean_key == e->d_kld_hkey.p &&
                     e->kld_clean_tmin == e->d_kld_htmin.p && e->kld_clean_cap == e->d_kld_hkey.cap;
  e->kld_clean_table = 0;
  const dim3 clear_grid(std::min(1024, blocks_for((int)std::max<unsigned>(table, 2u * (unsigned)n), 256)));
  if (!clean)
    hipLaunchKernelGGL(k_kld_clear, clear_grid, block, 0, e->stream, K, table, 4 + kMaxLevels, local ? 0 : 1);
  hipLaunchKernelGGL(k_kld_hash, grid, block, 0, e->stream, K);
  if (e->kld_persistent)
  {
    e->kld_last_form = 3;
    // the whole stream in one resident round of 1024-thread blocks: the level loop, the prefix sums and the stop test
    // run in ONE launch with grid barriers between the levels (k_kld_tree_persistent)
    const int pgrid = blocks_for(n, kKldBlock);
    if (e->kld_persist_blocks_per_cu < 0)
    {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_kld_tree_persistent),
                                                       kKldBlock, 0) != hipSuccess)
        nb = 0;
      e->kld_persist_blocks_per_cu = nb;
    }
    if (pgrid <= e->kld_persist_blocks_per_cu * e->n_cu)
    {
      HIPCHK(e, e->d_kld_bar.reserve(4));
      HIPCHK(e, e->d_kld_tiles.reserve((size_t)std::max(pgrid, tiles)));
      HIPCHK(e, hipMemsetAsync(e->d_kld_bar.p, 0, 4 * sizeof(unsigned), e->stream));
      KldPersistArgs P{};
      P.K = K;
      P.bar = e->d_kld_bar.p;
      P.level_waiting = e->d_kld_flags.p + 4;
      P.tile_sums = e->d_kld_tiles.p;
      P.max_levels = kMaxLevels;
      P.whole_stream = whole_stream ? 1 : 0;
      P.timeout_ticks = 2000000ll;  // 20 ms per barrier: a resident grid passes one in microseconds
      P.result_host = e->h_kld.p;
      e->kld_generation = (e->kld_generation % 0x3fffffff) + 1;
      P.generation = e->kld_generation;
      e->h_kld.p[0] = 0;  // (the level-per-launch form copies its flag words here)
      hipLaunchKernelGGL(k_kld_tree_persistent, dim3(pgrid), dim3(kKldBlock), 0, e->stream, P);
      HIPCHK(e, hipGetLastError());
      const auto t0 = std::chrono::steady_clock::now();
      bool seen = false;
      for (unsigned spins = 0; !seen; ++spins)
      {
        seen = __atomic_load_n(e->h_kld.p, __ATOMIC_ACQUIRE) == P.generation;
        if (!seen && (spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200))
          break;
        if (!seen)
          __builtin_ia32_pause();
      }
      if (!seen)
      {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (__atomic_load_n(e->h_kld.p, __ATOMIC_ACQUIRE) != P.generation)
          return e->fail(BPF_ERR_HIP, "k_kld_tree_persistent did not publish its result");
      }
      const int* res = e->h_kld.p;
      if (getenv("BPF_DEBUG"))
        fprintf(stderr, "[kld persistent] n %d blocks %d stop %d leaf %d bins %d status %d levels %d\n", n, pgrid, res[1],
                res[2], res[3], res[4], res[5]);
      if (res[4] == BPF_KLD_PERSIST_OK)
      {
        *stop_out = res[1];
        *leaf_out = res[2];
        *bins_out = res[3];
        *handled = true;
        return BPF_OK;
      }
      if (res[4] != BPF_KLD_PERSIST_TIMEOUT)
        return BPF_OK;  // a key outside the packing, or deeper than the level budget: not handled (host replay)
      // the grid was not resident as a whole (the GPU is shared with another process): let the launch drain, leave
      // this form alone from now on and build the tree again with one launch pair per level
      HIPCHK(e, hipStreamSynchronize(e->stream));
      e->kld_persistent = false;
      return kld_tree_on_device(e, maxs, handled, stop_out, leaf_out, bins_out, whole_stream);
    }
  }
  e->kld_last_form = local ? 2 : 1;
  if (local)
  {
    // the tree in LDS-sized pieces (kernels_kld2.hpp): no level loop, no host round trip until the result
    const size_t nn = (size_t)n;
    const int n_tiles = blocks_for(n, 256);
    HIPCHK(e, e->d_kld2_int.reserve(5 * nn + 3 * (size_t)kKld2Nodes + 8 + (size_t)n_tiles));
    HIPCHK(e, e->d_kld2_top.reserve((size_t)kKld2Nodes));
    if (!e->kld2_attr_set)
    {
      HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kld2_top),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kKld2LdsBytes));
      HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kld2_subtrees),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kKld2LdsBytes));
      e->kld2_attr_set = true;
    }
    Kld2Args L{};
    L.K = K;
    L.pk = reinterpret_cast<unsigned long long*>(e->d_kld2_int.p);  // first: 8-byte aligned
    int* base = e->d_kld2_int.p + 2 * nn;
    L.tkeys = base;
    L.bucket = base + nn;
    L.bk = base + 2 * nn;
    L.cnt = base + 3 * nn;
    L.off = L.cnt + kKld2Nodes;
    L.fill = L.off + kKld2Nodes + 1;
    L.n_tkeys = L.fill + kKld2Nodes;
    L.n_top = L.n_tkeys + 1;
    L.status = L.n_top + 1;
    int* tile_first = L.status + 4;
    L.tile_first = tile_first;
    L.n_tiles = n_tiles;
    L.top = e->d_kld2_top.p;
    L.counts = e->d_kld_counts.p;
    L.whole_stream = whole_stream ? 1 : 0;
    e->kld_generation = (e->kld_generation % 0x3fffffff) + 1;
    L.generation = e->kld_generation;
    // (eight 64-bit words behind the level loop's flag words in the pinned block)
    static_assert((4 + kMaxLevels) % 2 == 0, "the result words are 8-byte aligned");
    volatile unsigned long long* words = reinterpret_cast<volatile unsigned long long*>(e->h_kld.p + 4 + kMaxLevels);
    L.result_host = e->h_kld.p + 4 + kMaxLevels;
    int res[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    hipLaunchKernelGGL(k_kld2_init, dim3(n_tiles), dim3(256), 0, e->stream, K, tile_first);
    hipLaunchKernelGGL(k_kld2_compact, dim3(n_tiles), dim3(256), 0, e->stream, L);
    hipLaunchKernelGGL(k_kld2_top, dim3(1), dim3(kKld2Block), kKld2LdsBytes, e->stream, L);
    hipLaunchKernelGGL(k_kld2_route, dim3(blocks_for(n, kKld2Block)), dim3(kKld2Block), 0, e->stream, L);
    hipLaunchKernelGGL(k_kld2_offsets, dim3(1), dim3(1024), 0, e->stream, L);
    hipLaunchKernelGGL(k_kld2_scatter, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, L);
    hipLaunchKernelGGL(k_kld2_subtrees, dim3(blocks_for(n, kKld2Span) + 1), dim3(kKld2Block), kKld2LdsBytes, e->stream,
                       L);
    // prefix sums + stop test in one launch (look-back slots tagged with the generation: reserved and zeroed once)
    if (e->d_kld2_slots.cap < (size_t)tiles)
    {
      HIPCHK(e, e->d_kld2_slots.reserve((size_t)std::max(tiles, 1024)));
      HIPCHK(e, hipMemsetAsync(e->d_kld2_slots.p, 0, e->d_kld2_slots.cap * sizeof(unsigned long long), e->stream));
    }
    hipLaunchKernelGGL(k_kld2_scan, dim3(tiles), dim3(256), 0, e->stream, K, e->d_kld2_slots.p,
                       (unsigned)L.generation, e->d_kld_counts.p, L.status);
    hipLaunchKernelGGL(k_kld2_result, dim3(1), dim3(64), 0, e->stream, L);
    // the next build's start state now, behind the result: the GPU does it while this thread turns around
    hipLaunchKernelGGL(k_kld_clear, dim3(std::min(1024, blocks_for((int)table, 256))), block, 0, e->stream, K, table,
                       4 + kMaxLevels, 0);
    HIPCHK(e, hipGetLastError());
    // the result block in pinned memory, every word tagged with the generation: one poll instead of three copies
    // with a stream synchronisation each
    {
      auto all_there = [&]() {
        for (int k = 1; k < 8; ++k)
        {
          const unsigned long long w = __atomic_load_n(const_cast<unsigned long long*>(words + k), __ATOMIC_ACQUIRE);
          if ((unsigned)(w >> 32) != (unsigned)L.generation)
            return false;
          res[k] = (int)(unsigned)w;
        }
        return true;
      };
      const auto t0 = std::chrono::steady_clock::now();
      bool seen = false;
      for (unsigned spins = 0; !seen; ++spins)
      {
        seen = all_there();
        if (!seen && (spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(100))
          break;
        if (!seen)
          __builtin_ia32_pause();
      }
      if (!seen)
      {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (!all_there())
          return e->fail(BPF_ERR_HIP, "k_kld2_result did not publish its result");
      }
    }
    if (getenv("BPF_DEBUG"))
      fprintf(stderr, "[kld pieces] n %d tree keys %d status %d largest bucket %d stop %d leaf %d bins %d\n", n, res[6],
              res[5], res[7], res[2], res[3], res[4]);
    e->kld_clean_table = table;
    e->kld_clean_key = e->d_kld_hkey.p;
    e->kld_clean_tmin = e->d_kld_htmin.p;
    e->kld_clean_cap = e->d_kld_hkey.cap;
    if (res[1] != 0)
      return BPF_OK;  // a key outside the packing range: not handled
    if (res[5] != BPF_KLD2_OK)  // a bucket that does not fit a block, or a piece deeper than its budget
      return kld_tree_on_device(e, maxs, handled, stop_out, leaf_out, bins_out, whole_stream, false);
    *stop_out = res[2];
    *leaf_out = res[3];
    *bins_out = res[4];
    *handled = true;
    return BPF_OK;
  }
  hipLaunchKernelGGL(k_kld_init, grid, block, 0, e->stream, K);
  hipLaunchKernelGGL(k_kld_root_first, dim3(1), dim3(1024), 0, e->stream, K);
  int level = 0;
  bool done = false;
  int batch = std::min(24, (int)std::ceil(1.3 * std::log2((double)n + 1.0)) + 1);  // (see below)
  while (!done && level < kMaxLevels)
  {
    for (int q = 0; q < batch && level < kMaxLevels; ++q, ++level)
    {
      hipLaunchKernelGGL(k_kld_children, dim3(blocks_for(n, kKldBlock)), dim3(kKldBlock), 0, e->stream, K);
      hipLaunchKernelGGL(k_kld_descend, dim3(blocks_for(n, kKldBlock)), dim3(kKldBlock), 0, e->stream, K,
                         e->d_kld_flags.p + 4 + level);
    }
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipMemcpyAsync(e->h_kld.p, e->d_kld_flags.p, (4 + kMaxLevels) * sizeof(int), hipMemcpyDeviceToHost,
                             e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->h_kld.p[0] != 0)
      return BPF_OK;  // a key outside the packing range: not handled
    const int waiting = e->h_kld.p[4 + level - 1];  // keys that are not nodes yet
    done = waiting == 0;
    if (done)
      break;
    // A random-order tree of K keys is ~2.5 log2 K levels deep and the count of waiting keys falls faster and faster
    // near the end: ~1.3 log2(waiting) more levels finish it (measured on 59 k and 97 k keys).  (One block taking over
    // the last few thousand keys was tried: a level is ~6 dependent round trips to first[] / child[] either way, 8 us
    // in one block against 14 us as a launch pair, and the switch costs what it saves.)
    batch = std::min(24, (int)std::ceil(1.3 * std::log2((double)waiting + 1.0)) + 1);
  }
  if (getenv("BPF_DEBUG"))
  {
    fprintf(stderr, "[kld device] n %d levels %d done %d; keys waiting after each level:", n, level, (int)done);
    for (int l = 0; l < level; ++l)
      fprintf(stderr, " %d", e->h_kld.p[4 + l]);
    fprintf(stderr, "\n");
  }
  if (!done)
    return BPF_OK;  // deeper than the level budget: not handled
  hipLaunchKernelGGL(k_kld_scan_tiles, dim3(tiles), dim3(256), 0, e->stream, (const int2*)e->d_kld_delta.p, n,
                     e->d_kld_tiles.p);
  hipLaunchKernelGGL(k_kld_scan_offsets, dim3(1), dim3(1024), 0, e->stream, e->d_kld_tiles.p, tiles);
  hipLaunchKernelGGL(k_kld_scan_final, dim3(tiles), dim3(256), 0, e->stream, K, (const int2*)e->d_kld_tiles.p,
                     e->d_kld_counts.p);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(e->h_kld.p, e->d_kld_flags.p, 4 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  const int stop = whole_stream ? -1 : e->h_kld.p[2];  // whole_stream: the tree of all n keys, no stop rule
  const int M = (stop >= 1 && stop <= n) ? stop : n;
  int2 c;
  HIPCHK(e, hipMemcpyAsync(&c, e->d_kld_counts.p + (M - 1), sizeof(int2), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  *stop_out = (stop >= 1 && stop <= n) ? stop : -1;
  *leaf_out = c.x;
  *bins_out = c.y;
  *handled = true;
  return BPF_OK;
}

// the whole candidate stream [0, maxs) again with the keys on the device only, then the tree
int kld_on_device(bpf_engine* e, DrawArgs A, int maxs, bool* handled, int* stop_out, int* leaf_out, int* bins_out)
{
  A.m0 = 0;
  A.m1 = maxs;
  A.host_keys = nullptr;
  {
    ProfScope ps(e, BPF_K_DRAW);
    hipLaunchKernelGGL(k_draw_select, dim3(blocks_for(maxs, 256)), dim3(256), 0, e->stream, A);
  }
  return kld_tree_on_device(e, maxs, handled, stop_out, leaf_out, bins_out);
}

// Spin on the generation word a kernel publishes in pinned host memory (kernels of ~10 us); false if it
// takes implausibly long, and the caller falls back to a copy + stream synchronisation.
bool wait_generation(bpf_engine* e, unsigned generation)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins)
  {
    if (__atomic_load_n(e->h_done.p, __ATOMIC_ACQUIRE) == generation)
      return true;
    if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
      return false;
    __builtin_ia32_pause();
  }
}

int resample_multinomial(bpf_engine* e, double w_diff)
{
  SampleSet& a = e->sets[e->cur];
  SampleSet& b = e->sets[e->cur ^ 1];
  const int n = e->sample_count;
  const int maxs = e->max_samples;
  FreeSpaceDev free_space{};
  const int* chain = nullptr;
  if (w_diff > 0.0)
  {
    // :383-388: every draw first tests a uniform against w_diff; where each draw finds its stream elements is
    // resolved up front for all candidate draws
    int rcf = ensure_free_space(e, &free_space);
    if (rcf != BPF_OK)
      return rcf;
    rcf = build_draw_chain(e, w_diff, maxs);
    if (rcf != BPF_OK)
      return rcf;
    chain = e->d_chain.p;
  }
  e->fused_used = 0;
  {
    // tracking regime: the previous cycle's stop (+ 25 %) fits one block's window and no recovery draws are due
    const int w0 = std::min(std::max(1024, std::min(e->window_hint, maxs)), maxs);
    const bool long_stream0 = e->window_hint >= maxs && maxs >= e->kld_device_min;
    if (e->fused_resample && w_diff == 0.0 && w0 <= kFusedWindow && !long_stream0)
    {
      bool handled = false;
      int rcb = resample_block(e, w0, false, nullptr, &handled);
      if (rcb != BPF_OK)
        return rcb;
      if (handled)
      {
        const int M = e->sample_count;
        e->rng = lcg_skip_host(e->rng, 2ull * (uint64_t)M, e->jump);
        e->window_hint = std::max(1024, ((M + M / 4) + 1023) / 1024 * 1024);
        return BPF_OK;
      }
    }
  }
  HIPCHK(e, e->d_keys.reserve((size_t)maxs * 3));
  HIPCHK(e, e->d_src_index.reserve((size_t)maxs));
  HIPCHK(e, e->h_keys.reserve((size_t)maxs * 3));
  e->hist.clear();
  e->seen.reset((size_t)std::min(maxs, 1 << 20));
  int m0 = 0, stop = -1;
  int window = std::max(1024, std::min(e->window_hint, maxs));
  e->resample_windows = 0;
  int cached_leaf = -1, cached_limit = 0;
  e->kld_device_used = false;
  bool device_declined = false;
  // keep the host's first window short when the device tree can take over after it
  if (window > 4096 && maxs - 4096 >= e->kld_device_min)
    window = 4096;
  while (m0 < maxs && stop < 0)
  {
    const int m1 = std::min(maxs, m0 + window);
    DrawArgs A{};
    A.src = a.dev();
    A.n_src = n;
    A.cdf = e->d_cdf.p;
    A.dst = b.dev();
    A.m0 = m0;
    A.m1 = m1;
    A.rng_state = e->rng;
    A.jump = e->jump;
    A.keys = e->d_keys.p;
    A.src_index = e->d_src_index.p;
    A.miss_flag = e->d_flags.p;
    A.sharded = 0;
    A.chain = chain;
    A.free_space = free_space;
    A.guide = e->cdf_guide_valid ? e->d_cdf_guide.p : nullptr;
    // long stream ahead: the previous cycle ran to the end, or the windows so far found no stop and the bound for
    // the leaves seen so far (a lower estimate of where the stop will be) is still far away
    const bool long_stream = (m0 > 0 ? cached_limit - m0 >= e->kld_device_min : e->window_hint >= maxs) &&
                             maxs - m0 >= e->kld_device_min;
    if (long_stream && !device_declined)
    {
      // no stop inside the first window and a long stream ahead (a spread cloud): the ordered replay moves
      // to the device for the whole stream
      bool handled = false;
      int dstop = -1, dleaf = 0, dbins = 0;
      int rc = kld_on_device(e, A, maxs, &handled, &dstop, &dleaf, &dbins);
      if (rc != BPF_OK)
        return rc;
      if (handled)
      {
        e->resample_windows++;
        stop = dstop;
        e->kld_device_used = true;
        e->kld_leaf = dleaf;
        e->kld_bins = dbins;
        break;
      }
      device_declined = true;  // key range or depth outside what the device tree takes: host replay as before
    }
    const int wn = m1 - m0;
    // Keys go straight into pinned host memory and the last block publishes a generation number
    // there: the host polls that word instead of paying for a copy plus a stream synchronisation.
    const bool zero_copy = e->zero_copy_keys && wn <= (1 << 20);
    if (zero_copy)
    {
      A.host_keys = e->h_keys.p;
      A.host_stride = wn;
      A.done_counter = reinterpret_cast<unsigned*>(e->d_flags.p + 4);
      A.host_done = reinterpret_cast<volatile unsigned*>(e->h_done.p);
      A.generation = ++e->done_generation;
    }
    {
      ProfScope ps(e, BPF_K_DRAW);
      hipLaunchKernelGGL(k_draw_select, dim3(blocks_for(wn, 256)), dim3(256), 0, e->stream, A);
    }
    HIPCHK(e, hipGetLastError());
    const bool have_keys = zero_copy && wait_generation(e, A.generation);
    int k_stride = wn;
    if (!have_keys)
    {
      HIPCHK(e, hipMemcpyAsync(e->h_keys.p, e->d_keys.p, (size_t)wn * 3 * sizeof(int), hipMemcpyDeviceToHost,
                               e->stream));
      HIPCHK(e, hipStreamSynchronize(e->stream));
      k_stride = 0;  // AoS triples from the device buffer
    }
    e->resample_windows++;
    const int* keys = e->h_keys.p;
    for (int m = m0; m < m1; ++m)
    {
      const int o = m - m0;
      const int k[3] = { k_stride ? keys[o] : keys[3 * o], k_stride ? keys[k_stride + o] : keys[3 * o + 1],
                         k_stride ? keys[2 * k_stride + o] : keys[3 * o + 2] };
      if (e->seen.first_time(k[0], k[1], k[2]))
      {
        e->hist.insert(k[0], k[1], k[2]);
        const int lc = e->hist.leaf_count();
        if (lc != cached_leaf)
        {
          cached_leaf = lc;
          cached_limit = resample_limit(lc, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
        }
      }
      if (m + 1 > cached_limit)  // particle_filter.cpp:416
      {
        stop = m + 1;
        break;
      }
    }
    m0 = m1;
    // next window: up to a quarter past the bound for the leaves seen so far
    const int need = cached_limit - m0;
    window = std::max(1024, (need + need / 4 + 1023) / 1024 * 1024);
  }
  const int M = (stop > 0) ? stop : maxs;
  // the window that found the stop also inserted nothing past it: hist is exactly set b's tree
  if (chain != nullptr)
  {
    // the stream was consumed up to the element before draw M's test
    HIPCHK(e, e->h_chain_word.reserve(1));
    HIPCHK(e, hipMemcpyAsync(e->h_chain_word.p, chain + M, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const unsigned next_test = (unsigned)e->h_chain_word.p[0] & 0x7fffffffu;
    e->rng = lcg_skip_host(e->rng, (uint64_t)next_test - 1ull, e->jump);
  }
  else
    e->rng = lcg_skip_host(e->rng, 2ull * (uint64_t)M, e->jump);
  e->window_hint = std::max(1024, ((M + M / 4) + 1023) / 1024 * 1024);
  e->sample_count = M;
  return BPF_OK;
}

int resample_systematic(bpf_engine* e, double w_diff)
{
  SampleSet& a = e->sets[e->cur];
  SampleSet& b = e->sets[e->cur ^ 1];
  const int n = e->sample_count;
  e->fused_used = 0;
  int count = resample_limit(e->leaf_count, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
  FreeSpaceDev free_space{};
  int num_random = 0;
  if (w_diff > 0.0)
  {
    // particle_filter.cpp:295-306: room for random poses on top of the systematic ones
    count *= (1.0 + w_diff);
    if (count > e->max_samples)
      count = e->max_samples;
    num_random = (int)(w_diff * count);
    if (num_random > 0)
    {
      int rcf = ensure_free_space(e, &free_space);
      if (rcf != BPF_OK)
        return rcf;
    }
  }
  const int num_systematic = count - num_random;
  const uint64_t rng_before = e->rng;
  e->rng = lcg_skip_host(e->rng, 1, e->jump);
  const double start = std::ldexp((double)e->rng, -48);
  const double delta = 1.0 / num_systematic;
  HIPCHK(e, e->d_keys.reserve((size_t)e->max_samples * 3));
  HIPCHK(e, e->d_src_index.reserve((size_t)e->max_samples));
  HIPCHK(e, e->h_keys.reserve((size_t)e->max_samples * 3));
  SystematicArgs A{};
  A.src = a.dev();
  A.n_src = n;
  A.cdf = e->d_cdf.p;
  A.dst = b.dev();
  A.count = count;
  A.n_random = num_random;
  A.rng_state = rng_before;
  A.jump = e->jump;
  A.free_space = free_space;
  A.keys = e->d_keys.p;
  A.src_index = e->d_src_index.p;
  A.miss_flag = e->d_flags.p;
  // The targets are a serial floating-point chain (particle_filter.cpp:337-341): target += delta, and
  // target -= 1 once it passes 1.  A CPU core runs that dependency chain several times faster than a
  // GPU lane, with the same IEEE arithmetic, so the host forms the targets and uploads them.
  HIPCHK(e, e->h_targets.reserve((size_t)e->max_samples));
  {
    double t = start;
    double* out = e->h_targets.p;
    for (int i = 0; i < num_systematic; ++i)
    {
      out[i] = t;
      t += delta;
      if (t > 1.0)
        t -= 1.0;
    }
  }
  if (e->fused_resample && num_random == 0 && count <= kFusedWindow)
  {
    // the whole resample as one launch (k_resample_block): selection, the new set's histogram tree, weights,
    // updateConverged; the targets are read straight from the pinned buffer
    bool handled = false;
    int rcb = resample_block(e, count, true, e->h_targets.p, &handled);
    if (rcb != BPF_OK)
      return rcb;
    if (handled)
      return BPF_OK;
  }
  // the kernel reads the targets straight from the pinned buffer (28 KB for 3.5 k samples) and, like the
  // multinomial draw kernel, leaves the keys in pinned memory behind a generation word
  A.targets = e->h_targets.p;
  const bool zero_copy = e->zero_copy_keys && count <= (1 << 20);
  if (zero_copy)
  {
    A.host_keys = e->h_keys.p;
    A.host_stride = count;
    A.done_counter = reinterpret_cast<unsigned*>(e->d_flags.p + 4);
    A.host_done = reinterpret_cast<volatile unsigned*>(e->h_done.p);
    A.generation = ++e->done_generation;
  }
  {
    ProfScope ps(e, BPF_K_DRAW);
    hipLaunchKernelGGL(k_systematic_select, dim3(blocks_for(count, 256)), dim3(256), 0, e->stream, A);
  }
  HIPCHK(e, hipGetLastError());
  int k_stride = count;
  if (!(zero_copy && wait_generation(e, A.generation)))
  {
    if (zero_copy)  // the kernel wrote the host rows; wait for it the slow way
      HIPCHK(e, hipStreamSynchronize(e->stream));
    else
    {
      HIPCHK(e, hipMemcpyAsync(e->h_keys.p, e->d_keys.p, (size_t)count * 3 * sizeof(int), hipMemcpyDeviceToHost,
                               e->stream));
      HIPCHK(e, hipStreamSynchronize(e->stream));
      k_stride = 0;
    }
  }
  e->hist.clear();
  e->seen.reset((size_t)std::min(count, 1 << 20));
  const int* keys = e->h_keys.p;
  for (int m = 0; m < count; ++m)
  {
    const int k0 = k_stride ? keys[m] : keys[3 * m], k1 = k_stride ? keys[k_stride + m] : keys[3 * m + 1],
              k2 = k_stride ? keys[2 * k_stride + m] : keys[3 * m + 2];
    if (e->seen.first_time(k0, k1, k2))
      e->hist.insert(k0, k1, k2);
  }
  // :316-324: the random poses took two uniforms each, right after the systematic start
  e->rng = lcg_skip_host(e->rng, 2ull * (uint64_t)num_random, e->jump);
  e->resample_windows = 1;
  e->sample_count = count;
  return BPF_OK;
}

int upload_samples(bpf_engine* e, const double* aos, int n, SampleSet& dst)
{
  HIPCHK(e, e->d_aos.reserve((size_t)n));
  HIPCHK(e, dst.reserve((size_t)n));
  // the caller's (pageable) buffer straight to the runtime, which pipelines its own bounce buffers: a staging memcpy of
  // the whole set on this thread followed by one DMA serialises the two (0.37 against 0.31 ms per update at 100 k)
  H2D_OR_RETURN(h2d_from_host(e, e->d_aos.p, aos, (size_t)n * sizeof(double4), e->stream));
  hipLaunchKernelGGL(k_aos_to_soa, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->d_aos.p, dst.dev(), n);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

// the weights of a device-side set back into the caller's AoS records (only the weight of a record changes in a
// sensor update): 8 bytes per particle over PCIe instead of 32, no re-interleaving launch
int download_weights(bpf_engine* e, const SampleSet& src, int n, double* aos)
{
  HIPCHK(e, e->h_aos.reserve((size_t)n));
  double* hw = reinterpret_cast<double*>(e->h_aos.p);
  HIPCHK(e, hipMemcpyAsync(hw, src.w.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->h_scalars.p, e->d_scalars.p, sizeof(FilterScalars), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (int i = 0; i < n; ++i)
    aos[4 * (size_t)i + 3] = hw[i];
  return BPF_OK;
}
