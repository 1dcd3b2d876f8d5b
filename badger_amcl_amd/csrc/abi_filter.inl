// C-ABI: particle filter.
// ---------------------------------------------------------------------- particle filter
int bpf_pf_create(bpf_engine* e, int min_samples, int max_samples, double alpha_slow, double alpha_fast,
                  double global_localization_convergence_threshold)
{
  if (!e || max_samples <= 0 || min_samples < 0)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad filter sizes") : BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  int rc = ensure_scalars(e);
  if (rc != BPF_OK)
    return rc;
  e->min_samples = min_samples;
  e->max_samples = max_samples;
  e->alpha_slow = alpha_slow;
  e->alpha_fast = alpha_fast;
  e->conv_threshold = global_localization_convergence_threshold;
  e->pop_err = 0.01;
  e->pop_z = 3;
  e->dist_threshold = 0.5;
  e->resample_model = BPF_RESAMPLE_MULTINOMIAL;
  e->random_pose_mode = BPF_RANDOM_POSE_NONE;  // the constructor's random_pose_fn: none until one is set
  for (int k = 0; k < 2; ++k)
    HIPCHK(e, e->sets[k].reserve((size_t)max_samples));
  // ctor state (particle_filter.cpp:62-89): max_samples particles at the origin, weight 1/max
  e->cur = 0;
  e->sample_count = max_samples;
  HIPCHK(e, hipMemsetAsync(e->sets[0].x.p, 0, (size_t)max_samples * sizeof(double), e->stream));
  HIPCHK(e, hipMemsetAsync(e->sets[0].y.p, 0, (size_t)max_samples * sizeof(double), e->stream));
  HIPCHK(e, hipMemsetAsync(e->sets[0].th.p, 0, (size_t)max_samples * sizeof(double), e->stream));
  hipLaunchKernelGGL(k_fill, dim3(blocks_for(max_samples, 256)), dim3(256), 0, e->stream, e->sets[0].w.p,
                     1.0 / max_samples, max_samples);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemsetAsync(e->d_scalars.p, 0, sizeof(FilterScalars), e->stream));
  e->leaf_count = 0;
  e->bin_count = 0;
  e->converged = 0;
  e->converged_pending = false;
  e->window_hint = 4096;
  e->have_pf = true;
  return BPF_OK;
}

int bpf_pf_set_resample_model(bpf_engine* e, int resample_model)
{
  if (!e || (resample_model != BPF_RESAMPLE_MULTINOMIAL && resample_model != BPF_RESAMPLE_SYSTEMATIC))
    return BPF_ERR_INVALID_ARGUMENT;
  e->resample_model = resample_model;
  return BPF_OK;
}

int bpf_pf_set_population_size_parameters(bpf_engine* e, double pop_err, double pop_z)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->pop_err = pop_err;
  e->pop_z = pop_z;
  return BPF_OK;
}

int bpf_pf_set_decay_rates(bpf_engine* e, double alpha_slow, double alpha_fast)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->alpha_slow = alpha_slow;
  e->alpha_fast = alpha_fast;
  return BPF_OK;
}

int bpf_pf_srand48(bpf_engine* e, long seed)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->rng = ((((uint64_t)seed) & 0xFFFFFFFFull) << 16) | 0x330Eull;
  return BPF_OK;
}

int bpf_pf_set_rng_state(bpf_engine* e, uint64_t state48)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->rng = state48 & ((1ull << 48) - 1);
  return BPF_OK;
}

int bpf_pf_get_rng_state(const bpf_engine* e, uint64_t* state48)
{
  if (!e || !state48)
    return BPF_ERR_INVALID_ARGUMENT;
  *state48 = e->rng;
  return BPF_OK;
}

namespace
{
int finish_init(bpf_engine* e, int n);  // abi_motion.inl
int ensure_set_tree(bpf_engine* e);     // abi_motion.inl
}

int bpf_pf_set_samples(bpf_engine* e, const double* samples, int sample_count, int leaf_count)
{
  if (!e || !samples)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (sample_count <= 0 || sample_count > e->max_samples)
    return e->fail(BPF_ERR_CAPACITY, "sample_count outside (0, max_samples]");
  HIPCHK(e, hipSetDevice(e->device));
  (void)host_buffer_pinned(e, const_cast<double*>(samples), (size_t)sample_count * sizeof(double4));
  int rc = upload_samples(e, samples, sample_count, e->sets[e->cur]);
  if (rc != BPF_OK)
    return rc;
  e->sample_count = sample_count;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  e->set_epoch++;
  e->hist_matches_set = leaf_count < 0;
  // initWith*: w_slow_ = w_fast_ = 0, converged = false (particle_filter.cpp:127,157,164-168)
  HIPCHK(e, hipMemsetAsync(e->d_scalars.p, 0, sizeof(FilterScalars), e->stream));
  e->converged = 0;
  e->converged_pending = false;
  e->fused_partials = 0;
  e->tree_pending = false;
  e->spread_init = false;
  if (leaf_count >= 0)
  {
    e->leaf_count = leaf_count;
    e->bin_count = -1;
  }
  else
  {
    // the set's histogram tree, as the reference builds it when a set is created (on the device for large sets, as
    // after initWithGaussian / initWithPoseFn), is built when it is first needed: ensure_set_tree
    e->hist_matches_set = false;
    e->leaf_count = e->bin_count = -1;
    e->tree_pending = true;
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));  // the caller's buffer is only the call's
  return BPF_OK;
}

int bpf_pf_get_samples(bpf_engine* e, double* samples_out, int capacity, int* sample_count_out)
{
  if (!e || !samples_out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  const int n = e->sample_count;
  if (capacity < n)
    return e->fail(BPF_ERR_CAPACITY, "output too small");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->d_aos.reserve((size_t)n));
  HIPCHK(e, e->h_aos.reserve((size_t)n));
  hipLaunchKernelGGL(k_soa_to_aos, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->sets[e->cur].dev(),
                     e->d_aos.p, n);
  HIPCHK(e, hipGetLastError());
  if (host_buffer_pinned(e, samples_out, (size_t)n * sizeof(double4)))
  {
    // a registered buffer of the caller: the copy engine writes it directly
    HIPCHK(e, hipMemcpyAsync(samples_out, e->d_aos.p, (size_t)n * sizeof(double4), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
  }
  else
  {
    HIPCHK(e, hipMemcpyAsync(e->h_aos.p, e->d_aos.p, (size_t)n * sizeof(double4), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    std::memcpy(samples_out, e->h_aos.p, (size_t)n * sizeof(double4));
  }
  if (sample_count_out)
    *sample_count_out = n;
  return BPF_OK;
}

int bpf_pf_snapshot(bpf_engine* e)
{
  if (!e || !e->have_pf)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  {
    const int rct = ensure_set_tree(e);
    if (rct != BPF_OK)
      return rct;
  }
  const size_t n = (size_t)e->sample_count;
  HIPCHK(e, e->snap.reserve(n));
  SampleSet& s = e->sets[e->cur];
  HIPCHK(e, hipMemcpyAsync(e->snap.x.p, s.x.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->snap.y.p, s.y.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->snap.th.p, s.th.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->snap.w.p, s.w.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  e->snap_count = e->sample_count;
  e->snap_leaf = e->leaf_count;
  e->snap_bins = e->bin_count;
  return BPF_OK;
}

int bpf_pf_restore(bpf_engine* e)
{
  if (!e || !e->have_pf || e->snap_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  const int n = e->snap_count;
  hipLaunchKernelGGL(k_copy4, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->sets[e->cur].dev(),
                     e->snap.dev(), n);
  HIPCHK(e, hipGetLastError());
  e->sample_count = e->snap_count;
  e->tree_pending = false;
  e->leaf_count = e->snap_leaf;
  e->bin_count = e->snap_bins;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  e->set_epoch++;
  e->hist_matches_set = false;
  return BPF_OK;
}

int bpf_pf_fill_weights(bpf_engine* e, double weight)
{
  if (!e || !e->have_pf)
    return BPF_ERR_INVALID_ARGUMENT;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  e->set_epoch++;
  HIPCHK(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(k_fill, dim3(blocks_for(e->sample_count, 256)), dim3(256), 0, e->stream,
                     e->sets[e->cur].w.p, weight, e->sample_count);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

int bpf_pf_update_sensor_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                double range_max)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (e->pm.max_beams < 2)
    return BPF_OK;  // PlanarScanner::updateSensor returns false and touches nothing (:128-129)
  HIPCHK(e, hipSetDevice(e->device));
  if (e->pm.model == BPF_MODEL_LIKELIHOOD_FIELD_PROB && e->pm.do_beamskip && e->converged_pending)
  {
    int rc = fetch_scalars(e);
    if (rc != BPF_OK)
      return rc;
  }
  SampleSet& s = e->sets[e->cur];
  const int n = e->sample_count;
  bool forced_zero = false;
  int rc = score_planar(e, s.dev(), n, e->converged, ranges, angles, range_count, range_max, &forced_zero, true);
  if (rc != BPF_OK)
    return rc;
  if (e->fused_partials > 0)
  {
    // the scoring kernel left per-block weight partials: one launch folds them, normalises, updates the
    // running averages and leaves the tile sums for the CDF
    const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
    HIPCHK(e, e->d_tile_sums.reserve((size_t)nb));
    if (e->fused_resample && !e->cdf_serial && nb <= BPF_RED_BLOCK)
    {
      // ... and the resampling CDF in the same launch (kernels_fused.hpp): updateResample finds it ready
      rc = launch_normalize_cdf(e, s.w.p, n);
      if (rc != BPF_OK)
        return rc;
    }
    else
    {
      ProfScope ps(e, BPF_K_NORMALIZE);
      hipLaunchKernelGGL(k_normalize_fused, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, s.w.p, n,
                         e->d_block_partials.p, e->fused_partials, e->d_scalars.p, e->alpha_slow, e->alpha_fast,
                         e->d_tile_sums.p);
      HIPCHK(e, hipGetLastError());
      e->tile_sums_n = n;
    }
    e->fused_partials = 0;
  }
  else
  {
    rc = sum_into_slot(e, s.w.p, n, 0, 1, n);
    if (rc != BPF_OK)
      return rc;
    ProfScope ps(e, BPF_K_NORMALIZE);
    hipLaunchKernelGGL(k_normalize, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, s.w.p, n, e->d_scalars.p, 0,
                       0.0, n);
    HIPCHK(e, hipGetLastError());
  }
  e->last_status = BPF_OK;
  e->set_epoch++;
  return BPF_OK;
}

int bpf_pf_set_random_pose_generator(bpf_engine* e, int mode)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (mode != BPF_RANDOM_POSE_NONE && mode != BPF_RANDOM_POSE_FREE_SPACE_2D)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "unknown random pose generator");
  e->random_pose_mode = mode;
  return BPF_OK;
}

int bpf_pf_update_resample(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  // w_diff = max(0, 1 - w_fast/w_slow) (particle_filter.cpp:438-440).  With both decay rates
  // zero the two averages are always equal, so w_diff is 0 (or NaN before any update, which
  // the multinomial sampler treats as 0 too) and no read-back is needed.
  double w_diff = 0.0;
  if (e->alpha_slow != 0.0 || e->alpha_fast != 0.0)
  {
    int rc = fetch_scalars(e);
    if (rc != BPF_OK)
      return rc;
    const double ws = e->h_scalars.p->v[1], wf = e->h_scalars.p->v[2];
    w_diff = 1.0 - wf / ws;
    if (!(w_diff >= 0.0))
      w_diff = 0.0;
  }
  e->w_diff_last = w_diff;
  SampleSet& a = e->sets[e->cur];
  int rc = BPF_OK;
  if (e->resample_model == BPF_RESAMPLE_SYSTEMATIC)
  {
    rc = ensure_set_tree(e);  // its sample count comes from the current set's leaf count (particle_filter.cpp:276)
    if (rc != BPF_OK)
      return rc;
  }
  e->tree_pending = false;  // the multinomial resampler builds the new set's tree from its draws
  e->spread_init = false;   // from here on the resample's own outcome says whether the cloud is spread
  rc = build_cdf(e, a.w.p, e->sample_count);
  if (rc != BPF_OK)
    return rc;
  e->kld_device_used = false;
  rc = (e->resample_model == BPF_RESAMPLE_SYSTEMATIC) ? resample_systematic(e, w_diff)
                                                      : resample_multinomial(e, w_diff);
  if (rc != BPF_OK)
    return rc;
  if (w_diff > 0.0)  // "Reset averages, to avoid spiraling off into complete randomness" (particle_filter.cpp:453-455)
    HIPCHK(e, hipMemsetAsync(&e->d_scalars.p->v[1], 0, 2 * sizeof(double), e->stream));
  const int M = e->sample_count;
  SampleSet& b = e->sets[e->cur ^ 1];
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  e->cur ^= 1;
  e->leaf_count = e->kld_device_used ? e->kld_leaf : e->hist.leaf_count();
  e->bin_count = e->kld_device_used ? e->kld_bins : e->hist.bin_count();
  if (e->fused_used)
  {
    // k_resample_block already wrote the weights and counted the converged particles
    e->converged_pending = true;
    e->conv_n = M;
  }
  else if (M <= 8192)
  {
    // small resampled set: weights 1/M and updateConverged in one single-block launch
    ProfScope ps(e, BPF_K_FINALIZE);
    hipLaunchKernelGGL(k_resample_tail_small, dim3(1), dim3(1024), 0, e->stream, b.x.p, b.y.p, b.w.p, M,
                       e->dist_threshold, e->d_scalars.p, e->d_flags.p + 1);
    HIPCHK(e, hipGetLastError());
    e->converged_pending = true;
    e->conv_n = M;
  }
  else
  {
    {
      ProfScope ps(e, BPF_K_FINALIZE);
      // weight 1.0 each, total = M, then weight /= total (particle_filter.cpp:409,458-462)
      hipLaunchKernelGGL(k_fill, dim3(blocks_for(M, 256)), dim3(256), 0, e->stream, b.w.p, 1.0 / (double)M, M);
    }
    HIPCHK(e, hipGetLastError());
    rc = launch_converged(e);
    if (rc != BPF_OK)
      return rc;
  }
  // miss flag was copied? read it with the next fetch; report asynchronously via last_status
  e->last_status = BPF_OK;
  e->set_epoch++;
  e->hist_matches_set = !e->kld_device_used;  // the device tree leaves no host histogram behind
  return BPF_OK;
}

int bpf_pf_get_state(bpf_engine* e, bpf_pf_state* out)
{
  if (!e || !out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  int rc = ensure_set_tree(e);
  if (rc != BPF_OK)
    return rc;
  rc = fetch_scalars(e);
  if (rc != BPF_OK)
    return rc;
  if (e->h_flags.p[0] != 0 && e->last_status == BPF_OK)
    e->last_status = BPF_ERR_CDF_MISS;
  std::memset(out, 0, sizeof(*out));
  out->sample_count = e->sample_count;
  out->leaf_count = e->leaf_count;
  out->bin_count = e->bin_count;
  out->converged = e->converged;
  out->percent_converged = e->percent_converged;
  out->total = e->h_scalars.p->v[0];
  out->w_slow = e->h_scalars.p->v[1];
  out->w_fast = e->h_scalars.p->v[2];
  out->w_diff = e->w_diff_last;
  out->last_status = e->last_status;
  out->resample_windows = e->resample_windows;
  out->kld_on_device = e->fused_used ? 2 : (e->kld_device_used ? 1 : 0);
  out->reserved = 0;
  out->evals = e->evals_last;
  return BPF_OK;
}
