// C-ABI: motion update.
// ---------------------------------------------------------------------- motion update
namespace
{
double odom_angle_diff(double a, double b)
{
  // Odom::angleDiff (odom.cpp:308-311) = angles::shortest_angular_distance(b, a) = normalize_angle(a - b);
  // angles::normalize_angle in its Noetic form (third party)
  const double r = std::fmod((a - b) + M_PI, 2.0 * M_PI);
  return (r <= 0.0) ? r + M_PI : r - M_PI;
}

// the loop-invariant part of Odom::updateAction, with the host libm like the reference
MotionModelDev motion_constants(const bpf_engine* e, const double pose[3], const double delta[3],
                                const double absolute_motion[3])
{
  MotionModelDev M{};
  M.model = e->odom_model;
  const double a1 = e->odom_alpha[0], a2 = e->odom_alpha[1], a3 = e->odom_alpha[2], a4 = e->odom_alpha[3],
               a5 = e->odom_alpha[4];
  const double old_th = pose[2] - delta[2];  // odom.cpp:82-85
  const double delta_trans = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
  M.delta_trans = delta_trans;
  M.delta_rot = delta[2];
  M.half_rot = delta[2] / 2;
  M.bearing0 = odom_angle_diff(std::atan2(delta[1], delta[0]), old_th);
  if (M.model == BPF_ODOM_MODEL_OMNI || M.model == BPF_ODOM_MODEL_OMNI_CORRECTED)
  {
    const double delta_rot = delta[2];
    M.sd[0] = a3 * (delta_trans * delta_trans) + a1 * (delta_rot * delta_rot);  // :101-106 / :181-186
    M.sd[1] = a4 * (delta_rot * delta_rot) + a2 * (delta_trans * delta_trans);
    M.sd[2] = a1 * (delta_rot * delta_rot) + a5 * (delta_trans * delta_trans);
    if (M.model == BPF_ODOM_MODEL_OMNI_CORRECTED)
      for (double& v : M.sd)
        v = std::sqrt(v);
  }
  else if (M.model == BPF_ODOM_MODEL_DIFF || M.model == BPF_ODOM_MODEL_DIFF_CORRECTED)
  {
    M.rot1 = (delta_trans < 0.01) ? 0.0 : M.bearing0;  // :135-138 / :213-216
    M.rot2 = odom_angle_diff(delta[2], M.rot1);
    const double r1a = std::fabs(odom_angle_diff(M.rot1, 0.0)), r1b = std::fabs(odom_angle_diff(M.rot1, M_PI));
    const double r2a = std::fabs(odom_angle_diff(M.rot2, 0.0)), r2b = std::fabs(odom_angle_diff(M.rot2, M_PI));
    const double n1 = std::min(r1a, r1b), n2 = std::min(r2a, r2b);
    M.sd[0] = a1 * n1 * n1 + a2 * delta_trans * delta_trans;  // :156-162 / :233-243
    M.sd[1] = a3 * delta_trans * delta_trans + a4 * n1 * n1 + a4 * n2 * n2;
    M.sd[2] = a1 * n2 * n2 + a2 * delta_trans * delta_trans;
    if (M.model == BPF_ODOM_MODEL_DIFF_CORRECTED)
      for (double& v : M.sd)
        v = std::sqrt(v);
  }
  else
  {
    const double at2 = absolute_motion[0] * absolute_motion[0];  // :264-274
    const double as2 = absolute_motion[1] * absolute_motion[1];
    const double ar2 = absolute_motion[2] * absolute_motion[2];
    const double rot_sd = std::sqrt(a1 * ar2 + a2 * at2);
    const double trans_sd = std::sqrt(a3 * at2 + a4 * ar2);
    const double strafe_sd = std::sqrt(a4 * ar2 + a5 * as2);
    M.sd[0] = trans_sd;  // draw order :289-291
    M.sd[1] = strafe_sd;
    M.sd[2] = rot_sd;
  }
  return M;
}

// `need` Gaussians PDFGaussian::draw(sd[rank % 3]) from the filter's drand48 stream, ranks
// [first, first + count) materialised in d_gauss; *consumed_out = uniforms the whole update took.
// `after_gauss` is launched right behind the generation (optimistically: a rare second pass re-runs it).
int generate_gaussians(bpf_engine* e, long long need, long long first, long long count, const double sd[3],
                       long long* consumed_out, const std::function<void()>& after_gauss)
{
  // attempts are accepted with probability pi/4; 6 sigma of slack, doubled on the (never yet seen) shortfall
  long long attempts = (long long)std::ceil((double)need / 0.7853981633974483 + 6.0 * std::sqrt((double)need)) + 64;
  HIPCHK(e, e->d_motion_result.reserve(4));
  HIPCHK(e, e->h_motion_result.reserve(4));
  HIPCHK(e, e->d_gauss.reserve((size_t)std::max<long long>(count, 1)));
  long long zero_at = kNoZero;
  for (int round = 0; round < 8; ++round)
  {
    const int tiles = (int)((attempts + kMotionTile - 1) / kMotionTile);
    HIPCHK(e, e->d_motion_counts.reserve((size_t)tiles));
    HIPCHK(e, e->d_motion_offsets.reserve((size_t)tiles + 1));
    MotionRngArgs A{};
    A.rng_state = e->rng;
    A.zero_at = zero_at;
    A.n_attempts = attempts;
    A.need_total = need;
    A.gauss_first = first;
    A.gauss_count = count;
    A.tile_counts = e->d_motion_counts.p;
    A.tile_offsets = e->d_motion_offsets.p;
    A.gauss = e->d_gauss.p;
    A.result = e->d_motion_result.p;
    for (int k = 0; k < 3; ++k)
      A.sd[k] = sd[k];
    A.jump = e->jump;
    {
      ProfScope ps(e, BPF_K_MOTION);
      HIPCHK(e, hipMemsetAsync(e->d_motion_result.p, 0, 4 * sizeof(long long), e->stream));
      hipLaunchKernelGGL(k_motion_count, dim3(tiles), dim3(256), 0, e->stream, A);
      hipLaunchKernelGGL(k_motion_offsets, dim3(1), dim3(1024), 0, e->stream, A, tiles);
      hipLaunchKernelGGL(k_motion_gauss, dim3(tiles), dim3(256), 0, e->stream, A);
      after_gauss();
      HIPCHK(e, hipGetLastError());
    }
    HIPCHK(e, hipMemcpyAsync(e->h_motion_result.p, e->d_motion_result.p, 4 * sizeof(long long), hipMemcpyDeviceToHost,
                             e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const long long consumed = e->h_motion_result.p[0], zero_seen = e->h_motion_result.p[1],
                    accepted = e->h_motion_result.p[2];
    if (zero_at == kNoZero && zero_seen > 0)
    {
      zero_at = zero_seen;  // the stream's one exact 0.0 lies in the window: re-run with it skipped
      continue;
    }
    if (accepted < need || consumed <= 0)
    {
      attempts *= 2;
      continue;
    }
    *consumed_out = consumed;
    return BPF_OK;
  }
  return e->fail(BPF_ERR_HIP, "Gaussian stream did not fill (internal error)");
}

int update_action(bpf_engine* e, const double pose[3], const double delta[3], const double absolute_motion[3],
                  long long global_first, long long global_count)
{
  const int n = e->sample_count;
  if (global_first < 0 || global_first + n > global_count)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "shard range outside the global set");
  HIPCHK(e, hipSetDevice(e->device));
  {
    const int rct = ensure_set_tree(e);  // the tree belongs to the poses the set was created with
    if (rct != BPF_OK)
      return rct;
  }
  const MotionModelDev M = motion_constants(e, pose, delta, absolute_motion);
  SampleSet& src = e->sets[e->cur];
  SampleSet& dst = e->sets[e->cur ^ 1];
  long long consumed = 0;
  // optimistic: poses go to the other set, which becomes current only once the pass is known good
  int rc = generate_gaussians(e, 3 * global_count, 3 * global_first, 3ll * n, M.sd, &consumed, [&]() {
    hipLaunchKernelGGL(k_motion_apply, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, src.dev(), dst.dev(), n, M,
                       (const double*)e->d_gauss.p);
  });
  if (rc != BPF_OK)
    return rc;
  e->rng = lcg_skip_host(e->rng, (uint64_t)consumed, e->jump);
  e->cur ^= 1;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  e->fused_partials = 0;
  e->set_epoch++;
  e->hist_matches_set = false;
  return BPF_OK;
}

int build_set_tree(bpf_engine* e, int n);

// A set adopted from host memory (bpf_pf_set_samples) gets its histogram tree only when somebody asks for what the
// tree is for -- the leaf count (get_state, snapshot, the systematic resampler) -- or before the poses move
// (the reference builds the tree when the set is created, so it must be the tree of THESE poses).  A cycle that
// uploads the set, updates and resamples with the multinomial resampler never needs it: the resample builds the new
// set's tree from its draws.
int ensure_set_tree(bpf_engine* e)
{
  if (!e->tree_pending)
    return BPF_OK;
  return build_set_tree(e, e->sample_count);
}

// what initWithGaussian / initWithPoseFn leave besides the poses (particle_filter.cpp:126-131,157-162): the
// histogram tree of the set (leaf / bin counts), w_slow = w_fast = 0, converged = false
int finish_init(bpf_engine* e, int n)
{
  SampleSet& s = e->sets[e->cur];
  e->sample_count = n;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  e->fused_partials = 0;
  e->set_epoch++;
  e->hist_matches_set = false;
  HIPCHK(e, hipMemsetAsync(e->d_scalars.p, 0, sizeof(FilterScalars), e->stream));
  e->converged = 0;
  e->converged_pending = false;
  return build_set_tree(e, n);
}

// the histogram tree of the current set (PFKDTree of every sample, pf_kdtree.cpp:49-56): leaf and bin counts
int build_set_tree(bpf_engine* e, int n)
{
  SampleSet& s = e->sets[e->cur];
  e->tree_pending = false;
  HIPCHK(e, e->d_keys.reserve((size_t)n * 3));
  hipLaunchKernelGGL(k_set_keys, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, s.dev(), n, e->d_keys.p);
  HIPCHK(e, hipGetLastError());
  bool handled = false;
  int stop = -1, leaf = 0, bins = 0;
  if (n >= 8192)
  {
    int rc = kld_tree_on_device(e, n, &handled, &stop, &leaf, &bins, true);
    if (rc != BPF_OK)
      return rc;
  }
  if (!handled)
  {
    std::vector<int> keys((size_t)n * 3);
    H2D_OR_RETURN(d2h_to_host(e, keys.data(), e->d_keys.p, keys.size() * sizeof(int), e->stream));
    e->hist.clear();
    for (int i = 0; i < n; ++i)
      e->hist.insert(keys[3 * (size_t)i], keys[3 * (size_t)i + 1], keys[3 * (size_t)i + 2]);
    leaf = e->hist.leaf_count();
    bins = e->hist.bin_count();
    e->hist_matches_set = true;
  }
  e->leaf_count = leaf;
  e->bin_count = bins;
  return BPF_OK;
}
}  // namespace

int bpf_odom_set_model(bpf_engine* e, int model_type, double alpha1, double alpha2, double alpha3, double alpha4,
                       double alpha5)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (model_type < BPF_ODOM_MODEL_DIFF || model_type > BPF_ODOM_MODEL_GAUSSIAN)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "unknown odom model type");
  e->odom_model = model_type;
  e->odom_alpha[0] = alpha1;
  e->odom_alpha[1] = alpha2;
  e->odom_alpha[2] = alpha3;
  e->odom_alpha[3] = alpha4;
  e->odom_alpha[4] = alpha5;
  e->odom_configured = true;
  return BPF_OK;
}

int bpf_pf_init_with_gaussian(bpf_engine* e, const double mean[3], const double rotation[9], const double sigma[3])
{
  if (!e || !mean || !rotation || !sigma)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  const int n = e->max_samples;
  HIPCHK(e, e->d_init_rot.reserve(9));
  HIPCHK(e, hipMemcpyAsync(e->d_init_rot.p, rotation, 9 * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));  // `rotation` is the caller's memory
  SampleSet& dst = e->sets[e->cur];
  long long consumed = 0;
  int rc = generate_gaussians(e, 3ll * n, 0, 3ll * n, sigma, &consumed, [&]() {
    hipLaunchKernelGGL(k_init_gaussian, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, dst.dev(), n,
                       (const double*)e->d_gauss.p, mean[0], mean[1], mean[2], (const double*)e->d_init_rot.p,
                       1.0 / (double)n);
  });
  if (rc != BPF_OK)
    return rc;
  e->rng = lcg_skip_host(e->rng, (uint64_t)consumed, e->jump);
  e->spread_init = false;
  return finish_init(e, n);
}

int bpf_pf_init_with_random_poses(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  FreeSpaceDev fs{};
  int rc = ensure_free_space(e, &fs);
  if (rc != BPF_OK)
    return rc;
  const int n = e->max_samples;
  hipLaunchKernelGGL(k_init_free_space, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->sets[e->cur].dev(), n,
                     e->rng, e->jump, fs, 1.0 / (double)n);
  HIPCHK(e, hipGetLastError());
  e->rng = lcg_skip_host(e->rng, 2ull * (uint64_t)n, e->jump);
  e->spread_init = true;  // uniform over the free space: scored in tile order until a resample says otherwise
  return finish_init(e, n);
}

int bpf_pf_update_action(bpf_engine* e, const double pose[3], const double delta[3], const double absolute_motion[3])
{
  if (!e || !pose || !delta || !absolute_motion)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf || !e->odom_configured)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create and bpf_odom_set_model first");
  return update_action(e, pose, delta, absolute_motion, 0, e->sample_count);
}

int bpf_shard_update_action(bpf_engine* e, const double pose[3], const double delta[3],
                            const double absolute_motion[3], long long global_first, long long global_count)
{
  if (!e || !pose || !delta || !absolute_motion)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf || !e->odom_configured)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create and bpf_odom_set_model first");
  return update_action(e, pose, delta, absolute_motion, global_first, global_count);
}
