// C-ABI: cluster statistics.
// ---------------------------------------------------------------------- cluster statistics
namespace
{
// particle_filter.cpp:505-636 on the device (kernels_stats.hpp): bins, 26-connected components, fixed-point sums,
// moments, heaviest cluster; the host reads back one small result block (and the cluster array only if asked for one).
// *handled = false: a key outside the packing range or a non-finite term -- the caller evaluates on the host.
int compute_cluster_stats_device(bpf_engine* e, bool* handled)
{
  *handled = false;
  const int n = e->sample_count;
  if (n <= 0 || n >= (1 << 30))
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  if (n <= kStatBlockMax)
  {
    // the tracking regime: everything in one single-block launch (k_stats_block), the result in pinned memory
    HIPCHK(e, e->d_stats_clusters.reserve((size_t)std::max(n, kStatBlockClusters)));
    HIPCHK(e, e->h_stats_block.reserve(64));
    if (!e->stats_lds_attr_set)
    {
      HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_stats_block),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStatBlockLds));
      e->stats_lds_attr_set = true;
    }
    StatsBlockArgs B{};
    B.p = s.dev();
    B.n = n;
    B.clusters = reinterpret_cast<ClusterDev*>(e->d_stats_clusters.p);
    B.result_host = e->h_stats_block.p;
    e->stats_generation = (e->stats_generation % 0x3fffffff) + 1;
    B.generation = e->stats_generation;
    hipLaunchKernelGGL(k_stats_block, dim3(1), dim3(1024), kStatBlockLds, e->stream, B);
    HIPCHK(e, hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (unsigned spins = 0; !seen; ++spins)
    {
      seen = __atomic_load_n(e->h_stats_block.p, __ATOMIC_ACQUIRE) == B.generation;
      if (!seen && (spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50))
        break;
      if (!seen)
        __builtin_ia32_pause();
    }
    if (!seen)
    {
      HIPCHK(e, hipStreamSynchronize(e->stream));
      if (__atomic_load_n(e->h_stats_block.p, __ATOMIC_ACQUIRE) != B.generation)
        return e->fail(BPF_ERR_HIP, "k_stats_block did not publish its result");
    }
    if (e->h_stats_block.p[1] == 0)
    {
      StatsResult r;
      std::memcpy(&r, e->h_stats_block.p + 4, sizeof(r));
      e->stats_cluster_count = r.cluster_count;
      e->stats_best = r.best;
      e->stats_best_weight = r.best_weight;
      std::memcpy(e->stats_best_pose, r.best_pose, sizeof(r.best_pose));
      std::memcpy(e->set_mean, r.set_mean, sizeof(r.set_mean));
      std::memcpy(e->set_cov, r.set_cov, sizeof(r.set_cov));
      e->clusters.clear();
      e->stats_clusters_fetched = false;
      e->stats_on_device = true;
      e->stats_epoch = e->set_epoch;
      *handled = true;
      return BPF_OK;
    }
    if (e->h_stats_block.p[1] >= 10)
      return BPF_OK;  // key range / non-finite term: the host evaluation
    // more than 1024 bins or 64 clusters in a small set: the general device path below
  }
  unsigned table = 1024;
  while (table < 2u * (unsigned)n)
    table <<= 1;
  const int tiles = blocks_for(n, kStatTile);
  HIPCHK(e, e->d_keys.reserve((size_t)n * 3));
  HIPCHK(e, e->d_kld_hkey.reserve(table));
  HIPCHK(e, e->d_kld_htmin.reserve(table));
  HIPCHK(e, e->d_kld_slot.reserve((size_t)n));
  HIPCHK(e, e->d_stats_parent.reserve(table));
  HIPCHK(e, e->d_stats_label.reserve((size_t)n));
  HIPCHK(e, e->d_stats_root.reserve((size_t)n));
  HIPCHK(e, e->d_stats_tiles.reserve((size_t)tiles));
  HIPCHK(e, e->d_stats_flags.reserve(4));
  HIPCHK(e, e->d_stats_hi.reserve((size_t)kStatTerms * n));
  HIPCHK(e, e->d_stats_lo.reserve((size_t)kStatTerms * n));
  HIPCHK(e, e->d_stats_clusters.reserve((size_t)n));
  HIPCHK(e, e->d_stats_result.reserve(1));
  HIPCHK(e, e->h_stats_result.reserve(1));
  HIPCHK(e, e->h_stats_flags.reserve(4));
  e->kld_clean_table = 0;  // (the resampler's tables: in use here)
  HIPCHK(e, hipMemsetAsync(e->d_kld_hkey.p, 0xFF, (size_t)table * sizeof(unsigned long long), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_kld_htmin.p, 0x7F, (size_t)table * sizeof(int), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_stats_flags.p, 0, 4 * sizeof(int), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_stats_hi.p, 0, (size_t)kStatTerms * n * sizeof(long long), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_stats_lo.p, 0, (size_t)kStatTerms * n * sizeof(unsigned long long), e->stream));
  const dim3 grid(blocks_for(n, 256)), block(256);
  hipLaunchKernelGGL(k_set_keys, grid, block, 0, e->stream, s.dev(), n, e->d_keys.p);
  KldArgs K{};
  K.keys = e->d_keys.p;
  K.n = n;
  K.h_key = e->d_kld_hkey.p;
  K.h_tmin = e->d_kld_htmin.p;
  K.h_mask = table - 1;
  K.slot = e->d_kld_slot.p;
  K.flags = e->d_stats_flags.p;
  hipLaunchKernelGGL(k_kld_hash, grid, block, 0, e->stream, K);
  StatsArgs A{};
  A.p = s.dev();
  A.n = n;
  A.keys = e->d_keys.p;
  A.h_key = e->d_kld_hkey.p;
  A.h_tmin = e->d_kld_htmin.p;
  A.h_mask = table - 1;
  A.slot = e->d_kld_slot.p;
  A.parent = e->d_stats_parent.p;
  A.label = e->d_stats_label.p;
  A.flags = e->d_stats_flags.p;
  A.acc_hi = e->d_stats_hi.p;
  A.acc_lo = e->d_stats_lo.p;
  ClusterDev* clusters = reinterpret_cast<ClusterDev*>(e->d_stats_clusters.p);
  hipLaunchKernelGGL(k_stats_init, dim3(blocks_for((int)table, 256)), block, 0, e->stream, A);
  hipLaunchKernelGGL(k_stats_union, grid, block, 0, e->stream, A);
  hipLaunchKernelGGL(k_stats_roots, dim3(tiles), block, 0, e->stream, A, e->d_stats_root.p, e->d_stats_tiles.p);
  hipLaunchKernelGGL(k_stats_scan_offsets, dim3(1), dim3(1024), 0, e->stream, e->d_stats_tiles.p, tiles,
                     e->d_stats_flags.p);
  hipLaunchKernelGGL(k_stats_labels, dim3(tiles), block, 0, e->stream, A, (const int*)e->d_stats_root.p,
                     (const int*)e->d_stats_tiles.p);
  hipLaunchKernelGGL(k_stats_accumulate, grid, block, 0, e->stream, A, (const int*)e->d_stats_root.p);
  hipLaunchKernelGGL(k_stats_clusters, grid, block, 0, e->stream, A, clusters);
  hipLaunchKernelGGL(k_stats_set, dim3(1), dim3(1024), 0, e->stream, A, (const ClusterDev*)clusters,
                     e->d_stats_result.p);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(e->h_stats_result.p, e->d_stats_result.p, sizeof(StatsResult), hipMemcpyDeviceToHost,
                           e->stream));
  HIPCHK(e, hipMemcpyAsync(e->h_stats_flags.p, e->d_stats_flags.p, 4 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (e->h_stats_flags.p[0] != 0 || e->h_stats_flags.p[1] != 0)
    return BPF_OK;  // key range / non-finite weights: the host path takes the reference's route through them
  const StatsResult& r = *e->h_stats_result.p;
  e->stats_cluster_count = r.cluster_count;
  e->stats_best = r.best;
  e->stats_best_weight = r.best_weight;
  std::memcpy(e->stats_best_pose, r.best_pose, sizeof(r.best_pose));
  std::memcpy(e->set_mean, r.set_mean, sizeof(r.set_mean));
  std::memcpy(e->set_cov, r.set_cov, sizeof(r.set_cov));
  e->clusters.clear();
  e->stats_clusters_fetched = false;
  e->stats_on_device = true;
  e->stats_epoch = e->set_epoch;
  *handled = true;
  return BPF_OK;
}

// the cluster array itself, only when somebody asks for a cluster
int fetch_device_clusters(bpf_engine* e)
{
  if (!e->stats_on_device || e->stats_clusters_fetched)
    return BPF_OK;
  static_assert(sizeof(ClusterDev) == sizeof(bpf_cluster), "ClusterDev mirrors bpf_cluster");
  e->clusters.assign((size_t)e->stats_cluster_count, bpf_cluster{});
  if (e->stats_cluster_count > 0)
  {
    HIPCHK(e, hipMemcpyAsync(e->clusters.data(), e->d_stats_clusters.p,
                             (size_t)e->stats_cluster_count * sizeof(bpf_cluster), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
  }
  e->stats_clusters_fetched = true;
  return BPF_OK;
}

// particle_filter.cpp:505-636 on a host copy of the current set, in index order (BPF_OPT_STATS_HOST, and the fallback
// of the device evaluation)
int compute_cluster_stats(bpf_engine* e)
{
  if (e->stats_epoch == e->set_epoch)
    return BPF_OK;
  if (!e->stats_host)
  {
    bool handled = false;
    int rcd = compute_cluster_stats_device(e, &handled);
    if (rcd != BPF_OK)
      return rcd;
    if (handled)
      return BPF_OK;
  }
  e->stats_on_device = false;
  const int n = e->sample_count;
  std::vector<double> s((size_t)n * 4);
  int got = 0;
  int rc = bpf_pf_get_samples(e, s.data(), n, &got);
  if (rc != BPF_OK)
    return rc;
  if (!e->hist_matches_set)
  {
    // the histogram tree of this set is not at hand (set loaded with an explicit leaf count, or
    // restored): rebuild it the way initWith* / the resamplers do, by inserting every pose in order
    e->hist.clear();
    for (int i = 0; i < n; ++i)
    {
      int key[3];
      host_pose_key(s[4 * i], s[4 * i + 1], s[4 * i + 2], key);
      e->hist.insert(key[0], key[1], key[2]);
    }
    e->hist_matches_set = true;
  }
  e->hist.label_components();
  const int max_clusters = e->max_samples;  // cluster_max_count (particle_filter.cpp:84)
  struct Acc
  {
    int count = 0;
    double weight = 0, m[4] = { 0, 0, 0, 0 }, c[4] = { 0, 0, 0, 0 };
  };
  std::vector<Acc> acc;
  int cluster_count = 0;
  double weight = 0.0, m[4] = { 0, 0, 0, 0 }, c[4] = { 0, 0, 0, 0 };
  for (int i = 0; i < n; ++i)
  {
    const double* p = &s[4 * i];
    const double w = p[3];
    int key[3];
    host_pose_key(p[0], p[1], p[2], key);
    const int node = e->hist.find(key[0], key[1], key[2]);
    const int cidx = node < 0 ? -1 : e->hist.label_of(node);
    if (cidx < 0 || cidx >= max_clusters)
      continue;  // :574-576
    if (cidx + 1 > cluster_count)
      cluster_count = cidx + 1;
    if ((int)acc.size() < cluster_count)
      acc.resize(cluster_count);
    Acc& a = acc[cidx];
    a.count += 1;
    a.weight += w;
    a.m[0] += w * p[0];
    a.m[1] += w * p[1];
    a.m[2] += w * std::cos(p[2]);
    a.m[3] += w * std::sin(p[2]);
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 2; ++k)
        a.c[2 * j + k] += w * p[j] * p[k];
    weight += w;
    m[0] += w * p[0];
    m[1] += w * p[1];
    m[2] += w * std::cos(p[2]);
    m[3] += w * std::sin(p[2]);
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 2; ++k)
        c[2 * j + k] += w * p[j] * p[k];
  }
  e->clusters.assign((size_t)cluster_count, bpf_cluster{});
  for (int k = 0; k < cluster_count; ++k)
  {
    const Acc& a = acc[k];
    bpf_cluster& o = e->clusters[k];
    o.count = a.count;
    o.weight = a.weight;
    o.mean[0] = a.m[0] / a.weight;
    o.mean[1] = a.m[1] / a.weight;
    o.mean[2] = std::atan2(a.m[3], a.m[2]);
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 2; ++q)
        o.cov[2 * j + q] = a.c[2 * j + q] / a.weight - o.mean[j] * o.mean[q];
    o.cov[4] = -2 * std::log(std::sqrt(a.m[2] * a.m[2] + a.m[3] * a.m[3]));
  }
  e->set_mean[0] = m[0] / weight;
  e->set_mean[1] = m[1] / weight;
  e->set_mean[2] = std::atan2(m[3], m[2]);
  for (int j = 0; j < 2; ++j)
    for (int q = 0; q < 2; ++q)
      e->set_cov[2 * j + q] = c[2 * j + q] / weight - e->set_mean[j] * e->set_mean[q];
  e->set_cov[4] = -2 * std::log(std::sqrt(m[2] * m[2] + m[3] * m[3]));
  e->stats_epoch = e->set_epoch;
  return BPF_OK;
}
}  // namespace

int bpf_pf_compute_cluster_stats(bpf_engine* e, int* cluster_count_out, double set_mean[3], double set_cov[5])
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  int rc = compute_cluster_stats(e);
  if (rc != BPF_OK)
    return rc;
  if (cluster_count_out)
    *cluster_count_out = e->stats_on_device ? e->stats_cluster_count : (int)e->clusters.size();
  if (set_mean)
    std::memcpy(set_mean, e->set_mean, sizeof(e->set_mean));
  if (set_cov)
    std::memcpy(set_cov, e->set_cov, sizeof(e->set_cov));
  return BPF_OK;
}

int bpf_pf_get_cluster(bpf_engine* e, int cidx, bpf_cluster* out)
{
  if (!e || !out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  int rc = compute_cluster_stats(e);
  if (rc != BPF_OK)
    return rc;
  rc = fetch_device_clusters(e);
  if (rc != BPF_OK)
    return rc;
  if (cidx < 0 || cidx >= (int)e->clusters.size())
    return BPF_ERR_INVALID_ARGUMENT;  // getClusterStats returns false (particle_filter.cpp:642-643)
  *out = e->clusters[cidx];
  return BPF_OK;
}

int bpf_pf_get_max_weight_pose(bpf_engine* e, double* max_weight, double pose[3])
{
  if (!e || !max_weight || !pose)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  int rc = compute_cluster_stats(e);
  if (rc != BPF_OK)
    return rc;
  if (e->stats_on_device)
  {
    // found on the device; nothing of the set or of the cluster array crosses PCIe
    *max_weight = e->stats_best_weight;
    if (e->stats_best >= 0)
      std::memcpy(pose, e->stats_best_pose, 3 * sizeof(double));
    return BPF_OK;
  }
  double best = 0.0;
  int hyp = -1;
  for (size_t k = 0; k < e->clusters.size(); ++k)
    if (e->clusters[k].weight > best)  // node_2d.cpp:608-612
    {
      best = e->clusters[k].weight;
      hyp = (int)k;
    }
  *max_weight = best;
  if (hyp >= 0)
    std::memcpy(pose, e->clusters[hyp].mean, 3 * sizeof(double));
  return BPF_OK;
}
