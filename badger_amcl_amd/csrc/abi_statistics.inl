// C-ABI: cluster statistics.
// ---------------------------------------------------------------------- cluster statistics
namespace
{
// particle_filter.cpp:505-636 on a host copy of the current set, in index order
int compute_cluster_stats(bpf_engine* e)
{
  if (e->stats_epoch == e->set_epoch)
    return BPF_OK;
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
    *cluster_count_out = (int)e->clusters.size();
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
