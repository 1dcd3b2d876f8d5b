// Host-side pose histogram with the reference fork's kd-tree leaf-count semantic.
//
// PFKDTree (src/amcl/pf/pf_kdtree.cpp:97-150) is an unbalanced binary tree whose every
// node holds a bin key; a node counts as a "leaf" until the first different key is routed
// through it.  getLeafCount() is therefore an insertion-order-dependent quantity (about a
// third of the number of distinct bins), and ParticleFilter::resampleMultinomial tests it
// after every draw (particle_filter.cpp:416).  That stop rule is inherently sequential, so
// it stays on the host: the GPU produces the ordered key stream, this class replays it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace bpf
{

class KdHistogram
{
public:
  void clear()
  {
    kx_.clear(); ky_.clear(); kt_.clear();
    pivot_.clear(); lo_.clear(); hi_.clear(); label_.clear();
    leaves_ = 0;
  }

  int leaf_count() const { return leaves_; }
  int bin_count() const { return (int)kx_.size(); }

  // returns true when the key opened a new bin
  bool insert(int x, int y, int t)
  {
    if (kx_.empty())
    {
      add_node(x, y, t);
      return true;
    }
    int cur = 0;
    for (;;)
    {
      if (kx_[cur] == x && ky_[cur] == y && kt_[cur] == t)
        return false;
      int8_t pv = pivot_[cur];
      if (pv < 0)
      {
        // first different key through this node fixes its split axis: largest |delta|,
        // earliest axis on ties (pf_kdtree.cpp:133-146)
        const int d0 = std::abs(x - kx_[cur]), d1 = std::abs(y - ky_[cur]), d2 = std::abs(t - kt_[cur]);
        int best = 0;
        pv = -1;
        if (d0 > best) { best = d0; pv = 0; }
        if (d1 > best) { best = d1; pv = 1; }
        if (d2 > best) { best = d2; pv = 2; }
        pivot_[cur] = pv;
        leaves_ -= 1;
      }
      const bool up = (pv == 0) ? (x > kx_[cur]) : (pv == 1) ? (y > ky_[cur]) : (t > kt_[cur]);
      int& next = up ? hi_[cur] : lo_[cur];
      if (next < 0)
      {
        const int fresh = (int)kx_.size();
        next = fresh;  // written before the vectors may reallocate
        add_node(x, y, t);
        return true;
      }
      cur = next;
    }
  }

  int find(int x, int y, int t) const
  {
    int cur = kx_.empty() ? -1 : 0;
    while (cur >= 0)
    {
      if (kx_[cur] == x && ky_[cur] == y && kt_[cur] == t)
        return cur;
      const int8_t pv = pivot_[cur];
      if (pv < 0)
        return -1;
      const bool up = (pv == 0) ? (x > kx_[cur]) : (pv == 1) ? (y > ky_[cur]) : (t > kt_[cur]);
      cur = up ? hi_[cur] : lo_[cur];
    }
    return -1;
  }

  // PFKDTree::cluster (pf_kdtree.cpp:58-76,169-194): 26-connected components of occupied
  // bins; labels follow the creation order of each component's first bin.
  int label_components()
  {
    const int n = bin_count();
    label_.assign(n, -1);
    std::vector<int> stack;
    int next_label = 0;
    for (int seed = 0; seed < n; ++seed)
    {
      if (label_[seed] >= 0)
        continue;
      label_[seed] = next_label;
      stack.push_back(seed);
      while (!stack.empty())
      {
        const int cur = stack.back();
        stack.pop_back();
        for (int dx = -1; dx <= 1; ++dx)
          for (int dy = -1; dy <= 1; ++dy)
            for (int dt = -1; dt <= 1; ++dt)
            {
              if (!dx && !dy && !dt)
                continue;
              const int nb = find(kx_[cur] + dx, ky_[cur] + dy, kt_[cur] + dt);
              if (nb >= 0 && label_[nb] < 0)
              {
                label_[nb] = next_label;
                stack.push_back(nb);
              }
            }
      }
      ++next_label;
    }
    return next_label;
  }

  int label_of(int node) const { return label_[node]; }

private:
  void add_node(int x, int y, int t)
  {
    kx_.push_back(x); ky_.push_back(y); kt_.push_back(t);
    pivot_.push_back(-1); lo_.push_back(-1); hi_.push_back(-1);
    leaves_ += 1;
  }

  std::vector<int> kx_, ky_, kt_;
  std::vector<int8_t> pivot_;
  std::vector<int> lo_, hi_;
  std::vector<int> label_;
  int leaves_ = 0;
};

// Generation-stamped open-addressing set of bin keys: lets the replay skip the tree walk for a
// key it has already inserted (a repeated key only adds to the bin's weight, pf_kdtree.cpp:107-110,
// and changes neither the tree shape nor the leaf count).
class SeenKeys
{
public:
  void reset(size_t expected)
  {
    size_t want = 1024;
    while (want < expected * 4)
      want <<= 1;
    if (want != keys_.size())
    {
      keys_.assign(want, 0);
      gen_.assign(want, 0);
      cur_ = 0;
    }
    if (++cur_ == 0)
    {
      std::fill(gen_.begin(), gen_.end(), 0u);
      cur_ = 1;
    }
    used_ = 0;
  }

  // true when the key was not in the set (and is now); keys outside +-2^20 bins are never cached
  bool first_time(int x, int y, int t)
  {
    const int lim = 1 << 20;
    if (x < -lim || x >= lim || y < -lim || y >= lim || t < -lim || t >= lim || used_ * 2 > keys_.size())
      return true;
    const uint64_t k = ((uint64_t)(x + lim) << 42) | ((uint64_t)(y + lim) << 21) | (uint64_t)(t + lim);
    uint64_t h = k * 0x9E3779B97F4A7C15ull;
    size_t i = (size_t)(h >> 20) & (keys_.size() - 1);
    for (;;)
    {
      if (gen_[i] != cur_)
      {
        gen_[i] = cur_;
        keys_[i] = k;
        ++used_;
        return true;
      }
      if (keys_[i] == k)
        return false;
      i = (i + 1) & (keys_.size() - 1);
    }
  }

private:
  std::vector<uint64_t> keys_;
  std::vector<uint32_t> gen_;
  uint32_t cur_ = 0;
  size_t used_ = 0;
};

// ParticleFilter::resampleLimit (particle_filter.cpp:475-502)
inline int resample_limit(int k, int min_samples, int max_samples, double pop_err, double pop_z)
{
  if (k <= 1)
    return max_samples;
  const double kd = (double)k;
  const double b = 2 / (9 * (kd - 1));
  const double c = std::sqrt(2 / (9 * (kd - 1))) * pop_z;
  const double x = 1 - b + c;
  const int n = (int)std::ceil((k - 1) / (2 * pop_err) * x * x * x);
  if (n < min_samples)
    return min_samples;
  if (n > max_samples)
    return max_samples;
  return n;
}

// host-side bin key, same expression as the device pose_key()
inline void host_pose_key(double x, double y, double th, int key[3])
{
  const double cell_th = 10 * M_PI / 180;
  key[0] = (int)std::floor(x / 0.50);
  key[1] = (int)std::floor(y / 0.50);
  key[2] = (int)std::floor(th / cell_th);
}

}  // namespace bpf
