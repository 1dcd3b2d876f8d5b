// Cluster statistics of the resident set on the device:
// ParticleFilter::computeClusterStatsForSet (particle_filter.cpp:505-636) with PFKDTree::cluster /
// clusterNode / getCluster (pf_kdtree.cpp:58-90,152-194) and what Node2D::getMaxWeightPose reads
// (node_2d.cpp:588-617) -- without copying the set to the host.
//
//   bins        the occupied histogram bins = distinct keys of the set (hash table on the packed key, k_kld_hash);
//               a bin's identity is the index of the first sample that falls into it, which is also the order in
//               which the reference's tree creates its nodes.
//   clusters    26-connected components of the occupied bins (pf_kdtree.cpp:169-194) by lock-free union-find over the
//               bins: every bin looks its 26 neighbours up in the hash table and hooks the larger root under the
//               smaller, so a component's root is its earliest bin.  The reference numbers the components in the
//               creation order of their first node (:58-76): label = rank of the root among the roots = an
//               exclusive prefix sum over the samples of "this sample is the first of a root bin".
//   sums        per cluster: count, sum w, sum w x, w y, w cos, w sin, w x x, w x y, w y x, w y y.  The reference adds
//               them up sample by sample in index order; here they are accumulated as 128-bit FIXED-POINT numbers
//               (32.96, two 64-bit integer atomics per term with the carry counted by the thread whose add wrapped):
//               integer addition is associative, so the result does not depend on the order the atomics land in -- the
//               same bits every run -- and it is the exact sum of the terms rounded to 2^-96, closer to the true sum than
//               the reference's serial double chain.  Difference from the reference: summation rounding only
//               (a few ulp of the sums; 1e-12 relative is the tests' budget).  bpf_set_option(BPF_OPT_STATS_HOST, 1)
//               selects the bit-exact host evaluation instead.
//   finish      means, covariances, circular variance per cluster (device libm for atan2 / log / sqrt), the set's own
//               statistics as the sum over the clusters, and the heaviest cluster.
#pragma once
#include <climits>

#include "kernels_kld.hpp"
#include "kernels_pf.hpp"

namespace bpf
{

constexpr int kStatTerms = 10;  // w, wx, wy, wcos, wsin, wxx, wxy, wyx, wyy, count

struct StatsArgs
{
  ParticlesDev p;
  int n;
  const int* keys;                   // [3 n] bin keys (k_set_keys)
  const unsigned long long* h_key;   // hash table of the packed keys (k_kld_hash)
  const int* h_tmin;                 // first sample of the bin in a slot
  unsigned h_mask;
  const int* slot;                   // [n] table slot of each sample's bin
  int* parent;                       // [table] union-find over bins, values are bin ids (first-sample indices)
  int* label;                        // [n] cluster index of a root bin, at the root's first-sample index
  int* flags;                        // [0] key out of range, [1] non-finite term, [2] cluster count
  long long* acc_hi;                 // [kStatTerms][n]
  unsigned long long* acc_lo;        // [kStatTerms][n]
};

__device__ __forceinline__ int stats_find(const StatsArgs& A, int id)
{
  for (;;)
  {
    const int p = __hip_atomic_load(&A.parent[A.slot[id]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == id)
      return id;
    id = p;
  }
}

__global__ void k_stats_init(const StatsArgs A)
{
  const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s <= A.h_mask)
    A.parent[s] = A.h_tmin[s];  // occupied slot: the bin is its own root
}

// one thread per sample; the first sample of a bin visits the bin's 26 neighbours
__global__ void k_stats_union(const StatsArgs A)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n || A.h_tmin[A.slot[i]] != i)
    return;
  const int* k = &A.keys[3 * (size_t)i];
  for (int dx = -1; dx <= 1; ++dx)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dt = -1; dt <= 1; ++dt)
      {
        if (!dx && !dy && !dt)
          continue;
        const int nk[3] = { k[0] + dx, k[1] + dy, k[2] + dt };
        unsigned long long pk;
        if (!kld_pack(nk, &pk))
          continue;  // a key outside the packing range cannot be in the table
        unsigned h = (unsigned)((pk * 0x9E3779B97F4A7C15ull) >> 32) & A.h_mask;
        int other = -1;
        for (;;)
        {
          const unsigned long long held = A.h_key[h];
          if (held == kKldEmpty)
            break;
          if (held == pk)
          {
            other = A.h_tmin[h];
            break;
          }
          h = (h + 1) & A.h_mask;
        }
        if (other < 0)
          continue;
        // unite: the larger root goes under the smaller (atomicMin keeps whichever hook is smaller; a root that was
        // hooked meanwhile is followed and the union continues from there, so no link is lost)
        int a = stats_find(A, i), b = stats_find(A, other);
        while (a != b)
        {
          if (a < b)
          {
            const int t = a;
            a = b;
            b = t;
          }
          const int old = atomicMin(&A.parent[A.slot[a]], b);
          if (old == a)
            break;
          a = stats_find(A, old);
          b = stats_find(A, b);
        }
      }
}

// per sample: root of its bin; flag = the sample is the first of a root bin.  Tile sums of the flags for the scan.
constexpr int kStatTile = 2048;

__global__ __launch_bounds__(256) void k_stats_roots(const StatsArgs A, int* __restrict__ root_of, int* __restrict__ tile_sums)
{
  __shared__ int s_w[4];
  const int base = blockIdx.x * kStatTile;
  int cnt = 0;
  for (int j = threadIdx.x; j < kStatTile; j += 256)
  {
    const int i = base + j;
    if (i < A.n)
    {
      const int first = A.h_tmin[A.slot[i]];
      const int root = stats_find(A, first);
      root_of[i] = root;
      cnt += (root == i) ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0)
    s_w[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0)
    tile_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// one block: exclusive scan of the tile sums in place; the total is the cluster count
__global__ __launch_bounds__(1024) void k_stats_scan_offsets(int* tile_sums, int tiles, int* flags)
{
  __shared__ int s_part[1024];
  const int tid = threadIdx.x;
  const int per = (tiles + 1023) / 1024;
  const int lo = min(tid * per, tiles), hi = min(lo + per, tiles);
  int sum = 0;
  for (int i = lo; i < hi; ++i)
    sum += tile_sums[i];
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1)
  {
    const int v = (tid >= o) ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - sum;
  for (int i = lo; i < hi; ++i)
  {
    const int t = tile_sums[i];
    tile_sums[i] = run;
    run += t;
  }
  if (tid == 1023)
    flags[2] = s_part[1023];
}

// label[root sample] = rank of the root among the roots (sample order)
__global__ __launch_bounds__(256) void k_stats_labels(const StatsArgs A, const int* __restrict__ root_of,
                                                      const int* __restrict__ tile_offsets)
{
  __shared__ int s_w[4];
  constexpr int per = kStatTile / 256;
  const int base = blockIdx.x * kStatTile + threadIdx.x * per;
  int f[per];
  int sum = 0;
#pragma unroll
  for (int j = 0; j < per; ++j)
  {
    f[j] = (base + j < A.n && root_of[base + j] == base + j) ? 1 : 0;
    sum += f[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = sum;
  for (int o = 1; o < 64; o <<= 1)
  {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o)
      incl += u;
  }
  if (lane == 63)
    s_w[wave] = incl;
  __syncthreads();
  int run = tile_offsets[blockIdx.x] + incl - sum;
  for (int q = 0; q < wave; ++q)
    run += s_w[q];
#pragma unroll
  for (int j = 0; j < per; ++j)
    if (f[j])
    {
      A.label[base + j] = run;
      run += 1;
    }
}

// 32.96 fixed point: the 128-bit integer round-down of t * 2^96, as (hi, lo).  |t| < 2^31 (the largest term is
// w x y with a weight <= 1 and coordinates of a few km); the 96 fractional bits keep every bit of a term down to
// 2^-43 (a particle whose weight is 1e-8 still contributes 69 significant bits -- with 64 fractional bits a cluster of
// such stragglers came out only 1e-11 accurate).
struct Fx
{
  long long hi;
  unsigned long long lo;
};

__device__ __forceinline__ Fx fx_from(double t, bool* bad)
{
  Fx r;
  if (!(fabs(t) < 2.0e9))
  {
    *bad = true;
    r.hi = 0;
    r.lo = 0;
    return r;
  }
  const double s = t * 4294967296.0;  // exact
  double f = floor(s);
  double frac = s - f;  // exact; in [0, 1], 1 only when s is a tiny negative number
  if (frac >= 1.0)
  {
    f += 1.0;
    frac = 0.0;
  }
  r.hi = (long long)f;
  r.lo = (unsigned long long)(frac * 18446744073709551616.0);
  return r;
}

__device__ __forceinline__ Fx fx_add(Fx a, Fx b)
{
  Fx r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1 : 0);
  return r;
}

__device__ __forceinline__ double fx_to_double(long long hi, unsigned long long lo)
{
  return (double)hi * 2.3283064365386963e-10 + (double)lo * 1.2621774483536189e-29;  // 2^-32, 2^-96
}

__device__ __forceinline__ void fx_atomic_add(long long* hi, unsigned long long* lo, Fx v)
{
  const unsigned long long old = atomicAdd(lo, v.lo);
  const unsigned long long carry = (old + v.lo < old) ? 1ull : 0ull;
  atomicAdd(reinterpret_cast<unsigned long long*>(hi), (unsigned long long)v.hi + carry);
}

// every sample adds its terms to its cluster's accumulators; a wave whose samples all belong to one cluster (the
// tracking regime) adds them up in registers first and sends one set of atomics
__global__ __launch_bounds__(256) void k_stats_accumulate(const StatsArgs A, const int* __restrict__ root_of)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < A.n;
  int cidx = -1;
  Fx t[kStatTerms];
  bool bad = false;
  if (live)
  {
    cidx = A.label[root_of[i]];
    const double x = A.p.x[i], y = A.p.y[i], th = A.p.th[i], w = A.p.w[i];
    double s, c;
    sincos(th, &s, &c);
    // particle_filter.cpp:577-600: w, w x, w y, w cos, w sin, then c[j][k] += w * p[j] * p[k] (left to right)
    t[0] = fx_from(w, &bad);
    t[1] = fx_from(w * x, &bad);
    t[2] = fx_from(w * y, &bad);
    t[3] = fx_from(w * c, &bad);
    t[4] = fx_from(w * s, &bad);
    t[5] = fx_from(w * x * x, &bad);
    t[6] = fx_from(w * x * y, &bad);
    t[7] = fx_from(w * y * x, &bad);
    t[8] = fx_from(w * y * y, &bad);
    t[9].hi = 1ll << 32;  // the count: 1.0
    t[9].lo = 0;
  }
  else
  {
#pragma unroll
    for (int k = 0; k < kStatTerms; ++k)
    {
      t[k].hi = 0;
      t[k].lo = 0;
    }
  }
  if (bad)
    atomicExch(&A.flags[1], 1);
  const int first = __builtin_amdgcn_readfirstlane(cidx);
  const bool uniform = __builtin_amdgcn_ballot_w64(live && cidx != first) == 0 && first >= 0;
  if (uniform)
  {
#pragma unroll
    for (int k = 0; k < kStatTerms; ++k)
    {
      Fx v = t[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
      {
        Fx u;
        u.hi = __shfl_xor(v.hi, o, 64);
        u.lo = __shfl_xor(v.lo, o, 64);
        v = fx_add(v, u);
      }
      if ((threadIdx.x & 63) == 0)
        fx_atomic_add(&A.acc_hi[(size_t)k * A.n + first], &A.acc_lo[(size_t)k * A.n + first], v);
    }
  }
  else if (live && cidx >= 0)
  {
#pragma unroll
    for (int k = 0; k < kStatTerms; ++k)
      fx_atomic_add(&A.acc_hi[(size_t)k * A.n + cidx], &A.acc_lo[(size_t)k * A.n + cidx], t[k]);
  }
}

struct ClusterDev  // = bpf_cluster (include/badger_pf.h)
{
  int count;
  double weight;
  double mean[3];
  double cov[5];
};

struct StatsResult
{
  int cluster_count;
  int best;            // heaviest cluster (first of equals), -1 if none has weight > 0
  double best_weight;
  double best_pose[3];
  double set_mean[3];
  double set_cov[5];
};

__device__ __forceinline__ void stats_moments(const double* m, ClusterDev* o)
{
  // particle_filter.cpp:541-567 / 607-635 (normalizeCluster, computeSetStats)
  const double weight = m[0];
  o->weight = weight;
  o->mean[0] = m[1] / weight;
  o->mean[1] = m[2] / weight;
  o->mean[2] = atan2(m[4], m[3]);
  o->cov[0] = m[5] / weight - o->mean[0] * o->mean[0];
  o->cov[1] = m[6] / weight - o->mean[0] * o->mean[1];
  o->cov[2] = m[7] / weight - o->mean[1] * o->mean[0];
  o->cov[3] = m[8] / weight - o->mean[1] * o->mean[1];
  o->cov[4] = -2 * log(sqrt(m[3] * m[3] + m[4] * m[4]));
}

__global__ void k_stats_clusters(const StatsArgs A, ClusterDev* __restrict__ out)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= A.flags[2])
    return;
  double m[kStatTerms];
#pragma unroll
  for (int k = 0; k < kStatTerms; ++k)
    m[k] = fx_to_double(A.acc_hi[(size_t)k * A.n + c], A.acc_lo[(size_t)k * A.n + c]);
  ClusterDev o;
  o.count = (int)(A.acc_hi[(size_t)9 * A.n + c] >> 32);
  stats_moments(m, &o);
  out[c] = o;
}

// one block: the set's own sums = the clusters' sums added up (still integers: exact, any order), and the heaviest
// cluster, first of equals (node_2d.cpp:608-612 keeps the first strictly larger weight)
__global__ __launch_bounds__(1024) void k_stats_set(const StatsArgs A, const ClusterDev* __restrict__ clusters,
                                                    StatsResult* __restrict__ res)
{
  __shared__ long long s_hi[kStatTerms];
  __shared__ unsigned long long s_lo[kStatTerms];
  __shared__ double s_bw[16];
  __shared__ int s_bi[16];
  const int C = A.flags[2];
  const int tid = threadIdx.x;
  if (tid < kStatTerms)
  {
    s_hi[tid] = 0;
    s_lo[tid] = 0;
  }
  __syncthreads();
  Fx t[kStatTerms];
#pragma unroll
  for (int k = 0; k < kStatTerms; ++k)
  {
    t[k].hi = 0;
    t[k].lo = 0;
  }
  double bw = 0.0;
  int bi = INT_MAX;
  for (int c = tid; c < C; c += 1024)
  {
#pragma unroll
    for (int k = 0; k < kStatTerms; ++k)
    {
      Fx v;
      v.hi = A.acc_hi[(size_t)k * A.n + c];
      v.lo = A.acc_lo[(size_t)k * A.n + c];
      t[k] = fx_add(t[k], v);
    }
    const double w = clusters[c].weight;
    if (w > bw)  // ascending c within a thread: the first of equals stays
    {
      bw = w;
      bi = c;
    }
  }
#pragma unroll
  for (int k = 0; k < kStatTerms; ++k)
  {
    Fx v = t[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
      Fx u;
      u.hi = __shfl_xor(v.hi, o, 64);
      u.lo = __shfl_xor(v.lo, o, 64);
      v = fx_add(v, u);
    }
    if ((tid & 63) == 0)
      fx_atomic_add(&s_hi[k], &s_lo[k], v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
  {
    const double ow = __shfl_xor(bw, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ow > bw || (ow == bw && oi < bi))
    {
      bw = ow;
      bi = oi;
    }
  }
  if ((tid & 63) == 0)
  {
    s_bw[tid >> 6] = bw;
    s_bi[tid >> 6] = bi;
  }
  __syncthreads();
  if (tid == 0)
  {
    for (int q = 1; q < 16; ++q)
      if (s_bw[q] > bw || (s_bw[q] == bw && s_bi[q] < bi))
      {
        bw = s_bw[q];
        bi = s_bi[q];
      }
    double m[kStatTerms];
    for (int k = 0; k < kStatTerms; ++k)
      m[k] = fx_to_double(s_hi[k], s_lo[k]);
    ClusterDev o;
    stats_moments(m, &o);
    StatsResult r;
    r.cluster_count = C;
    r.best = (bw > 0.0 && bi != INT_MAX) ? bi : -1;
    r.best_weight = (r.best >= 0) ? bw : 0.0;
    for (int q = 0; q < 3; ++q)
    {
      r.best_pose[q] = (r.best >= 0) ? clusters[r.best].mean[q] : 0.0;
      r.set_mean[q] = o.mean[q];
    }
    for (int q = 0; q < 5; ++q)
      r.set_cov[q] = o.cov[q];
    *res = r;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same statistics for a set of at most 4096 samples (the tracking regime: ~2 000 samples in ~50 bins and one or
// two clusters) in ONE single-block launch, everything in LDS: the eleven launches above cost more than the host's
// serial loop at that size (0.17 ms against 0.10 ms); this one does not.  Same steps, same fixed-point sums, same
// results as the multi-launch path bit for bit.  Falls back (status != 0) when the set holds more than 1024 bins or 64
// clusters, or a key does not fit the packing.
constexpr int kStatBlockMax = 4096;
constexpr int kStatBlockBins = 1024;
constexpr int kStatBlockClusters = 64;
constexpr size_t kStatBlockLds = (size_t)kStatBlockMax * (8 + 8 + 4 + 4);  // keys, hash table, parent, label

struct StatsBlockArgs
{
  ParticlesDev p;
  int n;
  ClusterDev* clusters;        // [>= 64] device copy of the per-cluster results
  volatile int* result_host;   // pinned: StatsResult at byte 16, status at [1], then the generation at [0]
  int generation;
};

__device__ __forceinline__ int stats_block_find(const int* parent, int id)
{
  for (;;)
  {
    const int p = *reinterpret_cast<const volatile int*>(&parent[id]);
    if (p == id)
      return id;
    id = p;
  }
}

__global__ __launch_bounds__(1024) void k_stats_block(const StatsBlockArgs A)
{
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int W = kStatBlockMax;
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(smem);   // [W]
  int* s_hash = reinterpret_cast<int*>(smem + (size_t)W * 8);                 // [2 W] earliest sample of the key in a slot
  int* s_parent = reinterpret_cast<int*>(smem + (size_t)W * 16);              // [W] union-find, indexed by first-sample index
  int* s_label = reinterpret_cast<int*>(smem + (size_t)W * 20);               // [W] cluster index of a root, at the root's index
  __shared__ long long s_hi[kStatBlockClusters][kStatTerms];
  __shared__ unsigned long long s_lo[kStatBlockClusters][kStatTerms];
  __shared__ int s_list[kStatBlockBins];
  __shared__ int s_wsum[16];
  __shared__ int s_bins, s_bad, s_clusters;
  __shared__ StatsResult s_res;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int Q = 4;
  constexpr int kMask = 2 * W - 1;
  for (int s = tid; s < 2 * W; s += 1024)
    s_hash[s] = INT_MAX;
  for (int s = tid; s < kStatBlockClusters * kStatTerms; s += 1024)
  {
    (&s_hi[0][0])[s] = 0;
    (&s_lo[0][0])[s] = 0;
  }
  if (tid == 0)
  {
    s_bins = 0;
    s_bad = 0;
    s_clusters = 0;
  }
  // thread t owns samples 4t .. 4t + 3
  const int m0 = tid * Q;
  double x[Q], y[Q], th[Q], w[Q];
  unsigned long long pk[Q];
  int key[Q][3];
  bool bad_key = false;
#pragma unroll
  for (int q = 0; q < Q; ++q)
  {
    const int m = m0 + q;
    pk[q] = kKldEmpty;
    if (m < A.n)
    {
      x[q] = A.p.x[m];
      y[q] = A.p.y[m];
      th[q] = A.p.th[m];
      w[q] = A.p.w[m];
      pose_key(x[q], y[q], th[q], key[q]);
      if (!kld_pack(key[q], &pk[q]))
        bad_key = true;
      s_key[m] = pk[q];
    }
  }
  __syncthreads();
  if (bad_key)
    s_bad = 1;
  // ---- bins: a slot of the table holds the earliest sample with its key
  int slot[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q)
  {
    slot[q] = 0;
    const int m = m0 + q;
    if (m >= A.n || pk[q] == kKldEmpty)
      continue;
    unsigned h = (unsigned)((pk[q] * 0x9E3779B97F4A7C15ull) >> 40) & kMask;
    for (;;)
    {
      int held = *reinterpret_cast<volatile int*>(&s_hash[h]);
      if (held == INT_MAX)
      {
        held = atomicCAS(&s_hash[h], INT_MAX, m);
        if (held == INT_MAX)
          break;
      }
      if (s_key[held] == pk[q])
      {
        atomicMin(&s_hash[h], m);
        break;
      }
      h = (h + 1) & kMask;
    }
    slot[q] = (int)h;
  }
  __syncthreads();
  const bool usable = s_bad == 0;
  bool is_first[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q)
  {
    const int m = m0 + q;
    is_first[q] = usable && m < A.n && s_hash[slot[q]] == m;
    if (is_first[q])
    {
      s_parent[m] = m;  // the bin is its own root
      const int pos = atomicAdd(&s_bins, 1);
      if (pos < kStatBlockBins)
        s_list[pos] = m;
    }
  }
  __syncthreads();
  const int n_bins = s_bins;
  const bool fits = usable && n_bins <= kStatBlockBins;
  // ---- clusters: thread b unites bin b with its occupied neighbours (26-neighbourhood, pf_kdtree.cpp:169-194)
  if (fits && tid < n_bins)
  {
    const int i = s_list[tid];
    const unsigned long long mine = s_key[i];
    const int k0 = (int)(mine >> 40) - (1 << 23), k1 = (int)((mine >> 16) & 0xFFFFFFull) - (1 << 23),
              k2 = (int)(mine & 0xFFFFull) - (1 << 15);
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dt = -1; dt <= 1; ++dt)
        {
          if (!dx && !dy && !dt)
            continue;
          const int nk[3] = { k0 + dx, k1 + dy, k2 + dt };
          unsigned long long npk;
          if (!kld_pack(nk, &npk))
            continue;
          unsigned h = (unsigned)((npk * 0x9E3779B97F4A7C15ull) >> 40) & kMask;
          int other = -1;
          for (;;)
          {
            const int held = s_hash[h];
            if (held == INT_MAX)
              break;
            if (s_key[held] == npk)
            {
              other = held;
              break;
            }
            h = (h + 1) & kMask;
          }
          if (other < 0)
            continue;
          int a = stats_block_find(s_parent, i), b = stats_block_find(s_parent, other);
          while (a != b)
          {
            if (a < b)
            {
              const int t = a;
              a = b;
              b = t;
            }
            const int old = atomicMin(&s_parent[a], b);
            if (old == a)
              break;
            a = stats_block_find(s_parent, old);
            b = stats_block_find(s_parent, b);
          }
        }
  }
  __syncthreads();
  // ---- labels: rank of a root bin among the root bins, in sample order (prefix sum over the samples)
  int root[Q];
  int flags = 0, pre[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q)
  {
    const int m = m0 + q;
    root[q] = -1;
    if (fits && m < A.n)
      root[q] = stats_block_find(s_parent, s_hash[slot[q]]);
    pre[q] = flags;
    flags += (root[q] == m) ? 1 : 0;
  }
  int incl = flags;
  for (int o = 1; o < 64; o <<= 1)
  {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o)
      incl += u;
  }
  if (lane == 63)
    s_wsum[wave] = incl;
  __syncthreads();
  int base = incl - flags;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    base += (k < wave) ? s_wsum[k] : 0;
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (root[q] == m0 + q)
      s_label[m0 + q] = base + pre[q];
  if (tid == 1023)
    s_clusters = base + flags;
  __syncthreads();
  const int C = s_clusters;
  const bool ok = fits && C <= kStatBlockClusters;
  // ---- sums (32.96 fixed point, as k_stats_accumulate)
  if (ok)
  {
    bool bad = false;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      const int m = m0 + q;
      const bool live = m < A.n;
      const int cidx = live ? s_label[root[q]] : -1;
      Fx t[kStatTerms];
      if (live)
      {
        double sn, cs;
        sincos(th[q], &sn, &cs);
        t[0] = fx_from(w[q], &bad);
        t[1] = fx_from(w[q] * x[q], &bad);
        t[2] = fx_from(w[q] * y[q], &bad);
        t[3] = fx_from(w[q] * cs, &bad);
        t[4] = fx_from(w[q] * sn, &bad);
        t[5] = fx_from(w[q] * x[q] * x[q], &bad);
        t[6] = fx_from(w[q] * x[q] * y[q], &bad);
        t[7] = fx_from(w[q] * y[q] * x[q], &bad);
        t[8] = fx_from(w[q] * y[q] * y[q], &bad);
        t[9].hi = 1ll << 32;
        t[9].lo = 0;
      }
      else
      {
#pragma unroll
        for (int k = 0; k < kStatTerms; ++k)
        {
          t[k].hi = 0;
          t[k].lo = 0;
        }
      }
      const int first = __builtin_amdgcn_readfirstlane(cidx);
      const bool uniform = __builtin_amdgcn_ballot_w64(live && cidx != first) == 0 && first >= 0;
      if (uniform)
      {
#pragma unroll
        for (int k = 0; k < kStatTerms; ++k)
        {
          Fx v = t[k];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1)
          {
            Fx u;
            u.hi = __shfl_xor(v.hi, o, 64);
            u.lo = __shfl_xor(v.lo, o, 64);
            v = fx_add(v, u);
          }
          if (lane == 0)
            fx_atomic_add(&s_hi[first][k], &s_lo[first][k], v);
        }
      }
      else if (live && cidx >= 0)
      {
#pragma unroll
        for (int k = 0; k < kStatTerms; ++k)
          fx_atomic_add(&s_hi[cidx][k], &s_lo[cidx][k], t[k]);
      }
    }
    if (bad)
      s_bad = 2;
  }
  __syncthreads();
  // ---- moments per cluster, the set's sums, the heaviest cluster (one wave: at most 64 clusters)
  if (wave == 0)
  {
    const bool good = ok && s_bad == 0;
    double bw = 0.0;
    int bi = INT_MAX;
    double mean0 = 0.0, mean1 = 0.0, mean2 = 0.0;
    Fx tot[kStatTerms];
#pragma unroll
    for (int k = 0; k < kStatTerms; ++k)
    {
      tot[k].hi = 0;
      tot[k].lo = 0;
    }
    if (good && lane < C)
    {
      double mm[kStatTerms];
#pragma unroll
      for (int k = 0; k < kStatTerms; ++k)
      {
        tot[k].hi = s_hi[lane][k];
        tot[k].lo = s_lo[lane][k];
        mm[k] = fx_to_double(tot[k].hi, tot[k].lo);
      }
      ClusterDev o;
      o.count = (int)(s_hi[lane][9] >> 32);
      stats_moments(mm, &o);
      A.clusters[lane] = o;
      mean0 = o.mean[0];
      mean1 = o.mean[1];
      mean2 = o.mean[2];
      if (o.weight > 0.0)
      {
        bw = o.weight;
        bi = lane;
      }
    }
#pragma unroll
    for (int k = 0; k < kStatTerms; ++k)
    {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
      {
        Fx u;
        u.hi = __shfl_xor(tot[k].hi, o, 64);
        u.lo = __shfl_xor(tot[k].lo, o, 64);
        tot[k] = fx_add(tot[k], u);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
      const double ow = __shfl_xor(bw, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ow > bw || (ow == bw && oi < bi))  // the first of equals (node_2d.cpp:608-612 keeps the first strictly larger)
      {
        bw = ow;
        bi = oi;
      }
    }
    const int best = (good && bw > 0.0 && bi != INT_MAX) ? bi : -1;
    const int src_lane = best >= 0 ? best : 0;
    const double bp0 = __shfl(mean0, src_lane, 64), bp1 = __shfl(mean1, src_lane, 64), bp2 = __shfl(mean2, src_lane, 64);
    if (lane == 0)
    {
      StatsResult r;
      r.cluster_count = C;
      r.best = best;
      r.best_weight = (best >= 0) ? bw : 0.0;
      double mm[kStatTerms];
      for (int k = 0; k < kStatTerms; ++k)
        mm[k] = fx_to_double(tot[k].hi, tot[k].lo);
      ClusterDev so;
      stats_moments(mm, &so);
      r.best_pose[0] = best >= 0 ? bp0 : 0.0;
      r.best_pose[1] = best >= 0 ? bp1 : 0.0;
      r.best_pose[2] = best >= 0 ? bp2 : 0.0;
      for (int q = 0; q < 3; ++q)
        r.set_mean[q] = so.mean[q];
      for (int q = 0; q < 5; ++q)
        r.set_cov[q] = so.cov[q];
      s_res = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // the result block goes out as ONE wave-wide store (a chain of single-word stores to host memory costs ~0.3 us each)
    volatile int* out = A.result_host;
    constexpr int kWords = (int)(sizeof(StatsResult) / sizeof(int));
    static_assert(kWords <= 60, "one wave carries the result");
    if (lane < kWords)
      out[4 + lane] = reinterpret_cast<const int*>(&s_res)[lane];
    if (lane == 63)
      out[1] = good ? 0 : (s_bad ? 10 + s_bad : (fits ? 2 : 1));
    __threadfence_system();
    if (lane == 0)
      __hip_atomic_store(const_cast<int*>(out), A.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace bpf
