// Particle-filter kernels: weight total / normalisation, CDF, drand48 draws + CDF search +
// pose gather + histogram keys, systematic targets, convergence test, layout conversion,
// distance-LUT construction.
#pragma once
#include "device_types.hpp"
#include "kernels_mailbox.hpp"
#include "kernels_score.hpp"

namespace bpf
{

constexpr int kCdfGuide = 1024;  // CDF guide table: entries 0 .. kCdfGuide (k_scan_final, cdf_find_guided)

// ------------------------------------------------------------------ layout conversion
// PFSample AoS {x,y,theta,w} (32 B) <-> SoA.  One thread per particle; the AoS side moves
// as two 16-byte accesses per lane.
__global__ void k_aos_to_soa(const double4* __restrict__ aos, ParticlesDev p, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const double4 v = aos[i];
  p.x[i] = v.x;
  p.y[i] = v.y;
  p.th[i] = v.z;
  p.w[i] = v.w;
}

__global__ void k_soa_to_aos(ParticlesDev p, double4* __restrict__ aos, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  aos[i] = make_double4(p.x[i], p.y[i], p.th[i], p.w[i]);
}

__global__ void k_fill(double* __restrict__ dst, double value, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    dst[i] = value;
}

// ------------------------------------------------------------------ deterministic sum
// Fixed-shape two-level tree: block b sums elements b*2048 .. +2047 (thread t takes 8
// consecutive ones), then one block folds the partials.  Same shape every run, so the
// total is reproducible bit for bit (the reference sums serially, planar_scanner.cpp:679;
// the difference is rounding only).
#define BPF_RED_BLOCK 256
#define BPF_RED_PER_THREAD 8
#define BPF_RED_TILE (BPF_RED_BLOCK * BPF_RED_PER_THREAD)

__device__ __forceinline__ double block_sum_256(double v, double* s_wave)
{
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    s_wave[wave] = v;
  __syncthreads();
  const double r = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(BPF_RED_BLOCK) void k_sum_partials(const double* __restrict__ w, int n,
                                                               double* __restrict__ partials)
{
  __shared__ double s_wave[4];
  const size_t base = (size_t)blockIdx.x * BPF_RED_TILE + (size_t)threadIdx.x * BPF_RED_PER_THREAD;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
    if (base + k < (size_t)n)
      acc += w[base + k];
  const double tot = block_sum_256(acc, s_wave);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = tot;
}

// scalars block layout (double[16]):
//  [0] total of the last sensor update (local shard)   [1] w_slow  [2] w_fast
//  [3] mean x   [4] mean y   [5] converged-count (as double)   [6] total used to normalise
struct FilterScalars
{
  double v[16];
};

// Folds the partials; when update_averages != 0 also applies particle_filter.cpp:237-256:
// w_avg = total / n, first-time assignment or exponential update of w_slow / w_fast.
// done_flag != nullptr (the host-buffer seam: sc is then the pinned host block): the word the host polls is
// published behind the total.
__global__ __launch_bounds__(BPF_RED_BLOCK) void k_sum_final(const double* __restrict__ partials, int n_partials,
                                                            FilterScalars* sc, int slot, int update_averages,
                                                            int n_samples, double alpha_slow, double alpha_fast,
                                                            unsigned long long* done_flag = nullptr,
                                                            unsigned long long done_value = 0ull)
{
  __shared__ double s_wave[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += BPF_RED_BLOCK)
    acc += partials[i];
  const double tot = block_sum_256(acc, s_wave);
  if (threadIdx.x == 0)
  {
    sc->v[slot] = tot;
    if (update_averages)
    {
      sc->v[6] = tot;
      if (tot > 0.0)
      {
        const double w_avg = tot / n_samples;
        double ws = sc->v[1], wf = sc->v[2];
        if (ws == 0.0)
          ws = w_avg;
        else
          ws += alpha_slow * (w_avg - ws);
        if (wf == 0.0)
          wf = w_avg;
        else
          wf += alpha_fast * (w_avg - wf);
        sc->v[1] = ws;
        sc->v[2] = wf;
      }
    }
    if (done_flag != nullptr)
    {
      __threadfence_system();
      __hip_atomic_store(done_flag, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// particle_filter.cpp:241-246 / :258-266.  total comes from device memory (slot 6) unless
// use_host_total is set (sharded operation: the global total).
__global__ void k_normalize(double* __restrict__ w, int n, const FilterScalars* sc, int use_host_total,
                            double host_total, int global_n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const double total = use_host_total ? host_total : sc->v[6];
  if (total > 0.0)
    w[i] = w[i] / total;
  else
    w[i] = 1.0 / global_n;
}

// Fused tail of ParticleFilter::updateSensor (particle_filter.cpp:237-266): every block folds the
// scoring kernel's per-block weight partials into the total (same fixed tree in every block, so
// all blocks agree bit for bit), normalises its 2048-element tile, and leaves the tile's sum of
// normalised weights for the CDF scan.  Block 0 records the total and updates w_slow / w_fast.
__global__ __launch_bounds__(BPF_RED_BLOCK) void k_normalize_fused(double* __restrict__ w, int n,
                                                                  const double* __restrict__ block_partials,
                                                                  int n_partials, FilterScalars* sc,
                                                                  double alpha_slow, double alpha_fast,
                                                                  double* __restrict__ tile_sums)
{
  __shared__ double s_wave[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += BPF_RED_BLOCK)
    acc += block_partials[i];
  const double total = block_sum_256(acc, s_wave);
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    sc->v[0] = total;
    sc->v[6] = total;
    if (total > 0.0)
    {
      const double w_avg = total / n;
      double ws = sc->v[1], wf = sc->v[2];
      if (ws == 0.0)
        ws = w_avg;
      else
        ws += alpha_slow * (w_avg - ws);
      if (wf == 0.0)
        wf = w_avg;
      else
        wf += alpha_fast * (w_avg - wf);
      sc->v[1] = ws;
      sc->v[2] = wf;
    }
  }
  const size_t base = (size_t)blockIdx.x * BPF_RED_TILE + (size_t)threadIdx.x * BPF_RED_PER_THREAD;
  const double uniform = 1.0 / n;
  double tsum = 0.0;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
    if (base + k < (size_t)n)
    {
      const double v = (total > 0.0) ? w[base + k] / total : uniform;
      w[base + k] = v;
      tsum += v;
    }
  const double tile = block_sum_256(tsum, s_wave);
  if (threadIdx.x == 0)
    tile_sums[blockIdx.x] = tile;
}

// one launch instead of four device-to-device copies
// 16-byte words from pinned host memory to the device: the scan's staging block in front of a scoring launch that has
// no prep launch to carry it (HOST_MODE 2 of k_score_field)
__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16)
    dst[i] = src[i];
}

__global__ void k_copy4(ParticlesDev dst, ParticlesDev src, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  dst.x[i] = src.x[i];
  dst.y[i] = src.y[i];
  dst.th[i] = src.th[i];
  dst.w[i] = src.w[i];
}

// Tail of updateResample for a small resampled set (one block of 1024 threads): weights 1/M
// (particle_filter.cpp:409,458-462) and updateConverged (:170-220).
__device__ __forceinline__ void resample_tail_body(const double* __restrict__ x, const double* __restrict__ y,
                                                   double* __restrict__ w, int n, double thr, FilterScalars* sc,
                                                   int* __restrict__ count_out)
{
  __shared__ double s_x[16], s_y[16];
  __shared__ int s_c[16];
  const double weight = 1.0 / (double)n;
  double ax = 0.0, ay = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024)
  {
    w[i] = weight;
    ax += x[i];
    ay += y[i];
  }
  ax = wave_sum(ax);
  ay = wave_sum(ay);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
  {
    s_x[wave] = ax;
    s_y[wave] = ay;
  }
  __syncthreads();
  double sx = 0.0, sy = 0.0;
  for (int k = 0; k < 16; ++k)
  {
    sx += s_x[k];
    sy += s_y[k];
  }
  const double mx = sx / n, my = sy / n;
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 1024)
    if (fabs(x[i] - mx) <= thr && fabs(y[i] - my) <= thr)
      c++;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    c += __shfl_xor(c, off, 64);
  if (lane == 0)
    s_c[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    int tot = 0;
    for (int k = 0; k < 16; ++k)
      tot += s_c[k];
    *count_out = tot;
    sc->v[3] = sx;
    sc->v[4] = sy;
  }
}

__global__ __launch_bounds__(1024) void k_resample_tail_small(const double* __restrict__ x,
                                                             const double* __restrict__ y,
                                                             double* __restrict__ w, int n, double thr,
                                                             FilterScalars* sc, int* __restrict__ count_out)
{
  resample_tail_body(x, y, w, n, thr, sc, count_out);
}

// folds the scoring kernel's per-block weight partials into scalars[slot] (sharded path: the local total)
// With a mailbox (kernels_mailbox.hpp) the total also goes straight into every peer's memory.
__global__ __launch_bounds__(BPF_RED_BLOCK) void k_fold_partials(const double* __restrict__ partials, int n_partials,
                                                                FilterScalars* sc, int slot, const MailboxDev M,
                                                                int post_parity, unsigned long long post_gen)
{
  __shared__ double s_wave[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += BPF_RED_BLOCK)
    acc += partials[i];
  const double tot = block_sum_256(acc, s_wave);
  if (threadIdx.x == 0)
    sc->v[slot] = tot;
  if (M.world > 0)
    mb_post_total(M, post_parity, post_gen, tot);
}

// Sharded tail for a small resampled set (one block): this rank adopts poses [lo, hi) of the assembled
// window rows (weights 1/M, particle_filter.cpp:409,458-462) and evaluates updateConverged (:170-220)
// over all M poses, which every rank holds.
__global__ __launch_bounds__(1024) void k_shard_tail_small(const double* __restrict__ x_all,
                                                          const double* __restrict__ y_all,
                                                          const double* __restrict__ th_all, int m_total, int lo,
                                                          int hi, ParticlesDev dst, double thr, FilterScalars* sc,
                                                          int* __restrict__ count_out)
{
  __shared__ double s_x[16], s_y[16];
  __shared__ int s_c[16];
  const double weight = 1.0 / (double)m_total;
  double ax = 0.0, ay = 0.0;
  for (int i = threadIdx.x; i < m_total; i += 1024)
  {
    const double xv = x_all[i], yv = y_all[i];
    ax += xv;
    ay += yv;
    if (i >= lo && i < hi)
    {
      dst.x[i - lo] = xv;
      dst.y[i - lo] = yv;
      dst.th[i - lo] = th_all[i];
      dst.w[i - lo] = weight;
    }
  }
  ax = wave_sum(ax);
  ay = wave_sum(ay);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
  {
    s_x[wave] = ax;
    s_y[wave] = ay;
  }
  __syncthreads();
  double sx = 0.0, sy = 0.0;
  for (int k = 0; k < 16; ++k)
  {
    sx += s_x[k];
    sy += s_y[k];
  }
  const double mx = sx / m_total, my = sy / m_total;
  int c = 0;
  for (int i = threadIdx.x; i < m_total; i += 1024)
    if (fabs(x_all[i] - mx) <= thr && fabs(y_all[i] - my) <= thr)
      c++;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    c += __shfl_xor(c, off, 64);
  if (lane == 0)
    s_c[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    int tot = 0;
    for (int k = 0; k < 16; ++k)
      tot += s_c[k];
    *count_out = tot;
    sc->v[3] = sx;
    sc->v[4] = sy;
  }
}

// Sharded variant: the global total is the rank-ordered sum of the gathered per-shard totals;
// thread 0 of block 0 also applies the running-average update with the global numbers.  Tiles of
// 2048 weights per block; each block leaves its tile's sum of normalised weights for the CDF.
__global__ __launch_bounds__(BPF_RED_BLOCK) void k_normalize_gathered(double* __restrict__ w, int n,
                                                                     const double* totals, int world,
                                                                     int global_n, FilterScalars* sc,
                                                                     double alpha_slow, double alpha_fast,
                                                                     double* __restrict__ tile_sums,
                                                                     const MailboxDev M, int wait_parity,
                                                                     unsigned long long wait_gen,
                                                                     const double* __restrict__ fold_partials,
                                                                     int n_fold, const unsigned* totals_failed)
{
  __shared__ double s_wave[4];
  if (M.world > 0)
  {
    // `totals` are this rank's mailbox slots.  When the scoring stage left its fold to this launch, block 0 does
    // what k_fold_partials does (same summation shape) and posts the total to every peer, itself included ...
    if (fold_partials != nullptr && blockIdx.x == 0)
    {
      double acc = 0.0;
      for (int i = threadIdx.x; i < n_fold; i += BPF_RED_BLOCK)
        acc += fold_partials[i];
      const double tot = block_sum_256(acc, s_wave);
      if (threadIdx.x == 0)
        sc->v[0] = tot;
      mb_post_total(M, wait_parity, wait_gen, tot);
    }
    // ... then every block waits until all peers' totals of this update are in.  If they are not (a peer stalled past
    // the bound), the weights stay as the scoring stage left them -- scored, not normalised, the local total in
    // scalars[0] -- and the host finishes the update over its other transport.
    if (!mb_block_wait(M, mb_tot_gen(M.peer[M.rank], wait_parity, 0), wait_gen, 0))
      return;
  }
  else if (totals_failed != nullptr &&
           __hip_atomic_load(totals_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
    return;  // the one-block k_mailbox_wait in front of this launch ran out of time: same rule
  double total = 0.0;
  for (int r = 0; r < world; ++r)
    total += totals[r];
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    sc->v[6] = total;
    if (total > 0.0)
    {
      const double w_avg = total / global_n;
      double ws = sc->v[1], wf = sc->v[2];
      if (ws == 0.0)
        ws = w_avg;
      else
        ws += alpha_slow * (w_avg - ws);
      if (wf == 0.0)
        wf = w_avg;
      else
        wf += alpha_fast * (w_avg - wf);
      sc->v[1] = ws;
      sc->v[2] = wf;
    }
  }
  const size_t base = (size_t)blockIdx.x * BPF_RED_TILE + (size_t)threadIdx.x * BPF_RED_PER_THREAD;
  const double uniform = 1.0 / global_n;
  double tsum = 0.0;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
    if (base + k < (size_t)n)
    {
      const double v = (total > 0.0) ? w[base + k] / total : uniform;
      w[base + k] = v;
      tsum += v;
    }
  const double tile = block_sum_256(tsum, s_wave);
  if (threadIdx.x == 0)
    tile_sums[blockIdx.x] = tile;
}

// ------------------------------------------------------------------ CDF (inclusive scan)
// c[0] = 0, c[i+1] = c[i] + w[i]  (particle_filter.cpp:372-375), evaluated as a fixed-shape
// three-phase scan: per-2048-tile sums, scan of the tile sums by one block, then the
// in-tile prefix.  Rounding differs from the reference's serial chain by a few ulp of the
// running sum; see DESIGN.md "CDF edges".
__device__ __forceinline__ double wave_incl_scan(double v)
{
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const double o = __shfl_up(v, off, 64);
    if (lane >= off)
      v += o;
  }
  return v;
}

__global__ __launch_bounds__(BPF_RED_BLOCK) void k_scan_tile_offsets(double* __restrict__ partials, int n_partials,
                                                                    int* __restrict__ zero_word)
{
  // exclusive scan of the tile sums in place, serial per 256-chunk carry; n_partials is small
  __shared__ double s_wave[4];
  __shared__ double s_carry;
  if (threadIdx.x == 0)
  {
    s_carry = 0.0;
    if (zero_word != nullptr)
      *zero_word = 0;  // the CDF-miss flag of the draw kernels that follow
  }
  __syncthreads();
  for (int base = 0; base < n_partials; base += BPF_RED_BLOCK)
  {
    const int i = base + threadIdx.x;
    const double v = (i < n_partials) ? partials[i] : 0.0;
    const double incl = wave_incl_scan(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63)
      s_wave[wave] = incl;
    __syncthreads();
    double wave_off = 0.0;
    for (int k = 0; k < wave; ++k)
      wave_off += s_wave[k];
    const double carry = s_carry;
    if (i < n_partials)
      partials[i] = carry + (wave_off + (incl - v));
    __syncthreads();
    if (threadIdx.x == BPF_RED_BLOCK - 1)
      s_carry = carry + (wave_off + incl);
    __syncthreads();
  }
}

// tile_values holds exclusive tile offsets (offsets_ready != 0, after k_scan_tile_offsets) or the raw
// tile sums (offsets_ready == 0): then every block adds up the sums of the tiles before it, left to
// right -- the same running sum k_scan_tile_offsets forms -- which saves a launch when there are few tiles.
__global__ __launch_bounds__(BPF_RED_BLOCK) void k_scan_final(const double* __restrict__ w, int n,
                                                             const double* __restrict__ tile_values,
                                                             int offsets_ready, double* __restrict__ cdf,
                                                             int* __restrict__ zero_word, int* __restrict__ guide,
                                                             int* __restrict__ zero_word2 = nullptr,
                                                             double* __restrict__ sum_out = nullptr)
{
  __shared__ double s_wave[4];
  __shared__ double s_tile_off;
  if (threadIdx.x == 0)
  {
    double off = 0.0;
    if (offsets_ready)
      off = tile_values[blockIdx.x];
    else
      for (int t = 0; t < (int)blockIdx.x; ++t)
        off += tile_values[t];
    s_tile_off = off;
    if (blockIdx.x == 0 && zero_word != nullptr)
      *zero_word = 0;  // the CDF-miss flag of the draw kernels that follow
    if (blockIdx.x == 0 && zero_word2 != nullptr)
      *zero_word2 = 0;  // the caller's copy of that flag (sharded stages)
  }
  const size_t base = (size_t)blockIdx.x * BPF_RED_TILE + (size_t)threadIdx.x * BPF_RED_PER_THREAD;
  double v[BPF_RED_PER_THREAD];
  double run = 0.0;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
  {
    const double x = (base + k < (size_t)n) ? w[base + k] : 0.0;
    run += x;
    v[k] = run;
  }
  const double incl = wave_incl_scan(run);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63)
    s_wave[wave] = incl;
  __syncthreads();
  double off = s_tile_off;
  for (int k = 0; k < wave; ++k)
    off += s_wave[k];
  off += incl - run;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
    if (base + k < (size_t)n)
    {
      cdf[base + k + 1] = off + v[k];
      if (sum_out != nullptr && base + k == (size_t)n - 1)
        *sum_out = off + v[k];  // c[n]: the local CDF sum a shard publishes
    }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    cdf[0] = 0.0;
  if (guide != nullptr)
  {
    // guide[j] = the i with c[i] <= j / kCdfGuide < c[i+1] (n where j / kCdfGuide >= c[n]).  This thread's view of
    // c[base] can differ by an ulp from what its neighbour stored there, so an entry may stay unwritten or be
    // off by one now and then: the reader verifies the bracket against the stored CDF and falls back (below).
    double lo_c = (base == 0) ? 0.0 : off;
#pragma unroll
    for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
      if (base + k < (size_t)n)
      {
        const double hi_c = off + v[k];
        for (int j = (int)ceil(lo_c * (double)kCdfGuide); j <= kCdfGuide && (double)j / (double)kCdfGuide < hi_c; ++j)
          guide[j] = (int)(base + k);
        if (base + k == (size_t)n - 1)
          for (int j = (int)ceil(hi_c * (double)kCdfGuide); j <= kCdfGuide; ++j)
            guide[j] = n;
        lo_c = hi_c;
      }
  }
}

// Strict variant: the reference's serial chain, one lane.  Bit-exact, O(n) latency-bound;
// used by the parity tests (BPF_CDF_SERIAL) and as ground truth for the parallel scan.
__global__ void k_scan_serial(const double* __restrict__ w, int n, double* __restrict__ cdf)
{
  if (blockIdx.x != 0 || threadIdx.x != 0)
    return;
  double c = 0.0;
  cdf[0] = c;
  for (int i = 0; i < n; ++i)
  {
    c = c + w[i];
    cdf[i + 1] = c;
  }
}

// ------------------------------------------------------------------ drand48 + selection
__device__ __forceinline__ uint64_t lcg_skip(uint64_t x0, uint64_t n, const LcgJump& J)
{
  const uint64_t mask = (1ull << 48) - 1;
  uint64_t a = 1, c = 0;
  for (int j = 0; n != 0 && j < 48; ++j, n >>= 1)
    if (n & 1)
    {
      // compose: x -> A_j*(a*x + c) + C_j
      a = (a * J.A[j]) & mask;
      c = (c * J.A[j] + J.C[j]) & mask;
    }
  return (a * x0 + c) & mask;
}

// first i with c[i] <= r < c[i+1] (particle_filter.cpp:394-398) by bisection; n on a miss
__device__ __forceinline__ int cdf_find(const double* __restrict__ c, int n, double r)
{
  if (!(r < c[n]) || !(c[0] <= r))
    return n;
  int lo = 0, hi = n;
  while (hi - lo > 1)
  {
    const int mid = lo + ((hi - lo) >> 1);
    if (c[mid] <= r)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// The same answer with a head start: guide[j] brackets the interval of every r in [j, j+1) / kCdfGuide, so the
// bisection runs over a few elements instead of all n (17 dependent loads at n = 10^5 become ~8)
__device__ __forceinline__ int cdf_find_guided(const double* __restrict__ c, int n, double r,
                                               const int* __restrict__ guide)
{
  if (r >= 0.0 && r < 1.0)
  {
    const int j = (int)(r * (double)kCdfGuide);
    int lo = guide[j];
    if ((unsigned)lo < (unsigned)n)
    {
      const int g1 = guide[j + 1];
      int hi = ((unsigned)g1 < (unsigned)n) ? g1 + 1 : n;
      if (hi <= lo)
        hi = n;
      if (c[lo] <= r && r < c[hi])
      {
        while (hi - lo > 1)
        {
          const int mid = lo + ((hi - lo) >> 1);
          if (c[mid] <= r)
            lo = mid;
          else
            hi = mid;
        }
        return lo;
      }
    }
  }
  return cdf_find(c, n, r);
}

// PFKDTree::insertPose key (pf_kdtree.cpp:52-54): floor(pose / {0.5, 0.5, 10 deg})
__device__ __forceinline__ void pose_key(double x, double y, double th, int* key)
{
  const double cell_th = 10 * 3.14159265358979323846 / 180;
  key[0] = (int)floor(x / 0.50);
  key[1] = (int)floor(y / 0.50);
  key[2] = (int)floor(th / cell_th);
}

__device__ __forceinline__ uint64_t lcg_next(uint64_t x)
{
  return (0x5DEECE66Dull * x + 0xBull) & ((1ull << 48) - 1);
}

// Node::randomFreeSpacePose (node.cpp:823-845) from two uniforms, with OccupancyMap::convertMapToWorld
// (occupancy_map.cpp:75-88); free_ij = Node2D::updateFreeSpaceIndices (node_2d.cpp:317-337)
struct FreeSpaceDev
{
  const int2* ij;
  int n;
  int size_x, size_y;
  double origin_x, origin_y, resolution;
};

__device__ __forceinline__ void random_free_space_pose(const FreeSpaceDev& F, double r1, double r2, double* x,
                                                       double* y, double* th)
{
#pragma clang fp contract(off)
  const unsigned idx = (unsigned)(r1 * F.n);
  const int2 c = F.ij[idx];
  *x = F.origin_x + (c.x - F.size_x / 2) * F.resolution;
  *y = F.origin_y + (c.y - F.size_y / 2) * F.resolution;
  *th = r2 * 2 * 3.14159265358979323846 - 3.14159265358979323846;
}

struct DrawArgs
{
  ParticlesDev src;      // set a
  int n_src;
  const double* cdf;     // n_src + 1 (local shard CDF, starting at 0)
  ParticlesDev dst;      // set b (nullable x => no pose scatter)
  int m0, m1;            // global draw indices handled by this launch
  uint64_t rng_state;    // 48-bit state before draw 0
  LcgJump jump;
  int* keys;             // [3*(m - m0)]
  int* src_index;        // [m - m0]
  int* miss_flag;        // set to 1 when a search misses (reference ROS_ASSERT)
  // sharded selection: only draws with cdf_offset <= r < cdf_offset + c[n] belong here
  double cdf_offset;
  int sharded;
  int is_last_shard;
  // zero-copy hand-off of the keys to the host (nullable): keys also go, as three rows of
  // `host_stride` ints, straight into pinned host memory; the block that finishes last publishes
  // `generation` in *host_done so the host can poll instead of issuing a copy and a stream sync
  int* host_keys;
  int host_stride;
  unsigned* done_counter;       // device word, 0 between launches
  volatile unsigned* host_done; // pinned host word
  unsigned generation;
  // w_diff > 0 (kernels_recovery.hpp): position of draw m's test element in the stream, bit 31 = the draw is
  // a random free-space pose (nullptr: w_diff == 0, draw m tests element 2m+1 and takes r from 2m+2)
  const int* chain;
  FreeSpaceDev free_space;
  const int* guide;  // CDF guide table from k_scan_final (nullable)
};

// Multinomial resampler body (particle_filter.cpp:381-414): with w_diff == 0 draw m consumes stream elements
// 2m+1 (compared with w_diff) and 2m+2 (r); with w_diff > 0 the chain says where its elements are.
__device__ __forceinline__ void draw_select_body(const DrawArgs& A)
{
  const int m = A.m0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= A.m1)
    return;
  const int o = m - A.m0;
  uint64_t xs;
  if (A.chain != nullptr)
  {
    const int c = A.chain[m];
    xs = lcg_skip(A.rng_state, (uint64_t)(c & 0x7fffffff) + 1ull, A.jump);
    if (c < 0)
    {
      // :385-388: random_pose_fn_() = Node::randomFreeSpacePose, the next two stream elements
      const double r1 = ldexp((double)xs, -48), r2 = ldexp((double)lcg_next(xs), -48);
      double x, y, th;
      random_free_space_pose(A.free_space, r1, r2, &x, &y, &th);
      if (A.dst.x != nullptr)
      {
        A.dst.x[m] = x;
        A.dst.y[m] = y;
        A.dst.th[m] = th;
      }
      A.src_index[o] = -1;
      int key[3];
      pose_key(x, y, th, key);
      A.keys[3 * o] = key[0];
      A.keys[3 * o + 1] = key[1];
      A.keys[3 * o + 2] = key[2];
      if (A.host_keys != nullptr)
      {
        A.host_keys[o] = key[0];
        A.host_keys[A.host_stride + o] = key[1];
        A.host_keys[2 * A.host_stride + o] = key[2];
      }
      return;
    }
  }
  else
    xs = lcg_skip(A.rng_state, 2ull * (uint64_t)m + 2ull, A.jump);
  double r = ldexp((double)xs, -48);
  int i;
  if (A.sharded)
  {
    const double top = A.cdf_offset + A.cdf[A.n_src];
    const bool mine = (r >= A.cdf_offset) && (r < top || A.is_last_shard);
    if (!mine)
    {
      A.src_index[o] = -1;
      return;
    }
    // search on the shifted CDF: c_global[i] = cdf_offset + c_local[i]
    int lo = 0, hi = A.n_src;
    if (!(r < top))
    {
      atomicExch(A.miss_flag, 1);
      i = A.n_src - 1;
    }
    else
    {
      while (hi - lo > 1)
      {
        const int mid = lo + ((hi - lo) >> 1);
        if (A.cdf_offset + A.cdf[mid] <= r)
          lo = mid;
        else
          hi = mid;
      }
      i = lo;
    }
  }
  else
  {
    i = (A.guide != nullptr) ? cdf_find_guided(A.cdf, A.n_src, r, A.guide) : cdf_find(A.cdf, A.n_src, r);
    if (i >= A.n_src)
    {
      atomicExch(A.miss_flag, 1);
      i = A.n_src - 1;
    }
  }
  const double x = A.src.x[i], y = A.src.y[i], th = A.src.th[i];
  if (A.dst.x != nullptr)
  {
    A.dst.x[m] = x;
    A.dst.y[m] = y;
    A.dst.th[m] = th;
  }
  A.src_index[o] = i;
  int key[3];
  pose_key(x, y, th, key);
  A.keys[3 * o] = key[0];
  A.keys[3 * o + 1] = key[1];
  A.keys[3 * o + 2] = key[2];
  if (A.host_keys != nullptr)
  {
    A.host_keys[o] = key[0];
    A.host_keys[A.host_stride + o] = key[1];
    A.host_keys[2 * A.host_stride + o] = key[2];
  }
}

// Completion hand-off of k_draw_select's zero-copy keys: every thread makes its host-memory stores
// visible system-wide, the block checks in on a device counter, and the last block to do so
// publishes the generation number in pinned host memory (and re-arms the counter).
__device__ __forceinline__ void publish_when_last(unsigned* counter, volatile unsigned* host_done, unsigned generation)
{
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const unsigned prev = atomicAdd(counter, 1u);
    if (prev == gridDim.x - 1)
    {
      *counter = 0;
      __threadfence_system();
      __hip_atomic_store(const_cast<unsigned*>(host_done), generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void k_draw_select(const DrawArgs A)
{
  draw_select_body(A);
  if (A.host_keys != nullptr)
    publish_when_last(A.done_counter, A.host_done, A.generation);
}

// Sharded multinomial draws: every shard evaluates every draw of the window, keeps the ones whose
// r falls into its slice [offset, offset + sums[rank]) of the global CDF (the slices partition
// [0, total) exactly because every rank forms the same left-to-right running sum of `sums`),
// and writes pose bits + histogram key as int64 rows; draws owned by another shard are zeroed.
struct FusedJump
{
  uint64_t a, c;  // x -> a * x + c (mod 2^48) advances the drand48 state by 2 m + 2 elements (draw m)
};

struct WindowArgs
{
  const FusedJump* jump_table;  // nullable: the composed step per draw m < jump_table_n (instead of 48 square-and-multiply rounds)
  int jump_table_n;
  ParticlesDev src;
  int n_src;
  const double* cdf;     // local running sum, c[0] = 0
  const double* coarse;  // nullable: c[min(k << coarse_shift, n_src)], k = 0 .. ((n_src - 1) >> shift) + 1 (k_normalize_cdf's
  int coarse_shift;      // subsample); the block stages it in LDS and a draw brackets itself there first
  const double* sums;    // [world] per-shard CDF sums (or weight totals), rank order
  int sums_are_totals;   // 1: the shard's slice of [0,1) is total_r / sum(totals) by definition
  int rank, world;
  int m0, m1;
  uint64_t rng_state;
  LcgJump jump;
  long long* window;     // [6][stride]
  int stride;
  int* flags;
  const double* targets; // systematic resampling: r of draw m is targets[m - n_random] (nullptr: the drand48 stream)
  // w_diff > 0: `chain` as in DrawArgs (multinomial), or the first n_random samples (systematic) are random
  // free-space poses; they are written by the shard with write_random set, so the window sum holds them once
  const int* chain;
  int n_random;
  int write_random;
  FreeSpaceDev free_space;
  // mailbox exchange (mb.world > 0): a column is stored by its one owner into every peer's window instead of
  // being summed over the shards afterwards; `window` is not used
  MailboxDev mb;
  int mb_parity;
  unsigned long long mb_gen;
  unsigned* mb_counter;
};

// one draw of the window: true when this shard owns column o (out[] then holds pose bits + key)
// Slice [offset, top) of the global CDF owned by this shard.  With CDF sums the slices tile [0, total) exactly.
// With weight totals (one exchange less) the slice of shard q is defined as total_q / T, the same
// quotient on every rank; the shard's own running sum is used inside it and its last particle
// takes whatever rounding leaves between the end of that sum and the end of the slice.
__device__ __forceinline__ void shard_slice(const double* sums, int sums_are_totals, int rank, int world,
                                            double* offset_out, double* top_out)
{
  double T = 1.0;
  if (sums_are_totals)
  {
    T = 0.0;
    for (int r = 0; r < world; ++r)
      T += sums[r];
  }
  double offset = 0.0;
  for (int r = 0; r < rank; ++r)
    offset += sums_are_totals ? sums[r] / T : sums[r];
  *offset_out = offset;
  *top_out = offset + (sums_are_totals ? sums[rank] / T : sums[rank]);
}

__device__ __forceinline__ bool draw_window_column(const WindowArgs& A, int o, long long out[6], double offset,
                                                   double top, const double* s_coarse = nullptr)
{
  const int m = A.m0 + o;
  double r = 0.0;
  bool random = false;
  double rx = 0.0, ry = 0.0, rth = 0.0;
  if (A.targets != nullptr)
  {
    if (m < A.n_random)
    {
      random = true;  // particle_filter.cpp:316-324: stream elements 2m+2, 2m+3 (element 1 is the start)
      const uint64_t xs = lcg_skip(A.rng_state, 2ull * (uint64_t)m + 2ull, A.jump);
      random_free_space_pose(A.free_space, ldexp((double)xs, -48), ldexp((double)lcg_next(xs), -48), &rx, &ry, &rth);
    }
    else
      r = A.targets[m - A.n_random];
  }
  else if (A.chain != nullptr)
  {
    const int c = A.chain[m];
    const uint64_t xs = lcg_skip(A.rng_state, (uint64_t)(c & 0x7fffffff) + 1ull, A.jump);
    if (c < 0)
    {
      random = true;  // :385-388
      random_free_space_pose(A.free_space, ldexp((double)xs, -48), ldexp((double)lcg_next(xs), -48), &rx, &ry, &rth);
    }
    else
      r = ldexp((double)xs, -48);
  }
  else if (A.jump_table != nullptr && m < A.jump_table_n)
  {
    const FusedJump J = A.jump_table[m];
    r = ldexp((double)((J.a * A.rng_state + J.c) & ((1ull << 48) - 1)), -48);
  }
  else
  {
    const uint64_t xs = lcg_skip(A.rng_state, 2ull * (uint64_t)m + 2ull, A.jump);
    r = ldexp((double)xs, -48);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k)
    out[k] = 0;
  if (random)
  {
    if (A.write_random)
    {
      int key[3];
      pose_key(rx, ry, rth, key);
      out[0] = __double_as_longlong(rx);
      out[1] = __double_as_longlong(ry);
      out[2] = __double_as_longlong(rth);
      out[3] = key[0];
      out[4] = key[1];
      out[5] = key[2];
    }
    return A.write_random != 0;
  }
  const bool last = A.rank == A.world - 1;
  const bool mine = (r >= offset) && (r < top || last);
  if (mine && A.n_src <= 0)
  {
    // an empty shard (uneven first split, or fewer samples than ranks after a resample) owns no particle: a draw that
    // rounding sends here is the reference's failed search (ROS_ASSERT(i < sample_count)); no pose is read
    atomicExch(A.flags, 1);
    return false;
  }
  if (mine)
  {
    int i;
    if (!(r < top))
    {
      atomicExch(A.flags, 1);  // reference: ROS_ASSERT(i < sample_count)
      i = A.n_src - 1;
    }
    else if (!(r < offset + A.cdf[A.n_src]))
      i = A.n_src - 1;  // inside the slice but past the shard's running sum (rounding): last particle
    else
    {
      int lo = 0, hi = A.n_src;  // offset + c[lo] <= r < offset + c[hi]
      if (s_coarse != nullptr)
      {
        // the bracket of 2^shift CDF values first, from the subsample staged in LDS: ~12 of the ~17 bisection steps
        // without a trip to memory (same comparisons on the same values, so the same interval)
        int kl = 0, kh = ((A.n_src - 1) >> A.coarse_shift) + 1;
        while (kh - kl > 1)
        {
          const int mid = kl + ((kh - kl) >> 1);
          if (offset + s_coarse[mid] <= r)
            kl = mid;
          else
            kh = mid;
        }
        lo = kl << A.coarse_shift;
        hi = min(kh << A.coarse_shift, A.n_src);
      }
      while (hi - lo > 1)
      {
        const int mid = lo + ((hi - lo) >> 1);
        if (offset + A.cdf[mid] <= r)
          lo = mid;
        else
          hi = mid;
      }
      i = lo;
    }
    const double x = A.src.x[i], y = A.src.y[i], th = A.src.th[i];
    int key[3];
    pose_key(x, y, th, key);
    out[0] = __double_as_longlong(x);
    out[1] = __double_as_longlong(y);
    out[2] = __double_as_longlong(th);
    out[3] = key[0];
    out[4] = key[1];
    out[5] = key[2];
  }
  return mine;
}

// dynamic LDS: the CDF subsample when A.coarse is given ((((n_src - 1) >> shift) + 2) doubles), else none
__global__ void k_draw_window(const WindowArgs A)
{
  extern __shared__ __align__(16) unsigned char draw_smem[];
  double* s_coarse = nullptr;
  if (A.coarse != nullptr)
  {
    s_coarse = reinterpret_cast<double*>(draw_smem);
    const int n_coarse = ((A.n_src - 1) >> A.coarse_shift) + 1;
    for (int k = threadIdx.x; k <= n_coarse; k += blockDim.x)
      s_coarse[k] = A.coarse[k];
    __syncthreads();
  }
  // the shards' sums sit in uncached mailbox memory (or come from a collective): one round of loads per block, the
  // slice from LDS copies (same additions in the same order as every thread used to do from memory)
  __shared__ double s_sums[kMailboxMaxWorld];
  __shared__ double s_slice[2];
  if ((int)threadIdx.x < A.world && A.world <= kMailboxMaxWorld)
    s_sums[threadIdx.x] = A.sums[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0)
    shard_slice(A.world <= kMailboxMaxWorld ? s_sums : A.sums, A.sums_are_totals, A.rank, A.world, &s_slice[0],
                &s_slice[1]);
  __syncthreads();
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = A.m0 + o < A.m1;
  long long out[6] = { 0, 0, 0, 0, 0, 0 };
  const bool owned = live && draw_window_column(A, o, out, s_slice[0], s_slice[1], s_coarse);
  if (A.mb.world > 0)
  {
    if (owned)
      for (int r = 0; r < A.mb.world; ++r)
      {
        long long* win = mb_window(A.mb.peer[r], A.mb_parity, A.mb.max_window);
#pragma unroll
        for (int k = 0; k < 6; ++k)
          win[(size_t)k * (size_t)A.mb.max_window + o] = out[k];
      }
    mb_window_done_when_last(A.mb, A.mb_parity, A.mb_gen, A.mb_counter);
    return;
  }
  if (live)
  {
#pragma unroll
    for (int k = 0; k < 6; ++k)
      A.window[(size_t)k * A.stride + o] = out[k];
  }
}

__global__ void k_adopt(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ th,
                        ParticlesDev dst, int n, double weight)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  dst.x[i] = x[i];
  dst.y[i] = y[i];
  dst.th[i] = th[i];
  dst.w[i] = weight;
}

struct SystematicArgs
{
  ParticlesDev src;
  int n_src;
  const double* cdf;
  const double* targets;
  ParticlesDev dst;
  int count;
  int* keys;
  int* src_index;
  int* miss_flag;
  // zero-copy hand-off (as DrawArgs): keys as three rows in pinned host memory + generation word
  int* host_keys;
  int host_stride;
  unsigned* done_counter;
  volatile unsigned* host_done;
  unsigned generation;
  // w_diff > 0 (particle_filter.cpp:295-324): the first n_random samples are random free-space poses from stream
  // elements 2i+2, 2i+3 (element 1 is the systematic start); targets[] then belongs to samples n_random ..
  int n_random;
  uint64_t rng_state;
  LcgJump jump;
  FreeSpaceDev free_space;
};

// The reference walks the CDF cyclically from the previous hit (particle_filter.cpp:329-336);
// the interval that contains a target is unique, so each target is resolved independently.
__global__ void k_systematic_select(const SystematicArgs A)
{
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < A.count)
  {
    double x, y, th;
    int i = -1;
    if (m < A.n_random)
    {
      const uint64_t xs = lcg_skip(A.rng_state, 2ull * (uint64_t)m + 2ull, A.jump);
      random_free_space_pose(A.free_space, ldexp((double)xs, -48), ldexp((double)lcg_next(xs), -48), &x, &y, &th);
    }
    else
    {
      i = cdf_find(A.cdf, A.n_src, A.targets[m - A.n_random]);
      if (i >= A.n_src)
      {
        atomicExch(A.miss_flag, 1);  // the reference never leaves its while loop here
        i = A.n_src - 1;
      }
      x = A.src.x[i];
      y = A.src.y[i];
      th = A.src.th[i];
    }
    A.dst.x[m] = x;
    A.dst.y[m] = y;
    A.dst.th[m] = th;
    A.src_index[m] = i;
    int key[3];
    pose_key(x, y, th, key);
    if (A.host_keys != nullptr)
    {
      A.host_keys[m] = key[0];
      A.host_keys[A.host_stride + m] = key[1];
      A.host_keys[2 * A.host_stride + m] = key[2];
    }
    else
    {
      A.keys[3 * m] = key[0];
      A.keys[3 * m + 1] = key[1];
      A.keys[3 * m + 2] = key[2];
    }
  }
  if (A.host_keys != nullptr)
    publish_when_last(A.done_counter, A.host_done, A.generation);
}

// rows 3..5 of an int64 draw window -> three int rows in pinned host memory, then the generation word
// With a mailbox window every block first waits for the "done" words of all shards (kernels_mailbox.hpp).
__global__ void k_publish_window_keys(const long long* window, int stride, int n, int* host_keys,
                                      unsigned* done_counter, volatile unsigned* host_done, unsigned generation,
                                      const MailboxDev M, int wait_parity, unsigned long long wait_gen)
{
  if (M.world > 0)
    (void)mb_block_wait(M, mb_win_done(M.peer[M.rank], wait_parity, 0), wait_gen, 1);  // (the host checks the flag)
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n)
  {
    host_keys[q] = (int)window[(size_t)3 * stride + q];
    host_keys[n + q] = (int)window[(size_t)4 * stride + q];
    host_keys[2 * n + q] = (int)window[(size_t)5 * stride + q];
  }
  publish_when_last(done_counter, host_done, generation);
}

// rows 3..5 of an int64 draw window -> int key triples (AoS) on the device
__global__ void k_window_keys_to_aos(const long long* window, int stride, int n, int* __restrict__ keys,
                                     const MailboxDev M, int wait_parity, unsigned long long wait_gen)
{
  if (M.world > 0)
    (void)mb_block_wait(M, mb_win_done(M.peer[M.rank], wait_parity, 0), wait_gen, 1);  // (the host checks the flag)
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n)
  {
    keys[3 * (size_t)q] = (int)window[(size_t)3 * stride + q];
    keys[3 * (size_t)q + 1] = (int)window[(size_t)4 * stride + q];
    keys[3 * (size_t)q + 2] = (int)window[(size_t)5 * stride + q];
  }
}

// ------------------------------------------------------------------ updateConverged
// particle_filter.cpp:170-220: mean x / y, then the count of particles within
// dist_threshold of the mean on both axes.
__global__ __launch_bounds__(BPF_RED_BLOCK) void k_count_converged(const double* __restrict__ x,
                                                                  const double* __restrict__ y, int n,
                                                                  const FilterScalars* sc, double thr,
                                                                  int* __restrict__ count)
{
  __shared__ int s_cnt[4];
  const double mx = sc->v[3] / n, my = sc->v[4] / n;
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (fabs(x[i] - mx) <= thr && fabs(y[i] - my) <= thr)
      c++;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    c += __shfl_xor(c, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    s_cnt[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(count, s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
}

// The same in two launches instead of six stream operations (two sums of two launches each, a memset, the count):
// k_sum_xy_partials = k_sum_partials for x and y at once (same tiles, same shape; block 0 also zeroes the count),
// k_converged_count = every block folds the tile partials exactly as k_sum_final does -- so the means are the same
// bits -- and counts its share.
__global__ __launch_bounds__(BPF_RED_BLOCK) void k_sum_xy_partials(const double* __restrict__ x,
                                                                  const double* __restrict__ y, int n,
                                                                  double* __restrict__ px, double* __restrict__ py,
                                                                  int* __restrict__ count)
{
  __shared__ double s_wave[4];
  const size_t base = (size_t)blockIdx.x * BPF_RED_TILE + (size_t)threadIdx.x * BPF_RED_PER_THREAD;
  double ax = 0.0, ay = 0.0;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
    if (base + k < (size_t)n)
    {
      ax += x[base + k];
      ay += y[base + k];
    }
  const double tx = block_sum_256(ax, s_wave);
  const double ty = block_sum_256(ay, s_wave);
  if (threadIdx.x == 0)
  {
    px[blockIdx.x] = tx;
    py[blockIdx.x] = ty;
    if (blockIdx.x == 0)
      *count = 0;
  }
}

__global__ __launch_bounds__(BPF_RED_BLOCK) void k_converged_count(const double* __restrict__ x,
                                                                  const double* __restrict__ y, int n,
                                                                  const double* __restrict__ px,
                                                                  const double* __restrict__ py, int n_partials,
                                                                  FilterScalars* sc, double thr, int* __restrict__ count)
{
  __shared__ double s_wave[4];
  __shared__ int s_cnt[4];
  double ax = 0.0, ay = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += BPF_RED_BLOCK)
  {
    ax += px[i];
    ay += py[i];
  }
  const double sx = block_sum_256(ax, s_wave);
  const double sy = block_sum_256(ay, s_wave);
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    sc->v[3] = sx;
    sc->v[4] = sy;
  }
  const double mx = sx / n, my = sy / n;
  int c = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (fabs(x[i] - mx) <= thr && fabs(y[i] - my) <= thr)
      c++;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    c += __shfl_xor(c, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    s_cnt[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(count, s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
}

// ------------------------------------------------------------------ distance LUT (exact EDT)
// Separable capped Euclidean distance transform on the reference's integer lattice:
// value = float(sqrt(a^2+b^2) * res) for the nearest occupied cell if sqrt(a^2+b^2) <= radius,
// else float(max_dist)   (value lattice of occupancy_map.cpp:122-135,227-245).
__global__ void k_edt_rows(const int8_t* __restrict__ cells, int sx, int sy, int radius, int* __restrict__ dxrow)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= sx || y >= sy)
    return;
  int best = radius + 1;
  for (int d = 0; d <= radius && d < best; ++d)
  {
    const int xl = x - d, xr = x + d;
    if ((xl >= 0 && cells[xl + (size_t)y * sx] == 1) || (xr < sx && cells[xr + (size_t)y * sx] == 1))
      best = d;
  }
  dxrow[x + (size_t)y * sx] = best;
}

__global__ void k_edt_cols(const int* __restrict__ dxrow, int sx, int sy, int radius, double res, double max_dist,
                           float* __restrict__ lut)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= sx || y >= sy)
    return;
  const int r2 = radius * radius;
  int best = r2 + 1;
  for (int d = -radius; d <= radius; ++d)
  {
    const int yy = y + d;
    if (yy < 0 || yy >= sy)
      continue;
    const int a = dxrow[x + (size_t)yy * sx];
    if (a > radius)
      continue;
    const int v = a * a + d * d;
    if (v < best)
      best = v;
  }
  lut[x + (size_t)y * sx] = (best <= r2) ? (float)(sqrt((double)best) * res) : (float)max_dist;
}

}  // namespace bpf
