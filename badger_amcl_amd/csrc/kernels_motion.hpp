// Motion update on the device: Odom::updateAction (src/amcl/sensors/odom.cpp:74-301) with
// PDFGaussian::draw (src/amcl/pf/pdf_gaussian.cpp:77-97).
//
// The reference draws 3 Gaussians per particle, in particle order, from the ONE global drand48
// stream, each by polar Box-Muller with rejection -- a serial chain with a data-dependent number
// of uniforms per draw.  What makes it parallel:
//   * the stream is an LCG, so element k is reachable in O(log k) (jump tables, as the resampler);
//   * an *attempt* is two consecutive uniforms; whether it is accepted (0 < x1^2 + x2^2 <= 1)
//     depends on nothing else, and the j-th Gaussian of the whole update is the j-th accepted
//     attempt -- a stream compaction (count per tile, scan, scatter);
//   * the `r == 0.0` re-draws (pdf_gaussian.cpp:83-92) skip an element without consuming an
//     attempt.  The generator visits state 0 once per period (2^48), so a window holds at most one
//     such element; its position is found by the counting pass and handed to a re-run as
//     `zero_at`, which shifts every later element by one.
// The accept test and the uniforms are exact (integer LCG, products and sum rounded separately);
// log / sin / cos come from the device math library, so poses agree with a libm build of the
// reference to a few ulp, while the number of uniforms consumed -- the drand48 state handed to the
// resampler afterwards -- is exact.
#pragma once
#include "device_types.hpp"
#include "kernels_pf.hpp"
#include "kernels_score.hpp"

namespace bpf
{

constexpr int kMotionRun = 8;                      // attempts per thread
constexpr int kMotionTile = 256 * kMotionRun;      // attempts per block
constexpr long long kNoZero = 0x7fffffffffffffffll;

struct MotionRngArgs
{
  uint64_t rng_state;       // drand48 state before the update
  long long zero_at;        // raw index (1-based step count) of the element that is exactly 0.0, or kNoZero
  long long n_attempts;
  long long need_total;     // Gaussians the whole update consumes (3 x global particle count)
  long long gauss_first;    // first Gaussian rank this engine materialises (3 x first global index)
  long long gauss_count;
  int* tile_counts;         // [tiles]
  long long* tile_offsets;  // [tiles + 1]
  double* gauss;            // [gauss_count] PDFGaussian::draw(sd[rank % 3]) of this engine's ranks
  double sd[3];             // the three draw() arguments of a particle, in draw order
  long long* result;        // [0] uniforms consumed, [1] zero position seen (or -1), [2] attempts accepted
  LcgJump jump;
};

// Walks the kMotionRun attempts of one thread.  f(k, accepted, x2, w, raw_index_of_second_uniform).
template <class F>
__device__ __forceinline__ void motion_attempts(const MotionRngArgs& A, long long t0, F&& f)
{
#pragma clang fp contract(off)
  // effective element e (1-based) sits at raw index e, or e + 1 from the zero element on
  const long long e0 = 2 * t0 + 1;
  long long raw = e0 + (e0 >= A.zero_at ? 1 : 0);
  uint64_t x = lcg_skip(A.rng_state, (uint64_t)raw, A.jump);
  for (int k = 0; k < kMotionRun; ++k)
  {
    if (t0 + k >= A.n_attempts)
      break;
    if (k > 0)
    {
      ++raw;
      x = lcg_next(x);
      if (raw == A.zero_at)
      {
        ++raw;
        x = lcg_next(x);
      }
    }
    if (x == 0)
      atomicMax((unsigned long long*)&A.result[1], (unsigned long long)raw);  // result[1] starts at 0 = "none"
    const double x1 = 2.0 * ldexp((double)x, -48) - 1.0;
    ++raw;
    x = lcg_next(x);
    if (raw == A.zero_at)
    {
      ++raw;
      x = lcg_next(x);
    }
    if (x == 0)
      atomicMax((unsigned long long*)&A.result[1], (unsigned long long)raw);
    const double x2 = 2.0 * ldexp((double)x, -48) - 1.0;
    const double w = x1 * x1 + x2 * x2;
    const bool ok = !(w > 1.0 || w == 0.0);
    f(k, ok, x2, w, raw);
  }
}

__global__ __launch_bounds__(256) void k_motion_count(const MotionRngArgs A)
{
  __shared__ int s_w[4];
  const long long t0 = ((long long)blockIdx.x * 256 + threadIdx.x) * kMotionRun;
  int cnt = 0;
  if (t0 < A.n_attempts)
    motion_attempts(A, t0, [&](int, bool ok, double, double, long long) { cnt += ok ? 1 : 0; });
  for (int o = 32; o > 0; o >>= 1)
    cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0)
    s_w[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0)
    A.tile_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// one block: exclusive scan of the tile counts
__global__ __launch_bounds__(1024) void k_motion_offsets(const MotionRngArgs A, int tiles)
{
  __shared__ long long s_part[1024];
  const int tid = threadIdx.x;
  const int per = (tiles + 1023) / 1024;
  const int lo = min(tid * per, tiles), hi = min(lo + per, tiles);
  long long sum = 0;
  for (int i = lo; i < hi; ++i)
    sum += A.tile_counts[i];
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1)
  {
    const long long v = (tid >= o) ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  long long run = s_part[tid] - sum;
  for (int i = lo; i < hi; ++i)
  {
    A.tile_offsets[i] = run;
    run += A.tile_counts[i];
  }
  if (tid == 1023)
  {
    A.tile_offsets[tiles] = s_part[1023];
    A.result[2] = s_part[1023];
  }
}

__global__ __launch_bounds__(256) void k_motion_gauss(const MotionRngArgs A)
{
  __shared__ int s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long t0 = ((long long)blockIdx.x * 256 + tid) * kMotionRun;
  const long long base = A.tile_offsets[blockIdx.x];
  const long long end = A.tile_offsets[blockIdx.x + 1];
  // a tile is wanted if it holds ranks this engine materialises, or the update's last rank
  const bool overlaps = end > A.gauss_first && base < A.gauss_first + A.gauss_count;
  const bool holds_last = base < A.need_total && end >= A.need_total;
  if (!overlaps && !holds_last)
    return;
  int cnt = 0;
  if (t0 < A.n_attempts)
    motion_attempts(A, t0, [&](int, bool ok, double, double, long long) { cnt += ok ? 1 : 0; });
  // exclusive scan of cnt over the block
  int incl = cnt;
  for (int o = 1; o < 64; o <<= 1)
  {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o)
      incl += v;
  }
  if (lane == 63)
    s_w[wave] = incl;
  __syncthreads();
  int wave_base = 0;
  for (int q = 0; q < wave; ++q)
    wave_base += s_w[q];
  long long rank = base + wave_base + (incl - cnt);
  if (t0 < A.n_attempts)
    motion_attempts(A, t0, [&](int, bool ok, double x2, double w, long long raw) {
      if (!ok)
        return;
      const long long j = rank++;
      const long long o = j - A.gauss_first;
      if (o >= 0 && o < A.gauss_count)
        A.gauss[o] = (A.sd[j % 3] * x2 * sqrt(-2.0 * log(w) / w));  // pdf_gaussian.cpp:96
      if (j == A.need_total - 1)
        A.result[0] = raw;  // the update consumed the stream up to and including this uniform
    });
}

// per-update constants, formed on the host with libm exactly as odom.cpp forms them before its loop
struct MotionModelDev
{
  int model;          // OdomModelType: 0 diff, 1 omni, 2 diff-corrected, 3 omni-corrected, 4 gaussian
  double sd[3];       // the three draw() arguments in draw order (applied by k_motion_gauss)
  double delta_trans, delta_rot;
  double bearing0;    // angleDiff(atan2(dy, dx), old_pose[2])            (omni*, gaussian)
  double rot1, rot2;  // delta_rot1, delta_rot2                            (diff*)
  double half_rot;    // delta[2] / 2                                      (gaussian)
};

// normalize_angle_dev: kernels_score.hpp (angles::normalize_angle, Noetic form)
__global__ void k_motion_apply(const ParticlesDev src, const ParticlesDev dst, int n, const MotionModelDev M,
                               const double* __restrict__ gauss)
{
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  double x = src.x[i], y = src.y[i], th = src.th[i];
  const double g0 = gauss[3 * (size_t)i], g1 = gauss[3 * (size_t)i + 1], g2 = gauss[3 * (size_t)i + 2];
  if (M.model == 1 || M.model == 3)
  {
    // odom.cpp:104-122 / :190-208
    const double bearing = M.bearing0 + th;
    double sn, cs;
    sincos(bearing, &sn, &cs);
    const double trans_hat = M.delta_trans + g0;
    const double rot_hat = M.delta_rot + g1;
    const double strafe_hat = 0 + g2;
    x += (trans_hat * cs + strafe_hat * sn);
    y += (trans_hat * sn - strafe_hat * cs);
    th += rot_hat;
  }
  else if (M.model == 0 || M.model == 2)
  {
    // :151-168 / :230-250
    const double rot1_hat = normalize_angle_dev(M.rot1 - g0);
    const double trans_hat = M.delta_trans - g1;
    const double rot2_hat = normalize_angle_dev(M.rot2 - g2);
    double sn, cs;
    sincos(th + rot1_hat, &sn, &cs);
    x += trans_hat * cs;
    y += trans_hat * sn;
    th += rot1_hat + rot2_hat;
  }
  else
  {
    // :272-296: draws in the order trans, strafe, rot
    double sh, ch, sn, cs;
    sincos(th + M.half_rot, &sh, &ch);
    sincos(M.bearing0 + th, &sn, &cs);
    x += (M.delta_trans * cs);
    y += (M.delta_trans * sn);
    th += M.delta_rot;
    x += (g0 * ch + g1 * sh);
    y += (g0 * sh - g1 * ch);
    th += g2;
  }
  dst.x[i] = x;
  dst.y[i] = y;
  dst.th[i] = th;
  dst.w[i] = src.w[i];
}

// ParticleFilter::initWithGaussian (particle_filter.cpp:105-132): pose = x + cr * r, r[k] = draw(cd[k]) -- the
// sum over j is formed term by term as PDFGaussian::sample does (pdf_gaussian.cpp:63-68)
__global__ void k_init_gaussian(ParticlesDev dst, int n, const double* __restrict__ gauss, double mx, double my,
                                double mth, const double* __restrict__ cr9, double weight)
{
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const double r[3] = { gauss[3 * (size_t)i], gauss[3 * (size_t)i + 1], gauss[3 * (size_t)i + 2] };
  const double m[3] = { mx, my, mth };
  double v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
  {
    v[k] = m[k];
#pragma unroll
    for (int j = 0; j < 3; ++j)
      v[k] += cr9[3 * k + j] * r[j];
  }
  dst.x[i] = v[0];
  dst.y[i] = v[1];
  dst.th[i] = v[2];
  dst.w[i] = weight;
}

// ParticleFilter::initWithPoseFn (particle_filter.cpp:135-163) with pose_fn = Node::randomFreeSpacePose:
// sample i takes stream elements 2i+1, 2i+2
__global__ void k_init_free_space(ParticlesDev dst, int n, uint64_t rng_state, LcgJump jump, FreeSpaceDev F,
                                  double weight)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const uint64_t xs = lcg_skip(rng_state, 2ull * (uint64_t)i + 1ull, jump);
  double x, y, th;
  random_free_space_pose(F, ldexp((double)xs, -48), ldexp((double)lcg_next(xs), -48), &x, &y, &th);
  dst.x[i] = x;
  dst.y[i] = y;
  dst.th[i] = th;
  dst.w[i] = weight;
}

// histogram keys of a set (AoS triples), for the tree of a freshly initialised set
__global__ void k_set_keys(ParticlesDev p, int n, int* __restrict__ keys)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    pose_key(p.x[i], p.y[i], p.th[i], &keys[3 * (size_t)i]);
}

}  // namespace bpf
