// 3-D point-cloud scoring against the OctoMap distance LUT:
// PointCloudScanner::calcPointCloudModel{,Gompertz} + recalcWeight
// (src/amcl/sensors/point_cloud_scanner.cpp:132-229) with OctoMap::convertWorldToMap /
// getDistanceToObject (src/amcl/map/octomap.cpp:98-121,336-355).
//
// The reference transforms the whole cloud once per particle through tf2 / pcl (four full-cloud
// copies per particle, point_cloud_scanner.cpp:231-248).  Here the cloud is cut into chunks that
// live in LDS; a block owns one chunk and streams a slab of particles past it (wave = particle,
// lanes = points), so a point is fetched from HBM once per slab instead of once per particle.
// Per-particle partial sums go to a [chunk][particle] buffer and a finishing kernel folds them in
// chunk order and applies the model epilogue.
//
// Arithmetic of the per-point transform (third-party tf2_sensor_msgs, PARITY UNPINNED -- see
// DESIGN.md): float rotation matrix from the float quaternion, then R*p + t in float with
// separate multiplies and adds (no FMA contraction), as a baseline x86 build evaluates it.
#pragma once
#include "device_types.hpp"
#include "kernels_score.hpp"

namespace bpf
{

struct Map3dDev
{
  const uint32_t* pose_indices;
  const uint8_t* distance_ratios;
  int min_c[3], max_c[3];
  int width;
  double resolution;
  double inv_resolution;  // correctly rounded 1/resolution
};

struct CloudModelDev
{
  int model;  // 0 plain, 1 Gompertz
  GompertzDev g;
  double off_map_factor;
  double tf_xyz[3];
  double tf_quat[4];  // x y z w
};

constexpr int kCloudChunk = 2048;  // points per LDS chunk (24 KB as float SoA -> 6 blocks per CU)

// point_cloud_scanner.cpp:231-248 restated: q = q_yaw * q_scanner (double), t = R_yaw*t_s + (x,y,0),
// narrowed to float, Eigen's quaternion->matrix formula in float.
__global__ void k_cloud_affine(ParticlesDev p, int n, CloudModelDev M, float* __restrict__ affine)
{
// HIP's __fmul_rn / __fadd_rn are plain operators, and hipcc contracts a*b+c into an FMA by default;
// the stated semantic is separate roundings, so contraction is switched off for this function.
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const double px = p.x[i], py = p.y[i], pth = p.th[i];
  double sh, ch;
  sincos(pth * 0.5, &sh, &ch);
  const double yq[4] = { 0.0, 0.0, sh, ch };
  const double* s = M.tf_quat;
  double q[4];
  q[3] = yq[3] * s[3] - yq[0] * s[0] - yq[1] * s[1] - yq[2] * s[2];
  q[0] = yq[3] * s[0] + yq[0] * s[3] + yq[1] * s[2] - yq[2] * s[1];
  q[1] = yq[3] * s[1] + yq[1] * s[3] + yq[2] * s[0] - yq[0] * s[2];
  q[2] = yq[3] * s[2] + yq[2] * s[3] + yq[0] * s[1] - yq[1] * s[0];
  double sy, cy;
  sincos(pth, &sy, &cy);
  const double t0 = cy * M.tf_xyz[0] - sy * M.tf_xyz[1] + px;
  const double t1 = sy * M.tf_xyz[0] + cy * M.tf_xyz[1] + py;
  const double t2 = M.tf_xyz[2] + 0.0;
  const float x = (float)q[0], y = (float)q[1], z = (float)q[2], w = (float)q[3];
  // plain operators under `fp contract(off)`: each product and sum is rounded separately
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  float* a = affine + (size_t)i * 12;
  a[0] = 1.f - (tyy + tzz);
  a[1] = txy - twz;
  a[2] = txz + twy;
  a[3] = txy + twz;
  a[4] = 1.f - (txx + tzz);
  a[5] = tyz - twx;
  a[6] = txz - twy;
  a[7] = tyz + twx;
  a[8] = 1.f - (txx + tyy);
  a[9] = (float)t0;
  a[10] = (float)t1;
  a[11] = (float)t2;
}

// octomap.cpp:102-107: floor(v / resolution + 0.5).  The quotient is formed as a corrected
// reciprocal multiply (q = v*r; q += fma(-q, res, v)*r), which reproduces the correctly rounded
// division except in vanishingly rare double-rounding cases.
// When 1/resolution is exactly representable in <= 29 bits and rounds the same way (checked on the
// host: resolution 0.05 -> 20, 0.1 -> 10, 0.025 -> 40 ...), v * rinv is exact for a float v and equals
// the correctly rounded v / resolution, so one FMA gives the reference's value.
__device__ __forceinline__ int voxel_of_exact(float v, double rinv)
{
  const double f = floor(fma((double)v, rinv, 0.5));
  return (f == f) ? (int)f : 0x7fffffff;
}

__device__ __forceinline__ int voxel_of(float v, double res, double rinv)
{
#pragma clang fp contract(off)
  const double d = (double)v;
  double q = d * rinv;
  const double rem = fma(-q, res, d);
  q = fma(rem, rinv, q);
  const double f = floor(q + 0.5);
  return (f == f) ? (int)f : 0x7fffffff;
}

struct CloudScoreArgs
{
  int n;
  const float* affine;   // [n][12]
  const float* points;   // [3][n_points] SoA
  int n_points;
  Map3dDev map;
  const double* table;   // [257]: per distance ratio, [256] = off map
  double* partials;      // [n_chunks][n]
  int slabs;
  // graded partition (round_count[0] > 0, see k_score_field): the waves of slab y own round_count[r] particles each,
  // r = the placement round of the slab's blocks, starting at round_base[r] + ((y - round_first_slab[r]) * 4 + wave)
  // * round_count[r]
  int round_count[8];
  int round_base[8];
  int round_first_slab[8];
  int n_rounds;
};

#ifdef BPF_PHASE_TIMING
__device__ unsigned long long g_cloud_span[8192][2];  // diagnostic builds: per-wave start / end (100 MHz clock)
#endif

template <bool EXACT_RINV>
__global__ __launch_bounds__(256) void k_cloud_score(const CloudScoreArgs A)
{
#pragma clang fp contract(off)
#ifdef BPF_PHASE_TIMING
  const long long _w0 = wall_clock64();
#endif
  __shared__ float s_pts[3][kCloudChunk];
  __shared__ double s_table[257];
  const int chunk = blockIdx.x;
  const int p0 = chunk * kCloudChunk;
  const int np = min(kCloudChunk, A.n_points - p0);
  for (int i = threadIdx.x; i < np; i += 256)
  {
    s_pts[0][i] = A.points[p0 + i];
    s_pts[1][i] = A.points[(size_t)A.n_points + p0 + i];
    s_pts[2][i] = A.points[2 * (size_t)A.n_points + p0 + i];
  }
  for (int i = threadIdx.x; i < 257; i += 256)
    s_table[i] = A.table[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Map3dDev& M = A.map;
  const int span_x = M.max_c[0] - M.min_c[0], span_y = M.max_c[1] - M.min_c[1], span_z = M.max_c[2] - M.min_c[2];

  int j_begin = blockIdx.y * 4 + wave, j_end = A.n, j_step = A.slabs * 4;
  if (A.round_count[0] > 0)
  {
    int r = 0;
    while (r + 1 < A.n_rounds && (int)blockIdx.y >= A.round_first_slab[r + 1])
      ++r;
    j_begin = min(A.n, A.round_base[r] + (((int)blockIdx.y - A.round_first_slab[r]) * 4 + wave) * A.round_count[r]);
    j_end = min(A.n, j_begin + A.round_count[r]);
    j_step = 1;
  }
  for (int jv = j_begin; jv < j_end; jv += j_step)
  {
    // the particle index is wave-uniform: make it scalar so the affine comes in through s_load
    const int j = __builtin_amdgcn_readfirstlane(jv);
    const float* a = A.affine + (size_t)j * 12;
    float R[12];
#pragma unroll
    for (int k = 0; k < 12; ++k)
      R[k] = a[k];
    double acc = 0.0;
    // one evaluation up to the column: byte index into distance_ratios of the voxel, or -1 off the map
    auto locate = [&](int q, unsigned& col, unsigned& ck_out) -> bool {
      const float px = s_pts[0][q], py = s_pts[1][q], pz = s_pts[2][q];
      const float wx = ((R[0] * px + R[1] * py) + R[2] * pz) + R[9];
      const float wy = ((R[3] * px + R[4] * py) + R[5] * pz) + R[10];
      const float wz = ((R[6] * px + R[7] * py) + R[8] * pz) + R[11];
      const int ci = (EXACT_RINV ? voxel_of_exact(wx, M.inv_resolution) : voxel_of(wx, M.resolution, M.inv_resolution)) - M.min_c[0];
      const int cj = (EXACT_RINV ? voxel_of_exact(wy, M.inv_resolution) : voxel_of(wy, M.resolution, M.inv_resolution)) - M.min_c[1];
      const int ck = (EXACT_RINV ? voxel_of_exact(wz, M.inv_resolution) : voxel_of(wz, M.resolution, M.inv_resolution)) - M.min_c[2];
      const bool ok = (unsigned)ci <= (unsigned)span_x && (unsigned)cj <= (unsigned)span_y && (unsigned)ck <= (unsigned)span_z;
      col = ok ? (unsigned)cj * (unsigned)M.width + (unsigned)ci : 0u;
      ck_out = ok ? (unsigned)ck : 0u;
      return ok;
    };
    constexpr int U = 4;  // independent two-level gathers in flight per lane
    int q = lane;
    for (; q + 64 * (U - 1) < np; q += 64 * U)
    {
      unsigned col[U], ck[U], start[U], lvl[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        ok[u] = locate(q + 64 * u, col[u], ck[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        start[u] = M.pose_indices[col[u]];
#pragma unroll
      for (int u = 0; u < U; ++u)
        lvl[u] = M.distance_ratios[(size_t)start[u] + ck[u]];
#pragma unroll
      for (int u = 0; u < U; ++u)
        acc += s_table[ok[u] ? lvl[u] : 256u];
    }
    for (; q < np; q += 64)
    {
      unsigned col, ck;
      const bool ok = locate(q, col, ck);
      const unsigned start = M.pose_indices[col];
      const unsigned lvl = M.distance_ratios[(size_t)start + ck];
      acc += s_table[ok ? lvl : 256u];
    }
    const double tot = wave_sum(acc);
    if (lane == 0)
      A.partials[(size_t)chunk * A.n + j] = tot;
  }
#ifdef BPF_PHASE_TIMING
  if (lane == 0)
  {
    const int _w = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * 4 + wave;
    if (_w < 8192)
    {
      g_cloud_span[_w][0] = (unsigned long long)_w0;
      g_cloud_span[_w][1] = (unsigned long long)wall_clock64();
    }
  }
#endif
}

struct CloudFinishArgs
{
  ParticlesDev p;
  int n;
  const double* partials;
  int n_chunks;
  int n_points;
  Map3dDev map;
  CloudModelDev model;
};

__global__ void k_cloud_finish(const CloudFinishArgs A)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n)
    return;
  double sum = 0.0;
  for (int c = 0; c < A.n_chunks; ++c)
    sum += A.partials[(size_t)c * A.n + i];
  double pfac;
  if (A.model.model == 0)
    pfac = 1.0 + sum;  // point_cloud_scanner.cpp:142,160
  else
  {
    double v = sum / A.n_points;  // :197, no zero-count guard in the reference either
    v = v * A.model.g.input_scale + A.model.g.input_shift;
    v = A.model.g.a * exp(-1.0 * A.model.g.b * exp(-1.0 * A.model.g.c * v));
    pfac = v + A.model.g.output_shift;
  }
  double w = A.p.w[i] * pfac;
  // recalcWeight (:205-229): off-map factor on the robot's (x, y) cell; true division, once per particle
  const double fx = floor(A.p.x[i] / A.map.resolution + 0.5), fy = floor(A.p.y[i] / A.map.resolution + 0.5);
  const int ci = (fx == fx) ? (int)fx : 0x7fffffff, cj = (fy == fy) ? (int)fy : 0x7fffffff;
  const bool valid = ci <= A.map.max_c[0] && ci >= A.map.min_c[0] && cj <= A.map.max_c[1] && cj >= A.map.min_c[1];
  if (!valid)
    w *= A.model.off_map_factor;
  A.p.w[i] = w;
}

}  // namespace bpf
