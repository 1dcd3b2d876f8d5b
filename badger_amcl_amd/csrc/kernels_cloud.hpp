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

// The reference's two-level LUT (pose_indices -> column start in distance_ratios) and, when it fits, the same values
// as a DENSE volume laid out for the gathers of the scoring kernel: one plane per z cell, inside a plane 8 x 8-cell
// tiles (64 bytes) with the cells of a tile x-major, cell (i, j, k) of the map at grid position (i + 1, j + 1), so that
// the byte offset of grid position (x, y) is  8 x + y + (8 ntx - 1) (y & ~7)  (tile row  (y >> 3) ntx 64, tile
// (x >> 3) 64, inside (x & 7) 8 + (y & 7)).  The 64 lanes of a gather are 64 neighbouring points of the cloud, a
// short stretch of a wall at one height: a handful of tiles instead of one line per column.
struct Map3dDev
{
  const uint32_t* pose_indices;
  const uint8_t* distance_ratios;
  const uint8_t* dense;        // nullptr: two-level only
  unsigned dense_k;            // 8 ntx - 1
  unsigned dense_plane;        // bytes per z plane (< 2^24)
  int border_code;             // >= 0: a distance ratio no cell of the LUT holds; the dense volume's border cells (grid
                               // positions 0 and size + 1 in x and y) and its padding hold it, and the scoring kernel
                               // reads the off-map term under it (k_cloud_score, BORDER); -1: fewer than two free ratios
  int zero_code;               // with border_code: a second free ratio; one more plane behind the volume's last (plane
                               // index nz + 1) holds it everywhere, the kernel reads 0.0 under it: where the padding
                               // lanes of a chunk's last group of points gather from
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

constexpr int kCloudChunk = 2048;  // points per LDS chunk (24 KB as float SoA -> 6 blocks per CU); a multiple of 256

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

// The scoring kernel's forms: inputs are finite by construction (non-finite points and poses are replaced at staging
// by 1e30, which ends off the map exactly as the reference's (int)floor(NaN) = INT_MIN does), so the NaN test is
// gone, and the lower bound of the map is folded in.  Exact-reciprocal form: v * rinv has <= 29 significant bits
// and |0.5 - min| < 2^21, so fma(v, rinv, 0.5 - min) is exact (or, for |v| < 2^-22, stays strictly between the
// same two integers) and its floor is the reference's cell minus min.  v_cvt_i32_f64 saturates for huge values.
__device__ __forceinline__ int voxel_rel_exact(float v, double rinv, double half_minus_min)
{
  return (int)floor(fma((double)v, rinv, half_minus_min));
}

__device__ __forceinline__ int voxel_rel(float v, double res, double rinv, int min_c)
{
#pragma clang fp contract(off)
  const double d = (double)v;
  double q = d * rinv;
  const double rem = fma(-q, res, d);
  q = fma(rem, rinv, q);
  return (int)floor(q + 0.5) - min_c;
}

// The scoring kernel's own form, one step further: cell - min + 1 by TRUNCATION.  With the half folded in and one more
// added, every on-map coordinate maps to a value >= 1, where truncation and floor agree; everything below the map
// lands on 0 or a negative number (a huge unsigned), so "on the map" is 1 <= c1 <= span + 1 and the floor is not
// needed (one instruction per coordinate less).  half_minus_min_plus1 = 1.5 - min.
__device__ __forceinline__ unsigned voxel1_exact(float v, double rinv, double half_minus_min_plus1)
{
  return (unsigned)(int)fma((double)v, rinv, half_minus_min_plus1);
}

__device__ __forceinline__ unsigned voxel1(float v, double res, double rinv, int min_c)
{
  return (unsigned)(voxel_rel(v, res, rinv, min_c) + 1);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

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
  float planar_tz;       // PLANAR kernels: the z translation every particle shares, (float)tf_z
};

#ifdef BPF_PHASE_TIMING
__device__ unsigned long long g_cloud_span[8192][2];  // diagnostic builds: per-wave start / end (100 MHz clock)
#endif

// PLANAR: the scanner's mounting has no roll or pitch (tf quaternion x = y = 0).  The particles only add a yaw, so
// for EVERY particle the float matrix has a[2] = a[5] = a[6] = a[7] = 0 and a[8] = 1 exactly and t2 = (float)tf_z:
// a point's z voxel does not depend on the particle (wz = pz + t2: the products with 0 and 1 and the sums with 0 are
// exact) and its x, y are a 2-D rotation (the r2 * pz and r5 * pz terms add an exact zero).  The z voxel, its clamp and
// its range test are then formed once per chunk at staging and kept in LDS in place of pz.
// DENSE: one gather from Map3dDev::dense instead of the two-level pair.
// min(max(c, 1), hi) in ONE instruction (hi >= 1): the compiler keeps v_max_u32 + v_min_u32 because it cannot see
// that the bounds are ordered -- two of the kernel's ~29 vector instructions per evaluation
__device__ __forceinline__ unsigned clamp_1_to(unsigned c, unsigned hi)
{
  unsigned r;
  asm("v_med3_u32 %0, %1, 1, %2" : "=v"(r) : "v"(c), "s"(hi));
  return r;
}

// BORDER (with PLANAR and DENSE): an x or y cell off the map is clamped onto the volume's border (grid position 0 or
// size + 1), whose cells hold Map3dDev::border_code, and the table's entry under that code is the off-map term for the
// launch -- so the evaluation needs neither the two comparisons against the clamped cells nor the select of `bad`
// (3 of its ~24 vector instructions, in a kernel whose duration is its instruction count).  And the padding lanes behind
// a chunk's last point gather from one more plane that holds Map3dDev::zero_code, under which the table has 0.0: no
// `bad` offset is left at all, and the maximum in front of the table read goes too.  Same terms, same order.
template <bool EXACT_RINV, bool PLANAR, bool DENSE, bool BORDER = false>
__global__ __launch_bounds__(256, 6) void k_cloud_score(const CloudScoreArgs A)
{
#pragma clang fp contract(off)
  static_assert(!BORDER || (PLANAR && DENSE), "BORDER is a form of the planar dense kernel");
#ifdef BPF_PHASE_TIMING
  const long long _w0 = wall_clock64();
#endif
  __shared__ float s_pts[3][kCloudChunk];
  // [0, 256): the term per distance ratio; [256]: the off-map term; [257]: zero (padding lanes of the last group).
  // An evaluation's byte offset is max(ratio << 3, bad) with bad = 0 (on the map), 2048 or 2056.  (The off-map and zero
  // entries 256 times each, so that the offset is one v_lshl_or_b32: the 4 KB more LDS cost a block per CU and the
  // instruction saved bought nothing: 8.61 against 8.59 ms.)
  __shared__ double s_table[258];
  const int chunk = blockIdx.x;
  const int p0 = chunk * kCloudChunk;
  const int np = min(kCloudChunk, A.n_points - p0);
  const int np_pad = (np + 255) & ~255;  // whole groups of 256 points; the padding is finite and contributes zero
  const Map3dDev& M = A.map;
  const unsigned span_x = (unsigned)(M.max_c[0] - M.min_c[0]), span_y = (unsigned)(M.max_c[1] - M.min_c[1]),
                 span_z = (unsigned)(M.max_c[2] - M.min_c[2]);
  // cells are carried as cell - min + 1 (voxel1): on the map <=> 1 <= c1 <= span + 1
  const double hm0 = 1.5 - (double)M.min_c[0], hm1 = 1.5 - (double)M.min_c[1], hm2 = 1.5 - (double)M.min_c[2];
  for (int i = threadIdx.x; i < np_pad; i += 256)
  {
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
      const float v = (i < np) ? A.points[(size_t)k * A.n_points + p0 + i] : 0.f;
      const float f = (fabsf(v) <= 3.0e38f) ? v : 1.0e30f;  // NaN / inf: a finite coordinate that is off every map
      if (PLANAR && k == 2)
      {
        const float wz = f + A.planar_tz;
        const unsigned c1 = EXACT_RINV ? voxel1_exact(wz, M.inv_resolution, hm2)
                                       : voxel1(wz, M.resolution, M.inv_resolution, M.min_c[2]);
        const unsigned cc = min(max(c1, 1u), span_z + 1u);
        if (DENSE)
        {
          // the z plane's byte offset; a point whose z is off the map is off the map for every particle: its x is
          // replaced by a coordinate that fails the x test, so the loop needs no flag for it
          // (BORDER: the padding behind the chunk's last point gathers from the plane of zero_code and adds 0.0)
          s_pts[2][i] = __uint_as_float((BORDER && i >= np) ? (span_z + 2u) * M.dense_plane : cc * M.dense_plane);
          if (cc != c1)
            s_pts[0][i] = 1.0e30f;
        }
        else  // low half: the clamped z cell (+ 1); high half: 2048 (the table's off-map entry) when z is off the map
          s_pts[2][i] = __uint_as_float(cc | ((cc != c1) ? (2048u << 16) : 0u));
      }
      else
        s_pts[k][i] = f;
    }
  }
  for (int i = threadIdx.x; i < 258; i += 256)
    s_table[i] = (BORDER && i == A.map.border_code) ? A.table[256]
                 : ((BORDER && i == A.map.zero_code) ? 0.0 : ((i < 257) ? A.table[i] : 0.0));
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned width4 = (unsigned)M.width * 4u;
  const char* table_b = reinterpret_cast<const char*>(s_table);
  // the "+ 1" of the cells is taken back in the base addresses
  const char* pose_b = reinterpret_cast<const char*>(M.pose_indices) - (size_t)width4 - 4;
  const uint8_t* ratio_b = M.distance_ratios - 1;
  const uint8_t* dense_b = M.dense - M.dense_plane;  // plane index = z cell + 1
  const int n_groups = np_pad >> 8;  // wave-uniform: every lane walks the same groups

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
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 12; ++k)
    {
      R[k] = a[k];
      finite = finite && (fabsf(R[k]) <= 3.0e38f);
    }
    if (!finite)
    {
      // a NaN / inf pose puts every point off the map in the reference; same here through finite numbers
#pragma unroll
      for (int k = 0; k < 12; ++k)
        R[k] = 0.f;
      R[9] = 1.0e30f;
    }
    double acc = 0.0;
    // Cell of one transformed point relative to the map's lower corner -> byte offset of its column's entry in
    // pose_indices (24-bit multiply-add; bounds checked in bpf_map3d_set), level index, and `bad` (0 on the map,
    // 2048 = the table's off-map entry otherwise).  Off-map points still gather, from the clamped cell: selects on the addresses turn into
    // exec-mask branches, five scalar instructions per evaluation in a kernel that is bound by its instruction
    // count (measured: SQ_ACTIVE_INST_ANY x 4 cycles / SIMD = the kernel's duration).
    auto cell = [&](float wx, float wy, float wz, unsigned& col, unsigned& ck) -> unsigned {
      const unsigned ci = EXACT_RINV ? voxel1_exact(wx, M.inv_resolution, hm0)
                                     : voxel1(wx, M.resolution, M.inv_resolution, M.min_c[0]);
      const unsigned cj = EXACT_RINV ? voxel1_exact(wy, M.inv_resolution, hm1)
                                     : voxel1(wy, M.resolution, M.inv_resolution, M.min_c[1]);
      const unsigned cz = EXACT_RINV ? voxel1_exact(wz, M.inv_resolution, hm2)
                                     : voxel1(wz, M.resolution, M.inv_resolution, M.min_c[2]);
      const unsigned xi = clamp_1_to(ci, span_x + 1u), xj = clamp_1_to(cj, span_y + 1u);
      ck = clamp_1_to(cz, span_z + 1u);
      const bool ok = xi == ci && xj == cj && ck == cz;
      if (DENSE)
      {
        unsigned c2;  // (the multiply-add stated: see cell_xy)
        const unsigned t = (xi << 3) + xj, xh = xj & ~7u;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c2) : "v"(xh), "s"(M.dense_k), "v"(t));
        col = __umul24(ck, M.dense_plane) + c2;
      }
      else
        col = __umul24(xj, width4) + (xi << 2);
      return ok ? 0u : 2048u;
    };
    // PLANAR: x and y only; the z cell and its off-map flag come packed from LDS
    auto cell_xy = [&](float wx, float wy, unsigned& col) -> unsigned {
      const unsigned ci = EXACT_RINV ? voxel1_exact(wx, M.inv_resolution, hm0)
                                     : voxel1(wx, M.resolution, M.inv_resolution, M.min_c[0]);
      const unsigned cj = EXACT_RINV ? voxel1_exact(wy, M.inv_resolution, hm1)
                                     : voxel1(wy, M.resolution, M.inv_resolution, M.min_c[1]);
      if (BORDER)
      {
        // everything below the map is 0 or "negative" (a huge unsigned), everything above is > span + 1: one unsigned
        // minimum per axis lands all of it on a border cell
        const unsigned bi = min(ci, span_x + 2u), bj = min(cj, span_y + 2u);
        // 8 bi + bj + dense_k (bj & ~7) as shift-add, and, multiply-add: stated, because the compiler splits the
        // multiply-add into v_mul_u32_u24 + v_add3_u32 with the shift on its own (one operation more per evaluation)
        const unsigned t = (bi << 3) + bj, bh = bj & ~7u;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(col) : "v"(bh), "s"(M.dense_k), "v"(t));
        return 0u;
      }
      const unsigned xi = clamp_1_to(ci, span_x + 1u), xj = clamp_1_to(cj, span_y + 1u);
      const bool ok = xi == ci && xj == cj;
      if (DENSE)
      {
        const unsigned t = (xi << 3) + xj, xh = xj & ~7u;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(col) : "v"(xh), "s"(M.dense_k), "v"(t));
      }
      else
        col = __umul24(xj, width4) + (xi << 2);
      return ok ? 0u : 2048u;
    };
    // Two points at a time: the float transform R*p + t with separately rounded products and sums (no contraction),
    // as packed f32 operations -- lane halves = the two points (q and q + 64 sit 64 floats apart in LDS, which is
    // what ds_read2st64 fetches as a register pair).
    const f32x2 r0 = { R[0], R[0] }, r1 = { R[1], R[1] }, r2 = { R[2], R[2] }, r3 = { R[3], R[3] },
                r4 = { R[4], R[4] }, r5 = { R[5], R[5] }, r6 = { R[6], R[6] }, r7 = { R[7], R[7] },
                r8 = { R[8], R[8] }, t0 = { R[9], R[9] }, t1 = { R[10], R[10] }, t2 = { R[11], R[11] };
    constexpr int U = 4;  // points per lane and group: independent two-level gathers in flight (two pairs)
    // cells of the U points of group g and the first-level gathers for them
    auto stage1 = [&](int g, unsigned (&start)[U], unsigned (&ck)[U], unsigned (&bad)[U]) {
      const int q0 = (g << 8) + lane;
      unsigned col[U];
#pragma unroll
      for (int h = 0; h < U / 2; ++h)
      {
        const int qa = q0 + 128 * h, qb = qa + 64;
        const f32x2 px = { s_pts[0][qa], s_pts[0][qb] }, py = { s_pts[1][qa], s_pts[1][qb] },
                    pz = { s_pts[2][qa], s_pts[2][qb] };
        if (PLANAR)
        {
          const f32x2 wx = (r0 * px + r1 * py) + t0;
          const f32x2 wy = (r3 * px + r4 * py) + t1;
          const unsigned za = __float_as_uint(pz.x), zb = __float_as_uint(pz.y);
          if (DENSE)
          {
            bad[2 * h] = cell_xy(wx.x, wy.x, col[2 * h]);
            bad[2 * h + 1] = cell_xy(wx.y, wy.y, col[2 * h + 1]);
            col[2 * h] += za;
            col[2 * h + 1] += zb;
            ck[2 * h] = ck[2 * h + 1] = 0u;
          }
          else
          {
            bad[2 * h] = max(cell_xy(wx.x, wy.x, col[2 * h]), za >> 16);
            bad[2 * h + 1] = max(cell_xy(wx.y, wy.y, col[2 * h + 1]), zb >> 16);
            ck[2 * h] = za & 0xFFFFu;
            ck[2 * h + 1] = zb & 0xFFFFu;
          }
        }
        else
        {
          const f32x2 wx = ((r0 * px + r1 * py) + r2 * pz) + t0;
          const f32x2 wy = ((r3 * px + r4 * py) + r5 * pz) + t1;
          const f32x2 wz = ((r6 * px + r7 * py) + r8 * pz) + t2;
          bad[2 * h] = cell(wx.x, wy.x, wz.x, col[2 * h], ck[2 * h]);
          bad[2 * h + 1] = cell(wx.y, wy.y, wz.y, col[2 * h + 1], ck[2 * h + 1]);
        }
      }
      if (!BORDER && g == n_groups - 1 && np != np_pad)
      {
        // the padding lanes of the chunk's last group evaluate a harmless point and add the table's zero
#pragma unroll
        for (int u = 0; u < U; ++u)
          bad[u] = (q0 + 64 * u < np) ? bad[u] : 2056u;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        start[u] = DENSE ? (unsigned)dense_b[col[u]] : *reinterpret_cast<const uint32_t*>(pose_b + col[u]);
    };
    // Software pipeline over the groups: while the second-level gathers of group g are in flight, the cells and
    // first-level gathers of group g + 1 are formed and issued.
    // Two register sets in turn (a single set had to be copied aside before the next group overwrote it: eight moves
    // per group of four evaluations).
    unsigned start_a[U], ck_a[U], bad_a[U], start_b[U], ck_b[U], bad_b[U];
    auto consume = [&](const unsigned (&st)[U], const unsigned (&ck)[U], const unsigned (&bd)[U]) {
      unsigned lvl[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        lvl[u] = DENSE ? st[u] : (unsigned)ratio_b[st[u] + ck[u]];
#pragma unroll
      for (int u = 0; u < U; ++u)
        acc += *reinterpret_cast<const double*>(table_b + (BORDER ? lvl[u] << 3 : max(lvl[u] << 3, bd[u])));
    };
    stage1(0, start_a, ck_a, bad_a);
    for (int g = 0; g < n_groups; g += 2)
    {
      if (g + 1 < n_groups)
        stage1(g + 1, start_b, ck_b, bad_b);
      consume(start_a, ck_a, bad_a);
      if (g + 1 >= n_groups)
        break;
      if (g + 2 < n_groups)
        stage1(g + 2, start_a, ck_a, bad_a);
      consume(start_b, ck_b, bad_b);
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

// the two-level LUT re-laid as the dense tiled volume (Map3dDev); one thread per map cell
__global__ void k_dense3d_build(const uint32_t* __restrict__ pose_indices, const uint8_t* __restrict__ ratios, int w,
                                int h, int nz, unsigned dense_k, unsigned plane, uint8_t* __restrict__ dense)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)w * h * nz;
  if (t >= total)
    return;
  const int k = (int)(t % nz);
  const size_t col = t / nz;
  const int i = (int)(col % w), j = (int)(col / w);
  const unsigned x = (unsigned)i + 1u, y = (unsigned)j + 1u;
  dense[(size_t)k * plane + (x << 3) + y + (size_t)dense_k * (y & ~7u)] = ratios[(size_t)pose_indices[col] + k];
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
