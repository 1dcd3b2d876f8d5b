// Sensor scoring kernels (gfx950, wave64).
//
// Work mapping, both kernels: one wavefront scores one particle at a time with its 64
// lanes striding over the beams -- consecutive beams of one particle end on neighbouring
// map cells, so a wave's gathers fall into a handful of 8x8-cell tiles whatever the wall
// orientation.  A wave takes 16 particles per trip: lanes 0..15 do the per-particle
// trigonometry once, the results are broadcast through SGPRs (v_readlane), and the
// per-particle sums come back to lanes 0..15 for the epilogue (weight update +
// recalcWeight), which stores 16 consecutive weights.
#pragma once
#include "device_types.hpp"

namespace bpf
{

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off, 64);
  return v;
}

// broadcast a double from a wave-uniform lane index through scalar registers
__device__ __forceinline__ double lane_bcast(double v, int src_lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// angles::normalize_angle (third-party ROS `angles` header, Noetic form): fmod(a+pi, 2pi), <=0 ? +pi : -pi
__device__ __forceinline__ double normalize_angle_dev(double a)
{
  const double kPi = 3.14159265358979323846;
  const double r = fmod(a + kPi, 2.0 * kPi);
  return (r <= 0.0) ? r + kPi : r - kPi;
}

// OccupancyMap::convertWorldToMap for one axis (occupancy_map.cpp:96-97), exact division
__device__ __forceinline__ int world_to_cell(double v, double origin, double res, int half)
{
  const double f = floor((v - origin) / res + 0.5);
  // v_cvt_i32_f64 saturates and maps NaN to 0; send NaN off the map like x86's INT_MIN does
  return (f == f) ? (int)f + half : -1;
}

// The same with the division as a corrected reciprocal multiply (q = d*r; q += fma(-q, res, d)*r): the correctly
// rounded quotient but for vanishingly rare double roundings, at a fifth of the instructions of an IEEE division.
// Used where the conversion runs once per ray (beam model).
__device__ __forceinline__ int world_to_cell_rcp(double v, double origin, double res, double rinv, int half)
{
  const double d = v - origin;
  double q = d * rinv;
  q = fma(fma(-q, res, d), rinv, q);
  const double f = floor(q + 0.5);
  return (f == f) ? (int)f + half : -1;
}

constexpr int kLutPad = 1;  // border cells (holding the off-map level) on each side of the LUT image

// Clamp a padded cell coordinate into [0, size + 1]: negative values wrap to huge unsigned numbers
// and land on the far border, which holds the same off-map level as the near one.
__device__ __forceinline__ unsigned clamp_cell(int c, int size)
{
  return min((unsigned)c, (unsigned)(size + kLutPad));
}

// byte offset of padded cell (u, v) in lut_tiles (see MapDev): tile (v>>3, u>>3), cell (u&7)*8 + (v&7) inside it,
//   128*((v>>3)*ltx + (u>>3)) + 16*(u&7) + 2*(v&7)  =  16*u + 2*v + (16*ltx - 2)*(v & ~7)
// -- four VALU operations (shift, shift-add, and, 24-bit multiply-add), one masked term only.
__device__ __forceinline__ unsigned lut_byte_offset(const MapDev& m, unsigned u, unsigned v)
{
  const unsigned t = (u << 4) + (v << 1);
  const unsigned vh = v & ~7u, k = (unsigned)(16 * m.ltx - 2);
  // the compiler splits `vh * k + t` into v_mul_u32_u24 + v_add3_u32 with the two shifts (5 operations);
  // stating the multiply-add keeps it at shift, shift-add, and, multiply-add
  unsigned off;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(off) : "v"(vh), "s"(k), "v"(t));
  return off;
}

// the stored 16-bit value is (level index * 8) = byte offset of the level's term in the term table
__device__ __forceinline__ unsigned lut_level8(const MapDev& m, unsigned u, unsigned v)
{
  return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(m.lut_tiles) + lut_byte_offset(m, u, v));
}

// PlanarScanner::recalcWeight for one particle (planar_scanner.cpp:642-682)
__device__ __forceinline__ double recalc_factor(const MapDev& m, double px, double py, double off_map_factor,
                                                double non_free_factor, double non_free_radius)
{
  const int ci = world_to_cell(px, m.origin_x, m.resolution, m.half_x);
  const int cj = world_to_cell(py, m.origin_y, m.resolution, m.half_y);
  if (!((unsigned)ci < (unsigned)m.size_x && (unsigned)cj < (unsigned)m.size_y))
    return off_map_factor;
  if (m.cells8[ci + (size_t)cj * m.size_x] != -1)
    return non_free_factor;
  const unsigned idx = lut_level8(m, (unsigned)(ci + kLutPad), (unsigned)(cj + kLutPad)) >> 3;
  const double d = (double)m.levels[idx];
  if (d < non_free_radius)
  {
    const double frac = d / non_free_radius;
    double f = non_free_factor;
    f += frac * (1.0 - non_free_factor);
    return f;
  }
  return 1.0;
}

// scanner pose of a particle: PlanarScanner::coordAdd (planar_scanner.cpp:693-701)
struct ScannerPose
{
  double x, y, c, s, th;
};

__device__ __forceinline__ ScannerPose scanner_pose(double px, double py, double pth, double ax, double ay,
                                                    double ath)
{
  ScannerPose o;
  double sn, cs;
  sincos(pth, &sn, &cs);
  o.x = px + ax * cs - ay * sn;
  o.y = py + ax * sn + ay * cs;
  o.th = normalize_angle_dev(pth + ath);
  sincos(o.th, &o.s, &o.c);
  return o;
}

// ---------------------------------------------------------------------------------------
// Likelihood-field family: calcLikelihoodFieldModel (planar_scanner.cpp:236-323),
// calcLikelihoodFieldModelGompertz (:552-640) and the per-beam part of
// calcLikelihoodFieldModelProb (:325-533).
//
// Per evaluation the reference computes cos/sin(theta + bearing), the end point, its cell,
// the LUT distance z and a term f(z).  Here:
//   * angle addition: the particle's (cos, sin) are computed once, the beam arrives as
//     B = r*(cos b, sin b)/res, so the end point in cell units is 2 FMAs per axis;
//   * z only takes the LUT's few hundred distinct float values ("levels"), so f(z) is a
//     table indexed by the 16-bit level id of the cell, built on the host with the same
//     libm expression as the reference (no transcendental in the loop, identical terms).
// ---------------------------------------------------------------------------------------
// Per-particle quantities of the likelihood-field family, computed once per update by k_field_prep:
// scanner position in padded-cell units with the 0.5 rounding offset, size/2 and the border folded
// in (so the padded cell index of an end point is the truncation of Q + R(theta)*B), and cos / sin
// of the scanner heading.  A pose with a NaN / infinite / absurd component ends every beam off the
// map in the reference ((int)NaN is INT_MIN on x86); it is stored as a far-away point with a zero
// rotation so that the same happens here without per-beam tests.
__device__ __forceinline__ double4 field_prep_of(const MapDev& M, double px, double py, double pth, double ax,
                                                 double ay, double ath, bool* valid)
{
  ScannerPose sp = scanner_pose(px, py, pth, ax, ay, ath);
  double Qx = ((sp.x - M.origin_x) / M.resolution + 0.5) + (double)(M.half_x + kLutPad);
  double Qy = ((sp.y - M.origin_y) / M.resolution + 0.5) + (double)(M.half_y + kLutPad);
  *valid = fabs(sp.c) <= 1.0 && fabs(sp.s) <= 1.0 && fabs(Qx) < 1073741824.0 && fabs(Qy) < 1073741824.0;
  if (!*valid)
  {
    sp.c = 0.0;
    sp.s = 0.0;
    Qx = -1048576.0;
    Qy = -1048576.0;
  }
  return make_double4(Qx, Qy, sp.c, sp.s);
}

// One likelihood-field evaluation up to the cell: byte offset of the end point's LUT entry.
// |B| < 2^28 (checked when the beam table is staged) and |Q| < 2^30 keep the sum inside int range,
// so the conversion never saturates onto the map.
__device__ __forceinline__ unsigned field_cell(const MapDev& M, double c, double s, double qx, double qy,
                                               const double2 B)
{
  const double vx = fma(c, B.x, fma(-s, B.y, qx));
  const double vy = fma(s, B.x, fma(c, B.y, qy));
  return lut_byte_offset(M, clamp_cell((int)vx, M.size_x), clamp_cell((int)vy, M.size_y));
}

#ifndef BPF_FIELD_UNROLL
#define BPF_FIELD_UNROLL 8
#endif
constexpr int kFieldUnroll = BPF_FIELD_UNROLL;

// LDS layout of k_score_field: [term table (table_len doubles)] [beams (n_beams double2)]
// COUNT_ONLY: pass 1 of the prob model's beam skipping -- only the per-beam agreement counts.
#ifndef BPF_FIELD_WAVES
#define BPF_FIELD_WAVES 4
#endif
#ifdef BPF_PHASE_TIMING
// diagnostic builds only (tools/phase_timing.py): per-wave, per-phase core-clock totals (s_memtime) and the wave's
// start / end on the constant 100 MHz clock (s_memrealtime); one private row per wave, no atomics
constexpr int kPhaseWaves = 8192;
__device__ unsigned long long g_phase_cycles[kPhaseWaves][8];
#define PHASE_MARK(idx)                            \
  do                                               \
  {                                                \
    const long long _now = clock64();              \
    _ph[idx] += (unsigned long long)(_now - _t);   \
    _t = _now;                                     \
  } while (0)
#else
#define PHASE_MARK(idx)
#endif

// HOST_OUT: the host-buffer seam (abi_planar.inl) -- the new weights also go to a pinned array on the host (A.w_host,
// dense 8-byte stores: 0.8 MB per 100 k particles leave over PCIe while the kernel is still working, no download
// behind it).  k_seam_done, the launch behind it, folds the block partials and tells the host.  (A ticket per
// block for "the last block tells the host" was tried: 1 000 same-address atomics at the end of a one-round kernel
// serialise -- a 50 k-particle launch took 75 us instead of 40 -- and a system-scope release per block costs an L2
// write-back each.)
//
// HOST_MODE 3 has nothing to do with the host: TILE-SORTED scoring of a spread cloud.  With the particles all over the
// map every XCD's 4 MB L2 sees the whole LUT (8.3 MB for a 2000 x 2000 map): 747 MB of L2 fills per launch against
// 436 MB algorithmic, 6.3 TB/s, and the kernel waits for LUT lines (120 us where the converged cloud takes 73).  The
// prep launch therefore also bins the particles by map tile (k_field_prep_tile, k_tile_scatter: a
// counting sort into A.perm, prep values written in that order), this kernel walks slots in tile order and the
// graded partition hands each XCD -- blocks b and b + 8 share one -- a CONTIGUOUS eighth of the slots (A.xcd_local),
// i.e. an eighth of the map: its L2 then holds what its waves gather from.  85 us.  The order of the slots changes
// nothing in a particle's own arithmetic; the total comes from a fixed-shape sum over the weights in index order
// (no block partials in this mode), so the result does not depend on the order inside a tile either.
//
// HOST_MODE 2 (a REGISTERED buffer): no copy at all -- every wave reads the caller's 32-byte records from host memory
// itself when it reaches them (A.rec, zero-copy over PCIe), forms (Qx, Qy, cos, sin) with the prep launch's own
// function, and writes the records back WHOLE with the new weight (16 neighbouring records = 512 contiguous bytes per
// trip: full-line posted writes that leave while the kernel works; an 8-byte store per record at the records' stride
// is one PCIe packet each and takes 105 us per 100 k).  The transfer and the scoring overlap as far as one resident
// round of static shares lets them (a wave whose records arrive late still has its whole share to do): 160 us for
// 100 k x 1081 -- 127 us when only the weights are stored (to a pinned array), the whole records' way back costs the
// rest, and saves the calling thread the 40 us it takes to write 100 k weights into the records itself.
template <bool COUNT_ONLY, bool TABLE_IN_LDS, int HOST_MODE = 0>
__global__ __launch_bounds__(256, BPF_FIELD_WAVES) void k_score_field(const FieldScoreArgs A)
{
  constexpr bool HOST_OUT = HOST_MODE == 1;
  constexpr bool HOST_REC = HOST_MODE == 2;
  constexpr bool TILE_ORDER = HOST_MODE == 3;  // slot j of the launch is particle A.perm[j] (tile-sorted scoring)
#ifdef BPF_PHASE_TIMING
  unsigned long long _ph[6] = { 0, 0, 0, 0, 0, 0 };
  long long _t = clock64();
  const long long _w0 = wall_clock64();
#endif
  if (A.skip_if_set != nullptr && *A.skip_if_set != 0)
    return;  // kernels_window.hpp handles this update
  extern __shared__ __align__(16) unsigned char smem[];
  const int table_lds_len = (TABLE_IN_LDS && !COUNT_ONLY) ? A.table_len : 0;
  double* s_table = reinterpret_cast<double*>(smem);
  double2* s_beams = reinterpret_cast<double2*>(smem + (((size_t)table_lds_len * sizeof(double) + 15) & ~(size_t)15));
  // HOST_MODE 2: 4 waves x 16 records behind the beams (the launch asks for 2 KB more)
  double4* s_rec = reinterpret_cast<double4*>(reinterpret_cast<unsigned char*>(s_beams) +
                                              (((size_t)A.n_beams * sizeof(double2) + 31) & ~(size_t)31));
  (void)s_rec;

  const int tid = threadIdx.x;
  // staging: all of a thread's loads are issued before the first LDS store, so their latencies overlap
  // (a plain copy loop waits for every load in turn: ~8 round trips with 1000 blocks asking at once)
  for (int i0 = 0; i0 < A.n_beams; i0 += 4 * 256)
  {
    double2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = A.beams[min(i0 + q * 256 + tid, A.n_beams - 1)];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i0 + q * 256 + tid < A.n_beams)
        s_beams[i0 + q * 256 + tid] = v[q];
  }
  for (int i0 = 0; i0 < table_lds_len; i0 += 4 * 256)
  {
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = A.table[min(i0 + q * 256 + tid, table_lds_len - 1)];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i0 + q * 256 + tid < table_lds_len)
        s_table[i0 + q * 256 + tid] = v[q];
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  PHASE_MARK(0);  // LDS staging + barrier
  const MapDev& M = A.map;
  const char* table_b = reinterpret_cast<const char*>(TABLE_IN_LDS ? s_table : A.table);
  const char* __restrict__ tiles = reinterpret_cast<const char*>(M.lut_tiles);
  const int n_beams = A.n_beams;
  double wsum = 0.0;  // this lane's share of the block's weight total

  // Static partition: the grid is exactly one resident round of blocks and every wave owns a
  // contiguous range of per_wave particles (a grid with a few blocks more than fit would run a
  // second round for them and double the kernel time).  The range is walked 16 particles at a time.
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  int p_begin = min(A.n, wid * A.per_wave), p_end = min(A.n, p_begin + A.per_wave);
  if (A.share_count[0] > 0)
  {
    // Graded partition.  All blocks of the one resident round start together, and the SIMD's issue arbiter
    // favours its oldest wave: with equal shares the waves of the block placed first on a CU finish at 46 us,
    // those of the fourth at 86 us, and the SIMD idles more and more towards the end.  Blocks are placed in
    // blockIdx order, n_cu per round, so the round a block belongs to tells its age rank on its CU and the
    // shares are graded by it (tools/phase_timing.py shows the effect: every wave then ends within 70-80 us).
    const int round = min((int)blockIdx.x / A.blocks_per_round, 7);
    const int q = (int)blockIdx.x - round * A.blocks_per_round;
    if (A.xcd_local)
    {
      // Blocks b and b + 8 share an XCD (and its 4 MB L2): the x-th eighth of the particle range goes to the blocks
      // with b % 8 == x, split over the placement rounds by the same shares.  With the particles ordered by map tile
      // (tile-sorted scoring of a spread cloud) an XCD then only ever sees its own eighth of the map's LUT.
      const int x = q & 7, cu_slot = q >> 3, per_x = A.blocks_per_round >> 3;
      int xcd_total = 0, before = 0;
      for (int r = 0; r < 8; ++r)
      {
        const int c = A.share_count[r] * per_x * 4;
        xcd_total += c;
        if (r < round)
          before += c;
      }
      const int slot = __builtin_amdgcn_readfirstlane(cu_slot * 4 + wave);
      p_begin = min(A.n, x * xcd_total + before + slot * A.share_count[round]);
    }
    else
    {
      const int slot = __builtin_amdgcn_readfirstlane(q * 4 + wave);
      p_begin = min(A.n, A.share_base[round] + slot * A.share_count[round]);
    }
    p_end = min(A.n, p_begin + A.share_count[round]);
  }
  for (int base = p_begin; base < p_end; base += 16)
  {
    const int cnt = min(16, p_end - base);
    // one coalesced load brings the 16 particles' (Qx, Qy, cos, sin); each is then broadcast to the
    // wave through scalar registers (a scalar load per particle would expose its latency 16 times)
    double4 ql;
    if (HOST_REC)
    {
      // lane k (and its copies k + 16, 32, 48) reads record k of the trip; lanes < 16 park it in the wave's LDS stash
      // for the epilogue instead of holding eight more registers through the trip
      const double4 rec = A.rec[base + min(lane & 15, cnt - 1)];
      bool valid;
      ql = field_prep_of(M, rec.x, rec.y, rec.z, A.sp_x, A.sp_y, A.sp_th, &valid);
      if (lane < 16)
        s_rec[wave * 16 + lane] = rec;
    }
    else
      ql = A.prep[base + min(lane & 15, cnt - 1)];
    double mine = 0.0;
    if (!COUNT_ONLY)
    {
      // Beam-batch outer / particle inner: a lane's kFieldUnroll beams are read from LDS once per
      // 16 particles and stay in registers; every lane keeps one running sum per particle, so the
      // cross-lane reduction happens once per group (transposed, below) instead of once per particle.
      double accs[16];
#pragma unroll
      for (int k = 0; k < 16; ++k)
        accs[k] = 0.0;
      int b = lane;
      for (; b + 64 * (kFieldUnroll - 1) < n_beams; b += 64 * kFieldUnroll)
      {
        double2 B[kFieldUnroll];
#pragma unroll
        for (int u = 0; u < kFieldUnroll; ++u)
          B[u] = s_beams[b + 64 * u];
#pragma unroll
        for (int k = 0; k < 16; ++k)
        {
          if (k < cnt)
          {
            const double qx = lane_bcast(ql.x, k), qy = lane_bcast(ql.y, k);
            const double c = lane_bcast(ql.z, k), s = lane_bcast(ql.w, k);
            unsigned lv[kFieldUnroll];
#pragma unroll
            for (int u = 0; u < kFieldUnroll; ++u)
              lv[u] = *reinterpret_cast<const uint16_t*>(tiles + field_cell(M, c, s, qx, qy, B[u]));
#pragma unroll
            for (int u = 0; u < kFieldUnroll; ++u)
              accs[k] += *reinterpret_cast<const double*>(table_b + lv[u]);
          }
        }
      }
      PHASE_MARK(1);  // group start + full beam batches
      for (; b < n_beams; b += 64)
      {
        const double2 B1 = s_beams[b];
#pragma unroll
        for (int k = 0; k < 16; ++k)
        {
          if (k < cnt)
          {
            const double qx = lane_bcast(ql.x, k), qy = lane_bcast(ql.y, k);
            const double c = lane_bcast(ql.z, k), s = lane_bcast(ql.w, k);
            const unsigned lv = *reinterpret_cast<const uint16_t*>(tiles + field_cell(M, c, s, qx, qy, B1));
            accs[k] += *reinterpret_cast<const double*>(table_b + lv);
          }
        }
      }
      PHASE_MARK(2);  // remainder loop
      // transposed reduction: 16 per-particle partials x 64 lanes -> lane k (k < 16) holds particle k's sum.
      // Step h: lanes exchange the half of their values they do not keep (xor 32, 16, 8, 4 halve the
      // value count 16 -> 1), then two plain butterfly steps finish (xor 2, 1).
      {
        // after the exchanges lane L keeps the value of particle (L >> 2) & 15 ... built up bit by bit
        double v8[8], v4[4], v2[2], v1;
        const bool up32 = (lane & 32) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
          const double keep = up32 ? accs[i + 8] : accs[i];
          const double give = up32 ? accs[i] : accs[i + 8];
          v8[i] = keep + __shfl_xor(give, 32, 64);
        }
        const bool up16 = (lane & 16) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
          const double keep = up16 ? v8[i + 4] : v8[i];
          const double give = up16 ? v8[i] : v8[i + 4];
          v4[i] = keep + __shfl_xor(give, 16, 64);
        }
        const bool up8 = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
        {
          const double keep = up8 ? v4[i + 2] : v4[i];
          const double give = up8 ? v4[i] : v4[i + 2];
          v2[i] = keep + __shfl_xor(give, 8, 64);
        }
        const bool up4 = (lane & 4) != 0;
        {
          const double keep = up4 ? v2[1] : v2[0];
          const double give = up4 ? v2[0] : v2[1];
          v1 = keep + __shfl_xor(give, 4, 64);
        }
        v1 += __shfl_xor(v1, 2, 64);
        v1 += __shfl_xor(v1, 1, 64);
        // lane L now holds the total of particle p(L) = 8*[L&32] + 4*[L&16] + 2*[L&8] + 1*[L&4]
        const int owner = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
        // route it to lane == particle index for the epilogue
        const int src_lane = ((lane >> 3) & 1) * 32 + ((lane >> 2) & 1) * 16 + ((lane >> 1) & 1) * 8 + (lane & 1) * 4;
        mine = __shfl(v1, src_lane, 64);  // lane k (< 16) reads a lane whose owner is k
        (void)owner;
      }
      PHASE_MARK(3);  // transposed reduction
    }
    else
    {
      for (int k = 0; k < cnt; ++k)
      {
        const double qx = lane_bcast(ql.x, k), qy = lane_bcast(ql.y, k);
        const double c = lane_bcast(ql.z, k), s = lane_bcast(ql.w, k);
        for (int b = lane; b < n_beams; b += 64)
        {
          const unsigned lv = *reinterpret_cast<const uint16_t*>(tiles + field_cell(M, c, s, qx, qy, s_beams[b]));
          // skip_level <= K, so the border's off-map level never counts (planar_scanner.cpp:441-451)
          if ((int)(lv >> 3) < A.skip_level)
            atomicAdd(&A.obs_count[b], 1);
        }
      }
    }

    if (!COUNT_ONLY && lane < cnt)
    {
      const double sum = mine + A.extra_term;  // beams that end off the map for every pose
      double p;
      if (A.model == 1)  // likelihood field: p = 1 + sum pz^3
        p = 1.0 + sum;
      else if (A.model == 3)  // Gompertz of the mean pz (planar_scanner.cpp:540-550,624-633)
      {
        if (A.n_valid > 0)
        {
          double v = sum / A.n_valid;
          v = v * A.g.input_scale + A.g.input_shift;
          v = A.g.a * exp(-1.0 * A.g.b * exp(-1.0 * A.g.c * v));
          p = v + A.g.output_shift;
        }
        else
          p = 1.0;
      }
      else  // prob: exp(sum log pz)
        p = exp(sum);
      const int i = TILE_ORDER ? A.perm[base + lane] : base + lane;
      if (HOST_REC)
      {
        double4 rec = s_rec[wave * 16 + lane];
        double w = rec.w * p;
        w *= recalc_factor(M, rec.x, rec.y, A.off_map_factor, A.non_free_factor, A.non_free_radius);
        rec.w = w;
        A.rec[i] = rec;
        wsum += w;
      }
      else
      {
        double w = A.p.w[i] * p;
        w *= recalc_factor(M, A.p.x[i], A.p.y[i], A.off_map_factor, A.non_free_factor, A.non_free_radius);
        A.p.w[i] = w;
        if (HOST_OUT)  // plain stores: 16 neighbouring weights leave as one 128-byte write (a system-scope atomic
          A.w_host[i] = w;  // store per lane is its own PCIe packet: 50 k of them backed up for 25 us behind the launch)
        wsum += w;
      }
    }
    PHASE_MARK(4);  // epilogue
  }
  if (!COUNT_ONLY && A.block_partials != nullptr)
  {
    // fixed shape: lanes -> wave (xor tree), waves 0..3 in order: reproducible for a given grid.
    // The four partials reuse the head of the dynamic LDS block once every wave is done with the
    // tables: a static __shared__ array would push the term table off LDS address 0 and cost one
    // address add per table read in the inner loop.
    const double ws = wave_sum(wsum);
    __syncthreads();
    double* s_part = reinterpret_cast<double*>(smem);
    if (lane == 0)
      s_part[wave] = ws;
    __syncthreads();
    if (tid == 0)
    {
      A.block_partials[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    }
  }
  PHASE_MARK(5);  // block partial
#ifdef BPF_PHASE_TIMING
  if (lane == 0)
  {
    const int _w = blockIdx.x * 4 + wave;
    if (_w < kPhaseWaves)
    {
      for (int q = 0; q < 6; ++q)
        g_phase_cycles[_w][q] = _ph[q];
      g_phase_cycles[_w][6] = (unsigned long long)_w0;
      g_phase_cycles[_w][7] = (unsigned long long)wall_clock64();
    }
  }
#endif
}

// Behind a HOST_OUT scoring launch: the launch's weight total (its block partials folded in block order, fixed shape)
// and the word the host polls, both into pinned host memory.  A launch boundary orders them behind the weights.
__global__ __launch_bounds__(256) void k_seam_done(const double* __restrict__ partials, int n_partials,
                                                  double* total_host, unsigned long long* done_flag,
                                                  unsigned long long done_value)
{
  __shared__ double s_wave[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += 256)
    acc += partials[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0)
    s_wave[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __hip_atomic_store(total_host, (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(done_flag, done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------------------
// Beam model: calcBeamModel (planar_scanner.cpp:168-234) with OccupancyMap::calcRange
// (occupancy_map.cpp:257-364).  The Bresenham walk reads the 1-bit "not free" grid in 8x8
// tiles: a 64-bit word covers up to 8 steps in any direction and is re-fetched only when
// the ray enters another tile.
// ---------------------------------------------------------------------------------------
struct BeamRec
{
  double cb, sb;   // cos / sin of the bearing
  double obs;      // observed range
  double short_t;  // z_short*lambda*exp(-lambda*obs)
  double tail_t;   // z_max (obs == range_max) or z_rand/range_max (obs < range_max) or 0
};

// OccupancyMap::calcRange (occupancy_map.cpp:257-364) without visiting every cell.
//
// The reference walks an integer Bresenham line from the start cell towards the max-range end cell
// and stops at the first cell that is off the map or not FREE.  The cell reached after j steps is a
// closed form of j (major axis advances j, minor axis advances m_j = floor((2*j*dmin + dmaj)/(2*dmaj)),
// which is exactly what the reference's running error term produces), so the walk can jump:
// MapDev::cheb holds, per padded cell and per quadrant, the chessboard distance D to the nearest blocked cell of
// that quadrant (blocked = not FREE or outside the map; capped at 255).  A line that heads into quadrant q stays
// inside quadrant q of every cell it visits and every Bresenham step moves at most one cell per axis, so the next
// D-1 cells of the line are free and the D-th is the next candidate: jump D steps, look again.  (With one
// direction-blind distance per cell a ray running along a wall crawls; per quadrant a ray that is leaving an
// obstacle behind jumps as far as the space ahead allows: 5.8 look-ups per ray instead of 9.3 on the bench map.)  The first blocked cell found this way is the one the reference finds,
// and the returned distance uses the same integer deltas.  `walked` still counts the cells the
// reference would have visited (j_hit + 1), the unit of the kernel's algorithmic bytes.
// INT_DIV: the minor-axis advance by a 32-bit fixed-point reciprocal (one v_mul_hi_u32 per look-up instead of two
// conversions and an fp64 multiply-add); exact while (2 (dmaj + 1) dmin + dmaj) * 2 dmaj < 2^32, i.e. for rays of at
// most kIntDivRayCells cells -- the host picks the variant per launch from range_max / resolution.
constexpr int kIntDivRayCells = 1000;

// UNIFORM_START: (x0, y0) is the same for every lane of the wave (k_score_beam: wave = particle, read by v_readlane), so
// the start cell's offset can be the multiply-add's scalar addend.
template <bool INT_DIV, bool UNIFORM_START = false>
__device__ __forceinline__ double calc_range_skip(const MapDev& M, int x0, int y0, int x1, int y1, double range_max,
                                                  unsigned long long& walked)
{
  if (x0 == x1 && y0 == y1)
    return range_max;  // occupancy_map.cpp:279-280
  // a start outside map + ring is "not valid" at the very first test: distance 0 (:316-334)
  if (!((unsigned)(x0 + 1) <= (unsigned)(M.size_x + 1) && (unsigned)(y0 + 1) <= (unsigned)(M.size_y + 1)))
  {
    ++walked;
    return 0.0;
  }
  const int adx = abs(x1 - x0), ady = abs(y1 - y0);
  const bool steep = ady > adx;
  const int dmaj = steep ? ady : adx, dmin = steep ? adx : ady;
  const int sx = (x0 < x1) ? 1 : -1, sy = (y0 < y1) ? 1 : -1;
  const int maj_dx = steep ? 0 : sx, maj_dy = steep ? sy : 0;
  const int min_dx = steep ? sx : 0, min_dy = steep ? 0 : sy;
  // 1 / (2 dmaj) to a few ulp (hardware estimate + two Newton steps): the quotient below only has to land within
  // 1e-6 of the true one
  const double two_d = 2.0 * (double)dmaj;
  double inv2d = __builtin_amdgcn_rcp(two_d);
  inv2d = fma(fma(-two_d, inv2d, 1.0), inv2d, inv2d);
  inv2d = fma(fma(-two_d, inv2d, 1.0), inv2d, inv2d);
  // INT_DIV: magic = floor(2^32 / D) + 1 for D = 2 dmaj, so that floor(u / D) = mulhi(u, magic) for every u with
  // u * D < 2^32 (the error magic * D - 2^32 is at most D).  floor(2^32 / D) from the reciprocal, then set right by
  // one integer multiply (the estimate is within 1 of the true quotient: 2^32 * inv2d is good to ~1e-6 absolute).
  unsigned magic = 0;
  if (INT_DIV)
  {
    const unsigned D = 2u * (unsigned)dmaj;
    unsigned q = (unsigned)fma(4294967296.0, inv2d, -0.5);       // floor(2^32 / D) or one less / more
    const unsigned rem = 0u - q * D;                             // 2^32 - q D (mod 2^32), in [-(D), 2 D) as a signed value
    q += ((int)rem >= (int)D) ? 1u : 0u;
    q -= ((int)rem < 0) ? 1u : 0u;
    magic = q + 1u;
  }
  const int last = dmaj + 1;  // the reference tests cells j = 0 .. dmaj + 1
  const int stride = M.size_x + 2;
  // The walk itself is kept to a dozen instructions per visited cell (the kernel is bound by its instruction
  // count): the cell after j major and m minor steps sits at base + j * step_major + m * step_minor in the padded
  // chessboard-distance grid, all three formed once per ray, in BYTES (the grid has one 32-bit word per cell);
  // 24-bit multiply-adds (4 |step| <= 4 (size_x + 3) < 2^23, j, m <= range_max / resolution + 1, checked where the
  // map is set).
  const int step_major = 4 * (maj_dy * stride + maj_dx), step_minor = 4 * (min_dy * stride + min_dx);
  const int base = 4 * ((y0 + 1) * stride + (x0 + 1));
  const int two_dmin = 2 * dmin;
  const char* cheb = reinterpret_cast<const char*>(M.cheb);
  const unsigned qshift = ((sx < 0) ? 8u : 0u) + ((sy < 0) ? 16u : 0u);  // byte of the ray's quadrant
  int j = 0, m = 0;
  bool hit;
  for (;;)
  {
    // minor-axis advance after j steps, floor((2 j dmin + dmaj) / (2 dmaj)); fp64 variant: the 1e-6 absorbs the
    // reciprocal's rounding (fractional parts of the true quotient are multiples of 1/(2*dmaj) >= 1e-4)
    if (INT_DIV)
      m = (int)__umulhi((unsigned)(__mul24(j, two_dmin) + dmaj), magic);
    else
      m = (int)fma((double)(__mul24(j, two_dmin) + dmaj), inv2d, 1e-6);
    // base + j * step_major + m * step_minor as the two multiply-adds it is: written with __mul24 and additions the
    // compiler issues two multiplies and a three-operand add (one operation more per look-up of a ~10-operation trip)
    int t1, off_i;
    if (UNIFORM_START)
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(j), "v"(step_major), "s"(__builtin_amdgcn_readfirstlane(base)));
    else
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(j), "v"(step_major), "v"(base));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(off_i) : "v"(m), "v"(step_minor), "v"(t1));
    const unsigned off = (unsigned)off_i;
    const int d = (int)((*reinterpret_cast<const uint32_t*>(cheb + off) >> qshift) & 255u);
    hit = d == 0;
    if (hit || j >= last)  // (this shape compiles to one block; testing d after the loop does not)
      break;
    j = min(j + d, last);
  }
  if (hit)
  {
    walked += (unsigned long long)(j + 1);
    const int ddx = maj_dx * j + min_dx * m, ddy = maj_dy * j + min_dy * m;
    return sqrt((double)(ddx * ddx + ddy * ddy)) * M.resolution;
  }
  walked += (unsigned long long)(last + 1);
  return range_max;
}

struct BeamModelArgs
{
  ParticlesDev p;
  int n;
  const BeamRec* beams;
  int n_beams;
  MapDev map;
  double sp_x, sp_y, sp_th;
  double off_map_factor, non_free_factor, non_free_radius;
  double range_max, z_hit, denom;
  double inv_resolution;  // correctly rounded 1 / resolution
  unsigned long long* cells_walked;
  int per_wave;          // particles per grab (<= 16)
  int* next_particle;    // work counter, zero at launch
  double* block_partials;
};

// Beam model: calcBeamModel (planar_scanner.cpp:168-234).  Wave = particle, lanes = beams in bearing order (the 64
// rays of one trip are neighbouring bearings from one pose and pass much the same cells).
#ifndef BPF_BEAM_WAVES_ATTR
#define BPF_BEAM_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(6, 8)))
#endif
// BLOCK threads share one copy of the scan in LDS (43 KB at 1081 beams): the more waves per copy, the more of them a CU
// holds.
template <int BLOCK, bool INT_DIV>
__global__ __launch_bounds__(BLOCK) BPF_BEAM_WAVES_ATTR void k_score_beam(const BeamModelArgs A)
{
  extern __shared__ __align__(16) unsigned char smem[];
  BeamRec* s_beams = reinterpret_cast<BeamRec*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < A.n_beams; i += BLOCK)
    s_beams[i] = A.beams[i];
  __syncthreads();

  const int lane = tid & 63;
  const int sub = lane & 15;
  const int wave = tid >> 6;
  const MapDev& M = A.map;
  unsigned long long walked = 0;
  double wsum = 0.0;

  // Rays differ in length and so do the particles' costs: the waves of the one resident round take their particles
  // from a counter, A.per_wave at a time, until it runs past n (every wave reaches that exit).  Which wave scores
  // which particle therefore varies from run to run -- the weights do not, and the weight total is formed afterwards
  // by the fixed-shape sum over the weights, not from per-block partials.
  for (;;)
  {
    int base = 0;
    if (lane == 0)
      base = atomicAdd(A.next_particle, A.per_wave);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base >= A.n)
      break;
    const int p_end = min(A.n, base + A.per_wave);
    const int cnt = min(16, p_end - base);
    const int pi = base + min(sub, cnt - 1);
    const double px = A.p.x[pi], py = A.p.y[pi], pth = A.p.th[pi];
    const ScannerPose sp = scanner_pose(px, py, pth, A.sp_x, A.sp_y, A.sp_th);
    const int cx0 = world_to_cell(sp.x, M.origin_x, M.resolution, M.half_x);
    const int cy0 = world_to_cell(sp.y, M.origin_y, M.resolution, M.half_y);

    double mine = 0.0;
    for (int k = 0; k < cnt; ++k)
    {
      const double c = lane_bcast(sp.c, k), s = lane_bcast(sp.s, k);
      const double ox = lane_bcast(sp.x, k), oy = lane_bcast(sp.y, k);
      const int sx0 = __builtin_amdgcn_readlane(cx0, k), sy0 = __builtin_amdgcn_readlane(cy0, k);
      double acc = 0.0;
      for (int b = lane; b < A.n_beams; b += 64)
      {
        const BeamRec B = s_beams[b];
        const double ca = c * B.cb - s * B.sb;  // cos(theta + bearing)
        const double sa = s * B.cb + c * B.sb;
        const int x1 = world_to_cell_rcp(ox + A.range_max * ca, M.origin_x, M.resolution, A.inv_resolution, M.half_x);
        const int y1 = world_to_cell_rcp(oy + A.range_max * sa, M.origin_y, M.resolution, A.inv_resolution, M.half_y);
        const double map_range = calc_range_skip<INT_DIV, true>(M, sx0, sy0, x1, y1, A.range_max, walked);
        const double z = B.obs - map_range;
        double pz = 0.0;
        pz += A.z_hit * exp(-(z * z) / A.denom);
        if (z < 0)
          pz += B.short_t;
        pz += B.tail_t;
        acc += pz * pz * pz;
      }
      const double tot = wave_sum(acc);
      if (sub == k)
        mine = tot;
    }

    if (lane < cnt)
    {
      const int q = base + lane;
      double w = A.p.w[q] * (1.0 + mine);
      w *= recalc_factor(M, px, py, A.off_map_factor, A.non_free_factor, A.non_free_radius);
      A.p.w[q] = w;
      wsum += w;
    }
  }
  (void)wsum;
  (void)wave;
  if (A.cells_walked != nullptr)
  {
    // per-wave total, one atomic per wave
    unsigned long long wsum2 = walked;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      wsum2 += __shfl_xor(wsum2, off, 64);
    if (lane == 0)
      atomicAdd(A.cells_walked, wsum2);
  }
}

// OccupancyMap::calcRange (occupancy_map.cpp:257-364) for a batch of rays: origin (ox, oy), direction given as
// (cos, sin) of the ray's angle -- formed by the caller, as the reference forms them with libm -- and max range.
__global__ void k_calc_range(const MapDev M, const double* __restrict__ ox, const double* __restrict__ oy,
                             const double* __restrict__ ca, const double* __restrict__ sa,
                             const double* __restrict__ max_range, int n, double* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const double mr = max_range[i];
  const int x0 = world_to_cell(ox[i], M.origin_x, M.resolution, M.half_x);
  const int y0 = world_to_cell(oy[i], M.origin_y, M.resolution, M.half_y);
  const int x1 = world_to_cell(ox[i] + mr * ca[i], M.origin_x, M.resolution, M.half_x);
  const int y1 = world_to_cell(oy[i] + mr * sa[i], M.origin_y, M.resolution, M.half_y);
  unsigned long long walked = 0;
  // (rays of mixed lengths: the integer form where this ray allows it -- both forms return the same cell)
  out[i] = (fabs(mr) / M.resolution <= (double)kIntDivRayCells) ? calc_range_skip<true>(M, x0, y0, x1, y1, mr, walked)
                                                                : calc_range_skip<false>(M, x0, y0, x1, y1, mr, walked);
}

}  // namespace bpf
