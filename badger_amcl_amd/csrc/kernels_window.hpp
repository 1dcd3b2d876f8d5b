// LDS-window variant of the likelihood-field scoring (same arithmetic as k_score_field).
//
// Observation (rocprofv3 PMC, profiles/): k_score_field is bound by the L1/texture front end --
// a 64-lane gather of 2-byte LUT levels costs ~31 cache accesses.  When the particle cloud is
// concentrated (the tracking regime), all particles' end points of a given group of 64 consecutive
// beams fall into a few metres of map.  So:
//
//   k_field_prep     per particle: scanner pose in padded-cell units + cos/sin (once, not per wave),
//                    and per-block sums for the cloud's mean pose and spread.
//   k_field_windows  one block: mean pose + spread -> for every 64-beam chunk a map window
//                    (256 x 256 cells around where the chunk's beams end for the mean pose) and a
//                    device-side switch: use the windows only if they are expected to catch the cloud.
//   k_score_window   block = (chunk, particle slab): stages the chunk's window of the tiled LUT in
//                    LDS (128 KB) with the term table and the chunk's beams, then lanes = particles,
//                    loop over the chunk's beams (LDS broadcast).  The gather is an LDS read; an end
//                    point outside the window is added afterwards from the global image, so ANY
//                    window is correct -- only speed depends on it.  Each lane owns its particle's
//                    running sum: no cross-lane reduction.
//   k_field_finish   folds the per-chunk partial sums in chunk order, applies the model epilogue
//                    and recalcWeight, and emits the per-block weight partials for the normaliser.
//
// If the switch says "spread cloud", k_score_window / k_field_finish return at once and
// k_score_field (which checks the same switch) does the update.
#pragma once
#include "device_types.hpp"
#include "kernels_score.hpp"

namespace bpf
{

constexpr int kWinDim = 256;        // window edge in cells (256 x 256 x 2 B = 128 KB)
constexpr int kWinThreads = 1024;   // one block per CU
constexpr int kMaxChunks = 64;      // 4096 beams / 64
constexpr int kPrepStats = 8;       // per-block sums: Qx, Qy, cos, sin, Qx^2, Qy^2, count, spare

struct WindowDesc
{
  int u0, v0;  // first padded cell of the window (u0 multiple of 8)
  int wu, wv;  // extent in cells (wu multiple of 8)
};

struct WindowPlan
{
  int use_window;  // 1: the window path does this update; 0: k_score_field does
  int n_chunks;
  int covered;     // chunks whose 3-sigma footprint fits (diagnostic)
  int pad;
  WindowDesc d[kMaxChunks];
};

// ---------------------------------------------------------------------------------------
// The per-scan staging block (beam table + term table, ~25 KB, written by the host into pinned
// memory) rides along: the first blocks copy it into the device slot the scoring kernel reads, which
// saves a separate copy operation on the stream (stage_src == nullptr: already copied).
template <bool WITH_STATS>
__global__ __launch_bounds__(256) void k_field_prep(ParticlesDev p, int n, MapDev M, double ax, double ay, double ath,
                                                   double4* __restrict__ prep, double* __restrict__ stats,
                                                   const uint4* __restrict__ stage_src, uint4* __restrict__ stage_dst,
                                                   int stage_n16)
{
  __shared__ double s_red[4][kPrepStats];
  if (stage_src != nullptr)
  {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < stage_n16)
      stage_dst[q] = stage_src[q];
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v[kPrepStats] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  if (i < n)
  {
    bool valid;
    const double4 q = field_prep_of(M, p.x[i], p.y[i], p.th[i], ax, ay, ath, &valid);
    if (valid)
    {
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; v[4] = q.x * q.x; v[5] = q.y * q.y; v[6] = 1.0;
    }
    prep[i] = q;
  }
  if (!WITH_STATS)
    return;  // the cloud statistics only feed the window planner
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kPrepStats; ++k)
  {
    const double r = wave_sum(v[k]);
    if (lane == 0)
      s_red[wave][k] = r;
  }
  __syncthreads();
  if (threadIdx.x < kPrepStats)
    stats[(size_t)blockIdx.x * kPrepStats + threadIdx.x] =
        (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// Seam A with host buffers (PlanarScanner::applyModelToSampleSet on the caller's std::vector<PFSample>): a chunk of the
// 32-byte AoS records as the copy engine left them -> the SoA set AND the per-particle (Qx, Qy, cos, sin) in one pass,
// so a chunk costs one small launch in front of its scoring launch instead of two.  The scan's staging block rides
// along exactly as in k_field_prep.
__global__ __launch_bounds__(256) void k_field_prep_aos(const double4* __restrict__ aos, ParticlesDev p, int n, MapDev M,
                                                       double ax, double ay, double ath, double4* __restrict__ prep,
                                                       const uint4* __restrict__ stage_src,
                                                       uint4* __restrict__ stage_dst, int stage_n16)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (stage_src != nullptr && i < stage_n16)
    stage_dst[i] = stage_src[i];
  if (i >= n)
    return;
  const double4 v = aos[i];
  p.x[i] = v.x;
  p.y[i] = v.y;
  p.th[i] = v.z;
  p.w[i] = v.w;
  bool valid;
  prep[i] = field_prep_of(M, v.x, v.y, v.z, ax, ay, ath, &valid);
}

// ---------------------------------------------------------------------------------------
// Tile-sorted scoring of a spread cloud (HOST_MODE 3 of k_score_field): a counting sort of the particles by the map
// tile of their scanner position.  kTileBins bins of 2^shift x 2^shift cells, row-major, so that an eighth of the
// sorted order is an eighth of the map's rows.
constexpr int kTileBins = 4096;

__device__ __forceinline__ int tile_of(const double4 q, int shift, int tx_count, int ty_count)
{
  const int cx = min(max((int)q.x, 0) >> shift, tx_count - 1);
  const int cy = min(max((int)q.y, 0) >> shift, ty_count - 1);
  return cy * tx_count + cx;
}

// One atomic per lane -- a spread cloud's 64 lanes hit (nearly) 64 different bins, and the atomics of a wave overlap --
// except when the whole wave sits in ONE bin (a cloud that is not spread after all: 10^5 atomics on a handful of
// addresses would serialise): then the first lane adds the count.  Returns the lane's position in its bin.
__device__ __forceinline__ int wave_bin_add(int* counters, int bin, bool active)
{
  const int lane = threadIdx.x & 63;
  const unsigned long long act = __ballot(active);
  if (act == 0ull)
    return 0;
  const int leader = __ffsll((long long)act) - 1;
  const int b0 = __shfl(bin, leader, 64);
  if (__ballot(active && bin == b0) == act)
  {
    int prev = 0;
    if (lane == leader)
      prev = atomicAdd(&counters[b0], __popcll(act));
    return __shfl(prev, leader, 64) + __popcll(act & ((1ull << lane) - 1ull));
  }
  return active ? atomicAdd(&counters[bin], 1) : 0;
}

// `hist` is this update's half of a double-buffered histogram (zero on entry); block 0 puts the other half and the
// cursors back to zero for the next update, so there is no clearing launch and no offsets launch (k_tile_scatter
// forms the offsets itself).
__global__ __launch_bounds__(256) void k_field_prep_tile(ParticlesDev p, int n, MapDev M, double ax, double ay, double ath,
                                                        double4* __restrict__ prep, int* __restrict__ tile,
                                                        int* __restrict__ hist, int* __restrict__ hist_next,
                                                        int* __restrict__ cursor, int shift, int tx_count, int ty_count,
                                                        const uint4* __restrict__ stage_src,
                                                        uint4* __restrict__ stage_dst, int stage_n16)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (stage_src != nullptr && i < stage_n16)
    stage_dst[i] = stage_src[i];
  if (blockIdx.x == 0)
    for (int k = threadIdx.x; k < kTileBins; k += 256)
    {
      hist_next[k] = 0;
      cursor[k] = 0;
    }
  int t = 0;
  if (i < n)
  {
    bool valid;
    const double4 q = field_prep_of(M, p.x[i], p.y[i], p.th[i], ax, ay, ath, &valid);
    prep[i] = q;
    t = tile_of(q, shift, tx_count, ty_count);
    tile[i] = t;
  }
  (void)wave_bin_add(hist, t, i < n);
}

// every block forms the bins' offsets itself (exclusive prefix of the 4096 counts in LDS: 16 KB of reads per block,
// cheaper than a launch of its own), then places its particles: offset of the bin + the bin's cursor
__global__ __launch_bounds__(256) void k_tile_scatter(int n, const int* __restrict__ tile, const int* __restrict__ hist,
                                                     int* __restrict__ cursor, const double4* __restrict__ prep,
                                                     int* __restrict__ perm, double4* __restrict__ prep_sorted)
{
  __shared__ int s_off[kTileBins];
  __shared__ int s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int per = kTileBins / 256;
  int v[per], sum = 0;
#pragma unroll
  for (int q = 0; q < per; ++q)
  {
    v[q] = hist[tid * per + q];
    sum += v[q];
  }
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
  int run = incl - sum;
  for (int q = 0; q < wave; ++q)
    run += s_w[q];
#pragma unroll
  for (int q = 0; q < per; ++q)
  {
    s_off[tid * per + q] = run;
    run += v[q];
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + tid;
  const int t = i < n ? tile[i] : 0;
  const int rank = wave_bin_add(cursor, t, i < n);
  if (i < n)
  {
    const int pos = s_off[t] + rank;
    perm[pos] = i;
    prep_sorted[pos] = prep[i];
  }
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_field_windows(const double* __restrict__ stats, int n_stat_blocks,
                                                       const double2* __restrict__ beams, int n_beams, MapDev M,
                                                       WindowPlan* plan)
{
  __shared__ double s_tot[kPrepStats];
  __shared__ double s_w[16][kPrepStats];
  __shared__ int s_cover[kMaxChunks];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double acc[kPrepStats] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  for (int b = tid; b < n_stat_blocks; b += 1024)
#pragma unroll
    for (int k = 0; k < kPrepStats; ++k)
      acc[k] += stats[(size_t)b * kPrepStats + k];
#pragma unroll
  for (int k = 0; k < kPrepStats; ++k)
  {
    const double r = wave_sum(acc[k]);
    if (lane == 0)
      s_w[wave][k] = r;
  }
  __syncthreads();
  if (tid < kPrepStats)
  {
    double t = 0.0;
    for (int w = 0; w < 16; ++w)
      t += s_w[w][tid];
    s_tot[tid] = t;
  }
  __syncthreads();
  const double cnt = s_tot[6];
  const int n_chunks = (n_beams + 63) >> 6;
  if (tid < kMaxChunks)
    s_cover[tid] = 0;
  __syncthreads();
  if (cnt >= 1.0)
  {
    const double mx = s_tot[0] / cnt, my = s_tot[1] / cnt;
    double cm = s_tot[2] / cnt, sm = s_tot[3] / cnt;
    const double rlen = sqrt(cm * cm + sm * sm);
    const double sig_th = (rlen > 1e-12 && rlen < 1.0) ? sqrt(-2.0 * log(rlen)) : (rlen >= 1.0 ? 0.0 : 10.0);
    if (rlen > 1e-12)
    {
      cm /= rlen;
      sm /= rlen;
    }
    else
    {
      cm = 1.0;
      sm = 0.0;
    }
    const double sig_x = sqrt(fmax(s_tot[4] / cnt - mx * mx, 0.0)), sig_y = sqrt(fmax(s_tot[5] / cnt - my * my, 0.0));
    // wave w plans chunks w, w+16, ...: lane = beam of the chunk
    for (int c = wave; c < n_chunks; c += 16)
    {
      const int b = c * 64 + lane;
      double ex = 0, ey = 0, rho = 0;
      const bool have = b < n_beams;
      if (have)
      {
        const double2 B = beams[b];
        ex = mx + cm * B.x - sm * B.y;
        ey = my + sm * B.x + cm * B.y;
        rho = sqrt(B.x * B.x + B.y * B.y);
      }
      double lo_x = have ? ex : 1e300, hi_x = have ? ex : -1e300, lo_y = have ? ey : 1e300, hi_y = have ? ey : -1e300;
      double rmax = rho;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
      {
        lo_x = fmin(lo_x, __shfl_xor(lo_x, off, 64));
        hi_x = fmax(hi_x, __shfl_xor(hi_x, off, 64));
        lo_y = fmin(lo_y, __shfl_xor(lo_y, off, 64));
        hi_y = fmax(hi_y, __shfl_xor(hi_y, off, 64));
        rmax = fmax(rmax, __shfl_xor(rmax, off, 64));
      }
      if (lane == 0)
      {
        const double cx = 0.5 * (lo_x + hi_x), cy = 0.5 * (lo_y + hi_y);
        // clip a kWinDim square around the centre to the padded image [0, size+1]
        double fu0 = floor(cx) - kWinDim / 2, fv0 = floor(cy) - kWinDim / 2;
        fu0 = fmin(fmax(fu0, -1e9), 1e9);
        fv0 = fmin(fmax(fv0, -1e9), 1e9);
        int u0 = ((int)fu0) & ~7, v0 = (int)fv0;
        int u1 = u0 + kWinDim, v1 = v0 + kWinDim;  // exclusive
        const int min_u = 0, min_v = 0;  // u0 stays a multiple of 8
        const int max_u = ((M.size_x + 2 * kLutPad) + 7) & ~7, max_v = M.size_y + 2 * kLutPad;  // exclusive
        u0 = max(u0, min_u); v0 = max(v0, min_v);
        u1 = min(u1, max_u); v1 = min(v1, max_v);
        WindowDesc d;
        d.u0 = u0; d.v0 = v0;
        d.wu = max(u1 - u0, 0); d.wv = max(v1 - v0, 0);
        plan->d[c] = d;
        // does the 3-sigma footprint of the cloud fit the window?
        const double need_x = 0.5 * (hi_x - lo_x) + 3.0 * sig_x + 3.0 * sig_th * rmax + 2.0;
        const double need_y = 0.5 * (hi_y - lo_y) + 3.0 * sig_y + 3.0 * sig_th * rmax + 2.0;
        s_cover[c] = (need_x <= kWinDim / 2 && need_y <= kWinDim / 2 && d.wu > 0 && d.wv > 0) ? 1 : 0;
      }
    }
  }
  __syncthreads();
  if (tid == 0)
  {
    int covered = 0;
    for (int c = 0; c < n_chunks; ++c)
      covered += s_cover[c];
    plan->n_chunks = n_chunks;
    plan->covered = covered;
    // windows pay off when most chunks are caught; a miss is still scored correctly, only slower
    plan->use_window = (cnt >= 1.0 && covered * 4 >= n_chunks * 3) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------
struct WindowScoreArgs
{
  int n;
  const double4* prep;
  const double2* beams;
  int n_beams;
  const double* table;
  int table_len;
  MapDev map;
  const WindowPlan* plan;
  double* partials;  // [n_chunks][n]
  int slabs;
};

__global__ __launch_bounds__(kWinThreads) void k_score_window(const WindowScoreArgs A)
{
  if (A.plan->use_window == 0)
    return;
  extern __shared__ __align__(16) unsigned char smem[];
  // LDS: [window: wv rows x wu cells of uint16 = byte offset of the cell's term in s_table]
  //      [term table: table_len doubles + one 0.0 pad entry] [the chunk's 64 beams]
  uint16_t* s_win = reinterpret_cast<uint16_t*>(smem);
  double* s_table = reinterpret_cast<double*>(smem + (size_t)kWinDim * kWinDim * sizeof(uint16_t));
  double2* s_beams = reinterpret_cast<double2*>(s_table + A.table_len + 1);

  const int chunk = blockIdx.x;
  const WindowDesc D = A.plan->d[chunk];
  const MapDev& M = A.map;
  const char* __restrict__ tiles = reinterpret_cast<const char*>(M.lut_tiles);
  const int tid = threadIdx.x;
  const int stride = D.wu;  // cells per LDS row (multiple of 8)

  // stage the window cell by cell from the tiled image (entries are already level*8, the byte offset of the
  // level's term); consecutive threads walk v first, i.e. along a tile's 16-byte columns
  const int n_cells = D.wu * D.wv;
  for (int c = tid; c < n_cells; c += kWinThreads)
  {
    const int du = c / D.wv, dv = c - du * D.wv;
    s_win[dv * stride + du] = *reinterpret_cast<const uint16_t*>(
        tiles + lut_byte_offset(M, (unsigned)(D.u0 + du), (unsigned)(D.v0 + dv)));
  }
  for (int i = tid; i < A.table_len; i += kWinThreads)
    s_table[i] = A.table[i];
  if (tid == 0)
    s_table[A.table_len] = 0.0;  // what a window miss adds on the fast pass
  const int b0 = chunk * 64;
  const int nb = min(64, A.n_beams - b0);
  if (tid < nb)
    s_beams[tid] = A.beams[b0 + tid];
  __syncthreads();

  const unsigned wu = (unsigned)D.wu, wv = (unsigned)D.wv;
  const int u0 = D.u0, v0 = D.v0;
  const unsigned pad_off = (unsigned)A.table_len * 8u;
  const char* table_b = reinterpret_cast<const char*>(s_table);
  const unsigned stride2 = (unsigned)stride * 2u;
  const char* win_b = reinterpret_cast<const char*>(s_win);

  const int per_slab = (A.n + A.slabs - 1) / A.slabs;
  const int p_begin = blockIdx.y * per_slab, p_end = min(A.n, p_begin + per_slab);
  for (int p = p_begin + tid; p < p_end; p += kWinThreads)
  {
    const double4 q = A.prep[p];
    double acc = 0.0;
    bool missed = false;
#pragma unroll 8
    for (int k = 0; k < nb; ++k)
    {
      const double2 B = s_beams[k];  // same address in every lane: LDS broadcast
      const double vx = fma(q.z, B.x, fma(-q.w, B.y, q.x));
      const double vy = fma(q.w, B.x, fma(q.z, B.y, q.y));
      const unsigned du = (unsigned)((int)vx - u0), dv = (unsigned)((int)vy - v0);
      const bool inside = du < wu && dv < wv;
      // an out-of-window address wraps to some other LDS word (or reads 0 past the allocation);
      // the value is discarded below, and LDS reads cannot fault
      const unsigned off = *reinterpret_cast<const uint16_t*>(win_b + (dv * stride2 + (du << 1)));
      acc += *reinterpret_cast<const double*>(table_b + (inside ? off : pad_off));
      missed |= !inside;
    }
    if (missed)
    {
      // rare: end points outside the window go through the tiled image in global memory
      for (int k = 0; k < nb; ++k)
      {
        const double2 B = s_beams[k];
        const double vx = fma(q.z, B.x, fma(-q.w, B.y, q.x));
        const double vy = fma(q.w, B.x, fma(q.z, B.y, q.y));
        const int iu = (int)vx, iv = (int)vy;
        const unsigned du = (unsigned)(iu - u0), dv = (unsigned)(iv - v0);
        if (!(du < wu && dv < wv))
        {
          const unsigned lv = *reinterpret_cast<const uint16_t*>(
              tiles + lut_byte_offset(M, clamp_cell(iu, M.size_x), clamp_cell(iv, M.size_y)));
          acc += *reinterpret_cast<const double*>(table_b + lv);
        }
      }
    }
    A.partials[(size_t)chunk * A.n + p] = acc;
  }
}

// ---------------------------------------------------------------------------------------
struct FieldFinishArgs
{
  ParticlesDev p;
  int n;
  const double* partials;
  const WindowPlan* plan;
  MapDev map;
  double off_map_factor, non_free_factor, non_free_radius;
  int model;
  GompertzDev g;
  int n_valid;
  double extra_term;
  double* block_partials;
};

__global__ __launch_bounds__(256) void k_field_finish(const FieldFinishArgs A)
{
  if (A.plan->use_window == 0)
    return;
  __shared__ double s_part[4];
  const int n_chunks = A.plan->n_chunks;
  double wsum = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < A.n; i += gridDim.x * 256)
  {
    double sum = 0.0;
    for (int c = 0; c < n_chunks; ++c)
      sum += A.partials[(size_t)c * A.n + i];
    sum += A.extra_term;
    double pf;
    if (A.model == 1)
      pf = 1.0 + sum;
    else if (A.model == 3)
    {
      if (A.n_valid > 0)
      {
        double v = sum / A.n_valid;
        v = v * A.g.input_scale + A.g.input_shift;
        v = A.g.a * exp(-1.0 * A.g.b * exp(-1.0 * A.g.c * v));
        pf = v + A.g.output_shift;
      }
      else
        pf = 1.0;
    }
    else
      pf = exp(sum);
    double w = A.p.w[i] * pf;
    w *= recalc_factor(A.map, A.p.x[i], A.p.y[i], A.off_map_factor, A.non_free_factor, A.non_free_radius);
    A.p.w[i] = w;
    wsum += w;
  }
  const double ws = wave_sum(wsum);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    s_part[wave] = ws;
  __syncthreads();
  if (threadIdx.x == 0 && A.block_partials != nullptr)
    A.block_partials[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

}  // namespace bpf
