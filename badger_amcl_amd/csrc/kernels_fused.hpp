// The tracking-regime tail of the headline step in two launches instead of five plus a host round trip:
//
//   k_normalize_cdf   ParticleFilter::updateSensor's normalisation (particle_filter.cpp:237-266) AND the resampling
//                     CDF (:372-375) in one launch: every block normalises its 2048-weight tile, publishes the tile's
//                     sum, waits for the tiles before it (they were dispatched earlier, so they are running) and
//                     writes its slice of the CDF.  Same arithmetic, same summation shapes and therefore the same bits
//                     as k_normalize_fused followed by k_scan_final.
//   k_resample_block  resampleMultinomial / resampleSystematic (:269-420) for a candidate stream of at most 4096 draws
//                     in ONE block: drand48 jump-ahead, CDF search, pose gather, histogram keys, then the KLD stop rule
//                     -- the fork's insertion-order-dependent kd-tree leaf count (pf_kdtree.cpp:97-150) grown level by
//                     level in LDS exactly as kernels_kld.hpp grows it in HBM for long streams --, the weights 1/M and
//                     updateConverged (:170-220).  The keys never leave the CU; the host only reads back
//                     (M, leaf count, bin count) from pinned memory.
#pragma once
#include <climits>

#include "kernels_kld.hpp"
#include "kernels_pf.hpp"

namespace bpf
{

struct NormCdfArgs
{
  double* w;
  int n;
  const double* block_partials;  // the scoring kernel's per-block weight sums
  int n_partials;
  FilterScalars* sc;
  double alpha_slow, alpha_fast;
  // look-back slots: [2][256] tile sums as bit patterns, all-ones = "not there yet".  A launch publishes into the
  // half `generation & 1` and block 0 puts the other half back to all-ones for the next launch, so the value is its own
  // flag: one relaxed atomic store per tile, relaxed loads on the waiting side, no fence (an agent-scope release after a
  // tile's 16 KB of stores is an L2 write-back of ~6 us, an acquire in a spin loop an L1 invalidate per iteration)
  unsigned long long* tile_slots;
  unsigned generation;
  double* cdf;           // [n + 1]
  double* coarse;        // [((n - 1) >> coarse_shift) + 2]: c[min(k << coarse_shift, n)], what k_resample_block stages
  int coarse_shift;
  int* guide;            // CDF guide table (kCdfGuide + 2)
  int* zero_word;        // the CDF-miss flag of the draw kernels that follow
};

// block 0 empties the look-back half the NEXT launch publishes into (first thing in the kernel: also on a path that
// leaves early)
__device__ __forceinline__ void reset_next_lookback_half(const NormCdfArgs& A)
{
  if (blockIdx.x == 0)
    __hip_atomic_store(&A.tile_slots[(size_t)((A.generation & 1u) ^ 1u) * BPF_RED_BLOCK + threadIdx.x], ~0ull,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The tile of block b: normalise (w / total, or `uniform` when the total is not positive), publish the tile's sum, wait
// for the tiles before it and write the tile's slice of the CDF (+ subsample, + guide).  Summation shapes of
// k_normalize_fused (tile sum) and k_scan_final (running sums).
// this thread's eight scored weights: asked for first thing, so that they arrive while the total is being formed
__device__ __forceinline__ void load_tile_weights(const NormCdfArgs& A, double (&raw)[BPF_RED_PER_THREAD])
{
  const size_t base = (size_t)blockIdx.x * BPF_RED_TILE + (size_t)threadIdx.x * BPF_RED_PER_THREAD;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
    raw[k] = (base + k < (size_t)A.n) ? A.w[base + k] : 0.0;
}

__device__ __forceinline__ void normalize_tile_and_cdf(const NormCdfArgs& A, double total, double uniform,
                                                       const double (&raw)[BPF_RED_PER_THREAD], double* s_wave,
                                                       double* s_tiles, double* s_tile_off)
{
  const int tid = threadIdx.x, b = blockIdx.x;
  const size_t base = (size_t)b * BPF_RED_TILE + (size_t)tid * BPF_RED_PER_THREAD;
  double v[BPF_RED_PER_THREAD];
  double tsum = 0.0, run = 0.0;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
  {
    double x = 0.0;
    if (base + k < (size_t)A.n)
    {
      x = (total > 0.0) ? raw[k] / total : uniform;
      A.w[base + k] = x;
      tsum += x;
    }
    run += x;
    v[k] = run;
  }
  const double tile = block_sum_256(tsum, s_wave);
  unsigned long long* slots = A.tile_slots + (size_t)(A.generation & 1u) * BPF_RED_BLOCK;
  if (tid == 0)
  {
    unsigned long long bits = (unsigned long long)__double_as_longlong(tile);
    if (tile != tile)
      bits = 0x7FF8000000000000ull;  // a NaN sum (NaN weights) must not look like the empty slot
    __hip_atomic_store(&slots[b], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- the tiles before this one (blocks are dispatched in index order: they are resident or done)
  if (tid < b)
  {
    // bounded (every wave must reach its exit whatever happens to another block): after 50 ms the tile counts as empty
    // and the CDF-miss word is raised, which the resample that follows reports
    unsigned long long v;
    long long t0 = 0;
    for (unsigned spins = 0;
         (v = __hip_atomic_load(&slots[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull; ++spins)
    {
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 255u) == 255u)
      {
        const long long now = wall_clock64();
        if (t0 == 0)
          t0 = now;
        else if (now - t0 > 5000000ll)
        {
          atomicExch(A.zero_word, 1);
          v = 0ull;
          break;
        }
      }
    }
    s_tiles[tid] = __longlong_as_double((long long)v);
  }
  __syncthreads();
  if (tid == 0)
  {
    double off = 0.0;
    for (int t = 0; t < b; ++t)  // left to right, as k_scan_final adds them
      off += s_tiles[t];
    *s_tile_off = off;
  }
  const double incl = wave_incl_scan(run);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 63)
    s_wave[wave] = incl;
  __syncthreads();
  double off = *s_tile_off;
  for (int k = 0; k < wave; ++k)
    off += s_wave[k];
  off += incl - run;
  const unsigned cmask = (1u << A.coarse_shift) - 1u;
#pragma unroll
  for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
    if (base + k < (size_t)A.n)
    {
      const unsigned idx = (unsigned)(base + k + 1);
      A.cdf[idx] = off + v[k];
      if ((idx & cmask) == 0u)
        A.coarse[idx >> A.coarse_shift] = off + v[k];
      else if (idx == (unsigned)A.n)
        A.coarse[(idx >> A.coarse_shift) + 1] = off + v[k];
    }
  if (b == 0 && tid == 0)
  {
    A.cdf[0] = 0.0;
    A.coarse[0] = 0.0;
  }
  if (A.guide != nullptr)
  {
    double lo_c = (base == 0) ? 0.0 : off;
#pragma unroll
    for (int k = 0; k < BPF_RED_PER_THREAD; ++k)
      if (base + k < (size_t)A.n)
      {
        const double hi_c = off + v[k];
        for (int j = (int)ceil(lo_c * (double)kCdfGuide); j <= kCdfGuide && (double)j / (double)kCdfGuide < hi_c; ++j)
          A.guide[j] = (int)(base + k);
        if (base + k == (size_t)A.n - 1)
          for (int j = (int)ceil(hi_c * (double)kCdfGuide); j <= kCdfGuide; ++j)
            A.guide[j] = A.n;
        lo_c = hi_c;
      }
  }
}

// ParticleFilter::updateSensor's running averages (particle_filter.cpp:243-256) with the update's total over n samples
__device__ __forceinline__ void update_weight_averages(FilterScalars* sc, double total, int n, double alpha_slow,
                                                       double alpha_fast)
{
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

// grid = tiles (<= 256: thread t of a block fetches the sum of tile t), block = 256
__global__ __launch_bounds__(BPF_RED_BLOCK) void k_normalize_cdf(const NormCdfArgs A)
{
  __shared__ double s_wave[4];
  __shared__ double s_tiles[BPF_RED_BLOCK];
  __shared__ double s_tile_off;
  const int tid = threadIdx.x, b = blockIdx.x;
  reset_next_lookback_half(A);
  double raw[BPF_RED_PER_THREAD];
  load_tile_weights(A, raw);
  // ---- the total: every block folds the partials with the same fixed tree (k_normalize_fused)
  double acc = 0.0;
  for (int i = tid; i < A.n_partials; i += BPF_RED_BLOCK)
    acc += A.block_partials[i];
  const double total = block_sum_256(acc, s_wave);
  if (b == 0 && tid == 0)
  {
    FilterScalars* sc = A.sc;
    sc->v[0] = total;
    sc->v[6] = total;
    update_weight_averages(sc, total, A.n, A.alpha_slow, A.alpha_fast);
    *A.zero_word = 0;
  }
  normalize_tile_and_cdf(A, total, 1.0 / A.n, raw, s_wave, s_tiles, &s_tile_off);
}

// Sharded filter, mailbox mode (k_normalize_gathered + the CDF): block 0 folds the scoring kernel's partials into the
// LOCAL total and posts it to every peer; every block waits for the W totals of this update, forms the global total as
// their rank-ordered sum, normalises its tile by it and writes its slice of the local CDF.  If a peer's total does not
// arrive in time the weights stay scored but not normalised (the rule of k_normalize_gathered) and no CDF is written.
struct NormCdfGatherArgs
{
  NormCdfArgs n;             // n.n = local samples; n.block_partials / n.n_partials = the fold (may be 0: already posted)
  const double* totals;      // this rank's mailbox slots, [world]
  int world;
  int global_n;
  MailboxDev mb;
  int wait_parity;
  unsigned long long wait_gen;
  int* zero_word2;           // the caller's CDF-miss flag (nullable)
  double* sum_out;           // local CDF sum (scalars[7])
};

__global__ __launch_bounds__(BPF_RED_BLOCK) void k_normalize_gathered_cdf(const NormCdfGatherArgs G)
{
  __shared__ double s_wave[4];
  __shared__ double s_tiles[BPF_RED_BLOCK];
  __shared__ double s_tile_off;
  const NormCdfArgs& A = G.n;
  const int tid = threadIdx.x, b = blockIdx.x;
  reset_next_lookback_half(A);
  double raw[BPF_RED_PER_THREAD];
  load_tile_weights(A, raw);  // (the weights are final since the scoring launch; the wait below is for the totals)
  if (A.n_partials > 0 && b == 0)
  {
    double acc = 0.0;
    for (int i = tid; i < A.n_partials; i += BPF_RED_BLOCK)
      acc += A.block_partials[i];
    const double tot = block_sum_256(acc, s_wave);
    if (tid == 0)
      A.sc->v[0] = tot;
    mb_post_total(G.mb, G.wait_parity, G.wait_gen, tot);
  }
  if (!mb_block_wait(G.mb, mb_tot_gen(G.mb.peer[G.mb.rank], G.wait_parity, 0), G.wait_gen, 0))
    return;
  double total = 0.0;
  for (int r = 0; r < G.world; ++r)
    total += G.totals[r];
  if (b == 0 && tid == 0)
  {
    A.sc->v[6] = total;
    update_weight_averages(A.sc, total, G.global_n, A.alpha_slow, A.alpha_fast);
    *A.zero_word = 0;
    if (G.zero_word2 != nullptr)
      *G.zero_word2 = 0;
  }
  normalize_tile_and_cdf(A, total, 1.0 / G.global_n, raw, s_wave, s_tiles, &s_tile_off);
  if (G.sum_out != nullptr && b == (int)gridDim.x - 1 && tid == (A.n - 1 - b * BPF_RED_TILE) / BPF_RED_PER_THREAD)
    *G.sum_out = A.cdf[A.n];  // (this thread wrote it)
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int kFusedWindow = 4096;     // candidate draws one block takes (4 per thread)
constexpr int kFusedPerThread = 4;
constexpr int kFusedMaxLevels = 128;   // deeper histogram trees go back to the general path
constexpr int kFusedMaxBins = 1024;    // distinct histogram bins in the window (one tree key per thread)
constexpr int kFusedCoarse = 4096;     // entries of the CDF subsample staged in LDS (+ 1)
// LDS: packed keys 8 B, hash table / tree children / coarse CDF 8 B, first 4 B, staged x and y 16 B per draw
constexpr size_t kFusedLds = (size_t)kFusedWindow * (8 + 8 + 4 + 16) + 16;

enum
{
  BPF_FUSED_OK = 0,
  BPF_FUSED_NO_STOP = 1,    // no stop inside the window and the window is not the whole stream
  BPF_FUSED_KEY_RANGE = 2,  // a histogram key outside the 24 + 24 + 16 bit packing
  BPF_FUSED_TOO_DEEP = 3,
  BPF_FUSED_TOO_MANY_BINS = 4
};

constexpr int kFusedDrawsPerBlock = 128;  // draw phase: the window is spread over window / 128 blocks (CUs)

struct ResampleBlockArgs
{
  ParticlesDev src;   // set a
  int n_src;
  const double* cdf;  // [n_src + 1]
  const double* coarse;  // c[min(k << coarse_shift, n)], k = 0 .. ((n - 1) >> shift) + 1 (nullable: read from cdf)
  int coarse_shift;   // ((n_src - 1) >> shift) + 1 <= kFusedCoarse
  ParticlesDev dst;   // set b
  int window;         // candidate draws 0 .. window - 1, <= kFusedWindow
  int max_samples;
  int systematic;     // 1: draw m takes r = targets[m], the set size is `window`, no stop rule
  const double* targets;
  uint64_t rng_state;
  const FusedJump* jump;  // [kFusedWindow]
  unsigned long long* keys;  // [window] packed histogram keys, draw phase -> tree phase
  unsigned* counter;  // blocks that finished their draws (0 between launches)
  const int* limit;   // resampleLimit per leaf count, [0 .. window]
  int* miss_flag;
  double thr;         // updateConverged's distance threshold
  FilterScalars* sc;
  int* conv_count;
  volatile int* result_host;  // pinned: three result words (fused_publish)
  int generation;
  int debug;          // also copy the phase clocks out ([6], [8 ..])
};

// which side of node v key i goes to: split axis = largest |delta| between v's key and the first different key f
// routed through v, earliest axis on ties (pf_kdtree.cpp:133-146); high side if greater on that axis.  Packed keys:
// the offsets of kld_pack cancel in differences and comparisons.
__device__ __forceinline__ int fused_side(unsigned long long kv, unsigned long long kf, unsigned long long ki)
{
  const int v0 = (int)(kv >> 40), v1 = (int)((kv >> 16) & 0xFFFFFFull), v2 = (int)(kv & 0xFFFFull);
  const int f0 = (int)(kf >> 40), f1 = (int)((kf >> 16) & 0xFFFFFFull), f2 = (int)(kf & 0xFFFFull);
  const int i0 = (int)(ki >> 40), i1 = (int)((ki >> 16) & 0xFFFFFFull), i2 = (int)(ki & 0xFFFFull);
  const int d0 = abs(f0 - v0), d1 = abs(f1 - v1), d2 = abs(f2 - v2);
  int best = 0, cv = v0, ci = i0;
  if (d0 > best)
    best = d0;
  if (d1 > best)
  {
    best = d1;
    cv = v1;
    ci = i1;
  }
  if (d2 > best)
  {
    cv = v2;
    ci = i2;
  }
  return ci > cv ? 1 : 0;
}

__device__ __forceinline__ int wave_incl_scan_int(int v)
{
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const int o = __shfl_up(v, off, 64);
    if (lane >= off)
      v += o;
  }
  return v;
}

// The histogram tree of the window's distinct keys, level by level (kernels_kld.hpp): per level the earliest waiting
// key on either side of every node becomes that node's child, the others step down to it.  Thread k holds tree key
// s_list[k] (a draw index).  ONE_WAVE: at most 64 keys, wave 0 runs alone and LDS order within the wave replaces the
// block barriers.  Returns the number of levels used (kFusedMaxLevels: gave up).
template <bool ONE_WAVE>
__device__ __forceinline__ int fused_tree(int my, bool have, const unsigned long long* s_key, int* s_first,
                                          int* s_child, unsigned* s_nodelta)
{
  int cur = (have && my != 0) ? 0 : -1;  // every tree key but the root's (draw 0) waits at the root
  {
    int wmin = cur == 0 ? my : INT_MAX;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      wmin = min(wmin, __shfl_xor(wmin, off, 64));
    if ((threadIdx.x & 63) == 0 && wmin != INT_MAX)
      atomicMin(&s_first[0], wmin);
  }
  if (ONE_WAVE)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  else
    __syncthreads();
  const unsigned long long mine = have ? s_key[my] : 0ull;
  int level = 0;
  for (; level < kFusedMaxLevels; ++level)
  {
    int side = 0, fst = 0;
    if (cur >= 0)
    {
      fst = s_first[cur];
      side = fused_side(s_key[cur], s_key[fst], mine);
      atomicMin(&s_child[2 * cur + side], my);
    }
    if (ONE_WAVE)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    else
      __syncthreads();
    int still = 0;
    if (cur >= 0)
    {
      const int ch = s_child[2 * cur + side];
      if (ch == my)
      {
        cur = -1;  // my is a node now
        if (fst == my)
          atomicOr(&s_nodelta[my >> 5], 1u << (my & 31));  // its creation ends the parent's time as a leaf: +1 - 1
      }
      else
      {
        cur = ch;
        atomicMin(&s_first[ch], my);
        still = 1;
      }
    }
    if (ONE_WAVE)
    {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (__builtin_amdgcn_ballot_w64(still != 0) == 0)
        break;
    }
    else if (!__syncthreads_or(still))
      break;
  }
  return level;
}

// phase clock (100 MHz constant clock), kept in LDS and copied to the pinned result block [8 + k] at the end (a store
// to host memory in the middle would stall the next barrier); read with BPF_DEBUG only
#define BPF_FUSED_STAMP(k)                                                   \
  do                                                                         \
  {                                                                          \
    if (threadIdx.x == 0 && S.debug)                                         \
      S.stamp[k] = (int)(unsigned)wall_clock64();                            \
  } while (0)

// static LDS of the window kernels (k_resample_block and the sharded k_shard_stop_block share the stop rule below)
struct FusedStatics
{
  int stop, bad, leaf, bins, count, levels, last;
  int debug;  // phase clocks wanted (BPF_DEBUG); set by thread 0 first thing, read by thread 0 only
  int stamp[16];
  int wx[16];
  int limit[kFusedMaxBins + 1];
  int list[kFusedMaxBins];
  unsigned nodelta[kFusedWindow / 32];
};

// dynamic LDS (kFusedLds bytes): packed keys, the hash table (first the coarse CDF, last the tree's children), the
// first different key per node, the staged x and y of every draw
struct FusedLdsMap
{
  unsigned long long* key;  // [W]
  int* hash;                // [2 W]
  int* first;               // [W]
  double* x;                // [W]
  double* y;                // [W]
};

__device__ __forceinline__ FusedLdsMap fused_lds_map(unsigned char* smem)
{
  constexpr int W = kFusedWindow;
  FusedLdsMap L;
  L.key = reinterpret_cast<unsigned long long*>(smem);
  L.hash = reinterpret_cast<int*>(smem + (size_t)W * 8);
  L.first = reinterpret_cast<int*>(smem + (size_t)W * 16 + 16);
  L.x = reinterpret_cast<double*>(smem + (size_t)W * 20 + 16);
  L.y = L.x + W;
  return L;
}

// what the block that holds every draw of the window clears before it loads them
__device__ __forceinline__ void fused_stop_init(FusedStatics& S, const FusedLdsMap& L, int window, int systematic,
                                                const int* __restrict__ limit)
{
  constexpr int W = kFusedWindow;
  const int tid = threadIdx.x;
  for (int k = tid; k <= kFusedMaxBins; k += 1024)
    S.limit[k] = (k <= window && !systematic) ? limit[k] : INT_MAX;
  for (int s = tid; s < W; s += 1024)
    L.first[s] = INT_MAX;
  for (int s = tid; s < W / 32; s += 1024)
    S.nodelta[s] = 0u;
  for (int s = tid; s < 2 * W; s += 1024)
    L.hash[s] = INT_MAX;
  if (tid == 0)
  {
    S.stop = INT_MAX;
    S.bad = 0;
    S.leaf = 0;
    S.bins = 0;
    S.count = 0;
    S.levels = 0;
  }
}

// The result of a window kernel for the host, in pinned memory: three 64-bit words that each carry the launch's
// generation in their high half -- (gen | M), (gen | leaf count << 16 | bin count), (gen | status << 8 | levels) --
// so the host takes them when all three show the generation it waits for and the kernel needs no fence in front of a
// flag word (a system-scope release is a write-back of everything the block has just stored: ~5 us behind the 16 KB of
// weights; the host reads none of that).
__device__ __forceinline__ void fused_publish(volatile int* result_host, int generation, int M, int leaf, int bins,
                                              int status, int levels)
{
  unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<int*>(result_host));
  const unsigned long long g = (unsigned long long)(unsigned)generation << 32;
  __hip_atomic_store(&out[0], g | (unsigned)M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&out[1], g | ((unsigned)(leaf & 0xFFFF) << 16) | (unsigned)(bins & 0xFFFF), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&out[2], g | ((unsigned)(status & 0xFF) << 8) | (unsigned)(levels & 0xFF), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

// The stop rule over the window's draws, for the block that holds all of them (keys in L.key, thread t owns draws
// 4t .. 4t + 3: act / pk): repeated keys fold onto their first draw, the histogram tree grows over the distinct ones,
// and the first draw m with m + 1 > resampleLimit(leaves so far) ends the set (particle_filter.cpp:411-417).  Leaves
// S.leaf / S.bins / S.levels behind; *M_out = samples of the new set, *status_out = BPF_FUSED_*.
__device__ __forceinline__ void fused_stop_rule(FusedStatics& S, const FusedLdsMap& L, int window, int systematic,
                                                int max_samples, const bool (&act)[kFusedPerThread],
                                                const unsigned long long (&pk)[kFusedPerThread], int* M_out,
                                                int* status_out)
{
  constexpr int W = kFusedWindow;
  constexpr int Q = kFusedPerThread;
  constexpr int kHashMask = 2 * W - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m_base = tid * Q;
  const bool usable = S.bad == 0;
  // ---- repeated keys fold onto their first occurrence (PFKDTree::insertNode: equal key -> value +=):
  // open-addressing table of draw indices, a slot holds the earliest draw with its key
  bool is_first[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q)
    is_first[q] = false;
  if (usable)
  {
    int slot[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      slot[q] = 0;
      if (!act[q])
        continue;
      const int m = m_base + q;
      unsigned h = (unsigned)((pk[q] * 0x9E3779B97F4A7C15ull) >> 40) & kHashMask;
      for (;;)
      {
        int held = *reinterpret_cast<volatile int*>(&L.hash[h]);
        if (held == INT_MAX)
        {
          held = atomicCAS(&L.hash[h], INT_MAX, m);
          if (held == INT_MAX)
            break;
        }
        if (L.key[held] == pk[q])
        {
          atomicMin(&L.hash[h], m);
          break;
        }
        h = (h + 1) & kHashMask;
      }
      slot[q] = (int)h;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      is_first[q] = act[q] && L.hash[slot[q]] == m_base + q;
      if (is_first[q])
      {
        const int pos = atomicAdd(&S.count, 1);
        if (pos < kFusedMaxBins)
          S.list[pos] = m_base + q;
      }
    }
    __syncthreads();
    for (int s = tid; s < 2 * W; s += 1024)
      L.hash[s] = INT_MAX;
    __syncthreads();
  }
  const int n_bins = S.count;
  BPF_FUSED_STAMP(4);

  // ---- the tree
  if (usable && n_bins <= kFusedMaxBins)
  {
    const bool have = tid < n_bins;
    const int my = have ? S.list[tid] : INT_MAX;
    if (n_bins <= 64)
    {
      if (wave == 0)
      {
        const int lv = fused_tree<true>(my, have, L.key, L.first, L.hash, S.nodelta);
        // The stop rule straight from the (at most 64) tree keys, without a scan over the draws: the leaf count only
        // changes at a key's first draw m_k, so between two such draws it is constant and the first draw of that
        // stretch with m + 1 > resampleLimit(leaves) (particle_filter.cpp:416) is max(m_k, limit) -- if the stretch
        // is that long.  Every lane ranks its key against all others (values through v_readlane).
        const int mk = have ? my : INT_MAX;
        const int dk = (have && !((S.nodelta[my >> 5] >> (my & 31)) & 1u)) ? 1 : 0;
        int leaf_k = 0, bins_k = 0, next_k = window;
        for (int j = 0; j < n_bins; ++j)
        {
          const int mj = __builtin_amdgcn_readlane(mk, j);
          const int dj = __builtin_amdgcn_readlane(dk, j);
          leaf_k += (mj <= mk) ? dj : 0;
          bins_k += (mj <= mk) ? 1 : 0;
          next_k = (mj > mk) ? min(next_k, mj) : next_k;
        }
        int cand = INT_MAX;
        if (have && !systematic)
        {
          const int first_over = max(mk, S.limit[min(leaf_k, kFusedMaxBins)]);
          if (first_over < next_k)
            cand = first_over + 1;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
          cand = min(cand, __shfl_xor(cand, off, 64));
        const int M1 = (cand == INT_MAX ? window : cand) - 1;  // the last draw of the new set
        if (have && mk <= M1 && M1 < next_k)
        {
          S.leaf = leaf_k;
          S.bins = bins_k;
        }
        if (tid == 0)
        {
          S.levels = lv;
          S.stop = cand;
        }
      }
    }
    else
    {
      const int lv = fused_tree<false>(my, have, L.key, L.first, L.hash, S.nodelta);
      if (tid == 0)
        S.levels = lv;
    }
  }
  else if (usable && tid == 0)
    S.bad = BPF_FUSED_TOO_MANY_BINS;
  __syncthreads();
  if (S.levels >= kFusedMaxLevels && tid == 0)
    S.bad = BPF_FUSED_TOO_DEEP;
  BPF_FUSED_STAMP(5);

  // ---- more than 64 bins: leaf / bin count after every draw by prefix sums (both counts travel in one word, leaves
  // low, bins << 16: each stays below 4097) and the first draw with m + 1 > resampleLimit(leaves)
  const bool by_scan = !(usable && n_bins <= 64);
  int pxy[Q];
  int bxy = 0;
#pragma unroll
  for (int q = 0; q < Q; ++q)
    pxy[q] = 0;
  if (by_scan)
  {
    int sxy = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      const int m = m_base + q;
      if (is_first[q])
        sxy += (1 << 16) + (((S.nodelta[m >> 5] >> (m & 31)) & 1u) ? 0 : 1);
      pxy[q] = sxy;
    }
    const int ixy = wave_incl_scan_int(sxy);
    if (lane == 63)
      S.wx[wave] = ixy;
    __syncthreads();
    bxy = ixy - sxy;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      bxy += (k < wave) ? S.wx[k] : 0;
  }
  if (!systematic && by_scan)
  {
    int stop = INT_MAX;
#pragma unroll
    for (int q = 0; q < Q; ++q)
      if (act[q] && stop == INT_MAX &&
          m_base + q + 1 > S.limit[min((bxy + pxy[q]) & 0xFFFF, kFusedMaxBins)])  // particle_filter.cpp:416
        stop = m_base + q + 1;
    if (stop != INT_MAX)
      atomicMin(&S.stop, stop);
  }
  __syncthreads();
  int M = S.stop;
  int status = S.bad;
  if (M == INT_MAX)
  {
    M = window;
    // no stop: fine when the window is the loop's own bound, sample_count < max_samples (:381)
    if (!systematic && window < max_samples && status == 0)
      status = BPF_FUSED_NO_STOP;
  }
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (by_scan && m_base + q == M - 1)
    {
      S.leaf = (bxy + pxy[q]) & 0xFFFF;
      S.bins = (bxy + pxy[q]) >> 16;
    }
  __syncthreads();
  BPF_FUSED_STAMP(6);
  *M_out = M;
  *status_out = status;
}

// grid = ceil(window / 128) blocks of 1024 threads.  Draw phase: block b takes draws 128 b .. 128 b + 127 (its first two
// waves, one draw per lane; a CU's texture path takes about one cache line per clock, so 4096 random gathers want to be
// spread over many CUs).  The block that finishes last then holds every key and runs the rest alone.
__global__ __launch_bounds__(1024) void k_resample_block(const ResampleBlockArgs A)
{
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ FusedStatics S;
  const FusedLdsMap L = fused_lds_map(smem);
  double* s_coarse = reinterpret_cast<double*>(L.hash);  // [W + 1] (the 16-byte pad takes entry W)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int Q = kFusedPerThread;
  const bool stamps = blockIdx.x == 0;
  if (tid == 0)
    S.debug = A.debug;
  if (stamps)
    BPF_FUSED_STAMP(0);

  // ================================================================== draw phase (every block)
  const double* __restrict__ c = A.cdf;
  const int n = A.n_src;
  const int shift = A.coarse_shift;
  const int n_coarse = ((n - 1) >> shift) + 1;  // brackets; entry k = c[min(k << shift, n)], k = 0 .. n_coarse
  if (A.coarse != nullptr)
    for (int k = tid; k <= n_coarse; k += 1024)
      s_coarse[k] = A.coarse[k];
  else
    for (int k = tid; k <= n_coarse; k += 1024)
      s_coarse[k] = c[min(k << shift, n)];
  // draw m takes stream element 2m + 2 (element 2m + 1 is the test against w_diff = 0, particle_filter.cpp:383-393)
  const int m_draw = blockIdx.x * kFusedDrawsPerBlock + tid;
  const bool drawing = tid < kFusedDrawsPerBlock && m_draw < A.window;
  double r = 0.0;
  if (drawing)
  {
    if (A.systematic)
      r = A.targets[m_draw];
    else
    {
      const FusedJump J = A.jump[m_draw];
      r = ldexp((double)((J.a * A.rng_state + J.c) & ((1ull << 48) - 1)), -48);
    }
  }
  __syncthreads();
  if (stamps)
    BPF_FUSED_STAMP(8);
  // CDF search (first i with c[i] <= r < c[i+1], :394-398): bracket from the staged subsample, then inside it
  __shared__ int s_brk[kFusedDrawsPerBlock];
  double* s_vals = L.x;  // [draws per block][32] (the pose staging area is idle in this phase)
  static_assert(kFusedDrawsPerBlock * 32 <= 2 * kFusedWindow, "bracket staging must fit the L.x / L.y area");
  int lo = 0;
  bool miss = false;
  if (drawing)
  {
    int hi = n_coarse;
    miss = !(r < s_coarse[n_coarse]) || !(s_coarse[0] <= r);
    if (miss)
      atomicExch(A.miss_flag, 1);  // reference: ROS_ASSERT(i < sample_count), particle_filter.cpp:399
    else
      while (hi - lo > 1)
      {
        const int mid = lo + ((hi - lo) >> 1);
        if (s_coarse[mid] <= r)
          lo = mid;
        else
          hi = mid;
      }
    lo = lo << shift;
    s_brk[tid] = miss ? -1 : lo;
  }
  else if (tid < kFusedDrawsPerBlock)
    s_brk[tid] = -1;
  int i_sel = n - 1;
  if (shift <= 5)
  {
    // the block's 128 brackets (at most 32 values = 256 bytes each, aligned) in ONE round, eight threads per bracket
    // and 32 bytes per thread, then the bisection runs in LDS (d_cdf has the slack for a bracket that ends past n)
    __syncthreads();
    for (int t = tid; t < kFusedDrawsPerBlock * 8; t += 1024)
    {
      const int j = t >> 3, part = t & 7;
      const int base = s_brk[j];
      if (base >= 0 && part * 4 < (1 << shift))
      {
        const double4 v = *reinterpret_cast<const double4*>(c + base + part * 4);
        *reinterpret_cast<double4*>(s_vals + j * 32 + part * 4) = v;
      }
    }
    __syncthreads();
    if (drawing && !miss)
    {
      const double* vals = s_vals + tid * 32;
      int l = 0, h = min(1 << shift, n - lo);  // vals[l] <= r < vals[h] (vals[h] = c[lo + h], the next bracket's first)
      while (h - l > 1)
      {
        const int mid = l + ((h - l) >> 1);
        if (vals[mid] <= r)
          l = mid;
        else
          h = mid;
      }
      i_sel = lo + l;
    }
  }
  else if (drawing && !miss)
  {
    int hi = min(lo + (1 << shift), n);
    while (hi - lo > 1)
    {
      const int mid = lo + ((hi - lo) >> 1);
      if (c[mid] <= r)
        lo = mid;
      else
        hi = mid;
    }
    i_sel = lo;
  }
  if (stamps)
    BPF_FUSED_STAMP(9);
  if (drawing)
  {
    const int i = i_sel;
    const double x = A.src.x[i], y = A.src.y[i], th = A.src.th[i];
    // What the last block reads back (x, y, key) is stored write-through (agent-scope atomic stores: sc1) and read
    // there with agent-scope loads, so the hand-off needs no release / acquire fence, only every storing wave's
    // drain in front of the block's ticket (cdna_hip_programming.md, Guideline 16 form R1; fences by all 1 024
    // threads here took 2.8 us, and an acquire in the last block 1.5).  theta is not read in this launch.
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&A.dst.x[m_draw]),
                       (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&A.dst.y[m_draw]),
                       (unsigned long long)__double_as_longlong(y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    A.dst.th[m_draw] = th;
    int key[3];
    pose_key(x, y, th, key);
    unsigned long long pk1;
    if (!kld_pack(key, &pk1))
      pk1 = kKldEmpty;  // reported by the tree phase
    __hip_atomic_store(&A.keys[m_draw], pk1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (stamps)
      BPF_FUSED_STAMP(10);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores ...
  __syncthreads();                                    // ... before the block takes its ticket
  if (stamps)
    BPF_FUSED_STAMP(11);
  if (tid == 0)
  {
    const unsigned prev = __hip_atomic_fetch_add(A.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    S.last = prev == gridDim.x - 1;
    if (S.last)
      __hip_atomic_store(A.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!S.last)
  {
    if (stamps && tid == 0 && A.debug)
    {
      for (int k = 8; k < 12; ++k)
        A.result_host[20 + k] = S.stamp[k];  // diagnostics only: block 0's draw sub-phases
      A.result_host[20] = S.stamp[0];
    }
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the loads below the ticket)
  BPF_FUSED_STAMP(1);
  const long long shader_clk0 = clock64();

  // ================================================================== the last block: every draw of the window
  // thread t owns draws m = 4t .. 4t + 3 from here on; their loads are in flight while the tables are cleared
  const int m_base = tid * Q;
  bool act[Q];
  unsigned long long pk[Q];
  double lx[Q], ly[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q)
  {
    const int m = m_base + q;
    act[q] = m < A.window;
    pk[q] = act[q] ? __hip_atomic_load(&A.keys[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kKldEmpty;
    lx[q] = act[q] ? __hip_atomic_load(&A.dst.x[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    ly[q] = act[q] ? __hip_atomic_load(&A.dst.y[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  }
  fused_stop_init(S, L, A.window, A.systematic, A.limit);
  {
    bool bad_key = false;
#pragma unroll
    for (int q = 0; q < Q; ++q)
      if (act[q])
      {
        const int m = m_base + q;
        bad_key |= pk[q] == kKldEmpty;
        L.key[m] = pk[q];
        L.x[m] = lx[q];
        L.y[m] = ly[q];
      }
    __syncthreads();  // S.bad = 0 is in place
    if (bad_key)
      S.bad = BPF_FUSED_KEY_RANGE;
  }
  __syncthreads();
  BPF_FUSED_STAMP(3);

  int M, status;
  fused_stop_rule(S, L, A.window, A.systematic, A.max_samples, act, pk, &M, &status);
  // The host needs (M, leaf count, bin count, status) and nothing of what follows: it gets them now and spends its
  // turn-around (its bookkeeping, the next launches, which the stream orders behind this one) while the block writes
  // the weights and counts the converged samples.
  if (tid == 0 && !A.debug)
    fused_publish(A.result_host, A.generation, M, S.leaf, S.bins, status, S.levels);

  // ---- weights 1 / M (:409,458-462) and updateConverged (:170-220) in k_resample_tail_small's summation shape,
  // from the poses staged in LDS
  if (status == 0)
  {
    const double weight = 1.0 / (double)M;
    double ax = 0.0, ay = 0.0;
    for (int i = tid; i < M; i += 1024)
    {
      A.dst.w[i] = weight;
      ax += L.x[i];
      ay += L.y[i];
    }
    ax = wave_sum(ax);
    ay = wave_sum(ay);
    double* s_px = reinterpret_cast<double*>(L.key);  // the keys are done with
    double* s_py = s_px + 16;
    int* s_pc = reinterpret_cast<int*>(s_py + 16);
    if (lane == 0)
    {
      s_px[wave] = ax;
      s_py[wave] = ay;
    }
    __syncthreads();
    double sxx = 0.0, syy = 0.0;
    for (int k = 0; k < 16; ++k)
    {
      sxx += s_px[k];
      syy += s_py[k];
    }
    const double mx = sxx / M, my = syy / M;
    int cnt = 0;
    for (int i = tid; i < M; i += 1024)
      if (fabs(L.x[i] - mx) <= A.thr && fabs(L.y[i] - my) <= A.thr)
        cnt++;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0)
      s_pc[wave] = cnt;
    __syncthreads();
    if (tid == 0)
    {
      int tot = 0;
      for (int k = 0; k < 16; ++k)
        tot += s_pc[k];
      *A.conv_count = tot;
      A.sc->v[3] = sxx;
      A.sc->v[4] = syy;
    }
  }
  BPF_FUSED_STAMP(7);
  if (tid == 0)
  {
    volatile int* out = A.result_host;
    if (A.debug)
    {
      // with the phase clocks: everything at the end, so that the clocks are there when the result is
      out[6] = (int)(clock64() - shader_clk0);
      for (int k = 0; k < 12; ++k)
        out[8 + k] = S.stamp[k];
      __threadfence_system();
      fused_publish(out, A.generation, M, S.leaf, S.bins, status, S.levels);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Sharded filter: the same stop rule for a draw window that every shard holds after the exchange (rows x, y, theta as
// double bits and three int64 key rows; k_draw_window stored this shard's columns into all peers, or an integer
// all-reduce assembled them).  One block per rank, every rank redundantly: wait for the window (mailbox mode), stop
// rule, then this rank adopts samples [M rank / W, M (rank + 1) / W) with weights 1 / M and evaluates updateConverged
// over all M (k_shard_tail_small's shapes).  Replaces the key copy to pinned memory, the host's ordered replay and the
// tail launch of the stage-by-stage path.
struct ShardStopArgs
{
  const long long* window;  // [6][stride]
  int stride;
  int count;                // draws 0 .. count - 1 (<= kFusedWindow)
  int systematic;           // 1: the set is the whole window, no stop rule
  int max_samples;
  const int* limit;         // resampleLimit per leaf count, [0 .. count]
  int rank, world;
  ParticlesDev dst;         // this rank's other set
  double thr;
  FilterScalars* sc;
  int* conv_count;
  MailboxDev mb;            // world 0: nothing to wait for
  int wait_parity;
  unsigned long long wait_gen;
  volatile int* result_host;  // pinned: three result words (fused_publish)
  int generation;
  int debug;
};

constexpr int BPF_FUSED_EXCHANGE = 5;  // the window did not arrive (mailbox time-out): nothing was written

// the body of k_shard_stop_block (every thread of the 1 024-thread block calls it)
__device__ __forceinline__ void shard_stop_body(const ShardStopArgs& A, FusedStatics& S, const FusedLdsMap& L)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int Q = kFusedPerThread;
  volatile int* out = A.result_host;
  if (tid == 0)
    S.debug = A.debug;
  BPF_FUSED_STAMP(0);
  if (A.mb.world > 0 && !mb_block_wait(A.mb, mb_win_done(A.mb.peer[A.mb.rank], A.wait_parity, 0), A.wait_gen, 1))
  {
    if (tid == 0)
      fused_publish(out, A.generation, 0, 0, 0, BPF_FUSED_EXCHANGE, 0);
    return;
  }
  BPF_FUSED_STAMP(1);
  fused_stop_init(S, L, A.count, A.systematic, A.limit);
  const long long* __restrict__ wx = A.window;
  const long long* __restrict__ wy = A.window + (size_t)A.stride;
  const long long* __restrict__ wk = A.window + (size_t)3 * A.stride;
  const int m_base = tid * Q;
  bool act[Q];
  unsigned long long pk[Q];
  {
    bool bad_key = false;
    long long raw[Q][5];
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      const int m = min(m_base + q, A.count - 1);
      raw[q][0] = wx[m];
      raw[q][1] = wy[m];
      raw[q][2] = wk[m];
      raw[q][3] = wk[(size_t)A.stride + m];
      raw[q][4] = wk[(size_t)2 * A.stride + m];
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
    {
      const int m = m_base + q;
      act[q] = m < A.count;
      pk[q] = kKldEmpty;
      if (act[q])
      {
        // the rows carry the int keys sign-extended; a value outside int cannot be packed either
        int key[3];
        bool fits = true;
#pragma unroll
        for (int d = 0; d < 3; ++d)
        {
          key[d] = (int)raw[q][2 + d];
          fits &= (long long)key[d] == raw[q][2 + d];
        }
        unsigned long long p1;
        if (fits && kld_pack(key, &p1))
          pk[q] = p1;
        bad_key |= pk[q] == kKldEmpty;
        L.key[m] = pk[q];
        L.x[m] = __longlong_as_double(raw[q][0]);
        L.y[m] = __longlong_as_double(raw[q][1]);
      }
    }
    __syncthreads();  // S.bad = 0 is in place
    if (bad_key)
      S.bad = BPF_FUSED_KEY_RANGE;
  }
  __syncthreads();
  BPF_FUSED_STAMP(3);

  int M, status;
  fused_stop_rule(S, L, A.count, A.systematic, A.max_samples, act, pk, &M, &status);
  if (tid == 0 && !A.debug)
    fused_publish(out, A.generation, M, S.leaf, S.bins, status, S.levels);  // (the host's turn-around runs beside the tail)

  if (status == 0)
  {
    const int lo = (int)(((long long)M * A.rank) / A.world), hi = (int)(((long long)M * (A.rank + 1)) / A.world);
    const long long* __restrict__ wth = A.window + (size_t)2 * A.stride;
    const double weight = 1.0 / (double)M;
    double ax = 0.0, ay = 0.0;
    for (int i = tid; i < M; i += 1024)
    {
      const double xv = L.x[i], yv = L.y[i];
      ax += xv;
      ay += yv;
      if (i >= lo && i < hi)
      {
        A.dst.x[i - lo] = xv;
        A.dst.y[i - lo] = yv;
        A.dst.th[i - lo] = __longlong_as_double(wth[i]);
        A.dst.w[i - lo] = weight;
      }
    }
    ax = wave_sum(ax);
    ay = wave_sum(ay);
    double* s_px = reinterpret_cast<double*>(L.key);  // the keys are done with
    double* s_py = s_px + 16;
    int* s_pc = reinterpret_cast<int*>(s_py + 16);
    if (lane == 0)
    {
      s_px[wave] = ax;
      s_py[wave] = ay;
    }
    __syncthreads();
    double sxx = 0.0, syy = 0.0;
    for (int k = 0; k < 16; ++k)
    {
      sxx += s_px[k];
      syy += s_py[k];
    }
    const double mx = sxx / M, my = syy / M;
    int cnt = 0;
    for (int i = tid; i < M; i += 1024)
      if (fabs(L.x[i] - mx) <= A.thr && fabs(L.y[i] - my) <= A.thr)
        cnt++;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0)
      s_pc[wave] = cnt;
    __syncthreads();
    if (tid == 0)
    {
      int tot = 0;
      for (int k = 0; k < 16; ++k)
        tot += s_pc[k];
      *A.conv_count = tot;
      A.sc->v[3] = sxx;
      A.sc->v[4] = syy;
    }
  }
  BPF_FUSED_STAMP(7);
  if (tid == 0)
  {
    if (A.debug)
    {
      for (int k = 0; k < 8; ++k)
        out[8 + k] = S.stamp[k];
      __threadfence_system();
      fused_publish(out, A.generation, M, S.leaf, S.bins, status, S.levels);
    }
  }
}


__global__ __launch_bounds__(1024) void k_shard_stop_block(const ShardStopArgs A)
{
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ FusedStatics S;
  const FusedLdsMap L = fused_lds_map(smem);
  shard_stop_body(A, S, L);
}

// Mailbox mode, tracking regime: the draw window AND its consumer in one launch.  grid = ceil(draws / 128) blocks; every
// block resolves its draws (k_draw_window's column function), stores the columns this shard owns into all peers'
// windows and takes a ticket; the block that takes the last one posts this shard's "done" word to every peer and goes
// on as k_shard_stop_block does (wait for the W "done" words, stop rule, adoption, updateConverged).  One launch and
// one launch boundary less per resample than k_draw_window followed by k_shard_stop_block.
struct ShardResampleArgs
{
  WindowArgs W;
  ShardStopArgs S;
};

__global__ __launch_bounds__(1024) void k_shard_resample_block(const ShardResampleArgs A)
{
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ FusedStatics S;
  __shared__ double s_sums[kMailboxMaxWorld];
  __shared__ double s_slice[2];
  const FusedLdsMap L = fused_lds_map(smem);
  const WindowArgs& D = A.W;
  const int tid = threadIdx.x;
  double* s_coarse = nullptr;
  if (D.coarse != nullptr)
  {
    s_coarse = reinterpret_cast<double*>(L.hash);  // (as in k_resample_block: [W + 1] doubles)
    const int n_coarse = ((D.n_src - 1) >> D.coarse_shift) + 1;
    for (int k = tid; k <= n_coarse; k += 1024)
      s_coarse[k] = D.coarse[k];
  }
  if (tid < D.world)
    s_sums[tid] = D.sums[tid];
  __syncthreads();
  if (tid == 0)
    shard_slice(s_sums, D.sums_are_totals, D.rank, D.world, &s_slice[0], &s_slice[1]);
  __syncthreads();
  // 128 draws per block (its first two waves), as in k_resample_block: a CU's texture path takes about one cache line
  // per clock, so the window's random gathers want many CUs
  const int o = blockIdx.x * kFusedDrawsPerBlock + tid;
  const bool live = tid < kFusedDrawsPerBlock && D.m0 + o < D.m1;
  long long out[6] = { 0, 0, 0, 0, 0, 0 };
  const bool owned = live && draw_window_column(D, o, out, s_slice[0], s_slice[1], s_coarse);
  if (owned)
    for (int r = 0; r < D.mb.world; ++r)
    {
      long long* win = mb_window(D.mb.peer[r], D.mb_parity, D.mb.max_window);
#pragma unroll
      for (int k = 0; k < 6; ++k)
        win[(size_t)k * (size_t)D.mb.max_window + o] = out[k];
    }
  // every storing wave drains its peer stores, one system-scope release per block, the ticket; the last block posts
  // "done" (mb_window_done_when_last, with the block kept instead of retired)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0)
  {
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(D.mb_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    S.last = prev == gridDim.x - 1;
    if (S.last)
    {
      __hip_atomic_store(D.mb_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int r = 0; r < D.mb.world; ++r)
        __hip_atomic_store(mb_win_done(D.mb.peer[r], D.mb_parity, D.mb.rank), D.mb_gen, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (!S.last)
    return;
  shard_stop_body(A.S, S, L);
}

}  // namespace bpf
