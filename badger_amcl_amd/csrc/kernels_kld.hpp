// The KLD stop rule of resampleMultinomial on the device, for draw streams too long to replay on the
// host (a spread cloud: no early stop, 10^5 keys).
//
// Reference (particle_filter.cpp:401-418, pf_kdtree.cpp:97-150): draw m's pose is inserted into the
// histogram kd-tree and the loop stops as soon as  m + 1 > resampleLimit(leaf count).  In this fork every
// tree node holds a bin key and counts as a leaf until the first different key is routed through it,
// so the leaf count depends on the insertion order.  What makes it parallel:
//
//   * the tree after m insertions is the tree of the first m keys -- later keys only add descendants;
//   * node v's split axis is fixed by the EARLIEST different key routed through it, and its child on
//     either side is the earliest key on that side.  So the whole tree can be grown level by level
//     from the complete key stream: every key that is not yet a node sits at some node v; per level
//     one pass finds, per node, the earliest waiting key (atomicMin of draw indices), a second pass the
//     earliest on each side; those become v's children, the rest step down.
//   * v stops being a leaf exactly when its first child appears, i.e. at draw index first[v].  Hence
//     leaf_count after draw m = #{nodes created <= m} - #{nodes whose first child was created <= m},
//     a prefix sum over per-draw deltas, and the stop index is the first m with m + 1 > limit[leaf(m)].
//
// Repeated keys are folded first (hash table on the packed 64-bit key, atomicMin of the draw index):
// only a key's first occurrence is a tree key, as in PFKDTree::insertNode (equal key: value += ...).
#pragma once
#include <climits>

#include "device_types.hpp"

namespace bpf
{

constexpr unsigned long long kKldEmpty = ~0ull;

struct KldArgs
{
  const int* keys;              // [3 * n] bin keys in draw order (AoS)
  int n;                        // draws in the stream
  unsigned long long* h_key;    // hash table: packed key per slot (kKldEmpty = free)
  int* h_tmin;                  // hash table: earliest draw index with that key
  unsigned h_mask;              // table size - 1 (power of two)
  int* slot;                    // [n] table slot of each draw
  int* cur;                     // [n] node (draw index of its key) a waiting key currently sits at; -1: is a node / duplicate
  int* first;                   // [n] per node: earliest key routed through it (INT_MAX: still a leaf)
  int* child;                   // [2 * n] per node: earliest key on the low / high side
  int2* delta;                  // [n] per draw: (leaf-count change, new-bin flag)
  int* flags;                   // [0] keys out of packing range, [2] stop index
  const int* limit;             // resampleLimit per leaf count, [0 .. n]
};

__device__ __forceinline__ bool kld_pack(const int* k, unsigned long long* out)
{
  const long long a = (long long)k[0] + (1 << 23), b = (long long)k[1] + (1 << 23), c = (long long)k[2] + (1 << 15);
  if (a < 0 || a >= (1 << 24) - 1 || b < 0 || b >= (1 << 24) || c < 0 || c >= (1 << 16))
    return false;
  *out = ((unsigned long long)a << 40) | ((unsigned long long)b << 16) | (unsigned long long)c;
  return true;
}

// atomicMin(base[a], i) for the threads of a block, combined in LDS first.  Near the root of the tree thousands
// of keys wait at the same few nodes, and same-address atomics serialise in L2 (~80 ns each): a block folds
// its keys per target element in an LDS hash table (tag = element index, value = min draw index) and sends one
// atomic per distinct element; a key that finds no table slot in two probes goes to memory directly.
constexpr int kKldBlock = 1024;
constexpr int kKldCombine = 2048;

__device__ __forceinline__ void block_atomic_min(int* base, unsigned a, int i, bool active, unsigned* s_tag, int* s_val)
{
  for (int s = threadIdx.x; s < kKldCombine; s += kKldBlock)
  {
    s_tag[s] = 0xFFFFFFFFu;
    s_val[s] = INT_MAX;
  }
  __syncthreads();
  if (active)
  {
    unsigned h = (a * 2654435761u) >> 21;  // 11 bits
    bool placed = false;
#pragma unroll
    for (int probe = 0; probe < 2 && !placed; ++probe)
    {
      const unsigned prev = atomicCAS(&s_tag[h], 0xFFFFFFFFu, a);
      if (prev == 0xFFFFFFFFu || prev == a)
      {
        atomicMin(&s_val[h], i);
        placed = true;
      }
      h = (h + 1) & (kKldCombine - 1);
    }
    if (!placed)
      atomicMin(&base[a], i);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < kKldCombine; s += kKldBlock)
    if (s_tag[s] != 0xFFFFFFFFu)
      atomicMin(&base[s_tag[s]], s_val[s]);
}

// the tree's tables in their start state, one launch instead of six memsets (~4.4 us each): hash keys all-ones, the
// int tables 0x7F7F7F7F ("later than any draw"), the flag words zero except the stop index
// tree_tables = 0: the tree is grown in LDS-sized pieces (kernels_kld2.hpp), first[] / child[] are not used
__global__ void k_kld_clear(const KldArgs A, unsigned table, int n_flags, int tree_tables = 1)
{
  const unsigned stride = gridDim.x * blockDim.x;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  for (unsigned i = t; i < table; i += stride)
  {
    A.h_key[i] = kKldEmpty;
    A.h_tmin[i] = 0x7F7F7F7F;
  }
  for (unsigned i = t; tree_tables && i < (unsigned)A.n; i += stride)
    A.first[i] = 0x7F7F7F7F;
  for (unsigned i = t; tree_tables && i < 2u * (unsigned)A.n; i += stride)
    A.child[i] = 0x7F7F7F7F;
  for (unsigned i = t; i < (unsigned)n_flags; i += stride)
    A.flags[i] = (i == 2u) ? 0x7F7F7F7F : 0;
}

// pass 1: fold repeated keys; tmin[slot] = first draw with that key.  The block folds its own repeats in LDS first: a
// converged set of 10^5 samples has ~50 distinct keys, and 2 000 same-address atomics per table slot serialise in L2
// (~80 ns each: the launch took 257 us); with one insertion per distinct key and block it takes what a spread set does.
constexpr int kKldHashLds = 512;

__device__ __forceinline__ int kld_hash_insert(const KldArgs& A, unsigned long long pk, int m)
{
  unsigned h = (unsigned)((pk * 0x9E3779B97F4A7C15ull) >> 32) & A.h_mask;
  for (;;)
  {
    const unsigned long long prev = atomicCAS(&A.h_key[h], kKldEmpty, pk);
    if (prev == kKldEmpty || prev == pk)
      break;
    h = (h + 1) & A.h_mask;
  }
  atomicMin(&A.h_tmin[h], m);
  return (int)h;
}

__global__ __launch_bounds__(256) void k_kld_hash(const KldArgs A)
{
  __shared__ unsigned long long s_key[kKldHashLds];
  __shared__ int s_min[kKldHashLds];
  __shared__ int s_gslot[kKldHashLds];
  for (int s = threadIdx.x; s < kKldHashLds; s += blockDim.x)
  {
    s_key[s] = kKldEmpty;
    s_min[s] = INT_MAX;
  }
  __syncthreads();
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = m < A.n;
  unsigned long long pk = kKldEmpty;
  const bool ok = valid && kld_pack(&A.keys[3 * (size_t)m], &pk);
  if (valid && !ok)
  {
    atomicExch(&A.flags[0], 1);
    A.slot[m] = 0;
  }
  int ls = -1;
  if (ok)
  {
    unsigned h = (unsigned)((pk * 0x9E3779B97F4A7C15ull) >> 44) & (kKldHashLds - 1);
    for (int probe = 0; probe < 8; ++probe)
    {
      const unsigned long long prev = atomicCAS(&s_key[h], kKldEmpty, pk);
      if (prev == kKldEmpty || prev == pk)
      {
        atomicMin(&s_min[h], m);
        ls = (int)h;
        break;
      }
      h = (h + 1) & (kKldHashLds - 1);
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < kKldHashLds; s += blockDim.x)
    if (s_key[s] != kKldEmpty)
      s_gslot[s] = kld_hash_insert(A, s_key[s], s_min[s]);
  int gs = 0;
  if (ok && ls < 0)
    gs = kld_hash_insert(A, pk, m);  // the block's table had no room within eight probes
  __syncthreads();
  if (ok)
    A.slot[m] = ls >= 0 ? s_gslot[ls] : gs;
}

// pass 2: tree keys = first occurrences; all of them wait at the root (draw 0) and report to it
__global__ void k_kld_init(const KldArgs A)
{
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= A.n)
    return;
  const bool is_first = A.h_tmin[A.slot[m]] == m;
  A.delta[m] = make_int2(is_first ? 1 : 0, is_first ? 1 : 0);
  A.cur[m] = (is_first && m != 0) ? 0 : -1;
}

// the earliest key that waits at the root (one block)
__global__ __launch_bounds__(1024) void k_kld_root_first(const KldArgs A)
{
  __shared__ int s_min[16];
  int best = INT_MAX;
  for (int m = threadIdx.x; m < A.n; m += 1024)
    if (A.cur[m] == 0)
    {
      best = m;  // the first hit of a strided walk is this thread's smallest
      break;
    }
  for (int o = 32; o > 0; o >>= 1)
    best = min(best, __shfl_xor(best, o, 64));
  if ((threadIdx.x & 63) == 0)
    s_min[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    for (int w = 1; w < 16; ++w)
      best = min(best, s_min[w]);
    if (best != INT_MAX)
      A.first[0] = best;
  }
}

__device__ __forceinline__ int kld_side(const int* keys, int v, int f, int i)
{
  // split axis of v: largest |delta| between v's key and the first different key through it, earliest
  // axis on ties (pf_kdtree.cpp:133-146); a key goes to the high side if it is greater on that axis
  const int* kv = &keys[3 * (size_t)v];
  const int* kf = &keys[3 * (size_t)f];
  int best = 0, pv = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d)
  {
    const int s = abs(kf[d] - kv[d]);
    if (s > best)
    {
      best = s;
      pv = d;
    }
  }
  return keys[3 * (size_t)i + pv] > kv[pv] ? 1 : 0;
}

// level, first half: the earliest waiting key on each side of every node
__global__ __launch_bounds__(kKldBlock) void k_kld_children(const KldArgs A)
{
  __shared__ unsigned s_tag[kKldCombine];
  __shared__ int s_val[kKldCombine];
  const int i = blockIdx.x * kKldBlock + threadIdx.x;
  const int v = (i < A.n) ? A.cur[i] : -1;
  const bool waiting = v >= 0;
  unsigned a = 0;
  if (waiting)
    a = 2u * (unsigned)v + (unsigned)kld_side(A.keys, v, A.first[v], i);
  block_atomic_min(A.child, a, i, waiting, s_tag, s_val);
}

// level, second half: that key becomes the child node; the others step down to it and report
// `waiting` is set when a key is still not a node after this level (one word per level)
__global__ __launch_bounds__(kKldBlock) void k_kld_descend(const KldArgs A, int* waiting)
{
  __shared__ unsigned s_tag[kKldCombine];
  __shared__ int s_val[kKldCombine];
  const int i = blockIdx.x * kKldBlock + threadIdx.x;
  const int v = (i < A.n) ? A.cur[i] : -1;
  bool still = false;
  int c = 0;
  if (v >= 0)
  {
    const int f = A.first[v];
    const int side = kld_side(A.keys, v, f, i);
    c = A.child[2 * (size_t)v + side];
    if (c == i)
    {
      A.cur[i] = -1;  // i is a node now
      if (f == i)
        A.delta[i].x = 0;  // its creation ends the parent's time as a leaf: +1 - 1
    }
    else
    {
      A.cur[i] = c;
      still = true;
    }
  }
  block_atomic_min(A.first, (unsigned)c, i, still, s_tag, s_val);
  // keys of this block that still wait after this level (the host looks for zero; BPF_DEBUG prints the counts)
  const int cnt = __syncthreads_count(still ? 1 : 0);
  if (threadIdx.x == 0 && cnt > 0)
    atomicAdd(waiting, cnt);
}

// inclusive scan of delta (int2) in tiles of 2048, three launches
constexpr int kKldTile = 2048;

__global__ __launch_bounds__(256) void k_kld_scan_tiles(const int2* __restrict__ delta, int n, int2* __restrict__ tile_sums)
{
  __shared__ int2 s_w[4];
  const int base = blockIdx.x * kKldTile;
  int2 acc = make_int2(0, 0);
  for (int j = threadIdx.x; j < kKldTile; j += 256)
    if (base + j < n)
    {
      const int2 d = delta[base + j];
      acc.x += d.x;
      acc.y += d.y;
    }
  for (int o = 32; o > 0; o >>= 1)
  {
    acc.x += __shfl_xor(acc.x, o, 64);
    acc.y += __shfl_xor(acc.y, o, 64);
  }
  if ((threadIdx.x & 63) == 0)
    s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    tile_sums[blockIdx.x] = make_int2(s_w[0].x + s_w[1].x + s_w[2].x + s_w[3].x, s_w[0].y + s_w[1].y + s_w[2].y + s_w[3].y);
}

// one block: exclusive scan of the tile sums in place
__global__ __launch_bounds__(1024) void k_kld_scan_offsets(int2* tile_sums, int tiles)
{
  __shared__ int2 s_part[1024];
  const int tid = threadIdx.x;
  const int per = (tiles + 1023) / 1024;
  const int lo = min(tid * per, tiles), hi = min(lo + per, tiles);
  int2 sum = make_int2(0, 0);
  for (int i = lo; i < hi; ++i)
  {
    sum.x += tile_sums[i].x;
    sum.y += tile_sums[i].y;
  }
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1)
  {
    const int2 v = (tid >= o) ? s_part[tid - o] : make_int2(0, 0);
    __syncthreads();
    s_part[tid].x += v.x;
    s_part[tid].y += v.y;
    __syncthreads();
  }
  int2 run = make_int2(s_part[tid].x - sum.x, s_part[tid].y - sum.y);
  for (int i = lo; i < hi; ++i)
  {
    const int2 t = tile_sums[i];
    tile_sums[i] = run;
    run.x += t.x;
    run.y += t.y;
  }
}

// per tile: inclusive scan -> (leaf count, bin count) after every draw, and the stop test
// (particle_filter.cpp:416: sample_count > resampleLimit(leaf_count), sample_count = m + 1)
__global__ __launch_bounds__(256) void k_kld_scan_final(const KldArgs A, const int2* __restrict__ tile_offsets,
                                                        int2* __restrict__ counts)
{
  __shared__ int2 s_w[4];
  constexpr int per = kKldTile / 256;
  const int base = blockIdx.x * kKldTile + threadIdx.x * per;
  int2 v[per];
  int2 sum = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < per; ++j)
  {
    v[j] = (base + j < A.n) ? A.delta[base + j] : make_int2(0, 0);
    sum.x += v[j].x;
    sum.y += v[j].y;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int2 incl = sum;
  for (int o = 1; o < 64; o <<= 1)
  {
    const int ux = __shfl_up(incl.x, o, 64), uy = __shfl_up(incl.y, o, 64);
    if (lane >= o)
    {
      incl.x += ux;
      incl.y += uy;
    }
  }
  if (lane == 63)
    s_w[wave] = incl;
  __syncthreads();
  int2 run = tile_offsets[blockIdx.x];
  for (int q = 0; q < wave; ++q)
  {
    run.x += s_w[q].x;
    run.y += s_w[q].y;
  }
  run.x += incl.x - sum.x;
  run.y += incl.y - sum.y;
  int stop = INT_MAX;
#pragma unroll
  for (int j = 0; j < per; ++j)
  {
    const int m = base + j;
    if (m < A.n)
    {
      run.x += v[j].x;
      run.y += v[j].y;
      counts[m] = run;
      if (m + 1 > A.limit[run.x] && stop == INT_MAX)
        stop = m + 1;
    }
  }
  if (stop != INT_MAX)
    atomicMin(&A.flags[2], stop);
}

// ---------------------------------------------------------------------------------------------------------------
// The same tree in ONE launch (opt-in, BPF_OPT_KLD_PERSISTENT): the level loop above is launch-bound (a spread cloud's
// tree is ~40 levels deep: 80+ launches of ~7 us for microseconds of work each), so when the whole stream fits one
// resident round of blocks the grid can stay on the chip with grid-wide barriers between the levels (every thread
// keeps its key, its current node and its delta in registers); the prefix sums over the deltas, the stop test and the
// counts at the stop follow in the same launch and the result goes to pinned host memory.  Exact like the other form
// (tests run both) but NOT faster on MI355X: 0.98 ms per step of the 100 k spread cloud against 0.83 ms.  A level is
// ~9 dependent round trips between XCDs (first[], the atomics' completion, barrier arrival, barrier release, child[],
// atomics, barrier, barrier, level word), each 1.5-2 us through the Infinity Cache, which is what a launch boundary
// costs as well; with fences by all threads instead of thread 0 a barrier took 45 us.
//
// Grid barrier: arrival counter + generation word (agent scope).  A wait is bounded: if the grid is NOT co-resident
// after all (another process holds the CUs) the waiting blocks give up, status BPF_KLD_PERSIST_TIMEOUT is published
// once and the host takes the launch-per-level path; every wave reaches its exit either way.
constexpr int BPF_KLD_PERSIST_OK = 0, BPF_KLD_PERSIST_KEY_RANGE = 1, BPF_KLD_PERSIST_TOO_DEEP = 2,
              BPF_KLD_PERSIST_TIMEOUT = 3;

struct KldPersistArgs
{
  KldArgs K;
  unsigned* bar;          // [0] arrivals, [1] generation, [2] "somebody gave up", [3] result published; zeroed before the launch
  int* level_waiting;     // [max_levels] zeroed: a key still waits after level l
  int2* tile_sums;        // [gridDim.x]
  int max_levels;
  int whole_stream;       // 1: the tree of all n keys, no stop rule
  long long timeout_ticks;  // bound of one barrier wait (100 MHz wall clock)
  volatile int* result_host;  // pinned: [1] stop (-1: none), [2] leaf count, [3] bin count, [4] status, [5] levels; [0] generation
  int generation;
};

// agent-scope load: served past this CU's L1 (what another CU's atomics wrote is not in it)
__device__ __forceinline__ int kld_ld(const int* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Everything the blocks exchange inside the launch goes through agent-scope atomics (atomicMin into first / child,
// atomic stores of the tile sums and level words, atomic loads on the reading side), so the barrier only has to order
// them: the block barrier waits for every wave's outstanding memory operations, then thread 0 arrives with a release,
// spins on the generation word and acquires.  (Fences by all 1024 threads cost ~45 us per barrier: 1 600 L2
// write-back / invalidate requests in flight.)
__device__ __forceinline__ bool kld_grid_sync(const KldPersistArgs& P, unsigned& epoch, int* s_flag)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's atomics and stores are out
  __syncthreads();
  if (threadIdx.x == 0)
  {
    ++epoch;
    int ok = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(&P.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1)
    {
      __hip_atomic_store(&P.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&P.bar[1], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    else
    {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(&P.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch)
      {
        if (__hip_atomic_load(&P.bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
            wall_clock64() - t0 > P.timeout_ticks)
        {
          __hip_atomic_store(&P.bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

__device__ __forceinline__ void kld_persist_publish(const KldPersistArgs& P, int stop, int leaf, int bins, int status,
                                                    int levels)
{
  // once per launch, by whoever gets here first (the thread that holds the answer, or the first block to give up)
  if (__hip_atomic_exchange(&P.bar[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
    return;
  volatile int* out = P.result_host;
  out[1] = stop;
  out[2] = leaf;
  out[3] = bins;
  out[4] = status;
  out[5] = levels;
  __threadfence_system();
  __hip_atomic_store(const_cast<int*>(out), P.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// grid = ceil(n / 1024) blocks of 1024 threads, all resident at once (the host checks); after k_kld_hash
__global__ __launch_bounds__(kKldBlock) void k_kld_tree_persistent(const KldPersistArgs P)
{
  __shared__ unsigned s_tag[kKldCombine];
  __shared__ int s_val[kKldCombine];
  __shared__ int s_flag;
  __shared__ int s_red[16];
  __shared__ int2 s_scan[16];
  const KldArgs& A = P.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * kKldBlock + tid;
  const bool in = i < A.n;
  unsigned epoch = 0;
#define KLD_SYNC_OR_QUIT()                                                       \
  do                                                                             \
  {                                                                              \
    if (!kld_grid_sync(P, epoch, &s_flag))                                       \
    {                                                                            \
      if (tid == 0)                                                              \
        kld_persist_publish(P, -1, 0, 0, BPF_KLD_PERSIST_TIMEOUT, 0);            \
      return;                                                                    \
    }                                                                            \
  } while (0)

  // ---- tree keys = first occurrences; all of them but draw 0 wait at the root and report to it
  const bool is_first = in && A.h_tmin[A.slot[i]] == i;  // (written by the launch before)
  int dx = is_first ? 1 : 0;
  const int dy = dx;
  int cur = (is_first && i != 0) ? 0 : -1;
  {
    int best = cur == 0 ? i : INT_MAX;
    for (int o = 32; o > 0; o >>= 1)
      best = min(best, __shfl_xor(best, o, 64));
    if (lane == 0)
      s_red[wave] = best;
    __syncthreads();
    if (tid == 0)
    {
      for (int w = 1; w < 16; ++w)
        best = min(best, s_red[w]);
      if (best != INT_MAX)
        atomicMin(&A.first[0], best);
    }
  }
  KLD_SYNC_OR_QUIT();
  if (__hip_atomic_load(&A.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
  {
    // k_kld_hash met a key outside the packing range: not handled here
    if (i == 0)
      kld_persist_publish(P, -1, 0, 0, BPF_KLD_PERSIST_KEY_RANGE, 0);
    return;
  }

  // ---- the levels
  int level = 0;
  bool done = false;
  for (; level < P.max_levels; ++level)
  {
    int f = 0, side = 0;
    const bool waiting = cur >= 0;
    unsigned a = 0;
    if (waiting)
    {
      f = kld_ld(&A.first[cur]);
      side = kld_side(A.keys, cur, f, i);
      a = 2u * (unsigned)cur + (unsigned)side;
    }
    block_atomic_min(A.child, a, i, waiting, s_tag, s_val);
    KLD_SYNC_OR_QUIT();
    bool still = false;
    int c = 0;
    if (waiting)
    {
      c = kld_ld(&A.child[a]);
      if (c == i)
      {
        cur = -1;  // i is a node now
        if (f == i)
          dx = 0;  // its creation ends the parent's time as a leaf: +1 - 1
      }
      else
      {
        cur = c;
        still = true;
      }
    }
    block_atomic_min(A.first, (unsigned)c, i, still, s_tag, s_val);
    if (__syncthreads_or(still ? 1 : 0) && tid == 0)
      __hip_atomic_store(&P.level_waiting[level], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    KLD_SYNC_OR_QUIT();
    if (__hip_atomic_load(&P.level_waiting[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
    {
      done = true;
      ++level;
      break;
    }
  }
  if (!done)
  {
    if (i == 0)
      kld_persist_publish(P, -1, 0, 0, BPF_KLD_PERSIST_TOO_DEEP, level);
    return;
  }

  // ---- (leaf count, bin count) after every draw: inclusive scan of the deltas, one 1024-tile per block
  int2 incl = make_int2(dx, dy);
  for (int o = 1; o < 64; o <<= 1)
  {
    const int ux = __shfl_up(incl.x, o, 64), uy = __shfl_up(incl.y, o, 64);
    if (lane >= o)
    {
      incl.x += ux;
      incl.y += uy;
    }
  }
  if (lane == 63)
    s_scan[wave] = incl;
  __syncthreads();
  int2 run = incl;
  for (int q = 0; q < wave; ++q)
  {
    run.x += s_scan[q].x;
    run.y += s_scan[q].y;
  }
  if (tid == kKldBlock - 1)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&P.tile_sums[blockIdx.x]),
                       ((unsigned long long)(unsigned)run.y << 32) | (unsigned)run.x, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  KLD_SYNC_OR_QUIT();
  {
    // the tiles before this one (a few hundred at most): strided partial sums, then the block's total
    int2 acc = make_int2(0, 0);
    for (int t = tid; t < (int)blockIdx.x; t += kKldBlock)
    {
      const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&P.tile_sums[t]),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc.x += (int)(unsigned)v;
      acc.y += (int)(unsigned)(v >> 32);
    }
    for (int o = 32; o > 0; o >>= 1)
    {
      acc.x += __shfl_xor(acc.x, o, 64);
      acc.y += __shfl_xor(acc.y, o, 64);
    }
    __syncthreads();  // s_scan is read above
    if (lane == 0)
      s_scan[wave] = acc;
    __syncthreads();
    for (int q = 0; q < 16; ++q)
    {
      run.x += s_scan[q].x;
      run.y += s_scan[q].y;
    }
  }
  // the stop test (particle_filter.cpp:416: sample_count > resampleLimit(leaf_count), sample_count = i + 1)
  if (!P.whole_stream)
  {
    int stop = (in && i + 1 > A.limit[run.x]) ? i + 1 : INT_MAX;
    for (int o = 32; o > 0; o >>= 1)
      stop = min(stop, __shfl_xor(stop, o, 64));
    if (lane == 0)
      s_red[wave] = stop;
    __syncthreads();
    if (tid == 0)
    {
      for (int w = 1; w < 16; ++w)
        stop = min(stop, s_red[w]);
      if (stop != INT_MAX)
        atomicMin(&A.flags[2], stop);
    }
  }
  KLD_SYNC_OR_QUIT();
  const int stop = P.whole_stream ? INT_MAX : __hip_atomic_load(&A.flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool stopped = stop >= 1 && stop <= A.n;
  const int M = stopped ? stop : A.n;
  if (i == M - 1)
    kld_persist_publish(P, stopped ? stop : -1, run.x, run.y, BPF_KLD_PERSIST_OK, level);
#undef KLD_SYNC_OR_QUIT
}

}  // namespace bpf
