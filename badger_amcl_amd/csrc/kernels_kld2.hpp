// The histogram tree of a long draw stream (kernels_kld.hpp) WITHOUT the level-per-launch loop.
//
// The level-synchronous form spends ~14 us per level on a launch pair and a spread cloud's tree is ~40 levels deep
// (0.6 ms for 10^5 keys), because every level is a handful of dependent trips to first[] / child[] in device memory.
// Two facts about the tree (kernels_kld.hpp, top) let LDS-sized pieces of it be grown by single blocks instead:
//
//   * the tree after m insertions is the tree of the first m keys: the first K tree keys of the stream form the TOP
//     of the final tree ("T0"), whatever comes later;
//   * a later key is routed through T0 by T0's nodes alone until it reaches a node whose child slot on its side is
//     still empty (or a T0 leaf): all the keys that end at the same T0 node v form the subtree below v, and subtrees
//     below different nodes never see each other.
//
// So:  A  one block grows T0 from the first K = 2048 tree keys entirely in LDS (~35 levels at ~1 us);
//      B  every later key walks down an LDS copy of T0 to its node ("bucket"), buckets are counted, offset and filled;
//      C  blocks take runs of whole buckets (about 600 keys) and grow their subtrees in LDS, all at once.
// What the stop rule needs from the tree is one bit per key -- was it the first key routed through its parent (then
// its creation ends the parent's time as a leaf: leaf-count delta 0 instead of +1) -- which A and C write into
// delta[], and the prefix sums of kernels_kld.hpp finish as before.  Exactly the same tree: the same "earliest key"
// minima, only taken in LDS.  A stream whose buckets do not fit (a block's run of buckets with more than ~1 500 keys:
// a cloud that is neither spread nor converged) raises a status word and the host runs the level loop instead.
#pragma once
#include "kernels_kld.hpp"

namespace bpf
{

constexpr int kKld2Block = 1024;
constexpr int kKld2Nodes = 2048;       // node table of one block (44 B per node: 88 KB of LDS)
constexpr int kKld2Top = 2048;         // keys of T0 when the stream has more than kKld2Nodes tree keys
constexpr int kKld2Span = 512;         // bucket offsets per block of phase C (keys + roots of a block must fit the table)
constexpr int kKld2MaxIter = 255;
constexpr unsigned long long kKld2None = ~0ull;
constexpr int BPF_KLD2_OK = 0, BPF_KLD2_BUCKET_TOO_LARGE = 1, BPF_KLD2_TOO_DEEP = 2;

struct __align__(16) Kld2Top  // one T0 node as phase B reads it (one 16-byte LDS read per step of the walk)
{
  unsigned long long key;  // packed bin key (kld_pack)
  short child[2];          // T0 rank of the child on the low / high side, -1: empty
  int axis;                // split axis, -1: the node has no child yet
};

struct Kld2Args
{
  KldArgs K;
  const int* tile_first;  // [blocks of k_kld_init] first occurrences per 256 draws
  int n_tiles;
  int* tkeys;      // [n] draw indices of the tree keys (first occurrences) in draw order
  unsigned long long* pk;  // [n] their packed keys, same order
  int* n_tkeys;    // [1]
  Kld2Top* top;    // [kKld2Nodes]
  int* n_top;      // [1] nodes of T0
  int* bucket;     // [n] per tkeys position >= n_top: T0 rank of the node the key ends at
  int* cnt;        // [kKld2Nodes] keys per bucket
  int* off;        // [kKld2Nodes + 1] exclusive prefix of cnt
  int* fill;       // [kKld2Nodes]
  int* bk;         // [n] tkeys positions grouped by bucket
  int* status;     // [0] BPF_KLD2_*, [1] largest bucket
  // result for the host (k_kld2_result): eight 64-bit pinned words, (generation << 32) | value in words 1 .. 7
  volatile int* result_host;
  int generation;
  int whole_stream;
  const int2* counts;
};

__device__ __forceinline__ int kld2_field(unsigned long long pk, int axis)
{
  return axis == 0 ? (int)(pk >> 40) : (axis == 1 ? (int)((pk >> 16) & 0xFFFFFFull) : (int)(pk & 0xFFFFull));
}

// split axis of a node with key kv whose first different key is kf (pf_kdtree.cpp:133-146): largest |delta|, the
// earliest axis on ties.  The packing adds a constant per field, so differences are those of the raw keys.
__device__ __forceinline__ int kld2_axis(unsigned long long kv, unsigned long long kf)
{
  int best = 0, pv = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d)
  {
    const int s = abs(kld2_field(kf, d) - kld2_field(kv, d));
    if (s > best)
    {
      best = s;
      pv = d;
    }
  }
  return pv;
}

// k_kld_init with the count of first occurrences per block of 256 draws (what the compaction below starts from)
__global__ __launch_bounds__(256) void k_kld2_init(const KldArgs A, int* __restrict__ tile_first)
{
  const int m = blockIdx.x * 256 + threadIdx.x;
  bool is_first = false;
  if (m < A.n)
  {
    is_first = A.h_tmin[A.slot[m]] == m;
    A.delta[m] = make_int2(is_first ? 1 : 0, is_first ? 1 : 0);
    A.cur[m] = (is_first && m != 0) ? 0 : -1;
  }
  const int c = __syncthreads_count(is_first ? 1 : 0);
  if (threadIdx.x == 0)
    tile_first[blockIdx.x] = c;
}

// first occurrences in draw order: tkeys[], their count (block b writes the tree keys of draws [256 b, 256 b + 256));
// block 0 also clears the bucket tables
__global__ __launch_bounds__(256) void k_kld2_compact(const Kld2Args A)
{
  __shared__ int s_w[4];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int before = 0;
  for (int t = tid; t < (int)blockIdx.x; t += 256)
    before += A.tile_first[t];
  for (int o = 32; o > 0; o >>= 1)
    before += __shfl_xor(before, o, 64);
  if (lane == 0)
    s_w[wave] = before;
  __syncthreads();
  if (tid == 0)
    s_base = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
  __syncthreads();
  const int m = blockIdx.x * 256 + tid;
  const bool f = m < A.K.n && A.K.delta[m].y != 0;
  const unsigned long long ball = __ballot(f);
  const int in_wave = __popcll(ball & ((1ull << lane) - 1ull));
  if (lane == 0)
    s_w[wave] = __popcll(ball);
  __syncthreads();
  int wave_base = 0;
  for (int w = 0; w < wave; ++w)
    wave_base += s_w[w];
  if (f)
  {
    A.tkeys[s_base + wave_base + in_wave] = m;
    A.pk[s_base + wave_base + in_wave] = A.K.h_key[A.K.slot[m]];
  }
  if ((int)blockIdx.x == A.n_tiles - 1 && tid == 0)
    *A.n_tkeys = s_base + s_w[0] + s_w[1] + s_w[2] + s_w[3];
  if (blockIdx.x == 0)
  {
    for (int i = tid; i < kKld2Nodes; i += 256)
    {
      A.cnt[i] = 0;
      A.fill[i] = 0;
    }
    if (tid == 0)
    {
      A.status[0] = BPF_KLD2_OK;
      A.status[1] = 0;
    }
  }
}

// The LDS node table of one block.  Local ids: whatever the caller lays out; a key's `cur` is the local id of the node
// it waits at (-1: it is a node itself, or the entry is a root).  first / child hold (draw index << 32) | local id, so
// the minimum over draw indices brings the winner's local id along.
struct Kld2Lds
{
  unsigned long long* key;    // [kKld2Nodes]
  unsigned long long* first;  // earliest key waiting at the node (while its axis is unknown)
  unsigned long long* child;  // [2 * kKld2Nodes] earliest key per side
  int* idx;                   // draw index
  int* cur;
  int* axis;                  // -1 until the node's first different key is known
};

__device__ __forceinline__ Kld2Lds kld2_lds(unsigned char* smem)
{
  Kld2Lds L;
  L.key = reinterpret_cast<unsigned long long*>(smem);
  L.first = L.key + kKld2Nodes;
  L.child = L.first + kKld2Nodes;
  L.idx = reinterpret_cast<int*>(L.child + 2 * kKld2Nodes);
  L.cur = L.idx + kKld2Nodes;
  L.axis = L.cur + kKld2Nodes;
  return L;
}
constexpr size_t kKld2LdsBytes = (size_t)kKld2Nodes * (8 + 8 + 16 + 4 + 4 + 4);

__device__ __forceinline__ unsigned long long kld2_pair(int idx, int loc)
{
  return ((unsigned long long)(unsigned)idx << 32) | (unsigned)loc;
}

// Grows the subtrees of all the keys in the table (entries [0, n_entries); keys are those with cur >= 0) level by
// level in LDS: two block barriers per level.  delta[]: a key that is the first one routed through its parent gets
// leaf-count delta 0.  Returns false when it does not finish within kKld2MaxIter levels.
// A thread's own keys (key, draw index, current node) live in registers for the whole build, and a level's LDS reads
// are issued for all of a thread's keys before anything depends on them: a level is then three dependent LDS round
// trips (node record, first key's key, the minimum) + two (the child, the report) instead of twice as many.
// __syncthreads() also waits for the block's outstanding GLOBAL stores; the levels only exchange through LDS.
__device__ __forceinline__ void kld2_lds_barrier()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// s_more: three words of LDS, zero on entry (the levels' "anyone still waiting" flags, used in rotation: a level's
// flag is cleared two levels before it is set and read one barrier after).
__device__ __forceinline__ bool kld2_grow(const Kld2Lds& L, int n_entries, int2* __restrict__ delta, int* s_more)
{
  constexpr int kPer = kKld2Nodes / kKld2Block;
  const int tid = threadIdx.x;
  unsigned long long kj[kPer];
  int ij[kPer], cj[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q)
  {
    const int j = tid + q * kKld2Block;
    const bool in = j < n_entries;
    kj[q] = in ? L.key[j] : 0ull;
    ij[q] = in ? L.idx[j] : 0;
    cj[q] = in ? L.cur[j] : -1;
    // a key that arrives at a node whose axis is unknown reports to it (the earliest becomes the node's first key)
    if (cj[q] >= 0 && L.axis[cj[q]] < 0)
      atomicMin(&L.first[cj[q]], kld2_pair(ij[q], j));
  }
  bool is_first[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q)
    is_first[q] = false;
  bool done = false;
  kld2_lds_barrier();
  for (int iter = 0, r = 0; iter < kKld2MaxIter; ++iter, r = r == 2 ? 0 : r + 1)
  {
    // the node's axis from its first key (every waiting key works it out for itself; the first key's creation ends
    // the node's time as a leaf), then the earliest waiting key on each side of the node
    int slot[kPer], ax[kPer], fl[kPer];
    unsigned long long kc[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q)
    {
      ax[q] = 0;
      fl[q] = 0;
      kc[q] = 0ull;
      if (cj[q] >= 0)
      {
        ax[q] = L.axis[cj[q]];
        fl[q] = (int)(unsigned)(L.first[cj[q]] & 0xFFFFFFFFull);
        kc[q] = L.key[cj[q]];
      }
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q)
    {
      slot[q] = -1;
      if (cj[q] >= 0)
      {
        const int j = tid + q * kKld2Block;
        if (ax[q] < 0)
        {
          ax[q] = kld2_axis(kc[q], L.key[fl[q]]);
          is_first[q] = is_first[q] || fl[q] == j;
        }
        slot[q] = 2 * cj[q] + (kld2_field(kj[q], ax[q]) > kld2_field(kc[q], ax[q]) ? 1 : 0);
        atomicMin(&L.child[slot[q]], kld2_pair(ij[q], j));
      }
    }
    kld2_lds_barrier();
    // that key is the child node now; the others step down to it and report to it
    int still = 0;
    int ch[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q)
      ch[q] = slot[q] >= 0 ? (int)(unsigned)(L.child[slot[q]] & 0xFFFFFFFFull) : -1;
#pragma unroll
    for (int q = 0; q < kPer; ++q)
      if (slot[q] >= 0)
      {
        const int j = tid + q * kKld2Block;
        if (ch[q] == j)
          cj[q] = -1;
        else
        {
          cj[q] = ch[q];
          atomicMin(&L.first[ch[q]], kld2_pair(ij[q], j));
          still = 1;
        }
      }
    if (tid == 0)
      s_more[r == 2 ? 0 : r + 1] = 0;
    if (__any(still) && (tid & 63) == 0)
      s_more[r] = 1;
    kld2_lds_barrier();
    if (s_more[r] == 0)
    {
      done = true;
      break;
    }
  }
#pragma unroll
  for (int q = 0; q < kPer; ++q)
    if (is_first[q])
      delta[ij[q]].x = 0;
  return done;
}

// phase A: T0 from the first n_top tree keys (all of them when the stream has at most kKld2Nodes), one block
__global__ __launch_bounds__(kKld2Block) void k_kld2_top(const Kld2Args A)
{
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int s_more[3];
  const Kld2Lds L = kld2_lds(smem);
  const int tid = threadIdx.x;
  if (tid < 3)
    s_more[tid] = 0;
  const int n_t = *A.n_tkeys;
  const int n_top = n_t <= kKld2Nodes ? n_t : kKld2Top;
  for (int j = tid; j < n_top; j += kKld2Block)
  {
    L.key[j] = A.pk[j];
    L.idx[j] = A.tkeys[j];
    L.cur[j] = j == 0 ? -1 : 0;  // every tree key waits at the root, draw 0
    L.axis[j] = -1;
    L.first[j] = kKld2None;
    L.child[2 * j] = kKld2None;
    L.child[2 * j + 1] = kKld2None;
  }
  __syncthreads();
  const bool ok = n_top <= 1 || kld2_grow(L, n_top, A.K.delta, s_more);
  if (!ok && tid == 0)
    A.status[0] = BPF_KLD2_TOO_DEEP;
  for (int j = tid; j < n_top; j += kKld2Block)
  {
    Kld2Top t;
    t.key = L.key[j];
    // the axis of a node with children follows from its first key (the keys worked it out on the fly)
    t.axis = L.first[j] == kKld2None ? -1 : kld2_axis(L.key[j], L.key[(int)(unsigned)(L.first[j] & 0xFFFFFFFFull)]);
    t.child[0] = L.child[2 * j] == kKld2None ? (short)-1 : (short)(L.child[2 * j] & 0xFFFFull);
    t.child[1] = L.child[2 * j + 1] == kKld2None ? (short)-1 : (short)(L.child[2 * j + 1] & 0xFFFFull);
    A.top[j] = t;
  }
  if (tid == 0)
    *A.n_top = n_top;
}

// phase B: every later tree key walks down T0 (copied to LDS) to the node it ends at
__global__ __launch_bounds__(kKld2Block) void k_kld2_route(const Kld2Args A)
{
  __shared__ Kld2Top s_top[kKld2Top];
  __shared__ int s_cnt[kKld2Top];
  const int n_t = *A.n_tkeys, n_top = *A.n_top;
  if (n_top >= n_t || (int)(blockIdx.x * kKld2Block) >= n_t - n_top)
    return;
  for (int j = threadIdx.x; j < n_top; j += kKld2Block)
  {
    s_top[j] = A.top[j];
    s_cnt[j] = 0;
  }
  __syncthreads();
  const int p = n_top + blockIdx.x * kKld2Block + threadIdx.x;
  if (p < n_t)
  {
    const unsigned long long pk = A.pk[p];
    int v = 0;
    for (;;)
    {
      const Kld2Top t = s_top[v];
      if (t.axis < 0)
        break;
      const int c = t.child[kld2_field(pk, t.axis) > kld2_field(t.key, t.axis) ? 1 : 0];
      if (c < 0)
        break;
      v = c;
    }
    A.bucket[p] = v;
    atomicAdd(&s_cnt[v], 1);  // the block's keys per bucket in LDS first: one global atomic per bucket and block
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n_top; j += kKld2Block)
    if (s_cnt[j] != 0)
      atomicAdd(&A.cnt[j], s_cnt[j]);
}

// bucket offsets (one block): off[k] = keys in the buckets before k; the largest bucket decides whether phase C fits
__global__ __launch_bounds__(1024) void k_kld2_offsets(const Kld2Args A)
{
  __shared__ int s_part[1024];
  __shared__ int s_max[16];
  const int tid = threadIdx.x;
  constexpr int per = kKld2Nodes / 1024;
  int v[per], sum = 0, mx = 0;
#pragma unroll
  for (int q = 0; q < per; ++q)
  {
    v[q] = A.cnt[tid * per + q];
    sum += v[q];
    mx = max(mx, v[q]);
  }
  s_part[tid] = sum;
  for (int o = 32; o > 0; o >>= 1)
    mx = max(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0)
    s_max[tid >> 6] = mx;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1)
  {
    const int u = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += u;
    __syncthreads();
  }
  int run = s_part[tid] - sum;
#pragma unroll
  for (int q = 0; q < per; ++q)
  {
    A.off[tid * per + q] = run;
    run += v[q];
  }
  if (tid == 1023)
    A.off[kKld2Nodes] = run;
  if (tid == 0)
  {
    int m = 0;
    for (int w = 0; w < 16; ++w)
      m = max(m, s_max[w]);
    A.status[1] = m;
    if (m > kKld2Nodes - 2)
      A.status[0] = BPF_KLD2_BUCKET_TOO_LARGE;
  }
}

__global__ __launch_bounds__(256) void k_kld2_scatter(const Kld2Args A)
{
  const int n_t = *A.n_tkeys, n_top = *A.n_top;
  const int p = n_top + blockIdx.x * 256 + threadIdx.x;
  if (p >= n_t || A.status[0] != BPF_KLD2_OK)
    return;
  const int v = A.bucket[p];
  A.bk[A.off[v] + atomicAdd(&A.fill[v], 1)] = p;
}

// phase C: block b grows the subtrees of the buckets whose offset lies in [b, b + 1) * kKld2Span.
// Local ids: the key at position lo + j is entry j, the root of bucket k entry nk + (k - k0).
__global__ __launch_bounds__(kKld2Block) void k_kld2_subtrees(const Kld2Args A)
{
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int s_range[2];
  __shared__ int s_more[3];
  const int tid = threadIdx.x;
  if (A.status[0] != BPF_KLD2_OK)
    return;
  if (tid >= 64 && tid < 67)
    s_more[tid - 64] = 0;
  const int n_t = *A.n_tkeys, n_top = *A.n_top, total = n_t - n_top;
  const int w_lo = blockIdx.x * kKld2Span, w_hi = w_lo + kKld2Span;
  if (total <= 0 || w_lo >= total)
    return;
  if (tid < 2)
  {
    // first bucket whose offset is >= the window's edge (off[] is non-decreasing; empty buckets share an offset)
    const int edge = tid == 0 ? w_lo : w_hi;
    int a = 0, b = n_top;  // off[n_top .. kKld2Nodes] == total
    while (a < b)
    {
      const int mid = (a + b) >> 1;
      if (A.off[mid] < edge)
        a = mid + 1;
      else
        b = mid;
    }
    s_range[tid] = a;
  }
  __syncthreads();
  const int k0 = s_range[0], k1 = s_range[1];
  const int lo = A.off[k0], hi = A.off[k1];
  const int nk = hi - lo, nr = k1 - k0;
  if (nk <= 0)
    return;
  if (nk + nr > kKld2Nodes)
  {
    if (tid == 0)
      A.status[0] = BPF_KLD2_BUCKET_TOO_LARGE;  // this run of buckets does not fit one block's table
    return;
  }
  const Kld2Lds L = kld2_lds(smem);
  for (int j = tid; j < nk + nr; j += kKld2Block)
  {
    L.first[j] = kKld2None;
    L.child[2 * j] = kKld2None;
    L.child[2 * j + 1] = kKld2None;
    if (j < nk)
    {
      const int p = A.bk[lo + j];
      L.key[j] = A.pk[p];
      L.idx[j] = A.tkeys[p];
      L.cur[j] = nk + (A.bucket[p] - k0);
      L.axis[j] = -1;
    }
    else
    {
      const Kld2Top t = A.top[k0 + (j - nk)];
      L.key[j] = t.key;
      L.idx[j] = -1;
      L.cur[j] = -1;
      L.axis[j] = t.axis;
    }
  }
  __syncthreads();
  if (!kld2_grow(L, nk, A.K.delta, s_more) && tid == 0)
    A.status[0] = BPF_KLD2_TOO_DEEP;
}

// The prefix sums over delta[] (leaf count and bin count after every draw) and the stop test in ONE launch instead
// of three (tile sums, offsets, final): every block adds up its tile, publishes the pair behind the launch's
// generation in one 64-bit word -- (generation << 40) | (leaves << 20) | bins, the value is its own flag, nothing to
// reset -- and reads the tiles before it (blocks are dispatched in index order: they are resident or done; the wait
// is bounded all the same).  Tiles of kKldTile draws as in k_kld_scan_final, same counts[] and the same stop word.
__global__ __launch_bounds__(256) void k_kld2_scan(const KldArgs A, unsigned long long* __restrict__ slots,
                                                  unsigned generation, int2* __restrict__ counts, int* status)
{
  __shared__ int2 s_w[4];
  __shared__ int2 s_off;
  constexpr int per = kKldTile / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int base = b * kKldTile + tid * per;
  int2 v[per];
  int2 sum = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < per; ++j)
  {
    v[j] = (base + j < A.n) ? A.delta[base + j] : make_int2(0, 0);
    sum.x += v[j].x;
    sum.y += v[j].y;
  }
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
  const unsigned long long tag = (unsigned long long)(generation % 0xFFFFFFu + 1u) << 40;  // never the zeroed slot's
  if (tid == 0)
  {
    const int tx = s_w[0].x + s_w[1].x + s_w[2].x + s_w[3].x, ty = s_w[0].y + s_w[1].y + s_w[2].y + s_w[3].y;
    __hip_atomic_store(&slots[b], tag | ((unsigned long long)(unsigned)tx << 20) | (unsigned long long)(unsigned)ty,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the tiles before this one: thread t fetches tile t (strided for more than 256 tiles), the waves add up
  int2 before = make_int2(0, 0);
  for (int t = tid; t < b; t += 256)
  {
    unsigned long long w;
    long long t0 = 0;
    for (unsigned spins = 0; ((w = __hip_atomic_load(&slots[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 40) !=
                             (tag >> 40); ++spins)
    {
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 255u) == 255u)
      {
        const long long now = wall_clock64();
        if (t0 == 0)
          t0 = now;
        else if (now - t0 > 5000000ll)  // 50 ms: the tile counts as empty, the status word says so
        {
          atomicExch(status, BPF_KLD2_TOO_DEEP);
          w = tag;
          break;
        }
      }
    }
    before.x += (int)((w >> 20) & 0xFFFFFull);
    before.y += (int)(w & 0xFFFFFull);
  }
  for (int o = 32; o > 0; o >>= 1)
  {
    before.x += __shfl_xor(before.x, o, 64);
    before.y += __shfl_xor(before.y, o, 64);
  }
  __shared__ int2 s_b[4];
  if (lane == 0)
    s_b[wave] = before;
  __syncthreads();
  if (tid == 0)
    s_off = make_int2(s_b[0].x + s_b[1].x + s_b[2].x + s_b[3].x, s_b[0].y + s_b[1].y + s_b[2].y + s_b[3].y);
  __syncthreads();
  int2 run = s_off;
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

// what the host wants to know, in one pinned block behind a generation word: [1] key outside the packing, [2] stop
// index (flags[2]), [3] leaf count and [4] bin count at the stop (or at n), [5] status, [6] tree keys, [7] largest bucket
__global__ void k_kld2_result(const Kld2Args A)
{
  if (threadIdx.x != 0)
    return;
  const int n = A.K.n;
  const int stop = A.whole_stream ? -1 : A.K.flags[2];
  const int M = (stop >= 1 && stop <= n) ? stop : n;
  const int2 c = A.counts[M - 1];
  // Seven 64-bit words in pinned memory, each (generation << 32) | value, each a system-scope store of its own -- as
  // fused_publish does it.  (Plain stores followed by a flag word are NOT enough: they may stay in this XCD's L2 --
  // the default pinned allocation is coarse-grained -- while the flag, written through, is already in host memory; the
  // host then reads the PREVIOUS build's stop / leaf count.  A build over 60 000 draws followed by one over 100 showed
  // it as a sample count beyond the new filter's buffers.)
  unsigned long long* out = reinterpret_cast<unsigned long long*>(const_cast<int*>(A.result_host));
  const unsigned long long g = (unsigned long long)(unsigned)A.generation << 32;
  const int v[8] = { 0, A.K.flags[0], (stop >= 1 && stop <= n) ? stop : -1, c.x, c.y, A.status[0], *A.n_tkeys,
                     A.status[1] };
#pragma unroll
  for (int k = 1; k < 8; ++k)
    __hip_atomic_store(&out[k], g | (unsigned long long)(unsigned)v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace bpf
