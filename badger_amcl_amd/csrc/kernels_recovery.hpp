// Augmented-MCL recovery draws of resampleMultinomial (particle_filter.cpp:383-400, w_diff > 0):
//
//     if (drand48() < w_diff)  pose = random_pose_fn_();          // Node::randomFreeSpacePose: 2 more draws
//     else                   { r = drand48(); ... CDF search }    // 1 more draw
//
// Every candidate draw starts with a test on the next stream element and then consumes 2 or 1 further
// elements, so where draw m finds its elements depends on the outcome of every earlier test -- a pointer chase
// through the stream (next(q) = q + 3 or q + 2).  It is resolved in three passes over fixed segments of the
// stream: per segment and per possible entry offset (0, 1, 2) the number of draws that start inside it and the
// offset at which the walk leaves it; a scan that composes those little maps gives every segment its real
// entry offset and first draw index; a last pass walks each segment once more and records, per draw, the
// position of its test element and whether it is a random pose.
#pragma once
#include "device_types.hpp"
#include "kernels_pf.hpp"

namespace bpf
{

constexpr int kChainSeg = 128;  // stream positions per segment

struct ChainArgs
{
  uint64_t rng_state;  // drand48 state before the resample; stream position q (1-based) = q steps from it
  double w_diff;
  int n_seg;
  int max_draws;       // chain[] is filled for draws 0 .. max_draws (inclusive: where the stream would go on)
  uint64_t* seg_bits;  // [n_seg][2] outcome bit per position (1: r < w_diff)
  int* seg_cnt;        // [n_seg][3] draws that start in the segment, per entry offset
  int* seg_exit;       // [n_seg][3] offset into the next segment at which the walk leaves, per entry offset
  int* seg_entry;      // [n_seg] real entry offset
  int* seg_base;       // [n_seg] index of the first draw that starts in the segment
  int* chain;          // [max_draws + 1] position of draw m's test element, bit 31 = random pose
  LcgJump jump;
};

__device__ __forceinline__ bool chain_bit(const uint64_t b[2], int pos)
{
  return (b[pos >> 6] >> (pos & 63)) & 1ull;
}

__global__ void k_chain_segments(const ChainArgs A)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= A.n_seg)
    return;
  // outcomes of the tests at positions 1 + s*S + [0, S)
  uint64_t x = lcg_skip(A.rng_state, (uint64_t)s * kChainSeg + 1, A.jump);
  uint64_t b[2] = { 0, 0 };
  for (int p = 0; p < kChainSeg; ++p)
  {
    if (ldexp((double)x, -48) < A.w_diff)
      b[p >> 6] |= 1ull << (p & 63);
    x = lcg_next(x);
  }
  A.seg_bits[2 * (size_t)s] = b[0];
  A.seg_bits[2 * (size_t)s + 1] = b[1];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    int pos = a, cnt = 0;
    while (pos < kChainSeg)
    {
      ++cnt;
      pos += chain_bit(b, pos) ? 3 : 2;
    }
    A.seg_cnt[3 * (size_t)s + a] = cnt;
    A.seg_exit[3 * (size_t)s + a] = pos - kChainSeg;
  }
}

// a map {0,1,2} -> (draws, exit offset); (f then g)[a] = (f.cnt[a] + g.cnt[f.exit[a]], g.exit[f.exit[a]])
struct ChainMap
{
  int cnt[3];
  int exit[3];
};

__device__ __forceinline__ ChainMap chain_compose(const ChainMap& f, const ChainMap& g)
{
  ChainMap h;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    const int e = f.exit[a];
    h.cnt[a] = f.cnt[a] + g.cnt[e];
    h.exit[a] = g.exit[e];
  }
  return h;
}

__global__ __launch_bounds__(1024) void k_chain_scan(const ChainArgs A)
{
  __shared__ ChainMap s_map[1024];
  const int tid = threadIdx.x;
  const int per = (A.n_seg + 1023) / 1024;
  const int lo = min(tid * per, A.n_seg), hi = min(lo + per, A.n_seg);
  ChainMap mine;
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    mine.cnt[a] = 0;
    mine.exit[a] = a;  // identity
  }
  for (int s = lo; s < hi; ++s)
  {
    ChainMap g;
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
      g.cnt[a] = A.seg_cnt[3 * (size_t)s + a];
      g.exit[a] = A.seg_exit[3 * (size_t)s + a];
    }
    mine = chain_compose(mine, g);
  }
  s_map[tid] = mine;
  __syncthreads();
  // inclusive scan under composition (order matters: earlier chunks first)
  for (int o = 1; o < 1024; o <<= 1)
  {
    ChainMap left;
    const bool have = tid >= o;
    if (have)
      left = s_map[tid - o];
    __syncthreads();
    if (have)
      s_map[tid] = chain_compose(left, s_map[tid]);
    __syncthreads();
  }
  // the walk enters the first segment at offset 0 with draw 0; this thread's chunk starts where the
  // composition of all earlier chunks leaves it
  int entry = 0, base = 0;
  if (tid > 0)
  {
    entry = s_map[tid - 1].exit[0];
    base = s_map[tid - 1].cnt[0];
  }
  for (int s = lo; s < hi; ++s)
  {
    A.seg_entry[s] = entry;
    A.seg_base[s] = base;
    base += A.seg_cnt[3 * (size_t)s + entry];
    entry = A.seg_exit[3 * (size_t)s + entry];
  }
}

__global__ void k_chain_emit(const ChainArgs A)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= A.n_seg)
    return;
  const uint64_t b[2] = { A.seg_bits[2 * (size_t)s], A.seg_bits[2 * (size_t)s + 1] };
  int pos = A.seg_entry[s], m = A.seg_base[s];
  while (pos < kChainSeg && m <= A.max_draws)
  {
    const bool random = chain_bit(b, pos);
    A.chain[m] = (int)((unsigned)(s * kChainSeg + pos + 1) | (random ? 0x80000000u : 0u));
    ++m;
    pos += random ? 3 : 2;
  }
}

}  // namespace bpf
