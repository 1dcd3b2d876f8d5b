// Exchange between the engines of one node without a collective library ("mailbox").
//
// The sharded path has two exchanges per update, both tiny: W weight totals (8 B each) and one draw window
// in which every column has exactly one writer.  Neither is a reduction, so every rank can simply store its
// contribution into every peer's memory over xGMI: each engine owns one uncached device allocation, exports
// it through an IPC handle, and maps the W - 1 others.  A producer kernel writes its values into the same
// slot of every peer's mailbox and then a generation word with system-scope release; a consumer kernel spins
// (bounded) on the generation words in its OWN mailbox with system-scope acquire.  No host round trip, no
// extra launch: the post rides on the kernel that produces the value (k_fold_partials, k_draw_window) and
// the wait on the kernel that consumes it (k_normalize_gathered, k_publish_window_keys).
//
// Two parities per slot: generation g uses parity g & 1.  A rank can start exchange g + 2 only after it has
// seen every peer's word of g + 1, which that peer posts (in stream order) after it consumed g, so a slot
// is never overwritten while somebody still reads it.
//
// Layout of one mailbox (bytes):   0  tot_gen [2][16] u64     totals: generation words
//                                256  tot_val [2][16] f64     totals: values
//                                512  win_done[2][16] u64     window: "rank r has written all its columns"
//                                768  hello   [16]    u64     connect-time self-test
//                               4096  win     [2][6][max_window] i64
#pragma once
#include "device_types.hpp"

namespace bpf
{

constexpr int kMailboxMaxWorld = 16;
constexpr size_t kMailboxHeader = 4096;
constexpr long long kMailboxTimeoutTicks = 500000000ll;  // default: 5 s of the 100 MHz wall clock
constexpr int kMailboxFusedWaitBlocks = 64;               // consumers with more blocks get k_mailbox_wait in front

struct MailboxDev
{
  int rank, world;                // world == 0: no mailbox (the kernels skip their post / wait)
  long long max_window;           // columns per window row
  char* peer[kMailboxMaxWorld];   // every rank's mailbox as mapped into this process (own included)
  unsigned* host_error;           // pinned host words, set to 1 by a wait that ran out of time: [0] totals, [1] window
  unsigned* dev_error;            // the same two flags in device memory, for the kernels behind a k_mailbox_wait
  long long timeout_ticks;        // bound of a wait (100 MHz wall clock)
};

__device__ __forceinline__ unsigned long long* mb_tot_gen(char* base, int parity, int r)
{
  return reinterpret_cast<unsigned long long*>(base) + parity * kMailboxMaxWorld + r;
}
__device__ __forceinline__ double* mb_tot_val(char* base, int parity, int r)
{
  return reinterpret_cast<double*>(base + 256) + parity * kMailboxMaxWorld + r;
}
__device__ __forceinline__ unsigned long long* mb_win_done(char* base, int parity, int r)
{
  return reinterpret_cast<unsigned long long*>(base + 512) + parity * kMailboxMaxWorld + r;
}
__device__ __forceinline__ unsigned long long* mb_hello(char* base, int r)
{
  return reinterpret_cast<unsigned long long*>(base + 768) + r;
}
// "rank r gave up a wait": set by r in every peer's mailbox when one of its waits runs out, and looked at by every
// wait of every rank (even one whose words are all there), so that a rank that stalled finds out that its peer has
// stopped waiting for it and does not finish the step alone
__device__ __forceinline__ unsigned long long* mb_fail(char* base, int r)
{
  return reinterpret_cast<unsigned long long*>(base + 1024) + r;
}
__device__ __forceinline__ long long* mb_window(char* base, int parity, long long max_window)
{
  return reinterpret_cast<long long*>(base + kMailboxHeader) + (size_t)parity * 6 * (size_t)max_window;
}

// bounded spin until *slot >= gen; false when the time ran out (every wave reaches the exit either way)
__device__ __forceinline__ bool mb_spin_ge(const unsigned long long* slot, unsigned long long gen, long long timeout_ticks)
{
  // relaxed polls (an acquire load is a load plus a cache invalidate, per iteration), one acquire once the word is there
  const long long t0 = wall_clock64();
  for (;;)
  {
    if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= gen)
    {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed when the caller's barrier opens
      return true;
    }
    if (wall_clock64() - t0 > timeout_ticks)
      return false;
    __builtin_amdgcn_s_sleep(2);
  }
}

// Block-wide wait on `world` generation words of this rank's own mailbox (threads 0 .. world-1 spin, the
// rest wait at the barrier); every thread of the block must call it.  which: 0 = totals, 1 = window.  False (for the
// whole block) when a word did not arrive in time: the flags are raised and the caller must leave its data alone, so
// that the host can finish the update over another transport (ShardedFilter's recovery).  A rank that gives up says so
// in every peer's mailbox, and every wait also fails when a peer has said so: the rank whose stall caused the
// time-out finds its words all there (its peer posted them long ago) but must not finish the step alone.
__device__ __forceinline__ bool mb_block_wait(const MailboxDev& M, unsigned long long* first_slot,
                                              unsigned long long gen, int which)
{
  int ok = 1;
  if ((int)threadIdx.x < M.world)
  {
    if (!mb_spin_ge(first_slot + threadIdx.x, gen, M.timeout_ticks))
      ok = 0;
    // a peer that has given up a wait of its own no longer takes part in this step
    if (__hip_atomic_load(mb_fail(M.peer[M.rank], threadIdx.x), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull)
      ok = 0;
  }
  ok = __syncthreads_and(ok);
  if (!ok && (int)threadIdx.x < M.world)
  {
    __hip_atomic_store(mb_fail(M.peer[threadIdx.x], M.rank), 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0)
    {
      __hip_atomic_store(M.host_error + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(M.dev_error + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // The polling lanes (one wave) have acquired at system scope and drained before the barrier above: that covers the
  // block's CU.  What the peers wrote sits in this rank's uncached mailbox, which no cache holds anyway.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return ok != 0;
}

// thread t < world stores this rank's total into peer t's mailbox, then the generation word
__device__ __forceinline__ void mb_post_total(const MailboxDev& M, int parity, unsigned long long gen, double value)
{
  if ((int)threadIdx.x < M.world)
  {
    char* base = M.peer[threadIdx.x];
    *mb_tot_val(base, parity, M.rank) = value;
    __hip_atomic_store(mb_tot_gen(base, parity, M.rank), gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// After a grid has stored its window columns into the peers: the last block to get here posts "done".
// Every thread of every block must call it.
__device__ __forceinline__ void mb_window_done_when_last(const MailboxDev& M, int parity, unsigned long long gen,
                                                         unsigned* counter)
{
  // every storing wave drains its peer stores, then ONE system-scope release per block behind the barrier (a fence by
  // every thread costs 2-4x one lane's)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1)
    {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // every block's peer stores were out before its ticket: ONE more system-scope fence, then the W "done" words
      // relaxed (a release store per peer is a cache write-back per peer: 8 ranks, 8 of them in a row)
      __threadfence_system();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int r = 0; r < M.world; ++r)
        __hip_atomic_store(mb_win_done(M.peer[r], parity, M.rank), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// the local total of a scoring stage that did not go through k_fold_partials (3-D path, beam model)
__global__ void k_mailbox_post_total(const double* value, const MailboxDev M, int parity, unsigned long long gen)
{
  mb_post_total(M, parity, gen, *value);
}

// A wait of its own (one block) in front of a consumer whose grid is large: hundreds of blocks spinning on the
// words would hold the whole GPU for the length of the wait (and, where several ranks share one GPU, keep the very
// kernel they are waiting for from being scheduled).  which: 0 = totals words, 1 = window "done" words.
__global__ void k_mailbox_wait(const MailboxDev M, int which, int parity, unsigned long long gen)
{
  (void)mb_block_wait(M, which == 0 ? mb_tot_gen(M.peer[M.rank], parity, 0) : mb_win_done(M.peer[M.rank], parity, 0), gen,
                      which);
}

// Self-test of the window path with a payload that can be checked: rank r stores pattern(r, gen, column) into the
// columns c = r (mod world) of every peer's window, posts "done"; the check kernel waits for all ranks and counts the
// cells that do not hold what their owner must have written.
__device__ __forceinline__ long long mb_pattern(int rank, unsigned long long gen, int row, int col)
{
  return (long long)(((unsigned long long)(rank + 1) << 48) ^ (gen << 32) ^ ((unsigned long long)row << 24) ^
                     (unsigned long long)col * 0x9E3779B1ull);
}

__global__ void k_mailbox_selftest_write(const MailboxDev M, int n_cols, int parity, unsigned long long gen,
                                         unsigned* counter)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_cols && c % M.world == M.rank)
    for (int r = 0; r < M.world; ++r)
    {
      long long* win = mb_window(M.peer[r], parity, M.max_window);
#pragma unroll
      for (int k = 0; k < 6; ++k)
        win[(size_t)k * (size_t)M.max_window + c] = mb_pattern(M.rank, gen, k, c);
    }
  mb_window_done_when_last(M, parity, gen, counter);
}

__global__ void k_mailbox_selftest_check(const MailboxDev M, int n_cols, int parity, unsigned long long gen,
                                         int* mismatches)
{
  (void)mb_block_wait(M, mb_win_done(M.peer[M.rank], parity, 0), gen, 1);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols)
    return;
  const long long* win = mb_window(M.peer[M.rank], parity, M.max_window);
  int bad = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    bad += win[(size_t)k * (size_t)M.max_window + c] != mb_pattern(c % M.world, gen, k, c);
  if (bad)
    atomicAdd(mismatches, bad);
}

// connect-time self-test: one full round (post to every peer, wait for every peer); result[0] = 1 when all arrived
__global__ void k_mailbox_hello(const MailboxDev M, unsigned long long token, int* result)
{
  __shared__ int s_ok;
  if (threadIdx.x == 0)
    s_ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < M.world)
  {
    __hip_atomic_store(mb_hello(M.peer[threadIdx.x], M.rank), token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!mb_spin_ge(mb_hello(M.peer[M.rank], threadIdx.x), token, M.timeout_ticks))
      atomicExch(&s_ok, 0);
  }
  __syncthreads();
  if (threadIdx.x == 0)
    result[0] = s_ok;
}

}  // namespace bpf
