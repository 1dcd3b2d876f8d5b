// OctoMap::updateDistancesLUT (octomap.cpp:174-333) on the device, value for value and column for column.
//
// The reference is a FIFO brushfire: every queue entry (cell, source) tries its six neighbours in a fixed order; a
// neighbour whose stored distance exceeds the distance to the entry's source by more than one quantisation step takes
// that distance and is queued behind everything already there (:287-311).  The stored distances are quantised bytes,
// a cell can be improved several times, and the columns of the two-level LUT are allocated in the order in which
// cells are first written (:314-333) -- so the result depends on the processing order.  What makes it parallel:
//
//   * FIFO means generation by generation: every entry queued while generation g is processed is processed after all
//     of generation g, in the order it was queued.  The t-th attempt of a generation (t = 6 * entry + direction) is a
//     total order of everything that happens in it.
//   * attempts on different cells do not see each other (an attempt reads and writes only its target), so a
//     generation is: form all attempts, group them by target (radix sort of target << 32 | t), fold each target's
//     attempts in t order by one thread -- the sequential rule, on a handful of attempts --, and queue the successful
//     ones in t order (a prefix sum over the success flags) as generation g + 1.
//   * column allocation order = order of each column's first successful write, i.e. of (generation, t): one
//     atomicMin per write, one sort of the columns at the end.
//
// The distances live in a dense byte volume (x, y column major, z contiguous) while the brushfire runs; the two-level
// layout is cut from it at the end.  Sorts and scans are rocPRIM's device-wide primitives (map-load work, not the
// hot path).
#pragma once
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "device_types.hpp"

namespace bpf
{

struct Lut3dArgs
{
  int w, h, nz;        // cells per axis of the cropped map
  int td;              // radius + 2: offsets of td or more in any axis are outside the reference's table
  double resolution, max_dist, unit;  // unit = max_dist / 255 (max_distance_ratio_, octomap.cpp:57)
  unsigned char* vol;  // [w * h][nz] quantised distances, 255 = far
  unsigned long long* col_first;  // [w * h] order stamp of the first write into the column (~0: never)
};

constexpr unsigned long long kLut3dNever = ~0ull;

__device__ __forceinline__ unsigned char lut3d_quantise(const Lut3dArgs& A, double d)
{
  // setDistanceToObject, octomap.cpp:328-332
  d = fmin(d, A.max_dist);
  d = d / A.max_dist * 255;
  return (unsigned char)(int)floor(d);
}

// generation 0 (iterateObstacleCells, :208-249): occupied cells inside the bounds get distance 0 in list order; their
// FIFO order is descending Index3 order (i, then j, then k), formed by sorting the keys written here
__global__ void k_lut3d_seed(const Lut3dArgs A, const int* __restrict__ occ, size_t n_occ, int min_i, int min_j, int min_k,
                             unsigned long long* __restrict__ keys)
{
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_occ)
    return;
  const int i = occ[3 * q] - min_i, j = occ[3 * q + 1] - min_j, k = occ[3 * q + 2] - min_k;
  if (i < 0 || i >= A.w || j < 0 || j >= A.h || k < 0 || k >= A.nz)
  {
    keys[q] = kLut3dNever;  // skipped (:226-227); sorts to the end
    return;
  }
  const unsigned col = (unsigned)j * (unsigned)A.w + (unsigned)i;
  A.vol[(size_t)col * A.nz + k] = 0;
  atomicMin(&A.col_first[col], (unsigned long long)q);  // generation 0, list order
  // ascending sort of the complement = descending (i, j, k)
  const unsigned long long key = ((unsigned long long)i << 42) | ((unsigned long long)j << 21) | (unsigned long long)k;
  keys[q] = ~key & 0x7FFFFFFFFFFFFFFFull;
}

__global__ void k_lut3d_seed_entries(const Lut3dArgs A, const unsigned long long* __restrict__ sorted, size_t n,
                                     int* __restrict__ cell, int* __restrict__ src)
{
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n)
    return;
  const unsigned long long key = ~sorted[q] & 0x7FFFFFFFFFFFFFFFull;
  const int i = (int)(key >> 42), j = (int)((key >> 21) & 0x1FFFFF), k = (int)(key & 0x1FFFFF);
  const int id = (j * A.w + i) * A.nz + k;
  cell[q] = id;
  src[q] = id;
}

// attempt t = 6 * entry + direction -> (target cell, squared offsets to the entry's source); false: not made
__device__ __forceinline__ bool lut3d_attempt(const Lut3dArgs& A, int cell, int source, int dir, int* target, int* di,
                                              int* dj, int* dk)
{
  const int k = cell % A.nz, col = cell / A.nz, i = col % A.w, j = col / A.w;
  int ti = i, tj = j, tk = k;
  // SHIFTS order and the bounds tests of iterateEmptyCells (:256-279)
  switch (dir)
  {
    case 0: if (i <= 0) return false; ti = i - 1; break;
    case 1: if (j <= 0) return false; tj = j - 1; break;
    case 2: if (k <= 0) return false; tk = k - 1; break;
    case 3: if (i >= A.w - 1) return false; ti = i + 1; break;
    case 4: if (j >= A.h - 1) return false; tj = j + 1; break;
    default: if (k >= A.nz - 1) return false; tk = k + 1; break;
  }
  const int sk = source % A.nz, scol = source / A.nz, si = scol % A.w, sj = scol / A.w;
  *di = abs(ti - si);
  *dj = abs(tj - sj);
  *dk = abs(tk - sk);
  if (*di >= A.td || *dj >= A.td || *dk >= A.td)
    return false;  // outside the reference's table: such a cell was never improved on the way
  *target = (tj * A.w + ti) * A.nz + tk;
  return true;
}

__global__ void k_lut3d_attempts(const Lut3dArgs A, const int* __restrict__ cell, const int* __restrict__ src, size_t n_entries,
                                 unsigned long long* __restrict__ keys)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 6 * n_entries)
    return;
  const size_t p = t / 6;
  int target, di, dj, dk;
  keys[t] = lut3d_attempt(A, cell[p], src[p], (int)(t % 6), &target, &di, &dj, &dk)
                ? (((unsigned long long)(unsigned)target << 32) | (unsigned long long)t)
                : kLut3dNever;
}

// one thread per target: its attempts in t order under the reference's rule (enqueue, :287-311)
__global__ void k_lut3d_fold(const Lut3dArgs A, const unsigned long long* __restrict__ sorted, size_t n_attempts,
                             const int* __restrict__ cell, const int* __restrict__ src, unsigned generation,
                             int* __restrict__ pushed)
{
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_attempts)
    return;
  const unsigned long long key = sorted[q];
  if (key == kLut3dNever)
    return;
  const unsigned target = (unsigned)(key >> 32);
  if (q > 0 && (unsigned)(sorted[q - 1] >> 32) == target)
    return;  // not the head of its target's run
  unsigned char ratio = A.vol[target];
  const unsigned col = target / (unsigned)A.nz;
  bool wrote = false;
  for (size_t r = q; r < n_attempts; ++r)
  {
    const unsigned long long kr = sorted[r];
    if (kr == kLut3dNever || (unsigned)(kr >> 32) != target)
      break;
    const unsigned t = (unsigned)(kr & 0xFFFFFFFFull);
    const unsigned p = t / 6u;
    int tgt, di, dj, dk;
    (void)lut3d_attempt(A, cell[p], src[p], (int)(t % 6u), &tgt, &di, &dj, &dk);
    const double new_distance = sqrt((double)(di * di + dj * dj + dk * dk)) * A.resolution;  // cached table (:152-172)
    const double old_distance = ratio * A.unit;                                              // getDistanceToObject
    if (old_distance - new_distance > A.unit)
    {
      ratio = lut3d_quantise(A, new_distance);
      pushed[t] = 1;
      if (!wrote)
        atomicMin(&A.col_first[col], ((unsigned long long)generation << 40) | (unsigned long long)t);
      wrote = true;
    }
  }
  if (wrote)
    A.vol[target] = ratio;
}

__global__ void k_lut3d_emit(const Lut3dArgs A, const int* __restrict__ cell, const int* __restrict__ src, size_t n_entries,
                             const int* __restrict__ pushed, const int* __restrict__ position,
                             int* __restrict__ next_cell, int* __restrict__ next_src)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 6 * n_entries || !pushed[t])
    return;
  const size_t p = t / 6;
  int target, di, dj, dk;
  (void)lut3d_attempt(A, cell[p], src[p], (int)(t % 6), &target, &di, &dj, &dk);
  next_cell[position[t]] = target;
  next_src[position[t]] = src[p];
}

// (order stamp, column) of every column that was written, for the allocation-order sort
__global__ void k_lut3d_columns(const Lut3dArgs A, unsigned long long* __restrict__ stamps, int* __restrict__ cols)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= A.w * A.h)
    return;
  stamps[c] = A.col_first[c];
  cols[c] = c;
}

// the two-level layout (:314-355): column 0 is the shared all-255 run, the r-th allocated column starts at nz (r + 1)
__global__ void k_lut3d_layout(const Lut3dArgs A, const int* __restrict__ cols_sorted, int n_alloc,
                               unsigned* __restrict__ pose_indices, unsigned char* __restrict__ ratios)
{
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)(n_alloc + 1) * A.nz;
  if (q >= total)
    return;
  const size_t r = q / A.nz, k = q % A.nz;
  if (r == 0)
  {
    ratios[q] = 255;
    return;
  }
  const int col = cols_sorted[r - 1];
  ratios[q] = A.vol[(size_t)col * A.nz + k];
  if (k == 0)
    pose_indices[col] = (unsigned)(r * A.nz);
}

}  // namespace bpf
