// C-ABI: 3-D map + point cloud.
#ifndef BPF_CLOUD_BORDER
#define BPF_CLOUD_BORDER 1  // 0 (experiment builds): the planar dense kernel always in its plain form
#endif
// ---------------------------------------------------------------------- 3-D map + point cloud
int bpf_map3d_set(bpf_engine* e, const uint32_t* pose_indices, size_t n_pose_indices, const uint8_t* distance_ratios,
                  size_t n_distance_ratios, const int min_cells[3], const int max_cells[3], double resolution,
                  double max_dist)
{
  if (!e || !pose_indices || !distance_ratios || !min_cells || !max_cells || !(resolution > 0) || !(max_dist > 0))
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad 3-D map arguments") : BPF_ERR_INVALID_ARGUMENT;
  const long long w = (long long)max_cells[0] - min_cells[0] + 1, h = (long long)max_cells[1] - min_cells[1] + 1,
                  nz = (long long)max_cells[2] - min_cells[2] + 1;
  if (w <= 0 || h <= 0 || nz <= 0 || (size_t)(w * h) != n_pose_indices)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "pose_indices size does not match the cell bounds");
  // the scoring kernel forms the column's byte offset with a 24-bit multiply-add and 32-bit offsets, and folds the
  // lower cell bounds into a double addend (kernels_cloud.hpp)
  if (w >= (1 << 22) || h >= (1 << 24) || n_pose_indices >= (1ull << 30))
    return e->fail(BPF_ERR_CAPACITY, "3-D map wider than 2^22 cells, longer than 2^24 or with more than 2^30 columns");
  for (int d = 0; d < 3; ++d)
    if (std::abs((long long)min_cells[d]) >= (1 << 20) || std::abs((long long)max_cells[d]) >= (1 << 20))
      return e->fail(BPF_ERR_CAPACITY, "3-D map cell bounds beyond +-2^20");
  // every column start must leave room for a whole z column (octomap.cpp:315-333)
  for (size_t i = 0; i < n_pose_indices; ++i)
    if ((size_t)pose_indices[i] + (size_t)nz > n_distance_ratios)
      return e->fail(BPF_ERR_INVALID_ARGUMENT, "pose_indices entry points past distance_ratios");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, e->d_pose_indices.reserve(n_pose_indices));
  HIPCHK(e, e->d_ratios.reserve(n_distance_ratios));
  H2D_OR_RETURN(h2d_from_host_sync(e, e->d_pose_indices.p, pose_indices, n_pose_indices * sizeof(uint32_t)));
  H2D_OR_RETURN(h2d_from_host_sync(e, e->d_ratios.p, distance_ratios, n_distance_ratios));
  Map3dDev& M = e->map3;
  M.pose_indices = e->d_pose_indices.p;
  M.distance_ratios = e->d_ratios.p;
  for (int d = 0; d < 3; ++d)
  {
    M.min_c[d] = min_cells[d];
    M.max_c[d] = max_cells[d];
  }
  M.width = (int)w;
  M.resolution = resolution;
  M.inv_resolution = 1.0 / resolution;
  e->map3_max_dist = max_dist;
  e->n_pose_indices = n_pose_indices;
  e->n_ratios = n_distance_ratios;
  e->have_map3d = true;
  // the dense tiled copy for the scoring kernel's gathers, when a plane fits the 24-bit multiply and the volume 1 GiB
  M.dense = nullptr;
  M.dense_k = 0;
  M.dense_plane = 0;
  M.border_code = -1;
  M.zero_code = -1;
  {
    // grid positions 0 .. w + 1 and 0 .. h + 1: the map's cells at (i + 1, j + 1) with a border cell all round
    const long long ntx = (w + 2 + 7) / 8, nty = (h + 2 + 7) / 8;
    const long long plane = ntx * nty * 64;
    if (plane < (1 << 24) && 8 * ntx - 1 < (1 << 24) && plane * (nz + 1) <= (1ll << 30))
    {
      // two distance ratios that no entry of the LUT holds (the two highest such): one fills the border and the
      // padding -- the scoring kernel's BORDER form reads the off-map term under it --, the other one more plane behind
      // the last, under which the kernel reads 0.0 (the padding lanes of a chunk's last group of points).  With
      // max_dist / resolution = 6 a LUT holds some 30 distinct ratios; one that leaves fewer than two free keeps the
      // plain form.
      {
        bool used[256] = {};
        for (size_t i = 0; i < n_distance_ratios; ++i)
          used[distance_ratios[i]] = true;
        int free2[2] = { -1, -1 }, nf = 0;
        for (int c = 255; c >= 0 && nf < 2; --c)
          if (!used[c])
            free2[nf++] = c;
        if (nf == 2)
        {
          M.border_code = free2[0];
          M.zero_code = free2[1];
        }
      }
      HIPCHK(e, e->d_dense3d.reserve((size_t)(plane * (nz + 1))));
      HIPCHK(e, hipMemsetAsync(e->d_dense3d.p, M.border_code >= 0 ? M.border_code : 0xFF, (size_t)(plane * nz),
                               e->stream));
      HIPCHK(e, hipMemsetAsync(e->d_dense3d.p + (size_t)(plane * nz), M.zero_code >= 0 ? M.zero_code : 0xFF,
                               (size_t)plane, e->stream));
      const size_t total = (size_t)(w * h * nz);
      hipLaunchKernelGGL(k_dense3d_build, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream,
                         e->d_pose_indices.p, e->d_ratios.p, (int)w, (int)h, (int)nz, (unsigned)(8 * ntx - 1),
                         (unsigned)plane, e->d_dense3d.p);
      HIPCHK(e, hipGetLastError());
      HIPCHK(e, hipStreamSynchronize(e->stream));
      M.dense = e->d_dense3d.p;
      M.dense_k = (unsigned)(8 * ntx - 1);
      M.dense_plane = (unsigned)plane;
    }
  }
  return BPF_OK;
}

namespace
{
__global__ void k_count_valid_keys(const unsigned long long* __restrict__ keys, size_t n, unsigned long long* count)
{
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mine = (q < n && keys[q] != kLut3dNever) ? 1ull : 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine)
    atomicAdd(count, mine);
}

// OctoMap::updateDistancesLUT on the device (kernels_lut3d.hpp): the FIFO brushfire generation by generation.
// *handled = false when the dense working volume does not fit (2^31 cells / 1 GiB): the host builder takes over.
int build_lut3d_device(bpf_engine* e, const int* occupied_ijk, size_t n_occupied, const int min_cells[3],
                       const int max_cells[3], double resolution, double max_dist, bool* handled)
{
  *handled = false;
  const long long w = (long long)max_cells[0] - min_cells[0] + 1, h = (long long)max_cells[1] - min_cells[1] + 1,
                  nz = (long long)max_cells[2] - min_cells[2] + 1;
  const long long volume = w * h * nz;
  if (volume >= (1ll << 30) || w >= (1 << 21) || h >= (1 << 21) || nz >= (1 << 21) || n_occupied >= (1ull << 31))
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  hipStream_t st = e->stream;
  DevBuf<unsigned char> vol, tmp;
  DevBuf<unsigned long long> col_first, keys_a, keys_b, counter;
  DevBuf<int> occ, cell_a, src_a, cell_b, src_b, pushed, position, cols_a, cols_b;
  HIPCHK(e, vol.reserve((size_t)volume));
  HIPCHK(e, col_first.reserve((size_t)(w * h)));
  HIPCHK(e, counter.reserve(1));
  HIPCHK(e, hipMemsetAsync(vol.p, 0xFF, (size_t)volume, st));
  HIPCHK(e, hipMemsetAsync(col_first.p, 0xFF, (size_t)(w * h) * sizeof(unsigned long long), st));
  Lut3dArgs A{};
  A.w = (int)w;
  A.h = (int)h;
  A.nz = (int)nz;
  A.td = static_cast<int>(std::floor(max_dist / resolution)) + 2;  // CachedDistanceOctoMap (:152-172)
  A.resolution = resolution;
  A.max_dist = max_dist;
  A.unit = max_dist / 255;
  A.vol = vol.p;
  A.col_first = col_first.p;
  auto ensure_tmp = [&](size_t bytes) -> hipError_t { return tmp.reserve(bytes + 256); };
  auto count_valid = [&](const unsigned long long* keys, size_t n, unsigned long long* out) -> int {
    HIPCHK(e, hipMemsetAsync(counter.p, 0, sizeof(unsigned long long), st));
    if (n)
      hipLaunchKernelGGL(k_count_valid_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, n, counter.p);
    HIPCHK(e, hipMemcpyAsync(out, counter.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    return BPF_OK;
  };
  // ---- generation 0
  size_t n_entries = 0;
  if (n_occupied)
  {
    HIPCHK(e, occ.reserve(3 * n_occupied));
    H2D_OR_RETURN(h2d_from_host(e, occ.p, occupied_ijk, 3 * n_occupied * sizeof(int), st));
    HIPCHK(e, keys_a.reserve(n_occupied));
    HIPCHK(e, keys_b.reserve(n_occupied));
    hipLaunchKernelGGL(k_lut3d_seed, dim3((unsigned)((n_occupied + 255) / 256)), dim3(256), 0, st, A, (const int*)occ.p,
                       n_occupied, min_cells[0], min_cells[1], min_cells[2], keys_a.p);
    size_t bytes = 0;
    HIPCHK(e, rocprim::radix_sort_keys(nullptr, bytes, keys_a.p, keys_b.p, n_occupied, 0, 64, st));
    HIPCHK(e, ensure_tmp(bytes));
    HIPCHK(e, rocprim::radix_sort_keys(tmp.p, bytes, keys_a.p, keys_b.p, n_occupied, 0, 64, st));
    unsigned long long valid = 0;
    int rc = count_valid(keys_b.p, n_occupied, &valid);
    if (rc != BPF_OK)
      return rc;
    n_entries = (size_t)valid;
    HIPCHK(e, cell_a.reserve(std::max<size_t>(n_entries, 1)));
    HIPCHK(e, src_a.reserve(std::max<size_t>(n_entries, 1)));
    if (n_entries)
      hipLaunchKernelGGL(k_lut3d_seed_entries, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, st, A,
                         (const unsigned long long*)keys_b.p, n_entries, cell_a.p, src_a.p);
  }
  // ---- the brushfire, generation by generation (iterateEmptyCells)
  DevBuf<int>*cell_cur = &cell_a, *src_cur = &src_a, *cell_nxt = &cell_b, *src_nxt = &src_b;
  unsigned generation = 1;
  while (n_entries > 0)
  {
    if (generation >= (1u << 20) || 6 * n_entries >= (1ull << 32))
      return e->fail(BPF_ERR_CAPACITY, "3-D LUT builder: more than 2^32 attempts in one generation");
    const size_t n_att = 6 * n_entries;
    HIPCHK(e, keys_a.reserve(n_att));
    HIPCHK(e, keys_b.reserve(n_att));
    HIPCHK(e, pushed.reserve(n_att));
    HIPCHK(e, position.reserve(n_att));
    const dim3 grid((unsigned)((n_att + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_lut3d_attempts, grid, block, 0, st, A, (const int*)cell_cur->p, (const int*)src_cur->p, n_entries,
                       keys_a.p);
    size_t bytes = 0;
    HIPCHK(e, rocprim::radix_sort_keys(nullptr, bytes, keys_a.p, keys_b.p, n_att, 0, 64, st));
    HIPCHK(e, ensure_tmp(bytes));
    HIPCHK(e, rocprim::radix_sort_keys(tmp.p, bytes, keys_a.p, keys_b.p, n_att, 0, 64, st));
    HIPCHK(e, hipMemsetAsync(pushed.p, 0, n_att * sizeof(int), st));
    hipLaunchKernelGGL(k_lut3d_fold, grid, block, 0, st, A, (const unsigned long long*)keys_b.p, n_att,
                       (const int*)cell_cur->p, (const int*)src_cur->p, generation, pushed.p);
    bytes = 0;
    HIPCHK(e, rocprim::exclusive_scan(nullptr, bytes, pushed.p, position.p, 0, n_att, rocprim::plus<int>(), st));
    HIPCHK(e, ensure_tmp(bytes));
    HIPCHK(e, rocprim::exclusive_scan(tmp.p, bytes, pushed.p, position.p, 0, n_att, rocprim::plus<int>(), st));
    int last[2] = { 0, 0 };
    HIPCHK(e, hipMemcpyAsync(&last[0], position.p + (n_att - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipMemcpyAsync(&last[1], pushed.p + (n_att - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    const size_t n_next = (size_t)last[0] + (size_t)last[1];
    HIPCHK(e, cell_nxt->reserve(std::max<size_t>(n_next, 1)));
    HIPCHK(e, src_nxt->reserve(std::max<size_t>(n_next, 1)));
    if (n_next)
      hipLaunchKernelGGL(k_lut3d_emit, grid, block, 0, st, A, (const int*)cell_cur->p, (const int*)src_cur->p, n_entries,
                         (const int*)pushed.p, (const int*)position.p, cell_nxt->p, src_nxt->p);
    HIPCHK(e, hipGetLastError());
    std::swap(cell_cur, cell_nxt);
    std::swap(src_cur, src_nxt);
    n_entries = n_next;
    ++generation;
  }
  // ---- the two-level layout, columns in the order of their first write
  const size_t n_cols = (size_t)(w * h);
  HIPCHK(e, keys_a.reserve(n_cols));
  HIPCHK(e, keys_b.reserve(n_cols));
  HIPCHK(e, cols_a.reserve(n_cols));
  HIPCHK(e, cols_b.reserve(n_cols));
  hipLaunchKernelGGL(k_lut3d_columns, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, A, keys_a.p, cols_a.p);
  size_t bytes = 0;
  HIPCHK(e, rocprim::radix_sort_pairs(nullptr, bytes, keys_a.p, keys_b.p, cols_a.p, cols_b.p, n_cols, 0, 64, st));
  HIPCHK(e, ensure_tmp(bytes));
  HIPCHK(e, rocprim::radix_sort_pairs(tmp.p, bytes, keys_a.p, keys_b.p, cols_a.p, cols_b.p, n_cols, 0, 64, st));
  unsigned long long n_alloc = 0;
  int rc = count_valid(keys_b.p, n_cols, &n_alloc);
  if (rc != BPF_OK)
    return rc;
  const size_t n_ratios = (size_t)(n_alloc + 1) * (size_t)nz;
  if (n_ratios > 0xffffffffull)
    return e->fail(BPF_ERR_CAPACITY, "distance_ratios would pass the 32-bit column index range");
  DevBuf<unsigned> d_pose;
  DevBuf<unsigned char> d_ratios;
  HIPCHK(e, d_pose.reserve(n_cols));
  HIPCHK(e, d_ratios.reserve(n_ratios));
  HIPCHK(e, hipMemsetAsync(d_pose.p, 0, n_cols * sizeof(unsigned), st));
  hipLaunchKernelGGL(k_lut3d_layout, dim3((unsigned)((n_ratios + 255) / 256)), dim3(256), 0, st, A, (const int*)cols_b.p,
                     (int)n_alloc, d_pose.p, d_ratios.p);
  HIPCHK(e, hipGetLastError());
  std::vector<uint32_t> pose_indices(n_cols);
  std::vector<uint8_t> ratios(n_ratios);
  H2D_OR_RETURN(d2h_to_host(e, pose_indices.data(), d_pose.p, n_cols * sizeof(uint32_t), st));
  H2D_OR_RETURN(d2h_to_host(e, ratios.data(), d_ratios.p, n_ratios, st));
  HIPCHK(e, hipStreamSynchronize(st));
  e->lut3d_generations = (int)generation - 1;
  *handled = true;
  return bpf_map3d_set(e, pose_indices.data(), pose_indices.size(), ratios.data(), ratios.size(), min_cells, max_cells,
                       resolution, max_dist);
}
}  // namespace

int bpf_map3d_build_distances_lut(bpf_engine* e, const int* occupied_ijk, size_t n_occupied, const int min_cells[3],
                                  const int max_cells[3], double resolution, double max_dist)
{
  if (!e || (!occupied_ijk && n_occupied) || !min_cells || !max_cells || !(resolution > 0))
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad 3-D map arguments") : BPF_ERR_INVALID_ARGUMENT;
  if (max_dist == 0.0)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "max distance to object is 0 (octomap.cpp:177-181)");
  const long long w = (long long)max_cells[0] - min_cells[0] + 1, h = (long long)max_cells[1] - min_cells[1] + 1,
                  nz = (long long)max_cells[2] - min_cells[2] + 1;
  if (w <= 0 || h <= 0 || nz <= 0 || w * h > 0x7fffffffll)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "bad cell bounds");
  e->lut3d_generations = 0;
  if (!e->lut_host)
  {
    bool handled = false;
    int rcd = build_lut3d_device(e, occupied_ijk, n_occupied, min_cells, max_cells, resolution, max_dist, &handled);
    if (rcd != BPF_OK || handled)
      return rcd;
  }
  // the same on the host, serially (BPF_OPT_LUT_HOST, or a map too large for the dense working volume)
  struct Cell
  {
    int i, j, k, si, sj, sk;
  };
  struct Index3
  {
    int v[3];
    bool operator<(const Index3& o) const  // octomap.h:51-54
    {
      return v[0] != o.v[0] ? v[0] < o.v[0] : v[1] != o.v[1] ? v[1] < o.v[1] : v[2] < o.v[2];
    }
  };
  std::vector<uint32_t> pose_indices((size_t)(w * h), 0u);
  std::vector<uint8_t> ratios((size_t)nz, 255);  // the shared all-255 column 0 (octomap.cpp:189-190)
  const double ratio_unit = max_dist / 255;      // max_distance_ratio_ (octomap.cpp:57)
  auto column = [&](int i, int j) { return (size_t)(j - min_cells[1]) * (size_t)w + (size_t)(i - min_cells[0]); };
  auto get = [&](int i, int j, int k) {  // getDistanceToObject :336-350
    return ratios[(size_t)pose_indices[column(i, j)] + (size_t)(k - min_cells[2])] * ratio_unit;
  };
  bool too_big = false;
  auto set = [&](int i, int j, int k, double d) {  // setDistanceToObject :314-333
    uint32_t& start = pose_indices[column(i, j)];
    if (start == 0)
    {
      if (ratios.size() + (size_t)nz > 0xffffffffull)
      {
        too_big = true;
        return;
      }
      start = (uint32_t)ratios.size();
      ratios.resize(ratios.size() + (size_t)nz, 255);
    }
    d = std::min(d, max_dist);
    d = d / max_dist * 255;
    ratios[(size_t)start + (size_t)(k - min_cells[2])] = (uint8_t)static_cast<int>(std::floor(d));
  };
  // CachedDistanceOctoMap (:152-172)
  const int radius = static_cast<int>(std::floor(max_dist / resolution));
  const int td = radius + 2;
  std::vector<double> cached((size_t)td * td * td);
  for (int a = 0; a < td; ++a)
    for (int b = 0; b < td; ++b)
      for (int c = 0; c < td; ++c)
        cached[((size_t)a * td + b) * td + c] = std::sqrt((double)(a * a + b * b + c * c)) * resolution;
  // iterateObstacleCells (:208-249): zero distance in iteration order, FIFO seeded in descending Index3 order
  std::priority_queue<Index3> ordering;
  for (size_t q = 0; q < n_occupied; ++q)
  {
    const int* v = &occupied_ijk[3 * q];
    bool valid = true;
    for (int d = 0; d < 3; ++d)
      valid = valid && v[d] >= min_cells[d] && v[d] <= max_cells[d];
    if (!valid)
      continue;
    set(v[0], v[1], v[2], 0.0);
    ordering.push(Index3{ { v[0], v[1], v[2] } });
  }
  std::queue<Cell> fifo;
  while (!ordering.empty())
  {
    const Index3 s = ordering.top();
    ordering.pop();
    fifo.push(Cell{ s.v[0], s.v[1], s.v[2], s.v[0], s.v[1], s.v[2] });
  }
  // iterateEmptyCells / enqueue (:251-311)
  static const int kShifts[6][3] = { { -1, 0, 0 }, { 0, -1, 0 }, { 0, 0, -1 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
  while (!fifo.empty() && !too_big)
  {
    const Cell cur = fifo.front();
    const bool open[6] = { cur.i > min_cells[0], cur.j > min_cells[1], cur.k > min_cells[2],
                           cur.i < max_cells[0], cur.j < max_cells[1], cur.k < max_cells[2] };
    for (int s = 0; s < 6; ++s)
    {
      if (!open[s])
        continue;
      const int i = cur.i + kShifts[s][0], j = cur.j + kShifts[s][1], k = cur.k + kShifts[s][2];
      const int di = std::abs(i - cur.si), dj = std::abs(j - cur.sj), dk = std::abs(k - cur.sk);
      if (di >= td || dj >= td || dk >= td)
        continue;  // the reference indexes its table unchecked; a cell this far out was never improved on the way
      const double new_distance = cached[((size_t)di * td + dj) * td + dk];
      const double old_distance = get(i, j, k);
      if (old_distance - new_distance > ratio_unit)
      {
        set(i, j, k, new_distance);
        fifo.push(Cell{ i, j, k, cur.si, cur.sj, cur.sk });
      }
    }
    fifo.pop();
  }
  if (too_big)
    return e->fail(BPF_ERR_CAPACITY, "distance_ratios would pass the 32-bit column index range");
  return bpf_map3d_set(e, pose_indices.data(), pose_indices.size(), ratios.data(), ratios.size(), min_cells, max_cells,
                       resolution, max_dist);
}

int bpf_map3d_builder_generations(bpf_engine* e, int* generations_out)
{
  if (!e || !generations_out)
    return BPF_ERR_INVALID_ARGUMENT;
  *generations_out = e->lut3d_generations;
  return BPF_OK;
}

int bpf_map3d_get_distances_lut(bpf_engine* e, uint32_t* pose_indices, size_t pose_capacity, size_t* n_pose_indices,
                                uint8_t* distance_ratios, size_t ratios_capacity, size_t* n_distance_ratios)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_map3d)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no 3-D map set");
  if (n_pose_indices)
    *n_pose_indices = e->n_pose_indices;
  if (n_distance_ratios)
    *n_distance_ratios = e->n_ratios;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (pose_indices)
  {
    if (pose_capacity < e->n_pose_indices)
      return e->fail(BPF_ERR_CAPACITY, "pose_indices output too small");
    H2D_OR_RETURN(d2h_to_host(e, pose_indices, e->d_pose_indices.p, e->n_pose_indices * sizeof(uint32_t), e->stream));
  }
  if (distance_ratios)
  {
    if (ratios_capacity < e->n_ratios)
      return e->fail(BPF_ERR_CAPACITY, "distance_ratios output too small");
    H2D_OR_RETURN(d2h_to_host(e, distance_ratios, e->d_ratios.p, e->n_ratios, e->stream));
  }
  return BPF_OK;
}

int bpf_cloud_init(bpf_engine* e, int max_beams)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->cloud_max_beams = max_beams;
  if (!e->cloud_configured)
  {
    e->cm.off_map_factor = 1.0;  // point_cloud_scanner.cpp:36-38
    e->cm.tf_quat[3] = 1.0;
  }
  return BPF_OK;
}

int bpf_cloud_set_model(bpf_engine* e, double z_hit, double z_rand, double sigma_hit)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->cm.model = BPF_CLOUD_MODEL;
  e->cloud_z_hit = z_hit;
  e->cloud_z_rand = z_rand;
  e->cloud_sigma = sigma_hit;
  e->cloud_configured = true;
  return BPF_OK;
}

int bpf_cloud_set_model_gompertz(bpf_engine* e, double z_hit, double z_rand, double sigma_hit, double gompertz_a,
                                 double gompertz_b, double gompertz_c, double input_shift, double input_scale,
                                 double output_shift)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->cm.model = BPF_CLOUD_MODEL_GOMPERTZ;
  e->cm.g = GompertzDev{ gompertz_a, gompertz_b, gompertz_c, input_shift, input_scale, output_shift };
  e->cloud_z_hit = z_hit;
  e->cloud_z_rand = z_rand;
  e->cloud_sigma = sigma_hit;
  e->cloud_configured = true;
  return BPF_OK;
}

int bpf_cloud_set_map_factors(bpf_engine* e, double off_map_factor, double, double)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->cm.off_map_factor = off_map_factor;  // the 3-D recalcWeight only uses this one (:205-229)
  return BPF_OK;
}

int bpf_cloud_set_scanner_to_footprint_tf(bpf_engine* e, const double xyz[3], const double quat_xyzw[4])
{
  if (!e || !xyz || !quat_xyzw)
    return BPF_ERR_INVALID_ARGUMENT;
  std::memcpy(e->cm.tf_xyz, xyz, 3 * sizeof(double));
  std::memcpy(e->cm.tf_quat, quat_xyzw, 4 * sizeof(double));
  return BPF_OK;
}

namespace
{
int score_cloud(bpf_engine* e, ParticlesDev p, int n, const float* points_xyz, int n_points)
{
  if (!e->have_map3d)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no 3-D map set");
  if (!e->cloud_configured)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "point-cloud model not set");
  if (!points_xyz || n_points <= 0 || n <= 0)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "empty cloud or sample set");
  // stage points as float SoA; the table of per-ratio terms follows point_cloud_scanner.cpp:137-159,177-191
  HIPCHK(e, hipStreamSynchronize(e->stream));  // staging buffers are single-slot
  HIPCHK(e, e->h_points.reserve((size_t)n_points * 3));
  HIPCHK(e, e->d_points.reserve((size_t)n_points * 3));
  for (int q = 0; q < n_points; ++q)
  {
    e->h_points.p[q] = points_xyz[3 * q];
    e->h_points.p[(size_t)n_points + q] = points_xyz[3 * q + 1];
    e->h_points.p[2 * (size_t)n_points + q] = points_xyz[3 * q + 2];
  }
  HIPCHK(e, hipMemcpyAsync(e->d_points.p, e->h_points.p, (size_t)n_points * 3 * sizeof(float), hipMemcpyHostToDevice,
                           e->stream));
  HIPCHK(e, e->h_cloud_table.reserve(257));
  HIPCHK(e, e->d_cloud_table.reserve(257));
  const double denom = 2 * e->cloud_sigma * e->cloud_sigma;
  const double max_dist = e->map3_max_dist;
  const double rand_mult = 1.0 / max_dist;  // :140: 1/max_distance, not 1/range_max
  const double ratio = max_dist / 255;      // max_distance_ratio_, octomap.cpp:58
  for (int k = 0; k <= 256; ++k)
  {
    const double z = (k == 256) ? max_dist : k * ratio;
    double pz = e->cloud_z_hit * std::exp(-(z * z) / denom);
    if (e->cm.model == BPF_CLOUD_MODEL)
    {
      pz += e->cloud_z_rand * rand_mult;
      e->h_cloud_table.p[k] = pz * pz * pz;
    }
    else
    {
      pz += e->cloud_z_rand;
      e->h_cloud_table.p[k] = pz;
    }
  }
  HIPCHK(e, hipMemcpyAsync(e->d_cloud_table.p, e->h_cloud_table.p, 257 * sizeof(double), hipMemcpyHostToDevice,
                           e->stream));
  const int n_chunks = blocks_for(n_points, kCloudChunk);
  HIPCHK(e, e->d_affine.reserve((size_t)n * 12));
  HIPCHK(e, e->d_cloud_partials.reserve((size_t)n_chunks * n));
  hipLaunchKernelGGL(k_cloud_affine, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, p, n, e->cm, e->d_affine.p);
  CloudScoreArgs A{};
  A.n = n;
  A.affine = e->d_affine.p;
  A.points = e->d_points.p;
  A.n_points = n_points;
  A.map = e->map3;
  A.table = e->d_cloud_table.p;
  A.partials = e->d_cloud_partials.p;
  A.slabs = std::max(1, std::min(blocks_for(n, 4), std::max(1, (e->n_cu * 6) / n_chunks)));
  for (int k = 0; k < 8; ++k)
    A.round_count[k] = A.round_base[k] = A.round_first_slab[k] = 0;
  A.n_rounds = 0;
  if (e->graded_shares && n_chunks * A.slabs == e->n_cu * 6 && e->n_cu % n_chunks == 0 && n >= A.slabs * 4 * 64)
  {
    // graded partition as in the planar kernel: blocks are placed in linear order (x fastest), n_cu per round, and the
    // SIMD favours its oldest wave; with equal shares the six rounds end at 12.3 / 13.5 / 14.9 / 16.6 / 18.3 / 20.2 ms
    // (tools/cloud_span.py).  Shares ~ end^-1.5, the exponent that fits the planar kernel's measured optimum.
    static const double kShare[6] = { 0.246, 0.219, 0.185, 0.148, 0.114, 0.088 };
    const int slabs_per_round = e->n_cu / n_chunks;
    const double per_slot = (double)n / (slabs_per_round * 4);  // particles of one (slab, wave) slot over all rounds
    int base = 0;
    for (int r = 0; r < 6; ++r)
    {
      A.round_count[r] = std::max(1, (int)std::ceil(per_slot * kShare[r]));
      A.round_base[r] = base;
      A.round_first_slab[r] = r * slabs_per_round;
      base += A.round_count[r] * slabs_per_round * 4;
    }
    A.n_rounds = 6;
  }
  {
    // exact reciprocal?  1/res must fit 29 bits (so float * rinv is exact) and rinv*res must round to 1
    const double rinv = e->map3.inv_resolution;
    uint64_t bits;
    std::memcpy(&bits, &rinv, 8);
    const bool exact_rinv = (bits & ((1ull << 24) - 1)) == 0 && std::fabs(std::fma(rinv, e->map3.resolution, -1.0)) < 1.1e-16;
    // scanner mounted without roll or pitch: every particle's matrix has the z row (0, 0, 1) and no z term in x, y
    // (k_cloud_score, PLANAR); the packed z cell needs 16 bits
    const bool planar = e->cm.tf_quat[0] == 0.0 && e->cm.tf_quat[1] == 0.0 &&
                        (e->map3.max_c[2] - e->map3.min_c[2]) < 65534;
    A.planar_tz = (float)(e->cm.tf_xyz[2] + 0.0);
    const bool dense = e->map3.dense != nullptr && e->cloud_dense;
#define BPF_CLOUD_LAUNCH(X, P, D) \
  LAUNCH_TIMED(e, BPF_K_SCORE, (k_cloud_score<X, P, D>), dim3(n_chunks, A.slabs), dim3(256), 0, A)
    const bool border = BPF_CLOUD_BORDER != 0 && planar && dense && e->map3.border_code >= 0;
    if (exact_rinv && border)
      LAUNCH_TIMED(e, BPF_K_SCORE, (k_cloud_score<true, true, true, true>), dim3(n_chunks, A.slabs), dim3(256), 0, A);
    else if (border)
      LAUNCH_TIMED(e, BPF_K_SCORE, (k_cloud_score<false, true, true, true>), dim3(n_chunks, A.slabs), dim3(256), 0, A);
    else if (exact_rinv && planar && dense)
      BPF_CLOUD_LAUNCH(true, true, true);
    else if (exact_rinv && planar)
      BPF_CLOUD_LAUNCH(true, true, false);
    else if (exact_rinv && dense)
      BPF_CLOUD_LAUNCH(true, false, true);
    else if (exact_rinv)
      BPF_CLOUD_LAUNCH(true, false, false);
    else if (planar && dense)
      BPF_CLOUD_LAUNCH(false, true, true);
    else if (planar)
      BPF_CLOUD_LAUNCH(false, true, false);
    else if (dense)
      BPF_CLOUD_LAUNCH(false, false, true);
    else
      BPF_CLOUD_LAUNCH(false, false, false);
#undef BPF_CLOUD_LAUNCH
  }
  CloudFinishArgs F{};
  F.p = p;
  F.n = n;
  F.partials = e->d_cloud_partials.p;
  F.n_chunks = n_chunks;
  F.n_points = n_points;
  F.map = e->map3;
  F.model = e->cm;
  hipLaunchKernelGGL(k_cloud_finish, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, F);
  HIPCHK(e, hipGetLastError());
  e->evals_last = (long long)n * n_points;
  return BPF_OK;
}
}  // namespace

double bpf_cloud_apply_model_to_sample_set(bpf_engine* e, double* samples, int sample_count, const float* points_xyz,
                                           int n_points, int* status)
{
  int dummy;
  if (!status)
    status = &dummy;
  *status = BPF_OK;
  if (!e || !samples)
  {
    *status = BPF_ERR_INVALID_ARGUMENT;
    return 0.0;
  }
  if (e->cloud_max_beams < 2)
    return 0.0;  // point_cloud_scanner.cpp:109-110
  auto bail = [&](int code) { *status = code; return 0.0; };
  if (hipSetDevice(e->device) != hipSuccess)
    return bail(e->fail(BPF_ERR_HIP, "hipSetDevice"));
  int rc = ensure_scalars(e);
  if (rc != BPF_OK)
    return bail(rc);
  rc = upload_samples(e, samples, sample_count, e->scratch);
  if (rc != BPF_OK)
    return bail(rc);
  rc = score_cloud(e, e->scratch.dev(), sample_count, points_xyz, n_points);
  if (rc != BPF_OK)
    return bail(rc);
  rc = sum_into_slot(e, e->scratch.w.p, sample_count, 0, 0, sample_count);
  if (rc != BPF_OK)
    return bail(rc);
  rc = download_weights(e, e->scratch, sample_count, samples);
  if (rc != BPF_OK)
    return bail(rc);
  return e->h_scalars.p->v[0];
}

int bpf_pf_update_sensor_cloud(bpf_engine* e, const float* points_xyz, int n_points)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (e->cloud_max_beams < 2)
    return BPF_OK;  // PointCloudScanner::updateSensor returns false (:95-96)
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  const int n = e->sample_count;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  int rc = score_cloud(e, s.dev(), n, points_xyz, n_points);
  if (rc != BPF_OK)
    return rc;
  rc = sum_into_slot(e, s.w.p, n, 0, 1, n);
  if (rc != BPF_OK)
    return rc;
  {
    ProfScope ps(e, BPF_K_NORMALIZE);
    hipLaunchKernelGGL(k_normalize, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, s.w.p, n, e->d_scalars.p, 0,
                       0.0, n);
  }
  HIPCHK(e, hipGetLastError());
  e->last_status = BPF_OK;
  return BPF_OK;
}
