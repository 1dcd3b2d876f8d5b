// C-ABI: reference brushfire (host).
// ---------------------------------------------------------------------- reference brushfire (host)
int bpf_map2d_build_distances_lut_reference(bpf_engine* e, double max_dist)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_map)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no 2-D map set");
  if (max_dist == 0.0)
    return BPF_OK;  // occupancy_map.cpp:141-145
  HIPCHK(e, hipSetDevice(e->device));
  const int sx = e->map.size_x, sy = e->map.size_y;
  const double res = e->map.resolution;
  const int radius = (int)std::floor(max_dist / res);
  std::vector<float> lut((size_t)sx * sy);
  std::vector<bool> marked((size_t)sx * sy, false);
  struct Cell
  {
    int i, j, si, sj;
    const float* lut;
    int sx;
    bool operator<(const Cell& b) const { return lut[i + (size_t)j * sx] > lut[b.i + (size_t)b.j * sx]; }
  };
  std::priority_queue<Cell> q;
  for (int i = 0; i < sx; ++i)
    for (int j = 0; j < sy; ++j)
    {
      const size_t idx = i + (size_t)j * sx;
      if (e->h_cells8[idx] == 1)
      {
        lut[idx] = 0.0f;
        marked[idx] = true;
        q.push(Cell{ i, j, i, j, lut.data(), sx });
      }
      else
        lut[idx] = (float)max_dist;
    }
  auto visit = [&](int i, int j, const Cell& cur) {
    const size_t idx = i + (size_t)j * sx;
    if (marked[idx])
      return;
    const int di = std::abs(i - cur.si), dj = std::abs(j - cur.sj);
    const double d = std::sqrt((double)(di * di + dj * dj));
    if (d <= radius)
    {
      lut[idx] = (float)(d * res);
      q.push(Cell{ i, j, cur.si, cur.sj, lut.data(), sx });
      marked[idx] = true;
    }
  };
  while (!q.empty())
  {
    const Cell cur = q.top();
    if (cur.i > 0)
      visit(cur.i - 1, cur.j, cur);
    if (cur.j > 0)
      visit(cur.i, cur.j - 1, cur);
    if (cur.i < sx - 1)
      visit(cur.i + 1, cur.j, cur);
    if (cur.j < sy - 1)
      visit(cur.i, cur.j + 1, cur);
    q.pop();
  }
  e->map.max_dist = max_dist;
  return encode_lut(e, lut.data());
}
