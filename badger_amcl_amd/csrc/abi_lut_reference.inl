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
  std::vector<uint8_t> marked((size_t)sx * sy, 0);
  // The queue orders cells by their LUT value (occupancy_map.h:64-72 compares distances_lut_ of the two cells).  A
  // cell is pushed once, right after its value is written, and the value never changes afterwards (marked), so the
  // value travels in the entry: the same comparisons, hence the same libstdc++ heap and the same order among equal
  // distances, without a dependent load per comparison (1.6-2.0 -> 0.45 s for a 2000 x 2000 map).
  struct Cell
  {
    float key;
    short si, sj;  // the obstacle cell this wave front started from, relative to (i, j): |.| <= radius
    int i, j;
    bool operator<(const Cell& b) const { return key > b.key; }
  };
  if (radius > 30000)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "max_dist / resolution too large for the LUT builder");
  std::vector<Cell> store;
  store.reserve((size_t)sx * sy / 4 + 1024);
  std::priority_queue<Cell> q(std::less<Cell>(), std::move(store));
  // sqrt(di^2 + dj^2) * res for the offsets a front can have: the reference's expression, formed once
  const int tdim = radius + 2;
  std::vector<double> dist_cells((size_t)tdim * tdim);
  for (int a = 0; a < tdim; ++a)
    for (int b = 0; b < tdim; ++b)
      dist_cells[(size_t)a * tdim + b] = std::sqrt((double)(a * a + b * b));
  for (int i = 0; i < sx; ++i)
    for (int j = 0; j < sy; ++j)
    {
      const size_t idx = i + (size_t)j * sx;
      if (e->h_cells8[idx] == 1)
      {
        lut[idx] = 0.0f;
        marked[idx] = 1;
        q.push(Cell{ 0.0f, 0, 0, i, j });
      }
      else
        lut[idx] = (float)max_dist;
    }
  auto visit = [&](int i, int j, int si, int sj) {
    const size_t idx = i + (size_t)j * sx;
    if (marked[idx])
      return;
    const int di = std::abs(i - si), dj = std::abs(j - sj);
    if (di >= tdim || dj >= tdim)
      return;  // farther than the radius in one axis alone
    const double d = dist_cells[(size_t)di * tdim + dj];
    if (d <= radius)
    {
      const float v = (float)(d * res);
      lut[idx] = v;
      q.push(Cell{ v, (short)(si - i), (short)(sj - j), i, j });
      marked[idx] = 1;
    }
  };
  while (!q.empty())
  {
    const Cell cur = q.top();
    const int si = cur.i + cur.si, sj = cur.j + cur.sj;
    if (cur.i > 0)
      visit(cur.i - 1, cur.j, si, sj);
    if (cur.j > 0)
      visit(cur.i, cur.j - 1, si, sj);
    if (cur.i < sx - 1)
      visit(cur.i + 1, cur.j, si, sj);
    if (cur.j < sy - 1)
      visit(cur.i, cur.j + 1, si, sj);
    q.pop();
  }
  e->map.max_dist = max_dist;
  return encode_lut(e, lut.data());
}
