// C-ABI: 2-D map.
// ---------------------------------------------------------------------- 2-D map
int bpf_map2d_set(bpf_engine* e, const int32_t* cells, const float* dist_lut, int size_x, int size_y, float origin_x,
                  float origin_y, double resolution, double max_dist)
{
  if (!e || !cells || size_x <= 0 || size_y <= 0 || !(resolution > 0))
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad map arguments") : BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  const size_t ncell = (size_t)size_x * size_y;
  MapDev& M = e->map;
  M.size_x = size_x;
  M.size_y = size_y;
  M.ltx = (size_x + 2 + 7) / 8 + 1;  // padded cells 0 .. size+1, plus a spare tile column for 8-aligned windows
  M.lty = (size_y + 2 + 7) / 8;
  if ((size_t)M.ltx * 16 >= (1u << 24) || (size_t)M.ltx * M.lty * 128 >= (1ull << 32))
    return e->fail(BPF_ERR_CAPACITY, "map too large for the 32-bit tiled LUT addressing");
  M.half_x = size_x / 2;
  M.half_y = size_y / 2;
  M.origin_x = (double)origin_x;
  M.origin_y = (double)origin_y;
  M.resolution = resolution;
  M.max_dist = max_dist;
  M.n_levels = 0;
  e->h_cells8.resize(ncell);
  for (size_t i = 0; i < ncell; ++i)
    e->h_cells8[i] = (int8_t)cells[i];
  // Raycast grid, one 32-bit word per cell of the map padded by one cell all round: byte q = chessboard distance to
  // the nearest blocked cell (not FREE, or the ring) inside QUADRANT q of the cell (q bit 0: towards -x, bit 1:
  // towards -y; axes included), capped at 255.  A Bresenham line that heads into quadrant q only ever visits cells of
  // that quadrant, so the next D - 1 cells of the line are free.  D(c) = 0 if blocked, else
  // 1 + min(D(c + sx), D(c + sy), D(c + sx + sy)): exact for the chessboard metric restricted to a quadrant (each
  // of the three neighbours' quadrants lies inside c's), one sweep from the far corner per quadrant.
  const int pw = size_x + 2, ph = size_y + 2;
  std::vector<uint32_t> cheb((size_t)pw * ph, 0u);
  {
    std::vector<uint8_t> blocked((size_t)pw * ph, 1);
    for (int j = 0; j < size_y; ++j)
      for (int i = 0; i < size_x; ++i)
        if (cells[i + (size_t)j * size_x] == -1)
          blocked[(size_t)(j + 1) * pw + (i + 1)] = 0;
    std::vector<uint8_t> d((size_t)pw * ph);
    for (int q = 0; q < 4; ++q)
    {
      const int sx = (q & 1) ? -1 : 1, sy = (q & 2) ? -1 : 1;
      std::fill(d.begin(), d.end(), 0);
      for (int yy = 1; yy < ph - 1; ++yy)
      {
        const int y = (sy > 0) ? ph - 1 - yy : yy;
        for (int xx = 1; xx < pw - 1; ++xx)
        {
          const int x = (sx > 0) ? pw - 1 - xx : xx;
          const size_t c = (size_t)y * pw + x;
          if (blocked[c])
            continue;
          const int a = d[c + sx], b = d[(size_t)(y + sy) * pw + x], cc = d[(size_t)(y + sy) * pw + x + sx];
          d[c] = (uint8_t)std::min(255, 1 + std::min(a, std::min(b, cc)));
        }
      }
      for (size_t c = 0; c < cheb.size(); ++c)
        cheb[c] |= (uint32_t)d[c] << (8 * q);
    }
  }
  HIPCHK(e, e->d_cells8.reserve(ncell));
  H2D_OR_RETURN(h2d_from_host_sync(e, e->d_cells8.p, e->h_cells8.data(), ncell));
  HIPCHK(e, e->d_cheb.reserve(cheb.size()));
  H2D_OR_RETURN(h2d_from_host_sync(e, e->d_cheb.p, cheb.data(), cheb.size() * sizeof(uint32_t)));
  M.cells8 = e->d_cells8.p;
  M.cheb = e->d_cheb.p;
  M.lut_tiles = nullptr;
  M.levels = nullptr;
  e->have_map = true;
  e->have_lut = false;
  e->map_version++;
  if (dist_lut)
    return encode_lut(e, dist_lut);
  return BPF_OK;
}

int bpf_map2d_build_distances_lut(bpf_engine* e, double max_dist)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  return build_lut_device(e, max_dist);
}

int bpf_map2d_calc_range(bpf_engine* e, const double* ox, const double* oy, const double* cos_a, const double* sin_a,
                         const double* max_range, int n, double* range_out)
{
  if (!e || !ox || !oy || !cos_a || !sin_a || !max_range || !range_out || n <= 0)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "calc_range: null argument or n <= 0") : BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_map)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no 2-D map set");
  // the walk forms cell offsets with 24-bit multiply-adds and 32-bit offsets (calc_range_skip)
  if ((long long)(e->map.size_x + 2) * (long long)(e->map.size_y + 2) >= (1ll << 30) || e->map.size_x + 3 >= (1 << 21))
    return e->fail(BPF_ERR_CAPACITY, "calc_range: a map of 2^30 cells or more");
  for (int i = 0; i < n; ++i)
    if (!(std::fabs(max_range[i]) / e->map.resolution < kMaxRayCells))
      return e->fail(BPF_ERR_CAPACITY, "calc_range: max_range beyond 32 760 cells (or not finite)");
  HIPCHK(e, hipSetDevice(e->device));
  DevBuf<double> in, out;
  HIPCHK(e, in.reserve((size_t)5 * n));
  HIPCHK(e, out.reserve((size_t)n));
  const double* src[5] = { ox, oy, cos_a, sin_a, max_range };
  for (int k = 0; k < 5; ++k)
    H2D_OR_RETURN(h2d_from_host(e, in.p + (size_t)k * n, src[k], (size_t)n * sizeof(double), e->stream));
  hipLaunchKernelGGL(k_calc_range, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->map, in.p, in.p + n,
                     in.p + 2 * (size_t)n, in.p + 3 * (size_t)n, in.p + 4 * (size_t)n, n, out.p);
  HIPCHK(e, hipGetLastError());
  H2D_OR_RETURN(d2h_to_host(e, range_out, out.p, (size_t)n * sizeof(double), e->stream));
  return BPF_OK;
}

int bpf_map2d_get_distances_lut(bpf_engine* e, float* out, size_t capacity)
{
  if (!e || !out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_lut)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no distance LUT");
  const size_t ncell = (size_t)e->map.size_x * e->map.size_y;
  if (capacity < ncell)
    return e->fail(BPF_ERR_CAPACITY, "output too small");
  HIPCHK(e, hipSetDevice(e->device));
  H2D_OR_RETURN(d2h_to_host(e, out, e->d_lut_f32.p, ncell * sizeof(float), e->stream));
  return BPF_OK;
}
