#ifndef BPF_BEAM_BLOCK
#define BPF_BEAM_BLOCK 512
#endif
// Host half of the planar scoring path: scan staging ring, beam / term tables, kernel launches, beam skipping.
// ------------------------------------------------------------------ scan staging
int acquire_slot(bpf_engine* e, size_t bytes, ScanSlot** out)
{
  ScanSlot& s = e->ring[e->ring_next];
  e->ring_next = (e->ring_next + 1) % kRing;
  if (s.pending)
  {
    HIPCHK(e, hipEventSynchronize(s.done));
    s.pending = false;
  }
  if (!s.done)
    HIPCHK(e, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  HIPCHK(e, s.host.reserve((bytes + 31) & ~(size_t)15));  // whole 16-byte words are copied
  HIPCHK(e, s.dev.reserve((bytes + 31) & ~(size_t)15));
  *out = &s;
  return BPF_OK;
}

int release_slot(bpf_engine* e, ScanSlot* s)
{
  HIPCHK(e, hipEventRecord(s->done, e->stream));
  s->pending = true;
  return BPF_OK;
}

// Host half of calcLikelihoodFieldModel{,Prob,Gompertz}: beam decimation and validity
// (planar_scanner.cpp:265-282, :339-343,410-425, :578-597) and the per-level term table.
int stage_field_scan(bpf_engine* e, const double* ranges, const double* angles, int rc, double range_max,
                     ScanSlot** slot_out, FieldScan* fs, const std::vector<uint8_t>* keep_slot = nullptr)
{
  const PlanarModel& pm = e->pm;
  int step;
  if (pm.model == BPF_MODEL_LIKELIHOOD_FIELD_PROB)
    step = (int)std::ceil(rc / (double)pm.max_beams);
  else
    step = (rc - 1) / (pm.max_beams - 1);
  if (step < 1)
    step = 1;
  const int K = e->map.n_levels;
  fs->table_len = K + 1;
  if ((int)e->trig_angles.size() != rc || std::memcmp(e->trig_angles.data(), angles, (size_t)rc * sizeof(double)) != 0)
  {
    e->trig_angles.assign(angles, angles + rc);
    e->trig_cos.resize(rc);
    e->trig_sin.resize(rc);
    for (int i = 0; i < rc; ++i)
    {
      e->trig_cos[i] = std::cos(angles[i]);
      e->trig_sin[i] = std::sin(angles[i]);
    }
  }
  // This runs between the launch that wrote the poses and the prep launch, with the GPU waiting (9 us for 1081 beams as
  // one loop with its tests and push_backs): the products and the two divisions per beam first, for every decimated
  // beam and without a branch (the compiler vectorises it), then the tests, writing the kept beams straight into the
  // pinned slot: 4.7 us.  Same expressions, same bits.
  const int m_beams = (rc + step - 1) / step;
  if ((int)e->stage_bx.size() < m_beams)
  {
    e->stage_bx.resize(m_beams);
    e->stage_by.resize(m_beams);
  }
  const double res = e->map.resolution;
  {
    double* bx = e->stage_bx.data();
    double* by = e->stage_by.data();
    const double* tc = e->trig_cos.data();
    const double* ts = e->trig_sin.data();
    if (step == 1)
      for (int i = 0; i < rc; ++i)
      {
        bx[i] = (ranges[i] * tc[i]) / res;
        by[i] = (ranges[i] * ts[i]) / res;
      }
    else
      for (int k = 0, i = 0; i < rc; i += step, ++k)
      {
        bx[k] = (ranges[i] * tc[i]) / res;
        by[k] = (ranges[i] * ts[i]) / res;
      }
  }
  // the usual scan: every decimated beam is a valid return of sane length -- one branch-free pass finds that out, and
  // the kept beams are then all of them, in order (the loop with its tests below costs ~2 us per 1 000 beams with the
  // GPU waiting)
  bool all_kept = keep_slot == nullptr && m_beams <= kMaxBeams;
  if (all_kept)
  {
    int okc = 1;
    const double* bx = e->stage_bx.data();
    const double* by = e->stage_by.data();
    for (int k = 0, i = 0; i < rc; i += step, ++k)
      okc &= (int)(ranges[i] < range_max) & (int)(std::fabs(bx[k]) < 268435456.0) & (int)(std::fabs(by[k]) < 268435456.0);
    all_kept = okc != 0;  // (a NaN range fails the first test, a NaN or huge product the others)
  }
  fs->beams_off = 0;
  ScanSlot* s;
  int rcode = acquire_slot(e, (((size_t)m_beams * sizeof(double2) + 255) & ~(size_t)255) +
                                  (size_t)fs->table_len * sizeof(double), &s);  // room for every decimated beam
  if (rcode != BPF_OK)
    return rcode;
  double2* out = reinterpret_cast<double2*>(s->host.p + fs->beams_off);
  fs->slot_of.clear();
  fs->slot_of.reserve((size_t)m_beams);
  fs->n_valid = 0;
  fs->n_always_off = 0;
  int slot = 0, n_out = 0;
  if (all_kept)
  {
    const double* bx = e->stage_bx.data();
    const double* by = e->stage_by.data();
    for (int k = 0; k < m_beams; ++k)
    {
      out[k].x = bx[k];
      out[k].y = by[k];
    }
    fs->slot_of.resize((size_t)m_beams);
    for (int k = 0; k < m_beams; ++k)
      fs->slot_of[(size_t)k] = k;
    fs->n_valid = n_out = slot = m_beams;
  }
  for (int i = 0; !all_kept && i < rc; i += step, ++slot)
  {
    const double r = ranges[i];
    if (r >= range_max)
      continue;
    if (r != r)
      continue;
    ++fs->n_valid;
    if (keep_slot && !(slot < (int)keep_slot->size() && (*keep_slot)[slot]))
      continue;
    const double b_x = e->stage_bx[slot], b_y = e->stage_by[slot];
    // a non-finite or absurdly long beam (> 2^28 cells) ends off the map for every pose in the
    // reference ((int) of a NaN or huge double is INT_MIN on x86): it is not staged, its constant
    // off-map term is added in the epilogue instead
    if (!(std::fabs(b_x) < 268435456.0 && std::fabs(b_y) < 268435456.0))
    {
      ++fs->n_always_off;
      continue;
    }
    if (n_out < kMaxBeams)
    {
      out[n_out].x = b_x;
      out[n_out].y = b_y;
    }
    ++n_out;
    fs->slot_of.push_back(slot);
  }
  fs->n_slots = slot;
  fs->n_staged = n_out;
  if (fs->n_staged > kMaxBeams)
  {
    (void)release_slot(e, s);
    return e->fail(BPF_ERR_CAPACITY, "more than 4096 beams per scan after decimation");
  }
  fs->table_off = ((size_t)fs->n_staged * sizeof(double2) + 255) & ~(size_t)255;
  fs->bytes = fs->table_off + (size_t)fs->table_len * sizeof(double);
  double* table = reinterpret_cast<double*>(s->host.p + fs->table_off);
  bpf_engine::TermKey key;
  key.model = pm.model;
  key.map_version = e->map_version;
  key.z_hit = pm.z_hit;
  key.z_rand = pm.z_rand;
  key.sigma = pm.sigma_hit;
  key.range_max = range_max;
  const bool table_cached = key == e->term_key && (int)e->term_table.size() == K + 1;
  const double denom = 2 * pm.sigma_hit * pm.sigma_hit;
  const double rand_mult = 1.0 / range_max;
  for (int k = 0; k <= K && !table_cached; ++k)
  {
    const bool off_map = (k == K);
    const double z = off_map ? e->map.max_dist : (double)e->h_levels[k];
    double pz = 0.0;
    if (pm.model == BPF_MODEL_LIKELIHOOD_FIELD)
    {
      pz += pm.z_hit * std::exp(-(z * z) / denom);
      pz += pm.z_rand * rand_mult;
      table[k] = pz * pz * pz;
    }
    else if (pm.model == BPF_MODEL_LIKELIHOOD_FIELD_GOMPERTZ)
    {
      pz += pm.z_hit * std::exp(-(z * z) / denom);
      pz += pm.z_rand;
      table[k] = pz;
    }
    else
    {
      if (off_map)
      {
        const double max_dist_prob = std::exp(-(e->map.max_dist * e->map.max_dist) / denom);
        pz += pm.z_hit * max_dist_prob;
      }
      else
        pz += pm.z_hit * std::exp(-(z * z) / denom);
      pz += pm.z_rand * rand_mult;
      table[k] = std::log(pz);
    }
  }
  if (table_cached)
    std::memcpy(table, e->term_table.data(), (size_t)(K + 1) * sizeof(double));
  else
  {
    e->term_table.assign(table, table + K + 1);
    e->term_key = key;
  }
  fs->off_map_term = table[K];
  // the copy to the device slot is done by k_field_prep (launch_field) unless the staging block is
  // larger than what its grid covers
  fs->copy_pending = true;
  *slot_out = s;
  return BPF_OK;
}

// aos != nullptr: the n particles are still 32-byte records at `aos` (device memory); the prep launch unpacks them into
// `p` as it goes (k_field_prep_aos) -- the chunks of the pipelined host-buffer seam (abi_planar.inl).
// host_out != nullptr: the HOST_OUT form of the scoring kernel (weights to a pinned array as well, a done word).
struct FieldHostOut
{
  double4* rec;      // non-null: HOST_MODE 2 -- the caller's records in registered host memory, read and written in place
  double* w_host;
  double* total_host;
  double* partials;  // >= blocks of the launch
  unsigned long long* flag;
  unsigned long long value;
};
int launch_field(bpf_engine* e, ParticlesDev p, int n, ScanSlot* s, const FieldScan& fs, int* obs_count,
                 int skip_level, bool want_partials = false, const double4* aos = nullptr,
                 const FieldHostOut* host_out = nullptr)
{
  FieldScoreArgs A{};
  A.p = p;
  A.n = n;
  A.beams = reinterpret_cast<const double2*>(s->dev.p + fs.beams_off);
  A.n_beams = fs.n_staged;
  A.table = reinterpret_cast<const double*>(s->dev.p + fs.table_off);
  A.table_len = fs.table_len;
  A.map = e->map;
  A.sp_x = e->pm.pose[0];
  A.sp_y = e->pm.pose[1];
  A.sp_th = e->pm.pose[2];
  A.off_map_factor = e->pm.off_map_factor;
  A.non_free_factor = e->pm.non_free_factor;
  A.non_free_radius = e->pm.non_free_radius;
  A.model = e->pm.model;
  A.g = e->pm.g;
  A.n_valid = fs.n_valid;
  A.obs_count = obs_count;
  A.skip_level = skip_level;
  A.extra_term = 0.0;
  for (int k = 0; k < fs.n_always_off; ++k)
    A.extra_term += fs.off_map_term;
  const bool count_only = obs_count != nullptr;
  const bool table_lds = !count_only && fs.table_len <= kTableLdsMax;
  const size_t table_bytes = table_lds ? (((size_t)fs.table_len * sizeof(double) + 15) & ~(size_t)15) : 0;
  // at least the four block partials that reuse the head of the block (kernels_score.hpp)
  // (+ the record stash of HOST_MODE 2: 4 waves x 16 records behind the beams)
  const size_t lds = std::max<size_t>(32, (size_t)fs.n_staged * sizeof(double2) + table_bytes) +
                     (host_out != nullptr && host_out->rec != nullptr ? (size_t)(2048 + 32) : 0);
  // per-particle scanner pose / trig once per update (shared by both scoring forms)
  const int prep_blocks = blocks_for(n, 256);
  HIPCHK(e, e->d_prep.reserve((size_t)n));
  HIPCHK(e, e->d_prep_stats.reserve((size_t)prep_blocks * kPrepStats));
  const bool rec_mode = host_out != nullptr && host_out->rec != nullptr;
  // Tile-sorted scoring (HOST_MODE 3 of k_score_field): for a cloud that the previous resample found spread (it ran to
  // the end without a KLD stop: window_hint, as for the CDF's guide table) or that was just drawn uniformly over the
  // map, on a map whose LUT image does not fit an XCD's L2.
  const size_t lut_bytes = (size_t)e->map.ltx * e->map.lty * 128;
  const bool tile_mode = e->tile_sort && want_partials && !count_only && table_lds && host_out == nullptr &&
                         aos == nullptr && !e->window_enabled && e->graded_shares && (e->n_cu & 7) == 0 &&
                         n >= e->n_cu * 4 * 64 && lut_bytes > ((size_t)3 << 20) &&
                         (e->window_hint >= e->max_samples || e->spread_init);
  e->last_score_form = 0;
  if (host_out != nullptr)
  {
    A.w_host = host_out->w_host;
    A.rec = host_out->rec;
  }
  if (rec_mode)
  {
    // no prep launch to ride on: a small copy launch brings the scan's staging block over
    if (fs.copy_pending)
    {
      const int n16 = (int)((fs.bytes + 15) / 16);
      ProfScope pa(e, BPF_K_SCORE_AUX);
      hipLaunchKernelGGL(k_copy16, dim3(blocks_for(n16, 256)), dim3(256), 0, e->stream,
                         reinterpret_cast<const uint4*>(s->host.p), reinterpret_cast<uint4*>(s->dev.p), n16);
    }
  }
  else
  {
    const int n16 = (int)((fs.bytes + 15) / 16);
    const bool ride = fs.copy_pending && n16 <= prep_blocks * 256;
    if (fs.copy_pending && !ride)
      HIPCHK(e, hipMemcpyAsync(s->dev.p, s->host.p, fs.bytes, hipMemcpyHostToDevice, e->stream));
    ProfScope pa(e, BPF_K_SCORE_AUX);
    const uint4* src = ride ? reinterpret_cast<const uint4*>(s->host.p) : static_cast<const uint4*>(nullptr);
    if (tile_mode)
    {
      // prep + counting sort by map tile: bins of 2^shift cells with at most 64 bins per axis
      int shift = 5;
      while ((((e->map.size_x + 2) >> shift) + 1) > 64 || (((e->map.size_y + 2) >> shift) + 1) > 64)
        ++shift;
      const int txc = ((e->map.size_x + 2) >> shift) + 1, tyc = ((e->map.size_y + 2) >> shift) + 1;
      const bool fresh = e->d_tile_int.cap < (size_t)3 * kTileBins + 2 * (size_t)n;
      HIPCHK(e, e->d_tile_int.reserve((size_t)3 * kTileBins + 2 * (size_t)n));
      HIPCHK(e, e->d_prep_sorted.reserve((size_t)n));
      if (fresh)  // both halves of the histogram start at zero; from then on each update zeroes the other half
        HIPCHK(e, hipMemsetAsync(e->d_tile_int.p, 0, 3 * kTileBins * sizeof(int), e->stream));
      e->tile_parity ^= 1;
      int* hist = e->d_tile_int.p + (size_t)e->tile_parity * kTileBins;
      int* hist_next = e->d_tile_int.p + (size_t)(e->tile_parity ^ 1) * kTileBins;
      int* cursor = e->d_tile_int.p + 2 * kTileBins;
      int* tile = cursor + kTileBins;
      int* perm = tile + n;
      hipLaunchKernelGGL(k_field_prep_tile, dim3(prep_blocks), dim3(256), 0, e->stream, p, n, e->map, A.sp_x, A.sp_y,
                         A.sp_th, e->d_prep.p, tile, hist, hist_next, cursor, shift, txc, tyc, src,
                         reinterpret_cast<uint4*>(s->dev.p), n16);
      hipLaunchKernelGGL(k_tile_scatter, dim3(prep_blocks), dim3(256), 0, e->stream, n, (const int*)tile,
                         (const int*)hist, cursor, (const double4*)e->d_prep.p, perm, e->d_prep_sorted.p);
      A.perm = perm;
    }
    else if (aos != nullptr)
      hipLaunchKernelGGL(k_field_prep_aos, dim3(prep_blocks), dim3(256), 0, e->stream, aos, p, n, e->map, A.sp_x, A.sp_y,
                         A.sp_th, e->d_prep.p, src, reinterpret_cast<uint4*>(s->dev.p), n16);
    else if (e->window_enabled)
      hipLaunchKernelGGL(k_field_prep<true>, dim3(prep_blocks), dim3(256), 0, e->stream, p, n, e->map, A.sp_x, A.sp_y,
                         A.sp_th, e->d_prep.p, e->d_prep_stats.p, src, reinterpret_cast<uint4*>(s->dev.p), n16);
    else
      hipLaunchKernelGGL(k_field_prep<false>, dim3(prep_blocks), dim3(256), 0, e->stream, p, n, e->map, A.sp_x, A.sp_y,
                         A.sp_th, e->d_prep.p, e->d_prep_stats.p, src, reinterpret_cast<uint4*>(s->dev.p), n16);
  }
  A.prep = tile_mode ? e->d_prep_sorted.p : e->d_prep.p;
  // One resident round: blocks per CU = what registers, LDS and the SGPR rule admit (the occupancy
  // API can over-report by one block for SGPR-heavy kernels: MI355X_MICROARCH.md, residency).
  int api_blocks = 0;
  if (host_out != nullptr && (count_only || !table_lds))
    return e->fail(BPF_ERR_UNSUPPORTED, "host-out scoring kernel: table-in-LDS scoring form only");
  const void* kfn = count_only ? reinterpret_cast<const void*>(&k_score_field<true, false>)
                               : tile_mode ? reinterpret_cast<const void*>(&k_score_field<false, true, 3>)
                               : (rec_mode ? reinterpret_cast<const void*>(&k_score_field<false, true, 2>)
                                  : host_out != nullptr ? reinterpret_cast<const void*>(&k_score_field<false, true, 1>)
                                  : (table_lds ? reinterpret_cast<const void*>(&k_score_field<false, true>)
                                               : reinterpret_cast<const void*>(&k_score_field<false, false>)));
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&api_blocks, kfn, 256, lds) != hipSuccess || api_blocks < 1)
    api_blocks = 1;
  const int per_cu = std::max(1, std::min(api_blocks, 6));
  const int resident_waves = e->n_cu * per_cu * 4;
  A.per_wave = std::max(1, blocks_for(n, resident_waves));
  int grid = std::max(1, blocks_for(blocks_for(n, A.per_wave), 4));
  for (int k = 0; k < 8; ++k)
    A.share_count[k] = A.share_base[k] = 0;
  A.blocks_per_round = e->n_cu;
  if (e->graded_shares && !count_only && per_cu == 4 && n >= e->n_cu * 4 * 64)
  {
    // graded partition (see k_score_field): the share of a wave by the placement round of its block.  The shares
    // were measured on MI355X at 4 blocks per CU (tools/phase_timing.py): 40 / 28 / 19 / 13 % of a SIMD's particles
    // make every wave end within a few microseconds of the others (equal shares: 46 .. 86 us).
    // (HOST_MODE 2, whose records arrive over PCIe in dispatch order while the kernel runs, was measured with steeper
    // and with equal shares: 160 us with these, 169-177 with 46/28/16/10 ... 60/24/11/5, 173 with equal shares -- the
    // launch is bound by the 6.4 MB it moves over PCIe, not by the skew of its waves)
    static const double kShare[4] = { 0.40, 0.28, 0.19, 0.13 };
    const double per_simd = (double)n / (e->n_cu * 4);
    int base = 0;
    for (int k = 0; k < 4; ++k)
    {
      A.share_count[k] = std::max(1, (int)std::ceil(per_simd * kShare[k]));
      A.share_base[k] = base;
      base += A.share_count[k] * e->n_cu * 4;
    }
    grid = e->n_cu * per_cu;
    A.xcd_local = tile_mode ? 1 : 0;
  }
  const bool tile_run = tile_mode && A.xcd_local != 0;  // (the graded partition applies: four blocks per CU)
  if (tile_mode && !tile_run)
  {
    // the occupancy came out differently: walk the slots in tile order all the same, without the XCD-wise split
    A.xcd_local = 0;
  }
  if (host_out != nullptr && grid > kSeamMaxBlocks)
    return e->fail(BPF_ERR_CAPACITY, "host-out scoring launch: more blocks than the partials buffer holds");
  A.block_partials = host_out != nullptr ? host_out->partials : nullptr;
  A.skip_if_set = nullptr;
  e->last_used_window_path = false;
  if (tile_mode)
  {
    // no block partials: which particles share a block depends on the order inside a tile (atomics); the total is a
    // fixed-shape sum over the weights in index order, launched behind the scoring kernel below
    A.block_partials = nullptr;
  }
  else if (want_partials && !count_only)
  {
    HIPCHK(e, e->d_block_partials.reserve((size_t)grid));
    A.block_partials = e->d_block_partials.p;
    e->fused_partials = grid;
    // LDS-window path for big updates: the device decides (from the cloud's spread) whether the
    // window kernels or k_score_field do the work; the other one returns immediately.
    const int n_chunks = (fs.n_staged + 63) / 64;
    const size_t win_lds = (size_t)kWinDim * kWinDim * sizeof(uint16_t) + ((size_t)fs.table_len + 1) * 8 + 64 * 16;
    if (e->window_enabled && n >= 16384 && fs.n_staged >= 64 && n_chunks <= kMaxChunks && table_lds &&
        fs.table_len <= 2047 && win_lds <= 160 * 1024)
    {
      HIPCHK(e, e->d_chunk_partials.reserve((size_t)n_chunks * n));
      HIPCHK(e, e->d_plan.reserve(1));
      if (!e->window_lds_attr_set)
      {
        HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score_window),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        e->window_lds_attr_set = true;
      }
      {
        ProfScope pa(e, BPF_K_SCORE_AUX);
        hipLaunchKernelGGL(k_field_windows, dim3(1), dim3(1024), 0, e->stream, e->d_prep_stats.p, prep_blocks,
                           A.beams, fs.n_staged, e->map, e->d_plan.p);
      }
      WindowScoreArgs W{};
      W.n = n;
      W.prep = e->d_prep.p;
      W.beams = A.beams;
      W.n_beams = fs.n_staged;
      W.table = A.table;
      W.table_len = fs.table_len;
      W.map = e->map;
      W.plan = e->d_plan.p;
      W.partials = e->d_chunk_partials.p;
      W.slabs = std::max(1, e->n_cu / n_chunks);
      {
        ProfScope pw(e, BPF_K_SCORE_WINDOW);
        hipLaunchKernelGGL(k_score_window, dim3(n_chunks, W.slabs), dim3(kWinThreads), win_lds, e->stream, W);
      }
      FieldFinishArgs F{};
      F.p = p;
      F.n = n;
      F.partials = e->d_chunk_partials.p;
      F.plan = e->d_plan.p;
      F.map = e->map;
      F.off_map_factor = A.off_map_factor;
      F.non_free_factor = A.non_free_factor;
      F.non_free_radius = A.non_free_radius;
      F.model = A.model;
      F.g = A.g;
      F.n_valid = A.n_valid;
      F.extra_term = A.extra_term;
      F.block_partials = A.block_partials;
      {
        ProfScope pa(e, BPF_K_SCORE_AUX);
        hipLaunchKernelGGL(k_field_finish, dim3(grid), dim3(256), 0, e->stream, F);
      }
      HIPCHK(e, hipGetLastError());
      A.skip_if_set = &e->d_plan.p->use_window;
      e->last_used_window_path = true;
    }
  }
  if (count_only)
    LAUNCH_TIMED(e, BPF_K_SCORE, (k_score_field<true, false>), dim3(grid), dim3(256), lds, A);
  else if (host_out != nullptr)
  {
    if (rec_mode)
      LAUNCH_TIMED(e, BPF_K_SCORE, (k_score_field<false, true, 2>), dim3(grid), dim3(256), lds, A);
    else
      LAUNCH_TIMED(e, BPF_K_SCORE, (k_score_field<false, true, 1>), dim3(grid), dim3(256), lds, A);
    hipLaunchKernelGGL(k_seam_done, dim3(1), dim3(256), 0, e->stream, (const double*)host_out->partials, grid,
                       host_out->total_host, host_out->flag, host_out->value);
  }
  else if (tile_mode)
  {
    LAUNCH_TIMED(e, BPF_K_SCORE, (k_score_field<false, true, 3>), dim3(grid), dim3(256), lds, A);
    const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
    HIPCHK(e, e->d_block_partials.reserve((size_t)nb));
    hipLaunchKernelGGL(k_sum_partials, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, (const double*)p.w, n,
                       e->d_block_partials.p);
    e->fused_partials = nb;  // the normalise launch folds these instead of the scoring kernel's
    e->last_score_form = 3;
  }
  else if (table_lds)
    LAUNCH_TIMED(e, BPF_K_SCORE, (k_score_field<false, true>), dim3(grid), dim3(256), lds, A);
  else
    LAUNCH_TIMED(e, BPF_K_SCORE, (k_score_field<false, false>), dim3(grid), dim3(256), lds, A);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

// dst: where the scalars live (default: the engine's device block; the host-buffer seam passes the pinned host copy so
// that the total needs no copy behind it)
int sum_into_slot(bpf_engine* e, const double* v, int n, int slot, int update_averages, int n_samples,
                  FilterScalars* dst = nullptr, unsigned long long* done_flag = nullptr,
                  unsigned long long done_value = 0ull)
{
  const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
  HIPCHK(e, e->d_partials.reserve((size_t)nb));
  ProfScope ps(e, BPF_K_REDUCE);
  hipLaunchKernelGGL(k_sum_partials, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, v, n, e->d_partials.p);
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(BPF_RED_BLOCK), 0, e->stream, e->d_partials.p, nb,
                     dst ? dst : e->d_scalars.p, slot, update_averages, n_samples, e->alpha_slow, e->alpha_fast,
                     done_flag, done_value);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

int ensure_scalars(bpf_engine* e)
{
  if (e->d_scalars.p)
    return BPF_OK;
  HIPCHK(e, e->d_scalars.reserve(1));
  HIPCHK(e, hipMemsetAsync(e->d_scalars.p, 0, sizeof(FilterScalars), e->stream));
  HIPCHK(e, e->h_scalars.reserve(1));
  HIPCHK(e, e->d_flags.reserve(8));
  HIPCHK(e, hipMemsetAsync(e->d_flags.p, 0, 8 * sizeof(int), e->stream));
  HIPCHK(e, e->h_flags.reserve(8));
  HIPCHK(e, e->h_done.reserve(16));
  e->h_done.p[0] = 0;
  e->zero_copy_keys = getenv("BPF_NO_ZEROCOPY") == nullptr;
  return BPF_OK;
}

// Second half of beam skipping: mask from the (possibly shard-summed) counts in d_obs_count over
// `n_total` particles, then pass 2 over this engine's `n` particles.
int score_planar_beamskip_finish(bpf_engine* e, ParticlesDev p, int n, long long n_total, const double* ranges,
                                 const double* angles, int rc, double range_max, bool* forced_zero, bool want_partials)
{
  const PlanarModel& pm = e->pm;
  const FieldScan& fs = e->skip_fs;
  e->skip_pending = false;
  const int nv = std::max(fs.n_staged, 1);
  std::vector<int> counts((size_t)nv, 0);
  HIPCHK(e, hipMemcpyAsync(counts.data(), e->d_obs_count.p, (size_t)fs.n_staged * sizeof(int), hipMemcpyDeviceToHost,
                           e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  std::vector<int> obs_count((size_t)pm.max_beams, 0);
  for (int v = 0; v < fs.n_staged; ++v)
    if (fs.slot_of[v] < pm.max_beams)
      obs_count[fs.slot_of[v]] = counts[v];
  std::vector<uint8_t> mask_slot((size_t)pm.max_beams, 0);
  int skipped = 0;
  for (int b = 0; b < pm.max_beams; ++b)
  {
    if ((obs_count[b] / (double)n_total) > pm.beam_skip_threshold)
      mask_slot[b] = 1;
    else
      skipped++;
  }
  const bool error = skipped >= (pm.max_beams * pm.beam_skip_error_threshold);
  // A kept slot that was never written holds 0.0 in the reference's scratch matrix, and
  // log(0) = -inf zeroes every weight (planar_scanner.cpp:519-527).
  std::vector<uint8_t> visited((size_t)pm.max_beams, 0);
  for (int v = 0; v < fs.n_staged; ++v)
    if (fs.slot_of[v] < pm.max_beams)
      visited[fs.slot_of[v]] = 1;
  bool poisoned = false;
  for (int b = 0; b < pm.max_beams; ++b)
    if ((error || mask_slot[b]) && !visited[b])
      poisoned = true;
  if (poisoned)
  {
    hipLaunchKernelGGL(k_fill, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, p.w, 0.0, n);
    HIPCHK(e, hipGetLastError());
    *forced_zero = true;
    return BPF_OK;
  }
  // pass 2: stage only the kept beams (all of them when the error switch tripped) and score
  std::vector<uint8_t> keep((size_t)std::max(fs.n_slots, pm.max_beams), 0);
  for (size_t b = 0; b < keep.size(); ++b)
    keep[b] = error ? 1 : ((int)b < pm.max_beams ? mask_slot[b] : 0);
  ScanSlot* s2 = nullptr;
  FieldScan fs2;
  int rcode = stage_field_scan(e, ranges, angles, rc, range_max, &s2, &fs2, &keep);
  if (rcode != BPF_OK)
    return rcode;
  rcode = launch_field(e, p, n, s2, fs2, nullptr, 0, want_partials);
  if (rcode != BPF_OK)
    return rcode;
  return release_slot(e, s2);
}

// Scores `n` particles of `p` with the configured planar model (+ recalcWeight).  Leaves the
// weights un-normalised.  set_converged feeds the prob model's beam-skip switch.  defer_beamskip_pass2: stop
// after the counting pass of beam skipping (e->skip_pending is then set) so that a sharded driver can sum the
// counts over the shards before score_planar_beamskip_finish.
// aos != nullptr: the n particles are still 32-byte records at `aos` (device memory, as the copy engine left them);
// they are unpacked into `p` on the way (by the prep launch of the likelihood-field family, by a launch of its own
// for the other forms).
int score_planar(bpf_engine* e, ParticlesDev p, int n, int set_converged, const double* ranges,
                 const double* angles, int rc, double range_max, bool* forced_zero, bool want_partials = false,
                 bool defer_beamskip_pass2 = false, const double4* aos = nullptr)
{
  *forced_zero = false;
  e->fused_partials = 0;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  if (!e->have_map)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no 2-D map set");
  if (!e->have_lut)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "distance LUT missing (reference: isMapInitialized, node_2d.cpp:406-410)");
  if (!e->pm.configured)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "planar model not set");
  if (rc <= 0 || ranges == nullptr || angles == nullptr || n <= 0)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "empty scan or sample set");
  const PlanarModel& pm = e->pm;
  e->evals_last = 0;
  if (aos != nullptr && (pm.model == BPF_MODEL_BEAM ||
                         (pm.model == BPF_MODEL_LIKELIHOOD_FIELD_PROB && pm.do_beamskip && set_converged)))
  {
    hipLaunchKernelGGL(k_aos_to_soa, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, aos, p, n);
    HIPCHK(e, hipGetLastError());
    aos = nullptr;
  }

  if (pm.model == BPF_MODEL_BEAM)
  {
    const int step = (rc - 1) / (pm.max_beams - 1);  // planar_scanner.cpp:193, not clamped
    if (step < 1)
      return e->fail(BPF_ERR_BEAM_STEP, "beam model: range_count < max_beams makes the reference loop forever");
    std::vector<BeamRec> beams;
    for (int i = 0; i < rc; i += step)
    {
      BeamRec b;
      b.cb = std::cos(angles[i]);
      b.sb = std::sin(angles[i]);
      b.obs = ranges[i];
      b.short_t = pm.z_short * pm.lambda_short * std::exp(-pm.lambda_short * ranges[i]);
      b.tail_t = 0.0;
      if (ranges[i] == range_max)
        b.tail_t = pm.z_max * 1.0;
      if (ranges[i] < range_max)
        b.tail_t = pm.z_rand * 1.0 / range_max;
      beams.push_back(b);
    }
    if ((int)beams.size() > kMaxBeams)
      return e->fail(BPF_ERR_CAPACITY, "more than 4096 beams per scan after decimation");
    // the ray walk forms cell offsets with 24-bit multiply-adds and 32-bit offsets (calc_range_skip); its error term
    // j * 2 * dmin (j <= dmaj + 1, dmin <= dmaj <= range_max / resolution + 1) must stay below 2^31
    if (!(range_max / e->map.resolution < kMaxRayCells) ||
        (long long)(e->map.size_x + 2) * (long long)(e->map.size_y + 2) >= (1ll << 30) || e->map.size_x + 3 >= (1 << 21))
      return e->fail(BPF_ERR_CAPACITY, "beam model: range_max beyond 32 760 cells, or a map of 2^30 cells or more");
    // The beams stay in bearing order: one trip of a wave then casts 64 neighbouring bearings from one pose, which
    // pass much the same cells (measured: 2.73 ms against 2.98 ms with the beams ordered by observed range).
    const size_t bytes = beams.size() * sizeof(BeamRec);
    ScanSlot* s;
    int rcode = acquire_slot(e, bytes, &s);
    if (rcode != BPF_OK)
      return rcode;
    std::memcpy(s->host.p, beams.data(), bytes);
    HIPCHK(e, hipMemcpyAsync(s->dev.p, s->host.p, bytes, hipMemcpyHostToDevice, e->stream));
    BeamModelArgs A{};
    A.p = p;
    A.n = n;
    A.beams = reinterpret_cast<const BeamRec*>(s->dev.p);
    A.n_beams = (int)beams.size();
    A.map = e->map;
    A.sp_x = pm.pose[0];
    A.sp_y = pm.pose[1];
    A.sp_th = pm.pose[2];
    A.off_map_factor = pm.off_map_factor;
    A.non_free_factor = pm.non_free_factor;
    A.non_free_radius = pm.non_free_radius;
    A.range_max = range_max;
    A.z_hit = pm.z_hit;
    A.denom = 2 * pm.sigma_hit * pm.sigma_hit;
    A.inv_resolution = 1.0 / e->map.resolution;
    A.cells_walked = nullptr;
    if (e->count_cells)
    {
      if (!e->d_cells_walked.p)
      {
        HIPCHK(e, e->d_cells_walked.reserve(1));
        HIPCHK(e, hipMemsetAsync(e->d_cells_walked.p, 0, sizeof(unsigned long long), e->stream));
      }
      A.cells_walked = e->d_cells_walked.p;
    }
    constexpr int kBeamBlock = BPF_BEAM_BLOCK;
    constexpr int kBeamWaves = kBeamBlock / 64;
    int api_blocks = 0;
    // rays of at most kIntDivRayCells cells: the integer form of the walk (calc_range_skip)
    const bool int_div = range_max / e->map.resolution <= (double)kIntDivRayCells;
    const void* kfn = int_div ? reinterpret_cast<const void*>(&k_score_beam<kBeamBlock, true>)
                              : reinterpret_cast<const void*>(&k_score_beam<kBeamBlock, false>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&api_blocks, kfn, kBeamBlock, bytes) != hipSuccess || api_blocks < 1)
      api_blocks = 1;
    const int per_cu = std::max(1, std::min(api_blocks, 32 / kBeamWaves));
    // one resident round of blocks; the waves fetch their particles from a counter (k_score_beam), a few at a time
    // so that the round drains evenly: ~16 grabs per wave, at most 16 particles each (100 k particles: 2 per grab;
    // 1, 2 and 4 measure alike, 8 costs 8 %, 16 costs 23 %)
    A.per_wave = std::max(1, std::min(16, blocks_for(n, e->n_cu * per_cu * kBeamWaves * 16)));
    const int grid = std::max(1, std::min(e->n_cu * per_cu, blocks_for(blocks_for(n, A.per_wave), kBeamWaves)));
    HIPCHK(e, e->d_beam_counter.reserve(1));
    HIPCHK(e, hipMemsetAsync(e->d_beam_counter.p, 0, sizeof(int), e->stream));
    A.next_particle = e->d_beam_counter.p;
    A.block_partials = nullptr;  // dynamic assignment: the total comes from the fixed-shape sum over the weights
    (void)want_partials;
    if (int_div)
      LAUNCH_TIMED(e, BPF_K_SCORE, (k_score_beam<kBeamBlock, true>), dim3(grid), dim3(kBeamBlock), bytes, A);
    else
      LAUNCH_TIMED(e, BPF_K_SCORE, (k_score_beam<kBeamBlock, false>), dim3(grid), dim3(kBeamBlock), bytes, A);
    HIPCHK(e, hipGetLastError());
    e->evals_last = (long long)n * (long long)beams.size();
    return release_slot(e, s);
  }

  ScanSlot* s = nullptr;
  FieldScan fs;
  int rcode = stage_field_scan(e, ranges, angles, rc, range_max, &s, &fs);
  if (rcode != BPF_OK)
    return rcode;
  e->evals_last = (long long)n * fs.n_valid;

  const bool beamskip = pm.model == BPF_MODEL_LIKELIHOOD_FIELD_PROB && pm.do_beamskip && set_converged;
  if (!beamskip)
  {
    rcode = launch_field(e, p, n, s, fs, nullptr, 0, want_partials, aos);
    if (rcode != BPF_OK)
      return rcode;
    return release_slot(e, s);
  }

  // Beam skipping (planar_scanner.cpp:352-395,482-529): pass 1 counts, per beam, the particles
  // whose end point lies within beam_skip_distance of an obstacle; the host forms the mask;
  // pass 2 integrates the kept beams.  (The reference stores every pz in an N x max_beams
  // scratch matrix between the passes; re-evaluating is cheaper than 8 B x N x beams of HBM.)
  const int nv = std::max(fs.n_staged, 1);
  HIPCHK(e, e->d_obs_count.reserve((size_t)nv));
  HIPCHK(e, hipMemsetAsync(e->d_obs_count.p, 0, (size_t)nv * sizeof(int), e->stream));
  int skip_level = 0;  // levels are ascending: z < d  <=>  level index < first level >= d
  while (skip_level < e->map.n_levels && (double)e->h_levels[skip_level] < pm.beam_skip_distance)
    ++skip_level;
  rcode = launch_field(e, p, n, s, fs, e->d_obs_count.p, skip_level);
  if (rcode != BPF_OK)
    return rcode;
  rcode = release_slot(e, s);
  if (rcode != BPF_OK)
    return rcode;
  e->skip_fs = fs;
  e->skip_pending = true;
  if (defer_beamskip_pass2)
    return BPF_OK;  // sharded: the per-beam counts are summed over the shards first
  return score_planar_beamskip_finish(e, p, n, n, ranges, angles, rc, range_max, forced_zero, want_partials);
}
