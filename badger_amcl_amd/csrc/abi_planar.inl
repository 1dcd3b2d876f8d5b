// C-ABI: planar scanner.
// ---------------------------------------------------------------------- planar scanner
int bpf_planar_init(bpf_engine* e, int max_beams)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->pm.max_beams = max_beams;
  return BPF_OK;
}

int bpf_map2d_build_distances_lut_reference(bpf_engine* e, double max_dist);  // abi_lut_reference.inl

static int need_lut_for(bpf_engine* e, double max_dist)
{
  // setModelLikelihoodField* call map_->updateDistancesLUT(max_dist) (planar_scanner.cpp:74,91,112).
  // A LUT already there for the same max_dist is kept; otherwise it is built as the reference builds it (host
  // brushfire, the reference's values) unless the caller opted for the exact EDT on the device.
  if (!e->have_map)
    return BPF_OK;  // model may be set before the map; the LUT is then required at scoring time
  if (e->have_lut && e->map.max_dist == max_dist)
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->lut_exact_edt)
    return bpf_map2d_build_distances_lut_reference(e, max_dist);
  return build_lut_device(e, max_dist);
}

int bpf_planar_set_model_beam(bpf_engine* e, double z_hit, double z_short, double z_max, double z_rand,
                              double sigma_hit, double lambda_short)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  PlanarModel& p = e->pm;
  p.model = BPF_MODEL_BEAM;
  p.z_hit = z_hit; p.z_short = z_short; p.z_max = z_max; p.z_rand = z_rand;
  p.sigma_hit = sigma_hit; p.lambda_short = lambda_short;
  p.configured = true;
  return BPF_OK;
}

int bpf_planar_set_model_likelihood_field(bpf_engine* e, double z_hit, double z_rand, double sigma_hit,
                                          double max_distance_to_object)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  PlanarModel& p = e->pm;
  p.model = BPF_MODEL_LIKELIHOOD_FIELD;
  p.z_hit = z_hit; p.z_rand = z_rand; p.sigma_hit = sigma_hit;
  p.configured = true;
  return need_lut_for(e, max_distance_to_object);
}

int bpf_planar_set_model_likelihood_field_prob(bpf_engine* e, double z_hit, double z_rand, double sigma_hit,
                                               double max_distance_to_object, int do_beamskip,
                                               double beam_skip_distance, double beam_skip_threshold,
                                               double beam_skip_error_threshold)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  PlanarModel& p = e->pm;
  p.model = BPF_MODEL_LIKELIHOOD_FIELD_PROB;
  p.z_hit = z_hit; p.z_rand = z_rand; p.sigma_hit = sigma_hit;
  p.do_beamskip = do_beamskip;
  p.beam_skip_distance = beam_skip_distance;
  p.beam_skip_threshold = beam_skip_threshold;
  p.beam_skip_error_threshold = beam_skip_error_threshold;
  p.configured = true;
  return need_lut_for(e, max_distance_to_object);
}

int bpf_planar_set_model_likelihood_field_gompertz(bpf_engine* e, double z_hit, double z_rand, double sigma_hit,
                                                   double max_distance_to_object, double gompertz_a,
                                                   double gompertz_b, double gompertz_c, double input_shift,
                                                   double input_scale, double output_shift)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  PlanarModel& p = e->pm;
  p.model = BPF_MODEL_LIKELIHOOD_FIELD_GOMPERTZ;
  p.z_hit = z_hit; p.z_rand = z_rand; p.sigma_hit = sigma_hit;
  p.g = GompertzDev{ gompertz_a, gompertz_b, gompertz_c, input_shift, input_scale, output_shift };
  p.configured = true;
  return need_lut_for(e, max_distance_to_object);
}

int bpf_planar_set_map_factors(bpf_engine* e, double off_map_factor, double non_free_space_factor,
                               double non_free_space_radius)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->pm.off_map_factor = off_map_factor;
  e->pm.non_free_factor = non_free_space_factor;
  e->pm.non_free_radius = non_free_space_radius;
  return BPF_OK;
}

int bpf_planar_set_scanner_pose(bpf_engine* e, const double pose[3])
{
  if (!e || !pose)
    return BPF_ERR_INVALID_ARGUMENT;
  std::memcpy(e->pm.pose, pose, 3 * sizeof(double));
  return BPF_OK;
}

#define HIPCHK_OR(e, call, bail)                   \
  do                                               \
  {                                                \
    hipError_t _r = (call);                        \
    if (_r != hipSuccess)                          \
      return bail((e)->fail_hip(_r, #call));       \
  } while (0)

namespace
{
// the host's side of a done word (kernels of some tens of microseconds): spin, and give up after 50 ms so that the
// caller can fall back to a stream synchronisation
bool seam_wait_word(const unsigned long long* word, unsigned long long value)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins)
  {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == value)
      return true;
    if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50))
      return false;
    __builtin_ia32_pause();
  }
}

// Seam A on a REGISTERED host buffer, one scoring launch and no copy in either direction (HOST_MODE 2 of
// k_score_field): the waves read the caller's records over PCIe as they reach them and write them back whole with the
// new weight; k_seam_done folds the total and publishes the word this thread polls.  Nothing is left for the host to
// do but wait.  Same per-particle arithmetic as the resident form (field_prep_of, same beams and table): the same
// weights bit for bit; the total is the fixed-shape sum of the launch's block partials.
int apply_model_records(bpf_engine* e, double* samples, int n, int set_converged, const double* ranges,
                        const double* angles, int rc, double range_max, bool* done)
{
  *done = false;
  const PlanarModel& pm = e->pm;
  if (e->seam_chunks != 0 || !e->have_map || !e->have_lut || !pm.configured || rc <= 0 || !ranges || !angles || n < 4096)
    return BPF_OK;
  if (pm.model == BPF_MODEL_BEAM || (pm.model == BPF_MODEL_LIKELIHOOD_FIELD_PROB && pm.do_beamskip && set_converged))
    return BPF_OK;
  if (e->map.n_levels + 1 > kTableLdsMax)
    return BPF_OK;
  if (!host_buffer_pinned(e, samples, (size_t)n * sizeof(double4)))
    return BPF_OK;
  const bpf_engine::HostReg* r = host_reg_find(e, samples, (size_t)n * sizeof(double4));
  if (r == nullptr || r->dev_base == 0 || (reinterpret_cast<uintptr_t>(samples) & 15) != 0)
    return BPF_OK;  // (the records are read and written as 16-byte pairs)
  double4* rec = reinterpret_cast<double4*>(r->dev_base + (reinterpret_cast<uintptr_t>(samples) - r->base));
  int rcode = seam_resources(e);
  if (rcode != BPF_OK)
    return rcode;
  static const bool dbg = getenv("BPF_DEBUG_SEAM") != nullptr;
  auto now = []() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  const double t0 = dbg ? now() : 0.0;
  e->fused_partials = 0;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  ScanSlot* s = nullptr;
  FieldScan fs;
  rcode = stage_field_scan(e, ranges, angles, rc, range_max, &s, &fs);
  if (rcode != BPF_OK)
    return rcode;
  e->evals_last = (long long)n * fs.n_valid;
  const unsigned long long gen = ++e->seam_generation;
  const FieldHostOut out{ rec, nullptr, e->h_seam_totals.p, e->d_seam_partials.p, e->h_seam_flags.p, gen };
  rcode = launch_field(e, ParticlesDev{ nullptr, nullptr, nullptr, nullptr }, n, s, fs, nullptr, 0, false, nullptr, &out);
  if (rcode != BPF_OK)
    return rcode;
  rcode = release_slot(e, s);
  if (rcode != BPF_OK)
    return rcode;
  const double t1 = dbg ? now() : 0.0;
  if (!seam_wait_word(e->h_seam_flags.p, gen))
    HIPCHK(e, hipStreamSynchronize(e->stream));
  e->h_scalars.p->v[0] = *const_cast<const volatile double*>(e->h_seam_totals.p);
  if (dbg)
    fprintf(stderr, "seam: records in place, one launch: stage + issue %.1f us, waiting %.1f us\n", t1 - t0, now() - t1);
  e->last_seam_chunks = -1;
  e->last_seam_registered = true;
  *done = true;
  return BPF_OK;
}

// Seam A with the particles in HOST memory, pipelined.  What the plain sequence spends at 100 k x 1081 (measured,
// tools/ubench/pcie_probe.hip): 64 us for the 3.2 MB upload (the link's rate, pinned or not), 5 + 77 us for the
// launches, ~10 us for the total, 22-44 us for the download of the weights, ~30 us for this thread to write them into
// the caller's records, and a stream synchronisation.  Here the set goes up in a few chunks on a copy stream; chunk k
// is unpacked and scored (k_field_prep_aos + the HOST_OUT form of k_score_field on its range) while chunk k + 1 is
// still crossing PCIe; the scoring launch stores the weights into a pinned host array as it finishes them (dense
// 8-byte stores, they leave while the kernel works: no download) and a one-block launch behind it folds the chunk's
// total and publishes a word in pinned memory; this thread polls that word and writes chunk k's weights into the
// caller's records while chunk k + 1 is scored.  Left in series: the first chunk's upload, the scoring and the last
// chunk's write-back.  Few API calls on purpose -- each costs 2-5 us of this
// thread, and a cross-stream dependency ~10 us of latency, which is why the download stream and its events are gone.
// The per-particle arithmetic is that of the one-launch form (same kernels, same beams and table), so the weights are
// the same bits; the total is the sum of the chunks' totals, each the fixed-shape sum of its launch's block partials.
// Sets *done = false when the configuration is not one it takes (beam model, beam skipping, a small set, a table
// that does not fit LDS): the caller then runs the plain sequence.
int apply_model_pipelined(bpf_engine* e, double* samples, int n, int set_converged, const double* ranges,
                          const double* angles, int rc, double range_max, bool* done)
{
  *done = false;
  const PlanarModel& pm = e->pm;
  if (!e->have_map || !e->have_lut || !pm.configured || rc <= 0 || !ranges || !angles)
    return BPF_OK;  // the plain sequence reports what is missing
  if (pm.model == BPF_MODEL_BEAM || (pm.model == BPF_MODEL_LIKELIHOOD_FIELD_PROB && pm.do_beamskip && set_converged))
    return BPF_OK;
  if (e->map.n_levels + 1 > kTableLdsMax)
    return BPF_OK;
  // two chunks by default: a chunk costs ~35 us of pipeline granularity on this platform (a copy's fixed 9 us, a
  // cross-stream dependency's ~10 us, ~15 us of API calls by this thread), which more chunks do not win back
  int chunks = e->seam_chunks > 0 ? e->seam_chunks : (n >= 40000 ? 2 : 1);
  chunks = std::min(std::min(chunks, kSeamMaxChunks), n / 64);
  if (chunks < 2)
    return BPF_OK;
  int rcode = seam_resources(e);
  if (rcode != BPF_OK)
    return rcode;
  const bool pinned = host_buffer_pinned(e, samples, (size_t)n * sizeof(double4));
  static const bool dbg = getenv("BPF_DEBUG_SEAM") != nullptr;  // stage clocks of every call on stderr
  auto now = []() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  const double t0 = dbg ? now() : 0.0;
  e->fused_partials = 0;
  e->tile_sums_n = e->cdf_ready_n = e->cdf_coarse_n = -1;
  HIPCHK(e, e->d_aos.reserve((size_t)n));
  HIPCHK(e, e->scratch.reserve((size_t)n));
  // the weights come back through a FINE-grained pinned array: the scoring launches' stores are written through, so
  // they are in host memory when the launch behind them (k_seam_done) publishes the chunk's done word.  In a default
  // (coarse-grained) allocation they may still sit in some XCD's L2 at that point -- nothing but a system-scope
  // release or a stream synchronisation flushes them, and a kernel's end is neither when another kernel follows --
  // and this thread would read the previous call's weights (seen once in ~20 runs of the test suite).
  e->h_seam_w.coherent = true;
  HIPCHK(e, e->h_seam_w.reserve((size_t)n));
  ScanSlot* s = nullptr;
  FieldScan fs;
  rcode = stage_field_scan(e, ranges, angles, rc, range_max, &s, &fs);
  if (rcode != BPF_OK)
    return rcode;
  e->evals_last = (long long)n * fs.n_valid;
  double* hw = e->h_seam_w.p;
  const double4* src = reinterpret_cast<const double4*>(samples);
  const unsigned long long gen = ++e->seam_generation;
  // A pinned buffer's FIRST chunk is not copied at all: its prep launch reads the records from host memory itself
  // (zero-copy, ~25 us per MB), in stream order -- a copy on the other stream costs 9 us of set-up and ~18 us until
  // the dependent launch starts, with nothing to hide them behind at the start of the call.  The first chunk is
  // the smaller one then (40 %): the second chunk's copy (which runs beside it from the start) is there when it ends.
  const double4* dev_view = nullptr;
  if (pinned)
    if (const bpf_engine::HostReg* r = host_reg_find(e, samples, (size_t)n * sizeof(double4)))
      if (r->dev_base != 0)
        dev_view = reinterpret_cast<const double4*>(r->dev_base + (reinterpret_cast<uintptr_t>(samples) - r->base));
  const bool first_direct = dev_view != nullptr && chunks == 2 && e->seam_chunks == 0;
  auto lo_of = [&](int c) {
    const long long num = first_direct && c == 1 ? (long long)n * 2 / 5 : (long long)n * c / chunks;
    return (int)(num & ~63ll);
  };
  // every upload first: a pinned buffer costs this thread a microsecond per call, and the copy engine then has its
  // whole queue; a pageable one is staged by this thread inside the call, so its chunks are issued one ahead of the
  // launches instead (the staging of chunk k + 1 then runs beside the scoring of chunk k)
  auto upload = [&](int c) -> int {
    const int lo = lo_of(c), hi = c + 1 == chunks ? n : lo_of(c + 1);
    H2D_OR_RETURN(h2d_from_host(e, e->d_aos.p + lo, src + lo, (size_t)(hi - lo) * sizeof(double4), e->copy_up));
    HIPCHK(e, hipEventRecord(e->seam_ev[c], e->copy_up));
    return BPF_OK;
  };
  if (pinned)
    for (int c = first_direct ? 1 : 0; c < chunks; ++c)
      if ((rcode = upload(c)) != BPF_OK)
        return rcode;
  for (int c = 0; c < chunks; ++c)
  {
    const int lo = lo_of(c), hi = c + 1 == chunks ? n : lo_of(c + 1), cnt = hi - lo;
    const bool direct = first_direct && c == 0;
    if (!pinned && (rcode = upload(c)) != BPF_OK)
      return rcode;
    if (!direct)
      HIPCHK(e, hipStreamWaitEvent(e->stream, e->seam_ev[c], 0));
    ParticlesDev p = e->scratch.dev();
    p.x += lo; p.y += lo; p.th += lo; p.w += lo;
    FieldScan fc = fs;
    fc.copy_pending = fs.copy_pending && c == 0;  // the staging block rides with the first chunk's prep launch
    const FieldHostOut out{ nullptr, hw + lo, e->h_seam_totals.p + c, e->d_seam_partials.p + (size_t)c * kSeamMaxBlocks,
                            e->h_seam_flags.p + c, gen };
    rcode = launch_field(e, p, cnt, s, fc, nullptr, 0, false, direct ? dev_view + lo : e->d_aos.p + lo, &out);
    if (rcode != BPF_OK)
    {
      // the copy engine and the launches already issued read the caller's buffer: let them finish before returning
      (void)hipStreamSynchronize(e->copy_up);
      (void)hipStreamSynchronize(e->stream);
      return rcode;
    }
  }
  rcode = release_slot(e, s);
  if (rcode != BPF_OK)
    return rcode;
  const double t1 = dbg ? now() : 0.0;
  double t_wait = 0.0, t_scatter = 0.0;
  bool synced = false;
  double total = 0.0;
  for (int c = 0; c < chunks; ++c)
  {
    const double ta = dbg ? now() : 0.0;
    const unsigned long long* word = e->h_seam_flags.p + c;
    if (!synced && !seam_wait_word(word, gen))
    {
      HIPCHK(e, hipStreamSynchronize(e->stream));  // the slow way; everything is there afterwards
      synced = true;
    }
    const double tb = dbg ? now() : 0.0;
    {
      const int lo = lo_of(c), hi = c + 1 == chunks ? n : lo_of(c + 1);
      for (int i = lo; i < hi; ++i)
        samples[4 * (size_t)i + 3] = hw[i];
      total += *const_cast<const volatile double*>(e->h_seam_totals.p + c);  // chunk totals in chunk order
    }
    if (dbg)
    {
      t_wait += tb - ta;
      t_scatter += now() - tb;
    }
  }
  if (dbg)
    fprintf(stderr, "seam: %d chunks, %s: stage + issue %.1f us, waiting %.1f us, write-back %.1f us\n", chunks,
            pinned ? "pinned" : "pageable", t1 - t0, t_wait, t_scatter);
  e->h_scalars.p->v[0] = total;
  e->last_seam_chunks = chunks;
  e->last_seam_registered = pinned;
  *done = true;
  return BPF_OK;
}
}  // namespace

double bpf_planar_apply_model_to_sample_set(bpf_engine* e, double* samples, int sample_count, int set_converged,
                                            const double* ranges, const double* angles, int range_count,
                                            double range_max, int* status)
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
  if (e->pm.max_beams < 2)
    return 0.0;  // planar_scanner.cpp:144-145
  auto bail = [&](int code) { *status = code; return 0.0; };
  if (hipSetDevice(e->device) != hipSuccess)
    return bail(e->fail(BPF_ERR_HIP, "hipSetDevice"));
  int rc = ensure_scalars(e);
  if (rc != BPF_OK)
    return bail(rc);
  e->last_seam_chunks = 0;
  bool piped = false;
  rc = apply_model_records(e, samples, sample_count, set_converged, ranges, angles, range_count, range_max, &piped);
  if (rc != BPF_OK)
    return bail(rc);
  if (piped)
    return e->h_scalars.p->v[0];
  rc = apply_model_pipelined(e, samples, sample_count, set_converged, ranges, angles, range_count, range_max, &piped);
  if (!piped)
    e->last_seam_registered = host_reg_find(e, samples, (size_t)sample_count * sizeof(double4)) != nullptr;
  if (rc != BPF_OK)
    return bail(rc);
  if (piped)
    return e->h_scalars.p->v[0];
  // the plain sequence: the records go up as they are (one copy), the scoring path's prep launch unpacks them
  HIPCHK_OR(e, e->d_aos.reserve((size_t)sample_count), bail);
  HIPCHK_OR(e, e->scratch.reserve((size_t)sample_count), bail);
  {
    const int rcu = h2d_from_host(e, e->d_aos.p, samples, (size_t)sample_count * sizeof(double4), e->stream);
    if (rcu != BPF_OK)
      return bail(rcu);
  }
  // from here on the copy engine may still be reading the caller's (registered) buffer: no return before it is done
  auto bail_sync = [&](int code) {
    (void)hipStreamSynchronize(e->stream);
    return bail(code);
  };
  bool forced_zero = false;
  rc = score_planar(e, e->scratch.dev(), sample_count, set_converged, ranges, angles, range_count, range_max,
                    &forced_zero, false, false, e->d_aos.p);
  if (rc != BPF_OK)
    return bail_sync(rc);
  rc = sum_into_slot(e, e->scratch.w.p, sample_count, 0, 0, sample_count);
  if (rc != BPF_OK)
    return bail_sync(rc);
  rc = download_weights(e, e->scratch, sample_count, samples);
  if (rc != BPF_OK)
    return bail(rc);
  return e->h_scalars.p->v[0];
}
