// C-ABI: measurement.
// ---------------------------------------------------------------------- measurement
int bpf_device_memory_info(int device_ordinal, size_t* free_bytes, size_t* total_bytes)
{
  if (!free_bytes || !total_bytes)
    return BPF_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(device_ordinal) != hipSuccess || hipMemGetInfo(free_bytes, total_bytes) != hipSuccess)
    return BPF_ERR_HIP;
  return BPF_OK;
}

int bpf_profile_enable(bpf_engine* e, int on)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  if (on && e->ev_start.empty())
  {
    e->ev_start.resize(kEventPool);
    e->ev_stop.resize(kEventPool);
    e->ev_class.assign(kEventPool, 0);
    for (int i = 0; i < kEventPool; ++i)
    {
      HIPCHK(e, hipEventCreate(&e->ev_start[i]));
      HIPCHK(e, hipEventCreate(&e->ev_stop[i]));
    }
  }
  e->profiling = on != 0;
  e->profile_all = on == 2;
  e->timed_stride = on == 3 ? 1u : kTimedLaunchStride;
  return BPF_OK;
}

static int drain_events(bpf_engine* e)
{
  if (e->ev_used == 0)
    return BPF_OK;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (size_t i = 0; i < e->ev_used; ++i)
  {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_start[i], e->ev_stop[i]) == hipSuccess)
    {
      e->prof.ms[e->ev_class[i]] += ms;
      e->prof.launches[e->ev_class[i]] += 1;
    }
  }
  e->ev_used = 0;
  return BPF_OK;
}

int bpf_profile_reset(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  int rc = drain_events(e);
  std::memset(&e->prof, 0, sizeof(e->prof));
  e->timed_launches = 0;  // the first scoring launch after a reset is a timed one
  return rc;
}

int bpf_profile_get(bpf_engine* e, bpf_profile* out)
{
  if (!e || !out)
    return BPF_ERR_INVALID_ARGUMENT;
  int rc = drain_events(e);
  *out = e->prof;
  return rc;
}

int bpf_get_window_plan(bpf_engine* e, int* used_window, int* chunks_covered, int* chunks_total)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  int uw = 0, cov = 0, tot = 0;
  if (e->last_used_window_path && e->d_plan.p)
  {
    HIPCHK(e, hipSetDevice(e->device));
    WindowPlan plan;
    HIPCHK(e, hipMemcpyAsync(&plan, e->d_plan.p, sizeof(int) * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    uw = plan.use_window;
    cov = plan.covered;
    tot = plan.n_chunks;
  }
  if (used_window)
    *used_window = uw;
  if (chunks_covered)
    *chunks_covered = cov;
  if (chunks_total)
    *chunks_total = tot;
  return BPF_OK;
}

const char* bpf_score_kernel_name(const bpf_engine* e)
{
  if (e && !e->pm.configured && e->cloud_configured)
    return "k_cloud_score";
  if (e && e->pm.model == BPF_MODEL_BEAM)
    return "k_score_beam";
  return "k_score_field";
}

#ifdef BPF_PHASE_TIMING
int bpf_debug_cloud_span(unsigned long long* out, int n_waves)
{
  if (n_waves > 8192)
    return BPF_ERR_INVALID_ARGUMENT;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cloud_span), (size_t)n_waves * 2 * sizeof(unsigned long long)) != hipSuccess)
    return BPF_ERR_HIP;
  return BPF_OK;
}
// diagnostic builds only (not in badger_pf.h): rows of the last k_score_field launch, see tools/phase_timing.py
int bpf_debug_phase_cycles(unsigned long long* out, int n_waves)
{
  if (n_waves > kPhaseWaves)
    return BPF_ERR_INVALID_ARGUMENT;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), (size_t)n_waves * 8 * sizeof(unsigned long long)) !=
      hipSuccess)
    return BPF_ERR_HIP;
  return BPF_OK;
}
#endif
