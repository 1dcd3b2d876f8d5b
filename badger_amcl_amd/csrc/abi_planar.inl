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
  rc = upload_samples(e, samples, sample_count, e->scratch);
  if (rc != BPF_OK)
    return bail(rc);
  bool forced_zero = false;
  rc = score_planar(e, e->scratch.dev(), sample_count, set_converged, ranges, angles, range_count, range_max,
                    &forced_zero);
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
