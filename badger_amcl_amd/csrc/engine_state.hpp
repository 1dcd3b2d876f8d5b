// Engine state: device / pinned buffers, per-scan staging descriptors, the bpf_engine struct and the
// HIPCHK error macro.  Part of the one translation unit engine.hip.
#pragma once
namespace
{

constexpr int kRing = 4;            // in-flight scan uploads
constexpr int kMaxBeams = 4096;     // beams staged in LDS per launch
// calc_range_skip forms j * 2 * dmin with a 24-bit multiply whose 32-bit result must not wrap: j <= c + 2,
// dmin <= c + 1 for a ray of c = range_max / resolution cells -> 2 (c + 2)(c + 1) < 2^31 <=> c <= 32 765
constexpr double kMaxRayCells = 32760.0;
constexpr int kTableLdsMax = 2048;  // table entries that still go to LDS
constexpr int kEventPool = 8192;
constexpr int kSeamMaxChunks = 8;
constexpr int kSeamMaxBlocks = 4096;

// Device / pinned buffers free themselves with the engine (bpf_destroy selects the device first).
template <typename T>
struct DevBuf
{
  T* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  hipError_t reserve(size_t n)
  {
    if (n <= cap)
      return hipSuccess;
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t r = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
    if (r == hipSuccess)
      cap = n;
    return r;
  }
  void release()
  {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

template <typename T>
struct PinnedBuf
{
  T* p = nullptr;
  size_t cap = 0;
  // fine-grained (hipHostMallocCoherent): a kernel's plain stores are written through to host memory.  The default
  // allocation is coarse-grained on this platform: plain stores may stay in the GPU's L2s until a system-scope
  // release -- the END of a kernel is not one when another kernel follows in the queue -- so data that the host reads
  // behind a flag word (and not behind a stream synchronisation) must not live in a default allocation.
  bool coherent = false;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { release(); }
  hipError_t reserve(size_t n)
  {
    if (n <= cap)
      return hipSuccess;
    if (p)
      (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    hipError_t r = hipHostMalloc(reinterpret_cast<void**>(&p), n * sizeof(T),
                                 coherent ? hipHostMallocCoherent : hipHostMallocDefault);
    if (r == hipSuccess)
      cap = n;
    return r;
  }
  void release()
  {
    if (p)
      (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct SampleSet
{
  DevBuf<double> x, y, th, w;
  ParticlesDev dev() { return ParticlesDev{ x.p, y.p, th.p, w.p }; }
  hipError_t reserve(size_t n)
  {
    hipError_t r;
    if ((r = x.reserve(n)) != hipSuccess) return r;
    if ((r = y.reserve(n)) != hipSuccess) return r;
    if ((r = th.reserve(n)) != hipSuccess) return r;
    return w.reserve(n);
  }
  void release()
  {
    x.release(); y.release(); th.release(); w.release();
  }
};

struct PlanarModel
{
  bool configured = false;
  int model = BPF_MODEL_LIKELIHOOD_FIELD;
  int max_beams = 0;
  double z_hit = 0, z_short = 0, z_max = 0, z_rand = 0, sigma_hit = 0, lambda_short = 0;
  GompertzDev g{ 0, 0, 0, 0, 0, 0 };
  int do_beamskip = 0;
  double beam_skip_distance = 0, beam_skip_threshold = 0, beam_skip_error_threshold = 0;
  double off_map_factor = 1.0, non_free_factor = 1.0, non_free_radius = 0.0;  // planar_scanner.cpp:42-44
  double pose[3] = { 0, 0, 0 };
};

struct ScanSlot
{
  PinnedBuf<unsigned char> host;
  DevBuf<unsigned char> dev;
  hipEvent_t done = nullptr;
  bool pending = false;
};

// host-side description of one staged scan (see stage_field_scan)
struct FieldScan
{
  int n_valid = 0;             // beams that pass the range_max / NaN tests
  int n_staged = 0;            // of those, the ones uploaded (all, or the kept ones of beam skipping)
  bool copy_pending = false;   // pinned staging not yet copied to the device slot
  int n_always_off = 0;        // valid beams too long / non-finite to stage: off the map for every pose
  double off_map_term = 0.0;   // table[K]
  int n_slots = 0;             // beam_ind range of the prob model
  std::vector<int> slot_of;    // staged beam -> beam_ind
  size_t beams_off = 0, table_off = 0, bytes = 0;
  int table_len = 0;
};

}  // namespace

struct bpf_engine
{
  int device = 0;
  int n_cu = 256;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string last_error;

  // ---- 2-D map
  bool have_map = false, have_lut = false;
  MapDev map{};
  int map_version = 0;
  std::vector<int8_t> h_cells8;
  std::vector<float> h_levels;
  DevBuf<uint16_t> d_lut_tiles;
  DevBuf<uint32_t> d_cheb;
  DevBuf<int8_t> d_cells8;
  DevBuf<float> d_levels;
  DevBuf<float> d_lut_f32;
  DevBuf<int> d_edt_tmp;

  // ---- planar scanner
  PlanarModel pm;
  ScanSlot ring[kRing];
  int ring_next = 0;
  // host-side caches of scan staging: cos/sin of the bearings (a sensor's bearings are the same
  // arithmetic sequence scan after scan, node_2d.cpp:559) and the per-level term table (depends
  // only on the model parameters, range_max and the map)
  std::vector<double> trig_angles, trig_cos, trig_sin;
  std::vector<double> term_table;
  struct TermKey
  {
    int model = -1, map_version = -1;
    double z_hit = 0, z_rand = 0, sigma = 0, range_max = 0;
    bool operator==(const TermKey& o) const
    {
      return model == o.model && map_version == o.map_version && z_hit == o.z_hit && z_rand == o.z_rand &&
             sigma == o.sigma && range_max == o.range_max;
    }
  } term_key;
  DevBuf<int> d_obs_count;
  FieldScan skip_fs;          // staging of the counting pass of beam skipping, kept for its second half
  bool skip_pending = false;
  // LDS-window scoring path
  DevBuf<double4> d_prep;
  DevBuf<double> d_prep_stats, d_chunk_partials;
  DevBuf<WindowPlan> d_plan;
  bool window_lds_attr_set = false;
  bool beam_lds_attr_set = false;
  bool graded_shares = true;    // BPF_OPT_GRADED_SHARES
  DevBuf<int> d_beam_counter;   // work counter of k_score_beam
  bool window_enabled = false;  // measured: no gain on wide clouds; opt-in via BPF_OPT_WINDOW_PATH
  bool last_used_window_path = false;
  DevBuf<unsigned long long> d_cells_walked;

  // ---- 3-D map + point-cloud scanner
  bool have_map3d = false;
  Map3dDev map3{};
  double map3_max_dist = 0.0;
  DevBuf<uint32_t> d_pose_indices;
  DevBuf<uint8_t> d_ratios, d_dense3d;
  bool lut_host = false;      // BPF_OPT_LUT_HOST: the 3-D LUT builder on the host
  int lut3d_generations = 0;  // FIFO generations the device builder ran (0: host)
  bool cloud_dense = true;   // BPF_OPT_CLOUD_DENSE: use the dense tiled volume when there is one
  size_t n_pose_indices = 0, n_ratios = 0;
  bool cloud_configured = false;
  int cloud_max_beams = 0;
  double cloud_z_hit = 0, cloud_z_rand = 0, cloud_sigma = 0;
  CloudModelDev cm{};
  DevBuf<float> d_affine, d_points;
  DevBuf<double> d_cloud_partials, d_cloud_table;
  PinnedBuf<float> h_points;
  PinnedBuf<double> h_cloud_table;

  // ---- particle filter
  bool have_pf = false;
  int min_samples = 0, max_samples = 0;
  double alpha_slow = 0, alpha_fast = 0, conv_threshold = 0;
  double pop_err = 0.01, pop_z = 3, dist_threshold = 0.5;  // particle_filter.cpp:58-60
  int resample_model = BPF_RESAMPLE_MULTINOMIAL;
  uint64_t rng = 0;  // glibc's unseeded drand48 state
  LcgJump jump{};
  SampleSet sets[2];
  int cur = 0;
  int sample_count = 0;
  int leaf_count = 0, bin_count = 0;
  int converged = 0;
  float percent_converged = 0;
  bool converged_pending = false;
  int conv_n = 0;
  double w_diff_last = 0;
  int last_status = BPF_OK;
  int resample_windows = 0;
  int window_hint = 4096;
  long long evals_last = 0;
  bool cdf_serial = false;
  bool count_cells = false;
  KdHistogram hist;
  SeenKeys seen;
  DevBuf<int> d_cdf_guide;     // head start for the CDF bisection (k_scan_final / cdf_find_guided)
  bool cdf_guide_valid = false;
  DevBuf<double> d_cdf, d_partials, d_targets, d_block_partials, d_tile_sums;
  int fused_partials = 0;     // > 0: the last scoring launch left that many per-block weight partials
  int tile_sums_n = -1;       // >= 0: d_tile_sums holds the 2048-tile sums of the current weights for that n
  int cdf_ready_n = -1;       // >= 0: d_cdf (and the guide) already hold the CDF of the current weights (k_normalize_cdf)
  DevBuf<unsigned long long> d_tile_slots;  // k_normalize_cdf's look-back slots, [2][256]
  unsigned tile_generation = 0;
  bool fused_resample = true; // BPF_OPT_FUSED_RESAMPLE
  bool lut_exact_edt = false; // BPF_OPT_LUT_EXACT_EDT: implicit LUT builds take the device EDT instead of the reference's brushfire
  bool fused_lds_attr_set = false;
  bool shard_stop_attr_set = false;
  bool shard_resample_attr_set = false;
  bool kld_persistent = false;         // BPF_OPT_KLD_PERSISTENT: the device tree in one launch when its grid is resident
  int kld_generation = 0;
  int kld_persist_blocks_per_cu = -1;  // occupancy of k_kld_tree_persistent (-1: not asked yet)
  DevBuf<unsigned> d_kld_bar;
  bool shard_cdf_valid = false;        // k_normalize_gathered_cdf left the local CDF of the current weights behind
  void* shard_cdf_flags = nullptr;     // the caller's miss flag that launch cleared (null: none)
  void* shard_flags_last = nullptr;    // flags_dev of the last bpf_shard_build_cdf
  int fused_used = 0;         // the last resample ran as the one-block kernel
  int fused_generation = 0;
  PinnedBuf<int> h_fused;
  DevBuf<FusedJump> d_fused_jump;
  DevBuf<unsigned long long> d_fused_keys;
  DevBuf<unsigned> d_fused_counter;
  DevBuf<double> d_cdf_coarse;
  int cdf_coarse_n = -1;      // d_cdf_coarse holds the subsample of the CDF in d_cdf for that n
  DevBuf<FilterScalars> d_scalars;
  DevBuf<int> d_keys, d_src_index, d_flags;  // d_flags[0] miss, [1] converged count
  DevBuf<double4> d_aos;
  PinnedBuf<int> h_keys;
  PinnedBuf<unsigned> h_done;
  unsigned done_generation = 0;
  bool zero_copy_keys = true;
  PinnedBuf<double> h_targets;
  hipEvent_t targets_read = nullptr;  // recorded after the sharded systematic window kernel
  PinnedBuf<int> h_flags;
  PinnedBuf<FilterScalars> h_scalars;
  PinnedBuf<double4> h_aos;
  SampleSet scratch;  // Seam A host-buffer path
  SampleSet snap;
  int snap_count = 0, snap_leaf = 0, snap_bins = 0;

  // ---- w_diff > 0: random free-space poses (Node::randomFreeSpacePose) and the draw chain
  int random_pose_mode = BPF_RANDOM_POSE_NONE;
  std::vector<float> h_lut_f32;     // the LUT as floats (free-space test: distance > non_free_space_radius)
  DevBuf<int2> d_free_ij;
  int n_free = 0;
  int free_map_version = -1;
  double free_radius = -1.0;
  double shard_w_diff = 0.0;        // of the sharded resample in progress (bpf_shard_begin_resample)
  bool shard_chain = false;         // its draw chain is in d_chain
  int shard_n_random = 0;           // systematic: random poses at the head of the new set
  uint64_t shard_rng0 = 0;
  DevBuf<uint64_t> d_chain_bits;
  DevBuf<int> d_chain_cnt, d_chain_exit, d_chain_entry, d_chain_base, d_chain;
  PinnedBuf<int> h_chain_word;

  // ---- mailbox exchange of the sharded path (kernels_mailbox.hpp, abi_mailbox.inl)
  struct Mailbox
  {
    bool active = false;            // created AND connected
    int rank = 0, world = 0;
    long long max_window = 0;
    size_t bytes = 0;
    char* own = nullptr;            // this engine's mailbox (uncached device memory, exported by IPC handle)
    char* peer[kMailboxMaxWorld] = {};
    bool opened[kMailboxMaxWorld] = {};
    unsigned long long tot_gen = 0; // generation of the last totals post
    unsigned long long win_gen = 0; // generation of the window handed out last
    bool win_wait = false;          // that window's columns are in flight: its first consumer kernel has to wait
    unsigned long long hello = 0;
    int fold_deferred = 0;          // > 0: that many scoring partials wait to be folded and posted by the normalise launch
  } mb;
  int shard_rank = 0, shard_world = 1;  // of the shard exchange in use (mailbox or collective)
  // ---- RCCL collectives (libbadger_pf_rccl.so, loaded by bpf_shard_bootstrap when the mailbox cannot be used)
  struct Collective
  {
    bool active = false;
    void* lib = nullptr;
    void* comm = nullptr;
    struct Fn
    {
      const char* (*last_error)() = nullptr;
      int (*unique_id_bytes)() = nullptr;
      int (*unique_id)(void*) = nullptr;
      int (*init)(void**, int, int, const void*) = nullptr;
      int (*destroy)(void*) = nullptr;
      int (*allgather_f64)(void*, const double*, double*, size_t, void*) = nullptr;
      int (*allreduce_sum_i64)(void*, long long*, size_t, void*) = nullptr;
      int (*allreduce_sum_i32)(void*, int*, size_t, void*) = nullptr;
    } fn;
    DevBuf<double> totals;
    DevBuf<long long> window[2];
    int window_turn = 0;
  } coll;
  DevBuf<int> d_shard_flags;        // the CDF-miss flag word of the one-call sharded updates
  void* mb_totals = nullptr;        // bpf_shard_mailbox_update_sensor_planar: this update's totals (mailbox slots)
  bool mb_totals_valid = false;
  DevBuf<double> d_shard_out;       // [3][max_samples] poses of a resample that spans several windows
  PinnedBuf<unsigned> h_mb_error;
  DevBuf<unsigned> d_mb_error;
  int mb_timeout_ms = 5000;
  PinnedBuf<int> h_mb_result;
  DevBuf<unsigned> d_mb_counter;

  // ---- KLD stop rule on the device (long draw streams)
  int kld_device_min = 8192;  // draws left after the first window from which the device tree takes over
  bool kld_device_used = false;
  int kld_leaf = 0, kld_bins = 0;
  DevBuf<unsigned long long> d_kld_hkey;
  DevBuf<int> d_kld_htmin, d_kld_slot, d_kld_cur, d_kld_first, d_kld_child, d_kld_flags, d_kld_limit;
  DevBuf<int2> d_kld_delta, d_kld_tiles, d_kld_counts;
  // the tree in LDS-sized pieces (kernels_kld2.hpp)
  DevBuf<int> d_kld2_int;       // tkeys[n] bucket[n] bk[n] cnt[N] off[N + 1] fill[N] n_tkeys n_top status[2]
  DevBuf<Kld2Top> d_kld2_top;
  DevBuf<unsigned long long> d_kld2_slots;  // look-back slots of k_kld2_scan
  bool kld_local = true;        // BPF_OPT_KLD_LOCAL
  bool kld2_attr_set = false;
  // the histogram tree's hash tables were left in their start state behind the last build (pieces form): table size,
  // buffers and capacities they were cleared at (0: not clean)
  unsigned kld_clean_table = 0;
  const void* kld_clean_key = nullptr;
  const void* kld_clean_tmin = nullptr;
  size_t kld_clean_cap = 0;
  int kld_last_form = 0;        // diagnostics: 2 = LDS pieces, 1 = level loop, 3 = persistent
  PinnedBuf<int> h_kld;
  std::vector<int> kld_limit_host;
  double kld_limit_key[4] = { -1, -1, -1, -1 };  // pop_err, pop_z, min_samples, max_samples of the cached table

  // ---- motion model
  int odom_model = BPF_ODOM_MODEL_DIFF;
  double odom_alpha[5] = { 0, 0, 0, 0, 0 };
  bool odom_configured = false;
  DevBuf<int> d_motion_counts;
  DevBuf<long long> d_motion_offsets, d_motion_result;
  DevBuf<double> d_gauss, d_init_rot;
  PinnedBuf<long long> h_motion_result;

  // ---- cluster statistics on the device (kernels_stats.hpp)
  bool stats_host = false;          // BPF_OPT_STATS_HOST: the bit-exact host evaluation instead
  bool stats_on_device = false;     // the current statistics came from the device
  bool stats_clusters_fetched = false;
  int stats_cluster_count = 0, stats_best = -1;
  double stats_best_weight = 0.0, stats_best_pose[3] = { 0, 0, 0 };
  DevBuf<int> d_stats_parent, d_stats_label, d_stats_root, d_stats_tiles, d_stats_flags;
  DevBuf<long long> d_stats_hi;
  DevBuf<unsigned long long> d_stats_lo;
  DevBuf<bpf_cluster> d_stats_clusters;
  DevBuf<StatsResult> d_stats_result;
  PinnedBuf<StatsResult> h_stats_result;
  PinnedBuf<int> h_stats_flags;
  PinnedBuf<int> h_stats_block;     // k_stats_block: [0] generation, [1] status, [4 ..] StatsResult
  int stats_generation = 0;
  bool stats_lds_attr_set = false;

  // ---- cluster statistics (host, lazy)
  std::vector<bpf_cluster> clusters;
  double set_mean[3] = { 0, 0, 0 }, set_cov[5] = { 0, 0, 0, 0, 0 };
  long long stats_epoch = -1;   // value of set_epoch the statistics were computed for
  long long set_epoch = 0;      // bumped whenever the current set's poses / weights change
  bool hist_matches_set = false;

  // ---- host-buffer seam (abi_hostbuf.inl, abi_planar.inl): caller memory pinned with hipHostRegister, the copy
  // streams and events of the pipelined applyModelToSampleSet, the lazily built histogram tree of an adopted set
  struct HostReg
  {
    uintptr_t base;
    size_t bytes;
    bool automatic;  // made by BPF_OPT_HOST_AUTO_REGISTER, not by bpf_host_buffer_register
    uintptr_t dev_base;  // the range as a kernel addresses it (hipHostGetDevicePointer of base), 0: none
  };
  std::vector<HostReg> host_regs;
  bool host_auto_register = false;  // BPF_OPT_HOST_AUTO_REGISTER
  int seam_chunks = 0;              // BPF_OPT_SEAM_CHUNKS (0 = by size, 1 = the plain sequence)
  hipStream_t copy_up = nullptr;    // the uploads of the pipelined seam
  std::vector<hipEvent_t> seam_ev;  // [kSeamMaxChunks]: chunk uploaded
  DevBuf<double> d_seam_partials;   // [kSeamMaxChunks][kSeamMaxBlocks] block partials of the chunks' launches
  PinnedBuf<unsigned long long> h_seam_flags;  // [kSeamMaxChunks]: chunk c scored (the launch's generation)
  PinnedBuf<double> h_seam_totals;  // [kSeamMaxChunks]: weight total of chunk c
  unsigned long long seam_generation = 0;
  PinnedBuf<double> h_seam_w;       // [n] the chunked form's weights on their way back: FINE-grained (see PinnedBuf)
  int last_seam_chunks = 0;         // chunks the last applyModelToSampleSet used (0: the plain sequence)
  bool last_seam_registered = false;
  bool tree_pending = false;        // the current set's histogram tree (leaf / bin counts) has not been built yet

  std::vector<double> stage_bx, stage_by;  // stage_field_scan's scratch: beam end points of every decimated beam

  // ---- tile-sorted scoring of a spread cloud (kernels_window.hpp, HOST_MODE 3 of k_score_field)
  bool tile_sort = true;            // BPF_OPT_TILE_SORT
  // Pageable host memory on its way up (h2d_from_host, host_common.inl): two pinned halves that this thread fills
  // while the copy engine empties the other one
  PinnedBuf<unsigned char> h_bounce;
  hipEvent_t bounce_ev[2] = { nullptr, nullptr };
  bool bounce_busy[2] = { false, false };
  int bounce_next = 0;
  bool host_direct = false;         // BPF_OPT_HOST_DIRECT_PAGEABLE
  bool spread_init = false;         // the set was initialised with uniform random poses and not resampled since
  DevBuf<int> d_tile_int;           // hist[2][kTileBins] cursor[kTileBins] tile[n] perm[n]
  int tile_parity = 0;              // which half of hist the last tile-sorted update counted into
  DevBuf<double4> d_prep_sorted;
  int last_score_form = 0;          // diagnostics: 3 = the last scoring launch walked the particles in tile order

  // ---- profiling
  bool profiling = false;
  unsigned timed_launches = 0;  // scoring launches seen in profile mode 1 / 3 (every timed_stride-th is timed)
  unsigned timed_stride = 8;    // mode 1: kTimedLaunchStride, mode 3: 1
  bool profile_all = false;
  std::vector<hipEvent_t> ev_start, ev_stop;
  std::vector<int> ev_class;
  size_t ev_used = 0;
  bpf_profile prof{};

  int fail(int code, const std::string& msg)
  {
    last_error = msg;
    last_status = code;
    return code;
  }
  int fail_hip(hipError_t r, const char* what)
  {
    return fail(BPF_ERR_HIP, std::string(what) + ": " + hipGetErrorString(r));
  }
};

#define HIPCHK(e, call)                          \
  do                                             \
  {                                              \
    hipError_t _r = (call);                      \
    if (_r != hipSuccess)                        \
      return (e)->fail_hip(_r, #call);           \
  } while (0)
