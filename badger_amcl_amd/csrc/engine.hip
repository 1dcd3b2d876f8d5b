// libbadger_pf_hip.so -- host engine and C-ABI (include/badger_pf.h) for the MI355X
// sensor-update + resample path.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <queue>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/badger_pf.h"
#include "device_types.hpp"
#include "kdhist.hpp"
#include "kernels_cloud.hpp"
#include "kernels_motion.hpp"
#include "kernels_kld.hpp"
#include "kernels_recovery.hpp"
#include "kernels_pf.hpp"
#include "kernels_score.hpp"
#include "kernels_window.hpp"

using namespace bpf;

namespace
{

constexpr int kRing = 4;            // in-flight scan uploads
constexpr int kMaxBeams = 4096;     // beams staged in LDS per launch
constexpr int kTableLdsMax = 2048;  // table entries that still go to LDS
constexpr int kEventPool = 8192;

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
    hipError_t r = hipHostMalloc(reinterpret_cast<void**>(&p), n * sizeof(T), hipHostMallocDefault);
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
  DevBuf<uint8_t> d_cheb;
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
  bool window_enabled = false;  // measured: no gain on wide clouds; opt-in via BPF_OPT_WINDOW_PATH
  bool last_used_window_path = false;
  DevBuf<unsigned long long> d_cells_walked;

  // ---- 3-D map + point-cloud scanner
  bool have_map3d = false;
  Map3dDev map3{};
  double map3_max_dist = 0.0;
  DevBuf<uint32_t> d_pose_indices;
  DevBuf<uint8_t> d_ratios;
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
  DevBuf<double> d_cdf, d_partials, d_targets, d_block_partials, d_tile_sums;
  int fused_partials = 0;     // > 0: the last scoring launch left that many per-block weight partials
  int tile_sums_n = -1;       // >= 0: d_tile_sums holds the 2048-tile sums of the current weights for that n
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

  // ---- KLD stop rule on the device (long draw streams)
  int kld_device_min = 8192;  // draws left after the first window from which the device tree takes over
  bool kld_device_used = false;
  int kld_leaf = 0, kld_bins = 0;
  DevBuf<unsigned long long> d_kld_hkey;
  DevBuf<int> d_kld_htmin, d_kld_slot, d_kld_cur, d_kld_first, d_kld_child, d_kld_flags, d_kld_limit;
  DevBuf<int2> d_kld_delta, d_kld_tiles, d_kld_counts;
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

  // ---- cluster statistics (host, lazy)
  std::vector<bpf_cluster> clusters;
  double set_mean[3] = { 0, 0, 0 }, set_cov[5] = { 0, 0, 0, 0, 0 };
  long long stats_epoch = -1;   // value of set_epoch the statistics were computed for
  long long set_epoch = 0;      // bumped whenever the current set's poses / weights change
  bool hist_matches_set = false;

  // ---- profiling
  bool profiling = false;
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

namespace
{

// ------------------------------------------------------------------ profiling helpers
struct ProfScope
{
  bpf_engine* e;
  int idx = -1;
  ProfScope(bpf_engine* eng, int klass) : e(eng)
  {
    if (!e->profiling || e->ev_used >= e->ev_start.size() ||
        (klass != BPF_K_SCORE && klass != BPF_K_SCORE_WINDOW && !e->profile_all))
      return;
    idx = (int)e->ev_used++;
    e->ev_class[idx] = klass;
    (void)hipEventRecord(e->ev_start[idx], e->stream);
  }
  ~ProfScope()
  {
    if (idx >= 0)
      (void)hipEventRecord(e->ev_stop[idx], e->stream);
  }
};

void lcg_tables(LcgJump& J)
{
  const uint64_t mask = (1ull << 48) - 1;
  uint64_t a = 0x5DEECE66Dull, c = 0xBull;
  for (int j = 0; j < 48; ++j)
  {
    J.A[j] = a;
    J.C[j] = c;
    c = (c * a + c) & mask;  // apply the step twice: x -> a*(a*x + c) + c
    a = (a * a) & mask;
  }
}

uint64_t lcg_skip_host(uint64_t x0, uint64_t n, const LcgJump& J)
{
  const uint64_t mask = (1ull << 48) - 1;
  uint64_t a = 1, c = 0;
  for (int j = 0; n != 0 && j < 48; ++j, n >>= 1)
    if (n & 1)
    {
      a = (a * J.A[j]) & mask;
      c = (c * J.A[j] + J.C[j]) & mask;
    }
  return (a * x0 + c) & mask;
}

int blocks_for(int n, int per_block)
{
  return (n + per_block - 1) / per_block;
}

// ------------------------------------------------------------------ map encoding
int encode_lut(bpf_engine* e, const float* lut)
{
  const int sx = e->map.size_x, sy = e->map.size_y;
  const size_t ncell = (size_t)sx * sy;
  std::unordered_map<uint32_t, int> seen;
  seen.reserve(4096);
  std::vector<float> levels;
  uint32_t last_bits = 0;
  bool have_last = false;
  for (size_t i = 0; i < ncell; ++i)
  {
    uint32_t bits;
    std::memcpy(&bits, &lut[i], 4);
    if (have_last && bits == last_bits)
      continue;
    last_bits = bits;
    have_last = true;
    if (seen.emplace(bits, 0).second)
    {
      levels.push_back(lut[i]);
      if (levels.size() > 8190)
        return e->fail(BPF_ERR_LUT_LEVELS, "distance LUT holds more than 8190 distinct values");
    }
  }
  std::sort(levels.begin(), levels.end());
  for (size_t k = 0; k < levels.size(); ++k)
  {
    uint32_t bits;
    std::memcpy(&bits, &levels[k], 4);
    seen[bits] = (int)k;
  }
  // padded image: a border cell all round, everything outside the map holds the off-map level K;
  // entries are level*8 (byte offset of the level's term in the per-scan table)
  const int tx = e->map.ltx, ty = e->map.lty;
  const uint16_t off_map_level = (uint16_t)(levels.size() * 8);
  std::vector<uint16_t> tiles((size_t)tx * ty * 64, off_map_level);
  for (int j = 0; j < sy; ++j)
  {
    uint32_t prev_bits = 0;
    int prev_idx = -1;
    for (int i = 0; i < sx; ++i)
    {
      uint32_t bits;
      std::memcpy(&bits, &lut[i + (size_t)j * sx], 4);
      if (prev_idx < 0 || bits != prev_bits)
      {
        prev_idx = seen[bits];
        prev_bits = bits;
      }
      const int u = i + 1, v = j + 1;
      tiles[((size_t)(v >> 3) * tx + (u >> 3)) * 64 + ((u & 7) << 3) + (v & 7)] = (uint16_t)(prev_idx * 8);
    }
  }
  HIPCHK(e, e->d_lut_tiles.reserve(tiles.size()));
  HIPCHK(e, hipMemcpy(e->d_lut_tiles.p, tiles.data(), tiles.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  HIPCHK(e, e->d_levels.reserve(levels.size() + 1));
  HIPCHK(e, hipMemcpy(e->d_levels.p, levels.data(), levels.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(e, e->d_lut_f32.reserve(ncell));
  HIPCHK(e, hipMemcpy(e->d_lut_f32.p, lut, ncell * sizeof(float), hipMemcpyHostToDevice));
  e->h_levels = levels;
  e->h_lut_f32.assign(lut, lut + ncell);
  e->map.lut_tiles = e->d_lut_tiles.p;
  e->map.levels = e->d_levels.p;
  e->map.n_levels = (int)levels.size();
  e->have_lut = true;
  e->map_version++;
  return BPF_OK;
}

int build_lut_device(bpf_engine* e, double max_dist)
{
  if (!e->have_map)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no 2-D map set");
  if (max_dist == 0.0)
    return BPF_OK;  // occupancy_map.cpp:141-145: leaves the LUT untouched
  const int sx = e->map.size_x, sy = e->map.size_y;
  const size_t ncell = (size_t)sx * sy;
  const int radius = (int)std::floor(max_dist / e->map.resolution);
  HIPCHK(e, e->d_edt_tmp.reserve(ncell));
  HIPCHK(e, e->d_lut_f32.reserve(ncell));
  dim3 grid(blocks_for(sx, 256), sy), block(256);
  hipLaunchKernelGGL(k_edt_rows, grid, block, 0, e->stream, e->d_cells8.p, sx, sy, radius, e->d_edt_tmp.p);
  hipLaunchKernelGGL(k_edt_cols, grid, block, 0, e->stream, e->d_edt_tmp.p, sx, sy, radius, e->map.resolution,
                     max_dist, e->d_lut_f32.p);
  HIPCHK(e, hipGetLastError());
  std::vector<float> lut(ncell);
  HIPCHK(e, hipMemcpyAsync(lut.data(), e->d_lut_f32.p, ncell * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->map.max_dist = max_dist;
  return encode_lut(e, lut.data());
}

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
  std::vector<double2> beams;
  beams.reserve(rc / step + 1);
  fs->slot_of.clear();
  fs->n_valid = 0;
  fs->n_always_off = 0;
  int slot = 0;
  const double res = e->map.resolution;
  for (int i = 0; i < rc; i += step, ++slot)
  {
    const double r = ranges[i];
    if (r >= range_max)
      continue;
    if (r != r)
      continue;
    ++fs->n_valid;
    if (keep_slot && !(slot < (int)keep_slot->size() && (*keep_slot)[slot]))
      continue;
    double2 b;
    b.x = (r * e->trig_cos[i]) / res;
    b.y = (r * e->trig_sin[i]) / res;
    // a non-finite or absurdly long beam (> 2^28 cells) ends off the map for every pose in the
    // reference ((int) of a NaN or huge double is INT_MIN on x86): it is not staged, its constant
    // off-map term is added in the epilogue instead
    if (!(std::fabs(b.x) < 268435456.0 && std::fabs(b.y) < 268435456.0))
    {
      ++fs->n_always_off;
      continue;
    }
    beams.push_back(b);
    fs->slot_of.push_back(slot);
  }
  fs->n_slots = slot;
  fs->n_staged = (int)beams.size();
  if (fs->n_staged > kMaxBeams)
    return e->fail(BPF_ERR_CAPACITY, "more than 4096 beams per scan after decimation");
  fs->beams_off = 0;
  fs->table_off = ((size_t)fs->n_staged * sizeof(double2) + 255) & ~(size_t)255;
  fs->bytes = fs->table_off + (size_t)fs->table_len * sizeof(double);
  ScanSlot* s;
  int rcode = acquire_slot(e, fs->bytes, &s);
  if (rcode != BPF_OK)
    return rcode;
  std::memcpy(s->host.p + fs->beams_off, beams.data(), beams.size() * sizeof(double2));
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

int launch_field(bpf_engine* e, ParticlesDev p, int n, ScanSlot* s, const FieldScan& fs, int* obs_count,
                 int skip_level, bool want_partials = false)
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
  const size_t lds = std::max<size_t>(32, (size_t)fs.n_staged * sizeof(double2) + table_bytes);
  // per-particle scanner pose / trig once per update (shared by both scoring forms)
  const int prep_blocks = blocks_for(n, 256);
  HIPCHK(e, e->d_prep.reserve((size_t)n));
  HIPCHK(e, e->d_prep_stats.reserve((size_t)prep_blocks * kPrepStats));
  {
    const int n16 = (int)((fs.bytes + 15) / 16);
    const bool ride = fs.copy_pending && n16 <= prep_blocks * 256;
    if (fs.copy_pending && !ride)
      HIPCHK(e, hipMemcpyAsync(s->dev.p, s->host.p, fs.bytes, hipMemcpyHostToDevice, e->stream));
    ProfScope pa(e, BPF_K_SCORE_AUX);
    const uint4* src = ride ? reinterpret_cast<const uint4*>(s->host.p) : static_cast<const uint4*>(nullptr);
    if (e->window_enabled)
      hipLaunchKernelGGL(k_field_prep<true>, dim3(prep_blocks), dim3(256), 0, e->stream, p, n, e->map, A.sp_x, A.sp_y,
                         A.sp_th, e->d_prep.p, e->d_prep_stats.p, src, reinterpret_cast<uint4*>(s->dev.p), n16);
    else
      hipLaunchKernelGGL(k_field_prep<false>, dim3(prep_blocks), dim3(256), 0, e->stream, p, n, e->map, A.sp_x, A.sp_y,
                         A.sp_th, e->d_prep.p, e->d_prep_stats.p, src, reinterpret_cast<uint4*>(s->dev.p), n16);
  }
  A.prep = e->d_prep.p;
  // One resident round: blocks per CU = what registers, LDS and the SGPR rule admit (the occupancy
  // API can over-report by one block for SGPR-heavy kernels: MI355X_MICROARCH.md, residency).
  int api_blocks = 0;
  const void* kfn = count_only ? reinterpret_cast<const void*>(&k_score_field<true, false>)
                               : (table_lds ? reinterpret_cast<const void*>(&k_score_field<false, true>)
                                            : reinterpret_cast<const void*>(&k_score_field<false, false>));
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&api_blocks, kfn, 256, lds) != hipSuccess || api_blocks < 1)
    api_blocks = 1;
  const int per_cu = std::max(1, std::min(api_blocks, 6));
  const int resident_waves = e->n_cu * per_cu * 4;
  A.per_wave = std::max(1, blocks_for(n, resident_waves));
  const int grid = std::max(1, blocks_for(blocks_for(n, A.per_wave), 4));
  A.block_partials = nullptr;
  A.skip_if_set = nullptr;
  e->last_used_window_path = false;
  if (want_partials && !count_only)
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
  ProfScope ps(e, BPF_K_SCORE);
  if (count_only)
    hipLaunchKernelGGL((k_score_field<true, false>), dim3(grid), dim3(256), lds, e->stream, A);
  else if (table_lds)
    hipLaunchKernelGGL((k_score_field<false, true>), dim3(grid), dim3(256), lds, e->stream, A);
  else
    hipLaunchKernelGGL((k_score_field<false, false>), dim3(grid), dim3(256), lds, e->stream, A);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

int sum_into_slot(bpf_engine* e, const double* v, int n, int slot, int update_averages, int n_samples)
{
  const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
  HIPCHK(e, e->d_partials.reserve((size_t)nb));
  ProfScope ps(e, BPF_K_REDUCE);
  hipLaunchKernelGGL(k_sum_partials, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, v, n, e->d_partials.p);
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(BPF_RED_BLOCK), 0, e->stream, e->d_partials.p, nb, e->d_scalars.p,
                     slot, update_averages, n_samples, e->alpha_slow, e->alpha_fast);
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
int score_planar(bpf_engine* e, ParticlesDev p, int n, int set_converged, const double* ranges,
                 const double* angles, int rc, double range_max, bool* forced_zero, bool want_partials = false,
                 bool defer_beamskip_pass2 = false)
{
  *forced_zero = false;
  e->fused_partials = 0;
  e->tile_sums_n = -1;
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
    // Rays of similar length walk together: order the beams by observed range so that the 64 lanes
    // of one iteration finish their Bresenham walks at about the same step (the per-particle sum is
    // order-independent up to rounding).  NaN ranges sort last.
    std::stable_sort(beams.begin(), beams.end(), [](const BeamRec& a, const BeamRec& b) {
      const bool an = a.obs != a.obs, bn = b.obs != b.obs;
      if (an || bn)
        return !an && bn;
      return a.obs > b.obs;
    });
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
    int api_blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&api_blocks, reinterpret_cast<const void*>(&k_score_beam), 256,
                                                     bytes) != hipSuccess || api_blocks < 1)
      api_blocks = 1;
    const int per_cu = std::max(1, std::min(api_blocks, 6));
    // rays differ in length, so cut the set ~4x finer than one range per resident wave
    A.per_wave = std::max(1, blocks_for(n, e->n_cu * per_cu * 4 * 4));
    const int grid = std::max(1, blocks_for(blocks_for(n, A.per_wave), 4));
    A.block_partials = nullptr;
    if (want_partials)
    {
      HIPCHK(e, e->d_block_partials.reserve((size_t)grid));
      A.block_partials = e->d_block_partials.p;
      e->fused_partials = grid;
    }
    {
      ProfScope ps(e, BPF_K_SCORE);
      hipLaunchKernelGGL(k_score_beam, dim3(grid), dim3(256), bytes, e->stream, A);
    }
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
    rcode = launch_field(e, p, n, s, fs, nullptr, 0, want_partials);
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

int fetch_scalars(bpf_engine* e)
{
  HIPCHK(e, hipMemcpyAsync(e->h_scalars.p, e->d_scalars.p, sizeof(FilterScalars), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->h_flags.p, e->d_flags.p, 8 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (e->converged_pending)
  {
    // particle_filter.cpp:206-219, float arithmetic for the percentage
    const double pct = (float)e->h_flags.p[1] / (float)e->conv_n * 100;
    e->percent_converged = (float)pct;
    e->converged = pct >= e->conv_threshold;
    e->converged_pending = false;
  }
  return BPF_OK;
}

int build_cdf(bpf_engine* e, const double* w, int n)
{
  HIPCHK(e, e->d_cdf.reserve((size_t)n + 1));
  ProfScope ps(e, BPF_K_CDF);
  if (e->cdf_serial)
  {
    hipLaunchKernelGGL(k_scan_serial, dim3(1), dim3(64), 0, e->stream, w, n, e->d_cdf.p);
    HIPCHK(e, hipMemsetAsync(e->d_flags.p, 0, sizeof(int), e->stream));
  }
  else
  {
    const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
    double* tiles;
    if (e->tile_sums_n == n && w == e->sets[e->cur].w.p)
    {
      tiles = e->d_tile_sums.p;  // left behind by k_normalize_fused; consumed (scanned in place) here
      e->tile_sums_n = -1;
    }
    else
    {
      HIPCHK(e, e->d_partials.reserve((size_t)nb));
      tiles = e->d_partials.p;
      hipLaunchKernelGGL(k_sum_partials, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, w, n, tiles);
    }
    if (nb <= 256)
    {
      // few tiles: every block of the final pass forms its own offset (one launch less)
      hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, w, n, tiles, 0, e->d_cdf.p,
                         e->d_flags.p);
    }
    else
    {
      hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(BPF_RED_BLOCK), 0, e->stream, tiles, nb, e->d_flags.p);
      hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, w, n, tiles, 1, e->d_cdf.p,
                         nullptr);
    }
  }
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

// updateConverged (particle_filter.cpp:170-220) on the current set, result fetched lazily
int launch_converged(bpf_engine* e)
{
  SampleSet& s = e->sets[e->cur];
  const int n = e->sample_count;
  int rcode = sum_into_slot(e, s.x.p, n, 3, 0, n);
  if (rcode != BPF_OK)
    return rcode;
  rcode = sum_into_slot(e, s.y.p, n, 4, 0, n);
  if (rcode != BPF_OK)
    return rcode;
  HIPCHK(e, hipMemsetAsync(e->d_flags.p + 1, 0, sizeof(int), e->stream));
  const int grid = std::max(1, std::min(blocks_for(n, 256), 1024));
  hipLaunchKernelGGL(k_count_converged, dim3(grid), dim3(256), 0, e->stream, s.x.p, s.y.p, n, e->d_scalars.p,
                     e->dist_threshold, e->d_flags.p + 1);
  HIPCHK(e, hipGetLastError());
  e->converged_pending = true;
  e->conv_n = n;
  return BPF_OK;
}

// Node2D::updateFreeSpaceIndices (node_2d.cpp:317-337) for the current map and non_free_space_radius, cached
int ensure_free_space(bpf_engine* e, FreeSpaceDev* out)
{
  if (e->random_pose_mode != BPF_RANDOM_POSE_FREE_SPACE_2D)
    return e->fail(BPF_ERR_UNSUPPORTED,
                   "w_diff > 0: random pose injection needs a pose generator (bpf_pf_set_random_pose_generator); "
                   "the node's random_pose_fn_ callback (particle_filter.cpp:385-388) cannot be called from here");
  if (!e->have_map || !e->have_lut)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "random free-space poses need the 2-D map and its distance LUT");
  const double radius = e->pm.non_free_radius;
  if (e->free_map_version != e->map_version || e->free_radius != radius)
  {
    const int sx = e->map.size_x, sy = e->map.size_y;
    std::vector<int2> ij;
    ij.reserve((size_t)sx * sy / 2);
    for (int i = 0; i < sx; ++i)
      for (int j = 0; j < sy; ++j)
      {
        const size_t idx = i + (size_t)j * sx;
        if (e->h_cells8[idx] == -1 && (double)e->h_lut_f32[idx] > radius)
          ij.push_back(make_int2(i, j));
      }
    e->n_free = (int)ij.size();
    HIPCHK(e, e->d_free_ij.reserve(std::max<size_t>(ij.size(), 1)));
    if (!ij.empty())
      HIPCHK(e, hipMemcpy(e->d_free_ij.p, ij.data(), ij.size() * sizeof(int2), hipMemcpyHostToDevice));
    e->free_map_version = e->map_version;
    e->free_radius = radius;
  }
  if (e->n_free <= 0)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "the map has no free cell to draw random poses from");
  out->ij = e->d_free_ij.p;
  out->n = e->n_free;
  out->size_x = e->map.size_x;
  out->size_y = e->map.size_y;
  out->origin_x = e->map.origin_x;
  out->origin_y = e->map.origin_y;
  out->resolution = e->map.resolution;
  return BPF_OK;
}

// Where every candidate draw 0 .. max_draws finds its stream elements when w_diff > 0 (kernels_recovery.hpp)
int build_draw_chain(bpf_engine* e, double w_diff, int max_draws)
{
  const long long positions = 3ll * ((long long)max_draws + 1) + 3;
  if (positions >= 0x7fffffffll)
    return e->fail(BPF_ERR_CAPACITY, "draw chain would pass 31-bit stream positions");
  const int n_seg = (int)((positions + kChainSeg - 1) / kChainSeg);
  HIPCHK(e, e->d_chain_bits.reserve((size_t)2 * n_seg));
  HIPCHK(e, e->d_chain_cnt.reserve((size_t)3 * n_seg));
  HIPCHK(e, e->d_chain_exit.reserve((size_t)3 * n_seg));
  HIPCHK(e, e->d_chain_entry.reserve((size_t)n_seg));
  HIPCHK(e, e->d_chain_base.reserve((size_t)n_seg));
  HIPCHK(e, e->d_chain.reserve((size_t)max_draws + 1));
  ChainArgs C{};
  C.rng_state = e->rng;
  C.w_diff = w_diff;
  C.n_seg = n_seg;
  C.max_draws = max_draws;
  C.seg_bits = e->d_chain_bits.p;
  C.seg_cnt = e->d_chain_cnt.p;
  C.seg_exit = e->d_chain_exit.p;
  C.seg_entry = e->d_chain_entry.p;
  C.seg_base = e->d_chain_base.p;
  C.chain = e->d_chain.p;
  C.jump = e->jump;
  ProfScope ps(e, BPF_K_DRAW);
  hipLaunchKernelGGL(k_chain_segments, dim3(blocks_for(n_seg, 256)), dim3(256), 0, e->stream, C);
  hipLaunchKernelGGL(k_chain_scan, dim3(1), dim3(1024), 0, e->stream, C);
  hipLaunchKernelGGL(k_chain_emit, dim3(blocks_for(n_seg, 256)), dim3(256), 0, e->stream, C);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

// The KLD stop rule for the whole candidate stream [0, maxs) on the device (kernels_kld.hpp), keys in
// e->d_keys (AoS): grows the histogram tree level by level and scans the leaf count.
// Returns BPF_OK with *stop_out = stop count (or -1: no stop), *leaf_out / *bins_out at the final count;
// *handled = false when a key does not fit the 64-bit packing or the tree is deeper than the level budget
// (the caller then replays on the host as before).
int kld_tree_on_device(bpf_engine* e, int maxs, bool* handled, int* stop_out, int* leaf_out, int* bins_out,
                       bool whole_stream = false)
{
  *handled = false;
  const int n = maxs;
  if (n >= (1 << 30))
    return BPF_OK;  // element indices 2v + side are folded as 32-bit tags
  // resampleLimit per leaf count, cached per parameter set (host libm, as the reference evaluates it)
  if (e->kld_limit_key[0] != e->pop_err || e->kld_limit_key[1] != e->pop_z || e->kld_limit_key[2] != e->min_samples ||
      e->kld_limit_key[3] != e->max_samples || (int)e->kld_limit_host.size() < n + 1)
  {
    e->kld_limit_host.resize((size_t)n + 1);
    for (int k = 0; k <= n; ++k)
      e->kld_limit_host[k] = resample_limit(k, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
    HIPCHK(e, e->d_kld_limit.reserve((size_t)n + 1));
    HIPCHK(e, hipMemcpyAsync(e->d_kld_limit.p, e->kld_limit_host.data(), ((size_t)n + 1) * sizeof(int),
                             hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->kld_limit_key[0] = e->pop_err;
    e->kld_limit_key[1] = e->pop_z;
    e->kld_limit_key[2] = e->min_samples;
    e->kld_limit_key[3] = e->max_samples;
  }
  unsigned table = 1024;
  while (table < 2u * (unsigned)n)
    table <<= 1;
  constexpr int kMaxLevels = 256;
  const int tiles = blocks_for(n, kKldTile);
  HIPCHK(e, e->d_kld_hkey.reserve(table));
  HIPCHK(e, e->d_kld_htmin.reserve(table));
  HIPCHK(e, e->d_kld_slot.reserve((size_t)n));
  HIPCHK(e, e->d_kld_cur.reserve((size_t)n));
  HIPCHK(e, e->d_kld_first.reserve((size_t)n));
  HIPCHK(e, e->d_kld_child.reserve((size_t)2 * n));
  HIPCHK(e, e->d_kld_delta.reserve((size_t)n));
  HIPCHK(e, e->d_kld_counts.reserve((size_t)n));
  HIPCHK(e, e->d_kld_tiles.reserve((size_t)tiles));
  HIPCHK(e, e->d_kld_flags.reserve(4 + kMaxLevels));
  HIPCHK(e, e->h_kld.reserve(4 + kMaxLevels));
  ProfScope ps(e, BPF_K_DRAW);
  HIPCHK(e, hipMemsetAsync(e->d_kld_hkey.p, 0xFF, (size_t)table * sizeof(unsigned long long), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_kld_htmin.p, 0x7F, (size_t)table * sizeof(int), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_kld_first.p, 0x7F, (size_t)n * sizeof(int), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_kld_child.p, 0x7F, (size_t)2 * n * sizeof(int), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_kld_flags.p, 0, (4 + kMaxLevels) * sizeof(int), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_kld_flags.p + 2, 0x7F, sizeof(int), e->stream));
  KldArgs K{};
  K.keys = e->d_keys.p;
  K.n = n;
  K.h_key = e->d_kld_hkey.p;
  K.h_tmin = e->d_kld_htmin.p;
  K.h_mask = table - 1;
  K.slot = e->d_kld_slot.p;
  K.cur = e->d_kld_cur.p;
  K.first = e->d_kld_first.p;
  K.child = e->d_kld_child.p;
  K.delta = e->d_kld_delta.p;
  K.flags = e->d_kld_flags.p;
  K.limit = e->d_kld_limit.p;
  const dim3 grid(blocks_for(n, 256)), block(256);
  hipLaunchKernelGGL(k_kld_hash, grid, block, 0, e->stream, K);
  hipLaunchKernelGGL(k_kld_init, grid, block, 0, e->stream, K);
  hipLaunchKernelGGL(k_kld_root_first, dim3(1), dim3(1024), 0, e->stream, K);
  int level = 0;
  bool done = false;
  while (!done && level < kMaxLevels)
  {
    const int batch = (level == 0) ? 32 : 16;
    for (int q = 0; q < batch && level < kMaxLevels; ++q, ++level)
    {
      hipLaunchKernelGGL(k_kld_children, dim3(blocks_for(n, kKldBlock)), dim3(kKldBlock), 0, e->stream, K);
      hipLaunchKernelGGL(k_kld_descend, dim3(blocks_for(n, kKldBlock)), dim3(kKldBlock), 0, e->stream, K,
                         e->d_kld_flags.p + 4 + level);
    }
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipMemcpyAsync(e->h_kld.p, e->d_kld_flags.p, (4 + kMaxLevels) * sizeof(int), hipMemcpyDeviceToHost,
                             e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->h_kld.p[0] != 0)
      return BPF_OK;  // a key outside the packing range: not handled
    done = e->h_kld.p[4 + level - 1] == 0;
  }
  if (getenv("BPF_DEBUG"))
    fprintf(stderr, "[kld device] n %d levels %d done %d\n", n, level, (int)done);
  if (!done)
    return BPF_OK;  // deeper than the level budget: not handled
  hipLaunchKernelGGL(k_kld_scan_tiles, dim3(tiles), dim3(256), 0, e->stream, (const int2*)e->d_kld_delta.p, n,
                     e->d_kld_tiles.p);
  hipLaunchKernelGGL(k_kld_scan_offsets, dim3(1), dim3(1024), 0, e->stream, e->d_kld_tiles.p, tiles);
  hipLaunchKernelGGL(k_kld_scan_final, dim3(tiles), dim3(256), 0, e->stream, K, (const int2*)e->d_kld_tiles.p,
                     e->d_kld_counts.p);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(e->h_kld.p, e->d_kld_flags.p, 4 * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  const int stop = whole_stream ? -1 : e->h_kld.p[2];  // whole_stream: the tree of all n keys, no stop rule
  const int M = (stop >= 1 && stop <= n) ? stop : n;
  int2 c;
  HIPCHK(e, hipMemcpyAsync(&c, e->d_kld_counts.p + (M - 1), sizeof(int2), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  *stop_out = (stop >= 1 && stop <= n) ? stop : -1;
  *leaf_out = c.x;
  *bins_out = c.y;
  *handled = true;
  return BPF_OK;
}

// the whole candidate stream [0, maxs) again with the keys on the device only, then the tree
int kld_on_device(bpf_engine* e, DrawArgs A, int maxs, bool* handled, int* stop_out, int* leaf_out, int* bins_out)
{
  A.m0 = 0;
  A.m1 = maxs;
  A.host_keys = nullptr;
  {
    ProfScope ps(e, BPF_K_DRAW);
    hipLaunchKernelGGL(k_draw_select, dim3(blocks_for(maxs, 256)), dim3(256), 0, e->stream, A);
  }
  return kld_tree_on_device(e, maxs, handled, stop_out, leaf_out, bins_out);
}

// Spin on the generation word a kernel publishes in pinned host memory (kernels of ~10 us); false if it
// takes implausibly long, and the caller falls back to a copy + stream synchronisation.
bool wait_generation(bpf_engine* e, unsigned generation)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins)
  {
    if (__atomic_load_n(e->h_done.p, __ATOMIC_ACQUIRE) == generation)
      return true;
    if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
      return false;
    __builtin_ia32_pause();
  }
}

int resample_multinomial(bpf_engine* e, double w_diff)
{
  SampleSet& a = e->sets[e->cur];
  SampleSet& b = e->sets[e->cur ^ 1];
  const int n = e->sample_count;
  const int maxs = e->max_samples;
  FreeSpaceDev free_space{};
  const int* chain = nullptr;
  if (w_diff > 0.0)
  {
    // :383-388: every draw first tests a uniform against w_diff; where each draw finds its stream elements is
    // resolved up front for all candidate draws
    int rcf = ensure_free_space(e, &free_space);
    if (rcf != BPF_OK)
      return rcf;
    rcf = build_draw_chain(e, w_diff, maxs);
    if (rcf != BPF_OK)
      return rcf;
    chain = e->d_chain.p;
  }
  HIPCHK(e, e->d_keys.reserve((size_t)maxs * 3));
  HIPCHK(e, e->d_src_index.reserve((size_t)maxs));
  HIPCHK(e, e->h_keys.reserve((size_t)maxs * 3));
  e->hist.clear();
  e->seen.reset((size_t)std::min(maxs, 1 << 20));
  int m0 = 0, stop = -1;
  int window = std::max(1024, std::min(e->window_hint, maxs));
  e->resample_windows = 0;
  int cached_leaf = -1, cached_limit = 0;
  e->kld_device_used = false;
  bool device_declined = false;
  // keep the host's first window short when the device tree can take over after it
  if (window > 4096 && maxs - 4096 >= e->kld_device_min)
    window = 4096;
  while (m0 < maxs && stop < 0)
  {
    const int m1 = std::min(maxs, m0 + window);
    DrawArgs A{};
    A.src = a.dev();
    A.n_src = n;
    A.cdf = e->d_cdf.p;
    A.dst = b.dev();
    A.m0 = m0;
    A.m1 = m1;
    A.rng_state = e->rng;
    A.jump = e->jump;
    A.keys = e->d_keys.p;
    A.src_index = e->d_src_index.p;
    A.miss_flag = e->d_flags.p;
    A.sharded = 0;
    A.chain = chain;
    A.free_space = free_space;
    // long stream ahead: either the first window found no stop, or the previous cycle ran to the end
    const bool long_stream = (m0 > 0 || e->window_hint >= maxs) && maxs - m0 >= e->kld_device_min;
    if (long_stream && !device_declined)
    {
      // no stop inside the first window and a long stream ahead (a spread cloud): the ordered replay moves
      // to the device for the whole stream
      bool handled = false;
      int dstop = -1, dleaf = 0, dbins = 0;
      int rc = kld_on_device(e, A, maxs, &handled, &dstop, &dleaf, &dbins);
      if (rc != BPF_OK)
        return rc;
      if (handled)
      {
        e->resample_windows++;
        stop = dstop;
        e->kld_device_used = true;
        e->kld_leaf = dleaf;
        e->kld_bins = dbins;
        break;
      }
      device_declined = true;  // key range or depth outside what the device tree takes: host replay as before
    }
    const int wn = m1 - m0;
    // Keys go straight into pinned host memory and the last block publishes a generation number
    // there: the host polls that word instead of paying for a copy plus a stream synchronisation.
    const bool zero_copy = e->zero_copy_keys && wn <= (1 << 20);
    if (zero_copy)
    {
      A.host_keys = e->h_keys.p;
      A.host_stride = wn;
      A.done_counter = reinterpret_cast<unsigned*>(e->d_flags.p + 4);
      A.host_done = reinterpret_cast<volatile unsigned*>(e->h_done.p);
      A.generation = ++e->done_generation;
    }
    {
      ProfScope ps(e, BPF_K_DRAW);
      hipLaunchKernelGGL(k_draw_select, dim3(blocks_for(wn, 256)), dim3(256), 0, e->stream, A);
    }
    HIPCHK(e, hipGetLastError());
    const bool have_keys = zero_copy && wait_generation(e, A.generation);
    int k_stride = wn;
    if (!have_keys)
    {
      HIPCHK(e, hipMemcpyAsync(e->h_keys.p, e->d_keys.p, (size_t)wn * 3 * sizeof(int), hipMemcpyDeviceToHost,
                               e->stream));
      HIPCHK(e, hipStreamSynchronize(e->stream));
      k_stride = 0;  // AoS triples from the device buffer
    }
    e->resample_windows++;
    const int* keys = e->h_keys.p;
    for (int m = m0; m < m1; ++m)
    {
      const int o = m - m0;
      const int k[3] = { k_stride ? keys[o] : keys[3 * o], k_stride ? keys[k_stride + o] : keys[3 * o + 1],
                         k_stride ? keys[2 * k_stride + o] : keys[3 * o + 2] };
      if (e->seen.first_time(k[0], k[1], k[2]))
      {
        e->hist.insert(k[0], k[1], k[2]);
        const int lc = e->hist.leaf_count();
        if (lc != cached_leaf)
        {
          cached_leaf = lc;
          cached_limit = resample_limit(lc, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
        }
      }
      if (m + 1 > cached_limit)  // particle_filter.cpp:416
      {
        stop = m + 1;
        break;
      }
    }
    m0 = m1;
    window *= 4;
  }
  const int M = (stop > 0) ? stop : maxs;
  // the window that found the stop also inserted nothing past it: hist is exactly set b's tree
  if (chain != nullptr)
  {
    // the stream was consumed up to the element before draw M's test
    HIPCHK(e, e->h_chain_word.reserve(1));
    HIPCHK(e, hipMemcpyAsync(e->h_chain_word.p, chain + M, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const unsigned next_test = (unsigned)e->h_chain_word.p[0] & 0x7fffffffu;
    e->rng = lcg_skip_host(e->rng, (uint64_t)next_test - 1ull, e->jump);
  }
  else
    e->rng = lcg_skip_host(e->rng, 2ull * (uint64_t)M, e->jump);
  e->window_hint = std::max(1024, ((M + M / 4) + 1023) / 1024 * 1024);
  e->sample_count = M;
  return BPF_OK;
}

int resample_systematic(bpf_engine* e, double w_diff)
{
  SampleSet& a = e->sets[e->cur];
  SampleSet& b = e->sets[e->cur ^ 1];
  const int n = e->sample_count;
  int count = resample_limit(e->leaf_count, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
  FreeSpaceDev free_space{};
  int num_random = 0;
  if (w_diff > 0.0)
  {
    // particle_filter.cpp:295-306: room for random poses on top of the systematic ones
    count *= (1.0 + w_diff);
    if (count > e->max_samples)
      count = e->max_samples;
    num_random = (int)(w_diff * count);
    if (num_random > 0)
    {
      int rcf = ensure_free_space(e, &free_space);
      if (rcf != BPF_OK)
        return rcf;
    }
  }
  const int num_systematic = count - num_random;
  const uint64_t rng_before = e->rng;
  e->rng = lcg_skip_host(e->rng, 1, e->jump);
  const double start = std::ldexp((double)e->rng, -48);
  const double delta = 1.0 / num_systematic;
  HIPCHK(e, e->d_keys.reserve((size_t)e->max_samples * 3));
  HIPCHK(e, e->d_src_index.reserve((size_t)e->max_samples));
  HIPCHK(e, e->h_keys.reserve((size_t)e->max_samples * 3));
  SystematicArgs A{};
  A.src = a.dev();
  A.n_src = n;
  A.cdf = e->d_cdf.p;
  A.dst = b.dev();
  A.count = count;
  A.n_random = num_random;
  A.rng_state = rng_before;
  A.jump = e->jump;
  A.free_space = free_space;
  A.keys = e->d_keys.p;
  A.src_index = e->d_src_index.p;
  A.miss_flag = e->d_flags.p;
  // The targets are a serial floating-point chain (particle_filter.cpp:337-341): target += delta, and
  // target -= 1 once it passes 1.  A CPU core runs that dependency chain several times faster than a
  // GPU lane, with the same IEEE arithmetic, so the host forms the targets and uploads them.
  HIPCHK(e, e->h_targets.reserve((size_t)e->max_samples));
  {
    double t = start;
    double* out = e->h_targets.p;
    for (int i = 0; i < num_systematic; ++i)
    {
      out[i] = t;
      t += delta;
      if (t > 1.0)
        t -= 1.0;
    }
  }
  // the kernel reads the targets straight from the pinned buffer (28 KB for 3.5 k samples) and, like the
  // multinomial draw kernel, leaves the keys in pinned memory behind a generation word
  A.targets = e->h_targets.p;
  const bool zero_copy = e->zero_copy_keys && count <= (1 << 20);
  if (zero_copy)
  {
    A.host_keys = e->h_keys.p;
    A.host_stride = count;
    A.done_counter = reinterpret_cast<unsigned*>(e->d_flags.p + 4);
    A.host_done = reinterpret_cast<volatile unsigned*>(e->h_done.p);
    A.generation = ++e->done_generation;
  }
  {
    ProfScope ps(e, BPF_K_DRAW);
    hipLaunchKernelGGL(k_systematic_select, dim3(blocks_for(count, 256)), dim3(256), 0, e->stream, A);
  }
  HIPCHK(e, hipGetLastError());
  int k_stride = count;
  if (!(zero_copy && wait_generation(e, A.generation)))
  {
    if (zero_copy)  // the kernel wrote the host rows; wait for it the slow way
      HIPCHK(e, hipStreamSynchronize(e->stream));
    else
    {
      HIPCHK(e, hipMemcpyAsync(e->h_keys.p, e->d_keys.p, (size_t)count * 3 * sizeof(int), hipMemcpyDeviceToHost,
                               e->stream));
      HIPCHK(e, hipStreamSynchronize(e->stream));
      k_stride = 0;
    }
  }
  e->hist.clear();
  e->seen.reset((size_t)std::min(count, 1 << 20));
  const int* keys = e->h_keys.p;
  for (int m = 0; m < count; ++m)
  {
    const int k0 = k_stride ? keys[m] : keys[3 * m], k1 = k_stride ? keys[k_stride + m] : keys[3 * m + 1],
              k2 = k_stride ? keys[2 * k_stride + m] : keys[3 * m + 2];
    if (e->seen.first_time(k0, k1, k2))
      e->hist.insert(k0, k1, k2);
  }
  // :316-324: the random poses took two uniforms each, right after the systematic start
  e->rng = lcg_skip_host(e->rng, 2ull * (uint64_t)num_random, e->jump);
  e->resample_windows = 1;
  e->sample_count = count;
  return BPF_OK;
}

int upload_samples(bpf_engine* e, const double* aos, int n, SampleSet& dst)
{
  HIPCHK(e, e->h_aos.reserve((size_t)n));
  HIPCHK(e, e->d_aos.reserve((size_t)n));
  HIPCHK(e, dst.reserve((size_t)n));
  std::memcpy(e->h_aos.p, aos, (size_t)n * sizeof(double4));
  HIPCHK(e, hipMemcpyAsync(e->d_aos.p, e->h_aos.p, (size_t)n * sizeof(double4), hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_aos_to_soa, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->d_aos.p, dst.dev(), n);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

}  // namespace

// ====================================================================== C-ABI
extern "C" {

int bpf_create(int device_ordinal, bpf_engine** out)
{
  if (!out)
    return BPF_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return BPF_ERR_HIP;  // no GPU: the product path fails loudly, there is no CPU fallback
  if (device_ordinal < 0 || device_ordinal >= count)
    return BPF_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(device_ordinal) != hipSuccess)
    return BPF_ERR_HIP;
  bpf_engine* e = new bpf_engine();
  e->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess)
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess)
  {
    delete e;
    return BPF_ERR_HIP;
  }
  e->stream = e->own_stream;
  lcg_tables(e->jump);
  *out = e;
  return BPF_OK;
}

void bpf_destroy(bpf_engine* e)
{
  if (!e)
    return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  for (auto& s : e->ring)
  {
    s.host.release();
    s.dev.release();
    if (s.done)
      (void)hipEventDestroy(s.done);
  }
  if (e->targets_read)
    (void)hipEventDestroy(e->targets_read);
  for (auto ev : e->ev_start)
    (void)hipEventDestroy(ev);
  for (auto ev : e->ev_stop)
    (void)hipEventDestroy(ev);
  // every DevBuf / PinnedBuf member frees itself when the engine is deleted (the device is selected above)
  if (e->own_stream)
    (void)hipStreamDestroy(e->own_stream);
  delete e;
}

const char* bpf_error_string(int code)
{
  switch (code)
  {
    case BPF_OK: return "ok";
    case BPF_ERR_INVALID_ARGUMENT: return "invalid argument";
    case BPF_ERR_NOT_CONFIGURED: return "map, model or filter not configured";
    case BPF_ERR_HIP: return "HIP runtime error (or no GPU present)";
    case BPF_ERR_UNSUPPORTED: return "unsupported on the device path";
    case BPF_ERR_CDF_MISS: return "CDF search found no interval (reference asserts)";
    case BPF_ERR_LUT_LEVELS: return "distance LUT has too many distinct values";
    case BPF_ERR_BEAM_STEP: return "beam model step is zero (reference never returns)";
    case BPF_ERR_CAPACITY: return "capacity exceeded";
    default: return "unknown";
  }
}

const char* bpf_last_error_message(const bpf_engine* e)
{
  return e ? e->last_error.c_str() : "null engine";
}

int bpf_set_stream(bpf_engine* e, void* hip_stream)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->stream = (hip_stream == BPF_OWN_STREAM) ? e->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
  return BPF_OK;
}

int bpf_synchronize(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return BPF_OK;
}

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
  // chessboard distance to the nearest blocked cell on the padded grid: two raster sweeps of the
  // 8-neighbour recurrence D = min(D, neighbour + 1), which is exact for the Chebyshev metric
  const int pw = size_x + 2, ph = size_y + 2;
  std::vector<uint16_t> dist((size_t)pw * ph, 0);
  for (int j = 0; j < size_y; ++j)
    for (int i = 0; i < size_x; ++i)
      if (cells[i + (size_t)j * size_x] == -1)
        dist[(size_t)(j + 1) * pw + (i + 1)] = 0xFFFF;
  for (int y = 1; y < ph - 1; ++y)
    for (int x = 1; x < pw - 1; ++x)
    {
      uint16_t& d = dist[(size_t)y * pw + x];
      if (d == 0)
        continue;
      const uint16_t* up = &dist[(size_t)(y - 1) * pw + x];
      uint16_t best = std::min(std::min(up[-1], up[0]), std::min(up[1], (&d)[-1]));
      best = (uint16_t)std::min<int>(best + 1, 0xFFFF);
      d = std::min(d, best);
    }
  for (int y = ph - 2; y >= 1; --y)
    for (int x = pw - 2; x >= 1; --x)
    {
      uint16_t& d = dist[(size_t)y * pw + x];
      if (d == 0)
        continue;
      const uint16_t* dn = &dist[(size_t)(y + 1) * pw + x];
      uint16_t best = std::min(std::min(dn[-1], dn[0]), std::min(dn[1], (&d)[1]));
      best = (uint16_t)std::min<int>(best + 1, 0xFFFF);
      d = std::min(d, best);
    }
  std::vector<uint8_t> cheb((size_t)pw * ph);
  for (size_t q = 0; q < cheb.size(); ++q)
    cheb[q] = (uint8_t)std::min<int>(dist[q], 255);
  HIPCHK(e, e->d_cells8.reserve(ncell));
  HIPCHK(e, hipMemcpy(e->d_cells8.p, e->h_cells8.data(), ncell, hipMemcpyHostToDevice));
  HIPCHK(e, e->d_cheb.reserve(cheb.size()));
  HIPCHK(e, hipMemcpy(e->d_cheb.p, cheb.data(), cheb.size(), hipMemcpyHostToDevice));
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
  HIPCHK(e, hipMemcpyAsync(out, e->d_lut_f32.p, ncell * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return BPF_OK;
}

// ---------------------------------------------------------------------- planar scanner
int bpf_planar_init(bpf_engine* e, int max_beams)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->pm.max_beams = max_beams;
  return BPF_OK;
}

static int need_lut_for(bpf_engine* e, double max_dist)
{
  // setModelLikelihoodField* call map_->updateDistancesLUT(max_dist) (planar_scanner.cpp:74,91,112).
  // A host-provided LUT built for the same max_dist is kept; otherwise build on the device.
  if (!e->have_map)
    return BPF_OK;  // model may be set before the map; the LUT is then required at scoring time
  if (e->have_lut && e->map.max_dist == max_dist)
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
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
  hipLaunchKernelGGL(k_soa_to_aos, dim3(blocks_for(sample_count, 256)), dim3(256), 0, e->stream, e->scratch.dev(),
                     e->d_aos.p, sample_count);
  if (hipMemcpyAsync(e->h_aos.p, e->d_aos.p, (size_t)sample_count * sizeof(double4), hipMemcpyDeviceToHost,
                     e->stream) != hipSuccess ||
      hipMemcpyAsync(e->h_scalars.p, e->d_scalars.p, sizeof(FilterScalars), hipMemcpyDeviceToHost, e->stream) !=
          hipSuccess ||
      hipStreamSynchronize(e->stream) != hipSuccess)
    return bail(e->fail(BPF_ERR_HIP, "copy back"));
  for (int i = 0; i < sample_count; ++i)
    samples[4 * i + 3] = e->h_aos.p[i].w;
  return e->h_scalars.p->v[0];
}

// ---------------------------------------------------------------------- particle filter
int bpf_pf_create(bpf_engine* e, int min_samples, int max_samples, double alpha_slow, double alpha_fast,
                  double global_localization_convergence_threshold)
{
  if (!e || max_samples <= 0 || min_samples < 0)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad filter sizes") : BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  int rc = ensure_scalars(e);
  if (rc != BPF_OK)
    return rc;
  e->min_samples = min_samples;
  e->max_samples = max_samples;
  e->alpha_slow = alpha_slow;
  e->alpha_fast = alpha_fast;
  e->conv_threshold = global_localization_convergence_threshold;
  e->pop_err = 0.01;
  e->pop_z = 3;
  e->dist_threshold = 0.5;
  e->resample_model = BPF_RESAMPLE_MULTINOMIAL;
  e->random_pose_mode = BPF_RANDOM_POSE_NONE;  // the constructor's random_pose_fn: none until one is set
  for (int k = 0; k < 2; ++k)
    HIPCHK(e, e->sets[k].reserve((size_t)max_samples));
  // ctor state (particle_filter.cpp:62-89): max_samples particles at the origin, weight 1/max
  e->cur = 0;
  e->sample_count = max_samples;
  HIPCHK(e, hipMemsetAsync(e->sets[0].x.p, 0, (size_t)max_samples * sizeof(double), e->stream));
  HIPCHK(e, hipMemsetAsync(e->sets[0].y.p, 0, (size_t)max_samples * sizeof(double), e->stream));
  HIPCHK(e, hipMemsetAsync(e->sets[0].th.p, 0, (size_t)max_samples * sizeof(double), e->stream));
  hipLaunchKernelGGL(k_fill, dim3(blocks_for(max_samples, 256)), dim3(256), 0, e->stream, e->sets[0].w.p,
                     1.0 / max_samples, max_samples);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemsetAsync(e->d_scalars.p, 0, sizeof(FilterScalars), e->stream));
  e->leaf_count = 0;
  e->bin_count = 0;
  e->converged = 0;
  e->converged_pending = false;
  e->window_hint = 4096;
  e->have_pf = true;
  return BPF_OK;
}

int bpf_pf_set_resample_model(bpf_engine* e, int resample_model)
{
  if (!e || (resample_model != BPF_RESAMPLE_MULTINOMIAL && resample_model != BPF_RESAMPLE_SYSTEMATIC))
    return BPF_ERR_INVALID_ARGUMENT;
  e->resample_model = resample_model;
  return BPF_OK;
}

int bpf_pf_set_population_size_parameters(bpf_engine* e, double pop_err, double pop_z)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->pop_err = pop_err;
  e->pop_z = pop_z;
  return BPF_OK;
}

int bpf_pf_set_decay_rates(bpf_engine* e, double alpha_slow, double alpha_fast)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->alpha_slow = alpha_slow;
  e->alpha_fast = alpha_fast;
  return BPF_OK;
}

int bpf_pf_srand48(bpf_engine* e, long seed)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->rng = ((((uint64_t)seed) & 0xFFFFFFFFull) << 16) | 0x330Eull;
  return BPF_OK;
}

int bpf_pf_set_rng_state(bpf_engine* e, uint64_t state48)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->rng = state48 & ((1ull << 48) - 1);
  return BPF_OK;
}

int bpf_pf_get_rng_state(const bpf_engine* e, uint64_t* state48)
{
  if (!e || !state48)
    return BPF_ERR_INVALID_ARGUMENT;
  *state48 = e->rng;
  return BPF_OK;
}

int bpf_pf_set_samples(bpf_engine* e, const double* samples, int sample_count, int leaf_count)
{
  if (!e || !samples)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (sample_count <= 0 || sample_count > e->max_samples)
    return e->fail(BPF_ERR_CAPACITY, "sample_count outside (0, max_samples]");
  HIPCHK(e, hipSetDevice(e->device));
  int rc = upload_samples(e, samples, sample_count, e->sets[e->cur]);
  if (rc != BPF_OK)
    return rc;
  e->sample_count = sample_count;
  e->tile_sums_n = -1;
  e->set_epoch++;
  e->hist_matches_set = leaf_count < 0;
  // initWith*: w_slow_ = w_fast_ = 0, converged = false (particle_filter.cpp:127,157,164-168)
  HIPCHK(e, hipMemsetAsync(e->d_scalars.p, 0, sizeof(FilterScalars), e->stream));
  e->converged = 0;
  e->converged_pending = false;
  if (leaf_count >= 0)
  {
    e->leaf_count = leaf_count;
    e->bin_count = -1;
  }
  else
  {
    e->hist.clear();
    for (int i = 0; i < sample_count; ++i)
    {
      int key[3];
      host_pose_key(samples[4 * i], samples[4 * i + 1], samples[4 * i + 2], key);
      e->hist.insert(key[0], key[1], key[2]);
    }
    e->leaf_count = e->hist.leaf_count();
    e->bin_count = e->hist.bin_count();
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));  // h_aos staging is reused by the next call
  return BPF_OK;
}

int bpf_pf_get_samples(bpf_engine* e, double* samples_out, int capacity, int* sample_count_out)
{
  if (!e || !samples_out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  const int n = e->sample_count;
  if (capacity < n)
    return e->fail(BPF_ERR_CAPACITY, "output too small");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->d_aos.reserve((size_t)n));
  HIPCHK(e, e->h_aos.reserve((size_t)n));
  hipLaunchKernelGGL(k_soa_to_aos, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->sets[e->cur].dev(),
                     e->d_aos.p, n);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(e->h_aos.p, e->d_aos.p, (size_t)n * sizeof(double4), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  std::memcpy(samples_out, e->h_aos.p, (size_t)n * sizeof(double4));
  if (sample_count_out)
    *sample_count_out = n;
  return BPF_OK;
}

int bpf_pf_snapshot(bpf_engine* e)
{
  if (!e || !e->have_pf)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  const size_t n = (size_t)e->sample_count;
  HIPCHK(e, e->snap.reserve(n));
  SampleSet& s = e->sets[e->cur];
  HIPCHK(e, hipMemcpyAsync(e->snap.x.p, s.x.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->snap.y.p, s.y.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->snap.th.p, s.th.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->snap.w.p, s.w.p, n * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  e->snap_count = e->sample_count;
  e->snap_leaf = e->leaf_count;
  e->snap_bins = e->bin_count;
  return BPF_OK;
}

int bpf_pf_restore(bpf_engine* e)
{
  if (!e || !e->have_pf || e->snap_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  const int n = e->snap_count;
  hipLaunchKernelGGL(k_copy4, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->sets[e->cur].dev(),
                     e->snap.dev(), n);
  HIPCHK(e, hipGetLastError());
  e->sample_count = e->snap_count;
  e->leaf_count = e->snap_leaf;
  e->bin_count = e->snap_bins;
  e->tile_sums_n = -1;
  e->set_epoch++;
  e->hist_matches_set = false;
  return BPF_OK;
}

int bpf_pf_fill_weights(bpf_engine* e, double weight)
{
  if (!e || !e->have_pf)
    return BPF_ERR_INVALID_ARGUMENT;
  e->tile_sums_n = -1;
  e->set_epoch++;
  HIPCHK(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(k_fill, dim3(blocks_for(e->sample_count, 256)), dim3(256), 0, e->stream,
                     e->sets[e->cur].w.p, weight, e->sample_count);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

int bpf_pf_update_sensor_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                double range_max)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (e->pm.max_beams < 2)
    return BPF_OK;  // PlanarScanner::updateSensor returns false and touches nothing (:128-129)
  HIPCHK(e, hipSetDevice(e->device));
  if (e->pm.model == BPF_MODEL_LIKELIHOOD_FIELD_PROB && e->pm.do_beamskip && e->converged_pending)
  {
    int rc = fetch_scalars(e);
    if (rc != BPF_OK)
      return rc;
  }
  SampleSet& s = e->sets[e->cur];
  const int n = e->sample_count;
  bool forced_zero = false;
  int rc = score_planar(e, s.dev(), n, e->converged, ranges, angles, range_count, range_max, &forced_zero, true);
  if (rc != BPF_OK)
    return rc;
  if (e->fused_partials > 0)
  {
    // the scoring kernel left per-block weight partials: one launch folds them, normalises, updates the
    // running averages and leaves the tile sums for the CDF
    const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
    HIPCHK(e, e->d_tile_sums.reserve((size_t)nb));
    ProfScope ps(e, BPF_K_NORMALIZE);
    hipLaunchKernelGGL(k_normalize_fused, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, s.w.p, n,
                       e->d_block_partials.p, e->fused_partials, e->d_scalars.p, e->alpha_slow, e->alpha_fast,
                       e->d_tile_sums.p);
    HIPCHK(e, hipGetLastError());
    e->tile_sums_n = n;
    e->fused_partials = 0;
  }
  else
  {
    rc = sum_into_slot(e, s.w.p, n, 0, 1, n);
    if (rc != BPF_OK)
      return rc;
    ProfScope ps(e, BPF_K_NORMALIZE);
    hipLaunchKernelGGL(k_normalize, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, s.w.p, n, e->d_scalars.p, 0,
                       0.0, n);
    HIPCHK(e, hipGetLastError());
  }
  e->last_status = BPF_OK;
  e->set_epoch++;
  return BPF_OK;
}

int bpf_pf_set_random_pose_generator(bpf_engine* e, int mode)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (mode != BPF_RANDOM_POSE_NONE && mode != BPF_RANDOM_POSE_FREE_SPACE_2D)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "unknown random pose generator");
  e->random_pose_mode = mode;
  return BPF_OK;
}

int bpf_pf_update_resample(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  // w_diff = max(0, 1 - w_fast/w_slow) (particle_filter.cpp:438-440).  With both decay rates
  // zero the two averages are always equal, so w_diff is 0 (or NaN before any update, which
  // the multinomial sampler treats as 0 too) and no read-back is needed.
  double w_diff = 0.0;
  if (e->alpha_slow != 0.0 || e->alpha_fast != 0.0)
  {
    int rc = fetch_scalars(e);
    if (rc != BPF_OK)
      return rc;
    const double ws = e->h_scalars.p->v[1], wf = e->h_scalars.p->v[2];
    w_diff = 1.0 - wf / ws;
    if (!(w_diff >= 0.0))
      w_diff = 0.0;
  }
  e->w_diff_last = w_diff;
  SampleSet& a = e->sets[e->cur];
  int rc = build_cdf(e, a.w.p, e->sample_count);
  if (rc != BPF_OK)
    return rc;
  e->kld_device_used = false;
  rc = (e->resample_model == BPF_RESAMPLE_SYSTEMATIC) ? resample_systematic(e, w_diff)
                                                      : resample_multinomial(e, w_diff);
  if (rc != BPF_OK)
    return rc;
  if (w_diff > 0.0)  // "Reset averages, to avoid spiraling off into complete randomness" (particle_filter.cpp:453-455)
    HIPCHK(e, hipMemsetAsync(&e->d_scalars.p->v[1], 0, 2 * sizeof(double), e->stream));
  const int M = e->sample_count;
  SampleSet& b = e->sets[e->cur ^ 1];
  e->tile_sums_n = -1;
  e->cur ^= 1;
  e->leaf_count = e->kld_device_used ? e->kld_leaf : e->hist.leaf_count();
  e->bin_count = e->kld_device_used ? e->kld_bins : e->hist.bin_count();
  if (M <= 8192)
  {
    // small resampled set: weights 1/M and updateConverged in one single-block launch
    ProfScope ps(e, BPF_K_FINALIZE);
    hipLaunchKernelGGL(k_resample_tail_small, dim3(1), dim3(1024), 0, e->stream, b.x.p, b.y.p, b.w.p, M,
                       e->dist_threshold, e->d_scalars.p, e->d_flags.p + 1);
    HIPCHK(e, hipGetLastError());
    e->converged_pending = true;
    e->conv_n = M;
  }
  else
  {
    {
      ProfScope ps(e, BPF_K_FINALIZE);
      // weight 1.0 each, total = M, then weight /= total (particle_filter.cpp:409,458-462)
      hipLaunchKernelGGL(k_fill, dim3(blocks_for(M, 256)), dim3(256), 0, e->stream, b.w.p, 1.0 / (double)M, M);
    }
    HIPCHK(e, hipGetLastError());
    rc = launch_converged(e);
    if (rc != BPF_OK)
      return rc;
  }
  // miss flag was copied? read it with the next fetch; report asynchronously via last_status
  e->last_status = BPF_OK;
  e->set_epoch++;
  e->hist_matches_set = !e->kld_device_used;  // the device tree leaves no host histogram behind
  return BPF_OK;
}

int bpf_pf_get_state(bpf_engine* e, bpf_pf_state* out)
{
  if (!e || !out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  int rc = fetch_scalars(e);
  if (rc != BPF_OK)
    return rc;
  if (e->h_flags.p[0] != 0 && e->last_status == BPF_OK)
    e->last_status = BPF_ERR_CDF_MISS;
  std::memset(out, 0, sizeof(*out));
  out->sample_count = e->sample_count;
  out->leaf_count = e->leaf_count;
  out->bin_count = e->bin_count;
  out->converged = e->converged;
  out->percent_converged = e->percent_converged;
  out->total = e->h_scalars.p->v[0];
  out->w_slow = e->h_scalars.p->v[1];
  out->w_fast = e->h_scalars.p->v[2];
  out->w_diff = e->w_diff_last;
  out->last_status = e->last_status;
  out->resample_windows = e->resample_windows;
  out->kld_on_device = e->kld_device_used ? 1 : 0;
  out->reserved = 0;
  out->evals = e->evals_last;
  return BPF_OK;
}

// ---------------------------------------------------------------------- motion update
namespace
{
double odom_angle_diff(double a, double b)
{
  // Odom::angleDiff (odom.cpp:308-311) = angles::shortest_angular_distance(b, a) = normalize_angle(a - b);
  // angles::normalize_angle in its Noetic form (third party)
  const double r = std::fmod((a - b) + M_PI, 2.0 * M_PI);
  return (r <= 0.0) ? r + M_PI : r - M_PI;
}

// the loop-invariant part of Odom::updateAction, with the host libm like the reference
MotionModelDev motion_constants(const bpf_engine* e, const double pose[3], const double delta[3],
                                const double absolute_motion[3])
{
  MotionModelDev M{};
  M.model = e->odom_model;
  const double a1 = e->odom_alpha[0], a2 = e->odom_alpha[1], a3 = e->odom_alpha[2], a4 = e->odom_alpha[3],
               a5 = e->odom_alpha[4];
  const double old_th = pose[2] - delta[2];  // odom.cpp:82-85
  const double delta_trans = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
  M.delta_trans = delta_trans;
  M.delta_rot = delta[2];
  M.half_rot = delta[2] / 2;
  M.bearing0 = odom_angle_diff(std::atan2(delta[1], delta[0]), old_th);
  if (M.model == BPF_ODOM_MODEL_OMNI || M.model == BPF_ODOM_MODEL_OMNI_CORRECTED)
  {
    const double delta_rot = delta[2];
    M.sd[0] = a3 * (delta_trans * delta_trans) + a1 * (delta_rot * delta_rot);  // :101-106 / :181-186
    M.sd[1] = a4 * (delta_rot * delta_rot) + a2 * (delta_trans * delta_trans);
    M.sd[2] = a1 * (delta_rot * delta_rot) + a5 * (delta_trans * delta_trans);
    if (M.model == BPF_ODOM_MODEL_OMNI_CORRECTED)
      for (double& v : M.sd)
        v = std::sqrt(v);
  }
  else if (M.model == BPF_ODOM_MODEL_DIFF || M.model == BPF_ODOM_MODEL_DIFF_CORRECTED)
  {
    M.rot1 = (delta_trans < 0.01) ? 0.0 : M.bearing0;  // :135-138 / :213-216
    M.rot2 = odom_angle_diff(delta[2], M.rot1);
    const double r1a = std::fabs(odom_angle_diff(M.rot1, 0.0)), r1b = std::fabs(odom_angle_diff(M.rot1, M_PI));
    const double r2a = std::fabs(odom_angle_diff(M.rot2, 0.0)), r2b = std::fabs(odom_angle_diff(M.rot2, M_PI));
    const double n1 = std::min(r1a, r1b), n2 = std::min(r2a, r2b);
    M.sd[0] = a1 * n1 * n1 + a2 * delta_trans * delta_trans;  // :156-162 / :233-243
    M.sd[1] = a3 * delta_trans * delta_trans + a4 * n1 * n1 + a4 * n2 * n2;
    M.sd[2] = a1 * n2 * n2 + a2 * delta_trans * delta_trans;
    if (M.model == BPF_ODOM_MODEL_DIFF_CORRECTED)
      for (double& v : M.sd)
        v = std::sqrt(v);
  }
  else
  {
    const double at2 = absolute_motion[0] * absolute_motion[0];  // :264-274
    const double as2 = absolute_motion[1] * absolute_motion[1];
    const double ar2 = absolute_motion[2] * absolute_motion[2];
    const double rot_sd = std::sqrt(a1 * ar2 + a2 * at2);
    const double trans_sd = std::sqrt(a3 * at2 + a4 * ar2);
    const double strafe_sd = std::sqrt(a4 * ar2 + a5 * as2);
    M.sd[0] = trans_sd;  // draw order :289-291
    M.sd[1] = strafe_sd;
    M.sd[2] = rot_sd;
  }
  return M;
}

// `need` Gaussians PDFGaussian::draw(sd[rank % 3]) from the filter's drand48 stream, ranks
// [first, first + count) materialised in d_gauss; *consumed_out = uniforms the whole update took.
// `after_gauss` is launched right behind the generation (optimistically: a rare second pass re-runs it).
int generate_gaussians(bpf_engine* e, long long need, long long first, long long count, const double sd[3],
                       long long* consumed_out, const std::function<void()>& after_gauss)
{
  // attempts are accepted with probability pi/4; 6 sigma of slack, doubled on the (never yet seen) shortfall
  long long attempts = (long long)std::ceil((double)need / 0.7853981633974483 + 6.0 * std::sqrt((double)need)) + 64;
  HIPCHK(e, e->d_motion_result.reserve(4));
  HIPCHK(e, e->h_motion_result.reserve(4));
  HIPCHK(e, e->d_gauss.reserve((size_t)std::max<long long>(count, 1)));
  long long zero_at = kNoZero;
  for (int round = 0; round < 8; ++round)
  {
    const int tiles = (int)((attempts + kMotionTile - 1) / kMotionTile);
    HIPCHK(e, e->d_motion_counts.reserve((size_t)tiles));
    HIPCHK(e, e->d_motion_offsets.reserve((size_t)tiles + 1));
    MotionRngArgs A{};
    A.rng_state = e->rng;
    A.zero_at = zero_at;
    A.n_attempts = attempts;
    A.need_total = need;
    A.gauss_first = first;
    A.gauss_count = count;
    A.tile_counts = e->d_motion_counts.p;
    A.tile_offsets = e->d_motion_offsets.p;
    A.gauss = e->d_gauss.p;
    A.result = e->d_motion_result.p;
    for (int k = 0; k < 3; ++k)
      A.sd[k] = sd[k];
    A.jump = e->jump;
    {
      ProfScope ps(e, BPF_K_MOTION);
      HIPCHK(e, hipMemsetAsync(e->d_motion_result.p, 0, 4 * sizeof(long long), e->stream));
      hipLaunchKernelGGL(k_motion_count, dim3(tiles), dim3(256), 0, e->stream, A);
      hipLaunchKernelGGL(k_motion_offsets, dim3(1), dim3(1024), 0, e->stream, A, tiles);
      hipLaunchKernelGGL(k_motion_gauss, dim3(tiles), dim3(256), 0, e->stream, A);
      after_gauss();
      HIPCHK(e, hipGetLastError());
    }
    HIPCHK(e, hipMemcpyAsync(e->h_motion_result.p, e->d_motion_result.p, 4 * sizeof(long long), hipMemcpyDeviceToHost,
                             e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const long long consumed = e->h_motion_result.p[0], zero_seen = e->h_motion_result.p[1],
                    accepted = e->h_motion_result.p[2];
    if (zero_at == kNoZero && zero_seen > 0)
    {
      zero_at = zero_seen;  // the stream's one exact 0.0 lies in the window: re-run with it skipped
      continue;
    }
    if (accepted < need || consumed <= 0)
    {
      attempts *= 2;
      continue;
    }
    *consumed_out = consumed;
    return BPF_OK;
  }
  return e->fail(BPF_ERR_HIP, "Gaussian stream did not fill (internal error)");
}

int update_action(bpf_engine* e, const double pose[3], const double delta[3], const double absolute_motion[3],
                  long long global_first, long long global_count)
{
  const int n = e->sample_count;
  if (global_first < 0 || global_first + n > global_count)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "shard range outside the global set");
  HIPCHK(e, hipSetDevice(e->device));
  const MotionModelDev M = motion_constants(e, pose, delta, absolute_motion);
  SampleSet& src = e->sets[e->cur];
  SampleSet& dst = e->sets[e->cur ^ 1];
  long long consumed = 0;
  // optimistic: poses go to the other set, which becomes current only once the pass is known good
  int rc = generate_gaussians(e, 3 * global_count, 3 * global_first, 3ll * n, M.sd, &consumed, [&]() {
    hipLaunchKernelGGL(k_motion_apply, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, src.dev(), dst.dev(), n, M,
                       (const double*)e->d_gauss.p);
  });
  if (rc != BPF_OK)
    return rc;
  e->rng = lcg_skip_host(e->rng, (uint64_t)consumed, e->jump);
  e->cur ^= 1;
  e->tile_sums_n = -1;
  e->fused_partials = 0;
  e->set_epoch++;
  e->hist_matches_set = false;
  return BPF_OK;
}

// what initWithGaussian / initWithPoseFn leave besides the poses (particle_filter.cpp:126-131,157-162): the
// histogram tree of the set (leaf / bin counts), w_slow = w_fast = 0, converged = false
int finish_init(bpf_engine* e, int n)
{
  SampleSet& s = e->sets[e->cur];
  e->sample_count = n;
  e->tile_sums_n = -1;
  e->fused_partials = 0;
  e->set_epoch++;
  e->hist_matches_set = false;
  HIPCHK(e, hipMemsetAsync(e->d_scalars.p, 0, sizeof(FilterScalars), e->stream));
  e->converged = 0;
  e->converged_pending = false;
  HIPCHK(e, e->d_keys.reserve((size_t)n * 3));
  hipLaunchKernelGGL(k_set_keys, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, s.dev(), n, e->d_keys.p);
  HIPCHK(e, hipGetLastError());
  bool handled = false;
  int stop = -1, leaf = 0, bins = 0;
  if (n >= 8192)
  {
    int rc = kld_tree_on_device(e, n, &handled, &stop, &leaf, &bins, true);
    if (rc != BPF_OK)
      return rc;
  }
  if (!handled)
  {
    std::vector<int> keys((size_t)n * 3);
    HIPCHK(e, hipMemcpyAsync(keys.data(), e->d_keys.p, keys.size() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->hist.clear();
    for (int i = 0; i < n; ++i)
      e->hist.insert(keys[3 * (size_t)i], keys[3 * (size_t)i + 1], keys[3 * (size_t)i + 2]);
    leaf = e->hist.leaf_count();
    bins = e->hist.bin_count();
    e->hist_matches_set = true;
  }
  e->leaf_count = leaf;
  e->bin_count = bins;
  return BPF_OK;
}
}  // namespace

int bpf_odom_set_model(bpf_engine* e, int model_type, double alpha1, double alpha2, double alpha3, double alpha4,
                       double alpha5)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (model_type < BPF_ODOM_MODEL_DIFF || model_type > BPF_ODOM_MODEL_GAUSSIAN)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "unknown odom model type");
  e->odom_model = model_type;
  e->odom_alpha[0] = alpha1;
  e->odom_alpha[1] = alpha2;
  e->odom_alpha[2] = alpha3;
  e->odom_alpha[3] = alpha4;
  e->odom_alpha[4] = alpha5;
  e->odom_configured = true;
  return BPF_OK;
}

int bpf_pf_init_with_gaussian(bpf_engine* e, const double mean[3], const double rotation[9], const double sigma[3])
{
  if (!e || !mean || !rotation || !sigma)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  const int n = e->max_samples;
  HIPCHK(e, e->d_init_rot.reserve(9));
  HIPCHK(e, hipMemcpyAsync(e->d_init_rot.p, rotation, 9 * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));  // `rotation` is the caller's memory
  SampleSet& dst = e->sets[e->cur];
  long long consumed = 0;
  int rc = generate_gaussians(e, 3ll * n, 0, 3ll * n, sigma, &consumed, [&]() {
    hipLaunchKernelGGL(k_init_gaussian, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, dst.dev(), n,
                       (const double*)e->d_gauss.p, mean[0], mean[1], mean[2], (const double*)e->d_init_rot.p,
                       1.0 / (double)n);
  });
  if (rc != BPF_OK)
    return rc;
  e->rng = lcg_skip_host(e->rng, (uint64_t)consumed, e->jump);
  return finish_init(e, n);
}

int bpf_pf_init_with_random_poses(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  FreeSpaceDev fs{};
  int rc = ensure_free_space(e, &fs);
  if (rc != BPF_OK)
    return rc;
  const int n = e->max_samples;
  hipLaunchKernelGGL(k_init_free_space, dim3(blocks_for(n, 256)), dim3(256), 0, e->stream, e->sets[e->cur].dev(), n,
                     e->rng, e->jump, fs, 1.0 / (double)n);
  HIPCHK(e, hipGetLastError());
  e->rng = lcg_skip_host(e->rng, 2ull * (uint64_t)n, e->jump);
  return finish_init(e, n);
}

int bpf_pf_update_action(bpf_engine* e, const double pose[3], const double delta[3], const double absolute_motion[3])
{
  if (!e || !pose || !delta || !absolute_motion)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf || !e->odom_configured)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create and bpf_odom_set_model first");
  return update_action(e, pose, delta, absolute_motion, 0, e->sample_count);
}

int bpf_shard_update_action(bpf_engine* e, const double pose[3], const double delta[3],
                            const double absolute_motion[3], long long global_first, long long global_count)
{
  if (!e || !pose || !delta || !absolute_motion)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf || !e->odom_configured)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create and bpf_odom_set_model first");
  return update_action(e, pose, delta, absolute_motion, global_first, global_count);
}

// ---------------------------------------------------------------------- cluster statistics
namespace
{
// particle_filter.cpp:505-636 on a host copy of the current set, in index order
int compute_cluster_stats(bpf_engine* e)
{
  if (e->stats_epoch == e->set_epoch)
    return BPF_OK;
  const int n = e->sample_count;
  std::vector<double> s((size_t)n * 4);
  int got = 0;
  int rc = bpf_pf_get_samples(e, s.data(), n, &got);
  if (rc != BPF_OK)
    return rc;
  if (!e->hist_matches_set)
  {
    // the histogram tree of this set is not at hand (set loaded with an explicit leaf count, or
    // restored): rebuild it the way initWith* / the resamplers do, by inserting every pose in order
    e->hist.clear();
    for (int i = 0; i < n; ++i)
    {
      int key[3];
      host_pose_key(s[4 * i], s[4 * i + 1], s[4 * i + 2], key);
      e->hist.insert(key[0], key[1], key[2]);
    }
    e->hist_matches_set = true;
  }
  e->hist.label_components();
  const int max_clusters = e->max_samples;  // cluster_max_count (particle_filter.cpp:84)
  struct Acc
  {
    int count = 0;
    double weight = 0, m[4] = { 0, 0, 0, 0 }, c[4] = { 0, 0, 0, 0 };
  };
  std::vector<Acc> acc;
  int cluster_count = 0;
  double weight = 0.0, m[4] = { 0, 0, 0, 0 }, c[4] = { 0, 0, 0, 0 };
  for (int i = 0; i < n; ++i)
  {
    const double* p = &s[4 * i];
    const double w = p[3];
    int key[3];
    host_pose_key(p[0], p[1], p[2], key);
    const int node = e->hist.find(key[0], key[1], key[2]);
    const int cidx = node < 0 ? -1 : e->hist.label_of(node);
    if (cidx < 0 || cidx >= max_clusters)
      continue;  // :574-576
    if (cidx + 1 > cluster_count)
      cluster_count = cidx + 1;
    if ((int)acc.size() < cluster_count)
      acc.resize(cluster_count);
    Acc& a = acc[cidx];
    a.count += 1;
    a.weight += w;
    a.m[0] += w * p[0];
    a.m[1] += w * p[1];
    a.m[2] += w * std::cos(p[2]);
    a.m[3] += w * std::sin(p[2]);
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 2; ++k)
        a.c[2 * j + k] += w * p[j] * p[k];
    weight += w;
    m[0] += w * p[0];
    m[1] += w * p[1];
    m[2] += w * std::cos(p[2]);
    m[3] += w * std::sin(p[2]);
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 2; ++k)
        c[2 * j + k] += w * p[j] * p[k];
  }
  e->clusters.assign((size_t)cluster_count, bpf_cluster{});
  for (int k = 0; k < cluster_count; ++k)
  {
    const Acc& a = acc[k];
    bpf_cluster& o = e->clusters[k];
    o.count = a.count;
    o.weight = a.weight;
    o.mean[0] = a.m[0] / a.weight;
    o.mean[1] = a.m[1] / a.weight;
    o.mean[2] = std::atan2(a.m[3], a.m[2]);
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 2; ++q)
        o.cov[2 * j + q] = a.c[2 * j + q] / a.weight - o.mean[j] * o.mean[q];
    o.cov[4] = -2 * std::log(std::sqrt(a.m[2] * a.m[2] + a.m[3] * a.m[3]));
  }
  e->set_mean[0] = m[0] / weight;
  e->set_mean[1] = m[1] / weight;
  e->set_mean[2] = std::atan2(m[3], m[2]);
  for (int j = 0; j < 2; ++j)
    for (int q = 0; q < 2; ++q)
      e->set_cov[2 * j + q] = c[2 * j + q] / weight - e->set_mean[j] * e->set_mean[q];
  e->set_cov[4] = -2 * std::log(std::sqrt(m[2] * m[2] + m[3] * m[3]));
  e->stats_epoch = e->set_epoch;
  return BPF_OK;
}
}  // namespace

int bpf_pf_compute_cluster_stats(bpf_engine* e, int* cluster_count_out, double set_mean[3], double set_cov[5])
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  int rc = compute_cluster_stats(e);
  if (rc != BPF_OK)
    return rc;
  if (cluster_count_out)
    *cluster_count_out = (int)e->clusters.size();
  if (set_mean)
    std::memcpy(set_mean, e->set_mean, sizeof(e->set_mean));
  if (set_cov)
    std::memcpy(set_cov, e->set_cov, sizeof(e->set_cov));
  return BPF_OK;
}

int bpf_pf_get_cluster(bpf_engine* e, int cidx, bpf_cluster* out)
{
  if (!e || !out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  int rc = compute_cluster_stats(e);
  if (rc != BPF_OK)
    return rc;
  if (cidx < 0 || cidx >= (int)e->clusters.size())
    return BPF_ERR_INVALID_ARGUMENT;  // getClusterStats returns false (particle_filter.cpp:642-643)
  *out = e->clusters[cidx];
  return BPF_OK;
}

int bpf_pf_get_max_weight_pose(bpf_engine* e, double* max_weight, double pose[3])
{
  if (!e || !max_weight || !pose)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  int rc = compute_cluster_stats(e);
  if (rc != BPF_OK)
    return rc;
  double best = 0.0;
  int hyp = -1;
  for (size_t k = 0; k < e->clusters.size(); ++k)
    if (e->clusters[k].weight > best)  // node_2d.cpp:608-612
    {
      best = e->clusters[k].weight;
      hyp = (int)k;
    }
  *max_weight = best;
  if (hyp >= 0)
    std::memcpy(pose, e->clusters[hyp].mean, 3 * sizeof(double));
  return BPF_OK;
}

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

// ---------------------------------------------------------------------- wire formats
int bpf_wire_laserscan_to_planar(const float* ranges, int n, float msg_range_min, float msg_range_max,
                                 double sensor_min_range, double sensor_max_range, double angle_min,
                                 double angle_increment, double* ranges_out, double* angles_out, double* range_max_out)
{
  if (!ranges || n < 0 || !ranges_out || !angles_out || !range_max_out)
    return BPF_ERR_INVALID_ARGUMENT;
  // node_2d.cpp:535-543
  double range_max;
  if (sensor_max_range > 0.0)
    range_max = std::min(msg_range_max, static_cast<float>(sensor_max_range));
  else
    range_max = msg_range_max;
  double range_min;
  if (sensor_min_range > 0.0)
    range_min = std::max(msg_range_min, static_cast<float>(sensor_min_range));
  else
    range_min = msg_range_min;
  for (int i = 0; i < n; ++i)
  {
    // :548-558: short readings become max range; bearing = angle_min + i * increment
    if (ranges[i] <= range_min)
      ranges_out[i] = range_max;
    else
      ranges_out[i] = ranges[i];
    angles_out[i] = angle_min + (i * angle_increment);
  }
  *range_max_out = range_max;
  return BPF_OK;
}

namespace
{
struct Quat
{
  double x, y, z, w;
};
Quat quat_from_yaw(double yaw)
{
  // tf2::Quaternion::setRPY(0, 0, yaw): with zero roll / pitch the products reduce to this
  const double h = yaw * 0.5;
  return Quat{ 0.0, 0.0, std::sin(h), std::cos(h) };
}
Quat quat_mul(const Quat& a, const Quat& b)
{
  // tf2 operator*(Quaternion, Quaternion)
  return Quat{ a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
               a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z };
}
double quat_yaw(const Quat& q)
{
  // tf2::getYaw (tf2/impl/utils.h): gimbal-lock cases first, then the usual atan2
  const double sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, sqw = q.w * q.w;
  const double sarg = -2 * (q.x * q.z - q.w * q.y) / (sqx + sqy + sqz + sqw);
  if (sarg <= -0.99999)
    return -2 * std::atan2(q.y, q.x);
  if (sarg >= 0.99999)
    return 2 * std::atan2(q.y, q.x);
  return std::atan2(2 * (q.x * q.y + q.w * q.z), sqw + sqx - sqy - sqz);
}
}  // namespace

int bpf_wire_scan_angle_stats(double msg_angle_min, double msg_angle_increment, const double q_base_from_scanner[4],
                              double* angle_min_out, double* angle_increment_out)
{
  if (!q_base_from_scanner || !angle_min_out || !angle_increment_out)
    return BPF_ERR_INVALID_ARGUMENT;
  const Quat t{ q_base_from_scanner[0], q_base_from_scanner[1], q_base_from_scanner[2], q_base_from_scanner[3] };
  // node_2d.cpp:503-526: doTransform on a quaternion message is t.rotation * q
  const Quat min_q = quat_mul(t, quat_from_yaw(msg_angle_min));
  const Quat inc_q = quat_mul(t, quat_from_yaw(msg_angle_min + msg_angle_increment));
  const double amin = quat_yaw(min_q);
  double inc = quat_yaw(inc_q) - amin;
  const double r = std::fmod(inc + M_PI, 2.0 * M_PI);  // angles::normalize_angle, Noetic form
  inc = (r <= 0.0) ? r + M_PI : r - M_PI;
  *angle_min_out = amin;
  *angle_increment_out = inc;
  return BPF_OK;
}

int bpf_wire_occupancy_grid_to_cells(const int8_t* data, int width, int height, double msg_resolution,
                                     double msg_origin_x, double msg_origin_y, int map_scale_up_factor,
                                     int32_t* cells_out, int* size_x_out, int* size_y_out, float origin_out[2],
                                     double* resolution_out)
{
  if (!data || width <= 0 || height <= 0 || map_scale_up_factor < 1 || !cells_out || !size_x_out || !size_y_out ||
      !origin_out || !resolution_out)
    return BPF_ERR_INVALID_ARGUMENT;
  // node_2d.cpp:267-277
  const int f = map_scale_up_factor;
  const double resolution = msg_resolution / f;
  const int sx = width * f, sy = height * f;
  const double x_origin = msg_origin_x + (sx / 2) * resolution;
  const double y_origin = msg_origin_y + (sy / 2) * resolution;
  origin_out[0] = (float)x_origin;  // pcl::PointXYZ narrows to float
  origin_out[1] = (float)y_origin;
  for (int y = 0; y < sy; ++y)
  {
    int i = y * sx;
    const int msg_row = (y / f) * width;
    for (int x = 0; x < sx; ++x, ++i)
    {
      const int8_t v = data[msg_row + x / f];
      cells_out[i] = (v == 0) ? -1 : (v == 100 ? 1 : 0);  // :285-290
    }
  }
  *size_x_out = sx;
  *size_y_out = sy;
  *resolution_out = resolution;
  return BPF_OK;
}

int bpf_wire_decimate_cloud(const float* points_xyz, int n_points, int max_beams, float* out_xyz, int capacity)
{
  if (!points_xyz || !out_xyz || n_points < 0 || max_beams < 2)
    return -1;
  int step = (n_points - 1) / (max_beams - 1);  // node_3d.cpp:471-472
  step = std::max(step, 1);
  int k = 0;
  for (int i = 0; i < n_points; i += step)
  {
    if (k >= capacity)
      return -1;
    out_xyz[3 * k] = points_xyz[3 * i];
    out_xyz[3 * k + 1] = points_xyz[3 * i + 1];
    out_xyz[3 * k + 2] = points_xyz[3 * i + 2];
    ++k;
  }
  return k;
}

int bpf_wire_samples_to_pose_array(const double* samples, int sample_count, double* poses7_out)
{
  if (!samples || !poses7_out || sample_count < 0)
    return BPF_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < sample_count; ++i)
  {
    // tf2::Quaternion::setRPY(0, 0, yaw) (third party, tf2 LinearMath): with zero roll and pitch the
    // products reduce to (0, 0, sin(yaw/2), cos(yaw/2))
    const double h = samples[4 * i + 2] * 0.5;
    double* o = &poses7_out[7 * i];
    o[0] = samples[4 * i];
    o[1] = samples[4 * i + 1];
    o[2] = 0.0;
    o[3] = 0.0;
    o[4] = 0.0;
    o[5] = std::sin(h);
    o[6] = std::cos(h);
  }
  return BPF_OK;
}

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
  // every column start must leave room for a whole z column (octomap.cpp:315-333)
  for (size_t i = 0; i < n_pose_indices; ++i)
    if ((size_t)pose_indices[i] + (size_t)nz > n_distance_ratios)
      return e->fail(BPF_ERR_INVALID_ARGUMENT, "pose_indices entry points past distance_ratios");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, e->d_pose_indices.reserve(n_pose_indices));
  HIPCHK(e, e->d_ratios.reserve(n_distance_ratios));
  HIPCHK(e, hipMemcpy(e->d_pose_indices.p, pose_indices, n_pose_indices * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIPCHK(e, hipMemcpy(e->d_ratios.p, distance_ratios, n_distance_ratios, hipMemcpyHostToDevice));
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
  return BPF_OK;
}

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
    HIPCHK(e, hipMemcpy(pose_indices, e->d_pose_indices.p, e->n_pose_indices * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  if (distance_ratios)
  {
    if (ratios_capacity < e->n_ratios)
      return e->fail(BPF_ERR_CAPACITY, "distance_ratios output too small");
    HIPCHK(e, hipMemcpy(distance_ratios, e->d_ratios.p, e->n_ratios, hipMemcpyDeviceToHost));
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
  {
    // exact reciprocal?  1/res must fit 29 bits (so float * rinv is exact) and rinv*res must round to 1
    const double rinv = e->map3.inv_resolution;
    uint64_t bits;
    std::memcpy(&bits, &rinv, 8);
    const bool exact_rinv = (bits & ((1ull << 24) - 1)) == 0 && std::fabs(std::fma(rinv, e->map3.resolution, -1.0)) < 1.1e-16;
    ProfScope ps(e, BPF_K_SCORE);
    if (exact_rinv)
      hipLaunchKernelGGL(k_cloud_score<true>, dim3(n_chunks, A.slabs), dim3(256), 0, e->stream, A);
    else
      hipLaunchKernelGGL(k_cloud_score<false>, dim3(n_chunks, A.slabs), dim3(256), 0, e->stream, A);
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
  hipLaunchKernelGGL(k_soa_to_aos, dim3(blocks_for(sample_count, 256)), dim3(256), 0, e->stream, e->scratch.dev(),
                     e->d_aos.p, sample_count);
  if (hipMemcpyAsync(e->h_aos.p, e->d_aos.p, (size_t)sample_count * sizeof(double4), hipMemcpyDeviceToHost,
                     e->stream) != hipSuccess ||
      hipMemcpyAsync(e->h_scalars.p, e->d_scalars.p, sizeof(FilterScalars), hipMemcpyDeviceToHost, e->stream) !=
          hipSuccess ||
      hipStreamSynchronize(e->stream) != hipSuccess)
    return bail(e->fail(BPF_ERR_HIP, "copy back"));
  for (int i = 0; i < sample_count; ++i)
    samples[4 * i + 3] = e->h_aos.p[i].w;
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
  e->tile_sums_n = -1;
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

// ---------------------------------------------------------------------- sharded stages
namespace
{
// local weight total into scalars[0] after a sharded scoring stage
int shard_local_total(bpf_engine* e)
{
  SampleSet& s = e->sets[e->cur];
  if (e->fused_partials > 0)
  {
    // the scoring kernel left per-block partials: one small launch folds them into the local total
    ProfScope ps(e, BPF_K_REDUCE);
    hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(BPF_RED_BLOCK), 0, e->stream, e->d_block_partials.p,
                       e->fused_partials, e->d_scalars.p, 0);
    HIPCHK(e, hipGetLastError());
    e->fused_partials = 0;
    return BPF_OK;
  }
  return sum_into_slot(e, s.w.p, e->sample_count, 0, 0, e->sample_count);
}
}  // namespace

int bpf_shard_score_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                           double range_max)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (e->pm.max_beams < 2)
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  bool forced_zero = false;
  e->skip_pending = false;
  int rc = score_planar(e, s.dev(), e->sample_count, e->converged, ranges, angles, range_count, range_max,
                        &forced_zero, true, true);
  if (rc != BPF_OK)
    return rc;
  if (e->skip_pending)
    return BPF_SHARD_NEED_BEAM_COUNTS;  // sum bpf_shard_beam_counts_dev over the shards, then ..._finish
  return shard_local_total(e);
}

int bpf_shard_beam_counts_dev(bpf_engine* e, void** counts_dev, int* n_counts)
{
  if (!e || !counts_dev || !n_counts)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->skip_pending)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no beam-skip counting pass is pending");
  *counts_dev = e->d_obs_count.p;
  *n_counts = std::max(e->skip_fs.n_staged, 1);
  return BPF_OK;
}

int bpf_shard_score_planar_finish(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                  double range_max, long long global_count)
{
  if (!e || !ranges || !angles || global_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf || !e->skip_pending)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "no beam-skip counting pass is pending");
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  bool forced_zero = false;
  int rc = score_planar_beamskip_finish(e, s.dev(), e->sample_count, global_count, ranges, angles, range_count,
                                        range_max, &forced_zero, true);
  if (rc != BPF_OK)
    return rc;
  return shard_local_total(e);
}

int bpf_shard_score_cloud(bpf_engine* e, const float* points_xyz, int n_points)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  if (e->cloud_max_beams < 2)
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  e->tile_sums_n = -1;
  int rc = score_cloud(e, s.dev(), e->sample_count, points_xyz, n_points);
  if (rc != BPF_OK)
    return rc;
  return sum_into_slot(e, s.w.p, e->sample_count, 0, 0, e->sample_count);
}

int bpf_shard_scalars_dev(bpf_engine* e, void** dev_ptr)
{
  if (!e || !dev_ptr)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  int rc = ensure_scalars(e);
  if (rc != BPF_OK)
    return rc;
  *dev_ptr = e->d_scalars.p;
  return BPF_OK;
}

int bpf_shard_normalize_dev(bpf_engine* e, const void* totals_dev, int world, int global_sample_count)
{
  if (!e || !e->have_pf || !totals_dev || world <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& s = e->sets[e->cur];
  const int n = e->sample_count;
  const int nb = std::max(1, blocks_for(n, BPF_RED_TILE));
  HIPCHK(e, e->d_tile_sums.reserve((size_t)nb));
  ProfScope ps(e, BPF_K_NORMALIZE);
  hipLaunchKernelGGL(k_normalize_gathered, dim3(nb), dim3(BPF_RED_BLOCK), 0, e->stream, s.w.p, n,
                     static_cast<const double*>(totals_dev), world, global_sample_count, e->d_scalars.p,
                     e->alpha_slow, e->alpha_fast, e->d_tile_sums.p);
  HIPCHK(e, hipGetLastError());
  e->tile_sums_n = n;
  return BPF_OK;
}

int bpf_shard_build_cdf(bpf_engine* e, void* flags_dev)
{
  if (!e || !e->have_pf)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  if (flags_dev)
    HIPCHK(e, hipMemsetAsync(flags_dev, 0, sizeof(int), e->stream));
  int rc = build_cdf(e, e->sets[e->cur].w.p, e->sample_count);
  if (rc != BPF_OK)
    return rc;
  HIPCHK(e, hipMemcpyAsync(&e->d_scalars.p->v[7], e->d_cdf.p + e->sample_count, sizeof(double),
                           hipMemcpyDeviceToDevice, e->stream));
  return BPF_OK;
}

int bpf_shard_draw_window_dev(bpf_engine* e, uint64_t rng_state48, int m0, int m1, const void* sums_dev,
                              int sums_are_totals, int rank, int world, void* window_dev, int stride, void* flags_dev)
{
  if (!e || !e->have_pf || !sums_dev || !window_dev || !flags_dev || m1 <= m0 || stride < m1 - m0 || rank < 0 ||
      rank >= world)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad draw window arguments") : BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  WindowArgs A{};
  A.src = e->sets[e->cur].dev();
  A.n_src = e->sample_count;
  A.cdf = e->d_cdf.p;
  A.sums = static_cast<const double*>(sums_dev);
  A.sums_are_totals = sums_are_totals;
  A.rank = rank;
  A.world = world;
  A.m0 = m0;
  A.m1 = m1;
  A.rng_state = rng_state48;
  A.jump = e->jump;
  A.window = static_cast<long long*>(window_dev);
  A.stride = stride;
  A.flags = static_cast<int*>(flags_dev);
  if (e->shard_chain)
  {
    // w_diff > 0 (bpf_shard_begin_resample built the chain from this same stream state)
    if (rng_state48 != e->shard_rng0 || m1 > e->max_samples)
      return e->fail(BPF_ERR_INVALID_ARGUMENT, "draw window does not belong to the resample begun");
    A.chain = e->d_chain.p;
    A.write_random = rank == 0;
    int rcf = ensure_free_space(e, &A.free_space);
    if (rcf != BPF_OK)
      return rcf;
  }
  ProfScope ps(e, BPF_K_DRAW);
  hipLaunchKernelGGL(k_draw_window, dim3(blocks_for(m1 - m0, 256)), dim3(256), 0, e->stream, A);
  HIPCHK(e, hipGetLastError());
  return BPF_OK;
}

int bpf_shard_adopt_dev(bpf_engine* e, const void* x_dev, const void* y_dev, const void* theta_dev, int count,
                        int global_count, int leaf_count, int bin_count)
{
  if (!e || !e->have_pf || count < 0 || global_count <= 0 || (count > 0 && (!x_dev || !y_dev || !theta_dev)))
    return BPF_ERR_INVALID_ARGUMENT;
  if (count > e->max_samples)
    return e->fail(BPF_ERR_CAPACITY, "adopted shard larger than max_samples");
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& b = e->sets[e->cur ^ 1];
  if (count > 0)
  {
    ProfScope ps(e, BPF_K_FINALIZE);
    hipLaunchKernelGGL(k_adopt, dim3(blocks_for(count, 256)), dim3(256), 0, e->stream,
                       static_cast<const double*>(x_dev), static_cast<const double*>(y_dev),
                       static_cast<const double*>(theta_dev), b.dev(), count, 1.0 / (double)global_count);
    HIPCHK(e, hipGetLastError());
  }
  e->cur ^= 1;
  e->sample_count = count;
  e->leaf_count = leaf_count;
  e->bin_count = bin_count;
  e->tile_sums_n = -1;
  return BPF_OK;
}

int bpf_shard_tail_small_dev(bpf_engine* e, const void* x_all_dev, const void* y_all_dev, const void* theta_all_dev,
                             int global_count, int lo, int hi, int leaf_count, int bin_count)
{
  if (!e || !e->have_pf || !x_all_dev || !y_all_dev || !theta_all_dev || global_count <= 0 || lo < 0 || hi < lo ||
      hi > global_count)
    return BPF_ERR_INVALID_ARGUMENT;
  if (hi - lo > e->max_samples)
    return e->fail(BPF_ERR_CAPACITY, "adopted shard larger than max_samples");
  HIPCHK(e, hipSetDevice(e->device));
  SampleSet& b = e->sets[e->cur ^ 1];
  {
    ProfScope ps(e, BPF_K_FINALIZE);
    hipLaunchKernelGGL(k_shard_tail_small, dim3(1), dim3(1024), 0, e->stream, static_cast<const double*>(x_all_dev),
                       static_cast<const double*>(y_all_dev), static_cast<const double*>(theta_all_dev), global_count,
                       lo, hi, b.dev(), e->dist_threshold, e->d_scalars.p, e->d_flags.p + 1);
  }
  HIPCHK(e, hipGetLastError());
  e->cur ^= 1;
  e->sample_count = hi - lo;
  e->leaf_count = leaf_count;
  e->bin_count = bin_count;
  e->tile_sums_n = -1;
  e->converged_pending = true;
  e->conv_n = global_count;
  return BPF_OK;
}

int bpf_shard_converged_dev(bpf_engine* e, const void* x_all_dev, const void* y_all_dev, int global_count)
{
  if (!e || !e->have_pf || !x_all_dev || !y_all_dev || global_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  const double* x = static_cast<const double*>(x_all_dev);
  const double* y = static_cast<const double*>(y_all_dev);
  int rc = sum_into_slot(e, x, global_count, 3, 0, global_count);
  if (rc != BPF_OK)
    return rc;
  rc = sum_into_slot(e, y, global_count, 4, 0, global_count);
  if (rc != BPF_OK)
    return rc;
  HIPCHK(e, hipMemsetAsync(e->d_flags.p + 1, 0, sizeof(int), e->stream));
  const int grid = std::max(1, std::min(blocks_for(global_count, 256), 1024));
  hipLaunchKernelGGL(k_count_converged, dim3(grid), dim3(256), 0, e->stream, x, y, global_count, e->d_scalars.p,
                     e->dist_threshold, e->d_flags.p + 1);
  HIPCHK(e, hipGetLastError());
  e->converged_pending = true;
  e->conv_n = global_count;
  return BPF_OK;
}

uint64_t bpf_drand48_skip(uint64_t state48, uint64_t n)
{
  static LcgJump J;
  static bool init = false;
  if (!init)
  {
    lcg_tables(J);
    init = true;
  }
  return lcg_skip_host(state48 & ((1ull << 48) - 1), n, J);
}

int bpf_kld_reset(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  e->hist.clear();
  e->seen.reset((size_t)std::min(std::max(e->max_samples, 1024), 1 << 20));
  return BPF_OK;
}

int bpf_kld_feed(bpf_engine* e, const void* keys, int keys_are_int64, int stride, int n_keys, int first_draw_index,
                 int* stop_count_out)
{
  if (!e || !keys || !stop_count_out || stride < n_keys)
    return BPF_ERR_INVALID_ARGUMENT;
  *stop_count_out = -1;
  const long long* k64 = static_cast<const long long*>(keys);
  const int* k32 = static_cast<const int*>(keys);
  int cached_leaf = -1, cached_limit = 0;
  for (int q = 0; q < n_keys; ++q)
  {
    int k[3];
    for (int d = 0; d < 3; ++d)
      k[d] = keys_are_int64 ? (int)k64[(size_t)d * stride + q] : k32[(size_t)d * stride + q];
    if (e->seen.first_time(k[0], k[1], k[2]))
      e->hist.insert(k[0], k[1], k[2]);
    const int lc = e->hist.leaf_count();
    if (lc != cached_leaf)
    {
      cached_leaf = lc;
      cached_limit = resample_limit(lc, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
    }
    const int count = first_draw_index + q + 1;
    if (count > cached_limit)
    {
      *stop_count_out = count;
      break;
    }
  }
  return BPF_OK;
}

int bpf_kld_feed_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys, int first_draw_index,
                     int* stop_count_out)
{
  if (!e || !window_dev || !stop_count_out || stride < n_keys || n_keys <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->h_keys.reserve((size_t)n_keys * 3));
  const unsigned generation = ++e->done_generation;
  hipLaunchKernelGGL(k_publish_window_keys, dim3(blocks_for(n_keys, 256)), dim3(256), 0, e->stream,
                     static_cast<const long long*>(window_dev), stride, n_keys, e->h_keys.p,
                     reinterpret_cast<unsigned*>(e->d_flags.p + 4), reinterpret_cast<volatile unsigned*>(e->h_done.p),
                     generation);
  HIPCHK(e, hipGetLastError());
  if (!wait_generation(e, generation))
    HIPCHK(e, hipStreamSynchronize(e->stream));
  return bpf_kld_feed(e, e->h_keys.p, 0, n_keys, n_keys, first_draw_index, stop_count_out);
}

int bpf_shard_begin_resample(bpf_engine* e, uint64_t rng_state48, int leaf_count, double* w_diff_out,
                             int* systematic_count_out)
{
  if (!e || !w_diff_out || !systematic_count_out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  // w_diff = max(0, 1 - w_fast / w_slow) (particle_filter.cpp:438-440); the averages are the same on every shard
  double w_diff = 0.0;
  if (e->alpha_slow != 0.0 || e->alpha_fast != 0.0)
  {
    int rc = fetch_scalars(e);
    if (rc != BPF_OK)
      return rc;
    w_diff = 1.0 - e->h_scalars.p->v[2] / e->h_scalars.p->v[1];
    if (!(w_diff >= 0.0))
      w_diff = 0.0;
  }
  e->w_diff_last = w_diff;
  e->shard_w_diff = w_diff;
  e->shard_chain = false;
  e->shard_n_random = 0;
  e->shard_rng0 = rng_state48 & ((1ull << 48) - 1);
  int count = resample_limit(leaf_count, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
  if (w_diff > 0.0)
  {
    if (e->resample_model == BPF_RESAMPLE_SYSTEMATIC)
    {
      count *= (1.0 + w_diff);  // :295-306
      if (count > e->max_samples)
        count = e->max_samples;
      e->shard_n_random = (int)(w_diff * count);
    }
    else
    {
      FreeSpaceDev fs{};
      int rc = ensure_free_space(e, &fs);
      if (rc != BPF_OK)
        return rc;
      const uint64_t keep = e->rng;
      e->rng = e->shard_rng0;
      rc = build_draw_chain(e, w_diff, e->max_samples);
      e->rng = keep;
      if (rc != BPF_OK)
        return rc;
      e->shard_chain = true;
    }
  }
  *w_diff_out = w_diff;
  *systematic_count_out = count;
  return BPF_OK;
}

int bpf_shard_end_resample(bpf_engine* e, int sample_count, uint64_t* rng_state48_out)
{
  if (!e || !rng_state48_out || sample_count <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  uint64_t consumed;
  if (e->resample_model == BPF_RESAMPLE_SYSTEMATIC)
    consumed = 1ull + 2ull * (uint64_t)e->shard_n_random;
  else if (e->shard_chain)
  {
    if (sample_count > e->max_samples)
      return e->fail(BPF_ERR_INVALID_ARGUMENT, "sample_count beyond the chain");
    HIPCHK(e, e->h_chain_word.reserve(1));
    HIPCHK(e, hipMemcpyAsync(e->h_chain_word.p, e->d_chain.p + sample_count, sizeof(int), hipMemcpyDeviceToHost,
                             e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    consumed = (uint64_t)((unsigned)e->h_chain_word.p[0] & 0x7fffffffu) - 1ull;
  }
  else
    consumed = 2ull * (uint64_t)sample_count;
  *rng_state48_out = lcg_skip_host(e->shard_rng0, consumed, e->jump);
  if (e->shard_w_diff > 0.0)  // particle_filter.cpp:453-455
    HIPCHK(e, hipMemsetAsync(&e->d_scalars.p->v[1], 0, 2 * sizeof(double), e->stream));
  e->shard_chain = false;
  e->shard_n_random = 0;
  return BPF_OK;
}

int bpf_pf_resample_limit(bpf_engine* e, int leaf_count, int* count_out)
{
  if (!e || !count_out)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  *count_out = resample_limit(leaf_count, e->min_samples, e->max_samples, e->pop_err, e->pop_z);
  return BPF_OK;
}

int bpf_shard_systematic_window_dev(bpf_engine* e, uint64_t rng_state48, int count, const void* sums_dev,
                                    int sums_are_totals, int rank, int world, void* window_dev, int stride,
                                    void* flags_dev)
{
  if (!e || !e->have_pf || !sums_dev || !window_dev || !flags_dev || count <= 0 || stride < count || rank < 0 ||
      rank >= world)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "bad systematic window arguments") : BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->h_targets.reserve((size_t)std::max(count, e->max_samples)));
  // the reference's serial chain (particle_filter.cpp:337-341): target += delta, -= 1 once it passes 1
  const uint64_t st = lcg_skip_host(rng_state48 & ((1ull << 48) - 1), 1, e->jump);
  double t = std::ldexp((double)st, -48);
  const int n_random = e->shard_n_random;
  if (n_random < 0 || n_random >= count)
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "systematic window: random pose count out of range");
  const int n_systematic = count - n_random;
  const double delta = 1.0 / n_systematic;
  if (e->targets_read)  // a previous window kernel may still be reading the pinned targets
    HIPCHK(e, hipEventSynchronize(e->targets_read));
  for (int i = 0; i < n_systematic; ++i)
  {
    e->h_targets.p[i] = t;
    t += delta;
    if (t > 1.0)
      t -= 1.0;
  }
  WindowArgs A{};
  A.src = e->sets[e->cur].dev();
  A.n_src = e->sample_count;
  A.cdf = e->d_cdf.p;
  A.sums = static_cast<const double*>(sums_dev);
  A.sums_are_totals = sums_are_totals;
  A.rank = rank;
  A.world = world;
  A.m0 = 0;
  A.m1 = count;
  A.rng_state = rng_state48 & ((1ull << 48) - 1);
  A.jump = e->jump;
  A.n_random = n_random;
  A.write_random = rank == 0;
  if (n_random > 0)
  {
    int rcf = ensure_free_space(e, &A.free_space);
    if (rcf != BPF_OK)
      return rcf;
  }
  A.window = static_cast<long long*>(window_dev);
  A.stride = stride;
  A.flags = static_cast<int*>(flags_dev);
  A.targets = e->h_targets.p;
  {
    ProfScope ps(e, BPF_K_DRAW);
    hipLaunchKernelGGL(k_draw_window, dim3(blocks_for(count, 256)), dim3(256), 0, e->stream, A);
  }
  HIPCHK(e, hipGetLastError());
  if (!e->targets_read)
    HIPCHK(e, hipEventCreateWithFlags(&e->targets_read, hipEventDisableTiming));
  HIPCHK(e, hipEventRecord(e->targets_read, e->stream));
  return BPF_OK;
}

int bpf_kld_insert(bpf_engine* e, const void* keys, int keys_are_int64, int stride, int n_keys)
{
  if (!e || !keys || stride < n_keys)
    return BPF_ERR_INVALID_ARGUMENT;
  const long long* k64 = static_cast<const long long*>(keys);
  const int* k32 = static_cast<const int*>(keys);
  for (int q = 0; q < n_keys; ++q)
  {
    int k[3];
    for (int d = 0; d < 3; ++d)
      k[d] = keys_are_int64 ? (int)k64[(size_t)d * stride + q] : k32[(size_t)d * stride + q];
    if (e->seen.first_time(k[0], k[1], k[2]))
      e->hist.insert(k[0], k[1], k[2]);
  }
  return BPF_OK;
}

int bpf_kld_insert_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys)
{
  if (!e || !window_dev || stride < n_keys || n_keys <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->h_keys.reserve((size_t)n_keys * 3));
  const unsigned generation = ++e->done_generation;
  hipLaunchKernelGGL(k_publish_window_keys, dim3(blocks_for(n_keys, 256)), dim3(256), 0, e->stream,
                     static_cast<const long long*>(window_dev), stride, n_keys, e->h_keys.p,
                     reinterpret_cast<unsigned*>(e->d_flags.p + 4), reinterpret_cast<volatile unsigned*>(e->h_done.p),
                     generation);
  HIPCHK(e, hipGetLastError());
  if (!wait_generation(e, generation))
    HIPCHK(e, hipStreamSynchronize(e->stream));
  return bpf_kld_insert(e, e->h_keys.p, 0, n_keys, n_keys);
}

int bpf_kld_stop_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys, int* handled_out,
                     int* stop_count_out, int* leaf_count_out, int* bin_count_out)
{
  if (!e || !window_dev || !handled_out || !stop_count_out || !leaf_count_out || !bin_count_out || stride < n_keys ||
      n_keys <= 0)
    return BPF_ERR_INVALID_ARGUMENT;
  if (!e->have_pf)
    return e->fail(BPF_ERR_NOT_CONFIGURED, "bpf_pf_create first");
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->d_keys.reserve((size_t)n_keys * 3));
  hipLaunchKernelGGL(k_window_keys_to_aos, dim3(blocks_for(n_keys, 256)), dim3(256), 0, e->stream,
                     static_cast<const long long*>(window_dev), stride, n_keys, e->d_keys.p);
  HIPCHK(e, hipGetLastError());
  bool handled = false;
  *stop_count_out = -1;
  *leaf_count_out = *bin_count_out = 0;
  int rc = kld_tree_on_device(e, n_keys, &handled, stop_count_out, leaf_count_out, bin_count_out);
  *handled_out = handled ? 1 : 0;
  return rc;
}

int bpf_kld_leaf_count(bpf_engine* e, int* leaf_count_out, int* bin_count_out)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (leaf_count_out)
    *leaf_count_out = e->hist.leaf_count();
  if (bin_count_out)
    *bin_count_out = e->hist.bin_count();
  return BPF_OK;
}

int bpf_set_option(bpf_engine* e, int option, int value)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (option == BPF_OPT_CDF_SERIAL)
    e->cdf_serial = value != 0;
  else if (option == BPF_OPT_WINDOW_PATH)
    e->window_enabled = value != 0;
  else if (option == BPF_OPT_COUNT_CELLS)
    e->count_cells = value != 0;
  else if (option == BPF_OPT_KLD_DEVICE_MIN)
    e->kld_device_min = value > 0 ? value : 0x7fffffff;
  else
    return e->fail(BPF_ERR_INVALID_ARGUMENT, "unknown option");
  return BPF_OK;
}

int bpf_get_cells_walked(bpf_engine* e, unsigned long long* out, int reset)
{
  if (!e || !out)
    return BPF_ERR_INVALID_ARGUMENT;
  *out = 0;
  if (!e->d_cells_walked.p)
    return BPF_OK;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipMemcpyAsync(out, e->d_cells_walked.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (reset)
    HIPCHK(e, hipMemsetAsync(e->d_cells_walked.p, 0, sizeof(unsigned long long), e->stream));
  return BPF_OK;
}

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
  e->profile_all = on >= 2;
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

}  // extern "C"
