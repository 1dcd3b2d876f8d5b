// adapter.hpp -- header-only C++ host side above the C-ABI (badger_pf.h), shaped like the
// reference's classes so that Node2D-style code needs a type swap only:
//
//   badger_amcl_amd::OccupancyMap      <- OccupancyMap      (include/amcl/map/occupancy_map.h:54-123)
//   badger_amcl_amd::PlanarData        <- PlanarData        (include/amcl/sensors/planar_scanner.h:45-54)
//   badger_amcl_amd::PlanarScanner     <- PlanarScanner     (include/amcl/sensors/planar_scanner.h:57-168)
//   badger_amcl_amd::ParticleFilter    <- ParticleFilter    (include/amcl/pf/particle_filter.h:92-184)
//   badger_amcl_amd::PFSample / PFSampleSet                 (include/amcl/pf/particle_filter.h:41-87)
//   badger_amcl_amd::OdomData / Odom   <- OdomData / Odom   (include/amcl/sensors/odom.h:43-90)
//
// Same method names, argument meaning and return conventions (false / 0.0 on the reference's
// silent failures); conditions the reference asserts on or hangs in surface as std::runtime_error.
// No Eigen / ROS / PCL dependency: poses are plain double[3].
#pragma once
#include <array>
#include <cstdint>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../badger_pf.h"

namespace badger_amcl_amd
{

enum MapCellState { CELL_FREE = -1, CELL_UNKNOWN = 0, CELL_OCCUPIED = 1 };
enum PFResampleModelType { PF_RESAMPLE_MULTINOMIAL = BPF_RESAMPLE_MULTINOMIAL, PF_RESAMPLE_SYSTEMATIC = BPF_RESAMPLE_SYSTEMATIC };

class Engine
{
public:
  explicit Engine(int device = 0)
  {
    const int rc = bpf_create(device, &h_);
    if (rc != BPF_OK)
      throw std::runtime_error(std::string("bpf_create: ") + bpf_error_string(rc));
  }
  ~Engine() { bpf_destroy(h_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  bpf_engine* get() const { return h_; }
  void check(int rc) const
  {
    if (rc != BPF_OK)
      throw std::runtime_error(std::string("bpf: ") + bpf_last_error_message(h_) + " (" + bpf_error_string(rc) + ")");
  }

private:
  bpf_engine* h_ = nullptr;
};

struct PFSample
{
  std::array<double, 3> pose;  // x, y, theta
  double weight;
};

// Pins a sample buffer the caller owns for as long as this object lives (bpf_host_buffer_register /
// _unregister): what a ParticleFilter that keeps its `samples` vector in host memory (allocated once at max_samples,
// particle_filter.cpp:62-89) holds beside the vector, so that applyModelToSampleSet / initWithSamples / getCurrentSet
// move it at PCIe rate.  The vector must not be resized while it is pinned.
class PinnedSamples
{
public:
  PinnedSamples(std::shared_ptr<Engine> e, std::vector<PFSample>& samples) : e_(std::move(e)), ptr_(samples.data())
  {
    e_->check(bpf_host_buffer_register(e_->get(), ptr_, samples.size() * sizeof(PFSample)));
  }
  ~PinnedSamples() { (void)bpf_host_buffer_unregister(e_->get(), ptr_); }
  PinnedSamples(const PinnedSamples&) = delete;
  PinnedSamples& operator=(const PinnedSamples&) = delete;

private:
  std::shared_ptr<Engine> e_;
  void* ptr_;
};
static_assert(sizeof(PFSample) == 32, "PFSample must match the reference's 32-byte AoS record");

struct PFSampleSet
{
  int sample_count = 0;
  std::vector<PFSample> samples;
  int converged = 0;
  int leaf_count = 0;
};

class OccupancyMap
{
public:
  OccupancyMap(std::shared_ptr<Engine> e, double resolution) : e_(std::move(e)), resolution_(resolution) {}
  void setOrigin(float x, float y) { ox_ = x; oy_ = y; dirty_ = true; }
  std::vector<int> getSize() const { return { size_x_, size_y_ }; }
  void setSize(const std::vector<int>& size_vec)
  {
    size_x_ = size_vec[0];
    size_y_ = size_vec[1];
    cells_.assign((size_t)size_x_ * size_y_, CELL_UNKNOWN);
    dirty_ = true;
  }
  unsigned computeCellIndex(int i, int j) const { return i + j * unsigned(size_x_); }
  void setCellState(int index, MapCellState s) { cells_[index] = s; dirty_ = true; }
  MapCellState getCellState(int i, int j) const { return (MapCellState)cells_[computeCellIndex(i, j)]; }
  bool isValid(const std::vector<int>& c) const { return c[0] >= 0 && c[0] < size_x_ && c[1] >= 0 && c[1] < size_y_; }
  // adopt a host-built distances_lut_ (e.g. the reference's brushfire result)
  void setDistancesLUT(const std::vector<float>& lut, double max_distance_to_object)
  {
    lut_ = lut;
    max_dist_ = max_distance_to_object;
    dirty_ = true;
  }
  // OccupancyMap::updateDistancesLUT (occupancy_map.cpp:138-252): the reference's own priority-queue brushfire, the
  // reference's values bit for bit (host, once per map as in the reference, ~0.45 s per 2000^2 map)
  void updateDistancesLUT(double max_distance_to_object)
  {
    upload();
    e_->check(bpf_map2d_build_distances_lut_reference(e_->get(), max_distance_to_object));
    max_dist_ = max_distance_to_object;
    lut_.clear();
  }
  void updateDistancesLUTReference(double max_distance_to_object) { updateDistancesLUT(max_distance_to_object); }
  // NOT a reference method: the exact capped Euclidean distance transform, built on the device in milliseconds.  Its
  // values are <= the brushfire's and differ from them in < 1 % of the cells, so weights differ from the reference's
  // for particles whose beams end there -- an explicit choice of the caller (badger_pf.h, BPF_OPT_LUT_EXACT_EDT)
  void updateDistancesLUTExact(double max_distance_to_object)
  {
    upload();
    e_->check(bpf_map2d_build_distances_lut(e_->get(), max_distance_to_object));
    max_dist_ = max_distance_to_object;
    lut_.clear();
  }
  double getMaxDistanceToObject() const { return max_dist_; }
  // distances_lut_ as the engine holds it (index i + j * size_x, occupancy_map.cpp:107-110)
  std::vector<float> getDistancesLUT()
  {
    upload();
    std::vector<float> out((size_t)size_x_ * size_y_);
    e_->check(bpf_map2d_get_distances_lut(e_->get(), out.data(), out.size()));
    return out;
  }
  // OccupancyMap::calcRange (occupancy_map.cpp:257-364); cos / sin from libm here, as the reference forms them
  double calcRange(double ox, double oy, double oa, double max_range)
  {
    upload();
    const double ca = std::cos(oa), sa = std::sin(oa);
    double out = 0.0;
    e_->check(bpf_map2d_calc_range(e_->get(), &ox, &oy, &ca, &sa, &max_range, 1, &out));
    return out;
  }
  void upload()
  {
    if (!dirty_)
      return;
    e_->check(bpf_map2d_set(e_->get(), cells_.data(), lut_.empty() ? nullptr : lut_.data(), size_x_, size_y_, ox_, oy_,
                            resolution_, max_dist_));
    dirty_ = false;
  }

private:
  std::shared_ptr<Engine> e_;
  double resolution_;
  int size_x_ = 0, size_y_ = 0;
  float ox_ = 0, oy_ = 0;
  double max_dist_ = 0;
  std::vector<int32_t> cells_;
  std::vector<float> lut_;
  bool dirty_ = true;
};

struct PlanarData
{
  int range_count_ = 0;
  double range_max_ = 0;
  std::vector<double> ranges_, angles_;
};

class ParticleFilter
{
public:
  ParticleFilter(std::shared_ptr<Engine> e, int min_samples, int max_samples, double alpha_slow, double alpha_fast,
                 double global_localization_convergence_threshold)
      : e_(std::move(e)), max_samples_(max_samples)
  {
    e_->check(bpf_pf_create(e_->get(), min_samples, max_samples, alpha_slow, alpha_fast,
                            global_localization_convergence_threshold));
  }
  void setResampleModel(PFResampleModelType m) { e_->check(bpf_pf_set_resample_model(e_->get(), m)); }
  // random_pose_fn of the reference's constructor: true = Node::randomFreeSpacePose on the device (badger_pf.h)
  void setRandomFreeSpacePoseGenerator(bool on)
  {
    e_->check(bpf_pf_set_random_pose_generator(e_->get(), on ? BPF_RANDOM_POSE_FREE_SPACE_2D : BPF_RANDOM_POSE_NONE));
  }
  void setPopulationSizeParameters(double pop_err, double pop_z)
  {
    e_->check(bpf_pf_set_population_size_parameters(e_->get(), pop_err, pop_z));
  }
  void setDecayRates(double a_slow, double a_fast) { e_->check(bpf_pf_set_decay_rates(e_->get(), a_slow, a_fast)); }
  void srand48(long seed) { e_->check(bpf_pf_srand48(e_->get(), seed)); }
  // initWithPoseFn: pose_fn() returns {x, y, theta}
  template <typename PoseFn>
  void initWithPoseFn(PoseFn pose_fn)
  {
    std::vector<PFSample> s(max_samples_);
    for (auto& p : s)
    {
      p.pose = pose_fn();
      p.weight = 1.0 / max_samples_;
    }
    initWithSamples(s);
  }
  void initWithSamples(const std::vector<PFSample>& s, int leaf_count = -1)
  {
    e_->check(bpf_pf_set_samples(e_->get(), reinterpret_cast<const double*>(s.data()), (int)s.size(), leaf_count));
  }
  void updateResample() { e_->check(bpf_pf_update_resample(e_->get())); }
  std::shared_ptr<PFSampleSet> getCurrentSet()
  {
    auto set = std::make_shared<PFSampleSet>();
    set->samples.resize(max_samples_);
    int n = 0;
    e_->check(bpf_pf_get_samples(e_->get(), reinterpret_cast<double*>(set->samples.data()), max_samples_, &n));
    set->samples.resize(n);
    bpf_pf_state st;
    e_->check(bpf_pf_get_state(e_->get(), &st));
    set->sample_count = n;
    set->converged = st.converged;
    set->leaf_count = st.leaf_count;
    return set;
  }
  bool isConverged()
  {
    bpf_pf_state st;
    e_->check(bpf_pf_get_state(e_->get(), &st));
    return st.converged != 0;
  }
  bpf_pf_state getState()
  {
    bpf_pf_state st;
    e_->check(bpf_pf_get_state(e_->get(), &st));
    return st;
  }
  // ParticleFilter::getClusterStats(cidx, &weight, &mean) (particle_filter.cpp:638-649)
  bool getClusterStats(int cidx, double* weight, std::array<double, 3>* mean)
  {
    bpf_cluster c;
    const int rc = bpf_pf_get_cluster(e_->get(), cidx, &c);
    if (rc == BPF_ERR_INVALID_ARGUMENT)
      return false;
    e_->check(rc);
    *weight = c.weight;
    *mean = { c.mean[0], c.mean[1], c.mean[2] };
    return true;
  }
  // Node2D::getMaxWeightPose (node_2d.cpp:588-617)
  void getMaxWeightPose(double* max_weight_out, std::array<double, 3>* max_pose)
  {
    double pose[3] = { 0, 0, 0 };
    e_->check(bpf_pf_get_max_weight_pose(e_->get(), max_weight_out, pose));
    *max_pose = { pose[0], pose[1], pose[2] };
  }
  Engine& engine() { return *e_; }

private:
  std::shared_ptr<Engine> e_;
  int max_samples_;
};

enum OdomModelType
{
  ODOM_MODEL_DIFF = BPF_ODOM_MODEL_DIFF,
  ODOM_MODEL_OMNI = BPF_ODOM_MODEL_OMNI,
  ODOM_MODEL_DIFF_CORRECTED = BPF_ODOM_MODEL_DIFF_CORRECTED,
  ODOM_MODEL_OMNI_CORRECTED = BPF_ODOM_MODEL_OMNI_CORRECTED,
  ODOM_MODEL_GAUSSIAN = BPF_ODOM_MODEL_GAUSSIAN
};

struct OdomData
{
  std::array<double, 3> pose{}, delta{}, absolute_motion{};
};

class Odom
{
public:
  explicit Odom(std::shared_ptr<Engine> e) : e_(std::move(e)) {}
  void setModel(OdomModelType type, double alpha1, double alpha2, double alpha3, double alpha4, double alpha5 = 0)
  {
    e_->check(bpf_odom_set_model(e_->get(), type, alpha1, alpha2, alpha3, alpha4, alpha5));
  }
  bool updateAction(std::shared_ptr<ParticleFilter>, std::shared_ptr<OdomData> data)
  {
    e_->check(bpf_pf_update_action(e_->get(), data->pose.data(), data->delta.data(), data->absolute_motion.data()));
    return true;
  }

private:
  std::shared_ptr<Engine> e_;
};

class PlanarScanner
{
public:
  explicit PlanarScanner(std::shared_ptr<Engine> e) : e_(std::move(e))
  {
    e_->check(bpf_planar_set_map_factors(e_->get(), 1.0, 1.0, 0.0));  // planar_scanner.cpp:42-44
  }
  void init(int max_beams, std::shared_ptr<OccupancyMap> map)
  {
    max_beams_ = max_beams;
    map_ = std::move(map);
    map_->upload();
    e_->check(bpf_planar_init(e_->get(), max_beams));
  }
  void setModelBeam(double z_hit, double z_short, double z_max, double z_rand, double sigma_hit, double lambda_short)
  {
    map_->upload();
    e_->check(bpf_planar_set_model_beam(e_->get(), z_hit, z_short, z_max, z_rand, sigma_hit, lambda_short));
  }
  void setModelLikelihoodField(double z_hit, double z_rand, double sigma_hit, double max_distance_to_object)
  {
    map_->upload();
    e_->check(bpf_planar_set_model_likelihood_field(e_->get(), z_hit, z_rand, sigma_hit, max_distance_to_object));
  }
  void setModelLikelihoodFieldProb(double z_hit, double z_rand, double sigma_hit, double max_distance_to_object,
                                   bool do_beamskip, double beam_skip_distance, double beam_skip_threshold,
                                   double beam_skip_error_threshold)
  {
    map_->upload();
    e_->check(bpf_planar_set_model_likelihood_field_prob(e_->get(), z_hit, z_rand, sigma_hit, max_distance_to_object,
                                                         do_beamskip, beam_skip_distance, beam_skip_threshold,
                                                         beam_skip_error_threshold));
  }
  void setModelLikelihoodFieldGompertz(double z_hit, double z_rand, double sigma_hit, double max_distance_to_object,
                                       double gompertz_a, double gompertz_b, double gompertz_c, double input_shift,
                                       double input_scale, double output_shift)
  {
    map_->upload();
    e_->check(bpf_planar_set_model_likelihood_field_gompertz(e_->get(), z_hit, z_rand, sigma_hit,
                                                             max_distance_to_object, gompertz_a, gompertz_b,
                                                             gompertz_c, input_shift, input_scale, output_shift));
  }
  void setMapFactors(double off_map_factor, double non_free_space_factor, double non_free_space_radius)
  {
    e_->check(bpf_planar_set_map_factors(e_->get(), off_map_factor, non_free_space_factor, non_free_space_radius));
  }
  void setPlanarScannerPose(const std::array<double, 3>& pose) { e_->check(bpf_planar_set_scanner_pose(e_->get(), pose.data())); }

  // PlanarScanner::updateSensor(pf, data): false and no effect when max_beams_ < 2
  bool updateSensor(std::shared_ptr<ParticleFilter> pf, std::shared_ptr<PlanarData> data)
  {
    if (max_beams_ < 2)
      return false;
    e_->check(bpf_pf_update_sensor_planar(e_->get(), data->ranges_.data(), data->angles_.data(), data->range_count_,
                                          data->range_max_));
    (void)pf;
    return true;
  }
  // PlanarScanner::applyModelToSampleSet(data, set) on a host-resident set
  double applyModelToSampleSet(std::shared_ptr<PlanarData> data, std::shared_ptr<PFSampleSet> set)
  {
    int status = BPF_OK;
    const double total = bpf_planar_apply_model_to_sample_set(
        e_->get(), reinterpret_cast<double*>(set->samples.data()), set->sample_count, set->converged,
        data->ranges_.data(), data->angles_.data(), data->range_count_, data->range_max_, &status);
    e_->check(status);
    return total;
  }

private:
  std::shared_ptr<Engine> e_;
  std::shared_ptr<OccupancyMap> map_;
  int max_beams_ = 0;
};

}  // namespace badger_amcl_amd
