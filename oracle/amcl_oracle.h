/*
 * amcl_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, single-threaded, zero dependencies) of the
 * badger_amcl hot path: planar / point-cloud sensor scoring, weight
 * normalisation and KLD-adaptive resampling.  It exists so that the HIP path
 * can be checked against something that follows the reference's arithmetic
 * operation for operation.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; nothing in badger_amcl_amd/ does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   pinned by the reference's own gtest known answers
 *     (test/test_badger_amcl.cpp): drand48 + Gaussian draw (:29-49), kd-tree
 *     leaf count / cluster labels (:51-82), OctoMap and OccupancyMap
 *     world<->cell conversions (:84-129), isValid and calcRange (:131-171).
 *   PARITY UNPINNED (no reference test or fixture covers them, and the
 *     reference cannot be built here without stand-in headers): the four planar
 *     sensor models, recalcWeight, updateSensor normalisation, both resamplers,
 *     cluster statistics, updateConverged, the distance-LUT brushfire and the
 *     3-D point-cloud models.  They are restated line by line from the cited
 *     reference source and cross-checked only structurally.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 */
#ifndef AMCL_ORACLE_H
#define AMCL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG: glibc drand48 family (libc, used at particle_filter.cpp:309,385,393) */
void orc_srand48(uint64_t* state, long seed);
double orc_drand48(uint64_t* state);
/* pdf_gaussian.cpp:77-97 */
double orc_gaussian_draw(uint64_t* state, double sigma);
/* angles::normalize_angle (third-party `angles` package, Noetic form) */
double orc_normalize_angle(double a);
/* Odom::updateAction (src/amcl/sensors/odom.cpp:74-301); model = OdomModelType (odom.h:33-40):
 * 0 diff, 1 omni, 2 diff-corrected, 3 omni-corrected, 4 gaussian.  samples = [n][4]. */
void orc_odom_update_action(int model, const double alpha[5], const double pose[3], const double delta[3],
                            const double absolute_motion[3], double* samples, int n, uint64_t* rng);

/* ---- 2-D occupancy map (include/amcl/map/occupancy_map.h:93-102, map.h:48-53) */
typedef struct
{
  int size_x, size_y;
  float origin_x, origin_y; /* pcl::PointXYZ is float (map.h:48) */
  double resolution;
  double max_dist;          /* max_distance_to_object_ */
  const int32_t* cells;     /* -1 free, 0 unknown, +1 occupied; index i + j*size_x */
  const float* lut;         /* distances_lut_, same indexing */
} orc_map2d;

void orc_map2d_world_to_map(const orc_map2d* m, double x, double y, int* i, int* j);
void orc_map2d_map_to_world(const orc_map2d* m, int i, int j, double* x, double* y);
int orc_map2d_is_valid(const orc_map2d* m, int i, int j);
float orc_map2d_distance(const orc_map2d* m, int i, int j);
double orc_map2d_calc_range(const orc_map2d* m, double ox, double oy, double oa, double max_range,
                            long* cells_visited);
/* occupancy_map.cpp:122-252; lut_out has size_x*size_y floats */
void orc_map2d_build_lut(int size_x, int size_y, const int32_t* cells, double resolution,
                         double max_dist, float* lut_out);

/* ---- planar scanner (src/amcl/sensors/planar_scanner.cpp) */
enum
{
  ORC_MODEL_BEAM = 0,
  ORC_MODEL_LIKELIHOOD_FIELD = 1,
  ORC_MODEL_LIKELIHOOD_FIELD_PROB = 2,
  ORC_MODEL_LIKELIHOOD_FIELD_GOMPERTZ = 3
};

typedef struct
{
  int model;
  int max_beams;
  double z_hit, z_short, z_max, z_rand, sigma_hit, lambda_short;
  double gompertz_a, gompertz_b, gompertz_c, input_shift, input_scale, output_shift;
  int do_beamskip;
  double beam_skip_distance, beam_skip_threshold, beam_skip_error_threshold;
  double off_map_factor, non_free_space_factor, non_free_space_radius;
  double scanner_pose[3];
} orc_planar;

void orc_planar_defaults(orc_planar* p);

/* samples: AoS {x, y, theta, weight} doubles (PFSample, particle_filter.h:41-49).
 * Returns the total (planar_scanner.cpp:141-164).  stats[0] += beam evaluations,
 * stats[1] += cells visited by calcRange (beam model), either may be NULL. */
double orc_planar_apply(const orc_planar* p, const orc_map2d* m, double* samples, int sample_count,
                        int set_converged, const double* ranges, const double* angles,
                        int range_count, double range_max, long* stats);

/* ---- kd-tree histogram (src/amcl/pf/pf_kdtree.cpp) */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree* orc_kdtree_new(void);
void orc_kdtree_free(orc_kdtree* t);
void orc_kdtree_clear(orc_kdtree* t);
void orc_kdtree_insert(orc_kdtree* t, const double pose[3], double value);
void orc_kdtree_insert_key(orc_kdtree* t, const int key[3], double value);
int orc_kdtree_leaf_count(const orc_kdtree* t);
int orc_kdtree_node_count(const orc_kdtree* t);
void orc_kdtree_cluster(orc_kdtree* t);
int orc_kdtree_get_cluster(const orc_kdtree* t, const double pose[3]);

/* ---- particle filter core (src/amcl/pf/particle_filter.cpp) */
enum
{
  ORC_RESAMPLE_MULTINOMIAL = 0,
  ORC_RESAMPLE_SYSTEMATIC = 1
};

typedef struct
{
  int min_samples, max_samples;
  double pop_err, pop_z;
  double alpha_slow, alpha_fast;
  double w_slow, w_fast;
  double dist_threshold;
  double convergence_threshold; /* global_localization_convergence_threshold_ */
  int resample_model;
  uint64_t rng;                 /* the process-global drand48 state */
  int converged;
  /* random_pose_fn_ (particle_filter.cpp:40-47): NULL = not available (w_diff > 0 reports status 2); otherwise
   * Node::uniformPoseGenerator with its score check disabled (uniform_pose_starting_weight_threshold = 0, the
   * node's default, node.cpp:124) = Node::randomFreeSpacePose over this free-cell list (node.cpp:823-845) */
  const struct orc_free_space* random_source;
} orc_pf;

/* Node2D::updateFreeSpaceIndices (node_2d.cpp:317-337): FREE cells further than non_free_space_radius from an
 * obstacle, i outer / j inner, plus what convertMapToWorld needs */
typedef struct orc_free_space
{
  int n;
  const int* ij; /* [n][2] */
  int size_x, size_y;
  double origin_x, origin_y, resolution;
} orc_free_space;
/* ParticleFilter::initWithPoseFn (particle_filter.cpp:135-163) with pose_fn = Node::randomFreeSpacePose, and
 * ParticleFilter::initWithGaussian (:105-132) with PDFGaussian::sample (pdf_gaussian.cpp:52-70) given the
 * decomposition the constructor obtains from Eigen::EigenSolver (third party): cr = rotation (row-major 3x3),
 * cd = sqrt of the eigenvalues.  samples = [n][4] (n = max_samples); returns the kd-tree leaf count, node count
 * in *node_count. */
int orc_pf_init_with_free_space_poses(orc_pf* pf, const orc_free_space* fs, double* samples, int n, int* node_count);
int orc_pf_init_with_gaussian(orc_pf* pf, const double mean[3], const double cr[9], const double cd[3],
                              double* samples, int n, int* node_count);
/* returns the number of free cells; ij_out (capacity pairs) may be NULL to count only */
int orc_free_space_indices(const orc_map2d* m, double non_free_space_radius, int* ij_out, int capacity);
/* Node::randomFreeSpacePose: two drand48 draws */
void orc_random_free_space_pose(const orc_free_space* fs, uint64_t* rng, double pose[3]);

void orc_pf_init(orc_pf* pf, int min_samples, int max_samples, double alpha_slow, double alpha_fast,
                 double convergence_threshold);
/* particle_filter.cpp:237-266 given the sensor function's total */
void orc_pf_normalize(orc_pf* pf, double* samples, int sample_count, double total);
int orc_pf_resample_limit(const orc_pf* pf, int k);

typedef struct
{
  int sample_count;    /* M */
  int leaf_count;      /* set_b kdtree leaf count after the resample */
  int node_count;      /* distinct bins */
  int cluster_count;
  int converged;
  double mean[3];
  double cov[4];       /* (0,0) (0,1) (1,0) (1,1) */
  double cov_theta;    /* (2,2) */
  double w_diff;
  int status;          /* 0 ok; 1 CDF search miss (reference ROS_ASSERT) ; 2 unsupported w_diff>0 */
  float percent_converged;
} orc_resample_out;

/* particle_filter.cpp:423-471.  set_a: N x 4 doubles AoS; set_b: max_samples x 4.
 * prev_leaf_count = set_a's kdtree leaf count (systematic only).
 * idx_out (nullable): source index per output sample.  w_diff>0 needs a random
 * pose callback; the oracle takes none and reports status 2. */
void orc_pf_update_resample(orc_pf* pf, const double* set_a, int n_a, int prev_leaf_count,
                            double* set_b, int* idx_out, orc_resample_out* out);

/* particle_filter.cpp:170-220; returns converged flag, percent via *percent */
int orc_pf_update_converged(const orc_pf* pf, const double* samples, int n, float* percent);

/* particle_filter.cpp:505-636.  cluster arrays sized max_clusters:
 * count[], weight[], mean[3*], cov[5*] ((0,0),(0,1),(1,0),(1,1),(2,2)). */
int orc_pf_cluster_stats(orc_kdtree* t, const double* samples, int n, int max_clusters,
                         int* c_count, double* c_weight, double* c_mean, double* c_cov,
                         double set_mean[3], double set_cov[5]);

/* ---- 3-D map + point cloud scanner (octomap.cpp, point_cloud_scanner.cpp) */
typedef struct
{
  int min_cells[3], max_cells[3]; /* cropped_min_cells_, cropped_max_cells_ */
  int width;                      /* map_cells_width_ */
  int num_z;                      /* num_z_column_indices_ */
  double resolution, max_dist;
  const uint32_t* pose_indices;   /* per (x,y) column start into distance_ratios; 0 = shared empty */
  const uint8_t* distance_ratios;
} orc_map3d;

void orc_map3d_world_to_map(const orc_map3d* m, const double w[3], int c[3]);
void orc_map3d_map_to_world(double resolution, const int c[3], double w[3]);
double orc_map3d_distance(const orc_map3d* m, int i, int j, int k);
/* octomap.cpp:152-333 given the occupied voxel list (the octree walk itself is
 * third-party).  Returns number of uint8 entries written to ratios_out
 * (capacity ratios_cap); pose_indices_out has num_poses entries. */
size_t orc_map3d_build_lut(const int min_cells[3], const int max_cells[3], double resolution,
                           double max_dist, const int* occupied_ijk, size_t n_occupied,
                           uint32_t* pose_indices_out, uint8_t* ratios_out, size_t ratios_cap);

enum
{
  ORC_CLOUD_MODEL = 0,
  ORC_CLOUD_MODEL_GOMPERTZ = 1
};

typedef struct
{
  int model;
  int max_beams;
  double z_hit, z_rand, sigma_hit;
  double gompertz_a, gompertz_b, gompertz_c, input_shift, input_scale, output_shift;
  double off_map_factor;
  /* scanner -> footprint transform: translation + quaternion (x,y,z,w) */
  double tf_xyz[3];
  double tf_quat[4];
} orc_cloud;

/* points: n x 3 float32 in the scanner frame */
double orc_cloud_apply(const orc_cloud* p, const orc_map3d* m, double* samples, int sample_count,
                       const float* points, int n_points, long* stats);

/* ---- ROS-facing message shaping either side of the path ---- */
/* Node2D::updateLatestScanData, node_2d.cpp:531-560 */
void orc_wire_laserscan_to_planar(const float* scan_ranges, int range_count, float scan_range_min, float scan_range_max,
                                  double sensor_min_range, double sensor_max_range, double angle_min,
                                  double angle_increment, double* ranges_out, double* angles_out, double* range_max_out);
/* Node2D::getAngleStats, node_2d.cpp:497-529, given the rotation of the base <- scanner transform (x, y, z, w) */
void orc_wire_scan_angle_stats(double scan_angle_min, double scan_angle_increment, const double q_base_scanner[4],
                               double* angle_min_out, double* angle_increment_out);
/* Node2D::convertMap, node_2d.cpp:265-295: cells (MapCellState as int), size, float origin, resolution */
void orc_wire_convert_map(const int8_t* data, int width, int height, double msg_resolution, double origin_x,
                          double origin_y, int map_scale_up_factor, int32_t* cells_out, int size_out[2],
                          float origin_out[2], double* resolution_out);
/* Node3D::updateLatestScanData, node_3d.cpp:467-480: returns the number of points kept */
int orc_wire_decimate_cloud(const float* points_xyz, int data_count, int max_beams, float* out_xyz);
/* Node::publishParticleCloud, node.cpp:335-357: position (x, y, 0) + quaternion (x, y, z, w) per sample */
void orc_wire_pose_array(const double* samples, int sample_count, double* poses7_out);

#ifdef __cplusplus
}
#endif
#endif
