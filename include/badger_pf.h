/*
 * badger_pf.h -- C-ABI of libbadger_pf_hip.so, the MI355X (gfx950) engine for the
 * badger_amcl sensor-update + resample hot path.
 *
 * Everything here is plain C: opaque handle, pointers and sizes, int status codes.
 * No exception, C++ type or torch type crosses this boundary.  One engine is used
 * by one host thread at a time (the reference's seams are single-caller too:
 * SURVEY.md section 8(b), "Threading").
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * the reference checkout).  INTEGRATION.md shows the C++ binding a maintainer of
 * the reference would add on top of this header.
 *
 * Memory conventions
 *   "samples" buffers are the reference's PFSample array seen as doubles: AoS
 *   {x, y, theta, weight}, 32 bytes per particle (include/amcl/pf/particle_filter.h:41-49).
 *   Host pointers are caller-owned and only read/written during the call.
 *   Functions whose name contains `_dev` take DEVICE pointers the caller owns
 *   (e.g. a torch tensor's data_ptr()) and run asynchronously on the engine stream.
 */
#ifndef BADGER_PF_H
#define BADGER_PF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bpf_engine bpf_engine;

/* ---- status codes (the reference has none: it returns 0.0 / false or ROS_ASSERTs;
 * see SURVEY.md 8(b) "Error conventions") */
enum
{
  BPF_OK = 0,
  BPF_ERR_INVALID_ARGUMENT = 1,
  BPF_ERR_NOT_CONFIGURED = 2,     /* map / model / filter missing */
  BPF_ERR_HIP = 3,                /* a HIP runtime call failed; see bpf_last_error_message */
  BPF_ERR_UNSUPPORTED = 4,        /* e.g. w_diff > 0 needs the node's random_pose_fn_ callback */
  BPF_ERR_CDF_MISS = 5,           /* reference ROS_ASSERT(i < sample_count), particle_filter.cpp:399 */
  BPF_ERR_LUT_LEVELS = 6,         /* distance LUT holds more than 8190 distinct values */
  BPF_ERR_BEAM_STEP = 7,          /* beam model with range_count < max_beams: the reference never returns */
  BPF_ERR_CAPACITY = 8,
  BPF_ERR_EXCHANGE = 9            /* mailbox exchange: a peer's word did not arrive within 5 s */
};

enum
{
  BPF_MODEL_BEAM = 0,                       /* PLANAR_MODEL_BEAM */
  BPF_MODEL_LIKELIHOOD_FIELD = 1,           /* PLANAR_MODEL_LIKELIHOOD_FIELD */
  BPF_MODEL_LIKELIHOOD_FIELD_PROB = 2,      /* PLANAR_MODEL_LIKELIHOOD_FIELD_PROB */
  BPF_MODEL_LIKELIHOOD_FIELD_GOMPERTZ = 3   /* PLANAR_MODEL_LIKELIHOOD_FIELD_GOMPERTZ */
};

enum
{
  BPF_RESAMPLE_MULTINOMIAL = 0, /* PF_RESAMPLE_MULTINOMIAL */
  BPF_RESAMPLE_SYSTEMATIC = 1   /* PF_RESAMPLE_SYSTEMATIC */
};

enum
{
  BPF_CLOUD_MODEL = 0,          /* POINT_CLOUD_MODEL */
  BPF_CLOUD_MODEL_GOMPERTZ = 1  /* POINT_CLOUD_MODEL_GOMPERTZ */
};

/* ------------------------------------------------------------------ lifecycle */
int bpf_create(int device_ordinal, bpf_engine** out);
void bpf_destroy(bpf_engine* e);
const char* bpf_error_string(int code);
const char* bpf_last_error_message(const bpf_engine* e);
/* Run the engine's work on a caller-provided hipStream_t.  NULL is HIP's default (null) stream,
 * which is what torch.cuda.current_stream().cuda_stream reports unless a side stream is active;
 * BPF_OWN_STREAM restores the engine's own non-blocking stream. */
#define BPF_OWN_STREAM ((void*)(intptr_t)-1)
int bpf_set_stream(bpf_engine* e, void* hip_stream);
int bpf_synchronize(bpf_engine* e);

/* ------------------------------------------------------------------ 2-D map
 * OccupancyMap state (include/amcl/map/occupancy_map.h:93-102, map.h:48-53):
 * cells = cells_.data() (MapCellState is a 32-bit enum: -1 free, 0 unknown, +1
 * occupied; index i + j*size_x), dist_lut = distances_lut_.data() (may be NULL),
 * origin = origin_.x/.y (float), resolution_, max_distance_to_object_. */
int bpf_map2d_set(bpf_engine* e, const int32_t* cells, const float* dist_lut, int size_x, int size_y,
                  float origin_x, float origin_y, double resolution, double max_dist);
/* The FAST, explicitly named alternative to OccupancyMap::updateDistancesLUT (occupancy_map.cpp:138-160): the exact
 * Euclidean distance capped at max_dist, on the reference's (a,b)-integer lattice, built on the device in
 * milliseconds.  NOT the reference's values: its brushfire is approximate (>= this, equal in > 99 % of the cells).
 * The call that gives the reference's values is bpf_map2d_build_distances_lut_reference below, and that is what the
 * host mirrors' `updateDistancesLUT` and the implicit build of bpf_planar_set_model_likelihood_field* use unless
 * BPF_OPT_LUT_EXACT_EDT is set. */
int bpf_map2d_build_distances_lut(bpf_engine* e, double max_dist);
int bpf_map2d_get_distances_lut(bpf_engine* e, float* out, size_t capacity);
/* OccupancyMap::calcRange(ox, oy, oa, max_range) (occupancy_map.cpp:257-364) for n rays: the integer Bresenham
 * walk from the cell of (ox, oy) towards the cell of the max-range end point, distance to the first cell that is off
 * the map or not FREE (max_range if none; 0 for a start off the map).  The direction comes as cos(oa), sin(oa),
 * formed by the caller as the reference forms them (libm), so that the end cell is the reference's bit for bit.
 * Needs only the cells (bpf_map2d_set); host buffers in and out. */
int bpf_map2d_calc_range(bpf_engine* e, const double* ox, const double* oy, const double* cos_a, const double* sin_a,
                         const double* max_range, int n, double* range_out);
/* OccupancyMap::updateDistancesLUT exactly as the reference builds it (occupancy_map.cpp:138-252):
 * priority-queue brushfire on the host (std::priority_queue, so tie order matches a libstdc++
 * build of the reference), ~0.45 s for a 2000 x 2000 map, once per map as in the reference.  THE DEFAULT behind the
 * reference-named calls. */
int bpf_map2d_build_distances_lut_reference(bpf_engine* e, double max_dist);

/* ------------------------------------------------------------------ caller-owned host buffers
 * The reference keeps its particles in host memory: ParticleFilter allocates `samples` once at max_samples
 * (src/amcl/pf/particle_filter.cpp:62-89, include/amcl/pf/particle_filter.h:70-75) and every sensor model gets that
 * vector (planar_scanner.cpp:141-164).  Registering the buffer (hipHostRegister: pins it and maps it for the copy
 * engines, ~1 ms, once) lets the host-buffer entry points below move it at PCIe rate without a staging copy by the
 * calling thread.  [ptr, ptr + bytes) need not be aligned.  The owner must keep the memory allocated until
 * bpf_host_buffer_unregister (same ptr) or bpf_destroy.  Unregistered buffers work everywhere, through the runtime's
 * bounce buffers.  bpf_host_buffer_is_registered: 1 when the whole range lies inside one registered buffer. */
int bpf_host_buffer_register(bpf_engine* e, void* ptr, size_t bytes);
int bpf_host_buffer_unregister(bpf_engine* e, void* ptr);
int bpf_host_buffer_is_registered(bpf_engine* e, const void* ptr, size_t bytes);
/* 3 when the last likelihood-field scoring launch of a resident set walked the particles in map-tile order
 * (BPF_OPT_TILE_SORT), 0 otherwise */
int bpf_score_last_form(bpf_engine* e, int* form_out);
/* which form the last device-side histogram tree took: 2 = grown in LDS-sized pieces (kernels_kld2.hpp), 1 = one launch
 * pair per level (also after the pieces declined a stream), 3 = the persistent launch, 0 = none yet */
int bpf_kld_last_form(bpf_engine* e, int* form_out);
/* what the last bpf_planar_apply_model_to_sample_set did: chunks of the pipelined form (0 = the plain upload / score /
 * download sequence, -1 = one scoring launch that read and wrote the registered records in place); pinned = the buffer
 * lies in a registered range */
int bpf_seam_last_plan(bpf_engine* e, int* chunks_out, int* pinned_out);

/* ------------------------------------------------------------------ planar scanner
 * PlanarScanner::{init, setModel*, setMapFactors, setPlanarScannerPose}
 * (include/amcl/sensors/planar_scanner.h:62-93, planar_scanner.cpp:49-121,535-538). */
int bpf_planar_init(bpf_engine* e, int max_beams);
int bpf_planar_set_model_beam(bpf_engine* e, double z_hit, double z_short, double z_max, double z_rand,
                              double sigma_hit, double lambda_short);
int bpf_planar_set_model_likelihood_field(bpf_engine* e, double z_hit, double z_rand, double sigma_hit,
                                          double max_distance_to_object);
int bpf_planar_set_model_likelihood_field_prob(bpf_engine* e, double z_hit, double z_rand, double sigma_hit,
                                               double max_distance_to_object, int do_beamskip,
                                               double beam_skip_distance, double beam_skip_threshold,
                                               double beam_skip_error_threshold);
int bpf_planar_set_model_likelihood_field_gompertz(bpf_engine* e, double z_hit, double z_rand, double sigma_hit,
                                                   double max_distance_to_object, double gompertz_a,
                                                   double gompertz_b, double gompertz_c, double input_shift,
                                                   double input_scale, double output_shift);
int bpf_planar_set_map_factors(bpf_engine* e, double off_map_factor, double non_free_space_factor,
                               double non_free_space_radius);
int bpf_planar_set_scanner_pose(bpf_engine* e, const double pose[3]);

/* Seam A, host buffers: PlanarScanner::applyModelToSampleSet (planar_scanner.cpp:141-164).
 * Multiplies samples[i].weight in place for i < sample_count and returns their sum
 * (0.0 on failure, like the reference); *status (nullable) receives a BPF_* code.
 * ranges/angles/range_count/range_max are PlanarData (planar_scanner.h:45-54);
 * set_converged is PFSampleSet::converged (only the prob model reads it).
 * A registered, 16-byte aligned buffer (bpf_host_buffer_register) of 4 096 records or more under a likelihood-field
 * model is not copied at all: one scoring launch reads the records over PCIe as its waves reach them and writes them
 * back whole with the new weight.  Other sets of 40 k particles or more go through in two chunks
 * (BPF_OPT_SEAM_CHUNKS): chunk k is scored while chunk k + 1 crosses PCIe, the scoring launches store the weights into
 * pinned host memory themselves (no download) and the calling thread writes chunk k's weights into the records while
 * chunk k + 1 is scored.  Same weights bit for bit as the plain upload / score / download sequence. */
double bpf_planar_apply_model_to_sample_set(bpf_engine* e, double* samples, int sample_count, int set_converged,
                                            const double* ranges, const double* angles, int range_count,
                                            double range_max, int* status);

/* ------------------------------------------------------------------ particle filter
 * Device-resident ParticleFilter (include/amcl/pf/particle_filter.h:92-184).  The two
 * ping-pong sample sets live in HBM as structure-of-arrays. */
int bpf_pf_create(bpf_engine* e, int min_samples, int max_samples, double alpha_slow, double alpha_fast,
                  double global_localization_convergence_threshold);  /* ctor, particle_filter.cpp:38-98 */
int bpf_pf_set_resample_model(bpf_engine* e, int resample_model);             /* :100-103 */
int bpf_pf_set_population_size_parameters(bpf_engine* e, double pop_err, double pop_z); /* :651-655 */
int bpf_pf_set_decay_rates(bpf_engine* e, double alpha_slow, double alpha_fast);        /* :657-661 */
/* The reference draws from the process-global drand48 stream (particle_filter.cpp:309,385,393);
 * the engine carries that 48-bit state explicitly so host code can hand it over and take it back. */
int bpf_pf_srand48(bpf_engine* e, long seed);
int bpf_pf_set_rng_state(bpf_engine* e, uint64_t state48);
int bpf_pf_get_rng_state(const bpf_engine* e, uint64_t* state48);
/* Load the current set (what initWithPoseFn/initWithGaussian leave behind, :106-162):
 * resets w_slow/w_fast and converged.  leaf_count = that set's kd-tree leaf count, or -1
 * to have it computed from the poses -- when something first needs it (bpf_pf_get_state, the systematic resampler,
 * a snapshot) or before the poses move (a motion update): the tree is that of the poses handed over here, as in the
 * reference, but a cycle that uploads, updates and resamples with the multinomial resampler never builds it.
 * A registered `samples` buffer (bpf_host_buffer_register) is read by the copy engine directly. */
int bpf_pf_set_samples(bpf_engine* e, const double* samples, int sample_count, int leaf_count);
int bpf_pf_get_samples(bpf_engine* e, double* samples_out, int capacity, int* sample_count_out);
/* Keep a device-resident copy of the current set (poses, weights, counts) and put it back
 * later with one device-to-device copy -- stands in for the motion update, which rewrites
 * every pose of the set each cycle (Odom::updateAction, out of scope here). */
int bpf_pf_snapshot(bpf_engine* e);
int bpf_pf_restore(bpf_engine* e);
/* Overwrite the weights of the current set with one value (bench harness: restores 1/N). */
int bpf_pf_fill_weights(bpf_engine* e, double weight);

/* Seam A on the resident set: PlanarScanner::updateSensor(pf, data)
 * = ParticleFilter::updateSensor(applyModelToSampleSet, data)
 * (planar_scanner.cpp:125-137, particle_filter.cpp:223-267).  Asynchronous. */
int bpf_pf_update_sensor_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                double range_max);
/* random_pose_fn_ of the ParticleFilter constructor (particle_filter.cpp:40-47), used by both resamplers when
 * w_diff = max(0, 1 - w_fast / w_slow) > 0 (augmented-MCL recovery, :295-324 and :383-388).  The node passes
 * Node::uniformPoseGenerator; with its score check disabled (uniform_pose_starting_weight_threshold = 0, the
 * node's default, node.cpp:124,847-868) that is Node::randomFreeSpacePose (node.cpp:823-845): two drand48 draws,
 * a uniformly chosen cell of Node2D::updateFreeSpaceIndices (node_2d.cpp:317-337: FREE and further than
 * non_free_space_radius from an obstacle) and a uniform heading.  BPF_RANDOM_POSE_FREE_SPACE_2D evaluates exactly
 * that on the device from the engine's own 2-D map, drawing from the filter's drand48 stream in the reference's
 * order; with BPF_RANDOM_POSE_NONE (default) a resample that needs random poses returns BPF_ERR_UNSUPPORTED. */
enum
{
  BPF_RANDOM_POSE_NONE = 0,
  BPF_RANDOM_POSE_FREE_SPACE_2D = 1
};
int bpf_pf_set_random_pose_generator(bpf_engine* e, int mode);
/* Seam B: ParticleFilter::updateResample (particle_filter.cpp:423-471). */
int bpf_pf_update_resample(bpf_engine* e);

/* Engine options.  BPF_OPT_CDF_SERIAL = 1 builds the resampling CDF with the reference's
 * serial running sum (one lane, bit-exact, slow) instead of the parallel scan;
 * BPF_OPT_COUNT_CELLS = 1 makes the beam-model kernel count the cells its rays visit. */
enum
{
  BPF_OPT_CDF_SERIAL = 0,
  BPF_OPT_COUNT_CELLS = 1,
  BPF_OPT_WINDOW_PATH = 2,  /* default 0: 1 allows the LDS-window scoring kernels (device-side switch) */
  BPF_OPT_KLD_DEVICE_MIN = 3, /* default 8192: candidate draws left after the first window from which the KLD stop
                               * rule (ordered kd-tree replay) runs on the device instead of the host; 0 = never */
  BPF_OPT_GRADED_SHARES = 4,  /* default 1: the scoring kernel's waves own particle shares graded by the placement
                               * round of their block; 0 = equal shares.  Results do not
                               * depend on it beyond the summation order of the weight total. */
  BPF_OPT_CLOUD_DENSE = 6,    /* default 1: the 3-D scoring kernel gathers from a dense tiled copy of the LUT when the
                               * map allows one (a z plane below 16 MiB, the volume below 1 GiB); 0 = the reference's
                               * two-level layout.  Same results. */
  BPF_OPT_STATS_HOST = 7,     /* default 0: cluster statistics on the device (order-independent fixed-point sums: equal to
                               * the reference's up to summation rounding); 1 = on the host from a copy of the set, in
                               * the reference's serial order, bit for bit */
  BPF_OPT_LUT_HOST = 8,       /* default 0: bpf_map3d_build_distances_lut replays the reference's FIFO brushfire on the device,
                               * generation by generation (same bytes, same column order); 1 = the serial host builder */
  BPF_OPT_KLD_PERSISTENT = 9, /* default 0: one launch pair per level of the device-side histogram tree of a long draw
                               * stream; 1: ONE launch with grid barriers between the levels when the stream fits one
                               * resident round of blocks (measured slower: 0.98 against 0.83 ms per step of the spread
                               * cloud -- every level is ~9 dependent round trips through the Infinity Cache either way) */
  BPF_OPT_LUT_EXACT_EDT = 10, /* default 0: the implicit LUT build of bpf_planar_set_model_likelihood_field* (the reference calls
                               * map_->updateDistancesLUT there, planar_scanner.cpp:74,91,112) is the reference's
                               * brushfire on the host; 1 = the exact EDT on the device (milliseconds, values differ from
                               * the reference's in < 1 % of the cells) */
  BPF_OPT_HOST_AUTO_REGISTER = 11, /* default 0.  1 = the CALLER'S PROMISE that every host buffer of 64 KB or more it hands to
                               * bpf_planar_apply_model_to_sample_set / bpf_pf_set_samples / bpf_pf_get_samples stays
                               * allocated until bpf_destroy: the engine then pins each one on first sight
                               * (hipHostRegister, ~1 ms once) and keeps the registration, keyed by address range.  A
                               * buffer freed or re-allocated under a kept registration makes the next copy fault --
                               * hence off by default; bpf_host_buffer_register is the per-buffer, owner-controlled form */
  BPF_OPT_SEAM_CHUNKS = 12,   /* default 0 = by size and buffer: a registered, 16-byte aligned buffer of 4 096 records or
                               * more is read and written IN PLACE by one scoring launch (no copy); otherwise sets of 40 k
                               * particles or more go up in two chunks; k > 1 forces k chunks (at most 8) of the pipelined
                               * host-buffer seam (bpf_planar_apply_model_to_sample_set); 1 = the plain upload / score /
                               * download sequence.  Same weights either way (bpf_seam_last_plan says which ran). */
  BPF_OPT_KLD_LOCAL = 13,     /* default 1: the device-side histogram tree of a long draw stream is grown in LDS-sized pieces
                               * (one block grows the top from the first 2 048 keys, the later keys are routed through it
                               * and blocks grow the subtrees below its nodes: kernels_kld2.hpp) instead of one launch
                               * pair per level; 0 = the level loop.  Same tree. */
  BPF_OPT_TILE_SORT = 14,     /* default 1: a cloud the previous resample found spread (no KLD stop) or that was just drawn
                               * uniformly is scored in map-tile order, each XCD taking a contiguous eighth of it, when the
                               * map's LUT does not fit an XCD's L2 (tile-sorted scoring); 0 = index
                               * order always.  Same weights. */
  BPF_OPT_HOST_DIRECT_PAGEABLE = 15, /* default 0: pageable host memory (unregistered sample buffers, map and cloud arrays)
                               * goes up through the engine's own pinned bounce buffer, one host copy.  1 = it is handed
                               * to the HIP runtime as it is (faster by that copy), which pins ranges of more than a
                               * megabyte on the fly and KEEPS such pins, keyed by address and size: the caller's
                               * promise that no buffer it passes is ever freed and allocated again at the same address
                               * while the engine lives (a stale pin reads old pages or faults).  Registered buffers
                               * (bpf_host_buffer_register) are always direct. */
  BPF_OPT_FUSED_RESAMPLE = 5  /* default 1: normalisation + CDF in one launch, and a resample whose candidate stream
                               * fits 4096 draws as one single-block launch (draws, KLD stop rule, weights,
                               * updateConverged); 0 = the separate launches with the host's ordered replay.
                               * Same results either way. */
};
int bpf_set_option(bpf_engine* e, int option, int value);
/* cells visited by calcRange walks since the last reset (BPF_OPT_COUNT_CELLS) */
int bpf_get_cells_walked(bpf_engine* e, unsigned long long* out, int reset);

typedef struct
{
  int sample_count;       /* PFSampleSet::sample_count of the current set */
  int leaf_count;         /* its kd-tree leaf count (PFKDTree::getLeafCount) */
  int bin_count;          /* distinct occupied histogram bins (kd-tree node count) */
  int converged;          /* ParticleFilter::isConverged */
  float percent_converged;
  double total;           /* last sensor_fn total (particle_filter.cpp:235) */
  double w_slow, w_fast;
  double w_diff;          /* of the last updateResample */
  int last_status;        /* BPF_* of the last update_sensor / update_resample */
  int resample_windows;   /* candidate-draw windows used by the last multinomial resample */
  long long evals;        /* particle-beam evaluations of the last sensor update */
  int kld_on_device;      /* where the last resample's histogram tree was grown: 0 host (ordered replay), 1 device,
                           * level-synchronous in HBM (long streams), 2 device, inside the one-block resample kernel */
  int reserved;
} bpf_pf_state;
int bpf_pf_get_state(bpf_engine* e, bpf_pf_state* out);

/* ------------------------------------------------------------------ motion update
 * Odom::setModel / Odom::updateAction (src/amcl/sensors/odom.cpp:63-301, caller node.cpp:1053-1091)
 * on the resident set: 3 PDFGaussian::draw (pdf_gaussian.cpp:77-97) per particle from the filter's
 * drand48 stream, consumed in the reference's order (particle by particle, rejected attempts and
 * r == 0 re-draws included), so the state left for the resampler is exact.  Poses agree with a
 * libm build of the reference to a few ulp (device log / sin / cos). */
enum
{
  BPF_ODOM_MODEL_DIFF = 0,            /* OdomModelType, include/amcl/sensors/odom.h:33-40 */
  BPF_ODOM_MODEL_OMNI = 1,
  BPF_ODOM_MODEL_DIFF_CORRECTED = 2,
  BPF_ODOM_MODEL_OMNI_CORRECTED = 3,
  BPF_ODOM_MODEL_GAUSSIAN = 4
};
int bpf_odom_set_model(bpf_engine* e, int model_type, double alpha1, double alpha2, double alpha3,
                       double alpha4, double alpha5);
/* OdomData: pose, delta, absolute_motion (odom.h:43-52) */
int bpf_pf_update_action(bpf_engine* e, const double pose[3], const double delta[3],
                         const double absolute_motion[3]);
/* the same for one shard of a filter spread over several engines: this engine holds particles
 * [global_first, global_first + local count) of global_count; every rank passes the same rng state
 * (bpf_pf_set_rng_state) and ends with the same one. */
int bpf_shard_update_action(bpf_engine* e, const double pose[3], const double delta[3],
                            const double absolute_motion[3], long long global_first, long long global_count);

/* Initialisers on the resident set, drawing from the filter's drand48 stream like the reference:
 * ParticleFilter::initWithGaussian (particle_filter.cpp:105-132) given what PDFGaussian's constructor derives from
 * the covariance with Eigen::EigenSolver (third party; pdf_gaussian.cpp:32-46,100-131): `rotation` = cr_ (row-major
 * 3x3), `sigma` = cd_ (square roots of the eigenvalues); every sample is mean + cr * (draw(cd0), draw(cd1), draw(cd2)).
 * ParticleFilter::initWithPoseFn (:135-163) with the generator set by bpf_pf_set_random_pose_generator (global
 * localisation, node.cpp:870-882).  Both fill max_samples particles with weight 1/max_samples, build the set's
 * histogram tree (leaf count), zero w_slow / w_fast and clear the converged flag. */
int bpf_pf_init_with_gaussian(bpf_engine* e, const double mean[3], const double rotation[9], const double sigma[3]);
int bpf_pf_init_with_random_poses(bpf_engine* e);

/* ------------------------------------------------------------------ cluster statistics
 * ParticleFilter::computeClusterStatsForSet (particle_filter.cpp:505-636) with PFKDTree::cluster
 * (pf_kdtree.cpp:58-90,169-194), and what Node2D::getMaxWeightPose (node_2d.cpp:588-617) reads.
 * Evaluated lazily (first query after the set changed) on the device: occupied bins by hash table, clusters by
 * union-find over the bins' 26-neighbourhoods, labels in the reference's creation order, per-cluster sums as
 * order-independent 128-bit fixed-point accumulators (equal to the reference's serial sums up to summation rounding,
 * the same bits every run); the host reads back one small result block, the cluster array only on
 * bpf_pf_get_cluster.  BPF_OPT_STATS_HOST = 1 evaluates on the host from a copy of the set instead, bit for bit in the
 * reference's order. */
typedef struct
{
  int count;        /* PFCluster::count */
  double weight;    /* PFCluster::weight */
  double mean[3];   /* PFCluster::mean */
  double cov[5];    /* PFCluster::cov at (0,0) (0,1) (1,0) (1,1) (2,2); the other entries are unset in the reference */
} bpf_cluster;
/* cluster_count = PFSampleSet::cluster_count; set_mean / set_cov (nullable) = PFSampleSet::mean / cov */
int bpf_pf_compute_cluster_stats(bpf_engine* e, int* cluster_count_out, double set_mean[3], double set_cov[5]);
/* ParticleFilter::getClusterStats(cidx, &weight, &mean): returns BPF_ERR_INVALID_ARGUMENT for cidx >= cluster_count */
int bpf_pf_get_cluster(bpf_engine* e, int cidx, bpf_cluster* out);
/* Node2D::getMaxWeightPose: the heaviest cluster's weight and mean (weight 0 when there is none) */
int bpf_pf_get_max_weight_pose(bpf_engine* e, double* max_weight, double pose[3]);

/* ------------------------------------------------------------------ 3-D map + point cloud
 * OctoMap LUT state (include/amcl/map/octomap.h:96-110): pose_indices_, distance_ratios_,
 * cropped_min_cells_, cropped_max_cells_, resolution_, max_distance_to_object_.
 * Besides the two arrays the engine keeps a dense copy laid out for the scoring kernel's gathers when it fits 1 GiB;
 * this call also reads distance_ratios once on the host to find two byte values no entry holds: they mark the dense
 * copy's border cells (off the map) and one spare plane (a zero term), which is what lets the scoring kernel of a
 * planar mounting do without its on-the-map comparisons.  A LUT that uses more than 254 distinct ratios keeps the
 * comparisons; the weights are the same either way. */
int bpf_map3d_set(bpf_engine* e, const uint32_t* pose_indices, size_t n_pose_indices,
                  const uint8_t* distance_ratios, size_t n_distance_ratios, const int min_cells[3],
                  const int max_cells[3], double resolution, double max_dist);

/* OctoMap::updateDistancesLUT (octomap.cpp:175-333) from the occupied voxel indices instead of an octree
 * (the octomap library is the caller's): `occupied_ijk` = n x (i, j, k) map cells of the occupied leaves in
 * the octree's leaf-iteration order (that order fixes where each z column lands in distance_ratios);
 * min_cells / max_cells = cropped_min_cells_ / cropped_max_cells_.  FIFO brushfire with the reference's uint8
 * quantisation and seeding order (priority_queue<Index3>, octomap.h:49-55), run on the host; the result is
 * uploaded like bpf_map3d_set.  SURVEY 8(f) next-3. */
int bpf_map3d_build_distances_lut(bpf_engine* e, const int* occupied_ijk, size_t n_occupied, const int min_cells[3],
                                  const int max_cells[3], double resolution, double max_dist);
/* copies out what the builder (or bpf_map3d_set) holds; either pointer may be NULL to query the sizes only */
/* FIFO generations the last bpf_map3d_build_distances_lut ran on the device (0: it ran on the host) */
int bpf_map3d_builder_generations(bpf_engine* e, int* generations_out);
int bpf_map3d_get_distances_lut(bpf_engine* e, uint32_t* pose_indices, size_t pose_capacity, size_t* n_pose_indices,
                                uint8_t* distance_ratios, size_t ratios_capacity, size_t* n_distance_ratios);
/* PointCloudScanner::{init, setPointCloudModel, setPointCloudModelGompertz, setMapFactors,
 * setPointCloudScannerToFootprintTF} (point_cloud_scanner.cpp:48-90). */
int bpf_cloud_init(bpf_engine* e, int max_beams);
int bpf_cloud_set_model(bpf_engine* e, double z_hit, double z_rand, double sigma_hit);
int bpf_cloud_set_model_gompertz(bpf_engine* e, double z_hit, double z_rand, double sigma_hit, double gompertz_a,
                                 double gompertz_b, double gompertz_c, double input_shift, double input_scale,
                                 double output_shift);
int bpf_cloud_set_map_factors(bpf_engine* e, double off_map_factor, double non_free_space_factor,
                              double non_free_space_radius);
int bpf_cloud_set_scanner_to_footprint_tf(bpf_engine* e, const double xyz[3], const double quat_xyzw[4]);
/* PointCloudScanner::applyModelToSampleSet (point_cloud_scanner.cpp:106-129); points are
 * PointCloudData::points_ as packed float xyz triples in the scanner frame. */
double bpf_cloud_apply_model_to_sample_set(bpf_engine* e, double* samples, int sample_count, const float* points_xyz,
                                           int n_points, int* status);
/* PointCloudScanner::updateSensor(pf, data) on the resident set (:92-102). */
int bpf_pf_update_sensor_cloud(bpf_engine* e, const float* points_xyz, int n_points);

/* ------------------------------------------------------------------ sharded operation
 * One engine per GPU holds a contiguous shard (rank order = particle index order) of ONE
 * filter.  Scoring needs no exchange; normalisation needs the per-shard weight totals;
 * resampling needs the per-shard CDF sums and one sum-exchange of the candidate draw window.
 * The exchanges themselves (RCCL all-gather / all-reduce over xGMI) are issued by the host
 * layer on the same stream between these stage calls; every `_dev` pointer is device memory
 * and nothing here synchronises with the host.  badger_amcl_amd/sharded.py is the driver. */
/* score + recalcWeight on the local shard; the local weight total lands in scalars[0] */
int bpf_shard_score_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                           double range_max);
/* Beam skipping of the prob model (planar_scanner.cpp:352-395,482-529) needs, per beam, the number of particles
 * of the WHOLE set that agree with it.  When it is active (model prob, do_beamskip, set converged)
 * bpf_shard_score_planar stops after the counting pass and returns BPF_SHARD_NEED_BEAM_COUNTS (> 0, not an error):
 * sum the int32 counts of bpf_shard_beam_counts_dev over the shards in place (all-reduce), then call
 * bpf_shard_score_planar_finish with the same scan and the global particle count. */
#define BPF_SHARD_NEED_BEAM_COUNTS 100
int bpf_shard_beam_counts_dev(bpf_engine* e, void** counts_dev, int* n_counts);
int bpf_shard_score_planar_finish(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                  double range_max, long long global_count);
/* the same for the 3-D path (PointCloudScanner::applyModelToSampleSet on the local shard) */
int bpf_shard_score_cloud(bpf_engine* e, const float* points_xyz, int n_points);
/* device address of the engine's scalar block, double[16]: [0] local weight total,
 * [1] w_slow, [2] w_fast, [7] local CDF sum (after bpf_shard_build_cdf) */
int bpf_shard_scalars_dev(bpf_engine* e, void** dev_ptr);
/* ParticleFilter::updateSensor's normalisation (particle_filter.cpp:237-266) with the global
 * total = totals_dev[0] + ... + totals_dev[world-1] (added in rank order) and the global count */
int bpf_shard_normalize_dev(bpf_engine* e, const void* totals_dev, int world, int global_sample_count);
/* local running sum c[0..n] of the shard's weights; its last element is copied to scalars[7];
 * flags_dev (nullable): an int the call zeroes, the CDF-miss flag of the draw windows that follow */
int bpf_shard_build_cdf(bpf_engine* e, void* flags_dev);
/* Candidate draws m in [m0, m1) of the multinomial resampler (particle_filter.cpp:381-414) from
 * the drand48 state `rng_state48`.  The shard owns the draws whose r lies in
 * [offset, offset + sums_dev[rank]) with offset = sums_dev[0] + ... + sums_dev[rank-1].
 * sums_are_totals = 1: sums_dev holds the gathered WEIGHT TOTALS of the last sensor update instead
 * (no second exchange): shard q's slice is then total_q / sum(totals), formed identically on every
 * rank; inside it the shard's own running sum is used and its last particle absorbs the rounding.
 * window_dev is int64[6][stride]: rows 0-2 the bit patterns of the selected pose (x, y, theta),
 * rows 3-5 its histogram key; column m - m0.  Owned draws are written, all others are zeroed, so
 * an integer sum over the ranks assembles the window exactly.  flags_dev[0] is set on a CDF miss. */
int bpf_shard_draw_window_dev(bpf_engine* e, uint64_t rng_state48, int m0, int m1, const void* sums_dev,
                              int sums_are_totals, int rank, int world, void* window_dev, int stride, void* flags_dev);
/* Tail of the sharded resample for a small set (global_count <= 8192), one launch: adopt poses
 * [lo, hi) of the assembled arrays with weight 1/global_count, flip the sets, and evaluate
 * updateConverged over all global_count poses. */
int bpf_shard_tail_small_dev(bpf_engine* e, const void* x_all_dev, const void* y_all_dev, const void* theta_all_dev,
                             int global_count, int lo, int hi, int leaf_count, int bin_count);
/* Become the resampled shard: copy `count` poses from device arrays, weight 1/global_count each
 * (particle_filter.cpp:409,458-462), flip the ping-pong sets. */
int bpf_shard_adopt_dev(bpf_engine* e, const void* x_dev, const void* y_dev, const void* theta_dev, int count,
                        int global_count, int leaf_count, int bin_count);
/* updateConverged (particle_filter.cpp:170-220) over the WHOLE resampled set (every rank holds it
 * after the window exchange); fetched lazily by bpf_pf_get_state. */
int bpf_shard_converged_dev(bpf_engine* e, const void* x_all_dev, const void* y_all_dev, int global_count);
/* Advance a drand48 state by n draws (host arithmetic; the LCG jump the kernels use). */
uint64_t bpf_drand48_skip(uint64_t state48, uint64_t n);
/* Host-side exact KLD stop rule: replay ordered histogram keys through the fork's kd-tree
 * (pf_kdtree.cpp:97-150) and apply resampleLimit after each (particle_filter.cpp:416).
 * State persists in the engine between calls so windows can be fed one after another.
 * keys: int64 triples when keys_are_int64 != 0 (the window rows), else int32 triples, laid out
 * as three rows of `stride` (row-major [3][stride]). */
int bpf_kld_reset(bpf_engine* e);
int bpf_kld_feed(bpf_engine* e, const void* keys, int keys_are_int64, int stride, int n_keys, int first_draw_index,
                 int* stop_count_out);
/* the same with the keys still on the device: rows 3..5 of an assembled draw window (int64 [6][stride], see
 * bpf_shard_draw_window_dev).  A copy kernel on the engine's stream leaves them in pinned host memory behind a
 * generation word the host spins on (no D2H copy call, no stream synchronisation), then the replay runs. */
int bpf_kld_feed_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys, int first_draw_index,
                     int* stop_count_out);
/* Brackets of a sharded resample.  begin: w_diff = max(0, 1 - w_fast / w_slow) from the engine's averages (the
 * same on every shard); for the multinomial resampler with w_diff > 0 it resolves where every candidate draw finds
 * its stream elements (kernels_recovery.hpp) -- bpf_shard_draw_window_dev then follows that chain and shard 0
 * writes the random free-space poses; *systematic_count_out = resampleLimit(leaf_count), grown by (1 + w_diff)
 * (particle_filter.cpp:295-306), the `count` to pass to bpf_shard_systematic_window_dev, whose first
 * int(w_diff * count) samples are random poses.  end: the drand48 state after `sample_count` samples, and the
 * reset of the averages when w_diff > 0 (:453-455).  Needs bpf_pf_set_random_pose_generator when w_diff > 0. */
int bpf_shard_begin_resample(bpf_engine* e, uint64_t rng_state48, int leaf_count, double* w_diff_out,
                             int* systematic_count_out);
int bpf_shard_end_resample(bpf_engine* e, int sample_count, uint64_t* rng_state48_out);
/* Systematic resampling over shards (particle_filter.cpp:269-354): the targets
 * start + m / count are formed on the host exactly as the reference's serial chain, every shard resolves the
 * targets that fall into its slice of the global CDF and writes pose bits + key into its window columns
 * (zeros elsewhere), as bpf_shard_draw_window_dev does for the multinomial draws.  count = what
 * bpf_shard_begin_resample returned; rng_state48 = state BEFORE the one drand48. */
int bpf_pf_resample_limit(bpf_engine* e, int leaf_count, int* count_out);
int bpf_shard_systematic_window_dev(bpf_engine* e, uint64_t rng_state48, int count, const void* sums_dev,
                                    int sums_are_totals, int rank, int world, void* window_dev, int stride,
                                    void* flags_dev);
/* Mailbox exchange: the two small exchanges of the sharded path (W weight totals; one draw window whose every
 * column has exactly one writer) without a collective library.  Every engine of the node owns one uncached device
 * allocation, exported by IPC handle and mapped by the W - 1 others; a producer kernel stores its values into the
 * same slot of every peer's mailbox over xGMI and then a generation word (system-scope release), a consumer kernel
 * waits on the words in its own mailbox (bounded: 5 s, then BPF_ERR_EXCHANGE at the next host check).  The post
 * rides on the kernel that produces the value and the wait on the kernel that consumes it: no extra launch and no
 * host round trip.  Stands where torch.distributed / RCCL all-gather + all-reduce would (badger_amcl_amd/sharded.py
 * falls back to those when a mailbox cannot be set up); there is no counterpart in the reference.
 *   create   allocates for windows of up to max_window draws and returns the 64-byte IPC handle;
 *   connect  takes the world x 64 bytes of all ranks' handles in rank order (gather them with any host-side
 *            transport), maps the peers and runs one post-and-wait round with all of them -- every rank must call
 *            it at about the same time; BPF_ERR_EXCHANGE when a peer does not answer;
 *   totals   after a sharded scoring stage: device pointer to the W totals of this update, to be passed as
 *            totals_dev / sums_dev.  This rank's total is posted to the peers by the scoring stage or, for the field
 *            models, by the bpf_shard_normalize_dev launch that must follow (it folds the scoring kernel's partials,
 *            posts, and then waits in-kernel for all W totals);
 *   window   a fresh [6][stride] int64 window for the next exchange: bpf_shard_draw_window_dev /
 *            bpf_shard_systematic_window_dev given this pointer store every owned column into all peers' copies, and
 *            the first consumer (bpf_kld_feed_dev / bpf_kld_insert_dev / bpf_kld_stop_dev) waits for all shards.
 *            The window stays valid until the next-but-one call; copy out what has to live longer. */
/* A wait is bounded (default 5 s; set before create / connect).  When a bound runs out the consumer kernel leaves its
 * data alone -- after a failed wait for the totals the weights stay scored but NOT normalised, the local total in
 * bpf_shard_scalars_dev [0]; a failed window wait touches nothing of the current set -- and the next host check returns
 * BPF_ERR_EXCHANGE.  bpf_shard_mailbox_error_stage says which exchange it was, so that a driver can finish the update
 * over its other transport (all-gather the local totals, bpf_shard_normalize_dev, resample with collectives) and set
 * the mailbox up again: badger_amcl_amd/sharded.py does exactly that. */
int bpf_shard_mailbox_set_timeout_ms(bpf_engine* e, int timeout_ms);
int bpf_shard_mailbox_error_stage(bpf_engine* e, int* totals_failed, int* window_failed);
#define BPF_MAILBOX_HANDLE_BYTES 64
int bpf_shard_mailbox_create(bpf_engine* e, int rank, int world, long long max_window, void* handle_out);
int bpf_shard_mailbox_connect(bpf_engine* e, const void* handles);
/* Mailbox mode, the sharded sensor update and resample as one call each: what badger_amcl_amd/sharded.py does with
 * the stage functions above, in the same order, with the exchanges inside the kernels -- so that a host pays one
 * call per update.  update_sensor: bpf_shard_score_planar + totals + bpf_shard_normalize_dev (returns
 * BPF_SHARD_NEED_BEAM_COUNTS unchanged when beam skipping needs the caller's all-reduce: finish with the stage
 * functions then).  update_resample: multinomial (windows sized from *window_hint_io, follow-up windows, whole-stream
 * device tree) or systematic; *global_count_io / *leaf_count_io: the global sample count and the leaf count of the
 * current set's tree in, those of the new set out; this rank adopts its even share [M r / W, M (r + 1) / W);
 * flags_dev as in bpf_shard_build_cdf / bpf_shard_draw_window_dev (int32[>= 1] on the device, [0] = CDF-miss flag). */
int bpf_shard_mailbox_update_sensor_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                           double range_max, long long global_count);
int bpf_shard_mailbox_update_resample(bpf_engine* e, void* flags_dev, int* global_count_io, int* leaf_count_io,
                                      int* bin_count_out, int* windows_out, int* window_hint_io);
/* `rounds` full window exchanges with a checkable payload (every rank writes a pattern into its share of the columns
 * of every peer, every rank verifies all columns after the wait); all ranks must call it together, after connect.
 * BPF_ERR_EXCHANGE when a cell did not arrive as written or a wait ran out. */
int bpf_shard_mailbox_selftest(bpf_engine* e, int rounds);
int bpf_shard_mailbox_destroy(bpf_engine* e);
int bpf_shard_mailbox_totals(bpf_engine* e, void** totals_dev);
int bpf_shard_mailbox_window(bpf_engine* e, void** window_dev, int* stride);
/* Bring-up of a sharded filter from a plain C / C++ host (no Python, no MPI, no launcher): every rank of the node calls
 * bpf_shard_bootstrap with the same "host:port".  Rank 0 listens there, the others connect (TCP inside this library);
 * over those sockets the ranks gather the mailbox IPC handles, map each other's mailboxes, run the connect round and the
 * self-test and agree on the result.  When the mailbox cannot be used on every rank they all fall back to RCCL:
 * libbadger_pf_rccl.so (linked against librccl, loaded only then) joins a communicator whose unique id rank 0 hands out
 * over the same sockets, and the two exchanges become ncclAllGather (totals) and an integer ncclAllReduce (windows).
 * *mode_out: which one it is.  flags: BPF_BOOTSTRAP_FORCE_COLLECTIVE skips the mailbox, BPF_BOOTSTRAP_MAILBOX_ONLY
 * fails instead of falling back.  The sockets are closed before the call returns.  The engine must carry the GLOBAL
 * min / max sample counts (bpf_pf_create) and this rank's shard of the set. */
enum
{
  BPF_SHARD_EXCHANGE_MAILBOX = 1,
  BPF_SHARD_EXCHANGE_RCCL = 2
};
enum
{
  BPF_BOOTSTRAP_FORCE_COLLECTIVE = 1,
  BPF_BOOTSTRAP_MAILBOX_ONLY = 2
};
int bpf_shard_bootstrap(bpf_engine* e, int rank, int world, const char* host_port, long long max_window, int flags,
                        int* mode_out);
int bpf_shard_shutdown(bpf_engine* e);
/* The sharded sensor update and resample as one call each, over whichever exchange bpf_shard_bootstrap (or
 * bpf_shard_mailbox_connect) set up; arguments as bpf_shard_mailbox_update_sensor_planar / _update_resample, with the
 * CDF-miss flag word owned by the engine (*cdf_miss_out, nullable, reads it back). */
int bpf_shard_update_sensor_planar(bpf_engine* e, const double* ranges, const double* angles, int range_count,
                                   double range_max, long long global_count);
int bpf_shard_update_resample(bpf_engine* e, int* global_count_io, int* leaf_count_io, int* bin_count_out,
                              int* windows_out, int* window_hint_io, int* cdf_miss_out);
/* insert every key of the window into the engine's histogram tree (no stop rule): the tree of a systematic
 * resample, or of an initial set (keys as bpf_kld_feed / bpf_kld_feed_dev take them) */
int bpf_kld_insert(bpf_engine* e, const void* keys, int keys_are_int64, int stride, int n_keys);
int bpf_kld_insert_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys);
/* the stop rule for a WHOLE candidate stream (draws 0 .. n_keys-1, keys in rows 3..5 of the device window)
 * evaluated on the device (level-synchronous build of the same kd-tree).
 * *handled_out = 0 when a key does not fit the 64-bit packing or the tree is deeper than 256 levels: feed the
 * stream through bpf_kld_feed_dev instead.  *stop_count_out = -1: no stop up to n_keys. */
int bpf_kld_stop_dev(bpf_engine* e, const void* window_dev, int stride, int n_keys, int* handled_out,
                     int* stop_count_out, int* leaf_count_out, int* bin_count_out);
int bpf_kld_leaf_count(bpf_engine* e, int* leaf_count_out, int* bin_count_out);

/* ------------------------------------------------------------------ wire formats
 * The callers' data shaping, as plain host functions (no engine needed). */
/* Node2D::updateLatestScanData (node_2d.cpp:531-560): float LaserScan ranges -> PlanarData.
 * sensor_min_range / sensor_max_range <= 0 mean "not set".  ranges_out / angles_out hold n doubles. */
int bpf_wire_laserscan_to_planar(const float* ranges, int n, float msg_range_min, float msg_range_max,
                                 double sensor_min_range, double sensor_max_range, double angle_min,
                                 double angle_increment, double* ranges_out, double* angles_out,
                                 double* range_max_out);
/* Node2D::getAngleStats (node_2d.cpp:497-529): first bearing and bearing increment of a scanner in the base
 * frame (an upside-down scanner gets a negative increment).  q_base_from_scanner = rotation (x, y, z, w)
 * of the base_frame <- scan frame transform the node looks up.  Quaternion helpers are third-party tf2
 * (setRPY, operator*, getYaw), restated from their published form. */
int bpf_wire_scan_angle_stats(double msg_angle_min, double msg_angle_increment, const double q_base_from_scanner[4],
                              double* angle_min_out, double* angle_increment_out);
/* Node2D::convertMap (node_2d.cpp:265-295): nav_msgs/OccupancyGrid -> tri-state cells with integer
 * up-scaling and the centre origin (narrowed to float like pcl::PointXYZ).  cells_out holds
 * (width*scale) * (height*scale) int32. */
int bpf_wire_occupancy_grid_to_cells(const int8_t* data, int width, int height, double msg_resolution,
                                     double msg_origin_x, double msg_origin_y, int map_scale_up_factor,
                                     int32_t* cells_out, int* size_x_out, int* size_y_out, float origin_out[2],
                                     double* resolution_out);
/* Node3D::updateLatestScanData (node_3d.cpp:467-480): keep every step-th point, step = max((n-1)/(max_beams-1), 1).
 * Returns the number of points written (xyz triples); capacity in points. */
int bpf_wire_decimate_cloud(const float* points_xyz, int n_points, int max_beams, float* out_xyz, int capacity);
/* Node::publishParticleCloud (node.cpp:335-357): samples -> PoseArray entries {x, y, 0, qx, qy, qz, qw}
 * with q = setRPY(0, 0, theta). */
int bpf_wire_samples_to_pose_array(const double* samples, int sample_count, double* poses7_out);

/* ------------------------------------------------------------------ measurement */
/* free / total device memory as hipMemGetInfo reports it (diagnostic; used by the leak test) */
int bpf_device_memory_info(int device_ordinal, size_t* free_bytes, size_t* total_bytes);
enum
{
  BPF_K_SCORE = 0,     /* sensor scoring kernel, gather form (k_score_field / k_score_beam / k_cloud_score) */
  BPF_K_REDUCE = 1,
  BPF_K_NORMALIZE = 2,
  BPF_K_CDF = 3,
  BPF_K_DRAW = 4,
  BPF_K_FINALIZE = 5,
  BPF_K_SCORE_WINDOW = 6, /* sensor scoring kernel, LDS-window form (k_score_window) */
  BPF_K_SCORE_AUX = 7,    /* its helpers: k_field_prep, k_field_windows, k_field_finish */
  BPF_K_MOTION = 8,       /* k_motion_* */
  BPF_K_COUNT = 9
};
typedef struct
{
  double ms[BPF_K_COUNT];          /* accumulated HIP-event time per kernel class */
  long long launches[BPF_K_COUNT];
} bpf_profile;
/* on = 1: every 8th launch of the dominant (scoring) kernel is timed by a pair of hipEvents attached to the
 * dispatch itself (hipExtLaunchKernelGGL: the kernel's own start-to-end on the engine stream, which is what the
 * rocprofv3 kernel trace reports; a timed dispatch costs the update ~5 us, hence the sampling -- `launches` counts
 * the timed ones); on = 2: every scoring launch that way and every other kernel class bracketed by hipEventRecord
 * (costs host time per event, so not for timed regions); on = 3: as 1 but EVERY scoring launch (for scoring kernels
 * of a millisecond or more, where the 5 us do not show); 0: off. */
int bpf_profile_enable(bpf_engine* e, int on);
int bpf_profile_reset(bpf_engine* e);
int bpf_profile_get(bpf_engine* e, bpf_profile* out);
/* Name of the scoring kernel as rocprofv3 prints it, for matching profiles/ summaries. */
const char* bpf_score_kernel_name(const bpf_engine* e);
/* Decision of the last likelihood-field update: *used_window = 1 when the LDS-window kernels did
 * it, with how many of the 64-beam chunks the window plan expected to cover (synchronises). */
int bpf_get_window_plan(bpf_engine* e, int* used_window, int* chunks_covered, int* chunks_total);

#ifdef __cplusplus
}
#endif
#endif /* BADGER_PF_H */
