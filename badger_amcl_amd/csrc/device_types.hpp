// Device-visible plain structs shared by the kernels and the host engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bpf
{

// 2-D map as it lives in HBM.
//  * lut_tiles: the distance LUT re-encoded as 16-bit level indices in 8x8-cell tiles
//    (one tile = 128 B = one cache line), so that beam end points that are close in
//    EITHER axis share lines.  levels[idx] is the float distance of the reference's
//    distances_lut_ (include/amcl/map/occupancy_map.h:99).  The tiled image carries a
//    border of one cell on every side filled with the off-map level K: an end point is
//    clamped into [0, size+1] per axis (one unsigned min) and looked up without a bounds test.
//    The stored value is level*8, the byte offset of the level's term in the per-scan table.
//    Inside a tile the cells are stored u-major ((u&7)*8 + (v&7)), which makes the byte offset of padded
//    cell (u, v) a sum with a single masked term:  16*u + 2*v + (16*ltx - 2)*(v & ~7).
//  * cheb: for raycasts, one 32-bit word per cell of the map padded by one cell all round (row-major,
//    padded cell (x+1, y+1), row length size_x+2): byte q = chessboard distance to the nearest cell that is
//    not CELL_FREE or lies outside the map (the ring counts as blocked) within quadrant q of the cell (bit 0 of q:
//    towards -x, bit 1: towards -y, axes included), capped at 255.
//  * cells8: the tri-state grid narrowed to int8, row-major i + j*size_x.
struct MapDev
{
  const uint16_t* lut_tiles;
  const uint32_t* cheb;
  const int8_t* cells8;
  const float* levels;
  int size_x, size_y;
  int ltx, lty;           // tiles per row / column of the padded LUT image
  int half_x, half_y;     // size/2, the centre offset of convertWorldToMap
  int n_levels;        // K; index K is reserved for "off map"
  double origin_x, origin_y;  // float origin promoted to double (occupancy_map.cpp:96-97)
  double resolution;
  double max_dist;
};

struct ParticlesDev
{
  double* x;
  double* y;
  double* th;
  double* w;
};

struct GompertzDev
{
  double a, b, c, input_shift, input_scale, output_shift;
};

// Arguments of the likelihood-field family scoring kernel (LF / Gompertz / prob pass).
struct FieldScoreArgs
{
  ParticlesDev p;
  int n;
  const double4* prep;   // per particle (Qx, Qy, cos, sin) from k_field_prep
  double* w_host;        // HOST_MODE 1 (the host-buffer seam): pinned host array that receives the new weights too
  double4* rec;          // HOST_MODE 2: the caller's records in registered host memory, read and written in place
  const double2* beams;  // per valid beam: r*cos(bearing)/res, r*sin(bearing)/res
  int n_beams;
  const double* table;   // per LUT level (+1 off-map entry): the per-beam term
  int table_len;
  MapDev map;
  double sp_x, sp_y, sp_th;  // scanner pose in the robot frame
  double off_map_factor, non_free_factor, non_free_radius;
  double extra_term;         // sum of the terms of beams that are off the map for every pose
  int per_wave;              // particles owned by each wave (static partition)
  // graded partition (share_count[0] > 0): the waves of block b own share_count[r] particles each, r = b /
  // blocks_per_round, starting at share_base[r] + ((b % blocks_per_round) * 4 + wave) * share_count[r]
  int share_count[8];
  int share_base[8];
  int blocks_per_round;
  int xcd_local;             // graded partition: the eighth of the particle range of a block's XCD (see the kernel)
  const int* perm;           // HOST_MODE 3 (tile-sorted scoring): particle of slot j
  int model;
  GompertzDev g;
  int n_valid;               // beams that count towards the Gompertz mean
  double* block_partials;    // per-block sum of the updated weights (nullable)
  const int* skip_if_set;    // nullable: when *skip_if_set != 0 the window path does this update
  // prob model, counting pass
  int* obs_count;            // per staged beam: particles whose end point is near an obstacle
  int skip_level;            // levels below this index are "z < beam_skip_distance"
};

// drand48 jump-ahead tables: A[j] = a^(2^j), C[j] = c*(a^(2^j)-1)/(a-1)  (mod 2^48)
struct LcgJump
{
  uint64_t A[48];
  uint64_t C[48];
};

}  // namespace bpf
