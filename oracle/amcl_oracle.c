/*
 * amcl_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see amcl_oracle.h).
 *
 * CPU restatement of the badger_amcl sensor-update + resample hot path.
 * Written from the reference's behaviour, not copied: containers, control flow
 * and naming are this file's own; the arithmetic (operation order, types,
 * rounding points) follows the cited reference lines so that results agree
 * bit for bit wherever the same libm is used.
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off (no FMA contraction: the
 * reference is built for baseline x86-64, which has none).
 */
#include "amcl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* RNG: glibc drand48.  48-bit LCG X <- a*X + c mod 2^48,         */
/* a = 0x5DEECE66D, c = 0xB; srand48(s): X = (s << 16) | 0x330E; the result    */
/* is X / 2^48.  glibc's unseeded state is X = 0.                              */
/* ------------------------------------------------------------------------- */
#define ORC_LCG_A 0x5DEECE66DULL
#define ORC_LCG_C 0xBULL
#define ORC_MASK48 ((1ULL << 48) - 1)

void orc_srand48(uint64_t* state, long seed)
{
  *state = ((((uint64_t)seed) & 0xFFFFFFFFULL) << 16) | 0x330EULL;
}

double orc_drand48(uint64_t* state)
{
  *state = (ORC_LCG_A * (*state) + ORC_LCG_C) & ORC_MASK48;
  return ldexp((double)(*state), -48);
}

/* pdf_gaussian.cpp:77-97 -- polar Box-Muller; note it returns the x2 branch */
double orc_gaussian_draw(uint64_t* state, double sigma)
{
  double u, a, b, s;
  for (;;)
  {
    do
      u = orc_drand48(state);
    while (u == 0.0);
    a = 2.0 * u - 1.0;
    do
      u = orc_drand48(state);
    while (u == 0.0);
    b = 2.0 * u - 1.0;
    s = a * a + b * b;
    if (!(s > 1.0 || s == 0.0))
      break;
  }
  return sigma * b * sqrt(-2.0 * log(s) / s);
}

/* angles::normalize_angle -- third-party ROS `angles` header, version not
 * pinned by the reference (package.xml depends on `angles` unversioned).
 * Restated from the published Noetic form (angles 1.9.13):
 *   r = fmod(a + pi, 2 pi);  r <= 0 ? r + pi : r - pi.
 * Used once per particle at planar_scanner.cpp:699. */
double orc_normalize_angle(double a)
{
  const double r = fmod(a + M_PI, 2.0 * M_PI);
  return (r <= 0.0) ? r + M_PI : r - M_PI;
}

/* Odom::angleDiff (odom.cpp:308-311) = angles::shortest_angular_distance(b, a) = normalize_angle(a - b) */
static double odom_angle_diff(double a, double b)
{
  return orc_normalize_angle(a - b);
}

/* Odom::updateAction, odom.cpp:74-301.  All three draws of a particle come from the one global
 * drand48 stream in the order written in the reference. */
void orc_odom_update_action(int model, const double alpha[5], const double pose[3], const double delta[3],
                            const double absolute_motion[3], double* samples, int n, uint64_t* rng)
{
  const double a1 = alpha[0], a2 = alpha[1], a3 = alpha[2], a4 = alpha[3], a5 = alpha[4];
  const double old_th = pose[2] - delta[2]; /* :82-85, only the heading is used */
  if (model == 1 || model == 3) /* omni :94-124, omni-corrected :171-201 */
  {
    const double delta_trans = sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
    const double delta_rot = delta[2];
    double trans_sd = a3 * (delta_trans * delta_trans) + a1 * (delta_rot * delta_rot);
    double rot_sd = a4 * (delta_rot * delta_rot) + a2 * (delta_trans * delta_trans);
    double strafe_sd = a1 * (delta_rot * delta_rot) + a5 * (delta_trans * delta_trans);
    if (model == 3)
    {
      trans_sd = sqrt(trans_sd);
      rot_sd = sqrt(rot_sd);
      strafe_sd = sqrt(strafe_sd);
    }
    for (int i = 0; i < n; i++)
    {
      double* p = &samples[4 * i];
      const double turn_angle = atan2(delta[1], delta[0]);
      const double bearing = odom_angle_diff(turn_angle, old_th) + p[2];
      const double cs = cos(bearing), sn = sin(bearing);
      const double trans_hat = delta_trans + orc_gaussian_draw(rng, trans_sd);
      const double rot_hat = delta_rot + orc_gaussian_draw(rng, rot_sd);
      const double strafe_hat = 0 + orc_gaussian_draw(rng, strafe_sd);
      p[0] += (trans_hat * cs + strafe_hat * sn);
      p[1] += (trans_hat * sn - strafe_hat * cs);
      p[2] += rot_hat;
    }
  }
  else if (model == 0 || model == 2) /* diff :125-170, diff-corrected :202-252 */
  {
    const double delta_trans = sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
    double rot1;
    if (delta_trans < 0.01) /* :135-136 tests sqrt(dy*dy + dx*dx): same two squares, commuted sum */
      rot1 = 0.0;
    else
      rot1 = odom_angle_diff(atan2(delta[1], delta[0]), old_th);
    const double rot2 = odom_angle_diff(delta[2], rot1);
    /* std::min(a, b) = (b < a) ? b : a */
    const double r1a = fabs(odom_angle_diff(rot1, 0.0)), r1b = fabs(odom_angle_diff(rot1, M_PI));
    const double r2a = fabs(odom_angle_diff(rot2, 0.0)), r2b = fabs(odom_angle_diff(rot2, M_PI));
    const double rot1_noise = (r1b < r1a) ? r1b : r1a;
    const double rot2_noise = (r2b < r2a) ? r2b : r2a;
    double sd1 = a1 * rot1_noise * rot1_noise + a2 * delta_trans * delta_trans;
    double sd2 = a3 * delta_trans * delta_trans + a4 * rot1_noise * rot1_noise + a4 * rot2_noise * rot2_noise;
    double sd3 = a1 * rot2_noise * rot2_noise + a2 * delta_trans * delta_trans;
    if (model == 2)
    {
      sd1 = sqrt(sd1);
      sd2 = sqrt(sd2);
      sd3 = sqrt(sd3);
    }
    for (int i = 0; i < n; i++)
    {
      double* p = &samples[4 * i];
      const double rot1_hat = odom_angle_diff(rot1, orc_gaussian_draw(rng, sd1));
      const double trans_hat = delta_trans - orc_gaussian_draw(rng, sd2);
      const double rot2_hat = odom_angle_diff(rot2, orc_gaussian_draw(rng, sd3));
      p[0] += trans_hat * cos(p[2] + rot1_hat);
      p[1] += trans_hat * sin(p[2] + rot1_hat);
      p[2] += rot1_hat + rot2_hat;
    }
  }
  else /* gaussian :253-298 */
  {
    const double delta_trans = sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
    const double delta_rot = delta[2];
    const double at2 = absolute_motion[0] * absolute_motion[0];
    const double as2 = absolute_motion[1] * absolute_motion[1];
    const double ar2 = absolute_motion[2] * absolute_motion[2];
    const double rot_sd = sqrt(a1 * ar2 + a2 * at2);
    const double trans_sd = sqrt(a3 * at2 + a4 * ar2);
    const double strafe_sd = sqrt(a4 * ar2 + a5 * as2);
    for (int i = 0; i < n; i++)
    {
      double* p = &samples[4 * i];
      const double heading = p[2] + delta[2] / 2;
      const double ch = cos(heading), sh = sin(heading);
      const double ndata_angle = atan2(delta[1], delta[0]);
      const double bearing = odom_angle_diff(ndata_angle, old_th) + p[2];
      const double cs = cos(bearing), sn = sin(bearing);
      const double trans_hat = orc_gaussian_draw(rng, trans_sd);
      const double strafe_hat = orc_gaussian_draw(rng, strafe_sd);
      const double rot_hat = orc_gaussian_draw(rng, rot_sd);
      p[0] += (delta_trans * cs);
      p[1] += (delta_trans * sn);
      p[2] += delta_rot;
      p[0] += (trans_hat * ch + strafe_hat * sh);
      p[1] += (trans_hat * sh - strafe_hat * ch);
      p[2] += rot_hat;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* 2-D occupancy map                                                           */
/* ------------------------------------------------------------------------- */

/* occupancy_map.cpp:90-98: floor((x - origin)/res + 0.5) + size/2, the float
 * origin promoted to double, the double sum truncated into an int. */
void orc_map2d_world_to_map(const orc_map2d* m, double x, double y, int* i, int* j)
{
  *i = (int)(floor((x - m->origin_x) / m->resolution + 0.5) + m->size_x / 2);
  *j = (int)(floor((y - m->origin_y) / m->resolution + 0.5) + m->size_y / 2);
}

/* occupancy_map.cpp:75-88 */
void orc_map2d_map_to_world(const orc_map2d* m, int i, int j, double* x, double* y)
{
  *x = m->origin_x + (i - m->size_x / 2) * m->resolution;
  *y = m->origin_y + (j - m->size_y / 2) * m->resolution;
}

/* occupancy_map.cpp:100-105 */
int orc_map2d_is_valid(const orc_map2d* m, int i, int j)
{
  return i >= 0 && i < m->size_x && j >= 0 && j < m->size_y;
}

static unsigned cell_index(const orc_map2d* m, int i, int j)
{
  return (unsigned)i + (unsigned)j * (unsigned)m->size_x; /* occupancy_map.cpp:107-110 */
}

/* occupancy_map.cpp:64-73: float return; off-map gives float(max_dist) */
float orc_map2d_distance(const orc_map2d* m, int i, int j)
{
  if (orc_map2d_is_valid(m, i, j))
    return m->lut[cell_index(m, i, j)];
  return (float)m->max_dist;
}

static int blocked(const orc_map2d* m, int i, int j)
{
  return !orc_map2d_is_valid(m, i, j) || m->cells[cell_index(m, i, j)] != -1;
}

/* occupancy_map.cpp:257-364: integer Bresenham between the cell of (ox,oy)
 * and the cell of the max-range endpoint; first cell that is off-map or not
 * FREE ends the ray; distance from integer cell deltas times resolution. */
double orc_map2d_calc_range(const orc_map2d* m, double ox, double oy, double oa, double max_range,
                            long* cells_visited)
{
  int x0, y0, x1, y1;
  orc_map2d_world_to_map(m, ox, oy, &x0, &y0);
  orc_map2d_world_to_map(m, ox + max_range * cos(oa), oy + max_range * sin(oa), &x1, &y1);
  if (x0 == x1 && y0 == y1)
    return max_range;

  const int steep = abs(y1 - y0) > abs(x1 - x0);
  if (steep)
  {
    int t = x0; x0 = y0; y0 = t;
    t = x1; x1 = y1; y1 = t;
  }
  const int dx = abs(x1 - x0), dy = abs(y1 - y0);
  const int sx = (x0 < x1) ? 1 : -1, sy = (y0 < y1) ? 1 : -1;
  int err = 0, x = x0, y = y0;
  long visited = 0;
  for (;;)
  {
    ++visited;
    if (steep ? blocked(m, y, x) : blocked(m, x, y))
    {
      if (cells_visited)
        *cells_visited += visited;
      return sqrt((double)((x - x0) * (x - x0) + (y - y0) * (y - y0))) * m->resolution;
    }
    if (x == x1 + sx)
      break;
    x += sx;
    err += dy;
    if (2 * err >= dx)
    {
      y += sy;
      err -= dx;
    }
  }
  if (cells_visited)
    *cells_visited += visited;
  return max_range;
}

/* ---- distance LUT brushfire (occupancy_map.cpp:122-252).  The reference uses
 * std::priority_queue whose order among equal keys is decided by libstdc++'s
 * binary-heap routines (bits/stl_heap.h: __push_heap, __adjust_heap); those are
 * restated here so tie order -- and hence which source reaches a cell first --
 * matches a libstdc++ build.  Comparator (occupancy_map.h:111-114): a < b iff
 * lut(a) > lut(b) on the float LUT values. */
typedef struct
{
  int i, j, si, sj;
} bf_cell;

typedef struct
{
  bf_cell* v;
  size_t n, cap;
  const float* lut;
  int size_x;
} bf_heap;

static int bf_less(const bf_heap* h, const bf_cell* a, const bf_cell* b)
{
  return h->lut[a->i + (size_t)a->j * h->size_x] > h->lut[b->i + (size_t)b->j * h->size_x];
}

static void bf_sift_up(bf_heap* h, size_t hole, size_t top, bf_cell val)
{
  while (hole > top)
  {
    size_t parent = (hole - 1) / 2;
    if (!bf_less(h, &h->v[parent], &val))
      break;
    h->v[hole] = h->v[parent];
    hole = parent;
  }
  h->v[hole] = val;
}

static void bf_push(bf_heap* h, bf_cell c)
{
  if (h->n == h->cap)
  {
    h->cap = h->cap ? h->cap * 2 : 1024;
    h->v = (bf_cell*)realloc(h->v, h->cap * sizeof(bf_cell));
  }
  h->n++;
  bf_sift_up(h, h->n - 1, 0, c);
}

static void bf_pop(bf_heap* h)
{
  /* pop_heap: last element becomes the value to re-insert from the root */
  if (h->n > 1)
  {
    bf_cell val = h->v[h->n - 1];
    const size_t len = h->n - 1;
    size_t hole = 0, child = 0;
    h->v[h->n - 1] = h->v[0];
    while (child < (len - 1) / 2 && len >= 1)
    {
      child = 2 * (child + 1);
      if (bf_less(h, &h->v[child], &h->v[child - 1]))
        child--;
      h->v[hole] = h->v[child];
      hole = child;
    }
    if ((len & 1) == 0 && len >= 2 && child == (len - 2) / 2)
    {
      child = 2 * (child + 1);
      h->v[hole] = h->v[child - 1];
      hole = child - 1;
    }
    bf_sift_up(h, hole, 0, val);
  }
  h->n--;
}

void orc_map2d_build_lut(int size_x, int size_y, const int32_t* cells, double resolution,
                         double max_dist, float* lut)
{
  if (max_dist == 0.0)
    return; /* occupancy_map.cpp:141-145 */
  const int radius = (int)floor(max_dist / resolution); /* :124 */
  const int tdim = radius + 2;
  double* dtab = (double*)malloc(sizeof(double) * tdim * tdim);
  for (int a = 0; a < tdim; a++)
    for (int b = 0; b < tdim; b++)
      dtab[a * tdim + b] = sqrt((double)(a * a + b * b)); /* :131 */
  const size_t ncell = (size_t)size_x * size_y;
  unsigned char* marked = (unsigned char*)calloc(ncell, 1);
  bf_heap h = { NULL, 0, 0, lut, size_x };

  /* :162-187 column-major sweep (i outer, j inner) */
  for (int i = 0; i < size_x; i++)
    for (int j = 0; j < size_y; j++)
    {
      const size_t idx = i + (size_t)j * size_x;
      if (cells[idx] == 1)
      {
        lut[idx] = 0.0f;
        marked[idx] = 1;
        bf_cell c = { i, j, i, j };
        bf_push(&h, c);
      }
      else
        lut[idx] = (float)max_dist;
    }

  /* :189-214: neighbours are examined while the current cell is still the
   * heap top; it is popped afterwards. */
  static const int di[4] = { -1, 0, 1, 0 }, dj[4] = { 0, -1, 0, 1 };
  while (h.n)
  {
    const bf_cell cur = h.v[0];
    for (int k = 0; k < 4; k++)
    {
      const int ni = cur.i + di[k], nj = cur.j + dj[k];
      if (ni < 0 || nj < 0 || ni > size_x - 1 || nj > size_y - 1)
        continue;
      const size_t nidx = ni + (size_t)nj * size_x;
      if (marked[nidx])
        continue;
      /* :227-245 */
      const int a = abs(ni - cur.si), b = abs(nj - cur.sj);
      if (a >= tdim || b >= tdim)
        continue; /* unreachable in the reference: distance test stops growth first */
      const double d = dtab[a * tdim + b];
      if (d <= radius)
      {
        lut[nidx] = (float)(d * resolution);
        bf_cell c = { ni, nj, cur.si, cur.sj };
        bf_push(&h, c);
        marked[nidx] = 1;
      }
    }
    bf_pop(&h);
  }
  free(h.v);
  free(marked);
  free(dtab);
}

/* ------------------------------------------------------------------------- */
/* planar scanner                                                              */
/* ------------------------------------------------------------------------- */
void orc_planar_defaults(orc_planar* p)
{
  memset(p, 0, sizeof(*p));
  p->model = ORC_MODEL_LIKELIHOOD_FIELD;
  p->off_map_factor = 1.0;        /* planar_scanner.cpp:42-44 */
  p->non_free_space_factor = 1.0;
  p->non_free_space_radius = 0.0;
}

/* planar_scanner.cpp:693-701 */
static void coord_add(const double a[3], const double b[3], double c[3])
{
  c[0] = b[0] + a[0] * cos(b[2]) - a[1] * sin(b[2]);
  c[1] = b[1] + a[0] * sin(b[2]) + a[1] * cos(b[2]);
  c[2] = orc_normalize_angle(b[2] + a[2]);
}

/* z for a likelihood-field endpoint: planar_scanner.cpp:287-300 */
static double lf_endpoint_distance(const orc_map2d* m, const double pose[3], double r, double bearing,
                                   int* on_map)
{
  const double hx = pose[0] + r * cos(pose[2] + bearing);
  const double hy = pose[1] + r * sin(pose[2] + bearing);
  int ci, cj;
  orc_map2d_world_to_map(m, hx, hy, &ci, &cj);
  *on_map = orc_map2d_is_valid(m, ci, cj);
  if (!*on_map)
    return m->max_dist;
  return (double)orc_map2d_distance(m, ci, cj);
}

/* planar_scanner.cpp:168-234 */
static double model_beam(const orc_planar* p, const orc_map2d* m, double* s, int n, const double* ranges,
                         const double* angles, int rc, double range_max, long* stats)
{
  double total = 0.0;
  const int step = (rc - 1) / (p->max_beams - 1); /* :193, not clamped */
  if (step < 1)
    return NAN; /* the reference loops forever here; the oracle refuses */
  for (int j = 0; j < n; j++)
  {
    double pose[3];
    coord_add(p->scanner_pose, &s[4 * j], pose);
    double acc = 1.0;
    for (int i = 0; i < rc; i += step)
    {
      const double obs = ranges[i];
      long walked = 0;
      const double map_range = orc_map2d_calc_range(m, pose[0], pose[1], pose[2] + angles[i], range_max, &walked);
      double pz = 0.0;
      const double z = obs - map_range;
      pz += p->z_hit * exp(-(z * z) / (2 * p->sigma_hit * p->sigma_hit));
      if (z < 0)
        pz += p->z_short * p->lambda_short * exp(-p->lambda_short * obs);
      if (obs == range_max)
        pz += p->z_max * 1.0;
      if (obs < range_max)
        pz += p->z_rand * 1.0 / range_max;
      acc += pz * pz * pz;
      if (stats)
      {
        stats[0] += 1;
        stats[1] += walked;
      }
    }
    s[4 * j + 3] *= acc;
    total += s[4 * j + 3];
  }
  return total;
}

static int lf_step(int rc, int max_beams)
{
  int step = (rc - 1) / (max_beams - 1); /* planar_scanner.cpp:265-269 */
  return step < 1 ? 1 : step;
}

/* planar_scanner.cpp:236-323 */
static double model_lf(const orc_planar* p, const orc_map2d* m, double* s, int n, const double* ranges,
                       const double* angles, int rc, double range_max, long* stats)
{
  double total = 0.0;
  for (int j = 0; j < n; j++)
  {
    double pose[3];
    coord_add(p->scanner_pose, &s[4 * j], pose);
    double acc = 1.0;
    const double denom = 2 * p->sigma_hit * p->sigma_hit;
    const double rand_mult = 1.0 / range_max;
    const int step = lf_step(rc, p->max_beams);
    for (int i = 0; i < rc; i += step)
    {
      const double r = ranges[i];
      if (r >= range_max)
        continue;
      if (r != r)
        continue;
      int on_map;
      const double z = lf_endpoint_distance(m, pose, r, angles[i], &on_map);
      double pz = 0.0;
      pz += p->z_hit * exp(-(z * z) / denom);
      pz += p->z_rand * rand_mult;
      acc += pz * pz * pz;
      if (stats)
        stats[0] += 1;
    }
    s[4 * j + 3] *= acc;
    total += s[4 * j + 3];
  }
  return total;
}

/* planar_scanner.cpp:540-550 */
static double gompertz(const orc_planar* p, double v)
{
  v = v * p->input_scale + p->input_shift;
  v = p->gompertz_a * exp(-1.0 * p->gompertz_b * exp(-1.0 * p->gompertz_c * v));
  v += p->output_shift;
  return v;
}

/* planar_scanner.cpp:552-640 */
static double model_gompertz(const orc_planar* p, const orc_map2d* m, double* s, int n, const double* ranges,
                             const double* angles, int rc, double range_max, long* stats)
{
  double total = 0.0;
  for (int j = 0; j < n; j++)
  {
    double pose[3];
    coord_add(p->scanner_pose, &s[4 * j], pose);
    const double denom = 2 * p->sigma_hit * p->sigma_hit;
    const int step = lf_step(rc, p->max_beams);
    int valid = 0;
    double sum = 0.0;
    for (int i = 0; i < rc; i += step)
    {
      const double r = ranges[i];
      if (r >= range_max)
        continue;
      if (r != r)
        continue;
      valid++;
      int on_map;
      const double z = lf_endpoint_distance(m, pose, r, angles[i], &on_map);
      double pz = 0.0;
      pz += p->z_hit * exp(-(z * z) / denom);
      pz += p->z_rand;
      sum += pz;
      if (stats)
        stats[0] += 1;
    }
    double w = 1.0;
    if (valid > 0)
      w = gompertz(p, sum / valid);
    s[4 * j + 3] *= w;
    total += s[4 * j + 3];
  }
  return total;
}

/* planar_scanner.cpp:325-533 */
static double model_prob(const orc_planar* p, const orc_map2d* m, double* s, int n, int set_converged,
                         const double* ranges, const double* angles, int rc, double range_max, long* stats)
{
  double total = 0.0;
  int step = (int)ceil(rc / (double)p->max_beams); /* :339 */
  if (step < 1)
    step = 1;
  const double denom = 2 * p->sigma_hit * p->sigma_hit;
  const double rand_mult = 1.0 / range_max;
  const double max_dist_prob = exp(-(m->max_dist * m->max_dist) / denom);
  const int beamskip = p->do_beamskip && set_converged; /* :361 */
  const int mb = p->max_beams;
  int* obs_count = (int*)calloc(mb, sizeof(int));
  double* temp = NULL;
  if (beamskip)
    temp = (double*)calloc((size_t)n * mb, sizeof(double)); /* :684-690: zero-filled */

  for (int j = 0; j < n; j++)
  {
    double pose[3];
    coord_add(p->scanner_pose, &s[4 * j], pose);
    double log_p = 0;
    int b = 0;
    for (int i = 0; i < rc; i += step, b++)
    {
      const double r = ranges[i];
      if (r >= range_max)
        continue;
      if (r != r)
        continue;
      double pz = 0.0;
      const double hx = pose[0] + r * cos(pose[2] + angles[i]);
      const double hy = pose[1] + r * sin(pose[2] + angles[i]);
      int ci, cj;
      orc_map2d_world_to_map(m, hx, hy, &ci, &cj);
      if (!orc_map2d_is_valid(m, ci, cj))
        pz += p->z_hit * max_dist_prob;
      else
      {
        const double z = (double)orc_map2d_distance(m, ci, cj);
        if (z < p->beam_skip_distance)
          obs_count[b] += 1; /* counted whether or not beam skipping is active (:448-451) */
        pz += p->z_hit * exp(-(z * z) / denom);
      }
      pz += p->z_rand * rand_mult;
      if (!beamskip)
        log_p += log(pz);
      else
        temp[(size_t)j * mb + b] = pz;
      if (stats)
        stats[0] += 1;
    }
    if (!beamskip)
    {
      s[4 * j + 3] *= exp(log_p);
      total += s[4 * j + 3];
    }
  }

  if (beamskip)
  {
    /* :482-529 */
    unsigned char* mask = (unsigned char*)calloc(mb, 1);
    int skipped = 0;
    for (int b = 0; b < mb; b++)
    {
      if ((obs_count[b] / (double)n) > p->beam_skip_threshold)
        mask[b] = 1;
      else
        skipped++;
    }
    const int error = skipped >= (mb * p->beam_skip_error_threshold);
    for (int j = 0; j < n; j++)
    {
      double log_p = 0;
      for (int b = 0; b < mb; b++)
        if (error || mask[b])
          log_p += log(temp[(size_t)j * mb + b]); /* unvisited slots hold 0.0 -> log = -inf */
      s[4 * j + 3] *= exp(log_p);
      total += s[4 * j + 3];
    }
    free(mask);
    free(temp);
  }
  free(obs_count);
  return total;
}

/* planar_scanner.cpp:642-682: uses the ROBOT pose, not the scanner pose */
static double recalc_weight(const orc_planar* p, const orc_map2d* m, double* s, int n)
{
  double rv = 0.0;
  for (int j = 0; j < n; j++)
  {
    int ci, cj;
    orc_map2d_world_to_map(m, s[4 * j], s[4 * j + 1], &ci, &cj);
    if (!orc_map2d_is_valid(m, ci, cj))
      s[4 * j + 3] *= p->off_map_factor;
    else if (m->cells[cell_index(m, ci, cj)] != -1)
      s[4 * j + 3] *= p->non_free_space_factor;
    else
    {
      const double d = orc_map2d_distance(m, ci, cj);
      if (d < p->non_free_space_radius)
      {
        const double frac = orc_map2d_distance(m, ci, cj) / p->non_free_space_radius;
        double f = p->non_free_space_factor;
        f += frac * (1.0 - p->non_free_space_factor);
        s[4 * j + 3] *= f;
      }
    }
    rv += s[4 * j + 3];
  }
  return rv;
}

/* planar_scanner.cpp:141-164 */
double orc_planar_apply(const orc_planar* p, const orc_map2d* m, double* samples, int n, int set_converged,
                        const double* ranges, const double* angles, int rc, double range_max, long* stats)
{
  if (p->max_beams < 2)
    return 0.0;
  double rv = 0.0;
  switch (p->model)
  {
    case ORC_MODEL_BEAM:
      rv = model_beam(p, m, samples, n, ranges, angles, rc, range_max, stats);
      break;
    case ORC_MODEL_LIKELIHOOD_FIELD:
      rv = model_lf(p, m, samples, n, ranges, angles, rc, range_max, stats);
      break;
    case ORC_MODEL_LIKELIHOOD_FIELD_PROB:
      rv = model_prob(p, m, samples, n, set_converged, ranges, angles, rc, range_max, stats);
      break;
    case ORC_MODEL_LIKELIHOOD_FIELD_GOMPERTZ:
      rv = model_gompertz(p, m, samples, n, ranges, angles, rc, range_max, stats);
      break;
    default:
      break;
  }
  if (rv > 0.0)
    rv = recalc_weight(p, m, samples, n);
  return rv;
}

/* ------------------------------------------------------------------------- */
/* kd-tree histogram (pf_kdtree.cpp).  Nodes live in a growable array and are   */
/* linked by index; insertion is iterative.  Semantics kept: a node is a "leaf" */
/* until the first different key is routed through it, at which point its pivot */
/* dimension is fixed (largest |delta|, first maximum wins) and the leaf count  */
/* drops by one; every new node adds one.                                       */
/* ------------------------------------------------------------------------- */
typedef struct
{
  int key[3];
  int pivot; /* -1 while leaf */
  int child[2];
  int cluster;
  double value;
} kd_node;

struct orc_kdtree
{
  kd_node* nodes;
  int n, cap;
  int leaf_count;
  double cell[3];
};

orc_kdtree* orc_kdtree_new(void)
{
  orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof(orc_kdtree));
  t->cell[0] = 0.50; /* pf_kdtree.cpp:35-37 */
  t->cell[1] = 0.50;
  t->cell[2] = (10 * M_PI / 180);
  return t;
}

void orc_kdtree_free(orc_kdtree* t)
{
  if (!t)
    return;
  free(t->nodes);
  free(t);
}

void orc_kdtree_clear(orc_kdtree* t)
{
  t->n = 0;
  t->leaf_count = 0;
}

static void pose_key(const orc_kdtree* t, const double pose[3], int key[3])
{
  /* pf_kdtree.cpp:52-54: floor(pose / cell) with un-normalised theta */
  key[0] = (int)floor(pose[0] / t->cell[0]);
  key[1] = (int)floor(pose[1] / t->cell[1]);
  key[2] = (int)floor(pose[2] / t->cell[2]);
}

static int key_eq(const int a[3], const int b[3])
{
  return a[0] == b[0] && a[1] == b[1] && a[2] == b[2];
}

static int kd_new_node(orc_kdtree* t, const int key[3], double value)
{
  if (t->n == t->cap)
  {
    t->cap = t->cap ? t->cap * 2 : 256;
    t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * t->cap);
  }
  kd_node* nd = &t->nodes[t->n];
  memcpy(nd->key, key, sizeof(int) * 3);
  nd->pivot = -1;
  nd->child[0] = nd->child[1] = -1;
  nd->cluster = -1;
  nd->value = value;
  t->leaf_count += 1; /* pf_kdtree.cpp:128 */
  return t->n++;
}

/* pf_kdtree.cpp:97-150 */
void orc_kdtree_insert_key(orc_kdtree* t, const int key[3], double value)
{
  if (t->n == 0)
  {
    kd_new_node(t, key, value);
    return;
  }
  int cur = 0;
  for (;;)
  {
    kd_node* nd = &t->nodes[cur];
    if (key_eq(key, nd->key))
    {
      nd->value += value;
      return;
    }
    if (nd->pivot == -1)
    {
      int best = 0;
      for (int d = 0; d < 3; d++)
      {
        const int split = abs(key[d] - nd->key[d]);
        if (split > best)
        {
          best = split;
          nd->pivot = d;
        }
      }
      t->leaf_count -= 1;
    }
    const int side = key[nd->pivot] > nd->key[nd->pivot];
    if (nd->child[side] < 0)
    {
      const int fresh = kd_new_node(t, key, value); /* may realloc: re-index, do not hold nd */
      t->nodes[cur].child[side] = fresh;
      return;
    }
    cur = nd->child[side];
  }
}

void orc_kdtree_insert(orc_kdtree* t, const double pose[3], double value)
{
  int key[3];
  pose_key(t, pose, key);
  orc_kdtree_insert_key(t, key, value);
}

int orc_kdtree_leaf_count(const orc_kdtree* t)
{
  return t->leaf_count;
}

int orc_kdtree_node_count(const orc_kdtree* t)
{
  return t->n;
}

/* pf_kdtree.cpp:152-167; a leaf with a different key yields "not found" */
static int kd_find(const orc_kdtree* t, const int key[3])
{
  int cur = t->n ? 0 : -1;
  while (cur >= 0)
  {
    const kd_node* nd = &t->nodes[cur];
    if (key_eq(key, nd->key))
      return cur;
    if (nd->pivot < 0)
      return -1;
    cur = nd->child[key[nd->pivot] > nd->key[nd->pivot]];
  }
  return -1;
}

/* pf_kdtree.cpp:58-76,169-194: 26-neighbourhood connected components; labels
 * are handed out in node-creation order of each component's first node.  The
 * reference recurses; an explicit stack gives the same labelling. */
void orc_kdtree_cluster(orc_kdtree* t)
{
  for (int i = 0; i < t->n; i++)
    t->nodes[i].cluster = -1;
  int* stack = (int*)malloc(sizeof(int) * (t->n > 0 ? t->n : 1));
  int label = 0;
  for (int i = 0; i < t->n; i++)
  {
    if (t->nodes[i].cluster != -1)
      continue;
    t->nodes[i].cluster = label;
    int sp = 0;
    stack[sp++] = i;
    while (sp)
    {
      const int cur = stack[--sp];
      for (int k = 0; k < 27; k++)
      {
        int nk[3];
        nk[0] = t->nodes[cur].key[0] + (k / 9) - 1;
        nk[1] = t->nodes[cur].key[1] + ((k % 9) / 3) - 1;
        nk[2] = t->nodes[cur].key[2] + ((k % 9) % 3) - 1;
        if (k == 13)
          continue;
        const int nb = kd_find(t, nk);
        if (nb < 0 || t->nodes[nb].cluster >= 0)
          continue;
        t->nodes[nb].cluster = label;
        stack[sp++] = nb;
      }
    }
    label++;
  }
  free(stack);
}

int orc_kdtree_get_cluster(const orc_kdtree* t, const double pose[3])
{
  int key[3];
  pose_key(t, pose, key);
  const int nd = kd_find(t, key);
  return nd < 0 ? -1 : t->nodes[nd].cluster;
}

/* ------------------------------------------------------------------------- */
/* particle filter core                                                        */
/* ------------------------------------------------------------------------- */
void orc_pf_init(orc_pf* pf, int min_samples, int max_samples, double alpha_slow, double alpha_fast,
                 double convergence_threshold)
{
  memset(pf, 0, sizeof(*pf));
  pf->min_samples = min_samples;
  pf->max_samples = max_samples;
  pf->pop_err = 0.01; /* particle_filter.cpp:58-60 */
  pf->pop_z = 3;
  pf->dist_threshold = 0.5;
  pf->alpha_slow = alpha_slow;
  pf->alpha_fast = alpha_fast;
  pf->convergence_threshold = convergence_threshold;
  pf->resample_model = ORC_RESAMPLE_MULTINOMIAL;
  pf->rng = 0; /* glibc's unseeded drand48 state */
}

/* particle_filter.cpp:237-266 */
void orc_pf_normalize(orc_pf* pf, double* s, int n, double total)
{
  if (total > 0.0)
  {
    double w_avg = 0.0;
    for (int i = 0; i < n; i++)
    {
      w_avg += s[4 * i + 3];
      s[4 * i + 3] /= total;
    }
    w_avg /= n;
    if (pf->w_slow == 0.0)
      pf->w_slow = w_avg;
    else
      pf->w_slow += pf->alpha_slow * (w_avg - pf->w_slow);
    if (pf->w_fast == 0.0)
      pf->w_fast = w_avg;
    else
      pf->w_fast += pf->alpha_fast * (w_avg - pf->w_fast);
  }
  else
  {
    for (int i = 0; i < n; i++)
      s[4 * i + 3] = 1.0 / n;
  }
}

/* particle_filter.cpp:475-502 */
int orc_pf_resample_limit(const orc_pf* pf, int k)
{
  if (k <= 1)
    return pf->max_samples;
  const double kd = (double)k;
  const double b = 2 / (9 * (kd - 1));
  const double c = sqrt(2 / (9 * (kd - 1))) * pf->pop_z;
  const double x = 1 - b + c;
  const int n = (int)ceil((k - 1) / (2 * pf->pop_err) * x * x * x);
  if (n < pf->min_samples)
    return pf->min_samples;
  if (n > pf->max_samples)
    return pf->max_samples;
  return n;
}

/* particle_filter.cpp:170-220 */
int orc_pf_update_converged(const orc_pf* pf, const double* s, int n, float* percent)
{
  double mx = 0, my = 0;
  for (int i = 0; i < n; i++)
  {
    mx += s[4 * i];
    my += s[4 * i + 1];
  }
  mx /= n;
  my /= n;
  int inside = 0;
  for (int i = 0; i < n; i++)
    if (fabs(s[4 * i] - mx) <= pf->dist_threshold && fabs(s[4 * i + 1] - my) <= pf->dist_threshold)
      inside++;
  /* :206 float arithmetic, then widened to double for the comparison */
  const double pct = (float)inside / (float)n * 100;
  if (percent)
    *percent = (float)pct;
  return pct >= pf->convergence_threshold;
}

/* particle_filter.cpp:505-636 */
int orc_pf_cluster_stats(orc_kdtree* t, const double* s, int n, int max_clusters, int* c_count,
                         double* c_weight, double* c_mean, double* c_cov, double set_mean[3], double set_cov[5])
{
  orc_kdtree_cluster(t);
  double* cm = (double*)calloc((size_t)max_clusters * 4, sizeof(double));
  double* cc = (double*)calloc((size_t)max_clusters * 4, sizeof(double));
  for (int i = 0; i < max_clusters; i++)
  {
    c_count[i] = 0;
    c_weight[i] = 0;
  }
  int cluster_count = 0;
  double weight = 0.0, m[4] = { 0, 0, 0, 0 }, c[4] = { 0, 0, 0, 0 };
  for (int i = 0; i < n; i++)
  {
    const double* p = &s[4 * i];
    const double w = p[3];
    const int cidx = orc_kdtree_get_cluster(t, p);
    if (cidx < 0 || cidx >= max_clusters)
      continue; /* :574-576 (cidx<0 asserts in the reference) */
    if (cidx + 1 > cluster_count)
      cluster_count = cidx + 1;
    c_count[cidx] += 1;
    c_weight[cidx] += w;
    cm[4 * cidx + 0] += w * p[0];
    cm[4 * cidx + 1] += w * p[1];
    cm[4 * cidx + 2] += w * cos(p[2]);
    cm[4 * cidx + 3] += w * sin(p[2]);
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++)
        cc[4 * cidx + 2 * a + b] += w * p[a] * p[b];
    weight += w;
    m[0] += w * p[0];
    m[1] += w * p[1];
    m[2] += w * cos(p[2]);
    m[3] += w * sin(p[2]);
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++)
        c[2 * a + b] += w * p[a] * p[b];
  }
  for (int k = 0; k < cluster_count; k++)
  {
    double* mean = &c_mean[3 * k];
    mean[0] = cm[4 * k] / c_weight[k];
    mean[1] = cm[4 * k + 1] / c_weight[k];
    mean[2] = atan2(cm[4 * k + 3], cm[4 * k + 2]);
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++)
        c_cov[5 * k + 2 * a + b] = cc[4 * k + 2 * a + b] / c_weight[k] - mean[a] * mean[b];
    c_cov[5 * k + 4] = -2 * log(sqrt(cm[4 * k + 2] * cm[4 * k + 2] + cm[4 * k + 3] * cm[4 * k + 3]));
  }
  set_mean[0] = m[0] / weight;
  set_mean[1] = m[1] / weight;
  set_mean[2] = atan2(m[3], m[2]);
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++)
      set_cov[2 * a + b] = c[2 * a + b] / weight - set_mean[a] * set_mean[b];
  set_cov[4] = -2 * log(sqrt(m[2] * m[2] + m[3] * m[3]));
  free(cm);
  free(cc);
  return cluster_count;
}

/* first i with c[i] <= r < c[i+1] (particle_filter.cpp:394-398, linear scan in
 * the reference).  c is non-decreasing, so that i is (number of c[1..n] <= r)
 * whenever r < c[n]; otherwise the search misses. */
static int cdf_find(const double* c, int n, double r)
{
  if (!(r < c[n]) || !(c[0] <= r))
    return n;
  int lo = 0, hi = n; /* invariant: c[lo] <= r < c[hi] */
  while (hi - lo > 1)
  {
    const int mid = lo + (hi - lo) / 2;
    if (c[mid] <= r)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

/* Node2D::updateFreeSpaceIndices, node_2d.cpp:317-337 */
int orc_free_space_indices(const orc_map2d* m, double non_free_space_radius, int* ij_out, int capacity)
{
  int n = 0;
  for (int i = 0; i < m->size_x; i++)
    for (int j = 0; j < m->size_y; j++)
      if (m->cells[cell_index(m, i, j)] == -1 && (double)orc_map2d_distance(m, i, j) > non_free_space_radius)
      {
        if (ij_out && n < capacity)
        {
          ij_out[2 * n] = i;
          ij_out[2 * n + 1] = j;
        }
        n++;
      }
  return n;
}

/* Node::randomFreeSpacePose, node.cpp:823-845 (with OccupancyMap::convertMapToWorld, occupancy_map.cpp:75-88) */
void orc_random_free_space_pose(const orc_free_space* fs, uint64_t* rng, double pose[3])
{
  const unsigned int rand_index = (unsigned int)(orc_drand48(rng) * fs->n);
  const int i = fs->ij[2 * rand_index], j = fs->ij[2 * rand_index + 1];
  pose[0] = fs->origin_x + (i - fs->size_x / 2) * fs->resolution;
  pose[1] = fs->origin_y + (j - fs->size_y / 2) * fs->resolution;
  pose[2] = orc_drand48(rng) * 2 * M_PI - M_PI;
}

/* particle_filter.cpp:135-163 */
int orc_pf_init_with_free_space_poses(orc_pf* pf, const orc_free_space* fs, double* samples, int n, int* node_count)
{
  orc_kdtree* tree = orc_kdtree_new();
  for (int i = 0; i < n; i++)
  {
    double* s = &samples[4 * i];
    s[3] = 1.0 / pf->max_samples;
    orc_random_free_space_pose(fs, &pf->rng, s);
    orc_kdtree_insert(tree, s, s[3]);
  }
  pf->w_slow = pf->w_fast = 0.0;
  pf->converged = 0;
  const int leaf = orc_kdtree_leaf_count(tree);
  if (node_count)
    *node_count = orc_kdtree_node_count(tree);
  orc_kdtree_free(tree);
  return leaf;
}

/* particle_filter.cpp:105-132 + pdf_gaussian.cpp:52-70 */
int orc_pf_init_with_gaussian(orc_pf* pf, const double mean[3], const double cr[9], const double cd[3],
                              double* samples, int n, int* node_count)
{
  orc_kdtree* tree = orc_kdtree_new();
  for (int i = 0; i < n; i++)
  {
    double* s = &samples[4 * i];
    s[3] = 1.0 / pf->max_samples;
    double r[3];
    for (int k = 0; k < 3; k++)
      r[k] = orc_gaussian_draw(&pf->rng, cd[k]);
    for (int k = 0; k < 3; k++)
    {
      double v = mean[k];
      for (int j = 0; j < 3; j++)
        v += cr[3 * k + j] * r[j];
      s[k] = v;
    }
    orc_kdtree_insert(tree, s, s[3]);
  }
  pf->w_slow = pf->w_fast = 0.0;
  pf->converged = 0;
  const int leaf = orc_kdtree_leaf_count(tree);
  if (node_count)
    *node_count = orc_kdtree_node_count(tree);
  orc_kdtree_free(tree);
  return leaf;
}

/* particle_filter.cpp:356-420 */
static double resample_multinomial(orc_pf* pf, const double* a, int n_a, double w_diff, double* b, int* idx,
                                   orc_kdtree* tree, int* m_out, int* status)
{
  double* c = (double*)malloc(sizeof(double) * (n_a + 1));
  c[0] = 0.0;
  for (int i = 0; i < n_a; i++)
    c[i + 1] = c[i] + a[4 * i + 3];
  double total = 0;
  int m = 0;
  while (m < pf->max_samples)
  {
    double* out = &b[4 * m];
    if (orc_drand48(&pf->rng) < w_diff)
    {
      if (!pf->random_source || pf->random_source->n <= 0)
      {
        *status = 2; /* random_pose_fn_ is a node callback; no generator was handed to the oracle */
        break;
      }
      orc_random_free_space_pose(pf->random_source, &pf->rng, out); /* :385-388 */
      out[3] = 1.0;
      if (idx)
        idx[m] = -1;
    }
    else
    {
      const double r = orc_drand48(&pf->rng);
      int i = cdf_find(c, n_a, r);
      if (i >= n_a)
      {
        *status = 1; /* ROS_ASSERT(i < sample_count) */
        i = n_a - 1;
      }
      out[0] = a[4 * i];
      out[1] = a[4 * i + 1];
      out[2] = a[4 * i + 2];
      out[3] = 1.0;
      if (idx)
        idx[m] = i;
    }
    m++;
    total += 1.0;
    orc_kdtree_insert(tree, out, 1.0);
    if (m > orc_pf_resample_limit(pf, orc_kdtree_leaf_count(tree)))
      break;
  }
  free(c);
  *m_out = m;
  return total;
}

/* particle_filter.cpp:269-354 */
static double resample_systematic(orc_pf* pf, const double* a, int n_a, int prev_leaf_count, double w_diff,
                                  double* b, int* idx, orc_kdtree* tree, int* m_out, int* status)
{
  double* c = (double*)malloc(sizeof(double) * (n_a + 1));
  c[0] = 0.0;
  for (int i = 0; i < n_a; i++)
    c[i + 1] = c[i] + a[4 * i + 3];
  double total = 0;
  int new_count = orc_pf_resample_limit(pf, prev_leaf_count);
  if (w_diff > 0.0)
  {
    new_count = (int)(new_count * (1.0 + w_diff));
    if (new_count > pf->max_samples)
      new_count = pf->max_samples;
  }
  const int num_random = (int)(w_diff * new_count);
  const int num_sys = new_count - num_random;
  const double start = orc_drand48(&pf->rng);
  const double delta = 1.0 / num_sys;
  int ci;
  for (ci = 0; ci < n_a; ci++)
    if (c[ci] <= start && start < c[ci + 1])
      break;
  if (ci >= n_a)
    ci = 0; /* the reference would read c[n+1] once and then wrap to 0 */
  int i = 0;
  if (num_random > 0 && (!pf->random_source || pf->random_source->n <= 0))
  {
    *status = 2;
    free(c);
    *m_out = 0;
    return 0;
  }
  for (; i < num_random; ++i) /* :316-324 */
  {
    double* out = &b[4 * i];
    orc_random_free_space_pose(pf->random_source, &pf->rng, out);
    out[3] = 1.0;
    if (idx)
      idx[i] = -1;
    total += 1.0;
    orc_kdtree_insert(tree, out, 1.0);
  }
  double target = start;
  for (; i < new_count; ++i)
  {
    /* :329-336 cyclic forward walk; never terminates when no interval holds
     * the target (Appendix A16).  The oracle bounds it and reports a miss. */
    int guard = 0;
    while (!(c[ci] <= target && target < c[ci + 1]))
    {
      ci++;
      if (ci >= n_a)
        ci = 0;
      if (++guard > 2 * n_a + 2)
      {
        *status = 1;
        break;
      }
    }
    if (*status == 1)
      break;
    target += delta;
    if (target > 1.0)
      target -= 1.0;
    double* out = &b[4 * i];
    out[0] = a[4 * ci];
    out[1] = a[4 * ci + 1];
    out[2] = a[4 * ci + 2];
    out[3] = 1.0;
    if (idx)
      idx[i] = ci;
    total += 1.0;
    orc_kdtree_insert(tree, out, 1.0);
  }
  free(c);
  *m_out = i;
  return total;
}

/* particle_filter.cpp:423-471 */
void orc_pf_update_resample(orc_pf* pf, const double* set_a, int n_a, int prev_leaf_count, double* set_b,
                            int* idx_out, orc_resample_out* out)
{
  memset(out, 0, sizeof(*out));
  orc_kdtree* tree = orc_kdtree_new();
  double w_diff = 1.0 - pf->w_fast / pf->w_slow;
  if (w_diff < 0.0)
    w_diff = 0.0;
  out->w_diff = w_diff;
  int m = 0;
  double total;
  if (pf->resample_model == ORC_RESAMPLE_SYSTEMATIC)
    total = resample_systematic(pf, set_a, n_a, prev_leaf_count, w_diff, set_b, idx_out, tree, &m, &out->status);
  else
    total = resample_multinomial(pf, set_a, n_a, w_diff, set_b, idx_out, tree, &m, &out->status);
  if (w_diff > 0.0)
    pf->w_slow = pf->w_fast = 0.0;
  for (int i = 0; i < m; i++)
    set_b[4 * i + 3] /= total;
  out->sample_count = m;
  out->leaf_count = orc_kdtree_leaf_count(tree);
  out->node_count = orc_kdtree_node_count(tree);
  if (m > 0)
  {
    const int maxc = pf->max_samples;
    int* cnt = (int*)malloc(sizeof(int) * maxc);
    double* cw = (double*)malloc(sizeof(double) * maxc);
    double* cmn = (double*)malloc(sizeof(double) * 3 * maxc);
    double* ccv = (double*)malloc(sizeof(double) * 5 * maxc);
    double cov5[5];
    out->cluster_count = orc_pf_cluster_stats(tree, set_b, m, maxc, cnt, cw, cmn, ccv, out->mean, cov5);
    memcpy(out->cov, cov5, sizeof(double) * 4);
    out->cov_theta = cov5[4];
    free(cnt);
    free(cw);
    free(cmn);
    free(ccv);
    out->converged = orc_pf_update_converged(pf, set_b, m, &out->percent_converged);
    pf->converged = out->converged;
  }
  orc_kdtree_free(tree);
}

/* ------------------------------------------------------------------------- */
/* 3-D map + point-cloud scanner                                               */
/* ------------------------------------------------------------------------- */

/* octomap.cpp:98-109: no centre offset, origin at world zero */
void orc_map3d_world_to_map(const orc_map3d* m, const double w[3], int c[3])
{
  for (int d = 0; d < 3; d++)
    c[d] = (int)floor(w[d] / m->resolution + 0.5);
}

/* octomap.cpp:83-96 */
void orc_map3d_map_to_world(double resolution, const int c[3], double w[3])
{
  for (int d = 0; d < 3; d++)
    w[d] = c[d] * resolution;
}

static int pose_valid3(const orc_map3d* m, int i, int j)
{
  return i <= m->max_cells[0] && i >= m->min_cells[0] && j <= m->max_cells[1] && j >= m->min_cells[1];
}

/* octomap.cpp:336-350 (LUT created) */
double orc_map3d_distance(const orc_map3d* m, int i, int j, int k)
{
  if (!(pose_valid3(m, i, j) && k <= m->max_cells[2] && k >= m->min_cells[2]))
    return m->max_dist;
  const uint32_t col = (uint32_t)((j - m->min_cells[1]) * m->width + (i - m->min_cells[0]));
  const uint32_t start = m->pose_indices[col];
  const uint8_t ratio = m->distance_ratios[start + (uint32_t)(k - m->min_cells[2])];
  return ratio * (m->max_dist / 255);
}

/* octomap.cpp:152-333.  FIFO brushfire over 6-neighbours from the occupied
 * voxels (fed in lexicographic order through a max-priority queue on (i,j,k),
 * i.e. descending order -- :205-226), uint8 quantisation floor(d/max*255),
 * a column of num_z bytes allocated on first touch, column 0 being the shared
 * all-255 column. */
typedef struct
{
  int i, j, k, si, sj, sk;
} v3_cell;

static int cmp_ijk_desc(const void* pa, const void* pb)
{
  const int* a = (const int*)pa;
  const int* b = (const int*)pb;
  for (int d = 0; d < 3; d++)
    if (a[d] != b[d])
      return a[d] < b[d] ? 1 : -1;
  return 0;
}

typedef struct
{
  const int* mn;
  const int* mx;
  int width, num_z;
  double max_dist, ratio;
  uint32_t* pose_indices;
  uint8_t* ratios;
  size_t n_ratios, cap;
  int overflow;
} lut3_builder;

static double lut3_get(const lut3_builder* b, int i, int j, int k)
{
  const uint32_t col = (uint32_t)((j - b->mn[1]) * b->width + (i - b->mn[0]));
  return b->ratios[b->pose_indices[col] + (uint32_t)(k - b->mn[2])] * b->ratio;
}

static void lut3_set(lut3_builder* b, int i, int j, int k, double d)
{
  const uint32_t col = (uint32_t)((j - b->mn[1]) * b->width + (i - b->mn[0]));
  uint32_t start = b->pose_indices[col];
  if (start == 0)
  {
    start = (uint32_t)b->n_ratios;
    if (b->n_ratios + b->num_z > b->cap)
    {
      b->overflow = 1;
      return;
    }
    b->pose_indices[col] = start;
    memset(b->ratios + start, 255, b->num_z);
    b->n_ratios += b->num_z;
  }
  if (d > b->max_dist)
    d = b->max_dist;
  d = d / b->max_dist * 255;
  b->ratios[start + (uint32_t)(k - b->mn[2])] = (uint8_t)(int)floor(d);
}

size_t orc_map3d_build_lut(const int mn[3], const int mx[3], double resolution, double max_dist,
                           const int* occupied, size_t n_occ, uint32_t* pose_indices, uint8_t* ratios,
                           size_t ratios_cap)
{
  lut3_builder b;
  b.mn = mn;
  b.mx = mx;
  b.width = mx[0] - mn[0] + 1;
  b.num_z = mx[2] - mn[2] + 1;
  b.max_dist = max_dist;
  b.ratio = max_dist / 255;
  b.pose_indices = pose_indices;
  b.ratios = ratios;
  b.cap = ratios_cap;
  b.overflow = 0;
  const size_t num_poses = (size_t)b.width * (mx[1] - mn[1] + 1);
  memset(pose_indices, 0, sizeof(uint32_t) * num_poses);
  if ((size_t)b.num_z > ratios_cap)
    return 0;
  memset(ratios, 255, b.num_z);
  b.n_ratios = b.num_z;

  const int radius = (int)floor(max_dist / resolution);
  const int td = radius + 2;
  double* dtab = (double*)malloc(sizeof(double) * td * td * td);
  for (int a = 0; a < td; a++)
    for (int c = 0; c < td; c++)
      for (int e = 0; e < td; e++)
        dtab[(a * td + c) * td + e] = sqrt((double)(a * a + c * c + e * e)) * resolution;

  int* occ = (int*)malloc(sizeof(int) * 3 * (n_occ ? n_occ : 1));
  size_t kept = 0;
  for (size_t q = 0; q < n_occ; q++)
  {
    const int* v = &occupied[3 * q];
    if (v[0] < mn[0] || v[0] > mx[0] || v[1] < mn[1] || v[1] > mx[1] || v[2] < mn[2] || v[2] > mx[2])
      continue;
    lut3_set(&b, v[0], v[1], v[2], 0.0);
    memcpy(&occ[3 * kept++], v, sizeof(int) * 3);
  }
  qsort(occ, kept, sizeof(int) * 3, cmp_ijk_desc);

  size_t qcap = kept + 1024, head = 0, tail = 0;
  v3_cell* q = (v3_cell*)malloc(sizeof(v3_cell) * qcap);
  for (size_t s = 0; s < kept; s++)
  {
    v3_cell c = { occ[3 * s], occ[3 * s + 1], occ[3 * s + 2], occ[3 * s], occ[3 * s + 1], occ[3 * s + 2] };
    q[tail++] = c;
  }
  static const int sh[6][3] = { { -1, 0, 0 }, { 0, -1, 0 }, { 0, 0, -1 }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
  while (head < tail && !b.overflow)
  {
    const v3_cell cur = q[head];
    const int ok[6] = { cur.i > mn[0], cur.j > mn[1], cur.k > mn[2], cur.i < mx[0], cur.j < mx[1], cur.k < mx[2] };
    for (int s = 0; s < 6; s++)
    {
      if (!ok[s])
        continue;
      const int i = cur.i + sh[s][0], j = cur.j + sh[s][1], k = cur.k + sh[s][2];
      const int a = abs(i - cur.si), c = abs(j - cur.sj), e = abs(k - cur.sk);
      if (a >= td || c >= td || e >= td)
        continue;
      const double nd = dtab[(a * td + c) * td + e];
      const double od = lut3_get(&b, i, j, k);
      if (od - nd > b.ratio)
      {
        lut3_set(&b, i, j, k, nd);
        if (tail == qcap)
        {
          /* compact consumed prefix, then grow */
          memmove(q, q + head, sizeof(v3_cell) * (tail - head));
          tail -= head;
          head = 0;
          if (tail * 2 > qcap)
          {
            qcap *= 2;
            q = (v3_cell*)realloc(q, sizeof(v3_cell) * qcap);
          }
        }
        v3_cell c2 = { i, j, k, cur.si, cur.sj, cur.sk };
        q[tail++] = c2;
      }
    }
    head++;
  }
  free(q);
  free(occ);
  free(dtab);
  return b.overflow ? 0 : b.n_ratios;
}

/* point_cloud_scanner.cpp:231-248.  Third-party arithmetic (tf2 +
 * tf2_sensor_msgs, versions unpinned -- PARITY UNPINNED): the footprint->map
 * transform (yaw about z, translation (x, y, 0)) is composed with the
 * scanner->footprint transform in double precision; tf2_sensor_msgs then
 * narrows translation and quaternion to float, forms Eigen's float rotation
 * matrix and maps each float point as R*p + t in float.  Restated as: double
 * quaternion product q = q_yaw * q_s, t = R(q_yaw) * t_s + (x, y, 0); narrow;
 * Eigen toRotationMatrix formula in float; row-wise mul/add in float. */
static void cloud_affine(const orc_cloud* p, const double pose[3], float R[9], float T[3])
{
  const double h = pose[2] * 0.5;
  const double yq[4] = { 0.0, 0.0, sin(h), cos(h) }; /* x y z w */
  const double* s = p->tf_quat;
  double q[4];
  q[3] = yq[3] * s[3] - yq[0] * s[0] - yq[1] * s[1] - yq[2] * s[2];
  q[0] = yq[3] * s[0] + yq[0] * s[3] + yq[1] * s[2] - yq[2] * s[1];
  q[1] = yq[3] * s[1] + yq[1] * s[3] + yq[2] * s[0] - yq[0] * s[2];
  q[2] = yq[3] * s[2] + yq[2] * s[3] + yq[0] * s[1] - yq[1] * s[0];
  const double cy = cos(pose[2]), sy = sin(pose[2]);
  const double t[3] = { cy * p->tf_xyz[0] - sy * p->tf_xyz[1] + pose[0],
                        sy * p->tf_xyz[0] + cy * p->tf_xyz[1] + pose[1], p->tf_xyz[2] + 0.0 };
  const float x = (float)q[0], y = (float)q[1], z = (float)q[2], w = (float)q[3];
  const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
  T[0] = (float)t[0];
  T[1] = (float)t[1];
  T[2] = (float)t[2];
}

static double gompertz3(const orc_cloud* p, double v)
{
  v = v * p->input_scale + p->input_shift;
  v = p->gompertz_a * exp(-1.0 * p->gompertz_b * exp(-1.0 * p->gompertz_c * v));
  return v + p->output_shift;
}

/* point_cloud_scanner.cpp:106-229 */
double orc_cloud_apply(const orc_cloud* p, const orc_map3d* m, double* s, int n, const float* pts, int np,
                       long* stats)
{
  if (p->max_beams < 2)
    return 0.0;
  const double denom = 2 * p->sigma_hit * p->sigma_hit;
  const double rand_mult = 1.0 / m->max_dist; /* :140 */
  double total = 0.0;
  for (int j = 0; j < n; j++)
  {
    float R[9], T[3];
    cloud_affine(p, &s[4 * j], R, T);
    double acc = 1.0, sum = 0.0;
    int count = 0;
    for (int q = 0; q < np; q++)
    {
      const float px = pts[3 * q], py = pts[3 * q + 1], pz_ = pts[3 * q + 2];
      const float wx = ((R[0] * px + R[1] * py) + R[2] * pz_) + T[0];
      const float wy = ((R[3] * px + R[4] * py) + R[5] * pz_) + T[1];
      const float wz = ((R[6] * px + R[7] * py) + R[8] * pz_) + T[2];
      const double w3[3] = { wx, wy, wz };
      int c[3];
      orc_map3d_world_to_map(m, w3, c);
      const double z = orc_map3d_distance(m, c[0], c[1], c[2]);
      double pz = p->z_hit * exp(-(z * z) / denom);
      if (p->model == ORC_CLOUD_MODEL)
      {
        pz += p->z_rand * rand_mult;
        acc += pz * pz * pz;
      }
      else
      {
        pz += p->z_rand;
        sum += pz;
        count++;
      }
      if (stats)
        stats[0] += 1;
    }
    if (p->model == ORC_CLOUD_MODEL_GOMPERTZ)
      acc = gompertz3(p, sum / count); /* no zero-count guard (:197) */
    s[4 * j + 3] *= acc;
    total += s[4 * j + 3];
  }
  if (total > 0.0)
  {
    /* :205-229 off-map factor on the robot cell */
    double rv = 0.0;
    for (int j = 0; j < n; j++)
    {
      const double w3[3] = { s[4 * j], s[4 * j + 1], 0.0 };
      int c[3];
      orc_map3d_world_to_map(m, w3, c);
      if (!pose_valid3(m, c[0], c[1]))
        s[4 * j + 3] *= p->off_map_factor;
      rv += s[4 * j + 3];
    }
    total = rv;
  }
  return total;
}


/* ====================================================================================================
 * Message shaping either side of the hot path.  Restated from the node
 * sources; the tf2 pieces (third party, unpinned) from their published form.
 * ==================================================================================================== */

/* Node2D::updateLatestScanData, node_2d.cpp:531-560.  range_max_ is a double member assigned from a float
 * expression (std::min<float>), range_min likewise; short readings are mapped to max range (:552-555). */
void orc_wire_laserscan_to_planar(const float* scan_ranges, int range_count, float scan_range_min, float scan_range_max,
                                  double sensor_min_range, double sensor_max_range, double angle_min,
                                  double angle_increment, double* ranges_out, double* angles_out, double* range_max_out)
{
  double data_range_max; /* latest_scan_data_->range_max_ */
  if (sensor_max_range > 0.0) /* :535-538 */
  {
    const float cap = (float)sensor_max_range;
    const float m = (cap < scan_range_max) ? cap : scan_range_max; /* std::min(a, b) = b < a ? b : a, in float */
    data_range_max = m;
  }
  else
    data_range_max = scan_range_max;
  double range_min; /* :539-543 */
  if (sensor_min_range > 0.0)
  {
    const float lim = (float)sensor_min_range;
    const float m = (scan_range_min < lim) ? lim : scan_range_min; /* std::max(a, b) = a < b ? b : a, in float */
    range_min = m;
  }
  else
    range_min = scan_range_min;
  for (int i = 0; i < range_count; i++) /* :546-559 */
  {
    if (scan_ranges[i] <= range_min)
      ranges_out[i] = data_range_max;
    else
      ranges_out[i] = scan_ranges[i];
    angles_out[i] = angle_min + (i * angle_increment);
  }
  *range_max_out = data_range_max;
}

/* tf2::Quaternion::setRPY (tf2 LinearMath/Quaternion.h), the general form */
static void tf2_set_rpy(double roll, double pitch, double yaw, double q[4])
{
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}

/* tf2 operator*(const Quaternion&, const Quaternion&) */
static void tf2_quat_mul(const double a[4], const double b[4], double o[4])
{
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

/* tf2::getYaw -> tf2::impl::getYaw (tf2/impl/utils.h) */
static double tf2_get_yaw(const double q[4])
{
  const double sqx = q[0] * q[0], sqy = q[1] * q[1], sqz = q[2] * q[2], sqw = q[3] * q[3];
  const double sarg = -2 * (q[0] * q[2] - q[3] * q[1]) / (sqx + sqy + sqz + sqw);
  if (sarg <= -0.99999)
    return -2 * atan2(q[1], q[0]);
  if (sarg >= 0.99999)
    return 2 * atan2(q[1], q[0]);
  return atan2(2 * (q[0] * q[1] + q[3] * q[2]), sqw + sqx - sqy - sqz);
}

/* Node2D::getAngleStats, node_2d.cpp:497-529; tf2::doTransform on a Quaternion message multiplies by the
 * transform's rotation from the left (tf2_geometry_msgs). */
void orc_wire_scan_angle_stats(double scan_angle_min, double scan_angle_increment, const double q_base_scanner[4],
                               double* angle_min_out, double* angle_increment_out)
{
  double min_q[4], inc_q[4], tmin[4], tinc[4];
  tf2_set_rpy(0.0, 0.0, scan_angle_min, min_q);                          /* :504-505 */
  tf2_set_rpy(0.0, 0.0, scan_angle_min + scan_angle_increment, inc_q);   /* :506-507 */
  tf2_quat_mul(q_base_scanner, min_q, tmin);                             /* :515 */
  tf2_quat_mul(q_base_scanner, inc_q, tinc);                             /* :516 */
  *angle_min_out = tf2_get_yaw(tmin);                                    /* :527 */
  *angle_increment_out = tf2_get_yaw(tinc) - *angle_min_out;             /* :528 */
  *angle_increment_out = orc_normalize_angle(*angle_increment_out);      /* :530 */
}

/* Node2D::convertMap, node_2d.cpp:265-295 */
void orc_wire_convert_map(const int8_t* data, int width, int height, double msg_resolution, double origin_x,
                          double origin_y, int map_scale_up_factor, int32_t* cells_out, int size_out[2],
                          float origin_out[2], double* resolution_out)
{
  const double resolution = msg_resolution / map_scale_up_factor;
  size_out[0] = width * map_scale_up_factor;
  size_out[1] = height * map_scale_up_factor;
  const double x_origin = origin_x + (size_out[0] / 2) * resolution;
  const double y_origin = origin_y + (size_out[1] / 2) * resolution;
  origin_out[0] = (float)x_origin; /* pcl::PointXYZ(x_origin, y_origin, 0.0): float members */
  origin_out[1] = (float)y_origin;
  for (int y = 0; y < size_out[1]; y++)
  {
    int i = y * size_out[0];
    const int msg_row = (y / map_scale_up_factor) * width;
    for (int x = 0; x < size_out[0]; x++, i++)
    {
      const int msg_i = msg_row + x / map_scale_up_factor;
      if (data[msg_i] == 0)
        cells_out[i] = -1; /* CELL_FREE */
      else if (data[msg_i] == 100)
        cells_out[i] = 1;  /* CELL_OCCUPIED */
      else
        cells_out[i] = 0;  /* CELL_UNKNOWN */
    }
  }
  *resolution_out = resolution;
}

/* Node3D::updateLatestScanData, node_3d.cpp:467-480 */
int orc_wire_decimate_cloud(const float* points_xyz, int data_count, int max_beams, float* out_xyz)
{
  int step = (data_count - 1) / (max_beams - 1);
  if (step < 1)
    step = 1;
  int kept = 0;
  for (int i = 0; i < data_count; i += step)
  {
    memcpy(&out_xyz[3 * kept], &points_xyz[3 * i], 3 * sizeof(float));
    kept++;
  }
  return kept;
}

/* Node::publishParticleCloud, node.cpp:335-357: q.setRPY(0, 0, yaw); tf2::toMsg(Transform(q, (x, y, 0))) */
void orc_wire_pose_array(const double* samples, int sample_count, double* poses7_out)
{
  for (int i = 0; i < sample_count; i++)
  {
    double q[4];
    tf2_set_rpy(0.0, 0.0, samples[4 * i + 2], q);
    double* o = &poses7_out[7 * i];
    o[0] = samples[4 * i];
    o[1] = samples[4 * i + 1];
    o[2] = 0;
    o[3] = q[0];
    o[4] = q[1];
    o[5] = q[2];
    o[6] = q[3];
  }
}
