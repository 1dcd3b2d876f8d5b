// C-ABI: wire formats.
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
