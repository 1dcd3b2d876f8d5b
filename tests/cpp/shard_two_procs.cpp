// A sharded filter brought up and driven from plain C++ processes -- no Python, no MPI, no launcher between them:
// this program forks one process per rank (all of them on GPU 0 of the box) plus one that runs the same filter
// unsharded, each rank calls bpf_shard_bootstrap on 127.0.0.1:<port> (TCP rendez-vous inside the library, IPC handles
// over it, mailbox connect + self-test), then three cycles of bpf_shard_update_sensor_planar /
// bpf_shard_update_resample.  Every process writes what it ends up with; the Python test compares.
//
// usage: shard_two_procs cells lut samples ranges angles size world port flags out_prefix
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "badger_pf.h"

template <typename T>
static std::vector<T> slurp(const char* path)
{
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<T> v(n / sizeof(T));
  if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) std::exit(2);
  std::fclose(f);
  return v;
}

#define CHECK(e, call)                                                                                          \
  do                                                                                                            \
  {                                                                                                             \
    const int _rc = (call);                                                                                     \
    if (_rc != BPF_OK)                                                                                          \
    {                                                                                                           \
      std::fprintf(stderr, "rank %d: %s -> %d (%s)\n", rank, #call, _rc, (e) ? bpf_last_error_message(e) : ""); \
      return 10 + _rc;                                                                                          \
    }                                                                                                           \
  } while (0)

struct Inputs
{
  std::vector<int32_t> cells;
  std::vector<float> lut;
  std::vector<double> samples, ranges, angles;
  int size;
};

static int setup(bpf_engine* e, const Inputs& in, int rank, int n_global)
{
  const float origin = (float)((in.size / 2) * 0.05);
  CHECK(e, bpf_map2d_set(e, in.cells.data(), in.lut.data(), in.size, in.size, origin, origin, 0.05, 2.0));
  CHECK(e, bpf_planar_init(e, (int)in.ranges.size()));
  CHECK(e, bpf_planar_set_model_likelihood_field(e, 0.95, 0.05, 0.2, 2.0));
  CHECK(e, bpf_planar_set_map_factors(e, 0.95, 0.95, 0.3));
  const double pose[3] = { 0.1, -0.05, 0.2 };
  CHECK(e, bpf_planar_set_scanner_pose(e, pose));
  CHECK(e, bpf_pf_create(e, 100, n_global, 0.0, 0.0, 85.0));  // the GLOBAL bounds on every rank
  CHECK(e, bpf_pf_srand48(e, 42));
  return 0;
}

static void dump(const std::string& path, const std::vector<double>& v, int count)
{
  FILE* f = std::fopen(path.c_str(), "wb");
  std::fwrite(v.data(), sizeof(double), (size_t)count * 4, f);
  std::fclose(f);
}

static int run_rank(const Inputs& in, int rank, int world, int port, int flags, const std::string& prefix)
{
  const int n_global = (int)in.samples.size() / 4;
  bpf_engine* e = nullptr;
  CHECK(e, bpf_create(0, &e));
  if (int rc = setup(e, in, rank, n_global))
    return rc;
  const int lo = (int)((long long)n_global * rank / world), hi = (int)((long long)n_global * (rank + 1) / world);
  CHECK(e, bpf_pf_set_samples(e, in.samples.data() + 4 * (size_t)lo, hi - lo, 1));
  const std::string addr = "127.0.0.1:" + std::to_string(port);
  int mode = 0;
  CHECK(e, bpf_shard_bootstrap(e, rank, world, addr.c_str(), n_global, flags, &mode));
  int global = n_global, leaf = 1, bins = 0, windows = 0, hint = 4096, miss = 0;
  std::vector<double> local((size_t)n_global * 4);
  for (int cycle = 0; cycle < 3; ++cycle)
  {
    CHECK(e, bpf_shard_update_sensor_planar(e, in.ranges.data(), in.angles.data(), (int)in.ranges.size(), 30.0, global));
    CHECK(e, bpf_shard_update_resample(e, &global, &leaf, &bins, &windows, &hint, &miss));
    int got = 0;
    CHECK(e, bpf_pf_get_samples(e, local.data(), n_global, &got));
    uint64_t rng = 0;
    CHECK(e, bpf_pf_get_rng_state(e, &rng));
    dump(prefix + ".rank" + std::to_string(rank) + ".cycle" + std::to_string(cycle) + ".bin", local, got);
    std::printf("rank %d cycle %d mode %d M %d leaf %d bins %d windows %d local %d rng %llu miss %d\n", rank, cycle, mode,
                global, leaf, bins, windows, got, (unsigned long long)rng, miss);
    std::fflush(stdout);
  }
  CHECK(e, bpf_shard_shutdown(e));
  bpf_destroy(e);
  return 0;
}

// the same filter on one engine through the ordinary entry points
static int run_unsharded(const Inputs& in, const std::string& prefix)
{
  const int rank = -1;
  const int n = (int)in.samples.size() / 4;
  bpf_engine* e = nullptr;
  CHECK(e, bpf_create(0, &e));
  if (int rc = setup(e, in, rank, n))
    return rc;
  CHECK(e, bpf_pf_set_samples(e, in.samples.data(), n, 1));
  std::vector<double> all((size_t)n * 4);
  for (int cycle = 0; cycle < 3; ++cycle)
  {
    CHECK(e, bpf_pf_update_sensor_planar(e, in.ranges.data(), in.angles.data(), (int)in.ranges.size(), 30.0));
    CHECK(e, bpf_pf_update_resample(e));
    bpf_pf_state st;
    CHECK(e, bpf_pf_get_state(e, &st));
    int got = 0;
    CHECK(e, bpf_pf_get_samples(e, all.data(), n, &got));
    uint64_t rng = 0;
    CHECK(e, bpf_pf_get_rng_state(e, &rng));
    dump(prefix + ".single.cycle" + std::to_string(cycle) + ".bin", all, got);
    std::printf("single cycle %d M %d leaf %d bins %d rng %llu\n", cycle, st.sample_count, st.leaf_count, st.bin_count,
                (unsigned long long)rng);
    std::fflush(stdout);
  }
  bpf_destroy(e);
  return 0;
}

int main(int argc, char** argv)
{
  if (argc < 11)
  {
    std::fprintf(stderr, "usage: cells lut samples ranges angles size world port flags out_prefix\n");
    return 2;
  }
  Inputs in;
  in.cells = slurp<int32_t>(argv[1]);
  in.lut = slurp<float>(argv[2]);
  in.samples = slurp<double>(argv[3]);
  in.ranges = slurp<double>(argv[4]);
  in.angles = slurp<double>(argv[5]);
  in.size = std::atoi(argv[6]);
  const int world = std::atoi(argv[7]), port = std::atoi(argv[8]), flags = std::atoi(argv[9]);
  const std::string prefix = argv[10];
  // fork BEFORE anything touches the GPU: every child initialises HIP for itself
  std::vector<pid_t> kids;
  for (int r = -1; r < world; ++r)
  {
    const pid_t pid = fork();
    if (pid == 0)
      _exit(r < 0 ? run_unsharded(in, prefix) : run_rank(in, r, world, port, flags, prefix));
    kids.push_back(pid);
  }
  int worst = 0;
  for (pid_t pid : kids)
  {
    int status = 0;
    waitpid(pid, &status, 0);
    const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 99;
    if (code != 0)
      worst = code;
  }
  return worst;
}
