// Drives the hot path through the C++ adapter in the reference's call order and prints the
// results for the Python test to compare with the oracle.  Input arrays come from binary files
// written by the test (so both sides see identical bits).
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "badger_amcl_amd/adapter.hpp"

using namespace badger_amcl_amd;

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

int main(int argc, char** argv)
{
  if (argc < 9) { std::fprintf(stderr, "usage: cells lut|- samples ranges angles size out_weights out_resampled [out_lut]\n"); return 2; }
  auto cells = slurp<int32_t>(argv[1]);
  // "-": no LUT is handed over; the reference-named calls build it (setModelLikelihoodField -> updateDistancesLUT,
  // planar_scanner.cpp:74), which must give the reference's brushfire values
  const bool own_lut = std::string(argv[2]) != "-";
  std::vector<float> lut;
  if (own_lut)
    lut = slurp<float>(argv[2]);
  auto smp = slurp<double>(argv[3]);
  auto ranges = slurp<double>(argv[4]);
  auto angles = slurp<double>(argv[5]);
  const int size = std::atoi(argv[6]);
  const int n = (int)smp.size() / 4;

  auto eng = std::make_shared<Engine>(0);
  auto map = std::make_shared<OccupancyMap>(eng, 0.05);
  map->setSize({ size, size });
  const float origin = (float)((size / 2) * 0.05);
  map->setOrigin(origin, origin);
  for (int i = 0; i < size * size; ++i)
    map->setCellState(i, (MapCellState)cells[i]);
  if (own_lut)
    map->setDistancesLUT(lut, 2.0);

  auto scanner = std::make_shared<PlanarScanner>(eng);
  scanner->init((int)ranges.size(), map);
  scanner->setModelLikelihoodField(0.95, 0.05, 0.2, 2.0);
  scanner->setMapFactors(0.95, 0.95, 0.3);
  scanner->setPlanarScannerPose({ 0.1, -0.05, 0.2 });
  if (argc > 9)
  {
    auto built = map->getDistancesLUT();
    FILE* fl = std::fopen(argv[9], "wb");
    std::fwrite(built.data(), sizeof(float), built.size(), fl);
    std::fclose(fl);
  }

  auto pf = std::make_shared<ParticleFilter>(eng, 100, n, 0.0, 0.0, 85.0);
  pf->srand48(42);
  std::vector<PFSample> init(n);
  for (int i = 0; i < n; ++i)
    init[i] = PFSample{ { smp[4 * i], smp[4 * i + 1], smp[4 * i + 2] }, smp[4 * i + 3] };
  pf->initWithSamples(init);

  auto data = std::make_shared<PlanarData>();
  data->range_count_ = (int)ranges.size();
  data->range_max_ = 30.0;
  data->ranges_ = ranges;
  data->angles_ = angles;

  if (argc > 10)
  {
    // INTEGRATION Option 1: the set stays in a host vector that is allocated once (the reference's
    // ParticleFilter, particle_filter.cpp:62-89), pinned once, and scored where it lies
    auto host_set = std::make_shared<PFSampleSet>();
    host_set->samples = init;
    host_set->sample_count = n;
    PinnedSamples pin(eng, host_set->samples);
    const double total = scanner->applyModelToSampleSet(data, host_set);
    int chunks = 0, pinned = 0;
    bpf_seam_last_plan(eng->get(), &chunks, &pinned);
    FILE* fh = std::fopen(argv[10], "wb");
    std::fwrite(host_set->samples.data(), sizeof(PFSample), host_set->samples.size(), fh);
    std::fclose(fh);
    std::fprintf(stderr, "host set: total %.17g chunks %d pinned %d\n", total, chunks, pinned);
  }

  if (!scanner->updateSensor(pf, data)) return 3;
  auto set = pf->getCurrentSet();
  FILE* f = std::fopen(argv[7], "wb");
  std::fwrite(set->samples.data(), sizeof(PFSample), set->samples.size(), f);
  std::fclose(f);
  pf->updateResample();
  auto set2 = pf->getCurrentSet();
  f = std::fopen(argv[8], "wb");
  std::fwrite(set2->samples.data(), sizeof(PFSample), set2->samples.size(), f);
  std::fclose(f);
  std::printf("%d %d %d\n", set2->sample_count, set2->leaf_count, set2->converged);
  return 0;
}
