// C++ test program for the header-only host API (include/flatnav/**) -- compiled and linked against libflatnav_hip.so
// by tests/test_cpp_api.py.  Shape of the reference's own C++ tests (include/flatnav/tests/test_serialization.cpp:36-176:
// build, save, load, identical search for the index types) plus the error contract of SURVEY.md 8b.
// Exit code 0 = every check passed.  `--no-gpu`: only the parts that do not search (runs in the CPU-only container).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include <flatnav/distances/InnerProductDistance.h>
#include <flatnav/distances/SquaredL2Distance.h>
#include <flatnav/index/Index.h>

using flatnav::Index;
using flatnav::distances::InnerProductDistance;
using flatnav::distances::SquaredL2Distance;
using flatnav::util::DataType;

static int failures = 0;
#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                         \
    }                                                                     \
  } while (0)

template <typename dist_t, typename element_t>
static void roundTrip(DataType dt, int lo, int hi, bool gpu, const std::string& path) {
  const int N = 1500, dim = 40, M = 12, NQ = 64, K = 7;
  std::mt19937 rng(5);
  std::uniform_int_distribution<int> val(lo, hi);
  std::vector<element_t> data(static_cast<size_t>(N) * dim), queries(static_cast<size_t>(NQ) * dim);
  for (auto& v : data) v = static_cast<element_t>(val(rng));
  for (auto& v : queries) v = static_cast<element_t>(val(rng));
  Index<dist_t, int> index(dist_t::create(dim), N, M, true, dt);
  std::vector<int> labels(N);
  for (int i = 0; i < N; ++i) labels[i] = 1000 + 3 * i;
  index.template addBatch<element_t>(data.data(), labels, 48);
  CHECK(index.currentNumNodes() == static_cast<size_t>(N) && index.maxEdgesPerNode() == static_cast<size_t>(M));
  CHECK(index.distanceComputations() > 0);
  index.saveIndex(path);
  auto loaded = Index<dist_t, int>::loadIndex(path);
  CHECK(loaded->currentNumNodes() == static_cast<size_t>(N) && loaded->dataDimension() == static_cast<size_t>(dim));
  CHECK(std::memcmp(loaded->rawIndexMemory(), index.rawIndexMemory(), index.getTotalIndexMemory()) == 0);
  bool threw = false;
  try {
    int extra = 1;
    index.add(data.data(), extra, 48, 100);  // full (Index.h:355-360 of the reference)
  } catch (const std::runtime_error&) {
    threw = true;
  }
  CHECK(threw);
  if (!gpu) return;
  std::vector<float> d1(NQ * K), d2(NQ * K);
  std::vector<int> l1(NQ * K), l2(NQ * K);
  std::vector<int32_t> c1(NQ), c2(NQ);
  index.searchBatch(queries.data(), NQ, K, 40, 100, d1.data(), l1.data(), c1.data());
  loaded->searchBatch(queries.data(), NQ, K, 40, 100, d2.data(), l2.data(), c2.data());
  CHECK(d1 == d2 && l1 == l2 && c1 == c2);  // reloaded index: bit-identical results
  for (int q = 0; q < NQ; ++q) {            // single-query API == row of the batch
    auto r = index.search(queries.data() + static_cast<size_t>(q) * dim, K, 40);
    CHECK(static_cast<int>(r.size()) == c1[q]);
    for (size_t j = 0; j < r.size(); ++j) CHECK(r[j].first == d1[q * K + j] && r[j].second == l1[q * K + j]);
    for (size_t j = 1; j < r.size(); ++j) CHECK(r[j - 1].first <= r[j].first);
    for (size_t j = 0; j < r.size(); ++j) CHECK((r[j].second - 1000) % 3 == 0);
  }
  threw = false;
  try {
    index.search(queries.data(), K, 40, 0);  // num_initializations <= 0 (Index.h:847-849)
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);
}

int main(int argc, char** argv) {
  const bool gpu = !(argc > 1 && std::strcmp(argv[1], "--no-gpu") == 0);
  const std::string dir = argc > 2 ? argv[2] : "/tmp";
  try {
    roundTrip<SquaredL2Distance<DataType::float32>, float>(DataType::float32, 0, 255, gpu, dir + "/l2f.bin");
    roundTrip<InnerProductDistance<DataType::float32>, float>(DataType::float32, 0, 15, gpu, dir + "/ipf.bin");
    roundTrip<SquaredL2Distance<DataType::uint8>, uint8_t>(DataType::uint8, 0, 255, gpu, dir + "/l2u.bin");
    roundTrip<InnerProductDistance<DataType::uint8>, uint8_t>(DataType::uint8, 0, 31, gpu, dir + "/ipu.bin");
    roundTrip<SquaredL2Distance<DataType::int8>, int8_t>(DataType::int8, -128, 127, gpu, dir + "/l2i.bin");
    roundTrip<InnerProductDistance<DataType::int8>, int8_t>(DataType::int8, -30, 30, gpu, dir + "/ipi.bin");
    bool threw = false;
    try {
      Index<SquaredL2Distance<DataType::float32>, int>::loadIndex(dir + "/does-not-exist.bin");
    } catch (const std::runtime_error&) {
      threw = true;
    }
    CHECK(threw);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "unexpected exception: %s\n", e.what());
    return 2;
  }
  std::printf("%s: %d failure(s)\n", gpu ? "cpp api test (gpu)" : "cpp api test (no gpu)", failures);
  return failures ? 1 : 0;
}
