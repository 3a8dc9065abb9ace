// flatnav_construct -- build an index from an ann-benchmarks style .npy file and save it in flatnav's binary format.
// Same positional command line as the reference's tools/construct_npy.cpp:31-135
//   construct <quantize> <metric> <data> <M> <ef_construction> <build_num_threads> <outfile> [--device]
// (<quantize> must be 0: product quantisation is outside this build's scope).  --device builds on the GPU in batches
// (Index::addBatchDevice: same insertion rule, deterministic); default is the host builder like the reference.
// The .npy may hold float32 (any metric), uint8 or int8 rows; the index takes the file's element type.
#include <chrono>
#include <iostream>
#include <numeric>
#include <string>
#include <vector>

#include <flatnav/distances/InnerProductDistance.h>
#include <flatnav/distances/SquaredL2Distance.h>
#include <flatnav/index/Index.h>
#include <flatnav/util/NpyReader.h>

using flatnav::Index;
using flatnav::distances::InnerProductDistance;
using flatnav::distances::SquaredL2Distance;
using flatnav::util::DataType;

template <typename dist_t, typename element_t>
static int build(flatnav::util::NpyArray& file, DataType dt, int M, int efc, int threads, const std::string& out, bool device) {
  const int N = static_cast<int>(file.shape[0]), dim = static_cast<int>(file.shape[1]);
  Index<dist_t, int> index(dist_t::create(static_cast<size_t>(dim)), N, M, false, dt);
  index.setNumThreads(static_cast<uint32_t>(threads));
  std::vector<int> labels(static_cast<size_t>(N));
  std::iota(labels.begin(), labels.end(), 0);
  const auto t0 = std::chrono::steady_clock::now();
  if (device) index.template addBatchDevice<element_t>(file.data<element_t>(), labels, efc);
  else index.template addBatch<element_t>(file.data<element_t>(), labels, efc);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::clog << "Build time: " << ms << " milliseconds (" << (device ? "device" : "host") << " builder)" << std::endl;
  std::clog << "Saving index to: " << out << std::endl;
  index.saveIndex(out);
  return 0;
}

template <typename element_t, DataType dt>
static int dispatch(flatnav::util::NpyArray& file, int metric, int M, int efc, int threads, const std::string& out, bool device) {
  if (metric == 0) return build<SquaredL2Distance<dt>, element_t>(file, dt, M, efc, threads, out, device);
  if (metric == 1) return build<InnerProductDistance<dt>, element_t>(file, dt, M, efc, threads, out, device);
  std::cerr << "Invalid metric. Valid IDs are 0 (L2) and 1 (inner product)." << std::endl;
  return -1;
}

int main(int argc, char** argv) {
  if (argc < 8) {
    std::clog << "Usage:\nconstruct <quantize> <metric> <data> <M> <ef_construction> <build_num_threads> <outfile> [--device]\n"
                 "\t <quantize> int, must be 0 (no quantization)\n"
                 "\t <metric> int, 0 for L2, 1 for inner product (angular)\n"
                 "\t <data> npy file (2-D float32 / uint8 / int8)\n"
                 "\t <M>: int\n\t <ef_construction>: int\n\t <build_num_threads>: int\n"
                 "\t <outfile>: where to stash the index\n"
                 "\t --device: insert on the GPU in batches instead of on the host threads" << std::endl;
    return -1;
  }
  try {
    if (std::stoi(argv[1]) != 0) {
      std::cerr << "product quantization is not part of this build" << std::endl;
      return -1;
    }
    const int metric = std::stoi(argv[2]);
    flatnav::util::NpyArray file = flatnav::util::loadNpy(argv[3]);
    const int M = std::stoi(argv[4]), efc = std::stoi(argv[5]), threads = std::stoi(argv[6]);
    const std::string out = argv[7];
    const bool device = argc > 8 && std::string(argv[8]) == "--device";
    if (file.shape.size() != 2) return -1;
    std::clog << "Loading " << file.shape[1] << "-dimensional dataset with N = " << file.shape[0] << " (" << file.dtype << ")" << std::endl;
    if (file.dtype == "<f4") return dispatch<float, DataType::float32>(file, metric, M, efc, threads, out, device);
    if (file.dtype == "|u1") return dispatch<uint8_t, DataType::uint8>(file, metric, M, efc, threads, out, device);
    if (file.dtype == "|i1") return dispatch<int8_t, DataType::int8>(file, metric, M, efc, threads, out, device);
    std::cerr << "unsupported element type " << file.dtype << std::endl;
    return -1;
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
}
