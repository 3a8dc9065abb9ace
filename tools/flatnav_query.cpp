// flatnav_query -- load a saved index, answer a query file on the GPU, report recall and time per query.
// Same positional command line and output lines as the reference's tools/query_npy.cpp:25-160
//   query <space> <index> <queries> <gtruth> <ef_search,ef_search,...> <k> <reorder> <quantized> [--per-query] [--dtype f32|u8|i8]
// <space> 0 = L2, 1 = inner product; <quantized> must be 0.  Recall as in query_npy.cpp:53-62 (|top-k found ∩ first k
// ground-truth ids| / k).  Default: every ef value is ONE batched GPU launch over all queries ("Duration" = batch wall
// time / queries, host buffers, PCIe included); --per-query times the reference's protocol instead (a loop of
// single-query calls).  The element type of a .bin file is not stored with the metric: pass --dtype for non-float indexes.
#include <chrono>
#include <iostream>
#include <sstream>
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
static int run(const std::string& index_file, flatnav::util::NpyArray& qfile, const std::vector<int>& gt, size_t n_gt,
               const std::vector<int>& efs, int K, bool reorder, bool per_query) {
  auto index = Index<dist_t, int>::loadIndex(index_file);
  std::cout << "[INFO] Index loaded" << std::endl;
  index->getIndexSummary();
  if (reorder) {
    std::clog << "[INFO] Gorder Reordering: " << std::endl;
    const auto t0 = std::chrono::steady_clock::now();
    index->reorderGOrder();
    std::clog << "Reordering time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()
              << " seconds" << std::endl;
  }
  const size_t nq = qfile.shape[0], dim = qfile.shape[1];
  std::vector<element_t> queries = qfile.as<element_t>();
  index->syncDevice();  // upload outside the timed region, like the reference's load
  std::vector<float> dist(nq * static_cast<size_t>(K));
  std::vector<int> labels(nq * static_cast<size_t>(K));
  std::vector<int32_t> counts(nq);
  for (int ef : efs) {
    const auto t0 = std::chrono::steady_clock::now();
    if (per_query) {
      for (size_t i = 0; i < nq; ++i) {
        auto r = index->search(queries.data() + i * dim, K, ef);
        counts[i] = static_cast<int32_t>(r.size());
        for (size_t j = 0; j < r.size(); ++j) {
          dist[i * K + j] = r[j].first;
          labels[i * K + j] = r[j].second;
        }
      }
    } else {
      index->searchBatch(queries.data(), nq, K, ef, 100, dist.data(), labels.data(), counts.data());
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    double mean_recall = 0;
    for (size_t i = 0; i < nq; ++i) {
      int hits = 0;
      for (int j = 0; j < counts[i]; ++j)
        for (int l = 0; l < K; ++l) hits += labels[i * K + j] == gt[i * n_gt + static_cast<size_t>(l)];
      mean_recall += static_cast<double>(hits) / K;
    }
    std::cout << "[INFO] Mean Recall: " << mean_recall / static_cast<double>(nq) << ", Duration:" << ms / static_cast<double>(nq)
              << " milliseconds" << " (ef_search " << ef << ", " << (per_query ? "per-query calls" : "one batched launch")
              << ", " << static_cast<double>(nq) / ms * 1e3 << " queries/s)" << std::endl;
  }
  return 0;
}

template <typename element_t, DataType dt>
static int dispatch(int space, const std::string& index_file, flatnav::util::NpyArray& q, const std::vector<int>& gt, size_t n_gt,
                    const std::vector<int>& efs, int K, bool reorder, bool per_query) {
  if (space == 0) return run<SquaredL2Distance<dt>, element_t>(index_file, q, gt, n_gt, efs, K, reorder, per_query);
  if (space == 1) return run<InnerProductDistance<dt>, element_t>(index_file, q, gt, n_gt, efs, K, reorder, per_query);
  throw std::invalid_argument("Invalid space ID. Valid IDs are 0 and 1.");
}

int main(int argc, char** argv) {
  if (argc < 9) {
    std::clog << "Usage:\nquery <space> <index> <queries> <gtruth> <ef_search> <k> <Reorder ID> <Quantized> [--per-query] [--dtype f32|u8|i8]\n"
                 "\t <space>: 0 = L2, 1 = inner product\n\t <index>: .bin file written by construct / index.save()\n"
                 "\t <queries> <gtruth>: .npy files (rows of the index element type / integer ids)\n"
                 "\t <ef_search>: int,int,int,...\n\t <k>: number of neighbors\n"
                 "\t <Reorder ID>: 0 for no reordering, 1 for gorder\n\t <Quantized>: must be 0" << std::endl;
    return -1;
  }
  try {
    const int space = std::stoi(argv[1]);
    const std::string index_file = argv[2];
    std::vector<int> efs;
    std::stringstream ss(argv[5]);
    for (int e; ss >> e;) {
      efs.push_back(e);
      if (ss.peek() == ',') ss.ignore();
    }
    const int K = std::stoi(argv[6]);
    const bool reorder = std::stoi(argv[7]) != 0;
    if (std::stoi(argv[8]) != 0) {
      std::cerr << "product quantization is not part of this build" << std::endl;
      return -1;
    }
    bool per_query = false;
    std::string dtype = "f32";
    for (int i = 9; i < argc; ++i) {
      if (std::string(argv[i]) == "--per-query") per_query = true;
      if (std::string(argv[i]) == "--dtype" && i + 1 < argc) dtype = argv[++i];
    }
    flatnav::util::NpyArray q = flatnav::util::loadNpy(argv[3]);
    flatnav::util::NpyArray g = flatnav::util::loadNpy(argv[4]);
    if (q.shape.size() != 2 || g.shape.size() != 2) return -1;
    if (static_cast<size_t>(K) > g.shape[1]) {
      std::cerr << "K is larger than the number of precomputed ground truth neighbors" << std::endl;
      return -1;
    }
    std::clog << "Loading " << q.shape[0] << " queries" << std::endl;
    const std::vector<int> gt = g.as<int>();
    if (dtype == "f32") return dispatch<float, DataType::float32>(space, index_file, q, gt, g.shape[1], efs, K, reorder, per_query);
    if (dtype == "u8") return dispatch<uint8_t, DataType::uint8>(space, index_file, q, gt, g.shape[1], efs, K, reorder, per_query);
    if (dtype == "i8") return dispatch<int8_t, DataType::int8>(space, index_file, q, gt, g.shape[1], efs, K, reorder, per_query);
    std::cerr << "unknown --dtype " << dtype << std::endl;
    return -1;
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
}
