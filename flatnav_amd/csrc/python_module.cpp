// python_module.cpp -- pybind11 extension `flatnav_amd._core` (own implementation).
//
// Python surface of the reference's `flatnav._core` (python-bindings/src/flatnav/bindings.cpp:
// 426-539): submodules `index` (create(), IndexL2Float / IndexIPFloat / IndexL2Uint8 /
// IndexIPUint8 / IndexL2Int8 / IndexIPInt8) and `data_type` (DataType), enum MetricType,
// __version__.  Methods, argument names, defaults, returned dtypes/shapes and raised exception
// types follow the reference; `search` hands the whole batch to the GPU in one call
// (flatnav::Index::searchBatch -> C ABI fnv_search_batch) instead of looping on the host.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include <flatnav/distances/InnerProductDistance.h>
#include <flatnav/distances/SquaredL2Distance.h>
#include <flatnav/index/Index.h>

namespace py = pybind11;
using flatnav::Index;
using flatnav::distances::InnerProductDistance;
using flatnav::distances::MetricType;
using flatnav::distances::SquaredL2Distance;
using flatnav::util::DataType;

namespace {

template <typename T>
using dense_array = py::array_t<T, py::array::c_style | py::array::forcecast>;

// Index wrapper exposed to Python.  dist_t fixes metric and element type; labels are int32.
template <typename dist_t, DataType kType>
class PyIndex : public std::enable_shared_from_this<PyIndex<dist_t, kType>> {
  using element_t = typename flatnav::util::type_for_data_type<kType>::type;
  using index_t = Index<dist_t, int>;

  int _dim;
  int _next_label = 0;
  std::unique_ptr<index_t> _index;

 public:
  explicit PyIndex(std::unique_ptr<index_t> loaded) : _dim(static_cast<int>(loaded->dataDimension())), _index(std::move(loaded)) {}

  PyIndex(int dim, int dataset_size, int max_edges_per_node, bool verbose, bool collect_stats)
      : _dim(dim),
        _index(new index_t(dist_t::create(static_cast<size_t>(dim)), dataset_size, max_edges_per_node, collect_stats, kType)) {
    if (verbose) {
      const double gb = static_cast<double>(_index->getTotalIndexMemory() + _index->mutexesAllocatedMemory()) / 1e9;
      py::print("Total allocated index memory:", gb, "GB");
      _index->getIndexSummary();
    }
  }

  // add(data, ef_construction, num_initializations=100, labels=None)   [bindings.cpp:64-119, 326-335]
  // device=True (extension): the insertions' beam searches run on the GPU in batches (Index::addBatchDevice).
  void add(const py::array& data_any, int ef_construction, int num_initializations, py::object labels, bool device,
           uint32_t device_max_batch, bool device_wiring, uint32_t device_bootstrap) {
    dense_array<element_t> data = data_any.cast<dense_array<element_t>>();
    if (data.ndim() != 2 || data.shape(1) != _dim)
      throw std::invalid_argument("Data has incorrect dimensions. data.ndim() = `" + std::to_string(data.ndim()) +
                                  "`. Expected 2D array with dimensions (num_vectors, dim).");
    const size_t count = static_cast<size_t>(data.shape(0));
    std::vector<int> ids;
    if (labels.is_none()) {
      ids.resize(count);
      std::iota(ids.begin(), ids.end(), 0);
    } else {
      try {
        ids = py::cast<std::vector<int>>(labels);
      } catch (const py::cast_error&) {
        throw std::invalid_argument("Invalid labels provided.");
      }
      if (ids.size() != count) throw std::invalid_argument("Incorrect number of labels.");
    }
    void* raw = const_cast<element_t*>(data.data());
    py::gil_scoped_release release;  // the builder spawns its own threads
    if (device) {
      typename index_t::DeviceBuildOptions opt;
      if (device_max_batch) opt.max_batch = device_max_batch;
      if (device_bootstrap) opt.bootstrap = device_bootstrap;
      opt.wire_on_device = device_wiring;
      _index->template addBatchDevice<element_t>(raw, ids, ef_construction, num_initializations, opt);
    } else {
      _index->template addBatch<element_t>(raw, ids, ef_construction, num_initializations);
    }
  }

  // allocate_nodes(data) -> self   [bindings.cpp:308-324]: vectors only, no edges (used before
  // build_graph_links); labels continue from the wrapper's own counter.
  std::shared_ptr<PyIndex> allocateNodes(const dense_array<float>& data) {
    if (data.ndim() != 2 || data.shape(1) != _dim) throw std::invalid_argument("Data has incorrect dimensions.");
    for (py::ssize_t row = 0; row < data.shape(0); ++row) {
      uint32_t node;
      int label = _next_label++;
      _index->allocateNode(const_cast<float*>(data.data(row)), label, node);
    }
    return this->shared_from_this();
  }

  // search(queries, K, ef_search, num_initializations=100) -> (float32[Q,K], int32[Q,K])
  py::tuple search(const py::array& queries_any, int K, int ef_search, int num_initializations) {
    dense_array<element_t> queries = queries_any.cast<dense_array<element_t>>();
    if (queries.ndim() != 2 || queries.shape(1) != _dim) throw std::invalid_argument("Queries have incorrect dimensions.");
    if (K <= 0) throw std::invalid_argument("K must be positive.");
    const py::ssize_t nq = queries.shape(0);
    py::array_t<float> dist({nq, static_cast<py::ssize_t>(K)});
    py::array_t<int> labels({nq, static_cast<py::ssize_t>(K)});
    std::vector<int32_t> counts(static_cast<size_t>(nq));
    {
      const element_t* qptr = queries.data();
      float* dptr = dist.mutable_data();
      int* lptr = labels.mutable_data();
      py::gil_scoped_release release;  // the GPU works; other Python threads may run
      _index->searchBatch(qptr, static_cast<uint64_t>(nq), K, ef_search, num_initializations, dptr, lptr, counts.data());
    }
    for (py::ssize_t q = 0; q < nq; ++q)  // bindings.cpp:184-189
      if (counts[static_cast<size_t>(q)] != K)
        throw std::runtime_error("Search did not return the expected number of results. Expected " + std::to_string(K) +
                                 " but got " + std::to_string(counts[static_cast<size_t>(q)]) + ".");
    return py::make_tuple(dist, labels);
  }

  // search_single(query, K, ef_search, num_initializations=100) -> (float32[K], int32[K])
  py::tuple searchSingle(const py::array& query_any, int K, int ef_search, int num_initializations) {
    dense_array<element_t> query = query_any.cast<dense_array<element_t>>();
    if (query.ndim() != 1 || query.shape(0) != _dim) throw std::invalid_argument("Query has incorrect dimensions.");
    auto top = _index->search(query.data(), K, ef_search, num_initializations);
    if (static_cast<int>(top.size()) != K)  // bindings.cpp:134-137
      throw std::runtime_error("Search did not return the expected number of results. Expected " + std::to_string(K) +
                               " but got " + std::to_string(top.size()) + ".");
    py::array_t<float> dist(static_cast<py::ssize_t>(K));
    py::array_t<int> labels(static_cast<py::ssize_t>(K));
    for (int i = 0; i < K; ++i) {
      dist.mutable_data()[i] = top[static_cast<size_t>(i)].first;
      labels.mutable_data()[i] = top[static_cast<size_t>(i)].second;
    }
    return py::make_tuple(dist, labels);
  }

  uint64_t getQueryDistanceComputations() {  // returns AND resets (bindings.cpp:270-274)
    const uint64_t v = _index->distanceComputations();
    _index->resetStats();
    return v;
  }
  void buildGraphLinks(const std::string& mtx_filename) { _index->buildGraphLinks(mtx_filename); }
  std::vector<std::vector<uint32_t>> getGraphOutdegreeTable() { return _index->getGraphOutdegreeTable(); }
  uint32_t getMaxEdgesPerNode() { return static_cast<uint32_t>(_index->maxEdgesPerNode()); }
  void reorder(const std::vector<std::string>& strategies) {
    for (const auto& s : strategies) {
      std::string lower = s;
      std::transform(lower.begin(), lower.end(), lower.begin(), [](unsigned char c) { return std::tolower(c); });
      if (lower != "gorder" && lower != "rcm")
        throw std::invalid_argument("`" + s + "` is not a supported graph re-ordering strategy.");
    }
    _index->doGraphReordering(strategies);
  }
  void setNumThreads(uint32_t n) { _index->setNumThreads(n); }
  uint32_t getNumThreads() { return _index->getNumThreads(); }
  void save(const std::string& filename) { _index->saveIndex(filename); }
  static std::shared_ptr<PyIndex> loadIndex(const std::string& filename) {
    return std::make_shared<PyIndex>(index_t::loadIndex(filename));
  }

  // ---- additions for the GPU build (not in the reference) -------------------------------------
  void setDevice(int ordinal) { _index->setDevice(ordinal); }
  void setDevices(const std::vector<int>& ordinals) { _index->setDevices(ordinals); }
  std::vector<int> devices() { return _index->devices(); }
  void syncDevice() { _index->syncDevice(); }
  uintptr_t deviceHandle() { return reinterpret_cast<uintptr_t>(_index->deviceHandle()); }
  uint64_t currentNumNodes() { return _index->currentNumNodes(); }
  // zero-copy view of the AoS node store (for tests: graph-equality checks against the oracle)
  py::array_t<uint8_t> rawBlob() {
    const uint64_t n = _index->getTotalIndexMemory();
    return py::array_t<uint8_t>({static_cast<py::ssize_t>(n)}, {1},
                                reinterpret_cast<const uint8_t*>(_index->rawIndexMemory()), py::cast(this->shared_from_this()));
  }
  size_t nodeSizeBytes() { return _index->nodeSizeBytes(); }
  size_t dataSizeBytes() { return _index->dataSizeBytes(); }
};

template <typename dist_t, DataType kType>
void bindIndex(py::module_& m, const char* name) {
  using T = PyIndex<dist_t, kType>;
  py::class_<T, std::shared_ptr<T>>(m, name)
      .def("add", &T::add, py::arg("data"), py::arg("ef_construction"), py::arg("num_initializations") = 100,
           py::arg("labels") = py::none(), py::kw_only(), py::arg("device") = false, py::arg("device_max_batch") = 0,
           py::arg("device_wiring") = true, py::arg("device_bootstrap") = 0,
           "Insert vectors (rows of `data`, cast to the index data type) into the graph.  device=True: the insertions "
           "run on the GPU in batches (deterministic; same graph family as the host builder; device_max_batch=1 "
           "inserts one node at a time and reproduces the single-threaded host / reference graph byte for byte on "
           "data whose distances are exact).  device_bootstrap: nodes inserted on the host before the first batch "
           "(default 2048).")
      .def("allocate_nodes", &T::allocateNodes, py::arg("data"),
           "Store vectors without creating edges (follow with build_graph_links).")
      .def("search_single", &T::searchSingle, py::arg("query"), py::arg("K"), py::arg("ef_search"),
           py::arg("num_initializations") = 100, "k-NN of one query on the GPU -> (distances[K], labels[K]).")
      .def("search", &T::search, py::arg("queries"), py::arg("K"), py::arg("ef_search"),
           py::arg("num_initializations") = 100,
           "Batched k-NN on the GPU (one kernel launch) -> (distances[Q,K] float32, labels[Q,K] int32).")
      .def("get_query_distance_computations", &T::getQueryDistanceComputations,
           "Distance evaluations since the last call (needs collect_stats=True); resets the counter.")
      .def("save", &T::save, py::arg("filename"), "Write the index in flatnav's binary format.")
      .def("build_graph_links", &T::buildGraphLinks, py::arg("mtx_filename"),
           "Import edges from a MatrixMarket file written by the hnswlib fork's save_base_layer_graph.")
      .def("get_graph_outdegree_table", &T::getGraphOutdegreeTable, "Out-neighbour lists of every node.")
      .def("reorder", &T::reorder, py::arg("strategies"), "Relabel nodes with 'gorder' and/or 'rcm'.")
      .def("set_num_threads", &T::setNumThreads, py::arg("num_threads"), "Host threads used by add().")
      .def_static("load_index", &T::loadIndex, py::arg("filename"), "Load an index written by save().")
      .def_property_readonly("max_edges_per_node", &T::getMaxEdgesPerNode)
      .def_property_readonly("num_threads", &T::getNumThreads)
      // GPU-build additions
      .def("set_device", &T::setDevice, py::arg("ordinal"), "Use this one GPU.")
      .def("set_devices", &T::setDevices, py::arg("ordinals"),
           "GPUs that each hold a replica of the index; batched searches are sharded over them (default: the "
           "FLATNAV_DEVICES environment variable -- 'all' = every visible GPU --, else the primary GPU only).")
      .def_property_readonly("devices", &T::devices)
      .def("sync_device", &T::syncDevice, "Upload pending changes to HBM now instead of at the next search.")
      .def("device_handle", &T::deviceHandle, "fnv_index_t of the device mirror as an integer.")
      .def("_raw_blob", &T::rawBlob)
      .def_property_readonly("_node_size_bytes", &T::nodeSizeBytes)
      .def_property_readonly("_data_size_bytes", &T::dataSizeBytes)
      .def_property_readonly("_cur_num_nodes", &T::currentNumNodes);
}

template <DataType kType>
py::object makeIndex(const std::string& distance_type, int dim, int dataset_size, int max_edges_per_node, bool verbose,
                     bool collect_stats) {
  std::string lower = distance_type;
  std::transform(lower.begin(), lower.end(), lower.begin(), [](unsigned char c) { return std::tolower(c); });
  if (lower != "l2" && lower != "angular")  // bindings.cpp:397-407
    throw std::invalid_argument("Invalid distance type: `" + lower +
                                "` during index construction. Valid options include `l2` and `angular`.");
  if (lower == "l2")
    return py::cast(std::make_shared<PyIndex<SquaredL2Distance<kType>, kType>>(dim, dataset_size, max_edges_per_node,
                                                                              verbose, collect_stats));
  return py::cast(std::make_shared<PyIndex<InnerProductDistance<kType>, kType>>(dim, dataset_size, max_edges_per_node,
                                                                                verbose, collect_stats));
}

}  // namespace

PYBIND11_MODULE(_core, m) {
  m.doc() = "flatnav_amd: MI355X-native flat navigable-small-world index (search on gfx950 through a C ABI)";
  m.attr("__version__") = "0.1.0";

  auto data_type = m.def_submodule("data_type");
  py::enum_<DataType>(data_type, "DataType")
      .value("float32", DataType::float32)
      .value("int8", DataType::int8)
      .value("uint8", DataType::uint8)
      .export_values();

  py::enum_<MetricType>(m, "MetricType").value("L2", MetricType::L2).value("IP", MetricType::IP);

  auto index = m.def_submodule("index");
  bindIndex<SquaredL2Distance<DataType::float32>, DataType::float32>(index, "IndexL2Float");
  bindIndex<SquaredL2Distance<DataType::int8>, DataType::int8>(index, "IndexL2Int8");
  bindIndex<SquaredL2Distance<DataType::uint8>, DataType::uint8>(index, "IndexL2Uint8");
  bindIndex<InnerProductDistance<DataType::float32>, DataType::float32>(index, "IndexIPFloat");
  bindIndex<InnerProductDistance<DataType::int8>, DataType::int8>(index, "IndexIPInt8");
  bindIndex<InnerProductDistance<DataType::uint8>, DataType::uint8>(index, "IndexIPUint8");

  index.def(
      "create",
      [](const std::string& distance_type, int dim, int dataset_size, int max_edges_per_node, DataType index_data_type,
         bool verbose, bool collect_stats) -> py::object {
        switch (index_data_type) {
          case DataType::float32:
            return makeIndex<DataType::float32>(distance_type, dim, dataset_size, max_edges_per_node, verbose, collect_stats);
          case DataType::int8:
            return makeIndex<DataType::int8>(distance_type, dim, dataset_size, max_edges_per_node, verbose, collect_stats);
          case DataType::uint8:
            return makeIndex<DataType::uint8>(distance_type, dim, dataset_size, max_edges_per_node, verbose, collect_stats);
          default:
            throw std::runtime_error("Unsupported data type");
        }
      },
      py::arg("distance_type"), py::arg("dim"), py::arg("dataset_size"), py::arg("max_edges_per_node"),
      py::arg("index_data_type") = DataType::float32, py::arg("verbose") = false, py::arg("collect_stats") = false,
      "Create an index: distance_type 'l2' or 'angular' (1 - <x,y>, inputs are NOT normalised for you).");
}
