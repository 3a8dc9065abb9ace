// =============================================================================
//  flatnav_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
//  A CPU restatement of the reference's flat-NSW index (construction, beam
//  search, cereal binary file layout), written from the reference's documented
//  behaviour so the HIP search path can be checked against it.  Only tests/,
//  __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
//  library; the product (flatnav_amd/, include/) never links or imports it.
//
//  PINNING STATUS (read DESIGN.md "Oracle"):
//    * distances  -- pinned: compared against the reference's own SIMD/scalar
//      distance code compiled from /root/reference (oracle/_ref, see
//      oracle/ref_cereal_free.cpp) and against the reference's known answers
//      (include/flatnav/tests/test_distances.cpp:84-100).
//    * beam search / construction -- PARITY UNPINNED against an executed
//      reference: flatnav/index/Index.h cannot be compiled in this image
//      (its only dependency `cereal` is an empty, un-vendored submodule and
//      stand-in headers are not allowed), and the reference ships no golden
//      vectors for search.  The restatement follows Index.h statement by
//      statement and uses the very same libstdc++ containers
//      (std::priority_queue + the same comparators, std::sort with the same
//      lambda), so heap/tie behaviour is libstdc++'s by construction.
//
//  Every function cites the reference file:line it follows (paths relative to
//  /root/reference/include/flatnav/).
// =============================================================================
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace orc {

typedef uint32_t node_id_t;
typedef int32_t label_t;  // python-bindings/src/flatnav/bindings.cpp:363 (label_t = int)
typedef std::pair<float, node_id_t> dist_node_t;
typedef std::pair<float, label_t> dist_label_t;

// util/Datatype.h:11-24 -- ordinals are part of the file format.
enum : int { DT_UINT8 = 0, DT_INT8 = 4, DT_FLOAT32 = 9 };
// distances/DistanceInterface.h:14
enum : int { METRIC_L2 = 0, METRIC_IP = 1 };

static thread_local std::string g_last_error;

typedef float (*dist_fn_t)(const void*, const void*, size_t);

// ---------------------------------------------------------------------------
// Distances.  Definition: L2 = sum (x-y)^2 (no sqrt), IP = 1 - sum x*y
// (distances/L2DistanceDispatcher.h:10-17, IPDistanceDispatcher.h:10-16).
// Float: 16 independent partial sums + fixed pairwise tree (the reference's
// AVX-512 kernels also keep 16 partial sums, util/SquaredL2SimdExtensions.h:
// 8-27; its horizontal order is the compiler's, so float results agree only to
// rounding -- exact on integer-valued data).  Compiled with -ffp-contract=off
// and without -ffast-math so the order below is the order executed.
// ---------------------------------------------------------------------------
static inline float tree16(const float* a) {
  float b[8], c[4];
  for (int j = 0; j < 8; j++) b[j] = a[j] + a[j + 8];
  for (int j = 0; j < 4; j++) c[j] = b[j] + b[j + 4];
  return (c[0] + c[2]) + (c[1] + c[3]);
}

static float l2_f32(const void* xv, const void* yv, size_t d) {
  const float* x = (const float*)xv;
  const float* y = (const float*)yv;
  float acc[16] = {0};
  size_t i = 0;
  for (; i + 16 <= d; i += 16)
    for (int j = 0; j < 16; j++) {
      float t = x[i + j] - y[i + j];
      acc[j] += t * t;
    }
  for (int j = 0; i < d; i++, j++) {
    float t = x[i] - y[i];
    acc[j] += t * t;
  }
  return tree16(acc);
}

static float ip_f32(const void* xv, const void* yv, size_t d) {
  const float* x = (const float*)xv;
  const float* y = (const float*)yv;
  float acc[16] = {0};
  size_t i = 0;
  for (; i + 16 <= d; i += 16)
    for (int j = 0; j < 16; j++) acc[j] += x[i + j] * y[i + j];
  for (int j = 0; i < d; i++, j++) acc[j] += x[i] * y[i];
  return 1.0f - tree16(acc);
}

// Integer element types: operands promote to int before subtract/multiply
// (L2DistanceDispatcher.h:10-17).  The reference accumulates in float (scalar
// path) or int32 (AVX-512 u8, dim%64==0, SquaredL2SimdExtensions.h:32-76);
// both equal the exact integer while it is < 2^24.  The oracle accumulates
// exactly in int64 and converts once.
template <typename T>
static float l2_int(const void* xv, const void* yv, size_t d) {
  const T* x = (const T*)xv;
  const T* y = (const T*)yv;
  int64_t s = 0;
  for (size_t i = 0; i < d; i++) {
    int t = (int)x[i] - (int)y[i];
    s += (int64_t)(t * t);
  }
  return (float)s;
}
template <typename T>
static float ip_int(const void* xv, const void* yv, size_t d) {
  const T* x = (const T*)xv;
  const T* y = (const T*)yv;
  int64_t s = 0;
  for (size_t i = 0; i < d; i++) s += (int64_t)((int)x[i] * (int)y[i]);
  return 1.0f - (float)s;
}

static dist_fn_t pick_distance(int metric, int dtype) {
  if (metric == METRIC_L2) {
    if (dtype == DT_FLOAT32) return l2_f32;
    if (dtype == DT_UINT8) return l2_int<uint8_t>;
    if (dtype == DT_INT8) return l2_int<int8_t>;
  } else if (metric == METRIC_IP) {
    if (dtype == DT_FLOAT32) return ip_f32;
    if (dtype == DT_UINT8) return ip_int<uint8_t>;
    if (dtype == DT_INT8) return ip_int<int8_t>;
  }
  return nullptr;
}

static size_t dtype_size(int dtype) {
  switch (dtype) {
    case DT_FLOAT32: return 4;
    case DT_UINT8:
    case DT_INT8: return 1;
    default: return 0;
  }
}

// ---------------------------------------------------------------------------
// util/VisitedSetPool.h:16-50 -- byte table with an 8-bit epoch mark.
// ---------------------------------------------------------------------------
struct VisitedSet {
  uint8_t mark;
  std::vector<uint8_t> table;
  explicit VisitedSet(size_t n) : mark(1), table(n, 0) {}
  void clear() {
    mark++;
    if (mark == 0) {
      std::fill(table.begin(), table.end(), 0);
      mark = 1;
    }
  }
  bool isVisited(uint32_t i) const { return table[i] == mark; }
  void insert(uint32_t i) { table[i] = mark; }
};

// index/Index.h:47-53 -- heaps compare on distance only.
struct CompareByFirst {
  bool operator()(dist_node_t const& a, dist_node_t const& b) const noexcept { return a.first < b.first; }
};
typedef std::priority_queue<dist_node_t, std::vector<dist_node_t>, CompareByFirst> PriorityQueue;

// Per-call counters kept private to the caller (the reference's shared atomic
// _distance_computations, Index.h:83, is reproduced by summing these).
struct Stats {
  uint64_t n_init = 0;    // what the reference adds per initializeSearch (Index.h:857-859)
  uint64_t n_dist = 0;    // neighbour evaluations (Index.h:689-691)
  uint64_t n_hops = 0;    // candidates popped and expanded
  uint64_t n_admit = 0;   // admissions (Index.h:693-695)
  uint64_t max_cand = 0;  // high-water mark of the candidates heap
};

struct Index {
  int dtype = DT_FLOAT32;
  int metric = METRIC_L2;
  size_t M = 0, dim = 0, data_size = 0, node_size = 0, max_nodes = 0, cur_nodes = 0;
  std::vector<char> mem_owner;
  char* mem = nullptr;
  dist_fn_t dist = nullptr;
  std::mutex guard;
  std::unique_ptr<std::mutex[]> node_mutex;

  // index/Index.h:159-179
  Index(int metric_, int dtype_, size_t dim_, size_t max_nodes_, size_t M_)
      : dtype(dtype_), metric(metric_), M(M_), dim(dim_), max_nodes(max_nodes_) {
    dist = pick_distance(metric, dtype);
    if (!dist) throw std::invalid_argument("unsupported metric/dtype");
    data_size = dim * dtype_size(dtype);
    node_size = data_size + sizeof(node_id_t) * M + sizeof(label_t);
    mem_owner.resize((uint64_t)node_size * (uint64_t)max_nodes);
    mem = mem_owner.data();
    node_mutex.reset(new std::mutex[max_nodes]);
  }
  Index() {}

  // index/Index.h:555-573 -- node = [data][M links][label]
  char* nodeData(node_id_t n) const { return mem + (uint64_t)n * (uint64_t)node_size; }
  node_id_t* nodeLinks(node_id_t n) const {
    return reinterpret_cast<node_id_t*>(mem + (uint64_t)n * (uint64_t)node_size + data_size);
  }
  label_t* nodeLabel(node_id_t n) const {
    return reinterpret_cast<label_t*>(mem + (uint64_t)n * (uint64_t)node_size + data_size +
                                      M * sizeof(node_id_t));
  }

  // index/Index.h:845-870
  node_id_t initializeSearch(const void* query, int n_init, Stats* st) const {
    if (n_init <= 0) throw std::invalid_argument("num_initializations must be greater than 0.");
    int step = (int)(cur_nodes / (size_t)n_init);
    step = step ? step : 1;
    float min_dist = std::numeric_limits<float>::max();
    node_id_t entry = 0;
    if (st) st->n_init += (uint64_t)n_init;
    for (node_id_t node = 0; node < cur_nodes; node += step) {
      float d = dist(query, nodeData(node), dim);
      if (d < min_dist) {
        min_dist = d;
        entry = node;
      }
    }
    return entry;
  }

  // index/Index.h:606-659 (beamSearch) with 661-707 (processCandidateNode) inlined.
  PriorityQueue beamSearch(const void* query, node_id_t entry, int buffer_size, VisitedSet& visited,
                           Stats* st, bool lock_nodes) {
    PriorityQueue neighbors, candidates;
    visited.clear();
    float d0 = dist(query, nodeData(entry), dim);
    float max_dist = d0;
    candidates.emplace(-d0, entry);
    neighbors.emplace(d0, entry);
    visited.insert(entry);
    if (st) st->max_cand = std::max<uint64_t>(st->max_cand, 1);

    while (!candidates.empty()) {
      dist_node_t top = candidates.top();
      if (-top.first > max_dist && neighbors.size() >= (size_t)buffer_size) break;  // Index.h:630
      candidates.pop();
      node_id_t node = top.second;
      if (st) st->n_hops++;

      std::unique_lock<std::mutex> lock;
      if (lock_nodes) lock = std::unique_lock<std::mutex>(node_mutex[node]);  // Index.h:664
      node_id_t* links = nodeLinks(node);
      for (uint32_t i = 0; i < M; i++) {
        node_id_t nb = links[i];
        if (visited.isVisited(nb)) continue;  // Index.h:679-683
        visited.insert(nb);                   // marked before the distance test, Index.h:684
        float d = dist(query, nodeData(nb), dim);
        if (st) st->n_dist++;
        if (neighbors.size() < (size_t)buffer_size || d < max_dist) {  // Index.h:693
          candidates.emplace(-d, nb);
          neighbors.emplace(d, nb);
          if (st) {
            st->n_admit++;
            st->max_cand = std::max<uint64_t>(st->max_cand, candidates.size());
          }
          if (neighbors.size() > (size_t)buffer_size) neighbors.pop();
          if (!neighbors.empty()) max_dist = neighbors.top().first;
        }
      }
    }
    return neighbors;
  }

  // index/Index.h:387-409
  std::vector<dist_label_t> search(const void* query, int K, int ef, int n_init, VisitedSet& visited,
                                   Stats* st) {
    node_id_t entry = initializeSearch(query, n_init, st);
    PriorityQueue neighbors = beamSearch(query, entry, std::max(ef, K), visited, st, false);
    std::vector<dist_label_t> results;
    results.reserve(neighbors.size());
    while (!neighbors.empty()) {
      dist_node_t t = neighbors.top();
      results.emplace_back(t.first, *nodeLabel(t.second));
      neighbors.pop();
    }
    std::sort(results.begin(), results.end(),
              [](const dist_label_t& l, const dist_label_t& r) { return l.first < r.first; });
    if (results.size() > (size_t)K) results.resize(K);
    return results;
  }

  // index/Index.h:714-763 -- HNSW heuristic; note the default pair comparator here.
  void selectNeighbors(PriorityQueue& neighbors, int m) {
    if (neighbors.size() < (size_t)m) return;
    std::priority_queue<std::pair<float, node_id_t>> candidates;
    std::vector<dist_node_t> saved;
    saved.reserve(m);
    while (neighbors.size() > 0) {
      dist_node_t t = neighbors.top();
      candidates.emplace(-t.first, t.second);
      neighbors.pop();
    }
    while (candidates.size() > 0) {
      if (saved.size() >= (size_t)m) break;
      float d_query = -candidates.top().first;
      node_id_t cur = candidates.top().second;
      candidates.pop();
      bool keep = true;
      for (const auto& s : saved) {
        float cur_dist = dist(nodeData(s.second), nodeData(cur), dim);
        if (cur_dist < d_query) {
          keep = false;
          break;
        }
      }
      if (keep) saved.push_back(std::make_pair(-d_query, cur));
    }
    for (const dist_node_t& p : saved) neighbors.emplace(-p.first, p.second);
  }

  // index/Index.h:765-834
  void connectNeighbors(PriorityQueue& neighbors, node_id_t new_id) {
    std::unique_lock<std::mutex> lock(node_mutex[new_id]);
    node_id_t* new_links = nodeLinks(new_id);
    int i = 0;
    while (neighbors.size() > 0) {
      node_id_t nb = neighbors.top().second;
      new_links[i] = nb;
      std::unique_lock<std::mutex> nb_lock(node_mutex[nb]);
      node_id_t* nb_links = nodeLinks(nb);
      bool inserted = false;
      for (size_t j = 0; j < M; j++) {
        if (nb_links[j] == nb) {  // first self-loop slot
          nb_links[j] = new_id;
          inserted = true;
          break;
        }
      }
      if (!inserted) {
        float max_dist = dist(nodeData(nb), nodeData(new_id), dim);
        PriorityQueue cands;
        cands.emplace(max_dist, new_id);
        for (size_t j = 0; j < M; j++) {
          if (nb_links[j] != nb) {
            node_id_t l = nb_links[j];
            cands.emplace(dist(nodeData(nb), nodeData(l), dim), l);
          }
        }
        selectNeighbors(cands, (int)M);
        size_t j = 0;
        while (cands.size() > 0) {
          nb_links[j] = cands.top().second;
          cands.pop();
          j++;
        }
        while (j < M) nb_links[j++] = nb;
      }
      nb_lock.unlock();
      i++;
      neighbors.pop();
    }
  }

  // index/Index.h:262-272
  void allocateNode(const void* data, label_t label, node_id_t& new_id) {
    new_id = (node_id_t)cur_nodes;
    std::memcpy(nodeData(new_id), data, data_size);
    *nodeLabel(new_id) = label;
    std::fill_n(nodeLinks(new_id), M, new_id);
    cur_nodes++;
  }

  // index/Index.h:353-378
  void add(const void* data, label_t label, int efc, int n_init, VisitedSet& visited, Stats* st) {
    if (cur_nodes >= max_nodes)
      throw std::runtime_error("Maximum number of nodes reached. Consider increasing the `max_node_count` parameter to create a larger index.");
    std::unique_lock<std::mutex> g(guard);
    node_id_t entry = initializeSearch(data, n_init, st);
    node_id_t new_id;
    allocateNode(data, label, new_id);
    g.unlock();
    if (new_id == 0) return;
    PriorityQueue neighbors = beamSearch(data, entry, efc, visited, st, true);
    int sel = std::max((int)(M / 2), 1);
    selectNeighbors(neighbors, sel);
    connectNeighbors(neighbors, new_id);
  }
};

// util/Multithreading.h:19-48 -- dynamic self-scheduling over row indices.
template <typename F>
static void parallel_for(uint64_t n, int threads, F&& fn) {
  if (threads <= 1 || n <= 1) {
    for (uint64_t i = 0; i < n; i++) fn(i, 0);
    return;
  }
  std::atomic<uint64_t> cur(0);
  std::vector<std::thread> pool;
  std::vector<std::string> errors(threads);
  for (int t = 0; t < threads; t++) {
    pool.emplace_back([&, t] {
      try {
        while (true) {
          uint64_t i = cur.fetch_add(1);
          if (i >= n) break;
          fn(i, t);
        }
      } catch (const std::exception& e) {
        errors[t] = e.what();
        cur.store(n);
      }
    });
  }
  for (auto& th : pool) th.join();
  for (auto& e : errors)
    if (!e.empty()) throw std::runtime_error(e);
}

// ---------------------------------------------------------------------------
// File format: cereal BinaryOutputArchive of Index::serialize (Index.h:134-141)
// + distance serialize (SquaredL2Distance.h:54-57): little-endian, no framing.
//   int32 data_type | u64 M | u64 data_size | u64 node_size | u64 max_nodes |
//   u64 cur_nodes | u64 dimension | u64 data_size | blob[node_size*max_nodes]
// ---------------------------------------------------------------------------
static void save_index(const Index& ix, const std::string& path) {
  std::ofstream f(path, std::ios::binary);
  if (!f.is_open()) throw std::runtime_error("Unable to open file for writing: " + path);
  int32_t dt = ix.dtype;
  uint64_t h[7] = {ix.M, ix.data_size, ix.node_size, ix.max_nodes, ix.cur_nodes, ix.dim, ix.data_size};
  f.write((const char*)&dt, 4);
  f.write((const char*)h, sizeof(h));
  f.write(ix.mem, (std::streamsize)((uint64_t)ix.node_size * ix.max_nodes));
  if (!f) throw std::runtime_error("write failed: " + path);
}

static Index* load_index(const std::string& path, int metric) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) throw std::runtime_error("Unable to open file for reading: " + path);
  int32_t dt;
  uint64_t h[7];
  f.read((char*)&dt, 4);
  f.read((char*)h, sizeof(h));
  if (!f) throw std::runtime_error("truncated index header: " + path);
  std::unique_ptr<Index> ix(new Index());
  ix->dtype = dt;
  ix->metric = metric;
  ix->M = h[0];
  ix->data_size = h[1];
  ix->node_size = h[2];
  ix->max_nodes = h[3];
  ix->cur_nodes = h[4];
  ix->dim = h[5];
  ix->dist = pick_distance(metric, dt);
  if (!ix->dist) throw std::runtime_error("unsupported data type in file");
  if (ix->node_size != ix->data_size + 4 * ix->M + 4 || ix->cur_nodes > ix->max_nodes ||
      ix->data_size != ix->dim * dtype_size(dt))
    throw std::runtime_error("inconsistent index header: " + path);
  ix->mem_owner.resize((uint64_t)ix->node_size * ix->max_nodes);
  ix->mem = ix->mem_owner.data();
  f.read(ix->mem, (std::streamsize)ix->mem_owner.size());
  if (!f) throw std::runtime_error("truncated index blob: " + path);
  ix->node_mutex.reset(new std::mutex[ix->max_nodes]);
  return ix.release();
}

}  // namespace orc

// =============================================================================
// C ABI (ctypes).  All functions return 0 on success, non-zero on error with
// the message available from orc_last_error().
// =============================================================================
using namespace orc;

#define ORC_TRY try {
#define ORC_CATCH                              \
  }                                            \
  catch (const std::invalid_argument& e) {     \
    g_last_error = e.what();                   \
    return 1;                                  \
  }                                            \
  catch (const std::exception& e) {            \
    g_last_error = e.what();                   \
    return 2;                                  \
  }                                            \
  return 0;

extern "C" {

const char* orc_last_error() { return g_last_error.c_str(); }

int orc_create(int metric, int dtype, uint64_t dim, uint64_t max_nodes, uint64_t M, void** out) {
  ORC_TRY
  *out = new Index(metric, dtype, dim, max_nodes, M);
  ORC_CATCH
}

int orc_free(void* h) {
  delete (Index*)h;
  return 0;
}

// info[8] = {dtype, M, data_size, node_size, max_nodes, cur_nodes, dim, metric}
int orc_info(void* h, uint64_t* info) {
  Index* ix = (Index*)h;
  info[0] = (uint64_t)ix->dtype;
  info[1] = ix->M;
  info[2] = ix->data_size;
  info[3] = ix->node_size;
  info[4] = ix->max_nodes;
  info[5] = ix->cur_nodes;
  info[6] = ix->dim;
  info[7] = (uint64_t)ix->metric;
  return 0;
}

const void* orc_blob(void* h) { return ((Index*)h)->mem; }

// Replace the distance function (used by tests to run the restated search on
// top of the reference's own compiled distance kernels from oracle/_ref).
int orc_set_distance_fn(void* h, void* fn) {
  ((Index*)h)->dist = fn ? (dist_fn_t)fn : pick_distance(((Index*)h)->metric, ((Index*)h)->dtype);
  return 0;
}

// index/Index.h:301-329 (addBatch).  labels == NULL -> iota continuing from
// the current node count (bindings.cpp:86-101 uses iota from 0 per call).
int orc_add(void* h, const void* data, uint64_t n, const int32_t* labels, int efc, int n_init,
            int threads, uint64_t* dist_comps) {
  ORC_TRY
  Index* ix = (Index*)h;
  if (n_init <= 0) throw std::invalid_argument("num_initializations must be greater than 0.");
  if (threads < 1) threads = 1;
  std::vector<std::unique_ptr<VisitedSet>> vs;
  std::vector<Stats> st(threads);
  for (int t = 0; t < threads; t++) vs.emplace_back(new VisitedSet(ix->max_nodes));
  size_t row = ix->data_size;
  parallel_for(n, threads, [&](uint64_t i, int t) {
    label_t lab = labels ? labels[i] : (label_t)i;
    ix->add((const char*)data + i * row, lab, efc, n_init, *vs[t], &st[t]);
  });
  if (dist_comps) {
    uint64_t s = 0;
    for (auto& x : st) s += x.n_init + x.n_dist;
    *dist_comps = s;
  }
  ORC_CATCH
}

// Batched search = python-bindings/src/flatnav/bindings.cpp:161-228 over
// Index::search.  out_count[q] = number of results (< K possible, Index.h:
// 404-408); unused slots are filled with (+inf, -1).  Optional per-query
// counters (nullable): ndist, nhops, nadmit, maxcand.
int orc_search(void* h, const void* queries, uint64_t nq, int K, int ef, int n_init, int threads,
               float* out_d, int32_t* out_l, int32_t* out_count, uint64_t* ndist, uint64_t* nhops,
               uint64_t* nadmit, uint64_t* maxcand) {
  ORC_TRY
  Index* ix = (Index*)h;
  if (n_init <= 0) throw std::invalid_argument("num_initializations must be greater than 0.");
  if (K <= 0) throw std::invalid_argument("K must be positive");
  if (threads < 1) threads = 1;
  std::vector<std::unique_ptr<VisitedSet>> vs;
  for (int t = 0; t < threads; t++) vs.emplace_back(new VisitedSet(ix->max_nodes));
  size_t row = ix->data_size;
  parallel_for(nq, threads, [&](uint64_t q, int t) {
    Stats st;
    std::vector<dist_label_t> r = ix->search((const char*)queries + q * row, K, ef, n_init, *vs[t], &st);
    for (int k = 0; k < K; k++) {
      if ((size_t)k < r.size()) {
        out_d[q * K + k] = r[k].first;
        out_l[q * K + k] = r[k].second;
      } else {
        out_d[q * K + k] = std::numeric_limits<float>::infinity();
        out_l[q * K + k] = -1;
      }
    }
    if (out_count) out_count[q] = (int32_t)r.size();
    if (ndist) ndist[q] = st.n_dist;
    if (nhops) nhops[q] = st.n_hops;
    if (nadmit) nadmit[q] = st.n_admit;
    if (maxcand) maxcand[q] = st.max_cand;
  });
  ORC_CATCH
}

// The reference's `neighbors` heap driven by a LOG of evaluated neighbours instead of a traversal: entry 0 is the entry
// point, then every evaluated neighbour (distance, node id) in evaluation order.  Admission (Index.h:693-704) and result
// assembly (Index.h:393-408) are the reference's, on the real std::priority_queue / std::sort.  Used by
// tests/test_replay_model.py (a design check for the GPU path: when the merged-beam kernel's traversal is known to be the
// reference's and only the ORDER of equal distances among the results is open, replaying its log decides it).
int orc_replay_neighbors(const float* d, const uint32_t* ids, uint64_t n, int buffer_size, int K, float* out_d,
                         uint32_t* out_ids, int32_t* out_count) {
  ORC_TRY
  if (n == 0 || K <= 0 || buffer_size <= 0) throw std::invalid_argument("empty log");
  PriorityQueue neighbors;
  neighbors.emplace(d[0], ids[0]);
  float max_dist = d[0];
  for (uint64_t i = 1; i < n; i++) {
    if (neighbors.size() < (size_t)buffer_size || d[i] < max_dist) {
      neighbors.emplace(d[i], ids[i]);
      if (neighbors.size() > (size_t)buffer_size) neighbors.pop();
      if (!neighbors.empty()) max_dist = neighbors.top().first;
    }
  }
  std::vector<dist_node_t> results;
  while (!neighbors.empty()) {
    results.push_back(neighbors.top());
    neighbors.pop();
  }
  std::sort(results.begin(), results.end(), [](const dist_node_t& l, const dist_node_t& r) { return l.first < r.first; });
  if (results.size() > (size_t)K) results.resize(K);
  for (size_t k = 0; k < results.size(); k++) {
    out_d[k] = results[k].first;
    out_ids[k] = results[k].second;
  }
  *out_count = (int32_t)results.size();
  ORC_CATCH
}

// Design check for the GPU's mid-flight hand-over (round 5; csrc/kernels.hpp `resume_from_log`): the reference's search
// RESUMED from a log that the merged-beam traversal wrote.  The log is a stream of records: a hop header {node, number of
// neighbours evaluated} followed by that hop's evaluated neighbours that could still be admitted when the row began
// (distance < max_dist of that moment, or the beam not yet full), in link order.  Both heaps are the real
// std::priority_queue: per logged hop the reference's loop head runs (Index.h:625-633) -- if its top is NOT the logged node
// (equal keys: the reference expands another node first) the replay stops THERE --, then the logged neighbours go through
// the reference's admission (Index.h:693-704).  The visited set is then rebuilt as {entry} + every link of the nodes
// expanded so far (Index.h:679-684 marks every link it looks at), and the reference's own loop continues from that state
// with real distances.  is_hdr[i] != 0: record i is a header (ids[i] = node, d[i] = evaluated count as a float).
int orc_replay_search(void* h, const void* query, int K, int ef, uint32_t entry, const float* d, const uint32_t* ids,
                      const uint8_t* is_hdr, uint64_t n, float* out_d, int32_t* out_l, int32_t* out_count,
                      uint64_t* ndist, uint64_t* nhops, uint64_t* hops_replayed) {
  ORC_TRY
  Index* ix = (Index*)h;
  const size_t B = (size_t)std::max(ef, K);
  PriorityQueue neighbors, candidates;
  const float d0 = ix->dist(query, ix->nodeData(entry), ix->dim);
  float max_dist = d0;
  candidates.emplace(-d0, entry);
  neighbors.emplace(d0, entry);
  std::vector<node_id_t> expanded;
  uint64_t n_dist = 0, n_hops = 0, pos = 0;
  while (pos < n) {
    if (!is_hdr[pos]) throw std::runtime_error("log: header expected");
    if (candidates.empty()) break;
    dist_node_t top = candidates.top();
    if (-top.first > max_dist && neighbors.size() >= B) break;  // (the reference would stop; cannot happen for a beam member)
    if (top.second != ids[pos]) break;                          // equal keys: the reference expands another node first
    candidates.pop();
    n_hops++;
    n_dist += (uint64_t)d[pos];
    expanded.push_back(ids[pos]);
    pos++;
    for (; pos < n && !is_hdr[pos]; pos++) {
      if (neighbors.size() < B || d[pos] < max_dist) {  // Index.h:693
        candidates.emplace(-d[pos], ids[pos]);
        neighbors.emplace(d[pos], ids[pos]);
        if (neighbors.size() > B) neighbors.pop();
        if (!neighbors.empty()) max_dist = neighbors.top().first;
      }
    }
  }
  if (hops_replayed) *hops_replayed = n_hops;
  VisitedSet visited(ix->max_nodes);
  visited.clear();
  visited.insert(entry);
  for (node_id_t e : expanded) {
    node_id_t* links = ix->nodeLinks(e);
    for (uint32_t i = 0; i < ix->M; i++) visited.insert(links[i]);
  }
  while (!candidates.empty()) {  // Index.h:625-659 from the replayed state
    dist_node_t top = candidates.top();
    if (-top.first > max_dist && neighbors.size() >= B) break;
    candidates.pop();
    n_hops++;
    node_id_t* links = ix->nodeLinks(top.second);
    for (uint32_t i = 0; i < ix->M; i++) {
      node_id_t nb = links[i];
      if (visited.isVisited(nb)) continue;
      visited.insert(nb);
      float dd = ix->dist(query, ix->nodeData(nb), ix->dim);
      n_dist++;
      if (neighbors.size() < B || dd < max_dist) {
        candidates.emplace(-dd, nb);
        neighbors.emplace(dd, nb);
        if (neighbors.size() > B) neighbors.pop();
        if (!neighbors.empty()) max_dist = neighbors.top().first;
      }
    }
  }
  std::vector<dist_label_t> results;
  while (!neighbors.empty()) {
    dist_node_t t = neighbors.top();
    results.emplace_back(t.first, *ix->nodeLabel(t.second));
    neighbors.pop();
  }
  std::sort(results.begin(), results.end(), [](const dist_label_t& l, const dist_label_t& r) { return l.first < r.first; });
  if (results.size() > (size_t)K) results.resize(K);
  for (size_t k = 0; k < results.size(); k++) {
    out_d[k] = results[k].first;
    out_l[k] = results[k].second;
  }
  *out_count = (int32_t)results.size();
  if (ndist) *ndist = n_dist;
  if (nhops) *nhops = n_hops;
  ORC_CATCH
}

int orc_save(void* h, const char* path) {
  ORC_TRY
  save_index(*(Index*)h, path);
  ORC_CATCH
}

int orc_load(const char* path, int metric, void** out) {
  ORC_TRY
  *out = load_index(path, metric);
  ORC_CATCH
}

// Adopt a raw AoS blob (e.g. one produced by the product's host builder) so
// the oracle can search the very same graph.
int orc_from_blob(int metric, int dtype, uint64_t dim, uint64_t max_nodes, uint64_t cur_nodes, uint64_t M,
                  const void* blob, void** out) {
  ORC_TRY
  std::unique_ptr<Index> ix(new Index(metric, dtype, dim, max_nodes, M));
  if (cur_nodes > max_nodes) throw std::invalid_argument("cur_nodes > max_nodes");
  std::memcpy(ix->mem, blob, (uint64_t)ix->node_size * max_nodes);
  ix->cur_nodes = cur_nodes;
  *out = ix.release();
  ORC_CATCH
}

float orc_distance(int metric, int dtype, const void* x, const void* y, uint64_t d) {
  dist_fn_t f = pick_distance(metric, dtype);
  return f ? f(x, y, d) : std::numeric_limits<float>::quiet_NaN();
}

}  // extern "C"
