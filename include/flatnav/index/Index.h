// flatnav/index/Index.h -- host side of the MI355X flat-NSW index (header-only C++17, own code).
//
// Keeps the public surface of the reference's flatnav::Index<dist_t, label_t>
// (include/flatnav/index/Index.h:159-548 of the reference): same constructor, add / addBatch /
// allocateNode / buildGraphLinks / getGraphOutdegreeTable / search / saveIndex / loadIndex /
// setNumThreads / getters / resetStats / getIndexSummary, same exceptions, same binary file format.
//
// What is different underneath:
//   * search() and searchBatch() never run on the CPU.  The node store is mirrored to one GPU's
//     HBM through the C ABI (flatnav_hip.h: fnv_index_upload) and every query is answered by the
//     gfx950 beam-search kernel (fnv_search_batch).  If the device library or a GPU is missing the
//     call throws -- there is no CPU fallback.
//   * Batched search is a first-class call (searchBatch): the reference loops Index::search over
//     query rows with executeInParallel (bindings.cpp:198-211); here the batch is one kernel launch.
//   * Index construction (add) stays on the host in this round.  It uses flat-array binary heaps
//     moved with libstdc++'s exact algorithm (util/StlExact.h), so a single-threaded build yields
//     the same graph bytes as the reference's std::priority_queue code on tie-free and tied data.
//   * Per-query counters come back from the device; the shared atomic of the reference
//     (Index.h:83) is only summed into once per batch.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <flatnav/distances/DistanceInterface.h>
#include <flatnav/util/BinaryArchive.h>
#include <flatnav/util/Datatype.h>
#include <flatnav/util/Multithreading.h>
#include <flatnav/util/Reordering.h>
#include <flatnav/util/StlExact.h>
#include <flatnav_hip.h>

namespace flatnav {

using flatnav::distances::DistanceInterface;
using flatnav::distances::MetricType;
using flatnav::util::DataType;

namespace detail {

// Maps a C-ABI status to the exception the reference throws at the same point.
inline void throwOnDeviceError(int rc) {
  if (rc == FNV_OK) return;
  std::string msg = fnv_last_error();
  if (rc == FNV_ERR_INVALID) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

// Growable array of heap entries with the get/set face StlExact.h expects.
struct EntryArray {
  std::vector<fnv_stl::Entry> items;
  fnv_stl::Entry get(int i) const { return items[static_cast<size_t>(i)]; }
  void set(int i, fnv_stl::Entry e) {
    if (static_cast<size_t>(i) >= items.size()) items.resize(static_cast<size_t>(i) + 1);
    items[static_cast<size_t>(i)] = e;
  }
};

// A max-heap keyed on Entry::key only, with exactly std::priority_queue's element moves.
struct KeyHeap {
  EntryArray a;
  int n = 0;
  void clear() { n = 0; }
  bool empty() const { return n == 0; }
  int size() const { return n; }
  fnv_stl::Entry top() const { return a.items[0]; }
  void push(float key, uint32_t val) {
    fnv_stl::heap_push(a, n, fnv_stl::Entry{key, val});
    ++n;
  }
  void pop() {
    fnv_stl::heap_pop(a, n);
    --n;
  }
};

// One byte per node, test-and-set with backoff; guards a node's link row during construction.
class NodeLocks {
  std::unique_ptr<std::atomic<uint8_t>[]> _flags;

 public:
  void reset(size_t n) {
    _flags.reset(new std::atomic<uint8_t>[n]);
    for (size_t i = 0; i < n; ++i) _flags[i].store(0, std::memory_order_relaxed);
  }
  void lock(uint32_t node) {
    while (_flags[node].exchange(1, std::memory_order_acquire)) {
      while (_flags[node].load(std::memory_order_relaxed)) std::this_thread::yield();
    }
  }
  void unlock(uint32_t node) { _flags[node].store(0, std::memory_order_release); }
};

struct NodeGuard {
  NodeLocks& locks;
  uint32_t node;
  NodeGuard(NodeLocks& l, uint32_t n) : locks(l), node(n) { locks.lock(node); }
  ~NodeGuard() { locks.unlock(node); }
};

// Which link rows the host builder has changed since the device mirror was last brought up to date (one byte
// per node, set from any thread), so that the next search ships those rows instead of the whole index.
class DirtyRows {
  std::unique_ptr<std::atomic<uint8_t>[]> _flags;
  std::atomic<uint64_t> _count{0};
  size_t _n = 0;

 public:
  void reset(size_t n) {
    _flags.reset(new std::atomic<uint8_t>[n]);
    for (size_t i = 0; i < n; ++i) _flags[i].store(0, std::memory_order_relaxed);
    _count.store(0);
    _n = n;
  }
  void mark(uint32_t node) {
    if (!_flags[node].exchange(1, std::memory_order_relaxed)) _count.fetch_add(1, std::memory_order_relaxed);
  }
  uint64_t count() const { return _count.load(); }
  // ids < limit that are marked, ascending; clears every mark
  std::vector<uint32_t> drain(size_t limit) {
    std::vector<uint32_t> ids;
    if (_count.load() != 0)
      for (size_t i = 0; i < _n; ++i)
        if (_flags[i].load(std::memory_order_relaxed)) {
          _flags[i].store(0, std::memory_order_relaxed);
          if (i < limit) ids.push_back(static_cast<uint32_t>(i));
        }
    _count.store(0);
    return ids;
  }
};

// Per-thread working memory of the host builder.
struct BuildScratch {
  std::vector<uint32_t> stamp;  // visited epoch per node
  uint32_t epoch = 0;
  KeyHeap beam, frontier, prune_pool;
  std::vector<fnv_stl::Entry> ordered, kept;
  explicit BuildScratch(size_t nodes) : stamp(nodes, 0u) {}
  void newEpoch() {
    if (++epoch == 0) {
      std::fill(stamp.begin(), stamp.end(), 0u);
      epoch = 1;
    }
  }
  bool seen(uint32_t node) const { return stamp[node] == epoch; }
  void mark(uint32_t node) { stamp[node] = epoch; }
};

}  // namespace detail

template <typename dist_t, typename label_t>
class Index {
  static_assert(sizeof(label_t) == 4, "the device path carries labels as 32-bit values");

 public:
  typedef uint32_t node_id_t;
  typedef std::pair<float, label_t> dist_label_t;

 private:
  std::unique_ptr<char[]> _index_memory;
  size_t _M = 0;
  size_t _data_size_bytes = 0;
  size_t _node_size_bytes = 0;  // [data][M links][label], as in the reference (Index.h:61-63, 176)
  size_t _max_node_count = 0;
  size_t _cur_num_nodes = 0;
  std::unique_ptr<DistanceInterface<dist_t>> _distance;
  uint32_t _num_threads = 1;
  bool _collect_stats = false;
  DataType _data_type = DataType::float32;
  mutable std::atomic<uint64_t> _distance_computations{0};
  mutable std::atomic<uint64_t> _metric_hops{0};

  // host builder state
  std::mutex _index_data_guard;
  detail::NodeLocks _node_locks;
  std::mutex _scratch_guard;
  std::vector<std::unique_ptr<detail::BuildScratch>> _scratch_pool;

  // device mirror
  mutable std::mutex _device_guard;
  mutable fnv_index_t _device_index = nullptr;
  mutable bool _device_stale = true;       // the mirror misses host-side changes (new nodes and / or dirty link rows)
  mutable bool _device_rebuild = true;     // ... and they are not tracked: the whole store must be shipped again
  mutable size_t _device_capacity = 0;     // rows allocated on the device
  mutable size_t _device_synced_nodes = 0; // nodes [0, this) have their records on the device
  mutable detail::DirtyRows _dirty_rows;   // link rows of synced nodes that changed since
  int _device_ordinal = 0;
  // several GPUs: the mirror on _device_ordinal is the primary; replicas of it live on the other devices of
  // _device_list (filled by peer copies, refreshed whenever the primary changes); batches are sharded over all of them
  mutable std::vector<int> _device_list;   // empty: not decided yet (FLATNAV_DEVICES, else the primary GPU alone)
  mutable bool _device_list_defaulted = false;  // FLATNAV_DEVICES=all: "every visible GPU", whatever that turns out to be
  mutable std::vector<fnv_index_t> _replicas;
  mutable bool _replicas_stale = true;

  Index() = default;
  Index(const Index&) = delete;
  Index& operator=(const Index&) = delete;

  // ---- node store ---------------------------------------------------------------------------
  char* nodeData(node_id_t n) const { return _index_memory.get() + static_cast<uint64_t>(n) * _node_size_bytes; }
  node_id_t* nodeLinks(node_id_t n) const { return reinterpret_cast<node_id_t*>(nodeData(n) + _data_size_bytes); }
  label_t* nodeLabel(node_id_t n) const {
    return reinterpret_cast<label_t*>(nodeData(n) + _data_size_bytes + _M * sizeof(node_id_t));
  }
  uint64_t storeBytes() const { return static_cast<uint64_t>(_node_size_bytes) * static_cast<uint64_t>(_max_node_count); }

  void allocateStore() {
    _index_memory.reset(new char[storeBytes()]());  // zero-filled: saved files are deterministic
    _node_locks.reset(_max_node_count);
    _dirty_rows.reset(_max_node_count);
  }

  template <typename Archive>
  void serialize(Archive& archive) {
    // field order = reference Index::serialize (Index.h:134-141) + distance serialize
    archive(_data_type, _M, _data_size_bytes, _node_size_bytes, _max_node_count, _cur_num_nodes, *_distance);
    archive(flatnav::util::binary_data(_index_memory.get(), storeBytes()));
  }
  friend class flatnav::util::BinaryWriter;
  friend class flatnav::util::BinaryReader;

  // ---- host builder -------------------------------------------------------------------------
  std::unique_ptr<detail::BuildScratch> borrowScratch() {
    std::lock_guard<std::mutex> g(_scratch_guard);
    if (!_scratch_pool.empty()) {
      auto s = std::move(_scratch_pool.back());
      _scratch_pool.pop_back();
      return s;
    }
    return std::make_unique<detail::BuildScratch>(_max_node_count);
  }
  void returnScratch(std::unique_ptr<detail::BuildScratch> s) {
    std::lock_guard<std::mutex> g(_scratch_guard);
    _scratch_pool.push_back(std::move(s));
  }

  // Entry node for an insertion: closest of the nodes 0, s, 2s, ... (s = max(1, N / n_init)),
  // first minimum wins -- the rule of the reference's initializeSearch (Index.h:845-870), which
  // the GPU kernel applies to queries.
  node_id_t pickEntry(const void* point, int num_initializations) {
    if (num_initializations <= 0) throw std::invalid_argument("num_initializations must be greater than 0.");
    size_t step = _cur_num_nodes / static_cast<size_t>(num_initializations);
    if (step == 0) step = 1;
    if (_collect_stats) _distance_computations.fetch_add(static_cast<uint64_t>(num_initializations));
    float best = std::numeric_limits<float>::max();
    node_id_t entry = 0;
    for (size_t node = 0; node < _cur_num_nodes; node += step) {
      const float d = _distance->distance(point, nodeData(static_cast<node_id_t>(node)), true);
      if (d < best) {
        best = d;
        entry = static_cast<node_id_t>(node);
      }
    }
    return entry;
  }

  // Beam search used while inserting a point (reference: beamSearch + processCandidateNode,
  // Index.h:606-707).  Leaves the beam (a max-heap on distance, at most `width` entries) in s.beam.
  void insertionBeam(const void* point, node_id_t entry, int width, detail::BuildScratch& s) {
    s.newEpoch();
    s.beam.clear();
    s.frontier.clear();
    const float d0 = _distance->distance(point, nodeData(entry), true);
    float worst = d0;
    s.frontier.push(-d0, entry);
    s.beam.push(d0, entry);
    s.mark(entry);
    uint64_t evaluated = 0;
    while (!s.frontier.empty()) {
      const fnv_stl::Entry next = s.frontier.top();
      if (-next.key > worst && s.beam.size() >= width) break;
      s.frontier.pop();
      detail::NodeGuard guard(_node_locks, next.val);
      const node_id_t* links = nodeLinks(next.val);
      for (size_t i = 0; i < _M; ++i) {
        const node_id_t nb = links[i];
        if (s.seen(nb)) continue;
        s.mark(nb);
        const float d = _distance->distance(point, nodeData(nb), true);
        ++evaluated;
        if (s.beam.size() < width || d < worst) {
          s.frontier.push(-d, nb);
          s.beam.push(d, nb);
          if (s.beam.size() > width) s.beam.pop();
          worst = s.beam.top().key;
        }
      }
    }
    if (_collect_stats) _distance_computations.fetch_add(evaluated);
  }

  // HNSW-style diversity pruning (reference selectNeighbors, Index.h:714-763).  Candidates are
  // visited closest first, equal distances by DESCENDING id (the pop order of the reference's
  // std::priority_queue<std::pair<float, node_id_t>> over (-distance, id)); a candidate is kept
  // unless some already-kept node is strictly closer to it than the query point is.
  void pruneNeighbors(detail::KeyHeap& heap, int keep, detail::BuildScratch& s) {
    if (heap.size() < keep) return;
    s.ordered.assign(heap.a.items.begin(), heap.a.items.begin() + heap.size());
    std::sort(s.ordered.begin(), s.ordered.end(), [](const fnv_stl::Entry& l, const fnv_stl::Entry& r) {
      return l.key < r.key || (l.key == r.key && l.val > r.val);
    });
    s.kept.clear();
    for (const fnv_stl::Entry& c : s.ordered) {
      if (static_cast<int>(s.kept.size()) >= keep) break;
      bool diverse = true;
      for (const fnv_stl::Entry& k : s.kept) {
        if (_distance->distance(nodeData(k.val), nodeData(c.val)) < c.key) {
          diverse = false;
          break;
        }
      }
      if (diverse) s.kept.push_back(c);
    }
    heap.clear();
    for (const fnv_stl::Entry& k : s.kept) heap.push(k.key, k.val);
  }

  // Wire the new node to its selected neighbours and add back-links (reference connectNeighbors,
  // Index.h:765-834): a back-link takes the neighbour's first free (self-loop) slot, otherwise the
  // neighbour's row is re-pruned over {old links} + {new node}.
  // `pop_order` (optional): the neighbours in the order the reference would pop them, for callers that know it without
  // holding the heap itself (beams searched on the GPU); `selected` is then only emptied.
  void linkNeighbors(detail::KeyHeap& selected, node_id_t new_id, detail::BuildScratch& s,
                     const node_id_t* pop_order = nullptr) {
    detail::NodeGuard own(_node_locks, new_id);
    node_id_t* new_links = nodeLinks(new_id);
    _dirty_rows.mark(new_id);
    size_t slot = 0;
    while (!selected.empty()) {
      const node_id_t nb = pop_order ? pop_order[slot] : selected.top().val;
      new_links[slot++] = nb;
      _dirty_rows.mark(nb);
      {
        detail::NodeGuard theirs(_node_locks, nb);
        node_id_t* nb_links = nodeLinks(nb);
        size_t free_slot = _M;
        for (size_t j = 0; j < _M; ++j)
          if (nb_links[j] == nb) {
            free_slot = j;
            break;
          }
        if (free_slot < _M) {
          nb_links[free_slot] = new_id;
        } else {
          detail::KeyHeap& pool = s.prune_pool;
          pool.clear();
          pool.push(_distance->distance(nodeData(nb), nodeData(new_id)), new_id);
          for (size_t j = 0; j < _M; ++j)
            if (nb_links[j] != nb) pool.push(_distance->distance(nodeData(nb), nodeData(nb_links[j])), nb_links[j]);
          pruneNeighbors(pool, static_cast<int>(_M), s);
          size_t j = 0;
          while (!pool.empty()) {
            nb_links[j++] = pool.top().val;
            pool.pop();
          }
          while (j < _M) nb_links[j++] = nb;
        }
      }
      selected.pop();
    }
  }

  void markDeviceStale() { _device_stale = true; }             // appended nodes / dirty rows: tracked
  void markDeviceRebuild() { _device_stale = _device_rebuild = true; }  // anything else (reorder, graph import)

  // Brings the device mirror up to date.  A mirror that already holds a prefix of the nodes receives only what
  // changed -- the appended node records (fnv_index_write_nodes) and the link rows the builder touched
  // (fnv_index_write_links) -- unless that is most of the index anyway; otherwise the store is shipped whole.
  void ensureDevice() const {
    if (_device_index && !_device_stale) return;
    resolveDeviceList();
    Index* self = const_cast<Index*>(this);
    const int metric = self->deviceMetric();
    const uint32_t dim = static_cast<uint32_t>(self->_distance->dimension());
    const bool incremental = _device_index && !_device_rebuild && _device_capacity >= _cur_num_nodes &&
                             _device_synced_nodes > 0 && _device_synced_nodes <= _cur_num_nodes &&
                             _dirty_rows.count() <= _cur_num_nodes / 4;
    if (incremental) {
      // The marks are drained before the writes: if a write fails they are gone, so the mirror is declared untracked
      // (next call ships the whole store) before the error goes up -- never a silently diverged device graph.
      std::vector<node_id_t> ids = _dirty_rows.drain(_device_synced_nodes);
      try {
        if (_cur_num_nodes > _device_synced_nodes)
          detail::throwOnDeviceError(fnv_index_write_nodes(_device_index, _device_synced_nodes,
                                                           _cur_num_nodes - _device_synced_nodes,
                                                           nodeData(static_cast<node_id_t>(_device_synced_nodes)),
                                                           _node_size_bytes, _data_size_bytes));
        if (!ids.empty()) {
          std::vector<node_id_t> rows(ids.size() * _M);
          for (size_t t = 0; t < ids.size(); ++t) std::memcpy(rows.data() + t * _M, nodeLinks(ids[t]), _M * sizeof(node_id_t));
          detail::throwOnDeviceError(fnv_index_write_links(_device_index, ids.data(), rows.data(), ids.size()));
        }
        detail::throwOnDeviceError(fnv_index_set_live_nodes(_device_index, _cur_num_nodes));
      } catch (...) {
        const_cast<Index*>(this)->markDeviceRebuild();
        throw;
      }
    } else {
      if (_device_index) {
        fnv_index_free(_device_index);
        _device_index = nullptr;
        dropReplicas();
      }
      if (_cur_num_nodes == _max_node_count) {  // complete: exactly as many rows as nodes
        detail::throwOnDeviceError(fnv_index_upload(_index_memory.get(), _node_size_bytes, _data_size_bytes,
                                                    static_cast<uint32_t>(_M), _cur_num_nodes,
                                                    static_cast<int>(_data_type), metric, dim, _device_ordinal,
                                                    &_device_index));
        _device_capacity = _cur_num_nodes;
      } else {  // still growing: room for every node the store can hold, so that later additions are appended
        detail::throwOnDeviceError(fnv_index_alloc(static_cast<uint32_t>(_M), _max_node_count,
                                                   static_cast<int>(_data_type), metric, dim, _device_ordinal,
                                                   &_device_index));
        _device_capacity = _max_node_count;
        detail::throwOnDeviceError(fnv_index_write_nodes(_device_index, 0, _cur_num_nodes, _index_memory.get(),
                                                         _node_size_bytes, _data_size_bytes));
        detail::throwOnDeviceError(fnv_index_set_live_nodes(_device_index, _cur_num_nodes));
      }
      _dirty_rows.drain(0);
    }
    _device_synced_nodes = _cur_num_nodes;
    _device_stale = _device_rebuild = false;
    _replicas_stale = true;
  }

  void dropReplicas() const {
    for (fnv_index_t r : _replicas) fnv_index_free(r);
    _replicas.clear();
    _replicas_stale = true;
  }

  // The GPUs this index spreads its batches over: setDevices(), else the FLATNAV_DEVICES environment variable
  // ("0,1,2,3" -- an ordinal may repeat: tests put two replicas on one GPU -- or "all" = every visible device), else
  // the primary GPU alone: a library must not allocate a full index copy on every GPU of the node, nor move work to
  // devices the caller did not name, unasked.
  void resolveDeviceList() const {
    if (!_device_list.empty()) return;
    if (const char* env = std::getenv("FLATNAV_DEVICES")) {
      if (std::string(env) == "all") {  // the primary first, then every other visible GPU
        int count = 1;
        if (fnv_device_count(&count) != FNV_OK || count < 1) count = 1;
        _device_list.push_back(_device_ordinal);
        for (int d = 0; d < count; ++d)
          if (d != _device_ordinal) _device_list.push_back(d);
        _device_list_defaulted = true;  // a replication failure falls back to the primary with a warning
      } else {
        std::stringstream ss(env);
        for (std::string tok; std::getline(ss, tok, ',');)
          if (!tok.empty()) _device_list.push_back(std::stoi(tok));
      }
    }
    if (_device_list.empty()) _device_list.push_back(_device_ordinal);  // nobody asked for more: the primary GPU only
    const_cast<Index*>(this)->_device_ordinal = _device_list[0];
  }

  // Brings the replicas in line with the (up-to-date) primary mirror.
  void ensureReplicas() const {
    resolveDeviceList();
    if (_device_list.size() <= 1) return;
    if (!_replicas_stale && _replicas.size() + 1 == _device_list.size()) return;
    if (_replicas.size() + 1 != _device_list.size()) {
      dropReplicas();
      _replicas.resize(_device_list.size() - 1, nullptr);
      const int rc = fnv_replicate(_device_index, static_cast<int>(_replicas.size()), _device_list.data() + 1, _replicas.data());
      if (rc != FNV_OK) {
        _replicas.clear();
        if (_device_list_defaulted) {  // nobody asked for the other GPUs: serve from the primary alone, say so once
          std::fprintf(stderr, "flatnav: replicating the index over %zu visible GPUs failed (%s); searching on device %d only\n",
                       _device_list.size(), fnv_last_error(), _device_ordinal);
          _device_list.assign(1, _device_ordinal);
          _replicas_stale = false;
          return;
        }
        detail::throwOnDeviceError(rc);
      }
    } else {
      const int rc = fnv_replica_refresh(_device_index, static_cast<int>(_replicas.size()), _replicas.data());
      if (rc != FNV_OK) {  // e.g. the primary was re-created with another capacity: start over
        dropReplicas();
        return ensureReplicas();
      }
    }
    _replicas_stale = false;
  }

  int deviceMetric() { return _distance->metricType() == MetricType::L2 ? FNV_METRIC_L2 : FNV_METRIC_IP; }

  // Device index with room for every node the store can hold, holding nodes [0, _cur_num_nodes): what the
  // device-assisted builder appends to.
  void ensureDeviceAtCapacity() {
    if (_device_index && _device_capacity != _max_node_count) markDeviceRebuild();
    if (_cur_num_nodes == _max_node_count) return ensureDevice();
    ensureDevice();  // allocates at full capacity while the store is not full
  }

 public:
  // Reference constructor (Index.h:159-179).
  Index(std::unique_ptr<DistanceInterface<dist_t>> dist, int dataset_size, int max_edges_per_node,
        bool collect_stats = false, DataType data_type = DataType::float32)
      : _M(static_cast<size_t>(max_edges_per_node)),
        _max_node_count(static_cast<size_t>(dataset_size)),
        _distance(std::move(dist)),
        _collect_stats(collect_stats),
        _data_type(data_type) {
    _data_size_bytes = _distance->dataSize();
    _node_size_bytes = _data_size_bytes + sizeof(node_id_t) * _M + sizeof(label_t);
    allocateStore();
  }

  ~Index() {
    dropReplicas();
    if (_device_index) fnv_index_free(_device_index);
  }

  // ---- GPU placement (new) ------------------------------------------------------------------
  void setDevice(int ordinal) { setDevices(std::vector<int>{ordinal}); }
  // The GPUs that hold a copy of the index; batched searches are sharded over all of them (rows
  // [g * ceil(Q/G), ...) to the g-th), like the reference shards a batch over host threads
  // (bindings.cpp:198-211).  Default: FLATNAV_DEVICES ("all" = every visible GPU), else the primary GPU only.
  void setDevices(const std::vector<int>& ordinals) {
    if (ordinals.empty()) throw std::invalid_argument("setDevices: at least one device ordinal is required");
    std::lock_guard<std::mutex> g(_device_guard);
    _device_list_defaulted = false;
    if (ordinals == _device_list) return;
    dropReplicas();
    if (ordinals[0] != _device_ordinal || _device_list.empty()) markDeviceRebuild();
    _device_list = ordinals;
    _device_ordinal = ordinals[0];
  }
  std::vector<int> devices() const {
    std::lock_guard<std::mutex> g(_device_guard);
    resolveDeviceList();
    return _device_list;
  }
  int device() const { return _device_ordinal; }
  // Push pending host-side changes to HBM now (otherwise done lazily by the next search).
  void syncDevice() {
    std::lock_guard<std::mutex> g(_device_guard);
    ensureDevice();
  }
  // The C-ABI handle of the device mirror (valid until the next mutation); for callers that drive
  // fnv_search_batch_device on their own buffers / streams.
  fnv_index_t deviceHandle() {
    std::lock_guard<std::mutex> g(_device_guard);
    ensureDevice();
    return _device_index;
  }

  // ---- graph import (reference Index.h:187-251) ----------------------------------------------
  void buildGraphLinks(const std::string& mtx_filename) {
    std::ifstream input(mtx_filename);
    if (!input.is_open()) throw std::runtime_error("Unable to open file for reading: " + mtx_filename);
    std::string line;
    while (std::getline(input, line))
      if (line.empty() || line[0] != '%') break;
    std::istringstream header(line);
    long long rows = 0, cols = 0, edges = 0;
    header >> rows >> cols >> edges;
    if (static_cast<size_t>(cols) != _max_node_count)
      throw std::runtime_error("Number of vertices in the mtx file does not match the size allocated for the index.");
    if (static_cast<size_t>(edges) != _M)
      throw std::runtime_error("Number of edges in the mtx file does not match the number of links per node.");
    long long u, v;
    while (input >> u >> v) {
      --u;  // MatrixMarket is 1-based
      --v;
      node_id_t* links = nodeLinks(static_cast<node_id_t>(u));
      for (size_t i = 0; i < _M; ++i)
        if (links[i] == static_cast<node_id_t>(u)) {  // first free slot
          links[i] = static_cast<node_id_t>(v);
          break;
        }
    }
    markDeviceRebuild();
  }

  std::vector<std::vector<uint32_t>> getGraphOutdegreeTable() {
    std::vector<std::vector<uint32_t>> table(_cur_num_nodes);
    for (node_id_t node = 0; node < _cur_num_nodes; ++node) {
      const node_id_t* links = nodeLinks(node);
      for (size_t i = 0; i < _M; ++i)
        if (links[i] != node) table[node].push_back(links[i]);
    }
    return table;
  }

  // ---- construction (reference Index.h:262-378) ----------------------------------------------
  void allocateNode(void* data, label_t& label, node_id_t& new_node_id) {
    new_node_id = static_cast<node_id_t>(_cur_num_nodes);
    _distance->transformData(nodeData(new_node_id), data);
    *nodeLabel(new_node_id) = label;
    std::fill_n(nodeLinks(new_node_id), _M, new_node_id);  // self-loop == empty slot
    ++_cur_num_nodes;
    markDeviceStale();
  }

  template <typename data_type>
  void addBatch(void* data, std::vector<label_t>& labels, int ef_construction, int num_initializations = 100) {
    if (num_initializations <= 0) throw std::invalid_argument("num_initializations must be greater than 0.");
    const uint32_t total = static_cast<uint32_t>(labels.size());
    const uint64_t dim = _distance->dimension();
    auto insertRow = [&](uint32_t row) {
      void* vec = static_cast<data_type*>(data) + static_cast<uint64_t>(row) * dim;
      label_t label = labels[row];
      this->add(vec, label, ef_construction, num_initializations);
    };
    if (_num_threads == 1) {
      for (uint32_t row = 0; row < total; ++row) insertRow(row);
      return;
    }
    flatnav::executeInParallel(0, total, _num_threads, insertRow);
  }

  // ---- device-assisted construction (new) --------------------------------------
  // Same insertion rule as add() -- beam search of width ef_construction over the nodes present, diversity
  // pruning to M/2, back-links with re-pruning (reference Index.h:353-378, 714-834) -- but the beam searches
  // of a whole batch of new points run as ONE GPU launch (the search kernel, node ids as output) against the
  // graph as it stood before the batch; pruning and wiring stay on the host threads.  Points of one batch do
  // not see each other as candidates (later arrivals still link back to them), so batches are kept small
  // relative to the graph: at most cur/growth_divisor points, capped at max_batch; the first `bootstrap`
  // points take the host path.  The result is a graph of the same family and search quality -- not the
  // byte-identical one add() builds with one thread (the reference's own multi-threaded build is not
  // reproducible either).
  struct DeviceBuildOptions {
    uint32_t max_batch = 32768;
    uint32_t bootstrap = 2048;
    uint32_t growth_divisor = 4;
    bool wire_on_device = true;  // pruning + back-links by wire_batch_kernel (M <= 64); false: on the host threads
  };

  template <typename data_type>
  void addBatchDevice(void* data, std::vector<label_t>& labels, int ef_construction, int num_initializations = 100,
                      DeviceBuildOptions opt = DeviceBuildOptions()) {
    if (num_initializations <= 0) throw std::invalid_argument("num_initializations must be greater than 0.");
    if (ef_construction <= 0) throw std::invalid_argument("ef_construction must be positive");
    const uint64_t total = labels.size();
    const uint64_t dim = _distance->dimension();
    if (_cur_num_nodes + total > _max_node_count)
      throw std::runtime_error(
          "Maximum number of nodes reached. Consider increasing the `max_node_count` parameter to create a larger "
          "index.");
    auto rowPtr = [&](uint64_t row) { return static_cast<void*>(static_cast<data_type*>(data) + row * dim); };
    uint64_t row = 0;
    if (_cur_num_nodes < opt.bootstrap) {  // too small a graph to be worth a launch: host insertions
      const uint64_t nb = std::min<uint64_t>(total, opt.bootstrap - _cur_num_nodes);
      auto insertRow = [&](uint32_t r) {
        label_t label = labels[r];
        this->add(rowPtr(r), label, ef_construction, num_initializations);
      };
      // one thread: the device builder is deterministic from here on, so its start should be too
      for (uint32_t r = 0; r < nb; ++r) insertRow(r);
      row = nb;
    }
    if (row == total) return;

    std::lock_guard<std::mutex> dev_lock(_device_guard);
    ensureDeviceAtCapacity();
    detail::throwOnDeviceError(fnv_set_option(_device_index, "output_node_ids", 1));
    struct RestoreLabels {
      fnv_index_t ix;
      ~RestoreLabels() { fnv_set_option(ix, "output_node_ids", 0); }
    } restore{_device_index};

    // A device call that throws mid-batch leaves host store and mirror out of step in ways nothing tracks: the mirror
    // is rebuilt from the host store by the next ensureDevice().
    struct RebuildOnError {
      Index* self;
      bool armed = true;
      ~RebuildOnError() { if (armed) self->markDeviceRebuild(); }
    } rebuild_guard{this};

    const bool device_wiring = opt.wire_on_device && _M <= 64;
    const uint64_t first_device_node = _cur_num_nodes;
    const int width = ef_construction;
    const int keep = std::max(static_cast<int>(_M / 2), 1);
    std::vector<float> beam_dist;
    std::vector<int32_t> beam_ids, beam_count;
    std::vector<uint64_t> beam_evals;
    std::vector<node_id_t> touched, link_rows;
    std::mutex touched_guard;
    while (row < total) {
      const uint64_t cur = _cur_num_nodes;
      const uint64_t batch = std::min<uint64_t>(
          {total - row, opt.max_batch, std::max<uint64_t>(256, cur / std::max<uint32_t>(1, opt.growth_divisor))});
      // 1. store the new nodes (vector, label, empty link row) and mirror them to the device; searches
      //    still see only the `cur` nodes that are wired.
      auto storeNode = [&](uint32_t i) {  // allocateNode (Index.h:262-272) for slot cur + i
        const node_id_t id = static_cast<node_id_t>(cur + i);
        _distance->transformData(nodeData(id), rowPtr(row + i));
        *nodeLabel(id) = labels[row + i];
        std::fill_n(nodeLinks(id), _M, id);
      };
      for (uint32_t i = 0; i < batch; ++i) storeNode(i);  // a plain copy loop: measured faster than fanning out
      _cur_num_nodes += batch;
      markDeviceStale();
      detail::throwOnDeviceError(fnv_index_write_nodes(_device_index, cur, batch, nodeData(static_cast<node_id_t>(cur)),
                                                       _node_size_bytes, _data_size_bytes));
      if (device_wiring) {  // search + prune + wire in HBM; the host rows are refreshed once at the end
        uint64_t evals = 0;
        detail::throwOnDeviceError(
            fnv_index_insert_batch(_device_index, cur, batch, width, num_initializations, &evals));
        if (_collect_stats) _distance_computations.fetch_add(evals + batch * static_cast<uint64_t>(num_initializations));
        row += batch;
        continue;
      }
      // 2. the beam searches of the batch: one launch, full beams back (node ids, ascending distance)
      beam_dist.resize(batch * width);
      beam_ids.resize(batch * width);
      beam_count.resize(batch);
      beam_evals.resize(batch);
      detail::throwOnDeviceError(fnv_search_batch(_device_index, rowPtr(row),
                                                  batch, width, width, num_initializations, beam_dist.data(),
                                                  beam_ids.data(), beam_count.data(), beam_evals.data(), nullptr));
      // 3. prune + wire on the host threads
      touched.clear();
      auto wire = [&](uint32_t i) {
        auto scratch = borrowScratch();
        detail::KeyHeap& beam = scratch->beam;
        beam.clear();
        const float* bd = beam_dist.data() + static_cast<uint64_t>(i) * width;
        const int32_t* bi = beam_ids.data() + static_cast<uint64_t>(i) * width;
        for (int j = 0; j < beam_count[i]; ++j) beam.push(bd[j], static_cast<node_id_t>(bi[j]));
        pruneNeighbors(beam, keep, *scratch);
        node_id_t mine[1 + 64];
        int n_mine = 0;
        const node_id_t new_id = static_cast<node_id_t>(cur + i);
        mine[n_mine++] = new_id;
        for (int j = 0; j < beam.size() && n_mine < 65; ++j) mine[n_mine++] = beam.a.items[static_cast<size_t>(j)].val;
        // Fewer candidates than slots: the reference pops the search's own heap as it is (Index.h:715-717); the
        // device wrote that beam closest first with equal distances in reverse pop order, so read it backwards.
        node_id_t as_popped[64];
        const bool untouched = beam_count[i] < keep && beam_count[i] <= 64;
        for (int j = 0; untouched && j < beam_count[i]; ++j) as_popped[j] = static_cast<node_id_t>(bi[beam_count[i] - 1 - j]);
        linkNeighbors(beam, new_id, *scratch, untouched ? as_popped : nullptr);
        returnScratch(std::move(scratch));
        std::lock_guard<std::mutex> g(touched_guard);
        touched.insert(touched.end(), mine, mine + n_mine);
      };
      if (_num_threads == 1) for (uint32_t i = 0; i < batch; ++i) wire(i);
      else flatnav::executeInParallel(0, static_cast<uint32_t>(batch), _num_threads, wire);
      if (_collect_stats) {
        uint64_t evals = batch * static_cast<uint64_t>(num_initializations);
        for (uint64_t e : beam_evals) evals += e;
        _distance_computations.fetch_add(evals);
      }
      // 4. mirror every link row that changed, then let searches see the batch
      std::sort(touched.begin(), touched.end());
      touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
      link_rows.resize(touched.size() * _M);
      auto packRow = [&](uint32_t t) { std::memcpy(link_rows.data() + static_cast<uint64_t>(t) * _M, nodeLinks(touched[t]), _M * sizeof(node_id_t)); };
      if (_num_threads == 1) for (uint32_t t = 0; t < touched.size(); ++t) packRow(t);
      else flatnav::executeInParallel(0, static_cast<uint32_t>(touched.size()), _num_threads, packRow);
      detail::throwOnDeviceError(fnv_index_write_links(_device_index, touched.data(), link_rows.data(), touched.size()));
      detail::throwOnDeviceError(fnv_index_set_live_nodes(_device_index, cur + batch));
      row += batch;
    }
    if (device_wiring) {  // bring every link row home (old nodes gained back-links too)
      (void)first_device_node;
      link_rows.resize(_cur_num_nodes * _M);
      detail::throwOnDeviceError(fnv_index_read_links(_device_index, 0, _cur_num_nodes, link_rows.data()));
      auto unpackRow = [&](uint32_t n) { std::memcpy(nodeLinks(n), link_rows.data() + static_cast<uint64_t>(n) * _M, _M * sizeof(node_id_t)); };
      if (_num_threads == 1) for (uint32_t n = 0; n < _cur_num_nodes; ++n) unpackRow(n);
      else flatnav::executeInParallel(0, static_cast<uint32_t>(_cur_num_nodes), _num_threads, unpackRow);
    }
    // the device copy was kept in step; replicas on other GPUs were not
    _dirty_rows.drain(0);
    _device_synced_nodes = _cur_num_nodes;
    _device_stale = _device_rebuild = false;
    _replicas_stale = true;
    rebuild_guard.armed = false;
  }

  void add(void* data, label_t& label, int ef_construction, int num_initializations) {
    if (_cur_num_nodes >= _max_node_count)
      throw std::runtime_error(
          "Maximum number of nodes reached. Consider increasing the `max_node_count` parameter to create a larger "
          "index.");
    node_id_t entry, new_id;
    {
      std::lock_guard<std::mutex> g(_index_data_guard);
      entry = pickEntry(data, num_initializations);
      allocateNode(data, label, new_id);
    }
    if (new_id == 0) return;
    auto scratch = borrowScratch();
    insertionBeam(data, entry, ef_construction, *scratch);
    pruneNeighbors(scratch->beam, std::max(static_cast<int>(_M / 2), 1), *scratch);
    linkNeighbors(scratch->beam, new_id, *scratch);
    returnScratch(std::move(scratch));
  }

  // ---- search: always on the GPU -------------------------------------------------------------
  // Batched search (new; what bindings.cpp:161-228 does with a host loop).  queries = nq rows of
  // `dimension` elements of the index data type.  out_count may be null.  Returns nothing; rows
  // with fewer than K reachable results are padded with (+inf, -1) and flagged in out_count.
  void searchBatch(const void* queries, uint64_t nq, int K, int ef_search, int num_initializations, float* out_dist,
                   label_t* out_labels, int32_t* out_count = nullptr) {
    if (num_initializations <= 0) throw std::invalid_argument("num_initializations must be greater than 0.");
    std::lock_guard<std::mutex> g(_device_guard);
    ensureDevice();
    std::vector<uint64_t> ndist;
    if (_collect_stats) ndist.resize(nq);
    // small batches stay on the primary GPU; larger ones are sharded over every replica
    const bool spread = _device_list.size() > 1 && nq >= 64 * _device_list.size();
    if (spread) ensureReplicas();
    if (spread && !_replicas.empty()) {
      std::vector<fnv_index_t> all{_device_index};
      all.insert(all.end(), _replicas.begin(), _replicas.end());
      detail::throwOnDeviceError(fnv_search_batch_multi(all.data(), static_cast<int>(all.size()), queries, nq, K, ef_search,
                                                        num_initializations, out_dist, reinterpret_cast<int32_t*>(out_labels),
                                                        out_count, _collect_stats ? ndist.data() : nullptr, nullptr));
    } else {
      detail::throwOnDeviceError(fnv_search_batch(_device_index, queries, nq, K, ef_search, num_initializations, out_dist,
                                                  reinterpret_cast<int32_t*>(out_labels), out_count,
                                                  _collect_stats ? ndist.data() : nullptr, nullptr));
    }
    if (_collect_stats) {
      // reference accounting: + num_initializations per query (Index.h:857-859), + 1 per neighbour
      // evaluation (Index.h:689-691)
      uint64_t total = static_cast<uint64_t>(num_initializations) * nq;
      for (uint64_t v : ndist) total += v;
      _distance_computations.fetch_add(total);
    }
  }

  // Reference Index::search (Index.h:387-409): up to K (distance, label) pairs, ascending.
  std::vector<dist_label_t> search(const void* query, const int K, int ef_search, int num_initializations = 100) {
    std::vector<float> dist(static_cast<size_t>(std::max(K, 0)));
    std::vector<label_t> labels(static_cast<size_t>(std::max(K, 0)));
    int32_t count = 0;
    searchBatch(query, 1, K, ef_search, num_initializations, dist.data(), labels.data(), &count);
    std::vector<dist_label_t> results;
    results.reserve(static_cast<size_t>(count));
    for (int i = 0; i < count; ++i) results.emplace_back(dist[static_cast<size_t>(i)], labels[static_cast<size_t>(i)]);
    return results;
  }

  // ---- reordering (reference Index.h:412-440, 872-926) ---------------------------------------
  void doGraphReordering(const std::vector<std::string>& reordering_methods) {
    for (const auto& method : reordering_methods) {
      auto table = getGraphOutdegreeTable();
      std::vector<node_id_t> perm;
      if (method == "gorder") perm = util::gOrder<node_id_t>(table, 5);
      else if (method == "rcm") perm = util::rcmOrder<node_id_t>(table);
      else throw std::invalid_argument("Invalid reordering method: " + method);
      relabel(perm);
    }
  }
  void reorderGOrder(const int window_size = 5) {
    auto table = getGraphOutdegreeTable();
    relabel(util::gOrder<node_id_t>(table, window_size));
  }
  void reorderRCM() {
    auto table = getGraphOutdegreeTable();
    relabel(util::rcmOrder<node_id_t>(table));
  }

  // ---- persistence (reference Index.h:442-490) -----------------------------------------------
  static std::unique_ptr<Index<dist_t, label_t>> loadIndex(const std::string& filename) {
    std::ifstream stream(filename, std::ios::binary);
    if (!stream.is_open()) throw std::runtime_error("Unable to open file for reading: " + filename);
    flatnav::util::BinaryReader archive(stream);
    std::unique_ptr<Index<dist_t, label_t>> index(new Index<dist_t, label_t>());
    std::unique_ptr<DistanceInterface<dist_t>> dist = std::make_unique<dist_t>();
    archive(index->_data_type, index->_M, index->_data_size_bytes, index->_node_size_bytes, index->_max_node_count,
            index->_cur_num_nodes, *dist);
    if (index->_node_size_bytes != index->_data_size_bytes + sizeof(node_id_t) * index->_M + sizeof(label_t) ||
        index->_cur_num_nodes > index->_max_node_count)
      throw std::runtime_error("Corrupt index header: " + filename);
    index->_distance = std::move(dist);
    index->_num_threads = std::max<uint32_t>(1, std::thread::hardware_concurrency() / 2);
    index->allocateStore();
    archive(flatnav::util::binary_data(index->_index_memory.get(), index->storeBytes()));
    return index;
  }

  void saveIndex(const std::string& filename) {
    std::ofstream stream(filename, std::ios::binary);
    if (!stream.is_open()) throw std::runtime_error("Unable to open file for writing: " + filename);
    flatnav::util::BinaryWriter archive(stream);
    archive(*this);
  }

  // ---- knobs and getters (reference Index.h:492-548) -----------------------------------------
  inline void setNumThreads(uint32_t num_threads) {
    if (num_threads == 0 || num_threads > std::thread::hardware_concurrency())
      throw std::invalid_argument(
          "Number of threads must be greater than 0 and less than or equal to the number of hardware threads.");
    _num_threads = num_threads;
  }
  inline uint32_t getNumThreads() const { return _num_threads; }
  inline uint64_t getTotalIndexMemory() const { return storeBytes(); }
  inline uint64_t mutexesAllocatedMemory() const { return static_cast<uint64_t>(_max_node_count); }
  inline uint64_t visitedSetPoolAllocatedMemory() const {
    return static_cast<uint64_t>(_scratch_pool.size()) * _max_node_count * sizeof(uint32_t);
  }
  inline size_t maxEdgesPerNode() const { return _M; }
  inline size_t dataSizeBytes() const { return _data_size_bytes; }
  inline size_t nodeSizeBytes() const { return _node_size_bytes; }
  inline size_t maxNodeCount() const { return _max_node_count; }
  inline size_t currentNumNodes() const { return _cur_num_nodes; }
  inline size_t dataDimension() const { return _distance->dimension(); }
  inline uint64_t distanceComputations() const { return _distance_computations.load(); }
  inline DataType getDataType() const { return _data_type; }
  inline const char* rawIndexMemory() const { return _index_memory.get(); }

  void resetStats() {
    _distance_computations = 0;
    _metric_hops = 0;
  }

  void getIndexSummary() const {
    std::cout << "\nIndex Parameters\n-----------------------------\n"
              << "max_edges_per_node (M): " << _M << "\n"
              << "data_size_bytes: " << _data_size_bytes << "\n"
              << "node_size_bytes: " << _node_size_bytes << "\n"
              << "max_node_count: " << _max_node_count << "\n"
              << "cur_num_nodes: " << _cur_num_nodes << "\n"
              << std::flush;
    _distance->getSummary();
  }

 private:
  // Apply a permutation (perm[old] = new id): rewrite every link, then move the node records with
  // one gather into a fresh store (reference relabel(), Index.h:872-926, permutes in place).
  void relabel(const std::vector<node_id_t>& perm) {
    if (perm.size() != _cur_num_nodes) throw std::runtime_error("reordering permutation has the wrong size");
    for (node_id_t n = 0; n < _cur_num_nodes; ++n) {
      node_id_t* links = nodeLinks(n);
      for (size_t m = 0; m < _M; ++m) links[m] = perm[links[m]];
    }
    std::unique_ptr<char[]> fresh(new char[storeBytes()]());
    for (node_id_t n = 0; n < _cur_num_nodes; ++n)
      std::memcpy(fresh.get() + static_cast<uint64_t>(perm[n]) * _node_size_bytes, nodeData(n), _node_size_bytes);
    _index_memory = std::move(fresh);
    markDeviceRebuild();
  }
};

}  // namespace flatnav
