// flatnav/util/Reordering.h -- node relabelling heuristics of the host API (own implementation).
//
// Same entry points and contract as the reference (include/flatnav/util/Reordering.h:27-200):
//   gOrder<node_id_t>(outdegree_table, w)  and  rcmOrder<node_id_t>(outdegree_table)
// take table[node] = list of out-neighbours and return P with P[i] = NEW id of node i.
// Out of the GPU hot path: these run once on the CPU.  A relabelling keeps the
// graph isomorphic but moves which nodes the entry-point scan samples, so search results after a
// reorder are statistically -- not bitwise -- equivalent (same in the reference).
//
// gOrder: greedy sliding-window ordering (Wei et al., "Speedup Graph Processing by Graph
// Ordering"): repeatedly append the unplaced node with the highest locality score against the
// last `w` placed nodes; score(v) counts, for each window node u, edges u->v, edges v->u and
// common in-neighbours.  Scores change by +-1, so a bucket queue gives O(1) updates.
// rcmOrder: reverse Cuthill-McKee -- breadth-first from low-degree roots, neighbours expanded in
// ascending degree, final order reversed.
#pragma once
#include <algorithm>
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

namespace flatnav::util {

namespace detail {

// Max-priority bucket queue over nodes 0..n-1 whose keys only ever move by one.
// Buckets are doubly linked lists threaded through prev/next; pop() returns any node of the
// highest non-empty bucket (most recently inserted first).
template <typename node_id_t>
class UnitStepQueue {
  static constexpr int64_t NIL = -1;
  std::vector<int64_t> _prev, _next, _head;  // _head[key - _lo]
  std::vector<int> _key;
  std::vector<char> _present;
  int _lo, _top;

  std::size_t slot(int key) {
    if (key < _lo) {  // grow downwards
      _head.insert(_head.begin(), static_cast<std::size_t>(_lo - key), NIL);
      _lo = key;
    }
    std::size_t s = static_cast<std::size_t>(key - _lo);
    if (s >= _head.size()) _head.resize(s + 1, NIL);
    return s;
  }
  void unlink(node_id_t v) {
    const int64_t p = _prev[v], n = _next[v];
    if (p != NIL) _next[static_cast<std::size_t>(p)] = n;
    else _head[static_cast<std::size_t>(_key[v] - _lo)] = n;
    if (n != NIL) _prev[static_cast<std::size_t>(n)] = p;
  }
  void link(node_id_t v) {
    const std::size_t s = slot(_key[v]);
    _prev[v] = NIL;
    _next[v] = _head[s];
    if (_head[s] != NIL) _prev[static_cast<std::size_t>(_head[s])] = static_cast<int64_t>(v);
    _head[s] = static_cast<int64_t>(v);
    if (_key[v] > _top) _top = _key[v];
  }

 public:
  explicit UnitStepQueue(std::size_t n) : _prev(n, NIL), _next(n, NIL), _key(n, 0), _present(n, 1), _lo(0), _top(0) {
    _head.assign(1, NIL);
    for (std::size_t v = n; v-- > 0;) link(static_cast<node_id_t>(v));  // node 0 ends up first
  }
  void bump(node_id_t v, int delta) {
    if (!_present[v]) return;
    unlink(v);
    _key[v] += delta;
    link(v);
  }
  node_id_t pop() {
    while (_head[slot(_top)] == NIL) --_top;
    const node_id_t v = static_cast<node_id_t>(_head[static_cast<std::size_t>(_top - _lo)]);
    unlink(v);
    _present[v] = 0;
    return v;
  }
};

}  // namespace detail

template <typename node_id_t>
std::vector<node_id_t> gOrder(std::vector<std::vector<node_id_t>>& outdegree_table, const int w) {
  const std::size_t n = outdegree_table.size();
  std::vector<node_id_t> new_id(n, 0);
  if (n == 0) return new_id;
  std::vector<std::vector<node_id_t>> in_edges(n);
  for (std::size_t u = 0; u < n; ++u)
    for (node_id_t v : outdegree_table[u]) in_edges[v].push_back(static_cast<node_id_t>(u));

  detail::UnitStepQueue<node_id_t> queue(n);
  std::vector<node_id_t> order(n);
  // score contribution of a window node `x` to every other node: +-1 per edge x->u, per edge
  // u->x, and per shared in-neighbour relation (u->x and u->v).
  auto touch = [&](node_id_t x, int delta) {
    for (node_id_t u : outdegree_table[x]) queue.bump(u, delta);
    for (node_id_t u : in_edges[x]) {
      queue.bump(u, delta);
      for (node_id_t v : outdegree_table[u]) queue.bump(v, delta);
    }
  };
  queue.bump(0, 1);  // seed with node 0
  order[0] = queue.pop();
  for (std::size_t i = 1; i < n; ++i) {
    touch(order[i - 1], +1);
    if (i > static_cast<std::size_t>(w) + 1) touch(order[i - static_cast<std::size_t>(w) - 1], -1);
    order[i] = queue.pop();
  }
  for (std::size_t pos = 0; pos < n; ++pos) new_id[order[pos]] = static_cast<node_id_t>(pos);
  return new_id;
}

template <typename node_id_t>
std::vector<node_id_t> rcmOrder(std::vector<std::vector<node_id_t>>& outdegree_table) {
  const std::size_t n = outdegree_table.size();
  std::vector<std::size_t> degree(n);
  std::vector<node_id_t> roots(n);
  for (std::size_t v = 0; v < n; ++v) {
    degree[v] = outdegree_table[v].size();
    roots[v] = static_cast<node_id_t>(v);
  }
  auto by_degree = [&](node_id_t a, node_id_t b) { return degree[a] < degree[b]; };
  std::stable_sort(roots.begin(), roots.end(), by_degree);

  std::vector<char> placed(n, 0);
  std::vector<node_id_t> order;
  order.reserve(n);
  std::vector<node_id_t> fringe;
  for (node_id_t root : roots) {
    if (placed[root]) continue;
    std::queue<node_id_t> bfs;
    bfs.push(root);
    while (!bfs.empty()) {
      const node_id_t v = bfs.front();
      bfs.pop();
      if (placed[v]) continue;
      placed[v] = 1;
      order.push_back(v);
      fringe.assign(outdegree_table[v].begin(), outdegree_table[v].end());
      std::stable_sort(fringe.begin(), fringe.end(), by_degree);
      for (node_id_t u : fringe)
        if (!placed[u]) bfs.push(u);
    }
  }
  std::reverse(order.begin(), order.end());
  std::vector<node_id_t> new_id(n, 0);
  for (std::size_t pos = 0; pos < n; ++pos) new_id[order[pos]] = static_cast<node_id_t>(pos);
  return new_id;
}

}  // namespace flatnav::util
