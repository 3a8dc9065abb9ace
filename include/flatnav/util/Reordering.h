// flatnav/util/Reordering.h -- node relabelling heuristics of the host API (own implementation).
//
// Same entry points and contract as the reference (include/flatnav/util/Reordering.h:27-200):
//   gOrder<node_id_t>(outdegree_table, w)  and  rcmOrder<node_id_t>(outdegree_table)
// take table[node] = list of out-neighbours and return P with P[i] = NEW id of node i.
// Out of the GPU hot path: these run once on the CPU.
//
// Round 6: both return the SAME PERMUTATION as the reference, element for element -- pinned against the reference's own
// header compiled where it lies (oracle/_ref: ref_gorder / ref_rcm; tests/test_reference_pins.py).  What makes a
// permutation "the reference's" is how equal scores / equal degrees are resolved:
//
// gOrder (Wei et al., "Speedup Graph Processing by Graph Ordering"): greedy sliding window of w placed nodes; the score of
// an unplaced node counts, per window node x, edges x->v, edges v->x and shared in-neighbours.  The reference keeps the
// unplaced nodes in ONE array sorted by score (GorderPriorityQueue.h:14-109): +1 swaps the node with the LAST entry of
// its score run, -1 with the FIRST, pop takes the array's back.  `ScoreRuns` below holds the same array, and instead of
// the reference's two binary searches and hash map it keeps where each score's run starts and how long it is, so every
// operation is O(1) and lands on the same position.
//
// rcmOrder: reverse Cuthill-McKee: roots by ascending out-degree, breadth-first, every node's out-neighbours enqueued by
// ascending degree, final order reversed.  Equal degrees: the reference sorts (id, degree) pairs with std::sort on the
// degree alone (Reordering.h:133-136, 154-157, 176-179) -- introsort is a function of the comparison outcomes only, so
// std::sort over the ids with the same "degree less" predicate leaves equal-degree ids in the same places.
#pragma once
#include <algorithm>
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

namespace flatnav::util {

namespace detail {

// The unplaced nodes in one array ordered by ascending score; entries of equal score form a run.
template <typename node_id_t>
class ScoreRuns {
  std::vector<node_id_t> _at;    // position -> node
  std::vector<int64_t> _where;   // node -> position, -1 once placed
  std::vector<int> _score;       // node -> score
  // score s lives at runs[s - _base]: first position of its run, entries in it (first is meaningful while count > 0)
  struct Run {
    std::size_t first = 0, count = 0;
  };
  std::vector<Run> _runs;
  int _base = 0;
  std::size_t _size;

  Run& run(int s) {
    if (s < _base) {
      _runs.insert(_runs.begin(), static_cast<std::size_t>(_base - s), Run());
      _base = s;
    }
    const std::size_t k = static_cast<std::size_t>(s - _base);
    if (k >= _runs.size()) _runs.resize(k + 1);
    return _runs[k];
  }
  void exchange(std::size_t a, std::size_t b) {
    std::swap(_at[a], _at[b]);
    _where[_at[a]] = static_cast<int64_t>(a);
    _where[_at[b]] = static_cast<int64_t>(b);
  }

 public:
  explicit ScoreRuns(std::size_t n) : _at(n), _where(n), _score(n, 0), _runs(1), _size(n) {
    for (std::size_t v = 0; v < n; ++v) {
      _at[v] = static_cast<node_id_t>(v);
      _where[v] = static_cast<int64_t>(v);
    }
    _runs[0].count = n;
  }
  // score +1: the node trades places with the last entry of its run and becomes the first of the next run
  void raise(node_id_t v) {
    if (_where[v] < 0) return;
    const int s = _score[v];
    Run& from = run(s);
    const std::size_t j = from.first + from.count - 1;
    exchange(static_cast<std::size_t>(_where[v]), j);
    from.count--;
    Run& to = run(s + 1);  // (may reallocate: `from` is not used below)
    to.first = j;
    to.count++;
    _score[v] = s + 1;
  }
  // score -1: the node trades places with the first entry of its run and becomes the last of the run below
  void lower(node_id_t v) {
    if (_where[v] < 0) return;
    const int s = _score[v];
    const std::size_t j = run(s).first;
    exchange(static_cast<std::size_t>(_where[v]), j);
    Run& below = run(s - 1);  // (may reallocate)
    if (below.count == 0) below.first = j;
    below.count++;
    Run& from = run(s);
    from.first = j + 1;
    from.count--;
    _score[v] = s - 1;
  }
  // the back of the array: the last entry of the highest run
  node_id_t take_back() {
    const node_id_t v = _at[--_size];
    run(_score[v]).count--;
    _where[v] = -1;
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

  detail::ScoreRuns<node_id_t> unplaced(n);
  std::vector<node_id_t> order(n);
  // what a node x entering (up = true) or leaving the window does to every unplaced node's score: one step per edge x->u,
  // per edge u->x, and per out-edge of such a u (u is a shared in-neighbour) -- in this order, which decides the ties
  auto window_change = [&](node_id_t x, bool up) {
    auto step = [&](node_id_t v) { up ? unplaced.raise(v) : unplaced.lower(v); };
    for (node_id_t u : outdegree_table[x]) step(u);
    for (node_id_t u : in_edges[x]) {
      step(u);
      for (node_id_t v : outdegree_table[u]) step(v);
    }
  };
  unplaced.raise(0);  // node 0 seeds the order
  order[0] = unplaced.take_back();
  for (std::size_t i = 1; i < n; ++i) {
    window_change(order[i - 1], true);
    // (signed, as in the reference: a negative w makes every step i > w + 1 drop order[i - w - 1])
    if (static_cast<int64_t>(i) > static_cast<int64_t>(w) + 1)
      window_change(order[static_cast<std::size_t>(static_cast<int64_t>(i) - w - 1)], false);
    order[i] = unplaced.take_back();
  }
  for (std::size_t pos = 0; pos < n; ++pos) new_id[order[pos]] = static_cast<node_id_t>(pos);
  return new_id;
}

template <typename node_id_t>
std::vector<node_id_t> rcmOrder(std::vector<std::vector<node_id_t>>& outdegree_table) {
  const std::size_t n = outdegree_table.size();
  std::vector<int> degree(n);
  std::vector<node_id_t> roots(n);
  for (std::size_t v = 0; v < n; ++v) {
    degree[v] = static_cast<int>(outdegree_table[v].size());
    roots[v] = static_cast<node_id_t>(v);
  }
  auto by_degree = [&](node_id_t a, node_id_t b) { return degree[a] < degree[b]; };
  std::sort(roots.begin(), roots.end(), by_degree);  // NOT stable_sort: equal degrees as introsort leaves them (see top)

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
      // every out-neighbour is sorted (a placed one takes part in the sort's comparisons), the placed ones are skipped
      // when they come up -- dropping them here instead would change where equal degrees land
      fringe.assign(outdegree_table[v].begin(), outdegree_table[v].end());
      std::sort(fringe.begin(), fringe.end(), by_degree);
      for (node_id_t u : fringe) bfs.push(u);
    }
  }
  std::reverse(order.begin(), order.end());
  std::vector<node_id_t> new_id(n, 0);
  for (std::size_t pos = 0; pos < n; ++pos) new_id[order[pos]] = static_cast<node_id_t>(pos);
  return new_id;
}

}  // namespace flatnav::util
