// Host-side check that flatnav_amd/csrc/stl_exact.h performs exactly the element
// moves of libstdc++'s std::priority_queue / std::sort (the reference's containers,
// flatnav/index/Index.h:47-53, 402-403), including under heavy ties.
// Built and run by tests/test_stl_exact.py.  Exit code 0 = identical everywhere.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <random>
#include <utility>
#include <vector>

#include "../include/flatnav/util/StlExact.h"

typedef std::pair<float, uint32_t> P;
struct CompareByFirst {
  bool operator()(P const& a, P const& b) const noexcept { return a.first < b.first; }
};
struct PQ : std::priority_queue<P, std::vector<P>, CompareByFirst> {
  const std::vector<P>& raw() const { return c; }
};
struct Arr {
  std::vector<fnv_stl::Entry> v;
  fnv_stl::Entry get(int i) const { return v[i]; }
  void set(int i, fnv_stl::Entry e) { v[i] = e; }
};

static bool same(const PQ& pq, const Arr& a, int n) {
  const auto& c = pq.raw();
  if ((int)c.size() != n) return false;
  for (int i = 0; i < n; i++)
    if (c[i].first != a.v[i].key || c[i].second != a.v[i].val) return false;
  return true;
}

int main() {
  std::mt19937 rng(12345);
  // ---- heaps: random emplace/pop streams at several tie densities -------------
  for (int range : {1, 2, 4, 16, 256, 100000}) {
    for (int rep = 0; rep < 40; rep++) {
      PQ pq;
      Arr a;
      a.v.resize(5000);
      int n = 0;
      uint32_t next_id = 0;
      bool negate = rep & 1;
      for (int op = 0; op < 3000; op++) {
        bool push = n == 0 || (rng() % 100) < 55;
        if (push) {
          float k = (float)(rng() % range);
          if (negate) k = -k;
          pq.emplace(k, next_id);
          fnv_stl::heap_push(a, n, fnv_stl::Entry{k, next_id});
          n++;
          next_id++;
        } else {
          pq.pop();
          fnv_stl::heap_pop(a, n);
          n--;
        }
        if (!same(pq, a, n)) {
          std::printf("heap mismatch range=%d rep=%d op=%d\n", range, rep, op);
          return 1;
        }
      }
      // bounded-beam pattern: push then pop when over capacity
      while (n > 0) {
        if (pq.top().first != a.v[0].key || pq.top().second != a.v[0].val) return 2;
        pq.pop();
        fnv_stl::heap_pop(a, n);
        n--;
        if (!same(pq, a, n)) return 3;
      }
    }
  }
  // ---- sort: sizes around the introsort thresholds, several tie densities -----
  for (int range : {1, 2, 3, 8, 64, 1000, 1000000}) {
    for (int n = 0; n <= 700; n += (n < 40 ? 1 : 13)) {
      for (int rep = 0; rep < 6; rep++) {
        std::vector<P> ref(n);
        Arr a;
        a.v.resize(n);
        for (int i = 0; i < n; i++) {
          float k = (float)(rng() % range);
          ref[i] = P(k, (uint32_t)i);
        }
        if (rep == 1) std::sort(ref.begin(), ref.end(), [](const P& l, const P& r) { return l.first > r.first; });
        if (rep == 2) std::stable_sort(ref.begin(), ref.end(), [](const P& l, const P& r) { return l.first < r.first; });
        for (int i = 0; i < n; i++) a.v[i] = fnv_stl::Entry{ref[i].first, ref[i].second};
        std::sort(ref.begin(), ref.end(), [](const P& l, const P& r) { return l.first < r.first; });
        fnv_stl::sort_by_key(a, n);
        for (int i = 0; i < n; i++)
          if (ref[i].first != a.v[i].key || ref[i].second != a.v[i].val) {
            std::printf("sort mismatch range=%d n=%d rep=%d at %d\n", range, n, rep, i);
            return 4;
          }
      }
    }
  }
  // ---- adversarial: organ-pipe / sawtooth inputs that push introsort to heapsort
  for (int n : {200, 1000, 5000}) {
    std::vector<P> ref(n);
    Arr a;
    a.v.resize(n);
    for (int i = 0; i < n; i++) ref[i] = P((float)((i % 2) ? i : n - i), (uint32_t)i);
    for (int i = 0; i < n; i++) a.v[i] = fnv_stl::Entry{ref[i].first, ref[i].second};
    std::sort(ref.begin(), ref.end(), [](const P& l, const P& r) { return l.first < r.first; });
    fnv_stl::sort_by_key(a, n);
    for (int i = 0; i < n; i++)
      if (ref[i].first != a.v[i].key || ref[i].second != a.v[i].val) return 5;
  }
  std::printf("stl_exact: OK\n");
  return 0;
}
