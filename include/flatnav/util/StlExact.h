// flatnav/util/StlExact.h -- operation-for-operation restatement of the libstdc++ binary
// heap and introsort routines that the reference's search leans on, written so
// the SAME code runs on the host (unit-tested against the real std:: calls) and
// inside the HIP kernel.
//
// Why this exists: the reference keeps its beam in two std::priority_queue
// objects that compare on distance only (flatnav/index/Index.h:47-53) and ends
// with an unstable std::sort on distance only (Index.h:402-403).  Whenever two
// distances tie -- common on integer-valued data such as SIFT -- WHICH node is
// evicted, popped first, or survives the final truncate-to-K is decided by the
// element moves of these library routines, not by node ids.  Bit-exact ids
// therefore need the same moves.  Routines follow GCC 11 libstdc++
// bits/stl_heap.h (__push_heap :128-146, __adjust_heap :214-250, __pop_heap
// :253-266) and bits/stl_algo.h (__move_median_to_first :78-110,
// __unguarded_partition :1824-1846, __introsort_loop :1941-1963,
// __insertion_sort :1832-1853, __unguarded_linear_insert :1812-1829,
// __final_insertion_sort :1877-1893, __partial_sort/__heap_select for the
// depth-limit fallback), specialised to 8-byte {key, payload} entries compared
// on `key` only with operator< on float.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FNV_HD __host__ __device__ __forceinline__
#else
#define FNV_HD inline
#endif

namespace fnv_stl {

struct Entry {
  float key;      // distance (or negated distance in the candidates heap)
  uint32_t val;   // node id (or label in the final sort)
};

// ---- std::push_heap after emplace_back: `n` = size BEFORE the push ---------
// A is any random-access "array" type offering  Entry get(int) / void set(int, Entry).
template <class A>
FNV_HD void heap_push(A& a, int n, Entry v) {
  int hole = n;
  while (hole > 0) {
    int parent = (hole - 1) / 2;
    Entry p = a.get(parent);
    if (!(p.key < v.key)) break;
    a.set(hole, p);
    hole = parent;
  }
  a.set(hole, v);
}

// __adjust_heap(first, hole, len, value) followed by its trailing __push_heap.
template <class A>
FNV_HD void adjust_heap(A& a, int hole, int len, Entry v, int base = 0) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    Entry r = a.get(base + child);
    Entry l = a.get(base + child - 1);
    if (r.key < l.key) {
      child--;
      r = l;
    }
    a.set(base + hole, r);
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a.set(base + hole, a.get(base + child - 1));
    hole = child - 1;
  }
  // __push_heap(first, hole, top, value)
  while (hole > top) {
    int parent = (hole - 1) / 2;
    Entry p = a.get(base + parent);
    if (!(p.key < v.key)) break;
    a.set(base + hole, p);
    hole = parent;
  }
  a.set(base + hole, v);
}

// ---- priority_queue::pop(): std::pop_heap + pop_back; `n` = size BEFORE -----
// After the call the heap occupies [0, n-1) and slot n-1 holds the old top
// (exactly what std::pop_heap leaves there).
template <class A>
FNV_HD void heap_pop(A& a, int n) {
  if (n > 1) {
    Entry v = a.get(n - 1);
    a.set(n - 1, a.get(0));
    adjust_heap(a, 0, n - 1, v);
  }
}

// ---------------------------------------------------------------------------
// std::sort(first, last, [](l, r){ return l.key < r.key; })
// ---------------------------------------------------------------------------
template <class A>
FNV_HD void swap_entries(A& a, int i, int j) {
  Entry t = a.get(i);
  a.set(i, a.get(j));
  a.set(j, t);
}

template <class A>
FNV_HD void move_median_to_first(A& a, int result, int ia, int ib, int ic) {
  float ka = a.get(ia).key, kb = a.get(ib).key, kc = a.get(ic).key;
  if (ka < kb) {
    if (kb < kc) swap_entries(a, result, ib);
    else if (ka < kc) swap_entries(a, result, ic);
    else swap_entries(a, result, ia);
  } else if (ka < kc) swap_entries(a, result, ia);
  else if (kb < kc) swap_entries(a, result, ic);
  else swap_entries(a, result, ib);
}

template <class A>
FNV_HD int unguarded_partition(A& a, int first, int last, int pivot) {
  while (true) {
    float pk = a.get(pivot).key;
    while (a.get(first).key < pk) ++first;
    --last;
    while (pk < a.get(last).key) --last;
    if (!(first < last)) return first;
    swap_entries(a, first, last);
    ++first;
  }
}

template <class A>
FNV_HD void unguarded_linear_insert(A& a, int last) {
  Entry v = a.get(last);
  int next = last - 1;
  while (true) {
    Entry nx = a.get(next);
    if (!(v.key < nx.key)) break;
    a.set(last, nx);
    last = next;
    --next;
  }
  a.set(last, v);
}

template <class A>
FNV_HD void insertion_sort(A& a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    Entry v = a.get(i);
    if (v.key < a.get(first).key) {
      for (int j = i; j > first; --j) a.set(j, a.get(j - 1));  // move_backward(first, i, i+1)
      a.set(first, v);
    } else {
      unguarded_linear_insert(a, i);
    }
  }
}

// __partial_sort(first, last, last): heap-select over the whole range then sort_heap.
template <class A>
FNV_HD void heapsort_range(A& a, int first, int last) {
  int len = last - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      Entry v = a.get(first + parent);
      adjust_heap(a, parent, len, v, first);
      if (parent == 0) break;
      parent--;
    }
  }
  // __heap_select's loop over [middle, last) is empty (middle == last).
  while (last - first > 1) {  // __sort_heap
    --last;
    Entry v = a.get(last);
    a.set(last, a.get(first));
    adjust_heap(a, 0, last - first, v, first);
  }
}

FNV_HD int lg2_floor(int n) {  // std::__lg
  int k = 0;
  while (n > 1) {
    n >>= 1;
    k++;
  }
  return k;
}

// Explicit-stack form of __introsort_loop (the library recurses on the right
// part and loops on the left; the stack keeps the pending LEFT parts... no:
// it recurses into [cut,last) first and then continues with [first,cut)).  We
// reproduce that order exactly: process right part to completion before the
// left part.  The order in which disjoint sub-ranges are processed does not
// change the result (each call only touches its own range), but we keep it
// anyway.
// `frames`: caller-provided storage for the explicit stack, 3 ints per frame, `cap` frames.  The library's recursion
// depth is bounded by the depth limit 2*lg(n) (+1 for the frame in flight), so 64 frames cover every int-sized n.
// The device kernel hands in LDS (a local array would live in scratch memory).
template <class A>
FNV_HD void sort_by_key(A& a, int n, int* frames, int cap) {
  if (n <= 0) return;
  const int THRESH = 16;
  int sp = 0;
  frames[0] = 0;
  frames[1] = n;
  frames[2] = 2 * lg2_floor(n);
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = frames[3 * sp], last = frames[3 * sp + 1], depth = frames[3 * sp + 2];
    while (last - first > THRESH) {
      if (depth == 0) {
        heapsort_range(a, first, last);
        break;
      }
      --depth;
      int mid = first + (last - first) / 2;
      move_median_to_first(a, first, first + 1, mid, last - 1);
      int cut = unguarded_partition(a, first + 1, last, first);
      // library: recurse(cut, last, depth); last = cut;  -> right part first.
      // Push the LEFT remainder, continue with the right part now.
      if (sp < cap) {
        frames[3 * sp] = first;
        frames[3 * sp + 1] = cut;
        frames[3 * sp + 2] = depth;
        ++sp;
      }
      first = cut;
    }
  }
  // __final_insertion_sort
  if (n > THRESH) {
    insertion_sort(a, 0, THRESH);
    for (int i = THRESH; i != n; ++i) unguarded_linear_insert(a, i);
  } else {
    insertion_sort(a, 0, n);
  }
}

template <class A>
FNV_HD void sort_by_key(A& a, int n) {
  int frames[3 * 64];
  sort_by_key(a, n, frames, 64);
}

}  // namespace fnv_stl
