// heaps.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip).
// Heap entry packing, phase timers, and the wave-cooperative libstdc++-exact heap operations.
#pragma once
#include "search_params.h"
namespace fnv_dev {

// ---------------------------------------------------------------------------------------------
// Heaps: 8-byte entries {float key | uint32 id} packed in one 64-bit LDS word.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long pack(fnv_stl::Entry e) {
  return (unsigned long long)__float_as_uint(e.key) | ((unsigned long long)e.val << 32);
}
__device__ __forceinline__ fnv_stl::Entry unpack(unsigned long long v) {
  fnv_stl::Entry e;
  e.key = __uint_as_float((uint32_t)v);
  e.val = (uint32_t)(v >> 32);
  return e;
}

// The array starts 8 bytes past a 16-byte boundary, so the two children (2i+1, 2i+2) of any node
// form one aligned 16-byte pair: a single ds_read_b128 fetches both.
struct LdsHeap {
  unsigned long long* p;
  __device__ __forceinline__ fnv_stl::Entry get(int i) const { return unpack(p[i]); }
  __device__ __forceinline__ void set(int i, fnv_stl::Entry e) { p[i] = pack(e); }
  __device__ __forceinline__ bool leftChildWins(int i) const {  // key[2i+2] < key[2i+1]
    const uint4 c = *reinterpret_cast<const uint4*>(p + 2 * i + 1);
    return __uint_as_float(c.z) < __uint_as_float(c.x);
  }
  // Predicated store without touching EXEC: lanes with cond == false write into the scratch word just
  // below the array (p[-1]: the 8 bytes that pad the array to its 16n+8 start).  One VALU select instead of a
  // scalar saveexec / branch / restore sequence -- the scalar unit is shared by every wave of the CU.
  __device__ __forceinline__ void set_if(bool cond, int i, fnv_stl::Entry e) { p[cond ? i : -1] = pack(e); }
};

// Candidates heap: first `cap` entries in LDS, the rest in a per-slot HBM spill area (rare; the
// kernel fences around operations that reach into it).
struct CandHeap {
  unsigned long long* p;
  unsigned long long* spill;
  int cap;
  __device__ __forceinline__ fnv_stl::Entry get(int i) const { return unpack(i < cap ? p[i] : spill[i - cap]); }
  __device__ __forceinline__ void set(int i, fnv_stl::Entry e) {
    if (i < cap) p[i] = pack(e);
    else spill[i - cap] = pack(e);
  }
  __device__ __forceinline__ bool leftChildWins(int i) const { return get(2 * i + 2).key < get(2 * i + 1).key; }
  __device__ __forceinline__ void set_if(bool cond, int i, fnv_stl::Entry e) {
    if (cond) set(i, e);
  }
};

// ---------------------------------------------------------------------------------------------
// Phase timing (profiling builds only: -DFNV_PHASE_TIMING).  mark(i) charges the shader cycles
// since the previous mark to phase i, after draining outstanding memory operations so that a phase
// owns its own latency.  In product builds the struct is empty and every call folds away.
// ---------------------------------------------------------------------------------------------
constexpr int NPHASE = 16;
#ifdef FNV_PHASE_TIMING
struct PhaseTimer {
  unsigned long long t[NPHASE];
  unsigned long long last;
  __device__ __forceinline__ void start() {
    for (int i = 0; i < NPHASE; i++) t[i] = 0;
    last = clock64();
  }
  __device__ __forceinline__ void mark(int i) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long n = clock64();
    t[i] += n - last;
    last = n;
  }
  __device__ __forceinline__ void flush(unsigned long long* out, int lane) {
    if (lane == 0 && out)
      for (int i = 0; i < NPHASE; i++) atomicAdd(&out[i], t[i]);
  }
};
#else
struct PhaseTimer {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void flush(unsigned long long*, int) {}
};
#endif
// -DFNV_ASM_MARKS drops named comments into the ISA (tools/isa_regions.py counts instructions between them)
#ifdef FNV_ASM_MARKS
#define ISA_MARK(name) asm volatile("; ##MARK " name ::: "memory")
#else
#define ISA_MARK(name)
#endif
#define PH_DECL PhaseTimer ph; ph.start();
#define PH_MARK(i) do { ph.mark(i); ISA_MARK("phase" #i); } while (0)
#define PH_FLUSH ph.flush(cold_args()->phase_cycles, lane)

// ---------------------------------------------------------------------------------------------
// Wave-cooperative forms of the two libstdc++ heap operations (same element moves as
// fnv_stl::heap_push / heap_pop in flatnav/util/StlExact.h, which tests/ check against the real
// std::priority_queue), executed by all 64 lanes with O(1) LDS round trips instead of one per level.
//
//  push(n, v): the hole climbs the ancestor chain a_k = ((n+1) >> k) - 1 of index n while
//    heap[a_k].key < v.key.  All ancestors are read at once (lane j reads a_{j+1}); a ballot of the
//    comparisons gives t = length of the leading run of "true"; lanes j < t move their ancestor one
//    level down, lane t stores v.
//  pop(n): __adjust_heap walks the hole from the root to a leaf always taking the larger child
//    (right unless right < left), then sifts the former last element v back up.  Which child wins at
//    node i depends only on the array, so every internal node is judged in parallel (ballot ->
//    one 64-bit mask per 64 nodes, parked in lane r of two VGPRs), the root-to-leaf path is then a
//    scalar walk over those masks (v_readlane, no memory), the path's values are fetched in one
//    parallel read, the sift-up length comes from one more ballot, and the surviving moves are one
//    parallel write.  Moves that the sequential code does and then undoes are simply not performed.
// All lanes must call these with wave-uniform arguments.
// ---------------------------------------------------------------------------------------------
// Ordering between the lanes of ONE wave: the LDS executes a wave's instructions in issue order,
// so a later read by any lane sees an earlier write by any other lane; only the compiler must be
// kept from reordering the accesses (it reasons per thread).  No hardware wait is emitted.
__device__ __forceinline__ void wave_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// Returns true when v ended up at the root (it climbed the whole ancestor chain, or the heap was empty).
template <class H>
__device__ __forceinline__ bool coop_push(H& h, int n, fnv_stl::Entry v, int lane, PhaseTimer& ph, int phbase) {
  n = __builtin_amdgcn_readfirstlane(n);
  v.key = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.key)));
  v.val = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.val);
  const uint32_t m1 = (uint32_t)n + 1u;
  const int depth = 31 - __clz((int)m1);  // number of ancestors of index n
  // lane j looks at ancestor a_{j+1} = (m1 >> (j+1)) - 1; lanes past the root re-read the root (harmless)
  const uint32_t sh = (uint32_t)min(lane + 1, depth);
  const fnv_stl::Entry anc = h.get(max((int)(m1 >> sh) - 1, 0));
  const unsigned long long run = __ballot(lane < depth && anc.key < v.key);
  const int t = __ffsll((long long)~run) - 1;  // lanes >= depth vote false, so t <= depth
  h.set_if(lane <= t, (int)(m1 >> lane) - 1, lane < t ? anc : v);  // lanes < t: ancestor one level down; lane t: v
  wave_sync();
  ph.mark(phbase);
  return t == depth;
}

// The same element pushed onto two independent heaps (the reference's candidates.emplace(-d, id) and
// neighbors.emplace(d, id), Index.h:696-697): both ancestor chains are read before either vote, so the two pushes share
// ONE LDS round trip.  Returns whether vb reached the root of hb.
template <class HA, class HB>
__device__ __forceinline__ bool coop_push2(HA& ha, int na, fnv_stl::Entry va, HB& hb, int nb, fnv_stl::Entry vb, int lane,
                                           PhaseTimer& ph, int phbase) {
  na = __builtin_amdgcn_readfirstlane(na);
  nb = __builtin_amdgcn_readfirstlane(nb);
  va.key = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(va.key)));
  va.val = (uint32_t)__builtin_amdgcn_readfirstlane((int)va.val);
  vb.key = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vb.key)));
  vb.val = (uint32_t)__builtin_amdgcn_readfirstlane((int)vb.val);
  const uint32_t ma = (uint32_t)na + 1u, mb = (uint32_t)nb + 1u;
  const int da = 31 - __clz((int)ma), db = 31 - __clz((int)mb);
  const fnv_stl::Entry anca = ha.get(max((int)(ma >> (uint32_t)min(lane + 1, da)) - 1, 0));
  const fnv_stl::Entry ancb = hb.get(max((int)(mb >> (uint32_t)min(lane + 1, db)) - 1, 0));
  const unsigned long long runa = __ballot(lane < da && anca.key < va.key);
  const unsigned long long runb = __ballot(lane < db && ancb.key < vb.key);
  const int ta = __ffsll((long long)~runa) - 1, tb = __ffsll((long long)~runb) - 1;
  ha.set_if(lane <= ta, (int)(ma >> lane) - 1, lane < ta ? anca : va);
  hb.set_if(lane <= tb, (int)(mb >> lane) - 1, lane < tb ? ancb : vb);
  wave_sync();
  ph.mark(phbase);
  return tb == db;
}

// One step of the root-to-leaf walk in 1-based numbering: m <- 2*m + mask[bit].  Two scalar instructions
// (bit test into SCC, add-with-carry); the compiler's own sequence is shift/and/shift/or on 64-bit operands.
__device__ __forceinline__ uint32_t walk_step(uint32_t m, unsigned long long mask, uint32_t bit) {
  asm("s_bitcmp1_b64 %1, %2\n\ts_addc_u32 %0, %0, %0" : "+s"(m) : "s"(mask), "s"(bit) : "scc");
  return m;
}

// KEEP_TOP: also park the removed top in the vacated slot, as std::pop_heap does (only the result tail
// needs that; the beam loop never looks at the slot again).
// Returns the key of the new root (what priority_queue::top() reads next) -- from registers, not from memory;
// meaningless when n <= 1.
template <bool KEEP_TOP, class H>
__device__ __forceinline__ float coop_pop(H& h, int n, int lane, PhaseTimer& ph, int phbase) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n <= 1) return 0.f;  // std::pop_heap does nothing for a single element
  if (n > 8192) {      // more two-child nodes than 64 lanes x 64 mask bits: plain sequential form
    if (lane == 0) fnv_stl::heap_pop(h, n);
    wave_sync();
    return h.get(0).key;
  }
  const int len = n - 1;
  const fnv_stl::Entry v_raw = h.get(len);  // same address in all lanes (broadcast); used in phase 3
  fnv_stl::Entry top = v_raw;
  if (KEEP_TOP) top = h.get(0);
  const int two = (len - 1) / 2;  // nodes [0, two) have two children
  // phase 1: for every two-child node, does the RIGHT child win (i.e. NOT right.key < left.key)?
  // phase 2: walk root -> leaf in 1-based numbering (node m = index + 1; children 2m, 2m+1): the
  // next node is (m << 1) | right_wins(m), so after L steps `m` spells the whole path: the node at
  // depth j is m >> (L - j).  Every lane then derives its own path entry from that one scalar.
  uint32_t m = 1;  // 1-based position of the hole
  int L = 0;
  const uint32_t two1 = (uint32_t)two;  // nodes with 1-based number <= two have two children
  if (two <= WAVE - 1) {
    // <= 63 two-child nodes (heaps of <= 128 entries): one scalar mask, indexed by 1-based number
    const bool lw = h.leftChildWins(min(max(lane - 1, 0), max(two - 1, 0)));  // always a legal pair
    const unsigned long long rw = __ballot(lane >= 1 && lane <= two && !lw);
    ph.mark(phbase);
    while (m <= two1) m = walk_step(m, rw, m);
  } else if (two <= 4 * WAVE) {
    // <= 256 two-child nodes (heaps of <= 514 entries): four scalar masks, indexed by 0-based number
    // four independent 16-byte reads per lane, issued together (indices clamped to a legal pair)
    const bool w0 = h.leftChildWins(min(lane, two - 1)), w1 = h.leftChildWins(min(WAVE + lane, two - 1));
    const bool w2 = h.leftChildWins(min(2 * WAVE + lane, two - 1)), w3 = h.leftChildWins(min(3 * WAVE + lane, two - 1));
    const unsigned long long r0 = __ballot(lane < two && !w0), r1 = __ballot(WAVE + lane < two && !w1);
    const unsigned long long r2 = __ballot(2 * WAVE + lane < two && !w2), r3 = __ballot(3 * WAVE + lane < two && !w3);
    ph.mark(phbase);
    const unsigned long long r0s = r0 << 1;  // node of 1-based number m at bit m (numbers 1..63)
    while (m <= 63u) m = walk_step(m, r0s, m);  // levels 0-5: all of these nodes have two children here (two > 63)
    while (m <= two1) {
      const uint32_t i0 = m - 1, w = i0 >> 6;
      const unsigned long long rw = w == 0 ? r0 : w == 1 ? r1 : w == 2 ? r2 : r3;
      m = walk_step(m, rw, i0);  // the bit test uses the low 6 bits of i0
    }
  } else {
    int mlo = 0, mhi = 0;  // lane r keeps the mask of nodes [64r, 64r+64)
    for (int r = 0; r * WAVE < two; r++) {
      const int node = r * WAVE + lane;
      const unsigned long long rw = __ballot(node < two && !h.leftChildWins(node));
      if (lane == r) {
        mlo = (int)(uint32_t)rw;
        mhi = (int)(uint32_t)(rw >> 32);
      }
    }
    ph.mark(phbase);
    while (m <= two1) {
      const uint32_t i0 = m - 1, w = i0 >> 6;
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(mlo, (int)w);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane(mhi, (int)w);
      const unsigned long long rw = ((unsigned long long)hi << 32) | lo;
      m = (m << 1) | (uint32_t)((rw >> (i0 & 63)) & 1ull);
    }
  }
  if ((len & 1) == 0 && m - 1 == two1) m = m << 1;  // the one node with a single (left) child, stl_heap.h:235-241
  L = 31 - __clz((int)m);  // m spells the path behind a leading 1: its length is the depth reached
  // lane j (j <= L) owns the path node at depth j
  const int sh = L - lane;
  const int my_p = sh >= 0 ? (int)(m >> sh) - 1 : 0;
  const int my_next = sh >= 1 ? (int)(m >> (sh - 1)) - 1 : 0;
  ph.mark(phbase + 1);
  // phase 3: values on the path, sift-up length, surviving moves
  fnv_stl::Entry v;
  v.key = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_raw.key)));
  v.val = (uint32_t)__builtin_amdgcn_readfirstlane((int)v_raw.val);
  fnv_stl::Entry val = v;
  bool back = false;
  if (lane < L) {
    val = h.get(my_next);
    back = val.key < v.key;  // this moved-up element would be pushed back down by the sift-up
  }
  const unsigned long long fail = ~__ballot(back) & ((1ull << L) - 1ull);  // L <= 31
  const int jf = fail ? 63 - __clzll((long long)fail) : -1;  // deepest level whose move survives
  const fnv_stl::Entry put = lane <= jf ? val : v;
  h.set_if(lane <= jf + 1, my_p, put);
  if (KEEP_TOP && lane == 0) h.set(len, top);  // std::pop_heap parks the old top in the vacated slot
  wave_sync();
  ph.mark(phbase + 2);
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(put.key)));  // lane 0 writes the root
}

}  // namespace fnv_dev
