// beam_search.hip -- MI355X (gfx950 / CDNA4) implementation of flatnav's batched k-NN search
// and the C ABI declared in include/flatnav_hip.h.
//
// Reference path being replaced (paths relative to the reference repo):
//   Index::search              include/flatnav/index/Index.h:387-409
//   Index::initializeSearch    include/flatnav/index/Index.h:845-870
//   Index::beamSearch          include/flatnav/index/Index.h:606-659
//   Index::processCandidateNode include/flatnav/index/Index.h:661-707
//   distance dispatch          include/flatnav/distances/{L2,IP}DistanceDispatcher.h
//   VisitedSet                 include/flatnav/util/VisitedSetPool.h:16-50
//   batched loop               python-bindings/src/flatnav/bindings.cpp:161-228
//
// Execution model: one 64-lane wavefront = one query at a time; a
// persistent grid of query slots (as many as LDS lets stay resident) pulls query ids
// from an atomic dispenser.  Per query, in LDS: the query vector, the two binary
// heaps of the reference (moved with libstdc++'s exact algorithm, flatnav/util/StlExact.h,
// cooperatively by the 64 lanes), an exact 16-bit-tag visited set, and a 64-entry id staging area.  Per hop the wave
// loads one link row (M ids, coalesced), tests/marks all of them in the visited set
// in parallel, gathers the unvisited neighbours' vectors with 16-byte loads (G lanes
// per vector so each lane group reads whole 128-byte lines, PU*CU loads in flight per
// lane), reduces the distances across lanes with DPP, then replays the reference's
// sequential admission rule on the results.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/flatnav_hip.h"
#include <flatnav/util/StlExact.h>

namespace {

constexpr uint32_t EMPTY_ID = 0xFFFFFFFFu;
constexpr int WAVE = 64;
// Passes (vectors per lane group) whose loads are in flight together.  3 keeps the kernel at 124-128 VGPRs = 4 waves
// per SIMD (16 per CU); 4 needs 148 VGPRs (12 per CU) and measured 2-14 % slower on every configuration tried.
#ifndef FNV_PU
#define FNV_PU 3
#endif
constexpr int PU = FNV_PU;  // vector "passes" whose loads are issued back to back before any use
#ifndef FNV_MIN_WAVES_PER_SIMD
#define FNV_MIN_WAVES_PER_SIMD 4  // __launch_bounds__ 2nd argument: register budget 512/4 = 128 per lane
#endif

enum : int { ST_OK = 0, ST_CAND_OVERFLOW = 1 };
constexpr uint32_t OVF_LIST = 30;  // ids remembered for a cheap clean-up of the HBM visited bitmap

struct SearchParams {
  const uint8_t* vectors;   // [n_nodes][row_bytes]
  const uint32_t* links;    // [n_nodes][M]
  const int32_t* labels;    // [n_nodes]
  const uint8_t* queries;   // [nq][dim] elements, dense
  float* out_dist;          // [nq][K]
  int32_t* out_labels;      // [nq][K]
  int32_t* out_count;       // [nq] or null
  uint64_t* out_ndist;      // [nq] or null
  uint64_t* out_nhops;      // [nq] or null
  uint32_t* dispenser;      // next query id
  int32_t* status;          // sticky error flag for the whole launch
  uint32_t* ovf_bitmap;     // [nslots][bitmap_words] visited-set spill (all zero between queries)
  unsigned long long* cand_spill;  // [nslots][spill_entries]
  const uint32_t* entry_node;  // [nq] from entry_scan_kernel (null: scan inside the search kernel)
  const float* entry_dist;     // [nq]
  uint32_t* entry_node_out;    // entry_scan_kernel outputs
  float* entry_dist_out;
  uint32_t scan_tile_rows, scan_tile_stride;  // entry_scan_kernel: LDS tile geometry
  unsigned long long* phase_cycles;  // [16] profiling build only (FNV_PHASE_TIMING), else null
  uint64_t n_nodes;
  uint32_t nq, M, dim, row_bytes, nchunks, q_chunks;
  int K, B;
  uint32_t n_scan, scan_step;
  uint32_t vis_slots, vis_shift, vis_limit;
  uint32_t vis_tag16;      // 1: 16-bit-tag bucketed table (below), 0: 32-bit open addressing
  uint32_t vis_bytes;      // LDS bytes of the table
  uint32_t vis_nmask, vis_rshift, vis_rmask;  // tag16: 2^nbits-1, t = nbits-k, 2^t-1
  uint32_t vis_mult;       // tag16: buckets = vis_mult * 2^k with vis_mult in {1, 3}
  uint32_t off_ovf;        // LDS: [0] count, [1..OVF_LIST] ids that went to the HBM bitmap
  uint32_t cand_slots, spill_entries, bitmap_words;
  uint32_t off_q, off_nbr, off_cand, off_vis, off_stage_ids;
};

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
#define PH_FLUSH ph.flush(p.phase_cycles, lane)

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

template <class H>
__device__ __forceinline__ void coop_push(H& h, int n, fnv_stl::Entry v, int lane, PhaseTimer& ph, int phbase) {
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
}

// KEEP_TOP: also park the removed top in the vacated slot, as std::pop_heap does (only the result tail
// needs that; the beam loop never looks at the slot again).
template <bool KEEP_TOP, class H>
__device__ __forceinline__ void coop_pop(H& h, int n, int lane, PhaseTimer& ph, int phbase) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n <= 1) return;  // std::pop_heap does nothing for a single element
  if (n > 8192) {      // more two-child nodes than 64 lanes x 64 mask bits: plain sequential form
    if (lane == 0) fnv_stl::heap_pop(h, n);
    wave_sync();
    return;
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
    while (m <= two1) {
      m = (m << 1) | (uint32_t)((rw >> m) & 1ull);
      L++;
    }
  } else if (two <= 4 * WAVE) {
    // <= 256 two-child nodes (heaps of <= 514 entries): four scalar masks, indexed by 0-based number
    // four independent 16-byte reads per lane, issued together (indices clamped to a legal pair)
    const bool w0 = h.leftChildWins(min(lane, two - 1)), w1 = h.leftChildWins(min(WAVE + lane, two - 1));
    const bool w2 = h.leftChildWins(min(2 * WAVE + lane, two - 1)), w3 = h.leftChildWins(min(3 * WAVE + lane, two - 1));
    const unsigned long long r0 = __ballot(lane < two && !w0), r1 = __ballot(WAVE + lane < two && !w1);
    const unsigned long long r2 = __ballot(2 * WAVE + lane < two && !w2), r3 = __ballot(3 * WAVE + lane < two && !w3);
    ph.mark(phbase);
    while (m <= 64u) {  // nodes 0..63 (levels 0-5, two > 63 here): first mask only
      m = (m << 1) | (uint32_t)((r0 >> (m - 1)) & 1ull);
      L++;
    }
    while (m <= two1) {
      const uint32_t i0 = m - 1, w = i0 >> 6;
      const unsigned long long rw = w == 1 ? r1 : w == 2 ? r2 : r3;
      m = (m << 1) | (uint32_t)((rw >> (i0 & 63)) & 1ull);
      L++;
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
      L++;
    }
  }
  if ((len & 1) == 0 && m - 1 == two1) {  // the one node with a single (left) child, stl_heap.h:235-241
    m = m << 1;
    L++;
  }
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
  h.set_if(lane <= jf + 1, my_p, lane <= jf ? val : v);
  if (KEEP_TOP && lane == 0) h.set(len, top);  // std::pop_heap parks the old top in the vacated slot
  wave_sync();
  ph.mark(phbase + 2);
}

// ---------------------------------------------------------------------------------------------
// Cross-lane sums over aligned groups of G lanes (DPP inside a 16-lane row, bpermute above).
// Every step adds the same two operands in both partner lanes, so all lanes of a group end with
// bit-identical sums.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
template <int G, typename A>
__device__ __forceinline__ A group_sum(A v) {
  v += dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]  (lane ^ 1)
  v += dpp_mov<0x4E>(v);                       // quad_perm [2,3,0,1]  (lane ^ 2)
  if (G >= 8) v += dpp_mov<0x141>(v);          // row_half_mirror      (i <-> 7-i)
  if (G >= 16) v += dpp_mov<0x140>(v);         // row_mirror           (i <-> 15-i)
  if (G >= 32) v += __shfl_xor(v, 16, WAVE);
  if (G >= 64) v += __shfl_xor(v, 32, WAVE);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Distance kernels on one 16-byte chunk pair.  L2 = sum (x-y)^2, IP = 1 - sum x*y
// (L2DistanceDispatcher.h:10-17, IPDistanceDispatcher.h:10-16).  Integer element types
// accumulate exactly in int32 (the reference's float/int32 accumulations agree with that while
// the sum stays below 2^24).
// ---------------------------------------------------------------------------------------------
template <typename T, int METRIC>
struct Dist;

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int METRIC>
struct Dist<float, METRIC> {
  // two partial sums per lane so subtract and multiply-add issue as packed f32 (v_pk_add_f32 / v_pk_fma_f32)
  typedef f32x2 acc_t;
  static __device__ __forceinline__ f32x2 zero() { return f32x2{0.f, 0.f}; }
  static __device__ __forceinline__ f32x2 chunk(f32x2 acc, const uint4& x, const uint4& y) {
    const f32x2 x0 = {__uint_as_float(x.x), __uint_as_float(x.y)}, x1 = {__uint_as_float(x.z), __uint_as_float(x.w)};
    const f32x2 y0 = {__uint_as_float(y.x), __uint_as_float(y.y)}, y1 = {__uint_as_float(y.z), __uint_as_float(y.w)};
    if (METRIC == FNV_METRIC_L2) {
      const f32x2 t0 = x0 - y0, t1 = x1 - y1;
      acc = __builtin_elementwise_fma(t0, t0, acc);
      acc = __builtin_elementwise_fma(t1, t1, acc);
    } else {
      acc = __builtin_elementwise_fma(x0, y0, acc);
      acc = __builtin_elementwise_fma(x1, y1, acc);
    }
    return acc;
  }
  static __device__ __forceinline__ float lane_sum(f32x2 a) { return a.x + a.y; }
  static __device__ __forceinline__ float finish(float s) { return METRIC == FNV_METRIC_L2 ? s : 1.0f - s; }
};

template <typename T, int METRIC>
struct DistInt {
  typedef int acc_t;
  static __device__ __forceinline__ int zero() { return 0; }
  static __device__ __forceinline__ int lane_sum(int a) { return a; }
  static __device__ __forceinline__ int elem(uint32_t w, int k) {
    if (sizeof(T) == 1 && T(-1) < T(0)) return (int)(int8_t)(w >> (8 * k));
    return (int)((w >> (8 * k)) & 0xffu);
  }
  static __device__ __forceinline__ int chunk(int acc, const uint4& x, const uint4& y) {
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
    const uint32_t ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int a = elem(xs[i], k), b = elem(ys[i], k);
        if (METRIC == FNV_METRIC_L2) {
          int t = a - b;
          acc += t * t;
        } else {
          acc += a * b;
        }
      }
    }
    return acc;
  }
  static __device__ __forceinline__ float finish(int s) {
    return METRIC == FNV_METRIC_L2 ? (float)s : 1.0f - (float)s;
  }
};
template <int METRIC>
struct Dist<uint8_t, METRIC> : DistInt<uint8_t, METRIC> {};
template <int METRIC>
struct Dist<int8_t, METRIC> : DistInt<int8_t, METRIC> {};

// ---------------------------------------------------------------------------------------------
// Distances from the query (in LDS, zero padded to q_chunks) to one BATCH of up to PU * (64/G) nodes.
// Lane layout: g = lane % G walks the 16-byte chunks of a row (chunk g, g+G, ...), v = lane / G picks the
// vector of a pass; pass pu holds batch slot pu*(64/G) + v.  id[pu] is per lane (equal within a G-lane group) and
// must be a legal row for EVERY lane of passes < npass: callers give lanes beyond the last real slot the id of
// the last real one, whose loads coalesce with the real ones (no extra traffic, no EXEC juggling); their
// results are simply ignored.  `npass` (wave-uniform) = number of passes that hold at least one vector.  All
// PU*CU loads of an inner iteration are issued before the first use.  Results stay in registers: every
// lane of a group ends with the group's distance in out[pu].
// ---------------------------------------------------------------------------------------------
// Rows are addressed as rows + id * row_stride: the HBM vector table in the search kernel, an LDS tile in the
// entry-scan kernel (same arithmetic and summation order in both, so their distances agree bit for bit).
template <typename T, int METRIC, int G, int CU, bool FULL>
__device__ __forceinline__ void batch_dists(const uint8_t* rows, uint32_t row_stride, int nchunks, const uint4* qlds,
                                            const uint32_t (&id)[PU], int npass, float (&out)[PU], int lane) {
  typedef Dist<T, METRIC> D;
  typedef typename D::acc_t acc_t;
  const int g = lane % G;
  acc_t acc[PU];
  const uint8_t* rowp[PU];
#pragma unroll
  for (int pu = 0; pu < PU; pu++) {
    rowp[pu] = rows + (uint64_t)id[pu] * row_stride;
    acc[pu] = D::zero();
  }
  if (FULL) {
    // rows are a whole number of G*CU-chunk spans (e.g. d=128 f32: 32 chunks = 8 lanes x 4): no clamping, no
    // tail select; one address per pass, the CU loads use immediate offsets
#pragma unroll
    for (int pu = 0; pu < PU; pu++) rowp[pu] += g * 16;
    for (int c0 = 0; c0 < nchunks; c0 += G * CU) {
      uint4 y[PU][CU];
#pragma unroll
      for (int pu = 0; pu < PU; pu++) {
        if (pu < npass) {  // wave-uniform: skip passes that hold no vector at all
#pragma unroll
          for (int cu = 0; cu < CU; cu++)
            y[pu][cu] = *reinterpret_cast<const uint4*>(rowp[pu] + (c0 + cu * G) * 16);
        }
      }
#pragma unroll
      for (int cu = 0; cu < CU; cu++) {
        const uint4 x = qlds[c0 + cu * G + g];
#pragma unroll
        for (int pu = 0; pu < PU; pu++)
          if (pu < npass) acc[pu] = D::chunk(acc[pu], x, y[pu][cu]);
      }
    }
  } else {
  for (int c0 = 0; c0 < nchunks; c0 += G * CU) {
    uint4 y[PU][CU];
#pragma unroll
    for (int pu = 0; pu < PU; pu++) {
      if (pu < npass) {  // wave-uniform: skip passes that hold no vector at all
#pragma unroll
        for (int cu = 0; cu < CU; cu++) {
          const int c = c0 + cu * G + g;
          const int cc = c < nchunks ? c : nchunks - 1;  // clamp: always a legal address
          y[pu][cu] = *reinterpret_cast<const uint4*>(rowp[pu] + (uint32_t)cc * 16u);
        }
      }
    }
#pragma unroll
    for (int cu = 0; cu < CU; cu++) {
      const int c = c0 + cu * G + g;
      const uint4 x = qlds[c];  // zero beyond the row (q_chunks covers the last c0 block)
      const bool in_row = c < nchunks;
#pragma unroll
      for (int pu = 0; pu < PU; pu++) {
        if (pu < npass) {
          uint4 yy = y[pu][cu];
          if (!in_row) yy = x;  // x is zero there: (0-0)^2 = 0 and 0*0 = 0
          acc[pu] = D::chunk(acc[pu], x, yy);
        }
      }
    }
  }
  }
#pragma unroll
  for (int pu = 0; pu < PU; pu++) {
    out[pu] = 0.f;
    if (pu < npass) out[pu] = D::finish(group_sum<G>(D::lane_sum(acc[pu])));
  }
}

// ---------------------------------------------------------------------------------------------
// Exact visited set: open addressing (linear probing) over uint32 ids in LDS, filled to at most
// 3/4; once it would exceed that, the remaining insertions of the query go to a per-slot HBM
// bitmap (one bit per node) that the slot clears again before its next query.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool visited_insert_lds(uint32_t* tab, uint32_t slots_mask, uint32_t shift, uint32_t id) {
  uint32_t h = (id * 0x9E3779B1u) >> shift;
  while (true) {
    uint32_t cur = tab[h];
    if (cur == id) return false;
    if (cur == EMPTY_ID) {
      uint32_t old = atomicCAS(&tab[h], EMPTY_ID, id);
      if (old == EMPTY_ID) return true;
      if (old == id) return false;
    }
    h = (h + 1) & slots_mask;
  }
}
__device__ __forceinline__ bool visited_lookup_lds(const uint32_t* tab, uint32_t slots_mask, uint32_t shift,
                                                   uint32_t id) {
  uint32_t h = (id * 0x9E3779B1u) >> shift;
  while (true) {
    uint32_t cur = tab[h];
    if (cur == id) return true;
    if (cur == EMPTY_ID) return false;
    h = (h + 1) & slots_mask;
  }
}

// Exact visited set in 16 bits per element (used whenever the id width allows it).
// ids < 2^nbits.  Two multiplicative hashes h_k(id) = (id * A_k) mod 2^nbits, A_k odd, are bijections
// on nbits-bit integers, so (bucket = top bits of h_k, rem = remaining low bits, k) identifies the
// id uniquely: the table stores only tag = ((rem << 1) | k) + 1 (0 = empty) -- no false positives.
// A bucket is four 16-bit tags (8 bytes); an id may sit in either of its two buckets (inserted into
// the emptier one).  If both buckets are full the id is recorded in the slot's HBM bitmap instead;
// buckets never lose entries, so "both full -> ask the bitmap" stays consistent for the whole query.
__device__ __forceinline__ bool has_tag(uint32_t w, uint32_t tag) {
  return (w & 0xFFFFu) == tag || (w >> 16) == tag;
}
__device__ __forceinline__ int zero_halves(uint32_t w) { return ((w & 0xFFFFu) == 0u) + ((w >> 16) == 0u); }

// Bucket geometry: buckets = mult * 2^k (mult 1 or 3, so tables of 2^j or 3*2^j slots exist), t = nbits - k.
// x = h * mult; bucket = x >> t; the low t bits of x, divided by mult, number the ids inside the bucket.
__device__ __forceinline__ void tag16_slot(const SearchParams& p, uint32_t h, uint32_t which, uint32_t& bucket,
                                           uint32_t& tag) {
  const uint32_t x = h * p.vis_mult;
  bucket = x >> p.vis_rshift;
  uint32_t rem = x & p.vis_rmask;
  if (p.vis_mult == 3) rem = (rem * 43691u) >> 17;  // rem / 3, exact below 2^16
  tag = (rem << 1) + 1u + which;
}

// Called by ALL lanes (inactive ones pass act = false).  The probe is straight-line arithmetic (bitwise, no
// short-circuit branches) inside a wave-uniform retry loop that normally runs once, so EXEC is only touched
// around the CAS itself -- the scalar unit that manipulates EXEC is shared by every wave of the CU.
__device__ __forceinline__ bool visited_insert_tag16(uint32_t* tab, const SearchParams& p, bool act, uint32_t id,
                                                     uint32_t* bitmap, uint32_t* ovf_list, bool& used_bitmap) {
  uint32_t b1, b2, t1, t2;
  tag16_slot(p, (id * 0x9E3779B1u) & p.vis_nmask, 0u, b1, t1);
  tag16_slot(p, (id * 0x85EBCA6Bu) & p.vis_nmask, 1u, b2, t2);
  const uint32_t t1x = t1 | (t1 << 16), t2x = t2 | (t2 << 16);  // the tag in both halves of a word
  uint32_t pending = act ? 1u : 0u, isnew = 0u;
  while (__ballot(pending != 0u) != 0ull) {
    const uint2 B1 = *reinterpret_cast<const uint2*>(tab + 2 * b1);
    const uint2 B2 = *reinterpret_cast<const uint2*>(tab + 2 * b2);
    // zero16(w): bit 15 / 31 set iff the low / high half of w is zero (exact "has-zero-halfword" test)
#define FNV_ZERO16(w) ((~(((w) & 0x7FFF7FFFu) + 0x7FFF7FFFu) & ~(w)) & 0x80008000u)
    const uint32_t hit = FNV_ZERO16(B1.x ^ t1x) | FNV_ZERO16(B1.y ^ t1x) | FNV_ZERO16(B2.x ^ t2x) | FNV_ZERO16(B2.y ^ t2x);
    const uint32_t z1x = FNV_ZERO16(B1.x), z1y = FNV_ZERO16(B1.y), z2x = FNV_ZERO16(B2.x), z2y = FNV_ZERO16(B2.y);
#undef FNV_ZERO16
    const int e1 = __popc(z1x) + __popc(z1y), e2 = __popc(z2x) + __popc(z2y);
    const uint32_t found = hit != 0u ? 1u : 0u;
    const uint32_t full = (e1 | e2) == 0 ? 1u : 0u;
    const bool first = e1 >= e2;  // insert into the emptier bucket
    const uint32_t zx = first ? z1x : z2x;
    const uint32_t Bx = first ? B1.x : B2.x, By = first ? B1.y : B2.y;
    const uint32_t tag = first ? t1 : t2;
    const bool in_x = zx != 0u;
    const uint32_t oldw = in_x ? Bx : By;
    const uint32_t neww = oldw | ((oldw & 0xFFFFu) == 0u ? tag : tag << 16);
    const uint32_t try_cas = pending & (found ^ 1u) & (full ^ 1u);
    uint32_t got = ~oldw;
    if (try_cas) got = atomicCAS(tab + 2 * (first ? b1 : b2) + (in_x ? 0 : 1), oldw, neww);
    const uint32_t won = try_cas & (got == oldw ? 1u : 0u);
    isnew |= won;
    const uint32_t to_bitmap = pending & (found ^ 1u) & full;  // both buckets full: the HBM bitmap decides (rare)
    if (__ballot(to_bitmap != 0u) != 0ull) {
      if (to_bitmap) {
        const uint32_t bit = 1u << (id & 31);
        const uint32_t old = atomicOr(&bitmap[id >> 5], bit);
        used_bitmap = true;
        if (!(old & bit)) {
          const uint32_t pos = atomicAdd(&ovf_list[0], 1u);
          if (pos < OVF_LIST) ovf_list[1 + pos] = id;
          isnew = 1u;
        }
      }
    }
    pending = try_cas & (won ^ 1u);  // lost a race for that word: look again
  }
  return isnew != 0u;
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rfl(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// ---------------------------------------------------------------------------------------------
// The search kernel.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Entry-point selection (Index.h:845-870): argmin over nodes 0, s, 2s, ... ; strict '<', so the FIRST minimum
// wins.  Per lane the node index only grows, so '<' keeps the earliest; across lanes the tie goes to the
// smaller index.  `rows`/`stride` address row j of the scan set (HBM: j*step-th vector; LDS tile: j-th row).
// ---------------------------------------------------------------------------------------------
template <typename T, int METRIC, int G, int CU, bool FULL>
__device__ __forceinline__ void scan_rows(const uint8_t* rows, uint32_t stride_rows, uint32_t id_mul, int nchunks,
                                          const uint4* qlds, uint32_t count, uint32_t j_base, int lane, float& best_d,
                                          uint32_t& best_j) {
  constexpr int VPW = WAVE / G;
  const int v = lane / G;
  for (uint32_t j0 = 0; j0 < count; j0 += VPW * PU) {
    uint32_t sid[PU];
    bool sval[PU];
    float sd[PU];
#pragma unroll
    for (int pu = 0; pu < PU; pu++) {
      const uint32_t j = j0 + pu * VPW + v;
      sval[pu] = j < count;
      sid[pu] = min(j, count - 1) * id_mul;
    }
    const int npass = (int)min((uint32_t)PU, (count - j0 + VPW - 1) / VPW);
    batch_dists<T, METRIC, G, CU, FULL>(rows, stride_rows, nchunks, qlds, sid, npass, sd, lane);
#pragma unroll
    for (int pu = 0; pu < PU; pu++) {
      if (sval[pu] && sd[pu] < best_d) {  // strict '<': first minimum wins (Index.h:864)
        best_d = sd[pu];
        best_j = j_base + j0 + pu * VPW + v;
      }
    }
  }
}

__device__ __forceinline__ void wave_argmin(float& best_d, uint32_t& best_j) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float od = __shfl_xor(best_d, o, WAVE);
    const uint32_t oj = __shfl_xor(best_j, o, WAVE);
    if (od < best_d || (od == best_d && oj < best_j)) {
      best_d = od;
      best_j = oj;
    }
  }
}

// In-kernel variant (used when the batch kernel is switched off): scan straight from HBM / L2.
template <typename T, int METRIC, int G, int CU, bool FULL>
__device__ __forceinline__ uint32_t scan_entry_points(const SearchParams& p, const uint4* qlds, int lane, float& best_d) {
  best_d = std::numeric_limits<float>::max();
  uint32_t best_j = 0;
  scan_rows<T, METRIC, G, CU, FULL>(p.vectors, p.row_bytes, p.scan_step, (int)p.nchunks, qlds, p.n_scan, 0u, lane,
                                    best_d, best_j);
  wave_argmin(best_d, best_j);
  return best_j * p.scan_step;
}

// ---------------------------------------------------------------------------------------------
// K0: entry points for the whole batch.  Every query scans the SAME ceil(N/step) nodes, so a workgroup
// (4 waves) stages them once in LDS (tiles of scan_tile_rows rows, row stride padded by 16 bytes against
// bank conflicts) and runs SCAN_QPB queries against the tile; distances use the very same batch_dists code
// as the search kernel, so entry_dist equals what the search kernel would have computed, bit for bit.
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_WAVES = 4;
constexpr int SCAN_QPB = 32;

template <typename T, int METRIC, int G, int CU, bool FULL>
__global__ __launch_bounds__(SCAN_WAVES* WAVE) void entry_scan_kernel(const SearchParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
  const uint32_t qbytes = p.q_chunks * 16u;
  uint4* qlds = reinterpret_cast<uint4*>(smem + wave * qbytes);
  float* bd = reinterpret_cast<float*>(smem + SCAN_WAVES * qbytes);
  uint32_t* bj = reinterpret_cast<uint32_t*>(bd + SCAN_QPB);
  uint8_t* tile = reinterpret_cast<uint8_t*>(bj + SCAN_QPB);
  const uint32_t q0 = blockIdx.x * SCAN_QPB;
  const uint32_t nqb = min((uint32_t)SCAN_QPB, p.nq - q0);
  if (threadIdx.x < SCAN_QPB) {
    bd[threadIdx.x] = std::numeric_limits<float>::max();
    bj[threadIdx.x] = 0u;
  }
  for (uint32_t t0 = 0; t0 < p.n_scan; t0 += p.scan_tile_rows) {
    const uint32_t rows = min(p.scan_tile_rows, p.n_scan - t0);
    __syncthreads();  // everyone is done with the previous tile
    for (uint32_t c = threadIdx.x; c < rows * p.nchunks; c += SCAN_WAVES * WAVE) {
      const uint32_t r = c / p.nchunks, k = c % p.nchunks;
      *reinterpret_cast<uint4*>(tile + r * p.scan_tile_stride + k * 16u) =
          *reinterpret_cast<const uint4*>(p.vectors + (uint64_t)(t0 + r) * p.scan_step * p.row_bytes + k * 16u);
    }
    __syncthreads();
    for (uint32_t qq = wave; qq < nqb; qq += SCAN_WAVES) {
      const T* qsrc = reinterpret_cast<const T*>(p.queries) + (uint64_t)(q0 + qq) * p.dim;
      T* qdst = reinterpret_cast<T*>(qlds);
      const int padded = (int)(qbytes / sizeof(T));
      for (int i = lane; i < padded; i += WAVE) qdst[i] = i < (int)p.dim ? qsrc[i] : T(0);
      wave_sync();
      float best_d = std::numeric_limits<float>::max();
      uint32_t best_j = 0;
      scan_rows<T, METRIC, G, CU, FULL>(tile, p.scan_tile_stride, 1u, (int)p.nchunks, qlds, rows, t0, lane, best_d, best_j);
      wave_argmin(best_d, best_j);
      if (lane == 0 && best_d < bd[qq]) {  // later tiles hold larger indices: strict '<' keeps the first minimum
        bd[qq] = best_d;
        bj[qq] = best_j;
      }
      wave_sync();  // qlds is rewritten for the next query
    }
  }
  __syncthreads();
  if (threadIdx.x < nqb) {
    p.entry_node_out[q0 + threadIdx.x] = bj[threadIdx.x] * p.scan_step;
    p.entry_dist_out[q0 + threadIdx.x] = bd[threadIdx.x];
  }
}

template <typename T, int METRIC, int G, int CU, bool FULL>
__global__ __launch_bounds__(WAVE, FNV_MIN_WAVES_PER_SIMD) void beam_search_kernel(const SearchParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  uint4* qlds = reinterpret_cast<uint4*>(smem + p.off_q);
  LdsHeap nbr{reinterpret_cast<unsigned long long*>(smem + p.off_nbr)};
  LdsHeap cand{reinterpret_cast<unsigned long long*>(smem + p.off_cand)};  // while everything fits in LDS
  CandHeap cand_big{reinterpret_cast<unsigned long long*>(smem + p.off_cand),
                    p.cand_spill + (uint64_t)blockIdx.x * p.spill_entries, (int)p.cand_slots};
  uint32_t* vis = reinterpret_cast<uint32_t*>(smem + p.off_vis);
  uint32_t* stage_ids = reinterpret_cast<uint32_t*>(smem + p.off_stage_ids);
  uint32_t* bitmap = p.ovf_bitmap + (uint64_t)blockIdx.x * p.bitmap_words;
  uint32_t* ovf_list = reinterpret_cast<uint32_t*>(smem + p.off_ovf);
  const uint32_t vis_mask = p.vis_slots - 1;
  const int B = p.B;
  const int K = p.K;
  const int M = (int)p.M;

  while (true) {
    int qi = 0;
    if (lane == 0) qi = (int)atomicAdd(p.dispenser, 1u);
    qi = rfl(qi);
    if ((uint32_t)qi >= p.nq) break;
    PH_DECL

    // ---- stage the query (zero padded) and reset the visited table --------------------------
    {
      const T* qsrc = reinterpret_cast<const T*>(p.queries) + (uint64_t)qi * p.dim;
      T* qdst = reinterpret_cast<T*>(qlds);
      const int padded = (int)(p.q_chunks * 16u / sizeof(T));
      for (int i = lane; i < padded; i += WAVE) qdst[i] = i < (int)p.dim ? qsrc[i] : T(0);
      uint4* v4 = reinterpret_cast<uint4*>(vis);
      const uint32_t fill = p.vis_tag16 ? 0u : EMPTY_ID;
      for (uint32_t i = lane; i < p.vis_bytes / 16; i += WAVE) v4[i] = make_uint4(fill, fill, fill, fill);
      if (lane == 0) ovf_list[0] = 0u;
    }
    __syncthreads();
    PH_MARK(0);

    // ---- entry-point selection (Index.h:845-870): argmin over nodes 0, s, 2s, ... -----------
    float best_d;
    uint32_t entry;
    if (p.entry_node) {  // K0 ran: entry point and its distance were computed for the whole batch
      best_d = rfl(p.entry_dist[qi]);
      entry = (uint32_t)rfl((int)p.entry_node[qi]);
    } else {
      entry = scan_entry_points<T, METRIC, G, CU, FULL>(p, qlds, lane, best_d);
    }
    PH_MARK(1);

    // ---- beam search (Index.h:606-707) -------------------------------------------------------
    int nbr_n = 1, cand_n = 1;
    float max_dist = best_d;  // == distance(query, entry): same arithmetic, same bits
    if (lane == 0) {
      cand.set(0, fnv_stl::Entry{-best_d, entry});
      nbr.set(0, fnv_stl::Entry{best_d, entry});
    }
    uint32_t vis_count = 1;
    bool ovf = false;       // 32-bit table: switched to the bitmap; tag16: some id went to the bitmap
    if (lane == 0) {
      if (!p.vis_tag16) visited_insert_lds(vis, vis_mask, p.vis_shift, entry);
    }
    if (p.vis_tag16) visited_insert_tag16(vis, p, lane == 0, entry, bitmap, ovf_list, ovf);
    ovf = __ballot(ovf) != 0ull;
    int err = ST_OK;
    uint32_t n_dist = 0, n_hops = 0;
    __syncthreads();

    while (true) {
      if (cand_n <= 0) break;
      const fnv_stl::Entry ctop = cand.get(0);  // same address in every lane: LDS broadcast
      const float ctop_d = -rfl(ctop.key);
      if (ctop_d > max_dist && nbr_n >= B) break;  // Index.h:630
      const int node = rfl((int)ctop.val);
      // issue the link-row load now; the cooperative pop below hides most of its HBM latency
      uint32_t row_id = EMPTY_ID;
      if (lane < M) row_id = p.links[(uint64_t)(uint32_t)node * p.M + lane];
      if (cand_n <= (int)p.cand_slots) {
        coop_pop<false>(cand, cand_n, lane, ph, 8);
      } else {  // part of the heap lives in the HBM spill area
        __threadfence_block();
        coop_pop<false>(cand_big, cand_n, lane, ph, 8);
        __threadfence_block();
      }
      cand_n--;
      n_hops++;
      PH_MARK(2);

      for (int m0 = 0; m0 < M; m0 += WAVE) {
        if (!p.vis_tag16 && !ovf && vis_count + WAVE > p.vis_limit) ovf = true;
        const bool act = m0 + lane < M;
        uint32_t id = row_id;
        if (m0 > 0) id = act ? p.links[(uint64_t)(uint32_t)node * p.M + m0 + lane] : EMPTY_ID;
        PH_MARK(3);
        bool isnew = false;
        if (p.vis_tag16) {
          isnew = visited_insert_tag16(vis, p, act, id, bitmap, ovf_list, ovf);
        } else if (act) {
          if (!ovf) {
            isnew = visited_insert_lds(vis, vis_mask, p.vis_shift, id);
          } else if (!visited_lookup_lds(vis, vis_mask, p.vis_shift, id)) {
            uint32_t bit = 1u << (id & 31);
            uint32_t old = atomicOr(&bitmap[id >> 5], bit);
            isnew = !(old & bit);
          }
        }
        ovf = __ballot(ovf) != 0ull;  // wave-uniform
        const unsigned long long newmask = __ballot(isnew);
        const int n = __popcll(newmask);
        stage_ids[isnew ? __popcll(newmask & ((1ull << lane) - 1ull)) : WAVE] = id;  // keeps link order; slot 64 = bin
        vis_count += n;
        wave_sync();
        PH_MARK(4);
        if (n == 0) continue;
        n_dist += n;

        constexpr int VPW = WAVE / G;
        const int v = lane / G;
        const bool group_leader = (lane % G) == 0;
        for (int base = 0; base < n; base += VPW * PU) {
          // ---- distances of this batch, kept in registers: slot = base + pu*VPW + v lives in lane v*G
          uint32_t cid[PU];
          bool cval[PU];
          float cd[PU];
#pragma unroll
          for (int pu = 0; pu < PU; pu++) {
            const int slot = base + pu * VPW + v;
            cval[pu] = slot < n;
            cid[pu] = stage_ids[min(slot, n - 1)];  // lanes past the end re-read the last real id
          }
          const int npass = min(PU, (n - base + VPW - 1) / VPW);
          batch_dists<T, METRIC, G, CU, FULL>(p.vectors, p.row_bytes, (int)p.nchunks, qlds, cid, npass, cd, lane);
          PH_MARK(5);

          // ---- admissions in link order (Index.h:667-705).  Superset filter first: max_dist never grows
          // once the beam is full, so whatever fails here would also fail the sequential test.
#pragma unroll
          for (int pu = 0; pu < PU; pu++) {
            if (pu >= npass) break;
            unsigned long long pm = __ballot(group_leader && cval[pu] && (nbr_n < B || cd[pu] < max_dist));
            while (pm) {
              const int i = __ffsll((long long)pm) - 1;
              pm &= pm - 1;
              const float di = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cd[pu]), i));
              const uint32_t idi = (uint32_t)__builtin_amdgcn_readlane((int)cid[pu], i);
              if (nbr_n < B || di < max_dist) {  // Index.h:693
                if (cand_n >= (int)(p.cand_slots + p.spill_entries)) {
                  err = ST_CAND_OVERFLOW;
                  pm = 0;
                  break;
                }
                if (cand_n < (int)p.cand_slots) {
                  coop_push(cand, cand_n, fnv_stl::Entry{-di, idi}, lane, ph, 11);
                } else {
                  __threadfence_block();
                  coop_push(cand_big, cand_n, fnv_stl::Entry{-di, idi}, lane, ph, 11);
                  __threadfence_block();
                }
                coop_push(nbr, nbr_n, fnv_stl::Entry{di, idi}, lane, ph, 12);
                if (nbr_n + 1 > B) coop_pop<false>(nbr, nbr_n + 1, lane, ph, 13);
                cand_n++;
                if (nbr_n < B) nbr_n++;
                max_dist = rfl(nbr.get(0).key);
              }
            }
            if (err) break;
          }
          PH_MARK(6);
          if (err) break;
        }
        wave_sync();  // stage_ids is rewritten by the next row chunk
        if (err) break;
      }
      if (err) break;
    }
    PH_MARK(2);

    // ---- results (Index.h:393-408): drain, std::sort by distance, truncate to K --------------
    __syncthreads();
    const int n = nbr_n;
    const int cnt = n < K ? n : K;
    unsigned long long* res = reinterpret_cast<unsigned long long*>(smem + p.off_cand);  // candidates are dead now
    bool tie = false;
    for (int e = lane; e < n; e += WAVE) {
      const fnv_stl::Entry me = nbr.get(e);
      int rank = 0;
      bool eq = false;
      for (int j = 0; j < n; j++) {
        const float dj = nbr.get(j).key;
        rank += (dj < me.key || (dj == me.key && j < e)) ? 1 : 0;
        eq |= (dj == me.key && j != e);
      }
      if (rank < K) {
        res[rank] = pack(me);
        tie |= eq;  // a tie that reaches into the first K positions: order is the library's
      }
    }
    const bool any_tie = __ballot(tie) != 0ull;
    __syncthreads();
    if (any_tie) {
      // Exact replay of the reference's tail: pop everything (descending), std::sort ascending.
      for (int m = n; m > 1; m--) coop_pop<true>(nbr, m, lane, ph, 7);  // leaves nbr[] ascending
      __syncthreads();
      for (int i = lane; i < n; i += WAVE) res[i] = nbr.p[n - 1 - i];  // pop order = descending
      __syncthreads();
      if (lane == 0) {
        LdsHeap r{res};
        fnv_stl::sort_by_key(r, n);
      }
      __syncthreads();
    }
    for (int k = lane; k < K; k += WAVE) {
      float od = std::numeric_limits<float>::infinity();
      int32_t ol = -1;
      if (k < cnt && !err) {
        fnv_stl::Entry e = unpack(res[k]);
        od = e.key;
        ol = p.labels[e.val];
      }
      p.out_dist[(uint64_t)qi * K + k] = od;
      p.out_labels[(uint64_t)qi * K + k] = ol;
    }
    if (lane == 0) {
      if (p.out_count) p.out_count[qi] = err ? 0 : cnt;
      if (p.out_ndist) p.out_ndist[qi] = n_dist;
      if (p.out_nhops) p.out_nhops[qi] = n_hops;
      if (err) atomicMax(p.status, err);
    }
    PH_MARK(7);
    PH_FLUSH;
    if (ovf) {  // give the spill bitmap back zeroed
      __threadfence();
      const uint32_t listed = ovf_list[0];
      if (p.vis_tag16 && listed <= OVF_LIST) {  // few ids: clear just their words
        if ((uint32_t)lane < listed) bitmap[ovf_list[1 + lane] >> 5] = 0u;
      } else {
        for (uint32_t i = lane; i < p.bitmap_words; i += WAVE) bitmap[i] = 0u;
      }
      __threadfence();
    }
    __syncthreads();
  }
}

#if defined(FNV_PHASE_TIMING) || defined(FNV_MICROBENCH)
// Developer micro-benchmark (profiling builds only): cycles per cooperative heap operation on an
// LDS heap of `size` entries, `blocks` single-wave workgroups running concurrently.
__global__ __launch_bounds__(WAVE) void heap_microbench_kernel(int size, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  LdsHeap h{reinterpret_cast<unsigned long long*>(smem + 8)};
  PhaseTimer ph;
  ph.start();
  uint32_t rng = 12345u + blockIdx.x;
  int n = 0;
  for (int i = 0; i < size; i++) {
    rng = rng * 1664525u + 1013904223u;
    coop_push(h, n, fnv_stl::Entry{(float)(rng >> 8), (uint32_t)i}, lane, ph, 15);
    n++;
  }
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    rng = rng * 1664525u + 1013904223u;
    coop_push(h, n, fnv_stl::Entry{(float)(rng >> 8), (uint32_t)it}, lane, ph, 15);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = clock64();
  for (int it = 0; it < iters; it++) {
    coop_pop<true>(h, n + 1, lane, ph, 12);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t2 = clock64();
  // plain dependent LDS round trips for reference
  int idx = lane;
  for (int it = 0; it < iters; it++) idx = (int)(h.p[idx & 63] & 63);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t3 = clock64();
  if (lane == 0 && blockIdx.x == 0) {
    out[0] = (t1 - t0) / iters;
    out[1] = (t2 - t1) / iters;
    out[2] = (t3 - t2) / iters;
    out[3] = (unsigned long long)idx;
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// U1: AoS -> SoA re-layout of a staged block of nodes.  One thread per (node, 4-byte word) when
// everything is word aligned, else per byte.  Links: ids >= n_nodes are flagged; duplicates inside
// a row are replaced by the node's own id (== already visited, see header comment).
// ---------------------------------------------------------------------------------------------
__global__ void relayout_vectors_kernel(const uint8_t* __restrict__ aos, uint64_t node_size, uint64_t data_size,
                                        uint32_t row_bytes, uint64_t first_node, uint64_t count,
                                        uint8_t* __restrict__ vectors, int word_ok) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (word_ok) {
    const uint32_t wpr = row_bytes / 4;
    const uint64_t node = tid / wpr;
    const uint32_t w = (uint32_t)(tid % wpr);
    if (node >= count) return;
    uint32_t val = 0;
    if ((uint64_t)w * 4 < data_size) val = *reinterpret_cast<const uint32_t*>(aos + node * node_size + (uint64_t)w * 4);
    *reinterpret_cast<uint32_t*>(vectors + (first_node + node) * row_bytes + (uint64_t)w * 4) = val;
  } else {
    const uint64_t node = tid / row_bytes;
    const uint32_t b = (uint32_t)(tid % row_bytes);
    if (node >= count) return;
    vectors[(first_node + node) * row_bytes + b] = b < data_size ? aos[node * node_size + b] : (uint8_t)0;
  }
}

__global__ void relayout_links_kernel(const uint8_t* __restrict__ aos, uint64_t node_size, uint64_t data_size,
                                      uint32_t M, uint64_t first_node, uint64_t count, uint64_t n_nodes,
                                      uint32_t* __restrict__ links, int32_t* __restrict__ labels, int* bad_flag) {
  const uint64_t node = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= count) return;
  const uint8_t* base = aos + node * node_size + data_size;
  const uint32_t self = (uint32_t)(first_node + node);
  uint32_t* out = links + (first_node + node) * M;
  for (uint32_t i = 0; i < M; i++) {
    uint32_t id;
    memcpy(&id, base + (uint64_t)i * 4, 4);
    if ((uint64_t)id >= n_nodes) {
      atomicExch(bad_flag, 1);
      id = self;
    }
    for (uint32_t j = 0; j < i; j++) {
      uint32_t prev;
      memcpy(&prev, base + (uint64_t)j * 4, 4);
      if (prev == id) {
        id = self;
        break;
      }
    }
    out[i] = id;
  }
  int32_t lab;
  memcpy(&lab, base + (uint64_t)M * 4, 4);
  labels[first_node + node] = lab;
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess)                                                                               \
      return fail(FNV_ERR_NO_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e));        \
  } while (0)

size_t dtype_size(int dt) { return dt == FNV_DTYPE_FLOAT32 ? 4 : (dt == FNV_DTYPE_UINT8 || dt == FNV_DTYPE_INT8) ? 1 : 0; }

typedef void (*kernel_fn)(const SearchParams);

struct KernelCfg {
  int G, CU;
};
// chunks covered per inner iteration = G*CU: 8,16,32,64,128,256 (128 B ... 4 KiB of a row)
const KernelCfg kCfgs[] = {{8, 1}, {8, 2}, {8, 4}, {16, 4}, {32, 4}, {64, 4}};
constexpr int kNumCfgs = 6;

template <typename T, int METRIC, bool FULL>
kernel_fn pick_cfg(int c) {
  switch (c) {
    case 0: return beam_search_kernel<T, METRIC, 8, 1, FULL>;
    case 1: return beam_search_kernel<T, METRIC, 8, 2, FULL>;
    case 2: return beam_search_kernel<T, METRIC, 8, 4, FULL>;
    case 3: return beam_search_kernel<T, METRIC, 16, 4, FULL>;
    case 4: return beam_search_kernel<T, METRIC, 32, 4, FULL>;
    default: return beam_search_kernel<T, METRIC, 64, 4, FULL>;
  }
}

template <typename T, int METRIC, bool FULL>
kernel_fn pick_scan_cfg(int c) {
  switch (c) {
    case 0: return entry_scan_kernel<T, METRIC, 8, 1, FULL>;
    case 1: return entry_scan_kernel<T, METRIC, 8, 2, FULL>;
    case 2: return entry_scan_kernel<T, METRIC, 8, 4, FULL>;
    case 3: return entry_scan_kernel<T, METRIC, 16, 4, FULL>;
    case 4: return entry_scan_kernel<T, METRIC, 32, 4, FULL>;
    default: return entry_scan_kernel<T, METRIC, 64, 4, FULL>;
  }
}

template <typename T>
kernel_fn pick_scan_metric(int metric, int cfg, bool full) {
  if (metric == FNV_METRIC_L2)
    return full ? pick_scan_cfg<T, FNV_METRIC_L2, true>(cfg) : pick_scan_cfg<T, FNV_METRIC_L2, false>(cfg);
  return full ? pick_scan_cfg<T, FNV_METRIC_IP, true>(cfg) : pick_scan_cfg<T, FNV_METRIC_IP, false>(cfg);
}

kernel_fn pick_scan_kernel(int dtype, int metric, int cfg, bool full) {
  if (dtype == FNV_DTYPE_FLOAT32) return pick_scan_metric<float>(metric, cfg, full);
  if (dtype == FNV_DTYPE_UINT8) return pick_scan_metric<uint8_t>(metric, cfg, full);
  return pick_scan_metric<int8_t>(metric, cfg, full);
}

template <typename T>
kernel_fn pick_metric(int metric, int cfg, bool full) {
  if (metric == FNV_METRIC_L2) return full ? pick_cfg<T, FNV_METRIC_L2, true>(cfg) : pick_cfg<T, FNV_METRIC_L2, false>(cfg);
  return full ? pick_cfg<T, FNV_METRIC_IP, true>(cfg) : pick_cfg<T, FNV_METRIC_IP, false>(cfg);
}

kernel_fn pick_kernel(int dtype, int metric, int cfg, bool full) {
  if (dtype == FNV_DTYPE_FLOAT32) return pick_metric<float>(metric, cfg, full);
  if (dtype == FNV_DTYPE_UINT8) return pick_metric<uint8_t>(metric, cfg, full);
  return pick_metric<int8_t>(metric, cfg, full);
}

}  // namespace

struct fnv_index_s {
  int device = 0;
  int dtype = FNV_DTYPE_FLOAT32, metric = FNV_METRIC_L2;
  uint32_t M = 0, dim = 0, row_bytes = 0;
  uint64_t n_nodes = 0;
  uint8_t* d_vectors = nullptr;
  uint32_t* d_links = nullptr;
  int32_t* d_labels = nullptr;
  int num_cus = 0;
  // options
  int64_t visited_factor = 27, visited_slots = 0, cand_factor = 2, cand_slots = 0, spill_entries = 16384,
          blocks_per_cu = 0, visited_wide = 0, entry_kernel = 0;
  // workspace (grown on demand)
  uint32_t* d_dispenser = nullptr;  // [0] dispenser, [1] status
  unsigned long long* d_phase = nullptr;  // profiling builds only
  void* d_entry = nullptr;  // [nq] uint32 entry nodes | [nq] float entry distances (K0 output)
  size_t entry_bytes = 0;
  uint32_t* d_bitmap = nullptr;
  size_t bitmap_bytes = 0;
  unsigned long long* d_spill = nullptr;
  size_t spill_bytes = 0;
  // staging for the host-buffer entry point
  void* d_q = nullptr;
  size_t d_q_bytes = 0;
  void* d_out = nullptr;
  size_t d_out_bytes = 0;
  hipStream_t stream = nullptr;  // owned, used by fnv_search_batch
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t last_stream = nullptr;
  bool launched = false;
  uint64_t geom[6] = {0, 0, 0, 0, 0, 0};
  std::mutex mu;
};

namespace {

int index_common_init(fnv_index_s* ix) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, ix->device));
  ix->num_cus = prop.multiProcessorCount;
  HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&ix->ev0));
  HIP_TRY(hipEventCreate(&ix->ev1));
  HIP_TRY(hipMalloc(&ix->d_dispenser, 2 * sizeof(uint32_t)));
  HIP_TRY(hipMemset(ix->d_dispenser, 0, 2 * sizeof(uint32_t)));
#ifdef FNV_PHASE_TIMING
  HIP_TRY(hipMalloc(&ix->d_phase, NPHASE * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(ix->d_phase, 0, NPHASE * sizeof(unsigned long long)));
#endif
  return FNV_OK;
}

int validate_geometry(uint32_t M, uint64_t n_nodes, int data_type, int metric, uint32_t dim) {
  if (dtype_size(data_type) == 0) return fail(FNV_ERR_RUNTIME, "Unsupported data type");
  if (metric != FNV_METRIC_L2 && metric != FNV_METRIC_IP) return fail(FNV_ERR_INVALID, "Invalid metric");
  if (M == 0 || dim == 0) return fail(FNV_ERR_INVALID, "M and dim must be positive");
  if (n_nodes == 0) return fail(FNV_ERR_INVALID, "cannot upload an empty index");
  if (n_nodes >= 0xFFFFFFFFull) return fail(FNV_ERR_INVALID, "too many nodes for 32-bit node ids");
  return FNV_OK;
}

int alloc_buffers(fnv_index_s* ix) {
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipMalloc(&ix->d_vectors, ix->n_nodes * (uint64_t)ix->row_bytes));
  HIP_TRY(hipMalloc(&ix->d_links, ix->n_nodes * (uint64_t)ix->M * 4));
  HIP_TRY(hipMalloc(&ix->d_labels, ix->n_nodes * 4));
  return index_common_init(ix);
}

uint32_t pow2_ceil(uint64_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

extern "C" {

const char* fnv_last_error(void) { return g_err.c_str(); }
const char* fnv_version(void) { return "flatnav_hip gfx950 r1"; }

int fnv_device_count(int* count) {
  if (!count) return fail(FNV_ERR_INVALID, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(FNV_ERR_NO_DEVICE, std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
  }
  *count = n;
  return FNV_OK;
}

int fnv_index_alloc(uint32_t M, uint64_t n_nodes, int data_type, int metric, uint32_t dim, int device,
                    fnv_index_t* out) {
  if (!out) return fail(FNV_ERR_INVALID, "out is null");
  int rc = validate_geometry(M, n_nodes, data_type, metric, dim);
  if (rc) return rc;
  fnv_index_s* ix = new fnv_index_s();
  ix->device = device;
  ix->dtype = data_type;
  ix->metric = metric;
  ix->M = M;
  ix->dim = dim;
  ix->n_nodes = n_nodes;
  ix->row_bytes = (uint32_t)((dim * dtype_size(data_type) + 15) / 16 * 16);
  rc = alloc_buffers(ix);
  if (rc) {
    fnv_index_free(ix);
    return rc;
  }
  *out = ix;
  return FNV_OK;
}

int fnv_index_upload(const void* aos_blob, uint64_t node_size, uint64_t data_size, uint32_t M, uint64_t n_nodes,
                     int data_type, int metric, uint32_t dim, int device, fnv_index_t* out) {
  if (!aos_blob || !out) return fail(FNV_ERR_INVALID, "null argument");
  int rc = validate_geometry(M, n_nodes, data_type, metric, dim);
  if (rc) return rc;
  if (data_size != (uint64_t)dim * dtype_size(data_type) || node_size != data_size + 4ull * M + 4)
    return fail(FNV_ERR_INVALID, "node geometry does not match [data][M links][label] (Index.h:176)");
  fnv_index_t ix = nullptr;
  rc = fnv_index_alloc(M, n_nodes, data_type, metric, dim, device, &ix);
  if (rc) return rc;

  const uint64_t chunk_nodes = std::max<uint64_t>(1, (256ull << 20) / node_size);
  uint8_t* d_stage = nullptr;
  int* d_bad = nullptr;
  auto cleanup = [&]() {
    if (d_stage) (void)hipFree(d_stage);
    if (d_bad) (void)hipFree(d_bad);
  };
#define UP_TRY(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      cleanup();                                                                                       \
      fnv_index_free(ix);                                                                              \
      return fail(FNV_ERR_NO_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e));       \
    }                                                                                                  \
  } while (0)
  UP_TRY(hipMalloc(&d_stage, std::min(chunk_nodes, n_nodes) * node_size));
  UP_TRY(hipMalloc(&d_bad, sizeof(int)));
  UP_TRY(hipMemset(d_bad, 0, sizeof(int)));
  const int word_ok = (node_size % 4 == 0 && data_size % 4 == 0) ? 1 : 0;
  for (uint64_t first = 0; first < n_nodes; first += chunk_nodes) {
    const uint64_t count = std::min(chunk_nodes, n_nodes - first);
    UP_TRY(hipMemcpy(d_stage, (const uint8_t*)aos_blob + first * node_size, count * node_size, hipMemcpyHostToDevice));
    const uint64_t units = count * (word_ok ? ix->row_bytes / 4 : ix->row_bytes);
    hipLaunchKernelGGL(relayout_vectors_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, 0, d_stage,
                       node_size, data_size, ix->row_bytes, first, count, ix->d_vectors, word_ok);
    hipLaunchKernelGGL(relayout_links_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, d_stage,
                       node_size, data_size, M, first, count, n_nodes, ix->d_links, ix->d_labels, d_bad);
    UP_TRY(hipGetLastError());
    UP_TRY(hipDeviceSynchronize());
  }
  int bad = 0;
  UP_TRY(hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost));
  cleanup();
#undef UP_TRY
  if (bad) {
    fnv_index_free(ix);
    return fail(FNV_ERR_RUNTIME, "index blob holds link ids outside [0, n_nodes)");
  }
  *out = ix;
  return FNV_OK;
}

int fnv_index_device_buffers(fnv_index_t ix, void* ptrs[3], uint64_t sizes[3]) {
  if (!ix || !ptrs || !sizes) return fail(FNV_ERR_INVALID, "null argument");
  ptrs[0] = ix->d_vectors;
  ptrs[1] = ix->d_links;
  ptrs[2] = ix->d_labels;
  sizes[0] = ix->n_nodes * (uint64_t)ix->row_bytes;
  sizes[1] = ix->n_nodes * (uint64_t)ix->M * 4;
  sizes[2] = ix->n_nodes * 4;
  return FNV_OK;
}

int fnv_index_info(fnv_index_t ix, uint64_t info[8]) {
  if (!ix || !info) return fail(FNV_ERR_INVALID, "null argument");
  info[0] = (uint64_t)ix->dtype;
  info[1] = ix->M;
  info[2] = ix->row_bytes;
  info[3] = ix->n_nodes;
  info[4] = ix->dim;
  info[5] = (uint64_t)ix->metric;
  info[6] = (uint64_t)ix->device;
  info[7] = ix->n_nodes * ((uint64_t)ix->row_bytes + 4ull * ix->M + 4) + ix->bitmap_bytes + ix->spill_bytes;
  return FNV_OK;
}

int fnv_index_free(fnv_index_t ix) {
  if (!ix) return FNV_OK;
  (void)hipSetDevice(ix->device);
  if (ix->stream) (void)hipStreamSynchronize(ix->stream);
  void* bufs[] = {ix->d_vectors, ix->d_links, ix->d_labels, ix->d_dispenser, ix->d_bitmap, ix->d_spill, ix->d_q, ix->d_out, ix->d_phase, ix->d_entry};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (ix->ev0) (void)hipEventDestroy(ix->ev0);
  if (ix->ev1) (void)hipEventDestroy(ix->ev1);
  if (ix->stream) (void)hipStreamDestroy(ix->stream);
  delete ix;
  return FNV_OK;
}

int fnv_set_option(fnv_index_t ix, const char* name, int64_t value) {
  if (!ix || !name) return fail(FNV_ERR_INVALID, "null argument");
  std::string n(name);
  if (value < 0) return fail(FNV_ERR_INVALID, "option values must be non-negative");
  if (n == "visited_factor") ix->visited_factor = std::max<int64_t>(1, value);
  else if (n == "visited_slots") {
    if (value && (value & (value - 1)) && ((value % 3) || ((value / 3) & (value / 3 - 1))))
      return fail(FNV_ERR_INVALID, "visited_slots must be 2^j or 3*2^j");
    if (value && value < 256) return fail(FNV_ERR_INVALID, "visited_slots must be at least 256");
    ix->visited_slots = value;
  } else if (n == "cand_factor") ix->cand_factor = std::max<int64_t>(1, value);
  else if (n == "cand_slots") ix->cand_slots = value;
  else if (n == "spill_entries") ix->spill_entries = std::max<int64_t>(1, value);
  else if (n == "blocks_per_cu") ix->blocks_per_cu = value;
  else if (n == "visited_wide") ix->visited_wide = value;
  else if (n == "entry_kernel") ix->entry_kernel = value;
  else return fail(FNV_ERR_INVALID, "unknown option: " + n);
  return FNV_OK;
}

int fnv_search_batch_device(fnv_index_t ix, const void* d_queries, uint64_t nq, int K, int ef_search,
                            int num_initializations, float* d_out_dist, int32_t* d_out_labels,
                            int32_t* d_out_count, uint64_t* d_out_ndist, uint64_t* d_out_nhops, void* hip_stream) {
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  // Index.h:847-849
  if (num_initializations <= 0) return fail(FNV_ERR_INVALID, "num_initializations must be greater than 0.");
  if (K <= 0 || ef_search <= 0) return fail(FNV_ERR_INVALID, "K and ef_search must be positive");
  if (nq == 0) return FNV_OK;
  if (!d_queries || !d_out_dist || !d_out_labels) return fail(FNV_ERR_INVALID, "null buffer");
  if (nq > 0x7FFFFFFFull) return fail(FNV_ERR_INVALID, "too many queries in one batch");
  std::lock_guard<std::mutex> lock(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t stream = (hipStream_t)hip_stream;

  SearchParams p;
  memset(&p, 0, sizeof(p));
  p.vectors = ix->d_vectors;
  p.links = ix->d_links;
  p.labels = ix->d_labels;
  p.queries = (const uint8_t*)d_queries;
  p.out_dist = d_out_dist;
  p.out_labels = d_out_labels;
  p.out_count = d_out_count;
  p.out_ndist = d_out_ndist;
  p.out_nhops = d_out_nhops;
  p.n_nodes = ix->n_nodes;
  p.nq = (uint32_t)nq;
  p.M = ix->M;
  p.dim = ix->dim;
  p.row_bytes = ix->row_bytes;
  p.nchunks = ix->row_bytes / 16;
  p.K = K;
  p.B = std::max(ef_search, K);  // Index.h:392
  // Index.h:851-861: step = max(1, N / n_init); nodes 0, step, 2*step, ... < N
  uint64_t step = ix->n_nodes / (uint64_t)num_initializations;
  if (step == 0) step = 1;
  p.scan_step = (uint32_t)step;
  p.n_scan = (uint32_t)((ix->n_nodes + step - 1) / step);

  int cfg = kNumCfgs - 1;
  for (int c = 0; c < kNumCfgs; c++)
    if ((uint32_t)(kCfgs[c].G * kCfgs[c].CU) >= p.nchunks) {
      cfg = c;
      break;
    }
  const uint32_t per_iter = (uint32_t)(kCfgs[cfg].G * kCfgs[cfg].CU);
  p.q_chunks = (p.nchunks + per_iter - 1) / per_iter * per_iter;

  {
    // Visited-table geometry.  Slots = 2^j or 3*2^j.  16-bit tags whenever the per-bucket id range
    // fits 14 bits: buckets = mult*2^k, t = nbits - k, need t <= 14 (mult 1) or t <= 15 (mult 3).
    uint32_t nbits = 1;
    while (nbits < 32 && (1ull << nbits) < ix->n_nodes) nbits++;
    uint64_t want = ix->visited_slots ? (uint64_t)ix->visited_slots
                                      : (uint64_t)ix->visited_factor * (uint64_t)p.B + 600;
    want = std::max<uint64_t>(want, 256);
    uint32_t slots = 256;
    for (uint32_t base = 256;; base <<= 1) {  // candidates in increasing order: 2^j, 3*2^(j-1), 2^(j+1), ...
      if (base >= want || base >= (1u << 15)) {
        slots = base;
        break;
      }
      if ((uint64_t)base / 2 * 3 >= want) {
        slots = base / 2 * 3;
        break;
      }
    }
    if (ix->visited_slots) slots = (uint32_t)ix->visited_slots;
    const uint32_t mult = (slots % 3 == 0) ? 3u : 1u;
    uint32_t k = 0;
    for (uint32_t b = slots / 4 / mult; b > 1; b >>= 1) k++;
    bool can16 = !ix->visited_wide && nbits <= 30 && k <= nbits && (nbits - k) <= (mult == 3 ? 15u : 14u);
    if (!can16 && mult == 3) {  // the 32-bit table needs a power of two
      slots = pow2_ceil(slots);
    }
    p.vis_slots = slots;
    p.vis_tag16 = can16 ? 1u : 0u;
    p.vis_mult = mult;
    p.vis_nmask = (uint32_t)((1ull << nbits) - 1ull);
    p.vis_rshift = can16 ? nbits - k : 0;
    p.vis_rmask = can16 ? ((1u << p.vis_rshift) - 1u) : 0;
    p.vis_bytes = can16 ? slots * 2 : slots * 4;
    p.vis_shift = 32;
    for (uint32_t sft = p.vis_slots; sft > 1; sft >>= 1) p.vis_shift--;
    p.vis_limit = p.vis_slots / 4 * 3;
  }
  p.cand_slots = ix->cand_slots ? (uint32_t)ix->cand_slots : (uint32_t)(ix->cand_factor * p.B + 192);
  p.cand_slots = std::max<uint32_t>(p.cand_slots, (uint32_t)p.B + 1);  // also hosts the final result list
  p.spill_entries = (uint32_t)ix->spill_entries;
  p.bitmap_words = (uint32_t)((ix->n_nodes + 31) / 32);

  auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
  uint32_t off = 0;
  p.off_q = off;
  off = align16(off + p.q_chunks * 16);
  p.off_nbr = off + 8;  // heap arrays start at 16n + 8: child pairs are 16-byte aligned
  off = align16(off + 8 + ((uint32_t)p.B + 2) * 8);
  p.off_cand = off + 8;
  off = align16(off + 8 + (p.cand_slots + 1) * 8);
  p.off_vis = off;
  off = align16(off + p.vis_bytes);
  p.off_stage_ids = off;
  off = align16(off + (WAVE + 1) * 4);  // + one write-only slot for lanes with nothing to stage
  p.off_ovf = off;
  off = align16(off + (OVF_LIST + 2) * 4);
  const uint32_t lds_bytes = off;
  if (lds_bytes > 160u * 1024u)
    return fail(FNV_ERR_INVALID, "ef_search too large for the on-chip beam state (needs " + std::to_string(lds_bytes) +
                                     " bytes of LDS, 163840 available); lower ef_search or the *_slots options");

  const bool full = (p.nchunks % per_iter) == 0;  // rows are whole spans: the lean FULL kernels apply
  kernel_fn kern = pick_kernel(ix->dtype, ix->metric, cfg, full);
  HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  int bpc = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, WAVE, lds_bytes));
  if (bpc < 1) bpc = 1;
  if (ix->blocks_per_cu > 0) bpc = std::min<int>(bpc, (int)ix->blocks_per_cu);
  const uint32_t nslots = (uint32_t)std::min<uint64_t>(nq, (uint64_t)bpc * (uint64_t)ix->num_cus);

  // workspace
  const size_t need_bitmap = (size_t)nslots * p.bitmap_words * 4;
  if (need_bitmap > ix->bitmap_bytes) {
    if (ix->d_bitmap) HIP_TRY(hipFree(ix->d_bitmap));
    ix->d_bitmap = nullptr;
    ix->bitmap_bytes = 0;
    HIP_TRY(hipMalloc(&ix->d_bitmap, need_bitmap));
    HIP_TRY(hipMemset(ix->d_bitmap, 0, need_bitmap));
    ix->bitmap_bytes = need_bitmap;
  }
  const size_t need_spill = (size_t)nslots * p.spill_entries * 8;
  if (need_spill > ix->spill_bytes) {
    if (ix->d_spill) HIP_TRY(hipFree(ix->d_spill));
    ix->d_spill = nullptr;
    ix->spill_bytes = 0;
    HIP_TRY(hipMalloc(&ix->d_spill, need_spill));
    ix->spill_bytes = need_spill;
  }
  p.ovf_bitmap = ix->d_bitmap;
  p.cand_spill = ix->d_spill;
  p.dispenser = ix->d_dispenser;
  p.status = (int32_t*)(ix->d_dispenser + 1);
  p.phase_cycles = ix->d_phase;

  HIP_TRY(hipMemsetAsync(ix->d_dispenser, 0, 2 * sizeof(uint32_t), stream));
  HIP_TRY(hipEventRecord(ix->ev0, stream));
  if (ix->entry_kernel) {
    // K0: one pass over the shared entry-scan nodes for the whole batch (LDS-staged), same stream
    const size_t need_entry = (size_t)nq * 8;
    if (need_entry > ix->entry_bytes) {
      if (ix->d_entry) HIP_TRY(hipFree(ix->d_entry));
      ix->d_entry = nullptr;
      ix->entry_bytes = 0;
      HIP_TRY(hipMalloc(&ix->d_entry, need_entry));
      ix->entry_bytes = need_entry;
    }
    p.entry_node_out = (uint32_t*)ix->d_entry;
    p.entry_dist_out = (float*)((uint8_t*)ix->d_entry + (size_t)nq * 4);
    p.scan_tile_stride = p.row_bytes + 16;
    const uint32_t fixed = SCAN_WAVES * p.q_chunks * 16 + SCAN_QPB * 8;
    p.scan_tile_rows = std::max<uint32_t>(1, std::min<uint32_t>(p.n_scan, (64u * 1024u) / p.scan_tile_stride));
    const uint32_t scan_lds = fixed + p.scan_tile_rows * p.scan_tile_stride;
    kernel_fn scan = pick_scan_kernel(ix->dtype, ix->metric, cfg, full);
    HIP_TRY(hipFuncSetAttribute((const void*)scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_lds));
    hipLaunchKernelGGL(scan, dim3((unsigned)((nq + SCAN_QPB - 1) / SCAN_QPB)), dim3(SCAN_WAVES * WAVE), scan_lds, stream, p);
    HIP_TRY(hipGetLastError());
    p.entry_node = p.entry_node_out;
    p.entry_dist = p.entry_dist_out;
  }
  hipLaunchKernelGGL(kern, dim3(nslots), dim3(WAVE), lds_bytes, stream, p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ix->ev1, stream));
  ix->last_stream = stream;
  ix->launched = true;
  ix->geom[0] = nslots;
  ix->geom[1] = WAVE;
  ix->geom[2] = lds_bytes;
  ix->geom[3] = (uint64_t)bpc;
  ix->geom[4] = p.vis_slots;
  ix->geom[5] = p.cand_slots;
  return FNV_OK;
}

int fnv_search_status(fnv_index_t ix) {
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  if (!ix->launched) return FNV_OK;
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->last_stream));
  int32_t st = 0;
  HIP_TRY(hipMemcpy(&st, ix->d_dispenser + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (st == ST_CAND_OVERFLOW)
    return fail(FNV_ERR_CAPACITY, "candidate heap overflowed its HBM spill area; raise the spill_entries option");
  return FNV_OK;
}

int fnv_search_batch(fnv_index_t ix, const void* queries, uint64_t nq, int K, int ef_search, int num_initializations,
                     float* out_dist, int32_t* out_labels, int32_t* out_count, uint64_t* out_ndist,
                     uint64_t* out_nhops) {
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  if (num_initializations <= 0) return fail(FNV_ERR_INVALID, "num_initializations must be greater than 0.");
  if (K <= 0 || ef_search <= 0) return fail(FNV_ERR_INVALID, "K and ef_search must be positive");
  if (nq == 0) return FNV_OK;
  if (!queries || !out_dist || !out_labels) return fail(FNV_ERR_INVALID, "null buffer");
  HIP_TRY(hipSetDevice(ix->device));
  const size_t qbytes = (size_t)nq * ix->dim * dtype_size(ix->dtype);
  // one output slab: dist | labels | count | ndist | nhops
  const size_t o_dist = 0;
  const size_t o_lab = o_dist + (size_t)nq * K * 4;
  const size_t o_cnt = o_lab + (size_t)nq * K * 4;
  const size_t o_nd = (o_cnt + (size_t)nq * 4 + 7) & ~(size_t)7;
  const size_t o_nh = o_nd + (size_t)nq * 8;
  const size_t obytes = o_nh + (size_t)nq * 8;
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    if (qbytes > ix->d_q_bytes) {
      if (ix->d_q) HIP_TRY(hipFree(ix->d_q));
      ix->d_q = nullptr;
      ix->d_q_bytes = 0;
      HIP_TRY(hipMalloc(&ix->d_q, qbytes));
      ix->d_q_bytes = qbytes;
    }
    if (obytes > ix->d_out_bytes) {
      if (ix->d_out) HIP_TRY(hipFree(ix->d_out));
      ix->d_out = nullptr;
      ix->d_out_bytes = 0;
      HIP_TRY(hipMalloc(&ix->d_out, obytes));
      ix->d_out_bytes = obytes;
    }
  }
  uint8_t* o = (uint8_t*)ix->d_out;
  HIP_TRY(hipMemcpyAsync(ix->d_q, queries, qbytes, hipMemcpyHostToDevice, ix->stream));
  int rc = fnv_search_batch_device(ix, ix->d_q, nq, K, ef_search, num_initializations, (float*)(o + o_dist),
                                   (int32_t*)(o + o_lab), (int32_t*)(o + o_cnt), (uint64_t*)(o + o_nd),
                                   (uint64_t*)(o + o_nh), ix->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out_dist, o + o_dist, (size_t)nq * K * 4, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipMemcpyAsync(out_labels, o + o_lab, (size_t)nq * K * 4, hipMemcpyDeviceToHost, ix->stream));
  if (out_count) HIP_TRY(hipMemcpyAsync(out_count, o + o_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, ix->stream));
  if (out_ndist) HIP_TRY(hipMemcpyAsync(out_ndist, o + o_nd, (size_t)nq * 8, hipMemcpyDeviceToHost, ix->stream));
  if (out_nhops) HIP_TRY(hipMemcpyAsync(out_nhops, o + o_nh, (size_t)nq * 8, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return fnv_search_status(ix);
}

int fnv_last_kernel_ms(fnv_index_t ix, float* ms) {
  if (!ix || !ms) return fail(FNV_ERR_INVALID, "null argument");
  if (!ix->launched) return fail(FNV_ERR_RUNTIME, "no search has been launched on this index");
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipEventSynchronize(ix->ev1));
  HIP_TRY(hipEventElapsedTime(ms, ix->ev0, ix->ev1));
  return FNV_OK;
}

#ifdef FNV_PHASE_TIMING
int fnv_debug_heap_microbench(int size, int iters, int blocks, uint64_t out[4]) {
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc(&d, 4 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(heap_microbench_kernel, dim3(blocks), dim3(WAVE), (size + 4) * 8 + 64, 0, size, iters, d);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipFree(d));
  return FNV_OK;
}
// Profiling builds only: cumulative shader cycles per kernel phase (and reset).
int fnv_debug_phase_cycles(fnv_index_t ix, uint64_t out[16]) {
  if (!ix || !out) return fail(FNV_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, ix->d_phase, NPHASE * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(ix->d_phase, 0, NPHASE * sizeof(uint64_t)));
  return FNV_OK;
}
#endif

int fnv_last_launch_geometry(fnv_index_t ix, uint64_t geom[6]) {
  if (!ix || !geom) return fail(FNV_ERR_INVALID, "null argument");
  for (int i = 0; i < 6; i++) geom[i] = ix->geom[i];
  return FNV_OK;
}

}  // extern "C"
