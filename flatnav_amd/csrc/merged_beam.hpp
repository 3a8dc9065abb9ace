// merged_beam.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip / kernel_inst.hip).
//
// beam_search_merged_kernel -- the default search kernel: the same traversal as beam_search_kernel (the libstdc++-exact
// two-heap kernel) with the beam held as ONE SORTED ARRAY of at most B entries -- closest first, an "expanded" flag
// per entry (bit 31 of the id word) -- instead of the reference's two binary heaps (neighbors: B+1 entries,
// candidates: every admitted node, 2B+192 slots in the heap kernel), and ONE MERGE PER LINK ROW instead of one
// insertion per admitted neighbour.  Template parameter R:
//   R = 4, 2, 1  beams of up to R * 64 entries RESIDENT IN REGISTERS: entry e lives in lane e % 64 of register pair
//              (kr, ir)[e / 64]; entries beyond the beam's size hold {+inf, EMPTY_ID} (whose bit 31 is set, so they
//              never look unexpanded).  The merge's permutation goes through the LDS array (one scatter, one read
//              back of the chunks that moved).
//   R = 0      any beam width: the array stays in LDS (8 bytes per entry -- where the heap kernel needs (3B+194)*8,
//              which is what keeps 11-16 queries resident per CU at beam widths of 400-1200; the heap kernel: 3-5)
//              and is merged IN PLACE, 64-entry chunks from the top down, until a full chunk none of whose
//              entries moves.
//
// Why this is the same search.  The reference (Index.h:606-707) keeps `neighbors` (max-heap, <= B entries) and
// `candidates` (every admitted node, min-first).  A candidate that has been evicted from `neighbors` has a key
// >= max_dist and max_dist never grows once the beam is full, so when such a candidate reaches the top of
// `candidates` the stop test (Index.h:630) fires; it is never expanded.  The nodes that do get
// expanded are therefore exactly the not-yet-expanded members of `neighbors`, closest first -- which is what
// "first entry whose expanded flag is clear" picks here.
//
// The merge is the reference's admission loop (Index.h:693-704).  Taken one by one in link order, a neighbour is
// admitted iff the beam is not full or d < max_dist, and a full beam then drops its farthest member.  An element
// among the B smallest of beam U row is never dropped (when it is the farthest of a full beam and something closer
// arrives, B elements are closer than it) and never refused; an element outside is refused or dropped by the end of
// the row -- so the beam after the row is the B smallest of the union, whatever the order.  The row's distances are
// staged in LDS, lane j takes the j-th evaluated neighbour (link order), and every element computes its position in
// the stable merge (beam entries before candidates of equal key, candidates in link order) from wave ballots:
//         beam entry e      -> e + #{candidates with a smaller key}
//         candidate j       -> #{beam keys <= d_j} + #{candidates before j in (key, link order)}
//
// The ARRANGEMENT of the reference's heaps (libstdc++'s element moves) only decides something when equal keys meet at
// a decision:
//   (a) eviction: equal keys where the cut falls -- an element left outside the new beam has the key of the new
//       farthest member.  Which one the reference keeps is the library's choice, and the one it drops stays expandable
//       while max_dist equals its key.  Neither matters unless the search gets that far: `amb` remembers the key, the
//       query is handed over only if a node with a key >= it is about to be expanded or the search ends before
//       max_dist has dropped below it.  (A refused candidate with d == max_dist is flagged as well, which the reference
//       decides by its strict '<' -- conservative, never wrong.  Equal keys ABOVE the cut change nothing that lasts:
//       both are gone by the end of the row and nothing is expanded in between.)
//   (b) selection: the two closest unexpanded members have equal keys k (which one is expanded first).  Harmless if
//       every evaluated node with a key <= k is still in the beam when the search moves past k (max_dist > k, or the
//       beam is not full): then, whichever order the reference takes, each node with a key <= k has fewer than B
//       better nodes at its turn, so all of them get expanded, the same links get evaluated, and beam, visited set
//       and expanded set are the same once the last of them is done -- PROVIDED no equal keys meet where the beam is
//       cut while the tied nodes are being expanded (of two evaluated neighbours with the key of the farthest member,
//       the one whose row comes first is kept: `pend_cut`).  The check is therefore deferred (`pend`);
//   (d) result: equal keys among the first K results or across the K-th boundary (std::sort's order).
// Equal keys elsewhere in the beam decide nothing.  Each of the three spots is checked where it arises; a query
// that hits one, or meets a NaN / infinite distance that could be admitted, is abandoned and searched again -- by the
// same wave, right away -- with exact_query(), the libstdc++-exact two-heap search (its neighbours heap takes over the
// LDS array; its candidates heap lives in LDS when that costs no residency, else in the slot's HBM spill area).
// Otherwise results, their order and the per-query counters are identical by construction; the parity tests
// compare them bit for bit, tie-heavy inputs included.  History (profiles/r2_sorted_beam.md): a separate replay launch
// was tried first (one whole query latency at almost no parallelism, 0.5-0.7 ms on a 1.3 ms launch); the first
// in-kernel versions inserted admitted neighbours one by one (a wave shift in registers / a chunk walk in LDS per
// insertion) -- the merge costs a third of the instructions at four admissions per row.
#pragma once
#include "kernels.hpp"
namespace fnv_dev {

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// v with lane `l` (wave-uniform) replaced by the wave-uniform value s: one v_writelane_b32.  gfx950 reads at most one
// scalar register per vector instruction, so the lane select travels in M0 (which nothing else in these kernels uses).
__device__ __forceinline__ int writelane_i(int v, int s, int l) {
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(s), "s"(l) : "m0");
  return v;
}

constexpr uint32_t EXPANDED_BIT = 0x80000000u;  // beam entries: id in bits 0-30 (the host checks capacity < 2^31)
constexpr int NO_ENTRY = 1 << 30;               // LDS form: "no unexpanded entry" (compares >= every beam size)

#ifndef FNV_SORTED_WAVES_PER_SIMD
#define FNV_SORTED_WAVES_PER_SIMD 4
#endif

// order-preserving map float -> uint32 (no NaNs, -0 canonicalised by the caller)
__device__ __forceinline__ uint32_t float_ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}

// lane l of the (wave-uniform) r-th register of a small array, as a scalar.  One v_readlane per register and scalar
// selects: a select over the registers themselves makes the compiler keep the array in scratch memory.
template <int R>
__device__ __forceinline__ int lane_of(const uint32_t (&a)[R], int r, int l) {
  int s = __builtin_amdgcn_readlane((int)a[0], l);
#pragma unroll
  for (int k = 1; k < R; k++) {
    const int t = __builtin_amdgcn_readlane((int)a[k], l);
    s = r == k ? t : s;
  }
  return s;
}
template <int R>
__device__ __forceinline__ float lane_of(const float (&a)[R], int r, int l) {
  int s = __builtin_amdgcn_readlane(__float_as_int(a[0]), l);
#pragma unroll
  for (int k = 1; k < R; k++) {
    const int t = __builtin_amdgcn_readlane(__float_as_int(a[k]), l);
    s = r == k ? t : s;
  }
  return __int_as_float(s);
}

// DIRECT (round 5): the instantiations that small launches on small indexes run -- the visited set is a bitmap of all node ids
// in LDS (visited.hpp visited_insert_direct; search_params.h vis_w == 1), the tag-table code is compiled out.
template <typename T, int METRIC, int G, int CU, bool FULL, int R, bool DIRECT = false>
__global__ __launch_bounds__(WAVE, CU == 1 ? 5 : R == MB_R ? 3 : waves_per_simd<G>(FNV_SORTED_WAVES_PER_SIMD)) void beam_search_merged_kernel(const SearchParams p) {
  // (128-byte rows: 97 registers as compiled for four waves per SIMD -- one over the budget of five, which they fit.
  //  The four-chunk form -- beams of 129-256 entries -- is compiled for THREE waves per SIMD (168 registers, round 4): its
  //  LDS footprint, a 6144-slot table next to the beam, keeps at most 9-11 queries per CU resident anyway, and at four
  //  waves it kept 4-8 registers of the merge in scratch memory)
  constexpr int PU = passes<G, CU>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane_of_kernel = threadIdx.x;
  const float INF = std::numeric_limits<float>::infinity();

  while (true) {
    // (opaque per query, like the tail's below: lane-derived constants of the prologue are computed per query, not kept
    // alive -- in scratch, in the four-chunk form -- from the kernel's entry on)
    int lane = lane_of_kernel;
    asm volatile("" : "+v"(lane));
    const int item = next_query(lane);
    if (item < 0) break;
#ifdef FNV_TIMELINE  // developer build (tools/dev/launch_timeline.py): the per-query counters carry start / end clock readings instead
    const unsigned long long tl_start = wall_clock64();
#endif
    // shadows (search_params.h): items >= shadow_base are exact searches of the LAST queries, most recent first
    const uint32_t shadow_base = cold_args()->shadow_base;
    const bool shadow = shadow_base != 0u && (uint32_t)item >= shadow_base;
    const int qi = shadow ? 2 * (int)shadow_base - 1 - item : item;
    if (shadow) {  // claim the query (0 -> SH_SHADOW): answered already, or its own wave is searching it again -> nothing to do
      uint32_t old = 0u;
      if (lane == 0) old = atomicCAS(cold_args()->done_flags + qi, SH_NONE, SH_SHADOW);
      if (rfl((int)old) != (int)SH_NONE) continue;
    }
    PH_DECL
    // Per-query constants are re-read from the kernel arguments at the top of every query (a dozen scalar loads) and
    // again by the exact re-run below: nothing but the loop itself is then live across the two code paths, so the
    // register allocation of this loop does not pay for the heaps' (inlined together without this, the loop lost
    // 11 % to scalar-register spills).
    ColdArgs ca = cold_args();
    const uint8_t* const vectors = ca->vectors;
    const uint32_t* const links = ca->links;
    const uint32_t row_bytes = ca->row_bytes;
    const int nchunks = (int)ca->nchunks;
    const int B = ca->B;  // register forms: <= R * WAVE
    const int M = (int)ca->M;
    const VisGeom vg{ca->vis_nmask, ca->vis_rshift, ca->vis_rmask, ca->vis_mult, ca->vis_w};
    uint4* qlds = reinterpret_cast<uint4*>(smem + ca->off_q);
    uint32_t* vis = reinterpret_cast<uint32_t*>(smem + ca->off_vis);
    uint32_t* stage_ids = reinterpret_cast<uint32_t*>(smem + ca->off_stage_ids);
    float* stage_d = reinterpret_cast<float*>(smem + ca->off_stage_d);  // register forms: aliases `beam` (used between merges)
    uint32_t* ovf_list = reinterpret_cast<uint32_t*>(smem + ca->off_ovf);
    // [B + 2] at 16n + 8 (word -1 = write-only bin): the merge's permutation buffer (register forms) or the beam itself
    // (LDS form), and the neighbours heap of an exact re-run
    unsigned long long* beam = reinterpret_cast<unsigned long long*>(smem + ca->off_nbr);
    Query<G, CU> q;
    stage_query<T>(q, qlds, vis, ovf_list, qi, true, lane);
    __syncthreads();
    PH_MARK(0);

    float best_d;
    uint32_t entry = entry_point<T, METRIC, G, CU, FULL>(vectors, row_bytes, nchunks, q, qi, lane, best_d);
    PH_MARK(1);
    best_d = rfl(best_d);
    entry = (uint32_t)rfl((int)entry);
    uint32_t* const bitmap = cold_args()->ovf_bitmap + (uint64_t)blockIdx.x * cold_args()->bitmap_words;
    uint32_t* const ovf_glist = cold_args()->ovf_glist + (uint64_t)blockIdx.x * cold_args()->ovf_cap;

    constexpr int RR = R > 0 ? R : 1;
    float kr[RR];    // register form (R > 0)
    uint32_t ir[RR];
#pragma unroll
    for (int r = 0; r < RR; r++) {
      kr[r] = INF;
      ir[r] = EMPTY_ID;
    }
    if (lane == 0) {
      kr[0] = best_d;
      ir[0] = entry;
      if (R == 0) beam[0] = pack(fnv_stl::Entry{best_d, entry});
    }
    int cur = 0;  // LDS form (R == 0): index of the first unexpanded entry (>= n: none)
    int n = 1;
    float max_dist = best_d;
    bool ovf = false;
    visited_insert<DIRECT>(vis, vg, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    int tie = best_d != best_d ? 4 : 0;
    if ((uint32_t)item + ca->tail_exact >= ca->nq) tie = 5;  // last round of the launch: straight to the exact search
    if (shadow) tie = 5;
    float amb = INF;    // (a) pending: key at which the reference's eviction choice is unknown
    float pend = -INF;  // (b) pending: largest key at which two unexpanded members tied
    bool pend_cut = false;  // ... and, while it is pending, equal keys met where the beam is cut (see the header)
    uint32_t n_dist = 0, n_hops = 0;
    // Round 5: the hand-over log (kernels.hpp): one header + the row's admissible neighbours per hop, in the slot's HBM log
    // area.  Off (log_cap = 0) when the launch has no log area, for rows wider than LOG_MAX_LINKS and for work items that
    // go straight to the exact search; a log that would overflow stops (the query is then searched again from scratch).
    // (nor for queries that have an exact shadow -- every query of a small launch: the shadow answers the tied ones, and a lone
    //  wave's hop is 3.5 % longer with the log's two stores: r5_run4)
    uint32_t log_n = 0;
    const bool shadowed = shadow_base != 0u && (uint32_t)qi + (ca->nq - shadow_base) >= shadow_base;
    uint32_t log_cap = (tie || M > LOG_MAX_LINKS || shadowed) ? 0u : ca->log_entries;
    unsigned long long* const tie_log = ca->tie_log + (uint64_t)blockIdx.x * ca->log_entries;
    // Round 3: the link row of the RUNNER-UP is requested one hop ahead.  Once the beam has converged the next node to
    // expand is most often the current runner-up (a new neighbour rarely lands in front of every unexpanded member), and
    // then the hop starts with its link row already in a register: one dependent memory round trip per hop instead of
    // two.  A wrong guess costs one 4*M-byte request for a row that is expanded soon after anyway (it stays in L2).
    int spec_node = -1;
    uint32_t spec_row = EMPTY_ID;
    auto load_row = [&](const int nd) -> uint32_t { return lane < M ? links[(uint64_t)(uint32_t)nd * (uint32_t)M + lane] : EMPTY_ID; };
#ifdef FNV_SPEC_STATS  // developer build (tools/dev/spec_guess_stats.py): how often the runner-up guessed one hop ahead IS the next node
    uint32_t spec_hits = 0;
    auto take_row = [&](const int nd) -> uint32_t {
      spec_hits += nd == spec_node ? 1u : 0u;
      return nd == spec_node ? spec_row : load_row(nd);
    };
#else
    auto take_row = [&](const int nd) -> uint32_t { return nd == spec_node ? spec_row : load_row(nd); };
#endif
    // (not for 1-byte elements: their hop is bound by instruction issue, not by latency -- the guess bought the uint8 index
    // nothing and cost ~20 M scalar instructions per launch, 162 -> 143 M in profiles/r3_sq_counters.json)
    auto guess_next = [&](const int nd, const int runner_up) {
#ifndef FNV_NO_SPEC_ROW
      if constexpr (sizeof(T) > 1) {
        if (runner_up != spec_node || nd == spec_node) {  // (a guess that is still the runner-up keeps its row)
          spec_node = runner_up;
          if (runner_up >= 0) spec_row = load_row(runner_up);
        }
      }
#endif
    };
    __syncthreads();

    while (!tie) {
      // ---- pick the closest unexpanded member; (b) its runner-up must not have the same key -------------------
      int node, runner_up = -1;
      float key_c;
      uint32_t row0;  // the node's link row: requested as soon as the node is known (round 3), ahead of the tie bookkeeping
      if constexpr (R > 0) {
        int r1 = -1, l1 = 0, r2 = -1, l2 = 0;  // closest unexpanded member (chunk, lane) and its runner-up
#pragma unroll
        for (int r = 0; r < R; r++) {
          unsigned long long u = __ballot((int32_t)ir[r] >= 0);  // entries beyond the beam hold EMPTY_ID: bit 31 set
          if (r1 < 0 && u != 0ull) {
            r1 = r;
            l1 = __ffsll((long long)u) - 1;
            u &= u - 1ull;
          }
          if (r1 >= 0 && r2 < 0 && u != 0ull) {
            r2 = r;
            l2 = __ffsll((long long)u) - 1;
          }
        }
        if (r1 < 0) break;  // every beam member expanded: what is left in the reference's queue is stale
        node = lane_of(ir, r1, l1);
        row0 = take_row(node);
        if (r2 >= 0) runner_up = lane_of(ir, r2, l2);
        key_c = lane_of(kr, r1, l1);
        if (r2 >= 0 && lane_of(kr, r2, l2) == key_c) pend = fmaxf(pend, key_c);
#pragma unroll
        for (int r = 0; r < R; r++)
          if (r == r1) ir[r] |= lane == l1 ? EXPANDED_BIT : 0u;
      } else {
        if (cur >= n) break;
        // window of 64 entries starting at the first unexpanded one: lane 0 = the node to expand, the first other
        // unexpanded lane = the runner-up (the window slides on in the rare case that it holds none)
        const int c0 = cur;
        fnv_stl::Entry w = unpack(beam[min(c0 + lane, n - 1)]);
        node = __builtin_amdgcn_readlane((int)w.val, 0);
        row0 = take_row(node);
        key_c = readlane_f(w.key, 0);
        int c2 = NO_ENTRY;
        for (int base = c0;;) {
          const unsigned long long un = __ballot(base + lane < n && !(w.val & EXPANDED_BIT) && base + lane > c0);
          if (un) {
            const int l2 = __ffsll((long long)un) - 1;
            c2 = base + l2;
            runner_up = __builtin_amdgcn_readlane((int)w.val, l2);
            if (readlane_f(w.key, l2) == key_c) pend = fmaxf(pend, key_c);
            break;
          }
          base += WAVE;
          if (base >= n) break;
          w = unpack(beam[min(base + lane, n - 1)]);
        }
        if (lane == 0) beam[c0] = pack(fnv_stl::Entry{key_c, (uint32_t)node | EXPANDED_BIT});
        cur = c2;
      }
      if (key_c >= amb) {  // (a) became relevant
        tie = 1;
        break;
      }
      if (key_c > pend && pend > -INF) {  // (b) the search has moved past a tied key: was that tie harmless?
        if (pend_cut || (n >= B && !(max_dist > pend))) {
          tie = 2;
          break;
        }
        pend = -INF;
      }
      n_hops++;
      PH_MARK(2);
      PH_MARK(3);

      // One 64-link chunk of the row: visited test-and-mark, gather + distances of the new ones, merge.  Returns false
      // when the query has to be handed to the exact search (tie is set).  Rows of at most 64 links -- every
      // configuration of the benchmark -- are ONE call on the straight path of the hop loop; wider rows loop below.
      auto row_chunk = [&](const int m0, const uint32_t id) -> bool {
        const bool act = m0 + lane < M;
        bool isnew;
        isnew = visited_insert<DIRECT>(vis, vg, act, id, bitmap, ovf_list, ovf_glist, ovf);
        // the guess is requested HERE: after the node's own row has been consumed (the wait for that row is a wait for
        // every request in flight, so a guess issued earlier would be waited for as well) and ahead of the gathers
        if (m0 == 0) guess_next(node, runner_up);
        const unsigned long long newmask = __ballot(isnew);
        const int nn = __popcll(newmask);
        stage_ids[isnew ? __popcll(newmask & ((1ull << lane) - 1ull)) : WAVE] = id;  // keeps link order
        wave_sync();
        PH_MARK(4);
        if (log_cap != 0u && log_n + 1u + (uint32_t)WAVE > log_cap) log_cap = 0u;  // no room for another hop: the log ends here
        if (nn == 0) {
          if (log_cap != 0u) {
            if (lane == 0) tie_log[log_n] = log_header((uint32_t)node, 0u, 0u);
            log_n++;
          }
          return true;
        }
        n_dist += nn;

        // ---- distances of the row's unvisited neighbours, staged in link order ----------------------------------
        constexpr int VPW = WAVE / G;
        const int v = lane / G;
        const bool group_leader = (lane % G) == 0;
        for (int base = 0; base < nn; base += VPW * PU) {
          uint32_t cid[PU];
          float cd[PU];
#pragma unroll
          for (int pu = 0; pu < PU; pu++) cid[pu] = stage_ids[min(base + pu * VPW + v, nn - 1)];
          const int npass = min(PU, (nn - base + VPW - 1) / VPW);
          batch_dists<T, METRIC, G, CU, FULL>(vectors, row_bytes, nchunks, q, cid, npass, cd, lane);
#pragma unroll
          for (int pu = 0; pu < PU; pu++) {
            if (pu >= npass) break;
            const int slot = base + pu * VPW + v;
            stage_d[(group_leader && slot < nn) ? slot : WAVE] = cd[pu];
          }
        }
        wave_sync();
        PH_MARK(5);

        // ---- merge (see the header) ----------------------------------------------------------------------------
        // Wave votes are taken on plain comparisons and combined as lane masks in scalar registers (a vote on a
        // compound condition costs two more vector instructions: the compiler materialises the boolean first).
        const float d = stage_d[lane] + 0.0f;  // lane j: j-th neighbour in link order (-0 -> +0 for float_ord)
        const uint32_t cand_id = stage_ids[lane];
        const bool full0 = n >= B;
        const unsigned long long nnmask = nn >= WAVE ? ~0ull : (1ull << nn) - 1ull;  // lanes that hold a neighbour
        // superset of what one-by-one admission lets in
        const unsigned long long pm = (full0 ? __ballot(d < max_dist) : ~0ull) & nnmask;
        // (b) while a selection tie is pending the order of the tied expansions is the reference's choice; a neighbour
        // refused exactly at the cut (d == max_dist) would have been kept had its row come first
        if (pend > -INF && full0 && (__ballot(d == max_dist) & nnmask) != 0ull) pend_cut = true;
        if (pm != 0ull && (__ballot(!(d < INF)) & pm) != 0ull) {  // NaN / infinite distance that could be admitted
          tie = 4;
          return false;
        }
        if (log_cap != 0u) {  // header, then the candidates that could still be admitted, in link order -- ONE store: lane 63
          // never holds a neighbour (rows of at most 63 links are logged) and writes the header
          const uint32_t cl = (uint32_t)__popcll(pm);
          const bool hdr = lane == WAVE - 1;
          const uint32_t slot = hdr ? 0u : 1u + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
          const unsigned long long rec = hdr ? log_header((uint32_t)node, (uint32_t)nn, cl) : pack(fnv_stl::Entry{d, cand_id});
          if (hdr || ((pm >> lane) & 1ull) != 0ull) tie_log[log_n + slot] = rec;
          log_n += 1u + cl;
        }
        if (pm != 0ull) {
          const bool pass = ((pm >> lane) & 1ull) != 0ull;
          const int c = __popcll(pm);
          const unsigned long long key64 = ((unsigned long long)float_ord(d) << 32) | (uint32_t)lane;
          uint32_t rank = 0;   // candidate lane: candidates that precede it in (key, link order)
          int bpos = 0;        // candidate lane: beam keys <= its key
          const int n_new = min(B, n + c);
          float out_key = INF;  // smallest key among the elements this lane leaves outside the new beam (+inf: none)
          if constexpr (R > 0) {
            uint32_t le[R];  // beam lane: candidates whose key is >= the entry's key (they go after it)
#pragma unroll
            for (int r = 0; r < R; r++) le[r] = 0u;
            for (unsigned long long mm = pm; mm != 0ull; mm &= mm - 1ull) {
              const int i = __ffsll((long long)mm) - 1;
              const float di = readlane_f(d, i);
              const unsigned long long ki =
                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key64 >> 32), i) << 32) | (uint32_t)i;
              rank += ki < key64 ? 1u : 0u;
              int cnt = 0;
#pragma unroll
              for (int r = 0; r < R; r++) {
                if (r == 0 || r * WAVE < n) {  // wave-uniform (chunk 0 always holds entries)
                  const bool b = kr[r] <= di;  // entries beyond n hold +inf
                  cnt += __popcll(__ballot(b));
                  le[r] += b ? 1u : 0u;
                }
              }
              bpos = writelane_i(bpos, cnt, i);  // lane i keeps its own count
            }
            const int fpos = bpos + (int)rank;  // candidate's position in the stable merge
            // scatter through LDS; a full chunk none of whose entries moves keeps its registers (the candidates all
            // land above it)
            bool moved[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
              moved[r] = false;
              if (r == 0 || r * WAVE < n_new) {
                const int e = r * WAVE + lane;
                const int np = e + c - (int)le[r];
                const int live = n - r * WAVE;  // entries of this chunk (wave-uniform)
                const unsigned long long vmask = live >= WAVE ? ~0ull : live <= 0 ? 0ull : (1ull << live) - 1ull;
                const bool valid = e < n;
                moved[r] = live < WAVE || (__ballot((int)le[r] != c) & vmask) != 0ull;
                if (moved[r]) {
                  const bool keep = valid && np < B;
                  beam[keep ? np : -1] = pack(fnv_stl::Entry{kr[r], ir[r]});
                  if (valid && !keep) {
                    out_key = fminf(out_key, kr[r]);
                  }
                }
              }
            }
            {
              const bool keep = pass && fpos < B;
              beam[keep ? fpos : -1] = pack(fnv_stl::Entry{d, cand_id});
              if (pass && !keep) {
                out_key = fminf(out_key, d);
              }
            }
            wave_sync();
#pragma unroll
            for (int r = 0; r < R; r++) {
              if (moved[r]) {
                const int idx = r * WAVE + lane;
                const fnv_stl::Entry e = unpack(beam[idx < n_new ? idx : -1]);
                kr[r] = idx < n_new ? e.key : INF;
                ir[r] = idx < n_new ? e.val : EMPTY_ID;
              }
            }
            n = rfl(n_new);
            max_dist = lane_of(kr, (n - 1) >> 6, (n - 1) & (WAVE - 1));  // Index.h:702
          } else {
            // LDS form: the array is merged in place, 64-entry chunks from the top down -- an entry only ever moves
            // up, by at most c slots, into space the chunks above have already left; a full chunk none of whose
            // entries moves ends the walk (everything below is closer than every candidate as well)
            for (unsigned long long mm = pm; mm != 0ull; mm &= mm - 1ull) {
              const int i = __ffsll((long long)mm) - 1;
              const unsigned long long ki =
                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key64 >> 32), i) << 32) | (uint32_t)i;
              rank += ki < key64 ? 1u : 0u;
            }
            int shift_cur = 0;  // how far the first unexpanded entry moves
            for (int r = (n - 1) >> 6; r >= 0; r--) {
              const int e = r * WAVE + lane;
              const bool valid = e < n;
              const unsigned long long raw = beam[valid ? e : n - 1];
              const float key = valid ? unpack(raw).key : INF;
              int le = 0, cntv = 0;
              for (unsigned long long mm = pm; mm != 0ull; mm &= mm - 1ull) {
                const int i = __ffsll((long long)mm) - 1;
                const bool b = key <= readlane_f(d, i);
                le += b ? 1 : 0;
                const int cnt = __popcll(__ballot(b));
                cntv = writelane_i(cntv, cnt, i);
              }
              bpos += cntv;
              const int live = n - r * WAVE;
              const unsigned long long vmask = live >= WAVE ? ~0ull : (1ull << live) - 1ull;
              if ((r + 1) * WAVE <= n && (__ballot(le != c) & vmask) == 0ull) {
                bpos += r * WAVE;  // the chunks below: every entry is <= every candidate
                break;
              }
              const int np = e + c - le;
              const bool keep = valid && np < B;
              beam[keep ? np : -1] = raw;
              if (valid && !keep) {
                out_key = fminf(out_key, key);
              }
              if ((cur >> 6) == r && cur < n) shift_cur = c - __builtin_amdgcn_readlane(le, cur & (WAVE - 1));
            }
            const int fpos = bpos + (int)rank;
            {
              const bool keep = pass && fpos < B;
              beam[keep ? fpos : -1] = pack(fnv_stl::Entry{d, cand_id});
              if (pass && !keep) {
                out_key = fminf(out_key, d);
              }
            }
            wave_sync();
            // first unexpanded entry: the old one where it went (gone if pushed out), or the closest candidate
            int cur_new = cur < n ? cur + shift_cur : NO_ENTRY;
            if (cur_new >= B) cur_new = NO_ENTRY;
            const unsigned long long firstc = __ballot(rank == 0u) & pm;  // exactly one candidate lane
            const int fmin = __builtin_amdgcn_readlane(fpos, __ffsll((long long)firstc) - 1);
            if (fmin < B) cur_new = min(cur_new, fmin);
            cur = cur_new;
            n = rfl(n_new);
            max_dist = rfl(unpack(beam[n - 1]).key);  // Index.h:702
          }
          // (a) an element left outside has the key of the new farthest member
          if (__ballot(out_key == max_dist) != 0ull) {  // (out_key is +inf in lanes that left nothing outside; max_dist is finite)
            amb = max_dist;
            if (pend > -INF) pend_cut = true;  // (b) likewise: which of the two is inside depends on the order
          }
          if (max_dist < amb) amb = INF;  // every entry with that key is gone from both versions of the beam
          wave_sync();  // the LDS buffer is rewritten by the next merge
        }
        PH_MARK(6);
        wave_sync();  // stage_ids / stage_d are rewritten by the next row chunk
        return true;
      };

      if (row_chunk(0, row0))
        for (int m0 = WAVE; m0 < M; m0 += WAVE)
          if (!row_chunk(m0, m0 + lane < M ? links[(uint64_t)(uint32_t)node * (uint32_t)M + m0 + lane] : EMPTY_ID)) break;
    }

    // The query's tail.  The lane index is made opaque once more: what the tail derives from it (result slots, the
    // re-run's lane masks) is computed here, after the hop loop, instead of being hoisted above it and spilled to scratch
    // for its whole duration (round 4: the four- and two-chunk forms kept 11-20 such registers in scratch).
    const int lane_of_hop_loop = lane;
    {
    int lane = lane_of_hop_loop;
    asm volatile("" : "+v"(lane));
    ColdArgs c = cold_args();
    const int K = c->K;
    if (!tie && amb < INF) tie = 1;  // (a) still undecided when the search ended
    if (!tie && pend > -INF && (pend_cut || (n >= B && !(max_dist > pend)))) tie = 2;  // (b) likewise
    const int cnt = n < K ? n : K;
    if (!tie && R == 0) {  // (d), LDS form
      for (int k0 = 0; k0 < cnt && !tie; k0 += WAVE) {
        const int k = k0 + lane;
        const bool t = k < cnt && k + 1 < n && unpack(beam[min(k, n - 1)]).key == unpack(beam[min(k + 1, n - 1)]).key;
        if (__ballot(t) != 0ull) tie = 3;
      }
    }
    if (!tie && R > 0) {  // (d) equal keys inside the first K results or across the K-th boundary: std::sort's order
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (r * WAVE < cnt) {
          float nxt = __shfl_down(kr[r], 1, WAVE);
          if (r + 1 < R) {
            const float head = readlane_f(kr[r + 1 < R ? r + 1 : r], 0);
            nxt = lane == WAVE - 1 ? head : nxt;
          } else {
            nxt = lane == WAVE - 1 ? INF : nxt;
          }
          const int idx = r * WAVE + lane;
          if (__ballot(idx < cnt && idx + 1 < n && nxt == kr[r]) != 0ull) tie = 3;
        }
      }
    }
    if (tie) {  // equal keys met at a decision: the exact search takes over (results, counters and clean-up are exact_query's)
      // Round 5: not from scratch -- the reference's heaps are replayed from this query's log and the exact search continues
      // where the merged-beam pass stopped (kernels.hpp).  From scratch only: no / an overflowed log, a NaN / infinite distance
      // (the row was marked visited but not logged), work items that never ran the merged-beam pass.
      const bool resume = tie < 4 && log_cap != 0u;
      if (lane == 0 && tie < 5) {
        uint32_t* rc = c->redo_count;
        atomicAdd(rc, 1u);
        atomicAdd(rc + tie, 1u);
      }
      if (ovf && !resume) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, true, lane);
      // a query in the shadowed range: whoever claims it first searches it exactly -- a shadow that is already under way
      // (then this wave moves on: the answer comes one exact-search latency after the QUERY started, not after this pass
      // ended), else this wave itself (and no shadow will start on it any more)
      if (shadow_base != 0u && !shadow && (uint32_t)qi + (c->nq - shadow_base) >= shadow_base) {
        uint32_t old = 0u;
        if (lane == 0) old = atomicCAS(c->done_flags + qi, SH_NONE, SH_OWN_RERUN);
        if (rfl((int)old) == (int)SH_SHADOW) {
          if (ovf && resume) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, true, lane);
          PH_FLUSH;
          __syncthreads();
          continue;
        }
      }
      if (!resume) reset_visited(vis, ovf_list, true, lane);
      __syncthreads();
      ColdArgs xa = cold_args();
      ExactCtx x;
      x.vectors = xa->vectors;
      x.links = xa->links;
      x.row_bytes = xa->row_bytes;
      x.nchunks = (int)xa->nchunks;
      x.B = xa->B;
      x.M = (int)xa->M;
      x.cand_slots = (int)xa->cand_slots;
      x.tagged = true;
      x.vg = VisGeom{xa->vis_nmask, xa->vis_rshift, xa->vis_rmask, xa->vis_mult, xa->vis_w};
      x.nbr = reinterpret_cast<unsigned long long*>(smem + xa->off_nbr);
      x.cand = reinterpret_cast<unsigned long long*>(smem + xa->off_cand);
      x.vis = reinterpret_cast<uint32_t*>(smem + xa->off_vis);
      x.stage_ids = reinterpret_cast<uint32_t*>(smem + xa->off_stage_ids);
      x.ovf_list = reinterpret_cast<uint32_t*>(smem + xa->off_ovf);
      ExactResume rs{ExactState{1, 1, 0.f, ST_OK}, 0u, 0u, false};
      if (resume) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the log's stores have left this wave (the reader loads past the L1)
        const uint32_t taken = replay_log<DIRECT>(x, xa->tie_log + (uint64_t)blockIdx.x * xa->log_entries, log_n, entry, best_d, ovf, lane, ph, rs);
        if (lane == 0) {  // [5] queries resumed from their log, [6] hops taken from the logs, [7] hops the merged-beam passes had made
          uint32_t* rc = xa->redo_count;
          atomicAdd(rc + 5, 1u);
          atomicAdd(rc + 6, taken);
          atomicAdd(rc + 7, n_hops);
        }
      }
      exact_query<T, METRIC, G, CU, FULL, DIRECT>(x, q, qi, entry, best_d, lane, ph, shadow ? xa->done_flags + qi : nullptr, resume, rs);
#ifdef FNV_TIMELINE  // bits 60-61 of the end reading: 1 = searched twice (equal keys), 2 = sent straight to the exact search
      if (lane == 0 && !shadow && xa->out_ndist && xa->out_nhops) {
        xa->out_ndist[qi] = tl_start;
        xa->out_nhops[qi] = wall_clock64() | ((unsigned long long)(tie == 5 ? 2 : 1) << 60);
      }
#endif
      PH_FLUSH;
      continue;
    } else {
      const int32_t* labels = c->labels;  // null: construction wants node ids
      float* od_base = c->out_dist + (uint64_t)qi * K;
      int32_t* ol_base = c->out_labels + (uint64_t)qi * K;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int k = r * WAVE + lane;
        if (k < K) {
          const bool have = k < cnt;
          const uint32_t oi = ir[r] & ~EXPANDED_BIT;
          od_base[k] = have ? kr[r] : INF;
          ol_base[k] = have ? (labels ? labels[oi] : (int32_t)oi) : -1;
        }
      }
      if (R == 0) {
        for (int k = lane; k < K; k += WAVE) {
          const bool have = k < cnt;
          const fnv_stl::Entry e = unpack(beam[min(k, n - 1)]);
          const uint32_t oi = e.val & ~EXPANDED_BIT;
          od_base[k] = have ? e.key : INF;
          ol_base[k] = have ? (labels ? labels[oi] : (int32_t)oi) : -1;
        }
      }
      if (lane == 0) {
        if (c->out_count) c->out_count[qi] = cnt;
#ifdef FNV_SPEC_STATS
        atomicAdd(c->redo_count + 8, spec_hits);
        atomicAdd(c->redo_count + 9, n_hops);
#endif
#ifdef FNV_TIMELINE
        if (c->out_ndist) c->out_ndist[qi] = tl_start;
        if (c->out_nhops) c->out_nhops[qi] = wall_clock64();
#else
        if (c->out_ndist) c->out_ndist[qi] = n_dist;
        if (c->out_nhops) c->out_nhops[qi] = n_hops;
#endif
        // answered: a shadow stops / never starts (it writes the same bytes if it gets there first, so no ordering is needed)
        if (shadow_base != 0u) __hip_atomic_store(c->done_flags + qi, SH_ANSWERED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (ovf) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, true, lane);
    }
    PH_MARK(7);
    PH_FLUSH;
    __syncthreads();
  }
}

}  // namespace fnv_dev
