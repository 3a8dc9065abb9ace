// sorted_beam.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip).
//
// beam_search_sorted_kernel: the same traversal as beam_search_kernel (the libstdc++-exact two-heap kernel) with the
// beam held as ONE SORTED ARRAY of at most B entries -- closest first, an "expanded" flag per entry -- instead of the
// reference's two binary heaps (neighbors: B+1 entries, candidates: every admitted node, 2B+192 slots here).
// This kernel keeps the array in LDS, 8 bytes per entry {key | id, bit 31 = expanded}, for any beam width; an
// insertion reads/writes only the 64-entry chunks above the insertion point.  It needs B*8 bytes where the heap
// kernel needs (3B+194)*8, which is what keeps 11-16 queries resident per CU at beam widths of 400-1200 (the heap
// kernel: 3-5).  Beams of at most 256 entries are served by merged_beam.hpp (the array in registers, one merge per
// link row; same rules), which replaced this file's former register form (one wave shift per insertion).
//
// Why this is the same search.  The reference (Index.h:606-707) keeps `neighbors` (max-heap, <= B entries) and
// `candidates` (every admitted node, min-first).  A candidate that has been evicted from `neighbors` has a key
// >= max_dist and max_dist never grows once the beam is full, so when such a candidate reaches the top of
// `candidates` the stop test (Index.h:630) fires; it is never expanded.  The nodes that do get
// expanded are therefore exactly the not-yet-expanded members of `neighbors`, closest first -- which is what
// "first entry whose expanded flag is clear" picks here -- and the admission rule (strict <, link order, max_dist
// refreshed after every admission) is applied verbatim.  The ARRANGEMENT of the reference's heaps (libstdc++'s
// element moves) only decides something when equal keys meet at a decision:
//   (a) eviction: the two largest keys of a full beam are equal (which one goes; a stale candidate with
//       key == max_dist would still be expanded by the reference);
//   (b) selection: the two closest unexpanded members have equal keys k (which one is expanded first).  Harmless if
//       every evaluated node with a key <= k is still in the beam when the search moves past k (max_dist > k, or the
//       beam is not full): then, whichever order the reference takes, each node with a key <= k has fewer than B
//       better nodes at its turn, so all of them get expanded, the same links get evaluated, and beam, visited set
//       and expanded set are the same once the last of them is done.  The check is therefore deferred (`pend`);
//   (d) result: equal keys among the first K results or across the K-th boundary (std::sort's order).
// Equal keys elsewhere in the beam decide nothing.  Each of the three spots is checked where it arises; a query
// that hits one, or meets a NaN / infinite distance, is abandoned and searched again -- by the same wave, right
// away -- with exact_query(), the libstdc++-exact two-heap search (its neighbours heap takes over the beam's LDS
// array; its candidates heap lives in LDS when that costs no residency, else in the slot's HBM spill area).
// Otherwise results, their order and the per-query counters are identical by construction; the parity tests
// compare them bit for bit, tie-heavy inputs included.  (A separate replay launch was tried first: it costs one
// whole query latency at almost no parallelism, 0.5-0.7 ms on a 1.3 ms launch -- profiles/r2_sorted_beam.md.)
#pragma once
#include "kernels.hpp"
namespace fnv_dev {

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

constexpr uint32_t EXPANDED_BIT = 0x80000000u;  // beam entries: id in bits 0-30 (the host checks capacity < 2^31)
constexpr int NO_ENTRY = 1 << 30;               // "no unexpanded entry" (compares >= every beam size)

#ifndef FNV_SORTED_WAVES_PER_SIMD
#define FNV_SORTED_WAVES_PER_SIMD 4
#endif
template <typename T, int METRIC, int G, int CU, bool FULL>
__global__ __launch_bounds__(WAVE, FNV_SORTED_WAVES_PER_SIMD) void beam_search_sorted_kernel(const SearchParams p) {
  constexpr int PU = passes<G, CU>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const float INF = std::numeric_limits<float>::infinity();

  while (true) {
    const int qi = next_query(lane);
    if (qi < 0) break;
    PH_DECL
    // Per-query constants are re-read from the kernel arguments at the top of every query (a dozen scalar loads) and
    // again by the exact re-run below: nothing but the loop itself is then live across the two code paths, so the
    // register allocation of the sorted-beam loop does not pay for the heaps' (inlined together without this, the
    // loop lost 11 % to scalar-register spills).
    ColdArgs ca = cold_args();
    const uint8_t* const vectors = ca->vectors;
    const uint32_t* const links = ca->links;
    const uint32_t row_bytes = ca->row_bytes;
    const int nchunks = (int)ca->nchunks;
    const int B = ca->B;
    const int M = (int)ca->M;
    const VisGeom vg{ca->vis_nmask, ca->vis_rshift, ca->vis_rmask, ca->vis_mult, ca->vis_w};
    uint4* qlds = reinterpret_cast<uint4*>(smem + ca->off_q);
    uint32_t* vis = reinterpret_cast<uint32_t*>(smem + ca->off_vis);
    uint32_t* stage_ids = reinterpret_cast<uint32_t*>(smem + ca->off_stage_ids);
    uint32_t* ovf_list = reinterpret_cast<uint32_t*>(smem + ca->off_ovf);
    // [B + 2] at 16n + 8: the sorted beam (slot B = write-only bin), and the neighbours heap of an exact re-run
    unsigned long long* beam = reinterpret_cast<unsigned long long*>(smem + ca->off_nbr);
    stage_query<T>(qlds, vis, ovf_list, qi, true, lane);
    __syncthreads();
    PH_MARK(0);

    float best_d;
    uint32_t entry = entry_point<T, METRIC, G, CU, FULL>(vectors, row_bytes, nchunks, qlds, qi, lane, best_d);
    PH_MARK(1);
    // wave-uniform by construction; say so, or every loop below is compiled as divergent control flow
    best_d = rfl(best_d);
    entry = (uint32_t)rfl((int)entry);
    uint32_t* const bitmap = cold_args()->ovf_bitmap + (uint64_t)blockIdx.x * cold_args()->bitmap_words;
    uint32_t* const ovf_glist = cold_args()->ovf_glist + (uint64_t)blockIdx.x * cold_args()->ovf_cap;

    // ---- the beam --------------------------------------------------------------------------------------------
    int cur = 0;  // index of the first unexpanded entry (>= n: none)
    if (lane == 0) beam[0] = pack(fnv_stl::Entry{best_d, entry});
    int n = 1;
    float max_dist = best_d;
    float second = -INF;  // full beam: key of entry B-2 (the runner-up for eviction)
    bool ovf = false;
    if (vg.w == 16) visited_insert_tag16(vis, vg, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    else visited_insert_tagw(reinterpret_cast<unsigned long long*>(vis), vg, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    ovf = __ballot(ovf) != 0ull;
    int tie = best_d != best_d ? 4 : 0;  // why the query is handed to the exact kernel (0 = it is not); NaN entry: 4
    // The last queries of a launch go straight to the exact search: a query that is searched twice finishes a whole
    // exact-search latency late, and in the last round of a launch that lengthens the launch itself (one such query
    // costs as much as hundreds).  The exact search alone is slower than the sorted beam but never needs a second pass.
    if ((uint32_t)qi + ca->tail_exact >= ca->nq) tie = 5;
    float amb = INF;  // (a) pending (+inf = none): a key at which the reference's eviction choice is unknown (see below)
    float pend = -INF;  // (b) pending (-inf = none): largest key at which two unexpanded members tied
    uint32_t n_dist = 0, n_hops = 0;
    __syncthreads();

    while (!tie) {
      // ---- pick the closest unexpanded member; (b) its runner-up must not have the same key -------------------
      int node;
      float key_c;
      {
        if (cur >= n) break;
        // window of 64 entries starting at the first unexpanded one: lane 0 = the node to expand, the first other
        // unexpanded lane = the runner-up (the window slides on in the rare case that it holds none)
        const int c = cur;
        fnv_stl::Entry w = unpack(beam[min(c + lane, n - 1)]);
        node = __builtin_amdgcn_readlane((int)w.val, 0);
        key_c = readlane_f(w.key, 0);
        int c2 = NO_ENTRY;
        for (int base = c;;) {
          const unsigned long long un =
              __ballot(base + lane < n && !(w.val & EXPANDED_BIT) && base + lane > c);
          if (un) {
            const int l2 = __ffsll((long long)un) - 1;
            c2 = base + l2;
            if (readlane_f(w.key, l2) == key_c) pend = fmaxf(pend, key_c);
            break;
          }
          base += WAVE;
          if (base >= n) break;
          w = unpack(beam[min(base + lane, n - 1)]);
        }
        if (lane == 0) beam[c] = pack(fnv_stl::Entry{key_c, (uint32_t)node | EXPANDED_BIT});
        cur = c2;
      }
      if (key_c >= amb) {  // (a) became relevant
        tie = 1;
        break;
      }
      if (key_c > pend && pend > -INF) {  // (b) the search has moved past a tied key: was that tie harmless?
        if (n >= B && !(max_dist > pend)) {
          tie = 2;
          break;
        }
        pend = -INF;
      }
      n_hops++;
      PH_MARK(2);
      // link row of this node (requesting the likely NEXT node's row one hop ahead was tried: no gain for a lone
      // query -- the instruction chain, not this latency, bounds it -- and -2 % with every slot busy)
      const uint32_t row0 = lane < M ? links[(uint64_t)(uint32_t)node * (uint32_t)M + lane] : EMPTY_ID;
      PH_MARK(3);

      for (int m0 = 0; m0 < M; m0 += WAVE) {
        const bool act = m0 + lane < M;
        uint32_t id = row0;
        if (m0 > 0) id = act ? links[(uint64_t)(uint32_t)node * (uint32_t)M + m0 + lane] : EMPTY_ID;
        bool isnew;
        if (vg.w == 16) isnew = visited_insert_tag16(vis, vg, act, id, bitmap, ovf_list, ovf_glist, ovf);
        else isnew = visited_insert_tagw(reinterpret_cast<unsigned long long*>(vis), vg, act, id, bitmap, ovf_list, ovf_glist, ovf);
        ovf = __ballot(ovf) != 0ull;
        const unsigned long long newmask = __ballot(isnew);
        const int nn = __popcll(newmask);
        stage_ids[isnew ? __popcll(newmask & ((1ull << lane) - 1ull)) : WAVE] = id;  // keeps link order
        wave_sync();
        PH_MARK(4);
        if (nn == 0) continue;
        n_dist += nn;

        constexpr int VPW = WAVE / G;
        const int v = lane / G;
        const bool group_leader = (lane % G) == 0;
        for (int base = 0; base < nn && !tie; base += VPW * PU) {
          uint32_t cid[PU];
          bool cval[PU];
          float cd[PU];
#pragma unroll
          for (int pu = 0; pu < PU; pu++) {
            const int slot = base + pu * VPW + v;
            cval[pu] = slot < nn;
            cid[pu] = stage_ids[min(slot, nn - 1)];
          }
          const int npass = min(PU, (nn - base + VPW - 1) / VPW);
          batch_dists<T, METRIC, G, CU, FULL>(vectors, row_bytes, nchunks, qlds, cid, npass, cd, lane);
          PH_MARK(5);

          // admissions in link order (Index.h:667-705); superset filter first (max_dist never grows once full)
#pragma unroll
          for (int pu = 0; pu < PU; pu++) {
            if (pu >= npass) break;
            unsigned long long pm = __ballot(group_leader && cval[pu] && !(n >= B && cd[pu] >= max_dist));
            while (pm) {
              const int i = __ffsll((long long)pm) - 1;
              pm &= pm - 1;
              const float di = readlane_f(cd[pu], i);
              const uint32_t idi = (uint32_t)__builtin_amdgcn_readlane((int)cid[pu], i);
              if (!(n < B || di < max_dist)) continue;  // Index.h:693
              if (!(di < INF)) {  // NaN / infinite distance
                tie = 4;
                pm = 0;
                break;
              }
              // (a) full beam whose two largest keys are equal: which one the reference evicts is the library's
              // choice, and the one it evicts stays expandable while max_dist equals its key.  Neither matters
              // unless the search gets that far: remember the key, hand the query over only if a node with a key
              // >= it is about to be expanded or the search ends before max_dist has dropped below it.
              {
                if (n >= B && B >= 2 && second == max_dist) amb = max_dist;
                // entries farther than di move one slot up, top chunk first; the first chunk that holds a member
                // that is not farther fixes the position.  A full beam drops its last entry.
                int pos = 0;
                for (int base = (n - 1) & ~(WAVE - 1); base >= 0; base -= WAVE) {
                  const int idx = base + lane;
                  const unsigned long long e = beam[min(idx, n - 1)];
                  const bool le = idx < n && __uint_as_float((uint32_t)e) <= di;
                  const unsigned long long lem = __ballot(le);
                  beam[(idx < n && !le && idx + 1 < B) ? idx + 1 : B] = e;  // slot B = bin
                  if (lem) {
                    pos = base + __popcll(lem);
                    break;
                  }
                }
                if (lane == 0) beam[pos] = pack(fnv_stl::Entry{di, idi});
                if (n < B) n++;
                cur = min(cur, pos);  // the new entry is unexpanded; members ahead of it did not move
                // keys of the last two entries (wave-uniform addresses: LDS broadcast)
                max_dist = rfl(unpack(beam[n - 1]).key);  // Index.h:702
                second = n >= 2 ? rfl(unpack(beam[n - 2]).key) : -INF;
              }
              if (max_dist < amb) amb = INF;  // every entry with that key is gone from both versions of the beam
            }
            if (tie) break;
          }
          PH_MARK(6);
        }
        wave_sync();  // stage_ids is rewritten by the next row chunk
        if (tie) break;
      }
    }

    ColdArgs c = cold_args();
    const int K = c->K;
    if (!tie && amb < INF) tie = 1;  // (a) still undecided when the search ended
    if (!tie && pend > -INF && n >= B && !(max_dist > pend)) tie = 2;  // (b) likewise
    const int cnt = n < K ? n : K;
    if (!tie) {  // (d) equal keys inside the first K results or across the K-th boundary: std::sort's order
      for (int k0 = 0; k0 < cnt && !tie; k0 += WAVE) {
        const int k = k0 + lane;
        const bool t = k < cnt && k + 1 < n && unpack(beam[min(k, n - 1)]).key == unpack(beam[min(k + 1, n - 1)]).key;
        if (__ballot(t) != 0ull) tie = 3;
      }
    }
    if (tie) {  // search this query again, exactly (results, counters and clean-up are exact_query's)
      if (lane == 0 && tie < 5) {
        uint32_t* rc = c->redo_count;
        atomicAdd(rc, 1u);
        atomicAdd(rc + tie, 1u);  // by reason: [1] eviction tie, [2] selection tie, [3] result tie, [4] NaN/inf
      }
      if (ovf) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, true, lane);
      reset_visited(vis, ovf_list, true, lane);
      __syncthreads();
      ColdArgs xa = cold_args();  // fresh loads: see the note at the top of the loop
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
      x.qlds = reinterpret_cast<uint4*>(smem + xa->off_q);
      x.nbr = reinterpret_cast<unsigned long long*>(smem + xa->off_nbr);
      x.cand = reinterpret_cast<unsigned long long*>(smem + xa->off_cand);
      x.vis = reinterpret_cast<uint32_t*>(smem + xa->off_vis);
      x.stage_ids = reinterpret_cast<uint32_t*>(smem + xa->off_stage_ids);
      x.ovf_list = reinterpret_cast<uint32_t*>(smem + xa->off_ovf);
      exact_query<T, METRIC, G, CU, FULL>(x, qi, entry, best_d, lane, ph);
      PH_FLUSH;
      continue;
    } else {
      const int32_t* labels = c->labels;  // null: construction wants node ids
      float* od_base = c->out_dist + (uint64_t)qi * K;
      int32_t* ol_base = c->out_labels + (uint64_t)qi * K;
      for (int k = lane; k < K; k += WAVE) {
        const bool have = k < cnt;
        const fnv_stl::Entry e = unpack(beam[min(k, n - 1)]);
        const float od = e.key;
        const uint32_t oi = e.val & ~EXPANDED_BIT;
        od_base[k] = have ? od : INF;
        ol_base[k] = have ? (labels ? labels[oi] : (int32_t)oi) : -1;
      }
      if (lane == 0) {
        if (c->out_count) c->out_count[qi] = cnt;
        if (c->out_ndist) c->out_ndist[qi] = n_dist;
        if (c->out_nhops) c->out_nhops[qi] = n_hops;
      }
    }
    if (ovf) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, true, lane);
    PH_MARK(7);
    PH_FLUSH;
    __syncthreads();
  }
}

}  // namespace fnv_dev
