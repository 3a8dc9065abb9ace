// merged_beam.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip / kernel_inst.hip).
//
// beam_search_merged_kernel: the sorted-beam search (sorted_beam.hpp: same traversal as the reference's two heaps,
// same tie rules, same in-wave exact re-run) for beams of up to R * 64 entries (instantiated for R = 4 and R = 1), with
//   * the beam RESIDENT IN REGISTERS: entry e lives in lane e % 64 of register pair (kr, ir)[e / 64], closest first;
//     bit 31 of the id word is the "expanded" flag, entries beyond the beam's size hold {+inf, EMPTY_ID} (whose bit
//     31 is set, so they never look unexpanded);
//   * ONE MERGE PER LINK ROW instead of one insertion per admitted neighbour: the row's distances are staged in LDS,
//     lane j takes the j-th evaluated neighbour (link order), and every element of beam U candidates computes its
//     position in the stable merge (beam entries before candidates of equal key, candidates in link order -- the
//     arrangement the one-by-one insertions of sorted_beam.hpp produce) from wave ballots:
//         beam entry e      -> e + #{candidates with a smaller key}
//         candidate j       -> #{beam keys <= d_j} + #{candidates before j in (key, link order)}
//     The permutation itself goes through the LDS beam array (one scatter, one read back of the chunks that moved).
//
// Why the merge is the reference's admission loop (Index.h:693-704).  Taken one by one in link order, a neighbour is
// admitted iff the beam is not full or d < max_dist, and a full beam then drops its farthest member.  An element
// among the B smallest of beam U row is never dropped (when it is the farthest of a full beam and something closer
// arrives, B elements are closer than it) and never refused; an element outside is refused or dropped by the end of
// the row -- so the beam after the row is the B smallest of the union, whatever the order, PROVIDED no two keys are
// equal where that cut falls.  Equal keys above the cut change nothing that lasts (both are gone by the end of the
// row; nothing is expanded in between).  Equal keys AT the cut (an element left outside has the key of the new
// farthest member) are sorted_beam.hpp's case (a): which one the reference keeps is the library's choice, and the
// one it drops stays expandable while max_dist equals its key -- so `amb` is set to that key and the query is handed
// to the exact search only if the search gets that far (same deferred rule; this kernel flags a refused candidate
// with d == max_dist as well, which the reference decides by its strict '<' -- conservative, never wrong).
// A NaN / infinite distance that could be admitted hands the query over at once.
#pragma once
#include "sorted_beam.hpp"
namespace fnv_dev {

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

template <typename T, int METRIC, int G, int CU, bool FULL, int R>
__global__ __launch_bounds__(WAVE, (CU == 1 && R == 1) ? 5 : FNV_SORTED_WAVES_PER_SIMD) void beam_search_merged_kernel(const SearchParams p) {
  // (128-byte rows, one-chunk beam: 97 registers as compiled for four waves per SIMD -- one over the budget of five)
  constexpr int PU = passes<G, CU>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const float INF = std::numeric_limits<float>::infinity();

  while (true) {
    const int qi = next_query(lane);
    if (qi < 0) break;
    PH_DECL
    ColdArgs ca = cold_args();  // per-query constants: see sorted_beam.hpp
    const uint8_t* const vectors = ca->vectors;
    const uint32_t* const links = ca->links;
    const uint32_t row_bytes = ca->row_bytes;
    const int nchunks = (int)ca->nchunks;
    const int B = ca->B;  // <= R * WAVE
    const int M = (int)ca->M;
    const VisGeom vg{ca->vis_nmask, ca->vis_rshift, ca->vis_rmask, ca->vis_mult, ca->vis_w};
    uint4* qlds = reinterpret_cast<uint4*>(smem + ca->off_q);
    uint32_t* vis = reinterpret_cast<uint32_t*>(smem + ca->off_vis);
    uint32_t* stage_ids = reinterpret_cast<uint32_t*>(smem + ca->off_stage_ids);
    float* stage_d = reinterpret_cast<float*>(smem + ca->off_stage_d);  // aliases `beam` below: used between merges
    uint32_t* ovf_list = reinterpret_cast<uint32_t*>(smem + ca->off_ovf);
    // [B + 2] at 16n + 8 (word -1 = write-only bin): the merge's permutation buffer, and the neighbours heap of an
    // exact re-run
    unsigned long long* beam = reinterpret_cast<unsigned long long*>(smem + ca->off_nbr);
    stage_query<T>(qlds, vis, ovf_list, qi, true, lane);
    __syncthreads();
    PH_MARK(0);

    float best_d;
    uint32_t entry = entry_point<T, METRIC, G, CU, FULL>(vectors, row_bytes, nchunks, qlds, qi, lane, best_d);
    PH_MARK(1);
    best_d = rfl(best_d);
    entry = (uint32_t)rfl((int)entry);
    uint32_t* const bitmap = cold_args()->ovf_bitmap + (uint64_t)blockIdx.x * cold_args()->bitmap_words;
    uint32_t* const ovf_glist = cold_args()->ovf_glist + (uint64_t)blockIdx.x * cold_args()->ovf_cap;

    float kr[R];
    uint32_t ir[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      kr[r] = INF;
      ir[r] = EMPTY_ID;
    }
    if (lane == 0) {
      kr[0] = best_d;
      ir[0] = entry;
    }
    int n = 1;
    float max_dist = best_d;
    bool ovf = false;
    if (vg.w == 16) visited_insert_tag16(vis, vg, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    else visited_insert_tagw(reinterpret_cast<unsigned long long*>(vis), vg, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    ovf = __ballot(ovf) != 0ull;
    int tie = best_d != best_d ? 4 : 0;
    if ((uint32_t)qi + ca->tail_exact >= ca->nq) tie = 5;  // last round of the launch: straight to the exact search
    float amb = INF;    // (a) pending: key at which the reference's eviction choice is unknown
    float pend = -INF;  // (b) pending: largest key at which two unexpanded members tied
    uint32_t n_dist = 0, n_hops = 0;
    __syncthreads();

    while (!tie) {
      // ---- pick the closest unexpanded member; (b) its runner-up must not have the same key -------------------
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
      const int node = lane_of(ir, r1, l1);
      const float key_c = lane_of(kr, r1, l1);
      if (r2 >= 0 && lane_of(kr, r2, l2) == key_c) pend = fmaxf(pend, key_c);
#pragma unroll
      for (int r = 0; r < R; r++)
        if (r == r1) ir[r] |= lane == l1 ? EXPANDED_BIT : 0u;
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
          batch_dists<T, METRIC, G, CU, FULL>(vectors, row_bytes, nchunks, qlds, cid, npass, cd, lane);
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
        const float d = stage_d[lane] + 0.0f;  // lane j: j-th neighbour in link order (-0 -> +0 for float_ord)
        const uint32_t cand_id = stage_ids[lane];
        const bool full0 = n >= B;
        const bool pass = lane < nn && (!full0 || d < max_dist);  // superset of what one-by-one admission lets in
        const unsigned long long pm = __ballot(pass);
        if (pm != 0ull) {
          if (__ballot(pass && !(d < INF)) != 0ull) {  // NaN / infinite distance that could be admitted
            tie = 4;
            break;
          }
          const int c = __popcll(pm);
          const unsigned long long key64 = ((unsigned long long)float_ord(d) << 32) | (uint32_t)lane;
          uint32_t rank = 0;   // candidate lane: candidates that precede it in (key, link order)
          int bpos = 0;        // candidate lane: beam keys <= its key
          uint32_t le[R];      // beam lane: candidates whose key is >= the entry's key (they go after it)
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
              if (r * WAVE < n) {  // wave-uniform
                const bool b = kr[r] <= di;  // entries beyond n hold +inf
                cnt += __popcll(__ballot(b));
                le[r] += b ? 1u : 0u;
              }
            }
            bpos = lane == i ? cnt : bpos;
          }
          const int fpos = bpos + (int)rank;  // candidate's position in the stable merge
          const int n_new = min(B, n + c);
          bool out_eq = false;    // this lane's element is left outside the new beam (checked against the new max below)
          float out_key = 0.f;
          // scatter through LDS; a full chunk none of whose entries moves keeps its registers (the candidates all
          // land above it)
          bool moved[R];
#pragma unroll
          for (int r = 0; r < R; r++) {
            moved[r] = false;
            if (r * WAVE < n_new) {
              const int e = r * WAVE + lane;
              const int np = e + c - (int)le[r];
              const bool valid = e < n;
              moved[r] = (r + 1) * WAVE > n || __ballot(valid && (int)le[r] != c) != 0ull;
              if (moved[r]) {
                const bool keep = valid && np < B;
                beam[keep ? np : -1] = pack(fnv_stl::Entry{kr[r], ir[r]});
                if (valid && !keep) {
                  out_eq = true;
                  out_key = kr[r];
                }
              }
            }
          }
          {
            const bool keep = pass && fpos < B;
            beam[keep ? fpos : -1] = pack(fnv_stl::Entry{d, cand_id});
            if (pass && !keep) {
              out_eq = true;
              out_key = d;
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
          // (a) an element left outside has the key of the new farthest member
          if (__ballot(out_eq && out_key == max_dist) != 0ull) amb = max_dist;
          if (max_dist < amb) amb = INF;  // every entry with that key is gone from both versions of the beam
          wave_sync();  // the LDS buffer is rewritten by the next merge
        }
        PH_MARK(6);
        wave_sync();  // stage_ids / stage_d are rewritten by the next row chunk
      }
    }

    ColdArgs c = cold_args();
    const int K = c->K;
    if (!tie && amb < INF) tie = 1;  // (a) still undecided when the search ended
    if (!tie && pend > -INF && n >= B && !(max_dist > pend)) tie = 2;  // (b) likewise
    const int cnt = n < K ? n : K;
    if (!tie) {  // (d) equal keys inside the first K results or across the K-th boundary: std::sort's order
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
    if (tie) {  // search this query again, exactly (results, counters and clean-up are exact_query's)
      if (lane == 0 && tie < 5) {
        uint32_t* rc = c->redo_count;
        atomicAdd(rc, 1u);
        atomicAdd(rc + tie, 1u);
      }
      if (ovf) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, true, lane);
      reset_visited(vis, ovf_list, true, lane);
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
