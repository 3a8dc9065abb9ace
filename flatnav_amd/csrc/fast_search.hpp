// fast_search.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip).
//
// beam_search_fast_kernel: the same traversal as beam_search_kernel for beams of at most 64 entries, with the
// beam held as ONE SORTED ARRAY IN REGISTERS (lane i = i-th closest entry) instead of the two binary heaps.
//
// Why this is the same search.  The reference (Index.h:606-707) keeps `neighbors` (max-heap, <= B entries) and
// `candidates` (every admitted node, min-first).  A candidate that has been evicted from `neighbors` has a key
// >= max_dist and max_dist never grows once the beam is full, so when such a candidate reaches the top of
// `candidates` the stop test (Index.h:630) fires; it is never expanded.  The nodes that do get
// expanded are therefore exactly the not-yet-expanded members of `neighbors`, closest first -- which is what
// "first lane whose expanded bit is clear" picks here -- and the admission rule (strict <, link order, max_dist
// refreshed after every admission) is applied verbatim.  The ARRANGEMENT of the reference's heaps (libstdc++'s
// element moves) only decides something when equal keys meet at a decision:
//   (a) eviction: the two largest keys of a full beam are equal (which one goes; a stale candidate with
//       key == max_dist would still be expanded by the reference);
//   (b) selection: the two closest unexpanded members have equal keys (which one is expanded first);
//   (d) result: equal keys among the first K results or across the K-th boundary (std::sort's order).
// Equal keys elsewhere in the beam decide nothing.  Each of the three spots is checked where it arises (one
// readlane + compare); a query that hits one, or meets a NaN / infinite distance, is abandoned and queued on a redo
// list, and beam_search_kernel -- the libstdc++-exact replay -- runs those queries in a second launch on the same
// stream.  Otherwise results, their order and the per-query counters are identical by construction; the parity
// tests compare them bit for bit, tie-heavy inputs included.
//
// What it buys: an admission is one ballot + one wave shift (v_mov_dpp wave_shr:1) + selects (~20 instructions)
// instead of three cooperative heap operations (~250), and picking the next node is a find-first-set instead of a
// heap pop.  The kernel was issue-bound, with the heaps ~55 % of its instructions.
#pragma once
#include "kernels.hpp"
namespace fnv_dev {

__device__ __forceinline__ float wave_shr1(float v, float fill) {
  // lane i <- lane i-1, lane 0 <- fill
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t fill) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);
}

#ifndef FNV_FAST_WAVES_PER_SIMD
#define FNV_FAST_WAVES_PER_SIMD 4
#endif
template <typename T, int METRIC, int G, int CU, bool FULL>
__global__ __launch_bounds__(WAVE, FNV_FAST_WAVES_PER_SIMD) void beam_search_fast_kernel(const SearchParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  uint4* qlds = reinterpret_cast<uint4*>(smem + p.off_q);
  uint32_t* vis = reinterpret_cast<uint32_t*>(smem + p.off_vis);
  uint32_t* stage_ids = reinterpret_cast<uint32_t*>(smem + p.off_stage_ids);
  uint32_t* bitmap = p.ovf_bitmap + (uint64_t)blockIdx.x * p.bitmap_words;
  uint32_t* ovf_list = reinterpret_cast<uint32_t*>(smem + p.off_ovf);
  uint32_t* ovf_glist = p.ovf_glist + (uint64_t)blockIdx.x * p.ovf_cap;
  const int B = p.B;  // <= 64
  const int K = p.K;
  const int M = (int)p.M;
  const float INF = std::numeric_limits<float>::infinity();

  while (true) {
    int qi = 0;
    if (lane == 0) qi = (int)atomicAdd(p.dispenser, 1u);
    qi = rfl(qi);
    if ((uint32_t)qi >= p.nq) break;

    {  // stage the query (zero padded), reset the visited table
      const T* qsrc = reinterpret_cast<const T*>(p.queries) + (uint64_t)qi * p.dim;
      T* qdst = reinterpret_cast<T*>(qlds);
      const int padded = (int)(p.q_chunks * 16u / sizeof(T));
      for (int i = lane; i < padded; i += WAVE) qdst[i] = i < (int)p.dim ? qsrc[i] : T(0);
      uint4* v4 = reinterpret_cast<uint4*>(vis);
      for (uint32_t i = lane; i < p.vis_bytes / 16; i += WAVE) v4[i] = make_uint4(0u, 0u, 0u, 0u);
      if (lane == 0) ovf_list[0] = 0u;
    }
    __syncthreads();

    float best_d;
    uint32_t entry;
    if (p.entry_node) {
      best_d = rfl(p.entry_dist[qi]);
      entry = (uint32_t)rfl((int)p.entry_node[qi]);
    } else {
      entry = scan_entry_points<T, METRIC, G, CU, FULL>(p, qlds, lane, best_d);
    }
    // wave-uniform by construction; say so, or every loop below is compiled as divergent control flow
    best_d = rfl(best_d);
    entry = (uint32_t)rfl((int)entry);

    // the beam: lane i holds the i-th closest entry; lanes >= n hold +inf / junk that is never < a new key
    float kr = lane == 0 ? best_d : INF;
    uint32_t ir = lane == 0 ? entry : EMPTY_ID;
    int n = 1;
    unsigned long long expanded = 0ull;  // bit i: entry i has been expanded
    float max_dist = best_d;
    bool ovf = false;
    if (p.vis_w == 16) visited_insert_tag16(vis, p, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    else visited_insert_tagw(reinterpret_cast<unsigned long long*>(vis), p, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    ovf = __ballot(ovf) != 0ull;
    int tie = best_d != best_d ? 4 : 0;  // why the query is handed to the exact kernel (0 = it is not); NaN entry: 4
    float amb = INF;  // (a) pending (+inf = none): a key at which the reference's eviction choice is unknown (see below)
    uint32_t n_dist = 0, n_hops = 0;
    int pre_node = -1;      // node whose link row was loaded ahead of time (-1: none)
    uint32_t pre_row = 0u;  // ... lane i: its i-th link
    __syncthreads();

    while (!tie) {
      const unsigned long long valid = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
      const unsigned long long avail = ~expanded & valid;
      if (avail == 0ull) break;  // every beam member expanded: what is left in the reference's queue is stale
      const int c = __ffsll((long long)avail) - 1;
      const int node = __builtin_amdgcn_readlane((int)ir, c);
      const unsigned long long rest = avail & (avail - 1ull);
      if (rest != 0ull) {  // (b) the runner-up has the same key: the reference's pop order decides
        const int c2 = __ffsll((long long)rest) - 1;
        if (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(kr), c2)) ==
            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kr), c))) {
          tie = 2;
          break;
        }
      }
      if (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(kr), c)) >= amb) {  // (a) became relevant
        tie = 1;
        break;
      }
      expanded |= 1ull << c;
      n_hops++;
      // link row of this node: already in registers if the previous hop guessed it; and guess the next one now
      // (the runner-up, unless this hop admits something closer) so that its row load overlaps this hop's gather
      uint32_t row0 = pre_row;
      if (node != pre_node) row0 = lane < M ? p.links[(uint64_t)(uint32_t)node * p.M + lane] : EMPTY_ID;
      pre_node = -1;
      if (rest != 0ull) {
        pre_node = __builtin_amdgcn_readlane((int)ir, __ffsll((long long)rest) - 1);
        pre_row = lane < M ? p.links[(uint64_t)(uint32_t)pre_node * p.M + lane] : EMPTY_ID;
      }

      for (int m0 = 0; m0 < M; m0 += WAVE) {
        const bool act = m0 + lane < M;
        uint32_t id = row0;
        if (m0 > 0) id = act ? p.links[(uint64_t)(uint32_t)node * p.M + m0 + lane] : EMPTY_ID;
        bool isnew;
        if (p.vis_w == 16) isnew = visited_insert_tag16(vis, p, act, id, bitmap, ovf_list, ovf_glist, ovf);
        else isnew = visited_insert_tagw(reinterpret_cast<unsigned long long*>(vis), p, act, id, bitmap, ovf_list, ovf_glist, ovf);
        ovf = __ballot(ovf) != 0ull;
        const unsigned long long newmask = __ballot(isnew);
        const int nn = __popcll(newmask);
        stage_ids[isnew ? __popcll(newmask & ((1ull << lane) - 1ull)) : WAVE] = id;  // keeps link order
        wave_sync();
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
          batch_dists<T, METRIC, G, CU, FULL>(p.vectors, p.row_bytes, (int)p.nchunks, qlds, cid, npass, cd, lane);

          // admissions in link order (Index.h:667-705); superset filter first (max_dist never grows once full)
#pragma unroll
          for (int pu = 0; pu < PU; pu++) {
            if (pu >= npass) break;
            unsigned long long pm = __ballot(group_leader && cval[pu] && !(n >= B && cd[pu] >= max_dist));
            while (pm) {
              const int i = __ffsll((long long)pm) - 1;
              pm &= pm - 1;
              const float di = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cd[pu]), i));
              const uint32_t idi = (uint32_t)__builtin_amdgcn_readlane((int)cid[pu], i);
              if (n < B || di < max_dist) {  // Index.h:693
                if (!(di < INF)) {  // NaN / infinite distance
                  tie = 4;
                  pm = 0;
                  break;
                }
                // (a) full beam whose two largest keys are equal: which one the reference evicts is the library's
                // choice, and the one it evicts stays expandable while max_dist equals its key.  Neither matters
                // unless the search gets that far: remember the key, hand the query over only if a node with a key
                // >= it is about to be expanded or the search ends before max_dist has dropped below it.
                if (n >= B && B >= 2 && __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kr), B - 2)) == max_dist)
                  amb = max_dist;
                const int pos = __popcll(__ballot(kr <= di));  // after the members that are not farther
                const float sk = wave_shr1(kr, INF);
                const uint32_t si = wave_shr1(ir, EMPTY_ID);
                kr = lane > pos ? sk : (lane == pos ? di : kr);
                ir = lane > pos ? si : (lane == pos ? idi : ir);
                const unsigned long long low = (1ull << pos) - 1ull;
                expanded = (expanded & low) | ((expanded & ~low) << 1);
                if (n < B) n++;
                max_dist = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kr), n - 1));  // Index.h:702
                if (max_dist < amb) amb = INF;  // every entry with that key is gone from both versions of the beam
              }
            }
            if (tie) break;
          }
        }
        wave_sync();  // stage_ids is rewritten by the next row chunk
        if (tie) break;
      }
    }

    if (!tie && amb < INF) tie = 1;  // (a) still undecided when the search ended
    if (!tie) {  // (d) equal keys inside the first K results or across the K-th boundary: std::sort's order
      const float nxt = __shfl_down(kr, 1, WAVE);
      const int cnt0 = n < K ? n : K;
      if (__ballot(lane < cnt0 && lane + 1 < n && nxt == kr) != 0ull) tie = 3;
    }
    if (tie) {  // queue for the exact kernel (second launch on the same stream)
      if (lane == 0) {
        p.redo_list[atomicAdd(p.redo_count, 1u)] = (uint32_t)qi;
        atomicAdd(p.redo_count + tie, 1u);  // by reason: [1] eviction tie, [2] selection tie, [3] result tie, [4] NaN/inf
      }
    } else {
      const int cnt = n < K ? n : K;
      for (int k = lane; k < K; k += WAVE) {  // K <= B <= 64: one pass
        const bool have = k < cnt;
        p.out_dist[(uint64_t)qi * K + k] = have ? kr : INF;
        p.out_labels[(uint64_t)qi * K + k] = have ? (p.labels ? p.labels[ir] : (int32_t)ir) : -1;
      }
      if (lane == 0) {
        if (p.out_count) p.out_count[qi] = cnt;
        if (p.out_ndist) p.out_ndist[qi] = n_dist;
        if (p.out_nhops) p.out_nhops[qi] = n_hops;
      }
    }
    if (ovf) {  // give the spill bitmap back zeroed
      __threadfence();
      const uint32_t listed = ovf_list[0];
      if (listed <= OVF_LIST + p.ovf_cap) {
        if ((uint32_t)lane < min(listed, OVF_LIST)) bitmap[ovf_list[1 + lane] >> 5] = 0u;
        for (uint32_t i = OVF_LIST + lane; i < listed; i += WAVE) bitmap[ovf_glist[i - OVF_LIST] >> 5] = 0u;
      } else {
        uint4* b4 = reinterpret_cast<uint4*>(bitmap);
        for (uint32_t i = lane; i < p.bitmap_words / 4; i += WAVE) b4[i] = make_uint4(0u, 0u, 0u, 0u);
      }
      __threadfence();
    }
    __syncthreads();
  }
}

}  // namespace fnv_dev
