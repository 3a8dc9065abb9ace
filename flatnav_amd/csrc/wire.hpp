// wire.hpp -- part of the gfx950 engine (device code; included only by beam_search.hip).
// Device-side graph wiring for batched insertion: the reference's selectNeighbors
// (include/flatnav/index/Index.h:714-763) and connectNeighbors (:765-834) for a batch of new nodes whose
// beams (ef_construction nearest wired nodes) the search kernel has just produced.
//
// Three steps per batch, no locks, and the result does not depend on scheduling:
//   wire_select_kernel   one 64-lane wave per new node u: order its beam closest first (equal distances: larger
//                        id first, the pop order of the reference's (-distance, id) priority queue), keep a
//                        candidate unless an already kept node is strictly closer to it than u is, stop at M/2
//                        kept; write u's row (kept nodes in the order the reference pops them -- farthest first,
//                        equal distances as libstdc++'s heap leaves them -- then self-loops); record one
//                        back-link request per kept v: req_target[u's slot t] = v.
//   radix sort           the batch's requests by target id (stable: the requesters of one target stay in insertion
//                        order, i.e. ascending new-node id -- the order in which a sequential build meets them).
//   wire_connect_kernel  one wave per target v (= per run of equal keys): if the requesters fit v's free (self-loop)
//                        slots they take them in order (Index.h:789-797), otherwise {row(v)} + {requesters} is
//                        re-pruned with the same rule, keep <= M (Index.h:799-829).
// A wave owns its target's row outright, so hot targets cost one pruning pass over all their requesters
// instead of a lock hand-off per requester (a first version with per-node spin locks spent 3/4 of its time
// in hand-offs on hub nodes).  The reference re-prunes once per arriving back-link; pruning the union once is a
// different member of the same family of outcomes -- and exactly the reference's outcome when every target has
// one requester, in particular for batches of one node (the sequential mode the parity tests use).
// The pruning is evaluated "kept-major": when k is kept, d(k, c) is computed for every remaining candidate c
// in one gather (batch_dists, the search kernel's distance code) and c is struck out if d(k, c) < d(u, c).
// That is the same predicate as the reference's candidate-major loop, so given equal distance values the kept
// set is identical.
#pragma once
#include "distance.hpp"
#include "heaps.hpp"
#include "search_params.h"
namespace fnv_dev {

struct WireParams {
  const uint8_t* vectors;  // [capacity][row_bytes]
  const uint8_t* tails;    // split rows (distance.hpp): [capacity][tail_chunks * 16], else null
  uint32_t* links;         // [capacity][M]
  uint32_t* req_target;    // [count * keep] select: request r = i * keep + t (new node first_node + i) -> target node,
                           // EMPTY_ID for unused slots
  const uint32_t* sorted_target;  // [count * keep] connect: the requests sorted by target (EMPTY_ID last) ...
  const uint32_t* sorted_req;     // ... and their request numbers r, ascending within a target
  const float* beam_dist;  // [count][W] ascending
  const int32_t* beam_ids; // [count][W] node ids
  const int32_t* beam_count;
  uint32_t* dispenser;
  uint32_t first_node, count, W, M, keep, row_bytes, nchunks, q_chunks, tail_chunks;
  uint32_t cap;  // entries per LDS candidate array: max(W, 4 M)
  uint32_t off_q, off_ckey, off_cid, off_okey, off_oid, off_alive, off_kept, off_sel, off_stage_ids, off_stage_idx;
};

// Node `id`'s vector as the "query" of the distances that follow: its main-table chunks, then (split rows) its side-table chunks.
__device__ __forceinline__ void stage_vector(uint4* qlds, const WireParams& p, uint32_t id, int lane) {
  const uint4* src = reinterpret_cast<const uint4*>(p.vectors + (uint64_t)id * p.row_bytes);
  const int nchunks = (int)p.nchunks;
  for (int c = lane; c < nchunks; c += WAVE) qlds[c] = src[c];
  if ((uint32_t)lane < p.tail_chunks) qlds[nchunks + lane] = reinterpret_cast<const uint4*>(p.tails)[(uint64_t)id * p.tail_chunks + (uint32_t)lane];
  wave_sync();
}

// okey/oid = candidates ordered by (key ascending, id descending).
__device__ __forceinline__ void rank_order(const float* ckey, const uint32_t* cid, int C, float* okey, uint32_t* oid,
                                           int lane) {
  for (int j = lane; j < C; j += WAVE) {
    const float k = ckey[j];
    const uint32_t id = cid[j];
    int r = 0;
    for (int l = 0; l < C; l++) {
      const float kl = ckey[l];
      const uint32_t il = cid[l];
      r += (kl < k || (kl == k && (il > id || (il == id && l < j)))) ? 1 : 0;
    }
    okey[r] = k;
    oid[r] = id;
  }
  wave_sync();
}

// Diversity pruning over the ordered candidates; returns how many were kept, their positions in kept[].
template <typename T, int METRIC, int G, int CU, bool FULL>
__device__ __forceinline__ int prune_ordered(const WireParams& p, uint4* qlds, const float* okey, const uint32_t* oid,
                                             int C, int keep, uint32_t* alive, uint32_t* kept, uint32_t* stage_ids,
                                             uint32_t* stage_idx, int lane) {
  constexpr int PU = passes<G, CU>();
  constexpr int VPW = WAVE / G;
  const int v = lane / G;
  for (int j = lane; j < C; j += WAVE) alive[j] = 1u;
  wave_sync();
  int kept_n = 0, pos = 0;
  while (pos < C && kept_n < keep) {
    int found = -1;
    for (int base = pos; base < C; base += WAVE) {  // next candidate that has not been struck out
      const int j = base + lane;
      const unsigned long long m = __ballot(j < C && alive[j] != 0u);
      if (m) {
        found = base + __ffsll((long long)m) - 1;
        break;
      }
    }
    if (found < 0) break;
    if (lane == 0) kept[kept_n] = (uint32_t)found;
    kept_n++;
    pos = found + 1;
    if (kept_n >= keep || pos >= C) break;
    stage_vector(qlds, p, oid[found], lane);
    Query<G, CU> q;
    q.from_lds(qlds, lane, p.tails, p.tail_chunks);
    for (int base = pos; base < C; base += WAVE) {
      const int j = base + lane;
      const bool a = j < C && alive[j] != 0u;
      const unsigned long long m = __ballot(a);
      const int n = __popcll(m);
      if (n == 0) continue;
      const int slot = a ? __popcll(m & ((1ull << lane) - 1ull)) : WAVE;
      stage_ids[slot] = a ? oid[j] : 0u;
      stage_idx[slot] = (uint32_t)j;
      wave_sync();
      for (int b = 0; b < n; b += VPW * PU) {
        uint32_t id[PU];
        float d[PU];
#pragma unroll
        for (int pu = 0; pu < PU; pu++) id[pu] = stage_ids[min(b + pu * VPW + v, n - 1)];
        const int npass = min(PU, (n - b + VPW - 1) / VPW);
        batch_dists<T, METRIC, G, CU, FULL>(p.vectors, p.row_bytes, (int)p.nchunks, q, id, npass, d, lane);
#pragma unroll
        for (int pu = 0; pu < PU; pu++) {
          const int s = b + pu * VPW + v;
          if (pu < npass && s < n && (lane % G) == 0) {
            const uint32_t idx = stage_idx[s];
            if (d[pu] < okey[idx]) alive[idx] = 0u;  // a kept node is closer to it than the base point is
          }
        }
      }
      wave_sync();
    }
  }
  wave_sync();
  return kept_n;
}

// d(base, c) for the C candidates in cid[] -> ckey[] (base's vector is staged in qlds first).
template <typename T, int METRIC, int G, int CU, bool FULL>
__device__ __forceinline__ void keys_from(const WireParams& p, uint4* qlds, uint32_t base, const uint32_t* cid, int C,
                                          float* ckey, int lane) {
  constexpr int PU = passes<G, CU>();
  constexpr int VPW = WAVE / G;
  const int vgrp = lane / G;
  stage_vector(qlds, p, base, lane);
  Query<G, CU> q;
  q.from_lds(qlds, lane, p.tails, p.tail_chunks);
  for (int b = 0; b < C; b += VPW * PU) {
    uint32_t id[PU];
    float d[PU];
#pragma unroll
    for (int pu = 0; pu < PU; pu++) id[pu] = cid[min(b + pu * VPW + vgrp, C - 1)];
    const int npass = min(PU, (C - b + VPW - 1) / VPW);
    batch_dists<T, METRIC, G, CU, FULL>(p.vectors, p.row_bytes, (int)p.nchunks, q, id, npass, d, lane);
#pragma unroll
    for (int pu = 0; pu < PU; pu++) {
      const int s = b + pu * VPW + vgrp;
      if (pu < npass && s < C && (lane % G) == 0) ckey[s] = d[pu];
    }
  }
  wave_sync();
}

// The order in which the reference hands out the kept nodes: it pushes them (closest first) into a max-heap keyed
// on distance only and pops it empty (selectNeighbors' tail Index.h:757-761 + connectNeighbors' loop :775 / :818).
// Distinct keys: simply farthest first.  Equal keys among them: libstdc++'s heap moves decide -- replayed by lane 0
// on the {okey, oid} scratch arrays (dead by now), sel[] receives the ids in pop order.
__device__ __forceinline__ void pop_order(const float* okey_in, const uint32_t* oid_in, const uint32_t* kept, int kept_n,
                                          unsigned long long* heap, uint32_t* sel, int lane) {
  bool tie = false;
  for (int j = lane; j + 1 < kept_n; j += WAVE) tie |= okey_in[kept[j]] == okey_in[kept[j + 1]];  // kept[] ascends in key
  if (__ballot(tie) == 0ull) {
    for (int j = lane; j < kept_n; j += WAVE) sel[j] = oid_in[kept[kept_n - 1 - j]];
    wave_sync();
    return;
  }
  if (lane == 0) {
    LdsHeap h{heap};
    for (int j = 0; j < kept_n; j++) fnv_stl::heap_push(h, j, fnv_stl::Entry{okey_in[kept[j]], oid_in[kept[j]]});
    for (int m = kept_n; m >= 1; m--) {
      sel[kept_n - m] = h.get(0).val;
      fnv_stl::heap_pop(h, m);
    }
  }
  wave_sync();
}

#define FNV_WIRE_LDS                                                          \
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        \
  const int lane = threadIdx.x;                                               \
  uint4* qlds = reinterpret_cast<uint4*>(smem + p.off_q);                     \
  float* ckey = reinterpret_cast<float*>(smem + p.off_ckey);                  \
  uint32_t* cid = reinterpret_cast<uint32_t*>(smem + p.off_cid);              \
  float* okey = reinterpret_cast<float*>(smem + p.off_okey);                  \
  uint32_t* oid = reinterpret_cast<uint32_t*>(smem + p.off_oid);              \
  uint32_t* alive = reinterpret_cast<uint32_t*>(smem + p.off_alive);          \
  uint32_t* kept = reinterpret_cast<uint32_t*>(smem + p.off_kept);            \
  uint32_t* stage_ids = reinterpret_cast<uint32_t*>(smem + p.off_stage_ids);  \
  uint32_t* stage_idx = reinterpret_cast<uint32_t*>(smem + p.off_stage_idx);  \
  const int M = (int)p.M;                                                     \
  for (uint32_t c = p.nchunks + p.tail_chunks + lane; c < p.q_chunks; c += WAVE) qlds[c] = make_uint4(0u, 0u, 0u, 0u);

template <typename T, int METRIC, int G, int CU, bool FULL>
__global__ __launch_bounds__(WAVE, FNV_MIN_WAVES_PER_SIMD) void wire_select_kernel(const WireParams p) {
  FNV_WIRE_LDS
  uint32_t* sel = reinterpret_cast<uint32_t*>(smem + p.off_sel);
  while (true) {
    uint32_t i = 0;
    if (lane == 0) i = atomicAdd(p.dispenser, 1u);
    i = (uint32_t)rfl((int)i);
    if (i >= p.count) break;
    const uint32_t u = p.first_node + i;
    // ---- select (Index.h:714-763) ---------------------------------------------------------------
    const int C = min(max(rfl(p.beam_count[i]), 0), (int)p.W);
    for (int j = lane; j < C; j += WAVE) {
      ckey[j] = p.beam_dist[(uint64_t)i * p.W + j];
      cid[j] = (uint32_t)p.beam_ids[(uint64_t)i * p.W + j];
    }
    wave_sync();
    rank_order(ckey, cid, C, okey, oid, lane);
    int kept_n;
    if (C < (int)p.keep) {
      // Index.h:715-717: fewer candidates than slots -- all of them, and the beam's own heap is popped as it is: the
      // search wrote the beam closest first with equal distances in reverse pop order (kernels.hpp, result tail)
      for (int j = lane; j < C; j += WAVE) sel[j] = cid[C - 1 - j];
      kept_n = C;
      wave_sync();
    } else {
      kept_n = prune_ordered<T, METRIC, G, CU, FULL>(p, qlds, okey, oid, C, (int)p.keep, alive, kept, stage_ids,
                                                     stage_idx, lane);
      // ---- kept nodes in the reference's pop order ----------------------------------------------------------
      pop_order(okey, oid, kept, kept_n, reinterpret_cast<unsigned long long*>(alive), sel, lane);
    }
    // ---- u's own row: kept nodes, then empty (self-loop) slots ----------------------------------------
    for (int j = lane; j < M; j += WAVE) p.links[(uint64_t)u * p.M + j] = j < kept_n ? sel[j] : u;
    // ---- the back-link requests (Index.h:783: "add the reverse edge"), in the order the row lists them ----
    for (int t = lane; t < (int)p.keep; t += WAVE) p.req_target[i * p.keep + (uint32_t)t] = t < kept_n ? sel[t] : EMPTY_ID;
    wave_sync();
  }
}

// One unit of work = 64 consecutive positions of the sorted request list; the wave handles every target whose run
// of requests STARTS inside its block (runs may extend into the next blocks).
template <typename T, int METRIC, int G, int CU, bool FULL>
__global__ __launch_bounds__(WAVE, FNV_MIN_WAVES_PER_SIMD) void wire_connect_kernel(const WireParams p) {
  FNV_WIRE_LDS
  uint32_t* sel = reinterpret_cast<uint32_t*>(smem + p.off_sel);
  const uint32_t total = p.count * p.keep;
  const uint32_t nblocks = (total + WAVE - 1) / WAVE;
  while (true) {
    uint32_t blk = 0;
    if (lane == 0) blk = atomicAdd(p.dispenser, 1u);
    blk = (uint32_t)rfl((int)blk);
    if (blk >= nblocks) break;
    const uint32_t pos0 = blk * WAVE + lane;
    const uint32_t mykey = pos0 < total ? p.sorted_target[pos0] : EMPTY_ID;
    const uint32_t prevkey = (pos0 > 0 && pos0 < total) ? p.sorted_target[pos0 - 1] : EMPTY_ID;
    unsigned long long starts = __ballot(mykey != EMPTY_ID && (pos0 == 0 || prevkey != mykey));
    while (starts) {
      const int sl = __ffsll((long long)starts) - 1;
      starts &= starts - 1;
      const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)mykey, sl);
      uint32_t rpos = blk * WAVE + (uint32_t)sl;  // next unread request of v
      uint32_t* vrow = p.links + (uint64_t)v * p.M;
      // row(v): real members first, in slot order
      const uint32_t mine = lane < M ? vrow[lane] : v;
      const bool real = lane < M && mine != v;
      const unsigned long long realm = __ballot(real);
      int n = __popcll(realm);
      cid[real ? __popcll(realm & ((1ull << lane) - 1ull)) : (int)p.cap] = mine;  // slot cap = bin
      const int n_row = n;
      bool pruned = false;
      bool more = true;
      wave_sync();
      while (true) {
        // requesters in insertion order; as many as the candidate arrays hold
        while (more && n < (int)p.cap) {
          const int room = (int)p.cap - n;
          const uint32_t q = rpos + (uint32_t)lane;
          const bool mineq = lane < room && q < total && p.sorted_target[q] == v;
          const unsigned long long mm = __ballot(mineq);  // a prefix of the lanes (the list is sorted)
          const int got = __popcll(mm);
          if (mineq) cid[n + lane] = p.first_node + p.sorted_req[q] / p.keep;
          n += got;
          rpos += (uint32_t)got;
          if (got < min(room, WAVE)) more = false;  // ran into the next target / the end of the list
        }
        wave_sync();
        if (n > M) {  // does not fit: re-prune the union from v's point of view (Index.h:799-829)
          keys_from<T, METRIC, G, CU, FULL>(p, qlds, v, cid, n, ckey, lane);
          rank_order(ckey, cid, n, okey, oid, lane);
          const int k2 = prune_ordered<T, METRIC, G, CU, FULL>(p, qlds, okey, oid, n, M, alive, kept, stage_ids,
                                                               stage_idx, lane);
          pop_order(okey, oid, kept, k2, reinterpret_cast<unsigned long long*>(alive), sel, lane);
          for (int t = lane; t < k2; t += WAVE) cid[t] = sel[t];
          n = k2;
          pruned = true;
          wave_sync();
        }
        if (!more) break;
      }
      if (pruned) {
        if (lane < M) vrow[lane] = lane < n ? cid[lane] : v;
      } else {
        // everything fitted: requesters take the free slots in slot order (Index.h:789-797), members stay put
        const unsigned long long freem = __ballot(lane < M && !real);
        if (lane < M && !real) {
          const int k = __popcll(freem & ((1ull << lane) - 1ull));
          if (n_row + k < n) vrow[lane] = cid[n_row + k];
        }
      }
      wave_sync();
    }
  }
}
#undef FNV_WIRE_LDS

}  // namespace fnv_dev
