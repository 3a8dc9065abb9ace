// wire.hpp -- part of the gfx950 engine (device code; included only by beam_search.hip).
// Device-side graph wiring for batched insertion: the reference's selectNeighbors
// (include/flatnav/index/Index.h:714-763) and connectNeighbors (:765-834) for a batch of new nodes whose
// beams (ef_construction nearest wired nodes) the search kernel has just produced.
//
// One 64-lane wave wires one new node u:
//   1. order its beam closest first (equal distances: larger id first, the pop order of the reference's
//      (-distance, id) priority queue), keep a candidate unless an already kept node is strictly closer to it
//      than u is, stop at M/2 kept;
//   2. write u's row (kept nodes, farthest first -- the reference pops a max-heap -- then self-loops);
//   3. for each kept v, under v's lock: take v's first free (self-loop) slot, else re-prune {u} + row(v)
//      with the same rule, keep <= M.
// The pruning is evaluated "kept-major": when k is kept, d(k, c) is computed for every remaining candidate c
// in one gather (batch_dists, the search kernel's distance code) and c is struck out if d(k, c) < d(u, c).
// That is the same predicate as the reference's candidate-major loop, so given equal distance values the kept
// set is identical.  Locks are per-node spin locks in HBM; a wave holds one lock at a time, so no cycles.
#pragma once
#include "distance.hpp"
#include "heaps.hpp"
#include "search_params.h"
namespace fnv_dev {

struct WireParams {
  const uint8_t* vectors;  // [capacity][row_bytes]
  uint32_t* links;         // [capacity][M]
  uint32_t* locks;         // [capacity], 0 = free
  const float* beam_dist;  // [count][W] ascending
  const int32_t* beam_ids; // [count][W] node ids
  const int32_t* beam_count;
  uint32_t* dispenser;
  uint32_t first_node, count, W, M, keep, row_bytes, nchunks, q_chunks;
  uint32_t cap;  // entries per LDS candidate array: max(W, M + 1)
  uint32_t off_q, off_ckey, off_cid, off_okey, off_oid, off_alive, off_kept, off_sel, off_stage_ids, off_stage_idx;
};

__device__ __forceinline__ void stage_vector(uint4* qlds, const uint8_t* vectors, uint32_t row_bytes, int nchunks,
                                             uint32_t id, int lane) {
  const uint4* src = reinterpret_cast<const uint4*>(vectors + (uint64_t)id * row_bytes);
  for (int c = lane; c < nchunks; c += WAVE) qlds[c] = src[c];
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
    stage_vector(qlds, p.vectors, p.row_bytes, (int)p.nchunks, oid[found], lane);
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
        batch_dists<T, METRIC, G, CU, FULL>(p.vectors, p.row_bytes, (int)p.nchunks, qlds, id, npass, d, lane);
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

__device__ __forceinline__ void lock_node(uint32_t* locks, uint32_t node, int lane) {
  if (lane == 0) {
    while (true) {
      unsigned int expected = 0u;
      if (__hip_atomic_compare_exchange_strong(&locks[node], &expected, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT))
        break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void unlock_node(uint32_t* locks, uint32_t node, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (lane == 0) __hip_atomic_store(&locks[node], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int METRIC, int G, int CU, bool FULL>
__global__ __launch_bounds__(WAVE, FNV_MIN_WAVES_PER_SIMD) void wire_batch_kernel(const WireParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  constexpr int VPW = WAVE / G;
  const int vgrp = lane / G;
  uint4* qlds = reinterpret_cast<uint4*>(smem + p.off_q);
  float* ckey = reinterpret_cast<float*>(smem + p.off_ckey);
  uint32_t* cid = reinterpret_cast<uint32_t*>(smem + p.off_cid);
  float* okey = reinterpret_cast<float*>(smem + p.off_okey);
  uint32_t* oid = reinterpret_cast<uint32_t*>(smem + p.off_oid);
  uint32_t* alive = reinterpret_cast<uint32_t*>(smem + p.off_alive);
  uint32_t* kept = reinterpret_cast<uint32_t*>(smem + p.off_kept);
  uint32_t* sel = reinterpret_cast<uint32_t*>(smem + p.off_sel);
  uint32_t* stage_ids = reinterpret_cast<uint32_t*>(smem + p.off_stage_ids);
  uint32_t* stage_idx = reinterpret_cast<uint32_t*>(smem + p.off_stage_idx);
  const int M = (int)p.M;
  for (uint32_t c = p.nchunks + lane; c < p.q_chunks; c += WAVE) qlds[c] = make_uint4(0u, 0u, 0u, 0u);

  while (true) {
    uint32_t i = 0;
    if (lane == 0) i = atomicAdd(p.dispenser, 1u);
    i = (uint32_t)rfl((int)i);
    if (i >= p.count) break;
    const uint32_t u = p.first_node + i;

    // ---- 1. select (Index.h:714-763) ------------------------------------------------------------
    const int C = min(max(rfl(p.beam_count[i]), 0), (int)p.W);
    for (int j = lane; j < C; j += WAVE) {
      ckey[j] = p.beam_dist[(uint64_t)i * p.W + j];
      cid[j] = (uint32_t)p.beam_ids[(uint64_t)i * p.W + j];
    }
    wave_sync();
    rank_order(ckey, cid, C, okey, oid, lane);
    int kept_n;
    if (C < (int)p.keep) {  // Index.h:716-718: fewer candidates than slots -- all of them
      for (int j = lane; j < C; j += WAVE) kept[j] = (uint32_t)j;
      kept_n = C;
      wave_sync();
    } else {
      kept_n = prune_ordered<T, METRIC, G, CU, FULL>(p, qlds, okey, oid, C, (int)p.keep, alive, kept, stage_ids,
                                                     stage_idx, lane);
    }
    // ---- 2. u's own row: kept nodes farthest first, then empty (self-loop) slots -------------------
    for (int j = lane; j < kept_n; j += WAVE) sel[j] = oid[kept[kept_n - 1 - j]];
    wave_sync();
    for (int j = lane; j < M; j += WAVE) p.links[(uint64_t)u * p.M + j] = j < kept_n ? sel[j] : u;

    // ---- 3. back-links (Index.h:765-834) ---------------------------------------------------------
    for (int t = 0; t < kept_n; t++) {
      const uint32_t v = (uint32_t)rfl((int)sel[t]);
      lock_node(p.locks, v, lane);
      uint32_t* vrow = p.links + (uint64_t)v * p.M;
      const uint32_t r = lane < M ? vrow[lane] : v;
      const unsigned long long freem = __ballot(lane < M && r == v);
      if (freem) {
        if (lane == __ffsll((long long)freem) - 1) vrow[lane] = u;
      } else {
        // row is full: candidates = {u} + row(v), distances from v, keep <= M by the same rule
        const int C2 = M + 1;
        if (lane == 0) cid[0] = u;
        if (lane < M) cid[1 + lane] = r;
        stage_vector(qlds, p.vectors, p.row_bytes, (int)p.nchunks, v, lane);
        for (int b = 0; b < C2; b += VPW * PU) {
          uint32_t id[PU];
          float d[PU];
#pragma unroll
          for (int pu = 0; pu < PU; pu++) id[pu] = cid[min(b + pu * VPW + vgrp, C2 - 1)];
          const int npass = min(PU, (C2 - b + VPW - 1) / VPW);
          batch_dists<T, METRIC, G, CU, FULL>(p.vectors, p.row_bytes, (int)p.nchunks, qlds, id, npass, d, lane);
#pragma unroll
          for (int pu = 0; pu < PU; pu++) {
            const int s = b + pu * VPW + vgrp;
            if (pu < npass && s < C2 && (lane % G) == 0) ckey[s] = d[pu];
          }
        }
        wave_sync();
        rank_order(ckey, cid, C2, okey, oid, lane);
        const int k2 = prune_ordered<T, METRIC, G, CU, FULL>(p, qlds, okey, oid, C2, M, alive, kept, stage_ids,
                                                             stage_idx, lane);
        if (lane < M) vrow[lane] = lane < k2 ? oid[kept[k2 - 1 - lane]] : v;
      }
      unlock_node(p.locks, v, lane);
    }
    wave_sync();
  }
}

}  // namespace fnv_dev
