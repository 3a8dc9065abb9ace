// kernels.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip).
// Entry-point scan, the K0 batch kernel, the exact (two-heap) search of one query and its kernel.
#pragma once
#include "distance.hpp"
#include "heaps.hpp"
#include "visited.hpp"
namespace fnv_dev {

// ---------------------------------------------------------------------------------------------
// Entry-point selection (Index.h:845-870): argmin over nodes 0, s, 2s, ... ; strict '<', so the FIRST minimum
// wins.  Per lane the node index only grows, so '<' keeps the earliest; across lanes the tie goes to the
// smaller index.  `rows`/`stride` address row j of the scan set (HBM: j*step-th vector; LDS tile: j-th row).
// ---------------------------------------------------------------------------------------------
template <typename T, int METRIC, int G, int CU, bool FULL>
__device__ __forceinline__ void scan_rows(const uint8_t* rows, uint32_t stride_rows, uint32_t id_mul, int nchunks,
                                          const Query<G, CU>& q, uint32_t count, uint32_t j_base, int lane, float& best_d,
                                          uint32_t& best_j) {
  constexpr int PU = passes<G, CU>();
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
    batch_dists<T, METRIC, G, CU, FULL>(rows, stride_rows, nchunks, q, sid, npass, sd, lane);
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

// ---------------------------------------------------------------------------------------------
// K0: entry points for the whole batch.  Every query scans the SAME ceil(N/step) nodes, so a workgroup
// (4 waves) stages them once in LDS (tiles of scan_tile_rows rows, row stride padded by 16 bytes against
// bank conflicts) and runs SCAN_QPB queries against the tile; distances use the very same batch_dists code
// as the search kernel, so entry_dist equals what the search kernel would have computed, bit for bit.
// ---------------------------------------------------------------------------------------------

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
      Query<G, CU> q;
      q.from_lds(qlds, lane);
      float best_d = std::numeric_limits<float>::max();
      uint32_t best_j = 0;
      scan_rows<T, METRIC, G, CU, FULL>(tile, p.scan_tile_stride, 1u, (int)p.nchunks, q, rows, t0, lane, best_d, best_j);
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

// Per-query prologue / epilogue pieces shared by the search kernels (cold: once per query, so they read their
// parameters from the kernel-argument segment instead of keeping them in scalar registers -- see cold_args()).
__device__ __forceinline__ void reset_visited(uint32_t* vis, uint32_t* ovf_list, bool tagged, int lane) {
  uint4* v4 = reinterpret_cast<uint4*>(vis);
  const uint32_t fill = tagged ? 0u : EMPTY_ID;
  const uint32_t n16 = cold_args()->vis_bytes / 16;
  for (uint32_t i = lane; i < n16; i += WAVE) v4[i] = make_uint4(fill, fill, fill, fill);
  if (lane == 0) ovf_list[0] = 0u;
  static_assert(STASH <= (uint32_t)WAVE, "one store per lane empties the stash");
  if ((uint32_t)lane < STASH) ovf_list[OVF_LIST + 2 + lane] = 0u;  // the stash is empty
}

// The query of work item qi: staged in LDS (zero padded to q_chunks), or -- rows of one 192-chunk span, distance.hpp --
// straight from the caller's array into the lane's registers (no LDS is reserved for it then).
template <typename T, int G, int CU>
__device__ __forceinline__ void stage_query(Query<G, CU>& q, uint4* qlds, uint32_t* vis, uint32_t* ovf_list, int qi, bool tagged,
                                            int lane) {
  ColdArgs c = cold_args();
  const uint32_t dim = c->dim;
  const T* qsrc = reinterpret_cast<const T*>(c->queries) + (uint64_t)qi * dim;
  q.lds = qlds;
  q.tails = c->tails;  // (split rows, distance.hpp; dead code in every other row configuration)
  q.tail_chunks = c->tail_chunks;
  if constexpr (query_in_regs<G, CU>()) {
    constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
#pragma unroll
    for (int cu = 0; cu < CU; cu++) {
      T e[EPC];
      const int first = (cu * G + lane % G) * EPC;
#pragma unroll
      for (int k = 0; k < EPC; k++) e[k] = first + k < (int)dim ? qsrc[first + k] : T(0);
      q.r[cu] = __builtin_bit_cast(uint4, e);
    }
  } else {
    T* qdst = reinterpret_cast<T*>(qlds);
    const int padded = (int)(c->q_chunks * 16u / sizeof(T));
    for (int i = lane; i < padded; i += WAVE) qdst[i] = i < (int)dim ? qsrc[i] : T(0);
  }
  reset_visited(vis, ovf_list, tagged, lane);
}

// Give the slot's HBM visited bitmap back zeroed: word by word while every id that went there is on record (the
// first OVF_LIST in LDS, then ovf_cap more in the slot's HBM list), else the whole bitmap with 16-byte stores.
__device__ __forceinline__ void clear_spill_bitmap(uint32_t* bitmap, const uint32_t* ovf_list, const uint32_t* ovf_glist,
                                                   bool tagged, int lane) {
  ColdArgs c = cold_args();
  __threadfence();
  const uint32_t listed = ovf_list[0];
  if (tagged && listed <= OVF_LIST + c->ovf_cap) {
    if ((uint32_t)lane < min(listed, OVF_LIST)) bitmap[ovf_list[1 + lane] >> 5] = 0u;
    for (uint32_t i = OVF_LIST + lane; i < listed; i += WAVE) bitmap[ovf_glist[i - OVF_LIST] >> 5] = 0u;
  } else {
    uint4* b4 = reinterpret_cast<uint4*>(bitmap);
    const uint32_t n16 = c->bitmap_words / 4;
    for (uint32_t i = lane; i < n16; i += WAVE) b4[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __threadfence();
}

// Next query of this slot (-1: none left) from the launch's atomic dispenser.
__device__ __forceinline__ int next_query(int lane) {
  ColdArgs c = cold_args();
  int qi = 0;
  if (lane == 0) qi = (int)atomicAdd(c->dispenser, 1u);
  qi = rfl(qi);
  if ((uint32_t)qi >= c->nq) return -1;
  return qi;
}

// Entry point of query qi and its distance: from the batch kernel K0 if it ran, else the in-kernel scan.
template <typename T, int METRIC, int G, int CU, bool FULL>
__device__ __forceinline__ uint32_t entry_point(const uint8_t* vectors, uint32_t row_bytes, int nchunks, const Query<G, CU>& q,
                                                int qi, int lane, float& best_d) {
  ColdArgs c = cold_args();
  const uint32_t* en = c->entry_node;
  if (en) {
    best_d = rfl(c->entry_dist[qi]);
    return (uint32_t)rfl((int)en[qi]);
  }
  best_d = std::numeric_limits<float>::max();
  uint32_t best_j = 0;
  const uint32_t step = c->scan_step;
  scan_rows<T, METRIC, G, CU, FULL>(vectors, row_bytes, step, nchunks, q, c->n_scan, 0u, lane, best_d, best_j);
  wave_argmin(best_d, best_j);
  return best_j * step;
}

// ---------------------------------------------------------------------------------------------
// The exact search of ONE query (Index.h:606-707 + :393-408): the reference's two binary heaps moved with
// libstdc++'s element moves, link-order admissions, result tail.  Called by beam_search_kernel for every query
// and by beam_search_merged_kernel for the queries in which equal keys met at a decision.
// Expects the query staged (q), the visited table reset, entry / best_d chosen.
// ---------------------------------------------------------------------------------------------
struct ExactCtx {
  const uint8_t* vectors;
  const uint32_t* links;
  uint32_t row_bytes;
  int nchunks, B, M;
  int cand_slots;  // entries of the candidates heap that live in LDS (0: the whole heap is in the HBM spill area)
  bool tagged;
  VisGeom vg;
  unsigned long long* nbr;   // LDS, B + 2 entries, at 16n + 8
  unsigned long long* cand;  // LDS, cand_slots + 1 entries, at 16n + 8 (unused when cand_slots == 0)
  uint32_t* vis;
  uint32_t* stage_ids;
  uint32_t* ovf_list;
};

// ---------------------------------------------------------------------------------------------
// The hand-over log (round 5).  The merged-beam kernel writes, per hop, one HEADER record {node, evaluated neighbours,
// candidates that follow} and the row's neighbours that could still be admitted when the row began (distance < max_dist of
// that moment, or the beam not yet full) as CANDIDATE records {distance, id} in link order -- 8 bytes each, in the slot's
// HBM log area.  When equal keys meet at a decision, the query is not searched again from scratch: the reference's two
// heaps are REPLAYED from the log (no vector is loaded, no distance computed, the visited set is the one the merged-beam
// pass left), and the exact search continues from there.  Per logged hop the reference's loop head runs first
// (Index.h:625-633): if its top is not the logged node -- equal keys, the reference expands another node first -- the replay
// stops at that hop, the visited set is rebuilt as {entry} + every link of the nodes expanded so far (Index.h:679-684 marks
// every link it looks at) and the exact search continues from THAT state.  By induction over the hops the replayed state is
// the reference's: the same node is expanded against the same visited set, so the same neighbours are evaluated, and a
// neighbour the log leaves out is refused by the reference as well (max_dist never grows once the beam is full).
// CPU model of exactly this procedure against the oracle: oracle.replay_search, tests/test_replay_model.py.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t LOG_HDR = 0xFFC00000u;  // key bits of a header record: a NaN no distance ever is (candidates are finite)
constexpr int LOG_MAX_LINKS = 63;          // rows of up to 63 links are logged (a hop is at most 64 records)
__device__ __forceinline__ unsigned long long log_header(uint32_t node, uint32_t evaluated, uint32_t cands) {
  return (unsigned long long)(LOG_HDR | (evaluated << 8) | cands) | ((unsigned long long)node << 32);
}

// Sequential reader: records [base, base + 64) live in registers (lane i: record base + i), the next 64 are requested as
// soon as the window moves -- a replay consumes ~6 records per hop, so the load has ten hops to arrive.
struct LogReader {
  const unsigned long long* log;
  uint32_t n, base;
  unsigned long long cur, next;
  __device__ __forceinline__ void open(const unsigned long long* l, uint32_t count, int lane) {
    log = l;
    n = count;
    base = 0;
    cur = load(min((uint32_t)lane, n - 1));
    next = load(min(WAVE + (uint32_t)lane, n - 1));
  }
  // past this CU's vector L1 (which may still hold lines of this slot's log area from an earlier query's replay): the
  // records were written by this wave moments ago and sit in the XCD's L2
  __device__ __forceinline__ unsigned long long load(uint32_t i) const {
    return __hip_atomic_load(log + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ fnv_stl::Entry get(uint32_t r, int lane) {  // r wave-uniform, base <= r < n, never moves back
    while (r >= base + WAVE) {
      cur = next;
      base += WAVE;
      next = load(min(base + WAVE + (uint32_t)lane, n - 1));
    }
    const int l = (int)(r - base);
    fnv_stl::Entry e;
    e.key = __int_as_float(__builtin_amdgcn_readlane((int)(uint32_t)cur, l));
    e.val = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cur >> 32), l);
    return e;
  }
};

// The exact search's state between two admissions (wave-uniform).
struct ExactState {
  int nbr_n, cand_n;
  float max_dist;
  int err;
};

// One admission (Index.h:693-704) of (di, idi) into both heaps; the caller has checked nothing.
__device__ __forceinline__ void exact_admit(const ExactCtx& x, LdsHeap& nbr, LdsHeap& cand, unsigned long long* spill,
                                            ExactState& s, float di, uint32_t idi, int lane, PhaseTimer& ph) {
  const int B = x.B, cand_slots = x.cand_slots;
  if (!(s.nbr_n < B || di < s.max_dist)) return;  // Index.h:693
  // max_dist == neighbors.top().first throughout (Index.h:702), so the new top is known without reading
  // the heap back: a push changes it only if the new element climbs to the root, a pop reports it
  bool at_root;
  if (s.cand_n < cand_slots) {
    at_root = coop_push2(cand, s.cand_n, fnv_stl::Entry{-di, idi}, nbr, s.nbr_n, fnv_stl::Entry{di, idi}, lane, ph, 11);
  } else {
    const int spill_entries = (int)cold_args()->spill_entries;
    if (s.cand_n >= cand_slots + spill_entries) {
      s.err = ST_CAND_OVERFLOW;
      return;
    }
    CandHeap cand_big{cand.p, spill, cand_slots};
    __threadfence_block();
    coop_push(cand_big, s.cand_n, fnv_stl::Entry{-di, idi}, lane, ph, 11);
    __threadfence_block();
    at_root = coop_push(nbr, s.nbr_n, fnv_stl::Entry{di, idi}, lane, ph, 12);
  }
  if (at_root) s.max_dist = di;
  if (s.nbr_n + 1 > B) s.max_dist = coop_pop<false>(nbr, s.nbr_n + 1, lane, ph, 13);
  s.cand_n++;
  if (s.nbr_n < B) s.nbr_n++;
}

// The candidates heap's top (Index.h:626) and its removal (:634), wherever the heap lives.
__device__ __forceinline__ fnv_stl::Entry exact_cand_top(const ExactCtx& x, const LdsHeap& cand, const unsigned long long* spill) {
  return x.cand_slots > 0 ? cand.get(0) : unpack(spill[0]);  // same address in every lane: broadcast
}
__device__ __forceinline__ void exact_cand_pop(const ExactCtx& x, LdsHeap& cand, unsigned long long* spill, ExactState& s, int lane,
                                               PhaseTimer& ph) {
  if (s.cand_n <= x.cand_slots) {
    coop_pop<false>(cand, s.cand_n, lane, ph, 8);
  } else {  // part of the heap lives in the HBM spill area
    CandHeap cand_big{cand.p, spill, x.cand_slots};
    __threadfence_block();
    coop_pop<false>(cand_big, s.cand_n, lane, ph, 8);
    __threadfence_block();
  }
  s.cand_n--;
}

// What a search that was RESUMED from a log starts with (the heaps are in place, the visited set is current).
struct ExactResume {
  ExactState s;
  uint32_t n_dist, n_hops;
  bool ovf;
};

// `stop` (null: never) = a word another wavefront sets once the same query has been answered (shadow mode, search_params.h):
// polled once per hop; the search then gives up -- no results written, per-slot state left clean.
// `resumed` (false: a search from scratch): heaps, visited set and counters come from a replayed log (`rs`); the search starts
// at the loop head.  (A run-time flag, not a template parameter: the merged-beam kernel inlines this function ONCE for both;
// `rs` by value: a pointer to it would keep the struct in scratch memory.)
template <typename T, int METRIC, int G, int CU, bool FULL, bool DIRECT = false>
__device__ __forceinline__ void exact_query(const ExactCtx& x, const Query<G, CU>& q, int qi, uint32_t entry, float best_d, int lane,
                                            PhaseTimer& ph, const uint32_t* stop = nullptr, bool resumed = false,
                                            ExactResume rs = ExactResume{ExactState{1, 1, 0.f, ST_OK}, 0u, 0u, false}) {
  // (the lane index is made opaque here: everything derived from it below -- lane masks, group indices, chunk offsets --
  // is then computed per call instead of being hoisted to the kernel entry, where the merged-beam kernel, which inlines
  // this function for its rare re-runs, would have to keep it alive across its own hop loop and spill it to scratch)
  asm volatile("" : "+v"(lane));
  constexpr int PU = passes<G, CU>();
  const uint8_t* const vectors = x.vectors;
  const uint32_t* const links = x.links;
  const uint32_t row_bytes = x.row_bytes;
  const int nchunks = x.nchunks, B = x.B, M = x.M, cand_slots = x.cand_slots;
  const bool tagged = x.tagged;
  const VisGeom vg = x.vg;
  uint32_t* const vis = x.vis;
  uint32_t* const stage_ids = x.stage_ids;
  uint32_t* const ovf_list = x.ovf_list;
  LdsHeap nbr{x.nbr};
  LdsHeap cand{x.cand};  // while everything fits in LDS
  uint32_t* const bitmap = cold_args()->ovf_bitmap + (uint64_t)blockIdx.x * cold_args()->bitmap_words;
  uint32_t* const ovf_glist = cold_args()->ovf_glist + (uint64_t)blockIdx.x * cold_args()->ovf_cap;
  unsigned long long* const spill = cold_args()->cand_spill + (uint64_t)blockIdx.x * cold_args()->spill_entries;

  ExactState s{1, 1, best_d, ST_OK};  // max_dist == distance(query, entry): same arithmetic, same bits
  uint32_t vis_count = 1;
  bool ovf = false;       // 32-bit table: switched to the bitmap; tagged: some id went to the bitmap
  uint32_t n_dist = 0, n_hops = 0;
  if (resumed) {
    s = rs.s;
    n_dist = rs.n_dist;
    n_hops = rs.n_hops;
    ovf = rs.ovf;
  } else {
    if (lane == 0) {
      CandHeap c0{cand.p, spill, cand_slots};
      c0.set(0, fnv_stl::Entry{-best_d, entry});
      nbr.set(0, fnv_stl::Entry{best_d, entry});
    }
    if (!tagged) {
      if (lane == 0) visited_insert_lds(vis, cold_args()->vis_slots - 1, cold_args()->vis_shift, entry);
    } else {
      visited_insert<DIRECT>(vis, vg, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    }
  }
  // (ovf is wave-uniform: the visited inserts set it for every lane)
  bool aborted = false;
  if (cand_slots == 0) __threadfence_block();
  __syncthreads();

  while (true) {
    if (s.cand_n <= 0 || s.err) break;  // (s.err: a replayed log that overflowed the candidates heap -- no hop on that state)
    const fnv_stl::Entry ctop = exact_cand_top(x, cand, spill);
    const float ctop_d = -rfl(ctop.key);
    if (ctop_d > s.max_dist && s.nbr_n >= B) break;  // Index.h:630
    const int node = rfl((int)ctop.val);
    // issue the link-row load now; the cooperative pop below hides most of its HBM latency
    uint32_t row_id = EMPTY_ID;
    if (lane < M) row_id = links[(uint64_t)(uint32_t)node * (uint32_t)M + lane];
    uint32_t stop_now = 0u;  // (read past this CU's L1: another CU writes it)
    if (stop) stop_now = __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    exact_cand_pop(x, cand, spill, s, lane, ph);
    n_hops++;
    PH_MARK(2);
    if (stop && rfl((int)stop_now) == (int)SH_ANSWERED) {  // the merged-beam pass has answered this query
      aborted = true;
      break;
    }

    for (int m0 = 0; m0 < M; m0 += WAVE) {
      const bool act = m0 + lane < M;
      uint32_t id = row_id;
      if (m0 > 0) id = act ? links[(uint64_t)(uint32_t)node * (uint32_t)M + m0 + lane] : EMPTY_ID;
      PH_MARK(3);
      bool isnew = false;
      if (tagged) {
        isnew = visited_insert<DIRECT>(vis, vg, act, id, bitmap, ovf_list, ovf_glist, ovf);
      } else {  // 32-bit open addressing ("visited_wide", tests): hands over to the bitmap at 3/4 load
        const uint32_t vis_mask = cold_args()->vis_slots - 1, vis_shift = cold_args()->vis_shift;
        if (!ovf && vis_count + WAVE > cold_args()->vis_limit) ovf = true;
        if (act) {
          if (!ovf) {
            isnew = visited_insert_lds(vis, vis_mask, vis_shift, id);
          } else if (!visited_lookup_lds(vis, vis_mask, vis_shift, id)) {
            uint32_t bit = 1u << (id & 31);
            uint32_t old = atomicOr(&bitmap[id >> 5], bit);
            isnew = !(old & bit);
          }
        }
      }
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
        batch_dists<T, METRIC, G, CU, FULL>(vectors, row_bytes, nchunks, q, cid, npass, cd, lane);
        PH_MARK(5);

        // ---- admissions in link order (Index.h:667-705).  Superset filter first: max_dist never grows
        // once the beam is full, so whatever fails here would also fail the sequential test.
#pragma unroll
        for (int pu = 0; pu < PU; pu++) {
          if (pu >= npass) break;
          unsigned long long pm = __ballot(group_leader && cval[pu] && (s.nbr_n < B || cd[pu] < s.max_dist));
          while (pm) {
            const int i = __ffsll((long long)pm) - 1;
            pm &= pm - 1;
            const float di = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cd[pu]), i));
            const uint32_t idi = (uint32_t)__builtin_amdgcn_readlane((int)cid[pu], i);
            exact_admit(x, nbr, cand, spill, s, di, idi, lane, ph);
            if (s.err) pm = 0;
          }
          if (s.err) break;
        }
        PH_MARK(6);
        if (s.err) break;
      }
      wave_sync();  // stage_ids is rewritten by the next row chunk
      if (s.err) break;
    }
    if (s.err) break;
  }
  PH_MARK(2);
  const int err = s.err;
  const int nbr_n = s.nbr_n;

  // Shadow mode: a shadow that ran out of candidate-heap room AFTER the merged-beam pass answered the query has nothing to
  // report either -- the launch-wide status word must not fail a launch whose every query was answered.  (What remains: a
  // shadow that overflows while the merged-beam pass is still searching records the error even if that pass answers later;
  // the results it wrote are then overwritten by the right ones, the status stays set -- conservative.)
  if (err && stop) {
    const uint32_t answered = __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rfl((int)answered) == (int)SH_ANSWERED) aborted = true;
  }
  if (aborted) {  // nothing to report; hand the slot's HBM bitmap back clean
    if (ovf) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, tagged, lane);
    __syncthreads();
    return;
  }
  // ---- results (Index.h:393-408): drain, std::sort by distance, truncate to K --------------
  __syncthreads();
  const int K = cold_args()->K;
  const int n = nbr_n;
  const int cnt = n < K ? n : K;
  // candidates are dead now: their array takes the result list (LDS when it holds B + 1 entries, else the spill area)
  const bool res_global = cand_slots < B + 1;
  unsigned long long* res = res_global ? spill : cand.p;
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
  if (res_global) __threadfence_block();
  __syncthreads();
  if (any_tie) {
    // Exact replay of the reference's tail: pop everything (descending), std::sort ascending.
    for (int m = n; m > 1; m--) coop_pop<true>(nbr, m, lane, ph, 7);  // leaves nbr[] ascending
    __syncthreads();
    // Construction (node ids out, the whole beam asked for): the reference hands the beam's heap itself to
    // selectNeighbors / connectNeighbors, and when it holds fewer entries than there are slots to fill
    // (Index.h:715-717) it is popped as it is -- so the caller needs the POP ORDER, not std::sort's: the list
    // goes out closest first = pop order reversed (csrc/wire.hpp reads it backwards in that case).
    const bool pop_order_out = cold_args()->labels == nullptr && n <= K;
    for (int i = lane; i < n; i += WAVE) res[i] = pop_order_out ? nbr.p[i] : nbr.p[n - 1 - i];  // pop order = descending
    if (res_global) __threadfence_block();
    __syncthreads();
    if (lane == 0 && !pop_order_out) {
      LdsHeap r{res};
      // introsort's explicit stack lives in the (now dead) visited table: >= 512 bytes = 42 frames, the
      // library's depth limit 2*lg(n) needs at most 25 for beams that fit in LDS
      fnv_stl::sort_by_key(r, n, reinterpret_cast<int*>(vis), (int)min(64u, cold_args()->vis_bytes / 12u));
    }
    if (res_global) __threadfence_block();
    __syncthreads();
  }
  {
    ColdArgs c = cold_args();
    const int32_t* labels = c->labels;  // null: construction wants node ids
    float* od_base = c->out_dist + (uint64_t)qi * K;
    int32_t* ol_base = c->out_labels + (uint64_t)qi * K;
    for (int k = lane; k < K; k += WAVE) {
      float od = std::numeric_limits<float>::infinity();
      int32_t ol = -1;
      if (k < cnt && !err) {
        fnv_stl::Entry e = unpack(res[k]);
        od = e.key;
        ol = labels ? labels[e.val] : (int32_t)e.val;
      }
      od_base[k] = od;
      ol_base[k] = ol;
    }
    if (lane == 0) {
      if (c->out_count) c->out_count[qi] = err ? 0 : cnt;
      if (c->out_ndist) c->out_ndist[qi] = n_dist;
      if (c->out_nhops) c->out_nhops[qi] = n_hops;
      if (err) {
        atomicMax(c->status, err);
        if (c->host_status) __hip_atomic_store(c->host_status, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (every writer stores the same code)
      }
    }
  }
  PH_MARK(7);
  if (ovf) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, tagged, lane);
  __syncthreads();
}

// Replays the slot's hand-over log (`records` records, see above) into the reference's two heaps and leaves in `out` the
// state from which exact_query<..., RESUME> continues.  `ovf`: whether the merged-beam pass sent ids to the HBM bitmap.  The
// visited set is left as the merged-beam pass had it when every logged hop was the reference's; else it is rebuilt for the
// hops that were.  Returns the number of hops taken from the log.
template <bool DIRECT>
__device__ __forceinline__ uint32_t replay_log(const ExactCtx& x, const unsigned long long* log, uint32_t records, uint32_t entry,
                                               float best_d, bool ovf, int lane, PhaseTimer& ph, ExactResume& out) {
  asm volatile("" : "+v"(lane));
  const int B = x.B, M = x.M;
  LdsHeap nbr{x.nbr};
  LdsHeap cand{x.cand};
  unsigned long long* const spill = cold_args()->cand_spill + (uint64_t)blockIdx.x * cold_args()->spill_entries;
  ExactState s{1, 1, best_d, ST_OK};
  if (lane == 0) {
    CandHeap c0{cand.p, spill, x.cand_slots};
    c0.set(0, fnv_stl::Entry{-best_d, entry});
    nbr.set(0, fnv_stl::Entry{best_d, entry});
  }
  if (x.cand_slots == 0) __threadfence_block();
  __syncthreads();
  uint32_t pos = 0, hops = 0, n_dist = 0;
  bool all = true;  // every logged hop was the reference's
  if (records > 0) {
    LogReader rd;
    rd.open(log, records, lane);
    while (pos < records) {
      const fnv_stl::Entry h = rd.get(pos, lane);
      const uint32_t hb = __float_as_uint(h.key);
      const uint32_t nc = hb & 0xFFu, evaluated = (hb >> 8) & 0x3FFFu;
      // the reference's loop head (Index.h:625-633)
      bool mine = s.cand_n > 0;
      if (mine) {
        const fnv_stl::Entry ctop = exact_cand_top(x, cand, spill);
        const float ctop_d = -rfl(ctop.key);
        mine = !(ctop_d > s.max_dist && s.nbr_n >= B) && (uint32_t)rfl((int)ctop.val) == h.val;
      }
      if (!mine) {  // equal keys: the reference expands another node first (or would stop) -- the log ends here
        all = false;
        break;
      }
      exact_cand_pop(x, cand, spill, s, lane, ph);
      hops++;
      n_dist += evaluated;
      for (uint32_t i = 1; i <= nc; i++) {
        const fnv_stl::Entry c = rd.get(pos + i, lane);
        exact_admit(x, nbr, cand, spill, s, c.key, c.val, lane, ph);
        if (s.err) break;
      }
      if (s.err) break;
      pos += 1 + nc;
    }
  }
  if (!all && !s.err) {
    // Rebuild the visited set for the hops that were taken: {entry} + every link of their nodes.  (The rows are L2 hits more
    // often than not -- the merged-beam pass has just read them -- and row k + 1 is requested before row k is inserted.)
    uint32_t* const vis = x.vis;
    uint32_t* const ovf_list = x.ovf_list;
    uint32_t* const bitmap = cold_args()->ovf_bitmap + (uint64_t)blockIdx.x * cold_args()->bitmap_words;
    uint32_t* const ovf_glist = cold_args()->ovf_glist + (uint64_t)blockIdx.x * cold_args()->ovf_cap;
    if (ovf) clear_spill_bitmap(bitmap, ovf_list, ovf_glist, true, lane);
    reset_visited(vis, ovf_list, true, lane);
    __syncthreads();
    ovf = false;
    const VisGeom vg = x.vg;
    auto insert = [&](bool act, uint32_t id) {
      visited_insert<DIRECT>(vis, vg, act, id, bitmap, ovf_list, ovf_glist, ovf);
    };
    insert(lane == 0, entry);
    if (hops > 0) {
      LogReader rd;
      rd.open(log, records, lane);
      uint32_t p2 = 0;
      auto row_of = [&](uint32_t node) -> uint32_t { return lane < M ? x.links[(uint64_t)node * (uint32_t)M + lane] : EMPTY_ID; };
      fnv_stl::Entry h = rd.get(0, lane);
      uint32_t row = row_of(h.val);
      for (uint32_t k = 0; k < hops; k++) {
        p2 += 1 + (__float_as_uint(h.key) & 0xFFu);
        uint32_t next_row = EMPTY_ID;
        if (k + 1 < hops) {
          h = rd.get(p2, lane);
          next_row = row_of(h.val);
        }
        insert(lane < M, row);
        row = next_row;
      }
    }
    wave_sync();
  }
  out.s = s;
  out.n_dist = n_dist;
  out.n_hops = hops;
  out.ovf = ovf;
  return hops;
}

template <typename T, int METRIC, int G, int CU, bool FULL>
__global__ __launch_bounds__(WAVE, waves_per_simd<G>(FNV_MIN_WAVES_PER_SIMD)) void beam_search_kernel(const SearchParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  // hot parameters (scalar registers for the whole launch); everything else is re-read where it is used
  ExactCtx x;
  x.vectors = p.vectors;
  x.links = p.links;
  x.row_bytes = p.row_bytes;
  x.nchunks = (int)p.nchunks;
  x.B = p.B;
  x.M = (int)p.M;
  x.cand_slots = (int)p.cand_slots;
  x.tagged = p.vis_tag16 != 0u;
  x.vg = VisGeom{p.vis_nmask, p.vis_rshift, p.vis_rmask, p.vis_mult, p.vis_w};
  uint4* const qlds = reinterpret_cast<uint4*>(smem + p.off_q);
  x.nbr = reinterpret_cast<unsigned long long*>(smem + p.off_nbr);
  x.cand = reinterpret_cast<unsigned long long*>(smem + p.off_cand);
  x.vis = reinterpret_cast<uint32_t*>(smem + p.off_vis);
  x.stage_ids = reinterpret_cast<uint32_t*>(smem + p.off_stage_ids);
  x.ovf_list = reinterpret_cast<uint32_t*>(smem + p.off_ovf);

  while (true) {
    const int qi = next_query(lane);
    if (qi < 0) break;
    PH_DECL
    Query<G, CU> q;
    stage_query<T>(q, qlds, x.vis, x.ovf_list, qi, x.tagged, lane);
    __syncthreads();
    PH_MARK(0);
    // ---- entry-point selection (Index.h:845-870): argmin over nodes 0, s, 2s, ... -----------
    float best_d;
    const uint32_t entry = entry_point<T, METRIC, G, CU, FULL>(x.vectors, x.row_bytes, x.nchunks, q, qi, lane, best_d);
    PH_MARK(1);
    exact_query<T, METRIC, G, CU, FULL>(x, q, qi, entry, best_d, lane, ph);
    PH_FLUSH;
  }
}

}  // namespace fnv_dev
