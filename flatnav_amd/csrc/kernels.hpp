// kernels.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip).
// Entry-point scan, the beam-search kernel, the K0 batch kernel, developer micro-benchmark, AoS->SoA re-layout.
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
  uint32_t* ovf_glist = p.ovf_glist + (uint64_t)blockIdx.x * p.ovf_cap;
  const uint32_t vis_mask = p.vis_slots - 1;
  const int B = p.B;
  const int K = p.K;
  const int M = (int)p.M;

  while (true) {
    int qi = 0;
    if (lane == 0) qi = (int)atomicAdd(p.dispenser, 1u);
    qi = rfl(qi);
    if (p.redo_list) {  // second launch: only the queries the register-beam kernel handed over
      if ((uint32_t)qi >= *p.redo_count) break;
      qi = (int)p.redo_list[qi];
    } else if ((uint32_t)qi >= p.nq) {
      break;
    }
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
    if (p.vis_tag16) {
      if (p.vis_w == 16) visited_insert_tag16(vis, p, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
      else visited_insert_tagw(reinterpret_cast<unsigned long long*>(vis), p, lane == 0, entry, bitmap, ovf_list, ovf_glist, ovf);
    }
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
          if (p.vis_w == 16) isnew = visited_insert_tag16(vis, p, act, id, bitmap, ovf_list, ovf_glist, ovf);
          else isnew = visited_insert_tagw(reinterpret_cast<unsigned long long*>(vis), p, act, id, bitmap, ovf_list, ovf_glist, ovf);
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
        ol = p.labels ? p.labels[e.val] : (int32_t)e.val;  // null: construction wants node ids
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
      if (p.vis_tag16 && listed <= OVF_LIST + p.ovf_cap) {  // every id is on record: clear just their words
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

// Incremental construction: overwrite the link rows of `count` scattered nodes (same normalisation as above).
__global__ void scatter_links_kernel(const uint32_t* __restrict__ node_ids, const uint32_t* __restrict__ rows,
                                     uint64_t count, uint32_t M, uint64_t id_limit, uint32_t* __restrict__ links,
                                     int* bad_flag) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= count) return;
  const uint32_t self = node_ids[r];
  if ((uint64_t)self >= id_limit) {
    atomicExch(bad_flag, 1);
    return;
  }
  const uint32_t* in = rows + r * M;
  uint32_t* out = links + (uint64_t)self * M;
  for (uint32_t i = 0; i < M; i++) {
    uint32_t id = in[i];
    if ((uint64_t)id >= id_limit) {
      atomicExch(bad_flag, 1);
      id = self;
    }
    for (uint32_t j = 0; j < i; j++)
      if (in[j] == id) {
        id = self;
        break;
      }
    out[i] = id;
  }
}

}  // namespace fnv_dev
