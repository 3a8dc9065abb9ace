// search_params.h -- part of the gfx950 search engine (device code; included only by beam_search.hip).
// Launch constants and the kernel parameter block shared by host and device code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits>

#include <flatnav/util/StlExact.h>
#include "../../include/flatnav_hip.h"
namespace fnv_dev {

constexpr uint32_t EMPTY_ID = 0xFFFFFFFFu;
constexpr int WAVE = 64;
// Passes (vectors per lane group) whose loads are in flight together.  3 keeps the kernel at 124-128 VGPRs = 4 waves
// per SIMD (16 per CU); 4 needs 148 VGPRs (12 per CU) and measured 2-14 % slower on every configuration tried.
#ifndef FNV_PU
#define FNV_PU 3
#endif
constexpr int PU_DEFAULT = FNV_PU;  // vector "passes" whose loads are issued back to back before any use
// Rows of a whole number of 192-chunk spans (768-d, 1536-d float32 ...) run with G = 64, CU = 3 -- every lane loads
// exactly its three chunks, no clamping -- and four vectors in flight: the same 12 loads per lane as PU_DEFAULT x 4.
// Round 3 experiment (gpurun_out/r3_run11): rows of 3 KB and more (G = 64) keep so few queries resident (LDS: 8-10 per CU at
// 768-d) that half the register file is idle; compiling those instantiations for two waves per SIMD with 6 or 8 passes
// (18-24 loads per lane in flight, what a pure gather of 3 KB rows needs to reach its ceiling) gained nothing -- 85.7 k ->
// 86.0 k / 84.6 k queries/s at 3M x 768, ef=800, and cost a resident query at ef=200: that kernel's hop is paced by the
// on-chip work between two gathers (a 13-chunk LDS merge, the visited probe), not by bytes in flight.  Defaults unchanged;
// the knobs stay for the next look.  Which vectors are in flight together never changes a distance.
#ifndef FNV_PU_64_3
#define FNV_PU_64_3 4
#endif
#ifndef FNV_PU_64_4
#define FNV_PU_64_4 3
#endif
#ifndef FNV_WAVES_G64
#define FNV_WAVES_G64 4
#endif
template <int G, int CU>
constexpr int passes() {
  return (G == 64 && CU == 3) ? FNV_PU_64_3 : (G == 64 && CU == 4) ? FNV_PU_64_4 : PU_DEFAULT;
}
constexpr int passes_of(int G, int CU) { return (G == 64 && CU == 3) ? FNV_PU_64_3 : (G == 64 && CU == 4) ? FNV_PU_64_4 : PU_DEFAULT; }
template <int G>
constexpr int waves_per_simd(int deflt) {
  return G == 64 ? FNV_WAVES_G64 : deflt;
}
#ifndef FNV_MIN_WAVES_PER_SIMD
#define FNV_MIN_WAVES_PER_SIMD 4  // __launch_bounds__ 2nd argument: register budget 512/4 = 128 per lane
#endif

constexpr int MB_R = 4;                   // merged-beam kernel: 64-entry chunks of the beam held in registers
constexpr int MB_MAX_BEAM = MB_R * WAVE;  // ... = the widest beam it serves

enum : int { ST_OK = 0, ST_CAND_OVERFLOW = 1 };
enum : uint32_t { SH_NONE = 0u, SH_ANSWERED = 1u, SH_SHADOW = 2u, SH_OWN_RERUN = 3u };  // done_flags (exact shadows, below)
constexpr int SCAN_WAVES = 4;  // entry_scan_kernel (K0): waves per workgroup ...
constexpr int SCAN_QPB = 32;   // ... and queries per workgroup
constexpr uint32_t OVF_LIST = 30;  // ids remembered for a cheap clean-up of the HBM visited bitmap
// Round 3: a STASH of full ids behind the tag table (the "stash" of cuckoo hashing): an id whose two buckets are both full
// goes there first, and only when its stash bucket is full as well to the slot's HBM bitmap (visited.hpp).  At the load
// factors the layouts run at (40-65 %) a query overflows a few dozen ids: with the stash they never leave LDS, and a
// smaller table -- more resident queries -- no longer pays a dependent HBM round trip for them.  LDS: STASH words
// after the overflow list, at [OVF_LIST + 2 ...).
constexpr uint32_t STASH = 64;

struct SearchParams {
  const uint8_t* vectors;   // [n_nodes][row_bytes]
  const uint8_t* tails;     // split rows (round 6, distance.hpp): [n_nodes][tail_chunks * 16] -- the last chunks of every row, in a
                            // dense side table, when that lets the main table hold whole 128-byte lines only; else null
  const uint32_t* links;    // [n_nodes][M]
  const int32_t* labels;    // [n_nodes]
  const uint8_t* queries;   // [nq][dim] elements, dense
  float* out_dist;          // [nq][K]
  int32_t* out_labels;      // [nq][K]
  int32_t* out_count;       // [nq] or null
  uint64_t* out_ndist;      // [nq] or null
  uint64_t* out_nhops;      // [nq] or null
  uint32_t* dispenser;      // next query id
  uint32_t* redo_count;     // merged-beam kernel: [0] queries it handed to the exact search (equal keys at a decision),
                            // [1..4] by reason, [5] of them resumed from their log, [6] hops taken from the logs,
                            // [7] hops the merged-beam passes of the resumed queries had made
  int32_t* status;          // sticky error flag for the whole launch
  int32_t* host_status;     // (round 6, zero-copy small searches) the same flag in the caller's pinned result slab, or null
  uint32_t* ovf_bitmap;     // [nslots][bitmap_words] visited-set spill (all zero between queries)
  uint32_t* ovf_glist;      // [nslots][ovf_cap] ids sent to the bitmap beyond the first OVF_LIST (big indexes only)
  unsigned long long* cand_spill;  // [nslots][spill_entries]
  const uint32_t* entry_node;  // [nq] from entry_scan_kernel (null: scan inside the search kernel)
  const float* entry_dist;     // [nq]
  uint32_t* entry_node_out;    // entry_scan_kernel outputs
  float* entry_dist_out;
  uint32_t scan_tile_rows, scan_tile_stride;  // entry_scan_kernel: LDS tile geometry
  unsigned long long* phase_cycles;  // [16] profiling build only (FNV_PHASE_TIMING), else null
  uint64_t n_nodes;
  uint32_t nq, M, dim, row_bytes, nchunks, q_chunks;  // (split rows: row_bytes / nchunks describe the main table)
  uint32_t tail_chunks;    // split rows: 16-byte chunks per row in `tails` (1 or 2), else 0
  uint32_t q_lds_bytes;    // LDS the staged query takes per slot: q_chunks * 16, or 0 when it lives in registers (distance.hpp)
  int K, B;
  uint32_t n_scan, scan_step;
  uint32_t vis_slots, vis_shift, vis_limit;
  uint32_t vis_tag16;      // 1: bucketed tag table (below; tag width vis_w), 0: 32-bit open addressing
  uint32_t vis_w;          // 16: four tags per 8-byte bucket; 21 / 32: three / two tags per 64-bit bucket;
                           // 1 (round 5, small launches on small indexes): no table -- a bitmap of all node ids, vis_bytes long
  uint32_t vis_bytes;      // LDS bytes of the table
  uint32_t vis_nmask, vis_rshift, vis_rmask;  // tag16: 2^nbits-1, t = nbits-k, 2^t-1
  uint32_t vis_mult;       // tag16: buckets = vis_mult * 2^k with vis_mult in {1, 3}
  uint32_t off_ovf;        // LDS: [0] count, [1..OVF_LIST] ids that went to the HBM bitmap, [OVF_LIST + 2 ...) the stash
  uint32_t cand_slots, spill_entries, bitmap_words, ovf_cap;
  uint32_t off_q, off_nbr, off_cand, off_vis, off_stage_ids;
  uint32_t off_stage_d;     // LDS: [WAVE + 1] distances of a link row's unvisited neighbours (merged-beam kernel; = off_nbr:
                            // the permutation buffer is idle while they are staged)
  uint32_t tail_exact;     // merged-beam kernel: the last tail_exact queries of the launch skip the sorted pass
  // Exact shadows.  Work items >= shadow_base (= the number of queries) are exact (two-heap) searches of the LAST queries of
  // the launch, most recently dispensed first: item shadow_base + k shadows query shadow_base - 1 - k; nq (the dispenser's
  // limit) = queries + shadows.  A slot only ever pulls a shadow once every query has been handed out, i.e. when it would
  // otherwise go idle, and the queries it shadows first are the ones whose merged-beam search has only just begun.  One
  // word per query, done_flags[shadow_base] (zero at launch), settles who answers it:
  //   SH_NONE -> SH_ANSWERED    its merged-beam pass finished without a tie: a shadow stops at its next hop / never starts
  //   SH_NONE -> SH_SHADOW      a shadow claimed it: a merged-beam pass that meets a tie later does NOT search it again
  //   SH_NONE -> SH_OWN_RERUN   the merged-beam pass met a tie first and searches it again itself: no shadow starts
  // Both write the same bytes when both finish.  Two uses: small launches (at most a quarter of the slots: every query has a
  // shadow from the start; round 3) and -- round 4, "tail shadows" -- the end of ANY launch: instead of sending the whole last
  // round through the slower exact kernel so that no re-run becomes a straggler, every query runs the merged-beam kernel and
  // the slots that the drain leaves idle run the exact search of the queries still under way; a tie then costs one
  // exact-search latency from the query's start, paid by a slot that had nothing else to do.  0 = off.
  uint32_t shadow_base;
  uint32_t* done_flags;
  // Round 5: the hand-over log of the merged-beam kernel (kernels.hpp): log_entries 8-byte records per slot (0: no log --
  // a query in which equal keys meet at a decision is then searched again from scratch, as in rounds 2-4)
  unsigned long long* tie_log;  // [nslots][log_entries]
  uint32_t log_entries;
};

// Broadcast of lane 0's value into a scalar register ("this value is wave-uniform").
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rfl(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Register discipline.  The parameter block is ~330 bytes = 80+ scalar registers if every field stays live, and the
// search kernels are persistent (one loop over many queries), so the compiler would keep them all live and spill.
// Fields that are needed once per query or in rare branches are therefore NOT read from the by-value copy but
// re-loaded from the kernel-argument segment at the point of use (scalar loads that hit the scalar cache);
// the empty asm makes the pointer opaque so that the loads are not hoisted back to the kernel entry.
// The block must be the kernel's first (and only) argument.
typedef const __attribute__((address_space(4))) SearchParams* ColdArgs;
__device__ __forceinline__ ColdArgs cold_args() {
  ColdArgs k = (ColdArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return k;
}

// The fields of the visited-table geometry that the per-hop probe needs (kept in scalar registers).
struct VisGeom {
  uint32_t nmask, rshift, rmask, mult, w;
};

}  // namespace fnv_dev
