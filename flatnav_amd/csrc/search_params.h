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
  uint32_t* dispenser;      // next query id (exact replay of a redo list: next list position)
  uint32_t* redo_list;      // fast kernel: queries it abandoned (equal keys at a decision); exact kernel: non-null =
                            // run exactly these
  uint32_t* redo_count;
  int32_t* status;          // sticky error flag for the whole launch
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
  uint32_t nq, M, dim, row_bytes, nchunks, q_chunks;
  int K, B;
  uint32_t n_scan, scan_step;
  uint32_t vis_slots, vis_shift, vis_limit;
  uint32_t vis_tag16;      // 1: bucketed tag table (below; tag width vis_w), 0: 32-bit open addressing
  uint32_t vis_w;          // 16: four tags per 8-byte bucket; 21 / 32: three / two tags per 64-bit bucket
  unsigned long long vis_H, vis_Lo, vis_R;  // vis_w > 16: field MSBs, the other field bits, field replication multiplier
  uint32_t vis_bytes;      // LDS bytes of the table
  uint32_t vis_nmask, vis_rshift, vis_rmask;  // tag16: 2^nbits-1, t = nbits-k, 2^t-1
  uint32_t vis_mult;       // tag16: buckets = vis_mult * 2^k with vis_mult in {1, 3}
  uint32_t off_ovf;        // LDS: [0] count, [1..OVF_LIST] ids that went to the HBM bitmap
  uint32_t cand_slots, spill_entries, bitmap_words, ovf_cap;
  uint32_t off_q, off_nbr, off_cand, off_vis, off_stage_ids;
};

}  // namespace fnv_dev
