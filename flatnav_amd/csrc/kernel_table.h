// kernel_table.h -- part of the gfx950 search engine: how the host code finds a kernel instantiation.
// Every kernel is a template over <element type, metric, G lanes per vector, CU loads per lane, FULL rows>; the
// instantiations are compiled in separate translation units (kernel_inst.hip, once per element type x metric x
// kernel family, in parallel -- one unit with all of them takes a quarter of an hour) and registered here.
#pragma once
#include "search_params.h"
#include "wire.hpp"

namespace fnv_dev {

typedef void (*kernel_fn)(const SearchParams);
typedef void (*wire_fn)(const WireParams);

// chunks covered per inner iteration = G*CU: 8,16,32,64,128,256 (128 B ... 4 KiB of a row), and 192 for rows of
// exactly 3 KiB (768-d float32: no clamped loads, four vectors in flight -- passes<64,3>)
struct KernelCfg {
  int G, CU;
};
// ... and 24 for rows of three whole lines: FULL = plain 384-byte rows, non-FULL = SPLIT rows (three lines in the main table +
// one or two chunks in the side table; distance.hpp row_has_tail)
constexpr KernelCfg kCfgs[] = {{8, 1}, {8, 2}, {8, 4}, {16, 4}, {32, 4}, {64, 4}, {64, 3}, {8, 3}};
constexpr int kNumCfgs = 8;
constexpr int kCfgThreeLines = 7;

// Row configuration for rows of `nchunks` 16-byte chunks: the narrowest one that covers the row in one span; longer
// rows loop over 256-chunk spans.  Rows of exactly 192 chunks (768-d float32) have their own: every lane loads exactly
// its three chunks, and the query lives in registers instead of LDS (distance.hpp, query_in_regs).
inline bool cfg_query_in_regs(int cfg) { return kCfgs[cfg].G == 64 && kCfgs[cfg].CU == 3; }
inline int pick_row_cfg(uint32_t nchunks, uint32_t tail_chunks = 0) {
  if (tail_chunks) return kCfgThreeLines;  // (the host splits rows of exactly 24 main chunks only)
  if (nchunks == 192) return 6;
  if (nchunks == 24) return kCfgThreeLines;
  for (int c = 0; c < 6; c++)
    if ((uint32_t)(kCfgs[c].G * kCfgs[c].CU) >= nchunks) return c;
  return 5;
}

// All kernels for one (element type, metric): [row configuration][FULL rows].
struct KernelTable {
  kernel_fn exact[kNumCfgs][2];        // beam_search_kernel (two heaps, libstdc++-exact)
  kernel_fn scan[kNumCfgs][2];         // entry_scan_kernel (K0)
  kernel_fn merged[kNumCfgs][2];       // beam_search_merged_kernel (beam <= 256 in registers, one merge per link row)
  kernel_fn merged1[kNumCfgs][2];      // ... its one-chunk form (beam <= 64)
  kernel_fn merged0[kNumCfgs][2];      // ... its LDS form (any beam width)
  kernel_fn merged2[kNumCfgs][2];      // ... its two-chunk form (beam <= 128)
  kernel_fn merged_d[kNumCfgs][2], merged1_d[kNumCfgs][2], merged0_d[kNumCfgs][2], merged2_d[kNumCfgs][2];  // their DIRECT forms
  wire_fn select[kNumCfgs][2];         // wire_select_kernel
  wire_fn connect[kNumCfgs][2];        // wire_connect_kernel
};

// X(element type, type tag, metric ordinal, metric tag)
#define FNV_FOR_EACH_TYPE_METRIC(X) \
  X(float, f32, 0, l2) X(float, f32, 1, ip) X(uint8_t, u8, 0, l2) X(uint8_t, u8, 1, ip) X(int8_t, i8, 0, l2) X(int8_t, i8, 1, ip)

// one filler per kernel family and (type, metric), each defined by one compilation of kernel_inst.hip
#define FNV_DECLARE_FILLERS(T, tag, M, mtag)             \
  void fill_exact_##tag##_##mtag(KernelTable& t);        \
  void fill_merged_##tag##_##mtag(KernelTable& t);       \
  void fill_merged1_##tag##_##mtag(KernelTable& t);      \
  void fill_merged0_##tag##_##mtag(KernelTable& t);      \
  void fill_merged2_##tag##_##mtag(KernelTable& t);      \
  void fill_merged_d_##tag##_##mtag(KernelTable& t);     \
  void fill_merged1_d_##tag##_##mtag(KernelTable& t);    \
  void fill_merged0_d_##tag##_##mtag(KernelTable& t);    \
  void fill_merged2_d_##tag##_##mtag(KernelTable& t);    \
  void fill_wire_##tag##_##mtag(KernelTable& t);
FNV_FOR_EACH_TYPE_METRIC(FNV_DECLARE_FILLERS)
#undef FNV_DECLARE_FILLERS

}  // namespace fnv_dev
