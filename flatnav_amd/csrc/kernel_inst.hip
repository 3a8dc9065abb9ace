// kernel_inst.hip -- one compilation = the instantiations of ONE kernel family for ONE (element type, metric):
//   hipcc -c -DFNV_INST_T=float -DFNV_INST_TAG=f32 -DFNV_INST_METRIC=0 -DFNV_INST_MTAG=l2 -DFNV_INST_FAMILY=4 ...
// families: 0 exact two-heap kernel + entry scan, 3 wiring kernels, 4 merged beam (<= 256 entries in registers),
// 5 merged beam (<= 64 entries in registers), 6 merged beam (LDS, any width), 7 merged beam (<= 128 in registers), 8-11 the
// DIRECT forms of 4-7 (small launches on small indexes: the visited set is a bitmap in LDS).  flatnav_amd/build.py compiles the
// 60 combinations in parallel and links them with beam_search.hip.
#include <hip/hip_runtime.h>

#include "kernel_table.h"
#include "kernels.hpp"
#include "merged_beam.hpp"
#include "wire.hpp"

#define FNV_CAT_(a, b, c, d, e) a##b##c##d##e
#define FNV_CAT(a, b, c, d, e) FNV_CAT_(a, b, c, d, e)
#define FNV_FILLER(family) FNV_CAT(fill_, family, FNV_INST_TAG, _, FNV_INST_MTAG)

namespace fnv_dev {

typedef FNV_INST_T T;
constexpr int METRIC = FNV_INST_METRIC;

// ROW(slot, kernel template, extra template arguments...) fills slot[cfg][full] for the eight row configurations
#define FNV_ROW(slot, K, ...)                                          \
  slot[0][FULL] = K<T, METRIC, 8, 1, FULL __VA_ARGS__>;                \
  slot[1][FULL] = K<T, METRIC, 8, 2, FULL __VA_ARGS__>;                \
  slot[2][FULL] = K<T, METRIC, 8, 4, FULL __VA_ARGS__>;                \
  slot[3][FULL] = K<T, METRIC, 16, 4, FULL __VA_ARGS__>;               \
  slot[4][FULL] = K<T, METRIC, 32, 4, FULL __VA_ARGS__>;               \
  slot[5][FULL] = K<T, METRIC, 64, 4, FULL __VA_ARGS__>;               \
  slot[6][FULL] = K<T, METRIC, 64, 3, FULL __VA_ARGS__>;               \
  slot[7][FULL] = K<T, METRIC, 8, 3, FULL __VA_ARGS__>;

template <bool FULL>
static void fill_rows(KernelTable& t) {
#if FNV_INST_FAMILY == 0
  FNV_ROW(t.exact, beam_search_kernel)
  FNV_ROW(t.scan, entry_scan_kernel)
#elif FNV_INST_FAMILY == 4
#define FNV_COMMA_MB_R , MB_R
  FNV_ROW(t.merged, beam_search_merged_kernel, FNV_COMMA_MB_R)
#elif FNV_INST_FAMILY == 5
#define FNV_COMMA_ONE , 1
  FNV_ROW(t.merged1, beam_search_merged_kernel, FNV_COMMA_ONE)
#elif FNV_INST_FAMILY == 6
#define FNV_COMMA_ZERO , 0
  FNV_ROW(t.merged0, beam_search_merged_kernel, FNV_COMMA_ZERO)
#elif FNV_INST_FAMILY == 7
#define FNV_COMMA_TWO , 2
  FNV_ROW(t.merged2, beam_search_merged_kernel, FNV_COMMA_TWO)
#elif FNV_INST_FAMILY == 8
#define FNV_COMMA_MB_R_D , MB_R, true
  FNV_ROW(t.merged_d, beam_search_merged_kernel, FNV_COMMA_MB_R_D)
#elif FNV_INST_FAMILY == 9
#define FNV_COMMA_ONE_D , 1, true
  FNV_ROW(t.merged1_d, beam_search_merged_kernel, FNV_COMMA_ONE_D)
#elif FNV_INST_FAMILY == 10
#define FNV_COMMA_ZERO_D , 0, true
  FNV_ROW(t.merged0_d, beam_search_merged_kernel, FNV_COMMA_ZERO_D)
#elif FNV_INST_FAMILY == 11
#define FNV_COMMA_TWO_D , 2, true
  FNV_ROW(t.merged2_d, beam_search_merged_kernel, FNV_COMMA_TWO_D)
#else
  FNV_ROW(t.select, wire_select_kernel)
  FNV_ROW(t.connect, wire_connect_kernel)
#endif
}

#if FNV_INST_FAMILY == 0
void FNV_CAT(fill_exact_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#elif FNV_INST_FAMILY == 4
void FNV_CAT(fill_merged_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#elif FNV_INST_FAMILY == 5
void FNV_CAT(fill_merged1_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#elif FNV_INST_FAMILY == 6
void FNV_CAT(fill_merged0_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#elif FNV_INST_FAMILY == 7
void FNV_CAT(fill_merged2_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#elif FNV_INST_FAMILY == 8
void FNV_CAT(fill_merged_d_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#elif FNV_INST_FAMILY == 9
void FNV_CAT(fill_merged1_d_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#elif FNV_INST_FAMILY == 10
void FNV_CAT(fill_merged0_d_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#elif FNV_INST_FAMILY == 11
void FNV_CAT(fill_merged2_d_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#else
void FNV_CAT(fill_wire_, FNV_INST_TAG, _, FNV_INST_MTAG, )(KernelTable& t) {
#endif
  fill_rows<false>(t);
  fill_rows<true>(t);
}

}  // namespace fnv_dev
