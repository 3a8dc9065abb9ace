// =============================================================================
//  ref_cereal_free.cpp -- TEST INFRASTRUCTURE.  Thin extern "C" wrapper that
//  compiles the REFERENCE's own code from where it lies under
//  /root/reference/include (never copied into this repo) into
//  oracle/_ref/libflatnav_ref.so.  Everything of the reference that does not
//  include <cereal/...> is buildable in this image, and all of it is here:
//     flatnav/distances/L2DistanceDispatcher.h   (SquaredL2Impl<float|int8|uint8>)
//     flatnav/distances/IPDistanceDispatcher.h   (InnerProductImpl<...>)
//     flatnav/util/{Macros,SimdUtils,SquaredL2SimdExtensions,InnerProductSimdExtensions}.h
//     flatnav/util/VisitedSetPool.h
//     flatnav/util/Reordering.h + GorderPriorityQueue.h   (round 6: gOrder / rcmOrder)
//     flatnav/util/Multithreading.h                        (round 6: executeInParallel)
//     flatnav/util/Datatype.h                              (round 6: ordinals / names / sizes)
//  flatnav/index/Index.h and the Distance classes include <cereal/...>, an
//  empty un-vendored submodule, so the search itself cannot be built (see
//  DESIGN.md).  Built by oracle/Makefile with the reference's own flags
//  (python-bindings/setup.py:75-84: -Ofast -ffast-math -funroll-loops + SIMD).
// =============================================================================
// Standard headers first: the reference headers rely on transitive includes
// (util/Datatype.h uses size_t / std::forward / std::string_view unqualified).
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string_view>
#include <utility>

#include <flatnav/distances/IPDistanceDispatcher.h>
#include <flatnav/distances/L2DistanceDispatcher.h>
#include <flatnav/util/Datatype.h>
#include <flatnav/util/Multithreading.h>
#include <flatnav/util/Reordering.h>
#include <flatnav/util/SimdUtils.h>
#include <flatnav/util/VisitedSetPool.h>

#include <atomic>
#include <vector>

#include <cstddef>
#include <cstdint>

using flatnav::distances::IPDistanceDispatcher;
using flatnav::distances::L2DistanceDispatcher;

extern "C" {

// Signature matches the oracle's dist_fn_t so these can be injected with
// orc_set_distance_fn().
float ref_l2_f32(const void* x, const void* y, size_t d) {
  return L2DistanceDispatcher::dispatch((const float*)x, (const float*)y, d);
}
float ref_l2_u8(const void* x, const void* y, size_t d) {
  return L2DistanceDispatcher::dispatch((const uint8_t*)x, (const uint8_t*)y, d);
}
float ref_l2_i8(const void* x, const void* y, size_t d) {
  return L2DistanceDispatcher::dispatch((const int8_t*)x, (const int8_t*)y, d);
}
float ref_ip_f32(const void* x, const void* y, size_t d) {
  return IPDistanceDispatcher::dispatch((const float*)x, (const float*)y, d);
}
float ref_ip_u8(const void* x, const void* y, size_t d) {
  return IPDistanceDispatcher::dispatch((const uint8_t*)x, (const uint8_t*)y, d);
}
float ref_ip_i8(const void* x, const void* y, size_t d) {
  return IPDistanceDispatcher::dispatch((const int8_t*)x, (const int8_t*)y, d);
}

// Scalar definitions the reference's own tests compare the SIMD kernels with
// (include/flatnav/tests/test_distances.cpp:36-179).
float ref_default_l2_f32(const void* x, const void* y, size_t d) {
  return flatnav::distances::defaultSquaredL2<float>((const float*)x, (const float*)y, d);
}
float ref_default_ip_f32(const void* x, const void* y, size_t d) {
  return flatnav::distances::defaultInnerProduct<float>((const float*)x, (const float*)y, d);
}

// Known-answer helpers of test_distances.cpp:84-100 (reduce_add == 36 / 10).
float ref_reduce_add8(const float* v) {
#if defined(USE_AVX)
  flatnav::util::simd8float32 s(v);
  return s.reduce_add();
#else
  float t = 0;
  for (int i = 0; i < 8; i++) t += v[i];
  return t;
#endif
}
float ref_reduce_add4(const float* v) {
#if defined(USE_SSE)
  flatnav::util::simd4float32 s(v);
  return s.reduce_add();
#else
  return v[0] + v[1] + v[2] + v[3];
#endif
}

int ref_has_avx512() {
#if defined(USE_AVX512)
  return platformSupportsAvx512() ? 1 : 0;
#else
  return 0;
#endif
}

// util/VisitedSetPool.h:16-50 driven through a C surface so the oracle's
// restated VisitedSet can be cross-checked (epoch wrap after 255 clears).
void* ref_vs_new(uint32_t n) { return new flatnav::util::VisitedSet(n); }
void ref_vs_free(void* p) { delete (flatnav::util::VisitedSet*)p; }
void ref_vs_clear(void* p) { ((flatnav::util::VisitedSet*)p)->clear(); }
void ref_vs_insert(void* p, uint32_t i) { ((flatnav::util::VisitedSet*)p)->insert(i); }
int ref_vs_is_visited(void* p, uint32_t i) { return ((flatnav::util::VisitedSet*)p)->isVisited(i) ? 1 : 0; }
int ref_vs_mark(void* p) { return ((flatnav::util::VisitedSet*)p)->getMark(); }

// ---- round 6: the rest of the cereal-free reference ------------------------------------------------------------------
// util/Reordering.h:27-200 (+ GorderPriorityQueue.h:14-109).  The out-degree table arrives in CSR form: the out-edges of
// node v are flat[offsets[v] .. offsets[v+1]).  out[i] = NEW id of node i (the reference's return value).
static std::vector<std::vector<uint32_t>> table_from_csr(const uint32_t* flat, const uint64_t* offsets, uint32_t n) {
  std::vector<std::vector<uint32_t>> table(n);
  for (uint32_t v = 0; v < n; v++) table[v].assign(flat + offsets[v], flat + offsets[v + 1]);
  return table;
}
void ref_gorder(const uint32_t* flat, const uint64_t* offsets, uint32_t n, int w, uint32_t* out) {
  auto table = table_from_csr(flat, offsets, n);
  std::vector<uint32_t> p = flatnav::util::gOrder<uint32_t>(table, w);
  for (uint32_t v = 0; v < n; v++) out[v] = p[v];
}
void ref_rcm(const uint32_t* flat, const uint64_t* offsets, uint32_t n, uint32_t* out) {
  auto table = table_from_csr(flat, offsets, n);
  std::vector<uint32_t> p = flatnav::util::rcmOrder<uint32_t>(table);
  for (uint32_t v = 0; v < n; v++) out[v] = p[v];
}

// util/Multithreading.h:19-48: hits[i - start] += 1 + extra for every index the loop hands out (the forwarded argument
// arrives by value in every call); returns 0, or 1 if num_threads == 0 threw std::invalid_argument.
int ref_execute_in_parallel(uint32_t start, uint32_t end, uint32_t num_threads, uint32_t extra, uint32_t* hits) {
  std::atomic<uint32_t>* cells = reinterpret_cast<std::atomic<uint32_t>*>(hits);
  try {
    flatnav::executeInParallel(
        start, end, num_threads, [&](uint32_t i, uint32_t add) { cells[i - start].fetch_add(1 + add); }, extra);
  } catch (const std::invalid_argument&) {
    return 1;
  }
  return 0;
}

// util/Datatype.h:11-118: ordinal <-> name <-> size (the ordinal is the first int32 of a saved index, Index.h:136)
const char* ref_datatype_name(int ordinal) { return flatnav::util::name(static_cast<flatnav::util::DataType>(ordinal)); }
int ref_datatype_ordinal(const char* label) { return static_cast<int>(flatnav::util::type(label)); }
uint64_t ref_datatype_size(int ordinal) { return flatnav::util::size(static_cast<flatnav::util::DataType>(ordinal)); }
uint64_t ref_datatype_enum_bytes() { return sizeof(flatnav::util::DataType); }
// type_for_data_type<> (Datatype.h:121-186): element size of the C++ type each index tag maps to
uint64_t ref_datatype_ctype_bytes(int ordinal) {
  using flatnav::util::DataType;
  switch (static_cast<DataType>(ordinal)) {
    case DataType::float32: return sizeof(flatnav::util::type_for_data_type<DataType::float32>::type);
    case DataType::int8: return sizeof(flatnav::util::type_for_data_type<DataType::int8>::type);
    case DataType::uint8: return sizeof(flatnav::util::type_for_data_type<DataType::uint8>::type);
    default: return 0;
  }
}

}  // extern "C"
