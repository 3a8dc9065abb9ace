// =============================================================================
//  ref_distances.cpp -- TEST INFRASTRUCTURE.  Thin extern "C" wrapper that
//  compiles the REFERENCE's own distance code from where it lies under
//  /root/reference/include (never copied into this repo) into
//  oracle/_ref/libflatnav_ref.so.  Only the cereal-free part of the reference
//  is buildable in this image:
//     flatnav/distances/L2DistanceDispatcher.h   (SquaredL2Impl<float|int8|uint8>)
//     flatnav/distances/IPDistanceDispatcher.h   (InnerProductImpl<...>)
//     flatnav/util/{Macros,SimdUtils,SquaredL2SimdExtensions,InnerProductSimdExtensions}.h
//     flatnav/util/VisitedSetPool.h
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
#include <flatnav/util/SimdUtils.h>
#include <flatnav/util/VisitedSetPool.h>

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

}  // extern "C"
