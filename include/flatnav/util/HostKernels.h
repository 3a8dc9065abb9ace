// flatnav/util/HostKernels.h -- CPU distance kernels of the host API (own implementation).
//
// Used ONLY by index construction (Index::add), which stays on the CPU in this round; the
// search path never calls these -- it runs on the GPU through the C ABI (flatnav_hip.h).
// Semantics follow the reference's definitions (include/flatnav/distances/
// L2DistanceDispatcher.h:10-17, IPDistanceDispatcher.h:10-16): L2 = sum (x-y)^2 (no sqrt),
// IP = 1 - sum x*y; integer elements promote to int first.  Float kernels keep 16 independent
// partial sums (the shape the compiler turns into two AVX2 / one AVX-512 accumulator) and fold
// them in a fixed tree, so results do not depend on optimisation flags as long as
// -ffp-contract=off / no -ffast-math is used; on integer-valued data any order is exact.
#pragma once
#include <cstddef>
#include <cstdint>

namespace flatnav::util::host {

inline float fold16(const float* lanes) {
  float h[8], q[4];
  for (int i = 0; i < 8; ++i) h[i] = lanes[i] + lanes[i + 8];
  for (int i = 0; i < 4; ++i) q[i] = h[i] + h[i + 4];
  return (q[0] + q[2]) + (q[1] + q[3]);
}

inline float squaredL2(const float* a, const float* b, std::size_t dim) {
  float lanes[16] = {};
  std::size_t i = 0;
  for (; i + 16 <= dim; i += 16)
    for (int l = 0; l < 16; ++l) {
      const float diff = a[i + l] - b[i + l];
      lanes[l] += diff * diff;
    }
  for (int l = 0; i < dim; ++i, ++l) {
    const float diff = a[i] - b[i];
    lanes[l] += diff * diff;
  }
  return fold16(lanes);
}

inline float innerProductDistance(const float* a, const float* b, std::size_t dim) {
  float lanes[16] = {};
  std::size_t i = 0;
  for (; i + 16 <= dim; i += 16)
    for (int l = 0; l < 16; ++l) lanes[l] += a[i + l] * b[i + l];
  for (int l = 0; i < dim; ++i, ++l) lanes[l] += a[i] * b[i];
  return 1.0f - fold16(lanes);
}

// Exact integer accumulation; equal to the reference's float / int32 accumulation whenever the
// sum stays below 2^24 (always for d <= 258 with 8-bit elements).
template <typename Int8Like>
inline float squaredL2(const Int8Like* a, const Int8Like* b, std::size_t dim) {
  std::int64_t total = 0;
  for (std::size_t i = 0; i < dim; ++i) {
    const int diff = static_cast<int>(a[i]) - static_cast<int>(b[i]);
    total += diff * diff;
  }
  return static_cast<float>(total);
}

template <typename Int8Like>
inline float innerProductDistance(const Int8Like* a, const Int8Like* b, std::size_t dim) {
  std::int64_t total = 0;
  for (std::size_t i = 0; i < dim; ++i) total += static_cast<int>(a[i]) * static_cast<int>(b[i]);
  return 1.0f - static_cast<float>(total);
}

}  // namespace flatnav::util::host
