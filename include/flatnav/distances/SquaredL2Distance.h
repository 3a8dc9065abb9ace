// flatnav/distances/SquaredL2Distance.h -- squared Euclidean distance, sum (x-y)^2 without the root (own implementation).
//
// API as in the reference (include/flatnav/distances/SquaredL2Distance.h:24-72): SquaredL2Distance<DataType>::create(dim),
// DistanceInterface methods, and a serialize() that writes {dimension, data_size_bytes} -- the two
// u64 fields at byte offsets 44 and 52 of an index file.  distance() here runs on the CPU and is
// used by index construction only; search evaluates the same definition on the GPU.
#pragma once
#include <cstddef>
#include <cstring>
#include <iostream>
#include <memory>

#include <flatnav/distances/DistanceInterface.h>
#include <flatnav/util/Datatype.h>
#include <flatnav/util/HostKernels.h>

namespace flatnav::distances {

template <DataType data_type = DataType::float32>
class SquaredL2Distance : public DistanceInterface<SquaredL2Distance<data_type>> {
  friend class DistanceInterface<SquaredL2Distance<data_type>>;
  using element_t = typename flatnav::util::type_for_data_type<data_type>::type;

  std::size_t _dimension = 0;
  std::size_t _data_size_bytes = 0;

 public:
  static constexpr MetricType kMetric = MetricType::L2;

  SquaredL2Distance() = default;
  explicit SquaredL2Distance(std::size_t dim) : _dimension(dim), _data_size_bytes(dim * flatnav::util::size(data_type)) {}

  static std::unique_ptr<SquaredL2Distance<data_type>> create(std::size_t dim) {
    return std::make_unique<SquaredL2Distance<data_type>>(dim);
  }

  std::size_t getDimension() const { return _dimension; }

  float distanceImpl(const void* x, const void* y, bool /*asymmetric*/ = false) const {
    return flatnav::util::host::squaredL2(static_cast<const element_t*>(x), static_cast<const element_t*>(y), _dimension);
  }

  DataType getDataTypeImpl() const { return data_type; }

  template <typename Archive>
  void serialize(Archive& archive) {
    archive(_dimension, _data_size_bytes);
  }

 private:
  std::size_t dataSizeImpl() const { return _data_size_bytes; }
  void transformDataImpl(void* destination, const void* src) const { std::memcpy(destination, src, _data_size_bytes); }
  void getSummaryImpl() const {
    std::cout << "\nSquaredL2Distance Parameters\n-----------------------------\nDimension: " << _dimension << "\n"
              << std::flush;
  }
};

}  // namespace flatnav::distances
