// flatnav/distances/DistanceInterface.h -- CRTP distance facade of the host API (own code).
//
// Public names follow the reference (include/flatnav/distances/DistanceInterface.h:14-59):
// MetricType {L2, IP}; DistanceInterface<T>::{distance, dimension, dataSize, getSummary,
// getDataType, transformData, serialize}.  New here: metricType(), which the GPU upload needs
// because the metric is a template property in the reference and is not stored in index files.
#pragma once
#include <cstddef>

#include <flatnav/util/Datatype.h>

namespace flatnav::distances {

using flatnav::util::DataType;

enum class MetricType { L2, IP };

template <typename Derived>
class DistanceInterface {
  Derived& self() { return *static_cast<Derived*>(this); }

 public:
  // `asymmetric` distinguishes query-vs-node from node-vs-node calls for quantised metrics;
  // plain L2 / IP ignore it.
  float distance(const void* x, const void* y, bool asymmetric = false) { return self().distanceImpl(x, y, asymmetric); }
  std::size_t dimension() { return self().getDimension(); }
  std::size_t dataSize() { return self().dataSizeImpl(); }
  void getSummary() { self().getSummaryImpl(); }
  DataType getDataType() { return self().getDataTypeImpl(); }
  MetricType metricType() { return Derived::kMetric; }
  void transformData(void* destination, const void* src) { self().transformDataImpl(destination, src); }
  template <typename Archive>
  void serialize(Archive& archive) {
    self().template serialize<Archive>(archive);
  }
};

}  // namespace flatnav::distances
