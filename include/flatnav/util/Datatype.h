// flatnav/util/Datatype.h -- element-type tags of the host API (own implementation).
//
// Mirrors the public names of the reference's include/flatnav/util/Datatype.h:11-186 so user
// code compiles unchanged: flatnav::util::DataType, name(), type(), size(),
// type_for_data_type<>, for_each_data_type<>.  The ordinal of each enumerator is part of the
// on-disk index format (it is the first int32 of a saved index, Index.h:136 of the reference)
// and of the C ABI (FNV_DTYPE_*), so the order below must never change.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string_view>
#include <utility>

namespace flatnav::util {

enum class DataType : int {
  uint8 = 0, uint16 = 1, uint32 = 2, uint64 = 3,
  int8 = 4, int16 = 5, int32 = 6, int64 = 7,
  float16 = 8, float32 = 9, float64 = 10,
  undefined = 11
};

namespace detail {
struct DataTypeRow {
  DataType tag;
  const char* label;
  std::size_t bytes;
};
inline constexpr DataTypeRow kDataTypeTable[] = {
    {DataType::uint8, "uint8", 1},     {DataType::uint16, "uint16", 2},   {DataType::uint32, "uint32", 4},
    {DataType::uint64, "uint64", 8},   {DataType::int8, "int8", 1},       {DataType::int16, "int16", 2},
    {DataType::int32, "int32", 4},     {DataType::int64, "int64", 8},     {DataType::float16, "float16", 2},
    {DataType::float32, "float32", 4}, {DataType::float64, "float64", 8},
};
}  // namespace detail

inline constexpr const char* name(DataType t) {
  for (const auto& row : detail::kDataTypeTable)
    if (row.tag == t) return row.label;
  return "undefined";
}

inline constexpr DataType type(const std::string_view& label) {
  for (const auto& row : detail::kDataTypeTable)
    if (label == row.label) return row.tag;
  return DataType::undefined;
}

inline constexpr std::size_t size(DataType t) {
  for (const auto& row : detail::kDataTypeTable)
    if (row.tag == t) return row.bytes;
  return 0;
}

// DataType -> C++ element type, for the three element types an index can hold.
template <DataType>
struct type_for_data_type;
template <>
struct type_for_data_type<DataType::float32> { using type = float; };
template <>
struct type_for_data_type<DataType::int8> { using type = std::int8_t; };
template <>
struct type_for_data_type<DataType::uint8> { using type = std::uint8_t; };

// for_each_data_type<F, tags...>::apply(f) calls f.template operator()<tag>() for each tag.
template <typename F, DataType... tags>
struct for_each_data_type {
  static void apply(F&& f) { (f.template operator()<tags>(), ...); }
};

}  // namespace flatnav::util
