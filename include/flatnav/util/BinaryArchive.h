// flatnav/util/BinaryArchive.h -- byte-compatible stand-in for the two cereal archives the
// reference serialises its index with (cereal::BinaryOutputArchive / BinaryInputArchive,
// used at include/flatnav/index/Index.h:134-141, 449-476, 488-489 of the reference).
//
// cereal's binary archive writes every arithmetic value as its raw native-endian bytes, enums
// as their underlying type, objects by calling serialize(archive) / archive(members...), and
// cereal::binary_data(ptr, n) as n raw bytes -- no headers, no length prefixes.  That is all
// this class does, so files are interchangeable with reference-written ones:
//   int32 data_type | u64 M | u64 data_size | u64 node_size | u64 max_nodes | u64 cur_nodes |
//   u64 dimension | u64 data_size | node_size * max_nodes bytes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <type_traits>

namespace flatnav::util {

struct RawBytes {
  void* data;
  std::uint64_t size;
};
inline RawBytes binary_data(void* data, std::uint64_t size) { return RawBytes{data, size}; }

namespace detail {
template <typename T, typename Archive, typename = void>
struct has_serialize : std::false_type {};
template <typename T, typename Archive>
struct has_serialize<T, Archive, std::void_t<decltype(std::declval<T&>().serialize(std::declval<Archive&>()))>>
    : std::true_type {};
}  // namespace detail

class BinaryWriter {
  std::ostream& _out;

 public:
  explicit BinaryWriter(std::ostream& out) : _out(out) {}
  static constexpr bool is_loading = false;

  template <typename... Ts>
  BinaryWriter& operator()(Ts&&... values) {
    (put(values), ...);
    return *this;
  }

 private:
  void raw(const void* p, std::uint64_t n) {
    _out.write(static_cast<const char*>(p), static_cast<std::streamsize>(n));
    if (!_out) throw std::runtime_error("index file: write failed");
  }
  template <typename T>
  void put(T& v) {
    using U = std::remove_cv_t<std::remove_reference_t<T>>;
    if constexpr (std::is_same_v<U, RawBytes>) {
      raw(v.data, v.size);
    } else if constexpr (std::is_enum_v<U>) {
      auto u = static_cast<std::underlying_type_t<U>>(v);
      raw(&u, sizeof(u));
    } else if constexpr (std::is_arithmetic_v<U>) {
      raw(&v, sizeof(U));
    } else {
      const_cast<U&>(v).serialize(*this);
    }
  }
};

class BinaryReader {
  std::istream& _in;

 public:
  explicit BinaryReader(std::istream& in) : _in(in) {}
  static constexpr bool is_loading = true;

  template <typename... Ts>
  BinaryReader& operator()(Ts&&... values) {
    (get(values), ...);
    return *this;
  }

 private:
  void raw(void* p, std::uint64_t n) {
    _in.read(static_cast<char*>(p), static_cast<std::streamsize>(n));
    if (static_cast<std::uint64_t>(_in.gcount()) != n) throw std::runtime_error("index file: truncated");
  }
  template <typename T>
  void get(T& v) {
    using U = std::remove_cv_t<std::remove_reference_t<T>>;
    if constexpr (std::is_same_v<U, RawBytes>) {
      raw(v.data, v.size);
    } else if constexpr (std::is_enum_v<U>) {
      std::underlying_type_t<U> u;
      raw(&u, sizeof(u));
      v = static_cast<U>(u);
    } else if constexpr (std::is_arithmetic_v<U>) {
      raw(&v, sizeof(U));
    } else {
      v.serialize(*this);
    }
  }
};

}  // namespace flatnav::util
