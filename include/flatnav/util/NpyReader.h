// flatnav/util/NpyReader.h -- minimal reader / writer for NumPy .npy files (format versions 1.0 - 3.0), enough for the
// ann-benchmarks style inputs of the command-line tools (reference: tools/construct_npy.cpp, tools/query_npy.cpp load
// 2-D float32 / int32 arrays with the third-party cnpy; this is an own, dependency-free reader).
// Supported: little-endian or single-byte dtypes ('<f4', '<i4', '<u4', '<i8', '<f8', '|u1', '|i1'), C order.
#pragma once

#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace flatnav::util {

struct NpyArray {
  std::vector<size_t> shape;
  std::string dtype;  // e.g. "<f4"
  size_t word_size = 0;
  std::vector<char> bytes;

  size_t numValues() const {
    size_t n = 1;
    for (size_t s : shape) n *= s;
    return n;
  }
  template <typename T>
  T* data() {
    if (sizeof(T) != word_size) throw std::runtime_error("npy: element size mismatch for dtype " + dtype);
    return reinterpret_cast<T*>(bytes.data());
  }
  // Values converted to T (e.g. int64 ground truth -> int32).
  template <typename T>
  std::vector<T> as() const {
    std::vector<T> out(numValues());
    const char* p = bytes.data();
    for (size_t i = 0; i < out.size(); ++i, p += word_size) {
      if (dtype == "<f4") { float v; std::memcpy(&v, p, 4); out[i] = static_cast<T>(v); }
      else if (dtype == "<f8") { double v; std::memcpy(&v, p, 8); out[i] = static_cast<T>(v); }
      else if (dtype == "<i4") { int32_t v; std::memcpy(&v, p, 4); out[i] = static_cast<T>(v); }
      else if (dtype == "<u4") { uint32_t v; std::memcpy(&v, p, 4); out[i] = static_cast<T>(v); }
      else if (dtype == "<i8") { int64_t v; std::memcpy(&v, p, 8); out[i] = static_cast<T>(v); }
      else if (dtype == "|u1") { out[i] = static_cast<T>(static_cast<uint8_t>(*p)); }
      else if (dtype == "|i1") { out[i] = static_cast<T>(static_cast<int8_t>(*p)); }
      else throw std::runtime_error("npy: unsupported dtype " + dtype);
    }
    return out;
  }
};

namespace detail {
inline std::string headerValue(const std::string& header, const std::string& key) {
  const size_t k = header.find("'" + key + "'");
  if (k == std::string::npos) throw std::runtime_error("npy: header lacks '" + key + "'");
  size_t v = header.find(':', k);
  if (v == std::string::npos) throw std::runtime_error("npy: malformed header");
  ++v;
  while (v < header.size() && header[v] == ' ') ++v;
  size_t e = v;
  if (header[v] == '(') e = header.find(')', v) + 1;
  else if (header[v] == '\'') e = header.find('\'', v + 1) + 1;
  else while (e < header.size() && header[e] != ',' && header[e] != '}') ++e;
  return header.substr(v, e - v);
}
}  // namespace detail

inline NpyArray loadNpy(const std::string& filename) {
  std::ifstream in(filename, std::ios::binary);
  if (!in.is_open()) throw std::runtime_error("Unable to open file for reading: " + filename);
  char magic[6];
  in.read(magic, 6);
  if (!in || std::memcmp(magic, "\x93NUMPY", 6) != 0) throw std::runtime_error("npy: bad magic in " + filename);
  unsigned char ver[2];
  in.read(reinterpret_cast<char*>(ver), 2);
  uint32_t header_len = 0;
  if (ver[0] == 1) {
    unsigned char b[2];
    in.read(reinterpret_cast<char*>(b), 2);
    header_len = b[0] | (b[1] << 8);
  } else {
    unsigned char b[4];
    in.read(reinterpret_cast<char*>(b), 4);
    header_len = b[0] | (b[1] << 8) | (b[2] << 16) | (static_cast<uint32_t>(b[3]) << 24);
  }
  std::string header(header_len, ' ');
  in.read(&header[0], header_len);
  if (!in) throw std::runtime_error("npy: truncated header in " + filename);
  NpyArray a;
  std::string descr = detail::headerValue(header, "descr");
  a.dtype = descr.substr(1, descr.size() - 2);
  if (a.dtype.size() < 3 || a.dtype[0] == '>') throw std::runtime_error("npy: unsupported dtype " + a.dtype);
  if (a.dtype[0] == '=') a.dtype[0] = '<';
  a.word_size = static_cast<size_t>(std::stoul(a.dtype.substr(2)));
  if (a.word_size == 1) a.dtype[0] = '|';
  if (detail::headerValue(header, "fortran_order") != "False")
    throw std::runtime_error("npy: Fortran-ordered arrays are not supported");
  std::string shape = detail::headerValue(header, "shape");
  for (size_t i = 0; i < shape.size();) {
    if (shape[i] >= '0' && shape[i] <= '9') {
      size_t j = i;
      while (j < shape.size() && shape[j] >= '0' && shape[j] <= '9') ++j;
      a.shape.push_back(static_cast<size_t>(std::stoull(shape.substr(i, j - i))));
      i = j;
    } else {
      ++i;
    }
  }
  a.bytes.resize(a.numValues() * a.word_size);
  in.read(a.bytes.data(), static_cast<std::streamsize>(a.bytes.size()));
  if (static_cast<size_t>(in.gcount()) != a.bytes.size()) throw std::runtime_error("npy: truncated data in " + filename);
  return a;
}

// Writes a C-ordered array (version 1.0 header, padded to 64 bytes like NumPy does).
inline void saveNpy(const std::string& filename, const void* data, const std::vector<size_t>& shape,
                    const std::string& dtype, size_t word_size) {
  std::string dict = "{'descr': '" + dtype + "', 'fortran_order': False, 'shape': (";
  for (size_t i = 0; i < shape.size(); ++i) dict += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
  dict += "), }";
  size_t total = 10 + dict.size() + 1;
  const size_t pad = (64 - total % 64) % 64;
  dict += std::string(pad, ' ') + "\n";
  std::ofstream out(filename, std::ios::binary);
  if (!out.is_open()) throw std::runtime_error("Unable to open file for writing: " + filename);
  out.write("\x93NUMPY\x01\x00", 8);
  const uint16_t len = static_cast<uint16_t>(dict.size());
  const unsigned char lb[2] = {static_cast<unsigned char>(len & 0xff), static_cast<unsigned char>(len >> 8)};
  out.write(reinterpret_cast<const char*>(lb), 2);
  out.write(dict.data(), static_cast<std::streamsize>(dict.size()));
  size_t n = word_size;
  for (size_t s : shape) n *= s;
  out.write(static_cast<const char*>(data), static_cast<std::streamsize>(n));
}

}  // namespace flatnav::util
