// Test shim: the PRODUCT's header-only host utilities (include/flatnav/util/{Reordering,Multithreading,Datatype}.h) behind
// the same extern "C" surface oracle/ref_cereal_free.cpp gives the REFERENCE's headers, so that tests/test_reference_pins.py
// can drive both with the same inputs and compare bytes.  Built by the test with g++ (no GPU, no HIP library needed).
#include <flatnav/util/Datatype.h>
#include <flatnav/util/Multithreading.h>
#include <flatnav/util/Reordering.h>

#include <atomic>
#include <cstdint>
#include <stdexcept>
#include <vector>

static std::vector<std::vector<uint32_t>> table_from_csr(const uint32_t* flat, const uint64_t* offsets, uint32_t n) {
  std::vector<std::vector<uint32_t>> table(n);
  for (uint32_t v = 0; v < n; v++) table[v].assign(flat + offsets[v], flat + offsets[v + 1]);
  return table;
}

extern "C" {

void own_gorder(const uint32_t* flat, const uint64_t* offsets, uint32_t n, int w, uint32_t* out) {
  auto table = table_from_csr(flat, offsets, n);
  std::vector<uint32_t> p = flatnav::util::gOrder<uint32_t>(table, w);
  for (uint32_t v = 0; v < n; v++) out[v] = p[v];
}
void own_rcm(const uint32_t* flat, const uint64_t* offsets, uint32_t n, uint32_t* out) {
  auto table = table_from_csr(flat, offsets, n);
  std::vector<uint32_t> p = flatnav::util::rcmOrder<uint32_t>(table);
  for (uint32_t v = 0; v < n; v++) out[v] = p[v];
}
int own_execute_in_parallel(uint32_t start, uint32_t end, uint32_t num_threads, uint32_t extra, uint32_t* hits) {
  std::atomic<uint32_t>* cells = reinterpret_cast<std::atomic<uint32_t>*>(hits);
  try {
    flatnav::executeInParallel(
        start, end, num_threads, [&](uint32_t i, uint32_t add) { cells[i - start].fetch_add(1 + add); }, extra);
  } catch (const std::invalid_argument&) {
    return 1;
  }
  return 0;
}
const char* own_datatype_name(int ordinal) { return flatnav::util::name(static_cast<flatnav::util::DataType>(ordinal)); }
int own_datatype_ordinal(const char* label) { return static_cast<int>(flatnav::util::type(label)); }
uint64_t own_datatype_size(int ordinal) { return flatnav::util::size(static_cast<flatnav::util::DataType>(ordinal)); }
uint64_t own_datatype_enum_bytes() { return sizeof(flatnav::util::DataType); }
uint64_t own_datatype_ctype_bytes(int ordinal) {
  using flatnav::util::DataType;
  switch (static_cast<DataType>(ordinal)) {
    case DataType::float32: return sizeof(flatnav::util::type_for_data_type<DataType::float32>::type);
    case DataType::int8: return sizeof(flatnav::util::type_for_data_type<DataType::int8>::type);
    case DataType::uint8: return sizeof(flatnav::util::type_for_data_type<DataType::uint8>::type);
    default: return 0;
  }
}

}  // extern "C"
