// flatnav/util/Multithreading.h -- host work-sharing loop (own implementation).
//
// Same contract as the reference's flatnav::executeInParallel
// (include/flatnav/util/Multithreading.h:19-48): indices [start, end) are handed out one at a
// time from a shared atomic counter to `num_threads` std::threads; extra arguments are
// forwarded to every call.  Used by index construction only -- batched SEARCH parallelism is
// the GPU grid (one query per wavefront slot) plus query sharding across GPUs.
#pragma once
#include <atomic>
#include <cstdint>
#include <exception>
#include <stdexcept>
#include <thread>
#include <vector>

namespace flatnav {

template <typename Function, typename... Args>
void executeInParallel(uint32_t start_index, uint32_t end_index, uint32_t num_threads, Function function,
                       Args... additional_args) {
  if (num_threads == 0) throw std::invalid_argument("Invalid number of threads");
  std::atomic<uint32_t> next(start_index);
  std::exception_ptr first_error;
  std::atomic<bool> failed(false);
  auto worker = [&] {
    for (;;) {
      const uint32_t i = next.fetch_add(1);
      if (i >= end_index || failed.load(std::memory_order_relaxed)) return;
      try {
        function(i, additional_args...);
      } catch (...) {
        if (!failed.exchange(true)) first_error = std::current_exception();
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(num_threads);
  for (uint32_t t = 0; t < num_threads; ++t) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
  if (first_error) std::rethrow_exception(first_error);
}

}  // namespace flatnav
