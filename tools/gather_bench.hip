// Developer micro-benchmark: achievable HBM bandwidth for the search kernel's access pattern --
// random rows of `row_bytes` read as 16-byte chunks by G=8-lane groups (whole 128-byte lines), 16 loads in
// flight per lane -- with no other work.  Gives the practical ceiling to compare roofline.achieved with.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int CHUNKS_PER_LANE, int PASSES>
__global__ __launch_bounds__(64) void gather(const uint8_t* __restrict__ base, uint64_t n_rows, uint32_t row_bytes,
                                             int iters, uint32_t* out) {
  const int lane = threadIdx.x, g = lane & 7, v = lane >> 3;
  uint32_t rng = (blockIdx.x * 64 + lane / 8 * 8) * 2654435761u + 12345u;  // same per 8-lane group
  uint32_t acc = 0;
  const uint32_t nch = row_bytes / 16;
  for (int it = 0; it < iters; it++) {
    uint4 y[PASSES][CHUNKS_PER_LANE];
#pragma unroll
    for (int pu = 0; pu < PASSES; pu++) {
      rng = rng * 1664525u + 1013904223u;
      uint64_t row = ((uint64_t)(rng >> 4) * n_rows) >> 28;
      const uint8_t* p = base + row * row_bytes;
#pragma unroll
      for (int c = 0; c < CHUNKS_PER_LANE; c++) {
        uint32_t ch = (g + 8 * c) % nch;
        y[pu][c] = *reinterpret_cast<const uint4*>(p + ch * 16);
      }
    }
#pragma unroll
    for (int pu = 0; pu < PASSES; pu++)
#pragma unroll
      for (int c = 0; c < CHUNKS_PER_LANE; c++) acc ^= y[pu][c].x ^ y[pu][c].y ^ y[pu][c].z ^ y[pu][c].w;
  }
  if (acc == 0x12345678u) out[0] = acc + v;
}

template <int CPL, int PASSES>
static void run(const uint8_t* d, uint32_t* out, uint64_t table, uint32_t row_bytes, int waves_per_cu, hipEvent_t a, hipEvent_t b) {
  const int blocks = 256 * waves_per_cu;
  const int iters = (int)(400ull * 2048 / ((uint64_t)row_bytes * PASSES < 512 ? 512 : (uint64_t)row_bytes * PASSES));
  hipLaunchKernelGGL((gather<CPL, PASSES>), dim3(blocks), dim3(64), 0, 0, d, table / row_bytes, row_bytes, 20, out);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<CPL, PASSES>), dim3(blocks), dim3(64), 0, 0, d, table / row_bytes, row_bytes, iters, out);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  // algorithmic bytes: every 8-lane group reads one whole row per pass (chunk indices wrap inside short rows)
  const double gb = (double)blocks * 8 * PASSES * iters * (double)row_bytes / 1e9;
  printf("table %5.1f GB  rows %4u B  %2d loads in flight per lane  %2d waves/CU: %.2f TB/s of row bytes\n", table / 1e9,
         row_bytes, CPL * PASSES, waves_per_cu, gb / ms);
}

int main() {
  const uint64_t bytes = 32ull << 30;
  uint8_t* d; uint32_t* out;
  CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(d, 1, bytes));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (uint64_t table : {512ull << 20, 4ull << 30, 32ull << 30}) {
    for (int waves_per_cu : {8, 12, 16, 20, 24, 32}) {
      run<1, 12>(d, out, table, 128, waves_per_cu, a, b);  // uint8 x 128: one line per row, 12 rows in flight per group
      run<4, 3>(d, out, table, 400, waves_per_cu, a, b);   // float32 x 100: 400-byte rows straddle 4-5 lines
      run<4, 3>(d, out, table, 512, waves_per_cu, a, b);   // float32 x 128
      run<4, 4>(d, out, table, 512, waves_per_cu, a, b);
      run<24, 1>(d, out, table, 3072, waves_per_cu, a, b); // float32 x 768 (the search kernel: 64 lanes x 3 chunks x 4 rows)
    }
  }
  return 0;
}
