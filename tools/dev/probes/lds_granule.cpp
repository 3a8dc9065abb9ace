// Developer probe: how many single-wave workgroups does one CU keep resident as a function of the dynamic LDS size?
// (round 4: the launch timeline of the uint8 index showed 18 busy slots per CU where floor(160 KiB / 7712 B) = 21 and
// hipOccupancyMaxActiveBlocksPerMultiprocessor say 21.)  Every workgroup registers on its CU (HW_REG_HW_ID / XCC_ID), waits
// until the whole grid had time to start, and the host reads the per-CU maxima.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/probes/lds_granule.cpp -o /tmp/lds_granule && /tmp/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64, 8) void probe(unsigned* live, unsigned* peak, unsigned long long hold_ticks) {
  extern __shared__ unsigned char smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;  // gfx9 HW_ID layout
  const unsigned slot = (((xcc & 0xF) * 8 + se) * 2 + sh) * 16 + cu;
  if (threadIdx.x == 0) {
    smem[0] = 1;
    const unsigned now = atomicAdd(live + slot, 1u) + 1u;
    atomicMax(peak + slot, now);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(32);
    atomicSub(live + slot, 1u);
  }
}

int main() {
  const int NSLOT = 16 * 8 * 2 * 16;
  unsigned *live, *peak;
  hipMalloc(&live, NSLOT * 4);
  hipMalloc(&peak, NSLOT * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  std::vector<unsigned> h(NSLOT);
  for (int lds : {1024, 4096, 5120, 6400, 7600, 7680, 7681, 7712, 8192, 8960, 8961, 9100, 10144, 10240, 10241, 12768, 12800, 12801, 13584, 14080, 15552, 16640, 20480, 32768}) {
    hipMemset(live, 0, NSLOT * 4);
    hipMemset(peak, 0, NSLOT * 4);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)probe, 64, lds);
    hipLaunchKernelGGL(probe, dim3(256 * 40), dim3(64), lds, 0, live, peak, 20000ull /* 200 us at 100 MHz */);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), peak, NSLOT * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned> used;
    for (unsigned v : h) if (v) used.push_back(v);
    std::sort(used.begin(), used.end());
    printf("lds %6d B: hipOccupancy %2d, floor(163840/lds) %2d, floor(163840/ceil1280) %2d | CUs seen %zu, resident per CU min %u median %u max %u\n", lds, occ,
           163840 / lds, 163840 / (((lds + 1279) / 1280) * 1280), used.size(), used.empty() ? 0 : used.front(), used.empty() ? 0 : used[used.size() / 2], used.empty() ? 0 : used.back());
  }
  return 0;
}
