// relayout.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip, which owns these
// non-template kernels): AoS -> SoA re-layout at upload, link-row scatter for incremental construction, helpers.
#pragma once
#include "distance.hpp"
#include "heaps.hpp"
namespace fnv_dev {

#if defined(FNV_PHASE_TIMING) || defined(FNV_MICROBENCH)
// Developer micro-benchmark (profiling builds only): cycles per cooperative heap operation on an
// LDS heap of `size` entries, `blocks` single-wave workgroups running concurrently.
__global__ __launch_bounds__(WAVE) void heap_microbench_kernel(int size, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  LdsHeap h{reinterpret_cast<unsigned long long*>(smem + 8)};
  PhaseTimer ph;
  ph.start();
  uint32_t rng = 12345u + blockIdx.x;
  int n = 0;
  for (int i = 0; i < size; i++) {
    rng = rng * 1664525u + 1013904223u;
    coop_push(h, n, fnv_stl::Entry{(float)(rng >> 8), (uint32_t)i}, lane, ph, 15);
    n++;
  }
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    rng = rng * 1664525u + 1013904223u;
    coop_push(h, n, fnv_stl::Entry{(float)(rng >> 8), (uint32_t)it}, lane, ph, 15);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = clock64();
  for (int it = 0; it < iters; it++) {
    coop_pop<true>(h, n + 1, lane, ph, 12);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t2 = clock64();
  // plain dependent LDS round trips for reference
  int idx = lane;
  for (int it = 0; it < iters; it++) idx = (int)(h.p[idx & 63] & 63);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t3 = clock64();
  if (lane == 0 && blockIdx.x == 0) {
    out[0] = (t1 - t0) / iters;
    out[1] = (t2 - t1) / iters;
    out[2] = (t3 - t2) / iters;
    out[3] = (unsigned long long)idx;
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// U1: AoS -> SoA re-layout of a staged block of nodes.  One thread per (node, 4-byte word) when
// everything is word aligned, else per byte.  Links: ids >= n_nodes are flagged; duplicates inside
// a row are replaced by the node's own id (== already visited, see header comment).
// ---------------------------------------------------------------------------------------------
__global__ void relayout_vectors_kernel(const uint8_t* __restrict__ aos, uint64_t node_size, uint64_t data_size,
                                        uint32_t row_bytes, uint32_t tail_bytes, uint64_t first_node, uint64_t count,
                                        uint8_t* __restrict__ vectors, uint8_t* __restrict__ tails, int word_ok) {
  // split rows (distance.hpp): bytes [0, row_bytes) of a row go to the main table, [row_bytes, row_bytes + tail_bytes) to the
  // side table; zero from data_size on in either
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t row_all = row_bytes + tail_bytes;
  if (word_ok) {
    const uint32_t wpr = row_all / 4;
    const uint64_t node = tid / wpr;
    const uint32_t b = (uint32_t)(tid % wpr) * 4u;
    if (node >= count) return;
    uint32_t val = 0;
    if ((uint64_t)b < data_size) val = *reinterpret_cast<const uint32_t*>(aos + node * node_size + b);
    uint8_t* dst = b < row_bytes ? vectors + (first_node + node) * row_bytes + b : tails + (first_node + node) * tail_bytes + (b - row_bytes);
    *reinterpret_cast<uint32_t*>(dst) = val;
  } else {
    const uint64_t node = tid / row_all;
    const uint32_t b = (uint32_t)(tid % row_all);
    if (node >= count) return;
    uint8_t* dst = b < row_bytes ? vectors + (first_node + node) * row_bytes + b : tails + (first_node + node) * tail_bytes + (b - row_bytes);
    *dst = b < data_size ? aos[node * node_size + b] : (uint8_t)0;
  }
}

__global__ void relayout_links_kernel(const uint8_t* __restrict__ aos, uint64_t node_size, uint64_t data_size,
                                      uint32_t M, uint64_t first_node, uint64_t count, uint64_t n_nodes,
                                      uint32_t* __restrict__ links, int32_t* __restrict__ labels, int* bad_flag) {
  const uint64_t node = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= count) return;
  const uint8_t* base = aos + node * node_size + data_size;
  const uint32_t self = (uint32_t)(first_node + node);
  uint32_t* out = links + (first_node + node) * M;
  for (uint32_t i = 0; i < M; i++) {
    uint32_t id;
    memcpy(&id, base + (uint64_t)i * 4, 4);
    if ((uint64_t)id >= n_nodes) {
      atomicExch(bad_flag, 1);
      id = self;
    }
    for (uint32_t j = 0; j < i; j++) {
      uint32_t prev;
      memcpy(&prev, base + (uint64_t)j * 4, 4);
      if (prev == id) {
        id = self;
        break;
      }
    }
    out[i] = id;
  }
  int32_t lab;
  memcpy(&lab, base + (uint64_t)M * 4, 4);
  labels[first_node + node] = lab;
}

// ---------------------------------------------------------------------------------------------
// Gather ceiling (measurement aid, fnv_gather_ceiling): what the search kernel's access pattern reaches on THIS index's
// vector table with no other work -- random rows, read as 16-byte chunks by G-lane groups (whole 128-byte lines),
// PU x CU loads in flight per lane, exactly the (G, CU, passes) the search kernel uses for this row width.  The rate
// it reaches is the practical bound of roofline.achieved for this (row bytes, table size): part of a small table is
// served by the 256 MiB Infinity Cache, rows that are not whole lines pay for the lines they straddle.
// ---------------------------------------------------------------------------------------------
template <int G, int CU>
__global__ __launch_bounds__(WAVE) void gather_ceiling_kernel(const uint8_t* __restrict__ base, uint64_t n_rows,
                                                              uint32_t row_bytes, int iters, uint32_t* out) {
  constexpr int PU = passes<G, CU>();
  const int lane = threadIdx.x, g = lane % G;
  uint32_t rng = (blockIdx.x * WAVE + lane / G * G) * 2654435761u + 12345u;  // same within a G-lane group
  uint32_t acc = 0;
  const int nchunks = (int)(row_bytes / 16);
  for (int it = 0; it < iters; it++) {
    const uint8_t* rowp[PU];
#pragma unroll
    for (int pu = 0; pu < PU; pu++) {
      rng = rng * 1664525u + 1013904223u;
      const uint64_t row = ((uint64_t)(rng >> 4) * n_rows) >> 28;
      rowp[pu] = base + row * row_bytes;
    }
    for (int c0 = 0; c0 < nchunks; c0 += G * CU) {
      uint4 y[PU][CU];
#pragma unroll
      for (int pu = 0; pu < PU; pu++)
#pragma unroll
        for (int cu = 0; cu < CU; cu++) {
          const int c = c0 + cu * G + g;
          y[pu][cu] = *reinterpret_cast<const uint4*>(rowp[pu] + (uint32_t)(c < nchunks ? c : nchunks - 1) * 16u);
        }
#pragma unroll
      for (int pu = 0; pu < PU; pu++)
#pragma unroll
        for (int cu = 0; cu < CU; cu++) acc ^= y[pu][cu].x ^ y[pu][cu].y ^ y[pu][cu].z ^ y[pu][cu].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

__global__ void iota_kernel(uint32_t* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

// Incremental construction: overwrite the link rows of `count` scattered nodes (same normalisation as above).
__global__ void scatter_links_kernel(const uint32_t* __restrict__ node_ids, const uint32_t* __restrict__ rows,
                                     uint64_t count, uint32_t M, uint64_t id_limit, uint32_t* __restrict__ links,
                                     int* bad_flag) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= count) return;
  const uint32_t self = node_ids[r];
  if ((uint64_t)self >= id_limit) {
    atomicExch(bad_flag, 1);
    return;
  }
  const uint32_t* in = rows + r * M;
  uint32_t* out = links + (uint64_t)self * M;
  for (uint32_t i = 0; i < M; i++) {
    uint32_t id = in[i];
    if ((uint64_t)id >= id_limit) {
      atomicExch(bad_flag, 1);
      id = self;
    }
    for (uint32_t j = 0; j < i; j++)
      if (in[j] == id) {
        id = self;
        break;
      }
    out[i] = id;
  }
}

}  // namespace fnv_dev
