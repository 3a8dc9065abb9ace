// distance.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip).
// Cross-lane reductions, per-chunk distance arithmetic and the batched gather/distance primitive.
#pragma once
#include "search_params.h"
namespace fnv_dev {

// ---------------------------------------------------------------------------------------------
// Cross-lane sums over aligned groups of G lanes (DPP inside a 16-lane row, bpermute above).
// Every step adds the same two operands in both partner lanes, so all lanes of a group end with
// bit-identical sums.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
template <int G, typename A>
__device__ __forceinline__ A group_sum(A v) {
  v += dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]  (lane ^ 1)
  v += dpp_mov<0x4E>(v);                       // quad_perm [2,3,0,1]  (lane ^ 2)
  if (G >= 8) v += dpp_mov<0x141>(v);          // row_half_mirror      (i <-> 7-i)
  if (G >= 16) v += dpp_mov<0x140>(v);         // row_mirror           (i <-> 15-i)
  if (G >= 32) v += __shfl_xor(v, 16, WAVE);
  if (G >= 64) v += __shfl_xor(v, 32, WAVE);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Distance kernels on one 16-byte chunk pair.  L2 = sum (x-y)^2, IP = 1 - sum x*y
// (L2DistanceDispatcher.h:10-17, IPDistanceDispatcher.h:10-16).  Integer element types
// accumulate exactly in int32 (the reference's float/int32 accumulations agree with that while
// the sum stays below 2^24).
// ---------------------------------------------------------------------------------------------
template <typename T, int METRIC>
struct Dist;

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int METRIC>
struct Dist<float, METRIC> {
  // two partial sums per lane so subtract and multiply-add issue as packed f32 (v_pk_add_f32 / v_pk_fma_f32)
  typedef f32x2 acc_t;
  typedef int qacc_t;  // unused for float rows
  static __device__ __forceinline__ f32x2 zero() { return f32x2{0.f, 0.f}; }
  static __device__ __forceinline__ int qzero() { return 0; }
  static __device__ __forceinline__ int qchunk(int q, const uint4&) { return q; }
  static __device__ __forceinline__ f32x2 chunk(f32x2 acc, const uint4& x, const uint4& y) {
    const f32x2 x0 = {__uint_as_float(x.x), __uint_as_float(x.y)}, x1 = {__uint_as_float(x.z), __uint_as_float(x.w)};
    const f32x2 y0 = {__uint_as_float(y.x), __uint_as_float(y.y)}, y1 = {__uint_as_float(y.z), __uint_as_float(y.w)};
    if (METRIC == FNV_METRIC_L2) {
      const f32x2 t0 = x0 - y0, t1 = x1 - y1;
      acc = __builtin_elementwise_fma(t0, t0, acc);
      acc = __builtin_elementwise_fma(t1, t1, acc);
    } else {
      acc = __builtin_elementwise_fma(x0, y0, acc);
      acc = __builtin_elementwise_fma(x1, y1, acc);
    }
    return acc;
  }
  static __device__ __forceinline__ float lane_sum(f32x2 a, int) { return a.x + a.y; }
  static __device__ __forceinline__ float finish(float s) { return METRIC == FNV_METRIC_L2 ? s : 1.0f - s; }
};

// 1-byte element types: four products per instruction (v_dot4_u32_u8 / v_dot4_i32_i8), exact int32 accumulation --
// the arithmetic of the reference's AVX-512 uint8 path (SquaredL2SimdExtensions.h:32-76: widen, multiply, add into
// 32-bit lanes) and, while sums stay below 2^24, of its scalar float loops (L2DistanceDispatcher.h:10-17,
// IPDistanceDispatcher.h:79-93).  L2 is evaluated as  sum x^2 - 2 sum xy + sum y^2 : three dot products, every one
// exact, so the value is the same integer as sum (x-y)^2.  sum x^2 over a lane's chunks is shared by all PU passes.
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <typename T, int METRIC>
struct DistInt {
  static constexpr bool SIGNED = T(-1) < T(0);
  typedef i32x2 acc_t;  // {sum x*y, sum y*y}
  typedef int qacc_t;   // sum x*x
  static __device__ __forceinline__ i32x2 zero() { return i32x2{0, 0}; }
  static __device__ __forceinline__ int qzero() { return 0; }
  static __device__ __forceinline__ int dot(uint32_t a, uint32_t b, int c) {
    if (SIGNED) return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
    return (int)__builtin_amdgcn_udot4(a, b, (uint32_t)c, false);
  }
  static __device__ __forceinline__ int qchunk(int q, const uint4& x) {
    if (METRIC != FNV_METRIC_L2) return q;
    q = dot(x.x, x.x, q);
    q = dot(x.y, x.y, q);
    q = dot(x.z, x.z, q);
    return dot(x.w, x.w, q);
  }
  static __device__ __forceinline__ i32x2 chunk(i32x2 acc, const uint4& x, const uint4& y) {
    acc.x = dot(x.x, y.x, acc.x);
    acc.x = dot(x.y, y.y, acc.x);
    acc.x = dot(x.z, y.z, acc.x);
    acc.x = dot(x.w, y.w, acc.x);
    if (METRIC == FNV_METRIC_L2) {
      acc.y = dot(y.x, y.x, acc.y);
      acc.y = dot(y.y, y.y, acc.y);
      acc.y = dot(y.z, y.z, acc.y);
      acc.y = dot(y.w, y.w, acc.y);
    }
    return acc;
  }
  static __device__ __forceinline__ int lane_sum(i32x2 a, int q) {
    return METRIC == FNV_METRIC_L2 ? q + a.y - 2 * a.x : a.x;
  }
  static __device__ __forceinline__ float finish(int s) {
    return METRIC == FNV_METRIC_L2 ? (float)s : 1.0f - (float)s;
  }
};
template <int METRIC>
struct Dist<uint8_t, METRIC> : DistInt<uint8_t, METRIC> {};
template <int METRIC>
struct Dist<int8_t, METRIC> : DistInt<int8_t, METRIC> {};

// ---------------------------------------------------------------------------------------------
// Where a search keeps its query.  By default: staged in LDS, zero padded to q_chunks, read chunk by chunk inside the
// distance loop.  Rows of exactly ONE 192-chunk span (G = 64, CU = 3: 768-d float32, 3 KB) keep it in REGISTERS instead
// (round 4): lane g only ever multiplies with chunks g, g + 64, g + 128 of the query, which is 12 registers per lane and
// gives the 3 KB of LDS per resident query back -- at 768-d the LDS is what bounds the queries in flight per CU, and the
// queries in flight are what bound the bytes in flight (8 -> 10 per CU at ef = 800).  Same operands in the same order:
// no distance changes.
// ---------------------------------------------------------------------------------------------
template <int G, int CU>
constexpr bool query_in_regs() {
  return G == 64 && CU == 3;  // (the host picks this row configuration only for rows of exactly G * CU chunks)
}
// ---------------------------------------------------------------------------------------------
// SPLIT ROWS (round 6).  A 100-d float32 row is 400 bytes: three whole 128-byte lines and 16 bytes.  At a 512-byte stride
// (rounds 2-5) every gather fetched a fourth line of which 7/8 is padding -- 22 % of the HBM traffic of that configuration was
// zeros (profiles/r5_pmc_hbm_traffic.json: 1.22 x the algorithmic bytes).  Split: the MAIN table holds the whole lines of
// every row (stride 384), the last one or two chunks live in a dense side table of 16 / 32 bytes per row (1.18 M rows: 19 MB --
// resident in L2 / Infinity Cache, so the fourth request per vector no longer goes to HBM).  Chunk c of a row is still
// multiplied with chunk c of the query by the same lane as before (lane g of the vector's group: chunks g, g + 8, g + 16 from
// the main table, then chunk 24 + g from the side table), so every distance keeps its bits.  One row configuration has the
// form: G = 8, CU = 3 with FULL = false  (FULL = true is the plain 384-byte row: d = 96 float32 / 384 one-byte elements).
// The host picks it for rows of 3 lines + at most 32 bytes whose side table stays small (beam_search.hip row_layout).
// ---------------------------------------------------------------------------------------------
template <int G, int CU, bool FULL>
constexpr bool row_has_tail() {
  return G == 8 && CU == 3 && !FULL;
}

template <int G, int CU>
struct Query {
  const uint4* lds;                           // staged query; not read when the query lives in registers
  const uint8_t* tails;                       // split rows: the side table (else unused)
  uint32_t tail_chunks;                       // ... and its chunks per row
  uint4 r[query_in_regs<G, CU>() ? CU : 1];   // lane g: chunks g, g + G, ... of the query
  // registers from a query that has been staged in LDS (kernels that stage many different vectors: K0, wiring)
  __device__ __forceinline__ void from_lds(const uint4* staged, int lane, const uint8_t* tail_table = nullptr, uint32_t tail_chunks_ = 0u) {
    lds = staged;
    tails = tail_table;
    tail_chunks = tail_chunks_;
    if constexpr (query_in_regs<G, CU>()) {
#pragma unroll
      for (int cu = 0; cu < CU; cu++) r[cu] = staged[cu * G + lane % G];
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Distances from the query (in LDS, zero padded to q_chunks -- or in registers, above) to one BATCH of up to PU * (64/G) nodes.
// Lane layout: g = lane % G walks the 16-byte chunks of a row (chunk g, g+G, ...), v = lane / G picks the
// vector of a pass; pass pu holds batch slot pu*(64/G) + v.  id[pu] is per lane (equal within a G-lane group) and
// must be a legal row for EVERY lane of passes < npass: callers give lanes beyond the last real slot the id of
// the last real one, whose loads coalesce with the real ones (no extra traffic, no EXEC juggling); their
// results are simply ignored.  `npass` (wave-uniform) = number of passes that hold at least one vector.  All
// PU*CU loads of an inner iteration are issued before the first use.  Results stay in registers: every
// lane of a group ends with the group's distance in out[pu].
// ---------------------------------------------------------------------------------------------
// Rows are addressed as rows + id * row_stride: the HBM vector table in the search kernel, an LDS tile in the
// entry-scan kernel (same arithmetic and summation order in both, so their distances agree bit for bit).
template <typename T, int METRIC, int G, int CU, bool FULL>
__device__ __forceinline__ void batch_dists(const uint8_t* rows, uint32_t row_stride, int nchunks, const Query<G, CU>& q,
                                            const uint32_t (&id)[passes<G, CU>()], int npass, float (&out)[passes<G, CU>()], int lane) {
  constexpr int PU = passes<G, CU>();
  typedef Dist<T, METRIC> D;
  typedef typename D::acc_t acc_t;
  const int g = lane % G;
  acc_t acc[PU];
  typename D::qacc_t qacc = D::qzero();  // query-only term of the lane's chunks (1-byte L2: sum x^2)
  const uint8_t* rowp[PU];
#pragma unroll
  for (int pu = 0; pu < PU; pu++) {
    rowp[pu] = rows + (uint64_t)id[pu] * row_stride;
    acc[pu] = D::zero();
  }
  if constexpr (row_has_tail<G, CU, FULL>()) {
    // split rows (above): ONE span of G*CU chunks = the row's whole lines from the main table + up to G chunks from the side
    // table, all issued before the first use; lanes without a tail chunk multiply zeros (the query is zero beyond the row)
    const uint32_t tc = q.tail_chunks;
    const bool has_tail = (uint32_t)g < tc;
    uint4 y[PU][CU], yt[PU];
#pragma unroll
    for (int pu = 0; pu < PU; pu++) {
      if (pu < npass) {
#pragma unroll
        for (int cu = 0; cu < CU; cu++) y[pu][cu] = *reinterpret_cast<const uint4*>(rowp[pu] + (g + cu * G) * 16);
        yt[pu] = make_uint4(0u, 0u, 0u, 0u);
        if (has_tail) yt[pu] = *reinterpret_cast<const uint4*>(q.tails + ((uint64_t)id[pu] * tc + (uint32_t)g) * 16u);
      }
    }
#pragma unroll
    for (int cu = 0; cu < CU; cu++) {
      const uint4 x = q.lds[cu * G + g];
      qacc = D::qchunk(qacc, x);
#pragma unroll
      for (int pu = 0; pu < PU; pu++)
        if (pu < npass) acc[pu] = D::chunk(acc[pu], x, y[pu][cu]);
    }
    const uint4 xt = q.lds[G * CU + g];  // chunk 24 + g of the query: zero from the end of the row on (q_chunks = 32)
    qacc = D::qchunk(qacc, xt);
#pragma unroll
    for (int pu = 0; pu < PU; pu++)
      if (pu < npass) acc[pu] = D::chunk(acc[pu], xt, yt[pu]);
  } else if (FULL) {
    // rows are a whole number of G*CU-chunk spans (e.g. d=128 f32: 32 chunks = 8 lanes x 4): no clamping, no
    // tail select; one address per pass, the CU loads use immediate offsets
#pragma unroll
    for (int pu = 0; pu < PU; pu++) rowp[pu] += g * 16;
    for (int c0 = 0; c0 < nchunks; c0 += G * CU) {
      uint4 y[PU][CU];
#pragma unroll
      for (int pu = 0; pu < PU; pu++) {
        if (pu < npass) {  // wave-uniform: skip passes that hold no vector at all
#pragma unroll
          for (int cu = 0; cu < CU; cu++)
            y[pu][cu] = *reinterpret_cast<const uint4*>(rowp[pu] + (c0 + cu * G) * 16);
        }
      }
#pragma unroll
      for (int cu = 0; cu < CU; cu++) {
        uint4 x;
        if constexpr (query_in_regs<G, CU>()) x = q.r[cu];  // (one span: c0 == 0)
        else x = q.lds[c0 + cu * G + g];
        qacc = D::qchunk(qacc, x);
#pragma unroll
        for (int pu = 0; pu < PU; pu++)
          if (pu < npass) acc[pu] = D::chunk(acc[pu], x, y[pu][cu]);
      }
    }
  } else {
  for (int c0 = 0; c0 < nchunks; c0 += G * CU) {
    uint4 y[PU][CU];
#pragma unroll
    for (int pu = 0; pu < PU; pu++) {
      if (pu < npass) {  // wave-uniform: skip passes that hold no vector at all
#pragma unroll
        for (int cu = 0; cu < CU; cu++) {
          const int c = c0 + cu * G + g;
          const int cc = c < nchunks ? c : nchunks - 1;  // clamp: always a legal address
          y[pu][cu] = *reinterpret_cast<const uint4*>(rowp[pu] + (uint32_t)cc * 16u);
        }
      }
    }
#pragma unroll
    for (int cu = 0; cu < CU; cu++) {
      const int c = c0 + cu * G + g;
      uint4 x;  // zero beyond the row (q_chunks covers the last c0 block)
      if constexpr (query_in_regs<G, CU>()) x = q.r[cu];
      else x = q.lds[c];
      const bool in_row = c < nchunks;
      qacc = D::qchunk(qacc, x);
#pragma unroll
      for (int pu = 0; pu < PU; pu++) {
        if (pu < npass) {
          uint4 yy = y[pu][cu];
          if (!in_row) yy = x;  // x is zero there: (0-0)^2 = 0 and 0*0 = 0
          acc[pu] = D::chunk(acc[pu], x, yy);
        }
      }
    }
  }
  }
#pragma unroll
  for (int pu = 0; pu < PU; pu++) {
    out[pu] = 0.f;
    if (pu < npass) out[pu] = D::finish(group_sum<G>(D::lane_sum(acc[pu], qacc)));
  }
}

}  // namespace fnv_dev
