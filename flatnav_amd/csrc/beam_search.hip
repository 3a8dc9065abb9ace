// beam_search.hip -- MI355X (gfx950 / CDNA4) implementation of flatnav's batched k-NN search
// and the C ABI declared in include/flatnav_hip.h.
//
// Reference path being replaced (paths relative to the reference repo):
//   Index::search              include/flatnav/index/Index.h:387-409
//   Index::initializeSearch    include/flatnav/index/Index.h:845-870
//   Index::beamSearch          include/flatnav/index/Index.h:606-659
//   Index::processCandidateNode include/flatnav/index/Index.h:661-707
//   distance dispatch          include/flatnav/distances/{L2,IP}DistanceDispatcher.h
//   VisitedSet                 include/flatnav/util/VisitedSetPool.h:16-50
//   batched loop               python-bindings/src/flatnav/bindings.cpp:161-228
//
// Execution model: one 64-lane wavefront = one query at a time; a
// persistent grid of query slots (as many as LDS lets stay resident) pulls query ids
// from an atomic dispenser.  Per query, in LDS: the query vector, the two binary
// heaps of the reference (moved with libstdc++'s exact algorithm, flatnav/util/StlExact.h,
// cooperatively by the 64 lanes), an exact 16-bit-tag visited set, and a 64-entry id staging area.  Per hop the wave
// loads one link row (M ids, coalesced), tests/marks all of them in the visited set
// in parallel, gathers the unvisited neighbours' vectors with 16-byte loads (G lanes
// per vector so each lane group reads whole 128-byte lines, PU*CU loads in flight per
// lane), reduces the distances across lanes with DPP, then replays the reference's
// sequential admission rule on the results.
//
// Also here: incremental construction (fnv_index_write_nodes / write_links / insert_batch: Index::add,
// include/flatnav/index/Index.h:353-378 with selectNeighbors :714-763 and connectNeighbors :765-834 as the
// wire_select / wire_connect kernels of wire.hpp) and the merged-beam search kernel (merged_beam.hpp).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/flatnav_hip.h"
#include "kernel_table.h"
#ifdef FNV_DEV_FAST_BUILD
#include "kernels.hpp"
#include "merged_beam.hpp"
#endif
#include "relayout.hpp"

using namespace fnv_dev;

namespace {

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess)                                                                               \
      return fail(FNV_ERR_NO_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e));        \
  } while (0)

size_t dtype_size(int dt) { return dt == FNV_DTYPE_FLOAT32 ? 4 : (dt == FNV_DTYPE_UINT8 || dt == FNV_DTYPE_INT8) ? 1 : 0; }

// Every entry point works on its index's device and gives the calling thread its current device back on every exit
// path: a library call must not move a torch (or any other HIP) caller's allocations to another GPU.
struct DeviceScope {
  int prev = -1;
  bool moved = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) {
      err = hipSetDevice(device);
      moved = err == hipSuccess;
    }
  }
  ~DeviceScope() {
    if (moved && prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
#define ON_DEVICE(dev)                  \
  DeviceScope device_scope_(dev);       \
  HIP_TRY(device_scope_.err)

// hipFuncAttributeMaxDynamicSharedMemorySize is global per (kernel, device) while several handles (views, replicas on
// one device) launch concurrently with different beam widths: the limit is only ever RAISED, under one process-wide
// mutex -- set(A), set(B < A), launch(A) cannot happen.  (A larger limit than a launch needs costs nothing: residency
// follows the bytes the launch asks for.)
hipError_t raise_lds_limit(const void* kern, int device, uint32_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, uint32_t> limit;
  std::lock_guard<std::mutex> lock(mu);
  uint32_t& have = limit[{device, kern}];
  if (bytes <= have) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

// ---- kernel lookup ----------------------------------------------------------------------------------------------
#ifdef FNV_DEV_FAST_BUILD
// Developer builds (-DFNV_DEV_FAST_BUILD): ONE translation unit, ONE instantiation per kernel -- element type
// FNV_DEV_T, metric FNV_DEV_METRIC, G = FNV_DEV_G, CU = FNV_DEV_CU, FULL rows -- compiles in seconds (float/8/4: 128-d f32
// rows; unsigned char/8/1: 128-d u8 rows; float/64/3 with FNV_METRIC_IP: 768-d inner product).  Every table slot points at it.
#ifndef FNV_DEV_T
#define FNV_DEV_T float
#endif
#ifndef FNV_DEV_CU
#define FNV_DEV_CU 4
#endif
#ifndef FNV_DEV_G
#define FNV_DEV_G 8
#endif
#ifndef FNV_DEV_METRIC
#define FNV_DEV_METRIC FNV_METRIC_L2
#endif
const KernelTable& kernel_table(int, int) {
  static KernelTable t = [] {
    KernelTable k;
    for (int c = 0; c < kNumCfgs; c++)
      for (int f = 0; f < 2; f++) {
        k.exact[c][f] = beam_search_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true>;
        k.scan[c][f] = entry_scan_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true>;
        k.merged[c][f] = beam_search_merged_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true, MB_R>;
        k.merged1[c][f] = beam_search_merged_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true, 1>;
        k.merged0[c][f] = beam_search_merged_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true, 0>;
        k.merged2[c][f] = beam_search_merged_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true, 2>;
        k.merged_d[c][f] = beam_search_merged_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true, MB_R, true>;
        k.merged1_d[c][f] = beam_search_merged_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true, 1, true>;
        k.merged0_d[c][f] = beam_search_merged_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true, 0, true>;
        k.merged2_d[c][f] = beam_search_merged_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true, 2, true>;
        k.select[c][f] = wire_select_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true>;
        k.connect[c][f] = wire_connect_kernel<FNV_DEV_T, FNV_DEV_METRIC, FNV_DEV_G, FNV_DEV_CU, true>;
      }
    return k;
  }();
  return t;
}
#else
// Product builds: the instantiations live in kernel_inst.hip objects (one per family x element type x metric).
const KernelTable& kernel_table(int dtype, int metric) {
  static KernelTable tables[6];
  static std::once_flag once;
  std::call_once(once, [] {
    int i = 0;
#define FNV_FILL(T, tag, M, mtag)              \
    fill_exact_##tag##_##mtag(tables[i]);       \
    fill_merged_##tag##_##mtag(tables[i]);      \
    fill_merged1_##tag##_##mtag(tables[i]);     \
    fill_merged0_##tag##_##mtag(tables[i]);     \
    fill_merged2_##tag##_##mtag(tables[i]);     \
    fill_merged_d_##tag##_##mtag(tables[i]);    \
    fill_merged1_d_##tag##_##mtag(tables[i]);   \
    fill_merged0_d_##tag##_##mtag(tables[i]);   \
    fill_merged2_d_##tag##_##mtag(tables[i]);   \
    fill_wire_##tag##_##mtag(tables[i]);        \
    i++;
    FNV_FOR_EACH_TYPE_METRIC(FNV_FILL)
#undef FNV_FILL
  });
  const int t = dtype == FNV_DTYPE_FLOAT32 ? 0 : dtype == FNV_DTYPE_UINT8 ? 1 : 2;  // order of FNV_FOR_EACH_TYPE_METRIC
  return tables[2 * t + (metric == FNV_METRIC_IP ? 1 : 0)];
}
#endif

kernel_fn pick_kernel(int dtype, int metric, int cfg, bool full) { return kernel_table(dtype, metric).exact[cfg][full]; }
kernel_fn pick_scan_kernel(int dtype, int metric, int cfg, bool full) { return kernel_table(dtype, metric).scan[cfg][full]; }
kernel_fn pick_sorted_kernel(int dtype, int metric, int cfg, bool full, bool lds, int B, bool direct = false) {
  const KernelTable& t = kernel_table(dtype, metric);
  if (direct) return lds ? t.merged0_d[cfg][full] : B <= WAVE ? t.merged1_d[cfg][full] : B <= 2 * WAVE ? t.merged2_d[cfg][full] : t.merged_d[cfg][full];
  return lds ? t.merged0[cfg][full] : B <= WAVE ? t.merged1[cfg][full] : B <= 2 * WAVE ? t.merged2[cfg][full] : t.merged[cfg][full];
}
wire_fn pick_wire_kernel(int dtype, int metric, int cfg, bool full) { return kernel_table(dtype, metric).select[cfg][full]; }
wire_fn pick_connect_kernel(int dtype, int metric, int cfg, bool full) { return kernel_table(dtype, metric).connect[cfg][full]; }

}  // namespace

// What a search launch looks like for one (beam width, K) on one index: cached, because working it out costs
// several occupancy queries and a single-query search should not pay for them every time.
struct LaunchPlan {
  bool valid = false;
  int B = 0, K = 0, cfg = 0, mode = 0;
  bool full = false;
  uint64_t capacity = 0, options_version = 0;
  kernel_fn kern = nullptr, skern = nullptr;  // exact two-heap kernel; merged-beam kernel (mode != 0)
  kernel_fn skern_direct = nullptr;           // ... its DIRECT form (small launches on small indexes)
  SearchParams heaps, sorted;                 // geometry + LDS layout for each (per-call fields unset)
  uint32_t lds = 0, slds = 0;
  int bpc = 0, sbpc = 0;
};

// A host-buffer search that went through the pinned staging buffer: what search_host_finish copies where.
struct PinnedCall {
  bool active = false;
  const uint8_t* slab = nullptr;
  const int32_t* status = nullptr;
  uint64_t nq = 0;
  int K = 0;
  size_t o_lab = 0, o_cnt = 0, o_nd = 0, o_nh = 0;
  float* out_dist = nullptr;
  int32_t* out_labels = nullptr;
  int32_t* out_count = nullptr;
  uint64_t *out_ndist = nullptr, *out_nhops = nullptr;
};

// Everything fnv_set_option can change: one block, so that views and replicas start as exact copies of their source.
struct IndexOptions {
  int64_t visited_factor = 27, visited_slots = 0, visited_floor = 2048, occupancy_target = 13, occupancy_roomy = 9, cand_factor = 2,
          cand_slots = 0, spill_entries = 16384, blocks_per_cu = 0, visited_wide = 0,
          entry_kernel = 0, output_node_ids = 0, visited_tag_bits = 0, sorted_beam = 2,
          sorted_beam_min = 1, sorted_cand_lds = 2, sorted_tail_exact_pct = -1, beam_registers = 1,
          sorted_variant = -1, tune_layout = 1, shadow_exact = 1, tie_replay = 1, tie_log_entries = 0, visited_direct = 1,
          host_zero_copy = 1 << 20;  // (every call that fits the pinned staging buffer)
  int64_t overflow_list = -1;  // -1: automatic (a list in HBM only when the bitmap is larger than 512 KB)
};

// Kernel variants of one launch: 0 the exact two-heap kernel, 1 the merged-beam kernel, 2-5 the merged-beam kernel with
// the last 50 / 75 / 100 / 25 % of a round of queries sent straight to the exact search, 6 (round 4) the merged-beam kernel
// for every query plus exact shadows of the last ones on the slots the drain leaves idle (search_params.h).
constexpr int kNumVariants = 7;
constexpr int kVariantTailShadows = 6;
static const int kTailPct[kNumVariants] = {0, 0, 50, 75, 100, 25, 0};
static inline bool variant_allowed(int v, bool multi_round, bool try_tail, bool shadows_on, bool pinned_only = false) {
  if (v < 2) return true;
  // (tail shadows: measured in round 4 -- 0.5-3 % better than the merged-beam kernel alone, behind the best exact tail on every configuration:
  //  a shadow can only start when a slot falls idle, which is too late for the ties that end a launch -- so the variant can
  //  be pinned for A/B runs but is not part of the adaptive choice)
  if (v == kVariantTailShadows) return shadows_on && pinned_only;
  return multi_round && try_tail;  // an exact tail needs more than one round of queries
}

struct fnv_index_s : IndexOptions {
  bool owns_buffers = true;  // false: a view (fnv_index_view) of another handle's vectors / links / labels
  fnv_index_s* parent = nullptr;  // a view's source: its live node count is read at every launch
  std::atomic<int> n_views{0};    // live views of this handle's buffers (it cannot be freed before them)
  int device = 0;
  int dtype = FNV_DTYPE_FLOAT32, metric = FNV_METRIC_L2;
  uint32_t M = 0, dim = 0, row_bytes = 0;
  uint32_t tail_bytes = 0;  // split rows (distance.hpp, row_layout below): bytes per row in the side table that follows the main
                            // table in d_vectors' allocation ([capacity][row_bytes] main, then [capacity][tail_bytes]); 0: one table
  const uint8_t* tails() const { return tail_bytes ? d_vectors + capacity * (uint64_t)row_bytes : nullptr; }
  uint64_t vector_bytes() const { return capacity * ((uint64_t)row_bytes + tail_bytes); }
  std::atomic<uint64_t> n_nodes{0};  // live nodes: what a search sees (entry scan, id range); a view reads its source's at
                                     // every launch while the source may be growing -> atomic
  uint64_t capacity = 0;  // rows the device buffers hold (>= n_nodes; grows never)
  uint8_t* d_vectors = nullptr;
  uint32_t* d_links = nullptr;
  int32_t* d_labels = nullptr;
  int num_cus = 0;
  std::string gcn_arch;  // hipDeviceProp_t::gcnArchName: replicas on the same GPU model inherit the source's measurements
  uint64_t options_version = 0;
  LaunchPlan plan;
  // adaptive kernel choice ("sorted_beam" = 2): per beam width, the best time per query seen for each variant
  struct Tuner {
    // ms per query: [0] two-heap kernel, [1] merged-beam kernel, [2..5] merged-beam kernel whose last 50 / 75 / 100 / 25 %
    // of a round of queries go straight to the exact search ("sorted_tail_exact_pct"; launches of more than one round)
    float best[kNumVariants] = {-1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f};
    int samples[kNumVariants] = {0, 0, 0, 0, 0, 0, 0};
  };
  // Per beam width: the LDS layout fnv_tune measured to be the fastest (absent: the rules of configure_launch).
  struct LayoutChoice {
    int cand_lds = -1;       // where the exact search keeps its candidates heap: -1 = by rule, 0 = HBM, 1 = LDS
    uint32_t vis_slots = 0;  // visited-table slots: 0 = by rule
  };
  std::map<int, LayoutChoice> layouts;
  std::map<int, Tuner> tuner;
  int sample_B = 0, sample_kernel = -1;  // the launch between ev0 / ev1 is a sample for this entry (-1: it is not)
  uint64_t sample_nq = 0;
  int last_variant = 0;          // what the most recent launch ran: 0 two-heap kernel, 1 merged beam, 2-4 with exact tail
  bool last_exploratory = false;  // ... and whether the adaptive choice was still sampling (not its final pick)
  bool last_shadow = false;       // ... and whether every query had an exact shadow (small launches)
  // host-buffer searches: steady-clock time of launch / of completion (atomic: every lane's caller reports into the handle)
  std::atomic<uint64_t> t_enqueue_ns{0}, t_complete_ns{0};
  // workspace (grown on demand)
  uint32_t* d_dispenser = nullptr;  // [0] dispenser, [1] status, [3] queries a merged-beam launch handed to the exact search, [4..7] by
                                    // reason, [8] of them resumed from their log, [9] hops taken from logs, [10] hops of those queries
  unsigned long long* d_phase = nullptr;  // profiling builds only
  void* d_entry = nullptr;  // [nq] uint32 entry nodes | [nq] float entry distances (K0 output)
  size_t entry_bytes = 0;
  uint32_t* d_bitmap = nullptr;
  size_t bitmap_bytes = 0;
  void* d_wirebuf = nullptr;  // fnv_index_insert_batch: [count*keep] x {req_target, req_index, sorted_target, sorted_req} | sort scratch
  size_t wirebuf_bytes = 0;
  uint32_t* d_ovf = nullptr;  // [nslots][ovf_cap] ids whose bitmap words need clearing (big indexes)
  size_t ovf_bytes = 0;
  void* d_nodestage = nullptr;  // write_nodes: AoS staging chunk + bad flag
  size_t nodestage_bytes = 0;
  void* d_linkstage = nullptr;  // fnv_index_write_links: [count] ids | [count][M] rows | bad flag
  size_t linkstage_bytes = 0;
  unsigned long long* d_spill = nullptr;
  size_t spill_bytes = 0;
  uint32_t* d_done = nullptr;  // shadow mode: [nq] "answered" flags
  size_t done_bytes = 0;
  unsigned long long* d_tielog = nullptr;  // merged-beam kernel: [nslots][log_entries] hand-over log (round 5)
  size_t tielog_bytes = 0;
  // staging for the host-buffer entry point
  void* h_pin = nullptr;  // 1 MB of pinned host memory: staging of small host-buffer searches
  void* h_res = nullptr;  // pinned host memory for the result slab of larger host-buffer searches (grown on demand)
  size_t h_res_bytes = 0;
  PinnedCall pin;
  void* d_q = nullptr;
  size_t d_q_bytes = 0;
  void* d_out = nullptr;
  size_t d_out_bytes = 0;
  hipStream_t stream = nullptr;  // owned, used by fnv_search_batch
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t last_stream = nullptr;
  bool launched = false;
  uint64_t geom[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::mutex mu;       // launch configuration + workspace growth
  std::mutex host_mu;  // the host-buffer entry point owns d_q / d_out / stream for the whole call
  // Round 4: a SECOND LANE for concurrent host-buffer callers.  Two threads calling fnv_search_batch on one handle used to
  // take turns; now the second caller runs on a hidden view of the handle (own stream, workspace and staging, the same HBM
  // buffers), so its copies and its launch overlap the first caller's -- the reference's search is callable from several
  // threads at once (bindings.cpp:198-211 runs it under executeInParallel), and two launches in flight are what hides a
  // launch's ramp and drain.  Created on first contention; freed with the handle.
  // Round 5: EVERY lane is admitted by the HBM its launch workspace would really take (lane_workspace_bytes: slots x (visited
  // bitmap + overflow list + candidate spill area) for this batch size and the plan's residency) against a budget for all
  // hidden lanes together -- an eighth of the device's memory (36 GB of 288): one full-grid lane at 50M nodes (19 GB), three
  // at 10M, all seven at 1M; 1024-query batches at 50M nodes cost 6.4 GB each, so five of those.  A lane that does not fit
  // is not used (the caller waits for the handle, as before lanes existed); an idle lane's workspace is released when the
  // budget or the owner's own allocation needs the room (release_idle_lanes).
  static constexpr int kMaxLanes = 8;
  bool is_lane = false;          // a hidden lane: never samples or explores (it copies the owner's measurements)
  std::atomic<uint64_t> explored_launches{0};  // launches that were exploratory samples of the adaptive choice
  std::atomic<size_t> ws_bytes{0};  // launch workspace this handle holds (visited bitmaps + overflow lists + spill areas)
  size_t lane_budget_bytes = 0;  // owner: budget for the hidden lanes' workspaces together (set at creation)
  std::atomic<fnv_index_s*> lanes[kMaxLanes] = {};  // [0] unused; written under lane_mu, read anywhere
  std::mutex lane_mu;            // creation of lanes
  std::atomic<fnv_index_s*> last_served{nullptr};  // the lane that ran the handle's most recent launch (null: the handle itself):
                                                   // fnv_last_kernel_ms / _replayed_queries / _handover_stats read ITS events and counters
  uint64_t tune_epoch = 0;       // bumped whenever tuner / layouts change: a lane copies them when its own epoch lags
  uint64_t lane_epoch = ~0ull, lane_options = ~0ull;  // (on a lane: what it last copied from its owner)
  // (on a replica, round 6) the handle fnv_replica_refresh last copied options from, that handle's options_version then, and
  // the tune_epoch whose measurements this replica holds: fnv_search_batch_multi hands newer measurements over (pointer
  // compared, never followed, outside calls that were given the source itself)
  const fnv_index_s* replica_of = nullptr;
  uint64_t replica_options = ~0ull, replica_epoch = ~0ull;
};

namespace {

int index_common_init(fnv_index_s* ix) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, ix->device));
  ix->num_cus = prop.multiProcessorCount;
  ix->gcn_arch = prop.gcnArchName;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 0;
    (void)hipGetLastError();
    ix->lane_budget_bytes = total_b / 8;
    if (const char* env = getenv("FLATNAV_LANE_BUDGET_MB")) ix->lane_budget_bytes = (size_t)strtoull(env, nullptr, 10) << 20;
  }
  HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&ix->ev0));
  HIP_TRY(hipEventCreate(&ix->ev1));
  HIP_TRY(hipMalloc(&ix->d_dispenser, 16 * sizeof(uint32_t)));
  HIP_TRY(hipMemset(ix->d_dispenser, 0, 16 * sizeof(uint32_t)));
#ifdef FNV_PHASE_TIMING
  HIP_TRY(hipMalloc(&ix->d_phase, NPHASE * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(ix->d_phase, 0, NPHASE * sizeof(unsigned long long)));
#endif
  return FNV_OK;
}

int validate_geometry(uint32_t M, uint64_t n_nodes, int data_type, int metric, uint32_t dim) {
  if (dtype_size(data_type) == 0) return fail(FNV_ERR_RUNTIME, "Unsupported data type");
  if (metric != FNV_METRIC_L2 && metric != FNV_METRIC_IP) return fail(FNV_ERR_INVALID, "Invalid metric");
  if (M == 0 || dim == 0) return fail(FNV_ERR_INVALID, "M and dim must be positive");
  if (n_nodes == 0) return fail(FNV_ERR_INVALID, "cannot upload an empty index");
  if (n_nodes >= 0xFFFFFFFFull) return fail(FNV_ERR_INVALID, "too many nodes for 32-bit node ids");
  return FNV_OK;
}

int alloc_buffers(fnv_index_s* ix) {  // the caller is on ix->device
  HIP_TRY(hipMalloc(&ix->d_vectors, ix->vector_bytes()));
  HIP_TRY(hipMalloc(&ix->d_links, ix->capacity * (uint64_t)ix->M * 4));
  HIP_TRY(hipMalloc(&ix->d_labels, ix->capacity * 4));
  return index_common_init(ix);
}

// Row stride of the vector table.  Rows are 16-byte chunks; when rounding the stride up to whole 128-byte lines costs
// at most FLATNAV_ROW_PAD_PCT (default 30) per cent of padding it is done: a 100-d float32 row (400 bytes) at a
// 16-byte stride straddles 4-5 lines (4.0 on average = the 512 bytes the padded row occupies anyway), takes the clamped
// non-FULL distance path and costs the gather ~20 % of its rate (tools/gather_bench.hip: 5.9 vs 7.1 TB/s of row bytes);
// at a 512-byte stride it is exactly four lines and whole 8-lane x 4-chunk spans.  The padding is zero in rows and in
// the staged query, so every distance keeps its bits (zeros add nothing to either partial sum).
uint32_t row_stride_bytes(uint32_t dim, int data_type) {
  const uint64_t rb16 = ((uint64_t)dim * dtype_size(data_type) + 15) / 16 * 16;
  const uint64_t rb128 = (rb16 + 127) / 128 * 128;
  long pct = 30;
  if (const char* env = getenv("FLATNAV_ROW_PAD_PCT")) pct = strtol(env, nullptr, 10);
  if (pct > 0 && (rb128 - rb16) * 100 <= (uint64_t)pct * rb16) return (uint32_t)rb128;
  return (uint32_t)rb16;
}

// SPLIT ROWS (round 6, distance.hpp): a row of exactly three 128-byte lines plus at most 32 bytes (d = 97 ... 104 float32, 385 ... 416
// one-byte elements) keeps its whole lines in the main table (stride 384) and its last one or two chunks in a dense side
// table -- as long as that table stays small enough to live in L2 / Infinity Cache (FLATNAV_SPLIT_TAIL_MAX_MB, default 64 MB:
// 4 M rows of 16 bytes), where the fourth request of a gather no longer costs an HBM line that is 7/8 padding.
// FLATNAV_SPLIT_ROWS=0 turns it off (rows are then padded to four lines, as in rounds 2-5).  Every handle on the same
// buffers (views, fnv_index_adopt, replicas, the ranks of a broadcast) derives the same layout from (dim, type, capacity).
struct RowLayout {
  uint32_t row_bytes, tail_bytes;
};
RowLayout row_layout(uint32_t dim, int data_type, uint64_t capacity) {
  const uint64_t rb16 = ((uint64_t)dim * dtype_size(data_type) + 15) / 16 * 16;
  const uint64_t rem = rb16 % 128;
  long on = 1, max_mb = 64;
  if (const char* env = getenv("FLATNAV_SPLIT_ROWS")) on = strtol(env, nullptr, 10);
  if (const char* env = getenv("FLATNAV_SPLIT_TAIL_MAX_MB")) max_mb = strtol(env, nullptr, 10);
  if (on && rb16 - rem == 384 && rem > 0 && rem <= 32 && capacity * rem <= ((uint64_t)max_mb << 20))
    return RowLayout{384u, (uint32_t)rem};
  return RowLayout{row_stride_bytes(dim, data_type), 0u};
}

uint32_t pow2_ceil(uint64_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Copy AoS node records [data][M links][label] (reference Index.h:61-63) for nodes first..first+count-1 into the
// SoA device buffers, 256 MB at a time.  Link ids >= id_limit are flagged (and replaced by a self-loop).
int write_nodes_impl(fnv_index_s* ix, uint64_t first_node, uint64_t count_nodes, const void* aos_rows,
                     uint64_t node_size, uint64_t data_size, uint64_t id_limit, int* bad_out) {
  ON_DEVICE(ix->device);
  const uint64_t chunk_nodes = std::max<uint64_t>(1, (256ull << 20) / node_size);
  // staging area: [chunk of AoS records][bad flag]; kept on the index (a device build writes dozens of batches)
  const size_t need = std::min(chunk_nodes, count_nodes) * node_size + 16;
  if (need > ix->nodestage_bytes) {
    if (ix->d_nodestage) HIP_TRY(hipFree(ix->d_nodestage));
    ix->d_nodestage = nullptr;
    ix->nodestage_bytes = 0;
    HIP_TRY(hipMalloc(&ix->d_nodestage, need));
    ix->nodestage_bytes = need;
  }
  uint8_t* d_stage = (uint8_t*)ix->d_nodestage;
  int* d_bad = (int*)(d_stage + (need - 16));
  auto cleanup = [&]() {};
#define UP_TRY(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      cleanup();                                                                                       \
      return fail(FNV_ERR_NO_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e));       \
    }                                                                                                  \
  } while (0)
  UP_TRY(hipMemset(d_bad, 0, sizeof(int)));
  const int word_ok = (node_size % 4 == 0 && data_size % 4 == 0) ? 1 : 0;
  for (uint64_t done = 0; done < count_nodes; done += chunk_nodes) {
    const uint64_t count = std::min(chunk_nodes, count_nodes - done);
    const uint64_t first = first_node + done;
    UP_TRY(hipMemcpy(d_stage, (const uint8_t*)aos_rows + done * node_size, count * node_size, hipMemcpyHostToDevice));
    const uint32_t row_all = ix->row_bytes + ix->tail_bytes;  // bytes of a row over both tables
    const uint64_t units = count * (word_ok ? row_all / 4 : row_all);
    hipLaunchKernelGGL(relayout_vectors_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, 0, d_stage,
                       node_size, data_size, ix->row_bytes, ix->tail_bytes, first, count, ix->d_vectors,
                       const_cast<uint8_t*>(ix->tails()), word_ok);
    hipLaunchKernelGGL(relayout_links_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, d_stage,
                       node_size, data_size, ix->M, first, count, id_limit, ix->d_links, ix->d_labels, d_bad);
    UP_TRY(hipGetLastError());
    UP_TRY(hipDeviceSynchronize());
  }
  UP_TRY(hipMemcpy(bad_out, d_bad, sizeof(int), hipMemcpyDeviceToHost));
  cleanup();
#undef UP_TRY
  return FNV_OK;
}

}  // namespace

// Only the C ABI is exported (the library is built with -fvisibility=hidden).
#pragma GCC visibility push(default)
extern "C" {

const char* fnv_last_error(void) { return g_err.c_str(); }
const char* fnv_version(void) { return "flatnav_hip gfx950 r6"; }

int fnv_device_count(int* count) {
  if (!count) return fail(FNV_ERR_INVALID, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(FNV_ERR_NO_DEVICE, std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
  }
  *count = n;
  return FNV_OK;
}

int fnv_row_layout(uint32_t dim, int data_type, uint64_t capacity, uint32_t* row_bytes, uint32_t* tail_bytes) {
  if (!row_bytes || !tail_bytes) return fail(FNV_ERR_INVALID, "null argument");
  if (dtype_size(data_type) == 0) return fail(FNV_ERR_RUNTIME, "Unsupported data type");
  if (dim == 0 || capacity == 0) return fail(FNV_ERR_INVALID, "dim and capacity must be positive");
  const RowLayout lay = row_layout(dim, data_type, capacity);
  *row_bytes = lay.row_bytes;
  *tail_bytes = lay.tail_bytes;
  return FNV_OK;
}

int fnv_index_alloc(uint32_t M, uint64_t n_nodes, int data_type, int metric, uint32_t dim, int device,
                    fnv_index_t* out) {
  if (!out) return fail(FNV_ERR_INVALID, "out is null");
  int rc = validate_geometry(M, n_nodes, data_type, metric, dim);
  if (rc) return rc;
  ON_DEVICE(device);
  fnv_index_s* ix = new fnv_index_s();
  ix->device = device;
  ix->dtype = data_type;
  ix->metric = metric;
  ix->M = M;
  ix->dim = dim;
  ix->n_nodes = n_nodes;
  ix->capacity = n_nodes;
  const RowLayout lay = row_layout(dim, data_type, n_nodes);
  ix->row_bytes = lay.row_bytes;
  ix->tail_bytes = lay.tail_bytes;
  rc = alloc_buffers(ix);
  if (rc) {
    fnv_index_free(ix);
    return rc;
  }
  *out = ix;
  return FNV_OK;
}

int fnv_index_upload(const void* aos_blob, uint64_t node_size, uint64_t data_size, uint32_t M, uint64_t n_nodes,
                     int data_type, int metric, uint32_t dim, int device, fnv_index_t* out) {
  if (!aos_blob || !out) return fail(FNV_ERR_INVALID, "null argument");
  int rc = validate_geometry(M, n_nodes, data_type, metric, dim);
  if (rc) return rc;
  if (data_size != (uint64_t)dim * dtype_size(data_type) || node_size != data_size + 4ull * M + 4)
    return fail(FNV_ERR_INVALID, "node geometry does not match [data][M links][label] (Index.h:176)");
  fnv_index_t ix = nullptr;
  rc = fnv_index_alloc(M, n_nodes, data_type, metric, dim, device, &ix);
  if (rc) return rc;

  int bad = 0;
  rc = write_nodes_impl(ix, 0, n_nodes, aos_blob, node_size, data_size, n_nodes, &bad);
  if (ix->d_nodestage) {  // a one-off upload does not keep its (up to 256 MB) staging chunk
    (void)hipFree(ix->d_nodestage);
    ix->d_nodestage = nullptr;
    ix->nodestage_bytes = 0;
  }
  if (rc) {
    fnv_index_free(ix);
    return rc;
  }
  if (bad) {
    fnv_index_free(ix);
    return fail(FNV_ERR_RUNTIME, "index blob holds link ids outside [0, n_nodes)");
  }
  *out = ix;
  return FNV_OK;
}

int fnv_index_view(fnv_index_t src, fnv_index_t* out) {
  if (!src || !out) return fail(FNV_ERR_INVALID, "null argument");
  fnv_index_s* v = new fnv_index_s();
  v->owns_buffers = false;
  v->device = src->device;
  v->dtype = src->dtype;
  v->metric = src->metric;
  v->M = src->M;
  v->dim = src->dim;
  v->row_bytes = src->row_bytes;
  v->tail_bytes = src->tail_bytes;
  v->n_nodes = src->n_nodes.load();
  v->capacity = src->capacity;
  v->d_vectors = src->d_vectors;
  v->d_links = src->d_links;
  v->d_labels = src->d_labels;
  static_cast<IndexOptions&>(*v) = static_cast<const IndexOptions&>(*src);
  v->parent = src->parent ? src->parent : src;  // a view of a view hangs off the owner
  DeviceScope scope(v->device);
  if (scope.err != hipSuccess) {
    delete v;
    return fail(FNV_ERR_NO_DEVICE, "hipSetDevice failed");
  }
  int rc = index_common_init(v);
  if (rc) {
    v->parent = nullptr;
    fnv_index_free(v);
    return rc;
  }
  v->parent->n_views.fetch_add(1);
  *out = v;
  return FNV_OK;
}

int fnv_index_adopt(const void* vectors, const void* links, const void* labels, uint32_t M, uint64_t n_nodes, int data_type,
                    int metric, uint32_t dim, int device, fnv_index_t* out) {
  if (!vectors || !links || !labels || !out) return fail(FNV_ERR_INVALID, "null argument");
  int rc = validate_geometry(M, n_nodes, data_type, metric, dim);
  if (rc) return rc;
  if (((uintptr_t)vectors & 15u) || ((uintptr_t)links & 3u) || ((uintptr_t)labels & 3u))
    return fail(FNV_ERR_INVALID, "fnv_index_adopt: vectors must be 16-byte aligned, links and labels 4-byte aligned");
  fnv_index_s* v = new fnv_index_s();
  v->owns_buffers = false;
  v->device = device;
  v->dtype = data_type;
  v->metric = metric;
  v->M = M;
  v->dim = dim;
  const RowLayout lay = row_layout(dim, data_type, n_nodes);  // (the owner's layout: same dim, type and capacity)
  v->row_bytes = lay.row_bytes;
  v->tail_bytes = lay.tail_bytes;
  v->n_nodes = n_nodes;
  v->capacity = n_nodes;
  v->d_vectors = (uint8_t*)const_cast<void*>(vectors);
  v->d_links = (uint32_t*)const_cast<void*>(links);
  v->d_labels = (int32_t*)const_cast<void*>(labels);
  DeviceScope scope(device);
  if (scope.err != hipSuccess) {
    delete v;
    return fail(FNV_ERR_NO_DEVICE, "hipSetDevice failed");
  }
  rc = index_common_init(v);
  if (rc) {
    fnv_index_free(v);
    return rc;
  }
  *out = v;
  return FNV_OK;
}

int fnv_index_device_buffers(fnv_index_t ix, void* ptrs[3], uint64_t sizes[3]) {
  if (!ix || !ptrs || !sizes) return fail(FNV_ERR_INVALID, "null argument");
  ptrs[0] = ix->d_vectors;
  ptrs[1] = ix->d_links;
  ptrs[2] = ix->d_labels;
  sizes[0] = ix->vector_bytes();  // (split rows: the main table, then the side table)
  sizes[1] = ix->capacity * (uint64_t)ix->M * 4;
  sizes[2] = ix->capacity * 4;
  return FNV_OK;
}

int fnv_index_info(fnv_index_t ix, uint64_t info[8]) {
  if (!ix || !info) return fail(FNV_ERR_INVALID, "null argument");
  info[0] = (uint64_t)ix->dtype;
  info[1] = ix->M;
  info[2] = (uint64_t)ix->row_bytes | ((uint64_t)ix->tail_bytes << 32);
  info[3] = ix->parent ? ix->parent->n_nodes.load() : ix->n_nodes.load();
  info[4] = ix->dim;
  info[5] = (uint64_t)ix->metric;
  info[6] = (uint64_t)ix->device;
  info[7] = ix->capacity * ((uint64_t)ix->row_bytes + ix->tail_bytes + 4ull * ix->M + 4) + ix->bitmap_bytes + ix->spill_bytes;
  return FNV_OK;
}

int fnv_index_free(fnv_index_t ix) {
  if (!ix) return FNV_OK;
  for (std::atomic<fnv_index_s*>& l : ix->lanes) {  // the hidden lanes go first (they count as views of this handle)
    fnv_index_s* gone = l.exchange(nullptr);
    if (gone) (void)fnv_index_free(gone);
  }
  if (ix->n_views.load() > 0)
    return fail(FNV_ERR_INVALID, "fnv_index_free: the index still has live views (fnv_index_view) on its buffers; free them first");
  DeviceScope scope(ix->device);
  if (ix->stream) (void)hipStreamSynchronize(ix->stream);
  if (ix->parent) ix->parent->n_views.fetch_sub(1);
  if (!ix->owns_buffers) ix->d_vectors = nullptr, ix->d_links = nullptr, ix->d_labels = nullptr;
  void* bufs[] = {ix->d_vectors, ix->d_links, ix->d_labels, ix->d_dispenser, ix->d_bitmap, ix->d_ovf, ix->d_nodestage, ix->d_linkstage, ix->d_wirebuf, ix->d_spill, ix->d_q, ix->d_out, ix->d_phase, ix->d_entry, ix->d_done, ix->d_tielog};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (ix->h_pin) (void)hipHostFree(ix->h_pin);
  if (ix->h_res) (void)hipHostFree(ix->h_res);
  if (ix->ev0) (void)hipEventDestroy(ix->ev0);
  if (ix->ev1) (void)hipEventDestroy(ix->ev1);
  if (ix->stream) (void)hipStreamDestroy(ix->stream);
  delete ix;
  return FNV_OK;
}

int fnv_index_set_live_nodes(fnv_index_t ix, uint64_t n_live) {
  if (!ix) return fail(FNV_ERR_INVALID, "null argument");
  if (n_live == 0 || n_live > ix->capacity)
    return fail(FNV_ERR_INVALID, "live node count must be in [1, capacity]");
  std::lock_guard<std::mutex> lock(ix->mu);
  ix->n_nodes = n_live;
  return FNV_OK;
}

int fnv_index_write_nodes(fnv_index_t ix, uint64_t first_node, uint64_t count, const void* aos_rows,
                          uint64_t node_size, uint64_t data_size) {
  if (!ix || (!aos_rows && count)) return fail(FNV_ERR_INVALID, "null argument");
  if (data_size != (uint64_t)ix->dim * dtype_size(ix->dtype) || node_size != data_size + 4ull * ix->M + 4)
    return fail(FNV_ERR_INVALID, "node geometry does not match [data][M links][label] (Index.h:176)");
  if (first_node > ix->capacity || count > ix->capacity - first_node)
    return fail(FNV_ERR_RUNTIME, "Maximum number of nodes reached. (device index capacity)");
  if (count == 0) return FNV_OK;
  std::lock_guard<std::mutex> lock(ix->mu);
  int bad = 0;
  int rc = write_nodes_impl(ix, first_node, count, aos_rows, node_size, data_size, ix->capacity, &bad);
  if (rc) return rc;
  if (bad) return fail(FNV_ERR_RUNTIME, "node records hold link ids outside [0, capacity)");
  return FNV_OK;
}

int fnv_index_write_links(fnv_index_t ix, const uint32_t* node_ids, const uint32_t* link_rows, uint64_t count) {
  if (!ix || ((!node_ids || !link_rows) && count)) return fail(FNV_ERR_INVALID, "null argument");
  if (count == 0) return FNV_OK;
  std::lock_guard<std::mutex> lock(ix->mu);
  ON_DEVICE(ix->device);
  const size_t need = (size_t)count * (4 + 4ull * ix->M) + sizeof(int);
  if (need > ix->linkstage_bytes) {
    if (ix->d_linkstage) HIP_TRY(hipFree(ix->d_linkstage));
    ix->d_linkstage = nullptr;
    ix->linkstage_bytes = 0;
    HIP_TRY(hipMalloc(&ix->d_linkstage, need + need / 2));
    ix->linkstage_bytes = need + need / 2;
  }
  uint32_t* d_ids = (uint32_t*)ix->d_linkstage;
  uint32_t* d_rows = d_ids + count;
  int* d_bad = (int*)(d_rows + count * ix->M);
  HIP_TRY(hipMemcpy(d_ids, node_ids, count * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_rows, link_rows, count * 4ull * ix->M, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_bad, 0, sizeof(int)));
  hipLaunchKernelGGL(scatter_links_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, d_ids, d_rows, count,
                     ix->M, ix->capacity, ix->d_links, d_bad);
  HIP_TRY(hipGetLastError());
  int bad = 0;
  HIP_TRY(hipMemcpy(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost));
  if (bad) return fail(FNV_ERR_RUNTIME, "link rows hold node ids outside [0, capacity)");
  return FNV_OK;
}

int fnv_set_option(fnv_index_t ix, const char* name, int64_t value) {
  if (!ix || !name) return fail(FNV_ERR_INVALID, "null argument");
  std::string n(name);
  if (value < 0 && !((n == "sorted_tail_exact_pct" || n == "sorted_variant") && value == -1))
    return fail(FNV_ERR_INVALID, "option values must be non-negative");
  // Under the handle's mutex (round 5): a launch reads the options, the tuner and the layouts under it, and a concurrent
  // caller's lane copies them under it (sync_lane) -- an option may change while other threads search; each launch sees the
  // options either before or after the change.
  std::lock_guard<std::mutex> lock(ix->mu);
  if (n == "visited_factor") ix->visited_factor = std::max<int64_t>(1, value);
  else if (n == "visited_slots") {
    if (value && (value & (value - 1)) && ((value % 3) || ((value / 3) & (value / 3 - 1))))
      return fail(FNV_ERR_INVALID, "visited_slots must be 2^j or 3*2^j");
    if (value && value < 256) return fail(FNV_ERR_INVALID, "visited_slots must be at least 256");
    ix->visited_slots = value;
  } else if (n == "visited_floor") ix->visited_floor = std::max<int64_t>(256, value);
  else if (n == "occupancy_target") ix->occupancy_target = value;
  else if (n == "occupancy_roomy") ix->occupancy_roomy = std::max<int64_t>(1, value);
  else if (n == "cand_factor") ix->cand_factor = std::max<int64_t>(1, value);
  else if (n == "cand_slots") ix->cand_slots = value;
  else if (n == "spill_entries") ix->spill_entries = std::max<int64_t>(1, value);
  else if (n == "blocks_per_cu") ix->blocks_per_cu = value;
  else if (n == "visited_wide") ix->visited_wide = value;
  else if (n == "entry_kernel") ix->entry_kernel = value;
  else if (n == "output_node_ids") ix->output_node_ids = value;
  else if (n == "overflow_list") ix->overflow_list = value;
  else if (n == "sorted_beam") ix->sorted_beam = value;
  else if (n == "sorted_beam_min") ix->sorted_beam_min = value;
  else if (n == "sorted_cand_lds") ix->sorted_cand_lds = value;
  else if (n == "sorted_tail_exact_pct") ix->sorted_tail_exact_pct = value;
  else if (n == "beam_registers") ix->beam_registers = value;
  else if (n == "sorted_variant") {
    if (value >= kNumVariants) return fail(FNV_ERR_INVALID, "sorted_variant must be -1 (adaptive) or 0..6");
    ix->sorted_variant = value;
  }
  else if (n == "visited_tag_bits") ix->visited_tag_bits = value;
  else if (n == "tune_layout") ix->tune_layout = value;
  else if (n == "shadow_exact") ix->shadow_exact = value;
  else if (n == "tie_replay") ix->tie_replay = value;
  else if (n == "tie_log_entries") {
    if (value > (1 << 20)) return fail(FNV_ERR_INVALID, "tie_log_entries must be at most 1048576 records (8 MB per query slot)");
    ix->tie_log_entries = value;
  }
  else if (n == "visited_direct") ix->visited_direct = value;  // (read per launch)
  else if (n == "host_zero_copy") ix->host_zero_copy = value;  // (read per host-buffer call)
  else return fail(FNV_ERR_INVALID, "unknown option: " + n);
  // What fnv_tune measured (kernel variant, LDS layout) stays valid across options that change neither the launch plan
  // nor the kernel choice: Index.h::addBatchDevice flips output_node_ids around every device build, and a tune costs
  // dozens of launches.  (output_node_ids is read per launch; shadow_exact per launch; tune_layout by fnv_tune itself.)
  const bool keeps_tuning = n == "output_node_ids" || n == "shadow_exact" || n == "tune_layout" || n == "visited_direct" || n == "host_zero_copy";
  if (!keeps_tuning) {
    ix->options_version++;
    ix->tuner.clear();
    ix->layouts.clear();
    ix->sample_kernel = -1;
  }
  ix->tune_epoch++;  // (the second lane of the host entry point re-copies options and measurements)
  return FNV_OK;
}

// One batched search launch; node_ids: write node ids instead of labels (the device builder's beams).
static int search_device_impl(fnv_index_t ix, const void* d_queries, uint64_t nq, int K, int ef_search,
                              int num_initializations, float* d_out_dist, int32_t* d_out_labels, int32_t* d_out_count,
                              uint64_t* d_out_ndist, uint64_t* d_out_nhops, void* hip_stream, bool node_ids,
                              int force_variant = -1, bool ids_by_option = false, int32_t* host_status = nullptr);

int fnv_search_batch_device(fnv_index_t ix, const void* d_queries, uint64_t nq, int K, int ef_search,
                            int num_initializations, float* d_out_dist, int32_t* d_out_labels,
                            int32_t* d_out_count, uint64_t* d_out_ndist, uint64_t* d_out_nhops, void* hip_stream) {
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  return search_device_impl(ix, d_queries, nq, K, ef_search, num_initializations, d_out_dist, d_out_labels, d_out_count,
                            d_out_ndist, d_out_nhops, hip_stream, false, -1, /*ids_by_option=*/true);
}

// ---- launch configuration ---------------------------------------------------------------------------------
// How a query slot's LDS is laid out depends on the kernel: the two-heap kernel keeps {query, neighbours heap,
// candidates heap, visited table, staging}; the merged-beam kernel keeps {query, [beam array], visited table, staging}.
enum { MODE_HEAPS = 0, MODE_MERGED_REGS = 1, MODE_MERGED_LDS = 2 };

// Visited-table geometry for a table of `slots` (2^j or 3*2^j) and the LDS layout that follows from it; returns the
// bytes of LDS one query slot needs.  16-bit tags whenever the per-bucket id range fits 14 bits: buckets =
// mult*2^k, t = nbits - k, need t <= 14 (mult 1) or t <= 15 (mult 3).
static uint32_t lay_out(const fnv_index_s* ix, SearchParams& p, uint32_t slots, int mode) {
  uint32_t nbits = 1;
  while (nbits < 32 && (1ull << nbits) < ix->capacity) nbits++;
  const uint32_t mult = (slots % 3 == 0) ? 3u : 1u;
  uint32_t k = 0;
  for (uint32_t b = slots / 4 / mult; b > 1; b >>= 1) k++;
  const bool can16 = !ix->visited_wide && ix->visited_tag_bits <= 16 && nbits <= 30 && k <= nbits && (nbits - k) <= (mult == 3 ? 15u : 14u);
  // otherwise 64-bit buckets: three 21-bit tags (slots = 3 * 2^j) or two 32-bit tags (slots = 2^j)
  const uint32_t w = can16 ? 16u : (slots % 3 == 0 ? 21u : 32u);
  const uint32_t wbuckets = w == 21 ? slots / 3 : slots / 2;
  uint32_t wk = 0;
  for (uint32_t b = wbuckets; b > 1; b >>= 1) wk++;
  const bool canw = !can16 && !ix->visited_wide && wk <= nbits && (nbits - wk) <= w - 2;
  if (!can16 && !canw && mult == 3) slots = pow2_ceil(slots);  // the open-addressing table needs a power of two
  p.vis_slots = slots;
  p.vis_tag16 = (can16 || canw) ? 1u : 0u;
  p.vis_w = w;
  p.vis_mult = can16 ? mult : 1u;
  p.vis_nmask = (uint32_t)((1ull << nbits) - 1ull);
  p.vis_rshift = can16 ? nbits - k : (canw ? nbits - wk : 0);
  p.vis_rmask = p.vis_tag16 ? (uint32_t)((1ull << p.vis_rshift) - 1ull) : 0;
  p.vis_bytes = can16 ? slots * 2 : (canw ? wbuckets * 8 : slots * 4);
  p.vis_shift = 32;
  for (uint32_t sft = p.vis_slots; sft > 1; sft >>= 1) p.vis_shift--;
  p.vis_limit = p.vis_slots / 4 * 3;

  auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
  uint32_t off = 0;
  p.off_q = off;
  off = align16(off + p.q_lds_bytes);
  // neighbours heap (exact search) / sorted beam: arrays start at 16n + 8 so that child pairs are 16-byte aligned
  p.off_nbr = off + 8;
  // (merged-beam kernel: the same bytes stage a link row's distances, [WAVE + 1] floats, between two merges)
  off = align16(off + 8 + std::max<uint32_t>(((uint32_t)p.B + 2) * 8, mode == MODE_MERGED_REGS ? (WAVE + 1) * 4 : 0));
  p.off_stage_d = p.off_nbr;
  if (mode == MODE_MERGED_LDS) {  // LDS form: the array is the beam itself, the staging area its own
    p.off_stage_d = off;
    off = align16(off + (WAVE + 1) * 4);
  }
  p.off_cand = off + 8;  // candidates heap of the exact search: cand_slots entries in LDS (0: all of it in HBM)
  if (p.cand_slots) off = align16(off + 8 + (p.cand_slots + 1) * 8);
  p.off_vis = off;
  off = align16(off + p.vis_bytes);
  p.off_stage_ids = off;
  off = align16(off + (WAVE + 1) * 4);  // + one write-only slot for lanes with nothing to stage
  p.off_ovf = off;
  off = align16(off + (OVF_LIST + 2 + STASH) * 4);
  return off;
}

// Chooses the visited-table size for `kern` in `mode`, fills p's geometry/layout fields; outputs the LDS bytes per
// slot and the slots one CU keeps resident.
// Table sizes, ascending: 256, 384, 512, 768, ...  The roomy size (visited_factor * B + 600, <= 60 % load on the
// reference workloads) keeps every id in LDS; but LDS is also what limits how many queries a CU keeps in flight, and
// a lone wave issues slowly -- below ~13 resident queries per CU the loss of latency hiding costs more than sending
// part of the ids to the HBM bitmap (measured: profiles/r1_visited_sizing.md).  So: the largest size <= roomy that
// still leaves `occupancy_target` queries per CU, but never below visited_floor slots.
// gfx950 hands LDS out in 1280-byte granules (160 KiB = 128 of them): a workgroup that asks for 7712 bytes holds seven, and
// a CU keeps 18 such workgroups, not the floor(163840 / 7712) = 21 that hipOccupancyMaxActiveBlocksPerMultiprocessor reports.
// Measured in round 4 (tools/dev/probes/lds_granule.cpp: resident single-wave workgroups per CU against the dynamic LDS size
// -- 7680 bytes: 21, 7681: 18; 8960: 18, 8961: 16; 10240: 16, 10241: 14; 32768: 4) after the launch timeline of the uint8 index
// showed 18 busy slots per CU under a grid of 21 (profiles/r4_launch_timeline.md).
constexpr uint32_t kLdsGranule = 1280, kLdsPerCu = 160u * 1024u;
static inline uint32_t lds_allocated(uint32_t lds) { return (lds + kLdsGranule - 1) / kLdsGranule * kLdsGranule; }

static int configure_launch(fnv_index_s* ix, SearchParams& p, kernel_fn kern, int mode, uint32_t* lds_out, int* bpc_out,
                            bool grow_free = true, uint32_t forced_slots = 0) {
  // query slots one CU holds with this much LDS each, as the occupancy API counts them.  The table-size rules below were
  // calibrated against THIS number in rounds 1-3 and keep using it (same layouts as measured); what a CU really keeps
  // resident -- `really_resident` -- decides the granule trim at the end.
  auto resident = [&](uint32_t lds) -> int {
    if (lds > kLdsPerCu) return 0;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, WAVE, lds) != hipSuccess) n = 0;
    return n;
  };
  auto really_resident = [&](uint32_t lds) -> int {  // registers and wave slots: the API; LDS: whole granules
    return std::min<int>(resident(lds), (int)(kLdsPerCu / lds_allocated(std::max<uint32_t>(lds, 1u))));
  };
  uint32_t lds_bytes;
  if (forced_slots == 0) forced_slots = (uint32_t)ix->visited_slots;
  if (forced_slots) {
    lds_bytes = lay_out(ix, p, forced_slots, mode);
  } else {
    const uint64_t want = std::max<uint64_t>((uint64_t)ix->visited_factor * (uint64_t)p.B + 600, 256);
    std::vector<uint32_t> sizes;
    for (uint32_t base = 256; base <= (1u << 15); base <<= 1) {
      sizes.push_back(base);
      if (base >= want) break;
      if (base < (1u << 15)) {
        sizes.push_back(base / 2 * 3);
        if ((uint64_t)base / 2 * 3 >= want) break;
      }
    }
    size_t pick = sizes.size() - 1;  // roomy
    lds_bytes = lay_out(ix, p, sizes[pick], mode);
    (void)raise_lds_limit((const void*)kern, ix->device, std::min<uint32_t>(lds_bytes, 160u * 1024u));
    // (a table that holds every id is worth more than the last resident queries: it is kept down to
    // `occupancy_roomy` (9) of them -- measured with the merged-beam kernel at ef 160-200: -4...-18 % time at 9-11 resident
    // queries against a smaller table that overflows at 15; below that the smaller table wins again)
    const int target = (mode != MODE_HEAPS && resident(lds_bytes) >= (int)ix->occupancy_roomy) ? 0 : (int)ix->occupancy_target;
    const uint32_t roomy_tag16 = p.vis_tag16;
    for (size_t cand = pick; cand-- > 0 && sizes[cand] >= (uint32_t)ix->visited_floor && resident(lds_bytes) < target;) {
      const uint32_t smaller = lay_out(ix, p, sizes[cand], mode);
      // not a step down: the tag format lost (too few buckets for this id width), or -- wider tags per slot -- no
      // fewer bytes than the table already chosen
      if (p.vis_tag16 != roomy_tag16 || smaller >= lds_bytes) continue;
      pick = cand;
      lds_bytes = smaller;
    }
    lds_bytes = lay_out(ix, p, sizes[pick], mode);
    // A bigger table that costs no resident query is free: at ef=52 the 2048-slot table (60 % full at the end of a
    // query) already sends ids to the HBM bitmap; 3072 slots fit the same 16 queries per CU (-7 % kernel time).
    if (grow_free && pick + 1 == sizes.size()) {
      for (int step = 0; step < 2; step++) {
        const uint32_t have = p.vis_slots;
        const uint32_t next = (have & (have - 1)) == 0 ? have / 2 * 3 : have / 3 * 4;
        if (next > (1u << 15)) break;
        SearchParams q = p;
        const uint32_t bytes = lay_out(ix, q, next, mode);
        if (q.vis_tag16 != p.vis_tag16 || q.vis_slots != next || bytes > kLdsPerCu || resident(bytes) < resident(lds_bytes)) break;
        p = q;
        lds_bytes = bytes;
      }
    }
  }
  if (lds_bytes > 160u * 1024u)
    return fail(FNV_ERR_INVALID, "ef_search too large for the on-chip beam state (needs " + std::to_string(lds_bytes) +
                                     " bytes of LDS, 163840 available); lower ef_search or the *_slots options");
  HIP_TRY(raise_lds_limit((const void*)kern, ix->device, lds_bytes));
  // A layout that ends a few bytes into a granule pays a whole granule per slot for them.  If dropping at most an eighth of
  // the exact search's LDS heap entries (its overflow continues in the slot's HBM spill area; the heap is sized by rule of
  // thumb: cand_factor * B + 192) brings the slot one granule down AND that keeps one more query resident, do so
  // (the uint8 index at ef=52: 7712 -> 7680 bytes, 18 -> 21 slots per CU, +3 % queries/s, profiles/r4_launch_timeline.md).
  if (ix->cand_slots == 0 && p.cand_slots > (uint32_t)p.B + 1) {
    const uint32_t lower = lds_allocated(lds_bytes) - kLdsGranule;
    const uint32_t over = lds_bytes - lower, entries = (over + 7) / 8;
    if (lower > 0 && entries <= p.cand_slots / 8 && p.cand_slots - entries >= (uint32_t)p.B + 1 && really_resident(lower) > really_resident(lds_bytes)) {
      SearchParams q = p;
      q.cand_slots = p.cand_slots - entries;
      uint32_t bytes = lay_out(ix, q, p.vis_slots, mode);
      for (int i = 0; i < 2 && bytes > lower && q.cand_slots > (uint32_t)p.B + 2; i++) {  // (16-byte alignment of what follows the heap)
        q.cand_slots--;
        bytes = lay_out(ix, q, p.vis_slots, mode);
      }
      if (bytes <= lower && q.vis_slots == p.vis_slots && q.vis_tag16 == p.vis_tag16) {
        p = q;
        lds_bytes = bytes;
      }
    }
  }
  // The GRID is the slots a CU really keeps resident (round 5).  Rounds 1-4 launched the occupancy API's count, also where
  // that is one more than the LDS granules allow (the surplus workgroup starts when the first slot exits, finds the dispenser
  // empty and leaves): sizing the grid by `really_resident` lost 0.7-2.9 % then, because the exact tail is a percentage of the
  // grid and 75 % of the API's count sat nearer the best tail length.  With the hand-over the configurations where the two
  // counts differ run without a tail, and the two grids measure the same (c4 ef=110: 2.0857 vs 2.0837 ms, 10M x 768 ef=670:
  // 102.37 vs 102.40 ms; gpurun r5 run 24) -- so `blocks_per_cu` now says what it means.  The table-size rules above keep
  // comparing the API's counts (the layouts they choose are the measured ones).
  int bpc = really_resident(lds_bytes);
  if (bpc < 1) bpc = 1;
  if (ix->blocks_per_cu > 0) bpc = std::min<int>(bpc, (int)ix->blocks_per_cu);
  *lds_out = lds_bytes;
  *bpc_out = bpc;
  return FNV_OK;
}

static int grow(void** buf, size_t* have, size_t need, bool zero = false) {
  if (need <= *have) return FNV_OK;
  if (*buf) HIP_TRY(hipFree(*buf));
  *buf = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(buf, need));
  if (zero) HIP_TRY(hipMemset(*buf, 0, need));
  *have = need;
  return FNV_OK;
}

// Set while fnv_tune / fnv_index_insert_batch run: this thread holds ix->lane_mu AND every lane's host_mu of that handle
// (taken up front, so that no host-buffer search runs on a lane meanwhile).  release_idle_lanes then frees with the locks
// it already has -- try_lock on a std::mutex the calling thread owns is undefined behaviour.
static thread_local const fnv_index_s* t_holds_lanes_of = nullptr;
struct HoldsLanes {
  const fnv_index_s* before;
  explicit HoldsLanes(const fnv_index_s* ix) : before(t_holds_lanes_of) { t_holds_lanes_of = ix; }
  ~HoldsLanes() { t_holds_lanes_of = before; }
};

static size_t free_lane_workspace(fnv_index_s* l) {  // the caller owns l->host_mu
  std::lock_guard<std::mutex> lk(l->mu);
  DeviceScope scope(l->device);
  size_t freed = 0;
  void** bufs[] = {(void**)&l->d_bitmap, (void**)&l->d_ovf, (void**)&l->d_spill, (void**)&l->d_tielog};
  size_t* sizes[] = {&l->bitmap_bytes, &l->ovf_bytes, &l->spill_bytes, &l->tielog_bytes};
  for (int i = 0; i < 4; i++) {
    if (*bufs[i]) (void)hipFree(*bufs[i]);
    *bufs[i] = nullptr;
    freed += *sizes[i];
    *sizes[i] = 0;
  }
  l->ws_bytes = 0;
  return freed;
}

// Frees the launch workspace of every hidden lane of `ix` that is idle right now (nobody inside a call on it), except
// `keep`; returns the bytes released.  The caller holds ix->lane_mu.  `all_held`: it also holds every lane's host_mu.
static size_t release_idle_lanes_locked(fnv_index_s* ix, const fnv_index_s* keep, bool all_held = false) {
  size_t freed = 0;
  for (std::atomic<fnv_index_s*>& slot : ix->lanes) {
    fnv_index_s* l = slot.load();
    if (!l || l == keep) continue;
    if (all_held) {
      freed += free_lane_workspace(l);
      continue;
    }
    std::unique_lock<std::mutex> idle(l->host_mu, std::try_to_lock);
    if (idle.owns_lock()) freed += free_lane_workspace(l);
  }
  (void)hipGetLastError();
  return freed;
}
// ... for a caller that holds nothing of the lanes.  Never blocks: a lane in use, or a handle whose lanes are being created /
// held by another thread's fnv_tune, is left alone.  The lanes re-grow on their next call.
static size_t release_idle_lanes(fnv_index_s* ix, const fnv_index_s* keep) {
  if (t_holds_lanes_of == ix) return release_idle_lanes_locked(ix, keep, /*all_held=*/true);
  std::unique_lock<std::mutex> lanes_lock(ix->lane_mu, std::try_to_lock);
  if (!lanes_lock.owns_lock()) return 0;
  return release_idle_lanes_locked(ix, keep);
}

// Records of the hand-over log per query slot (kernels.hpp): a query logs ~6 records per beam entry on the reference workloads
// (1M x 128 at ef=52: ~310; a hop is a header + the row's admissible neighbours); 24 per entry + 512, in [1024, 16384] records
// of 8 bytes per slot = 8-128 KB, or what "tie_log_entries" says (in [WAVE + 2, 2^20]).  A log that overflows ends (the query
// is searched again from scratch if equal keys meet).
static uint32_t log_entries_for(const fnv_index_s* ix, int B) {
  if (!ix->tie_replay) return 0u;
  if (ix->tie_log_entries) return (uint32_t)std::min<int64_t>(1 << 20, std::max<int64_t>(WAVE + 2, ix->tie_log_entries));
  return std::min<uint32_t>(16384u, std::max<uint32_t>(1024u, pow2_ceil(24ull * (uint64_t)B + 512)));
}

// HBM a launch of `nq` queries needs as per-slot workspace on a handle of this index (what search_device_impl grows).
static size_t launch_workspace_bytes(fnv_index_s* ix, uint64_t nq) {
  std::lock_guard<std::mutex> lock(ix->mu);
  const uint64_t bitmap_words = ((ix->capacity + 31) / 32 + 3) / 4 * 4;
  const uint64_t ovf_cap = ix->overflow_list >= 0 ? (uint64_t)ix->overflow_list : (bitmap_words * 4 > (512u << 10) ? 16384u : 0u);
  const int per_cu = ix->plan.valid ? std::max(ix->plan.bpc, ix->plan.sbpc) : 32;  // (no plan yet: the hardware's 32 waves per CU)
  const uint64_t slots = std::min<uint64_t>(2 * nq, (uint64_t)per_cu * (uint64_t)ix->num_cus);  // (small launches: a shadow per query)
  // (the plan's own formula -- plan.sorted is unset while the plan runs the two-heap kernel only; no plan yet: the largest automatic log)
  const uint64_t log_entries = ix->plan.valid ? log_entries_for(ix, ix->plan.B) : (ix->tie_replay ? std::max<uint64_t>(16384u, (uint64_t)ix->tie_log_entries) : 0u);
  return (size_t)(slots * (bitmap_words * 4 + ovf_cap * 4 + (uint64_t)ix->spill_entries * 8 + log_entries * 8));
}

static int search_device_impl(fnv_index_t ix, const void* d_queries, uint64_t nq, int K, int ef_search,
                              int num_initializations, float* d_out_dist, int32_t* d_out_labels, int32_t* d_out_count,
                              uint64_t* d_out_ndist, uint64_t* d_out_nhops, void* hip_stream, bool node_ids,
                              int force_variant,  // >= 0: fnv_tune's launches (an argument, not index state: a concurrent
                                                  // caller's launch on the same handle is never forced)
                              bool ids_by_option,  // node_ids = the "output_node_ids" option, read under the handle's mutex
                              int32_t* host_status) {  // zero-copy small searches: the error flag's copy in the caller's pinned slab
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  // Index.h:847-849
  if (num_initializations <= 0) return fail(FNV_ERR_INVALID, "num_initializations must be greater than 0.");
  if (K <= 0 || ef_search <= 0) return fail(FNV_ERR_INVALID, "K and ef_search must be positive");
  if (nq == 0) return FNV_OK;
  if (!d_queries || !d_out_dist || !d_out_labels) return fail(FNV_ERR_INVALID, "null buffer");
  if (nq > 0x7FFFFFFFull) return fail(FNV_ERR_INVALID, "too many queries in one batch");
  std::lock_guard<std::mutex> lock(ix->mu);
  if (ids_by_option) node_ids = ix->output_node_ids != 0;
  ON_DEVICE(ix->device);
  hipStream_t stream = (hipStream_t)hip_stream;

  // ---- launch plan: depends on (beam width, K, live geometry, options) only -> cached between calls ------------
  const int B = std::max(ef_search, K);  // Index.h:392
  LaunchPlan& plan = ix->plan;
  if (!(plan.valid && plan.B == B && plan.K == K && plan.capacity == ix->capacity &&
        plan.options_version == ix->options_version)) {
    plan = LaunchPlan();
    SearchParams p;
    memset(&p, 0, sizeof(p));
    p.vectors = ix->d_vectors;
    p.tails = ix->tails();
    p.tail_chunks = ix->tail_bytes / 16;
    p.links = ix->d_links;
    p.M = ix->M;
    p.dim = ix->dim;
    p.row_bytes = ix->row_bytes;
    p.nchunks = ix->row_bytes / 16;
    p.K = K;
    p.B = B;
    const int cfg = pick_row_cfg(p.nchunks, p.tail_chunks);
    const uint32_t per_iter = (uint32_t)(kCfgs[cfg].G * kCfgs[cfg].CU);
    p.q_chunks = (p.nchunks + per_iter - 1) / per_iter * per_iter;
    if (p.tail_chunks) p.q_chunks = p.nchunks + (uint32_t)kCfgs[cfg].G;  // split rows: lane g also reads query chunk 24 + g (zero past the row)
    p.q_lds_bytes = cfg_query_in_regs(cfg) ? 0u : p.q_chunks * 16u;
    p.cand_slots = ix->cand_slots ? (uint32_t)ix->cand_slots : (uint32_t)(ix->cand_factor * p.B + 192);
    p.cand_slots = std::max<uint32_t>(p.cand_slots, (uint32_t)p.B + 1);  // also hosts the final result list
    p.spill_entries = (uint32_t)ix->spill_entries;
    p.bitmap_words = (uint32_t)(((ix->capacity + 31) / 32 + 3) / 4 * 4);  // whole 16-byte groups: wide clears
    p.ovf_cap = ix->overflow_list >= 0 ? (uint32_t)ix->overflow_list
                                       : ((uint64_t)p.bitmap_words * 4 > (512u << 10) ? 16384u : 0u);
    p.log_entries = log_entries_for(ix, p.B);
    const bool full = p.tail_chunks == 0 && (p.nchunks % per_iter) == 0;  // rows are whole spans: the lean FULL kernels apply
                                                                            // (the three-line configuration's non-FULL form IS the split-row kernel)
    plan.cfg = cfg;
    plan.full = full;

    // the exact two-heap kernel: always configured (it also replays what a merged-beam kernel hands over)
    plan.heaps = p;
    plan.kern = pick_kernel(ix->dtype, ix->metric, cfg, full);
    int rc = configure_launch(ix, plan.heaps, plan.kern, MODE_HEAPS, &plan.lds, &plan.bpc);
    if (rc) return rc;

    // Merged-beam kernel (merged_beam.hpp): the beam as one sorted array, one merge per link row -- in registers for
    // beams of at most 256 entries ("beam_registers" = 0: never), else in LDS; queries in which equal keys meet at a
    // decision are searched again by the same wave with the exact two-heap code.  Same results.  "sorted_beam":
    // 0 = never, 1 = always, 2 (default) = adaptive: measured against the two-heap kernel per beam width (below).
    const bool tagged = plan.heaps.vis_tag16 != 0;
    const bool want = ix->sorted_beam != 0 && B >= ix->sorted_beam_min && ix->capacity < (1ull << 31);
    plan.mode = (!tagged || !want) ? MODE_HEAPS : (B <= MB_MAX_BEAM && ix->beam_registers != 0) ? MODE_MERGED_REGS : MODE_MERGED_LDS;
    if (plan.mode != MODE_HEAPS) {
      plan.skern = pick_sorted_kernel(ix->dtype, ix->metric, cfg, full, plan.mode == MODE_MERGED_LDS, B);
      plan.skern_direct = pick_sorted_kernel(ix->dtype, ix->metric, cfg, full, plan.mode == MODE_MERGED_LDS, B, true);
      // a layout that fnv_tune measured for this beam width overrides the rules below (heap home, table size)
      fnv_index_s::LayoutChoice lc;
      if (auto it = ix->layouts.find(B); it != ix->layouts.end()) lc = it->second;
      const int64_t cand_lds_mode = lc.cand_lds >= 0 ? lc.cand_lds : ix->sorted_cand_lds;
      const uint32_t forced = lc.vis_slots;
      // the exact re-run's candidates heap: in LDS if that costs neither resident queries nor visited-table
      // slots, else entirely in the slot's HBM spill area (slower for the few queries that need it)
      SearchParams with = p, without = p;
      without.cand_slots = 0;
      uint32_t lds_w = 0, lds_wo = 0;
      int bpc_w = 0, bpc_wo = 0;
      rc = configure_launch(ix, without, plan.skern, plan.mode, &lds_wo, &bpc_wo, false, forced);
      if (rc) return rc;
      const int rc_w = configure_launch(ix, with, plan.skern, plan.mode, &lds_w, &bpc_w, false, forced);
      // (an exact re-run whose candidates heap lives in HBM pays a global round trip per heap operation: a handful of
      // such queries per launch are stragglers that cost 10 % of it -- measured at ef=100 on float data with 5 re-runs
      // in 10 000 queries -- so up to beams of 128 the LDS home is worth going down to 9 resident queries; wider beams'
      // heaps cost more LDS than the stragglers cost time)
      bool keep_lds = rc_w == FNV_OK && (cand_lds_mode == 1 ||
                                         (cand_lds_mode == 2 && ((bpc_w >= bpc_wo && with.vis_slots >= without.vis_slots) ||
                                                                 (B <= 2 * WAVE && bpc_w >= (int)ix->occupancy_roomy))));
      // both candidates once more with the free table growth; an LDS home that costs residency AND table slots is not taken
      SearchParams fin_w = p, fin_wo = p;
      // (p's own heap size, not `with`'s: configure_launch may already have trimmed that one by up to an eighth to fit an LDS
      //  granule, and the trim must be applied once, to the final layout)
      fin_w.cand_slots = p.cand_slots;
      fin_wo.cand_slots = 0u;
      uint32_t flds_w = 0, flds_wo = 0;
      int fbpc_w = 0, fbpc_wo = 0;
      rc = configure_launch(ix, fin_wo, plan.skern, plan.mode, &flds_wo, &fbpc_wo, true, forced);
      if (rc) return rc;
      if (keep_lds) {
        rc = configure_launch(ix, fin_w, plan.skern, plan.mode, &flds_w, &fbpc_w, true, forced);
        if (rc) return rc;
        if (cand_lds_mode == 2 && fin_w.vis_slots < fin_wo.vis_slots && fbpc_w < fbpc_wo) keep_lds = false;
      }
      plan.sorted = keep_lds ? fin_w : fin_wo;
      plan.slds = keep_lds ? flds_w : flds_wo;
      plan.sbpc = keep_lds ? fbpc_w : fbpc_wo;
      if (!plan.sorted.vis_tag16) plan.mode = MODE_HEAPS;
      if ((uint64_t)plan.sorted.cand_slots + plan.sorted.spill_entries < 3ull * (uint64_t)B + 256) plan.mode = MODE_HEAPS;
    }
    plan.B = B;
    plan.K = K;
    plan.capacity = ix->capacity;
    plan.options_version = ix->options_version;
    plan.valid = true;
  }
  // Adaptive choice ("sorted_beam" = 2): both kernels give the same answers; which one is faster depends on how often
  // equal keys force the merged-beam kernel to search a query twice (rarely on float data, often on integer-valued
  // data with wide beams) -- so it is measured: launches of at least 2048 queries are timed by the events that bracket
  // them anyway, harvested when a later call finds them complete, first one kernel, then the other, then the faster.
  bool sorted = plan.mode != MODE_HEAPS;
  bool sample = false;
  int variant = sorted ? 1 : 0;
  // The merged-beam kernel's stragglers: a query that is searched twice finishes a whole exact-search latency late, and
  // in the last round of a launch that lengthens the launch itself (one such query costs as much as hundreds).  With
  // "sorted_tail_exact_pct" = p the last p % of one round of queries skip the sorted pass (the exact search is slower but
  // never needs a second one): -15 % on the integer-valued SIFT stand-in at ef=52, +0-4 % on float data without ties --
  // so by default (-1) it is one more variant that the adaptive choice measures.
  const uint64_t round_slots = (uint64_t)plan.sbpc * (uint64_t)ix->num_cus;
  const bool multi_round = sorted && nq > round_slots;
  int64_t tail_pct = ix->sorted_tail_exact_pct < 0 ? 0 : ix->sorted_tail_exact_pct;
  bool exploratory = false;
  const int pinned = force_variant >= 0 ? force_variant : (int)ix->sorted_variant;  // fnv_tune / "sorted_variant"
  const bool shadows_on = ix->shadow_exact != 0;
  const bool try_tail = ix->sorted_tail_exact_pct < 0;
  if (sorted && pinned >= 0) {
    variant = variant_allowed(pinned, multi_round, true, shadows_on, true) ? pinned : 1;
    sorted = variant != 0;
    if (variant >= 2) tail_pct = kTailPct[variant];
  } else if (sorted && ix->sorted_beam == 2 && ix->is_lane) {
    // a hidden lane runs what its owner has measured (copied by sync_lane) and never explores or samples: a production
    // caller that happens to land on a lane must not pay for 3 x variants exploratory launches per lane
    if (nq >= 2048) {
      if (auto it = ix->tuner.find(2 * B + (multi_round ? 1 : 0)); it != ix->tuner.end()) {
        const fnv_index_s::Tuner& t = it->second;
        int best = -1;
        for (int v = 0; v < kNumVariants; v++)
          if (variant_allowed(v, multi_round, try_tail, shadows_on) && t.samples[v] > 0 && (best < 0 || t.best[v] < t.best[best])) best = v;
        if (best >= 0) variant = best;
      }
      sorted = variant != 0;
      if (variant >= 2) tail_pct = kTailPct[variant];
    }
  } else if (sorted && ix->sorted_beam == 2) {
    if (ix->sample_kernel >= 0 && ix->launched && hipEventQuery(ix->ev1) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ix->ev0, ix->ev1) == hipSuccess && ms > 0.f) {
        fnv_index_s::Tuner& t = ix->tuner[ix->sample_B];
        const float per_q = ms / (float)ix->sample_nq;
        if (t.samples[ix->sample_kernel] == 0 || per_q < t.best[ix->sample_kernel]) t.best[ix->sample_kernel] = per_q;
        t.samples[ix->sample_kernel]++;
        ix->tune_epoch++;  // (lanes re-copy the measurements)
      }
      ix->sample_kernel = -1;
    }
    if (nq >= 2048) {
      fnv_index_s::Tuner& t = ix->tuner[2 * B + (multi_round ? 1 : 0)];
      // three samples each (the first launch of a kernel is a cold one; the best of the rest decides), then the fastest
      // (fnv_tune takes all the samples in one call, so that no caller's launch is an exploratory one)
      variant = -1;
      for (int v : {1, 0, 6, 4, 3, 2, 5})
        if (variant_allowed(v, multi_round, try_tail, shadows_on) && variant < 0 && t.samples[v] < 3) variant = v;
      exploratory = variant >= 0;
      if (variant < 0) {
        variant = 0;
        for (int v = 1; v < kNumVariants; v++)
          if (variant_allowed(v, multi_round, try_tail, shadows_on) && t.samples[v] > 0 &&
              (t.best[v] < t.best[variant] || (v == 1 && t.best[1] <= t.best[0])))
            variant = v;
      }
      sample = t.samples[variant] < 4;
      sorted = variant != 0;
      if (variant >= 2) tail_pct = kTailPct[variant];
    }
  }
  const int bpc = sorted ? plan.sbpc : plan.bpc;
  uint32_t lds_bytes = sorted ? plan.slds : plan.lds;
  // Shadow mode (search_params.h): a launch that fills at most a quarter of the resident slots starts, next to the
  // merged-beam search of every query, an exact search of the same query on another slot.  A query in which equal keys
  // meet at a decision is then answered after ONE exact-search latency from the start of the launch instead of a
  // merged-beam pass plus a re-run (batch of 64 at ef=100 on the integer-valued data: p50 0.80 -> 0.47 ms); the shadow of
  // a query that needs none stops at its next hop.  Same bytes either way.
  const bool shadow = sorted && ix->shadow_exact != 0 && 4 * nq <= (uint64_t)bpc * (uint64_t)ix->num_cus;
  const uint32_t nslots = shadow ? (uint32_t)(2 * nq) : (uint32_t)std::min<uint64_t>(nq, (uint64_t)bpc * (uint64_t)ix->num_cus);
  // Tail shadows (variant 6, search_params.h): one exact shadow per slot at most -- of the queries dispensed last
  const uint32_t tail_shadows = (sorted && !shadow && variant == kVariantTailShadows) ? (uint32_t)std::min<uint64_t>(nq, nslots) : 0u;
  const uint32_t max_slots = std::max<uint32_t>(nslots, (uint32_t)std::min<uint64_t>(nq, (uint64_t)std::max(plan.bpc, plan.sbpc) * (uint64_t)ix->num_cus));

  // ---- workspace (grown on demand; sized for whichever kernel keeps more slots resident) -------------------------
  int rc = FNV_OK;
  for (int attempt = 0; attempt < 2; attempt++) {
    rc = grow((void**)&ix->d_bitmap, &ix->bitmap_bytes, (size_t)max_slots * plan.heaps.bitmap_words * 4, true);
    if (!rc) rc = grow((void**)&ix->d_ovf, &ix->ovf_bytes, (size_t)max_slots * plan.heaps.ovf_cap * 4);
    if (!rc) rc = grow((void**)&ix->d_spill, &ix->spill_bytes, (size_t)max_slots * plan.heaps.spill_entries * 8);
    if (!rc && (shadow || tail_shadows)) rc = grow((void**)&ix->d_done, &ix->done_bytes, (size_t)nq * 4);
    if (!rc && sorted) rc = grow((void**)&ix->d_tielog, &ix->tielog_bytes, (size_t)max_slots * plan.sorted.log_entries * 8);
    ix->ws_bytes = ix->bitmap_bytes + ix->ovf_bytes + ix->spill_bytes + ix->tielog_bytes;
    // out of memory on the handle itself: the hidden lanes' idle workspaces go first, then once more (from inside fnv_tune /
    // fnv_index_insert_batch, which hold the lanes, with the locks already held: t_holds_lanes_of)
    if (rc != FNV_ERR_NO_DEVICE || attempt || ix->is_lane || ix->parent) break;
    (void)hipGetLastError();
    if (release_idle_lanes(ix, nullptr) == 0) break;
  }
  if (rc) return rc;

  SearchParams p = sorted ? plan.sorted : plan.heaps;
  // Small launches on small indexes (round 5): when a bitmap of ALL node ids fits the LDS of the slots the launch needs, the
  // visited set is that bitmap (csrc/visited.hpp visited_insert_direct: one LDS round trip per link row, nothing overflows)
  // instead of the tag table, and the launch runs the kernel's DIRECT instantiation -- a launch that leaves the GPU mostly idle
  // is a chain of dependent latencies, and the tag table is 1.8 k of a lone hop's 7.9 k cycles.  The table is the last but
  // two of the slot's LDS areas: only what follows it moves.
  // Only in launches that fill at most a quarter of the slots, and not when the caller has pinned the table's shape
  // ("visited_slots", "visited_tag_bits", "visited_wide"); "visited_direct" = 0 turns it off.
  const bool small_launch = 4 * nq <= (uint64_t)bpc * (uint64_t)ix->num_cus;  // (the condition of shadow mode)
  bool direct = false;
  if (sorted && small_launch && ix->visited_direct != 0 && ix->visited_slots == 0 && ix->visited_tag_bits == 0 && ix->visited_wide == 0 && p.vis_tag16) {
    const uint64_t ids = std::max<uint64_t>(ix->capacity, ix->parent ? ix->parent->capacity : 0);
    const uint64_t bytes = ((ids + 7) / 8 + 15) / 16 * 16;
    auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
    if (bytes + p.off_vis + 1024 <= kLdsPerCu) {
      SearchParams d = p;
      d.vis_w = 1;
      d.vis_bytes = (uint32_t)bytes;
      d.vis_slots = (uint32_t)(bytes * 8);  // (what fnv_last_launch_geometry reports: one slot per node id)
      uint32_t off = align16(d.off_vis + d.vis_bytes);
      d.off_stage_ids = off;
      off = align16(off + (WAVE + 1) * 4);
      d.off_ovf = off;
      off = align16(off + (OVF_LIST + 2 + STASH) * 4);
      const uint64_t per_cu = std::min<uint64_t>((uint64_t)bpc, kLdsPerCu / lds_allocated(off));
      if (off <= kLdsPerCu && (uint64_t)nslots <= per_cu * (uint64_t)ix->num_cus) {
        p = d;
        lds_bytes = off;
        direct = true;
      }
    }
  }
  p.labels = node_ids ? nullptr : ix->d_labels;
  p.queries = (const uint8_t*)d_queries;
  p.out_dist = d_out_dist;
  p.out_labels = d_out_labels;
  p.out_count = d_out_count;
  p.out_ndist = d_out_ndist;
  p.out_nhops = d_out_nhops;
  const uint64_t live = ix->parent ? ix->parent->n_nodes.load() : ix->n_nodes.load();  // a view follows its source's growth
  p.n_nodes = live;
  p.nq = (uint32_t)nq;
  // Index.h:851-861: step = max(1, N / n_init); nodes 0, step, 2*step, ... < N
  uint64_t step = live / (uint64_t)num_initializations;
  if (step == 0) step = 1;
  p.scan_step = (uint32_t)step;
  p.n_scan = (uint32_t)((live + step - 1) / step);
  p.ovf_bitmap = ix->d_bitmap;
  p.ovf_glist = ix->d_ovf;
  p.cand_spill = ix->d_spill;
  p.tie_log = ix->d_tielog;
  if (!sorted) p.log_entries = 0u;
  p.dispenser = ix->d_dispenser;
  p.status = (int32_t*)(ix->d_dispenser + 1);
  p.host_status = host_status;
  p.redo_count = ix->d_dispenser + 3;  // [3] queries searched exactly after a tie, [4..7] by reason
  p.phase_cycles = ix->d_phase;
  p.tail_exact = multi_round && sorted ? (uint32_t)std::min<uint64_t>((uint64_t)tail_pct * nslots / 100, nq) : 0u;

  HIP_TRY(hipMemsetAsync(ix->d_dispenser, 0, 16 * sizeof(uint32_t), stream));
  HIP_TRY(hipEventRecord(ix->ev0, stream));
  if (ix->entry_kernel && !ix->tail_bytes) {  // (split rows: K0's LDS tiles hold one table's rows; the in-kernel scan serves them)
    // K0: one pass over the shared entry-scan nodes for the whole batch (LDS-staged), same stream
    rc = grow((void**)&ix->d_entry, &ix->entry_bytes, (size_t)nq * 8);
    if (rc) return rc;
    p.entry_node_out = (uint32_t*)ix->d_entry;
    p.entry_dist_out = (float*)((uint8_t*)ix->d_entry + (size_t)nq * 4);
    p.scan_tile_stride = p.row_bytes + 16;
    const uint32_t fixed = SCAN_WAVES * p.q_chunks * 16 + SCAN_QPB * 8;
    p.scan_tile_rows = std::max<uint32_t>(1, std::min<uint32_t>(p.n_scan, (64u * 1024u) / p.scan_tile_stride));
    const uint32_t scan_lds = fixed + p.scan_tile_rows * p.scan_tile_stride;
    kernel_fn scan = pick_scan_kernel(ix->dtype, ix->metric, plan.cfg, plan.full);
    HIP_TRY(raise_lds_limit((const void*)scan, ix->device, scan_lds));
    hipLaunchKernelGGL(scan, dim3((unsigned)((nq + SCAN_QPB - 1) / SCAN_QPB)), dim3(SCAN_WAVES * WAVE), scan_lds, stream, p);
    HIP_TRY(hipGetLastError());
    p.entry_node = p.entry_node_out;
    p.entry_dist = p.entry_dist_out;
  }
  if (tail_shadows) {  // the queries, then exact shadows of the last `tail_shadows` of them (pulled by slots that would go idle)
    HIP_TRY(hipMemsetAsync(ix->d_done, 0, (size_t)nq * 4, stream));
    p.shadow_base = (uint32_t)nq;
    p.done_flags = ix->d_done;
    p.nq = (uint32_t)(nq + tail_shadows);
    p.tail_exact = 0u;
  }
  if (shadow) {  // from here on the launch has 2 nq work items: the queries, then their exact shadows
    HIP_TRY(hipMemsetAsync(ix->d_done, 0, (size_t)nq * 4, stream));
    p.shadow_base = (uint32_t)nq;
    p.done_flags = ix->d_done;
    p.nq = (uint32_t)(2 * nq);
    p.tail_exact = 0u;
  }
  kernel_fn kern = !sorted ? plan.kern : direct ? plan.skern_direct : plan.skern;
  HIP_TRY(raise_lds_limit((const void*)kern, ix->device, lds_bytes));
  hipLaunchKernelGGL(kern, dim3(nslots), dim3(WAVE), lds_bytes, stream, p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ix->ev1, stream));
  ix->sample_kernel = sample ? variant : -1;
  ix->sample_B = 2 * B + (multi_round ? 1 : 0);
  ix->sample_nq = nq;
  ix->last_variant = sorted ? std::max(variant, 1) : 0;
  ix->last_exploratory = exploratory;
  if (exploratory) ix->explored_launches++;
  ix->last_shadow = shadow;
  ix->last_stream = stream;
  ix->launched = true;
  if (!ix->is_lane) ix->last_served = nullptr;  // the handle's own launch is its most recent one
  ix->geom[0] = nslots;
  ix->geom[1] = WAVE;
  ix->geom[2] = lds_bytes;
  ix->geom[3] = (uint64_t)bpc;
  ix->geom[4] = p.vis_slots;
  ix->geom[5] = p.cand_slots;
  ix->geom[6] = (uint64_t)(sorted ? plan.mode : MODE_HEAPS);
  ix->geom[7] = p.tail_exact;
  return FNV_OK;
}

int fnv_search_status(fnv_index_t ix) {
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  if (!ix->launched) return FNV_OK;
  ON_DEVICE(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->last_stream));
  int32_t st = 0;
  HIP_TRY(hipMemcpy(&st, ix->d_dispenser + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (st == ST_CAND_OVERFLOW)
    return fail(FNV_ERR_CAPACITY, "candidate heap overflowed its HBM spill area; raise the spill_entries option");
  return FNV_OK;
}

static uint64_t now_ns() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Host-buffer search in two halves, so that several devices can be kept busy by one caller: enqueue (H2D of the
// queries, the search launch, D2H of the results -- all asynchronous on the index's own stream; the caller holds
// ix->host_mu) and finish (wait, report a capacity error).
static int search_host_enqueue(fnv_index_t ix, const void* queries, uint64_t nq, int K, int ef_search,
                               int num_initializations, float* out_dist, int32_t* out_labels, int32_t* out_count,
                               uint64_t* out_ndist, uint64_t* out_nhops) {
  ON_DEVICE(ix->device);
  const size_t qbytes = (size_t)nq * ix->dim * dtype_size(ix->dtype);
  // one output slab: dist | labels | count | ndist | nhops
  const size_t o_dist = 0;
  const size_t o_lab = o_dist + (size_t)nq * K * 4;
  const size_t o_cnt = o_lab + (size_t)nq * K * 4;
  const size_t o_nd = (o_cnt + (size_t)nq * 4 + 7) & ~(size_t)7;
  const size_t o_nh = o_nd + (size_t)nq * 8;
  const size_t obytes = o_nh + (size_t)nq * 8;
  // Small batches (a single query above all) go through one pinned host buffer: one H2D copy, one D2H copy of the
  // whole slab plus the status word, no pageable-memory copies (each of those synchronises) -- the host side of a
  // one-query search drops from ~10 driver round trips to 4 asynchronous calls and one wait.
  const size_t qoff = 0, ooff = (qbytes + 63) & ~(size_t)63, soff = ooff + ((obytes + 63) & ~(size_t)63);
  const bool pinned = soff + 64 <= (1u << 20);
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    int rc = grow(&ix->d_q, &ix->d_q_bytes, qbytes);
    if (!rc) rc = grow(&ix->d_out, &ix->d_out_bytes, obytes);
    if (rc) return rc;
    if (pinned && !ix->h_pin) HIP_TRY(hipHostMalloc(&ix->h_pin, 1u << 20, hipHostMallocDefault));
    // larger batches (round 4): the queries still go in as one pageable copy, but the five result arrays come back as ONE
    // slab into pinned memory (grown on demand, up to 256 MB of results) and are scattered by the CPU -- five pageable
    // device-to-host copies each synchronise with the runtime's staging
    const size_t res_need = ((obytes + 63) & ~(size_t)63) + 64;
    if (!pinned && res_need <= (256u << 20) && res_need > ix->h_res_bytes) {
      // best effort: pinned memory is commonly capped (ulimit -l, containers) -- without the slab the five pageable
      // copies below still work
      void* old = ix->h_res;
      ix->h_res = nullptr;
      ix->h_res_bytes = 0;
      if (old) (void)hipHostFree(old);
      void* slab = nullptr;
      if (hipHostMalloc(&slab, res_need + res_need / 4, hipHostMallocDefault) == hipSuccess && slab) {
        ix->h_res = slab;
        ix->h_res_bytes = res_need + res_need / 4;
      } else {
        (void)hipGetLastError();
      }
    }
  }
  const bool pinned_results = !pinned && ix->h_res && ((obytes + 63) & ~(size_t)63) + 64 <= ix->h_res_bytes;
  uint8_t* o = (uint8_t*)ix->d_out;
  ix->pin = PinnedCall();
  // A caller whose arrays are ALREADY pinned (hipHostMalloc / hipHostRegister / torch pin_memory) needs no staging at any batch
  // size (round 6): the kernel reads the queries from and writes the results into the caller's own memory -- each query
  // crosses PCIe once while the launch runs, nothing is copied before or after it.  (Batches that fit the staging buffer take
  // the path below whatever their memory is: one attribute query per array costs more than their copies.)
  if (!pinned && ix->host_zero_copy != 0) {
    auto device_view = [](const void* host) -> void* {
      if (!host) return nullptr;
      hipPointerAttribute_t a;
      if (hipPointerGetAttributes(&a, host) != hipSuccess || a.type != hipMemoryTypeHost || !a.devicePointer) {
        (void)hipGetLastError();  // (pageable memory is "invalid value" to the runtime)
        return nullptr;
      }
      return a.devicePointer;
    };
    void* dq = device_view(queries);
    void* dd = dq ? device_view(out_dist) : nullptr;
    void* dl = dd ? device_view(out_labels) : nullptr;
    void* dc = out_count ? device_view(out_count) : nullptr;
    void* dn = out_ndist ? device_view(out_ndist) : nullptr;
    void* dh = out_nhops ? device_view(out_nhops) : nullptr;
    if (dq && dd && dl && (!out_count || dc) && (!out_ndist || dn) && (!out_nhops || dh)) {
      int rc0 = search_device_impl(ix, dq, nq, K, ef_search, num_initializations, (float*)dd, (int32_t*)dl, (int32_t*)dc, (uint64_t*)dn,
                                   (uint64_t*)dh, ix->stream, false, -1, /*ids_by_option=*/true);
      if (rc0) return rc0;
      ix->t_enqueue_ns = now_ns();
      return FNV_OK;  // (search_host_finish: wait + the launch's status word)
    }
  }
  // ZERO-COPY (round 6, "host_zero_copy" = the largest batch that takes it; default: every call that fits the 1 MB pinned
  // staging buffer): the kernel reads the queries straight from the pinned buffer and writes results, counters and its error
  // flag straight into it -- three stream operations (one copy in, two out) fewer per call; each query is read over PCIe
  // once (when it is staged), its K results are posted writes.  Measured on 1M x 128 (profiles/r6_host_zero_copy.txt): kernel
  // time +0-3 %, wall time of a call -13 ... -40 us at every batch size from 1 to 1024 queries (one query: 0.157 -> 0.143 ms
  // at ef=50; 64: 0.24-0.26 -> 0.215 ms; 1024: 0.446 -> 0.405 ms).
  if (pinned && nq <= (uint64_t)ix->host_zero_copy) {
    uint8_t* h = (uint8_t*)ix->h_pin;
    memcpy(h + qoff, queries, qbytes);
    *(int32_t*)(h + soff) = ST_OK;
    uint8_t* ho = h + ooff;
    int rc0 = search_device_impl(ix, h + qoff, nq, K, ef_search, num_initializations, (float*)(ho + o_dist), (int32_t*)(ho + o_lab),
                                 (int32_t*)(ho + o_cnt), (uint64_t*)(ho + o_nd), (uint64_t*)(ho + o_nh), ix->stream, false, -1,
                                 /*ids_by_option=*/true, (int32_t*)(h + soff));
    if (rc0) return rc0;
    ix->t_enqueue_ns = now_ns();
    PinnedCall& c = ix->pin;
    c.active = true;
    c.slab = ho;
    c.status = (const int32_t*)(h + soff);
    c.nq = nq;
    c.K = K;
    c.o_lab = o_lab; c.o_cnt = o_cnt; c.o_nd = o_nd; c.o_nh = o_nh;
    c.out_dist = out_dist; c.out_labels = out_labels; c.out_count = out_count; c.out_ndist = out_ndist; c.out_nhops = out_nhops;
    return FNV_OK;
  }
  if (pinned) {
    uint8_t* h = (uint8_t*)ix->h_pin;
    memcpy(h + qoff, queries, qbytes);
    HIP_TRY(hipMemcpyAsync(ix->d_q, h + qoff, qbytes, hipMemcpyHostToDevice, ix->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(ix->d_q, queries, qbytes, hipMemcpyHostToDevice, ix->stream));
  }
  int rc = fnv_search_batch_device(ix, ix->d_q, nq, K, ef_search, num_initializations, (float*)(o + o_dist),
                                   (int32_t*)(o + o_lab), (int32_t*)(o + o_cnt), (uint64_t*)(o + o_nd),
                                   (uint64_t*)(o + o_nh), ix->stream);
  if (rc) return rc;
  ix->t_enqueue_ns = now_ns();
  if (pinned) {
    uint8_t* h = (uint8_t*)ix->h_pin;
    HIP_TRY(hipMemcpyAsync(h + ooff, o, obytes, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipMemcpyAsync(h + soff, ix->d_dispenser + 1, sizeof(int32_t), hipMemcpyDeviceToHost, ix->stream));
    PinnedCall& c = ix->pin;
    c.active = true;
    c.slab = h + ooff;
    c.status = (const int32_t*)(h + soff);
    c.nq = nq;
    c.K = K;
    c.o_lab = o_lab; c.o_cnt = o_cnt; c.o_nd = o_nd; c.o_nh = o_nh;
    c.out_dist = out_dist; c.out_labels = out_labels; c.out_count = out_count; c.out_ndist = out_ndist; c.out_nhops = out_nhops;
    return FNV_OK;
  }
  if (pinned_results) {
    uint8_t* h = (uint8_t*)ix->h_res;
    const size_t soff2 = (obytes + 63) & ~(size_t)63;
    HIP_TRY(hipMemcpyAsync(h, o, obytes, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipMemcpyAsync(h + soff2, ix->d_dispenser + 1, sizeof(int32_t), hipMemcpyDeviceToHost, ix->stream));
    PinnedCall& c = ix->pin;
    c.active = true;
    c.slab = h;
    c.status = (const int32_t*)(h + soff2);
    c.nq = nq;
    c.K = K;
    c.o_lab = o_lab; c.o_cnt = o_cnt; c.o_nd = o_nd; c.o_nh = o_nh;
    c.out_dist = out_dist; c.out_labels = out_labels; c.out_count = out_count; c.out_ndist = out_ndist; c.out_nhops = out_nhops;
    return FNV_OK;
  }
  HIP_TRY(hipMemcpyAsync(out_dist, o + o_dist, (size_t)nq * K * 4, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipMemcpyAsync(out_labels, o + o_lab, (size_t)nq * K * 4, hipMemcpyDeviceToHost, ix->stream));
  if (out_count) HIP_TRY(hipMemcpyAsync(out_count, o + o_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, ix->stream));
  if (out_ndist) HIP_TRY(hipMemcpyAsync(out_ndist, o + o_nd, (size_t)nq * 8, hipMemcpyDeviceToHost, ix->stream));
  if (out_nhops) HIP_TRY(hipMemcpyAsync(out_nhops, o + o_nh, (size_t)nq * 8, hipMemcpyDeviceToHost, ix->stream));
  return FNV_OK;
}

static int search_host_finish(fnv_index_t ix) {
  ON_DEVICE(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->t_complete_ns = now_ns();
  PinnedCall& c = ix->pin;
  if (!c.active) return fnv_search_status(ix);
  c.active = false;
  const size_t nk = (size_t)c.nq * c.K * 4;
  memcpy(c.out_dist, c.slab, nk);
  memcpy(c.out_labels, c.slab + c.o_lab, nk);
  if (c.out_count) memcpy(c.out_count, c.slab + c.o_cnt, (size_t)c.nq * 4);
  if (c.out_ndist) memcpy(c.out_ndist, c.slab + c.o_nd, (size_t)c.nq * 8);
  if (c.out_nhops) memcpy(c.out_nhops, c.slab + c.o_nh, (size_t)c.nq * 8);
  if (*c.status == ST_CAND_OVERFLOW)
    return fail(FNV_ERR_CAPACITY, "candidate heap overflowed its HBM spill area; raise the spill_entries option");
  return FNV_OK;
}

static int check_search_args(fnv_index_t ix, const void* queries, uint64_t nq, int K, int ef_search,
                             int num_initializations, const float* out_dist, const int32_t* out_labels) {
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  if (num_initializations <= 0) return fail(FNV_ERR_INVALID, "num_initializations must be greater than 0.");
  if (K <= 0 || ef_search <= 0) return fail(FNV_ERR_INVALID, "K and ef_search must be positive");
  if (nq && (!queries || !out_dist || !out_labels)) return fail(FNV_ERR_INVALID, "null buffer");
  return FNV_OK;
}

// Hidden lane `which` (1 ... kMaxLanes - 1) of a handle (fnv_index_s::lanes): created on first contention, by a caller that
// holds ix->lane_mu and has been admitted (a lane costs a stream, two events, a dispenser and 1 MB of pinned host memory).
static fnv_index_s* create_lane_locked(fnv_index_t ix, int which) {
  fnv_index_t v = nullptr;
  if (fnv_index_view(ix, &v) != FNV_OK) return nullptr;
  v->is_lane = true;
  ix->lanes[which] = v;
  return v;
}
// The lane answers exactly like its owner: same options, same measured layouts and kernel choice (copied when they changed).
static void sync_lane(fnv_index_t ix, fnv_index_s* lane) {
  std::lock_guard<std::mutex> l1(ix->mu);
  std::lock_guard<std::mutex> l2(lane->mu);
  if (lane->lane_epoch == ix->tune_epoch && lane->lane_options == ix->options_version) return;
  static_cast<IndexOptions&>(*lane) = static_cast<const IndexOptions&>(*ix);
  lane->layouts = ix->layouts;
  lane->tuner = ix->tuner;
  lane->sample_kernel = -1;
  lane->options_version++;
  lane->plan.valid = false;
  lane->lane_epoch = ix->tune_epoch;
  lane->lane_options = ix->options_version;
}

// Large host-buffer searches take the plain path above: one pageable copy in, one launch, copies out (0.80-0.93 of the
// device-resident rate).  Round 4 tried to hide the copies inside one call, twice, and measured both slower:
// chunks of the batch as separate copy + launch + copy pipelines on two streams (every chunk pays a whole query latency
// and its own stragglers: 2.4-3.3 ms against 1.28 ms), and ONE launch that starts before its queries are there -- reading
// them from pinned host memory behind a gate word, results written straight back (every touch of host memory from a
// running wave costs microseconds: 1.7-2.6 ms).  Neither is in the tree.
int fnv_search_batch(fnv_index_t ix, const void* queries, uint64_t nq, int K, int ef_search, int num_initializations,
                     float* out_dist, int32_t* out_labels, int32_t* out_count, uint64_t* out_ndist,
                     uint64_t* out_nhops) {
  int rc = check_search_args(ix, queries, nq, K, ef_search, num_initializations, out_dist, out_labels);
  if (rc || nq == 0) return rc;
  // one caller at a time per lane (a lane's staging areas, stream and workspace are its caller's for the whole call); a
  // caller that finds the handle busy takes the second lane, a further ones take more lanes when their batches are small, else wait for the first
  fnv_index_s* lane = ix;
  std::unique_lock<std::mutex> host_lock(ix->host_mu, std::try_to_lock);
  if (!host_lock.owns_lock() && !ix->parent) {
    // A lane is taken only if the HBM its launch workspace needs fits the budget for all hidden lanes together (round 5;
    // fnv_index_s: an eighth of the device's memory) -- measured on 1M x 128 (r4_run18): 7.8 / 10.4 / 11.2 / 11.5 M queries/s
    // from 1 / 2 / 3 / 4 caller threads on four lanes; at 50M nodes one full-grid lane is 19 GB and the second does not fit.
    // Admission, reservation and creation happen under lane_mu: two concurrent callers cannot both pass the budget
    // test, and a batch that fits no lane (large index, FLATNAV_LANE_BUDGET_MB=0) creates none.  Lock order: lane_mu, then a
    // lane's host_mu by try_lock only (fnv_tune takes lane_mu, then every lane's host_mu, blocking).
    const size_t need = launch_workspace_bytes(ix, nq);
    if (need <= ix->lane_budget_bytes) {
      std::lock_guard<std::mutex> admission(ix->lane_mu);
      for (int which = 1; which < fnv_index_s::kMaxLanes; which++) {
        fnv_index_s* l = ix->lanes[which].load();
        std::unique_lock<std::mutex> lock2;
        if (l) {
          lock2 = std::unique_lock<std::mutex>(l->host_mu, std::try_to_lock);
          if (!lock2.owns_lock()) continue;
        }
        auto others = [&]() {
          size_t sum = 0;
          for (int o = 1; o < fnv_index_s::kMaxLanes; o++)
            if (fnv_index_s* other = o != which ? ix->lanes[o].load() : nullptr) sum += other->ws_bytes.load();
          return sum;
        };
        const size_t mine = std::max<size_t>(l ? l->ws_bytes.load() : 0, need);
        if (others() + mine > ix->lane_budget_bytes) (void)release_idle_lanes_locked(ix, l);  // idle lanes give their room back first
        if (others() + mine > ix->lane_budget_bytes) {
          if (!l) break;  // no further lane would fit either
          continue;       // does not fit: not this lane
        }
        if (!l) {
          l = create_lane_locked(ix, which);
          if (!l) break;
          lock2 = std::unique_lock<std::mutex>(l->host_mu);
        }
        l->ws_bytes = mine;  // the reservation: visible to the next caller's budget test before lane_mu is released
        lane = l;
        host_lock = std::move(lock2);
        break;
      }
    }
  }
  if (!host_lock.owns_lock()) host_lock = std::unique_lock<std::mutex>(ix->host_mu);
  if (lane != ix) sync_lane(ix, lane);
  rc = search_host_enqueue(lane, queries, nq, K, ef_search, num_initializations, out_dist, out_labels, out_count,
                           out_ndist, out_nhops);
  if (rc == FNV_ERR_NO_DEVICE && lane != ix) {
    // a lane could not get its workspace (another copy of the per-slot bitmaps and spill areas: 19 GB at 50M nodes): the
    // call waits for the handle's own lane instead, like any caller did before there were lanes
    (void)hipGetLastError();
    host_lock.unlock();  // (never hold a lane while waiting for the handle: fnv_tune takes them in the other order)
    host_lock = std::unique_lock<std::mutex>(ix->host_mu);
    lane = ix;
    rc = search_host_enqueue(ix, queries, nq, K, ef_search, num_initializations, out_dist, out_labels, out_count, out_ndist,
                             out_nhops);
  }
  if (rc) return rc;
  rc = search_host_finish(lane);
  if (lane != ix) {  // what fnv_last_launch_info / _geometry report for the handle: the most recent call, whichever lane
    std::lock_guard<std::mutex> l1(ix->mu);  // served it -- the whole record, under the handle's mutex (same order as sync_lane)
    std::lock_guard<std::mutex> l2(lane->mu);
    ix->t_enqueue_ns = lane->t_enqueue_ns.load();
    ix->t_complete_ns = lane->t_complete_ns.load();
    ix->last_variant = lane->last_variant;
    ix->last_exploratory = lane->last_exploratory;
    ix->last_shadow = lane->last_shadow;
    for (int i = 0; i < 8; i++) ix->geom[i] = lane->geom[i];
    ix->last_served = lane;  // (lanes live as long as the handle)
  }
  return rc;
}

// ---- several GPUs behind one call -----------------------------------------------------------------
int fnv_replicate(fnv_index_t src, int n_devices, const int* devices, fnv_index_t* out) {
  if (!src || !out || n_devices <= 0) return fail(FNV_ERR_INVALID, "null argument");
  int visible = 0;
  HIP_TRY(hipGetDeviceCount(&visible));
  for (int i = 0; i < n_devices; i++) out[i] = nullptr;
  auto undo = [&]() {
    for (int i = 0; i < n_devices; i++) {
      if (out[i]) fnv_index_free(out[i]);
      out[i] = nullptr;
    }
  };
  for (int i = 0; i < n_devices; i++) {
    const int dev = devices ? devices[i] : i;
    if (dev < 0 || dev >= visible) {
      undo();
      return fail(FNV_ERR_INVALID, "fnv_replicate: device ordinal " + std::to_string(dev) + " is not visible");
    }
    int rc = fnv_index_alloc(src->M, src->capacity, src->dtype, src->metric, src->dim, dev, &out[i]);
    if (rc) {
      undo();
      return rc;
    }
  }
  int rc = fnv_replica_refresh(src, n_devices, out);  // also hands the source's options to every replica
  if (rc) undo();
  return rc;
}

// What the source measured -- kernel variant per (beam width, batch class), LDS layout per beam width -- holds for a replica
// on the same GPU model under the same options (round 6; before, a refresh cleared it and each of the seven replicas of an
// 8-GPU run re-explored for up to 18 launches of its own).  Another GPU model starts empty and measures for itself.
// Both handles' mutexes are held by the caller (source first).
static void inherit_measurements(const fnv_index_s* src, fnv_index_s* r) {
  const bool same_model = r->num_cus == src->num_cus && r->gcn_arch == src->gcn_arch;
  if (same_model) {
    r->tuner = src->tuner;
    r->layouts = src->layouts;
  } else {
    r->tuner.clear();
    r->layouts.clear();  // measured under the old options (a stale table size could even change the kernel mode)
  }
  r->sample_kernel = -1;
  r->replica_epoch = src->tune_epoch;
  r->tune_epoch++;  // (the replica's own lanes re-copy)
}

int fnv_replica_refresh(fnv_index_t src, int n_replicas, fnv_index_t* replicas) {
  if (!src || (!replicas && n_replicas)) return fail(FNV_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(src->mu);
  int caller_device = -1;
  (void)hipGetDevice(&caller_device);
  struct Restore {
    int dev;
    ~Restore() { if (dev >= 0) (void)hipSetDevice(dev); }
  } restore{caller_device};
  const uint64_t live = src->n_nodes;
  const size_t bytes[4] = {(size_t)live * src->row_bytes, (size_t)live * src->M * 4, (size_t)live * 4, (size_t)live * src->tail_bytes};
  for (int i = 0; i < n_replicas; i++) {
    fnv_index_t r = replicas[i];
    if (!r || r->capacity < live || r->row_bytes != src->row_bytes || r->tail_bytes != src->tail_bytes || r->M != src->M ||
        r->dtype != src->dtype || r->metric != src->metric)
      return fail(FNV_ERR_INVALID, "fnv_replica_refresh: replica geometry does not match the source index");
  }
  // A replica answers its shard of a batch exactly as the source would: same options (kernel choice, node ids vs
  // labels, table sizes ...), whatever they were when the replica was made.
  for (int i = 0; i < n_replicas; i++) {
    fnv_index_t r = replicas[i];
    std::lock_guard<std::mutex> rl(r->mu);
    static_cast<IndexOptions&>(*r) = static_cast<const IndexOptions&>(*src);
    r->options_version++;
    r->replica_of = src;
    r->replica_options = src->options_version;
    inherit_measurements(src, r);
    r->plan.valid = false;
  }
  // Doubling tree of peer copies over xGMI: in every round each index that already holds the data feeds one that
  // does not (1 -> 2 -> 4 -> 8 holders: three rounds for eight GPUs, every link busy once per round).  The copies
  // of a round run concurrently on the destination indexes' streams.
  for (int i = 0; i < n_replicas; i++)  // direct xGMI copies where the platform allows them (else staged by the runtime)
    if (replicas[i]->device != src->device) {
      (void)hipSetDevice(replicas[i]->device);
      (void)hipDeviceEnablePeerAccess(src->device, 0);
      for (int j = 0; j < n_replicas; j++)
        if (replicas[j]->device != replicas[i]->device) (void)hipDeviceEnablePeerAccess(replicas[j]->device, 0);
    }
  (void)hipGetLastError();  // "peer access already enabled" is not an error
  std::vector<fnv_index_t> have{src}, need(replicas, replicas + n_replicas);
  size_t next = 0;
  while (next < need.size()) {
    const size_t senders = have.size();
    std::vector<fnv_index_t> round;
    for (size_t s = 0; s < senders && next < need.size(); s++, next++) {
      fnv_index_t from = have[s], to = need[next];
      const void* srcp[4] = {from->d_vectors, from->d_links, from->d_labels, from->tails()};
      void* dstp[4] = {to->d_vectors, to->d_links, to->d_labels, const_cast<uint8_t*>(to->tails())};
      HIP_TRY(hipSetDevice(to->device));
      for (int b = 0; b < 4; b++) {
        if (bytes[b] == 0) continue;  // (no side table)
        if (from->device == to->device)
          HIP_TRY(hipMemcpyAsync(dstp[b], srcp[b], bytes[b], hipMemcpyDeviceToDevice, to->stream));
        else
          HIP_TRY(hipMemcpyPeerAsync(dstp[b], to->device, srcp[b], from->device, bytes[b], to->stream));
      }
      round.push_back(to);
    }
    for (fnv_index_t to : round) {
      HIP_TRY(hipSetDevice(to->device));
      HIP_TRY(hipStreamSynchronize(to->stream));
      to->n_nodes = live;
      have.push_back(to);
    }
  }
  return FNV_OK;
}

int fnv_search_batch_multi(fnv_index_t* indexes, int n_indexes, const void* queries, uint64_t nq, int K, int ef_search,
                           int num_initializations, float* out_dist, int32_t* out_labels, int32_t* out_count,
                           uint64_t* out_ndist, uint64_t* out_nhops) {
  if (!indexes || n_indexes <= 0) return fail(FNV_ERR_INVALID, "null argument");
  int rc = check_search_args(indexes[0], queries, nq, K, ef_search, num_initializations, out_dist, out_labels);
  if (rc || nq == 0) return rc;
  for (int g = 1; g < n_indexes; g++)
    if (!indexes[g] || indexes[g]->dim != indexes[0]->dim || indexes[g]->dtype != indexes[0]->dtype)
      return fail(FNV_ERR_INVALID, "fnv_search_batch_multi: indexes differ in geometry");
  for (int g = 1; g < n_indexes; g++)
    if (indexes[g]->output_node_ids != indexes[0]->output_node_ids)
      return fail(FNV_ERR_INVALID, "fnv_search_batch_multi: indexes differ in the output_node_ids option (labels and node ids would mix)");
  // Rows [g * ceil(Q/G), ...) go to index g.  Every shard is driven by its own host thread (shard 0 by
  // the caller's): a shard of the bench shape is 5 MB of pageable queries, and a pageable hipMemcpyAsync blocks its
  // host thread until the copy has been staged -- issued from ONE thread, GPU g+1 would not be launched before GPU g's
  // copy (and, on one stream per device, its kernel) had been waited for.  With a thread per device the staging copies,
  // launches and waits of all devices overlap, and hipSetDevice (per-thread state) never touches the caller's device.
  // Replicas of indexes[0] take over what it has measured since their last refresh (fnv_tune on the source after
  // fnv_replicate), as long as its options are still the ones they were given.
  for (int g = 1; g < n_indexes; g++) {
    fnv_index_t r = indexes[g], src = indexes[0];
    if (r == src || r->replica_of != src) continue;
    std::lock_guard<std::mutex> l1(src->mu);
    std::lock_guard<std::mutex> l2(r->mu);
    if (r->replica_options == src->options_version && r->replica_epoch != src->tune_epoch) {
      inherit_measurements(src, r);
      r->plan.valid = false;  // (a layout choice changes the plan)
    }
  }
  const uint64_t per = (nq + (uint64_t)n_indexes - 1) / (uint64_t)n_indexes;
  const size_t qrow = (size_t)indexes[0]->dim * dtype_size(indexes[0]->dtype);
  int shards = 0;
  for (int g = 0; g < n_indexes; g++)
    if (std::min<uint64_t>(nq, (uint64_t)g * per) < nq) shards = g + 1;
  std::vector<int> rcs((size_t)shards, FNV_OK);
  std::vector<std::string> msgs((size_t)shards);
  auto run_shard = [&](int g) {
    const uint64_t lo = std::min<uint64_t>(nq, (uint64_t)g * per), hi = std::min<uint64_t>(nq, lo + per);
    rcs[g] = fnv_search_batch(indexes[g], (const uint8_t*)queries + lo * qrow, hi - lo, K, ef_search, num_initializations,
                              out_dist + lo * K, out_labels + lo * K, out_count ? out_count + lo : nullptr,
                              out_ndist ? out_ndist + lo : nullptr, out_nhops ? out_nhops + lo : nullptr);
    if (rcs[g]) msgs[g] = g_err;  // thread-local: carried back by hand
  };
  std::vector<std::thread> workers;
  for (int g = 1; g < shards; g++) workers.emplace_back(run_shard, g);
  run_shard(0);
  for (std::thread& w : workers) w.join();
  for (int g = 0; g < shards; g++)
    if (rcs[g]) return fail(rcs[g], msgs[g]);
  return FNV_OK;
}

// ---- the adaptive kernel choice, taken in one go --------------------------------------------------------------------
// Runs every variant the adaptive choice ("sorted_beam" = 2) would try for this (beam width, batch size class) on the
// caller's queries -- a cold launch plus three timed ones each -- and settles the choice, so that the caller's own
// launches never are exploratory ones.  Results are discarded (scratch buffers of the index).
int fnv_tune(fnv_index_t ix, const void* queries, uint64_t nq, int queries_on_device, int K, int ef_search,
             int num_initializations) {
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  if (num_initializations <= 0) return fail(FNV_ERR_INVALID, "num_initializations must be greater than 0.");
  if (K <= 0 || ef_search <= 0) return fail(FNV_ERR_INVALID, "K and ef_search must be positive");
  if (!queries || nq == 0) return fail(FNV_ERR_INVALID, "fnv_tune needs a batch of queries");
  std::lock_guard<std::mutex> host_lock(ix->host_mu);
  // (no host-buffer search on the hidden lanes meanwhile either: the existing ones are locked, and none is created -- a caller
  //  that wants one waits on lane_mu, holding nothing)
  std::lock_guard<std::mutex> no_new_lanes(ix->lane_mu);
  std::vector<std::unique_lock<std::mutex>> lane_locks;
  for (std::atomic<fnv_index_s*>& slot : ix->lanes)
    if (fnv_index_s* l = slot.load()) lane_locks.emplace_back(l->host_mu);
  HoldsLanes holds_lanes(ix);
  ON_DEVICE(ix->device);
  const size_t qbytes = (size_t)nq * ix->dim * dtype_size(ix->dtype);
  const size_t o_lab = (size_t)nq * K * 4, obytes = 2 * o_lab;
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    int rc = grow(&ix->d_q, &ix->d_q_bytes, 2 * qbytes);
    if (!rc) rc = grow(&ix->d_out, &ix->d_out_bytes, obytes);
    if (rc) return rc;
  }
  // The batch twice in a row: a launch on rows [off, off + nq) searches the same queries in a rotated order.  Which
  // queries tie -- and so which ones are stragglers of the last round -- is a property of the queries: timing several
  // rotations and averaging measures a variant's EXPECTED time, not its luck with one order.
  uint8_t* dq2 = (uint8_t*)ix->d_q;
  HIP_TRY(hipMemcpy(dq2, queries, qbytes, queries_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dq2 + qbytes, dq2, qbytes, hipMemcpyDeviceToDevice));
  const size_t qrow_bytes = (size_t)ix->dim * dtype_size(ix->dtype);
  uint8_t* o = (uint8_t*)ix->d_out;
  auto launch = [&](int variant, uint64_t rotate = 0) -> int {
    return search_device_impl(ix, dq2 + (rotate % nq) * qrow_bytes, nq, K, ef_search, num_initializations, (float*)o,
                              (int32_t*)(o + o_lab), nullptr, nullptr, nullptr, ix->stream, false, variant, /*ids_by_option=*/true);
  };
  // what kind of launch is this?  (one probing launch of the merged-beam kernel tells: plan, round size)
  const int B = std::max(ef_search, K);
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->layouts.erase(B);
    ix->tuner.erase(2 * B);
    ix->tuner.erase(2 * B + 1);
    ix->plan.valid = false;
    ix->tune_epoch++;  // (whatever follows -- including the early return below -- lanes drop their copy of the old choice)
  }
  int rc = launch(1);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ix->stream));
  if (ix->geom[6] == MODE_HEAPS || ix->sorted_beam != 2 || ix->sorted_variant >= 0 || nq < 2048) return FNV_OK;  // nothing to choose
  // Every timed launch starts from cold caches: the SAME queries visit the same nodes launch after launch, so the index
  // rows and -- above all -- the words of the per-slot HBM visited bitmaps they touch would sit in the 256 MiB Infinity
  // Cache by the second launch, which flatters exactly the layouts that spill most (small visited tables); a caller's
  // consecutive batches are different queries.  (Measured: without this the layout choice for 100-d rows at ef=200
  // flipped between boxes to a 4096-slot table that is 10 % slower under the bench protocol.)
  struct Flush {
    void* p = nullptr;
    size_t bytes = 768u << 20;
    ~Flush() { if (p) (void)hipFree(p); }
  } flush;
  const char* flush_env = getenv("FLATNAV_TUNE_FLUSH");  // developer knob: 0 = time warm launches
  if (flush_env && flush_env[0] == '0') flush.p = nullptr;
  else if (hipMalloc(&flush.p, flush.bytes) != hipSuccess) flush.p = nullptr;  // (no room: tune warm, as before)
  (void)hipGetLastError();
  // mean of `reps` timed launches of one variant, each on another rotation of the batch (after a cold launch), ms per query
  auto time_variant = [&](int v, int reps, float* out) -> int {
    float sum = 0.f;
    for (int rep = 0; rep <= reps; rep++) {
      if (rep > 0 && flush.p) HIP_TRY(hipMemsetAsync(flush.p, rep, flush.bytes, ix->stream));
      const int r = launch(v, rep == 0 ? 0 : (uint64_t)(rep - 1) * nq / (uint64_t)reps);
      if (r) return r;
      HIP_TRY(hipStreamSynchronize(ix->stream));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
      if (rep > 0) sum += ms;  // the first launch of a kernel is a cold one
    }
    *out = sum / (float)reps / (float)nq;
    return FNV_OK;
  };

  // ---- 1. the LDS layout ------------------------------------------------------------------------------------------
  // Resident queries, visited-table slots and the LDS home of the exact search's candidates heap trade against each
  // other, and which trade wins depends on the data (how often queries tie, how many ids a query visits): rules cover
  // the common cases (configure_launch), this measures the neighbours of the rule's choice -- heap home flipped, the
  // table one size up / down -- with the merged-beam kernel alone and with its whole last round straight to the exact
  // search, and keeps the fastest.  Skipped for whatever the caller pinned with an option.
  // (from the plan, not from the probing launch's geometry record: a DIRECT launch reports its bitmap's bits there)
  uint32_t base_slots = 0;
  bool base_heap_lds = false;
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    base_slots = ix->plan.sorted.vis_slots;
    base_heap_lds = ix->plan.sorted.cand_slots != 0;
  }
  std::vector<fnv_index_s::LayoutChoice> cands(1);  // [0]: the rules
  if (ix->tune_layout && ix->sorted_cand_lds == 2) {
    fnv_index_s::LayoutChoice c;
    c.cand_lds = base_heap_lds ? 0 : 1;
    cands.push_back(c);
  }
  if (ix->tune_layout && ix->visited_slots == 0 && base_slots >= 512) {
    const bool pow2 = (base_slots & (base_slots - 1)) == 0;
    const uint32_t up = pow2 ? base_slots / 2 * 3 : base_slots / 3 * 4, down = pow2 ? base_slots / 4 * 3 : base_slots / 3 * 2;
    // ... and two sizes up (round 5): beyond 2^24 nodes the tag format alternates with the size -- three 21-bit tags per
    // 8-byte bucket at 3 * 2^j slots, two 32-bit tags at 2^j -- so one size up from 3072 slots (4096: same bytes as 6144, a
    // third fewer tags) is a step DOWN in tags per byte and hides the layout that is 18 % faster on 50M x 128 Gaussian rows
    // (6144 slots at 8 queries per CU: 3.06 ms against the rules' 3072 slots at 12 per CU: 3.74 ms; profiles/r5_table_sizes_50m.txt)
    // (only where tags are wide: with 16-bit tags every size has the same format and the neighbours above suffice)
    const uint32_t up2 = ix->plan.sorted.vis_w != 16u ? base_slots * 2 : 0u;  // 3 * 2^j -> 3 * 2^(j+1): the same tag format
    // (two sizes DOWN was measured in round 6 and is not a candidate: on 50M x 128 uint8 -- 128-byte rows, 12 resident queries
    //  per CU -- 1536 slots keep 18-20 queries resident and are 32-48 % SLOWER than 3072 slots at 12: what a smaller table sends
    //  to the HBM bitmap costs more than the queries in flight it buys; profiles/r6_table_sizes_50m_uint8.txt)
    for (uint32_t slots : {up, down, up2}) {
      if (slots < 256 || slots > (1u << 15)) continue;
      fnv_index_s::LayoutChoice c;
      c.vis_slots = slots;
      cands.push_back(c);
      if (ix->sorted_cand_lds == 2) {
        c.cand_lds = base_heap_lds ? 0 : 1;
        cands.push_back(c);
      }
    }
  }
  const bool tune_log = getenv("FLATNAV_TUNE_LOG") != nullptr;  // developer aid: every measurement on stderr
  size_t best_layout = 0;
  float best_layout_t = -1.f;
  // one layout's time: the better of the merged-beam kernel alone and with 75 % of its last round straight to the exact search
  auto time_layout = [&](size_t li, float* out) -> int {
    {
      std::lock_guard<std::mutex> lock(ix->mu);
      if (li == 0) ix->layouts.erase(B);
      else ix->layouts[B] = cands[li];
      ix->plan.valid = false;
    }
    float t1 = -1.f, t4 = -1.f;
    int r = time_variant(1, 4, &t1);
    if (r) return r;
    const bool multi = nq > (uint64_t)ix->plan.sbpc * (uint64_t)ix->num_cus;
    if (multi && ix->sorted_tail_exact_pct < 0 && (r = time_variant(3, 4, &t4)) != FNV_OK) return r;
    *out = (t4 > 0.f && t4 < t1) ? t4 : t1;
    if (tune_log)
      fprintf(stderr, "fnv_tune B=%d layout %zu (heap in LDS %d, table %u): merged %.4f ms, 75%% tail exact %.4f ms -> %u slots, %d per CU\n", B,
              li, (int)cands[li].cand_lds, cands[li].vis_slots, t1 * (float)nq, t4 * (float)nq, (unsigned)ix->geom[4], (int)ix->geom[3]);
    return FNV_OK;
  };
  // The device settles first.  After the GPU has idled (the caller computed something on the host) the first launches run
  // up to 70 % slower than the steady state, for about 0.3 s of work -- clocks and power state (measured inside bench.py:
  // the rules' layout, timed first, lost to a neighbour it beats by 10 %; launches of 8.3 ms that take 4.8 ms half a second
  // later).  The slow state is steady while it lasts, so "two launches agree" does not end it: launches repeat for half a
  // second, then until two in a row agree.
  {
    const auto t_begin = std::chrono::steady_clock::now();
    float prev = -1.f;
    for (int i = 0; i < 400; i++) {
      float t = 0.f;
      if (time_variant(1, 1, &t) != FNV_OK) break;
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      if (tune_log) fprintf(stderr, "fnv_tune B=%d settling: %.4f ms (at %.2f s)\n", B, t * (float)nq, waited);
      if ((waited > 0.5 && prev > 0.f && fabsf(t - prev) < 0.02f * prev) || waited > 2.0) break;
      prev = t;
    }
  }
  bool won_by_duel = false;
  for (size_t li = 0; li < cands.size(); li++) {
    float t = -1.f;
    if (time_layout(li, &t) != FNV_OK) continue;  // e.g. a layout that does not fit LDS: not a candidate
    if (best_layout_t < 0.f || t < best_layout_t * 0.95f) {  // a neighbour must win by 5 %: less is box-to-box noise (round 4: a 2 % margin picked layouts that lost 5 % under the bench protocol)
      best_layout_t = t;
      best_layout = li;
      won_by_duel = false;
    } else if (t < best_layout_t * 0.985f) {
      // Inside the margin (round 6): ONE measurement cannot tell 2-4 % from drift, several that agree can -- the holder and the
      // challenger are timed twice more, alternately; the challenger takes over if it wins BOTH rounds by 1 % (round 5 saw
      // layouts that were 2-2.5 % faster on every box stay unused: 10M x 768 at 8192 slots; the means of four cold launches
      // repeat within 0.3 % -- 90.4 / 90.8 ms against 93.0 / 92.3 ms in the tune log behind profiles/r6_bench.json).
      float sum_b = 0.f, sum_c = 0.f;
      bool wins = true;
      int rounds = 0;
      for (; rounds < 2 && wins; rounds++) {
        float tb = -1.f, tc = -1.f;
        if (time_layout(best_layout, &tb) != FNV_OK || time_layout(li, &tc) != FNV_OK) wins = false;
        else wins = tc < tb * 0.99f;
        sum_b += tb;
        sum_c += tc;
      }
      if (tune_log) fprintf(stderr, "fnv_tune B=%d duel (%d round%s): layout %zu %.4f ms against layout %zu %.4f ms -> %s\n", B, rounds, rounds == 1 ? "" : "s",
                            li, sum_c / (float)rounds * (float)nq, best_layout, sum_b / (float)rounds * (float)nq,
                            wins ? "challenger" : "holder");
      if (wins) {
        best_layout_t = sum_c / (float)rounds;
        best_layout = li;
        won_by_duel = true;
      }
    }
  }
  if (best_layout != 0 && !won_by_duel) {  // a neighbour won on one measurement: the rules' layout was timed first, so it is timed once more, now last
    float again = -1.f;
    if (time_layout(0, &again) == FNV_OK && !(best_layout_t < again * 0.95f)) best_layout = 0;
  }
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    if (best_layout == 0) ix->layouts.erase(B);
    else ix->layouts[B] = cands[best_layout];
    ix->plan.valid = false;
    ix->tune_epoch++;
  }
  rc = launch(1);  // re-plan with the chosen layout
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ix->stream));

  // ---- 2. the kernel variant on that layout --------------------------------------------------------------------------
  const bool multi_round = nq > (uint64_t)ix->plan.sbpc * (uint64_t)ix->num_cus;
  fnv_index_s::Tuner t;
  for (int v = 0; v < kNumVariants; v++) {
    if (!variant_allowed(v, multi_round, ix->sorted_tail_exact_pct < 0, ix->shadow_exact != 0)) continue;
    rc = time_variant(v, 4, &t.best[v]);
    if (rc) return rc;
    if (tune_log) fprintf(stderr, "fnv_tune B=%d variant %d: %.4f ms\n", B, v, t.best[v] * (float)nq);
    t.samples[v] = 4;  // settled: later launches neither explore nor sample
  }
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->tuner[2 * B + (multi_round ? 1 : 0)] = t;
    ix->sample_kernel = -1;
    ix->tune_epoch++;
  }
  int32_t st = 0;
  HIP_TRY(hipMemcpy(&st, ix->d_dispenser + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (st == ST_CAND_OVERFLOW)
    return fail(FNV_ERR_CAPACITY, "candidate heap overflowed its HBM spill area; raise the spill_entries option");
  return FNV_OK;
}

// Measurement aid: the rate (GB/s of row bytes) a pure gather of random rows of this index's vector table reaches with
// the search kernel's own load pattern for this row width (relayout.hpp: gather_ceiling_kernel) at `waves_per_cu`
// resident waves (0: 16).  ~20-40 ms of GPU time.
int fnv_gather_ceiling(fnv_index_t ix, int waves_per_cu, double* gbps_out) {
  if (!ix || !gbps_out) return fail(FNV_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(ix->mu);
  ON_DEVICE(ix->device);
  const uint64_t n_rows = ix->parent ? ix->parent->n_nodes.load() : ix->n_nodes.load();
  const uint32_t nchunks = ix->row_bytes / 16;
  const int cfg = pick_row_cfg(nchunks, ix->tail_bytes / 16);
  typedef void (*gather_fn)(const uint8_t*, uint64_t, uint32_t, int, uint32_t*);
  static const gather_fn kGather[kNumCfgs] = {gather_ceiling_kernel<8, 1>,  gather_ceiling_kernel<8, 2>,  gather_ceiling_kernel<8, 4>,
                                              gather_ceiling_kernel<16, 4>, gather_ceiling_kernel<32, 4>, gather_ceiling_kernel<64, 4>,
                                              gather_ceiling_kernel<64, 3>, gather_ceiling_kernel<8, 3>};  // (split rows: the main table's lines)
  const int G = kCfgs[cfg].G, CU = kCfgs[cfg].CU;
  const int PU = passes_of(G, CU);
  const int wpc = waves_per_cu > 0 ? std::min(waves_per_cu, 32) : 16;
  const uint32_t blocks = (uint32_t)(ix->num_cus * wpc);
  const double rows_per_iter = (double)blocks * (WAVE / G) * PU;  // rows one iteration of the whole grid reads
  const int iters = (int)std::max<double>(8.0, 120e9 / (rows_per_iter * (double)ix->row_bytes));  // ~120 GB ~ 20 ms
  uint32_t* d_sink = ix->d_dispenser + 2;  // an unused word of the workspace
  hipLaunchKernelGGL(kGather[cfg], dim3(blocks), dim3(WAVE), 0, ix->stream, ix->d_vectors, n_rows, ix->row_bytes, iters / 8 + 1, d_sink);  // warm
  HIP_TRY(hipEventRecord(ix->ev0, ix->stream));
  hipLaunchKernelGGL(kGather[cfg], dim3(blocks), dim3(WAVE), 0, ix->stream, ix->d_vectors, n_rows, ix->row_bytes, iters, d_sink);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
  HIP_TRY(hipEventSynchronize(ix->ev1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
  ix->sample_kernel = -1;  // ev0 / ev1 no longer bracket a search launch
  *gbps_out = rows_per_iter * (double)iters * (double)ix->row_bytes / 1e9 / ((double)ms / 1e3);
  return FNV_OK;
}

int fnv_last_launch_info(fnv_index_t ix, uint64_t info[4]) {
  if (!ix || !info) return fail(FNV_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(ix->mu);
  info[0] = (uint64_t)ix->last_variant;
  info[1] = (ix->last_exploratory ? 1u : 0u) | (ix->last_shadow ? 2u : 0u);
  info[2] = ix->t_enqueue_ns;
  info[3] = ix->t_complete_ns;
  return FNV_OK;
}

int fnv_index_insert_batch(fnv_index_t ix, uint64_t first_node, uint64_t count, int ef_construction,
                           int num_initializations, uint64_t* evals_out) {
  if (!ix) return fail(FNV_ERR_INVALID, "index is null");
  if (num_initializations <= 0) return fail(FNV_ERR_INVALID, "num_initializations must be greater than 0.");
  if (ef_construction <= 0) return fail(FNV_ERR_INVALID, "ef_construction must be positive");
  if (ix->M > (uint32_t)WAVE) return fail(FNV_ERR_INVALID, "device-side wiring supports max_edges_per_node <= 64");
  if (first_node != ix->n_nodes || first_node == 0)
    return fail(FNV_ERR_INVALID, "insert_batch: first_node must equal the live node count (and the graph be non-empty)");
  if (count > ix->capacity - first_node)
    return fail(FNV_ERR_RUNTIME, "Maximum number of nodes reached. (device index capacity)");
  if (evals_out) *evals_out = 0;
  if (count == 0) return FNV_OK;
  std::lock_guard<std::mutex> host_lock(ix->host_mu);
  // (no host-buffer search on the hidden lanes meanwhile either: the existing ones are locked, and none is created -- a caller
  //  that wants one waits on lane_mu, holding nothing)
  std::lock_guard<std::mutex> no_new_lanes(ix->lane_mu);
  std::vector<std::unique_lock<std::mutex>> lane_locks;
  for (std::atomic<fnv_index_s*>& slot : ix->lanes)
    if (fnv_index_s* l = slot.load()) lane_locks.emplace_back(l->host_mu);
  HoldsLanes holds_lanes(ix);
  ON_DEVICE(ix->device);
  const int W = ef_construction;
  const uint32_t keep = std::max<uint32_t>(ix->M / 2, 1);  // Index.h:373
  const size_t esize = dtype_size(ix->dtype);
  const size_t qrow = (size_t)ix->dim * esize;
  const size_t qbytes = count * qrow;
  const size_t o_dist = 0;
  const size_t o_lab = o_dist + (size_t)count * W * 4;
  const size_t o_cnt = o_lab + (size_t)count * W * 4;
  const size_t o_nd = (o_cnt + (size_t)count * 4 + 7) & ~(size_t)7;
  const size_t obytes = o_nd + (size_t)count * 8;
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    int rcg = grow(&ix->d_q, &ix->d_q_bytes, qbytes);
    if (!rcg) rcg = grow(&ix->d_out, &ix->d_out_bytes, obytes);
    if (rcg) return rcg;
  }
  const size_t nreq = (size_t)count * keep;
  size_t sort_bytes = 0;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)nreq, 0, 32, ix->stream));
  const size_t req_bytes = (nreq * 4 + 255) & ~(size_t)255;
  {
    std::lock_guard<std::mutex> lock(ix->mu);
    int rcg = grow(&ix->d_wirebuf, &ix->wirebuf_bytes, 4 * req_bytes + sort_bytes + 256);
    if (rcg) return rcg;
  }
  uint8_t* o = (uint8_t*)ix->d_out;
  // the new nodes' vectors are the queries (Index.h:371: beamSearch(data, entry, ef_construction)); dense rows
  if (!ix->tail_bytes) {
    HIP_TRY(hipMemcpy2DAsync(ix->d_q, qrow, ix->d_vectors + first_node * (uint64_t)ix->row_bytes, ix->row_bytes, qrow,
                             count, hipMemcpyDeviceToDevice, ix->stream));
  } else {  // split rows: the main table's part of every row, then what the side table holds of it
    const size_t main_part = std::min<size_t>(qrow, ix->row_bytes), tail_part = qrow - main_part;
    HIP_TRY(hipMemcpy2DAsync(ix->d_q, qrow, ix->d_vectors + first_node * (uint64_t)ix->row_bytes, ix->row_bytes, main_part,
                             count, hipMemcpyDeviceToDevice, ix->stream));
    if (tail_part)
      HIP_TRY(hipMemcpy2DAsync((uint8_t*)ix->d_q + main_part, qrow, ix->tails() + first_node * (uint64_t)ix->tail_bytes,
                               ix->tail_bytes, tail_part, count, hipMemcpyDeviceToDevice, ix->stream));
  }
  int rc = search_device_impl(ix, ix->d_q, count, W, W, num_initializations, (float*)(o + o_dist), (int32_t*)(o + o_lab),
                              (int32_t*)(o + o_cnt), (uint64_t*)(o + o_nd), nullptr, ix->stream, /*node_ids=*/true);
  if (rc) return rc;

  std::lock_guard<std::mutex> lock(ix->mu);
  WireParams w;
  memset(&w, 0, sizeof(w));
  uint8_t* wb = (uint8_t*)ix->d_wirebuf;
  uint32_t* req_target = (uint32_t*)wb;
  uint32_t* req_index = (uint32_t*)(wb + req_bytes);
  uint32_t* sorted_target = (uint32_t*)(wb + 2 * req_bytes);
  uint32_t* sorted_req = (uint32_t*)(wb + 3 * req_bytes);
  void* sort_tmp = wb + 4 * req_bytes;
  w.vectors = ix->d_vectors;
  w.tails = ix->tails();
  w.tail_chunks = ix->tail_bytes / 16;
  w.links = ix->d_links;
  w.req_target = req_target;
  w.sorted_target = sorted_target;
  w.sorted_req = sorted_req;
  w.beam_dist = (const float*)(o + o_dist);
  w.beam_ids = (const int32_t*)(o + o_lab);
  w.beam_count = (const int32_t*)(o + o_cnt);
  w.dispenser = ix->d_dispenser;
  w.first_node = (uint32_t)first_node;
  w.count = (uint32_t)count;
  w.W = (uint32_t)W;
  w.M = ix->M;
  w.keep = keep;
  w.row_bytes = ix->row_bytes;
  w.nchunks = ix->row_bytes / 16;
  const int cfg = pick_row_cfg(w.nchunks, w.tail_chunks);
  const uint32_t per_iter = (uint32_t)(kCfgs[cfg].G * kCfgs[cfg].CU);
  w.q_chunks = (w.nchunks + per_iter - 1) / per_iter * per_iter;
  if (w.tail_chunks) w.q_chunks = w.nchunks + (uint32_t)kCfgs[cfg].G;
  const bool full = w.tail_chunks == 0 && (w.nchunks % per_iter) == 0;
  w.cap = std::max<uint32_t>((uint32_t)W, 4 * ix->M);  // connect prunes row + up to cap - M requesters at a time
  auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
  uint32_t off = 0;
  w.off_q = off;
  off = align16(off + w.q_chunks * 16);
  uint32_t* offs[] = {&w.off_ckey, &w.off_cid, &w.off_okey, &w.off_oid, &w.off_alive, &w.off_kept, &w.off_sel};
  for (uint32_t* f : offs) {
    *f = off;
    off = align16(off + (w.cap + 1) * 4);  // + a write-only bin slot
  }
  w.off_stage_ids = off;
  off = align16(off + (WAVE + 1) * 4);
  w.off_stage_idx = off;
  off = align16(off + (WAVE + 1) * 4);
  const uint32_t lds_bytes = off;
  if (lds_bytes > 160u * 1024u) return fail(FNV_ERR_INVALID, "ef_construction too large for the on-chip wiring state");
  wire_fn kernels[2] = {pick_wire_kernel(ix->dtype, ix->metric, cfg, full),
                        pick_connect_kernel(ix->dtype, ix->metric, cfg, full)};
  for (int phase = 0; phase < 2; phase++) {
    wire_fn kern = kernels[phase];
    HIP_TRY(raise_lds_limit((const void*)kern, ix->device, lds_bytes));
    int bpc = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, WAVE, lds_bytes));
    if (bpc < 1) bpc = 1;
    // select: one unit of work per new node; connect: per 64 positions of the sorted request list
    const uint64_t units = phase == 0 ? count : (nreq + WAVE - 1) / WAVE;
    const uint32_t nslots = (uint32_t)std::min<uint64_t>(units, (uint64_t)bpc * (uint64_t)ix->num_cus);
    HIP_TRY(hipMemsetAsync(ix->d_dispenser, 0, sizeof(uint32_t), ix->stream));
    hipLaunchKernelGGL(kern, dim3(nslots), dim3(WAVE), lds_bytes, ix->stream, w);
    HIP_TRY(hipGetLastError());
    if (phase == 0) {  // group the batch's back-link requests by target (stable: insertion order within a target)
      hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((nreq + 255) / 256)), dim3(256), 0, ix->stream, req_index, (uint32_t)nreq);
      HIP_TRY(hipGetLastError());
      size_t tmp = sort_bytes;
      HIP_TRY(hipcub::DeviceRadixSort::SortPairs(sort_tmp, tmp, (const uint32_t*)req_target, sorted_target,
                                                 (const uint32_t*)req_index, sorted_req, (int)nreq, 0, 32, ix->stream));
    }
  }
  std::vector<uint64_t> nd;
  if (evals_out) {
    nd.resize(count);
    HIP_TRY(hipMemcpyAsync(nd.data(), o + o_nd, count * 8, hipMemcpyDeviceToHost, ix->stream));
  }
  HIP_TRY(hipStreamSynchronize(ix->stream));
  if (evals_out) {
    uint64_t total = 0;
    for (uint64_t e : nd) total += e;
    *evals_out = total;
  }
  int32_t st = 0;
  HIP_TRY(hipMemcpy(&st, ix->d_dispenser + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (st == ST_CAND_OVERFLOW)
    return fail(FNV_ERR_CAPACITY, "candidate heap overflowed its HBM spill area; raise spill_entries");
  ix->n_nodes = first_node + count;
  return FNV_OK;
}

int fnv_index_read_links(fnv_index_t ix, uint64_t first_node, uint64_t count, uint32_t* out_rows) {
  if (!ix || (!out_rows && count)) return fail(FNV_ERR_INVALID, "null argument");
  if (first_node > ix->capacity || count > ix->capacity - first_node)
    return fail(FNV_ERR_INVALID, "node range outside the device index");
  if (count == 0) return FNV_OK;
  std::lock_guard<std::mutex> lock(ix->mu);
  ON_DEVICE(ix->device);
  HIP_TRY(hipMemcpy(out_rows, ix->d_links + first_node * (uint64_t)ix->M, count * 4ull * ix->M, hipMemcpyDeviceToHost));
  return FNV_OK;
}

// The handle whose events / counters describe the most recent launch of `ix`: a hidden lane if that served the last host-buffer call.
static fnv_index_s* last_launcher(fnv_index_t ix) {
  fnv_index_s* lane = ix->last_served.load();
  return lane ? lane : ix;
}

int fnv_last_kernel_ms(fnv_index_t ix, float* ms) {
  if (!ix || !ms) return fail(FNV_ERR_INVALID, "null argument");
  ix = last_launcher(ix);
  if (!ix->launched) return fail(FNV_ERR_RUNTIME, "no search has been launched on this index");
  ON_DEVICE(ix->device);
  HIP_TRY(hipEventSynchronize(ix->ev1));
  HIP_TRY(hipEventElapsedTime(ms, ix->ev0, ix->ev1));
  return FNV_OK;
}

#ifdef FNV_PHASE_TIMING
int fnv_debug_heap_microbench(int size, int iters, int blocks, uint64_t out[4]) {
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc(&d, 4 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(heap_microbench_kernel, dim3(blocks), dim3(WAVE), (size + 4) * 8 + 64, 0, size, iters, d);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipFree(d));
  return FNV_OK;
}
// Profiling builds only: cumulative shader cycles per kernel phase (and reset).
int fnv_debug_phase_cycles(fnv_index_t ix, uint64_t out[16]) {
  if (!ix || !out) return fail(FNV_ERR_INVALID, "null argument");
  ON_DEVICE(ix->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, ix->d_phase, NPHASE * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(ix->d_phase, 0, NPHASE * sizeof(uint64_t)));
  return FNV_OK;
}
#endif

#ifdef FNV_SPEC_STATS
// Developer builds only (-DFNV_SPEC_STATS): {hops whose node was the runner-up guessed one hop ahead, hops} of the last launch's
// queries that finished in the merged-beam kernel.
int fnv_debug_spec_stats(fnv_index_t ix, uint64_t out[2]) {
  if (!ix || !out) return fail(FNV_ERR_INVALID, "null argument");
  ix = last_launcher(ix);
  ON_DEVICE(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->last_stream));
  uint32_t w[2];
  HIP_TRY(hipMemcpy(w, ix->d_dispenser + 11, sizeof(w), hipMemcpyDeviceToHost));
  out[0] = w[0];
  out[1] = w[1];
  return FNV_OK;
}
#endif

int fnv_last_replayed_queries(fnv_index_t ix, uint64_t out[5]) {
  if (!ix || !out) return fail(FNV_ERR_INVALID, "null argument");
  for (int i = 0; i < 5; i++) out[i] = 0;
  ix = last_launcher(ix);
  if (!ix->launched) return FNV_OK;
  ON_DEVICE(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->last_stream));
  uint32_t w[5];
  HIP_TRY(hipMemcpy(w, ix->d_dispenser + 3, sizeof(w), hipMemcpyDeviceToHost));
  for (int i = 0; i < 5; i++) out[i] = w[i];
  return FNV_OK;
}

int fnv_last_handover_stats(fnv_index_t ix, uint64_t out[4]) {
  if (!ix || !out) return fail(FNV_ERR_INVALID, "null argument");
  for (int i = 0; i < 4; i++) out[i] = 0;
  ix = last_launcher(ix);
  if (!ix->launched) return FNV_OK;
  ON_DEVICE(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->last_stream));
  uint32_t w[8];
  HIP_TRY(hipMemcpy(w, ix->d_dispenser + 3, sizeof(w), hipMemcpyDeviceToHost));
  out[0] = w[5];         // queries resumed from their hand-over log
  out[1] = w[6];         // hops taken from the logs
  out[2] = w[7];         // hops the merged-beam passes of those queries had made
  out[3] = w[0] - w[5];  // queries searched again from scratch
  return FNV_OK;
}

int fnv_lane_info(fnv_index_t ix, uint64_t info[16]) {
  if (!ix || !info) return fail(FNV_ERR_INVALID, "null argument");
  for (int i = 0; i < fnv_index_s::kMaxLanes; i++) {
    const fnv_index_s* l = i == 0 ? ix : ix->lanes[i].load();
    info[2 * i] = l ? (uint64_t)l->ws_bytes.load() : 0;
    info[2 * i + 1] = l ? l->explored_launches.load() : 0;
  }
  return FNV_OK;
}

int fnv_last_launch_geometry(fnv_index_t ix, uint64_t geom[8]) {
  if (!ix || !geom) return fail(FNV_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lock(ix->mu);
  for (int i = 0; i < 8; i++) geom[i] = ix->geom[i];
  return FNV_OK;
}

}  // extern "C"
#pragma GCC visibility pop
