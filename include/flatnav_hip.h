/* flatnav_hip.h -- C ABI of the MI355X (gfx950) flat-NSW search engine.
 *
 * This is the drop-in boundary for ONE path of BlaiseMuhirwa/flatnav: batched
 * k-NN search (greedy beam traversal over the flat navigable-small-world graph).
 * Everything here is plain C: opaque handle, pointers, sizes, int status codes.
 * No torch / pybind / C++ types cross this line.
 *
 * Reference interfaces each entry point stands in for (paths relative to the
 * reference repository root):
 *
 *   fnv_index_upload        the host index memory that search reads:
 *                           include/flatnav/index/Index.h:56 (_index_memory),
 *                           :61-63 and :555-573 (AoS node = [data][M links][label]),
 *                           as produced by Index::add / Index::loadIndex (:353, :442).
 *   fnv_search_batch        the batched search loop of the Python binding,
 *                           python-bindings/src/flatnav/bindings.cpp:161-228
 *                           (searchImpl: for each query Index::search + copy K results),
 *                           i.e. Index::search, include/flatnav/index/Index.h:387-409 with
 *                           initializeSearch :845-870, beamSearch :606-659,
 *                           processCandidateNode :661-707 and the distance dispatchers
 *                           include/flatnav/distances/L2DistanceDispatcher.h:121-126,
 *                           IPDistanceDispatcher.h:95-100.
 *   fnv_search_batch_device same, for callers that already keep queries/results in HBM
 *                           (flatnav::executeInParallel over query rows,
 *                           include/flatnav/util/Multithreading.h:19-48, becomes the GPU grid).
 *   out_ndist / out_nhops   Index::_distance_computations, Index.h:83, 689-691, 857-859
 *                           (kept per query instead of one shared atomic).
 *   fnv_replicate / fnv_search_batch_multi
 *                           the batch parallelism of the binding (bindings.cpp:198-211:
 *                           executeInParallel over query rows on one shared index) across
 *                           the GPUs of a node: index replicated by peer copies over xGMI at
 *                           load, query rows sharded, no per-query collective.
 *   fnv_index_device_buffers / fnv_index_alloc
 *                           the same for one-process-per-GPU callers: they broadcast the three
 *                           device buffers with RCCL (one ncclBroadcast each at load).
 *
 * Status codes mirror the exception the reference would throw at that point:
 *   FNV_ERR_INVALID  -> std::invalid_argument (Python ValueError)
 *   FNV_ERR_RUNTIME  -> std::runtime_error    (Python RuntimeError)
 */
#ifndef FLATNAV_HIP_H
#define FLATNAV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fnv_index_s* fnv_index_t;

enum {
  FNV_OK = 0,
  FNV_ERR_INVALID = 1,   /* bad argument (std::invalid_argument in the reference)            */
  FNV_ERR_RUNTIME = 2,   /* runtime failure (std::runtime_error in the reference)             */
  FNV_ERR_NO_DEVICE = 3, /* no usable gfx950 device / HIP runtime error                       */
  FNV_ERR_CAPACITY = 4   /* a per-query on-device structure overflowed its spill area         */
};

/* flatnav::util::DataType ordinals (include/flatnav/util/Datatype.h:11-24); also the file format's. */
enum { FNV_DTYPE_UINT8 = 0, FNV_DTYPE_INT8 = 4, FNV_DTYPE_FLOAT32 = 9 };
/* flatnav::distances::MetricType (include/flatnav/distances/DistanceInterface.h:14). */
enum { FNV_METRIC_L2 = 0, FNV_METRIC_IP = 1 };

/* Message of the last failing call on this thread (never NULL). */
const char* fnv_last_error(void);

/* Library / build identification, e.g. "flatnav_hip gfx950 r3". */
const char* fnv_version(void);

/* Number of visible HIP devices. */
int fnv_device_count(int* count);

/* Upload a host AoS index blob (exactly the bytes flatnav keeps in _index_memory / writes to its
 * .bin file after the 60-byte header) to `device` and re-lay it out for the GPU:
 *   vectors [n_nodes][row_bytes]  (row_bytes = data_size rounded up to 16 bytes, and on to whole 128-byte lines
 *                                  when that pads by at most FLATNAV_ROW_PAD_PCT per cent -- environment variable,
 *                                  default 30, 0 = never: 100-d float32 rows take 512 bytes = exactly four lines
 *                                  instead of straddling four to five; zero padded, distances keep their bits)
 *   links   [n_nodes][M] uint32   (duplicate ids inside a row are replaced by the node's own id,
 *                                  which the search treats exactly like the reference treats an
 *                                  already-visited link)
 *   labels  [n_nodes] int32
 * n_nodes is the reference's _cur_num_nodes (NOT max_node_count). The blob is not retained. */
int fnv_index_upload(const void* aos_blob, uint64_t node_size_bytes, uint64_t data_size_bytes, uint32_t M,
                     uint64_t n_nodes, int data_type, int metric, uint32_t dim, int device,
                     fnv_index_t* out);

/* Allocate an EMPTY device index of the given geometry (buffers uninitialised) so that a replica
 * can be filled by a collective (RCCL broadcast) or a peer copy. */
int fnv_index_alloc(uint32_t M, uint64_t n_nodes, int data_type, int metric, uint32_t dim, int device,
                    fnv_index_t* out);

/* A second handle on the SAME device buffers (vectors / links / labels are shared, not copied) with its own workspace,
 * stream and options (copied from `src` at creation): lets callers keep several searches in flight on one index
 * (fnv_search_batch_device allows one launch in flight per handle).  Two launches in flight on two handles / two streams
 * overlap one launch's drain (its last, slowest queries at falling occupancy) with the next one's start: +25-40 %
 * queries/s on back-to-back 10 000-query batches (round 3: 9.7 -> 11.9 M on 1M x 128 float32, 12.7 -> 19.3 M on the uint8
 * index; INTEGRATION.md recommends it to callers that always have the next batch ready).  The view reads the source's live node
 * count at every launch (it follows fnv_index_set_live_nodes / fnv_index_insert_batch on the source); the source
 * counts its views and fnv_index_free(source) fails with FNV_ERR_INVALID while any is alive: free views first. */
int fnv_index_view(fnv_index_t src, fnv_index_t* out);

/* A handle on device buffers that SOMEBODY ELSE owns and keeps alive -- laid out as fnv_index_device_buffers reports
 * them ([n_nodes][row_bytes] vectors at the library's row stride, see fnv_index_info[2]; [n_nodes][M] uint32 links;
 * [n_nodes] int32 labels), all on `device`: e.g. buffers that another process shares over HIP IPC, that a collective
 * filled inside a framework's allocation, or that another build of this library uploaded (tools/dev/knob_sweep.py A/Bs
 * library builds on one copy of a 32 GB index this way).  `n_nodes` is the CAPACITY the buffers were laid out for: with split
 * rows (fnv_row_layout) the side table starts at vectors + n_nodes * row_bytes; fewer live nodes: fnv_index_set_live_nodes.
 * The handle has its own workspace, stream and options and never frees the buffers; there is no reference interface for this (the reference's index lives in one process's heap). */
int fnv_index_adopt(const void* vectors, const void* links, const void* labels, uint32_t M, uint64_t n_nodes,
                    int data_type, int metric, uint32_t dim, int device, fnv_index_t* out);

/* How the library lays out the vector table of an index of this geometry, without a handle and without a GPU (round 6): what a
 * caller that fills buffers for fnv_index_adopt itself has to know.  *row_bytes = stride of the table (16-byte chunks, whole
 * 128-byte lines when that pads <= 30 %: FLATNAV_ROW_PAD_PCT); *tail_bytes != 0: split rows -- the table holds the row's three whole
 * lines, the last 16 / 32 bytes of row i live at vectors + capacity * row_bytes + i * tail_bytes (FLATNAV_SPLIT_ROWS,
 * FLATNAV_SPLIT_TAIL_MAX_MB; see fnv_index_info).  The vectors buffer is capacity * (row_bytes + tail_bytes) bytes, zero beyond
 * each row's dim elements.  (No reference interface: the reference's layout is the AoS node record, Index.h:61-63, 555-573.) */
int fnv_row_layout(uint32_t dim, int data_type, uint64_t capacity, uint32_t* row_bytes, uint32_t* tail_bytes);

/* Device pointers and byte sizes of the three index buffers: [0]=vectors (split rows: table + side table) [1]=links [2]=labels. */
int fnv_index_device_buffers(fnv_index_t index, void* ptrs[3], uint64_t sizes[3]);

/* info[8] = {data_type, M, row_bytes | tail_bytes << 32, n_nodes, dim, metric, device, total_device_bytes}.
 * row_bytes = stride of the vector table.  tail_bytes != 0 ("split rows", round 6): rows of three whole 128-byte lines plus at
 * most 32 bytes (e.g. 100-d float32) keep their whole lines in the table (stride 384) and their last 16 / 32 bytes in a dense
 * side table that FOLLOWS the table in the same buffer: row i's tail at vectors + capacity * row_bytes + i * tail_bytes. */
int fnv_index_info(fnv_index_t index, uint64_t info[8]);

int fnv_index_free(fnv_index_t index);

/* ---- incremental construction --------------------------------------------------
 * The reference inserts a point by running the SAME beam search over the nodes present so far
 * (Index::add, include/flatnav/index/Index.h:353-378: initializeSearch + beamSearch with
 * ef_construction), then selectNeighbors / connectNeighbors (:714-834).  These three calls let a
 * host builder keep a device index in step with its node store so that the beam searches of a
 * whole batch of insertions run as one fnv_search_batch launch:
 *   - allocate at full capacity (fnv_index_alloc), searches see only nodes [0, n_live);
 *   - fnv_index_write_nodes copies AoS node records first..first+count-1 (same layout as
 *     fnv_index_upload) into the device buffers;
 *   - fnv_index_write_links overwrites the link rows (M ids each) of `count` scattered nodes after
 *     the host has wired a batch.
 * With option "output_node_ids" = 1 searches return node ids instead of labels.
 * Callers must not run these concurrently with a search on the same handle. */
int fnv_index_set_live_nodes(fnv_index_t index, uint64_t n_live);
int fnv_index_write_nodes(fnv_index_t index, uint64_t first_node, uint64_t count, const void* aos_rows,
                          uint64_t node_size_bytes, uint64_t data_size_bytes);
int fnv_index_write_links(fnv_index_t index, const uint32_t* node_ids, const uint32_t* link_rows,
                          uint64_t count);

/* Whole insertions on the device: nodes first_node..first_node+count-1 (records already written with
 * fnv_index_write_nodes, first_node == live count) are searched for with beam width ef_construction
 * against the live graph (Index.h:367-371), then wired by two kernels: wire_select_kernel (selectNeighbors
 * to M/2, Index.h:714-763; own row; one back-link request per kept neighbour) and wire_connect_kernel (one
 * wavefront per target node: requesters take free slots, else row + requesters are re-pruned once,
 * connectNeighbors Index.h:765-834).  On return the batch is live.  evals_out (nullable) receives the distance
 * evaluations of the beam searches (what the reference adds to its counter, Index.h:689-691).
 * max_edges_per_node <= 64.  fnv_index_read_links copies link rows back for the host node store. */
int fnv_index_insert_batch(fnv_index_t index, uint64_t first_node, uint64_t count, int ef_construction,
                           int num_initializations, uint64_t* evals_out);
int fnv_index_read_links(fnv_index_t index, uint64_t first_node, uint64_t count, uint32_t* out_rows);

/* Tuning / test knobs (all optional).  Names:
 *   "visited_factor"  roomy LDS visited-table size = visited_factor * beam width + 600 slots, rounded up to
 *                     2^j or 3*2^j (default 27); used as is while "occupancy_target" queries fit per CU
 *   "occupancy_roomy" the merged-beam kernel keeps the roomy table while at least this many queries stay resident per
 *                     CU (default 9: a table that holds every id beats the last few resident queries)
 *   "occupancy_target" resident queries per CU below which the table is shrunk step by step (ids that
 *                     find both their buckets full go to the per-slot HBM bitmap, results unchanged);
 *                     default 13, 0 = never shrink
 *   "visited_floor"   the table is not shrunk below this many slots (default 2048)
 *   "visited_tag_bits" 0 = automatic tag width of the bucketed table (16-bit tags while the id range allows it at
 *                     the chosen size, else three 21-bit or two 32-bit tags per 64-bit bucket); 21 / 32 = never
 *                     use 16-bit tags (tests)
 *   "overflow_list"   ids per query slot remembered in HBM so that only the touched words of the overflow
 *                     bitmap are cleared; default: 16384 when the bitmap (N/8 bytes) exceeds 512 KB, else 0
 *                     (a small bitmap is cleared whole)
 *   "visited_slots"   force the LDS visited-table size (2^j or 3*2^j; 0 = automatic as above)
 *   "cand_factor"     LDS candidate-heap capacity = cand_factor * beam width + 192 (default 2)
 *   "cand_slots"      force the LDS candidate-heap capacity (0 = from factor)
 *   "spill_entries"   per-slot HBM spill capacity of the candidate heap (default 16384)
 *   "blocks_per_cu"   cap resident query slots per CU (0 = occupancy limit)
 *   "output_node_ids" 1 = out_labels receives node ids, not labels (used by the device-assisted builder)
 *   "sorted_beam"     the merged-beam kernel (csrc/merged_beam.hpp: the beam as one sorted array instead of the
 *                     reference's two heaps, one merge per link row; a query in which equal distances meet at a
 *                     decision is searched again by the same wavefront with the exact two-heap code, so results are
 *                     the same either way):
 *                     0 = never (two-heap kernel only), 1 = always, 2 (default) = adaptive -- launches of >= 2048
 *                     queries are timed per beam width, first each variant three times, then the fastest serves that beam
 *                     width (which one wins depends on how often the data ties).  Needs capacity < 2^31 nodes.
 *   "sorted_tail_exact_pct"  the last p % of one round of queries (one round = as many queries as stay resident)
 *                     of a merged-beam launch go straight to the exact search: a query that is searched twice
 *                     finishes late, and in the last round that lengthens the whole launch.  -1 (default) = one more
 *                     variant for the adaptive choice to measure (it tries 0, 25, 50, 75 and 100); >= 0 = fixed
 *   "beam_registers"  != 0 (default): beams of at most 256 entries keep the sorted array in registers (the merge's
 *                     permutation goes through LDS); 0 = the array always lives in LDS, as it does for wider beams
 *   "sorted_variant"  -1 (default) = the adaptive choice above; 0..6 = pin what a merged-beam-capable launch runs:
 *                     0 two-heap kernel, 1 merged-beam kernel, 2 / 3 / 4 / 5 merged-beam kernel with the last 50 / 75 /
 *                     100 / 25 % of a round straight to the exact search (launches of one round or less run 1 instead),
 *                     6 (round 4) merged-beam kernel for every query plus exact "tail shadows": the slots that run out of
 *                     queries search the most recently started ones exactly, so a tie in the last round costs one
 *                     exact-search latency from the query's start without the whole round paying for the slower kernel
 *                     (needs "shadow_exact" != 0, else 1 runs)
 *   "shadow_exact"    1 (default): a launch that fills at most a quarter of the resident query slots (a single query, a
 *                     batch of 64 ...) starts, next to the merged-beam search of every query, an exact two-heap search of
 *                     the same query on another slot; a query in which equal distances meet at a decision is then
 *                     answered after one exact-search latency from the start of the launch instead of a merged-beam pass
 *                     plus a re-run, the shadow of a query that needs none stops at its next hop.  0 = off.  Same bytes.
 *   "tie_replay"      1 (default): a query in which equal distances meet at a decision is resumed from its hand-over log
 *                     (fnv_last_handover_stats) instead of being searched again from scratch; 0 = from scratch, as in
 *                     rounds 2-4.  "tie_log_entries": 8-byte log records per resident query slot in HBM (0 = automatic:
 *                     24 per beam entry + 512, in [1024, 16384]).  Same bytes.
 *   "visited_direct"  1 (default): a launch that fills at most a quarter of the resident query slots keeps its visited set as a
 *                     plain bitmap of ALL node ids in LDS whenever that fits the slots it needs (up to ~1.2 M nodes at one
 *                     query per CU): one LDS round trip per link row, nothing overflows.  0 = the tag table always; also off
 *                     while "visited_slots", "visited_tag_bits" or "visited_wide" pin the table's shape.  Same bytes.
 *   "host_zero_copy"  (round 6; default 2^20 = every call that fits the handle's 1 MB pinned staging buffer) fnv_search_batch calls
 *                     of at most this many queries run zero-copy: the kernel reads the queries from the pinned buffer and
 *                     writes results, counters and its error flag straight into it -- three stream operations fewer per
 *                     call (one query at ef=50 on 1M x 128: 0.157 -> 0.143 ms wall).  Larger calls whose arrays (queries AND
 *                     every output passed) are already pinned host memory -- hipHostMalloc, hipHostRegister, torch's
 *                     pin_memory -- run zero-copy on the caller's own memory.  0 = every call copies in and out.  Same bytes.
 *   "tune_layout"     1 (default): fnv_tune also measures the LDS layout (see fnv_tune); 0 = kernel variants only
 *   "sorted_beam_min" smallest beam width the merged-beam kernel is used for (default 1)
 *   "sorted_cand_lds" where the exact re-run of the merged-beam kernel keeps its candidates heap: 2 (default) = in LDS
 *                     when that costs neither resident queries nor visited-table slots, or -- beams of at most 128
 *                     entries -- leaves "occupancy_roomy" resident queries; else in the slot's HBM spill area;
 *                     0 = always HBM, 1 = always LDS (tests)
 *   "entry_kernel"    1 = entry points of the whole batch come from the LDS-staged entry_scan_kernel (K0);
 *                     0 (default) = every query scans them inside the search kernel.  Same results bit for
 *                     bit; measured equally fast on MI355X (the shared scan rows are L2 hits either way)
 *   "visited_wide"    1 = always use the 32-bit open-addressing visited table (default 0: the 16-bit-tag
 *                     bucketed table whenever node-id width allows it) */
int fnv_set_option(fnv_index_t index, const char* name, int64_t value);

/* Thread safety: fnv_search_batch may be called concurrently on one index: the first caller uses the handle's own
 * stream / workspace / staging, a concurrent caller runs on a hidden lane (a view of the handle, created on first
 * contention: its copies and its launch overlap the first caller's -- round 4), up to seven of them; further callers
 * wait.  HBM: every lane owns a launch workspace like the handle's -- per resident query slot a visited bitmap of
 * capacity / 8 bytes, a 128 KB candidate spill area and (indexes beyond 4M nodes) a 64 KB overflow list; a full grid is
 * 0.6 GB at 1M nodes, 3.5 GB at 10M, 19 GB at 50M, a 1024-query batch at 50M nodes 6.4 GB.  A lane is only used while
 * the workspaces of all hidden lanes together stay within an eighth of the device's memory (FLATNAV_LANE_BUDGET_MB
 * overrides; 0 = no lanes): idle lanes give their workspace back when another lane or the handle itself needs the room,
 * a caller whose batch fits no lane waits for the handle (round 5).  Lanes run the kernel variant and LDS layout the
 * handle has measured and never take exploratory samples themselves.
 * What may race with a search: fnv_set_option -- it takes the handle's mutex, every launch sees the options either before
 * or after the change (the reference's only runtime knob, setNumThreads, Index.h:492-502, is not meant to be called during
 * a search either); fnv_last_launch_info / fnv_last_launch_geometry -- consistent records of ONE launch, the most recent
 * to complete on any lane; fnv_tune and fnv_index_insert_batch keep every lane out while they run.  NOT allowed while
 * searches are in flight: fnv_index_free, fnv_replica_refresh towards the handle.
 * fnv_search_batch_device shares one per-index workspace, so at most one such launch may be in flight per index
 * (launches on the same stream are naturally ordered).  Different indexes are independent.
 *
 * Batched search, host buffers.  queries: [nq][dim] elements of the index data type, C-contiguous.
 * out_dist/out_labels: [nq][K].  Rows with fewer than K reachable results are padded with
 * (+inf, -1) and reported through out_count[q] (nullable) -- the reference's binding raises
 * RuntimeError in that case (bindings.cpp:184-189); the host wrapper does the same.
 * out_ndist / out_nhops (nullable): per-query neighbour distance evaluations and expanded hops.
 * Copies: batches up to ~1 MB of queries + results go through one pinned staging buffer (one copy in, one out);
 * larger ones take one pageable copy in and ONE copy of the whole result slab out into pinned memory (round 4),
 * scattered to the five arrays by the CPU: 0.84-0.99 of the device-resident rate (PCIe-inclusive). */
int fnv_search_batch(fnv_index_t index, const void* queries, uint64_t nq, int K, int ef_search,
                     int num_initializations, float* out_dist, int32_t* out_labels, int32_t* out_count,
                     uint64_t* out_ndist, uint64_t* out_nhops);

/* Same, but every buffer already lives in this index's device memory and the work is enqueued on
 * `hip_stream` (a hipStream_t; NULL = the null stream) without synchronising. */
int fnv_search_batch_device(fnv_index_t index, const void* d_queries, uint64_t nq, int K, int ef_search,
                            int num_initializations, float* d_out_dist, int32_t* d_out_labels,
                            int32_t* d_out_count, uint64_t* d_out_ndist, uint64_t* d_out_nhops,
                            void* hip_stream);

/* ---- several GPUs of one node: index replicated, query rows sharded, no per-query collective -------
 * The reference parallelises a batch over host threads that share one index in memory (executeInParallel over rows,
 * python-bindings/src/flatnav/bindings.cpp:198-211, include/flatnav/util/Multithreading.h:19-48); here every GPU
 * holds a replica in its own HBM.
 * fnv_replicate: allocates an index of the same geometry and capacity on each of devices[0..n) (NULL = 0..n-1; a
 *   device may appear more than once and may be the source's own) and fills them from `src` with peer copies over
 *   xGMI, as a doubling tree (1 -> 2 -> 4 -> 8 holders).  The replicas are independent handles (own workspace and
 *   stream); free each with fnv_index_free.  One process drives all GPUs -- processes that own one GPU each
 *   (torch.distributed) broadcast the buffers of fnv_index_device_buffers with RCCL instead.
 * fnv_replica_refresh: copies the live rows of `src` into existing replicas again (after the source grew or was
 *   re-wired); replicas also take over the source's options (fnv_set_option) at every refresh and -- round 6 -- what the
 *   source has MEASURED (fnv_tune: kernel variant, LDS layout) when they sit on the same GPU model: tune the source, not
 *   every replica.  A source tuned after its replicas were made hands the measurements over at the next
 *   fnv_search_batch_multi whose indexes[0] it is (as long as its options are still the ones the replicas were given).
 * fnv_search_batch_multi: fnv_search_batch over several handles of the same index: rows [g*ceil(Q/G), ...) go to
 *   indexes[g]; every shard is driven by its own host thread (staging copies, launch and wait of all devices
 *   overlap), results land in the caller's row ranges.  The calling thread's current HIP device is left as it was
 *   (by every entry point of this library). */
int fnv_replicate(fnv_index_t src, int n_devices, const int* devices, fnv_index_t* out);
int fnv_replica_refresh(fnv_index_t src, int n_replicas, fnv_index_t* replicas);
int fnv_search_batch_multi(fnv_index_t* indexes, int n_indexes, const void* queries, uint64_t nq, int K,
                           int ef_search, int num_initializations, float* out_dist, int32_t* out_labels,
                           int32_t* out_count, uint64_t* out_ndist, uint64_t* out_nhops);

/* Wait for the most recent fnv_search_batch_device launch on this index and report whether any
 * query hit a capacity limit (FNV_ERR_CAPACITY); fnv_search_batch calls this itself. */
int fnv_search_status(fnv_index_t index);

/* Duration (ms, HIP events on the launch stream) of the search kernel of the most recent
 * fnv_search_batch[_device] call on this index; synchronises with that launch. */
int fnv_last_kernel_ms(fnv_index_t index, float* ms);

/* The merged-beam kernel searches a query again with the exact (libstdc++-replay) two-heap code when equal
 * distances meet at a decision.  out[5] = queries of the most recent search that were: {total, eviction tie,
 * selection tie, result tie, NaN/inf}; synchronises with that launch.  Results do not depend on it. */
int fnv_last_replayed_queries(fnv_index_t index, uint64_t out[5]);

/* Launch geometry of the most recent search: geom[8] = {grid_blocks, block_threads, lds_bytes,
 * blocks_per_cu, visited_slots, cand_slots (LDS entries of the exact search's candidates heap), kernel: 0 = two-heap
 * kernel, 1 = merged-beam kernel with the beam in registers, 2 = with the beam in LDS, tail_exact: the last that-many queries of the launch
 * went straight to the exact search (merged-beam kernel, see the "sorted_tail_exact_pct" option)}.
 * blocks_per_cu = the query slots a CU keeps resident = the grid's share per CU (round 5; gfx950 hands LDS out in 1280-byte
 * granules, so this is min(what the HIP occupancy API counts, 163840 / (ceil(lds_bytes / 1280) * 1280)); rounds 1-4 launched the
 * API's count). */
int fnv_last_launch_geometry(fnv_index_t index, uint64_t geom[8]);

/* The adaptive kernel choice ("sorted_beam" = 2) measures its variants on the caller's launches: up to 18 launches per
 * (beam width, batch-size class) run a variant that is being sampled, not the final pick.  fnv_tune takes all those
 * samples in ONE call -- every variant, a cold launch plus three timed ones, on the given batch (host pointer, or
 * device pointer with queries_on_device != 0; results are discarded) -- so that the first launch afterwards with
 * the same K / ef_search and a batch of the same class (more than one round of resident queries or not) already runs
 * the final variant.  Before that it measures the per-query LDS layout for this beam width: the rules' choice against
 * its neighbours (candidates heap of the exact search in LDS or in HBM, visited table one size up / down -- what the
 * "sorted_cand_lds" / "visited_slots" options would force; options the caller has set are respected) and keeps the
 * fastest until an option changes.  Results never depend on any of this.
 * No-op when there is nothing to choose (pinned variant, non-adaptive mode, batches < 2048).
 * There is no reference counterpart: the reference has one search routine (Index.h:606-707).
 * fnv_last_launch_info: info[4] = {variant of the most recent launch (numbering of "sorted_variant"), bit 0: that
 * launch was an exploratory sample of the adaptive choice, bit 1: it ran in shadow mode ("shadow_exact"), host
 * steady-clock ns at which the most recent
 * host-buffer search was enqueued, ... at which it completed (both 0 before the first one)}. */
int fnv_tune(fnv_index_t index, const void* queries, uint64_t nq, int queries_on_device, int K, int ef_search,
             int num_initializations);
int fnv_last_launch_info(fnv_index_t index, uint64_t info[4]);

/* The merged-beam kernel's hand-overs of the most recent launch (round 5).  A query in which equal distances meet at a decision
 * (fnv_last_replayed_queries counts them by reason) is not searched again from scratch: the reference's two heaps
 * (Index.h:618-619) are replayed from a log the merged-beam pass wrote -- per hop the expanded node and the neighbours that could
 * still be admitted -- and the exact search continues where that pass stopped; if the reference would have expanded another
 * node of equal distance first, from that hop.  out[0] = queries resumed from their log, out[1] = hops taken from the logs,
 * out[2] = hops the merged-beam passes of those queries had made, out[3] = queries searched again from scratch (no / an
 * overflowed log: options "tie_replay" = 0, "tie_log_entries"; NaN / infinite distances).  Same bytes either way. */
int fnv_last_handover_stats(fnv_index_t index, uint64_t out[4]);

/* What the handle (i = 0) and its hidden lanes (i = 1 ... 7; see fnv_search_batch) hold and did: info[2 i] = bytes of launch
 * workspace in HBM (visited bitmaps, overflow lists, candidate spill areas), info[2 i + 1] = launches that were exploratory
 * samples of the adaptive kernel choice (always 0 on a lane).  No reference counterpart (the reference's per-search state is
 * a VisitedSet from a pool, VisitedSetPool.h:16-50, and it has one search routine). */
int fnv_lane_info(fnv_index_t index, uint64_t info[16]);

/* Measurement aid (bench.py's roofline.gather_ceiling): GB/s of row bytes that a pure gather of random rows of this
 * index's vector table reaches with the search kernel's own load pattern for this row width (lane groups, loads in
 * flight) and `waves_per_cu` resident wavefronts per CU (0 = 16) -- the practical bound of the algorithmic rate for
 * this (row bytes, table size): small tables are partly served by the 256 MiB Infinity Cache, rows that are not whole
 * 128-byte lines pay for the lines they straddle.  About 20-40 ms of GPU time; no reference counterpart. */
int fnv_gather_ceiling(fnv_index_t index, int waves_per_cu, double* gbps_out);

#ifdef __cplusplus
}
#endif
#endif /* FLATNAV_HIP_H */
