/*
 * ndbhip.h — C ABI of the MI355X (gfx950) vector-distance engine that sits
 * behind NeuronDB's index access methods.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, int status
 * codes, no PostgreSQL and no torch types.  Each entry point names the
 * reference interface it replaces (paths relative to NeuronDB/ in the
 * reference tree).  The reference-side binding a maintainer would add
 * (ivf_am.c / hnsw_am.c calling these) is shown in INTEGRATION.md.
 *
 * Conventions (same as the reference's GPU vtable,
 * include/neurondb_gpu_backend.h:24-26 and src/gpu/common/gpu_distance.c:50-51):
 *   - every function returns 0 on success and a negative NDBHIP_ERR_* code on
 *     failure; nothing longjmps or throws across this boundary;
 *   - ndbhip_last_error() returns a thread-local message for the last failure;
 *   - inputs are copied (or consumed) before a function returns: the library
 *     never keeps a pointer into palloc'd or buffer-page memory;
 *   - there is NO CPU fallback inside the library: without a usable HIP device
 *     every compute entry point fails with NDBHIP_ERR_NODEVICE and the caller
 *     decides (neurondb.compute_mode, src/gpu/common/gpu_core.c:268-284).
 *   - HIP is initialised lazily by ndbhip_init() in the calling process, never
 *     at library load (PostgreSQL loads the library pre-fork:
 *     src/worker/worker_init.c:77-84).
 *   - the library serves ONE thread per process (a PostgreSQL backend is single-
 *     threaded): the runtime context, the mode switches (ndbhip_set_scan_mode,
 *     ndbhip_hnsw_set_*_mode) and a mirror's scratch buffers are process-wide /
 *     per-mirror state without locks.  Several backends = several processes, each
 *     with its own context and mirrors.
 *   - a failing call releases what it had built only as far as cheap; after
 *     NDBHIP_ERR_HIP / NDBHIP_ERR_NOMEM destroy the mirror and rebuild it.
 *
 * Heap TIDs cross the boundary as PostgreSQL ItemPointerData images
 * (6 bytes: bi_hi, bi_lo, ip_posid, each little-endian uint16).  On the device
 * they are held as uint64 = bi_hi | bi_lo << 16 | ip_posid << 32.
 */
#ifndef NDBHIP_H
#define NDBHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDBHIP_ABI_VERSION 1

/* status codes */
#define NDBHIP_OK               0
#define NDBHIP_ERR_INVALID     (-1)	/* bad argument (reference: ereport(ERROR) on bad params) */
#define NDBHIP_ERR_NODEVICE    (-2)	/* no HIP device / not initialised */
#define NDBHIP_ERR_HIP         (-3)	/* a HIP runtime call failed */
#define NDBHIP_ERR_NOMEM       (-4)
#define NDBHIP_ERR_STATE       (-5)	/* index not loaded / wrong lifecycle order */
#define NDBHIP_ERR_UNSUPPORTED (-6)	/* e.g. hnsw strategy not in {1,2,3}: hnsw_am.c:1339-1343 */

/* ordering-operator strategy numbers as the AMs switch on them
 * (src/index/ivf_am.c:1559-1591, src/index/hnsw_am.c:1310-1344) */
#define NDBHIP_STRATEGY_L2      1
#define NDBHIP_STRATEGY_COSINE  2
#define NDBHIP_STRATEGY_IP      3	/* hnsw: in the reference; ivf: new surface (quirk Q2) */

#define NDBHIP_INVALID_BLOCK 0xFFFFFFFFu
#define NDBHIP_HNSW_MAX_LEVEL 16	/* src/index/hnsw_am.c:85 */
#define NDBHIP_MAX_K       1024		/* neurondb.hnsw_k upper bound is 1000: src/util/neurondb_guc.c:174 */
#define NDBHIP_MAX_NPROBE  1024		/* neurondb.ivf_probes upper bound is 1000: src/util/neurondb_guc.c:187 */
#define NDBHIP_MAX_EF      1024

typedef struct ndbhip_ivf ndbhip_ivf;	/* device-resident mirror of one ivf index */
typedef struct ndbhip_hnsw ndbhip_hnsw;	/* device-resident mirror of one hnsw index */

/* ------------------------------------------------------------------ */
/* Runtime (replaces ndb_gpu_backend.init/shutdown/device_count/set_device/
 * stream_*: include/neurondb_gpu_backend.h:28-65; lazy init pattern of
 * ndb_gpu_init_if_needed: src/gpu/common/gpu_core.c:240-310)           */
/* ------------------------------------------------------------------ */
int			ndbhip_abi_version(void);
int			ndbhip_device_count(void);		/* >= 0, or NDBHIP_ERR_NODEVICE; does not create a context */
int			ndbhip_init(int device);		/* idempotent per process */
int			ndbhip_shutdown(void);
const char *ndbhip_last_error(void);
/* Run all work on this hipStream_t.  NULL = the library's own stream, which is NON-BLOCKING: it does not
 * synchronise with the legacy default stream (handle 0) either — a caller that produces or consumes device
 * buffers on another stream passes that stream here (neurondb_amd/_lib.py: use_torch_stream). */
int			ndbhip_set_stream(void *hip_stream);
/* The CALLING THREAD's stream: every entry point this thread calls afterwards launches on it instead of the process-wide
 * one (NULL: back to that).  Two host threads, each with a stream and a mirror of its own, keep two batches in flight:
 * the per-query chains of one batch (round trips to memory) run under the other batch's sweep (profiles/r05_overlap.txt).
 * Handles are not shared between threads that search at the same time; ndbhip_ivf_share / ndbhip_hnsw_share and the
 * destroy of a shared handle are made while no other thread searches on the source (a share copies the source's
 * tables as they stand).
 * GPU_MAX_HW_QUEUES: the HIP runtime maps a process's streams onto 4 hardware queues unless the environment says
 * otherwise WHEN THE RUNTIME INITIALISES — two streams that land on one queue run one after the other and nothing
 * overlaps.  A process that keeps batches in flight on streams of its own exports GPU_MAX_HW_QUEUES=8 before its first
 * HIP call (bench.py sets it for itself; under rocprofv3 export it in the shell: tools/round_profiles.sh). */
int			ndbhip_set_thread_stream(void *hip_stream);
int			ndbhip_get_stream(void **out_hip_stream);	/* the stream every asynchronous entry point is ordered on */
int			ndbhip_synchronize(void);

/* Per-process counters (replaces GPUStats, include/neurondb_gpu.h:34-42). */
typedef struct ndbhip_stats
{
	uint64_t	queries;			/* queries searched */
	uint64_t	rows_scored;		/* distance evaluations on the device */
	uint64_t	bytes_scored;		/* algorithmic bytes = rows_scored * dim * elem size */
	uint64_t	scan_launches;		/* launches of the dominant kernel (list scan / hnsw walk) */
	double		scan_kernel_ms;		/* HIP-event time of those launches (only while profiling is on) */
	uint64_t	rows_rescored;		/* screened scan: candidates given the reference's arithmetic in the second pass */
	uint64_t	rows_emitted;		/* fp16 matrix-core screen: candidates the bound pass could not exclude */
	uint64_t	screen16_batches;	/* sub-batches served by the fp16 matrix-core screen */
	uint64_t	screen16_fallbacks;	/* ... that overflowed a query's record capacity and were rerun on the fp32 screen */
	uint64_t	pairs_pruned;		/* (query, list) pairs excluded by |q - centroid| - list radius before the sweep */
	uint64_t	rows_swept;			/* candidate rows of the pairs the sweep did multiply (<= rows_scored) */
	uint64_t	plane_bytes;		/* bytes of row planes the sweeps had to read: every touched 128-row tile once per batch */
	uint64_t	cent_screen_batches;	/* sub-batches whose centroid scan ran on the matrix cores (k_cent_select) */
	uint64_t	prepares;			/* full layouts of the sweep's operands (planes, sublists, radii): once per mirror unless ... */
	uint64_t	prepare_updates;	/* ... appends / deletes were folded into the existing layout instead (rows added in spare blocks,
									 * deleted rows left as holes and the survivors renumbered) */
	uint64_t	dense_sweeps;		/* sweeps that ran the dense tile's kernel (256 pairs x 256 rows: buckets probed by hundreds of
									 * queries — a table without cluster structure; csrc/ndbhip_screen16d.h) */
	uint64_t	wave_sweeps;		/* sweeps that ran as wave-autonomous register streams (k_s16c_wsweep, csrc/ndbhip_screen16w.h: sparse pair
								 * tables — a bucket probed by a handful of queries) */
	uint64_t	sub_restricted;		/* batches that scored the regrouped lists' centres for the PROBED lists only (round 6, k_subdist_lists:
								 * tables with tens of thousands of sublists) instead of multiplying every query by every centre */
}			ndbhip_stats;
int			ndbhip_stats_get(ndbhip_stats *out);
int			ndbhip_stats_reset(void);
int			ndbhip_profile(int on);			/* bracket the dominant kernel with HIP events */
/* List-scan kernel choice (results are bit-identical either way): 0 = auto
 * (query-grouped scan for batches of >= 5 queries when dim % 64 == 0, per-query
 * scan otherwise; batches of >= 128 queries over float4 rows are screened — L2, inner product, cosine), 1 = always per-query,
 * 2 = always grouped, 3 = grouped and screened whenever the recipe allows, 4 = grouped, never screened.
 * Screened = a fused-multiply-add pass bounds every candidate's distance from below, and only the candidates
 * that can still be among the k nearest get the reference's own arithmetic. */
int			ndbhip_set_scan_mode(int mode);
/* ... 5 = screened by the fp16 matrix-core pass whenever it applies (L2 / inner product, k <= 64, float4 or
 * halfvec rows, any dim), which is also what auto mode picks for batches of >= 128 queries: the bound pass runs
 * as a query-tile x row-tile contraction on v_mfma_f32_32x32x16_f16 over rows and queries split into two fp16
 * planes, emits the candidates it cannot exclude, and the reference's arithmetic decides among those
 * (csrc/ndbhip_screen16.h; the error term is derived in csrc/ndbhip_common.h).
 * Rows or queries holding NaN / infinity (or sums beyond fp32) are outside the parity contract — the index paths
 * of the reference never test for them (ivf_am.c:1550-1592) and x86 / gfx950 do not even agree on the NaN they
 * produce — but they only affect the candidates they take part in: such a row is always handed to the
 * reference's arithmetic, never allowed to distort another row's bound. */

/* Process-wide switches (round 1 read some of them from the environment; nothing in the library reads the
 * environment any more).  Results are bit-identical whichever way they are set.
 *   "screen"            1   auto mode screens batches of >= 128 queries (0 = never)
 *   "screen16"          1   ... on the fp16 matrix cores (0 = the fp32 bound pass)
 *   "screen16_records"  8192  candidates a query may emit before it is swept again / its batch falls back
 *   "screen16_prune"    1     L2: a (query, list) pair whose |q - centroid| - list radius already exceeds the query's threshold is not swept
 *   "screen16_sublists" 1     L2 and inner product, float4 and fp16 rows: lists longer than "screen16_sub_min" (256) rows are regrouped, inside the library's
 *                             own copy of the rows, into sublists of about "screen16_sub_rows" (128) rows — kept per list only
 *                             where that shrinks the radius — and a (query, probe) pair expands only to the sublists the
 *                             triangle inequality cannot exclude (csrc/ndbhip_screen16.h, "Sublists")
 *   "screen16_tighten"  1     a query's threshold is lowered inside the sweep every 128 emitted records (0: only between the two rounds)
 *   "cent_screen16"     1     screened batches, 256 .. 4096 centroids: |q - centroid|^2 of every pair from the matrix-core sweep, the
 *                             reference's arithmetic only for the centroids near the nprobe-th (k_cent_select); 0: the exact centroid scan
 *   "screen16_centered" 1     L2, float4 rows: the sweep multiplies (query - centre) with (row - centre) of the row's list / sublist,
 *                             ONE fp16 plane of each (2 bytes per row element instead of 4); the error term scales with the distances
 *                             to the centre instead of the vectors' norms (csrc/ndbhip_screen16c.h, csrc/ndbhip_common.h (8));
 *                             0: the two-plane sweep over the rows as they are
 *   "screen16c_qb"      0     (query, list) pairs per tile of the centred sweep / 32: 8 (256 pairs x 256 rows, 8 waves) | 4 | 1, 0 = chosen from the previous batch's pairs per list
 *   "screen16c_seeds"   0     rows per query whose upper bounds make its first threshold: 32 | 64, 0 = 32 for k <= 20, else 64
 *   "build_prepare"     0     ndbhip_ivf_build / _build_device end with ndbhip_ivf_prepare(ix, this strategy 1 .. 3): the index leaves the
 *                             build searchable at full speed (0: the first batched scan, or an explicit ndbhip_ivf_prepare, pays for it)
 *   "hnsw_intended_waves" 16  waves per CU walking the intended HNSW (build and search); each owns a visited bitmap of one bit per node
 *   "hnsw_intended_host_groups" 0  1 = the back-links of an intended build batch are grouped by target on the host (rounds 3-4) instead of on the device
 *   "hnsw_intended_occ4"   0  the intended search held to 128 registers (four walkers a SIMD): 1 = where that costs no scratch (walk rows, dim <= 768), 2 = everywhere, 0 = nowhere
 *   "screen16_cosine"   1     cosine batches run the matrix-core sweep over NORMALISED planes (rows and queries divided by their
 *                             norms; sublists regrouped in that space); 0: the round-1 fp32 screen
 *   "screen16_cosine_centered" 1  ... as the centred L2 sweep (|q^ - x^|^2 = 2 x cosine distance); 0: as the inner product of
 *                             two-plane normalised rows
 *   "screen16_ip_centered" 1  inner-product batches run the centred one-plane sweep over the L2 layout's planes (b = |q - x|^2 + M^2 - |x|^2
 *                             orders a query's rows like -q.x; fp32 and fp16 mirrors); 0: the two-plane sweep of round 2
 *   "screen16_redo"     1     queries whose records / survivors overflow go to the exact path alone (0: their whole batch does)
 *   "screen16_slack"    1     the centred planes keep spare 32-row blocks per bucket and take appends in place (0: every append lays them out again)
 *   "screen_min_nq"     5     batches of at least this many queries take the screened (matrix-core) scan; smaller ones the exact scans
 *                             (measured crossover, profiles/r04_small_batch.txt; the first such batch on an index pays its preparation once)
 *   "screen16_stage"    1     k_s16_finalize's survivors and k_cent_select's candidate centroids get the reference's sequential sum from rows
 *                             streamed through LDS by DMA (s16_exact_staged: a chunk of 256 bytes per row, several chunks ahead) instead of
 *                             rows loaded 16 bytes at a time by the lane that sums them; 0 = off, 2..13 = that ring depth (1: by batch size)
 *   "screen16c_nbuf"    0     ring depth of the centred sweep: 2 | 3, 0 = the tile geometry's default
 *   "screen16c_dense"   1     the dense tile (256 pairs x 256 rows) runs k_s16c_dense (csrc/ndbhip_screen16d.h: loader and prefetcher
 *                             waves, chunk-major pair planes, the matrix pipe screens its own accumulator blocks, queued records);
 *                             0 = k_s16c_sweep<8, 2> (A/B)
 *   "screen16c_bigk"    1     64 < k <= 256 on the centred fp16 screen (L2, sublists): thresholds from the buckets' radii (k_s16c_thr_radius); 0: the fp32 screen serves k > 64
 *   "screen16c_sample"  2048  rows of the mirror sampled for the first thresholds of a batch on a table without cluster structure
 *                             (k_s16c_seed_sample: all queries x the sample as one matrix on the matrix cores); 0 = seeds only, 256..2048
 *   "screen16c_dense_min" 24, "screen16c_dense_min_sub" 100  the dense tile's kernel from this many (query, bucket) pairs a bucket in the
 *                             previous batch on this mirror: whole lists / layouts with sublists (0 = never); below: the 128-pair ring,
 *                             below 24 the 32-pair tile (sparse tables)
 *   "screen16c_wave"    2     the 32-pair tile as wave-autonomous register streams (k_s16c_wsweep): chunks in flight a wave, 2 | 3 | 4; 0 = the LDS ring
 *   "screen16c_wave_blocks" 2 blocks of k_s16c_wsweep a compute unit (1 .. 3), "screen16c_wave_min_nq" 1024: batches below take the ring
 *   "screen16c_plane_seeds" 1 L2 / inner product: a query's first threshold from block 0 of the nearest sublist's planes (0: from float4 rows)
 *   "screen16_sweep_queue" 1  sweeps of different streams (steps in flight) take turns on the device; 0 = launched as they come
 *   "screen16_sub_restrict" 0 from this many regrouped-list centres on a batch scores the centres of its PROBED lists only (0 = never)
 *   "build_single_sweep" 1    build assignment: row minima and candidate records in ONE matrix sweep + a resolve kernel (0: two sweeps)
 *   "kmeans_screen16"   1     k-means iterations assign on the matrix cores like the build does (0: exact kernels)
 *   "slow_call_log"     0     a search call slower than this many microseconds reports its host-side phases on stderr
 *   "screen16c_dense_split" 3  k_s16c_dense: 32-row blocks (of a tile's eight) a loader wave multiplies, its SIMD's multiplier the rest (3 | 4)
 *   "screen16c_dense_sync" 16  k_s16c_dense on tables of whole full tiles: the blocks of an XCD meet before every this many-th item, so
 *                             that the blocks sharing an operand tile ask for its chunks while the XCD's L2 still has them (0 = never)
 *   "screen16c_dense_small" 1  tables below 400 pairs a bucket: the kernel whose tiles of <= 128 members run with one pair block a wave
 *   "screen16c_dense_spare" 0  compute units k_s16c_dense leaves to other steps' kernels on a mirror with shares (measured: no gain)
 *   "screen16c_tight"   128   k_s16c_dense tightens a query's threshold every this many records (power of two, 8..1024)
 *   "screen16c_pfd"     0     chunks k_s16c_dense's prefetcher waves touch ahead of its loaders (0 = none: measured slower on MI355X), 0..10
 *   "screen16c_rot"     0     k_s16c_dense takes an item's chunks in an order rotated by its row tile (1) / pair tile (2); 0 = in order
 *   "screen16c_epi"     1     k_s16c_sweep: the matrix pipe screens accumulator blocks before the per-element test; 0 = every element (A/B)
 *   "screen16c_pf"      0     timing variants of k_s16c_sweep<8, 2>: 3 = in-wave L2 prefetch, 16 / 32 / 48 = non-temporal rows / pairs / both
 *   "build_screen16"    1     build / ndbhip_ivf_assign_device (>= 4096 rows): the assignment is screened on the matrix cores (0: exact kernels)
 *   "block_cache"       1     keep up to 4 freed packed-row blocks (>= 64 MiB) for the next build (0: release them now, stop caching)
 *   "screen16_waves"    4   tile geometry of the fp16 sweep: 4 waves, 128 x 128, ring of 2 (measured faster) | 8 waves, 256 x 128, ring of 3
 *   "screen16_debug"    0   timing experiments of the sweep (1 no DMA, 2 DMA of cache-hot lines; 4 / 8 / 16: k_sub_pairs without its counter atomics / centre distances / table reads — all WRONG results)
 *   "scr_coop" 2, "scr_ch" 16, "scr_mfma" 1, "gchunk" 32   A/B switches of the fp32 screened / grouped kernels (docs/DESIGN_rounds_1_3.md 3b, 3c)
 *   "screen16_fin_threads" 64 threads of a k_s16_finalize block (one block per query; 64 / 128 / 256), "probe_select_threads" 256,
 *   "probe_select_radix" 0    launch shapes kept for A/B (docs/DESIGN_rounds_1_3.md 3e has the measurements)
 *   "debug_s16", "debug_build", "hnsw_trace"  0   progress / timing lines on stderr
 *   "hnsw_nofast"       0   hnsw build walks score rows in the reference's own summation order only */
int			ndbhip_set_option(const char *name, int value);

/* Synthetic data for benches and full-size tests (a counter-based generator in the repo): element
 * (row, d) is a pure function of the seeds — any slice, any order — and ndbhip_gen_rows_host (pure C, needs no
 * device: what an oracle run is fed) returns the SAME BITS as ndbhip_gen_rows_device (csrc/ndbhip_gen.h says why).
 * kind 0: i.i.d. N(0,1); kind 1: mixture of `components` Gaussians (sigma) around centers ~ N(0,1) drawn from
 * center_seed.  out: [nrows][dim] floats, rows first_row .. first_row + nrows. */
int			ndbhip_gen_rows_device(int kind, uint64_t seed, uint64_t center_seed, int64_t first_row, int64_t nrows,
								   int dim, int components, float sigma, float *d_out);
int			ndbhip_gen_rows_host(int kind, uint64_t seed, uint64_t center_seed, int64_t first_row, int64_t nrows,
								 int dim, int components, float sigma, float *out);

/* The matrix-core instruction the bound pass rests on, in isolation, so that its accumulation-error model
 * (csrc/ndbhip_common.h (4)) is checked on the part the library runs on (tests/test_gpu_mfma_model.py):
 * per tile t, D = C + chain x (A.B) with A [ntiles][32][16], B [ntiles][16][32] fp16 bit patterns and
 * C, D [ntiles][32][32] floats, all device pointers; asynchronous on the library's stream. */
int			ndbhip_mfma_probe(const uint16_t *d_a, const uint16_t *d_b, const float *d_c, float *d_d,
							  int ntiles, int chain);
/* The fp32 matrix instruction that screens the centred sweep's accumulator blocks (v_mfma_f32_32x32x2_f32,
 * csrc/ndbhip_screen16c.h "pass 0"): per tile t, D = C + A.B with A [ntiles][32][2], B [ntiles][2][32],
 * C, D [ntiles][32][32] floats. */
int			ndbhip_mfma_probe_f32(const float *d_a, const float *d_b, const float *d_c, float *d_d, int ntiles);
/* Profiling builds of the library (make PHASES=1: -DNDB_PHASES) stamp a 100 MHz clock at marked places of the per-batch
 * kernels (block 0 only); this copies the 64 stamps to out (zeros from an ordinary build).  tools/phase_probe.py */
int			ndbhip_debug_phases(unsigned long long *out);
/* profiling builds: per-wave trace of the last register-streaming sweep (4 words a wave: first request, end — 100 MHz
 * clock —, items, ticks inside the stream's waits), n words; zeros in a release build */
int			ndbhip_debug_trace(unsigned long long *out, int n);
int			ndbhip_debug_h2_phases(unsigned long long *out);	/* [8]: the intended HNSW search's phase clocks (csrc/ndbhip_hnsw2.h), read and reset */

/* ------------------------------------------------------------------ */
/* IVF mirror lifecycle.  Replaces the page walk of ivfSelectClusters /
 * ivfCollectCandidates (src/index/ivf_am.c:1597-1909): centroids page →
 * `centroids`, list page chains → one packed row block per list, in chain
 * order, dead / dim-mismatched items already dropped by the packer.       */
/* ------------------------------------------------------------------ */
int			ndbhip_ivf_create(int dim, int nlists, ndbhip_ivf **out);
int			ndbhip_ivf_destroy(ndbhip_ivf *ix);

/* centroids: host, row-major [ncentroids * dim] (ncentroids = items on the
 * centroid page(s), "maxoff" in ivf_am.c:1646-1652; may be < nlists). */
int			ndbhip_ivf_set_centroids(ndbhip_ivf *ix, const float *centroids, int ncentroids);

/*
 * Load every list in one call (host pointers).
 *   list_len[ncentroids]  GLOBAL number of live entries of each list
 *   owned[ncentroids]     nullable; owned[L]==0 → this process holds no rows of
 *                         list L (multi-GPU sharding, one process per GPU)
 *   rows                  the owned lists' vectors, list-major, chain order
 *   tids6                 matching ItemPointerData images, 6 bytes per row
 *   nrows                 number of rows in `rows`/`tids6` (= sum of owned list_len)
 */
int			ndbhip_ivf_load(ndbhip_ivf *ix, const int64_t *list_len, const uint8_t *owned,
							const float *rows, const uint8_t *tids6, int64_t nrows);
/* halfvec column (src/types/quantization.c, VectorF16): the rows as IEEE fp16 images, kept as fp16 in
 * HBM (half the bytes per row) and decoded on the fly exactly like the reference's fp16_to_float
 * (quantization.c:170-218, incl. its subnormal quirk) — results are bit-identical to indexing the expanded
 * float4 values, which is what the reference stores (hnsw_am.c:1436-1451 / ivf_am.c:168-177).
 * Needs dim % 64 == 0.  Queries stay float4 (ndbhip_extract_vector expands a halfvec query). */
int			ndbhip_ivf_load_f16(ndbhip_ivf *ix, const int64_t *list_len, const uint8_t *owned,
								const uint16_t *rows_f16, const uint8_t *tids6, int64_t nrows);
/* Same, rows/tids already in HBM (d_tids as uint64 device format). The arrays
 * are adopted without a copy and must outlive the index. */
int			ndbhip_ivf_load_device(ndbhip_ivf *ix, const int64_t *list_len, const uint8_t *owned,
								   const float *d_rows, const uint64_t *d_tids, int64_t nrows);
/* aminsert: append one entry to the tail of list `list_id`
 * (src/index/ivf_am.c:954-1157) — see ndbhip_ivf_assign for the list choice. */
int			ndbhip_ivf_append(ndbhip_ivf *ix, int list_id, const float *vec, const uint8_t *tid6);
/* ivfinsert (src/index/ivf_am.c:797-1167) for one host row: nearest centroid by the insert-time rule
 * (:905-935) on the device, then ndbhip_ivf_append to that list; *list_out (optional) = the list chosen. */
int			ndbhip_ivf_insert(ndbhip_ivf *ix, const float *vec, const uint8_t *tid6, int *list_out);
/* ambulkdelete (src/index/ivf_am.c:1172-1357): every entry whose heapPtr is one of the n given TIDs is
 * dropped from the mirror (the reference marks its line pointer dead and scans skip it, :1816-1822);
 * survivors keep their list and their order inside it.  *removed = entries dropped.  Runs on the device
 * (mark by binary search, prefix scan, move).  Not for sharded mirrors: delete on the full mirror and shard
 * again, the other ranks' candidate positions change with it. */
int			ndbhip_ivf_delete(ndbhip_ivf *ix, const uint8_t *tids6, int64_t n, int64_t *removed);

/* Read the mirror back to the host (any pointer may be NULL): centroids
 * [ncentroids*dim], list_len [ncentroids] (owned lists only), rows/tids6 of
 * the resident rows.  Used by the page writer of ambuild and by tests. */
int			ndbhip_ivf_export(const ndbhip_ivf *ix, float *centroids, int64_t *list_len, float *rows,
							  uint8_t *tids6);
int			ndbhip_ivf_ncentroids(const ndbhip_ivf *ix);
int			ndbhip_ivf_dim(const ndbhip_ivf *ix);
int			ndbhip_ivf_shape(const ndbhip_ivf *ix, int *dim, int *nlists);
/* IvfMetaPageData.nprobe (ivf_am.c:75-89): set from the pages by ndbhip_ivf_load_pages, default 10; what
 * ivfrescan uses (:1487-1513) */
int			ndbhip_ivf_get_nprobe(const ndbhip_ivf *ix, int *nprobe);
int			ndbhip_ivf_set_nprobe(ndbhip_ivf *ix, int nprobe);

/* ------------------------------------------------------------------ */
/* Index pages <-> mirror.  PostgreSQL-free codec of the ivf
 * relation's 8 KB pages, layouts as in src/index/ivf_am.c:62-106, 241-256,
 * 640-711, 954-1157 (meta page, centroid items, list page chains).  `pages` is
 * the relation image, block b at pages + b*8192.  The reader follows exactly the
 * walk of ivfSelectClusters / ivfCollectCandidates: centroid items in offset
 * order, each list's chain via IvfListPageHeader.nextBlock, dead line pointers
 * and entries with a foreign dim skipped.  Format version 2 chains several
 * centroid pages (the reference fits only 2 centroids at dim 768 on its single
 * page: quirk Q6); single-page indexes are written as version 1.  Host code.  */
/* ------------------------------------------------------------------ */
int			ndbhip_ivf_pages_info(const uint8_t *pages, uint32_t nblocks, int *dim, int *nlists,
								  int *ncentroids, int64_t *live_rows, int *version);
int			ndbhip_ivf_pages_unpack(const uint8_t *pages, uint32_t nblocks, float *centroids,
									int64_t *list_len, float *rows, uint8_t *tids6);
/* pages -> new device mirror (the packer behind ambeginscan's mirror cache) */
int			ndbhip_ivf_load_pages(ndbhip_ivf **out, const uint8_t *pages, uint32_t nblocks);
int64_t		ndbhip_ivf_pages_needed(int dim, int ncentroids, const int64_t *list_len);
int			ndbhip_ivf_pages_pack(int dim, int nlists, int nprobe, int ncentroids, const float *centroids,
								  const int64_t *list_len, const float *rows, const uint8_t *tids6,
								  uint8_t *pages, uint32_t nblocks_cap, uint32_t *nblocks_out);
/* device mirror -> pages (ambuild after ndbhip_ivf_build_device) */
int			ndbhip_ivf_write_pages(const ndbhip_ivf *ix, int nprobe, uint8_t *pages, uint32_t nblocks_cap,
								   uint32_t *nblocks_out);

/* Multi-GPU: a new mirror holding only the lists with owned[L] != 0 (one process
 * per GPU keeps its share); list lengths stay global, so candidate positions —
 * and therefore the merged result — are identical to the unsharded index. */
int			ndbhip_ivf_shard(const ndbhip_ivf *src, const uint8_t *owned, ndbhip_ivf **out);
/* The same for slices: the new mirror holds positions [lo[c], lo[c] + len[c]) of every list c (len 0 = none).
 * Slices let several ranks share one long, popular list — a list-granular shard cannot scale past the work of
 * its heaviest list; candidates keep their positions in the reference's candidates[], so the merged result is
 * the unsharded one.
 * tail[c] != 0 marks the one rank that takes later appends to list c (ndbhip_ivf_append lands on the list's
 * tail page); NULL = the rank holding the list's last row, and nobody for an empty list. */
int			ndbhip_ivf_shard_slices(const ndbhip_ivf *src, const int64_t *lo, const int64_t *len, const uint8_t *tail,
									ndbhip_ivf **out);
/* A second HANDLE on the same mirror (round 5): rows, TIDs, planes, sublists and matrices — everything a search reads —
 * are the source's own arrays; everything a batch writes (per-batch scratch, the pinned result block) is the new
 * handle's.  What several batches in flight need — a host thread, a stream (ndbhip_set_thread_stream) and a handle each —
 * without a second copy of a mirror that is 1.55 x its table.  While a share lives both handles are FROZEN: loads,
 * appends, deletes, builds and anything that would lay the planes out again return NDBHIP_ERR_STATE, so run
 * ndbhip_ivf_prepare (or one batch of every kind the shares will serve) on the source first; ndbhip_ivf_destroy of the
 * source is refused until its shares are destroyed.
 * Threading: ndbhip_ivf_share copies the source's handle, and ndbhip_ivf_destroy of a share updates the source's count of
 * shares: call both while NO search is running on the source (the bench and neurondb_amd/ivf.py create all shares before
 * the lanes start and close them after the lanes have joined).  Searches on different handles of one mirror may run at the
 * same time from different threads; searches on ONE handle may not. */
int			ndbhip_ivf_share(ndbhip_ivf *src, ndbhip_ivf **out);
/* A halfvec twin of a float4 mirror (same centroids, lists, TIDs): rows narrowed on the device with the
 * reference's own encoder float4_to_fp16 (src/types/quantization.c:141-168: mantissa truncated, subnormal
 * results flushed to zero) when reference_encoder != 0 — what a halfvec column cast by the reference holds —
 * else round-to-nearest-even. */
int			ndbhip_ivf_to_f16(const ndbhip_ivf *src, int reference_encoder, ndbhip_ivf **out);
int64_t		ndbhip_ivf_nrows(const ndbhip_ivf *ix);			/* rows resident on this device */
int64_t		ndbhip_ivf_max_candidates(const ndbhip_ivf *ix, int nprobe);	/* sum of the nprobe longest lists */

/* ------------------------------------------------------------------ */
/* IVF search = ivfgettuple's first-call work (src/index/ivf_am.c:1976-1999:
 * ivfSelectClusters + ivfCollectCandidates) for nq queries at once.
 *   strategy        1 L2, 2 cosine, 3 -IP (new), anything else L2 (:1583)
 *   nprobe          so->nprobe (the reference pins it to 10: quirk Q4)
 *   k               so->k (the reference pins it to 10: quirk Q3)
 *   max_candidates  k*10 reproduces the reference's cap (:1743); <= 0 = no cap
 * Results per query q: out_count[q] <= k entries at out_*[q*k ...], in the
 * exact order the reference's selection sort returns them.                 */
/* ------------------------------------------------------------------ */
int			ndbhip_ivf_search(ndbhip_ivf *ix, const float *queries, int nq, int strategy,
							  int nprobe, int k, int64_t max_candidates,
							  uint8_t *out_tids6, float *out_dist, int *out_count);
/* Device-pointer form (inputs and outputs in HBM, asynchronous on the stream). */
/* ndbhip_ivf_search for queries scattered in host memory the device can read (hipHostRegister'd / hipHostMalloc'd):
 * query i = the dim floats at d_base + offsets[i] (bytes; d_base = the DEVICE pointer of that memory, offsets a host
 * array).  A kernel gathers them over PCIe: no CPU copy of the queries.  The device-owner service serves its request
 * ring this way (csrc/ndb_service.cpp). */
int			ndbhip_ivf_search_mapped(ndbhip_ivf *ix, const void *d_base, const int64_t *offsets, int nq, int strategy,
									 int nprobe, int k, int64_t max_candidates, uint8_t *out_tids6, float *out_dist,
									 int *out_count);
int			ndbhip_ivf_search_device(ndbhip_ivf *ix, const float *d_queries, int nq, int strategy,
									 int nprobe, int k, int64_t max_candidates,
									 uint64_t *d_out_tids, float *d_out_dist, int *d_out_count);

/* Centroid selection only (ivfSelectClusters, :1597-1717): out_probes[nq*nprobe]. */
int			ndbhip_ivf_select_clusters(ndbhip_ivf *ix, const float *queries, int nq, int nprobe,
									   int *out_probes);

/*
 * Sharded search (one process per GPU).  Each rank scans only the lists it
 * owns and emits, per query, the candidates that can still reach the global
 * top-k: fixed-size records {key, pos, tid}, NDBHIP_PARTIAL_CAP(k) per query.
 * After an all-gather of the records, ndbhip_merge_topk_* replays the
 * reference's selection sort on the union, so the merged result equals the
 * single-process result entry for entry.
 */
typedef struct ndbhip_cand
{
	uint32_t	key;			/* order-preserving image of the float4 distance */
	uint32_t	pos;			/* index in the reference's candidates[] array */
	uint64_t	tid;
}			ndbhip_cand;
#define NDBHIP_PARTIAL_CAP(k) (3 * (k))
int			ndbhip_ivf_search_partial_device(ndbhip_ivf *ix, const float *d_queries, int nq, int strategy,
											 int nprobe, int k, int64_t max_candidates,
											 ndbhip_cand *d_out_cand, int *d_out_ncand, int64_t *d_out_total);
/* The two halves of ndbhip_ivf_search_partial_device, so that the ranks of a sharded search can split the
 * centroid scan + ivfSelectClusters (ivf_am.c:1597-1717) by QUERIES instead of all repeating it: each rank
 * selects for its slice, the probe lists ([nq][nprobe] int32, same values ndbhip_ivf_select_clusters
 * returns) are all-gathered, and every rank scans its own lists for all queries. */
int			ndbhip_ivf_select_clusters_device(ndbhip_ivf *ix, const float *d_queries, int nq, int nprobe,
											  int *d_out_probes);
int			ndbhip_ivf_search_partial_probes_device(ndbhip_ivf *ix, const float *d_queries, int nq, int strategy,
													int nprobe, int k, int64_t max_candidates,
													const int *d_probes, ndbhip_cand *d_out_cand,
													int *d_out_ncand, int64_t *d_out_total);
/* d_cand: [world][nq][cap] records, d_ncand: [world][nq], d_total: [nq]
 * (number of candidates over all ranks, identical on every rank). */
int			ndbhip_merge_topk_device(const ndbhip_cand *d_cand, const int *d_ncand, const int64_t *d_total,
									 int world, int nq, int k, int cap,
									 uint64_t *d_out_tids, float *d_out_dist, int *d_out_count);
/* ------------------------------------------------------------------ */
/* The exchange itself, in C (csrc/ndbhip_comm.cpp): a PostgreSQL backend cannot import torch.  One process per
 * GPU; the reference has no multi-GPU path of its own (no NCCL / RCCL / MPI call anywhere in the tree), the
 * merge rule it would have to follow is src/util/distributed.c:204-244 (smaller distance first, ties by the
 * order a single backend meets them), which the replay merge reproduces exactly.
 *
 *   rank 0:  ndbhip_comm_unique_id(id);  hand the 128 bytes to the other ranks (any side channel: the
 *            postmaster's shared memory, a file, MPI, torch.distributed ...)
 *   all:     ndbhip_comm_init(id, rank, world);          RCCL communicator on this process's device
 *   all:     ndbhip_ivf_search_sharded(shard, ...);      per batch: select (split by queries) -> all-gather of
 *            probes -> scan of the own lists -> all-gather of <= 3k records per query -> replay merge;
 *            asynchronous on the library's stream, every rank gets the full result
 *
 * ndbhip_comm_init_shm is the same group over a POSIX shared-memory segment on the host (name "/...", created
 * by rank 0, `slot_bytes` >= the largest per-rank message: nq * 3k * 16): for ranks that cannot form an RCCL
 * communicator (several processes on one device, no librccl).  Its collectives block the calling thread.
 * librccl is opened with dlopen by ndbhip_comm_unique_id / ndbhip_comm_init, never at library load. */
#define NDBHIP_COMM_ID_BYTES 128
int			ndbhip_comm_unique_id(void *out_id);
int			ndbhip_comm_init(const void *unique_id, int rank, int world);
int			ndbhip_comm_init_shm(const char *name, int rank, int world, size_t slot_bytes);
int			ndbhip_comm_rank(void);
int			ndbhip_comm_world(void);
int			ndbhip_comm_destroy(void);
/* d_recv[r * bytes .. (r + 1) * bytes) = rank r's d_send[0 .. bytes); without a communicator: a copy */
int			ndbhip_comm_allgather(const void *d_send, void *d_recv, size_t bytes);
int			ndbhip_ivf_search_sharded(ndbhip_ivf *shard, const float *d_queries, int nq, int strategy, int nprobe,
									  int k, int64_t max_candidates, uint64_t *d_out_tids, float *d_out_dist,
									  int *d_out_count);
/* Personalised exchange (csrc/ndbhip_comm.cpp): bytes [send_off[p], send_off[p + 1]) of d_send go to rank p and
 * land at [recv_off[r], recv_off[r + 1]) of p's d_recv (r = the sender); host arrays of world + 1 byte offsets. */
int			ndbhip_comm_alltoallv(const void *d_send, const size_t *send_off, void *d_recv, const size_t *recv_off);
/* In-place element-wise minimum of n floats over the ranks (device pointer, the library's stream): what a sharded
 * screened scan does to its queries' first thresholds between the seeds and the sweep, so that a rank that does not
 * hold a query's own list still sweeps against the bound of the rank that does.  RCCL: ncclAllReduce(ncclMin). */
int			ndbhip_comm_allreduce_min_f32(float *d_buf, size_t n);
/*
 * ivfbuild over the communicator's ranks, each holding a contiguous slice of the table in heap order (rank 0 the
 * first rows, rank 1 the next ...): rank 0 runs the k-means on the sample (the table's first min(10000, 100 lists)
 * rows, which its slice must hold), the centroids are broadcast, every rank assigns its own rows, the list
 * histograms are all-gathered, whole lists are dealt to ranks by length (longest first to the least loaded rank,
 * the same deal on every rank) and every row travels once, to its list's owner.  `ix` becomes this rank's shard:
 * all centroids, the global list lengths, and the rows of its own lists in heap order — the mirror
 * ndbhip_ivf_shard(full, owned) would cut out of the single-process build, row for row.  out_owned (optional,
 * [nlists] bytes): which lists this rank holds.  Without a communicator: ndbhip_ivf_build_device.
 */
int			ndbhip_ivf_build_sharded(ndbhip_ivf *ix, const float *d_rows, const uint64_t *d_tids, int64_t nrows_local,
									 int max_iter, int *out_iters, uint8_t *out_owned);

/* Host form of the same merge (results already on the host, e.g. gathered by
 * the PostgreSQL backend from several device-owner processes). Pure C, needs
 * no device. */
int			ndbhip_merge_topk_host(const ndbhip_cand *cand, const int *ncand, const int64_t *total,
								   int world, int nq, int k, int cap,
								   uint64_t *out_tids, float *out_dist, int *out_count);

/* ------------------------------------------------------------------ */
/* IVF build (src/index/ivf_am.c:501-745, 2070-2294) and insert-time
 * assignment (:905-935).                                               */
/* ------------------------------------------------------------------ */
/* kmeans_init + kmeans_run on the first n sample rows. d_samples in HBM,
 * centroids out (device) [k*dim]; returns iterations in *out_iters. */
int			ndbhip_kmeans_device(const float *d_samples, int n, int dim, int k, int max_iter,
								 float threshold, float *d_centroids, int *d_assign, int *d_counts,
								 int *out_iters, float *out_cost);
/* Nearest centroid by sqrtf(fp32 L2), strict <, first wins (:915-934). */
int			ndbhip_ivf_assign_device(const float *d_centroids, int ncentroids, int dim,
									 const float *d_rows, int64_t nrows, int *d_out_list);
/* The same from host memory (rows in heap order, heapPtrs 6 bytes each): staged by the library. */
int			ndbhip_ivf_build(ndbhip_ivf *ix, const float *rows, const uint8_t *tids6, int64_t nrows, int max_iter,
							 int *out_iters);
/* Whole build in HBM: sample first min(10000, 100*nlists) rows, k-means,
 * assign all nrows, pack lists in insertion (heap) order, adopt into ix. */
int			ndbhip_ivf_build_device(ndbhip_ivf *ix, const float *d_rows, const uint64_t *d_tids,
									int64_t nrows, int max_iter, int *out_iters);
/* Optional last step of a build (or of a load): prepare now what the first batched scan of operator class
 * `strategy` would otherwise prepare lazily — sublists of the long lists, the rows' fp16 planes in the matrix-core
 * sweep's layout, norms, radii — so that an index is searchable at full speed when ambuild returns (ivf_am.c:501-745
 * has no such step: its lists are read from the buffer manager as they are).  Later appends / deletes are folded
 * in by the next scan. */
int			ndbhip_ivf_prepare(ndbhip_ivf *ix, int strategy);

/* ------------------------------------------------------------------ */
/* HNSW mirror and search (src/index/hnsw_am.c:1545-2080).  Node b = block
 * number b (block 0 is the meta page and is never a node).
 *   vecs       [nblocks * dim], row b = node b's vector
 *   levels     [nblocks]
 *   ncount     [nblocks * 16]  neighborCount[level]
 *   nbr_off    [nblocks + 1]   start of node b's neighbour slots in `nbrs`
 *   nbrs       node b holds (levels[b]+1) * 2m slots, level-major, unused = 0xFFFFFFFF
 *   tids6      [nblocks * 6]   heapPtr of each node                     */
/* ------------------------------------------------------------------ */
int			ndbhip_hnsw_create(int dim, int m, ndbhip_hnsw **out);
int			ndbhip_hnsw_destroy(ndbhip_hnsw *g);
int			ndbhip_hnsw_load(ndbhip_hnsw *g, uint32_t nblocks, const float *vecs, const int32_t *levels,
							 const int16_t *ncount, const int64_t *nbr_off, const uint32_t *nbrs,
							 const uint8_t *tids6, uint32_t entry_point, int entry_level);
/* hnswbuild (src/index/hnsw_am.c:343-415): hnswInsertNode (:2091-2670) for rows 0..n-1 in order, on
 * rows already in HBM; node i+1 = row i.  levels[i] (host) = the level hnswGetRandomLevel (:1143-1161)
 * drew for row i — injected, because the reference draws it from random().  Links are always built with
 * L2 and ef = k = ef_construction, as the reference does (quirk Q12). */
int			ndbhip_hnsw_build_device(ndbhip_hnsw *g, const float *d_rows, const uint64_t *d_tids, uint32_t n,
									 const int32_t *levels, int ef_construction);
/* hnswinsert (src/index/hnsw_am.c:478-538) = the same hnswInsertNode for n MORE rows on top of the graph the
 * mirror holds (built here or loaded; a loaded graph is first given the dense 16-level layout): node
 * nblocks + i = row i.  On an empty mirror this is ndbhip_hnsw_build_device. */
int			ndbhip_hnsw_insert_device(ndbhip_hnsw *g, const float *d_rows, const uint64_t *d_tids, uint32_t n,
									  const int32_t *levels, int ef_construction);
/* ... and for host rows / heapPtrs (6 bytes each), staged by the library */
int			ndbhip_hnsw_insert(ndbhip_hnsw *g, const float *rows, const uint8_t *tids6, uint32_t n,
							   const int32_t *levels, int ef_construction);
/* How the last ndbhip_hnsw_build_device ran: out[0] walks (one per insert and linked level), out[1] walks
 * that had to run again because an earlier insert of their batch wrote a list they had read, out[2] walks
 * whose read-set log overflowed, out[3] speculate+commit rounds, out[4] batches, out[5] largest batch.
 * All zero after a sequential (one-wave) build. */
int			ndbhip_hnsw_build_stats(const ndbhip_hnsw *g, int64_t out[6]);
/* How ndbhip_hnsw_build_device schedules the inserts.  optimistic = 0: one wave inserts row after row.
 * optimistic = 2: as 1, but the commit is done walk by walk by one wave; 3: as 1 with the sorted-replay
 * chunked commit whatever m (both kept for cross-checking).
 * optimistic = 1 (default): the walks of up to min(batch_max, nodes so far / batch_div) inserts run in
 * parallel against the graph as it stands and are committed in insert order up to the first walk that
 * read a list an earlier insert of the batch has written since; the rest runs again in the next round.
 * Both produce the same graph, slot for slot
 * (defaults: batch_div 64, batch_max 1024). */
int			ndbhip_hnsw_set_build_mode(int optimistic, int batch_div, int batch_max);

/* hnswbulkdelete (src/index/hnsw_am.c:544-720) with the callback = "heapPtr is one of the n TIDs": live
 * hits are processed in block order — unlinked from the lists of the nodes THEIR lists name
 * (hnswRemoveNodeFromNeighbor, :2747-2840), the entry point moves to the hit's first valid neighbour (top
 * level first) or becomes invalid, the line pointer is marked dead.  As in the reference nothing else changes:
 * links INTO a dead node stay and hnswSearch does not test the dead flag, so it can still be walked and
 * returned.  A loaded (packed) graph is converted to the dense 16-level layout first.  m <= 32. */
int			ndbhip_hnsw_delete(ndbhip_hnsw *g, const uint8_t *tids6, int64_t n, int64_t *removed);

/* Read the graph back in the dense 16-level layout (any pointer may be NULL):
 * levels [nblocks], ncount [nblocks*16], nbrs [nblocks*16*2m]. */
int			ndbhip_hnsw_export(const ndbhip_hnsw *g, uint32_t *nblocks, int32_t *levels, int16_t *ncount,
							   uint32_t *nbrs, uint32_t *entry_point, int *entry_level);
/* Which hnswSearch kernel serves queries: 0 auto (block-cooperative when dim % 4 == 0), 1 one wave per query
 * scoring every row in the reference's own summation order, 2 block-cooperative (one 256-thread block per
 * query; partial sums accepted only when the float4 result is provably the sequential one, else redone in
 * order).  Both return the same bits. */
int			ndbhip_hnsw_set_search_mode(int mode);
/* hnswSearch for nq queries: strategy in {1,2,3}; ef = neurondb.hnsw_ef_search;
 * k = neurondb.hnsw_k.  out_blocks/out_dist [nq*k]; out_tids6 nullable
 * (hnswgettuple's node->heapPtr lookup, :1009-1053); out_scored nullable
 * [nq] = hnswComputeDistance calls per query. */
int			ndbhip_hnsw_search(ndbhip_hnsw *g, const float *queries, int nq, int strategy, int ef, int k,
							   uint32_t *out_blocks, float *out_dist, int *out_count,
							   uint8_t *out_tids6, int64_t *out_scored);
int			ndbhip_hnsw_search_device(ndbhip_hnsw *g, const float *d_queries, int nq, int strategy, int ef,
									  int k, uint32_t *d_out_blocks, float *d_out_dist, int *d_out_count,
									  uint64_t *d_out_tids, int64_t *d_out_scored);
/* hnsw_search_layer (src/scan/hnsw_scan.c:379-477): the best-first search the reference ships next to
 * hnswSearch and never calls, rule for rule — compute_l2_distance (fp32 sequential + sqrtf,
 * :105-118) whatever `strategy` says (the reference's body never reads that argument), hill climb on the upper
 * layers (:485-636), at layer 0 a min-heap of at most 2 * ef candidates, the bound results[k - 1], k unsorted
 * result slots (:645-844).  Same outputs as ndbhip_hnsw_search, results in SLOT order (not sorted), as the
 * reference returns them (:826-830).  A packed mirror is converted to the dense layout first. */
int			ndbhip_hnsw_search_layer(ndbhip_hnsw *g, const float *queries, int nq, int strategy, int ef, int k,
									 uint32_t *out_blocks, float *out_dist, int *out_count,
									 uint8_t *out_tids6, int64_t *out_scored);
int			ndbhip_hnsw_search_layer_device(ndbhip_hnsw *g, const float *d_queries, int nq, int strategy, int ef,
											int k, uint32_t *d_out_blocks, float *d_out_dist, int *d_out_count,
											uint64_t *d_out_tids, int64_t *d_out_scored);

/* The `intended` HNSW: what the reference's index would be with its search-and-link bugs repaired —
 * the greedy descent's result used as the next level's entry point (hnsw_am.c:2155-2286 drops it), every level searched
 * on its own links with the best-first layer search src/scan/hnsw_scan.c:379-483 specifies, neighbours chosen among
 * the ef_construction nearest by the textbook heuristic (or the nearest m), a full list PRUNED when a back-link
 * arrives (hnsw_am.c:2503-2513 is the unreachable branch).  Same page-level data model (levels, 16 x 2m slots, entry
 * point), same level draws; every build-time comparison is L2 like hnswInsertNode's.  The build is batch-synchronous
 * (batches of clamp(nodes so far / batch_div, 1, batch_max): the members search the graph as it stood when their
 * batch began, their links are applied in insertion order), which is part of the definition:
 * oracle/ndb_oracle_hnsw2.c states it sequentially and the device graph equals it slot for slot.  Build-time distances
 * are squared L2 in fp64 (a fixed 64-way summation tree).
 * The SEARCH takes the operator class's strategy like hnswSearch does (src/index/hnsw_am.c:918-921 hands sk_strategy down;
 * neurondb--1.0.sql:2941-2965: <-> 1, <=> 2, <#> 3): 1 = L2, distances (float) sqrt(d2); 2 = cosine, 3 = negative inner
 * product: descent and layer search order by that metric's fp64 key (1 - dot / (|q| |x|), -dot: the same summation tree),
 * the result set's (at most ef) entries are then scored with hnswComputeDistance's own arithmetic (hnsw_am.c:1321-1337,
 * bit for bit what ndbhip_hnsw_search returns for the same pair), ordered by (that float4, block), and the k nearest
 * returned with those values.  Any other strategy: NDBHIP_ERR_INVALID (the reference's ERROR, :1339-1343). */
int			ndbhip_hnsw_build_intended_device(ndbhip_hnsw *g, const float *d_rows, const uint64_t *d_tids, uint32_t n,
											  const int32_t *levels, int ef_construction, int batch_div, int batch_max);
/* hnswinsert under `intended` (src/index/hnsw_am.c:478-538 / hnswInsertNode :2091-2670 with the repairs above): n MORE rows on
 * top of the graph the mirror holds (built here in either mode, or loaded): node nblocks + i = row i; the batch schedule goes
 * on from the relation's size (one row = one batch = the sequential textbook insert).  oracle: ndbo_h2_build on a graph that
 * is not empty.  The _device form takes rows and packed heapPtrs in HBM; the other stages host rows / 6-byte heapPtrs. */
int			ndbhip_hnsw_insert_intended_device(ndbhip_hnsw *g, const float *d_rows, const uint64_t *d_tids, uint32_t n,
											   const int32_t *levels, int ef_construction, int batch_div, int batch_max);
int			ndbhip_hnsw_insert_intended(ndbhip_hnsw *g, const float *rows, const uint8_t *tids6, uint32_t n,
										const int32_t *levels, int ef_construction, int batch_div, int batch_max);
int			ndbhip_hnsw_search_intended_device(ndbhip_hnsw *g, const float *d_queries, int nq, int strategy, int ef, int k,
											   uint32_t *d_out_blocks, float *d_out_dist, int *d_out_count,
											   uint64_t *d_out_tids, int64_t *d_out_evals);
/* The same search with the walk on fp16 WALK ROWS (round 5; SURVEY 8f-4, src/index/hnsw_am.c:1436-1451 reads halfvec node
 * vectors): every element of the graph's rows through the reference's float4_to_fp16 (src/types/quantization.c:141-168) —
 * what a halfvec column of the same data holds; the twin is made on the device at the first call and again after rows were
 * appended (+ 0.5 x the rows' bytes).  Descent and layer search run on the halves (half the bytes per evaluated row); the
 * result set's ef entries are then scored against the float4 rows with the definition's arithmetic, ordered by that, and
 * the k nearest returned with those distances: oracle/ndb_oracle_hnsw2.c ndbo_h2_search_w16, equal id for id and bit for
 * bit.  dim % 4 == 0 and dim <= 1024, else NDBHIP_ERR_UNSUPPORTED (use the float4 walk). */
int			ndbhip_hnsw_search_intended_w16_device(ndbhip_hnsw *g, const float *d_queries, int nq, int strategy, int ef, int k,
												   uint32_t *d_out_blocks, float *d_out_dist, int *d_out_count,
												   uint64_t *d_out_tids, int64_t *d_out_evals);
/* The intended search from HOST pointers (replaces the body of hnswgettuple's hnswSearch call, src/index/hnsw_am.c:998-1001,
 * when neurondb.ref_compat is off): queries [nq x dim] are copied in, results copied out before return; out_tids6 nullable =
 * node->heapPtr of every result (:1009-1053); out_evals nullable [nq].  walk16 != 0: the walk on fp16 walk rows. */
int			ndbhip_hnsw_search_intended(ndbhip_hnsw *g, const float *queries, int nq, int strategy, int ef, int k, int walk16,
										uint32_t *out_blocks, float *out_dist, int *out_count, uint8_t *out_tids6,
										int64_t *out_evals);
/* A second HANDLE on the same graph (rows, levels, neighbour lists, TIDs, walk rows) with a workspace of its own — visited
 * maps, result blocks —: two batches of searches in flight, a host thread, a stream (ndbhip_set_thread_stream) and a handle
 * each (a batch ends with its longest walks; the next one's fill the device meanwhile).  Both handles are frozen while the
 * share lives (loads, inserts, builds, deletes: NDBHIP_ERR_STATE); ndbhip_hnsw_destroy of the source is refused until its
 * shares are gone.  Walk rows are made by the source's first ndbhip_hnsw_search_intended_w16_device, before sharing.
 * Threading as for ndbhip_ivf_share: share and destroy while no search runs on the source. */
int			ndbhip_hnsw_share(ndbhip_hnsw *src, ndbhip_hnsw **out);
/* bit 0: the heuristic (else the nearest); bit 1: a new node takes up to 2m links at level 0 instead of m; bit 2 (with
 * bit 0): the places the heuristic leaves empty go to the nearest candidates it passed over (keepPrunedConnections). */
int			ndbhip_hnsw_set_intended_select(int select);

/* ------------------------------------------------------------------ */
/* hnsw relation pages <-> mirror, PostgreSQL-free (src/index/hnsw_am.c:108-181, 1091-1110, 2288-2332):
 * block 0 = HnswMetaPageData, every other block = ONE item = HnswNodeData (48 B) + vector + neighbour slots
 * of levels 0..level; a line pointer hnswbulkdelete marked dead (:693) comes back in dead[].  The array
 * layout is ndbhip_hnsw_export's (dense: nbrs [nblocks][16][2m]).  pack/unpack/info are pure host code. */
/* ------------------------------------------------------------------ */
int			ndbhip_hnsw_pages_info(const uint8_t *pages, uint32_t nblocks, int *dim, int *m, int *ef_construction,
								   int *ef_search, uint32_t *entry_point, int *entry_level);
int			ndbhip_hnsw_pages_unpack(const uint8_t *pages, uint32_t nblocks, float *vecs, int32_t *levels,
									 int16_t *ncount, uint32_t *nbrs, uint8_t *tids6, uint8_t *dead);
int			ndbhip_hnsw_pages_pack(int dim, int m, int ef_construction, int ef_search, uint32_t nblocks,
								   const float *vecs, const int32_t *levels, const int16_t *ncount,
								   const uint32_t *nbrs, const uint8_t *tids6, const uint8_t *dead,
								   uint32_t entry_point, int entry_level, uint8_t *pages, uint32_t nblocks_cap);
int			ndbhip_hnsw_load_pages(ndbhip_hnsw **out, const uint8_t *pages, uint32_t nblocks);
/* pages == NULL: only *nblocks_out (= blocks the relation needs) */
int			ndbhip_hnsw_write_pages(const ndbhip_hnsw *g, int ef_construction, int ef_search, uint8_t *pages,
									uint32_t nblocks_cap, uint32_t *nblocks_out);
int			ndbhip_hnsw_shape(const ndbhip_hnsw *g, int *dim, int *m);
/* efConstruction / efSearch of the meta page (hnsw_am.c:108-120; defaults 200 / 64): taken from the pages by
 * ndbhip_hnsw_load_pages, from the argument by a build, and read back by the AM callbacks of ndb_am.h */
int			ndbhip_hnsw_get_meta(const ndbhip_hnsw *g, int *ef_construction, int *ef_search);
int			ndbhip_hnsw_set_meta(ndbhip_hnsw *g, int ef_construction, int ef_search);
/* the rest of the mirror ndbhip_hnsw_export does not return: vectors [nblocks*dim], heapPtrs [nblocks*6],
 * dead flags [nblocks] (each may be NULL); ndbhip_hnsw_set_dead_flags restores the latter after a load */
int			ndbhip_hnsw_export_rows(const ndbhip_hnsw *g, float *vecs, uint8_t *tids6, uint8_t *dead);
int			ndbhip_hnsw_set_dead_flags(ndbhip_hnsw *g, const uint8_t *dead);

/* ------------------------------------------------------------------ */
/* Datum -> dense float4[] (replaces ivfExtractVectorData, src/index/ivf_am.c:117-218,
 * and hnswExtractVectorData, src/index/hnsw_am.c:1402-1519).  `datum` is the
 * DETOASTED varlena image of the indexed value (vector / halfvec / sparsevec /
 * bit), `kind` says which.  Pure host code; needs no device.  halfvec elements
 * decode exactly like the reference's fp16_to_float (src/types/quantization.c:
 * 170-218), sparsevec scatters into zeros dropping out-of-range indices, bit
 * maps 1 -> +1.0f and 0 -> -1.0f.  out may be NULL to query the dimension.   */
/* ------------------------------------------------------------------ */
#define NDBHIP_TYPE_VECTOR    0
#define NDBHIP_TYPE_HALFVEC   1
#define NDBHIP_TYPE_SPARSEVEC 2
#define NDBHIP_TYPE_BIT       3
int			ndbhip_extract_vector(int kind, const void *datum, size_t datum_len, float *out, int out_cap,
								  int *out_dim);

/* ------------------------------------------------------------------ */
/* Batch distance (replaces neurondb_gpu_batch_l2_distance & co:
 * include/neurondb_gpu.h:94-106, src/gpu/common/gpu_batch.c:27-83, whose
 * body is a CPU loop).  results[q*nv + v], host pointers.  recipe selects
 * whose arithmetic is reproduced, bit for bit:
 *   0 ivf   ivfComputeDistance (fp32 sequential), strategy 1/2/3, 4 = squared L2
 *   1 hnsw  hnswComputeDistance (fp64 accumulate), strategy 1/2/3
 *   2 the SQL operators <-> <=> <#> as a default x86-64 build runs them: the scalar
 *     kernels of src/vector/vector_distance.c (Kahan double L2, double cosine / IP)
 *     through the dispatchers of vector_distance_simd.c:467-613 (strategy 3 = +dot, Q15)
 *   3 / 4 the same operators in an AVX2 / AVX-512 build: 8 / 16 fp32 lane accumulators,
 *     FMA cosine, the fixed horizontal-sum tree (vector_distance_simd.c:84-392)
 *   5 the halfvec operators (src/types/quantization.c:1985-2116): queries and vectors are
 *     uint16 fp16 images, decoded per element like fp16_to_float (strategy 3 = -dot)       */
/* ------------------------------------------------------------------ */
int			ndbhip_batch_distance(const float *queries, const float *vectors, float *results,
								  int nq, int nv, int dim, int strategy, int recipe);

/* ------------------------------------------------------------------ */
/* The launcher shapes of the reference's GPU vtable (struct ndb_gpu_backend,
 * include/neurondb_gpu_backend.h:54-84; ROCm bodies src/gpu/rocm/gpu_backend_rocm.c:752-1000), host
 * pointers in and out like there — with the arithmetic of the CPU functions the reference falls back to,
 * so an answer does not depend on which side produced it.  include/ndb_backend.h wraps them in a vtable.
 *   pair_distance  out[i] = l2_distance / cosine_distance / inner_product (A[i], B[i])  (launch_l2_distance,
 *                  launch_cosine: n PAIRS, not a matrix); strategy 1 / 2 / 3, scalar double kernels of
 *                  src/vector/vector_distance.c:93-227 (3 = +dot, the operator's sign)
 *   kmeans_assign  idx[i] = first minimum of the fp32 squared L2 to the k centroids (launch_kmeans_assign;
 *                  ivf_am.c:2157-2180, 2274-2294)
 *   kmeans_update  C[c] = members of c added in sample order / (float) count, empty clusters keep their
 *                  centroid (launch_kmeans_update; ivf_am.c:2182-2213); n is bounded by the LDS-resident
 *                  member list (37 k rows — the reference's k-means never sees more than 10 000)
 *   quant_fp16     float4_to_fp16 (launch_quant_fp16; src/types/quantization.c:141-168)     */
/* ------------------------------------------------------------------ */
int			ndbhip_pair_distance(const float *A, const float *B, float *out, int n, int dim, int strategy);
int			ndbhip_kmeans_assign(const float *X, const float *C, int *idx, int n, int dim, int k);
int			ndbhip_kmeans_update(const float *X, const int *idx, float *C, int n, int dim, int k);
int			ndbhip_quant_fp16(const float *in, uint16_t *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif							/* NDBHIP_H */
