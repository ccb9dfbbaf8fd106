/*
 * ndb_oracle.h — CPU ORACLE for the NeuronDB vector-distance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a PostgreSQL-free, literal C restatement
 * of the reference's algorithms (loop order, operand types, tie rules).  It is
 * the checker for the HIP path; it is never linked into, imported by or called
 * from the product (neurondb_amd/, include/).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - scalar distance recipes: PINNED to the reference's own known-answer
 *     cases (NeuronDB/t/005_distances_comprehensive.t, sql/03_distance_metrics.sql,
 *     sql/10_gpu_distance_wrappers.sql) in tests/test_oracle_known_answers.py.
 *   - kNN id lists, k-means centroids, HNSW graphs, the unused best-first search of
 *     src/scan/hnsw_scan.c (ndbo_hnsw_search_layer): PARITY UNPINNED — the
 *     reference holds no golden output for them (its tree has no expected/ .out files)
 *     and its sources cannot be compiled here (every file includes postgres.h).
 *     These functions follow the cited reference lines statement by statement.
 *
 * All citations are relative to /root/reference/NeuronDB/.
 *
 * Build: -O2 -ffp-contract=off (no FMA contraction: the reference's default
 * PGXS build targets baseline x86-64, which has no FMA).
 */
#ifndef NDB_ORACLE_H
#define NDB_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDBO_INVALID_BLOCK 0xFFFFFFFFu	/* InvalidBlockNumber */
#define NDBO_HNSW_MAX_LEVEL 16			/* src/index/hnsw_am.c:85 */

/* ItemPointerData: 6 bytes {bi_hi, bi_lo, offset} */
typedef struct ndbo_tid
{
	uint16_t	bi_hi;
	uint16_t	bi_lo;
	uint16_t	posid;
}			ndbo_tid;

/* ------------------------------------------------------------------ */
/* Scalar distance recipes                                             */
/* ------------------------------------------------------------------ */

/* src/index/ivf_am.c:1550-1592  (strategy 1 = L2, 2 = cosine, other = L2).
 * strategy 3 (negative inner product, fp32 sequential) is NOT in the reference
 * (quirk Q2); it is provided for the "intended" mode and documented as new. */
float		ndbo_ivf_distance(const float *vec1, const float *vec2, int dim, int strategy);

/* src/index/ivf_am.c:2255-2269 (squared L2, fp32 sequential) */
float		ndbo_ivf_l2sq(const float *v1, const float *v2, int dim);

/* src/index/hnsw_am.c:1301-1345 (1 = L2, 2 = cosine, 3 = -IP).
 * Returns NaN and sets *err = 1 for an unsupported strategy (reference: ERROR). */
float		ndbo_hnsw_distance(const float *vec1, const float *vec2, int dim, int strategy, int *err);

/* Operator kernels, scalar path: src/vector/vector_distance.c:93-122, 145-157, 180-213 */
float		ndbo_op_l2_scalar(const float *a, const float *b, int dim);
float		ndbo_op_ip_scalar(const float *a, const float *b, int dim);	/* returns -sum */
float		ndbo_op_cosine_scalar(const float *a, const float *b, int dim);

/* Operator kernels, lane-exact emulation of the AVX2 (lanes=8) / AVX-512
 * (lanes=16) paths incl. the horizontal-sum tree:
 * src/vector/vector_distance_simd.c:84-137, 158-392 */
float		ndbo_op_l2_simd(const float *a, const float *b, int dim, int lanes);
float		ndbo_op_ip_simd(const float *a, const float *b, int dim, int lanes);	/* +sum (Q15) */
float		ndbo_op_cosine_simd(const float *a, const float *b, int dim, int lanes);	/* uses fmaf */

/* Dispatchers: src/vector/vector_distance_simd.c:467-613.  simd = 0 (scalar
 * build, the reference default: Q16), 8 (AVX2 build) or 16 (AVX-512 build). */
float		ndbo_op_l2(const float *a, const float *b, int dim, int simd);
float		ndbo_op_ip(const float *a, const float *b, int dim, int simd);	/* `<#>`: +dot (Q15) */
float		ndbo_op_cosine(const float *a, const float *b, int dim, int simd);

/* src/types/quantization.c:141-168, 170-218 (note the subnormal quirk Q20) */
uint16_t	ndbo_float4_to_fp16(float f);
float		ndbo_fp16_to_float(uint16_t h);

/* src/types/quantization.c:1985-2116 */
float		ndbo_halfvec_l2(const uint16_t *a, const uint16_t *b, int dim);
float		ndbo_halfvec_cosine(const uint16_t *a, const uint16_t *b, int dim);
float		ndbo_halfvec_ip(const uint16_t *a, const uint16_t *b, int dim);	/* -sum */

/* src/index/ivf_am.c:117-218 on detoasted datum images: kind 0 vector, 1 halfvec,
 * 2 sparsevec, 3 bit; returns -1 for an unsupported kind (reference: ERROR) */
int			ndbo_extract_vector(int kind, const unsigned char *datum, float *out, int *out_dim);

/* ------------------------------------------------------------------ */
/* IVF                                                                 */
/* ------------------------------------------------------------------ */

/*
 * Flat, PG-free image of an IVF index.  Rows of list L are
 * vecs[list_off[L] .. list_off[L+1]) in page-chain order (the order
 * ivfCollectCandidates meets them: src/index/ivf_am.c:1793-1840).
 * `live` (nullable) models !ItemIdIsValid||ItemIdIsDead / dim mismatch skips
 * (src/index/ivf_am.c:1816-1822): a row with live[i]==0 is skipped.
 */
typedef struct ndbo_ivf
{
	int			dim;
	int			nlists;			/* meta->nlists */
	int			maxoff;			/* centroid items present on the centroid page(s) */
	const float *centroids;		/* [maxoff * dim] */
	const int  *centroid_dim;	/* nullable; centroid->dim per item */
	const int64_t *list_off;	/* [maxoff + 1] */
	const float *vecs;			/* [N * dim] */
	const ndbo_tid *tids;		/* [N] */
	const uint8_t *live;		/* nullable [N] */
}			ndbo_ivf;

/* src/index/ivf_am.c:1597-1717.  selected[] has so_nprobe entries, zero-filled
 * on entry by the caller exactly as ivfgettuple's palloc0 does (:1978).
 * Returns the clamped nprobe actually filled. */
int			ndbo_ivf_select_clusters(const ndbo_ivf *ix, const float *query, int nprobe, int *selected);

/* src/index/ivf_am.c:1722-1909.  max_candidates = k*10 reproduces the
 * reference (Q3); max_candidates <= 0 means "no cap" (intended mode).
 * out_tids/out_dist need room for k entries.  Returns resultCount.
 * *n_scored (nullable) receives the number of distance evaluations. */
int			ndbo_ivf_collect_candidates(const ndbo_ivf *ix, const float *query, int strategy,
										const int *selected, int nprobe, int k, int64_t max_candidates,
										ndbo_tid *out_tids, float *out_dist, int64_t *n_scored);

/* ivfgettuple first-call work: src/index/ivf_am.c:1976-1999 */
int			ndbo_ivf_search(const ndbo_ivf *ix, const float *query, int strategy, int nprobe, int k,
							int64_t max_candidates, ndbo_tid *out_tids, float *out_dist, int64_t *n_scored);

/* ndb_oracle_mt.c: the same single-query functions on one thread per host core (CPU baseline only: no arithmetic of
 * its own).  ndbo_mt_spread_copy: a copy whose pages are first touched by the workers, 2 MiB stripes round-robin (a
 * row array filled by one thread sits on one NUMA node).  The batch drivers return the parallel section's wall time. */
void	   *ndbo_mt_spread_copy(const void *src, size_t bytes, int nthreads);
double		ndbo_mt_ivf_search_batch(const ndbo_ivf *ix, const float *queries, int nq, int strategy, int nprobe, int k,
									 int64_t max_candidates, int nthreads, ndbo_tid *out_tids, float *out_dist,
									 int *out_count, int64_t *n_scored);
double		ndbo_mt_ivf_assign_batch(const float *rows, int64_t n, int dim, const float *cents, int k, int nthreads,
									 int *out);

/* k-means: src/index/ivf_am.c:2070-2294.  data = first n sample rows
 * (row-major [n*dim]); centroids out [k*dim]; assignments out [n]; counts out [k].
 * Returns the number of Lloyd iterations executed. */
int			ndbo_kmeans(const float *data, int n, int dim, int k, int max_iter, float threshold,
						float *centroids, int *assignments, int *counts, float *final_cost);

/* One Lloyd step pieces (exposed so the HIP kernels can be checked one by one) */
void		ndbo_kmeans_assign(const float *data, int n, int dim, const float *centroids, int k,
							   int *assignments, int *counts);
void		ndbo_kmeans_update(const float *data, int n, int dim, const int *assignments,
							   const int *counts, int k, float *centroids);
float		ndbo_kmeans_cost(const float *data, int n, int dim, const int *assignments,
							 const float *centroids);

/* insert-time assignment: src/index/ivf_am.c:905-935 (sqrtf compare, strict <) */
int			ndbo_ivf_assign(const float *centroids, const int *centroid_dim, int nlists, int maxoff,
							int dim, const float *vec, float *min_dist);

/* ------------------------------------------------------------------ */
/* HNSW                                                                */
/* ------------------------------------------------------------------ */

/*
 * PG-free image of the HNSW index: node b (block number b, 1-based; block 0 is
 * the meta page) lives at slot b.  Every node carries neighbour storage for
 * all HNSW_MAX_LEVEL levels so that the reference's out-of-node writes
 * (quirk Q12/Q21, src/index/hnsw_am.c:2487) land in a defined place; reads are
 * still guarded by node->level exactly where the reference guards them.
 */
typedef struct ndbo_hnsw
{
	int			dim;
	int			m;
	int			ef_construction;
	uint32_t	entry_point;	/* InvalidBlockNumber when empty */
	int			entry_level;
	int			max_level;
	int64_t		inserted;
	uint32_t	nblocks;		/* RelationGetNumberOfBlocks: nodes + 1 */
	uint32_t	cap_blocks;
	float	   *vecs;			/* [cap_blocks * dim], row b = node b */
	ndbo_tid   *heap_tids;		/* [cap_blocks] */
	int		   *levels;			/* [cap_blocks] */
	int16_t    *ncount;			/* [cap_blocks * HNSW_MAX_LEVEL] */
	uint32_t   *nbrs;			/* [cap_blocks * HNSW_MAX_LEVEL * 2m] */
	uint8_t    *dead;			/* [cap_blocks] line pointer marked dead by hnswbulkdelete */
}			ndbo_hnsw;

ndbo_hnsw  *ndbo_hnsw_create(int dim, int m, int ef_construction, uint32_t cap_nodes);
void		ndbo_hnsw_free(ndbo_hnsw *g);

/* src/index/hnsw_am.c:1545-2080.  out arrays need room for k entries.
 * Returns resultCount; *n_scored counts hnswComputeDistance calls. */
int			ndbo_hnsw_search(const ndbo_hnsw *g, const float *query, int strategy, int ef_search, int k,
							 uint32_t *out_blocks, float *out_dist, int64_t *n_scored);
/* src/scan/hnsw_scan.c:379-477 hnsw_search_layer (the unused best-first search): always
 * compute_l2_distance; results in slot order, not sorted */
int			ndbo_hnsw_search_layer(const ndbo_hnsw *g, const float *query, int ef_search, int k,
								   uint32_t *out_blocks, float *out_dist, int64_t *n_scored);

/* src/index/hnsw_am.c:2091-2670 with the level injected (hnswGetRandomLevel
 * uses random(): :1143-1161).  Returns the new block number. */
uint32_t	ndbo_hnsw_insert(ndbo_hnsw *g, const float *vec, ndbo_tid heap_tid, int level);

/* hnswbulkdelete (src/index/hnsw_am.c:544-720) with the callback = "heapPtr is one of tids[0..n)".
 * Blocks are visited in ascending order; a node whose line pointer is already dead is skipped (:601);
 * a hit is unlinked from the lists of the nodes ITS lists name (hnswRemoveNodeFromNeighbor, :2747-2840:
 * first occurrence among the first count entries, shift down, count--), the entry point moves to the hit's
 * first valid neighbour (top level first) or to InvalidBlockNumber, and the line pointer is marked dead.
 * Nothing else changes: links INTO the dead node from other nodes stay, and hnswSearch does not test the dead
 * flag, so the node can still be walked and returned.  Returns tuples_removed. */
int64_t		ndbo_hnsw_bulkdelete(ndbo_hnsw *g, const ndbo_tid *tids, int64_t n);

/* ndb_oracle_hnsw2.c — the `intended` HNSW (its header says what it keeps of the reference and what it
 * repairs): squared L2 in fp64 summed by a fixed 64-way tree, best-first layer search, links pruned to the nearest
 * `cap`, batch-synchronous build schedule.  parity unpinned (nothing in the reference pins it). */
double		ndbo_h2_dist2(const float *a, const float *b, int dim);
int			ndbo_h2_search_layer(const ndbo_hnsw *g, const float *q, const uint32_t *ep, const double *epd, int nep, int ef,
								 int level, uint32_t *out_ids, double *out_d2, int64_t *evals, uint8_t *visited);
int			ndbo_h2_build(ndbo_hnsw *g, const float *vecs, const ndbo_tid *tids, int64_t n, const int *levels,
						  int batch_div, int batch_max, int select);
int			ndbo_h2_search(const ndbo_hnsw *g, const float *query, int ef, int k, uint32_t *out_blocks, float *out_dist,
						   int64_t *evals);
/* the same search with the walk on fp16 WALK ROWS (the reference's float4_to_fp16 of every element, groups-of-four
 * summation tree) and the final result set re-scored on the float4 rows: ndb_oracle_hnsw2.c "WALK ROWS" */
void		ndbo_h2_walk_rows(const float *vecs, int64_t nel, uint16_t *out);
double		ndbo_h2_dist2_w16(const float *q, const uint16_t *w, int dim);
int			ndbo_h2_search_w16(const ndbo_hnsw *g, const uint16_t *w16, const float *query, int ef, int k,
							   uint32_t *out_blocks, float *out_dist, int64_t *evals);
/* round 6: the intended search under the operator class's strategy (1 L2, 2 cosine, 3 negative inner product); w16 NULL = walk
 * on the float4 rows */
double		ndbo_h2_rinv(const ndbo_hnsw *g, const uint16_t *w16, uint32_t e);
double		ndbo_h2_walk_key(const ndbo_hnsw *g, const uint16_t *w16, const float *q, uint32_t e, int strategy);
int			ndbo_h2_search_s(const ndbo_hnsw *g, const uint16_t *w16, int strategy, const float *query, int ef, int k,
							 uint32_t *out_blocks, float *out_dist, int64_t *evals);

/* level = (int)(-log(r) * ml), clamped [0, 15]: src/index/hnsw_am.c:1143-1161 */
int			ndbo_hnsw_level_from_uniform(double r, float ml);

/* ------------------------------------------------------------------ */
/* Shard merge (src/util/distributed.c:204-244 rule applied to the      */
/* candidate array of ivfCollectCandidates)                             */
/* ------------------------------------------------------------------ */

/* Selection sort with index swaps over (dist[i]) exactly as
 * src/index/ivf_am.c:1861-1881 / src/index/hnsw_am.c:1984-2004: returns the
 * first min(k,n) indices into order[]. */
int			ndbo_selection_topk(const float *dist, int64_t n, int k, int64_t *order);

#ifdef __cplusplus
}
#endif
#endif							/* NDB_ORACLE_H */
