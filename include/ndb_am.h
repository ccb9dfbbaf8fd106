/*
 * ndb_am.h — the index access-method callbacks of the reference, PostgreSQL-free.
 *
 * The reference's scan entry points are static functions reached through IndexAmRoutine
 * (src/index/ivf_am.c:390-432, src/index/hnsw_am.c:293-335).  This header declares the same
 * four per-AM callbacks with the same meaning of every argument and the same state machine,
 * over the device mirror instead of the buffer manager, so that the glue in INTEGRATION.md is
 * a field-by-field forwarding.  What PostgreSQL supplies is replaced by the smallest plain-C
 * equivalent:
 *
 *   IndexScanDesc            -> ndb_index_scan (only the fields the callbacks touch)
 *   ScanKey                  -> ndb_scan_key   (sk_strategy + the DETOASTED datum image of sk_argument)
 *   ItemPointerData          -> ndb_item_pointer (6 bytes, same layout)
 *   GUCs read with GetConfigOption -> ndb_am_set_guc / ndb_am_get_guc
 *   ereport(ERROR)           -> negative NDBHIP_ERR_* return, text in ndbhip_last_error()
 *
 * Every callback cites the reference lines it mirrors in ndb_am.cpp.
 */
#ifndef NDB_AM_H
#define NDB_AM_H

#include <stddef.h>
#include <stdint.h>

#include "ndbhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ndb_item_pointer
{
	uint16_t	bi_hi;			/* ItemPointerData: ip_blkid.bi_hi, bi_lo, ip_posid */
	uint16_t	bi_lo;
	uint16_t	posid;
}			ndb_item_pointer;

/* one ORDER BY key: orderbys[0] of amrescan (ivf_am.c:1455-1536, hnsw_am.c:904-976) */
typedef struct ndb_scan_key
{
	int			sk_strategy;	/* operator strategy number: 1 <-> (L2), 2 <=> (cosine), 3 <#> (inner product) */
	int			sk_type;		/* NDBHIP_TYPE_VECTOR / HALFVEC / SPARSEVEC / BIT: the indexed column's type */
	const void *sk_argument;	/* detoasted varlena image of the query value, NULL = SK_ISNULL */
	size_t		sk_len;
}			ndb_scan_key;

typedef struct ndb_index_scan
{
	void	   *indexRelation;	/* ndbhip_ivf * or ndbhip_hnsw *: the mirror of the index relation */
	int			numberOfKeys;
	int			numberOfOrderBys;
	ndb_item_pointer xs_heaptid;	/* set by amgettuple when it returns true */
	float		xs_orderbyval;		/* xs_orderbyvals[0] (ivf sets it: ivf_am.c:2013-2020; hnsw does not, Q13) */
	int			xs_orderbynull;		/* xs_orderbynulls[0]: 1 when no value was set */
	int			xs_recheckorderby;	/* always 0 */
	void	   *opaque;
}			ndb_index_scan;

#define NDB_FORWARD_SCAN_DIRECTION 1

/* GUCs the callbacks read.  Names as in the reference where it has them (src/util/neurondb_guc.c):
 *   neurondb.hnsw_ef_search (64), neurondb.hnsw_k (10)             hnsw_am.c:923-936
 *   neurondb.ivf_probes (10)   defined by the reference but never read (Q4)
 *   neurondb.ivf_k (10)        the reference hard-codes k = 10 (Q3)
 *   neurondb.ref_compat (0)    1 = keep Q1/Q3/Q4: strategy 1, nprobe 10, k 10 and the k*10 candidate cap
 * Returns NDBHIP_ERR_INVALID for an unknown name or an out-of-range value. */
int			ndb_am_set_guc(const char *name, int value);
int			ndb_am_get_guc(const char *name, int *value);
/* GUC neurondb.device_service (a string: the shared-memory name of the device-owner process, include/ndb_service.h;
 * NULL or "" detaches).  While set, an ivf scan opened with ndb_ivfbeginscan_service (or index == NULL) is
 * answered by that process: the backend itself never initialises HIP and holds no mirror.  Connect failure =
 * NDBHIP_ERR_NODEVICE.  hnsw scans are not served: ndb_hnswbeginscan(NULL, ..) fails. */
int			ndb_am_use_service(const char *name);

/* ivf: src/index/ivf_am.c:1412-1437 / 1439-1545 / 1911-2027 / 2029-2048 */
ndb_index_scan *ndb_ivfbeginscan(ndbhip_ivf *index, int nkeys, int norderbys);
/* a scan the device-owner process answers (ndb_am_use_service), on the index with key `index_key` at generation
 * `index_version` (include/ndb_service.h: ndb_gen_get): a service holding another index or another generation is
 * NDBHIP_ERR_NODEVICE from ndb_ivfgettuple.  ndb_ivfbeginscan(NULL, ..) is the unkeyed form (key 0, generation 0). */
ndb_index_scan *ndb_ivfbeginscan_service(uint64_t index_key, uint64_t index_version, int nkeys, int norderbys);
int			ndb_ivfrescan(ndb_index_scan *scan, const ndb_scan_key *keys, int nkeys,
						  const ndb_scan_key *orderbys, int norderbys);
/* 1 = a tuple is in xs_heaptid / xs_orderbyval, 0 = no more tuples, < 0 = the reference's ERROR */
int			ndb_ivfgettuple(ndb_index_scan *scan, int direction);
void		ndb_ivfendscan(ndb_index_scan *scan);

/* hnsw: src/index/hnsw_am.c:880-902 / 904-976 / 978-1056 / 1058-1084 */
ndb_index_scan *ndb_hnswbeginscan(ndbhip_hnsw *index, int nkeys, int norderbys);
int			ndb_hnswrescan(ndb_index_scan *scan, const ndb_scan_key *keys, int nkeys,
						   const ndb_scan_key *orderbys, int norderbys);
int			ndb_hnswgettuple(ndb_index_scan *scan, int direction);
void		ndb_hnswendscan(ndb_index_scan *scan);

/* aminsert.  value = the detoasted datum of values[0] (NULL = isnull[0]: nothing is inserted, returns 0);
 * returns 1 when an entry was added, 0 when not, < 0 for the reference's ERROR.
 *   ivfinsert  src/index/ivf_am.c:797-1167
 *   hnswinsert src/index/hnsw_am.c:478-538; `level` = what hnswGetRandomLevel (:1143-1161) drew in the
 *              backend — it uses random(), so the draw stays with the caller; ndb_hnsw_level_from_uniform
 *              is the formula for a caller-supplied uniform r in (0, 1]. */
int			ndb_ivfinsert(ndbhip_ivf *index, const void *value, size_t value_len, int value_type,
						  const ndb_item_pointer *ht_ctid);
int			ndb_hnswinsert(ndbhip_hnsw *index, const void *value, size_t value_len, int value_type,
						   const ndb_item_pointer *ht_ctid, int level);
int			ndb_hnsw_level_from_uniform(double r, float ml);

/* ambulkdelete (ivf_am.c:1172-1357, hnsw_am.c:544-720): callback(itemptr, state) != 0 = delete this entry.
 * *tuples_removed as IndexBulkDeleteResult.tuples_removed. */
typedef int (*ndb_bulkdelete_callback) (const ndb_item_pointer *itemptr, void *state);
int			ndb_ivfbulkdelete(ndbhip_ivf *index, ndb_bulkdelete_callback callback, void *callback_state,
							  int64_t *tuples_removed);
int			ndb_hnswbulkdelete(ndbhip_hnsw *index, ndb_bulkdelete_callback callback, void *callback_state,
							   int64_t *tuples_removed);

#ifdef __cplusplus
}
#endif
#endif							/* NDB_AM_H */
