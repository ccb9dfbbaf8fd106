/*
 * ndb_am.cpp — the reference's index-AM scan callbacks over the device mirror (see include/ndb_am.h).
 * Host code only: every distance, selection and walk happens in libndbhip's HIP kernels; there is no
 * CPU fallback here — a failing device call is returned as the reference's ERROR would be raised.
 *
 * Reference paths are relative to NeuronDB/.
 */
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/ndb_am.h"
#include "../../include/ndb_service.h"

int			ndbhip_pages_fail(int code, const char *msg);	/* sets the thread-local error text (ndbhip.hip) */

#define IVF_DEFAULT_NPROBE 10		/* src/index/ivf_am.c:60 */
#define IVF_DEFAULT_K 10			/* so->k = 10: ivf_am.c:1421 */
#define HNSW_DEFAULT_EF_SEARCH 64	/* src/index/hnsw_am.c:83 */
#define HNSW_DEFAULT_K 10			/* hnsw_am.c:974 */

/* neurondb.device_service: when set, this backend does not touch the device — its index scans go to the
 * device-owner process through the shared-memory ring (include/ndb_service.h) */
static ndb_client *am_client = nullptr;

/* ---- GUCs ------------------------------------------------------------------------------------ */
static int	guc_ivf_probes = IVF_DEFAULT_NPROBE;
static int	guc_ivf_k = IVF_DEFAULT_K;
static int	guc_hnsw_ef_search = HNSW_DEFAULT_EF_SEARCH;
static int	guc_hnsw_k = HNSW_DEFAULT_K;
static int	guc_ref_compat = 0;

static int *
guc_slot(const char *name, int *lo, int *hi)
{
	if (!name)
		return nullptr;
	if (!strcmp(name, "neurondb.ivf_probes")) { *lo = 1; *hi = NDBHIP_MAX_NPROBE; return &guc_ivf_probes; }
	if (!strcmp(name, "neurondb.ivf_k")) { *lo = 1; *hi = NDBHIP_MAX_K; return &guc_ivf_k; }
	if (!strcmp(name, "neurondb.hnsw_ef_search")) { *lo = 0; *hi = NDBHIP_MAX_EF; return &guc_hnsw_ef_search; }
	if (!strcmp(name, "neurondb.hnsw_k")) { *lo = 0; *hi = NDBHIP_MAX_K; return &guc_hnsw_k; }
	if (!strcmp(name, "neurondb.ref_compat")) { *lo = 0; *hi = 1; return &guc_ref_compat; }
	return nullptr;
}

extern "C" int
ndb_am_use_service(const char *name)
{
	if (am_client)
	{
		ndb_client_disconnect(am_client);
		am_client = nullptr;
	}
	if (!name || !name[0])
		return NDBHIP_OK;
	return ndb_client_connect(name, &am_client);
}

extern "C" int
ndb_am_set_guc(const char *name, int value)
{
	int			lo = 0, hi = 0;
	int		   *slot = guc_slot(name, &lo, &hi);

	if (!slot)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "unrecognized configuration parameter");
	if (value < lo || value > hi)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "value is outside the valid range for the parameter");
	*slot = value;
	return NDBHIP_OK;
}

extern "C" int
ndb_am_get_guc(const char *name, int *value)
{
	int			lo = 0, hi = 0;
	int		   *slot = guc_slot(name, &lo, &hi);

	if (!slot || !value)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "unrecognized configuration parameter");
	*value = *slot;
	return NDBHIP_OK;
}

/* ---- shared scan state (IvfScanOpaqueData: ivf_am.c:266-281; HnswScanOpaqueData: hnsw_am.c:205-216) ---- */
struct ScanOpaque
{
	int			strategy = 1;
	int			nprobe = IVF_DEFAULT_NPROBE;
	int			efSearch = HNSW_DEFAULT_EF_SEARCH;
	int			k = 10;
	bool		firstCall = true;
	int			resultCount = 0;
	int			currentResult = 0;
	bool		haveQuery = false;			/* so->queryVector != NULL */
	std::vector<float> queryVector;
	std::vector<uint8_t> results;			/* heapPtrs, 6 bytes each */
	std::vector<float> distances;
	uint64_t	serviceKey = 0;				/* a scan answered by the device-owner process: which index it is on, */
	uint64_t	serviceVersion = 0;			/* and the generation the backend read for it (include/ndb_service.h) */
};

static ndb_index_scan *
begin_scan(void *index, int nkeys, int norderbys, bool service_ok)
{
	if (!index && !(am_client && service_ok))	/* (a NULL ivf index is the device-owner process's: ndb_am_use_service) */
	{
		ndbhip_pages_fail(NDBHIP_ERR_INVALID, "index is NULL");
		return nullptr;
	}
	ndb_index_scan *scan = (ndb_index_scan *) calloc(1, sizeof(ndb_index_scan));	/* RelationGetIndexScan */

	if (!scan)
		return nullptr;
	scan->indexRelation = index;
	scan->numberOfKeys = nkeys;
	scan->numberOfOrderBys = norderbys;
	scan->xs_orderbynull = 1;
	scan->opaque = new ScanOpaque();
	return scan;
}

/* sk_argument -> so->queryVector (ivfExtractVectorData: ivf_am.c:117-218, hnswExtractVectorData: hnsw_am.c:1402-1519);
 * a NULL argument leaves the previous query in place, as both rescans do (ivf_am.c:1501, hnsw_am.c:941) */
static int
take_query(ScanOpaque *so, const ndb_scan_key *orderbys, int norderbys)
{
	if (norderbys <= 0 || !orderbys || !orderbys[0].sk_argument)
		return NDBHIP_OK;
	int			dim = 0;
	int			rc = ndbhip_extract_vector(orderbys[0].sk_type, orderbys[0].sk_argument, orderbys[0].sk_len, nullptr, 0,
										   &dim);

	if (rc)
		return rc;				/* "unsupported type" / malformed datum: the reference's ERROR */
	so->queryVector.assign((size_t) (dim > 0 ? dim : 1), 0.0f);
	rc = ndbhip_extract_vector(orderbys[0].sk_type, orderbys[0].sk_argument, orderbys[0].sk_len,
							   so->queryVector.data(), dim, &dim);
	if (rc)
		return rc;
	so->queryVector.resize((size_t) dim);
	so->haveQuery = true;
	return NDBHIP_OK;
}

static void
end_scan(ndb_index_scan *scan)
{
	if (!scan)
		return;
	delete (ScanOpaque *) scan->opaque;
	scan->opaque = nullptr;
	free(scan);					/* IndexScanEnd */
}

/* ---- ivf -------------------------------------------------------------------------------------- */

/* ivfbeginscan: src/index/ivf_am.c:1412-1437 */
extern "C" ndb_index_scan *
ndb_ivfbeginscan(ndbhip_ivf *index, int nkeys, int norderbys)
{
	return begin_scan(index, nkeys, norderbys, true);
}

/* the same for a backend without a mirror: the scan is on index `index_key` (relfilenode / OID) whose generation
 * the backend read as `index_version` (ndb_gen_get); the device-owner process answers only if that is the index
 * and the generation it holds, otherwise ndb_ivfgettuple returns NDBHIP_ERR_NODEVICE and the backend runs its
 * CPU scan */
extern "C" ndb_index_scan *
ndb_ivfbeginscan_service(uint64_t index_key, uint64_t index_version, int nkeys, int norderbys)
{
	ndb_index_scan *scan = begin_scan(nullptr, nkeys, norderbys, true);

	if (scan)
	{
		((ScanOpaque *) scan->opaque)->serviceKey = index_key;
		((ScanOpaque *) scan->opaque)->serviceVersion = index_version;
	}
	return scan;
}

/* ivfrescan: src/index/ivf_am.c:1439-1545 */
extern "C" int
ndb_ivfrescan(ndb_index_scan *scan, const ndb_scan_key *keys, int nkeys, const ndb_scan_key *orderbys, int norderbys)
{
	(void) keys;
	(void) nkeys;
	if (!scan || !scan->opaque)	/* so == NULL: return (:1453) */
		return NDBHIP_OK;
	ScanOpaque *so = (ScanOpaque *) scan->opaque;

	so->firstCall = true;		/* :1456-1461 */
	so->currentResult = 0;
	so->resultCount = 0;
	so->results.clear();		/* :1464-1478 */
	so->distances.clear();
	so->strategy = norderbys > 0 && orderbys ? orderbys[0].sk_strategy : 1;	/* :1481-1484 */
	/*
	 * :1487-1513 takes nprobe from the reloptions / meta page and :1421 pins k = 10; the opclasses
	 * register every metric under strategy 1 (Q1).  ref_compat keeps exactly that; otherwise the two
	 * GUCs the reference defines but never reads take effect.
	 */
	if (guc_ref_compat)
	{
		int			np = IVF_DEFAULT_NPROBE;

		if (!scan->indexRelation && am_client)
		{
			const int	snp = ndb_client_meta_nprobe(am_client);	/* the owner published its index's reloptions / meta->nprobe */

			if (snp > 0)
				np = snp;
		}
		else
			(void) ndbhip_ivf_get_nprobe((ndbhip_ivf *) scan->indexRelation, &np);	/* reloptions / meta->nprobe */
		so->strategy = 1;
		so->nprobe = np > 0 ? np : IVF_DEFAULT_NPROBE;	/* :1512-1513 */
		so->k = IVF_DEFAULT_K;
	}
	else
	{
		so->nprobe = guc_ivf_probes > 0 ? guc_ivf_probes : IVF_DEFAULT_NPROBE;	/* :1512-1513 */
		so->k = guc_ivf_k;
	}
	return take_query(so, orderbys, norderbys);	/* :1516-1536 */
}

/* ivfgettuple: src/index/ivf_am.c:1911-2027 */
extern "C" int
ndb_ivfgettuple(ndb_index_scan *scan, int direction)
{
	(void) direction;
	if (!scan || !scan->opaque)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "scan is NULL");
	ScanOpaque *so = (ScanOpaque *) scan->opaque;
	ndbhip_ivf *ix = (ndbhip_ivf *) scan->indexRelation;

	if (!so->haveQuery)			/* :1921-1925 */
		return 0;
	if (so->firstCall && !ix && am_client)
	{
		/* a backend without a mirror of its own: the device-owner process answers (its mirror knows whether the
		 * index is empty; a query of another dimension is "no rows", :1961-1972) */
		int			count = 0;

		so->firstCall = false;
		so->currentResult = 0;
		so->resultCount = 0;
		if ((int) so->queryVector.size() != ndb_client_dim(am_client))
			return 0;
		so->results.assign((size_t) so->k * 6, 0);
		so->distances.assign((size_t) so->k, 0.0f);
		const int	rc = ndb_client_search_index(am_client, so->serviceKey, so->serviceVersion, so->queryVector.data(),
												 so->strategy, so->nprobe, so->k, guc_ref_compat ? (int64_t) so->k * 10 : 0,
												 so->results.data(), so->distances.data(), &count, 30000);

		if (rc)
			return rc;			/* NDBHIP_ERR_NODEVICE: the caller applies neurondb.compute_mode (CPU scan or ERROR) */
		so->resultCount = count;
	}
	else if (so->firstCall)
	{
		int			dim = 0;
		int			rc = ndbhip_ivf_shape(ix, &dim, nullptr);

		if (rc)
			return rc;
		so->firstCall = false;
		so->currentResult = 0;
		so->resultCount = 0;
		if (ndbhip_ivf_nrows(ix) <= 0)	/* "Index is empty" (:1950-1958) */
			return 0;
		if (dim > 0 && (int) so->queryVector.size() != dim)	/* :1961-1972 */
			return 0;
		so->results.assign((size_t) so->k * 6, 0);
		so->distances.assign((size_t) so->k, 0.0f);
		int			count = 0;

		/* ivfSelectClusters + ivfCollectCandidates (:1976-1999) on the device; :1743 caps the candidates
		 * at k * 10 in the reference */
		rc = ndbhip_ivf_search(ix, so->queryVector.data(), 1, so->strategy, so->nprobe, so->k,
							   guc_ref_compat ? (int64_t) so->k * 10 : 0, so->results.data(), so->distances.data(),
							   &count);
		if (rc)
			return rc;
		so->resultCount = count;
	}
	if (so->currentResult < so->resultCount)	/* :2011-2024 */
	{
		memcpy(&scan->xs_heaptid, so->results.data() + (size_t) so->currentResult * 6, 6);
		if (scan->numberOfOrderBys > 0)
		{
			scan->xs_orderbyval = so->distances[(size_t) so->currentResult];
			scan->xs_orderbynull = 0;
		}
		scan->xs_recheckorderby = 0;
		so->currentResult++;
		return 1;
	}
	return 0;
}

/* ivfendscan: src/index/ivf_am.c:2029-2048 */
extern "C" void
ndb_ivfendscan(ndb_index_scan *scan)
{
	end_scan(scan);
}

/* ---- hnsw ------------------------------------------------------------------------------------- */

/* hnswbeginscan: src/index/hnsw_am.c:880-902 */
extern "C" ndb_index_scan *
ndb_hnswbeginscan(ndbhip_hnsw *index, int nkeys, int norderbys)
{
	return begin_scan(index, nkeys, norderbys, false);	/* (the device service answers ivf scans only) */
}

/* hnswrescan: src/index/hnsw_am.c:904-976 */
extern "C" int
ndb_hnswrescan(ndb_index_scan *scan, const ndb_scan_key *keys, int nkeys, const ndb_scan_key *orderbys, int norderbys)
{
	(void) keys;
	(void) nkeys;
	if (!scan || !scan->opaque)
		return NDBHIP_OK;
	ScanOpaque *so = (ScanOpaque *) scan->opaque;

	so->firstCall = true;		/* :913-915 */
	so->currentResult = 0;
	so->resultCount = 0;
	so->strategy = norderbys > 0 && orderbys ? orderbys[0].sk_strategy : 1;	/* :918-921 */
	/* :923-936: the GUC when positive, else the meta page's efSearch */
	so->efSearch = guc_hnsw_ef_search;
	if (so->efSearch <= 0)
	{
		int			efs = HNSW_DEFAULT_EF_SEARCH;

		(void) ndbhip_hnsw_get_meta((ndbhip_hnsw *) scan->indexRelation, nullptr, &efs);
		so->efSearch = efs;
	}
	if (norderbys > 0 && orderbys && orderbys[0].sk_argument)
	{
		int			rc = take_query(so, orderbys, norderbys);	/* :941-971 */

		if (rc)
			return rc;
		so->k = guc_hnsw_k > 0 ? guc_hnsw_k : HNSW_DEFAULT_K;	/* :974, only on this path */
	}
	return NDBHIP_OK;
}

/* hnswgettuple: src/index/hnsw_am.c:978-1056 */
extern "C" int
ndb_hnswgettuple(ndb_index_scan *scan, int direction)
{
	(void) direction;
	if (!scan || !scan->opaque)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "scan is NULL");
	ScanOpaque *so = (ScanOpaque *) scan->opaque;
	ndbhip_hnsw *g = (ndbhip_hnsw *) scan->indexRelation;

	if (so->firstCall)
	{
		if (!so->haveQuery)		/* :992-996 */
			return 0;
		std::vector<uint32_t> blocks((size_t) so->k);
		int			count = 0;

		so->results.assign((size_t) so->k * 6, 0);
		so->distances.assign((size_t) so->k, 0.0f);
		/* hnswSearch (:998-1001); a strategy outside 1..3 is the ERROR of hnswComputeDistance (:1339-1343).
		 * The heapPtr of every result comes back with it: saves the node re-read of :1009-1053.
		 * neurondb.ref_compat = 1: the reference's walk as it stands (BFS-until-ef at level 0, Q10: recall ~ 0 beyond a few
		 * thousand nodes); 0 (default): the `intended` search — greedy descent kept, best-first layer search of
		 * src/scan/hnsw_scan.c:379-483 — ordered by the operator class's metric like hnswSearch's strategy argument
		 * (:918-921), distances of strategies 2 / 3 in hnswComputeDistance's arithmetic. */
		int			rc = guc_ref_compat
			? ndbhip_hnsw_search(g, so->queryVector.data(), 1, so->strategy, so->efSearch, so->k,
								 blocks.data(), so->distances.data(), &count, so->results.data(), nullptr)
			: ndbhip_hnsw_search_intended(g, so->queryVector.data(), 1, so->strategy, so->efSearch, so->k, 0,
										  blocks.data(), so->distances.data(), &count, so->results.data(), nullptr);

		if (rc)
			return rc;
		so->resultCount = count;
		so->firstCall = false;
		so->currentResult = 0;
	}
	if (so->currentResult < so->resultCount)
	{
		memcpy(&scan->xs_heaptid, so->results.data() + (size_t) so->currentResult * 6, 6);
		/* no xs_orderbyvals: hnswgettuple never sets them (Q13) */
		so->currentResult++;
		return 1;
	}
	return 0;
}

/* hnswendscan: src/index/hnsw_am.c:1058-1084 */
extern "C" void
ndb_hnswendscan(ndb_index_scan *scan)
{
	end_scan(scan);
}

/* ---- aminsert / ambulkdelete ------------------------------------------------------------------ */

static int
datum_to_row(const void *value, size_t len, int type, std::vector<float> &row)
{
	int			dim = 0;
	int			rc = ndbhip_extract_vector(type, value, len, nullptr, 0, &dim);

	if (rc)
		return rc;
	row.assign((size_t) (dim > 0 ? dim : 1), 0.0f);
	rc = ndbhip_extract_vector(type, value, len, row.data(), dim, &dim);
	if (!rc)
		row.resize((size_t) dim);
	return rc;
}

/* ivfinsert: src/index/ivf_am.c:797-1167 */
extern "C" int
ndb_ivfinsert(ndbhip_ivf *index, const void *value, size_t value_len, int value_type, const ndb_item_pointer *ht_ctid)
{
	if (!index || !ht_ctid)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!value)					/* isnull[0]: "don't insert NULLs" (:817-818) */
		return 0;
	std::vector<float> row;
	int			rc = datum_to_row(value, value_len, value_type, row);
	int			dim = 0;

	if (rc)
		return rc;
	rc = ndbhip_ivf_shape(index, &dim, nullptr);
	if (rc)
		return rc;
	if ((int) row.size() != dim)	/* centroids of another dimension are skipped (:921-923): nothing can be nearest */
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf: vector dimension does not match the index");
	rc = ndbhip_ivf_insert(index, row.data(), (const uint8_t *) ht_ctid, nullptr);
	return rc ? rc : 1;
}

/* hnswGetRandomLevel's formula (hnsw_am.c:1143-1161): (int)(-log(r) * ml), clamped to [0, 15] */
extern "C" int
ndb_hnsw_level_from_uniform(double r, float ml)
{
	if (!(r > 0.0))
		return 0;
	int			level = (int) (-__builtin_log(r) * ml);

	return level < 0 ? 0 : (level > 15 ? 15 : level);
}

/* hnswinsert: src/index/hnsw_am.c:478-538 */
extern "C" int
ndb_hnswinsert(ndbhip_hnsw *index, const void *value, size_t value_len, int value_type,
			   const ndb_item_pointer *ht_ctid, int level)
{
	if (!index || !ht_ctid)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!value)					/* isnull[0] (:497-498) */
		return 0;
	std::vector<float> row;
	int			rc = datum_to_row(value, value_len, value_type, row);
	int			dim = 0, m = 0;

	if (rc)
		return rc;
	rc = ndbhip_hnsw_shape(index, &dim, &m);
	if (rc)
		return rc;
	if ((int) row.size() != dim)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw: vector dimension does not match the index");
	const int32_t lv = level;
	int			efc = 200;		/* HNSW_DEFAULT_EF_CONSTRUCTION; the mirror carries meta->efConstruction (:2369-2378) */

	(void) ndbhip_hnsw_get_meta(index, &efc, nullptr);
	/* neurondb.ref_compat = 1: hnswInsertNode as it stands (:2091-2670); 0: the `intended` insert — one row = one batch of the
	 * definition's schedule, i.e. the sequential textbook insert (descent kept, every level on its own links, pruning) */
	rc = guc_ref_compat
		? ndbhip_hnsw_insert(index, row.data(), (const uint8_t *) ht_ctid, 1, &lv, efc)
		: ndbhip_hnsw_insert_intended(index, row.data(), (const uint8_t *) ht_ctid, 1, &lv, efc, 64, 1024);
	return rc ? rc : 1;
}

/* ivfbulkdelete: src/index/ivf_am.c:1172-1357 */
extern "C" int
ndb_ivfbulkdelete(ndbhip_ivf *index, ndb_bulkdelete_callback callback, void *callback_state, int64_t *tuples_removed)
{
	if (!index || !callback)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "bad arguments");
	const int64_t n = ndbhip_ivf_nrows(index);
	std::vector<uint8_t> tids((size_t) (n > 0 ? n : 1) * 6), hit;
	int			rc = ndbhip_ivf_export(index, nullptr, nullptr, nullptr, tids.data());

	if (rc)
		return rc;
	for (int64_t i = 0; i < n; i++)	/* every live entry, list by list, chain order (:1226-1290) */
	{
		ndb_item_pointer ip;

		memcpy(&ip, tids.data() + (size_t) i * 6, 6);
		if (callback(&ip, callback_state))
			hit.insert(hit.end(), tids.begin() + (size_t) i * 6, tids.begin() + (size_t) i * 6 + 6);
	}
	int64_t		removed = 0;

	rc = ndbhip_ivf_delete(index, hit.empty() ? nullptr : hit.data(), (int64_t) (hit.size() / 6), &removed);
	if (tuples_removed)
		*tuples_removed = removed;
	return rc;
}

/* hnswbulkdelete: src/index/hnsw_am.c:544-720 */
extern "C" int
ndb_hnswbulkdelete(ndbhip_hnsw *index, ndb_bulkdelete_callback callback, void *callback_state,
				   int64_t *tuples_removed)
{
	if (!index || !callback)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "bad arguments");
	uint32_t	nb = 0;
	int			rc = ndbhip_hnsw_export(index, &nb, nullptr, nullptr, nullptr, nullptr, nullptr);

	if (rc)
		return rc;
	std::vector<uint8_t> tids((size_t) (nb > 0 ? nb : 1) * 6), dead(nb > 0 ? nb : 1), hit;

	rc = ndbhip_hnsw_export_rows(index, nullptr, tids.data(), dead.data());
	if (rc)
		return rc;
	for (uint32_t b = 1; b < nb; b++)	/* blocks in order; dead line pointers are not offered again (:586-601) */
	{
		ndb_item_pointer ip;

		if (dead[b])
			continue;
		memcpy(&ip, tids.data() + (size_t) b * 6, 6);
		if (callback(&ip, callback_state))
			hit.insert(hit.end(), tids.begin() + (size_t) b * 6, tids.begin() + (size_t) b * 6 + 6);
	}
	int64_t		removed = 0;

	rc = ndbhip_hnsw_delete(index, hit.empty() ? nullptr : hit.data(), (int64_t) (hit.size() / 6), &removed);
	if (tuples_removed)
		*tuples_removed = removed;
	return rc;
}
