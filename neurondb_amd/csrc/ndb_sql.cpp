/*
 * ndb_sql.cpp — the reference's SQL-level batch / "GPU" functions for this path over the device library
 * (see include/ndb_sql.h).  Host code only: argument unpacking and the reference's validation order; every
 * distance and every search runs in libndbhip's HIP kernels, and a failing device call is returned as the
 * reference's ERROR would be raised — there is no CPU fallback.
 *
 * Reference paths are relative to NeuronDB/.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "../../include/ndb_sql.h"

int			ndbhip_pages_fail(int code, const char *msg);	/* sets the thread-local error text (ndbhip.hip) */

#define VECTOR_MAX_DIM 16000		/* include/neurondb.h:113 */
#define OPERATOR_RECIPE 2			/* ndbhip_batch_distance: the scalar double kernels of src/vector/vector_distance.c */

static int
failf(int code, const char *fmt, int a, int b = 0)
{
	char		buf[160];

	snprintf(buf, sizeof buf, fmt, a, b);
	return ndbhip_pages_fail(code, buf);
}

/* PG_GETARG_VECTOR_P + NDB_CHECK_VECTOR_VALID (include/neurondb_validation.h:345-356): dim of a vector datum */
static int
vector_dim(const void *datum, size_t len, int *dim)
{
	if (!datum)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "neurondb: vector is NULL");
	int			rc = ndbhip_extract_vector(NDBHIP_TYPE_VECTOR, datum, len, nullptr, 0, dim);

	if (rc)
		return rc;
	if (*dim <= 0 || *dim > 32767)
		return failf(NDBHIP_ERR_INVALID, "neurondb: invalid vector dimension %d", *dim);
	return NDBHIP_OK;
}

static const float *
vector_data(const void *datum)
{
	return (const float *) ((const char *) datum + 8);	/* Vector.data: include/neurondb.h:35-41 */
}

/* the shared body of the three *_batch functions (src/vector/vector_batch.c:37-412) */
static int
distance_batch(const char *fn, int strategy, const void *const *vecs, const size_t *vec_lens, int nvec,
			   const void *query, size_t query_len, float *out)
{
	char		msg[160];
	int			qdim = 0;

	if (!vecs || !vec_lens || !query || !out)	/* :68-71 */
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "vector array and query vector must not be NULL");
	int			rc = vector_dim(query, query_len, &qdim);	/* :65 */

	if (rc)
		return rc;
	if (qdim > VECTOR_MAX_DIM)	/* :73-77 */
		return failf(NDBHIP_ERR_INVALID, "invalid query vector dimension: %d", qdim);
	if (nvec <= 0)				/* :93-97 */
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "vector array must not be empty");

	/* elements that are scored, packed; the others keep elems[i] = (Datum) 0 -> 0.0 (:121-150, :154) */
	std::vector<int> live;
	std::vector<float> rows;

	live.reserve((size_t) nvec);
	for (int i = 0; i < nvec; i++)
	{
		out[i] = 0.0f;
		if (!vecs[i])			/* isnull (:121-126) */
			continue;
		int			dim = 0;

		rc = vector_dim(vecs[i], vec_lens[i], &dim);	/* NDB_CHECK_VECTOR_VALID(vec) (:135) raises for dim <= 0 */
		if (rc)
			return rc;
		if (dim > VECTOR_MAX_DIM || dim != qdim)	/* :136-148 */
			continue;
		live.push_back(i);
		const float *d = vector_data(vecs[i]);

		rows.insert(rows.end(), d, d + dim);
	}
	if (live.empty())
		return NDBHIP_OK;
	std::vector<float> res(live.size());

	/* out = f(vec, query): a = the array element, b = the query, as the reference passes them */
	rc = ndbhip_batch_distance(vector_data(query), rows.data(), res.data(), 1, (int) live.size(), qdim, strategy,
							   OPERATOR_RECIPE);
	if (rc)
		return rc;
	for (size_t j = 0; j < live.size(); j++)
	{
		/* l2_distance / cosine_distance raise on a NaN or infinite result (vector_distance.c:116-119, 219-223);
		 * inner_product_distance does not */
		if (strategy != 3 && (isnan(res[j]) || isinf(res[j])))
		{
			snprintf(msg, sizeof msg, "%s distance calculation resulted in NaN or Infinity",
					 strategy == 1 ? "L2" : "cosine");
			return ndbhip_pages_fail(NDBHIP_ERR_INVALID, msg);
		}
		out[live[j]] = res[j];
	}
	(void) fn;
	return NDBHIP_OK;
}

extern "C" int
ndb_vector_l2_distance_batch(const void *const *vecs, const size_t *vec_lens, int nvec, const void *query,
							 size_t query_len, float *out)
{
	return distance_batch("vector_l2_distance_batch", 1, vecs, vec_lens, nvec, query, query_len, out);
}

extern "C" int
ndb_vector_cosine_distance_batch(const void *const *vecs, const size_t *vec_lens, int nvec, const void *query,
								 size_t query_len, float *out)
{
	return distance_batch("vector_cosine_distance_batch", 2, vecs, vec_lens, nvec, query, query_len, out);
}

/* elems[i] = -inner_product_distance(vec, query) = +dot (vector_batch.c:404; recipe 2 / strategy 3 is +dot) */
extern "C" int
ndb_vector_inner_product_batch(const void *const *vecs, const size_t *vec_lens, int nvec, const void *query,
							   size_t query_len, float *out)
{
	return distance_batch("vector_inner_product_batch", 3, vecs, vec_lens, nvec, query, query_len, out);
}

/* vector_*_distance_gpu (gpu_sql.c:90-160): one pair, the CPU functions' arithmetic */
static int
distance_pair(int strategy, const void *a, size_t a_len, const void *b, size_t b_len, float *out)
{
	int			da = 0, db = 0;

	if (!out)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "bad arguments");
	int			rc = vector_dim(a, a_len, &da);

	if (rc)
		return rc;
	rc = vector_dim(b, b_len, &db);
	if (rc)
		return rc;
	if (da != db)				/* check_dimensions: vector_distance.c:42-47 */
		return failf(NDBHIP_ERR_INVALID, "vector dimensions must match: %d vs %d", da, db);
	float		r = 0.0f;

	rc = ndbhip_batch_distance(vector_data(b), vector_data(a), &r, 1, 1, da, strategy, OPERATOR_RECIPE);
	if (rc)
		return rc;
	if (strategy != 3 && (isnan(r) || isinf(r)))
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, strategy == 1 ?
								 "L2 distance calculation resulted in NaN or Infinity" :
								 "cosine distance calculation resulted in NaN or Infinity");
	*out = strategy == 3 ? -r : r;	/* inner_product_distance returns (float4) (-sum): vector_distance.c:145-157 */
	return NDBHIP_OK;
}

extern "C" int
ndb_vector_l2_distance_gpu(const void *a, size_t a_len, const void *b, size_t b_len, float *out)
{
	return distance_pair(1, a, a_len, b, b_len, out);
}

extern "C" int
ndb_vector_cosine_distance_gpu(const void *a, size_t a_len, const void *b, size_t b_len, float *out)
{
	return distance_pair(2, a, a_len, b, b_len, out);
}

extern "C" int
ndb_vector_inner_product_gpu(const void *a, size_t a_len, const void *b, size_t b_len, float *out)
{
	return distance_pair(3, a, a_len, b, b_len, out);
}

/* ---- ivf_knn_search_gpu / hnsw_knn_search_gpu, repaired (Q19) ---------------------------------- */

/* the non-NULL queries packed as rows of `dim` floats; which[j] = their index in the SQL array */
static int
pack_queries(const char *fn, const void *const *queries, const size_t *query_lens, int nq, int dim,
			 std::vector<float> &rows, std::vector<int> &which)
{
	char		msg[160];

	for (int q = 0; q < nq; q++)
	{
		if (!queries[q])
			continue;
		int			qd = 0;
		int			rc = vector_dim(queries[q], query_lens[q], &qd);

		if (rc)
			return rc;
		if (qd > 10000)			/* gpu_sql.c:988-992, 574-578 */
		{
			snprintf(msg, sizeof msg, "%s: invalid query dimension %d", fn, qd);
			return ndbhip_pages_fail(NDBHIP_ERR_INVALID, msg);
		}
		if (qd != dim)			/* every list entry / node of another dimension is skipped: no rows */
			continue;
		const float *d = vector_data(queries[q]);

		rows.insert(rows.end(), d, d + dim);
		which.push_back(q);
	}
	return NDBHIP_OK;
}

static void
emit_rows(const std::vector<int> &which, int k, const uint8_t *tids6, const float *dist, const int *count,
		  ndb_knn_row *rows, int64_t *nrows)
{
	int64_t		n = 0;

	for (size_t j = 0; j < which.size(); j++)
		for (int i = 0; i < count[j]; i++)
		{
			ndb_knn_row *r = &rows[n++];

			memset(r, 0, sizeof *r);
			r->query_no = which[j];
			memcpy(&r->heaptid, tids6 + ((size_t) j * k + i) * 6, 6);
			r->id = ((int64_t) r->heaptid.bi_hi << 16) | r->heaptid.bi_lo;	/* ItemPointerGetBlockNumber */
			r->distance = dist[(size_t) j * k + i];
		}
	*nrows = n;
}

extern "C" int
ndb_ivf_knn_search_gpu(ndbhip_ivf *index, int strategy, const void *const *queries, const size_t *query_lens,
					   int nq, int k, int nprobe, ndb_knn_row *rows, int64_t *nrows)
{
	int			dim = 0;

	if (!index)					/* "index name cannot be NULL" / "does not exist" (gpu_sql.c:968-972, 1010-1014) */
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf_knn_search_gpu: index cannot be NULL");
	if (!queries || !query_lens || !rows || !nrows || nq < 0)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf_knn_search_gpu: query vector cannot be NULL");
	if (k <= 0 || k > NDBHIP_MAX_K)	/* :978-981 */
		return failf(NDBHIP_ERR_INVALID, "ivf_knn_search_gpu: k must be between 1 and %d", NDBHIP_MAX_K);
	if (nprobe <= 0 || nprobe > 1000)	/* :983-986 */
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf_knn_search_gpu: nprobe must be between 1 and 1000");
	*nrows = 0;
	int			rc = ndbhip_ivf_shape(index, &dim, nullptr);

	if (rc)
		return rc;
	std::vector<float> packed;
	std::vector<int> which;

	rc = pack_queries("ivf_knn_search_gpu", queries, query_lens, nq, dim, packed, which);
	if (rc || which.empty())
		return rc;
	const size_t n = which.size();
	std::vector<uint8_t> tids(n * k * 6);
	std::vector<float> dist(n * k);
	std::vector<int> count(n);

	rc = ndbhip_ivf_search(index, packed.data(), (int) n, strategy, nprobe, k, 0, tids.data(), dist.data(),
						   count.data());
	if (rc)
		return rc;
	emit_rows(which, k, tids.data(), dist.data(), count.data(), rows, nrows);
	return NDBHIP_OK;
}

extern "C" int
ndb_hnsw_knn_search_gpu(ndbhip_hnsw *index, int strategy, const void *const *queries, const size_t *query_lens,
						int nq, int k, int ef_search, ndb_knn_row *rows, int64_t *nrows)
{
	int			dim = 0;

	if (!index)					/* gpu_sql.c:552-556, 590-594 */
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw_knn_search_gpu: index cannot be NULL");
	if (!queries || !query_lens || !rows || !nrows || nq < 0)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw_knn_search_gpu: query vector cannot be NULL");
	if (k <= 0 || k > NDBHIP_MAX_K)	/* :564-567 */
		return failf(NDBHIP_ERR_INVALID, "hnsw_knn_search_gpu: k must be between 1 and %d", NDBHIP_MAX_K);
	if (ef_search <= 0 || ef_search > NDBHIP_MAX_EF)	/* :569-572 */
		return failf(NDBHIP_ERR_INVALID, "hnsw_knn_search_gpu: ef_search must be between 1 and %d", NDBHIP_MAX_EF);
	*nrows = 0;
	int			rc = ndbhip_hnsw_shape(index, &dim, nullptr);

	if (rc)
		return rc;
	std::vector<float> packed;
	std::vector<int> which;

	rc = pack_queries("hnsw_knn_search_gpu", queries, query_lens, nq, dim, packed, which);
	if (rc || which.empty())
		return rc;
	const size_t n = which.size();
	std::vector<uint32_t> blocks(n * k);
	std::vector<uint8_t> tids(n * k * 6);
	std::vector<float> dist(n * k);
	std::vector<int> count(n);

	/* the access method's search for every query: neurondb.ref_compat = 1 the reference's walk, else the `intended`
	 * search under the same strategy — the choice ndb_hnswgettuple makes, on the same (float4) rows, so that the rows
	 * equal an index scan's */
	int			compat = 0;

	(void) ndb_am_get_guc("neurondb.ref_compat", &compat);
	if (compat)
		rc = ndbhip_hnsw_search(index, packed.data(), (int) n, strategy, ef_search, k, blocks.data(), dist.data(),
								count.data(), tids.data(), nullptr);
	else
		rc = ndbhip_hnsw_search_intended(index, packed.data(), (int) n, strategy, ef_search, k, 0, blocks.data(), dist.data(),
										 count.data(), tids.data(), nullptr);
	if (rc)
		return rc;
	emit_rows(which, k, tids.data(), dist.data(), count.data(), rows, nrows);
	return NDBHIP_OK;
}
