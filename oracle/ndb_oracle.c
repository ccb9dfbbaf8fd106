/*
 * ndb_oracle.c — CPU ORACLE (test infrastructure, see ndb_oracle.h).
 *
 * Literal, PostgreSQL-free restatement of the reference's hot path.  Each
 * function cites the reference lines it follows (relative to
 * /root/reference/NeuronDB/).  Buffer-manager calls (ReadBuffer / LockBuffer /
 * PageGetItem) are replaced by array indexing; every arithmetic statement,
 * loop bound, comparison operator and tie rule is kept as written there.
 *
 * Compile with -ffp-contract=off: `sum += diff * diff` must stay a rounded
 * multiply followed by a rounded add, as on the reference's baseline x86-64
 * build.
 */
#include "ndb_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ================================================================== */
/* Scalar distance recipes                                             */
/* ================================================================== */

/* src/index/ivf_am.c:1550-1592 */
float
ndbo_ivf_distance(const float *vec1, const float *vec2, int dim, int strategy)
{
	int			i;
	float		sum = 0.0f;
	float		dot_product = 0.0f;
	float		norm1 = 0.0f;
	float		norm2 = 0.0f;

	switch (strategy)
	{
		case 2:					/* cosine: :1570-1581 */
			for (i = 0; i < dim; i++)
			{
				dot_product += vec1[i] * vec2[i];
				norm1 += vec1[i] * vec1[i];
				norm2 += vec2[i] * vec2[i];
			}
			norm1 = sqrtf(norm1);
			norm2 = sqrtf(norm2);
			if (norm1 == 0.0f || norm2 == 0.0f)
				return 1.0f;
			return 1.0f - (dot_product / (norm1 * norm2));

		case 3:					/* NOT in the reference (Q2): -dot, fp32 sequential */
			for (i = 0; i < dim; i++)
				dot_product += vec1[i] * vec2[i];
			return -dot_product;

		case 1:					/* L2: :1561-1568 */
		default:				/* :1583-1590 */
			for (i = 0; i < dim; i++)
			{
				float		diff = vec1[i] - vec2[i];

				sum += diff * diff;
			}
			return sqrtf(sum);
	}
}

/* src/index/ivf_am.c:2255-2269 */
float
ndbo_ivf_l2sq(const float *v1, const float *v2, int dim)
{
	float		sum = 0.0;
	int			i;

	for (i = 0; i < dim; i++)
	{
		float		diff = v1[i] - v2[i];

		sum += diff * diff;
	}
	return sum;
}

/* src/index/hnsw_am.c:1301-1345 */
float
ndbo_hnsw_distance(const float *vec1, const float *vec2, int dim, int strategy, int *err)
{
	int			i;
	double		sum = 0.0,
				dot_product = 0.0,
				norm1 = 0.0,
				norm2 = 0.0;

	if (err)
		*err = 0;
	switch (strategy)
	{
		case 1:					/* :1312-1319 — fp32 subtract, widened, fp64 accumulate */
			for (i = 0; i < dim; i++)
			{
				double		d = vec1[i] - vec2[i];

				sum += d * d;
			}
			return (float) sqrt(sum);

		case 2:					/* :1321-1332 — fp32 products widened */
			for (i = 0; i < dim; i++)
			{
				dot_product += vec1[i] * vec2[i];
				norm1 += vec1[i] * vec1[i];
				norm2 += vec2[i] * vec2[i];
			}
			norm1 = sqrt(norm1);
			norm2 = sqrt(norm2);
			if (norm1 == 0.0 || norm2 == 0.0)
				return 2.0f;
			return (float) (1.0f - (dot_product / (norm1 * norm2)));

		case 3:					/* :1334-1337 */
			for (i = 0; i < dim; i++)
				dot_product += vec1[i] * vec2[i];
			return (float) (-dot_product);

		default:				/* :1339-1343 ereport(ERROR) */
			if (err)
				*err = 1;
			return NAN;
	}
}

/* src/vector/vector_distance.c:93-122 (Kahan in double) */
float
ndbo_op_l2_scalar(const float *a, const float *b, int dim)
{
	double		c = 0.0;
	double		sum = 0.0;
	int			i;

	for (i = 0; i < dim; i++)
	{
		double		diff = (double) a[i] - (double) b[i];
		double		y = (diff * diff) - c;
		double		t = sum + y;

		c = (t - sum) - y;
		sum = t;
	}
	return (float) sqrt(sum);
}

/* src/vector/vector_distance.c:145-157 */
float
ndbo_op_ip_scalar(const float *a, const float *b, int dim)
{
	double		sum = 0.0;
	int			i;

	for (i = 0; i < dim; i++)
		sum += (double) a[i] * (double) b[i];
	return (float) (-sum);
}

/* src/vector/vector_distance.c:180-213 */
float
ndbo_op_cosine_scalar(const float *a, const float *b, int dim)
{
	double		dot = 0.0,
				norm_a = 0.0,
				norm_b = 0.0;
	int			i;

	for (i = 0; i < dim; i++)
	{
		double		va = (double) a[i];
		double		vb = (double) b[i];

		dot += va * vb;
		norm_a += va * va;
		norm_b += vb * vb;
	}
	if (norm_a == 0.0 || norm_b == 0.0)
		return 1.0;
	return (float) (1.0 - (dot / (sqrt(norm_a) * sqrt(norm_b))));
}

/*
 * Horizontal sums, src/vector/vector_distance_simd.c:84-137.
 * AVX2 (8 lanes):   s4[j] = v[j] + v[j+4]           (extractf128 + add_ps)
 *                   t[0] = s4[0]+s4[1]; t[2] = s4[2]+s4[3]   (movehdup + add_ps)
 *                   r = t[0] + t[2]                  (movehl + add_ss)
 * AVX-512 (16):     s8[j] = v[j] + v[j+8], then the 8-lane tree.
 */
static float
hsum_lanes(const float *v, int lanes)
{
	float		s8[8];
	float		s4[4];
	float		t0,
				t2;
	int			j;

	if (lanes == 16)
	{
		for (j = 0; j < 8; j++)
			s8[j] = v[j] + v[j + 8];
	}
	else
	{
		for (j = 0; j < 8; j++)
			s8[j] = v[j];
	}
	for (j = 0; j < 4; j++)
		s4[j] = s8[j] + s8[j + 4];
	t0 = s4[0] + s4[1];
	t2 = s4[2] + s4[3];
	return t0 + t2;
}

/* src/vector/vector_distance_simd.c:158-187 (AVX2), 195-225 (AVX-512) */
float
ndbo_op_l2_simd(const float *a, const float *b, int dim, int lanes)
{
	float		acc[16];
	int			i,
				j;
	int			simd_end = (dim / lanes) * lanes;
	float		sum;

	for (j = 0; j < 16; j++)
		acc[j] = 0.0f;
	for (i = 0; i < simd_end; i += lanes)
		for (j = 0; j < lanes; j++)
		{
			float		diff = a[i + j] - b[i + j];
			float		sq = diff * diff;

			acc[j] = acc[j] + sq;
		}
	sum = hsum_lanes(acc, lanes);
	for (i = simd_end; i < dim; i++)
	{
		float		diff = a[i] - b[i];

		sum += diff * diff;
	}
	return sqrtf(sum);
}

/* src/vector/vector_distance_simd.c:233-256 (AVX2), 262-288 (AVX-512): returns +sum */
float
ndbo_op_ip_simd(const float *a, const float *b, int dim, int lanes)
{
	float		acc[16];
	int			i,
				j;
	int			simd_end = (dim / lanes) * lanes;
	float		sum;

	for (j = 0; j < 16; j++)
		acc[j] = 0.0f;
	for (i = 0; i < simd_end; i += lanes)
		for (j = 0; j < lanes; j++)
		{
			float		prod = a[i + j] * b[i + j];

			acc[j] = acc[j] + prod;
		}
	sum = hsum_lanes(acc, lanes);
	for (i = simd_end; i < dim; i++)
		sum += a[i] * b[i];
	return sum;
}

/* src/vector/vector_distance_simd.c:296-340 (AVX2), 346-392 (AVX-512): fmadd lanes */
float
ndbo_op_cosine_simd(const float *a, const float *b, int dim, int lanes)
{
	float		dotv[16],
				nav[16],
				nbv[16];
	int			i,
				j;
	int			simd_end = (dim / lanes) * lanes;
	float		dot,
				norm_a,
				norm_b,
				similarity;

	for (j = 0; j < 16; j++)
		dotv[j] = nav[j] = nbv[j] = 0.0f;
	for (i = 0; i < simd_end; i += lanes)
		for (j = 0; j < lanes; j++)
		{
			float		va = a[i + j];
			float		vb = b[i + j];

			dotv[j] = fmaf(va, vb, dotv[j]);
			nav[j] = fmaf(va, va, nav[j]);
			nbv[j] = fmaf(vb, vb, nbv[j]);
		}
	dot = hsum_lanes(dotv, lanes);
	norm_a = hsum_lanes(nav, lanes);
	norm_b = hsum_lanes(nbv, lanes);
	for (i = simd_end; i < dim; i++)
	{
		float		va = a[i];
		float		vb = b[i];

		dot += va * vb;
		norm_a += va * va;
		norm_b += vb * vb;
	}
	if (norm_a == 0.0f || norm_b == 0.0f)
		return 1.0f;
	similarity = dot / (sqrtf(norm_a) * sqrtf(norm_b));
	return 1.0f - similarity;
}

/* Dispatch: src/vector/vector_distance_simd.c:467-509 */
float
ndbo_op_l2(const float *a, const float *b, int dim, int simd)
{
	if (simd == 16 && dim >= 16)
		return ndbo_op_l2_simd(a, b, dim, 16);
	if (simd == 8 && dim >= 8)
		return ndbo_op_l2_simd(a, b, dim, 8);
	return ndbo_op_l2_scalar(a, b, dim);
}

/* :511-558 — SIMD paths return +sum, the scalar path -(-sum) (Q15) */
float
ndbo_op_ip(const float *a, const float *b, int dim, int simd)
{
	if (simd == 16 && dim >= 16)
		return ndbo_op_ip_simd(a, b, dim, 16);
	if (simd == 8 && dim >= 8)
		return ndbo_op_ip_simd(a, b, dim, 8);
	return -ndbo_op_ip_scalar(a, b, dim);
}

/* :566-613 */
float
ndbo_op_cosine(const float *a, const float *b, int dim, int simd)
{
	if (simd == 16 && dim >= 16)
		return ndbo_op_cosine_simd(a, b, dim, 16);
	if (simd == 8 && dim >= 8)
		return ndbo_op_cosine_simd(a, b, dim, 8);
	return ndbo_op_cosine_scalar(a, b, dim);
}

/* src/types/quantization.c:141-168 (truncating, flush-to-zero) */
uint16_t
ndbo_float4_to_fp16(float f)
{
	uint32_t	u;
	uint16_t	sign;
	uint32_t	mantissa;
	int16_t		exp;

	memcpy(&u, &f, sizeof(uint32_t));
	sign = (u >> 16) & 0x8000;
	mantissa = u & 0x7fffff;
	exp = ((u >> 23) & 0xff) - 127 + 15;

	if (exp <= 0)
		return sign;
	else if (exp >= 31)
		return sign | 0x7c00;
	else
		return sign | (exp << 10) | (mantissa >> 13);
}

/* src/types/quantization.c:170-218.  Subnormal branch kept verbatim, including
 * its exponent arithmetic (quirk Q20: decodes subnormals 2^-10 too small). */
float
ndbo_fp16_to_float(uint16_t h)
{
	uint32_t	sign = (uint32_t) (h & 0x8000) << 16;
	uint32_t	exp = (h & 0x7c00) >> 10;
	uint32_t	mantissa = h & 0x03ff;
	uint32_t	f;
	float		ret;

	if (exp == 0)
	{
		if (mantissa == 0)
			f = sign;
		else
		{
			uint32_t	m = mantissa;
			uint32_t	exponent;

			exp = 1;
			while ((m & 0x0400) == 0)
			{
				m <<= 1;
				exp--;
			}
			m &= 0x03ff;
			exponent = 127 - 15 - (10 - exp);
			f = sign | (exponent << 23) | (m << 13);
		}
	}
	else if (exp == 0x1f)
		f = sign | 0x7f800000 | (mantissa << 13);
	else
	{
		uint32_t	exponent = exp + 127 - 15;

		f = sign | (exponent << 23) | (mantissa << 13);
	}
	memcpy(&ret, &f, 4);
	return ret;
}

/* src/types/quantization.c:1985-2027 */
float
ndbo_halfvec_l2(const uint16_t *a, const uint16_t *b, int dim)
{
	double		sum = 0.0;
	int			i;

	for (i = 0; i < dim; i++)
	{
		float		va = ndbo_fp16_to_float(a[i]);
		float		vb = ndbo_fp16_to_float(b[i]);
		double		diff = (double) va - (double) vb;

		sum += diff * diff;
	}
	return (float) sqrt(sum);
}

/* src/types/quantization.c:2030-2077 */
float
ndbo_halfvec_cosine(const uint16_t *a, const uint16_t *b, int dim)
{
	double		dot = 0.0,
				norm_a = 0.0,
				norm_b = 0.0;
	int			i;

	for (i = 0; i < dim; i++)
	{
		float		va = ndbo_fp16_to_float(a[i]);
		float		vb = ndbo_fp16_to_float(b[i]);

		dot += (double) va * (double) vb;
		norm_a += (double) va * (double) va;
		norm_b += (double) vb * (double) vb;
	}
	if (norm_a == 0.0 || norm_b == 0.0)
		return 1.0;
	return (float) (1.0 - (dot / (sqrt(norm_a) * sqrt(norm_b))));
}

/* src/types/quantization.c:2080-2116 */
float
ndbo_halfvec_ip(const uint16_t *a, const uint16_t *b, int dim)
{
	double		sum = 0.0;
	int			i;

	for (i = 0; i < dim; i++)
	{
		float		va = ndbo_fp16_to_float(a[i]);
		float		vb = ndbo_fp16_to_float(b[i]);

		sum += (double) va * (double) vb;
	}
	return (float) (-sum);
}

/* ================================================================== */
/* ivfExtractVectorData: src/index/ivf_am.c:117-218 (datum images)      */
/* ================================================================== */
int
ndbo_extract_vector(int kind, const unsigned char *datum, float *out, int *out_dim)
{
	int			i;

	switch (kind)
	{
		case 0:					/* vector: :158-167 */
		{
			int16_t		dim;

			memcpy(&dim, datum + 4, 2);
			*out_dim = dim;
			for (i = 0; i < dim; i++)
				memcpy(&out[i], datum + 8 + 4 * (size_t) i, 4);
			return 0;
		}
		case 1:					/* halfvec: :168-177 */
		{
			int16_t		dim;

			memcpy(&dim, datum + 4, 2);
			*out_dim = dim;
			for (i = 0; i < dim; i++)
			{
				uint16_t	h;

				memcpy(&h, datum + 6 + 2 * (size_t) i, 2);
				out[i] = ndbo_fp16_to_float(h);
			}
			return 0;
		}
		case 2:					/* sparsevec: :178-194 */
		{
			int32_t		total_dim, nnz;
			const unsigned char *indices = datum + 12;
			const unsigned char *values;

			memcpy(&total_dim, datum + 4, 4);
			memcpy(&nnz, datum + 8, 4);
			values = indices + 4 * (size_t) nnz;
			*out_dim = total_dim;
			memset(out, 0, (size_t) total_dim * sizeof(float));
			for (i = 0; i < nnz; i++)
			{
				int32_t		ix;

				memcpy(&ix, indices + 4 * (size_t) i, 4);
				if (ix >= 0 && ix < total_dim)
					memcpy(&out[ix], values + 4 * (size_t) i, 4);
			}
			return 0;
		}
		case 3:					/* bit: :195-212 */
		{
			int32_t		nbits;
			const unsigned char *bit_data = datum + 8;

			memcpy(&nbits, datum + 4, 4);
			*out_dim = nbits;
			for (i = 0; i < nbits; i++)
			{
				int			byte_idx = i / 8;
				int			bit_idx = i % 8;
				int			bit_val = (bit_data[byte_idx] >> (8 - 1 - bit_idx)) & 1;

				out[i] = bit_val ? 1.0f : -1.0f;
			}
			return 0;
		}
		default:				/* :213-220 ereport(ERROR) */
			return -1;
	}
}

/* ================================================================== */
/* Top-k by selection sort with index swaps                            */
/* src/index/ivf_am.c:1856-1881 ; src/index/hnsw_am.c:1977-2004        */
/* ================================================================== */
int
ndbo_selection_topk(const float *dist, int64_t n, int k, int64_t *order)
{
	int64_t    *indices;
	int64_t		i,
				j;
	int			actualK = (k < n) ? k : (int) n;

	if (n <= 0)
		return 0;
	indices = (int64_t *) malloc((size_t) n * sizeof(int64_t));
	for (i = 0; i < n; i++)
		indices[i] = i;

	for (i = 0; i < actualK; i++)
	{
		int64_t		bestIdx = i;
		float		bestDist = dist[indices[i]];

		for (j = i + 1; j < n; j++)
		{
			if (dist[indices[j]] < bestDist)
			{
				bestDist = dist[indices[j]];
				bestIdx = j;
			}
		}
		if (bestIdx != i)
		{
			int64_t		temp = indices[i];

			indices[i] = indices[bestIdx];
			indices[bestIdx] = temp;
		}
	}
	for (i = 0; i < actualK; i++)
		order[i] = indices[i];
	free(indices);
	return actualK;
}

/* ================================================================== */
/* IVF                                                                 */
/* ================================================================== */

/* src/index/ivf_am.c:1597-1717 */
int
ndbo_ivf_select_clusters(const ndbo_ivf *ix, const float *query, int nprobe, int *selected)
{
	float	   *clusterDistances;
	int			i,
				j;
	int			nlists;
	int			maxoff = ix->maxoff;
	int			dim = ix->dim;

	/* :1616-1622 — no centroids block */
	if (ix->centroids == NULL || maxoff <= 0)
	{
		for (i = 0; i < nprobe; i++)
			selected[i] = -1;
		return nprobe;
	}

	nlists = ix->nlists;
	if (nprobe > nlists)
		nprobe = nlists;
	/* :1648-1652 */
	if (nlists > maxoff)
		nlists = maxoff;
	if (nprobe > nlists)
		nprobe = nlists;

	clusterDistances = (float *) malloc((size_t) (nlists > 0 ? nlists : 1) * sizeof(float));
	for (i = 0; i < nlists; i++)
		clusterDistances[i] = FLT_MAX;

	/* :1660-1681 */
	for (i = 0; i < nlists && i < maxoff; i++)
	{
		if (ix->centroid_dim != NULL && ix->centroid_dim[i] != dim)
		{
			clusterDistances[i] = FLT_MAX;
			continue;
		}
		clusterDistances[i] = ndbo_ivf_distance(query, ix->centroids + (size_t) i * dim, dim, 1);
	}

	/* :1686-1714 */
	for (i = 0; i < nprobe; i++)
	{
		int			bestIdx = -1;
		float		bestDist = FLT_MAX;

		for (j = 0; j < nlists; j++)
		{
			int			alreadySelected = 0;
			int			k;

			for (k = 0; k < i; k++)
			{
				if (selected[k] == j)
				{
					alreadySelected = 1;
					break;
				}
			}
			if (!alreadySelected && clusterDistances[j] < bestDist)
			{
				bestDist = clusterDistances[j];
				bestIdx = j;
			}
		}
		selected[i] = bestIdx;
	}
	free(clusterDistances);
	return nprobe;
}

/* src/index/ivf_am.c:1722-1909 */
int
ndbo_ivf_collect_candidates(const ndbo_ivf *ix, const float *query, int strategy,
							const int *selected, int nprobe, int k, int64_t max_candidates,
							ndbo_tid *out_tids, float *out_dist, int64_t *n_scored)
{
	int64_t		cap;
	int64_t		candidateCount = 0;
	int64_t		maxCandidates;
	int64_t    *cand_row;
	float	   *candidateDistances;
	int			i;
	int			maxoff = ix->maxoff;
	int			dim = ix->dim;
	int			resultCount = 0;

	/* upper bound on rows met, to size the candidate arrays when uncapped */
	cap = 0;
	for (i = 0; i < nprobe; i++)
	{
		int			c = selected[i];

		if (c < 0 || c >= maxoff)
			continue;
		cap += ix->list_off[c + 1] - ix->list_off[c];
	}
	maxCandidates = (max_candidates > 0) ? max_candidates : cap;	/* :1743 k*10 */
	if (maxCandidates < 1)
		maxCandidates = 1;

	cand_row = (int64_t *) malloc((size_t) maxCandidates * sizeof(int64_t));
	candidateDistances = (float *) malloc((size_t) maxCandidates * sizeof(float));

	/* :1764-1842 */
	for (i = 0; i < nprobe && candidateCount < maxCandidates; i++)
	{
		int			clusterId = selected[i];
		int64_t		r;

		if (clusterId < 0 || clusterId >= maxoff)
			continue;
		/* firstBlock == InvalidBlockNumber ⇔ empty list (:1778) */
		for (r = ix->list_off[clusterId];
			 r < ix->list_off[clusterId + 1] && candidateCount < maxCandidates; r++)
		{
			if (ix->live != NULL && !ix->live[r])
				continue;		/* :1816-1822 */
			candidateDistances[candidateCount] =
				ndbo_ivf_distance(query, ix->vecs + (size_t) r * dim, dim, strategy);
			cand_row[candidateCount] = r;
			candidateCount++;
		}
	}
	if (n_scored)
		*n_scored = candidateCount;

	/* :1847-1899 */
	if (candidateCount > 0)
	{
		int64_t    *order = (int64_t *) malloc((size_t) (k > 0 ? k : 1) * sizeof(int64_t));
		int			actualK = ndbo_selection_topk(candidateDistances, candidateCount, k, order);

		for (i = 0; i < actualK; i++)
		{
			out_tids[i] = ix->tids[cand_row[order[i]]];
			out_dist[i] = candidateDistances[order[i]];
		}
		resultCount = actualK;
		free(order);
	}
	free(cand_row);
	free(candidateDistances);
	return resultCount;
}

/* src/index/ivf_am.c:1976-1999 */
int
ndbo_ivf_search(const ndbo_ivf *ix, const float *query, int strategy, int nprobe, int k,
				int64_t max_candidates, ndbo_tid *out_tids, float *out_dist, int64_t *n_scored)
{
	int		   *selectedClusters;
	int			n;

	if (nprobe < 1)
		return 0;
	selectedClusters = (int *) calloc((size_t) nprobe, sizeof(int));	/* palloc0, :1978 */
	ndbo_ivf_select_clusters(ix, query, nprobe, selectedClusters);
	/* note: collect receives so->nprobe, not the clamped value (:1990-1999) */
	n = ndbo_ivf_collect_candidates(ix, query, strategy, selectedClusters, nprobe, k,
									max_candidates, out_tids, out_dist, n_scored);
	free(selectedClusters);
	return n;
}

/* src/index/ivf_am.c:2274-2294 */
static int
find_nearest_centroid(const float *centroids, int k, int dim, const float *vector)
{
	int			best = 0;
	float		bestDist = FLT_MAX;
	int			c;

	for (c = 0; c < k; c++)
	{
		float		dist = ndbo_ivf_l2sq(vector, centroids + (size_t) c * dim, dim);

		if (dist < bestDist)
		{
			bestDist = dist;
			best = c;
		}
	}
	return best;
}

/* src/index/ivf_am.c:2164-2177 */
void
ndbo_kmeans_assign(const float *data, int n, int dim, const float *centroids, int k,
				   int *assignments, int *counts)
{
	int			i;

	memset(counts, 0, (size_t) k * sizeof(int));
	for (i = 0; i < n; i++)
	{
		assignments[i] = find_nearest_centroid(centroids, k, dim, data + (size_t) i * dim);
		counts[assignments[i]]++;
	}
}

/* src/index/ivf_am.c:2182-2213 */
void
ndbo_kmeans_update(const float *data, int n, int dim, const int *assignments,
				   const int *counts, int k, float *centroids)
{
	int			i,
				j,
				c;

	for (c = 0; c < k; c++)
		for (j = 0; j < dim; j++)
			centroids[(size_t) c * dim + j] = 0.0;

	for (i = 0; i < n; i++)
	{
		c = assignments[i];
		for (j = 0; j < dim; j++)
			centroids[(size_t) c * dim + j] += data[(size_t) i * dim + j];
	}

	for (c = 0; c < k; c++)
	{
		if (counts[c] > 0)
		{
			for (j = 0; j < dim; j++)
				centroids[(size_t) c * dim + j] /= counts[c];
		}
	}
}

/* src/index/ivf_am.c:2218-2233 */
float
ndbo_kmeans_cost(const float *data, int n, int dim, const int *assignments, const float *centroids)
{
	float		cost = 0.0;
	int			i,
				c;

	for (i = 0; i < n; i++)
	{
		c = assignments[i];
		cost += ndbo_ivf_l2sq(data + (size_t) i * dim, centroids + (size_t) c * dim, dim);
	}
	return cost;
}

/* kmeans_init + kmeans_run: src/index/ivf_am.c:2070-2159 */
int
ndbo_kmeans(const float *data, int n, int dim, int k, int max_iter, float threshold,
			float *centroids, int *assignments, int *counts, float *final_cost)
{
	int			i,
				j,
				iter;
	float		prevCost = FLT_MAX;
	float		cost = 0.0f;
	int			iters_done = 0;

	/* :2092-2104 — palloc0 then copy the first k samples */
	for (i = 0; i < k; i++)
		for (j = 0; j < dim; j++)
			centroids[(size_t) i * dim + j] = (i < n) ? data[(size_t) i * dim + j] : 0.0f;

	for (iter = 0; iter < max_iter; iter++)
	{
		ndbo_kmeans_assign(data, n, dim, centroids, k, assignments, counts);
		ndbo_kmeans_update(data, n, dim, assignments, counts, k, centroids);
		cost = ndbo_kmeans_cost(data, n, dim, assignments, centroids);
		iters_done = iter + 1;
		/* :2141 — fabs() of the float difference, compared with the float4 threshold */
		if (fabs(prevCost - cost) < threshold)
			break;
		prevCost = cost;
	}
	if (final_cost)
		*final_cost = cost;
	return iters_done;
}

/* src/index/ivf_am.c:905-935 */
int
ndbo_ivf_assign(const float *centroids, const int *centroid_dim, int nlists, int maxoff,
				int dim, const float *vec, float *min_dist_out)
{
	int			i,
				k;
	int			min_idx = 0;
	float		min_dist = FLT_MAX;
	float		dist;
	float		accum;

	for (i = 0; i < nlists && i < maxoff; i++)
	{
		const float *centroidVector = centroids + (size_t) i * dim;

		if (centroid_dim != NULL && centroid_dim[i] != dim)
			continue;
		accum = 0.0f;
		for (k = 0; k < dim; k++)
		{
			float		diff = vec[k] - centroidVector[k];

			accum += diff * diff;
		}
		dist = sqrtf(accum);
		if (dist < min_dist)
		{
			min_dist = dist;
			min_idx = i;
		}
	}
	if (min_dist_out)
		*min_dist_out = min_dist;
	return min_idx;
}

/* ================================================================== */
/* HNSW                                                                */
/* ================================================================== */

ndbo_hnsw *
ndbo_hnsw_create(int dim, int m, int ef_construction, uint32_t cap_nodes)
{
	ndbo_hnsw  *g = (ndbo_hnsw *) calloc(1, sizeof(ndbo_hnsw));
	size_t		cap = (size_t) cap_nodes + 1;

	g->dim = dim;
	g->m = m;
	g->ef_construction = ef_construction;
	g->entry_point = NDBO_INVALID_BLOCK;
	g->entry_level = -1;		/* hnswInitMetaPage */
	g->max_level = -1;
	g->inserted = 0;
	g->nblocks = 1;				/* meta page */
	g->cap_blocks = (uint32_t) cap;
	g->vecs = (float *) calloc(cap * dim, sizeof(float));
	g->heap_tids = (ndbo_tid *) calloc(cap, sizeof(ndbo_tid));
	g->levels = (int *) calloc(cap, sizeof(int));
	g->ncount = (int16_t *) calloc(cap * NDBO_HNSW_MAX_LEVEL, sizeof(int16_t));
	g->nbrs = (uint32_t *) malloc(cap * NDBO_HNSW_MAX_LEVEL * 2 * (size_t) m * sizeof(uint32_t));
	memset(g->nbrs, 0xFF, cap * NDBO_HNSW_MAX_LEVEL * 2 * (size_t) m * sizeof(uint32_t));
	g->dead = (uint8_t *) calloc(cap, 1);
	return g;
}

void
ndbo_hnsw_free(ndbo_hnsw *g)
{
	if (!g)
		return;
	free(g->vecs);
	free(g->heap_tids);
	free(g->levels);
	free(g->ncount);
	free(g->nbrs);
	free(g->dead);
	free(g);
}

static inline const float *
hnsw_vec(const ndbo_hnsw *g, uint32_t b)
{
	return g->vecs + (size_t) b * g->dim;
}

static inline uint32_t *
hnsw_nbrs(const ndbo_hnsw *g, uint32_t b, int level)
{
	return g->nbrs + ((size_t) b * NDBO_HNSW_MAX_LEVEL + level) * 2 * (size_t) g->m;
}

/* hnswValidateBlockNumber: block must exist and not be the meta page */
static inline int
hnsw_valid_block(const ndbo_hnsw *g, uint32_t b)
{
	return b != NDBO_INVALID_BLOCK && b < g->nblocks && b != 0;
}

/* hnswValidateNeighborCount: clamp to [0, 2m] */
static inline int
hnsw_clamp_ncount(int c, int m)
{
	if (c < 0)
		return 0;
	if (c > m * 2)
		return m * 2;
	return c;
}

/* src/index/hnsw_am.c:1545-2080 */
int
ndbo_hnsw_search(const ndbo_hnsw *g, const float *query, int strategy, int efSearch, int k,
				 uint32_t *out_blocks, float *out_dist, int64_t *n_scored)
{
	uint32_t	current;
	int			currentLevel;
	float		currentDist;
	int			level;
	int			i,
				j,
				l;
	uint32_t   *candidates;
	float	   *candidateDists;
	int			candidateCount = 0;
	uint8_t    *visitedSet;
	uint32_t	numBlocks = g->nblocks;
	int64_t		scored = 0;
	int			dim = g->dim;
	int			m = g->m;
	int			topKCount;
	int			err;

	if (n_scored)
		*n_scored = 0;
	if (g->entry_point == NDBO_INVALID_BLOCK)	/* :1593-1599 */
		return 0;

	current = g->entry_point;
	currentLevel = g->entry_level;
	if (currentLevel < 0 || currentLevel >= NDBO_HNSW_MAX_LEVEL)	/* :1609-1613 */
		currentLevel = 0;

	visitedSet = (uint8_t *) calloc(numBlocks, 1);	/* :1619-1631 */
	candidates = (uint32_t *) malloc((size_t) efSearch * sizeof(uint32_t));
	candidateDists = (float *) malloc((size_t) efSearch * sizeof(float));

	/* greedy descent: :1638-1750 */
	for (level = currentLevel; level > 0; level--)
	{
		int			foundBetter;

		do
		{
			foundBetter = 0;
			if (!hnsw_valid_block(g, current))
				break;
			currentDist = ndbo_hnsw_distance(query, hnsw_vec(g, current), dim, strategy, &err);
			scored++;
			if (g->levels[current] >= level)
			{
				uint32_t   *neighbors = hnsw_nbrs(g, current, level);
				int			neighborCount =
					hnsw_clamp_ncount(g->ncount[(size_t) current * NDBO_HNSW_MAX_LEVEL + level], m);
				uint32_t	node = current;

				(void) node;
				for (i = 0; i < neighborCount; i++)
				{
					float		neighborDist;

					if (neighbors[i] == NDBO_INVALID_BLOCK)
						continue;
					if (!hnsw_valid_block(g, neighbors[i]))
						continue;
					neighborDist = ndbo_hnsw_distance(query, hnsw_vec(g, neighbors[i]), dim, strategy, &err);
					scored++;
					if (neighborDist < currentDist)
					{
						current = neighbors[i];
						currentDist = neighborDist;
						foundBetter = 1;
					}
				}
			}
		} while (foundBetter);
	}

	if (!hnsw_valid_block(g, current))	/* :1752-1763 */
	{
		free(visitedSet);
		free(candidates);
		free(candidateDists);
		if (n_scored)
			*n_scored = scored;
		return 0;
	}

	/* :1765-1831 */
	candidates[0] = current;
	candidateDists[0] = ndbo_hnsw_distance(query, hnsw_vec(g, current), dim, strategy, &err);
	scored++;
	candidateCount = 1;
	if (current < numBlocks)
		visitedSet[current] = 1;

	/* level-0 expansion: :1833-1975 */
	for (i = 0; i < candidateCount && candidateCount < efSearch; i++)
	{
		uint32_t	candidate = candidates[i];
		uint32_t   *neighbors;
		int			neighborCount;

		if (!hnsw_valid_block(g, candidate))
			continue;
		neighbors = hnsw_nbrs(g, candidate, 0);
		neighborCount = hnsw_clamp_ncount(g->ncount[(size_t) candidate * NDBO_HNSW_MAX_LEVEL + 0], m);

		for (j = 0; j < neighborCount; j++)
		{
			float		neighborDist;

			if (neighbors[j] == NDBO_INVALID_BLOCK)
				continue;
			if (!hnsw_valid_block(g, neighbors[j]))
				continue;
			if (neighbors[j] < numBlocks && visitedSet[neighbors[j]])
				continue;

			neighborDist = ndbo_hnsw_distance(query, hnsw_vec(g, neighbors[j]), dim, strategy, &err);
			scored++;
			if (neighbors[j] < numBlocks)
				visitedSet[neighbors[j]] = 1;

			if (candidateCount < efSearch)	/* :1948-1953 */
			{
				candidates[candidateCount] = neighbors[j];
				candidateDists[candidateCount] = neighborDist;
				candidateCount++;
			}
			else				/* :1954-1972 */
			{
				int			worstIdx = 0;
				float		worstDist = candidateDists[0];

				for (l = 1; l < candidateCount && l < efSearch; l++)
				{
					if (candidateDists[l] > worstDist)
					{
						worstDist = candidateDists[l];
						worstIdx = l;
					}
				}
				if (neighborDist < worstDist)
				{
					candidates[worstIdx] = neighbors[j];
					candidateDists[worstIdx] = neighborDist;
				}
			}
		}
	}

	/* :1977-2013 */
	{
		int64_t    *order = (int64_t *) malloc((size_t) (k > 0 ? k : 1) * sizeof(int64_t));

		topKCount = ndbo_selection_topk(candidateDists, candidateCount, k, order);
		for (i = 0; i < topKCount; i++)
		{
			out_blocks[i] = candidates[order[i]];
			out_dist[i] = candidateDists[order[i]];
		}
		free(order);
	}

	free(visitedSet);
	free(candidates);
	free(candidateDists);
	if (n_scored)
		*n_scored = scored;
	return topKCount;
}

/* src/index/hnsw_am.c:1143-1161 with r injected */
/* hnswRemoveNodeFromNeighbor: src/index/hnsw_am.c:2747-2840 */
static void
hnsw_remove_from_neighbor(ndbo_hnsw *g, uint32_t neighborBlkno, uint32_t nodeBlkno, int level)
{
	uint32_t   *neighbors;
	int			neighborCount;
	int			i,
				j;

	if (!hnsw_valid_block(g, neighborBlkno))	/* :2761, and PageIsEmpty(meta page) at :2778 */
		return;
	if (level < 0 || level >= NDBO_HNSW_MAX_LEVEL)
		return;
	neighbors = hnsw_nbrs(g, neighborBlkno, level);
	neighborCount = hnsw_clamp_ncount(g->ncount[(size_t) neighborBlkno * NDBO_HNSW_MAX_LEVEL + level], g->m);
	for (i = 0; i < neighborCount; i++)
	{
		if (neighbors[i] == nodeBlkno)
		{
			for (j = i; j < neighborCount - 1; j++)		/* :2826-2828 */
				neighbors[j] = neighbors[j + 1];
			neighbors[neighborCount - 1] = NDBO_INVALID_BLOCK;
			g->ncount[(size_t) neighborBlkno * NDBO_HNSW_MAX_LEVEL + level]--;	/* the raw count, :2829 */
			break;
		}
	}
}

static int
tid_in_set(const ndbo_tid *tids, int64_t n, ndbo_tid t)
{
	int64_t		i;

	for (i = 0; i < n; i++)
		if (tids[i].bi_hi == t.bi_hi && tids[i].bi_lo == t.bi_lo && tids[i].posid == t.posid)
			return 1;
	return 0;
}

/* ------------------------------------------------------------------ */
/* src/scan/hnsw_scan.c — the best-first search the reference ships but  */
/* never calls.  Restated with its own rules:             */
/*   - distances are compute_l2_distance (:105-118): fp32 sequential,    */
/*     sqrtf, whatever the index's operator class;                       */
/*   - the upper layers are a hill climb that scans the neighbours of     */
/*     the node it STARTED the pass from while `best` moves (:485-636);  */
/*   - layer 0 (:645-844): a binary min-heap of at most 2*efSearch        */
/*     candidates (an insert into a full heap is dropped, :241-242), the  */
/*     entry point enters with distance 0.0 (:668-671), a node counts as  */
/*     visited only once it was offered to the heap, the bound is         */
/*     results[k-1] — the k-th SLOT, not the worst result (:684-686,      */
/*     :744-746) — and results are k unsorted slots where a better node   */
/*     replaces the first worst one (:333-365); they are returned in slot */
/*     order (:826-830).                                                  */
/* ------------------------------------------------------------------ */
typedef struct
{
	uint32_t	block;
	float		distance;
}			scan_elem;

/* hnswInsertCandidate: hnsw_scan.c:235-266 */
static void
scan_heap_insert(scan_elem *h, int *count, int cap, uint32_t block, float distance)
{
	int			i,
				parent;

	if (*count >= cap)
		return;
	i = (*count)++;
	h[i].block = block;
	h[i].distance = distance;
	while (i > 0)
	{
		scan_elem	t;

		parent = (i - 1) / 2;
		if (h[i].distance >= h[parent].distance)
			break;
		t = h[i];
		h[i] = h[parent];
		h[parent] = t;
		i = parent;
	}
}

/* hnswExtractMinCandidate: hnsw_scan.c:271-327 */
static int
scan_heap_pop(scan_elem *h, int *count, uint32_t *block, float *distance)
{
	int			i,
				left,
				right,
				smallest;

	if (*count == 0)
		return 0;
	*block = h[0].block;
	*distance = h[0].distance;
	(*count)--;
	if (*count > 0)
	{
		h[0] = h[*count];
		i = 0;
		for (;;)
		{
			scan_elem	t;

			smallest = i;
			left = 2 * i + 1;
			right = 2 * i + 2;
			if (left < *count && h[left].distance < h[smallest].distance)
				smallest = left;
			if (right < *count && h[right].distance < h[smallest].distance)
				smallest = right;
			if (smallest == i)
				break;
			t = h[i];
			h[i] = h[smallest];
			h[smallest] = t;
			i = smallest;
		}
	}
	return 1;
}

/* a block hnsw_scan.c can read a node from: inside the relation (:562-566, :756-760) and not the meta page,
 * which holds no item (PageIsEmpty, :520-524, :700-704) */
static inline int
scan_readable(const ndbo_hnsw *g, uint32_t b)
{
	return b < g->nblocks && b != 0;
}

/* hnsw_search_layer: src/scan/hnsw_scan.c:379-477 (+ :485-636, :645-844).  Returns resultCount. */
int
ndbo_hnsw_search_layer(const ndbo_hnsw *g, const float *query, int efSearch, int k,
					   uint32_t *out_blocks, float *out_dist, int64_t *n_scored)
{
	uint32_t	currentEntry = g->entry_point;
	int			currentLevel = g->entry_level;
	int			dim = g->dim,
				m = g->m;
	int64_t		scored = 0;
	scan_elem  *cand,
			   *results;
	uint8_t    *visited;
	int			candCount = 0,
				candCap = efSearch * 2,
				resultCount = 0;
	uint32_t	block;
	float		distance;
	int			i;

	if (n_scored)
		*n_scored = 0;
	if (currentEntry == NDBO_INVALID_BLOCK || currentLevel < 0)	/* :396-402 */
		return 0;

	/* Step 1 (:448-457): hnswSearchLayerGreedy per upper layer */
	while (currentLevel > 0)
	{
		uint32_t	best = currentEntry;
		int			changed = 1;

		while (changed)
		{
			const uint32_t *neighbors;
			int			neighborCount;
			float		bestDist;

			changed = 0;
			if (!scan_readable(g, best))	/* :520-524 */
				break;
			if (g->levels[best] < 0 || g->levels[best] >= NDBO_HNSW_MAX_LEVEL)	/* :535-540 */
				break;
			neighbors = hnsw_nbrs(g, best, currentLevel);	/* :549-550: no test of the node's own level */
			neighborCount = hnsw_clamp_ncount(g->ncount[(size_t) best * NDBO_HNSW_MAX_LEVEL + currentLevel], m);
			bestDist = ndbo_ivf_distance(query, hnsw_vec(g, best), dim, 1);
			scored++;
			for (i = 0; i < neighborCount; i++)
			{
				float		neighborDist;

				if (neighbors[i] == NDBO_INVALID_BLOCK)
					continue;
				if (!scan_readable(g, neighbors[i]))
					continue;
				neighborDist = ndbo_ivf_distance(query, hnsw_vec(g, neighbors[i]), dim, 1);
				scored++;
				if (neighborDist < bestDist)	/* :617-622 */
				{
					best = neighbors[i];
					bestDist = neighborDist;
					changed = 1;
				}
			}
		}
		currentEntry = best;
		currentLevel--;
	}

	/* Step 2 (:460-469): hnswSearchLayer0 */
	if (efSearch < 1 || k < 1)
		return 0;
	cand = (scan_elem *) malloc((size_t) candCap * sizeof(scan_elem));
	results = (scan_elem *) malloc((size_t) k * sizeof(scan_elem));
	visited = (uint8_t *) calloc(g->nblocks ? g->nblocks : 1, 1);	/* membership is all hnswIsVisited asks (:201-212) */

	scan_heap_insert(cand, &candCount, candCap, currentEntry, 0.0f);	/* :668-671 */
	if (currentEntry < g->nblocks)
		visited[currentEntry] = 1;

	while (scan_heap_pop(cand, &candCount, &block, &distance))
	{
		const uint32_t *neighbors;
		int			neighborCount;
		float		furthestDist;
		int			j;

		if (resultCount >= k && distance > results[k - 1].distance)	/* :684-686 */
			continue;
		if (!scan_readable(g, block))	/* :700-704 */
			continue;
		if (g->levels[block] < 0 || g->levels[block] >= NDBO_HNSW_MAX_LEVEL)	/* :715-720 */
			continue;
		neighbors = hnsw_nbrs(g, block, 0);
		neighborCount = hnsw_clamp_ncount(g->ncount[(size_t) block * NDBO_HNSW_MAX_LEVEL + 0], m);
		distance = ndbo_ivf_distance(query, hnsw_vec(g, block), dim, 1);	/* :741 */
		scored++;
		furthestDist = (resultCount >= k) ? results[k - 1].distance : FLT_MAX;	/* :744-746 */

		for (j = 0; j < neighborCount; j++)
		{
			float		neighborDist;

			if (neighbors[j] == NDBO_INVALID_BLOCK)
				continue;
			if (!scan_readable(g, neighbors[j]))
				continue;
			if (visited[neighbors[j]])	/* :763-764 */
				continue;
			neighborDist = ndbo_ivf_distance(query, hnsw_vec(g, neighbors[j]), dim, 1);
			scored++;
			if (neighborDist < furthestDist || resultCount < k)	/* :804-810 */
			{
				scan_heap_insert(cand, &candCount, candCap, neighbors[j], neighborDist);
				visited[neighbors[j]] = 1;
			}
		}

		/* hnswAddResult: :333-365 */
		if (resultCount < k)
		{
			results[resultCount].block = block;
			results[resultCount].distance = distance;
			resultCount++;
		}
		else
		{
			int			worstIdx = 0;
			float		worstDist = results[0].distance;

			for (i = 1; i < resultCount; i++)
				if (results[i].distance > worstDist)
				{
					worstDist = results[i].distance;
					worstIdx = i;
				}
			if (distance < worstDist)
			{
				results[worstIdx].block = block;
				results[worstIdx].distance = distance;
			}
		}
	}

	for (i = 0; i < resultCount; i++)	/* :826-830 */
	{
		out_blocks[i] = results[i].block;
		out_dist[i] = results[i].distance;
	}
	free(cand);
	free(results);
	free(visited);
	if (n_scored)
		*n_scored = scored;
	return resultCount;
}

/* src/index/hnsw_am.c:544-720 */
int64_t
ndbo_hnsw_bulkdelete(ndbo_hnsw *g, const ndbo_tid *tids, int64_t n)
{
	uint32_t	blkno;
	int64_t		removed = 0;

	for (blkno = 1; blkno < g->nblocks; blkno++)	/* :586 */
	{
		int			level,
					i,
					nodeLevel = g->levels[blkno];

		if (g->dead[blkno])			/* ItemIdIsDead: :601 */
			continue;
		if (nodeLevel < 0 || nodeLevel >= NDBO_HNSW_MAX_LEVEL)	/* :610-615 */
			continue;
		if (!tid_in_set(tids, n, g->heap_tids[blkno]))	/* callback: :618 */
			continue;
		for (level = 0; level <= nodeLevel; level++)	/* :620-640 */
		{
			const uint32_t *neighbors = hnsw_nbrs(g, blkno, level);
			int			neighborCount = hnsw_clamp_ncount(g->ncount[(size_t) blkno * NDBO_HNSW_MAX_LEVEL + level], g->m);

			for (i = 0; i < neighborCount; i++)
				if (neighbors[i] != NDBO_INVALID_BLOCK && neighbors[i] < g->nblocks)
					hnsw_remove_from_neighbor(g, neighbors[i], blkno, level);
		}
		if (g->entry_point == blkno)	/* :642-690 */
		{
			int			foundNewEntry = 0;

			for (level = nodeLevel; level >= 0 && !foundNewEntry; level--)
			{
				const uint32_t *neighbors = hnsw_nbrs(g, blkno, level);
				int			neighborCount = hnsw_clamp_ncount(g->ncount[(size_t) blkno * NDBO_HNSW_MAX_LEVEL + level], g->m);

				for (i = 0; i < neighborCount && !foundNewEntry; i++)
				{
					const uint32_t nb = neighbors[i];

					/* valid block whose page holds an item (the meta page does not) with a sane level */
					if (hnsw_valid_block(g, nb) && g->levels[nb] >= 0 && g->levels[nb] < NDBO_HNSW_MAX_LEVEL)
					{
						g->entry_point = nb;
						g->entry_level = g->levels[nb];
						foundNewEntry = 1;
					}
				}
			}
			if (!foundNewEntry)
			{
				g->entry_point = NDBO_INVALID_BLOCK;
				g->entry_level = -1;
			}
		}
		g->dead[blkno] = 1;			/* ItemIdSetDead: :693 */
		removed++;
		g->inserted--;
		if (g->inserted < 0)
			g->inserted = 0;
	}
	return removed;
}

int
ndbo_hnsw_level_from_uniform(double r, float ml)
{
	int			level = (int) (-log(r) * ml);

	if (level > NDBO_HNSW_MAX_LEVEL - 1)
		level = NDBO_HNSW_MAX_LEVEL - 1;
	if (level < 0)
		level = 0;
	return level;
}

/* src/index/hnsw_am.c:2091-2670 */
uint32_t
ndbo_hnsw_insert(ndbo_hnsw *g, const float *vector, ndbo_tid heap_tid, int level)
{
	uint32_t	blkno;
	int			dim = g->dim;
	int			m = g->m;
	int			i;
	int			err;

	if (level >= NDBO_HNSW_MAX_LEVEL)
		level = NDBO_HNSW_MAX_LEVEL - 1;
	if (level < 0)
		level = 0;
	if (g->nblocks >= g->cap_blocks)
		return NDBO_INVALID_BLOCK;

	/* Step 3 (:2155-2286): greedy entry search — its result (bestEntry) is never
	 * used afterwards, and it has no side effect; only the distance count differs.
	 * It is omitted here because it cannot influence the graph. */

	/* Step 4 (:2288-2332): new page, one node */
	blkno = g->nblocks;
	g->nblocks++;
	memcpy(g->vecs + (size_t) blkno * dim, vector, (size_t) dim * sizeof(float));
	g->heap_tids[blkno] = heap_tid;
	g->levels[blkno] = level;
	for (i = 0; i < NDBO_HNSW_MAX_LEVEL; i++)
		g->ncount[(size_t) blkno * NDBO_HNSW_MAX_LEVEL + i] = 0;
	memset(hnsw_nbrs(g, blkno, 0), 0xFF, (size_t) NDBO_HNSW_MAX_LEVEL * 2 * m * sizeof(uint32_t));

	/* Step 5 (:2334-2640) */
	{
		int			entryLevel = g->entry_level;
		int			efConstruction = g->ef_construction;

		if (g->entry_point != NDBO_INVALID_BLOCK && entryLevel >= 0)
		{
			int			currentLevel;
			int			maxLevel = (level < entryLevel) ? level : entryLevel;
			int			idx,
						j;
			uint32_t   *candidates = (uint32_t *) malloc((size_t) efConstruction * sizeof(uint32_t));
			float	   *candidateDistances = (float *) malloc((size_t) efConstruction * sizeof(float));
			uint32_t   *selectedNeighbors = (uint32_t *) malloc((size_t) (m > 0 ? m : 1) * sizeof(uint32_t));
			float	   *selectedDistances = (float *) malloc((size_t) (m > 0 ? m : 1) * sizeof(float));

			for (currentLevel = maxLevel; currentLevel >= 0; currentLevel--)
			{
				int			candidateCount;
				int			selectedCount;

				/* :2369-2378 — always L2, ef = k = efConstruction */
				candidateCount = ndbo_hnsw_search(g, vector, 1, efConstruction, efConstruction,
												  candidates, candidateDistances, NULL);
				selectedCount = (m < candidateCount) ? m : candidateCount;

				/* :2391-2414 — selection sort that swaps the arrays themselves */
				for (idx = 0; idx < selectedCount; idx++)
				{
					int			bestIdx = idx;
					float		bestDist = candidateDistances[idx];

					for (j = idx + 1; j < candidateCount; j++)
					{
						if (candidateDistances[j] < bestDist)
						{
							bestDist = candidateDistances[j];
							bestIdx = j;
						}
					}
					if (bestIdx != idx)
					{
						uint32_t	tempBlk = candidates[idx];
						float		tempDist = candidateDistances[idx];

						candidates[idx] = candidates[bestIdx];
						candidateDistances[idx] = candidateDistances[bestIdx];
						candidates[bestIdx] = tempBlk;
						candidateDistances[bestIdx] = tempDist;
					}
					selectedNeighbors[idx] = candidates[idx];
					selectedDistances[idx] = candidateDistances[idx];
				}

				/* :2445-2616 */
				{
					uint32_t   *newNodeNeighbors = hnsw_nbrs(g, blkno, currentLevel);

					for (idx = 0; idx < selectedCount; idx++)
					{
						uint32_t	nb;
						uint32_t   *neighborNeighbors;
						int			neighborNeighborCount;
						int			insertPos;
						int16_t    *nbcount;

						if (idx < m)
						{
							newNodeNeighbors[idx] = selectedNeighbors[idx];
							g->ncount[(size_t) blkno * NDBO_HNSW_MAX_LEVEL + currentLevel] = idx + 1;
						}

						nb = selectedNeighbors[idx];
						/* :2487 — written at currentLevel whatever the neighbour's own level (Q12) */
						neighborNeighbors = hnsw_nbrs(g, nb, currentLevel);
						nbcount = &g->ncount[(size_t) nb * NDBO_HNSW_MAX_LEVEL + currentLevel];
						neighborNeighborCount = hnsw_clamp_ncount(*nbcount, m);

						insertPos = neighborNeighborCount;
						for (j = 0; j < neighborNeighborCount; j++)
						{
							if (neighborNeighbors[j] == NDBO_INVALID_BLOCK)
							{
								insertPos = j;
								break;
							}
						}
						if (insertPos < m * 2)
						{
							neighborNeighbors[insertPos] = blkno;
							if (insertPos >= neighborNeighborCount)
								*nbcount = insertPos + 1;
						}
						/* :2513 `neighborCount > m*2` can never hold after the guarded
						 * append above, so the prune branch (:2516-2612) is unreachable. */
						(void) selectedDistances;
						(void) err;
					}
				}
			}
			free(candidates);
			free(candidateDistances);
			free(selectedNeighbors);
			free(selectedDistances);
		}
	}

	/* Step 6 (:2642-2663) */
	if (g->entry_point == NDBO_INVALID_BLOCK || level > g->entry_level)
	{
		g->entry_point = blkno;
		g->entry_level = level;
	}
	g->inserted++;
	if (level > g->max_level)
		g->max_level = level;
	return blkno;
}
