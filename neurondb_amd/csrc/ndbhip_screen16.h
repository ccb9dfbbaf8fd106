/*
 * ndbhip_screen16.h — the screened list scan on fp16 matrix cores (part of ndbhip.hip's translation unit).
 *
 * What it computes is ivfCollectCandidates (src/index/ivf_am.c:1722-1909) for a batch of >= 128 queries:
 * every probed list entry scored with ivfComputeDistance (ivf_am.c:1550-1592), the k smallest by the
 * reference's selection sort — ids, ranks and float4 bits identical to the reference's.  How:
 *
 *   k_s16_row_prep   once per version of the mirror: |x|^2 (fp64 -> fp32), scale exponent, the row split into
 *                    two fp16 planes  x 2^(14-e) = hi + lo  (same bytes per row as the fp32 row)
 *   k_s16_qprep      per batch: the same for every query
 *   k_s16_seed       per query: the reference's own arithmetic for its first 64 candidates; their k-th
 *                    smallest distance is an upper bound thr of the query's k-th distance
 *   k_s16_sweep      the bound pass: tiles of 128 rows x 128 queries, operands DMA'd into LDS
 *                    (global_load_lds_dwordx4), three v_mfma_f32_32x32x16_f16 per 16 dimensions
 *                    (hi*hi + hi*lo + lo*hi), fp32 accumulate; a ~ |q - x|^2 with |a - |q - x|^2| <= E_q
 *                    (ndbhip_common.h derives E_q); a candidate is EMITTED (position, a) unless
 *                    a - E_q > thr^2 (1 + m), i.e. unless it provably lies beyond the k-th distance.
 *                    Nothing else is written: no [nq x candidates] distance array.
 *   k_s16_finalize   per query: tighten thr with the k-th smallest emitted a (+ E_q), give the survivors the
 *                    reference's own sequential arithmetic (one lane per candidate), and replay the
 *                    reference's selection sort over them (block_sort_cut / block_replay_emit).
 *
 * For L2, two cheaper bounds run first (docs/DESIGN_rounds_1_3.md 3e): k_s16_pair_prune drops a (query, probe) pair whose list lies
 * wholly beyond the threshold (|q - centroid| - list radius), and on mirrors whose long lists were regrouped into
 * sublists (ivf_s16_build_sublists: rows reordered inside the planes only, pos_of = their place in the list) the
 * pair expands just to the sublists that survive the same test against the sublist centres (k_sub_pairs; the
 * centre distances come from this sweep's MODE 3), and the seeds are the nearest sublist's rows (k_s16_seed_sub).
 *
 * Every value the selection can pick or tie with is exact; everything else is provably larger, so the result
 * is the exact path's bit for bit.  If a query emits more than its record capacity (adversarial data: nearly
 * everything ties) the host reruns the batch through the older screened path, which has no capacity.
 */
#ifndef NDBHIP_SCREEN16_H
#define NDBHIP_SCREEN16_H

typedef _Float16 ndb_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 ndb_h2 __attribute__((ext_vector_type(2)));

#define S16_QT 128				/* queries per tile */
#define S16_CH 32				/* dimensions per staged chunk (two MFMA k-steps) */
#define S16_SEED 64				/* candidates scored exactly per query for the first threshold */
#define S16_SURV_CAP 256		/* survivors per query the finalize stage holds in LDS (typically 10-30; more: fall back) */
#define S16_NB_LOG2 7
#define S16_NB (1 << S16_NB_LOG2)	/* hash buckets of candidate positions per query (>= 2 x the largest k) */

/* per-query record the sweep reads (one per member of the tile, in LDS) */
struct S16Q
{
	float		q2;				/* |q|^2 */
	float		thrE;			/* emit unless a > thrE */
	int			eq;				/* scale exponent of the query */
	uint32_t	la;				/* first local candidate position of this (query, probe) */
	uint32_t	nrow;			/* rows of the list visible to this (query, probe); 0 = padding member */
	uint32_t	qid;
};

__device__ __forceinline__ double
wave_sum_f64(double v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		const uint32_t lo = __shfl_xor((uint32_t) __double2loint(v), off, 64);
		const uint32_t hi = __shfl_xor((uint32_t) __double2hiint(v), off, 64);

		v += __hiloint2double((int) hi, (int) lo);
	}
	return v;
}

/* e with 2^(e-1) <= sqrt(s) < 2^e (s > 0, finite); 0 for s == 0 */
__device__ __forceinline__ int
s16_exponent(double s)
{
	if (!(s > 0.0))
		return 0;
	return ilogb(__builtin_sqrt(s)) + 1;
}

/* v = x 2^(14-e) split into hi + lo halves (ndbhip_common.h (2), (3)); the scaling runs in fp64 so that no
 * exponent of an fp32 vector can overflow it */
__device__ __forceinline__ void
s16_split(float x, int e, _Float16 &hi, _Float16 &lo)
{
	const float v = (float) ldexp((double) x, 14 - e);

	hi = (_Float16) v;
	lo = (_Float16) (v - (float) hi);
}

/*
 * One wave per row.  The planes are stored the way the sweep wants them in LDS, so that a 32-row block's share
 * of a chunk is ONE contiguous piece the DMA copies verbatim (no per-lane gather, full DRAM pages instead of
 * 128 bytes out of every 3 KB row):
 *
 *   planes[(blk * nchunk + c) * ROW_BLK + image of the block's 32 rows for chunk c]
 *
 * blk = blk_off[list] + (position in the list) / 32 (every list starts a new block; the tail of a list's last
 * block stays zero), nchunk = dimp / 32 with dimp = dim rounded up to 64 (the tail dimensions are zero).  Image,
 * float4 rows: [r][8 slots of 16 bytes] = 4096 bytes, logical slot s = 2 * kstep + khalf for the hi plane,
 * 4 + 2 * kstep + khalf for the lo plane, stored at slot s ^ ((r >> 1) & 7).  H16 rows (halfvec mirror): the
 * mirror's own fp16 values are the hi plane and there is no lo plane: [r][4 slots] = 2048 bytes, slot
 * s ^ ((r >> 2) & 3); decoded like fp16_to_float for the norm (SUBFIX: quirk Q20), scale exponent 14.
 * rn2[row] = |x|^2, NaN for a row whose norm is not a finite fp32 (the sweep emits every candidate of such a
 * row, so it only affects itself: the reference's arithmetic decides).  rexp[row] = e.  xmax_bits: largest
 * finite rn2 (bits order like values for non-negative floats).
 */
template <int H16>
__global__ __launch_bounds__(256) void
k_s16_row_prep(const void *__restrict__ vecs, int64_t nrows, int dim, int dimp, const int64_t *__restrict__ loc_off,
			   const uint32_t *__restrict__ blk_off, int ncent, unsigned char *__restrict__ planes,
			   float *__restrict__ rn2, int16_t *__restrict__ rexp, uint32_t *__restrict__ xmax_bits,
			   const int64_t *__restrict__ perm = nullptr /* plane row -> mirror row (sublists: rows regrouped inside a list) */ )
{
	const int	lane = threadIdx.x & 63;
	const int64_t prow = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);	/* place in the planes' order */

	if (prow >= nrows)
		return;
	const int64_t row = perm ? perm[prow] : prow;						/* the row it holds */
	double		s = 0.0;

	if constexpr (H16 != 0)
	{
		const uint16_t *x = (const uint16_t *) vecs + (size_t) row * dim;

		for (int i = lane; i < dim; i += 64)
		{
			const float v = (H16 == 1) ? h2f_ref(x[i]) : __half2float(__ushort_as_half(x[i]));

			s += (double) v * (double) v;
		}
	}
	else
	{
		const float *x = (const float *) vecs + (size_t) row * dim;

		for (int i = lane; i < dim; i += 64)
			s += (double) x[i] * (double) x[i];
	}
	s = wave_sum_f64(s);
	const bool	ok = s <= 3.0e38;	/* false for NaN, inf and sums beyond fp32 */
	const float n2 = ok ? (float) s : __uint_as_float(0x7FC00000u);
	const int	e = ok ? s16_exponent(s) : 0;

	if (lane == 0)
	{
		rn2[prow] = n2;
		if (rexp)
			rexp[prow] = (int16_t) e;
		/* one address for every row of the index: only a row that would raise the maximum goes to the atomic unit
		 * (a stale read is merely smaller; a million same-address atomics cost 10 ms) */
		if (ok && __float_as_uint(n2) > __atomic_load_n(xmax_bits, __ATOMIC_RELAXED))
			atomicMax(xmax_bits, __float_as_uint(n2));
	}
	/* the row's list (largest L with loc_off[L] <= row) and its place in the blocked planes */
	int			lo = 0, hi = ncent;

	while (hi - lo > 1)
	{
		const int	mid = (lo + hi) >> 1;

		if (loc_off[mid] <= prow)
			lo = mid;
		else
			hi = mid;
	}
	while (lo + 1 < ncent && loc_off[lo + 1] <= prow)
		lo++;
	const uint32_t pos = (uint32_t) (prow - loc_off[lo]);
	const size_t blk = (size_t) blk_off[lo] + (pos >> 5);
	const int	rr = (int) (pos & 31u);
	const int	nchunk = dimp / 32;
	constexpr int ROW_BLK = H16 ? 2048 : 4096;
	constexpr int ROW_CHUNK = H16 ? 64 : 128;
	unsigned char *img = planes + blk * (size_t) nchunk * ROW_BLK + (size_t) rr * ROW_CHUNK;

	/* lane handles the element pairs (2p, 2p + 1): pair j = p % 16 of chunk c = p / 16 lives in logical slot j / 4 */
	for (int p = lane; p < dimp / 2; p += 64)
	{
		const int	i = 2 * p, c = p >> 4, j = p & 15;
		unsigned char *rowimg = img + (size_t) c * ROW_BLK;

		if constexpr (H16 != 0)
		{
			const uint16_t *x = (const uint16_t *) vecs + (size_t) row * dim;
			const uint32_t v = (i < dim ? (uint32_t) x[i] : 0u) | ((i + 1 < dim ? (uint32_t) x[i + 1] : 0u) << 16);

			*reinterpret_cast<uint32_t *>(rowimg + 16 * ((j >> 2) ^ ((rr >> 2) & 3)) + 4 * (j & 3)) = v;
		}
		else
		{
			const float *x = (const float *) vecs + (size_t) row * dim;
			_Float16	h0 = 0, l0 = 0, h1 = 0, l1 = 0;

			if (ok && i < dim)
				s16_split(x[i], e, h0, l0);
			if (ok && i + 1 < dim)
				s16_split(x[i + 1], e, h1, l1);
			ndb_h2		h, l;

			h.x = h0; h.y = h1; l.x = l0; l.y = l1;
			*reinterpret_cast<ndb_h2 *>(rowimg + 16 * ((j >> 2) ^ ((rr >> 1) & 7)) + 4 * (j & 3)) = h;
			*reinterpret_cast<ndb_h2 *>(rowimg + 16 * ((4 + (j >> 2)) ^ ((rr >> 1) & 7)) + 4 * (j & 3)) = l;
		}
	}
}

/*
 * Radius of every list around its centroid: lrad_bits[L] = bits of the largest |x - c_L| over the rows of L held
 * here, in fp64, rounded to fp32 and up by 2^-20 (bits order like values for non-negative floats); +inf when a
 * row or the centroid is not finite.  One wave per row.
 */
template <int H16>
__global__ __launch_bounds__(256) void
k_s16_list_radius(const void *__restrict__ vecs, int64_t nrows, int dim, const int64_t *__restrict__ loc_off, int ncent,
				  const float *__restrict__ cents, uint32_t *__restrict__ lrad_bits)
{
	const int	lane = threadIdx.x & 63;
	const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);

	if (row >= nrows)
		return;
	int			lo = 0, hi = ncent;

	while (hi - lo > 1)
	{
		const int	mid = (lo + hi) >> 1;

		if (loc_off[mid] <= row)
			lo = mid;
		else
			hi = mid;
	}
	while (lo + 1 < ncent && loc_off[lo + 1] <= row)
		lo++;
	const float *c = cents + (size_t) lo * dim;
	double		s = 0.0;

	for (int i = lane; i < dim; i += 64)
	{
		float		v;

		if constexpr (H16 != 0)
		{
			const uint16_t h = ((const uint16_t *) vecs)[(size_t) row * dim + i];

			v = (H16 == 1) ? h2f_ref(h) : __half2float(__ushort_as_half(h));
		}
		else
			v = ((const float *) vecs)[(size_t) row * dim + i];
		const double d = (double) v - (double) c[i];

		s += d * d;
	}
	s = wave_sum_f64(s);
	if (lane == 0)
	{
		const double r = __builtin_sqrt(s) * (1.0 + 9.5367431640625e-7);
		const uint32_t bits = (r <= 3.0e38) ? __float_as_uint(__double2float_ru(r)) : 0x7F800000u;	/* NaN fails the test: +inf */

		if (bits > __atomic_load_n(&lrad_bits[lo], __ATOMIC_RELAXED))
			atomicMax(&lrad_bits[lo], bits);
	}
}

/*
 * drop[q][p] = 1 when no row of the p-th probed list can be among query q's k nearest: every row x of list L
 * satisfies |q - x| >= |q - c_L| - radius_L (triangle inequality, real numbers), so if that lower bound, squared,
 * exceeds thrE = thr^2 (1 + m) + E (qthr[q].x, the value the sweep compares a candidate's bound with) the rows'
 * real squared distances D do too, hence D > thr^2 (1 + m) and the reference's float4 distance exceeds thr
 * (ndbhip_common.h (7)).  |q - c_L| is computed in fp64 (error ~1e-13 relative) and shaved by 1e-9; the radius is
 * rounded up by k_s16_list_radius.
 */
__global__ __launch_bounds__(256) void
k_s16_pair_prune(const float *__restrict__ queries, uint32_t nq, int npr, int dim, const int *__restrict__ probes, int ncent,
				 const float *__restrict__ cents, const uint32_t *__restrict__ lrad_bits, const float2 *__restrict__ qthr,
				 const unsigned int *__restrict__ active, uint8_t *__restrict__ drop,
				 float *__restrict__ pdist = nullptr /* [nq][npr]: |q - c_L|, rounded down (sublists decide with it later) */ )
{
	/* one block per query: wave w takes probes w, w + 4, ...; the query's elements stay in registers (dim <= 768:
	 * 12 per lane) or are re-read from L1 */
	const int	lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t q = blockIdx.x;

	if (active && !active[q])
		return;					/* (round 1 only rebuilds the pairs of the queries still active) */
	const float *x = queries + (size_t) q * dim;
	const double te = (double) qthr[q].x;
	float		xr[12];
	const bool	inreg = dim <= 768;

	if (inreg)
#pragma unroll
		for (int j = 0; j < 12; j++)
			xr[j] = lane + 64 * j < dim ? x[lane + 64 * j] : 0.0f;
	for (int p = wave; p < npr; p += 4)
	{
		const int	L = probes[(size_t) q * npr + p];
		bool		out = false;

		if (L >= 0 && L < ncent)
		{
			const float *c = cents + (size_t) L * dim;
			double		s = 0.0;

			if (inreg)
			{
#pragma unroll
				for (int j = 0; j < 12; j++)
				{
					const double t = (double) xr[j] - (double) (lane + 64 * j < dim ? c[lane + 64 * j] : 0.0f);

					s += t * t;
				}
			}
			else
				for (int d = lane; d < dim; d += 64)
				{
					const double t = (double) x[d] - (double) c[d];

					s += t * t;
				}
			s = wave_sum_f64(s);
			const double rad = (double) __uint_as_float(lrad_bits[L]);
			const double lb = __builtin_sqrt(s) * (1.0 - 1e-9) - rad;

			out = lb > 0.0 && lb * lb * (1.0 - 1e-9) > te && te >= 0.0;		/* (NaN or inf anywhere: false) */
			if (pdist && lane == 0)
				pdist[(size_t) q * npr + p] = __double2float_rd(__builtin_sqrt(s) * (1.0 - 1e-9));
		}
		else if (pdist && lane == 0)
			pdist[(size_t) q * npr + p] = 0.0f;		/* (no such list: never excluded) */
		if (lane == 0)
			drop[(size_t) q * npr + p] = out ? 1 : 0;
	}
}

/* statistics of one batch: counters[0] += pairs dropped, counters[1] += candidate rows of the pairs kept
 * (grid-stride; two atomics per block) */
__global__ __launch_bounds__(256) void
k_s16_prune_stats(const uint8_t *__restrict__ drop, const uint32_t *__restrict__ loc_cand_off, uint32_t nq, int npr,
				  unsigned long long *__restrict__ counters, int count_rows)
{
	__shared__ unsigned long long sd[256], sr[256];
	unsigned long long nd = 0, nr = 0;

	for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < nq; q += gridDim.x * 256)
	{
		const uint32_t *co = loc_cand_off + (size_t) q * (npr + 1);

		for (int p = 0; p < npr; p++)
			if (drop && drop[(size_t) q * npr + p])
				nd++;
			else
				nr += co[p + 1] - co[p];
	}
	sd[threadIdx.x] = nd;
	sr[threadIdx.x] = nr;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1)
	{
		if ((int) threadIdx.x < o)
		{
			sd[threadIdx.x] += sd[threadIdx.x + o];
			sr[threadIdx.x] += sr[threadIdx.x + o];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0)
	{
		atomicAdd(&counters[0], sd[0]);
		if (count_rows)
			atomicAdd(&counters[1], sr[0]);
	}
}

/* sublists: rows the sweep multiplies = sum over sublists of (pairs kept) x (rows); one block */
__global__ __launch_bounds__(256) void
k_s16_swept_rows(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ sub_len, int nsub,
				 unsigned long long *__restrict__ counter)
{
	__shared__ unsigned long long sr[256];
	unsigned long long nr = 0;

	for (int s2 = threadIdx.x; s2 < nsub; s2 += 256)
		nr += (unsigned long long) cnt[s2] * sub_len[s2];
	sr[threadIdx.x] = nr;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1)
	{
		if ((int) threadIdx.x < o)
			sr[threadIdx.x] += sr[threadIdx.x + o];
		__syncthreads();
	}
	if (threadIdx.x == 0)
		atomicAdd(counter, sr[0]);
}

template <int R> __device__ __forceinline__ float s16_e(int dim, float q2, float x2max, bool sub);

/* ---------------------------------------------------------------------------------------------------------
 * Sublists.  A list of the reference's index can be huge and unrelated to any one query (its k-means runs on the
 * table's first 10 000 rows: components without a sample pile up in a few lists every query probes).  The rows
 * of such a list are regrouped INSIDE THE PLANES — which are this library's own copy — by their nearest of a few
 * sample rows of the list ("sublists"), every sublist with a centre and a radius; a (query, probe) pair then
 * expands only to the sublists the triangle inequality cannot exclude, and the query's first threshold comes from
 * the rows of the sublist nearest to it (k_s16_seed_sub).  Positions, candidate caps and ties stay
 * defined on the row's index in its list (pos_of), so nothing downstream changes.
 * --------------------------------------------------------------------------------------------------------- */

/* list-major identity: plane row = mirror row */
__global__ void
k_s16_identity_perm(int64_t nrows, const int64_t *__restrict__ loc_off, int ncent, int64_t *__restrict__ perm,
					uint32_t *__restrict__ pos_of)
{
	const int64_t row = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;

	if (row >= nrows)
		return;
	int			lo = 0, hi = ncent;

	while (hi - lo > 1)
	{
		const int	mid = (lo + hi) >> 1;

		if (loc_off[mid] <= row)
			lo = mid;
		else
			hi = mid;
	}
	while (lo + 1 < ncent && loc_off[lo + 1] <= row)
		lo++;
	perm[row] = row;
	pos_of[row] = (uint32_t) (row - loc_off[lo]);
}

/* a regrouped list: sorted[j] = index in the list of the row that comes j-th in the planes */
__global__ void
k_s16_sub_perm(const uint64_t *__restrict__ sorted, int64_t n, int64_t first_row, int64_t *__restrict__ perm,
			   uint32_t *__restrict__ pos_of)
{
	const int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;

	if (j >= n)
		return;
	perm[first_row + j] = first_row + (int64_t) sorted[j];
	pos_of[first_row + j] = (uint32_t) sorted[j];
}

/* radius of every sublist around its centre (cptr[sub]): like k_s16_list_radius, over the planes' order */
__global__ __launch_bounds__(256) void
k_s16_sub_radius(const float *__restrict__ vecs, int64_t nrows, int dim, const int64_t *__restrict__ sub_loc, int nsub,
				 const int64_t *__restrict__ perm, const float *const *__restrict__ cptr, uint32_t *__restrict__ rad_bits)
{
	const int	lane = threadIdx.x & 63;
	const int64_t prow = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);

	if (prow >= nrows)
		return;
	int			lo = 0, hi = nsub;

	while (hi - lo > 1)
	{
		const int	mid = (lo + hi) >> 1;

		if (sub_loc[mid] <= prow)
			lo = mid;
		else
			hi = mid;
	}
	while (lo + 1 < nsub && sub_loc[lo + 1] <= prow)
		lo++;
	const float *x = vecs + (size_t) perm[prow] * dim, *c = cptr[lo];
	double		s = 0.0;

	for (int i = lane; i < dim; i += 64)
	{
		const double d = (double) x[i] - (double) c[i];

		s += d * d;
	}
	s = wave_sum_f64(s);
	if (lane == 0)
	{
		const double r = __builtin_sqrt(s) * (1.0 + 9.5367431640625e-7);
		const uint32_t bits = (r <= 3.0e38) ? __float_as_uint(__double2float_ru(r)) : 0x7F800000u;

		if (bits > __atomic_load_n(&rad_bits[lo], __ATOMIC_RELAXED))
			atomicMax(&rad_bits[lo], bits);
	}
}

/* can sublist `s` be left out for a query whose squared threshold is te?  d = the reference's float4 L2 distance
 * of the query to the sublist's centre (within (dim + 3) 2^-24 of the real one: shaved by 1e-3), rad = the
 * sublist's radius, rounded up.  Same argument as k_s16_pair_prune. */
/* the same with a = the matrix-core sweep's |q - c|^2 (k_s16_sweep MODE 3), |a - |q - c|^2| <= e */
__device__ __forceinline__ bool
s16_sub_excluded_a(float a, float e, uint32_t rad_bits, float te)
{
	const double alo = (double) a - (double) e * (1.0 + 1e-6);
	const double lb = __builtin_sqrt(alo > 0.0 ? alo : 0.0) - (double) __uint_as_float(rad_bits);

	return lb > 0.0 && lb * lb * (1.0 - 1e-9) > (double) te && te >= 0.0f;	/* NaN / inf: false */
}

__device__ __forceinline__ bool
s16_sub_excluded(float d, uint32_t rad_bits, float te)
{
	const double lb = (double) d * (1.0 - 1e-3) - (double) __uint_as_float(rad_bits);

	return lb > 0.0 && lb * lb * (1.0 - 1e-9) > (double) te && te >= 0.0f;	/* NaN / inf: false */
}

/* Inner product: the reference's value of a row is -(q.x) and -(q.x) = -(q.c) - q.(x - c) >= -(q.c) - |q| rad for every row
 * of a sublist with centre c and radius rad, where q.c = (|q|^2 + |c|^2 - |q - c|^2) / 2.  alo = a lower bound of
 * |q - c|^2 (the matrix-core sweep's a - e, or the centroid scan's float4 distance squared and shaved), q2 and c2 the
 * norms as stored (fp64 sums rounded to fp32: relative 2^-24 each).  The sublist is left out when that lower bound
 * exceeds thrE = thr + e, the reference value bounding the k-th from above plus the query's error term (which covers
 * the reference's own rounding gamma |q||x|: a row with true value > thrE has a reference value > thr). */
__device__ __forceinline__ bool
s16_sub_excluded_ip(double alo, float q2, float c2, uint32_t rad_bits, float thrE)
{
	const double lbdot = 0.5 * (alo - ((double) q2 + (double) c2) * (1.0 + 3e-7));
	const double qn = __builtin_sqrt((double) q2) * (1.0 + 1e-7);
	const double lb = lbdot - qn * (double) __uint_as_float(rad_bits);

	return lb - __builtin_fabs(lb) * 1e-9 - 1e-30 > (double) thrE;	/* NaN / inf anywhere: false */
}

/* rows (float4, or fp16 as the reference decodes them: H16 1 = with quirk Q20, 2 = no subnormals in the mirror) divided
 * by their norm, as fp32: out = fl32(x_i / sqrt(sum x_i^2)) with the sum and the quotient in fp64.  A zero row stays zero
 * (the reference's cosine distance of a zero vector is exactly 1 = 1 - 0); a row whose norm is outside the range the error
 * model covers becomes NaN (its plane row is marked, every candidate of it is emitted).  One wave per row. */
template <int H16>
__global__ __launch_bounds__(256) void
k_rows_normalise(const void *__restrict__ src, int64_t n, int dim, float *__restrict__ out,
				 int zero_is_nan = 0 /* the centred form (|q^ - x^|^2 = 2 x cosine distance) holds for UNIT vectors only: a zero
									  * vector, whose reference distance is exactly 1, must not pass for one at squared
									  * distance 1 = "cosine distance 0.5" — it is marked like a vector out of range */ )
{
	const int	lane = threadIdx.x & 63;
	const int64_t r = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);

	if (r >= n)
		return;
	auto		at = [&](int i) -> float {
		if constexpr (H16 == 0)
			return ((const float *) src)[(size_t) r * dim + i];
		else if constexpr (H16 == 1)
			return h2f_ref(((const uint16_t *) src)[(size_t) r * dim + i]);
		else
			return __half2float(__ushort_as_half(((const uint16_t *) src)[(size_t) r * dim + i]));
	};
	double		s = 0.0;

	for (int i = lane; i < dim; i += 64)
	{
		const double v = (double) at(i);

		s += v * v;
	}
	s = wave_sum_f64(s);
	/* the error model of s16_e<R_IVF_COS> assumes the reference's fp32 sums neither overflow nor lose their terms to
	 * underflow: |x|^2 within [1e-28, 1e37] (terms below 2^-126 then add up to < 1e-7 of the sum), or exactly zero;
	 * anything else is left to the exact arithmetic (NaN marks the row / the query) */
	const bool	ok = (s == 0.0 && !zero_is_nan) || (s >= 1.0e-28 && s <= 1.0e37);
	const double inv = (ok && s > 0.0) ? 1.0 / __builtin_sqrt(s) : 0.0;

	for (int i = lane; i < dim; i += 64)
		out[(size_t) r * dim + i] = ok ? (float) ((double) at(i) * inv) : __uint_as_float(0x7FC00000u);
}

/* vectors' squared norms (fp64 sums, stored as fp32): one wave each */
__global__ __launch_bounds__(256) void
k_vec_norm2(const float *__restrict__ v, int n, int dim, float *__restrict__ out)
{
	const int	lane = threadIdx.x & 63;
	const int	i = blockIdx.x * 4 + (threadIdx.x >> 6);

	if (i >= n)
		return;
	double		s = 0.0;

	for (int j = lane; j < dim; j += 64)
		s += (double) v[(size_t) i * dim + j] * (double) v[(size_t) i * dim + j];
	s = wave_sum_f64(s);
	if (lane == 0)
		out[i] = s <= 3.0e38 ? (float) s : __uint_as_float(0x7FC00000u);
}

/* (query, probe) -> the sublists of the probed list that stay.  The distance of the query to the sublist's centre is
 * the centroid scan's own value (cdist, the reference's float4 distance, shaved by 1e-3) or pdist[q][p]
 * (k_s16_pair_prune, when the probes came from elsewhere) for a list that is its own single sublist; for the sublists of a
 * regrouped one, subdist[q][gidx] holds the SQUARED distance as the matrix-core sweep computes it (MODE 3, every
 * query against every such centre, within the sweep's own error bound).  FILL = 0: count. */
#define S16_QP_CAP 256			/* (query, sublist) pairs per query the count pass of k_sub_pairs remembers for the fill pass */
template <int FILL>
__global__ __launch_bounds__(256) void
k_sub_pairs(const int *__restrict__ probes, const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t nq,
			const uint32_t *__restrict__ sub_first, const int *__restrict__ sub_gidx, const uint32_t *__restrict__ sub_len,
			const uint32_t *__restrict__ sub_rad, const float *__restrict__ subdist, uint32_t sstride,
			const float2 *__restrict__ qthr, int prune /* 0: nothing is excluded */,
			const float *__restrict__ pdist /* |q - centroid| per (query, probe) ... */,
			const float *__restrict__ cdist /* ... or, when the centroid scan ran here, its [nq][cstride] distances */,
			uint32_t cstride, const float *__restrict__ qn2, const uint32_t *__restrict__ cxmax_bits, int dim,
			const unsigned int *__restrict__ active,
			uint32_t *__restrict__ cnt, const uint32_t *__restrict__ pair_off, uint32_t *__restrict__ fill,
			PairRec *__restrict__ pairs, int ip = 0 /* inner product: s16_sub_excluded_ip, with ... */,
			const float *__restrict__ cn2_sub = nullptr /* ... |c|^2 of the regrouped lists' centres (by gidx) */,
			const float *__restrict__ cn2_list = nullptr /* ... and of the lists' centroids */,
			uint4 *__restrict__ qpairs = nullptr /* [nq][S16_QP_CAP] (sublist, probe, rank in the sublist's run) of the pairs the
												  * count pass kept: the fill pass replays them instead of testing everything
												  * again and needs no atomics ... */,
			uint32_t *__restrict__ qpn = nullptr /* ... (their number per query) ... */,
			uint32_t *__restrict__ qovf = nullptr /* ... unless some query of the batch kept more than S16_QP_CAP (set here; the
												   * fill pass then places every pair with its own counter, as it used to) */,
			int dbg = 0 /* timing experiments, WRONG results: 4 no counter atomics, 8 no reads of subdist, 16 no table reads */,
			uint32_t *__restrict__ cntx = nullptr /* [8][ncs] the count pass's counters, one set per XCD (see below) */,
			uint32_t ncs = 0,
			float ipc_m2 = -1.0f /* >= 0: inner product on the centred sweep — the thresholds are in b's domain (s16c_ip_*), this is M^2 */ )
{
	__shared__ uint32_t s_np;

	if (qpairs && FILL)
	{
		/* (uniform per block) */
		const uint32_t q0 = blockIdx.x;

		if (q0 >= nq || (active && !active[q0]))
			return;
		if (*qovf == 0)
		{
			const uint32_t n0 = qpn[q0];

			for (uint32_t i = threadIdx.x; i < n0; i += blockDim.x)
			{
				const uint4 sp = qpairs[(size_t) q0 * S16_QP_CAP + i];
				PairRec		r;

				r.q = q0;
				r.p = sp.y;
				/* (k_pair_offsets turned cntx[x][s] into the start of XCD x's pairs inside sublist s's run) */
				pairs[pair_off[sp.x] + (cntx ? cntx[(size_t) sp.w * ncs + sp.x] : 0u) + sp.z] = r;
			}
			return;
		}
	}
	if (threadIdx.x == 0)
		s_np = 0;
	__syncthreads();
	uint32_t	xcd = 0;

	if (cntx && !FILL)
	{
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcd));
		xcd &= 7u;
	}
	/* One block (4 waves) per query.  The (probe, sublist) tests of 64 probes at a time are laid end to end and dealt to
	 * the block's 256 lanes (every wave works the same prefix out for itself: lane p learns where probe p's sublists
	 * start and how many there are, a prefix sum over the lanes gives every test its number, and a test finds its probe
	 * by bisection over the 64 starts in LDS).  The kernel is a chain of dependent loads per test (sublist -> centre
	 * index -> distance) and every wave of a 4096-query batch is resident at once, so its time is the LONGEST WAVE's:
	 * one wave per query walking its probes one after the other took 89 + 77 us (count + fill), 64 lanes over the
	 * flattened tests 71 + 66 us, 256 lanes what is measured now. */
	__shared__ uint32_t s_off[4][65], s_s0[4][64];
	__shared__ float s_pd[4][64], s_c2[4][64];
	const int	lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t q = blockIdx.x;

	if (q >= nq || (active && !active[q]))
		return;
	const uint32_t *co = loc_cand_off + (size_t) q * (npr + 1);
	float		te = qthr[q].x;

	if (ipc_m2 >= 0.0f)
	{
		/* T = |q|^2 + M^2 + 2 (thr + ev) back to the reference's domain, upwards: the value a row's real -q.x must exceed
		 * for the row to be out (what s16_sub_excluded_ip compares with) */
		const double tv = 0.5 * ((double) te - (double) qn2[q] * (1.0 - 3e-7) - (double) ipc_m2 * (1.0 - 3e-7));

		te = (float) (tv + __builtin_fabs(tv) * 1e-6 + 1e-30);
	}
	/* error of the centre distances: the sweep's own bound with the largest centre norm in the rows' place */
	const float ec = prune ? s16_e<R_IVF_L2>(dim, qn2[q], __uint_as_float(*cxmax_bits), false) : 0.0f;

	for (int p0 = 0; p0 < npr; p0 += 64)
	{
		const int	p = p0 + lane;
		uint32_t	n = 0, s0 = 0;
		float		pd = 0.0f, c2l = 0.0f;

		if (p < npr && co[p + 1] != co[p])
		{
			const int	L = probes[(size_t) q * npr + p];

			s0 = sub_first[L];
			n = sub_first[L + 1] - s0;
			pd = (!prune || ip >= 2) ? 0.0f : (cdist ? cdist[(size_t) q * cstride + L] : pdist[(size_t) q * npr + p]);
			if (ip == 1 && prune)
				c2l = cn2_list[L];
		}
		/* inclusive prefix sum of n over the lanes */
		uint32_t	inc = n;

#pragma unroll
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t v = (uint32_t) __shfl_up((int) inc, off, 64);

			if (lane >= off)
				inc += v;
		}
		s_off[w][lane] = inc - n;
		s_s0[w][lane] = s0;
		s_pd[w][lane] = pd;
		s_c2[w][lane] = c2l;
		const uint32_t T = (uint32_t) __shfl((int) inc, 63, 64);

		if (lane == 0)
			s_off[w][64] = T;
		__builtin_amdgcn_wave_barrier();
		for (uint32_t t = (uint32_t) threadIdx.x; t < T; t += 256)
		{
			/* the probe whose tests include number t: the last lane pl with s_off[pl] <= t (empty probes share their
			 * successor's start and are skipped by "last") */
			int			lo = 0, hi = 64;

			while (hi - lo > 1)
			{
				const int	mid = (lo + hi) >> 1;

				if (s_off[w][mid] <= t)
					lo = mid;
				else
					hi = mid;
			}
			const uint32_t s = s_s0[w][lo] + (t - s_off[w][lo]);
			/* (the three table reads go out together: behind the `continue` they would be three round trips) */
			const uint32_t slen = (dbg & 16) ? 100u : sub_len[s];
			const int	gi = (dbg & 16) ? (int) (s & 1023u) : sub_gidx[s];
			const uint32_t srad = (dbg & 16) ? 0u : sub_rad[s];

			if (slen == 0)
				continue;

			if (prune && ip >= 2 && gi < 0)
				;				/* cosine: the centroid scan's distance to a list that is its own sublist is in the rows' own space: kept */
			else if (prune && ip == 3)
			{
				/* cosine on the centred sweep: the L2 test in the normalised space */
				if (s16_sub_excluded_a(subdist[(size_t) q * sstride + gi], ec, srad, te))
					continue;
			}
			else if (prune && ip)
			{
				const double pdl = (double) s_pd[w][lo];
				/* (the centroid scan's float4 distance is within (dim + 3) 2^-24 of the real one, relatively; squared: twice that) */
				const double alo = gi < 0 ? pdl * pdl * (1.0 - 2.2 * (double) (dim + 8) * 5.9604645e-8)
					: (double) subdist[(size_t) q * sstride + gi] - (double) ec * (1.0 + 1e-6);

				if (s16_sub_excluded_ip(alo, qn2[q], gi < 0 ? s_c2[w][lo] : cn2_sub[gi], srad, te))
					continue;
			}
			else if (prune && (gi < 0 ? s16_sub_excluded(s_pd[w][lo], srad, te)
							   : s16_sub_excluded_a((dbg & 8) ? (float) (s & 255u) * te * 0.02f : subdist[(size_t) q * sstride + gi], ec, srad, te)))
				continue;
			if (FILL)
			{
				PairRec		r;

				r.q = q;
				r.p = (uint32_t) (p0 + lo);
				pairs[pair_off[s] + atomicAdd(&fill[s], 1u)] = r;
			}
			else
			{
				/*
				 * One counter per (XCD, sublist), bumped where this block runs: 50 k device-scope atomics with a return
				 * value — they execute at the memory side, for every XCD — were half of this kernel (64 -> 32 us without
				 * them); an atomic that only has to be coherent among the blocks of ONE XCD executes in that XCD's L2.
				 * The scope is `workgroup` to get exactly that instruction (no sc1); what makes it correct is that every
				 * block that touches cntx[x][.] runs on XCD x (it asked the hardware which one it is on), and that the
				 * next kernel reads the counters after this one's end-of-kernel write-back.
				 */
				const uint32_t rank = (dbg & 4) ? 0u
					: (cntx ? __hip_atomic_fetch_add(&cntx[(size_t) xcd * ncs + s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
					   : atomicAdd(&cnt[s], 1u));

				if (qpairs)
				{
					const uint32_t i = atomicAdd(&s_np, 1u);

					if (i < S16_QP_CAP)
						qpairs[(size_t) q * S16_QP_CAP + i] = make_uint4(s, (uint32_t) (p0 + lo), rank, xcd);
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	if (qpairs && !FILL)
	{
		__syncthreads();
		if (threadIdx.x == 0)
		{
			qpn[q] = s_np;
			if (s_np > S16_QP_CAP)
				*qovf = 1u;
		}
	}
}

/* One wave per query: the same split for the batch's queries; qn2 / qexp like rn2 / rexp.  A query whose norm is
 * not finite gets qn2 = NaN: every candidate of it is emitted and the batch falls back to the older path. */
__global__ __launch_bounds__(256) void
k_s16_qprep(const float *__restrict__ queries, uint32_t nq, int dim, int dimp, ndb_h2 *__restrict__ qplanes,
			float *__restrict__ qn2, int *__restrict__ qexp)
{
	const int	lane = threadIdx.x & 63;
	const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);

	if (q >= nq)
		return;
	const float *x = queries + (size_t) q * dim;
	double		s = 0.0;

	for (int i = lane; i < dim; i += 64)
		s += (double) x[i] * (double) x[i];
	s = wave_sum_f64(s);
	const bool	ok = s <= 3.0e38;
	const int	e = ok ? s16_exponent(s) : 0;

	if (lane == 0)
	{
		qn2[q] = ok ? (float) s : __uint_as_float(0x7FC00000u);
		qexp[q] = e;
	}
	ndb_h2	   *out = qplanes + (size_t) q * dimp;

	for (int p = lane; p < dimp / 2; p += 64)
	{
		const int	i = 2 * p, c = i >> 5, j = (i & 31) >> 1;
		_Float16	h0 = 0, l0 = 0, h1 = 0, l1 = 0;

		if (ok && i < dim)
			s16_split(x[i], e, h0, l0);
		if (ok && i + 1 < dim)
			s16_split(x[i + 1], e, h1, l1);
		ndb_h2		h, l;

		h.x = h0; h.y = h1; l.x = l0; l.y = l1;
		out[c * 32 + j] = h;
		out[c * 32 + 16 + j] = l;
	}
}

/* x rounded towards +inf by more than any rounding of the expression that produced it can have lost */
__device__ __forceinline__ float
s16_up(float x)
{
	return x + fabsf(x) * 4.8e-7f + 1e-37f;
}

/* E of a query (ndbhip_common.h (6), (7)); sub = the mirror holds fp16 subnormals that the reference decodes
 * 2^-10 too small (quirk Q20) while the matrix cores take them at face value: |q.x - q.x'| <= 2^-14 sqrt(dim) |q| */
template <int R>
__device__ __forceinline__ float
s16_e(int dim, float q2, float x2max, bool sub)
{
	float		e;

	if (R == R_IVF_COS)
	{
		/* COSINE runs as the inner product of NORMALISED rows and queries (k_rows_normalise: q^ = fl32(q / |q|), norms
		 * within 1 +- 4u): a = -(q^.x^) as the sweep computes it is within cdot |q^||x^| of the fp32 vectors' product,
		 * that within 6u of the real cosine.  The reference's 1 - dot / (sqrt(n1) sqrt(n2)): the dot product's sequential
		 * fp32 sum is within gamma_dim |q||x| of the real one; n1 and n2 (sums of non-negative terms) within gamma_dim
		 * relatively, their roots within gamma_dim / 2 + u each, the product of the roots within gamma_dim + 3u; so the
		 * quotient is within (gamma_dim |q||x| + |dot| (gamma_dim + 3u)) / (|q||x|) (1 + ..) + u <= 2 gamma_dim + 5u of the
		 * real cosine (|dot| <= |q||x|), and the difference adds 2u: 2.01 gamma_(dim + 8) + 16u covers it with the second-
		 * order terms.  |(1 + a) - reference| <= e.  The norms and the subnormal term play no part: the planes hold
		 * decoded, normalised values. */
		e = ndb_s16_cdot(dim) * 1.00001f + 2.01f * ndb_s16_gamma(dim + 8) + 16.0f * NDB_S16_U;
		return s16_up(e * 1.00001f) + NDB_S16_ABS;
	}
	if (R == R_IVF_L2)
		e = (ndb_s16_cdot(dim) + NDB_S16_NORMS) * (q2 + x2max);
	else
		e = (ndb_s16_cdot(dim) + ndb_s16_gamma(dim)) * __builtin_sqrtf(q2) * __builtin_sqrtf(x2max) * 1.000001f;
	if (sub)
		e += (R == R_IVF_L2 ? 2.0f : 1.0f) * 6.1035156e-5f * __builtin_sqrtf((float) dim) * __builtin_sqrtf(q2) * 1.000001f;
	return s16_up(e * 1.00001f) + NDB_S16_ABS;
}

/* "emit unless a > thrE" for a float4 reference value thr that bounds the k-th distance from above */
template <int R>
__device__ __forceinline__ float
s16_thr_from_ref(float thr, float e, int dim)
{
	if (R == R_IVF_L2)
		return s16_up(s16_up(thr * thr) * (1.0f + ndb_s16_refslack(dim)) + e);
	if (R == R_IVF_COS)
		return s16_up(s16_up(thr - 1.0f) + e);	/* the sweep's a = -(q^.x^) = (cosine distance) - 1 */
	return s16_up(thr + e);
}

/* the same from the k-th smallest emitted a: the k-th reference value is at most (a_k + E)(1 + m) resp. a_k + E */
template <int R>
__device__ __forceinline__ float
s16_thr_from_a(float ak, float e, int dim)
{
	if (R == R_IVF_L2)
	{
		const float t = s16_up(fmaxf(ak + e, 0.0f)) * (1.0f + 2.5f * ndb_s16_refslack(dim));

		return s16_up(s16_up(t) + e);
	}
	return s16_up(s16_up(ak + e) + e);
}

/*
 * COSINE ON THE CENTRED SWEEP (ndbhip_screen16c.h): |q^ - x^|^2 = 2 (1 - q^.x^) = 2 x (cosine distance) for the normalised
 * vectors, so the centred one-plane L2 sweep over the NORMALISED planes bounds the cosine distance; its thresholds T live
 * in that squared-L2 domain.  e_ref = what separates the reference's 1 - dot / (sqrt(n1) sqrt(n2)) from the real cosine
 * distance (s16_e<R_IVF_COS>: 2.01 gamma_(dim + 8) + 16u) plus the normalisation's roundings (|q^_f|, |x^_f| within 4u of 1:
 * 32u on the squared distance).
 *   from a reference value thr that bounds the k-th from above: a row can be among the k only if its reference value is
 *     <= thr, i.e. its real cosine distance <= thr + e_ref, i.e. |q^ - x^|^2 <= 2 (thr + e_ref):  T = 2 (thr + e_ref);
 *   from the k-th smallest upper bound U of |q^ - x^|^2 over distinct candidates: those k rows have reference values
 *     <= U / 2 + e_ref, which is such a thr:  T = U + 4 e_ref.
 */
__device__ __forceinline__ float
s16c_cos_eref(int dim)
{
	return (2.01f * ndb_s16_gamma(dim + 8) + 48.0f * NDB_S16_U) * 1.0001f;
}
__device__ __forceinline__ float
s16c_cos_t_from_ref(float thr, int dim)
{
	return s16_up(s16_up(2.0f * (fmaxf(thr, 0.0f) + s16c_cos_eref(dim))) * 1.000001f);
}
__device__ __forceinline__ float
s16c_cos_t_from_ub(float ub, int dim)
{
	return s16_up(s16_up(fmaxf(ub, 0.0f)) * 1.000001f + 4.0f * s16c_cos_eref(dim));
}

/*
 * INNER PRODUCT ON THE CENTRED SWEEP (ndbhip_screen16c.h; quantization.c:2075-2116 / hnsw_am.c:1334-1337 are where the
 * reference's -sum q_i x_i comes from).  With M^2 = the largest |x|^2 of the mirror,
 *     b(q, x) = |q - x|^2 + (M^2 - |x|^2) = |q|^2 + M^2 - 2 q.x  >= 0
 * orders the rows of a query like -q.x does, and its first term is what the centred L2 sweep bounds (a - E <= |q - x|^2
 * <= a + E) from the SAME planes an L2 search uses; the second is a constant of the row (`rnx`, per padded plane row).  The
 * thresholds T of such a batch live in b's domain.  ev = gamma_dim |q| M (+) bounds what separates the reference's
 * sequential fp32 sum from the real -q.x for every row (|x| <= M).
 *   from a reference value thr that bounds the k-th from above: a row can be among the k only if its reference value is
 *     <= thr, i.e. -q.x <= thr + ev, i.e. b <= |q|^2 + M^2 + 2 (thr + ev):  T = that (|q|^2, M^2 as stored: fp64 sums
 *     rounded to fp32, taken 2^-21 up);
 *   from the k-th smallest upper bound U of b over distinct candidates: those k rows have -q.x <= (U - |q|^2 - M^2) / 2,
 *     hence reference values <= that + ev, which is such a thr:  T = U + 4 ev  (|q|^2 and M^2 cancel).
 * T >= 0 always (b is), so its bits order like its values, as the sweep's atomicMin wants.
 */
__device__ __forceinline__ float
s16c_ip_ev(int dim, float q2, float m2)
{
	return s16_up(ndb_s16_gamma(dim) * __builtin_sqrtf(q2) * __builtin_sqrtf(m2) * 1.00001f) + NDB_S16_ABS;
}
__device__ __forceinline__ float
s16c_ip_t_from_ref(float thr, float q2, float m2, int dim)
{
	const float t = s16_up(s16_up(q2 * 1.0000005f) + s16_up(m2 * 1.0000005f) + 2.0f * s16_up(thr + s16c_ip_ev(dim, q2, m2)));

	return t == t ? fmaxf(t, 0.0f) : __uint_as_float(0x7F800000u);
}
__device__ __forceinline__ float
s16c_ip_t_from_ub(float ub, float ev)
{
	return s16_up(s16_up(fmaxf(ub, 0.0f)) + 4.0f * ev);
}

typedef __attribute__((address_space(3))) void *ndb_lds_ptr;

/* -DNDB_PHASES (profiling builds only): block 0's first lane stamps the 100 MHz clock at marked places of the per-batch
 * kernels; ndbhip_debug_phases() reads the stamps (tools/phase_probe.py) */
#ifdef NDB_PHASES
__device__ unsigned long long g_phases[64];
#define NDB_PHASE(I) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phases[I] = wall_clock64(); } while (0)
#else
#define NDB_PHASE(I) ((void) 0)
#endif

template <int N> __device__ __forceinline__ void
s16_wait_vm()
{
	static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
	asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

/* wait until all but the newest `ahead` chunks' requests (IPC instructions each) have landed; ahead * IPC <= 63 */
template <int IPC, int K> __device__ __forceinline__ void
s16_wait_vm_k()
{
	if constexpr (K * IPC < 64)
		s16_wait_vm<K * IPC>();
	else
		s16_wait_vm<0>();
}
template <int IPC> __device__ __forceinline__ void
s16_wait_vm_chunks(int ahead)
{
	switch (ahead)
	{
		case 1: s16_wait_vm_k<IPC, 1>(); break;
		case 2: s16_wait_vm_k<IPC, 2>(); break;
		case 3: s16_wait_vm_k<IPC, 3>(); break;
		case 4: s16_wait_vm_k<IPC, 4>(); break;
		case 5: s16_wait_vm_k<IPC, 5>(); break;
		case 6: s16_wait_vm_k<IPC, 6>(); break;
		case 7: s16_wait_vm_k<IPC, 7>(); break;
		case 8: s16_wait_vm_k<IPC, 8>(); break;
		case 9: s16_wait_vm_k<IPC, 9>(); break;
		case 10: s16_wait_vm_k<IPC, 10>(); break;
		case 11: s16_wait_vm_k<IPC, 11>(); break;
		case 12: s16_wait_vm_k<IPC, 12>(); break;
		default: s16_wait_vm<0>(); break;
	}
}

/* 16 bytes per lane from 64 unrelated addresses to la + 16 lane */
__device__ __forceinline__ void
s16_dma16_at(const unsigned char *p, uint32_t la)
{
	asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
				 :: "s"(la), "v"(p) : "memory");
}

/*
 * The reference's sequential arithmetic for up to 16 NG rows per wave, one lane per row, with the rows STREAMED through
 * LDS instead of loaded by the lane that sums them.  A lane that reads its own row asks for 16 bytes of 64 different
 * lines per instruction and then waits for them before its next 4 (fp16: 8) steps: a batch of 256 queries spent 0.12 ms
 * in k_s16_finalize and 0.095 ms in k_cent_select on 1536 dimensions — a chain of ~200 memory round trips — and a
 * batch of 4096 is bound by the 16-byte requests themselves.  Here the wave copies a CHUNK (256 bytes of every row: 64
 * floats or 128 halves) per step by LDS DMA, 4 lanes per row and instruction (64 contiguous bytes of 16 rows), nbuf - 1
 * chunks ahead of the one being summed, and the query's chunk with it; the lanes then read their row from LDS
 * (16-byte units, swizzled by the row so that the 16 rows of a group fall into different banks).
 *   rowp  this lane's row (lanes without one: any readable row, e.g. the wave's first; their result means nothing)
 *   qq    the query (wave-uniform), dim floats
 *   row bytes must be a multiple of 16 (dim % 4 == 0; fp16: dim % 8 == 0) and rows 16-byte aligned: s16_staged_ok
 *   ring  the wave's LDS ring, nbuf * S16X_SLOT(NG) bytes, 16-byte aligned; nbuf >= 2, (nbuf - 1) * (4 NG + 1) <= 63
 * Every lane of the wave calls it (uniform control flow).  The sum is Acc<R>'s: the same steps in the same order as
 * scr_exact / scr_exact_h, bit for bit.
 */
#define S16X_SLOT(NG) ((4 * (NG) + 1) * 1024)
__host__ __device__ __forceinline__ bool
s16_staged_ok(int dim, int h16)
{
	return h16 ? (dim % 8) == 0 : (dim % 4) == 0;
}
__host__ __device__ __forceinline__ int
s16_staged_nbuf(int ng, int want)
{
	const int	most = 63 / (4 * ng + 1) + 1;

	return want < 2 ? 2 : (want > most ? most : want);
}

template <int R, int H16, int NG>
__device__ __forceinline__ float
s16_exact_staged(const float *__restrict__ qq, const unsigned char *rowp, int dim, unsigned char *ring, int nbuf)
{
	constexpr int IPC = 4 * NG + 1;				/* DMA instructions per chunk */
	constexpr int CD = H16 ? 128 : 64;			/* dimensions per chunk */
	constexpr int UD = H16 ? 8 : 4;				/* ... per 16-byte unit of a row */
	const int	lane = threadIdx.x & 63;
	const uint32_t ring_la = (uint32_t) (uintptr_t) (ndb_lds_ptr) ring;
	const uint32_t rowbytes = (uint32_t) dim * (H16 ? 2u : 4u);
	const int	nchunk = (dim + CD - 1) / CD;
	/* the rows this lane copies from: row 16 g + lane / 4 for g = 0 .. NG - 1, unit (lane & 3) ^ swizzle of every piece */
	const unsigned char *src[NG];
	const uint32_t cu16 = (uint32_t) ((lane & 3) ^ ((lane >> 4) & 3)) * 16u;

#pragma unroll
	for (int g = 0; g < NG; g++)
	{
		const uint64_t v = (uint64_t) rowp;
		const int	from = 16 * g + (lane >> 2);
		const uint32_t lo = (uint32_t) __shfl((int) (uint32_t) v, from, 64), hi = (uint32_t) __shfl((int) (uint32_t) (v >> 32), from, 64);

		src[g] = (const unsigned char *) (((uint64_t) hi << 32) | lo);
	}
	const unsigned char *qb = (const unsigned char *) qq;
	const uint32_t qbytes = (uint32_t) dim * 4u;

	int			si = 0;					/* ring slot of the next chunk to request */
	auto		issue = [&](int c) {
		const uint32_t la = ring_la + (uint32_t) si * (uint32_t) S16X_SLOT(NG);

		si = si + 1 == nbuf ? 0 : si + 1;

#pragma unroll
		for (int g = 0; g < NG; g++)
#pragma unroll
			for (int p = 0; p < 4; p++)
			{
				uint32_t	off = (uint32_t) c * 256u + (uint32_t) p * 64u + cu16;

				off = off < rowbytes ? off : 0u;		/* beyond the row: anything readable, never looked at */
				s16_dma16_at(src[g] + off, la + (uint32_t) (g * 4 + p) * 1024u);
			}
		{
			uint32_t	off = (uint32_t) c * (uint32_t) (CD * 4) + (uint32_t) lane * 16u;

			off = ((uint32_t) lane * 16u < (uint32_t) (CD * 4) && off < qbytes) ? off : 0u;
			s16_dma16_at(qb + off, la + (uint32_t) NG * 4096u);
		}
	};

	for (int c = 0; c < nbuf - 1 && c < nchunk; c++)
		issue(c);
	Acc<R>		acc;
	const int	g_me = lane >> 4, rr = lane & 15, sw = (rr >> 2) & 3;
	int			sc = 0;					/* ring slot of chunk c */

	for (int c = 0; c < nchunk; c++)
	{
		if (c + nbuf - 1 < nchunk)
		{
			issue(c + nbuf - 1);			/* into the slot of chunk c - 1: its reads were consumed by the sums above */
			s16_wait_vm_chunks<IPC>(nbuf - 1);
		}
		else
			s16_wait_vm<0>();
		const unsigned char *slot = ring + (size_t) sc * S16X_SLOT(NG);

		sc = sc + 1 == nbuf ? 0 : sc + 1;
		const unsigned char *mine = slot + (NG > 1 ? g_me * 4096 : 0) + rr * 64;
		const unsigned char *qs = slot + NG * 4096;
		const int	d0 = c * CD;

		/* (a whole chunk inside the row: no test between the units, so that their LDS reads are issued together) */
		auto		sum_chunk = [&](auto whole) {
#pragma unroll
			for (int p = 0; p < 4; p++)
#pragma unroll
				for (int u = 0; u < 4; u++)
				{
					const int	d = d0 + (p * 4 + u) * UD;

					if (!decltype(whole)::value && d >= dim)		/* uniform */
						continue;
					const float4 raw = *reinterpret_cast<const float4 *>(mine + p * 1024 + ((u ^ sw) * 16));

					if constexpr (H16 != 0)
					{
						const float4 q0 = *reinterpret_cast<const float4 *>(qs + (p * 4 + u) * 32);
						const float4 q1 = *reinterpret_cast<const float4 *>(qs + (p * 4 + u) * 32 + 16);
						float		v[8];

						decode8<H16 == 1>(raw, v);
						acc.step(q0.x, v[0]);
						acc.step(q0.y, v[1]);
						acc.step(q0.z, v[2]);
						acc.step(q0.w, v[3]);
						acc.step(q1.x, v[4]);
						acc.step(q1.y, v[5]);
						acc.step(q1.z, v[6]);
						acc.step(q1.w, v[7]);
					}
					else
					{
						const float4 q0 = *reinterpret_cast<const float4 *>(qs + (p * 4 + u) * 16);

						acc.step(q0.x, raw.x);
						acc.step(q0.y, raw.y);
						acc.step(q0.z, raw.z);
						acc.step(q0.w, raw.w);
					}
				}
		};

		if (d0 + CD <= dim)
			sum_chunk(std::true_type{});
		else
			sum_chunk(std::false_type{});
		/* (the compiler must not start the next chunk's reads before the wait, nor hold this chunk's past the refill) */
		asm volatile("" ::: "memory");
	}
	return acc.fin();
}

/*
 * ivfSelectClusters (src/index/ivf_am.c:1597-1717) for a batch, screened: amat[q][c] = a ~ |q - centroid c|^2 from
 * the two-plane sweep's MODE 3 (s16mat_run over the centroids' planes; |a - D| <= E = s16_e(dim, |q|^2, largest
 * centroid norm)).  The reference's float4 distance d of a centroid satisfies (a - E)(1 - m) <= d^2 <= (a + E)(1 + m)
 * (ndbhip_common.h (7)); with U = the nprobe-th smallest upper bound, at least nprobe centroids have d^2 <= U, so a
 * centroid whose lower bound exceeds U is neither among the nprobe nearest nor tied with one of them.  The others —
 * nprobe plus a few — get the reference's own sequential sqrtf(sum (q - c)^2) (one lane each) and the selection
 * ("nprobe times the first strict minimum" = ascending (distance, index), valid = below FLT_MAX) runs over those.
 * cdist[q][c] receives the exact distance of every centroid examined (the probed ones among them: what the sublist
 * code reads).  A query with more than NDB_CSEL_CAP candidates (or fewer than nprobe finite bounds) is marked in
 * `full` after ALL its distances were computed exactly: k_probe_select serves it from cdist as before.
 * One wave per query; PER = centroids per lane.
 */
#define NDB_CSEL_CAP 256
#define S16_OVER_CAP 512		/* queries of a batch that may go to the exact path on their own (more: the whole batch does) */

template <int PER, bool STAGED = false /* the candidates' centroids come through LDS (dynamic: stage_nbuf slots, NG = 4) */>
__global__ __launch_bounds__(64) void
k_cent_select(const float *__restrict__ amat, uint32_t astride, const float *__restrict__ qn2, const uint32_t *__restrict__ cmax_bits,
			  const float *__restrict__ queries, const float *__restrict__ cents, int dim, int ncmp, int ncent, int npr,
			  const uint32_t *__restrict__ glob_len, const uint32_t *__restrict__ own_lo, const uint32_t *__restrict__ own_len,
			  uint64_t cap, float *__restrict__ cdist, uint32_t cstride, int *__restrict__ probes,
			  uint32_t *__restrict__ cand_off, uint32_t *__restrict__ loc_cand_off, uint8_t *__restrict__ full,
			  int stage_nbuf = 0 /* STAGED: slots of the ring */ )
{
	extern __shared__ __attribute__((aligned(16))) unsigned char csel_ring[];
	__shared__ uint32_t s_idx[NDB_CSEL_CAP], s_key[NDB_CSEL_CAP];
	__shared__ int s_sel[NDBHIP_MAX_NPROBE];
	__shared__ uint32_t s_len[NDBHIP_MAX_NPROBE];
	const uint32_t q = blockIdx.x;
	const int	lane = threadIdx.x;
	const float *av = amat + (size_t) q * astride;
	const float *qq = queries + (size_t) q * dim;
	const int	npr_eff = max(0, min(npr, ncmp));
	const float e = s16_e<R_IVF_L2>(dim, qn2[q], __uint_as_float(*cmax_bits), false);
	const float m = ndb_s16_refslack(dim);
	uint32_t	ub[PER], lb[PER];

	NDB_PHASE(0);

#pragma unroll
	for (int j = 0; j < PER; j++)
	{
		const int	c = j * 64 + lane;
		const float a = c < ncmp ? av[c] : 0.0f;
		const float hi = s16_up(s16_up(fmaxf(a + e, 0.0f)) * (1.0f + m));
		const float lo0 = fmaxf(a - e, 0.0f) * (1.0f - m);
		const float lo = fmaxf(lo0 - lo0 * 4.8e-7f - 1e-37f, 0.0f);
		/* a NaN or an infinity anywhere (fmaxf drops a NaN operand: test the inputs): no bound at all — the centroid
		 * is always examined and never counts towards the nprobe bounds U rests on */
		const bool	fin = (__float_as_uint(a) & 0x7FFFFFFFu) < 0x7F800000u && (__float_as_uint(e) & 0x7FFFFFFFu) < 0x7F800000u &&
			hi < __uint_as_float(0x7F800000u);

		ub[j] = c < ncmp ? (fin ? __float_as_uint(hi) : 0x7F800000u) : 0xFFFFFFFFu;
		lb[j] = c < ncmp ? (fin ? __float_as_uint(lo) : 0u) : 0xFFFFFFFFu;
	}
	/* U = the npr_eff-th smallest upper bound (non-negative floats: the bits order like the values): the largest v
	 * with #(ub < v) < npr_eff */
	uint32_t	U = 0;

	NDB_PHASE(1);

	if (npr_eff > 0)
	{
		for (int bit = 30; bit >= 0; bit--)
		{
			const uint32_t t = U | (1u << bit);
			uint32_t	mine = 0, n = 0;

#pragma unroll
			for (int j = 0; j < PER; j++)
				mine += ub[j] < t ? 1u : 0u;
			/* the wave's total, bit by bit on the scalar side (mine <= PER <= 64: 7 independent ballots instead of six
			 * dependent cross-lane steps) */
#pragma unroll
			for (int b = 0; b < 7; b++)
				n += (uint32_t) __popcll(__ballot((mine >> b) & 1u)) << b;
			if (n < (uint32_t) npr_eff)
				U = t;
		}
	}
	/* the candidates, in index order */
	uint32_t	total = 0;

	NDB_PHASE(2);

#pragma unroll
	for (int j = 0; j < PER; j++)
	{
		const bool	in = lb[j] <= U && j * 64 + lane < ncmp;
		const unsigned long long mk = __ballot(in);

		if (in)
		{
			const uint32_t slot = total + (uint32_t) __popcll(mk & ((1ull << lane) - 1ull));

			if (slot < NDB_CSEL_CAP)
				s_idx[slot] = (uint32_t) (j * 64 + lane);
		}
		total += (uint32_t) __popcll(mk);
	}
	if (total > NDB_CSEL_CAP || U >= 0x7F800000u || npr_eff == 0)
	{
		/* nothing to gain here: every distance exactly, the old selection kernel does the rest */
		for (int c = lane; c < ncmp; c += 64)
			cdist[(size_t) q * cstride + c] = scr_exact<R_IVF_L2>(qq, cents + (size_t) c * dim, dim);
		if (lane == 0)
			full[q] = 1;
		return;
	}
	__syncthreads();
	NDB_PHASE(3);
	for (uint32_t i0 = 0; i0 < total; i0 += 64)		/* uniform */
	{
		const uint32_t i = i0 + lane;
		const uint32_t c = s_idx[i < total ? i : 0];
		float		v = 0.0f;

		if constexpr (STAGED)
			v = s16_exact_staged<R_IVF_L2, 0, 4>(qq, (const unsigned char *) (cents + (size_t) c * dim), dim, csel_ring, stage_nbuf);
		else if (i < total)
			v = scr_exact<R_IVF_L2>(qq, cents + (size_t) c * dim, dim);

		if (i >= total)
			continue;
		cdist[(size_t) q * cstride + c] = v;
		/* valid = strictly below FLT_MAX (bestDist starts at FLT_MAX: ivf_am.c:1689, 1706); the others never win */
		s_key[i] = v < FLT_MAX ? ndb_key_from_bits(__float_as_uint(v)) : 0xFFFFFFFFu;
	}
	__syncthreads();
	/* rank of every valid candidate by (distance, index); the list is in index order, so ties go to the earlier slot */
	uint32_t	nvalid = 0;

	NDB_PHASE(4);

	for (uint32_t i = lane; i < total; i += 64)
		nvalid += s_key[i] != 0xFFFFFFFFu ? 1u : 0u;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		nvalid += (uint32_t) __shfl_xor((int) nvalid, off, 64);
	const uint32_t kk = min((uint32_t) npr_eff, nvalid);

	for (uint32_t i = lane; i < total; i += 64)
	{
		const uint32_t ki = s_key[i];

		if (ki == 0xFFFFFFFFu)
			continue;
		uint32_t	rank = 0;

		for (uint32_t o = 0; o < total; o++)
		{
			const uint32_t ko = s_key[o];

			rank += (ko < ki || (ko == ki && o < i)) ? 1u : 0u;
		}
		if (rank < kk)
			s_sel[rank] = (int) s_idx[i];
	}
	__syncthreads();
	NDB_PHASE(5);
	const bool	in_lds = npr <= NDB_CSEL_CAP;

	for (int i = lane; i < npr; i += 64)
	{
		int			c;

		if ((uint32_t) i < kk)
			c = s_sel[i];
		else if (i < npr_eff)
			c = -1;
		else
			c = 0;
		probes[(size_t) q * npr + i] = c;
		s_sel[i] = c;
		const bool	held = c >= 0 && c < ncent;

		s_len[i] = held ? glob_len[c] : 0u;	/* ivf_am.c:1768-1779 */
		/* the probes' own-row ranges, read by all lanes into the (now free) candidate arrays: lane 0's running sums
		 * below then touch LDS only — two dependent global loads per probe were 64 round trips in a row */
		if (in_lds)
		{
			s_idx[i] = held ? own_lo[c] : 0u;
			s_key[i] = held ? own_len[c] : 0u;
		}
	}
	__syncthreads();
	NDB_PHASE(6);
	if (lane == 0)
	{
		uint64_t	acc = 0, mine = 0;
		uint32_t   *co = cand_off + (size_t) q * (npr + 1);
		uint32_t   *lco = loc_cand_off ? loc_cand_off + (size_t) q * (npr + 1) : nullptr;

		full[q] = 0;
		co[0] = 0;
		if (lco)
			lco[0] = 0;
		for (int i = 0; i < npr; i++)
		{
			uint64_t	l = s_len[i];

			if (cap > 0 && acc + l > cap)
				l = cap - acc;	/* candidateCount < maxCandidates guards: ivf_am.c:1764, 1793, 1811 */
			acc += l;
			if (l > 0 && in_lds)
			{
				const uint64_t lo = s_idx[i], hi = lo + s_key[i];

				mine += l > lo ? ((l < hi ? l : hi) - lo) : 0;
			}
			else if (l > 0)
				mine += ndb_local_part(l, own_lo, own_len, s_sel[i]);
			co[i + 1] = (uint32_t) acc;
			if (lco)
				lco[i + 1] = (uint32_t) mine;
		}
	}
	NDB_PHASE(7);
}

/*
 * First threshold: one wave per query scores its first S16_SEED candidates (probe order: the nearest list
 * first) with the reference's arithmetic; the k-th smallest of those distances bounds the query's k-th
 * distance.  Fewer than k candidates: +inf (everything is emitted).  Also E_q.
 * qinfo[q] = {thrE, E}.
 */
template <int R, int H16>
__global__ __launch_bounds__(64) void
k_s16_seed(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
		   const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t k, const float *__restrict__ qn2,
		   const uint32_t *__restrict__ xmax_bits, int sub, float2 *__restrict__ qthr,
		   int cen = 0 /* the centred sweep's threshold: thr^2 (1 + m) without an error term (ndbhip_screen16c.h) */ )
{
	const uint32_t q = blockIdx.x;
	const uint32_t lane = threadIdx.x;
	const uint32_t *lco = loc_cand_off + (size_t) q * (npr + 1);
	const uint32_t all = lco[npr];
	const uint32_t n = min(all, (uint32_t) S16_SEED);
	const int	dim = ix.dim;
	float		v = 0.0f;

	if (lane < n)
	{
		const uint32_t p = find_probe(lco, npr, lane);
		const int	L = probes[(size_t) q * npr + p];
		const size_t row = (size_t) ix.loc_off[L] + (lane - lco[p]);

		if constexpr (H16 != 0)
			v = scr_exact_h<R, H16 == 1>(queries + (size_t) q * dim, (const uint16_t *) ix.vecs + row * (size_t) dim, dim);
		else
			v = scr_exact<R>(queries + (size_t) q * dim, ix.vecs + row * (size_t) dim, dim);
	}
	/* rank of every lane's value among the n (ties by lane); NaN keys sort last */
	const uint32_t key = lane < n ? ndb_key_from_bits(__float_as_uint(v)) : 0xFFFFFFFFu;
	uint32_t	rank = 0;

	for (int j = 0; j < S16_SEED; j++)
	{
		const uint32_t kj = (uint32_t) __shfl((int) key, j, 64);

		rank += (kj < key || (kj == key && (uint32_t) j < lane)) ? 1u : 0u;
	}
	const float e = s16_e<R>(dim, qn2[q], __uint_as_float(*xmax_bits), sub != 0);
	const unsigned long long pick = __ballot(lane < n && rank == k - 1);
	float		thrE = __uint_as_float(0x7F800000u);	/* +inf */

	if (n >= k && pick)
	{
		const float thr = __shfl(v, __ffsll((long long) pick) - 1, 64);

		if (thr == thr)			/* a NaN distance bounds nothing */
			/* (cen == 2, inner product on the centred sweep: xmax_bits holds M^2, the threshold goes to b's domain) */
			thrE = (R == R_IVF_COS && cen) ? s16c_cos_t_from_ref(thr, dim)
				: (R == R_IVF_IP && cen == 2) ? s16c_ip_t_from_ref(thr, qn2[q], __uint_as_float(*xmax_bits), dim)
				: s16_thr_from_ref<R>(thr, cen ? 0.0f : e, dim);
	}
	if (lane == 0)
		qthr[q] = make_float2(thrE, e);
}

/*
 * The first threshold when the planes are regrouped (L2, float4 rows): k_s16_seed scores the query's first 64
 * candidates, which belong to its nearest LIST — and say nothing about the query when that list is one of the
 * long mixed ones.  Here the seeds are the first 64 rows of the SUBLIST whose centre is nearest to the query
 * (among the sublists of its probed lists that hold at least k visible rows): the same exact arithmetic, the same
 * rule (the k-th smallest of distinct candidates bounds the k-th distance); without such a sublist, k_s16_seed's
 * own candidates.  One wave per query.
 */
template <int R, int H16 = 0>
__global__ __launch_bounds__(64) void
k_s16_seed_sub(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
			   const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t k, const uint32_t *__restrict__ sub_first,
			   const int *__restrict__ sub_gidx, const uint32_t *__restrict__ sub_len, const int64_t *__restrict__ sub_loc,
			   const int64_t *__restrict__ perm, const uint32_t *__restrict__ pos_of, const float *__restrict__ subdist,
			   uint32_t sstride, const float *__restrict__ pdist, const float *__restrict__ cdist, uint32_t cstride,
			   const float *__restrict__ qn2, const uint32_t *__restrict__ xmax_bits, float2 *__restrict__ qthr,
			   int cen = 0, const float *__restrict__ cn2_sub = nullptr, const float *__restrict__ cn2_list = nullptr,
			   int e_sub = 0 /* the mirror holds fp16 subnormals (s16_e) */,
			   uint32_t nseed = S16_SEED /* rows scored (<= S16_SEED): any subset of a query's candidates gives a valid threshold;
										  * 32 halve this kernel's row traffic where k leaves room (k <= 20) */ )
{
	const uint32_t q = blockIdx.x;
	const int	lane = threadIdx.x;
	const uint32_t *lco = loc_cand_off + (size_t) q * (npr + 1);
	const int	dim = ix.dim;
	const float e = s16_e<R>(dim, qn2[q], __uint_as_float(*xmax_bits), e_sub != 0);
	float		bd = __uint_as_float(0x7F800000u);
	uint32_t	bs = 0xFFFFFFFFu, bp = 0;

	/* probes one after the other (wave-uniform), a list's sublists spread over the lanes: coalesced reads, a
	 * handful of iterations even for a list of 200 sublists */
	for (int p = 0; p < npr; p++)
	{
		const uint32_t vis = lco[p + 1] - lco[p];
		const int	L = probes[(size_t) q * npr + p];

		if (vis < k || L < 0 || L >= ix.ncent)
			continue;
		const uint32_t s0 = sub_first[L], s1 = sub_first[L + 1];
		const float pd = R == R_IVF_COS ? 0.0f : (cdist ? cdist[(size_t) q * cstride + L] : pdist[(size_t) q * npr + p]);

		for (uint32_t s = s0 + (uint32_t) lane; s < s1; s += 64)
		{
			if (sub_len[s] < k)
				continue;
			const int	gi = sub_gidx[s];

			if (R == R_IVF_COS && gi < 0)
				continue;			/* (the distances to such a list's centroid are in the rows' own space) */
			float		dd = gi < 0 ? pd * pd : fmaxf(subdist[(size_t) q * sstride + gi], 0.0f);	/* both squared */

			/* inner product: "nearest" = the largest q.c = (|q|^2 + |c|^2 - |q - c|^2) / 2 (any choice is valid;
			 * this one finds the query's own neighbourhood) */
			if ((R == R_IVF_IP || R == R_IVF_COS) && cn2_sub)
				dd = dd - (gi < 0 ? cn2_list[L] : cn2_sub[gi]);

			if (dd < bd)
			{
				bd = dd;
				bs = s;
				bp = (uint32_t) p;
			}
		}
	}
	/* the wave's nearest (ties: the lower lane) */
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		const float od = __shfl_xor(bd, off, 64);
		const uint32_t os = (uint32_t) __shfl_xor((int) bs, off, 64), op = (uint32_t) __shfl_xor((int) bp, off, 64);
		const bool	take = od < bd || (od == bd && os < bs);

		if (take)
		{
			bd = od;
			bs = os;
			bp = op;
		}
	}
	bool		ok = false;
	float		v = 0.0f;

	if (bs != 0xFFFFFFFFu)		/* uniform */
	{
		const uint32_t vis = lco[bp + 1] - lco[bp];
		const uint32_t n = min(sub_len[bs], min(nseed, (uint32_t) S16_SEED));

		if ((uint32_t) lane < n)
		{
			const int64_t prow = sub_loc[bs] + lane;

			ok = pos_of[prow] < vis;			/* a candidate of this (query, probe) under the candidate cap */
			if (ok)
			{
				if (H16)
					v = scr_exact_h<R, H16 == 1>(queries + (size_t) q * dim, (const uint16_t *) ix.vecs + (size_t) perm[prow] * (size_t) dim, dim);
				else
					v = scr_exact<R>(queries + (size_t) q * dim, ix.vecs + (size_t) perm[prow] * (size_t) dim, dim);
			}
		}
	}
	if (__ballot(ok) == 0 || (uint32_t) __popcll(__ballot(ok)) < k)
	{
		/* no sublist holds k visible rows (short lists, a tight candidate cap): k_s16_seed's rule, the query's
		 * first 64 candidates */
		const uint32_t n = min(lco[npr], min(nseed, (uint32_t) S16_SEED));

		ok = (uint32_t) lane < n;
		v = 0.0f;
		if (ok)
		{
			const uint32_t p = find_probe(lco, npr, (uint32_t) lane);
			const int	L = probes[(size_t) q * npr + p];
			const size_t row = (size_t) ix.loc_off[L] + ((uint32_t) lane - lco[p]);

			if (H16)
				v = scr_exact_h<R, H16 == 1>(queries + (size_t) q * dim, (const uint16_t *) ix.vecs + row * (size_t) dim, dim);
			else
				v = scr_exact<R>(queries + (size_t) q * dim, ix.vecs + row * (size_t) dim, dim);
		}
	}
	const uint32_t key = ok ? ndb_key_from_bits(__float_as_uint(v)) : 0xFFFFFFFFu;
	uint32_t	rank = 0;

	for (int j = 0; j < S16_SEED; j++)
	{
		const uint32_t kj = (uint32_t) __shfl((int) key, j, 64);

		rank += (kj < key || (kj == key && j < lane)) ? 1u : 0u;
	}
	const unsigned long long have = __ballot(ok);
	const unsigned long long pick = __ballot(ok && rank == k - 1);
	float		thrE = __uint_as_float(0x7F800000u);	/* +inf */

	if ((uint32_t) __popcll(have) >= k && pick)
	{
		const float thr = __shfl(v, __ffsll((long long) pick) - 1, 64);

		if (thr == thr)			/* a NaN distance bounds nothing */
			/* (cen == 2, inner product on the centred sweep: xmax_bits holds M^2, the threshold goes to b's domain) */
			thrE = (R == R_IVF_COS && cen) ? s16c_cos_t_from_ref(thr, dim)
				: (R == R_IVF_IP && cen == 2) ? s16c_ip_t_from_ref(thr, qn2[q], __uint_as_float(*xmax_bits), dim)
				: s16_thr_from_ref<R>(thr, cen ? 0.0f : e, dim);
	}
	if (lane == 0)
		qthr[q] = make_float2(thrE, e);
}


/* ------------------------------------------------------------------------------------------------------------
 * The sweep.  Work item = (list, RT-row tile, 128-query tile) from the same queues as the other grouped scans
 * (k_pair_offsets with 8 groups of 16 queries and RT / 64 tiles of 64 rows per item).  NW waves per block; wave
 * (wq, wr) = (w & 1, w >> 1) owns the 64 queries x 64 rows sub-tile as 2 x 2 MFMA blocks, so RT = 32 NW:
 * 128 rows for 4 waves, 256 for 8.
 *
 * LDS image of a 32-row (or 32-query) block for one 32-dimension chunk: [r][8 slots of 16 bytes], logical slot
 * s = 2 * kstep + khalf for the hi plane, 4 + 2 * kstep + khalf for the lo plane, stored at slot s ^ ((r >> 1) & 7):
 * the 16 lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) then hit 16 distinct 16-byte bank
 * slots (SQ_LDS_BANK_CONFLICT = 0).  The DMA writes LDS linearly in lane order (lane i -> byte 16 i of the
 * instruction's 1 KiB), so the swizzle is applied to the global address each lane reads: 8 rows x 128
 * contiguous bytes per instruction.  H16 rows: the mirror's own fp16 row is the hi plane (64 bytes per row and
 * chunk, slot s ^ ((r >> 2) & 3)).
 *
 * Pipeline: a ring of NBUF chunk buffers, the DMA runs NBUF - 1 chunks ahead.  Per chunk: wait until the chunk's
 * own DMA has landed (s_waitcnt vmcnt leaves the later chunks' requests in flight), one barrier, issue the DMA of
 * chunk c + NBUF - 1 into the buffer every wave finished reading before that barrier, then 8 ds_read_b128 +
 * 12 MFMAs per k-step.  Every 64 dimensions the block accumulator is added to the running sum and restarted
 * from zero (ndbhip_common.h (4)).
 * ------------------------------------------------------------------------------------------------------------ */
template <int H16, int NW> struct S16Geom
{
	static constexpr int RT = 32 * NW;						/* rows per tile */
	static constexpr int ROW_BLK = H16 ? 2048 : 4096;		/* bytes of a 32-row block per chunk */
	static constexpr int ROWS_BYTES = NW * ROW_BLK;
	static constexpr int Q_OFF = ROWS_BYTES;
	static constexpr int BUF = ROWS_BYTES + 4 * 4096;		/* + 128 queries x 128 bytes */
	static constexpr int ROW_CHUNK = H16 ? 64 : 128;		/* bytes of a row per chunk */
	static constexpr int ROW_DMA = H16 ? 2 : 4;				/* DMA instructions per wave, chunk and row block */
	static constexpr int Q_DMA = 16 / NW;					/* ... and for the wave's share of the 4 query blocks */
	static constexpr int PER = ROW_DMA + Q_DMA;
};

typedef const __attribute__((address_space(1))) void *ndb_glb_ptr;

/*
 * One LDS-DMA instruction: every lane fetches 16 bytes from base + voff (base wave-uniform) and the wave's 1 KiB
 * lands at LDS address `la` (wave-uniform), lane i at la + 16 i.  Written as inline asm on purpose: with the
 * __builtin_amdgcn_global_load_lds form hipcc tracks "an LDS DMA may be pending" per LDS object and puts
 * s_waitcnt vmcnt(0) in front of every later ds_read it cannot prove disjoint — all of them once the chunk
 * buffers are a ring with a run-time index — which waits for the chunk just requested and serialises the whole
 * pipeline (measured: the 3-deep ring ran slower than the 2-deep one).  The kernel orders DMA against reads
 * itself: s_waitcnt vmcnt(n) + barrier before a buffer is read, a barrier before it is refilled.  M0 is
 * written here and nowhere else in the kernel (tools/check_asm_hazards.py checks that — M0 is a reserved
 * register hipcc does not model as a clobber, so the guarantee has to come from the generated code); one wait
 * state between the SALU write of M0 and its use.
 */
template <bool NT = false>
__device__ __forceinline__ void
s16_dma16(const unsigned char *base, uint32_t voff, uint32_t la)
{
	if constexpr (NT)
		asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt"
					 :: "s"(la), "v"(voff), "s"(base) : "memory");
	else
		asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
					 :: "s"(la), "v"(voff), "s"(base) : "memory");
}

/* the same for a piece that is contiguous on both sides: N instructions, lane i of instruction j copies the 16
 * bytes at base + 1024 j + 16 i to la + 1024 j + 16 i (the instruction offset applies to both addresses) */
template <int N, bool NT = false>
__device__ __forceinline__ void
s16_dma_linear(const unsigned char *base, uint32_t lane16, uint32_t la)
{
	if constexpr (N == 4 && NT)
		/* non-temporal: a stream that is read once (by the blocks sharing it now) should not push out of the L2 what
		 * is read again */
		asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
					 "global_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
					 "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 nt"
					 :: "s"(la), "v"(lane16), "s"(base) : "memory");
	else if constexpr (N == 4)
		asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
					 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
					 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
					 :: "s"(la), "v"(lane16), "s"(base) : "memory");
	else
		asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
					 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
					 :: "s"(la), "v"(lane16), "s"(base) : "memory");
}

__device__ __forceinline__ const unsigned char *
s16_uniform_ptr(const unsigned char *p)
{
	const uint64_t v = (uint64_t) p;
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) v), hi = __builtin_amdgcn_readfirstlane((uint32_t) (v >> 32));

	return (const unsigned char *) (((uint64_t) hi << 32) | lo);
}


/* one work item of the sweep, expanded once per batch (k_s16_items) so that a block finds its next item with one
 * load instead of a binary search over the lists */
struct S16Desc
{
	uint32_t	L;				/* list */
	uint32_t	t2;				/* row tile of the list */
	uint32_t	qt;				/* 128-query tile of the list's (query, probe) pairs */
	uint32_t	pad;
};

/* Consecutive items of a list walk super-tiles of up to 8 row tiles x 8 query tiles (query tile fastest inside):
 * the blocks of an XCD that pull them together then share 8 + 8 operand tiles through that XCD's L2 — with
 * "query tile fastest" across ALL of a list's query tiles (32 of them where every query probes the list) the
 * 12.6 MB of query planes went round the 4 MB L2 once per row tile (hit rate 0.45 -> 0.59) */
__global__ void
k_s16_items(const uint32_t *__restrict__ item_off, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ own_len,
			int ncent, uint32_t rt, uint32_t cap, S16Desc *__restrict__ desc, unsigned int *__restrict__ flags,
			uint32_t qtile = S16_QT /* (query, probe) pairs per tile */,
			unsigned long long *__restrict__ plane_bytes = nullptr /* statistics: += bytes of row planes the items read, every row tile once */,
			uint32_t blk_bytes = 0 /* bytes of a 32-row block over all chunks */ )
{
	const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long mine = 0;

	if (item == 0 && item_off[ncent] > cap)
		atomicAdd(flags, 1u);		/* cannot happen (the host sizes the table by an upper bound); if it does, fall back */
	const bool	live = item < item_off[ncent] && item < cap;

	if (!live && !plane_bytes)
		return;
	if (live)
	{
	uint32_t	lo = 0, hi = (uint32_t) ncent;

	while (hi - lo > 1)
	{
		const uint32_t mid = (lo + hi) >> 1;

		if (item_off[mid] <= item)
			lo = mid;
		else
			hi = mid;
	}
	while (lo + 1 < (uint32_t) ncent && item_off[lo + 1] <= item)
		lo++;
	const uint32_t L = lo, local = item - item_off[L];
	const uint32_t nqt = (cnt[L] + qtile - 1) / qtile, nrt = (own_len[L] + rt - 1) / rt;
	const uint32_t band = local / (8u * nqt);			/* bands of 8 row tiles: all but the last are full */
	const uint32_t lb = local - band * 8u * nqt;
	const uint32_t br = min(8u, nrt - 8u * band);
	const uint32_t grp = lb / (br * 8u);				/* groups of 8 query tiles: all but the last are full */
	const uint32_t lg = lb - grp * br * 8u;
	const uint32_t gq = min(8u, nqt - 8u * grp);
	S16Desc		d;

	d.L = L;
	d.t2 = 8u * band + lg / gq;
	d.qt = 8u * grp + lg % gq;
	d.pad = 0;
	desc[item] = d;
	if (plane_bytes && d.qt == 0)
		mine = (unsigned long long) min(rt / 32u, (own_len[L] - d.t2 * rt + 31u) / 32u) * blk_bytes;
	}
	if (plane_bytes)
	{
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
		{
			const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) mine, off, 64);
			const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (mine >> 32), off, 64);

			mine += ((unsigned long long) hi << 32) | lo;
		}
		if ((threadIdx.x & 63) == 0 && mine != 0)
			atomicAdd(plane_bytes, mine);
	}
}

#define S16_NOITEM 0xFFFFFFFFu
#ifndef S16_TIGHT
#define S16_TIGHT 128			/* a query's threshold is tightened every time it has emitted this many more records */
#endif
#define S16_TIGHT_Q 8			/* such queries a block handles per item (more: the next multiple gets them) */
#ifndef S16_SETPRIO
#define S16_SETPRIO 0
#endif

/* MODE 1 / 2 (the build's assignment, ndbhip_build.h: assign_rows_s16): the "queries" are the distinct centroids,
 * one list holds a slab of rows.  MODE 1 leaves bmin[row] = the row's smallest a over all centroids (order key).
 * MODE 2 records, per row, the centroids whose a lies within the error bound of that minimum: ecount[row] = how
 * many, erec[row][S16_ASSIGN_SLOTS] = (centroid, a).  qthr has one slot per centroid like in a search, of which
 * only qthr[0].x = the largest centroid norm is used. */
/* MODE 4 (round 6): both in ONE sweep.  An item first takes its tile's row minima (MODE 1), publishes them with a returning
 * atomicMin on bmin[row] and tests its elements against the threshold of min(what bmin held, the tile's own) — the row's
 * minimum SO FAR, never below the final one, so every centroid the final threshold admits is recorded (thresholds grow
 * with the minimum they come from), along with a few the final one no longer admits: a row's tiles leave a record whenever
 * they set a new low, 2.7 of 8 tiles on average.  k_s16_assign_resolve then applies MODE 2's test with the FINAL minimum to
 * the records: the same candidate set as the two sweeps', hence the same lists; the matrix is multiplied once. */
#define S16_ASSIGN_SLOTS 16

template <int R, int H16, int NW, int NBUF, int DBG = 0, int MODE = 0>
__global__ __launch_bounds__(64 * NW, (NW == 4 && NBUF == 2) ? 2 : 1) void
k_s16_sweep(IvfDev ix, const unsigned char *__restrict__ planes, const uint32_t *__restrict__ blk_off,
			const float *__restrict__ rn2, const int16_t *__restrict__ rexp,
			const unsigned char *__restrict__ qplanes, uint32_t qrowbytes, const float *__restrict__ qn2,
			const int *__restrict__ qexp, float2 *qthr,
			const uint32_t *__restrict__ loc_cand_off, int npr, const uint32_t *__restrict__ cnt,
			const uint32_t *__restrict__ pair_off, const S16Desc *__restrict__ desc,
			const PairRec *__restrict__ pairs, unsigned int *__restrict__ next_item,
			const uint32_t *__restrict__ runs, unsigned int *__restrict__ ecount, uint2 *__restrict__ erec,
			uint32_t ecap, uint32_t *__restrict__ bmin, int polite, int nchunk, uint32_t desc_cap, uint32_t topk,
			const uint32_t *__restrict__ pos_of = nullptr /* plane row -> the row's index in its list (sublists) */ )
{
	typedef S16Geom<H16, NW> G;
	__shared__ uint32_t s_tn, s_tq[S16_TIGHT_Q], s_tkeys[S16_NB];	/* queries whose threshold is due for tightening */
	__shared__ __attribute__((aligned(1024))) unsigned char ring[NBUF * G::BUF];
	__shared__ S16Q qinfo[2][S16_QT];
	__shared__ uint32_t s_desc[2][5];		/* item (S16_NOITEM = none), L, t2, qt, members of the current / next item's tile */
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const int	wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	wq = wave & 1, wr = wave >> 1;
	const int	r32 = lane & 31, kh = lane >> 5;
	/* work queues: thread 0 only */
	uint32_t	hop = 0;

	/* next item of this block's queues, or S16_NOITEM; `got` = the value a previously issued atomicAdd on the
	 * current hop's head returned (0xFFFFFFFF: none issued) */
	auto		pop = [&](uint32_t got) -> uint32_t {
		for (; hop < 8; hop++)
		{
			const uint32_t xq = (blockIdx.x + hop) & 7u;
			const uint32_t run_lo = runs[xq], run_hi = runs[xq + 1];

			if (run_lo != run_hi)
			{
				if (got == 0xFFFFFFFFu)
					got = (polite && run_lo + __hip_atomic_load(next_item + xq * NDB_QHEAD_STRIDE, __ATOMIC_RELAXED,
																__HIP_MEMORY_SCOPE_AGENT) >= run_hi)
						? run_hi : atomicAdd(next_item + xq * NDB_QHEAD_STRIDE, 1u);
				if (got < run_hi - run_lo && run_lo + got < desc_cap)	/* (the table holds every item: k_s16_items flags it otherwise) */
					return run_lo + got;
			}
			got = 0xFFFFFFFFu;
		}
		return S16_NOITEM;
	};
	/* thread 0: publish item `it` (and its descriptor) in slot sl */
	auto		put_desc = [&](int sl, uint32_t it, const S16Desc &d) {
		s_desc[sl][0] = it;
		s_desc[sl][1] = d.L;
		s_desc[sl][2] = d.t2;
		s_desc[sl][3] = d.qt;
		s_desc[sl][4] = it == S16_NOITEM ? 0u : min((uint32_t) S16_QT, cnt[d.L] - d.qt * S16_QT);
	};
	/* threads < 128: the tile's member record */
	auto		load_pair = [&](int sl, PairRec &pr) -> bool {
		const uint32_t L = s_desc[sl][1], qt = s_desc[sl][3];

		if (s_desc[sl][0] == S16_NOITEM || qt * S16_QT + (uint32_t) tid >= cnt[L])
			return false;
		pr = pairs[pair_off[L] + qt * S16_QT + (uint32_t) tid];
		return true;
	};
	auto		load_qinfo = [&](bool have, const PairRec &pr, S16Q &qi) {
		qi.q2 = 0.0f; qi.thrE = 0.0f; qi.eq = 0; qi.la = 0; qi.nrow = 0; qi.qid = 0;
		if (have)
		{
			const uint32_t *lq = loc_cand_off + (size_t) pr.q * (npr + 1);

			qi.qid = pr.q;
			qi.la = lq[pr.p];
			qi.nrow = lq[pr.p + 1] - qi.la;
			qi.q2 = qn2[pr.q];
			qi.eq = qexp[pr.q];
			qi.thrE = qthr[pr.q].x;
		}
	};

	/* ---- first item: everything synchronously ---- */
	if (tid == 0)
	{
		s_tn = 0;
		const uint32_t it = pop(0xFFFFFFFFu);
		S16Desc		d = {0, 0, 0, 0};

		if (it != S16_NOITEM)
			d = desc[it];
		put_desc(0, it, d);
	}
	__syncthreads();
	if (s_desc[0][0] == S16_NOITEM)
		return;					/* uniform */
	if (tid < S16_QT)
	{
		PairRec		pr = {0, 0};
		S16Q		qi;
		const bool	have = load_pair(0, pr);

		load_qinfo(have, pr, qi);
		qinfo[0][tid] = qi;
	}
	__syncthreads();

	uint32_t	voff_q[G::Q_DMA];
	bool		qdma = true;	/* this wave's share of the query area holds a member (wave-uniform) */
	const unsigned char *rbase;
	const uint32_t lane16 = (uint32_t) lane * 16u;

	/* DMA addresses for the item in slot sl: wave w stages 32-row block w of the tile (one contiguous piece of
	 * the blocked planes per chunk) and its share of the 4 query blocks (gathered by query id) */
	auto		set_dma = [&](int sl) {
		const uint32_t L = s_desc[sl][1], t2 = s_desc[sl][2];
		/* a tile's last blocks may lie beyond the list's last block: they read that one again (results of rows
		 * >= the list's length are never looked at) */
		const uint32_t nb = blk_off[L + 1] - blk_off[L];
		const uint32_t b = min(t2 * (uint32_t) NW + (uint32_t) wave, nb - 1u);
		/* 1 KiB pieces of the 16 KiB query area, in order: piece -> query block piece / 4, rows 8 (piece % 4) .. + 7 */
#pragma unroll
		for (int j = 0; j < G::Q_DMA; j++)
		{
			const int	piece = wave * G::Q_DMA + j;
			const int	rr = 8 * (piece & 3) + (lane >> 3);

			voff_q[j] = qinfo[sl][32 * (piece >> 2) + rr].qid * qrowbytes + 16u * (uint32_t) ((lane & 7) ^ ((rr >> 1) & 7));
		}
		rbase = planes + ((size_t) blk_off[L] + b) * (size_t) nchunk * G::ROW_BLK;
		/* most lists are probed by a handful of queries: their tiles leave three of the four query blocks empty,
		 * and an empty block is neither fetched nor multiplied (pieces wave * Q_DMA .. belong to block
		 * (wave * Q_DMA) / 4) */
		qdma = s_desc[sl][4] > (uint32_t) (32 * ((wave * G::Q_DMA) >> 2));
	};
	const uint32_t ring_la = (uint32_t) (uintptr_t) (ndb_lds_ptr) ring;
	auto		issue = [&](int c, int bufi) {
		/* DBG (timing experiments only, results are garbage): 1 = no DMA at all, 2 = every chunk re-reads
		 * chunk 0 of the mirror's first rows and of query 0 (always cache hits) */
		if constexpr (DBG == 1)
			return;
		/* (3: only the queries are always hits, 4: only the rows) */
		const unsigned char *rb = s16_uniform_ptr((DBG == 2 || DBG == 4) ? planes : rbase + (size_t) c * G::ROW_BLK);
		const unsigned char *qb = s16_uniform_ptr((DBG == 2 || DBG == 3) ? qplanes : qplanes + (size_t) c * 128);
		const uint32_t la = ring_la + (uint32_t) bufi * G::BUF;

		s16_dma_linear<G::ROW_DMA>(rb, lane16, la + wave * G::ROW_BLK);
		if (qdma)
		{
#pragma unroll
			for (int j = 0; j < G::Q_DMA; j++)
				s16_dma16(qb, voff_q[j], la + G::Q_OFF + (wave * G::Q_DMA + j) * 1024);
		}
	};

	/* fragment addresses (bytes inside a buffer) */
	const int	qsw = (r32 >> 1) & 7;
	const int	qfrag = G::Q_OFF + (2 * wq) * 4096 + r32 * 128;
	const int	rsw = H16 ? ((r32 >> 2) & 3) : ((r32 >> 1) & 7);
	const int	rfrag = (2 * wr) * G::ROW_BLK + r32 * G::ROW_CHUNK;
	int			cur = 0;

	set_dma(0);
	/* nchunk is even (the planes are padded to 64 dimensions) and >= 2 */
#pragma unroll
	for (int p = 0; p < NBUF - 1; p++)
		if (p < nchunk)
			issue(p, p);

	for (;;)
	{
		const int	nxt = cur ^ 1;
		ndb_f16acc	run[2][2], blk[2][2];

#pragma unroll
		for (int a = 0; a < 2; a++)
#pragma unroll
			for (int b = 0; b < 2; b++)
#pragma unroll
				for (int i = 0; i < 16; i++)
				{
					run[a][b][i] = 0.0f;
					blk[a][b][i] = 0.0f;
				}

		/* query blocks of this wave's half that hold a member of the current item's tile */
		const uint32_t nmem_cur = s_desc[cur][4];
		const int	na = nmem_cur > (uint32_t) (64 * wq + 32) ? 2 : (nmem_cur > (uint32_t) (64 * wq) ? 1 : 0);

		/* the first query block only */
		auto		compute_half = [&](const unsigned char *buf) {
			ndb_h8		ah[2], al[2], bh[2][2], bl[2][2];

#pragma unroll
			for (int s = 0; s < 2; s++)
			{
				ah[s] = *reinterpret_cast<const ndb_h8 *>(buf + qfrag + (((2 * s + kh) ^ qsw) * 16));
				al[s] = *reinterpret_cast<const ndb_h8 *>(buf + qfrag + (((4 + 2 * s + kh) ^ qsw) * 16));
#pragma unroll
				for (int b = 0; b < 2; b++)
				{
					bh[s][b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * G::ROW_BLK + (((2 * s + kh) ^ rsw) * 16));
					if constexpr (H16 == 0)
						bl[s][b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * G::ROW_BLK + (((4 + 2 * s + kh) ^ rsw) * 16));
				}
			}
#pragma unroll
			for (int s = 0; s < 2; s++)
			{
#pragma unroll
				for (int b = 0; b < 2; b++)
					blk[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s][b], blk[0][b], 0, 0, 0);
				if constexpr (H16 == 0)
				{
#pragma unroll
					for (int b = 0; b < 2; b++)
						blk[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s][b], blk[0][b], 0, 0, 0);
				}
#pragma unroll
				for (int b = 0; b < 2; b++)
					blk[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s][b], blk[0][b], 0, 0, 0);
			}
		};
		/* both k-steps' fragments are requested before the first k-step is multiplied: the second set arrives
		 * under the first 12 MFMAs instead of after them */
		/* a short query tile (most lists are probed by a handful of queries) leaves query blocks without a
		 * member: a wave multiplies only the blocks of its half that have one (na = 0, 1 or 2, wave-uniform) */
		auto		compute = [&](const unsigned char *buf) {
			if (na == 0)
				return;
			if (na == 1)
			{
				compute_half(buf);
				return;
			}
			ndb_h8		ah[2][2], al[2][2], bh[2][2], bl[2][2];

#pragma unroll
			for (int s = 0; s < 2; s++)
#pragma unroll
				for (int b = 0; b < 2; b++)
				{
					ah[s][b] = *reinterpret_cast<const ndb_h8 *>(buf + qfrag + b * 4096 + (((2 * s + kh) ^ qsw) * 16));
					al[s][b] = *reinterpret_cast<const ndb_h8 *>(buf + qfrag + b * 4096 + (((4 + 2 * s + kh) ^ qsw) * 16));
					bh[s][b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * G::ROW_BLK + (((2 * s + kh) ^ rsw) * 16));
					if constexpr (H16 == 0)
						bl[s][b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * G::ROW_BLK + (((4 + 2 * s + kh) ^ rsw) * 16));
				}
#if S16_SETPRIO == 1
			__builtin_amdgcn_s_setprio(1);
#elif S16_SETPRIO == 2
			__builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
			for (int s = 0; s < 2; s++)
			{
				/* the three products of a block are spread over the step, so that no MFMA follows the one it
				 * depends on */
#pragma unroll
				for (int a = 0; a < 2; a++)
#pragma unroll
					for (int b = 0; b < 2; b++)
						blk[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][a], bh[s][b], blk[a][b], 0, 0, 0);
				if constexpr (H16 == 0)
				{
#pragma unroll
					for (int a = 0; a < 2; a++)
#pragma unroll
						for (int b = 0; b < 2; b++)
							blk[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][a], bl[s][b], blk[a][b], 0, 0, 0);
				}
#pragma unroll
				for (int a = 0; a < 2; a++)
#pragma unroll
					for (int b = 0; b < 2; b++)
						blk[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][a], bh[s][b], blk[a][b], 0, 0, 0);
			}
#if S16_SETPRIO == 1
			__builtin_amdgcn_s_setprio(0);
#elif S16_SETPRIO == 2
			__builtin_amdgcn_s_setprio(1);		/* requests (DMA, fragment reads) go ahead of the other wave's MFMAs */
#endif
		};
		auto		flush = [&]() {
#pragma unroll
			for (int a = 0; a < 2; a++)
#pragma unroll
				for (int b = 0; b < 2; b++)
#pragma unroll
					for (int i = 0; i < 16; i++)
					{
						run[a][b][i] = run[a][b][i] + blk[a][b][i];
						blk[a][b][i] = 0.0f;
					}
		};

		/*
		 * The next item is prepared while this one is multiplied, one stage per chunk, each stage right after
		 * the chunk's barrier: a stage consumes what the previous stage requested one chunk earlier (so the
		 * request had a chunk of MFMAs to arrive) and what it writes to LDS is published by the next barrier.
		 *   0  thread 0: atomicAdd on its queue head          3  threads < 128: the member's (query, probe) record
		 *   1  thread 0: the item's descriptor                4  threads < 128: its offsets, norm, exponent, threshold
		 *   2  thread 0: descriptor -> LDS                    5  threads < 128: record -> LDS
		 * Items with fewer than 6 chunks run the remaining stages after their last chunk.
		 */
		uint32_t	got = 0xFFFFFFFFu, nit = S16_NOITEM;
		S16Desc		nd = {0, 0, 0, 0};
		PairRec		npair = {0, 0};
		bool		nhave = false;
		S16Q		nqi;

		auto		stage = [&](int st) {
			if (st == 0)
			{
				if (tid == 0 && hop < 8 && !polite)
				{
					const uint32_t xq = (blockIdx.x + hop) & 7u;

					if (runs[xq] != runs[xq + 1])
						got = atomicAdd(next_item + xq * NDB_QHEAD_STRIDE, 1u);
				}
			}
			else if (st == 1)
			{
				if (tid == 0)
				{
					nit = pop(got);
					if (nit != S16_NOITEM)
						nd = desc[nit];
				}
			}
			else if (st == 2)
			{
				if (tid == 0)
					put_desc(nxt, nit, nd);
			}
			else if (st == 3)
			{
				if (tid < S16_QT)
					nhave = load_pair(nxt, npair);
			}
			else if (st == 4)
			{
				if (tid < S16_QT)
					load_qinfo(nhave, npair, nqi);
			}
			else if (st == 5)
			{
				if (tid < S16_QT)
					qinfo[nxt][tid] = nqi;
			}
		};

		if constexpr (NBUF == 2)
		{
			for (int c = 0; c < nchunk; c += 2)
			{
				s16_wait_vm<0>();
				__syncthreads();
				if (c < 6)
					stage(c);
				issue(c + 1, 1);
				compute(ring);
				s16_wait_vm<0>();
				__syncthreads();
				if (c + 1 < 6)
					stage(c + 1);
				if (c + 2 < nchunk)
					issue(c + 2, 0);
				compute(ring + G::BUF);
				flush();			/* NDB_S16_FLUSH_DIMS = 2 chunks */
			}
		}
		else
		{
			int			bc = 0;			/* buffer of chunk c; the DMA target is (bc + NBUF - 1) % NBUF */

			for (int c = 0; c < nchunk; c++)
			{
				/* the chunks after c that are already requested: min(NBUF - 2, nchunk - 1 - c) */
				if (c + NBUF - 2 < nchunk)
				{
					if (qdma)
						s16_wait_vm<G::PER * (NBUF - 2)>();
					else
						s16_wait_vm<G::ROW_DMA * (NBUF - 2)>();	/* this wave requests rows only */
				}
				else
					s16_wait_vm<0>();		/* (the tail of a ring deeper than 3 drains a little early) */
				__syncthreads();
				if (c < 6)
					stage(c);
				const int	bt = bc == 0 ? NBUF - 1 : bc - 1;

				if (c + NBUF - 1 < nchunk)
					issue(c + NBUF - 1, bt);
				compute(ring + bc * G::BUF);
				if (c & 1)
					flush();
				bc = bc + 1 == NBUF ? 0 : bc + 1;
			}
		}
		for (int st = nchunk; st < 6; st++)
		{
			__syncthreads();
			stage(st);
		}
		__syncthreads();		/* every wave has finished reading the ring; the next item's records are published */

		const bool	more = s_desc[nxt][0] != S16_NOITEM;	/* uniform */
		const uint32_t L = s_desc[cur][1], t2 = s_desc[cur][2];
		const uint32_t len = ix.own_len[L];

		if (more)
		{
			/* the next item's first chunks travel while this item's results are looked at */
			set_dma(nxt);
#pragma unroll
			for (int p = 0; p < NBUF - 1; p++)
				if (p < nchunk)
					issue(p, p);
		}

		/* epilogue: element (reg, lane) of block (a, b) = query 32 (2 wq + a) + (reg & 3) + 8 (reg >> 2) + 4 kh,
		 * row 32 (2 wr + b) + r32 */
		if constexpr (MODE == 1)
		{
			/* build, first sweep: the smallest a of every row over all centroids (rowmin = bmin[row]) */
			__shared__ uint32_t s_rowmin[G::RT];
			auto		aval = [&](int a, int b, int reg, const S16Q &qi, float x2, int ex) {
				const float dot = ldexpf(run[a][b][reg], qi.eq + ex - 28);

				return __builtin_fmaf(-2.0f, dot, qi.q2 + x2);
			};

			if (tid < G::RT)
				s_rowmin[tid] = 0xFFFFFFFFu;
			__syncthreads();
#pragma unroll
			for (int b = 0; b < 2; b++)
			{
				const uint32_t ridx = t2 * G::RT + (uint32_t) (32 * (2 * wr + b) + r32);
				const size_t grow = (size_t) ix.loc_off[L] + (ridx < len ? ridx : len - 1);
				const float x2 = rn2[grow];
				const int	ex = (int) rexp[grow];
				uint32_t	mn = 0xFFFFFFFFu;

#pragma unroll
				for (int a = 0; a < 2; a++)
#pragma unroll
					for (int reg = 0; reg < 16; reg++)
					{
						const int	m = 32 * (2 * wq + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
						const S16Q	qi = qinfo[cur][m];

						if (qi.nrow != 0)
						{
							const float av = aval(a, b, reg, qi, x2, ex);

							/* a NaN (a row or centroid beyond fp32) bounds nothing: it is left out of the minimum
							 * and emitted by the second sweep, which sends the row to the reference's arithmetic */
							if (av == av)
								mn = min(mn, ndb_key_from_bits(__float_as_uint(av)));
						}
					}
				mn = min(mn, (uint32_t) __shfl_xor((int) mn, 32, 64));
				if (kh == 0 && ridx < len)
					atomicMin(&s_rowmin[32 * (2 * wr + b) + r32], mn);
			}
			__syncthreads();
			if (tid < G::RT && t2 * G::RT + (uint32_t) tid < len)
				atomicMin(&bmin[(size_t) ix.loc_off[L] + t2 * G::RT + tid], s_rowmin[tid]);
			__syncthreads();		/* s_rowmin is reused by the next item */
		}
		else if constexpr (MODE == 3)
		{
			/* distance matrix (sublist centres as the rows of one list): a of every (query, row) pair, no test;
			 * erec is a float array [query][ecap] */
			float	   *out = reinterpret_cast<float *>(erec);

#pragma unroll
			for (int b = 0; b < 2; b++)
			{
				const uint32_t ridx = t2 * G::RT + (uint32_t) (32 * (2 * wr + b) + r32);

				if (ridx >= len)
					continue;
				const size_t grow = (size_t) ix.loc_off[L] + ridx;
				const float x2 = rn2[grow];
				const int	ex = (int) rexp[grow];

#pragma unroll
				for (int a = 0; a < 2; a++)
#pragma unroll
					for (int reg = 0; reg < 16; reg++)
					{
						const int	m = 32 * (2 * wq + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
						const S16Q	qi = qinfo[cur][m];

						if (qi.nrow != 0)
						{
							const float dot = ldexpf(run[a][b][reg], qi.eq + ex - 28);

							out[(size_t) qi.qid * ecap + ridx] = __builtin_fmaf(-2.0f, dot, qi.q2 + x2);
						}
					}
			}
		}
		else if constexpr (MODE == 2 || MODE == 4)
		{
			/* build, second sweep: every centroid within reach of the row's minimum, S16_ASSIGN_SLOTS records a row */
			const float c2max = qthr[0].x;
			__shared__ uint32_t s_rowlow[MODE == 4 ? G::RT : 1];

			if constexpr (MODE == 4)
			{
				/* the tile's row minima (MODE 1's code), published; s_rowlow = the rows' minima so far, this tile included */
				if (tid < G::RT)
					s_rowlow[tid] = 0xFFFFFFFFu;
				__syncthreads();
#pragma unroll
				for (int b = 0; b < 2; b++)
				{
					const uint32_t ridx = t2 * G::RT + (uint32_t) (32 * (2 * wr + b) + r32);
					const size_t grow = (size_t) ix.loc_off[L] + (ridx < len ? ridx : len - 1);
					const float x2 = rn2[grow];
					const int	ex = (int) rexp[grow];
					uint32_t	mn = 0xFFFFFFFFu;

#pragma unroll
					for (int a = 0; a < 2; a++)
#pragma unroll
						for (int reg = 0; reg < 16; reg++)
						{
							const int	m = 32 * (2 * wq + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
							const S16Q	qi = qinfo[cur][m];

							if (qi.nrow != 0)
							{
								const float dot = ldexpf(run[a][b][reg], qi.eq + ex - 28);
								const float av = __builtin_fmaf(-2.0f, dot, qi.q2 + x2);

								if (av == av)		/* (a NaN bounds nothing: left out of the minimum, emitted below) */
									mn = min(mn, ndb_key_from_bits(__float_as_uint(av)));
							}
						}
					mn = min(mn, (uint32_t) __shfl_xor((int) mn, 32, 64));
					if (kh == 0 && ridx < len)
						atomicMin(&s_rowlow[32 * (2 * wr + b) + r32], mn);
				}
				__syncthreads();
				if (tid < G::RT && t2 * G::RT + (uint32_t) tid < len)
				{
					const uint32_t mine = s_rowlow[tid];
					const uint32_t old = atomicMin(&bmin[(size_t) ix.loc_off[L] + t2 * G::RT + tid], mine);

					s_rowlow[tid] = min(old, mine);
				}
				__syncthreads();
			}

#pragma unroll
			for (int b = 0; b < 2; b++)
			{
				const uint32_t ridx = t2 * G::RT + (uint32_t) (32 * (2 * wr + b) + r32);

				if (ridx >= len)
					continue;
				const size_t grow = (size_t) ix.loc_off[L] + ridx;
				const float x2 = rn2[grow];
				const int	ex = (int) rexp[grow];
				const uint32_t gmin = MODE == 4 ? s_rowlow[32 * (2 * wr + b) + r32] : bmin[grow];
				const uint32_t gb = (gmin & 0x80000000u) ? (gmin & 0x7FFFFFFFu) : ~gmin;
				const float e_r = s16_e<R_IVF_L2>(ix.dim, x2, c2max, false);
				const float thr = gmin == 0xFFFFFFFFu ? __uint_as_float(0x7F800000u) : s16_thr_from_a<R_IVF_L2>(__uint_as_float(gb), e_r, ix.dim);

#pragma unroll
				for (int a = 0; a < 2; a++)
#pragma unroll
					for (int reg = 0; reg < 16; reg++)
					{
						const int	m = 32 * (2 * wq + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
						const S16Q	qi = qinfo[cur][m];

						if (qi.nrow != 0)
						{
							const float dot = ldexpf(run[a][b][reg], qi.eq + ex - 28);
							const float av = __builtin_fmaf(-2.0f, dot, qi.q2 + x2);

							if (!(av > thr))
							{
								const uint32_t slot = atomicAdd(&ecount[grow], 1u);

								if (slot < S16_ASSIGN_SLOTS)
									erec[grow * S16_ASSIGN_SLOTS + slot] = make_uint2(qi.qid, __float_as_uint(av));
							}
						}
					}
			}
			if constexpr (MODE == 4)
				__syncthreads();		/* s_rowlow is reused by the next item */
		}
		else
#pragma unroll
		for (int b = 0; b < 2; b++)
		{
			const uint32_t ridx = t2 * G::RT + (uint32_t) (32 * (2 * wr + b) + r32);
			const bool	rok = ridx < len;
			const size_t grow = (size_t) ix.loc_off[L] + (rok ? ridx : len - 1);
			const float x2 = rn2[grow];
			const int	ex = H16 ? 14 : (int) rexp[grow];
			/* the row's index in its list: what positions, candidate caps and ties are defined on */
			const uint32_t porig = pos_of ? pos_of[grow] : ridx;

#pragma unroll
			for (int a = 0; a < 2; a++)
#pragma unroll
				for (int reg = 0; reg < 16; reg++)
				{
					const int	m = 32 * (2 * wq + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
					const S16Q	qi = qinfo[cur][m];

					if (rok && porig < qi.nrow)
					{
						const float dot = ldexpf(run[a][b][reg], qi.eq + ex - 28);
						float		av;

						if (R == R_IVF_L2)
							av = __builtin_fmaf(-2.0f, dot, qi.q2 + x2);
						else
							av = -dot;
						if (DBG ? av == 1234.5f : !(av > qi.thrE))
						{
							const uint32_t slot = atomicAdd(&ecount[qi.qid], 1u);
							const uint32_t pos = qi.la + porig, ab = __float_as_uint(av);

							if (slot < ecap)
								erec[(size_t) qi.qid * ecap + slot] = make_uint2(pos, ab);
							/* the smallest a of every hash bucket of positions, kept whether or not the record fit:
							 * k non-empty buckets are k distinct candidates (k_s16_retarget) */
							if ((ab & 0x7FFFFFFFu) < 0x7F800000u)
								atomicMin(&bmin[(size_t) qi.qid * S16_NB + ((pos * 2654435761u) >> (32 - S16_NB_LOG2))],
										  ndb_key_from_bits(ab));
							if (DBG == 0 && (slot & (S16_TIGHT - 1)) == S16_TIGHT - 1)
							{
								const uint32_t ti = atomicAdd(&s_tn, 1u);

								if (ti < S16_TIGHT_Q)
									s_tq[ti] = qi.qid;
							}
						}
					}
				}
		}
		if constexpr (DBG == 0)
		{
			/*
			 * A query that keeps emitting has a loose threshold (its seeds missed its own cluster).  The bucket
			 * minima its emissions have left are real candidates: the k-th smallest of them bounds its k-th
			 * distance like in k_s16_retarget, so the threshold is lowered HERE, while the sweep is still running,
			 * and the items that follow emit against it.  Any value read in between is a valid bound (the slot
			 * only ever decreases), so the result does not depend on who sees which.
			 */
			__syncthreads();
			const uint32_t tn = min(s_tn, (uint32_t) S16_TIGHT_Q);	/* uniform */

			for (uint32_t j = 0; j < tn; j++)
			{
				const uint32_t q = s_tq[j];
				uint32_t	mine = 0xFFFFFFFFu;

				if (tid < S16_NB)
				{
					mine = __hip_atomic_load(&bmin[(size_t) q * S16_NB + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					s_tkeys[tid] = mine;
				}
				__syncthreads();
				if (tid < S16_NB)
				{
					uint32_t	rank = 0;

					for (uint32_t o = 0; o < S16_NB; o++)
					{
						const uint32_t ok = s_tkeys[o];

						rank += (ok < mine || (ok == mine && o < (uint32_t) tid)) ? 1u : 0u;
					}
					if (rank == topk - 1 && mine != 0xFFFFFFFFu)
					{
						const uint32_t tb = (mine & 0x80000000u) ? (mine & 0x7FFFFFFFu) : ~mine;
						const float nt = s16_thr_from_a<R>(__uint_as_float(tb), qthr[q].y, ix.dim);
						unsigned int *px = reinterpret_cast<unsigned int *>(&qthr[q].x);
						unsigned int old = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

						while (nt < __uint_as_float(old))
						{
							const unsigned int prev = atomicCAS(px, old, __float_as_uint(nt));

							if (prev == old)
								break;
							old = prev;
						}
					}
				}
				__syncthreads();
			}
			if (tid == 0 && s_tn != 0)
				s_tn = 0;
			/* (the next write of s_tn is an emission of the next item, behind that item's barriers) */
		}
		if (!more)
			break;
		cur = nxt;
		/* (no barrier: qinfo[cur ^ 1] and s_desc[cur ^ 1] are next written in stages 2 and 5, behind barriers) */
	}
}

/*
 * Between the two rounds: a query that emitted more than `ecap` candidates in round 0 lost records, so it is
 * swept again in round 1 — against a threshold that makes the second attempt short.  Every emitted candidate,
 * kept or not, left its a in one of S16_NB hash buckets of its position (smallest a per bucket): k non-empty
 * buckets are k distinct candidates with value <= a + E each, so the k-th smallest bucket minimum bounds the k-th
 * value like the k-th smallest emitted a does (and equals it unless two of the k best share a bucket).
 * flags[0] counts the queries marked active.
 */
template <int R>
__global__ __launch_bounds__(S16_NB) void
k_s16_retarget(int dim, uint32_t k, float2 *__restrict__ qthr, unsigned int *__restrict__ ecount, uint32_t ecap,
			   const uint32_t *__restrict__ bmin, unsigned int *__restrict__ active,
			   unsigned int *__restrict__ flags, int cen = 0 /* bmin holds upper bounds, the threshold no error term */,
			   const float *__restrict__ qev = nullptr /* inner product on the centred sweep: ev per query (s16c_ip_ev) */ )
{
	__shared__ uint32_t keys[S16_NB];
	const uint32_t q = blockIdx.x;
	const uint32_t nraw = ecount[q];
	const uint32_t t = threadIdx.x;

	if (nraw <= ecap)
	{
		if (t == 0)
			active[q] = 0;
		return;					/* uniform */
	}
	const uint32_t mine = bmin[(size_t) q * S16_NB + t];

	keys[t] = mine;
	__syncthreads();
	uint32_t	rank = 0;

	for (uint32_t j = 0; j < S16_NB; j++)
	{
		const uint32_t o = keys[j];

		rank += (o < mine || (o == mine && j < t)) ? 1u : 0u;
	}
	if (rank == k - 1 && mine != 0xFFFFFFFFu)
	{
		const uint32_t tb = (mine & 0x80000000u) ? (mine & 0x7FFFFFFFu) : ~mine;
		const float2 o = qthr[q];

		const float ak = __uint_as_float(tb);
		const float nt = cen ? (R == R_IVF_COS ? s16c_cos_t_from_ub(ak, dim)
								: (R == R_IVF_IP && qev) ? s16c_ip_t_from_ub(ak, qev[q])
								: s16_up(s16_up(fmaxf(ak, 0.0f)) * (1.0f + 2.5f * ndb_s16_refslack(dim))))
			: s16_thr_from_a<R>(ak, o.y, dim);

		qthr[q] = make_float2(fminf(o.x, nt), o.y);
	}
	if (t == 0)
	{
		active[q] = 1;
		ecount[q] = 0;
		atomicAdd(&flags[0], 1u);
	}
}

/*
 * Finalize: one block per query.  rec_counts[q] = survivors rescored (statistics); flags[0] != 0 when some query
 * overflowed its records or its survivor list (the host then reruns the batch on the older path).
 */
template <int R, int H16, bool STAGED = false /* small batches: the survivors' rows come through LDS (stage_nbuf, stage_nbuf4) */>
__global__ __launch_bounds__(256) void
k_s16_finalize(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
			   const uint32_t *__restrict__ cand_off, const uint32_t *__restrict__ loc_cand_off, int npr,
			   uint32_t k, const float2 *__restrict__ qthr, const unsigned int *__restrict__ ecount,
			   const uint2 *__restrict__ erec, uint32_t ecap, int partial, ndbhip_cand *__restrict__ out_cand,
			   int *__restrict__ out_ncand, int64_t *__restrict__ out_total, uint64_t *__restrict__ out_tids,
			   float *__restrict__ out_dist, int *__restrict__ out_count, unsigned int *__restrict__ rec_counts,
			   unsigned int *__restrict__ flags,
			   const float *__restrict__ eub = nullptr /* the centred sweep (ndbhip_screen16c.h): a record holds the
														 * candidate's LOWER bound, eub its upper bound, and the
														 * threshold carries no error term */,
			   uint32_t surv_cap = S16_SURV_CAP /* survivors the block's LDS holds (a shard's k-th local bound is looser
												 * than the index's: it gets more room) */,
			   uint32_t *__restrict__ over_q = nullptr /* [S16_OVER_CAP] the queries counted in flags[0]: what the host
														* hands to the exact path one by one instead of the whole batch */,
			   const float *__restrict__ qev = nullptr /* inner product on the centred sweep: ev per query (s16c_ip_ev); the
														 * records' bounds and the threshold are in b's domain */,
			   int stage_nbuf = 0 /* > 0: up to 16 survivors' rows are streamed through a ring of that many chunk slots behind
								   * the top-k arrays (s16_exact_staged, NG = 1; the host checked s16_staged_ok) */,
			   int stage_nbuf4 = 0 /* > 0: more than 16 go 64 at a time through that many 64-row slots (NG = 4) */ )
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	TopkSmem	s = carve_topk_smem(smem_raw, surv_cap, k);
	unsigned char *stage_ring = smem_raw + ((topk_smem_bytes(surv_cap, k) + 15) & ~(size_t) 15);
	const uint32_t q = blockIdx.x;
	const uint32_t tid = threadIdx.x;
	const uint32_t *co = cand_off + (size_t) q * (npr + 1);
	const uint32_t *lco = loc_cand_off + (size_t) q * (npr + 1);
	const uint32_t gtotal = co[npr];
	const uint32_t nraw = ecount[q];
	const int	dim = ix.dim;

	if (nraw > ecap)
	{
		if (tid == 0)
		{
			const uint32_t sl = atomicAdd(&flags[0], 1u);

			if (over_q && sl < S16_OVER_CAP)
				over_q[sl] = q;
			rec_counts[q] = 0;
		}
		return;					/* uniform */
	}
	NDB_PHASE(16);
	const uint2 *rec = erec + (size_t) q * ecap;	/* a few hundred records, read from L2 as often as needed */
	const float *ubq = eub ? eub + (size_t) q * ecap : nullptr;
	float		thrE = qthr[q].x;
	const float e = qthr[q].y;

	if (nraw >= k)
	{
		uint32_t	T, m_less, kk0, cnt_eq;
		/* only finite values stand for a candidate whose distance is known to within E (a NaN or an infinity is
		 * what a row or a product beyond fp32 leaves behind: emitted, never counted) */
		auto		ld = [&](uint32_t i, uint32_t &bits) -> bool {
			bits = ubq ? __float_as_uint(ubq[i]) : rec[i].y;
			return (bits & 0x7FFFFFFFu) < 0x7F800000u;
		};

		block_radix_select(ld, nraw, k, s.hist, s.sh, T, m_less, kk0, cnt_eq);
		if (kk0 >= k)
		{
			/* T = order key of the k-th smallest a: back to the float */
			const uint32_t tb = (T & 0x80000000u) ? (T & 0x7FFFFFFFu) : ~T;

			if (ubq && R == R_IVF_COS)
				thrE = fminf(thrE, s16c_cos_t_from_ub(__uint_as_float(tb), dim));
			else if (ubq && R == R_IVF_IP && qev)
				thrE = fminf(thrE, s16c_ip_t_from_ub(__uint_as_float(tb), qev[q]));
			else if (ubq)
				thrE = fminf(thrE, s16_up(s16_up(fmaxf(__uint_as_float(tb), 0.0f)) * (1.0f + 2.5f * ndb_s16_refslack(dim))));
			else
				thrE = fminf(thrE, s16_thr_from_a<R>(__uint_as_float(tb), e, dim));
		}
	}
	NDB_PHASE(17);
	if (tid == 0)
		s.sh[0] = 0;
	__syncthreads();
	for (uint32_t i = tid; i < nraw; i += blockDim.x)
	{
		const uint2 r = rec[i];

		if (!(__uint_as_float(r.y) > thrE))
		{
			const uint32_t slot = atomicAdd(&s.sh[0], 1u);

			if (slot < surv_cap)
				s.e_pos[slot] = r.x;
		}
	}
	__syncthreads();
	const uint32_t ns = s.sh[0];

	__syncthreads();
	if (ns > surv_cap)
	{
		if (tid == 0)
		{
			const uint32_t sl = atomicAdd(&flags[0], 1u);

			if (over_q && sl < S16_OVER_CAP)
				over_q[sl] = q;
			rec_counts[q] = 0;
		}
		return;
	}
	if (tid == 0)
		rec_counts[q] = ns;
	NDB_PHASE(18);
	/* the reference's arithmetic for every survivor: one lane per candidate */
	/* ... their rows streamed through LDS (first wave): up to 16 with 4 lanes copying per row and the deep ring, up to 64
	 * in one pass when the ring has room for 64-row slots (small batches: LDS is free); otherwise every lane loads the row
	 * it sums, below */
	const bool	st1 = STAGED && stage_nbuf > 0 && ns <= 16, st4 = STAGED && stage_nbuf4 > 0 && !st1;

	if (st1 || st4)
	{
		const uint32_t per = st1 ? 16u : 64u;

		if (tid < 64)
			for (uint32_t j0 = 0; j0 < ns; j0 += per)		/* uniform */
			{
				const uint32_t j = j0 + tid;
				const bool	mine = tid < per && j < ns;
				uint32_t	i = 0, p = 0;
				int			L = 0;
				size_t		row = 0;

				if (mine)
				{
					i = s.e_pos[j];
					p = find_probe(lco, npr, i);
					L = probes[(size_t) q * npr + p];
					row = (size_t) ix.loc_off[L] + (i - lco[p]);
				}
				/* (lane 0 always has a row here: the others' copies read it again) */
				const uint32_t r0lo = (uint32_t) __shfl((int) (uint32_t) row, 0, 64), r0hi = (uint32_t) __shfl((int) (uint32_t) ((uint64_t) row >> 32), 0, 64);
				const size_t rsrc = mine ? row : (size_t) (((uint64_t) r0hi << 32) | r0lo);
				const unsigned char *rp = (const unsigned char *) ix.vecs + rsrc * (size_t) dim * (H16 != 0 ? 2 : 4);
				float		v = 0.0f;

				if constexpr (STAGED)
					v = st1 ? s16_exact_staged<R, H16, 1>(queries + (size_t) q * dim, rp, dim, stage_ring, stage_nbuf)
						: s16_exact_staged<R, H16, 4>(queries + (size_t) q * dim, rp, dim, stage_ring, stage_nbuf4);

				if (mine)
				{
					s.e_bits[j] = __float_as_uint(v);
					s.e_id[j] = ix.tids[row];
					s.e_pos[j] = co[p] + ix.own_lo[L] + (i - lco[p]);
				}
			}
	}
	else
		for (uint32_t j = tid; j < ns; j += blockDim.x)
		{
			const uint32_t i = s.e_pos[j];
			const uint32_t p = find_probe(lco, npr, i);
			const int	L = probes[(size_t) q * npr + p];
			const size_t row = (size_t) ix.loc_off[L] + (i - lco[p]);
			float		v;

			if constexpr (H16 != 0)
				v = scr_exact_h<R, H16 == 1>(queries + (size_t) q * dim, (const uint16_t *) ix.vecs + row * (size_t) dim, dim);
			else
				v = scr_exact<R>(queries + (size_t) q * dim, ix.vecs + row * (size_t) dim, dim);
			s.e_bits[j] = __float_as_uint(v);
			s.e_id[j] = ix.tids[row];
			s.e_pos[j] = co[p] + ix.own_lo[L] + (i - lco[p]);
		}
	__syncthreads();
	NDB_PHASE(19);
	uint32_t	kk;
	const uint32_t npad = next_pow2(ns > 0 ? ns : 1);
	const uint32_t cut = block_sort_cut(s.e_bits, s.e_pos, ns, npad, k, partial ? (uint64_t) ns : (uint64_t) gtotal, s.fs, kk);

	NDB_PHASE(20);

	if (partial)
	{
		for (uint32_t j = tid; j < cut; j += blockDim.x)
		{
			const uint32_t en = s.fs.perm[j];
			ndbhip_cand c;

			c.key = s.e_bits[en];
			c.pos = s.e_pos[en];
			c.tid = s.e_id[en];
			out_cand[(size_t) q * (3 * k) + j] = c;
		}
		if (tid == 0)
		{
			out_ncand[q] = (int) cut;
			out_total[q] = (int64_t) gtotal;
		}
		return;
	}
	block_replay_emit(s.e_bits, s.e_id, cut, kk, s.fs, out_tids + (size_t) q * k, out_dist + (size_t) q * k,
					  out_count + q);
	NDB_PHASE(21);
}

/*
 * Build: the list of every row from the second sweep's records.  One candidate: that centroid (the bound proved
 * every other one farther in the reference's arithmetic).  Several: the reference's own loop (ivf_am.c:905-935:
 * sqrtf of the sequential sum, minDist from FLT_MAX, strict <, so the first minimum in centroid order wins) over
 * the candidates in centroid order.  More than S16_ASSIGN_SLOTS (duplicated centroids, rows beyond fp32): the row
 * goes on the overflow list, which the host sends through the exact assignment.
 */
template <bool SQRT>		/* true: the insert rule (sqrtf of the sum, ivf_am.c:905-935); false: kmeans_assign's squared distance (:2157-2180, 2255-2269) */
__global__ void __launch_bounds__(256)
k_s16_assign_resolve(const float *__restrict__ rows, int64_t nrows, int dim, const float *__restrict__ cents,
					 const unsigned int *__restrict__ acnt, const uint2 *__restrict__ arec,
					 int *__restrict__ out_list, int64_t row_base, unsigned int *__restrict__ over_n,
					 int64_t *__restrict__ over_rows, unsigned long long *__restrict__ multi_n,
					 const uint32_t *__restrict__ rowmin = nullptr /* MODE 4's records: the rows' FINAL minima (keys), ... */ ,
					 const float *__restrict__ rn2 = nullptr /* ... their norms ... */ ,
					 const float2 *__restrict__ aux = nullptr /* ... and [0].x = the largest centroid norm: MODE 2's test is applied here */ )
{
	const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;

	if (r >= nrows)
		return;
	uint32_t	n = acnt[r];
	uint2		rec[S16_ASSIGN_SLOTS];

	if (n >= 1 && n <= S16_ASSIGN_SLOTS)
	{
		uint32_t	nv = 0;
		float		thr = __uint_as_float(0x7F800000u);

		if (rowmin)
		{
			const uint32_t gmin = rowmin[r];
			const uint32_t gb = (gmin & 0x80000000u) ? (gmin & 0x7FFFFFFFu) : ~gmin;

			if (gmin != 0xFFFFFFFFu)
				thr = s16_thr_from_a<R_IVF_L2>(__uint_as_float(gb), s16_e<R_IVF_L2>(dim, rn2[r], aux[0].x, false), dim);
		}
		for (uint32_t j = 0; j < n; j++)
		{
			const uint2 e = arec[(size_t) r * S16_ASSIGN_SLOTS + j];

			if (!rowmin || !(__uint_as_float(e.y) > thr))
				rec[nv++] = e;
		}
		n = nv;
	}
	if (n == 1)
	{
		out_list[r] = (int) rec[0].x;
		return;
	}
	if (n == 0 || n > S16_ASSIGN_SLOTS)
	{
		/* n == 0 cannot happen for a finite row (its minimum is within its own threshold); treated like overflow */
		out_list[r] = -1;
		over_rows[atomicAdd(over_n, 1u)] = row_base + r;
		if (n == 0)
			atomicAdd(over_n + 1, 1u);
		return;
	}
	int			cand[S16_ASSIGN_SLOTS];

	for (uint32_t j = 0; j < n; j++)
	{
		/* kept in centroid order (insertion: a handful of entries) */
		const int	c = (int) rec[j].x;
		int			p = (int) j;

		while (p > 0 && cand[p - 1] > c)
		{
			cand[p] = cand[p - 1];
			p--;
		}
		cand[p] = c;
	}
	const float *x = rows + (size_t) r * dim;
	float		best = 0.0f;
	int			bi = -1;

	for (uint32_t j = 0; j < n; j++)
	{
		const float *cv = cents + (size_t) cand[j] * dim;
		float		dv;

		if ((dim & 3) == 0)
			dv = scr_exact<SQRT ? R_IVF_L2 : R_IVF_L2SQ>(x, cv, dim);	/* 16-byte pieces, 32 loads in flight, then the sequential chain */
		else
		{
			Acc<SQRT ? R_IVF_L2 : R_IVF_L2SQ> acc;

			for (int d = 0; d < dim; d++)
				acc.step(x[d], cv[d]);
			dv = acc.fin();
		}
		/* minDist starts at FLT_MAX and the test is strict: the first minimum wins; a NaN distance never does,
		 * and a row whose every distance is NaN or +inf keeps the initial list 0 */
		if (dv < (bi < 0 ? __uint_as_float(0x7F7FFFFFu) : best))
		{
			best = dv;
			bi = cand[j];
		}
	}
	if (bi < 0)
	{
		out_list[r] = -1;
		over_rows[atomicAdd(over_n, 1u)] = row_base + r;
		return;
	}
	out_list[r] = bi;
	if (multi_n)
		atomicAdd(multi_n, 1ull);
}

/* The instruction the error model of ndbhip_common.h (4) is about, in isolation (ndbhip_mfma_probe): one wave per
 * tile, D = C + chain x (A.B); A [nt][32][16], B [nt][16][32] fp16 bits, C / D [nt][32][32]. */
__global__ __launch_bounds__(64) void
k_s16_mfma_probe(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B, const float *__restrict__ C,
				 float *__restrict__ D, int chain)
{
	const size_t t = blockIdx.x;
	const int	lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
	ndb_h8		a, b;
	ndb_f16acc	acc;

#pragma unroll
	for (int e = 0; e < 8; e++)
	{
		a[e] = __builtin_bit_cast(_Float16, A[(t * 32 + i) * 16 + kh * 8 + e]);
		b[e] = __builtin_bit_cast(_Float16, B[(t * 16 + kh * 8 + e) * 32 + i]);
	}
#pragma unroll
	for (int r = 0; r < 16; r++)
		acc[r] = C[(t * 32 + ((r & 3) + 8 * (r >> 2) + 4 * kh)) * 32 + i];
	for (int c = 0; c < chain; c++)
		acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 16; r++)
		D[(t * 32 + ((r & 3) + 8 * (r >> 2) + 4 * kh)) * 32 + i] = acc[r];
}

/* the same for the fp32 instruction of the centred sweep's pass 0 (ndbhip_mfma_probe_f32): A [nt][32][2], B [nt][2][32] */
__global__ __launch_bounds__(64) void
k_s16_mfma_probe_f32(const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ C,
					 float *__restrict__ D)
{
	const size_t t = blockIdx.x;
	const int	lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
	ndb_f16acc	acc;

#pragma unroll
	for (int r = 0; r < 16; r++)
		acc[r] = C[(t * 32 + ((r & 3) + 8 * (r >> 2) + 4 * kh)) * 32 + i];
	acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(t * 32 + i) * 2 + kh], B[(t * 2 + kh) * 32 + i], acc, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 16; r++)
		D[(t * 32 + ((r & 3) + 8 * (r >> 2) + 4 * kh)) * 32 + i] = acc[r];
}

#endif							/* NDBHIP_SCREEN16_H */
