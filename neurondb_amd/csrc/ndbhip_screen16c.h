/*
 * ndbhip_screen16c.h — the CENTRED one-plane sweep of the screened list scan (part of ndbhip.hip's translation unit;
 * L2, float4 rows).  Same job and same surroundings as k_s16_sweep (ndbhip_screen16.h: seeds, pair tables, items,
 * finalize — ids, ranks and float4 bits of ivfCollectCandidates, src/index/ivf_am.c:1722-1909, unchanged), other
 * operands:
 *
 *   k_s16c_row_prep  once per version of the mirror: every row MINUS THE CENTRE OF ITS BUCKET (the list's centroid,
 *                    or the sample row a sublist was grouped around), |x - c|^2, scale exponent, ONE fp16 plane —
 *                    2 bytes per element, half of what the two-plane sweep streams
 *   k_s16c_qcprep    per batch: for every surviving (query, bucket) pair q - c, its norm, exponent and fp16 plane,
 *                    stored in pair order, so a tile's 128 query rows are 128 consecutive rows
 *   k_s16c_sweep     tiles of 128 rows x 128 (or 32) pairs, one v_mfma_f32_32x32x16_f16 per 16 dimensions and
 *                    32 x 32 block, one accumulator chain; a = |q - c|^2 + |x - c|^2 - 2 (q - c).(x - c) ~ |q - x|^2
 *                    within E = c_E (|q - c|^2 + |x - c|^2) of the real value (ndbhip_common.h (8)): the error
 *                    scales with the distances to the centre, not with the vectors' norms, which is what lets
 *                    one 11-bit plane decide inside a cluster.  An element is left out when a - E > T; an emitted
 *                    one leaves (position, a - E) and a + E.
 *
 * The thresholds (qthr[q].x) of this path are T = thr^2 (1 + m) WITHOUT an error term: E belongs to the element.
 */
#ifndef NDBHIP_SCREEN16C_H
#define NDBHIP_SCREEN16C_H

#define S16C_CH 64				/* dimensions per staged chunk (four MFMA k-steps) */

/* byte offset, inside a chunk's 4 KiB image of a 32-row block, of the 16-byte unit (row rr, logical slot t = 2 kstep + khalf) */
__device__ __forceinline__ uint32_t
s16c_unit(int rr, int t)
{
	return (uint32_t) (t >> 1) * 1024u + (uint32_t) (((t & 1) << 5) + rr) * 16u;
}

/* T from a float4 reference distance that bounds the k-th distance from above / from the k-th smallest upper
 * bound `ub` of distinct candidates (ndbhip_common.h (7), (8)) */
__device__ __forceinline__ float
s16c_t_from_ref(float thr, int dim)
{
	return s16_up(s16_up(thr * thr) * (1.0f + ndb_s16_refslack(dim)));
}
__device__ __forceinline__ float
s16c_t_from_ub(float ub, int dim)
{
	return s16_up(s16_up(fmaxf(ub, 0.0f)) * (1.0f + 2.5f * ndb_s16_refslack(dim)));
}

/*
 * One wave per plane row.  planes[(blk * nchunk + c) * 4096 + image]: the 32 rows of block blk for the 64-dimension
 * chunk c in FRAGMENT-MAJOR order (r5): the 16-byte unit of row r, logical slot t = elements 8 t .. 8 t + 7 of the
 * chunk (t = 2 kstep + khalf), sits at s16c_unit(r, t) = kstep * 1024 + (32 khalf + r) * 16 — lane (r, khalf) of a
 * wave finds its operand of k-step `kstep` at image + 1024 kstep + 16 lane, so that a wave reads a k-step's 1 KiB with
 * ONE fully coalesced 16-byte-per-lane load straight into the registers the matrix instruction takes
 * (k_s16c_wsweep, ndbhip_screen16w.h), and the LDS-staged sweeps (a straight 4 KiB DMA, then ds_read_b128 at
 * 1024 kstep + 16 lane: consecutive lanes, consecutive banks) need no swizzle.  rn2[padded row] = |x - c|^2 as computed (NaN: not a finite fp32,
 * the row's elements are always emitted), rexp[..] its scale exponent, pposof[..] the row's index in its list.
 */
__global__ __launch_bounds__(256) void
k_s16c_row_prep(const float *__restrict__ vecs, int64_t nrows, int dim, int dimp, const int64_t *__restrict__ loc_off,
				const uint32_t *__restrict__ blk_off, int nb, const float *__restrict__ cents /* [nb][dim], or ... */,
				const float *const *__restrict__ cptr /* ... a pointer per bucket */, unsigned char *__restrict__ planes,
				float *__restrict__ rn2, int16_t *__restrict__ rexp, const int64_t *__restrict__ perm,
				const int64_t *__restrict__ prow_off /* [nb + 1] first PADDED plane row (32 x first block) of every bucket */,
				const uint32_t *__restrict__ pos_of /* dense plane row -> index in its list, or NULL (= place in the bucket) */,
				uint32_t *__restrict__ pposof /* padded plane row -> index in its list */ )
{
	const int	lane = threadIdx.x & 63;
	const int64_t prow = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);

	if (prow >= nrows)
		return;
	const int64_t row = perm ? perm[prow] : prow;
	int			lo = 0, hi = nb;

	while (hi - lo > 1)
	{
		const int	mid = (lo + hi) >> 1;

		if (loc_off[mid] <= prow)
			lo = mid;
		else
			hi = mid;
	}
	while (lo + 1 < nb && loc_off[lo + 1] <= prow)
		lo++;
	const float *x = vecs + (size_t) row * dim;
	const float *c = cptr ? cptr[lo] : cents + (size_t) lo * dim;
	double		s = 0.0;

	for (int i = lane; i < dim; i += 64)
	{
		const float d = x[i] - c[i];

		s += (double) d * (double) d;
	}
	s = wave_sum_f64(s);
	const bool	ok = s <= 3.0e38;
	const int	e = ok ? s16_exponent(s) : 0;

	const uint32_t pos = (uint32_t) (prow - loc_off[lo]);

	if (lane == 0)
	{
		/* norms, exponents and list positions are indexed by padded plane row: a bucket's rows stay where they are
		 * when rows are added behind them */
		const int64_t pp = prow_off[lo] + pos;

		rn2[pp] = ok ? (float) s : __uint_as_float(0x7FC00000u);
		rexp[pp] = (int16_t) e;
		pposof[pp] = pos_of ? pos_of[prow] : pos;
	}
	const size_t blk = (size_t) blk_off[lo] + (pos >> 5);
	const int	rr = (int) (pos & 31u);
	const int	nchunk = dimp / S16C_CH;
	unsigned char *img = planes + blk * (size_t) nchunk * 4096;

	for (int p = lane; p < dimp / 2; p += 64)
	{
		const int	i = 2 * p, ch = i >> 6, j = (i & 63) >> 1;
		_Float16	h0 = 0, h1 = 0;

		if (ok && i < dim)
			h0 = (_Float16) (float) ldexp((double) (x[i] - c[i]), 14 - e);
		if (ok && i + 1 < dim)
			h1 = (_Float16) (float) ldexp((double) (x[i + 1] - c[i + 1]), 14 - e);
		ndb_h2		h;

		h.x = h0; h.y = h1;
		*reinterpret_cast<ndb_h2 *>(img + (size_t) ch * 4096 + s16c_unit(rr, j >> 2) + 4 * (j & 3)) = h;
	}
}

/* a row inserted after the planes were laid out: where it sits in the mirror and where it goes in the planes */
struct S16CApp
{
	int64_t		row;			/* mirror row */
	int64_t		pp;				/* padded plane row (in the spare blocks of its bucket) */
	uint32_t	bucket;
	uint32_t	list;
	uint32_t	pos;			/* index in its list */
	uint32_t	pad;
};

/* k_s16c_row_prep for those rows (one wave each), plus what the bounds need: the bucket's and the list's radius grow
 * by atomicMax (float bits of |x - c| rounded up; +inf for a row that is not a finite fp32) */
__global__ __launch_bounds__(256) void
k_s16c_row_append(const float *__restrict__ vecs, int dim, int dimp, const S16CApp *__restrict__ recs, uint32_t n,
				  const float *__restrict__ cents, const float *const *__restrict__ cptr, unsigned char *__restrict__ planes,
				  float *__restrict__ rn2, int16_t *__restrict__ rexp, uint32_t *__restrict__ pposof,
				  uint32_t *__restrict__ sub_rad /* per bucket, or NULL */, uint32_t *__restrict__ lrad /* per list */ )
{
	const int	lane = threadIdx.x & 63;
	const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);

	if (i >= n)
		return;
	const S16CApp a = recs[i];
	const float *x = vecs + (size_t) a.row * dim;
	const float *c = cptr ? cptr[a.bucket] : cents + (size_t) a.bucket * dim;
	double		s = 0.0;

	for (int j = lane; j < dim; j += 64)
	{
		const float d = x[j] - c[j];

		s += (double) d * (double) d;
	}
	s = wave_sum_f64(s);
	const bool	ok = s <= 3.0e38;
	const int	e = ok ? s16_exponent(s) : 0;

	if (lane == 0)
	{
		const double r = __builtin_sqrt(s) * (1.0 + 9.5367431640625e-7);
		const uint32_t bits = (r <= 3.0e38) ? __float_as_uint(__double2float_ru(r)) : 0x7F800000u;

		rn2[a.pp] = ok ? (float) s : __uint_as_float(0x7FC00000u);
		rexp[a.pp] = (int16_t) e;
		pposof[a.pp] = a.pos;
		if (sub_rad)
			atomicMax(&sub_rad[a.bucket], bits);
		atomicMax(&lrad[a.list], bits);
	}
	const size_t blk = (size_t) (a.pp >> 5);
	const int	rr = (int) (a.pp & 31);
	const int	nchunk = dimp / S16C_CH;
	unsigned char *img = planes + blk * (size_t) nchunk * 4096;

	for (int p = lane; p < dimp / 2; p += 64)
	{
		const int	k2 = 2 * p, ch = k2 >> 6, j = (k2 & 63) >> 1;
		_Float16	h0 = 0, h1 = 0;

		if (ok && k2 < dim)
			h0 = (_Float16) (float) ldexp((double) (x[k2] - c[k2]), 14 - e);
		if (ok && k2 + 1 < dim)
			h1 = (_Float16) (float) ldexp((double) (x[k2 + 1] - c[k2 + 1]), 14 - e);
		ndb_h2		h;

		h.x = h0; h.y = h1;
		*reinterpret_cast<ndb_h2 *>(img + (size_t) ch * 4096 + s16c_unit(rr, j >> 2) + 4 * (j & 3)) = h;
	}
}

/* dst[idx[i]] = val[i] */
__global__ void
k_s16c_set_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ val, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;

	if (i < n)
		dst[idx[i]] = val[i];
}

/*
 * k_s16c_qcprep's work for G consecutive pair records j .. j + G - 1 of one wave (dims in multiples of 4, up to 256 NT):
 * 16 bytes per lane and load, the loads of all G pairs in flight together and their fp64 sums folded side by side — one
 * pair after the other was two memory round trips and twelve cross-lane steps per pair, 0.47 ms of a 4096-query batch on
 * 10 M rows —, every q_i - c_i kept in registers for the plane.  `lo` = the bucket of pair j - 1 on entry, of the last
 * pair on return.
 */
template <int NT, int G>
__device__ __forceinline__ void
s16c_qc_group(const float *__restrict__ queries, int dim, int dimp, const PairRec *__restrict__ pairs,
			  const uint32_t *__restrict__ pair_off, int nb, const float *__restrict__ cents,
			  const float *const *__restrict__ cptr, _Float16 *__restrict__ qcplanes, size_t chunk_plane,
			  float *__restrict__ qcn2, int *__restrict__ qcexp, uint32_t *__restrict__ pqid, uint32_t *__restrict__ pla,
			  uint32_t *__restrict__ pnrow, const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t j, uint32_t &lo,
			  int lane)
{
	PairRec		pr[G];
	const float *q[G], *c[G];
	float4		dv[G][NT];
	double		s[G];

#pragma unroll
	for (int g = 0; g < G; g++)
	{
		while (lo + 1 < (uint32_t) nb && pair_off[lo + 1] <= j + (uint32_t) g)
			lo++;
		pr[g] = pairs[j + g];
		q[g] = queries + (size_t) pr[g].q * dim;
		c[g] = cptr ? cptr[lo] : cents + (size_t) lo * dim;
	}
#pragma unroll
	for (int g = 0; g < G; g++)
#pragma unroll
		for (int t = 0; t < NT; t++)
		{
			const int	i = t * 256 + lane * 4;

			dv[g][t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (i < dim)
			{
				const float4 qv = *reinterpret_cast<const float4 *>(q[g] + i), cv = *reinterpret_cast<const float4 *>(c[g] + i);

				dv[g][t] = make_float4(qv.x - cv.x, qv.y - cv.y, qv.z - cv.z, qv.w - cv.w);
			}
		}
#pragma unroll
	for (int g = 0; g < G; g++)
	{
		s[g] = 0.0;
#pragma unroll
		for (int t = 0; t < NT; t++)
			s[g] += (double) dv[g][t].x * (double) dv[g][t].x + (double) dv[g][t].y * (double) dv[g][t].y +
				(double) dv[g][t].z * (double) dv[g][t].z + (double) dv[g][t].w * (double) dv[g][t].w;
	}
#pragma unroll
	for (int g = 0; g < G; g++)
		s[g] = wave_sum_f64(s[g]);
#pragma unroll
	for (int g = 0; g < G; g++)
	{
		const uint32_t jj = j + (uint32_t) g;
		const bool	ok = s[g] <= 3.0e38;
		const int	e = ok ? s16_exponent(s[g]) : 0;

		if (lane == 0)
		{
			const uint32_t *lq = loc_cand_off + (size_t) pr[g].q * (npr + 1);

			qcn2[jj] = ok ? (float) s[g] : __uint_as_float(0x7FC00000u);
			qcexp[jj] = e;
			pqid[jj] = pr[g].q;
			pla[jj] = lq[pr[g].p];
			pnrow[jj] = lq[pr[g].p + 1] - lq[pr[g].p];
		}
		_Float16   *out = qcplanes + (chunk_plane ? (size_t) jj * 64 : (size_t) jj * dimp);
		const float sc = ldexpf(1.0f, 14 - e);

#pragma unroll
		for (int t = 0; t < NT; t++)
		{
			const int	i = t * 256 + lane * 4;

			if (i >= dimp)
				continue;
			ndb_h2		h01, h23;

			h01.x = ok ? (_Float16) (dv[g][t].x * sc) : (_Float16) 0;
			h01.y = ok ? (_Float16) (dv[g][t].y * sc) : (_Float16) 0;
			h23.x = ok ? (_Float16) (dv[g][t].z * sc) : (_Float16) 0;
			h23.y = ok ? (_Float16) (dv[g][t].w * sc) : (_Float16) 0;
			_Float16   *o = chunk_plane ? out + (size_t) (i >> 6) * chunk_plane + (i & 63) : out + i;
			ndb_h2		pk[2] = {h01, h23};

			*reinterpret_cast<uint2 *>(o) = *reinterpret_cast<const uint2 *>(pk);
		}
	}
}

/*
 * One wave per (query, bucket) pair record, in the pair tables' order (slot j = pair_off[bucket] + i): the plane of
 * q - c in natural element order (the sweep's DMA applies the LDS swizzle), qcn2[j] = |q - c|^2 as computed (NaN:
 * not a finite fp32 — every element of the pair is emitted), qcexp[j], and the pair's query / first candidate
 * position / visible rows (pqid, pla, pnrow).  Persistent grid.
 */
__global__ __launch_bounds__(256) void
k_s16c_qcprep(const float *__restrict__ queries, int dim, int dimp, const PairRec *__restrict__ pairs,
			  const uint32_t *__restrict__ pair_off, int nb, const float *__restrict__ cents,
			  const float *const *__restrict__ cptr, _Float16 *__restrict__ qcplanes,
			  size_t chunk_plane /* 0: [pair][dimp]; else the dense sweep's layout, [64-dim chunk][pair][64] with this many halfs per chunk plane */,
			  float *__restrict__ qcn2,
			  int *__restrict__ qcexp, uint32_t *__restrict__ pqid, uint32_t *__restrict__ pla, uint32_t *__restrict__ pnrow,
			  const uint32_t *__restrict__ loc_cand_off, int npr,
			  uint32_t cap, unsigned int *__restrict__ flags /* [0]++ when the pairs exceed cap */,
			  unsigned int *__restrict__ dens /* statistics or NULL: [0] = pairs, [1] = buckets with pairs */,
			  const uint32_t *__restrict__ cnt)
{
	const int	lane = threadIdx.x & 63;
	const uint32_t total = min(pair_off[nb], cap);
	const uint32_t nw = gridDim.x * 4;

	if (blockIdx.x == 0)
	{
		if (threadIdx.x == 0 && pair_off[nb] > cap)
			atomicAdd(flags, 1u);
		if (dens)
		{
			uint32_t	n = 0;

			for (int b = threadIdx.x; b < nb; b += 256)
				n += cnt[b] != 0 ? 1u : 0u;
			for (int off = 32; off > 0; off >>= 1)
				n += (uint32_t) __shfl_xor((int) n, off, 64);
			if (lane == 0)
				atomicAdd(&dens[1], n);
			if (threadIdx.x == 0)
				dens[0] = pair_off[nb];
		}
	}

	/* every wave takes a contiguous run of the pair records: one bisection for the run's first bucket, then the buckets are
	 * walked along with the records (a bisection per record — ten dependent loads — was most of this kernel's time) */
	const uint32_t per = (total + nw - 1) / nw;
	const uint32_t j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * per, j1 = min(total, j0 + per);
	uint32_t	lo = 0;

	if (j0 < j1)
	{
		uint32_t	hi = (uint32_t) nb;

		while (hi - lo > 1)
		{
			const uint32_t mid = (lo + hi) >> 1;

			if (pair_off[mid] <= j0)
				lo = mid;
			else
				hi = mid;
		}
	}
	uint32_t	jv = j0;

	if ((dim & 3) == 0 && dim <= 2048)
	{
#define S16C_QC_ARGS queries, dim, dimp, pairs, pair_off, nb, cents, cptr, qcplanes, chunk_plane, qcn2, qcexp, pqid, pla, pnrow, loc_cand_off, npr
		if (dim <= 1024)
		{
			for (; jv + 4 <= j1; jv += 4)
				s16c_qc_group<4, 4>(S16C_QC_ARGS, jv, lo, lane);
			for (; jv < j1; jv++)
				s16c_qc_group<4, 1>(S16C_QC_ARGS, jv, lo, lane);
		}
		else
		{
			for (; jv + 2 <= j1; jv += 2)
				s16c_qc_group<8, 2>(S16C_QC_ARGS, jv, lo, lane);
			for (; jv < j1; jv++)
				s16c_qc_group<8, 1>(S16C_QC_ARGS, jv, lo, lane);
		}
#undef S16C_QC_ARGS
	}
	for (uint32_t j = jv; j < j1; j++)
	{
		while (lo + 1 < (uint32_t) nb && pair_off[lo + 1] <= j)
			lo++;
		const PairRec pr = pairs[j];
		const float *q = queries + (size_t) pr.q * dim;
		const float *c = cptr ? cptr[lo] : cents + (size_t) lo * dim;
		double		s = 0.0;
		for (int i = lane; i < dim; i += 64)
		{
			const float d = q[i] - c[i];

			s += (double) d * (double) d;
		}
		s = wave_sum_f64(s);
		const bool	ok = s <= 3.0e38;
		const int	e = ok ? s16_exponent(s) : 0;

		if (lane == 0)
		{
			const uint32_t *lq = loc_cand_off + (size_t) pr.q * (npr + 1);

			qcn2[j] = ok ? (float) s : __uint_as_float(0x7FC00000u);
			qcexp[j] = e;
			/* what the sweep needs of the pair besides its plane, in the same order: the query, the first candidate
			 * position of this (query, probe) and how many rows of the list it may see */
			pqid[j] = pr.q;
			pla[j] = lq[pr.p];
			pnrow[j] = lq[pr.p + 1] - lq[pr.p];
		}
		ndb_h2	   *out = reinterpret_cast<ndb_h2 *>(qcplanes + (chunk_plane ? (size_t) j * 64 : (size_t) j * dimp));
		/* (q - c) 2^(14 - e): the scale is a power of two between 2^-126 and 2^127 for every exponent an fp32 vector's norm
		 * can have, the product exact (below 2^-126 it is flushed: inside the 2^-25 the error model allows an element) */
		const float sc = ldexpf(1.0f, 14 - e);

		for (int p = lane; p < dimp / 2; p += 64)
		{
			const int	i = 2 * p;
			_Float16	h0 = 0, h1 = 0;

			if (ok && i < dim)
				h0 = (_Float16) ((q[i] - c[i]) * sc);
			if (ok && i + 1 < dim)
				h1 = (_Float16) ((q[i + 1] - c[i + 1]) * sc);
			ndb_h2		h;

			h.x = h0; h.y = h1;
			if (chunk_plane)
				*reinterpret_cast<ndb_h2 *>(reinterpret_cast<_Float16 *>(out) + (size_t) (i >> 6) * chunk_plane + (i & 63)) = h;
			else
				out[p] = h;
		}
	}
}

/*
 * First thresholds of the centred path.  k_s16_seed / k_s16_seed_sub give `ns` candidates the reference's own
 * sequential arithmetic, one lane per row — 768 dependent steps a lane, 0.15 ms a batch.  A threshold only has to be
 * an UPPER bound of the k-th distance, so here the whole wave sums one row's (q_i - x_i)^2 at a time: every term is
 * the reference's own fl(fl(q_i - x_i)^2) >= 0, and a sum of n non-negative floats in ANY order is within
 * (1 - n u) .. (1 + n u) of the real sum, so D <= S (1 + m), m = 2 (dim + 16) u (ndbhip_common.h (7)); the k-th
 * smallest of those upper bounds over distinct candidates bounds the k-th distance, T = s16c_t_from_ub.
 * Seeds: the first `ns` rows of the sublist whose centre is nearest to the query, among the sublists of its probed
 * lists that hold at least k visible rows (SUB; k_s16_seed_sub's rule); without one, the query's first `ns` candidates.
 * One wave per query.
 */
/*
 * IP (inner product, thresholds in b's domain: ndbhip_screen16.h s16c_ip_*): the wave sums -q_i x_i instead; a sum of dim
 * rounded products in any order is within gamma_(dim + 1) |q||x| <= 1.01 ev of the real -q.x, the reference's own value
 * within ev of it, so v + 2.02 ev bounds the reference's value of the row from above and the k-th smallest of those is a
 * reference bound: T = s16c_ip_t_from_ref.  The nearest sublist is the one with the largest q.c = (|q|^2 + |c|^2 -
 * |q - c|^2) / 2 (cn2_sub / cn2_list: the centres' norms).  H16: the mirror's rows are fp16 (decoded like the reference
 * decodes them).
 */
#define S16C_SEED_THREADS 256	/* 4 waves a query: one wave alone was 0.33 ms of dependent loads for 256 queries on 10M rows */
template <bool SUB, bool IP = false, int H16 = 0 /* 0: float4 rows; 1: fp16 decoded like the reference (quirk Q20); 2: fp16 without subnormals (the plain conversion is the reference's) */,
		  bool PL = false /* the seeds' bounds from the sweep's own fp16 planes (SUB, L2): see "plane seeds" below */>
__global__ __launch_bounds__(S16C_SEED_THREADS) void
k_s16c_seed(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
			const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t k, uint32_t ns,
			const uint32_t *__restrict__ sub_first, const int *__restrict__ sub_gidx, const uint32_t *__restrict__ sub_len,
			const int64_t *__restrict__ prow_off /* first padded plane row of every sublist */,
			const uint32_t *__restrict__ pposof /* padded plane row -> index in its list */,
			const float *__restrict__ subdist, uint32_t sstride, const float *__restrict__ pdist,
			const float *__restrict__ cdist, uint32_t cstride, float2 *__restrict__ qthr,
			const float *__restrict__ qn2 = nullptr, const uint32_t *__restrict__ m2_bits = nullptr /* IP: |q|^2, M^2 */,
			const float *__restrict__ cn2_sub = nullptr, const float *__restrict__ cn2_list = nullptr /* IP: |c|^2 */,
			/* PL: the centred planes and what goes with them (k_s16c_row_prep), the buckets' centres, c_E */
			const unsigned char *__restrict__ planes = nullptr, const uint32_t *__restrict__ blk_off = nullptr,
			const float *__restrict__ rn2 = nullptr, const int16_t *__restrict__ rexp = nullptr,
			const float *const *__restrict__ cptr = nullptr, float cE = 0.0f,
			const float *__restrict__ rnx = nullptr /* PL with IP: M^2 - |x|^2 per padded plane row */ )
{
	const uint32_t q = blockIdx.x;
	const int	lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	constexpr int NW = S16C_SEED_THREADS / 64;
	const uint32_t *lco = loc_cand_off + (size_t) q * (npr + 1);
	const int	dim = ix.dim;
	const float *qq = queries + (size_t) q * dim;
	bool		ok = false;
	int64_t		row = 0;

	if constexpr (SUB)
	{
		float		bd = __uint_as_float(0x7F800000u);
		uint32_t	bs = 0xFFFFFFFFu, bp = 0;
		/* the (probe, sublist) candidates of 64 probes at a time laid end to end and dealt to the lanes (as in
		 * k_sub_pairs: walking the probes one after the other was 32 dependent trips of mostly idle lanes) */
		__shared__ uint32_t s_off[65], s_s0[64], s_bs[NW], s_bp[NW];
		__shared__ float s_pd[64], s_bd[NW];

		for (int p0 = 0; p0 < npr; p0 += 64)
		{
			const int	p = p0 + lane;
			uint32_t	n = 0, s0 = 0;
			float		pd = 0.0f;

			if (p < npr)
			{
				const uint32_t vis = lco[p + 1] - lco[p];
				const int	L = probes[(size_t) q * npr + p];

				if (vis >= k && L >= 0 && L < ix.ncent)
				{
					s0 = sub_first[L];
					n = sub_first[L + 1] - s0;
					pd = cdist ? cdist[(size_t) q * cstride + L] : pdist[(size_t) q * npr + p];
				}
			}
			uint32_t	inc = n;

#pragma unroll
			for (int off = 1; off < 64; off <<= 1)
			{
				const uint32_t v = (uint32_t) __shfl_up((int) inc, off, 64);

				if (lane >= off)
					inc += v;
			}
			const uint32_t T = (uint32_t) __shfl((int) inc, 63, 64);

			if (wv == 0)			/* (every wave holds the same values) */
			{
				s_off[lane] = inc - n;
				s_s0[lane] = s0;
				s_pd[lane] = pd;
				if (lane == 0)
					s_off[64] = T;
			}
			__syncthreads();
			for (uint32_t t = threadIdx.x; t < T; t += S16C_SEED_THREADS)
			{
				int			lo = 0, hi = 64;

				while (hi - lo > 1)
				{
					const int	mid = (lo + hi) >> 1;

					if (s_off[mid] <= t)
						lo = mid;
					else
						hi = mid;
				}
				const uint32_t sx = s_s0[lo] + (t - s_off[lo]);

				if (sub_len[sx] < k)
					continue;
				const int	gi = sub_gidx[sx];
				const float pdl = s_pd[lo];
				float		dd = gi < 0 ? pdl * pdl : fmaxf(subdist[(size_t) q * sstride + gi], 0.0f);	/* both squared */

				if constexpr (IP)
					/* (|q - c|^2 - |c|^2 orders the centres like -q.c does; any choice of seed rows is a valid one) */
					dd -= gi < 0 ? (cn2_list ? cn2_list[probes[(size_t) q * npr + p0 + lo]] : 0.0f) : (cn2_sub ? cn2_sub[gi] : 0.0f);

				/* (a total order, the probe included: a list probed twice — the reference's probe slots beyond nlists all
				 * read list 0, ivf_am.c:1978 — offers the same sublist at the same distance under two probes with different
				 * numbers of visible rows, and every lane of every wave has to end up with the SAME probe: round 6's fuzz
				 * found lanes that disagreed on which seed rows are candidates, and a threshold of 0) */
				if (dd < bd || (dd == bd && (sx < bs || (sx == bs && (uint32_t) (p0 + lo) < bp))))
				{
					bd = dd;
					bs = sx;
					bp = (uint32_t) (p0 + lo);
				}
			}
			__syncthreads();
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
		{
			const float od = __shfl_xor(bd, off, 64);
			const uint32_t os = (uint32_t) __shfl_xor((int) bs, off, 64), op = (uint32_t) __shfl_xor((int) bp, off, 64);

			if (od < bd || (od == bd && (os < bs || (os == bs && op < bp))))
			{
				bd = od;
				bs = os;
				bp = op;
			}
		}
		if (lane == 0)
		{
			s_bd[wv] = bd;
			s_bs[wv] = bs;
			s_bp[wv] = bp;
		}
		__syncthreads();
#pragma unroll
		for (int w = 0; w < NW; w++)
		{
			const float od = s_bd[w];
			const uint32_t os = s_bs[w], op = s_bp[w];

			if (od < bd || (od == bd && (os < bs || (os == bs && op < bp))))
			{
				bd = od;
				bs = os;
				bp = op;
			}
		}
		if (bs != 0xFFFFFFFFu)		/* uniform over the block */
		{
			const uint32_t vis = lco[bp + 1] - lco[bp];

			if ((uint32_t) lane < min(sub_len[bs], ns))
			{
				const uint32_t pos = pposof[prow_off[bs] + lane];

				ok = pos < vis;			/* a candidate of this (query, probe) under the candidate cap */
				row = ix.loc_off[probes[(size_t) q * npr + bp]] + pos;
			}
		}
		if constexpr (PL)
		{
			/*
			 * Plane seeds (r5).  The seed rows are the first rows of the nearest sublist: block 0 of that bucket's planes.
			 * Summing (q_i - x_i)^2 over their float4 rows read 32 x 3 KB a query — 440 MB a batch of 4096, 85 us at C2 —
			 * for a number that only has to BOUND the k-th distance from above; the sweep's own arithmetic gives such a
			 * bound from half the bytes: a = Q2 + X2 - 2 (q - c).(x - c) from the block's fp16 image (one
			 * v_mfma_f32_32x32x16_f16 per 16 dimensions in ONE accumulator chain, the query's q - c as the only member of
			 * the pair operand: the very chain k_s16c_sweep runs), |a - |q - x|^2| <= E = c_E (Q2 + X2) + ABS
			 * (ndbhip_common.h (8)), so a + E bounds the real squared distance from above — what s16c_t_from_ub takes.
			 * q - c, its norm and its exponent are made exactly as k_s16c_qcprep makes them (fp64 sum, 2^(14 - e), round
			 * to nearest fp16).
			 */
			const int	dimp = (dim + 63) & ~63;
			const bool	enough = (uint32_t) __popcll(__ballot(ok)) >= k;		/* (every wave holds the same `ok`) */

			if (bs != 0xFFFFFFFFu && enough && ns <= 32u && dimp <= 2048 && (dim & 3) == 0)		/* uniform over the block */
			{
				__shared__ double s_sum[NW];
				__shared__ __attribute__((aligned(16))) _Float16 s_qc[2048];
				const float *c = cptr[bs];
				const int	i0 = (int) threadIdx.x * 4;
				float4		dv[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
				double		sm = 0.0;

#pragma unroll
				for (int t = 0; t < 2; t++)
				{
					const int	i = i0 + t * 4 * S16C_SEED_THREADS;

					if (i < dim)
					{
						const float4 qv = *reinterpret_cast<const float4 *>(qq + i), cv = *reinterpret_cast<const float4 *>(c + i);

						dv[t] = make_float4(qv.x - cv.x, qv.y - cv.y, qv.z - cv.z, qv.w - cv.w);
						sm += (double) dv[t].x * (double) dv[t].x + (double) dv[t].y * (double) dv[t].y +
							(double) dv[t].z * (double) dv[t].z + (double) dv[t].w * (double) dv[t].w;
					}
				}
				sm = wave_sum_f64(sm);
				if (lane == 0)
					s_sum[wv] = sm;
				__syncthreads();
				double		tot = 0.0;

#pragma unroll
				for (int w = 0; w < NW; w++)
					tot += s_sum[w];
				const bool	qok = tot <= 3.0e38;
				const int	eq = qok ? s16_exponent(tot) : 0;
				const float Q2 = qok ? (float) tot : __uint_as_float(0x7FC00000u);
				const float sc = ldexpf(1.0f, 14 - eq);

#pragma unroll
				for (int t = 0; t < 2; t++)
				{
					const int	i = i0 + t * 4 * S16C_SEED_THREADS;

					if (i < dimp)
					{
						ndb_h2		h01, h23;

						h01.x = qok ? (_Float16) (dv[t].x * sc) : (_Float16) 0;
						h01.y = qok ? (_Float16) (dv[t].y * sc) : (_Float16) 0;
						h23.x = qok ? (_Float16) (dv[t].z * sc) : (_Float16) 0;
						h23.y = qok ? (_Float16) (dv[t].w * sc) : (_Float16) 0;
						ndb_h2		pk[2] = {h01, h23};

						*reinterpret_cast<uint2 *>(s_qc + i) = *reinterpret_cast<const uint2 *>(pk);
					}
				}
				__syncthreads();
				if (wv)
					return;
				/* one wave, one chain: block 0 of the bucket, chunk by chunk, k-step by k-step */
				const int	nchunk = dimp / S16C_CH, r32 = lane & 31, kh = lane >> 5;
				const unsigned char *blk = planes + (size_t) blk_off[bs] * (size_t) nchunk * 4096;
				ndb_f16acc	acc;
				ndb_h8		zero;

#pragma unroll
				for (int i = 0; i < 16; i++)
					acc[i] = 0.0f;
#pragma unroll
				for (int i = 0; i < 8; i++)
					zero[i] = (_Float16) 0;
				for (int ch = 0; ch < nchunk; ch++)
				{
					ndb_h8		bh[4], ah[4];

#pragma unroll
					for (int st = 0; st < 4; st++)
					{
						bh[st] = *reinterpret_cast<const ndb_h8 *>(blk + (size_t) ch * 4096 + st * 1024 + lane * 16);
						ah[st] = r32 == 0 ? *reinterpret_cast<const ndb_h8 *>(s_qc + ch * 64 + st * 16 + kh * 8) : zero;
					}
#pragma unroll
					for (int st = 0; st < 4; st++)
						acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], bh[st], acc, 0, 0, 0);
				}
				/* element (reg 0, lane r < 32) = member 0, row r of the block */
				const int64_t pp = prow_off[bs] + (lane & 31);
				const float X2 = rn2[pp];
				const int	ex = (int) rexp[pp];
				const float t1 = ldexpf(acc[0], eq + ex - 27);
				const float nn = Q2 + X2;
				const float av = nn - t1;
				const float er = s16_up(s16_up(cE * nn) + NDB_S16_ABS);
				/* (inner product: the bound of b = |q - x|^2 + M^2 - |x|^2, the row's constant on top as the sweep's epilogue
				 * puts it there; the threshold converts as k_s16_finalize's k-th upper bound does) */
				const float ub = IP ? fmaxf(s16_up(s16_up(s16_up(av + er)) + rnx[pp] * 1.000001f), 0.0f) : fmaxf(s16_up(s16_up(av + er)), 0.0f);
				const bool	good = ok && lane < 32 && ub == ub && (ub - ub) == 0.0f;
				const uint32_t key = good ? __float_as_uint(ub) : 0xFFFFFFFFu;
				uint32_t	rank = 0;

				for (int j = 0; j < 64; j++)
				{
					const uint32_t kj = (uint32_t) __shfl((int) key, j, 64);

					rank += (kj < key || (kj == key && j < lane)) ? 1u : 0u;
				}
				const unsigned long long pick = __ballot(good && rank == k - 1);
				float		t = __uint_as_float(0x7F800000u);	/* +inf: fewer than k usable seeds */

				if (pick)
				{
					const float ubk = __shfl(ub, __ffsll((long long) pick) - 1, 64);

					t = IP ? s16c_ip_t_from_ub(ubk, s16c_ip_ev(dim, qn2[q], __uint_as_float(*m2_bits))) : s16c_t_from_ub(ubk, dim);
				}
				if (lane == 0)
					qthr[q] = make_float2(t, 0.0f);
				return;
			}
		}
	}
	if (!SUB || (uint32_t) __popcll(__ballot(ok)) < k)
	{
		/* the query's first candidates (probe order) */
		ok = (uint32_t) lane < min(lco[npr], ns);
		row = 0;
		if (ok)
		{
			const uint32_t p = find_probe(lco, npr, (uint32_t) lane);
			const int	L = probes[(size_t) q * npr + p];

			row = ix.loc_off[L] + ((uint32_t) lane - lco[p]);
		}
	}
	const unsigned long long have = __ballot(ok);
	const int	n = have ? 64 - __builtin_clzll(have) : 0;		/* slots 0 .. n - 1 may hold a seed */
	float		v = 0.0f;
	const bool	vec4 = (dim & 3) == 0;

	/* SG seed rows at a time per wave: their loads are in flight together (a row is 3 KB: 12 coalesced loads per lane) */
	constexpr int SG = 8;
	__shared__ float s_v[64];

	for (int j0 = wv * SG; j0 < n; j0 += NW * SG)
	{
		float		part[SG];

#pragma unroll
		for (int u = 0; u < SG; u++)
			part[u] = 0.0f;

#pragma unroll
		for (int u = 0; u < SG; u++)
		{
			const int	j = j0 + u;

			if (j >= n || !((have >> j) & 1ull))
				continue;			/* uniform */
			const uint32_t rlo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) row, j);
			const uint32_t rhi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (row >> 32), j);
			const size_t ro = (size_t) (((uint64_t) rhi << 32) | rlo) * (size_t) dim;
			const float *x = ix.vecs + ro;
			const uint16_t *xh = (const uint16_t *) ix.vecs + ro;
			float		a = 0.0f;

			if (vec4)
				for (int i = lane * 4; i < dim; i += 256)
				{
					float4		xv;
					const float4 qv = *reinterpret_cast<const float4 *>(qq + i);

					if constexpr (H16 == 1)
					{
						const uint2 hb = *reinterpret_cast<const uint2 *>(xh + i);

						xv = make_float4(h2f_ref(hb.x & 0xFFFFu), h2f_ref(hb.x >> 16), h2f_ref(hb.y & 0xFFFFu), h2f_ref(hb.y >> 16));
					}
					else if constexpr (H16 == 2)
					{
						const ndb_h2 h01 = *reinterpret_cast<const ndb_h2 *>(xh + i), h23 = *reinterpret_cast<const ndb_h2 *>(xh + i + 2);

						xv = make_float4((float) h01.x, (float) h01.y, (float) h23.x, (float) h23.y);
					}
					else
						xv = *reinterpret_cast<const float4 *>(x + i);
					if constexpr (IP)
					{
						a -= qv.x * xv.x;
						a -= qv.y * xv.y;
						a -= qv.z * xv.z;
						a -= qv.w * xv.w;
					}
					else
					{
						const float d0 = qv.x - xv.x, d1 = qv.y - xv.y, d2 = qv.z - xv.z, d3 = qv.w - xv.w;

						a += d0 * d0;
						a += d1 * d1;
						a += d2 * d2;
						a += d3 * d3;
					}
				}
			else
				for (int i = lane; i < dim; i += 64)
				{
					const float xv = H16 == 1 ? h2f_ref(xh[i]) : (H16 == 2 ? (float) reinterpret_cast<const _Float16 *>(xh)[i] : x[i]);

					if constexpr (IP)
						a -= qq[i] * xv;
					else
					{
						const float d0 = qq[i] - xv;

						a += d0 * d0;
					}
				}
			part[u] = a;
		}
#pragma unroll
		for (int u = 0; u < SG; u++)
		{
			float		a = part[u];

#pragma unroll
			for (int off = 32; off > 0; off >>= 1)
				a += __shfl_xor(a, off, 64);
			if (lane == j0 + u)
				s_v[lane] = a;
		}
	}
	__syncthreads();
	if (wv)
		return;
	v = lane < n ? s_v[lane] : 0.0f;		/* (slots without a seed were never written: `ok` masks them below) */
	/* upper bound of the real squared distance (IP: of the reference's value); NaN (a row or query beyond fp32) bounds nothing */
	float		ub;

	if constexpr (IP)
		ub = s16_up(v + 2.02f * s16c_ip_ev(dim, qn2[q], __uint_as_float(*m2_bits)));
	else
		ub = s16_up(v * (1.0f + ndb_s16_refslack(dim)));
	const bool	good = ok && ub == ub && (!IP || (ub - ub) == 0.0f);
	/* (L2: ub >= 0, its bits order like the values; IP: the order-preserving key of any float) */
	const uint32_t key = good ? (IP ? ndb_key_from_bits(__float_as_uint(ub)) : __float_as_uint(ub)) : 0xFFFFFFFFu;
	uint32_t	rank = 0;

	for (int j = 0; j < 64; j++)
	{
		const uint32_t kj = (uint32_t) __shfl((int) key, j, 64);

		rank += (kj < key || (kj == key && j < lane)) ? 1u : 0u;
	}
	const unsigned long long pick = __ballot(good && rank == k - 1);
	float		t = __uint_as_float(0x7F800000u);	/* +inf: fewer than k seeds */

	if (pick)
	{
		const float uk = __shfl(ub, __ffsll((long long) pick) - 1, 64);

		t = IP ? s16c_ip_t_from_ref(uk, qn2[q], __uint_as_float(*m2_bits), dim) : s16c_t_from_ub(uk, dim);
	}
	if (lane == 0)
		qthr[q] = make_float2(t, 0.0f);
}

/*
 * Thresholds for tables WITHOUT cluster structure (the i.i.d. table of BASELINE.md: whole lists as buckets, every query
 * probing most of the table).  k_s16c_seed's 32 seed rows are 32 arbitrary rows there — their k-th smallest distance is
 * about the median of all distances — and a query's first tiles, multiplied by 32 blocks at a time before any tightening
 * can reach them, emitted half of their rows: 2 100 records per query for 13 survivors, 2.7 of the sweep's 7.4 ms
 * (profiles/r04_dense_probe.txt).  A threshold taken from n0 of the query's candidates passes 1 / n0-th of the rest as
 * the sweep begins, so: a fixed SAMPLE of the mirror's rows (`ns`, every nrows / ns-th row: k_seed_gather, once per
 * version of the mirror), all queries against all of them as one dense matrix on the matrix cores (s16mat_run: the
 * two-plane sweep's MODE 3, as the centroid scan runs), and per query the k-th smallest UPPER bound a + E of its own
 * candidates among them — a sample row counts for a query that probes its list and sees its position under the candidate
 * cap, exactly what makes a row a seed in k_s16c_seed.
 */
__global__ __launch_bounds__(256) void
k_seed_gather(const float *__restrict__ vecs, int64_t nrows, int dim, const int64_t *__restrict__ loc_off, int ncent,
			  uint32_t ns, float *__restrict__ out, int *__restrict__ slist, uint32_t *__restrict__ spos)
{
	const uint32_t i = blockIdx.x;
	const int64_t row = (int64_t) (((uint64_t) i * (uint64_t) nrows) / ns);
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
	if (threadIdx.x == 0)
	{
		slist[i] = lo;
		spos[i] = (uint32_t) (row - loc_off[lo]);
	}
	for (int j = threadIdx.x; j < dim; j += 256)
		out[(size_t) i * dim + j] = vecs[(size_t) row * dim + j];
}

/* one block per query; amat[q][0 .. ns) = a ~ |q - sample|^2 within E = s16_e(dim, |q|^2, largest sample norm) */
__global__ __launch_bounds__(256) void
k_s16c_seed_sample(const float *__restrict__ amat, uint32_t astride, uint32_t ns, const int *__restrict__ slist,
				   const uint32_t *__restrict__ spos, const int *__restrict__ probes, const uint32_t *__restrict__ loc_cand_off,
				   int npr, uint32_t k, const float *__restrict__ qn2, const uint32_t *__restrict__ xmax_bits, int dim,
				   float2 *__restrict__ qthr)
{
	constexpr int PER = 8;			/* ns <= 2048 */
	__shared__ int s_pl[512];
	__shared__ uint32_t s_vis[512];
	__shared__ unsigned long long s_red[4];
	const uint32_t q = blockIdx.x;
	const int	tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const uint32_t *lco = loc_cand_off + (size_t) q * (npr + 1);

	for (int p = tid; p < npr; p += 256)
	{
		s_pl[p] = probes[(size_t) q * npr + p];
		s_vis[p] = lco[p + 1] - lco[p];
	}
	__syncthreads();
	const float e = s16_e<R_IVF_L2>(dim, qn2[q], __uint_as_float(*xmax_bits), false);
	unsigned long long key[PER];
	int			sl[PER];
	uint32_t	sp[PER];
	bool		ok[PER];

#pragma unroll
	for (int u = 0; u < PER; u++)
	{
		const uint32_t i = (uint32_t) tid + 256u * (uint32_t) u;

		sl[u] = i < ns ? slist[i] : -1;
		sp[u] = i < ns ? spos[i] : 0xFFFFFFFFu;
		ok[u] = false;
	}
	/* (probes outermost: one broadcast read of a probe serves the thread's PER samples; a list can be probed more than
	 * once — ivf_am.c:1978 —: any probe that sees the row will do) */
	for (int p = 0; p < npr; p++)
	{
		const int	L = s_pl[p];
		const uint32_t vis = s_vis[p];

#pragma unroll
		for (int u = 0; u < PER; u++)
			ok[u] = ok[u] || (sl[u] == L && sp[u] < vis);
	}
#pragma unroll
	for (int u = 0; u < PER; u++)
	{
		const uint32_t i = (uint32_t) tid + 256u * (uint32_t) u;

		key[u] = ~0ull;
		if (i < ns)
		{
			const float ub = s16_up(fmaxf(amat[(size_t) q * astride + i] + e, 0.0f));

			if (ok[u] && ub == ub && ub < 3.0e38f)
				key[u] = ((unsigned long long) __float_as_uint(ub) << 32) | i;		/* ub >= 0: the bits order like the values */
		}
	}
	/* the k-th smallest (k <= 64, distinct keys): every wave takes its own k smallest out one at a time (no barrier), the
	 * 4 k keys that leaves are ranked by counting */
	__shared__ unsigned long long s_top[4 * 64];
	unsigned long long kth;

	for (uint32_t r = 0; r < k; r++)
	{
		unsigned long long m = key[0];

#pragma unroll
		for (int u = 1; u < PER; u++)
			m = key[u] < m ? key[u] : m;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
		{
			const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) m, off, 64);
			const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (m >> 32), off, 64);
			const unsigned long long o = ((unsigned long long) hi << 32) | lo;

			m = o < m ? o : m;
		}
		if (lane == 0)
			s_top[w * 64 + r] = m;
#pragma unroll
		for (int u = 0; u < PER; u++)
			if (key[u] == m && m != ~0ull)
				key[u] = ~0ull;
	}
	if (tid == 0)
		s_red[0] = ~0ull;
	__syncthreads();
	{
		/* thread t < 4 k (<= 256): the rank of its key among the 4 k (keys are distinct but for the "none" value) */
		const uint32_t nt = 4u * k;
		const unsigned long long mine = (uint32_t) tid < nt ? s_top[((uint32_t) tid / k) * 64 + ((uint32_t) tid % k)] : ~0ull;
		uint32_t	rank = 0;

		for (uint32_t o = 0; o < nt; o++)
			rank += s_top[(o / k) * 64 + (o % k)] < mine ? 1u : 0u;
		if ((uint32_t) tid < nt && mine != ~0ull && rank == k - 1)
			s_red[0] = mine;
	}
	__syncthreads();
	kth = s_red[0];			/* "none": fewer than k candidates in the sample, no threshold from it */
	if (tid == 0 && kth != ~0ull)
	{
		const float t = s16c_t_from_ub(__uint_as_float((uint32_t) (kth >> 32)), dim);

		/* T >= 0 (or +inf): its bits order like the values */
		atomicMin(reinterpret_cast<unsigned int *>(&qthr[q].x), __float_as_uint(t));
	}
}

/*
 * Thresholds for 64 < k <= NDB_S16_MAXK (round 5).  The seed kernels bound the k-th distance by the k-th smallest of at
 * most 64 rows' distances — nothing for k > 64, and without a finite first threshold every pair survives and every
 * query overflows its record buffer.  What bounds the k-th distance of a larger k needs no row at
 * all: EVERY row of a bucket with centre c and radius rad lies within |q - c| + rad of the query, so with the buckets of
 * the query's probed lists in ascending order of U = (|q - c| + rad)^2, the first U at which the buckets so far hold k
 * of the query's candidates (live rows, visible under the candidate cap: counted from pposof, a bucket at a time) is an
 * upper bound of its k-th smallest REAL squared distance — what s16c_t_from_ub takes.  |q - c|: from the matrix-core
 * centre distances (a + their error bound, the one k_sub_pairs prunes with), or the centroid scan's float4 distance for
 * a list that is its own bucket (within 1e-3 of the real one, as s16_sub_excluded takes it).  k = 100 on a clustered
 * table: U of the query's own sublist, a few times its real k-th distance; the emissions that costs (the rows of the
 * neighbouring sublists) are cut by k_s16_finalize's k-th UPPER bound as ever.
 * One block per query; up to S16C_RAD_CAP buckets (more: the threshold stays where it was).  L2 only.
 */
/* inner product: the largest M^2 - |x|^2 of every bucket's plane rows (holes and padding hold 0, which M^2 - |x|^2 never
 * falls below); one wave a bucket */
__global__ __launch_bounds__(64) void
k_ipc_bucket_max(const float *__restrict__ rnx, const int64_t *__restrict__ prow_off, const uint32_t *__restrict__ blen, int nb,
				 float *__restrict__ out)
{
	const int	b = blockIdx.x, lane = threadIdx.x;

	if (b >= nb)
		return;
	const int64_t r0 = prow_off[b];
	float		m = 0.0f;

	for (uint32_t i = (uint32_t) lane; i < blen[b]; i += 64)
	{
		const float v = rnx[r0 + i];

		m = v > m ? v : m;			/* (a NaN — a row beyond fp32 — never wins: its bucket's radius is +inf and rules it out) */
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		const float o = __shfl_xor(m, off, 64);

		m = o > m ? o : m;
	}
	if (lane == 0)
		out[b] = m;
}

#define S16C_RAD_CAP 1024
#define S16C_RAD_STEPS 32		/* buckets taken in order before giving up (k <= 256 needs a handful of ~128-row sublists) */
__global__ __launch_bounds__(256) void
k_s16c_thr_radius(const int *__restrict__ probes, const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t k,
				  const uint32_t *__restrict__ sub_first, const int *__restrict__ sub_gidx, const uint32_t *__restrict__ sub_len,
				  const uint32_t *__restrict__ sub_rad, const int64_t *__restrict__ prow_off, const uint32_t *__restrict__ pposof,
				  const float *__restrict__ subdist, uint32_t sstride, const float *__restrict__ pdist,
				  const float *__restrict__ cdist, uint32_t cstride, const float *__restrict__ qn2,
				  const uint32_t *__restrict__ cxmax_bits, int dim, float2 *__restrict__ qthr,
				  const float *__restrict__ rnxmax = nullptr /* inner product on the centred sweep (thresholds in b's domain,
															  * b = |q - x|^2 + M^2 - |x|^2): every bucket's largest M^2 - |x|^2
															  * (k_ipc_bucket_max) goes on top of (|q - c| + rad)^2 ... */,
				  const float *__restrict__ qev = nullptr /* ... and the threshold converts with s16c_ip_t_from_ub (ev per query) */,
				  int cos = 0 /* cosine on the centred sweep: centres, radii and distances are those of the NORMALISED rows and
							   * queries (|q^ - x^|^2 = 2 x the cosine distance), the threshold converts with s16c_cos_t_from_ub; a
							   * list that is its own bucket has no centre distance in that space (the centroid scan's is in the
							   * rows' own): it is not taken */ )
{
	__shared__ uint32_t s_off[65], s_s0[64];
	__shared__ float s_pd[64];
	__shared__ uint32_t s_key[S16C_RAD_CAP], s_sx[S16C_RAD_CAP];
	__shared__ uint16_t s_pr[S16C_RAD_CAP];
	__shared__ unsigned long long s_best[4];
	__shared__ uint32_t s_cnt[4];
	const uint32_t q = blockIdx.x;
	const int	tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t *lco = loc_cand_off + (size_t) q * (npr + 1);
	const float ec = s16_e<R_IVF_L2>(dim, qn2[q], __uint_as_float(*cxmax_bits), false);
	uint32_t	nb = 0;
	bool		toomany = false;

	for (int p0 = 0; p0 < npr; p0 += 64)
	{
		const int	p = p0 + lane;
		uint32_t	n = 0, s0 = 0;
		float		pd = 0.0f;

		if (p < npr && lco[p + 1] != lco[p])
		{
			const int	L = probes[(size_t) q * npr + p];

			s0 = sub_first[L];
			n = sub_first[L + 1] - s0;
			pd = cdist ? cdist[(size_t) q * cstride + L] : pdist[(size_t) q * npr + p];
		}
		uint32_t	inc = n;

#pragma unroll
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t v = (uint32_t) __shfl_up((int) inc, off, 64);

			if (lane >= off)
				inc += v;
		}
		const uint32_t T = (uint32_t) __shfl((int) inc, 63, 64);

		__syncthreads();			/* (the previous round's readers are done with the tables) */
		if (wv == 0)
		{
			s_off[lane] = inc - n;
			s_s0[lane] = s0;
			s_pd[lane] = pd;
			if (lane == 0)
				s_off[64] = T;
		}
		__syncthreads();
		if (nb + T > (uint32_t) S16C_RAD_CAP)
		{
			toomany = true;			/* uniform */
			break;
		}
		for (uint32_t t = (uint32_t) tid; t < T; t += 256)
		{
			int			lo = 0, hi = 64;

			while (hi - lo > 1)
			{
				const int	mid = (lo + hi) >> 1;

				if (s_off[mid] <= t)
					lo = mid;
				else
					hi = mid;
			}
			const uint32_t sx = s_s0[lo] + (t - s_off[lo]);
			const int	gi = sub_gidx[sx];
			const double rad = (double) __uint_as_float(sub_rad[sx]);
			double		du;

			if (gi < 0)
				du = (double) s_pd[lo] * (1.0 + 1e-3);
			else
				du = __builtin_sqrt((double) fmaxf(subdist[(size_t) q * sstride + gi], 0.0f) + (double) ec * (1.0 + 1e-6)) * (1.0 + 1e-9);
			const double u = ((du + rad) * (du + rad) + (rnxmax ? (double) rnxmax[sx] * (1.0 + 1e-6) : 0.0)) * (1.0 + 1e-9);
			const float uf = s16_up((float) u);
			/* (a centre or a radius beyond fp32, an empty bucket: never taken) */
			const bool	good = sub_len[sx] > 0 && uf == uf && uf < 3.0e38f && uf >= 0.0f && !(cos && gi < 0);

			s_key[nb + t] = good ? __float_as_uint(uf) : 0xFFFFFFFFu;		/* uf >= 0: the bits order like the values */
			s_sx[nb + t] = sx;
			s_pr[nb + t] = (uint16_t) (p0 + lo);
		}
		nb += T;
	}
	__syncthreads();
	if (toomany || nb == 0)
		return;
	uint32_t	have = 0;

	for (int step = 0; step < S16C_RAD_STEPS; step++)
	{
		/* the nearest bucket not taken yet: (key, index) minimum over the block */
		unsigned long long m = ~0ull;

		for (uint32_t i = (uint32_t) tid; i < nb; i += 256)
		{
			const unsigned long long v = ((unsigned long long) s_key[i] << 32) | i;

			m = v < m ? v : m;
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
		{
			const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) m, off, 64);
			const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (m >> 32), off, 64);
			const unsigned long long o = ((unsigned long long) hi << 32) | lo;

			m = o < m ? o : m;
		}
		if (lane == 0)
			s_best[wv] = m;
		__syncthreads();
		m = s_best[0];
#pragma unroll
		for (int w = 1; w < 4; w++)
			m = s_best[w] < m ? s_best[w] : m;
		if ((uint32_t) (m >> 32) == 0xFFFFFFFFu)
			return;					/* nothing usable left (uniform) */
		const uint32_t bi = (uint32_t) m, sx = s_sx[bi], pr = s_pr[bi];
		const uint32_t vis = lco[pr + 1] - lco[pr], len = sub_len[sx];
		const int64_t r0 = prow_off[sx];
		uint32_t	c = 0;

		for (uint32_t i = (uint32_t) tid; i < len; i += 256)
			c += pposof[r0 + i] < vis ? 1u : 0u;		/* (a deleted row's hole: 0xFFFFFFFF) */
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
			c += (uint32_t) __shfl_xor((int) c, off, 64);
		if (lane == 0)
			s_cnt[wv] = c;
		__syncthreads();
		have += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
		if (tid == 0)
			s_key[bi] = 0xFFFFFFFFu;		/* taken */
		if (have >= k)
		{
			if (tid == 0)
			{
				const float ubk = __uint_as_float((uint32_t) (m >> 32));
				const float t = cos ? s16c_cos_t_from_ub(ubk, dim) : (rnxmax ? s16c_ip_t_from_ub(ubk, qev[q]) : s16c_t_from_ub(ubk, dim));
				const float2 o = qthr[q];

				qthr[q] = make_float2(fminf(o.x, t), o.y);
			}
			return;
		}
		__syncthreads();
	}
}

/*
 * Geometry.  QB = 32-pair blocks per tile (4: 128 pairs; 1: 32 pairs, for batches whose buckets are probed by a
 * handful of queries each — most of a 128-pair tile would be padding, and the LDS it reserves is better spent on a
 * deeper ring, because that regime is bound by the rows' bytes); 4 waves, 128 rows per tile.  QB = 8 is the dense
 * form, for buckets with hundreds of pairs AND hundreds of rows (an i.i.d. table: every product is needed, the sweep
 * is a dense contraction): 8 waves, 256 pairs x 256 rows per tile, a wave owns 64 pairs x 128 rows — per 16 dims 6
 * ds_read_b128 feed 8 MFMAs (4 feed 4 in the 128 x 128 tile, which ran into the LDS's 128 B/clk: profiles/r03_pmc_gauss_sq),
 * and a tile's operands (2 x 393 KB) serve 4 x the products of a 128 x 128 tile's (2 x 196 KB): half the bytes per
 * product through L2 and HBM, which at 4.4 TB/s were the other limit.
 * Wave w = (wq, wr) = (w % NWQ, w / NWQ) owns AQ pair blocks x BR row blocks.
 */
template <int QB> struct S16CGeom
{
	static constexpr int NW = QB == 8 ? 8 : 4;	/* waves per block */
	static constexpr int RB = QB == 8 ? 8 : 4;	/* 32-row blocks per tile: one per wave and chunk to fetch */
	static constexpr int RT = 32 * RB;
	static constexpr int NWQ = QB == 8 ? 4 : (QB >= 2 ? 2 : 1);
	static constexpr int NWR = NW / NWQ;
	static constexpr int AQ = QB / NWQ;
	static constexpr int BR = RB / NWR;
	static constexpr int Q_OFF = RB * 4096;
	static constexpr int BUF = Q_OFF + QB * 4096;
	static constexpr int Q_DMA = 4 * QB / NW;	/* 1 KiB pieces of the pair area per wave and chunk (4 QB pieces in all) */
	static constexpr int PER = 4 + Q_DMA;
	static constexpr int QT = 32 * QB;
	/* the epilogue keeps one bit per accumulator element in a 64-bit mask: BG row blocks (x AQ pair blocks x 16
	 * registers) at a time, NG times */
	static constexpr int BG = 64 / (AQ * 16) < BR ? 64 / (AQ * 16) : BR;
	static constexpr int NG = (BR + BG - 1) / BG;
	static_assert(BR % BG == 0, "the epilogue walks the row blocks BG at a time");
};

/* a 4-byte-per-lane LDS DMA: lane i of the wave copies the dword at base + voff to LDS address la + 4 i */
__device__ __forceinline__ void
s16_dma4(const void *base, uint32_t voff, uint32_t la)
{
	asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
				 :: "s"(la), "v"(voff), "s"(base) : "memory");
}

/*
 * The sweep.  Work item = (bucket, 128-row tile, QT-pair tile), expanded by k_s16_items into 8 runs, one per XCD.
 * A block walks the items of its XCD's run at a fixed stride (item = run start + block / 8 + j blocks / 8): the next
 * item is known without asking anyone, so its descriptor comes through the scalar cache (s_load: no entry in the
 * vector-memory counter), its members' constants come by LDS DMA like the operands, and the operand stream never
 * stops at an item's end — the ring's look-ahead simply runs into the next item's first chunks while the current
 * item's last chunks are multiplied and its results looked at.  (The first version popped items from atomic queues
 * and fetched descriptors and member records with ordinary loads, one dependent stage per chunk: every stage's wait
 * drained the wave's vector-memory counter, i.e. waited for the operand chunks in flight as well, and an item's first
 * half ran with a ring of depth one.)
 *
 * Ring of NBUF chunk buffers; chunk g of the block's stream lives in buffer g % NBUF.  Per chunk: wait until the
 * chunk's own DMA has landed (s_waitcnt vmcnt leaves the later chunks' requests in flight), barrier, request chunk
 * g + NBUF - 1 into the buffer everybody has finished reading, 16 ds_read_b128 + 4 AQ BR MFMAs.
 */
template <int QB, int NBUF, int DBG = 0, int EPI = 1, int VAR = 0>
__global__ __launch_bounds__(64 * S16CGeom<QB>::NW, ((S16CGeom<QB>::BUF * NBUF + 8192) * 2 <= 160 * 1024) ? 2 : 1) void
k_s16c_sweep(int dim, int nbuckets, const int64_t *__restrict__ loc_off, const uint32_t *__restrict__ own_len,
			 const unsigned char *__restrict__ planes, const uint32_t *__restrict__ blk_off,
			 const float *__restrict__ rn2, const int16_t *__restrict__ rexp,
			 const unsigned char *__restrict__ qcplanes, uint32_t qrowbytes, const float *__restrict__ qcn2,
			 const int *__restrict__ qcexp, const uint32_t *__restrict__ pqid, const uint32_t *__restrict__ pla,
			 const uint32_t *__restrict__ pnrow, float2 *qthr, const uint32_t *__restrict__ cnt,
			 const uint32_t *__restrict__ pair_off, const S16Desc *__restrict__ desc,
			 const uint32_t *__restrict__ runs, unsigned int *__restrict__ ecount, uint2 *__restrict__ erec,
			 float *__restrict__ eub, uint32_t ecap, uint32_t *__restrict__ bmin, int nchunk,
			 uint32_t desc_cap, uint32_t topk, const uint32_t *__restrict__ pos_of, float cE,
			 uint32_t qc_cap /* rows of qcplanes: more pairs than that and nothing is swept (k_s16c_qcprep raised the flag) */,
			 int cosine = 0 /* the planes are normalised vectors' and the thresholds bound cosine distances (s16c_cos_t_from_ub) */,
			 const float *__restrict__ rnx = nullptr /* inner product (ndbhip_screen16.h: s16c_ip_*): M^2 - |x|^2 per padded plane
													  * row, added to every bound of the row; the thresholds are in b's domain */,
			 const float *__restrict__ qev = nullptr /* ... and ev per query, for the tightening */ )
{
	typedef S16CGeom<QB> G;
	/* VAR: bits 0-3 = chunks an L2 prefetch runs ahead (0: none), bit 4 = the rows' stream is non-temporal, bit 5 = the pairs' */
	constexpr int PF = VAR & 15;
	constexpr bool RNT = (VAR & 16) != 0, QNT = (VAR & 32) != 0;
	__shared__ uint32_t s_tn, s_tq[S16_TIGHT_Q], s_tkeys[S16_NB];
	__shared__ __attribute__((aligned(1024))) unsigned char ring[NBUF * G::BUF];
	/* per member of the two items in flight (by the item's parity): DMA'd from the pair-ordered arrays */
	__shared__ __attribute__((aligned(256))) float s_q2[2][64 * ((G::QT + 63) / 64)];
	__shared__ __attribute__((aligned(256))) int s_eq[2][64 * ((G::QT + 63) / 64)];
	__shared__ __attribute__((aligned(256))) uint32_t s_la[2][64 * ((G::QT + 63) / 64)], s_nrow[2][64 * ((G::QT + 63) / 64)],
		s_qid[2][64 * ((G::QT + 63) / 64)];
	__shared__ float s_t2[G::QT];
	/* EPI: the members' two operands of the test instruction (k = 0: -v, k = 1: -u, below), and whether some exponent
	 * of the item (by the item's parity) lies outside the range the test instruction is proved for */
	__shared__ float s_nuv[2][G::QT];
	__shared__ uint32_t s_wild[2];
	/* PF: where the L2 prefetch's 4-byte-per-lane LDS DMA lands (nobody reads it) */
	__shared__ __attribute__((aligned(256))) uint32_t s_sink[PF ? 64 : 1];
	static_assert(PF == 0 || (NBUF == 2 && G::Q_DMA == 4), "the L2 prefetch is written for the ring of two and 32 pair rows per wave");
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const int	wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	wq = wave % G::NWQ, wr = wave / G::NWQ;
	const int	r32 = lane & 31, kh = lane >> 5;

	if (pair_off[nbuckets] > qc_cap)
		return;					/* uniform */
	if (tid < 2)
		s_wild[tid] = 0;		/* (the first chunk's barrier comes before anybody looks) */
	/* this block's items: those of run (block % 8) at stride (blocks in that XCD) */
	const uint32_t xq = blockIdx.x & 7u;
	const uint32_t stride = (gridDim.x - xq + 7u) >> 3;
	const uint32_t run_hi = min(runs[xq + 1], desc_cap);
	uint32_t	it_c = runs[xq] + (blockIdx.x >> 3);		/* the item being multiplied */

	if (it_c >= run_hi)
		return;
	if (tid == 0)
		s_tn = 0;

	/* ---- fetch side: the item whose chunks are being requested, and the addresses its DMA needs ---- */
	uint32_t	it_f = it_c, f_c = 0, f_par = 0;
	uint32_t	voff_q[G::Q_DMA];
	bool		qdma = true;
	const unsigned char *rbase = planes, *qbase = qcplanes;
	const uint32_t lane16 = (uint32_t) lane * 16u;
	const uint32_t ring_la = (uint32_t) (uintptr_t) (ndb_lds_ptr) ring;
	const uint32_t mem_la[5] = {(uint32_t) (uintptr_t) (ndb_lds_ptr) &s_q2[0][0], (uint32_t) (uintptr_t) (ndb_lds_ptr) &s_eq[0][0],
		(uint32_t) (uintptr_t) (ndb_lds_ptr) &s_la[0][0], (uint32_t) (uintptr_t) (ndb_lds_ptr) &s_nrow[0][0],
		(uint32_t) (uintptr_t) (ndb_lds_ptr) &s_qid[0][0]};
	constexpr uint32_t MEM_STRIDE = 4u * 64u * ((G::QT + 63) / 64);	/* bytes between the two parities of a member array */

	/* item `it` becomes the fetch item: addresses of its operand pieces (wave w: 32-row block w of the tile, one
	 * contiguous 4 KiB piece per chunk, and its share of the tile's pair rows — consecutive rows of qcplanes, 128
	 * bytes per chunk, 16-byte slots XOR-swizzled at the source), and its members' constants requested into the
	 * member arrays of parity `par` (waves 0 .. QT / 64: five 4-byte-per-lane DMA instructions each) */
	auto		enter = [&](uint32_t it, uint32_t par) {
		const S16Desc d = desc[it];			/* uniform address: scalar loads */
		const uint32_t L = d.L, nmem = min((uint32_t) G::QT, cnt[L] - d.qt * G::QT);
		const uint32_t nbk = blk_off[L + 1] - blk_off[L];
		const uint32_t b = min(d.t2 * (uint32_t) G::RB + (uint32_t) wave, nbk - 1u);
		const uint32_t slot0 = pair_off[L] + d.qt * G::QT;

#pragma unroll
		for (int j = 0; j < G::Q_DMA; j++)
		{
			const int	piece = wave * G::Q_DMA + j;
			const int	rr = 8 * (piece & 3) + (lane >> 3);
			const uint32_t mem = min((uint32_t) (32 * (piece >> 2) + rr), nmem - 1u);

			voff_q[j] = mem * qrowbytes + 16u * (uint32_t) ((lane & 7) ^ ((rr >> 1) & 7));
		}
		rbase = planes + ((size_t) blk_off[L] + b) * (size_t) nchunk * 4096;
		qbase = qcplanes + (size_t) slot0 * qrowbytes;
		/* a pair block without a member is neither fetched nor multiplied */
		qdma = nmem > (uint32_t) (32 * ((wave * G::Q_DMA) >> 2));
		if (wave * 64 < G::QT)
		{
			/* members beyond the tile's count read the last member's constants (never looked at: nrow is tested
			 * against the member count below) */
			const uint32_t mo = 4u * min((uint32_t) (wave * 64 + lane), nmem - 1u);
			const void *src[5] = {qcn2 + slot0, qcexp + slot0, pla + slot0, pnrow + slot0, pqid + slot0};

#pragma unroll
			for (int a = 0; a < 5; a++)
				s16_dma4(s16_uniform_ptr((const unsigned char *) src[a]), mo, mem_la[a] + par * MEM_STRIDE + (uint32_t) wave * 256u);
		}
	};
	auto		issue = [&](uint32_t c, uint32_t bufi) {
		if constexpr (DBG == 1)
			return;
		/* (DBG 3 / 4: only the rows' / only the pairs' operands come from where they are) */
		const unsigned char *rb = s16_uniform_ptr((DBG == 2 || DBG == 4) ? planes : rbase + (size_t) c * 4096);
		const unsigned char *qb = s16_uniform_ptr((DBG == 2 || DBG == 3) ? qcplanes : qbase + (size_t) c * 128);
		const uint32_t la = ring_la + bufi * G::BUF;

		s16_dma_linear<4, RNT>(rb, lane16, la + wave * 4096);
		if (qdma)
		{
#pragma unroll
			for (int j = 0; j < G::Q_DMA; j++)
				s16_dma16<QNT>(qb, voff_q[j], la + G::Q_OFF + (wave * G::Q_DMA + j) * 1024);
		}
	};
	/* request the next chunk of the stream (none left: nothing); returns whether this wave's request included pair
	 * rows (what the matching wait has to count) */
	uint32_t	g_f = 0;		/* chunks requested so far */
	auto		fetch_next = [&]() -> bool {
		if (it_f == S16_NOITEM)
			return false;
		if (f_c == (uint32_t) nchunk)
		{
			/* the next item, entered only now that its first chunk is due: NBUF - 1 <= nchunk chunks ahead of the
			 * multiplication, i.e. while the item before it is being multiplied — whose member arrays have the
			 * other parity, and whose predecessor (this parity) has been looked at */
			f_c = 0;
			it_f = it_f + stride < run_hi ? it_f + stride : S16_NOITEM;
			f_par ^= 1u;
			if (it_f == S16_NOITEM)
				return false;
			enter(it_f, f_par);
		}
		const bool	had_q = qdma;

		issue(f_c, g_f % NBUF);
		g_f++;
		f_c++;
		return had_q;
	};

	/*
	 * PF > 0: an L2 prefetch runs PF chunks ahead of the operand stream.  The ring holds one chunk in flight (64 KiB a
	 * block: all the LDS there is), which covers an L2 hit's latency and not a miss's; two thirds of the requests hit
	 * (the blocks of an XCD share row and pair tiles), but a chunk is as late as its latest line, so nearly every chunk
	 * waited for HBM: 8.2 ms a batch on the i.i.d. table against 3.6 ms with every request a hit, at 1.9 TB/s — a
	 * quarter of what HBM delivers.  So every wave touches, with one 4-byte-per-lane LDS DMA for the row block and one
	 * for the pair rows (a line per lane; the data lands in a sink nobody reads), the 64 lines it will ask for PF chunks
	 * later: by then they are in this XCD's L2.  The cursor (it_p, p_c) walks the same static schedule as the stream.
	 */
	uint32_t	it_p = it_c, p_c = 0, voff_qp = 0;
	const unsigned char *rbase_p = planes, *qbase_p = qcplanes;
	const uint32_t sink_la = (uint32_t) (uintptr_t) (ndb_lds_ptr) s_sink;
	auto		enter_p = [&](uint32_t it) {
		const S16Desc d = desc[it];			/* uniform address: scalar loads */
		const uint32_t L = d.L, nmem = min((uint32_t) G::QT, cnt[L] - d.qt * G::QT);
		const uint32_t nbk = blk_off[L + 1] - blk_off[L];
		const uint32_t b = min(d.t2 * (uint32_t) G::RB + (uint32_t) wave, nbk - 1u);

		rbase_p = planes + ((size_t) blk_off[L] + b) * (size_t) nchunk * 4096;
		qbase_p = qcplanes + (size_t) (pair_off[L] + d.qt * G::QT) * qrowbytes;
		voff_qp = min((uint32_t) (32 * wave + r32), nmem - 1u) * qrowbytes;
	};
	auto		prefetch_next = [&]() -> bool {
		if constexpr (PF == 0)
			return false;
		if (it_p == S16_NOITEM)
			return false;
		if (p_c == (uint32_t) nchunk)
		{
			p_c = 0;
			it_p = it_p + stride < run_hi ? it_p + stride : S16_NOITEM;
			if (it_p == S16_NOITEM)
				return false;
			enter_p(it_p);
		}
		s16_dma4(s16_uniform_ptr(rbase_p + (size_t) p_c * 4096), (uint32_t) r32 * 128u, sink_la);
		s16_dma4(s16_uniform_ptr(qbase_p + (size_t) p_c * 128), voff_qp, sink_la);
		p_c++;
		return true;
	};
	bool		pf_last = false;		/* the newest two requests of this wave are a prefetch (they may stay in flight across the chunk's wait) */

	const int	sw = (r32 >> 1) & 7;
	const int	qfrag = G::Q_OFF + (G::AQ * wq) * 4096 + r32 * 128;
	const int	rfrag = (G::BR * wr) * 4096 + lane * 16;		/* (fragment-major row images: s16c_unit) */
	uint32_t	c_par = 0, g_c = 0;		/* parity of the item being multiplied; chunks consumed so far */
	/* did the request for chunk g (g = g_c + 1 .. g_c + NBUF - 2, the ones that stay in flight across a wait) include
	 * pair rows: one bit per ring slot */
	uint32_t	qbits = 0;

	if constexpr (PF > 0)
	{
		/* (the first PF + NBUF - 1 chunks are touched too: the stream asks for the first of them right behind) */
		enter_p(it_c);
		for (int p = 0; p < PF + NBUF - 1; p++)
			(void) prefetch_next();
	}
	enter(it_c, 0);
#pragma unroll
	for (int p = 0; p < NBUF - 1; p++)
	{
		const uint32_t slot = g_f % NBUF;
		const bool	hq = fetch_next();

		qbits = (qbits & ~(1u << slot)) | ((hq ? 1u : 0u) << slot);
	}

	for (;;)
	{
		/* the item being multiplied: its descriptor again (scalar cache) */
		const S16Desc dc = desc[it_c];
		const uint32_t L = dc.L, t2 = dc.t2;
		const uint32_t nmem_cur = min((uint32_t) G::QT, cnt[L] - dc.qt * G::QT);
		const uint32_t len = own_len[L];
		ndb_f16acc	acc[G::AQ][G::BR];

#pragma unroll
		for (int a = 0; a < G::AQ; a++)
#pragma unroll
			for (int b = 0; b < G::BR; b++)
#pragma unroll
				for (int i = 0; i < 16; i++)
					acc[a][b][i] = 0.0f;

		/* pair blocks of this wave that hold a member (wave-uniform) */
		int			na = 0;

#pragma unroll
		for (int a = 0; a < G::AQ; a++)
			na += nmem_cur > (uint32_t) (32 * (G::AQ * wq + a)) ? 1 : 0;

		auto		compute = [&](const unsigned char *buf) {
			if (QB != 8 && na == 0)
				return;
#pragma unroll
			for (int s = 0; s < 4; s++)
			{
				ndb_h8		ah[G::AQ], bh[G::BR];

#pragma unroll
				for (int a = 0; a < G::AQ; a++)
					ah[a] = *reinterpret_cast<const ndb_h8 *>(buf + qfrag + a * 4096 + (((2 * s + kh) ^ sw) * 16));
#pragma unroll
				for (int b = 0; b < G::BR; b++)
					bh[b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * 4096 + s * 1024);
#pragma unroll
				for (int a = 0; a < G::AQ; a++)
				{
					/* (the dense tile multiplies its empty pair blocks too: without the test the k-steps are straight-line
					 * code and the next one's ds_reads are issued under this one's MFMAs) */
					if (QB == 8 || a < na)
					{
#pragma unroll
						for (int b = 0; b < G::BR; b++)
							acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
					}
				}
			}
		};

		/* one chunk of the stream: wait for it, barrier, request the chunk NBUF - 1 ahead, multiply */
		auto		chunk = [&]() {
			/* chunk g_c must have landed; the requests for the NBUF - 2 chunks after it may stay in flight: 4 row
			 * pieces each, plus Q_DMA pair pieces where the request had them (anything else the wave has asked for —
			 * member constants — is older than those and is waited for along with the chunk) */
			if constexpr (NBUF == 2 && PF > 0)
			{
				/* (requests retire in order: leaving the newest two — the prefetch — leaves nothing of the chunk) */
				if (pf_last)
					s16_wait_vm<2>();
				else
					s16_wait_vm<0>();
			}
			else if constexpr (NBUF == 2)
				s16_wait_vm<0>();
			else
			{
				static_assert(NBUF == 3, "ring depths 2 and 3");
				const uint32_t later = g_f - g_c - 1;		/* chunks requested after g_c: NBUF - 2, or fewer at the stream's end */

				if (later == 0)
					s16_wait_vm<0>();
				else if ((qbits >> ((g_c + 1) % NBUF)) & 1u)
					s16_wait_vm<G::PER>();
				else
					s16_wait_vm<4>();
			}
			__syncthreads();
			{
				const uint32_t slot = g_f % NBUF;
				const bool	hq = fetch_next();

				qbits = (qbits & ~(1u << slot)) | ((hq ? 1u : 0u) << slot);
			}
			pf_last = prefetch_next();
			compute(ring + (g_c % NBUF) * G::BUF);
			g_c++;
		};

		chunk();
		/*
		 * Everything else the item reads with ordinary loads — its members' thresholds as they stand now (in-sweep
		 * tightening; a stale value is a valid, looser bound) and its rows' norms / exponents / list positions — is
		 * requested HERE, once, behind the first chunk's barrier (the member arrays have landed), and looked at after
		 * the last chunk.  The vector-memory counter retires in order: a load consumed in the epilogue would wait for
		 * every operand chunk requested before it, and one whose result is not consumed on every path would leave the
		 * compiler waiting for "it" — i.e. for the whole ring — wherever its register is reused inside the loop.
		 * Requested here, these loads are older than the stream and cost nothing; the empty asm statements behind
		 * the loop pin their first use there, so that no arithmetic on them (and no wait) is hoisted into the loop.
		 */
		float		tfresh = 0.0f, x2r[G::BR], rxr[G::BR];
		int			exr[G::BR];
		uint32_t	porr[G::BR];
		bool		rokr[G::BR];

		if (tid < G::QT && (uint32_t) tid < nmem_cur)
			tfresh = qthr[s_qid[c_par][tid]].x;
#pragma unroll
		for (int b = 0; b < G::BR; b++)
		{
			const uint32_t ridx = t2 * G::RT + (uint32_t) (32 * (G::BR * wr + b) + r32);
			const size_t grow = (size_t) loc_off[L] + (ridx < len ? ridx : len - 1);

			rokr[b] = ridx < len;
			x2r[b] = rn2[grow];
			rxr[b] = rnx ? rnx[grow] : 0.0f;
			exr[b] = (int) rexp[grow];
			porr[b] = pos_of ? pos_of[grow] : ridx;
		}
		for (int c = 1; c < nchunk; c++)
			chunk();
		asm volatile("" : "+v"(tfresh));
#pragma unroll
		for (int b = 0; b < G::BR; b++)
		{
			asm volatile("" : "+v"(x2r[b]), "+v"(exr[b]), "+v"(porr[b]), "+v"(rxr[b]));
			exr[b] -= 27;
		}
		if (tid < G::QT)
		{
			/* what the test subtracts: T rounded up with the slack its fused form needs (see below) */
			s_t2[tid] = s16_up(tfresh * 1.000001f) + NDB_S16_ABS;
			if constexpr (EPI != 0)
			{
				/* (an index the compiler cannot see through: the dense tile runs at the register file's limit, and an
				 * LDS address computed in the prologue for this block would live — in scratch — across the whole sweep) */
				int			tix = tid;

				asm volatile("" : "+v"(tix));
				/* the member's operands of the test instruction (see "pass 0" below): u = (KB Q2 - TB) 2^(27 - eq) and
				 * v = KB 2^(27 - eq), negated.  A member the tile does not have never emits (u = +inf); a NaN — a norm
				 * that is not a finite fp32, a threshold that is +inf — always does (u = -inf). */
				const bool	valid = (uint32_t) tix < nmem_cur;
				const int	eq = s_eq[c_par][tix];
				const float KB = (1.0f - cE) * 0.9999962f;
				const float TB = s16_up(tfresh * 1.000004f) + NDB_S16_ABS;
				float		cm = __builtin_fmaf(s_q2[c_par][tix], KB, -TB);

				if (!(cm == cm))
					cm = -__builtin_inff();
				s_nuv[0][tix] = -ldexpf(KB, 27 - eq);
				s_nuv[1][tix] = valid ? -ldexpf(cm, 27 - eq) : -__builtin_inff();
				if (valid && (eq < -20 || eq > 20))
					s_wild[c_par] = 1u;
			}
		}
		if constexpr (EPI != 0)
		{
			bool		w = false;

#pragma unroll
			for (int b = 0; b < G::BR; b++)
				w = w || (rokr[b] && (exr[b] + 27 < -20 || exr[b] + 27 > 20 || rxr[b] > 1.0e30f));
			if (w)
				s_wild[c_par] = 1u;
			if (tid == 0)
				s_wild[c_par ^ 1u] = 0;		/* read by the item before this one, set next by the item after it */
		}
		__syncthreads();

		/*
		 * Epilogue: element (reg, lane) of block (a, b) = member 32 (AQ wq + a) + (reg & 3) + 8 (reg >> 2) + 4 kh,
		 * row 32 (BR wr + b) + r32.  acc = (q - c).(x - c) 2^(28 - eq - ex);  t1 = 2 dot (a power-of-two scaling,
		 * exact);  the element is LEFT OUT when  a - E > T,  a = n - t1,  n = Q2 + X2,  E = cE n + ABS,  i.e. when
		 *     t1 < n (1 - cE) - ABS - T.
		 * Computed as t1 < fma(fl(Q2 + X2), K, -T2) with K = (1 - cE)(1 - 2^-20) and T2 = T (1 + 2^-20) (+ up) + ABS:
		 * fl(Q2 + X2) K <= n (1 + u)(1 - cE)(1 - 2^-20) and the fma's one rounding moves the result by at most
		 * u (n + T2), less than the 15 u (n (1 - cE) + T) the two 2^-20 factors give away (cE < 1/2), so the computed
		 * right-hand side never exceeds the real one.  NaN anywhere: the comparison is false, the element emitted.
		 */
		const float K = (1.0f - cE) * 0.99999905f;
		/* (a generic lambda per group of row blocks, not a loop: the group's index must be a compile-time constant for
		 * the accumulators to stay in registers, and the body is too large for the unroller to be trusted with) */
		auto		epilogue = [&](auto bgc) {
		constexpr int bg = decltype(bgc)::value;
		/* pass 1: which elements cannot be left out — one bit each, bit ((b - BG bg) AQ + a) 16 + reg, no memory traffic */
		unsigned long long emask = 0;

#pragma unroll
		for (int bb = 0; bb < G::BG; bb++)
		{
			const int	b = bg * G::BG + bb;	/* (BR is a multiple of BG) */
			const bool	rok = rokr[b];

#pragma unroll
			for (int a = 0; a < G::AQ; a++)
			{
				if (a >= na)
					continue;
#pragma unroll
				for (int reg = 0; reg < 16; reg++)
				{
					const int	m = 32 * (G::AQ * wq + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
					const float t1 = ldexpf(acc[a][b][reg], s_eq[c_par][m] + exr[b]);
					const float n = s_q2[c_par][m] + x2r[b];
					/* (inner product: the row's constant joins the bound, 2^-20 down: that covers the one more rounding) */
					const float rhs = __builtin_fmaf(n, K, -s_t2[m]) + rxr[b] * 0.99999905f;

					if ((DBG ? t1 == 1234.5f : !(t1 < rhs)) && rok && (uint32_t) m < nmem_cur && porr[b] < s_nrow[c_par][m])
						emask |= 1ull << ((bb * G::AQ + a) * 16 + reg);
				}
			}
		}
		/*
		 * pass 2 (only where something is emitted): the record slots.  The lanes of one half (kh) that emit the same
		 * element belong to ONE member = one query, so the slots of a member are handed out with a single atomicAdd:
		 * lane ml (the member's index among this wave's AQ x 32) first counts what its member emits (ballots), takes
		 * the slots in one go — every member's request travels at the same time — and the records are then written
		 * at base + rank.
		 */
		uint32_t	any_lo = (uint32_t) emask, any_hi = (uint32_t) (emask >> 32);

#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
		{
			any_lo |= (uint32_t) __shfl_xor((int) any_lo, off, 64);
			any_hi |= (uint32_t) __shfl_xor((int) any_hi, off, 64);
		}
		const unsigned long long anym = ((unsigned long long) any_hi << 32) | any_lo;	/* uniform */

		if (anym)
		{
			uint32_t	mycnt = 0;

			for (unsigned long long rest = anym; rest; rest &= rest - 1)
			{
				const int	e = __builtin_ctzll(rest);
				const int	reg = e & 15, a = (e >> 4) % G::AQ;
				const unsigned long long bal = __ballot((emask >> e) & 1ull);
				const int	ml0 = 32 * a + (reg & 3) + 8 * (reg >> 2);		/* kh = 0; kh = 1: + 4 */

				if (lane == ml0)
					mycnt += (uint32_t) __popcll(bal & 0xFFFFFFFFull);
				if (lane == ml0 + 4)
					mycnt += (uint32_t) __popcll(bal >> 32);
			}
			const int	mym = 32 * G::AQ * wq + lane;			/* the member lane `lane` keeps the books of */
			uint32_t	base = 0;

			if (mycnt != 0)
				base = atomicAdd(&ecount[s_qid[c_par][mym % G::QT]], mycnt);
#pragma unroll
			for (int bb = 0; bb < G::BG; bb++)
#pragma unroll
			for (int a = 0; a < G::AQ; a++)
#pragma unroll
			for (int reg = 0; reg < 16; reg++)
			{
				/* (unrolled: the accumulators are registers, an index known only at run time would send them to scratch) */
				const int	b = bg * G::BG + bb;
				const int	e = (bb * G::AQ + a) * 16 + reg;

				if (!((anym >> e) & 1ull))
					continue;			/* uniform */
				const bool	mine = (emask >> e) & 1ull;
				const unsigned long long bal = __ballot(mine);
				const int	ml = 32 * a + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
				const unsigned long long half = kh ? (bal >> 32) : (bal & 0xFFFFFFFFull);
				const uint32_t hb = (uint32_t) __shfl((int) base, ml, 64);

				if (mine)
				{
					const int	m = 32 * G::AQ * wq + ml;
					const uint32_t slot = hb + (uint32_t) __popcll(half & ((1ull << r32) - 1ull));
					const uint32_t q = s_qid[c_par][m];
					const float t1 = ldexpf(acc[a][b][reg], s_eq[c_par][m] + exr[b]);
					const float n = s_q2[c_par][m] + x2r[b];
					const float av = n - t1;
					const float er = s16_up(s16_up(cE * n) + NDB_S16_ABS);
					const float lbv = (av - er) + rxr[b] * 0.99999905f, ubv = s16_up(s16_up(av + er) + rxr[b] * 1.000001f);
					const float lb = lbv - fabsf(lbv) * 4.8e-7f - 1e-37f;
					const uint32_t pos = s_la[c_par][m] + porr[b], ub_bits = __float_as_uint(ubv);

					if (slot < ecap)
					{
						erec[(size_t) q * ecap + slot] = make_uint2(pos, __float_as_uint(lb));
						eub[(size_t) q * ecap + slot] = ubv;
					}
					/* the smallest upper bound of every hash bucket of positions (kept whether or not the record
					 * fit): k non-empty buckets are k distinct candidates */
					if ((ub_bits & 0x7FFFFFFFu) < 0x7F800000u)
						atomicMin(&bmin[(size_t) q * S16_NB + ((pos * 2654435761u) >> (32 - S16_NB_LOG2))],
								  ndb_key_from_bits(ub_bits));
					if (DBG == 0 && (slot & (S16_TIGHT - 1)) == S16_TIGHT - 1)
					{
						const uint32_t ti = atomicAdd(&s_tn, 1u);

						if (ti < S16_TIGHT_Q)
							s_tq[ti] = q;
					}
				}
				/* the two members' books move on by what their halves just took */
				const uint32_t nlo = (uint32_t) __popcll(bal & 0xFFFFFFFFull), nhi = (uint32_t) __popcll(bal >> 32);
				const int	ml0 = 32 * a + (reg & 3) + 8 * (reg >> 2);

				if (lane == ml0)
					base += nlo;
				if (lane == ml0 + 4)
					base += nhi;
			}
		}
		};
		/*
		 * Pass 0 (EPI): the test above costs a dozen vector instructions per accumulator element — at 128 elements a
		 * lane and two waves a SIMD as much time as the item's matrix instructions —, and all but a few in a million
		 * elements are left out.  So the matrix pipe evaluates the test itself: one v_mfma_f32_32x32x2_f32 per 32 x 32
		 * block (fp32 operands, D = fma(a1, b1, fma(a0, b0, C)): 4 % of the block's 4 nchunk fp16 instructions) with
		 *     A[member][k] = (-v, -u),   v = KB 2^(27 - eq),  u = (KB Q2 - TB) 2^(27 - eq)          (per member, LDS)
		 *     B[k][row]    = (w, s),     w = X2 2^-ex,        s = 2^-ex                               (per row)
		 *     C            = the block's accumulators = t1 P / 2... = dot 2^(28 - eq - ex) = t1 P,  P = 2^(27 - eq - ex)
		 * leaves fin = (t1 - (KB (Q2 + X2) - TB)) P up to two fp32 roundings, into registers of its own (the
		 * accumulators stay as they are), and the element may be left out when fin < 0:
		 *   - every scaling is by a power of two and exact while |eq|, |ex| <= 20 (u, v, w, s normal or zero, no
		 *     overflow: |u| < 2^(27 + eq) when it is positive); an item with an exponent outside that range
		 *     (`s_wild`) does not use the instruction's verdict at all;
		 *   - KB = (1 - cE)(1 - 2^-18) and TB = T (1 + 2^-18) (+ up) + ABS give away (1 - cE) n 2^-18 + T 2^-18
		 *     >= (32 n + 64 T) u of the real right-hand side n (1 - cE) - ABS - T  (n = Q2 + X2, cE < 1/2), while
		 *     everything that is rounded moves the computed one by less: KB itself 2 u n, the fma that makes
		 *     KB Q2 - TB u (Q2 + TB), the instruction's two steps — taken at 2 u each of |C| + |a0 b0| + |a1 b1|
		 *     <= (2 n + TB) P, twice what a correctly rounded fma chain costs — 4 u (2 n + TB) P: (11 n + 5 T) u P in all.
		 *     Hence fin < 0 implies t1 < n (1 - cE) - ABS - T in real numbers: what pass 1 tests;
		 *   - a row the tile does not have, or a hole, never emits (w = +inf: fin = -inf), a row whose norm is not a
		 *     finite fp32 always does (w = -inf); where the two kinds of infinity meet (NaN), one side is a member or
		 *     row that pass 1 throws out anyway.
		 * "Some fin >= 0 or NaN" = the largest of the 16 AQ BG bit patterns, taken as integers, is >= 0 (the part
		 * makes +NaN; a -NaN can only come from the meeting of infinities above).  Only then — a few times per
		 * thousand blocks — the group goes through passes 1 and 2, which decide as they always did.
		 */
		auto		screen = [&](auto bgc) -> bool {
			constexpr int bg = decltype(bgc)::value;

			if constexpr (EPI == 0)
				return true;
			else
			{
				int			mx = (int) 0x80000000;

#pragma unroll
				for (int bb = 0; bb < G::BG; bb++)
				{
					const int	b = bg * G::BG + bb;
					/* (exr is the row's exponent - 27) */
					const bool	nan = !(x2r[b] == x2r[b]) || !(rxr[b] == rxr[b]);
					const bool	dead = !rokr[b] || porr[b] == 0xFFFFFFFFu;
					/* (inner product: KB (X2 + rx' / KB) = KB X2 + rx' with rx' = the row's constant 2^-18 down — the quotient's
					 * and the sum's roundings and the instruction's own on one more term are inside that) */
					const float xw = rnx ? x2r[b] + rxr[b] * 0.999996f / ((1.0f - cE) * 0.9999962f) : x2r[b];
					const float w = dead ? __builtin_inff() : (nan ? -__builtin_inff() : ldexpf(xw, -(exr[b] + 27)));
					const float wb = kh ? ldexpf(1.0f, -(exr[b] + 27)) : w;

#pragma unroll
					for (int a = 0; a < G::AQ; a++)
					{
						if (a >= na)
							continue;
						const float ua = s_nuv[kh][32 * (G::AQ * wq + a) + r32];
						const ndb_f16acc fin = __builtin_amdgcn_mfma_f32_32x32x2f32(ua, wb, acc[a][b], 0, 0, 0);

#pragma unroll
						for (int reg = 0; reg < 16; reg++)
							mx = max(mx, __float_as_int(fin[reg]));
					}
				}
				const bool	hit = DBG ? mx == 0x12345678 : (mx >= 0 || s_wild[c_par] != 0);

				return __ballot(hit) != 0ull;
			}
		};
		if (screen(std::integral_constant<int, 0>{}))
			epilogue(std::integral_constant<int, 0>{});
		if constexpr (G::NG > 1)
		{
			if (screen(std::integral_constant<int, 1>{}))
				epilogue(std::integral_constant<int, 1>{});
		}
		if constexpr (G::NG > 2)
		{
			if (screen(std::integral_constant<int, 2>{}))
				epilogue(std::integral_constant<int, 2>{});
			if (screen(std::integral_constant<int, 3>{}))
				epilogue(std::integral_constant<int, 3>{});
		}
		static_assert(G::NG == 1 || G::NG == 2 || G::NG == 4, "groups of row blocks in the epilogue");
		if constexpr (DBG == 0)
		{
			/* a query that keeps emitting has a loose threshold: the k-th smallest bucket minimum bounds its k-th
			 * distance, so T is lowered here, while the sweep runs (monotone; any value read meanwhile is valid) */
			__syncthreads();
			const uint32_t tn = min(s_tn, (uint32_t) S16_TIGHT_Q);

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
					if (topk != 0 && rank == topk - 1 && mine != 0xFFFFFFFFu)
					{
						const uint32_t tb = (mine & 0x80000000u) ? (mine & 0x7FFFFFFFu) : ~mine;
						const float nt = qev ? s16c_ip_t_from_ub(__uint_as_float(tb), qev[q])
							: cosine ? s16c_cos_t_from_ub(__uint_as_float(tb), dim) : s16c_t_from_ub(__uint_as_float(tb), dim);

						/* T >= 0 (or +inf): its bits order like the values */
						atomicMin(reinterpret_cast<unsigned int *>(&qthr[q].x), __float_as_uint(nt));
					}
				}
				__syncthreads();
			}
			if (tid == 0 && s_tn != 0)
				s_tn = 0;
		}
		it_c += stride;
		if (it_c >= run_hi)
			break;
		c_par ^= 1u;
		/* (no barrier: the next item's first chunk starts with one, and the member arrays of parity c_par ^ 1 are
		 * requested again only by an `enter` behind that barrier) */
	}
}

#endif							/* NDBHIP_SCREEN16C_H */
