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
 * Every value the selection can pick or tie with is exact; everything else is provably larger, so the result
 * is the exact path's bit for bit.  If a query emits more than its record capacity (adversarial data: nearly
 * everything ties) the host reruns the batch through the older screened path, which has no capacity.
 */
#ifndef NDBHIP_SCREEN16_H
#define NDBHIP_SCREEN16_H

typedef _Float16 ndb_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 ndb_h2 __attribute__((ext_vector_type(2)));

#define S16_QT 128				/* queries per tile */
#define S16_RT 128				/* rows per tile */
#define S16_CH 32				/* dimensions per staged chunk (two MFMA k-steps) */
#define S16_SEED 64				/* candidates scored exactly per query for the first threshold */
#define S16_SURV_CAP 1024		/* survivors per query the finalize stage holds (= NDB_TOPK_FAST_CAP) */
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
 * One wave per row.  planes: [row][chunk of 32 dims][hi 32 halves | lo 32 halves], dimp = dim rounded up to 32
 * (the tail is zero).  rn2[row] = |x|^2, NaN for a row whose norm is not a finite fp32 (the sweep emits every
 * candidate of such a row, so it only affects itself: the reference's arithmetic decides).  rexp[row] = e.
 * xmax_bits: largest finite rn2 (bits order like values for non-negative floats).
 * H16 rows (halfvec mirror): decoded like fp16_to_float (SUBFIX: quirk Q20), norm only — the mirror itself is
 * the hi plane, there is no lo plane and e = 14.
 */
template <int H16>
__global__ __launch_bounds__(256) void
k_s16_row_prep(const void *__restrict__ vecs, int64_t nrows, int dim, int dimp, ndb_h2 *__restrict__ planes,
			   float *__restrict__ rn2, int16_t *__restrict__ rexp, uint32_t *__restrict__ xmax_bits)
{
	const int	lane = threadIdx.x & 63;
	const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);

	if (row >= nrows)
		return;
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
		rn2[row] = n2;
		if (rexp)
			rexp[row] = (int16_t) e;
		if (ok)
			atomicMax(xmax_bits, __float_as_uint(n2));
	}
	if constexpr (H16 == 0)
	{
		const float *x = (const float *) vecs + (size_t) row * dim;
		ndb_h2	   *out = planes + (size_t) row * dimp;	/* dimp * 4 bytes per row = dimp h2 */

		/* lane handles the element pairs (2p, 2p+1): hi pair p of the chunk, lo pair 16 + p */
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
 * First threshold: one wave per query scores its first S16_SEED candidates (probe order: the nearest list
 * first) with the reference's arithmetic; the k-th smallest of those distances bounds the query's k-th
 * distance.  Fewer than k candidates: +inf (everything is emitted).  Also E_q.
 * qinfo[q] = {thrE, E}.
 */
template <int R, int H16>
__global__ __launch_bounds__(64) void
k_s16_seed(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
		   const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t k, const float *__restrict__ qn2,
		   const uint32_t *__restrict__ xmax_bits, int sub, float2 *__restrict__ qthr)
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
			thrE = s16_thr_from_ref<R>(thr, e, dim);
	}
	if (lane == 0)
		qthr[q] = make_float2(thrE, e);
}

/* ------------------------------------------------------------------------------------------------------------
 * The sweep.  Work item = (list, 128-row tile, 128-query tile) from the same queues as the other grouped scans
 * (k_pair_offsets with 8 groups of 16 queries and 2 tiles of 64 rows per item).  256 threads = 4 waves; wave
 * (wq, wr) owns the 64 queries x 64 rows sub-tile as 2 x 2 MFMA blocks.
 *
 * LDS image of a 32-row (or 32-query) block for one 32-dimension chunk: [r][8 slots of 16 bytes], logical slot
 * s = 2 * kstep + khalf for the hi plane, 4 + 2 * kstep + khalf for the lo plane, stored at slot s ^ ((r >> 1) & 7):
 * the 16 lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) then hit 16 distinct 16-byte bank
 * slots.  The DMA writes LDS linearly in lane order (lane i -> byte 16 i of the instruction's 1 KiB), so the
 * swizzle is applied to the global address each lane reads: 8 rows x 128 contiguous bytes per instruction.
 * H16 rows: the mirror's own fp16 row is the hi plane (64 bytes per row and chunk, slot s ^ ((r >> 2) & 3)).
 *
 * Pipeline per chunk: wait for the chunk's DMA, one barrier, issue the next chunk's DMA into the other buffer
 * (which every wave has finished reading), then 8 ds_read_b128 + 12 MFMAs per k-step.  Every 64 dimensions the
 * block accumulator is added to the running sum and restarted from zero (ndbhip_common.h (4)).
 * ------------------------------------------------------------------------------------------------------------ */
template <int H16> struct S16Geom
{
	static constexpr int ROW_BLK = H16 ? 2048 : 4096;		/* bytes of a 32-row block per chunk */
	static constexpr int ROWS_BYTES = 4 * ROW_BLK;			/* 128 rows */
	static constexpr int Q_OFF = ROWS_BYTES;
	static constexpr int BUF = ROWS_BYTES + 4 * 4096;		/* + 128 queries x 128 bytes */
	static constexpr int ROW_CHUNK = H16 ? 64 : 128;		/* bytes of a row per chunk */
};

typedef __attribute__((address_space(3))) void *ndb_lds_ptr;
typedef const __attribute__((address_space(1))) void *ndb_glb_ptr;

__device__ __forceinline__ void
s16_dma16(const unsigned char *gp, unsigned char *lp)
{
	__builtin_amdgcn_global_load_lds((ndb_glb_ptr) gp, (ndb_lds_ptr) lp, 16, 0, 0);
}

template <int R, int H16>
__global__ __launch_bounds__(256, 2) void
k_s16_sweep(IvfDev ix, const unsigned char *__restrict__ planes, uint32_t rowbytes,
			const float *__restrict__ rn2, const int16_t *__restrict__ rexp,
			const unsigned char *__restrict__ qplanes, uint32_t qrowbytes, const float *__restrict__ qn2,
			const int *__restrict__ qexp, const float2 *__restrict__ qthr,
			const uint32_t *__restrict__ loc_cand_off, int npr, const uint32_t *__restrict__ cnt,
			const uint32_t *__restrict__ pair_off, const uint32_t *__restrict__ item_off,
			const PairRec *__restrict__ pairs, unsigned int *__restrict__ next_item,
			const uint32_t *__restrict__ runs, unsigned int *__restrict__ ecount, uint2 *__restrict__ erec,
			uint32_t ecap, uint32_t *__restrict__ bmin, int polite, int nchunk)
{
	typedef S16Geom<H16> G;
	__shared__ __attribute__((aligned(1024))) unsigned char bufA[G::BUF];
	__shared__ __attribute__((aligned(1024))) unsigned char bufB[G::BUF];
	__shared__ S16Q qinfo[S16_QT];
	__shared__ uint32_t s_item;
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const int	wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	wq = wave & 1, wr = wave >> 1;
	const int	r32 = lane & 31, kh = lane >> 5;

	for (uint32_t hop = 0; hop < 8; hop++)
	{
	const uint32_t xq = (blockIdx.x + hop) & 7u;
	const uint32_t run_lo = runs[xq], run_hi = runs[xq + 1];

	if (run_lo == run_hi)
		continue;
	for (;;)
	{
		if (tid == 0)
			s_item = (polite && run_lo + __hip_atomic_load(next_item + xq * NDB_QHEAD_STRIDE, __ATOMIC_RELAXED,
															__HIP_MEMORY_SCOPE_AGENT) >= run_hi)
				? run_hi : run_lo + atomicAdd(next_item + xq * NDB_QHEAD_STRIDE, 1u);
		__syncthreads();
		const uint32_t item = s_item;

		if (item >= run_hi)
		{
			__syncthreads();
			break;				/* uniform */
		}
		uint32_t	lo = 0, hi = (uint32_t) ix.ncent;

		while (hi - lo > 1)
		{
			const uint32_t mid = (lo + hi) >> 1;

			if (item_off[mid] <= item)
				lo = mid;
			else
				hi = mid;
		}
		while (lo + 1 < (uint32_t) ix.ncent && item_off[lo + 1] <= item)
			lo++;
		const uint32_t L = lo;
		const uint32_t len = ix.own_len[L];
		const uint32_t local = item - item_off[L];
		const uint32_t nmemL = cnt[L];
		const uint32_t nqt = (nmemL + S16_QT - 1) / S16_QT;
		const uint32_t qt = local % nqt;		/* query tile fastest: neighbours stream the same rows */
		const uint32_t t2 = local / nqt;		/* 128-row tile */
		const uint32_t nmem = min((uint32_t) S16_QT, nmemL - qt * S16_QT);
		const size_t row0 = (size_t) ix.loc_off[L] + (size_t) t2 * S16_RT;

		if (tid < S16_QT)
		{
			S16Q		qi;

			qi.q2 = 0.0f; qi.thrE = 0.0f; qi.eq = 0; qi.la = 0; qi.nrow = 0; qi.qid = 0;
			if ((uint32_t) tid < nmem)
			{
				const PairRec pr = pairs[pair_off[L] + qt * S16_QT + (uint32_t) tid];
				const uint32_t *lq = loc_cand_off + (size_t) pr.q * (npr + 1);

				qi.qid = pr.q;
				qi.la = lq[pr.p];
				qi.nrow = lq[pr.p + 1] - qi.la;
				qi.q2 = qn2[pr.q];
				qi.eq = qexp[pr.q];
				qi.thrE = qthr[pr.q].x;
			}
			qinfo[tid] = qi;
		}
		__syncthreads();		/* also: everybody has read s_item */

		/* DMA addresses of this lane: wave w stages 32-row block w and 32-query block w */
		uint32_t	voff_r[4], voff_q[4];

#pragma unroll
		for (int j = 0; j < 4; j++)
		{
			if constexpr (H16 != 0)
			{
				/* 16 rows x 4 slots per instruction; two instructions per 32-row block and chunk */
				const int	rr = 16 * (j & 1) + (lane >> 2);		/* j = 0, 1 only */
				const uint32_t ridx = t2 * S16_RT + (uint32_t) (32 * wave + rr);
				const uint32_t rc = ridx < len ? (uint32_t) (32 * wave + rr) : (len - 1 - t2 * S16_RT);

				voff_r[j] = rc * rowbytes + 16u * (uint32_t) ((lane & 3) ^ ((rr >> 2) & 3));
			}
			else
			{
				const int	rr = 8 * j + (lane >> 3);
				const uint32_t ridx = t2 * S16_RT + (uint32_t) (32 * wave + rr);
				const uint32_t rc = ridx < len ? (uint32_t) (32 * wave + rr) : (len - 1 - t2 * S16_RT);

				voff_r[j] = rc * rowbytes + 16u * (uint32_t) ((lane & 7) ^ ((rr >> 1) & 7));
			}
			{
				const int	rr = 8 * j + (lane >> 3);

				voff_q[j] = qinfo[32 * wave + rr].qid * qrowbytes + 16u * (uint32_t) ((lane & 7) ^ ((rr >> 1) & 7));
			}
		}
		const unsigned char *rbase = planes + row0 * (size_t) rowbytes;

		auto		issue = [&](int c, unsigned char *buf) {
			const unsigned char *rb = rbase + (size_t) c * G::ROW_CHUNK;
			const unsigned char *qb = qplanes + (size_t) c * 128;

#pragma unroll
			for (int j = 0; j < (H16 ? 2 : 4); j++)
				s16_dma16(rb + voff_r[j], buf + wave * G::ROW_BLK + j * 1024);
#pragma unroll
			for (int j = 0; j < 4; j++)
				s16_dma16(qb + voff_q[j], buf + G::Q_OFF + wave * 4096 + j * 1024);
		};

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
		/* fragment addresses (bytes inside a buffer) */
		const int	qsw = (r32 >> 1) & 7;
		const int	qfrag = G::Q_OFF + (2 * wq) * 4096 + r32 * 128;
		const int	rsw = H16 ? ((r32 >> 2) & 3) : ((r32 >> 1) & 7);
		const int	rfrag = (2 * wr) * G::ROW_BLK + r32 * G::ROW_CHUNK;

		auto		compute = [&](const unsigned char *buf) {
#pragma unroll
			for (int s = 0; s < 2; s++)
			{
				ndb_h8		ah[2], al[2], bh[2], bl[2];

#pragma unroll
				for (int b = 0; b < 2; b++)
				{
					ah[b] = *reinterpret_cast<const ndb_h8 *>(buf + qfrag + b * 4096 + (((2 * s + kh) ^ qsw) * 16));
					al[b] = *reinterpret_cast<const ndb_h8 *>(buf + qfrag + b * 4096 + (((4 + 2 * s + kh) ^ qsw) * 16));
					bh[b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * G::ROW_BLK + (((2 * s + kh) ^ rsw) * 16));
					if constexpr (H16 == 0)
						bl[b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * G::ROW_BLK + (((4 + 2 * s + kh) ^ rsw) * 16));
				}
#pragma unroll
				for (int a = 0; a < 2; a++)
#pragma unroll
					for (int b = 0; b < 2; b++)
					{
						blk[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], blk[a][b], 0, 0, 0);
						if constexpr (H16 == 0)
							blk[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], blk[a][b], 0, 0, 0);
						blk[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], blk[a][b], 0, 0, 0);
					}
			}
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

		issue(0, bufA);
		for (int c = 0; c < nchunk; c += 2)
		{
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			if (c + 1 < nchunk)
				issue(c + 1, bufB);
			compute(bufA);
			if (c + 1 >= nchunk)
			{
				flush();
				break;
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			if (c + 2 < nchunk)
				issue(c + 2, bufA);
			compute(bufB);
			flush();			/* NDB_S16_FLUSH_DIMS = 2 chunks */
		}

		/* epilogue: element (reg, lane) of block (a, b) = query 32 (2 wq + a) + (reg & 3) + 8 (reg >> 2) + 4 kh,
		 * row 32 (2 wr + b) + r32 */
#pragma unroll
		for (int b = 0; b < 2; b++)
		{
			const uint32_t ridx = t2 * S16_RT + (uint32_t) (32 * (2 * wr + b) + r32);
			const bool	rok = ridx < len;
			const size_t grow = (size_t) ix.loc_off[L] + (rok ? ridx : len - 1);
			const float x2 = rn2[grow];
			const int	ex = H16 ? 14 : (int) rexp[grow];

#pragma unroll
			for (int a = 0; a < 2; a++)
#pragma unroll
				for (int reg = 0; reg < 16; reg++)
				{
					const int	m = 32 * (2 * wq + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
					const S16Q	qi = qinfo[m];

					if (ridx < qi.nrow)
					{
						const float dot = ldexpf(run[a][b][reg], qi.eq + ex - 28);
						float		av;

						if (R == R_IVF_L2)
							av = __builtin_fmaf(-2.0f, dot, qi.q2 + x2);
						else
							av = -dot;
						if (!(av > qi.thrE))
						{
							const uint32_t slot = atomicAdd(&ecount[qi.qid], 1u);
							const uint32_t pos = qi.la + ridx, ab = __float_as_uint(av);

							if (slot < ecap)
								erec[(size_t) qi.qid * ecap + slot] = make_uint2(pos, ab);
							/* the smallest a of every hash bucket of positions, kept whether or not the record fit:
							 * k non-empty buckets are k distinct candidates (k_s16_retarget) */
							if ((ab & 0x7FFFFFFFu) < 0x7F800000u)
								atomicMin(&bmin[(size_t) qi.qid * S16_NB + ((pos * 2654435761u) >> (32 - S16_NB_LOG2))],
										  ndb_key_from_bits(ab));
						}
					}
				}
		}
		__syncthreads();		/* qinfo, s_item and the buffers are reused by the next item */
	}
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
			   unsigned int *__restrict__ flags)
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

		qthr[q] = make_float2(fminf(o.x, s16_thr_from_a<R>(__uint_as_float(tb), o.y, dim)), o.y);
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
template <int R, int H16>
__global__ __launch_bounds__(256) void
k_s16_finalize(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
			   const uint32_t *__restrict__ cand_off, const uint32_t *__restrict__ loc_cand_off, int npr,
			   uint32_t k, const float2 *__restrict__ qthr, const unsigned int *__restrict__ ecount,
			   const uint2 *__restrict__ erec, uint32_t ecap, int partial, ndbhip_cand *__restrict__ out_cand,
			   int *__restrict__ out_ncand, int64_t *__restrict__ out_total, uint64_t *__restrict__ out_tids,
			   float *__restrict__ out_dist, int *__restrict__ out_count, unsigned int *__restrict__ rec_counts,
			   unsigned int *__restrict__ flags)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	TopkSmem	s = carve_topk_smem(smem_raw, S16_SURV_CAP, k);
	uint32_t   *r_pos = (uint32_t *) (smem_raw + topk_smem_bytes(S16_SURV_CAP, k));
	uint32_t   *r_a = r_pos + ecap;
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
			atomicAdd(&flags[0], 1u);
			rec_counts[q] = 0;
		}
		return;					/* uniform */
	}
	for (uint32_t i = tid; i < nraw; i += 256)
	{
		const uint2 r = erec[(size_t) q * ecap + i];

		r_pos[i] = r.x;
		r_a[i] = r.y;
	}
	__syncthreads();
	float		thrE = qthr[q].x;
	const float e = qthr[q].y;

	if (nraw >= k)
	{
		uint32_t	T, m_less, kk0, cnt_eq;
		/* only finite values stand for a candidate whose distance is known to within E (a NaN or an infinity is
		 * what a row or a product beyond fp32 leaves behind: emitted, never counted) */
		auto		ld = [&](uint32_t i, uint32_t &bits) -> bool {
			bits = r_a[i];
			return (bits & 0x7FFFFFFFu) < 0x7F800000u;
		};

		block_radix_select(ld, nraw, k, s.hist, s.sh, T, m_less, kk0, cnt_eq);
		if (kk0 >= k)
		{
			/* T = order key of the k-th smallest a: back to the float */
			const uint32_t tb = (T & 0x80000000u) ? (T & 0x7FFFFFFFu) : ~T;

			thrE = fminf(thrE, s16_thr_from_a<R>(__uint_as_float(tb), e, dim));
		}
	}
	if (tid == 0)
		s.sh[0] = 0;
	__syncthreads();
	for (uint32_t i = tid; i < nraw; i += 256)
	{
		const float av = __uint_as_float(r_a[i]);

		if (!(av > thrE))
		{
			const uint32_t slot = atomicAdd(&s.sh[0], 1u);

			if (slot < S16_SURV_CAP)
				s.e_pos[slot] = r_pos[i];
		}
	}
	__syncthreads();
	const uint32_t ns = s.sh[0];

	__syncthreads();
	if (ns > S16_SURV_CAP)
	{
		if (tid == 0)
		{
			atomicAdd(&flags[0], 1u);
			rec_counts[q] = 0;
		}
		return;
	}
	if (tid == 0)
		rec_counts[q] = ns;
	/* the reference's arithmetic for every survivor: one lane per candidate */
	for (uint32_t j = tid; j < ns; j += 256)
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
	uint32_t	kk;
	const uint32_t npad = next_pow2(ns > 0 ? ns : 1);
	const uint32_t cut = block_sort_cut(s.e_bits, s.e_pos, ns, npad, k, partial ? (uint64_t) ns : (uint64_t) gtotal, s.fs, kk);

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

#endif							/* NDBHIP_SCREEN16_H */
