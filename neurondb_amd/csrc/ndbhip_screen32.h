/*
 * ndbhip_screen32.h — the fp32 screened list scan (part of ndbhip.hip's translation unit): a fused-multiply-add
 * bound pass over (list, row tile, query groups) items — on the vector ALU (k_ivf_bound_coop / _coop2) or on fp32
 * MFMA (k_ivf_bound_mfma) — into the [nq x candidates] distance buffer, k_ivf_survivors / k_ivf_rescore_list for
 * the candidates that can still matter.  Serves cosine, k > 64 and the batches the fp16 matrix-core screen
 * (ndbhip_screen16.h) hands back; docs/DESIGN_rounds_1_3.md 3c.
 */
#ifndef NDBHIP_SCREEN32_H
#define NDBHIP_SCREEN32_H

/* ------------------------------------------------------------------ */
/* Screened L2 scan (grouped path): see GAcc<R_SCR_L2>.                 */
/* ------------------------------------------------------------------ */
#define NDB_SCR_U 5.9604645e-8f		/* 2^-24 */

/* largest FINITE float of a non-negative array (bits order like values).  A row whose norm is NaN or infinite
 * must not reach the bound's constant: it would turn every query's E into NaN and with it every provisional
 * distance of the batch.  Such a row's own provisional distance is 0 (NaN) or inf, i.e. it is handed
 * to the reference's arithmetic or ordered last, like the exact scan does. */
__global__ void
k_max_nonneg(const float *__restrict__ v, int64_t n, uint32_t *__restrict__ out_bits)
{
	uint32_t	m = 0;

	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
	{
		const uint32_t b = __float_as_uint(v[i]);

		if ((b & 0x7F800000u) != 0x7F800000u)
			m = max(m, b & 0x7FFFFFFFu);
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		m = max(m, (uint32_t) __shfl_xor((int) m, off, 64));
	if ((threadIdx.x & 63) == 0)
		atomicMax(out_bits, m);
}

/* qe[q] = |q|^2 (already there), qe[nq + q] = E of query q: gamma_(dim+8) * 2 * (|q|^2 + max |x|^2), inflated by
 * 1 % for the rounding of the norms themselves, plus an absolute floor for underflow */
__global__ void
k_screen_eq(float *__restrict__ qe, uint32_t nq, int dim, const float *__restrict__ xxmax)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;

	if (q >= nq)
		return;
	const float nu = (float) (dim + 8) * NDB_SCR_U;
	const float gam = nu / (1.0f - nu);

	qe[nq + q] = gam * 2.02f * (qe[q] + *xxmax) + 1e-30f;
}

/*
 * Second pass of the screened scan.  lk = the k-th smallest provisional distance (a lower bound of that
 * candidate's distance; first-pass top-k).  With l = lk^2 the k-th smallest LOWER bound of the squared
 * distances, l + 2E is the k-th smallest UPPER bound, so the k-th smallest real squared distance is at most
 * l + 2E, the reference's k-th sequential sum T at most (l + 2E)(1 + gamma), and every candidate whose float4
 * distance can be <= the k-th float4 distance has a lower bound <= thr (slack m covers the sequential sum's own
 * rounding and the two sqrtf roundings).  k_ivf_survivors (one block per query) finds those candidates through
 * the tile minima — a few dozen per query — and lists them; k_ivf_rescore_list gives each the reference's own
 * arithmetic, one lane per candidate.  The rest keep their provisional value, which is above the k-th
 * distance.  Tile minima are recomputed over what the buffer then holds.
 */
struct ScrRec
{
	uint32_t	q, pos, row, slot;
};

template <int R>
__device__ __forceinline__ float
scr_exact(const float *__restrict__ qq, const float *__restrict__ x, int dim)
{
	Acc<R>		acc;
	int			i = 0;

	for (; i + 64 <= dim; i += 64)	/* 16 + 16 loads in flight, then the reference's chain */
	{
		float4		xv[16], qv[16];

#pragma unroll
		for (int u = 0; u < 16; u++)
		{
			xv[u] = *reinterpret_cast<const float4 *>(x + i + 4 * u);
			qv[u] = *reinterpret_cast<const float4 *>(qq + i + 4 * u);
		}
#pragma unroll
		for (int u = 0; u < 16; u++)
		{
			acc.step(qv[u].x, xv[u].x);
			acc.step(qv[u].y, xv[u].y);
			acc.step(qv[u].z, xv[u].z);
			acc.step(qv[u].w, xv[u].w);
		}
	}
	for (; i < dim; i++)
		acc.step(qq[i], x[i]);
	return acc.fin();
}

/* the same over an fp16 row (halfvec column): every element decoded like fp16_to_float (SUBFIX: with the Q20
 * subnormal quirk), then the reference's chain */
template <int R, bool SUBFIX>
__device__ __forceinline__ float
scr_exact_h(const float *__restrict__ qq, const uint16_t *__restrict__ x, int dim)
{
	Acc<R>		acc;

	for (int i = 0; i < dim; i += 8)	/* fp16 mirrors have dim % 64 == 0 */
	{
		const float4 raw = *reinterpret_cast<const float4 *>(x + i);
		const float4 q0 = *reinterpret_cast<const float4 *>(qq + i);
		const float4 q1 = *reinterpret_cast<const float4 *>(qq + i + 4);
		float		v[8];

		decode8<SUBFIX>(raw, v);
		acc.step(q0.x, v[0]);
		acc.step(q0.y, v[1]);
		acc.step(q0.z, v[2]);
		acc.step(q0.w, v[3]);
		acc.step(q1.x, v[4]);
		acc.step(q1.y, v[5]);
		acc.step(q1.z, v[6]);
		acc.step(q1.w, v[7]);
	}
	return acc.fin();
}

/* |x|^2 of every fp16 row, the sequential unfused chain over the decoded values (= the reference's norm2) */
template <bool SUBFIX>
__global__ void
k_row_norms_h(const uint16_t *__restrict__ vecs, int64_t nrows, int dim, float *__restrict__ out)
{
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;

	if (r >= nrows)
		return;
	const uint16_t *x = vecs + (size_t) r * dim;
	float		n2 = 0.0f;

	for (int i = 0; i < dim; i += 8)
	{
		const float4 raw = *reinterpret_cast<const float4 *>(x + i);
		float		v[8];

		decode8<SUBFIX>(raw, v);
#pragma unroll
		for (int u = 0; u < 8; u++)
			n2 = n2 + v[u] * v[u];
	}
	out[r] = n2;
}

template <int R, int H16>
__global__ __launch_bounds__(256) void
k_ivf_survivors(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
				const uint32_t *__restrict__ loc_cand_off, int npr, float *__restrict__ dist, uint32_t stride,
				uint32_t *__restrict__ tmin, uint32_t tstride, const float *__restrict__ qe, uint32_t nq, uint32_t k,
				const float *__restrict__ first_dist, const int *__restrict__ first_count,
				ScrRec *__restrict__ recs_all, uint32_t rec_cap, unsigned int *__restrict__ rec_counts,
				unsigned long long *__restrict__ counters)
{
	/* the query's own slice of the list and an LDS counter: one global counter for all blocks would serialise */
	__shared__ unsigned int s_count;
	const uint32_t q = blockIdx.x;
	ScrRec	   *recs = recs_all + (size_t) q * rec_cap;

	if (threadIdx.x == 0)
		s_count = 0;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t *lco = loc_cand_off + (size_t) q * (npr + 1);
	const int	dim = ix.dim;
	float		thr = FLT_MAX;

	if (first_count[q] >= (int) k)
	{
		const float lk = first_dist[(size_t) q * k + (k - 1)];
		const float e = qe[nq + q];
		const float m = (float) (16 * dim + 64) * NDB_SCR_U;

		if (R == R_IVF_L2)
		{
			const float l2 = lk * lk * 1.0000039f;		/* undo the kernel's round-down (2^-20) and sqrtf's */
			const float t2 = (l2 + 2.0f * e) * (1.0f + m);

			thr = __builtin_sqrtf(t2) * 1.000001f;
		}
		else
		{
			/* inner product / cosine: the k-th smallest lower bound + 2E is the k-th smallest upper bound; the
			 * values are signed, so the slack is absolute as well as relative */
			const float e2 = (R == R_IVF_COS) ? 4.0f * ((float) (dim + 8) * NDB_SCR_U) / (1.0f - (float) (dim + 8) * NDB_SCR_U) : e;
			const float u = lk + 2.0f * e2;

			thr = u + (fabsf(u) + fabsf(lk) + 2.0f * e2) * 2e-6f + 1e-36f;
		}
	}
	const uint32_t kthr = ndb_key_from_bits(__float_as_uint(thr));
	uint32_t   *tm = tmin + (size_t) q * tstride;
	uint32_t	unit = 0;

	/* units of 64 tile slots, dealt to the block's 4 waves in turn */
	for (int pp = 0; pp < npr; pp++)
	{
		const uint32_t la = lco[pp], nrow = lco[pp + 1] - la;
		const uint32_t ntile = (nrow + 63u) >> 6;

		for (uint32_t tbase = 0; tbase < ntile; tbase += 64, unit++)
		{
			if ((unit & 3u) != wave)
				continue;
			const uint32_t tt = tbase + lane;
			unsigned long long hits = __ballot(tt < ntile && tm[(la >> 6) + pp + tt] <= kthr);

			while (hits)
			{
				const uint32_t t = tbase + (uint32_t) (__ffsll((long long) hits) - 1);

				hits &= hits - 1ull;
				const uint32_t ridx = t * 64 + lane;
				const bool	valid = ridx < nrow;
				float	   *dp = dist + (size_t) q * stride + la + ridx;
				float		v = valid ? *dp : FLT_MAX;
				const bool	surv = valid && v <= thr;
				const unsigned long long sm = __ballot(surv);
				const int	L = probes[(size_t) q * npr + pp];
				const uint32_t row = (uint32_t) ix.loc_off[L] + ridx;
				const uint32_t slot = (la >> 6) + (uint32_t) pp + t;
				uint32_t	base = 0;

				if (lane == 0 && sm)
					base = atomicAdd(&s_count, (unsigned int) __popcll(sm));
				base = __shfl(base, 0, 64);
				if (surv)
				{
					const uint32_t at = base + (uint32_t) __popcll(sm & ((1ull << lane) - 1ull));

					if (at < rec_cap)
					{
						ScrRec		r;

						r.q = q; r.pos = la + ridx; r.row = row; r.slot = slot;
						recs[at] = r;
						v = FLT_MAX;	/* its exact value is min-ed into the tile by k_ivf_rescore_list */
					}
					else
					{
						/* list full: do it here */
						if constexpr (H16 != 0)
							v = scr_exact_h<R, H16 == 1>(queries + (size_t) q * dim,
														 (const uint16_t *) ix.vecs + (size_t) row * (size_t) dim, dim);
						else
							v = scr_exact<R>(queries + (size_t) q * dim, ix.vecs + (size_t) row * (size_t) dim, dim);
						*dp = v;
					}
				}
				if (counters && base + (uint32_t) __popcll(sm) > rec_cap)	/* wave-uniform */
				{
					const uint32_t first_over = base > rec_cap ? base : rec_cap;

					if (lane == 0)
						atomicAdd(&counters[3], (unsigned long long) (base + (uint32_t) __popcll(sm) - first_over));
				}
				/* the tile's minimum over what stays as it is */
				uint32_t	mk = (valid && v != FLT_MAX) ? ndb_key_from_bits(__float_as_uint(v)) : 0xFFFFFFFFu;

#pragma unroll
				for (int off = 32; off > 0; off >>= 1)
					mk = min(mk, (uint32_t) __shfl_xor((int) mk, off, 64));
				if (lane == 0)
					tm[slot] = mk;
			}
		}
	}
	__syncthreads();
	if (threadIdx.x == 0)
		rec_counts[q] = min(s_count, rec_cap);
}

/* one lane per listed candidate: the reference's arithmetic, the value into the distance buffer and into its
 * tile's minimum */
/* counters[3] += sum of v[0..n): any number of blocks, one atomic each (a single block of 256 threads read 4096 values
 * in 16 dependent trips: 6 us of a 1 ms step for a statistic) */
__global__ __launch_bounds__(256) void
k_sum_u32(const unsigned int *__restrict__ v, uint32_t n, unsigned long long *__restrict__ out)
{
	__shared__ unsigned long long part[4];
	unsigned long long s = 0;

	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
		s += v[i];
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		const uint32_t lo = __shfl_xor((uint32_t) s, off, 64);
		const uint32_t hi = __shfl_xor((uint32_t) (s >> 32), off, 64);

		s += ((unsigned long long) hi << 32) | lo;
	}
	if ((threadIdx.x & 63) == 0)
		part[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0 && out)
		atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

template <int R, int H16>
__global__ __launch_bounds__(64) void
k_ivf_rescore_list(IvfDev ix, const float *__restrict__ queries, float *__restrict__ dist, uint32_t stride,
				   uint32_t *__restrict__ tmin, uint32_t tstride, const ScrRec *__restrict__ recs, uint32_t rec_cap,
				   const unsigned int *__restrict__ rec_counts, unsigned long long *__restrict__ counters)
{
	const uint32_t q = blockIdx.y;
	const uint32_t n = rec_counts[q];
	const uint32_t i = blockIdx.x * 64 + threadIdx.x;

	(void) counters;			/* counted by k_sum_u32: one atomic per launch, not one per query on one line */
	if (i >= n)
		return;
	const ScrRec r = recs[(size_t) q * rec_cap + i];
	float		v;

	if constexpr (H16 != 0)
		v = scr_exact_h<R, H16 == 1>(queries + (size_t) r.q * ix.dim,
									 (const uint16_t *) ix.vecs + (size_t) r.row * (size_t) ix.dim, ix.dim);
	else
		v = scr_exact<R>(queries + (size_t) r.q * ix.dim, ix.vecs + (size_t) r.row * (size_t) ix.dim, ix.dim);

	dist[(size_t) r.q * stride + r.pos] = v;
	atomicMin(&tmin[(size_t) r.q * tstride + r.slot], ndb_key_from_bits(__float_as_uint(v)));
}

/* a row piece fetched outside the compiler's view: hipcc drains every outstanding vector load in front of each
 * `asm volatile` of the query stream, so a C++ load issued ahead of the arithmetic is waited for at once; this
 * one is only waited for where ndb_gwait says so */
typedef float ndb_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void
ndb_gload4(ndb_f4 &v, const float *p)
{
	asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}

__device__ __forceinline__ void
ndb_gwait(ndb_f4 &v)
{
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory");
}

/*
 * The bound pass of the screened scan, cooperative form: one 256-thread block per (list, 64-row tile, FOUR
 * consecutive query groups).  The four waves of the block score the same rows for four different groups, so
 * a 16-float chunk of the tile is fetched once — one float4 per thread — into a double-buffered LDS tile and
 * consumed by all four; with single-wave blocks the sibling waves drift apart over the 48 chunks of an item and
 * the lines the first one brought in are gone when the others arrive (a row tile came from HBM ~4 times per
 * batch).  Everything else — work queues, query stream through SGPRs, epilogue — is k_ivf_scan_grouped's.
 */
__global__ __launch_bounds__(256, 8) void
k_ivf_bound_coop(IvfDev ix, const float *__restrict__ qblock, const uint32_t *__restrict__ loc_cand_off, int npr,
				 const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ pair_off,
				 const uint32_t *__restrict__ item_off, const uint32_t *__restrict__ grp_off,
				 const PairRec *__restrict__ pairs, unsigned int *__restrict__ next_item,
				 const uint32_t *__restrict__ runs, float *__restrict__ dist, uint32_t stride,
				 const float *__restrict__ qnorm, uint32_t *__restrict__ tmin, uint32_t tstride, int polite,
				 uint32_t nq_all)
{
	constexpr int CH = 16;
	__shared__ __attribute__((aligned(16))) float tile[2][64 * CH];
	__shared__ uint32_t s_item;
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	dim = ix.dim;
	const int	srow = tid >> 2, sslot = tid & 3;	/* staging: thread = (row of the tile, 16-byte slot) */

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

		__syncthreads();		/* everybody has read it before thread 0 can write the next one */
		if (item >= run_hi)
			break;				/* uniform: every thread leaves */
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
		const uint32_t ngrp = (cnt[L] + NDB_QG - 1) / NDB_QG;
		const uint32_t nquad = (ngrp + 3u) >> 2;
		const uint32_t quad = local % nquad;
		const uint32_t t = local / nquad;
		const uint32_t gi = quad * 4u + wave;
		const bool	active = gi < ngrp;		/* wave-uniform */
		const uint32_t g0 = (active ? gi : 0u) * NDB_QG;
		const uint32_t nmem = active ? min((uint32_t) NDB_QG, cnt[L] - g0) : 0u;
		const PairRec *mem = pairs + pair_off[L] + g0;
		const float *__restrict__ qb = qblock + (size_t) (grp_off[L] + (active ? gi : 0u)) * (size_t) dim * NDB_QG;
		const uint32_t ridx = t * 64 + lane;
		const uint32_t sr = t * 64 + (uint32_t) srow;
		const float *srcrow = ix.vecs + ((size_t) ix.loc_off[L] + (sr < len ? sr : len - 1)) * (size_t) dim +
			((sslot ^ tile_swz<CH>(srow)) * 4);
		GAcc<R_SCR_L2> acc;

		acc.init();
		const float *qs = qb;
		ndb_f16		qa0, qa1, qb0, qb1;

		asm volatile("s_nop 4" ::: "memory");
		if (active)
			sload2x16(qa0, qa1, qs);
		ndb_f4		st;

		ndb_gload4(st, srcrow);
		for (int c = 0; c < dim; c += CH)
		{
			float	   *tb = tile[(c / CH) & 1];

			ndb_gwait(st);
			*reinterpret_cast<ndb_f4 *>(tb + srow * CH + sslot * 4) = st;
			__syncthreads();
			if (c + CH < dim)
				ndb_gload4(st, srcrow + c + CH);	/* in flight while this chunk is consumed */
			if (active)
			{
				float4		x[CH / 4];

#pragma unroll
				for (int p = 0; p < CH / 4; p++)
					x[p] = *reinterpret_cast<const float4 *>(tb + lane * CH + ((p ^ tile_swz<CH>(lane)) * 4));
				const float *qnext = (c + CH >= dim) ? qs - 2 * NDB_QG : qs;

				ndb_static_for<0, CH / 4>([&](auto pc) {
					constexpr int p = decltype(pc)::value;

					swait2(qa0, qa1);
					sload2x16_at<(4 * p + 2) * 64>(qb0, qb1, qs);
					acc.step(qa0, x[p].x);
					acc.step(qa1, x[p].y);
					swait2(qb0, qb1);
					if constexpr (p == CH / 4 - 1)
						sload2x16_at<CH * 64>(qa0, qa1, qnext);
					else
						sload2x16_at<(4 * p + 4) * 64>(qa0, qa1, qs);
					acc.step(qb0, x[p].z);
					acc.step(qb1, x[p].w);
				});
				qs += CH * NDB_QG;
			}
			/* double-buffered tile: the barrier of the next chunk keeps any wave from running two chunks ahead */
		}
		if (active)
		{
			swait2(qa0, qa1);
#pragma unroll
			for (int j = 0; j < NDB_QG; j++)
			{
				if ((uint32_t) j < nmem)
				{
					const uint32_t qid = mem[j].q;
					const uint32_t pp = mem[j].p;
					const uint32_t *lq = loc_cand_off + (size_t) qid * (npr + 1);
					const uint32_t la = lq[pp];
					const uint32_t nrow = lq[pp + 1] - la;
					const float dv = acc.bound(j, qnorm[qid], qnorm[nq_all + qid]);

					if (ridx < nrow)
						dist[(size_t) qid * stride + la + ridx] = dv;
					uint32_t	mk = ridx < nrow ? ndb_key_from_bits(__float_as_uint(dv)) : 0xFFFFFFFFu;

#pragma unroll
					for (int off = 32; off > 0; off >>= 1)
						mk = min(mk, (uint32_t) __shfl_xor((int) mk, off, 64));
					if (lane == 0 && t * 64u < nrow)
						tmin[(size_t) qid * tstride + (la >> 6) + pp + t] = mk;
				}
			}
		}
		__syncthreads();		/* s_item and the tile are reused by the next item */
	}
	}
}

/*
 * The same with TWO 64-row tiles per item (128 rows x 4 query groups per block): a lane scores rows r and r + 64
 * against the same query values, so the query stream — as many bytes per item as the row tile itself with one
 * tile per item, and re-read for every tile of the list — is fetched half as often, and the scalar loads per
 * vector instruction halve.
 */
/* four halfs (one 16-byte slot of decoded floats) of an fp16 row */
template <bool SUBFIX>
__device__ __forceinline__ float4
ndb_decode4(const uint16_t *p)
{
	const uint2 raw = *reinterpret_cast<const uint2 *>(p);
	float4		o;

	if (SUBFIX)
	{
		o.x = h2f_ref(raw.x & 0xFFFFu);
		o.y = h2f_ref(raw.x >> 16);
		o.z = h2f_ref(raw.y & 0xFFFFu);
		o.w = h2f_ref(raw.y >> 16);
	}
	else
	{
		o.x = __half2float(__ushort_as_half((unsigned short) (raw.x & 0xFFFFu)));
		o.y = __half2float(__ushort_as_half((unsigned short) (raw.x >> 16)));
		o.z = __half2float(__ushort_as_half((unsigned short) (raw.y & 0xFFFFu)));
		o.w = __half2float(__ushort_as_half((unsigned short) (raw.y >> 16)));
	}
	return o;
}

template <int R, int H16>
__global__ __launch_bounds__(256, NDB_COOP2_WAVES) void
k_ivf_bound_coop2(IvfDev ix, const float *__restrict__ qblock, const uint32_t *__restrict__ loc_cand_off, int npr,
				 const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ pair_off,
				 const uint32_t *__restrict__ item_off, const uint32_t *__restrict__ grp_off,
				 const PairRec *__restrict__ pairs, unsigned int *__restrict__ next_item,
				 const uint32_t *__restrict__ runs, float *__restrict__ dist, uint32_t stride,
				 const float *__restrict__ qnorm, uint32_t *__restrict__ tmin, uint32_t tstride, int polite,
				 uint32_t nq_all, const float *__restrict__ rnorm)
{
	constexpr int CH = 16;
	__shared__ __attribute__((aligned(16))) float tile[2][128 * CH];
	__shared__ uint32_t s_item;
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	dim = ix.dim;
	const int	srow = tid >> 2, sslot = tid & 3;	/* staging: thread = (row of the tile, 16-byte slot) */

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

		__syncthreads();		/* everybody has read it before thread 0 can write the next one */
		if (item >= run_hi)
			break;				/* uniform: every thread leaves */
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
		const uint32_t ngrp = (cnt[L] + NDB_QG - 1) / NDB_QG;
		const uint32_t nquad = (ngrp + 3u) >> 2;
		const uint32_t quad = local % nquad;
		const uint32_t t2 = local / nquad;		/* 128-row tile */
		const uint32_t gi = quad * 4u + wave;
		const bool	active = gi < ngrp;		/* wave-uniform */
		const uint32_t g0 = (active ? gi : 0u) * NDB_QG;
		const uint32_t nmem = active ? min((uint32_t) NDB_QG, cnt[L] - g0) : 0u;
		const PairRec *mem = pairs + pair_off[L] + g0;
		const float *__restrict__ qb = qblock + (size_t) (grp_off[L] + (active ? gi : 0u)) * (size_t) dim * NDB_QG;
		const uint32_t sr0 = t2 * 128 + (uint32_t) srow, sr1 = sr0 + 64;
		const int	spiece = (sslot ^ tile_swz<CH>(srow)) * 4;
		const float *src0 = ix.vecs + ((size_t) ix.loc_off[L] + (sr0 < len ? sr0 : len - 1)) * (size_t) dim + spiece;
		const float *src1 = ix.vecs + ((size_t) ix.loc_off[L] + (sr1 < len ? sr1 : len - 1)) * (size_t) dim + spiece;
		const uint16_t *h0 = (const uint16_t *) ix.vecs + ((size_t) ix.loc_off[L] + (sr0 < len ? sr0 : len - 1)) * (size_t) dim + spiece;
		const uint16_t *h1 = (const uint16_t *) ix.vecs + ((size_t) ix.loc_off[L] + (sr1 < len ? sr1 : len - 1)) * (size_t) dim + spiece;
		GAcc<R_SCR_L2> acc0, acc1;

		acc0.init();
		acc1.init();
		const float *qs = qb;
		ndb_f16		qa0, qa1, qb0, qb1;

		asm volatile("s_nop 4" ::: "memory");
		/* the query stream is primed per chunk, not carried over the loop edge: the compiler copies loop-carried
		 * registers at the edge, and a copy of a register an asm load is still filling copies garbage (this is
		 * what broke the first version of this kernel; tools/check_asm_hazards.py finds it in the ISA) */
		for (int c = 0; c < dim; c += CH)
		{
			float	   *tb = tile[(c / CH) & 1];
			/* plain loads: 5 waves per SIMD hide them, and nothing asm-loaded then lives across the loop edge */
			float4		st0, st1;

			if constexpr (H16 != 0)
			{
				/* fp16 rows: the slot's four halfs, decoded like fp16_to_float here; from LDS on it is the float4 path */
				st0 = ndb_decode4<H16 == 1>(h0 + c);
				st1 = ndb_decode4<H16 == 1>(h1 + c);
			}
			else
			{
				st0 = *reinterpret_cast<const float4 *>(src0 + c);
				st1 = *reinterpret_cast<const float4 *>(src1 + c);
			}

			*reinterpret_cast<float4 *>(tb + srow * CH + sslot * 4) = st0;
			*reinterpret_cast<float4 *>(tb + (64 + srow) * CH + sslot * 4) = st1;
			__syncthreads();
			if (active)
			{
				sload2x16(qa0, qa1, qs);
				ndb_static_for<0, CH / 4>([&](auto pc) {
					constexpr int p = decltype(pc)::value;
					const float4 x0 = *reinterpret_cast<const float4 *>(tb + lane * CH + ((p ^ tile_swz<CH>(lane)) * 4));
					const float4 x1 = *reinterpret_cast<const float4 *>(tb + (64 + lane) * CH + ((p ^ tile_swz<CH>(lane)) * 4));

					swait2(qa0, qa1);
					sload2x16_at<(4 * p + 2) * 64>(qb0, qb1, qs);
					acc0.step_dot(qa0, x0.x);
					acc1.step_dot(qa0, x1.x);
					acc0.step_dot(qa1, x0.y);
					acc1.step_dot(qa1, x1.y);
					swait2(qb0, qb1);
					if constexpr (p < CH / 4 - 1)
						sload2x16_at<(4 * p + 4) * 64>(qa0, qa1, qs);
					acc0.step_dot(qb0, x0.z);
					acc1.step_dot(qb0, x1.z);
					acc0.step_dot(qb1, x0.w);
					acc1.step_dot(qb1, x1.w);
				});
				qs += CH * NDB_QG;
			}
			/* double-buffered tile: the barrier of the next chunk keeps any wave from running two chunks ahead */
		}
		if (active)
		{
			/* |x|^2 of this lane's two rows: the exact kernel's sequential sum against a zero query, kept per row
			 * (relative error gamma_dim, like the fused chain it replaces) */
			const uint32_t r0 = t2 * 128 + lane, r1 = r0 + 64;
			const float rn0 = rnorm[(size_t) ix.loc_off[L] + (r0 < len ? r0 : len - 1)];
			const float rn1 = rnorm[(size_t) ix.loc_off[L] + (r1 < len ? r1 : len - 1)];

#pragma unroll
			for (int j = 0; j < NDB_QG; j++)
			{
				if ((uint32_t) j < nmem)
				{
					const uint32_t qid = mem[j].q;
					const uint32_t pp = mem[j].p;
					const uint32_t *lq = loc_cand_off + (size_t) qid * (npr + 1);
					const uint32_t la = lq[pp];
					const uint32_t nrow = lq[pp + 1] - la;
					const float qn = qnorm[qid], qe = qnorm[nq_all + qid];

#pragma unroll
					for (int u = 0; u < 2; u++)
					{
						const uint32_t t = t2 * 2u + (uint32_t) u;
						const uint32_t ridx = t * 64 + lane;
						float		dv;

						if (R == R_IVF_L2)
							dv = u ? acc1.bound_n2(j, qn, qe, rn1) : acc0.bound_n2(j, qn, qe, rn0);
						else if (R == R_IVF_IP)
							dv = u ? acc1.bound_ip(j, qe) : acc0.bound_ip(j, qe);
						else
						{
							const float nu = (float) (dim + 8) * NDB_SCR_U;
							const float ec = 4.0f * nu / (1.0f - nu);

							dv = u ? acc1.bound_cos(j, qn, rn1, ec) : acc0.bound_cos(j, qn, rn0, ec);
						}

						if (ridx < nrow)
							dist[(size_t) qid * stride + la + ridx] = dv;
						uint32_t	mk = ridx < nrow ? ndb_key_from_bits(__float_as_uint(dv)) : 0xFFFFFFFFu;

#pragma unroll
						for (int off = 32; off > 0; off >>= 1)
							mk = min(mk, (uint32_t) __shfl_xor((int) mk, off, 64));
						if (lane == 0 && t * 64u < nrow)
							tmin[(size_t) qid * tstride + (la >> 6) + pp + t] = mk;
					}
				}
			}
		}
		__syncthreads();		/* s_item and the tile are reused by the next item */
	}
	}
}

/*
 * The bound pass on the matrix cores: v_mfma_f32_32x32x2_f32.  Its numerics are a k-ordered f32 fmaf chain per
 * output element (one rounding per product-and-add, no wider accumulator: cdna_hip_programming.md, "FP32-input
 * MFMA") — the kind of chain GAcc::step_dot runs (dot = fma(q_d, x_d, dot)), in a permuted dimension order
 * (below) — so the error term E_q of the screened scan, which holds for any order of the dim fused
 * multiply-adds, holds unchanged.  What changes is who does the work: one MFMA (64 cycles of the
 * matrix pipe, two operand registers) replaces 1024 packed FMAs' worth of issue slots, operand moves and
 * scalar-load waits.
 *
 * Same items as the two-tile kernel: (list, 128 rows, four query groups = 64 queries).  Wave w scores the
 * 32 queries of groups {2(w&1), 2(w&1)+1} against the 64 rows of tile half (w>>1): A = queries (M = 32),
 * B = rows (N = 32, two blocks), so a result's column — the lane — is the row and a half-wave stores 128
 * contiguous bytes of a query's distance array.  Rows are staged through LDS in 16-dimension chunks (a
 * lane's eight values of a chunk are two 16-byte reads); the query values come straight from the
 * [group][dim][16] block, one dword per lane and step; both are fetched two chunks ahead.
 */
typedef float ndb_f16acc __attribute__((ext_vector_type(16)));

#ifndef NDB_MFMA_BLOCKS
#define NDB_MFMA_BLOCKS 4		/* measured per 4096 queries: 2 -> 9.5 ms, 3 -> 8.7 ms, 4 -> 8.6 ms */
#endif

__device__ __forceinline__ float
scr_bound_l2(float dot, float qn, float e, float rn2)
{
	const float a = (qn + rn2) - 2.0f * dot;
	const float l = a - e;

	return __builtin_sqrtf(fmaxf(l, 0.0f) * 0.99999905f);
}
__device__ __forceinline__ float
scr_bound_ip(float dot, float e)
{
	const float l = -dot - e;

	return l - fabsf(l) * 2.4e-7f - 1e-37f;
}
__device__ __forceinline__ float
scr_bound_cos(float dot, float qn, float rn2, float e)
{
	const float a = __builtin_sqrtf(qn), b = __builtin_sqrtf(rn2);
	const float c = (a == 0.0f || b == 0.0f) ? 1.0f : 1.0f - (dot / (a * b));
	const float l = c - e;

	return l - fabsf(l) * 2.4e-7f - 1e-37f;
}

template <int R, int H16>
__global__ __launch_bounds__(256, NDB_MFMA_BLOCKS) void
k_ivf_bound_mfma(IvfDev ix, const float *__restrict__ qblock, const uint32_t *__restrict__ loc_cand_off, int npr,
				 const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ pair_off,
				 const uint32_t *__restrict__ item_off, const uint32_t *__restrict__ grp_off,
				 const PairRec *__restrict__ pairs, unsigned int *__restrict__ next_item,
				 const uint32_t *__restrict__ runs, float *__restrict__ dist, uint32_t stride,
				 const float *__restrict__ qnorm, uint32_t *__restrict__ tmin, uint32_t tstride, int polite,
				 uint32_t nq_all, const float *__restrict__ rnorm)
{
	constexpr int CH = 16;
	__shared__ __attribute__((aligned(16))) float tile[2][128 * CH];
	__shared__ uint32_t s_item;
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t rhalf = wave >> 1;
	const int	kh = lane >> 5, ln = lane & 31;
	const int	dim = ix.dim;
	const int	srow = tid >> 2, sslot = tid & 3;	/* staging: thread = (row of the tile, four dimensions) */
	/* Within a 16-dimension chunk, step s of the MFMA sequence multiplies dimension s (k = 0, lanes 0-31) and
	 * dimension 8 + s (k = 1, lanes 32-63): a lane's eight values are 32 contiguous bytes of the row, the tile
	 * keeps the row's natural layout and staging is a straight 16-byte copy.  The chain of an output element
	 * then runs 0, 8, 1, 9, ... instead of 0, 1, 2, ...: a different order of the same fused multiply-adds, to
	 * which the error term applies unchanged (gamma_n bounds recursive summation in any order).  16-byte slots
	 * are XOR-swizzled by the row so that 16 consecutive rows reading one logical slot cover all 64 banks */
	const int	woff = srow * CH + ((sslot ^ ((srow >> 2) & 3)) * 4);	/* rows srow and srow + 64 share the swizzle */

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

		__syncthreads();		/* everybody has read it before thread 0 can write the next one */
		if (item >= run_hi)
			break;				/* uniform: every thread leaves */
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
		const uint32_t ngrp = (nmemL + NDB_QG - 1) / NDB_QG;
		const uint32_t nquad = (ngrp + 3u) >> 2;
		const uint32_t quad = local % nquad;
		const uint32_t t2 = local / nquad;		/* 128-row tile */
		/* which waves take the quad's upper two groups alternates from item to item: a quad with one or two
		 * groups leaves two waves without work, and wave i of every block runs on SIMD i — always idling the
		 * same two SIMDs would leave the other two as the bottleneck of the four blocks that share the CU */
		const uint32_t qhalf = (wave ^ t2 ^ L) & 1u;
		const uint32_t gw = quad * 4u + qhalf * 2u;	/* this wave's first group */
		const bool	active = gw < ngrp;		/* wave-uniform */
		/* the lane's query column of A: group gw + (ln >> 4), member ln & 15; a missing second group reads the
		 * first one again (its results are not stored) */
		const uint32_t ga = (active && gw + (uint32_t) (ln >> 4) < ngrp) ? gw + (uint32_t) (ln >> 4) : (active ? gw : 0u);
		const float *__restrict__ qp = qblock + (size_t) (grp_off[L] + ga) * (size_t) dim * NDB_QG + (ln & 15) + kh * 8 * NDB_QG;
		const uint32_t sr0 = t2 * 128 + (uint32_t) srow, sr1 = sr0 + 64;
		const float *src0 = ix.vecs + ((size_t) ix.loc_off[L] + (sr0 < len ? sr0 : len - 1)) * (size_t) dim + sslot * 4;
		const float *src1 = ix.vecs + ((size_t) ix.loc_off[L] + (sr1 < len ? sr1 : len - 1)) * (size_t) dim + sslot * 4;
		const uint16_t *h0 = (const uint16_t *) ix.vecs + ((size_t) ix.loc_off[L] + (sr0 < len ? sr0 : len - 1)) * (size_t) dim + sslot * 4;
		const uint16_t *h1 = (const uint16_t *) ix.vecs + ((size_t) ix.loc_off[L] + (sr1 < len ? sr1 : len - 1)) * (size_t) dim + sslot * 4;
		ndb_f16acc	acc0, acc1;

#pragma unroll
		for (int i = 0; i < 16; i++)
		{
			acc0[i] = 0.0f;
			acc1[i] = 0.0f;
		}
		/* Two register sets rotate (the chunk loop is unrolled by two): the rows of chunk c + 2 are fetched while
		 * chunk c is multiplied and the set fetched one chunk earlier is written to LDS, so a row fetch has two
		 * chunks of MFMAs to arrive; the query values of chunk c + 2 go into the registers chunk c has just
		 * used.  Nothing is copied between the sets: a copy would wait for its load. */
		float4		sa0, sa1, sb0, sb1;
		float		qa[CH / 2], qb[CH / 2];
		const int	clast = dim - CH;

		auto fetch_rows = [&](int c, float4 &st0, float4 &st1) {
			if constexpr (H16 != 0)
			{
				st0 = ndb_decode4<H16 == 1>(h0 + c);
				st1 = ndb_decode4<H16 == 1>(h1 + c);
			}
			else
			{
				st0 = *reinterpret_cast<const float4 *>(src0 + c);
				st1 = *reinterpret_cast<const float4 *>(src1 + c);
			}
		};
		auto store_rows = [&](float *tb, const float4 &st0, const float4 &st1) {
			*reinterpret_cast<float4 *>(tb + woff) = st0;
			*reinterpret_cast<float4 *>(tb + 64 * CH + woff) = st1;
		};
		auto load_q = [&](int c, float (&q)[CH / 2]) {
#pragma unroll
			for (int s = 0; s < CH / 2; s++)
				q[s] = qp[(size_t) (c + s) * NDB_QG];
		};
		const int	r0 = (int) rhalf * 64 + ln, r1 = r0 + 32;
		const int	ro0a = r0 * CH + (((kh * 2) ^ ((r0 >> 2) & 3)) * 4), ro0b = r0 * CH + (((kh * 2 + 1) ^ ((r0 >> 2) & 3)) * 4);
		const int	ro1a = r1 * CH + (((kh * 2) ^ ((r1 >> 2) & 3)) * 4), ro1b = r1 * CH + (((kh * 2 + 1) ^ ((r1 >> 2) & 3)) * 4);
		auto multiply = [&](const float *tb, const float (&q)[CH / 2]) {
			const float4 xa0 = *reinterpret_cast<const float4 *>(tb + ro0a);
			const float4 xb0 = *reinterpret_cast<const float4 *>(tb + ro0b);
			const float4 xa1 = *reinterpret_cast<const float4 *>(tb + ro1a);
			const float4 xb1 = *reinterpret_cast<const float4 *>(tb + ro1b);
			const float x0[8] = {xa0.x, xa0.y, xa0.z, xa0.w, xb0.x, xb0.y, xb0.z, xb0.w};
			const float x1[8] = {xa1.x, xa1.y, xa1.z, xa1.w, xb1.x, xb1.y, xb1.z, xb1.w};

#pragma unroll
			for (int s = 0; s < CH / 2; s++)
			{
				acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q[s], x0[s], acc0, 0, 0, 0);
				acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q[s], x1[s], acc1, 0, 0, 0);
			}
		};

		/* chunk indices past the end fetch the last chunk again (never multiplied) instead of branching */
		fetch_rows(0, sa0, sa1);
		load_q(0, qa);
		fetch_rows(min(CH, clast), sb0, sb1);
		load_q(min(CH, clast), qb);
		store_rows(tile[0], sa0, sa1);
		__syncthreads();
		for (int c = 0; c < dim; c += 2 * CH)
		{
			/* even chunk c: tile[0]; set a refills with chunk c + 2, set b (chunk c + 1) goes to tile[1] */
			fetch_rows(min(c + 2 * CH, clast), sa0, sa1);
			if (active)
				multiply(tile[0], qa);
			load_q(min(c + 2 * CH, clast), qa);
			store_rows(tile[1], sb0, sb1);
			__syncthreads();
			if (c + CH >= dim)
				break;			/* uniform: an odd number of chunks */
			/* odd chunk c + 1: tile[1]; set b refills with chunk c + 3, set a (chunk c + 2) goes to tile[0] */
			fetch_rows(min(c + 3 * CH, clast), sb0, sb1);
			if (active)
				multiply(tile[1], qb);
			load_q(min(c + 3 * CH, clast), qb);
			store_rows(tile[0], sa0, sa1);
			__syncthreads();
		}
		if (active)
		{
			const uint32_t rb = t2 * 128 + rhalf * 64 + (uint32_t) ln;	/* row of acc0; acc1: + 32 */
			const float rn0 = rnorm[(size_t) ix.loc_off[L] + (rb < len ? rb : len - 1)];
			const float rn1 = rnorm[(size_t) ix.loc_off[L] + (rb + 32 < len ? rb + 32 : len - 1)];
			const uint32_t t = t2 * 2u + rhalf;
			const uint32_t ridx0 = t * 64 + (uint32_t) ln, ridx1 = ridx0 + 32;

#pragma unroll
			for (int reg = 0; reg < 16; reg++)
			{
				const uint32_t m = (uint32_t) ((reg & 3) + 8 * (reg >> 2) + 4 * kh);	/* query row of C */
				const uint32_t mi = (gw + (m >> 4)) * NDB_QG + (m & 15);		/* member index in the list's pairs */
				const bool	qv = mi < nmemL;		/* uniform over the half-wave */
				uint32_t	mk = 0xFFFFFFFFu;
				uint32_t	qid = 0, pp = 0, la = 0, nrow = 0;

				if (qv)
				{
					const PairRec pr = pairs[pair_off[L] + mi];

					qid = pr.q;
					pp = pr.p;
					const uint32_t *lq = loc_cand_off + (size_t) qid * (npr + 1);

					la = lq[pp];
					nrow = lq[pp + 1] - la;
					const float qn = qnorm[qid], qe = qnorm[nq_all + qid];
					float		d0, d1;

					if (R == R_IVF_L2)
					{
						d0 = scr_bound_l2(acc0[reg], qn, qe, rn0);
						d1 = scr_bound_l2(acc1[reg], qn, qe, rn1);
					}
					else if (R == R_IVF_IP)
					{
						d0 = scr_bound_ip(acc0[reg], qe);
						d1 = scr_bound_ip(acc1[reg], qe);
					}
					else
					{
						const float nu = (float) (dim + 8) * NDB_SCR_U;
						const float ec = 4.0f * nu / (1.0f - nu);

						d0 = scr_bound_cos(acc0[reg], qn, rn0, ec);
						d1 = scr_bound_cos(acc1[reg], qn, rn1, ec);
					}
					if (ridx0 < nrow)
					{
						dist[(size_t) qid * stride + la + ridx0] = d0;
						mk = ndb_key_from_bits(__float_as_uint(d0));
					}
					if (ridx1 < nrow)
					{
						dist[(size_t) qid * stride + la + ridx1] = d1;
						mk = min(mk, ndb_key_from_bits(__float_as_uint(d1)));
					}
				}
				/* minimum over the half-wave's 32 lanes (both halves shuffle; they hold different queries) */
#pragma unroll
				for (int off = 16; off > 0; off >>= 1)
					mk = min(mk, (uint32_t) __shfl_xor((int) mk, off, 64));
				if (qv && ln == 0 && t * 64u < nrow)
					tmin[(size_t) qid * tstride + (la >> 6) + pp + t] = mk;
			}
		}
		__syncthreads();		/* s_item and the tile are reused by the next item */
	}
	}
}


#endif							/* NDBHIP_SCREEN32_H */
