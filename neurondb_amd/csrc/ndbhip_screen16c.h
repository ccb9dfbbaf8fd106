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
 * chunk c, [r][8 slots of 16 bytes], logical slot s = elements 8 s .. 8 s + 7 of the chunk (s = 2 kstep + khalf),
 * stored at slot s ^ ((r >> 1) & 7) — the geometry of k_s16_row_prep's image with the second plane's slots
 * holding the chunk's next 32 dimensions instead.  rn2[prow] = |x - c|^2 as computed (NaN: not a finite fp32,
 * the row's elements are always emitted), rexp[prow] its scale exponent.
 */
__global__ __launch_bounds__(256) void
k_s16c_row_prep(const float *__restrict__ vecs, int64_t nrows, int dim, int dimp, const int64_t *__restrict__ loc_off,
				const uint32_t *__restrict__ blk_off, int nb, const float *__restrict__ cents /* [nb][dim], or ... */,
				const float *const *__restrict__ cptr /* ... a pointer per bucket */, unsigned char *__restrict__ planes,
				float *__restrict__ rn2, int16_t *__restrict__ rexp, const int64_t *__restrict__ perm)
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

	if (lane == 0)
	{
		rn2[prow] = ok ? (float) s : __uint_as_float(0x7FC00000u);
		rexp[prow] = (int16_t) e;
	}
	const uint32_t pos = (uint32_t) (prow - loc_off[lo]);
	const size_t blk = (size_t) blk_off[lo] + (pos >> 5);
	const int	rr = (int) (pos & 31u);
	const int	nchunk = dimp / S16C_CH;
	unsigned char *img = planes + blk * (size_t) nchunk * 4096 + (size_t) rr * 128;

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
		*reinterpret_cast<ndb_h2 *>(img + (size_t) ch * 4096 + 16 * ((j >> 2) ^ ((rr >> 1) & 7)) + 4 * (j & 3)) = h;
	}
}

/*
 * One wave per (query, bucket) pair record, in the pair tables' order (slot j = pair_off[bucket] + i): the plane of
 * q - c in natural element order (the sweep's DMA applies the LDS swizzle), qcn2[j] = |q - c|^2 as computed (NaN:
 * not a finite fp32 — every element of the pair is emitted), qcexp[j].  Persistent grid.
 */
__global__ __launch_bounds__(256) void
k_s16c_qcprep(const float *__restrict__ queries, int dim, int dimp, const PairRec *__restrict__ pairs,
			  const uint32_t *__restrict__ pair_off, int nb, const float *__restrict__ cents,
			  const float *const *__restrict__ cptr, _Float16 *__restrict__ qcplanes, float *__restrict__ qcn2,
			  int *__restrict__ qcexp, uint32_t cap, unsigned int *__restrict__ flags /* [0]++ when the pairs exceed cap */,
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

	for (uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6); j < total; j += nw)
	{
		uint32_t	lo = 0, hi = (uint32_t) nb;

		while (hi - lo > 1)
		{
			const uint32_t mid = (lo + hi) >> 1;

			if (pair_off[mid] <= j)
				lo = mid;
			else
				hi = mid;
		}
		while (lo + 1 < (uint32_t) nb && pair_off[lo + 1] <= j)
			lo++;
		const float *q = queries + (size_t) pairs[j].q * dim;
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
			qcn2[j] = ok ? (float) s : __uint_as_float(0x7FC00000u);
			qcexp[j] = e;
		}
		ndb_h2	   *out = reinterpret_cast<ndb_h2 *>(qcplanes + (size_t) j * dimp);

		for (int p = lane; p < dimp / 2; p += 64)
		{
			const int	i = 2 * p;
			_Float16	h0 = 0, h1 = 0;

			if (ok && i < dim)
				h0 = (_Float16) (float) ldexp((double) (q[i] - c[i]), 14 - e);
			if (ok && i + 1 < dim)
				h1 = (_Float16) (float) ldexp((double) (q[i + 1] - c[i + 1]), 14 - e);
			ndb_h2		h;

			h.x = h0; h.y = h1;
			out[p] = h;
		}
	}
}

/*
 * Geometry.  QB = 32-pair blocks per tile (4: 128 pairs, the dense form; 1: 32 pairs, for batches whose buckets are
 * probed by a handful of queries each — most of a 128-pair tile would be padding, and the LDS it reserves is
 * better spent on a deeper ring, because that regime is bound by the rows' bytes).  4 waves, 128 rows per tile.
 * Wave w = (wq, wr) = (w % NWQ, w / NWQ) owns AQ pair blocks x BR row blocks.
 */
template <int QB> struct S16CGeom
{
	static constexpr int NW = 4;
	static constexpr int RT = 128;
	static constexpr int NWQ = QB >= 2 ? 2 : 1;
	static constexpr int NWR = NW / NWQ;
	static constexpr int AQ = QB / NWQ;
	static constexpr int BR = 4 / NWR;
	static constexpr int Q_OFF = 4 * 4096;
	static constexpr int BUF = Q_OFF + QB * 4096;
	static constexpr int Q_DMA = QB;			/* 1 KiB pieces of the pair area per wave and chunk (4 QB pieces, 4 waves) */
	static constexpr int PER = 4 + Q_DMA;
	static constexpr int QT = 32 * QB;
};

template <int QB, int NBUF, int DBG = 0>
__global__ __launch_bounds__(256, ((S16CGeom<QB>::BUF * NBUF + 8192) * 2 <= 160 * 1024) ? 2 : 1) void
k_s16c_sweep(IvfDev ix, const unsigned char *__restrict__ planes, const uint32_t *__restrict__ blk_off,
			 const float *__restrict__ rn2, const int16_t *__restrict__ rexp,
			 const unsigned char *__restrict__ qcplanes, uint32_t qrowbytes, const float *__restrict__ qcn2,
			 const int *__restrict__ qcexp, float2 *qthr,
			 const uint32_t *__restrict__ loc_cand_off, int npr, const uint32_t *__restrict__ cnt,
			 const uint32_t *__restrict__ pair_off, const S16Desc *__restrict__ desc,
			 const PairRec *__restrict__ pairs, unsigned int *__restrict__ next_item,
			 const uint32_t *__restrict__ runs, unsigned int *__restrict__ ecount, uint2 *__restrict__ erec,
			 float *__restrict__ eub, uint32_t ecap, uint32_t *__restrict__ bmin, int polite, int nchunk,
			 uint32_t desc_cap, uint32_t topk, const uint32_t *__restrict__ pos_of, float cE,
			 uint32_t qc_cap /* rows of qcplanes: more pairs than that and nothing is swept (k_s16c_qcprep raised the flag) */ )
{
	typedef S16CGeom<QB> G;
	__shared__ uint32_t s_tn, s_tq[S16_TIGHT_Q], s_tkeys[S16_NB];
	__shared__ __attribute__((aligned(1024))) unsigned char ring[NBUF * G::BUF];
	/* per member of the current / next item's tile */
	__shared__ float s_q2[2][G::QT], s_t2[2][G::QT];
	__shared__ int s_eq[2][G::QT];
	__shared__ uint32_t s_la[2][G::QT], s_nrow[2][G::QT], s_qid[2][G::QT];
	__shared__ uint32_t s_desc[2][5];		/* item (S16_NOITEM = none), bucket, row tile, pair tile, members */
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const int	wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	wq = wave % G::NWQ, wr = wave / G::NWQ;
	const int	r32 = lane & 31, kh = lane >> 5;
	uint32_t	hop = 0;

	if (pair_off[ix.ncent] > qc_cap)
		return;					/* uniform */
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
				if (got < run_hi - run_lo && run_lo + got < desc_cap)
					return run_lo + got;
			}
			got = 0xFFFFFFFFu;
		}
		return S16_NOITEM;
	};
	auto		put_desc = [&](int sl, uint32_t it, const S16Desc &d) {
		s_desc[sl][0] = it;
		s_desc[sl][1] = d.L;
		s_desc[sl][2] = d.t2;
		s_desc[sl][3] = d.qt;
		s_desc[sl][4] = it == S16_NOITEM ? 0u : min((uint32_t) G::QT, cnt[d.L] - d.qt * G::QT);
	};
	/* threads < QT: the member's pair slot and record */
	auto		load_pair = [&](int sl, PairRec &pr, uint32_t &slot) -> bool {
		const uint32_t L = s_desc[sl][1], qt = s_desc[sl][3];

		if (s_desc[sl][0] == S16_NOITEM || qt * G::QT + (uint32_t) tid >= cnt[L])
			return false;
		slot = pair_off[L] + qt * G::QT + (uint32_t) tid;
		pr = pairs[slot];
		return true;
	};
	struct Mem { float q2, t2; int eq; uint32_t la, nrow, qid; };
	auto		load_mem = [&](bool have, const PairRec &pr, uint32_t slot, Mem &m) {
		m.q2 = 0.0f; m.t2 = 0.0f; m.eq = 0; m.la = 0; m.nrow = 0; m.qid = 0;
		if (have)
		{
			const uint32_t *lq = loc_cand_off + (size_t) pr.q * (npr + 1);
			const float t = __hip_atomic_load(&qthr[pr.q].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

			m.qid = pr.q;
			m.la = lq[pr.p];
			m.nrow = lq[pr.p + 1] - m.la;
			m.q2 = qcn2[slot];
			m.eq = qcexp[slot];
			/* what the test subtracts: T rounded up with the slack its fused form needs (see the epilogue) */
			m.t2 = s16_up(t * 1.000001f) + NDB_S16_ABS;
		}
	};
	auto		put_mem = [&](int sl, const Mem &m) {
		s_q2[sl][tid] = m.q2;
		s_t2[sl][tid] = m.t2;
		s_eq[sl][tid] = m.eq;
		s_la[sl][tid] = m.la;
		s_nrow[sl][tid] = m.nrow;
		s_qid[sl][tid] = m.qid;
	};

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
		return;
	if (tid < G::QT)
	{
		PairRec		pr = {0, 0};
		uint32_t	slot = 0;
		Mem			m;
		const bool	have = load_pair(0, pr, slot);

		load_mem(have, pr, slot, m);
		put_mem(0, m);
	}
	__syncthreads();

	uint32_t	voff_q[G::Q_DMA];
	bool		qdma = true;
	const unsigned char *rbase, *qbase;
	const uint32_t lane16 = (uint32_t) lane * 16u;

	/* wave w stages 32-row block w of the tile (one contiguous 4 KiB piece of the blocked planes per chunk) and its
	 * share of the tile's pair rows, which are consecutive rows of qcplanes: piece -> pair block piece / 4, rows
	 * 8 (piece % 4) .. + 7, 128 bytes each, the 16-byte slots XOR-swizzled at the source */
	auto		set_dma = [&](int sl) {
		const uint32_t L = s_desc[sl][1], t2 = s_desc[sl][2], qt = s_desc[sl][3], nmem = s_desc[sl][4];
		const uint32_t nbk = blk_off[L + 1] - blk_off[L];
		const uint32_t b = min(t2 * 4u + (uint32_t) wave, nbk - 1u);

#pragma unroll
		for (int j = 0; j < G::Q_DMA; j++)
		{
			const int	piece = wave * G::Q_DMA + j;
			const int	rr = 8 * (piece & 3) + (lane >> 3);
			const uint32_t mem = min((uint32_t) (32 * (piece >> 2) + rr), nmem > 0 ? nmem - 1u : 0u);

			voff_q[j] = mem * qrowbytes + 16u * (uint32_t) ((lane & 7) ^ ((rr >> 1) & 7));
		}
		rbase = planes + ((size_t) blk_off[L] + b) * (size_t) nchunk * 4096;
		qbase = qcplanes + ((size_t) pair_off[L] + (size_t) qt * G::QT) * qrowbytes;
		/* a pair block without a member is neither fetched nor multiplied */
		qdma = nmem > (uint32_t) (32 * ((wave * G::Q_DMA) >> 2));
	};
	const uint32_t ring_la = (uint32_t) (uintptr_t) (ndb_lds_ptr) ring;
	auto		issue = [&](int c, int bufi) {
		if constexpr (DBG == 1)
			return;
		const unsigned char *rb = s16_uniform_ptr(DBG == 2 ? planes : rbase + (size_t) c * 4096);
		const unsigned char *qb = s16_uniform_ptr(DBG == 2 ? qcplanes : qbase + (size_t) c * 128);
		const uint32_t la = ring_la + (uint32_t) bufi * G::BUF;

		s16_dma_linear<4>(rb, lane16, la + wave * 4096);
		if (qdma)
		{
#pragma unroll
			for (int j = 0; j < G::Q_DMA; j++)
				s16_dma16(qb, voff_q[j], la + G::Q_OFF + (wave * G::Q_DMA + j) * 1024);
		}
	};

	const int	sw = (r32 >> 1) & 7;
	const int	qfrag = G::Q_OFF + (G::AQ * wq) * 4096 + r32 * 128;
	const int	rfrag = (G::BR * wr) * 4096 + r32 * 128;
	int			cur = 0;

	set_dma(0);
#pragma unroll
	for (int p = 0; p < NBUF - 1; p++)
		if (p < nchunk)
			issue(p, p);

	for (;;)
	{
		const int	nxt = cur ^ 1;
		ndb_f16acc	acc[G::AQ][G::BR];

#pragma unroll
		for (int a = 0; a < G::AQ; a++)
#pragma unroll
			for (int b = 0; b < G::BR; b++)
#pragma unroll
				for (int i = 0; i < 16; i++)
					acc[a][b][i] = 0.0f;

		const uint32_t nmem_cur = s_desc[cur][4];
		/* pair blocks of this wave that hold a member (wave-uniform) */
		int			na = 0;

#pragma unroll
		for (int a = 0; a < G::AQ; a++)
			na += nmem_cur > (uint32_t) (32 * (G::AQ * wq + a)) ? 1 : 0;

		auto		compute = [&](const unsigned char *buf) {
			if (na == 0)
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
					bh[b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * 4096 + (((2 * s + kh) ^ sw) * 16));
#pragma unroll
				for (int a = 0; a < G::AQ; a++)
				{
					if (a < na)
					{
#pragma unroll
						for (int b = 0; b < G::BR; b++)
							acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
					}
				}
			}
		};

		uint32_t	got = 0xFFFFFFFFu, nit = S16_NOITEM;
		S16Desc		nd = {0, 0, 0, 0};
		PairRec		npair = {0, 0};
		uint32_t	nslot = 0;
		bool		nhave = false;
		Mem			nm;

		/* the next item is prepared while this one is multiplied, one stage per chunk (k_s16_sweep's scheme) */
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
				if (tid < G::QT)
					nhave = load_pair(nxt, npair, nslot);
			}
			else if (st == 4)
			{
				if (tid < G::QT)
					load_mem(nhave, npair, nslot, nm);
			}
			else if (st == 5)
			{
				if (tid < G::QT)
					put_mem(nxt, nm);
			}
		};

		{
			int			bc = 0;

			for (int c = 0; c < nchunk; c++)
			{
				if (c + NBUF - 2 < nchunk)
				{
					if (qdma)
						s16_wait_vm<G::PER * (NBUF - 2)>();
					else
						s16_wait_vm<4 * (NBUF - 2)>();
				}
				else
					s16_wait_vm<0>();
				__syncthreads();
				if (c < 6)
					stage(c);
				const int	bt = bc == 0 ? NBUF - 1 : bc - 1;

				if (c + NBUF - 1 < nchunk)
					issue(c + NBUF - 1, bt);
				compute(ring + bc * G::BUF);
				bc = bc + 1 == NBUF ? 0 : bc + 1;
			}
		}
		for (int st = nchunk; st < 6; st++)
		{
			__syncthreads();
			stage(st);
		}
		__syncthreads();

		const bool	more = s_desc[nxt][0] != S16_NOITEM;
		const uint32_t L = s_desc[cur][1], t2 = s_desc[cur][2];
		const uint32_t len = ix.own_len[L];

		if (more)
		{
			set_dma(nxt);
#pragma unroll
			for (int p = 0; p < NBUF - 1; p++)
				if (p < nchunk)
					issue(p, p);
		}

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

#pragma unroll
		for (int b = 0; b < G::BR; b++)
		{
			const uint32_t ridx = t2 * G::RT + (uint32_t) (32 * (G::BR * wr + b) + r32);
			const bool	rok = ridx < len;
			const size_t grow = (size_t) ix.loc_off[L] + (rok ? ridx : len - 1);
			const float x2 = rn2[grow];
			const int	ex = (int) rexp[grow] - 27;
			const uint32_t porig = pos_of ? pos_of[grow] : ridx;

#pragma unroll
			for (int a = 0; a < G::AQ; a++)
			{
				if (a >= na)
					continue;
#pragma unroll
				for (int reg = 0; reg < 16; reg++)
				{
					const int	m = 32 * (G::AQ * wq + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
					const float t1 = ldexpf(acc[a][b][reg], s_eq[cur][m] + ex);
					const float n = s_q2[cur][m] + x2;
					const float rhs = __builtin_fmaf(n, K, -s_t2[cur][m]);

					if (DBG ? t1 == 1234.5f : !(t1 < rhs))
					{
						if (rok && porig < s_nrow[cur][m])
						{
							const uint32_t q = s_qid[cur][m];
							const float av = n - t1;
							const float e = s16_up(s16_up(cE * n) + NDB_S16_ABS);
							const float lbv = av - e, ubv = s16_up(av + e);
							const float lb = lbv - fabsf(lbv) * 4.8e-7f - 1e-37f;
							const uint32_t slot = atomicAdd(&ecount[q], 1u);
							const uint32_t pos = s_la[cur][m] + porig, ub_bits = __float_as_uint(ubv);

							if (slot < ecap)
							{
								erec[(size_t) q * ecap + slot] = make_uint2(pos, __float_as_uint(lb));
								eub[(size_t) q * ecap + slot] = ubv;
							}
							/* the smallest upper bound of every hash bucket of positions (kept whether or not the
							 * record fit): k non-empty buckets are k distinct candidates */
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
					}
				}
			}
		}
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
						const float nt = s16c_t_from_ub(__uint_as_float(tb), ix.dim);

						/* T >= 0 (or +inf): its bits order like the values */
						atomicMin(reinterpret_cast<unsigned int *>(&qthr[q].x), __float_as_uint(nt));
					}
				}
				__syncthreads();
			}
			if (tid == 0 && s_tn != 0)
				s_tn = 0;
		}
		if (!more)
			break;
		cur = nxt;
	}
}

#endif							/* NDBHIP_SCREEN16C_H */
