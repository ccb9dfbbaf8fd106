/*
 * ndbhip_hnsw.hip — the HNSW half of libndbhip.so: device mirror of the one-node-per-page graph, hnswSearch
 * (src/index/hnsw_am.c:1545-2080) as k_hnsw_search / k_hnsw_search_fast, the reference's unused best-first
 * search (src/scan/hnsw_scan.c) as k_hnsw_scan_layer, hnswInsertNode / hnswbuild (hnsw_am.c:2091-2670, 343-415) as
 * optimistic batches, hnswbulkdelete, and the host entry points of include/ndbhip.h for them.
 * Written for wave64 / CDNA4 only.
 */
#include "ndbhip_internal.h"

int			g_hnsw_trace = 0, g_hnsw_nofast = 0;	/* ndbhip_set_option("hnsw_trace" / "hnsw_nofast") */
int			g_h2_waves = 16;		/* waves per CU of the intended HNSW's walk kernels ("hnsw_intended_waves"): each owns a visited bitmap */

/* hnswbuild: 0 the one-wave sequential kernel, 1 optimistic batches with the chunked block-wide commit (hashed
 * when m <= 16, else sorted), 2 optimistic batches with the one-wave commit, 3 optimistic batches with the
 * sorted chunked commit; batch = min(max, nodes so far / div) walks */
static int	g_hnsw_search_mode = 0;
static int	g_hnsw_spec = 1;
static int	g_hnsw_batch_div = 64;
static int	g_hnsw_batch_max = 1024;


/* ================================================================== */
/* HNSW: hnswSearch (src/index/hnsw_am.c:1545-2080)                    */
/* ================================================================== */

struct HnswDev
{
	const float *vecs;			/* [nblocks * dim], row b = node b (row 0 = meta page, unused) */
	const int  *levels;			/* [nblocks] */
	const int16_t *ncount;		/* [nblocks * 16] */
	const int64_t *nbr_off;		/* [nblocks + 1] (packed layout) */
	const uint32_t *nbrs;
	const uint64_t *tids;		/* [nblocks] */
	int64_t		dense_stride;	/* != 0: node b's slots start at b * dense_stride (16 levels x 2m each) */
	uint32_t	nblocks;
	int			dim;
	int			m;
	uint32_t	entry_point;
	int			entry_level;
};

/* hnswValidateBlockNumber (:1228-1241) + "the meta page holds no node" (PageIsEmpty checks) */
__device__ __forceinline__ bool
hnsw_valid(uint32_t nblocks, uint32_t b)
{
	return b != NDBHIP_INVALID_BLOCK && b < nblocks && b != 0;
}

__device__ __forceinline__ int
hnsw_clamp(int c, int m)
{
	return c < 0 ? 0 : (c > 2 * m ? 2 * m : c);
}

/* Graph metadata read.  MUT = the graph is being modified by this kernel (build): go through an
 * agent-scope load so that neither the scalar cache nor the CU's L1 can serve a stale value. */
template <bool MUT, class T>
__device__ __forceinline__ T
gload(const T *p)
{
	if (MUT)
		return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return *p;
}

__device__ __forceinline__ const uint32_t *
hnsw_nbr_base(const HnswDev &g, uint32_t b)
{
	return g.nbrs + (g.dense_stride ? (int64_t) b * g.dense_stride : g.nbr_off[b]);
}

struct HnswLds
{
	float	   *tile;
	uint64_t   *e_id;
	FinalizeScratch fs;
	uint32_t   *cand, *cdist, *e_pos, *visited;	/* visited: hash set of vmask + 1 slots */
	int		   *s_count;
	uint32_t	npad, vmask;
};

/* slots of the visited hash set: at most ef + 2m + 63 blocks are ever scored at level 0; load factor <= 1/2 */
__host__ __device__ static inline uint32_t
hnsw_vslots(uint32_t ef, uint32_t m)
{
	return next_pow2(2u * (ef + 2u * m + 64u));
}

__host__ __device__ static inline size_t
hnsw_smem_bytes(uint32_t ef, uint32_t k, uint32_t m, size_t tile_bytes = (size_t) NDB_TILE_FLOATS * 4)
{
	const uint32_t npad = next_pow2(ef < 4 ? 4 : ef);

	return tile_bytes + (size_t) ef * (4 + 4 + 4 + 8) + (size_t) hnsw_vslots(ef, m) * 4 +
		(size_t) npad * (8 + 4 + 4 + 1) + (size_t) k * 4 + 128;
}

__device__ static inline HnswLds
carve_hnsw_lds(unsigned char *sp, uint32_t ef, uint32_t k, uint32_t m, size_t tile_bytes = (size_t) NDB_TILE_FLOATS * 4)
{
	HnswLds		L;

	L.npad = next_pow2(ef < 4 ? 4 : ef);
	L.tile = (float *) sp;				sp += tile_bytes;	/* staging tile, or the block-cooperative scorer's region */
	L.e_id = (uint64_t *) sp;			sp += (size_t) ef * 8;
	L.fs.comp = (uint64_t *) sp;		sp += (size_t) L.npad * 8;
	L.cand = (uint32_t *) sp;			sp += (size_t) ef * 4;
	L.cdist = (uint32_t *) sp;			sp += (size_t) ef * 4;	/* float bits */
	L.e_pos = (uint32_t *) sp;			sp += (size_t) ef * 4;
	L.vmask = hnsw_vslots(ef, m) - 1u;
	L.visited = (uint32_t *) sp;		sp += (size_t) (L.vmask + 1u) * 4;
	L.fs.perm = (uint32_t *) sp;		sp += (size_t) L.npad * 4;
	L.fs.curpos = (uint32_t *) sp;		sp += (size_t) L.npad * 4;
	L.fs.order = (uint32_t *) sp;		sp += (size_t) k * 4;
	L.fs.taken = (uint8_t *) sp;		/* npad bytes (multiple of 4), then one int */
	L.s_count = (int *) (L.fs.taken + L.npad);
	return L;
}

/*
 * hnswSearch's walk for ONE query by ONE wave (hnsw_am.c:1593-1975): greedy descent, then the level-0
 * "BFS until ef candidates" loop.  The walk is the reference's, statement for statement; only the
 * distance evaluations of one neighbour list are batched (one lane per neighbour) — they do not depend on
 * the sequential state — and the sequential bookkeeping (visited marks, append / replace-worst, first-min
 * ties) is then replayed in neighbour order.  Leaves candidates[0..cc) / their distances in L.cand /
 * L.cdist.  Returns false when the reference returns "no results" before level 0.
 */
/*
 * Block-cooperative scorer: the 64 lanes of the walking wave each hold (at most) one row to score; the
 * whole 256-thread block scores them together, K = 4..KMAX threads per row, thread `part` summing the float4
 * pieces part, part + K, ... in fp64 (a row's K threads read K consecutive float4 = one coalesced line per
 * step, and all of a thread's loads are in flight at once: the walk is a chain of dependent fetches, so
 * latency is what it costs).
 *
 * That is NOT the reference's summation order, so a result is only accepted when it provably cannot matter.
 * Every term is what the reference adds ((double)(q-x) squared; the fp32 product q*x widened), so:
 *   sums of terms >= 0 (L2's sum, cosine's row norm): any fp64 summation order lies within n*u of the exact
 *     sum (u = 2^-53), hence the reference's sequential sum s* is in [s(1-eps), s(1+eps)], eps = 3*dim*u;
 *   signed sums (the dot product): |s - s*| <= E = 3*dim*u*A with A = the sum of |terms|, accumulated
 *     beside it;
 *   the query's own norm (cosine) is computed ONCE per walk in the reference's order: exact.
 * sqrt, *, /, 1 - x and the narrowing to float are correctly rounded, hence monotone in each argument; the
 * distance is therefore bracketed by its values at the interval end points, and when those agree as floats
 * that float IS the reference's result, bit for bit.  Otherwise (1e-6 .. 1e-5 of the rows) the lane redoes
 * its row in the reference's order.  Zero norms are exact either way (a sum of squares is 0 iff every term is).
 */
#define NDB_HNSW_FAST_MAX_DIM 1920	/* build: partial sums + rows + ctl + q must fit the 16 KiB tile region */

struct HnswFast
{
	float	   *q;				/* [dim] the query / inserted vector, in LDS */
	uint32_t   *rows;			/* [64] compacted rows to score */
	uint32_t   *ctl;			/* [0] 1 = score, 0 = helpers may leave; [1] rows to score */
	double	   *part;			/* [nacc][64 * KMAX] partial sums */
	double		qnorm;			/* cosine: the query's sum of squares, reference order */
};

template <int R> struct FastAcc;
template <> struct FastAcc<R_HNSW_L2> { static constexpr int N = 1; };
template <> struct FastAcc<R_HNSW_IP> { static constexpr int N = 2; };	/* dot, sum |terms| */
template <> struct FastAcc<R_HNSW_COS> { static constexpr int N = 3; };	/* dot, sum |terms|, row norm */

__host__ __device__ static inline size_t
hnsw_fast_bytes(int nacc, int kmax, int dim)
{
	return (size_t) nacc * 64 * kmax * 8 + 64 * 4 + 16 + (((size_t) dim * 4 + 15) & ~(size_t) 15);
}

__device__ __forceinline__ HnswFast
carve_hnsw_fast(void *base, int nacc, int kmax, int dim)
{
	HnswFast	F;

	F.part = (double *) base;
	F.rows = (uint32_t *) (F.part + (size_t) nacc * 64 * kmax);
	F.ctl = F.rows + 64;
	F.q = (float *) (F.ctl + 4);	/* 16-byte aligned */
	F.qnorm = 0.0;
	return F;
}

template <int R, int KMAX>
__device__ __forceinline__ void
hnsw_fast_part(const float *__restrict__ vecs, int dim, const HnswFast &F)
{
	const uint32_t na = F.ctl[1];
	const uint32_t r2 = next_pow2(na);
	const uint32_t K = (256u / r2) > (uint32_t) KMAX ? (uint32_t) KMAX : (256u / r2);	/* a power of two >= 4 */
	const uint32_t part = threadIdx.x & (K - 1u);
	const uint32_t slot = threadIdx.x / K;

	if (slot >= na)
		return;
	const float4 *x = reinterpret_cast<const float4 *>(vecs + (size_t) F.rows[slot] * dim);
	const float4 *q4 = reinterpret_cast<const float4 *>(F.q);
	const int	nf4 = dim >> 2;
	double		s0 = 0.0, s1 = 0.0, s2 = 0.0;
	constexpr int U = 12;
	auto		term = [&](float qv, float xv) {
		if (R == R_HNSW_L2)
		{
			const double d = (double) (qv - xv);

			s0 = s0 + d * d;
		}
		else
		{
			const double t = (double) (qv * xv);	/* fp32 product, widened: hnsw_am.c:1322-1326 */

			s0 = s0 + t;
			s1 = s1 + __builtin_fabs(t);
			if (R == R_HNSW_COS)
				s2 = s2 + (double) (xv * xv);
		}
	};

	for (int f0 = (int) part; f0 < nf4; f0 += U * (int) K)
	{
		float4		buf[U];

#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int	f = f0 + u * (int) K;

			if (f < nf4)
				buf[u] = x[f];
		}
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int	f = f0 + u * (int) K;

			if (f < nf4)
			{
				const float4 qq = q4[f];

				term(qq.x, buf[u].x);
				term(qq.y, buf[u].y);
				term(qq.z, buf[u].z);
				term(qq.w, buf[u].w);
			}
		}
	}
	const uint32_t o = slot * (uint32_t) KMAX + part;

	F.part[o] = s0;
	if (R != R_HNSW_L2)
		F.part[64u * KMAX + o] = s1;
	if (R == R_HNSW_COS)
		F.part[2u * 64u * KMAX + o] = s2;
}

/* helper waves of a walk: score on demand until released */
template <int R, int KMAX>
__device__ void
hnsw_fast_helper(const float *__restrict__ vecs, int dim, const HnswFast &F)
{
	for (;;)
	{
		__syncthreads();
		if (F.ctl[0] == 0u)
			return;
		hnsw_fast_part<R, KMAX>(vecs, dim, F);
		__syncthreads();
	}
}

/* the walking wave: this lane's row (if act) -> its float4 distance to the query under recipe R */
template <int R, int KMAX>
__device__ float
hnsw_fast_score(const float *__restrict__ vecs, int dim, const HnswFast &F, uint32_t row, bool act)
{
	const uint32_t lane = threadIdx.x;
	const unsigned long long mask = __ballot(act);
	const uint32_t na = (uint32_t) __popcll(mask);
	const uint32_t slot = (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
	float		r = 0.0f;

	if (na == 0)
		return r;
	if (act)
		F.rows[slot] = row;
	if (lane == 0)
	{
		F.ctl[0] = 1u;
		F.ctl[1] = na;
	}
	__syncthreads();
	hnsw_fast_part<R, KMAX>(vecs, dim, F);
	__syncthreads();
	if (act)
	{
		const uint32_t r2 = next_pow2(na);
		const uint32_t K = (256u / r2) > (uint32_t) KMAX ? (uint32_t) KMAX : (256u / r2);
		const double eps = 3.0 * (double) dim * 1.1102230246251565e-16;
		double		s0 = 0.0, s1 = 0.0, s2 = 0.0;
		bool		sure;

		for (uint32_t p = 0; p < K; p++)
		{
			s0 = s0 + F.part[slot * KMAX + p];
			if (R != R_HNSW_L2)
				s1 = s1 + F.part[64u * KMAX + slot * KMAX + p];
			if (R == R_HNSW_COS)
				s2 = s2 + F.part[2u * 64u * KMAX + slot * KMAX + p];
		}
		if (R == R_HNSW_L2)
		{
			const float lo = (float) __builtin_sqrt(s0 * (1.0 - eps));
			const float hi = (float) __builtin_sqrt(s0 * (1.0 + eps));

			r = lo;
			sure = lo == hi;
		}
		else if (R == R_HNSW_IP)
		{
			const double E = eps * s1;
			const float lo = (float) (-(s0 + E));
			const float hi = (float) (-(s0 - E));

			r = lo;
			sure = lo == hi;
		}
		else
		{
			if (F.qnorm == 0.0 || s2 == 0.0)	/* :1331-1332, exact */
			{
				r = 2.0f;
				sure = true;
			}
			else
			{
				const double a = __builtin_sqrt(F.qnorm);
				const double E = eps * s1;
				const double blo = __builtin_sqrt(s2 * (1.0 - eps)), bhi = __builtin_sqrt(s2 * (1.0 + eps));
				const float f0 = (float) (1.0 - ((s0 - E) / (a * blo)));
				const float f1 = (float) (1.0 - ((s0 - E) / (a * bhi)));
				const float f2 = (float) (1.0 - ((s0 + E) / (a * blo)));
				const float f3 = (float) (1.0 - ((s0 + E) / (a * bhi)));

				r = f0;
				sure = f0 == f1 && f0 == f2 && f0 == f3;
			}
		}
		if (!sure)
		{
			Acc<R>		acc;
			const float *x = vecs + (size_t) row * dim;

			for (int d = 0; d < dim; d++)
				acc.step(F.q[d], x[d]);
			r = acc.fin();
		}
	}
	return r;
}

#define NDB_HNSW_RS_CAP 256u		/* read-set entries logged per speculative walk */
#define NDB_HNSW_RS_NODE_BITS 28

template <int R, bool MUT, bool LOG = false, bool FAST = false, int KMAX = 16>
__device__ bool
hnsw_walk(const HnswDev &g, const float *__restrict__ q, uint32_t ef, HnswLds &L, uint32_t &cc_out,
		  long long &scored, uint32_t *__restrict__ rs = nullptr, uint32_t *rs_count = nullptr,
		  const HnswFast *F = nullptr)
{
	/* this lane's row -> its distance; every lane of the wave calls it together */
	auto		score = [&](uint32_t row, uint32_t idle_row, bool act) -> float {
		if (FAST)
			return hnsw_fast_score<R, KMAX>(g.vecs, g.dim, *F, row, act);
		return score_rows<R>(q, g.vecs, act ? row : idle_row, g.dim, L.tile);
	};
	uint32_t	rs_local = 0;
	uint32_t   &rs_n = LOG ? *rs_count : rs_local;	/* wave-uniform; the caller publishes it */

	/* LOG: record every (node, level) whose neighbour list this walk reads — the only mutable data
	 * a walk depends on (vectors and node levels never change once written) */
	auto		log_read = [&](uint32_t node, int level) {
		if (LOG)
		{
			if (threadIdx.x == 0 && rs_n < NDB_HNSW_RS_CAP)
				rs[rs_n] = node | ((uint32_t) level << NDB_HNSW_RS_NODE_BITS);
			rs_n++;
		}
	};

	const uint32_t lane = threadIdx.x;
	const int	m2 = 2 * g.m;
	const uint32_t nblocks = g.nblocks;
	uint32_t	cur = g.entry_point;
	int			curLevel = g.entry_level;
	uint32_t   *cand = L.cand, *cdist = L.cdist;

	cc_out = 0;
	if (cur == NDBHIP_INVALID_BLOCK)	/* :1593-1599 */
		return false;
	if (curLevel < 0 || curLevel >= NDBHIP_HNSW_MAX_LEVEL)	/* :1609-1613 */
		curLevel = 0;

	/* ---- greedy descent (:1638-1750) ---- */
	for (int level = curLevel; level > 0; level--)
	{
		bool		found;

		do
		{
			found = false;
			if (!hnsw_valid(nblocks, cur))
				break;
			log_read(cur, level);
			const int	nc = (gload<MUT>(&g.levels[cur]) >= level)
				? hnsw_clamp(gload<MUT>(&g.ncount[(size_t) cur * NDBHIP_HNSW_MAX_LEVEL + level]), g.m) : 0;
			const uint32_t *nb = hnsw_nbr_base(g, cur) + (size_t) level * m2;
			const uint32_t node = cur;
			float		currentDist = 0.0f;

			/* batch 0: lane 0 = the node itself (currentDist, :1683), lanes 1.. = neighbours */
			for (int j0 = -1; j0 < nc; j0 += 64)
			{
				const int	j = j0 + (int) lane;
				uint32_t	my = (j < 0) ? node : ((j < nc) ? gload<MUT>(&nb[j]) : NDBHIP_INVALID_BLOCK);
				const bool	act = hnsw_valid(nblocks, my);
				const float d = score(my, node, act);
				const unsigned long long am = __ballot(act);

				scored += __popcll(am);
				if (j0 < 0)
					currentDist = __shfl(d, 0, 64);
				/* sequential `if (neighborDist < currentDist)` over the batch = first strict minimum */
				const bool	isnb = act && j >= 0;
				uint64_t	key = isnb ? (((uint64_t) ndb_key_from_bits(__float_as_uint(d)) << 32) | lane)
					: ~0ull;
				const uint64_t best = wave_min_u64(key);

				if (best != ~0ull)
				{
					const uint32_t bl = (uint32_t) best & 63u;
					const float bd = __shfl(d, bl, 64);

					if (bd < currentDist)
					{
						cur = __shfl(my, bl, 64);
						currentDist = bd;
						found = true;
					}
				}
			}
		} while (found);
	}

	if (!hnsw_valid(nblocks, cur))	/* :1752-1763 */
		return false;

	/* ---- level 0 (:1765-1975) ---- */
	/*
	 * visitedSet (:1619-1631, a bool per block in the reference) is a membership test and nothing else, so
	 * it lives in LDS as an open-addressing hash set of the blocks scored so far (0 = empty: block 0 is
	 * the meta page and never a node).
	 */
	const uint32_t vmask = L.vmask;
	const uint32_t vshift = 32u - (uint32_t) __popc(vmask);
	uint32_t   *vhash = L.visited;
	auto		v_insert = [&](uint32_t key) {
		uint32_t	h = (key * 2654435761u) >> vshift;

		for (;;)
		{
			const uint32_t prev = atomicCAS(&vhash[h], 0u, key);

			if (prev == 0u || prev == key)
				break;
			h = (h + 1u) & vmask;
		}
	};
	auto		v_contains = [&](uint32_t key) -> bool {
		uint32_t	h = (key * 2654435761u) >> vshift;

		for (;;)
		{
			const uint32_t v = vhash[h];

			if (v == key)
				return true;
			if (v == 0u)
				return false;
			h = (h + 1u) & vmask;
		}
	};
	uint32_t	cc = 1;

	for (uint32_t t = lane; t <= vmask; t += 64)
		vhash[t] = 0u;
	wave_lds_sync();
	{
		const float d0 = score(cur, cur, lane == 0);

		scored += 1;
		if (lane == 0)
		{
			cand[0] = cur;
			cdist[0] = __float_as_uint(d0);
			v_insert(cur);
		}
		wave_lds_sync();
	}
	for (uint32_t i = 0; i < cc && cc < ef; i++)
	{
		const uint32_t c = cand[i];

		if (!hnsw_valid(nblocks, c))
			continue;
		log_read(c, 0);
		const uint32_t *nb = hnsw_nbr_base(g, c);
		/* the list and its count are fetched together (one round trip): slots past the count exist in
		 * both layouts, they are just not neighbours */
		const uint32_t raw0 = ((int) lane < m2) ? gload<MUT>(&nb[lane]) : NDBHIP_INVALID_BLOCK;
		const int	nc = hnsw_clamp(gload<MUT>(&g.ncount[(size_t) c * NDBHIP_HNSW_MAX_LEVEL + 0]), g.m);

		for (int j0 = 0; j0 < nc; j0 += 64)
		{
			const int	j = j0 + (int) lane;
			const uint32_t my = (j < nc) ? (j0 == 0 ? raw0 : gload<MUT>(&nb[j])) : NDBHIP_INVALID_BLOCK;
			bool		ok = hnsw_valid(nblocks, my);

			/* visitedSet test (:1891) against everything scored so far */
			if (ok)
				ok = !v_contains(my);
			/* a block repeated inside this batch is visited by the time its 2nd copy is met */
			for (unsigned long long rem = __ballot(ok); rem; rem &= rem - 1)
			{
				const int	l = __ffsll((long long) rem) - 1;
				const uint32_t other = (uint32_t) __builtin_amdgcn_readlane((int) my, l);

				if ((int) lane > l && other == my)
					ok = false;
			}
			const unsigned long long mask0 = __ballot(ok);

			if (mask0 == 0ull)
				continue;
			const float d = score(my, c, ok);
			const uint32_t nok = (uint32_t) __popcll(mask0);
			const uint32_t rank = (uint32_t) __popcll(mask0 & ((1ull << lane) - 1ull));
			/* while there is room the scored neighbours are appended in list order (:1948-1953) — all at
			 * once; what does not fit goes through replace-worst one by one, as the reference does */
			const uint32_t napp = cc < ef ? (nok < ef - cc ? nok : ef - cc) : 0u;

			scored += nok;
			if (ok)
			{
				v_insert(my);
				if (rank < napp)
				{
					cand[cc + rank] = my;
					cdist[cc + rank] = __float_as_uint(d);
				}
			}
			cc += napp;
			wave_lds_sync();
			unsigned long long mask = mask0;

			for (uint32_t r = 0; r < napp; r++)
				mask &= mask - 1;
			while (mask)
			{
				const int	l = __ffsll((long long) mask) - 1;

				mask &= mask - 1;
				const uint32_t nbk = (uint32_t) __builtin_amdgcn_readlane((int) my, l);
				const float nd = __uint_as_float((uint32_t) __builtin_amdgcn_readlane((int) __float_as_uint(d), l));
				/* :1954-1972: first maximum, strict > */
				uint64_t	wk = 0;

				for (uint32_t t = lane; t < cc; t += 64)
				{
					const uint64_t kk2 = ((uint64_t) ndb_key_from_bits(cdist[t]) << 32) | (0xFFFFFFFFu - t);

					wk = kk2 > wk ? kk2 : wk;
				}
#pragma unroll
				for (int off = 32; off > 0; off >>= 1)
				{
					const uint32_t lo = __shfl_xor((uint32_t) wk, off, 64);
					const uint32_t hi = __shfl_xor((uint32_t) (wk >> 32), off, 64);
					const uint64_t o = ((uint64_t) hi << 32) | lo;

					wk = o > wk ? o : wk;
				}
				const uint32_t widx = 0xFFFFFFFFu - (uint32_t) wk;
				const float wd = __uint_as_float(cdist[widx]);

				if (nd < wd && lane == 0)
				{
					cand[widx] = nbk;
					cdist[widx] = __float_as_uint(nd);
				}
				wave_lds_sync();
			}
		}
	}
	wave_lds_sync();
	cc_out = cc;
	return true;
}

/* top-k of the walk's candidates by the reference's selection sort (:1977-2013); returns kk,
 * result i = candidate L.fs.perm[L.fs.order[i]] */
__device__ uint32_t
hnsw_topk(HnswLds &L, uint32_t cc, uint32_t k, float *out_dist)
{
	for (uint32_t t = threadIdx.x; t < cc; t += blockDim.x)
	{
		L.e_pos[t] = t;
		L.e_id[t] = L.cand[t];
	}
	__syncthreads();
	block_finalize_topk(L.cdist, L.e_pos, L.e_id, cc, next_pow2(cc > 0 ? cc : 1), k, (uint64_t) cc, L.fs,
						(uint64_t *) nullptr, out_dist, L.s_count);
	__syncthreads();
	return (uint32_t) *L.s_count;
}

/* One wave per query. */
template <int R>
__global__ __launch_bounds__(64) void
k_hnsw_search(HnswDev g, const float *__restrict__ queries, uint32_t ef, uint32_t k,
			  uint32_t *__restrict__ out_blocks, float *__restrict__ out_dist, int *__restrict__ out_count,
			  uint64_t *__restrict__ out_tids, long long *__restrict__ out_scored)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	HnswLds		L = carve_hnsw_lds(smem_raw, ef, k, (uint32_t) g.m);
	const uint32_t lane = threadIdx.x;
	const uint32_t qi = blockIdx.x;
	long long	scored = 0;
	uint32_t	cc = 0;
	const bool	ok = hnsw_walk<R, false>(g, queries + (size_t) qi * g.dim, ef, L, cc, scored);
	uint32_t	kk = 0;

	if (ok)
	{
		kk = hnsw_topk(L, cc, k, out_dist + (size_t) qi * k);
		for (uint32_t i2 = lane; i2 < kk; i2 += 64)
		{
			const uint32_t b = L.cand[L.fs.perm[L.fs.order[i2]]];

			out_blocks[(size_t) qi * k + i2] = b;
			if (out_tids)
				out_tids[(size_t) qi * k + i2] = g.tids[b];
		}
	}
	if (lane == 0)
	{
		out_count[qi] = (int) kk;
		if (out_scored) out_scored[qi] = scored;
	}
}

/*
 * The same search with the block-cooperative scorer: one 256-thread block per query, wave 0 walks, the other
 * three help it score (hnsw_fast_score<R>).  A walk is a chain of dependent fetches, so what a batch of
 * queries costs is walks in flight x latency of one: spreading a neighbour list's rows over the block takes
 * the fetch from 12 staged chunks to one round trip.  dim % 4 == 0.
 */
#define NDB_HNSW_SEARCH_KMAX 8
template <int R>
__global__ __launch_bounds__(256) void
k_hnsw_search_fast(HnswDev g, const float *__restrict__ queries, uint32_t ef, uint32_t k,
				   uint32_t *__restrict__ out_blocks, float *__restrict__ out_dist, int *__restrict__ out_count,
				   uint64_t *__restrict__ out_tids, long long *__restrict__ out_scored)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	constexpr int NACC = FastAcc<R>::N;
	HnswLds		L = carve_hnsw_lds(smem_raw, ef, k, (uint32_t) g.m, hnsw_fast_bytes(NACC, NDB_HNSW_SEARCH_KMAX, g.dim));
	HnswFast	F = carve_hnsw_fast(L.tile, NACC, NDB_HNSW_SEARCH_KMAX, g.dim);
	const uint32_t qi = blockIdx.x;
	const float *q = queries + (size_t) qi * g.dim;
	long long	scored = 0;
	uint32_t	cc = 0;
	bool		ok = false;

	for (int d = threadIdx.x; d < g.dim; d += 256)
		F.q[d] = q[d];
	if (threadIdx.x == 0)
		F.ctl[0] = 1u;
	__syncthreads();
	if (R == R_HNSW_COS && threadIdx.x < 64)
	{
		/* norm1 in the reference's order (hnsw_am.c:1322-1326): one chain, once per query; every lane of
		 * the walking wave computes it (LDS broadcast reads) so that no exchange is needed */
		double		n1 = 0.0;

		for (int d = 0; d < g.dim; d++)
			n1 = n1 + (double) (F.q[d] * F.q[d]);
		F.qnorm = n1;
	}
	if (threadIdx.x >= 64)
		hnsw_fast_helper<R, NDB_HNSW_SEARCH_KMAX>(g.vecs, g.dim, F);
	else
	{
		ok = hnsw_walk<R, false, false, true, NDB_HNSW_SEARCH_KMAX>(g, q, ef, L, cc, scored, nullptr, nullptr, &F);
		if (threadIdx.x == 0)
		{
			F.ctl[0] = 0u;
			F.ctl[2] = ok ? 1u : 0u;
			F.ctl[3] = cc;
		}
		__syncthreads();		/* releases the helpers */
	}
	ok = F.ctl[2] != 0u;
	cc = F.ctl[3];
	uint32_t	kk = 0;

	if (ok)
	{
		kk = hnsw_topk(L, cc, k, out_dist + (size_t) qi * k);
		for (uint32_t i2 = threadIdx.x; i2 < kk; i2 += 256)
		{
			const uint32_t b = L.cand[L.fs.perm[L.fs.order[i2]]];

			out_blocks[(size_t) qi * k + i2] = b;
			if (out_tids)
				out_tids[(size_t) qi * k + i2] = g.tids[b];
		}
	}
	if (threadIdx.x == 0)
	{
		out_count[qi] = (int) kk;
		if (out_scored) out_scored[qi] = scored;
	}
}

/*
 * src/scan/hnsw_scan.c: hnsw_search_layer (:379-477) — the best-first search the reference ships next to
 * hnswSearch and never calls, restated rule for rule (oracle: ndbo_hnsw_search_layer):
 * compute_l2_distance (:105-118, fp32 sequential + sqrtf = Acc<R_IVF_L2>) whatever the operator class; a hill
 * climb per upper layer that keeps scanning the neighbours of the node the pass started from (:485-636);
 * at layer 0 (:645-844) a binary min-heap of at most 2 * efSearch candidates (inserts into a full heap are
 * dropped), the entry point pushed with distance 0.0, "visited" = was offered to the heap, the bound
 * results[k - 1] (the k-th slot, not the worst), k unsorted result slots where a better node replaces the
 * first worst one, returned in slot order.
 *
 * One wave per query, persistent blocks.  The distance evaluations of one neighbour list are batched, one
 * lane per neighbour (they do not depend on the sequential state: results and the bound only change after
 * the list); heap and result bookkeeping is replayed in neighbour order by lane 0 in LDS.  The visited set
 * is a bitmap in global memory owned by the block (all-zero between queries: the wave clears the words it
 * set, from a log, or the whole map when the log overflowed).
 */
#define NDB_SCAN_VLOG 4096u

__device__ __forceinline__ bool
scan_readable(uint32_t nblocks, uint32_t b)
{
	return b < nblocks && b != 0;	/* :562-566 / :756-760; the meta page holds no item (PageIsEmpty) */
}

__global__ __launch_bounds__(64) void
k_hnsw_scan_layer(HnswDev g, const float *__restrict__ queries, uint32_t nq, uint32_t ef, uint32_t k,
				  uint32_t *__restrict__ vbits_all, uint32_t vwords, uint32_t *__restrict__ vlog_all,
				  uint32_t *__restrict__ out_blocks, float *__restrict__ out_dist, int *__restrict__ out_count,
				  uint64_t *__restrict__ out_tids, long long *__restrict__ out_scored)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	float	   *tile = (float *) smem_raw;
	uint2	   *heap = (uint2 *) (smem_raw + (size_t) NDB_TILE_FLOATS * 4);	/* .x block, .y float bits */
	uint2	   *res = heap + 2u * ef;
	const uint32_t lane = threadIdx.x;
	const uint32_t nblocks = g.nblocks;
	const int	m2 = 2 * g.m;
	const uint32_t cap = 2u * ef;
	uint32_t   *vbits = vbits_all + (size_t) blockIdx.x * vwords;
	uint32_t   *vlog = vlog_all + (size_t) blockIdx.x * NDB_SCAN_VLOG;

	auto		v_test = [&](uint32_t b) -> bool {
		return (__hip_atomic_load(&vbits[b >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (b & 31u)) & 1u;
	};

	for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x)
	{
		const float *q = queries + (size_t) qi * g.dim;
		long long	scored = 0;
		uint32_t	entry = g.entry_point;
		int			level = g.entry_level;
		uint32_t	candCount = 0, resCount = 0, vcount = 0;

		if (entry == NDBHIP_INVALID_BLOCK || level < 0)	/* :396-402 */
		{
			if (lane == 0)
			{
				out_count[qi] = 0;
				if (out_scored) out_scored[qi] = 0;
			}
			continue;
		}

		/* ---- hnswSearchLayerGreedy per upper layer (:448-457, :485-636) ---- */
		for (; level > 0; level--)
		{
			uint32_t	best = entry;
			bool		changed = true;

			while (changed)
			{
				changed = false;
				if (!scan_readable(nblocks, best))
					break;
				const int	lv = g.levels[best];

				if (lv < 0 || lv >= NDBHIP_HNSW_MAX_LEVEL)	/* :535-540 */
					break;
				const int	nc = hnsw_clamp(g.ncount[(size_t) best * NDBHIP_HNSW_MAX_LEVEL + level], g.m);
				const uint32_t *nb = hnsw_nbr_base(g, best) + (size_t) level * m2;	/* :549: no test of the node's level */
				const uint32_t node = best;
				float		bestDist = 0.0f;

				for (int j0 = -1; j0 < nc; j0 += 64)
				{
					const int	j = j0 + (int) lane;
					const uint32_t my = (j < 0) ? node : ((j < nc) ? nb[j] : NDBHIP_INVALID_BLOCK);
					const bool	act = my != NDBHIP_INVALID_BLOCK && scan_readable(nblocks, my);
					const float d = score_rows<R_IVF_L2>(q, g.vecs, act ? my : node, g.dim, tile);

					scored += __popcll(__ballot(act));
					if (j0 < 0)
						bestDist = __shfl(d, 0, 64);
					/* `if (neighborDist < bestDist)` in neighbour order = the first strict minimum */
					const bool	isnb = act && j >= 0;
					const uint64_t key = isnb ? (((uint64_t) ndb_key_from_bits(__float_as_uint(d)) << 32) | lane) : ~0ull;
					const uint64_t mn = wave_min_u64(key);

					if (mn != ~0ull)
					{
						const uint32_t bl = (uint32_t) mn & 63u;
						const float bd = __shfl(d, bl, 64);

						if (bd < bestDist)
						{
							best = __shfl(my, bl, 64);
							bestDist = bd;
							changed = true;
						}
					}
				}
			}
			entry = best;
		}

		/* ---- hnswSearchLayer0 (:645-844) ---- */
		auto		heap_insert = [&](uint32_t block, uint32_t dbits) {	/* hnswInsertCandidate :235-266 */
			if (candCount >= cap)
				return;
			if (lane == 0)
			{
				uint32_t	i = candCount;
				const float d = __uint_as_float(dbits);

				while (i > 0)
				{
					const uint32_t parent = (i - 1u) / 2u;
					const uint2 pe = heap[parent];

					if (d >= __uint_as_float(pe.y))
						break;
					heap[i] = pe;
					i = parent;
				}
				heap[i] = make_uint2(block, dbits);
			}
			candCount++;
			wave_lds_sync();
		};
		auto		mark = [&](uint32_t block) {	/* hnswMarkVisited :217-230 */
			if (lane == 0)
			{
				if (block < nblocks)
					__hip_atomic_fetch_or(&vbits[block >> 5], 1u << (block & 31u), __ATOMIC_RELAXED,
										  __HIP_MEMORY_SCOPE_AGENT);
				if (vcount < NDB_SCAN_VLOG)
					vlog[vcount] = block;
			}
			vcount++;
		};

		heap_insert(entry, 0u);	/* distance 0.0: :668-671 */
		mark(entry);

		while (candCount > 0)
		{
			/* hnswExtractMinCandidate :271-327 */
			const uint2 top = heap[0];
			const uint32_t block = top.x;
			float		distance = __uint_as_float(top.y);

			candCount--;
			wave_lds_sync();
			if (candCount > 0 && lane == 0)
			{
				const uint2 last = heap[candCount];
				const float ld = __uint_as_float(last.y);
				uint32_t	i = 0;

				for (;;)
				{
					const uint32_t left = 2u * i + 1u, right = left + 1u;
					uint32_t	smallest = i;
					float		sd = ld;

					if (left < candCount && __uint_as_float(heap[left].y) < sd)
					{
						smallest = left;
						sd = __uint_as_float(heap[left].y);
					}
					if (right < candCount && __uint_as_float(heap[right].y) < sd)
						smallest = right;
					if (smallest == i)
						break;
					heap[i] = heap[smallest];
					i = smallest;
				}
				heap[i] = last;
			}
			wave_lds_sync();

			if (resCount >= k && distance > __uint_as_float(res[k - 1u].y))	/* :684-686 */
				continue;
			if (!scan_readable(nblocks, block))
				continue;
			const int	lv = g.levels[block];

			if (lv < 0 || lv >= NDBHIP_HNSW_MAX_LEVEL)	/* :715-720 */
				continue;
			const int	nc = hnsw_clamp(g.ncount[(size_t) block * NDBHIP_HNSW_MAX_LEVEL + 0], g.m);
			const uint32_t *nb = hnsw_nbr_base(g, block);
			const float furthest = resCount >= k ? __uint_as_float(res[k - 1u].y) : FLT_MAX;	/* :744-746 */
			const bool	open = resCount < k;

			for (int j0 = -1; j0 < nc; j0 += 64)
			{
				const int	j = j0 + (int) lane;
				const uint32_t my = (j < 0) ? block : ((j < nc) ? nb[j] : NDBHIP_INVALID_BLOCK);
				/* a neighbour is scored unless invalid, unreadable or already visited (:749-764) */
				bool		act = my != NDBHIP_INVALID_BLOCK && scan_readable(nblocks, my);

				if (act && j >= 0 && v_test(my))
					act = false;
				const float d = score_rows<R_IVF_L2>(q, g.vecs, act ? my : block, g.dim, tile);

				if (j0 < 0)
					distance = __shfl(d, 0, 64);	/* the node itself: :741 */
				const bool	take = act && j >= 0 && (d < furthest || open);	/* :804-810 */
				unsigned long long tm = __ballot(take);
				unsigned long long am = __ballot(act);

				/* replay in neighbour order; a block listed twice is scored again only if its first
				 * occurrence was not offered to the heap (it is "visited" from then on) */
				unsigned long long rest = tm;

				while (rest)
				{
					const int	idx = __ffsll((long long) rest) - 1;
					const uint32_t b = __shfl(my, idx, 64);
					const uint32_t db = __shfl(__float_as_uint(d), idx, 64);
					const unsigned long long same = __ballot(act && my == b) & ~((2ull << idx) - 1ull);

					rest &= rest - 1ull;
					heap_insert(b, db);
					mark(b);
					am &= ~same;		/* later occurrences: visited, neither scored nor offered */
					rest &= ~same;
				}
				scored += __popcll(am);
			}

			/* hnswAddResult :333-365 */
			if (resCount < k)
			{
				if (lane == 0)
					res[resCount] = make_uint2(block, __float_as_uint(distance));
				resCount++;
			}
			else
			{
				/* the first slot holding the largest distance */
				uint64_t	bestk = ~0ull;

				for (uint32_t i = lane; i < resCount; i += 64)
				{
					const uint64_t c = ((uint64_t) (~ndb_key_from_bits(res[i].y)) << 32) | i;

					bestk = c < bestk ? c : bestk;
				}
				bestk = wave_min_u64(bestk);
				const uint32_t wi = (uint32_t) bestk;

				if (lane == 0 && distance < __uint_as_float(res[wi].y))
					res[wi] = make_uint2(block, __float_as_uint(distance));
			}
			wave_lds_sync();
		}

		for (uint32_t i = lane; i < resCount; i += 64)	/* :826-830: slot order */
		{
			const uint2 r = res[i];

			out_blocks[(size_t) qi * k + i] = r.x;
			out_dist[(size_t) qi * k + i] = __uint_as_float(r.y);
			if (out_tids)
				out_tids[(size_t) qi * k + i] = r.x < nblocks ? g.tids[r.x] : 0ull;
		}
		if (lane == 0)
		{
			out_count[qi] = (int) resCount;
			if (out_scored) out_scored[qi] = scored;
		}
		/* leave the bitmap all-zero for the next query */
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
		if (vcount <= NDB_SCAN_VLOG)
			for (uint32_t i = lane; i < vcount; i += 64)
			{
				const uint32_t b = __hip_atomic_load(&vlog[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

				if (b < nblocks)
					__hip_atomic_store(&vbits[b >> 5], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		else
			for (uint32_t i = lane; i < vwords; i += 64)
				__hip_atomic_store(&vbits[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
	}
}

/*
 * hnswbuild (hnsw_am.c:343-415) = hnswInsertNode for every heap row in order (:2091-2670).  The inserts
 * depend on each other (each one searches the graph the previous ones left), so ONE wave walks them in
 * order inside ONE launch; the graph lives in the dense 16-level layout so that the reference's writes
 * at `currentLevel` into nodes allocated with fewer levels (Q12 / Q21) land in a defined slot, exactly
 * like the oracle's model.  levels[i] = the level drawn for row i (hnswGetRandomLevel uses random():
 * injected by the caller).
 */
__global__ __launch_bounds__(64) void
k_hnsw_build(float *vecs, int *levels_out, int16_t *ncount, uint32_t *nbrs, uint64_t *tids_out,
			 const float *__restrict__ rows, const uint64_t *__restrict__ tids_in,
			 const int *__restrict__ levels_in, uint32_t n, int dim, int m, uint32_t efc,
			 uint32_t *entry_io /* [0] entry point, [1] entry level (as int) */, uint32_t base)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	const uint32_t ksel = (uint32_t) m < efc ? (uint32_t) m : efc;
	HnswLds		L = carve_hnsw_lds(smem_raw, efc, efc, (uint32_t) m);
	const uint32_t lane = threadIdx.x;
	const int	m2 = 2 * m;
	const int64_t stride = (int64_t) NDBHIP_HNSW_MAX_LEVEL * m2;
	uint32_t	entry = entry_io[0];
	int			entry_level = (int) entry_io[1];
	long long	scored = 0;

	for (uint32_t i = 0; i < n; i++)
	{
		const uint32_t blk = base + i + 1;
		int			level = levels_in[i];

		if (level >= NDBHIP_HNSW_MAX_LEVEL) level = NDBHIP_HNSW_MAX_LEVEL - 1;
		if (level < 0) level = 0;
		/* Step 4 (:2288-2332): the node's page */
		for (int j = lane; j < dim; j += 64)
			vecs[(size_t) blk * dim + j] = rows[(size_t) i * dim + j];
		for (int j = lane; j < NDBHIP_HNSW_MAX_LEVEL; j += 64)
			ncount[(size_t) blk * NDBHIP_HNSW_MAX_LEVEL + j] = 0;
		for (int64_t j = lane; j < stride; j += 64)
			nbrs[(size_t) blk * stride + j] = NDBHIP_INVALID_BLOCK;
		if (lane == 0)
		{
			levels_out[blk] = level;
			tids_out[blk] = tids_in[i];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

		/* Step 5 (:2334-2640) */
		if (entry != NDBHIP_INVALID_BLOCK && entry_level >= 0)
		{
			HnswDev		g;

			g.vecs = vecs; g.levels = levels_out; g.ncount = ncount; g.nbr_off = nullptr; g.nbrs = nbrs;
			g.tids = tids_out; g.dense_stride = stride; g.nblocks = blk + 1; g.dim = dim; g.m = m;
			g.entry_point = entry; g.entry_level = entry_level;
			const int	maxLevel = level < entry_level ? level : entry_level;

			for (int cl = maxLevel; cl >= 0; cl--)
			{
				uint32_t	cc = 0;
				/* always L2, ef = k = efConstruction (:2369-2378); only the first m results are used,
				 * and the second selection sort (:2391-2414) over already sorted results is the identity */
				const bool	ok = hnsw_walk<R_HNSW_L2, true>(g, rows + (size_t) i * dim, efc, L, cc, scored);
				uint32_t	kk = 0;

				if (ok)
					kk = hnsw_topk(L, cc, ksel, (float *) L.fs.curpos /* scratch: distances not needed */);
				const uint32_t nsel = kk;	/* = Min(m, candidateCount) */

				for (uint32_t idx = 0; idx < nsel; idx++)
				{
					const uint32_t nbk = L.cand[L.fs.perm[L.fs.order[idx]]];
					uint32_t   *newn = nbrs + (size_t) blk * stride + (size_t) cl * m2;
					uint32_t   *nn = nbrs + (size_t) nbk * stride + (size_t) cl * m2;
					int16_t    *ncp = &ncount[(size_t) nbk * NDBHIP_HNSW_MAX_LEVEL + cl];

					if (lane == 0)
					{
						newn[idx] = nbk;		/* :2452-2456 */
						ncount[(size_t) blk * NDBHIP_HNSW_MAX_LEVEL + cl] = (int16_t) (idx + 1);
					}
					/* back-link (:2487-2511): first InvalidBlockNumber slot among the first count, else count */
					const int	cnt = hnsw_clamp(gload<true>(ncp), m);
					int			pos = cnt;

					for (int j0 = 0; j0 < cnt; j0 += 64)
					{
						const int	j = j0 + (int) lane;
						const bool	inv = j < cnt && gload<true>(&nn[j]) == NDBHIP_INVALID_BLOCK;
						const unsigned long long mk = __ballot(inv);

						if (mk)
						{
							pos = j0 + __ffsll((long long) mk) - 1;
							break;
						}
					}
					if (lane == 0 && pos < m2)
					{
						nn[pos] = blk;
						if (pos >= cnt)
							*ncp = (int16_t) (pos + 1);
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				}
			}
		}
		/* Step 6 (:2642-2663) */
		if (entry == NDBHIP_INVALID_BLOCK || level > entry_level)
		{
			entry = blk;
			entry_level = level;
		}
	}
	if (lane == 0)
	{
		entry_io[0] = entry;
		entry_io[1] = (uint32_t) entry_level;
	}
}

/* ------------------------------------------------------------------ */
/* Optimistic batched hnswbuild                                         */
/*                                                                      */
/* hnswInsertNode is sequential by definition: insert i searches the    */
/* graph inserts 0..i-1 left.  But a walk only READS the neighbour      */
/* lists of the few nodes it passes (descent path + the level-0 nodes   */
/* it expands), and an insert only WRITES the lists of the <= m nodes   */
/* it back-links (and not even those once they are full).  So a batch   */
/* of inserts proceeds in ROUNDS:                                       */
/*   speculate  every not yet committed walk whose result is missing or */
/*              stale runs, one wave each and all in parallel, against  */
/*              the graph as it stands, logging the (node, level) lists */
/*              it read;                                                */
/*   commit     ONE wave applies the walks' selections in insert order  */
/*              for as long as every list a walk read is unwritten      */
/*              since that walk ran — such a walk saw exactly the graph */
/*              the sequential run would have shown it — and stops at   */
/*              the first stale one, which the next round redoes.       */
/* The first walk of a round's commit ran in that very round with       */
/* nothing written since, so every round commits at least one walk; the */
/* result is the sequential graph, slot for slot (tests: device build   */
/* == oracle).  A "walk" is one (insert, level) pair = one hnswSearch   */
/* call of hnswInsertNode's level loop (:2360-2520).  Staleness is      */
/* tracked per node in two classes, level 0 and levels >= 1: stamp[c]   */
/* [node] = the last round that wrote such a list.                      */
/* ------------------------------------------------------------------ */

struct HnswTask
{
	uint32_t	row;			/* heap row i; its node is block i + 1 */
	int32_t		cl;				/* level being linked */
};

/* Step 4 (:2288-2332) for every row at once: a node page is unreachable until its own insert links it,
 * and nobody writes into it before that (back-links only go to older nodes). */
__global__ __launch_bounds__(256) void
k_hnsw_init_nodes(float *vecs, int *levels_out, int16_t *ncount, uint32_t *nbrs, uint64_t *tids_out,
				  const float *__restrict__ rows, const uint64_t *__restrict__ tids_in,
				  const int *__restrict__ levels_in, uint32_t n, int dim, int64_t stride, uint32_t base)
{
	const uint32_t i = blockIdx.x;
	const uint32_t blk = base + i + 1;	/* `base` nodes exist already (hnswinsert into a built graph) */

	if (i >= n)
		return;
	for (int j = threadIdx.x; j < dim; j += 256)
		vecs[(size_t) blk * dim + j] = rows[(size_t) i * dim + j];
	for (int64_t j = threadIdx.x; j < stride; j += 256)
		nbrs[(size_t) blk * stride + j] = NDBHIP_INVALID_BLOCK;
	if (threadIdx.x < NDBHIP_HNSW_MAX_LEVEL)
		ncount[(size_t) blk * NDBHIP_HNSW_MAX_LEVEL + threadIdx.x] = 0;
	if (threadIdx.x == 0)
	{
		int			level = levels_in[i];

		if (level >= NDBHIP_HNSW_MAX_LEVEL) level = NDBHIP_HNSW_MAX_LEVEL - 1;
		if (level < 0) level = 0;
		levels_out[blk] = level;
		tids_out[blk] = tids_in[i];
	}
}

/* per-batch state of the rounds */
struct HnswRounds
{
	uint32_t   *next;			/* [1] first uncommitted walk of the batch */
	uint32_t   *spec_round;		/* [batch] round each walk last ran in (0 = never) */
	uint32_t   *sel;			/* [batch * ksel] its selection */
	int		   *nsel;			/* [batch] */
	uint32_t   *rs;				/* [batch * NDB_HNSW_RS_CAP] its read set */
	uint32_t   *rsn;			/* [batch] entries logged (> cap: overflowed, never validates) */
	uint32_t   *stamp0;			/* [nblocks] last round that wrote the node's level-0 list */
	uint32_t   *stampU;			/* [nblocks] ... one of its upper-level lists */
	unsigned long long *stats;	/* [0] walks run, [1] commit stops on a stale walk, [2] read-set overflows */
};

/* has any list this walk read been written in round `since` or later? (wave-uniform) */
template <bool MUT>
__device__ __forceinline__ bool
hnsw_walk_is_stale(const HnswRounds &R, uint32_t t, uint32_t since)
{
	const uint32_t rsn = R.rsn[t];

	if (rsn > NDB_HNSW_RS_CAP)
		return true;
	for (uint32_t e0 = 0; e0 < rsn; e0 += 64)
	{
		bool		hit = false;

		if (e0 + (threadIdx.x & 63u) < rsn)	/* every wave of the block checks the whole log */
		{
			const uint32_t enc = R.rs[(size_t) t * NDB_HNSW_RS_CAP + e0 + (threadIdx.x & 63u)];
			const uint32_t node = enc & ((1u << NDB_HNSW_RS_NODE_BITS) - 1u);
			const uint32_t *st = (enc >> NDB_HNSW_RS_NODE_BITS) ? R.stampU : R.stamp0;

			hit = gload<MUT>(&st[node]) >= since;
		}
		if (__ballot(hit) != 0ull)
			return true;
	}
	return false;
}

/*
 * One block per walk of the batch: (re)run it if it is uncommitted and has no valid result.  Wave 0 walks;
 * FAST: three more waves help it score (hnsw_fast_score), else the block is that one wave.
 */
template <bool FAST>
__global__ __launch_bounds__(FAST ? 256 : 64) void
k_hnsw_spec(HnswDev g, const float *__restrict__ rows, const HnswTask *__restrict__ tasks, uint32_t efc,
			uint32_t ksel, HnswRounds R, uint32_t round, uint32_t base)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	const uint32_t t = blockIdx.x;

	if (t < *R.next)
		return;
	const uint32_t ran = R.spec_round[t];

	if (ran != 0 && !hnsw_walk_is_stale<false>(R, t, ran))	/* block-uniform: every wave sees the same lists */
		return;
	HnswLds		L = carve_hnsw_lds(smem_raw, efc, efc, (uint32_t) g.m);
	const HnswTask task = tasks[t];
	const float *q = rows + (size_t) (task.row - base) * g.dim;	/* rows[] holds the new rows only */
	HnswFast	F = carve_hnsw_fast(L.tile, 1, 16, g.dim);	/* 8 KiB of partial sums + the row, inside the tile region */

	if (FAST)
	{
		for (int d = threadIdx.x; d < g.dim; d += 256)
			F.q[d] = q[d];
		if (threadIdx.x == 0)
			F.ctl[0] = 1u;
		__syncthreads();
	}
	long long	scored = 0;
	uint32_t	cc = 0, rsn = 0;
	bool		ok = false;

	g.nblocks = task.row + 2;	/* the relation ends at this row's own page */
	if (FAST && threadIdx.x >= 64)
		hnsw_fast_helper<R_HNSW_L2, 16>(g.vecs, g.dim, F);
	else
	{
		ok = hnsw_walk<R_HNSW_L2, false, true, FAST>(g, q, efc, L, cc, scored,
													   R.rs + (size_t) t * NDB_HNSW_RS_CAP, &rsn, &F);
		if (FAST)
		{
			if (threadIdx.x == 0)
			{
				F.ctl[0] = 0u;
				F.ctl[2] = ok ? 1u : 0u;
				F.ctl[3] = cc;
			}
			__syncthreads();	/* releases the helpers */
		}
	}
	if (FAST)
	{
		ok = F.ctl[2] != 0u;	/* the whole block selects together */
		cc = F.ctl[3];
	}
	uint32_t	kk = 0;

	if (ok)
		kk = hnsw_topk(L, cc, ksel, (float *) L.fs.curpos);
	for (uint32_t i = threadIdx.x; i < kk; i += blockDim.x)
		R.sel[(size_t) t * ksel + i] = L.cand[L.fs.perm[L.fs.order[i]]];
	if (threadIdx.x == 0)
	{
		R.nsel[t] = (int) kk;
		R.rsn[t] = rsn;
		R.spec_round[t] = round;
		atomicAdd(&R.stats[0], 1ull);
		if (rsn > NDB_HNSW_RS_CAP)
			atomicAdd(&R.stats[2], 1ull);
	}
}

template <class T>
__device__ __forceinline__ void
gstore(T *p, T v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* publish this wave's global writes to its own later (cache-bypassing) reads */
__device__ __forceinline__ void
hnsw_publish()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

/*
 * The linking half of one level of hnswInsertNode (:2416-2520): node blk takes sel[0..nsel) as its level-cl
 * neighbours, and every selected node gets blk written into the first InvalidBlockNumber slot among its
 * first `count` level-cl slots, else appended (dropped when the 2m slots are full).  The selected nodes are
 * distinct (the walk never scores a block twice), so the back-links are independent and run one per lane —
 * unless blk selected ITSELF (reachable through its own upper-level back-links, quirk Q12), where the
 * reference's statement order decides which write survives: that case is replayed by one lane in order.
 * Every list actually written is stamped with `round`; a back-link dropped because the list is full writes
 * nothing — which is what keeps saturated hub nodes from serialising the build.
 */
__device__ void
hnsw_link(uint32_t *nbrs, int16_t *ncount, uint32_t blk, int cl, int m, int64_t stride, const uint32_t *sel,
		  uint32_t nsel, uint32_t *stamp0, uint32_t *stampU, uint32_t round)
{
	const uint32_t lane = threadIdx.x;
	const int	m2 = 2 * m;
	uint32_t   *newn = nbrs + (size_t) blk * stride + (size_t) cl * m2;
	int16_t    *newc = &ncount[(size_t) blk * NDBHIP_HNSW_MAX_LEVEL + cl];
	uint32_t   *stamp = cl ? stampU : stamp0;
	bool		self = false;

	if (nsel == 0)
		return;
	for (uint32_t i0 = 0; i0 < nsel; i0 += 64)
		self = self || __ballot(i0 + lane < nsel && sel[i0 + lane] == blk) != 0ull;
	if (self)
	{
		if (lane == 0)
			for (uint32_t idx = 0; idx < nsel; idx++)
			{
				const uint32_t nbk = sel[idx];
				uint32_t   *nn = nbrs + (size_t) nbk * stride + (size_t) cl * m2;
				int16_t    *ncp = &ncount[(size_t) nbk * NDBHIP_HNSW_MAX_LEVEL + cl];

				gstore(&newn[idx], nbk);			/* :2452-2456 */
				gstore(newc, (int16_t) (idx + 1));
				const int	cnt = hnsw_clamp(gload<true>(ncp), m);
				int			pos = cnt;

				for (int j = 0; j < cnt; j++)
					if (gload<true>(&nn[j]) == NDBHIP_INVALID_BLOCK)
					{
						pos = j;
						break;
					}
				if (pos < m2)
				{
					gstore(&nn[pos], blk);
					if (pos >= cnt)
						gstore(ncp, (int16_t) (pos + 1));
					gstore(&stamp[nbk], round);
				}
			}
	}
	else
	{
		for (uint32_t i0 = 0; i0 < nsel; i0 += 64)
		{
			const uint32_t idx = i0 + lane;

			if (idx < nsel)
			{
				const uint32_t nbk = sel[idx];
				uint32_t   *nn = nbrs + (size_t) nbk * stride + (size_t) cl * m2;
				int16_t    *ncp = &ncount[(size_t) nbk * NDBHIP_HNSW_MAX_LEVEL + cl];
				const int	cnt = hnsw_clamp(gload<true>(ncp), m);
				int			pos = cnt;

				gstore(&newn[idx], nbk);
				for (int j = cnt - 1; j >= 0; j--)	/* first invalid slot = the lowest one */
					if (gload<true>(&nn[j]) == NDBHIP_INVALID_BLOCK)
						pos = j;
				if (pos < m2)
				{
					gstore(&nn[pos], blk);
					if (pos >= cnt)
						gstore(ncp, (int16_t) (pos + 1));
					gstore(&stamp[nbk], round);
				}
			}
		}
		if (lane == 0)
			gstore(newc, (int16_t) nsel);
	}
	/* blk's own list changed too (only reachable through a stamped list, but a stale check is cheap) */
	if (lane == 0)
		gstore(&stamp[blk], round);
}

/* ONE wave commits the batch's walks in insert order until it meets a stale one */
__global__ __launch_bounds__(64) void
k_hnsw_commit(int16_t *ncount, uint32_t *nbrs, const HnswTask *__restrict__ tasks, uint32_t ntasks, int m,
			  uint32_t ksel, HnswRounds R, uint32_t round)
{
	__shared__ uint32_t sel[NDBHIP_MAX_EF];
	const uint32_t lane = threadIdx.x;
	const int64_t stride = (int64_t) NDBHIP_HNSW_MAX_LEVEL * 2 * m;
	const uint32_t first = *R.next;
	uint32_t	t = first;

	for (; t < ntasks; t++)
	{
		const uint32_t ran = R.spec_round[t];

		/* the round's first walk ran in this round with nothing written since: valid by construction
		 * (also what lets a walk whose read set overflowed the log get through) */
		if (!(t == first && ran == round) && (ran == 0 || hnsw_walk_is_stale<true>(R, t, ran)))
			break;
		const HnswTask task = tasks[t];
		const uint32_t nsel = (uint32_t) R.nsel[t];

		for (uint32_t i = lane; i < nsel; i += 64)
			sel[i] = R.sel[(size_t) t * ksel + i];
		__syncthreads();
		hnsw_link(nbrs, ncount, task.row + 1, task.cl, m, stride, sel, nsel, R.stamp0, R.stampU, round);
		hnsw_publish();
		__syncthreads();
	}
	if (lane == 0)
	{
		*R.next = t;
		if (t < ntasks)
			atomicAdd(&R.stats[1], 1ull);
	}
}

/*
 * The same commit, a chunk of walks at a time by a whole block.  What makes that legal: the lists of different
 * (node, level) pairs evolve independently — a back-link goes to the first InvalidBlockNumber slot of ITS
 * list, else to the tail — so the requests of a chunk are sorted by (node, level, walk) and every list replays
 * its own requests in walk order (one thread per list), assuming for the moment that every walk of the chunk
 * commits.  That replay yields, per list, the first walk that really writes it.  A walk is stale if a list it
 * read was written before the chunk since it ran (stamps), or is first written inside the chunk by an EARLIER
 * walk; `stop` = the first stale walk.  For every walk up to `stop` the assumption held (all its predecessors
 * do commit), so its verdict and its slot positions are the sequential ones; the writes of walks < stop are
 * then applied, all at once.  A node's own list (hnsw_am.c:2452-2456) takes part as a request that always
 * writes.  A walk that selected its own node (quirk Q12) is committed alone through hnsw_link.
 */
#define NDB_HC_TASKS 64u			/* walks per chunk */
#define NDB_HC_REQ 2048u			/* requests per chunk, padded (a power of two) */
#define NDB_HC_MAXSEL 31u			/* NDB_HC_TASKS * (NDB_HC_MAXSEL + 1) <= NDB_HC_REQ */
#define NDB_HC_NONE 0xFFu

__device__ __forceinline__ uint64_t
hc_key(uint32_t node, int level, uint32_t j, uint32_t own)
{
	return ((uint64_t) node << 16) | ((uint64_t) level << 12) | ((uint64_t) j << 4) | own;
}

__global__ __launch_bounds__(256) void
k_hnsw_commit_par(int16_t *ncount, uint32_t *nbrs, const HnswTask *__restrict__ tasks, uint32_t ntasks, int m,
				  uint32_t ksel, HnswRounds R, uint32_t round)
{
	__shared__ uint64_t key[NDB_HC_REQ];
	__shared__ uint32_t sel[NDB_HC_TASKS * NDB_HC_MAXSEL];
	__shared__ uint16_t fw[NDB_HC_REQ];			/* at a run head: first walk of the chunk that writes this list */
	__shared__ uint8_t pos[NDB_HC_REQ];			/* slot a back-link request lands in, NDB_HC_NONE = dropped */
	__shared__ uint8_t cnt0s[NDB_HC_REQ];		/* at a run head: the list's count before the chunk */
	__shared__ uint32_t t_ran[NDB_HC_TASKS], t_rsn[NDB_HC_TASKS], t_nsel[NDB_HC_TASKS], t_blk[NDB_HC_TASKS];
	__shared__ int t_cl[NDB_HC_TASKS];
	__shared__ uint32_t t_stale[NDB_HC_TASKS], t_off[NDB_HC_TASKS + 1];
	__shared__ uint32_t s_stop, s_self;
	const uint32_t tid = threadIdx.x;
	const int	m2 = 2 * m;
	const int64_t stride = (int64_t) NDBHIP_HNSW_MAX_LEVEL * m2;
	const uint32_t first = *R.next;
	/* as many walks as keep the padded request count at 1024 when they fit (60 walks at m = 16): the sort is
	 * the chunk's biggest fixed cost */
	const uint32_t cmax = min(NDB_HC_TASKS, (1024u / (ksel + 1u)) >= 16u ? 1024u / (ksel + 1u) : NDB_HC_REQ / (ksel + 1u));
	uint32_t	cur = first;
	bool		stopped = false;

	while (cur < ntasks && !stopped)
	{
		uint32_t	C = min(cmax, ntasks - cur);

		/* ---- the chunk's walks ---- */
		if (tid < C)
		{
			const uint32_t t = cur + tid;
			const HnswTask task = tasks[t];

			t_ran[tid] = R.spec_round[t];
			t_rsn[tid] = R.rsn[t];
			t_nsel[tid] = (uint32_t) R.nsel[t];
			t_blk[tid] = task.row + 1;
			t_cl[tid] = task.cl;
			/* never run, or its read-set log overflowed: cannot be validated (unless it opens the round) */
			t_stale[tid] = (t_ran[tid] == 0 || t_rsn[tid] > NDB_HNSW_RS_CAP) ? 1u : 0u;
		}
		if (tid == 0)
		{
			s_stop = C;
			s_self = C;
		}
		__syncthreads();
		for (uint32_t e = tid; e < C * ksel; e += 256)
		{
			const uint32_t j = e / ksel, idx = e % ksel;

			if (idx < t_nsel[j])
			{
				const uint32_t v = R.sel[(size_t) (cur + j) * ksel + idx];

				sel[j * NDB_HC_MAXSEL + idx] = v;
				if (v == t_blk[j])
					atomicMin(&s_self, j);
			}
		}
		__syncthreads();
		const bool	solo = s_self == 0;	/* the chunk's first walk selected its own node: commit it alone */

		if (solo)
			C = 1;
		else if (s_self < C)
			C = s_self;					/* ... a later one: it will open the next chunk */
		const bool	opens_round = cur == first && t_ran[0] == round;	/* valid by construction */

		if (tid == 0)
		{
			s_stop = C;
			if (opens_round)
				t_stale[0] = 0;
		}
		__syncthreads();

		/* ---- stale against what was written before this chunk ---- */
		/* (walk, read-set entry) pairs are spread over the block, 8 per thread in flight */
		if (tid == 0)
		{
			uint32_t	acc = 0;

			for (uint32_t j = 0; j < C; j++)
			{
				t_off[j] = acc;
				acc += t_rsn[j] > NDB_HNSW_RS_CAP ? 0u : t_rsn[j];
			}
			t_off[C] = acc;
		}
		__syncthreads();
		const uint32_t npairs = t_off[C];
		auto		pair_walk = [&](uint32_t p) -> uint32_t {	/* largest j with t_off[j] <= p */
			uint32_t	lo = 0, hi = C;

			while (hi - lo > 1)
			{
				const uint32_t mid = (lo + hi) >> 1;

				if (t_off[mid] <= p)
					lo = mid;
				else
					hi = mid;
			}
			return lo;
		};

		for (uint32_t base = 0; base < npairs; base += 256u * 8u)
		{
			uint32_t	enc[8], jj[8], stv[8];

#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t p = base + (uint32_t) u * 256u + tid;

				jj[u] = 0xFFFFFFFFu;
				enc[u] = 0;
				if (p < npairs)
				{
					jj[u] = pair_walk(p);
					enc[u] = R.rs[(size_t) (cur + jj[u]) * NDB_HNSW_RS_CAP + (p - t_off[jj[u]])];
				}
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t node = enc[u] & ((1u << NDB_HNSW_RS_NODE_BITS) - 1u);
				const uint32_t *st = (enc[u] >> NDB_HNSW_RS_NODE_BITS) ? R.stampU : R.stamp0;

				stv[u] = jj[u] != 0xFFFFFFFFu ? gload<true>(&st[node]) : 0u;
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
				if (jj[u] != 0xFFFFFFFFu && !(jj[u] == 0 && opens_round) && stv[u] >= t_ran[jj[u]])
					t_stale[jj[u]] = 1u;
		}
		__syncthreads();
		if (solo)
		{
			if (!t_stale[0])
			{
				if (tid < 64)
				{
					hnsw_link(nbrs, ncount, t_blk[0], t_cl[0], m, stride, sel, t_nsel[0], R.stamp0, R.stampU, round);
					hnsw_publish();
				}
				cur += 1;
			}
			else
				stopped = true;
			__syncthreads();
			continue;
		}

		/* ---- requests, sorted by (node, level, walk) ---- */
		const uint32_t nreq = C * (ksel + 1u);
		uint32_t	npad = 2;

		while (npad < nreq)
			npad <<= 1;
		for (uint32_t e = tid; e < npad; e += 256)
		{
			uint64_t	kv = ~0ull;

			if (e < nreq)
			{
				const uint32_t j = e / (ksel + 1u), idx = e % (ksel + 1u);

				if (idx < t_nsel[j])
					kv = hc_key(sel[j * NDB_HC_MAXSEL + idx], t_cl[j], j, 0u);
				else if (idx == ksel && t_nsel[j] > 0)
					kv = hc_key(t_blk[j], t_cl[j], j, 1u);	/* the node's own list */
			}
			key[e] = kv;
		}
		for (uint32_t size = 2; size <= npad; size <<= 1)
			for (uint32_t sd = size >> 1; sd > 0; sd >>= 1)
			{
				__syncthreads();
				for (uint32_t t = tid; t < (npad >> 1); t += 256)
				{
					const uint32_t lo = 2 * t - (t & (sd - 1));
					const uint32_t hi = lo + sd;
					const bool	up = ((lo & size) == 0);
					const uint64_t a = key[lo], b = key[hi];

					if ((a > b) == up)
					{
						key[lo] = b;
						key[hi] = a;
					}
				}
			}
		__syncthreads();

		/* ---- every list replays its requests in walk order ---- */
		for (uint32_t i = tid; i < npad; i += 256)
		{
			const uint64_t k0 = key[i];

			if (k0 == ~0ull || (i > 0 && (key[i - 1] >> 12) == (k0 >> 12)))
				continue;
			const uint32_t X = (uint32_t) (k0 >> 16);
			const int	cl = (int) ((k0 >> 12) & 15u);
			const uint32_t *nn = nbrs + (size_t) X * stride + (size_t) cl * m2;
			/* count and all 2m slots in one round trip (plain loads: every wave passed hnsw_publish's acquire
			 * after the previous chunk's stores); the holes below the count become a bit mask */
			const int16_t craw = ncount[(size_t) X * NDBHIP_HNSW_MAX_LEVEL + cl];
			unsigned long long inv = 0ull;

#pragma unroll 16
			for (int q = 0; q < m2; q++)
				inv |= (unsigned long long) (nn[q] == NDBHIP_INVALID_BLOCK) << q;
			int			c0 = hnsw_clamp(craw, m);
			int			cnt = c0;
			unsigned long long holes = c0 >= 64 ? inv : (inv & ((1ull << c0) - 1ull));
			uint32_t	firstw = 0xFFFFu;

			cnt0s[i] = (uint8_t) c0;
			for (uint32_t r = i; r < npad && (key[r] >> 12) == (k0 >> 12); r++)
			{
				const uint32_t j = (uint32_t) (key[r] >> 4) & 0xFFu;

				if (key[r] & 1u)
				{
					/* own list: slots 0..nsel-1 written, count = nsel (:2452-2456) */
					cnt = (int) t_nsel[j];
					holes = 0ull;
					pos[r] = NDB_HC_NONE;
					firstw = min(firstw, j);
					continue;
				}
				int			p;

				if (holes)				/* first InvalidBlockNumber among the first `count` slots (:2487-2511) */
				{
					p = __ffsll((long long) holes) - 1;
					holes &= holes - 1;
				}
				else
					p = cnt;
				if (p < m2)
				{
					pos[r] = (uint8_t) p;
					if (p >= cnt)
						cnt = p + 1;
					firstw = min(firstw, j);
				}
				else
					pos[r] = NDB_HC_NONE;
			}
			fw[i] = (uint16_t) firstw;
		}
		__syncthreads();

		/* ---- stale against the chunk's own earlier walks ---- */
		for (uint32_t base = 0; base < npairs; base += 256u * 8u)
		{
			uint32_t	enc[8], jj[8];

#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t p = base + (uint32_t) u * 256u + tid;

				jj[u] = 0xFFFFFFFFu;
				enc[u] = 0;
				if (p < npairs)
				{
					jj[u] = pair_walk(p);
					enc[u] = R.rs[(size_t) (cur + jj[u]) * NDB_HNSW_RS_CAP + (p - t_off[jj[u]])];
				}
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				if (jj[u] == 0xFFFFFFFFu || jj[u] == 0)
					continue;
				const uint64_t want = ((uint64_t) (enc[u] & ((1u << NDB_HNSW_RS_NODE_BITS) - 1u)) << 4) |
					(enc[u] >> NDB_HNSW_RS_NODE_BITS);	/* (node, level) = key >> 12 */
				uint32_t	lo = 0, hi = npad;

				while (lo < hi)
				{
					const uint32_t mid = (lo + hi) >> 1;

					if ((key[mid] >> 12) < want)
						lo = mid + 1;
					else
						hi = mid;
				}
				if (lo < npad && (key[lo] >> 12) == want && fw[lo] < jj[u])
					t_stale[jj[u]] = 1u;
			}
		}
		__syncthreads();
		if (tid < C && t_stale[tid])
			atomicMin(&s_stop, tid);
		__syncthreads();
		const uint32_t stop = s_stop;

		/* ---- apply the walks before `stop` ---- */
		for (uint32_t i = tid; i < npad; i += 256)
		{
			const uint64_t k0 = key[i];

			if (k0 == ~0ull)
				continue;
			const uint32_t X = (uint32_t) (k0 >> 16);
			const int	cl = (int) ((k0 >> 12) & 15u);
			const uint32_t j = (uint32_t) (k0 >> 4) & 0xFFu;

			if (j < stop && !(k0 & 1u) && pos[i] != NDB_HC_NONE)
				gstore(&nbrs[(size_t) X * stride + (size_t) cl * m2 + pos[i]], t_blk[j]);
			if (i > 0 && (key[i - 1] >> 12) == (k0 >> 12))
				continue;
			/* run head: the list's final count and its stamp */
			int			cnt = cnt0s[i];
			bool		wrote = false;

			for (uint32_t r = i; r < npad && (key[r] >> 12) == (k0 >> 12); r++)
			{
				const uint32_t jr = (uint32_t) (key[r] >> 4) & 0xFFu;

				if (jr >= stop)
					break;
				if (key[r] & 1u)
				{
					cnt = (int) t_nsel[jr];
					wrote = true;
				}
				else if (pos[r] != NDB_HC_NONE)
				{
					if ((int) pos[r] >= cnt)
						cnt = (int) pos[r] + 1;
					wrote = true;
				}
			}
			if (wrote)
			{
				gstore(&ncount[(size_t) X * NDBHIP_HNSW_MAX_LEVEL + cl], (int16_t) cnt);
				gstore(cl ? &R.stampU[X] : &R.stamp0[X], round);
			}
		}
		for (uint32_t e = tid; e < stop * ksel; e += 256)
		{
			const uint32_t j = e / ksel, idx = e % ksel;

			if (idx < t_nsel[j])
				gstore(&nbrs[(size_t) t_blk[j] * stride + (size_t) t_cl[j] * m2 + idx], sel[j * NDB_HC_MAXSEL + idx]);
		}
		hnsw_publish();
		__syncthreads();
		cur += stop;
		if (stop < C)
			stopped = true;
	}
	if (tid == 0)
	{
		*R.next = cur;
		if (cur < ntasks)
			atomicAdd(&R.stats[1], 1ull);
	}
}

#define NDB_HH_BITS 11
#define NDB_HH_SLOTS (1u << NDB_HH_BITS)	/* >= 2 x the chunk's distinct lists (64 walks x 17) */
#define NDB_HH_MAXSEL 16u					/* ksel <= 16 and 2m <= 32: the default m = 16 */

/*
 * The chunked commit without the sort: with at most 64 walks per chunk "who back-links into this list, in
 * walk order" is one 64-bit mask per list, kept in an LDS hash table keyed by (node, level).  A list's free
 * places are known up front — its holes below the count, then the tail up to 2m — so request number r (the
 * r-th set bit of the mask) lands in the r-th free place or is dropped, in closed form; no replay loop, no
 * sort, and the first writer of a list is the mask's lowest bit (if there is room at all).
 */
__global__ __launch_bounds__(256) void
k_hnsw_commit_hash(int16_t *ncount, uint32_t *nbrs, const HnswTask *__restrict__ tasks, uint32_t ntasks, int m,
				  uint32_t ksel, HnswRounds R, uint32_t round)
{
	__shared__ uint64_t tkey[NDB_HH_SLOTS];		/* (node << 4 | level) + 1, 0 = empty */
	__shared__ uint64_t tmask[NDB_HH_SLOTS];	/* walks of the chunk that back-link into this list */
	__shared__ uint32_t tholes[NDB_HH_SLOTS];	/* InvalidBlockNumber slots below the list's count */
	__shared__ uint8_t tcnt0[NDB_HH_SLOTS];		/* the list's count before the chunk (own list: the walk's nsel) */
	__shared__ uint8_t tfw[NDB_HH_SLOTS];		/* first walk of the chunk that writes the list, NDB_HC_NONE = none */
	__shared__ uint8_t town[NDB_HH_SLOTS];		/* the walk whose own list this is, NDB_HC_NONE = nobody's */
	__shared__ uint16_t rslot[NDB_HC_TASKS * (NDB_HH_MAXSEL + 1)];
	__shared__ uint32_t sel[NDB_HC_TASKS * NDB_HC_MAXSEL];
	__shared__ uint32_t t_ran[NDB_HC_TASKS], t_rsn[NDB_HC_TASKS], t_nsel[NDB_HC_TASKS], t_blk[NDB_HC_TASKS];
	__shared__ int t_cl[NDB_HC_TASKS];
	__shared__ uint32_t t_stale[NDB_HC_TASKS], t_off[NDB_HC_TASKS + 1];
	__shared__ uint32_t s_stop, s_self;
	const uint32_t tid = threadIdx.x;
	const int	m2 = 2 * m;
	const int64_t stride = (int64_t) NDBHIP_HNSW_MAX_LEVEL * m2;
	const uint32_t first = *R.next;
	const uint32_t cmax = NDB_HC_TASKS;			/* <= 64 walks: one bit each in tmask */
	uint32_t	cur = first;
	bool		stopped = false;

	while (cur < ntasks && !stopped)
	{
		uint32_t	C = min(cmax, ntasks - cur);

		/* ---- the chunk's walks ---- */
		if (tid < C)
		{
			const uint32_t t = cur + tid;
			const HnswTask task = tasks[t];

			t_ran[tid] = R.spec_round[t];
			t_rsn[tid] = R.rsn[t];
			t_nsel[tid] = (uint32_t) R.nsel[t];
			t_blk[tid] = task.row + 1;
			t_cl[tid] = task.cl;
			/* never run, or its read-set log overflowed: cannot be validated (unless it opens the round) */
			t_stale[tid] = (t_ran[tid] == 0 || t_rsn[tid] > NDB_HNSW_RS_CAP) ? 1u : 0u;
		}
		if (tid == 0)
		{
			s_stop = C;
			s_self = C;
		}
		__syncthreads();
		for (uint32_t e = tid; e < C * ksel; e += 256)
		{
			const uint32_t j = e / ksel, idx = e % ksel;

			if (idx < t_nsel[j])
			{
				const uint32_t v = R.sel[(size_t) (cur + j) * ksel + idx];

				sel[j * NDB_HC_MAXSEL + idx] = v;
				if (v == t_blk[j])
					atomicMin(&s_self, j);
			}
		}
		__syncthreads();
		const bool	solo = s_self == 0;	/* the chunk's first walk selected its own node: commit it alone */

		if (solo)
			C = 1;
		else if (s_self < C)
			C = s_self;					/* ... a later one: it will open the next chunk */
		const bool	opens_round = cur == first && t_ran[0] == round;	/* valid by construction */

		if (tid == 0)
		{
			s_stop = C;
			if (opens_round)
				t_stale[0] = 0;
		}
		__syncthreads();

		/* ---- stale against what was written before this chunk ---- */
		/* (walk, read-set entry) pairs are spread over the block, 8 per thread in flight */
		if (tid == 0)
		{
			uint32_t	acc = 0;

			for (uint32_t j = 0; j < C; j++)
			{
				t_off[j] = acc;
				acc += t_rsn[j] > NDB_HNSW_RS_CAP ? 0u : t_rsn[j];
			}
			t_off[C] = acc;
		}
		__syncthreads();
		const uint32_t npairs = t_off[C];
		auto		pair_walk = [&](uint32_t p) -> uint32_t {	/* largest j with t_off[j] <= p */
			uint32_t	lo = 0, hi = C;

			while (hi - lo > 1)
			{
				const uint32_t mid = (lo + hi) >> 1;

				if (t_off[mid] <= p)
					lo = mid;
				else
					hi = mid;
			}
			return lo;
		};

		for (uint32_t base = 0; base < npairs; base += 256u * 8u)
		{
			uint32_t	enc[8], jj[8], stv[8];

#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t p = base + (uint32_t) u * 256u + tid;

				jj[u] = 0xFFFFFFFFu;
				enc[u] = 0;
				if (p < npairs)
				{
					jj[u] = pair_walk(p);
					enc[u] = R.rs[(size_t) (cur + jj[u]) * NDB_HNSW_RS_CAP + (p - t_off[jj[u]])];
				}
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t node = enc[u] & ((1u << NDB_HNSW_RS_NODE_BITS) - 1u);
				const uint32_t *st = (enc[u] >> NDB_HNSW_RS_NODE_BITS) ? R.stampU : R.stamp0;

				stv[u] = jj[u] != 0xFFFFFFFFu ? gload<true>(&st[node]) : 0u;
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
				if (jj[u] != 0xFFFFFFFFu && !(jj[u] == 0 && opens_round) && stv[u] >= t_ran[jj[u]])
					t_stale[jj[u]] = 1u;
		}
		__syncthreads();
		if (solo)
		{
			if (!t_stale[0])
			{
				if (tid < 64)
				{
					hnsw_link(nbrs, ncount, t_blk[0], t_cl[0], m, stride, sel, t_nsel[0], R.stamp0, R.stampU, round);
					hnsw_publish();
				}
				cur += 1;
			}
			else
				stopped = true;
			__syncthreads();
			continue;
		}

		/* ---- the chunk's requests, hashed by (node, level): who asks, in walk order, is a bit mask ---- */
		for (uint32_t i = tid; i < NDB_HH_SLOTS; i += 256)
		{
			tkey[i] = 0ull;
			tmask[i] = 0ull;
			town[i] = NDB_HC_NONE;
			tfw[i] = NDB_HC_NONE;
		}
		__syncthreads();
		auto		slot_of = [&](uint32_t node, uint32_t level, bool insert) -> uint32_t {
			const uint64_t kv = (((uint64_t) node << 4) | level) + 1ull;
			uint32_t	h = (uint32_t) ((kv * 0x9E3779B97F4A7C15ull) >> (64 - NDB_HH_BITS));

			for (;;)
			{
				uint64_t	cur = tkey[h];

				if (cur == kv)
					return h;
				if (cur == 0ull)
				{
					if (!insert)
						return NDB_HH_SLOTS;
					cur = atomicCAS((unsigned long long *) &tkey[h], 0ull, (unsigned long long) kv);
					if (cur == 0ull || cur == kv)
						return h;
				}
				h = (h + 1u) & (NDB_HH_SLOTS - 1u);
			}
		};
		const uint32_t nreq = C * (ksel + 1u);

		for (uint32_t e = tid; e < nreq; e += 256)
		{
			const uint32_t j = e / (ksel + 1u), idx = e % (ksel + 1u);

			if (idx < t_nsel[j])
			{
				const uint32_t sl = slot_of(sel[j * NDB_HC_MAXSEL + idx], (uint32_t) t_cl[j], true);

				atomicOr((unsigned long long *) &tmask[sl], 1ull << j);
				rslot[e] = (uint16_t) sl;
			}
			else if (idx == ksel && t_nsel[j] > 0)
			{
				const uint32_t sl = slot_of(t_blk[j], (uint32_t) t_cl[j], true);	/* the node's own list */

				town[sl] = (uint8_t) j;
			}
		}
		__syncthreads();

		/* ---- per list: what it holds now, hence which requests will write and where ---- */
		for (uint32_t i = tid; i < NDB_HH_SLOTS; i += 256)
		{
			const uint64_t kv = tkey[i];

			if (kv == 0ull)
				continue;
			const uint32_t X = (uint32_t) ((kv - 1ull) >> 4);
			const int	cl = (int) ((kv - 1ull) & 15ull);
			const uint32_t *nn = nbrs + (size_t) X * stride + (size_t) cl * m2;
			int			c0;
			uint32_t	holes = 0;

			if (town[i] != NDB_HC_NONE)
			{
				c0 = (int) t_nsel[town[i]];		/* slots 0..nsel-1 written, count = nsel (:2452-2456) */
				tfw[i] = town[i];
			}
			else
			{
				/* count and the 2m slots in one round trip (plain loads: every wave passed hnsw_publish's
				 * acquire after the previous chunk's stores) */
				const int16_t craw = ncount[(size_t) X * NDBHIP_HNSW_MAX_LEVEL + cl];
				uint32_t	inv = 0;

#pragma unroll 16
				for (int q = 0; q < m2; q++)
					inv |= (uint32_t) (nn[q] == NDBHIP_INVALID_BLOCK) << q;
				c0 = hnsw_clamp(craw, m);
				holes = c0 >= 32 ? inv : (inv & ((1u << c0) - 1u));
				if (tmask[i] != 0ull && (__popc(holes) + (m2 - c0)) > 0)
					tfw[i] = (uint8_t) (__ffsll((long long) tmask[i]) - 1);
			}
			tcnt0[i] = (uint8_t) c0;
			tholes[i] = holes;
		}
		__syncthreads();

		/* ---- stale against the chunk's own earlier walks ---- */
		for (uint32_t base = 0; base < npairs; base += 256u * 8u)
		{
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const uint32_t p = base + (uint32_t) u * 256u + tid;

				if (p >= npairs)
					continue;
				const uint32_t j = pair_walk(p);

				if (j == 0)
					continue;
				const uint32_t enc = R.rs[(size_t) (cur + j) * NDB_HNSW_RS_CAP + (p - t_off[j])];
				const uint32_t sl = slot_of(enc & ((1u << NDB_HNSW_RS_NODE_BITS) - 1u), enc >> NDB_HNSW_RS_NODE_BITS,
											false);

				if (sl < NDB_HH_SLOTS && tfw[sl] < j)
					t_stale[j] = 1u;
			}
		}
		__syncthreads();
		if (tid < C && t_stale[tid])
			atomicMin(&s_stop, tid);
		__syncthreads();
		const uint32_t stop = s_stop;
		const uint64_t below_stop = stop >= 64 ? ~0ull : ((1ull << stop) - 1ull);

		/* ---- apply the walks before `stop`: every request knows its rank among the list's requests ---- */
		for (uint32_t e = tid; e < stop * (ksel + 1u); e += 256)
		{
			const uint32_t j = e / (ksel + 1u), idx = e % (ksel + 1u);

			if (idx >= t_nsel[j])
				continue;
			const uint32_t sl = rslot[e];
			const uint64_t kv = tkey[sl] - 1ull;
			const uint32_t X = (uint32_t) (kv >> 4);
			const int	cl = (int) (kv & 15ull);
			const uint32_t r = (uint32_t) __popcll(tmask[sl] & ((1ull << j) - 1ull));
			uint32_t	holes = tholes[sl];
			const uint32_t nh = (uint32_t) __popc(holes);
			int			p;

			if (r < nh)			/* first InvalidBlockNumber among the first `count` slots (:2487-2511) */
			{
				for (uint32_t z = 0; z < r; z++)
					holes &= holes - 1;
				p = __ffs((int) holes) - 1;
			}
			else
				p = (int) tcnt0[sl] + (int) (r - nh);
			if (p < m2)
				gstore(&nbrs[(size_t) X * stride + (size_t) cl * m2 + p], t_blk[j]);
			/* the node's own list: sel is what it links to */
			gstore(&nbrs[(size_t) t_blk[j] * stride + (size_t) t_cl[j] * m2 + idx], sel[j * NDB_HC_MAXSEL + idx]);
		}
		for (uint32_t i = tid; i < NDB_HH_SLOTS; i += 256)
		{
			const uint64_t kv = tkey[i];

			if (kv == 0ull)
				continue;
			const uint32_t X = (uint32_t) ((kv - 1ull) >> 4);
			const int	cl = (int) ((kv - 1ull) & 15ull);
			const bool	own = town[i] != NDB_HC_NONE && town[i] < stop;
			const int	nh = __popc(tholes[i]);
			const int	w = __popcll(tmask[i] & below_stop);	/* requests of committed walks, in order */
			const int	room = nh + (m2 - (int) tcnt0[i]);
			const int	writes = w < room ? w : room;

			if (town[i] != NDB_HC_NONE && !own)
				continue;			/* this node's own walk did not commit: nothing of its list exists yet */
			if (writes > 0 || own)
			{
				const int	appended = writes > nh ? writes - nh : 0;

				gstore(&ncount[(size_t) X * NDBHIP_HNSW_MAX_LEVEL + cl], (int16_t) ((int) tcnt0[i] + appended));
				gstore(cl ? &R.stampU[X] : &R.stamp0[X], round);
			}
		}
		hnsw_publish();
		__syncthreads();
		cur += stop;
		if (stop < C)
			stopped = true;
	}
	if (tid == 0)
	{
		*R.next = cur;
		if (cur < ntasks)
			atomicAdd(&R.stats[1], 1ull);
	}
}

int
set_kernel_attributes_hnsw()
{
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_search<R_HNSW_L2>, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_search<R_HNSW_COS>, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_search<R_HNSW_IP>, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_search_fast<R_HNSW_L2>, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_search_fast<R_HNSW_COS>, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_search_fast<R_HNSW_IP>, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_build, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_spec<false>, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_spec<true>, hipFuncAttributeMaxDynamicSharedMemorySize, NDB_TOPK_MAX_SMEM));
	return NDBHIP_OK;
}

struct ndbhip_hnsw
{
	int64_t		build_stats[6] = {0, 0, 0, 0, 0, 0};
	int			dim = 0, m = 0;
	uint32_t	nblocks = 0;
	uint32_t	entry_point = NDBHIP_INVALID_BLOCK;
	int			entry_level = -1;
	float	   *d_vecs = nullptr;
	uint16_t   *d_vecs16 = nullptr;		/* walk rows of the intended search (made on first use, ndbhip_hnsw2.h) */
	uint32_t	w16_blocks = 0;			/* blocks they cover: fewer than nblocks = stale (rows were appended) */
	/* the intended search under strategy 2 (cosine): 1 / |row| of every node, for the float4 rows [0] and the walk rows [1]
	 * (made on first use; stale like the walk rows) */
	double	   *d_rinv[2] = {nullptr, nullptr};
	uint32_t	rinv_blocks[2] = {0, 0};
	int		   *d_levels = nullptr;
	int16_t    *d_ncount = nullptr;
	int64_t    *d_nbr_off = nullptr;
	uint32_t   *d_nbrs = nullptr;
	uint64_t   *d_tids = nullptr;
	uint8_t    *d_dead = nullptr;		/* [nblocks] line pointer marked dead by bulkdelete (allocated on first use) */
	uint32_t	cap_blocks = 0;			/* blocks the dense arrays have room for (hnswinsert grows them geometrically) */
	int			ef_construction = 200;	/* HnswMetaPageData.efConstruction / efSearch (hnsw_am.c:108-120), defaults :82-83 */
	int			ef_search = 64;
	bool		loaded = false;
	bool		dense = false;			/* neighbour slots in the 16-level dense layout (device-built graphs) */
	/* host-call workspace */
	float	   *w_q = nullptr;		size_t w_q_n = 0;
	uint32_t   *w_ob = nullptr;		size_t w_ob_n = 0;
	float	   *w_od = nullptr;		size_t w_od_n = 0;
	int		   *w_oc = nullptr;		size_t w_oc_n = 0;
	uint64_t   *w_ot = nullptr;		size_t w_ot_n = 0;
	long long  *w_os = nullptr;		size_t w_os_n = 0;
	uint32_t   *w_vbits = nullptr;	size_t w_vbits_n = 0;	/* hnsw_search_layer: per-block visited bitmaps, all-zero at rest */
	uint32_t   *w_vlog = nullptr;	size_t w_vlog_n = 0;
	void	   *pin = nullptr;		size_t pin_n = 0;		/* pinned host block of the host-pointer search: queries + results */
	/* ndbhip_hnsw_share: a second handle on the same graph with a workspace of its own (see ndbhip_ivf_share) */
	ndbhip_hnsw *shared_of = nullptr;
	int			nshares = 0;
};

static inline bool
hnsw_frozen(const ndbhip_hnsw *h)
{
	return h->shared_of != nullptr || h->nshares > 0;
}
#define HNSW_NOT_FROZEN(h, what)                                                                                        \
	do {                                                                                                                \
		if ((h) && hnsw_frozen(h))                                                                                      \
			return fail(NDBHIP_ERR_STATE, "%s: the graph is shared (ndbhip_hnsw_share): destroy the shares first%s", what, \
						(h)->shared_of ? ", and do this on the handle they were made from" : "");                        \
	} while (0)

extern "C" int
ndbhip_hnsw_create(int dim, int m, ndbhip_hnsw **out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!out || dim < 1 || dim > 32767)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (m < 2 || m > 128)		/* HNSW_MIN_M / HNSW_MAX_M: hnsw_am.c:90-91 */
		return fail(NDBHIP_ERR_INVALID, "m %d out of range 2..128", m);
	ndbhip_hnsw *g2 = new (std::nothrow) ndbhip_hnsw();

	if (!g2)
		return fail(NDBHIP_ERR_NOMEM, "out of host memory");
	g2->dim = dim;
	g2->m = m;
	*out = g2;
	return NDBHIP_OK;
}

static void
hnsw_free_dev(ndbhip_hnsw *h)
{
	void	   *ptrs[] = {h->d_vecs, h->d_levels, h->d_ncount, h->d_nbr_off, h->d_nbrs, h->d_tids, h->d_dead};

	for (void *p : ptrs)
		if (p) (void) hipFree(p);
	if (h->d_vecs16) (void) hipFree(h->d_vecs16);
	h->d_vecs16 = nullptr;
	h->w16_blocks = 0;
	for (int r = 0; r < 2; r++)
	{
		if (h->d_rinv[r]) (void) hipFree(h->d_rinv[r]);
		h->d_rinv[r] = nullptr;
		h->rinv_blocks[r] = 0;
	}
	h->d_dead = nullptr;
	h->cap_blocks = 0;
	h->d_vecs = nullptr; h->d_levels = nullptr; h->d_ncount = nullptr;
	h->d_nbr_off = nullptr; h->d_nbrs = nullptr; h->d_tids = nullptr;
	h->loaded = false;
}

extern "C" int
ndbhip_hnsw_destroy(ndbhip_hnsw *h)
{
	if (!h)
		return NDBHIP_OK;
	if (h->nshares > 0)
		return fail(NDBHIP_ERR_STATE, "ndbhip_hnsw_destroy: %d shares of this graph are alive (ndbhip_hnsw_share): destroy them first", h->nshares);
	if (h->shared_of)
	{
		if (g.inited)
		{
			(void) hipDeviceSynchronize();		/* (its last batch may have run on another thread's stream) */
			void	   *ptrs[] = {h->w_q, h->w_ob, h->w_od, h->w_oc, h->w_ot, h->w_os, h->w_vbits, h->w_vlog};

			for (void *p : ptrs)
				if (p) (void) hipFree(p);
			if (h->pin) (void) hipHostFree(h->pin);
		}
		{
			std::lock_guard<std::mutex> lk(ndbhip_g_mtx);

			h->shared_of->nshares--;
		}
		delete h;
		return NDBHIP_OK;
	}
	if (g.inited)
	{
		(void) hipStreamSynchronize(g.stream);
		hnsw_free_dev(h);
		void	   *ptrs[] = {h->w_q, h->w_ob, h->w_od, h->w_oc, h->w_ot, h->w_os, h->w_vbits, h->w_vlog};

		for (void *p : ptrs)
			if (p) (void) hipFree(p);
		if (h->pin) (void) hipHostFree(h->pin);
	}
	delete h;
	return NDBHIP_OK;
}

static int hnsw_densify(ndbhip_hnsw *h);

/* A second handle on the same graph (node rows, levels, neighbour lists, TIDs, the fp16 walk rows if they exist): a
 * workspace of its own — visited maps, result blocks — and nothing else.  Two batches of searches in flight (a host
 * thread, a stream — ndbhip_set_thread_stream — and a handle each): a batch ends with its longest walks, the next one's
 * fill the device meanwhile.  Both handles are frozen while the share lives (NDBHIP_ERR_STATE from loads, inserts, builds,
 * deletes); walk rows are made on the source by its first ndbhip_hnsw_search_intended_w16_device, before sharing. */
extern "C" int
ndbhip_hnsw_share(ndbhip_hnsw *src, ndbhip_hnsw **out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!src || !out)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!src->loaded)
		return fail(NDBHIP_ERR_STATE, "graph not loaded");
	if (src->shared_of)
		return fail(NDBHIP_ERR_STATE, "ndbhip_hnsw_share: make shares from the handle that owns the graph");
	if (!src->nshares)
	{
		const int	rc = hnsw_densify(src);		/* (the layout every search reads: made once, here, not under a share) */

		if (rc)
			return rc;
	}
	HIP_TRY(hipStreamSynchronize(g.stream));
	ndbhip_hnsw *h = new (std::nothrow) ndbhip_hnsw(*src);

	if (!h)
		return fail(NDBHIP_ERR_NOMEM, "out of host memory");
	h->w_q = nullptr; h->w_q_n = 0;
	h->w_ob = nullptr; h->w_ob_n = 0;
	h->w_od = nullptr; h->w_od_n = 0;
	h->w_oc = nullptr; h->w_oc_n = 0;
	h->w_ot = nullptr; h->w_ot_n = 0;
	h->w_os = nullptr; h->w_os_n = 0;
	h->w_vbits = nullptr; h->w_vbits_n = 0;
	h->w_vlog = nullptr; h->w_vlog_n = 0;
	h->pin = nullptr; h->pin_n = 0;
	h->shared_of = src;
	h->nshares = 0;
	{
		std::lock_guard<std::mutex> lk(ndbhip_g_mtx);		/* (shares may be made and destroyed by different threads) */

		src->nshares++;
	}
	*out = h;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_load(ndbhip_hnsw *h, uint32_t nblocks, const float *vecs, const int32_t *levels,
				 const int16_t *ncount, const int64_t *nbr_off, const uint32_t *nbrs, const uint8_t *tids6,
				 uint32_t entry_point, int entry_level)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	HNSW_NOT_FROZEN(h, "ndbhip_hnsw_load");
	if (!h || nblocks < 1 || !vecs || !levels || !ncount || !nbr_off || !tids6)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	const int64_t nn = nbr_off[nblocks];

	if (nn < 0 || (nn > 0 && !nbrs))
		return fail(NDBHIP_ERR_INVALID, "bad neighbour arrays");
	for (uint32_t b = 1; b < nblocks; b++)
	{
		if (levels[b] < 0 || levels[b] >= NDBHIP_HNSW_MAX_LEVEL)
			return fail(NDBHIP_ERR_INVALID, "node %u: level %d out of range", b, levels[b]);
		if (nbr_off[b + 1] - nbr_off[b] != (int64_t) (levels[b] + 1) * 2 * h->m)
			return fail(NDBHIP_ERR_INVALID, "node %u: neighbour slots do not match (level+1)*2m", b);
	}
	hnsw_free_dev(h);
	std::vector<uint64_t> t64(nblocks);

	for (uint32_t b = 0; b < nblocks; b++)
		t64[b] = ndb_tid_pack(tids6 + 6 * (size_t) b);
	HIP_TRY(hipMalloc((void **) &h->d_vecs, (size_t) nblocks * h->dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &h->d_levels, (size_t) nblocks * sizeof(int)));
	HIP_TRY(hipMalloc((void **) &h->d_ncount, (size_t) nblocks * 16 * sizeof(int16_t)));
	HIP_TRY(hipMalloc((void **) &h->d_nbr_off, (size_t) (nblocks + 1) * sizeof(int64_t)));
	HIP_TRY(hipMalloc((void **) &h->d_nbrs, (size_t) std::max<int64_t>(nn, 1) * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **) &h->d_tids, (size_t) nblocks * sizeof(uint64_t)));
	HIP_TRY(hipMemcpyAsync(h->d_vecs, vecs, (size_t) nblocks * h->dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(h->d_levels, levels, (size_t) nblocks * sizeof(int), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(h->d_ncount, ncount, (size_t) nblocks * 16 * sizeof(int16_t), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(h->d_nbr_off, nbr_off, (size_t) (nblocks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, g.stream));
	if (nn > 0)
		HIP_TRY(hipMemcpyAsync(h->d_nbrs, nbrs, (size_t) nn * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(h->d_tids, t64.data(), (size_t) nblocks * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	h->nblocks = nblocks;
	h->cap_blocks = nblocks;
	h->entry_point = entry_point;
	h->entry_level = entry_level;
	h->loaded = true;
	h->dense = false;
	return NDBHIP_OK;
}

static int hnsw_densify(ndbhip_hnsw *h);

/* room for nb blocks in the dense arrays of a mirror that holds base nodes (blocks 0 .. base): the relation grows by pages;
 * the arrays grow geometrically so that a stream of single-row hnswinsert calls does not copy the graph every time */
static int
hnsw_grow_dense(ndbhip_hnsw *h, uint32_t base, uint32_t nb)
{
	const size_t stride = (size_t) NDBHIP_HNSW_MAX_LEVEL * 2 * h->m;
	{
		/* the relation grows by n pages; the arrays grow geometrically so that a stream of single-row
		 * hnswinsert calls does not copy the graph every time */
		int			rc = hnsw_densify(h);

		if (rc)
			return rc;
		const uint32_t ob = base + 1;

		if (h->cap_blocks < nb)
		{
			const uint64_t want = std::max<uint64_t>(nb, (uint64_t) h->cap_blocks + h->cap_blocks / 2 + 1024);
			const uint32_t cap = (uint32_t) std::min<uint64_t>(want, 0xFFFFFFF0ull);
			float	   *nv = nullptr;
			int		   *nl = nullptr;
			int16_t    *nc = nullptr;
			uint32_t   *nn = nullptr;
			uint64_t   *nt = nullptr;

			HIP_TRY(hipMalloc((void **) &nv, (size_t) cap * h->dim * sizeof(float)));
			HIP_TRY(hipMalloc((void **) &nl, (size_t) cap * sizeof(int)));
			HIP_TRY(hipMalloc((void **) &nc, (size_t) cap * 16 * sizeof(int16_t)));
			HIP_TRY(hipMalloc((void **) &nn, (size_t) cap * stride * sizeof(uint32_t)));
			HIP_TRY(hipMalloc((void **) &nt, (size_t) cap * sizeof(uint64_t)));
			HIP_TRY(hipMemcpyAsync(nv, h->d_vecs, (size_t) ob * h->dim * sizeof(float), hipMemcpyDeviceToDevice, g.stream));
			HIP_TRY(hipMemcpyAsync(nl, h->d_levels, (size_t) ob * sizeof(int), hipMemcpyDeviceToDevice, g.stream));
			HIP_TRY(hipMemcpyAsync(nc, h->d_ncount, (size_t) ob * 16 * sizeof(int16_t), hipMemcpyDeviceToDevice, g.stream));
			HIP_TRY(hipMemcpyAsync(nn, h->d_nbrs, (size_t) ob * stride * sizeof(uint32_t), hipMemcpyDeviceToDevice, g.stream));
			HIP_TRY(hipMemcpyAsync(nt, h->d_tids, (size_t) ob * sizeof(uint64_t), hipMemcpyDeviceToDevice, g.stream));
			if (h->d_dead)
			{
				uint8_t    *nd = nullptr;

				HIP_TRY(hipMalloc((void **) &nd, (size_t) cap));
				HIP_TRY(hipMemsetAsync(nd, 0, (size_t) cap, g.stream));
				HIP_TRY(hipMemcpyAsync(nd, h->d_dead, (size_t) ob, hipMemcpyDeviceToDevice, g.stream));
				HIP_TRY(hipStreamSynchronize(g.stream));
				HIP_TRY(hipFree(h->d_dead));
				h->d_dead = nd;
			}
			HIP_TRY(hipStreamSynchronize(g.stream));
			HIP_TRY(hipFree(h->d_vecs)); HIP_TRY(hipFree(h->d_levels)); HIP_TRY(hipFree(h->d_ncount));
			HIP_TRY(hipFree(h->d_nbrs)); HIP_TRY(hipFree(h->d_tids));
			h->d_vecs = nv; h->d_levels = nl; h->d_ncount = nc; h->d_nbrs = nn; h->d_tids = nt;
			h->cap_blocks = cap;
		}
	}
	return 0;
}

/* hnswInsertNode for rows 0..n-1 on top of the `base` nodes the mirror already holds (0: build from nothing) */
static int
hnsw_insert_rows(ndbhip_hnsw *h, const float *d_rows, const uint64_t *d_tids, uint32_t n, const int32_t *levels,
				 int ef_construction, uint32_t base)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!h || !d_rows || !d_tids || !levels || n < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (ef_construction < 4 || ef_construction > NDBHIP_MAX_EF)	/* HNSW_MIN_EF_CONSTRUCTION: hnsw_am.c:92 */
		return fail(NDBHIP_ERR_INVALID, "ef_construction %d out of range 4..%d", ef_construction, NDBHIP_MAX_EF);
	const size_t smem = hnsw_smem_bytes((uint32_t) ef_construction, (uint32_t) ef_construction, (uint32_t) h->m);

	if (smem > NDB_TOPK_MAX_SMEM)
		return fail(NDBHIP_ERR_UNSUPPORTED, "ef_construction too large for the LDS-resident candidate set");
	if ((uint64_t) base + n + 1 > 0xFFFFFFF0ull)
		return fail(NDBHIP_ERR_UNSUPPORTED, "more than 2^32 blocks");
	const uint32_t nb = base + n + 1;
	const size_t stride = (size_t) NDBHIP_HNSW_MAX_LEVEL * 2 * h->m;
	int		   *d_lv_in = nullptr;
	uint32_t   *d_entry = nullptr;
	uint32_t	entry[2] = {NDBHIP_INVALID_BLOCK, (uint32_t) -1};

	if (base == 0)
	{
		hnsw_free_dev(h);
		HIP_TRY(hipMalloc((void **) &h->d_vecs, (size_t) nb * h->dim * sizeof(float)));
		HIP_TRY(hipMalloc((void **) &h->d_levels, (size_t) nb * sizeof(int)));
		HIP_TRY(hipMalloc((void **) &h->d_ncount, (size_t) nb * 16 * sizeof(int16_t)));
		HIP_TRY(hipMalloc((void **) &h->d_nbrs, (size_t) nb * stride * sizeof(uint32_t)));
		HIP_TRY(hipMalloc((void **) &h->d_tids, (size_t) nb * sizeof(uint64_t)));
		HIP_TRY(hipMemsetAsync(h->d_vecs, 0, (size_t) h->dim * sizeof(float), g.stream));	/* row 0 = meta page */
		HIP_TRY(hipMemsetAsync(h->d_levels, 0, sizeof(int), g.stream));
		HIP_TRY(hipMemsetAsync(h->d_ncount, 0, 16 * sizeof(int16_t), g.stream));
		HIP_TRY(hipMemsetAsync(h->d_nbrs, 0xFF, stride * sizeof(uint32_t), g.stream));
		HIP_TRY(hipMemsetAsync(h->d_tids, 0, sizeof(uint64_t), g.stream));
		h->cap_blocks = nb;
	}
	else
	{
		int			rc = hnsw_grow_dense(h, base, nb);

		if (rc)
			return rc;
		entry[0] = h->entry_point;
		entry[1] = (uint32_t) h->entry_level;
	}
	HIP_TRY(hipMalloc((void **) &d_lv_in, (size_t) n * sizeof(int)));
	HIP_TRY(hipMalloc((void **) &d_entry, 2 * sizeof(uint32_t)));
	HIP_TRY(hipMemcpyAsync(d_lv_in, levels, (size_t) n * sizeof(int), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_entry, entry, sizeof(entry), hipMemcpyHostToDevice, g.stream));
	const bool	spec = g_hnsw_spec && nb < (1u << NDB_HNSW_RS_NODE_BITS);

	memset(h->build_stats, 0, sizeof(h->build_stats));
	if (!spec)
	{
		hipLaunchKernelGGL(k_hnsw_build, dim3(1), dim3(64), smem, g.stream, h->d_vecs, h->d_levels, h->d_ncount,
						   h->d_nbrs, h->d_tids, d_rows, d_tids, (const int *) d_lv_in, n, h->dim, h->m,
						   (uint32_t) ef_construction, d_entry, base);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(entry, d_entry, sizeof(entry), hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
	}
	else
	{
		/*
		 * The entry point is a pure function of the drawn levels (Step 6, :2642-2663: the first node of each
		 * new maximum level), so the host knows it for every insert and cuts the batches so that it is
		 * constant inside one.
		 */
		const uint32_t ksel = (uint32_t) std::min(h->m, ef_construction);
		std::vector<HnswTask> tasks;
		struct Batch { size_t t0, t1; uint32_t entry; int entry_level; };
		std::vector<Batch> batches;
		uint32_t	e_pt = entry[0];
		int			e_lv = (int) entry[1];
		size_t		maxb = 0;

		tasks.reserve((size_t) n + n / 8);
		for (uint32_t i = 0; i < n;)
		{
			const size_t want = std::min<size_t>((size_t) g_hnsw_batch_max,
												 std::max<size_t>(1, ((size_t) base + i) / (size_t) g_hnsw_batch_div));
			Batch		b{tasks.size(), tasks.size(), e_pt, e_lv};

			while (i < n && tasks.size() - b.t0 < want)
			{
				int			level = levels[i];

				if (level >= NDBHIP_HNSW_MAX_LEVEL) level = NDBHIP_HNSW_MAX_LEVEL - 1;
				if (level < 0) level = 0;
				if (e_pt != NDBHIP_INVALID_BLOCK && e_lv >= 0)
					for (int cl = std::min(level, e_lv); cl >= 0; cl--)
						tasks.push_back(HnswTask{base + i, cl});
				i++;
				if (e_pt == NDBHIP_INVALID_BLOCK || level > e_lv)
				{
					e_pt = base + i;	/* block of row i-1 */
					e_lv = level;
					break;		/* the entry point changes: close the batch */
				}
			}
			b.t1 = tasks.size();
			if (b.t1 > b.t0)
				batches.push_back(b);
			maxb = std::max(maxb, b.t1 - b.t0);
		}
		entry[0] = e_pt;
		entry[1] = (uint32_t) e_lv;

		HnswTask   *d_tasks = nullptr;
		uint32_t   *d_u32 = nullptr;
		unsigned long long *d_stats = nullptr;
		const size_t ntot = std::max<size_t>(tasks.size(), 1);
		HnswRounds	R;

		maxb = std::max<size_t>(maxb, 1);
		/* one allocation: next | spec_round | nsel | rsn | sel | rs | stamp0 | stampU */
		const size_t n_u32 = 1 + 3 * maxb + maxb * ksel + maxb * NDB_HNSW_RS_CAP + (size_t) 2 * nb;

		HIP_TRY(hipMalloc((void **) &d_tasks, ntot * sizeof(HnswTask)));
		HIP_TRY(hipMalloc((void **) &d_u32, n_u32 * sizeof(uint32_t)));
		HIP_TRY(hipMalloc((void **) &d_stats, 4 * sizeof(unsigned long long)));
		R.next = d_u32;
		R.spec_round = R.next + 1;
		R.nsel = (int *) (R.spec_round + maxb);
		R.rsn = (uint32_t *) R.nsel + maxb;
		R.sel = R.rsn + maxb;
		R.rs = R.sel + maxb * ksel;
		R.stamp0 = R.rs + maxb * NDB_HNSW_RS_CAP;
		R.stampU = R.stamp0 + nb;
		R.stats = d_stats;
		HIP_TRY(hipMemsetAsync(R.stamp0, 0, (size_t) 2 * nb * sizeof(uint32_t), g.stream));
		HIP_TRY(hipMemsetAsync(d_stats, 0, 4 * sizeof(unsigned long long), g.stream));
		if (!tasks.empty())
			HIP_TRY(hipMemcpyAsync(d_tasks, tasks.data(), tasks.size() * sizeof(HnswTask), hipMemcpyHostToDevice,
								   g.stream));
		hipLaunchKernelGGL(k_hnsw_init_nodes, dim3(n), dim3(256), 0, g.stream, h->d_vecs, h->d_levels, h->d_ncount,
						   h->d_nbrs, h->d_tids, d_rows, d_tids, (const int *) d_lv_in, n, h->dim, (int64_t) stride,
						   base);
		HIP_TRY(hipGetLastError());

		HnswDev		gd;

		gd.vecs = h->d_vecs; gd.levels = h->d_levels; gd.ncount = h->d_ncount; gd.nbr_off = nullptr;
		gd.nbrs = h->d_nbrs; gd.tids = h->d_tids; gd.dense_stride = (int64_t) stride; gd.nblocks = nb;
		gd.dim = h->dim; gd.m = h->m;
		uint32_t	round = 0;
		int64_t		nrounds = 0;
		const bool	trace = g_hnsw_trace != 0;
		/* the chunked commit keeps a list's slots in a 64-bit mask and a chunk's requests in LDS */
		/* commit kernel: 1 = hashed closed-form chunks (m <= 16), else / 3 = sorted-replay chunks (m <= 32),
		 * 2 = one wave, walk by walk */
		const bool	hash_commit = g_hnsw_spec == 1 && ksel <= NDB_HH_MAXSEL && 2 * h->m <= 32;
		const bool	par_commit = !hash_commit && (g_hnsw_spec == 1 || g_hnsw_spec == 3) && ksel <= NDB_HC_MAXSEL &&
			2 * h->m <= 64;
		const bool	fast = (h->dim % 4) == 0 && h->dim <= NDB_HNSW_FAST_MAX_DIM && !g_hnsw_nofast;
		uint32_t   *h_next = nullptr;

		HIP_TRY(hipHostMalloc((void **) &h_next, sizeof(uint32_t), hipHostMallocDefault));
		for (const Batch &b : batches)
		{
			const uint32_t nt = (uint32_t) (b.t1 - b.t0);
			int			burst = 2;	/* rounds queued between looks at `next` */

			gd.entry_point = b.entry;
			gd.entry_level = b.entry_level;
			HIP_TRY(hipMemsetAsync(R.next, 0, (1 + (size_t) nt) * sizeof(uint32_t), g.stream));	/* next, spec_round[] */
			for (;;)
			{
				for (int r = 0; r < burst; r++)
				{
					round++;
					nrounds++;
					if (fast)
						hipLaunchKernelGGL(k_hnsw_spec<true>, dim3(nt), dim3(256), smem, g.stream, gd, d_rows,
										   (const HnswTask *) (d_tasks + b.t0), (uint32_t) ef_construction, ksel,
										   R, round, base);
					else
						hipLaunchKernelGGL(k_hnsw_spec<false>, dim3(nt), dim3(64), smem, g.stream, gd, d_rows,
										   (const HnswTask *) (d_tasks + b.t0), (uint32_t) ef_construction, ksel,
										   R, round, base);
					if (hash_commit)
						hipLaunchKernelGGL(k_hnsw_commit_hash, dim3(1), dim3(256), 0, g.stream, h->d_ncount, h->d_nbrs,
										   (const HnswTask *) (d_tasks + b.t0), nt, h->m, ksel, R, round);
					else if (par_commit)
						hipLaunchKernelGGL(k_hnsw_commit_par, dim3(1), dim3(256), 0, g.stream, h->d_ncount, h->d_nbrs,
										   (const HnswTask *) (d_tasks + b.t0), nt, h->m, ksel, R, round);
					else
						hipLaunchKernelGGL(k_hnsw_commit, dim3(1), dim3(64), 0, g.stream, h->d_ncount, h->d_nbrs,
										   (const HnswTask *) (d_tasks + b.t0), nt, h->m, ksel, R, round);
				}
				HIP_TRY(hipMemcpyAsync(h_next, R.next, sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
				HIP_TRY(hipStreamSynchronize(g.stream));
				if (*h_next >= nt)
					break;
				burst = std::min(burst * 2, 16);
			}
			if (trace)
				fprintf(stderr, "hnsw batch: first row %u walks %u rounds so far %lld\n", tasks[b.t0].row, nt,
						(long long) nrounds);
		}
		HIP_TRY(hipGetLastError());
		unsigned long long st[4] = {0, 0, 0, 0};

		HIP_TRY(hipMemcpyAsync(st, d_stats, sizeof(st), hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
		h->build_stats[0] = (int64_t) tasks.size();
		h->build_stats[1] = (int64_t) st[0] - (int64_t) tasks.size();	/* walks run again */
		h->build_stats[2] = (int64_t) st[2];
		h->build_stats[3] = nrounds;
		h->build_stats[4] = (int64_t) batches.size();
		h->build_stats[5] = (int64_t) maxb;
		HIP_TRY(hipHostFree(h_next));
		HIP_TRY(hipFree(d_tasks));
		HIP_TRY(hipFree(d_u32));
		HIP_TRY(hipFree(d_stats));
	}
	HIP_TRY(hipFree(d_lv_in));
	HIP_TRY(hipFree(d_entry));
	h->ef_construction = ef_construction;
	h->nblocks = nb;
	h->entry_point = entry[0];
	h->entry_level = (int) entry[1];
	h->loaded = true;
	h->dense = true;
	return NDBHIP_OK;
}

/* hnswbuild on rows already in HBM: node i+1 = row i, levels[i] = its drawn level (host array). */
extern "C" int
ndbhip_hnsw_build_device(ndbhip_hnsw *h, const float *d_rows, const uint64_t *d_tids, uint32_t n,
						 const int32_t *levels, int ef_construction)
{
	HNSW_NOT_FROZEN(h, "ndbhip_hnsw_build_device");
	return hnsw_insert_rows(h, d_rows, d_tids, n, levels, ef_construction, 0);
}

/* hnswinsert (src/index/hnsw_am.c:478-538): n more rows on top of the graph the mirror holds */
extern "C" int
ndbhip_hnsw_insert_device(ndbhip_hnsw *h, const float *d_rows, const uint64_t *d_tids, uint32_t n,
						  const int32_t *levels, int ef_construction)
{
	HNSW_NOT_FROZEN(h, "ndbhip_hnsw_insert_device");
	if (!h)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!h->loaded || h->nblocks < 1)
		return hnsw_insert_rows(h, d_rows, d_tids, n, levels, ef_construction, 0);
	return hnsw_insert_rows(h, d_rows, d_tids, n, levels, ef_construction, h->nblocks - 1);
}

/* the same for host rows: staged to the device, then ndbhip_hnsw_insert_device */
extern "C" int
ndbhip_hnsw_insert(ndbhip_hnsw *h, const float *rows, const uint8_t *tids6, uint32_t n, const int32_t *levels,
				   int ef_construction)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	HNSW_NOT_FROZEN(h, "ndbhip_hnsw_insert");
	if (!h || !rows || !tids6 || !levels || n < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	float	   *d_rows = nullptr;
	uint64_t   *d_tids = nullptr;
	std::vector<uint64_t> t64(n);

	for (uint32_t i = 0; i < n; i++)
		t64[i] = ndb_tid_pack(tids6 + (size_t) i * 6);
	HIP_TRY(hipMalloc((void **) &d_rows, (size_t) n * h->dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_tids, (size_t) n * sizeof(uint64_t)));
	HIP_TRY(hipMemcpyAsync(d_rows, rows, (size_t) n * h->dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_tids, t64.data(), (size_t) n * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
	const int	rc = ndbhip_hnsw_insert_device(h, d_rows, d_tids, n, levels, ef_construction);

	(void) hipStreamSynchronize(g.stream);
	(void) hipFree(d_rows);
	(void) hipFree(d_tids);
	return rc;
}

extern "C" int
ndbhip_hnsw_set_search_mode(int mode)
{
	if (mode < 0 || mode > 2)
		return fail(NDBHIP_ERR_INVALID, "search mode must be 0 (auto), 1 (one wave per query) or 2 (block-cooperative)");
	g_hnsw_search_mode = mode;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_set_build_mode(int optimistic, int batch_div, int batch_max)
{
	if (batch_div < 1 || batch_max < 1 || batch_max > 65535)
		return fail(NDBHIP_ERR_INVALID, "batch_div >= 1 and 1 <= batch_max <= 65535 required");
	g_hnsw_spec = optimistic < 0 ? 0 : (optimistic > 3 ? 3 : optimistic);
	g_hnsw_batch_div = batch_div;
	g_hnsw_batch_max = batch_max;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_build_stats(const ndbhip_hnsw *h, int64_t out[6])
{
	if (!h || !out)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	memcpy(out, h->build_stats, sizeof(h->build_stats));
	return NDBHIP_OK;
}

/* ------------------------------------------------------------------ */
/* hnswbulkdelete on the mirror (src/index/hnsw_am.c:544-720)           */
/* ------------------------------------------------------------------ */

/* packed (loaded) neighbour slots -> the dense 16-level layout the writers use */
__global__ __launch_bounds__(256) void
k_hnsw_densify(const int *__restrict__ levels, const int64_t *__restrict__ nbr_off,
			   const uint32_t *__restrict__ packed, uint32_t nblocks, int m2, uint32_t *__restrict__ dense)
{
	const uint32_t b = blockIdx.x;
	const int64_t stride = (int64_t) NDBHIP_HNSW_MAX_LEVEL * m2;

	if (b >= nblocks)
		return;
	int			lv = levels[b];

	lv = lv < 0 ? -1 : (lv >= NDBHIP_HNSW_MAX_LEVEL ? NDBHIP_HNSW_MAX_LEVEL - 1 : lv);
	const int64_t have = b == 0 ? 0 : (int64_t) (lv + 1) * m2;

	for (int64_t j = threadIdx.x; j < stride; j += 256)
		dense[(size_t) b * stride + j] = j < have ? packed[nbr_off[b] + j] : NDBHIP_INVALID_BLOCK;
}

/* hit[b] = node b is live, has a sane level and its heapPtr is in the sorted set */
__global__ __launch_bounds__(256) void
k_hnsw_delete_mark(const uint64_t *__restrict__ tids, const int *__restrict__ levels,
				   const uint8_t *__restrict__ dead, uint32_t nblocks, const uint64_t *__restrict__ set,
				   int64_t nset, uint8_t *__restrict__ hit)
{
	const uint32_t b = blockIdx.x * 256 + threadIdx.x;

	if (b >= nblocks)
		return;
	bool		h = false;

	if (b != 0 && !dead[b] && levels[b] >= 0 && levels[b] < NDBHIP_HNSW_MAX_LEVEL)
	{
		const uint64_t t = tids[b];
		int64_t		lo = 0, hi = nset;

		while (lo < hi)
		{
			const int64_t mid = (lo + hi) >> 1;

			if (set[mid] < t)
				lo = mid + 1;
			else
				hi = mid;
		}
		h = lo < nset && set[lo] == t;
	}
	hit[b] = h ? 1 : 0;
}

/* ONE wave unlinks the hit nodes in block order, statement for statement (:618-699) */
__global__ __launch_bounds__(64) void
k_hnsw_delete_seq(const int *__restrict__ levels, int16_t *ncount, uint32_t *nbrs, uint8_t *dead,
				  const uint32_t *__restrict__ victims, uint32_t nvict, uint32_t nblocks, int m,
				  uint32_t *entry_io)
{
	const uint32_t lane = threadIdx.x;
	const int	m2 = 2 * m;
	const int64_t stride = (int64_t) NDBHIP_HNSW_MAX_LEVEL * m2;
	uint32_t	entry = entry_io[0];
	int			entry_level = (int) entry_io[1];

	for (uint32_t v = 0; v < nvict; v++)
	{
		const uint32_t blk = victims[v];
		const int	nodeLevel = levels[blk];

		for (int level = 0; level <= nodeLevel; level++)
		{
			const int	nc = hnsw_clamp(gload<true>(&ncount[(size_t) blk * NDBHIP_HNSW_MAX_LEVEL + level]), m);
			const uint32_t *mine = nbrs + (size_t) blk * stride + (size_t) level * m2;

			for (int i = 0; i < nc; i++)
			{
				const uint32_t nb = gload<true>(&mine[i]);

				/* :630-638, then hnswRemoveNodeFromNeighbor (:2747-2840) */
				if (nb == NDBHIP_INVALID_BLOCK || nb >= nblocks || nb == 0)
					continue;
				int16_t    *ncp = &ncount[(size_t) nb * NDBHIP_HNSW_MAX_LEVEL + level];
				uint32_t   *nn = nbrs + (size_t) nb * stride + (size_t) level * m2;
				const int16_t raw = gload<true>(ncp);
				const int	cnt = hnsw_clamp(raw, m);
				const uint32_t val = (int) lane < cnt ? gload<true>(&nn[lane]) : NDBHIP_INVALID_BLOCK;
				const unsigned long long match = __ballot((int) lane < cnt && val == blk);

				if (match)
				{
					const int	idx = __ffsll((long long) match) - 1;
					const uint32_t next = __shfl_down(val, 1, 64);

					if ((int) lane >= idx && (int) lane < cnt - 1)
						gstore(&nn[lane], next);
					if ((int) lane == cnt - 1)
						gstore(&nn[lane], (uint32_t) NDBHIP_INVALID_BLOCK);
					if (lane == 0)
						gstore(ncp, (int16_t) (raw - 1));
					hnsw_publish();
				}
			}
		}
		if (entry == blk)	/* :642-690 */
		{
			bool		found = false;

			for (int level = nodeLevel; level >= 0 && !found; level--)
			{
				const int	nc = hnsw_clamp(gload<true>(&ncount[(size_t) blk * NDBHIP_HNSW_MAX_LEVEL + level]), m);
				const uint32_t *mine = nbrs + (size_t) blk * stride + (size_t) level * m2;

				for (int i = 0; i < nc && !found; i++)
				{
					const uint32_t nb = gload<true>(&mine[i]);

					if (hnsw_valid(nblocks, nb) && levels[nb] >= 0 && levels[nb] < NDBHIP_HNSW_MAX_LEVEL)
					{
						entry = nb;
						entry_level = levels[nb];
						found = true;
					}
				}
			}
			if (!found)
			{
				entry = NDBHIP_INVALID_BLOCK;
				entry_level = -1;
			}
		}
		if (lane == 0)
			dead[blk] = 1;
	}
	if (lane == 0)
	{
		entry_io[0] = entry;
		entry_io[1] = (uint32_t) entry_level;
	}
}

/* loaded graphs hold (level+1)*2m slots per node; the reference's writers put entries at `level` into
 * whatever node a list names (Q12/Q21), so before the mirror is modified every node gets all 16 levels */
static int
hnsw_densify(ndbhip_hnsw *h)
{
	if (h->dense)
		return 0;
	if (hnsw_frozen(h))
		return fail(NDBHIP_ERR_STATE, "the graph is shared (ndbhip_hnsw_share) and its neighbour lists are not in the dense layout");
	const uint32_t nb = h->nblocks;
	const int	m2 = 2 * h->m;
	const size_t stride = (size_t) NDBHIP_HNSW_MAX_LEVEL * m2;
	uint32_t   *d_dense = nullptr;

	HIP_TRY(hipMalloc((void **) &d_dense, (size_t) nb * stride * sizeof(uint32_t)));
	hipLaunchKernelGGL(k_hnsw_densify, dim3(nb), dim3(256), 0, g.stream, (const int *) h->d_levels,
					   (const int64_t *) h->d_nbr_off, (const uint32_t *) h->d_nbrs, nb, m2, d_dense);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipFree(h->d_nbrs));
	HIP_TRY(hipFree(h->d_nbr_off));
	h->d_nbrs = d_dense;
	h->d_nbr_off = nullptr;
	h->dense = true;
	return 0;
}

extern "C" int
ndbhip_hnsw_delete(ndbhip_hnsw *h, const uint8_t *tids6, int64_t n, int64_t *removed)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	HNSW_NOT_FROZEN(h, "ndbhip_hnsw_delete");
	if (!h || n < 0 || (n > 0 && !tids6))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!h->loaded)
		return fail(NDBHIP_ERR_STATE, "hnsw mirror not loaded");
	if (2 * h->m > 64)
		return fail(NDBHIP_ERR_UNSUPPORTED, "bulkdelete on the mirror supports m <= 32");
	if (removed)
		*removed = 0;
	if (n == 0 || h->nblocks < 2)
		return NDBHIP_OK;
	const uint32_t nb = h->nblocks;

	{
		int			rc = hnsw_densify(h);

		if (rc)
			return rc;
	}
	if (!h->d_dead)
	{
		/* (as long as the other arrays: later inserts fill blocks below cap_blocks without reallocating) */
		const size_t dcap = std::max<size_t>(nb, h->cap_blocks);

		HIP_TRY(hipMalloc((void **) &h->d_dead, dcap));
		HIP_TRY(hipMemsetAsync(h->d_dead, 0, dcap, g.stream));
	}
	std::vector<uint64_t> set((size_t) n);

	for (int64_t i = 0; i < n; i++)
		set[(size_t) i] = ndb_tid_pack(tids6 + 6 * i);
	std::sort(set.begin(), set.end());
	uint64_t   *d_set = nullptr;
	uint8_t    *d_hit = nullptr;
	uint32_t   *d_vict = nullptr, *d_entry = nullptr;
	std::vector<uint8_t> hit((size_t) nb);

	HIP_TRY(hipMalloc((void **) &d_set, (size_t) n * sizeof(uint64_t)));
	HIP_TRY(hipMalloc((void **) &d_hit, (size_t) nb));
	HIP_TRY(hipMemcpyAsync(d_set, set.data(), (size_t) n * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
	hipLaunchKernelGGL(k_hnsw_delete_mark, dim3((nb + 255) / 256), dim3(256), 0, g.stream,
					   (const uint64_t *) h->d_tids, (const int *) h->d_levels, (const uint8_t *) h->d_dead, nb,
					   (const uint64_t *) d_set, n, d_hit);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(hit.data(), d_hit, (size_t) nb, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	std::vector<uint32_t> victims;

	for (uint32_t b = 1; b < nb; b++)	/* ascending block order: :586 */
		if (hit[b])
			victims.push_back(b);
	if (!victims.empty())
	{
		uint32_t	entry[2] = {h->entry_point, (uint32_t) h->entry_level};

		HIP_TRY(hipMalloc((void **) &d_vict, victims.size() * sizeof(uint32_t)));
		HIP_TRY(hipMalloc((void **) &d_entry, sizeof(entry)));
		HIP_TRY(hipMemcpyAsync(d_vict, victims.data(), victims.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
							   g.stream));
		HIP_TRY(hipMemcpyAsync(d_entry, entry, sizeof(entry), hipMemcpyHostToDevice, g.stream));
		hipLaunchKernelGGL(k_hnsw_delete_seq, dim3(1), dim3(64), 0, g.stream, (const int *) h->d_levels,
						   h->d_ncount, h->d_nbrs, h->d_dead, (const uint32_t *) d_vict, (uint32_t) victims.size(),
						   nb, h->m, d_entry);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(entry, d_entry, sizeof(entry), hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
		h->entry_point = entry[0];
		h->entry_level = (int) entry[1];
		HIP_TRY(hipFree(d_vict));
		HIP_TRY(hipFree(d_entry));
	}
	if (removed)
		*removed = (int64_t) victims.size();
	HIP_TRY(hipFree(d_set));
	HIP_TRY(hipFree(d_hit));
	return NDBHIP_OK;
}

/* the two search-width fields of the meta page the AM callbacks read (hnsw_am.c:923-936, 2369-2378) */
extern "C" int
ndbhip_hnsw_get_meta(const ndbhip_hnsw *h, int *ef_construction, int *ef_search)
{
	if (!h)
		return fail(NDBHIP_ERR_INVALID, "graph is NULL");
	if (ef_construction) *ef_construction = h->ef_construction;
	if (ef_search) *ef_search = h->ef_search;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_set_meta(ndbhip_hnsw *h, int ef_construction, int ef_search)
{
	if (!h)
		return fail(NDBHIP_ERR_INVALID, "graph is NULL");
	if (ef_construction < 4 || ef_construction > NDBHIP_MAX_EF || ef_search < 4 || ef_search > NDBHIP_MAX_EF)
		return fail(NDBHIP_ERR_INVALID, "ef_construction / ef_search out of range 4..%d", NDBHIP_MAX_EF);
	h->ef_construction = ef_construction;
	h->ef_search = ef_search;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_shape(const ndbhip_hnsw *h, int *dim, int *m)
{
	if (!h)
		return fail(NDBHIP_ERR_INVALID, "graph is NULL");
	if (dim) *dim = h->dim;
	if (m) *m = h->m;
	return NDBHIP_OK;
}

/* vectors [nblocks * dim], heapPtrs [nblocks * 6], dead flags [nblocks] (any may be NULL) */
extern "C" int
ndbhip_hnsw_export_rows(const ndbhip_hnsw *h, float *vecs, uint8_t *tids6, uint8_t *dead)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!h || !h->loaded)
		return fail(NDBHIP_ERR_STATE, "hnsw mirror not loaded");
	const uint32_t nb = h->nblocks;

	HIP_TRY(hipStreamSynchronize(g.stream));
	if (vecs)
		HIP_TRY(hipMemcpy(vecs, h->d_vecs, (size_t) nb * h->dim * sizeof(float), hipMemcpyDeviceToHost));
	if (tids6)
	{
		std::vector<uint64_t> t64(nb);

		HIP_TRY(hipMemcpy(t64.data(), h->d_tids, (size_t) nb * sizeof(uint64_t), hipMemcpyDeviceToHost));
		for (uint32_t b = 0; b < nb; b++)
			ndb_tid_unpack(t64[b], tids6 + (size_t) b * 6);
	}
	if (dead)
	{
		if (h->d_dead)
			HIP_TRY(hipMemcpy(dead, h->d_dead, (size_t) nb, hipMemcpyDeviceToHost));
		else
			memset(dead, 0, (size_t) nb);
	}
	return NDBHIP_OK;
}

/* line pointers hnswbulkdelete had marked dead before the mirror was packed ([nblocks]) */
extern "C" int
ndbhip_hnsw_set_dead_flags(ndbhip_hnsw *h, const uint8_t *dead)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	HNSW_NOT_FROZEN(h, "ndbhip_hnsw_set_dead_flags");
	if (!h || !h->loaded || !dead)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!h->d_dead)
	{
		const size_t dcap = std::max<size_t>(h->nblocks, h->cap_blocks);

		HIP_TRY(hipMalloc((void **) &h->d_dead, dcap));
		HIP_TRY(hipMemsetAsync(h->d_dead, 0, dcap, g.stream));
	}
	HIP_TRY(hipMemcpyAsync(h->d_dead, dead, (size_t) h->nblocks, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	return NDBHIP_OK;
}

/* Read a graph back in the dense layout: levels [nblocks], ncount [nblocks*16],
 * nbrs [nblocks*16*2m] (slots a packed graph does not hold come back as 0xFFFFFFFF). */
extern "C" int
ndbhip_hnsw_export(const ndbhip_hnsw *h, uint32_t *nblocks, int32_t *levels, int16_t *ncount, uint32_t *nbrs,
				   uint32_t *entry_point, int *entry_level)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!h || !h->loaded)
		return fail(NDBHIP_ERR_STATE, "hnsw mirror not loaded");
	const uint32_t nb = h->nblocks;
	const size_t stride = (size_t) NDBHIP_HNSW_MAX_LEVEL * 2 * h->m;

	if (nblocks) *nblocks = nb;
	if (entry_point) *entry_point = h->entry_point;
	if (entry_level) *entry_level = h->entry_level;
	HIP_TRY(hipStreamSynchronize(g.stream));
	std::vector<int32_t> lv(nb);

	HIP_TRY(hipMemcpy(lv.data(), h->d_levels, (size_t) nb * sizeof(int), hipMemcpyDeviceToHost));
	if (levels)
		memcpy(levels, lv.data(), (size_t) nb * sizeof(int));
	if (ncount)
		HIP_TRY(hipMemcpy(ncount, h->d_ncount, (size_t) nb * 16 * sizeof(int16_t), hipMemcpyDeviceToHost));
	if (nbrs)
	{
		if (h->dense)
			HIP_TRY(hipMemcpy(nbrs, h->d_nbrs, (size_t) nb * stride * sizeof(uint32_t), hipMemcpyDeviceToHost));
		else
		{
			std::vector<int64_t> off((size_t) nb + 1);

			HIP_TRY(hipMemcpy(off.data(), h->d_nbr_off, off.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
			std::vector<uint32_t> packed((size_t) std::max<int64_t>(off[nb], 1));

			if (off[nb] > 0)
				HIP_TRY(hipMemcpy(packed.data(), h->d_nbrs, (size_t) off[nb] * sizeof(uint32_t), hipMemcpyDeviceToHost));
			memset(nbrs, 0xFF, (size_t) nb * stride * sizeof(uint32_t));
			for (uint32_t b = 1; b < nb; b++)
				memcpy(nbrs + (size_t) b * stride, packed.data() + off[b], (size_t) (off[b + 1] - off[b]) * sizeof(uint32_t));
		}
	}
	return NDBHIP_OK;
}

static int
hnsw_check(ndbhip_hnsw *h, int nq, int strategy, int ef, int k)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!h || !h->loaded)
		return fail(NDBHIP_ERR_STATE, "hnsw mirror not loaded");
	if (strategy < 1 || strategy > 3)	/* hnsw_am.c:1339-1343 */
		return fail(NDBHIP_ERR_UNSUPPORTED, "hnsw: unsupported distance strategy %d", strategy);
	if (nq < 0 || ef < 1 || ef > NDBHIP_MAX_EF || k < 1 || k > NDBHIP_MAX_K)
		return fail(NDBHIP_ERR_INVALID, "nq/ef/k out of range (ef <= %d, k <= %d)", NDBHIP_MAX_EF, NDBHIP_MAX_K);
	return 0;
}

extern "C" int
ndbhip_hnsw_search_device(ndbhip_hnsw *h, const float *d_queries, int nq, int strategy, int ef, int k,
						  uint32_t *d_out_blocks, float *d_out_dist, int *d_out_count, uint64_t *d_out_tids,
						  int64_t *d_out_scored)
{
	int			rc = hnsw_check(h, nq, strategy, ef, k);

	if (rc)
		return rc;
	if (nq == 0)
		return NDBHIP_OK;
	if (!d_queries || !d_out_blocks || !d_out_dist || !d_out_count)
		return fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	HnswDev		d;

	d.vecs = h->d_vecs; d.levels = h->d_levels; d.ncount = h->d_ncount; d.nbr_off = h->d_nbr_off;
	d.nbrs = h->d_nbrs; d.tids = h->d_tids; d.nblocks = h->nblocks; d.dim = h->dim; d.m = h->m;
	d.dense_stride = h->dense ? (int64_t) NDBHIP_HNSW_MAX_LEVEL * 2 * h->m : 0;
	d.entry_point = h->entry_point; d.entry_level = h->entry_level;
	const int	nacc = strategy == 1 ? FastAcc<R_HNSW_L2>::N : (strategy == 2 ? FastAcc<R_HNSW_COS>::N : FastAcc<R_HNSW_IP>::N);
	const size_t smem_fast = hnsw_smem_bytes((uint32_t) ef, (uint32_t) k, (uint32_t) h->m,
											 hnsw_fast_bytes(nacc, NDB_HNSW_SEARCH_KMAX, h->dim));
	/* g_hnsw_search_mode: 0 auto, 1 one wave per query (the literal per-lane recipe), 2 block-cooperative */
	const bool	fast = (h->dim % 4) == 0 && smem_fast <= NDB_TOPK_MAX_SMEM && g_hnsw_search_mode != 1;
	const size_t smem = fast ? smem_fast : hnsw_smem_bytes((uint32_t) ef, (uint32_t) k, (uint32_t) h->m);

	if (g_hnsw_search_mode == 2 && !fast)
		return fail(NDBHIP_ERR_UNSUPPORTED, "the block-cooperative search needs dim %% 4 == 0 and an LDS-resident state");
	if (smem > NDB_TOPK_MAX_SMEM)
		return fail(NDBHIP_ERR_UNSUPPORTED, "ef/k too large for the LDS-resident candidate set");
	ScanTimer	t;

	if (t.start()) return NDBHIP_ERR_HIP;
#define LAUNCH_HNSW_SEARCH(RR)                                                                                       \
	do {                                                                                                             \
		if (fast)                                                                                                    \
			hipLaunchKernelGGL(k_hnsw_search_fast<RR>, dim3(nq), dim3(256), smem, g.stream, d, d_queries,            \
							   (uint32_t) ef, (uint32_t) k, d_out_blocks, d_out_dist, d_out_count, d_out_tids,         \
							   (long long *) d_out_scored);                                                          \
		else                                                                                                         \
			hipLaunchKernelGGL(k_hnsw_search<RR>, dim3(nq), dim3(64), smem, g.stream, d, d_queries,                  \
							   (uint32_t) ef, (uint32_t) k, d_out_blocks, d_out_dist, d_out_count, d_out_tids,         \
							   (long long *) d_out_scored);                                                          \
	} while (0)
	switch (strategy)
	{
		case 1: LAUNCH_HNSW_SEARCH(R_HNSW_L2); break;
		case 2: LAUNCH_HNSW_SEARCH(R_HNSW_COS); break;
		default: LAUNCH_HNSW_SEARCH(R_HNSW_IP); break;
	}
#undef LAUNCH_HNSW_SEARCH
	if (t.stop()) return NDBHIP_ERR_HIP;
	HIP_TRY(hipGetLastError());
	NDB_STAT_ADD(queries, (uint64_t) nq);
	return NDBHIP_OK;
}

/* hnsw_search_layer (src/scan/hnsw_scan.c:379-477) for nq queries: see k_hnsw_scan_layer */
extern "C" int
ndbhip_hnsw_search_layer_device(ndbhip_hnsw *h, const float *d_queries, int nq, int strategy, int ef, int k,
								uint32_t *d_out_blocks, float *d_out_dist, int *d_out_count,
								uint64_t *d_out_tids, int64_t *d_out_scored)
{
	/* `strategy` is an argument of the reference's function that its body never reads (:384): every
	 * distance is compute_l2_distance */
	int			rc = hnsw_check(h, nq, 1, ef, k);

	(void) strategy;
	if (rc)
		return rc;
	if (nq == 0)
		return NDBHIP_OK;
	if (!d_queries || !d_out_blocks || !d_out_dist || !d_out_count)
		return fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	if (!h->dense)				/* layer reads are not guarded by the node's own level (:549): dense slots */
	{
		rc = hnsw_densify(h);
		if (rc)
			return rc;
	}
	const size_t smem = (size_t) NDB_TILE_FLOATS * 4 + ((size_t) 2 * ef + (size_t) k) * 8;

	if (smem > NDB_TOPK_MAX_SMEM)
		return fail(NDBHIP_ERR_UNSUPPORTED, "ef/k too large for the LDS-resident candidate heap");
	static bool attr_set = false;

	if (!attr_set)
	{
		HIP_TRY(hipFuncSetAttribute((const void *) k_hnsw_scan_layer, hipFuncAttributeMaxDynamicSharedMemorySize,
									NDB_TOPK_MAX_SMEM));
		attr_set = true;
	}
	/* persistent single-wave blocks, each with its own visited bitmap (1 bit per block of the relation) */
	const uint32_t vwords = (h->nblocks + 31u) / 32u;
	uint32_t	grid = (uint32_t) std::min<int64_t>(nq, (int64_t) g.num_cus * 8);
	const size_t max_bitmap_bytes = (size_t) 1 << 30;

	while (grid > 1 && (size_t) grid * vwords * 4 > max_bitmap_bytes)
		grid /= 2;
	const size_t want = (size_t) grid * vwords;

	if (want > h->w_vbits_n)
	{
		if (grow(h->w_vbits, h->w_vbits_n, want)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemsetAsync(h->w_vbits, 0, want * 4, g.stream));	/* every query leaves its map zero */
	}
	if (grow(h->w_vlog, h->w_vlog_n, (size_t) grid * NDB_SCAN_VLOG)) return NDBHIP_ERR_HIP;
	HnswDev		d;

	d.vecs = h->d_vecs; d.levels = h->d_levels; d.ncount = h->d_ncount; d.nbr_off = h->d_nbr_off;
	d.nbrs = h->d_nbrs; d.tids = h->d_tids; d.nblocks = h->nblocks; d.dim = h->dim; d.m = h->m;
	d.dense_stride = (int64_t) NDBHIP_HNSW_MAX_LEVEL * 2 * h->m;
	d.entry_point = h->entry_point; d.entry_level = h->entry_level;
	ScanTimer	t;

	if (t.start()) return NDBHIP_ERR_HIP;
	hipLaunchKernelGGL(k_hnsw_scan_layer, dim3(grid), dim3(64), smem, g.stream, d, d_queries, (uint32_t) nq,
					   (uint32_t) ef, (uint32_t) k, h->w_vbits, vwords, h->w_vlog, d_out_blocks, d_out_dist,
					   d_out_count, d_out_tids, (long long *) d_out_scored);
	if (t.stop()) return NDBHIP_ERR_HIP;
	HIP_TRY(hipGetLastError());
	NDB_STAT_ADD(queries, (uint64_t) nq);
	return NDBHIP_OK;
}

/* mode: 0 hnswSearch, 1 hnsw_search_layer, 2 / 3 the `intended` search on float4 / fp16 walk rows */
static int
hnsw_search_host(ndbhip_hnsw *h, int mode, const float *queries, int nq, int strategy, int ef, int k,
				 uint32_t *out_blocks, float *out_dist, int *out_count, uint8_t *out_tids6, int64_t *out_scored)
{
	const bool	scan_layer = mode == 1;
	int			rc = hnsw_check(h, nq, scan_layer ? 1 : strategy, ef, k);

	if (rc)
		return rc;
	if (nq == 0)
		return NDBHIP_OK;
	if (!queries || !out_blocks || !out_dist || !out_count)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	if (grow(h->w_q, h->w_q_n, (size_t) nq * h->dim)) return NDBHIP_ERR_HIP;
	/* one device block for the results — [TIDs | evaluation counts | blocks | distances | counts] — and one
	 * pinned host block for the queries and the results: one H2D, one clear, the walk, one D2H per call */
	const size_t nk = (size_t) nq * k;
	const size_t out_bytes = nk * 8 + (size_t) nq * 8 + nk * 4 + nk * 4 + (size_t) nq * 4;

	if (grow(h->w_ot, h->w_ot_n, (out_bytes + 7) / 8)) return NDBHIP_ERR_HIP;
	uint64_t   *d_tid = h->w_ot;
	long long  *d_sc = (long long *) (d_tid + nk);
	uint32_t   *d_blk = (uint32_t *) (d_sc + nq);
	float	   *d_dist = (float *) (d_blk + nk);
	int		   *d_cnt = (int *) (d_dist + nk);
	const size_t q_bytes = ((size_t) nq * h->dim * sizeof(float) + 7) & ~(size_t) 7;

	if (q_bytes + out_bytes > h->pin_n)
	{
		if (h->pin) HIP_TRY(hipHostFree(h->pin));
		h->pin = nullptr;
		h->pin_n = 0;
		HIP_TRY(hipHostMalloc((void **) &h->pin, q_bytes + out_bytes, hipHostMallocDefault));
		h->pin_n = q_bytes + out_bytes;
	}
	unsigned char *h_out = (unsigned char *) h->pin + q_bytes;

	memcpy(h->pin, queries, (size_t) nq * h->dim * sizeof(float));
	HIP_TRY(hipMemcpyAsync(h->w_q, h->pin, (size_t) nq * h->dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemsetAsync(d_tid, 0, out_bytes, g.stream));
	rc = mode == 3 ? ndbhip_hnsw_search_intended_w16_device(h, h->w_q, nq, strategy, ef, k, d_blk, d_dist, d_cnt, d_tid, (int64_t *) d_sc)
		: mode == 2 ? ndbhip_hnsw_search_intended_device(h, h->w_q, nq, strategy, ef, k, d_blk, d_dist, d_cnt, d_tid, (int64_t *) d_sc)
		: scan_layer
		? ndbhip_hnsw_search_layer_device(h, h->w_q, nq, strategy, ef, k, d_blk, d_dist, d_cnt, d_tid, (int64_t *) d_sc)
		: ndbhip_hnsw_search_device(h, h->w_q, nq, strategy, ef, k, d_blk, d_dist, d_cnt, d_tid, (int64_t *) d_sc);
	if (rc)
		return rc;
	HIP_TRY(hipMemcpyAsync(h_out, d_tid, out_bytes, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	const uint64_t *t64 = (const uint64_t *) h_out;
	const long long *sc = (const long long *) (h_out + nk * 8);

	memcpy(out_blocks, h_out + nk * 8 + (size_t) nq * 8, nk * 4);
	memcpy(out_dist, h_out + nk * 8 + (size_t) nq * 8 + nk * 4, nk * 4);
	memcpy(out_count, h_out + nk * 8 + (size_t) nq * 8 + nk * 8, (size_t) nq * 4);
	uint64_t	tot = 0;

	for (int q2 = 0; q2 < nq; q2++)
	{
		tot += (uint64_t) sc[q2];
		if (out_scored)
			out_scored[q2] = sc[q2];
		if (out_tids6)
			for (int i = 0; i < k; i++)
				ndb_tid_unpack(i < out_count[q2] ? t64[(size_t) q2 * k + i] : 0, out_tids6 + ((size_t) q2 * k + i) * 6);
	}
	g.host_rows += tot;
	g.host_bytes += tot * (uint64_t) h->dim * 4;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_search(ndbhip_hnsw *h, const float *queries, int nq, int strategy, int ef, int k,
				   uint32_t *out_blocks, float *out_dist, int *out_count, uint8_t *out_tids6, int64_t *out_scored)
{
	return hnsw_search_host(h, 0, queries, nq, strategy, ef, k, out_blocks, out_dist, out_count, out_tids6,
							out_scored);
}

extern "C" int
ndbhip_hnsw_search_layer(ndbhip_hnsw *h, const float *queries, int nq, int strategy, int ef, int k,
						 uint32_t *out_blocks, float *out_dist, int *out_count, uint8_t *out_tids6,
						 int64_t *out_scored)
{
	return hnsw_search_host(h, 1, queries, nq, strategy, ef, k, out_blocks, out_dist, out_count, out_tids6,
							out_scored);
}

/* the `intended` search from host pointers (what ndb_hnswgettuple calls when neurondb.ref_compat is off): one upload, the
 * walk, one download; heap TIDs come back with the blocks like ndbhip_hnsw_search's */
extern "C" int
ndbhip_hnsw_search_intended(ndbhip_hnsw *h, const float *queries, int nq, int strategy, int ef, int k, int walk16,
							uint32_t *out_blocks, float *out_dist, int *out_count, uint8_t *out_tids6, int64_t *out_evals)
{
	return hnsw_search_host(h, walk16 ? 3 : 2, queries, nq, strategy, ef, k, out_blocks, out_dist, out_count, out_tids6,
							out_evals);
}

/* ================================================================== */
/* the `intended` HNSW (ndbhip_hnsw2.h; oracle/ndb_oracle_hnsw2.c is its sequential definition)                   */
/* ================================================================== */
/* rocPRIM, AMD's own device-primitive library (header-only, wave64-tuned radix sort and scan: /opt/rocm/include/rocprim) —
 * called directly; round 5 went through hipCUB, the CUB-compatibility wrapper around the same kernels.  Used by the intended
 * build only (back-links grouped on the device), never on a search path. */
#include <rocprim/rocprim.hpp>
#include "ndbhip_hnsw2.h"

int			g_h2_host_groups = 0;	/* option hnsw_intended_host_groups: 1 = a build batch's back-links grouped on the host (rounds 3-4) */
int			g_h2_occ4 = 0;			/* option hnsw_intended_occ4 (measured: more walkers a SIMD buy nothing) */
static int	g_h2_select = 1;		/* 1: the heuristic (ndbhip_hnsw_set_intended_select(0): the nearest m) */

/* device temporaries of one call, freed on every way out */
struct H2Tmp
{
	std::vector<void *> owned;
	template <class T> int alloc(T *&p, size_t bytes)
	{
		p = nullptr;
		HIP_TRY(hipMalloc((void **) &p, bytes ? bytes : 16));
		owned.push_back((void *) p);
		return 0;
	}
	~H2Tmp()
	{
		for (auto o : owned)
			if (o)
				(void) hipFree(o);
	}
};

static int
h2_workspace(ndbhip_hnsw *h, uint32_t nwaves, uint32_t nblocks, uint32_t *nwords_out)
{
	const uint32_t nwords = (nblocks + 31u) / 32u + 1u;

	if (h->w_vbits_n < (size_t) nwaves * nwords)
	{
		if (grow(h->w_vbits, h->w_vbits_n, (size_t) nwaves * nwords)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemsetAsync(h->w_vbits, 0, h->w_vbits_n * sizeof(uint32_t), g.stream));	/* all zero at rest */
	}
	/* (+ 64 words behind the logs: the "next query" counter of a search launch) */
	if (grow(h->w_vlog, h->w_vlog_n, (size_t) nwaves * H2_LOG_CAP + 64)) return NDBHIP_ERR_HIP;
	*nwords_out = nwords;
	return 0;
}

static H2Graph
h2_graph(const ndbhip_hnsw *h, uint32_t nvisible)
{
	H2Graph		gr;

	gr.vecs = h->d_vecs;
	gr.vecs16 = h->d_vecs16;
	gr.rinv = nullptr;
	gr.levels = h->d_levels;
	gr.ncount = h->d_ncount;
	gr.nbrs = h->d_nbrs;
	gr.stride = (int64_t) NDBHIP_HNSW_MAX_LEVEL * 2 * h->m;
	gr.nvisible = nvisible;
	gr.dim = h->dim;
	gr.m = h->m;
	return gr;
}

/*
 * hnswbuild, `intended` mode: node i + 1 = row i, levels[i] its drawn level (host array), every comparison L2.
 * Batches of clamp(nodes so far / batch_div, 1, batch_max) inserts: the members search the graph as it stood when the
 * batch began (k_h2_insert_search), the host groups their back-links by target, every target replays its requests in
 * insertion order (k_h2_apply).  The graph is the one ndbo_h2_build leaves, slot for slot.
 */
/* base = nodes the mirror holds already (0: hnswbuild from nothing; > 0: round 6, rows appended — hnswinsert under `intended`:
 * row i becomes block base + 1 + i and the schedule goes on from the relation's size: oracle ndbo_h2_build on a graph that is
 * not empty) */
static int
h2_build_rows(ndbhip_hnsw *h, const float *d_rows, const uint64_t *d_tids, uint32_t n,
			  const int32_t *levels, int ef_construction, int batch_div, int batch_max, uint32_t base)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	HNSW_NOT_FROZEN(h, "ndbhip_hnsw_build_intended_device");
	if (!h || !d_rows || !d_tids || !levels || n < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (ef_construction < 4 || ef_construction > NDBHIP_MAX_EF)
		return fail(NDBHIP_ERR_INVALID, "ef_construction %d out of range 4..%d", ef_construction, NDBHIP_MAX_EF);
	if (h->m > 64)
		return fail(NDBHIP_ERR_UNSUPPORTED, "intended build: m <= 64");
	if ((uint64_t) base + n + 1 > 0xFFFFFFF0ull)
		return fail(NDBHIP_ERR_UNSUPPORTED, "more than 2^32 blocks");
	if (batch_div < 1) batch_div = 1;
	if (batch_max < 1) batch_max = 1;
	const uint32_t nb = base + n + 1;
	const int	m = 2 * h->m;		/* row width of the selections: a level-0 row may hold 2m */
	const size_t stride = (size_t) NDBHIP_HNSW_MAX_LEVEL * 2 * h->m;
	std::vector<int> lev(n);

	for (uint32_t i = 0; i < n; i++)
		lev[i] = levels[i] < 0 ? 0 : (levels[i] > NDBHIP_HNSW_MAX_LEVEL - 1 ? NDBHIP_HNSW_MAX_LEVEL - 1 : levels[i]);
	if (base == 0)
	{
		hnsw_free_dev(h);
		HIP_TRY(hipMalloc((void **) &h->d_vecs, (size_t) nb * h->dim * sizeof(float)));
		HIP_TRY(hipMalloc((void **) &h->d_levels, (size_t) nb * sizeof(int)));
		HIP_TRY(hipMalloc((void **) &h->d_ncount, (size_t) nb * 16 * sizeof(int16_t)));
		HIP_TRY(hipMalloc((void **) &h->d_nbrs, (size_t) nb * stride * sizeof(uint32_t)));
		HIP_TRY(hipMalloc((void **) &h->d_tids, (size_t) nb * sizeof(uint64_t)));
		h->cap_blocks = nb;
		HIP_TRY(hipMemsetAsync(h->d_vecs, 0, (size_t) h->dim * sizeof(float), g.stream));
		HIP_TRY(hipMemsetAsync(h->d_levels, 0, sizeof(int), g.stream));
		HIP_TRY(hipMemsetAsync(h->d_ncount, 0, 16 * sizeof(int16_t), g.stream));
		HIP_TRY(hipMemsetAsync(h->d_nbrs, 0xFF, stride * sizeof(uint32_t), g.stream));
		HIP_TRY(hipMemsetAsync(h->d_tids, 0, sizeof(uint64_t), g.stream));
	}
	else
	{
		int			rc = hnsw_grow_dense(h, base, nb);

		if (rc)
			return rc;
	}
	/* every page is laid out before anything is linked: a node is unreachable until its own insert links it */
	const size_t b1 = (size_t) base + 1;

	HIP_TRY(hipMemcpyAsync(h->d_vecs + b1 * h->dim, d_rows, (size_t) n * h->dim * sizeof(float), hipMemcpyDeviceToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(h->d_levels + b1, lev.data(), (size_t) n * sizeof(int), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemsetAsync(h->d_ncount + b1 * 16, 0, (size_t) n * 16 * sizeof(int16_t), g.stream));
	HIP_TRY(hipMemsetAsync(h->d_nbrs + b1 * stride, 0xFF, (size_t) n * stride * sizeof(uint32_t), g.stream));
	HIP_TRY(hipMemcpyAsync(h->d_tids + b1, d_tids, (size_t) n * sizeof(uint64_t), hipMemcpyDeviceToDevice, g.stream));
	if (h->d_dead)
		HIP_TRY(hipMemsetAsync(h->d_dead + b1, 0, (size_t) n, g.stream));
	h->dense = true;
	h->nblocks = nb;
	h->ef_construction = ef_construction;

	const uint32_t nwaves = (uint32_t) std::min<int64_t>((int64_t) g.num_cus * g_h2_waves, std::max(1, batch_max));
	uint32_t	nwords = 0;

	if (h2_workspace(h, nwaves, nb, &nwords)) return NDBHIP_ERR_HIP;
	const size_t smem = h2_smem_bytes((uint32_t) ef_construction, false);

	HIP_TRY(hipFuncSetAttribute((const void *) k_h2_insert_search, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem));
	/* per-batch scratch: selections (compact: a member has min(level, entry level) + 1 levels), groups, requests */
	H2Tmp		tmp;
	uint32_t   *d_off = nullptr, *d_sid = nullptr;
	double	   *d_sd2 = nullptr;
	int		   *d_sn = nullptr;
	H2Group    *d_grp = nullptr;
	H2Req	   *d_req = nullptr;
	const size_t bm = (size_t) batch_max, lvcap = bm * NDBHIP_HNSW_MAX_LEVEL;

	uint32_t   *d_next = nullptr;	/* the batch's next unassigned member (k_h2_insert_search deals them to whichever wave is free) */

	if (tmp.alloc(d_next, 16)) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_off, bm * 4)) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_sid, lvcap * m * 4)) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_sd2, lvcap * m * 8)) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_sn, lvcap * 4)) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_grp, lvcap * m * sizeof(H2Group))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_req, lvcap * m * sizeof(H2Req))) return NDBHIP_ERR_HIP;
	/* back-links grouped on the device (ndbhip_hnsw2.h k_h2_bl_*): every member's first selection row for the whole build
	 * (the schedule and the entry level at every batch follow from the level draws alone), sort and scan scratch */
	const bool	devgrp = g_h2_host_groups == 0;
	uint32_t   *d_offall = nullptr, *d_rowmem = nullptr, *d_vals = nullptr, *d_vals2 = nullptr, *d_flags = nullptr, *d_gpos = nullptr,
			   *d_counts = nullptr;
	int		   *d_rowlc = nullptr;
	unsigned long long *d_keys = nullptr, *d_keys2 = nullptr, *d_total = nullptr;
	void	   *d_cub = nullptr;
	size_t		cub_bytes = 0;
	const unsigned long long padkey = ((unsigned long long) nb << 8) | 0xFFull;
	int			end_bit = 1;

	while (end_bit < 64 && (padkey >> end_bit) != 0)
		end_bit++;
	if (devgrp)
	{
		const size_t maxitems = lvcap * (size_t) m;
		size_t		b1 = 0, b2 = 0;
		std::vector<uint32_t> off_all(n);
		uint32_t	e = base ? h->entry_point : NDBHIP_INVALID_BLOCK, dn = 0;
		int			el = base ? h->entry_level : -1;

		while (dn < n)
		{
			uint32_t	b = (uint32_t) std::min<int64_t>(std::max<int64_t>(((int64_t) base + dn) / batch_div, 1), batch_max);
			uint32_t	nlev = 0;

			b = std::min(b, n - dn);
			for (uint32_t i = 0; i < b; i++)
			{
				off_all[dn + i] = nlev;
				nlev += (uint32_t) std::min(lev[dn + i], std::max(el, 0)) + 1u;
			}
			/* (the entry point changes AFTER the batch: its members see the entry of its start) */
			for (uint32_t i = 0; i < b; i++)
				if (e == NDBHIP_INVALID_BLOCK || lev[dn + i] > el)
				{
					e = base + dn + 1 + i;
					el = lev[dn + i];
				}
			dn += b;
		}
		if (tmp.alloc(d_offall, (size_t) n * 4)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemcpy(d_offall, off_all.data(), (size_t) n * 4, hipMemcpyHostToDevice));
		if (tmp.alloc(d_rowmem, lvcap * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_rowlc, lvcap * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_keys, maxitems * 8)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_keys2, maxitems * 8)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_vals, maxitems * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_vals2, maxitems * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_flags, maxitems * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_gpos, maxitems * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_counts, 16)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_total, 16)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemsetAsync(d_total, 0, 8, g.stream));
		HIP_TRY(rocprim::radix_sort_pairs(nullptr, b1, d_keys, d_keys2, d_vals, d_vals2, (size_t) maxitems, 0u, (unsigned int) end_bit, (hipStream_t) g.stream));
		HIP_TRY(rocprim::exclusive_scan(nullptr, b2, d_flags, d_gpos, 0u, (size_t) maxitems, rocprim::plus<uint32_t>(), (hipStream_t) g.stream));
		cub_bytes = std::max(b1, b2);
		if (tmp.alloc(d_cub, cub_bytes)) return NDBHIP_ERR_HIP;
	}
	std::vector<uint32_t> off(bm), sid;
	std::vector<double> sd2;
	std::vector<int> sn;
	std::vector<H2Group> grp;
	std::vector<H2Req> req;
	struct Key { uint64_t key; uint32_t seq; uint32_t x; double d2; };
	std::vector<Key> keys, keys2;
	std::vector<uint32_t> radix_cnt(65537);
	uint32_t	entry = base ? h->entry_point : NDBHIP_INVALID_BLOCK;
	int			entry_level = base ? h->entry_level : -1;
	uint32_t	done = 0;
	int64_t		nbatches = 0, maxbatch = 0, nprunes = 0;

	memset(h->build_stats, 0, sizeof(h->build_stats));
	while (done < n)
	{
		uint32_t	b = (uint32_t) std::min<int64_t>(std::max<int64_t>(((int64_t) base + done) / batch_div, 1), batch_max);

		b = std::min(b, n - done);
		const uint32_t first = base + done + 1;

		if (entry != NDBHIP_INVALID_BLOCK)
		{
			uint32_t	nlev = 0;

			for (uint32_t i = 0; i < b; i++)
			{
				off[i] = nlev;
				nlev += (uint32_t) std::min(lev[done + i], entry_level) + 1u;
			}
			const uint32_t *d_off_b = devgrp ? d_offall + done : d_off;

			if (!devgrp)
				HIP_TRY(hipMemcpyAsync(d_off, off.data(), (size_t) b * 4, hipMemcpyHostToDevice, g.stream));
			HIP_TRY(hipMemsetAsync(d_next, 0, 4, g.stream));
			hipLaunchKernelGGL(k_h2_insert_search, dim3(std::min(b, nwaves)), dim3(64), smem, g.stream, h2_graph(h, first), first, b,
							   (uint32_t) ef_construction, g_h2_select, entry, entry_level, d_off_b, d_sid, d_sd2,
							   d_sn, h->w_vbits, h->w_vlog, nwords, d_next);
			HIP_TRY(hipGetLastError());
			if (devgrp)
			{
				const uint32_t nitems = nlev * (uint32_t) m;
				size_t		cb = cub_bytes;

				hipLaunchKernelGGL(k_h2_bl_rows, dim3((b + 255) / 256), dim3(256), 0, g.stream, b, d_off_b, (const int *) h->d_levels + first,
								   entry_level, d_rowmem, d_rowlc);
				hipLaunchKernelGGL(k_h2_bl_keys, dim3((nitems + 255) / 256), dim3(256), 0, g.stream, nitems, (uint32_t) m, (const int *) d_sn,
								   (const uint32_t *) d_sid, (const int *) d_rowlc, padkey, d_keys, d_vals);
				HIP_TRY(rocprim::radix_sort_pairs(d_cub, cb, d_keys, d_keys2, d_vals, d_vals2, (size_t) nitems, 0u, (unsigned int) end_bit, (hipStream_t) g.stream));
				hipLaunchKernelGGL(k_h2_bl_heads, dim3((nitems + 255) / 256), dim3(256), 0, g.stream, nitems, (const unsigned long long *) d_keys2,
								   padkey, d_flags);
				cb = cub_bytes;
				HIP_TRY(rocprim::exclusive_scan(d_cub, cb, d_flags, d_gpos, 0u, (size_t) nitems, rocprim::plus<uint32_t>(), (hipStream_t) g.stream));
				HIP_TRY(hipMemsetAsync(d_counts, 0, 8, g.stream));
				hipLaunchKernelGGL(k_h2_bl_groups, dim3((nitems + 255) / 256), dim3(256), 0, g.stream, nitems, (uint32_t) m, first,
								   (const unsigned long long *) d_keys2, (const uint32_t *) d_vals2, padkey, (const uint32_t *) d_flags,
								   (const uint32_t *) d_gpos, (const uint32_t *) d_rowmem, (const double *) d_sd2, d_grp, d_req, d_counts, d_total);
				hipLaunchKernelGGL(k_h2_apply, dim3((unsigned) std::min<size_t>((size_t) nitems, (size_t) g.num_cus * 16)), dim3(64), 0, g.stream,
								   h2_graph(h, first + b), (const H2Group *) d_grp, 0u, (const uint32_t *) d_counts, (const H2Req *) d_req, g_h2_select);
				HIP_TRY(hipGetLastError());
			}
			else
			{
			sid.resize((size_t) nlev * m);
			sd2.resize((size_t) nlev * m);
			sn.resize(nlev);
			HIP_TRY(hipMemcpyAsync(sid.data(), d_sid, sid.size() * 4, hipMemcpyDeviceToHost, g.stream));
			HIP_TRY(hipMemcpyAsync(sd2.data(), d_sd2, sd2.size() * 8, hipMemcpyDeviceToHost, g.stream));
			HIP_TRY(hipMemcpyAsync(sn.data(), d_sn, sn.size() * 4, hipMemcpyDeviceToHost, g.stream));
			HIP_TRY(hipStreamSynchronize(g.stream));
			/* the back-links, grouped by target (node, level), inside a group in insertion order */
			keys.clear();
			for (uint32_t i = 0; i < b; i++)
			{
				const int	top = std::min(lev[done + i], entry_level);

				for (int lc = top; lc >= 0; lc--)
				{
					const size_t row = (size_t) off[i] + (size_t) (top - lc);

					for (int j = 0; j < sn[row]; j++)
					{
						Key			kx;

						kx.key = ((uint64_t) sid[row * m + j] << 8) | (uint64_t) lc;
						kx.seq = (uint32_t) keys.size();
						kx.x = first + i;
						kx.d2 = sd2[row * m + j];
						keys.push_back(kx);
					}
				}
			}
			/* by (target, insertion order): the keys were made in insertion order, so a STABLE sort by target — least
			 * significant digit first, 16 bits a pass over (node << 8 | level) — is that order (a comparison sort of the
			 * ~140 k requests of a full batch was 10 ms of every batch on the host, with the device waiting) */
			if (keys.size() > 4096)
			{
				uint64_t	maxkey = 0;

				for (const Key &kx : keys)
					maxkey = std::max(maxkey, kx.key);
				keys2.resize(keys.size());
				for (int shift = 0; shift < 64 && (maxkey >> shift) != 0; shift += 16)
				{
					std::fill(radix_cnt.begin(), radix_cnt.end(), 0u);
					for (const Key &kx : keys)
						radix_cnt[(size_t) ((kx.key >> shift) & 0xFFFFu) + 1]++;
					for (size_t d = 1; d <= 65536; d++)
						radix_cnt[d] += radix_cnt[d - 1];
					for (const Key &kx : keys)
						keys2[radix_cnt[(size_t) ((kx.key >> shift) & 0xFFFFu)]++] = kx;
					keys.swap(keys2);
				}
			}
			else
				std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &c) { return a.key < c.key || (a.key == c.key && a.seq < c.seq); });
			grp.clear();
			req.resize(keys.size());
			for (size_t r = 0; r < keys.size(); r++)
			{
				req[r].x = keys[r].x;
				req[r].pad = 0;
				req[r].d2 = keys[r].d2;
				if (r == 0 || keys[r].key != keys[r - 1].key)
				{
					H2Group		gr;

					gr.node = (uint32_t) (keys[r].key >> 8);
					gr.level = (int) (keys[r].key & 0xFF);
					gr.r0 = (uint32_t) r;
					gr.r1 = (uint32_t) r;
					grp.push_back(gr);
				}
				grp.back().r1 = (uint32_t) r + 1;
			}
			if (!grp.empty())
			{
				HIP_TRY(hipMemcpyAsync(d_grp, grp.data(), grp.size() * sizeof(H2Group), hipMemcpyHostToDevice, g.stream));
				HIP_TRY(hipMemcpyAsync(d_req, req.data(), req.size() * sizeof(H2Req), hipMemcpyHostToDevice, g.stream));
				hipLaunchKernelGGL(k_h2_apply, dim3((unsigned) std::min<size_t>(grp.size(), (size_t) g.num_cus * 16)), dim3(64), 0, g.stream,
								   h2_graph(h, first + b), (const H2Group *) d_grp, (uint32_t) grp.size(), (const uint32_t *) nullptr,
								   (const H2Req *) d_req, g_h2_select);
				HIP_TRY(hipGetLastError());
				HIP_TRY(hipStreamSynchronize(g.stream));		/* grp / req are reused by the next batch */
			}
			nprunes += (int64_t) keys.size();
			}
		}
		/* the entry point: the first node of every new top level, in insertion order */
		for (uint32_t i = 0; i < b; i++)
			if (entry == NDBHIP_INVALID_BLOCK || lev[done + i] > entry_level)
			{
				entry = first + i;
				entry_level = lev[done + i];
			}
		done += b;
		nbatches++;
		maxbatch = std::max<int64_t>(maxbatch, b);
	}
	HIP_TRY(hipStreamSynchronize(g.stream));		/* (the scratch above is freed on return) */
	if (devgrp)
	{
		unsigned long long tot = 0;

		HIP_TRY(hipMemcpy(&tot, d_total, 8, hipMemcpyDeviceToHost));
		nprunes = (int64_t) tot;
	}
	h->entry_point = entry;
	h->entry_level = entry_level;
	h->loaded = true;
	h->build_stats[4] = nbatches;
	h->build_stats[5] = maxbatch;
	h->build_stats[0] = nprunes;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_build_intended_device(ndbhip_hnsw *h, const float *d_rows, const uint64_t *d_tids, uint32_t n,
								  const int32_t *levels, int ef_construction, int batch_div, int batch_max)
{
	return h2_build_rows(h, d_rows, d_tids, n, levels, ef_construction, batch_div, batch_max, 0);
}

/* hnswinsert under `intended` (src/index/hnsw_am.c:478-538): n MORE rows on top of the graph the mirror holds — built here in
 * either mode, or loaded (a loaded graph is first given the dense layout): node nblocks + i = row i, the batch schedule goes
 * on from the relation's size.  On an empty mirror this is ndbhip_hnsw_build_intended_device. */
extern "C" int
ndbhip_hnsw_insert_intended_device(ndbhip_hnsw *h, const float *d_rows, const uint64_t *d_tids, uint32_t n,
								   const int32_t *levels, int ef_construction, int batch_div, int batch_max)
{
	if (!h)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	return h2_build_rows(h, d_rows, d_tids, n, levels, ef_construction, batch_div, batch_max,
						 (!h->loaded || h->nblocks < 1) ? 0u : h->nblocks - 1);
}

/* ... and for host rows / heapPtrs (6 bytes each), staged by the library: what ndb_hnswinsert calls when neurondb.ref_compat
 * is off */
extern "C" int
ndbhip_hnsw_insert_intended(ndbhip_hnsw *h, const float *rows, const uint8_t *tids6, uint32_t n, const int32_t *levels,
							int ef_construction, int batch_div, int batch_max)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	HNSW_NOT_FROZEN(h, "ndbhip_hnsw_insert_intended");
	if (!h || !rows || !tids6 || !levels || n < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	float	   *d_rows = nullptr;
	uint64_t   *d_tids = nullptr;
	std::vector<uint64_t> t64(n);

	for (uint32_t i = 0; i < n; i++)
		t64[i] = ndb_tid_pack(tids6 + (size_t) i * 6);
	HIP_TRY(hipMalloc((void **) &d_rows, (size_t) n * h->dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_tids, (size_t) n * sizeof(uint64_t)));
	HIP_TRY(hipMemcpyAsync(d_rows, rows, (size_t) n * h->dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_tids, t64.data(), (size_t) n * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
	const int	rc = ndbhip_hnsw_insert_intended_device(h, d_rows, d_tids, n, levels, ef_construction, batch_div, batch_max);

	(void) hipStreamSynchronize(g.stream);
	(void) hipFree(d_rows);
	(void) hipFree(d_tids);
	return rc;
}

/* kNN search of the `intended` mode on a dense mirror (built by ndbhip_hnsw_build_intended_device, or any graph):
 * greedy descent, best-first layer search with ef at level 0, the k nearest ascending.  strategy = the operator class's
 * (hnsw_am.c:918-921: 1 L2 — distances (float) sqrt(d2) —, 2 cosine, 3 negative inner product: the walk orders by that
 * metric's fp64 key, the result set is scored with hnswComputeDistance's own arithmetic, :1321-1337, and ordered by
 * that float4; oracle/ndb_oracle_hnsw2.c "THE OPERATOR CLASS'S METRIC").  Device pointers, asynchronous. */
static int
h2_search_run(ndbhip_hnsw *h, bool w16, const float *d_queries, int nq, int strategy, int ef, int k, uint32_t *d_out_blocks,
			  float *d_out_dist, int *d_out_count, uint64_t *d_out_tids, int64_t *d_out_evals)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!h || !h->loaded)
		return fail(NDBHIP_ERR_STATE, "graph not loaded");
	if (nq < 0 || (nq > 0 && (!d_queries || !d_out_blocks || !d_out_dist || !d_out_count)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (k < 1 || k > NDBHIP_MAX_K || ef < 1 || ef > NDBHIP_MAX_EF)
		return fail(NDBHIP_ERR_INVALID, "k or ef out of range");
	if (strategy < 1 || strategy > 3)
		return fail(NDBHIP_ERR_INVALID, "hnsw: unsupported distance strategy %d", strategy);		/* hnsw_am.c:1339-1343 */
	if (nq == 0)
		return NDBHIP_OK;
	int			rc = hnsw_densify(h);

	if (rc)
		return rc;
	if (w16)
	{
		if (h->dim % 4 != 0 || h->dim > 64 * H2_QREG)
			return fail(NDBHIP_ERR_UNSUPPORTED, "walk rows need dim %% 4 == 0 and dim <= %d (dim = %d)", 64 * H2_QREG, h->dim);
		if ((!h->d_vecs16 || h->w16_blocks != h->nblocks) && hnsw_frozen(h))
			return fail(NDBHIP_ERR_STATE, "the graph is shared (ndbhip_hnsw_share) and has no walk rows: run ndbhip_hnsw_search_intended_w16_device on the source before sharing");
		if (!h->d_vecs16 || h->w16_blocks != h->nblocks)
		{
			/* (rows are only ever appended: a twin that covers fewer blocks than the graph is stale as a whole — made again) */
			const size_t nel = (size_t) h->nblocks * h->dim;

			if (h->d_vecs16) { HIP_TRY(hipStreamSynchronize(g.stream)); HIP_TRY(hipFree(h->d_vecs16)); h->d_vecs16 = nullptr; }
			HIP_TRY(hipMalloc((void **) &h->d_vecs16, nel * sizeof(uint16_t)));
			hipLaunchKernelGGL(k_h2_walk_rows, dim3((unsigned) std::min<size_t>((nel + 255) / 256, (size_t) 1 << 20)), dim3(256), 0, g.stream,
							   (const float *) h->d_vecs, h->d_vecs16, nel);
			HIP_TRY(hipGetLastError());
			h->w16_blocks = h->nblocks;
		}
	}
	const int	rv = w16 ? 1 : 0;

	if (strategy == 2 && (!h->d_rinv[rv] || h->rinv_blocks[rv] != h->nblocks))
	{
		/* the nodes' factors of the cosine walk key: 1 / |row| of the rows this walk reads */
		if (hnsw_frozen(h))
			return fail(NDBHIP_ERR_STATE, "the graph is shared (ndbhip_hnsw_share) and has no row norms for this walk: run a cosine search of the same kind on the source before sharing");
		if (h->d_rinv[rv]) { HIP_TRY(hipStreamSynchronize(g.stream)); HIP_TRY(hipFree(h->d_rinv[rv])); h->d_rinv[rv] = nullptr; }
		HIP_TRY(hipMalloc((void **) &h->d_rinv[rv], (size_t) h->nblocks * sizeof(double)));
		const unsigned nb4 = (unsigned) std::min<uint32_t>((h->nblocks + 3u) / 4u, 1u << 16);

		if (w16)
			hipLaunchKernelGGL(k_h2_rinv<1>, dim3(nb4), dim3(256), 0, g.stream, (const float *) h->d_vecs, (const uint16_t *) h->d_vecs16, h->dim, h->nblocks, h->d_rinv[rv]);
		else
			hipLaunchKernelGGL(k_h2_rinv<0>, dim3(nb4), dim3(256), 0, g.stream, (const float *) h->d_vecs, (const uint16_t *) nullptr, h->dim, h->nblocks, h->d_rinv[rv]);
		HIP_TRY(hipGetLastError());
		h->rinv_blocks[rv] = h->nblocks;
	}
	const uint32_t efe = (uint32_t) std::max(ef, k);
	const uint32_t nwaves = (uint32_t) std::min<int64_t>((int64_t) g.num_cus * g_h2_waves, nq);
	uint32_t	nwords = 0;

	if (h2_workspace(h, nwaves, h->nblocks, &nwords)) return NDBHIP_ERR_HIP;
	const size_t smem = h2_smem_bytes(efe);

	uint32_t   *d_next = h->w_vlog + (size_t) nwaves * H2_LOG_CAP;

	HIP_TRY(hipMemsetAsync(d_next, 0, 4, g.stream));
	H2Graph		gr = h2_graph(h, h->nblocks);

	if (strategy == 2)
		gr.rinv = h->d_rinv[rv];
#define H2_SEARCH_L(KK, ...) do { \
		HIP_TRY(hipFuncSetAttribute((const void *) KK<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem)); \
		hipLaunchKernelGGL(HIP_KERNEL_NAME(KK<__VA_ARGS__>), dim3(nwaves), dim3(64), smem, g.stream, gr, d_queries, (uint32_t) nq, efe, \
						   (uint32_t) k, h->entry_point, h->entry_level, (const uint64_t *) h->d_tids, h->w_vbits, h->w_vlog, nwords, \
						   d_out_blocks, d_out_dist, d_out_count, d_out_tids, (long long *) d_out_evals, d_next); } while (0)
	/* (g_h2_occ4: 1 = four walkers a SIMD where that costs no scratch — walk rows of dim <= 768; 2 = everywhere; 0 = nowhere;
	 * L2 only: the other strategies' walks hold a second partial sum per row) */
	const int	ng = !w16 ? 0 : (h->dim <= 256 ? 1 : (h->dim <= 512 ? 2 : (h->dim <= 768 ? 3 : 4)));

#define H2_SEARCH_S(NGG) do { \
		if (strategy == 2) H2_SEARCH_L(k_h2_search, NGG, 2); \
		else if (strategy == 3) H2_SEARCH_L(k_h2_search, NGG, 3); \
		else if (g_h2_occ4 >= ((NGG) >= 1 && (NGG) <= 3 ? 1 : 2)) H2_SEARCH_L(k_h2_search4, NGG); \
		else H2_SEARCH_L(k_h2_search, NGG, 1); } while (0)
	switch (ng)
	{
		case 0: H2_SEARCH_S(0); break;
		case 1: H2_SEARCH_S(1); break;
		case 2: H2_SEARCH_S(2); break;
		case 3: H2_SEARCH_S(3); break;
		default: H2_SEARCH_S(4); break;
	}
#undef H2_SEARCH_S
#undef H2_SEARCH_L
	HIP_TRY(hipGetLastError());
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_search_intended_device(ndbhip_hnsw *h, const float *d_queries, int nq, int strategy, int ef, int k, uint32_t *d_out_blocks,
								   float *d_out_dist, int *d_out_count, uint64_t *d_out_tids, int64_t *d_out_evals)
{
	return h2_search_run(h, false, d_queries, nq, strategy, ef, k, d_out_blocks, d_out_dist, d_out_count, d_out_tids, d_out_evals);
}

/* The same search with the WALK on fp16 walk rows — every element of the graph's rows through the reference's own
 * float4_to_fp16 (src/types/quantization.c:141-168), what a halfvec column of the same data holds; made on the device at
 * the first call and again after rows were appended (+ 0.5 x the rows' bytes) — and the result set's ef entries scored
 * against the float4 rows with the definition's arithmetic: oracle/ndb_oracle_hnsw2.c ndbo_h2_search_w16, equal id for id
 * and bit for bit.  A walk fetches half the bytes per evaluated row. */
extern "C" int
ndbhip_hnsw_search_intended_w16_device(ndbhip_hnsw *h, const float *d_queries, int nq, int strategy, int ef, int k, uint32_t *d_out_blocks,
									   float *d_out_dist, int *d_out_count, uint64_t *d_out_tids, int64_t *d_out_evals)
{
	return h2_search_run(h, true, d_queries, nq, strategy, ef, k, d_out_blocks, d_out_dist, d_out_count, d_out_tids, d_out_evals);
}

extern "C" int
ndbhip_hnsw_set_intended_select(int select)
{
	if (select < 0 || select > 7 || (select & 5) == 4)
		return fail(NDBHIP_ERR_INVALID, "select: bit 0 = the heuristic (else the nearest), bit 1 = up to 2m links for a new node at level 0, "
					"bit 2 (with bit 0) = places the heuristic leaves empty go to the nearest candidates it passed over");
	g_h2_select = select;
	return NDBHIP_OK;
}

/* profiling builds (-DNDB_PHASES): the intended search's phase clocks (ndbhip_hnsw2.h), read and reset; otherwise zeros */
extern "C" int
ndbhip_debug_h2_phases(unsigned long long *out)
{
	if (!out)
		return NDBHIP_ERR_INVALID;
#ifdef NDB_PHASES
	unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};

	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h2_phases), sizeof(zero)));
	HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_h2_phases), zero, sizeof(zero)));
#else
	memset(out, 0, 8 * sizeof(unsigned long long));
#endif
	return NDBHIP_OK;
}

