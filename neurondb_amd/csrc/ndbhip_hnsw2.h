/*
 * ndbhip_hnsw2.h — the `intended` HNSW on the device (part of ndbhip_hnsw.hip's translation unit): the graph SURVEY 8f-2
 * asks for next to the bug-compatible hnswInsertNode / hnswSearch, built and searched in HBM.
 *
 * The algorithm and its relation to the reference (what is kept: page-level data model, injected level draws, L2 for
 * every build-time comparison, the best-first layer search of src/scan/hnsw_scan.c:379-483, 645-844 as the search's
 * specification; what is repaired: the greedy descent's result is used, every level is searched on its own links, a
 * full list is pruned instead of the back-link being dropped — src/index/hnsw_am.c:2155-2286, 2503-2513) are stated
 * once, in oracle/ndb_oracle_hnsw2.c, whose sequential run this file reproduces slot for slot
 * (tests/test_gpu_hnsw2.py).  Three things make that possible:
 *   - one arithmetic: squared L2 in fp64, terms (double) fl32(a_i - b_i) squared, 64 strided partial sums — ONE LANE
 *     EACH — folded by the xor butterfly 32 .. 1; every comparison is on (d2, block number), so nothing ties;
 *   - order-free steps: the result set of a layer search after expanding a node is the best ef of (what it was) +
 *     (the node's unvisited neighbours), whatever order they are offered in;
 *   - a batch-synchronous schedule that is PART OF THE DEFINITION: the members of a batch search the graph as it
 *     stood when the batch began (k_h2_insert_search, one wave per member, all in parallel), then their links are
 *     applied in insertion order — lists of different (node, level) pairs evolve independently, so the requests
 *     are grouped by target on the host and every target replays its own requests in order (k_h2_apply, one wave
 *     per target).
 * Distance evaluations dominate: a wave reads a 3 KB row with twelve coalesced 256-byte loads, four rows in flight.
 */
#ifndef NDBHIP_HNSW2_H
#define NDBHIP_HNSW2_H

/* profiling builds (make EXTRA=-DNDB_PHASES): 100 MHz clock ticks block 0's wave spends in the parts of a layer search,
 * summed over the launch — [0] pick, [1] neighbour list + visited marks, [2] row distances, [3] offers, [4] expansions */
#ifdef NDB_PHASES
__device__ unsigned long long g_h2_phases[8];
#define H2_PH_DECL unsigned long long h2_ph_t = wall_clock64(), h2_ph_acc[5] = {0, 0, 0, 0, 0}
#define H2_PH(I) do { const unsigned long long h2_now = wall_clock64(); h2_ph_acc[I] += h2_now - h2_ph_t; h2_ph_t = h2_now; } while (0)
#define H2_PH_COUNT(I) (h2_ph_acc[I]++)
#define H2_PH_FLUSH do { if (blockIdx.x == 0 && lane == 0) for (int h2_i = 0; h2_i < 5; h2_i++) atomicAdd(&g_h2_phases[h2_i], h2_ph_acc[h2_i]); } while (0)
#else
#define H2_PH_DECL ((void) 0)
#define H2_PH(I) ((void) 0)
#define H2_PH_COUNT(I) ((void) 0)
#define H2_PH_FLUSH ((void) 0)
#endif

#ifndef H2_JG_DEF
#define H2_JG_DEF 6			/* strides of a row batch requested together (h2_dist2x4) */
#endif
#define H2_QREG 16				/* query elements a lane keeps in registers (dim <= 1024); beyond: re-read (L1 / L2) */
#define H2_LOG_CAP 8192			/* visited blocks a wave logs for clearing its bitmap; more: the whole map is cleared */

struct H2Graph
{
	const float *vecs;
	const uint16_t *vecs16;		/* walk rows (round 5): float4_to_fp16 of every element of vecs, same stride; nullptr = none */
	const double *rinv;			/* strategy 2 (round 6): 1 / sqrt(sum of squares) of the rows THIS walk reads (vecs, or vecs16), 0 for a
								 * row of zeros; nullptr for the other strategies */
	const int  *levels;
	int16_t    *ncount;
	uint32_t   *nbrs;
	int64_t		stride;			/* 16 levels x 2m slots */
	uint32_t	nvisible;		/* blocks below this are linked (what a frozen search may meet) */
	int			dim;
	int			m;
};

__device__ __forceinline__ double
h2_wave_fold(double p)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		p = p + __shfl_xor(p, off, 64);
	return p;
}

__device__ __forceinline__ bool
h2_less(double d, uint32_t id, double e, uint32_t jd)
{
	return d < e || (d == e && id < jd);
}

/* the wave's query: elements lane, lane + 64, ... in registers while they fit */
struct H2Query
{
	const float *q;
	float		r[H2_QREG];
	int			dim;
	__device__ __forceinline__ void load(const float *qq, int d, int lane)
	{
		q = qq;
		dim = d;
#pragma unroll
		for (int j = 0; j < H2_QREG; j++)
			r[j] = lane + 64 * j < d ? qq[lane + 64 * j] : 0.0f;
	}

	/* for walk rows: the lane's GROUPS of four elements, group lane, lane + 64, ... (d % 4 == 0, d <= 64 * H2_QREG) */
	__device__ __forceinline__ void load16(const float *qq, int d, int lane)
	{
		q = qq;
		dim = d;
#pragma unroll
		for (int j = 0; j < H2_QREG / 4; j++)
		{
			const int	i = (lane + 64 * j) * 4;
#pragma unroll
			for (int t = 0; t < 4; t++)
				r[4 * j + t] = i < d ? qq[i + t] : 0.0f;
		}
	}
};

/*
 * The walk key under the operator class's strategy (oracle/ndb_oracle_hnsw2.c "THE OPERATOR CLASS'S METRIC"; S = 1 L2 — the
 * definition's d2, what every build-time comparison uses —, 2 cosine, 3 negative inner product): a lane's term per element
 * and the wave's key out of the folded partial sums.
 */
template <int S>
__device__ __forceinline__ void
h2_term(double &p, float qv, float xv)
{
	if constexpr (S == 1)
	{
		const float d = qv - xv;

		p += (double) d * (double) d;
	}
	else
		p += (double) qv * (double) xv;		/* (a product of two float4 values is exact in fp64) */
}

/* S = 1: d2; 3: -dot; 2: -dot as well — the node's factor rinv is multiplied in by h2_ids_d2, which knows the node */
template <int S>
__device__ __forceinline__ double
h2_key_of(double p)
{
	if constexpr (S == 1)
		return h2_wave_fold(p);
	else
		return -h2_wave_fold(p);
}

/* d2(query, row x) by the whole wave (every lane returns it) */
__device__ __forceinline__ double
h2_dist2(const H2Query &Q, const float *__restrict__ x, int lane)
{
	double		p = 0.0;

	if (Q.dim <= 64 * H2_QREG)
	{
#pragma unroll
		for (int j = 0; j < H2_QREG; j++)
			if (lane + 64 * j < Q.dim)
			{
				const float d = Q.r[j] - x[lane + 64 * j];

				p += (double) d * (double) d;
			}
	}
	else
		for (int i = lane; i < Q.dim; i += 64)
		{
			const float d = Q.q[i] - x[i];

			p += (double) d * (double) d;
		}
	return h2_wave_fold(p);
}

/* H2_NR rows at a time: the loads of all of them are in flight before the first sum is folded (a walk is a chain of
 * dependent fetches: what it can overlap is the rows of ONE expansion — up to 2m unvisited neighbours, offered to the
 * set in any order with the same result) */
#define H2_NR 8
template <int S = 1>
__device__ __forceinline__ void
h2_dist2x4(const H2Query &Q, const float *const x[H2_NR], int n, int lane, double out[H2_NR])
{
	double		p[H2_NR];

#pragma unroll
	for (int u = 0; u < H2_NR; u++)
		p[u] = 0.0;

	if (Q.dim <= 64 * H2_QREG)
	{
		/* H2_JG strides of all the rows requested together: 12 strides one after the other, each waiting for its loads, were
		 * twelve memory round trips per expansion — 20 of its 28 us (phase clocks, tools/h2_bench.py on a profiling build).
		 * A lane still adds its elements lane, lane + 64, ... in that order. */
		constexpr int H2_JG = H2_JG_DEF;

#pragma unroll
		for (int j0 = 0; j0 < H2_QREG; j0 += H2_JG)
		{
			if (64 * j0 >= Q.dim)		/* uniform */
				break;
			float		v[H2_NR][H2_JG];

#pragma unroll
			for (int u = 0; u < H2_NR; u++)
#pragma unroll
				for (int jj = 0; jj < H2_JG; jj++)
				{
					const int	i = lane + 64 * (j0 + jj);

					/* (beyond the row: 0 against the query's 0 — a term +0.0 that leaves the sum as it is) */
					if constexpr (S == 1)
						v[u][jj] = (u < n && j0 + jj < H2_QREG && i < Q.dim) ? x[u][i] : 0.0f;
					else
					{
						/* (the other strategies widen the loaded value at once, and the compiler then puts every guarded load in a
						 * branch of its own with a full wait behind it — 326 waits in the kernel against 88, the float4 walk at a
						 * quarter of its rate —: the load is made unconditional (a row the wave does not need reads row 0, an element
						 * beyond the row its first stride), the guard picks the value) */
						const bool	ok = u < n && j0 + jj < H2_QREG && i < Q.dim;
						const float raw = x[u][i < Q.dim ? i : lane];

						v[u][jj] = ok ? raw : 0.0f;
					}
				}
#pragma unroll
			for (int jj = 0; jj < H2_JG; jj++)
				if (j0 + jj < H2_QREG && lane + 64 * (j0 + jj) < Q.dim)
				{
#pragma unroll
					for (int u = 0; u < H2_NR; u++)
						h2_term<S>(p[u], Q.r[j0 + jj], v[u][jj]);
				}
		}
	}
	else
		for (int i = lane; i < Q.dim; i += 64)
		{
			const float qv = Q.q[i];

#pragma unroll
			for (int u = 0; u < H2_NR; u++)
				if (u < n)
					h2_term<S>(p[u], qv, x[u][i]);
		}
#pragma unroll
	for (int u = 0; u < H2_NR; u++)
		out[u] = h2_key_of<S>(p[u]);
}

/*
 * The same on WALK ROWS (oracle/ndb_oracle_hnsw2.c "WALK ROWS", ndbo_h2_dist2_w16): fp16 images of the rows, a lane
 * asks for 8 bytes — the four halves of group lane + 64 j — per request (a 768-dim row is three requests of 512
 * contiguous bytes instead of twelve of 256), decodes them (the reference's fp16_to_float; the encoder leaves no
 * subnormal, so the hardware conversion is that function) and adds the four terms in increasing element order to its
 * fp64 partial: element i goes to partial (i / 4) mod 64.  Query registers in load16's order.
 */
template <int NG, int S = 1>
__device__ __forceinline__ void
h2w_dist2x(const H2Query &Q, const uint16_t *const x[H2_NR], int n, int lane, double out[H2_NR])
{
	double		p[H2_NR];
	uint2		v[H2_NR][NG];

#pragma unroll
	for (int u = 0; u < H2_NR; u++)
	{
		p[u] = 0.0;
#pragma unroll
		for (int j = 0; j < NG; j++)
		{
			const int	i = (lane + 64 * j) * 4;

			v[u][j] = (u < n && i < Q.dim) ? *(const uint2 *) (x[u] + i) : make_uint2(0u, 0u);
		}
	}
#pragma unroll
	for (int j = 0; j < NG; j++)
		if ((lane + 64 * j) * 4 < Q.dim)
		{
#pragma unroll
			for (int u = 0; u < H2_NR; u++)
			{
				const uint32_t w[2] = {v[u][j].x, v[u][j].y};

#pragma unroll
				for (int t = 0; t < 4; t++)
				{
					const unsigned short h = (unsigned short) ((t & 1) ? (w[t >> 1] >> 16) : (w[t >> 1] & 0xFFFFu));

					h2_term<S>(p[u], Q.r[4 * j + t], __half2float(__ushort_as_half(h)));
				}
			}
		}
#pragma unroll
	for (int u = 0; u < H2_NR; u++)
		out[u] = h2_key_of<S>(p[u]);
}

/* d2(query, node ids[u]) for u < n: on the walk rows (W16; Q loaded by load16) or on the float4 rows (Q loaded by load) */
template <int W16, int S = 1>		/* 0: float4 rows; NG = 1 .. 4: walk rows of dim <= 256 NG (groups of four a lane holds) */
__device__ __forceinline__ void
h2_ids_d2(const H2Graph &g, const H2Query &Q, const uint32_t ids[H2_NR], int n, int lane, double d[H2_NR])
{
	double		ri[S == 2 ? H2_NR : 1];

	if constexpr (S == 2)
	{
		/* the nodes' factors (wave-uniform addresses: they travel with the rows' loads) */
#pragma unroll
		for (int u = 0; u < H2_NR; u++)
			ri[u] = u < n ? g.rinv[ids[u]] : 0.0;
	}
	if constexpr (W16 != 0)
	{
		const uint16_t *x[H2_NR];

#pragma unroll
		for (int u = 0; u < H2_NR; u++)
			x[u] = g.vecs16 + (size_t) ids[u] * g.dim;
		h2w_dist2x<W16, S>(Q, x, n, lane, d);
	}
	else
	{
		const float *x[H2_NR];

#pragma unroll
		for (int u = 0; u < H2_NR; u++)
			x[u] = g.vecs + (size_t) ids[u] * g.dim;
		h2_dist2x4<S>(Q, x, n, lane, d);
	}
	if constexpr (S == 2)
	{
#pragma unroll
		for (int u = 0; u < H2_NR; u++)
			d[u] = d[u] * ri[u];		/* (-dot) * rinv: oracle ndbo_h2_walk_key */
	}
}

/* d2 of two rows of the graph */
__device__ __forceinline__ double
h2_dist2_rows(const float *__restrict__ a, const float *__restrict__ b, int dim, int lane)
{
	double		p = 0.0;

	for (int i = lane; i < dim; i += 64)
	{
		const float d = a[i] - b[i];

		p += (double) d * (double) d;
	}
	return h2_wave_fold(p);
}

/* a wave's visited set: one bit per block in global memory (all zero at rest), the blocks it set logged for clearing */
#define H2_HV_LOG2 11
#define H2_HV (1u << H2_HV_LOG2)	/* slots of the LDS visited table of a search; three quarters full = it moves to the bitmap */
/* (an expansion marks at most 2 m = 64 blocks between two checks of the fill: 3/4 of the table + 64 stays below its size,
 * so the probe loop always finds a free slot) */
static_assert(H2_HV == 2048u && (H2_HV / 4u) * 3u + 64u < H2_HV, "the visited table's hash shift and its fill bound go with H2_HV_LOG2");
#define H2_HV_MAX_EF 256			/* searches up to this ef start on the table (8 KB of LDS a walker) */
struct H2Visited
{
	uint32_t   *bits;
	uint32_t   *log;
	uint32_t	nwords;
	uint32_t	nlog;			/* uniform */
	/* hv != NULL: the visited blocks of this search are an open-addressing table in LDS instead (block numbers are >= 1:
	 * 0 = empty slot) — a search at ef 64 meets ~850 nodes, and marking 32 neighbours per expansion in a bitmap of one
	 * bit per node (0.5 GB over the walkers) was 32 atomics on 32 random lines of HBM, a sixth of the walk's traffic.
	 * More than 3/4 H2_HV marks: `over` is set, and before its next expansion the search moves the table's blocks into the
	 * bitmap and carries on there (migrate). */
	uint32_t   *hv;
	uint32_t	nhv;			/* uniform */
	bool		over;			/* uniform */

	__device__ __forceinline__ bool mark_lds(bool act, uint32_t b)
	{
		bool		fresh = false;

		if (act)
		{
			uint32_t	h = (b * 2654435761u) >> (32 - H2_HV_LOG2);

			for (;;)
			{
				const uint32_t old = atomicCAS(&hv[h], 0u, b);

				if (old == 0u)
				{
					fresh = true;
					break;
				}
				if (old == b)
					break;
				h = (h + 1u) & (H2_HV - 1u);
			}
		}
		nhv += (uint32_t) __popcll(__ballot(fresh));
		if (nhv > (H2_HV / 4u) * 3u)
			over = true;
		return fresh;
	}
	__device__ __forceinline__ void clear_lds(int lane)
	{
		for (uint32_t i = lane; i < H2_HV; i += 64)
			hv[i] = 0u;
		nhv = 0;
		over = false;
		__threadfence_block();
	}

	/* the table is three quarters full: its blocks go into the wave's bitmap and the search carries on there (same set of
	 * visited blocks: nothing observable changes) */
	__device__ __forceinline__ void migrate(int lane)
	{
		uint32_t   *const t = hv;

		hv = nullptr;
		for (uint32_t i0 = 0; i0 < H2_HV; i0 += 64)
		{
			const uint32_t b = t[i0 + lane];

			(void) mark(b != 0u, b, lane);
			t[i0 + lane] = 0u;
		}
		nhv = 0;
		over = false;
		__threadfence_block();
	}

	/* lanes with act: was block b unvisited (and now marked)?  One lane wins where several name the same block. */
	__device__ __forceinline__ bool mark(bool act, uint32_t b, int lane)
	{
		if (hv)
			return mark_lds(act, b);
		bool		fresh = false;

		if (act)
		{
			const uint32_t bit = 1u << (b & 31u);

			fresh = !(atomicOr(&bits[b >> 5], bit) & bit);
		}
		const unsigned long long mk = __ballot(fresh);

		if (fresh)
		{
			const uint32_t slot = nlog + (uint32_t) __popcll(mk & ((1ull << lane) - 1ull));

			if (slot < H2_LOG_CAP)
				log[slot] = b;
		}
		nlog += (uint32_t) __popcll(mk);
		return fresh;
	}
	__device__ __forceinline__ void clear(int lane)
	{
		if (hv)
		{
			clear_lds(lane);
			return;
		}
		if (nlog <= H2_LOG_CAP)
			for (uint32_t i = lane; i < nlog; i += 64)
				bits[log[i] >> 5] = 0;
		else
			for (uint32_t i = lane; i < nwords; i += 64)
				bits[i] = 0;
		nlog = 0;
		__threadfence_block();
	}
};

/* the wave's result set of a layer search, in LDS: wd / wid / wx[ef] (wx: already expanded) */
struct H2Set
{
	double	   *wd;
	uint32_t   *wid;
	uint8_t    *wx;
	uint32_t	ef;
	uint32_t	nw;				/* uniform */
	/* the farthest entry of a FULL set, remembered between offers (worst_i < 0: not known) — most offers to a full set
	 * lose against it and need nothing else */
	double		worst_d;
	uint32_t	worst_id;
	int			worst_i;
	/* ef <= 64 (uniform `inreg`): while a layer is searched entry i lives in lane i's registers — no LDS round trip per
	 * offer or pick — and is written to wd / wid / wx when the search ends (h2_set_spill) */
	bool		inreg;
	double		rd;
	uint32_t	rid;
	uint32_t	rx;
};

__device__ __forceinline__ double
h2_readlane_f64(double v, int l)
{
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

/* block number of entry i (uniform i) */
__device__ __forceinline__ uint32_t
h2_id_at(const H2Set &W, int i)
{
	return W.inreg ? (uint32_t) __builtin_amdgcn_readlane((int) W.rid, i) : W.wid[i];
}

__device__ __forceinline__ void
h2_set_spill(const H2Set &W, int lane)
{
	if (!W.inreg)
		return;
	if ((uint32_t) lane < W.nw)
	{
		W.wd[lane] = W.rd;
		W.wid[lane] = W.rid;
		W.wx[lane] = (uint8_t) W.rx;
	}
	__threadfence_block();
}

/*
 * The best of the lanes' (d, id, index) triples under the total order h2_less (FAR: its reverse), index < 0 = the lane has
 * none; wave-uniform result.  (d2, block) is a total order, so the pairing of the reduction is free: four steps inside the
 * rows of 16 lanes and two broadcasts across them by data-parallel-primitive moves (no LDS round trip) — the xor
 * butterfly over ds_bpermute this replaces was 24 dependent round trips per call, once per distance evaluation.
 */
template <bool FAR>
__device__ __forceinline__ int
h2_best_lane(double bd, uint32_t bid, int bi)
{
#define H2_DPP_STEP(CTRL, RMASK)                                                                                              \
	{                                                                                                                         \
		const int	lo = __builtin_amdgcn_update_dpp((int) __double2loint(bd), (int) __double2loint(bd), CTRL, RMASK, 0xF, false); \
		const int	hi = __builtin_amdgcn_update_dpp((int) __double2hiint(bd), (int) __double2hiint(bd), CTRL, RMASK, 0xF, false); \
		const uint32_t oid = (uint32_t) __builtin_amdgcn_update_dpp((int) bid, (int) bid, CTRL, RMASK, 0xF, false);             \
		const int	oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, RMASK, 0xF, false);                                         \
		const double od = __hiloint2double(hi, lo);                                                                            \
		const bool	take = oi >= 0 && (bi < 0 || (FAR ? h2_less(bd, bid, od, oid) : h2_less(od, oid, bd, bid)));               \
                                                                                                                              \
		bd = take ? od : bd;                                                                                                   \
		bid = take ? oid : bid;                                                                                                \
		bi = take ? oi : bi;                                                                                                   \
	}
	H2_DPP_STEP(0xB1, 0xF)		/* quad_perm [1, 0, 3, 2] */
	H2_DPP_STEP(0x4E, 0xF)		/* quad_perm [2, 3, 0, 1] */
	H2_DPP_STEP(0x141, 0xF)		/* row_half_mirror */
	H2_DPP_STEP(0x140, 0xF)		/* row_mirror: every lane of a row holds the row's best */
	H2_DPP_STEP(0x142, 0xA)		/* row_bcast:15 into rows 1 and 3 */
	H2_DPP_STEP(0x143, 0xC)		/* row_bcast:31 into rows 2 and 3: lane 63 holds the wave's best */
#undef H2_DPP_STEP
	return __builtin_amdgcn_readlane(bi, 63);
}

/* nearest unexpanded entry (index, or -1), wave-uniform */
__device__ __forceinline__ int
h2_pick(const H2Set &W, int lane)
{
	double		bd = 0.0;
	uint32_t	bid = 0;
	int			bi = -1;

	if (W.inreg)
	{
		/* (the lanes hold the entries in ascending order: the nearest unexpanded one is the first) */
		const unsigned long long m = __ballot((uint32_t) lane < W.nw && !W.rx);

		return m ? (int) __builtin_ctzll(m) : -1;
	}
	for (uint32_t i = lane; i < W.nw; i += 64)
		if (!W.wx[i] && (bi < 0 || h2_less(W.wd[i], W.wid[i], bd, bid)))
		{
			bd = W.wd[i];
			bid = W.wid[i];
			bi = (int) i;
		}
	return h2_best_lane<false>(bd, bid, bi);
}

/* farthest entry (index), wave-uniform; nw >= 1 */
__device__ __forceinline__ int
h2_worst(const H2Set &W, int lane)
{
	double		bd = 0.0;
	uint32_t	bid = 0;
	int			bi = -1;

	if (W.inreg)
	{
		if ((uint32_t) lane < W.nw)
		{
			bd = W.rd;
			bid = W.rid;
			bi = lane;
		}
		return h2_best_lane<true>(bd, bid, bi);
	}
	for (uint32_t i = lane; i < W.nw; i += 64)
		if (bi < 0 || h2_less(bd, bid, W.wd[i], W.wid[i]))
		{
			bd = W.wd[i];
			bid = W.wid[i];
			bi = (int) i;
		}
	return h2_best_lane<true>(bd, bid, bi);
}

/* offer (d, id) to the set: appended while there is room, else it replaces the farthest entry it beats (uniform args) */
__device__ __forceinline__ void
h2_offer(H2Set &W, double d, uint32_t id, int lane)
{
	if (W.inreg)
	{
		/* lanes 0 .. nw - 1 hold the entries in ascending (d2, block) order: the new one goes where the entries before it are
		 * the nearer ones, everything behind moves up a lane (one whole-wave shift; the farthest falls off a full set).  An
		 * arg-max over the set per accepted offer, however it is reduced, was a third of a search's instructions. */
		const int	pos = (int) __popcll(__ballot((uint32_t) lane < W.nw && h2_less(W.rd, W.rid, d, id)));

		if (W.nw >= W.ef && pos >= (int) W.ef)
			return;
		const int	slo = __builtin_amdgcn_update_dpp(0, (int) __double2loint(W.rd), 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
		const int	shi = __builtin_amdgcn_update_dpp(0, (int) __double2hiint(W.rd), 0x138, 0xF, 0xF, false);
		const int	sid = __builtin_amdgcn_update_dpp(0, (int) W.rid, 0x138, 0xF, 0xF, false);
		const int	sx = __builtin_amdgcn_update_dpp(0, (int) W.rx, 0x138, 0xF, 0xF, false);

		if (lane > pos)
		{
			W.rd = __hiloint2double(shi, slo);
			W.rid = (uint32_t) sid;
			W.rx = (uint32_t) sx;
		}
		else if (lane == pos)
		{
			W.rd = d;
			W.rid = id;
			W.rx = 0;
		}
		if (W.nw < W.ef)
			W.nw++;
		return;
	}
	if (W.nw < W.ef)
	{
		if (lane == 0)
		{
			W.wd[W.nw] = d;
			W.wid[W.nw] = id;
			W.wx[W.nw] = 0;
		}
		W.nw++;
		W.worst_i = -1;
		__threadfence_block();
		return;
	}
	if (W.worst_i < 0)
	{
		const int	w = h2_worst(W, lane);

		W.worst_i = w;
		W.worst_d = W.wd[w];
		W.worst_id = W.wid[w];
	}
	if (h2_less(d, id, W.worst_d, W.worst_id))
	{
		if (lane == 0)
		{
			W.wd[W.worst_i] = d;
			W.wid[W.worst_i] = id;
			W.wx[W.worst_i] = 0;
		}
		W.worst_i = -1;
		__threadfence_block();
	}
}

/*
 * Best-first search of one layer from the single entry point (ep, epd); leaves the result set in W (unsorted).
 * `evals` counts distance evaluations.  A node's neighbour list is read one slot per lane, the unvisited ones are
 * scored four rows at a time and offered to the set.
 */
template <int W16 = 0, int S = 1>
__device__ __forceinline__ void
h2_search_layer(const H2Graph &g, const H2Query &Q, uint32_t ep, double epd, int level, H2Set &W, H2Visited &V, int lane,
				long long &evals)
{
	W.nw = 0;
	W.worst_i = -1;
	W.inreg = W.ef <= 64;
	W.rd = 0.0;
	W.rid = 0;
	W.rx = 0;
	(void) V.mark(lane == 0, ep, lane);
	h2_offer(W, epd, ep, lane);
	H2_PH_DECL;
	uint32_t	pf_id = NDBHIP_INVALID_BLOCK, pf_e0 = NDBHIP_INVALID_BLOCK;
	int			pf_cnt = 0;

	for (;;)
	{
		const int	bi = h2_pick(W, lane);

		H2_PH(0);
		if (bi < 0)
			break;
		if (V.hv && V.over)			/* uniform */
			V.migrate(lane);
		H2_PH_COUNT(4);
		if (W.inreg)
		{
			if (lane == bi)
				W.rx = 1;
		}
		else
		{
			if (lane == 0)
				W.wx[bi] = 1;
			__threadfence_block();
		}
		const uint32_t c = h2_id_at(W, bi);
		uint32_t	e0;
		int			cnt;

		if (W.inreg && c == pf_id)		/* uniform: the list asked for during the previous expansion's rows */
		{
			e0 = pf_e0;
			cnt = pf_cnt;
		}
		else
		{
			/* (the slot is read whatever the count says — every list has 2m slots — so that both loads are in flight together) */
			const uint32_t *nb = g.nbrs + (size_t) c * g.stride + (size_t) level * 2 * g.m;

			e0 = lane < 2 * g.m ? nb[lane] : NDBHIP_INVALID_BLOCK;
			cnt = min((int) g.ncount[(size_t) c * NDBHIP_HNSW_MAX_LEVEL + level], 2 * g.m);
			/* a node page holds lists for levels 0 .. its own only (hnsw_am.c:124-181); the dense layout a reference-compatible
			 * build leaves has the out-of-item writes of Q12 / Q21 above that: not part of the index */
			if (level > 0 && g.levels[c] < level)
				cnt = 0;
		}
		const uint32_t e = lane < cnt ? e0 : NDBHIP_INVALID_BLOCK;
		const bool	fresh = V.mark(e != NDBHIP_INVALID_BLOCK && e < g.nvisible && e != 0, e, lane);
		unsigned long long todo = __ballot(fresh);

		if (W.inreg)
		{
			/* An expansion is two dependent round trips — the node's list, then its neighbours' rows.  The list of the node
			 * that is next in line NOW (the nearest unexpanded entry; c is marked) is asked for before this expansion's rows:
			 * unless one of these rows turns out nearer, the next expansion finds its list in registers.  A frozen graph: the
			 * list read now is the list read then. */
			const unsigned long long m2 = __ballot((uint32_t) lane < W.nw && !W.rx);

			pf_id = NDBHIP_INVALID_BLOCK;
			if (m2)
			{
				pf_id = (uint32_t) __builtin_amdgcn_readlane((int) W.rid, (int) __builtin_ctzll(m2));
				const uint32_t *nb2 = g.nbrs + (size_t) pf_id * g.stride + (size_t) level * 2 * g.m;

				pf_e0 = lane < 2 * g.m ? nb2[lane] : NDBHIP_INVALID_BLOCK;
				pf_cnt = min((int) g.ncount[(size_t) pf_id * NDBHIP_HNSW_MAX_LEVEL + level], 2 * g.m);
				if (level > 0 && g.levels[pf_id] < level)
					pf_cnt = 0;
			}
		}
		H2_PH(1);
		while (todo)
		{
			uint32_t	ids[H2_NR];
			double		d[H2_NR];
			int			n = 0;

#pragma unroll
			for (int u = 0; u < H2_NR; u++)
			{
				ids[u] = 0;
				if (todo)
				{
					const int	l = __builtin_ctzll(todo);

					todo &= todo - 1;
					ids[u] = (uint32_t) __builtin_amdgcn_readlane((int) e, l);
					n = u + 1;
				}
			}
			h2_ids_d2<W16, S>(g, Q, ids, n, lane, d);
			evals += n;
			H2_PH(2);
			for (int u = 0; u < n; u++)
				h2_offer(W, d[u], ids[u], lane);
			H2_PH(3);
		}
	}
	h2_set_spill(W, lane);
	H2_PH_FLUSH;
}

/* greedy step of the upper layers: from (cur, curd) move to the nearest neighbour at `level` while one is nearer */
template <int W16 = 0, int S = 1>
__device__ __forceinline__ void
h2_greedy(const H2Graph &g, const H2Query &Q, int level, uint32_t &cur, double &curd, int lane, long long &evals)
{
	for (;;)
	{
		const int	cnt0 = min((int) g.ncount[(size_t) cur * NDBHIP_HNSW_MAX_LEVEL + level], 2 * g.m);
		const int	cnt = g.levels[cur] >= level ? cnt0 : 0;	/* (lists exist up to the node's own level: see h2_search_layer) */
		const uint32_t *nb = g.nbrs + (size_t) cur * g.stride + (size_t) level * 2 * g.m;
		const uint32_t e = lane < cnt ? nb[lane] : NDBHIP_INVALID_BLOCK;
		unsigned long long todo = __ballot(e != NDBHIP_INVALID_BLOCK && e < g.nvisible && e != 0);
		uint32_t	bid = cur;
		double		bd = curd;

		while (todo)
		{
			uint32_t	ids[H2_NR];
			double		d[H2_NR];
			int			n = 0;

#pragma unroll
			for (int u = 0; u < H2_NR; u++)
			{
				ids[u] = 0;
				if (todo)
				{
					const int	l = __builtin_ctzll(todo);

					todo &= todo - 1;
					ids[u] = (uint32_t) __builtin_amdgcn_readlane((int) e, l);
					n = u + 1;
				}
			}
			h2_ids_d2<W16, S>(g, Q, ids, n, lane, d);
			evals += n;
			for (int u = 0; u < n; u++)
				if (h2_less(d[u], ids[u], bd, bid))
				{
					bd = d[u];
					bid = ids[u];
				}
		}
		if (bid == cur)
			return;
		cur = bid;
		curd = bd;
	}
}

/* the set's entries ascending by (d2, id) into sid / sd (LDS): every lane ranks its entries by counting */
__device__ __forceinline__ void
h2_sort(const H2Set &W, uint32_t *sid, double *sd, int lane)
{
	for (uint32_t i = lane; i < W.nw; i += 64)
	{
		const double d = W.wd[i];
		const uint32_t id = W.wid[i];
		uint32_t	rank = 0;

		for (uint32_t j = 0; j < W.nw; j++)
			rank += h2_less(W.wd[j], W.wid[j], d, id) ? 1u : 0u;
		sid[rank] = id;
		sd[rank] = d;
	}
	__threadfence_block();
}

/*
 * The links of one node out of candidates ascending by (d2 to the node, id): select 0 = the first M; 1 = the
 * heuristic — a candidate is taken unless it is nearer to one already taken than to the node.  out / outd in LDS or
 * global; returns how many (uniform).
 */
__device__ __forceinline__ int
h2_select(const H2Graph &g, const uint32_t *cid, const double *cd, int nc, int M, int select, uint32_t *out, double *outd,
		  int lane)
{
	int			n = 0;

	for (int i = 0; i < nc && n < M; i++)
	{
		const uint32_t c = cid[i];
		const double dc = cd[i];
		bool		ok = true;

		if ((select & 1) && n > 0)
		{
			/* the candidate against everything taken so far, H2_NR rows at a time with their loads in flight together (one
			 * row pair after the other, stopping at the first that is nearer, was up to 16 dependent round trips per
			 * candidate — most of a build).  Same arithmetic (the candidate in the query's place: h2_dist2x4 = h2_dist2_rows
			 * term for term), and whether ANY taken row is nearer does not depend on the order they are looked at. */
			H2Query		C;

			C.load(g.vecs + (size_t) c * g.dim, g.dim, lane);
			for (int j0 = 0; j0 < n && ok; j0 += H2_NR)
			{
				const float *x[H2_NR];
				double		d[H2_NR];
				const int	nn = min(H2_NR, n - j0);

#pragma unroll
				for (int u = 0; u < H2_NR; u++)
					x[u] = g.vecs + (size_t) (u < nn ? out[j0 + u] : c) * g.dim;
				h2_dist2x4(C, x, nn, lane, d);
#pragma unroll
				for (int u = 0; u < H2_NR; u++)
					if (u < nn && d[u] < dc)
						ok = false;
			}
		}
		if (ok)
		{
			if (lane == 0)
			{
				out[n] = c;
				outd[n] = dc;
			}
			n++;
			__threadfence_block();
		}
	}
	if ((select & 5) == 5)
	{
		/* keepPrunedConnections: the places left go to the nearest candidates the heuristic passed over, in order */
		const int	n0 = n;

		for (int i = 0; i < nc && n < M; i++)
		{
			const uint32_t c = cid[i];
			bool		taken = false;

			for (int j = 0; j < n0; j++)
				taken = taken || out[j] == c;		/* (uniform: every lane reads the same LDS words) */
			if (!taken)
			{
				if (lane == 0)
				{
					out[n] = c;
					outd[n] = cd[i];
				}
				n++;
				__threadfence_block();
			}
		}
	}
	return n;
}

/* walk rows of rows [first, first + n) x dim: the reference's float4_to_fp16 (src/types/quantization.c:141-168: mantissa
 * truncated, subnormal results flushed to signed zero, overflow and NaN -> infinity) */
__global__ __launch_bounds__(256) void
k_h2_walk_rows(const float *__restrict__ src, uint16_t *__restrict__ dst, size_t nel)
{
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < nel; i += (size_t) gridDim.x * 256)
	{
		const uint32_t u = __float_as_uint(src[i]);
		const uint16_t sign = (uint16_t) ((u >> 16) & 0x8000u);
		const int	e = (int) ((u >> 23) & 0xffu) - 127 + 15;

		dst[i] = e <= 0 ? sign : (e >= 31 ? (uint16_t) (sign | 0x7c00u) : (uint16_t) (sign | ((uint32_t) e << 10) | ((u & 0x7fffffu) >> 13)));
	}
}

/* rinv of rows [0, nblocks): a wave per row; the tree of ndbo_h2_dist2 / _w16 — element i into partial i mod 64 (float4 rows)
 * or (i / 4) mod 64 (walk rows), a lane's elements in increasing order, the butterfly — then 1 / sqrt, 0 for a row of zeros
 * (oracle ndbo_h2_rinv; fp64 sqrt and divide are the IEEE results on gfx950: tests/test_gpu_hnsw2.py) */
template <int W16>
__global__ __launch_bounds__(256) void
k_h2_rinv(const float *__restrict__ vecs, const uint16_t *__restrict__ vecs16, int dim, uint32_t nblocks, double *__restrict__ out)
{
	const int	lane = threadIdx.x & 63;

	for (uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nblocks; b += gridDim.x * 4u)
	{
		double		p = 0.0;

		if (W16)
		{
			const uint16_t *w = vecs16 + (size_t) b * dim;

			for (int g0 = lane; g0 * 4 < dim; g0 += 64)
#pragma unroll
				for (int t = 0; t < 4; t++)
					if (g0 * 4 + t < dim)
					{
						const double xv = (double) __half2float(__ushort_as_half(w[g0 * 4 + t]));

						p += xv * xv;
					}
		}
		else
		{
			const float *x = vecs + (size_t) b * dim;

			for (int i = lane; i < dim; i += 64)
				p += (double) x[i] * (double) x[i];
		}
		p = h2_wave_fold(p);
		if (lane == 0)
			out[b] = p > 0.0 ? 1.0 / __builtin_sqrt(p) : 0.0;
	}
}

__host__ __device__ static inline size_t
h2_smem_bytes(uint32_t ef, bool table = true /* with the LDS visited table (the search; the build's walks outgrow it at once,
											   * and its 8 KB a wave would cost the build a third of its walkers) */ )
{
	return (((size_t) ef * (8 + 4 + 1 + 8 + 4) + 64 + 15) & ~(size_t) 15) + ((table && ef <= H2_HV_MAX_EF) ? (size_t) H2_HV * 4 : 0);
}

/* kNN queries: greedy descent to level 1, layer search with ef at level 0, the k nearest ascending, distances as
 * (float) sqrt(d2).  Persistent grid of one-wave blocks; block b owns visited map b.
 * W16 (ndbo_h2_search_w16): descent and layer search on the fp16 walk rows (g.vecs16), then the result set's entries
 * scored against the float4 rows with the definition's arithmetic and ordered by that. */
template <int W16, int S>		/* W16: 0, or the walk rows' groups a lane (dim <= 256 W16); S: the operator class's strategy */
__device__ __forceinline__ void
h2_search_body(H2Graph g, const float *__restrict__ queries, uint32_t nq, uint32_t ef, uint32_t k, uint32_t entry, int entry_level,
			const uint64_t *__restrict__ tids, uint32_t *__restrict__ vbits, uint32_t *__restrict__ vlog, uint32_t nwords,
			uint32_t *__restrict__ out_blocks, float *__restrict__ out_dist, int *__restrict__ out_count,
			uint64_t *__restrict__ out_tids, long long *__restrict__ out_evals,
			uint32_t *__restrict__ next /* zero at launch: queries are dealt to the waves as they come free (a query costs
										  * 400 .. 2500 evaluations: two a wave in a fixed order left the batch waiting for the
										  * unluckiest wave) */ )
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int	lane = threadIdx.x;
	H2Set		W;
	H2Visited	V;

	W.wd = (double *) smem;
	double	   *sd = W.wd + ef;
	W.wid = (uint32_t *) (sd + ef);
	uint32_t   *sid = W.wid + ef;
	W.wx = (uint8_t *) (sid + ef);
	W.ef = ef;
	V.bits = vbits + (size_t) blockIdx.x * nwords;
	V.log = vlog + (size_t) blockIdx.x * H2_LOG_CAP;
	V.nwords = nwords;
	V.nlog = 0;
	/* the LDS visited table behind the set's arrays (h2_smem_bytes) */
	uint32_t   *const hv_lds = ef <= H2_HV_MAX_EF ? (uint32_t *) (smem + (((size_t) ef * (8 + 4 + 1 + 8 + 4) + 64 + 15) & ~(size_t) 15)) : nullptr;

	V.hv = hv_lds;
	V.nhv = 0;
	V.over = false;
	if (hv_lds)
		V.clear_lds(lane);
	for (;;)
	{
		uint32_t	q = 0;

		if (lane == 0)
			q = atomicAdd(next, 1u);
		q = (uint32_t) __builtin_amdgcn_readfirstlane((int) q);
		if (q >= nq)
			break;
		H2Query		Q;
		long long	evals = 0;
		uint32_t	n = 0;

		if (W16)
			Q.load16(queries + (size_t) q * g.dim, g.dim, lane);
		else
			Q.load(queries + (size_t) q * g.dim, g.dim, lane);
		if (entry != NDBHIP_INVALID_BLOCK)
		{
			uint32_t	cur = entry;
			double		curd;

			if (W16 || S != 1)
			{
				uint32_t	ids[H2_NR];
				double		d[H2_NR];

#pragma unroll
				for (int u = 0; u < H2_NR; u++)
					ids[u] = u == 0 ? cur : 0u;
				h2_ids_d2<W16, S>(g, Q, ids, 1, lane, d);
				curd = d[0];
			}
			else
				curd = h2_dist2(Q, g.vecs + (size_t) cur * g.dim, lane);
			evals = 1;
			for (int lc = entry_level; lc >= 1; lc--)
				h2_greedy<W16, S>(g, Q, lc, cur, curd, lane, evals);
			h2_search_layer<W16, S>(g, Q, cur, curd, 0, W, V, lane, evals);
			V.clear(lane);			/* (the LDS table, or — after a migration — the bitmap) */
			V.hv = hv_lds;
			if constexpr (S != 1)
			{
				/* strategies 2, 3: the result set under hnswComputeDistance's own arithmetic on the float4 rows (hnsw_am.c:1321-1337:
				 * sequential, fp32 products widened, fp64 sums) — a lane per entry, its row 16 bytes at a time —, kept as the
				 * float4's fp64 image (exact, same order, same ties) so that the sort below orders (that float4, block) */
				const float *qq = queries + (size_t) q * g.dim;

				for (uint32_t i = lane; i < W.nw; i += 64)
				{
					const float *x = g.vecs + (size_t) W.wid[i] * g.dim;
					Acc<S == 2 ? R_HNSW_COS : R_HNSW_IP> a;
					int			t = 0;

					if ((g.dim & 3) == 0)
						for (; t < g.dim; t += 4)
						{
							const float4 xv = *(const float4 *) (x + t);		/* (rows of dim % 4 == 0 are 16-byte aligned; the query may not be) */

							a.step(qq[t], xv.x);
							a.step(qq[t + 1], xv.y);
							a.step(qq[t + 2], xv.z);
							a.step(qq[t + 3], xv.w);
						}
					for (; t < g.dim; t++)
						a.step(qq[t], x[t]);
					W.wd[i] = (double) a.fin();
				}
				evals += W.nw;
				__threadfence_block();
			}
			else if (W16)
			{
				/* the result set against the float4 rows, H2_NR at a time (one evaluation each, counted) */
				Q.load(queries + (size_t) q * g.dim, g.dim, lane);
				for (uint32_t i0 = 0; i0 < W.nw; i0 += H2_NR)
				{
					uint32_t	ids[H2_NR];
					double		d[H2_NR];
					const int	n2 = (int) min((uint32_t) H2_NR, W.nw - i0);

#pragma unroll
					for (int u = 0; u < H2_NR; u++)
						ids[u] = u < n2 ? W.wid[i0 + u] : 0u;
					h2_ids_d2<0>(g, Q, ids, n2, lane, d);
#pragma unroll
					for (int u = 0; u < H2_NR; u++)
						if (lane == u && u < n2)
							W.wd[i0 + u] = d[u];
				}
				evals += W.nw;
				__threadfence_block();
			}
			h2_sort(W, sid, sd, lane);
			n = min(W.nw, k);
			for (uint32_t i = lane; i < n; i += 64)
			{
				out_blocks[(size_t) q * k + i] = sid[i];
				out_dist[(size_t) q * k + i] = S != 1 ? (float) sd[i] : (float) __builtin_sqrt(sd[i]);
				if (out_tids)
					out_tids[(size_t) q * k + i] = tids[sid[i]];
			}
		}
		if (lane == 0)
		{
			out_count[q] = (int) n;
			if (out_evals)
				out_evals[q] = evals;
		}
		__threadfence_block();
	}
}

#define H2_SEARCH_PARAMS H2Graph g, const float *__restrict__ queries, uint32_t nq, uint32_t ef, uint32_t k, uint32_t entry, int entry_level, \
	const uint64_t *__restrict__ tids, uint32_t *__restrict__ vbits, uint32_t *__restrict__ vlog, uint32_t nwords, \
	uint32_t *__restrict__ out_blocks, float *__restrict__ out_dist, int *__restrict__ out_count, uint64_t *__restrict__ out_tids, \
	long long *__restrict__ out_evals, uint32_t *__restrict__ next
#define H2_SEARCH_ARGS g, queries, nq, ef, k, entry, entry_level, tids, vbits, vlog, nwords, out_blocks, out_dist, out_count, out_tids, out_evals, next
template <int W16, int S = 1>
__global__ __launch_bounds__(64) void
k_h2_search(H2_SEARCH_PARAMS)
{
	h2_search_body<W16, S>(H2_SEARCH_ARGS);
}

/* the same held to 128 registers: four walkers a SIMD instead of three (a walk waits on memory most of its time; what a
 * compute unit gets done scales with the walks it holds).  Walk rows of dim <= 768 fit without scratch. */
template <int W16>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void
k_h2_search4(H2_SEARCH_PARAMS)
{
	h2_search_body<W16, 1>(H2_SEARCH_ARGS);
}

/*
 * Search phase of a batch of inserts against the graph as it stands (frozen: nothing it reads is written meanwhile).
 * Member i = block first + i, level lev[i]; its selections of levels top .. 0 (top = min(level, entry level)) go to
 * sel_ids / sel_d2 [sel_off[i] + (top - lc)][m] and sel_n, AND into its own (not yet reachable) neighbour lists.
 */
__global__ __launch_bounds__(64) void
k_h2_insert_search(H2Graph g, uint32_t first, uint32_t nmem, uint32_t efc, int select, uint32_t entry, int entry_level,
				   const uint32_t *__restrict__ sel_off, uint32_t *__restrict__ sel_ids, double *__restrict__ sel_d2,
				   int *__restrict__ sel_n, uint32_t *__restrict__ vbits, uint32_t *__restrict__ vlog, uint32_t nwords,
				   uint32_t *__restrict__ next /* zero at launch: members are dealt to the waves as they come free — a member
												 * costs 2 .. 10 ms, and with two a wave in a fixed order the batch waited
												 * for the wave that drew two expensive ones */ )
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ uint32_t s_selid[128];
	__shared__ double s_seld[128];
	const int	lane = threadIdx.x;
	H2Set		W;
	H2Visited	V;

	W.wd = (double *) smem;
	double	   *sd = W.wd + efc;
	W.wid = (uint32_t *) (sd + efc);
	uint32_t   *sid = W.wid + efc;
	W.wx = (uint8_t *) (sid + efc);
	W.ef = efc;
	V.bits = vbits + (size_t) blockIdx.x * nwords;
	V.log = vlog + (size_t) blockIdx.x * H2_LOG_CAP;
	V.nwords = nwords;
	V.nlog = 0;
	V.hv = nullptr;			/* (ef_construction walks meet thousands of nodes: the bitmap from the start) */
	V.nhv = 0;
	V.over = false;
	for (;;)
	{
		uint32_t	i = 0;

		if (lane == 0)
			i = atomicAdd(next, 1u);
		i = (uint32_t) __builtin_amdgcn_readfirstlane((int) i);
		if (i >= nmem)
			break;
		const uint32_t x = first + i;
		const int	level = g.levels[x];
		H2Query		Q;
		long long	evals = 0;

		if (entry == NDBHIP_INVALID_BLOCK)
			continue;
		Q.load(g.vecs + (size_t) x * g.dim, g.dim, lane);
		uint32_t	cur = entry;
		double		curd = h2_dist2(Q, g.vecs + (size_t) cur * g.dim, lane);

		for (int lc = entry_level; lc > level; lc--)
			h2_greedy(g, Q, lc, cur, curd, lane, evals);
		const int	top = min(level, entry_level);

		for (int lc = top; lc >= 0; lc--)
		{
			const size_t so = ((size_t) sel_off[i] + (size_t) (top - lc)) * 2 * g.m;	/* rows of 2m: level 0 may take that many */

#ifdef NDB_PHASES
			const unsigned long long ph_t0 = wall_clock64();
#endif
			h2_search_layer(g, Q, cur, curd, lc, W, V, lane, evals);
			V.clear(lane);
#ifdef NDB_PHASES
			const unsigned long long ph_t1 = wall_clock64();
#endif
			h2_sort(W, sid, sd, lane);
#ifdef NDB_PHASES
			const unsigned long long ph_t2 = wall_clock64();
#endif
			const int	n = h2_select(g, sid, sd, (int) W.nw, (lc == 0 && (select & 2)) ? 2 * g.m : g.m, select & 5, s_selid, s_seld, lane);
#ifdef NDB_PHASES
			if (blockIdx.x == 0 && lane == 0)
			{
				/* [5] layer search + clearing, [6] sort, [7] selection: block 0's wave, all its members and levels */
				atomicAdd(&g_h2_phases[5], ph_t1 - ph_t0);
				atomicAdd(&g_h2_phases[6], ph_t2 - ph_t1);
				atomicAdd(&g_h2_phases[7], wall_clock64() - ph_t2);
			}
#endif

			if (lane == 0)
			{
				sel_n[sel_off[i] + (uint32_t) (top - lc)] = n;
				g.ncount[(size_t) x * NDBHIP_HNSW_MAX_LEVEL + lc] = (int16_t) n;
			}
			for (int j = lane; j < n; j += 64)
			{
				sel_ids[so + j] = s_selid[j];
				sel_d2[so + j] = s_seld[j];
				g.nbrs[(size_t) x * g.stride + (size_t) lc * 2 * g.m + j] = s_selid[j];
			}
			cur = sid[0];
			curd = sd[0];
			__threadfence_block();
		}
	}
}

/* one target (node, level) and its back-link requests req[r0 .. r1), in insertion order */
struct H2Group
{
	uint32_t	node;
	int			level;
	uint32_t	r0, r1;
};
struct H2Req
{
	uint32_t	x;
	uint32_t	pad;
	double		d2;				/* d2(x, target) */
};

/*
 * Apply phase: every target replays its requests in order — appended while the list has room, else the list becomes
 * what the selection rule keeps of (list + x) around the target, ascending by (d2 to the target, id).  One wave per
 * target; scratch in LDS: cid / cd / kid / kd [2m + 1].
 */
__global__ __launch_bounds__(64) void
k_h2_apply(H2Graph g, const H2Group *__restrict__ groups, uint32_t ngroups_host, const uint32_t *__restrict__ ngroups_dev,
		   const H2Req *__restrict__ req, int select)
{
	__shared__ uint32_t cid[260], kid[260];
	__shared__ double cd[260], kd[260];
	const int	lane = threadIdx.x;
	const uint32_t ngroups = ngroups_dev ? *ngroups_dev : ngroups_host;		/* (grouped on the device: the count lives there) */

	for (uint32_t gi = blockIdx.x; gi < ngroups; gi += gridDim.x)
	{
		const H2Group gr = groups[gi];
		const int	cap = gr.level == 0 ? 2 * g.m : g.m;
		uint32_t   *nb = g.nbrs + (size_t) gr.node * g.stride + (size_t) gr.level * 2 * g.m;
		int16_t    *pc = &g.ncount[(size_t) gr.node * NDBHIP_HNSW_MAX_LEVEL + gr.level];
		int			cnt = *pc;
		const float *ev = g.vecs + (size_t) gr.node * g.dim;
		H2Query		C;
		bool		Cloaded = false;

		C.q = ev;
		C.dim = g.dim;
		for (uint32_t r = gr.r0; r < gr.r1; r++)
		{
			const uint32_t x = req[r].x;
			const double dxe = req[r].d2;

			if (cnt < cap)
			{
				if (lane == 0)
					nb[cnt] = x;
				cnt++;
				__threadfence_block();
				continue;
			}
			/* candidates = the list + x with their distances to the target: the list's rows H2_NR at a time, their loads in
			 * flight together (one row pair after the other was cnt dependent round trips per request — most of this
			 * kernel); the target in the query's place: h2_dist2x4 = h2_dist2_rows term for term */
			if (!Cloaded)
			{
				C.load(ev, g.dim, lane);
				Cloaded = true;
			}
			for (int i0 = 0; i0 < cnt; i0 += H2_NR)
			{
				uint32_t	ids[H2_NR];
				double		d[H2_NR];
				const int	n2 = min(H2_NR, cnt - i0);

#pragma unroll
				for (int u = 0; u < H2_NR; u++)
					ids[u] = u < n2 ? nb[i0 + u] : 0u;
				h2_ids_d2<0>(g, C, ids, n2, lane, d);
#pragma unroll
				for (int u = 0; u < H2_NR; u++)
					if (lane == u && u < n2)
					{
						kid[i0 + u] = ids[u];
						kd[i0 + u] = d[u];
					}
			}
			if (lane == 0)
			{
				kid[cnt] = x;
				kd[cnt] = dxe;
			}
			__threadfence_block();
			/* ... ascending by (d2, id) ... */
			for (int i0 = lane; i0 <= cnt; i0 += 64)
			{
				const double d = kd[i0];
				const uint32_t id = kid[i0];
				int			rank = 0;

				for (int j = 0; j <= cnt; j++)
					rank += h2_less(kd[j], kid[j], d, id) ? 1 : 0;
				cid[rank] = id;
				cd[rank] = d;
			}
			__threadfence_block();
			/* ... and what the rule keeps of them */
			const int	n = h2_select(g, cid, cd, cnt + 1, cap, select & 5, kid, kd, lane);

			for (int j = lane; j < 2 * g.m; j += 64)
				nb[j] = j < n ? kid[j] : NDBHIP_INVALID_BLOCK;
			cnt = n;
			__threadfence_block();
		}
		if (lane == 0)
			*pc = (int16_t) cnt;
	}
}

/*
 * A batch's back-links grouped by target ON THE DEVICE (round 5; the host did this between the search and the apply
 * kernel of every batch — selections down, 140 k keys built and radix-sorted, groups and requests up — with the device
 * waiting: a third of a build).  The selections of a batch lie as rows [off[i] + (top - lc)][m] in insertion order —
 * member i ascending, level descending, slot ascending — so the item index row * m + j IS the insertion order:
 *   k_h2_bl_rows    row -> (member, level)
 *   k_h2_bl_keys    key[idx] = target << 8 | level (a slot beyond the row's count: `pad`, which sorts behind every key)
 *   a STABLE radix sort of (key, idx) (rocprim::radix_sort_pairs) = by (target, insertion order): the host's order
 *   k_h2_bl_heads   flags the first request of every target; an exclusive sum numbers the groups
 *   k_h2_bl_groups  requests (x, d2) in sorted order, groups (target, level, r0, r1), the counts k_h2_apply reads
 * Nothing comes back to the host: a build is one queue of launches.
 */
__global__ __launch_bounds__(256) void
k_h2_bl_rows(uint32_t b, const uint32_t *__restrict__ off, const int *__restrict__ levels, int entry_level,
			 uint32_t *__restrict__ rowmem, int *__restrict__ rowlc)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;

	if (i >= b)
		return;
	const int	top = min(levels[i], entry_level);

	for (int t = 0; t <= top; t++)
	{
		rowmem[off[i] + t] = i;
		rowlc[off[i] + t] = top - t;
	}
}

__global__ __launch_bounds__(256) void
k_h2_bl_keys(uint32_t nitems, uint32_t m, const int *__restrict__ sn, const uint32_t *__restrict__ sid, const int *__restrict__ rowlc,
			 unsigned long long pad, unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals)
{
	const uint32_t idx = blockIdx.x * 256u + threadIdx.x;

	if (idx >= nitems)
		return;
	const uint32_t row = idx / m, j = idx - row * m;

	keys[idx] = (int) j < sn[row] ? (((unsigned long long) sid[idx] << 8) | (unsigned long long) rowlc[row]) : pad;
	vals[idx] = idx;
}

__global__ __launch_bounds__(256) void
k_h2_bl_heads(uint32_t nitems, const unsigned long long *__restrict__ keys, unsigned long long pad, uint32_t *__restrict__ flags)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;

	if (r < nitems)
		flags[r] = (keys[r] != pad && (r == 0 || keys[r] != keys[r - 1])) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void
k_h2_bl_groups(uint32_t nitems, uint32_t m, uint32_t first, const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals,
			   unsigned long long pad, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ gpos,
			   const uint32_t *__restrict__ rowmem, const double *__restrict__ sd2, H2Group *__restrict__ grp, H2Req *__restrict__ req,
			   uint32_t *__restrict__ counts /* [0] groups, [1] requests of this batch (zeroed before) */ ,
			   unsigned long long *__restrict__ total)
{
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;

	if (r >= nitems)
		return;
	const unsigned long long key = keys[r];

	if (key == pad)
		return;
	const uint32_t idx = vals[r];
	const uint32_t gi = gpos[r] + flags[r] - 1u;
	H2Req		rq;

	rq.x = first + rowmem[idx / m];
	rq.pad = 0;
	rq.d2 = sd2[idx];
	req[r] = rq;
	if (flags[r])
	{
		grp[gi].node = (uint32_t) (key >> 8);
		grp[gi].level = (int) (key & 0xFFull);
		grp[gi].r0 = r;
	}
	const unsigned long long nextkey = r + 1 < nitems ? keys[r + 1] : pad;

	if (nextkey != key)
		grp[gi].r1 = r + 1;
	if (nextkey == pad)
	{
		counts[0] = gi + 1;
		counts[1] = r + 1;
		atomicAdd(total, (unsigned long long) (r + 1));
	}
}

#endif							/* NDBHIP_HNSW2_H */
