/*
 * ndbhip_build.h — ivfbuild / ivfinsert on the device (part of ndbhip.hip's translation unit): k-means with the
 * reference's rules (src/index/ivf_am.c:2070-2294), the insert-time assignment (:905-935), list packing,
 * ndbhip_ivf_build_device / _assign_device / export and the GPU plugin's k-means launchers.
 */
#ifndef NDBHIP_BUILD_H
#define NDBHIP_BUILD_H

/* ================================================================== */
/* IVF build: k-means (ivf_am.c:2070-2294), insert-time assignment     */
/* (:905-935), list packing                                            */
/* ================================================================== */
#define NDB_CGROUP 64			/* centroids handled by one wave */

/*
 * Nearest-centroid search for 64 rows x one group of NDB_CGROUP centroids.
 * The row tile is staged ONCE per 64-float chunk and every centroid of the
 * group is accumulated against it: acc[c] lives in LDS ([c][lane], conflict
 * free), the centroid chunk arrives through the scalar cache.  Each
 * (row, centroid) sum is still the reference's sequential fp32 chain
 * (vector_distance_l2 / the accum loop of ivfinsert).
 * SQRT = false: compare squared sums (find_nearest_centroid, :2274-2294)
 * SQRT = true : compare sqrtf(sum)     (ivfinsert, :915-934)
 * grid = (ceil(nrows/64), ngroups), block = 64.
 */
template <bool SQRT>
__global__ __launch_bounds__(64) void
k_assign_partial(const float *__restrict__ rows, uint32_t nrows, int dim,
				 const float *__restrict__ cents, int ncent,
				 float *__restrict__ part_dist, int *__restrict__ part_idx)
{
	__shared__ __attribute__((aligned(16))) float tile[NDB_TILE_FLOATS];
	__shared__ float accs[NDB_CGROUP * 64];
	const int	lane = threadIdx.x;
	const int	grp = lane >> 4;
	const int	slot = lane & 15;
	const uint32_t r = blockIdx.x * 64 + lane;
	const uint32_t row = (r < nrows) ? r : (nrows - 1);
	const int	c0 = blockIdx.y * NDB_CGROUP;
	const int	gc = (ncent - c0 < NDB_CGROUP) ? (ncent - c0) : NDB_CGROUP;
	uint32_t	rows16[16];

#pragma unroll
	for (int i = 0; i < 16; i++)
		rows16[i] = __shfl(row, 4 * i + grp, 64);
	for (int cl = 0; cl < gc; cl++)
		accs[cl * 64 + lane] = 0.0f;

	for (int c = 0; c < dim; c += NDB_CHUNK)
	{
		const bool	full = (dim - c) >= NDB_CHUNK;
		const int	npieces = full ? 16 : ((dim - c) >> 2);
		float4		x[16];

		if (full)
			stage_chunk<true>(x, rows, rows16, dim, c, tile, lane, grp, slot);
		else
			stage_chunk<false>(x, rows, rows16, dim, c, tile, lane, grp, slot);

		for (int cl = 0; cl < gc; cl++)
		{
			const float *__restrict__ q = cents + (size_t) (c0 + cl) * (size_t) dim + c;
			float		a = accs[cl * 64 + lane];

			if (full)
			{
#pragma unroll
				for (int p = 0; p < 16; p++)
				{
					const float4 qq = *reinterpret_cast<const float4 *>(q + p * 4);
					float		d;

					d = x[p].x - qq.x; a = a + d * d;
					d = x[p].y - qq.y; a = a + d * d;
					d = x[p].z - qq.z; a = a + d * d;
					d = x[p].w - qq.w; a = a + d * d;
				}
			}
			else
			{
#pragma unroll
				for (int p = 0; p < 16; p++)
					if (p < npieces)
					{
						const float4 qq = *reinterpret_cast<const float4 *>(q + p * 4);
						float		d;

						d = x[p].x - qq.x; a = a + d * d;
						d = x[p].y - qq.y; a = a + d * d;
						d = x[p].z - qq.z; a = a + d * d;
						d = x[p].w - qq.w; a = a + d * d;
					}
			}
			accs[cl * 64 + lane] = a;
		}
	}
	float		best = FLT_MAX;
	int			bidx = -1;

	for (int cl = 0; cl < gc; cl++)
	{
		float		d = accs[cl * 64 + lane];

		if (SQRT)
			d = __builtin_sqrtf(d);
		if (d < best)
		{
			best = d;
			bidx = c0 + cl;
		}
	}
	if (r < nrows)
	{
		part_dist[(size_t) blockIdx.y * nrows + r] = best;
		part_idx[(size_t) blockIdx.y * nrows + r] = bidx;
	}
}

/* dim % 4 != 0: one lane walks its row against every centroid of the group directly */
template <bool SQRT>
__global__ __launch_bounds__(64) void
k_assign_partial_direct(const float *__restrict__ rows, uint32_t nrows, int dim,
						const float *__restrict__ cents, int ncent,
						float *__restrict__ part_dist, int *__restrict__ part_idx)
{
	const uint32_t r = blockIdx.x * 64 + threadIdx.x;
	const int	c0 = blockIdx.y * NDB_CGROUP;
	const int	gc = (ncent - c0 < NDB_CGROUP) ? (ncent - c0) : NDB_CGROUP;

	if (r >= nrows)
		return;
	float		best = FLT_MAX;
	int			bidx = -1;

	for (int cl = 0; cl < gc; cl++)
	{
		const float *q = cents + (size_t) (c0 + cl) * dim;
		const float *x = rows + (size_t) r * dim;
		float		a = 0.0f;

		for (int i = 0; i < dim; i++)
		{
			const float d = x[i] - q[i];

			a = a + d * d;
		}
		if (SQRT)
			a = __builtin_sqrtf(a);
		if (a < best)
		{
			best = a;
			bidx = c0 + cl;
		}
	}
	part_dist[(size_t) blockIdx.y * nrows + r] = best;
	part_idx[(size_t) blockIdx.y * nrows + r] = bidx;
}

/* cblock[g][d][j] = cents[16 g + j][d] (j beyond the last centroid repeats centroid 0 and is ignored) */
__global__ void
k_interleave16(const float *__restrict__ cents, int ncent, int dim, float *__restrict__ cblock)
{
	const int	g16 = blockIdx.y;
	const int	d = blockIdx.x * blockDim.x + threadIdx.x;

	if (d >= dim)
		return;
	float		v[NDB_QG];

#pragma unroll
	for (int j = 0; j < NDB_QG; j++)
	{
		const int	c = g16 * NDB_QG + j;

		v[j] = cents[(size_t) (c < ncent ? c : 0) * dim + d];
	}
	float4	   *dst = reinterpret_cast<float4 *>(cblock + ((size_t) g16 * dim + d) * NDB_QG);

#pragma unroll
	for (int j = 0; j < NDB_QG / 4; j++)
		dst[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

/*
 * Nearest centroid, fast form (dim % 64 == 0): one wave = 64 rows x 16 centroids, the row chunk
 * staged once per chunk, the 16 centroids' values of a dimension arriving as ONE scalar load and
 * the arithmetic running on centroid pairs (v_pk_*_f32) — the same engine as k_ivf_scan_grouped.
 * Every (row, centroid) sum is still the sequential fp32 chain of vector_distance_l2 / ivfinsert
 * ((x-c)^2 == (c-x)^2 exactly).  grid = (row tiles, centroid groups of 16), block = 64.
 */
template <bool SQRT, int CH>
__global__ __launch_bounds__(64, (CH == 32 ? NDB_G32_WAVES : NDB_GROUPED_WAVES_PER_SIMD)) void
k_assign_grouped(const float *__restrict__ rows, uint32_t nrows, int dim, const float *__restrict__ cblock,
				 int ncent, float *__restrict__ part_dist, int *__restrict__ part_idx,
				 float *__restrict__ all_dist, uint32_t all_stride)
{
	__shared__ __attribute__((aligned(16))) float tile[64 * CH];
	const int	lane = threadIdx.x;
	/*
	 * 1-D grid, XCD-aware: block b runs on XCD b % 8 (observed; speed only).  XCD x takes the row tiles
	 * congruent to x mod 8 and walks each one through ALL centroid groups before the next, so a tile's
	 * rows come from HBM once and from that XCD's L2 for the other groups.
	 */
	const uint32_t ngroups = ((uint32_t) ncent + NDB_QG - 1) / NDB_QG;
	const uint32_t seq = blockIdx.x >> 3;
	const uint32_t tileno = (seq / ngroups) * 8u + (blockIdx.x & 7u);
	const uint32_t cgrp = seq % ngroups;

	if (tileno * 64u >= nrows)
		return;
	const uint32_t r = tileno * 64 + lane;
	const uint32_t row = (r < nrows) ? r : (nrows - 1);
	const int	c0 = (int) cgrp * NDB_QG;
	const int	gc = (ncent - c0 < NDB_QG) ? (ncent - c0) : NDB_QG;
	uint32_t	rowsN[CH / 4];
	GAcc<R_IVF_L2> acc;
	const float *qs = cblock + (size_t) cgrp * (size_t) dim * NDB_QG;
	ndb_f16		qa0, qa1, qb0, qb1;

	acc.init();
	rows_for_loads<CH>(rowsN, row, lane);
	sload2x16(qa0, qa1, qs);
	for (int c = 0; c < dim; c += CH)
	{
		float4		x[CH / 4];

		stage_chunk_w<CH>(x, rows, rowsN, dim, c, tile, lane);
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
	swait2(qa0, qa1);
	if (all_dist)
	{
		/* every distance, not the nearest: the query x centroid scan of ivfSelectClusters (rows = queries) */
		if (r < nrows)
		{
#pragma unroll
			for (int j = 0; j < NDB_QG; j++)
			{
				float		d = (j & 1) ? acc.s[j >> 1].y : acc.s[j >> 1].x;

				if (SQRT)
					d = __builtin_sqrtf(d);
				if (j < gc)
					all_dist[(size_t) r * all_stride + c0 + j] = d;
			}
		}
		return;
	}
	float		best = FLT_MAX;
	int			bidx = -1;

#pragma unroll
	for (int j = 0; j < NDB_QG; j++)
	{
		float		d = (j & 1) ? acc.s[j >> 1].y : acc.s[j >> 1].x;

		if (SQRT)
			d = __builtin_sqrtf(d);
		if (j < gc && d < best)
		{
			best = d;
			bidx = c0 + j;
		}
	}
	if (r < nrows)
	{
		part_dist[(size_t) cgrp * nrows + r] = best;
		part_idx[(size_t) cgrp * nrows + r] = bidx;
	}
}

/* first strict minimum over the groups, in centroid order; none below FLT_MAX -> 0
 * (best = 0 / min_idx = 0 initialisers: ivf_am.c:2277, 812) */
__global__ void
k_assign_combine(const float *__restrict__ part_dist, const int *__restrict__ part_idx, int ngroups,
				 uint32_t nrows, int *__restrict__ out_list, int *__restrict__ counts)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;

	if (r >= nrows)
		return;
	float		best = FLT_MAX;
	int			bidx = 0;

	for (int g2 = 0; g2 < ngroups; g2++)
	{
		const float d = part_dist[(size_t) g2 * nrows + r];
		const int	i = part_idx[(size_t) g2 * nrows + r];

		if (i >= 0 && d < best)
		{
			best = d;
			bidx = i;
		}
	}
	out_list[r] = bidx;
	if (counts)
		atomicAdd(&counts[bidx], 1);
}

/* kmeans_update_centroids (:2182-2213): block = centroid.  The members are first compacted IN SAMPLE
 * ORDER into LDS (ballot + popcount prefix), then thread = coordinate adds them in that order and
 * divides by (float) count — the reference's summation order, without scanning all n samples per
 * coordinate.  Dynamic LDS: n uint32. */
__global__ __launch_bounds__(256) void
k_kmeans_update(const float *__restrict__ data, int n, int dim, const int *__restrict__ assign,
				const int *__restrict__ counts, float *__restrict__ cents)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	uint32_t   *members = (uint32_t *) smem_raw;
	uint32_t   *sh = members + n;		/* 8 words */
	const int	c = blockIdx.x;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	uint32_t	base = 0;

	for (int start = 0; start < n; start += 256)
	{
		const int	i = start + (int) tid;
		const bool	mine = i < n && assign[i] == c;
		const unsigned long long m = __ballot(mine);
		const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

		if (lane == 0)
			sh[wave] = __popcll(m);
		__syncthreads();
		uint32_t	woff = 0, tot = 0;

		for (uint32_t w = 0; w < 4; w++)
		{
			if (w < wave)
				woff += sh[w];
			tot += sh[w];
		}
		if (mine)
			members[base + woff + __popcll(m & below)] = (uint32_t) i;
		base += tot;
		__syncthreads();
	}
	const uint32_t cnt = base;		/* == counts[c] */

	for (int j = tid; j < dim; j += 256)
	{
		float		s = 0.0f;

		for (uint32_t k2 = 0; k2 < cnt; k2++)
			s = s + data[(size_t) members[k2] * dim + j];
		if (counts[c] > 0)
			s = s / (float) counts[c];
		cents[(size_t) c * dim + j] = s;
	}
}

/* per-sample squared distance to its own centroid (:2225-2230) */
__global__ void
k_kmeans_point_cost(const float *__restrict__ data, int n, int dim, const int *__restrict__ assign,
					const float *__restrict__ cents, float *__restrict__ pc)
{
	const int	i = blockIdx.x * blockDim.x + threadIdx.x;

	if (i >= n)
		return;
	const float *x = data + (size_t) i * dim;
	const float *q = cents + (size_t) assign[i] * dim;

	if ((dim & 3) == 0)
	{
		/* (round 6: the same sequential chain with the loads 16 bytes at a time and 32 of them in flight — a lane that asks
		 * for one float of its own row per step had the kernel at 186 us for 10000 x 768) */
		pc[i] = scr_exact<R_IVF_L2SQ>(x, q, dim);
		return;
	}
	float		s = 0.0f;

	for (int j = 0; j < dim; j++)
	{
		const float d = x[j] - q[j];

		s = s + d * d;
	}
	pc[i] = s;
}

/* cost += d_i strictly in sample order, in fp32 (:2221-2232): the block stages the terms in LDS,
 * one lane then adds them in order (the sum is order-dependent and decides the stopping iteration) */
__global__ __launch_bounds__(256) void
k_seq_sum(const float *__restrict__ pc, int n, float *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	float	   *v = (float *) smem_raw;

	for (int i = threadIdx.x; i < n; i += 256)
		v[i] = pc[i];
	__syncthreads();
	if (threadIdx.x == 0)
	{
		float		s = 0.0f;

		for (int i = 0; i < n; i++)
			s = s + v[i];
		*out = s;
	}
}

/* kmeans_init (:2092-2104): first k samples, zeros beyond n */
__global__ void
k_kmeans_init(const float *__restrict__ data, int n, int dim, int k, float *__restrict__ cents)
{
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;

	if (i >= (size_t) k * dim)
		return;
	const int	c = (int) (i / dim);

	cents[i] = (c < n) ? data[i] : 0.0f;
}


static int
set_kernel_attributes_build()
{
	HIP_TRY(hipFuncSetAttribute((const void *) k_kmeans_update, hipFuncAttributeMaxDynamicSharedMemorySize,
								NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_seq_sum, hipFuncAttributeMaxDynamicSharedMemorySize,
								NDB_TOPK_MAX_SMEM));
	return set_kernel_attributes_hnsw();
}

/*
 * hipFree waits for every stream of the device — including the one a table is still arriving on (ndbhip_ivf_build):
 * one small free after the k-means would hold the build up until the last byte of the upload.  While
 * g_defer_frees is set, the build's temporaries are parked here and released together at its end.
 */
static bool g_defer_frees = false;
static std::vector<void *> g_deferred;

static hipError_t
free_or_defer(void *p)
{
	if (!p)
		return hipSuccess;
	if (g_defer_frees)
	{
		g_deferred.push_back(p);
		return hipSuccess;
	}
	return hipFree(p);
}

static void
flush_deferred(void)
{
	for (void *p : g_deferred)
		(void) hipFree(p);
	g_deferred.clear();
}

/* scratch of assign_rows, reusable across calls (the k-means loop calls it once per iteration) */
struct AssignWs
{
	float	   *pd = nullptr;
	int		   *pi = nullptr;
	float	   *cblock = nullptr;
	size_t		pn = 0, cn = 0;
	int release()
	{
		if (pd) HIP_TRY(free_or_defer(pd));
		if (pi) HIP_TRY(free_or_defer(pi));
		if (cblock) HIP_TRY(free_or_defer(cblock));
		pd = nullptr; pi = nullptr; cblock = nullptr; pn = cn = 0;
		return 0;
	}
};

static int
assign_rows(const float *d_rows, int64_t nrows, int dim, const float *d_cents, int ncent, bool use_sqrt,
			int *d_out_list, int *d_counts, AssignWs *ws = nullptr)
{
	AssignWs	local;

	if (!ws)
		ws = &local;
	const bool	fast = (dim % NDB_CHUNK) == 0;
	const int	gsize = fast ? NDB_QG : NDB_CGROUP;
	const int	ngroups = (ncent + gsize - 1) / gsize;
	const int64_t chunk = 1 << 18;
	const int64_t cmax = std::min<int64_t>(chunk, nrows);

	if (nrows <= 0)
		return 0;
	if (ws->pn < (size_t) ngroups * cmax)
	{
		if (ws->pd) HIP_TRY(hipFree(ws->pd));
		if (ws->pi) HIP_TRY(hipFree(ws->pi));
		ws->pd = nullptr; ws->pi = nullptr;
		HIP_TRY(hipMalloc((void **) &ws->pd, (size_t) ngroups * cmax * sizeof(float)));
		HIP_TRY(hipMalloc((void **) &ws->pi, (size_t) ngroups * cmax * sizeof(int)));
		ws->pn = (size_t) ngroups * cmax;
	}
	if (fast && ws->cn < (size_t) ngroups * dim * NDB_QG)
	{
		if (ws->cblock) HIP_TRY(hipFree(ws->cblock));
		ws->cblock = nullptr;
		HIP_TRY(hipMalloc((void **) &ws->cblock, (size_t) ngroups * dim * NDB_QG * sizeof(float)));
		ws->cn = (size_t) ngroups * dim * NDB_QG;
	}
	float	   *pd = ws->pd, *cblock = ws->cblock;
	int		   *pi = ws->pi;

	if (fast)
	{
		hipLaunchKernelGGL(k_interleave16, dim3((dim + 255) / 256, ngroups), dim3(256), 0, g.stream, d_cents,
						   ncent, dim, cblock);
	}
	for (int64_t r0 = 0; r0 < nrows; r0 += chunk)
	{
		const uint32_t n = (uint32_t) std::min<int64_t>(chunk, nrows - r0);
		dim3		grid((n + 63) / 64, ngroups);
		const float *rows = d_rows + (size_t) r0 * dim;

		if (fast)
		{
			/* 1-D, XCD-aware: ceil(tiles / 8) * 8 tiles x ngroups blocks (k_assign_grouped decodes it) */
			const dim3	g1((unsigned) ((((size_t) (n + 63) / 64 + 7) / 8) * 8 * (size_t) ngroups));

			if (use_sqrt)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assign_grouped<true, 32>), g1, dim3(64), 0, g.stream, rows, n, dim,
								   (const float *) cblock, ncent, pd, pi);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assign_grouped<false, 32>), g1, dim3(64), 0, g.stream, rows, n, dim,
								   (const float *) cblock, ncent, pd, pi);
		}
		else if ((dim & 3) == 0)
		{
			if (use_sqrt)
				hipLaunchKernelGGL(k_assign_partial<true>, grid, dim3(64), 0, g.stream, rows, n, dim, d_cents,
								   ncent, pd, pi);
			else
				hipLaunchKernelGGL(k_assign_partial<false>, grid, dim3(64), 0, g.stream, rows, n, dim, d_cents,
								   ncent, pd, pi);
		}
		else
		{
			if (use_sqrt)
				hipLaunchKernelGGL(k_assign_partial_direct<true>, grid, dim3(64), 0, g.stream, rows, n, dim,
								   d_cents, ncent, pd, pi);
			else
				hipLaunchKernelGGL(k_assign_partial_direct<false>, grid, dim3(64), 0, g.stream, rows, n, dim,
								   d_cents, ncent, pd, pi);
		}
		hipLaunchKernelGGL(k_assign_combine, dim3((n + 255) / 256), dim3(256), 0, g.stream, (const float *) pd,
						   (const int *) pi, ngroups, n, d_out_list + r0, d_counts);
	}
	HIP_TRY(hipGetLastError());
	if (ws == &local)
	{
		HIP_TRY(hipStreamSynchronize(g.stream));
		return local.release();
	}
	return 0;
}

static int	assign_rows_s16(const float *d_rows, int64_t nrows, int dim, const float *d_cents, int k, int *d_out_list,
							unsigned long long *stats, const std::function<int()> &while_running,
							const std::function<int(int64_t)> &rows_until = nullptr, bool use_sqrt = true);

extern "C" int
ndbhip_ivf_assign_device(const float *d_centroids, int ncentroids, int dim, const float *d_rows,
						 int64_t nrows, int *d_out_list)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!d_centroids || ncentroids < 1 || dim < 1 || nrows < 0 || (nrows > 0 && (!d_rows || !d_out_list)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (nrows > 0xFFFFFFFFll)
		return fail(NDBHIP_ERR_UNSUPPORTED, "too many rows");
	if (g_build_s16 && nrows >= 4096)
	{
		const int	rc = assign_rows_s16(d_rows, nrows, dim, d_centroids, ncentroids, d_out_list, nullptr, nullptr);

		if (rc != 1)
			return rc;
	}
	return assign_rows(d_rows, nrows, dim, d_centroids, ncentroids, true, d_out_list, nullptr);
}

/* ivfinsert (src/index/ivf_am.c:797-1167) for one host row: nearest centroid by the insert-time rule
 * (sqrtf of the fp32 sum, strict <, first minimum: :905-935) on the device, then the entry goes to the tail
 * of that list (ndbhip_ivf_append). */
extern "C" int
ndbhip_ivf_insert(ndbhip_ivf *ix, const float *vec, const uint8_t *tid6, int *list_out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_insert");
	if (!ix || !vec || !tid6)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!ix->loaded || ix->ncent < 1)
		return fail(NDBHIP_ERR_STATE, "index has no centroids/lists loaded");
	if (ix->sharded)
		return fail(NDBHIP_ERR_UNSUPPORTED, "insert into the unsharded mirror");
	const int	ncmp = std::min(ix->nlists, ix->ncent);	/* i < nlist && i < maxoff: :917 */
	int			list = 0;

	if (grow(ix->w_q, ix->w_q_n, (size_t) ix->dim)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_ocnt, ix->w_ocnt_n, (size_t) 1)) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemcpyAsync(ix->w_q, vec, (size_t) ix->dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	int			rc = assign_rows(ix->w_q, 1, ix->dim, ix->d_centroids, ncmp, true, ix->w_ocnt, nullptr);

	if (rc)
		return rc;
	HIP_TRY(hipMemcpyAsync(&list, ix->w_ocnt, sizeof(int), hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	if (list_out)
		*list_out = list;
	return ndbhip_ivf_append(ix, list, vec, tid6);
}

__global__ void k_count_members(const int *__restrict__ idx, int n, int k, int *__restrict__ counts);

extern "C" int
ndbhip_kmeans_device(const float *d_samples, int n, int dim, int k, int max_iter, float threshold,
					 float *d_centroids, int *d_assign, int *d_counts, int *out_iters, float *out_cost)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!d_samples || !d_centroids || !d_assign || !d_counts || n < 1 || dim < 1 || k < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if ((size_t) n * 4 + 64 > NDB_TOPK_MAX_SMEM)	/* the reference samples at most 10000 rows (ivf_am.c:580) */
		return fail(NDBHIP_ERR_UNSUPPORTED, "k-means sample of %d rows exceeds the LDS-resident limit", n);
	float	   *d_pc = nullptr;
	/* the iteration's cost lands in pinned host memory straight from the kernel: no copy engine in the loop */
	float	   *d_cost = g.pin_words;
	float		prevCost = FLT_MAX, cost = 0.0f;
	int			iters = 0;
	AssignWs	ws;

	const auto	tk0 = std::chrono::steady_clock::now();

	HIP_TRY(hipMalloc((void **) &d_pc, (size_t) n * sizeof(float)));
	if (g_debug_build & 2)
		fprintf(stderr, "k-means: scratch allocated at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk0).count());
	hipLaunchKernelGGL(k_kmeans_init, dim3((unsigned) (((size_t) k * dim + 255) / 256)), dim3(256), 0, g.stream,
					   d_samples, n, dim, k, d_centroids);
	for (int iter = 0; iter < max_iter; iter++)
	{
		int			rc;

		HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t) k * sizeof(int), g.stream));
		/* round 6: the Lloyd assignment through the matrix-core screen too (kmeans_assign's squared distances decide among
		 * the centroids inside the bound's window): 0.43 -> 0.2 ms an iteration at 10000 x 1024 x 768 */
		rc = (g_build_s16 && g_kmeans_s16 && n >= 4096) ? assign_rows_s16(d_samples, n, dim, d_centroids, k, d_assign, nullptr, nullptr, nullptr, false) : 1;
		if (rc == 0)
			hipLaunchKernelGGL(k_count_members, dim3((n + 255) / 256), dim3(256), 0, g.stream, (const int *) d_assign, n, k, d_counts);
		else if (rc == 1)
			rc = assign_rows(d_samples, n, dim, d_centroids, k, false, d_assign, d_counts, &ws);
		if (rc)
			return rc;
		hipLaunchKernelGGL(k_kmeans_update, dim3(k), dim3(256), (size_t) n * 4 + 64, g.stream, d_samples, n, dim,
						   (const int *) d_assign, (const int *) d_counts, d_centroids);
		hipLaunchKernelGGL(k_kmeans_point_cost, dim3((n + 255) / 256), dim3(256), 0, g.stream, d_samples, n, dim,
						   (const int *) d_assign, (const float *) d_centroids, d_pc);
		hipLaunchKernelGGL(k_seq_sum, dim3(1), dim3(256), (size_t) n * 4, g.stream, (const float *) d_pc, n, d_cost);
		HIP_TRY(hipStreamSynchronize(g.stream));
		cost = *(volatile float *) d_cost;
		if (g_debug_build & 2)
			fprintf(stderr, "k-means: iteration %d done at %.2f ms\n", iter, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk0).count());
		iters = iter + 1;
		/* fabs(prevCost - cost) < threshold, float difference widened (ivf_am.c:2141) */
		if (fabs((double) (float) (prevCost - cost)) < (double) threshold)
			break;
		prevCost = cost;
	}
	HIP_TRY(free_or_defer(d_pc));
	if (ws.release())
		return NDBHIP_ERR_HIP;
	if (out_iters)
		*out_iters = iters;
	if (out_cost)
		*out_cost = cost;
	return NDBHIP_OK;
}

/* One Lloyd half-step each, from host memory: kmeans_assign (ivf_am.c:2157-2180: first minimum of the fp32
 * squared L2 over the k centroids) and kmeans_update_centroids (:2182-2213: members added in sample order,
 * divided by (float) count; an empty cluster keeps its centroid).  The shapes of the GPU vtable's
 * launch_kmeans_assign / launch_kmeans_update (include/neurondb_gpu_backend.h:66-79). */
extern "C" int
ndbhip_kmeans_assign(const float *X, const float *C, int *idx, int n, int dim, int k)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!X || !C || !idx || n < 1 || dim < 1 || dim > 32767 || k < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	float	   *d_x = nullptr, *d_c = nullptr;
	int		   *d_i = nullptr;

	HIP_TRY(hipMalloc((void **) &d_x, (size_t) n * dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_c, (size_t) k * dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_i, (size_t) n * sizeof(int)));
	HIP_TRY(hipMemcpyAsync(d_x, X, (size_t) n * dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_c, C, (size_t) k * dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	int			rc = assign_rows(d_x, n, dim, d_c, k, false, d_i, nullptr);

	if (!rc)
	{
		HIP_TRY(hipMemcpyAsync(idx, d_i, (size_t) n * sizeof(int), hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
	}
	HIP_TRY(hipFree(d_x));
	HIP_TRY(hipFree(d_c));
	HIP_TRY(hipFree(d_i));
	return rc;
}

__global__ void
k_count_members(const int *__restrict__ idx, int n, int k, int *__restrict__ counts)
{
	const int	i = blockIdx.x * blockDim.x + threadIdx.x;

	if (i < n && idx[i] >= 0 && idx[i] < k)
		atomicAdd(&counts[idx[i]], 1);
}

extern "C" int
ndbhip_kmeans_update(const float *X, const int *idx, float *C, int n, int dim, int k)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!X || !C || !idx || n < 1 || dim < 1 || dim > 32767 || k < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if ((size_t) n * 4 + 64 > NDB_TOPK_MAX_SMEM)
		return fail(NDBHIP_ERR_UNSUPPORTED, "k-means update of %d rows exceeds the LDS-resident member list", n);
	float	   *d_x = nullptr, *d_c = nullptr;
	int		   *d_i = nullptr, *d_n = nullptr;

	HIP_TRY(hipMalloc((void **) &d_x, (size_t) n * dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_c, (size_t) k * dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_i, (size_t) n * sizeof(int)));
	HIP_TRY(hipMalloc((void **) &d_n, (size_t) k * sizeof(int)));
	HIP_TRY(hipMemcpyAsync(d_x, X, (size_t) n * dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_c, C, (size_t) k * dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_i, idx, (size_t) n * sizeof(int), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemsetAsync(d_n, 0, (size_t) k * sizeof(int), g.stream));
	hipLaunchKernelGGL(k_count_members, dim3((n + 255) / 256), dim3(256), 0, g.stream, (const int *) d_i, n, k, d_n);
	hipLaunchKernelGGL(k_kmeans_update, dim3(k), dim3(256), (size_t) n * 4 + 64, g.stream, (const float *) d_x, n, dim,
					   (const int *) d_i, (const int *) d_n, d_c);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(C, d_c, (size_t) k * dim * sizeof(float), hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipFree(d_x));
	HIP_TRY(hipFree(d_c));
	HIP_TRY(hipFree(d_i));
	HIP_TRY(hipFree(d_n));
	return NDBHIP_OK;
}

/* ---- list packing: stable counting sort of rows by list id (heap order kept inside a list) ---- */

#define NDB_PACK_BLOCK 256

/* per-block histogram: hist[list * nblocks + block] */
__global__ __launch_bounds__(NDB_PACK_BLOCK) void
k_pack_hist(const int *__restrict__ lists, int64_t nrows, int nlists, uint32_t nblocks,
			uint32_t *__restrict__ hist)
{
	const int64_t r = (int64_t) blockIdx.x * NDB_PACK_BLOCK + threadIdx.x;

	if (r < nrows)
		atomicAdd(&hist[(size_t) lists[r] * nblocks + blockIdx.x], 1u);
}

/* exclusive scan of hist[list][block] in (list-major, block) order, on the device:
 * pass A: one thread per list adds up its blocks -> list_len; pass B (single thread): list bases;
 * pass C: one thread per list walks its blocks again writing the running offsets */
__global__ void
k_pack_list_totals(const uint32_t *__restrict__ hist, int nlists, uint32_t nblocks, int64_t *__restrict__ list_len)
{
	const int	L = blockIdx.x * blockDim.x + threadIdx.x;

	if (L >= nlists)
		return;
	int64_t		t = 0;

	for (uint32_t b = 0; b < nblocks; b++)
		t += hist[(size_t) L * nblocks + b];
	list_len[L] = t;
}

__global__ void
k_pack_offsets(const uint32_t *__restrict__ hist, int nlists, uint32_t nblocks,
			   const int64_t *__restrict__ list_len, int64_t *__restrict__ scanned)
{
	__shared__ int64_t base_sh;
	const int	L = blockIdx.x;

	if (threadIdx.x == 0)
	{
		int64_t		b0 = 0;

		for (int l2 = 0; l2 < L; l2++)
			b0 += list_len[l2];
		base_sh = b0;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		int64_t		acc = base_sh;

		for (uint32_t b = 0; b < nblocks; b++)
		{
			scanned[(size_t) L * nblocks + b] = acc;
			acc += hist[(size_t) L * nblocks + b];
		}
	}
}

/* dest = scanned[list][block] + rank of the row among earlier same-list rows of its block; copies the row */
__global__ __launch_bounds__(NDB_PACK_BLOCK) void
k_pack_scatter(const int *__restrict__ lists, int64_t nrows, int dim, uint32_t nblocks,
			   const int64_t *__restrict__ scanned, const float *__restrict__ rows,
			   const uint64_t *__restrict__ tids, float *__restrict__ out_rows, uint64_t *__restrict__ out_tids)
{
	__shared__ int sl[NDB_PACK_BLOCK];
	__shared__ int64_t sdest[NDB_PACK_BLOCK];
	const int	t = threadIdx.x;
	const int64_t r0 = (int64_t) blockIdx.x * NDB_PACK_BLOCK;
	const int64_t r = r0 + t;
	const int	L = (r < nrows) ? lists[r] : -1;

	sl[t] = L;
	__syncthreads();
	if (r < nrows)
	{
		int			rank = 0;

		for (int u = 0; u < t; u++)
			rank += (sl[u] == L);
		sdest[t] = scanned[(size_t) L * nblocks + blockIdx.x] + rank;
		out_tids[sdest[t]] = tids[r];
	}
	__syncthreads();
	const int	nb = (int) ((nrows - r0 < NDB_PACK_BLOCK) ? (nrows - r0) : NDB_PACK_BLOCK);

	if ((dim & 3) == 0)
	{
		const int	d4 = dim >> 2;

		for (int rr = 0; rr < nb; rr++)
		{
			const float4 *src = reinterpret_cast<const float4 *>(rows + (size_t) (r0 + rr) * dim);
			float4	   *dst = reinterpret_cast<float4 *>(out_rows + (size_t) sdest[rr] * dim);

			for (int j = t; j < d4; j += NDB_PACK_BLOCK)
				dst[j] = src[j];
		}
	}
	else
	{
		for (int rr = 0; rr < nb; rr++)
			for (int j = t; j < dim; j += NDB_PACK_BLOCK)
				out_rows[(size_t) sdest[rr] * dim + j] = rows[(size_t) (r0 + rr) * dim + j];
	}
}

__global__ void __launch_bounds__(256)
k_rows_gather(const float *__restrict__ rows, int dim, const int64_t *__restrict__ which, float *__restrict__ out)
{
	const float *src = rows + (size_t) which[blockIdx.x] * dim;
	float	   *dst = out + (size_t) blockIdx.x * dim;

	for (int j = threadIdx.x; j < dim; j += 256)
		dst[j] = src[j];
}

__global__ void
k_list_scatter(const int *__restrict__ lists, const int64_t *__restrict__ which, uint32_t n, int *__restrict__ out_list)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;

	if (i < n)
		out_list[which[i]] = lists[i];
}

/* 64-bit content hash of every centroid (wave per centroid; a sum of mixed words, so lane order does not matter) */
__global__ void __launch_bounds__(256)
k_cent_hash(const float *__restrict__ cents, int k, int dim, unsigned long long *__restrict__ hash)
{
	const int	lane = threadIdx.x & 63;
	const int	c = blockIdx.x * 4 + (threadIdx.x >> 6);

	if (c >= k)
		return;
	const uint32_t *x = reinterpret_cast<const uint32_t *>(cents) + (size_t) c * dim;
	unsigned long long h = 0;

	for (int i = lane; i < dim; i += 64)
	{
		unsigned long long z = ((unsigned long long) x[i] << 32 | (uint32_t) i) + 0x9E3779B97F4A7C15ull;

		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		h += z ^ (z >> 31);
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		const uint32_t lo = __shfl_xor((uint32_t) h, off, 64);
		const uint32_t hi = __shfl_xor((uint32_t) (h >> 32), off, 64);

		h += ((unsigned long long) hi << 32) | lo;
	}
	if (lane == 0)
		hash[c] = h;
}

/* dup[c] = 1 when a centroid with a smaller id holds the same bits.  Such a centroid can never be chosen: its
 * distance to any row is the earlier one's, and the insert rule's test is a strict < (ivf_am.c:905-935). */
__global__ void __launch_bounds__(256)
k_cent_dups(const float *__restrict__ cents, int k, int dim, const unsigned long long *__restrict__ hash,
			unsigned char *__restrict__ dup)
{
	const int	c = blockIdx.x * 256 + threadIdx.x;

	if (c >= k)
		return;
	const unsigned long long h = hash[c];
	unsigned char d = 0;

	for (int j = 0; j < c && !d; j++)
		if (hash[j] == h)
		{
			const uint32_t *a = reinterpret_cast<const uint32_t *>(cents) + (size_t) c * dim;
			const uint32_t *b = reinterpret_cast<const uint32_t *>(cents) + (size_t) j * dim;
			int			i = 0;

			while (i < dim && a[i] == b[i])
				i++;
			d = i == dim;
		}
	dup[c] = d;
}

/* words of the sweep's tables for one slab shape: blk_off[2] | own_len[1] | cnt[1] | pair_off[2] | item_off[2] |
 * runs[9] | loc_cand_off[k][2] */
#define NDB_ASG_META(k) ((size_t) 17 + 2 * (size_t) (k))

/* One wave: the centroids that are not duplicates, in id order, become the sweep's "(query, probe) pairs"; the
 * tables of the full slab (meta_a) and of the last, shorter one (meta_b) */
__global__ void __launch_bounds__(64)
k_assign_tables(const unsigned char *__restrict__ dup, int k, uint32_t rows_a, uint32_t rows_b,
				PairRec *__restrict__ pairs, uint32_t *__restrict__ meta_a, uint32_t *__restrict__ meta_b)
{
	const int	lane = threadIdx.x;
	uint32_t	total = 0;

	for (int c0 = 0; c0 < k; c0 += 64)
	{
		const int	c = c0 + lane;
		const bool	keep = c < k && dup[c] == 0;
		const unsigned long long m = __ballot(keep);

		if (keep)
		{
			PairRec		pr;

			pr.q = (uint32_t) c;
			pr.p = 0;
			pairs[total + (uint32_t) __popcll(m & ((1ull << lane) - 1ull))] = pr;
		}
		total += (uint32_t) __popcll(m);
	}
	const uint32_t nqt = (total + S16_QT - 1) / S16_QT;

	for (int v = 0; v < 2; v++)
	{
		uint32_t   *m32 = v ? meta_b : meta_a;
		const uint32_t n = v ? rows_b : rows_a;
		const uint32_t nitems = ((n + 127u) / 128u) * nqt;

		if (lane == 0)
		{
			m32[0] = 0; m32[1] = (n + 31u) / 32u;
			m32[2] = n;
			m32[3] = total;
			m32[4] = 0; m32[5] = total;
			m32[6] = 0; m32[7] = nitems;
		}
		if (lane <= 8)
			m32[8 + lane] = (uint32_t) (((unsigned long long) nitems * (uint32_t) lane) >> 3);
		for (int c = lane; c < k; c += 64)
		{
			m32[17 + 2 * c] = 0;
			m32[17 + 2 * c + 1] = n;
		}
	}
}

/*
 * The assignment of every row to its nearest centroid (the insert-time rule, ivf_am.c:905-935) through the fp16
 * matrix-core sweep: a slab of rows is one "list", the distinct centroids are the "queries".  Sweep 1
 * (k_s16_sweep MODE 1) leaves every row's smallest bound a_min; sweep 2 (MODE 2) records the centroids whose a
 * lies within the error bound of it, i.e. every centroid that can be the nearest in the reference's arithmetic;
 * k_s16_assign_resolve decides by that arithmetic where more than one is left.  Rows with more candidates than
 * record slots go through assign_rows.  Same list ids as assign_rows (tests/test_gpu_build.py compares both with
 * the oracle).  Returns 1 when the shape is outside what the sweep handles (the caller then runs assign_rows).
 * Temporaries are sized by the slab (2^20 rows), not by the table.
 */
static int
assign_rows_s16(const float *d_rows, int64_t nrows, int dim, const float *d_cents, int k, int *d_out_list,
				unsigned long long *stats /* host [2] or NULL: rows decided among several candidates, rows sent to the exact assignment */ ,
				const std::function<int()> &while_running /* host work done while the kernels run */ ,
				const std::function<int(int64_t)> &rows_until /* nullable: returns once rows [0, n) are on the device (a table still arriving) */ ,
				bool use_sqrt /* true: the insert rule; false (round 6): kmeans_assign's squared distances — the Lloyd iterations */ )
{
	const int	dimp = (dim + 63) & ~63;
	const uint32_t qrowbytes = (uint32_t) dimp * 4u;
	const uint32_t nqt = (uint32_t) ((k + S16_QT - 1) / S16_QT);
	/* rows per slab: about 512 MiB of fp16 planes, whatever the table's size (so that the scratch a small build
	 * leaves behind fits the next large one) */
	const int64_t slab_cap = std::min<int64_t>((int64_t) 1 << 20,
											   std::max<int64_t>(4096, (((int64_t) 512 << 20) / ((int64_t) dimp * 4)) & ~(int64_t) 127));
	const int64_t slab = std::min<int64_t>(nrows, slab_cap);
	const int64_t tail = nrows % slab == 0 ? slab : nrows % slab;
	const uint64_t nb = (uint64_t) ((slab_cap + 31) / 32);
	const uint32_t nrt = (uint32_t) ((slab + 127) / 128), nrt_cap = (uint32_t) ((slab_cap + 127) / 128);

	if (nrows < 1 || k < 1 || k > 65535 || (size_t) k * qrowbytes >= ((size_t) 1 << 32) ||
		(size_t) nrt_cap * nqt > 0x7FFFFFFFull)
		return 1;
	DevGuard	tmp;
	unsigned char *planes = nullptr, *qplanes = nullptr, *dup = nullptr;
	float	   *rn2 = nullptr, *qn2 = nullptr;
	int16_t    *rexp = nullptr;
	int		   *qexp = nullptr;
	float2	   *aux = nullptr;
	uint32_t   *xmax = nullptr, *rowmin = nullptr, *meta = nullptr;
	int64_t    *locoff = nullptr;
	unsigned int *acnt = nullptr, *heads = nullptr, *over_n = nullptr;
	unsigned long long *hash = nullptr;
	int64_t    *over_rows = nullptr;
	uint2	   *arec = nullptr;
	PairRec    *pairs = nullptr;
	S16Desc    *desc = nullptr;
	const size_t blk_bytes = (size_t) (dimp / S16_CH) * 4096;
	const uint32_t nitems = nrt * nqt;		/* upper bound: duplicates only shrink it */
	const uint32_t nitems_cap = nrt_cap * nqt;
	const size_t mw = NDB_ASG_META(k);
	/* everything but the list of overflowing rows comes out of the arena kept in g: laid out twice, first to size it */
	size_t		used = 0;
	auto		take = [&](auto *&p, size_t bytes, bool commit) {
		if (commit)
			p = reinterpret_cast<std::remove_reference_t<decltype(p)>>(g.asg_arena + used);
		used += (bytes + 255) & ~(size_t) 255;
	};
	auto		layout = [&](bool commit) {
		used = 0;
		take(planes, (size_t) (nb + 8) * blk_bytes, commit);
		take(qplanes, (size_t) k * qrowbytes, commit);
		take(rn2, (size_t) slab_cap * 4, commit);
		take(rexp, (size_t) slab_cap * 2, commit);
		take(qn2, (size_t) k * 4, commit);
		take(qexp, (size_t) k * 4, commit);
		take(aux, (size_t) k * sizeof(float2), commit);	/* the sweep's per-query slot; only [0].x is used */
		take(xmax, 4, commit);
		take(rowmin, (size_t) slab_cap * 4, commit);
		take(acnt, (size_t) slab_cap * 4, commit);
		take(arec, (size_t) slab_cap * S16_ASSIGN_SLOTS * sizeof(uint2), commit);
		take(over_n, 16, commit);
		take(pairs, (size_t) k * sizeof(PairRec), commit);
		take(desc, (size_t) 2 * nitems_cap * sizeof(S16Desc), commit);
		take(locoff, 4 * sizeof(int64_t), commit);
		take(meta, 2 * mw * 4, commit);
		take(heads, 2 * 8 * NDB_QHEAD_STRIDE * 4, commit);
		take(hash, (size_t) k * 8, commit);
		take(dup, (size_t) k, commit);
	};

	layout(false);
	if (used > g.asg_arena_cap)
	{
		HIP_TRY(hipStreamSynchronize(g.stream));
		if (g.asg_arena)
			HIP_TRY(hipFree(g.asg_arena));
		g.asg_arena = nullptr;
		g.asg_arena_cap = 0;
		HIP_TRY(hipMalloc((void **) &g.asg_arena, used));
		g.asg_arena_cap = used;
	}
	layout(true);
	if (tmp.alloc(over_rows, (size_t) nrows * sizeof(int64_t))) return NDBHIP_ERR_HIP;
	const int64_t hloc[4] = {0, slab, 0, tail};

	HIP_TRY(hipMemcpyAsync(locoff, hloc, sizeof(hloc), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemsetAsync(over_n, 0, 16, g.stream));
	/* (the arena holds an earlier build's planes: rows past a slab's end are never looked at, zeroed all the same) */
	HIP_TRY(hipMemsetAsync(planes, 0, (size_t) ((slab + 31) / 32 + 8) * blk_bytes, g.stream));
	HIP_TRY(hipMemsetAsync(xmax, 0, 4, g.stream));
	HIP_TRY(hipMemsetAsync(heads, 0, 2 * 8 * NDB_QHEAD_STRIDE * 4, g.stream));
	/* the centroids: duplicates dropped, split into fp16 planes, the sweep's tables for both slab shapes */
	hipLaunchKernelGGL(k_cent_hash, dim3((k + 3) / 4), dim3(256), 0, g.stream, d_cents, k, dim, hash);
	hipLaunchKernelGGL(k_cent_dups, dim3((k + 255) / 256), dim3(256), 0, g.stream, d_cents, k, dim,
					   (const unsigned long long *) hash, dup);
	hipLaunchKernelGGL(k_assign_tables, dim3(1), dim3(64), 0, g.stream, (const unsigned char *) dup, k, (uint32_t) slab,
					   (uint32_t) tail, pairs, meta, meta + mw);
	hipLaunchKernelGGL(k_s16_qprep, dim3((k + 3) / 4), dim3(256), 0, g.stream, d_cents, (uint32_t) k, dim, dimp,
					   (ndb_h2 *) qplanes, qn2, qexp);
	/* aux[0].x = the largest finite centroid norm (bits order like values for non-negative floats) */
	HIP_TRY(hipMemsetAsync(aux, 0, (size_t) k * sizeof(float2), g.stream));
	hipLaunchKernelGGL(k_max_nonneg, dim3(4), dim3(256), 0, g.stream, (const float *) qn2, (int64_t) k, (uint32_t *) aux);
	for (int v = 0; v < 2; v++)
		hipLaunchKernelGGL(k_s16_items, dim3((nitems + 255) / 256), dim3(256), 0, g.stream, meta + v * mw + 6, meta + v * mw + 3,
						   meta + v * mw + 2, 1, 128u, nitems, desc + (size_t) v * nitems, heads + 8 * NDB_QHEAD_STRIDE - 1);
	HIP_TRY(hipGetLastError());

	bool		overlapped = false;

	for (int64_t s0 = 0; s0 < nrows; s0 += slab)
	{
		const int64_t ns = std::min<int64_t>(slab, nrows - s0);
		const int	v = ns == slab ? 0 : 1;
		const uint32_t *m32 = meta + v * mw;
		const float *srows = d_rows + (size_t) s0 * dim;
		IvfDev		dv = {};

		dv.vecs = srows;
		dv.loc_off = locoff + 2 * v;
		dv.own_len = m32 + 2;
		dv.glob_len = m32 + 2;
		dv.dim = dim;
		dv.ncent = 1;
		dv.nlists = 1;
		if (rows_until)
		{
			const int	urc = rows_until(s0 + ns);

			if (urc != 0)
			{
				(void) hipStreamSynchronize(g.stream);
				return urc;
			}
		}
		if (s0 > 0)
			HIP_TRY(hipMemsetAsync(heads, 0, 2 * 8 * NDB_QHEAD_STRIDE * 4, g.stream));
		HIP_TRY(hipMemsetAsync(rowmin, 0xFF, (size_t) ns * 4, g.stream));
		HIP_TRY(hipMemsetAsync(acnt, 0, (size_t) ns * 4, g.stream));
		hipLaunchKernelGGL(k_s16_row_prep<0>, dim3((unsigned) ((ns + 3) / 4)), dim3(256), 0, g.stream, (const void *) srows,
						   ns, dim, dimp, (const int64_t *) (locoff + 2 * v), m32, 1, planes, rn2, rexp, xmax);
		if (g_build_single_sweep)
		{
			/* round 6, ONE sweep (MODE 4): an item tests its elements against the row's minimum so far and the resolve kernel
			 * applies the test again with the final one — the candidate sets of the two sweeps below, the matrix multiplied once */
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_L2, 0, 4, 2, 0, 4>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, dv,
							   (const unsigned char *) planes, m32, (const float *) rn2, (const int16_t *) rexp,
							   (const unsigned char *) qplanes, qrowbytes, (const float *) qn2, (const int *) qexp,
							   (float2 *) aux, m32 + 17, 1, m32 + 3, m32 + 4, (const S16Desc *) (desc + (size_t) v * nitems),
							   (const PairRec *) pairs, heads, m32 + 8, acnt, arec, 1u, rowmin, 0, dimp / S16_CH, nitems, 0u);
			if (use_sqrt)
				hipLaunchKernelGGL(k_s16_assign_resolve<true>, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, g.stream, srows, ns, dim,
								   d_cents, (const unsigned int *) acnt, (const uint2 *) arec, d_out_list + s0, s0, over_n, over_rows,
								   (unsigned long long *) (over_n + 2), (const uint32_t *) rowmin, (const float *) rn2, (const float2 *) aux);
			else
				hipLaunchKernelGGL(k_s16_assign_resolve<false>, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, g.stream, srows, ns, dim,
								   d_cents, (const unsigned int *) acnt, (const uint2 *) arec, d_out_list + s0, s0, over_n, over_rows,
								   (unsigned long long *) (over_n + 2), (const uint32_t *) rowmin, (const float *) rn2, (const float2 *) aux);
		}
		else
		{
		/* two sweeps over the same items: the minimum of every row, then the centroids within reach of it */
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_L2, 0, 4, 2, 0, 1>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, dv,
						   (const unsigned char *) planes, m32, (const float *) rn2, (const int16_t *) rexp,
						   (const unsigned char *) qplanes, qrowbytes, (const float *) qn2, (const int *) qexp,
						   (float2 *) aux, m32 + 17, 1, m32 + 3, m32 + 4, (const S16Desc *) (desc + (size_t) v * nitems),
						   (const PairRec *) pairs, heads, m32 + 8, acnt, arec, 1u, rowmin, 0, dimp / S16_CH, nitems, 0u);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_L2, 0, 4, 2, 0, 2>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, dv,
						   (const unsigned char *) planes, m32, (const float *) rn2, (const int16_t *) rexp,
						   (const unsigned char *) qplanes, qrowbytes, (const float *) qn2, (const int *) qexp,
						   (float2 *) aux, m32 + 17, 1, m32 + 3, m32 + 4, (const S16Desc *) (desc + (size_t) v * nitems),
						   (const PairRec *) pairs, heads + 8 * NDB_QHEAD_STRIDE, m32 + 8, acnt, arec, 1u, rowmin, 0,
						   dimp / S16_CH, nitems, 0u);
		if (use_sqrt)
			hipLaunchKernelGGL(k_s16_assign_resolve<true>, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, g.stream, srows, ns, dim,
							   d_cents, (const unsigned int *) acnt, (const uint2 *) arec, d_out_list + s0, s0, over_n, over_rows,
							   (unsigned long long *) (over_n + 2), (const uint32_t *) nullptr, (const float *) nullptr, (const float2 *) nullptr);
		else
			hipLaunchKernelGGL(k_s16_assign_resolve<false>, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, g.stream, srows, ns, dim,
							   d_cents, (const unsigned int *) acnt, (const uint2 *) arec, d_out_list + s0, s0, over_n, over_rows,
							   (unsigned long long *) (over_n + 2), (const uint32_t *) nullptr, (const float *) nullptr, (const float2 *) nullptr);
		}
		HIP_TRY(hipGetLastError());
		if (!overlapped && while_running)
		{
			const int	wrc = while_running();

			overlapped = true;
			if (wrc != 0)
			{
				(void) hipStreamSynchronize(g.stream);
				return wrc;
			}
		}
	}
	struct { unsigned int over, none; unsigned long long multi; } hs;
	float	   *orows = nullptr;		/* (function scope: DevGuard keeps their addresses) */
	int		   *olist = nullptr;

	HIP_TRY(hipMemcpyAsync(&hs, over_n, 16, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));	/* the temporaries go out of scope */
	if (g_debug_build)
		fprintf(stderr, "build: assign_rows_s16: %lld rows, %llu decided among several candidates, %u sent to the exact assignment (%u with no record)\n",
				(long long) nrows, hs.multi, hs.over, hs.none);
	if (stats)
	{
		stats[0] = hs.multi;
		stats[1] = hs.over;
	}
	if (hs.over > 0)
	{
		/* rows the bound could not narrow to S16_ASSIGN_SLOTS centroids: the exact assignment on a packed copy */
		if (tmp.alloc(orows, (size_t) hs.over * dim * sizeof(float))) return NDBHIP_ERR_HIP;
		if (tmp.alloc(olist, (size_t) hs.over * sizeof(int))) return NDBHIP_ERR_HIP;
		hipLaunchKernelGGL(k_rows_gather, dim3(hs.over), dim3(256), 0, g.stream, d_rows, dim, (const int64_t *) over_rows, orows);
		const int	rc = assign_rows(orows, (int64_t) hs.over, dim, d_cents, k, use_sqrt, olist, nullptr);

		if (rc != 0)
			return rc;
		hipLaunchKernelGGL(k_list_scatter, dim3((hs.over + 255) / 256), dim3(256), 0, g.stream, (const int *) olist,
						   (const int64_t *) over_rows, hs.over, d_out_list);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(g.stream));
	}
	return 0;
}

static int	ivf_build_rows(ndbhip_ivf *ix, const float *d_rows, const uint64_t *d_tids, int64_t nrows, int max_iter,
						   int *out_iters, const std::function<int(int64_t)> &rows_until, const std::function<int()> &tids_ready);

/*
 * ivfbuild (src/index/ivf_am.c:501-745) for host rows in heap order.  The table is the input of a CREATE INDEX:
 * it arrives over PCIe, and the arrival is most of the wall time (3 GB at 1M x 768 against ~25 ms of device
 * work).  So: the k-means sample (the first min(10000, 100 lists) rows, ivf_am.c:580) goes first, the rest
 * follows on the upload lanes (ndbhip.hip: upload_rows) WHILE the k-means runs, and only the assignment of every
 * row waits for the last byte.
 */
extern "C" int
ndbhip_ivf_build(ndbhip_ivf *ix, const float *rows, const uint8_t *tids6, int64_t nrows, int max_iter, int *out_iters)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_build");
	if (!ix || !rows || !tids6 || nrows < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	float	   *d_rows = nullptr;
	uint64_t   *d_tids = nullptr;
	const size_t row_bytes = (size_t) ix->dim * sizeof(float);
	const int64_t ns = std::min<int64_t>(std::min<int64_t>(10000, (int64_t) ix->nlists * 100), nrows);
	std::vector<uint64_t> t64;
	UploadJob	rest;
	int			rc;

	const auto	tb0 = std::chrono::steady_clock::now();
	auto		since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(); };
	const bool	tl = (g_debug_build & 2) != 0;

	if (big_alloc((void **) &d_rows, (size_t) nrows * row_bytes)) return NDBHIP_ERR_HIP;
	if (big_alloc((void **) &d_tids, (size_t) nrows * sizeof(uint64_t)))
	{
		big_free(d_rows);
		return NDBHIP_ERR_HIP;
	}
	/* the rows in heap order on a thread of their own; the build asks for as many as its next step reads: the
	 * sample for the k-means, then the assignment slab by slab — the device works on what has arrived while the
	 * rest is on the wire; the TIDs (packed on this thread meanwhile) follow the last row */
	if (tl) fprintf(stderr, "host build: %.2f ms device blocks allocated\n", since());
	g_defer_frees = true;
	rest.start(d_rows, rows, (size_t) nrows * row_bytes);
	rc = ivf_build_rows(ix, d_rows, d_tids, nrows, max_iter, out_iters,
						[&](int64_t upto) -> int {
							if (tl) fprintf(stderr, "host build: %.2f ms waiting for rows up to %lld\n", since(), (long long) upto);
							const int	r = rest.wait_for((size_t) upto * row_bytes);

							if (tl) fprintf(stderr, "host build: %.2f ms   ... arrived\n", since());
							return r;
						},
						[&]() -> int {
							if (tl) fprintf(stderr, "host build: %.2f ms assignment launched; packing TIDs\n", since());
							t64.resize((size_t) nrows);
							for (int64_t i = 0; i < nrows; i++)
								t64[(size_t) i] = ndb_tid_pack(tids6 + (size_t) i * 6);
							int			r = rest.wait();

							if (tl) fprintf(stderr, "host build: %.2f ms TIDs packed\n", since());
							if (r == 0)
								r = upload_rows(d_tids, t64.data(), (size_t) nrows * sizeof(uint64_t));
							if (tl) fprintf(stderr, "host build: %.2f ms TIDs uploaded\n", since());
							return r;
						});
	(void) ns;
	(void) rest.wait();
	g_defer_frees = false;
	flush_deferred();
	if (tl) fprintf(stderr, "host build: %.2f ms index built\n", since());
	(void) hipStreamSynchronize(g.stream);
	big_free(d_rows);
	big_free(d_tids);
	return rc;
}

extern "C" int
ndbhip_ivf_build_device(ndbhip_ivf *ix, const float *d_rows, const uint64_t *d_tids, int64_t nrows,
						int max_iter, int *out_iters)
{
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_build_device");
	return ivf_build_rows(ix, d_rows, d_tids, nrows, max_iter, out_iters, nullptr, nullptr);
}

static int	pack_by_list(const int *d_list, int64_t nrows, int dim, int k, const float *d_rows, const uint64_t *d_tids,
						 float *d_prow, uint64_t *d_ptid, std::vector<int64_t> &list_len);

/* rows_until(n) (nullable): returns once rows [0, n) are on the device; tids_ready() (nullable): once all TIDs are —
 * for a table that is still arriving (ndbhip_ivf_build) */
static int
ivf_build_rows(ndbhip_ivf *ix, const float *d_rows, const uint64_t *d_tids, int64_t nrows, int max_iter, int *out_iters,
			   const std::function<int(int64_t)> &rows_until, const std::function<int()> &tids_ready)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !d_rows || !d_tids || nrows < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (nrows > 0xFFFFFFFFll)
		return fail(NDBHIP_ERR_UNSUPPORTED, "too many rows");
	const int	dim = ix->dim;
	const int	k = ix->nlists;
	/* maxSamples = Min(10000, nlists * 100): the FIRST rows in heap order (ivf_am.c:580, 486-495) */
	const int	ns = (int) std::min<int64_t>(std::min<int64_t>(10000, (int64_t) k * 100), nrows);

	if (ns < k)					/* ivf_am.c:596-601 */
		return fail(NDBHIP_ERR_INVALID, "ivf: not enough sample vectors (%d < %d)", ns, k);

	float	   *d_cent = nullptr;
	int		   *d_sasg = nullptr, *d_scnt = nullptr, *d_list = nullptr;
	int			iters = 0, rc;
	const bool	dbg = (g_debug_build & 1) != 0;
	auto		now = [&]() { if (dbg) (void) hipStreamSynchronize(g.stream); return std::chrono::steady_clock::now(); };
	auto		t_start = now();
	auto		lap = [&](const char *what) {
		if (!dbg) return;
		auto		t = now();
		fprintf(stderr, "build: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_start).count());
		t_start = t;
	};

	HIP_TRY(hipMalloc((void **) &d_cent, (size_t) k * dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_sasg, (size_t) ns * sizeof(int)));
	HIP_TRY(hipMalloc((void **) &d_scnt, (size_t) k * sizeof(int)));
	if (rows_until && (rc = rows_until(ns)) != 0)
		return rc;
	rc = ndbhip_kmeans_device(d_rows, ns, dim, k, max_iter, 0.001f, d_cent, d_sasg, d_scnt, &iters, nullptr);
	if (rc)
		return rc;
	lap("k-means on the sample");
	HIP_TRY(free_or_defer(d_sasg));
	HIP_TRY(free_or_defer(d_scnt));

	/* every row goes to the list ivfinsert would choose (Q5: the reference leaves this to later INSERTs) */
	HIP_TRY(hipMalloc((void **) &d_list, (size_t) nrows * sizeof(int)));
	lap("free + malloc list ids");
	AssignWs	aws;				/* own workspace: assign_rows then returns without waiting for its kernels */

	/* The packed mirror is allocated while the assignment kernels run: a fresh multi-GB hipMalloc is host-side
	 * work (page-table setup) that took 0.3 ms in one process and 63 ms in the next on the same box — as much
	 * as the rest of the build — and it needs nothing the GPU is busy with. */
	float	   *d_prow = nullptr;
	uint64_t   *d_ptid = nullptr;
	auto		alloc_mirror = [&]() -> int {
		if (d_prow)
			return 0;
		if (big_alloc((void **) &d_prow, (size_t) nrows * dim * sizeof(float))) return NDBHIP_ERR_HIP;
		if (big_alloc((void **) &d_ptid, (size_t) nrows * sizeof(uint64_t))) return NDBHIP_ERR_HIP;
		return 0;
	};

	rc = g_build_s16 ? assign_rows_s16(d_rows, nrows, dim, d_cent, k, d_list, nullptr, alloc_mirror, rows_until) : 1;
	if (rc < 0)
		return rc;
	if (rc == 1)				/* a shape the sweep does not take: the exact assignment on the vector ALU */
	{
		if (rows_until && (rc = rows_until(nrows)) != 0)
			return rc;
		rc = assign_rows(d_rows, nrows, dim, d_cent, k, true, d_list, nullptr, &aws);
	}
	if (rc)
		return rc;
	if (tids_ready && (rc = tids_ready()) != 0)
		return rc;
	rc = alloc_mirror();
	if (rc)
		return rc;
	lap("assign every row (+ malloc of the packed rows under it)");

	std::vector<int64_t> list_len;

	rc = pack_by_list(d_list, nrows, dim, k, d_rows, d_tids, d_prow, d_ptid, list_len);
	if (rc)
		return rc;
	lap("histograms / offsets / scatter");
	if (aws.release()) return NDBHIP_ERR_HIP;
	HIP_TRY(hipFree(d_list));
	lap("frees");

	/* adopt: centroids + packed lists become the index */
	ix->dm_cent_valid = false; ix->dm_all_valid = false;
	if (ix->d_centroids)
		HIP_TRY(hipFree(ix->d_centroids));
	ix->d_centroids = d_cent;
	ix->ncent = k;
	rc = ivf_set_layout(ix, list_len.data(), nullptr, nrows);
	if (rc)
		return rc;
	ivf_free_rows(ix);
	ix->d_vecs = d_prow;
	ix->d_tids = d_ptid;
	ix->own_rows = true;
	ix->nrows = nrows;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->cap_rows = nrows;
	ix->loaded = true;
	lap("adopt (layout upload, old rows freed)");
	if (out_iters)
		*out_iters = iters;
	if (g_build_prepare)
	{
		rc = ndbhip_ivf_prepare(ix, g_build_prepare);	/* so that the first query does not pay for it */
		lap("prepare (sublists, planes, norms, radii)");
		if (rc)
			return rc;
	}
	return NDBHIP_OK;
}

/* stable scatter of rows and TIDs by list id (list-major, input order inside a list): hist per 256-row block,
 * offsets, scatter; list_len out (host).  Synchronises the stream. */
static int
pack_by_list(const int *d_list, int64_t nrows, int dim, int k, const float *d_rows, const uint64_t *d_tids,
			 float *d_prow, uint64_t *d_ptid, std::vector<int64_t> &list_len)
{
	list_len.assign((size_t) k, 0);
	if (nrows == 0)
		return 0;
	const uint32_t nblocks = (uint32_t) ((nrows + NDB_PACK_BLOCK - 1) / NDB_PACK_BLOCK);
	const size_t nh = (size_t) k * nblocks;
	DevGuard	tmp;
	uint32_t   *d_hist = nullptr;
	int64_t    *d_scan = nullptr, *d_llen = nullptr;

	if (tmp.alloc(d_hist, nh * sizeof(uint32_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_scan, nh * sizeof(int64_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_llen, (size_t) k * sizeof(int64_t))) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemsetAsync(d_hist, 0, nh * sizeof(uint32_t), g.stream));
	hipLaunchKernelGGL(k_pack_hist, dim3(nblocks), dim3(NDB_PACK_BLOCK), 0, g.stream, d_list, nrows, k, nblocks, d_hist);
	hipLaunchKernelGGL(k_pack_list_totals, dim3((k + 63) / 64), dim3(64), 0, g.stream, (const uint32_t *) d_hist, k,
					   nblocks, d_llen);
	hipLaunchKernelGGL(k_pack_offsets, dim3(k), dim3(64), 0, g.stream, (const uint32_t *) d_hist, k, nblocks,
					   (const int64_t *) d_llen, d_scan);
	HIP_TRY(hipMemcpyAsync(list_len.data(), d_llen, (size_t) k * sizeof(int64_t), hipMemcpyDeviceToHost, g.stream));
	hipLaunchKernelGGL(k_pack_scatter, dim3(nblocks), dim3(NDB_PACK_BLOCK), 0, g.stream, d_list, nrows, dim, nblocks,
					   (const int64_t *) d_scan, d_rows, d_tids, d_prow, d_ptid);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(g.stream));
	return 0;
}

__global__ void
k_map_lists(const int *__restrict__ lists, int64_t n, const int *__restrict__ slot_of, int *__restrict__ out)
{
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;

	if (i < n)
		out[i] = slot_of[lists[i]];
}

extern "C" int ndbhip_comm_rank(void);
extern "C" int ndbhip_comm_world(void);
extern "C" int ndbhip_comm_allgather(const void *d_send, void *d_recv, size_t bytes);
extern "C" int ndbhip_comm_alltoallv(const void *d_send, const size_t *send_off, void *d_recv, const size_t *recv_off);

/* ivfbuild over the ranks of the communicator: include/ndbhip.h */
extern "C" int
ndbhip_ivf_build_sharded(ndbhip_ivf *ix, const float *d_rows, const uint64_t *d_tids, int64_t nrows_local,
						 int max_iter, int *out_iters, uint8_t *out_owned)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_build_sharded");
	const int	W = ndbhip_comm_world(), me = ndbhip_comm_rank();

	if (W <= 1)
	{
		const int	rc = ndbhip_ivf_build_device(ix, d_rows, d_tids, nrows_local, max_iter, out_iters);

		if (rc == 0 && out_owned)
			memset(out_owned, 1, (size_t) ix->nlists);
		return rc;
	}
	if (!ix || nrows_local < 0 || (nrows_local > 0 && (!d_rows || !d_tids)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	const int	dim = ix->dim, k = ix->nlists;
	DevGuard	tmp;
	int64_t    *d_i64 = nullptr, *d_i64_all = nullptr;
	int			rc;

	auto		trace = [&](const char *what) {
		if (g_debug_build)
			fprintf(stderr, "build_sharded[%d]: %s (last HIP error: %s)\n", me, what, hipGetErrorName(hipPeekAtLastError()));
	};

	/* 1. how many rows everyone holds */
	if (tmp.alloc(d_i64, (size_t) (k + 2) * sizeof(int64_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_i64_all, (size_t) W * (k + 2) * sizeof(int64_t))) return NDBHIP_ERR_HIP;
	std::vector<int64_t> h_send((size_t) k + 2, 0), h_all((size_t) W * (k + 2), 0);

	h_send[0] = nrows_local;
	HIP_TRY(hipMemcpyAsync(d_i64, h_send.data(), sizeof(int64_t), hipMemcpyHostToDevice, g.stream));
	if ((rc = ndbhip_comm_allgather(d_i64, d_i64_all, sizeof(int64_t))) != 0) return rc;
	HIP_TRY(hipMemcpyAsync(h_all.data(), d_i64_all, (size_t) W * sizeof(int64_t), hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	std::vector<int64_t> nloc(h_all.begin(), h_all.begin() + W);
	int64_t		ntotal = 0;

	for (int r = 0; r < W; r++)
		ntotal += nloc[(size_t) r];
	if (ntotal < 1 || ntotal > 0xFFFFFFFFll)
		return fail(NDBHIP_ERR_INVALID, "table of %lld rows", (long long) ntotal);
	const int	ns = (int) std::min<int64_t>(std::min<int64_t>(10000, (int64_t) k * 100), ntotal);

	if (ns < k)					/* ivf_am.c:596-601 */
		return fail(NDBHIP_ERR_INVALID, "ivf: not enough sample vectors (%d < %d)", ns, k);
	if (nloc[0] < ns)
		return fail(NDBHIP_ERR_UNSUPPORTED, "rank 0 holds %lld rows, the sample is the table's first %d", (long long) nloc[0], ns);

	trace("row counts gathered");
	/* 2. k-means on rank 0, centroids (and the iteration count) to everyone */
	float	   *d_cent = nullptr, *d_cent_all = nullptr;
	int			iters = 0;

	HIP_TRY(hipMalloc((void **) &d_cent, ((size_t) k * dim + 1) * sizeof(float)));
	if (tmp.alloc(d_cent_all, (size_t) W * ((size_t) k * dim + 1) * sizeof(float))) { (void) hipFree(d_cent); return NDBHIP_ERR_HIP; }
	auto		bail = [&](int code) { (void) hipStreamSynchronize(g.stream); (void) hipFree(d_cent); return code; };

	int		   *d_sasg = nullptr, *d_scnt = nullptr;	/* (function scope: DevGuard keeps their addresses) */

	if (me == 0)
	{
		if (tmp.alloc(d_sasg, (size_t) ns * sizeof(int))) return bail(NDBHIP_ERR_HIP);
		if (tmp.alloc(d_scnt, (size_t) k * sizeof(int))) return bail(NDBHIP_ERR_HIP);
		rc = ndbhip_kmeans_device(d_rows, ns, dim, k, max_iter, 0.001f, d_cent, d_sasg, d_scnt, &iters, nullptr);
		if (rc)
			return bail(rc);		/* (the other ranks are left in the all-gather: a failed build ends the job) */
		const float fi = (float) iters;

		if (hipMemcpyAsync(d_cent + (size_t) k * dim, &fi, sizeof(float), hipMemcpyHostToDevice, g.stream) != hipSuccess ||
			hipStreamSynchronize(g.stream) != hipSuccess)
			return bail(NDBHIP_ERR_HIP);
	}
	if ((rc = ndbhip_comm_allgather(d_cent, d_cent_all, ((size_t) k * dim + 1) * sizeof(float))) != 0) return bail(rc);
	if (me != 0)
	{
		float		fi = 0.0f;

		if (hipMemcpyAsync(d_cent, d_cent_all, ((size_t) k * dim + 1) * sizeof(float), hipMemcpyDeviceToDevice, g.stream) != hipSuccess ||
			hipMemcpyAsync(&fi, d_cent_all + (size_t) k * dim, sizeof(float), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
			hipStreamSynchronize(g.stream) != hipSuccess)
			return bail(NDBHIP_ERR_HIP);
		iters = (int) fi;
	}

	trace("centroids broadcast");
	/* 3. every rank assigns its own rows (the insert rule, screened on the matrix cores where the shape allows) */
	int		   *d_list = nullptr, *d_cnt = nullptr;
	const int64_t nl = nrows_local;

	if (tmp.alloc(d_list, (size_t) std::max<int64_t>(nl, 1) * sizeof(int))) return bail(NDBHIP_ERR_HIP);
	if (tmp.alloc(d_cnt, (size_t) k * sizeof(int))) return bail(NDBHIP_ERR_HIP);
	if (nl > 0)
	{
		rc = g_build_s16 ? assign_rows_s16(d_rows, nl, dim, d_cent, k, d_list, nullptr, nullptr) : 1;
		if (rc == 1)
			rc = assign_rows(d_rows, nl, dim, d_cent, k, true, d_list, nullptr);
		if (rc)
			return bail(rc);
	}

	trace("rows assigned");
	/* 4. list histograms of every rank */
	if (hipMemsetAsync(d_cnt, 0, (size_t) k * sizeof(int), g.stream) != hipSuccess) return bail(NDBHIP_ERR_HIP);
	if (nl > 0)
		hipLaunchKernelGGL(k_count_members, dim3((unsigned) ((nl + 255) / 256)), dim3(256), 0, g.stream, (const int *) d_list,
						   (int) nl, k, d_cnt);
	std::vector<int> hcnt((size_t) k);

	if (hipMemcpyAsync(hcnt.data(), d_cnt, (size_t) k * sizeof(int), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
		hipStreamSynchronize(g.stream) != hipSuccess)
		return bail(NDBHIP_ERR_HIP);
	for (int L = 0; L < k; L++)
		h_send[(size_t) L] = hcnt[(size_t) L];
	if (hipMemcpyAsync(d_i64, h_send.data(), (size_t) k * sizeof(int64_t), hipMemcpyHostToDevice, g.stream) != hipSuccess)
		return bail(NDBHIP_ERR_HIP);
	if ((rc = ndbhip_comm_allgather(d_i64, d_i64_all, (size_t) k * sizeof(int64_t))) != 0) return bail(rc);
	if (hipMemcpyAsync(h_all.data(), d_i64_all, (size_t) W * k * sizeof(int64_t), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
		hipStreamSynchronize(g.stream) != hipSuccess)
		return bail(NDBHIP_ERR_HIP);
	auto		cnt = [&](int r, int L) -> int64_t { return h_all[(size_t) r * k + (size_t) L]; };
	std::vector<int64_t> glob((size_t) k, 0);

	for (int r = 0; r < W; r++)
		for (int L = 0; L < k; L++)
			glob[(size_t) L] += cnt(r, L);

	trace("histograms gathered");
	/* 5. the deal: longest list first, to the rank with the fewest rows so far (ties: the lower rank); every rank
	 * computes the same one */
	std::vector<int> order((size_t) k), owner((size_t) k, 0);
	std::vector<int64_t> load((size_t) W, 0);

	for (int L = 0; L < k; L++)
		order[(size_t) L] = L;
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return glob[(size_t) a] > glob[(size_t) b]; });
	for (int L : order)
	{
		int			best = 0;

		for (int r = 1; r < W; r++)
			if (load[(size_t) r] < load[(size_t) best])
				best = r;
		owner[(size_t) L] = best;
		load[(size_t) best] += glob[(size_t) L];
	}

	/* 6. local rows ordered by (owner, list, heap order): a stable pack by slot_of[list] */
	std::vector<int> by_owner((size_t) k), slot_of((size_t) k);

	for (int L = 0; L < k; L++)
		by_owner[(size_t) L] = L;
	std::stable_sort(by_owner.begin(), by_owner.end(), [&](int a, int b) { return owner[(size_t) a] < owner[(size_t) b]; });
	for (int s2 = 0; s2 < k; s2++)
		slot_of[(size_t) by_owner[(size_t) s2]] = s2;
	int		   *d_slot_of = nullptr, *d_slots = nullptr;
	float	   *d_srow = nullptr;
	uint64_t   *d_stid = nullptr;
	std::vector<int64_t> slot_len;

	if (tmp.alloc(d_slot_of, (size_t) k * sizeof(int))) return bail(NDBHIP_ERR_HIP);
	if (tmp.alloc(d_slots, (size_t) std::max<int64_t>(nl, 1) * sizeof(int))) return bail(NDBHIP_ERR_HIP);
	if (tmp.alloc(d_srow, (size_t) std::max<int64_t>(nl, 1) * dim * sizeof(float))) return bail(NDBHIP_ERR_HIP);
	if (tmp.alloc(d_stid, (size_t) std::max<int64_t>(nl, 1) * sizeof(uint64_t))) return bail(NDBHIP_ERR_HIP);
	if (hipMemcpyAsync(d_slot_of, slot_of.data(), (size_t) k * sizeof(int), hipMemcpyHostToDevice, g.stream) != hipSuccess)
		return bail(NDBHIP_ERR_HIP);
	if (nl > 0)
		hipLaunchKernelGGL(k_map_lists, dim3((unsigned) ((nl + 255) / 256)), dim3(256), 0, g.stream, (const int *) d_list, nl,
						   (const int *) d_slot_of, d_slots);
	if ((rc = pack_by_list(d_slots, nl, dim, k, d_rows, d_tids, d_srow, d_stid, slot_len)) != 0) return bail(rc);

	trace("packed by owner");
	/* 7. every row to its list's owner */
	std::vector<size_t> so_rows((size_t) W + 1, 0), ro_rows((size_t) W + 1, 0), so((size_t) W + 1), ro((size_t) W + 1);

	for (int L = 0; L < k; L++)
		so_rows[(size_t) owner[(size_t) L] + 1] += (size_t) cnt(me, L);
	for (int r = 0; r < W; r++)
	{
		so_rows[(size_t) r + 1] += so_rows[(size_t) r];
		size_t		from_r = 0;

		for (int L = 0; L < k; L++)
			if (owner[(size_t) L] == me)
				from_r += (size_t) cnt(r, L);
		ro_rows[(size_t) r + 1] = ro_rows[(size_t) r] + from_r;
	}
	const int64_t nmine = (int64_t) ro_rows[(size_t) W];
	float	   *d_rrow = nullptr, *d_prow = nullptr;
	uint64_t   *d_rtid = nullptr, *d_ptid = nullptr;
	int		   *d_rlist = nullptr;

	if (tmp.alloc(d_rrow, (size_t) std::max<int64_t>(nmine, 1) * dim * sizeof(float))) return bail(NDBHIP_ERR_HIP);
	if (tmp.alloc(d_rtid, (size_t) std::max<int64_t>(nmine, 1) * sizeof(uint64_t))) return bail(NDBHIP_ERR_HIP);
	if (tmp.alloc(d_rlist, (size_t) std::max<int64_t>(nmine, 1) * sizeof(int))) return bail(NDBHIP_ERR_HIP);
	for (int r = 0; r <= W; r++)
	{
		so[(size_t) r] = so_rows[(size_t) r] * (size_t) dim * sizeof(float);
		ro[(size_t) r] = ro_rows[(size_t) r] * (size_t) dim * sizeof(float);
	}
	if ((rc = ndbhip_comm_alltoallv(d_srow, so.data(), d_rrow, ro.data())) != 0) return bail(rc);
	for (int r = 0; r <= W; r++)
	{
		so[(size_t) r] = so_rows[(size_t) r] * sizeof(uint64_t);
		ro[(size_t) r] = ro_rows[(size_t) r] * sizeof(uint64_t);
	}
	if ((rc = ndbhip_comm_alltoallv(d_stid, so.data(), d_rtid, ro.data())) != 0) return bail(rc);

	trace("rows exchanged");
	/* 8. what arrived is ordered (source rank, list, heap order); sources hold consecutive heap ranges, so a stable
	 * pack by list puts every list in heap order */
	std::vector<int> rlist((size_t) std::max<int64_t>(nmine, 1));
	size_t		w = 0;

	for (int r = 0; r < W; r++)
		for (int L = 0; L < k; L++)
			if (owner[(size_t) L] == me)
				for (int64_t j = 0; j < cnt(r, L); j++)
					rlist[w++] = L;
	if (hipMemcpyAsync(d_rlist, rlist.data(), (size_t) std::max<int64_t>(nmine, 1) * sizeof(int), hipMemcpyHostToDevice, g.stream) != hipSuccess)
		return bail(NDBHIP_ERR_HIP);
	if (big_alloc((void **) &d_prow, (size_t) std::max<int64_t>(nmine, 1) * dim * sizeof(float))) return bail(NDBHIP_ERR_HIP);
	if (big_alloc((void **) &d_ptid, (size_t) std::max<int64_t>(nmine, 1) * sizeof(uint64_t)))
	{
		big_free(d_prow);
		return bail(NDBHIP_ERR_HIP);
	}
	std::vector<int64_t> own_len;

	if ((rc = pack_by_list(d_rlist, nmine, dim, k, d_rrow, d_rtid, d_prow, d_ptid, own_len)) != 0)
	{
		big_free(d_prow);
		big_free(d_ptid);
		return bail(rc);
	}

	trace("packed by list");
	/* 9. adopt: all centroids, global lengths, the rows of the lists held here */
	std::vector<uint8_t> owned((size_t) k);

	for (int L = 0; L < k; L++)
		owned[(size_t) L] = owner[(size_t) L] == me;
	ix->dm_cent_valid = false; ix->dm_all_valid = false;
	if (ix->d_centroids)
		(void) hipFree(ix->d_centroids);
	ix->d_centroids = d_cent;
	ix->ncent = k;
	rc = ivf_set_layout(ix, glob.data(), owned.data(), nmine);
	if (rc)
	{
		big_free(d_prow);
		big_free(d_ptid);
		return rc;
	}
	ivf_free_rows(ix);
	ix->d_vecs = d_prow;
	ix->d_tids = d_ptid;
	ix->own_rows = true;
	ix->nrows = nmine;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->cap_rows = std::max<int64_t>(nmine, 1);
	ix->f16 = false;
	ix->loaded = true;
	trace("adopted");
	if (out_iters)
		*out_iters = iters;
	if (out_owned)
		memcpy(out_owned, owned.data(), (size_t) k);
	return NDBHIP_OK;
}

/* radius of the sublists one list was just assigned to: rad_bits[sid[i]] = max |row i - cents[sid[i]]| (rounded up) */
__global__ __launch_bounds__(256) void
k_s16_assigned_radius(const float *__restrict__ rows, int64_t n, int dim, const float *__restrict__ cents,
					  const int *__restrict__ sid, uint32_t *__restrict__ rad_bits)
{
	const int	lane = threadIdx.x & 63;
	const int64_t i = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);

	if (i >= n)
		return;
	const int	sd = sid[i];
	const float *x = rows + (size_t) i * dim, *c = cents + (size_t) sd * dim;
	double		s = 0.0;

	for (int d = lane; d < dim; d += 64)
	{
		const double t = (double) x[d] - (double) c[d];

		s += t * t;
	}
	s = wave_sum_f64(s);
	if (lane == 0)
	{
		const double r = __builtin_sqrt(s) * (1.0 + 9.5367431640625e-7);
		const uint32_t bits = (r <= 3.0e38) ? __float_as_uint(__double2float_ru(r)) : 0x7F800000u;

		if (bits > __atomic_load_n(&rad_bits[sd], __ATOMIC_RELAXED))
			atomicMax(&rad_bits[sd], bits);
	}
}

/*
 * A dense block of n vectors (sublist centres; the index's centroids) as ONE list of the two-plane sweep, whose
 * MODE 3 gives every query's squared distance to every one of them within the sweep's own error bound
 * s16_e(dim, |q|^2, xmax) — a (queries x vectors x dim) contraction at matrix-core speed where the exact recipe
 * would run on the vector ALU.  The sweep's tables depend on (n, batch size) only and are kept.
 */
static int
s16mat_prepare(S16Mat &M, const float *d_src, int n, int dim)
{
	const int	dimp = (dim + 63) & ~63;
	const size_t cblk = ((size_t) n + 31) / 32;
	const int64_t hloc[2] = {0, (int64_t) n};
	const uint32_t hblk[2] = {0, (uint32_t) cblk};

	if (grow(M.planes, M.planes_n, (cblk + 8) * (size_t) (dimp / S16_CH) * 4096)) return NDBHIP_ERR_HIP;
	if (grow(M.rn2, M.rn2_n, (size_t) n)) return NDBHIP_ERR_HIP;
	if (grow(M.rexp, M.rexp_n, (size_t) n)) return NDBHIP_ERR_HIP;
	if (grow(M.xmax, M.xmax_n, (size_t) 4)) return NDBHIP_ERR_HIP;		/* [0] max norm, [2..3] block offsets */
	if (grow(M.loc, M.loc_n, (size_t) 2)) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemsetAsync(M.planes, 0, (cblk + 8) * (size_t) (dimp / S16_CH) * 4096, g.stream));
	HIP_TRY(hipMemsetAsync(M.xmax, 0, 16, g.stream));
	HIP_TRY(hipMemcpyAsync(M.loc, hloc, sizeof(hloc), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(M.xmax + 2, hblk, sizeof(hblk), hipMemcpyHostToDevice, g.stream));
	hipLaunchKernelGGL(k_s16_row_prep<0>, dim3((unsigned) ((n + 3) / 4)), dim3(256), 0, g.stream, (const void *) d_src,
					   (int64_t) n, dim, dimp, (const int64_t *) M.loc, (const uint32_t *) (M.xmax + 2), 1,
					   M.planes, M.rn2, M.rexp, M.xmax, (const int64_t *) nullptr);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(g.stream));		/* hloc / hblk are locals */
	M.nq = -1;
	M.n = n;
	M.src = d_src;
	return 0;
}

static int
s16mat_run(S16Mat &M, int dim, const unsigned char *qplanes, const float *qn2, const int *qexp, float2 *qthr, int nq,
		   float *out, uint32_t stride)
{
	const int	ng = M.n, dimp = (dim + 63) & ~63;
	const uint32_t nrt = (uint32_t) ((ng + 127) / 128), nqt = (uint32_t) ((nq + S16_QT - 1) / S16_QT), nitems = nrt * nqt;
	const size_t mw = NDB_ASG_META(nq);

	if (M.nq != nq)
	{
		if (grow(M.meta, M.meta_n, 2 * mw)) return NDBHIP_ERR_HIP;
		if (grow(M.pairs, M.pairs_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		if (grow(M.desc, M.desc_n, (size_t) nitems)) return NDBHIP_ERR_HIP;
		if (grow(M.heads, M.heads_n, (size_t) 8 * NDB_QHEAD_STRIDE)) return NDBHIP_ERR_HIP;
		if (grow(M.zero, M.zero_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemsetAsync(M.zero, 0, (size_t) nq, g.stream));
		/* the build's table kernel with the roles it has there: the "rows" are the vectors, the "centroids" the queries */
		hipLaunchKernelGGL(k_assign_tables, dim3(1), dim3(64), 0, g.stream, (const unsigned char *) M.zero, nq, (uint32_t) ng,
						   (uint32_t) ng, M.pairs, M.meta, M.meta + mw);
		hipLaunchKernelGGL(k_s16_items, dim3((nitems + 255) / 256), dim3(256), 0, g.stream, (const uint32_t *) (M.meta + 6),
						   (const uint32_t *) (M.meta + 3), (const uint32_t *) (M.meta + 2), 1, 128u, nitems,
						   (S16Desc *) M.desc, M.heads + 8 * NDB_QHEAD_STRIDE - 1);
		M.nq = nq;
	}
	HIP_TRY(hipMemsetAsync(M.heads, 0, (size_t) 8 * NDB_QHEAD_STRIDE * sizeof(unsigned int), g.stream));
	const uint32_t *m32 = M.meta;
	IvfDev		dv = {};

	dv.vecs = M.src;
	dv.loc_off = M.loc;
	dv.own_len = m32 + 2;
	dv.glob_len = m32 + 2;
	dv.dim = dim;
	dv.ncent = 1;
	dv.nlists = 1;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_L2, 0, 4, 2, 0, 3>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, dv,
					   (const unsigned char *) M.planes, m32, (const float *) M.rn2, (const int16_t *) M.rexp,
					   qplanes, (uint32_t) dimp * 4u, qn2, qexp, qthr, m32 + 17, 1, m32 + 3, m32 + 4, (const S16Desc *) M.desc,
					   (const PairRec *) M.pairs, M.heads, m32 + 8, (unsigned int *) nullptr,
					   reinterpret_cast<uint2 *>(out), stride, (uint32_t *) nullptr, 0, dimp / S16_CH, nitems, 0u,
					   (const uint32_t *) nullptr);
	HIP_TRY(hipGetLastError());
	return 0;
}

/* ---- lists of a few hundred to 2048 rows ("mid" lists): all of them regrouped in three launches ---- */
struct MidDesc
{
	int64_t		row0;			/* first mirror row of the list */
	uint32_t	len;
	uint32_t	cent0;			/* first of its sample rows in the gathered centre block */
	uint32_t	S;				/* sample rows = sublists (<= 32) */
	uint32_t	sorted0;		/* first slot of its rows in the order array */
};

/* one wave per row of the mid lists: the nearest of its own list's sample rows (fp32 sums: any assignment gives valid
 * sublists, whose radii are measured afterwards), that sublist's size and an upper estimate of its radius */
__global__ __launch_bounds__(256) void
k_s16_mid_assign(const float *__restrict__ vecs, int dim, const MidDesc *__restrict__ md, const uint32_t *__restrict__ mrow_off,
				 int nmid, const float *__restrict__ cents, uint8_t *__restrict__ sid, uint32_t *__restrict__ rad_bits,
				 uint32_t *__restrict__ cnt)
{
	const int	lane = threadIdx.x & 63;
	const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);

	if (w >= mrow_off[nmid])
		return;
	int			lo = 0, hi = nmid;

	while (hi - lo > 1)
	{
		const int	mid = (lo + hi) >> 1;

		if (mrow_off[mid] <= w)
			lo = mid;
		else
			hi = mid;
	}
	const MidDesc d = md[lo];
	const float *x = vecs + (size_t) (d.row0 + (w - mrow_off[lo])) * dim;
	float		best = __uint_as_float(0x7F800000u);
	uint32_t	bi = 0;

	for (uint32_t j = 0; j < d.S; j++)
	{
		const float *c = cents + (size_t) (d.cent0 + j) * dim;
		float		a = 0.0f;

		for (int i = lane; i < dim; i += 64)
		{
			const float t = x[i] - c[i];

			a = __builtin_fmaf(t, t, a);
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
			a += __shfl_xor(a, off, 64);
		if (a < best)
		{
			best = a;
			bi = j;
		}
	}
	if (lane == 0)
	{
		const float r = __builtin_sqrtf(best) * 1.001f;

		sid[w] = (uint8_t) bi;
		atomicAdd(&cnt[d.cent0 + bi], 1u);
		if (r == r && __float_as_uint(r) > __atomic_load_n(&rad_bits[d.cent0 + bi], __ATOMIC_RELAXED))
			atomicMax(&rad_bits[d.cent0 + bi], __float_as_uint(r));
		if (!(r == r))
			atomicMax(&rad_bits[d.cent0 + bi], 0x7F800000u);
	}
}

/* one block per mid list: its rows in sublist order, stable (sorted[sorted0 + j] = index in the list of the row that
 * comes j-th) */
__global__ __launch_bounds__(256) void
k_s16_mid_sort(const MidDesc *__restrict__ md, const uint32_t *__restrict__ mrow_off, const uint8_t *__restrict__ sid,
			   uint64_t *__restrict__ sorted)
{
	__shared__ uint8_t s_sid[2048];
	__shared__ uint32_t s_part[256], s_base;
	const MidDesc d = md[blockIdx.x];
	const uint8_t *mine = sid + mrow_off[blockIdx.x];
	const uint32_t t = threadIdx.x, per = (d.len + 255) / 256, r0 = t * per, r1 = min(d.len, r0 + per);

	for (uint32_t r = t; r < d.len; r += 256)
		s_sid[r] = mine[r];
	if (t == 0)
		s_base = 0;
	__syncthreads();
	for (uint32_t v = 0; v < d.S; v++)
	{
		uint32_t	n = 0;

		for (uint32_t r = r0; r < r1; r++)
			n += s_sid[r] == v ? 1u : 0u;
		s_part[t] = n;
		__syncthreads();
		for (int off = 1; off < 256; off <<= 1)
		{
			const uint32_t a = t >= (uint32_t) off ? s_part[t - off] : 0u;

			__syncthreads();
			s_part[t] += a;
			__syncthreads();
		}
		uint32_t	pos = s_base + s_part[t] - n;

		for (uint32_t r = r0; r < r1; r++)
			if (s_sid[r] == v)
				sorted[d.sorted0 + pos++] = (uint64_t) r;
		__syncthreads();
		if (t == 255)
			s_base += s_part[255];
		__syncthreads();
	}
}

/*
 * Sublists of the matrix-core screen (ndbhip_screen16.h, "Sublists"): every list longer than screen16_sub_min rows
 * is regrouped, inside the planes only, by the nearest of len / screen16_sub_rows of its own rows (taken at equal
 * strides; no k-means: a sample row of every cluster the list mixes is enough, and the assignment is one screened
 * pass).  A regrouping is kept per list only where it shrinks the radius
 * (row-weighted mean sublist radius < 0.6 x the list's own).  Outputs the d_sub_* tables, the planes' order (d_perm, d_posof) and bo = first 32-row block of every
 * sublist.  Leaves ix->s16_sub false when no list is long enough.
 */
/* an fp16 mirror's rows as the reference decodes them (fp16_to_float, quirk Q20 for subnormals): the values the
 * exact arithmetic sees, so radii measured on them are radii of what is scored */
__global__ void
k_rows_decode_f16(const uint16_t *__restrict__ src, size_t n, float *__restrict__ out)
{
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;

	if (i < n)
		out[i] = h2f_ref(src[i]);
}

static int
ivf_s16_build_sublists(ndbhip_ivf *ix, std::vector<uint32_t> &bo, bool slack, const float *rows32, const float *list_centres)
{
	/* the lists' own centres: their centroids, or (cosine) the centroids divided by their norms — the space rows32 is in */
	const float *lcent = list_centres ? list_centres : (const float *) ix->d_centroids;

	const int	nc = ix->ncent, dim = ix->dim;
	std::vector<int> giant, midl;
	/* an fp16 mirror is regrouped on a transient fp32 copy of its decoded rows (the copy goes when this returns;
	 * the planes, the seeds and the exact re-score read the fp16 rows themselves) */
	struct Rows32
	{
		float	   *p = nullptr;
		~Rows32() { if (p) big_free(p); }
	}			dec;
	const float *vecs32 = (const float *) ix->d_vecs;

	if (rows32)
		vecs32 = rows32;
	else if (ix->f16)
	{
		const size_t ne = (size_t) ix->nrows * dim;

		if (ne == 0)
			return 0;
		if (big_alloc((void **) &dec.p, ne * sizeof(float))) return NDBHIP_ERR_HIP;
		hipLaunchKernelGGL(k_rows_decode_f16, dim3((unsigned) ((ne + 255) / 256)), dim3(256), 0, g.stream,
						   (const uint16_t *) ix->d_vecs, ne, dec.p);
		vecs32 = dec.p;
	}

	for (int c = 0; c < nc; c++)
		if (ix->own_len[c] > (int64_t) g_s16_sub_min)
		{
			/* up to 2048 rows and 32 sample rows: regrouped together with all the other lists of that size */
			if (ix->own_len[c] <= 2048 && (ix->own_len[c] + g_s16_sub_rows - 1) / g_s16_sub_rows <= 32)
				midl.push_back(c);
			else
				giant.push_back(c);
		}
	if ((giant.empty() && midl.empty()) || ix->nrows < 1)
		return 0;
	/* the lists' own radii (around their centroids): what regrouping has to beat */
	if (grow(ix->d_lrad, ix->d_lrad_n, (size_t) nc)) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemsetAsync(ix->d_lrad, 0, (size_t) nc * sizeof(uint32_t), g.stream));
	hipLaunchKernelGGL(k_s16_list_radius<0>, dim3((unsigned) ((ix->nrows + 3) / 4)), dim3(256), 0, g.stream, (const void *) vecs32,
					   ix->nrows, dim, (const int64_t *) ix->d_loc_off, nc, lcent, ix->d_lrad);
	std::vector<float> lrad((size_t) nc);

	HIP_TRY(hipMemcpyAsync(lrad.data(), ix->d_lrad, (size_t) nc * 4, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));

	/* phase A: every long list assigned to sample rows of its own; kept only where that shrinks the radius */
	DevGuard	tmp;
	int64_t		maxlen = 0, sumlen = 0;
	size_t		maxS = 0;
	std::vector<uint32_t> nsub_of((size_t) nc, 1);

	for (int c : giant)
	{
		nsub_of[(size_t) c] = (uint32_t) std::min<int64_t>(4096, (ix->own_len[c] + g_s16_sub_rows - 1) / g_s16_sub_rows);
		maxlen = std::max<int64_t>(maxlen, ix->own_len[c]);
		sumlen += ix->own_len[c];
		maxS = std::max<size_t>(maxS, nsub_of[(size_t) c]);
	}
	int		   *d_sid = nullptr;
	float	   *d_dummy = nullptr, *d_dummy2 = nullptr, *d_cents = nullptr;
	uint64_t   *d_iota = nullptr, *d_sorted_all = nullptr;
	int64_t    *d_which = nullptr;
	uint32_t   *d_radtmp = nullptr;
	size_t		ncent_all = 0;

	for (int c : giant)
		ncent_all += nsub_of[(size_t) c];
	size_t		mid_rows = 0, mid_cents = 0;

	for (int c : midl)
	{
		nsub_of[(size_t) c] = (uint32_t) ((ix->own_len[c] + g_s16_sub_rows - 1) / g_s16_sub_rows);
		mid_rows += (size_t) ix->own_len[c];
		mid_cents += nsub_of[(size_t) c];
	}
	ncent_all += mid_cents;
	maxlen = std::max<int64_t>(maxlen, 1);
	if (tmp.alloc(d_sid, (size_t) maxlen * sizeof(int))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_dummy, (size_t) maxlen * sizeof(float))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_dummy2, (size_t) maxlen * sizeof(float))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_iota, (size_t) maxlen * sizeof(uint64_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_sorted_all, (size_t) sumlen * sizeof(uint64_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_which, maxS * sizeof(int64_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_radtmp, maxS * sizeof(uint32_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_cents, ncent_all * (size_t) dim * sizeof(float))) return NDBHIP_ERR_HIP;
	{
		std::vector<uint64_t> iota((size_t) maxlen);

		for (int64_t j = 0; j < maxlen; j++)
			iota[(size_t) j] = (uint64_t) j;
		HIP_TRY(hipMemcpyAsync(d_iota, iota.data(), (size_t) maxlen * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipMemsetAsync(d_dummy, 0, (size_t) maxlen * sizeof(float), g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
	}
	struct Kept { int c; size_t cent0; const uint64_t *sorted; std::vector<int64_t> lens; };
	std::vector<Kept> kept;
	size_t		cbase = 0, sbase = 0;

	if (!midl.empty())
	{
		/* the mid lists, all at once: sample rows gathered, every row to the nearest sample of its own list, the
		 * lists' rows ordered by sublist; kept, as for the long lists, only where that shrinks the radius */
		const int	nmid = (int) midl.size();
		std::vector<MidDesc> md((size_t) nmid);
		std::vector<uint32_t> mro((size_t) nmid + 1, 0);
		std::vector<int64_t> which(mid_cents);
		MidDesc    *d_md = nullptr;
		uint32_t   *d_mro = nullptr, *d_mrad = nullptr, *d_mcnt = nullptr;
		int64_t    *d_mwhich = nullptr, *d_keep = nullptr;
		float	   *d_mcents = nullptr;
		uint8_t    *d_msid = nullptr;
		uint64_t   *d_msorted = nullptr;
		size_t		c0 = 0;

		for (int m = 0; m < nmid; m++)
		{
			const int	c = midl[(size_t) m];
			const int64_t len = ix->own_len[c];
			const uint32_t S = nsub_of[(size_t) c];

			md[(size_t) m].row0 = ix->loc_off[c];
			md[(size_t) m].len = (uint32_t) len;
			md[(size_t) m].cent0 = (uint32_t) c0;
			md[(size_t) m].S = S;
			md[(size_t) m].sorted0 = mro[(size_t) m];
			mro[(size_t) m + 1] = mro[(size_t) m] + (uint32_t) len;
			for (uint32_t j = 0; j < S; j++)
				which[c0 + j] = ix->loc_off[c] + (int64_t) (((__int128) len * j) / S);
			c0 += S;
		}
		if (tmp.alloc(d_md, (size_t) nmid * sizeof(MidDesc))) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_mro, ((size_t) nmid + 1) * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_mrad, mid_cents * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_mcnt, mid_cents * 4)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_mwhich, mid_cents * 8)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_keep, mid_cents * 8)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_mcents, mid_cents * (size_t) dim * sizeof(float))) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_msid, mid_rows)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(d_msorted, mid_rows * 8)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemcpyAsync(d_md, md.data(), (size_t) nmid * sizeof(MidDesc), hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipMemcpyAsync(d_mro, mro.data(), ((size_t) nmid + 1) * 4, hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipMemcpyAsync(d_mwhich, which.data(), mid_cents * 8, hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipMemsetAsync(d_mrad, 0, mid_cents * 4, g.stream));
		HIP_TRY(hipMemsetAsync(d_mcnt, 0, mid_cents * 4, g.stream));
		hipLaunchKernelGGL(k_rows_gather, dim3((unsigned) mid_cents), dim3(256), 0, g.stream, vecs32, dim,
						   (const int64_t *) d_mwhich, d_mcents);
		hipLaunchKernelGGL(k_s16_mid_assign, dim3((unsigned) ((mid_rows + 3) / 4)), dim3(256), 0, g.stream, vecs32,
						   dim, (const MidDesc *) d_md, (const uint32_t *) d_mro, nmid, (const float *) d_mcents, d_msid, d_mrad, d_mcnt);
		hipLaunchKernelGGL(k_s16_mid_sort, dim3((unsigned) nmid), dim3(256), 0, g.stream, (const MidDesc *) d_md,
						   (const uint32_t *) d_mro, (const uint8_t *) d_msid, d_msorted);
		std::vector<uint32_t> mrad(mid_cents), mcnt(mid_cents);
		std::vector<int64_t> keepidx;

		HIP_TRY(hipMemcpyAsync(mrad.data(), d_mrad, mid_cents * 4, hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipMemcpyAsync(mcnt.data(), d_mcnt, mid_cents * 4, hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
		for (int m = 0; m < nmid; m++)
		{
			const int	c = midl[(size_t) m];
			const MidDesc &d = md[(size_t) m];
			double		wr = 0.0;

			for (uint32_t j = 0; j < d.S; j++)
			{
				float		r;

				memcpy(&r, &mrad[d.cent0 + j], 4);
				wr += (double) mcnt[d.cent0 + j] * (std::isfinite(r) ? (double) r : 3.0e38);
			}
			wr /= (double) d.len;
			if (std::isfinite(lrad[(size_t) c]) && wr < 0.6 * (double) lrad[(size_t) c])
			{
				Kept		kp;

				kp.c = c;
				kp.cent0 = cbase;
				kp.sorted = d_msorted + d.sorted0;
				for (uint32_t j = 0; j < d.S; j++)
				{
					kp.lens.push_back((int64_t) mcnt[d.cent0 + j]);
					keepidx.push_back((int64_t) (d.cent0 + j));
				}
				kept.push_back(std::move(kp));
				cbase += d.S;
			}
			else
				nsub_of[(size_t) c] = 1;
		}
		if (!keepidx.empty())
		{
			HIP_TRY(hipMemcpyAsync(d_keep, keepidx.data(), keepidx.size() * 8, hipMemcpyHostToDevice, g.stream));
			hipLaunchKernelGGL(k_rows_gather, dim3((unsigned) keepidx.size()), dim3(256), 0, g.stream, (const float *) d_mcents, dim,
							   (const int64_t *) d_keep, d_cents);
			HIP_TRY(hipStreamSynchronize(g.stream));		/* keepidx is a local */
		}
		if (g_debug_s16)
			fprintf(stderr, "s16 sublists: %zu of %d lists of %d .. 2048 rows are worth regrouping\n", kept.size(), nmid, g_s16_sub_min);
	}

	for (int c : giant)
	{
		const int64_t len = ix->own_len[c], row0 = ix->loc_off[c];
		const int	S = (int) nsub_of[(size_t) c];
		const float *lrows = vecs32 + (size_t) row0 * dim;
		float	   *cents = d_cents + cbase * (size_t) dim;
		std::vector<int64_t> which((size_t) S);
		std::vector<uint32_t> rad((size_t) S);

		for (int j = 0; j < S; j++)
			which[(size_t) j] = (int64_t) (((__int128) len * j) / S);
		HIP_TRY(hipMemcpyAsync(d_which, which.data(), (size_t) S * sizeof(int64_t), hipMemcpyHostToDevice, g.stream));
		hipLaunchKernelGGL(k_rows_gather, dim3(S), dim3(256), 0, g.stream, lrows, dim, (const int64_t *) d_which, cents);
		HIP_TRY(hipStreamSynchronize(g.stream));		/* `which` is a local */
		int			rc = g_build_s16 ? assign_rows_s16(lrows, len, dim, cents, S, d_sid, nullptr, nullptr) : 1;

		if (rc == 1)
			rc = assign_rows(lrows, len, dim, cents, S, true, d_sid, nullptr);
		if (rc)
			return rc;
		HIP_TRY(hipMemsetAsync(d_radtmp, 0, (size_t) S * 4, g.stream));
		hipLaunchKernelGGL(k_s16_assigned_radius, dim3((unsigned) ((len + 3) / 4)), dim3(256), 0, g.stream, lrows, len, dim,
						   (const float *) cents, (const int *) d_sid, d_radtmp);
		HIP_TRY(hipMemcpyAsync(rad.data(), d_radtmp, (size_t) S * 4, hipMemcpyDeviceToHost, g.stream));
		/* the list's rows in sublist order: a stable pack of (dummy row, index in list) by sublist id */
		Kept		kp;

		kp.c = c;
		kp.cent0 = cbase;
		kp.sorted = d_sorted_all + sbase;
		rc = pack_by_list(d_sid, len, 1, S, d_dummy, d_iota, d_dummy2, d_sorted_all + sbase, kp.lens);	/* (synchronises) */
		if (rc)
			return rc;
		double		wr = 0.0;

		for (int j = 0; j < S; j++)
		{
			float		r;

			memcpy(&r, &rad[(size_t) j], 4);
			wr += (double) kp.lens[(size_t) j] * (std::isfinite(r) ? (double) r : 3.0e38);
		}
		wr /= (double) len;
		/* worth it when a sublist's rows sit well inside their list's ball: clustered rows do (a component's spread
		 * against the distance between components), rows without structure do not (every pair is equally far) */
		if (std::isfinite(lrad[(size_t) c]) && wr < 0.6 * (double) lrad[(size_t) c])
		{
			kept.push_back(std::move(kp));
			cbase += (size_t) S;
			sbase += (size_t) len;
		}
		else
			nsub_of[(size_t) c] = 1;
	}
	if (g_debug_s16)
		fprintf(stderr, "s16 sublists: %zu of %zu long lists are worth regrouping\n", kept.size(), giant.size());
	if (kept.empty())
		return 0;

	/* phase B: the tables */
	std::vector<uint32_t> first((size_t) nc + 1, 0);
	size_t		nsub = 0;
	const size_t nsub_g = cbase;

	/* slack (the centred planes take appends in place): a regrouped list gets one more, empty sublist around its own
	 * centroid — the rows inserted later land there, in insertion order — and every bucket that takes appends has
	 * spare 32-row blocks behind it */
	std::vector<uint32_t> nsub_eff(nsub_of);

	for (int c = 0; c < nc; c++)
	{
		if (slack && nsub_of[(size_t) c] > 1)
			nsub_eff[(size_t) c]++;
		first[(size_t) c] = (uint32_t) nsub;
		nsub += nsub_eff[(size_t) c];
	}
	first[(size_t) nc] = (uint32_t) nsub;
	if (grow(ix->d_perm, ix->d_perm_n, (size_t) ix->nrows)) return NDBHIP_ERR_HIP;
	if (grow(ix->d_posof, ix->d_posof_n, (size_t) ix->nrows)) return NDBHIP_ERR_HIP;
	/* cosine (list_centres given): the lists' own normalised centres join the matrix of centres, so that a list that is its
	 * own sublist has a distance in the normalised space too (the centroid scan's is in the rows' own space) */
	const size_t ncol = nsub_g + (list_centres ? (size_t) nc : 0);

	if (grow(ix->d_subcent, ix->d_subcent_n, ncol * (size_t) dim)) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemcpyAsync(ix->d_subcent, d_cents, nsub_g * (size_t) dim * sizeof(float), hipMemcpyDeviceToDevice, g.stream));
	if (list_centres)
		HIP_TRY(hipMemcpyAsync(ix->d_subcent + nsub_g * (size_t) dim, list_centres, (size_t) nc * dim * sizeof(float),
							   hipMemcpyDeviceToDevice, g.stream));
	hipLaunchKernelGGL(k_s16_identity_perm, dim3((unsigned) ((ix->nrows + 255) / 256)), dim3(256), 0, g.stream, ix->nrows,
					   (const int64_t *) ix->d_loc_off, nc, ix->d_perm, ix->d_posof);
	std::vector<uint32_t> sub_len(nsub, 0);
	std::vector<int> sub_gidx(nsub, -1);

	for (const Kept &kp : kept)
	{
		const int64_t len = ix->own_len[kp.c], row0 = ix->loc_off[kp.c];

		hipLaunchKernelGGL(k_s16_sub_perm, dim3((unsigned) ((len + 255) / 256)), dim3(256), 0, g.stream,
						   kp.sorted, len, row0, ix->d_perm, ix->d_posof);
		for (size_t j = 0; j < kp.lens.size(); j++)
		{
			sub_len[first[(size_t) kp.c] + j] = (uint32_t) kp.lens[j];
			sub_gidx[first[(size_t) kp.c] + j] = (int) (kp.cent0 + j);
		}
	}
	for (int c = 0; c < nc; c++)
		if (nsub_of[(size_t) c] == 1)
			sub_len[first[(size_t) c]] = (uint32_t) ix->own_len[c];
	if (list_centres)
		for (int c = 0; c < nc; c++)
			for (uint32_t s2 = first[(size_t) c]; s2 < first[(size_t) c + 1]; s2++)
				if (sub_gidx[s2] < 0)
					sub_gidx[s2] = (int) (nsub_g + (size_t) c);		/* (whole lists and the tail buckets: the list's own centre) */
	/* offsets: sublists are consecutive in the planes, every one starts a new 32-row block */
	std::vector<int64_t> sub_loc(nsub + 1, 0);
	std::vector<const float *> cptr(nsub);
	uint64_t	nb = 0;

	bo.assign(nsub + 1, 0);
	ix->s16_tail.assign((size_t) nc, -1);
	for (int c = 0; c < nc; c++)
		for (uint32_t j = 0; j < nsub_eff[(size_t) c]; j++)
		{
			const size_t s2 = first[(size_t) c] + j;
			const bool	tail = j + 1 == nsub_eff[(size_t) c];

			sub_loc[s2 + 1] = sub_loc[s2] + (int64_t) sub_len[s2];
			bo[s2] = (uint32_t) nb;
			nb += (uint64_t) ((sub_len[s2] + 31) / 32);
			if (slack && tail)
			{
				nb += std::max<uint64_t>(2, (uint64_t) ix->own_len[c] / 256);
				ix->s16_tail[(size_t) c] = (int) s2;
			}
		}
	bo[nsub] = (uint32_t) nb;
	ix->s16_blen.assign(sub_len.begin(), sub_len.end());
	if (nb + 8 > 0xFFFFFFFFull)
		return fail(NDBHIP_ERR_UNSUPPORTED, "more than 2^32 row blocks");
	for (int c = 0; c < nc; c++)
		for (uint32_t j = 0; j < nsub_eff[(size_t) c]; j++)
		{
			const size_t s2 = first[(size_t) c] + j;

			cptr[s2] = sub_gidx[s2] >= 0 ? ix->d_subcent + (size_t) sub_gidx[s2] * dim : lcent + (size_t) c * dim;
		}
	{
		/* the units of ivf_s16_sub_distances_probed: a regrouped list's sublists (those with a centre of their own) sixteen at
		 * a time, eight beyond 1024 dimensions (the tile of centres lives in LDS) */
		std::vector<uint2> units;
		const uint32_t uc = dim > 1024 ? 8u : 16u;

		for (int c = 0; c < nc; c++)
		{
			uint32_t	s2 = first[(size_t) c];
			const uint32_t s3 = first[(size_t) c + 1];

			while (s2 < s3 && sub_gidx[s2] >= 0)
			{
				units.push_back(make_uint2((uint32_t) c, s2));
				s2 += uc;
			}
		}
		if (grow(ix->d_sub_units, ix->d_sub_units_n, units.size() + 1)) return NDBHIP_ERR_HIP;
		if (!units.empty())
			HIP_TRY(hipMemcpyAsync(ix->d_sub_units, units.data(), units.size() * sizeof(uint2), hipMemcpyHostToDevice, g.stream));
		ix->nsub_units = (uint32_t) units.size();
		ix->sub_unit_c = uc;
		HIP_TRY(hipStreamSynchronize(g.stream));		/* (units is a local) */
	}
	if (grow(ix->d_sub_first, ix->d_sub_first_n, (size_t) nc + 1)) return NDBHIP_ERR_HIP;
	if (grow(ix->d_sub_len, ix->d_sub_len_n, nsub)) return NDBHIP_ERR_HIP;
	if (grow(ix->d_sub_loc, ix->d_sub_loc_n, nsub + 1)) return NDBHIP_ERR_HIP;
	if (grow(ix->d_sub_blk, ix->d_sub_blk_n, nsub + 1)) return NDBHIP_ERR_HIP;
	if (grow(ix->d_sub_rad, ix->d_sub_rad_n, nsub)) return NDBHIP_ERR_HIP;
	if (grow(ix->d_sub_gidx, ix->d_sub_gidx_n, nsub)) return NDBHIP_ERR_HIP;
	if (grow(ix->d_sub_cptr, ix->d_sub_cptr_n, nsub)) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemcpyAsync(ix->d_sub_first, first.data(), ((size_t) nc + 1) * 4, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_sub_len, sub_len.data(), nsub * 4, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_sub_loc, sub_loc.data(), (nsub + 1) * 8, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_sub_blk, bo.data(), (nsub + 1) * 4, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_sub_gidx, sub_gidx.data(), nsub * 4, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_sub_cptr, cptr.data(), nsub * sizeof(const float *), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemsetAsync(ix->d_sub_rad, 0, nsub * 4, g.stream));
	hipLaunchKernelGGL(k_s16_sub_radius, dim3((unsigned) ((ix->nrows + 3) / 4)), dim3(256), 0, g.stream, vecs32,
					   ix->nrows, dim, (const int64_t *) ix->d_sub_loc, (int) nsub, (const int64_t *) ix->d_perm,
					   (const float *const *) ix->d_sub_cptr, ix->d_sub_rad);
	/* the centres of the regrouped lists as one list of the matrix-core sweep: planes, norms, exponents */
	{
		const int	rc = s16mat_prepare(ix->dm_sub, ix->d_subcent, (int) ncol, dim);

		if (rc)
			return rc;
	}
	HIP_TRY(hipStreamSynchronize(g.stream));			/* the host tables are locals */
	ix->nsub = (int) nsub;
	ix->nsub_g = (int) ncol;
	ix->s16_sub = true;
	if (g_debug_s16)
		fprintf(stderr, "s16 sublists: %zu lists regrouped into %zu sublists (%zu in all), %llu row blocks\n", kept.size(), nsub_g,
				nsub, (unsigned long long) nb);
	return 0;
}

/* squared distances of every query of the batch to every centre of the regrouped lists, as the matrix-core sweep
 * computes them (k_s16_sweep MODE 3 over the centres' planes; the queries' planes are the batch's): w_subdist
 * [nq][*sstride], each within s16_e(dim, |q|^2, largest centre norm) of the real value. */
static int
ivf_s16_sub_distances(ndbhip_ivf *ix, const float *d_q, int nq, uint32_t *sstride)
{
	const uint32_t st = (uint32_t) ((ix->nsub_g + 63) & ~63);

	(void) d_q;
	if (grow(ix->w_subdist, ix->w_subdist_n, (size_t) nq * st)) return NDBHIP_ERR_HIP;
	*sstride = st;
	return s16mat_run(ix->dm_sub, ix->dim, ix->w_qplanes, ix->w_qn2, ix->w_qexp, ix->w_qthr, nq, ix->w_subdist, st);
}

/*
 * Round 6: the same distances for the centres of each query's PROBED lists only (10M x 768, lists 4096 has
 * 78 k centres — every query x every centre was 0.79 of a 4.3 ms step, on every rank of a sharded search — while a query can
 * use the 32 x ~19 of its probed lists).  List-major, so that a tile of centres is read once for all the queries that probe
 * its list: the batch's (query, probe) pairs bucketed by list (k_lq_count / k_lq_scan / k_lq_fill), then k_subdist_lists —
 * a work item = (a UNIT of <= 16 sublists of one list, 256 of the list's queries); the tile of centres in LDS, a wave per
 * query, every lane the elements lane, lane + 64, ... of (q - c)^2 in fp32, the wave's sum by the xor butterfly:
 * |a - |q - c|^2| <= (dim / 64 + 8) 2^-24 a, far inside the s16_e bound the consumers (k_sub_pairs, the seeds,
 * k_s16c_thr_radius) assume for the matrix-core sweep's values.  Entries of lists a query does not probe are never read
 * (the consumers walk sub_first[L] .. sub_first[L + 1] of probed lists) and are left as they are.
 */
__global__ __launch_bounds__(256) void
k_lq_count(const int *__restrict__ probes, uint32_t n, int nlists, uint32_t *__restrict__ cnt)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;

	if (i < n)
	{
		const int	L = probes[i];

		if (L >= 0 && L < nlists)
			atomicAdd(&cnt[L], 1u);
	}
}

/* one block: off[0 .. nlists] = exclusive sums of cnt (cursor = a copy), then uoff[0 .. nunits] = exclusive sums of the units'
 * work items (ceil(queries of the unit's list / 256)); uoff[nunits + 1] = the work counter, zeroed */
__global__ __launch_bounds__(1024) void
k_lq_scan(const uint32_t *__restrict__ cnt, int nlists, uint32_t *__restrict__ off, uint32_t *__restrict__ cursor,
		  const uint2 *__restrict__ units, uint32_t nunits, uint32_t *__restrict__ uoff)
{
	__shared__ uint32_t s_w[16], s_run;
	const int	tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

	for (int pass = 0; pass < 2; pass++)
	{
		const uint32_t n = pass == 0 ? (uint32_t) nlists : nunits;

		if (tid == 0)
			s_run = 0;
		__syncthreads();
		for (uint32_t i0 = 0; i0 < n; i0 += 1024)
		{
			const uint32_t i = i0 + (uint32_t) tid;
			uint32_t	v = 0;

			if (i < n)
			{
				if (pass == 0)
					v = cnt[i];
				else
				{
					const uint32_t L = units[i].x;

					v = (off[L + 1] - off[L] + 255u) >> 8;
				}
			}
			uint32_t	inc = v;

#pragma unroll
			for (int o = 1; o < 64; o <<= 1)
			{
				const uint32_t t = (uint32_t) __shfl_up((int) inc, o, 64);

				if (lane >= o)
					inc += t;
			}
			if (lane == 63)
				s_w[w] = inc;
			__syncthreads();
			uint32_t	base = s_run;

			for (int k2 = 0; k2 < w; k2++)
				base += s_w[k2];
			if (i < n)
			{
				if (pass == 0)
				{
					off[i] = base + inc - v;
					cursor[i] = base + inc - v;
				}
				else
					uoff[i] = base + inc - v;
			}
			__syncthreads();
			if (tid == 1023)
				s_run = base + inc;
			__syncthreads();
		}
		if (tid == 0)
		{
			if (pass == 0)
				off[nlists] = s_run;
			else
			{
				uoff[nunits] = s_run;
				uoff[nunits + 1] = 0;
			}
		}
		__syncthreads();		/* (pass 1 reads off[]) */
	}
}

__global__ __launch_bounds__(256) void
k_lq_fill(const int *__restrict__ probes, uint32_t n, int npr, int nlists, uint32_t *__restrict__ cursor, uint32_t *__restrict__ lq)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;

	if (i < n)
	{
		const int	L = probes[i];

		if (L >= 0 && L < nlists)
			lq[atomicAdd(&cursor[L], 1u)] = i / (uint32_t) npr;
	}
}

#define SUBD_QREG 32			/* query elements a lane holds: dim <= 2048 */
__global__ __launch_bounds__(256) void
k_subdist_lists(const float *__restrict__ q, int dim, const uint32_t *__restrict__ lq_off, const uint32_t *__restrict__ lq,
				const uint2 *__restrict__ units, uint32_t nunits, uint32_t unit_c, const uint32_t *__restrict__ uoff,
				uint32_t *__restrict__ next, const uint32_t *__restrict__ sub_first, const int *__restrict__ sub_gidx,
				const float *const *__restrict__ sub_cptr, float *__restrict__ subdist, uint32_t sstride)
{
	extern __shared__ __attribute__((aligned(16))) float s_c[];		/* [unit_c][dim] */
	__shared__ uint32_t s_item;
	__shared__ int s_g[16];
	const int	tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const uint32_t total = uoff[nunits];

	for (;;)
	{
		__syncthreads();		/* (the tile and s_item of the item before) */
		if (tid == 0)
			s_item = atomicAdd(next, 1u);
		__syncthreads();
		const uint32_t item = s_item;

		if (item >= total)
			break;
		/* the unit of this item: the last u with uoff[u] <= item */
		uint32_t	lo = 0, hi = nunits;

		while (hi - lo > 1)
		{
			const uint32_t mid = (lo + hi) >> 1;

			if (uoff[mid] <= item)
				lo = mid;
			else
				hi = mid;
		}
		while (lo + 1 < nunits && uoff[lo + 1] <= item)
			lo++;
		const uint2 un = units[lo];
		const uint32_t L = un.x, s0 = un.y, s1 = min(s0 + unit_c, sub_first[L + 1]);
		const uint32_t q0 = lq_off[L] + ((item - uoff[lo]) << 8), q1 = min(q0 + 256u, lq_off[L + 1]);

		/* the tile of centres (the sublists of the unit that have one) */
		if (tid < 16)
			s_g[tid] = s0 + (uint32_t) tid < s1 ? sub_gidx[s0 + tid] : -1;
		for (uint32_t t = 0; s0 + t < s1; t++)
		{
			if (sub_gidx[s0 + t] < 0)		/* uniform */
				continue;
			const float *c = sub_cptr[s0 + t];

			for (int i = tid; i < dim; i += 256)
				s_c[(size_t) t * dim + i] = c[i];
		}
		__syncthreads();
		for (uint32_t qi = q0 + (uint32_t) w; qi < q1; qi += 4)
		{
			const uint32_t qq = lq[qi];
			const float *qv = q + (size_t) qq * dim;
			float		qr[SUBD_QREG];

#pragma unroll
			for (int j = 0; j < SUBD_QREG; j++)
				qr[j] = lane + 64 * j < dim ? qv[lane + 64 * j] : 0.0f;
			for (uint32_t t = 0; s0 + t < s1; t++)
			{
				const int	gi = s_g[t];

				if (gi < 0)				/* uniform */
					continue;
				const float *c = s_c + (size_t) t * dim;
				float		p = 0.0f;

#pragma unroll
				for (int j = 0; j < SUBD_QREG; j++)
					if (lane + 64 * j < dim)
					{
						const float d = qr[j] - c[lane + 64 * j];

						p = p + d * d;
					}
#pragma unroll
				for (int o = 32; o > 0; o >>= 1)
					p = p + __shfl_xor(p, o, 64);
				if (lane == 0)
					subdist[(size_t) qq * sstride + (uint32_t) gi] = p;
			}
		}
	}
}

static int
ivf_s16_sub_distances_probed(ndbhip_ivf *ix, const float *d_q, int nq, const int *w_probes, int npr, uint32_t *sstride)
{
	const uint32_t st = (uint32_t) ((ix->nsub_g + 63) & ~63);
	const int	nl = ix->ncent;
	const uint32_t npairs = (uint32_t) nq * (uint32_t) npr;

	if (ix->dim > 64 * SUBD_QREG)
		return fail(NDBHIP_ERR_UNSUPPORTED, "ivf_s16_sub_distances_probed: dim <= %d", 64 * SUBD_QREG);
	if (grow(ix->w_subdist, ix->w_subdist_n, (size_t) nq * st)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_lqoff, ix->w_lqoff_n, (size_t) 3 * (nl + 1))) return NDBHIP_ERR_HIP;
	if (grow(ix->w_lq, ix->w_lq_n, (size_t) npairs + 1)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_uoff, ix->w_uoff_n, (size_t) ix->nsub_units + 4)) return NDBHIP_ERR_HIP;
	*sstride = st;
	if (ix->nsub_units == 0 || npairs == 0)
		return 0;
	uint32_t   *cnt = ix->w_lqoff, *off = ix->w_lqoff + (nl + 1), *cursor = ix->w_lqoff + 2 * (nl + 1);

	HIP_TRY(hipMemsetAsync(cnt, 0, (size_t) (nl + 1) * 4, g.stream));
	hipLaunchKernelGGL(k_lq_count, dim3((npairs + 255) / 256), dim3(256), 0, g.stream, w_probes, npairs, nl, cnt);
	hipLaunchKernelGGL(k_lq_scan, dim3(1), dim3(1024), 0, g.stream, (const uint32_t *) cnt, nl, off, cursor,
					   (const uint2 *) ix->d_sub_units, ix->nsub_units, ix->w_uoff);
	hipLaunchKernelGGL(k_lq_fill, dim3((npairs + 255) / 256), dim3(256), 0, g.stream, w_probes, npairs, npr, nl, cursor, ix->w_lq);
	const size_t smem = (size_t) ix->sub_unit_c * ix->dim * sizeof(float);

	if (smem > 65536)
	{
		static bool attr = false;

		if (!attr)
		{
			HIP_TRY(hipFuncSetAttribute((const void *) k_subdist_lists, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2048 * 4));
			attr = true;
		}
	}
	hipLaunchKernelGGL(k_subdist_lists, dim3((unsigned) (g.num_cus * (smem > 49152 ? 2 : 3))), dim3(256), smem, g.stream, d_q, ix->dim,
					   (const uint32_t *) off, (const uint32_t *) ix->w_lq, (const uint2 *) ix->d_sub_units, ix->nsub_units,
					   ix->sub_unit_c, (const uint32_t *) ix->w_uoff, ix->w_uoff + ix->nsub_units + 1,
					   (const uint32_t *) ix->d_sub_first, (const int *) ix->d_sub_gidx, (const float *const *) ix->d_sub_cptr,
					   ix->w_subdist, st);
	HIP_TRY(hipGetLastError());
	return 0;
}

/* read the index image back (tests, bench cpu baseline, PostgreSQL page writer) */
extern "C" int
ndbhip_ivf_export(const ndbhip_ivf *cix, float *centroids, int64_t *list_len, float *rows, uint8_t *tids6)
{
	ndbhip_ivf *ix = const_cast<ndbhip_ivf *>(cix);

	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !ix->loaded)
		return fail(NDBHIP_ERR_STATE, "index not loaded");
	if (ivf_flush(ix))
		return NDBHIP_ERR_HIP;
	if (centroids)
		HIP_TRY(hipMemcpy(centroids, ix->d_centroids, (size_t) ix->ncent * ix->dim * 4, hipMemcpyDeviceToHost));
	if (list_len)
		for (int c = 0; c < ix->ncent; c++)
			list_len[c] = ix->own_len[c];
	if (rows && ix->f16)
		return fail(NDBHIP_ERR_UNSUPPORTED, "fp16 mirror: rows are not exported as float4");
	if (rows && ix->nrows > 0)
		HIP_TRY(hipMemcpy(rows, ix->d_vecs, (size_t) ix->nrows * ix->dim * 4, hipMemcpyDeviceToHost));
	if (tids6 && ix->nrows > 0)
	{
		std::vector<uint64_t> t((size_t) ix->nrows);

		HIP_TRY(hipMemcpy(t.data(), ix->d_tids, t.size() * 8, hipMemcpyDeviceToHost));
		for (int64_t r = 0; r < ix->nrows; r++)
			ndb_tid_unpack(t[(size_t) r], tids6 + 6 * r);
	}
	return NDBHIP_OK;
}

/* New index holding only the lists with owned[L] != 0 (device-to-device copy);
 * list lengths stay global so candidate positions are identical on every rank. */
/* New mirror holding positions [lo[c], lo[c] + len[c]) of every list c of `src` (device-to-device copy).
 * Whole lists are the usual shard; a slice lets several ranks share one long, popular list (its candidates keep
 * their positions in the reference's candidates[], so the merged result is unchanged). */
extern "C" int
ndbhip_ivf_shard_slices(const ndbhip_ivf *src, const int64_t *lo, const int64_t *len, const uint8_t *tail,
						ndbhip_ivf **out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!src || !src->loaded || !lo || !len || !out)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	for (int c = 0; c < src->ncent; c++)
		if (len[c] < 0 || lo[c] < 0 ||
			(len[c] > 0 && (lo[c] < src->own_lo[c] || lo[c] + len[c] > src->own_lo[c] + src->own_len[c])))
			return fail(NDBHIP_ERR_INVALID, "slice of list %d is not resident in the source index", c);
	if (!src->pend_list.empty())
		return fail(NDBHIP_ERR_STATE, "source index has pending appends: search or export it first");
	ndbhip_ivf *ix = nullptr;
	int			rc = ndbhip_ivf_create(src->dim, src->nlists, &ix);

	if (rc)
		return rc;
	HIP_TRY(hipMalloc((void **) &ix->d_centroids, (size_t) src->ncent * src->dim * sizeof(float)));
	HIP_TRY(hipMemcpyAsync(ix->d_centroids, src->d_centroids, (size_t) src->ncent * src->dim * sizeof(float),
						   hipMemcpyDeviceToDevice, g.stream));
	ix->ncent = src->ncent;
	ix->meta_nprobe = src->meta_nprobe;
	int64_t		nrows = 0;

	for (int c = 0; c < src->ncent; c++)
		nrows += len[c];
	rc = ivf_set_layout(ix, src->glob_len.data(), tail, nrows, lo, len);
	if (rc)
		return rc;
	const int64_t cap = nrows > 0 ? nrows : 1;
	const size_t esz = src->f16 ? sizeof(uint16_t) : sizeof(float);	/* rows are fp16 images or float4 */

	HIP_TRY(hipMalloc((void **) &ix->d_vecs, (size_t) cap * ix->dim * esz));
	HIP_TRY(hipMalloc((void **) &ix->d_tids, (size_t) cap * sizeof(uint64_t)));
	ix->own_rows = true;
	ix->cap_rows = cap;
	ix->f16 = src->f16;
	for (int c = 0; c < src->ncent; c++)
	{
		const int64_t n = len[c];

		if (n == 0)
			continue;
		const int64_t from = src->loc_off[c] + (lo[c] - src->own_lo[c]);

		HIP_TRY(hipMemcpyAsync((char *) ix->d_vecs + (size_t) ix->loc_off[c] * ix->dim * esz,
							   (const char *) src->d_vecs + (size_t) from * src->dim * esz,
							   (size_t) n * ix->dim * esz, hipMemcpyDeviceToDevice, g.stream));
		HIP_TRY(hipMemcpyAsync(ix->d_tids + ix->loc_off[c], src->d_tids + from, (size_t) n * sizeof(uint64_t),
							   hipMemcpyDeviceToDevice, g.stream));
	}
	HIP_TRY(hipStreamSynchronize(g.stream));
	ix->nrows = nrows;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->loaded = true;
	ix->f16_sub = src->f16_sub;	/* a shard holds a subset of the source's rows */
	*out = ix;
	return NDBHIP_OK;
}

/* the lists with owned[L] != 0, whole */
extern "C" int
ndbhip_ivf_shard(const ndbhip_ivf *src, const uint8_t *owned, ndbhip_ivf **out)
{
	if (!src || !src->loaded || !owned || !out)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	std::vector<int64_t> lo((size_t) src->ncent, 0), len((size_t) src->ncent, 0);

	for (int c = 0; c < src->ncent; c++)
	{
		if (owned[c] && src->own_len[c] != src->glob_len[c])
			return fail(NDBHIP_ERR_INVALID, "list %d is not resident in the source index", c);
		len[(size_t) c] = owned[c] ? src->glob_len[c] : 0;
	}
	return ndbhip_ivf_shard_slices(src, lo.data(), len.data(), owned, out);
}

/* float4 -> IEEE half image.  REF = the reference's float4_to_fp16 (src/types/quantization.c:141-168:
 * mantissa truncated, subnormal results flushed to signed zero, overflow and NaN -> infinity); else
 * round-to-nearest-even (v_cvt_f16_f32). */
template <bool REF>
__global__ __launch_bounds__(256) void
k_rows_to_f16(const float *__restrict__ src, uint16_t *__restrict__ dst, size_t n)
{
	/* grid-stride: a launch carries at most 2^32 - 1 work-items per dimension, a 10M x 1536 mirror has 1.5e10
	 * elements (found by the full-size oracle replay of tests/test_gpu_fullsize.py: with one thread per element
	 * the launch wrapped and left most of the twin unwritten) */
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
	{
		if (REF)
		{
			const uint32_t u = __float_as_uint(src[i]);
			const uint16_t sign = (uint16_t) ((u >> 16) & 0x8000u);
			const int	e = (int) ((u >> 23) & 0xffu) - 127 + 15;

			dst[i] = e <= 0 ? sign : (e >= 31 ? (uint16_t) (sign | 0x7c00u)
									  : (uint16_t) (sign | ((uint32_t) e << 10) | ((u & 0x7fffffu) >> 13)));
		}
		else
			dst[i] = __half_as_ushort(__float2half_rn(src[i]));
	}
}

/* A halfvec twin of a float4 mirror: same centroids, lists and TIDs, rows narrowed on the device. */
extern "C" int
ndbhip_ivf_to_f16(const ndbhip_ivf *src, int reference_encoder, ndbhip_ivf **out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!src || !src->loaded || !out)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (src->f16)
		return fail(NDBHIP_ERR_STATE, "the mirror already holds fp16 rows");
	if (src->dim % 64 != 0)
		return fail(NDBHIP_ERR_UNSUPPORTED, "fp16 rows need dim %% 64 == 0 (dim = %d)", src->dim);
	if (!src->pend_list.empty())
		return fail(NDBHIP_ERR_STATE, "source index has pending appends: search or export it first");
	ndbhip_ivf *ix = nullptr;
	int			rc = ndbhip_ivf_create(src->dim, src->nlists, &ix);

	if (rc)
		return rc;
	HIP_TRY(hipMalloc((void **) &ix->d_centroids, (size_t) src->ncent * src->dim * sizeof(float)));
	HIP_TRY(hipMemcpyAsync(ix->d_centroids, src->d_centroids, (size_t) src->ncent * src->dim * sizeof(float),
						   hipMemcpyDeviceToDevice, g.stream));
	ix->ncent = src->ncent;
	rc = ivf_set_layout(ix, src->glob_len.data(), src->owned.data(), src->nrows, src->own_lo.data(), src->own_len.data());
	if (rc)
		return rc;
	const int64_t cap = src->nrows > 0 ? src->nrows : 1;
	const size_t nel = (size_t) src->nrows * src->dim;

	HIP_TRY(hipMalloc((void **) &ix->d_vecs, (size_t) cap * ix->dim * sizeof(uint16_t)));
	HIP_TRY(hipMalloc((void **) &ix->d_tids, (size_t) cap * sizeof(uint64_t)));
	ix->own_rows = true;
	ix->cap_rows = cap;
	if (nel > 0)
	{
		const dim3	grid((unsigned) std::min<size_t>((nel + 255) / 256, (size_t) 1 << 22));

		if (reference_encoder)
			hipLaunchKernelGGL(k_rows_to_f16<true>, grid, dim3(256), 0, g.stream, (const float *) src->d_vecs,
							   (uint16_t *) ix->d_vecs, nel);
		else
			hipLaunchKernelGGL(k_rows_to_f16<false>, grid, dim3(256), 0, g.stream, (const float *) src->d_vecs,
							   (uint16_t *) ix->d_vecs, nel);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(ix->d_tids, src->d_tids, (size_t) src->nrows * sizeof(uint64_t),
							   hipMemcpyDeviceToDevice, g.stream));
	}
	HIP_TRY(hipStreamSynchronize(g.stream));
	ix->nrows = src->nrows;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->f16 = true;
	ix->loaded = true;
	*out = ix;
	return ivf_note_f16_subnormals(ix);	/* the reference's encoder flushes them; round-to-nearest may not */
}

/* float4_to_fp16 (src/types/quantization.c:141-168: mantissa truncated, subnormal results flushed) for n
 * values from host memory: the GPU vtable's launch_quant_fp16 with the CPU encoder's bits */
extern "C" int
ndbhip_quant_fp16(const float *in, uint16_t *out, int64_t n)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!in || !out || n < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	float	   *d_in = nullptr;
	uint16_t   *d_out = nullptr;

	HIP_TRY(hipMalloc((void **) &d_in, (size_t) n * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_out, (size_t) n * sizeof(uint16_t)));
	HIP_TRY(hipMemcpyAsync(d_in, in, (size_t) n * sizeof(float), hipMemcpyHostToDevice, g.stream));
	hipLaunchKernelGGL(k_rows_to_f16<true>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, g.stream,
					   (const float *) d_in, d_out, (size_t) n);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(out, d_out, (size_t) n * sizeof(uint16_t), hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipFree(d_in));
	HIP_TRY(hipFree(d_out));
	return NDBHIP_OK;
}


/* nprobe as the meta page / reloptions carry it: what ivfrescan reads (ivf_am.c:1487-1513) */
extern "C" int
ndbhip_ivf_get_nprobe(const ndbhip_ivf *ix, int *nprobe)
{
	if (!ix || !nprobe)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	*nprobe = ix->meta_nprobe;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_set_nprobe(ndbhip_ivf *ix, int nprobe)
{
	if (!ix)
		return fail(NDBHIP_ERR_INVALID, "index is NULL");
	ix->meta_nprobe = nprobe;	/* <= 0 is legal on the page: ivfrescan then takes the default (:1512-1513) */
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_shape(const ndbhip_ivf *ix, int *dim, int *nlists)
{
	if (!ix)
		return fail(NDBHIP_ERR_INVALID, "index is NULL");
	if (dim) *dim = ix->dim;
	if (nlists) *nlists = ix->nlists;
	return NDBHIP_OK;
}

/* error text for the PostgreSQL-free page codec (ndbhip_pages.cpp) */
int
ndbhip_pages_fail(int code, const char *msg)
{
	return fail(code, "%s", msg);
}

extern "C" int
ndbhip_ivf_dim(const ndbhip_ivf *ix)
{
	return ix ? ix->dim : fail(NDBHIP_ERR_INVALID, "index is NULL");
}

extern "C" int
ndbhip_ivf_ncentroids(const ndbhip_ivf *ix)
{
	return ix ? ix->ncent : -1;
}


#endif							/* NDBHIP_BUILD_H */
