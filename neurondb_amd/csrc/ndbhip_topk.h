/*
 * ndbhip_topk.h — k_ivf_topk / k_merge_topk (part of ndbhip.hip's translation unit): the k smallest of a query's
 * candidate distances in the order the reference's selection sort returns them (ivf_am.c:1856-1899), per query,
 * per position range or per shard, and the replay merge of such partial results.  The block-level primitives
 * they stand on are in ndbhip_internal.h.
 */
#ifndef NDBHIP_TOPK_H
#define NDBHIP_TOPK_H

/*
 * Top-k of one query's candidate distances, reproducing ivf_am.c:1856-1899.
 * One block (256 threads) per query.
 *
 * Fast path (k <= 64), two streaming passes and no histogram:
 *   1. every thread keeps the minimum key of its strided share; the k-th smallest of
 *      the 256 thread minima is an upper bound U of the k-th smallest candidate
 *      (the k smallest minima are k distinct candidates <= U);
 *   2. every candidate with key <= U is gathered (a superset of "everything <= T");
 *      block_sort_cut trims it to the tie-complete subset and the replay finishes.
 *   If more than NDB_TOPK_FAST_CAP candidates are <= U (massive ties) the radix
 *   path below is used instead.
 * Radix path: 4-pass LDS-histogram select + ordered compaction (any k, any ties).
 *
 * partial != 0: emit the tie-complete subset for the shard merge instead of results.
 */
/* the k-th smallest of the 256 threads' values mn (all threads get it; `otherwise` when fewer than k threads hold
 * one): the thread whose (value, thread) pair has rank k - 1 publishes it — no sort.  Ends with a barrier. */
static __device__ uint32_t
block_kth_of_minima(uint64_t *comp, uint32_t *slot, uint32_t mn, uint32_t k, uint32_t nth, uint32_t otherwise)
{
	const uint32_t tid = threadIdx.x;
	const uint64_t mine = ((uint64_t) mn << 32) | tid;

	comp[tid] = mine;
	__syncthreads();
	if (nth >= k && lds_rank_u64(comp, 256, mine) == k - 1)
		*slot = mn;
	__syncthreads();
	return nth >= k ? *slot : otherwise;
}

__global__ __launch_bounds__(256) void
k_ivf_topk(IvfDev ix, const int *__restrict__ probes, const uint32_t *__restrict__ cand_off,
		   const uint32_t *__restrict__ loc_cand_off, int npr, const float *__restrict__ dist, uint32_t stride, uint32_t k, int partial,
		   ndbhip_cand *__restrict__ out_cand, int *__restrict__ out_ncand, int64_t *__restrict__ out_total,
		   uint64_t *__restrict__ out_tids, float *__restrict__ out_dist, int *__restrict__ out_count,
		   uint32_t nq, const uint32_t *__restrict__ tmin, uint32_t tstride)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	const uint32_t ecap = topk_entry_cap(k);
	TopkSmem	s = carve_topk_smem(smem_raw, ecap, k);
	const uint32_t q = blockIdx.x;
	const uint32_t tid = threadIdx.x;
	const uint32_t *co = cand_off + (size_t) q * (npr + 1);		/* positions in the reference's candidates[] */
	const uint32_t *lco = loc_cand_off + (size_t) q * (npr + 1);	/* positions among the rows held here */
	const uint32_t gtotal = co[npr];
	/*
	 * gridDim.y > 1 (partial mode only): a query's candidates are cut into gridDim.y position ranges, one
	 * block each — for small batches one block per query cannot keep enough loads in flight (a single
	 * query: 130 k candidates in 129 us).  Every range emits its tie-complete subset exactly like a rank of
	 * a sharded search does, and k_merge_topk replays the union; rec. layout [(range * nq + q) * 3k + j].
	 */
	const uint32_t all = lco[npr];
	const uint32_t per = (all + gridDim.y - 1) / gridDim.y;
	const uint32_t lo = min(all, blockIdx.y * per);
	const uint32_t total = min(all, lo + per) - lo;
	const float *d = dist + (size_t) q * stride + lo;
	const size_t oq = (size_t) blockIdx.y * nq + q;
	uint32_t	ns = 0;
	bool		have = false;

	auto		ld = [&](uint32_t i, uint32_t &bits) -> bool {
		bits = __float_as_uint(d[i]);
		return true;
	};
	/* local position (inside this block's range) -> (TID, position in candidates[]) */
	/* the fast paths search a copy of the probes' offsets in LDS (the histogram area, which only the radix path uses):
	 * the binary search is five dependent reads, and from memory each of them is a trip to L2 */
	const uint32_t *lcs = lco;
	auto		tid_of = [&](uint32_t i0, uint32_t &gpos) -> uint64_t {
		const uint32_t i = i0 + lo;
		const uint32_t p = find_probe(lcs, npr, i);
		const int	L = probes[(size_t) q * npr + p];
		const uint32_t at = i - lcs[p];

		gpos = co[p] + ix.own_lo[L] + at;	/* a split list: this mirror starts at position own_lo */
		return ix.tids[ix.loc_off[L] + at];
	};

	if (npr + 1 <= 256 && k <= NDB_TOPK_FAST_MAXK && ecap == NDB_TOPK_FAST_CAP)
	{
		for (uint32_t i = tid; i <= (uint32_t) npr; i += 256)
			s.hist[i] = lco[i];
		lcs = s.hist;			/* (every use is behind a barrier of the fast paths) */
	}

	if (tmin && gridDim.y == 1 && k <= NDB_TOPK_FAST_MAXK && ecap == NDB_TOPK_FAST_CAP)
	{
		/*
		 * Tile path (the grouped scan left the smallest key of every 64-candidate tile): the k-th smallest
		 * of the thread minima over TILE minima bounds the k-th candidate just like the minima over
		 * candidates do, and a tile whose minimum is above the bound holds nothing to gather — so the
		 * distance buffer is only read where it matters (a few tiles of 256 B instead of all of it twice).
		 */
		const uint32_t *tm = tmin + (size_t) q * tstride;
		const uint32_t nslots = min(tstride, (all >> 6) + (uint32_t) npr + 1u);
		uint32_t	mn = 0xFFFFFFFFu;

		for (uint32_t sidx = tid; sidx < nslots; sidx += 256)
			mn = min(mn, tm[sidx]);
		const uint32_t nth = (uint32_t) __syncthreads_count(mn != 0xFFFFFFFFu);

		const uint32_t U = block_kth_of_minima(s.fs.comp, s.sh + 8, mn, k, nth, 0xFFFFFFFEu);	/* 0xFFFFFFFF = empty slot */
		uint32_t   *tlist = s.fs.curpos;	/* tiles to open (curpos is replay scratch, free until then) */

		__syncthreads();
		if (tid == 0)
		{
			s.sh[0] = 0;		/* gathered candidates */
			s.sh[1] = 0;		/* tiles to open */
		}
		__syncthreads();
		for (uint32_t sidx = tid; sidx < nslots; sidx += 256)
			if (tm[sidx] <= U)
			{
				const uint32_t at = atomicAdd(&s.sh[1], 1u);

				if (at < NDB_TOPK_FAST_CAP)
					tlist[at] = sidx;
			}
		__syncthreads();
		const uint32_t ntl = s.sh[1];

		if (ntl <= NDB_TOPK_FAST_CAP)
		{
			const uint32_t lane = tid & 63u, wave = tid >> 6;

			for (uint32_t ti = wave; ti < ntl; ti += 4)
			{
				const uint32_t sidx = tlist[ti];
				/* slot -> (probe, tile): the largest p with (lco[p] >> 6) + p <= slot */
				uint32_t	lo2 = 0, hi2 = (uint32_t) npr;

				while (hi2 - lo2 > 1)
				{
					const uint32_t mid = (lo2 + hi2) >> 1;

					if ((lcs[mid] >> 6) + mid <= sidx)
						lo2 = mid;
					else
						hi2 = mid;
				}
				const uint32_t base = lcs[lo2] + ((sidx - ((lcs[lo2] >> 6) + lo2)) << 6);
				const uint32_t i = base + lane;

				if (base < lcs[lo2 + 1] && i < lcs[lo2 + 1])
				{
					const uint32_t b0 = __float_as_uint(d[i]);

					if (ndb_key_from_bits(b0) <= U)
					{
						const uint32_t slot = atomicAdd(&s.sh[0], 1u);

						if (slot < NDB_TOPK_FAST_CAP)
						{
							s.e_bits[slot] = b0;
							s.e_pos[slot] = i;
						}
					}
				}
			}
			__syncthreads();
			const uint32_t got = s.sh[0];

			__syncthreads();
			if (got <= NDB_TOPK_FAST_CAP)
			{
				ns = got;
				have = true;
				for (uint32_t j = tid; j < ns; j += 256)
				{
					uint32_t	gpos;

					s.e_id[j] = tid_of(s.e_pos[j], gpos);
					s.e_pos[j] = gpos;
				}
				__syncthreads();
			}
		}
		__syncthreads();
	}
	else if (k <= NDB_TOPK_FAST_MAXK && ecap == NDB_TOPK_FAST_CAP)
	{
		/* pass 1: thread minima (4 independent loads in flight per thread) */
		uint32_t	mn = 0xFFFFFFFFu;
		uint32_t	nvalid = 0;
		uint32_t	i = tid;

		for (; i + 3 * 256 < total; i += 4 * 256)
		{
			const uint32_t b0 = __float_as_uint(d[i]), b1 = __float_as_uint(d[i + 256]);
			const uint32_t b2 = __float_as_uint(d[i + 512]), b3 = __float_as_uint(d[i + 768]);

			mn = min(min(mn, ndb_key_from_bits(b0)), min(ndb_key_from_bits(b1), min(ndb_key_from_bits(b2), ndb_key_from_bits(b3))));
			nvalid += 4;
		}
		for (; i < total; i += 256)
		{
			const uint32_t b0 = __float_as_uint(d[i]);

			mn = min(mn, ndb_key_from_bits(b0));
			nvalid++;
		}
		/* sort the 256 minima; threads without a candidate carry 0xFFFFFFFF and sort last */
		const uint32_t nth = (uint32_t) __syncthreads_count(nvalid > 0);

		/* U: the k-th smallest thread minimum bounds the k-th smallest candidate (the k smallest
		 * minima are k distinct candidates <= U); with fewer than k non-empty threads gather all */
		const uint32_t U = block_kth_of_minima(s.fs.comp, s.sh + 8, mn, k, nth, 0xFFFFFFFFu);

		/* pass 2: gather every candidate with key <= U */
		if (tid == 0)
			s.sh[0] = 0;
		__syncthreads();
		auto		keep = [&](uint32_t b0, uint32_t at) {
			if (ndb_key_from_bits(b0) <= U)
			{
				const uint32_t slot = atomicAdd(&s.sh[0], 1u);

				if (slot < NDB_TOPK_FAST_CAP)
				{
					s.e_bits[slot] = b0;
					s.e_pos[slot] = at;
				}
			}
		};
		/* (four loads in flight per thread, like pass 1: with the LDS atomic between them the loads would go out one by one) */
		for (i = tid; i + 3 * 256 < total; i += 4 * 256)
		{
			const uint32_t b0 = __float_as_uint(d[i]), b1 = __float_as_uint(d[i + 256]);
			const uint32_t b2 = __float_as_uint(d[i + 512]), b3 = __float_as_uint(d[i + 768]);

			keep(b0, i);
			keep(b1, i + 256);
			keep(b2, i + 512);
			keep(b3, i + 768);
		}
		for (; i < total; i += 256)
			keep(__float_as_uint(d[i]), i);
		__syncthreads();
		const uint32_t got = s.sh[0];

		__syncthreads();
		if (got <= NDB_TOPK_FAST_CAP)
		{
			ns = got;
			have = true;
			for (uint32_t j = tid; j < ns; j += 256)
			{
				uint32_t	gpos;

				s.e_id[j] = tid_of(s.e_pos[j], gpos);
				s.e_pos[j] = gpos;
			}
			__syncthreads();
		}
	}

	if (!have)
	{
		uint32_t	T, m_less, kk0, cnt_eq;

		lcs = lco;

		block_radix_select(ld, total, k, s.hist, s.sh, T, m_less, kk0, cnt_eq);
		const uint32_t n_eq = cnt_eq < 2 * k ? cnt_eq : 2 * k;

		ns = (kk0 > 0) ? (m_less + n_eq) : 0;
		if (kk0 > 0)
		{
			auto		emit = [&](int cls, uint32_t rank, uint32_t i, uint32_t bits) {
				const uint32_t slot = cls ? (m_less + rank) : rank;

				uint32_t	gpos;

				s.e_bits[slot] = bits;
				s.e_id[slot] = tid_of(i, gpos);
				s.e_pos[slot] = gpos;
			};
			block_ordered_gather(ld, total, T, n_eq, s.sh, emit);
		}
		__syncthreads();
	}

	/* number of candidates this rank holds = what bounds kk locally; globally `total` */
	uint32_t	kk;
	const uint32_t npad = next_pow2(ns > 0 ? ns : 1);
	const uint32_t cut = block_sort_cut(s.e_bits, s.e_pos, ns, npad, k, partial ? (uint64_t) ns : (uint64_t) gtotal,
										s.fs, kk);

	if (partial)
	{
		for (uint32_t j = tid; j < cut; j += blockDim.x)
		{
			const uint32_t e = s.fs.perm[j];
			ndbhip_cand c;

			c.key = s.e_bits[e];	/* raw float4 bits; the merge derives the order key */
			c.pos = s.e_pos[e];
			c.tid = s.e_id[e];
			out_cand[oq * (3 * k) + j] = c;
		}
		if (tid == 0)
		{
			out_ncand[oq] = (int) cut;
			out_total[q] = (int64_t) gtotal;
		}
		return;
	}
	block_replay_emit(s.e_bits, s.e_id, cut, kk, s.fs, out_tids + (size_t) q * k, out_dist + (size_t) q * k,
					  out_count + q);
}

/*
 * Shard merge: union of the ranks' partial records for one query, then the
 * same replay.  cand[(w * nq + q) * cap + j], ncand[w * nq + q].
 */
__global__ __launch_bounds__(256) void
k_merge_topk(const ndbhip_cand *__restrict__ cand, const int *__restrict__ ncand,
			 const int64_t *__restrict__ total, int world, int nq, uint32_t k, uint32_t cap,
			 uint64_t *__restrict__ out_tids, float *__restrict__ out_dist, int *__restrict__ out_count)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	const uint32_t capall = cap * (uint32_t) world;
	TopkSmem	s = carve_topk_smem(smem_raw, capall, k);
	const uint32_t q = blockIdx.x;
	uint32_t   *woff = s.hist;		/* 65 words: the histogram area is unused in the merge */

	/* (the counts are read by as many threads as there are ranks or ranges: one thread reading them in turn pays the
	 * memory latency `world` times — 16 ranges were ~10 us of a single query) */
	uint32_t   *wcnt = s.hist + 80;

	for (int w0 = 0; w0 < world; w0 += 64)
	{
		const int	w = w0 + (int) threadIdx.x;

		__syncthreads();
		if (threadIdx.x < 64 && w < world)
		{
			/* a count from a peer is data, not a promise: more than `cap` records (or a negative count) would
			 * overrun the LDS arrays sized for world x cap */
			const int	nc_w = ncand[(size_t) w * nq + q];

			wcnt[threadIdx.x] = (uint32_t) (nc_w < 0 ? 0 : (nc_w > (int) cap ? (int) cap : nc_w));
		}
		__syncthreads();
		if (threadIdx.x == 0)
		{
			uint32_t	acc = w0 ? woff[w0] : 0;

			for (int i = 0; i < 64 && w0 + i < world; i++)
			{
				woff[w0 + i] = acc;
				acc += wcnt[i];
			}
			woff[w0 + (world - w0 < 64 ? world - w0 : 64)] = acc;
		}
	}
	__syncthreads();
	const uint32_t n = woff[world];

	/* (one flat loop over ranks x records: a loop per rank would wait for memory once per rank) */
	for (uint32_t idx = threadIdx.x; idx < capall; idx += blockDim.x)
	{
		const uint32_t w = idx / cap, j = idx - w * cap;

		if (j < woff[w + 1] - woff[w])
		{
			const ndbhip_cand c = cand[((size_t) w * nq + q) * cap + j];

			s.e_bits[woff[w] + j] = c.key;
			s.e_pos[woff[w] + j] = c.pos;
			s.e_id[woff[w] + j] = c.tid;
		}
	}
	__syncthreads();
	block_finalize_topk(s.e_bits, s.e_pos, s.e_id, n, next_pow2(n > 0 ? n : 1), k,
						(uint64_t) total[q], s.fs,
						out_tids + (size_t) q * k, out_dist + (size_t) q * k, out_count + q);
}


static int
set_kernel_attributes()
{
	HIP_TRY(hipFuncSetAttribute((const void *) k_ivf_topk, hipFuncAttributeMaxDynamicSharedMemorySize,
								NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_merge_topk, hipFuncAttributeMaxDynamicSharedMemorySize,
								NDB_TOPK_MAX_SMEM));
	return set_kernel_attributes_build();
	return NDBHIP_OK;
}

#endif							/* NDBHIP_TOPK_H */
