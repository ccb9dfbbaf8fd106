/*
 * ndbhip_screen16d.h — the DENSE form of the centred sweep (part of ndbhip.hip's translation unit; L2, float4 rows):
 * k_s16c_sweep<8, 2>'s tile — 256 (query, bucket) pairs x 256 rows, 8 waves, a wave owns 64 pairs x 128 rows, one
 * v_mfma_f32_32x32x16_f16 per 16 dimensions and 32 x 32 block — for buckets that are whole lists probed by hundreds of
 * queries (an i.i.d. table: the reference's build rule, src/index/ivf_am.c:580, 2098-2103, leaves 175 non-empty lists
 * there and every query's 32 probes cover four fifths of the table: the batch is a 4096 x 794 k x 768 contraction).
 * Same operands, same items, same records as k_s16c_sweep (ndbhip_screen16c.h); what decides a survivor is still the
 * reference's own arithmetic in k_s16_finalize (ivf_am.c:1561-1568).  What is different is who does what:
 *
 *   - The ring holds ONE chunk (64 KiB: all the LDS there is) in flight per block.  That covers the latency of an L2
 *     hit and not of a miss, and although two thirds of the requests hit — the 32 blocks of an XCD walk 4 row tiles x 8
 *     pair tiles together — a chunk is as late as its latest line and the first block to ask for a line waits for HBM:
 *     measured 7.6 ms per 4096 queries with the operands where they are against 3.7 ms with every request a hit
 *     (profiles/r04_dense_probe.txt), at 2 TB/s of HBM traffic — a quarter of what HBM delivers.  More bytes in flight
 *     is the only cure, and the L2 is where they fit: a PREFETCH touches the lines of the chunk `pfd` steps ahead.
 *   - A wave's vector-memory requests retire in order (s_waitcnt vmcnt counts them in order), so a wave that
 *     prefetches and then waits for its operand chunk waits for its prefetch as well (measured: slower than none).
 *     Hence two kinds of waves: waves 0-3 are the LOADERS — they request the whole chunk (two row blocks and eight pair
 *     pieces each), the members' and the rows' constants (LDS DMA into per-parity arrays), and they alone wait —;
 *     waves 4-7 are the PREFETCHERS: two 4-byte-per-lane LDS DMAs each per chunk (one line per lane: 512 lines =
 *     the chunk; the data lands in a sink nobody reads) and never a wait on the vector-memory counter: nothing else
 *     they do is a vector-memory read (everything per item comes through LDS or the scalar cache).
 *   - Every SIMD runs one loader and one prefetcher; all eight multiply.
 *   - The pair planes are CHUNK-MAJOR here, [64-dim chunk][pair][128 bytes] (k_s16c_qcprep's second layout): the 256
 *     pairs of a tile are one contiguous 32 KiB piece per chunk.  In k_s16c_sweep's [pair][dimp] layout the same bytes are
 *     256 separate 128-byte lines 1536 bytes apart, and that alone made this sweep twice as slow once both operands
 *     came from memory (7.5 against 4.8 ms per 4096 queries, profiles/r04_dense_probe.txt; padding the stride to an odd
 *     number of lines changed nothing: it is the scatter, not the L2's channels).
 */
#ifndef NDBHIP_SCREEN16D_H
#define NDBHIP_SCREEN16D_H

#define S16D_T 256				/* rows per tile = pairs per tile */
#define S16D_BUF 65536			/* a chunk in the ring: 8 row blocks, then 8 pair blocks, 4 KiB each */
#define S16D_QOFF 32768

/* profiling builds (-DNDB_PHASES): 100 MHz clock ticks wave 0 and wave 4 of block 0 spend in the parts of the sweep, summed
 * over the launch: g_phases[32 + 8 w + i], w = 0 (a loader) / 1 (a multiplier), i = 0 waiting for the chunk's DMA, 1 at
 * the chunk's barrier, 2 requesting the next chunk, 3 multiplying, 4 the item's results (pass 0, records), 5 tightening,
 * 6 items */
#ifdef NDB_PHASES
#define S16D_PH_DECL unsigned long long d_ph_t = wall_clock64(), d_ph[7] = {0, 0, 0, 0, 0, 0, 0}
#define S16D_PH(I) do { const unsigned long long d_now = wall_clock64(); d_ph[I] += d_now - d_ph_t; d_ph_t = d_now; } while (0)
#define S16D_PH_FLUSH do { if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) for (int d_i = 0; d_i < 7; d_i++) atomicAdd(&g_phases[32 + 8 * (wave >> 2) + d_i], d_ph[d_i]); } while (0)
#else
#define S16D_PH_DECL ((void) 0)
#define S16D_PH(I) ((void) 0)
#define S16D_PH_FLUSH ((void) 0)
#endif

template <int DBG = 0, int NBL = 4 /* 32-row blocks of the tile's eight that a loader wave multiplies; its SIMD's multiplier takes the rest */,
		  bool SMALL = false /* tiles of <= 128 members take the one-pair-block wave map (a kernel of its own: the third copy of the item
							  * costs the full tiles' path a few spilled registers, 1 % on the i.i.d. table, which has no such tile) */>
__global__ __launch_bounds__(512, 1) void
k_s16c_dense(int dim, int nbuckets, const int64_t *__restrict__ loc_off, const uint32_t *__restrict__ own_len,
			 const unsigned char *__restrict__ planes, const uint32_t *__restrict__ blk_off,
			 const float *__restrict__ rn2, const int16_t *__restrict__ rexp,
			 const unsigned char *__restrict__ qcplanes, size_t qplane /* bytes of one chunk plane of the pair planes */, const float *__restrict__ qcn2,
			 const int *__restrict__ qcexp, const uint32_t *__restrict__ pqid, const uint32_t *__restrict__ pla,
			 const uint32_t *__restrict__ pnrow, float2 *qthr, const uint32_t *__restrict__ cnt,
			 const uint32_t *__restrict__ pair_off, const S16Desc *__restrict__ desc,
			 const uint32_t *__restrict__ runs, unsigned int *__restrict__ ecount, uint2 *__restrict__ erec,
			 float *__restrict__ eub, uint32_t ecap, uint32_t *__restrict__ bmin, int nchunk,
			 uint32_t desc_cap, uint32_t topk, const uint32_t *__restrict__ pos_of, float cE,
			 uint32_t qc_cap, int cosine, int pfd /* chunks the prefetch runs ahead of the operand stream (0: none) */,
			 int rot /* 1: an item's chunks start at a rotation given by its row tile (see `enter`) */,
			 uint32_t tight /* a query's threshold is tightened every time it has emitted this many more records (a power of two) */,
			 unsigned int *__restrict__ xsync = nullptr /* [8][NDB_QHEAD_STRIDE], zeroed: blocks of each XCD that have reached their next meeting */,
			 uint32_t sync_every = 0 /* the blocks of an XCD meet before every this many-th item (0: never) */ )
{
	constexpr int T = S16D_T;
	__shared__ uint32_t s_tn, s_tq[S16_TIGHT_Q], s_tkeys[S16_NB];
	__shared__ __attribute__((aligned(1024))) unsigned char ring[2 * S16D_BUF];
	/* per member / per row of the two items in flight (by the item's parity): DMA'd by the loaders */
	__shared__ __attribute__((aligned(256))) float s_q2[2][T];
	__shared__ __attribute__((aligned(256))) int s_eq[2][T];
	__shared__ __attribute__((aligned(256))) uint32_t s_la[2][T], s_nrow[2][T], s_qid[2][T];
	__shared__ __attribute__((aligned(256))) float s_x2[2][T];
	__shared__ __attribute__((aligned(256))) uint32_t s_por[2][T];
	__shared__ __attribute__((aligned(256))) uint32_t s_exw[2][T];		/* [4 loaders][64 dwords, 32 of them used: 64 int16 exponents] */
	__shared__ float s_t2[T];
	__shared__ float s_nuv[2][T];
	__shared__ uint32_t s_wild[2];
	__shared__ __attribute__((aligned(256))) uint32_t s_sink[64];
	__shared__ uint2 s_hq[8][64];			/* per wave: the elements that stay, until the wave writes their records */
	__shared__ __attribute__((aligned(256))) float s_tf[T];			/* the members' thresholds as the item's first chunk found them */
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const int	wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	wq = wave & 3, wr = wave >> 2;
	const int	r32 = lane & 31, kh = lane >> 5;
	const bool	loader = wave < 4;		/* uniform */
	const int	lw = wave & 3;			/* which of the four loaders / prefetchers */

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

	const uint32_t lane16 = (uint32_t) lane * 16u;
	const uint32_t ring_la = (uint32_t) (uintptr_t) (ndb_lds_ptr) ring;
	const uint32_t mem_la[5] = {(uint32_t) (uintptr_t) (ndb_lds_ptr) &s_q2[0][0], (uint32_t) (uintptr_t) (ndb_lds_ptr) &s_eq[0][0],
		(uint32_t) (uintptr_t) (ndb_lds_ptr) &s_la[0][0], (uint32_t) (uintptr_t) (ndb_lds_ptr) &s_nrow[0][0],
		(uint32_t) (uintptr_t) (ndb_lds_ptr) &s_qid[0][0]};
	const uint32_t row_la[3] = {(uint32_t) (uintptr_t) (ndb_lds_ptr) &s_x2[0][0], (uint32_t) (uintptr_t) (ndb_lds_ptr) &s_por[0][0],
		(uint32_t) (uintptr_t) (ndb_lds_ptr) &s_exw[0][0]};
	const uint32_t sink_la = (uint32_t) (uintptr_t) (ndb_lds_ptr) s_sink;
	const uint32_t tf_la = (uint32_t) (uintptr_t) (ndb_lds_ptr) s_tf;
	constexpr uint32_t PAR = 4u * T;	/* bytes between the two parities of a per-item array */

	/* ---- the stream: the item whose chunks are being requested (loaders) / touched (prefetchers) ---- */
	uint32_t	it_f = it_c, f_c = 0, f_par = 0;
	uint32_t	f_rot = 0;				/* the item's chunks are taken in the order f_rot, f_rot + 1, ... (mod nchunk): see `enter` */
	const unsigned char *sb0 = planes, *sb1 = planes, *sq = qcplanes;	/* loader: its two row blocks, its 64 pair rows; prefetcher: sb0 = its row lines, sq = its pair lines */
	uint32_t	npc = 0;				/* loader: 1 KiB pair pieces that hold a member (0 .. 8) */
	/* loader: a piece = 8 pair rows x 128 bytes; the 16-byte slots are XOR-swizzled at the source (the LDS image is
	 * lane-linear): piece j of the wave holds rows 8 j .. 8 j + 7 of its 64, row rr = 8 (j & 3) + (lane >> 3) of its
	 * 32-row block swizzles by (rr >> 1) & 7 = 4 (j & 1) + (lane >> 4) */
	const uint32_t voff_e = (uint32_t) (lane >> 3) * 128u + 16u * (uint32_t) ((lane & 7) ^ (lane >> 4));
	const uint32_t voff_o = (uint32_t) (lane >> 3) * 128u + 16u * (uint32_t) ((lane & 7) ^ (4 + (lane >> 4)));
	uint32_t	voff_rp = 0, voff_qp = 0;	/* prefetcher: a line per lane */

	auto		enter = [&](uint32_t it, uint32_t par) {
		const S16Desc d = desc[it];			/* uniform address: scalar loads */
		const uint32_t L = d.L, nmem = min((uint32_t) T, cnt[L] - d.qt * T);
		const uint32_t nbk = blk_off[L + 1] - blk_off[L];
		const uint32_t b0 = min(d.t2 * 8u + 2u * (uint32_t) lw, nbk - 1u), b1 = min(d.t2 * 8u + 2u * (uint32_t) lw + 1u, nbk - 1u);
		const uint32_t slot0 = pair_off[L] + d.qt * T;

		/*
		 * The order of an item's chunks is free (one accumulator chain; the error bound of ndbhip_common.h (8) holds for
		 * any order), and it decides what the XCD's L2 keeps.  The 32 blocks of an XCD multiply 4 row tiles x 8 pair tiles
		 * at a time and a block keeps its pair tile for its next item: 3.1 MB of pair planes that are read again and
		 * again, next to 1.5 MB of row planes per item that are read once (by 8 blocks at a time) — 4.7 MB through a
		 * 4 MB L2.  With every block starting at chunk 0 a pair line is touched by its 4 blocks together and then not
		 * for a whole item, by when everything else has gone through the cache: least-recently-used throws out exactly
		 * what comes back (measured: pairs alone or rows alone cost nothing or 0.9 ms, both together 3.8 ms).  With the
		 * start rotated by the row tile (the 4 blocks that share a pair tile have 4 consecutive row tiles), a pair line is
		 * touched every nchunk / 4 steps: it stays, and what the cache drops is the rows' dead lines.
		 */
		f_rot = rot == 1 ? ((d.t2 & 3u) * (uint32_t) nchunk) >> 2 : (rot == 2 ? ((d.qt & 7u) * (uint32_t) nchunk) >> 3 : 0u);
		sb0 = planes + ((size_t) blk_off[L] + b0) * (size_t) nchunk * 4096;
		if (!loader)
		{
			/* lanes 0-31: the 32 lines of block b0's chunk image, lanes 32-63: block b1's; a line per pair row */
			voff_rp = (uint32_t) kh * (b1 - b0) * (uint32_t) nchunk * 4096u + (uint32_t) r32 * 128u;
			voff_qp = min((uint32_t) (64 * lw + lane), nmem - 1u) * 128u;
			sq = qcplanes + (size_t) slot0 * 128;
			return;
		}
		sb1 = planes + ((size_t) blk_off[L] + b1) * (size_t) nchunk * 4096;
		sq = qcplanes + ((size_t) slot0 + 64u * (uint32_t) lw) * 128;
		npc = nmem > 64u * (uint32_t) lw ? min(8u, (nmem - 64u * (uint32_t) lw + 7u) >> 3) : 0u;
		{
			/* the constants of members 64 lw .. 64 lw + 63 (beyond the tile's count: the last member's, never looked at) */
			const uint32_t mo = 4u * min((uint32_t) (64 * lw + lane), nmem - 1u);
			const void *src[5] = {qcn2 + slot0, qcexp + slot0, pla + slot0, pnrow + slot0, pqid + slot0};

#pragma unroll
			for (int a = 0; a < 5; a++)
				s16_dma4(s16_uniform_ptr((const unsigned char *) src[a]), mo, mem_la[a] + par * PAR + (uint32_t) lw * 256u);
		}
		{
			/* ... and of rows 64 lw .. 64 lw + 63 of the tile: |x - c|^2, index in the list, scale exponents (int16: two a
			 * dword; a bucket's first padded plane row is a multiple of 32).  Rows the bucket does not have read its last
			 * row's (never looked at) */
			const uint32_t len = own_len[L];
			const uint32_t r0 = d.t2 * T + 64u * (uint32_t) lw;
			const uint32_t ro = 4u * min(r0 + (uint32_t) lane, len - 1u);
			const uint32_t eo = 4u * min((r0 >> 1) + (uint32_t) r32, (len - 1u) >> 1);
			const size_t g0 = (size_t) loc_off[L];

			s16_dma4(s16_uniform_ptr((const unsigned char *) (rn2 + g0)), ro, row_la[0] + par * PAR + (uint32_t) lw * 256u);
			s16_dma4(s16_uniform_ptr((const unsigned char *) (pos_of + g0)), ro, row_la[1] + par * PAR + (uint32_t) lw * 256u);
			s16_dma4(s16_uniform_ptr((const unsigned char *) (rexp + g0)), eo, row_la[2] + par * PAR + (uint32_t) lw * 256u);
		}
	};
	/* loader: request chunk c of the fetch item into ring buffer bufi */
	auto		issue = [&](uint32_t c, uint32_t bufi) {
		if constexpr (DBG == 1)
			return;
		const uint32_t la = ring_la + bufi * S16D_BUF;
		const bool	rfix = DBG == 2 || DBG == 4, qfix = DBG == 2 || DBG == 3;

		s16_dma_linear<4>(s16_uniform_ptr(rfix ? planes : sb0 + (size_t) c * 4096), lane16, la + (uint32_t) (2 * lw) * 4096u);
		s16_dma_linear<4>(s16_uniform_ptr(rfix ? planes : sb1 + (size_t) c * 4096), lane16, la + (uint32_t) (2 * lw + 1) * 4096u);
#pragma unroll
		for (int j = 0; j < 8; j++)
			if ((uint32_t) j < npc)		/* uniform: a piece without a member is neither fetched nor looked at */
				s16_dma16(s16_uniform_ptr(qfix ? qcplanes : sq + (size_t) c * qplane + (size_t) (1024 * j)),
						  (j & 1) ? voff_o : voff_e, la + S16D_QOFF + (uint32_t) (8 * lw + j) * 1024u);
	};
	/* prefetcher: touch the lines of chunk c of its item */
	auto		touch = [&](uint32_t c) {
		if constexpr (DBG != 0 && DBG != 6 && DBG != 7)
			return;
		s16_dma4(s16_uniform_ptr(sb0 + (size_t) c * 4096), voff_rp, sink_la);
		s16_dma4(s16_uniform_ptr(sq + (size_t) c * qplane), voff_qp, sink_la);
	};
	/* the stream's next chunk (none left: nothing): the loaders request it into the ring, the prefetchers touch it */
	uint32_t	g_f = 0;		/* chunks the stream has handed out so far */
	auto		step = [&]() {
		if (it_f == S16_NOITEM)
			return;
		if (f_c == (uint32_t) nchunk)
		{
			/* the next item, entered only now that its first chunk is due: for the loaders one chunk ahead of the
			 * multiplication, i.e. while the item before it is being multiplied — whose per-item arrays have the other
			 * parity, and whose predecessor (this parity) has been looked at */
			f_c = 0;
			it_f = it_f + stride < run_hi ? it_f + stride : S16_NOITEM;
			f_par ^= 1u;
			if (it_f == S16_NOITEM)
				return;
			enter(it_f, f_par);
		}
		{
			const uint32_t cc = f_c + f_rot < (uint32_t) nchunk ? f_c + f_rot : f_c + f_rot - (uint32_t) nchunk;

			if (loader)
				issue(cc, g_f & 1u);
			else
				touch(cc);
		}
		g_f++;
		f_c++;
	};

	const int	sw = (r32 >> 1) & 7;
	uint32_t	c_par = 0, g_c = 0;		/* parity of the item being multiplied; chunks consumed so far */
	uint32_t	n_item = 0;				/* items this block has begun */
	bool		lonely = false;			/* (wave 0, lane 0) this block has given up waiting for its XCD's others */

	S16D_PH_DECL;
	enter(it_c, 0);
	if (loader)
		step();
	else if (pfd > 0)
		for (int p = 0; p <= pfd; p++)
			step();

	/*
	 * The items, for a wave that multiplies NB of the tile's eight 32-row blocks from block rb0 on (its SIMD's other wave
	 * has the rest; both have the same 64 pairs).  NBL = 4: four and four.  NBL = 3: a loader's 16 requests per chunk cost
	 * it ~1360 cycles in which it issues no matrix instruction, and its sibling's 32 are done after 1024 — with 24 for the
	 * loader and 40 for the multiplier the pipe has work until the loader's requests are out.
	 */
	auto		run = [&](auto nbc0, const int rb00) {
	for (;;)
	{
		/* the item being multiplied: its descriptor again (scalar cache) */
		const S16Desc dc = desc[it_c];
		const uint32_t L = dc.L, t2 = dc.t2;
		const uint32_t nmem_cur = min((uint32_t) T, cnt[L] - dc.qt * T);
		const uint32_t len = own_len[L];

		/*
		 * One item for a wave that has NA pair blocks from block pb0 and NB row blocks from block rb0.  A tile with more than
		 * 128 members: two pair blocks (its SIMD's pair of 64) and the role's row blocks.  A tile with 128 members at most
		 * (SMALL; a bucket probed by a hundred queries — the middle of the sigma sweep — is mostly such tiles): the upper two
		 * of the four 64-pair groups are empty, so every wave takes ONE 32-pair block and four row blocks — the eight
		 * waves cover 4 x 8 blocks, nobody multiplies an empty block, the loaders skip the empty pair pieces as before.
		 */
		auto		item = [&](auto nac, auto nbc, const int pb0, const int rb0) {
		constexpr int NA = decltype(nac)::value, NB = decltype(nbc)::value;
		const int	rfrag = rb0 * 4096 + lane * 16;		/* (fragment-major row images: s16c_unit) */
		const int	qfrag = S16D_QOFF + pb0 * 4096 + r32 * 128;
		ndb_f16acc	acc[NA][NB];

#pragma unroll
		for (int a = 0; a < NA; a++)
#pragma unroll
			for (int b = 0; b < NB; b++)
#pragma unroll
				for (int i = 0; i < 16; i++)
					acc[a][b][i] = 0.0f;

		/* pair blocks of this wave that hold a member (wave-uniform; the empty ones are multiplied all the same: without
		 * the test the k-steps are straight-line code, the next one's ds_reads issued under this one's MFMAs) */
		const int	na = (nmem_cur > (uint32_t) (32 * pb0) ? 1 : 0) + ((NA > 1 && nmem_cur > (uint32_t) (32 * (pb0 + 1))) ? 1 : 0);

		auto		compute = [&](const unsigned char *buf) {
#pragma unroll
			for (int s = 0; s < 4; s++)
			{
				ndb_h8		ah[NA], bh[NB];

#pragma unroll
				for (int a = 0; a < NA; a++)
					ah[a] = *reinterpret_cast<const ndb_h8 *>(buf + qfrag + a * 4096 + (((2 * s + kh) ^ sw) * 16));
#pragma unroll
				for (int b = 0; b < NB; b++)
					bh[b] = *reinterpret_cast<const ndb_h8 *>(buf + rfrag + b * 4096 + s * 1024);
#pragma unroll
				for (int a = 0; a < NA; a++)
#pragma unroll
					for (int b = 0; b < NB; b++)
						acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
			}
		};
		/* one chunk of the stream: (loaders) wait for it, barrier, hand out the next one, multiply */
		auto		chunk = [&]() {
			if (loader)
				s16_wait_vm<0>();
			S16D_PH(0);
			__syncthreads();
			S16D_PH(1);
			if (loader || pfd > 0)
				step();
			S16D_PH(2);
			compute(ring + (g_c & 1u) * S16D_BUF);
			S16D_PH(3);
			g_c++;
		};

		/*
		 * The 32 blocks of an XCD multiply 4 row tiles x 8 pair tiles at a time and a tile's chunk is asked for by the 8 (4)
		 * blocks that share it; it is fetched from memory ONCE only if they ask within the few microseconds a line survives
		 * in the 4 MB L2 under this stream.  Blocks start together and their items take the same time to within a
		 * microsecond or two, but that drifts: every `sync_every` items the blocks of an XCD meet again (a counter in the
		 * XCD's L2; the last generation's participants are the blocks that still have an item).  Wave 0 waits, the others
		 * wait for it at the first chunk's barrier.  Measured on the i.i.d. table (PMC FETCH_SIZE, same box): never 19.1 GB a
		 * launch, every item 9.8 GB (+ 9 % time), every 8th 10.3 GB (+ 1.2 %), every 16th 11.2 GB (no time), every 32nd
		 * 12.6 GB (no time).
		 */
		if (sync_every != 0 && n_item % sync_every == 0 && wave == 0)
		{
			if (lane == 0)
			{
				const uint32_t base = it_c - (blockIdx.x >> 3);		/* the first item of this generation in the run */
				const uint32_t part = min(stride, run_hi - base);
				const uint32_t target = (n_item / sync_every) * stride + part;
				unsigned int *ctr = xsync + xq * NDB_QHEAD_STRIDE;

				atomicAdd(ctr, 1u);
				/* (bounded: two sweeps of different processes — or of two streams with "screen16_sweep_queue" 0 — can each
				 * hold part of the device and wait for blocks that cannot start; a block that has waited some milliseconds
				 * goes on alone and from then on only reports its arrival) */
				if (!lonely)
				{
					int			spins = 0;

					while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < 8192)
					{
						__builtin_amdgcn_s_sleep(2);
						spins++;
					}
					lonely = spins >= 8192;
				}
			}
			__builtin_amdgcn_wave_barrier();
		}
		n_item++;
		chunk();
		/* the members' thresholds as they stand now (in-sweep tightening; a stale value is a valid, looser bound): the one
		 * ordinary vector-memory read of an item, by the loaders' threads, behind the first chunk's barrier (the member
		 * arrays have landed) and looked at after the last chunk */
		if (loader)
			/* (thread = member: the loaders' 256 threads; a gather by LDS DMA like everything else — an ordinary load
			 * would bring the compiler's own waits on the vector-memory counter into paths the prefetchers take too) */
			s16_dma4(s16_uniform_ptr((const unsigned char *) qthr), 8u * s_qid[c_par][tid], tf_la + (uint32_t) lw * 256u);
		/*
		 * What the item's END needs besides the accumulators — the members' operands of pass 0's test instruction, whether
		 * an exponent is outside its range — depends on nothing the chunks compute: the members' and rows' constants landed
		 * with the first chunk, the thresholds with the second.  So from three chunks up it is made behind the SECOND chunk's
		 * barrier, by the multiplier waves' threads (who wait for the loaders at every chunk anyway), and a wave goes from
		 * its last matrix instruction straight into pass 0: no preparation and no block-wide barrier in the part of an item
		 * where the matrix pipe idles (round 6; before, the loaders made it after the last chunk, a barrier behind it).
		 */
		const bool	early = nchunk >= 3;		/* uniform */
		auto		wildcheck = [&]() {
			/* does one of this lane's rows (row 32 (rb0 + b) + r32 of the tile) have an exponent outside pass 0's range */
			bool		wildrow = false;

#pragma unroll
			for (int b = 0; b < NB; b++)
			{
				const int	ri = 32 * (rb0 + b) + r32;
				const int	ex = (int) reinterpret_cast<const int16_t *>(&s_exw[c_par][0])[(ri >> 6) * 128 + (ri & 63)];

				wildrow = wildrow || (t2 * T + (uint32_t) ri < len && (ex < -20 || ex > 20));
			}
			if (wildrow)
				s_wild[c_par] = 1u;
		};
		auto		prep = [&](int tix /* the member */ ) {
			/* what pass 1 subtracts: T rounded up with the slack its fused form needs; and the member's operands of the
			 * test instruction (pass 0 — ndbhip_screen16c.h has the derivation): u = (KB Q2 - TB) 2^(27 - eq) and
			 * v = KB 2^(27 - eq), negated.  A member the tile does not have never emits (u = +inf); a NaN — a norm that
			 * is not a finite fp32, a threshold that is +inf — always does (u = -inf). */
			/* (an index the compiler cannot see through: the kernel runs at the register file's limit, and an LDS address
			 * computed in the prologue for this block would live — in scratch — across the whole sweep) */
			asm volatile("" : "+v"(tix));
			const bool	valid = (uint32_t) tix < nmem_cur;
			const int	eq = s_eq[c_par][tix];
			const float KB = (1.0f - cE) * 0.9999962f;
			const float tfresh = s_tf[tix];
			const float TB = s16_up(tfresh * 1.000004f) + NDB_S16_ABS;
			float		cm = __builtin_fmaf(s_q2[c_par][tix], KB, -TB);

			if (!(cm == cm))
				cm = -__builtin_inff();
			s_t2[tix] = s16_up(tfresh * 1.000001f) + NDB_S16_ABS;
			s_nuv[0][tix] = -ldexpf(KB, 27 - eq);
			s_nuv[1][tix] = valid ? -ldexpf(cm, 27 - eq) : -__builtin_inff();
			if (valid && (eq < -20 || eq > 20))
				s_wild[c_par] = 1u;
		};

		for (int c = 1; c < nchunk; c++)
		{
			chunk();
			if (c == 1 && early)
			{
				/* (behind the second chunk's barrier: the loaders waited for the thresholds' gather in front of it; the third
				 * chunk's barrier, at least, comes before anybody looks) */
				wildcheck();
				if (!loader)
					prep(tid - 256);
			}
		}
		if (nchunk == 1 && loader)
			s16_wait_vm<0>();		/* (otherwise the later chunks' waits have covered the gather) */
		if (!early)
		{
			wildcheck();
			if (loader)
				prep(tid);
		}
		if (tid == 0)
			s_wild[c_par ^ 1u] = 0;		/* read by the item before this one, set next by the item after it */
		if (!early)
			__syncthreads();

		const float K = (1.0f - cE) * 0.99999905f;
		const bool	wild = s_wild[c_par] != 0;		/* uniform */
		/*
		 * The results.  Pass 0 (ndbhip_screen16c.h has the derivation): one v_mfma_f32_32x32x2_f32 per 32 x 32 block leaves
		 * fin = (t1 - (KB (Q2 + X2) - TB)) P in registers of its own, and an element may be left out when fin < 0.  On this
		 * kind of table a few elements in ten thousand stay — but that is one or two per 64 x 64 group of blocks, so a
		 * per-element pass over every group with a hit (k_s16c_sweep's passes 1 and 2) ran for most groups: 2.7 of the
		 * sweep's 7.4 ms.  Here only the elements that stay are touched: the largest of a block's 16 bit patterns per
		 * lane says whether the lane has one (a block without any costs 9 instructions a lane), a lane that has takes them
		 * out one at a time and QUEUES them — (member, row, accumulator) in a 64-entry queue per wave, filled by ballot
		 * rank — and the queue is emptied by the whole wave at once, an entry a lane: validity (the row exists and is no
		 * hole, the member exists, the position is under the candidate cap), the record slot (one returning atomic per
		 * entry, all in flight together), the record.  Every element with fin >= 0 is emitted — a superset, by the margin
		 * of pass 0's slack, of what k_s16c_sweep's fused test keeps —; in an item with an exponent outside pass 0's
		 * range every element is queued and the flush applies that test itself.
		 */
		uint32_t	hq_n = 0;			/* entries in this wave's queue (uniform) */
		auto		flush = [&]() {
			__builtin_amdgcn_wave_barrier();
			if ((uint32_t) lane < hq_n)
			{
				const uint2 h = s_hq[wave][lane];
				const int	m = (int) (h.x & 255u), ri = (int) ((h.x >> 8) & 255u);
				const float accv = __uint_as_float(h.y);
				const int	ex = (int) reinterpret_cast<const int16_t *>(&s_exw[c_par][0])[(ri >> 6) * 128 + (ri & 63)];
				const float x2 = s_x2[c_par][ri];
				const uint32_t por = s_por[c_par][ri];
				const float t1 = ldexpf(accv, s_eq[c_par][m] + ex - 27);
				const float n = s_q2[c_par][m] + x2;
				bool		keep = t2 * T + (uint32_t) ri < len && (uint32_t) m < nmem_cur && por < s_nrow[c_par][m];

				if (wild)
					keep = keep && !(t1 < __builtin_fmaf(n, K, -s_t2[m]));
				if (keep)
				{
					const uint32_t q = s_qid[c_par][m];
					/* (DBG 7: timing only — the records without the returning atomic's round trip) */
					uint32_t	slot = DBG == 7 ? (uint32_t) lane : atomicAdd(&ecount[q], 1u);

					/* (looked at here: a returning atomic still pending at the loop's edge would put the compiler's wait for
					 * the vector-memory counter into every item's first chunk) */
					asm volatile("" : "+v"(slot));
					const float av = n - t1;
					const float er = s16_up(s16_up(cE * n) + NDB_S16_ABS);
					const float lbv = av - er, ubv = s16_up(av + er);
					const float lb = lbv - fabsf(lbv) * 4.8e-7f - 1e-37f;
					const uint32_t pos = s_la[c_par][m] + por, ub_bits = __float_as_uint(ubv);

					if (slot < ecap)
					{
						erec[(size_t) q * ecap + slot] = make_uint2(pos, __float_as_uint(lb));
						eub[(size_t) q * ecap + slot] = ubv;
					}
					/* the smallest upper bound of every hash bucket of positions (kept whether or not the record fit):
					 * k non-empty buckets are k distinct candidates */
					if ((ub_bits & 0x7FFFFFFFu) < 0x7F800000u)
						atomicMin(&bmin[(size_t) q * S16_NB + ((pos * 2654435761u) >> (32 - S16_NB_LOG2))],
								  ndb_key_from_bits(ub_bits));
					if ((slot & (tight - 1u)) == tight - 1u)
					{
						const uint32_t ti = atomicAdd(&s_tn, 1u);

						if (ti < S16_TIGHT_Q)
							s_tq[ti] = q;
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
			hq_n = 0;
		};
		/* pass 0's operands, once per item: the rows' (w, 2^-ex) by row block, the members' (v, u) by pair block */
		float		wbv[NB], uav[NA];
		uint32_t	deadm = 0;			/* bit b: this lane's row of row block b is no row (beyond the bucket, or a hole) */

#pragma unroll
		for (int b = 0; b < NB; b++)
		{
			const int	ri = 32 * (rb0 + b) + r32;
			const int	ex = (int) reinterpret_cast<const int16_t *>(&s_exw[c_par][0])[(ri >> 6) * 128 + (ri & 63)];
			const float x2 = s_x2[c_par][ri];
			const bool	nan = !(x2 == x2);
			const bool	dead = !(t2 * T + (uint32_t) ri < len) || s_por[c_par][ri] == 0xFFFFFFFFu;
			const float w = dead ? __builtin_inff() : (nan ? -__builtin_inff() : ldexpf(x2, -ex));

			wbv[b] = kh ? ldexpf(1.0f, -ex) : w;
			deadm |= dead ? 1u << b : 0u;
		}
#pragma unroll
		for (int a = 0; a < NA; a++)
			uav[a] = s_nuv[kh][32 * (pb0 + a) + r32];
		/* block n of the wave's NA NB: pair block n % NA, row block n / NA.  The test instruction of block n + 1 is issued
		 * before block n's result is looked at (two result windows: the operand fragments' registers are free by now) */
		auto		pass0 = [&](auto nc) {
			constexpr int a = decltype(nc)::value % NA, b = decltype(nc)::value / NA;

			return __builtin_amdgcn_mfma_f32_32x32x2f32(uav[a], wbv[b], acc[a][b], 0, 0, 0);
		};
		auto		look = [&](auto nc, const ndb_f16acc &fin) {
			constexpr int a = decltype(nc)::value % NA, b = decltype(nc)::value / NA;
			const int	ri = 32 * (rb0 + b) + r32;
			int			mx = (int) 0x80000000;

#pragma unroll
			for (int reg = 0; reg < 16; reg++)
				mx = max(mx, __float_as_int(fin[reg]));
			if constexpr (DBG != 0 && DBG != 7)
			{
				if (mx != 0x12345678)
					return;
			}
			if (__ballot(mx >= 0 || wild) == 0ull)
				return;
			/* the elements that stay: one bit each */
			uint32_t	mask = 0;

			if (wild)
				mask = ((deadm >> b) & 1u) ? 0u : 0xFFFFu;
			else
			{
				/* the sixteen sign bits, one v_alignbit_b32 each: mk = (mk << 1) | sign, register 15 first */
				uint32_t	mk = 0;

#pragma unroll
				for (int reg = 15; reg >= 0; reg--)
					mk = __builtin_amdgcn_alignbit(mk, __float_as_uint(fin[reg]), 31);
				mask = ~mk & 0xFFFFu;
			}
			for (;;)
			{
				const unsigned long long any = __ballot(mask != 0);

				if (any == 0ull)
					break;
				/* the register of the first lane that has one (uniform: the accumulator is read by a relative move, no chain
				 * of selects), and every lane that has an element in that register */
				const int	reg = __builtin_amdgcn_readfirstlane(__ffs((int) __builtin_amdgcn_readlane((int) mask, __ffsll((long long) any) - 1)) - 1);
				const bool	has = (mask >> reg) & 1u;
				const unsigned long long bal = __ballot(has);
				const uint32_t cnt = (uint32_t) __popcll(bal);

				if (hq_n + cnt > 64u)
					flush();
				const float v = acc[a][b][reg];

				if (has)
				{
					const int	m = 32 * (pb0 + a) + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
					const uint32_t idx = hq_n + (uint32_t) __popcll(bal & ((1ull << lane) - 1ull));

					mask &= ~(1u << reg);
					s_hq[wave][idx] = make_uint2((uint32_t) m | ((uint32_t) ri << 8), __float_as_uint(v));
				}
				hq_n += cnt;
			}
		};
		{
			ndb_f16acc	fin0, fin1;

			/* (a = 1 blocks of a tile with <= 32 members here are not looked at: uniform) */
#define S16D_P0(N, F) do { if constexpr ((N) < NA * NB) { if (((N) % NA) < na) F = pass0(std::integral_constant<int, (N)>{}); } } while (0)
#define S16D_LK(N, F) do { if constexpr ((N) < NA * NB) { if (((N) % NA) < na) look(std::integral_constant<int, (N)>{}, F); } } while (0)
			if constexpr (NB <= 4)
			{
				S16D_P0(0, fin0);
				S16D_P0(1, fin1); S16D_LK(0, fin0);
				S16D_P0(2, fin0); S16D_LK(1, fin1);
				S16D_P0(3, fin1); S16D_LK(2, fin0);
				S16D_P0(4, fin0); S16D_LK(3, fin1);
				S16D_P0(5, fin1); S16D_LK(4, fin0);
				S16D_P0(6, fin0); S16D_LK(5, fin1);
				S16D_P0(7, fin1); S16D_LK(6, fin0);
				S16D_LK(7, fin1);
			}
			else
			{
				/* (five row blocks: 160 accumulator registers leave room for one window) */
				S16D_P0(0, fin0); S16D_LK(0, fin0); S16D_P0(1, fin0); S16D_LK(1, fin0);
				S16D_P0(2, fin0); S16D_LK(2, fin0); S16D_P0(3, fin0); S16D_LK(3, fin0);
				S16D_P0(4, fin0); S16D_LK(4, fin0); S16D_P0(5, fin0); S16D_LK(5, fin0);
				S16D_P0(6, fin0); S16D_LK(6, fin0); S16D_P0(7, fin0); S16D_LK(7, fin0);
				S16D_P0(8, fin0); S16D_LK(8, fin0); S16D_P0(9, fin0); S16D_LK(9, fin0);
			}
#undef S16D_P0
#undef S16D_LK
		}
		if (hq_n != 0)
			flush();
		S16D_PH(4);
#ifdef NDB_PHASES
		d_ph[6]++;
#endif
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
						const float nt = cosine ? s16c_cos_t_from_ub(__uint_as_float(tb), dim) : s16c_t_from_ub(__uint_as_float(tb), dim);

						/* T >= 0 (or +inf): its bits order like the values */
						atomicMin(reinterpret_cast<unsigned int *>(&qthr[q].x), __float_as_uint(nt));
					}
				}
				__syncthreads();
			}
			if (tid == 0 && s_tn != 0)
				s_tn = 0;
		}
		S16D_PH(5);
		};		/* item */
		if constexpr (SMALL)
		{
			if (nmem_cur <= 128u)
				item(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, wave & 3, 4 * (wave >> 2));
			else
				item(std::integral_constant<int, 2>{}, nbc0, 2 * wq, rb00);
		}
		else
			item(std::integral_constant<int, 2>{}, nbc0, 2 * wq, rb00);
		it_c += stride;
		if (it_c >= run_hi)
			break;
		c_par ^= 1u;
		/* (no barrier: the next item's first chunk starts with one, and the per-item arrays of parity c_par ^ 1 are
		 * requested again only by an `enter` behind that barrier) */
	}
	};
	if constexpr (NBL == 4)
		run(std::integral_constant<int, 4>{}, 4 * wr);
	else if (loader)
		run(std::integral_constant<int, NBL>{}, 0);
	else
		run(std::integral_constant<int, 8 - NBL>{}, NBL);
	S16D_PH_FLUSH;
}

#endif							/* NDBHIP_SCREEN16D_H */
