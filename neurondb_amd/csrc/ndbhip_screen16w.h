/*
 * ndbhip_screen16w.h — the centred sweep for SPARSE pair tables (a bucket is probed by a handful of queries: clustered
 * tables, where the triangle inequality leaves ~1 % of the (query, sublist) pairs) as WAVE-AUTONOMOUS REGISTER STREAMS
 * (round 5; part of ndbhip.hip's translation unit).  Same job, same arguments, same records and the same per-element
 * arithmetic as k_s16c_sweep<1, NBUF> (ndbhip_screen16c.h; bounds of ivfCollectCandidates' distances,
 * src/index/ivf_am.c:1722-1909 — the values still come from Acc<R> in k_s16_finalize), another way through the machine:
 *
 *   - That regime is bound by the rows' bytes (each 32-row block image is read once per batch, 4 matrix instructions
 *     per 4 KiB), i.e. by how many bytes the device keeps in flight.  The LDS ring of k_s16c_sweep<1, 3> holds 2 chunks
 *     x 20 KiB x 2 blocks = 80 KiB a compute unit in flight at best, and every item boundary drained it (the compiler
 *     puts s_waitcnt vmcnt(0) in front of the item's ordinary loads, and the vector-memory counter retires in order):
 *     3.8 TB/s, 0.48 of the roof.
 *   - Here nothing goes through LDS.  The row planes are fragment-major (s16c_unit): lane l of a wave finds its operand of
 *     k-step s at image + 1024 s + 16 l, so a k-step's 1 KiB is ONE coalesced global_load_dwordx4 into the very registers
 *     v_mfma_f32_32x32x16_f16 reads; the pair operand (a few members, rows of the pair planes in natural order) is
 *     16 bytes per lane from member min(lane & 31, nmem - 1)'s row.  A wave owns one 32-row block x 32 pairs, keeps D
 *     chunks (D x 8 loads) in flight in its registers and never meets another wave: no barrier, no ring in LDS.
 *     8 waves a compute unit x 5 chunks x 4 KiB of rows = 160 KiB of row bytes in flight, twice the ring's.
 *   - The stream never stops at an item's end: the slot of the chunk just multiplied takes the stream's chunk D further
 *     on, which is the NEXT item's once this one has none left (descriptor through the scalar cache).
 *   - Work items are k_s16_items' (bucket, 128-row tile, 32-pair tile); wave w of a block takes 32-row block w of the
 *     tile and skips tiles that have no such block (the LDS kernel re-read the tile's last block instead).
 *
 * WHY THE STREAM IS INLINE ASM WITH REGISTERS OF ITS OWN.  Written as plain loads into arrays the compiler software-
 * pipelines only by accident: one conditional request anywhere in the loop and its waitcnt pass falls back to
 * s_waitcnt vmcnt(0) at every use (measured on three formulations; a flat_load sneaking in does the same), and values
 * loaded next to a full ring get spilled to scratch the moment they arrive (scratch traffic retires through the same
 * in-order counter).  So the kernel is compiled with amdgpu_num_vgpr(88): the compiler owns v0-v87 and NEVER touches
 * v88-v255 (they are reserved registers to it); the stream's loads, waits and matrix instructions are inline asm on
 *     v88-v94   the item's constants in flight (|x - c|^2, M^2 - |x|^2, position, |q - c|^2, query, the two exponents)
 *     v95       the member's threshold in flight
 *     v96 + 32 j + 4 s ..  (rows: operand B)  and  v112 + 32 j + 4 s ..  (pairs: operand A)   of slot j, k-step s
 * and every wait is `s_waitcnt vmcnt(n)` with n = the stream's own requests issued after the one needed.  That is
 * safe whatever else the compiler has in flight (the rare emission path's loads, atomics, stores): vector-memory loads
 * retire in order, so other requests among the newest n only make the wait longer, never shorter — the count assumes
 * nothing that was not issued.  The accumulators are an ordinary variable ("+v"): the compiler does not know the asm
 * wrote them with matrix instructions, so the 18 wait states between the last one and the first ordinary read are
 * written out (s_nop) where the item's products are complete.
 */
#ifndef NDBHIP_SCREEN16W_H
#define NDBHIP_SCREEN16W_H

#define S16W_CVGPR 88			/* the compiler's registers: v0 .. v87 */
#define S16W_MAXD 5				/* slots: v96 .. v255 */

/* ---- generated (tools/gen_s16w_asm.py): slot j's eight requests (rows at base + voff, pairs at qbase + pvoff; 16 bytes a
 * lane each) and its four matrix instructions ---- */
#define S16W_LD0(vo, rb, pv, qb) asm volatile("global_load_dwordx4 v[96:99], %0, %1\n\t" "global_load_dwordx4 v[112:115], %2, %3\n\t" "global_load_dwordx4 v[100:103], %0, %1 offset:1024\n\t" "global_load_dwordx4 v[116:119], %2, %3 offset:32\n\t" "global_load_dwordx4 v[104:107], %0, %1 offset:2048\n\t" "global_load_dwordx4 v[120:123], %2, %3 offset:64\n\t" "global_load_dwordx4 v[108:111], %0, %1 offset:3072\n\t" "global_load_dwordx4 v[124:127], %2, %3 offset:96" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_MM0(acc) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[112:115], v[96:99], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[116:119], v[100:103], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[120:123], v[104:107], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[124:127], v[108:111], %0" : "+v"(acc))
#define S16W_LD1(vo, rb, pv, qb) asm volatile("global_load_dwordx4 v[128:131], %0, %1\n\t" "global_load_dwordx4 v[144:147], %2, %3\n\t" "global_load_dwordx4 v[132:135], %0, %1 offset:1024\n\t" "global_load_dwordx4 v[148:151], %2, %3 offset:32\n\t" "global_load_dwordx4 v[136:139], %0, %1 offset:2048\n\t" "global_load_dwordx4 v[152:155], %2, %3 offset:64\n\t" "global_load_dwordx4 v[140:143], %0, %1 offset:3072\n\t" "global_load_dwordx4 v[156:159], %2, %3 offset:96" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_MM1(acc) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[144:147], v[128:131], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[148:151], v[132:135], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[152:155], v[136:139], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[156:159], v[140:143], %0" : "+v"(acc))
#define S16W_LD2(vo, rb, pv, qb) asm volatile("global_load_dwordx4 v[160:163], %0, %1\n\t" "global_load_dwordx4 v[176:179], %2, %3\n\t" "global_load_dwordx4 v[164:167], %0, %1 offset:1024\n\t" "global_load_dwordx4 v[180:183], %2, %3 offset:32\n\t" "global_load_dwordx4 v[168:171], %0, %1 offset:2048\n\t" "global_load_dwordx4 v[184:187], %2, %3 offset:64\n\t" "global_load_dwordx4 v[172:175], %0, %1 offset:3072\n\t" "global_load_dwordx4 v[188:191], %2, %3 offset:96" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_MM2(acc) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[176:179], v[160:163], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[180:183], v[164:167], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[184:187], v[168:171], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[188:191], v[172:175], %0" : "+v"(acc))
#define S16W_LD3(vo, rb, pv, qb) asm volatile("global_load_dwordx4 v[192:195], %0, %1\n\t" "global_load_dwordx4 v[208:211], %2, %3\n\t" "global_load_dwordx4 v[196:199], %0, %1 offset:1024\n\t" "global_load_dwordx4 v[212:215], %2, %3 offset:32\n\t" "global_load_dwordx4 v[200:203], %0, %1 offset:2048\n\t" "global_load_dwordx4 v[216:219], %2, %3 offset:64\n\t" "global_load_dwordx4 v[204:207], %0, %1 offset:3072\n\t" "global_load_dwordx4 v[220:223], %2, %3 offset:96" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_MM3(acc) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[208:211], v[192:195], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[212:215], v[196:199], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[216:219], v[200:203], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[220:223], v[204:207], %0" : "+v"(acc))
#define S16W_LD4(vo, rb, pv, qb) asm volatile("global_load_dwordx4 v[224:227], %0, %1\n\t" "global_load_dwordx4 v[240:243], %2, %3\n\t" "global_load_dwordx4 v[228:231], %0, %1 offset:1024\n\t" "global_load_dwordx4 v[244:247], %2, %3 offset:32\n\t" "global_load_dwordx4 v[232:235], %0, %1 offset:2048\n\t" "global_load_dwordx4 v[248:251], %2, %3 offset:64\n\t" "global_load_dwordx4 v[236:239], %0, %1 offset:3072\n\t" "global_load_dwordx4 v[252:255], %2, %3 offset:96" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_MM4(acc) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[240:243], v[224:227], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[244:247], v[228:231], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[248:251], v[232:235], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[252:255], v[236:239], %0" : "+v"(acc))
/* ---- end of generated ---- */

template <int N> __device__ __forceinline__ void
s16w_wait()
{
	static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
	asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

#ifdef NDB_PHASES
#define S16W_PH_DECL unsigned long long w_ph_t = wall_clock64(), w_ph[6] = {0, 0, 0, 0, 0, 0}
#define S16W_PH(I) do { const unsigned long long w_now = wall_clock64(); w_ph[I] += w_now - w_ph_t; w_ph_t = w_now; } while (0)
#define S16W_PH_FLUSH do { if (blockIdx.x == 0 && wave == 0 && lane == 0) for (int w_i = 0; w_i < 6; w_i++) atomicAdd(&g_phases[48 + w_i], w_ph[w_i]); } while (0)
#else
#define S16W_PH_DECL ((void) 0)
#define S16W_PH(I) ((void) 0)
#define S16W_PH_FLUSH ((void) 0)
#endif

/* D = chunks (of 64 dimensions: 4 KiB of rows + the pairs' 16 bytes a lane and k-step) a wave keeps in flight; an item
 * must have at least D chunks (the host falls back to k_s16c_sweep otherwise) */
template <int D, bool IPX>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(S16W_CVGPR))) void
k_s16c_wsweep(int dim, int nbuckets, const int64_t *__restrict__ loc_off, const uint32_t *__restrict__ own_len,
			  const unsigned char *__restrict__ planes, const uint32_t *__restrict__ blk_off,
			  const float *__restrict__ rn2, const int16_t *__restrict__ rexp,
			  const unsigned char *__restrict__ qcplanes, uint32_t qrowbytes, const float *__restrict__ qcn2,
			  const int *__restrict__ qcexp, const uint32_t *__restrict__ pqid, const uint32_t *__restrict__ pla,
			  const uint32_t *__restrict__ pnrow, float2 *qthr, const uint32_t *__restrict__ cnt,
			  const uint32_t *__restrict__ pair_off, const S16Desc *__restrict__ desc,
			  const uint32_t *__restrict__ runs, unsigned int *__restrict__ ecount, uint2 *__restrict__ erec,
			  float *__restrict__ eub, uint32_t ecap, uint32_t *__restrict__ bmin, int nchunk,
			  uint32_t desc_cap, uint32_t topk, const uint32_t *__restrict__ pos_of, float cE,
			  uint32_t qc_cap, int cosine, const float *__restrict__ rnx, const float *__restrict__ qev)
{
	static_assert(D >= 2 && D <= S16W_MAXD, "chunks a wave keeps in flight");
	constexpr int NC = IPX ? 7 : 6;		/* requests of an item's constants */
	constexpr int NX = NC + 1;			/* ... plus the member's threshold: what sits between an item's last chunk and the next item's first */
	/* the tightening's scratch, per wave: queries that crossed a multiple of S16_TIGHT records, and a query's bucket minima */
	__shared__ uint32_t s_tn[4], s_tq[4][S16_TIGHT_Q], s_tkeys[4][S16_NB];
	/* ... and, where an item has something to emit, its members' constants */
	__shared__ float s_mq2[4][32], s_mt2[4][32];
	__shared__ int s_meq[4][32];
	__shared__ uint32_t s_mqid[4][32], s_mnrow[4][32], s_mla[4][32];
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const int	wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	r32 = lane & 31, kh = lane >> 5;

	/* The stream's registers, named where the compiler looks: the kernel's allocation must include them, and whatever
	 * the compiler takes outside its own budget (the registers it spills scalars into) must not be one of them. */
	asm volatile("" :::
				 "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103",
				 "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
				 "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135",
				 "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151",
				 "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167",
				 "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183",
				 "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199",
				 "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215",
				 "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231",
				 "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247",
				 "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
	if (pair_off[nbuckets] > qc_cap)
		return;					/* uniform: more pairs than the pair planes hold, the batch goes to the older path */
	const uint32_t xq = blockIdx.x & 7u;
	const uint32_t stride = (gridDim.x - xq + 7u) >> 3;
	const uint32_t run_hi = min(runs[xq + 1], desc_cap);
	const uint32_t lane16 = (uint32_t) lane * 16u;

	if (lane == 0)
		s_tn[wave] = 0;

	/* an item as this wave sees it (all wave-uniform: scalar registers) */
	struct Item
	{
		uint32_t	it;				/* S16_NOITEM: none */
		uint32_t	nmem;
		const unsigned char *rbase, *qbase;
	};
	/* the first item at or after `it` (stride of the block's schedule) that has a 32-row block for this wave: scalar work
	 * only (descriptors through the scalar cache) */
	auto		find = [&](uint32_t it) -> Item {
		Item		f;
		S16Desc		d;

		d.L = d.t2 = d.qt = d.pad = 0;
		f.it = S16_NOITEM;
		while (it < run_hi)
		{
			d = desc[it];			/* uniform address: scalar loads */
			if (d.t2 * 128u + 32u * (uint32_t) wave < own_len[d.L])
			{
				f.it = it;
				break;
			}
			it += stride;
		}
		f.nmem = 1;
		f.rbase = planes;
		f.qbase = qcplanes;
		if (f.it != S16_NOITEM)
		{
			const uint32_t L = d.L;

			f.nmem = min(32u, cnt[L] - d.qt * 32u);
			f.rbase = planes + ((size_t) blk_off[L] + d.t2 * 4u + (uint32_t) wave) * (size_t) nchunk * 4096;
			f.qbase = qcplanes + (size_t) (pair_off[L] + d.qt * 32u) * qrowbytes;
		}
		return f;
	};
	auto		pvoff_of = [&](const Item &f) -> uint32_t {
		return min((uint32_t) r32, f.nmem - 1u) * qrowbytes + (uint32_t) kh * 16u;
	};
	/* request the item's constants of this lane into v88 .. v94: row 32 wave + r32 of the tile (beyond the bucket: its last
	 * row's), member min(r32, nmem - 1).  NC requests. */
	auto		request_consts = [&](const Item &f) {
		const S16Desc d = desc[f.it];
		const uint32_t L = d.L, len = own_len[L];
		const uint32_t slot0 = pair_off[L] + d.qt * 32u;
		const uint32_t ridx = d.t2 * 128u + 32u * (uint32_t) wave + (uint32_t) r32;
		const size_t grow = (size_t) loc_off[L] + (ridx < len ? ridx : len - 1u);
		const uint32_t mo = slot0 + min((uint32_t) r32, f.nmem - 1u);

		if constexpr (IPX)
			asm volatile("global_load_dword v89, %0, off" :: "v"(rnx + grow) : "memory");
		asm volatile("global_load_dword v88, %0, off\n\t"
					 "global_load_dword v90, %1, off\n\t"
					 "global_load_dword v91, %2, off\n\t"
					 "global_load_dword v92, %3, off\n\t"
					 "global_load_sshort v93, %4, off\n\t"
					 "global_load_dword v94, %5, off"
					 :: "v"(rn2 + grow), "v"(pos_of + grow), "v"(qcn2 + mo), "v"(pqid + mo), "v"(rexp + grow), "v"(qcexp + mo)
					 : "memory");
	};
	/* slot `sl` of the ring: wait until its chunk has landed — W = the stream's requests issued after that chunk's —,
	 * multiply it, request chunk c of item f into it */
	ndb_f16acc	acc;
	auto		step = [&](uint32_t sl, bool boundary, const Item &f, uint32_t pv, uint32_t c) {
		const unsigned char *rb = f.rbase + (size_t) c * 4096;
		const unsigned char *qb = f.qbase + (size_t) c * 128;

		if (boundary)
			s16w_wait<8 * (D - 1) + NX>();
		else
			s16w_wait<8 * (D - 1)>();
		switch (sl)
		{
			case 0: S16W_MM0(acc); S16W_LD0(lane16, rb, pv, qb); break;
			case 1: S16W_MM1(acc); S16W_LD1(lane16, rb, pv, qb); break;
			case 2: if constexpr (D > 2) { S16W_MM2(acc); S16W_LD2(lane16, rb, pv, qb); } break;
			case 3: if constexpr (D > 3) { S16W_MM3(acc); S16W_LD3(lane16, rb, pv, qb); } break;
			default: if constexpr (D > 4) { S16W_MM4(acc); S16W_LD4(lane16, rb, pv, qb); } break;
		}
	};

	Item		cur = find(runs[xq] + (blockIdx.x >> 3));

	if (cur.it == S16_NOITEM)
		return;
	S16W_PH_DECL;
	/* the stream's first requests: the first item's constants, then its first D chunks into slots 0 .. D - 1 */
	request_consts(cur);
	{
		const uint32_t pv = pvoff_of(cur);

		S16W_LD0(lane16, cur.rbase, pv, cur.qbase);
		S16W_LD1(lane16, cur.rbase + 4096, pv, cur.qbase + 128);
		if constexpr (D > 2) S16W_LD2(lane16, cur.rbase + 2 * 4096, pv, cur.qbase + 2 * 128);
		if constexpr (D > 3) S16W_LD3(lane16, cur.rbase + 3 * 4096, pv, cur.qbase + 3 * 128);
		if constexpr (D > 4) S16W_LD4(lane16, cur.rbase + 4 * 4096, pv, cur.qbase + 4 * 128);
	}
	uint32_t	sl = 0;				/* the slot that holds the stream's next chunk */

	for (;;)
	{
		const uint32_t pv = pvoff_of(cur);
		uint32_t	c = 0;

#pragma unroll
		for (int i = 0; i < 16; i++)
			acc[i] = 0.0f;
		/* all but the item's last D chunks: the slot takes the item's chunk D further on */
		for (; c + D < (uint32_t) nchunk; c++)
		{
			step(sl, false, cur, pv, c + D);
			sl = sl + 1 == D ? 0 : sl + 1;
		}
		/*
		 * The item's last D chunks; the slots take the next item's first D.  Before the first of them is requested:
		 * this item's constants (requested before its first chunk, so landed with it) move to ordinary registers, the
		 * members' thresholds are requested — as they stand now: in-sweep tightening; a stale value is a valid, looser
		 * bound; lanes beyond the members read the last member's — and then the next item's constants (the same
		 * registers).  No next item: the slots re-read this item's first chunks and nobody looks; every path through
		 * here issues the same requests, which is what the waits' counts rely on.
		 */
		float		x2, rx = 0.0f, q2;
		uint32_t	por, qid;
		int			ex, eq;

		/* (the constants are older than the item's first chunk: landed unless the item has only D chunks and none has
		 * been waited for yet) */
		s16w_wait<8 * D>();
		if constexpr (IPX)
			asm volatile("v_mov_b32 %0, v89" : "=v"(rx));
		asm volatile("v_mov_b32 %0, v88\n\tv_mov_b32 %1, v90\n\tv_mov_b32 %2, v91\n\tv_mov_b32 %3, v92\n\tv_mov_b32 %4, v93\n\tv_mov_b32 %5, v94"
					 : "=v"(x2), "=v"(por), "=v"(q2), "=v"(qid), "=v"(ex), "=v"(eq));
		asm volatile("global_load_dword v95, %0, off" :: "v"(&qthr[qid].x) : "memory");
		const Item	nxt = find(cur.it + stride);
		const Item	fi = nxt.it != S16_NOITEM ? nxt : cur;

		request_consts(fi);
		{
			const uint32_t pvn = pvoff_of(fi);

			/* (between each of these chunks and the newest request lie D - 1 chunks AND the NX requests above) */
			for (uint32_t cn = 0; cn < D; cn++)
			{
				step(sl, true, fi, pvn, cn);
				sl = sl + 1 == D ? 0 : sl + 1;
			}
		}
		/* the item's products are complete: the matrix pipe's results may be read by ordinary instructions 18 wait states
		 * after the last one; the thresholds (requested before the next item's constants and first D chunks) have landed */
		asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));
		s16w_wait<NC + 8 * D>();
		float		tfresh;

		asm volatile("v_mov_b32 %0, v95" : "=v"(tfresh));
		S16W_PH(0);
		{
			const S16Desc d = desc[cur.it];

			if (d.t2 * 128u + 32u * (uint32_t) wave + (uint32_t) r32 >= own_len[d.L])
				por = 0xFFFFFFFFu;		/* the tile has no such row (or, as stored, it is a deleted row's hole): never emitted */
		}

		/*
		 * The item's results.  Element (reg, lane) = member (reg & 3) + 8 (reg >> 2) + 4 kh, row r32 of the block;
		 * acc = (q - c).(x - c) 2^(28 - eq - ex).  Everything below is k_s16c_sweep's epilogue (pass 0 on the matrix pipe,
		 * pass 1 per element, pass 2 the records: see there for the bounds), with a member's constants living in lane
		 * `member` (and member + 32) instead of LDS.
		 */
		const bool	valid = (uint32_t) r32 < cur.nmem;

		if (!valid)
			tfresh = 0.0f;
		const bool	rok = por != 0xFFFFFFFFu;		/* (a hole never emits: it does not matter to `wild`) */
		const float KB = (1.0f - cE) * 0.9999962f;
		const float TB = s16_up(tfresh * 1.000004f) + NDB_S16_ABS;
		const float t2m = s16_up(tfresh * 1.000001f) + NDB_S16_ABS;		/* what pass 1 subtracts for this lane's member */
		float		cm = __builtin_fmaf(q2, KB, -TB);

		if (!(cm == cm))
			cm = -__builtin_inff();
		const float nu0 = -ldexpf(KB, 27 - eq);
		const float nu1 = valid ? -ldexpf(cm, 27 - eq) : -__builtin_inff();
		const bool	wild_l = (valid && (eq < -20 || eq > 20)) || (rok && (ex < -20 || ex > 20 || rx > 1.0e30f));
		const bool	wild = __ballot(wild_l) != 0ull;
		bool		hit;

		{
			const bool	nan = !(x2 == x2) || !(rx == rx);
			const float xw = IPX ? x2 + rx * 0.999996f / ((1.0f - cE) * 0.9999962f) : x2;
			const float w = !rok ? __builtin_inff() : (nan ? -__builtin_inff() : ldexpf(xw, -ex));
			const float wb = kh ? ldexpf(1.0f, -ex) : w;
			const float ua = kh ? nu1 : nu0;
			const ndb_f16acc fin = __builtin_amdgcn_mfma_f32_32x32x2f32(ua, wb, acc, 0, 0, 0);
			int			mx = (int) 0x80000000;

#pragma unroll
			for (int reg = 0; reg < 16; reg++)
				mx = max(mx, __float_as_int(fin[reg]));
			hit = __ballot(mx >= 0 || wild) != 0ull;
		}
		S16W_PH(1);
		if (hit)
		{
			/* (copies the compiler cannot see through: everything below that depends only on the lane — member indices,
			 * LDS addresses, lane masks — would otherwise be computed once in front of the sweep and kept, in scratch,
			 * across the stream's registers) */
			int			lane_o = lane, wave_o = wave;

			asm volatile("" : "+v"(lane_o));
			asm volatile("" : "+s"(wave_o));
			const int	r32_o = lane_o & 31, kh_o = lane_o >> 5;
			/* a few times per thousand items: the members' constants go to this wave_o's LDS scratch, where every lane_o finds
			 * the member of each of its 16 elements (lane_o m < 32 holds member m's) */
			const float K = (1.0f - cE) * 0.99999905f;

			{
				const S16Desc d = desc[cur.it];
				const uint32_t slot0 = pair_off[d.L] + d.qt * 32u;
				const uint32_t mo = min((uint32_t) r32_o, cur.nmem - 1u);

				if (lane_o < 32)
				{
					s_mq2[wave_o][lane_o] = q2;
					s_mt2[wave_o][lane_o] = t2m;
					s_meq[wave_o][lane_o] = eq;
					s_mqid[wave_o][lane_o] = qid;
					s_mnrow[wave_o][lane_o] = pnrow[slot0 + mo];
					s_mla[wave_o][lane_o] = pla[slot0 + mo];
				}
			}
			__builtin_amdgcn_wave_barrier();
			unsigned int emask = 0;		/* bit reg: the element cannot be left out */

#pragma unroll
			for (int reg = 0; reg < 16; reg++)
			{
				const int	m = (reg & 3) + 8 * (reg >> 2) + 4 * kh_o;
				const float t1 = ldexpf(acc[reg], s_meq[wave_o][m] + ex - 27);
				const float n = s_mq2[wave_o][m] + x2;
				const float rhs = __builtin_fmaf(n, K, -s_mt2[wave_o][m]) + rx * 0.99999905f;

				if (!(t1 < rhs) && (uint32_t) m < cur.nmem && por < s_mnrow[wave_o][m])
					emask |= 1u << reg;
			}
			uint32_t	anym = emask;

#pragma unroll
			for (int off = 32; off > 0; off >>= 1)
				anym |= (uint32_t) __shfl_xor((int) anym, off, 64);
			if (anym)
			{
				/* the slots of a member are handed out with a single atomicAdd by lane_o `member` (see k_s16c_sweep) */
				uint32_t	mycnt = 0;

				for (uint32_t rest = anym; rest; rest &= rest - 1)
				{
					const int	reg = __builtin_ctz(rest);
					const unsigned long long bal = __ballot((emask >> reg) & 1u);
					const int	ml0 = (reg & 3) + 8 * (reg >> 2);

					if (lane_o == ml0)
						mycnt += (uint32_t) __popcll(bal & 0xFFFFFFFFull);
					if (lane_o == ml0 + 4)
						mycnt += (uint32_t) __popcll(bal >> 32);
				}
				uint32_t	base = 0;

				if (mycnt != 0)
				{
					base = atomicAdd(&ecount[qid], mycnt);
					/* (looked at here, on every path: a result the compiler still counts as in flight at the loop's back
					 * edge makes it wait for ALL of the stream, vmcnt(0), wherever that register is touched next) */
					asm volatile("" : "+v"(base));
				}
#pragma unroll
				for (int reg = 0; reg < 16; reg++)
				{
					if (!((anym >> reg) & 1u))
						continue;			/* uniform */
					const bool	mine = (emask >> reg) & 1u;
					const unsigned long long bal = __ballot(mine);
					const int	ml0 = (reg & 3) + 8 * (reg >> 2), ml = ml0 + 4 * kh_o;
					const unsigned long long half = kh_o ? (bal >> 32) : (bal & 0xFFFFFFFFull);
					const uint32_t hb = (uint32_t) __shfl((int) base, ml, 64);

					if (mine)
					{
						const uint32_t slot = hb + (uint32_t) __popcll(half & ((1ull << r32_o) - 1ull));
						const uint32_t q = s_mqid[wave_o][ml];
						const float t1 = ldexpf(acc[reg], s_meq[wave_o][ml] + ex - 27);
						const float n = s_mq2[wave_o][ml] + x2;
						const float av = n - t1;
						const float er = s16_up(s16_up(cE * n) + NDB_S16_ABS);
						const float lbv = (av - er) + rx * 0.99999905f, ubv = s16_up(s16_up(av + er) + rx * 1.000001f);
						const float lb = lbv - fabsf(lbv) * 4.8e-7f - 1e-37f;
						const uint32_t pos = s_mla[wave_o][ml] + por, ub_bits = __float_as_uint(ubv);

						if (slot < ecap)
						{
							erec[(size_t) q * ecap + slot] = make_uint2(pos, __float_as_uint(lb));
							eub[(size_t) q * ecap + slot] = ubv;
						}
						if ((ub_bits & 0x7FFFFFFFu) < 0x7F800000u)
							atomicMin(&bmin[(size_t) q * S16_NB + ((pos * 2654435761u) >> (32 - S16_NB_LOG2))],
									  ndb_key_from_bits(ub_bits));
						if ((slot & (S16_TIGHT - 1)) == S16_TIGHT - 1)
						{
							const uint32_t ti = atomicAdd(&s_tn[wave_o], 1u);

							if (ti < S16_TIGHT_Q)
								s_tq[wave_o][ti] = q;
						}
					}
					const uint32_t nlo = (uint32_t) __popcll(bal & 0xFFFFFFFFull), nhi = (uint32_t) __popcll(bal >> 32);

					if (lane_o == ml0)
						base += nlo;
					if (lane_o == ml0 + 4)
						base += nhi;
				}
				/* a query that keeps emitting has a loose threshold: the k-th smallest bucket minimum bounds its k-th distance,
				 * so T is lowered here, while the sweep runs (monotone; any value read meanwhile is valid) */
				__builtin_amdgcn_wave_barrier();
				const uint32_t tn = min(s_tn[wave_o], (uint32_t) S16_TIGHT_Q);

				for (uint32_t jq = 0; jq < tn; jq++)
				{
					const uint32_t q = s_tq[wave_o][jq];
					uint32_t	k0, k1;

					static_assert(S16_NB == 128, "two bucket minima a lane_o");
					k0 = __hip_atomic_load(&bmin[(size_t) q * S16_NB + lane_o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					k1 = __hip_atomic_load(&bmin[(size_t) q * S16_NB + 64 + lane_o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					s_tkeys[wave_o][lane_o] = k0;
					s_tkeys[wave_o][64 + lane_o] = k1;
					__builtin_amdgcn_wave_barrier();
					uint32_t	rank0 = 0, rank1 = 0;

					for (uint32_t o = 0; o < S16_NB; o++)
					{
						const uint32_t ok = s_tkeys[wave_o][o];

						rank0 += (ok < k0 || (ok == k0 && o < (uint32_t) lane_o)) ? 1u : 0u;
						rank1 += (ok < k1 || (ok == k1 && o < (uint32_t) lane_o + 64u)) ? 1u : 0u;
					}
					uint32_t	mine = 0xFFFFFFFFu;

					if (topk != 0 && rank0 == topk - 1)
						mine = k0;
					if (topk != 0 && rank1 == topk - 1)
						mine = k1;
					if (mine != 0xFFFFFFFFu)
					{
						const uint32_t tb = (mine & 0x80000000u) ? (mine & 0x7FFFFFFFu) : ~mine;
						const float nt = qev ? s16c_ip_t_from_ub(__uint_as_float(tb), qev[q])
							: cosine ? s16c_cos_t_from_ub(__uint_as_float(tb), dim) : s16c_t_from_ub(__uint_as_float(tb), dim);

						atomicMin(reinterpret_cast<unsigned int *>(&qthr[q].x), __float_as_uint(nt));
					}
					__builtin_amdgcn_wave_barrier();
				}
				if (lane_o == 0 && tn != 0)
					s_tn[wave_o] = 0;
			}
			__builtin_amdgcn_wave_barrier();
		}
		S16W_PH(2);
#ifdef NDB_PHASES
		w_ph[5]++;
#endif
		if (nxt.it == S16_NOITEM)
			break;
		cur = nxt;
	}
	/* (the requests still in flight — the dummy chunks, the dummy constants — land in registers nobody reads; the wave
	 * must not end before they have: a new wave could be given those registers) */
	s16w_wait<0>();
	S16W_PH_FLUSH;
}

#endif							/* NDBHIP_SCREEN16W_H */
