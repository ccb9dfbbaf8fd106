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
 *     8 waves a compute unit x 3 chunks x 4 KiB of rows = 96 KiB of row bytes in flight (more changes nothing: the
 *     waves of a depth-5 build never waited for data, depth 2 waited 15 % of the time at the same speed).
 *   - The stream never stops at an item's end: the slot of the chunk just multiplied takes the stream's chunk D further
 *     on, which is the NEXT item's once this one has none left (descriptor through the scalar cache).
 *   - Work items are (bucket, ONE 32-row block, 32-pair tile) — k_s16_items with a row tile of 32 —, all of the same
 *     cost, in 8 runs (one per XCD: consecutive items share their pair tile through that XCD's L2); a wave takes the next
 *     item of its XCD's run with a fetch-and-add, an item ahead of the one it multiplies, and turns to the fullest other
 *     run when its own is empty.  (Tiles of 128 rows with
 *     wave w taking block w left the waves between 6 and 20 items each — sublists have 40 to 200 rows —: 368 us for
 *     296 us of mean work.  Equal shares, 16 or 17 items a wave, still ended between 233 and 351 us: the memory system
 *     does not serve all waves alike.  profiles/r05_wave_trace.txt)
 *   - The pairs' requests run under the mask of the lanes whose pair row exists (a handful of 32): the address path's
 *     time goes with the active lanes, and with all 64 lanes asking — most for the last member's row again — the pairs
 *     cost the address path as much as the rows (a timing build without them streamed the rows a third faster).
 *
 * WHY THE STREAM IS INLINE ASM WITH REGISTERS OF ITS OWN.  Written as plain loads into arrays the compiler software-
 * pipelines only by accident: one conditional request anywhere in the loop and its waitcnt pass falls back to
 * s_waitcnt vmcnt(0) at every use (measured on three formulations; a flat_load sneaking in does the same), and values
 * loaded next to a full ring get spilled to scratch the moment they arrive (scratch traffic retires through the same
 * in-order counter).  So the kernel is compiled with amdgpu_num_vgpr(44): the compiler owns v0-v87 (the attribute counts
 * accumulation registers too, and a kernel without any gets twice the number as ordinary ones) and NEVER touches
 * v88-v255 (they are reserved registers to it; tools/check_asm_hazards.py audits the generated code for that); the
 * stream's loads, waits and matrix instructions are inline asm on
 *     v88-v94, v97, v98   the item's constants in flight (|x - c|^2, M^2 - |x|^2, position, |q - c|^2, query, the two
 *               exponents, the member's visible rows and first candidate position)
 *     v95       the member's threshold in flight
 *     v96       the index of the next item, in flight (lane 0's fetch-and-add on its run's queue head)
 *     v128 + 32 j + 4 s ..  (rows: operand B)  and  v144 + 32 j + 4 s ..  (pairs: operand A)   of slot j, k-step s
 *     (three blocks a compute unit: the two slots start at v104, and the kernel ends at v167)
 * and every wait is `s_waitcnt vmcnt(n)` with n = the stream's own requests issued after the one needed.  That is
 * safe whatever else the compiler has in flight (the rare emission path's loads, atomics, stores): vector-memory loads
 * retire in order, so other requests among the newest n only make the wait longer, never shorter — the count assumes
 * nothing that was not issued.  The accumulators are an ordinary variable ("+v"): the compiler does not know the asm
 * wrote them with matrix instructions (it may copy them between register tuples right behind a step, and does), so
 * every step ends with the wait states the matrix pipe needs before an ordinary instruction reads its results.
 */
#ifndef NDBHIP_SCREEN16W_H
#define NDBHIP_SCREEN16W_H

/* the rows' requests carry the non-temporal hint: a stream that is read once should not push the pair tiles, which the
 * four row blocks of a sublist share, out of the L2 (A/B on one box: 0.369 -> 0.347 ms; -DS16W_NO_NT: without) */
#ifdef S16W_NO_NT
#define S16W_RNT ""
#else
#define S16W_RNT " nt"
#endif
/* -DS16W_NOPAIRS (timing experiments only: the results are garbage): the pairs' requests are left out */
#ifdef S16W_NOPAIRS
#define S16W_P(x) ""
#else
#define S16W_P(x) x
#endif
#define S16W_NUM_VGPR 44		/* amdgpu_num_vgpr: on gfx950 the compiler takes TWICE that many registers, v0 .. v87, when the kernel uses no accumulation registers (measured: tools/r05 notes in DESIGN.md; with 88 it could have taken v0 .. v175) */
#define S16W_MAXD 4				/* slots: v128 .. v255 */

/* ---- generated (tools/gen_s16w_asm.py): slot j's eight requests (rows at rb + vo, pairs at qb + pv; 16 bytes a
 * lane each), and a step on slot j: four matrix instructions, the slot's next eight requests, the matrix pipe's wait states ---- */
#define S16W_LD0(vo, rb, pv, qb) asm volatile("s_nop 4\n\t" "global_load_dwordx4 v[128:131], %0, %1" S16W_RNT "\n\t" "global_load_dwordx4 v[132:135], %0, %1 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[136:139], %0, %1 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[140:143], %0, %1 offset:3072" S16W_RNT "\n\t" S16W_P("global_load_dwordx4 v[144:147], %2, %3\n\t") S16W_P("global_load_dwordx4 v[148:151], %2, %3 offset:32\n\t") S16W_P("global_load_dwordx4 v[152:155], %2, %3 offset:64\n\t") S16W_P("global_load_dwordx4 v[156:159], %2, %3 offset:96\n\t") "s_nop 0" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_ST0(acc, vo, rb, pv, qb, mk) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[144:147], v[128:131], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[148:151], v[132:135], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[152:155], v[136:139], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[156:159], v[140:143], %0\n\t" "global_load_dwordx4 v[128:131], %1, %2" S16W_RNT "\n\t" "global_load_dwordx4 v[132:135], %1, %2 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[136:139], %1, %2 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[140:143], %1, %2 offset:3072" S16W_RNT "\n\t" "s_mov_b64 exec, %5\n\t" S16W_P("global_load_dwordx4 v[144:147], %3, %4\n\t") S16W_P("global_load_dwordx4 v[148:151], %3, %4 offset:32\n\t") S16W_P("global_load_dwordx4 v[152:155], %3, %4 offset:64\n\t") S16W_P("global_load_dwordx4 v[156:159], %3, %4 offset:96\n\t") "s_mov_b64 exec, -1\n\t" "s_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(vo), "s"(rb), "v"(pv), "s"(qb), "s"(mk) : "memory")
#define S16W_LD1(vo, rb, pv, qb) asm volatile("s_nop 4\n\t" "global_load_dwordx4 v[160:163], %0, %1" S16W_RNT "\n\t" "global_load_dwordx4 v[164:167], %0, %1 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[168:171], %0, %1 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[172:175], %0, %1 offset:3072" S16W_RNT "\n\t" S16W_P("global_load_dwordx4 v[176:179], %2, %3\n\t") S16W_P("global_load_dwordx4 v[180:183], %2, %3 offset:32\n\t") S16W_P("global_load_dwordx4 v[184:187], %2, %3 offset:64\n\t") S16W_P("global_load_dwordx4 v[188:191], %2, %3 offset:96\n\t") "s_nop 0" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_ST1(acc, vo, rb, pv, qb, mk) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[176:179], v[160:163], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[180:183], v[164:167], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[184:187], v[168:171], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[188:191], v[172:175], %0\n\t" "global_load_dwordx4 v[160:163], %1, %2" S16W_RNT "\n\t" "global_load_dwordx4 v[164:167], %1, %2 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[168:171], %1, %2 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[172:175], %1, %2 offset:3072" S16W_RNT "\n\t" "s_mov_b64 exec, %5\n\t" S16W_P("global_load_dwordx4 v[176:179], %3, %4\n\t") S16W_P("global_load_dwordx4 v[180:183], %3, %4 offset:32\n\t") S16W_P("global_load_dwordx4 v[184:187], %3, %4 offset:64\n\t") S16W_P("global_load_dwordx4 v[188:191], %3, %4 offset:96\n\t") "s_mov_b64 exec, -1\n\t" "s_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(vo), "s"(rb), "v"(pv), "s"(qb), "s"(mk) : "memory")
#define S16W_LD2(vo, rb, pv, qb) asm volatile("s_nop 4\n\t" "global_load_dwordx4 v[192:195], %0, %1" S16W_RNT "\n\t" "global_load_dwordx4 v[196:199], %0, %1 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[200:203], %0, %1 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[204:207], %0, %1 offset:3072" S16W_RNT "\n\t" S16W_P("global_load_dwordx4 v[208:211], %2, %3\n\t") S16W_P("global_load_dwordx4 v[212:215], %2, %3 offset:32\n\t") S16W_P("global_load_dwordx4 v[216:219], %2, %3 offset:64\n\t") S16W_P("global_load_dwordx4 v[220:223], %2, %3 offset:96\n\t") "s_nop 0" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_ST2(acc, vo, rb, pv, qb, mk) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[208:211], v[192:195], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[212:215], v[196:199], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[216:219], v[200:203], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[220:223], v[204:207], %0\n\t" "global_load_dwordx4 v[192:195], %1, %2" S16W_RNT "\n\t" "global_load_dwordx4 v[196:199], %1, %2 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[200:203], %1, %2 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[204:207], %1, %2 offset:3072" S16W_RNT "\n\t" "s_mov_b64 exec, %5\n\t" S16W_P("global_load_dwordx4 v[208:211], %3, %4\n\t") S16W_P("global_load_dwordx4 v[212:215], %3, %4 offset:32\n\t") S16W_P("global_load_dwordx4 v[216:219], %3, %4 offset:64\n\t") S16W_P("global_load_dwordx4 v[220:223], %3, %4 offset:96\n\t") "s_mov_b64 exec, -1\n\t" "s_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(vo), "s"(rb), "v"(pv), "s"(qb), "s"(mk) : "memory")
#define S16W_LD3(vo, rb, pv, qb) asm volatile("s_nop 4\n\t" "global_load_dwordx4 v[224:227], %0, %1" S16W_RNT "\n\t" "global_load_dwordx4 v[228:231], %0, %1 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[232:235], %0, %1 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[236:239], %0, %1 offset:3072" S16W_RNT "\n\t" S16W_P("global_load_dwordx4 v[240:243], %2, %3\n\t") S16W_P("global_load_dwordx4 v[244:247], %2, %3 offset:32\n\t") S16W_P("global_load_dwordx4 v[248:251], %2, %3 offset:64\n\t") S16W_P("global_load_dwordx4 v[252:255], %2, %3 offset:96\n\t") "s_nop 0" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W_ST3(acc, vo, rb, pv, qb, mk) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[240:243], v[224:227], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[244:247], v[228:231], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[248:251], v[232:235], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[252:255], v[236:239], %0\n\t" "global_load_dwordx4 v[224:227], %1, %2" S16W_RNT "\n\t" "global_load_dwordx4 v[228:231], %1, %2 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[232:235], %1, %2 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[236:239], %1, %2 offset:3072" S16W_RNT "\n\t" "s_mov_b64 exec, %5\n\t" S16W_P("global_load_dwordx4 v[240:243], %3, %4\n\t") S16W_P("global_load_dwordx4 v[244:247], %3, %4 offset:32\n\t") S16W_P("global_load_dwordx4 v[248:251], %3, %4 offset:64\n\t") S16W_P("global_load_dwordx4 v[252:255], %3, %4 offset:96\n\t") "s_mov_b64 exec, -1\n\t" "s_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(vo), "s"(rb), "v"(pv), "s"(qb), "s"(mk) : "memory")
#define S16W3_LD0(vo, rb, pv, qb) asm volatile("s_nop 4\n\t" "global_load_dwordx4 v[104:107], %0, %1" S16W_RNT "\n\t" "global_load_dwordx4 v[108:111], %0, %1 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[112:115], %0, %1 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[116:119], %0, %1 offset:3072" S16W_RNT "\n\t" S16W_P("global_load_dwordx4 v[120:123], %2, %3\n\t") S16W_P("global_load_dwordx4 v[124:127], %2, %3 offset:32\n\t") S16W_P("global_load_dwordx4 v[128:131], %2, %3 offset:64\n\t") S16W_P("global_load_dwordx4 v[132:135], %2, %3 offset:96\n\t") "s_nop 0" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W3_ST0(acc, vo, rb, pv, qb, mk) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[120:123], v[104:107], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[124:127], v[108:111], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[128:131], v[112:115], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[132:135], v[116:119], %0\n\t" "global_load_dwordx4 v[104:107], %1, %2" S16W_RNT "\n\t" "global_load_dwordx4 v[108:111], %1, %2 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[112:115], %1, %2 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[116:119], %1, %2 offset:3072" S16W_RNT "\n\t" "s_mov_b64 exec, %5\n\t" S16W_P("global_load_dwordx4 v[120:123], %3, %4\n\t") S16W_P("global_load_dwordx4 v[124:127], %3, %4 offset:32\n\t") S16W_P("global_load_dwordx4 v[128:131], %3, %4 offset:64\n\t") S16W_P("global_load_dwordx4 v[132:135], %3, %4 offset:96\n\t") "s_mov_b64 exec, -1\n\t" "s_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(vo), "s"(rb), "v"(pv), "s"(qb), "s"(mk) : "memory")
#define S16W3_LD1(vo, rb, pv, qb) asm volatile("s_nop 4\n\t" "global_load_dwordx4 v[136:139], %0, %1" S16W_RNT "\n\t" "global_load_dwordx4 v[140:143], %0, %1 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[144:147], %0, %1 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[148:151], %0, %1 offset:3072" S16W_RNT "\n\t" S16W_P("global_load_dwordx4 v[152:155], %2, %3\n\t") S16W_P("global_load_dwordx4 v[156:159], %2, %3 offset:32\n\t") S16W_P("global_load_dwordx4 v[160:163], %2, %3 offset:64\n\t") S16W_P("global_load_dwordx4 v[164:167], %2, %3 offset:96\n\t") "s_nop 0" :: "v"(vo), "s"(rb), "v"(pv), "s"(qb) : "memory")
#define S16W3_ST1(acc, vo, rb, pv, qb, mk) asm volatile("s_nop 1\n\t" "v_mfma_f32_32x32x16_f16 %0, v[152:155], v[136:139], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[156:159], v[140:143], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[160:163], v[144:147], %0\n\t" "v_mfma_f32_32x32x16_f16 %0, v[164:167], v[148:151], %0\n\t" "global_load_dwordx4 v[136:139], %1, %2" S16W_RNT "\n\t" "global_load_dwordx4 v[140:143], %1, %2 offset:1024" S16W_RNT "\n\t" "global_load_dwordx4 v[144:147], %1, %2 offset:2048" S16W_RNT "\n\t" "global_load_dwordx4 v[148:151], %1, %2 offset:3072" S16W_RNT "\n\t" "s_mov_b64 exec, %5\n\t" S16W_P("global_load_dwordx4 v[152:155], %3, %4\n\t") S16W_P("global_load_dwordx4 v[156:159], %3, %4 offset:32\n\t") S16W_P("global_load_dwordx4 v[160:163], %3, %4 offset:64\n\t") S16W_P("global_load_dwordx4 v[164:167], %3, %4 offset:96\n\t") "s_mov_b64 exec, -1\n\t" "s_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(vo), "s"(rb), "v"(pv), "s"(qb), "s"(mk) : "memory")
/* ---- end of generated ---- */

template <int N> __device__ __forceinline__ void
s16w_wait()
{
	static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
	asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

/* profiling builds (-DNDB_PHASES): every wave leaves g_wtrace[8 (4 block + wave) ..] = its first request's and its end's
 * 100 MHz clock, its items, the ticks it spent inside the stream's waits, the ticks inside the emission path and how many
 * items took it, the ticks looking for work in other runs (ndbhip_debug_trace; NDB_TRACE=file bench.py) */
#ifdef NDB_PHASES
#define S16W_TRACE_N 24576
__device__ unsigned long long g_wtrace[S16W_TRACE_N];
#define S16W_PH_DECL unsigned long long w_t0 = wall_clock64(), w_wait = 0, w_items = 0, w_rare = 0, w_hits = 0, w_steal = 0, w_sub[4] = {0, 0, 0, 0}, w_subt = 0, w_last = 0, w_maxi = 0, w_tight = 0, w_irare = 0, w_iemit = 0, w_mx_nmem = 0, w_mx_rare = 0, w_mx_emit = 0
#define S16W_WAIT_T(STMT) do { const unsigned long long w_a = wall_clock64(); STMT; w_wait += wall_clock64() - w_a; } while (0)
#define S16W_PH_ITEM do { const unsigned long long w_e = wall_clock64(); w_items++; if (w_last && w_e - w_last > w_maxi) { w_maxi = w_e - w_last; w_mx_nmem = cur.nmem; w_mx_rare = w_irare; w_mx_emit = w_iemit; } w_last = w_e; w_irare = 0; w_iemit = 0; } while (0)
#define S16W_RARE_T(A) do { const unsigned long long w_b = wall_clock64(); w_rare += w_b - (A); w_irare = w_b - (A); w_hits++; } while (0)
#define S16W_EMIT_N(M) do { unsigned int w_pc = __popc(M); for (int w_o = 32; w_o > 0; w_o >>= 1) w_pc += (unsigned int) __shfl_xor((int) w_pc, w_o, 64); w_iemit = w_pc; } while (0)
#define S16W_STEAL_T(A) do { w_steal += wall_clock64() - (A); } while (0)
#define S16W_NOW wall_clock64()
#define S16W_SUB(I) do { const unsigned long long w_c = wall_clock64(); w_sub[I] += w_c - w_subt; w_subt = w_c; } while (0)
#define S16W_PH_FLUSH do { const uint32_t w_i = 12u * (blockIdx.x * 4u + (uint32_t) wave); if (lane == 0 && w_i + 11 < S16W_TRACE_N) { g_wtrace[w_i + 8] = w_sub[0]; g_wtrace[w_i + 9] = w_mx_nmem; g_wtrace[w_i + 10] = w_mx_rare; g_wtrace[w_i + 11] = w_mx_emit; g_wtrace[w_i] = w_t0; g_wtrace[w_i + 1] = wall_clock64(); g_wtrace[w_i + 2] = w_items; g_wtrace[w_i + 3] = w_wait; g_wtrace[w_i + 4] = w_rare; g_wtrace[w_i + 5] = w_hits; g_wtrace[w_i + 6] = w_steal; g_wtrace[w_i + 7] = w_maxi; } } while (0)
#else
#define S16W_PH_DECL ((void) 0)
#define S16W_WAIT_T(STMT) STMT
#define S16W_PH_ITEM ((void) 0)
#define S16W_RARE_T(A) ((void) 0)
#define S16W_EMIT_N(M) ((void) 0)
#define S16W_STEAL_T(A) ((void) 0)
#define S16W_NOW 0ull
#define S16W_SUB(I) ((void) 0)
#define S16W_PH_FLUSH ((void) 0)
#endif

/* D = chunks (of 64 dimensions: 4 KiB of rows + the pairs' 16 bytes a lane and k-step) a wave keeps in flight; an item
 * must have at least D chunks (the host falls back to k_s16c_sweep otherwise) */
/* BLK = blocks of 4 waves a compute unit: 2 (256 registers a lane, D up to 4) or 3 (168 registers, D = 2: the emission
 * path behind nearly every item of a clustered table is a chain of memory round trips — 9 of an item's 20 us —, and what
 * hides it is other waves streaming meanwhile) */
template <int D, bool IPX, int BLK>
__global__ __launch_bounds__(256, BLK) __attribute__((amdgpu_num_vgpr(S16W_NUM_VGPR))) void
k_s16c_wsweep(int dim, int nbuckets, const int64_t *__restrict__ loc_off, const uint32_t *__restrict__ own_len,
			  const unsigned char *__restrict__ planes, const uint32_t *__restrict__ blk_off,
			  const float *__restrict__ rn2, const int16_t *__restrict__ rexp,
			  const unsigned char *__restrict__ qcplanes, uint32_t qrowbytes, const float *__restrict__ qcn2,
			  const int *__restrict__ qcexp, const uint32_t *__restrict__ pqid,
			  const uint32_t *__restrict__ pebase /* per pair: its first (pair, 32-row block) word */,
			  const uint32_t *__restrict__ pnrow, const float2 *__restrict__ qthr, const uint32_t *__restrict__ cnt,
			  const uint32_t *__restrict__ pair_off, const S16Desc *__restrict__ desc,
			  const uint32_t *__restrict__ runs,
			  uint32_t *__restrict__ wmask /* [words] per (pair, 32-row block): the rows whose element cannot be left out */,
			  float2 *__restrict__ wrec /* [32 words] their (lower, upper) bounds, at 32 word + row */,
			  uint32_t wcap /* words there are: more and nothing is swept (k_s16w_pairinfo raised the flag) */,
			  const uint32_t *__restrict__ wtotal /* -> the words the batch needs (the offsets' grand total) */,
			  int nchunk, uint32_t desc_cap, const uint32_t *__restrict__ pos_of, float cE, uint32_t qc_cap,
			  const float *__restrict__ rnx, unsigned int *__restrict__ heads /* [8][NDB_QHEAD_STRIDE] zeroed: items taken from each XCD's run */ )
{
	static_assert(D >= 2 && D <= S16W_MAXD, "chunks a wave keeps in flight");
	static_assert(BLK == 2 || (BLK == 3 && D == 2), "three blocks a compute unit leave the stream two slots");
	constexpr int NC = IPX ? 9 : 8;		/* requests of an item's constants */
	constexpr int NX = NC + 2;			/* ... plus the member's threshold and the fetch-and-add: what sits between an item's last chunk and the next item's first */
	/* an item's members' constants, per wave: where every lane finds the member of each of its 16 elements */
	__shared__ float s_mq2[4][32], s_mt2[4][32];
	__shared__ int s_meq[4][32];
	__shared__ uint32_t s_mnrow[4][32], s_mbase[4][32];
	const int	tid = threadIdx.x;
	const int	lane = tid & 63;
	const int	wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int	r32 = lane & 31, kh = lane >> 5;

	/* The stream's registers, named where the compiler looks: the kernel's allocation must include them, and whatever
	 * the compiler takes outside its own budget (the registers it spills scalars into) must not be one of them. */
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
	if constexpr (BLK == 3)
		asm volatile("" :::
					 "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103",
					 "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
					 "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135",
					 "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151",
					 "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167");
	else
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
#pragma clang diagnostic pop
	if (pair_off[nbuckets] > qc_cap || *wtotal > wcap)
		return;					/* uniform: more pairs than the pair planes or the masks hold, the batch goes to the older path */
	/* this wave's items: those it takes from run (block % 8), its XCD's */
	const uint32_t xq = blockIdx.x & 7u;
	/* the run the wave draws from: its XCD's — later, once that is empty, the fullest other one's (`steal`) */
	uint32_t	run_lo = runs[xq], run_hi = min(runs[xq + 1], desc_cap);
	unsigned int *head = heads + xq * NDB_QHEAD_STRIDE;
	const uint32_t lane16 = (uint32_t) lane * 16u;

	/* an item as the stream needs it (all wave-uniform: scalar registers) */
	struct Item
	{
		uint32_t	it;				/* S16_NOITEM: none */
		uint32_t	nmem;
		const unsigned char *rbase, *qbase;
	};
	auto		find = [&](uint32_t it) -> Item {
		Item		f;

		f.it = S16_NOITEM;
		f.nmem = 1;
		f.rbase = planes;
		f.qbase = qcplanes;
		if (it != S16_NOITEM)
		{
			const S16Desc d = desc[it];			/* uniform address: scalar loads */
			const uint32_t L = d.L;

			f.it = it;
			f.nmem = min(32u, cnt[L] - d.qt * 32u);
			f.rbase = planes + ((size_t) blk_off[L] + d.t2) * (size_t) nchunk * 4096;
			f.qbase = qcplanes + (size_t) (pair_off[L] + d.qt * 32u) * qrowbytes;
		}
		return f;
	};
	auto		pvoff_of = [&](const Item &f) -> uint32_t {
		return min((uint32_t) r32, f.nmem - 1u) * qrowbytes + (uint32_t) kh * 16u;
	};
	/* the lanes whose pair row exists: r32 < nmem, both k-halves */
	auto		mask_of = [&](const Item &f) -> unsigned long long {
		const unsigned long long lo = f.nmem >= 32u ? 0xFFFFFFFFull : ((1ull << f.nmem) - 1ull);

		return lo | (lo << 32);
	};
	/* lane 0 takes the next item of the run: the index comes back in v86, behind everything requested before (1 request) */
	auto		request_grab = [&]() {
		asm volatile("s_mov_b64 exec, 1\n\tglobal_atomic_add v96, %0, %1, off sc0\n\ts_mov_b64 exec, -1"
					 :: "v"(head), "v"(1u) : "memory");
	};
	auto		read_grab = [&]() -> uint32_t {
		uint32_t	g;

		asm volatile("v_readfirstlane_b32 %0, v96\n\ts_nop 4" : "=s"(g));
		/* (the counter runs past the run's length: S16_NOITEM then, whatever the sum would wrap to) */
		return g < run_hi - run_lo ? run_lo + g : S16_NOITEM;
	};
	/*
	 * The wave's run is empty (the XCDs do not finish together: their runs are equal, the bandwidth they get is not):
	 * it turns to the run with the most items left — lane x < 8 looks at head x — and takes from there from now on.
	 * Returns the item taken, or S16_NOITEM when every run is empty.  Synchronous (the stream is drained: this is the
	 * sweep's last stretch); the request still in flight on the old head is lost with it, which is an index nobody needs.
	 */
	auto		steal = [&]() -> uint32_t {
		for (int round = 0; round < 8; round++)
		{
			uint32_t	left = 0;

			if (lane < 8)
			{
				const uint32_t lo = runs[lane], hi = min(runs[lane + 1], desc_cap);
				const uint32_t taken = __hip_atomic_load(&heads[(uint32_t) lane * NDB_QHEAD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

				left = hi > lo && taken < hi - lo ? hi - lo - taken : 0u;
			}
			uint32_t	best = left, arg = (uint32_t) lane;

#pragma unroll
			for (int off = 4; off > 0; off >>= 1)
			{
				const uint32_t ob = (uint32_t) __shfl_xor((int) best, off, 64), oa = (uint32_t) __shfl_xor((int) arg, off, 64);

				if (ob > best || (ob == best && oa < arg))
				{
					best = ob;
					arg = oa;
				}
			}
			best = (uint32_t) __builtin_amdgcn_readfirstlane((int) best);
			arg = (uint32_t) __builtin_amdgcn_readfirstlane((int) arg);
			if (best == 0)
				return S16_NOITEM;
			run_lo = runs[arg];
			run_hi = min(runs[arg + 1], desc_cap);
			head = heads + arg * NDB_QHEAD_STRIDE;
			s16w_wait<0>();			/* (the old head's request has landed in v96: the new one must not be overtaken by it) */
			request_grab();
			s16w_wait<0>();
			const uint32_t it = read_grab();

			if (it != S16_NOITEM)
				return it;
		}
		return S16_NOITEM;
	};
	/* request the item's constants of this lane into v88 .. v94: row r32 of the block (beyond the bucket: its last row's),
	 * member min(r32, nmem - 1).  NC requests. */
	auto		request_consts = [&](const Item &f) {
		const S16Desc d = desc[f.it];
		const uint32_t L = d.L, len = own_len[L];
		const uint32_t slot0 = pair_off[L] + d.qt * 32u;
		const uint32_t ridx = d.t2 * 32u + (uint32_t) r32;
		const size_t grow = (size_t) loc_off[L] + (ridx < len ? ridx : len - 1u);
		const uint32_t mo = slot0 + min((uint32_t) r32, f.nmem - 1u);

		if constexpr (IPX)
			asm volatile("global_load_dword v89, %0, off" :: "v"(rnx + grow) : "memory");
		asm volatile("global_load_dword v88, %0, off\n\t"
					 "global_load_dword v90, %1, off\n\t"
					 "global_load_dword v91, %2, off\n\t"
					 "global_load_dword v92, %3, off\n\t"
					 "global_load_sshort v93, %4, off\n\t"
					 "global_load_dword v94, %5, off\n\t"
					 "global_load_dword v97, %6, off\n\t"
					 "global_load_dword v98, %7, off"
					 :: "v"(rn2 + grow), "v"(pos_of + grow), "v"(qcn2 + mo), "v"(pqid + mo), "v"(rexp + grow), "v"(qcexp + mo),
						"v"(pnrow + mo), "v"(pebase + mo)
					 : "memory");
	};
	/* slot `sl` of the ring: wait until its chunk has landed — W = the stream's requests issued after that chunk's —,
	 * multiply it, request chunk c of item f into it */
	S16W_PH_DECL;
	ndb_f16acc	acc;
	auto		step = [&](uint32_t sl, bool boundary, const Item &f, uint32_t pv, unsigned long long mk, uint32_t c) {
		const unsigned char *rb = f.rbase + (size_t) c * 4096;
		const unsigned char *qb = f.qbase + (size_t) c * 128;

		if (boundary)
			S16W_WAIT_T(s16w_wait<8 * (D - 1) + NX>());
		else
			S16W_WAIT_T(s16w_wait<8 * (D - 1)>());
		switch (sl)
		{
			case 0: if constexpr (BLK == 3) S16W3_ST0(acc, lane16, rb, pv, qb, mk); else S16W_ST0(acc, lane16, rb, pv, qb, mk); break;
			case 1: if constexpr (BLK == 3) S16W3_ST1(acc, lane16, rb, pv, qb, mk); else S16W_ST1(acc, lane16, rb, pv, qb, mk); break;
			case 2: if constexpr (D > 2) S16W_ST2(acc, lane16, rb, pv, qb, mk); break;
			default: if constexpr (D > 3) S16W_ST3(acc, lane16, rb, pv, qb, mk); break;
		}
	};

	/* the first item; the index of the second is asked for now and read when the first item's last chunks are due */
	request_grab();
	s16w_wait<0>();
	Item		cur = find(read_grab());

	if (cur.it == S16_NOITEM)
		return;
	request_grab();
	/* the stream's first requests: the first item's constants, then its first D chunks into slots 0 .. D - 1 */
	request_consts(cur);
	{
		const uint32_t pv = pvoff_of(cur);

		if constexpr (BLK == 3)
		{
			S16W3_LD0(lane16, cur.rbase, pv, cur.qbase);
			S16W3_LD1(lane16, cur.rbase + 4096, pv, cur.qbase + 128);
		}
		else
		{
			S16W_LD0(lane16, cur.rbase, pv, cur.qbase);
			S16W_LD1(lane16, cur.rbase + 4096, pv, cur.qbase + 128);
		}
		if constexpr (D > 2) S16W_LD2(lane16, cur.rbase + 2 * 4096, pv, cur.qbase + 2 * 128);
		if constexpr (D > 3) S16W_LD3(lane16, cur.rbase + 3 * 4096, pv, cur.qbase + 3 * 128);
	}
	uint32_t	sl = 0;				/* the slot that holds the stream's next chunk */

	for (;;)
	{
		const uint32_t pv = pvoff_of(cur);
		const unsigned long long mk = mask_of(cur);
		uint32_t	c = 0;

#pragma unroll
		for (int i = 0; i < 16; i++)
			acc[i] = 0.0f;
		/* all but the item's last D chunks: the slot takes the item's chunk D further on */
		for (; c + D < (uint32_t) nchunk; c++)
		{
			step(sl, false, cur, pv, mk, c + D);
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
		uint32_t	por, qid, nrow_m, base_m;
		int			ex, eq;

		/* (the constants are older than the item's first chunk: landed unless the item has only D chunks and none has
		 * been waited for yet) */
		s16w_wait<8 * D>();
		if constexpr (IPX)
			asm volatile("v_mov_b32 %0, v89" : "=v"(rx));
		asm volatile("v_mov_b32 %0, v88\n\tv_mov_b32 %1, v90\n\tv_mov_b32 %2, v91\n\tv_mov_b32 %3, v92\n\tv_mov_b32 %4, v93\n\tv_mov_b32 %5, v94"
					 : "=v"(x2), "=v"(por), "=v"(q2), "=v"(qid), "=v"(ex), "=v"(eq));
		asm volatile("v_mov_b32 %0, v97\n\tv_mov_b32 %1, v98" : "=v"(nrow_m), "=v"(base_m));
		asm volatile("global_load_dword v95, %0, off" :: "v"(&qthr[qid].x) : "memory");
		/* the next item (asked for an item ago: landed long since; from another XCD's run when this one is empty), and the
		 * request for the one after it */
		uint32_t	it_next = read_grab();

		if (it_next == S16_NOITEM)
		{
			const unsigned long long w_steal_a = S16W_NOW;

			it_next = steal();
			S16W_STEAL_T(w_steal_a);
		}
		const Item	nxt = find(it_next);
		const Item	fi = nxt.it != S16_NOITEM ? nxt : cur;

		request_grab();
		request_consts(fi);
		{
			const uint32_t pvn = pvoff_of(fi);
			const unsigned long long mkn = mask_of(fi);

			/* (between each of these chunks and the newest request lie D - 1 chunks AND the NX requests above) */
			for (uint32_t cn = 0; cn < D; cn++)
			{
				step(sl, true, fi, pvn, mkn, cn);
				sl = sl + 1 == D ? 0 : sl + 1;
			}
		}
		/* the item's products are complete; the thresholds (requested before the next item's constants and first D chunks)
		 * have landed */
		s16w_wait<NC + 1 + 8 * D>();
		float		tfresh;

		asm volatile("v_mov_b32 %0, v95" : "=v"(tfresh));
		const S16Desc dcur = desc[cur.it];
		const uint32_t ridx = dcur.t2 * 32u + (uint32_t) r32;

		if (ridx >= own_len[dcur.L])
			por = 0xFFFFFFFFu;		/* the block has no such row (or, as stored, it is a deleted row's hole): never emitted */

		/*
		 * The item's results.  Element (reg, lane) = member (reg & 3) + 8 (reg >> 2) + 4 kh, row r32 of the block;
		 * acc = (q - c).(x - c) 2^(28 - eq - ex);  t1 = 2 dot (a power-of-two scaling, exact);  the element is LEFT OUT
		 * when  a - E > T,  a = n - t1,  n = Q2 + X2,  E = cE n + ABS — k_s16c_sweep's per-element test, to the letter
		 * (see there for why the fused form below never exceeds the real right-hand side).
		 *
		 * What is NOT k_s16c_sweep's: where the results go.  On a clustered table the pairs that reach the sweep are the
		 * queries' own neighbourhoods — 96 % of the items have something to emit (measured) —, and a slot in the query's
		 * record array costs a fetch-and-add with its round trip, per member and item: 9 of an item's 20 us went into
		 * that chain (loads of the members' constants, the counts, the atomic, the records, the tightening), none of it
		 * hidden by anything.  So an element has a PLACE OF ITS OWN instead: the pair's words start at pebase, word
		 * (pair, block) holds the block's row mask and wrec[32 word + row] the row's two bounds.  Every (pair, block) word is
		 * written by exactly one item (zero when nothing stays), no counter is touched, nothing is waited for;
		 * k_s16w_collect turns a query's words into the record list k_s16_finalize reads.
		 */
		const bool	valid = (uint32_t) r32 < cur.nmem;

		if (!valid)
			tfresh = 0.0f;
		{
			const unsigned long long w_rare_a = S16W_NOW;
			const float K = (1.0f - cE) * 0.99999905f;
			const float t2m = s16_up(tfresh * 1.000001f) + NDB_S16_ABS;		/* what the test subtracts for this lane's member */

			if (lane < 32)
			{
				s_mq2[wave][lane] = q2;
				s_mt2[wave][lane] = t2m;
				s_meq[wave][lane] = eq;
				s_mnrow[wave][lane] = nrow_m;
				s_mbase[wave][lane] = base_m;
			}
			__builtin_amdgcn_wave_barrier();
			uint32_t	mymask = 0;		/* lane m < 32: the rows of the block that stay for member m */

#pragma unroll
			for (int reg = 0; reg < 16; reg++)
			{
				const int	m0 = (reg & 3) + 8 * (reg >> 2), m = m0 + 4 * kh;
				const float t1 = ldexpf(acc[reg], s_meq[wave][m] + ex - 27);
				const float n = s_mq2[wave][m] + x2;
				const float rhs = __builtin_fmaf(n, K, -s_mt2[wave][m]) + rx * 0.99999905f;
				const bool	stays = !(t1 < rhs) && (uint32_t) m < cur.nmem && por < s_mnrow[wave][m];
				const unsigned long long bal = __ballot(stays);

				if (lane == m0)
					mymask = (uint32_t) bal;
				if (lane == m0 + 4)
					mymask = (uint32_t) (bal >> 32);
				if (bal != 0ull && stays)
				{
					const float av = n - t1;
					const float er = s16_up(s16_up(cE * n) + NDB_S16_ABS);
					const float lbv = (av - er) + rx * 0.99999905f, ubv = s16_up(s16_up(av + er) + rx * 1.000001f);
					const float lb = lbv - fabsf(lbv) * 4.8e-7f - 1e-37f;

					wrec[(size_t) 32u * (s_mbase[wave][m] + dcur.t2) + (uint32_t) r32] = make_float2(lb, ubv);
				}
			}
			if (lane < 32 && valid)
				wmask[base_m + dcur.t2] = mymask;
			__builtin_amdgcn_wave_barrier();
			S16W_RARE_T(w_rare_a);
		}
		S16W_PH_ITEM;
		if (nxt.it == S16_NOITEM)
			break;
		cur = nxt;
	}
	/* (the requests still in flight — the dummy chunks, the dummy constants — land in registers nobody reads; the wave
	 * must not end before they have: a new wave could be given those registers) */
	s16w_wait<0>();
	S16W_PH_FLUSH;
}


/*
 * Before the sweep, one thread per pair (in the pair tables' order): where the pair's words start — wbase[bucket] (the
 * offsets' third column with wmode = 1: pairs before the bucket x their buckets' 32-row blocks) plus the pair's rank in
 * its bucket x the bucket's blocks —, its bucket, and the pair's entry in ITS QUERY's list (a fetch-and-add per pair;
 * the order inside a query's list is whatever it comes out as: k_s16_finalize does not care about the order of the
 * records, the sweep's own slots never had one).  flags[0] is raised when the batch needs more words than there are,
 * flags[1] when a query has more pairs than its list holds: the batch then goes to the older path.
 */
__global__ __launch_bounds__(256) void
k_s16w_pairinfo(const PairRec *__restrict__ pairs, const uint32_t *__restrict__ pair_off, int nb,
				const uint32_t *__restrict__ wbase /* [nb + 1] */, const uint32_t *__restrict__ own_len, uint32_t qc_cap,
				uint32_t wcap, uint32_t *__restrict__ pebase, uint32_t *__restrict__ pbkt, uint32_t *__restrict__ qslot,
				uint32_t *__restrict__ qsn, uint32_t qcap, unsigned int *__restrict__ flag_words, unsigned int *__restrict__ flag_list,
				unsigned int *__restrict__ words_needed /* = the batch's words (the host sizes the next batch's arrays by it) */ )
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t total = pair_off[nb];

	if (j == 0)
	{
		*words_needed = wbase[nb];
		if (wbase[nb] > wcap)
			atomicAdd(flag_words, 1u);
	}
	if (j >= total || total > qc_cap || wbase[nb] > wcap)
		return;
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
	pebase[j] = wbase[lo] + (j - pair_off[lo]) * ((own_len[lo] + 31u) >> 5);
	pbkt[j] = lo;
	const uint32_t q = pairs[j].q;
	const uint32_t i = atomicAdd(&qsn[q], 1u);

	if (i < qcap)
		qslot[(size_t) q * qcap + i] = j;
	else if (i == qcap)
		atomicAdd(flag_list, 1u);
}

/*
 * After the sweep, one wave per query: its pairs' words -> the record list k_s16_finalize reads (erec[q][i] = (candidate
 * position, lower bound), eub[q][i] = upper bound, ecount[q]; the positions and bounds the sweep's own slots held, in
 * another order).  Three steps, each with all its loads in flight together: the pairs (bucket, first word, blocks), the
 * words (row mask -> count, running sums over the wave), the records (record r -> its word by bisection over the sums
 * -> the set bit of its rank -> bounds and list position; words and records S16W_COLLECT_WORDS words at a time).  Records
 * beyond ecap are counted and not written (the query overflows: k_s16_finalize hands it to the exact path, as ever).
 */
#define S16W_COLLECT_WORDS 512		/* (pair, block) words a query's wave keeps at a time (more: in several goes); 9 KB of LDS a wave with the pairs' arrays: 17 waves a compute unit, the batch's 4096 in one go */
__global__ __launch_bounds__(64) void
k_s16w_collect(uint32_t nq, const uint32_t *__restrict__ qslot, const uint32_t *__restrict__ qsn, uint32_t qcap,
			   const uint32_t *__restrict__ pbkt, const uint32_t *__restrict__ pebase, const uint32_t *__restrict__ pla,
			   const int64_t *__restrict__ prow_off, const uint32_t *__restrict__ own_len, const uint32_t *__restrict__ pos_of,
			   const uint32_t *__restrict__ wmask, const float2 *__restrict__ wrec, unsigned int *__restrict__ ecount,
			   uint2 *__restrict__ erec, float *__restrict__ eub, uint32_t ecap, const unsigned int *__restrict__ active,
			   const uint32_t *__restrict__ pair_off, int nb, uint32_t qc_cap, const uint32_t *__restrict__ wtotal, uint32_t wcap)
{
	__shared__ uint32_t s_first[S16_QP_CAP + 1], s_word[S16_QP_CAP], s_la[S16_QP_CAP];
	__shared__ uint32_t s_prow[S16_QP_CAP];		/* (padded plane rows: fewer than 2^32, ivf_s16_prepare checks the blocks) */
	__shared__ uint32_t s_off[S16W_COLLECT_WORDS + 1], s_mask[S16W_COLLECT_WORDS];
	__shared__ uint16_t s_pair[S16W_COLLECT_WORDS];
	const uint32_t q = blockIdx.x;
	const int	lane = threadIdx.x;

	if (q >= nq || (active && !active[q]) || pair_off[nb] > qc_cap || *wtotal > wcap)
		return;
	const uint32_t n = min(qsn[q], min(qcap, (uint32_t) S16_QP_CAP));
	/* the pairs: their blocks, summed over the lanes */
	uint32_t	run = 0;

	for (uint32_t i0 = 0; i0 < n; i0 += 64)
	{
		const uint32_t i = i0 + (uint32_t) lane;
		uint32_t	nbk = 0;

		if (i < n)
		{
			const uint32_t j = qslot[(size_t) q * qcap + i], L = pbkt[j];

			nbk = (own_len[L] + 31u) >> 5;
			s_word[i] = pebase[j];
			s_la[i] = pla[j];
			s_prow[i] = (uint32_t) prow_off[L];
		}
		uint32_t	inc = nbk;

#pragma unroll
		for (int off = 1; off < 64; off <<= 1)
		{
			const uint32_t v = (uint32_t) __shfl_up((int) inc, off, 64);

			if (lane >= off)
				inc += v;
		}
		if (i < n)
			s_first[i] = run + inc - nbk;
		run += (uint32_t) __shfl((int) inc, 63, 64);
	}
	if (lane == 0)
		s_first[n] = run;
	__builtin_amdgcn_wave_barrier();
	const uint32_t nwords = run;
	/* the words, S16W_COLLECT_WORDS at a time: masks and where each word's records start, then the records */
	uint32_t	nrec = 0;

	for (uint32_t c0 = 0; c0 < nwords; c0 += S16W_COLLECT_WORDS)
	{
		const uint32_t cw = min((uint32_t) S16W_COLLECT_WORDS, nwords - c0);
		const uint32_t rec0 = nrec;

		for (uint32_t w0 = 0; w0 < cw; w0 += 64)
		{
			const uint32_t wl = w0 + (uint32_t) lane, w = c0 + wl;
			uint32_t	mk = 0, pi = 0;

			if (wl < cw)
			{
				uint32_t	lo = 0, hi = n;

				while (hi - lo > 1)
				{
					const uint32_t mid = (lo + hi) >> 1;

					if (s_first[mid] <= w)
						lo = mid;
					else
						hi = mid;
				}
				while (lo + 1 < n && s_first[lo + 1] <= w)
					lo++;
				pi = lo;
				mk = wmask[s_word[pi] + (w - s_first[pi])];
			}
			const uint32_t c = (uint32_t) __popc(mk);
			uint32_t	inc = c;

#pragma unroll
			for (int off = 1; off < 64; off <<= 1)
			{
				const uint32_t v = (uint32_t) __shfl_up((int) inc, off, 64);

				if (lane >= off)
					inc += v;
			}
			if (wl < cw)
			{
				s_off[wl] = nrec + inc - c;
				s_mask[wl] = mk;
				s_pair[wl] = (uint16_t) pi;
			}
			nrec += (uint32_t) __shfl((int) inc, 63, 64);
		}
		if (lane == 0)
			s_off[cw] = nrec;
		__builtin_amdgcn_wave_barrier();
		/* the chunk's records */
		const uint32_t rend = min(nrec, ecap);

		for (uint32_t r = rec0 + (uint32_t) lane; r < rend; r += 64)
		{
			uint32_t	lo = 0, hi = cw;

			while (hi - lo > 1)
			{
				const uint32_t mid = (lo + hi) >> 1;

				if (s_off[mid] <= r)
					lo = mid;
				else
					hi = mid;
			}
			while (lo + 1 < cw && s_off[lo + 1] <= r)
				lo++;
			uint32_t	mk = s_mask[lo];

			for (uint32_t t = r - s_off[lo]; t > 0; t--)
				mk &= mk - 1u;
			const uint32_t bit = (uint32_t) __builtin_ctz(mk);
			const uint32_t pi = s_pair[lo], blk = c0 + lo - s_first[pi];
			const float2 lu = wrec[(size_t) 32u * (s_word[pi] + blk) + bit];
			const uint32_t por = pos_of[(size_t) s_prow[pi] + 32u * blk + bit];

			erec[(size_t) q * ecap + r] = make_uint2(s_la[pi] + por, __float_as_uint(lu.x));
			eub[(size_t) q * ecap + r] = lu.y;
		}
		__builtin_amdgcn_wave_barrier();
	}
	if (lane == 0)
		ecount[q] = nrec;
}

#endif							/* NDBHIP_SCREEN16W_H */
