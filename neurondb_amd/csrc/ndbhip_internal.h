/*
 * ndbhip_internal.h — what the translation units of libndbhip.so share (ndbhip.hip: runtime, IVF mirror and
 * scans; ndbhip_hnsw.hip: the HNSW mirror, search and build): the process context, error reporting, the
 * event timer of the dominant kernel, device allocation helpers and the block-level selection primitives
 * (radix select, ordered compaction, bitonic sort, the replay of the reference's selection sort).
 * Not part of the ABI.
 */
#ifndef NDBHIP_INTERNAL_H
#define NDBHIP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <vector>
#include <algorithm>
#include <mutex>
#include <chrono>
#include <type_traits>
#include <functional>
#include "../../include/ndbhip.h"
#include "ndbhip_kernels.h"
#pragma clang fp contract(off)

#include <utility>

/* ================================================================== */
/* context / errors                                                    */
/* ================================================================== */

extern thread_local char ndbhip_g_err[512];
#define g_err ndbhip_g_err


static int
fail(int code, const char *fmt, ...)
{
	va_list		ap;

	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr)                                                              \
	do {                                                                           \
		hipError_t _e = (expr);                                                    \
		if (_e != hipSuccess)                                                      \
			return fail(NDBHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
						hipGetErrorString(_e), __FILE__, __LINE__);                \
	} while (0)

/*
 * The stream everything is launched on.  Process-wide (ndbhip_set_stream), unless the CALLING THREAD has given itself a
 * stream of its own (ndbhip_set_thread_stream): two host threads, each with its own stream and its own mirror, keep two
 * batches in flight, and the per-query chains of one batch (selection, seeds, pair tables, finalize: waves waiting for
 * memory round trips) run under the other batch's sweep.  `g.stream` reads as the calling thread's stream everywhere.
 */
extern thread_local hipStream_t ndbhip_tl_stream;
struct StreamRef
{
	hipStream_t dflt = nullptr;
	operator hipStream_t() const { return ndbhip_tl_stream ? ndbhip_tl_stream : dflt; }
	StreamRef &operator=(hipStream_t s) { dflt = s; return *this; }
};

struct Ctx
{
	bool		inited = false;
	int			device = -1;
	hipStream_t own_stream = nullptr;
	StreamRef	stream;
	bool		profile = false;
	int			num_cus = 256;
	ndbhip_stats stats = {};
	unsigned long long *d_counters = nullptr;	/* [0] candidate rows scored (all ranks' view), [1] rows scored here */
	uint64_t	host_rows = 0, host_bytes = 0;	/* counted on the host (batch distance) */
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;	/* profiling events not yet read */
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
	/* scratch of the build's screened assignment (ndbhip_build.h: assign_rows_s16), kept between calls: a fresh
	 * multi-GB hipMalloc costs anything from 0.3 to 60 ms on this runtime, which is as much as a whole build */
	unsigned char *asg_arena = nullptr;
	size_t		asg_arena_cap = 0;
	float	   *pin_words = nullptr;	/* 1024 words of pinned, device-visible host memory: kernels leave single results here
									 * (a copy engine busy with a table upload would hold a 4-byte readback up for its whole queue) */
	/* large device blocks (an index's packed rows and TIDs) handed back by a destroyed or rebuilt index, kept for
	 * the next build of similar size for the same reason (big_alloc / big_free in ndbhip.hip) */
	std::vector<std::pair<void *, size_t>> big_live, big_cached;
	bool		big_cache_on = true;
};
extern Ctx	g;
/* the library's counters are bumped by every thread that has a batch in flight (ndbhip_set_thread_stream: the bench's lanes,
 * a service with several owner threads): relaxed atomic adds, read by ndbhip_stats_get as plain loads */
#define NDB_STAT_ADD(FIELD, N) __atomic_fetch_add(&g.stats.FIELD, (uint64_t) (N), __ATOMIC_RELAXED)
extern std::mutex ndbhip_g_mtx;		/* g's event pools and block cache, when several threads search at once */
int			big_alloc(void **out, size_t bytes);
void		big_free(void *p);
void		big_cache_flush(void);


static int
need_init()
{
	if (!g.inited)
		return fail(NDBHIP_ERR_NODEVICE, "ndbhip_init() has not succeeded in this process");
	/* HIP's current device is per thread, and the caller (torch, another library) may have changed it: an entry
	 * point called from another thread than ndbhip_init's must not allocate and launch on device 0 */
	if (hipSetDevice(g.device) != hipSuccess)
		return fail(NDBHIP_ERR_HIP, "hipSetDevice(%d) failed", g.device);
	return 0;
}


/* bracket the dominant kernel with events when profiling */
struct ScanTimer
{
	std::pair<hipEvent_t, hipEvent_t> ev{};
	bool		on = false;
	int start()
	{
		std::lock_guard<std::mutex> lk(ndbhip_g_mtx);

		NDB_STAT_ADD(scan_launches, 1);
		if (!g.profile)
			return 0;
		if (!g.pool.empty()) { ev = g.pool.back(); g.pool.pop_back(); }
		else
		{
			HIP_TRY(hipEventCreate(&ev.first));
			HIP_TRY(hipEventCreate(&ev.second));
		}
		HIP_TRY(hipEventRecord(ev.first, g.stream));
		on = true;
		return 0;
	}
	int stop()
	{
		if (!on)
			return 0;
		HIP_TRY(hipEventRecord(ev.second, g.stream));
		std::lock_guard<std::mutex> lk(ndbhip_g_mtx);

		g.pending.push_back(ev);
		return 0;
	}
};

/* grow-only device workspace */
template <class T>
static int
grow(T *&p, size_t &have, size_t want)
{
	if (want <= have)
		return 0;
	if (p)
		HIP_TRY(hipFree(p));
	p = nullptr;
	have = 0;
	HIP_TRY(hipMalloc((void **) &p, want * sizeof(T)));
	have = want;
	return 0;
}

/* ================================================================== */
/* block-level primitives                                              */
/* ================================================================== */

/*
 * Radix select over the order-preserving keys of the valid elements of a
 * sequence.  f(i, bits) -> valid.  On return (all threads):
 *   kk      = min(k_want, number of valid elements)
 *   T       = key of the kk-th smallest valid element (undefined if kk == 0)
 *   m_less  = number of valid elements with key < T
 *   cnt_eq  = number of valid elements with key == T
 * hist: 256 words of LDS; sh: 8 words of LDS.  Ends with a barrier.
 */
template <class F>
static __device__ void
block_radix_select(F f, uint32_t n, uint32_t k_want, uint32_t *hist, uint32_t *sh,
				   uint32_t &T, uint32_t &m_less, uint32_t &kk, uint32_t &cnt_eq)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t nthr = blockDim.x;
	uint32_t	prefix = 0,
				mask = 0;

	kk = 0;
	m_less = 0;
	cnt_eq = 0;
	T = 0;
	for (int pass = 0; pass < 4; pass++)
	{
		const int	shift = 24 - 8 * pass;

		for (uint32_t b = tid; b < 256; b += nthr)
			hist[b] = 0;
		__syncthreads();
		for (uint32_t i = tid; i < n; i += nthr)
		{
			uint32_t	bits;

			if (f(i, bits))
			{
				const uint32_t key = ndb_key_from_bits(bits);

				if ((key & mask) == prefix)
					atomicAdd(&hist[(key >> shift) & 255u], 1u);
			}
		}
		__syncthreads();
		/*
		 * Which bin holds the rem-th smallest: the first wave, four bins a lane — the lanes' sums, a prefix over the
		 * lanes, and the one lane whose range contains rem walks its four bins.  (One thread walking the 256 bins,
		 * a dependent LDS read each and twice in the first pass, was ~8 us a pass: five of those were a third of
		 * k_s16_finalize, whose 4096 one-wave blocks have nothing else to run meanwhile.)
		 */
		if (tid < 64)
		{
			const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
			const uint32_t mine = c0 + c1 + c2 + c3;
			uint32_t	inc = mine;

#pragma unroll
			for (int off = 1; off < 64; off <<= 1)
			{
				const uint32_t v = (uint32_t) __shfl_up((int) inc, off, 64);

				if (tid >= (uint32_t) off)
					inc += v;
			}
			const uint32_t nv = (uint32_t) __shfl((int) inc, 63, 64);
			const uint32_t kkv = (k_want < nv) ? k_want : nv;
			const uint32_t rem = (pass == 0) ? kkv : sh[1];		/* (sh[1] is rewritten below by the lane that finds the bin: all lanes have read it) */
			const uint32_t excl = inc - mine;

			__builtin_amdgcn_wave_barrier();
			if (pass == 0 && tid == 0)
				sh[3] = kkv;		/* kk */
			if (rem == 0)
			{
				if (tid == 0)
				{
					sh[0] = 0;
					sh[1] = 0;
					sh[2] = 0;
				}
			}
			else if (excl < rem && rem <= inc)
			{
				uint32_t	cum = excl, b = 4 * tid, c = c0;

				if (cum + c < rem) { cum += c; b++; c = c1; }
				if (cum + c < rem) { cum += c; b++; c = c2; }
				if (cum + c < rem) { cum += c; b++; c = c3; }
				sh[0] = b;
				sh[1] = rem - cum;	/* rank inside this bin, 1-based */
				sh[2] = c;
			}
		}
		__syncthreads();
		prefix |= sh[0] << shift;
		mask |= 0xFFu << shift;
		kk = sh[3];
		if (pass == 3)
		{
			cnt_eq = sh[2];
			m_less = kk - sh[1];
		}
		__syncthreads();
		if (kk == 0)
			return;
	}
	T = prefix;
}

/*
 * In-order compaction of the elements with key < T (class 0, all of them) and
 * key == T (class 1, the first eq_cap by index).  emit(cls, rank, i, bits).
 * sh: 16 words of LDS.  Block size must be a multiple of 64, at most 512.
 */
template <class F, class E>
static __device__ void
block_ordered_gather(F f, uint32_t n, uint32_t T, uint32_t eq_cap, uint32_t *sh, E emit)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t nthr = blockDim.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = tid >> 6;
	const uint32_t nwave = nthr >> 6;
	uint32_t	base_lt = 0,
				base_eq = 0;

	for (uint32_t start = 0; start < n; start += nthr)
	{
		const uint32_t i = start + tid;
		uint32_t	bits = 0;
		bool		valid = (i < n) && f(i, bits);
		const uint32_t key = ndb_key_from_bits(bits);
		const bool	is_lt = valid && key < T;
		const bool	is_eq = valid && key == T;
		const unsigned long long m_lt = __ballot(is_lt);
		const unsigned long long m_eq = __ballot(is_eq);
		const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
		const uint32_t r_lt = __popcll(m_lt & below);
		const uint32_t r_eq = __popcll(m_eq & below);

		if (lane == 0)
		{
			sh[wave * 2 + 0] = __popcll(m_lt);
			sh[wave * 2 + 1] = __popcll(m_eq);
		}
		__syncthreads();
		uint32_t	w_lt = 0, w_eq = 0, t_lt = 0, t_eq = 0;

		for (uint32_t w = 0; w < nwave; w++)
		{
			if (w < wave)
			{
				w_lt += sh[w * 2 + 0];
				w_eq += sh[w * 2 + 1];
			}
			t_lt += sh[w * 2 + 0];
			t_eq += sh[w * 2 + 1];
		}
		if (is_lt)
			emit(0, base_lt + w_lt + r_lt, i, bits);
		if (is_eq && base_eq + w_eq + r_eq < eq_cap)
			emit(1, base_eq + w_eq + r_eq, i, bits);
		base_lt += t_lt;
		base_eq += t_eq;
		__syncthreads();
	}
}

/* Bitonic sort of npad (power of two) 64-bit keys with a 32-bit payload, in LDS. */
static __device__ void
block_bitonic_sort(uint64_t *comp, uint32_t *payload, uint32_t npad)
{
	for (uint32_t size = 2; size <= npad; size <<= 1)
	{
		for (uint32_t stride = size >> 1; stride > 0; stride >>= 1)
		{
			__syncthreads();
			for (uint32_t t = threadIdx.x; t < (npad >> 1); t += blockDim.x)
			{
				const uint32_t lo = 2 * t - (t & (stride - 1));
				const uint32_t hi = lo + stride;
				const bool	up = ((lo & size) == 0);
				const uint64_t a = comp[lo], b = comp[hi];

				if ((a > b) == up)
				{
					const uint32_t pa = payload[lo], pb = payload[hi];

					comp[lo] = b;
					comp[hi] = a;
					payload[lo] = pb;
					payload[hi] = pa;
				}
			}
		}
	}
	__syncthreads();
}

/*
 * Rank of the value v among the DISTINCT keys[0 .. n): how many are below it (every caller's keys carry an index or a
 * position in their low word).  No barrier and no dependent LDS read inside — 16 reads are in flight per step — which
 * is what makes counting beat the sorting network for a few hundred keys: the network's log^2 stages each cost an LDS
 * round trip and a barrier (55 stages for 1024 keys were 30 us of a single query's probe selection).
 */
__device__ __forceinline__ uint32_t
lds_rank_u64(const uint64_t *keys, uint32_t n, uint64_t v)
{
	uint32_t	r = 0;
	uint32_t	t = 0;

	for (; t + 16 <= n; t += 16)
	{
		uint64_t	w[16];

#pragma unroll
		for (uint32_t i = 0; i < 16; i++)
			w[i] = keys[t + i];
#pragma unroll
		for (uint32_t i = 0; i < 16; i++)
			r += w[i] < v ? 1u : 0u;
	}
	if (t < n)
	{
		uint64_t	w[16];

#pragma unroll
		for (uint32_t i = 0; i < 16; i++)
			w[i] = (t + i < n) ? keys[t + i] : ~0ull;
#pragma unroll
		for (uint32_t i = 0; i < 16; i++)
			r += w[i] < v ? 1u : 0u;
	}
	return r;
}

/* up to this many keys are ordered by counting ranks (each thread reads all of them once per key it owns) */
#define NDB_RANK_SORT_MAX 512

/* comp[0 .. n), distinct keys, sorted ascending in place, payload[] along with it (n <= NDB_RANK_SORT_MAX and n <= 8 x blockDim);
 * entries from n on are left alone.  Starts and ends with a barrier. */
static __device__ void
block_rank_sort(uint64_t *comp, uint32_t *payload, uint32_t n)
{
	uint64_t	v[8];
	uint32_t	pl[8], r[8];

	__syncthreads();
#pragma unroll
	for (uint32_t e = 0; e < 8; e++)
	{
		const uint32_t j = threadIdx.x + e * blockDim.x;

		if (j < n)
		{
			v[e] = comp[j];
			pl[e] = payload[j];
			r[e] = lds_rank_u64(comp, n, v[e]);
		}
	}
	__syncthreads();
#pragma unroll
	for (uint32_t e = 0; e < 8; e++)
	{
		const uint32_t j = threadIdx.x + e * blockDim.x;

		if (j < n)
		{
			comp[r[e]] = v[e];
			payload[r[e]] = pl[e];
		}
	}
	__syncthreads();
}

__device__ __forceinline__ uint64_t
wave_min_u64(uint64_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		const uint32_t lo = __shfl_xor((uint32_t) v, off, 64);
		const uint32_t hi = __shfl_xor((uint32_t) (v >> 32), off, 64);
		const uint64_t o = ((uint64_t) hi << 32) | lo;

		v = (o < v) ? o : v;
	}
	return v;
}

/*
 * Final stage shared by IVF top-k, the shard merge and HNSW: given n entries
 * (dist bits, position in the reference's candidates[] array, payload id) in
 * LDS, replay the reference's selection sort (ivf_am.c:1856-1881) and write the
 * first kk = min(k, total) results.
 *
 * LDS scratch (npad = next pow2 >= n): comp[npad] u64, perm[npad] u32,
 * curpos[npad] u32, taken[npad] u8, order[k] u32.
 */
struct FinalizeScratch
{
	uint64_t   *comp;
	uint32_t   *perm;
	uint32_t   *curpos;
	uint8_t    *taken;
	uint32_t   *order;
};

/* Sort the n entries by (order key, position) and cut to the tie-complete prefix:
 * everything below T (= k-th smallest) plus the first 2k entries equal to T.
 * Returns (all threads) ns = prefix length; fills s.comp / s.perm. Ends with a barrier. */
static __device__ uint32_t
block_sort_cut(const uint32_t *e_bits, const uint32_t *e_pos, uint32_t n, uint32_t npad, uint32_t k,
			   uint64_t total, FinalizeScratch s, uint32_t &kk_out)
{
	const uint32_t tid = threadIdx.x;
	uint32_t	kk = (uint32_t) ((uint64_t) k < total ? (uint64_t) k : total);

	if (kk > n)
		kk = n;
	for (uint32_t j = tid; j < npad; j += blockDim.x)
	{
		if (j < n)
		{
			s.comp[j] = ((uint64_t) ndb_key_from_bits(e_bits[j]) << 32) | e_pos[j];
			s.perm[j] = j;
		}
		else
		{
			s.comp[j] = ~0ull;
			s.perm[j] = 0xFFFFFFFFu;
		}
	}
	if (n <= NDB_RANK_SORT_MAX && n <= 8 * blockDim.x)
		block_rank_sort(s.comp, s.perm, n);		/* (the padding is already where a sort would leave it) */
	else
		block_bitonic_sort(s.comp, s.perm, npad);

	uint32_t	ns = n;

	if (kk > 0)
	{
		const uint32_t Tkey = (uint32_t) (s.comp[kk - 1] >> 32);
		/* first index whose key >= T: binary search, every thread redundantly */
		uint32_t	lo = 0, hi = kk - 1;

		while (lo < hi)
		{
			const uint32_t mid = (lo + hi) >> 1;

			if ((uint32_t) (s.comp[mid] >> 32) < Tkey)
				lo = mid + 1;
			else
				hi = mid;
		}
		/* (entries with key > T inside [kk, ns) are harmless: they lose to every tie) */
		if (lo + 2 * k < ns)
			ns = lo + 2 * k;
	}
	kk_out = kk;
	__syncthreads();
	return ns;
}

/* Replay the reference's selection sort on the sorted prefix [0, ns) and write kk results. */
static __device__ void
block_replay_emit(const uint32_t *e_bits, const uint64_t *e_id, uint32_t ns, uint32_t kk, FinalizeScratch s,
				  uint64_t *out_id, float *out_dist, int *out_count)
{
	const uint32_t tid = threadIdx.x;

	for (uint32_t j = tid; j < ns; j += blockDim.x)
	{
		s.curpos[j] = (uint32_t) s.comp[j];
		s.taken[j] = 0;
	}
	__syncthreads();

	if (tid < 64)
	{
		for (uint32_t i = 0; i < kk; i++)
		{
			uint64_t	best = ~0ull;

			for (uint32_t j = tid; j < ns; j += 64)
				if (!s.taken[j])
				{
					const uint64_t c = (s.comp[j] & 0xFFFFFFFF00000000ull) | s.curpos[j];

					best = (c < best) ? c : best;
				}
			best = wave_min_u64(best);
			const uint32_t bpos = (uint32_t) best;

			for (uint32_t j = tid; j < ns; j += 64)
				if (!s.taken[j])
				{
					const uint64_t c = (s.comp[j] & 0xFFFFFFFF00000000ull) | s.curpos[j];

					if (c == best)
					{
						s.taken[j] = 1;
						s.order[i] = j;
					}
					else if (s.curpos[j] == i)
						s.curpos[j] = bpos;	/* the loser parked in slot i moves to the winner's slot */
				}
			wave_lds_sync();
		}
	}
	__syncthreads();
	for (uint32_t i = tid; i < kk; i += blockDim.x)
	{
		const uint32_t e = s.perm[s.order[i]];

		if (out_id)
			out_id[i] = e_id[e];
		out_dist[i] = ndb_u2f(e_bits[e]);
	}
	if (tid == 0)
		*out_count = (int) kk;
}

static __device__ void
block_finalize_topk(const uint32_t *e_bits, const uint32_t *e_pos, const uint64_t *e_id, uint32_t n,
					uint32_t npad, uint32_t k, uint64_t total, FinalizeScratch s,
					uint64_t *out_id, float *out_dist, int *out_count)
{
	uint32_t	kk;
	const uint32_t ns = block_sort_cut(e_bits, e_pos, n, npad, k, total, s, kk);

	block_replay_emit(e_bits, e_id, ns, kk, s, out_id, out_dist, out_count);
}

/* dynamic LDS layout of k_ivf_topk / k_merge_topk */
struct TopkSmem
{
	uint32_t   *hist;			/* 256 */
	uint32_t   *sh;				/* 16 */
	uint32_t   *e_bits;			/* cap */
	uint32_t   *e_pos;			/* cap */
	uint64_t   *e_id;			/* cap */
	FinalizeScratch fs;
};

__host__ __device__ static inline uint32_t
next_pow2(uint32_t v)
{
	uint32_t	p = 1;

	while (p < v)
		p <<= 1;
	return p;
}

__host__ __device__ static inline size_t
topk_smem_bytes(uint32_t cap, uint32_t k)
{
	const uint32_t npad = next_pow2(cap);

	return (size_t) (256 + 16) * 4 + (size_t) cap * (4 + 4 + 8) + (size_t) npad * (8 + 4 + 4 + 1) +
		(size_t) k * 4 + 64;
}

__device__ static inline TopkSmem
carve_topk_smem(unsigned char *base, uint32_t cap, uint32_t k)
{
	TopkSmem	s;
	const uint32_t npad = next_pow2(cap);
	unsigned char *p = base;

	s.e_id = (uint64_t *) p;			p += (size_t) cap * 8;
	s.fs.comp = (uint64_t *) p;			p += (size_t) npad * 8;
	s.hist = (uint32_t *) p;			p += 256 * 4;
	s.sh = (uint32_t *) p;				p += 16 * 4;
	s.e_bits = (uint32_t *) p;			p += (size_t) cap * 4;
	s.e_pos = (uint32_t *) p;			p += (size_t) cap * 4;
	s.fs.perm = (uint32_t *) p;			p += (size_t) npad * 4;
	s.fs.curpos = (uint32_t *) p;		p += (size_t) npad * 4;
	s.fs.order = (uint32_t *) p;		p += (size_t) k * 4;
	s.fs.taken = (uint8_t *) p;
	return s;
}

#define NDB_TOPK_FAST_MAXK 64		/* fast path: k <= 64 (256 thread minima bound the k-th value) */
#define NDB_S16_MAXK 256			/* the centred fp16 screen serves k up to this (L2 over sublists: thresholds from the buckets' radii, k_s16c_thr_radius) */
#define NDB_TOPK_FAST_CAP 1024		/* candidates <= U the fast path can hold before falling back */

__host__ __device__ static inline uint32_t
topk_entry_cap(uint32_t k)
{
	return (k <= NDB_TOPK_FAST_MAXK && 3 * k < NDB_TOPK_FAST_CAP) ? NDB_TOPK_FAST_CAP : 3 * k;
}

#define NDB_TOPK_MAX_SMEM (150 * 1024)

int			set_kernel_attributes_hnsw();
extern int	g_hnsw_trace, g_hnsw_nofast, g_h2_waves, g_h2_occ4, g_h2_host_groups;

#endif							/* NDBHIP_INTERNAL_H */
