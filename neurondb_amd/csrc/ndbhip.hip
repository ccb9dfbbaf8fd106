/*
 * ndbhip.hip — MI355X (gfx950) implementation of include/ndbhip.h.
 *
 * Kernel pipeline of one IVF search batch (reference call stack:
 * ivfgettuple -> ivfSelectClusters -> ivfCollectCandidates,
 * NeuronDB/src/index/ivf_am.c:1911-2027, 1597-1717, 1722-1909):
 *
 *   k_rows_scan      query x every centroid        (HOT LOOP 1, :1660-1681)
 *   k_probe_select   nprobe first-min selection     (:1686-1714) + candidate offsets
 *   k_ivf_scan       query x every probed entry     (HOT LOOP 2, :1810-1834)  <- dominant, HBM-bound
 *   k_ivf_topk       k-th value, tie-complete subset, selection-sort replay (:1856-1899)
 *
 * Written for wave64 / CDNA4 only.
 */
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

#include "../../include/ndbhip.h"
#include "ndbhip_kernels.h"

#pragma clang fp contract(off)

/* ================================================================== */
/* context / errors                                                    */
/* ================================================================== */

static thread_local char g_err[512];

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

/* list-scan kernel choice: 0 auto (grouped for batches >= NDB_GROUPED_MIN_NQ queries and dim % 64 == 0),
 * 1 always per-query (k_ivf_scan), 2 always grouped (k_ivf_scan_grouped) */
static int	g_scan_mode = 0;
/* screened L2 scan in auto mode (NDBHIP_SCREEN=0 turns it off); batches below this many queries keep the
 * exact scan (the two extra passes cost more than they save there) */
static bool g_screen_auto = true;
#define NDB_SCREEN_MIN_NQ 128
/* measured crossover on MI355X (tools/small_batch_probe.py, 1M x 768, probes 32): the grouped path costs 0.38 ms for 1..16
 * queries, the per-query path 0.18 / 0.24 / 0.35 / 0.50 ms for 1 / 2 / 4 / 7 */
#define NDB_GROUPED_MIN_NQ 5
/* rows staged per step by the grouped kernels: 64 floats (16 KiB tile, 3 waves/SIMD) or 32 (8 KiB, 4 waves/SIMD);
 * NDBHIP_GCHUNK overrides for experiments */
static int	g_gchunk = 32;
/* hnswbuild: 0 the one-wave sequential kernel, 1 optimistic batches with the chunked block-wide commit (hashed
 * when m <= 16, else sorted), 2 optimistic batches with the one-wave commit, 3 optimistic batches with the
 * sorted chunked commit; batch = min(max, nodes so far / div) walks */
static int	g_hnsw_search_mode = 0;
static int	g_hnsw_spec = 1;
static int	g_hnsw_batch_div = 64;
static int	g_hnsw_batch_max = 1024;

struct Ctx
{
	bool		inited = false;
	int			device = -1;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	bool		profile = false;
	int			num_cus = 256;
	ndbhip_stats stats = {};
	unsigned long long *d_counters = nullptr;	/* [0] candidate rows scored (all ranks' view), [1] rows scored here */
	uint64_t	host_rows = 0, host_bytes = 0;	/* counted on the host (batch distance) */
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;	/* profiling events not yet read */
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};
static Ctx	g;

static int
need_init()
{
	if (!g.inited)
		return fail(NDBHIP_ERR_NODEVICE, "ndbhip_init() has not succeeded in this process");
	return 0;
}

static int	set_kernel_attributes();
static int	set_kernel_attributes_build();

extern "C" int
ndbhip_abi_version(void)
{
	return NDBHIP_ABI_VERSION;
}

extern "C" const char *
ndbhip_last_error(void)
{
	return g_err;
}

extern "C" int
ndbhip_device_count(void)
{
	int			n = 0;
	hipError_t	e = hipGetDeviceCount(&n);

	if (e != hipSuccess)
	{
		(void) hipGetLastError();
		return fail(NDBHIP_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
	}
	return n;
}

extern "C" int
ndbhip_init(int device)
{
	int			n;

	if (g.inited)
	{
		if (device == g.device)
			return NDBHIP_OK;
		return fail(NDBHIP_ERR_STATE, "already initialised on device %d", g.device);
	}
	n = ndbhip_device_count();
	if (n <= 0)
		return fail(NDBHIP_ERR_NODEVICE, "no HIP device visible");
	if (device < 0 || device >= n)
		return fail(NDBHIP_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
	HIP_TRY(hipSetDevice(device));
	{
		hipDeviceProp_t prop;

		HIP_TRY(hipGetDeviceProperties(&prop, device));
		if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
			return fail(NDBHIP_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only",
						device, prop.gcnArchName);
		g.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	}
	HIP_TRY(hipStreamCreateWithFlags(&g.own_stream, hipStreamNonBlocking));
	g.stream = g.own_stream;
	HIP_TRY(hipMalloc((void **) &g.d_counters, 8 * sizeof(unsigned long long)));
	HIP_TRY(hipMemset(g.d_counters, 0, 8 * sizeof(unsigned long long)));
	g.device = device;
	g.inited = true;
	{
		if (getenv("NDBHIP_SCREEN"))
			g_screen_auto = atoi(getenv("NDBHIP_SCREEN")) != 0;
		const char *e = getenv("NDBHIP_GCHUNK");

		if (e && atoi(e) == 32)
			g_gchunk = 32;
		else if (e && atoi(e) == 64)
			g_gchunk = 64;
	}
	return set_kernel_attributes();
}

extern "C" int
ndbhip_shutdown(void)
{
	if (!g.inited)
		return NDBHIP_OK;
	(void) hipStreamSynchronize(g.stream);
	for (auto &p : g.pending) { (void) hipEventDestroy(p.first); (void) hipEventDestroy(p.second); }
	for (auto &p : g.pool) { (void) hipEventDestroy(p.first); (void) hipEventDestroy(p.second); }
	g.pending.clear();
	g.pool.clear();
	if (g.own_stream)
		(void) hipStreamDestroy(g.own_stream);
	if (g.d_counters)
		(void) hipFree(g.d_counters);
	g = Ctx();
	return NDBHIP_OK;
}

extern "C" int
ndbhip_set_stream(void *s)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	g.stream = s ? (hipStream_t) s : g.own_stream;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_get_stream(void **out_hip_stream)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!out_hip_stream)
		return fail(NDBHIP_ERR_INVALID, "out is NULL");
	*out_hip_stream = (void *) g.stream;
	return NDBHIP_OK;
}

/* the other translation units of the library report errors through the same thread-local message */
extern "C" int
ndbhip_internal_fail(int code, const char *fmt, ...)
{
	va_list		ap;

	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

extern "C" int
ndbhip_synchronize(void)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	HIP_TRY(hipStreamSynchronize(g.stream));
	return NDBHIP_OK;
}

static int
drain_profile_events()
{
	for (auto &p : g.pending)
	{
		float		ms = 0.f;

		HIP_TRY(hipEventSynchronize(p.second));
		HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
		g.stats.scan_kernel_ms += ms;
		g.pool.push_back(p);
	}
	g.pending.clear();
	return 0;
}

extern "C" int
ndbhip_stats_get(ndbhip_stats *out)
{
	if (!out)
		return fail(NDBHIP_ERR_INVALID, "out is NULL");
	if (g.inited)
	{
		unsigned long long c[8];

		if (drain_profile_events())
			return NDBHIP_ERR_HIP;
		HIP_TRY(hipStreamSynchronize(g.stream));
		HIP_TRY(hipMemcpy(c, g.d_counters, sizeof(c), hipMemcpyDeviceToHost));
		g.stats.rows_scored = g.host_rows + c[1];
		g.stats.bytes_scored = g.host_bytes + c[2];
		g.stats.rows_rescored = c[3];
		g.stats.rows_emitted = c[4];
	}
	*out = g.stats;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_stats_reset(void)
{
	if (g.inited)
	{
		if (drain_profile_events())
			return NDBHIP_ERR_HIP;
		HIP_TRY(hipStreamSynchronize(g.stream));
		HIP_TRY(hipMemset(g.d_counters, 0, 8 * sizeof(unsigned long long)));
	}
	g.host_rows = g.host_bytes = 0;
	g.stats = ndbhip_stats();
	return NDBHIP_OK;
}

extern "C" int
ndbhip_set_scan_mode(int mode)
{
	if (mode < 0 || mode > 5)
		return fail(NDBHIP_ERR_INVALID, "scan mode must be 0 (auto), 1 (per-query), 2 (grouped), 3 (grouped, screened by "
					"the fp32 bound pass), 4 (grouped, never screened) or 5 (screened by the fp16 matrix-core pass)");
	g_scan_mode = mode;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_profile(int on)
{
	g.profile = on != 0;
	return NDBHIP_OK;
}

/* bracket the dominant kernel with events when profiling */
struct ScanTimer
{
	std::pair<hipEvent_t, hipEvent_t> ev{};
	bool		on = false;
	int start()
	{
		g.stats.scan_launches++;
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
		g.pending.push_back(ev);
		return 0;
	}
};

/* ================================================================== */
/* device-side index views                                             */
/* ================================================================== */

struct IvfDev
{
	const float *vecs;			/* [nrows_local * dim] */
	const uint64_t *tids;		/* [nrows_local] */
	const float *centroids;		/* [ncent * dim] */
	const int64_t *loc_off;		/* [ncent + 1] local row offsets */
	const uint32_t *glob_len;	/* [ncent] global live entries per list */
	const uint8_t *owned;		/* [ncent] this mirror holds some of the list's rows */
	const uint32_t *own_lo;		/* [ncent] first list position held here (0 unless a list is split over ranks) */
	const uint32_t *own_len;	/* [ncent] rows of the list held here: positions own_lo .. own_lo + own_len */
	int			dim;
	int			ncent;			/* centroid items present ("maxoff") */
	int			nlists;			/* meta->nlists */
	int			f16;			/* rows are fp16 (vecs points at them; dim % 64 == 0) */
};

/* of the first l positions of list c, how many does this mirror hold */
__device__ __forceinline__ uint64_t
ndb_local_part(uint64_t l, const uint32_t *__restrict__ own_lo, const uint32_t *__restrict__ own_len, int c)
{
	const uint64_t lo = own_lo[c];
	const uint64_t hi = lo + own_len[c];

	return l > lo ? ((l < hi ? l : hi) - lo) : 0;
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
__device__ void
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
		if (tid == 0)
		{
			uint32_t	rem;
			uint32_t	cum = 0;

			if (pass == 0)
			{
				uint32_t	nv = 0;

				for (int b = 0; b < 256; b++)
					nv += hist[b];
				sh[3] = (k_want < nv) ? k_want : nv;	/* kk */
				rem = sh[3];
			}
			else
				rem = sh[1];
			sh[0] = 0;
			sh[2] = 0;
			if (rem > 0)
			{
				for (int b = 0; b < 256; b++)
				{
					const uint32_t c = hist[b];

					if (cum + c >= rem)
					{
						sh[0] = (uint32_t) b;
						sh[1] = rem - cum;	/* rank inside this bin, 1-based */
						sh[2] = c;
						break;
					}
					cum += c;
				}
			}
			else
				sh[1] = 0;
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
__device__ void
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
__device__ void
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
__device__ uint32_t
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
__device__ void
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

__device__ void
block_finalize_topk(const uint32_t *e_bits, const uint32_t *e_pos, const uint64_t *e_id, uint32_t n,
					uint32_t npad, uint32_t k, uint64_t total, FinalizeScratch s,
					uint64_t *out_id, float *out_dist, int *out_count)
{
	uint32_t	kk;
	const uint32_t ns = block_sort_cut(e_bits, e_pos, n, npad, k, total, s, kk);

	block_replay_emit(e_bits, e_id, ns, kk, s, out_id, out_dist, out_count);
}

/* ================================================================== */
/* kernels                                                             */
/* ================================================================== */

/* out[q * out_stride + r] = dist(query q, base row r), r < nrows.
 * grid = (ceil(nrows / 256), nq), block = 256 (4 independent waves). */
template <int R>
__global__ __launch_bounds__(256) void
k_rows_scan(const float *__restrict__ base, uint32_t nrows, int dim,
			const float *__restrict__ queries, float *__restrict__ out, uint32_t out_stride)
{
	__shared__ __attribute__((aligned(16))) float tiles[4 * NDB_TILE_FLOATS];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t q = blockIdx.y;
	const uint32_t r0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 64;	/* block = 1..4 waves of 64 rows */

	if (r0 >= nrows)
		return;
	const uint32_t r = r0 + lane;
	const uint32_t row = (r < nrows) ? r : (nrows - 1);
	const float d = score_rows<R>(queries + (size_t) q * dim, base, row, dim,
								  tiles + wave * NDB_TILE_FLOATS);

	if (r < nrows)
		out[(size_t) q * out_stride + r] = d;
}

/*
 * ivfSelectClusters selection + candidate offsets.  One block per query.
 *   cdist[q * cstride + c], c < ncmp (= min(nlists, ncent))
 *   probes[q * npr + i]:  i < nsel: i-th nearest centroid (first-min ties);
 *                         nsel <= i < npr_eff: -1 (bestIdx stays -1);
 *                         npr_eff <= i < npr: 0 (palloc0 slots never written: ivf_am.c:1978)
 *   cand_off[q * (npr+1) + i]: start of probe i's entries in candidates[],
 *                         capped at `cap` (ivf_am.c:1743) when cap > 0.
 */
__global__ __launch_bounds__(256) void
k_probe_select(const float *__restrict__ cdist, uint32_t cstride, int ncmp, int ncent, int npr,
			   const uint32_t *__restrict__ glob_len, const uint32_t *__restrict__ own_lo,
			   const uint32_t *__restrict__ own_len, uint64_t cap,
			   int dim, int *__restrict__ probes, uint32_t *__restrict__ cand_off,
			   uint32_t *__restrict__ loc_cand_off, unsigned long long *__restrict__ counters)
{
	__shared__ uint32_t hist[256];
	__shared__ uint32_t sh[16];
	__shared__ uint64_t comp[NDBHIP_MAX_NPROBE];
	__shared__ uint64_t full[2048];
	__shared__ uint32_t perm[NDBHIP_MAX_NPROBE];
	__shared__ uint32_t lens[NDBHIP_MAX_NPROBE];
	__shared__ int selc[NDBHIP_MAX_NPROBE];
	const uint32_t q = blockIdx.x;
	const uint32_t tid = threadIdx.x;
	const float *d = cdist + (size_t) q * cstride;
	int			npr_eff = npr < ncmp ? npr : ncmp;
	uint32_t	T, m_less, kk, cnt_eq;

	if (npr_eff < 0)
		npr_eff = 0;
	/* valid = strictly below FLT_MAX (bestDist starts at FLT_MAX: ivf_am.c:1689, 1706) */
	auto		ld = [&](uint32_t i, uint32_t &bits) -> bool {
		const float v = d[i];

		bits = __float_as_uint(v);
		return v < FLT_MAX;
	};

	if (ncmp <= 2048)
	{
		/* few centroids (the reference's build fits them on ONE page: <= 185 at dim 4): sort all of them by
		 * (distance, index) in LDS — the nprobe-times "first strict minimum" selection (:1685-1714) is the
		 * head of that order — instead of four histogram passes with their barriers */
		uint32_t	np2 = 2;

		while (np2 < (uint32_t) ncmp)
			np2 <<= 1;
		if (tid == 0)
			sh[0] = 0;
		__syncthreads();
		for (uint32_t j = tid; j < np2; j += blockDim.x)
		{
			uint32_t	bits = 0;
			const bool	ok = j < (uint32_t) ncmp && ld(j, bits);

			full[j] = ok ? (((uint64_t) ndb_key_from_bits(bits) << 32) | j) : ~0ull;
			if (ok)
				atomicAdd(&sh[0], 1u);
		}
		for (uint32_t size = 2; size <= np2; size <<= 1)
			for (uint32_t sd = size >> 1; sd > 0; sd >>= 1)
			{
				__syncthreads();
				for (uint32_t t = tid; t < (np2 >> 1); t += blockDim.x)
				{
					const uint32_t lo = 2 * t - (t & (sd - 1));
					const uint32_t hi = lo + sd;
					const bool	up = ((lo & size) == 0);
					const uint64_t a = full[lo], b = full[hi];

					if ((a > b) == up)
					{
						full[lo] = b;
						full[hi] = a;
					}
				}
			}
		__syncthreads();
		kk = min((uint32_t) npr_eff, sh[0]);
		for (uint32_t j = tid; j < kk; j += blockDim.x)
			perm[j] = (uint32_t) full[j];
		__syncthreads();
	}
	else
	{
	block_radix_select(ld, (uint32_t) ncmp, (uint32_t) npr_eff, hist, sh, T, m_less, kk, cnt_eq);

	uint32_t	npad = 1;

	while (npad < kk)
		npad <<= 1;
	for (uint32_t j = tid; j < npad; j += blockDim.x)
	{
		comp[j] = ~0ull;
		perm[j] = 0;
	}
	__syncthreads();
	if (kk > 0)
	{
		auto		emit = [&](int cls, uint32_t rank, uint32_t i, uint32_t bits) {
			const uint32_t slot = cls ? (m_less + rank) : rank;

			comp[slot] = ((uint64_t) ndb_key_from_bits(bits) << 32) | i;
			perm[slot] = i;
		};
		block_ordered_gather(ld, (uint32_t) ncmp, T, kk - m_less, sh, emit);
		block_bitonic_sort(comp, perm, npad);
	}
	__syncthreads();
	}
	for (uint32_t i = tid; i < (uint32_t) npr; i += blockDim.x)
	{
		int			c;

		if (i < kk)
			c = (int) perm[i];
		else if (i < (uint32_t) npr_eff)
			c = -1;
		else
			c = 0;
		probes[(size_t) q * npr + i] = c;
		selc[i] = c;
		lens[i] = (c >= 0 && c < ncent) ? glob_len[c] : 0u;	/* ivf_am.c:1768-1779 */
	}
	__syncthreads();
	if (tid == 0)
	{
		uint64_t	acc = 0, mine = 0;
		uint32_t   *co = cand_off + (size_t) q * (npr + 1);
		uint32_t   *lco = loc_cand_off ? loc_cand_off + (size_t) q * (npr + 1) : nullptr;

		co[0] = 0;
		if (lco)
			lco[0] = 0;
		for (int i = 0; i < npr; i++)
		{
			uint64_t	l = lens[i];

			if (cap > 0 && acc + l > cap)
				l = cap - acc;	/* candidateCount < maxCandidates guards: ivf_am.c:1764, 1793, 1811 */
			acc += l;
			if (l > 0)
				mine += ndb_local_part(l, own_lo, own_len, selc[i]);
			co[i + 1] = (uint32_t) acc;
			if (lco)			/* positions of the rows THIS rank holds (sharded mirrors) */
				lco[i + 1] = (uint32_t) mine;
		}
		(void) counters;		/* summed by k_sum_candidates: three atomics per query on one line serialise */
		(void) dim;
	}
}

/* the tail of k_probe_select alone, for probes chosen elsewhere (another rank): candidates[] offsets of
 * every probe, globally and among the rows held here */
__global__ __launch_bounds__(256) void
k_probe_offsets(const int *__restrict__ probes, uint32_t nq, int npr, int ncent,
				const uint32_t *__restrict__ glob_len, const uint32_t *__restrict__ own_lo,
				const uint32_t *__restrict__ own_len, uint64_t cap, int dim,
				uint32_t *__restrict__ cand_off, uint32_t *__restrict__ loc_cand_off,
				unsigned long long *__restrict__ counters)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;

	if (q >= nq)
		return;
	uint64_t	acc = 0, mine = 0;
	uint32_t   *co = cand_off + (size_t) q * (npr + 1);
	uint32_t   *lco = loc_cand_off ? loc_cand_off + (size_t) q * (npr + 1) : nullptr;

	co[0] = 0;
	if (lco)
		lco[0] = 0;
	for (int i = 0; i < npr; i++)
	{
		const int	c = probes[(size_t) q * npr + i];
		uint64_t	l = (c >= 0 && c < ncent) ? glob_len[c] : 0u;

		if (cap > 0 && acc + l > cap)
			l = cap - acc;
		acc += l;
		if (l > 0)
			mine += ndb_local_part(l, own_lo, own_len, c);
		co[i + 1] = (uint32_t) acc;
		if (lco)
			lco[i + 1] = (uint32_t) mine;
	}
	(void) counters;
	(void) dim;
}

/* counters[0] += candidates of all queries (every rank's view), [1] += those held here, [2] += their bytes:
 * one block over the per-query offset tables the two kernels above leave */
__global__ __launch_bounds__(256) void
k_sum_candidates(const uint32_t *__restrict__ cand_off, const uint32_t *__restrict__ loc_cand_off, uint32_t nq, int npr,
				 int row_bytes, unsigned long long *__restrict__ counters)
{
	__shared__ unsigned long long pa[4], pm[4];
	unsigned long long a = 0, m = 0;

	for (uint32_t q = threadIdx.x; q < nq; q += 256)
	{
		a += cand_off[(size_t) q * (npr + 1) + npr];
		m += (loc_cand_off ? loc_cand_off : cand_off)[(size_t) q * (npr + 1) + npr];
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		a += ((unsigned long long) (uint32_t) __shfl_xor((uint32_t) (a >> 32), off, 64) << 32) | (uint32_t) __shfl_xor((uint32_t) a, off, 64);
		m += ((unsigned long long) (uint32_t) __shfl_xor((uint32_t) (m >> 32), off, 64) << 32) | (uint32_t) __shfl_xor((uint32_t) m, off, 64);
	}
	if ((threadIdx.x & 63) == 0)
	{
		pa[threadIdx.x >> 6] = a;
		pm[threadIdx.x >> 6] = m;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const unsigned long long ta = pa[0] + pa[1] + pa[2] + pa[3], tm = pm[0] + pm[1] + pm[2] + pm[3];

		atomicAdd(&counters[0], ta);
		atomicAdd(&counters[1], tm);
		atomicAdd(&counters[2], tm * (unsigned long long) row_bytes);
	}
}

/* position -> (probe index) : largest p with co[p] <= pos */
__device__ __forceinline__ uint32_t
find_probe(const uint32_t *__restrict__ co, int npr, uint32_t pos)
{
	uint32_t	lo = 0, hi = (uint32_t) npr;	/* invariant: co[lo] <= pos < co[hi] */

	while (hi - lo > 1)
	{
		const uint32_t mid = (lo + hi) >> 1;

		if (co[mid] <= pos)
			lo = mid;
		else
			hi = mid;
	}
	return lo;
}

/*
 * HOT LOOP 2: score every entry of every probed list (ivf_am.c:1810-1834).
 * candidates[] position pos = cand_off[p] + index inside list probes[p].
 * grid = (ceil(stride / 256), nq), block = 256 = 4 independent 64-row tiles.
 * Writes dist[q * stride + local pos]; on a sharded mirror the positions count only
 * the rows held here (loc_cand_off), so a rank's scan and top-k cost what its lists cost.
 */
template <int R>
__global__ __launch_bounds__(256) void
k_ivf_scan(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
		   const uint32_t *__restrict__ loc_cand_off, int npr, float *__restrict__ dist, uint32_t stride)
{
	__shared__ __attribute__((aligned(16))) float tiles[4 * NDB_TILE_FLOATS];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t q = blockIdx.y;
	const uint32_t *co = loc_cand_off + (size_t) q * (npr + 1);	/* rows held here, in candidates[] order */
	const uint32_t total = co[npr];
	const uint32_t pos0 = (blockIdx.x * 4 + wave) * 64;

	if (pos0 >= total)
		return;
	const uint32_t pos = pos0 + lane;
	const bool	valid = pos < total;
	const uint32_t spos = valid ? pos : (total - 1);
	const uint32_t p = find_probe(co, npr, spos);
	const int	L = probes[(size_t) q * npr + p];
	const uint32_t row = (uint32_t) (ix.loc_off[L] + (spos - co[p]));
	const float d = score_rows<R>(queries + (size_t) q * ix.dim, ix.vecs, row, ix.dim,
								  tiles + wave * NDB_TILE_FLOATS);

	if (valid)
		dist[(size_t) q * stride + pos] = d;
}

/* the same for fp16 rows (halfvec columns) */
template <int R>
__global__ __launch_bounds__(256) void
k_ivf_scan_h(IvfDev ix, const float *__restrict__ queries, const int *__restrict__ probes,
		   const uint32_t *__restrict__ loc_cand_off, int npr, float *__restrict__ dist, uint32_t stride)
{
	__shared__ __attribute__((aligned(16))) float tiles[4 * NDB_TILE_FLOATS];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t q = blockIdx.y;
	const uint32_t *co = loc_cand_off + (size_t) q * (npr + 1);	/* rows held here, in candidates[] order */
	const uint32_t total = co[npr];
	const uint32_t pos0 = (blockIdx.x * 4 + wave) * 64;

	if (pos0 >= total)
		return;
	const uint32_t pos = pos0 + lane;
	const bool	valid = pos < total;
	const uint32_t spos = valid ? pos : (total - 1);
	const uint32_t p = find_probe(co, npr, spos);
	const int	L = probes[(size_t) q * npr + p];
	const uint32_t row = (uint32_t) (ix.loc_off[L] + (spos - co[p]));
	const float d = score_rows_f16<R>(queries + (size_t) q * ix.dim, ix.vecs, row, ix.dim,
								  tiles + wave * NDB_TILE_FLOATS);

	if (valid)
		dist[(size_t) q * stride + pos] = d;
}

/* ------------------------------------------------------------------ */
/* Query-grouped list scan.  In a batch many queries probe the same list
 * (nq * nprobe / nlists on average), so the (query, probe) pairs are bucketed
 * by list and one work item = (list, 64-row tile, group of <= NDB_QG queries):
 * the row chunk is staged ONCE into registers and every query of the group is
 * accumulated against it with its own register accumulator.  Each (row, query)
 * sum is still the reference's sequential chain, so the distances are the same
 * bits as k_ivf_scan's; HBM/fabric traffic drops by the group size and the
 * kernel becomes bound by the fp32 vector ALU instead of HBM.
 * Requires dim % 64 == 0 (otherwise the per-query kernel is used).          */
/* ------------------------------------------------------------------ */
#define NDB_QG 16
#define NDB_QHEAD_STRIDE 32u		/* words between the scan's work-queue heads: one 128-byte line each */
#ifndef NDB_COOP2_WAVES
#define NDB_COOP2_WAVES 5		/* measured: 4 -> 12.1 ms, 5 -> 11.0 ms, 6 (45 scratch spills) -> 11.7 ms per 4096 queries */
#endif
#ifndef NDB_G16_WAVES
#define NDB_G16_WAVES 8
#endif
#ifndef NDB_G32_WAVES
#define NDB_G32_WAVES 5
#endif
#ifndef NDB_GROUPED_WAVES_PER_SIMD
#define NDB_GROUPED_WAVES_PER_SIMD 3		/* caps the kernel at 168 VGPRs; LDS (16 KiB/wave) allows 10 waves/CU */
#endif

struct PairRec
{
	uint32_t	q;				/* query index inside the sub-batch */
	uint32_t	p;				/* probe index */
};

/* pass 1: how many (query, probe) pairs hit each owned list */
__global__ void
k_pair_count(const int *__restrict__ probes, const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t nq,
			 uint32_t *__restrict__ cnt, const unsigned int *__restrict__ active = nullptr)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;

	if (i >= nq * (uint32_t) npr)
		return;
	const uint32_t q = i / npr, p = i % npr;
	const uint32_t *co = loc_cand_off + (size_t) q * (npr + 1);	/* rows held HERE for this (query, probe) */

	if (co[p + 1] == co[p] || (active && !active[q]))
		return;
	atomicAdd(&cnt[probes[(size_t) q * npr + p]], 1u);
}

/* pass 2 (one block of 1024 threads): per-list pair / work-item / group offsets */
__global__ __launch_bounds__(1024) void
k_pair_offsets(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ glob_len, int ncent,
			   uint32_t *__restrict__ pair_off, uint32_t *__restrict__ item_off,
			   uint32_t *__restrict__ grp_off, uint32_t *__restrict__ runs, uint32_t gdiv, uint32_t rt)
{
	__shared__ uint32_t sa[1024], sb[1024], sc[1024];
	const int	t = threadIdx.x;
	const int	per = (ncent + 1023) / 1024;
	const int	l0 = t * per, l1 = min(ncent, l0 + per);
	uint32_t	a = 0, b = 0, c2 = 0;

	for (int L = l0; L < l1; L++)
	{
		const uint32_t c = cnt[L];
		const uint32_t ng = (c + NDB_QG - 1) / NDB_QG;

		a += c;
		b += ((((glob_len[L] + 63u) >> 6) + rt - 1u) / rt) * ((ng + gdiv - 1u) / gdiv);
		c2 += ng;
	}
	sa[t] = a;
	sb[t] = b;
	sc[t] = c2;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1)
	{
		const uint32_t va = (t >= off) ? sa[t - off] : 0u;
		const uint32_t vb = (t >= off) ? sb[t - off] : 0u;
		const uint32_t vc = (t >= off) ? sc[t - off] : 0u;

		__syncthreads();
		sa[t] += va;
		sb[t] += vb;
		sc[t] += vc;
		__syncthreads();
	}
	a = sa[t] - a;				/* exclusive prefix of this thread's first list */
	b = sb[t] - b;
	c2 = sc[t] - c2;
	for (int L = l0; L < l1; L++)
	{
		const uint32_t c = cnt[L];
		const uint32_t ng = (c + NDB_QG - 1) / NDB_QG;

		pair_off[L] = a;
		item_off[L] = b;
		grp_off[L] = c2;
		a += c;
		b += ((((glob_len[L] + 63u) >> 6) + rt - 1u) / rt) * ((ng + gdiv - 1u) / gdiv);
		c2 += ng;
	}
	if (t == 1023)
	{
		pair_off[ncent] = sa[1023];
		item_off[ncent] = sb[1023];
		grp_off[ncent] = sc[1023];
	}
	/* the scan's 8 work queues (one per XCD): runs of whole lists with about the same number of items;
	 * runs[x] = first item of run x, runs[8] = nitems.  Computed once here instead of by every block. */
	__threadfence_block();
	__syncthreads();
	if (t <= 8)
	{
		const uint32_t nitems = sb[1023];
		uint32_t	r = t == 0 ? 0u : nitems;

		if (t > 0 && t < 8)
		{
			const uint32_t target = (uint32_t) (((uint64_t) nitems * (uint32_t) t) >> 3);
			uint32_t	lo = 0, hi = (uint32_t) ncent;	/* smallest L with item_off[L] >= target */

			while (lo < hi)
			{
				const uint32_t mid = (lo + hi) >> 1;

				if (__hip_atomic_load(&item_off[mid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
					lo = mid + 1;
				else
					hi = mid;
			}
			r = __hip_atomic_load(&item_off[lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		runs[t] = r;
	}
}

/* pass 3: bucket the pairs */
__global__ void
k_pair_fill(const int *__restrict__ probes, const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t nq,
			const uint32_t *__restrict__ pair_off,
			uint32_t *__restrict__ fill, PairRec *__restrict__ pairs, const unsigned int *__restrict__ active = nullptr)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;

	if (i >= nq * (uint32_t) npr)
		return;
	const uint32_t q = i / npr, p = i % npr;
	const uint32_t *co = loc_cand_off + (size_t) q * (npr + 1);

	if (co[p + 1] == co[p] || (active && !active[q]))
		return;
	const int	L = probes[(size_t) q * npr + p];
	const uint32_t slot = pair_off[L] + atomicAdd(&fill[L], 1u);
	PairRec		r;

	r.q = q;
	r.p = p;
	pairs[slot] = r;
}

/*
 * pass 4: interleave every group's queries per dimension:
 *   qblock[group][d][j] = queries[member j][d]   (j < 16; short groups are padded with member 0)
 * so that ONE s_load_dwordx16 brings the 16 queries' values of a dimension and the
 * arithmetic runs on query pairs with packed fp32 instructions.
 * thread = (group, d): 16 coalesced reads (one per member), one 64-byte write.
 */
__global__ void
k_group_pack(const float *__restrict__ queries, int dim, int ncent, const uint32_t *__restrict__ cnt,
			 const uint32_t *__restrict__ pair_off, const uint32_t *__restrict__ grp_off,
			 const PairRec *__restrict__ pairs, float *__restrict__ qblock)
{
	const uint32_t grp = blockIdx.y;
	const uint32_t ngroups = grp_off[ncent];

	if (grp >= ngroups)
		return;
	uint32_t	lo = 0, hi = (uint32_t) ncent;

	while (hi - lo > 1)
	{
		const uint32_t mid = (lo + hi) >> 1;

		if (grp_off[mid] <= grp)
			lo = mid;
		else
			hi = mid;
	}
	/* lists without pairs share their offset with the next one: skip forward to the owner */
	while (lo + 1 < (uint32_t) ncent && grp_off[lo + 1] <= grp)
		lo++;
	const uint32_t L = lo;
	const uint32_t g0 = (grp - grp_off[L]) * NDB_QG;
	const uint32_t nmem = min((uint32_t) NDB_QG, cnt[L] - g0);
	const PairRec *mem = pairs + pair_off[L] + g0;
	const int	d = blockIdx.x * blockDim.x + threadIdx.x;

	if (d >= dim)
		return;
	float		v[NDB_QG];

#pragma unroll
	for (int j = 0; j < NDB_QG; j++)
	{
		const uint32_t qid = mem[(uint32_t) j < nmem ? j : 0].q;

		v[j] = queries[(size_t) qid * dim + d];
	}
	float4	   *dst = reinterpret_cast<float4 *>(qblock + ((size_t) grp * dim + d) * NDB_QG);

#pragma unroll
	for (int j = 0; j < NDB_QG / 4; j++)
		dst[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

/* compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) */
template <int B, int E, class F>
__device__ __forceinline__ void
ndb_static_for(F &&f)
{
	if constexpr (B < E)
	{
		f(std::integral_constant<int, B>{});
		ndb_static_for<B + 1, E>(f);
	}
}

typedef float ndb_f2 __attribute__((ext_vector_type(2)));
typedef float ndb_f16 __attribute__((ext_vector_type(16)));

/*
 * Scalar (SMEM) loads the compiler does not schedule: hipcc sinks every s_load next to its
 * first use and then waits lgkmcnt(0) — the full scalar-cache latency once per dimension.
 * These helpers issue the loads early and wait late (cdna_hip_programming.md 5.7 form ii:
 * "=s" loads, one wait statement that names every destination "+s").  SMEM returns out of
 * order, so the only legal wait is lgkmcnt(0): the pipeline is "wait current -> issue next ->
 * compute current".
 */
__device__ __forceinline__ void
sload2x16(ndb_f16 &a, ndb_f16 &b, const float *p)
{
	asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40"
				 : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
}

/* the same at a compile-time byte offset from one base pointer: no per-batch address arithmetic (hipcc
 * materialised, and then spilled to VGPR lanes, a 64-bit address per batch) */
template <int OFF>
__device__ __forceinline__ void
sload2x16_at(ndb_f16 &a, ndb_f16 &b, const float *base)
{
	asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"
				 : "=&s"(a), "=&s"(b) : "s"(base), "n"(OFF), "n"(OFF + 64) : "memory");
}

__device__ __forceinline__ void
swait2(ndb_f16 &a, ndb_f16 &b)
{
	asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) :: "memory");
}

/* accumulators of 16 queries as 8 packed pairs; STEP = one dimension for all 16 queries */
template <int R> struct GAcc;

template <> struct GAcc<R_IVF_L2>
{
	ndb_f2		s[NDB_QG / 2];
	__device__ __forceinline__ void init()
	{
#pragma unroll
		for (int i = 0; i < NDB_QG / 2; i++)
			s[i] = (ndb_f2) (0.0f);
	}
	__device__ __forceinline__ void step(const ndb_f16 &q, float x)
	{
		const ndb_f2 xx = (ndb_f2) (x);

		/* four pairs at a time, phase by phase: a packed multiply must not be followed directly by the add
		 * that consumes it (one wait state on gfx950), and with a single temporary hipcc pads every such
		 * pair with s_nop (406 per 768 packed ops in the chunk loop) */
#pragma unroll
		for (int i0 = 0; i0 < NDB_QG / 2; i0 += 4)
		{
			ndb_f2		d[4];

#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				ndb_f2		qp;

				qp.x = q[2 * (i0 + u)];
				qp.y = q[2 * (i0 + u) + 1];
				d[u] = qp - xx;
			}
#pragma unroll
			for (int u = 0; u < 4; u++)
				d[u] = d[u] * d[u];
#pragma unroll
			for (int u = 0; u < 4; u++)
				s[i0 + u] = s[i0 + u] + d[u];
		}
	}
	__device__ __forceinline__ float fin(int j, float) const
	{
		return __builtin_sqrtf((j & 1) ? s[j >> 1].y : s[j >> 1].x);
	}
};

template <> struct GAcc<R_IVF_IP>
{
	ndb_f2		s[NDB_QG / 2];
	__device__ __forceinline__ void init()
	{
#pragma unroll
		for (int i = 0; i < NDB_QG / 2; i++)
			s[i] = (ndb_f2) (0.0f);
	}
	__device__ __forceinline__ void step(const ndb_f16 &q, float x)
	{
		const ndb_f2 xx = (ndb_f2) (x);

#pragma unroll
		for (int i0 = 0; i0 < NDB_QG / 2; i0 += 4)	/* products first, sums after: see GAcc<R_IVF_L2>::step */
		{
			ndb_f2		d[4];

#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				ndb_f2		qp;

				qp.x = q[2 * (i0 + u)];
				qp.y = q[2 * (i0 + u) + 1];
				d[u] = qp * xx;
			}
#pragma unroll
			for (int u = 0; u < 4; u++)
				s[i0 + u] = s[i0 + u] + d[u];
		}
	}
	__device__ __forceinline__ float fin(int j, float) const
	{
		return -((j & 1) ? s[j >> 1].y : s[j >> 1].x);
	}
};

/* cosine (ivf_am.c:1570-1581): dot per (row, query) as packed pairs, the row's own norm chain in the
 * lane, the query's norm chain precomputed once per query (k_query_norms) — three independent
 * sequential fp32 chains, exactly the reference's */
template <> struct GAcc<R_IVF_COS>
{
	ndb_f2		s[NDB_QG / 2];
	float		n2;
	__device__ __forceinline__ void init()
	{
#pragma unroll
		for (int i = 0; i < NDB_QG / 2; i++)
			s[i] = (ndb_f2) (0.0f);
		n2 = 0.0f;
	}
	__device__ __forceinline__ void step(const ndb_f16 &q, float x)
	{
		const ndb_f2 xx = (ndb_f2) (x);

#pragma unroll
		for (int i0 = 0; i0 < NDB_QG / 2; i0 += 4)	/* products first, sums after: see GAcc<R_IVF_L2>::step */
		{
			ndb_f2		d[4];

#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				ndb_f2		qp;

				qp.x = q[2 * (i0 + u)];
				qp.y = q[2 * (i0 + u) + 1];
				d[u] = qp * xx;
			}
#pragma unroll
			for (int u = 0; u < 4; u++)
				s[i0 + u] = s[i0 + u] + d[u];
		}
		n2 = n2 + x * x;
	}
	__device__ __forceinline__ float fin(int j, float n1) const
	{
		const float dot = (j & 1) ? s[j >> 1].y : s[j >> 1].x;
		const float a = __builtin_sqrtf(n1);
		const float b = __builtin_sqrtf(n2);

		if (a == 0.0f || b == 0.0f)
			return 1.0f;
		return 1.0f - (dot / (a * b));
	}
};

/*
 * Screening recipe for L2 (not a reference recipe: a BOUND on one).  Per (row, query) one fused
 * multiply-add per dimension instead of subtract / multiply / add: dot += q * x, and the row's norm
 * n2 += x * x once per row.  a = (|q|^2 + n2) - 2 dot approximates the squared distance with
 *   |a - D| <= E = gamma_(dim+8) * 2 * (|q|^2 + |x|^2)          (D = the real squared distance)
 * (sequential-FMA dot: gamma_dim * sum |q_i x_i| <= gamma_dim * |q||x|; the two norm chains gamma_dim each;
 * three roundings to combine), so l = a - E is a LOWER bound of D and l + 2E an upper bound.  The kernel
 * stores sqrt(max(l, 0)) rounded down as the candidate's provisional distance; k_ivf_rescore then replaces
 * it by the reference's own sequential sqrtf(sum (q - x)^2) for every candidate that can still be among the
 * k nearest, and the top-k runs on that: ids, ranks and float4 bits are the exact path's (proof in DESIGN.md).
 */
template <> struct GAcc<R_SCR_L2>
{
	ndb_f2		s[NDB_QG / 2];
	float		n2;
	__device__ __forceinline__ void init()
	{
#pragma unroll
		for (int i = 0; i < NDB_QG / 2; i++)
			s[i] = (ndb_f2) (0.0f);
		n2 = 0.0f;
	}
	__device__ __forceinline__ void step(const ndb_f16 &q, float x)
	{
		const ndb_f2 xx = (ndb_f2) (x);

#pragma unroll
		for (int i = 0; i < NDB_QG / 2; i++)
		{
			ndb_f2		qp;

			qp.x = q[2 * i];
			qp.y = q[2 * i + 1];
			s[i] = __builtin_elementwise_fma(qp, xx, s[i]);
		}
		n2 = __builtin_fmaf(x, x, n2);
	}
	/* lower bound of the distance, in the distance's own domain: qn = |q|^2, e = E of this query */
	__device__ __forceinline__ float bound(int j, float qn, float e) const
	{
		const float dot = (j & 1) ? s[j >> 1].y : s[j >> 1].x;
		const float a = (qn + n2) - 2.0f * dot;
		const float l = a - e;

		return __builtin_sqrtf(fmaxf(l, 0.0f) * 0.99999905f);	/* 1 - 2^-20: sqrtf's own rounding stays below */
	}
	/* the same with the row's norm supplied (k_ivf_bound_coop2 reads it from the per-row norms the mirror keeps
	 * for the bound's constant: one instruction per dimension less) */
	__device__ __forceinline__ void step_dot(const ndb_f16 &q, float x)
	{
		const ndb_f2 xx = (ndb_f2) (x);

#pragma unroll
		for (int i = 0; i < NDB_QG / 2; i++)
		{
			ndb_f2		qp;

			qp.x = q[2 * i];
			qp.y = q[2 * i + 1];
			s[i] = __builtin_elementwise_fma(qp, xx, s[i]);
		}
	}
	__device__ __forceinline__ float bound_n2(int j, float qn, float e, float rn2) const
	{
		const float dot = (j & 1) ? s[j >> 1].y : s[j >> 1].x;
		const float a = (qn + rn2) - 2.0f * dot;
		const float l = a - e;

		return __builtin_sqrtf(fmaxf(l, 0.0f) * 0.99999905f);
	}
	/* inner product (the -dot recipe): the fused and the reference's unfused chain are both within
	 * gamma_dim |q||x| of the real dot product, so -dot - e with e >= gamma (|q|^2 + |x|^2) is a lower bound
	 * of the reference's value; rounded down */
	__device__ __forceinline__ float bound_ip(int j, float e) const
	{
		const float dot = (j & 1) ? s[j >> 1].y : s[j >> 1].x;
		const float l = -dot - e;

		return l - fabsf(l) * 2.4e-7f - 1e-37f;
	}
	/* cosine: the norms ARE the reference's (same sequential chains, bit for bit), only the dot product differs
	 * by at most gamma_dim |q||x|, i.e. gamma_dim in the quotient; e = 4 gamma_(dim+8) also covers the quotient's
	 * own roundings.  A zero norm is the reference's exact 1.0 */
	__device__ __forceinline__ float bound_cos(int j, float qn, float rn2, float e) const
	{
		const float dot = (j & 1) ? s[j >> 1].y : s[j >> 1].x;
		const float a = __builtin_sqrtf(qn), b = __builtin_sqrtf(rn2);
		const float c = (a == 0.0f || b == 0.0f) ? 1.0f : 1.0f - (dot / (a * b));
		const float l = c - e;

		return l - fabsf(l) * 2.4e-7f - 1e-37f;
	}
	__device__ __forceinline__ float fin(int, float) const { return 0.0f; }
};

/* norm1 of every query: `norm1 += vec1[i] * vec1[i]` in dimension order (ivf_am.c:1574) */
__global__ void
k_query_norms(const float *__restrict__ queries, uint32_t nq, int dim, float *__restrict__ out)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;

	if (q >= nq)
		return;
	const float *v = queries + (size_t) q * dim;
	float		n1 = 0.0f;

	for (int i = 0; i < dim; i++)
		n1 = n1 + v[i] * v[i];
	out[q] = n1;
}

/*
 * Persistent kernel: every wave pulls work items (list, 64-row tile, query group)
 * from a global counter.  block = 256 (4 independent waves, 16 KiB LDS tile each).
 */
/* H16: 0 = float4 rows, 1 = fp16 rows decoded like fp16_to_float incl. the subnormal quirk (Q20), 2 = fp16 rows
 * of a mirror that holds no subnormal (the hardware conversion alone is exact there) */
template <int R, int CH, int H16>
__global__ __launch_bounds__(64, (H16 ? 4 : (CH == 16 ? NDB_G16_WAVES : (CH == 32 ? NDB_G32_WAVES : NDB_GROUPED_WAVES_PER_SIMD)))) void
k_ivf_scan_grouped(IvfDev ix, const float *__restrict__ qblock, const uint32_t *__restrict__ cand_off,
				   const uint32_t *__restrict__ loc_cand_off, int npr, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ pair_off,
				   const uint32_t *__restrict__ item_off, const uint32_t *__restrict__ grp_off,
				   const PairRec *__restrict__ pairs, unsigned int *__restrict__ next_item,
				   const uint32_t *__restrict__ runs, float *__restrict__ dist, uint32_t stride,
				   const float *__restrict__ qnorm, uint32_t *__restrict__ tmin, uint32_t tstride, int polite,
				   uint32_t nq_all)
{
	__shared__ __attribute__((aligned(16))) float tile[64 * CH];
	const int	lane = threadIdx.x & 63;
	const int	dim = ix.dim;
	/*
	 * XCD-aware work queues.  The items (list-major) are cut into 8 runs of whole lists with about the same
	 * number of items, one per XCD; a block serves the run of ITS XCD first (block b runs on XCD b % 8 —
	 * observed, used for speed only) and then helps the others.  All query groups of a list are therefore
	 * scored through one XCD's L2, and — consecutive items being the same row tile for consecutive groups —
	 * at about the same time: a tile comes from HBM once, not once per group.
	 */
	/* next_item: 8 queue heads, one per 128-byte line (NDB_QHEAD_STRIDE words apart: every block of the grid
	 * hits them, and device-scope atomics on one line serialise); runs: the 9 run bounds k_pair_offsets left
	 * (read-only here and away from the hot lines) */
	for (uint32_t hop = 0; hop < 8; hop++)
	{
	const uint32_t xq = (blockIdx.x + hop) & 7u;
	const uint32_t run_lo = runs[xq], run_hi = runs[xq + 1];

	if (run_lo == run_hi)
		continue;

	for (;;)
	{
		uint32_t	item = 0;

		/* polite (small batches): look before taking.  A small batch has fewer items than the grid has blocks;
		 * 8 failing read-modify-writes per block on eight hot lines then cost more than the scan itself (0.54 ms
		 * for 8 queries), and a load does not serialise on the line like a read-modify-write does.  Large
		 * batches skip the look: under load it queues behind the other blocks' atomics (4 % of the step). */
		if (lane == 0)
			item = (polite && run_lo + __hip_atomic_load(next_item + xq * NDB_QHEAD_STRIDE, __ATOMIC_RELAXED,
														  __HIP_MEMORY_SCOPE_AGENT) >= run_hi)
				? run_hi : run_lo + atomicAdd(next_item + xq * NDB_QHEAD_STRIDE, 1u);
		item = __builtin_amdgcn_readfirstlane(item);
		if (item >= run_hi)
			break;
		/* list of this item: the L with item_off[L] <= item < item_off[L+1] */
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
		const uint32_t len = ix.own_len[L];	/* the rows of the list held here */
		const uint32_t local = item - item_off[L];
		/* consecutive items = the same 64-row tile for the list's consecutive query groups: the waves that
		 * pull them stream the same rows at about the same time, so all but the first find them in cache */
		const uint32_t ngrp = (cnt[L] + NDB_QG - 1) / NDB_QG;
		const uint32_t gi = local % ngrp;
		const uint32_t t = local / ngrp;
		const uint32_t g0 = gi * NDB_QG;
		const uint32_t nmem = min((uint32_t) NDB_QG, cnt[L] - g0);
		const PairRec *mem = pairs + pair_off[L] + g0;
		const float *__restrict__ qb = qblock + (size_t) (grp_off[L] + gi) * (size_t) dim * NDB_QG;
		const uint32_t ridx = t * 64 + lane;
		const uint32_t row = (uint32_t) ix.loc_off[L] + (ridx < len ? ridx : len - 1);
		uint32_t	rowsN[CH / 4];
		GAcc<R>		acc;

		acc.init();
		rows_for_loads<CH>(rowsN, row, lane);

		/* query stream of this group: [dim][16] floats, consumed 2 dimensions (128 B) per batch,
		 * double buffered in SGPRs: A = even batch, B = odd batch */
		const float *qs = qb;
		ndb_f16		qa0, qa1, qb0, qb1;

		asm volatile("s_nop 4" ::: "memory");	/* the base pointer may come from v_readfirstlane */
		sload2x16(qa0, qa1, qs);
		if constexpr (H16)
		{
			/* fp16 rows: 64 dimensions (128 raw bytes per row) per step */
			for (int c = 0; c < dim; c += 64)
			{
				float4		raw[8];

				stage_chunk_w<32>(raw, ix.vecs, rowsN, dim >> 1, c >> 1, tile, lane);
				const float *qnext = (c + 64 >= dim) ? qs - 2 * NDB_QG : qs;

				ndb_static_for<0, 8>([&](auto pc) {
					constexpr int p = decltype(pc)::value;
					float		x[8];

					decode8<H16 == 1>(raw[p], x);
					swait2(qa0, qa1);
					sload2x16_at<(8 * p + 2) * 64>(qb0, qb1, qs);
					acc.step(qa0, x[0]);
					acc.step(qa1, x[1]);
					swait2(qb0, qb1);
					sload2x16_at<(8 * p + 4) * 64>(qa0, qa1, qs);
					acc.step(qb0, x[2]);
					acc.step(qb1, x[3]);
					swait2(qa0, qa1);
					sload2x16_at<(8 * p + 6) * 64>(qb0, qb1, qs);
					acc.step(qa0, x[4]);
					acc.step(qa1, x[5]);
					swait2(qb0, qb1);
					if constexpr (p == 7)
						sload2x16_at<64 * 64>(qa0, qa1, qnext);
					else
						sload2x16_at<(8 * p + 8) * 64>(qa0, qa1, qs);
					acc.step(qb0, x[6]);
					acc.step(qb1, x[7]);
				});
				qs += 64 * NDB_QG;
			}
		}
		else
		{
		for (int c = 0; c < dim; c += CH)
		{
			float4		x[CH / 4];

			stage_chunk_w<CH>(x, ix.vecs, rowsN, dim, c, tile, lane);
			/* the chunk's query values sit at fixed byte offsets from qs (16 queries x 4 B = 64 B per
			 * dimension); the batch issued last belongs to the next chunk — on the last chunk it re-reads
			 * this group's final 128 B instead of running past the block */
			const float *qnext = (c + CH >= dim) ? qs - 2 * NDB_QG : qs;

			ndb_static_for<0, CH / 4>([&](auto pc) {
				constexpr int p = decltype(pc)::value;

				/* dims 4p, 4p+1 from A; 4p+2, 4p+3 from B */
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
		}
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
				const uint32_t nrow = lq[pp + 1] - la;	/* may be capped below len (ivf_am.c:1743) */

				float		dv;

				if constexpr (R == R_SCR_L2)
					dv = acc.bound(j, qnorm[qid], qnorm[nq_all + qid]);	/* [|q|^2 ... | E ...] */
				else
					dv = acc.fin(j, R == R_IVF_COS ? qnorm[qid] : 0.0f);

				if (ridx < nrow)
					dist[(size_t) qid * stride + la + ridx] = dv;
				/* the smallest order key of this (query, 64-candidate tile): k_ivf_topk bounds the k-th
				 * candidate with these and then only opens the tiles that can hold one */
				uint32_t	mk = ridx < nrow ? ndb_key_from_bits(__float_as_uint(dv)) : 0xFFFFFFFFu;

#pragma unroll
				for (int off = 32; off > 0; off >>= 1)
					mk = min(mk, (uint32_t) __shfl_xor((int) mk, off, 64));
				if (lane == 0 && t * 64u < nrow)	/* a tile wholly beyond a capped list (:1743) has no slot */
					tmin[(size_t) qid * tstride + (la >> 6) + pp + t] = mk;
			}
		}
	}
	}
}

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
/* counters[3] += sum of v[0..n): one block */
__global__ __launch_bounds__(256) void
k_sum_u32(const unsigned int *__restrict__ v, uint32_t n, unsigned long long *__restrict__ out)
{
	__shared__ unsigned long long part[4];
	unsigned long long s = 0;

	for (uint32_t i = threadIdx.x; i < n; i += 256)
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
#define NDB_TOPK_FAST_CAP 1024		/* candidates <= U the fast path can hold before falling back */

__host__ __device__ static inline uint32_t
topk_entry_cap(uint32_t k)
{
	return (k <= NDB_TOPK_FAST_MAXK && 3 * k < NDB_TOPK_FAST_CAP) ? NDB_TOPK_FAST_CAP : 3 * k;
}

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
	auto		tid_of = [&](uint32_t i0, uint32_t &gpos) -> uint64_t {
		const uint32_t i = i0 + lo;
		const uint32_t p = find_probe(lco, npr, i);
		const int	L = probes[(size_t) q * npr + p];

		gpos = co[p] + ix.own_lo[L] + (i - lco[p]);	/* a split list: this mirror starts at position own_lo */
		return ix.tids[ix.loc_off[L] + (i - lco[p])];
	};

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

		s.fs.comp[tid] = ((uint64_t) mn << 32) | tid;
		s.fs.perm[tid] = tid;
		block_bitonic_sort(s.fs.comp, s.fs.perm, 256);
		const uint32_t U = (nth >= k) ? (uint32_t) (s.fs.comp[k - 1] >> 32) : 0xFFFFFFFEu;	/* 0xFFFFFFFF = empty slot */
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

					if ((lco[mid] >> 6) + mid <= sidx)
						lo2 = mid;
					else
						hi2 = mid;
				}
				const uint32_t base = lco[lo2] + ((sidx - ((lco[lo2] >> 6) + lo2)) << 6);
				const uint32_t i = base + lane;

				if (base < lco[lo2 + 1] && i < lco[lo2 + 1])
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

		s.fs.comp[tid] = ((uint64_t) mn << 32) | tid;
		s.fs.perm[tid] = tid;
		block_bitonic_sort(s.fs.comp, s.fs.perm, 256);
		/* U: the k-th smallest thread minimum bounds the k-th smallest candidate (the k smallest
		 * minima are k distinct candidates <= U); with fewer than k non-empty threads gather all */
		const uint32_t U = (nth >= k) ? (uint32_t) (s.fs.comp[k - 1] >> 32) : 0xFFFFFFFFu;
		__syncthreads();

		/* pass 2: gather every candidate with key <= U */
		if (tid == 0)
			s.sh[0] = 0;
		__syncthreads();
		for (i = tid; i < total; i += 256)
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

	if (threadIdx.x == 0)
	{
		uint32_t	acc = 0;

		for (int w = 0; w < world; w++)
		{
			/* a count from a peer is data, not a promise: more than `cap` records (or a negative count) would
			 * overrun the LDS arrays sized for world x cap */
			const int	nc_w = ncand[(size_t) w * nq + q];

			woff[w] = acc;
			acc += (uint32_t) (nc_w < 0 ? 0 : (nc_w > (int) cap ? (int) cap : nc_w));
		}
		woff[world] = acc;
	}
	__syncthreads();
	const uint32_t n = woff[world];

	for (int w = 0; w < world; w++)
	{
		const uint32_t cnt = woff[w + 1] - woff[w];
		const ndbhip_cand *src = cand + ((size_t) w * nq + q) * cap;

		for (uint32_t j = threadIdx.x; j < cnt; j += blockDim.x)
		{
			const ndbhip_cand c = src[j];

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

#define NDB_TOPK_MAX_SMEM (150 * 1024)

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

/* ================================================================== */
/* host side: IVF mirror                                               */
/* ================================================================== */

struct ndbhip_ivf
{
	int			dim = 0;
	int			nlists = 0;
	int			ncent = 0;
	float	   *d_centroids = nullptr;
	float	   *d_vecs = nullptr;
	uint64_t   *d_tids = nullptr;
	bool		own_rows = false;
	int64_t		nrows = 0;
	int64_t		cap_rows = 0;
	float	   *d_vecs_alt = nullptr;	/* second row buffer of ivf_flush (appends): the new layout is gathered into it */
	uint64_t   *d_tids_alt = nullptr;
	size_t		alt_cap = 0;
	int64_t    *d_loc_off = nullptr;
	uint32_t   *d_glob_len = nullptr;
	uint8_t    *d_owned = nullptr;
	uint32_t   *d_own_lo = nullptr, *d_own_len = nullptr;
	std::vector<int64_t> own_lo, own_len;	/* list positions [own_lo, own_lo + own_len) are held here */
	std::vector<int64_t> glob_len;
	std::vector<int64_t> loc_off;
	std::vector<uint8_t> owned;
	bool		loaded = false;
	bool		sharded = false;		/* some list is not held here */
	bool		f16 = false;			/* rows held as fp16 (halfvec column): d_vecs points at uint16 data */
	bool		f16_sub = true;			/* ... and some element is an fp16 subnormal (decode needs the Q20 fix) */
	int			meta_nprobe = 10;		/* IvfMetaPageData.nprobe (ivf_am.c:75-89), IVF_DEFAULT_NPROBE */
	/* aminsert: entries appended since the last repack (flushed before the next search) */
	std::vector<int> pend_list;
	std::vector<float> pend_rows;
	std::vector<uint64_t> pend_tids;
	/* workspace (grown on demand) */
	float	   *w_cdist = nullptr;	size_t w_cdist_n = 0;
	int		   *w_probes = nullptr;	size_t w_probes_n = 0;
	uint32_t   *w_candoff = nullptr; size_t w_candoff_n = 0;
	float	   *w_dist = nullptr;	size_t w_dist_n = 0;
	float	   *w_q = nullptr;		size_t w_q_n = 0;
	uint64_t   *w_otid = nullptr;	size_t w_otid_n = 0;
	float	   *w_odist = nullptr;	size_t w_odist_n = 0;
	int		   *w_ocnt = nullptr;	size_t w_ocnt_n = 0;
	uint32_t   *w_gcnt = nullptr;	size_t w_gcnt_n = 0;	/* [3*ncent + 3]: cnt, fill, + next_item */
	uint32_t   *w_goff = nullptr;	size_t w_goff_n = 0;	/* [2*(ncent+1)]: pair_off, item_off */
	PairRec    *w_pairs = nullptr;	size_t w_pairs_n = 0;
	float	   *w_qblock = nullptr;	size_t w_qblock_n = 0;	/* [groups][dim][16] interleaved queries */
	float	   *w_qnorm = nullptr;	size_t w_qnorm_n = 0;	/* [2 nq] sum of squares of every query (cosine, screening) | screening E */
	/* screened L2 scan: largest row norm^2 of the rows held here (valid while norm_valid), first-pass top-k scratch */
	float	   *d_xxmax = nullptr;
	bool		norm_valid = false;
	float	   *w_rnorm = nullptr;	size_t w_rnorm_n = 0;
	uint64_t   *w_scrt = nullptr;	size_t w_scrt_n = 0;
	float	   *w_scrd = nullptr;	size_t w_scrd_n = 0;
	int		   *w_scrc = nullptr;	size_t w_scrc_n = 0;
	uint32_t   *w_screc = nullptr;	size_t w_screc_n = 0;	/* survivor records (4 words each) + their count */
	void	   *pin = nullptr;		size_t pin_n = 0;		/* pinned host block of ndbhip_ivf_search: query + results */
	float	   *w_cblock = nullptr;	size_t w_cblock_n = 0;	/* centroids interleaved 16 per block (batch centroid scan) */
	uint32_t   *w_tmin = nullptr;	size_t w_tmin_n = 0;	/* [nq][tstride] smallest order key per 64-candidate tile */
	/* fp16-MFMA screened scan (ndbhip_screen16.h): per-row split planes / norms / exponents, valid while s16_valid */
	unsigned char *d_planes = nullptr; size_t d_planes_n = 0;
	float	   *d_rn2 = nullptr;	size_t d_rn2_n = 0;
	int16_t    *d_rexp = nullptr;	size_t d_rexp_n = 0;
	uint32_t   *d_xmax16 = nullptr;
	uint32_t   *d_blkoff = nullptr;	size_t d_blkoff_n = 0;	/* [ncent + 1] first 32-row block of every list in d_planes */
	bool		s16_valid = false;
	unsigned char *w_qplanes = nullptr; size_t w_qplanes_n = 0;
	float	   *w_qn2 = nullptr;	size_t w_qn2_n = 0;
	int		   *w_qexp = nullptr;	size_t w_qexp_n = 0;
	float2	   *w_qthr = nullptr;	size_t w_qthr_n = 0;
	unsigned int *w_ecount = nullptr; size_t w_ecount_n = 0;	/* [nq] emitted per query | [nq] survivors | [nq] seeds | 4 flags */
	uint2	   *w_erec = nullptr;	size_t w_erec_n = 0;
	uint32_t   *w_s16desc = nullptr; size_t w_s16desc_n = 0;	/* S16Desc per work item of the sweep */
	uint32_t   *w_bmin = nullptr;	size_t w_bmin_n = 0;	/* [nq][S16_NB] smallest emitted a per hash bucket of positions */
	/* split top-k of small batches: per-range records, counts, totals */
	ndbhip_cand *w_scand = nullptr;	size_t w_scand_n = 0;
	int		   *w_sncand = nullptr;	size_t w_sncand_n = 0;
	int64_t    *w_stotal = nullptr;	size_t w_stotal_n = 0;
};

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

/* temporaries of one call: freed on every way out of the scope unless keep() hands one over (the
 * HIP_TRY early returns of ivf_flush / ndbhip_ivf_delete leaked their device buffers) */
struct DevGuard
{
	std::vector<void **> owned;
	template <class T> int alloc(T *&p, size_t bytes)
	{
		p = nullptr;
		if (bytes == 0)
			bytes = 16;
		HIP_TRY(hipMalloc((void **) &p, bytes));
		owned.push_back((void **) &p);
		return 0;
	}
	template <class T> void keep(T *&p)
	{
		for (auto &o : owned)
			if (o == (void **) &p)
				o = nullptr;
	}
	~DevGuard()
	{
		for (auto o : owned)
			if (o && *o)
			{
				(void) hipFree(*o);
				*o = nullptr;
			}
	}
};

extern "C" int
ndbhip_ivf_create(int dim, int nlists, ndbhip_ivf **out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!out)
		return fail(NDBHIP_ERR_INVALID, "out is NULL");
	if (dim < 1 || dim > 32767)
		return fail(NDBHIP_ERR_INVALID, "dim %d out of range 1..32767", dim);
	if (nlists < 1)
		return fail(NDBHIP_ERR_INVALID, "nlists %d must be >= 1", nlists);
	ndbhip_ivf *ix = new (std::nothrow) ndbhip_ivf();

	if (!ix)
		return fail(NDBHIP_ERR_NOMEM, "out of host memory");
	ix->dim = dim;
	ix->nlists = nlists;
	*out = ix;
	return NDBHIP_OK;
}

static void
ivf_free_rows(ndbhip_ivf *ix)
{
	if (ix->own_rows)
	{
		if (ix->d_vecs) (void) hipFree(ix->d_vecs);
		if (ix->d_tids) (void) hipFree(ix->d_tids);
	}
	if (ix->d_vecs_alt) (void) hipFree(ix->d_vecs_alt);
	if (ix->d_tids_alt) (void) hipFree(ix->d_tids_alt);
	ix->d_vecs_alt = nullptr;
	ix->d_tids_alt = nullptr;
	ix->alt_cap = 0;
	ix->d_vecs = nullptr;
	ix->d_tids = nullptr;
	ix->own_rows = false;
	ix->nrows = 0;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->cap_rows = 0;
}

extern "C" int
ndbhip_ivf_destroy(ndbhip_ivf *ix)
{
	if (!ix)
		return NDBHIP_OK;
	if (g.inited)
	{
		(void) hipStreamSynchronize(g.stream);
		ivf_free_rows(ix);
		void	   *ptrs[] = {ix->d_centroids, ix->d_loc_off, ix->d_glob_len, ix->d_owned, ix->d_own_lo, ix->d_own_len, ix->w_cdist,
			ix->w_probes, ix->w_candoff, ix->w_dist, ix->w_q, ix->w_otid, ix->w_odist, ix->w_ocnt,
			ix->w_gcnt, ix->w_goff, ix->w_pairs, ix->w_qblock, ix->w_qnorm, ix->w_scand, ix->w_sncand, ix->w_stotal, ix->w_tmin, ix->w_cblock,
			ix->d_xxmax, ix->w_rnorm, ix->w_scrt, ix->w_scrd, ix->w_scrc, ix->w_screc,
			ix->d_planes, ix->d_rn2, ix->d_rexp, ix->d_xmax16, ix->w_qplanes, ix->w_qn2, ix->w_qexp, ix->w_qthr,
			ix->w_ecount, ix->w_erec, ix->w_bmin, ix->w_s16desc, ix->d_blkoff};

		for (void *p : ptrs)
			if (p) (void) hipFree(p);
		if (ix->pin) (void) hipHostFree(ix->pin);
	}
	delete ix;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_set_centroids(ndbhip_ivf *ix, const float *centroids, int ncent)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !centroids || ncent < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (ix->d_centroids)
		HIP_TRY(hipFree(ix->d_centroids));
	ix->d_centroids = nullptr;
	HIP_TRY(hipMalloc((void **) &ix->d_centroids, (size_t) ncent * ix->dim * sizeof(float)));
	HIP_TRY(hipMemcpyAsync(ix->d_centroids, centroids, (size_t) ncent * ix->dim * sizeof(float),
						   hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));	/* caller's buffer may be palloc'd: copy before return */
	ix->ncent = ncent;
	ix->loaded = false;
	return NDBHIP_OK;
}

/* own_lo / own_len (optional): the slice of every list this mirror holds; without them a list is held
 * whole (owned[c] != 0, or owned == NULL) or not at all */
static int
ivf_set_layout(ndbhip_ivf *ix, const int64_t *list_len, const uint8_t *owned, int64_t nrows,
			   const int64_t *own_lo = nullptr, const int64_t *own_len = nullptr)
{
	const int	nc = ix->ncent;
	int64_t		acc = 0;

	if (nc < 1)
		return fail(NDBHIP_ERR_STATE, "set centroids before loading lists");
	ix->glob_len.assign(list_len, list_len + nc);
	ix->owned.resize(nc);
	ix->own_lo.resize(nc);
	ix->own_len.resize(nc);
	ix->loc_off.resize(nc + 1);
	std::vector<uint32_t> gl32(nc), lo32(nc), ln32(nc);

	ix->sharded = false;
	for (int c = 0; c < nc; c++)
	{
		if (list_len[c] < 0 || list_len[c] > 0xFFFFFFFFll)
			return fail(NDBHIP_ERR_INVALID, "list_len[%d] out of range", c);
		if (own_len)
		{
			const int64_t lo = own_lo ? own_lo[c] : 0;

			if (lo < 0 || own_len[c] < 0 || lo + own_len[c] > list_len[c])
				return fail(NDBHIP_ERR_INVALID, "slice of list %d is outside the list", c);
			ix->own_lo[c] = own_len[c] > 0 ? lo : 0;
			ix->own_len[c] = own_len[c];
		}
		else
		{
			ix->own_lo[c] = 0;
			ix->own_len[c] = (owned ? (owned[c] != 0) : true) ? list_len[c] : 0;
		}
		/* owned = this mirror takes the appends to list c: it holds the list's tail (or is told so) */
		ix->owned[c] = owned ? owned[c] != 0 : (!own_len || (own_len[c] > 0 && ix->own_lo[c] + own_len[c] == list_len[c]));
		if (ix->own_len[c] != list_len[c])
			ix->sharded = true;
		ix->loc_off[c] = acc;
		acc += ix->own_len[c];
		gl32[c] = (uint32_t) list_len[c];
		lo32[c] = (uint32_t) ix->own_lo[c];
		ln32[c] = (uint32_t) ix->own_len[c];
	}
	ix->loc_off[nc] = acc;
	if (acc != nrows)
		return fail(NDBHIP_ERR_INVALID, "nrows %lld does not match the held slices' total %lld",
					(long long) nrows, (long long) acc);
	if (acc > 0xFFFFFFFFll)
		return fail(NDBHIP_ERR_UNSUPPORTED, "more than 2^32 rows on one device");
	void	  **ptrs[] = {(void **) &ix->d_loc_off, (void **) &ix->d_glob_len, (void **) &ix->d_owned,
		(void **) &ix->d_own_lo, (void **) &ix->d_own_len};

	for (void **p : ptrs)
		if (*p) { HIP_TRY(hipFree(*p)); *p = nullptr; }
	HIP_TRY(hipMalloc((void **) &ix->d_loc_off, (size_t) (nc + 1) * sizeof(int64_t)));
	HIP_TRY(hipMalloc((void **) &ix->d_glob_len, (size_t) nc * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **) &ix->d_own_lo, (size_t) nc * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **) &ix->d_own_len, (size_t) nc * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **) &ix->d_owned, (size_t) nc));
	HIP_TRY(hipMemcpyAsync(ix->d_loc_off, ix->loc_off.data(), (size_t) (nc + 1) * sizeof(int64_t),
						   hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_glob_len, gl32.data(), (size_t) nc * sizeof(uint32_t),
						   hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_own_lo, lo32.data(), (size_t) nc * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_own_len, ln32.data(), (size_t) nc * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(ix->d_owned, ix->owned.data(), (size_t) nc, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	return 0;
}

extern "C" int
ndbhip_ivf_load(ndbhip_ivf *ix, const int64_t *list_len, const uint8_t *owned,
				const float *rows, const uint8_t *tids6, int64_t nrows)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !list_len || nrows < 0 || (nrows > 0 && (!rows || !tids6)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	int			rc = ivf_set_layout(ix, list_len, owned, nrows);

	if (rc)
		return rc;
	ivf_free_rows(ix);
	const int64_t cap = nrows > 0 ? nrows : 1;

	HIP_TRY(hipMalloc((void **) &ix->d_vecs, (size_t) cap * ix->dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &ix->d_tids, (size_t) cap * sizeof(uint64_t)));
	ix->own_rows = true;
	ix->cap_rows = cap;
	ix->f16 = false;
	if (nrows > 0)
	{
		std::vector<uint64_t> t64((size_t) nrows);

		for (int64_t r = 0; r < nrows; r++)
			t64[(size_t) r] = ndb_tid_pack(tids6 + 6 * r);
		HIP_TRY(hipMemcpyAsync(ix->d_vecs, rows, (size_t) nrows * ix->dim * sizeof(float),
							   hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipMemcpyAsync(ix->d_tids, t64.data(), (size_t) nrows * sizeof(uint64_t),
							   hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
	}
	ix->nrows = nrows;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->loaded = true;
	return NDBHIP_OK;
}

/* does any element have a zero exponent and a non-zero mantissa? (sets *flag) */
__global__ __launch_bounds__(256) void
k_f16_has_subnormal(const uint16_t *__restrict__ v, size_t n, int *__restrict__ flag)
{
	size_t		i = (size_t) blockIdx.x * 256 + threadIdx.x;
	bool		sub = false;

	for (; i < n; i += (size_t) gridDim.x * 256)
	{
		const uint32_t h = v[i];

		sub = sub || ((h & 0x7C00u) == 0u && (h & 0x03FFu) != 0u);
	}
	if (__syncthreads_or(sub) && threadIdx.x == 0)
		*flag = 1;
}

/* fp16 mirrors: look once whether the Q20 subnormal fix can ever matter for these rows */
static int
ivf_note_f16_subnormals(ndbhip_ivf *ix)
{
	ix->f16_sub = true;
	if (!ix->f16 || ix->nrows <= 0)
		return 0;
	int		   *d_flag = nullptr;
	int			flag = 0;
	const size_t n = (size_t) ix->nrows * ix->dim;

	HIP_TRY(hipMalloc((void **) &d_flag, sizeof(int)));
	HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), g.stream));
	hipLaunchKernelGGL(k_f16_has_subnormal, dim3((unsigned) std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0,
					   g.stream, (const uint16_t *) ix->d_vecs, n, d_flag);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipFree(d_flag));
	ix->f16_sub = flag != 0;
	return 0;
}

/* halfvec column: rows as IEEE fp16 images (uint16), decoded on the fly exactly like fp16_to_float */
extern "C" int
ndbhip_ivf_load_f16(ndbhip_ivf *ix, const int64_t *list_len, const uint8_t *owned,
					const uint16_t *rows_f16, const uint8_t *tids6, int64_t nrows)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !list_len || nrows < 0 || (nrows > 0 && (!rows_f16 || !tids6)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (ix->dim % 64 != 0)
		return fail(NDBHIP_ERR_UNSUPPORTED, "fp16 rows need dim %% 64 == 0 (dim = %d)", ix->dim);
	int			rc = ivf_set_layout(ix, list_len, owned, nrows);

	if (rc)
		return rc;
	ivf_free_rows(ix);
	const int64_t cap = nrows > 0 ? nrows : 1;

	HIP_TRY(hipMalloc((void **) &ix->d_vecs, (size_t) cap * ix->dim * sizeof(uint16_t)));
	HIP_TRY(hipMalloc((void **) &ix->d_tids, (size_t) cap * sizeof(uint64_t)));
	ix->own_rows = true;
	ix->cap_rows = cap;
	if (nrows > 0)
	{
		std::vector<uint64_t> t64((size_t) nrows);

		for (int64_t r = 0; r < nrows; r++)
			t64[(size_t) r] = ndb_tid_pack(tids6 + 6 * r);
		HIP_TRY(hipMemcpyAsync(ix->d_vecs, rows_f16, (size_t) nrows * ix->dim * sizeof(uint16_t),
							   hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipMemcpyAsync(ix->d_tids, t64.data(), (size_t) nrows * sizeof(uint64_t), hipMemcpyHostToDevice,
							   g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
	}
	ix->nrows = nrows;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->f16 = true;
	ix->loaded = true;
	return ivf_note_f16_subnormals(ix);
}

extern "C" int
ndbhip_ivf_load_device(ndbhip_ivf *ix, const int64_t *list_len, const uint8_t *owned,
					   const float *d_rows, const uint64_t *d_tids, int64_t nrows)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !list_len || nrows < 0 || (nrows > 0 && (!d_rows || !d_tids)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (((uintptr_t) d_rows & 15) != 0)
		return fail(NDBHIP_ERR_INVALID, "d_rows must be 16-byte aligned");
	int			rc = ivf_set_layout(ix, list_len, owned, nrows);

	if (rc)
		return rc;
	ivf_free_rows(ix);
	ix->d_vecs = const_cast<float *>(d_rows);
	ix->d_tids = const_cast<uint64_t *>(d_tids);
	ix->own_rows = false;
	ix->f16 = false;
	ix->nrows = nrows;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->cap_rows = nrows;
	ix->loaded = true;
	return NDBHIP_OK;
}

extern "C" int64_t
ndbhip_ivf_nrows(const ndbhip_ivf *ix)
{
	if (!ix)
		return -1;
	int64_t		n = ix->nrows;

	for (int c : ix->pend_list)
		if (ix->owned[c])
			n++;
	return n;
}

extern "C" int64_t
ndbhip_ivf_max_candidates(const ndbhip_ivf *ix, int nprobe)
{
	if (!ix || nprobe < 1)
		return 0;
	std::vector<int64_t> v(ix->glob_len);
	int			n = std::min<int>(nprobe, (int) v.size());

	std::partial_sort(v.begin(), v.begin() + n, v.end(), std::greater<int64_t>());
	int64_t		s = 0;

	for (int i = 0; i < n; i++)
		s += v[i];
	/* nprobe > nlists: the never-written probe slots re-scan list 0 (ivf_am.c:1978, 1990-1999) */
	if (nprobe > n && !ix->glob_len.empty())
		s += (int64_t) (nprobe - n) * ix->glob_len[0];
	return s;
}

/* the same bound over the rows THIS mirror holds: sizes the candidate-distance buffer */
static int64_t
ivf_local_max_candidates(const ndbhip_ivf *ix, int nprobe)
{
	if (!ix->sharded)
		return ndbhip_ivf_max_candidates(ix, nprobe);
	std::vector<int64_t> v(ix->glob_len.size());

	for (size_t i = 0; i < v.size(); i++)
		v[i] = ix->own_len[i];
	int			n = std::min<int>(nprobe, (int) v.size());

	std::partial_sort(v.begin(), v.begin() + n, v.end(), std::greater<int64_t>());
	int64_t		s = 0;

	for (int i = 0; i < n; i++)
		s += v[i];
	if (nprobe > n && !v.empty())
		s += (int64_t) (nprobe - n) * ix->own_len[0];
	return s;
}

/* one block per 64 new rows: row r of the new layout comes from the old mirror or from the staged appends */
__global__ __launch_bounds__(256) void
k_flush_gather(const float *__restrict__ old_rows, const uint64_t *__restrict__ old_tids,
			   const float *__restrict__ stage_rows, const uint64_t *__restrict__ stage_tids,
			   const int64_t *__restrict__ new_off, const int64_t *__restrict__ old_off, const int64_t *__restrict__ old_own,
			   const int64_t *__restrict__ stage_off, int ncent, int dim, int64_t nnew, float *__restrict__ out_rows,
			   uint64_t *__restrict__ out_tids)
{
	const int64_t r0 = (int64_t) blockIdx.x * 64;

	for (int rr = threadIdx.x >> 6; rr < 64; rr += 4)
	{
		const int64_t r = r0 + rr;

		if (r >= nnew)
			break;
		int			lo = 0, hi = ncent;	/* largest c with new_off[c] <= r (empty lists share an offset: skip forward) */

		while (hi - lo > 1)
		{
			const int	mid = (lo + hi) >> 1;

			if (new_off[mid] <= r)
				lo = mid;
			else
				hi = mid;
		}
		while (lo + 1 < ncent && new_off[lo + 1] <= r)
			lo++;
		const int64_t idx = r - new_off[lo];
		const bool	from_old = idx < old_own[lo];
		const float *src = from_old ? old_rows + (size_t) (old_off[lo] + idx) * dim
			: stage_rows + (size_t) (stage_off[lo] + idx - old_own[lo]) * dim;
		float	   *dst = out_rows + (size_t) r * dim;

		for (int d = threadIdx.x & 63; d < dim; d += 64)
			dst[d] = src[d];
		if ((threadIdx.x & 63) == 0)
			out_tids[r] = from_old ? old_tids[old_off[lo] + idx] : stage_tids[stage_off[lo] + idx - old_own[lo]];
	}
}

/*
 * The first search after ndbhip_ivf_append() folds the pending entries into the mirror: every list keeps its
 * order, the appended entries follow in insertion order (the page chain's order, ivf_am.c:985-1120).  One gather
 * kernel writes the new layout into the mirror's second buffer (kept between flushes, grown with 1/8 slack: no
 * multi-GB hipMalloc and no per-list copies on the scan that follows an INSERT), the buffers swap.  O(N) bytes
 * still move; the norms / fp16 planes of the batched scans are rebuilt lazily by the next batch that needs them.
 */
static int
ivf_flush(ndbhip_ivf *ix)
{
	const size_t P = ix->pend_list.size();

	if (P == 0)
		return 0;
	const int	nc = ix->ncent;
	const int	dim = ix->dim;
	std::vector<int64_t> add((size_t) nc, 0), new_len(ix->glob_len);
	std::vector<int64_t> new_off((size_t) nc + 1, 0);
	int64_t		nown = 0;

	for (size_t i = 0; i < P; i++)
	{
		add[ix->pend_list[i]]++;
		new_len[ix->pend_list[i]]++;
	}
	std::vector<int64_t> new_own(ix->own_len);

	for (int c = 0; c < nc; c++)
	{
		new_off[c] = nown;
		if (ix->owned[c])
			new_own[c] += add[c];
		nown += new_own[c];
	}
	new_off[nc] = nown;
	if (nown > 0xFFFFFFFFll)
		return fail(NDBHIP_ERR_UNSUPPORTED, "more than 2^32 rows on one device");

	/* pending rows of owned lists, grouped by list (stable) */
	std::vector<int64_t> stage_off((size_t) nc + 1, 0);

	for (int c = 0; c < nc; c++)
		stage_off[c + 1] = stage_off[c] + (ix->owned[c] ? add[c] : 0);
	const int64_t nstage = stage_off[nc];
	std::vector<float> srows((size_t) std::max<int64_t>(nstage, 1) * dim);
	std::vector<uint64_t> stids((size_t) std::max<int64_t>(nstage, 1));
	std::vector<int64_t> cur(stage_off.begin(), stage_off.end() - 1);

	for (size_t i = 0; i < P; i++)
	{
		const int	c = ix->pend_list[i];

		if (!ix->owned[c])
			continue;
		memcpy(&srows[(size_t) cur[c] * dim], &ix->pend_rows[i * dim], (size_t) dim * sizeof(float));
		stids[(size_t) cur[c]] = ix->pend_tids[i];
		cur[c]++;
	}
	if (nown > 0)
	{
		DevGuard	tmp;
		float	   *stage_d = nullptr;
		uint64_t   *stids_d = nullptr;
		int64_t    *meta_d = nullptr;	/* new_off | old_off | old_own | stage_off, nc + 1 each */
		std::vector<int64_t> meta((size_t) 4 * (nc + 1), 0);

		for (int c = 0; c <= nc; c++)
		{
			meta[c] = new_off[c];
			meta[(size_t) (nc + 1) + c] = ix->loc_off[c];
			meta[(size_t) 2 * (nc + 1) + c] = c < nc ? ix->own_len[c] : 0;
			meta[(size_t) 3 * (nc + 1) + c] = stage_off[c];
		}
		if (tmp.alloc(stage_d, (size_t) std::max<int64_t>(nstage, 1) * dim * sizeof(float))) return NDBHIP_ERR_HIP;
		if (tmp.alloc(stids_d, (size_t) std::max<int64_t>(nstage, 1) * sizeof(uint64_t))) return NDBHIP_ERR_HIP;
		if (tmp.alloc(meta_d, meta.size() * sizeof(int64_t))) return NDBHIP_ERR_HIP;
		/* the second buffer: kept between flushes */
		if (!ix->own_rows || (int64_t) ix->alt_cap < nown)
		{
			const int64_t cap = nown + nown / 8 + 1024;

			if (ix->d_vecs_alt) HIP_TRY(hipFree(ix->d_vecs_alt));
			if (ix->d_tids_alt) HIP_TRY(hipFree(ix->d_tids_alt));
			ix->d_vecs_alt = nullptr;
			ix->d_tids_alt = nullptr;
			ix->alt_cap = 0;
			HIP_TRY(hipMalloc((void **) &ix->d_vecs_alt, (size_t) cap * dim * sizeof(float)));
			HIP_TRY(hipMalloc((void **) &ix->d_tids_alt, (size_t) cap * sizeof(uint64_t)));
			ix->alt_cap = (size_t) cap;
		}
		HIP_TRY(hipMemcpyAsync(stage_d, srows.data(), (size_t) std::max<int64_t>(nstage, 1) * dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipMemcpyAsync(stids_d, stids.data(), (size_t) std::max<int64_t>(nstage, 1) * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipMemcpyAsync(meta_d, meta.data(), meta.size() * sizeof(int64_t), hipMemcpyHostToDevice, g.stream));
		hipLaunchKernelGGL(k_flush_gather, dim3((unsigned) ((nown + 63) / 64)), dim3(256), 0, g.stream,
						   (const float *) ix->d_vecs, (const uint64_t *) ix->d_tids, (const float *) stage_d,
						   (const uint64_t *) stids_d, (const int64_t *) meta_d, (const int64_t *) meta_d + (nc + 1),
						   (const int64_t *) meta_d + 2 * (nc + 1), (const int64_t *) meta_d + 3 * (nc + 1), nc, dim, nown,
						   ix->d_vecs_alt, ix->d_tids_alt);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(g.stream));
	}
	std::vector<uint8_t> owned(ix->owned);
	std::vector<int64_t> lo(ix->own_lo);
	int			rc = ivf_set_layout(ix, new_len.data(), owned.data(), nown, lo.data(), new_own.data());

	if (rc)
		return rc;
	if (nown > 0)
	{
		/* swap: the old mirror becomes the second buffer (if the library owns it) */
		float	   *ov = ix->own_rows ? ix->d_vecs : nullptr;
		uint64_t   *ot = ix->own_rows ? ix->d_tids : nullptr;
		const size_t ocap = ix->own_rows ? (size_t) ix->cap_rows : 0;

		ix->d_vecs = ix->d_vecs_alt;
		ix->d_tids = ix->d_tids_alt;
		ix->cap_rows = (int64_t) ix->alt_cap;
		ix->d_vecs_alt = ov;
		ix->d_tids_alt = ot;
		ix->alt_cap = ocap;
		ix->own_rows = true;
	}
	ix->nrows = nown;
	ix->norm_valid = false; ix->s16_valid = false;
	ix->pend_list.clear();
	ix->pend_rows.clear();
	ix->pend_tids.clear();
	return 0;
}

/* ------------------------------------------------------------------ */
/* ambulkdelete (src/index/ivf_am.c:1172-1357): the callback is a set    */
/* test on heapPtr; entries it hits get a dead line pointer and every    */
/* later scan skips them (:1816-1822).  On the mirror: drop those rows,  */
/* survivors keep their list and their order inside it.                  */
/* ------------------------------------------------------------------ */

/* keep[r] = heapPtr of row r is NOT in the sorted dead set; block_sum[b] = keeps in rows [256b, 256b+256) */
__global__ __launch_bounds__(256) void
k_delete_mark(const uint64_t *__restrict__ tids, int64_t nrows, const uint64_t *__restrict__ dead, int64_t ndead,
			  uint8_t *__restrict__ keep, uint32_t *__restrict__ block_sum)
{
	__shared__ uint32_t wsum[4];
	const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
	bool		k = false;

	if (r < nrows)
	{
		const uint64_t t = tids[r];
		int64_t		lo = 0, hi = ndead;

		while (lo < hi)
		{
			const int64_t mid = (lo + hi) >> 1;

			if (dead[mid] < t)
				lo = mid + 1;
			else
				hi = mid;
		}
		k = !(lo < ndead && dead[lo] == t);
		keep[r] = k ? 1 : 0;
	}
	const unsigned long long b = __ballot(k);

	if ((threadIdx.x & 63) == 0)
		wsum[threadIdx.x >> 6] = (uint32_t) __popcll(b);
	__syncthreads();
	if (threadIdx.x == 0)
		block_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

/* exclusive scan of block_sum in place (one block); total -> *total */
__global__ __launch_bounds__(1024) void
k_delete_scan(uint32_t *__restrict__ block_sum, uint32_t nblocks, uint32_t *__restrict__ total)
{
	__shared__ uint32_t sh[1024];
	__shared__ uint32_t carry;

	if (threadIdx.x == 0)
		carry = 0;
	__syncthreads();
	for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024)
	{
		const uint32_t i = b0 + threadIdx.x;
		const uint32_t v = i < nblocks ? block_sum[i] : 0u;

		sh[threadIdx.x] = v;
		__syncthreads();
		for (uint32_t off = 1; off < 1024; off <<= 1)
		{
			const uint32_t add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;

			__syncthreads();
			sh[threadIdx.x] += add;
			__syncthreads();
		}
		if (i < nblocks)
			block_sum[i] = carry + sh[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == 1023)
			carry += sh[1023];
		__syncthreads();
	}
	if (threadIdx.x == 0)
		*total = carry;
}

/* new position of every kept row; pref[r] = survivors before row r (pref[nrows] = total) */
__global__ __launch_bounds__(256) void
k_delete_positions(const uint8_t *__restrict__ keep, int64_t nrows, const uint32_t *__restrict__ block_base,
				   uint32_t *__restrict__ pref)
{
	__shared__ uint32_t wsum[4];
	const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
	const bool	k = r < nrows && keep[r];
	const unsigned long long b = __ballot(k);
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;

	if (lane == 0)
		wsum[w] = (uint32_t) __popcll(b);
	__syncthreads();
	uint32_t	base = block_base[blockIdx.x];

	for (uint32_t i = 0; i < w; i++)
		base += wsum[i];
	if (r < nrows)
		pref[r] = base + (uint32_t) __popcll(b & ((1ull << lane) - 1ull));
	if (r == nrows - 1)
		pref[nrows] = base + (uint32_t) __popcll(b & ((1ull << lane) - 1ull)) + (k ? 1u : 0u);
}

/* one block per row: survivors move to pref[r] (rows are `row_bytes` bytes, a multiple of 4) */
__global__ __launch_bounds__(256) void
k_delete_move(const uint8_t *__restrict__ keep, const uint32_t *__restrict__ pref, const uint32_t *__restrict__ src,
			  const uint64_t *__restrict__ src_tids, uint32_t *__restrict__ dst, uint64_t *__restrict__ dst_tids,
			  uint32_t row_words)
{
	const size_t r = blockIdx.x;

	if (!keep[r])
		return;
	const size_t d = pref[r];

	for (uint32_t j = threadIdx.x; j < row_words; j += 256)
		dst[d * row_words + j] = src[r * row_words + j];
	if (threadIdx.x == 0)
		dst_tids[d] = src_tids[r];
}

__global__ void
k_delete_list_len(const uint32_t *__restrict__ pref, const int64_t *__restrict__ loc_off, int ncent,
				  int64_t *__restrict__ new_len)
{
	const int	L = blockIdx.x * blockDim.x + threadIdx.x;

	if (L < ncent)
		new_len[L] = (int64_t) pref[loc_off[L + 1]] - (int64_t) pref[loc_off[L]];
}

extern "C" int
ndbhip_ivf_delete(ndbhip_ivf *ix, const uint8_t *tids6, int64_t n, int64_t *removed)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || n < 0 || (n > 0 && !tids6))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!ix->loaded)
		return fail(NDBHIP_ERR_STATE, "index has no lists loaded");
	if (ix->sharded)
		return fail(NDBHIP_ERR_UNSUPPORTED, "delete on the unsharded mirror and shard again: the other ranks' "
					"list lengths (global candidate positions) must change with it");
	if (!ix->own_rows)
		return fail(NDBHIP_ERR_STATE, "the rows belong to the caller (ndbhip_ivf_load_device): rebuild the layout there");
	int			rc = ivf_flush(ix);

	if (rc)
		return rc;
	if (removed)
		*removed = 0;
	if (n == 0 || ix->nrows == 0)
		return NDBHIP_OK;
	std::vector<uint64_t> dead((size_t) n);

	for (int64_t i = 0; i < n; i++)
		dead[(size_t) i] = ndb_tid_pack(tids6 + 6 * i);
	std::sort(dead.begin(), dead.end());
	const int64_t nrows = ix->nrows;
	const uint32_t nblk = (uint32_t) ((nrows + 255) / 256);
	const size_t esz = ix->f16 ? sizeof(uint16_t) : sizeof(float);
	const uint32_t row_words = (uint32_t) ((size_t) ix->dim * esz / 4);
	uint64_t   *d_dead = nullptr;
	uint8_t    *d_keep = nullptr;
	uint32_t   *d_bs = nullptr, *d_pref = nullptr, *d_total = nullptr;
	int64_t    *d_newlen = nullptr;
	uint32_t	total = 0;

	if (((size_t) ix->dim * esz) % 4 != 0)
		return fail(NDBHIP_ERR_UNSUPPORTED, "row size must be a multiple of 4 bytes");
	DevGuard	tmp;				/* freed on every way out */

	if (tmp.alloc(d_dead, (size_t) n * sizeof(uint64_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_keep, (size_t) nrows)) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_bs, (size_t) nblk * sizeof(uint32_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_pref, ((size_t) nrows + 1) * sizeof(uint32_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_total, sizeof(uint32_t))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_newlen, (size_t) ix->ncent * sizeof(int64_t))) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemcpyAsync(d_dead, dead.data(), (size_t) n * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
	hipLaunchKernelGGL(k_delete_mark, dim3(nblk), dim3(256), 0, g.stream, (const uint64_t *) ix->d_tids, nrows,
					   (const uint64_t *) d_dead, n, d_keep, d_bs);
	hipLaunchKernelGGL(k_delete_scan, dim3(1), dim3(1024), 0, g.stream, d_bs, nblk, d_total);
	hipLaunchKernelGGL(k_delete_positions, dim3(nblk), dim3(256), 0, g.stream, (const uint8_t *) d_keep, nrows,
					   (const uint32_t *) d_bs, d_pref);
	hipLaunchKernelGGL(k_delete_list_len, dim3((ix->ncent + 255) / 256), dim3(256), 0, g.stream,
					   (const uint32_t *) d_pref, (const int64_t *) ix->d_loc_off, ix->ncent, d_newlen);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, g.stream));
	std::vector<int64_t> newlen((size_t) ix->ncent);

	HIP_TRY(hipMemcpyAsync(newlen.data(), d_newlen, (size_t) ix->ncent * sizeof(int64_t), hipMemcpyDeviceToHost,
						   g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	if ((int64_t) total < nrows)
	{
		const int64_t cap = total > 0 ? (int64_t) total : 1;
		unsigned char *nv = nullptr;
		uint64_t   *nt = nullptr;

		if (tmp.alloc(nv, (size_t) cap * ix->dim * esz)) return NDBHIP_ERR_HIP;
		if (tmp.alloc(nt, (size_t) cap * sizeof(uint64_t))) return NDBHIP_ERR_HIP;
		hipLaunchKernelGGL(k_delete_move, dim3((unsigned) nrows), dim3(256), 0, g.stream, (const uint8_t *) d_keep,
						   (const uint32_t *) d_pref, (const uint32_t *) ix->d_vecs, (const uint64_t *) ix->d_tids,
						   (uint32_t *) nv, nt, row_words);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(g.stream));
		ivf_free_rows(ix);
		ix->d_vecs = (float *) nv;
		ix->d_tids = nt;
		tmp.keep(nv);				/* the mirror owns them now */
		tmp.keep(nt);
		ix->own_rows = true;
		ix->cap_rows = cap;
		ix->nrows = (int64_t) total;
		ix->norm_valid = false; ix->s16_valid = false;
		rc = ivf_set_layout(ix, newlen.data(), nullptr, (int64_t) total);
	}
	if (removed)
		*removed = nrows - (int64_t) total;
	return rc;
}

extern "C" int
ndbhip_ivf_append(ndbhip_ivf *ix, int list_id, const float *vec, const uint8_t *tid6)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !vec || !tid6)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!ix->loaded)
		return fail(NDBHIP_ERR_STATE, "index has no lists loaded");
	if (list_id < 0 || list_id >= ix->ncent)
		return fail(NDBHIP_ERR_INVALID, "list %d out of range 0..%d", list_id, ix->ncent - 1);
	if (ix->f16)
		return fail(NDBHIP_ERR_UNSUPPORTED, "append to an fp16 mirror is not implemented: reload the list");
	ix->pend_list.push_back(list_id);
	ix->pend_rows.insert(ix->pend_rows.end(), vec, vec + ix->dim);	/* copied: caller's memory may be palloc'd */
	ix->pend_tids.push_back(ndb_tid_pack(tid6));
	return NDBHIP_OK;
}

static IvfDev
ivf_dev(const ndbhip_ivf *ix)
{
	IvfDev		d;

	d.vecs = ix->d_vecs;
	d.tids = ix->d_tids;
	d.centroids = ix->d_centroids;
	d.loc_off = ix->d_loc_off;
	d.glob_len = ix->d_glob_len;
	d.owned = ix->d_owned;
	d.own_lo = ix->d_own_lo;
	d.own_len = ix->d_own_len;
	d.dim = ix->dim;
	d.ncent = ix->ncent;
	d.nlists = ix->nlists;
	d.f16 = ix->f16 ? 1 : 0;
	return d;
}

static int
ivf_recipe(int strategy)
{
	switch (strategy)
	{
		case NDBHIP_STRATEGY_COSINE: return R_IVF_COS;
		case NDBHIP_STRATEGY_IP: return R_IVF_IP;
		default: return R_IVF_L2;	/* ivf_am.c:1561, 1583 */
	}
}

#define LAUNCH_BY_RECIPE(R, KERNEL, GRID, BLOCK, ...)                                            \
	do {                                                                                         \
		switch (R) {                                                                             \
			case R_IVF_L2: hipLaunchKernelGGL(KERNEL<R_IVF_L2>, GRID, BLOCK, 0, g.stream, __VA_ARGS__); break; \
			case R_IVF_COS: hipLaunchKernelGGL(KERNEL<R_IVF_COS>, GRID, BLOCK, 0, g.stream, __VA_ARGS__); break; \
			case R_IVF_IP: hipLaunchKernelGGL(KERNEL<R_IVF_IP>, GRID, BLOCK, 0, g.stream, __VA_ARGS__); break; \
			case R_IVF_L2SQ: hipLaunchKernelGGL(KERNEL<R_IVF_L2SQ>, GRID, BLOCK, 0, g.stream, __VA_ARGS__); break; \
			case R_HNSW_L2: hipLaunchKernelGGL(KERNEL<R_HNSW_L2>, GRID, BLOCK, 0, g.stream, __VA_ARGS__); break; \
			case R_HNSW_COS: hipLaunchKernelGGL(KERNEL<R_HNSW_COS>, GRID, BLOCK, 0, g.stream, __VA_ARGS__); break; \
			default: hipLaunchKernelGGL(KERNEL<R_HNSW_IP>, GRID, BLOCK, 0, g.stream, __VA_ARGS__); break; \
		}                                                                                        \
	} while (0)

/* defined with the build kernels below; the batch centroid scan of the search reuses them */
#include "ndbhip_screen16.h"

/* the fp16-MFMA screened scan in auto mode (ndbhip_set_option("screen16", 0) turns it off: the older fp32 bound
 * pass then serves batches of >= 128 queries); records a query may emit before the batch falls back */
static int	g_s16_auto = 1;
static int	g_s16_waves = 4;
static int	g_s16_debug = 0;		/* timing experiments (wrong results): see k_s16_sweep's DBG */
static uint32_t g_s16_ecap = 2048;

static bool
ivf_s16_eligible(const ndbhip_ivf *ix, int nq, int R, int k)
{
	const size_t dimp = (size_t) ((ix->dim + 63) & ~63);

	if (R != R_IVF_L2 && R != R_IVF_IP)
		return false;
	if (k > NDB_TOPK_FAST_MAXK || ix->nrows < 1)
		return false;
	if (ix->f16 && (ix->dim % 64) != 0)
		return false;
	if ((size_t) nq * dimp * 4 >= ((size_t) 1 << 32) || (size_t) 256 * dimp * 4 >= ((size_t) 1 << 31))
		return false;
	return true;
}

/* Runs the sweep + finalize for one sub-batch whose probes / candidate offsets are already on the device.
 * Returns 0, a negative error, or 1 when some query overflowed (nothing usable was written: rerun on the older path). */
static int
ivf_s16_run(ndbhip_ivf *ix, const IvfDev &d, const float *d_q, int nq, int R, int npr, int k, const int *w_probes,
			const uint32_t *lco, int partial, ndbhip_cand *d_cand, int *d_ncand, int64_t *d_total,
			uint64_t *d_otid, float *d_odist, int *d_ocnt)
{
	const int	dim = ix->dim, dimp = (dim + 63) & ~63;	/* two chunks per accumulator block */
	const uint32_t qrowbytes = (uint32_t) dimp * 4u;
	const int	nc = ix->ncent;
	const int	H = !ix->f16 ? 0 : (ix->f16_sub ? 1 : 2);
	const uint32_t ecap = g_s16_ecap;

	if (!ix->s16_valid)
	{
		/* every list starts a new 32-row block of the blocked planes */
		std::vector<uint32_t> bo((size_t) nc + 1);
		uint64_t	nb = 0;

		for (int c = 0; c < nc; c++)
		{
			bo[c] = (uint32_t) nb;
			nb += (uint64_t) ((ix->own_len[c] + 31) / 32);
		}
		bo[nc] = (uint32_t) nb;
		if (nb + 8 > 0xFFFFFFFFull)
			return fail(NDBHIP_ERR_UNSUPPORTED, "more than 2^32 row blocks");
		const size_t blk_bytes = (size_t) (dimp / S16_CH) * (ix->f16 ? 2048 : 4096);

		if (grow(ix->d_blkoff, ix->d_blkoff_n, (size_t) nc + 1)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemcpyAsync(ix->d_blkoff, bo.data(), ((size_t) nc + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));		/* bo is a local */
		if (grow(ix->d_rn2, ix->d_rn2_n, (size_t) ix->nrows)) return NDBHIP_ERR_HIP;
		if (grow(ix->d_rexp, ix->d_rexp_n, (size_t) ix->nrows)) return NDBHIP_ERR_HIP;
		if (grow(ix->d_planes, ix->d_planes_n, (size_t) (nb + 8) * blk_bytes)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemsetAsync(ix->d_planes, 0, (size_t) (nb + 8) * blk_bytes, g.stream));
		if (!ix->d_xmax16)
			HIP_TRY(hipMalloc((void **) &ix->d_xmax16, sizeof(uint32_t)));
		HIP_TRY(hipMemsetAsync(ix->d_xmax16, 0, sizeof(uint32_t), g.stream));
		const dim3	gp((unsigned) ((ix->nrows + 3) / 4));

#define S16_PREP_L(HH)                                                                                          \
		hipLaunchKernelGGL(k_s16_row_prep<HH>, gp, dim3(256), 0, g.stream, (const void *) ix->d_vecs, ix->nrows, dim, dimp, \
						   (const int64_t *) ix->d_loc_off, (const uint32_t *) ix->d_blkoff, nc, ix->d_planes, ix->d_rn2, \
						   ix->d_rexp, ix->d_xmax16)
		if (!ix->f16)
			S16_PREP_L(0);
		else if (ix->f16_sub)
			S16_PREP_L(1);
		else
			S16_PREP_L(2);
		HIP_TRY(hipGetLastError());
		ix->s16_valid = true;
	}
	if (grow(ix->w_qplanes, ix->w_qplanes_n, (size_t) nq * qrowbytes)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_qn2, ix->w_qn2_n, (size_t) nq)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_qexp, ix->w_qexp_n, (size_t) nq)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_qthr, ix->w_qthr_n, (size_t) nq)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_ecount, ix->w_ecount_n, (size_t) 3 * nq + 4)) return NDBHIP_ERR_HIP;	/* emitted | survivors | active | flags */
	if (grow(ix->w_erec, ix->w_erec_n, (size_t) nq * ecap)) return NDBHIP_ERR_HIP;
	unsigned int *ecount = ix->w_ecount, *surv = ix->w_ecount + nq, *flags = ix->w_ecount + 3 * (size_t) nq;

	if (grow(ix->w_bmin, ix->w_bmin_n, (size_t) nq * S16_NB)) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemsetAsync(ix->w_ecount, 0, ((size_t) 3 * nq + 4) * sizeof(unsigned int), g.stream));
	HIP_TRY(hipMemsetAsync(ix->w_bmin, 0xFF, (size_t) nq * S16_NB * sizeof(uint32_t), g.stream));
	hipLaunchKernelGGL(k_s16_qprep, dim3((nq + 3) / 4), dim3(256), 0, g.stream, d_q, (uint32_t) nq, dim, dimp,
					   (ndb_h2 *) ix->w_qplanes, ix->w_qn2, ix->w_qexp);
#define S16_BY_RH(KERNEL, ...)                                                                  \
	do {                                                                                        \
		if (R == R_IVF_IP)                                                                      \
		{                                                                                       \
			if (H == 0) KERNEL(R_IVF_IP, 0, __VA_ARGS__);                                        \
			else if (H == 1) KERNEL(R_IVF_IP, 1, __VA_ARGS__);                                   \
			else KERNEL(R_IVF_IP, 2, __VA_ARGS__);                                               \
		}                                                                                       \
		else                                                                                    \
		{                                                                                       \
			if (H == 0) KERNEL(R_IVF_L2, 0, __VA_ARGS__);                                        \
			else if (H == 1) KERNEL(R_IVF_L2, 1, __VA_ARGS__);                                   \
			else KERNEL(R_IVF_L2, 2, __VA_ARGS__);                                               \
		}                                                                                       \
	} while (0)
#define S16_SEED_L(RR, HH, ...) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_seed<RR, HH>), dim3(nq), dim3(64), 0, g.stream, __VA_ARGS__)
	S16_BY_RH(S16_SEED_L, d, d_q, w_probes, lco, npr, (uint32_t) k, (const float *) ix->w_qn2,
			  (const uint32_t *) ix->d_xmax16, (int) (ix->f16 && ix->f16_sub), ix->w_qthr);

	/* the (query, probe) pairs bucketed by list; items of 128 rows x 128 queries */
	uint32_t   *cnt = ix->w_gcnt, *fill = ix->w_gcnt + nc;
	unsigned int *next_item = ix->w_gcnt + 2 * nc;
	uint32_t   *pair_off = ix->w_goff, *item_off = ix->w_goff + (nc + 1), *grp_off = ix->w_goff + 2 * (nc + 1);
	uint32_t   *runs = ix->w_goff + 3 * (nc + 1) + 32;
	const uint32_t npairs = (uint32_t) nq * (uint32_t) npr;
	ScanTimer	t;

	/* tile geometry: 8 waves, 256 rows x 128 queries, ring of 3 chunk buffers, one block per CU (default), or
	 * 4 waves, 128 x 128, ring of 2, two blocks per CU (ndbhip_set_option("screen16_waves", 4)) */
	const int	s16_rt = g_s16_waves == 8 ? 256 : 128;
#define S16_SWEEP_L(RR, HH, ...)                                                                                  \
	do {                                                                                                          \
		if (g_s16_debug == 1 && HH == 0 && RR == R_IVF_L2)                                                         \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_L2, 0, 4, 2, 1>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, __VA_ARGS__); \
		else if (g_s16_debug == 2 && HH == 0 && RR == R_IVF_L2)                                                    \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_L2, 0, 4, 2, 2>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, __VA_ARGS__); \
		else if (g_s16_waves == 8)                                                                                \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<RR, HH, 8, 3>), dim3(g.num_cus), dim3(512), 0, g.stream, __VA_ARGS__); \
		else                                                                                                      \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<RR, HH, 4, 2>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, __VA_ARGS__); \
	} while (0)
	/*
	 * Round 0 sweeps every (query, probe) pair against the seed threshold.  A query that emits more than its
	 * record capacity (its nearest list is huge, or the data has no cluster structure) keeps the first `ecap`
	 * records — any subset is valid evidence — and k_s16_retarget turns their k-th smallest into a far tighter
	 * threshold (a sample of 2048 values below the seed threshold puts its 10th smallest ~200x deeper); round 1
	 * sweeps those queries alone against it.  Whatever still overflows after that sends the batch to the older path.
	 */
	unsigned int *active = ix->w_ecount + 2 * (size_t) nq;

	for (int round = 0; round < 2; round++)
	{
		const unsigned int *act = round ? active : (const unsigned int *) nullptr;
		uint32_t	desc_cap = 0;

		if (round)
		{
			if (R == R_IVF_IP)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_retarget<R_IVF_IP>), dim3(nq), dim3(S16_NB), 0, g.stream, dim, (uint32_t) k,
								   ix->w_qthr, ecount, ecap, (const uint32_t *) ix->w_bmin, active, flags + 1);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_retarget<R_IVF_L2>), dim3(nq), dim3(S16_NB), 0, g.stream, dim, (uint32_t) k,
								   ix->w_qthr, ecount, ecap, (const uint32_t *) ix->w_bmin, active, flags + 1);
		}
		HIP_TRY(hipMemsetAsync(ix->w_gcnt, 0, (size_t) (2 * nc + 8 * NDB_QHEAD_STRIDE) * sizeof(uint32_t), g.stream));
		hipLaunchKernelGGL(k_pair_count, dim3((npairs + 255) / 256), dim3(256), 0, g.stream, w_probes, lco, npr,
						   (uint32_t) nq, cnt, act);
		hipLaunchKernelGGL(k_pair_offsets, dim3(1), dim3(1024), 0, g.stream, (const uint32_t *) cnt, d.own_len, nc,
						   pair_off, item_off, grp_off, runs, (uint32_t) (S16_QT / NDB_QG), (uint32_t) (s16_rt / 64));
		hipLaunchKernelGGL(k_pair_fill, dim3((npairs + 255) / 256), dim3(256), 0, g.stream, w_probes, lco, npr,
						   (uint32_t) nq, (const uint32_t *) pair_off, fill, ix->w_pairs, act);
		{
			/* items <= (row tiles) x (query tiles of the fullest list); a query probes a list once — except list 0,
			 * which the reference scans again for every probe slot beyond nlists (ivf_am.c:1978, palloc0) */
			const int	ncmp = std::min(ix->nlists, ix->ncent);
			const size_t dup = npr > ncmp ? (size_t) (npr - ncmp + 1) : 1;
			const size_t cap_items = ((size_t) ix->nrows / (size_t) s16_rt + (size_t) nc) *
				(((size_t) nq * dup + S16_QT - 1) / S16_QT);

			if (grow(ix->w_s16desc, ix->w_s16desc_n, cap_items * 4)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_s16_items, dim3((unsigned) ((cap_items + 255) / 256)), dim3(256), 0, g.stream,
							   (const uint32_t *) item_off, (const uint32_t *) cnt, d.own_len, nc, (uint32_t) s16_rt,
							   (uint32_t) std::min<size_t>(cap_items, 0xFFFFFFFFu), (S16Desc *) ix->w_s16desc, flags);
			desc_cap = (uint32_t) std::min<size_t>(cap_items, 0xFFFFFFFFu);
		}
		if (round == 0 && t.start()) return NDBHIP_ERR_HIP;
		S16_BY_RH(S16_SWEEP_L, d, (const unsigned char *) ix->d_planes, (const uint32_t *) ix->d_blkoff,
				  (const float *) ix->d_rn2, (const int16_t *) ix->d_rexp, (const unsigned char *) ix->w_qplanes, qrowbytes,
				  (const float *) ix->w_qn2, (const int *) ix->w_qexp, (const float2 *) ix->w_qthr, lco, npr,
				  (const uint32_t *) cnt, (const uint32_t *) pair_off, (const S16Desc *) ix->w_s16desc,
				  (const PairRec *) ix->w_pairs, next_item, (const uint32_t *) runs, ecount, ix->w_erec, ecap,
				  ix->w_bmin, nq < 1024 ? 1 : 0, dimp / S16_CH, desc_cap);
		if (round == 0 && t.stop()) return NDBHIP_ERR_HIP;
	}
	const size_t fsmem = topk_smem_bytes(S16_SURV_CAP, (uint32_t) k);

#define S16_FIN_L(RR, HH, ...) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_finalize<RR, HH>), dim3(nq), dim3(256), fsmem, g.stream, __VA_ARGS__)
	S16_BY_RH(S16_FIN_L, d, d_q, w_probes, (const uint32_t *) ix->w_candoff, lco, npr, (uint32_t) k,
			  (const float2 *) ix->w_qthr, (const unsigned int *) ecount, (const uint2 *) ix->w_erec, ecap, partial,
			  d_cand, d_ncand, d_total, d_otid, d_odist, d_ocnt, surv, flags);
	hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(256), 0, g.stream, (const unsigned int *) surv, (uint32_t) nq,
					   g.d_counters + 3);
	hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(256), 0, g.stream, (const unsigned int *) ecount, (uint32_t) nq,
					   g.d_counters + 4);
	HIP_TRY(hipGetLastError());
	unsigned int over = 0;

	HIP_TRY(hipMemcpyAsync(&over, flags, sizeof(over), hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	if (getenv("NDBHIP_DEBUG_S16"))
	{
		std::vector<unsigned int> h((size_t) 3 * nq + 4);
		std::vector<float2> th((size_t) nq);

		HIP_TRY(hipMemcpy(h.data(), ix->w_ecount, h.size() * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(th.data(), ix->w_qthr, th.size() * 8, hipMemcpyDeviceToHost));
		unsigned int mx = 0, nact = 0, nover = 0, mxs = 0;
		int			arg = -1;
		for (int q = 0; q < nq; q++)
		{
			if (h[q] > mx) { mx = h[q]; arg = q; }
			nact += h[2 * (size_t) nq + q];
			nover += h[q] > ecap;
			mxs = std::max(mxs, h[(size_t) nq + q]);
		}
		fprintf(stderr, "s16 debug: nq %d max ecount %u (q %d, thrE %g E %g active %u) active %u still-over %u max surv %u flags %u %u\n",
				nq, mx, arg, arg >= 0 ? th[arg].x : 0.f, arg >= 0 ? th[arg].y : 0.f, arg >= 0 ? h[2 * (size_t) nq + arg] : 0u,
				nact, nover, mxs, h[3 * (size_t) nq], h[3 * (size_t) nq + 1]);
	}
	if (over)
	{
		g.stats.screen16_fallbacks++;
		return 1;
	}
	g.stats.screen16_batches++;
	return 0;
}

/* ------------------------------------------------------------------ */
/* synthetic data (ndbhip_gen.h): the same bits on host and device       */
/* ------------------------------------------------------------------ */
#include "ndbhip_gen.h"

__global__ void
k_gen_rows(int kind, uint64_t seed, uint64_t center_seed, uint64_t first_row, uint64_t n_elems, int dim, int components,
		   float sigma, float *__restrict__ out)
{
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (uint64_t) gridDim.x * blockDim.x)
		out[i] = ndb_gen_element(kind, seed, center_seed, first_row + i / (uint64_t) dim, (int) (i % (uint64_t) dim), dim,
								 components, sigma);
}

static int
gen_check(int kind, int64_t first_row, int64_t nrows, int dim, int components, const void *out)
{
	if ((kind != 0 && kind != 1) || first_row < 0 || nrows < 0 || dim < 1 || (kind == 1 && components < 1) || (nrows > 0 && !out))
		return fail(NDBHIP_ERR_INVALID, "bad generator arguments");
	return 0;
}

extern "C" int
ndbhip_gen_rows_device(int kind, uint64_t seed, uint64_t center_seed, int64_t first_row, int64_t nrows, int dim,
					   int components, float sigma, float *d_out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (gen_check(kind, first_row, nrows, dim, components, d_out))
		return NDBHIP_ERR_INVALID;
	if (nrows == 0)
		return NDBHIP_OK;
	const uint64_t n = (uint64_t) nrows * (uint64_t) dim;

	hipLaunchKernelGGL(k_gen_rows, dim3((unsigned) std::min<uint64_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, g.stream, kind,
					   seed, center_seed, (uint64_t) first_row, n, dim, components, sigma, d_out);
	HIP_TRY(hipGetLastError());
	return NDBHIP_OK;
}

extern "C" int
ndbhip_gen_rows_host(int kind, uint64_t seed, uint64_t center_seed, int64_t first_row, int64_t nrows, int dim,
					 int components, float sigma, float *out)
{
	if (gen_check(kind, first_row, nrows, dim, components, out))
		return NDBHIP_ERR_INVALID;
	for (int64_t r = 0; r < nrows; r++)
		for (int d = 0; d < dim; d++)
			out[(size_t) r * dim + d] = ndb_gen_element(kind, seed, center_seed, (uint64_t) (first_row + r), d, dim, components, sigma);
	return NDBHIP_OK;
}

extern "C" int
ndbhip_mfma_probe(const uint16_t *d_a, const uint16_t *d_b, const float *d_c, float *d_d, int ntiles, int chain)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (ntiles < 0 || chain < 0 || (ntiles > 0 && (!d_a || !d_b || !d_c || !d_d)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (ntiles == 0)
		return NDBHIP_OK;
	hipLaunchKernelGGL(k_s16_mfma_probe, dim3(ntiles), dim3(64), 0, g.stream, d_a, d_b, d_c, d_d, chain);
	HIP_TRY(hipGetLastError());
	return NDBHIP_OK;
}

extern "C" int
ndbhip_set_option(const char *name, int value)
{
	if (!name)
		return fail(NDBHIP_ERR_INVALID, "name is NULL");
	if (!strcmp(name, "screen16"))
		g_s16_auto = value != 0;
	else if (!strcmp(name, "screen16_records"))
	{
		if (value < 64 || value > 16384)
			return fail(NDBHIP_ERR_INVALID, "screen16_records must be 64..16384");
		g_s16_ecap = (uint32_t) value;
	}
	else if (!strcmp(name, "screen16_waves"))
	{
		if (value != 4 && value != 8)
			return fail(NDBHIP_ERR_INVALID, "screen16_waves must be 4 or 8");
		g_s16_waves = value;
	}
	else if (!strcmp(name, "screen16_debug"))
		g_s16_debug = value;
	else if (!strcmp(name, "screen"))
		g_screen_auto = value != 0;
	else
		return fail(NDBHIP_ERR_INVALID, "unknown option '%s'", name);
	return NDBHIP_OK;
}

__global__ void k_interleave16(const float *__restrict__ cents, int ncent, int dim, float *__restrict__ cblock);
template <bool SQRT, int CH>
__global__ void k_assign_grouped(const float *__restrict__ rows, uint32_t nrows, int dim,
								 const float *__restrict__ cblock, int ncent, float *__restrict__ part_dist,
								 int *__restrict__ part_idx, float *__restrict__ all_dist = nullptr,
								 uint32_t all_stride = 0);

/* queries already on the device; runs select (+ scan + topk when `full`) for one sub-batch */
static int
ivf_search_chunk(ndbhip_ivf *ix, const float *d_q, int nq, int strategy, int npr, int k,
				 int64_t max_candidates, uint32_t stride, bool full, int partial,
				 ndbhip_cand *d_cand, int *d_ncand, int64_t *d_total,
				 uint64_t *d_otid, float *d_odist, int *d_ocnt, const int *d_probes_in = nullptr,
				 int *d_probes_out = nullptr)
{
	const IvfDev d = ivf_dev(ix);
	const int	ncmp = std::min(ix->nlists, ix->ncent);
	const uint32_t cstride = (uint32_t) ((ncmp + 63) & ~63);
	/*
	 * Two position spaces: cand_off = place in the reference's candidates[] (what the selection
	 * replay orders by); lco = place among the rows held HERE (what addresses w_dist).  They only
	 * differ on a sharded mirror, whose scan / top-k then cost what its own lists cost.
	 */
	uint32_t   *lco_w = ix->sharded ? ix->w_candoff + (size_t) nq * (npr + 1) : nullptr;
	const uint32_t *lco = ix->sharded ? lco_w : ix->w_candoff;

	int		   *w_probes = d_probes_out ? d_probes_out : ix->w_probes;

	if (d_probes_in)
	{
		/* probes chosen elsewhere (each rank of a sharded search selects for its slice of the queries) */
		w_probes = const_cast<int *>(d_probes_in);
		hipLaunchKernelGGL(k_probe_offsets, dim3((nq + 255) / 256), dim3(256), 0, g.stream, d_probes_in,
						   (uint32_t) nq, npr, ix->ncent, (const uint32_t *) d.glob_len, d.own_lo, d.own_len,
						   (uint64_t) (max_candidates > 0 ? max_candidates : 0), ix->dim * (ix->f16 ? 2 : 4),
						   ix->w_candoff, lco_w, full ? g.d_counters : (unsigned long long *) nullptr);
	}
	else
	{
		/* HOT LOOP 1: query x centroid, always L2 (ivf_am.c:1676-1680); a small batch spreads its
		 * 64-centroid tiles over more CUs (one wave per block) */
		if (nq >= 64 && (ix->dim % NDB_CHUNK) == 0 && true)
		{
			/* a batch: the same 64 rows x 16 columns engine as the list scan and the build's assignment, the
			 * queries as rows and the centroids (interleaved 16 per block, 3 MB at 1024 x 768: redone per
			 * call, 4 us) as the scalar operand; (c - q)^2 == (q - c)^2 exactly */
			const int	ngroups = (ncmp + NDB_QG - 1) / NDB_QG;

			if (grow(ix->w_cblock, ix->w_cblock_n, (size_t) ngroups * ix->dim * NDB_QG)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_interleave16, dim3((ix->dim + 255) / 256, ngroups), dim3(256), 0, g.stream,
							   (const float *) d.centroids, ncmp, ix->dim, ix->w_cblock);
			const dim3	g1((unsigned) ((((size_t) (nq + 63) / 64 + 7) / 8) * 8 * (size_t) ngroups));

			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_assign_grouped<true, 32>), g1, dim3(64), 0, g.stream, d_q,
							   (uint32_t) nq, ix->dim, (const float *) ix->w_cblock, ncmp, (float *) nullptr,
							   (int *) nullptr, ix->w_cdist, cstride);
		}
		else
		{
		const int	rsw = nq <= 16 ? 1 : 4;
		dim3		grid((ncmp + 64 * rsw - 1) / (64 * rsw), nq);

		hipLaunchKernelGGL(k_rows_scan<R_IVF_L2>, grid, dim3(64 * rsw), 0, g.stream, (const float *) d.centroids,
						   (uint32_t) ncmp, ix->dim, d_q, ix->w_cdist, cstride);
		}
		hipLaunchKernelGGL(k_probe_select, dim3(nq), dim3(256), 0, g.stream, (const float *) ix->w_cdist, cstride,
						   ncmp, ix->ncent, npr, (const uint32_t *) d.glob_len, d.own_lo, d.own_len,
						   (uint64_t) (max_candidates > 0 ? max_candidates : 0), ix->dim * (ix->f16 ? 2 : 4),
						   w_probes, ix->w_candoff, lco_w, full ? g.d_counters : (unsigned long long *) nullptr);
	}
	HIP_TRY(hipGetLastError());
	if (!full)
		return 0;
	hipLaunchKernelGGL(k_sum_candidates, dim3(1), dim3(256), 0, g.stream, (const uint32_t *) ix->w_candoff,
					   (const uint32_t *) lco_w, (uint32_t) nq, npr, ix->dim * (ix->f16 ? 2 : 4), g.d_counters);

	/* HOT LOOP 2 */
	const int	R = ivf_recipe(strategy);

	/* batches of >= 128 queries: the bound pass on fp16 matrix cores (ndbhip_screen16.h); mode 5 forces it */
	if ((g_scan_mode == 5 || (g_scan_mode == 0 && g_screen_auto && g_s16_auto && nq >= NDB_SCREEN_MIN_NQ)) &&
		ivf_s16_eligible(ix, nq, R, k))
	{
		const int	rc = ivf_s16_run(ix, d, d_q, nq, R, npr, k, w_probes, lco, partial, d_cand, d_ncand, d_total,
									 d_otid, d_odist, d_ocnt);

		if (rc <= 0)
			return rc;
		/* some query emitted more than its record capacity: the older path has none */
	}
	/* one slot per (probe, 64-candidate tile): slot = (local offset of the probe >> 6) + probe + tile */
	const uint32_t tstride = (((stride >> 6) + (uint32_t) npr + 2u) + 63u) & ~63u;
	const bool	grouped = (ix->dim % NDB_CHUNK) == 0 &&
		(g_scan_mode >= 2 || (g_scan_mode == 0 && nq >= NDB_GROUPED_MIN_NQ));
	bool		screen = false;
	int			coop = 0;			/* bound pass: 0 one wave per item, 1 a block per 4 query groups of a tile */

	if (grouped)
	{
		const int	nc = ix->ncent;
		uint32_t   *cnt = ix->w_gcnt, *fill = ix->w_gcnt + nc;
		unsigned int *next_item = ix->w_gcnt + 2 * nc;
		uint32_t   *pair_off = ix->w_goff, *item_off = ix->w_goff + (nc + 1), *grp_off = ix->w_goff + 2 * (nc + 1);
		uint32_t   *runs = ix->w_goff + 3 * (nc + 1) + 32;	/* 9 words, on a line of their own */
		const uint32_t npairs = (uint32_t) nq * (uint32_t) npr;
		const uint32_t maxgroups = npairs / NDB_QG + (uint32_t) nc;
		ScanTimer	t;
		/* screened L2 scan (GAcc<R_SCR_L2>): float4 rows, 32-float chunks; mode 0 = auto, 3 = always, 4 = never */
		{
			/* 2 (default): a block per 128 rows x 4 query groups, 11 ms per 4096 queries; 1: per 64 rows x 4 groups,
			 * 13.1 ms; 0: the single-wave bound pass, 13.7 ms (NDBHIP_SCR_COOP for A/B).  Inner product and cosine
			 * are screened by the two-tile kernel only (its pass is the plain dot product; the norms come from
			 * the per-row norms) */
			static const int scr_coop = getenv("NDBHIP_SCR_COOP") ? atoi(getenv("NDBHIP_SCR_COOP")) : 2;
			const bool	want = g_scan_mode == 3 || (g_scan_mode == 0 && g_screen_auto && nq >= NDB_SCREEN_MIN_NQ);
			const bool	two_tile = scr_coop == 2 && (ix->dim % 16) == 0;

			/* fp16 rows (decoded when the tile is staged), inner product and cosine: the two-tile kernel only */
			screen = want && ((R == R_IVF_L2 && !ix->f16) || two_tile);
			coop = (screen && (ix->dim % 16) == 0) ? scr_coop : 0;
		}

		HIP_TRY(hipMemsetAsync(ix->w_gcnt, 0, (size_t) (2 * nc + 8 * NDB_QHEAD_STRIDE) * sizeof(uint32_t), g.stream));	/* + 8 queue heads */
		hipLaunchKernelGGL(k_pair_count, dim3((npairs + 255) / 256), dim3(256), 0, g.stream,
						   (const int *) w_probes, lco, npr, (uint32_t) nq, cnt);
		hipLaunchKernelGGL(k_pair_offsets, dim3(1), dim3(1024), 0, g.stream, (const uint32_t *) cnt,
						   d.own_len, nc, pair_off, item_off, grp_off, runs, coop ? 4u : 1u, coop == 2 ? 2u : 1u);
		hipLaunchKernelGGL(k_pair_fill, dim3((npairs + 255) / 256), dim3(256), 0, g.stream,
						   (const int *) w_probes, lco, npr, (uint32_t) nq, (const uint32_t *) pair_off, fill,
						   ix->w_pairs);
		hipLaunchKernelGGL(k_group_pack, dim3((ix->dim + 255) / 256, maxgroups), dim3(256), 0, g.stream, d_q,
						   ix->dim, nc, (const uint32_t *) cnt, (const uint32_t *) pair_off,
						   (const uint32_t *) grp_off, (const PairRec *) ix->w_pairs, ix->w_qblock);
		const dim3	pgrid(g.num_cus * 10);	/* one wave per block; LDS admits 10 per CU */

		if (R == R_IVF_COS || screen)
			hipLaunchKernelGGL(k_query_norms, dim3((nq + 63) / 64), dim3(64), 0, g.stream, d_q, (uint32_t) nq,
							   ix->dim, ix->w_qnorm);
		if (screen)
		{
			/* the largest row norm of the rows held here, once per version of the mirror */
			if (!ix->norm_valid)
			{
				if (!ix->d_xxmax)
					HIP_TRY(hipMalloc((void **) &ix->d_xxmax, sizeof(float)));
				HIP_TRY(hipMemsetAsync(ix->d_xxmax, 0, sizeof(float), g.stream));
				if (ix->nrows > 0)
				{
					if (grow(ix->w_rnorm, ix->w_rnorm_n, (size_t) ix->nrows + (size_t) ix->dim)) return NDBHIP_ERR_HIP;
					float	   *zero = ix->w_rnorm + ix->nrows;	/* a zero query: sum (0 - x)^2 = the row's norm^2 */

					HIP_TRY(hipMemsetAsync(zero, 0, (size_t) ix->dim * sizeof(float), g.stream));
					if (ix->f16)
					{
						const dim3	gn((unsigned) ((ix->nrows + 255) / 256));

						if (ix->f16_sub)
							hipLaunchKernelGGL(k_row_norms_h<true>, gn, dim3(256), 0, g.stream, (const uint16_t *) ix->d_vecs,
											   ix->nrows, ix->dim, ix->w_rnorm);
						else
							hipLaunchKernelGGL(k_row_norms_h<false>, gn, dim3(256), 0, g.stream, (const uint16_t *) ix->d_vecs,
											   ix->nrows, ix->dim, ix->w_rnorm);
					}
					else
					for (int64_t r0 = 0; r0 < ix->nrows; r0 += (int64_t) 1 << 30)
					{
						const uint32_t nr = (uint32_t) std::min<int64_t>((int64_t) 1 << 30, ix->nrows - r0);

						hipLaunchKernelGGL(k_rows_scan<R_IVF_L2SQ>, dim3((nr + 255) / 256, 1), dim3(256), 0, g.stream,
										   (const float *) ix->d_vecs + (size_t) r0 * ix->dim, nr, ix->dim,
										   (const float *) zero, ix->w_rnorm + r0, nr);
					}
					hipLaunchKernelGGL(k_max_nonneg, dim3(1024), dim3(256), 0, g.stream, (const float *) ix->w_rnorm,
									   ix->nrows, (uint32_t *) ix->d_xxmax);
				}
				ix->norm_valid = true;
			}
			hipLaunchKernelGGL(k_screen_eq, dim3((nq + 255) / 256), dim3(256), 0, g.stream, ix->w_qnorm, (uint32_t) nq,
							   ix->dim, (const float *) ix->d_xxmax);
		}
		HIP_TRY(hipMemsetAsync(ix->w_tmin, 0xFF, (size_t) nq * tstride * sizeof(uint32_t), g.stream));
		if (t.start()) return NDBHIP_ERR_HIP;	/* events bracket the dominant kernel only */

#define LAUNCH_GROUPED(RR, CC, GRID) LAUNCH_GROUPED_H(RR, CC, 0, GRID)
#define LAUNCH_GROUPED_H(RR, CC, HH, GRID)                                                                       \
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ivf_scan_grouped<RR, CC, HH>), GRID, dim3(64), 0, g.stream, d,    \
						   (const float *) ix->w_qblock, (const uint32_t *) ix->w_candoff, lco, npr,           \
						   (const uint32_t *) cnt, (const uint32_t *) pair_off, (const uint32_t *) item_off,   \
						   (const uint32_t *) grp_off, (const PairRec *) ix->w_pairs, next_item,               \
						   (const uint32_t *) runs, ix->w_dist, stride, (const float *) ix->w_qnorm, ix->w_tmin, tstride, \
						   nq < 1024 ? 1 : 0, (uint32_t) nq)
		if (ix->f16 && !screen)
		{
			const dim3	g16(g.num_cus * 16);	/* 8 KiB tile, 4 waves per SIMD */

			if (ix->f16_sub)
			{
				if (R == R_IVF_IP)
					LAUNCH_GROUPED_H(R_IVF_IP, 32, 1, g16);
				else if (R == R_IVF_COS)
					LAUNCH_GROUPED_H(R_IVF_COS, 32, 1, g16);
				else
					LAUNCH_GROUPED_H(R_IVF_L2, 32, 1, g16);
			}
			else
			{
				if (R == R_IVF_IP)
					LAUNCH_GROUPED_H(R_IVF_IP, 32, 2, g16);
				else if (R == R_IVF_COS)
					LAUNCH_GROUPED_H(R_IVF_COS, 32, 2, g16);
				else
					LAUNCH_GROUPED_H(R_IVF_L2, 32, 2, g16);
			}
		}
		else if (screen)
		{
			const dim3	g32(g.num_cus * 4 * NDB_G32_WAVES);

			/* the bound pass waits on row-tile fetches, not on the ALU: 16-float chunks (4 KiB tile) let 8 waves
			 * per SIMD overlap them instead of 5 (16.5 -> 13.6 ms per 4096 queries); NDBHIP_SCR_CH=32 for A/B */
			static const int scr_ch = getenv("NDBHIP_SCR_CH") ? atoi(getenv("NDBHIP_SCR_CH")) : 16;

#define LAUNCH_COOP2_H(RR, HH)                                                                                  \
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ivf_bound_coop2<RR, HH>), dim3(g.num_cus * NDB_COOP2_WAVES), dim3(256), 0, \
								   g.stream, d, (const float *) ix->w_qblock, lco, npr, (const uint32_t *) cnt,        \
								   (const uint32_t *) pair_off, (const uint32_t *) item_off, (const uint32_t *) grp_off, \
								   (const PairRec *) ix->w_pairs, next_item, (const uint32_t *) runs, ix->w_dist, stride, \
								   (const float *) ix->w_qnorm, ix->w_tmin, tstride, nq < 1024 ? 1 : 0, (uint32_t) nq, \
								   (const float *) ix->w_rnorm)
#define LAUNCH_COOP2(RR)                                                                                       \
				do {                                                                                                   \
					if (!ix->f16) LAUNCH_COOP2_H(RR, 0);                                                               \
					else if (ix->f16_sub) LAUNCH_COOP2_H(RR, 1);                                                       \
					else LAUNCH_COOP2_H(RR, 2);                                                                        \
				} while (0)
			/* the same items on the matrix cores (k_ivf_bound_mfma: the same fmaf chains, hence the same bits);
			 * NDBHIP_SCR_MFMA=0 keeps the vector-ALU kernel for A/B */
			static const int scr_mfma = getenv("NDBHIP_SCR_MFMA") ? atoi(getenv("NDBHIP_SCR_MFMA")) : 1;
#define LAUNCH_MFMA_H(RR, HH)                                                                                   \
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ivf_bound_mfma<RR, HH>), dim3(g.num_cus * NDB_MFMA_BLOCKS), dim3(256), 0, \
								   g.stream, d, (const float *) ix->w_qblock, lco, npr, (const uint32_t *) cnt,        \
								   (const uint32_t *) pair_off, (const uint32_t *) item_off, (const uint32_t *) grp_off, \
								   (const PairRec *) ix->w_pairs, next_item, (const uint32_t *) runs, ix->w_dist, stride, \
								   (const float *) ix->w_qnorm, ix->w_tmin, tstride, nq < 1024 ? 1 : 0, (uint32_t) nq, \
								   (const float *) ix->w_rnorm)
#define LAUNCH_MFMA(RR)                                                                                        \
				do {                                                                                                   \
					if (!ix->f16) LAUNCH_MFMA_H(RR, 0);                                                                \
					else if (ix->f16_sub) LAUNCH_MFMA_H(RR, 1);                                                        \
					else LAUNCH_MFMA_H(RR, 2);                                                                         \
				} while (0)
			if (coop == 2 && scr_mfma)
			{
				if (R == R_IVF_IP)
					LAUNCH_MFMA(R_IVF_IP);
				else if (R == R_IVF_COS)
					LAUNCH_MFMA(R_IVF_COS);
				else
					LAUNCH_MFMA(R_IVF_L2);
			}
			else if (coop == 2)
			{
				if (R == R_IVF_IP)
					LAUNCH_COOP2(R_IVF_IP);
				else if (R == R_IVF_COS)
					LAUNCH_COOP2(R_IVF_COS);
				else
					LAUNCH_COOP2(R_IVF_L2);
			}
			else if (coop)
				hipLaunchKernelGGL(k_ivf_bound_coop, dim3(g.num_cus * 8), dim3(256), 0, g.stream, d,
								   (const float *) ix->w_qblock, lco, npr, (const uint32_t *) cnt,
								   (const uint32_t *) pair_off, (const uint32_t *) item_off, (const uint32_t *) grp_off,
								   (const PairRec *) ix->w_pairs, next_item, (const uint32_t *) runs, ix->w_dist, stride,
								   (const float *) ix->w_qnorm, ix->w_tmin, tstride, nq < 1024 ? 1 : 0, (uint32_t) nq);
			else if (scr_ch == 16)
			{
				const dim3	g16w(g.num_cus * 4 * NDB_G16_WAVES);	/* 4 KiB tile */

				LAUNCH_GROUPED(R_SCR_L2, 16, g16w);
			}
			else if (g_gchunk == 32)
				LAUNCH_GROUPED(R_SCR_L2, 32, g32);
			else
				LAUNCH_GROUPED(R_SCR_L2, 64, pgrid);
		}
		else if (g_gchunk == 32)
		{
			const dim3	g32(g.num_cus * 4 * NDB_G32_WAVES);	/* 8 KiB LDS per wave, VGPRs capped for NDB_G32_WAVES per SIMD */

			if (R == R_IVF_IP)
				LAUNCH_GROUPED(R_IVF_IP, 32, g32);
			else if (R == R_IVF_COS)
				LAUNCH_GROUPED(R_IVF_COS, 32, g32);
			else
				LAUNCH_GROUPED(R_IVF_L2, 32, g32);
		}
		else
		{
			if (R == R_IVF_IP)
				LAUNCH_GROUPED(R_IVF_IP, 64, pgrid);
			else if (R == R_IVF_COS)
				LAUNCH_GROUPED(R_IVF_COS, 64, pgrid);
			else
				LAUNCH_GROUPED(R_IVF_L2, 64, pgrid);
		}
		if (t.stop()) return NDBHIP_ERR_HIP;
	}
	else
	{
		dim3		grid((stride + 255) / 256, nq);
		ScanTimer	t;

		if (t.start()) return NDBHIP_ERR_HIP;
		if (ix->f16)
			LAUNCH_BY_RECIPE(R, k_ivf_scan_h, grid, dim3(256), d, d_q, (const int *) w_probes, lco, npr,
							 ix->w_dist, stride);
		else
			LAUNCH_BY_RECIPE(R, k_ivf_scan, grid, dim3(256), d, d_q, (const int *) w_probes, lco, npr,
							 ix->w_dist, stride);
		if (t.stop()) return NDBHIP_ERR_HIP;
	}
	{
		const size_t smem = topk_smem_bytes(topk_entry_cap((uint32_t) k), (uint32_t) k);
		/*
		 * Small batches: one block per query is latency-bound (its 256 threads cannot keep enough of the
		 * distance buffer in flight), so the query's candidates are cut into position ranges — partial
		 * top-k per range, then the same replay merge the sharded search uses.
		 */
		uint32_t	nsplit = 1;

		if (!partial && nq <= 512 && !grouped)	/* the grouped scan leaves tile minima: one block per query is cheap */
		{
			const uint32_t by_work = stride / 2048u;					/* >= 2048 candidates per block */
			const uint32_t by_merge = 2048u / (3u * (uint32_t) k);		/* records the merge stage sorts in LDS */
			const uint32_t by_grid = 1024u / (uint32_t) nq;

			/* 16 ranges: the merge sorts 16 x 3k records instead of 64 x 3k (68 -> ~25 us per query) while a
			 * range's partial top-k stays short; measured p50 276 -> 242 us for one 1M x 768 query */
			nsplit = std::min(std::min(by_work, by_merge), std::min(by_grid, 16u));
			if (nsplit < 2 || topk_smem_bytes(3u * (uint32_t) k * nsplit, (uint32_t) k) > NDB_TOPK_MAX_SMEM)
				nsplit = 1;
		}
		if (nsplit > 1)
		{
			const size_t nrec = (size_t) nsplit * nq * 3 * k;

			if (grow(ix->w_scand, ix->w_scand_n, nrec)) return NDBHIP_ERR_HIP;
			if (grow(ix->w_sncand, ix->w_sncand_n, (size_t) nsplit * nq)) return NDBHIP_ERR_HIP;
			if (grow(ix->w_stotal, ix->w_stotal_n, (size_t) nq)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_ivf_topk, dim3(nq, nsplit), dim3(256), smem, g.stream, d, (const int *) w_probes,
							   (const uint32_t *) ix->w_candoff, lco, npr, (const float *) ix->w_dist, stride,
							   (uint32_t) k, 1, ix->w_scand, ix->w_sncand, ix->w_stotal, (uint64_t *) nullptr,
							   (float *) nullptr, (int *) nullptr, (uint32_t) nq, (const uint32_t *) nullptr, 0u);
			hipLaunchKernelGGL(k_merge_topk, dim3(nq), dim3(256),
							   topk_smem_bytes(3u * (uint32_t) k * nsplit, (uint32_t) k), g.stream,
							   (const ndbhip_cand *) ix->w_scand, (const int *) ix->w_sncand,
							   (const int64_t *) ix->w_stotal, (int) nsplit, nq, (uint32_t) k, 3u * (uint32_t) k,
							   d_otid, d_odist, d_ocnt);
		}
		else
		{
			if (screen)
			{
				/* first pass: the k smallest provisional distances -> the survivors' threshold; then the
				 * reference's arithmetic for the survivors; the top-k below sees exact values wherever it matters */
				if (grow(ix->w_scrt, ix->w_scrt_n, (size_t) nq * k)) return NDBHIP_ERR_HIP;
				if (grow(ix->w_scrd, ix->w_scrd_n, (size_t) nq * k)) return NDBHIP_ERR_HIP;
				if (grow(ix->w_scrc, ix->w_scrc_n, (size_t) nq)) return NDBHIP_ERR_HIP;
				hipLaunchKernelGGL(k_ivf_topk, dim3(nq), dim3(256), smem, g.stream, d, (const int *) w_probes,
								   (const uint32_t *) ix->w_candoff, lco, npr, (const float *) ix->w_dist, stride,
								   (uint32_t) k, 0, (ndbhip_cand *) nullptr, (int *) nullptr, (int64_t *) nullptr,
								   ix->w_scrt, ix->w_scrd, ix->w_scrc, (uint32_t) nq, (const uint32_t *) ix->w_tmin,
								   tstride);
				const uint32_t rec_cap = 256;	/* survivors listed per query; more are rescored in place */

				if (grow(ix->w_screc, ix->w_screc_n, (size_t) nq * rec_cap * 4 + (size_t) nq)) return NDBHIP_ERR_HIP;
				unsigned int *rec_counts = (unsigned int *) (ix->w_screc + (size_t) nq * rec_cap * 4);

#define LAUNCH_SECOND_PASS(RR)                                                                                 \
				do {                                                                                                   \
					if (!ix->f16) LAUNCH_SECOND_PASS_H(RR, 0);                                                         \
					else if (ix->f16_sub) LAUNCH_SECOND_PASS_H(RR, 1);                                                 \
					else LAUNCH_SECOND_PASS_H(RR, 2);                                                                  \
				} while (0)
#define LAUNCH_SECOND_PASS_H(RR, HH)                                                                           \
				do {                                                                                                   \
					hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ivf_survivors<RR, HH>), dim3(nq), dim3(256), 0, g.stream, d, d_q, \
									   (const int *) w_probes, lco, npr, ix->w_dist, stride, ix->w_tmin, tstride,       \
									   (const float *) ix->w_qnorm, (uint32_t) nq, (uint32_t) k,                          \
									   (const float *) ix->w_scrd, (const int *) ix->w_scrc, (ScrRec *) ix->w_screc,      \
									   rec_cap, rec_counts, g.d_counters);                                             \
					hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ivf_rescore_list<RR, HH>), dim3(rec_cap / 64, nq), dim3(64), 0, \
									   g.stream, d, d_q, ix->w_dist, stride, ix->w_tmin, tstride,                         \
									   (const ScrRec *) ix->w_screc, rec_cap, (const unsigned int *) rec_counts,          \
									   g.d_counters);                                                                  \
				} while (0)
				if (R == R_IVF_IP)
					LAUNCH_SECOND_PASS(R_IVF_IP);
				else if (R == R_IVF_COS)
					LAUNCH_SECOND_PASS(R_IVF_COS);
				else
					LAUNCH_SECOND_PASS(R_IVF_L2);
				hipLaunchKernelGGL(k_sum_u32, dim3(1), dim3(256), 0, g.stream, (const unsigned int *) rec_counts,
								   (uint32_t) nq, g.d_counters + 3);
			}
			hipLaunchKernelGGL(k_ivf_topk, dim3(nq), dim3(256), smem, g.stream, d, (const int *) w_probes,
							   (const uint32_t *) ix->w_candoff, lco, npr, (const float *) ix->w_dist, stride,
							   (uint32_t) k, partial, d_cand, d_ncand, d_total, d_otid, d_odist, d_ocnt, (uint32_t) nq,
							   grouped ? (const uint32_t *) ix->w_tmin : (const uint32_t *) nullptr, tstride);
		}
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

static int
ivf_check_search_args(ndbhip_ivf *ix, int nq, int nprobe, int k)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix)
		return fail(NDBHIP_ERR_INVALID, "index is NULL");
	if (!ix->loaded || ix->ncent < 1)
		return fail(NDBHIP_ERR_STATE, "index has no centroids/lists loaded");
	if (nq < 0)
		return fail(NDBHIP_ERR_INVALID, "nq < 0");
	if (nprobe < 1 || nprobe > NDBHIP_MAX_NPROBE)
		return fail(NDBHIP_ERR_INVALID, "nprobe %d out of range 1..%d", nprobe, NDBHIP_MAX_NPROBE);
	if (k < 1 || k > NDBHIP_MAX_K)
		return fail(NDBHIP_ERR_INVALID, "k %d out of range 1..%d", k, NDBHIP_MAX_K);
	return 0;
}

/* per-sub-batch budget for the candidate-distance buffer */
static size_t g_dist_budget_bytes = (size_t) 8 << 30;	/* of 288 GB: a 4096-query step of the 1M x 768 workload needs 2.1 GB */

static int
ivf_search_device_impl(ndbhip_ivf *ix, const float *d_queries, int nq, int strategy, int nprobe, int k,
					   int64_t max_candidates, int partial, ndbhip_cand *d_cand, int *d_ncand,
					   int64_t *d_total, uint64_t *d_otid, float *d_odist, int *d_ocnt,
					   const int *d_probes_in = nullptr)
{
	int			rc = ivf_check_search_args(ix, nq, nprobe, k);

	if (rc)
		return rc;
	rc = ivf_flush(ix);
	if (rc)
		return rc;
	if (nq == 0)
		return NDBHIP_OK;
	if (topk_smem_bytes(topk_entry_cap((uint32_t) k), (uint32_t) k) > NDB_TOPK_MAX_SMEM)
		return fail(NDBHIP_ERR_UNSUPPORTED, "k too large for the LDS top-k stage");

	int64_t		maxc = ivf_local_max_candidates(ix, nprobe);

	if (max_candidates > 0 && maxc > max_candidates)
		maxc = max_candidates;
	if (maxc > 0xFFFFFF00ll)
		return fail(NDBHIP_ERR_UNSUPPORTED, "more than 2^32 candidates per query");
	const uint32_t stride = (uint32_t) std::max<int64_t>(64, (maxc + 63) & ~63ll);
	int			qb = (int) std::max<size_t>(1, g_dist_budget_bytes / ((size_t) stride * 4));

	qb = std::min(qb, std::min(nq, 65535));
	const int	ncmp = std::min(ix->nlists, ix->ncent);
	const size_t cstride = (size_t) ((ncmp + 63) & ~63);

	if (grow(ix->w_cdist, ix->w_cdist_n, (size_t) qb * cstride)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_probes, ix->w_probes_n, (size_t) qb * nprobe)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_candoff, ix->w_candoff_n, (size_t) 2 * qb * (nprobe + 1))) return NDBHIP_ERR_HIP;
	if (grow(ix->w_dist, ix->w_dist_n, (size_t) qb * stride)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_gcnt, ix->w_gcnt_n, (size_t) 2 * ix->ncent + 8 * NDB_QHEAD_STRIDE + 16)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_goff, ix->w_goff_n, (size_t) 3 * (ix->ncent + 1) + 80)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_pairs, ix->w_pairs_n, (size_t) qb * nprobe)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_qnorm, ix->w_qnorm_n, (size_t) 2 * qb)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_tmin, ix->w_tmin_n, (size_t) qb * ((((stride >> 6) + (size_t) nprobe + 2) + 63) & ~(size_t) 63)))
		return NDBHIP_ERR_HIP;
	if ((ix->dim % NDB_CHUNK) == 0 && g_scan_mode != 1 && (qb >= NDB_GROUPED_MIN_NQ || g_scan_mode == 2))
		if (grow(ix->w_qblock, ix->w_qblock_n,
				 ((size_t) qb * nprobe / NDB_QG + (size_t) ix->ncent) * (size_t) ix->dim * NDB_QG))
			return NDBHIP_ERR_HIP;

	const uint32_t cap = 3u * (uint32_t) k;

	for (int q0 = 0; q0 < nq; q0 += qb)
	{
		const int	n = std::min(qb, nq - q0);

		rc = ivf_search_chunk(ix, d_queries + (size_t) q0 * ix->dim, n, strategy, nprobe, k, max_candidates,
							  stride, true, partial,
							  d_cand ? d_cand + (size_t) q0 * cap : nullptr,
							  d_ncand ? d_ncand + q0 : nullptr, d_total ? d_total + q0 : nullptr,
							  d_otid ? d_otid + (size_t) q0 * k : nullptr,
							  d_odist ? d_odist + (size_t) q0 * k : nullptr, d_ocnt ? d_ocnt + q0 : nullptr,
							  d_probes_in ? d_probes_in + (size_t) q0 * nprobe : nullptr);
		if (rc)
			return rc;
	}
	g.stats.queries += (uint64_t) nq;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_search_device(ndbhip_ivf *ix, const float *d_queries, int nq, int strategy, int nprobe, int k,
						 int64_t max_candidates, uint64_t *d_out_tids, float *d_out_dist, int *d_out_count)
{
	if (nq > 0 && (!d_queries || !d_out_tids || !d_out_dist || !d_out_count))
		return fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	return ivf_search_device_impl(ix, d_queries, nq, strategy, nprobe, k, max_candidates, 0, nullptr, nullptr,
								  nullptr, d_out_tids, d_out_dist, d_out_count);
}

extern "C" int
ndbhip_ivf_search_partial_device(ndbhip_ivf *ix, const float *d_queries, int nq, int strategy, int nprobe,
								 int k, int64_t max_candidates, ndbhip_cand *d_out_cand, int *d_out_ncand,
								 int64_t *d_out_total)
{
	if (nq > 0 && (!d_queries || !d_out_cand || !d_out_ncand || !d_out_total))
		return fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	return ivf_search_device_impl(ix, d_queries, nq, strategy, nprobe, k, max_candidates, 1, d_out_cand,
								  d_out_ncand, d_out_total, nullptr, nullptr, nullptr);
}

extern "C" int
ndbhip_ivf_search_partial_probes_device(ndbhip_ivf *ix, const float *d_queries, int nq, int strategy, int nprobe,
										int k, int64_t max_candidates, const int *d_probes,
										ndbhip_cand *d_out_cand, int *d_out_ncand, int64_t *d_out_total)
{
	if (nq > 0 && (!d_queries || !d_probes || !d_out_cand || !d_out_ncand || !d_out_total))
		return fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	return ivf_search_device_impl(ix, d_queries, nq, strategy, nprobe, k, max_candidates, 1, d_out_cand,
								  d_out_ncand, d_out_total, nullptr, nullptr, nullptr, d_probes);
}

extern "C" int
ndbhip_ivf_select_clusters_device(ndbhip_ivf *ix, const float *d_queries, int nq, int nprobe, int *d_out_probes)
{
	int			rc = ivf_check_search_args(ix, nq, nprobe, 1);

	if (rc)
		return rc;
	if (nq == 0)
		return NDBHIP_OK;
	if (!d_queries || !d_out_probes)
		return fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	rc = ivf_flush(ix);
	if (rc)
		return rc;
	const int	ncmp = std::min(ix->nlists, ix->ncent);
	const size_t cstride = (size_t) ((ncmp + 63) & ~63);
	const int	qb = std::min(nq, 4096);

	if (grow(ix->w_cdist, ix->w_cdist_n, (size_t) qb * cstride)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_candoff, ix->w_candoff_n, (size_t) 2 * qb * (nprobe + 1))) return NDBHIP_ERR_HIP;
	for (int q0 = 0; q0 < nq; q0 += qb)
	{
		const int	n = std::min(qb, nq - q0);

		rc = ivf_search_chunk(ix, d_queries + (size_t) q0 * ix->dim, n, 1, nprobe, 1, 0, 0, false, 0, nullptr,
							  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
							  d_out_probes + (size_t) q0 * nprobe);
		if (rc)
			return rc;
	}
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_search(ndbhip_ivf *ix, const float *queries, int nq, int strategy, int nprobe, int k,
				  int64_t max_candidates, uint8_t *out_tids6, float *out_dist, int *out_count)
{
	int			rc = ivf_check_search_args(ix, nq, nprobe, k);

	if (rc)
		return rc;
	if (nq == 0)
		return NDBHIP_OK;
	if (!queries || !out_tids6 || !out_dist || !out_count)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	if (grow(ix->w_q, ix->w_q_n, (size_t) nq * ix->dim)) return NDBHIP_ERR_HIP;
	/* one device block for the results — [TIDs | distances | counts] — and one pinned host block for the query
	 * and the results: a call is one H2D, the kernels, one D2H (what one amgettuple costs: every pageable copy
	 * is ~10 us of it) */
	const size_t nk = (size_t) nq * k;
	const size_t out_words = nk * 2 + nk + (size_t) nq;		/* in 4-byte words */

	if (grow(ix->w_otid, ix->w_otid_n, (out_words + 1) / 2)) return NDBHIP_ERR_HIP;	/* uint64 units */
	uint64_t   *d_tid = ix->w_otid;
	float	   *d_dist = (float *) (ix->w_otid + nk);
	int		   *d_cnt = (int *) (d_dist + nk);
	const size_t pin_bytes = (size_t) nq * ix->dim * sizeof(float) + 8 + out_words * 4;

	if (pin_bytes > ix->pin_n)
	{
		if (ix->pin) HIP_TRY(hipHostFree(ix->pin));
		ix->pin = nullptr;
		ix->pin_n = 0;
		HIP_TRY(hipHostMalloc((void **) &ix->pin, pin_bytes, hipHostMallocDefault));
		ix->pin_n = pin_bytes;
	}
	float	   *h_q = (float *) ix->pin;
	unsigned char *h_out = (unsigned char *) ix->pin + (((size_t) nq * ix->dim * sizeof(float) + 7) & ~(size_t) 7);

	memcpy(h_q, queries, (size_t) nq * ix->dim * sizeof(float));
	HIP_TRY(hipMemcpyAsync(ix->w_q, h_q, (size_t) nq * ix->dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	rc = ivf_search_device_impl(ix, ix->w_q, nq, strategy, nprobe, k, max_candidates, 0, nullptr, nullptr,
								nullptr, d_tid, d_dist, d_cnt);
	if (rc)
		return rc;
	HIP_TRY(hipMemcpyAsync(h_out, d_tid, out_words * 4, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	const uint64_t *t64 = (const uint64_t *) h_out;

	memcpy(out_dist, h_out + nk * 8, nk * 4);
	memcpy(out_count, h_out + nk * 8 + nk * 4, (size_t) nq * 4);
	for (int q = 0; q < nq; q++)
		for (int i = 0; i < k; i++)
		{
			if (i < out_count[q])
				ndb_tid_unpack(t64[(size_t) q * k + i], out_tids6 + ((size_t) q * k + i) * 6);
			else
			{
				memset(out_tids6 + ((size_t) q * k + i) * 6, 0, 6);
				out_dist[(size_t) q * k + i] = 0.0f;
			}
		}
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_select_clusters(ndbhip_ivf *ix, const float *queries, int nq, int nprobe, int *out_probes)
{
	int			rc = ivf_check_search_args(ix, nq, nprobe, 1);

	if (rc)
		return rc;
	if (nq == 0)
		return NDBHIP_OK;
	if (!queries || !out_probes)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	rc = ivf_flush(ix);
	if (rc)
		return rc;
	const int	ncmp = std::min(ix->nlists, ix->ncent);
	const size_t cstride = (size_t) ((ncmp + 63) & ~63);
	const int	qb = 4096;

	if (grow(ix->w_q, ix->w_q_n, (size_t) nq * ix->dim)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_cdist, ix->w_cdist_n, (size_t) std::min(qb, nq) * cstride)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_probes, ix->w_probes_n, (size_t) std::min(qb, nq) * nprobe)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_candoff, ix->w_candoff_n, (size_t) 2 * std::min(qb, nq) * (nprobe + 1))) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemcpyAsync(ix->w_q, queries, (size_t) nq * ix->dim * sizeof(float), hipMemcpyHostToDevice,
						   g.stream));
	for (int q0 = 0; q0 < nq; q0 += qb)
	{
		const int	n = std::min(qb, nq - q0);

		rc = ivf_search_chunk(ix, ix->w_q + (size_t) q0 * ix->dim, n, 1, nprobe, 1, 0, 0, false, 0, nullptr,
							  nullptr, nullptr, nullptr, nullptr, nullptr);
		if (rc)
			return rc;
		HIP_TRY(hipMemcpyAsync(out_probes + (size_t) q0 * nprobe, ix->w_probes, (size_t) n * nprobe * 4,
							   hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
	}
	return NDBHIP_OK;
}

/* ================================================================== */
/* shard merge                                                         */
/* ================================================================== */

extern "C" int
ndbhip_merge_topk_device(const ndbhip_cand *d_cand, const int *d_ncand, const int64_t *d_total, int world,
						 int nq, int k, int cap, uint64_t *d_out_tids, float *d_out_dist, int *d_out_count)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (world < 1 || world > 64 || nq < 0 || k < 1 || cap < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (nq == 0)
		return NDBHIP_OK;
	const size_t smem = topk_smem_bytes((uint32_t) cap * world, (uint32_t) k);

	if (smem > NDB_TOPK_MAX_SMEM)
		return fail(NDBHIP_ERR_UNSUPPORTED, "world*cap=%d records do not fit the LDS merge stage", cap * world);
	hipLaunchKernelGGL(k_merge_topk, dim3(nq), dim3(256), smem, g.stream, d_cand, d_ncand, d_total, world, nq,
					   (uint32_t) k, (uint32_t) cap, d_out_tids, d_out_dist, d_out_count);
	HIP_TRY(hipGetLastError());
	return NDBHIP_OK;
}

extern "C" int
ndbhip_merge_topk_host(const ndbhip_cand *cand, const int *ncand, const int64_t *total, int world, int nq,
					   int k, int cap, uint64_t *out_tids, float *out_dist, int *out_count)
{
	if (!cand || !ncand || !total || !out_tids || !out_dist || !out_count || world < 1 || nq < 0 || k < 1 ||
		cap < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	std::vector<uint32_t> key, pos;
	std::vector<uint64_t> comp;
	std::vector<uint32_t> idx;
	std::vector<uint8_t> taken;
	std::vector<int> order((size_t) k);
	std::vector<const ndbhip_cand *> ent;

	for (int q = 0; q < nq; q++)
	{
		ent.clear();
		for (int w = 0; w < world; w++)
		{
			const int	n = ncand[(size_t) w * nq + q];

			if (n < 0 || n > cap)
				return fail(NDBHIP_ERR_INVALID, "ncand out of range");
			for (int j = 0; j < n; j++)
				ent.push_back(cand + ((size_t) w * nq + q) * cap + j);
		}
		const int	n = (int) ent.size();
		int64_t		kk64 = std::min<int64_t>(k, total[q]);
		int			kk = (int) std::min<int64_t>(kk64, n);

		/* sort by (order key, position), keep the tie-complete prefix, replay */
		idx.resize(n);
		for (int j = 0; j < n; j++)
			idx[j] = j;
		std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
			const uint64_t ca = ((uint64_t) ndb_key_from_bits(ent[a]->key) << 32) | ent[a]->pos;
			const uint64_t cb = ((uint64_t) ndb_key_from_bits(ent[b]->key) << 32) | ent[b]->pos;

			return ca < cb;
		});
		int			ns = n;

		if (kk > 0)
		{
			const uint32_t Tkey = ndb_key_from_bits(ent[idx[kk - 1]]->key);
			int			first = 0;

			while (first < kk && ndb_key_from_bits(ent[idx[first]]->key) < Tkey)
				first++;
			ns = std::min(n, first + 2 * k);
		}
		key.resize(ns);
		pos.resize(ns);
		taken.resize(ns ? ns : 1);
		for (int j = 0; j < ns; j++)
		{
			key[j] = ndb_key_from_bits(ent[idx[j]]->key);
			pos[j] = ent[idx[j]]->pos;
		}
		const int	got = ndb_replay_selection_host(key.data(), pos.data(), taken.data(), ns, k, total[q],
													order.data());

		for (int i = 0; i < k; i++)
		{
			if (i < got)
			{
				const ndbhip_cand *c = ent[idx[order[i]]];

				out_tids[(size_t) q * k + i] = c->tid;
				out_dist[(size_t) q * k + i] = ndb_u2f(c->key);
			}
			else
			{
				out_tids[(size_t) q * k + i] = 0;
				out_dist[(size_t) q * k + i] = 0.0f;
			}
		}
		out_count[q] = got;
	}
	return NDBHIP_OK;
}

/* ================================================================== */
/* operator kernels (<->, <=>, <#> without an index): one lane = one    */
/* (query, row) pair in the reference's own order and width.            */
/*   OP_SCALAR  src/vector/vector_distance.c:93-122 (Kahan, double),    */
/*              145-157, 180-213 — what a default x86-64 build runs     */
/*              (Q16)                                                   */
/*   OP_SIMD    src/vector/vector_distance_simd.c:158-392: LANES fp32   */
/*              accumulators (8 = AVX2, 16 = AVX-512), cosine with FMA, */
/*              the fixed horizontal-sum tree (:84-137), scalar tail    */
/*   halfvec    src/types/quantization.c:1985-2116: both operands       */
/*              decoded per element (fp16_to_float incl. Q20), double   */
/* ================================================================== */

template <int LANES>
__device__ __forceinline__ float
op_hsum(const float (&v)[16])
{
	float		s8[8], s4[4];

#pragma unroll
	for (int j = 0; j < 8; j++)
		s8[j] = LANES == 16 ? v[j] + v[j + 8] : v[j];
#pragma unroll
	for (int j = 0; j < 4; j++)
		s4[j] = s8[j] + s8[j + 4];
	const float t0 = s4[0] + s4[1];
	const float t2 = s4[2] + s4[3];

	return t0 + t2;
}

/* strategy 1 L2, 2 cosine, 3 inner product (the dispatcher's sign: +sum from the SIMD paths, Q15) */
template <int LANES>
__device__ float
op_simd_pair(const float *__restrict__ a, const float *__restrict__ b, int dim, int strategy)
{
	float		acc0[16], acc1[16], acc2[16];
	const int	simd_end = (dim / LANES) * LANES;

#pragma unroll
	for (int j = 0; j < 16; j++)
		acc0[j] = acc1[j] = acc2[j] = 0.0f;
	for (int i = 0; i < simd_end; i += LANES)
	{
#pragma unroll
		for (int j = 0; j < LANES; j++)
		{
			const float va = a[i + j], vb = b[i + j];

			if (strategy == 1)
			{
				const float diff = va - vb;
				const float sq = diff * diff;

				acc0[j] = acc0[j] + sq;
			}
			else if (strategy == 3)
			{
				const float prod = va * vb;

				acc0[j] = acc0[j] + prod;
			}
			else
			{
				acc0[j] = __builtin_fmaf(va, vb, acc0[j]);	/* _mm256_fmadd_ps */
				acc1[j] = __builtin_fmaf(va, va, acc1[j]);
				acc2[j] = __builtin_fmaf(vb, vb, acc2[j]);
			}
		}
	}
	float		s0 = op_hsum<LANES>(acc0);

	if (strategy == 1)
	{
		for (int i = simd_end; i < dim; i++)
		{
			const float diff = a[i] - b[i];

			s0 = s0 + diff * diff;
		}
		return __builtin_sqrtf(s0);
	}
	if (strategy == 3)
	{
		for (int i = simd_end; i < dim; i++)
			s0 = s0 + a[i] * b[i];
		return s0;
	}
	float		na = op_hsum<LANES>(acc1), nb = op_hsum<LANES>(acc2);

	for (int i = simd_end; i < dim; i++)
	{
		const float va = a[i], vb = b[i];

		s0 = s0 + va * vb;
		na = na + va * va;
		nb = nb + vb * vb;
	}
	if (na == 0.0f || nb == 0.0f)
		return 1.0f;
	return 1.0f - (s0 / (__builtin_sqrtf(na) * __builtin_sqrtf(nb)));
}

/* H16: operands are fp16 images (halfvec), else float4; scalar double paths */
template <bool H16>
__device__ float
op_scalar_pair(const void *__restrict__ pa, const void *__restrict__ pb, int dim, int strategy)
{
	auto		ld = [&](const void *p, int i) -> double {
		if (H16)
			return (double) h2f_ref(((const uint16_t *) p)[i]);
		return (double) ((const float *) p)[i];
	};

	if (strategy == 1)
	{
		double		c = 0.0, sum = 0.0;

		for (int i = 0; i < dim; i++)
		{
			const double diff = ld(pa, i) - ld(pb, i);

			if (H16)
				sum = sum + diff * diff;	/* quantization.c:1997-2004: plain double sum */
			else
			{
				const double y = (diff * diff) - c;	/* Kahan: vector_distance.c:104-115 */
				const double t = sum + y;

				c = (t - sum) - y;
				sum = t;
			}
		}
		return (float) __builtin_sqrt(sum);
	}
	if (strategy == 3)
	{
		double		sum = 0.0;

		for (int i = 0; i < dim; i++)
			sum = sum + ld(pa, i) * ld(pb, i);
		/* halfvec_inner_product returns -sum (:2114); the float4 dispatcher negates the scalar kernel's -sum
		 * back to +sum (vector_distance_simd.c:511-558, Q15) */
		return H16 ? (float) (-sum) : -((float) (-sum));
	}
	double		dot = 0.0, na = 0.0, nb = 0.0;

	for (int i = 0; i < dim; i++)
	{
		const double va = ld(pa, i), vb = ld(pb, i);

		dot = dot + va * vb;
		na = na + va * va;
		nb = nb + vb * vb;
	}
	if (na == 0.0 || nb == 0.0)
		return 1.0f;
	return (float) (1.0 - (dot / (__builtin_sqrt(na) * __builtin_sqrt(nb))));
}

/* MODE 0 scalar float4, 8 / 16 SIMD emulation (falls back to scalar below LANES dims, as the dispatchers
 * do), 1 halfvec */
template <int MODE>
__global__ __launch_bounds__(256) void
k_op_distance(const void *__restrict__ queries, const void *__restrict__ vectors, float *__restrict__ out,
			  uint32_t nv, int dim, int strategy)
{
	const uint32_t v = blockIdx.x * 256 + threadIdx.x;
	const uint32_t q = blockIdx.y;
	constexpr size_t esz = MODE == 1 ? 2 : 4;

	if (v >= nv)
		return;
	const char *a = (const char *) queries + (size_t) q * dim * esz;
	const char *b = (const char *) vectors + (size_t) v * dim * esz;
	float		r;

	if (MODE == 1)
		r = op_scalar_pair<true>(a, b, dim, strategy);
	else if (MODE == 8 && dim >= 8)
		r = op_simd_pair<8>((const float *) a, (const float *) b, dim, strategy);
	else if (MODE == 16 && dim >= 16)
		r = op_simd_pair<16>((const float *) a, (const float *) b, dim, strategy);
	else
		r = op_scalar_pair<false>(a, b, dim, strategy);
	out[(size_t) q * nv + v] = r;
}

/* ================================================================== */
/* batch distance                                                      */
/* ================================================================== */

extern "C" int
ndbhip_batch_distance(const float *queries, const float *vectors, float *results, int nq, int nv, int dim,
					  int strategy, int recipe)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (nq < 0 || nv < 0 || dim < 1 || dim > 32767)
		return fail(NDBHIP_ERR_INVALID, "bad sizes");
	if (nq == 0 || nv == 0)
		return NDBHIP_OK;
	if (!queries || !vectors || !results)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	int			R = 0;

	if (recipe == 0)
		R = (strategy == 4) ? R_IVF_L2SQ : ivf_recipe(strategy);
	else if (recipe == 1)
	{
		if (strategy < 1 || strategy > 3)
			return fail(NDBHIP_ERR_UNSUPPORTED, "hnsw: unsupported distance strategy %d", strategy);
		R = R_HNSW_L2 + (strategy - 1);
	}
	else if (recipe >= 2 && recipe <= 5)
	{
		if (strategy < 1 || strategy > 3)
			return fail(NDBHIP_ERR_UNSUPPORTED, "operator kernels: strategy must be 1 (<->), 2 (<=>) or 3 (<#>)");
	}
	else
		return fail(NDBHIP_ERR_INVALID, "recipe must be 0 (ivf), 1 (hnsw), 2 (operator, scalar build), "
					"3 (operator, AVX2 build), 4 (operator, AVX-512 build) or 5 (halfvec operators)");
	const size_t esz = recipe == 5 ? 2 : 4;	/* recipe 5: queries / vectors are fp16 images */
	float	   *d_q = nullptr, *d_v = nullptr, *d_o = nullptr;

	HIP_TRY(hipMalloc((void **) &d_q, (size_t) nq * dim * esz));
	HIP_TRY(hipMalloc((void **) &d_v, (size_t) nv * dim * esz));
	HIP_TRY(hipMalloc((void **) &d_o, (size_t) nq * nv * 4));
	HIP_TRY(hipMemcpyAsync(d_q, queries, (size_t) nq * dim * esz, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_v, vectors, (size_t) nv * dim * esz, hipMemcpyHostToDevice, g.stream));
	for (int q0 = 0; q0 < nq; q0 += 65535)
	{
		const int	n = std::min(65535, nq - q0);
		dim3		grid((nv + 255) / 256, n);
		const void *qp = (const char *) d_q + (size_t) q0 * dim * esz;
		float	   *op = d_o + (size_t) q0 * nv;

		if (recipe <= 1)
			LAUNCH_BY_RECIPE(R, k_rows_scan, grid, dim3(256), (const float *) d_v, (uint32_t) nv, dim,
							 (const float *) qp, op, (uint32_t) nv);
		else if (recipe == 2)
			hipLaunchKernelGGL(k_op_distance<0>, grid, dim3(256), 0, g.stream, qp, (const void *) d_v, op,
							   (uint32_t) nv, dim, strategy);
		else if (recipe == 3)
			hipLaunchKernelGGL(k_op_distance<8>, grid, dim3(256), 0, g.stream, qp, (const void *) d_v, op,
							   (uint32_t) nv, dim, strategy);
		else if (recipe == 4)
			hipLaunchKernelGGL(k_op_distance<16>, grid, dim3(256), 0, g.stream, qp, (const void *) d_v, op,
							   (uint32_t) nv, dim, strategy);
		else
			hipLaunchKernelGGL(k_op_distance<1>, grid, dim3(256), 0, g.stream, qp, (const void *) d_v, op,
							   (uint32_t) nv, dim, strategy);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(results, d_o, (size_t) nq * nv * 4, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipFree(d_q));
	HIP_TRY(hipFree(d_v));
	HIP_TRY(hipFree(d_o));
	g.host_rows += (uint64_t) nq * nv;
	g.host_bytes += (uint64_t) nq * nv * dim * esz;
	return NDBHIP_OK;
}


/* out[i] = the SQL operator's distance of the PAIR (A[i], B[i]): the pairwise shape of the reference's GPU
 * vtable launchers (include/neurondb_gpu_backend.h:54-65), with the arithmetic of the CPU functions they fall
 * back to (src/vector/vector_distance.c:93-227), so a result does not depend on whether the device served it */
__global__ __launch_bounds__(256) void
k_op_pairs(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, uint32_t n, int dim,
		   int strategy)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;

	if (i < n)
		out[i] = op_scalar_pair<false>((const char *) (a + (size_t) i * dim), (const char *) (b + (size_t) i * dim),
									   dim, strategy);
}

extern "C" int
ndbhip_pair_distance(const float *A, const float *B, float *out, int n, int dim, int strategy)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (n < 0 || dim < 1 || dim > 32767)
		return fail(NDBHIP_ERR_INVALID, "bad sizes");
	if (strategy < 1 || strategy > 3)
		return fail(NDBHIP_ERR_UNSUPPORTED, "operator kernels: strategy must be 1 (<->), 2 (<=>) or 3 (<#>)");
	if (n == 0)
		return NDBHIP_OK;
	if (!A || !B || !out)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	float	   *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
	const size_t bytes = (size_t) n * dim * sizeof(float);

	HIP_TRY(hipMalloc((void **) &d_a, bytes));
	HIP_TRY(hipMalloc((void **) &d_b, bytes));
	HIP_TRY(hipMalloc((void **) &d_o, (size_t) n * sizeof(float)));
	HIP_TRY(hipMemcpyAsync(d_a, A, bytes, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_b, B, bytes, hipMemcpyHostToDevice, g.stream));
	hipLaunchKernelGGL(k_op_pairs, dim3((n + 255) / 256), dim3(256), 0, g.stream, (const float *) d_a,
					   (const float *) d_b, d_o, (uint32_t) n, dim, strategy);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(out, d_o, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipFree(d_a));
	HIP_TRY(hipFree(d_b));
	HIP_TRY(hipFree(d_o));
	g.host_rows += (uint64_t) n;
	g.host_bytes += (uint64_t) n * dim * 8;
	return NDBHIP_OK;
}

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
set_kernel_attributes_hnsw();

static int
set_kernel_attributes_build()
{
	HIP_TRY(hipFuncSetAttribute((const void *) k_kmeans_update, hipFuncAttributeMaxDynamicSharedMemorySize,
								NDB_TOPK_MAX_SMEM));
	HIP_TRY(hipFuncSetAttribute((const void *) k_seq_sum, hipFuncAttributeMaxDynamicSharedMemorySize,
								NDB_TOPK_MAX_SMEM));
	return set_kernel_attributes_hnsw();
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
		if (pd) HIP_TRY(hipFree(pd));
		if (pi) HIP_TRY(hipFree(pi));
		if (cblock) HIP_TRY(hipFree(cblock));
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

extern "C" int
ndbhip_ivf_assign_device(const float *d_centroids, int ncentroids, int dim, const float *d_rows,
						 int64_t nrows, int *d_out_list)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!d_centroids || ncentroids < 1 || dim < 1 || nrows < 0 || (nrows > 0 && (!d_rows || !d_out_list)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (nrows > 0xFFFFFFFFll)
		return fail(NDBHIP_ERR_UNSUPPORTED, "too many rows");
	return assign_rows(d_rows, nrows, dim, d_centroids, ncentroids, true, d_out_list, nullptr);
}

/* ivfinsert (src/index/ivf_am.c:797-1167) for one host row: nearest centroid by the insert-time rule
 * (sqrtf of the fp32 sum, strict <, first minimum: :905-935) on the device, then the entry goes to the tail
 * of that list (ndbhip_ivf_append). */
extern "C" int
ndbhip_ivf_insert(ndbhip_ivf *ix, const float *vec, const uint8_t *tid6, int *list_out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
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

extern "C" int
ndbhip_kmeans_device(const float *d_samples, int n, int dim, int k, int max_iter, float threshold,
					 float *d_centroids, int *d_assign, int *d_counts, int *out_iters, float *out_cost)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!d_samples || !d_centroids || !d_assign || !d_counts || n < 1 || dim < 1 || k < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if ((size_t) n * 4 + 64 > NDB_TOPK_MAX_SMEM)	/* the reference samples at most 10000 rows (ivf_am.c:580) */
		return fail(NDBHIP_ERR_UNSUPPORTED, "k-means sample of %d rows exceeds the LDS-resident limit", n);
	float	   *d_pc = nullptr, *d_cost = nullptr;
	float		prevCost = FLT_MAX, cost = 0.0f;
	int			iters = 0;
	AssignWs	ws;

	HIP_TRY(hipMalloc((void **) &d_pc, (size_t) n * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_cost, sizeof(float)));
	hipLaunchKernelGGL(k_kmeans_init, dim3((unsigned) (((size_t) k * dim + 255) / 256)), dim3(256), 0, g.stream,
					   d_samples, n, dim, k, d_centroids);
	for (int iter = 0; iter < max_iter; iter++)
	{
		int			rc;

		HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t) k * sizeof(int), g.stream));
		rc = assign_rows(d_samples, n, dim, d_centroids, k, false, d_assign, d_counts, &ws);
		if (rc)
			return rc;
		hipLaunchKernelGGL(k_kmeans_update, dim3(k), dim3(256), (size_t) n * 4 + 64, g.stream, d_samples, n, dim,
						   (const int *) d_assign, (const int *) d_counts, d_centroids);
		hipLaunchKernelGGL(k_kmeans_point_cost, dim3((n + 255) / 256), dim3(256), 0, g.stream, d_samples, n, dim,
						   (const int *) d_assign, (const float *) d_centroids, d_pc);
		hipLaunchKernelGGL(k_seq_sum, dim3(1), dim3(256), (size_t) n * 4, g.stream, (const float *) d_pc, n, d_cost);
		HIP_TRY(hipMemcpyAsync(&cost, d_cost, sizeof(float), hipMemcpyDeviceToHost, g.stream));
		HIP_TRY(hipStreamSynchronize(g.stream));
		iters = iter + 1;
		/* fabs(prevCost - cost) < threshold, float difference widened (ivf_am.c:2141) */
		if (fabs((double) (float) (prevCost - cost)) < (double) threshold)
			break;
		prevCost = cost;
	}
	HIP_TRY(hipFree(d_pc));
	HIP_TRY(hipFree(d_cost));
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

extern "C" int ndbhip_ivf_build_device(ndbhip_ivf *ix, const float *d_rows, const uint64_t *d_tids, int64_t nrows,
									   int max_iter, int *out_iters);

/* ivfbuild (src/index/ivf_am.c:501-745) for host rows in heap order: staged H2D, then ndbhip_ivf_build_device */
extern "C" int
ndbhip_ivf_build(ndbhip_ivf *ix, const float *rows, const uint8_t *tids6, int64_t nrows, int max_iter, int *out_iters)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !rows || !tids6 || nrows < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	float	   *d_rows = nullptr;
	uint64_t   *d_tids = nullptr;
	std::vector<uint64_t> t64((size_t) nrows);

	for (int64_t i = 0; i < nrows; i++)
		t64[(size_t) i] = ndb_tid_pack(tids6 + (size_t) i * 6);
	HIP_TRY(hipMalloc((void **) &d_rows, (size_t) nrows * ix->dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_tids, (size_t) nrows * sizeof(uint64_t)));
	HIP_TRY(hipMemcpyAsync(d_rows, rows, (size_t) nrows * ix->dim * sizeof(float), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_tids, t64.data(), (size_t) nrows * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream));
	const int	rc = ndbhip_ivf_build_device(ix, d_rows, d_tids, nrows, max_iter, out_iters);

	(void) hipStreamSynchronize(g.stream);
	(void) hipFree(d_rows);
	(void) hipFree(d_tids);
	return rc;
}

extern "C" int
ndbhip_ivf_build_device(ndbhip_ivf *ix, const float *d_rows, const uint64_t *d_tids, int64_t nrows,
						int max_iter, int *out_iters)
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
	const bool	dbg = getenv("NDBHIP_DEBUG_BUILD") != nullptr;
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
	rc = ndbhip_kmeans_device(d_rows, ns, dim, k, max_iter, 0.001f, d_cent, d_sasg, d_scnt, &iters, nullptr);
	if (rc)
		return rc;
	lap("k-means on the sample");
	HIP_TRY(hipFree(d_sasg));
	HIP_TRY(hipFree(d_scnt));

	/* every row goes to the list ivfinsert would choose (Q5: the reference leaves this to later INSERTs) */
	HIP_TRY(hipMalloc((void **) &d_list, (size_t) nrows * sizeof(int)));
	lap("free + malloc list ids");
	AssignWs	aws;				/* own workspace: assign_rows then returns without waiting for its kernels */

	rc = assign_rows(d_rows, nrows, dim, d_cent, k, true, d_list, nullptr, &aws);
	if (rc)
		return rc;
	/* The packed mirror is allocated while the assignment kernels run: a fresh multi-GB hipMalloc is host-side
	 * work (page-table setup) that took 0.3 ms in one process and 63 ms in the next on the same box — as much
	 * as the rest of the build — and it needs nothing the GPU is busy with. */
	float	   *d_prow = nullptr;
	uint64_t   *d_ptid = nullptr;

	HIP_TRY(hipMalloc((void **) &d_prow, (size_t) nrows * dim * sizeof(float)));
	HIP_TRY(hipMalloc((void **) &d_ptid, (size_t) nrows * sizeof(uint64_t)));
	lap("assign every row (+ malloc of the packed rows under it)");

	const uint32_t nblocks = (uint32_t) ((nrows + NDB_PACK_BLOCK - 1) / NDB_PACK_BLOCK);
	const size_t nh = (size_t) k * nblocks;
	uint32_t   *d_hist = nullptr;
	int64_t    *d_scan = nullptr;

	HIP_TRY(hipMalloc((void **) &d_hist, nh * sizeof(uint32_t)));
	HIP_TRY(hipMalloc((void **) &d_scan, nh * sizeof(int64_t)));
	HIP_TRY(hipMemsetAsync(d_hist, 0, nh * sizeof(uint32_t), g.stream));
	hipLaunchKernelGGL(k_pack_hist, dim3(nblocks), dim3(NDB_PACK_BLOCK), 0, g.stream, (const int *) d_list, nrows,
					   k, nblocks, d_hist);
	std::vector<int64_t> list_len((size_t) k, 0);
	int64_t    *d_llen = nullptr;

	HIP_TRY(hipMalloc((void **) &d_llen, (size_t) k * sizeof(int64_t)));
	hipLaunchKernelGGL(k_pack_list_totals, dim3((k + 63) / 64), dim3(64), 0, g.stream, (const uint32_t *) d_hist, k,
					   nblocks, d_llen);
	hipLaunchKernelGGL(k_pack_offsets, dim3(k), dim3(64), 0, g.stream, (const uint32_t *) d_hist, k, nblocks,
					   (const int64_t *) d_llen, d_scan);
	HIP_TRY(hipMemcpyAsync(list_len.data(), d_llen, (size_t) k * sizeof(int64_t), hipMemcpyDeviceToHost, g.stream));

	lap("histograms / offsets");
	hipLaunchKernelGGL(k_pack_scatter, dim3(nblocks), dim3(NDB_PACK_BLOCK), 0, g.stream, (const int *) d_list,
					   nrows, dim, nblocks, (const int64_t *) d_scan, d_rows, d_tids, d_prow, d_ptid);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(g.stream));
	lap("scatter");
	if (aws.release()) return NDBHIP_ERR_HIP;
	HIP_TRY(hipFree(d_hist));
	HIP_TRY(hipFree(d_scan));
	HIP_TRY(hipFree(d_list));
	HIP_TRY(hipFree(d_llen));
	lap("frees");

	/* adopt: centroids + packed lists become the index */
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
	return NDBHIP_OK;
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
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;

	if (i >= n)
		return;
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
		const dim3	grid((unsigned) ((nel + 255) / 256));

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

static int
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
};

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

extern "C" int
ndbhip_hnsw_load(ndbhip_hnsw *h, uint32_t nblocks, const float *vecs, const int32_t *levels,
				 const int16_t *ncount, const int64_t *nbr_off, const uint32_t *nbrs, const uint8_t *tids6,
				 uint32_t entry_point, int entry_level)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
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
		entry[0] = h->entry_point;
		entry[1] = (uint32_t) h->entry_level;
	}
	HIP_TRY(hipMalloc((void **) &d_lv_in, (size_t) n * sizeof(int)));
	HIP_TRY(hipMalloc((void **) &d_entry, 2 * sizeof(uint32_t)));
	HIP_TRY(hipMemcpyAsync(d_lv_in, levels, (size_t) n * sizeof(int), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_entry, entry, sizeof(entry), hipMemcpyHostToDevice, g.stream));
	{	/* experiment knobs */
		const char *e;

		if ((e = getenv("NDBHIP_HNSW_SPEC")) != nullptr) g_hnsw_spec = atoi(e);
		if ((e = getenv("NDBHIP_HNSW_BATCH_DIV")) != nullptr && atoi(e) > 0) g_hnsw_batch_div = atoi(e);
		if ((e = getenv("NDBHIP_HNSW_BATCH_MAX")) != nullptr && atoi(e) > 0) g_hnsw_batch_max = atoi(e);
	}
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
		const bool	trace = getenv("NDBHIP_HNSW_TRACE") != nullptr;
		/* the chunked commit keeps a list's slots in a 64-bit mask and a chunk's requests in LDS */
		/* commit kernel: 1 = hashed closed-form chunks (m <= 16), else / 3 = sorted-replay chunks (m <= 32),
		 * 2 = one wave, walk by walk */
		const bool	hash_commit = g_hnsw_spec == 1 && ksel <= NDB_HH_MAXSEL && 2 * h->m <= 32;
		const bool	par_commit = !hash_commit && (g_hnsw_spec == 1 || g_hnsw_spec == 3) && ksel <= NDB_HC_MAXSEL &&
			2 * h->m <= 64;
		const bool	fast = (h->dim % 4) == 0 && h->dim <= NDB_HNSW_FAST_MAX_DIM && getenv("NDBHIP_HNSW_NOFAST") == nullptr;
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
	return hnsw_insert_rows(h, d_rows, d_tids, n, levels, ef_construction, 0);
}

/* hnswinsert (src/index/hnsw_am.c:478-538): n more rows on top of the graph the mirror holds */
extern "C" int
ndbhip_hnsw_insert_device(ndbhip_hnsw *h, const float *d_rows, const uint64_t *d_tids, uint32_t n,
						  const int32_t *levels, int ef_construction)
{
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
		HIP_TRY(hipMalloc((void **) &h->d_dead, (size_t) nb));
		HIP_TRY(hipMemsetAsync(h->d_dead, 0, (size_t) nb, g.stream));
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
	if (!h || !h->loaded || !dead)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!h->d_dead)
		HIP_TRY(hipMalloc((void **) &h->d_dead, (size_t) h->nblocks));
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
	g.stats.queries += (uint64_t) nq;
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
	g.stats.queries += (uint64_t) nq;
	return NDBHIP_OK;
}

static int
hnsw_search_host(ndbhip_hnsw *h, bool scan_layer, const float *queries, int nq, int strategy, int ef, int k,
				 uint32_t *out_blocks, float *out_dist, int *out_count, uint8_t *out_tids6, int64_t *out_scored)
{
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
	rc = scan_layer
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
	return hnsw_search_host(h, false, queries, nq, strategy, ef, k, out_blocks, out_dist, out_count, out_tids6,
							out_scored);
}

extern "C" int
ndbhip_hnsw_search_layer(ndbhip_hnsw *h, const float *queries, int nq, int strategy, int ef, int k,
						 uint32_t *out_blocks, float *out_dist, int *out_count, uint8_t *out_tids6,
						 int64_t *out_scored)
{
	return hnsw_search_host(h, true, queries, nq, strategy, ef, k, out_blocks, out_dist, out_count, out_tids6,
							out_scored);
}


/* ================================================================== */
/* Datum -> dense float4[] (ivfExtractVectorData ivf_am.c:117-218,      */
/* hnswExtractVectorData hnsw_am.c:1402-1519).  Host-side staging of    */
/* queries and inserted rows; operates on detoasted datum images.       */
/* ================================================================== */

/* fp16 -> fp32 exactly as the reference's fp16_to_float (quantization.c:170-218), including its
 * subnormal exponent arithmetic (quirk Q20), so halfvec columns index the same values */
static float
fp16_image_to_float(uint16_t h)
{
	const uint32_t sign = (uint32_t) (h & 0x8000u) << 16;
	uint32_t	exp = (h & 0x7c00u) >> 10;
	const uint32_t mant = h & 0x03ffu;
	uint32_t	f;
	float		out;

	if (exp == 0)
	{
		if (mant == 0)
			f = sign;
		else
		{
			uint32_t	m = mant;

			exp = 1;
			while ((m & 0x0400u) == 0)
			{
				m <<= 1;
				exp--;
			}
			m &= 0x03ffu;
			f = sign | ((127u - 15u - (10u - exp)) << 23) | (m << 13);
		}
	}
	else if (exp == 0x1f)
		f = sign | 0x7f800000u | (mant << 13);
	else
		f = sign | ((exp + 127u - 15u) << 23) | (mant << 13);
	memcpy(&out, &f, 4);
	return out;
}

extern "C" int
ndbhip_extract_vector(int kind, const void *datum, size_t datum_len, float *out, int out_cap, int *out_dim)
{
	const uint8_t *p = (const uint8_t *) datum;

	if (!datum || !out_dim)
		return fail(NDBHIP_ERR_INVALID, "ivf: out_dim cannot be NULL");
	switch (kind)
	{
		case NDBHIP_TYPE_VECTOR:	/* Vector {int32 vl_len_; int16 dim; int16 unused; float4 data[]} */
		{
			int16_t		dim;

			if (datum_len < 8)
				return fail(NDBHIP_ERR_INVALID, "vector datum too short");
			memcpy(&dim, p + 4, 2);
			if (dim < 0 || datum_len < 8 + (size_t) dim * 4)
				return fail(NDBHIP_ERR_INVALID, "vector datum truncated");
			*out_dim = dim;
			if (out && out_cap >= dim)
				memcpy(out, p + 8, (size_t) dim * 4);
			else if (out)
				return fail(NDBHIP_ERR_INVALID, "output buffer too small for %d dimensions", (int) dim);
			return NDBHIP_OK;
		}
		case NDBHIP_TYPE_HALFVEC:	/* VectorF16 {int32 vl_len_; int16 dim; int16 data[]} */
		{
			int16_t		dim;

			if (datum_len < 6)
				return fail(NDBHIP_ERR_INVALID, "halfvec datum too short");
			memcpy(&dim, p + 4, 2);
			if (dim < 0 || datum_len < 6 + (size_t) dim * 2)
				return fail(NDBHIP_ERR_INVALID, "halfvec datum truncated");
			*out_dim = dim;
			if (out && out_cap < dim)
				return fail(NDBHIP_ERR_INVALID, "output buffer too small for %d dimensions", (int) dim);
			for (int i = 0; out && i < dim; i++)
			{
				uint16_t	h;

				memcpy(&h, p + 6 + 2 * (size_t) i, 2);
				out[i] = fp16_image_to_float(h);
			}
			return NDBHIP_OK;
		}
		case NDBHIP_TYPE_SPARSEVEC:	/* VectorMap {int32 vl_len_; int32 total_dim; int32 nnz; int32 idx[]; float4 val[]} */
		{
			int32_t		total_dim, nnz;

			if (datum_len < 12)
				return fail(NDBHIP_ERR_INVALID, "sparsevec datum too short");
			memcpy(&total_dim, p + 4, 4);
			memcpy(&nnz, p + 8, 4);
			if (total_dim < 0 || nnz < 0 || datum_len < 12 + (size_t) nnz * 8)
				return fail(NDBHIP_ERR_INVALID, "sparsevec datum truncated");
			*out_dim = total_dim;
			if (out && out_cap < total_dim)
				return fail(NDBHIP_ERR_INVALID, "output buffer too small for %d dimensions", total_dim);
			if (out)
			{
				memset(out, 0, (size_t) total_dim * 4);
				for (int i = 0; i < nnz; i++)
				{
					int32_t		ix;

					memcpy(&ix, p + 12 + 4 * (size_t) i, 4);
					if (ix >= 0 && ix < total_dim)	/* out-of-range indices are dropped (ivf_am.c:187-188) */
						memcpy(&out[ix], p + 12 + 4 * (size_t) nnz + 4 * (size_t) i, 4);
				}
			}
			return NDBHIP_OK;
		}
		case NDBHIP_TYPE_BIT:		/* VarBit {int32 vl_len_; int32 bit_len; bits8 bit_dat[]}: 1 -> +1.0, 0 -> -1.0 */
		{
			int32_t		nbits;

			if (datum_len < 8)
				return fail(NDBHIP_ERR_INVALID, "bit datum too short");
			memcpy(&nbits, p + 4, 4);
			if (nbits < 0 || datum_len < 8 + ((size_t) nbits + 7) / 8)
				return fail(NDBHIP_ERR_INVALID, "bit datum truncated");
			*out_dim = nbits;
			if (out && out_cap < nbits)
				return fail(NDBHIP_ERR_INVALID, "output buffer too small for %d dimensions", nbits);
			for (int i = 0; out && i < nbits; i++)
				out[i] = ((p[8 + i / 8] >> (7 - (i % 8))) & 1) ? 1.0f : -1.0f;
			return NDBHIP_OK;
		}
		default:				/* ereport(ERROR, "ivf: unsupported type OID"): ivf_am.c:208-214 */
			return fail(NDBHIP_ERR_UNSUPPORTED, "ivf: unsupported type kind %d", kind);
	}
}
