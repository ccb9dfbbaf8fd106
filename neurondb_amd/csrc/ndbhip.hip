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



#include "ndbhip_internal.h"
#include <atomic>
#include <thread>

/* ================================================================== */
/* context / errors                                                    */
/* ================================================================== */

thread_local char ndbhip_g_err[512];
thread_local hipStream_t ndbhip_tl_stream = nullptr;
std::mutex	ndbhip_g_mtx;

/* list-scan kernel choice: 0 auto (grouped for batches >= NDB_GROUPED_MIN_NQ queries and dim % 64 == 0),
 * 1 always per-query (k_ivf_scan), 2 always grouped (k_ivf_scan_grouped) */
static int	g_scan_mode = 0;
/* screened L2 scan in auto mode (ndbhip_set_option("screen", 0) turns it off); batches below this many queries keep the
 * exact scan (the two extra passes cost more than they save there) */
static bool g_screen_auto = true;
/* measured crossover on MI355X (tools/small_batch_probe.py, 1M x 768, probes 32; profiles/r04_small_batch.txt): the
 * exact scans cost 0.17 / 0.25 / 0.35 / 0.34 / 0.39 ms for 2 / 4 / 8 / 16 / 24 queries, the matrix-core screen 0.25-0.27 ms
 * for anything from 2 to 32 — it wins from 5 queries up (32 until round 4: the per-batch chain was twice as long) */
#define NDB_SCREEN_MIN_NQ 5
static int	g_build_prepare = 0;	/* "build_prepare": strategy (1 .. 3) a build prepares the index for before it returns, 0 = the first batched scan or ndbhip_ivf_prepare does */
static int	g_screen_min_nq = NDB_SCREEN_MIN_NQ;	/* batches of at least this many queries take the screened path ("screen_min_nq") */
/* measured crossover on MI355X (tools/small_batch_probe.py, 1M x 768, probes 32): the grouped path costs 0.38 ms for 1..16
 * queries, the per-query path 0.18 / 0.24 / 0.35 / 0.50 ms for 1 / 2 / 4 / 7 */
#define NDB_GROUPED_MIN_NQ 5
/* rows staged per step by the grouped kernels: 64 floats (16 KiB tile, 3 waves/SIMD) or 32 (8 KiB, 4 waves/SIMD);
 * ndbhip_set_option("gchunk", 32 | 64) for experiments */
static int	g_gchunk = 32;
/* A/B switches of the fp32 screened path (all results are bit-identical): bound pass 0 one wave per item,
 * 1 a block per 64 rows x 4 query groups, 2 per 128 rows x 4 groups ("scr_coop"); its chunk 16 | 32 floats
 * ("scr_ch"); on fp32 MFMA or the vector ALU ("scr_mfma"); timing prints of the fp16 sweep / the build */
static int	g_scr_coop = 2, g_scr_ch = 16, g_scr_mfma = 1, g_debug_s16 = 0, g_debug_build = 0;

Ctx			g;

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
	HIP_TRY(hipHostMalloc((void **) &g.pin_words, 4096, hipHostMallocDefault));
	memset(g.pin_words, 0, 4096);
	g.device = device;
	g.inited = true;
	return set_kernel_attributes();
}

static void upload_release(void);

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
	if (g.asg_arena)
		(void) hipFree(g.asg_arena);
	if (g.pin_words)
		(void) hipHostFree(g.pin_words);
	big_cache_flush();
	upload_release();
	g = Ctx();
	return NDBHIP_OK;
}

/*
 * Host rows -> device.  Measured on the MI355X box (tools/host_build_probe.py): the runtime's own hipMemcpy from
 * pageable memory carries 3 GB at 56 GB/s — the link's rate — while a pinned-staging copy of this library's own
 * (CPU memcpy into pinned buffers + DMA, 1 .. 8 threads) reached 20 .. 39 GB/s, bounded by the CPU copies, and
 * pinning the caller's pages first (hipHostRegister) + asynchronous pieces was no faster than the runtime.  So
 * the upload is the runtime's, cut into NDB_UP_CHUNK pieces on a side stream by a thread of its own, and what
 * this adds is ORDER: the pieces go in row order and `done` says how far they have got, so that device work on
 * the first rows (the k-means of a build, then the assignment slab by slab) runs while the rest is on the wire.
 */
#define NDB_UP_CHUNK ((size_t) 32 << 20)
static hipStream_t g_up_stream = nullptr;
static std::mutex g_up_mutex;		/* one upload at a time */

static void
upload_release(void)
{
	if (g_up_stream)
		(void) hipStreamDestroy(g_up_stream);
	g_up_stream = nullptr;
}

/* bytes of host memory at src -> device memory at d_dst in chunks; *done (nullable) = bytes that have arrived */
static int
upload_rows(void *d_dst, const void *src, size_t bytes, std::atomic<size_t> *done = nullptr)
{
	std::lock_guard<std::mutex> hold(g_up_mutex);

	if (!g_up_stream && hipStreamCreateWithFlags(&g_up_stream, hipStreamNonBlocking) != hipSuccess)
		return fail(NDBHIP_ERR_HIP, "no stream for the upload");
	for (size_t off = 0; off < bytes; off += NDB_UP_CHUNK)
	{
		const size_t n = std::min(NDB_UP_CHUNK, bytes - off);

		if (hipMemcpyAsync((char *) d_dst + off, (const char *) src + off, n, hipMemcpyHostToDevice, g_up_stream) != hipSuccess ||
			hipStreamSynchronize(g_up_stream) != hipSuccess)
			return fail(NDBHIP_ERR_HIP, "upload of %zu bytes failed: %s", bytes, hipGetErrorString(hipGetLastError()));
		if (done)
			done->store(off + n, std::memory_order_release);
	}
	return 0;
}

/* upload_rows on a thread of its own */
struct UploadJob
{
	std::thread th;
	std::atomic<size_t> done{0};
	std::atomic<int> finished{0};
	size_t		total = 0;
	int			rc = 0;
	bool		running = false;
	void start(void *d_dst, const void *src, size_t bytes)
	{
		int			dev = 0;

		(void) hipGetDevice(&dev);
		total = bytes;
		running = true;
		th = std::thread([=]() {
			(void) hipSetDevice(dev);
			rc = upload_rows(d_dst, src, bytes, &done);
			finished.store(1, std::memory_order_release);
		});
	}
	/* blocks until the first `bytes` bytes have arrived (or the upload has failed) */
	int wait_for(size_t bytes)
	{
		if (bytes > total)
			bytes = total;
		while (running && done.load(std::memory_order_acquire) < bytes && !finished.load(std::memory_order_acquire))
			std::this_thread::sleep_for(std::chrono::microseconds(50));
		return finished.load(std::memory_order_acquire) ? rc : 0;
	}
	int wait()
	{
		if (running)
		{
			th.join();
			running = false;
		}
		return rc;
	}
	~UploadJob() { (void) wait(); }
};

/*
 * Allocation of an index's large device blocks.  hipMalloc of a few GB costs between 0.3 and 60 ms on this runtime
 * (page-table work on the host), as much as a whole 1 M-row build, and hipFree about as much; a REINDEX frees
 * exactly the blocks its successor needs.  Up to NDB_BIG_CACHE_SLOTS freed blocks of >= 64 MiB are therefore kept
 * and handed to the next request of (nearly) the same size; ndbhip_set_option("block_cache", 0) or
 * ndbhip_shutdown() releases them.
 */
#define NDB_BIG_CACHE_SLOTS 4
#define NDB_BIG_CACHE_MIN ((size_t) 64 << 20)

int
big_alloc(void **out, size_t bytes)
{
	std::lock_guard<std::mutex> lk(ndbhip_g_mtx);

	*out = nullptr;
	if (bytes == 0)
		bytes = 16;
	for (size_t i = 0; i < g.big_cached.size(); i++)
	{
		const size_t have = g.big_cached[i].second;

		if (have >= bytes && have - bytes <= bytes / 8)
		{
			*out = g.big_cached[i].first;
			g.big_live.push_back(g.big_cached[i]);
			g.big_cached.erase(g.big_cached.begin() + (long) i);
			return 0;
		}
	}
	if (hipMalloc(out, bytes) != hipSuccess)
	{
		(void) hipGetLastError();
		for (auto &b : g.big_cached)		/* the cache must never be the reason an allocation fails */
			(void) hipFree(b.first);
		g.big_cached.clear();
		HIP_TRY(hipMalloc(out, bytes));
	}
	if (bytes >= NDB_BIG_CACHE_MIN)
		g.big_live.push_back(std::make_pair(*out, bytes));
	return 0;
}

void
big_free(void *p)
{
	if (!p)
		return;
	std::lock_guard<std::mutex> lk(ndbhip_g_mtx);

	for (size_t i = 0; i < g.big_live.size(); i++)
		if (g.big_live[i].first == p)
		{
			const std::pair<void *, size_t> b = g.big_live[i];

			g.big_live.erase(g.big_live.begin() + (long) i);
			if (!g.big_cache_on)
				break;
			if (g.big_cached.size() >= NDB_BIG_CACHE_SLOTS)
			{
				(void) hipFree(g.big_cached.front().first);
				g.big_cached.erase(g.big_cached.begin());
			}
			g.big_cached.push_back(b);
			return;
		}
	(void) hipFree(p);
}

void
big_cache_flush(void)
{
	std::lock_guard<std::mutex> lk(ndbhip_g_mtx);

	for (auto &b : g.big_cached)
		(void) hipFree(b.first);
	g.big_cached.clear();
}

extern "C" int
ndbhip_set_stream(void *s)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	g.stream = s ? (hipStream_t) s : g.own_stream;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_set_thread_stream(void *s)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	ndbhip_tl_stream = (hipStream_t) s;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_get_stream(void **out_hip_stream)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!out_hip_stream)
		return fail(NDBHIP_ERR_INVALID, "out is NULL");
	*out_hip_stream = (void *) (hipStream_t) g.stream;
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
	std::lock_guard<std::mutex> lk(ndbhip_g_mtx);

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
		g.stats.pairs_pruned = c[5];
		g.stats.rows_swept = c[6];
		g.stats.plane_bytes = c[7];
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
			   uint32_t *__restrict__ loc_cand_off, unsigned long long *__restrict__ counters, int full_sort = 1,
			   const uint8_t *__restrict__ only = nullptr /* serve just the queries marked here (k_cent_select's leftovers) */,
			   int sum_here = 0 /* add this query's candidate counts to counters[0..2] (dim = bytes per row) */ )
{
	__shared__ uint32_t hist[256];
	__shared__ uint32_t sh[16];
	__shared__ uint64_t comp[NDBHIP_MAX_NPROBE + 2];
	__shared__ uint64_t full[2048];
	__shared__ uint64_t s_bound;
	__shared__ uint32_t perm[NDBHIP_MAX_NPROBE];
	__shared__ uint32_t lens[NDBHIP_MAX_NPROBE];
	__shared__ int selc[NDBHIP_MAX_NPROBE];
	const uint32_t q = blockIdx.x;
	const uint32_t tid = threadIdx.x;
	const float *d = cdist + (size_t) q * cstride;
	int			npr_eff = npr < ncmp ? npr : ncmp;
	uint32_t	T, m_less, kk, cnt_eq;

	if (only && !only[q])
		return;					/* uniform */
	if (npr_eff < 0)
		npr_eff = 0;
	/* valid = strictly below FLT_MAX (bestDist starts at FLT_MAX: ivf_am.c:1689, 1706) */
	auto		ld = [&](uint32_t i, uint32_t &bits) -> bool {
		const float v = d[i];

		bits = __float_as_uint(v);
		return v < FLT_MAX;
	};

	if (ncmp <= 2048 && full_sort)
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
		uint32_t	nok = 0;

		for (uint32_t j0 = 0; j0 < np2; j0 += blockDim.x)
		{
			const uint32_t j = j0 + tid;
			uint32_t	bits = 0;
			const bool	ok = j < (uint32_t) ncmp && ld(j, bits);

			if (j < np2)
				full[j] = ok ? (((uint64_t) ndb_key_from_bits(bits) << 32) | j) : ~0ull;
			nok += (uint32_t) __popcll(__ballot(ok));	/* (one LDS atomic per key on one word serialises: 1024 of them cost ~10 us) */
		}
		if ((tid & 63u) == 0 && nok)
			atomicAdd(&sh[0], nok);
		__syncthreads();
		/*
		 * The head of that order without the sort (a single query spent 30 of its 176 us in the 55 barrier-separated
		 * stages of the 1024-key network): the kk-th smallest of the threads' minima over their strided shares bounds the
		 * kk-th smallest key (the kk smallest minima are kk distinct keys), at most kk threads hold keys at or below it,
		 * so at most kk x (keys per thread) keys are gathered — and both "the kk-th smallest of the minima" and "the
		 * order of the gathered" are ranks by counting, a loop of broadcast LDS reads with no barrier inside.
		 */
		const uint32_t per = (np2 + blockDim.x - 1) / blockDim.x;
		const uint32_t want = min((uint32_t) npr_eff, sh[0]);
		uint64_t	mymin = ~0ull;

		for (uint32_t j = tid; j < np2; j += blockDim.x)
			mymin = min(mymin, full[j]);
		const uint32_t nth = (uint32_t) __syncthreads_count(mymin != ~0ull);
		const bool	by_rank = want > 0 && nth >= want && want * per <= 512u && blockDim.x <= NDBHIP_MAX_NPROBE;

		if (by_rank)
		{
			comp[tid] = mymin;
			if (tid == 0)
				sh[1] = 0;
			__syncthreads();
			const uint32_t r = lds_rank_u64(comp, blockDim.x, mymin);

			if (r == want - 1)
				s_bound = mymin;
			__syncthreads();
			const uint64_t U = s_bound;

			/* (every thread finished reading comp[] before that barrier: it can be reused) */
			for (uint32_t j = tid; j < np2; j += blockDim.x)
			{
				const uint64_t v = full[j];

				if (v <= U && v != ~0ull)
					comp[atomicAdd(&sh[1], 1u)] = v;
			}
			__syncthreads();
			const uint32_t m = sh[1];

			for (uint32_t c = tid; c < m; c += blockDim.x)
			{
				const uint64_t v = comp[c];
				const uint32_t r = lds_rank_u64(comp, m, v);

				if (r < want)
					perm[r] = (uint32_t) v;
			}
			__syncthreads();
			kk = want;
		}
		else
		{
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
	/* the selection is done with full[] and comp[]: the probes' own-row ranges and the running sums live there now */
	uint32_t   *olo = (uint32_t *) full, *olen = olo + NDBHIP_MAX_NPROBE;
	uint32_t   *pa = (uint32_t *) comp, *pm = pa + (NDBHIP_MAX_NPROBE + 1);

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
		const bool	held = c >= 0 && c < ncent;

		lens[i] = held ? glob_len[c] : 0u;	/* ivf_am.c:1768-1779 */
		olo[i] = held ? own_lo[c] : 0u;		/* (read here, by all threads: thread 0's running sums below then touch LDS only —
											 * 32 dependent pairs of global loads were 15 of a single query's 176 us) */
		olen[i] = held ? own_len[c] : 0u;
	}
	__syncthreads();
	if (tid == 0)
	{
		uint64_t	acc = 0, mine = 0;

		pa[0] = 0;
		pm[0] = 0;
		for (int i = 0; i < npr; i++)
		{
			uint64_t	l = lens[i];

			if (cap > 0 && acc + l > cap)
				l = cap - acc;	/* candidateCount < maxCandidates guards: ivf_am.c:1764, 1793, 1811 */
			acc += l;
			if (l > 0)
			{
				/* of the first l positions of the list, how many does this mirror hold (ndb_local_part) */
				const uint64_t lo = olo[i], hi = lo + olen[i];

				mine += l > lo ? ((l < hi ? l : hi) - lo) : 0;
			}
			pa[i + 1] = (uint32_t) acc;
			pm[i + 1] = (uint32_t) mine;
		}
		/* (batches: summed by k_sum_candidates — three atomics per query on one line serialise; a handful of queries
		 * add their own and save the launch) */
		if (counters && sum_here)
		{
			const uint64_t m = loc_cand_off ? mine : acc;

			atomicAdd(&counters[0], (unsigned long long) acc);
			atomicAdd(&counters[1], (unsigned long long) m);
			atomicAdd(&counters[2], (unsigned long long) m * (unsigned long long) dim);
		}
	}
	__syncthreads();
	{
		uint32_t   *co = cand_off + (size_t) q * (npr + 1);
		uint32_t   *lco = loc_cand_off ? loc_cand_off + (size_t) q * (npr + 1) : nullptr;

		for (uint32_t i = tid; i <= (uint32_t) npr; i += blockDim.x)
		{
			co[i] = pa[i];
			if (lco)
				lco[i] = pm[i];
		}
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

	for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < nq; q += gridDim.x * 256)
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
			 uint32_t *__restrict__ cnt, const unsigned int *__restrict__ active = nullptr,
			 const uint8_t *__restrict__ drop = nullptr /* [nq][npr]: pairs a bound has already excluded */ )
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;

	if (i >= nq * (uint32_t) npr)
		return;
	const uint32_t q = i / npr, p = i % npr;
	const uint32_t *co = loc_cand_off + (size_t) q * (npr + 1);	/* rows held HERE for this (query, probe) */

	if (co[p + 1] == co[p] || (active && !active[q]) || (drop && drop[i]))
		return;
	atomicAdd(&cnt[probes[(size_t) q * npr + p]], 1u);
}

/* k_sub_pairs' per-XCD counts summed per list and turned into each XCD's start inside the list's run of pairs — what
 * k_pair_offsets does itself for a few thousand lists; with the tens of thousands of buckets of a 10 M-row index that is
 * megabytes through ONE compute unit (154 us of a 1.4 ms step at C5), so there it runs here, over the whole device */
__global__ __launch_bounds__(256) void
k_pair_xsum(uint32_t *__restrict__ cntx, uint32_t xstride, int ncent, uint32_t *__restrict__ cnt)
{
	const int	L = blockIdx.x * 256 + threadIdx.x;

	if (L >= ncent)
		return;
	uint32_t	v[8], run = 0;

#pragma unroll
	for (int x = 0; x < 8; x++)
		v[x] = cntx[(size_t) x * xstride + L];
#pragma unroll
	for (int x = 0; x < 8; x++)
	{
		cntx[(size_t) x * xstride + L] = run;
		run += v[x];
	}
	cnt[L] = run;
}

/* pass 2 (one block of 1024 threads): per-list pair / work-item / group offsets */
__global__ __launch_bounds__(1024) void
k_pair_offsets(uint32_t *__restrict__ cnt, const uint32_t *__restrict__ glob_len, int ncent,
			   uint32_t *__restrict__ pair_off, uint32_t *__restrict__ item_off,
			   uint32_t *__restrict__ grp_off, uint32_t *__restrict__ runs, uint32_t gdiv, uint32_t rt /* rows per item / 32 */,
			   int exact_runs = 0 /* runs of exactly nitems / 8 items (a list may straddle two runs): for a sweep whose blocks
								   * walk their XCD's run at a fixed stride and cannot help another run out */,
			   unsigned long long *__restrict__ swept = nullptr /* statistics: += sum of pairs x rows over the lists */,
			   uint32_t *__restrict__ cntx = nullptr /* [8][xstride] counts per XCD (k_sub_pairs): summed into cnt[] here and
													  * replaced by each XCD's start inside the list's run of pairs */,
			   uint32_t xstride = 0,
			   int wmode = 0 /* 1: grp_off[] = the list's first (pair, 32-row block) word instead — pairs before it x their
							  * lists' blocks: where the register-streaming sweep leaves a pair's row masks and bounds
							  * (ndbhip_screen16w.h) */ )
{
	__shared__ uint32_t sa[1024], sb[1024], sc[1024];
	const int	t = threadIdx.x;
	const int	per = (ncent + 1023) / 1024;
	const int	l0 = t * per, l1 = min(ncent, l0 + per);
	uint32_t	a = 0, b = 0, c2 = 0;
	unsigned long long sw = 0;

	if (cntx)
	{
		/* (lists dealt round the threads here, so that the eight rows of counters are read in full lines) */
		for (int L = t; L < ncent; L += 1024)
		{
			uint32_t	v[8], run = 0;

#pragma unroll
			for (int x = 0; x < 8; x++)
				v[x] = cntx[(size_t) x * xstride + L];
#pragma unroll
			for (int x = 0; x < 8; x++)
			{
				cntx[(size_t) x * xstride + L] = run;
				run += v[x];
			}
			cnt[L] = run;
		}
		__threadfence_block();
		__syncthreads();
	}

	for (int L = l0; L < l1; L++)
	{
		const uint32_t c = cnt[L];
		const uint32_t ng = (c + NDB_QG - 1) / NDB_QG;

		sw += (unsigned long long) c * glob_len[L];
		a += c;
		b += ((((glob_len[L] + 31u) >> 5) + rt - 1u) / rt) * ((ng + gdiv - 1u) / gdiv);
		c2 += wmode ? c * ((glob_len[L] + 31u) >> 5) : ng;
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
		b += ((((glob_len[L] + 31u) >> 5) + rt - 1u) / rt) * ((ng + gdiv - 1u) / gdiv);
		c2 += wmode ? c * ((glob_len[L] + 31u) >> 5) : ng;
	}
	if (swept)
	{
		/* (one wave-level sum, then 16 atomics) */
		for (int off = 32; off > 0; off >>= 1)
		{
			const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) sw, off, 64);
			const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (sw >> 32), off, 64);

			sw += ((unsigned long long) hi << 32) | lo;
		}
		if ((t & 63) == 0 && sw != 0)
			atomicAdd(swept, sw);
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

		if (t > 0 && t < 8 && exact_runs)
			r = (uint32_t) (((uint64_t) nitems * (uint32_t) t) >> 3);
		else if (t > 0 && t < 8)
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

/*
 * The same for the tens of thousands of buckets a 10 M-row index has (one block walking them all was 130 us of a 1.4 ms
 * step at C5: one compute unit's address path): two launches over ncent / 1024 blocks, a list per thread.
 * k_pair_part leaves every block's three totals; k_pair_scan adds up the totals of the blocks before its own, scans its
 * 1024 lists and writes their offsets; the last block writes the grand totals and the eight runs (exact runs only: the
 * centred sweep's static schedule).
 */
__device__ __forceinline__ void
pair_triple(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ glob_len, int L, int ncent, uint32_t gdiv, uint32_t rt,
			uint32_t &a, uint32_t &b, uint32_t &c2, int wmode)
{
	a = b = c2 = 0;
	if (L < ncent)
	{
		const uint32_t c = cnt[L];
		const uint32_t ng = (c + NDB_QG - 1) / NDB_QG;

		a = c;
		b = ((((glob_len[L] + 31u) >> 5) + rt - 1u) / rt) * ((ng + gdiv - 1u) / gdiv);
		c2 = wmode ? c * ((glob_len[L] + 31u) >> 5) : ng;
	}
}

__global__ __launch_bounds__(1024) void
k_pair_part(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ glob_len, int ncent, uint32_t gdiv, uint32_t rt,
			uint32_t *__restrict__ part /* [blocks][4] */, unsigned long long *__restrict__ swept, int wmode = 0)
{
	__shared__ uint32_t sa[16], sb[16], sc[16];
	const int	t = threadIdx.x, L = blockIdx.x * 1024 + t;
	uint32_t	a, b, c2;
	unsigned long long sw = 0;

	pair_triple(cnt, glob_len, L, ncent, gdiv, rt, a, b, c2, wmode);
	if (swept && L < ncent)
		sw = (unsigned long long) a * glob_len[L];
	for (int off = 32; off > 0; off >>= 1)
	{
		a += (uint32_t) __shfl_xor((int) a, off, 64);
		b += (uint32_t) __shfl_xor((int) b, off, 64);
		c2 += (uint32_t) __shfl_xor((int) c2, off, 64);
		if (swept)
		{
			const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) sw, off, 64);
			const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (sw >> 32), off, 64);

			sw += ((unsigned long long) hi << 32) | lo;
		}
	}
	if ((t & 63) == 0)
	{
		sa[t >> 6] = a;
		sb[t >> 6] = b;
		sc[t >> 6] = c2;
		if (swept && sw != 0)
			atomicAdd(swept, sw);
	}
	__syncthreads();
	if (t == 0)
	{
		uint32_t	A = 0, B = 0, C = 0;

		for (int w = 0; w < 16; w++)
		{
			A += sa[w];
			B += sb[w];
			C += sc[w];
		}
		part[4 * blockIdx.x] = A;
		part[4 * blockIdx.x + 1] = B;
		part[4 * blockIdx.x + 2] = C;
	}
}

__global__ __launch_bounds__(1024) void
k_pair_scan(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ glob_len, int ncent, uint32_t gdiv, uint32_t rt,
			const uint32_t *__restrict__ part, uint32_t *__restrict__ pair_off, uint32_t *__restrict__ item_off,
			uint32_t *__restrict__ grp_off, uint32_t *__restrict__ runs, int wmode = 0)
{
	__shared__ uint32_t sa[1024], sb[1024], sc[1024], base[3];
	const int	t = threadIdx.x, L = blockIdx.x * 1024 + t;
	uint32_t	a, b, c2;

	/* the totals of the blocks before this one (a few dozen) */
	{
		uint32_t	pa = 0, pb = 0, pc = 0;

		for (int j = t; j < (int) blockIdx.x; j += 1024)
		{
			pa += part[4 * j];
			pb += part[4 * j + 1];
			pc += part[4 * j + 2];
		}
		sa[t] = pa;
		sb[t] = pb;
		sc[t] = pc;
		__syncthreads();
		if (t == 0)
		{
			uint32_t	A = 0, B = 0, C = 0;
			const int	nn = min((int) blockIdx.x, 1024);

			for (int j = 0; j < nn; j++)
			{
				A += sa[j];
				B += sb[j];
				C += sc[j];
			}
			base[0] = A;
			base[1] = B;
			base[2] = C;
		}
		__syncthreads();
	}
	pair_triple(cnt, glob_len, L, ncent, gdiv, rt, a, b, c2, wmode);
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
	if (L < ncent)
	{
		pair_off[L] = base[0] + sa[t] - a;
		item_off[L] = base[1] + sb[t] - b;
		grp_off[L] = base[2] + sc[t] - c2;
	}
	if (blockIdx.x == gridDim.x - 1 && t == 1023)
	{
		const uint32_t nitems = base[1] + sb[1023];

		pair_off[ncent] = base[0] + sa[1023];
		item_off[ncent] = nitems;
		grp_off[ncent] = base[2] + sc[1023];
		for (uint32_t x = 0; x <= 8; x++)
			runs[x] = x == 0 ? 0u : (x == 8 ? nitems : (uint32_t) (((uint64_t) nitems * x) >> 3));
	}
}

/* pass 3: bucket the pairs */
__global__ void
k_pair_fill(const int *__restrict__ probes, const uint32_t *__restrict__ loc_cand_off, int npr, uint32_t nq,
			const uint32_t *__restrict__ pair_off,
			uint32_t *__restrict__ fill, PairRec *__restrict__ pairs, const unsigned int *__restrict__ active = nullptr,
			const uint8_t *__restrict__ drop = nullptr)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;

	if (i >= nq * (uint32_t) npr)
		return;
	const uint32_t q = i / npr, p = i % npr;
	const uint32_t *co = loc_cand_off + (size_t) q * (npr + 1);

	if (co[p + 1] == co[p] || (active && !active[q]) || (drop && drop[i]))
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

#include "ndbhip_screen32.h"

#include "ndbhip_topk.h"

/* ================================================================== */
/* host side: IVF mirror                                               */
/* ================================================================== */

/* see s16mat_prepare / s16mat_run (ndbhip_build.h) */
struct S16Mat
{
	unsigned char *planes = nullptr; size_t planes_n = 0;
	float	   *rn2 = nullptr;		size_t rn2_n = 0;
	int16_t    *rexp = nullptr;		size_t rexp_n = 0;
	uint32_t   *xmax = nullptr;		size_t xmax_n = 0;		/* [0] largest norm (float bits), [2..3] block offsets */
	int64_t    *loc = nullptr;		size_t loc_n = 0;		/* {0, n} */
	uint32_t   *meta = nullptr;		size_t meta_n = 0;		/* the sweep's tables for (n vectors x nq queries) */
	PairRec    *pairs = nullptr;	size_t pairs_n = 0;
	uint4	   *desc = nullptr;		size_t desc_n = 0;		/* S16Desc[] */
	unsigned int *heads = nullptr;	size_t heads_n = 0;
	unsigned char *zero = nullptr;	size_t zero_n = 0;
	int			nq = -1;			/* batch size the tables were built for */
	int			n = 0;
	const float *src = nullptr;
	void release()
	{
		void	   *ptrs[] = {planes, rn2, rexp, xmax, loc, meta, pairs, desc, heads, zero};

		for (void *p : ptrs)
			if (p) (void) hipFree(p);
	}
};

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
	/* the centred one-plane sweep (ndbhip_screen16c.h): upper bounds of the emitted records, the batch's (query,
	 * bucket) pair planes / norms / exponents, and what the last batch looked like (pairs per bucket with pairs) */
	float	   *w_eub = nullptr;	size_t w_eub_n = 0;
	_Float16   *w_qcplanes = nullptr; size_t w_qcplanes_n = 0;
	float	   *w_qcn2 = nullptr;	size_t w_qcn2_n = 0;
	int		   *w_qcexp = nullptr;	size_t w_qcexp_n = 0;
	/* ... its planes take appends in place: rows / capacity of every bucket (list or sublist), the bucket of every list
	 * that takes its appends, first padded plane row (32 x first block) of every bucket, a row's index in its list */
	std::vector<uint32_t> s16_blen, s16_bcap;
	std::vector<int> s16_tail;
	std::vector<int64_t> s16_prow;
	int64_t    *d_prow_off = nullptr;	size_t d_prow_off_n = 0;
	uint32_t   *d_pposof = nullptr;	size_t d_pposof_n = 0;
	bool		s16_cen_layout = false;
	uint32_t   *w_pslot = nullptr;	size_t w_pslot_n = 0;	/* [3][qc_cap] per pair slot: query, first candidate position, visible rows */
	int64_t    *w_qoffs = nullptr;	size_t w_qoffs_n = 0;	/* [nq] byte offsets of scattered queries (ndbhip_ivf_search_mapped) */
	float	   *w_amat = nullptr;	size_t w_amat_n = 0;	/* [nq][astride] the sweep's |q - centroid|^2 (k_cent_select) */
	uint8_t    *w_cfull = nullptr;	size_t w_cfull_n = 0;	/* [nq] queries k_cent_select left to k_probe_select */
	int			qc_mult = 4;			/* rows of the pair planes / (queries x probes) */
	float		s16c_density = -1.0f;	/* pairs per bucket that had any, previous batch (-1: none yet) */
	/* sublists (ndbhip_screen16.h): the planes' own grouping of the rows of long lists */
	bool		s16_sub = false;
	int			s16_sub_cfg = -1;			/* sublist settings the planes were laid out under */
	int			nsub = 0, nsub_g = 0;	/* sublists in all / centres of regrouped lists */
	uint32_t   *d_sub_first = nullptr;	size_t d_sub_first_n = 0;	/* [ncent + 1] first sublist of every list */
	uint32_t   *d_sub_len = nullptr;	size_t d_sub_len_n = 0;		/* [nsub] */
	int64_t    *d_sub_loc = nullptr;	size_t d_sub_loc_n = 0;		/* [nsub + 1] first plane row */
	uint32_t   *d_sub_blk = nullptr;	size_t d_sub_blk_n = 0;		/* [nsub + 1] first 32-row block */
	uint32_t   *d_sub_rad = nullptr;	size_t d_sub_rad_n = 0;		/* [nsub] radius (float bits, rounded up) */
	int		   *d_sub_gidx = nullptr;	size_t d_sub_gidx_n = 0;	/* [nsub] column of its centre in w_subdist, -1: the list's own centroid */
	float	   *d_subcent = nullptr;	size_t d_subcent_n = 0;		/* [nsub_g][dim] */
	const float **d_sub_cptr = nullptr;	size_t d_sub_cptr_n = 0;	/* [nsub] centre of every sublist */
	int64_t    *d_perm = nullptr;		size_t d_perm_n = 0;		/* [nrows] plane row -> mirror row */
	uint32_t   *d_posof = nullptr;		size_t d_posof_n = 0;		/* [nrows] plane row -> index in its list */
	float	   *w_subdist = nullptr;	size_t w_subdist_n = 0;		/* [nq][sstride] squared distances to the centres (sweep MODE 3) */
	/* round 6, the centres of PROBED lists only (ivf_s16_sub_distances_probed): the batch's (query, probe) pairs by list */
	uint32_t   *w_lqoff = nullptr;	size_t w_lqoff_n = 0;		/* [2 (nlists + 1)] first pair of every list | fill cursors */
	uint32_t   *w_lq = nullptr;		size_t w_lq_n = 0;			/* [nq npr] the queries, list by list */
	uint32_t   *w_uoff = nullptr;	size_t w_uoff_n = 0;		/* [nunits + 2] first work item of every unit (a unit's queries 256 at a time) | the work counter */
	/* ... and, per layout: the regrouped lists' sublists sixteen (dim > 1024: eight) at a time — a UNIT (list, first sublist) */
	uint2	   *d_sub_units = nullptr;	size_t d_sub_units_n = 0;
	uint32_t	nsub_units = 0, sub_unit_c = 16;
	S16Mat		dm_sub;				/* the centres of the regrouped lists: every query's squared distance to every one of them */
	S16Mat		dm_cent;			/* the index's centroids: the batch centroid scan on the matrix cores (ndbhip_screen16.h) */
	bool		dm_cent_valid = false;
	/* centroids AND the regrouped lists' centres as one matrix (columns 0 .. ncmp - 1, then the centres): a screened
	 * batch that needs both gets them from one launch instead of two of the same fixed latency */
	S16Mat		dm_all;
	/* inner product on the centred sweep: M^2 (device word + host copy), M^2 - |x|^2 per padded plane row, ev per query */
	bool		ipc_valid = false;
	float		ipc_m2 = 0.0f;
	uint32_t   *d_ipc_m2 = nullptr; size_t d_ipc_m2_n = 0;
	float	   *d_rnx = nullptr; size_t d_rnx_n = 0;
	float	   *d_bkt_rnxmax = nullptr; size_t d_bkt_rnxmax_n = 0;	/* [buckets] the largest of them per bucket (k > 64: k_s16c_thr_radius) */
	float	   *w_qev = nullptr; size_t w_qev_n = 0;
	uint32_t   *w_ppart = nullptr; size_t w_ppart_n = 0;	/* k_pair_part's block totals */
	/* a sample of the mirror's rows as a matrix of their own: first thresholds of a dense batch (k_s16c_seed_sample) */
	S16Mat		dm_seed;
	bool		seed_valid = false;
	int			seed_n = 0;
	float	   *d_seedrows = nullptr; size_t d_seedrows_n = 0;
	int		   *d_seed_list = nullptr; size_t d_seed_list_n = 0;
	uint32_t   *d_seed_pos = nullptr; size_t d_seed_pos_n = 0;
	float	   *w_seedmat = nullptr; size_t w_seedmat_n = 0;
	float	   *d_allcent = nullptr;	size_t d_allcent_n = 0;
	bool		dm_all_valid = false;
	int			dm_all_ncmp = 0;
	const float *bat_subdist = nullptr;	/* this batch's distances to the centres, when dm_all served (else NULL) */
	uint32_t	bat_sstride = 0;
	bool		bat_restrict = false;	/* this batch: the centroid scan multiplied the centroids alone; the regrouped lists' centres are
									 * scored for the probed lists only (ivf_s16_sub_distances_probed) */
	float	   *w_pdist = nullptr;		size_t w_pdist_n = 0;		/* [nq][npr] |q - centroid of the probed list| */
	uint32_t   *d_lrad = nullptr;	size_t d_lrad_n = 0;	/* [ncent] list radius around its centroid (float bits, rounded up) */
	float	   *d_cn2 = nullptr;	size_t d_cn2_n = 0;		/* [ncent] |centroid|^2 (the inner product's sublist bound) */
	/* centred planes: rows of every bucket IN THE PLANES (holes of deleted rows included: the list's own length shrinks,
	 * the bucket's does not) when the buckets are the lists themselves (regrouped planes keep that in d_sub_len), and the
	 * list every bucket belongs to */
	uint4	   *w_qpairs = nullptr;	size_t w_qpairs_n = 0;	/* k_sub_pairs: what the count pass kept, per query */
	uint32_t   *w_qpn = nullptr;	size_t w_qpn_n = 0;
	/* the register-streaming sweep (ndbhip_screen16w.h): per query the slots of its pairs (+ how many), per (pair, 32-row
	 * block) word the rows that stay, per row of a word its two bounds */
	uint32_t   *w_qslot = nullptr;	size_t w_qslot_n = 0;
	uint32_t   *w_wmask = nullptr;	size_t w_wmask_n = 0;
	float2	   *w_wrec = nullptr;	size_t w_wrec_n = 0;
	bool		s16w_off = false;	/* a query of some batch had more pairs than its list holds: the LDS ring from then on */
	uint32_t	s16w_maxlw = 1;		/* the most (pair, block) words a (query, probe) pair can need: blocks of the fullest list's buckets */
	uint32_t	s16w_avgblk = 4;	/* 32-row blocks of a bucket on average (rounded up) */
	uint32_t	wc_mult = 4;		/* buckets per (query, probe) pair the word arrays are sized for to start with; x 4 after a batch that did not fit */
	uint32_t	wc_words = 0;		/* the most words a batch of this mirror has needed so far (read back with the batch's flags) */
	uint32_t   *w_overq = nullptr;	size_t w_overq_n = 0;	/* queries of the last batch whose records / survivors overflowed */
	std::vector<uint32_t> redo;		/* ... on the host: ivf_s16_run returned 2, these go to the exact path one sub-batch */
	float	   *w_redo_q = nullptr;	size_t w_redo_q_n = 0;
	int		   *w_redo_p = nullptr;	size_t w_redo_p_n = 0;
	unsigned char *w_redo_out = nullptr;	size_t w_redo_out_n = 0;
	int64_t    *w_redo_idx = nullptr;	size_t w_redo_idx_n = 0;
	bool		s16_planes_f32 = true;	/* the planes hold two-plane float4-style rows (always, except an fp16 mirror's own plane) */
	float	   *w_qhat = nullptr;	size_t w_qhat_n = 0;	/* cosine: the batch's queries divided by their norms */
	float	   *d_cent_hat = nullptr;	size_t d_cent_hat_n = 0;	/* cosine: the centroids divided by their norms (bucket centres of lists that are their own sublist) */
	bool		s16_cos_layout = false;		/* the planes are the normalised rows' */
	/* cosine: planes / norms / exponents of the queries AS THEY ARE, for the centroid scan (always L2 on the rows' own space) */
	unsigned char *w_qplanes_o = nullptr;	size_t w_qplanes_o_n = 0;
	float	   *w_qn2_o = nullptr;	size_t w_qn2_o_n = 0;
	int		   *w_qexp_o = nullptr;	size_t w_qexp_o_n = 0;
	uint32_t   *d_plen = nullptr;	size_t d_plen_n = 0;
	uint32_t   *d_bucket_list = nullptr;	size_t d_bucket_list_n = 0;
	uint8_t    *w_drop = nullptr;	size_t w_drop_n = 0;	/* [nq][npr] pairs excluded before the sweep */
	uint32_t   *w_s16desc = nullptr; size_t w_s16desc_n = 0;	/* S16Desc per work item of the sweep */
	/* ndbhip_ivf_share: a second handle on the same rows, planes and tables with scratch of its own (two batches in flight on
	 * two streams without a second copy of the mirror).  shared_of: this handle borrows everything persistent from that one;
	 * nshares: handles borrowing from this one.  Either way the persistent state is frozen (ivf_frozen). */
	ndbhip_ivf *shared_of = nullptr;
	int			nshares = 0;
	bool		s16_bigk_off = false;	/* this mirror's layout cannot serve k > 64 on the fp16 screen (no sublists, not centred): found out once */
	uint32_t   *w_bmin = nullptr;	size_t w_bmin_n = 0;	/* [nq][S16_NB] smallest emitted a per hash bucket of positions */
	/* split top-k of small batches: per-range records, counts, totals */
	ndbhip_cand *w_scand = nullptr;	size_t w_scand_n = 0;
	int		   *w_sncand = nullptr;	size_t w_sncand_n = 0;
	int64_t    *w_stotal = nullptr;	size_t w_stotal_n = 0;
};


/* every per-batch scratch array of a handle (pointer + element count `_n`): what ndbhip_ivf_share gives a new handle afresh
 * and the only device memory such a handle frees.  (w_rnorm is not among them: the rows' norms, made once while norm_valid
 * is false, are read by every later batch of the fp32 screen — persistent whatever its name says.) */
#define NDB_IVF_SCRATCH(F) \
	F(w_cdist) F(w_probes) F(w_candoff) F(w_dist) F(w_q) F(w_otid) F(w_odist) F(w_ocnt) \
	F(w_gcnt) F(w_goff) F(w_pairs) F(w_qblock) F(w_qnorm) F(w_scrt) F(w_scrd) \
	F(w_scrc) F(w_screc) F(w_cblock) F(w_tmin) F(w_qplanes) F(w_qn2) F(w_qexp) F(w_qthr) \
	F(w_ecount) F(w_erec) F(w_eub) F(w_qcplanes) F(w_qcn2) F(w_qcexp) F(w_pslot) F(w_qoffs) \
	F(w_amat) F(w_cfull) F(w_subdist) F(w_qev) F(w_ppart) F(w_seedmat) F(w_pdist) F(w_qpairs) \
	F(w_qpn) F(w_qslot) F(w_wmask) F(w_wrec) F(w_overq) F(w_redo_q) F(w_redo_p) F(w_redo_out) \
	F(w_redo_idx) F(w_qhat) F(w_qplanes_o) F(w_qn2_o) F(w_qexp_o) F(w_drop) F(w_s16desc) F(w_bmin) \
	F(w_scand) F(w_sncand) F(w_stotal) F(w_lqoff) F(w_lq) F(w_uoff)

/* a handle whose persistent state other handles read, or which reads another's: nothing may change or be built in it */
static inline bool
ivf_frozen(const ndbhip_ivf *ix)
{
	return ix->shared_of != nullptr || ix->nshares > 0;
}
#define IVF_NOT_FROZEN(ix, what)                                                                                        \
	do {                                                                                                                \
		if (ivf_frozen(ix))                                                                                             \
			return fail(NDBHIP_ERR_STATE, "%s: the mirror is shared (ndbhip_ivf_share): destroy the shares first%s", what, \
						(ix)->shared_of ? ", and do this on the handle they were made from" : "");                       \
	} while (0)

struct ndbhip_ivf;
static int	ivf_s16c_append(ndbhip_ivf *ix, const std::vector<int64_t> &add, const std::vector<int64_t> &new_own);

/* temporaries of one call: freed on every way out of the scope unless keep() hands one over (the
 * HIP_TRY early returns of ivf_flush / ndbhip_ivf_delete leaked their device buffers) */
struct DevGuard
{
	std::vector<void *> owned;	/* the blocks themselves, not the addresses of the caller's variables: a variable
								 * declared in an inner scope is gone by the time the guard goes */
	template <class T> int alloc(T *&p, size_t bytes)
	{
		p = nullptr;
		if (bytes == 0)
			bytes = 16;
		HIP_TRY(hipMalloc((void **) &p, bytes));
		owned.push_back((void *) p);
		return 0;
	}
	template <class T> void keep(T *&p)
	{
		for (auto &o : owned)
			if (o == (void *) p)
				o = nullptr;
	}
	~DevGuard()
	{
		for (auto o : owned)
			if (o)
				(void) hipFree(o);
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
		big_free(ix->d_vecs);	/* (blocks that did not come from big_alloc go straight to hipFree) */
		big_free(ix->d_tids);
	}
	big_free(ix->d_vecs_alt);
	big_free(ix->d_tids_alt);
	ix->d_vecs_alt = nullptr;
	ix->d_tids_alt = nullptr;
	ix->alt_cap = 0;
	ix->d_vecs = nullptr;
	ix->d_tids = nullptr;
	ix->own_rows = false;
	ix->nrows = 0;
	ix->norm_valid = false; ix->s16_valid = false; ix->seed_valid = false; ix->ipc_valid = false;
	ix->cap_rows = 0;
}

extern "C" int
ndbhip_ivf_destroy(ndbhip_ivf *ix)
{
	if (!ix)
		return NDBHIP_OK;
	if (ix->nshares > 0)
		return fail(NDBHIP_ERR_STATE, "ndbhip_ivf_destroy: %d shares of this mirror are alive (ndbhip_ivf_share): destroy them first", ix->nshares);
	if (ix->shared_of)
	{
		/* a share: its scratch, the matrices' per-batch tables, its pinned block — nothing persistent is its own */
		if (g.inited)
		{
			(void) hipDeviceSynchronize();		/* (its last batch may have run on another thread's stream) */
#define F(name) if (ix->name) (void) hipFree((void *) ix->name);
			NDB_IVF_SCRATCH(F)
#undef F
			for (S16Mat *m : {&ix->dm_sub, &ix->dm_cent, &ix->dm_all, &ix->dm_seed})
			{
				void	   *ptrs[] = {m->meta, m->pairs, m->desc, m->heads, m->zero};

				for (void *p : ptrs)
					if (p) (void) hipFree(p);
			}
			if (ix->pin) (void) hipHostFree(ix->pin);
		}
		{
			std::lock_guard<std::mutex> lk(ndbhip_g_mtx);

			ix->shared_of->nshares--;
		}
		delete ix;
		return NDBHIP_OK;
	}
	if (g.inited)
	{
		(void) hipStreamSynchronize(g.stream);
		ivf_free_rows(ix);
		void	   *ptrs[] = {ix->d_centroids, ix->d_loc_off, ix->d_glob_len, ix->d_owned, ix->d_own_lo, ix->d_own_len, ix->w_cdist,
			ix->w_probes, ix->w_candoff, ix->w_dist, ix->w_q, ix->w_otid, ix->w_odist, ix->w_ocnt,
			ix->w_gcnt, ix->w_goff, ix->w_pairs, ix->w_qblock, ix->w_qnorm, ix->w_scand, ix->w_sncand, ix->w_stotal, ix->w_tmin, ix->w_cblock,
			ix->d_xxmax, ix->w_rnorm, ix->w_scrt, ix->w_scrd, ix->w_scrc, ix->w_screc,
			ix->d_planes, ix->d_rn2, ix->d_rexp, ix->d_xmax16, ix->w_qplanes, ix->w_qn2, ix->w_qexp, ix->w_qthr,
			ix->w_ecount, ix->w_erec, ix->w_bmin, ix->w_s16desc, ix->d_blkoff, ix->d_lrad, ix->w_drop,
			ix->d_sub_first, ix->d_sub_len, ix->d_sub_loc, ix->d_sub_blk, ix->d_sub_rad, ix->d_sub_gidx, ix->d_subcent,
			(void *) ix->d_sub_cptr, ix->d_perm, ix->d_posof, ix->w_subdist, ix->w_pdist,
			ix->w_eub, ix->w_qcplanes, ix->w_qcn2, ix->w_qcexp, ix->w_amat, ix->w_cfull, ix->w_pslot, ix->d_prow_off, ix->d_pposof, ix->d_cn2, ix->d_plen, ix->d_bucket_list, ix->w_qhat, ix->d_allcent, ix->w_overq, ix->w_redo_q, ix->w_redo_p, ix->w_redo_out, ix->w_redo_idx, ix->w_qplanes_o, ix->w_qn2_o, ix->w_qexp_o, ix->d_cent_hat, ix->w_qpairs, ix->w_qpn, ix->d_seedrows, ix->d_seed_list, ix->d_seed_pos, ix->w_seedmat, ix->d_ipc_m2, ix->d_rnx, ix->d_bkt_rnxmax, ix->w_qev, ix->w_ppart, ix->w_qoffs, ix->w_qslot, ix->w_wmask, ix->w_wrec, ix->w_lqoff, ix->w_lq, ix->w_uoff, (void *) ix->d_sub_units};

		for (void *p : ptrs)
			if (p) (void) hipFree(p);
		ix->dm_sub.release();
		ix->dm_cent.release();
		ix->dm_all.release();
		ix->dm_seed.release();
		if (ix->pin) (void) hipHostFree(ix->pin);
	}
	delete ix;
	return NDBHIP_OK;
}

/* A second handle on the same mirror: rows, TIDs, planes, sublists, matrices — everything a search READS — are the
 * source's own arrays; everything a batch WRITES (the NDB_IVF_SCRATCH arrays, the matrices' per-batch tables, the pinned
 * result block, the heuristics' memory of earlier batches) is the new handle's.  What two batches in flight need
 * (ndbhip_set_thread_stream: a thread, a stream and a handle each) without a second copy of a mirror that is 1.55 x its
 * table.  Both handles are frozen while the share lives: loads, appends, deletes, builds and anything that would lay
 * the planes out again return NDBHIP_ERR_STATE — run a batch of every kind the shares will serve on the source first
 * (ndbhip_ivf_prepare, or one search), so that nothing is left to build. */
extern "C" int
ndbhip_ivf_share(ndbhip_ivf *src, ndbhip_ivf **out)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!src || !out)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (!src->loaded)
		return fail(NDBHIP_ERR_STATE, "index not loaded");
	if (src->shared_of)
		return fail(NDBHIP_ERR_STATE, "ndbhip_ivf_share: make shares from the handle that owns the mirror");
	if (!src->pend_list.empty())
		return fail(NDBHIP_ERR_STATE, "source index has pending appends: search or export it first");
	HIP_TRY(hipStreamSynchronize(g.stream));		/* (whatever the source still has in flight has written its tables) */
	ndbhip_ivf *ix = new (std::nothrow) ndbhip_ivf(*src);

	if (!ix)
		return fail(NDBHIP_ERR_NOMEM, "out of host memory");
#define F(name) ix->name = nullptr; ix->name##_n = 0;
	NDB_IVF_SCRATCH(F)
#undef F
	for (S16Mat *m : {&ix->dm_sub, &ix->dm_cent, &ix->dm_all, &ix->dm_seed})
	{
		m->meta = nullptr; m->meta_n = 0;
		m->pairs = nullptr; m->pairs_n = 0;
		m->desc = nullptr; m->desc_n = 0;
		m->heads = nullptr; m->heads_n = 0;
		m->zero = nullptr; m->zero_n = 0;
		m->nq = -1;
	}
	ix->pin = nullptr;
	ix->pin_n = 0;
	ix->d_vecs_alt = nullptr;		/* (the appends' second row buffer: a share takes no appends) */
	ix->d_tids_alt = nullptr;
	ix->alt_cap = 0;
	ix->redo.clear();
	ix->bat_subdist = nullptr;
	ix->shared_of = src;
	ix->nshares = 0;
	{
		std::lock_guard<std::mutex> lk(ndbhip_g_mtx);		/* (shares may be made and destroyed by different threads) */

		src->nshares++;
	}
	*out = ix;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_set_centroids(ndbhip_ivf *ix, const float *centroids, int ncent)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix || !centroids || ncent < 1)
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	IVF_NOT_FROZEN(ix, "ndbhip_ivf_set_centroids");
	ix->dm_cent_valid = false; ix->dm_all_valid = false;
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
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_load");
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
	ix->norm_valid = false; ix->s16_valid = false; ix->seed_valid = false; ix->ipc_valid = false;
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
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_load_f16");
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
	ix->norm_valid = false; ix->s16_valid = false; ix->seed_valid = false; ix->ipc_valid = false;
	ix->f16 = true;
	ix->loaded = true;
	return ivf_note_f16_subnormals(ix);
}

extern "C" int
ndbhip_ivf_load_device(ndbhip_ivf *ix, const int64_t *list_len, const uint8_t *owned,
					   const float *d_rows, const uint64_t *d_tids, int64_t nrows)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_load_device");
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
	ix->norm_valid = false; ix->s16_valid = false; ix->seed_valid = false; ix->ipc_valid = false;
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

			big_free(ix->d_vecs_alt);	/* (either buffer may be a block the build took from big_alloc) */
			big_free(ix->d_tids_alt);
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
	ix->norm_valid = false;
	ix->seed_valid = false;
	ix->ipc_valid = false;
	/* the centred planes take the new rows in the spare blocks of their lists (ndbhip_screen16c.h); anything else —
	 * other layouts, a list that has outgrown its spare blocks — is laid out again by the next screened batch */
	if (ix->s16_valid && ivf_s16c_append(ix, add, new_own) != 0)
		ix->s16_valid = false;
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
			  uint32_t row_words, size_t nrows)
{
	/* (blocks stride over the rows: a launch carries at most 2^32 - 1 work-items, i.e. 16.7 M blocks of 256) */
	for (size_t r = blockIdx.x; r < nrows; r += gridDim.x)
	{
		if (!keep[r])
			continue;
		const size_t d = pref[r];

		for (uint32_t j = threadIdx.x; j < row_words; j += 256)
			dst[d * row_words + j] = src[r * row_words + j];
		if (threadIdx.x == 0)
			dst_tids[d] = src_tids[r];
	}
}

__global__ void
k_delete_list_len(const uint32_t *__restrict__ pref, const int64_t *__restrict__ loc_off, int ncent,
				  int64_t *__restrict__ new_len)
{
	const int	L = blockIdx.x * blockDim.x + threadIdx.x;

	if (L < ncent)
		new_len[L] = (int64_t) pref[loc_off[L + 1]] - (int64_t) pref[loc_off[L]];
}

/* Centred planes after a delete: the planes keep the deleted rows as HOLES (position 0xFFFFFFFF: no candidate cap
 * reaches it, so the sweep never emits it and the seeds skip it) and every surviving row learns its new index in its
 * list — old index minus the deleted rows before it, which the compaction's own prefix sums give: pref[r] = new mirror
 * row of old row r.  One thread per padded plane row. */
__global__ void
k_s16c_delete_fix(const int64_t *__restrict__ prow_off, int nb, const uint32_t *__restrict__ plen,
				  const uint32_t *__restrict__ bucket_list, const int64_t *__restrict__ loc_off_old,
				  const uint8_t *__restrict__ keep, const uint32_t *__restrict__ pref, uint32_t *__restrict__ pposof)
{
	const int64_t pp = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;

	if (pp >= prow_off[nb])
		return;
	int			lo = 0, hi = nb;

	while (hi - lo > 1)
	{
		const int	mid = (lo + hi) >> 1;

		if (prow_off[mid] <= pp)
			lo = mid;
		else
			hi = mid;
	}
	while (lo + 1 < nb && prow_off[lo + 1] <= pp)
		lo++;
	if (pp - prow_off[lo] >= (int64_t) plen[lo])
		return;					/* spare rows */
	const uint32_t p = pposof[pp];

	if (p == 0xFFFFFFFFu)
		return;					/* a hole already */
	const int64_t l0 = loc_off_old[bucket_list[lo]], r = l0 + p;

	pposof[pp] = keep[r] ? pref[r] - pref[l0] : 0xFFFFFFFFu;
}

extern "C" int
ndbhip_ivf_delete(ndbhip_ivf *ix, const uint8_t *tids6, int64_t n, int64_t *removed)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_delete");
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
		hipLaunchKernelGGL(k_delete_move, dim3((unsigned) std::min<int64_t>(nrows, 1 << 23)), dim3(256), 0, g.stream,
						   (const uint8_t *) d_keep, (const uint32_t *) d_pref, (const uint32_t *) ix->d_vecs,
						   (const uint64_t *) ix->d_tids, (uint32_t *) nv, nt, row_words, (size_t) nrows);
		/* centred planes stay: holes for the deleted rows, new list positions for the others (the lists' radii remain
		 * upper bounds; a bucket emptied of live rows simply emits nothing) */
		const bool	planes_stay = ix->s16_valid && ix->s16_cen_layout && !ix->s16_cos_layout && !ix->f16 && ix->d_plen && ix->d_bucket_list &&
			!ix->s16_prow.empty();

		if (planes_stay)
		{
			const int	nbk = (int) ix->s16_prow.size() - 1;
			const int64_t npp = ix->s16_prow.back();

			if (npp > 0)
				hipLaunchKernelGGL(k_s16c_delete_fix, dim3((unsigned) ((npp + 255) / 256)), dim3(256), 0, g.stream,
								   (const int64_t *) ix->d_prow_off, nbk,
								   ix->s16_sub ? (const uint32_t *) ix->d_sub_len : (const uint32_t *) ix->d_plen,
								   (const uint32_t *) ix->d_bucket_list, (const int64_t *) ix->d_loc_off, (const uint8_t *) d_keep,
								   (const uint32_t *) d_pref, ix->d_pposof);
		}
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
		ix->norm_valid = false;
		ix->s16_valid = planes_stay;	/* (ivf_free_rows has just cleared it) */
		if (planes_stay)
			NDB_STAT_ADD(prepare_updates, 1);
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
	if (ix) IVF_NOT_FROZEN(ix, "ndbhip_ivf_append");
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
#include "ndbhip_screen16c.h"
#include "ndbhip_screen16d.h"
#include "ndbhip_screen16w.h"

/* ---- inner product on the centred sweep (ndbhip_screen16.h: s16c_ip_*): M^2 and the rows' constants M^2 - |x|^2 ---- */
/* one wave per mirror row: |x|^2 (fp64 sum, rounded to fp32), and the largest of them (float bits; NaN / inf rows are
 * not counted: their plane rows are marked and always emitted) */
template <bool F16>		/* the mirror holds fp16 rows: as the reference decodes them (fp16_to_float, quantization.c:170-218) */
__global__ __launch_bounds__(256) void
k_ipc_norms(const void *__restrict__ vecs, int64_t nrows, int dim, float *__restrict__ x2, uint32_t *__restrict__ m2_bits)
{
	const int	lane = threadIdx.x & 63;
	const int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);

	if (row >= nrows)
		return;
	double		s = 0.0;

	for (int i = lane; i < dim; i += 64)
	{
		const float v = F16 ? h2f_ref(((const uint16_t *) vecs)[(size_t) row * dim + i]) : ((const float *) vecs)[(size_t) row * dim + i];

		s += (double) v * (double) v;
	}
	s = wave_sum_f64(s);
	if (lane == 0)
	{
		const bool	ok = s <= 3.0e38;
		const float v = ok ? __double2float_ru(s) : __uint_as_float(0x7FC00000u);

		x2[row] = v;
		if (ok && __float_as_uint(v) > __atomic_load_n(m2_bits, __ATOMIC_RELAXED))
			atomicMax(m2_bits, __float_as_uint(v));
	}
}

/* one thread per padded plane row: which mirror row sits there (bucket -> list, index in the list), its constant */
__global__ __launch_bounds__(256) void
k_ipc_fill(const float *__restrict__ x2, const uint32_t *__restrict__ m2_bits, const int64_t *__restrict__ prow_off, int nb,
		   const uint32_t *__restrict__ bucket_list, const int64_t *__restrict__ loc_off, const uint32_t *__restrict__ own_len,
		   const uint32_t *__restrict__ pposof, float *__restrict__ rnx)
{
	const int64_t pp = (int64_t) blockIdx.x * 256 + threadIdx.x;

	if (pp >= prow_off[nb])
		return;
	int			lo = 0, hi = nb;

	while (hi - lo > 1)
	{
		const int	mid = (lo + hi) >> 1;

		if (prow_off[mid] <= pp)
			lo = mid;
		else
			hi = mid;
	}
	while (lo + 1 < nb && prow_off[lo + 1] <= pp)
		lo++;
	const uint32_t L = bucket_list[lo], pos = pposof[pp];
	float		r = 0.0f;

	if (pos < own_len[L])
	{
		const float v = x2[(size_t) loc_off[L] + pos];

		/* (rounded down; the sweep takes it another 2^-20 down or up as the bound needs) */
		r = v == v ? __double2float_rd((double) __uint_as_float(*m2_bits) - (double) v) : __uint_as_float(0x7FC00000u);
		r = r < 0.0f ? 0.0f : r;
	}
	rnx[pp] = r;
}

__global__ void
k_ipc_qev(const float *__restrict__ qn2, const uint32_t *__restrict__ m2_bits, int dim, uint32_t nq, float *__restrict__ qev)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;

	if (q < nq)
		qev[q] = s16c_ip_ev(dim, qn2[q], __uint_as_float(*m2_bits));
}

/* the fp16-MFMA screened scan in auto mode (ndbhip_set_option("screen16", 0) turns it off: the older fp32 bound
 * pass then serves batches of >= 128 queries); records a query may emit before the batch falls back */
static int	g_s16_auto = 1;
static int	g_build_s16 = 1;		/* ndbhip_set_option("build_screen16", 0): the build assigns on the vector ALU only */
static int	g_kmeans_s16 = 1;		/* "kmeans_screen16": the Lloyd iterations' assignment through the screen too (0: the vector ALU, rounds 1-5) */
static int	g_build_single_sweep = 1;	/* "build_single_sweep": the screened assignment multiplies the matrix once (k_s16_sweep MODE 4); 0 = two sweeps (rounds 2-5) */
static int	g_s16_waves = 4;
static int	g_s16_debug = 0;		/* timing experiments (wrong results): see k_s16_sweep's DBG */
static uint32_t g_s16_ecap = 8192;
static int	g_probe_sel_radix = 0;	/* batches of >= 512 queries: radix select instead of the full LDS sort ("probe_select_radix") */
static int	g_probe_sel_threads = 256;	/* threads of a k_probe_select block for batches of >= 512 queries ("probe_select_threads") */
static int	g_s16_stage = 1;		/* "screen16_stage": 0 = every lane loads the row it sums (k_s16_finalize, k_cent_select); 1 = rows
									 * streamed through LDS (s16_exact_staged), ring depth by batch size; n >= 2 = that depth */
static int	g_s16_fin_threads = 64;	/* threads of a k_s16_finalize block (one block per query; "screen16_fin_threads": 64 / 128 / 256) */
static int	g_s16_prune = 1;	/* (query, list) pairs excluded by |q - centroid| - list radius before the sweep ("screen16_prune") */
static int	g_s16_tighten = 1;	/* thresholds tightened inside the sweep (ndbhip_set_option("screen16_tighten", 0): only between the rounds) */
static int	g_s16_cen = 1;		/* L2 on float4 rows: the centred one-plane sweep (ndbhip_screen16c.h; "screen16_centered", 0: the two-plane sweep) */
static int	g_s16c_qb = 0;		/* pairs per tile of the centred sweep / 32: 4 or 1; 0 = from the previous batch's pairs per bucket ("screen16c_qb") */
/* a sharded search (ndbhip_comm.cpp) exchanges the queries' first thresholds between the seeds and the sweep: the
 * minimum over the ranks, in place, of the (threshold, unused) pairs — every rank must call it once per sub-batch */
static int	(*g_thr_hook) (float *, size_t) = nullptr;

/*
 * The exchange is an element-wise MINIMUM over the ranks.  A threshold (x = thr + e, y = e: the reference bound plus
 * this rank's error term, which grows with the largest row norm IT holds) cannot simply be min'd: rank B would sweep
 * against thr_A + e_A with e_A < e_B.  Packed as (x, -y) the minimum gives x* = the smallest thr + e and e_max = the
 * largest error term; x* + e_max >= thr_r* + e_B for every rank B, so (x* + e_max, e_max) is valid everywhere (looser
 * than the best possible by at most the smallest e).  The centred path's thresholds carry y = 0 and pass unchanged.
 */
__global__ void
k_thr_pack(float2 *__restrict__ qthr, uint32_t nq)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;

	if (q < nq)
		qthr[q].y = -qthr[q].y;
}

__global__ void
k_thr_unpack(float2 *__restrict__ qthr, uint32_t nq)
{
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;

	if (q >= nq)
		return;
	const float2 v = qthr[q];
	const float e = -v.y;

	qthr[q] = make_float2(e > 0.0f ? s16_up(v.x + e) : v.x, e);
}

/* the queries' thresholds through the hook (a sharded search: ndbhip_comm.cpp) */
static int
ivf_exchange_thresholds(float2 *qthr, int nq)
{
	hipLaunchKernelGGL(k_thr_pack, dim3((nq + 255) / 256), dim3(256), 0, g.stream, qthr, (uint32_t) nq);
	const int	rc = g_thr_hook((float *) qthr, (size_t) 2 * nq);

	if (rc)
		return rc;
	hipLaunchKernelGGL(k_thr_unpack, dim3((nq + 255) / 256), dim3(256), 0, g.stream, qthr, (uint32_t) nq);
	return 0;
}

extern "C" void
ndbhip_internal_set_thr_hook(int (*fn) (float *, size_t))
{
	g_thr_hook = fn;
}

static int	g_s16_slack = 1;	/* the centred planes keep spare blocks and take appends in place ("screen16_slack", 0: every append lays the planes out again) */
static int	g_s16c_seeds = 0;	/* rows whose upper bounds give a query its first threshold, 0 = 32 (k <= 20) or 64 ("screen16c_seeds") */
static int	g_s16c_epi = 1;		/* 1: the sweep's matrix pipe screens its own accumulator blocks before the per-element test; 0: every element tested (A/B, "screen16c_epi") */
static int	g_s16c_pf = 0;		/* variants of k_s16c_sweep<8, 2> for A/B: 3 = an in-wave L2 prefetch, 16 / 32 / 48 = non-temporal rows / pairs / both ("screen16c_pf") */
static int	g_s16c_dense = 1;	/* dense buckets (tile of 256 x 256) run k_s16c_dense (ndbhip_screen16d.h: loader and prefetcher waves); 0: k_s16c_sweep<8, 2> ("screen16c_dense") */
static int	g_s16c_sample = 2048;	/* rows of the mirror sampled for a dense batch's first thresholds, 0 = none ("screen16c_sample") */
/* The dense tile's kernel (ndbhip_screen16d.h) from this many pairs per bucket with pairs (the previous batch's): round 6,
 * measured on 1M x 768 at 512 .. 4096 queries a batch (tools/dense_probe.py, SIGMA = 0.5 / 1.0): whole lists 47 pairs a bucket
 * 0.82 against the ring's 0.86 ms, 74: 1.30 / 1.58, 133: 2.14 / 2.90 (25: 0.56 / 0.48 — the ring's); layouts with sublists
 * (buckets of ~128 rows fill half a row tile) 131: 1.56 / 1.79, 72: 1.04 / 1.01, 45: 0.76 / 0.59.  Until round 5 the rule
 * was 320 and whole lists only: a balanced table at 4096 queries (128 pairs a list) stayed on the 128 x 128 ring at 0.2 of
 * the matrix peak.  Later in round 6, with the kernel's one-pair-block map for tiles of <= 128 members (SMALL): whole lists
 * at 24 / 32 / 48 pairs a bucket 0.60 / 0.75 / 0.96 against the ring's 0.66 / 0.83 / 1.18 ms — from 24 (where the 32-pair
 * tile's sparse sweeps end); with sublists the two meet around 64 (0.94 / 0.97 ms) and the ring wins below (0.80 / 0.77 at
 * 48): 100 stays. */
static int	g_s16c_dense_min = 24;		/* "screen16c_dense_min" */
static int	g_s16c_dense_min_sub = 100;	/* "screen16c_dense_min_sub": the same for regrouped planes (sublists); 0 = never */
/* "screen16_sub_restrict": from this many regrouped-list centres on, a batch scores the centres of its PROBED lists only
 * (k_subdist_lists, list-major, fp32 vector ALU) instead of multiplying every query by every centre along with the
 * centroids: 0 = never, the default.  MEASURED (round 6, 10M x 768, lists 4096 = C4: 78 k centres, the full matrix 0.79 of a
 * 4.3 ms step): the step went from 4.67 to 22.3 ms.  The reference's build rule leaves that table a few lists of hundreds
 * of thousands of rows, every query probes them, and they are the lists with thousands of sublists: the (query, centre)
 * pairs a batch needs are a quarter of the full matrix, not the 1 % that 32 probes x 19 sublists of a BALANCED index
 * would be, and the vector ALU multiplies at a twentieth of the matrix cores' rate.  Same results either way
 * (tests/test_gpu_screen16.py); kept for tables whose lists are balanced. */
static int	g_sub_restrict = 0;
static int	g_s16c_tight = 128;	/* k_s16c_dense tightens a query's threshold every this many records (power of two; "screen16c_tight") */
static int	g_s16c_rot = 0;		/* k_s16c_dense takes an item's chunks in an order rotated by its row tile ("screen16c_rot") */
static int	g_s16c_dense_spare = 0;	/* compute units k_s16c_dense leaves free on a mirror that has shares, i.e. steps in flight: the tile fills a CU's LDS and registers, so other steps' kernels run only where it is not ("screen16c_dense_spare") */
static int	g_s16c_dense_split = 3;	/* 32-row blocks of a tile's eight that k_s16c_dense's loader waves multiply: 4 (as many as the multipliers) or 3 ("screen16c_dense_split") */
static int	g_s16c_dense_small = 1;	/* k_s16c_dense: tiles of <= 128 members take the one-pair-block wave map ("screen16c_dense_small") */
static int	g_s16c_dense_sync = 16;	/* k_s16c_dense: the blocks of an XCD meet before every this many-th item, so that the blocks sharing an operand tile ask for its chunks within the L2's memory (0: never; "screen16c_dense_sync") */
static int	g_s16c_pfd = 0;		/* chunks k_s16c_dense's prefetchers run ahead of its loaders, 0 = no prefetch ("screen16c_pfd") */
static int	g_s16c_wave = 2;	/* sparse pair tables (32-pair tiles): k_s16c_wsweep (ndbhip_screen16w.h: wave-autonomous register streams) with this many chunks a wave in flight (2 .. 4; at most the chunks of a row); 0: k_s16c_sweep<1, NBUF>, the LDS ring ("screen16c_wave") */
static int	g_s16c_plseed = 1;	/* first thresholds from the sweep's own planes (block 0 of the nearest sublist) instead of float4 rows ("screen16c_plane_seeds") */
static int	g_s16c_wave_min_nq = 1024;	/* batches from this many queries up take k_s16c_wsweep ("screen16c_wave_min_nq") */
static int	g_s16c_wblk = 2;	/* blocks of 4 waves per compute unit that k_s16c_wsweep is launched with (1, 2, or 3 when two chunks are in flight per wave: the registers of that form allow three; "screen16c_wave_blocks") */
static int	g_s16c_nbuf = 0;	/* ring depth of the centred sweep, 0 = the geometry's default ("screen16c_nbuf") */

static long	g_slow_call_us = 0;	/* "slow_call_log": host-pointer searches slower than this many microseconds report their phases on stderr */
static int	g_s16c_bigk = 1;	/* 64 < k <= 256 on the centred fp16 screen ("screen16c_bigk"; 0: the fp32 screen, as before round 5) */
static int	g_s16_redo = 1;		/* queries whose records / survivors overflow go to the exact path alone ("screen16_redo"; 0: the whole batch does) */
static int	g_s16_cos = 1;		/* cosine on the matrix-core sweep, as the inner product of normalised planes ("screen16_cosine") */

static bool
ivf_s16_eligible(const ndbhip_ivf *ix, int nq, int R, int k)
{
	const size_t dimp = (size_t) ((ix->dim + 63) & ~63);

	if (R != R_IVF_L2 && R != R_IVF_IP && !(R == R_IVF_COS && g_s16_cos))
		return false;
	if (ix->nrows < 1)
		return false;
	/* (64 < k <= 256: L2 on the centred planes over sublists only — ivf_s16_run sends anything else back) */
	if (k > NDB_TOPK_FAST_MAXK && !(g_s16c_bigk && k <= NDB_S16_MAXK && (R == R_IVF_L2 || R == R_IVF_IP || (R == R_IVF_COS && g_s16_cos)) && !ix->s16_bigk_off))
		return false;
	if (ix->f16 && (ix->dim % 64) != 0)
		return false;
	if ((size_t) nq * dimp * 4 >= ((size_t) 1 << 32) || (size_t) 256 * dimp * 4 >= ((size_t) 1 << 31))
		return false;
	return true;
}

static int	ivf_s16_build_sublists(ndbhip_ivf *ix, std::vector<uint32_t> &blk_off_host, bool slack,
								   const float *rows32 = nullptr /* the rows to regroup (cosine: their normalised copy) */,
								   const float *list_centres = nullptr /* (cosine: the normalised centroids) */ );	/* ndbhip_build.h */
static int	ivf_s16_sub_distances(ndbhip_ivf *ix, const float *d_q, int nq, uint32_t *sstride);
static int	ivf_s16_sub_distances_probed(ndbhip_ivf *ix, const float *d_q, int nq, const int *w_probes, int npr, uint32_t *sstride);
static int	s16mat_prepare(S16Mat &M, const float *d_src, int n, int dim);	/* ndbhip_build.h */
__global__ void k_rows_decode_f16(const uint16_t *__restrict__ src, size_t n, float *__restrict__ out);	/* ndbhip_build.h */
static int	s16mat_run(S16Mat &M, int dim, const unsigned char *qplanes, const float *qn2, const int *qexp, float2 *qthr,
					   int nq, float *out, uint32_t stride);
static int	g_cent_s16 = 1;		/* screened batches: the centroid scan on the matrix cores + exact arithmetic near the nprobe-th ("cent_screen16") */
static int	g_s16_sublists = 1;	/* long lists regrouped into sublists ("screen16_sublists") */
static int	g_s16_sub_min = 256;	/* lists longer than this are regrouped where that shrinks their radius ("screen16_sub_min") */
static int	g_s16_sub_rows = 128;	/* ... into sublists of about this many rows ("screen16_sub_rows") */

/* does the centred one-plane sweep (ndbhip_screen16c.h) serve this recipe on this mirror */
static int	g_s16_ip_cen = 1;	/* inner product on the CENTRED sweep: |q - x|^2 + M^2 - |x|^2 ("screen16_ip_centered") */
static int	g_s16_cos_cen = 1;	/* cosine on the CENTRED sweep: |q^ - x^|^2 = 2 x cosine distance ("screen16_cosine_centered") */

static bool
ivf_s16_centered(const ndbhip_ivf *ix, int R)
{
	/* (cosine: the planes come from a normalised fp32 copy whatever the mirror holds; L2 on an fp16 mirror: from a
	 * transient copy of the rows as the reference decodes them) */
	/* (inner product: on the L2 layout's own planes, every bound shifted by the row's M^2 - |x|^2 (s16c_ip_*); an fp16 mirror's planes come
	 * from its decoded rows as for L2; not on a shard — the thresholds carry M^2, which every rank has its own of) */
	return g_s16_cen != 0 && (R == R_IVF_L2 || (R == R_IVF_COS && g_s16_cos && g_s16_cos_cen) ||
							  (R == R_IVF_IP && g_s16_ip_cen && !g_thr_hook));
}

/*
 * The sweep's operands, once per version of the mirror (and per layout: sublist settings, centred or not): sublists
 * of the long lists, the rows' fp16 planes / norms / exponents in the blocked layout, list radii.  The first
 * screened batch runs it if nobody did (ndbhip_ivf_prepare: a build that wants its index searchable at once).
 */
static int
ivf_s16_prepare(ndbhip_ivf *ix, int R)
{
	const int	dim = ix->dim, dimp = (dim + 63) & ~63;
	const int	nc = ix->ncent;
	/* sublists only pay where a bound can exclude them: L2 (an index has one operator class, hence one strategy;
	 * a caller that alternates strategies on one mirror has its planes laid out again at every change) */
	const int	sub_cfg = (g_s16_sublists && g_s16_prune && (R == R_IVF_L2 || R == R_IVF_IP || R == R_IVF_COS)) ? (g_s16_sub_min * 131 + g_s16_sub_rows) : 0;
	/* cosine: the planes (and the sublists) are those of the rows divided by their norms */
	const bool	cosn = R == R_IVF_COS;

	/* L2 on float4 rows: the planes hold the rows minus their bucket's centre, one fp16 plane (ndbhip_screen16c.h) */
	const bool	cen = ivf_s16_centered(ix, R);
	const int	lay_cfg = (sub_cfg * 2 + (cen ? 1 : 0)) * 2 + (cosn ? 1 : 0);

	if (ix->s16_valid && ix->s16_sub_cfg != lay_cfg && !ivf_frozen(ix))
		ix->s16_valid = false;	/* the planes were laid out under other settings */
	if (ix->s16_valid && ix->s16_sub_cfg != lay_cfg)
		return fail(NDBHIP_ERR_STATE, "the mirror is shared (ndbhip_ivf_share) and its planes were laid out for another kind of search or under other settings");
	if (!ix->s16_valid && ivf_frozen(ix))
		return fail(NDBHIP_ERR_STATE, "the mirror is shared (ndbhip_ivf_share) and has no planes yet: ndbhip_ivf_prepare on the source before sharing");
	if (!ix->s16_valid)
	{
		std::vector<uint32_t> bo;
		uint64_t	nb = 0;

		NDB_STAT_ADD(prepares, 1);
		ix->ipc_valid = false;		/* (the rows' constants are indexed by padded plane row) */
		ix->s16_sub = false;
		ix->s16_sub_cfg = lay_cfg;
		ix->s16_planes_f32 = cosn || !ix->f16;
		/* cosine: a transient fp32 copy of the rows divided by their norms (decoded like the reference decodes an fp16
		 * mirror) — what is regrouped and what the planes are made of; released when this returns */
		struct Hat
		{
			float	   *p = nullptr;
			~Hat() { if (p) big_free(p); }
		}			hat;

		const bool	dech = !cosn && cen && ix->f16;		/* the centred planes of an fp16 mirror: made from its decoded rows */

		if (dech)
		{
			const size_t ne = (size_t) ix->nrows * dim;

			if (big_alloc((void **) &hat.p, ne * sizeof(float))) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_rows_decode_f16, dim3((unsigned) ((ne + 255) / 256)), dim3(256), 0, g.stream,
							   (const uint16_t *) ix->d_vecs, ne, hat.p);
		}
		if (cosn)
		{
			if (big_alloc((void **) &hat.p, (size_t) ix->nrows * dim * sizeof(float))) return NDBHIP_ERR_HIP;
			const dim3	gn((unsigned) ((ix->nrows + 3) / 4));

			if (!ix->f16)
				hipLaunchKernelGGL(k_rows_normalise<0>, gn, dim3(256), 0, g.stream, (const void *) ix->d_vecs, ix->nrows, dim, hat.p, cen ? 1 : 0);
			else if (ix->f16_sub)
				hipLaunchKernelGGL(k_rows_normalise<1>, gn, dim3(256), 0, g.stream, (const void *) ix->d_vecs, ix->nrows, dim, hat.p, cen ? 1 : 0);
			else
				hipLaunchKernelGGL(k_rows_normalise<2>, gn, dim3(256), 0, g.stream, (const void *) ix->d_vecs, ix->nrows, dim, hat.p, cen ? 1 : 0);
		}
		ix->s16_cos_layout = cosn;
		if (cosn)
		{
			if (grow(ix->d_cent_hat, ix->d_cent_hat_n, (size_t) nc * dim)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_rows_normalise<0>, dim3((nc + 3) / 4), dim3(256), 0, g.stream, (const void *) ix->d_centroids,
							   (int64_t) nc, dim, ix->d_cent_hat);
		}
		if (sub_cfg != 0)
		{
			/* long lists regrouped into sublists: sets ix->s16_sub and the d_sub_* tables, bo = their block offsets */
			const int	rc = ivf_s16_build_sublists(ix, bo, cen && g_s16_slack != 0, hat.p, cosn ? ix->d_cent_hat : (const float *) nullptr);

			if (rc)
				return rc;
		}
		if (!ix->s16_sub)
		{
			/* every list starts a new 32-row block of the blocked planes (the centred planes: with spare blocks
			 * behind it, for the rows inserted later) */
			bo.assign((size_t) nc + 1, 0);
			ix->s16_tail.assign((size_t) nc, -1);
			ix->s16_blen.assign((size_t) nc, 0);
			for (int c = 0; c < nc; c++)
			{
				bo[c] = (uint32_t) nb;
				nb += (uint64_t) ((ix->own_len[c] + 31) / 32);
				ix->s16_blen[(size_t) c] = (uint32_t) ix->own_len[c];
				if (cen && g_s16_slack)
				{
					nb += std::max<uint64_t>(2, (uint64_t) ix->own_len[c] / 256);
					ix->s16_tail[(size_t) c] = c;
				}
			}
			bo[nc] = (uint32_t) nb;
			if (nb + 8 > 0xFFFFFFFFull)
				return fail(NDBHIP_ERR_UNSUPPORTED, "more than 2^32 row blocks");
			if (grow(ix->d_blkoff, ix->d_blkoff_n, (size_t) nc + 1)) return NDBHIP_ERR_HIP;
			HIP_TRY(hipMemcpyAsync(ix->d_blkoff, bo.data(), ((size_t) nc + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
			HIP_TRY(hipStreamSynchronize(g.stream));		/* bo is a local */
		}
		nb = bo.back();
		const size_t nbk = bo.size() - 1;			/* buckets */

		ix->s16_cen_layout = cen;
		ix->s16_bcap.assign(nbk, 0);
		for (size_t b2 = 0; b2 < nbk; b2++)
			ix->s16_bcap[b2] = (bo[b2 + 1] - bo[b2]) * 32u;
		if (cen)
		{
			/* the centred path indexes norms / exponents / list positions by PADDED plane row (32 x block + row) */
			std::vector<int64_t> po(nbk + 1);

			for (size_t b2 = 0; b2 <= nbk; b2++)
				po[b2] = (int64_t) bo[b2] * 32;
			ix->s16_prow = po;
			if (grow(ix->d_prow_off, ix->d_prow_off_n, nbk + 1)) return NDBHIP_ERR_HIP;
			if (grow(ix->d_pposof, ix->d_pposof_n, (size_t) (nb + 8) * 32)) return NDBHIP_ERR_HIP;
			HIP_TRY(hipMemcpyAsync(ix->d_prow_off, po.data(), (nbk + 1) * sizeof(int64_t), hipMemcpyHostToDevice, g.stream));
			HIP_TRY(hipMemsetAsync(ix->d_pposof, 0xFF, (size_t) (nb + 8) * 32 * sizeof(uint32_t), g.stream));
			/* bucket -> list, and the buckets' row counts in the planes (what a delete must not shrink) */
			std::vector<uint32_t> blist(nbk);

			if (ix->s16_sub)
			{
				std::vector<uint32_t> first((size_t) nc + 1);

				HIP_TRY(hipMemcpyAsync(first.data(), ix->d_sub_first, ((size_t) nc + 1) * 4, hipMemcpyDeviceToHost, g.stream));
				HIP_TRY(hipStreamSynchronize(g.stream));
				for (int c = 0; c < nc; c++)
					for (uint32_t b2 = first[(size_t) c]; b2 < first[(size_t) c + 1] && b2 < nbk; b2++)
						blist[b2] = (uint32_t) c;
			}
			else
				for (size_t b2 = 0; b2 < nbk; b2++)
					blist[b2] = (uint32_t) b2;
			{
				/* the most 32-row blocks the buckets of ONE list have (spare blocks included): what a (query, probe)
				 * pair can need of (pair, block) words at most (ndbhip_screen16w.h) */
				std::vector<uint64_t> lw((size_t) nc, 0);
				uint64_t	mx = 1;

				for (size_t b2 = 0; b2 < nbk; b2++)
					if (blist[b2] < (uint32_t) nc)
						lw[blist[b2]] += ix->s16_bcap[b2] / 32u;
				for (int c = 0; c < nc; c++)
					mx = std::max(mx, lw[(size_t) c]);
				ix->s16w_maxlw = (uint32_t) std::min<uint64_t>(mx, 0xFFFFFFu);
				ix->s16w_avgblk = (uint32_t) std::max<uint64_t>(1, ((uint64_t) nb + nbk - 1) / std::max<size_t>(nbk, 1));
			}
			if (grow(ix->d_bucket_list, ix->d_bucket_list_n, nbk)) return NDBHIP_ERR_HIP;
			if (grow(ix->d_plen, ix->d_plen_n, nbk)) return NDBHIP_ERR_HIP;
			HIP_TRY(hipMemcpyAsync(ix->d_bucket_list, blist.data(), nbk * 4, hipMemcpyHostToDevice, g.stream));
			HIP_TRY(hipMemcpyAsync(ix->d_plen, ix->s16_blen.data(), nbk * 4, hipMemcpyHostToDevice, g.stream));
			HIP_TRY(hipStreamSynchronize(g.stream));		/* po, blist are locals */
		}
		const size_t blk_bytes = cen ? (size_t) (dimp / S16C_CH) * 4096 : (size_t) (dimp / S16_CH) * (ix->s16_planes_f32 ? 4096 : 2048);

		if (grow(ix->d_rn2, ix->d_rn2_n, cen ? (size_t) (nb + 8) * 32 : (size_t) ix->nrows)) return NDBHIP_ERR_HIP;
		if (grow(ix->d_rexp, ix->d_rexp_n, cen ? (size_t) (nb + 8) * 32 : (size_t) ix->nrows)) return NDBHIP_ERR_HIP;
		if (grow(ix->d_planes, ix->d_planes_n, (size_t) (nb + 8) * blk_bytes)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemsetAsync(ix->d_planes, 0, (size_t) (nb + 8) * blk_bytes, g.stream));
		if (!ix->d_xmax16)
			HIP_TRY(hipMalloc((void **) &ix->d_xmax16, sizeof(uint32_t)));
		HIP_TRY(hipMemsetAsync(ix->d_xmax16, 0, sizeof(uint32_t), g.stream));
		const dim3	gp((unsigned) ((ix->nrows + 3) / 4));

#define S16_PREP_L(HH)                                                                                          \
		hipLaunchKernelGGL(k_s16_row_prep<HH>, gp, dim3(256), 0, g.stream, cosn ? (const void *) hat.p : (const void *) ix->d_vecs, ix->nrows, dim, dimp, \
						   ix->s16_sub ? (const int64_t *) ix->d_sub_loc : (const int64_t *) ix->d_loc_off,              \
						   ix->s16_sub ? (const uint32_t *) ix->d_sub_blk : (const uint32_t *) ix->d_blkoff,            \
						   ix->s16_sub ? ix->nsub : nc, ix->d_planes, ix->d_rn2, ix->d_rexp, ix->d_xmax16,               \
						   ix->s16_sub ? (const int64_t *) ix->d_perm : (const int64_t *) nullptr)
		if (cen)
			hipLaunchKernelGGL(k_s16c_row_prep, gp, dim3(256), 0, g.stream, hat.p ? (const float *) hat.p : (const float *) ix->d_vecs, ix->nrows, dim, dimp,
							   ix->s16_sub ? (const int64_t *) ix->d_sub_loc : (const int64_t *) ix->d_loc_off,
							   ix->s16_sub ? (const uint32_t *) ix->d_sub_blk : (const uint32_t *) ix->d_blkoff,
							   ix->s16_sub ? ix->nsub : nc, cosn ? (const float *) ix->d_cent_hat : (const float *) ix->d_centroids,
							   ix->s16_sub ? (const float *const *) ix->d_sub_cptr : (const float *const *) nullptr,
							   ix->d_planes, ix->d_rn2, ix->d_rexp,
							   ix->s16_sub ? (const int64_t *) ix->d_perm : (const int64_t *) nullptr,
							   (const int64_t *) ix->d_prow_off, ix->s16_sub ? (const uint32_t *) ix->d_posof : (const uint32_t *) nullptr,
							   ix->d_pposof);
		else if (ix->s16_planes_f32)
			S16_PREP_L(0);
		else if (ix->f16_sub)
			S16_PREP_L(1);
		else
			S16_PREP_L(2);
		/* radius of every list around its centroid (largest |x - c| over the rows held here, rounded up; +inf for a
		 * list with a row or centroid beyond fp32): lets a whole (query, list) pair be excluded by the triangle
		 * inequality before any of its rows is scored (k_s16_pair_prune) */
		if (grow(ix->d_lrad, ix->d_lrad_n, (size_t) nc)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemsetAsync(ix->d_lrad, 0, (size_t) nc * sizeof(uint32_t), g.stream));
		if (!ix->f16)
			hipLaunchKernelGGL(k_s16_list_radius<0>, gp, dim3(256), 0, g.stream, (const void *) ix->d_vecs, ix->nrows, dim,
							   (const int64_t *) ix->d_loc_off, nc, (const float *) ix->d_centroids, ix->d_lrad);
		else if (ix->f16_sub)
			hipLaunchKernelGGL(k_s16_list_radius<1>, gp, dim3(256), 0, g.stream, (const void *) ix->d_vecs, ix->nrows, dim,
							   (const int64_t *) ix->d_loc_off, nc, (const float *) ix->d_centroids, ix->d_lrad);
		else
			hipLaunchKernelGGL(k_s16_list_radius<2>, gp, dim3(256), 0, g.stream, (const void *) ix->d_vecs, ix->nrows, dim,
							   (const int64_t *) ix->d_loc_off, nc, (const float *) ix->d_centroids, ix->d_lrad);
		/* |centroid|^2: the inner product's sublist bound turns |q - c|^2 into q.c with it */
		if (grow(ix->d_cn2, ix->d_cn2_n, (size_t) nc)) return NDBHIP_ERR_HIP;
		hipLaunchKernelGGL(k_vec_norm2, dim3((nc + 3) / 4), dim3(256), 0, g.stream, (const float *) ix->d_centroids, nc, dim, ix->d_cn2);
		HIP_TRY(hipGetLastError());
		if (hat.p)
			HIP_TRY(hipStreamSynchronize(g.stream));	/* the transient copy goes with this scope */
		ix->dm_all_valid = false;
		{
			const int	ncmp = std::min(ix->nlists, ix->ncent);

			if (ix->s16_sub && ix->nsub_g > 0 && !cosn && g_cent_s16 && ncmp >= 256 && ncmp <= 4096)
			{
				const size_t nall = (size_t) ncmp + (size_t) ix->nsub_g;

				if (grow(ix->d_allcent, ix->d_allcent_n, nall * (size_t) dim)) return NDBHIP_ERR_HIP;
				HIP_TRY(hipMemcpyAsync(ix->d_allcent, ix->d_centroids, (size_t) ncmp * dim * sizeof(float), hipMemcpyDeviceToDevice, g.stream));
				HIP_TRY(hipMemcpyAsync(ix->d_allcent + (size_t) ncmp * dim, ix->d_subcent, (size_t) ix->nsub_g * dim * sizeof(float),
									   hipMemcpyDeviceToDevice, g.stream));
				const int	rc = s16mat_prepare(ix->dm_all, ix->d_allcent, (int) nall, dim);

				if (rc)
					return rc;
				ix->dm_all_valid = true;
				ix->dm_all_ncmp = ncmp;
			}
		}
		ix->s16_valid = true;
		ix->s16_bigk_off = false;		/* (a new layout: k > 64 gets another try) */
		ix->s16w_off = false;			/* ... and so does the register-streaming sweep, with word arrays sized afresh */
		ix->wc_words = 0;
		ix->wc_mult = 4;
	}
	return 0;
}

/*
 * Rows appended since the planes were laid out (ivf_flush has just put them behind their lists in the mirror): the
 * centred layout keeps spare 32-row blocks behind the bucket of every list that takes appends — the list itself, or
 * the extra sublist around the centroid of a regrouped list — so the new rows' planes, norms, exponents and list
 * positions are written there (k_s16c_row_append), the bucket's and the list's radius grow by a max, and nothing
 * else moves.  Returns 0 when done, 1 when the layout cannot take them (the caller invalidates it: the next
 * screened batch lays everything out again, with fresh spare blocks), negative on error.
 */
static int
ivf_s16c_append(ndbhip_ivf *ix, const std::vector<int64_t> &add, const std::vector<int64_t> &new_own)
{
	const int	nc = ix->ncent, dim = ix->dim, dimp = (dim + 63) & ~63;

	if (!ix->s16_cen_layout || ix->s16_cos_layout || !g_s16_slack || ix->f16 || (int) ix->s16_tail.size() != nc)
		return 1;				/* (the cosine layout's planes are normalised rows: laid out again) */
	std::vector<S16CApp> recs;
	std::vector<uint32_t> bidx, bval;

	for (int c = 0; c < nc; c++)
	{
		if (add[c] <= 0 || !ix->owned[c])
			continue;
		const int	b = ix->s16_tail[(size_t) c];

		if (b < 0 || (int64_t) ix->s16_blen[(size_t) b] + add[c] > (int64_t) ix->s16_bcap[(size_t) b])
			return 1;
		const int64_t old_own = new_own[c] - add[c];

		for (int64_t i = 0; i < add[c]; i++)
		{
			S16CApp		a;

			a.pos = (uint32_t) (old_own + i);
			a.row = ix->loc_off[c] + old_own + i;
			a.pp = ix->s16_prow[(size_t) b] + (int64_t) ix->s16_blen[(size_t) b] + i;
			a.bucket = (uint32_t) b;
			a.list = (uint32_t) c;
			a.pad = 0;
			recs.push_back(a);
		}
		ix->s16_blen[(size_t) b] += (uint32_t) add[c];
		bidx.push_back((uint32_t) b);
		bval.push_back(ix->s16_blen[(size_t) b]);
	}
	if (recs.empty())
		return 0;
	DevGuard	tmp;
	S16CApp    *d_recs = nullptr;
	uint32_t   *d_bidx = nullptr, *d_bval = nullptr;

	if (tmp.alloc(d_recs, recs.size() * sizeof(S16CApp))) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_bidx, bidx.size() * 4)) return NDBHIP_ERR_HIP;
	if (tmp.alloc(d_bval, bval.size() * 4)) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemcpyAsync(d_recs, recs.data(), recs.size() * sizeof(S16CApp), hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_bidx, bidx.data(), bidx.size() * 4, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_bval, bval.data(), bval.size() * 4, hipMemcpyHostToDevice, g.stream));
	hipLaunchKernelGGL(k_s16c_row_append, dim3((unsigned) ((recs.size() + 3) / 4)), dim3(256), 0, g.stream, (const float *) ix->d_vecs,
					   dim, dimp, (const S16CApp *) d_recs, (uint32_t) recs.size(), (const float *) ix->d_centroids,
					   ix->s16_sub ? (const float *const *) ix->d_sub_cptr : (const float *const *) nullptr, ix->d_planes,
					   ix->d_rn2, ix->d_rexp, ix->d_pposof, ix->s16_sub ? ix->d_sub_rad : (uint32_t *) nullptr, ix->d_lrad);
	/* the sweep's own table of bucket lengths: the sublists' (regrouped planes), or the lists' rows in the planes */
	hipLaunchKernelGGL(k_s16c_set_u32, dim3((unsigned) ((bidx.size() + 255) / 256)), dim3(256), 0, g.stream,
					   ix->s16_sub ? ix->d_sub_len : ix->d_plen,
					   (const uint32_t *) d_bidx, (const uint32_t *) d_bval, (uint32_t) bidx.size());
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(g.stream));		/* the host arrays are locals */
	NDB_STAT_ADD(prepare_updates, 1);
	return 0;
}

/*
 * Batches in flight on several streams (ndbhip_set_thread_stream): the point is that one batch's per-query chains run
 * under ANOTHER batch's sweep — not that two sweeps share the device, each then taking twice as long with nothing
 * gained.  So the sweeps queue up: a sweep is launched behind the event that ends the sweep launched before it,
 * whichever stream that was on (a ring of events; the mutex only orders the launches, nobody waits on the host).
 * "screen16_sweep_queue" 0 turns that off.
 */
static int	g_s16_sweepq = 1;
static std::mutex g_sweepq_mtx;
static hipEvent_t g_sweepq_ev[8];
static unsigned g_sweepq_n = 0;
static hipStream_t g_sweepq_last = nullptr;

struct SweepTurn
{
	std::unique_lock<std::mutex> lk;
	bool		on = false;
	int begin()
	{
		if (!g_s16_sweepq || !ndbhip_tl_stream)
			return 0;			/* one stream: nothing to queue behind */
		lk = std::unique_lock<std::mutex>(g_sweepq_mtx);
		on = true;
		if (g_sweepq_n > 0 && g_sweepq_last != (hipStream_t) g.stream)
			HIP_TRY(hipStreamWaitEvent(g.stream, g_sweepq_ev[(g_sweepq_n - 1) & 7u], 0));
		return 0;
	}
	int end()
	{
		if (!on)
			return 0;
		hipEvent_t &e = g_sweepq_ev[g_sweepq_n & 7u];

		if (!e)
			HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(e, g.stream));
		g_sweepq_last = (hipStream_t) g.stream;
		g_sweepq_n++;
		lk.unlock();
		on = false;
		return 0;
	}
};

/* Runs the sweep + finalize for one sub-batch whose probes / candidate offsets are already on the device.
 * Returns 0, a negative error, or 1 when some query overflowed (nothing usable was written: rerun on the older path). */
static int
ivf_s16_run(ndbhip_ivf *ix, const IvfDev &d, const float *d_q, int nq, int R, int npr, int k, const int *w_probes,
			const uint32_t *lco, int partial, ndbhip_cand *d_cand, int *d_ncand, int64_t *d_total,
			uint64_t *d_otid, float *d_odist, int *d_ocnt,
			const float *cdist /* the centroid scan's [nq][cstride] L2 distances, or NULL (probes chosen elsewhere) */,
			uint32_t cstride)
{
	const int	dim = ix->dim, dimp = (dim + 63) & ~63;	/* two chunks per accumulator block */
	const uint32_t qrowbytes = (uint32_t) dimp * 4u;
	const int	nc = ix->ncent;
	const int	H = !ix->f16 ? 0 : (ix->f16_sub ? 1 : 2);
	/* records a query may leave: the option's value, less for batches whose record array would pass 1 GiB */
	const uint32_t ecap = std::max<uint32_t>(std::min<uint32_t>(g_s16_ecap, (uint32_t) ((((size_t) 1 << 30) / 8) / (size_t) nq)), 64u);

	const bool	cen = ivf_s16_centered(ix, R);
	const bool	cosb = cen && R == R_IVF_COS;	/* cosine on the centred sweep: normalised planes, thresholds in their squared-L2 domain */
	/* seeds by the exact-arithmetic kernels (k_s16_seed / k_s16_seed_sub with the centred thresholds) instead of
	 * k_s16c_seed, which reads float4 rows: cosine (the reference's cosine values), fp16 mirrors */
	const bool	ipc = cen && R == R_IVF_IP;		/* inner product on the centred sweep: the L2 planes, bounds and thresholds in b's domain */
	/* (inner product and fp16 mirrors take k_s16c_seed's wave-wide sums too: its IP / H16 forms) */
	const bool	xseed = cosb;
	/* A sharded search exchanges thresholds between the seeds and the sweep (g_thr_hook): a collective.  A rank that
	 * fails before it gets there (an allocation, the preparation) must still take part — with +inf, the identity of the
	 * minimum — or its peers wait for it forever; it reports its error afterwards. */
	struct ThrJoin
	{
		ndbhip_ivf *ix; int nq; bool done;
		~ThrJoin()
		{
			if (done || !g_thr_hook)
				return;
			float	   *buf = nullptr;
			bool		own = false;

			if (ix->w_qthr && ix->w_qthr_n >= (size_t) nq)
				buf = (float *) ix->w_qthr;
			else if (hipMalloc((void **) &buf, (size_t) 2 * nq * sizeof(float)) == hipSuccess)
				own = true;
			if (buf)
			{
				(void) hipMemsetD32Async((hipDeviceptr_t) buf, 0x7F800000, (size_t) 2 * nq, g.stream);
				(void) g_thr_hook(buf, (size_t) 2 * nq);
				if (own)
				{
					(void) hipStreamSynchronize(g.stream);
					(void) hipFree(buf);
				}
			}
		}
	} thr_join = {ix, nq, false};
	{
		const int	rc = ivf_s16_prepare(ix, R);

		if (rc)
			return rc;
	}
	if (ipc)
	{
		if (!ix->ipc_valid && ivf_frozen(ix))
			return fail(NDBHIP_ERR_STATE, "the mirror is shared (ndbhip_ivf_share): run an inner-product batch on the source before sharing");
		if (!ix->ipc_valid)
		{
			/* once per version of the mirror / layout of the planes: M^2 and every plane row's M^2 - |x|^2 */
			const size_t npp = (size_t) ix->s16_prow.back() + 8 * 32;
			float	   *x2 = nullptr;

			if (grow(ix->d_ipc_m2, ix->d_ipc_m2_n, (size_t) 1)) return NDBHIP_ERR_HIP;
			if (grow(ix->d_rnx, ix->d_rnx_n, npp)) return NDBHIP_ERR_HIP;
			if (big_alloc((void **) &x2, (size_t) ix->nrows * sizeof(float))) return NDBHIP_ERR_HIP;
			HIP_TRY(hipMemsetAsync(ix->d_ipc_m2, 0, sizeof(uint32_t), g.stream));
			HIP_TRY(hipMemsetAsync(ix->d_rnx, 0, npp * sizeof(float), g.stream));
			if (ix->f16)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ipc_norms<true>), dim3((unsigned) ((ix->nrows + 3) / 4)), dim3(256), 0, g.stream,
								   (const void *) ix->d_vecs, ix->nrows, dim, x2, ix->d_ipc_m2);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ipc_norms<false>), dim3((unsigned) ((ix->nrows + 3) / 4)), dim3(256), 0, g.stream,
								   (const void *) ix->d_vecs, ix->nrows, dim, x2, ix->d_ipc_m2);
			hipLaunchKernelGGL(k_ipc_fill, dim3((unsigned) ((ix->s16_prow.back() + 255) / 256)), dim3(256), 0, g.stream, (const float *) x2,
							   (const uint32_t *) ix->d_ipc_m2, (const int64_t *) ix->d_prow_off, (int) ix->s16_prow.size() - 1,
							   (const uint32_t *) ix->d_bucket_list, (const int64_t *) d.loc_off, d.own_len,
							   (const uint32_t *) ix->d_pposof, ix->d_rnx);
			{
				const int	nbk = (int) ix->s16_prow.size() - 1;

				if (grow(ix->d_bkt_rnxmax, ix->d_bkt_rnxmax_n, (size_t) std::max(nbk, 1))) return NDBHIP_ERR_HIP;
				hipLaunchKernelGGL(k_ipc_bucket_max, dim3((unsigned) std::max(nbk, 1)), dim3(64), 0, g.stream, (const float *) ix->d_rnx,
								   (const int64_t *) ix->d_prow_off, ix->s16_sub ? (const uint32_t *) ix->d_sub_len : (const uint32_t *) ix->d_plen,
								   nbk, ix->d_bkt_rnxmax);
			}
			HIP_TRY(hipMemcpyAsync(&ix->ipc_m2, ix->d_ipc_m2, sizeof(float), hipMemcpyDeviceToHost, g.stream));
			HIP_TRY(hipStreamSynchronize(g.stream));
			big_free(x2);
			ix->ipc_valid = true;
		}
		if (grow(ix->w_qev, ix->w_qev_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		hipLaunchKernelGGL(k_ipc_qev, dim3((nq + 255) / 256), dim3(256), 0, g.stream, (const float *) ix->w_qn2,
						   (const uint32_t *) ix->d_ipc_m2, dim, (uint32_t) nq, ix->w_qev);
	}
	if (grow(ix->w_qplanes, ix->w_qplanes_n, (size_t) nq * qrowbytes)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_qn2, ix->w_qn2_n, (size_t) nq)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_qexp, ix->w_qexp_n, (size_t) nq)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_qthr, ix->w_qthr_n, (size_t) nq)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_ecount, ix->w_ecount_n, (size_t) 3 * nq + 8)) return NDBHIP_ERR_HIP;	/* emitted | survivors | active | flags (4) | pairs, buckets with pairs */
	if (grow(ix->w_erec, ix->w_erec_n, (size_t) nq * ecap)) return NDBHIP_ERR_HIP;
	if (cen && grow(ix->w_eub, ix->w_eub_n, (size_t) nq * ecap)) return NDBHIP_ERR_HIP;
	unsigned int *ecount = ix->w_ecount, *surv = ix->w_ecount + nq, *flags = ix->w_ecount + 3 * (size_t) nq;

	if (grow(ix->w_bmin, ix->w_bmin_n, (size_t) nq * S16_NB)) return NDBHIP_ERR_HIP;
	HIP_TRY(hipMemsetAsync(ix->w_ecount, 0, ((size_t) 3 * nq + 8) * sizeof(unsigned int), g.stream));
	HIP_TRY(hipMemsetAsync(ix->w_bmin, 0xFF, (size_t) nq * S16_NB * sizeof(uint32_t), g.stream));
	/* (the queries' planes, norms and exponents — k_s16_qprep — are the caller's: ivf_search_chunk, which may
	 * already have needed them for the centroid scan) */
#define S16_BY_RH(KERNEL, ...)                                                                  \
	do {                                                                                        \
		if (R == R_IVF_COS)                                                                     \
		{                                                                                       \
			if (H == 0) KERNEL(R_IVF_COS, 0, __VA_ARGS__);                                       \
			else if (H == 1) KERNEL(R_IVF_COS, 1, __VA_ARGS__);                                  \
			else KERNEL(R_IVF_COS, 2, __VA_ARGS__);                                              \
		}                                                                                       \
		else if (R == R_IVF_IP)                                                                 \
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
	/* (with regrouped planes and L2 the seeds come from the nearest sublist instead: k_s16_seed_sub, below) */
	const bool	seed_by_sublist = ix->s16_sub && (R == R_IVF_L2 || (R == R_IVF_IP && cdist) || R == R_IVF_COS) && g_s16_prune && ix->nsub_g > 0;

	if (k > NDB_TOPK_FAST_MAXK && !(cen && seed_by_sublist && (R == R_IVF_L2 || cosb || ipc)))
	{
		/* 64 < k: thresholds come from the sublists' radii (k_s16c_thr_radius) — a layout without sublists, or not centred,
		 * has nothing to take them from: this mirror's batches with k > 64 go to the fp32 screen, this one included */
		ix->s16_bigk_off = true;
		NDB_STAT_ADD(screen16_fallbacks, 1);
		return 1;
	}
	/* (centred path: upper bounds summed by the whole wave instead of the reference's chain per lane: k_s16c_seed) */
	const uint32_t cseeds = g_s16c_seeds ? (uint32_t) g_s16c_seeds : (k <= 20 ? 32u : 64u);

#define S16C_SEED_L(SUBB, PLL, ...)                                                                                     \
	do {                                                                                                                \
		if (ipc && H == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_seed<SUBB, true, 1, PLL>), dim3(nq), dim3(S16C_SEED_THREADS), 0, g.stream, __VA_ARGS__); \
		else if (ipc && H == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_seed<SUBB, true, 2, PLL>), dim3(nq), dim3(S16C_SEED_THREADS), 0, g.stream, __VA_ARGS__); \
		else if (ipc) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_seed<SUBB, true, 0, PLL>), dim3(nq), dim3(S16C_SEED_THREADS), 0, g.stream, __VA_ARGS__);     \
		else if (H == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_seed<SUBB, false, 1, PLL>), dim3(nq), dim3(S16C_SEED_THREADS), 0, g.stream, __VA_ARGS__); \
		else if (H == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_seed<SUBB, false, 2, PLL>), dim3(nq), dim3(S16C_SEED_THREADS), 0, g.stream, __VA_ARGS__); \
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_seed<SUBB, false, 0, PLL>), dim3(nq), dim3(S16C_SEED_THREADS), 0, g.stream, __VA_ARGS__);             \
	} while (0)
	if (!seed_by_sublist && cen && !xseed)
		S16C_SEED_L(false, false, d, d_q, w_probes, lco, npr,
					(uint32_t) k, cseeds, (const uint32_t *) nullptr, (const int *) nullptr, (const uint32_t *) nullptr,
					(const int64_t *) nullptr, (const uint32_t *) nullptr, (const float *) nullptr,
					0u, (const float *) nullptr, (const float *) nullptr, 0u, ix->w_qthr,
					(const float *) ix->w_qn2, (const uint32_t *) ix->d_ipc_m2, (const float *) nullptr, (const float *) nullptr);
	else if (!seed_by_sublist)
		S16_BY_RH(S16_SEED_L, d, d_q, w_probes, lco, npr, (uint32_t) k, (const float *) ix->w_qn2,
				  ipc ? (const uint32_t *) ix->d_ipc_m2 : (const uint32_t *) ix->d_xmax16, (int) (ix->f16 && ix->f16_sub), ix->w_qthr,
				  ipc ? 2 : (cen ? 1 : 0));
	/* a table without cluster structure (what the previous batch's pairs per bucket say, as for the tile size below):
	 * thresholds from a sample of the mirror's rows (k_s16c_seed_sample) on top of the seeds' */
	/* (float4 mirrors: k_seed_gather copies fp32 rows) */
	if (!seed_by_sublist && cen && !xseed && !ix->f16 && R == R_IVF_L2 && !ix->s16_sub && g_s16c_sample > 0 && k <= 64 && npr <= 512 &&
		ix->nrows >= 16 * (int64_t) g_s16c_sample &&
		(g_s16c_qb == 8 || (g_s16c_qb == 0 && ix->s16c_density >= (float) g_s16c_dense_min)))
	{
		const uint32_t ns = (uint32_t) g_s16c_sample, sstr = (ns + 63u) & ~63u;

		if ((!ix->seed_valid || ix->seed_n != (int) ns) && ivf_frozen(ix))
			return fail(NDBHIP_ERR_STATE, "the mirror is shared (ndbhip_ivf_share): run a batch on the source before sharing (its row sample is not there yet)");
		if (!ix->seed_valid || ix->seed_n != (int) ns)
		{
			if (grow(ix->d_seedrows, ix->d_seedrows_n, (size_t) ns * dim)) return NDBHIP_ERR_HIP;
			if (grow(ix->d_seed_list, ix->d_seed_list_n, (size_t) ns)) return NDBHIP_ERR_HIP;
			if (grow(ix->d_seed_pos, ix->d_seed_pos_n, (size_t) ns)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_seed_gather, dim3(ns), dim3(256), 0, g.stream, (const float *) ix->d_vecs, (int64_t) ix->nrows, dim,
							   d.loc_off, nc, ns, ix->d_seedrows, ix->d_seed_list, ix->d_seed_pos);
			const int	rc = s16mat_prepare(ix->dm_seed, ix->d_seedrows, (int) ns, dim);

			if (rc)
				return rc;
			ix->seed_valid = true;
			ix->seed_n = (int) ns;
		}
		if (grow(ix->w_seedmat, ix->w_seedmat_n, (size_t) nq * sstr)) return NDBHIP_ERR_HIP;
		const int	rc = s16mat_run(ix->dm_seed, dim, (const unsigned char *) ix->w_qplanes, ix->w_qn2, ix->w_qexp, ix->w_qthr, nq,
									ix->w_seedmat, sstr);

		if (rc)
			return rc;
		hipLaunchKernelGGL(k_s16c_seed_sample, dim3(nq), dim3(256), 0, g.stream, (const float *) ix->w_seedmat, sstr, ns,
						   (const int *) ix->d_seed_list, (const uint32_t *) ix->d_seed_pos, w_probes, lco, npr, (uint32_t) k,
						   (const float *) ix->w_qn2, (const uint32_t *) ix->dm_seed.xmax, dim, ix->w_qthr);
	}
	if (g_thr_hook && !seed_by_sublist)
	{
		/* sharded search: the smallest threshold any rank found for a query serves all of them */
		thr_join.done = true;
		const int	rc = ivf_exchange_thresholds(ix->w_qthr, nq);

		if (rc)
			return rc;
	}

	/* the (query, probe) pairs bucketed by list — by sublist when the planes are regrouped (`ncs` buckets) —; items
	 * of 128 rows x 128 queries */
	const bool	sub = ix->s16_sub;
	const int	ncs = sub ? ix->nsub : nc;
	const int	ncmp0 = std::min(ix->nlists, ix->ncent);
	const size_t dup0 = npr > ncmp0 ? (size_t) (npr - ncmp0 + 1) : 1;
	/* a (query, probe) pair expands to at most the sublists of its list: per query, npr buckets plus the extra
	 * sublists of the regrouped lists — dup0 times over, because list 0 can be probed that often by one query
	 * (ivf_am.c:1978: the probe slots beyond nlists all read list 0) */
	const size_t pairs_cap = (size_t) nq * ((size_t) npr + dup0 * (size_t) (ncs - nc));

	if (sub)
	{
		if (grow(ix->w_gcnt, ix->w_gcnt_n, (size_t) 2 * ncs + 8 * NDB_QHEAD_STRIDE + 48 + (size_t) 8 * ((ncs + 31) & ~31))) return NDBHIP_ERR_HIP;
		if (grow(ix->w_goff, ix->w_goff_n, (size_t) 3 * (ncs + 1) + 80)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_pairs, ix->w_pairs_n, pairs_cap)) return NDBHIP_ERR_HIP;
	}
	uint32_t   *cnt = ix->w_gcnt, *fill = ix->w_gcnt + ncs;
	unsigned int *next_item = ix->w_gcnt + 2 * ncs;
	/* [8][ncsx]: k_sub_pairs counts per XCD; every XCD's row starts on a 128-byte line and is whole lines long, so no line
	 * is ever dirty in two L2s */
	const int	ncsx = (ncs + 31) & ~31;
	uint32_t   *cntx = ix->w_gcnt + (((size_t) 2 * ncs + 8 * NDB_QHEAD_STRIDE + 16 + 31) & ~(size_t) 31);
	uint32_t   *pair_off = ix->w_goff, *item_off = ix->w_goff + (ncs + 1), *grp_off = ix->w_goff + 2 * (ncs + 1);
	uint32_t   *runs = ix->w_goff + 3 * (ncs + 1) + 32;
	const uint32_t npairs = (uint32_t) nq * (uint32_t) npr;
	ScanTimer	t;
	/* the sweep sees the sublists as its lists */
	IvfDev		ds = d;
	uint32_t	sstride = 0;
	/* distances of the batch's queries to the regrouped lists' centres: the centroid scan's matrix when it covered them
	 * (ix->bat_subdist), else a MODE 3 run of their own */
	const float *subdist = ix->w_subdist, *sub_rn2 = ix->dm_sub.rn2;
	const uint32_t *sub_xmax = ix->dm_sub.xmax;

	if (sub)
	{
		ds.loc_off = ix->d_sub_loc;
		ds.own_len = ix->d_sub_len;
		ds.ncent = ncs;
	}
	else if (cen)
		ds.own_len = ix->d_plen;	/* the lists' rows in the planes, holes of deleted rows included */

	/* tile geometry: 8 waves, 256 rows x 128 queries, ring of 3 chunk buffers, one block per CU (default), or
	 * 4 waves, 128 x 128, ring of 2, two blocks per CU (ndbhip_set_option("screen16_waves", 4)) */
	/* the centred sweep's tile holds 128 pairs; 32 where the buckets are probed by a handful of queries each; 256 pairs
	 * x 256 rows where they are probed by hundreds and are whole lists (what the previous batch on this mirror looked
	 * like; before any: regrouped planes mean clustered rows, i.e. few) */
	const int	c_qb = !cen ? 4 : (g_s16c_qb == 1 || g_s16c_qb == 4 || g_s16c_qb == 8) ? g_s16c_qb :
		(ix->s16c_density >= 0.0f ? (ix->s16c_density < 24.0f ? 1 :
									(dimp / S16C_CH >= 2 && (sub ? (g_s16c_dense_min_sub > 0 && ix->s16c_density >= (float) g_s16c_dense_min_sub)
											 : ix->s16c_density >= (float) g_s16c_dense_min)) ? 8 : 4)
		 : (ix->s16_sub ? 1 : 4));
	/* the register-streaming sweep (ndbhip_screen16w.h: sparse pair tables) takes items of ONE 32-row block — a wave each,
	 * dealt round-robin: the sweep is as long as its busiest wave, and tiles of 128 rows over sublists of 40 to 200
	 * left waves with 6 to 20 items (profiles/r05_wave_trace.txt) */
	const int	nchk_w = dimp / S16C_CH;
	/* (buckets of 16 blocks on average at most: a pair has a word per block of its bucket, and whole lists of thousands of
	 * rows that happen to be probed by few queries — a small batch on an unclustered table — are the LDS ring's) */
	/* (and batches of a thousand queries at least: below that the ring is as fast — 0.156 against 0.163-0.177 ms at C5's
	 * 256 — and needs no pair lists and no collect behind it) */
	const int	wd = (!cen || c_qb != 1 || !g_s16c_wave || !g_s16c_epi || g_s16_debug != 0 || nchk_w < 2 || ix->s16w_off || ix->s16w_avgblk > 16 ||
					  nq < g_s16c_wave_min_nq) ? 0
		: std::min(g_s16c_wave, nchk_w);
	const int	s16_rt = cen ? (wd ? 32 : (c_qb == 8 ? 256 : 128)) : (g_s16_waves == 8 ? 256 : 128);
	const uint32_t s16_qt = (uint32_t) (32 * c_qb);
	/* rows of the pair planes: every pair there can be, up to qc_mult x (queries x probes) (at least 65 536; qc_mult starts at 4 and doubles, up to 16, after a batch that did not fit) — sublists
	 * multiply the pairs of a probed list, the exclusion bounds remove most again; a batch with more than that goes to
	 * the older path (flags[2]) */
	const uint32_t qc_cap = (uint32_t) std::min<size_t>(std::min<size_t>(pairs_cap, std::max<size_t>((size_t) ix->qc_mult * nq * npr + 1024, (size_t) 1 << 16)),
														   0x7FFFFFFFu);
	const uint32_t qcrowbytes = (uint32_t) dimp * 2u;
	/* the dense tile's kernel (ndbhip_screen16d.h) reads chunk-major pair planes: [64-dim chunk][pair][64 halfs] */
	const bool	dense_k = cen && c_qb == 8 && g_s16c_dense && !ipc;	/* (inner product: k_s16c_sweep<8, 2>, which takes the rows' constants) */
	const size_t qc_plane = ((size_t) qc_cap + 256) * 64;		/* halfs per chunk plane */

	if (cen)
	{
		/* (+ 256 rows: the dense sweep's loaders fetch whole 8-row pieces of a tile's last, partly filled pair block) */
		if (grow(ix->w_qcplanes, ix->w_qcplanes_n, ((size_t) qc_cap + 256) * dimp)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_qcn2, ix->w_qcn2_n, (size_t) qc_cap)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_qcexp, ix->w_qcexp_n, (size_t) qc_cap)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_pslot, ix->w_pslot_n, (size_t) 5 * qc_cap)) return NDBHIP_ERR_HIP;		/* query, first position, visible rows | first word, bucket */
	}
#define S16_SWEEP_L(RR, HH, ...)                                                                                  \
	do {                                                                                                          \
		if (g_s16_debug == 1 && HH == 0 && RR == R_IVF_L2)                                                         \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_L2, 0, 4, 2, 1>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, __VA_ARGS__); \
		else if (g_s16_debug == 2 && HH == 0 && RR == R_IVF_L2)                                                    \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_L2, 0, 4, 2, 2>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, __VA_ARGS__); \
		else if (RR == R_IVF_COS)		/* (the inner product of the normalised planes, which are float4-style whatever the mirror holds) */ \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<R_IVF_IP, 0, 4, 2>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, __VA_ARGS__); \
		else if (g_s16_waves == 8)                                                                                \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<(RR == R_IVF_COS ? R_IVF_IP : RR), HH, 8, 3>), dim3(g.num_cus), dim3(512), 0, g.stream, __VA_ARGS__); \
		else                                                                                                      \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_sweep<(RR == R_IVF_COS ? R_IVF_IP : RR), HH, 4, 2>), dim3(g.num_cus * 2), dim3(256), 0, g.stream, __VA_ARGS__); \
	} while (0)
	/*
	 * Round 0 sweeps every (query, probe) pair against the seed threshold.  A query that emits more than its
	 * record capacity (its nearest list is huge, or the data has no cluster structure) keeps the first `ecap`
	 * records — any subset is valid evidence — and k_s16_retarget turns their k-th smallest into a far tighter
	 * threshold (a sample of 2048 values below the seed threshold puts its 10th smallest ~200x deeper); round 1
	 * sweeps those queries alone against it.  Whatever still overflows after that sends the batch to the older path.
	 */
	unsigned int *active = ix->w_ecount + 2 * (size_t) nq;

	/* a shard's finalize decides with the k-th LOCAL bound, looser than the whole index's: four times the room */
	/* (k > 64: the cut at the k-th upper bound leaves k survivors and those inside the bounds' error) */
	const uint32_t surv_cap = std::max(partial ? 4u * S16_SURV_CAP : (uint32_t) S16_SURV_CAP, k > NDB_TOPK_FAST_MAXK ? std::min(6u * (uint32_t) k, 1280u) : 0u);
	/* the survivors' rows through LDS (s16_exact_staged) for a small batch, which has a CU to itself: 8 chunks deep for up to
	 * 16 survivors, 64-row slots for more.  A large batch lives on 16 blocks per CU overlapping each other's phases, and the
	 * ring's LDS would halve them (measured at 4096 queries: 128 us with a 10 KB ring, 103 us without) */
	const size_t fsmem0 = (topk_smem_bytes(surv_cap, (uint32_t) k) + 15) & ~(size_t) 15;
	const size_t fin_room = fsmem0 < 65536 ? 65536 - fsmem0 : 0;	/* (within the 64 KB a kernel gets unasked) */
	const bool	fin_ok = g_s16_stage && s16_staged_ok(ix->dim, ix->f16 ? 1 : 0);
	const bool	fin_small = nq <= 512;
	int			fin_nbuf = (fin_ok && (fin_small || g_s16_stage >= 2)) ? s16_staged_nbuf(1, g_s16_stage >= 2 ? g_s16_stage : 8) : 0;
	int			fin_nbuf4 = (fin_ok && fin_small) ? s16_staged_nbuf(4, g_s16_stage >= 2 ? g_s16_stage : 3) : 0;

	fin_nbuf = std::min(fin_nbuf, (int) (fin_room / S16X_SLOT(1)));
	fin_nbuf4 = std::min(fin_nbuf4, (int) (fin_room / S16X_SLOT(4)));
	if (fin_nbuf < 2) fin_nbuf = 0;
	if (fin_nbuf4 < 2) fin_nbuf4 = 0;
	const size_t fsmem = fsmem0 + std::max((size_t) fin_nbuf * S16X_SLOT(1), (size_t) fin_nbuf4 * S16X_SLOT(4));
	unsigned int over = 0, over_n = 0, fl8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	bool		over_pairs = false;

	if (grow(ix->w_overq, ix->w_overq_n, (size_t) S16_OVER_CAP)) return NDBHIP_ERR_HIP;
	if (ix->s16_sub)
	{
		if (grow(ix->w_qpairs, ix->w_qpairs_n, (size_t) nq * S16_QP_CAP)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_qpn, ix->w_qpn_n, (size_t) nq)) return NDBHIP_ERR_HIP;
	}

	for (int round = 0; round < 2; round++)
	{
		const unsigned int *act = round ? active : (const unsigned int *) nullptr;
		uint32_t	desc_cap = 0;

		if (round)
		{
			/* some query overflowed its records in round 0 (the host has just read the flag): it alone is swept again,
			 * against the threshold its first records give — rare enough that its nine launches are not worth
			 * queueing for every batch */
			HIP_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(unsigned int), g.stream));
			if (cosb)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_retarget<R_IVF_COS>), dim3(nq), dim3(S16_NB), 0, g.stream, dim, (uint32_t) k,
								   ix->w_qthr, ecount, ecap, (const uint32_t *) ix->w_bmin, active, flags + 1, 1);
			else if (R == R_IVF_IP || R == R_IVF_COS)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_retarget<R_IVF_IP>), dim3(nq), dim3(S16_NB), 0, g.stream, dim, (uint32_t) k,
								   ix->w_qthr, ecount, ecap, (const uint32_t *) ix->w_bmin, active, flags + 1, ipc ? 1 : 0,
								   ipc ? (const float *) ix->w_qev : (const float *) nullptr);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_retarget<R_IVF_L2>), dim3(nq), dim3(S16_NB), 0, g.stream, dim, (uint32_t) k,
								   ix->w_qthr, ecount, ecap, (const uint32_t *) ix->w_bmin, active, flags + 1, cen ? 1 : 0);
		}
		HIP_TRY(hipMemsetAsync(ix->w_gcnt, 0, (size_t) (2 * ncs + 8 * NDB_QHEAD_STRIDE + (sub ? 48 + 8 * ncsx : 16)) * sizeof(uint32_t), g.stream));	/* + k_sub_pairs' overflow flag and per-XCD counts */
		const uint8_t *drop = nullptr;
		const float *pdist = nullptr;

		/* inner product: sublists are excluded by -(q.c) - |q| rad (k_sub_pairs); that needs the centroid scan's
		 * distances of this call */
		const bool	prune = g_s16_prune && (R == R_IVF_L2 || (R == R_IVF_IP && sub && cdist) || (R == R_IVF_COS && sub));
		const int	ipb = R == R_IVF_IP ? 1 : (R == R_IVF_COS ? (cosb ? 3 : 2) : 0);

		if (prune && !(sub && cdist) && R == R_IVF_L2)
		{
			/* (query, list) pairs whose every row lies beyond the query's current threshold: |q - c| - radius
			 * (with sublists and the centroid scan's distances at hand, k_sub_pairs applies the same test itself) */
			if (grow(ix->w_drop, ix->w_drop_n, (size_t) npairs)) return NDBHIP_ERR_HIP;
			if (sub && grow(ix->w_pdist, ix->w_pdist_n, (size_t) npairs)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_s16_pair_prune, dim3(nq), dim3(256), 0, g.stream, d_q, (uint32_t) nq, npr, dim,
							   w_probes, nc, (const float *) ix->d_centroids, (const uint32_t *) ix->d_lrad,
							   (const float2 *) ix->w_qthr, act, ix->w_drop, sub ? ix->w_pdist : (float *) nullptr);
			drop = ix->w_drop;
			pdist = sub ? ix->w_pdist : nullptr;
		}
		if (round == 0 && !sub)
			hipLaunchKernelGGL(k_s16_prune_stats, dim3((nq + 255) / 256), dim3(256), 0, g.stream, drop,
							   lco, (uint32_t) nq, npr, g.d_counters + 5, 1);
		if (sub)
		{
			/* distances of every query to the centres of the regrouped lists (once per batch), then the expansion */
			if (round == 0 && prune && ix->nsub_g > 0)
			{
				if (ix->bat_restrict && cdist)
				{
					/* the centroid scan multiplied the centroids alone: the centres of the probed lists, list by list */
					const int	rc = ivf_s16_sub_distances_probed(ix, d_q, nq, w_probes, npr, &sstride);

					if (rc)
						return rc;
					NDB_STAT_ADD(sub_restricted, 1);
					subdist = ix->w_subdist;
					sub_xmax = ix->dm_all.xmax;
					sub_rn2 = ix->dm_all.rn2 + ix->dm_all_ncmp;
				}
				else if (ix->bat_subdist && cdist)
				{
					/* the centroid scan of this call multiplied the centres along with the centroids */
					subdist = ix->bat_subdist;
					sstride = ix->bat_sstride;
					sub_xmax = ix->dm_all.xmax;
					sub_rn2 = ix->dm_all.rn2 + ix->dm_all_ncmp;
				}
				else
				{
					const int	rc = ivf_s16_sub_distances(ix, d_q, nq, &sstride);

					if (rc)
						return rc;
					subdist = ix->w_subdist;
					sub_xmax = ix->dm_sub.xmax;
					sub_rn2 = ix->dm_sub.rn2;
				}
				/* ... which also say where the query's own neighbourhood is: seeds from the nearest sublist */
				if (cen && !xseed)
					{
#define S16C_SEED_SUB_ARGS d, d_q, w_probes, lco, npr,                                                               \
								(uint32_t) k, cseeds, (const uint32_t *) ix->d_sub_first, (const int *) ix->d_sub_gidx,       \
								(const uint32_t *) ix->d_sub_len, (const int64_t *) ix->d_prow_off,                          \
								(const uint32_t *) ix->d_pposof,                                                            \
								subdist, sstride, pdist, cdist, cstride, ix->w_qthr,                                         \
								(const float *) ix->w_qn2, (const uint32_t *) ix->d_ipc_m2, sub_rn2, (const float *) ix->d_cn2, \
								(const unsigned char *) ix->d_planes, (const uint32_t *) ix->d_sub_blk, (const float *) ix->d_rn2, \
								(const int16_t *) ix->d_rexp, (const float *const *) ix->d_sub_cptr, ndb_s16c_ce(dim),             \
								ipc ? (const float *) ix->d_rnx : (const float *) nullptr
						/* (L2 and inner product on the centred planes: the seeds' bounds from block 0 of the nearest sublist's planes) */
						if (g_s16c_plseed && !cosb)
							S16C_SEED_L(true, true, S16C_SEED_SUB_ARGS);
						else
							S16C_SEED_L(true, false, S16C_SEED_SUB_ARGS);
#undef S16C_SEED_SUB_ARGS
					}
				else
				{
#define S16_SEEDSUB_L(RR, HH, ...) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_seed_sub<RR, HH>), dim3(nq), dim3(64), 0, g.stream, __VA_ARGS__)
					S16_BY_RH(S16_SEEDSUB_L, d, d_q, w_probes, lco,
							  npr, (uint32_t) k, (const uint32_t *) ix->d_sub_first, (const int *) ix->d_sub_gidx,
							  (const uint32_t *) ix->d_sub_len, (const int64_t *) ix->d_sub_loc,
							  (const int64_t *) ix->d_perm, (const uint32_t *) ix->d_posof,
							  subdist, sstride, pdist, cdist, cstride, (const float *) ix->w_qn2,
							  ipc ? (const uint32_t *) ix->d_ipc_m2 : (const uint32_t *) ix->d_xmax16, ix->w_qthr, ipc ? 2 : (xseed ? 1 : 0),
							  ipb ? sub_rn2 : (const float *) nullptr,
							  ipb ? (const float *) ix->d_cn2 : (const float *) nullptr, H == 1 ? 1 : 0,
							  /* (the two-plane sweep pays for a looser threshold with emissions: measured 2.66 -> 2.59 M q/s at 32) */
							  xseed ? cseeds : (uint32_t) S16_SEED);
				}
				/* k > 64: no seed kernel holds k rows; the threshold comes from the buckets' radii (L2, and cosine in the
				 * normalised rows' space) */
				if (k > NDB_TOPK_FAST_MAXK && cen)
					hipLaunchKernelGGL(k_s16c_thr_radius, dim3(nq), dim3(256), 0, g.stream, w_probes, lco, npr, (uint32_t) k,
									   (const uint32_t *) ix->d_sub_first, (const int *) ix->d_sub_gidx, (const uint32_t *) ix->d_sub_len,
									   (const uint32_t *) ix->d_sub_rad, (const int64_t *) ix->d_prow_off, (const uint32_t *) ix->d_pposof,
									   subdist, sstride, pdist, cdist, cstride, (const float *) ix->w_qn2, (const uint32_t *) sub_xmax, dim,
									   ix->w_qthr, ipc ? (const float *) ix->d_bkt_rnxmax : (const float *) nullptr,
									   ipc ? (const float *) ix->w_qev : (const float *) nullptr, cosb ? 1 : 0);
				if (g_thr_hook)
				{
					thr_join.done = true;
					const int	rc2 = ivf_exchange_thresholds(ix->w_qthr, nq);

					if (rc2)
						return rc2;
				}
			}
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sub_pairs<0>), dim3(nq), dim3(256), 0, g.stream, w_probes, lco,
							   npr, (uint32_t) nq, (const uint32_t *) ix->d_sub_first, (const int *) ix->d_sub_gidx,
							   (const uint32_t *) ix->d_sub_len, (const uint32_t *) ix->d_sub_rad, subdist,
							   sstride, (const float2 *) ix->w_qthr, prune ? 1 : 0, pdist, cdist, cstride, (const float *) ix->w_qn2,
							   (const uint32_t *) sub_xmax, dim, act, cnt, (const uint32_t *) nullptr, (uint32_t *) nullptr,
							   (PairRec *) nullptr, ipb, sub_rn2, (const float *) ix->d_cn2, ix->w_qpairs, ix->w_qpn, ix->w_gcnt + 2 * ncs + 8 * NDB_QHEAD_STRIDE,
							   g_s16_debug & 28, cntx, (uint32_t) ncsx, ipc ? ix->ipc_m2 : -1.0f);
		}
		else
			hipLaunchKernelGGL(k_pair_count, dim3((npairs + 255) / 256), dim3(256), 0, g.stream, w_probes, lco, npr,
							   (uint32_t) nq, cnt, act, drop);
#ifdef NDB_DEBUG_SUBPAIRS
		if (sub && getenv("NDB_DEBUG_Q") && atoi(getenv("NDB_DEBUG_Q")) < nq)
		{
			/* (diagnostic builds only) what k_sub_pairs saw for one query */
			const int	dq = atoi(getenv("NDB_DEBUG_Q"));
			float2		th;
			float		q2 = 0, xm = 0;
			std::vector<int> pr(npr);
			std::vector<uint32_t> lc(npr + 1), sf(nc + 1);

			HIP_TRY(hipStreamSynchronize(g.stream));
			HIP_TRY(hipMemcpy(&th, ix->w_qthr + dq, 8, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(&q2, ix->w_qn2 + dq, 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(&xm, sub_xmax, 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(pr.data(), w_probes + (size_t) dq * npr, 4 * npr, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(lc.data(), lco + (size_t) dq * (npr + 1), 4 * (npr + 1), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(sf.data(), ix->d_sub_first, 4 * (nc + 1), hipMemcpyDeviceToHost));
			float		ecd = (ndb_s16_cdot(dim) + NDB_S16_NORMS) * (q2 + xm) * 1.00001f;

			ecd = ecd + fabsf(ecd) * 4.8e-7f + 1e-37f + NDB_S16_ABS;
			fprintf(stderr, "DBG q %d: T %.9g  |q|^2 %.9g xmax %.9g ec %.9g prune %d cdist %d subdist %p sstride %u R %d\n", dq, th.x, q2, xm, ecd, (int) prune, cdist ? 1 : 0, (const void *) subdist, sstride, R);
			for (int p = 0; p < npr; p++)
			{
				fprintf(stderr, "DBG  probe %d list %d visible %u\n", p, pr[p], lc[p + 1] - lc[p]);
				if (pr[p] < 0 || pr[p] >= nc)
					continue;
				for (uint32_t s2 = sf[pr[p]]; s2 < sf[pr[p] + 1]; s2++)
				{
					uint32_t	sl = 0, sr = 0;
					int			gi = 0;
					float		a = -1.0f;
					int64_t		po = 0;

					HIP_TRY(hipMemcpy(&sl, ix->d_sub_len + s2, 4, hipMemcpyDeviceToHost));
					HIP_TRY(hipMemcpy(&sr, ix->d_sub_rad + s2, 4, hipMemcpyDeviceToHost));
					HIP_TRY(hipMemcpy(&gi, ix->d_sub_gidx + s2, 4, hipMemcpyDeviceToHost));
					HIP_TRY(hipMemcpy(&po, ix->d_prow_off + s2, 8, hipMemcpyDeviceToHost));
					if (gi >= 0 && subdist)
						HIP_TRY(hipMemcpy(&a, subdist + (size_t) dq * sstride + gi, 4, hipMemcpyDeviceToHost));
					float		rad;
					memcpy(&rad, &sr, 4);
					const double alo = (double) a - (double) ecd * (1.0 + 1e-6);
					const double lb = sqrt(alo > 0.0 ? alo : 0.0) - (double) rad;
					std::vector<uint32_t> pp(sl);
					std::vector<float> cc(dim);
					const float *cp = nullptr;

					HIP_TRY(hipMemcpy(pp.data(), ix->d_pposof + po, 4 * sl, hipMemcpyDeviceToHost));
					HIP_TRY(hipMemcpy(&cp, ix->d_sub_cptr + s2, sizeof(cp), hipMemcpyDeviceToHost));
					HIP_TRY(hipMemcpy(cc.data(), cp, 4 * dim, hipMemcpyDeviceToHost));
					fprintf(stderr, "DBG   sub %u len %u gi %d rad %.9g a %.9g lb^2 %.9g %s  pos:", s2, sl, gi, rad, a, lb > 0 ? lb * lb : 0.0,
							(lb > 0.0 && lb * lb * (1.0 - 1e-9) > (double) th.x) ? "EXCLUDED" : "kept");
					for (uint32_t r = 0; r < sl; r++)
						fprintf(stderr, " %u", pp[r]);
					fprintf(stderr, "\nDBG   centre:");
					for (int i = 0; i < dim; i++)
						fprintf(stderr, " %.9g", cc[i]);
					fprintf(stderr, "\n");
				}
			}
		}
#endif
		const bool	xsum_apart = sub && ncs > 8192;

		if (xsum_apart)
			hipLaunchKernelGGL(k_pair_xsum, dim3((ncs + 255) / 256), dim3(256), 0, g.stream, cntx, (uint32_t) ncsx, ncs, cnt);
		if (xsum_apart && cen)
		{
			const unsigned nblk = (unsigned) ((ncs + 1023) / 1024);

			if (grow(ix->w_ppart, ix->w_ppart_n, (size_t) 4 * nblk)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_pair_part, dim3(nblk), dim3(1024), 0, g.stream, (const uint32_t *) cnt, ds.own_len, ncs,
							   (uint32_t) (s16_qt / NDB_QG), (uint32_t) (s16_rt / 32), ix->w_ppart,
							   (sub && round == 0) ? g.d_counters + 6 : (unsigned long long *) nullptr, wd ? 1 : 0);
			hipLaunchKernelGGL(k_pair_scan, dim3(nblk), dim3(1024), 0, g.stream, (const uint32_t *) cnt, ds.own_len, ncs,
							   (uint32_t) (s16_qt / NDB_QG), (uint32_t) (s16_rt / 32), (const uint32_t *) ix->w_ppart,
							   pair_off, item_off, grp_off, runs, wd ? 1 : 0);
		}
		else
		hipLaunchKernelGGL(k_pair_offsets, dim3(1), dim3(1024), 0, g.stream, cnt, ds.own_len, ncs,
						   pair_off, item_off, grp_off, runs, (uint32_t) (s16_qt / NDB_QG), (uint32_t) (s16_rt / 32), cen ? 1 : 0,
						   (sub && round == 0) ? g.d_counters + 6 : (unsigned long long *) nullptr,
						   (sub && !xsum_apart) ? cntx : (uint32_t *) nullptr, (uint32_t) ncsx, wd ? 1 : 0);
		if (sub)
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sub_pairs<1>), dim3(nq), dim3(256), 0, g.stream, w_probes, lco,
							   npr, (uint32_t) nq, (const uint32_t *) ix->d_sub_first, (const int *) ix->d_sub_gidx,
							   (const uint32_t *) ix->d_sub_len, (const uint32_t *) ix->d_sub_rad, subdist,
							   sstride, (const float2 *) ix->w_qthr, prune ? 1 : 0, pdist, cdist, cstride, (const float *) ix->w_qn2,
							   (const uint32_t *) sub_xmax, dim, act, cnt, (const uint32_t *) pair_off, fill, ix->w_pairs,
							   ipb, sub_rn2, (const float *) ix->d_cn2, ix->w_qpairs, ix->w_qpn, ix->w_gcnt + 2 * ncs + 8 * NDB_QHEAD_STRIDE,
							   0, cntx, (uint32_t) ncsx, ipc ? ix->ipc_m2 : -1.0f);
		else
			hipLaunchKernelGGL(k_pair_fill, dim3((npairs + 255) / 256), dim3(256), 0, g.stream, w_probes, lco, npr,
							   (uint32_t) nq, (const uint32_t *) pair_off, fill, ix->w_pairs, act, drop);
		{
			/* items <= (row tiles) x (query tiles of the fullest list); a query probes a list once — except list 0,
			 * which the reference scans again for every probe slot beyond nlists (ivf_am.c:1978, palloc0) */
			const int	ncmp = std::min(ix->nlists, ix->ncent);
			const size_t dup = npr > ncmp ? (size_t) (npr - ncmp + 1) : 1;
			const size_t cap_items = ((size_t) ix->nrows / (size_t) s16_rt + (size_t) ncs) *
				(((size_t) nq * dup + s16_qt - 1) / s16_qt);

			if (grow(ix->w_s16desc, ix->w_s16desc_n, cap_items * 4)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_s16_items, dim3((unsigned) ((cap_items + 255) / 256)), dim3(256), 0, g.stream,
							   (const uint32_t *) item_off, (const uint32_t *) cnt, ds.own_len, ncs, (uint32_t) s16_rt,
							   (uint32_t) std::min<size_t>(cap_items, 0xFFFFFFFFu), (S16Desc *) ix->w_s16desc, flags, s16_qt,
							   round == 0 ? g.d_counters + 7 : (unsigned long long *) nullptr,
							   (uint32_t) (cen ? (dimp / S16C_CH) * 4096 : (dimp / S16_CH) * (ix->s16_planes_f32 ? 4096 : 2048)));
			desc_cap = (uint32_t) std::min<size_t>(cap_items, 0xFFFFFFFFu);
		}
		if (cen)
		{
			/* q - c of every pair that is left, in the pair tables' order (more pairs than the planes hold: flags[0],
			 * the batch goes to the older path and the sweep below returns at once) */
			hipLaunchKernelGGL(k_s16c_qcprep, dim3(g.num_cus * 8), dim3(256), 0, g.stream, cosb ? (const float *) ix->w_qhat : d_q, dim, dimp,
							   (const PairRec *) ix->w_pairs, (const uint32_t *) pair_off, ncs,
							   cosb ? (const float *) ix->d_cent_hat : (const float *) ix->d_centroids,
							   sub ? (const float *const *) ix->d_sub_cptr : (const float *const *) nullptr,
							   ix->w_qcplanes, dense_k ? qc_plane : (size_t) 0, ix->w_qcn2, ix->w_qcexp, ix->w_pslot, ix->w_pslot + qc_cap, ix->w_pslot + 2 * (size_t) qc_cap,
							   lco, npr, qc_cap, flags + 2, round == 0 ? flags + 4 : (unsigned int *) nullptr,
							   (const uint32_t *) cnt);
		}
		/* the register-streaming sweep leaves its results per (pair, 32-row block) word (ndbhip_screen16w.h): the words'
		 * places (grp_off is their per-bucket start with wmode = 1), and every query's list of its pairs */
		/* at most every (query, probe) pair x the fullest list's blocks; to start with, wc_mult buckets of average size a
		 * (query, probe) pair (C2 needs 1 word a pair, C4 — 19 sublists a list, 2.3 of them kept per probe — 9), then half
		 * as much again as the fullest batch so far needed; four times the start after a batch that did not fit */
		const size_t wbound = (size_t) nq * ((size_t) npr + dup0) * (size_t) ix->s16w_maxlw;
		const size_t wwant = std::max<size_t>(std::max<size_t>((size_t) ix->wc_mult * nq * npr * ix->s16w_avgblk, (size_t) ix->wc_words + ix->wc_words / 2),
											  (size_t) 1 << 16);
		const uint32_t wcap = (uint32_t) std::min<size_t>(std::min<size_t>(wbound, wwant), 0x07FFFFFFu);

		if (wd)
		{
			if (grow(ix->w_wmask, ix->w_wmask_n, (size_t) wcap)) return NDBHIP_ERR_HIP;
			if (grow(ix->w_wrec, ix->w_wrec_n, (size_t) 32 * wcap)) return NDBHIP_ERR_HIP;
			if (grow(ix->w_qslot, ix->w_qslot_n, (size_t) nq * S16_QP_CAP + (size_t) nq)) return NDBHIP_ERR_HIP;
			uint32_t   *qsn = ix->w_qslot + (size_t) nq * S16_QP_CAP;

			HIP_TRY(hipMemsetAsync(qsn, 0, (size_t) nq * sizeof(uint32_t), g.stream));
			hipLaunchKernelGGL(k_s16w_pairinfo, dim3((qc_cap + 255) / 256), dim3(256), 0, g.stream, (const PairRec *) ix->w_pairs,
							   (const uint32_t *) pair_off, ncs, (const uint32_t *) grp_off, (const uint32_t *) ds.own_len, qc_cap, wcap,
							   ix->w_pslot + 3 * (size_t) qc_cap, ix->w_pslot + 4 * (size_t) qc_cap, ix->w_qslot, qsn,
							   (uint32_t) S16_QP_CAP, flags + 2, flags + 6, flags + 7);
		}
		if (g_debug_s16 && round == 0)
		{
			uint32_t	np = 0, ni = 0;

			HIP_TRY(hipStreamSynchronize(g.stream));
			HIP_TRY(hipMemcpy(&np, pair_off + ncs, 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(&ni, item_off + ncs, 4, hipMemcpyDeviceToHost));
			fprintf(stderr, "s16 debug: %u (query, probe, %s) triples in %u items of %d x %d, %d buckets\n", np,
					sub ? "sublist" : "list", ni, s16_rt, S16_QT, ncs);
		}
		SweepTurn	turn;

		if (turn.begin()) return NDBHIP_ERR_HIP;
		if (round == 0 && t.start()) return NDBHIP_ERR_HIP;
		if (cen)
		{
			const float cE = ndb_s16c_ce(dim);
			/* (a ring of 3 looks two chunks ahead, which must not reach past the NEXT item: dims <= 64 are one chunk) */
			const int	nbuf = dimp / S16C_CH < 2 ? 2 : (g_s16c_nbuf ? g_s16c_nbuf : (c_qb == 1 ? 3 : 2));

#define S16C_SWEEP_L(QB, NB, DB) S16C_SWEEP_LP(QB, NB, DB, 1, 0)
#define S16C_SWEEP_LE(QB, NB, DB, EP) S16C_SWEEP_LP(QB, NB, DB, EP, 0)
#define S16C_SWEEP_LP(QB, NB, DB, EP, PFD)                                                                           \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_sweep<QB, NB, DB, EP, PFD>), dim3(g.num_cus * ((QB != 8 && (QB == 1 || NB == 2)) ? 2 : 1)), dim3(QB == 8 ? 512 : 256), 0, g.stream, \
							   dim, ncs, (const int64_t *) ix->d_prow_off, (const uint32_t *) ds.own_len,                     \
							   (const unsigned char *) ix->d_planes, sub ? (const uint32_t *) ix->d_sub_blk : (const uint32_t *) ix->d_blkoff, \
							   (const float *) ix->d_rn2, (const int16_t *) ix->d_rexp, (const unsigned char *) ix->w_qcplanes, qcrowbytes, \
							   (const float *) ix->w_qcn2, (const int *) ix->w_qcexp, (const uint32_t *) ix->w_pslot,            \
							   (const uint32_t *) (ix->w_pslot + qc_cap), (const uint32_t *) (ix->w_pslot + 2 * (size_t) qc_cap), \
							   (float2 *) ix->w_qthr, (const uint32_t *) cnt, (const uint32_t *) pair_off,                    \
							   (const S16Desc *) ix->w_s16desc, (const uint32_t *) runs, ecount, ix->w_erec, ix->w_eub, ecap, \
							   ix->w_bmin, dimp / S16C_CH, desc_cap, g_s16_tighten ? (uint32_t) k : 0u,                       \
							   (const uint32_t *) ix->d_pposof, cE, qc_cap, cosb ? 1 : 0,                                       \
							   ipc ? (const float *) ix->d_rnx : (const float *) nullptr, ipc ? (const float *) ix->w_qev : (const float *) nullptr)
#define S16C_DENSE_L(DB) S16C_DENSE_LS(DB, 4, false)
#define S16C_DENSE_LS(DB, NBLL, SMALLL)                                                                              \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_dense<DB, NBLL, SMALLL>), dim3(dense_grid), dim3(512), 0, g.stream,   \
							   dim, ncs, (const int64_t *) ix->d_prow_off, (const uint32_t *) ds.own_len,                     \
							   (const unsigned char *) ix->d_planes, sub ? (const uint32_t *) ix->d_sub_blk : (const uint32_t *) ix->d_blkoff, \
							   (const float *) ix->d_rn2, (const int16_t *) ix->d_rexp, (const unsigned char *) ix->w_qcplanes, qc_plane * 2, \
							   (const float *) ix->w_qcn2, (const int *) ix->w_qcexp, (const uint32_t *) ix->w_pslot,            \
							   (const uint32_t *) (ix->w_pslot + qc_cap), (const uint32_t *) (ix->w_pslot + 2 * (size_t) qc_cap), \
							   (float2 *) ix->w_qthr, (const uint32_t *) cnt, (const uint32_t *) pair_off,                    \
							   (const S16Desc *) ix->w_s16desc, (const uint32_t *) runs, ecount, ix->w_erec, ix->w_eub, ecap, \
							   ix->w_bmin, dimp / S16C_CH, desc_cap, g_s16_tighten ? (uint32_t) k : 0u,                       \
							   (const uint32_t *) ix->d_pposof, cE, qc_cap, cosb ? 1 : 0, g_s16c_pfd, g_s16c_rot, (uint32_t) g_s16c_tight,   \
							   next_item, dense_sync)
			/* chunks in flight per wave of the register-streaming sweep: the option's value if it divides the item's chunks */
#define S16C_WSWEEP_L(DD, IPXX, BLKK)                                                                                   \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16c_wsweep<DD, IPXX, BLKK>), dim3(g.num_cus * g_s16c_wblk), dim3(256), 0, g.stream, \
							   dim, ncs, (const int64_t *) ix->d_prow_off, (const uint32_t *) ds.own_len,                     \
							   (const unsigned char *) ix->d_planes, sub ? (const uint32_t *) ix->d_sub_blk : (const uint32_t *) ix->d_blkoff, \
							   (const float *) ix->d_rn2, (const int16_t *) ix->d_rexp, (const unsigned char *) ix->w_qcplanes, qcrowbytes, \
							   (const float *) ix->w_qcn2, (const int *) ix->w_qcexp, (const uint32_t *) ix->w_pslot,            \
							   (const uint32_t *) (ix->w_pslot + 3 * (size_t) qc_cap), (const uint32_t *) (ix->w_pslot + 2 * (size_t) qc_cap), \
							   (const float2 *) ix->w_qthr, (const uint32_t *) cnt, (const uint32_t *) pair_off,               \
							   (const S16Desc *) ix->w_s16desc, (const uint32_t *) runs, ix->w_wmask, ix->w_wrec, wcap,        \
							   (const uint32_t *) (grp_off + ncs), dimp / S16C_CH, desc_cap, (const uint32_t *) ix->d_pposof, cE, qc_cap, \
							   ipc ? (const float *) ix->d_rnx : (const float *) nullptr, next_item)
#define S16C_WSWEEP_D(DD, BLKK) do { if (ipc) S16C_WSWEEP_L(DD, true, BLKK); else S16C_WSWEEP_L(DD, false, BLKK); } while (0)
			const unsigned dense_grid = (unsigned) std::max(8, g.num_cus - (ivf_frozen(ix) ? g_s16c_dense_spare : 0));

			/* the XCD's blocks meet only where the items are alike — whole lists probed by hundreds of queries, every tile
			 * full: the i.i.d. table; on the middle of the sigma sweep (128 pairs a list, tiles of every size) the meetings
			 * cost 9-12 % (measured) */
			const uint32_t dense_sync = (dense_k && !sub && ix->s16c_density >= 400.0f) ? (uint32_t) g_s16c_dense_sync : 0u;
			/* ... and where the buckets are probed by a hundred queries or two, most tiles have 128 members at most: the kernel
			 * with the one-pair-block wave map for those (sigma 0.5: sweep 1.48 -> 1.40 ms, sigma 1.0: 1.98 -> 1.95) */
			const bool	dense_small = dense_k && g_s16c_dense_small && ix->s16c_density < 400.0f;

			if (dense_k)
			{
				NDB_STAT_ADD(dense_sweeps, 1);
				if (dense_sync)
					HIP_TRY(hipMemsetAsync(next_item, 0, (size_t) 8 * NDB_QHEAD_STRIDE * sizeof(unsigned int), g.stream));
			}
			/* (two chunks in flight: the 168-register form whatever the blocks — at two blocks a compute unit it leaves a
			 * third of the register file to the other steps' kernels, see --inflight) */
			if (wd)
				NDB_STAT_ADD(wave_sweeps, 1);
			if (wd == 2)
				S16C_WSWEEP_D(2, 3);
			else if (wd == 3)
				S16C_WSWEEP_D(3, 2);
			else if (wd == 4)
				S16C_WSWEEP_D(4, 2);
			else if (dense_k && g_s16_debug == 6)
				S16C_DENSE_L(6);
			else if (dense_k && g_s16_debug == 7)
				S16C_DENSE_L(7);
			else if (dense_k && (g_s16_debug < 1 || g_s16_debug > 4) && g_s16c_dense_split == 3 && dense_small)
				S16C_DENSE_LS(0, 3, true);
			else if (dense_k && (g_s16_debug < 1 || g_s16_debug > 4) && g_s16c_dense_split == 3)
				S16C_DENSE_LS(0, 3, false);
			else if (dense_k && (g_s16_debug < 1 || g_s16_debug > 4))
				S16C_DENSE_L(0);
			else if (dense_k && g_s16_debug == 1)
				S16C_DENSE_L(1);
			else if (dense_k && g_s16_debug == 2)
				S16C_DENSE_L(2);
			else if (dense_k && g_s16_debug == 3)
				S16C_DENSE_L(3);
			else if (dense_k && g_s16_debug == 4)
				S16C_DENSE_L(4);
			else if (g_s16_debug == 1 && c_qb == 8)
				S16C_SWEEP_L(8, 2, 1);
			else if (g_s16_debug == 2 && c_qb == 8)
				S16C_SWEEP_L(8, 2, 2);
			else if (g_s16_debug == 3 && c_qb == 8)
				S16C_SWEEP_L(8, 2, 3);
			else if (g_s16_debug == 4 && c_qb == 8)
				S16C_SWEEP_L(8, 2, 4);
			else if (g_s16_debug == 1)
				S16C_SWEEP_L(4, 2, 1);
			else if (g_s16_debug == 2)
				S16C_SWEEP_L(4, 2, 2);
			else if (c_qb == 8 && !g_s16c_epi)
				S16C_SWEEP_LE(8, 2, 0, 0);
			else if (c_qb == 8 && g_s16c_pf == 3)
				S16C_SWEEP_LP(8, 2, 0, 1, 3);
			else if (c_qb == 8 && g_s16c_pf == 16)
				S16C_SWEEP_LP(8, 2, 0, 1, 16);
			else if (c_qb == 8 && g_s16c_pf == 32)
				S16C_SWEEP_LP(8, 2, 0, 1, 32);
			else if (c_qb == 8 && g_s16c_pf == 48)
				S16C_SWEEP_LP(8, 2, 0, 1, 48);
			else if (c_qb == 8)
				S16C_SWEEP_L(8, 2, 0);
			else if (c_qb == 1 && nbuf == 3 && !g_s16c_epi)
				S16C_SWEEP_LE(1, 3, 0, 0);
			else if (c_qb == 4 && nbuf == 2 && !g_s16c_epi)
				S16C_SWEEP_LE(4, 2, 0, 0);
			else if (c_qb == 1 && nbuf == 2)
				S16C_SWEEP_L(1, 2, 0);
			else if (c_qb == 1)
				S16C_SWEEP_L(1, 3, 0);
			else if (nbuf == 3)
				S16C_SWEEP_L(4, 3, 0);
			else
				S16C_SWEEP_L(4, 2, 0);
		}
		else
		S16_BY_RH(S16_SWEEP_L, ds, (const unsigned char *) ix->d_planes,
				  sub ? (const uint32_t *) ix->d_sub_blk : (const uint32_t *) ix->d_blkoff,
				  (const float *) ix->d_rn2, (const int16_t *) ix->d_rexp, (const unsigned char *) ix->w_qplanes, qrowbytes,
				  (const float *) ix->w_qn2, (const int *) ix->w_qexp, (float2 *) ix->w_qthr, lco, npr,
				  (const uint32_t *) cnt, (const uint32_t *) pair_off, (const S16Desc *) ix->w_s16desc,
				  (const PairRec *) ix->w_pairs, next_item, (const uint32_t *) runs, ecount, ix->w_erec, ecap,
				  ix->w_bmin, nq < 1024 ? 1 : 0, dimp / S16_CH, desc_cap, g_s16_tighten ? (uint32_t) k : 0u,
				  sub ? (const uint32_t *) ix->d_posof : (const uint32_t *) nullptr);
		if (round == 0 && t.stop()) return NDBHIP_ERR_HIP;
		if (turn.end()) return NDBHIP_ERR_HIP;
		if (wd)
			hipLaunchKernelGGL(k_s16w_collect, dim3(nq), dim3(64), 0, g.stream, (uint32_t) nq, (const uint32_t *) ix->w_qslot,
							   (const uint32_t *) (ix->w_qslot + (size_t) nq * S16_QP_CAP), (uint32_t) S16_QP_CAP,
							   (const uint32_t *) (ix->w_pslot + 4 * (size_t) qc_cap), (const uint32_t *) (ix->w_pslot + 3 * (size_t) qc_cap),
							   (const uint32_t *) (ix->w_pslot + qc_cap), (const int64_t *) ix->d_prow_off, (const uint32_t *) ds.own_len,
							   (const uint32_t *) ix->d_pposof, (const uint32_t *) ix->w_wmask, (const float2 *) ix->w_wrec, ecount,
							   ix->w_erec, ix->w_eub, ecap, act, (const uint32_t *) pair_off, ncs, qc_cap,
							   (const uint32_t *) (grp_off + ncs), wcap);

#define S16_FIN_L(RR, HH, ...)                                                                                                      \
	do {                                                                                                                            \
		if (fin_nbuf || fin_nbuf4)                                                                                                  \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_finalize<RR, HH, true>), dim3(nq), dim3(g_s16_fin_threads), fsmem, g.stream, __VA_ARGS__); \
		else                                                                                                                        \
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s16_finalize<RR, HH, false>), dim3(nq), dim3(g_s16_fin_threads), fsmem, g.stream, __VA_ARGS__); \
	} while (0)
	S16_BY_RH(S16_FIN_L, d, d_q, w_probes, (const uint32_t *) ix->w_candoff, lco, npr, (uint32_t) k,
			  (const float2 *) ix->w_qthr, (const unsigned int *) ecount, (const uint2 *) ix->w_erec, ecap, partial,
			  d_cand, d_ncand, d_total, d_otid, d_odist, d_ocnt, surv, flags, cen ? (const float *) ix->w_eub : (const float *) nullptr,
			  surv_cap, ix->w_overq, ipc ? (const float *) ix->w_qev : (const float *) nullptr, fin_nbuf, fin_nbuf4);
		HIP_TRY(hipGetLastError());
		{
			unsigned int f[8];

			HIP_TRY(hipMemcpyAsync(f, flags, sizeof(f), hipMemcpyDeviceToHost, g.stream));
			HIP_TRY(hipStreamSynchronize(g.stream));
			if (wd && f[7] > ix->wc_words)
				ix->wc_words = f[7];
			if (f[6])
			{
				/* a query with more pairs than its list holds (ndbhip_screen16w.h): this mirror's batches take the LDS ring
				 * from now on; this one goes to the older path */
				ix->s16w_off = true;
				f[2] = 1;
			}
			over = f[0] | f[2];
			over_n = f[0];
			over_pairs = f[2] != 0;
			if (g_debug_s16)
				fprintf(stderr, "s16 debug: round %d flags %u %u %u\n", round, f[0], f[1], f[2]);
			if (round == 0)
				memcpy(fl8, f, sizeof(fl8));
			if (f[2])
			{
				/* more pairs than the pair planes hold: nothing was swept, the older path serves the batch — and
				 * the next batch gets planes twice as large */
				if (ix->qc_mult < 16)
					ix->qc_mult *= 2;
				/* (the word arrays only when THEY were what did not fit: f[7] = the words the batch needed — already remembered in
				 * wc_words above, which sizes the next batch; a pair-plane overflow alone must not quadruple 256 bytes a word) */
				if (wd && f[7] > wcap && ix->wc_mult < 4096)
					ix->wc_mult *= 4;
				break;
			}
		}
		if (!over)
			break;
	}
	/* statistics: survivors given the reference's arithmetic, records emitted (as the last round left them) */
	hipLaunchKernelGGL(k_sum_u32, dim3(std::min(64, (nq + 255) / 256)), dim3(256), 0, g.stream, (const unsigned int *) surv, (uint32_t) nq,
					   g.d_counters + 3);
	hipLaunchKernelGGL(k_sum_u32, dim3(std::min(64, (nq + 255) / 256)), dim3(256), 0, g.stream, (const unsigned int *) ecount, (uint32_t) nq,
					   g.d_counters + 4);
	if (cen && fl8[5] > 0)
		ix->s16c_density = (float) fl8[4] / (float) fl8[5];		/* pairs per bucket with pairs: the next batch's tile size */
		if (getenv("NDB_DENSITY"))
			fprintf(stderr, "[ndbhip] pairs %llu buckets with pairs %llu density %.1f sub %d\n", (unsigned long long) fl8[4], (unsigned long long) fl8[5], ix->s16c_density, (int) ix->s16_sub);
	if (g_debug_s16)
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
		fprintf(stderr, "s16 debug: cen %d qb %d flags %u %u %u %u pairs %u buckets-with-pairs %u qc_cap %u\n", (int) cen, c_qb, fl8[0], fl8[1],
				fl8[2], fl8[3], fl8[4], fl8[5], qc_cap);
		{
			float		tmin = 3e38f, tmax = 0.0f;
			int			ninf = 0;

			for (int q = 0; q < nq; q++)
			{
				if (!(th[q].x < 3e38f))
					ninf++;
				else
				{
					tmin = std::min(tmin, th[q].x);
					tmax = std::max(tmax, th[q].x);
				}
			}
			fprintf(stderr, "s16 debug: thresholds after the batch: min %g max %g, %d infinite; hook %s, seeds by sublist %d\n", tmin, tmax, ninf,
					g_thr_hook ? "set" : "none", (int) seed_by_sublist);
		}
	}
	if (over)
	{
		/* a handful of queries with more records or survivors than their buffers hold (a dense neighbourhood in a
		 * bucket the centring serves badly): they alone go to the exact path, the others' results stand */
		if (g_s16_redo && over_n > 0 && over_n <= (unsigned int) S16_OVER_CAP && over_n <= (unsigned int) std::max(1, nq / 8) && !over_pairs)
		{
			ix->redo.resize(over_n);
			HIP_TRY(hipMemcpyAsync(ix->redo.data(), ix->w_overq, (size_t) over_n * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
			HIP_TRY(hipStreamSynchronize(g.stream));
			NDB_STAT_ADD(screen16_batches, 1);
			return 2;
		}
		NDB_STAT_ADD(screen16_fallbacks, 1);
		return 1;
	}
	NDB_STAT_ADD(screen16_batches, 1);
	return 0;
}

/* ambuild's last step (optional): everything the first batched scan would otherwise prepare lazily — sublists of
 * the long lists, the rows' fp16 planes, norms and radii — for the operator class `strategy` */
extern "C" int
ndbhip_ivf_prepare(ndbhip_ivf *ix, int strategy)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (!ix)
		return fail(NDBHIP_ERR_INVALID, "index is NULL");
	if (!ix->loaded || ix->ncent < 1)
		return fail(NDBHIP_ERR_STATE, "index has no centroids/lists loaded");
	int			rc = ivf_flush(ix);

	if (rc)
		return rc;
	const int	R = ivf_recipe(strategy);

	if (ix->nrows < 1 || !ivf_s16_eligible(ix, NDB_SCREEN_MIN_NQ, R, 10))
		return NDBHIP_OK;		/* nothing is prepared ahead for the other scan paths */
	rc = ivf_s16_prepare(ix, R);
	if (rc)
		return rc;
	HIP_TRY(hipStreamSynchronize(g.stream));
	return NDBHIP_OK;
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
ndbhip_mfma_probe_f32(const float *d_a, const float *d_b, const float *d_c, float *d_d, int ntiles)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (ntiles < 0 || (ntiles > 0 && (!d_a || !d_b || !d_c || !d_d)))
		return fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (ntiles == 0)
		return NDBHIP_OK;
	hipLaunchKernelGGL(k_s16_mfma_probe_f32, dim3(ntiles), dim3(64), 0, g.stream, d_a, d_b, d_c, d_d);
	HIP_TRY(hipGetLastError());
	return NDBHIP_OK;
}

/* profiling builds (-DNDB_PHASES): the 64 clock stamps of block 0 (100 MHz); otherwise zeros */
extern "C" int
ndbhip_debug_phases(unsigned long long *out)
{
	if (!out)
		return fail(NDBHIP_ERR_INVALID, "ndbhip_debug_phases: out is NULL");
#ifdef NDB_PHASES
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phases), 64 * sizeof(unsigned long long)));
#else
	memset(out, 0, 64 * sizeof(unsigned long long));
#endif
	return NDBHIP_OK;
}


/* profiling builds (-DNDB_PHASES): the register-streaming sweep's per-wave trace of its last launch — n words of
 * {first request, end (100 MHz clock), items, ticks inside the stream's waits} per wave; otherwise zeros */
extern "C" int
ndbhip_debug_trace(unsigned long long *out, int n)
{
	if (!out || n < 0)
		return fail(NDBHIP_ERR_INVALID, "ndbhip_debug_trace: bad arguments");
	memset(out, 0, (size_t) n * sizeof(unsigned long long));
#ifdef NDB_PHASES
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wtrace), (size_t) std::min(n, S16W_TRACE_N) * sizeof(unsigned long long)));
#endif
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
	else if (!strcmp(name, "block_cache"))
	{
		g.big_cache_on = value != 0;
		if (!g.big_cache_on)
			big_cache_flush();
	}
	else if (!strcmp(name, "screen16_sublists"))
		g_s16_sublists = value != 0;
	else if (!strcmp(name, "screen16_sub_min"))
	{
		if (value < 256)
			return fail(NDBHIP_ERR_INVALID, "screen16_sub_min must be >= 256");
		g_s16_sub_min = value;
	}
	else if (!strcmp(name, "screen16_sub_rows"))
	{
		if (value < 32 || value > 65536)
			return fail(NDBHIP_ERR_INVALID, "screen16_sub_rows must be 32..65536");
		g_s16_sub_rows = value;
	}
	else if (!strcmp(name, "probe_select_radix"))
		g_probe_sel_radix = value != 0;
	else if (!strcmp(name, "probe_select_threads"))
	{
		if (value != 64 && value != 128 && value != 256)
			return fail(NDBHIP_ERR_INVALID, "probe_select_threads must be 64, 128 or 256");
		g_probe_sel_threads = value;
	}
	else if (!strcmp(name, "screen16_fin_threads"))
	{
		if (value != 64 && value != 128 && value != 256)
			return fail(NDBHIP_ERR_INVALID, "screen16_fin_threads must be 64, 128 or 256");
		g_s16_fin_threads = value;
	}
	else if (!strcmp(name, "screen16_stage"))
	{
		if (value < 0 || value > 13)
			return fail(NDBHIP_ERR_INVALID, "screen16_stage must be 0 .. 13");
		g_s16_stage = value;
	}
	else if (!strcmp(name, "screen16_prune"))
		g_s16_prune = value != 0;
	else if (!strcmp(name, "screen16_tighten"))
		g_s16_tighten = value != 0;
	else if (!strcmp(name, "screen16_waves"))
	{
		if (value != 4 && value != 8)
			return fail(NDBHIP_ERR_INVALID, "screen16_waves must be 4 or 8");
		g_s16_waves = value;
	}
	else if (!strcmp(name, "screen16_debug"))
		g_s16_debug = value;
	else if (!strcmp(name, "screen16_centered"))
		g_s16_cen = value != 0;
	else if (!strcmp(name, "cent_screen16"))
		g_cent_s16 = value != 0;
	else if (!strcmp(name, "screen16c_qb"))
	{
		if (value != 0 && value != 1 && value != 4 && value != 8)
			return fail(NDBHIP_ERR_INVALID, "screen16c_qb must be 0 (auto), 1, 4 or 8");
		g_s16c_qb = value;
	}
	else if (!strcmp(name, "screen16_slack"))
		g_s16_slack = value != 0;
	else if (!strcmp(name, "screen16_cosine_centered"))
		g_s16_cos_cen = value != 0;
	else if (!strcmp(name, "screen16_redo"))
		g_s16_redo = value != 0;
	else if (!strcmp(name, "screen16_cosine"))
		g_s16_cos = value != 0;
	else if (!strcmp(name, "build_prepare"))
	{
		if (value < 0 || value > 3)
			return fail(NDBHIP_ERR_INVALID, "build_prepare must be 0 (off) or a strategy 1 .. 3");
		g_build_prepare = value;
	}
	else if (!strcmp(name, "screen_min_nq"))
	{
		if (value < 1 || value > 65536)
			return fail(NDBHIP_ERR_INVALID, "screen_min_nq must be 1 .. 65536");
		g_screen_min_nq = value;
	}
	else if (!strcmp(name, "screen16c_seeds"))
	{
		if (value != 0 && value != 16 && value != 24 && value != 32 && value != 48 && value != 64)
			return fail(NDBHIP_ERR_INVALID, "screen16c_seeds must be 0 (default), 16, 24, 32, 48 or 64");
		g_s16c_seeds = value;
	}
	else if (!strcmp(name, "screen16_ip_centered"))
		g_s16_ip_cen = value != 0;
	else if (!strcmp(name, "screen16c_sample"))
	{
		if (value != 0 && (value < 256 || value > 2048))
			return fail(NDBHIP_ERR_INVALID, "screen16c_sample must be 0 or 256..2048");
		g_s16c_sample = value;
	}
	else if (!strcmp(name, "screen16c_dense_min_sub"))
	{
		if (value != 0 && (value < 24 || value > 100000))
			return fail(NDBHIP_ERR_INVALID, "screen16c_dense_min_sub must be 0 (never) or 24 .. 100000 pairs per bucket");
		g_s16c_dense_min_sub = value;
	}
	else if (!strcmp(name, "screen16c_dense_min"))
	{
		if (value < 24 || value > 100000)
			return fail(NDBHIP_ERR_INVALID, "screen16c_dense_min must be 24 .. 100000 pairs per bucket");
		g_s16c_dense_min = value;
	}
	else if (!strcmp(name, "screen16_sub_restrict"))
	{
		if (value < 0)
			return fail(NDBHIP_ERR_INVALID, "screen16_sub_restrict must be 0 (never) or a number of centres");
		g_sub_restrict = value;
	}
	else if (!strcmp(name, "screen16c_tight"))
	{
		if (value < 8 || value > 1024 || (value & (value - 1)))
			return fail(NDBHIP_ERR_INVALID, "screen16c_tight must be a power of two, 8..1024");
		g_s16c_tight = value;
	}
	else if (!strcmp(name, "screen16c_rot"))
	{
		if (value < 0 || value > 2)
			return fail(NDBHIP_ERR_INVALID, "screen16c_rot must be 0, 1 or 2");
		g_s16c_rot = value;
	}
	else if (!strcmp(name, "screen16c_dense"))
		g_s16c_dense = value != 0;
	else if (!strcmp(name, "screen16c_dense_small"))
		g_s16c_dense_small = value != 0;
	else if (!strcmp(name, "screen16c_dense_sync"))
	{
		if (value < 0 || value > 1024)
			return fail(NDBHIP_ERR_INVALID, "screen16c_dense_sync must be 0..1024");
		g_s16c_dense_sync = value;
	}
	else if (!strcmp(name, "screen16c_dense_split"))
	{
		if (value != 3 && value != 4)
			return fail(NDBHIP_ERR_INVALID, "screen16c_dense_split must be 3 or 4");
		g_s16c_dense_split = value;
	}
	else if (!strcmp(name, "screen16c_dense_spare"))
	{
		if (value < 0 || value > 128)
			return fail(NDBHIP_ERR_INVALID, "screen16c_dense_spare must be 0..128");
		g_s16c_dense_spare = value;
	}
	else if (!strcmp(name, "screen16c_pfd"))
	{
		if (value < 0 || value > 10)
			return fail(NDBHIP_ERR_INVALID, "screen16c_pfd must be 0..10");
		g_s16c_pfd = value;
	}
	else if (!strcmp(name, "screen16c_pf"))
	{
		if (value != 0 && value != 3 && value != 16 && value != 32 && value != 48)
			return fail(NDBHIP_ERR_INVALID, "screen16c_pf must be 0, 3, 16, 32 or 48");
		g_s16c_pf = value;
	}
	else if (!strcmp(name, "screen16c_epi"))
		g_s16c_epi = value != 0;
	else if (!strcmp(name, "screen16c_wave"))
	{
		if (value != 0 && (value < 2 || value > 4))
			return fail(NDBHIP_ERR_INVALID, "screen16c_wave must be 0 (the LDS ring) or 2 .. 4 chunks in flight");
		g_s16c_wave = value;
	}
	else if (!strcmp(name, "screen16c_plane_seeds"))
		g_s16c_plseed = value != 0;
	else if (!strcmp(name, "screen16c_bigk"))
		g_s16c_bigk = value != 0;
	else if (!strcmp(name, "slow_call_log"))
		g_slow_call_us = value;
	else if (!strcmp(name, "screen16c_wave_min_nq"))
	{
		if (value < 1)
			return fail(NDBHIP_ERR_INVALID, "screen16c_wave_min_nq must be >= 1");
		g_s16c_wave_min_nq = value;
	}
	else if (!strcmp(name, "screen16_sweep_queue"))
		g_s16_sweepq = value != 0;
	else if (!strcmp(name, "screen16c_wave_blocks"))
	{
		if (value < 1 || value > 3)
			return fail(NDBHIP_ERR_INVALID, "screen16c_wave_blocks must be 1, 2 or 3 (3: with screen16c_wave = 2)");
		g_s16c_wblk = value;
	}
	else if (!strcmp(name, "screen16c_nbuf"))
	{
		if (value != 0 && value != 2 && value != 3)
			return fail(NDBHIP_ERR_INVALID, "screen16c_nbuf must be 0 (default), 2 or 3");
		g_s16c_nbuf = value;
	}
	else if (!strcmp(name, "kmeans_screen16"))
		g_kmeans_s16 = value != 0;
	else if (!strcmp(name, "build_single_sweep"))
		g_build_single_sweep = value != 0;
	else if (!strcmp(name, "build_screen16"))
		g_build_s16 = value != 0;
	else if (!strcmp(name, "gchunk"))
	{
		if (value != 32 && value != 64)
			return fail(NDBHIP_ERR_INVALID, "gchunk must be 32 or 64");
		g_gchunk = value;
	}
	else if (!strcmp(name, "scr_coop"))
		g_scr_coop = value;
	else if (!strcmp(name, "scr_ch"))
		g_scr_ch = value;
	else if (!strcmp(name, "scr_mfma"))
		g_scr_mfma = value != 0;
	else if (!strcmp(name, "debug_s16"))
		g_debug_s16 = value;
	else if (!strcmp(name, "debug_build"))
		g_debug_build = value;
	else if (!strcmp(name, "hnsw_intended_waves"))
	{
		if (value < 1 || value > 32)
			return fail(NDBHIP_ERR_INVALID, "hnsw_intended_waves must be 1 .. 32 (waves per CU)");
		g_h2_waves = value;
	}
	else if (!strcmp(name, "hnsw_intended_host_groups"))
		g_h2_host_groups = value != 0;
	else if (!strcmp(name, "hnsw_intended_occ4"))
	{
		if (value < 0 || value > 2)
			return fail(NDBHIP_ERR_INVALID, "hnsw_intended_occ4 must be 0 (three walkers a SIMD), 1 (four where that costs no scratch) or 2 (four everywhere)");
		g_h2_occ4 = value;
	}
	else if (!strcmp(name, "hnsw_trace"))
		g_hnsw_trace = value;
	else if (!strcmp(name, "hnsw_nofast"))
		g_hnsw_nofast = value;
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

/* per-sub-batch budget for the candidate-distance buffer of the paths that keep one (the fp16 matrix-core screen
 * does not) */
static size_t g_dist_budget_bytes = (size_t) 8 << 30;	/* of 288 GB: a 4096-query step of the 1M x 768 workload needs 2.1 GB */

static int
ivf_dist_budget_queries(uint32_t stride, int nq)
{
	const int	qb = (int) std::min<size_t>(65535, std::max<size_t>(1, g_dist_budget_bytes / ((size_t) stride * 4)));

	return std::min(qb, std::max(nq, 1));
}

/* buffers only the exact / grouped / fp32-screened scans use */
static int
ivf_grow_scan_buffers(ndbhip_ivf *ix, int qb, uint32_t stride, int nprobe)
{
	if (grow(ix->w_dist, ix->w_dist_n, (size_t) qb * stride)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_tmin, ix->w_tmin_n, (size_t) qb * ((((stride >> 6) + (size_t) nprobe + 2) + 63) & ~(size_t) 63)))
		return NDBHIP_ERR_HIP;
	/* (the same predicate as `grouped` in ivf_search_chunk: modes 2 .. 5 run the grouped scan for any batch size) */
	if ((ix->dim % NDB_CHUNK) == 0 && g_scan_mode != 1 && (qb >= NDB_GROUPED_MIN_NQ || g_scan_mode >= 2))
		if (grow(ix->w_qblock, ix->w_qblock_n,
				 ((size_t) qb * nprobe / NDB_QG + (size_t) ix->ncent) * (size_t) ix->dim * NDB_QG))
			return NDBHIP_ERR_HIP;
	return 0;
}

/* does a sub-batch of nq queries go to the fp16 matrix-core screen (ndbhip_screen16.h)?  mode 5 forces it */
static bool
ivf_s16_wanted(const ndbhip_ivf *ix, int nq, int R, int k)
{
	return (g_scan_mode == 5 || (g_scan_mode == 0 && g_screen_auto && g_s16_auto && nq >= g_screen_min_nq)) &&
		ivf_s16_eligible(ix, nq, R, k);
}

/* row which[i] of src (words dwords each) -> row i of dst */
__global__ void
k_rows_pick(const uint32_t *__restrict__ src, uint32_t words, const int64_t *__restrict__ which, uint32_t *__restrict__ dst)
{
	const uint32_t *s = src + (size_t) which[blockIdx.x] * words;
	uint32_t   *d = dst + (size_t) blockIdx.x * words;

	for (uint32_t j = threadIdx.x; j < words; j += blockDim.x)
		d[j] = s[j];
}

/* row i of src (words dwords each) -> row which[i] of dst */
__global__ void
k_rows_scatter(const uint32_t *__restrict__ src, uint32_t words, const int64_t *__restrict__ which, uint32_t *__restrict__ dst)
{
	const uint32_t *s = src + (size_t) blockIdx.x * words;
	uint32_t   *d = dst + (size_t) which[blockIdx.x] * words;

	for (uint32_t j = threadIdx.x; j < words; j += blockDim.x)
		d[j] = s[j];
}

/* queries already on the device; runs select (+ scan + topk when `full`) for one sub-batch */
static int
ivf_search_chunk(ndbhip_ivf *ix, const float *d_q, int nq, int strategy, int npr, int k,
				 int64_t max_candidates, uint32_t stride, bool full, int partial,
				 ndbhip_cand *d_cand, int *d_ncand, int64_t *d_total,
				 uint64_t *d_otid, float *d_odist, int *d_ocnt, const int *d_probes_in = nullptr,
				 int *d_probes_out = nullptr, bool allow_s16 = true)
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
	bool		summed = false;	/* k_probe_select added the candidate counts itself */

	if (d_probes_in)
	{
		/* probes chosen elsewhere (each rank of a sharded search selects for its slice of the queries) */
		w_probes = const_cast<int *>(d_probes_in);
		hipLaunchKernelGGL(k_probe_offsets, dim3((nq + 255) / 256), dim3(256), 0, g.stream, d_probes_in,
						   (uint32_t) nq, npr, ix->ncent, (const uint32_t *) d.glob_len, d.own_lo, d.own_len,
						   (uint64_t) (max_candidates > 0 ? max_candidates : 0), ix->dim * (ix->f16 ? 2 : 4),
						   ix->w_candoff, lco_w, full ? g.d_counters : (unsigned long long *) nullptr);
	}
	/* the matrix-core screen serves this sub-batch: the queries' fp16 planes first (the centroid scan below and the
	 * sweep both multiply them) */
	const bool	s16_here = full && allow_s16 && ivf_s16_wanted(ix, nq, ivf_recipe(strategy), k);
	bool		cent_done = false;
	/* the query planes the centroid scan multiplies (cosine: not the sweep's) */
	const unsigned char *cq_planes = nullptr;
	const float *cq_n2 = nullptr;
	const int  *cq_exp = nullptr;

	ix->bat_subdist = nullptr;
	ix->bat_restrict = false;

	if (s16_here)
	{
		const int	dimp = (ix->dim + 63) & ~63;

		if (grow(ix->w_qplanes, ix->w_qplanes_n, (size_t) nq * dimp * 4)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_qn2, ix->w_qn2_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_qexp, ix->w_qexp_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_qthr, ix->w_qthr_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		const float *qsrc = d_q;

		cq_planes = ix->w_qplanes;
		cq_n2 = ix->w_qn2;
		cq_exp = ix->w_qexp;
		if (ivf_recipe(strategy) == R_IVF_COS)
		{
			/* cosine: the planes are those of q / |q| (the exact arithmetic keeps reading d_q itself) */
			if (grow(ix->w_qhat, ix->w_qhat_n, (size_t) nq * ix->dim)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_rows_normalise<0>, dim3((nq + 3) / 4), dim3(256), 0, g.stream, (const void *) d_q, (int64_t) nq, ix->dim, ix->w_qhat,
							   ivf_s16_centered(ix, R_IVF_COS) ? 1 : 0);
			qsrc = ix->w_qhat;
			/* ... and the centroid scan, which is L2 in the rows' own space, multiplies planes of the queries as they are */
			if (grow(ix->w_qplanes_o, ix->w_qplanes_o_n, (size_t) nq * dimp * 4)) return NDBHIP_ERR_HIP;
			if (grow(ix->w_qn2_o, ix->w_qn2_o_n, (size_t) nq)) return NDBHIP_ERR_HIP;
			if (grow(ix->w_qexp_o, ix->w_qexp_o_n, (size_t) nq)) return NDBHIP_ERR_HIP;
			hipLaunchKernelGGL(k_s16_qprep, dim3((nq + 3) / 4), dim3(256), 0, g.stream, d_q, (uint32_t) nq, ix->dim, dimp,
							   (ndb_h2 *) ix->w_qplanes_o, ix->w_qn2_o, ix->w_qexp_o);
			cq_planes = ix->w_qplanes_o;
			cq_n2 = ix->w_qn2_o;
			cq_exp = ix->w_qexp_o;
		}
		hipLaunchKernelGGL(k_s16_qprep, dim3((nq + 3) / 4), dim3(256), 0, g.stream, qsrc, (uint32_t) nq, ix->dim, dimp,
						   (ndb_h2 *) ix->w_qplanes, ix->w_qn2, ix->w_qexp);
	}
	if (g_thr_hook && !s16_here && full && allow_s16)
	{
		/* a sharded search whose other ranks exchange thresholds while this one serves the sub-batch another way
		 * (no rows here, a recipe the screen does not take): it joins the exchange with +inf */
		if (grow(ix->w_qthr, ix->w_qthr_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) ix->w_qthr, 0x7F800000, (size_t) 2 * nq, g.stream));
		const int	rc = g_thr_hook((float *) ix->w_qthr, (size_t) 2 * nq);

		if (rc)
			return rc;
	}
	if (!d_probes_in && s16_here && g_cent_s16 && ncmp >= 256 && ncmp <= 4096 && npr <= NDBHIP_MAX_NPROBE)
	{
		/* HOT LOOP 1 for a screened batch: |q - centroid|^2 of every pair from the two-plane sweep (MODE 3), the
		 * reference's arithmetic for the centroids near the nprobe-th only (k_cent_select) */
		/* the planes laid out and their centres known: the centroids and the regrouped lists' centres are one matrix
		 * (dm_all), the sweep's sublist code reads its columns past the centroids' */
		const bool	both0 = ix->s16_valid && ix->s16_sub && ix->dm_all_valid && ix->dm_all_ncmp == ncmp && ix->nsub_g > 0 &&
			ix->dm_all.n == ncmp + ix->nsub_g && g_s16_prune && ivf_recipe(strategy) != R_IVF_COS;
		/* many centres: the matrix is the centroids' alone and the centres of the probed lists follow once the probes are
		 * known (float4 queries; the norms and the error bound stay dm_all's) */
		const bool	restr = both0 && g_sub_restrict > 0 && ix->nsub_g >= g_sub_restrict && !ix->f16;
		const bool	both = both0 && !restr;
		S16Mat	   &cm = both ? ix->dm_all : ix->dm_cent;
		const uint32_t astride = (uint32_t) (((both ? ncmp + ix->nsub_g : ncmp) + 63) & ~63);

		if (!both && (!ix->dm_cent_valid || ix->dm_cent.n != ncmp || ix->dm_cent.src != ix->d_centroids) && ivf_frozen(ix))
			return fail(NDBHIP_ERR_STATE, "the mirror is shared (ndbhip_ivf_share): run a batch on the source before sharing (the centroid matrix is not there yet)");
		if (!both && (!ix->dm_cent_valid || ix->dm_cent.n != ncmp || ix->dm_cent.src != ix->d_centroids))
		{
			const int	rc = s16mat_prepare(ix->dm_cent, ix->d_centroids, ncmp, ix->dim);

			if (rc)
				return rc;
			ix->dm_cent_valid = true;
		}
		if (grow(ix->w_amat, ix->w_amat_n, (size_t) nq * astride)) return NDBHIP_ERR_HIP;
		if (grow(ix->w_cfull, ix->w_cfull_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		{
			const int	rc = s16mat_run(cm, ix->dim, cq_planes, cq_n2, cq_exp, ix->w_qthr, nq, ix->w_amat, astride);

			if (rc)
				return rc;
		}
		if (both)
		{
			ix->bat_subdist = ix->w_amat + ncmp;
			ix->bat_sstride = astride;
		}
		ix->bat_restrict = restr;
#define CENT_SELECT_L(PER, ST)                                                                                      \
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cent_select<PER, ST>), dim3(nq), dim3(64), (size_t) cs_nbuf * S16X_SLOT(4), g.stream, (const float *) ix->w_amat, astride, \
						   cq_n2, (const uint32_t *) cm.xmax, d_q, (const float *) d.centroids, ix->dim, \
						   ncmp, ix->ncent, npr, (const uint32_t *) d.glob_len, d.own_lo, d.own_len,                          \
						   (uint64_t) (max_candidates > 0 ? max_candidates : 0), ix->w_cdist, cstride, w_probes, ix->w_candoff, \
						   lco_w, ix->w_cfull, cs_nbuf)
		/* the candidates' centroids through LDS while a CU has few queries to work on (4 chunks deep: 68 KB a block); a large
		 * batch keeps its 16 blocks per CU (measured: 3 staged blocks a CU took 238 us where 16 lane-per-row ones take 84) */
		const int	cs_nbuf = (g_s16_stage && s16_staged_ok(ix->dim, 0) && (nq <= 512 || g_s16_stage >= 2))
			? s16_staged_nbuf(4, g_s16_stage >= 2 ? g_s16_stage : 4) : 0;

		{
			static bool cs_attr = false;	/* static + dynamic LDS beyond 64 KB has to be asked for */

			if (!cs_attr)
			{
				HIP_TRY(hipFuncSetAttribute((const void *) k_cent_select<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * S16X_SLOT(4)));
				HIP_TRY(hipFuncSetAttribute((const void *) k_cent_select<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * S16X_SLOT(4)));
				HIP_TRY(hipFuncSetAttribute((const void *) k_cent_select<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * S16X_SLOT(4)));
				cs_attr = true;
			}
		}
		if (cs_nbuf)
		{
			if (ncmp <= 1024)
				CENT_SELECT_L(16, true);
			else if (ncmp <= 2048)
				CENT_SELECT_L(32, true);
			else
				CENT_SELECT_L(64, true);
		}
		else if (ncmp <= 1024)
			CENT_SELECT_L(16, false);
		else if (ncmp <= 2048)
			CENT_SELECT_L(32, false);
		else
			CENT_SELECT_L(64, false);
		/* (a query with too many near-ties, or fewer than nprobe finite bounds, had all its distances computed
		 * exactly and is selected the old way) */
		hipLaunchKernelGGL(k_probe_select, dim3(nq), dim3(256), 0, g.stream, (const float *) ix->w_cdist, cstride,
						   ncmp, ix->ncent, npr, (const uint32_t *) d.glob_len, d.own_lo, d.own_len,
						   (uint64_t) (max_candidates > 0 ? max_candidates : 0), ix->dim * (ix->f16 ? 2 : 4),
						   w_probes, ix->w_candoff, lco_w, g.d_counters, 1, (const uint8_t *) ix->w_cfull);
		NDB_STAT_ADD(cent_screen_batches, 1);
		cent_done = true;
	}
	if (!d_probes_in && !cent_done)
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
		/* (measured for batches of 4096: 256 threads 1.85 ms per step, 128: 1.88, 64: 1.97 — the sort wants the threads;
		 * "probe_select_threads" for A/B) */
		hipLaunchKernelGGL(k_probe_select, dim3(nq), dim3(nq >= 512 ? g_probe_sel_threads : 256), 0, g.stream, (const float *) ix->w_cdist, cstride,
						   ncmp, ix->ncent, npr, (const uint32_t *) d.glob_len, d.own_lo, d.own_len,
						   (uint64_t) (max_candidates > 0 ? max_candidates : 0), ix->dim * (ix->f16 ? 2 : 4),
						   w_probes, ix->w_candoff, lco_w, full ? g.d_counters : (unsigned long long *) nullptr,
						   (nq >= 512 && g_probe_sel_radix) ? 0 : 1, (const uint8_t *) nullptr, (full && nq <= 16) ? 1 : 0);
		summed = full && nq <= 16;
	}
	HIP_TRY(hipGetLastError());
	if (!full)
		return 0;
	if (!summed)
	hipLaunchKernelGGL(k_sum_candidates, dim3(std::min(64, (nq + 255) / 256)), dim3(256), 0, g.stream, (const uint32_t *) ix->w_candoff,
					   (const uint32_t *) lco_w, (uint32_t) nq, npr, ix->dim * (ix->f16 ? 2 : 4), g.d_counters);

	/* HOT LOOP 2 */
	const int	R = ivf_recipe(strategy);

	/* batches of >= 128 queries: the bound pass on fp16 matrix cores (ndbhip_screen16.h); mode 5 forces it */
	if (allow_s16 && ivf_s16_wanted(ix, nq, R, k))
	{
		const int	rc = ivf_s16_run(ix, d, d_q, nq, R, npr, k, w_probes, lco, partial, d_cand, d_ncand, d_total,
									 d_otid, d_odist, d_ocnt, d_probes_in ? (const float *) nullptr : (const float *) ix->w_cdist,
									 cstride);

		if (rc <= 0)
			return rc;
		if (rc == 2)
		{
			/* the sweep served the batch but for ix->redo (a few queries whose records or survivors overflowed): those
			 * alone through the paths below, as a sub-batch of their own, their rows put back where they belong */
			const std::vector<uint32_t> redo = ix->redo;
			const int	nr = (int) redo.size();

			if (g_debug_s16)
				fprintf(stderr, "s16 debug: %d queries go to the exact path alone (first %u), stride %u, partial %d\n", nr, redo[0], stride, partial);
			const uint32_t cap3 = 3u * (uint32_t) k;
			const size_t ob_t = (size_t) nr * k * sizeof(uint64_t), ob_d = (size_t) nr * k * sizeof(float), ob_c = (size_t) nr * sizeof(int),
				ob_cand = (size_t) nr * cap3 * sizeof(ndbhip_cand), ob_tot = (size_t) nr * sizeof(int64_t);
			std::vector<int64_t> idx(redo.begin(), redo.end());

			if (grow(ix->w_redo_q, ix->w_redo_q_n, (size_t) nr * ix->dim)) return NDBHIP_ERR_HIP;
			if (grow(ix->w_redo_idx, ix->w_redo_idx_n, (size_t) nr)) return NDBHIP_ERR_HIP;
			if (grow(ix->w_redo_out, ix->w_redo_out_n, ob_t + ob_d + ob_c + ob_cand + ob_tot + 64)) return NDBHIP_ERR_HIP;
			if (d_probes_in && grow(ix->w_redo_p, ix->w_redo_p_n, (size_t) nr * npr)) return NDBHIP_ERR_HIP;
			HIP_TRY(hipMemcpyAsync(ix->w_redo_idx, idx.data(), (size_t) nr * sizeof(int64_t), hipMemcpyHostToDevice, g.stream));
			hipLaunchKernelGGL(k_rows_pick, dim3(nr), dim3(256), 0, g.stream, (const uint32_t *) d_q, (uint32_t) ix->dim,
							   (const int64_t *) ix->w_redo_idx, (uint32_t *) ix->w_redo_q);
			if (d_probes_in)
				hipLaunchKernelGGL(k_rows_pick, dim3(nr), dim3(64), 0, g.stream, (const uint32_t *) d_probes_in, (uint32_t) npr,
								   (const int64_t *) ix->w_redo_idx, (uint32_t *) ix->w_redo_p);
			HIP_TRY(hipStreamSynchronize(g.stream));		/* idx is a local */
			if (g_debug_s16) fprintf(stderr, "s16 debug: redo queries gathered\n");
			/* (8-byte pieces first: every piece's offset is a multiple of its alignment) */
			unsigned char *ob = ix->w_redo_out;
			uint64_t   *r_tid = (uint64_t *) ob;
			ndbhip_cand *r_cand = (ndbhip_cand *) (ob + ob_t);
			int64_t    *r_tot = (int64_t *) (ob + ob_t + ob_cand);
			float	   *r_dist = (float *) (ob + ob_t + ob_cand + ob_tot);
			int		   *r_cnt = (int *) (ob + ob_t + ob_cand + ob_tot + ob_d);
			/* the buffers above are the mirror's: the recursion must not run into this branch again (allow_s16 = false) */
			const float *rq = ix->w_redo_q;
			const int  *rp = d_probes_in ? (const int *) ix->w_redo_p : (const int *) nullptr;
			const int	rc2 = ivf_search_chunk(ix, rq, nr, strategy, npr, k, max_candidates, stride, full, partial,
											   partial ? r_cand : (ndbhip_cand *) nullptr, partial ? r_cnt : (int *) nullptr,
											   partial ? r_tot : (int64_t *) nullptr, partial ? (uint64_t *) nullptr : r_tid,
											   partial ? (float *) nullptr : r_dist, partial ? (int *) nullptr : r_cnt, rp,
											   (int *) nullptr, false);

			if (rc2)
				return rc2;
			if (g_debug_s16) { (void) hipStreamSynchronize(g.stream); fprintf(stderr, "s16 debug: redo sub-batch done\n"); }
			if (grow(ix->w_redo_idx, ix->w_redo_idx_n, (size_t) nr)) return NDBHIP_ERR_HIP;
			HIP_TRY(hipMemcpyAsync(ix->w_redo_idx, idx.data(), (size_t) nr * sizeof(int64_t), hipMemcpyHostToDevice, g.stream));
			if (partial)
			{
				hipLaunchKernelGGL(k_rows_scatter, dim3(nr), dim3(64), 0, g.stream, (const uint32_t *) r_cand,
								   (uint32_t) (cap3 * sizeof(ndbhip_cand) / 4), (const int64_t *) ix->w_redo_idx, (uint32_t *) d_cand);
				hipLaunchKernelGGL(k_rows_scatter, dim3(nr), dim3(64), 0, g.stream, (const uint32_t *) r_cnt, 1u,
								   (const int64_t *) ix->w_redo_idx, (uint32_t *) d_ncand);
				hipLaunchKernelGGL(k_rows_scatter, dim3(nr), dim3(64), 0, g.stream, (const uint32_t *) r_tot, 2u,
								   (const int64_t *) ix->w_redo_idx, (uint32_t *) d_total);
			}
			else
			{
				hipLaunchKernelGGL(k_rows_scatter, dim3(nr), dim3(64), 0, g.stream, (const uint32_t *) r_tid, (uint32_t) (2 * k),
								   (const int64_t *) ix->w_redo_idx, (uint32_t *) d_otid);
				hipLaunchKernelGGL(k_rows_scatter, dim3(nr), dim3(64), 0, g.stream, (const uint32_t *) r_dist, (uint32_t) k,
								   (const int64_t *) ix->w_redo_idx, (uint32_t *) d_odist);
				hipLaunchKernelGGL(k_rows_scatter, dim3(nr), dim3(64), 0, g.stream, (const uint32_t *) r_cnt, 1u,
								   (const int64_t *) ix->w_redo_idx, (uint32_t *) d_ocnt);
			}
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipStreamSynchronize(g.stream));		/* idx */
			return 0;
		}
		/* some query emitted more than its record capacity: the older path has none */
	}
	/* The paths below keep a [queries x candidates] distance array, which the caller did not size when it expected
	 * the matrix-core screen to serve the batch: sub-batches that fit the array's budget, then the array itself */
	{
		const int	qb_old = ivf_dist_budget_queries(stride, nq);

		if (nq > qb_old)
		{
			const uint32_t cap3 = 3u * (uint32_t) k;

			for (int q0 = 0; q0 < nq; q0 += qb_old)
			{
				const int	n = std::min(qb_old, nq - q0);
				const int	rc = ivf_search_chunk(ix, d_q + (size_t) q0 * ix->dim, n, strategy, npr, k, max_candidates,
												  stride, full, partial, d_cand ? d_cand + (size_t) q0 * cap3 : nullptr,
												  d_ncand ? d_ncand + q0 : nullptr, d_total ? d_total + q0 : nullptr,
												  d_otid ? d_otid + (size_t) q0 * k : nullptr,
												  d_odist ? d_odist + (size_t) q0 * k : nullptr, d_ocnt ? d_ocnt + q0 : nullptr,
												  d_probes_in ? d_probes_in + (size_t) q0 * npr : nullptr,
												  d_probes_out ? d_probes_out + (size_t) q0 * npr : nullptr, false);

				if (rc)
					return rc;
			}
			return 0;
		}
		if (ivf_grow_scan_buffers(ix, nq, stride, npr))
			return NDBHIP_ERR_HIP;
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
			const int	scr_coop = g_scr_coop;
			/* (the fp32 screen — what serves k > 64 — pays from 32 queries up, as measured in round 1) */
			const bool	want = g_scan_mode == 3 || (g_scan_mode == 0 && g_screen_auto && nq >= std::max(g_screen_min_nq, 32));
			const bool	two_tile = scr_coop == 2 && (ix->dim % 16) == 0;

			/* fp16 rows (decoded when the tile is staged), inner product and cosine: the two-tile kernel only */
			screen = want && ((R == R_IVF_L2 && !ix->f16) || two_tile);
			coop = (screen && (ix->dim % 16) == 0) ? scr_coop : 0;
		}

		HIP_TRY(hipMemsetAsync(ix->w_gcnt, 0, (size_t) (2 * nc + 8 * NDB_QHEAD_STRIDE) * sizeof(uint32_t), g.stream));	/* + 8 queue heads */
		hipLaunchKernelGGL(k_pair_count, dim3((npairs + 255) / 256), dim3(256), 0, g.stream,
						   (const int *) w_probes, lco, npr, (uint32_t) nq, cnt);
		hipLaunchKernelGGL(k_pair_offsets, dim3(1), dim3(1024), 0, g.stream, cnt,
						   d.own_len, nc, pair_off, item_off, grp_off, runs, coop ? 4u : 1u, coop == 2 ? 4u : 2u);		/* (items of 128 or 64 rows) */
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
			if (!ix->norm_valid && ivf_frozen(ix))
				return fail(NDBHIP_ERR_STATE, "the mirror is shared (ndbhip_ivf_share): run a batch of this kind on the source before sharing (the row norms are not there yet)");
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
			const int	scr_ch = g_scr_ch;

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
			const int	scr_mfma = g_scr_mfma;
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
	/* sub-batches: what the distance array's budget allows — or, when the matrix-core screen serves the batch (it
	 * keeps no such array), what its 32-bit query-plane offsets allow */
	const size_t dimp4 = (size_t) ((ix->dim + 63) & ~63) * 4;
	const int	qb_s16 = (int) std::min<size_t>(std::min(nq, 65535), (((size_t) 1 << 32) - 1) / dimp4);
	const bool	s16 = ivf_s16_wanted(ix, qb_s16, ivf_recipe(strategy), k);
	const int	qb = s16 ? qb_s16 : ivf_dist_budget_queries(stride, nq);
	const int	ncmp = std::min(ix->nlists, ix->ncent);
	const size_t cstride = (size_t) ((ncmp + 63) & ~63);

	if (grow(ix->w_cdist, ix->w_cdist_n, (size_t) qb * cstride)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_probes, ix->w_probes_n, (size_t) qb * nprobe)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_candoff, ix->w_candoff_n, (size_t) 2 * qb * (nprobe + 1))) return NDBHIP_ERR_HIP;
	if (!s16 && ivf_grow_scan_buffers(ix, qb, stride, nprobe)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_gcnt, ix->w_gcnt_n, (size_t) 2 * ix->ncent + 8 * NDB_QHEAD_STRIDE + 16)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_goff, ix->w_goff_n, (size_t) 3 * (ix->ncent + 1) + 80)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_pairs, ix->w_pairs_n, (size_t) qb * nprobe)) return NDBHIP_ERR_HIP;
	if (grow(ix->w_qnorm, ix->w_qnorm_n, (size_t) 2 * qb)) return NDBHIP_ERR_HIP;

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
	NDB_STAT_ADD(queries, (uint64_t) nq);
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

/* row i of out = the dim floats at base + off[i] (bytes): the device-owner service's queries straight out of its request
 * ring (host memory registered with the device: the reads cross PCIe once, no CPU copy) */
__global__ void
k_gather_mapped(const unsigned char *__restrict__ base, const int64_t *__restrict__ off, int dim, float *__restrict__ out)
{
	const float *src = (const float *) (base + off[blockIdx.x]);
	float	   *dst = out + (size_t) blockIdx.x * dim;

	if ((dim & 3) == 0 && (((uintptr_t) src) & 15) == 0)
		for (int i = threadIdx.x * 4; i < dim; i += blockDim.x * 4)
			*reinterpret_cast<float4 *>(dst + i) = *reinterpret_cast<const float4 *>(src + i);
	else
		for (int i = threadIdx.x; i < dim; i += blockDim.x)
			dst[i] = src[i];
}

/* ndbhip_ivf_search with the queries either in one host array (queries) or scattered in device-visible host memory
 * (d_base + offsets[i]) */
static int
ivf_search_host(ndbhip_ivf *ix, const float *queries, const void *d_base, const int64_t *offsets, int nq, int strategy,
				int nprobe, int k, int64_t max_candidates, uint8_t *out_tids6, float *out_dist, int *out_count)
{
	int			rc = ivf_check_search_args(ix, nq, nprobe, k);

	if (rc)
		return rc;
	if (nq == 0)
		return NDBHIP_OK;
	if ((!queries && (!d_base || !offsets)) || !out_tids6 || !out_dist || !out_count)
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
	const size_t in_bytes = queries ? (size_t) nq * ix->dim * sizeof(float) : (size_t) nq * sizeof(int64_t);
	const size_t pin_bytes = in_bytes + 8 + out_words * 4;

	if (pin_bytes > ix->pin_n)
	{
		if (ix->pin) HIP_TRY(hipHostFree(ix->pin));
		ix->pin = nullptr;
		ix->pin_n = 0;
		HIP_TRY(hipHostMalloc((void **) &ix->pin, pin_bytes, hipHostMallocDefault));
		ix->pin_n = pin_bytes;
	}
	unsigned char *h_out = (unsigned char *) ix->pin + ((in_bytes + 7) & ~(size_t) 7);
	/* ("slow_call_log" = microseconds: a call that takes longer says where the time went — host-side phases, stderr) */
	const auto	lt0 = std::chrono::steady_clock::now();

	if (queries)
	{
		float	   *h_q = (float *) ix->pin;

		memcpy(h_q, queries, in_bytes);
		HIP_TRY(hipMemcpyAsync(ix->w_q, h_q, in_bytes, hipMemcpyHostToDevice, g.stream));
	}
	else
	{
		if (grow(ix->w_qoffs, ix->w_qoffs_n, (size_t) nq)) return NDBHIP_ERR_HIP;
		memcpy(ix->pin, offsets, in_bytes);
		HIP_TRY(hipMemcpyAsync(ix->w_qoffs, ix->pin, in_bytes, hipMemcpyHostToDevice, g.stream));
		hipLaunchKernelGGL(k_gather_mapped, dim3(nq), dim3(256), 0, g.stream, (const unsigned char *) d_base,
						   (const int64_t *) ix->w_qoffs, ix->dim, ix->w_q);
		HIP_TRY(hipGetLastError());
	}
	const auto	lt1 = std::chrono::steady_clock::now();

	rc = ivf_search_device_impl(ix, ix->w_q, nq, strategy, nprobe, k, max_candidates, 0, nullptr, nullptr,
								nullptr, d_tid, d_dist, d_cnt);
	if (rc)
		return rc;
	const auto	lt2 = std::chrono::steady_clock::now();

	HIP_TRY(hipMemcpyAsync(h_out, d_tid, out_words * 4, hipMemcpyDeviceToHost, g.stream));
	const auto	lt3 = std::chrono::steady_clock::now();

	HIP_TRY(hipStreamSynchronize(g.stream));
	if (g_slow_call_us > 0)
	{
		const auto	lt4 = std::chrono::steady_clock::now();
		auto		us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
			return (long) std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
		};

		if (us(lt0, lt4) > g_slow_call_us)
			fprintf(stderr, "ndbhip slow call: %ld us for %d queries: copy in %ld, launches %ld, copy out %ld, wait %ld\n", us(lt0, lt4), nq,
					us(lt0, lt1), us(lt1, lt2), us(lt2, lt3), us(lt3, lt4));
	}
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
ndbhip_ivf_search(ndbhip_ivf *ix, const float *queries, int nq, int strategy, int nprobe, int k,
				  int64_t max_candidates, uint8_t *out_tids6, float *out_dist, int *out_count)
{
	if (nq > 0 && !queries)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	return ivf_search_host(ix, queries, nullptr, nullptr, nq, strategy, nprobe, k, max_candidates, out_tids6, out_dist, out_count);
}

extern "C" int
ndbhip_ivf_search_mapped(ndbhip_ivf *ix, const void *d_base, const int64_t *offsets, int nq, int strategy, int nprobe, int k,
						 int64_t max_candidates, uint8_t *out_tids6, float *out_dist, int *out_count)
{
	if (nq > 0 && (!d_base || !offsets))
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	return ivf_search_host(ix, nullptr, d_base, offsets, nq, strategy, nprobe, k, max_candidates, out_tids6, out_dist, out_count);
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

#include "ndbhip_ops.h"

#include "ndbhip_build.h"

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
