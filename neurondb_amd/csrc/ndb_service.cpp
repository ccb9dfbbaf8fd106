/*
 * ndb_service.cpp — see include/ndb_service.h: a shared-memory submission ring between PostgreSQL backends
 * (clients) and the one process that owns the device and the index mirror.
 */
#include <errno.h>
#include <signal.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/ndb_service.h"

extern "C" int ndbhip_internal_fail(int code, const char *fmt, ...);

namespace
{
enum : uint32_t { S_FREE = 0, S_CLAIMED = 1, S_READY = 2, S_RUNNING = 3, S_DONE = 4 };
/* A slot's state word is (pid of the backend that holds it << 3) | state, 0 = free.  Every transition is a compare-
 * and-swap (or a store by whoever owns the slot in that state) on the WHOLE word, so that nobody can act on a slot
 * that has changed hands since it was looked at: a backend whose half-filled slot was taken back (its pid looked dead,
 * or it was slow) fails to publish it and claims another, instead of publishing into a slot a third backend has
 * claimed meanwhile; the owner's once-a-second sweep frees exactly the holder it examined. */
inline uint32_t st_of(uint32_t w) { return w & 7u; }
inline int32_t pid_of(uint32_t w) { return (int32_t) (w >> 3); }
inline uint32_t word_of(int32_t pid, uint32_t st) { return ((uint32_t) pid << 3) | st; }
const uint32_t MAGIC = 0x4E445356u;		/* "NDSV" */

struct Header
{
	std::atomic<uint32_t> magic;
	uint32_t	dim, max_k, nslots;
	uint64_t	slot_bytes;
	std::atomic<uint32_t> submitted;	/* bumped by every submit: the owner sleeps on it */
	std::atomic<uint32_t> stop;
	std::atomic<uint32_t> cursor;		/* where clients start looking for a free slot */
	std::atomic<uint64_t> seq;			/* arrival order */
	/* which index the owner's mirror is, and at which generation of it (ndb_service_publish); a request for
	 * another key or a newer generation is refused with NDBHIP_ERR_NODEVICE instead of being answered from the
	 * wrong rows */
	std::atomic<uint64_t> index_key, index_version;
	std::atomic<uint64_t> wanted_version;	/* newest generation a backend has asked for (> index_version: reload) */
	std::atomic<int32_t> meta_nprobe;		/* the served index's reloptions / meta-page nprobe (ivf_am.c:1487-1513) */
	std::atomic<int32_t> owner_pid;
	/* bumped once per completed BATCH: the backends of a batch sleep on this one word and are woken by one system call
	 * (a wake-up per slot — a thousand futex calls per thousand-query batch — was most of the owner's time per batch:
	 * 4.35 ms against the 0.6 ms the device needed, profiles/r02_service_bench.txt) */
	std::atomic<uint32_t> completed;
};
static_assert(sizeof(Header) <= 4096, "the header has one page");

bool
pid_gone(int32_t pid)
{
	return pid > 0 && kill((pid_t) pid, 0) != 0 && errno == ESRCH;
}

uint64_t
now_ms()
{
	return (uint64_t) std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Slot
{
	std::atomic<uint32_t> state;
	int32_t		status;
	int32_t		strategy, nprobe, k, count;
	int64_t		max_candidates;
	uint64_t	seq;
	uint64_t	index_key, index_version;	/* what the backend's scan is on (0, 0: whatever the owner serves — only
											 * accepted while the owner itself publishes key 0) */
	int32_t		client_pid;					/* so that the owner can take back the slot of a backend that died */
	uint64_t	t_ms;						/* when the slot was claimed / finished (steady clock) */
	/* followed by: float query[dim]; float dist[max_k]; uint8_t tids6[max_k][6] */
};

static_assert(std::atomic<uint32_t>::is_always_lock_free, "futex words must be plain 32-bit atomics");

size_t
slot_bytes_for(int dim, int max_k)
{
	size_t		b = sizeof(Slot) + (size_t) dim * 4 + (size_t) max_k * 4 + (size_t) max_k * 6;

	return (b + 63) & ~(size_t) 63;
}

long
futex(std::atomic<uint32_t> *addr, int op, uint32_t val, const struct timespec *ts)
{
	return syscall(SYS_futex, (uint32_t *) addr, op, val, ts, nullptr, 0);
}

struct Map
{
	void	   *base = nullptr;
	size_t		bytes = 0;
	Header	   *h = nullptr;
	unsigned char *slots = nullptr;
	Slot *slot(uint32_t i) const { return (Slot *) (slots + (size_t) i * h->slot_bytes); }
	/* one byte per slot behind the slots: "this slot may hold a request" — set by the backend after it published the slot,
	 * cleared by the owner when it takes it.  The owner's scans walk these bytes (nslots / 64 cache lines) instead of one
	 * line per slot, 3.5 KB apart and last written by another core: at 8192 slots the three scans of a poll cost more than
	 * the device's work on the batch.  Only a hint: the state word decides */
	std::atomic<uint8_t> *hint(uint32_t i) const { return (std::atomic<uint8_t> *) (slots + (size_t) h->nslots * h->slot_bytes) + i; }
	float *query(Slot *s) const { return (float *) (s + 1); }
	float *dist(Slot *s) const { return query(s) + h->dim; }
	uint8_t *tids(Slot *s) const { return (uint8_t *) (dist(s) + h->max_k); }
};
}	/* namespace */

struct ndb_service
{
	Map			m;
	std::string name;
};

struct ndb_client
{
	Map			m;
};

extern "C" int
ndb_service_create(const char *name, int dim, int max_k, int nslots, ndb_service **out)
{
	if (!name || name[0] != '/' || !out || dim < 1 || dim > 32767 || max_k < 1 || max_k > NDBHIP_MAX_K || nslots < 1 || nslots > (1 << 20))
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad service arguments (name must start with '/')");
	const size_t sb = slot_bytes_for(dim, max_k);
	const size_t bytes = 4096 + sb * (size_t) nslots + (((size_t) nslots + 63) & ~(size_t) 63);	/* header, slots, hints */

	(void) shm_unlink(name);
	const int	fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);

	if (fd < 0 || ftruncate(fd, (off_t) bytes) != 0)
	{
		if (fd >= 0) close(fd);
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "shm_open(%s): %s", name, strerror(errno));
	}
	void	   *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);

	close(fd);
	if (p == MAP_FAILED)
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "mmap(%s): %s", name, strerror(errno));
	memset(p, 0, bytes);
	ndb_service *s = new (std::nothrow) ndb_service();

	if (!s)
	{
		munmap(p, bytes);
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "out of host memory");
	}
	s->name = name;
	s->m.base = p;
	s->m.bytes = bytes;
	s->m.h = (Header *) p;
	s->m.slots = (unsigned char *) p + 4096;
	s->m.h->dim = (uint32_t) dim;
	s->m.h->max_k = (uint32_t) max_k;
	s->m.h->nslots = (uint32_t) nslots;
	s->m.h->slot_bytes = sb;
	s->m.h->owner_pid.store((int32_t) getpid());
	s->m.h->meta_nprobe.store(0);
	s->m.h->magic.store(MAGIC, std::memory_order_release);
	*out = s;
	return NDBHIP_OK;
}

extern "C" int
ndb_service_destroy(ndb_service *s)
{
	if (!s)
		return NDBHIP_OK;
	s->m.h->stop.store(1);
	s->m.h->magic.store(0);
	munmap(s->m.base, s->m.bytes);
	(void) shm_unlink(s->name.c_str());
	delete s;
	return NDBHIP_OK;
}

extern "C" int
ndb_service_stop(ndb_service *s)
{
	if (!s)
		return NDBHIP_ERR_INVALID;
	s->m.h->stop.store(1);
	s->m.h->submitted.fetch_add(1);
	futex(&s->m.h->submitted, FUTEX_WAKE, 1 << 30, nullptr);
	return NDBHIP_OK;
}

extern "C" int
ndb_service_stopped(const ndb_service *s)
{
	return s ? (int) s->m.h->stop.load() : 1;
}

extern "C" int
ndb_service_publish(ndb_service *s, uint64_t index_key, uint64_t index_version, int meta_nprobe)
{
	if (!s)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "service is NULL");
	Header	   *h = s->m.h;

	h->meta_nprobe.store(meta_nprobe);
	h->index_key.store(index_key, std::memory_order_release);
	h->index_version.store(index_version, std::memory_order_release);
	return NDBHIP_OK;
}

extern "C" int
ndb_service_reload_wanted(const ndb_service *s, uint64_t *version)
{
	if (!s)
		return 0;
	const uint64_t w = s->m.h->wanted_version.load(std::memory_order_acquire);

	if (version)
		*version = w;
	return w > s->m.h->index_version.load(std::memory_order_acquire) ? 1 : 0;
}

/* Slots whose backend is gone (SIGKILLed between claim and read, or gave up on a RUNNING ticket and exited):
 * CLAIMED / DONE slots of a dead pid, and DONE slots nobody read for a minute, go back to FREE.  Owner only. */
static int
reclaim_slots(ndb_service *s)
{
	Header	   *h = s->m.h;
	const uint64_t t = now_ms();
	int			n = 0;

	for (uint32_t i = 0; i < h->nslots; i++)
	{
		Slot	   *sl = s->m.slot(i);
		uint32_t	w = sl->state.load(std::memory_order_acquire);
		const uint32_t st = st_of(w);

		if (st != S_CLAIMED && st != S_DONE && st != S_READY)
			continue;
		/* (t_ms of a DONE slot is the owner's own stamp; a CLAIMED slot's holder is judged by its pid alone — the pid
		 * is part of the word, so it is the holder's, not a predecessor's) */
		if (!(pid_gone(pid_of(w)) || (st == S_DONE && t - sl->t_ms > 60000)))
			continue;
		if (sl->state.compare_exchange_strong(w, S_FREE, std::memory_order_acq_rel))
			n++;
	}
	return n;
}

extern "C" int
ndb_service_reclaim(ndb_service *s)
{
	if (!s)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "service is NULL");
	return reclaim_slots(s);
}

extern "C" int
ndb_service_poll(ndb_service *s, int max_batch, int wait_us, int linger_us, int *slot_ids, float *queries,
				 int *strategy, int *nprobe, int *k, int64_t *max_candidates)
{
	if (!s || max_batch < 1 || !slot_ids || !strategy || !nprobe || !k || !max_candidates)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad poll arguments");
	Header	   *h = s->m.h;
	const auto	t0 = std::chrono::steady_clock::now();
	auto		us_since = [&](std::chrono::steady_clock::time_point t) {
		return (int64_t) std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count();
	};
	int			n = 0;
	bool		have_key = false;
	auto		last_gain = t0;
	static thread_local uint64_t last_reclaim = 0;

	if (now_ms() - last_reclaim > 1000)
	{
		last_reclaim = now_ms();
		(void) reclaim_slots(s);
	}
	for (;;)
	{
		if (h->stop.load(std::memory_order_acquire) && n == 0)
			return 0;
		const uint32_t seen = h->submitted.load(std::memory_order_acquire);
		const uint64_t skey = h->index_key.load(std::memory_order_acquire), sver = h->index_version.load(std::memory_order_acquire);

		/* requests on another index, or on a generation of this one the mirror does not hold: refused now
		 * (the backend runs its CPU scan, as after a failed ndb_gpu_init_if_needed), never answered from the
		 * wrong rows */
		for (uint32_t i = 0; i < h->nslots; i++)
		{
			if (!s->m.hint(i)->load(std::memory_order_acquire))
				continue;
			Slot	   *sl = s->m.slot(i);
			uint32_t	w = sl->state.load(std::memory_order_acquire);

			if (st_of(w) != S_READY)
			{
				/* a stale hint (the backend withdrew its request): cleared — and set again should the slot have been
				 * published in between (the backend stores the state first, the hint second).  The clear and the re-read
				 * are sequentially consistent (as are the backend's publish and its hint): with release / acquire alone
				 * the re-read may be satisfied BEFORE the clear becomes visible (store -> load reordering, legal and real
				 * on x86), the backend's state = READY, hint = 1 slip in between, and the clear then lands on top of the
				 * fresh hint — a READY slot no scan would ever look at again */
				s->m.hint(i)->exchange(0, std::memory_order_seq_cst);
				if (st_of(sl->state.load(std::memory_order_seq_cst)) == S_READY)
					s->m.hint(i)->store(1, std::memory_order_seq_cst);
				continue;
			}
			/* (generation 0 of a named index = its backend could not learn the generation, ndb_gen_get: never served) */
			if (st_of(w) != S_READY || (sl->index_key == skey && sl->index_version == sver && !(sl->index_key != 0 && sl->index_version == 0)))
				continue;
			const uint64_t want = sl->index_version;
			const bool	same_key = sl->index_key == skey;
			const int32_t holder = pid_of(w);

			if (!sl->state.compare_exchange_strong(w, word_of(holder, S_RUNNING), std::memory_order_acq_rel))
				continue;
			s->m.hint(i)->store(0, std::memory_order_release);
			if (same_key)
			{
				uint64_t	w = h->wanted_version.load();

				while (want > w && !h->wanted_version.compare_exchange_weak(w, want))
					;
			}
			sl->status = NDBHIP_ERR_NODEVICE;
			sl->count = 0;
			sl->t_ms = now_ms();
			sl->state.store(word_of(holder, S_DONE), std::memory_order_release);
			h->completed.fetch_add(1, std::memory_order_release);
			futex(&h->completed, FUTEX_WAKE, 1 << 30, nullptr);
		}
		/* oldest first: a backend must not starve behind newer arrivals with another parameter set */
		if (!have_key)
		{
			uint64_t	best = ~0ull;
			int			bi = -1;

			for (uint32_t i = 0; i < h->nslots; i++)
			{
				if (!s->m.hint(i)->load(std::memory_order_acquire))
					continue;
				Slot	   *sl = s->m.slot(i);

				if (st_of(sl->state.load(std::memory_order_acquire)) == S_READY && sl->seq < best)
				{
					best = sl->seq;
					bi = (int) i;
				}
			}
			if (bi >= 0)
			{
				Slot	   *sl = s->m.slot((uint32_t) bi);

				*strategy = sl->strategy; *nprobe = sl->nprobe; *k = sl->k; *max_candidates = sl->max_candidates;
				have_key = true;
			}
		}
		int			gained = 0;

		if (have_key)
			for (uint32_t i = 0; i < h->nslots && n < max_batch; i++)
			{
				if (!s->m.hint(i)->load(std::memory_order_acquire))
					continue;
				Slot	   *sl = s->m.slot(i);
				uint32_t	w = sl->state.load(std::memory_order_acquire);

				if (st_of(w) != S_READY)
					continue;
				if (sl->strategy != *strategy || sl->nprobe != *nprobe || sl->k != *k || sl->max_candidates != *max_candidates)
					continue;
				if (!sl->state.compare_exchange_strong(w, word_of(pid_of(w), S_RUNNING), std::memory_order_acq_rel))
					continue;
				s->m.hint(i)->store(0, std::memory_order_release);
				if (queries)
					memcpy(queries + (size_t) n * h->dim, s->m.query(sl), (size_t) h->dim * 4);
				slot_ids[n++] = (int) i;
				gained++;
			}
		if (gained)
			last_gain = std::chrono::steady_clock::now();
		else if (n == 0)
			have_key = false;	/* the slot the parameters came from was withdrawn: look again */
		if (n >= max_batch)
			return n;
		if (n > 0)
		{
			if (us_since(last_gain) >= linger_us)
				return n;
			/* linger: a short sleep, not a futex — arrivals are expected within microseconds */
			struct timespec ts = {0, 5 * 1000};

			nanosleep(&ts, nullptr);
			continue;
		}
		const int64_t left = (int64_t) wait_us - us_since(t0);

		if (left <= 0)
			return 0;
		struct timespec ts = {(time_t) (left / 1000000), (long) (left % 1000000) * 1000};

		futex(&h->submitted, FUTEX_WAIT, seen, &ts);	/* returns at once if something was submitted meanwhile */
	}
}

extern "C" int64_t
ndb_service_query_offset(const ndb_service *s, int slot_id)
{
	if (!s || slot_id < 0 || (uint32_t) slot_id >= s->m.h->nslots)
		return -1;
	return (int64_t) ((unsigned char *) s->m.query(s->m.slot((uint32_t) slot_id)) - (unsigned char *) s->m.base);
}

extern "C" int
ndb_service_segment(const ndb_service *s, void **base, size_t *bytes)
{
	if (!s || !base || !bytes)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad arguments");
	*base = s->m.base;
	*bytes = s->m.bytes;
	return NDBHIP_OK;
}

extern "C" int
ndb_service_complete(ndb_service *s, int n, const int *slot_ids, const uint8_t *tids6, const float *dist,
					 const int *count, int k, int status)
{
	if (!s || n < 0 || (n > 0 && !slot_ids))
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad complete arguments");
	Header	   *h = s->m.h;

	for (int i = 0; i < n; i++)
	{
		Slot	   *sl = s->m.slot((uint32_t) slot_ids[i]);

		sl->status = status;
		sl->count = 0;
		if (status == 0 && tids6 && dist && count)
		{
			const int	c = count[i] < (int) h->max_k ? count[i] : (int) h->max_k;

			sl->count = c;
			memcpy(s->m.dist(sl), dist + (size_t) i * k, (size_t) c * 4);
			memcpy(s->m.tids(sl), tids6 + (size_t) i * k * 6, (size_t) c * 6);
		}
		sl->t_ms = now_ms();
		/* (RUNNING slots are the owner's: the holder's pid stays in the word) */
		sl->state.store(word_of(pid_of(sl->state.load(std::memory_order_relaxed)), S_DONE), std::memory_order_release);
	}
	if (n > 0)
	{
		h->completed.fetch_add(1, std::memory_order_release);
		futex(&h->completed, FUTEX_WAKE, 1 << 30, nullptr);		/* every sleeper looks at its own slot again */
	}
	return NDBHIP_OK;
}

extern "C" int
ndb_service_serve_ivf(ndb_service *s, ndbhip_ivf *ix, int max_batch, int linger_us, int64_t max_batches,
					  ndb_service_stats *stats)
{
	if (!s || !ix || max_batch < 1)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad serve arguments");
	Header	   *h = s->m.h;

	if (ndbhip_ivf_dim(ix) != (int) h->dim)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "service dim %u != index dim %d", h->dim, ndbhip_ivf_dim(ix));
	/* the batch's queries and results in PINNED host memory: ndbhip_ivf_search's copies are then one DMA each instead of the
	 * runtime's staged copy of pageable memory (3 MB of queries per 1024-query batch) */
	struct Pinned
	{
		void	   *p = nullptr;
		bool		pinned = false;
		explicit Pinned(size_t bytes)
		{
			if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess)
				pinned = true;
			else
			{
				(void) hipGetLastError();
				p = malloc(bytes ? bytes : 1);
			}
		}
		~Pinned() { if (pinned) (void) hipHostFree(p); else free(p); }
	};
	std::vector<int> ids((size_t) max_batch);
	Pinned		q_mem((size_t) max_batch * h->dim * 4), dist_mem((size_t) max_batch * h->max_k * 4),
		tids_mem((size_t) max_batch * h->max_k * 6), cnt_mem((size_t) max_batch * 4);

	if (!q_mem.p || !dist_mem.p || !tids_mem.p || !cnt_mem.p)
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "out of host memory");
	struct { void *p; float *data() { return (float *) p; } } q{q_mem.p}, dist{dist_mem.p};
	struct { void *p; uint8_t *data() { return (uint8_t *) p; } } tids{tids_mem.p};
	struct { void *p; int *data() { return (int *) p; } } cnt{cnt_mem.p};
	ndb_service_stats st = {0, 0, 0, 0.0};
	int			rc_last = NDBHIP_OK;
	/* the ring registered with the device: a batch's queries are gathered by a kernel straight out of the slots (3 MB per
	 * 1024 queries that no CPU copies twice on the critical path); where that fails the queries are copied as before */
	void	   *d_ring = nullptr;
	bool		ring_reg = false;

	/* (the poll's linger sleeps 5 us at a time: with the default 50 us timer slack each of those is ten times as long) */
	/* (the calling thread's own setting is put back before this returns: a library call leaves no process-visible change) */
	const int	old_slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);

	(void) prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
	std::vector<int64_t> offs((size_t) max_batch);

	if (hipHostRegister(s->m.base, s->m.bytes, hipHostRegisterMapped) == hipSuccess)
	{
		ring_reg = true;
		if (hipHostGetDevicePointer(&d_ring, s->m.base, 0) != hipSuccess)
			d_ring = nullptr;
	}
	if (!d_ring)
		(void) hipGetLastError();

	while (!h->stop.load(std::memory_order_acquire) && (max_batches <= 0 || (int64_t) st.batches < max_batches))
	{
		if (h->wanted_version.load(std::memory_order_acquire) > h->index_version.load(std::memory_order_acquire))
			break;				/* the index has moved on: the owner reloads its mirror, publishes, and serves again */
		int			strategy, nprobe, k;
		int64_t		cap;
		const int	n = ndb_service_poll(s, max_batch, 50 * 1000, linger_us, ids.data(), d_ring ? nullptr : q.data(), &strategy, &nprobe, &k, &cap);

		if (n <= 0)
			continue;
		const auto	t0 = std::chrono::steady_clock::now();
		int			rc;

		if (k < 1 || k > (int) h->max_k)
			rc = ndbhip_internal_fail(NDBHIP_ERR_INVALID, "k %d beyond the service's max_k %u", k, h->max_k);
		else if (d_ring)
		{
			for (int i = 0; i < n; i++)
				offs[(size_t) i] = ndb_service_query_offset(s, ids[(size_t) i]);
			rc = ndbhip_ivf_search_mapped(ix, d_ring, offs.data(), n, strategy, nprobe, k, cap, tids.data(), dist.data(), cnt.data());
		}
		else
			rc = ndbhip_ivf_search(ix, q.data(), n, strategy, nprobe, k, cap, tids.data(), dist.data(), cnt.data());

		ndb_service_complete(s, n, ids.data(), tids.data(), dist.data(), cnt.data(), k, rc);
		st.busy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		st.batches++;
		st.queries += (uint64_t) n;
		if ((uint64_t) n > st.max_batch)
			st.max_batch = (uint64_t) n;
		if (rc)
			rc_last = rc;
	}
	if (ring_reg)
		(void) hipHostUnregister(s->m.base);
	if (old_slack > 0)
		(void) prctl(PR_SET_TIMERSLACK, (unsigned long) old_slack, 0, 0, 0);
	if (stats)
		*stats = st;
	return rc_last;
}

/* ------------------------------------------------------------------ */
/* backend side                                                        */
/* ------------------------------------------------------------------ */
extern "C" int
ndb_client_connect(const char *name, ndb_client **out)
{
	if (!name || !out)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad connect arguments");
	const int	fd = shm_open(name, O_RDWR, 0600);
	struct stat st;

	if (fd < 0 || fstat(fd, &st) != 0 || (size_t) st.st_size < 4096)
	{
		if (fd >= 0) close(fd);
		return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "no service segment %s", name);
	}
	void	   *p = mmap(nullptr, (size_t) st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);

	close(fd);
	if (p == MAP_FAILED)
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "mmap(%s): %s", name, strerror(errno));
	Header	   *h = (Header *) p;

	if (h->magic.load(std::memory_order_acquire) != MAGIC ||
		4096 + h->slot_bytes * (size_t) h->nslots + (size_t) h->nslots > (size_t) st.st_size)
	{
		munmap(p, (size_t) st.st_size);
		return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "service segment %s is not initialised", name);
	}
	ndb_client *c = new (std::nothrow) ndb_client();

	if (!c)
	{
		munmap(p, (size_t) st.st_size);
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "out of host memory");
	}
	c->m.base = p;
	c->m.bytes = (size_t) st.st_size;
	c->m.h = h;
	c->m.slots = (unsigned char *) p + 4096;
	*out = c;
	return NDBHIP_OK;
}

extern "C" int
ndb_client_disconnect(ndb_client *c)
{
	if (!c)
		return NDBHIP_OK;
	munmap(c->m.base, c->m.bytes);
	delete c;
	return NDBHIP_OK;
}

extern "C" int
ndb_client_dim(const ndb_client *c)
{
	return c ? (int) c->m.h->dim : NDBHIP_ERR_INVALID;
}

extern "C" int
ndb_client_meta_nprobe(const ndb_client *c)
{
	return c ? (int) c->m.h->meta_nprobe.load() : NDBHIP_ERR_INVALID;
}

extern "C" int
ndb_client_index(const ndb_client *c, uint64_t *index_key, uint64_t *index_version)
{
	if (!c)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "client is NULL");
	if (index_key)
		*index_key = c->m.h->index_key.load(std::memory_order_acquire);
	if (index_version)
		*index_version = c->m.h->index_version.load(std::memory_order_acquire);
	return NDBHIP_OK;
}

extern "C" int
ndb_client_stop_service(ndb_client *c)
{
	if (!c)
		return NDBHIP_ERR_INVALID;
	c->m.h->stop.store(1);
	c->m.h->submitted.fetch_add(1);
	futex(&c->m.h->submitted, FUTEX_WAKE, 1 << 30, nullptr);
	return NDBHIP_OK;
}

extern "C" int
ndb_client_submit(ndb_client *c, const float *query, int strategy, int nprobe, int k, int64_t max_candidates, int *ticket)
{
	return ndb_client_submit_index(c, 0, 0, query, strategy, nprobe, k, max_candidates, ticket);
}

extern "C" int
ndb_client_submit_index(ndb_client *c, uint64_t index_key, uint64_t index_version, const float *query, int strategy,
						int nprobe, int k, int64_t max_candidates, int *ticket)
{
	if (!c || !query || !ticket)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad submit arguments");
	Header	   *h = c->m.h;

	if (h->magic.load(std::memory_order_acquire) != MAGIC || h->stop.load() || pid_gone(h->owner_pid.load()))
		return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "the device service is gone");
	if (h->index_key.load(std::memory_order_acquire) != index_key)
		return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "the device service holds another index (key %llu, this scan is on %llu)",
									(unsigned long long) h->index_key.load(), (unsigned long long) index_key);
	if (h->index_version.load(std::memory_order_acquire) != index_version)
	{
		/* tell the owner (its serve loop returns for a reload); this scan runs on the CPU */
		uint64_t	w = h->wanted_version.load();

		while (index_version > w && !h->wanted_version.compare_exchange_weak(w, index_version))
			;
		h->submitted.fetch_add(1, std::memory_order_release);
		futex(&h->submitted, FUTEX_WAKE, 1, nullptr);
		return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "the device service holds generation %llu of the index, this scan needs %llu",
									(unsigned long long) h->index_version.load(), (unsigned long long) index_version);
	}
	if (k < 1 || k > (int) h->max_k)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "k %d beyond the service's max_k %u", k, h->max_k);
	/* a free slot: start at the shared cursor, bounded number of rounds */
	for (int round = 0; round < 2000; round++)
	{
		const uint32_t start = h->cursor.fetch_add(1, std::memory_order_relaxed);

		for (uint32_t j = 0; j < h->nslots; j++)
		{
			const uint32_t i = (start + j) % h->nslots;
			Slot	   *sl = c->m.slot(i);
			uint32_t	st = S_FREE;
			const int32_t me = (int32_t) getpid();

			if (sl->state.load(std::memory_order_relaxed) != S_FREE ||
				!sl->state.compare_exchange_strong(st, word_of(me, S_CLAIMED), std::memory_order_acq_rel))
				continue;
			memcpy(c->m.query(sl), query, (size_t) h->dim * 4);
			sl->strategy = strategy;
			sl->nprobe = nprobe;
			sl->k = k;
			sl->max_candidates = max_candidates;
			sl->index_key = index_key;
			sl->index_version = index_version;
			sl->client_pid = (int32_t) getpid();
			sl->t_ms = now_ms();
			sl->status = 0;
			sl->count = 0;
			sl->seq = h->seq.fetch_add(1, std::memory_order_relaxed);
			{
				/* published only if the slot is still this backend's claim: the owner may have taken it back (this
				 * process looked dead to it, e.g. across a pid namespace) and somebody else may hold it now */
				uint32_t	mine = word_of(me, S_CLAIMED);

				if (!sl->state.compare_exchange_strong(mine, word_of(me, S_READY), std::memory_order_seq_cst))
					continue;
			}
			c->m.hint(i)->store(1, std::memory_order_seq_cst);		/* (seq_cst with the owner's stale-hint repair: see there) */
			h->submitted.fetch_add(1, std::memory_order_release);
			futex(&h->submitted, FUTEX_WAKE, 1, nullptr);
			*ticket = (int) i;
			return NDBHIP_OK;
		}
		struct timespec ts = {0, 50 * 1000};	/* every slot busy: more backends than slots */

		nanosleep(&ts, nullptr);
	}
	return ndbhip_internal_fail(NDBHIP_ERR_STATE, "no free request slot");
}

extern "C" int
ndb_client_wait(ndb_client *c, int ticket, uint8_t *tids6, float *dist, int *count, int timeout_ms)
{
	if (!c || ticket < 0 || (uint32_t) ticket >= c->m.h->nslots || !count)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad wait arguments");
	Header	   *h = c->m.h;
	Slot	   *sl = c->m.slot((uint32_t) ticket);
	const auto	t0 = std::chrono::steady_clock::now();

	for (int spins = 0;; spins++)
	{
		/* (read before the slot: a batch completed between the two reads makes the sleep below return at once) */
		const uint32_t seen_done = h->completed.load(std::memory_order_acquire);
		const uint32_t w = sl->state.load(std::memory_order_acquire);
		const uint32_t st = st_of(w);

		if (pid_of(w) != (int32_t) getpid() && w != S_FREE)
			return ndbhip_internal_fail(NDBHIP_ERR_STATE, "ticket %d belongs to another backend now", ticket);
		if (st == S_DONE)
			break;
		if (st != S_READY && st != S_RUNNING)
			return ndbhip_internal_fail(NDBHIP_ERR_STATE, "ticket %d is not in flight", ticket);
		const int64_t waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();

		if ((timeout_ms >= 0 && waited >= timeout_ms) || (h->stop.load() && st == S_READY))
		{
			/* give the slot back only if the owner has not taken it; a RUNNING slot is the owner's */
			uint32_t	exp = word_of((int32_t) getpid(), S_READY);

			if (sl->state.compare_exchange_strong(exp, S_FREE))
				return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "the device service did not answer");
			if (timeout_ms >= 0 && waited >= 4 * (int64_t) timeout_ms + 1000)
				return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "the device service took the request and never finished it");
		}
		if ((spins & 63) == 63 && pid_gone(h->owner_pid.load()))
		{
			/* the owner was killed: nobody will finish (or reclaim) this slot; a new owner starts from a fresh segment */
			uint32_t	exp = w;

			(void) sl->state.compare_exchange_strong(exp, S_FREE);
			return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "the device service's process is gone");
		}
		if (spins < 200)
			continue;			/* a batch returns within a fraction of a millisecond: spin first */
		struct timespec ts = {0, 2 * 1000 * 1000};

		futex(&h->completed, FUTEX_WAIT, seen_done, &ts);
	}
	const int	rc = sl->status;
	const int	n = sl->count;

	*count = n;
	if (rc == 0)
	{
		if (dist)
			memcpy(dist, c->m.dist(sl), (size_t) n * 4);
		if (tids6)
			memcpy(tids6, c->m.tids(sl), (size_t) n * 6);
	}
	{
		/* given back only if it is still this backend's finished request: the owner takes DONE slots back after a minute
		 * (a backend stopped in a debugger, say), and then what was copied above may be somebody else's */
		uint32_t	mine = word_of((int32_t) getpid(), S_DONE);

		if (!sl->state.compare_exchange_strong(mine, S_FREE, std::memory_order_acq_rel))
			return ndbhip_internal_fail(NDBHIP_ERR_NODEVICE, "the device service took ticket %d back before its answer was read", ticket);
	}
	if (rc)
		return ndbhip_internal_fail(rc, "the device service reported error %d for this query", rc);
	return NDBHIP_OK;
}

extern "C" int
ndb_client_search(ndb_client *c, const float *query, int strategy, int nprobe, int k, int64_t max_candidates,
				  uint8_t *tids6, float *dist, int *count, int timeout_ms)
{
	return ndb_client_search_index(c, 0, 0, query, strategy, nprobe, k, max_candidates, tids6, dist, count, timeout_ms);
}

extern "C" int
ndb_client_search_index(ndb_client *c, uint64_t index_key, uint64_t index_version, const float *query, int strategy,
						int nprobe, int k, int64_t max_candidates, uint8_t *tids6, float *dist, int *count, int timeout_ms)
{
	int			ticket = -1;
	const int	rc = ndb_client_submit_index(c, index_key, index_version, query, strategy, nprobe, k, max_candidates, &ticket);

	if (rc)
		return rc;
	return ndb_client_wait(c, ticket, tids6, dist, count, timeout_ms);
}

/* ------------------------------------------------------------------ */
/* index generations                                                   */
/* ------------------------------------------------------------------ */
/*
 * A counter per index that every aminsert / ambulkdelete of every backend bumps and every scan reads: the
 * version stamp of the device mirrors.  It never repeats (the reference's meta->insertedVectors goes down again
 * in ivfbulkdelete, ivf_am.c:1346, and its pages carry no LSN — every change is MarkBufferDirty without WAL,
 * ivf_am.c:1137-1156 — so nothing on the pages can serve).  A POSIX shm segment with an open-addressing table of
 * (key, generation) pairs; lock-free: a key is claimed by CAS on an empty cell and never removed.
 */
namespace
{
const uint32_t GEN_MAGIC = 0x4E444753u;	/* "NDGS" */

struct GenHeader
{
	std::atomic<uint32_t> magic;
	uint32_t	ncells;
};

struct GenCell
{
	std::atomic<uint64_t> key;		/* 0 = empty */
	std::atomic<uint64_t> gen;
};
}

struct ndb_gen
{
	void	   *base = nullptr;
	size_t		bytes = 0;
	GenHeader  *h = nullptr;
	GenCell    *cells = nullptr;
};

extern "C" int
ndb_gen_attach(const char *name, int ncells, ndb_gen **out)
{
	if (!name || name[0] != '/' || !out || ncells < 16 || ncells > (1 << 24) || (ncells & (ncells - 1)))
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad generation-table arguments (name \"/...\", ncells a power of two >= 16)");
	const size_t bytes = 4096 + sizeof(GenCell) * (size_t) ncells;
	bool		creator = true;
	int			fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);

	if (fd < 0 && errno == EEXIST)
	{
		creator = false;
		fd = shm_open(name, O_RDWR, 0600);
	}
	if (fd < 0)
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "shm_open(%s): %s", name, strerror(errno));
	if (creator && ftruncate(fd, (off_t) bytes) != 0)
	{
		close(fd);
		(void) shm_unlink(name);
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "ftruncate(%s): %s", name, strerror(errno));
	}
	struct stat st;

	/* a late attacher may see the segment before the creator has sized it */
	for (int tries = 0; tries < 2000; tries++)
	{
		if (fstat(fd, &st) == 0 && (size_t) st.st_size >= 4096 + sizeof(GenCell) * 16)
			break;
		struct timespec ts = {0, 1000 * 1000};

		nanosleep(&ts, nullptr);
	}
	if (fstat(fd, &st) != 0 || (size_t) st.st_size < 4096 + sizeof(GenCell) * 16)
	{
		close(fd);
		return ndbhip_internal_fail(NDBHIP_ERR_STATE, "generation table %s was never sized", name);
	}
	void	   *p = mmap(nullptr, (size_t) st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);

	close(fd);
	if (p == MAP_FAILED)
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "mmap(%s): %s", name, strerror(errno));
	GenHeader  *h = (GenHeader *) p;

	if (creator)
	{
		h->ncells = (uint32_t) ncells;	/* (the segment is zero-filled: every cell empty) */
		h->magic.store(GEN_MAGIC, std::memory_order_release);
	}
	else
	{
		for (int tries = 0; tries < 2000 && h->magic.load(std::memory_order_acquire) != GEN_MAGIC; tries++)
		{
			struct timespec ts = {0, 1000 * 1000};

			nanosleep(&ts, nullptr);
		}
		if (h->magic.load(std::memory_order_acquire) != GEN_MAGIC ||
			4096 + sizeof(GenCell) * (size_t) h->ncells > (size_t) st.st_size)
		{
			munmap(p, (size_t) st.st_size);
			return ndbhip_internal_fail(NDBHIP_ERR_STATE, "generation table %s is not initialised", name);
		}
	}
	ndb_gen    *g = new (std::nothrow) ndb_gen();

	if (!g)
	{
		munmap(p, (size_t) st.st_size);
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "out of host memory");
	}
	g->base = p;
	g->bytes = (size_t) st.st_size;
	g->h = h;
	g->cells = (GenCell *) ((unsigned char *) p + 4096);
	*out = g;
	return NDBHIP_OK;
}

extern "C" int
ndb_gen_detach(ndb_gen *g, const char *unlink_name)
{
	if (g)
	{
		munmap(g->base, g->bytes);
		delete g;
	}
	if (unlink_name)
		(void) shm_unlink(unlink_name);
	return NDBHIP_OK;
}

/* the cell of `key`; NULL when it has none — *full says whether that is because the table has no empty cell left (the
 * key may have been changed without anybody being able to count it) */
static GenCell *
gen_cell(ndb_gen *g, uint64_t key, bool create, bool *full = nullptr)
{
	if (full)
		*full = false;
	const uint32_t mask = g->h->ncells - 1;
	uint64_t	x = key * 0x9E3779B97F4A7C15ull;
	uint32_t	i = (uint32_t) (x >> 32) & mask;

	for (uint32_t probe = 0; probe <= mask; probe++, i = (i + 1) & mask)
	{
		GenCell    *c = &g->cells[i];
		uint64_t	k = c->key.load(std::memory_order_acquire);

		if (k == key)
			return c;
		if (k == 0)
		{
			if (!create)
				return nullptr;
			if (c->key.compare_exchange_strong(k, key, std::memory_order_acq_rel) || k == key)
				return c;
		}
	}
	if (full)
		*full = true;
	return nullptr;
}

/* generation of `key` (an index's relfilenode / OID, != 0); 1 for an index nobody has changed since the table
 * was created; 0 = UNKNOWN — no table, or the key has no cell and the table is full, so its changes could not be
 * counted: a caller must treat every mirror of it as stale (reload for every scan), never compare 0 with 0 */
extern "C" uint64_t
ndb_gen_get(ndb_gen *g, uint64_t key)
{
	if (!g || key == 0)
		return 0;
	bool		full = false;
	GenCell    *c = gen_cell(g, key, false, &full);

	if (c)
		return c->gen.load(std::memory_order_acquire) + 1;
	return full ? 0 : 1;
}

/* one more change to `key`: returns the new generation (0: the table is full — callers then treat every scan
 * as stale, i.e. rebuild their mirror each time) */
extern "C" uint64_t
ndb_gen_bump(ndb_gen *g, uint64_t key)
{
	if (!g || key == 0)
		return 0;
	GenCell    *c = gen_cell(g, key, true);

	return c ? c->gen.fetch_add(1, std::memory_order_acq_rel) + 2 : 0;
}
