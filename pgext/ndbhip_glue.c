/*
 * pgext/ndbhip_glue.c — the PostgreSQL side of the drop-in boundary.
 *
 * Lives in the reference tree as src/index/ndbhip_glue.c (pgext/Makefile copies / builds it when pg_config
 * exists; the build image of this repository has no PostgreSQL, so it is compile-checked only where one is
 * installed).  Everything below the C ABI (include/ndbhip.h, ndb_am.h, ndb_service.h, ndb_backend.h) is
 * exercised PostgreSQL-free by tests/ and bench.py; this file is the thin forwarding layer:
 *
 *   GUCs            neurondb.ivf_probes / ivf_k / hnsw_ef_search / hnsw_k / ref_compat -> ndb_am_set_guc
 *                   neurondb.device_service (string)                                   -> ndb_am_use_service
 *   lazy init       ndb_hip_ready(): the contract of ndb_gpu_init_if_needed (src/gpu/common/gpu_core.c:240-310)
 *   mirror cache    ndb_hip_ivf_mirror(Relation): index pages -> ndbhip_ivf_load_pages, keyed by (relid, stamp);
 *                   the stamp is the index's generation in a shared counter table (ndb_gen_*, include/ndb_service.h)
 *                   that every backend's aminsert / ambulkdelete bumps — it never repeats
 *   AM callbacks    ndbhip_ivfrescan / ndbhip_ivfgettuple / ndbhip_ivfendscan (and hnsw): called from the
 *                   handlers of src/index/ivf_am.c:385-435 / hnsw_am.c:287-338 (pgext/reference.patch)
 *   plugin vtable   ndbhip_register_backend(): memcpy of the prefix image + ndb_gpu_register_backend
 */
#include "postgres.h"

#include "access/genam.h"
#include "access/relscan.h"
#include "fmgr.h"
#include "storage/bufmgr.h"
#include "utils/guc.h"
#include "utils/memutils.h"
#include "utils/rel.h"

#include "neurondb_gpu_backend.h"	/* the reference's: struct ndb_gpu_backend, ndb_gpu_register_backend */

#include "ndbhip.h"
#include "ndb_am.h"
#include "ndb_backend.h"
#include "ndb_service.h"

extern int	neurondb_compute_mode;	/* 0 cpu, 1 gpu, 2 auto: src/util/neurondb_guc.c:213 */
extern int	neurondb_gpu_device;

/* ---- the prefix image really is one (a changed reference header fails the build, not a query) ---- */
#define SAME_OFFSET(m) _Static_assert(offsetof(ndb_hip_backend, m) == offsetof(ndb_gpu_backend, m), "ndb_hip_backend." #m)
SAME_OFFSET(name); SAME_OFFSET(provider); SAME_OFFSET(kind); SAME_OFFSET(features); SAME_OFFSET(priority);
SAME_OFFSET(init); SAME_OFFSET(shutdown); SAME_OFFSET(is_available);
SAME_OFFSET(device_count); SAME_OFFSET(device_info); SAME_OFFSET(set_device);
SAME_OFFSET(mem_alloc); SAME_OFFSET(mem_free); SAME_OFFSET(memcpy_h2d); SAME_OFFSET(memcpy_d2h);
SAME_OFFSET(launch_l2_distance); SAME_OFFSET(launch_cosine); SAME_OFFSET(launch_kmeans_assign);
SAME_OFFSET(launch_kmeans_update); SAME_OFFSET(launch_quant_fp16); SAME_OFFSET(launch_quant_int8);
SAME_OFFSET(launch_quant_int4); SAME_OFFSET(launch_quant_fp8_e4m3); SAME_OFFSET(launch_quant_fp8_e5m2);
SAME_OFFSET(launch_quant_binary); SAME_OFFSET(launch_pq_encode);
_Static_assert(sizeof(ndb_hip_device_info) == sizeof(NDBGpuDeviceInfo), "ndb_hip_device_info");

void
ndbhip_register_backend(void)
{
	static ndb_gpu_backend b;	/* zero: every ML / LLM launcher NULL = "not provided" */

	memcpy(&b, ndb_hip_backend_get(), sizeof(ndb_hip_backend));
	ndb_hip_backend_streams(&b.stream_create, &b.stream_destroy, &b.stream_synchronize);
	ndb_gpu_register_backend(&b);	/* src/gpu/common/gpu_backend_registry.c:91-131 */
}

/* ---- GUCs ---- */
static int	guc_ivf_probes = 10, guc_ivf_k = 10, guc_hnsw_ef = 64, guc_hnsw_k = 10;
static bool guc_ref_compat = false;
static char *guc_device_service = NULL;

static void assign_ivf_probes(int v, void *extra) { (void) ndb_am_set_guc("neurondb.ivf_probes", v); }
static void assign_ivf_k(int v, void *extra) { (void) ndb_am_set_guc("neurondb.ivf_k", v); }
static void assign_hnsw_ef(int v, void *extra) { (void) ndb_am_set_guc("neurondb.hnsw_ef_search", v); }
static void assign_hnsw_k(int v, void *extra) { (void) ndb_am_set_guc("neurondb.hnsw_k", v); }
static void assign_ref_compat(bool v, void *extra) { (void) ndb_am_set_guc("neurondb.ref_compat", v ? 1 : 0); }
static void assign_device_service(const char *v, void *extra) { (void) ndb_am_use_service(v); }

/* called from _PG_init (src/worker/worker_init.c:90-107): defines GUCs only — no HIP call before fork */
void
ndbhip_define_gucs(void)
{
	/* neurondb.ivf_probes exists in the reference (src/util/neurondb_guc.c:187-198) but nothing reads it
	 * (quirk Q4): the assign hook is what makes it take effect */
	DefineCustomIntVariable("neurondb.ivf_probes", "lists probed by an ivf index scan", NULL, &guc_ivf_probes, 10, 1,
							NDBHIP_MAX_NPROBE, PGC_USERSET, 0, NULL, assign_ivf_probes, NULL);
	DefineCustomIntVariable("neurondb.ivf_k", "rows an ivf index scan returns (the reference pins 10: ivf_am.c:1422)",
							NULL, &guc_ivf_k, 10, 1, NDBHIP_MAX_K, PGC_USERSET, 0, NULL, assign_ivf_k, NULL);
	DefineCustomIntVariable("neurondb.hnsw_ef_search", "hnsw_am.c:923-936", NULL, &guc_hnsw_ef, 64, 0, NDBHIP_MAX_EF,
							PGC_USERSET, 0, NULL, assign_hnsw_ef, NULL);
	DefineCustomIntVariable("neurondb.hnsw_k", "hnsw_am.c:974", NULL, &guc_hnsw_k, 10, 0, NDBHIP_MAX_K, PGC_USERSET, 0,
							NULL, assign_hnsw_k, NULL);
	DefineCustomBoolVariable("neurondb.ref_compat", "keep the reference's quirks (k = 10, candidate cap k * 10, L2 for "
							 "every opclass)", NULL, &guc_ref_compat, false, PGC_USERSET, 0, NULL, assign_ref_compat, NULL);
	DefineCustomIntVariable("neurondb.generation_cells", "indexes the cluster-wide generation table can hold (a power of two; "
							"read when the first backend creates the table)", NULL, &guc_gen_cells, 65536, 1024, 1 << 24,
							PGC_POSTMASTER, 0, NULL, NULL, NULL);
	DefineCustomStringVariable("neurondb.device_service", "shared-memory name of the device-owner process "
							   "(include/ndb_service.h); empty = this backend drives the device itself", NULL,
							   &guc_device_service, "", PGC_SUSET, 0, NULL, assign_device_service, NULL);
}

/* ---- lazy, per-process initialisation: never from _PG_init (the library is loaded pre-fork) ---- */
bool
ndb_hip_ready(void)
{
	static int	state = 0;		/* 0 unknown, 1 ready, -1 unavailable */

	if (neurondb_compute_mode == 0)
		return false;
	if (guc_device_service && guc_device_service[0])
		return true;			/* the device-owner process holds the device: this backend never touches HIP */
	if (state == 0)
	{
		int			rc = ndbhip_init(neurondb_gpu_device);

		state = (rc == NDBHIP_OK) ? 1 : -1;
		if (rc != NDBHIP_OK)
			ereport(neurondb_compute_mode == 1 ? ERROR : WARNING,
					(errmsg("neurondb: HIP device unavailable: %s", ndbhip_last_error())));
	}
	return state == 1;
}

/* ---- mirror cache: one mirror per index and per version of its pages (rules and tests: ndb_mirror_cache.h) ---- */
#define NDB_MC_DESTROY_IVF(p) ndbhip_ivf_destroy((ndbhip_ivf *) (p))
#define NDB_MC_DESTROY_HNSW(p) ndbhip_hnsw_destroy((ndbhip_hnsw *) (p))
#include "ndb_mirror_cache.h"

static uint64 index_key(Relation index);

/* ambeginscan / amendscan: a scan holds the raw mirror pointer from its first rescan to its end */
void
ndb_hip_mirror_pin(const void *mirror)
{
	ndb_mc_pin(mirror);
}

void
ndb_hip_mirror_unpin(const void *mirror)
{
	ndb_mc_unpin(mirror);
}

/*
 * A pin that cannot leak.  amendscan is not called when a query is cancelled or an ereport(ERROR) unwinds through the
 * executor, and a rescan (LATERAL, nested loop) must not pin a second time: the scan takes ONE scoped pin when it first
 * takes the mirror (ivfrescan / hnswrescan, inside `if (!so->ndb)`), tied to the memory context the scan's opaque state
 * lives in.  The pin is released exactly once — by ndb_hip_mirror_unpin_scoped from amendscan or the fallback path, or
 * by the context's reset callback when the query's memory goes away on abort, whichever comes first.
 */
typedef struct ScopedPin
{
	MemoryContextCallback cb;
	const void *mirror;
	bool		released;
} ScopedPin;

static void
scoped_pin_release(void *arg)
{
	ScopedPin  *p = (ScopedPin *) arg;

	if (!p->released)
	{
		p->released = true;
		ndb_hip_mirror_unpin(p->mirror);
	}
}

void *
ndb_hip_mirror_pin_scoped(const void *mirror, MemoryContext scan_context)
{
	ScopedPin  *p;

	if (!mirror)
		return NULL;
	p = (ScopedPin *) MemoryContextAllocZero(scan_context, sizeof(ScopedPin));
	p->mirror = mirror;
	p->cb.func = scoped_pin_release;
	p->cb.arg = p;
	MemoryContextRegisterResetCallback(scan_context, &p->cb);	/* before the pin: nothing may throw between the two */
	ndb_hip_mirror_pin(mirror);
	return p;
}

void
ndb_hip_mirror_unpin_scoped(void *pin)
{
	if (pin)
		scoped_pin_release(pin);
}

static void
mirror_reset(void *arg)			/* MemoryContextCallback on CacheMemoryContext: ERROR unwinds by longjmp */
{
	(void) arg;
	ndb_mc_reset();
}

/* the relation's blocks, copied under SHARE locks: no pointer into a buffer page outlives its lock */
static uint8 *
copy_relation_pages(Relation index, BlockNumber *nblocks)
{
	BlockNumber n = RelationGetNumberOfBlocks(index);
	uint8	   *pages = (uint8 *) palloc((Size) n * BLCKSZ);

	for (BlockNumber b = 0; b < n; b++)
	{
		Buffer		buf = ReadBufferExtended(index, MAIN_FORKNUM, b, RBM_NORMAL, NULL);

		LockBuffer(buf, BUFFER_LOCK_SHARE);
		memcpy(pages + (Size) b * BLCKSZ, BufferGetPage(buf), BLCKSZ);
		UnlockReleaseBuffer(buf);
	}
	*nblocks = n;
	return pages;
}

/* NULL: the index's stale mirror is still pinned by scans of this backend and no retired slot is free — the caller does
 * not search a device mirror for this call (neurondb.compute_mode decides: CPU path or ERROR) */
static NdbMirrorEntry *
mirror_slot(Relation index, uint64 stamp)
{
	static bool registered = false;
	NdbMirrorEntry *e;
	int			full = 0;

	if (!registered)
	{
		MemoryContextCallback *cb = MemoryContextAlloc(CacheMemoryContext, sizeof(*cb));

		cb->func = mirror_reset;
		cb->arg = NULL;
		MemoryContextRegisterResetCallback(CacheMemoryContext, cb);
		registered = true;
	}
	e = ndb_mc_slot((uint32_t) RelationGetRelid(index), index_key(index), stamp, &full);
	if (!e && full)
		ereport(ERROR, (errmsg("neurondb: more than %d device-mirrored indexes in one backend", NDB_MAX_MIRRORS)));
	return e;
}

ndbhip_ivf *
ndb_hip_ivf_mirror(Relation index, uint64 stamp)
{
	NdbMirrorEntry *e = mirror_slot(index, stamp);

	if (!e)
		return NULL;			/* a stale mirror that cannot be dropped yet is never served: the caller applies neurondb.compute_mode */
	if (!e->ivf)
	{
		BlockNumber n;
		uint8	   *pages = copy_relation_pages(index, &n);
		int			rc = ndbhip_ivf_load_pages((ndbhip_ivf **) &e->ivf, pages, (uint32) n);

		pfree(pages);
		if (rc != NDBHIP_OK)
		{
			e->ivf = NULL;
			return NULL;		/* the caller applies neurondb.compute_mode */
		}
	}
	return (ndbhip_ivf *) e->ivf;
}

ndbhip_hnsw *
ndb_hip_hnsw_mirror(Relation index, uint64 stamp)
{
	NdbMirrorEntry *e = mirror_slot(index, stamp);

	if (!e)
		return NULL;
	if (!e->hnsw)
	{
		BlockNumber n;
		uint8	   *pages = copy_relation_pages(index, &n);
		int			rc = ndbhip_hnsw_load_pages((ndbhip_hnsw **) &e->hnsw, pages, (uint32) n);

		pfree(pages);
		if (rc != NDBHIP_OK)
		{
			e->hnsw = NULL;
			return NULL;
		}
	}
	return (ndbhip_hnsw *) e->hnsw;
}

bool
guc_device_service_set(void)
{
	return guc_device_service && guc_device_service[0];
}

/*
 * Version stamp of an index's pages: its generation in the cluster-wide counter table.  Nothing ON the pages can
 * serve: meta->insertedVectors goes up in ivfinsert (ivf_am.c:1122-1157) and down again in ivfbulkdelete
 * (:1346), so an UPDATE + VACUUM elsewhere brings an old value back, and the reference changes its pages with
 * MarkBufferDirty alone (no WAL record, ivf_am.c:1137-1156), so their LSNs never move.  The table lives in POSIX
 * shared memory named after the postmaster's port (one per cluster); a backend attaches on first use.  Keys are
 * (database OID << 32 | index relfilenumber): a REINDEX gives the index a new key and with it generation 1, and
 * every mirror of the old file is stale by key.
 */
static ndb_gen *gen_table = NULL;
static int	guc_gen_cells = 65536;		/* neurondb.generation_cells: indexes (relfilenodes) the counter table can hold */

static uint64
index_key(Relation index)
{
	return ((uint64) MyDatabaseId << 32) | (uint64) RelationGetSmgr(index)->smgr_rlocator.locator.relNumber;
}

static ndb_gen *
generations(void)
{
	if (!gen_table)
	{
		char		name[64];

		snprintf(name, sizeof(name), "/ndbhip_gen_%d", PostPortNumber);
		/* never an ERROR: this runs inside aminsert and ambulkdelete of backends that may not use the accelerator at
		 * all.  Without the table every stamp is 0 = "unknown": mirrors are rebuilt for every scan, nothing is stale. */
		if (ndb_gen_attach(name, guc_gen_cells, &gen_table) != NDBHIP_OK)
		{
			static bool warned = false;

			if (!warned)
				ereport(WARNING, (errmsg("neurondb: %s; device mirrors will be rebuilt for every scan", ndbhip_last_error())));
			warned = true;
			gen_table = NULL;
		}
	}
	return gen_table;
}

uint64
ivf_stamp(Relation index)
{
	/* (ndb_gen_get: 0 for a NULL table, and 0 for a key that has no cell in a table that is full) */
	return ndb_gen_get(generations(), index_key(index));
}

/* the scan of a backend without a mirror of its own (neurondb.device_service): which index, which generation */
ndb_index_scan *
ndb_hip_ivf_service_scan(Relation index, int nkeys, int norderbys)
{
	return ndb_ivfbeginscan_service(index_key(index), ivf_stamp(index), nkeys, norderbys);
}

/* aminsert: keep a cached mirror in step instead of rebuilding it (ivf_am.c:954-1157 appended the entry to
 * list `list_id`); without a cached mirror there is nothing to do — the next scan packs the pages */
void
ndb_hip_ivf_note_insert(Relation index, int list_id, const float *vec, ItemPointer heap_tid)
{
	/* every backend's mirror of this index (and the device-owner process's) is now one change behind */
	const uint64 gen = ndb_gen_bump(generations(), index_key(index));

	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
		if (ndb_mc_mirrors[i].relid == (uint32_t) RelationGetRelid(index) && ndb_mc_mirrors[i].ivf)
		{
			/* this backend's own mirror follows along — only if nobody else changed the index in between
			 * (gen is exactly one past the generation the mirror holds; a never-fresh entry, stamp 0, does not follow) and
			 * the append itself worked */
			if (gen != 0 && ndb_mc_mirrors[i].stamp != 0 && gen == ndb_mc_mirrors[i].stamp + 1 &&
				ndb_mc_mirrors[i].key == index_key(index) &&
				ndbhip_ivf_append((ndbhip_ivf *) ndb_mc_mirrors[i].ivf, list_id, vec, (const uint8_t *) heap_tid) == NDBHIP_OK)
				ndb_mc_mirrors[i].stamp = gen;
			else
				(void) ndb_mc_drop(&ndb_mc_mirrors[i]);	/* out of step: rebuild on the next scan (an open scan keeps its copy until it
															 * ends; a mirror that cannot be retired stays, never fresh) */
		}
}

/* ambulkdelete: line pointers were killed (ivf_am.c:1172-1357); the mirror of this backend is dropped and the
 * next scan packs the surviving entries (ndbhip_ivf_delete does the same in place for a caller that batches) */
void
ndb_hip_ivf_note_delete(Relation index)
{
	(void) ndb_gen_bump(generations(), index_key(index));	/* other backends and the owner: reload before the next scan */
	ndb_mc_invalidate((uint32_t) RelationGetRelid(index));
}

/* ---- the scan callbacks: IndexScanDesc <-> ndb_index_scan ---- */
static ndb_scan_key
key_from(ScanKey orderby, Oid vector_oid, Oid halfvec_oid, Oid sparsevec_oid)
{
	ndb_scan_key k;
	struct varlena *d = PG_DETOAST_DATUM(orderby->sk_argument);	/* amrescan may get a toasted datum */
	Oid			t = orderby->sk_subtype;

	k.sk_strategy = orderby->sk_strategy;
	k.sk_type = t == halfvec_oid ? NDBHIP_TYPE_HALFVEC : t == sparsevec_oid ? NDBHIP_TYPE_SPARSEVEC : NDBHIP_TYPE_VECTOR;
	k.sk_argument = d;
	k.sk_len = VARSIZE_ANY(d);
	return k;
}

/* amrescan of both AMs; `device_scan` = so->ndb (a ndb_index_scan *) kept in the AM's scan opaque */
void
ndbhip_rescan(ndb_index_scan *device_scan, bool hnsw, ScanKey orderbys, int norderbys, Oid vector_oid,
			  Oid halfvec_oid, Oid sparsevec_oid)
{
	ndb_scan_key k;
	int			rc;

	if (norderbys > 0)
		k = key_from(&orderbys[0], vector_oid, halfvec_oid, sparsevec_oid);
	rc = hnsw ? ndb_hnswrescan(device_scan, NULL, 0, norderbys > 0 ? &k : NULL, norderbys)
		: ndb_ivfrescan(device_scan, NULL, 0, norderbys > 0 ? &k : NULL, norderbys);
	if (rc != NDBHIP_OK)
		ereport(ERROR, (errmsg("neurondb: %s", ndbhip_last_error())));
}

/* amgettuple: true + xs_heaptid / xs_orderbyvals[0] set (ivf_am.c:2013-2020), or false at the end.
 * A device failure (negative return) is ERROR under compute_mode = gpu and `*fallback = true` otherwise:
 * the caller then runs its unchanged CPU block (reference convention, src/gpu/common/gpu_distance.c:50-51). */
bool
ndbhip_gettuple(ndb_index_scan *device_scan, bool hnsw, IndexScanDesc scan, ScanDirection dir, bool *fallback)
{
	int			rc = hnsw ? ndb_hnswgettuple(device_scan, (int) dir) : ndb_ivfgettuple(device_scan, (int) dir);

	*fallback = false;
	if (rc < 0)
	{
		if (neurondb_compute_mode == 1)
			ereport(ERROR, (errmsg("neurondb: %s", ndbhip_last_error())));
		*fallback = true;
		return false;
	}
	if (rc == 0)
		return false;
	memcpy(&scan->xs_heaptid, &device_scan->xs_heaptid, sizeof(ItemPointerData));
	if (scan->numberOfOrderBys > 0 && !hnsw)	/* hnswgettuple does not set it (Q13) */
	{
		scan->xs_orderbyvals[0] = Float4GetDatum(device_scan->xs_orderbyval);
		scan->xs_orderbynulls[0] = false;
	}
	scan->xs_recheckorderby = false;
	return true;
}
