/*
 * ndb_mirror_cache.h — the per-backend cache of device mirrors behind pgext/ndbhip_glue.c, free of PostgreSQL types so that
 * its rules can be compiled and tested without a server (tests/test_mirror_cache.py builds tests/mirror_cache_harness.c
 * around this file with counting destroy hooks).
 *
 * One mirror per index (relation OID) and per version of its pages: an entry is FRESH only for the same file
 * (key = database << 32 | relfilenumber) at the same generation (stamp; 0 = unknown, never fresh).  A scan of this backend
 * PINS the raw mirror pointer from its first rescan to its end; a mirror that goes stale while pinned is RETIRED (kept
 * alive, out of the lookup's sight) and destroyed by its last unpin.  When every retired slot is held too, the stale mirror
 * stays in its entry, marked never-fresh, and the lookup answers NULL — the caller applies neurondb.compute_mode (CPU path) —
 * until the pins are gone: a stale mirror is never served (the round-5 code re-stamped the entry after a failed
 * drop and served old pages as fresh).
 *
 * The includer defines NDB_MC_DESTROY_IVF(p) / NDB_MC_DESTROY_HNSW(p) before including this file.
 */
#ifndef NDB_MIRROR_CACHE_H
#define NDB_MIRROR_CACHE_H

#include <stdbool.h>
#include <stdint.h>
#include <string.h>

#ifndef NDB_MAX_MIRRORS
#define NDB_MAX_MIRRORS 32
#endif

typedef struct NdbMirrorEntry
{
	uint32_t	relid;			/* 0 (InvalidOid) = free */
	uint64_t	key;			/* (database, relfilenumber) of the file the pages were read from */
	uint64_t	stamp;			/* that key's generation when the pages were read; 0 = unknown, never fresh */
	void	   *ivf;
	void	   *hnsw;
	int			pins;			/* open scans of this backend that hold `ivf` / `hnsw` */
} NdbMirrorEntry;

typedef struct NdbRetired
{
	void	   *ivf;
	void	   *hnsw;
	int			pins;
} NdbRetired;

static NdbMirrorEntry ndb_mc_mirrors[NDB_MAX_MIRRORS];
static NdbRetired ndb_mc_retired[NDB_MAX_MIRRORS];

/* true: the entry holds no mirror any more (destroyed, or retired until its scans end); false: every retired slot is held
 * by an open scan too — the mirror stays in the entry, which is marked never-fresh.  Never an ERROR from here (this runs
 * inside aminsert, ambulkdelete and every scan's lookup). */
static bool
ndb_mc_drop(NdbMirrorEntry *e)
{
	if (e->pins > 0 && (e->ivf || e->hnsw))
	{
		for (int i = 0; i < NDB_MAX_MIRRORS; i++)
			if (!ndb_mc_retired[i].ivf && !ndb_mc_retired[i].hnsw)
			{
				ndb_mc_retired[i].ivf = e->ivf;
				ndb_mc_retired[i].hnsw = e->hnsw;
				ndb_mc_retired[i].pins = e->pins;
				e->ivf = NULL;
				e->hnsw = NULL;
				e->pins = 0;
				return true;
			}
		e->stamp = 0;			/* matches no lookup; dropped for good by a later call once its scans have ended */
		return false;
	}
	if (e->ivf) NDB_MC_DESTROY_IVF(e->ivf);
	if (e->hnsw) NDB_MC_DESTROY_HNSW(e->hnsw);
	e->ivf = NULL;
	e->hnsw = NULL;
	e->pins = 0;
	return true;
}

static void
ndb_mc_pin(const void *mirror)
{
	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
		if (mirror && ((const void *) ndb_mc_mirrors[i].ivf == mirror || (const void *) ndb_mc_mirrors[i].hnsw == mirror))
		{
			ndb_mc_mirrors[i].pins++;
			return;
		}
}

static void
ndb_mc_unpin(const void *mirror)
{
	if (!mirror)
		return;
	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
	{
		if ((const void *) ndb_mc_mirrors[i].ivf == mirror || (const void *) ndb_mc_mirrors[i].hnsw == mirror)
		{
			if (ndb_mc_mirrors[i].pins > 0)
				ndb_mc_mirrors[i].pins--;
			return;
		}
		if ((const void *) ndb_mc_retired[i].ivf == mirror || (const void *) ndb_mc_retired[i].hnsw == mirror)
		{
			if (--ndb_mc_retired[i].pins <= 0)
			{
				if (ndb_mc_retired[i].ivf) NDB_MC_DESTROY_IVF(ndb_mc_retired[i].ivf);
				if (ndb_mc_retired[i].hnsw) NDB_MC_DESTROY_HNSW(ndb_mc_retired[i].hnsw);
				memset(&ndb_mc_retired[i], 0, sizeof(ndb_mc_retired[i]));
			}
			return;
		}
	}
}

/*
 * The entry of index `relid` for the file `key` at generation `stamp`: its mirror pointers are NULL when it has to be
 * (re)loaded.  NULL = no entry can be handed out: *full != 0 — more than NDB_MAX_MIRRORS indexes in one backend —, or
 * (*full == 0) the index's stale mirror could not be dropped yet (all retired slots pinned): the caller must NOT search a
 * device mirror for this call.
 * Fresh only for the SAME file at the SAME generation: a new relfilenode (TRUNCATE, VACUUM FULL, CLUSTER, REINDEX) starts
 * its own counter at 1 again, so the stamp alone would keep the old file's mirror — and its heap TIDs — alive.
 */
static NdbMirrorEntry *
ndb_mc_slot(uint32_t relid, uint64_t key, uint64_t stamp, int *full)
{
	NdbMirrorEntry *free_slot = NULL;

	*full = 0;
	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
	{
		NdbMirrorEntry *e = &ndb_mc_mirrors[i];

		if (e->relid == relid && relid != 0)
		{
			if (e->key != key || e->stamp != stamp || stamp == 0)
			{
				if (!ndb_mc_drop(e))
					return NULL;	/* (key and stamp stay as they are: stamp 0, the mirror is not fresh for anybody) */
				e->key = key;
				e->stamp = stamp;
			}
			return e;
		}
		if (!free_slot && e->relid == 0)
			free_slot = e;
	}
	if (!free_slot)
	{
		*full = 1;
		return NULL;
	}
	free_slot->relid = relid;
	free_slot->key = key;
	free_slot->stamp = stamp;
	return free_slot;
}

/* the index's pages changed in a way the mirror cannot follow (ambulkdelete, a failed append): rebuild on the next scan */
static void
ndb_mc_invalidate(uint32_t relid)
{
	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
		if (ndb_mc_mirrors[i].relid == relid && relid != 0)
		{
			if (ndb_mc_drop(&ndb_mc_mirrors[i]))
				memset(&ndb_mc_mirrors[i], 0, sizeof(ndb_mc_mirrors[i]));
			/* else: the entry keeps the pinned mirror, never fresh (stamp 0), until a later lookup can drop it */
		}
}

/* backend exit / cache reset: everything goes */
static void
ndb_mc_reset(void)
{
	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
	{
		if (ndb_mc_mirrors[i].ivf) NDB_MC_DESTROY_IVF(ndb_mc_mirrors[i].ivf);
		if (ndb_mc_mirrors[i].hnsw) NDB_MC_DESTROY_HNSW(ndb_mc_mirrors[i].hnsw);
		memset(&ndb_mc_mirrors[i], 0, sizeof(ndb_mc_mirrors[i]));
		if (ndb_mc_retired[i].ivf) NDB_MC_DESTROY_IVF(ndb_mc_retired[i].ivf);
		if (ndb_mc_retired[i].hnsw) NDB_MC_DESTROY_HNSW(ndb_mc_retired[i].hnsw);
		memset(&ndb_mc_retired[i], 0, sizeof(ndb_mc_retired[i]));
	}
}

#endif							/* NDB_MIRROR_CACHE_H */
