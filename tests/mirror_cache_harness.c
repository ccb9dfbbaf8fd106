/* tests/mirror_cache_harness.c — pgext/ndb_mirror_cache.h (the backend's mirror cache, PostgreSQL-free) driven the way
 * ndbhip_glue.c drives it, with counting destroy hooks.  Built and run by tests/test_mirror_cache.py; exits 0 when every
 * expectation holds.  ADVICE r5: with all retired slots held, a stale mirror must never be handed out as fresh. */
#include <stdio.h>
#include <stdlib.h>

static int	destroyed_ivf = 0, destroyed_hnsw = 0;
static char alive[4096];		/* alive[id]: mirror `id` exists */

#define NDB_MC_DESTROY_IVF(p) do { destroyed_ivf++; alive[(size_t) (p)] = 0; } while (0)
#define NDB_MC_DESTROY_HNSW(p) do { destroyed_hnsw++; alive[(size_t) (p)] = 0; } while (0)
#include "../pgext/ndb_mirror_cache.h"

static size_t next_id = 1;

#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "mirror_cache_harness: line %d: %s\n", __LINE__, #c); exit(1); } } while (0)

/* ndb_hip_ivf_mirror(): the mirror of (relid, key) at `stamp`, loading one when the entry has none; NULL = CPU path */
static void *
lookup(uint32_t relid, uint64_t key, uint64_t stamp)
{
	int			full = 0;
	NdbMirrorEntry *e = ndb_mc_slot(relid, key, stamp, &full);

	if (!e)
		return NULL;
	if (!e->ivf)
	{
		e->ivf = (void *) next_id;		/* "ndbhip_ivf_load_pages" of the pages as they are NOW: always current */
		alive[next_id++] = 1;
	}
	return e->ivf;
}

int
main(void)
{
	void	   *m[NDB_MAX_MIRRORS + 2];
	void	   *cur,
			   *again;

	/* 1. fresh lookups return the same mirror; a new generation drops the old one (no pins: destroyed at once) */
	cur = lookup(7, 100, 1);
	EXPECT(cur && lookup(7, 100, 1) == cur && destroyed_ivf == 0);
	again = lookup(7, 100, 2);
	EXPECT(again && again != cur && destroyed_ivf == 1 && !alive[(size_t) cur] && alive[(size_t) again]);
	/* a new relfilenode at the same generation number is another file: not fresh */
	cur = lookup(7, 101, 2);
	EXPECT(cur != again && destroyed_ivf == 2);
	/* generation 0 = unknown: never fresh */
	again = lookup(7, 101, 0);
	EXPECT(again != cur && destroyed_ivf == 3);
	cur = lookup(7, 101, 0);
	EXPECT(cur != again && destroyed_ivf == 4);

	/* 2. a pinned mirror that goes stale is retired, alive until its last unpin */
	cur = lookup(8, 200, 1);
	ndb_mc_pin(cur);
	ndb_mc_pin(cur);
	again = lookup(8, 200, 2);
	EXPECT(again != cur && alive[(size_t) cur] && destroyed_ivf == 4);
	ndb_mc_unpin(cur);
	EXPECT(alive[(size_t) cur]);
	ndb_mc_unpin(cur);
	EXPECT(!alive[(size_t) cur] && destroyed_ivf == 5);

	/* 3. fill EVERY retired slot: index 9 .. 9 + N - 1, each pinned once and then made stale */
	for (int i = 0; i < NDB_MAX_MIRRORS - 2; i++)		/* (entries 7 and 8 are in use) */
	{
		m[i] = lookup(100 + (uint32_t) i, 1000 + (uint64_t) i, 1);
		EXPECT(m[i]);
		ndb_mc_pin(m[i]);
	}
	for (int i = 0; i < NDB_MAX_MIRRORS - 2; i++)
		EXPECT(lookup(100 + (uint32_t) i, 1000 + (uint64_t) i, 2) != m[i] && alive[(size_t) m[i]]);	/* retired: N - 2 slots taken */
	/* two more through index 7 and 8 */
	cur = lookup(7, 101, 5);
	ndb_mc_pin(cur);
	EXPECT(lookup(7, 101, 6) != cur && alive[(size_t) cur]);
	m[NDB_MAX_MIRRORS - 2] = cur;
	cur = lookup(8, 200, 5);
	ndb_mc_pin(cur);
	EXPECT(lookup(8, 200, 6) != cur && alive[(size_t) cur]);
	m[NDB_MAX_MIRRORS - 1] = cur;
	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
		EXPECT(ndb_mc_retired[i].ivf != NULL);			/* the retired table is full */

	/* 4. THE case: index 100's current mirror (generation 2) is pinned by a scan; the index changes (generation 3) */
	cur = lookup(100, 1000, 2);
	EXPECT(cur);
	ndb_mc_pin(cur);
	{
		const int	d0 = destroyed_ivf;

		again = lookup(100, 1000, 3);
		EXPECT(again == NULL);			/* cannot retire, cannot destroy: NOT served — the caller takes the CPU path */
		EXPECT(lookup(100, 1000, 3) == NULL && lookup(100, 1000, 4) == NULL);	/* ... however often it asks */
		EXPECT(alive[(size_t) cur] && destroyed_ivf == d0);			/* the scan's copy is untouched */
		/* ambulkdelete / aminsert of the same index meanwhile: nothing is lost, nothing becomes fresh */
		ndb_mc_invalidate(100);
		EXPECT(lookup(100, 1000, 4) == NULL && alive[(size_t) cur]);
		/* the scan ends: the next lookup drops the stale mirror and loads the pages as they are now */
		ndb_mc_unpin(cur);
		again = lookup(100, 1000, 4);
		EXPECT(again && again != cur && !alive[(size_t) cur] && destroyed_ivf == d0 + 1);
		EXPECT(lookup(100, 1000, 4) == again);
	}

	/* 5. retired mirrors go with their last unpin; the table empties */
	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
	{
		ndb_mc_unpin(m[i]);
		EXPECT(!alive[(size_t) m[i]]);
	}
	for (int i = 0; i < NDB_MAX_MIRRORS; i++)
		EXPECT(ndb_mc_retired[i].ivf == NULL && ndb_mc_retired[i].hnsw == NULL);

	/* 6. the table of entries is bounded: index number N + 1 is refused with *full */
	{
		int			full = 0;

		for (uint32_t r = 500; r < 500 + NDB_MAX_MIRRORS; r++)
			(void) ndb_mc_slot(r, r, 1, &full);
		EXPECT(ndb_mc_slot(9999, 1, 1, &full) == NULL && full == 1);
	}
	ndb_mc_reset();
	for (size_t i = 1; i < next_id; i++)
		EXPECT(!alive[i]);				/* nothing leaks */
	printf("mirror_cache_harness: OK (%d mirrors destroyed)\n", destroyed_ivf + destroyed_hnsw);
	return 0;
}
