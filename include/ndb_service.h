/*
 * ndb_service.h — one device-owner process serving the index scans of many PostgreSQL backends.
 *
 * Why: `ORDER BY v <-> $q LIMIT k` reaches the access method as ONE query per amrescan
 * (src/index/ivf_am.c:1439-1545; the work happens in the first amgettuple, :1911-2027), and a backend is a
 * single-threaded process.  Called from each backend on its own, the device path costs ~0.2 ms per query
 * (~5 k queries/s per backend) and every backend would upload its own copy of the index (3 GB at 1M x 768).
 * The batched kernels need tens to hundreds of queries per launch to run at their rate.  So the backends do not touch the
 * device: they hand their query to a shared-memory ring, ONE process owns the device and the mirror, coalesces
 * whatever is waiting into one launch of ndbhip_ivf_search and writes every backend's rows back.  This is the
 * reference's own lazy, per-process GPU initialisation contract (src/gpu/common/gpu_core.c:240-310:
 * ndb_gpu_init_if_needed — nothing touches the device before a backend needs it, failure falls back to the CPU
 * unless compute_mode forbids it) applied to the one process that needs the device at all; a backend whose
 * connect or wait fails falls back to its CPU scan exactly like a failed ndb_gpu_init_if_needed does.
 *
 * Shared state: a POSIX shm segment `name`: header + `nslots` request slots (query [dim] floats in, <= max_k
 * rows out).  Slot life cycle (one 32-bit state word per slot, futex-woken):
 *   FREE -> CLAIMED (backend, CAS) -> READY (backend filled the query) -> RUNNING (owner took it into a batch)
 *        -> DONE (owner wrote rows + status) -> FREE (backend read them).
 * Results of a slot are exactly what ndbhip_ivf_search returns for that query alone (ids, ranks, float4 bits:
 * batches never change results — tests/test_service.py checks it against the oracle).
 *
 * All functions return 0 or a negative NDBHIP_ERR_* code (message: ndbhip_last_error()).  PG-free, plain C ABI.
 */
#ifndef NDB_SERVICE_H
#define NDB_SERVICE_H

#include <stddef.h>
#include <stdint.h>
#include "ndbhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ndb_service ndb_service;	/* the owner's end */
typedef struct ndb_client ndb_client;		/* a backend's end */

typedef struct ndb_service_stats
{
	uint64_t	batches;			/* launches */
	uint64_t	queries;			/* queries served */
	uint64_t	max_batch;			/* largest batch */
	double		busy_s;				/* time inside the executor */
}			ndb_service_stats;

/* ---- owner ---- */
/* create the segment (name "/..."; an old one of that name is replaced) */
int			ndb_service_create(const char *name, int dim, int max_k, int nslots, ndb_service **out);
int			ndb_service_destroy(ndb_service *s);
/* Gather up to max_batch READY requests that share (strategy, nprobe, k, max_candidates) with the oldest one:
 * waits up to wait_us for the first, then lingers up to linger_us while more keep arriving.  Returns the number
 * gathered (0 = nothing within wait_us, or stopped); fills slot_ids[], queries [n][dim] and the four parameters.
 * queries == NULL: the queries are not copied — an executor that can read the ring itself takes them from
 * ndb_service_query_offset(s, slot_ids[i]) (ndb_service_serve_ivf does: the device gathers them). */
int			ndb_service_poll(ndb_service *s, int max_batch, int wait_us, int linger_us, int *slot_ids, float *queries,
							 int *strategy, int *nprobe, int *k, int64_t *max_candidates);
/* where slot `slot_id`'s query lies in the segment (bytes from its start); ndb_service_segment: the owner's mapping */
int64_t		ndb_service_query_offset(const ndb_service *s, int slot_id);
int			ndb_service_segment(const ndb_service *s, void **base, size_t *bytes);
/* rows of a gathered batch (tids6 [n][k][6], dist [n][k], count [n]); status != 0 is handed to the backends
 * as the error of their wait */
int			ndb_service_complete(ndb_service *s, int n, const int *slot_ids, const uint8_t *tids6, const float *dist,
								 const int *count, int k, int status);
/* The device executor: poll -> ndbhip_ivf_search(ix, batch) -> complete, until ndb_service_stop() or
 * max_batches launches (0 = unbounded).  Needs ndbhip_init() in this process. */
int			ndb_service_serve_ivf(ndb_service *s, ndbhip_ivf *ix, int max_batch, int linger_us, int64_t max_batches,
								  ndb_service_stats *stats);
/* Which index the owner's mirror is (key: its relfilenode / OID; 0 = unkeyed, for a single-index deployment and
 * the tests) and at which generation (ndb_gen_get at the time the pages were read), plus the index's own nprobe
 * (reloptions / meta page, ivf_am.c:1487-1513) for backends in neurondb.ref_compat.  A request whose key or
 * generation differs is refused with NDBHIP_ERR_NODEVICE — the backend runs its CPU scan — and a newer generation
 * makes ndb_service_serve_ivf return so that the owner can reload: while (!stopped) { load; publish; serve; }. */
int			ndb_service_publish(ndb_service *s, uint64_t index_key, uint64_t index_version, int meta_nprobe);
int			ndb_service_reload_wanted(const ndb_service *s, uint64_t *version);	/* 1: a backend has seen *version > the published one */
/* slots of backends that died mid-request go back to FREE (the poll loop does this once a second by itself);
 * returns how many */
int			ndb_service_reclaim(ndb_service *s);
int			ndb_service_stop(ndb_service *s);		/* callable from another thread / a signal handler of the owner */
int			ndb_service_stopped(const ndb_service *s);

/* ---- backend (what ivfrescan + the first ivfgettuple call instead of ndbhip_ivf_search) ---- */
int			ndb_client_connect(const char *name, ndb_client **out);
int			ndb_client_disconnect(ndb_client *c);
int			ndb_client_dim(const ndb_client *c);
int			ndb_client_stop_service(ndb_client *c);	/* ask the owner's loop to end (tests, shutdown) */
/* one query: submit returns a ticket; wait blocks (timeout_ms < 0: forever) and frees the slot.  A backend keeps
 * one ticket in flight; a caller with many independent queries (a batched SQL function) may keep several. */
int			ndb_client_submit(ndb_client *c, const float *query, int strategy, int nprobe, int k,
							  int64_t max_candidates, int *ticket);
int			ndb_client_wait(ndb_client *c, int ticket, uint8_t *tids6, float *dist, int *count, int timeout_ms);
int			ndb_client_search(ndb_client *c, const float *query, int strategy, int nprobe, int k, int64_t max_candidates,
							  uint8_t *tids6, float *dist, int *count, int timeout_ms);
/* the same for a scan on index `index_key` at generation `index_version` (the plain forms above are key 0,
 * generation 0 and are only answered by an owner that published exactly that) */
int			ndb_client_submit_index(ndb_client *c, uint64_t index_key, uint64_t index_version, const float *query,
									int strategy, int nprobe, int k, int64_t max_candidates, int *ticket);
int			ndb_client_search_index(ndb_client *c, uint64_t index_key, uint64_t index_version, const float *query,
									int strategy, int nprobe, int k, int64_t max_candidates, uint8_t *tids6, float *dist,
									int *count, int timeout_ms);
int			ndb_client_index(const ndb_client *c, uint64_t *index_key, uint64_t *index_version);	/* what the owner serves */
int			ndb_client_meta_nprobe(const ndb_client *c);	/* the served index's own nprobe (0: not published) */

/* ---- index generations: the version stamp of every device mirror (backends' own and the owner's) ----
 * A shared table (POSIX shm `name`, created by whoever attaches first; ncells a power of two) of one counter per
 * index.  Every aminsert / ambulkdelete bumps its index's counter after changing the pages; a scan compares the
 * counter with the generation its mirror was loaded at.  Counters only grow, so a stamp never comes back (the
 * reference's meta->insertedVectors does: +1 by ivfinsert, -n by ivfbulkdelete, ivf_am.c:1346). */
typedef struct ndb_gen ndb_gen;
int			ndb_gen_attach(const char *name, int ncells, ndb_gen **out);
int			ndb_gen_detach(ndb_gen *g, const char *unlink_name);	/* unlink_name != NULL also removes the segment */
uint64_t	ndb_gen_get(ndb_gen *g, uint64_t key);		/* >= 1; 0 = unknown (key 0, no table, or no cell for the key in a FULL
													 * table): the caller reloads for every scan and never treats 0 as a match */
uint64_t	ndb_gen_bump(ndb_gen *g, uint64_t key);	/* the new generation; 0 = table full */

#ifdef __cplusplus
}
#endif
#endif							/* NDB_SERVICE_H */
