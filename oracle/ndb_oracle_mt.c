/*
 * ndb_oracle_mt.c — TEST INFRASTRUCTURE (see ndb_oracle.h): a pthread driver around the oracle's single-query
 * functions, for the CPU baseline bench.py reports next to the device numbers.  It adds no arithmetic of its own:
 * every query goes through ndbo_ivf_search (= ivfgettuple's first call, src/index/ivf_am.c:1976-1999) or
 * ndbo_ivf_assign exactly as one PostgreSQL backend would run it; the driver only spreads the queries over one
 * thread per host core (one backend per core: SURVEY 8d "CPU baseline timing" (2)).
 *
 * Memory placement matters more than the thread count on a many-socket host: a Python caller allocates and fills
 * the row array from ONE thread, so first-touch puts every page on that thread's NUMA node and 256 cores then read
 * 3 GB per query through one node's memory controllers (the round-2 harness measured 6x one thread on 256 cores).
 * ndbo_mt_image_clone copies the image into memory whose pages are first touched by the worker threads
 * themselves, 2 MiB at a time round-robin, so the rows are spread over the nodes the workers run on.
 */
#define _GNU_SOURCE
#include "ndb_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct mt_copy_job
{
	char	   *dst;
	const char *src;
	size_t		bytes;
	int			tid,
				nthreads;
}			mt_copy_job;

#define MT_STRIPE ((size_t) 2 << 20)

static void *
mt_copy_worker(void *arg)
{
	mt_copy_job *j = (mt_copy_job *) arg;
	size_t		off;

	for (off = (size_t) j->tid * MT_STRIPE; off < j->bytes; off += (size_t) j->nthreads * MT_STRIPE)
	{
		size_t		n = j->bytes - off < MT_STRIPE ? j->bytes - off : MT_STRIPE;

		memcpy(j->dst + off, j->src + off, n);	/* first touch of these pages: they land on this thread's node */
	}
	return NULL;
}

/* a copy of `bytes` bytes whose pages were first touched stripe by stripe by `nthreads` threads; free() it */
void *
ndbo_mt_spread_copy(const void *src, size_t bytes, int nthreads)
{
	char	   *dst = NULL;
	pthread_t  *th;
	mt_copy_job *jobs;
	int			t;

	if (bytes == 0)
		return NULL;
	if (posix_memalign((void **) &dst, 4096, bytes) != 0)
		return NULL;
	if (nthreads < 1)
		nthreads = 1;
	th = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) nthreads);
	jobs = (mt_copy_job *) malloc(sizeof(mt_copy_job) * (size_t) nthreads);
	for (t = 0; t < nthreads; t++)
	{
		jobs[t].dst = dst;
		jobs[t].src = (const char *) src;
		jobs[t].bytes = bytes;
		jobs[t].tid = t;
		jobs[t].nthreads = nthreads;
		pthread_create(&th[t], NULL, mt_copy_worker, &jobs[t]);
	}
	for (t = 0; t < nthreads; t++)
		pthread_join(th[t], NULL);
	free(th);
	free(jobs);
	return dst;
}

typedef struct mt_search_job
{
	const ndbo_ivf *ix;
	const float *queries;
	int			nq,
				strategy,
				nprobe,
				k;
	int64_t		max_candidates;
	ndbo_tid   *out_tids;
	float	   *out_dist;
	int		   *out_count;
	atomic_int *next;
	int64_t		scored;
}			mt_search_job;

static void *
mt_search_worker(void *arg)
{
	mt_search_job *j = (mt_search_job *) arg;

	for (;;)
	{
		const int	q = atomic_fetch_add(j->next, 1);
		int64_t		ns = 0;

		if (q >= j->nq)
			break;
		j->out_count[q] = ndbo_ivf_search(j->ix, j->queries + (size_t) q * j->ix->dim, j->strategy, j->nprobe, j->k,
										  j->max_candidates, j->out_tids + (size_t) q * j->k,
										  j->out_dist + (size_t) q * j->k, &ns);
		j->scored += ns;
	}
	return NULL;
}

/*
 * nq queries through ndbo_ivf_search on `nthreads` threads (queries handed out one at a time).  out_tids [nq][k],
 * out_dist [nq][k], out_count [nq].  Returns the wall time of the parallel section in seconds (CLOCK_MONOTONIC);
 * *n_scored (nullable) = distance evaluations in all.
 */
double
ndbo_mt_ivf_search_batch(const ndbo_ivf *ix, const float *queries, int nq, int strategy, int nprobe, int k,
						 int64_t max_candidates, int nthreads, ndbo_tid *out_tids, float *out_dist, int *out_count,
						 int64_t *n_scored)
{
	pthread_t  *th;
	mt_search_job *jobs;
	atomic_int	next;
	struct timespec t0,
				t1;
	int			t;
	int64_t		total = 0;

	if (nthreads < 1)
		nthreads = 1;
	if (nthreads > nq)
		nthreads = nq > 0 ? nq : 1;
	atomic_init(&next, 0);
	th = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) nthreads);
	jobs = (mt_search_job *) malloc(sizeof(mt_search_job) * (size_t) nthreads);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (t = 0; t < nthreads; t++)
	{
		jobs[t].ix = ix;
		jobs[t].queries = queries;
		jobs[t].nq = nq;
		jobs[t].strategy = strategy;
		jobs[t].nprobe = nprobe;
		jobs[t].k = k;
		jobs[t].max_candidates = max_candidates;
		jobs[t].out_tids = out_tids;
		jobs[t].out_dist = out_dist;
		jobs[t].out_count = out_count;
		jobs[t].next = &next;
		jobs[t].scored = 0;
		pthread_create(&th[t], NULL, mt_search_worker, &jobs[t]);
	}
	for (t = 0; t < nthreads; t++)
	{
		pthread_join(th[t], NULL);
		total += jobs[t].scored;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(th);
	free(jobs);
	if (n_scored)
		*n_scored = total;
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

typedef struct mt_assign_job
{
	const float *rows;
	const float *cents;
	int64_t		n;
	int			dim,
				k;
	int		   *out;
	atomic_llong *next;
}			mt_assign_job;

static void *
mt_assign_worker(void *arg)
{
	mt_assign_job *j = (mt_assign_job *) arg;

	for (;;)
	{
		const long long r0 = atomic_fetch_add(j->next, 256);
		long long	r;

		if (r0 >= j->n)
			break;
		for (r = r0; r < r0 + 256 && r < j->n; r++)
			j->out[r] = ndbo_ivf_assign(j->cents, NULL, j->k, j->k, j->dim, j->rows + (size_t) r * j->dim, NULL);
	}
	return NULL;
}

/* the insert-time assignment (ndbo_ivf_assign = src/index/ivf_am.c:905-935) of n rows on nthreads threads;
 * returns the wall time in seconds */
double
ndbo_mt_ivf_assign_batch(const float *rows, int64_t n, int dim, const float *cents, int k, int nthreads, int *out)
{
	pthread_t  *th;
	mt_assign_job *jobs;
	atomic_llong next;
	struct timespec t0,
				t1;
	int			t;

	if (nthreads < 1)
		nthreads = 1;
	atomic_init(&next, 0);
	th = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) nthreads);
	jobs = (mt_assign_job *) malloc(sizeof(mt_assign_job) * (size_t) nthreads);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (t = 0; t < nthreads; t++)
	{
		jobs[t].rows = rows;
		jobs[t].cents = cents;
		jobs[t].n = n;
		jobs[t].dim = dim;
		jobs[t].k = k;
		jobs[t].out = out;
		jobs[t].next = &next;
		pthread_create(&th[t], NULL, mt_assign_worker, &jobs[t]);
	}
	for (t = 0; t < nthreads; t++)
		pthread_join(th[t], NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(th);
	free(jobs);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}
