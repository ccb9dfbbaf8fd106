/*
 * The device-owner ring (include/ndb_service.h) under stress with NO device: one owner thread that answers every
 * request with rows computed from the query itself, THREADS backend threads that keep INFLIGHT single-query requests
 * in flight each and check that every answer is the one for THEIR query (routing), a backend that withdraws by timing
 * out on purpose, and a reclaim pass running alongside.  All in one process, so that a host-side sanitizer build of the
 * library (tools/san_build.sh: -fsanitize=address,undefined or thread) sees both sides of every slot word, hint byte and
 * futex.  Exit code 0 = every answer routed correctly.
 *
 * usage: service_stress [THREADS [INFLIGHT [QUERIES_PER_THREAD]]]
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "ndb_service.h"
#include "ndbhip.h"

#define DIM 32
#define MAXK 8

static const char *NAME = "/ndb_service_stress";
static ndb_service *svc;
static volatile int owner_rc = 0;

/* the "executor": neighbour j of a query is TID {q[0] as int, j}, distance q[1] + j */
static void
answer(const float *q, int k, uint8_t *tids6, float *dist, int *count)
{
	const uint32_t id = (uint32_t) q[0];

	for (int j = 0; j < k; j++)
	{
		uint8_t    *t = tids6 + 6 * j;

		t[0] = (uint8_t) (id >> 16);
		t[1] = (uint8_t) (id >> 24);
		t[2] = (uint8_t) id;
		t[3] = (uint8_t) (id >> 8);
		t[4] = (uint8_t) (j + 1);
		t[5] = 0;
		dist[j] = q[1] + (float) j;
	}
	*count = k;
}

static void *
owner(void *arg)
{
	(void) arg;
	enum { MAXB = 64 };
	int			ids[MAXB], cnt[MAXB];
	float	   *q = (float *) malloc(sizeof(float) * MAXB * DIM);
	float	   *dist = (float *) malloc(sizeof(float) * MAXB * MAXK);
	uint8_t    *tids = (uint8_t *) malloc((size_t) MAXB * MAXK * 6);
	long		rounds = 0;

	while (!ndb_service_stopped(svc))
	{
		int			strategy, nprobe, k;
		int64_t		cap;
		const int	n = ndb_service_poll(svc, MAXB, 2000, 20, ids, q, &strategy, &nprobe, &k, &cap);

		if (n < 0)
		{
			owner_rc = n;
			break;
		}
		if ((++rounds & 63) == 0)
			(void) ndb_service_reclaim(svc);
		if (n == 0)
			continue;
		for (int i = 0; i < n; i++)
			answer(q + (size_t) i * DIM, k, tids + (size_t) i * k * 6, dist + (size_t) i * k, &cnt[i]);
		if (ndb_service_complete(svc, n, ids, tids, dist, cnt, k, 0))
			owner_rc = -1;
	}
	free(q);
	free(dist);
	free(tids);
	return NULL;
}

typedef struct
{
	int			rank, inflight, nq, bad;
} backend_t;

static void *
backend(void *arg)
{
	backend_t  *b = (backend_t *) arg;
	ndb_client *c = NULL;
	int		   *ticket = (int *) malloc(sizeof(int) * (size_t) b->inflight);
	float	   *qs = (float *) malloc(sizeof(float) * (size_t) b->inflight * DIM);
	uint8_t		t1[MAXK * 6], e1[MAXK * 6];
	float		d1[MAXK], ed[MAXK];
	int			sent = 0, done = 0, head = 0, held = 0;

	if (ndb_client_connect(NAME, &c))
	{
		b->bad = 1000000;
		return NULL;
	}
	while (done < b->nq)
	{
		while (held < b->inflight && sent < b->nq)
		{
			const int	s = (head + held) % b->inflight;
			float	   *q = qs + (size_t) s * DIM;

			memset(q, 0, sizeof(float) * DIM);
			q[0] = (float) (b->rank * 100000 + sent);
			q[1] = (float) (sent % 97);
			if (ndb_client_submit(c, q, 1, 4, 1 + sent % MAXK, 0, &ticket[s]))
			{
				if (b->bad++ < 3)
					fprintf(stderr, "backend %d: submit %d failed: %s\n", b->rank, sent, ndbhip_last_error());
				break;
			}
			held++;
			sent++;
		}
		if (held == 0)
			break;
		{
			const float *q = qs + (size_t) head * DIM;
			const int	k = 1 + (done % MAXK);
			int			cnt = -1, ecnt = 0;
			/* every 53rd wait gives up at once (a cancelled query): the slot must come back, the next answers must
			 * still be this backend's own */
			const int	rc = ndb_client_wait(c, ticket[head], t1, d1, &cnt, (done % 53) == 52 ? 0 : 5000);

			if (rc == 0)
			{
				answer(q, k, e1, ed, &ecnt);
				if (cnt != ecnt || memcmp(t1, e1, (size_t) k * 6) || memcmp(d1, ed, sizeof(float) * (size_t) k))
				{
					if (b->bad++ < 3)
						fprintf(stderr, "backend %d: answer %d is not its own (count %d, expected %d; id %u, expected %u)\n", b->rank, done, cnt, ecnt,
								(unsigned) t1[2] | ((unsigned) t1[3] << 8) | ((unsigned) t1[0] << 16), (unsigned) (uint32_t) q[0]);
				}
			}
			else if ((done % 53) != 52)
			{
				if (b->bad++ < 3)
					fprintf(stderr, "backend %d: wait %d failed (%d): %s\n", b->rank, done, rc, ndbhip_last_error());
			}
			head = (head + 1) % b->inflight;
			held--;
			done++;
		}
	}
	(void) ndb_client_disconnect(c);
	free(ticket);
	free(qs);
	return NULL;
}

int
main(int argc, char **argv)
{
	const int	threads = argc > 1 ? atoi(argv[1]) : 8, inflight = argc > 2 ? atoi(argv[2]) : 4, nq = argc > 3 ? atoi(argv[3]) : 2000;
	pthread_t	ot, *bt = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) threads);
	backend_t  *bs = (backend_t *) calloc((size_t) threads, sizeof(backend_t));
	int			bad = 0;

	if (ndb_service_create(NAME, DIM, MAXK, threads * inflight + 1, &svc))	/* (a backend fills its window before it waits: fewer slots than requests in flight would be a deadlock of the test's own making) */
	{
		fprintf(stderr, "ndb_service_create: %s\n", ndbhip_last_error());
		return 2;
	}
	(void) ndb_service_publish(svc, 0, 0, 4);
	pthread_create(&ot, NULL, owner, NULL);
	for (int i = 0; i < threads; i++)
	{
		bs[i].rank = i;
		bs[i].inflight = inflight;
		bs[i].nq = nq;
		pthread_create(&bt[i], NULL, backend, &bs[i]);
	}
	for (int i = 0; i < threads; i++)
	{
		pthread_join(bt[i], NULL);
		bad += bs[i].bad;
	}
	(void) ndb_service_stop(svc);
	pthread_join(ot, NULL);
	(void) ndb_service_destroy(svc);
	printf("service_stress: %d backends x %d in flight x %d queries: %d bad answers, owner rc %d\n", threads, inflight, nq, bad, owner_rc);
	free(bt);
	free(bs);
	return (bad || owner_rc) ? 1 : 0;
}
