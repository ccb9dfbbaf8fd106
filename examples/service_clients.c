/*
 * Backends of the device-owner service in plain C: THREADS connections to the ring `name`, each keeping INFLIGHT
 * single-query requests in flight (ndb_client_submit / ndb_client_wait, include/ndb_service.h) — what that many
 * PostgreSQL backends running ivfrescan + ivfgettuple put on the ring, without a Python interpreter per backend
 * between the futex and the next request.  Queries come from the bench's generator (ndbhip_gen_rows_host: the
 * clustered table's own components).  The first CHECK queries of every thread are written with their answers to
 * OUT for the caller's oracle check (tools/service_bench.py --clients c).
 *
 * usage: service_clients NAME THREADS INFLIGHT QUERIES NPROBE K DIM COMPONENTS CHECK OUT
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ndb_service.h"
#include "ndbhip.h"

typedef struct
{
	const char *name;
	int			rank, inflight, nq, nprobe, k, dim, components, check;
	double		wall;
	int			rc;
	float	   *q;				/* [nq][dim] */
	uint8_t    *tids;			/* [check][k][6] */
	float	   *dist;			/* [check][k] */
	int		   *cnt;			/* [check] */
	pthread_barrier_t *start;
} backend_t;

static double
now(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void *
backend(void *arg)
{
	backend_t  *b = (backend_t *) arg;
	ndb_client *c = NULL;
	int		   *ticket = (int *) malloc(sizeof(int) * (size_t) b->inflight);
	int		   *which = (int *) malloc(sizeof(int) * (size_t) b->inflight);
	uint8_t    *t1 = (uint8_t *) malloc((size_t) b->k * 6);
	float	   *d1 = (float *) malloc(sizeof(float) * (size_t) b->k);
	int			head = 0, held = 0, next = 0, done = 0;

	b->rc = ndb_client_connect(b->name, &c);
	pthread_barrier_wait(b->start);
	if (b->rc)
		return NULL;
	const double t0 = now();

	while (done < b->nq && !b->rc)
	{
		while (next < b->nq && held < b->inflight && !b->rc)
		{
			const int	slot = (head + held) % b->inflight;

			b->rc = ndb_client_submit(c, b->q + (size_t) next * b->dim, 1, b->nprobe, b->k, 0, &ticket[slot]);
			which[slot] = next++;
			held++;
		}
		if (b->rc)
			break;
		const int	i = which[head];
		int			n = 0;

		b->rc = ndb_client_wait(c, ticket[head], i < b->check ? b->tids + (size_t) i * b->k * 6 : t1,
								i < b->check ? b->dist + (size_t) i * b->k : d1, &n, 20000);
		if (i < b->check)
			b->cnt[i] = n;
		head = (head + 1) % b->inflight;
		held--;
		done++;
	}
	b->wall = now() - t0;
	ndb_client_disconnect(c);
	free(ticket); free(which); free(t1); free(d1);
	return NULL;
}

int
main(int argc, char **argv)
{
	if (argc < 11)
	{
		fprintf(stderr, "usage: %s NAME THREADS INFLIGHT QUERIES NPROBE K DIM COMPONENTS CHECK OUT\n", argv[0]);
		return 2;
	}
	const char *name = argv[1];
	const int	nt = atoi(argv[2]), inflight = atoi(argv[3]), nq = atoi(argv[4]), nprobe = atoi(argv[5]), k = atoi(argv[6]),
		dim = atoi(argv[7]), comps = atoi(argv[8]);
	int			check = atoi(argv[9]);
	backend_t  *b = (backend_t *) calloc((size_t) nt, sizeof(backend_t));
	pthread_t  *th = (pthread_t *) calloc((size_t) nt, sizeof(pthread_t));
	pthread_barrier_t start;

	if (check > nq)
		check = nq;
	pthread_barrier_init(&start, NULL, (unsigned) nt);
	for (int r = 0; r < nt; r++)
	{
		b[r].name = name; b[r].rank = r; b[r].inflight = inflight; b[r].nq = nq; b[r].nprobe = nprobe; b[r].k = k;
		b[r].dim = dim; b[r].components = comps; b[r].check = check; b[r].start = &start;
		b[r].q = (float *) malloc(sizeof(float) * (size_t) nq * dim);
		b[r].tids = (uint8_t *) calloc((size_t) check * k, 6);
		b[r].dist = (float *) calloc((size_t) check * k, sizeof(float));
		b[r].cnt = (int *) calloc((size_t) check, sizeof(int));
		if (ndbhip_gen_rows_host(1, 0x5EED0002ull, 0x5EEDC0DEull, (int64_t) r * nq, nq, dim, comps, 0.1f, b[r].q))
		{
			fprintf(stderr, "generator failed\n");
			return 1;
		}
	}
	for (int r = 0; r < nt; r++)
		pthread_create(&th[r], NULL, backend, &b[r]);
	double		wall = 0.0;
	int			rc = 0;

	for (int r = 0; r < nt; r++)
	{
		pthread_join(th[r], NULL);
		if (b[r].wall > wall)
			wall = b[r].wall;
		if (b[r].rc)
			rc = b[r].rc;
	}
	FILE	   *f = fopen(argv[10], "wb");

	if (f)
	{
		for (int r = 0; r < nt; r++)
		{
			fwrite(b[r].q, sizeof(float), (size_t) check * dim, f);
			fwrite(b[r].cnt, sizeof(int), (size_t) check, f);
			fwrite(b[r].tids, 6, (size_t) check * k, f);
			fwrite(b[r].dist, sizeof(float), (size_t) check * k, f);
		}
		fclose(f);
	}
	printf("{\"rc\": %d, \"wall_s\": %.6f, \"queries\": %lld, \"queries_per_s\": %.1f}\n", rc, wall,
		   (long long) nt * nq, wall > 0 ? (double) nt * nq / wall : 0.0);
	return rc ? 1 : 0;
}
