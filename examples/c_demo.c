/*
 * c_demo.c — the C ABI from plain C, the way the PostgreSQL glue of INTEGRATION.md uses it: no Python, no
 * torch.  ambuild (ndbhip_ivf_build) on host rows, then the AM scan callbacks of ndb_am.h driven like the
 * executor drives ivf_am.c (rescan with the ORDER BY datum, gettuple until false), aminsert, ambulkdelete,
 * and the same for hnsw.  Self-checking: exits 0 and prints "c_demo: OK" when every answer is the expected one.
 *
 *   gcc -O2 -Iinclude examples/c_demo.c -Lneurondb_amd/lib -lndbhip -Wl,-rpath,$PWD/neurondb_amd/lib -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ndb_am.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, ndbhip_last_error()); return 1; } } while (0)
#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "c_demo: expectation failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

enum { N = 6000, DIM = 64, NLISTS = 16 };

static unsigned long long rng_state = 0x5EED0001ull;
static float
frand(void)
{
	/* splitmix64 -> sum of uniforms: cheap, roughly bell-shaped coordinates */
	float		s = 0.0f;
	int			i;

	for (i = 0; i < 4; i++)
	{
		unsigned long long z = (rng_state += 0x9E3779B97F4A7C15ull);

		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		s += (float) (z >> 40) / (float) (1 << 24);
	}
	return s - 2.0f;
}

/* the varlena image of a `vector` value (include/neurondb.h:35-41): int32 vl_len_, int16 dim, int16 unused */
static size_t
vector_datum(const float *v, int dim, unsigned char *out)
{
	int			len = 8 + 4 * dim;
	short		d = (short) dim, z = 0;

	memcpy(out, &len, 4);
	memcpy(out + 4, &d, 2);
	memcpy(out + 6, &z, 2);
	memcpy(out + 8, v, (size_t) 4 * dim);
	return (size_t) len;
}

static int
kill_odd_offsets(const ndb_item_pointer *ip, void *state)
{
	(*(long *) state)++;
	return (ip->posid & 1) ? 1 : 0;
}

int
main(void)
{
	static float rows[N * DIM];
	static unsigned char tids[N * 6];
	unsigned char datum[8 + 4 * DIM];
	ndbhip_ivf *ivf = NULL;
	ndbhip_hnsw *hn = NULL;
	ndb_index_scan *scan;
	ndb_scan_key key;
	int			i, iters = 0, n, rc;
	long		seen = 0;
	int64_t		removed = 0;

	if (ndbhip_device_count() < 1)
	{
		fprintf(stderr, "c_demo: no HIP device: %s\n", ndbhip_last_error());
		return 2;
	}
	CHECK(ndbhip_init(0));
	for (i = 0; i < N * DIM; i++)
		rows[i] = frand();
	for (i = 0; i < N; i++)
	{
		ndb_item_pointer ip = {0, (uint16_t) (i / 64), (uint16_t) (i % 64 + 1)};

		memcpy(tids + 6 * i, &ip, 6);
	}

	/* ---- ivf: ambuild, scan, aminsert, ambulkdelete ---- */
	CHECK(ndbhip_ivf_create(DIM, NLISTS, &ivf));
	CHECK(ndbhip_ivf_build(ivf, rows, tids, N, 50, &iters));
	EXPECT(iters >= 1 && ndbhip_ivf_nrows(ivf) == N);
	CHECK(ndb_am_set_guc("neurondb.ivf_probes", NLISTS));	/* probe everything: the answer must be exact */
	scan = ndb_ivfbeginscan(ivf, 0, 1);
	EXPECT(scan != NULL);
	key.sk_strategy = NDBHIP_STRATEGY_L2;
	key.sk_type = NDBHIP_TYPE_VECTOR;
	key.sk_argument = datum;
	key.sk_len = vector_datum(rows + 1234 * DIM, DIM, datum);
	CHECK(ndb_ivfrescan(scan, NULL, 0, &key, 1));
	n = 0;
	while ((rc = ndb_ivfgettuple(scan, NDB_FORWARD_SCAN_DIRECTION)) == 1)
	{
		if (n == 0)				/* the row itself, at distance 0 */
			EXPECT(scan->xs_heaptid.bi_lo == 1234 / 64 && scan->xs_heaptid.posid == 1234 % 64 + 1 &&
				   scan->xs_orderbyval == 0.0f && scan->xs_orderbynull == 0);
		else
			EXPECT(scan->xs_orderbyval >= 0.0f);
		n++;
	}
	CHECK(rc);
	EXPECT(n == 10);
	{
		ndb_item_pointer ip = {0, 900, 7};
		float		v[DIM];

		for (i = 0; i < DIM; i++)
			v[i] = 40.0f + (float) i;	/* far from everything */
		key.sk_len = vector_datum(v, DIM, datum);
		EXPECT(ndb_ivfinsert(ivf, datum, key.sk_len, NDBHIP_TYPE_VECTOR, &ip) == 1);
		EXPECT(ndb_ivfinsert(ivf, NULL, 0, NDBHIP_TYPE_VECTOR, &ip) == 0);	/* NULL value */
		CHECK(ndb_ivfrescan(scan, NULL, 0, &key, 1));
		EXPECT(ndb_ivfgettuple(scan, NDB_FORWARD_SCAN_DIRECTION) == 1);
		EXPECT(scan->xs_heaptid.bi_lo == 900 && scan->xs_heaptid.posid == 7 && scan->xs_orderbyval == 0.0f);
	}
	CHECK(ndb_ivfbulkdelete(ivf, kill_odd_offsets, &seen, &removed));
	EXPECT(seen == N + 1 && removed == N / 2 + 1 && ndbhip_ivf_nrows(ivf) == N + 1 - removed);
	ndb_ivfendscan(scan);

	/* ---- hnsw: build from host rows, scan, insert, bulkdelete ---- */
	{
		static int32_t levels[2000];

		for (i = 0; i < 2000; i++)
			levels[i] = ndb_hnsw_level_from_uniform(((double) (i * 7919 % 2000) + 0.5) / 2000.0, 0.36f);
		CHECK(ndbhip_hnsw_create(DIM, 16, &hn));
		CHECK(ndbhip_hnsw_insert(hn, rows, tids, 2000, levels, 200));
		scan = ndb_hnswbeginscan(hn, 0, 1);
		EXPECT(scan != NULL);
		key.sk_strategy = NDBHIP_STRATEGY_COSINE;
		key.sk_len = vector_datum(rows + 3 * DIM, DIM, datum);
		CHECK(ndb_hnswrescan(scan, NULL, 0, &key, 1));
		n = 0;
		while ((rc = ndb_hnswgettuple(scan, NDB_FORWARD_SCAN_DIRECTION)) == 1)
			n++;
		CHECK(rc);
		EXPECT(n == 10);
		{
			ndb_item_pointer ip = {0, 901, 3};

			EXPECT(ndb_hnswinsert(hn, datum, key.sk_len, NDBHIP_TYPE_VECTOR, &ip, 0) == 1);
		}
		seen = 0;
		CHECK(ndb_hnswbulkdelete(hn, kill_odd_offsets, &seen, &removed));
		EXPECT(seen == 2001 && removed == 1001);
		key.sk_strategy = 9;	/* no such operator: the reference raises ERROR (hnsw_am.c:1339-1343) */
		CHECK(ndb_hnswrescan(scan, NULL, 0, &key, 1));
		EXPECT(ndb_hnswgettuple(scan, NDB_FORWARD_SCAN_DIRECTION) < 0);
		ndb_hnswendscan(scan);
	}
	CHECK(ndbhip_hnsw_destroy(hn));
	CHECK(ndbhip_ivf_destroy(ivf));
	CHECK(ndbhip_shutdown());
	printf("c_demo: OK\n");
	return 0;
}
