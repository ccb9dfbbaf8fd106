/*
 * ndb_oracle_hnsw2.c — TEST INFRASTRUCTURE (see ndb_oracle.h).  The `intended` HNSW: what SURVEY 8f-2 asks for next
 * to the bug-compatible hnswInsertNode / hnswSearch of ndb_oracle.c.  parity unpinned: the reference holds no test,
 * fixture or caller for this algorithm — it is the textbook one its dead file sketches.
 *
 * What it keeps from the reference: the page-level data model (node = block, `levels`, dense 16-level neighbour
 * arrays of 2m slots, entry point / entry level in the meta page: src/index/hnsw_am.c:108-181), the injected level
 * draws (hnswGetRandomLevel, :1143-1161), L2 for every build-time comparison (hnswInsertNode always passes strategy
 * 1, :2360-2520) and the best-first layer search of src/scan/hnsw_scan.c:379-483, 645-844 as the specification of
 * the search: a candidate set ordered by distance, a result set of at most ef, expansion of the nearest unexpanded
 * candidate until none is nearer than the worst result.
 * What it repairs (each one is a reason recall@10 of the reference's graph is 0.0: DESIGN.md section 7):
 *   - the greedy descent's result is USED as the next level's entry point (hnsw_am.c:2155-2286 computes and drops it);
 *   - every level is searched on ITS OWN links (the reference reruns the level-0 walk for every level);
 *   - a full neighbour list is PRUNED to its nearest `cap` entries when a back-link arrives (hnsw_am.c:2503-2513 is
 *     the unreachable branch; the reference drops the link);
 *   - hnsw_scan.c's quirks (the bound read from an unsorted slot, inserts into a full heap dropped, the entry point
 *     entered at distance 0) are not reproduced.
 *
 * Arithmetic (one definition for host and device, so that a graph can be compared slot for slot): the squared L2
 * distance in fp64, every term (double) fl32(a_i - b_i) squared, summed by a FIXED tree — 64 strided partial sums
 * (element i goes to partial i mod 64, in increasing i) folded by the butterfly 32, 16, 8, 4, 2, 1 — which is what a
 * 64-lane wave computes with one lane per partial; no fused multiply-add (-ffp-contract=off on both sides).
 * Every comparison is on the pair (d2, block number): no ties anywhere.
 *
 * Build schedule (part of the definition, so that the device can run a batch's searches in parallel and still
 * produce THIS graph): inserts are taken in batches of clamp(nodes so far / batch_div, 1, batch_max); every member of
 * a batch searches the graph as it stood when the batch began (members do not see one another), then the members'
 * links are applied one after the other in insertion order.  batch_max = 1 is the sequential textbook insert.
 */
#include "ndb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define H2_MAXLEV NDBO_HNSW_MAX_LEVEL

double
ndbo_h2_dist2(const float *a, const float *b, int dim)
{
	double		p[64];
	int			i,
				off;

	for (i = 0; i < 64; i++)
		p[i] = 0.0;
	for (i = 0; i < dim; i++)
	{
		const float d = a[i] - b[i];
		const double dd = (double) d;

		p[i & 63] += dd * dd;
	}
	for (off = 32; off > 0; off >>= 1)
		for (i = 0; i < off; i++)
			p[i] = p[i] + p[i + off];	/* lane i and lane i ^ off compute the same sum: the butterfly's value */
	return p[0];
}

static inline const float *
h2_vec(const ndbo_hnsw *g, uint32_t b)
{
	return g->vecs + (size_t) b * g->dim;
}

/*
 * WALK ROWS (round 5; SURVEY 8f-4: fp16 node storage).  A walk is bound by the bytes of the rows it fetches at random;
 * it can run on what a halfvec column of the same rows would hold — every element through the reference's own
 * float4_to_fp16 (src/types/quantization.c:141-168: mantissa truncated, subnormal results flushed to signed zero)
 * and back through fp16_to_float (:170-218) — as long as the rows it RETURNS are ordered by the definition's distance
 * to the float4 rows: ndbo_h2_search_w16 walks on the halves and re-scores the ef entries of the final result set.
 * Arithmetic on the halves (again one definition for host and device): the squared L2 distance in fp64, every term
 * (double) fl32(q_i - w_i) squared, summed by a fixed tree whose 64 partial sums take the elements in GROUPS OF FOUR —
 * element i goes to partial (i / 4) mod 64, in increasing i — folded by the same butterfly: what a 64-lane wave computes
 * when every lane loads 8 bytes (four halves) of the row per request.
 */
void
ndbo_h2_walk_rows(const float *vecs, int64_t nel, uint16_t *out)
{
	int64_t		i;

	for (i = 0; i < nel; i++)
		out[i] = ndbo_float4_to_fp16(vecs[i]);
}

double
ndbo_h2_dist2_w16(const float *q, const uint16_t *w, int dim)
{
	double		p[64];
	int			i,
				off;

	for (i = 0; i < 64; i++)
		p[i] = 0.0;
	for (i = 0; i < dim; i++)
	{
		const float d = q[i] - ndbo_fp16_to_float(w[i]);
		const double dd = (double) d;

		p[(i >> 2) & 63] += dd * dd;
	}
	for (off = 32; off > 0; off >>= 1)
		for (i = 0; i < off; i++)
			p[i] = p[i] + p[i + off];
	return p[0];
}

/* d2(q, node e) as a walk sees it: on the walk rows when there are any, else the definition's */
static inline double
h2_walk_d2(const ndbo_hnsw *g, const uint16_t *w16, const float *q, uint32_t e)
{
	return w16 ? ndbo_h2_dist2_w16(q, w16 + (size_t) e * g->dim, g->dim) : ndbo_h2_dist2(q, h2_vec(g, e), g->dim);
}

/*
 * THE OPERATOR CLASS'S METRIC (round 6; SURVEY Q1: "metric comes from the opclass", hnsw_am.c:918-921 hands sk_strategy to
 * hnswSearch, neurondb--1.0.sql:2941-2965 binds <-> / <=> / <#> to strategies 1 / 2 / 3).  The graph's links stay L2 (Q12:
 * hnswInsertNode always passes strategy 1); what the strategy decides is the ORDER a search walks and returns in:
 *   walk key (descent, layer search; fp64, the fixed 64-partial tree of ndbo_h2_dist2 / _w16 — element i to partial i mod 64 on
 *   float4 rows, (i / 4) mod 64 on walk rows — products of two float4 values are exact in fp64):
 *     1  sum (double) fl32(q_i - x_i) squared                                   (ndbo_h2_dist2[_w16]; unchanged)
 *     2  -dot * rinv(x) with dot = sum q_i x_i and rinv(x) = 1 / sqrt(nx), nx = sum x_i x_i by that tree over the row the walk
 *        reads (0 for a row of zeros): the order of 1 - dot / (|q| |x|) — hnswComputeDistance case 2, hnsw_am.c:1321-1332 —
 *        without the query's own norm (one positive factor for all of a query's keys) and with the row's factor a
 *        constant of the node: the device keeps rinv per node and an evaluation costs what an inner product costs
 *        (the first version, 1 - dot / (sqrt(nq) sqrt(nx)) with nx summed again at every evaluation, walked at 0.7 of the
 *        L2 walk's rate: round 6 bench, 2.1 against 3.1 M queries/s)
 *     3  -dot                                                                   (case 3, :1334-1337)
 *   every comparison on (key, block), as before;
 *   returned distances: strategy 1 as before ((float) sqrt(d2) of the float4 rows); strategies 2 and 3: the (at most ef) entries
 *   of the final result set are scored with hnswComputeDistance's OWN arithmetic on the float4 rows (ndbo_hnsw_distance:
 *   sequential, fp32 products widened and summed in fp64, hnsw_am.c:1301-1345), ordered by (that float4, block), and the k
 *   nearest returned with those values — what `ORDER BY v <=> $q` compares.  One evaluation each, counted.
 */
static void
h2_dot_nx(const float *q, const float *x, const uint16_t *w, int dim, double *dot, double *nx)
{
	double		pd[64],
				pn[64];
	int			i,
				off;

	for (i = 0; i < 64; i++)
		pd[i] = pn[i] = 0.0;
	for (i = 0; i < dim; i++)
	{
		const double xv = (double) (w ? ndbo_fp16_to_float(w[i]) : x[i]);
		const int	p = w ? ((i >> 2) & 63) : (i & 63);

		pd[p] += (double) q[i] * xv;
		pn[p] += xv * xv;
	}
	for (off = 32; off > 0; off >>= 1)
		for (i = 0; i < off; i++)
		{
			pd[i] = pd[i] + pd[i + off];
			pn[i] = pn[i] + pn[i + off];
		}
	*dot = pd[0];
	*nx = pn[0];
}

/* rinv of node e's row as the walk reads it (w16: its walk row): 1 / sqrt(sum of squares by the tree), 0 for a row of zeros */
double
ndbo_h2_rinv(const ndbo_hnsw *g, const uint16_t *w16, uint32_t e)
{
	double		dot,
				nx;
	const float *x = h2_vec(g, e);

	h2_dot_nx(x, w16 ? NULL : x, w16 ? w16 + (size_t) e * g->dim : NULL, g->dim, &dot, &nx);
	return nx > 0.0 ? 1.0 / sqrt(nx) : 0.0;
}

/* the walk key of (query, node e) under `strategy` */
double
ndbo_h2_walk_key(const ndbo_hnsw *g, const uint16_t *w16, const float *q, uint32_t e, int strategy)
{
	double		dot,
				nx;

	if (strategy == 1)
		return h2_walk_d2(g, w16, q, e);
	h2_dot_nx(q, w16 ? NULL : h2_vec(g, e), w16 ? w16 + (size_t) e * g->dim : NULL, g->dim, &dot, &nx);
	if (strategy == 3)
		return -dot;
	return -dot * (nx > 0.0 ? 1.0 / sqrt(nx) : 0.0);
}

/* (strategy, query norm) of the walk in progress: the layer search and the greedy step below score through this */
typedef struct h2_metric
{
	int			strategy;
}			h2_metric;

static inline double
h2_key(const ndbo_hnsw *g, const uint16_t *w16, const float *q, uint32_t e, const h2_metric *mt)
{
	return mt ? ndbo_h2_walk_key(g, w16, q, e, mt->strategy) : h2_walk_d2(g, w16, q, e);
}

static inline uint32_t *
h2_nbrs(const ndbo_hnsw *g, uint32_t b, int level)
{
	return g->nbrs + ((size_t) b * H2_MAXLEV + level) * 2 * (size_t) g->m;
}

static inline int
h2_cap(const ndbo_hnsw *g, int level)
{
	return level == 0 ? 2 * g->m : g->m;
}

/* (d2, id) < (e2, jd) */
static inline int
h2_less(double d2, uint32_t id, double e2, uint32_t jd)
{
	return d2 < e2 || (d2 == e2 && id < jd);
}

/*
 * Best-first search of one layer.  Entry points ep[0..nep) with distances; results: the (at most ef) nearest found,
 * ascending by (d2, id).  `visited`: nblocks bytes, all zero on entry and on return.
 * The candidate set is the result set's unexpanded part: an element pushed out of the results is farther than
 * everything left in them, so the textbook loop would stop before expanding it.
 */
static int
h2_search_layer_w(const ndbo_hnsw *g, const uint16_t *w16, const h2_metric *mt, const float *q, const uint32_t *ep, const double *epd, int nep, int ef,
				  int level, uint32_t *out_ids, double *out_d2, int64_t *evals, uint8_t *visited)
{
	uint32_t   *wid = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) (ef + 1));
	double	   *wd = (double *) malloc(sizeof(double) * (size_t) (ef + 1));
	uint8_t    *wx = (uint8_t *) malloc((size_t) (ef + 1));
	uint32_t   *log = NULL;
	size_t		nlog = 0,
				caplog = 0;
	int			nw = 0,
				i;

#define H2_MARK(b) do { if (nlog == caplog) { caplog = caplog ? 2 * caplog : 1024; \
		log = (uint32_t *) realloc(log, caplog * sizeof(uint32_t)); } log[nlog++] = (b); visited[b] = 1; } while (0)
	for (i = 0; i < nep && nw < ef; i++)
	{
		if (visited[ep[i]])
			continue;
		H2_MARK(ep[i]);
		wid[nw] = ep[i];
		wd[nw] = epd[i];
		wx[nw] = 0;
		nw++;
	}
	for (;;)
	{
		int			best = -1,
					j;
		uint32_t	c;
		const uint32_t *nb;
		int			cnt;

		for (i = 0; i < nw; i++)
			if (!wx[i] && (best < 0 || h2_less(wd[i], wid[i], wd[best], wid[best])))
				best = i;
		if (best < 0)
			break;
		wx[best] = 1;
		c = wid[best];
		nb = h2_nbrs(g, c, level);
		/* a node page holds lists for levels 0 .. its own only (hnsw_am.c:124-181: neighbors[level + 1][2m]); what the
		 * reference's inserts wrote beyond that (Q12 / Q21) is not part of the index */
		cnt = g->levels[c] >= level ? g->ncount[(size_t) c * H2_MAXLEV + level] : 0;
		for (j = 0; j < cnt; j++)
		{
			const uint32_t e = nb[j];
			double		d;
			int			worst = 0;

			if (e == NDBO_INVALID_BLOCK || e >= g->nblocks || visited[e])
				continue;
			H2_MARK(e);
			d = h2_key(g, w16, q, e, mt);
			if (evals)
				(*evals)++;
			if (nw < ef)
			{
				wid[nw] = e;
				wd[nw] = d;
				wx[nw] = 0;
				nw++;
				continue;
			}
			for (i = 1; i < nw; i++)
				if (h2_less(wd[worst], wid[worst], wd[i], wid[i]))
					worst = i;
			if (h2_less(d, e, wd[worst], wid[worst]))
			{
				wid[worst] = e;
				wd[worst] = d;
				wx[worst] = 0;
			}
		}
	}
	/* ascending (d2, id): insertion sort (ef <= a few hundred) */
	for (i = 0; i < nw; i++)
	{
		int			j = i;
		const uint32_t id = wid[i];
		const double d = wd[i];

		while (j > 0 && h2_less(d, id, out_d2[j - 1], out_ids[j - 1]))
		{
			out_d2[j] = out_d2[j - 1];
			out_ids[j] = out_ids[j - 1];
			j--;
		}
		out_d2[j] = d;
		out_ids[j] = id;
	}
	for (i = 0; i < (int) nlog; i++)
		visited[log[i]] = 0;
	free(log);
	free(wid);
	free(wd);
	free(wx);
	return nw;
#undef H2_MARK
}

int
ndbo_h2_search_layer(const ndbo_hnsw *g, const float *q, const uint32_t *ep, const double *epd, int nep, int ef,
					 int level, uint32_t *out_ids, double *out_d2, int64_t *evals, uint8_t *visited)
{
	return h2_search_layer_w(g, NULL, NULL, q, ep, epd, nep, ef, level, out_ids, out_d2, evals, visited);
}

/* greedy step of the upper layers: from `cur`, move to the nearest neighbour at `level` while one is nearer */
static void
h2_greedy_w(const ndbo_hnsw *g, const uint16_t *w16, const h2_metric *mt, const float *q, int level, uint32_t *cur, double *curd, int64_t *evals)
{
	for (;;)
	{
		const uint32_t *nb = h2_nbrs(g, *cur, level);
		const int	cnt = g->levels[*cur] >= level ? g->ncount[(size_t) (*cur) * H2_MAXLEV + level] : 0;
		uint32_t	bid = *cur;
		double		bd = *curd;
		int			j;

		for (j = 0; j < cnt; j++)
		{
			const uint32_t e = nb[j];
			double		d;

			if (e == NDBO_INVALID_BLOCK || e >= g->nblocks)
				continue;
			d = h2_key(g, w16, q, e, mt);
			if (evals)
				(*evals)++;
			if (h2_less(d, e, bd, bid))
			{
				bd = d;
				bid = e;
			}
		}
		if (bid == *cur)
			return;
		*cur = bid;
		*curd = bd;
	}
}

static void
h2_greedy(const ndbo_hnsw *g, const float *q, int level, uint32_t *cur, double *curd, int64_t *evals)
{
	h2_greedy_w(g, NULL, NULL, q, level, cur, curd, evals);
}

/* kNN query: greedy descent to level 1, best-first search with ef at level 0, the k nearest ascending.  Distances
 * come back as (float) sqrt(d2).  Returns the count.
 * w16 != NULL (ndbo_h2_search_w16): descent and layer search on the walk rows; the result set's (at most ef) entries
 * are then scored against the float4 rows with the definition's arithmetic (ndbo_h2_dist2), ordered by that
 * (d2, block), and the k nearest returned with THOSE distances — an evaluation each, counted. */
static int
h2_search_w(const ndbo_hnsw *g, const uint16_t *w16, int strategy, const float *query, int ef, int k, uint32_t *out_blocks, float *out_dist,
			int64_t *evals)
{
	h2_metric	mtv;
	const h2_metric *mt = NULL;
	uint8_t    *visited;
	uint32_t   *ids;
	double	   *d2;
	uint32_t	cur;
	double		curd;
	int			lc,
				n,
				i;

	if (g->entry_point == NDBO_INVALID_BLOCK || k < 1)
		return 0;
	if (ef < k)
		ef = k;
	if (strategy != 1)
	{
		mtv.strategy = strategy;
		mt = &mtv;
	}
	cur = g->entry_point;
	curd = h2_key(g, w16, query, cur, mt);
	if (evals)
		(*evals)++;
	for (lc = g->entry_level; lc >= 1; lc--)
		h2_greedy_w(g, w16, mt, query, lc, &cur, &curd, evals);
	visited = (uint8_t *) calloc(g->nblocks, 1);
	ids = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) ef);
	d2 = (double *) malloc(sizeof(double) * (size_t) ef);
	n = h2_search_layer_w(g, w16, mt, query, &cur, &curd, 1, ef, 0, ids, d2, evals, visited);
	if (mt)
	{
		/* the result set under hnswComputeDistance's arithmetic (float4 rows), ascending (that float4, block) */
		float	   *fd = (float *) malloc(sizeof(float) * (size_t) (n > 0 ? n : 1));

		for (i = 0; i < n; i++)
			fd[i] = ndbo_hnsw_distance(query, h2_vec(g, ids[i]), g->dim, strategy, NULL);
		if (evals)
			*evals += n;
		for (i = 1; i < n; i++)
		{
			const uint32_t id = ids[i];
			const float d = fd[i];
			int			j = i;

			while (j > 0 && (d < fd[j - 1] || (d == fd[j - 1] && id < ids[j - 1])))
			{
				fd[j] = fd[j - 1];
				ids[j] = ids[j - 1];
				j--;
			}
			fd[j] = d;
			ids[j] = id;
		}
		if (n > k)
			n = k;
		for (i = 0; i < n; i++)
		{
			out_blocks[i] = ids[i];
			out_dist[i] = fd[i];
		}
		free(fd);
		free(visited);
		free(ids);
		free(d2);
		return n;
	}
	if (w16)
	{
		/* re-score on the float4 rows, then ascending (d2, id) again (insertion sort) */
		for (i = 0; i < n; i++)
			d2[i] = ndbo_h2_dist2(query, h2_vec(g, ids[i]), g->dim);
		if (evals)
			*evals += n;
		for (i = 1; i < n; i++)
		{
			const uint32_t id = ids[i];
			const double d = d2[i];
			int			j = i;

			while (j > 0 && h2_less(d, id, d2[j - 1], ids[j - 1]))
			{
				d2[j] = d2[j - 1];
				ids[j] = ids[j - 1];
				j--;
			}
			d2[j] = d;
			ids[j] = id;
		}
	}
	if (n > k)
		n = k;
	for (i = 0; i < n; i++)
	{
		out_blocks[i] = ids[i];
		out_dist[i] = (float) sqrt(d2[i]);
	}
	free(visited);
	free(ids);
	free(d2);
	return n;
}

int
ndbo_h2_search(const ndbo_hnsw *g, const float *query, int ef, int k, uint32_t *out_blocks, float *out_dist,
			   int64_t *evals)
{
	return h2_search_w(g, NULL, 1, query, ef, k, out_blocks, out_dist, evals);
}

/* the same under the operator class's strategy (1 L2 = ndbo_h2_search, 2 cosine, 3 negative inner product; see
 * "THE OPERATOR CLASS'S METRIC" above); w16 NULL = walk on the float4 rows.  -1: unknown strategy (the reference's ERROR) */
int
ndbo_h2_search_s(const ndbo_hnsw *g, const uint16_t *w16, int strategy, const float *query, int ef, int k, uint32_t *out_blocks,
				 float *out_dist, int64_t *evals)
{
	if (strategy < 1 || strategy > 3)
		return -1;
	return h2_search_w(g, w16, strategy, query, ef, k, out_blocks, out_dist, evals);
}

/* w16: [nblocks][dim] walk rows of g's vectors (ndbo_h2_walk_rows over g->vecs) */
int
ndbo_h2_search_w16(const ndbo_hnsw *g, const uint16_t *w16, const float *query, int ef, int k, uint32_t *out_blocks,
				   float *out_dist, int64_t *evals)
{
	if (!w16)
		return -1;
	return h2_search_w(g, w16, 1, query, ef, k, out_blocks, out_dist, evals);
}

/* what the search phase of one insert leaves: per level lc <= min(level, entry level at that time) the selected
 * neighbours, ascending by (d2, id) */
typedef struct h2_sel
{
	int			top;			/* highest level with a selection (-1: the graph was empty) */
	int			n[H2_MAXLEV];
	uint32_t   *ids;			/* [H2_MAXLEV][2m] */
	double	   *d2;
}			h2_sel;

/* the m the new node links to out of the layer search's results (ascending): select bit 0 clear = the nearest m;
 * set = the textbook heuristic — a candidate is taken unless it is nearer to one already taken than to the new node;
 * bit 2 (with bit 0): the places the heuristic leaves empty go to the nearest candidates it passed over, in order
 * (the paper's keepPrunedConnections: a node inside a tight cluster keeps m links instead of a handful) */
static int
h2_select(const ndbo_hnsw *g, const float *base, const uint32_t *cid, const double *cd2, int nc, int M, int select,
		  uint32_t *out, double *outd)
{
	int			n = 0,
				i,
				j;

	for (i = 0; i < nc && n < M; i++)
	{
		int			ok = 1;

		if (select & 1)
			for (j = 0; j < n; j++)
				if (ndbo_h2_dist2(h2_vec(g, cid[i]), h2_vec(g, out[j]), g->dim) < cd2[i])
				{
					ok = 0;
					break;
				}
		if (ok)
		{
			out[n] = cid[i];
			outd[n] = cd2[i];
			n++;
		}
	}
	if ((select & 5) == 5)
	{
		const int	n0 = n;

		for (i = 0; i < nc && n < M; i++)
		{
			int			taken = 0;

			for (j = 0; j < n0; j++)
				if (out[j] == cid[i])
					taken = 1;
			if (!taken)
			{
				out[n] = cid[i];
				outd[n] = cd2[i];
				n++;
			}
		}
	}
	(void) base;
	return n;
}

static void
h2_insert_search(const ndbo_hnsw *g, const float *vec, int level, int select, h2_sel *s, uint8_t *visited)
{
	const int	efc = g->ef_construction, m = g->m, w = 2 * g->m;
	/* bit 1 of `select`: a new node takes up to 2m links at level 0 (the list's capacity there) instead of m */
	const int	m0 = (select & 2) ? 2 * m : m;
	uint32_t	cur;
	double		curd;
	uint32_t   *wid;
	double	   *wd;
	int			lc;

	memset(s->n, 0, sizeof(s->n));
	s->top = -1;
	if (g->entry_point == NDBO_INVALID_BLOCK)
		return;
	cur = g->entry_point;
	curd = ndbo_h2_dist2(vec, h2_vec(g, cur), g->dim);
	for (lc = g->entry_level; lc > level; lc--)
		h2_greedy(g, vec, lc, &cur, &curd, NULL);
	wid = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) efc);
	wd = (double *) malloc(sizeof(double) * (size_t) efc);
	s->top = level < g->entry_level ? level : g->entry_level;
	for (lc = s->top; lc >= 0; lc--)
	{
		const int	nw = ndbo_h2_search_layer(g, vec, &cur, &curd, 1, efc, lc, wid, wd, NULL, visited);

		s->n[lc] = h2_select(g, vec, wid, wd, nw, lc == 0 ? m0 : m, select & 5, s->ids + (size_t) lc * w, s->d2 + (size_t) lc * w);
		cur = wid[0];			/* the nearest found is the next level's entry point */
		curd = wd[0];
	}
	free(wid);
	free(wd);
}

/* a back-link x -> into e's list at `level`: appended while there is room, else the list becomes the `cap` entries
 * of (list + x) chosen by the same rule as above around e, ascending by (d2 to e, id) */
static void
h2_backlink(ndbo_hnsw *g, uint32_t e, int level, uint32_t x, double dxe, int select)
{
	const int	cap = h2_cap(g, level);
	uint32_t   *nb = h2_nbrs(g, e, level);
	int16_t    *pc = &g->ncount[(size_t) e * H2_MAXLEV + level];
	uint32_t	cid[2 * 64 + 1],
				kid[2 * 64 + 1];
	double		cd[2 * 64 + 1],
				kd[2 * 64 + 1];
	int			cnt = *pc,
				i,
				n;

	if (cnt < cap)
	{
		nb[cnt] = x;
		*pc = (int16_t) (cnt + 1);
		return;
	}
	/* candidates ascending by (d2 to e, id) */
	n = 0;
	for (i = 0; i <= cnt; i++)
	{
		const uint32_t id = i < cnt ? nb[i] : x;
		const double d = i < cnt ? ndbo_h2_dist2(h2_vec(g, e), h2_vec(g, id), g->dim) : dxe;
		int			j = n;

		while (j > 0 && h2_less(d, id, cd[j - 1], cid[j - 1]))
		{
			cd[j] = cd[j - 1];
			cid[j] = cid[j - 1];
			j--;
		}
		cd[j] = d;
		cid[j] = id;
		n++;
	}
	n = h2_select(g, h2_vec(g, e), cid, cd, n, cap, select & 5, kid, kd);
	for (i = 0; i < 2 * g->m; i++)
		nb[i] = i < n ? kid[i] : NDBO_INVALID_BLOCK;
	*pc = (int16_t) n;
}

static void
h2_insert_apply(ndbo_hnsw *g, uint32_t x, int level, const h2_sel *s, int select)
{
	const int	m = 2 * g->m;		/* row width of the selections */
	int			lc,
				i;

	for (lc = s->top; lc >= 0; lc--)
	{
		uint32_t   *nb = h2_nbrs(g, x, lc);

		for (i = 0; i < s->n[lc]; i++)
			nb[i] = s->ids[(size_t) lc * m + i];
		g->ncount[(size_t) x * H2_MAXLEV + lc] = (int16_t) s->n[lc];
		for (i = 0; i < s->n[lc]; i++)
			h2_backlink(g, s->ids[(size_t) lc * m + i], lc, x, s->d2[(size_t) lc * m + i], select);
	}
	if (g->entry_point == NDBO_INVALID_BLOCK || level > g->entry_level)
	{
		g->entry_point = x;
		g->entry_level = level;
		if (level > g->max_level)
			g->max_level = level;
	}
}

/*
 * Build over n rows, levels injected, the batch schedule of the header.  On an empty graph row i becomes block i + 1; on a graph
 * that holds nodes already (round 6: hnswinsert under `intended`, src/index/hnsw_am.c:478-538 — one row, one batch — and bulk
 * appends) row i becomes block nblocks + i and the schedule goes on from the nodes inserted so far.  select: bit 0: 0 = the
 * nearest, 1 = the heuristic; bit 1: a new node takes up to 2m links at level 0 instead of m.  Returns the number of
 * batches, -1 when the graph's arrays cannot hold the rows.
 */
int
ndbo_h2_build(ndbo_hnsw *g, const float *vecs, const ndbo_tid *tids, int64_t n, const int *levels, int batch_div,
			  int batch_max, int select)
{
	const int	m = g->m;
	const int64_t base = g->nblocks > 0 ? (int64_t) g->nblocks - 1 : 0;	/* nodes there already: blocks 1 .. base */
	int64_t		done = 0;
	int			nbatches = 0;
	uint8_t    *visited;
	h2_sel	   *sel = NULL;
	int64_t		capsel = 0;

	if (base + n + 1 > (int64_t) g->cap_blocks)
		return -1;
	visited = (uint8_t *) calloc((size_t) (base + n) + 2, 1);

	if (batch_div < 1)
		batch_div = 1;
	if (batch_max < 1)
		batch_max = 1;
	while (done < n)
	{
		int64_t		b = (base + done) / batch_div,	/* (nodes the relation holds: = g->inserted on a graph nothing was deleted from) */
					i;

		if (b < 1)
			b = 1;
		if (b > batch_max)
			b = batch_max;
		if (b > n - done)
			b = n - done;
		if (b > capsel)
		{
			sel = (h2_sel *) realloc(sel, sizeof(h2_sel) * (size_t) b);
			for (i = capsel; i < b; i++)
			{
				sel[i].ids = (uint32_t *) malloc(sizeof(uint32_t) * H2_MAXLEV * 2 * (size_t) m);
				sel[i].d2 = (double *) malloc(sizeof(double) * H2_MAXLEV * 2 * (size_t) m);
			}
			capsel = b;
		}
		/* the members' vectors are in place before anyone searches (a member is not reachable until it is linked) */
		for (i = 0; i < b; i++)
		{
			const uint32_t x = (uint32_t) (base + done + i + 1);
			int			lev = levels[done + i];

			if (lev < 0)
				lev = 0;
			if (lev > H2_MAXLEV - 1)
				lev = H2_MAXLEV - 1;
			memcpy(g->vecs + (size_t) x * g->dim, vecs + (size_t) (done + i) * g->dim, sizeof(float) * (size_t) g->dim);
			if (tids)
				g->heap_tids[x] = tids[done + i];
			g->levels[x] = lev;
		}
		for (i = 0; i < b; i++)
			h2_insert_search(g, vecs + (size_t) (done + i) * g->dim, g->levels[base + done + i + 1], select, &sel[i], visited);
		for (i = 0; i < b; i++)
		{
			const uint32_t x = (uint32_t) (base + done + i + 1);

			g->nblocks = x + 1;		/* (visible to validity checks only once it can be named) */
			h2_insert_apply(g, x, g->levels[x], &sel[i], select);
			g->inserted++;
		}
		done += b;
		nbatches++;
	}
	{
		int64_t		i;

		for (i = 0; i < capsel; i++)
		{
			free(sel[i].ids);
			free(sel[i].d2);
		}
	}
	free(sel);
	free(visited);
	return nbatches;
}
