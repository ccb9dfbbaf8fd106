/*
 * ndbhip_pages.cpp — PostgreSQL-free codec for the ivf index relation's 8 KB
 * pages: the packer that turns index pages into
 * the flat arrays of the HBM mirror, and the writer ambuild uses after a device
 * build.  Pure host code.
 *
 * Layouts follow the reference (paths relative to NeuronDB/):
 *   page header 24 B, line pointers 4 B (lp_off:15 | lp_flags:2 | lp_len:15),
 *   items grow down from pd_special, MAXALIGN = 8 (PostgreSQL bufpage.h);
 *   block 0: IvfMetaPageData at PageGetContents (src/index/ivf_am.c:75-89, 555-577)
 *   centroid page: special = MAXALIGN(sizeof(IvfCentroidData)) = 24 B, item =
 *     IvfCentroidData{int listId; int dim; int64 memberCount; BlockNumber firstBlock}
 *     + float4[dim] at +24 (:94-106, 640-711)
 *   list page: special = IvfListPageHeader{BlockNumber nextBlock; int32 entryCount},
 *     item = IvfListEntryData{ItemPointerData heapPtr; int16 dim} + float4[dim] at +8,
 *     entry size MAXALIGN(8) + MAXALIGN(4*dim) (:241-256, 977-979, 1101-1120)
 *
 * Format version 2 (this build): the reference PageAddItem()s every centroid onto ONE
 * page (quirk Q6: 15 lists at dim 128, 2 at dim 768).  Version 2 chains centroid pages:
 * the first 4 bytes of the centroid page's special space (24 zero bytes the reference
 * never touches) hold the next centroid block, 0 / InvalidBlockNumber = end.  An index
 * whose centroids fit one page is written as version 1, bit-compatible with the reference.
 */
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../include/ndbhip.h"

#define PG_BLCKSZ 8192
#define PG_PAGE_HEADER 24
#define PG_ITEMID 4
#define PG_MAXALIGN(x) (((size_t) (x) + 7) & ~(size_t) 7)
#define LP_NORMAL 1
#define LP_DEAD 3

#define IVF_MAGIC 0x49564646u
#define IVF_CENTROID_HDR 24		/* MAXALIGN(sizeof(IvfCentroidData)) */
#define IVF_ENTRY_HDR 8			/* MAXALIGN(sizeof(IvfListEntryData)) */

extern "C" const char *ndbhip_last_error(void);
int			ndbhip_pages_fail(int code, const char *msg);	/* sets the thread-local error text (ndbhip.hip) */

static inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void wr16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }
static inline void wr32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline void wr64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

struct PageView
{
	const uint8_t *p;
	uint16_t	lower, upper, special;
	bool		ok;
	int			nitems;
};

static PageView
page_view(const uint8_t *p)
{
	PageView	v;

	v.p = p;
	v.lower = rd16(p + 12);
	v.upper = rd16(p + 14);
	v.special = rd16(p + 16);
	/* PageIsNew: pd_upper == 0; sanity of the pointers */
	v.ok = v.upper != 0 && v.lower >= PG_PAGE_HEADER && v.lower <= v.upper && v.upper <= v.special &&
		v.special <= PG_BLCKSZ;
	v.nitems = v.ok ? (v.lower - PG_PAGE_HEADER) / PG_ITEMID : 0;	/* PageGetMaxOffsetNumber */
	return v;
}

/* line pointer i (0-based) -> item offset/len/flags */
static inline void
item_id(const PageView &v, int i, uint32_t &off, uint32_t &flags, uint32_t &len)
{
	const uint32_t w = rd32(v.p + PG_PAGE_HEADER + PG_ITEMID * i);

	off = w & 0x7FFFu;
	flags = (w >> 15) & 3u;
	len = (w >> 17) & 0x7FFFu;
}

struct ndbhip_ivf_page_info_s
{
	int			dim, nlists, nprobe, version, ncentroids;
	int64_t		inserted, live_rows;
};

/* walk: count (rows == NULL) or unpack */
static int
ivf_walk(const uint8_t *pages, uint32_t nblocks, ndbhip_ivf_page_info_s *info, float *centroids,
		 int64_t *list_len, float *rows, uint8_t *tids6)
{
	if (!pages || nblocks < 1)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "no pages");
	const PageView mv = page_view(pages);

	if (!mv.ok)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "block 0 is not an initialised page");
	const uint8_t *meta = pages + PG_PAGE_HEADER;

	if (rd32(meta) != IVF_MAGIC)	/* ivfgettuple: "Invalid magic number in metadata" (ivf_am.c:1943-1948) */
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf: invalid magic number in metadata");
	const int	version = (int) rd32(meta + 4);
	const int	nlists = (int) rd32(meta + 8);
	const int	nprobe = (int) rd32(meta + 12);
	const int	dim = (int) rd32(meta + 16);
	const uint32_t cblock0 = rd32(meta + 20);
	int64_t		inserted;

	memcpy(&inserted, meta + 24, 8);
	if (dim < 0 || dim > 32767)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf: bad dimension in metadata");
	int			nc = 0;
	int64_t		live = 0;
	std::vector<uint32_t> first_block;

	/* centroid page chain */
	for (uint32_t cb = cblock0, hops = 0; cb != NDBHIP_INVALID_BLOCK && cb != 0; hops++)
	{
		if (cb >= nblocks || hops > nblocks)
			return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf: centroid block out of range");
		const PageView cv = page_view(pages + (size_t) cb * PG_BLCKSZ);

		if (!cv.ok)
			break;
		for (int i = 0; i < cv.nitems; i++)
		{
			uint32_t	off, flags, len;

			item_id(cv, i, off, flags, len);
			if (off + IVF_CENTROID_HDR + (size_t) dim * 4 > PG_BLCKSZ)
				return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf: centroid item out of page");
			const uint8_t *c = cv.p + off;

			if ((int) rd32(c + 4) != dim)
				return ndbhip_pages_fail(NDBHIP_ERR_UNSUPPORTED, "ivf: centroid with a different dimension");
			if (centroids)
				memcpy(centroids + (size_t) nc * dim, c + IVF_CENTROID_HDR, (size_t) dim * 4);
			first_block.push_back(rd32(c + 16));
			nc++;
		}
		/* version 2 chain pointer in the special space; version 1 pages hold zeros there */
		cb = (version >= 2 && cv.special + 4 <= PG_BLCKSZ) ? rd32(cv.p + cv.special) : NDBHIP_INVALID_BLOCK;
	}

	/* list chains, in the order ivfCollectCandidates walks them (ivf_am.c:1793-1840) */
	for (int L = 0; L < nc; L++)
	{
		int64_t		n = 0;
		uint32_t	hops = 0;

		for (uint32_t lb = first_block[L]; lb != NDBHIP_INVALID_BLOCK; hops++)
		{
			if (lb >= nblocks || lb == 0 || hops > nblocks)
				return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf: list block out of range");
			const PageView lv = page_view(pages + (size_t) lb * PG_BLCKSZ);

			if (lv.ok)
			{
				for (int i = 0; i < lv.nitems; i++)
				{
					uint32_t	off, flags, len;

					item_id(lv, i, off, flags, len);
					if (flags == LP_DEAD || flags != LP_NORMAL)		/* ItemIdIsDead / unused pointer */
						continue;
					if (off + IVF_ENTRY_HDR + (size_t) dim * 4 > PG_BLCKSZ)
						return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "ivf: list entry out of page");
					const uint8_t *e = lv.p + off;

					if ((int16_t) rd16(e + 6) != dim)				/* entry->dim != dim: skipped (:1821) */
						continue;
					if (rows)
					{
						memcpy(rows + (size_t) (live + n) * dim, e + IVF_ENTRY_HDR, (size_t) dim * 4);
						memcpy(tids6 + (size_t) (live + n) * 6, e, 6);
					}
					n++;
				}
			}
			lb = (lv.special + 4 <= PG_BLCKSZ && lv.special >= PG_PAGE_HEADER) ? rd32(lv.p + lv.special)
				: NDBHIP_INVALID_BLOCK;
		}
		if (list_len)
			list_len[L] = n;
		live += n;
	}
	if (info)
	{
		info->dim = dim;
		info->nlists = nlists;
		info->nprobe = nprobe;
		info->version = version;
		info->ncentroids = nc;
		info->inserted = inserted;
		info->live_rows = live;
	}
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_pages_info(const uint8_t *pages, uint32_t nblocks, int *dim, int *nlists, int *ncentroids,
					  int64_t *live_rows, int *version)
{
	ndbhip_ivf_page_info_s info;
	int			rc = ivf_walk(pages, nblocks, &info, nullptr, nullptr, nullptr, nullptr);

	if (rc)
		return rc;
	if (dim) *dim = info.dim;
	if (nlists) *nlists = info.nlists;
	if (ncentroids) *ncentroids = info.ncentroids;
	if (live_rows) *live_rows = info.live_rows;
	if (version) *version = info.version;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_ivf_pages_unpack(const uint8_t *pages, uint32_t nblocks, float *centroids, int64_t *list_len,
						float *rows, uint8_t *tids6)
{
	if (!centroids || !list_len || !rows || !tids6)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "NULL output");
	return ivf_walk(pages, nblocks, nullptr, centroids, list_len, rows, tids6);
}

extern "C" int
ndbhip_ivf_load_pages(ndbhip_ivf **out, const uint8_t *pages, uint32_t nblocks)
{
	ndbhip_ivf_page_info_s info;
	int			rc = ivf_walk(pages, nblocks, &info, nullptr, nullptr, nullptr, nullptr);

	if (rc)
		return rc;
	if (!out)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "out is NULL");
	if (info.ncentroids < 1 || info.dim < 1)
		return ndbhip_pages_fail(NDBHIP_ERR_STATE, "ivf: index has no centroids block");	/* ivf_am.c:1616-1622 */
	std::vector<float> cent((size_t) info.ncentroids * info.dim);
	std::vector<int64_t> ll((size_t) info.ncentroids);
	std::vector<float> rows((size_t) (info.live_rows > 0 ? info.live_rows : 1) * info.dim);
	std::vector<uint8_t> tids((size_t) (info.live_rows > 0 ? info.live_rows : 1) * 6);

	rc = ivf_walk(pages, nblocks, nullptr, cent.data(), ll.data(), rows.data(), tids.data());
	if (rc)
		return rc;
	ndbhip_ivf *ix = nullptr;

	rc = ndbhip_ivf_create(info.dim, info.nlists > 0 ? info.nlists : info.ncentroids, &ix);
	if (rc)
		return rc;
	rc = ndbhip_ivf_set_centroids(ix, cent.data(), info.ncentroids);
	if (!rc)
		rc = ndbhip_ivf_load(ix, ll.data(), nullptr, rows.data(), tids.data(), info.live_rows);
	if (!rc)
		rc = ndbhip_ivf_set_nprobe(ix, info.nprobe);
	if (rc)
	{
		ndbhip_ivf_destroy(ix);
		return rc;
	}
	*out = ix;
	return NDBHIP_OK;
}

/* ------------------------------------------------------------------ */
/* writer                                                              */
/* ------------------------------------------------------------------ */

static void
page_init(uint8_t *p, size_t special_size)
{
	const uint16_t sp = (uint16_t) (PG_BLCKSZ - PG_MAXALIGN(special_size));

	memset(p, 0, PG_BLCKSZ);
	wr16(p + 12, PG_PAGE_HEADER);	/* pd_lower */
	wr16(p + 14, sp);				/* pd_upper */
	wr16(p + 16, sp);				/* pd_special */
	wr16(p + 18, PG_BLCKSZ | 4);	/* pd_pagesize_version */
}

/* PageGetFreeSpace */
static size_t
page_free(const uint8_t *p)
{
	const int	space = (int) rd16(p + 14) - (int) rd16(p + 12);

	return space < (int) PG_ITEMID ? 0 : (size_t) (space - PG_ITEMID);
}

/* PageAddItem at the next offset; returns the data pointer */
static uint8_t *
page_add(uint8_t *p, size_t size)
{
	const uint16_t lower = rd16(p + 12);
	const uint16_t upper = (uint16_t) (rd16(p + 14) - PG_MAXALIGN(size));

	wr32(p + lower, (uint32_t) upper | ((uint32_t) LP_NORMAL << 15) | ((uint32_t) size << 17));
	wr16(p + 12, lower + PG_ITEMID);
	wr16(p + 14, upper);
	return p + upper;
}

static int64_t
ivf_blocks_needed(int dim, int ncent, const int64_t *list_len)
{
	const size_t csize = PG_MAXALIGN(IVF_CENTROID_HDR + (size_t) dim * 4);
	const size_t cper = (PG_BLCKSZ - PG_PAGE_HEADER - IVF_CENTROID_HDR) / (csize + PG_ITEMID);
	const size_t esize = IVF_ENTRY_HDR + PG_MAXALIGN((size_t) dim * 4);
	const size_t eper = (PG_BLCKSZ - PG_PAGE_HEADER - 8) / (esize + PG_ITEMID);
	int64_t		n = 1;

	if (cper < 1 || eper < 1)
		return -1;
	n += (ncent + (int64_t) cper - 1) / (int64_t) cper;
	for (int L = 0; L < ncent; L++)
		n += (list_len[L] + (int64_t) eper - 1) / (int64_t) eper;
	return n;
}

extern "C" int64_t
ndbhip_ivf_pages_needed(int dim, int ncentroids, const int64_t *list_len)
{
	if (dim < 1 || ncentroids < 1 || !list_len)
		return -1;
	return ivf_blocks_needed(dim, ncentroids, list_len);
}

extern "C" int
ndbhip_ivf_pages_pack(int dim, int nlists, int nprobe, int ncentroids, const float *centroids,
					  const int64_t *list_len, const float *rows, const uint8_t *tids6, uint8_t *pages,
					  uint32_t nblocks_cap, uint32_t *nblocks_out)
{
	if (dim < 1 || dim > 32767 || ncentroids < 1 || !centroids || !list_len || !pages)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "bad arguments");
	const int64_t need = ivf_blocks_needed(dim, ncentroids, list_len);

	if (need < 0)
		return ndbhip_pages_fail(NDBHIP_ERR_UNSUPPORTED, "ivf: one entry does not fit an 8 KB page");
	if ((int64_t) nblocks_cap < need)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "page buffer too small");
	const size_t csize = PG_MAXALIGN(IVF_CENTROID_HDR + (size_t) dim * 4);
	const size_t esize = IVF_ENTRY_HDR + PG_MAXALIGN((size_t) dim * 4);
	uint32_t	nb = 1;
	int64_t		total = 0;

	for (int L = 0; L < ncentroids; L++)
		total += list_len[L];
	if (total > 0 && (!rows || !tids6))
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "rows/tids missing");

	/* centroid pages first (block 1..), then the list chains; firstBlock patched afterwards */
	std::vector<uint8_t *> cent_item((size_t) ncentroids);
	uint32_t	cblock0 = nb;
	uint8_t    *cp = nullptr;
	int			ncpages = 0;

	for (int c = 0; c < ncentroids; c++)
	{
		if (!cp || page_free(cp) < csize)
		{
			uint8_t    *np = pages + (size_t) nb * PG_BLCKSZ;

			page_init(np, IVF_CENTROID_HDR);
			if (cp)
				wr32(cp + rd16(cp + 16), nb);	/* version-2 chain pointer */
			cp = np;
			nb++;
			ncpages++;
		}
		uint8_t    *it = page_add(cp, csize);

		memset(it, 0, csize);
		wr32(it, (uint32_t) c);				/* listId */
		wr32(it + 4, (uint32_t) dim);
		wr64(it + 8, (uint64_t) list_len[c]);	/* memberCount */
		wr32(it + 16, NDBHIP_INVALID_BLOCK);	/* firstBlock */
		memcpy(it + IVF_CENTROID_HDR, centroids + (size_t) c * dim, (size_t) dim * 4);
		cent_item[c] = it;
	}
	int64_t		r = 0;

	for (int L = 0; L < ncentroids; L++)
	{
		uint8_t    *lp = nullptr;

		for (int64_t i = 0; i < list_len[L]; i++, r++)
		{
			if (!lp || page_free(lp) < esize)	/* ivfinsert: PageGetFreeSpace(listPage) < entrySize (:1072) */
			{
				uint8_t    *np = pages + (size_t) nb * PG_BLCKSZ;

				page_init(np, 8);
				wr32(np + rd16(np + 16), NDBHIP_INVALID_BLOCK);	/* nextBlock */
				if (lp)
					wr32(lp + rd16(lp + 16), nb);
				else
					wr32(cent_item[L] + 16, nb);
				lp = np;
				nb++;
			}
			uint8_t    *it = page_add(lp, esize);

			memset(it, 0, esize);
			memcpy(it, tids6 + (size_t) r * 6, 6);
			wr16(it + 6, (uint16_t) dim);
			memcpy(it + IVF_ENTRY_HDR, rows + (size_t) r * dim, (size_t) dim * 4);
			const uint32_t cnt = rd32(lp + rd16(lp + 16) + 4) + 1;	/* entryCount++ */

			wr32(lp + rd16(lp + 16) + 4, cnt);
		}
	}
	/* meta page */
	page_init(pages, 32);
	uint8_t    *meta = pages + PG_PAGE_HEADER;

	wr32(meta, IVF_MAGIC);
	wr32(meta + 4, ncpages > 1 ? 2u : 1u);
	wr32(meta + 8, (uint32_t) nlists);
	wr32(meta + 12, (uint32_t) nprobe);
	wr32(meta + 16, (uint32_t) dim);
	wr32(meta + 20, cblock0);
	wr64(meta + 24, (uint64_t) total);
	if (nblocks_out)
		*nblocks_out = nb;
	return NDBHIP_OK;
}

/* ambuild after a device build: mirror -> index pages */
extern "C" int
ndbhip_ivf_write_pages(const ndbhip_ivf *ix, int nprobe, uint8_t *pages, uint32_t nblocks_cap,
					   uint32_t *nblocks_out)
{
	const int	nc = ndbhip_ivf_ncentroids(ix);
	const int64_t n = ndbhip_ivf_nrows(ix);
	int			dim = 0, nlists = 0;

	if (nc < 1 || n < 0)
		return ndbhip_pages_fail(NDBHIP_ERR_STATE, "index not loaded");
	if (ndbhip_ivf_shape(ix, &dim, &nlists))
		return NDBHIP_ERR_STATE;
	std::vector<float> cent((size_t) nc * dim);
	std::vector<int64_t> ll((size_t) nc);
	std::vector<float> rows((size_t) (n > 0 ? n : 1) * dim);
	std::vector<uint8_t> tids((size_t) (n > 0 ? n : 1) * 6);
	int			rc = ndbhip_ivf_export(ix, cent.data(), ll.data(), rows.data(), tids.data());

	if (rc)
		return rc;
	return ndbhip_ivf_pages_pack(dim, nlists, nprobe, nc, cent.data(), ll.data(), rows.data(), tids.data(), pages,
								 nblocks_cap, nblocks_out);
}

/* ================================================================== */
/* hnsw relation pages                                                  */
/*   block 0: HnswMetaPageData at PageGetContents (src/index/hnsw_am.c:108-120, 1091-1110):              */
/*     magic 0x48534E57 @0, version @4, entryPoint @8, entryLevel @12, maxLevel @16, int16 m @20,        */
/*     efConstruction @22, efSearch @24, float4 ml @28, int64 insertedVectors @32 (40 B)                 */
/*   block b >= 1: ONE item (hnsw_am.c:2288-2332) = HnswNodeData{ItemPointerData heapPtr @0; int level   */
/*     @8; int16 dim @12; int16 neighborCount[16] @14} (48 B) + float4 vector[dim] @48 +                  */
/*     BlockNumber neighbors[level+1][2m] (:124-181); hnswbulkdelete leaves the item with LP_DEAD (:693) */
/* ================================================================== */

#define HNSW_MAGIC 0x48534E57u
#define HNSW_NODE_HDR 48
#define HNSW_LEVELS 16

struct HnswPagesInfo
{
	int			dim, m, ef_construction, ef_search, entry_level, max_level;
	uint32_t	entry_point, nblocks;
	int64_t		inserted;
};

/* count / validate (outputs NULL) or unpack into the dense arrays of ndbhip_hnsw_export's layout */
static int
hnsw_walk_pages(const uint8_t *pages, uint32_t nblocks, HnswPagesInfo *info, float *vecs, int32_t *levels,
				int16_t *ncount, uint32_t *nbrs, uint8_t *tids6, uint8_t *dead)
{
	if (!pages || nblocks < 1)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "no pages");
	const PageView mv = page_view(pages);

	if (!mv.ok)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "block 0 is not an initialised page");
	const uint8_t *meta = pages + PG_PAGE_HEADER;

	if (rd32(meta) != HNSW_MAGIC)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw: invalid magic number in metadata");
	const int	m = (int16_t) rd16(meta + 20);

	if (m < 2 || m > 128)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw: m in the meta page out of range");
	int			dim = 0;
	const size_t stride = (size_t) HNSW_LEVELS * 2 * m;

	for (uint32_t b = 1; b < nblocks; b++)
	{
		const PageView v = page_view(pages + (size_t) b * PG_BLCKSZ);
		uint32_t	off, flags, len;

		if (!v.ok || v.nitems < 1)
			return ndbhip_pages_fail(NDBHIP_ERR_UNSUPPORTED, "hnsw: empty node page (the reference never leaves one)");
		item_id(v, 0, off, flags, len);
		if ((flags != LP_NORMAL && flags != LP_DEAD) || off + HNSW_NODE_HDR > PG_BLCKSZ)
			return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw: bad line pointer on a node page");
		const uint8_t *node = v.p + off;
		const int	level = (int32_t) rd32(node + 8);
		const int	ndim = (int16_t) rd16(node + 12);

		if (level < 0 || level >= HNSW_LEVELS || ndim < 1)
			return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw: node level / dim out of range");
		if (dim == 0)
			dim = ndim;
		if (ndim != dim)
			return ndbhip_pages_fail(NDBHIP_ERR_UNSUPPORTED, "hnsw: nodes of different dimensions");
		const size_t need = HNSW_NODE_HDR + (size_t) ndim * 4 + (size_t) (level + 1) * 2 * m * 4;

		if (need > len || off + need > PG_BLCKSZ)
			return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw: node item shorter than its level needs");
		if (!vecs)
			continue;
		memcpy(tids6 + (size_t) b * 6, node, 6);
		levels[b] = level;
		memcpy(ncount + (size_t) b * HNSW_LEVELS, node + 14, HNSW_LEVELS * 2);
		memcpy(vecs + (size_t) b * dim, node + HNSW_NODE_HDR, (size_t) dim * 4);
		memcpy(nbrs + (size_t) b * stride, node + HNSW_NODE_HDR + (size_t) dim * 4, (size_t) (level + 1) * 2 * m * 4);
		dead[b] = flags == LP_DEAD;
	}
	if (info)
	{
		info->dim = dim;
		info->m = m;
		info->entry_point = rd32(meta + 8);
		info->entry_level = (int32_t) rd32(meta + 12);
		info->max_level = (int32_t) rd32(meta + 16);
		info->ef_construction = (int16_t) rd16(meta + 22);
		info->ef_search = (int16_t) rd16(meta + 24);
		info->nblocks = nblocks;
		memcpy(&info->inserted, meta + 32, 8);
	}
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_pages_info(const uint8_t *pages, uint32_t nblocks, int *dim, int *m, int *ef_construction,
					   int *ef_search, uint32_t *entry_point, int *entry_level)
{
	HnswPagesInfo info;
	int			rc = hnsw_walk_pages(pages, nblocks, &info, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);

	if (rc)
		return rc;
	if (dim) *dim = info.dim;
	if (m) *m = info.m;
	if (ef_construction) *ef_construction = info.ef_construction;
	if (ef_search) *ef_search = info.ef_search;
	if (entry_point) *entry_point = info.entry_point;
	if (entry_level) *entry_level = info.entry_level;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_pages_unpack(const uint8_t *pages, uint32_t nblocks, float *vecs, int32_t *levels, int16_t *ncount,
						 uint32_t *nbrs, uint8_t *tids6, uint8_t *dead)
{
	HnswPagesInfo info;
	int			rc = hnsw_walk_pages(pages, nblocks, &info, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);

	if (rc)
		return rc;
	if (!vecs || !levels || !ncount || !nbrs || !tids6 || !dead)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "NULL output");
	const size_t stride = (size_t) HNSW_LEVELS * 2 * info.m;

	memset(vecs, 0, (size_t) info.dim * 4);
	levels[0] = 0;
	memset(ncount, 0, HNSW_LEVELS * 2);
	memset(nbrs, 0xFF, (size_t) nblocks * stride * 4);
	memset(tids6, 0, 6);
	dead[0] = 0;
	return hnsw_walk_pages(pages, nblocks, nullptr, vecs, levels, ncount, nbrs, tids6, dead);
}

/* dense arrays -> relation image: block 0 meta + one node page per block */
extern "C" int
ndbhip_hnsw_pages_pack(int dim, int m, int ef_construction, int ef_search, uint32_t nblocks, const float *vecs,
					   const int32_t *levels, const int16_t *ncount, const uint32_t *nbrs, const uint8_t *tids6,
					   const uint8_t *dead, uint32_t entry_point, int entry_level, uint8_t *pages,
					   uint32_t nblocks_cap)
{
	if (dim < 1 || m < 2 || m > 128 || nblocks < 1 || !pages || (nblocks > 1 && (!vecs || !levels || !ncount || !nbrs || !tids6)))
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (nblocks_cap < nblocks)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "page buffer too small");
	const size_t stride = (size_t) HNSW_LEVELS * 2 * m;
	int			max_level = -1;
	int64_t		live = 0;

	for (uint32_t b = 1; b < nblocks; b++)
	{
		uint8_t    *p = pages + (size_t) b * PG_BLCKSZ;
		const int	level = levels[b];

		if (level < 0 || level >= HNSW_LEVELS)
			return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "hnsw: node level out of range");
		const size_t size = PG_MAXALIGN(HNSW_NODE_HDR + (size_t) dim * 4 + (size_t) (level + 1) * 2 * m * 4);

		page_init(p, 0);
		if (page_free(p) < size)	/* "node size exceeds free space" (:2306-2312) */
			return ndbhip_pages_fail(NDBHIP_ERR_UNSUPPORTED, "hnsw: node does not fit one page");
		uint8_t    *node = page_add(p, size);

		memset(node, 0, size);
		memcpy(node, tids6 + (size_t) b * 6, 6);
		wr32(node + 8, (uint32_t) level);
		wr16(node + 12, (uint16_t) dim);
		memcpy(node + 14, ncount + (size_t) b * HNSW_LEVELS, HNSW_LEVELS * 2);
		memcpy(node + HNSW_NODE_HDR, vecs + (size_t) b * dim, (size_t) dim * 4);
		memcpy(node + HNSW_NODE_HDR + (size_t) dim * 4, nbrs + (size_t) b * stride, (size_t) (level + 1) * 2 * m * 4);
		if (dead && dead[b])
		{
			const uint32_t w = rd32(p + PG_PAGE_HEADER);

			wr32(p + PG_PAGE_HEADER, (w & ~(3u << 15)) | ((uint32_t) LP_DEAD << 15));
		}
		else
			live++;
		if (level > max_level)
			max_level = level;
	}
	uint8_t    *p0 = pages;

	page_init(p0, 40);				/* PageInit(page, size, sizeof(HnswMetaPageData)): :1097 */
	uint8_t    *meta = p0 + PG_PAGE_HEADER;

	wr32(meta, HNSW_MAGIC);
	wr32(meta + 4, 1);
	wr32(meta + 8, entry_point);
	wr32(meta + 12, (uint32_t) entry_level);
	wr32(meta + 16, (uint32_t) max_level);
	wr16(meta + 20, (uint16_t) m);
	wr16(meta + 22, (uint16_t) ef_construction);
	wr16(meta + 24, (uint16_t) ef_search);
	{
		const float ml = 0.36f;		/* HNSW_DEFAULT_ML: :84 */

		memcpy(meta + 28, &ml, 4);
	}
	wr64(meta + 32, (uint64_t) live);
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_load_pages(ndbhip_hnsw **out, const uint8_t *pages, uint32_t nblocks)
{
	HnswPagesInfo info;
	int			rc = hnsw_walk_pages(pages, nblocks, &info, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);

	if (rc)
		return rc;
	if (!out)
		return ndbhip_pages_fail(NDBHIP_ERR_INVALID, "out is NULL");
	if (nblocks < 2 || info.dim < 1)
		return ndbhip_pages_fail(NDBHIP_ERR_STATE, "hnsw: the relation holds no node");
	const size_t stride = (size_t) HNSW_LEVELS * 2 * info.m;
	std::vector<float> vecs((size_t) nblocks * info.dim);
	std::vector<int32_t> levels(nblocks);
	std::vector<int16_t> ncount((size_t) nblocks * HNSW_LEVELS);
	std::vector<uint32_t> nbrs((size_t) nblocks * stride);
	std::vector<uint8_t> tids((size_t) nblocks * 6), dead(nblocks);

	rc = ndbhip_hnsw_pages_unpack(pages, nblocks, vecs.data(), levels.data(), ncount.data(), nbrs.data(),
								  tids.data(), dead.data());
	if (rc)
		return rc;
	/* ndbhip_hnsw_load takes the packed slots: (level+1)*2m per node */
	std::vector<int64_t> off((size_t) nblocks + 1);
	std::vector<uint32_t> packed;

	off[0] = 0;
	off[1] = 0;
	for (uint32_t b = 1; b < nblocks; b++)
	{
		const size_t n = (size_t) (levels[b] + 1) * 2 * info.m;

		packed.insert(packed.end(), nbrs.begin() + (size_t) b * stride, nbrs.begin() + (size_t) b * stride + n);
		off[b + 1] = (int64_t) packed.size();
	}
	ndbhip_hnsw *h = nullptr;

	rc = ndbhip_hnsw_create(info.dim, info.m, &h);
	if (rc)
		return rc;
	rc = ndbhip_hnsw_load(h, nblocks, vecs.data(), levels.data(), ncount.data(), off.data(),
						  packed.empty() ? nullptr : packed.data(), tids.data(), info.entry_point, info.entry_level);
	if (!rc)
		rc = ndbhip_hnsw_set_dead_flags(h, dead.data());
	if (!rc && info.ef_construction >= 4 && info.ef_construction <= NDBHIP_MAX_EF && info.ef_search >= 4 &&
		info.ef_search <= NDBHIP_MAX_EF)
		rc = ndbhip_hnsw_set_meta(h, info.ef_construction, info.ef_search);
	if (rc)
	{
		ndbhip_hnsw_destroy(h);
		return rc;
	}
	*out = h;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_hnsw_write_pages(const ndbhip_hnsw *h, int ef_construction, int ef_search, uint8_t *pages,
						uint32_t nblocks_cap, uint32_t *nblocks_out)
{
	uint32_t	nb = 0, entry = 0;
	int			entry_level = -1, dim = 0, m = 0;
	int			rc = ndbhip_hnsw_export(h, &nb, nullptr, nullptr, nullptr, &entry, &entry_level);

	if (rc)
		return rc;
	rc = ndbhip_hnsw_shape(h, &dim, &m);
	if (rc)
		return rc;
	if (nblocks_out)
		*nblocks_out = nb;
	if (!pages)
		return NDBHIP_OK;		/* size query */
	const size_t stride = (size_t) HNSW_LEVELS * 2 * m;
	std::vector<float> vecs((size_t) nb * dim);
	std::vector<int32_t> levels(nb);
	std::vector<int16_t> ncount((size_t) nb * HNSW_LEVELS);
	std::vector<uint32_t> nbrs((size_t) nb * stride);
	std::vector<uint8_t> tids((size_t) nb * 6), dead(nb);

	rc = ndbhip_hnsw_export(h, nullptr, levels.data(), ncount.data(), nbrs.data(), nullptr, nullptr);
	if (!rc)
		rc = ndbhip_hnsw_export_rows(h, vecs.data(), tids.data(), dead.data());
	if (rc)
		return rc;
	return ndbhip_hnsw_pages_pack(dim, m, ef_construction, ef_search, nb, vecs.data(), levels.data(), ncount.data(),
								  nbrs.data(), tids.data(), dead.data(), entry, entry_level, pages, nblocks_cap);
}
