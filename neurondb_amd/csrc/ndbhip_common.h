/*
 * ndbhip_common.h — helpers shared by host and device code.
 *
 * The reference picks its top-k with a selection sort that swaps entries of an
 * index array (src/index/ivf_am.c:1856-1881, src/index/hnsw_am.c:1977-2004).
 * With ties that is NOT "smallest (distance, position) first": an element
 * sitting in one of the first k slots is moved to the winner's slot when it
 * loses, and may then be met later than an equal-distance element that
 * originally followed it.  replay_selection() reproduces that exactly from a
 * sparse subset of the candidate array (see DESIGN.md "Top-k replay"):
 *
 *   S = { all candidates with dist < T } ∪ { the first 2k candidates, by
 *         position, with dist == T },   T = k-th smallest distance.
 *
 * Elements with dist > T are never selected and only ever act as "the loser
 * parked in slot i", which any remaining element of S beats; an element of the
 * tie class that is selected has fewer than 2k tie-class elements before it
 * (at most k-1 selected earlier, at most k displaced past it).  So |S| <= 3k
 * is enough whatever the input.
 */
#ifndef NDBHIP_COMMON_H
#define NDBHIP_COMMON_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NDB_HD __host__ __device__
#else
#define NDB_HD
#endif


NDB_HD static inline uint32_t
ndb_f2u(float f)
{
	uint32_t	u;

#if defined(__HIP_DEVICE_COMPILE__)
	u = __float_as_uint(f);
#else
	memcpy(&u, &f, 4);
#endif
	return u;
}

NDB_HD static inline float
ndb_u2f(uint32_t u)
{
	float		f;

#if defined(__HIP_DEVICE_COMPILE__)
	f = __uint_as_float(u);
#else
	memcpy(&f, &u, 4);
#endif
	return f;
}

/*
 * Order-preserving map float -> uint32 under the C `<` operator for non-NaN
 * values: a < b  <=>  key(a) < key(b), and -0.0f == +0.0f map to one key
 * (C compares them equal, so the reference treats them as a tie).
 * NaN never compares less; we place (positive) NaNs above +inf.
 */
NDB_HD static inline uint32_t
ndb_key_from_bits(uint32_t u)
{
	if (u == 0x80000000u)
		u = 0;
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

/* pack / unpack ItemPointerData images */
NDB_HD static inline uint64_t
ndb_tid_pack(const uint8_t *t6)
{
	return (uint64_t) t6[0] | ((uint64_t) t6[1] << 8) | ((uint64_t) t6[2] << 16) |
		((uint64_t) t6[3] << 24) | ((uint64_t) t6[4] << 32) | ((uint64_t) t6[5] << 40);
}

NDB_HD static inline void
ndb_tid_unpack(uint64_t v, uint8_t *t6)
{
	t6[0] = (uint8_t) v;
	t6[1] = (uint8_t) (v >> 8);
	t6[2] = (uint8_t) (v >> 16);
	t6[3] = (uint8_t) (v >> 24);
	t6[4] = (uint8_t) (v >> 32);
	t6[5] = (uint8_t) (v >> 40);
}

/*
 * Serial replay (host; also the specification of the wave-parallel device
 * version in ndbhip.hip).  key[]/pos[] describe n entries of S (any order);
 * pos[] is overwritten.  Writes the indices (into the entry arrays) of the
 * selected entries to order[0..kk).  kk = min(k, total) with total = size of
 * the full candidate array.
 */
static inline int
ndb_replay_selection_host(const uint32_t *key, uint32_t *pos, uint8_t *taken, int n,
						  int k, int64_t total, int *order)
{
	int			kk = (int) ((int64_t) k < total ? (int64_t) k : total);
	int			i,
				e;

	if (kk > n)
		kk = n;
	for (e = 0; e < n; e++)
		taken[e] = 0;
	for (i = 0; i < kk; i++)
	{
		int			best = -1;
		uint64_t	bestc = ~(uint64_t) 0;

		for (e = 0; e < n; e++)
		{
			uint64_t	c;

			if (taken[e])
				continue;
			c = ((uint64_t) key[e] << 32) | pos[e];
			if (best < 0 || c < bestc)
			{
				best = e;
				bestc = c;
			}
		}
		if (best < 0)
			break;
		taken[best] = 1;
		/* the loser parked in slot i moves to the winner's slot */
		for (e = 0; e < n; e++)
			if (!taken[e] && pos[e] == (uint32_t) i)
				pos[e] = pos[best];
		order[i] = best;
	}
	return i;
}

#endif							/* NDBHIP_COMMON_H */
