/*
 * ndbhip_gen.h — the counter-based synthetic-data generator of the bench and the full-size tests (SURVEY 8d:
 * "a counter-based generator implemented in the repo ... so any slice can be regenerated on the GPU box without
 * shipping data"), written once for host C++ and device code.
 *
 * Element (row r, dimension d) of a data set is a pure function of (seed, r, d): no state, any slice in any
 * order, and the SAME BITS on the host (ndbhip_gen_rows_host: what an oracle run is fed) and on the device
 * (ndbhip_gen_rows_device: what the mirror is built from).  Bit equality across gcc / x86-64 and hipcc / gfx950
 * is by construction: integer hashing (splitmix64), then only IEEE-754 double +, -, *, / and sqrt — all correctly
 * rounded on both sides — with contraction off; the logarithm is a fixed atanh series instead of libm / ocml.
 *
 *   z(seed, i)  ~ N(0, 1): Marsaglia's polar method on the draws of counter i (u1, u2 uniform in (-1, 1), accept
 *                 0 < s = u1^2 + u2^2 < 1, z = u1 sqrt(-2 ln s / s)), cast to float
 *   gauss       x[r][d] = z(seed, r * dim + d)
 *   clustered   x[r][d] = c[comp(r)][d] + sigma * z(seed, r * dim + d) in float (product and sum rounded
 *               separately), c[j][d] = z(center_seed, j * dim + d), comp(r) = splitmix64(seed ^ COMP ^ r) mod components
 */
#ifndef NDBHIP_GEN_H
#define NDBHIP_GEN_H

#include <stdint.h>
#include <string.h>
#include "ndbhip_common.h"

#pragma clang fp contract(off)

NDB_HD static inline uint64_t
ndb_splitmix64(uint64_t x)
{
	uint64_t	z = x + 0x9E3779B97F4A7C15ull;

	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

NDB_HD static inline double
ndb_gen_u2d(uint64_t u)
{
	double		d;

#if defined(__HIP_DEVICE_COMPILE__)
	d = __longlong_as_double((long long) u);
#else
	memcpy(&d, &u, 8);
#endif
	return d;
}

NDB_HD static inline uint64_t
ndb_gen_d2u(double d)
{
	uint64_t	u;

#if defined(__HIP_DEVICE_COMPILE__)
	u = (uint64_t) __double_as_longlong(d);
#else
	memcpy(&u, &d, 8);
#endif
	return u;
}

/* ln x for a normal x in (0, 1]: x = m 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m - 1) / (m + 1)) by
 * its series to s^21 (|s| <= 0.1716: the first omitted term is < 1e-18); +, -, *, / only */
NDB_HD static inline double
ndb_gen_log(double x)
{
	const uint64_t b = ndb_gen_d2u(x);
	int			e = (int) ((b >> 52) & 0x7FFu) - 1022;
	double		m = ndb_gen_u2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FE0000000000000ull);	/* [0.5, 1) */

	if (m < 0.70710678118654752440)
	{
		m = m * 2.0;
		e -= 1;
	}
	const double s = (m - 1.0) / (m + 1.0);
	const double s2 = s * s;
	double		p = 1.0 / 21.0;

	p = p * s2 + 1.0 / 19.0;
	p = p * s2 + 1.0 / 17.0;
	p = p * s2 + 1.0 / 15.0;
	p = p * s2 + 1.0 / 13.0;
	p = p * s2 + 1.0 / 11.0;
	p = p * s2 + 1.0 / 9.0;
	p = p * s2 + 1.0 / 7.0;
	p = p * s2 + 1.0 / 5.0;
	p = p * s2 + 1.0 / 3.0;
	p = p * s2 + 1.0;
	return (double) e * 0.69314718055994530942 + 2.0 * s * p;
}

NDB_HD static inline double
ndb_gen_sqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_sqrt(x);	/* ocml: correctly rounded */
#else
	return __builtin_sqrt(x);
#endif
}

/* uniform in (-1, 1): 53 bits, never an end point */
NDB_HD static inline double
ndb_gen_uniform(uint64_t bits)
{
	return ((double) (bits >> 11) + 0.5) * (1.0 / 4503599627370496.0) - 1.0;
}

NDB_HD static inline float
ndb_gen_normal(uint64_t seed, uint64_t i)
{
	const uint64_t h0 = ndb_splitmix64(seed ^ (i * 0xD1342543DE82EF95ull));

	for (uint64_t t = 0;; t += 2)
	{
		const double u1 = ndb_gen_uniform(ndb_splitmix64(h0 + t * 0x9E3779B97F4A7C15ull));
		const double u2 = ndb_gen_uniform(ndb_splitmix64(h0 + (t + 1) * 0x9E3779B97F4A7C15ull));
		const double s = u1 * u1 + u2 * u2;

		if (s > 0.0 && s < 1.0)
			return (float) (u1 * ndb_gen_sqrt(-2.0 * ndb_gen_log(s) / s));
	}
}

#define NDB_GEN_COMP 0x636F6D70636F6D70ull

/* kind 0 = i.i.d. N(0, 1), 1 = mixture of `components` Gaussians (sigma) around centers ~ N(0, 1) */
NDB_HD static inline float
ndb_gen_element(int kind, uint64_t seed, uint64_t center_seed, uint64_t row, int d, int dim, int components, float sigma)
{
	const float z = ndb_gen_normal(seed, row * (uint64_t) dim + (uint64_t) d);

	if (kind == 0)
		return z;
	const uint64_t comp = ndb_splitmix64(seed ^ NDB_GEN_COMP ^ row) % (uint64_t) components;
	const float c = ndb_gen_normal(center_seed, comp * (uint64_t) dim + (uint64_t) d);
	const float sz = sigma * z;

	return c + sz;
}

#endif							/* NDBHIP_GEN_H */
