/*
 * ndbhip_kernels.h — device code of the MI355X (gfx950, wave64) distance engine.
 *
 * Design (DESIGN.md has the long form):
 *
 *  * Exact arithmetic.  The reference scores a candidate with a strictly
 *    sequential loop (`sum += diff*diff`, src/index/ivf_am.c:1562-1568; fp64
 *    accumulate in src/index/hnsw_am.c:1312-1337).  A lane-parallel reduction
 *    rounds differently and can flip neighbour ranks, so here ONE LANE OWNS ONE
 *    ROW and walks its dimensions in order with the same operand types and the
 *    same unfused multiply/add — distances come out bit-identical, not merely
 *    the ids.  Parallelism is across rows (64 rows per wave), never inside one.
 *
 *  * HBM-bound by construction.  A lane reading its own row would touch a new
 *    3 KB-strided line per lane.  Instead a wave stages a 64-row x 256-byte
 *    chunk through a wave-private 16 KiB LDS tile: 16 lanes fetch one row's
 *    256 contiguous bytes (4 rows = 1 KiB per load instruction, fully
 *    coalesced), the tile is written with a 16-byte-slot XOR swizzle
 *    (slot ^= row & 15) and each lane then reads its row back with
 *    conflict-free ds_read_b128.  Waves never synchronise with each other
 *    (no s_barrier): each is an independent streaming engine, latency is
 *    hidden by 8+ waves per CU.
 *
 *  * The query is wave-uniform: it is read through the scalar cache into SGPRs
 *    and used as the scalar operand of v_sub/v_mul — no LDS traffic for it.
 */
#ifndef NDBHIP_KERNELS_H
#define NDBHIP_KERNELS_H

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "ndbhip_common.h"

#pragma clang fp contract(off)

#define NDB_WAVE 64
#define NDB_CHUNK 64						/* floats of each row staged per step (256 B) */
#define NDB_TILE_FLOATS (NDB_WAVE * NDB_CHUNK)	/* 16 KiB per wave */

/* rounding recipes */
enum
{
	R_IVF_L2 = 0,				/* ivf_am.c:1561-1568 */
	R_IVF_COS = 1,				/* ivf_am.c:1570-1581 */
	R_IVF_IP = 2,				/* new (quirk Q2): -dot, fp32 sequential */
	R_IVF_L2SQ = 3,				/* ivf_am.c:2255-2269 (k-means) */
	R_SCR_L2 = 32,				/* grouped scan only: fused dot + norms, a bound on L2 (screening, ndbhip.hip) */
	R_HNSW_L2 = 4,				/* hnsw_am.c:1312-1319 */
	R_HNSW_COS = 5,				/* hnsw_am.c:1321-1332 */
	R_HNSW_IP = 6,				/* hnsw_am.c:1334-1337 */
	R_COUNT = 7
};

template <int R> struct Acc;

template <> struct Acc<R_IVF_L2>
{
	float		s = 0.0f;
	__device__ __forceinline__ void step(float q, float x)
	{
		float		d = q - x;

		s = s + d * d;
	}
	__device__ __forceinline__ float fin() const { return __builtin_sqrtf(s); }
};

template <> struct Acc<R_IVF_L2SQ>
{
	float		s = 0.0f;
	__device__ __forceinline__ void step(float q, float x)
	{
		float		d = q - x;

		s = s + d * d;
	}
	__device__ __forceinline__ float fin() const { return s; }
};

template <> struct Acc<R_IVF_COS>
{
	float		dot = 0.0f, n1 = 0.0f, n2 = 0.0f;
	__device__ __forceinline__ void step(float q, float x)
	{
		dot = dot + q * x;
		n1 = n1 + q * q;
		n2 = n2 + x * x;
	}
	__device__ __forceinline__ float fin() const
	{
		float		a = __builtin_sqrtf(n1);
		float		b = __builtin_sqrtf(n2);

		if (a == 0.0f || b == 0.0f)
			return 1.0f;
		return 1.0f - (dot / (a * b));
	}
};

template <> struct Acc<R_IVF_IP>
{
	float		dot = 0.0f;
	__device__ __forceinline__ void step(float q, float x) { dot = dot + q * x; }
	__device__ __forceinline__ float fin() const { return -dot; }
};

template <> struct Acc<R_HNSW_L2>
{
	double		s = 0.0;
	__device__ __forceinline__ void step(float q, float x)
	{
		double		d = (double) (q - x);	/* fp32 subtract, then widened */

		s = s + d * d;
	}
	__device__ __forceinline__ float fin() const { return (float) __builtin_sqrt(s); }
};

template <> struct Acc<R_HNSW_COS>
{
	double		dot = 0.0, n1 = 0.0, n2 = 0.0;
	__device__ __forceinline__ void step(float q, float x)
	{
		dot = dot + (double) (q * x);	/* fp32 products, widened */
		n1 = n1 + (double) (q * q);
		n2 = n2 + (double) (x * x);
	}
	__device__ __forceinline__ float fin() const
	{
		double		a = __builtin_sqrt(n1);
		double		b = __builtin_sqrt(n2);

		if (a == 0.0 || b == 0.0)
			return 2.0f;
		return (float) (1.0 - (dot / (a * b)));
	}
};

template <> struct Acc<R_HNSW_IP>
{
	double		dot = 0.0;
	__device__ __forceinline__ void step(float q, float x) { dot = dot + (double) (q * x); }
	__device__ __forceinline__ float fin() const { return (float) (-dot); }
};

/* Order LDS writes before the same wave's later LDS reads (different lanes).
 * LDS instructions of one wave execute in issue order; this only pins the
 * compiler's ordering. No instruction is emitted. */
__device__ __forceinline__ void
wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/*
 * Score 64 rows (one per lane) against the wave-uniform query q.
 *   base   row-major [* x dim] fp32, 16-byte aligned, dim % 4 == 0
 *   row    this lane's row number (any valid row for idle lanes)
 *   tile   this wave's private 16 KiB LDS tile
 * Returns this lane's distance, bit-identical to the CPU recipe R.
 */
/* Stage one 64-row x 64-float chunk through the wave's LDS tile and hand every
 * lane ITS row's 64 floats back in x[0..15] (x[p] = dims c+4p .. c+4p+3).
 * FULL = the chunk lies completely inside the row. */
template <bool FULL>
__device__ __forceinline__ void
stage_chunk(float4 (&x)[16], const float *__restrict__ base, const uint32_t (&rows16)[16],
			int dim, int c, float *tile, int lane, int grp, int slot)
{
#pragma unroll
	for (int i = 0; i < 16; i++)
	{
		const int	r = 4 * i + grp;
		const int	piece = slot ^ (r & 15);
		const int	col = c + piece * 4;
		const float *src = base + (size_t) rows16[i] * (size_t) dim + col;

		if (FULL || col < dim)
			x[i] = *reinterpret_cast<const float4 *>(src);
		else
			x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	}
#pragma unroll
	for (int i = 0; i < 16; i++)
		*reinterpret_cast<float4 *>(tile + (4 * i + grp) * NDB_CHUNK + slot * 4) = x[i];
	wave_lds_sync();
#pragma unroll
	for (int p = 0; p < 16; p++)
		x[p] = *reinterpret_cast<const float4 *>(tile + lane * NDB_CHUNK + ((p ^ (lane & 15)) * 4));
	wave_lds_sync();
}

/*
 * Generic-width staging (CH = 64 or 32 floats of every row per step).  CH = 32 halves the LDS tile
 * (8 KiB per wave) and the row registers, so more waves fit per SIMD — the grouped kernels, which are
 * bound by the vector ALU and by scalar-load waits rather than by HBM, use it.
 *   lanes per row PP = CH/4, rows per load instruction 64/PP, load instructions per chunk PP.
 *   tile is linear in (row, slot); piece p of row r sits in slot p ^ swz(r):
 *     CH = 64: swz = r & 15          (row stride 256 B = one bank row)
 *     CH = 32: swz = (r >> 1) & 7    (two rows per 256-B bank row: the row parity picks the half)
 *   Either way the 16 lanes of every ds_read_b128 lane group land on 16 distinct 16-B slots of the
 *   bank row, and every ds_write_b128 8-lane group writes 128 contiguous bytes: conflict-free.
 */
template <int CH>
__device__ __forceinline__ int
tile_swz(int r)
{
	/* CH = 16: four rows per 256-B bank row, the row's position in it picks the quarter */
	return CH == 64 ? (r & 15) : (CH == 32 ? ((r >> 1) & 7) : ((r >> 2) & 3));
}

template <int CH>
__device__ __forceinline__ void
stage_chunk_w(float4 (&x)[CH / 4], const float *__restrict__ base, const uint32_t (&rowsN)[CH / 4],
			  int dim, int c, float *tile, int lane)
{
	constexpr int PP = CH / 4;			/* pieces (= lanes) per row chunk, load instructions per chunk */
	constexpr int RPI = 64 / PP;		/* rows per load instruction */
	const int	grp = lane / PP;
	const int	slot = lane % PP;

#pragma unroll
	for (int i = 0; i < PP; i++)
	{
		const int	r = RPI * i + grp;
		const int	piece = slot ^ tile_swz<CH>(r);

		x[i] = *reinterpret_cast<const float4 *>(base + (size_t) rowsN[i] * (size_t) dim + c + piece * 4);
	}
#pragma unroll
	for (int i = 0; i < PP; i++)
		*reinterpret_cast<float4 *>(tile + (RPI * i + grp) * CH + slot * 4) = x[i];
	wave_lds_sync();
#pragma unroll
	for (int p = 0; p < PP; p++)
		x[p] = *reinterpret_cast<const float4 *>(tile + lane * CH + ((p ^ tile_swz<CH>(lane)) * 4));
	wave_lds_sync();
}

/* rowsN[i] = row handled by this lane's i-th load instruction */
template <int CH>
__device__ __forceinline__ void
rows_for_loads(uint32_t (&rowsN)[CH / 4], uint32_t row, int lane)
{
	constexpr int PP = CH / 4;
	constexpr int RPI = 64 / PP;

#pragma unroll
	for (int i = 0; i < PP; i++)
		rowsN[i] = __shfl(row, RPI * i + lane / PP, 64);
}

/*
 * fp16 rows (halfvec columns: the indexed values ARE half precision, so keeping them as fp16 in HBM
 * is lossless and halves bytes and capacity per row).  Decode = the reference's fp16_to_float
 * (src/types/quantization.c:170-218): normals, zeros and infinities are IEEE; SUBNORMALS come out
 * 2^-10 too small in the reference (quirk Q20: exponent = 127-15-(10-exp)), reproduced here by an
 * exact power-of-two scaling of the hardware conversion.
 */
__device__ __forceinline__ float
h2f_ref(uint32_t h16)
{
	const float f = __half2float(__ushort_as_half((unsigned short) h16));

	/* (the subnormals are the values below 2^-14; a zero scaled is the same zero: one compare instead of two mask tests) */
	return __builtin_fabsf(f) < 0x1p-14f ? f * 0x1p-10f : f;
}

/* 8 halves (one 16-byte piece) -> 8 floats, element order preserved.  SUBFIX = false: the mirror is known
 * to hold no fp16 subnormal (checked once at load), so the hardware conversion alone IS fp16_to_float */
template <bool SUBFIX = true>
__device__ __forceinline__ void
decode8(const float4 &raw, float (&out)[8])
{
	const uint32_t w[4] = {__float_as_uint(raw.x), __float_as_uint(raw.y), __float_as_uint(raw.z),
		__float_as_uint(raw.w)};

#pragma unroll
	for (int i = 0; i < 4; i++)
	{
		if (SUBFIX)
		{
			out[2 * i] = h2f_ref(w[i] & 0xFFFFu);
			out[2 * i + 1] = h2f_ref(w[i] >> 16);
		}
		else
		{
			out[2 * i] = __half2float(__ushort_as_half((unsigned short) (w[i] & 0xFFFFu)));
			out[2 * i + 1] = __half2float(__ushort_as_half((unsigned short) (w[i] >> 16)));
		}
	}
}

/* 64 fp16 rows x 64 dimensions per step: the 128 raw bytes of every row pass through the 8 KiB
 * tile exactly like a 32-float chunk; rows are addressed in units of floats (dim/2 per row). */
template <int R>
__device__ __forceinline__ float
score_rows_f16(const float *__restrict__ q, const float *__restrict__ base16, uint32_t row, int dim,
			   float *tile)
{
	const int	lane = threadIdx.x & (NDB_WAVE - 1);
	uint32_t	rowsN[8];
	Acc<R>		acc;

	rows_for_loads<32>(rowsN, row, lane);
	for (int c = 0; c < dim; c += 64)
	{
		float4		raw[8];

		stage_chunk_w<32>(raw, base16, rowsN, dim >> 1, c >> 1, tile, lane);
#pragma unroll
		for (int p = 0; p < 8; p++)
		{
			float		x[8];
			const float4 q0 = *reinterpret_cast<const float4 *>(q + c + p * 8);
			const float4 q1 = *reinterpret_cast<const float4 *>(q + c + p * 8 + 4);

			decode8(raw[p], x);
			acc.step(q0.x, x[0]);
			acc.step(q0.y, x[1]);
			acc.step(q0.z, x[2]);
			acc.step(q0.w, x[3]);
			acc.step(q1.x, x[4]);
			acc.step(q1.y, x[5]);
			acc.step(q1.z, x[6]);
			acc.step(q1.w, x[7]);
		}
	}
	return acc.fin();
}

/* One 64-row x 64-float step: stage through LDS, then every lane walks its row.
 * FULL = the chunk lies completely inside the row (no per-piece bounds checks). */
template <int R, bool FULL>
__device__ __forceinline__ void
score_chunk(Acc<R> &acc, const float *__restrict__ q, const float *__restrict__ base,
			const uint32_t (&rows16)[16], int dim, int c, float *tile, int lane, int grp, int slot)
{
	float4		v[16];

	/* global -> registers: instruction i covers rows 4i..4i+3, 256 B each */
#pragma unroll
	for (int i = 0; i < 16; i++)
	{
		const int	r = 4 * i + grp;
		const int	piece = slot ^ (r & 15);
		const int	col = c + piece * 4;
		const float *src = base + (size_t) rows16[i] * (size_t) dim + col;

		if (FULL || col < dim)
			v[i] = *reinterpret_cast<const float4 *>(src);
		else
			v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
	}
	/* registers -> LDS, linear in (row, slot): piece p of row r sits in slot p ^ (r & 15) */
#pragma unroll
	for (int i = 0; i < 16; i++)
	{
		const int	r = 4 * i + grp;

		*reinterpret_cast<float4 *>(tile + r * NDB_CHUNK + slot * 4) = v[i];
	}
	wave_lds_sync();

	/* each lane walks ITS row's chunk in dimension order */
	const int	npieces = FULL ? 16 : ((dim - c) >> 2);
#pragma unroll
	for (int p = 0; p < 16; p++)
	{
		if (FULL || p < npieces)
		{
			const float4 x = *reinterpret_cast<const float4 *>(
				tile + lane * NDB_CHUNK + ((p ^ (lane & 15)) * 4));
			const float4 qq = *reinterpret_cast<const float4 *>(q + c + p * 4);

			acc.step(qq.x, x.x);
			acc.step(qq.y, x.y);
			acc.step(qq.z, x.z);
			acc.step(qq.w, x.w);
		}
	}
	wave_lds_sync();
}

template <int R>
__device__ __forceinline__ float
score_rows_tiled(const float *__restrict__ q, const float *__restrict__ base,
				 uint32_t row, int dim, float *tile)
{
	const int	lane = threadIdx.x & (NDB_WAVE - 1);
	const int	grp = lane >> 4;	/* which of the 4 rows of a load instruction */
	const int	slot = lane & 15;	/* 16-byte slot inside the 256-byte row chunk */
	uint32_t	rows16[16];
	Acc<R>		acc;
	int			c = 0;

#pragma unroll
	for (int i = 0; i < 16; i++)
		rows16[i] = __shfl(row, 4 * i + grp, NDB_WAVE);

	for (; c + NDB_CHUNK <= dim; c += NDB_CHUNK)
		score_chunk<R, true>(acc, q, base, rows16, dim, c, tile, lane, grp, slot);
	if (c < dim)
		score_chunk<R, false>(acc, q, base, rows16, dim, c, tile, lane, grp, slot);
	return acc.fin();
}

/* Any dimension, no alignment requirement: the lane reads its row directly.
 * Used only when dim % 4 != 0 (tiny test shapes). */
template <int R>
__device__ __forceinline__ float
score_row_direct(const float *__restrict__ q, const float *__restrict__ rowp, int dim)
{
	Acc<R>		acc;

	for (int i = 0; i < dim; i++)
		acc.step(q[i], rowp[i]);
	return acc.fin();
}

template <int R>
__device__ __forceinline__ float
score_rows(const float *__restrict__ q, const float *__restrict__ base, uint32_t row, int dim,
		   float *tile)
{
	if ((dim & 3) == 0)
		return score_rows_tiled<R>(q, base, row, dim, tile);
	return score_row_direct<R>(q, base + (size_t) row * (size_t) dim, dim);
}

#endif							/* NDBHIP_KERNELS_H */
