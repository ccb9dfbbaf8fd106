/*
 * ndbhip_ops.h — the SQL-operator kernels (<->, <=>, <#> without an index: src/vector/vector_distance*.c,
 * src/types/quantization.c:1985-2116) and ndbhip_batch_distance (part of ndbhip.hip's translation unit).
 */
#ifndef NDBHIP_OPS_H
#define NDBHIP_OPS_H

/* ================================================================== */
/* operator kernels (<->, <=>, <#> without an index): one lane = one    */
/* (query, row) pair in the reference's own order and width.            */
/*   OP_SCALAR  src/vector/vector_distance.c:93-122 (Kahan, double),    */
/*              145-157, 180-213 — what a default x86-64 build runs     */
/*              (Q16)                                                   */
/*   OP_SIMD    src/vector/vector_distance_simd.c:158-392: LANES fp32   */
/*              accumulators (8 = AVX2, 16 = AVX-512), cosine with FMA, */
/*              the fixed horizontal-sum tree (:84-137), scalar tail    */
/*   halfvec    src/types/quantization.c:1985-2116: both operands       */
/*              decoded per element (fp16_to_float incl. Q20), double   */
/* ================================================================== */

template <int LANES>
__device__ __forceinline__ float
op_hsum(const float (&v)[16])
{
	float		s8[8], s4[4];

#pragma unroll
	for (int j = 0; j < 8; j++)
		s8[j] = LANES == 16 ? v[j] + v[j + 8] : v[j];
#pragma unroll
	for (int j = 0; j < 4; j++)
		s4[j] = s8[j] + s8[j + 4];
	const float t0 = s4[0] + s4[1];
	const float t2 = s4[2] + s4[3];

	return t0 + t2;
}

/* strategy 1 L2, 2 cosine, 3 inner product (the dispatcher's sign: +sum from the SIMD paths, Q15) */
template <int LANES>
__device__ float
op_simd_pair(const float *__restrict__ a, const float *__restrict__ b, int dim, int strategy)
{
	float		acc0[16], acc1[16], acc2[16];
	const int	simd_end = (dim / LANES) * LANES;

#pragma unroll
	for (int j = 0; j < 16; j++)
		acc0[j] = acc1[j] = acc2[j] = 0.0f;
	for (int i = 0; i < simd_end; i += LANES)
	{
#pragma unroll
		for (int j = 0; j < LANES; j++)
		{
			const float va = a[i + j], vb = b[i + j];

			if (strategy == 1)
			{
				const float diff = va - vb;
				const float sq = diff * diff;

				acc0[j] = acc0[j] + sq;
			}
			else if (strategy == 3)
			{
				const float prod = va * vb;

				acc0[j] = acc0[j] + prod;
			}
			else
			{
				acc0[j] = __builtin_fmaf(va, vb, acc0[j]);	/* _mm256_fmadd_ps */
				acc1[j] = __builtin_fmaf(va, va, acc1[j]);
				acc2[j] = __builtin_fmaf(vb, vb, acc2[j]);
			}
		}
	}
	float		s0 = op_hsum<LANES>(acc0);

	if (strategy == 1)
	{
		for (int i = simd_end; i < dim; i++)
		{
			const float diff = a[i] - b[i];

			s0 = s0 + diff * diff;
		}
		return __builtin_sqrtf(s0);
	}
	if (strategy == 3)
	{
		for (int i = simd_end; i < dim; i++)
			s0 = s0 + a[i] * b[i];
		return s0;
	}
	float		na = op_hsum<LANES>(acc1), nb = op_hsum<LANES>(acc2);

	for (int i = simd_end; i < dim; i++)
	{
		const float va = a[i], vb = b[i];

		s0 = s0 + va * vb;
		na = na + va * va;
		nb = nb + vb * vb;
	}
	if (na == 0.0f || nb == 0.0f)
		return 1.0f;
	return 1.0f - (s0 / (__builtin_sqrtf(na) * __builtin_sqrtf(nb)));
}

/* H16: operands are fp16 images (halfvec), else float4; scalar double paths */
template <bool H16>
__device__ float
op_scalar_pair(const void *__restrict__ pa, const void *__restrict__ pb, int dim, int strategy)
{
	auto		ld = [&](const void *p, int i) -> double {
		if (H16)
			return (double) h2f_ref(((const uint16_t *) p)[i]);
		return (double) ((const float *) p)[i];
	};

	if (strategy == 1)
	{
		double		c = 0.0, sum = 0.0;

		for (int i = 0; i < dim; i++)
		{
			const double diff = ld(pa, i) - ld(pb, i);

			if (H16)
				sum = sum + diff * diff;	/* quantization.c:1997-2004: plain double sum */
			else
			{
				const double y = (diff * diff) - c;	/* Kahan: vector_distance.c:104-115 */
				const double t = sum + y;

				c = (t - sum) - y;
				sum = t;
			}
		}
		return (float) __builtin_sqrt(sum);
	}
	if (strategy == 3)
	{
		double		sum = 0.0;

		for (int i = 0; i < dim; i++)
			sum = sum + ld(pa, i) * ld(pb, i);
		/* halfvec_inner_product returns -sum (:2114); the float4 dispatcher negates the scalar kernel's -sum
		 * back to +sum (vector_distance_simd.c:511-558, Q15) */
		return H16 ? (float) (-sum) : -((float) (-sum));
	}
	double		dot = 0.0, na = 0.0, nb = 0.0;

	for (int i = 0; i < dim; i++)
	{
		const double va = ld(pa, i), vb = ld(pb, i);

		dot = dot + va * vb;
		na = na + va * va;
		nb = nb + vb * vb;
	}
	if (na == 0.0 || nb == 0.0)
		return 1.0f;
	return (float) (1.0 - (dot / (__builtin_sqrt(na) * __builtin_sqrt(nb))));
}

/* MODE 0 scalar float4, 8 / 16 SIMD emulation (falls back to scalar below LANES dims, as the dispatchers
 * do), 1 halfvec */
template <int MODE>
__global__ __launch_bounds__(256) void
k_op_distance(const void *__restrict__ queries, const void *__restrict__ vectors, float *__restrict__ out,
			  uint32_t nv, int dim, int strategy)
{
	const uint32_t v = blockIdx.x * 256 + threadIdx.x;
	const uint32_t q = blockIdx.y;
	constexpr size_t esz = MODE == 1 ? 2 : 4;

	if (v >= nv)
		return;
	const char *a = (const char *) queries + (size_t) q * dim * esz;
	const char *b = (const char *) vectors + (size_t) v * dim * esz;
	float		r;

	if (MODE == 1)
		r = op_scalar_pair<true>(a, b, dim, strategy);
	else if (MODE == 8 && dim >= 8)
		r = op_simd_pair<8>((const float *) a, (const float *) b, dim, strategy);
	else if (MODE == 16 && dim >= 16)
		r = op_simd_pair<16>((const float *) a, (const float *) b, dim, strategy);
	else
		r = op_scalar_pair<false>(a, b, dim, strategy);
	out[(size_t) q * nv + v] = r;
}

/* ================================================================== */
/* batch distance                                                      */
/* ================================================================== */

extern "C" int
ndbhip_batch_distance(const float *queries, const float *vectors, float *results, int nq, int nv, int dim,
					  int strategy, int recipe)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (nq < 0 || nv < 0 || dim < 1 || dim > 32767)
		return fail(NDBHIP_ERR_INVALID, "bad sizes");
	if (nq == 0 || nv == 0)
		return NDBHIP_OK;
	if (!queries || !vectors || !results)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	int			R = 0;

	if (recipe == 0)
		R = (strategy == 4) ? R_IVF_L2SQ : ivf_recipe(strategy);
	else if (recipe == 1)
	{
		if (strategy < 1 || strategy > 3)
			return fail(NDBHIP_ERR_UNSUPPORTED, "hnsw: unsupported distance strategy %d", strategy);
		R = R_HNSW_L2 + (strategy - 1);
	}
	else if (recipe >= 2 && recipe <= 5)
	{
		if (strategy < 1 || strategy > 3)
			return fail(NDBHIP_ERR_UNSUPPORTED, "operator kernels: strategy must be 1 (<->), 2 (<=>) or 3 (<#>)");
	}
	else
		return fail(NDBHIP_ERR_INVALID, "recipe must be 0 (ivf), 1 (hnsw), 2 (operator, scalar build), "
					"3 (operator, AVX2 build), 4 (operator, AVX-512 build) or 5 (halfvec operators)");
	const size_t esz = recipe == 5 ? 2 : 4;	/* recipe 5: queries / vectors are fp16 images */
	float	   *d_q = nullptr, *d_v = nullptr, *d_o = nullptr;

	HIP_TRY(hipMalloc((void **) &d_q, (size_t) nq * dim * esz));
	HIP_TRY(hipMalloc((void **) &d_v, (size_t) nv * dim * esz));
	HIP_TRY(hipMalloc((void **) &d_o, (size_t) nq * nv * 4));
	HIP_TRY(hipMemcpyAsync(d_q, queries, (size_t) nq * dim * esz, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_v, vectors, (size_t) nv * dim * esz, hipMemcpyHostToDevice, g.stream));
	for (int q0 = 0; q0 < nq; q0 += 65535)
	{
		const int	n = std::min(65535, nq - q0);
		dim3		grid((nv + 255) / 256, n);
		const void *qp = (const char *) d_q + (size_t) q0 * dim * esz;
		float	   *op = d_o + (size_t) q0 * nv;

		if (recipe <= 1)
			LAUNCH_BY_RECIPE(R, k_rows_scan, grid, dim3(256), (const float *) d_v, (uint32_t) nv, dim,
							 (const float *) qp, op, (uint32_t) nv);
		else if (recipe == 2)
			hipLaunchKernelGGL(k_op_distance<0>, grid, dim3(256), 0, g.stream, qp, (const void *) d_v, op,
							   (uint32_t) nv, dim, strategy);
		else if (recipe == 3)
			hipLaunchKernelGGL(k_op_distance<8>, grid, dim3(256), 0, g.stream, qp, (const void *) d_v, op,
							   (uint32_t) nv, dim, strategy);
		else if (recipe == 4)
			hipLaunchKernelGGL(k_op_distance<16>, grid, dim3(256), 0, g.stream, qp, (const void *) d_v, op,
							   (uint32_t) nv, dim, strategy);
		else
			hipLaunchKernelGGL(k_op_distance<1>, grid, dim3(256), 0, g.stream, qp, (const void *) d_v, op,
							   (uint32_t) nv, dim, strategy);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(results, d_o, (size_t) nq * nv * 4, hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipFree(d_q));
	HIP_TRY(hipFree(d_v));
	HIP_TRY(hipFree(d_o));
	g.host_rows += (uint64_t) nq * nv;
	g.host_bytes += (uint64_t) nq * nv * dim * esz;
	return NDBHIP_OK;
}


/* out[i] = the SQL operator's distance of the PAIR (A[i], B[i]): the pairwise shape of the reference's GPU
 * vtable launchers (include/neurondb_gpu_backend.h:54-65), with the arithmetic of the CPU functions they fall
 * back to (src/vector/vector_distance.c:93-227), so a result does not depend on whether the device served it */
__global__ __launch_bounds__(256) void
k_op_pairs(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, uint32_t n, int dim,
		   int strategy)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;

	if (i < n)
		out[i] = op_scalar_pair<false>((const char *) (a + (size_t) i * dim), (const char *) (b + (size_t) i * dim),
									   dim, strategy);
}

extern "C" int
ndbhip_pair_distance(const float *A, const float *B, float *out, int n, int dim, int strategy)
{
	if (need_init()) return NDBHIP_ERR_NODEVICE;
	if (n < 0 || dim < 1 || dim > 32767)
		return fail(NDBHIP_ERR_INVALID, "bad sizes");
	if (strategy < 1 || strategy > 3)
		return fail(NDBHIP_ERR_UNSUPPORTED, "operator kernels: strategy must be 1 (<->), 2 (<=>) or 3 (<#>)");
	if (n == 0)
		return NDBHIP_OK;
	if (!A || !B || !out)
		return fail(NDBHIP_ERR_INVALID, "NULL pointer");
	float	   *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
	const size_t bytes = (size_t) n * dim * sizeof(float);

	HIP_TRY(hipMalloc((void **) &d_a, bytes));
	HIP_TRY(hipMalloc((void **) &d_b, bytes));
	HIP_TRY(hipMalloc((void **) &d_o, (size_t) n * sizeof(float)));
	HIP_TRY(hipMemcpyAsync(d_a, A, bytes, hipMemcpyHostToDevice, g.stream));
	HIP_TRY(hipMemcpyAsync(d_b, B, bytes, hipMemcpyHostToDevice, g.stream));
	hipLaunchKernelGGL(k_op_pairs, dim3((n + 255) / 256), dim3(256), 0, g.stream, (const float *) d_a,
					   (const float *) d_b, d_o, (uint32_t) n, dim, strategy);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(out, d_o, (size_t) n * sizeof(float), hipMemcpyDeviceToHost, g.stream));
	HIP_TRY(hipStreamSynchronize(g.stream));
	HIP_TRY(hipFree(d_a));
	HIP_TRY(hipFree(d_b));
	HIP_TRY(hipFree(d_o));
	g.host_rows += (uint64_t) n;
	g.host_bytes += (uint64_t) n * dim * 8;
	return NDBHIP_OK;
}

#endif							/* NDBHIP_OPS_H */
