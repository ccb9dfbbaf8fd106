/*
 * mfma_probe.hip — what does v_mfma_f32_32x32x16_{f16,bf16} do to the low-order bits on gfx950?
 *
 * The screened scan's bound pass (csrc/ndbhip_screen16.hip) computes dot products with fp16-input MFMA and has
 * to bound the instruction's accumulation error.  The ISA text gives no rule for it, so ndbhip_common.h assumes
 * a MODEL that covers every plausible implementation (any order, any tree, truncation or round-to-nearest,
 * an aligned multi-operand adder with >= 24 kept bits):
 *
 *      | D - (C + sum_k a_k b_k) |  <=  NDB_MFMA_THETA * ( |C| + sum_k |a_k b_k| ),   NDB_MFMA_THETA = 17 * 2^-23
 *
 * and this program (run on the GPU box: tools/run_mfma_probe.sh, output committed under profiles/) measures
 * how far the hardware stays inside it: directed cases that separate the implementations from each other, and
 * random / adversarial stress against exact fp64 arithmetic (a product of two halfs is exact in fp32 and in
 * fp64; 17 fp64 additions err by < 2^-49 of sum|terms|).  The same stress runs in tests/test_gpu_mfma_model.py
 * through the library's own probe entry point, so a part that breaks the model fails the GPU suite.
 *
 * Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_probe.hip -o gpurun_out/mfma_probe
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

/* A [nt][32][16], B [nt][16][32] (bits), C/D [nt][32][32]; one wave per tile; chain = how many times the same
 * A/B is accumulated (D = C + chain * A.B in exact arithmetic) */
template <bool BF>
__global__ void k_mfma(const uint16_t *A, const uint16_t *B, const float *C, float *D, int chain)
{
	const int	t = blockIdx.x, lane = threadIdx.x;
	const int	i = lane & 31, kh = lane >> 5;
	s8			a, b;
	f16v		acc;

	for (int e = 0; e < 8; e++)
	{
		a[e] = (short) A[((size_t) t * 32 + i) * 16 + kh * 8 + e];
		b[e] = (short) B[((size_t) t * 16 + kh * 8 + e) * 32 + i];
	}
	for (int r = 0; r < 16; r++)
		acc[r] = C[((size_t) t * 32 + ((r & 3) + 8 * (r >> 2) + 4 * kh)) * 32 + i];
	for (int c = 0; c < chain; c++)
	{
		if (BF)
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
														  __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), acc, 0, 0, 0);
		else
			acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), acc, 0, 0, 0);
	}
	for (int r = 0; r < 16; r++)
		D[((size_t) t * 32 + ((r & 3) + 8 * (r >> 2) + 4 * kh)) * 32 + i] = acc[r];
}

static uint16_t f2h(float f)		/* round to nearest even, with subnormals */
{
	_Float16	h = (_Float16) f;
	uint16_t	u;

	memcpy(&u, &h, 2);
	return u;
}
static float h2f(uint16_t u)
{
	_Float16	h;

	memcpy(&h, &u, 2);
	return (float) h;
}
static uint16_t f2bf(float f)
{
	uint32_t	u;

	memcpy(&u, &f, 4);
	u += 0x7FFFu + ((u >> 16) & 1u);
	return (uint16_t) (u >> 16);
}
static float bf2f(uint16_t b)
{
	uint32_t	u = (uint32_t) b << 16;
	float		f;

	memcpy(&f, &u, 4);
	return f;
}

struct Dev
{
	uint16_t   *A, *B;
	float	   *C, *D;
	int			nt;
};

static void run(Dev &d, bool bf, const std::vector<uint16_t> &A, const std::vector<uint16_t> &B,
				const std::vector<float> &C, std::vector<float> &D, int nt, int chain)
{
	CK(hipMemcpy(d.A, A.data(), A.size() * 2, hipMemcpyHostToDevice));
	CK(hipMemcpy(d.B, B.data(), B.size() * 2, hipMemcpyHostToDevice));
	CK(hipMemcpy(d.C, C.data(), C.size() * 4, hipMemcpyHostToDevice));
	if (bf)
		hipLaunchKernelGGL(k_mfma<true>, dim3(nt), dim3(64), 0, 0, d.A, d.B, d.C, d.D, chain);
	else
		hipLaunchKernelGGL(k_mfma<false>, dim3(nt), dim3(64), 0, 0, d.A, d.B, d.C, d.D, chain);
	CK(hipDeviceSynchronize());
	D.resize((size_t) nt * 1024);
	CK(hipMemcpy(D.data(), d.D, D.size() * 4, hipMemcpyDeviceToHost));
}

/* one directed case: element (0,0) of the tile gets C = c and the products a[k]*b[k] */
static float directed(Dev &d, bool bf, float c, const float *a, const float *b)
{
	std::vector<uint16_t> A(512, 0), B(512, 0);
	std::vector<float> C(1024, 0.f), D;

	for (int k = 0; k < 16; k++)
	{
		A[k] = bf ? f2bf(a[k]) : f2h(a[k]);
		B[k * 32] = bf ? f2bf(b[k]) : f2h(b[k]);
	}
	C[0] = c;
	run(d, bf, A, B, C, D, 1, 1);
	return D[0];
}

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
	uint64_t	z = (rng_s += 0x9E3779B97F4A7C15ull);

	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static double urand() { return (double) (rnd() >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
	const int	nt = argc > 1 ? atoi(argv[1]) : 4096;
	Dev			d;

	d.nt = nt;
	CK(hipMalloc((void **) &d.A, (size_t) nt * 512 * 2));
	CK(hipMalloc((void **) &d.B, (size_t) nt * 512 * 2));
	CK(hipMalloc((void **) &d.C, (size_t) nt * 1024 * 4));
	CK(hipMalloc((void **) &d.D, (size_t) nt * 1024 * 4));

	for (int bf = 0; bf < 2; bf++)
	{
		const char *nm = bf ? "bf16" : "f16";
		float		a[16], b[16];

		printf("== v_mfma_f32_32x32x16_%s: directed cases (element [0][0], k = 0..15)\n", nm);
		for (int k = 0; k < 16; k++) { a[k] = 1.f; b[k] = 1.f; }
		printf("T1  C=2^24, 16 x (1*1)             exact 16777232   got %.1f   (16777216: C first, each +1 lost; 16777232: products summed first or wide adder)\n",
			   directed(d, bf, 16777216.f, a, b));
		for (int k = 0; k < 16; k++) { a[k] = 0.000244140625f; b[k] = 0.000244140625f; }	/* 2^-12 * 2^-12 = 2^-24 */
		printf("T2  C=1, 16 x 2^-24                exact 1+2^-20    got 1+%.3g\n", (double) directed(d, bf, 1.f, a, b) - 1.0);
		for (int k = 0; k < 16; k++) { a[k] = 1.f; b[k] = 1.f; }
		a[0] = 4096.f; b[0] = 4096.f;
		printf("T3  C=0, 2^24 + 15 x 1             exact 16777231   got %.1f\n", directed(d, bf, 0.f, a, b));
		a[15] = 4096.f; b[15] = 4096.f; a[0] = 1.f; b[0] = 1.f;
		printf("T3b C=0, 15 x 1 + 2^24 (k=15)      exact 16777231   got %.1f\n", directed(d, bf, 0.f, a, b));
		for (int k = 0; k < 16; k++) { a[k] = 0.03125f; b[k] = 0.03125f; }	/* 2^-10 */
		a[0] = 4096.f; b[0] = -4096.f; a[1] = 4096.f; b[1] = 4096.f;
		printf("T4  C=0, -2^24 + 2^24 + 14 x 2^-10 exact %.10g got %.10g\n", 14.0 / 1024.0, directed(d, bf, 0.f, a, b));
		for (int k = 0; k < 16; k++) { a[k] = 0.0009765625f; b[k] = 0.0009765625f; }	/* 2^-20 */
		a[0] = 4096.f; b[0] = -4096.f;
		printf("T5  C=2^24, -2^24 + 15 x 2^-20     exact %.10g got %.10g\n", 15.0 / 1048576.0, directed(d, bf, 16777216.f, a, b));
		for (int k = 0; k < 16; k++) { a[k] = 0.f; b[k] = 0.f; }
		a[0] = 1.f; b[0] = 1.f;
		printf("T6a C=2^24 + 1                     RNE 16777216 RZ 16777216   got %.1f\n", directed(d, bf, 16777216.f, a, b));
		a[0] = 3.f;
		printf("T6b C=2^24 + 3                     RNE 16777220 RZ 16777218   got %.1f\n", directed(d, bf, 16777216.f, a, b));
		a[0] = 1.f; a[1] = 1.f; b[1] = 1.f; a[2] = 1.f; b[2] = 1.f;
		printf("T6c C=2^24 + 1 + 1 + 1             seq-RNE 16777216; fused RNE 16777220   got %.1f\n", directed(d, bf, 16777216.f, a, b));
		a[0] = -1.f; a[1] = 0.f; a[2] = 0.f;
		printf("T6d C=-2^24-... C=2^24+2, + (-1)   exact 16777217 RNE 16777216 or 16777218  got %.1f\n", directed(d, bf, 16777218.f, a, b));
		if (!bf)
		{
			for (int k = 0; k < 16; k++) { a[k] = 0.f; b[k] = 0.f; }
			a[0] = 5.9604645e-8f; b[0] = 1.f;		/* 2^-24: the smallest fp16 subnormal */
			printf("T7  fp16 subnormal input 2^-24 * 1  exact 5.96e-8   got %.6g   (0 = inputs flushed)\n", directed(d, bf, 0.f, a, b));
			a[0] = 3.0517578e-5f; b[0] = 1.f;		/* 2^-15: subnormal */
			printf("T7b fp16 subnormal input 2^-15 * 1  exact 3.05e-5   got %.6g\n", directed(d, bf, 0.f, a, b));
			a[0] = 6.1035156e-5f; b[0] = 6.1035156e-5f;	/* 2^-14 * 2^-14 = 2^-28 */
			printf("T7c product 2^-28 of two normals    exact 3.73e-9   got %.6g\n", directed(d, bf, 0.f, a, b));
			a[0] = 65504.f; b[0] = 65504.f; a[1] = 65504.f; b[1] = 65504.f;
			printf("T8  2 x 65504^2                     exact %.9g got %.9g\n", 2.0 * 65504.0 * 65504.0, directed(d, bf, 0.f, a, b));
		}
		else
		{
			for (int k = 0; k < 16; k++) { a[k] = 0.f; b[k] = 0.f; }
			a[0] = 1e-39f; b[0] = 1.f;				/* bf16 subnormal */
			printf("T7  bf16 subnormal input 1e-39 * 1  got %.6g   (0 = inputs flushed)\n", directed(d, bf, 0.f, a, b));
			a[0] = 1e-20f; b[0] = 1e-20f;
			printf("T7c product 1e-40 (subnormal fp32)  got %.6g\n", directed(d, bf, 0.f, a, b));
		}
		/* a product below the accumulator's half ulp, many of them: T9 separates "align to the largest and
		 * truncate" from a chain of fp32 additions */
		for (int k = 0; k < 16; k++) { a[k] = 1.f; b[k] = 0.5f; }
		printf("T9  C=2^24 + 16 x 0.5              exact 16777224   got %.1f\n", directed(d, bf, 16777216.f, a, b));
		for (int k = 0; k < 16; k++) { a[k] = 1.f; b[k] = 0.25f; }
		printf("T9b C=2^24 + 16 x 0.25             exact 16777220   got %.1f\n", directed(d, bf, 16777216.f, a, b));
		for (int k = 0; k < 16; k++) { a[k] = 1.f; b[k] = 0.0625f; }
		printf("T9c C=2^24 + 16 x 2^-4             exact 16777217 (RNE of exact: 16777216 tie->even, or 16777218)   got %.1f\n", directed(d, bf, 16777216.f, a, b));
		for (int k = 0; k < 16; k++) { a[k] = 1.f; b[k] = 0.09375f; }
		printf("T9d C=2^24 + 16 x 0.09375          exact 16777217.5 -> 16777218   got %.1f\n", directed(d, bf, 16777216.f, a, b));

		/* random / adversarial stress */
		static const char *kinds[] = {"uniform [-1,1)", "exponents spread over 2^+-12", "C huge vs products", "cancelling pairs + dust",
			"positive terms only", "one big product + dust", "chain of 48 (768 dims), uniform", "chain of 48, positive"};
		for (int kind = 0; kind < 8; kind++)
		{
			std::vector<uint16_t> A((size_t) nt * 512), B((size_t) nt * 512);
			std::vector<float> C((size_t) nt * 1024), D;
			const int	chain = kind >= 6 ? 48 : 1;

			for (size_t i = 0; i < A.size(); i++)
			{
				double		x = urand() * 2 - 1, y = urand() * 2 - 1;

				if (kind == 1) { x = ldexp(x, (int) (rnd() % 25) - 12); y = ldexp(y, (int) (rnd() % 25) - 12); }
				if (kind == 4 || kind == 7) { x = fabs(x); y = fabs(y); }
				if (kind == 5) { x = ldexp(x, -10); }
				if (bf)
				{
					/* keep bf16 tests in a range where fp32 neither overflows nor underflows */
					A[i] = f2bf((float) x); B[i] = f2bf((float) y);
				}
				else
				{
					A[i] = f2h((float) x); B[i] = f2h((float) y);
				}
			}
			if (kind == 3)
				for (int t = 0; t < nt; t++)
					for (int i = 0; i < 32; i++)
						for (int k = 0; k < 14; k += 2)		/* a[k+1] = -a[k] on the same b: exact cancellation */
						{
							A[((size_t) t * 32 + i) * 16 + k + 1] = A[((size_t) t * 32 + i) * 16 + k] ^ 0x8000u;
						}
			if (kind == 3)
				for (int t = 0; t < nt; t++)
					for (int j = 0; j < 32; j++)
						for (int k = 0; k < 14; k += 2)
							B[((size_t) t * 16 + k + 1) * 32 + j] = B[((size_t) t * 16 + k) * 32 + j];
			if (kind == 5)
				for (int t = 0; t < nt; t++)
					for (int i = 0; i < 32; i++)
						A[((size_t) t * 32 + i) * 16 + (rnd() & 15)] = bf ? f2bf(700.f) : f2h(700.f);
			for (size_t i = 0; i < C.size(); i++)
			{
				double		c = urand() * 2 - 1;

				if (kind == 1) c = ldexp(c, (int) (rnd() % 25) - 12);
				if (kind == 2) c = ldexp(c, 20);
				if (kind == 3) c = 0;
				if (kind == 4 || kind == 7) c = fabs(c);
				if (kind >= 6) c = 0;
				C[i] = (float) c;
			}
			run(d, bf, A, B, C, D, nt, chain);
			double		worst = 0, worst_u = 0, sum_ratio = 0;
			size_t		n = 0;

			for (int t = 0; t < nt; t++)
				for (int i = 0; i < 32; i++)
					for (int j = 0; j < 32; j++)
					{
						double		exact = C[((size_t) t * 32 + i) * 32 + j], mag = fabs(exact), dot = 0, dmag = 0;

						for (int k = 0; k < 16; k++)
						{
							const double av = bf ? bf2f(A[((size_t) t * 32 + i) * 16 + k]) : h2f(A[((size_t) t * 32 + i) * 16 + k]);
							const double bv = bf ? bf2f(B[((size_t) t * 16 + k) * 32 + j]) : h2f(B[((size_t) t * 16 + k) * 32 + j]);

							dot += av * bv;
							dmag += fabs(av * bv);
						}
						exact += chain * dot;
						mag += chain * dmag;
						const double got = D[((size_t) t * 32 + i) * 32 + j];
						const double err = fabs(got - exact);
						const double ratio = mag > 0 ? err / mag : (err > 0 ? 1e9 : 0);
						/* error in units of the result's own ulp/2: <= 1 means "correctly rounded result of the exact sum" */
						const double ulp_half = ldexp(1.0, (exact != 0 ? ilogb(exact) : -126) - 24);

						if (ratio > worst) worst = ratio;
						if (err / ulp_half > worst_u) worst_u = err / ulp_half;
						sum_ratio += ratio;
						n++;
					}
			printf("S%d %-34s chain %2d: max err / (|C| + sum|p|) = %8.3f x 2^-24   mean %6.3f x 2^-24   max err / half-ulp(result) = %9.2f   model %d x 2^-24\n",
				   kind, kinds[kind], chain, worst * 16777216.0, sum_ratio / n * 16777216.0, worst_u, 34 * chain);
		}
	}
	return 0;
}
