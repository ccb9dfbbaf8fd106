/*
 * ndb_backend.h — the reference's GPU plugin vtable for this path (struct ndb_gpu_backend,
 * include/neurondb_gpu_backend.h:28-110) filled from the device library, without PostgreSQL: same member
 * names, argument lists and return convention (0 = success, negative = failure, the caller falls back to its
 * CPU function: :24-26, src/gpu/common/gpu_distance.c:50-51).  Only the members a distance / k-means caller
 * uses are present; the ML, LLM and PQ launchers of the reference vtable are out of scope.
 * The struct is a prefix image of the reference's, so it registers with one memcpy (see below, INTEGRATION.md §11
 * and pgext/ndbhip_glue.c).
 *
 * The launchers take HOST pointers, like the ROCm backend's (src/gpu/rocm/gpu_backend_rocm.c:752-1000), and
 * return results computed with the arithmetic of the CPU fallbacks, bit for bit (include/ndbhip.h,
 * "launcher shapes").  `stream` is accepted for signature compatibility; the calls are synchronous, as the
 * reference's are (they return host results).
 */
#ifndef NDB_BACKEND_H
#define NDB_BACKEND_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "ndbhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ndb_stream_t;		/* include/neurondb_gpu_backend.h: opaque stream handle */

/* image of NDBGpuDeviceInfo (include/neurondb_gpu_types.h:45-54), member for member */
typedef struct ndb_hip_device_info
{
	int			device_id;
	char		name[256];
	size_t		total_memory_bytes;
	size_t		free_memory_bytes;
	int			compute_major;		/* gfx950 -> 9 */
	int			compute_minor;		/* gfx950 -> 5 */
	bool		is_available;
}			ndb_hip_device_info;

/* NDBGpuBackendKind (include/neurondb_gpu_types.h:20-26) */
typedef enum ndb_hip_backend_kind
{
	NDB_HIP_BACKEND_NONE = 0,
	NDB_HIP_BACKEND_CUDA,
	NDB_HIP_BACKEND_ROCM,
	NDB_HIP_BACKEND_METAL
}			ndb_hip_backend_kind;

/*
 * A PREFIX IMAGE of struct ndb_gpu_backend (include/neurondb_gpu_backend.h:28-110): the same members in the same
 * order with the same types from `name` through `launch_pq_encode`, i.e. everything the reference declares
 * before its random-forest / ML / LLM launchers (:112-349, out of scope).  So a maintainer registers it without
 * touching a pointer:
 *
 *     static ndb_gpu_backend b;                                  // zero: every ML launcher NULL = "not provided"
 *     memcpy(&b, ndb_hip_backend_get(), sizeof(ndb_hip_backend));
 *     ndb_hip_backend_streams(&b.stream_create, &b.stream_destroy, &b.stream_synchronize);   // the reference's LAST three members (:351-353)
 *     ndb_gpu_register_backend(&b);                              // src/gpu/common/gpu_backend_registry.c:91-131
 *
 * (pgext/ndbhip_glue.c does exactly that, with a _Static_assert on every offset.)  The quantisation launchers
 * other than fp16 and launch_pq_encode are NULL: the reference's callers treat a NULL launcher as "fall back
 * to the CPU" (src/gpu/common/gpu_backend_registry.c looks the pointer up before calling).
 */
typedef struct ndb_hip_backend
{
	/* Identity */
	const char *name;
	const char *provider;
	ndb_hip_backend_kind kind;
	unsigned int features;
	int			priority;

	/* Lifecycle */
	int			(*init) (void);
	void		(*shutdown) (void);
	int			(*is_available) (void);

	/* Device management */
	int			(*device_count) (void);
	int			(*device_info) (int device_id, ndb_hip_device_info *info);
	int			(*set_device) (int device_id);

	/* Memory helpers */
	int			(*mem_alloc) (void **ptr, size_t bytes);
	int			(*mem_free) (void *ptr);
	int			(*memcpy_h2d) (void *dst, const void *src, size_t bytes);
	int			(*memcpy_d2h) (void *dst, const void *src, size_t bytes);

	/* Launchers */
	int			(*launch_l2_distance) (const float *A, const float *B, float *out, int n, int d, ndb_stream_t stream);
	int			(*launch_cosine) (const float *A, const float *B, float *out, int n, int d, ndb_stream_t stream);
	int			(*launch_kmeans_assign) (const float *X, const float *C, int *idx, int n, int d, int k,
										 ndb_stream_t stream);
	int			(*launch_kmeans_update) (const float *X, const int *idx, float *C, int n, int d, int k,
										 ndb_stream_t stream);
	int			(*launch_quant_fp16) (const float *in, void *out, int n, ndb_stream_t stream);
	int			(*launch_quant_int8) (const float *in, int8_t *out, int n, float scale, ndb_stream_t stream);			/* NULL */
	int			(*launch_quant_int4) (const float *in, unsigned char *out, int n, float scale, ndb_stream_t stream);	/* NULL */
	int			(*launch_quant_fp8_e4m3) (const float *in, unsigned char *out, int n, ndb_stream_t stream);			/* NULL */
	int			(*launch_quant_fp8_e5m2) (const float *in, unsigned char *out, int n, ndb_stream_t stream);			/* NULL */
	int			(*launch_quant_binary) (const float *in, uint8_t *out, int n, ndb_stream_t stream);					/* NULL */
	int			(*launch_pq_encode) (const float *X, const float *codebooks, uint8_t *codes, int n, int d, int m,
									 int ks, ndb_stream_t stream);														/* NULL */
}			ndb_hip_backend;

/* the reference's last three members (stream_create / stream_destroy / stream_synchronize, :351-353) */
void		ndb_hip_backend_streams(int (**create) (ndb_stream_t *), int (**destroy) (ndb_stream_t),
									int (**synchronize) (ndb_stream_t));

/* the one instance (static storage) */
const ndb_hip_backend *ndb_hip_backend_get(void);

#ifdef __cplusplus
}
#endif
#endif							/* NDB_BACKEND_H */
