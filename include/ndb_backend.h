/*
 * ndb_backend.h — the reference's GPU plugin vtable for this path (struct ndb_gpu_backend,
 * include/neurondb_gpu_backend.h:28-110) filled from the device library, without PostgreSQL: same member
 * names, argument lists and return convention (0 = success, negative = failure, the caller falls back to its
 * CPU function: :24-26, src/gpu/common/gpu_distance.c:50-51).  Only the members a distance / k-means caller
 * uses are present; the ML, LLM and PQ launchers of the reference vtable are out of scope.
 * A maintainer registers it by copying the pointers into an ndb_gpu_backend and calling
 * ndb_gpu_register_backend (src/gpu/common/gpu_backend_registry.c:91-131) — INTEGRATION.md §11.
 *
 * The launchers take HOST pointers, like the ROCm backend's (src/gpu/rocm/gpu_backend_rocm.c:752-1000), and
 * return results computed with the arithmetic of the CPU fallbacks, bit for bit (include/ndbhip.h,
 * "launcher shapes").  `stream` is accepted for signature compatibility; the calls are synchronous, as the
 * reference's are (they return host results).
 */
#ifndef NDB_BACKEND_H
#define NDB_BACKEND_H

#include <stddef.h>
#include <stdint.h>

#include "ndbhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ndb_stream_t;		/* include/neurondb_gpu_backend.h: opaque stream handle */

typedef struct ndb_hip_device_info
{
	int			device_id;
	char		name[256];
	size_t		total_memory_bytes;
	size_t		free_memory_bytes;
	int			compute_units;
	int			is_available;
}			ndb_hip_device_info;

typedef struct ndb_hip_backend
{
	/* Identity */
	const char *name;
	const char *provider;
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

	/* Streams */
	int			(*stream_create) (ndb_stream_t *stream);
	int			(*stream_destroy) (ndb_stream_t stream);
	int			(*stream_synchronize) (ndb_stream_t stream);
}			ndb_hip_backend;

/* the one instance (static storage) */
const ndb_hip_backend *ndb_hip_backend_get(void);

#ifdef __cplusplus
}
#endif
#endif							/* NDB_BACKEND_H */
