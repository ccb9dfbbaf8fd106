/*
 * ndb_backend.cpp — struct ndb_gpu_backend's members for this path over the device library (see
 * include/ndb_backend.h).  Host code only; every member returns 0 or a negative code, never throws.
 * Reference paths are relative to NeuronDB/.
 */
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "../../include/ndb_backend.h"

static int	be_device = 0;

static int
be_init(void)
{
	return ndbhip_init(be_device) == NDBHIP_OK ? 0 : -1;	/* lazy, per backend: src/gpu/common/gpu_core.c:240-310 */
}

static void
be_shutdown(void)
{
	(void) ndbhip_shutdown();
}

static int
be_is_available(void)
{
	return ndbhip_device_count() > 0;
}

static int
be_device_count(void)
{
	const int	n = ndbhip_device_count();

	return n < 0 ? 0 : n;
}

static int
be_device_info(int device_id, ndb_hip_device_info *info)
{
	hipDeviceProp_t prop;
	size_t		free_b = 0, total_b = 0;

	if (!info || device_id < 0 || device_id >= be_device_count())
		return -1;
	if (hipGetDeviceProperties(&prop, device_id) != hipSuccess)
		return -1;
	memset(info, 0, sizeof *info);
	info->device_id = device_id;
	strncpy(info->name, prop.name, sizeof info->name - 1);
	info->total_memory_bytes = prop.totalGlobalMem;
	info->compute_major = prop.major;	/* as ndb_rocm_device_info does: gpu_backend_rocm.c:336-337 */
	info->compute_minor = prop.minor;
	info->is_available = true;
	if (device_id == be_device && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
		info->free_memory_bytes = free_b;
	return 0;
}

static int
be_set_device(int device_id)
{
	if (device_id < 0 || device_id >= be_device_count())
		return -1;
	be_device = device_id;		/* takes effect at init: the library binds one device per process */
	return 0;
}

static int
be_mem_alloc(void **ptr, size_t bytes)
{
	return (ptr && hipMalloc(ptr, bytes) == hipSuccess) ? 0 : -1;
}

static int
be_mem_free(void *ptr)
{
	return hipFree(ptr) == hipSuccess ? 0 : -1;
}

static int
be_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
	return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}

static int
be_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
	return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

/* ndb_rocm_launch_l2_distance: src/gpu/rocm/gpu_backend_rocm.c:752-825 (n pairs, host pointers) */
static int
be_launch_l2_distance(const float *A, const float *B, float *out, int n, int d, ndb_stream_t stream)
{
	(void) stream;
	if (!A || !B || !out || n <= 0 || d <= 0)	/* :764-766 */
		return -1;
	return ndbhip_pair_distance(A, B, out, n, d, 1) == NDBHIP_OK ? 0 : -1;
}

/* ndb_rocm_launch_cosine: gpu_backend_rocm.c:827-903 */
static int
be_launch_cosine(const float *A, const float *B, float *out, int n, int d, ndb_stream_t stream)
{
	(void) stream;
	if (!A || !B || !out || n <= 0 || d <= 0)
		return -1;
	return ndbhip_pair_distance(A, B, out, n, d, 2) == NDBHIP_OK ? 0 : -1;
}

/* ndb_rocm_launch_kmeans_assign: gpu_backend_rocm.c:905-935 */
static int
be_launch_kmeans_assign(const float *X, const float *C, int *idx, int n, int d, int k, ndb_stream_t stream)
{
	(void) stream;
	if (!X || !C || !idx || n <= 0 || d <= 0 || k <= 0)
		return -1;
	return ndbhip_kmeans_assign(X, C, idx, n, d, k) == NDBHIP_OK ? 0 : -1;
}

/* ndb_rocm_launch_kmeans_update: gpu_backend_rocm.c:937-975 */
static int
be_launch_kmeans_update(const float *X, const int *idx, float *C, int n, int d, int k, ndb_stream_t stream)
{
	(void) stream;
	if (!X || !C || !idx || n <= 0 || d <= 0 || k <= 0)
		return -1;
	return ndbhip_kmeans_update(X, idx, C, n, d, k) == NDBHIP_OK ? 0 : -1;
}

/* ndb_rocm_launch_quant_fp16: gpu_backend_rocm.c:977-992 */
static int
be_launch_quant_fp16(const float *in, void *out, int n, ndb_stream_t stream)
{
	(void) stream;
	if (!in || !out || n <= 0)
		return -1;
	return ndbhip_quant_fp16(in, (uint16_t *) out, n) == NDBHIP_OK ? 0 : -1;
}

static int
be_stream_create(ndb_stream_t *stream)
{
	hipStream_t s = nullptr;

	if (!stream || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
		return -1;
	*stream = (ndb_stream_t) s;
	return 0;
}

static int
be_stream_destroy(ndb_stream_t stream)
{
	return hipStreamDestroy((hipStream_t) stream) == hipSuccess ? 0 : -1;
}

static int
be_stream_synchronize(ndb_stream_t stream)
{
	return hipStreamSynchronize((hipStream_t) stream) == hipSuccess ? 0 : -1;
}

static const ndb_hip_backend the_backend = {
	"ndbhip", "AMD", NDB_HIP_BACKEND_ROCM, 0u, 100,
	be_init, be_shutdown, be_is_available,
	be_device_count, be_device_info, be_set_device,
	be_mem_alloc, be_mem_free, be_memcpy_h2d, be_memcpy_d2h,
	be_launch_l2_distance, be_launch_cosine, be_launch_kmeans_assign, be_launch_kmeans_update,
	be_launch_quant_fp16,
	nullptr, nullptr, nullptr, nullptr, nullptr,	/* int8 / int4 / fp8 / binary quantisers: out of scope -> CPU */
	nullptr,										/* launch_pq_encode */
};

extern "C" void
ndb_hip_backend_streams(int (**create) (ndb_stream_t *), int (**destroy) (ndb_stream_t), int (**synchronize) (ndb_stream_t))
{
	if (create) *create = be_stream_create;
	if (destroy) *destroy = be_stream_destroy;
	if (synchronize) *synchronize = be_stream_synchronize;
}

extern "C" const ndb_hip_backend *
ndb_hip_backend_get(void)
{
	return &the_backend;
}
