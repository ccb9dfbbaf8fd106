/*
 * ndbhip_comm.cpp — the multi-GPU exchange of the sharded IVF search, inside the C ABI.
 *
 * One process per GPU; every rank holds a shard of the lists (ndbhip_ivf_shard / _shard_slices) and the same
 * centroids.  ndbhip_ivf_search_sharded runs one batch:
 *
 *   1. this rank selects the probes of ITS slice of the queries (the selection is a pure function of query and
 *      centroids: ivfSelectClusters, src/index/ivf_am.c:1597-1717) and the slices are all-gathered
 *      (nq x nprobe x 4 bytes);
 *   2. it scans the probed lists it holds for ALL queries and keeps, per query, its tie-complete subset
 *      (<= 3k records of 16 bytes: float4 bits, position in the reference's candidates[], TID);
 *   3. the records are all-gathered (nq x 3k x 16 bytes per rank) and the union is merged by replaying the
 *      reference's selection sort (ivf_am.c:1856-1881; merge order of src/util/distributed.c:204-244: the
 *      smaller distance first, ties by position = by the order a single backend would have met them).
 *
 * Two transports behind ndbhip_comm_allgather:
 *   RCCL  (ndbhip_comm_init)      ncclAllGather on the library's stream, over xGMI between the GPUs of a node.
 *         librccl is opened with dlopen when a communicator is created, so the library loads without it
 *         (a PostgreSQL backend that never shards does not pull it in).
 *   SHM   (ndbhip_comm_init_shm)  a POSIX shared-memory segment on the host: D2H, process-shared barrier, H2D.
 *         For ranks that cannot form an RCCL communicator — several backends on ONE device (RCCL refuses two
 *         ranks per device; tests/test_gpu_dist.py runs the whole flow that way) or a node without RCCL.
 *
 * Everything here sits on top of the public ABI (include/ndbhip.h): the comm layer owns no kernels.
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <string>

#include "../../include/ndbhip.h"

extern "C" int ndbhip_internal_fail(int code, const char *fmt, ...);
extern "C" void ndbhip_internal_set_thr_hook(int (*fn) (float *, size_t));

namespace
{
struct ShmHeader
{
	volatile uint32_t magic;		/* set last by rank 0 */
	uint32_t	world;
	uint64_t	slot_bytes;
	pthread_barrier_t barrier;
};

struct RcclApi
{
	void	   *handle = nullptr;
	ncclResult_t (*GetUniqueId) (ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank) (ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*AllGather) (const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*CommDestroy) (ncclComm_t) = nullptr;
	const char *(*GetErrorString) (ncclResult_t) = nullptr;
	/* point-to-point, for ndbhip_comm_alltoallv (the distributed build's row routing) */
	ncclResult_t (*Send) (const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv) (void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart) (void) = nullptr;
	ncclResult_t (*GroupEnd) (void) = nullptr;
	/* the per-query thresholds' minimum over the ranks (ndbhip_comm_allreduce_min_f32) */
	ncclResult_t (*AllReduce) (const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
};

struct Comm
{
	int			kind = 0;			/* 0 none, 1 RCCL, 2 SHM */
	int			rank = 0, world = 1;
	ncclComm_t	nccl = nullptr;
	/* SHM */
	std::string shm_name;
	ShmHeader  *hdr = nullptr;
	unsigned char *slots = nullptr;
	size_t		map_bytes = 0, slot_bytes = 0;
	/* workspace of ndbhip_ivf_search_sharded */
	int		   *probes_mine = nullptr;	size_t probes_mine_n = 0;
	int		   *probes_all = nullptr;	size_t probes_all_n = 0;
	ndbhip_cand *cand = nullptr;		size_t cand_n = 0;
	ndbhip_cand *cand_all = nullptr;	size_t cand_all_n = 0;
	int		   *ncand = nullptr;		size_t ncand_n = 0;
	int		   *ncand_all = nullptr;	size_t ncand_all_n = 0;
	int64_t    *total = nullptr;		size_t total_n = 0;
};

RcclApi		rccl;
Comm		comm;

int
load_rccl()
{
	if (rccl.handle)
		return 0;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};

	for (const char *n : names)
	{
		rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
		if (rccl.handle)
			break;
	}
	if (!rccl.handle)
		return ndbhip_internal_fail(NDBHIP_ERR_UNSUPPORTED, "librccl not found: %s", dlerror());
#define SYM(field, name)                                                              \
	do {                                                                              \
		*(void **) (&rccl.field) = dlsym(rccl.handle, name);                          \
		if (!rccl.field)                                                              \
			return ndbhip_internal_fail(NDBHIP_ERR_UNSUPPORTED, "librccl has no %s", name); \
	} while (0)
	SYM(GetUniqueId, "ncclGetUniqueId");
	SYM(CommInitRank, "ncclCommInitRank");
	SYM(AllGather, "ncclAllGather");
	SYM(CommDestroy, "ncclCommDestroy");
	SYM(GetErrorString, "ncclGetErrorString");
	SYM(Send, "ncclSend");
	SYM(Recv, "ncclRecv");
	SYM(GroupStart, "ncclGroupStart");
	SYM(GroupEnd, "ncclGroupEnd");
	SYM(AllReduce, "ncclAllReduce");
#undef SYM
	return 0;
}

template <class T>
int
grow_dev(T *&p, size_t &have, size_t want)
{
	if (want <= have)
		return 0;
	if (p)
		(void) hipFree(p);
	p = nullptr;
	have = 0;
	if (hipMalloc((void **) &p, want * sizeof(T)) != hipSuccess)
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "hipMalloc of %zu bytes failed", want * sizeof(T));
	have = want;
	return 0;
}

void
free_workspace()
{
	void	   *ptrs[] = {comm.probes_mine, comm.probes_all, comm.cand, comm.cand_all, comm.ncand, comm.ncand_all, comm.total};

	for (void *p : ptrs)
		if (p)
			(void) hipFree(p);
	comm.probes_mine = comm.probes_all = nullptr;
	comm.cand = comm.cand_all = nullptr;
	comm.ncand = comm.ncand_all = nullptr;
	comm.total = nullptr;
	comm.probes_mine_n = comm.probes_all_n = comm.cand_n = comm.cand_all_n = comm.ncand_n = comm.ncand_all_n = comm.total_n = 0;
}
}	/* namespace */

extern "C" int
ndbhip_comm_unique_id(void *out_id)
{
	if (!out_id)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "out_id is NULL");
	if (load_rccl())
		return NDBHIP_ERR_UNSUPPORTED;
	ncclUniqueId id;
	const ncclResult_t r = rccl.GetUniqueId(&id);

	if (r != ncclSuccess)
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "ncclGetUniqueId: %s", rccl.GetErrorString(r));
	static_assert(sizeof(id) == NDBHIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
	memcpy(out_id, &id, sizeof(id));
	return NDBHIP_OK;
}

extern "C" int
ndbhip_comm_init(const void *unique_id, int rank, int world)
{
	void	   *stream = nullptr;

	if (ndbhip_get_stream(&stream))
		return NDBHIP_ERR_NODEVICE;
	if (comm.kind)
		return ndbhip_internal_fail(NDBHIP_ERR_STATE, "a communicator already exists (ndbhip_comm_destroy first)");
	if (!unique_id || world < 1 || rank < 0 || rank >= world)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad communicator arguments");
	if (load_rccl())
		return NDBHIP_ERR_UNSUPPORTED;
	ncclUniqueId id;

	memcpy(&id, unique_id, sizeof(id));
	const ncclResult_t r = rccl.CommInitRank(&comm.nccl, world, id, rank);

	if (r != ncclSuccess)
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl.GetErrorString(r));
	comm.kind = 1;
	comm.rank = rank;
	comm.world = world;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_comm_init_shm(const char *name, int rank, int world, size_t slot_bytes)
{
	void	   *stream = nullptr;

	if (ndbhip_get_stream(&stream))
		return NDBHIP_ERR_NODEVICE;
	if (comm.kind)
		return ndbhip_internal_fail(NDBHIP_ERR_STATE, "a communicator already exists (ndbhip_comm_destroy first)");
	if (!name || name[0] != '/' || world < 1 || rank < 0 || rank >= world || slot_bytes < 4096)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad communicator arguments (name must start with '/')");
	const size_t bytes = 4096 + (size_t) world * slot_bytes;
	int			fd = -1;

	if (rank == 0)
	{
		(void) shm_unlink(name);
		fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
		if (fd < 0 || ftruncate(fd, (off_t) bytes) != 0)
		{
			if (fd >= 0) close(fd);
			return ndbhip_internal_fail(NDBHIP_ERR_HIP, "shm_open(%s): %s", name, strerror(errno));
		}
	}
	else
	{
		/* wait (bounded) until rank 0 has created and sized the segment */
		for (int tries = 0; tries < 3000; tries++)
		{
			struct stat st;

			fd = shm_open(name, O_RDWR, 0600);
			if (fd >= 0 && fstat(fd, &st) == 0 && (size_t) st.st_size >= bytes)
				break;
			if (fd >= 0) { close(fd); fd = -1; }
			struct timespec ts = {0, 10 * 1000 * 1000};

			nanosleep(&ts, nullptr);
		}
		if (fd < 0)
			return ndbhip_internal_fail(NDBHIP_ERR_HIP, "shm segment %s did not appear", name);
	}
	void	   *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);

	close(fd);
	if (m == MAP_FAILED)
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "mmap(%s): %s", name, strerror(errno));
	ShmHeader  *h = (ShmHeader *) m;

	if (rank == 0)
	{
		pthread_barrierattr_t at;

		pthread_barrierattr_init(&at);
		pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
		pthread_barrier_init(&h->barrier, &at, (unsigned) world);
		pthread_barrierattr_destroy(&at);
		h->world = (uint32_t) world;
		h->slot_bytes = slot_bytes;
		__sync_synchronize();
		h->magic = 0x4E444243u;
	}
	else
	{
		for (int tries = 0; tries < 3000 && h->magic != 0x4E444243u; tries++)
		{
			struct timespec ts = {0, 10 * 1000 * 1000};

			nanosleep(&ts, nullptr);
		}
		if (h->magic != 0x4E444243u || h->world != (uint32_t) world || h->slot_bytes != slot_bytes)
		{
			munmap(m, bytes);
			return ndbhip_internal_fail(NDBHIP_ERR_STATE, "shm segment %s was not initialised for this group", name);
		}
	}
	comm.kind = 2;
	comm.rank = rank;
	comm.world = world;
	comm.shm_name = name;
	comm.hdr = h;
	comm.slots = (unsigned char *) m + 4096;
	comm.map_bytes = bytes;
	comm.slot_bytes = slot_bytes;
	return NDBHIP_OK;
}

extern "C" int
ndbhip_comm_rank(void)
{
	return comm.kind ? comm.rank : 0;
}

extern "C" int
ndbhip_comm_world(void)
{
	return comm.kind ? comm.world : 1;
}

extern "C" int
ndbhip_comm_destroy(void)
{
	free_workspace();
	if (comm.kind == 1 && comm.nccl)
		(void) rccl.CommDestroy(comm.nccl);
	if (comm.kind == 2 && comm.hdr)
	{
		munmap((void *) comm.hdr, comm.map_bytes);
		if (comm.rank == 0)
			(void) shm_unlink(comm.shm_name.c_str());
	}
	comm = Comm();
	return NDBHIP_OK;
}

/* recv[r * bytes .. (r + 1) * bytes) = rank r's send; device pointers; ordered on the library's stream (RCCL:
 * asynchronous; SHM: returns after the exchange) */
extern "C" int
ndbhip_comm_allgather(const void *d_send, void *d_recv, size_t bytes)
{
	void	   *sv = nullptr;

	if (ndbhip_get_stream(&sv))
		return NDBHIP_ERR_NODEVICE;
	hipStream_t stream = (hipStream_t) sv;

	if (bytes == 0)
		return NDBHIP_OK;
	if (!d_send || !d_recv)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	if (comm.kind == 0 || comm.world == 1)
	{
		if (hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
			return ndbhip_internal_fail(NDBHIP_ERR_HIP, "hipMemcpyAsync failed");
		return NDBHIP_OK;
	}
	if (comm.kind == 1)
	{
		const ncclResult_t r = rccl.AllGather(d_send, d_recv, bytes, ncclInt8, comm.nccl, stream);

		if (r != ncclSuccess)
			return ndbhip_internal_fail(NDBHIP_ERR_HIP, "ncclAllGather: %s", rccl.GetErrorString(r));
		return NDBHIP_OK;
	}
	if (bytes > comm.slot_bytes)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "all-gather of %zu bytes per rank exceeds the segment's slots (%zu)",
									bytes, comm.slot_bytes);
	if (hipMemcpyAsync(comm.slots + (size_t) comm.rank * comm.slot_bytes, d_send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess ||
		hipStreamSynchronize(stream) != hipSuccess)
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "device-to-host copy failed");
	pthread_barrier_wait(&comm.hdr->barrier);		/* every slot is written */
	for (int r = 0; r < comm.world; r++)
		if (hipMemcpyAsync((unsigned char *) d_recv + (size_t) r * bytes, comm.slots + (size_t) r * comm.slot_bytes, bytes,
						   hipMemcpyHostToDevice, stream) != hipSuccess)
			return ndbhip_internal_fail(NDBHIP_ERR_HIP, "host-to-device copy failed");
	if (hipStreamSynchronize(stream) != hipSuccess)
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "stream synchronisation failed");
	pthread_barrier_wait(&comm.hdr->barrier);		/* every slot has been read: it may be overwritten */
	return NDBHIP_OK;
}

/*
 * In-place element-wise minimum of n floats over the ranks (device pointer, the library's stream).  One use: the
 * per-query thresholds of a sharded screened scan — a rank that does not hold a query's own list would otherwise
 * sweep its rows against a threshold nothing has tightened.  RCCL: ncclAllReduce(ncclMin); SHM: through the slots.
 */
extern "C" int
ndbhip_comm_allreduce_min_f32(float *d_buf, size_t n)
{
	void	   *sv = nullptr;

	if (ndbhip_get_stream(&sv))
		return NDBHIP_ERR_NODEVICE;
	hipStream_t stream = (hipStream_t) sv;

	if (n == 0 || comm.kind == 0 || comm.world == 1)
		return NDBHIP_OK;
	if (!d_buf)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	if (comm.kind == 1)
	{
		const ncclResult_t r = rccl.AllReduce(d_buf, d_buf, n, ncclFloat, ncclMin, comm.nccl, stream);

		if (r != ncclSuccess)
			return ndbhip_internal_fail(NDBHIP_ERR_HIP, "ncclAllReduce: %s", rccl.GetErrorString(r));
		return NDBHIP_OK;
	}
	const size_t bytes = n * sizeof(float);

	if (bytes > comm.slot_bytes)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "all-reduce of %zu bytes exceeds the segment's slots (%zu)", bytes,
									comm.slot_bytes);
	float	   *mine = (float *) (comm.slots + (size_t) comm.rank * comm.slot_bytes);
	/* A collective: whatever goes wrong on this rank, it passes BOTH barriers — its peers are waiting in them — and
	 * reports afterwards.  So everything that can fail is done before the first barrier, and a rank that could not
	 * produce its slot contributes +inf (the identity of min: the others' bound stands). */
	float	   *red = (float *) malloc(bytes);
	bool		ok = red != nullptr;

	if (hipMemcpyAsync(mine, d_buf, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess ||
		hipStreamSynchronize(stream) != hipSuccess)
	{
		ok = false;
		for (size_t i = 0; i < n; i++)
			mine[i] = __builtin_inff();
	}
	pthread_barrier_wait(&comm.hdr->barrier);		/* every slot is written */
	if (red)
	{
		memcpy(red, comm.slots, bytes);
		for (int r = 1; r < comm.world; r++)
		{
			const float *o = (const float *) (comm.slots + (size_t) r * comm.slot_bytes);

			for (size_t i = 0; i < n; i++)
				if (o[i] < red[i])
					red[i] = o[i];
		}
	}
	pthread_barrier_wait(&comm.hdr->barrier);		/* every slot has been read: it may be overwritten */
	if (!red)
		return ndbhip_internal_fail(NDBHIP_ERR_NOMEM, "out of host memory");
	ok = ok && hipMemcpyAsync(d_buf, red, bytes, hipMemcpyHostToDevice, stream) == hipSuccess &&
		hipStreamSynchronize(stream) == hipSuccess;
	free(red);
	if (!ok)
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "a copy between the device and the exchange segment failed");
	return NDBHIP_OK;
}

/*
 * Personalised exchange: bytes [send_off[p], send_off[p + 1]) of this rank's d_send go to rank p and arrive at
 * [recv_off[r], recv_off[r + 1]) of p's d_recv (r = this rank); host arrays of world + 1 byte offsets, the sizes
 * agreed by the caller (rank r's piece for p is as long as p expects from r).  Device pointers, ordered on the
 * library's stream.  RCCL: grouped ncclSend / ncclRecv.  SHM: the whole send buffer travels through the rank's
 * slot, so it must fit one (with its offset table).
 */
extern "C" int
ndbhip_comm_alltoallv(const void *d_send, const size_t *send_off, void *d_recv, const size_t *recv_off)
{
	void	   *sv = nullptr;

	if (ndbhip_get_stream(&sv))
		return NDBHIP_ERR_NODEVICE;
	hipStream_t stream = (hipStream_t) sv;

	if (!send_off || !recv_off)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "NULL offset table");
	const int	W = comm.kind == 0 ? 1 : comm.world, me = comm.kind == 0 ? 0 : comm.rank;

	if ((send_off[W] > 0 && !d_send) || (recv_off[W] > 0 && !d_recv))
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "NULL device pointer");
	if (send_off[me + 1] - send_off[me] != recv_off[me + 1] - recv_off[me])
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "a rank's piece for itself has two sizes");
	if (W == 1)
	{
		if (send_off[1] > send_off[0] &&
			hipMemcpyAsync((unsigned char *) d_recv + recv_off[0], (const unsigned char *) d_send + send_off[0],
						   send_off[1] - send_off[0], hipMemcpyDeviceToDevice, stream) != hipSuccess)
			return ndbhip_internal_fail(NDBHIP_ERR_HIP, "hipMemcpyAsync failed");
		return NDBHIP_OK;
	}
	if (comm.kind == 1)
	{
		ncclResult_t r = rccl.GroupStart();

		for (int p = 0; p < W && r == ncclSuccess; p++)
		{
			const size_t ns = send_off[p + 1] - send_off[p], nr = recv_off[p + 1] - recv_off[p];

			if (ns > 0)
				r = rccl.Send((const unsigned char *) d_send + send_off[p], ns, ncclInt8, p, comm.nccl, stream);
			if (r == ncclSuccess && nr > 0)
				r = rccl.Recv((unsigned char *) d_recv + recv_off[p], nr, ncclInt8, p, comm.nccl, stream);
		}
		const ncclResult_t e = rccl.GroupEnd();

		if (r == ncclSuccess)
			r = e;
		if (r != ncclSuccess)
			return ndbhip_internal_fail(NDBHIP_ERR_HIP, "ncclSend/ncclRecv: %s", rccl.GetErrorString(r));
		return NDBHIP_OK;
	}
	/* SHM: slot = [world + 1 offsets][send buffer] */
	const size_t head = (size_t) (W + 1) * sizeof(uint64_t);

	if (head + send_off[W] > comm.slot_bytes)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "exchange of %zu bytes from one rank exceeds the segment's slots (%zu)",
									send_off[W], comm.slot_bytes);
	unsigned char *mine = comm.slots + (size_t) me * comm.slot_bytes;

	for (int p = 0; p <= W; p++)
		((uint64_t *) mine)[p] = (uint64_t) send_off[p];
	if (send_off[W] > 0 &&
		hipMemcpyAsync(mine + head, d_send, send_off[W], hipMemcpyDeviceToHost, stream) != hipSuccess)
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "device-to-host copy failed");
	if (hipStreamSynchronize(stream) != hipSuccess)
		return ndbhip_internal_fail(NDBHIP_ERR_HIP, "stream synchronisation failed");
	pthread_barrier_wait(&comm.hdr->barrier);		/* every slot is written */
	int			rc = NDBHIP_OK;

	for (int p = 0; p < W; p++)
	{
		const unsigned char *theirs = comm.slots + (size_t) p * comm.slot_bytes;
		const uint64_t lo = ((const uint64_t *) theirs)[me], hi = ((const uint64_t *) theirs)[me + 1];
		const size_t nr = recv_off[p + 1] - recv_off[p];

		if (hi - lo != nr)
			rc = ndbhip_internal_fail(NDBHIP_ERR_INVALID, "rank %d sends %llu bytes where %zu are expected", p,
									  (unsigned long long) (hi - lo), nr);
		else if (nr > 0 && hipMemcpyAsync((unsigned char *) d_recv + recv_off[p], theirs + head + lo, nr,
										  hipMemcpyHostToDevice, stream) != hipSuccess)
			rc = ndbhip_internal_fail(NDBHIP_ERR_HIP, "host-to-device copy failed");
	}
	if (hipStreamSynchronize(stream) != hipSuccess && rc == NDBHIP_OK)
		rc = ndbhip_internal_fail(NDBHIP_ERR_HIP, "stream synchronisation failed");
	pthread_barrier_wait(&comm.hdr->barrier);		/* every slot has been read: it may be overwritten */
	return rc;
}

extern "C" int
ndbhip_ivf_search_sharded(ndbhip_ivf *shard, const float *d_queries, int nq, int strategy, int nprobe, int k,
						  int64_t max_candidates, uint64_t *d_out_tids, float *d_out_dist, int *d_out_count)
{
	void	   *sv = nullptr;

	if (ndbhip_get_stream(&sv))
		return NDBHIP_ERR_NODEVICE;
	if (!shard || nq < 0 || (nq > 0 && (!d_queries || !d_out_tids || !d_out_dist || !d_out_count)))
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "bad arguments");
	if (nprobe < 1 || nprobe > NDBHIP_MAX_NPROBE || k < 1 || k > NDBHIP_MAX_K)
		return ndbhip_internal_fail(NDBHIP_ERR_INVALID, "nprobe / k out of range");
	if (nq == 0)
		return NDBHIP_OK;
	const int	world = ndbhip_comm_world(), rank = ndbhip_comm_rank();
	const int	dim = ndbhip_ivf_dim(shard);
	const int	s = (nq + world - 1) / world;			/* padded slice: every rank sends the same count */
	const int	lo = std::min(rank * s, nq), hi = std::min((rank + 1) * s, nq);
	const size_t cap = (size_t) NDBHIP_PARTIAL_CAP(k);
	int			rc;

	if (dim < 1)
		return NDBHIP_ERR_INVALID;
	if ((rc = grow_dev(comm.probes_mine, comm.probes_mine_n, (size_t) s * nprobe)) != 0) return rc;
	if ((rc = grow_dev(comm.probes_all, comm.probes_all_n, (size_t) world * s * nprobe)) != 0) return rc;
	if ((rc = grow_dev(comm.cand, comm.cand_n, (size_t) nq * cap)) != 0) return rc;
	if ((rc = grow_dev(comm.cand_all, comm.cand_all_n, (size_t) world * nq * cap)) != 0) return rc;
	if ((rc = grow_dev(comm.ncand, comm.ncand_n, (size_t) nq)) != 0) return rc;
	if ((rc = grow_dev(comm.ncand_all, comm.ncand_all_n, (size_t) world * nq)) != 0) return rc;
	if ((rc = grow_dev(comm.total, comm.total_n, (size_t) nq)) != 0) return rc;

	/* 1. cluster selection, split by queries */
	if (hi > lo)
	{
		rc = ndbhip_ivf_select_clusters_device(shard, d_queries + (size_t) lo * dim, hi - lo, nprobe, comm.probes_mine);
		if (rc)
			return rc;
	}
	rc = ndbhip_comm_allgather(comm.probes_mine, comm.probes_all, (size_t) s * nprobe * sizeof(int));
	if (rc)
		return rc;
	/* 2. this shard's lists for all queries; the queries' first thresholds are exchanged on the way (minimum over the
	 * ranks: every rank then sweeps against the bound of the rank that holds the query's own list) */
	ndbhip_internal_set_thr_hook(world > 1 ? ndbhip_comm_allreduce_min_f32 : nullptr);
	rc = ndbhip_ivf_search_partial_probes_device(shard, d_queries, nq, strategy, nprobe, k, max_candidates,
												 comm.probes_all, comm.cand, comm.ncand, comm.total);
	ndbhip_internal_set_thr_hook(nullptr);
	if (rc)
		return rc;
	/* 3. records of every rank, then the replay merge */
	rc = ndbhip_comm_allgather(comm.cand, comm.cand_all, (size_t) nq * cap * sizeof(ndbhip_cand));
	if (rc)
		return rc;
	rc = ndbhip_comm_allgather(comm.ncand, comm.ncand_all, (size_t) nq * sizeof(int));
	if (rc)
		return rc;
	return ndbhip_merge_topk_device(comm.cand_all, comm.ncand_all, comm.total, world, nq, k, (int) cap,
									d_out_tids, d_out_dist, d_out_count);
}
