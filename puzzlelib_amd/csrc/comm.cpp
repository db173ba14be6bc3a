// Data-parallel exchange over RCCL (xGMI): gradient all-reduce and parameter broadcast, one communicator per process
// (one process per GPU). Replaces the IPC star of NodeInfo.{sumTensor,broadcastBuffer} — Grid.py:54-63,103-157 and
// Cuda/Source/Core/Buffer.c:61-98 (cudaIpc*MemHandle). librccl is loaded lazily with dlopen so that single-GPU users
// (and the CPU-only build check) never need it; a missing library is a loud PZ_ERR_COMM, not a fallback.
#include "common.h"

#include <dlfcn.h>
#include <chrono>
#include <thread>

namespace {

// minimal RCCL/NCCL ABI surface (stable C ABI; values from the public nccl.h)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclFloat32 = 7, ncclUint8 = 1, ncclSum = 0 };

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
	ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
	ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
};

Rccl g_rccl;

int load_rccl() {
	if (g_rccl.lib) return PZ_OK;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
	void *lib = nullptr;
	for (const char *n : names)
		if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
	if (!lib) {
		pz::set_error("cannot load librccl: %s", dlerror());
		return PZ_ERR_COMM;
	}

#define PZ_SYM(field, name)                                             \
	*(void **)(&g_rccl.field) = dlsym(lib, name);                       \
	if (!g_rccl.field) {                                                \
		pz::set_error("librccl lacks symbol %s", name);                 \
		return PZ_ERR_COMM;                                             \
	}
	PZ_SYM(GetUniqueId, "ncclGetUniqueId")
	PZ_SYM(CommInitRank, "ncclCommInitRank")
	PZ_SYM(CommDestroy, "ncclCommDestroy")
	PZ_SYM(AllReduce, "ncclAllReduce")
	PZ_SYM(Broadcast, "ncclBroadcast")
	PZ_SYM(GetErrorString, "ncclGetErrorString")
	PZ_SYM(CommGetAsyncError, "ncclCommGetAsyncError")
	PZ_SYM(CommAbort, "ncclCommAbort")
	PZ_SYM(CommCount, "ncclCommCount")
	PZ_SYM(CommUserRank, "ncclCommUserRank")
	PZ_SYM(GroupStart, "ncclGroupStart")
	PZ_SYM(GroupEnd, "ncclGroupEnd")
#undef PZ_SYM

	g_rccl.lib = lib;
	return PZ_OK;
}

}  // namespace

struct pz_comm {
	ncclComm_t comm;
	int nranks, rank;
};

#define PZ_NCCL(call)                                                                                \
	do {                                                                                             \
		ncclResult_t r_ = (call);                                                                    \
		if (r_ != 0) {                                                                               \
			pz::set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
			return PZ_ERR_COMM;                                                                      \
		}                                                                                            \
	} while (0)

extern "C" {

int pz_comm_unique_id(char id[PZ_COMM_ID_BYTES]) {
	if (int rc = load_rccl()) return rc;
	static_assert(sizeof(ncclUniqueId) == PZ_COMM_ID_BYTES, "unique id size");
	ncclUniqueId uid;
	PZ_NCCL(g_rccl.GetUniqueId(&uid));
	memcpy(id, &uid, sizeof(uid));
	return PZ_OK;
}

int pz_comm_init_rank(pz_comm_t *comm, int nranks, const char id[PZ_COMM_ID_BYTES], int rank) {
	PZ_REQUIRE(comm != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "pz_comm_init_rank: bad arguments");
	if (int rc = load_rccl()) return rc;
	ncclUniqueId uid;
	memcpy(&uid, id, sizeof(uid));
	ncclComm_t c;
	PZ_NCCL(g_rccl.CommInitRank(&c, nranks, uid, rank));
	*comm = new pz_comm{c, nranks, rank};
	return PZ_OK;
}

int pz_comm_probe(void) {
	return load_rccl();
}

int pz_comm_info(pz_comm_t comm, int *nranks, int *rank) {
	// what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank), not what the caller asked for
	PZ_REQUIRE(comm != nullptr && comm->comm && nranks && rank, "pz_comm_info: null argument or aborted communicator");
	PZ_NCCL(g_rccl.CommCount(comm->comm, nranks));
	PZ_NCCL(g_rccl.CommUserRank(comm->comm, rank));
	return PZ_OK;
}

int pz_comm_async_error(pz_comm_t comm) {
	PZ_REQUIRE(comm != nullptr, "pz_comm_async_error: null communicator");
	if (!comm->comm) {
		pz::set_error("RCCL communicator of rank %d/%d was aborted", comm->rank, comm->nranks);
		return PZ_ERR_COMM;
	}
	ncclResult_t state = 0;
	PZ_NCCL(g_rccl.CommGetAsyncError(comm->comm, &state));
	if (state != 0 && state != 7 /* ncclInProgress */) {
		pz::set_error("RCCL communicator of rank %d/%d reports an asynchronous error: %s", comm->rank, comm->nranks,
		              g_rccl.GetErrorString(state));
		return PZ_ERR_COMM;
	}
	return PZ_OK;
}

int pz_comm_wait_event(pz_comm_t comm, pz_event_t event, double timeout_s) {
	PZ_REQUIRE(comm != nullptr && event != nullptr, "pz_comm_wait_event: null argument");
	if (!comm->comm) {
		pz::set_error("RCCL communicator of rank %d/%d was aborted", comm->rank, comm->nranks);
		return PZ_ERR_COMM;
	}
	const auto t0 = std::chrono::steady_clock::now();
	for (;;) {
		const hipError_t rc = hipEventQuery((hipEvent_t)event);
		if (rc == hipSuccess) return PZ_OK;
		if (rc != hipErrorNotReady) PZ_HIP(rc);
		if (int err = pz_comm_async_error(comm)) return err;
		const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (timeout_s > 0.0 && waited > timeout_s) {
			// a collective some rank never joined would otherwise block the stream (and every later sync) for good
			g_rccl.CommAbort(comm->comm);
			comm->comm = nullptr;
			pz::set_error("RCCL collective of rank %d/%d did not complete within %.1f s: communicator aborted", comm->rank,
			              comm->nranks, timeout_s);
			return PZ_ERR_COMM;
		}
		std::this_thread::sleep_for(std::chrono::microseconds(50));
	}
}

int pz_comm_destroy(pz_comm_t comm) {
	if (!comm) return PZ_OK;
	if (comm->comm) PZ_NCCL(g_rccl.CommDestroy(comm->comm));
	delete comm;
	return PZ_OK;
}

int pz_comm_allreduce_sum_f32(pz_comm_t comm, const float *send, float *recv, size_t count, pz_stream_t stream) {
	PZ_REQUIRE(comm != nullptr && comm->comm && send && recv, "pz_comm_allreduce_sum_f32: null argument or aborted communicator");
	if (count == 0) return PZ_OK;
	PZ_NCCL(g_rccl.AllReduce(send, recv, count, ncclFloat32, ncclSum, comm->comm, pz::as_stream(stream)));
	return PZ_OK;
}

int pz_comm_allreduce_sum_f32_ranges(pz_comm_t comm, float *base, const size_t *offsets, const size_t *counts, int nranges,
                                     pz_stream_t stream) {
	PZ_REQUIRE(comm != nullptr && comm->comm && base && offsets && counts && nranges >= 0,
	           "pz_comm_allreduce_sum_f32_ranges: null argument or aborted communicator");
	if (nranges == 0) return PZ_OK;
	// one group: RCCL batches the ranges' collectives into as few launches as it can, every rank lists the same ranges in the same order
	PZ_NCCL(g_rccl.GroupStart());
	for (int i = 0; i < nranges; ++i) {
		if (counts[i] == 0) continue;
		const ncclResult_t r = g_rccl.AllReduce(base + offsets[i], base + offsets[i], counts[i], ncclFloat32, ncclSum, comm->comm,
		                                        pz::as_stream(stream));
		if (r != 0) {
			g_rccl.GroupEnd();
			pz::set_error("ncclAllReduce (range %d of %d) failed: %s", i, nranges, g_rccl.GetErrorString(r));
			return PZ_ERR_COMM;
		}
	}
	PZ_NCCL(g_rccl.GroupEnd());
	return PZ_OK;
}

int pz_comm_broadcast(pz_comm_t comm, void *buf, size_t nbytes, int root, pz_stream_t stream) {
	PZ_REQUIRE(comm != nullptr && comm->comm && buf, "pz_comm_broadcast: null argument or aborted communicator");
	if (nbytes == 0) return PZ_OK;
	PZ_NCCL(g_rccl.Broadcast(buf, buf, nbytes, ncclUint8, root, comm->comm, pz::as_stream(stream)));
	return PZ_OK;
}

}  // extern "C"
