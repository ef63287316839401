// pbdx_comm.cpp -- the collective of a MULTI-PROCESS host (one process per GPU) without torch: RCCL over xGMI, loaded at run time.
//
// the ensemble of independent instances is the only thing that shards, and it needs no collective on the data path.  What the ranks of
// `bench.py --gpus N` exchange through torch.distributed (backend nccl = RCCL) is control data: a barrier, the maximum of the per-rank times, the sum of
// the projection counts, the checksums.  A C or C++ host with one process per GPU has no torch; this file gives it the same four operations on the same
// library: librccl.so is opened with dlopen when a communicator is asked for (libpbdx.so does NOT link it: a single-GPU host never loads it), the
// 128-byte unique id of rank 0 travels to the other ranks by whatever channel the host has (a file, MPI, its own sockets), every rank calls
// pbdx_comm_create with its HIP device.  Values cross the boundary by value (host arrays in, host arrays out); the device staging buffer is the
// communicator's own.  No reference counterpart (Simulation/TimeStepController.cpp:75-241 is one process on one model).
#include "pbdx_internal.h"
#include "pbdx_device.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <string.h>
#include <mutex>

using namespace pbdx;

namespace {

// the part of RCCL's C API this file uses (rccl.h: NCCL_UNIQUE_ID_BYTES 128, ncclUint64 5, ncclFloat64 8, ncclSum 0, ncclMax 2)
struct RcclId { char internal[128]; };
typedef int (*fn_get_id)(RcclId *);
typedef int (*fn_init_rank)(void **comm, int nranks, RcclId id, int rank);
typedef int (*fn_all_reduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream);
typedef int (*fn_all_gather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t stream);
typedef int (*fn_destroy)(void *comm);
typedef const char *(*fn_error_string)(int);
constexpr int kUint64 = 5, kFloat64 = 8, kSum = 0, kMax = 2;

struct Rccl
{
	void *handle = nullptr;
	fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_all_reduce all_reduce = nullptr; fn_all_gather all_gather = nullptr;
	fn_destroy destroy = nullptr; fn_error_string error_string = nullptr;
	std::string why;
};
Rccl &rccl()
{
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		const char *names[] = { getenv("PBDX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
		for (const char *n : names)
		{
			if (!n || !*n) continue;
			r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
			if (r.handle) break;
			r.why = dlerror();
		}
		if (!r.handle) return;
		r.get_id = (fn_get_id)dlsym(r.handle, "ncclGetUniqueId");
		r.init_rank = (fn_init_rank)dlsym(r.handle, "ncclCommInitRank");
		r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
		r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
		r.destroy = (fn_destroy)dlsym(r.handle, "ncclCommDestroy");
		r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
		if (!r.get_id || !r.init_rank || !r.all_reduce || !r.all_gather || !r.destroy)
		{
			r.why = "librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclAllGather / ncclCommDestroy";
			dlclose(r.handle); r.handle = nullptr;
		}
	});
	return r;
}
int rccl_fail(const char *what, int rc)
{
	Rccl &r = rccl();
	set_error("%s failed: %s (RCCL result %d)", what, r.error_string ? r.error_string(rc) : "?", rc);
	return PBDX_ERR_HIP;
}

} // namespace

struct pbdx_comm
{
	void *comm = nullptr;
	int device = 0, world = 1, rank = 0;
	hipStream_t stream = nullptr;
	char *d_buf = nullptr;           // staging: send at 0, receive behind it
	size_t buf_bytes = 0;
	char *h_mir = nullptr;           // page-locked mirror of the staging buffer (no caller address reaches the GPU: pbdx_hostio.hip)
};

#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
	set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return PBDX_ERR_HIP; } } while (0)

namespace {
int ensure_buffers(pbdx_comm *c, size_t bytes)
{
	if (bytes <= c->buf_bytes) return PBDX_OK;
	if (c->d_buf) { (void)hipFree(c->d_buf); c->d_buf = nullptr; }
	if (c->h_mir) { (void)hipHostFree(c->h_mir); c->h_mir = nullptr; }
	c->buf_bytes = 0;
	const size_t want = (bytes + 4095) & ~(size_t)4095;
	HIPCHECK(hipMalloc(reinterpret_cast<void **>(&c->d_buf), want));
	HIPCHECK(hipHostMalloc(reinterpret_cast<void **>(&c->h_mir), want, hipHostMallocDefault));
	c->buf_bytes = want;
	return PBDX_OK;
}
// send_bytes from `in` -> collective -> recv_bytes into `out` (both host arrays of the caller; the copies go through the communicator's own mirror)
template <class Op> int round_trip(pbdx_comm *c, const void *in, size_t send_bytes, void *out, size_t recv_bytes, Op &&op)
{
	if (!c || !c->comm) { set_error("pbdx_comm: no communicator"); return PBDX_ERR_INVALID; }
	ENTER_DEVICE(c->device);
	int r = ensure_buffers(c, send_bytes + recv_bytes);
	if (r) return r;
	memcpy(c->h_mir, in, send_bytes);
	HIPCHECK(hipMemcpyAsync(c->d_buf, c->h_mir, send_bytes, hipMemcpyHostToDevice, c->stream));
	const int rc = op(c->d_buf, c->d_buf + send_bytes);
	if (rc) return rccl_fail("RCCL collective", rc);
	HIPCHECK(hipMemcpyAsync(c->h_mir + send_bytes, c->d_buf + send_bytes, recv_bytes, hipMemcpyDeviceToHost, c->stream));
	HIPCHECK(hipStreamSynchronize(c->stream));
	memcpy(out, c->h_mir + send_bytes, recv_bytes);
	return PBDX_OK;
}
} // namespace

extern "C" {

int pbdx_comm_available(void) { return rccl().handle ? 1 : 0; }

int pbdx_comm_unique_id(void *id, size_t bytes)
{
	if (!id || bytes < PBDX_COMM_ID_BYTES) { set_error("pbdx_comm_unique_id: the id takes %d bytes", PBDX_COMM_ID_BYTES); return PBDX_ERR_INVALID; }
	Rccl &r = rccl();
	if (!r.handle) { set_error("pbdx_comm: RCCL is not available (%s)", r.why.c_str()); return PBDX_ERR_UNSUPPORTED; }
	RcclId u;
	memset(&u, 0, sizeof(u));
	const int rc = r.get_id(&u);
	if (rc) return rccl_fail("ncclGetUniqueId", rc);
	memcpy(id, &u, sizeof(u));
	return PBDX_OK;
}

int pbdx_comm_create(pbdx_comm **out, const void *id, size_t bytes, int world, int rank, int device)
{
	if (!out || !id || bytes < PBDX_COMM_ID_BYTES || world < 1 || rank < 0 || rank >= world) { set_error("pbdx_comm_create: bad arguments"); return PBDX_ERR_INVALID; }
	*out = nullptr;
	Rccl &r = rccl();
	if (!r.handle) { set_error("pbdx_comm: RCCL is not available (%s)", r.why.c_str()); return PBDX_ERR_UNSUPPORTED; }
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { (void)hipGetLastError(); set_error("pbdx_comm_create: no HIP device %d", device); return PBDX_ERR_NO_DEVICE; }
	pbdx_comm *c = new (std::nothrow) pbdx_comm();
	if (!c) { set_error("out of memory"); return PBDX_ERR_ALLOC; }
	c->device = device; c->world = world; c->rank = rank;
	DeviceScope scope(device);
	if (scope.err != hipSuccess) { set_error("pbdx_comm_create: cannot select device %d: %s", device, hipGetErrorString(scope.err)); delete c; return PBDX_ERR_HIP; }
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { set_error("pbdx_comm_create: no stream: %s", hipGetErrorString(hipGetLastError())); delete c; return PBDX_ERR_HIP; }
	RcclId u;
	memcpy(&u, id, sizeof(u));
	const int rc = r.init_rank(&c->comm, world, u, rank);      // (RCCL binds the communicator to the CURRENT device: selected above)
	if (rc) { (void)hipStreamDestroy(c->stream); delete c; return rccl_fail("ncclCommInitRank", rc); }
	*out = c;
	return PBDX_OK;
}

void pbdx_comm_destroy(pbdx_comm *c)
{
	if (!c) return;
	DeviceScope scope(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->comm && rccl().destroy) (void)rccl().destroy(c->comm);
	if (c->d_buf) (void)hipFree(c->d_buf);
	if (c->h_mir) (void)hipHostFree(c->h_mir);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

int pbdx_comm_world(const pbdx_comm *c) { return c ? c->world : 0; }
int pbdx_comm_rank(const pbdx_comm *c) { return c ? c->rank : -1; }

int pbdx_comm_all_reduce_sum_u64(pbdx_comm *c, uint64_t *values, uint32_t n)
{
	if (!n) return PBDX_OK;
	if (!values) return PBDX_ERR_INVALID;
	return round_trip(c, values, (size_t)n * 8, values, (size_t)n * 8, [&](void *s, void *d) { return rccl().all_reduce(s, d, n, kUint64, kSum, c->comm, c->stream); });
}
int pbdx_comm_all_reduce_max_f64(pbdx_comm *c, double *values, uint32_t n)
{
	if (!n) return PBDX_OK;
	if (!values) return PBDX_ERR_INVALID;
	return round_trip(c, values, (size_t)n * 8, values, (size_t)n * 8, [&](void *s, void *d) { return rccl().all_reduce(s, d, n, kFloat64, kMax, c->comm, c->stream); });
}
int pbdx_comm_all_gather_u64(pbdx_comm *c, const uint64_t *mine, uint32_t n, uint64_t *all)
{
	if (!n) return PBDX_OK;
	if (!mine || !all || !c) return PBDX_ERR_INVALID;
	return round_trip(c, mine, (size_t)n * 8, all, (size_t)n * 8 * (size_t)c->world, [&](void *s, void *d) { return rccl().all_gather(s, d, n, kUint64, c->comm, c->stream); });
}
int pbdx_comm_barrier(pbdx_comm *c)
{
	uint64_t one = 1;
	const int r = pbdx_comm_all_reduce_sum_u64(c, &one, 1);
	if (r) return r;
	if (c && one != (uint64_t)c->world) { set_error("pbdx_comm_barrier: %llu of %d ranks arrived", (unsigned long long)one, c->world); return PBDX_ERR_HIP; }
	return PBDX_OK;
}

} // extern "C"
