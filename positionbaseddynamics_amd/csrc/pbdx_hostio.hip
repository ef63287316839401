// Synchronous copies between HOST memory the engine does not own (caller arrays, std::vectors on the heap) and the device, through a
// page-locked bounce buffer the library owns.
//
// Why not hipMemcpy on the caller's pointer: for anything but small sizes the runtime page-locks the caller's pages for the duration of the
// copy, i.e. maps a piece of the host's heap into the GPU's address space at its host address, keeps such mappings in a cache, and their
// validity then depends on what the host's allocator does with that heap afterwards.  A GPU test run of round 5 ended in "Memory access fault
// by GPU" at a heap address (profiles/HISTORY.md [9]); since then no address of memory the library does not own is handed to the GPU: the
// device copies to / from the bounce buffer, host threads copy between the bounce buffer and the caller's memory, one half of the buffer while
// the other half is on the bus.  (Particle transfers with PBDX_OPT_PIN_HOST have a mirror of their own per engine: pbdx_solver.hip.)
#include <hip/hip_runtime.h>
#include "pbdx_internal.h"
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <string.h>
#include <unistd.h>
#include <dlfcn.h>

namespace pbdx {

void host_copy(void *dst, const void *src, size_t bytes);

// memcpy by a team of worker threads that is started on first use and stays (a thread start per 8 MiB half costs more than the half's copy).
// One job at a time; the calling thread works too.  All job state is guarded by one mutex, and a job's fields change only when the previous
// job is complete, so a worker that wakes late finds nothing to take.  After fork() the child has no workers: it copies on its own thread.
namespace {
class CopyTeam
{
public:
	explicit CopyTeam(unsigned workers) : pid_(getpid())
	{
		// the workers run code of this library until the process ends: a host that dlclose()s the library (or a plug-in linked to it) must not
		// take their code away -- pin the library in memory
		Dl_info info;
		if (dladdr(reinterpret_cast<const void *>(&pbdx::host_copy), &info) && info.dli_fname) (void)dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD | RTLD_NODELETE);
		for (unsigned t = 0; t < workers; t++) std::thread([this]() { work(); }).detach();
	}
	void run(char *dst, const char *src, size_t bytes, size_t per)
	{
		std::lock_guard<std::mutex> one_job(jobs_);
		std::unique_lock<std::mutex> lock(m_);
		dst_ = dst; src_ = src; bytes_ = bytes; per_ = per; parts_ = (unsigned)((bytes + per - 1) / per); next_ = 0; done_ = 0;
		wake_.notify_all();
		while (next_ < parts_) take(lock);
		finished_.wait(lock, [this]() { return done_ == parts_; });
		parts_ = 0; next_ = 0;
	}
	pid_t pid() const { return pid_; }
private:
	void take(std::unique_lock<std::mutex> &lock)
	{
		const size_t lo = per_ * next_++, hi = std::min(bytes_, lo + per_);
		char *d = dst_; const char *s = src_;
		lock.unlock();
		memcpy(d + lo, s + lo, hi - lo);
		lock.lock();
		if (++done_ == parts_) finished_.notify_all();
	}
	void work()
	{
		std::unique_lock<std::mutex> lock(m_);
		for (;;)
		{
			wake_.wait(lock, [this]() { return next_ < parts_; });
			take(lock);
		}
	}
	std::mutex jobs_, m_;
	std::condition_variable wake_, finished_;
	char *dst_ = nullptr; const char *src_ = nullptr;
	size_t bytes_ = 0, per_ = 0;
	unsigned parts_ = 0, next_ = 0, done_ = 0;
	const pid_t pid_;
};
} // namespace

#ifndef PBDX_COPY_TEAM
#define PBDX_COPY_TEAM 1        // 0: threads started per call (A/B builds: scripts/build_variant.sh spawn -DPBDX_COPY_TEAM=0)
#endif

void host_copy(void *dst, const void *src, size_t bytes)
{
#if !PBDX_COPY_TEAM
	{
		const size_t kSlice = (size_t)1 << 20;
		unsigned threads = (unsigned)std::min<size_t>(bytes / kSlice, 16);
		const unsigned hw = std::thread::hardware_concurrency();
		if (hw && threads > hw) threads = hw;
		if (threads < 2) { memcpy(dst, src, bytes); return; }
		std::vector<std::thread> team;
		team.reserve(threads - 1);
		const size_t per = ((bytes + threads - 1) / threads + 63) & ~(size_t)63;
		for (unsigned t = 1; t < threads; t++)
		{
			const size_t lo = std::min(bytes, per * t), hi = std::min(bytes, per * (t + 1));
			if (hi > lo) team.emplace_back([=]() { memcpy(static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, hi - lo); });
		}
		memcpy(dst, src, std::min(bytes, per));
		for (std::thread &t : team) t.join();
		return;
	}
#endif
	const size_t kPart = (size_t)256 << 10;
	if (bytes < 4 * kPart) { memcpy(dst, src, bytes); return; }
	static CopyTeam *team = []()
	{
		const unsigned hw = std::thread::hardware_concurrency();
		return new CopyTeam(std::max(1u, std::min(hw ? hw / 2 : 4u, 16u)) - 0u);         // never destroyed: the workers sleep until the process ends
	}();
	if (team->pid() != getpid()) { memcpy(dst, src, bytes); return; }                  // (forked child)
	team->run(static_cast<char *>(dst), static_cast<const char *>(src), bytes, kPart);
}

namespace {

constexpr size_t kHalf = (size_t)8 << 20;
constexpr int kMaxDevices = 64;
struct Bounce
{
	std::mutex mutex;
	char *buf = nullptr;                 // two halves
	hipStream_t stream = nullptr;        // (blocking stream: ordered against the null stream like hipMemcpy)
	hipEvent_t ev[2] = { nullptr, nullptr };
};
Bounce g_bounce[kMaxDevices];

hipError_t prepare(Bounce &b)
{
	if (b.buf) return hipSuccess;
	hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&b.buf), 2 * kHalf, hipHostMallocDefault);
	if (e == hipSuccess) e = hipStreamCreate(&b.stream);
	for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&b.ev[i], hipEventDisableTiming);
	if (e != hipSuccess)
	{
		if (b.buf) { (void)hipHostFree(b.buf); b.buf = nullptr; }
		if (b.stream) { (void)hipStreamDestroy(b.stream); b.stream = nullptr; }
		for (hipEvent_t &v : b.ev) if (v) { (void)hipEventDestroy(v); v = nullptr; }
	}
	return e;
}

} // namespace

std::recursive_mutex &device_mutex(int device)
{
	static std::recursive_mutex m[kMaxDevices + 1];
	return m[device >= 0 && device < kMaxDevices ? device : kMaxDevices];
}

// dst (device) := src (host), complete on return.  Current device.
hipError_t copy_to_device(void *dst, const void *src, size_t bytes)
{
	if (!bytes) return hipSuccess;
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
	Bounce &b = g_bounce[dev];
	std::lock_guard<std::mutex> lock(b.mutex);
	if ((e = prepare(b)) != hipSuccess) return e;
	if ((e = hipDeviceSynchronize()) != hipSuccess) return e;          // (what hipMemcpy on the null stream waits for)
	// (an error in the middle: copies already enqueued still read / write the halves -- they are waited for before the mutex lets the next caller at them)
	auto fail = [&](hipError_t err) { (void)hipStreamSynchronize(b.stream); return err; };
	bool busy[2] = { false, false };
	int i = 0;
	for (size_t off = 0; off < bytes; off += kHalf, i ^= 1)
	{
		const size_t n = std::min(kHalf, bytes - off);
		if (busy[i] && (e = hipEventSynchronize(b.ev[i])) != hipSuccess) return fail(e);
		host_copy(b.buf + kHalf * i, static_cast<const char *>(src) + off, n);
		if ((e = hipMemcpyAsync(static_cast<char *>(dst) + off, b.buf + kHalf * i, n, hipMemcpyHostToDevice, b.stream)) != hipSuccess) return fail(e);
		if ((e = hipEventRecord(b.ev[i], b.stream)) != hipSuccess) return fail(e);
		busy[i] = true;
	}
	return hipStreamSynchronize(b.stream);
}

// dst (host) := src (device), complete on return.  Current device.
hipError_t copy_from_device(void *dst, const void *src, size_t bytes)
{
	if (!bytes) return hipSuccess;
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
	Bounce &b = g_bounce[dev];
	std::lock_guard<std::mutex> lock(b.mutex);
	if ((e = prepare(b)) != hipSuccess) return e;
	if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
	auto fail = [&](hipError_t err) { (void)hipStreamSynchronize(b.stream); return err; };
	// chunk k is on the bus while chunk k - 1 leaves the buffer
	const size_t chunks = (bytes + kHalf - 1) / kHalf;
	for (size_t k = 0; k <= chunks; k++)
	{
		if (k < chunks)
		{
			const size_t off = k * kHalf, n = std::min(kHalf, bytes - off);
			if ((e = hipMemcpyAsync(b.buf + kHalf * (k & 1), static_cast<const char *>(src) + off, n, hipMemcpyDeviceToHost, b.stream)) != hipSuccess) return fail(e);
			if ((e = hipEventRecord(b.ev[k & 1], b.stream)) != hipSuccess) return fail(e);
		}
		if (k > 0)
		{
			const size_t off = (k - 1) * kHalf, n = std::min(kHalf, bytes - off);
			if ((e = hipEventSynchronize(b.ev[(k - 1) & 1])) != hipSuccess) return fail(e);
			host_copy(static_cast<char *>(dst) + off, b.buf + kHalf * ((k - 1) & 1), n);
		}
	}
	return hipSuccess;
}

} // namespace pbdx

extern "C" void pbdx_debug_host_copy(void *dst, const void *src, uint64_t bytes)
{
	if (dst && src && bytes) pbdx::host_copy(dst, src, (size_t)bytes);
}

// ---- page-locked memory THE LIBRARY owns (pbdx_model's particle arrays) ----------------------------------------------------------------------
// The host mirror of the reference's ParticleData (pbdx_model: mass, inverse mass, x0, x, v, a, oldX, lastX) is the library's own memory, so it may be
// page-locked and handed to the copy engine as it is -- which is what the reference's python binding does with its own arrays (zero-copy getVertices,
// pyPBD/ParticleDataModule.cpp:54-58).  A step of the host-in / host-out contract then moves its 56 + 48 MB at the bus rate and no host thread touches
// them (round 5: 3.6-3.8 ms per step through a mirror and a team of copying threads, 4.8-5.1 through the bounce buffer).  Allocations are registered by
// address range; the direct copies below REFUSE any address that is not inside one -- a caller's pointer still cannot reach the GPU by address.
namespace pbdx {
namespace {
std::mutex g_pinned_mu;
std::vector<std::pair<uintptr_t, size_t>> &pinned_ranges() { static std::vector<std::pair<uintptr_t, size_t>> r; return r; }
bool g_have_device_known = false, g_have_device = false;
bool have_device()
{
	if (!g_have_device_known)
	{
		int n = 0;
		g_have_device = hipGetDeviceCount(&n) == hipSuccess && n > 0;
		if (!g_have_device) (void)hipGetLastError();
		g_have_device_known = true;
	}
	return g_have_device;
}
}
void *pinned_alloc(size_t bytes)
{
	if (!bytes) return nullptr;
	void *p = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_pinned_mu);
		if (!getenv("PBDX_NO_PINNED_MODEL") && have_device() && hipHostMalloc(&p, bytes, hipHostMallocPortable) == hipSuccess && p)
		{
			pinned_ranges().push_back({ (uintptr_t)p, bytes });
			return p;
		}
	}
	(void)hipGetLastError();
	return malloc(bytes);
}
void pinned_free(void *p)
{
	if (!p) return;
	{
		std::lock_guard<std::mutex> lk(g_pinned_mu);
		auto &r = pinned_ranges();
		for (size_t i = 0; i < r.size(); i++)
			if (r[i].first == (uintptr_t)p)
			{
				r[i] = r.back(); r.pop_back();
				(void)hipHostFree(p);
				return;
			}
	}
	free(p);
}
bool is_library_pinned(const void *p, size_t bytes)
{
	if (!p) return false;
	std::lock_guard<std::mutex> lk(g_pinned_mu);
	for (const auto &r : pinned_ranges())
		if ((uintptr_t)p >= r.first && (uintptr_t)p + bytes <= r.first + r.second) return true;
	return false;
}
// asynchronous copies on `stream` with a host side inside the library's own page-locked memory (checked); hipErrorInvalidValue otherwise
hipError_t copy_pinned_to_device_async(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
	if (!is_library_pinned(src, bytes)) return hipErrorInvalidValue;
	return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
}
hipError_t copy_device_to_pinned_async(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
	if (!is_library_pinned(dst, bytes)) return hipErrorInvalidValue;
	return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
}
} // namespace pbdx
