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
#include <mutex>
#include <thread>
#include <vector>
#include <string.h>

namespace pbdx {

void host_copy(void *dst, const void *src, size_t bytes)
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
	bool busy[2] = { false, false };
	int i = 0;
	for (size_t off = 0; off < bytes; off += kHalf, i ^= 1)
	{
		const size_t n = std::min(kHalf, bytes - off);
		if (busy[i] && (e = hipEventSynchronize(b.ev[i])) != hipSuccess) return e;
		host_copy(b.buf + kHalf * i, static_cast<const char *>(src) + off, n);
		if ((e = hipMemcpyAsync(static_cast<char *>(dst) + off, b.buf + kHalf * i, n, hipMemcpyHostToDevice, b.stream)) != hipSuccess) return e;
		if ((e = hipEventRecord(b.ev[i], b.stream)) != hipSuccess) return e;
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
	// chunk k is on the bus while chunk k - 1 leaves the buffer
	const size_t chunks = (bytes + kHalf - 1) / kHalf;
	for (size_t k = 0; k <= chunks; k++)
	{
		if (k < chunks)
		{
			const size_t off = k * kHalf, n = std::min(kHalf, bytes - off);
			if ((e = hipMemcpyAsync(b.buf + kHalf * (k & 1), static_cast<const char *>(src) + off, n, hipMemcpyDeviceToHost, b.stream)) != hipSuccess) return e;
			if ((e = hipEventRecord(b.ev[k & 1], b.stream)) != hipSuccess) return e;
		}
		if (k > 0)
		{
			const size_t off = (k - 1) * kHalf, n = std::min(kHalf, bytes - off);
			if ((e = hipEventSynchronize(b.ev[(k - 1) & 1])) != hipSuccess) return e;
			host_copy(static_cast<char *>(dst) + off, b.buf + kHalf * ((k - 1) & 1), n);
		}
	}
	return hipSuccess;
}

} // namespace pbdx
