// pbdx_device.h -- selecting a solver's HIP device for the duration of one entry point.
#ifndef PBDX_DEVICE_H
#define PBDX_DEVICE_H
#include <hip/hip_runtime.h>
#include <mutex>

// One host thread at a time inside the library per DEVICE (process-wide, re-entrant).  Two engines that share a device and are driven from two host threads
// -- an ensemble with a device listed more than once, the plug-in's helper thread next to a second controller -- otherwise meet inside the runtime: while one
// thread captures a substep into a hipGraph, another thread's device-wide synchronisation or null-stream copy is "not permitted when stream is capturing"
// and poisons the capture (soak of round 6: 8 of 12 runs of the GPU suite lost `test_single_process_ensemble_with_one_device_listed_eight_times` that way,
// profiles/HISTORY.md [10]).  Engines on different devices do not share a mutex and run concurrently as before; on one device the work of two engines
// was serialised by the hardware anyway (the persistent launch needs every CU).
namespace pbdx { std::recursive_mutex &device_mutex(int device); }

namespace {
// Every entry point runs on ITS solver's device and leaves the calling thread's current HIP device as it found it: a host that
// keeps a device of its own current (torch's rank device, a second solver on another GPU of the same process) is not disturbed by
// a call into the engine (tests/test_distributed.py: two solvers on two devices in one process).
struct DeviceScope
{
	int prev = -1; bool switched = false; hipError_t err = hipSuccess;
	std::recursive_mutex &mutex;
	explicit DeviceScope(int device) : mutex(pbdx::device_mutex(device))
	{
		mutex.lock();
		err = hipGetDevice(&prev);
		if (err == hipSuccess && prev != device) { err = hipSetDevice(device); switched = (err == hipSuccess); }
	}
	~DeviceScope() { if (switched) (void)hipSetDevice(prev); mutex.unlock(); }
	DeviceScope(const DeviceScope &) = delete;
	DeviceScope &operator=(const DeviceScope &) = delete;
};
#define ENTER_DEVICE(device) DeviceScope device_scope_(device); HIPCHECK(device_scope_.err)
}

// pbdx_hostio.hip: synchronous copies between host memory the library does not own and the CURRENT device, through the library's page-locked
// bounce buffer -- instead of hipMemcpy on the caller's pointer, which has the runtime map the caller's heap pages into the GPU's address space
namespace pbdx {
hipError_t copy_to_device(void *dst, const void *src, size_t bytes);
hipError_t copy_from_device(void *dst, const void *src, size_t bytes);
// asynchronous copies whose host side lies inside page-locked memory THE LIBRARY owns (pinned_alloc: checked, hipErrorInvalidValue otherwise)
hipError_t copy_pinned_to_device_async(void *dst, const void *src, size_t bytes, hipStream_t stream);
hipError_t copy_device_to_pinned_async(void *dst, const void *src, size_t bytes, hipStream_t stream);
}

#endif
