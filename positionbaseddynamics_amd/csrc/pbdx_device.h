// pbdx_device.h -- selecting a solver's HIP device for the duration of one entry point.
#ifndef PBDX_DEVICE_H
#define PBDX_DEVICE_H
#include <hip/hip_runtime.h>

namespace {
// Every entry point runs on ITS solver's device and leaves the calling thread's current HIP device as it found it: a host that
// keeps a device of its own current (torch's rank device, a second solver on another GPU of the same process) is not disturbed by
// a call into the engine (tests/test_distributed.py: two solvers on two devices in one process).
struct DeviceScope
{
	int prev = -1; bool switched = false; hipError_t err = hipSuccess;
	explicit DeviceScope(int device)
	{
		err = hipGetDevice(&prev);
		if (err == hipSuccess && prev != device) { err = hipSetDevice(device); switched = (err == hipSuccess); }
	}
	~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
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
