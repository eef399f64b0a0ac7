// TEST INFRASTRUCTURE.  A stand-in for libamdhip64 so that the HOST side of the product -- structure building, colouring, strip
// partition, incremental placement, the world chain's bookkeeping, the reference-side binding's pack / unpack -- can run under
// AddressSanitizer + UndefinedBehaviorSanitizer on a box without a GPU (tests/hostcheck/Makefile, tests/test_hostcheck.py).
// "Device" memory is heap memory (so an out-of-bounds hipMemcpy into a device table is an ASan report), copies are memcpy,
// streams / events / graphs are inert handles, and kernels are never run: hipLaunchKernel returns success and does nothing.
// Results of a step are therefore meaningless; what is checked is that the host code touches only memory it owns.
// The product never links this file.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

static int s_dummy[16];

extern "C" {

hipError_t hipGetDeviceCount(int* count)
{
	*count = 1;
	return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* device)
{
	*device = 0;
	return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
int hipGetStreamDeviceId(hipStream_t) { return 0; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* prop, int)
{
	memset(prop, 0, sizeof(*prop));
	prop->multiProcessorCount = 256, prop->warpSize = 64, prop->maxThreadsPerBlock = 1024;
	strcpy(prop->gcnArchName, "gfx950:sramecc+:xnack-");
	prop->sharedMemPerBlock = 160 * 1024, prop->regsPerBlock = 131072;
	return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t attr, int)
{
	*value = attr == hipDeviceAttributeMultiprocessorCount ? 256 : attr == hipDeviceAttributeWarpSize ? 64 : 1024;
	return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char* pciBusId, int len, int)
{
	strncpy(pciBusId, "0000:00:00.0", (size_t)len);
	return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip_stub"; }
const char* hipGetErrorName(hipError_t) { return "hip_stub"; }

hipError_t hipMalloc(void** ptr, size_t size)
{
	*ptr = calloc(size ? size : 1, 1);
	return *ptr ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* ptr)
{
	free(ptr);
	return hipSuccess;
}
hipError_t hipHostMalloc(void** ptr, size_t size, unsigned int)
{
	*ptr = calloc(size ? size : 1, 1);
	return *ptr ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* ptr)
{
	free(ptr);
	return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void** devPtr, void* hstPtr, unsigned int)
{
	*devPtr = hstPtr;
	return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind)
{
	if (bytes)
	{
		memmove(dst, src, bytes);
	}
	return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t)
{
	if (bytes)
	{
		memmove(dst, src, bytes);
	}
	return hipSuccess;
}
hipError_t hipMemset(void* dst, int value, size_t bytes)
{
	if (bytes)
	{
		memset(dst, value, bytes);
	}
	return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t)
{
	if (bytes)
	{
		memset(dst, value, bytes);
	}
	return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t* stream, unsigned int)
{
	*stream = (hipStream_t)s_dummy;
	return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* stream)
{
	*stream = (hipStream_t)s_dummy;
	return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* dst, int, const void* src, int, size_t bytes, hipStream_t)
{
	if (bytes)
	{
		memmove(dst, src, bytes);
	}
	return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int* can, int, int)
{
	*can = 0;
	return hipSuccess;
}
hipError_t hipDeviceEnablePeerAccess(int, unsigned int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* event)
{
	*event = (hipEvent_t)s_dummy;
	return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* event, unsigned)
{
	*event = (hipEvent_t)s_dummy;
	return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t)
{
	*ms = 0.0f;
	return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* graph)
{
	*graph = (hipGraph_t)s_dummy;
	return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t, hipGraphNode_t*, char*, size_t)
{
	*exec = (hipGraphExec_t)s_dummy;
	return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes* attr, const void*)
{
	memset(attr, 0, sizeof(*attr));
	attr->maxThreadsPerBlock = 1024;
	return hipSuccess;
}
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* numBlocks, const void*, int, size_t)
{
	*numBlocks = 1;
	return hipSuccess;
}

// what clang's host-side code for `kernel<<<...>>>(...)` and for a translation unit with kernels calls
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shared, hipStream_t* stream)
{
	*grid = dim3(1), *block = dim3(1), *shared = 0, *stream = nullptr;
	return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { return (void**)s_dummy; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned int, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void*, void*, void*, const char*, size_t, unsigned) {}
}
