// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library: how many KiB does the
// counter report per byte actually requested when (a) every lane streams consecutive 16-byte words (the SoA arrays), (b) every lane reads
// ONE 152-byte AoS record of its own -- the wire contact, as prepareSoftFromWire reads it in the resident-island kernels --, (c) the same
// for the 88-byte wire body?  MI355X_MICROARCH.md's rule (double FETCH_SIZE for wide coalesced streams) was derived on (a).
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o pmc -- tools/fetch_calib.bin        (then tools/rocpd_summary.py on the .db)
// Each kernel reads `bytes` in total (printed); the buffer (1 GiB) is far larger than L2 + MALL, every byte is read once.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void streamFloat4(const float4* in, size_t n, float* out)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	float acc = 0.0f;
	for (; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		const float4 v = in[i];
		acc += v.x + v.y + v.z + v.w;
	}
	if (acc == 123.456f)
	{
		*out = acc;
	}
}

template <int WORDS> struct Record
{
	float w[WORDS];
};

template <int WORDS> __global__ void recordPerLane(const Record<WORDS>* in, size_t n, float* out)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	float acc = 0.0f;
	for (; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		const Record<WORDS> r = in[i];
#pragma unroll
		for (int k = 0; k < WORDS; ++k)
		{
			acc += r.w[k];
		}
	}
	if (acc == 123.456f)
	{
		*out = acc;
	}
}

int main()
{
	const size_t bytes = 1ull << 30;
	void* buf = nullptr;
	float* out = nullptr;
	if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess)
	{
		fprintf(stderr, "hipMalloc failed\n");
		return 1;
	}
	(void)hipMemset(buf, 0, bytes);
	(void)hipDeviceSynchronize();
	const dim3 grid(256 * 8), block(256);
	streamFloat4<<<grid, block>>>((const float4*)buf, bytes / 16, out);
	recordPerLane<38><<<grid, block>>>((const Record<38>*)buf, bytes / 152, out);
	recordPerLane<22><<<grid, block>>>((const Record<22>*)buf, bytes / 88, out);
	if (hipDeviceSynchronize() != hipSuccess)
	{
		fprintf(stderr, "kernel failed\n");
		return 1;
	}
	printf("streamFloat4: %zu bytes; recordPerLane<38> (152-byte records): %zu bytes; recordPerLane<22> (88-byte records): %zu bytes\n", (bytes / 16) * 16,
		   (bytes / 152) * 152, (bytes / 88) * 88);
	return 0;
}
