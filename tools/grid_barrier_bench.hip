// Microbenchmark: what does one device-wide barrier cost on MI355X next to one dependent kernel launch in a hipGraph?
// Decides whether a one-launch "op list" kernel (every colour batch of a step separated by grid barriers) can beat the
// graph of per-colour launches.   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_bench.hip -o /tmp/gbb && /tmp/gbb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void gridBarrier(unsigned* counter, unsigned target)
{
	__syncthreads();
	if (threadIdx.x == 0)
	{
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
		{
			__builtin_amdgcn_s_sleep(1);
		}
	}
	__syncthreads();
}

// every phase: each thread gathers two values another workgroup wrote in the phase before (a stand-in for two bodies)
__global__ void __launch_bounds__(256) phasesKernel(float* a, float* b, int n, int phases, unsigned* counter, int work)
{
	const int tid = blockIdx.x * blockDim.x + threadIdx.x;
	const int stride = gridDim.x * blockDim.x;
	for (int p = 0; p < phases; ++p)
	{
		const float* src = (p & 1) ? b : a;
		float* dst = (p & 1) ? a : b;
		if (work)
		{
			for (int i = tid; i < n; i += stride)
			{
				int j = (i + 4099) % n, k = (i + n - 7717 % n) % n;
				dst[i] = __builtin_nontemporal_load(src + j) * 0.5f + __builtin_nontemporal_load(src + k) * 0.5f + 1.0f;
			}
		}
		gridBarrier(counter, (unsigned)(p + 1) * gridDim.x);
	}
}

__global__ void __launch_bounds__(256) onePhaseKernel(const float* src, float* dst, int n)
{
	const int tid = blockIdx.x * blockDim.x + threadIdx.x;
	const int stride = gridDim.x * blockDim.x;
	for (int i = tid; i < n; i += stride)
	{
		int j = (i + 4099) % n, k = (i + n - 7717 % n) % n;
		dst[i] = src[j] * 0.5f + src[k] * 0.5f + 1.0f;
	}
}

int main()
{
	const int n = 20000, phases = 200;
	float *a, *b;
	unsigned* counter;
	CHECK(hipMalloc(&a, n * sizeof(float)));
	CHECK(hipMalloc(&b, n * sizeof(float)));
	CHECK(hipMalloc(&counter, 4));
	CHECK(hipMemset(a, 0, n * sizeof(float)));
	CHECK(hipMemset(b, 0, n * sizeof(float)));
	hipStream_t stream;
	CHECK(hipStreamCreate(&stream));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int work = 0; work < 2; ++work)
	{
		for (int grid : {20, 40, 79, 128, 256, 512})
		{
			float best = 1e9f;
			for (int rep = 0; rep < 5; ++rep)
			{
				CHECK(hipMemsetAsync(counter, 0, 4, stream));
				CHECK(hipEventRecord(e0, stream));
				phasesKernel<<<grid, 256, 0, stream>>>(a, b, n, phases, counter, work);
				CHECK(hipEventRecord(e1, stream));
				CHECK(hipStreamSynchronize(stream));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				best = ms < best ? ms : best;
			}
			printf("one launch, %3d workgroups x 256, work=%d: %.3f us per phase\n", grid, work, 1e3f * best / phases);
		}
	}
	// the same phases as a captured graph of dependent launches
	for (int grid : {20, 79, 256})
	{
		hipGraph_t graph;
		hipGraphExec_t exec;
		CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
		for (int p = 0; p < phases; ++p)
		{
			onePhaseKernel<<<grid, 256, 0, stream>>>((p & 1) ? b : a, (p & 1) ? a : b, n);
		}
		CHECK(hipStreamEndCapture(stream, &graph));
		CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
		float best = 1e9f;
		for (int rep = 0; rep < 5; ++rep)
		{
			CHECK(hipEventRecord(e0, stream));
			CHECK(hipGraphLaunch(exec, stream));
			CHECK(hipEventRecord(e1, stream));
			CHECK(hipStreamSynchronize(stream));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			best = ms < best ? ms : best;
		}
		printf("graph of %d launches, %3d workgroups x 256: %.3f us per phase\n", phases, grid, 1e3f * best / phases);
	}
	std::vector<float> h(n);
	CHECK(hipMemcpy(h.data(), a, n * sizeof(float), hipMemcpyDeviceToHost));
	printf("check %.1f\n", h[123]);
	return 0;
}
