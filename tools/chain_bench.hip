// Microbenchmark behind DESIGN.md section 5: what does one phase of a dependent kernel chain cost on
// MI355X, as a function of the number of dependent memory round trips inside the kernel?
//   hipcc --offload-arch=gfx950 -O3 tools/chain_bench.hip -o /tmp/chain_bench && /tmp/chain_bench
// Every variant is a hipGraph of 512 kernels launched back to back on one stream, 10k threads each
// (the size of one colour batch of the base-200 pyramid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void kEmpty(float4* a, const int* idx, float4* b, int n) {}
__global__ void kOneTrip(float4* a, const int* idx, float4* b, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		float4 v = a[i];
		v.x += 1.0f;
		a[i] = v;
	}
}
__global__ void kTwoTrips(float4* a, const int* idx, float4* b, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		int j = idx[i];
		float4 v = b[j];
		float4 w = a[i];
		v.x += w.x;
		b[j] = v;
	}
}
// two-trip kernel with a big by-value argument block like ContactView + BodyView
struct Big
{
	float4* p[32];
};
__global__ void kTwoTripsBigArgs(Big big, const int* idx, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		int j = idx[i];
		float4 v = big.p[1][j];
		float4 w = big.p[0][i];
		for (int q = 2; q < 14; ++q)
		{
			float4 t = big.p[q][i];
			w.x += t.x;
		}
		v.x += w.x;
		big.p[1][j] = v;
	}
}

template <class F> float timeGraph(hipStream_t s, int nKernels, F enqueue)
{
	hipGraph_t g;
	hipGraphExec_t ge;
	hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
	for (int i = 0; i < nKernels; ++i)
	{
		enqueue();
	}
	hipStreamEndCapture(s, &g);
	hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int w = 0; w < 3; ++w)
	{
		hipGraphLaunch(ge, s);
	}
	hipStreamSynchronize(s);
	hipEventRecord(e0, s);
	const int reps = 10;
	for (int r = 0; r < reps; ++r)
	{
		hipGraphLaunch(ge, s);
	}
	hipEventRecord(e1, s);
	hipStreamSynchronize(s);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	hipGraphExecDestroy(ge);
	hipGraphDestroy(g);
	return 1e3f * ms / (reps * nKernels);
}

int main()
{
	const int n = 10000, nb = 20000, K = 512;
	float4 *a, *b;
	int* idx;
	hipMalloc(&a, 16 * sizeof(float4) * n);
	hipMalloc(&b, sizeof(float4) * nb);
	hipMalloc(&idx, sizeof(int) * n);
	std::vector<int> h(n);
	for (int i = 0; i < n; ++i)
	{
		h[i] = (int)((i * 7919u) % nb);
	}
	hipMemcpy(idx, h.data(), sizeof(int) * n, hipMemcpyHostToDevice);
	hipMemset(a, 0, 16 * sizeof(float4) * n);
	hipMemset(b, 0, sizeof(float4) * nb);
	hipStream_t s;
	hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	dim3 grid((n + 255) / 256), block(256);
	Big big;
	for (int q = 0; q < 32; ++q)
	{
		big.p[q] = a + (size_t)(q % 16) * n;
	}
	big.p[1] = b;
	printf("us per dependent kernel (graph of %d, %d threads each)\n", K, n);
	printf("  empty kernel              %.2f\n", timeGraph(s, K, [&] { kEmpty<<<grid, block, 0, s>>>(a, idx, b, n); }));
	printf("  1 round trip (rmw a[i])   %.2f\n", timeGraph(s, K, [&] { kOneTrip<<<grid, block, 0, s>>>(a, idx, b, n); }));
	printf("  2 round trips (gather)    %.2f\n", timeGraph(s, K, [&] { kTwoTrips<<<grid, block, 0, s>>>(a, idx, b, n); }));
	printf("  2 trips, 14 arrays, 272 B of kernel args  %.2f\n", timeGraph(s, K, [&] { kTwoTripsBigArgs<<<grid, block, 0, s>>>(big, idx, n); }));
	dim3 grid1(1);
	printf("  1 workgroup, 2 round trips %.2f\n", timeGraph(s, K, [&] { kTwoTrips<<<grid1, block, 0, s>>>(a, idx, b, 256); }));
	return 0;
}
