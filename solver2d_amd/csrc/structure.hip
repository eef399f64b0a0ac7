// Constraint-graph structure on the device, behind the same C-ABI:
//   s2amd_find_islands       connected components over the movable bodies (the islands of solver2d_amd/islands.py;
//                            the reference has none) -- lock-free union-find: one CAS-hooking pass over the
//                            edges (roots only ever point to a smaller body index, so the final root of a component is
//                            its lowest index whatever the thread timing), a flatten pass, an exclusive scan over the
//                            roots for "numbered by lowest body index".
//   s2amd_color_constraints  a proper colouring of the contact constraints (no two constraints of one colour share a
//                            writable body) by Jones-Plassmann rounds with a fixed priority hash: a constraint takes the
//                            lowest colour unused by its already-coloured neighbours once every neighbour of higher
//                            priority is coloured.  The outcome equals a sequential greedy colouring in descending
//                            priority order, so it is deterministic and has an exact CPU statement
//                            (tests/structure_ref.py) -- unlike the host's pool-order greedy (graph_coloring.cpp: colorGraph),
//                            whose dependency chains are as long as the pool.
// Integer work; results are compared exactly.  Host arrays in and out like the other stage calls.

#include "launch.h"
#include "s2_device.h"

#include "solver2d_amd.h"

#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include <string>
#include <vector>

#define S2_ST_BLOCK 256

int s2amdFail(int code, const std::string& msg);
hipStream_t s2amdStream(s2amdSolver* s);
int s2amdDevice(s2amdSolver* s);
void s2amdRecordDeviceMs(s2amdSolver* s, float ms);

namespace
{

__device__ __forceinline__ bool movableBody(const s2amdBody& b) { return b.type != S2AMD_BODY_FREE && (b.invMass != 0.0f || b.invI != 0.0f); }
__device__ __forceinline__ bool ownedBody(const s2amdBody& b) { return b.type != S2AMD_BODY_FREE && b.type != S2AMD_BODY_STATIC; }

// ---- islands ----
// Reads go through agent-scope atomics: another CU's hook must become visible inside this launch, and a CU's
// vector L1 is never refreshed by other CUs' stores (a plain re-read could spin on a stale "root" for ever).
__device__ __forceinline__ int loadParent(int* parent, int i) { return __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int findRoot(int* parent, int i)
{
	int p = loadParent(parent, i);
	while (p != i)
	{
		int gp = loadParent(parent, p);
		if (gp != p)
		{
			// path halving: a benign race, every value written is an ancestor of i (and i is not a root any more)
			__hip_atomic_store(parent + i, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		i = p;
		p = gp;
	}
	return i;
}

__device__ __forceinline__ void hook(int* parent, int a, int b)
{
	for (;;)
	{
		int ra = findRoot(parent, a), rb = findRoot(parent, b);
		if (ra == rb)
		{
			return;
		}
		int hi = ra > rb ? ra : rb, lo = ra > rb ? rb : ra;
		// hi is a root only while parent[hi] == hi
		if (atomicCAS(&parent[hi], hi, lo) == hi)
		{
			return;
		}
	}
}

__global__ __launch_bounds__(S2_ST_BLOCK) void initParentKernel(int* parent, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		parent[i] = i;
	}
}

__global__ __launch_bounds__(S2_ST_BLOCK) void hookEdgesKernel(const s2amdBody* bodies, const s2amdContact* contacts, int nc, const s2amdJoint* joints,
															   int nj, int* parent)
{
	int e = blockIdx.x * blockDim.x + threadIdx.x;
	int a = -1, b = -1;
	if (e < nc)
	{
		if (contacts[e].pointCount > 0)
		{
			a = contacts[e].bodyA, b = contacts[e].bodyB;
		}
	}
	else if (e < nc + nj)
	{
		const s2amdJoint& j = joints[e - nc];
		if (j.type == S2AMD_JOINT_REVOLUTE)
		{
			a = j.bodyA, b = j.bodyB;
		}
	}
	if (a >= 0 && b >= 0 && movableBody(bodies[a]) && movableBody(bodies[b]))
	{
		hook(parent, a, b);
	}
}

__global__ __launch_bounds__(S2_ST_BLOCK) void flattenKernel(const s2amdBody* bodies, int* parent, int n, unsigned int* isRoot)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
	{
		return;
	}
	int r = findRoot(parent, i);
	parent[i] = r; // every thread writes the final root of its own body only after reading ancestors: converges to the root
	isRoot[i] = (ownedBody(bodies[i]) && r == i) ? 1u : 0u;
}

__global__ __launch_bounds__(S2_ST_BLOCK) void labelKernel(const s2amdBody* bodies, const int* parent, const unsigned int* rootRank, int n, int32_t* island)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
	{
		return;
	}
	int r = parent[i];
	// a second hop covers a parent written before its own flatten finished
	r = parent[r];
	island[i] = ownedBody(bodies[i]) ? (int32_t)rootRank[r] : -1;
}

// ---- colouring ----
__host__ __device__ __forceinline__ uint32_t priorityOf(uint32_t k)
{
	// murmur3 finaliser: a bijection on 32 bits, so priorities are distinct
	k ^= k >> 16;
	k *= 0x85ebca6bu;
	k ^= k >> 13;
	k *= 0xc2b2ae35u;
	k ^= k >> 16;
	return k;
}

__global__ __launch_bounds__(S2_ST_BLOCK) void countIncidenceKernel(const s2amdBody* bodies, const s2amdContact* contacts, int nc, unsigned int* degree)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nc || contacts[k].pointCount <= 0)
	{
		return;
	}
	int a = contacts[k].bodyA, b = contacts[k].bodyB;
	if (movableBody(bodies[a]))
	{
		atomicAdd(&degree[a], 1u);
	}
	if (b != a && movableBody(bodies[b]))
	{
		atomicAdd(&degree[b], 1u);
	}
}

__global__ __launch_bounds__(S2_ST_BLOCK) void fillIncidenceKernel(const s2amdBody* bodies, const s2amdContact* contacts, int nc, const unsigned int* offsets,
																   unsigned int* cursor, int* incident)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nc || contacts[k].pointCount <= 0)
	{
		return;
	}
	int a = contacts[k].bodyA, b = contacts[k].bodyB;
	if (movableBody(bodies[a]))
	{
		incident[offsets[a] + atomicAdd(&cursor[a], 1u)] = k;
	}
	if (b != a && movableBody(bodies[b]))
	{
		incident[offsets[b] + atomicAdd(&cursor[b], 1u)] = k;
	}
}

// {movable end A or -1, movable end B or -1 (also when B == A)}; {-2, -2} for an inactive slot: the rounds below then
// never touch the 152-byte contact records or the bodies again
__global__ __launch_bounds__(S2_ST_BLOCK) void compactEndsKernel(const s2amdBody* bodies, const s2amdContact* contacts, int nc, int2* ends)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nc)
	{
		return;
	}
	if (contacts[k].pointCount <= 0)
	{
		ends[k] = make_int2(-2, -2);
		return;
	}
	int a = contacts[k].bodyA, b = contacts[k].bodyB;
	ends[k] = make_int2(movableBody(bodies[a]) ? a : -1, (b != a && movableBody(bodies[b])) ? b : -1);
}

#define S2_COLOR_WORDS 4 // 256 colours

// one Jones-Plassmann round: colours written in a round are not read in the same round by constraints that
// depend on them (they see -1 and wait), so a round is race-free
__global__ __launch_bounds__(S2_ST_BLOCK) void colourRoundKernel(const int2* endsOf, int nc, const unsigned int* offsets, const int* incident,
																 const int32_t* colourIn, int32_t* colourOut, unsigned int* remaining, unsigned int* overflow)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= nc)
	{
		return;
	}
	int32_t mine = colourIn[k];
	const int2 e2 = endsOf[k];
	if (e2.x == -2 || mine >= 0)
	{
		colourOut[k] = e2.x == -2 ? -1 : mine;
		return;
	}
	const uint32_t pk = priorityOf((uint32_t)k);
	unsigned long long used[S2_COLOR_WORDS] = {0ull, 0ull, 0ull, 0ull};
	bool wait = false;
	int ends[2] = {e2.x, e2.y};
	for (int side = 0; side < 2 && !wait; ++side)
	{
		int body = ends[side];
		if (body < 0)
		{
			continue;
		}
		for (unsigned int e = offsets[body]; e < offsets[body + 1]; ++e)
		{
			int n = incident[e];
			if (n == k)
			{
				continue;
			}
			int32_t cn = colourIn[n];
			if (cn < 0)
			{
				if (priorityOf((uint32_t)n) > pk)
				{
					wait = true;
					break;
				}
			}
			else if (cn < 64 * S2_COLOR_WORDS)
			{
				used[cn >> 6] |= 1ull << (cn & 63);
			}
		}
	}
	if (wait)
	{
		colourOut[k] = -1;
		*remaining = 1u; // "somebody still waits": a flag, every writer stores the same value (tens of thousands of atomics on one word cost more than the round)
		return;
	}
	int32_t chosen = -1;
	for (int w = 0; w < S2_COLOR_WORDS && chosen < 0; ++w)
	{
		if (~used[w])
		{
			chosen = w * 64 + __builtin_ctzll(~used[w]);
		}
	}
	if (chosen < 0)
	{
		atomicAdd(overflow, 1u);
		chosen = 64 * S2_COLOR_WORDS - 1;
	}
	colourOut[k] = chosen;
}

struct Scratch
{
	void* p = nullptr;
	~Scratch()
	{
		if (p)
		{
			(void)hipFree(p);
		}
	}
};

dim3 gridOf(size_t n) { return dim3((unsigned)((n + S2_ST_BLOCK - 1) / S2_ST_BLOCK)); }

} // namespace

#define ST_TRY(expr)                                                                                                             \
	do                                                                                                                           \
	{                                                                                                                            \
		hipError_t _e = (expr);                                                                                                  \
		if (_e != hipSuccess)                                                                                                    \
		{                                                                                                                        \
			return s2amdFail(S2AMD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
		}                                                                                                                        \
	} while (0)

static int checkIndices(const s2amdContact* contacts, int nc, const s2amdJoint* joints, int nj, int nb)
{
	for (int i = 0; i < nc; ++i)
	{
		if (contacts[i].pointCount > 0 && (contacts[i].bodyA < 0 || contacts[i].bodyA >= nb || contacts[i].bodyB < 0 || contacts[i].bodyB >= nb))
		{
			return s2amdFail(S2AMD_E_INVALID, "contact " + std::to_string(i) + " names a body outside the body array");
		}
	}
	for (int i = 0; i < nj; ++i)
	{
		if (joints[i].type == S2AMD_JOINT_REVOLUTE && (joints[i].bodyA < 0 || joints[i].bodyA >= nb || joints[i].bodyB < 0 || joints[i].bodyB >= nb))
		{
			return s2amdFail(S2AMD_E_INVALID, "joint " + std::to_string(i) + " names a body outside the body array");
		}
	}
	return S2AMD_OK;
}

#pragma GCC visibility push(default)
extern "C"
{

int s2amd_find_islands(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
					   const s2amdJoint* joints, int32_t jointCapacity, int32_t* islandOfBody, int32_t* islandCount)
{
	if (!solver || !islandCount || bodyCapacity < 0 || contactCapacity < 0 || jointCapacity < 0 || (bodyCapacity > 0 && (!bodies || !islandOfBody)) ||
		(contactCapacity > 0 && !contacts) || (jointCapacity > 0 && !joints))
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument");
	}
	*islandCount = 0;
	if (bodyCapacity == 0)
	{
		return S2AMD_OK;
	}
	int rc = checkIndices(contacts, contactCapacity, joints, jointCapacity, bodyCapacity);
	if (rc)
	{
		return rc;
	}
	ST_TRY(hipSetDevice(s2amdDevice(solver)));
	hipStream_t st = s2amdStream(solver);
	const int nb = bodyCapacity, nc = contactCapacity, nj = jointCapacity;
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	size_t tmpScan = 0;
	ST_TRY(rocprim::exclusive_scan(nullptr, tmpScan, (unsigned int*)nullptr, (unsigned int*)nullptr, 0u, (size_t)nb + 1, rocprim::plus<unsigned int>(), st));
	size_t bB = al((size_t)nb * sizeof(s2amdBody)), bC = al((size_t)std::max(nc, 1) * sizeof(s2amdContact)), bJ = al((size_t)std::max(nj, 1) * sizeof(s2amdJoint));
	size_t bI = al(((size_t)nb + 1) * 4);
	Scratch buf;
	ST_TRY(hipMalloc(&buf.p, bB + bC + bJ + 4 * bI + al(tmpScan + 256)));
	char* base = (char*)buf.p;
	s2amdBody* dB = (s2amdBody*)base;
	s2amdContact* dC = (s2amdContact*)(base + bB);
	s2amdJoint* dJ = (s2amdJoint*)(base + bB + bC);
	int* dParent = (int*)(base + bB + bC + bJ);
	unsigned int* dIsRoot = (unsigned int*)((char*)dParent + bI);
	unsigned int* dRank = (unsigned int*)((char*)dIsRoot + bI);
	int32_t* dIsland = (int32_t*)((char*)dRank + bI);
	void* dTmp = (char*)dIsland + bI;
	ST_TRY(hipMemcpyAsync(dB, bodies, (size_t)nb * sizeof(s2amdBody), hipMemcpyHostToDevice, st));
	if (nc > 0)
	{
		ST_TRY(hipMemcpyAsync(dC, contacts, (size_t)nc * sizeof(s2amdContact), hipMemcpyHostToDevice, st));
	}
	if (nj > 0)
	{
		ST_TRY(hipMemcpyAsync(dJ, joints, (size_t)nj * sizeof(s2amdJoint), hipMemcpyHostToDevice, st));
	}
	hipEvent_t e0 = nullptr, e1 = nullptr;
	ST_TRY(hipEventCreate(&e0));
	ST_TRY(hipEventCreate(&e1));
	ST_TRY(hipEventRecord(e0, st));
	initParentKernel<<<gridOf((size_t)nb), dim3(S2_ST_BLOCK), 0, st>>>(dParent, nb);
	if (nc + nj > 0)
	{
		hookEdgesKernel<<<gridOf((size_t)nc + nj), dim3(S2_ST_BLOCK), 0, st>>>(dB, dC, nc, dJ, nj, dParent);
	}
	ST_TRY(hipMemsetAsync(dIsRoot, 0, bI, st));
	flattenKernel<<<gridOf((size_t)nb), dim3(S2_ST_BLOCK), 0, st>>>(dB, dParent, nb, dIsRoot);
	size_t tmp = tmpScan + 256;
	ST_TRY(rocprim::exclusive_scan(dTmp, tmp, dIsRoot, dRank, 0u, (size_t)nb + 1, rocprim::plus<unsigned int>(), st));
	labelKernel<<<gridOf((size_t)nb), dim3(S2_ST_BLOCK), 0, st>>>(dB, dParent, dRank, nb, dIsland);
	ST_TRY(hipEventRecord(e1, st));
	ST_TRY(hipGetLastError());
	unsigned int count = 0;
	ST_TRY(hipMemcpyAsync(islandOfBody, dIsland, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
	ST_TRY(hipMemcpyAsync(&count, dRank + nb, 4, hipMemcpyDeviceToHost, st));
	ST_TRY(hipStreamSynchronize(st));
	float ms = 0.0f;
	(void)hipEventElapsedTime(&ms, e0, e1);
	s2amdRecordDeviceMs(solver, ms);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	*islandCount = (int32_t)count;
	return S2AMD_OK;
}

int s2amd_color_constraints(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
							int32_t* colorOfContact, int32_t* colorCount, int32_t* rounds)
{
	if (!solver || !colorCount || bodyCapacity < 0 || contactCapacity < 0 || (bodyCapacity > 0 && !bodies) ||
		(contactCapacity > 0 && (!contacts || !colorOfContact)))
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument");
	}
	*colorCount = 0;
	if (rounds)
	{
		*rounds = 0;
	}
	if (contactCapacity == 0)
	{
		return S2AMD_OK;
	}
	int rc = checkIndices(contacts, contactCapacity, nullptr, 0, bodyCapacity);
	if (rc)
	{
		return rc;
	}
	ST_TRY(hipSetDevice(s2amdDevice(solver)));
	hipStream_t st = s2amdStream(solver);
	const int nb = bodyCapacity, nc = contactCapacity;
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	size_t tmpScan = 0;
	ST_TRY(rocprim::exclusive_scan(nullptr, tmpScan, (unsigned int*)nullptr, (unsigned int*)nullptr, 0u, (size_t)nb + 1, rocprim::plus<unsigned int>(), st));
	size_t bB = al((size_t)nb * sizeof(s2amdBody)), bC = al((size_t)nc * sizeof(s2amdContact)), bN = al(((size_t)nb + 1) * 4);
	size_t bK = al((size_t)nc * 4), bInc = al((size_t)nc * 2 * 4);
	Scratch buf;
	ST_TRY(hipMalloc(&buf.p, bB + bC + 3 * bN + 2 * bK + 2 * bInc + 256 + al(tmpScan + 256)));
	char* base = (char*)buf.p;
	s2amdBody* dB = (s2amdBody*)base;
	s2amdContact* dC = (s2amdContact*)(base + bB);
	unsigned int* dDegree = (unsigned int*)(base + bB + bC);
	unsigned int* dOffsets = (unsigned int*)((char*)dDegree + bN);
	unsigned int* dCursor = (unsigned int*)((char*)dOffsets + bN);
	int32_t* dColourA = (int32_t*)((char*)dCursor + bN);
	int32_t* dColourB = (int32_t*)((char*)dColourA + bK);
	int* dIncident = (int*)((char*)dColourB + bK);
	int2* dEnds = (int2*)((char*)dIncident + bInc);
	unsigned int* dCounters = (unsigned int*)((char*)dEnds + bInc); // [1] overflow, [2 + r] constraints still waiting after round r of a burst
	void* dTmp = (char*)dCounters + 256;
	ST_TRY(hipMemcpyAsync(dB, bodies, (size_t)nb * sizeof(s2amdBody), hipMemcpyHostToDevice, st));
	ST_TRY(hipMemcpyAsync(dC, contacts, (size_t)nc * sizeof(s2amdContact), hipMemcpyHostToDevice, st));
	hipEvent_t e0 = nullptr, e1 = nullptr;
	ST_TRY(hipEventCreate(&e0));
	ST_TRY(hipEventCreate(&e1));
	ST_TRY(hipEventRecord(e0, st));
	ST_TRY(hipMemsetAsync(dDegree, 0, 3 * bN, st));
	ST_TRY(hipMemsetAsync(dColourA, 0xff, 2 * bK, st)); // -1: uncoloured
	ST_TRY(hipMemsetAsync(dCounters, 0, 256, st));
	countIncidenceKernel<<<gridOf((size_t)nc), dim3(S2_ST_BLOCK), 0, st>>>(dB, dC, nc, dDegree);
	size_t tmp = tmpScan + 256;
	ST_TRY(rocprim::exclusive_scan(dTmp, tmp, dDegree, dOffsets, 0u, (size_t)nb + 1, rocprim::plus<unsigned int>(), st));
	fillIncidenceKernel<<<gridOf((size_t)nc), dim3(S2_ST_BLOCK), 0, st>>>(dB, dC, nc, dOffsets, dCursor, dIncident);
	compactEndsKernel<<<gridOf((size_t)nc), dim3(S2_ST_BLOCK), 0, st>>>(dB, dC, nc, dEnds);
	int done = 0;
	unsigned int host[2] = {1u, 0u};
	const int kMaxRounds = 4096, kMaxBurst = 24; // rounds between looks at the counter: a read-back costs ten rounds
	while (host[0] != 0u && done < kMaxRounds)
	{
		const int kBurst = done == 0 ? kMaxBurst : 8; // bounded-degree contact graphs finish in about twenty rounds
		ST_TRY(hipMemsetAsync(dCounters + 2, 0, kMaxBurst * 4, st));
		for (int r = 0; r < kBurst; ++r)
		{
			colourRoundKernel<<<gridOf((size_t)nc), dim3(S2_ST_BLOCK), 0, st>>>(dEnds, nc, dOffsets, dIncident, dColourA, dColourB, dCounters + 2 + r,
																				 dCounters + 1);
			std::swap(dColourA, dColourB);
			done += 1;
		}
		unsigned int burst[2 + 24];
		ST_TRY(hipMemcpyAsync(burst, dCounters, sizeof(burst), hipMemcpyDeviceToHost, st));
		ST_TRY(hipStreamSynchronize(st));
		host[0] = burst[2 + kBurst - 1]; // constraints still waiting after the last round of the burst
		host[1] = burst[1];
		if (host[0] == 0u)
		{
			// report the round that finished the job, not the end of the burst
			for (int r = kBurst - 1; r > 0 && burst[2 + r - 1] == 0u; --r)
			{
				done -= 1;
			}
		}
	}
	ST_TRY(hipEventRecord(e1, st));
	ST_TRY(hipGetLastError());
	ST_TRY(hipMemcpyAsync(colorOfContact, dColourA, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
	ST_TRY(hipStreamSynchronize(st));
	float ms = 0.0f;
	(void)hipEventElapsedTime(&ms, e0, e1);
	s2amdRecordDeviceMs(solver, ms);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	if (host[0] != 0u)
	{
		return s2amdFail(S2AMD_E_DEVICE, "colouring did not converge");
	}
	if (host[1] != 0u)
	{
		return s2amdFail(S2AMD_E_CAPACITY, "a body carries more than 255 active contacts: more colours than the device colouring supports");
	}
	int32_t maxColour = -1;
	for (int i = 0; i < nc; ++i)
	{
		maxColour = std::max(maxColour, colorOfContact[i]);
	}
	*colorCount = maxColour + 1;
	if (rounds)
	{
		*rounds = done;
	}
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop

S2_DEFINE_WARM(structure)
