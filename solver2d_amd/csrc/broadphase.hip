// Stages either side of the solver, behind the same C-ABI:
//   s2amd_refit_shapes  == Stage 4 of s2World_Step (src/world.c:259-301): body origins, tight AABBs,
//                          fat-AABB re-inflation; one thread per body, one per shape.
//   s2amd_find_pairs    == the pair discovery of s2UpdateBroadPhasePairs (src/broad_phase.c:166-307).
//                          The reference walks three dynamic AABB trees per moved proxy (pointer
//                          chasing, CPU); here: sort the fat AABBs by lower x (rocPRIM radix sort),
//                          one binary search per shape for the end of its x-overlap run, a prefix sum
//                          over the run lengths, then ONE thread per candidate (shape, later shape)
//                          so a 400 m ground box and a 1 m brick cost the same per thread.  The
//                          result is the reference's pair SET with its A/B orientation, sorted.
// Host arrays in, host arrays out (the callers of these stages keep their worlds on the host today).

#include "launch.h"
#include "s2_device.h"
#include "refit_ops.h"

#include "solver2d_amd.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <string>
#include <vector>

#define S2_BLOCK 256

int s2amdFail(int code, const std::string& msg);
hipStream_t s2amdStream(s2amdSolver* s);
int s2amdDevice(s2amdSolver* s);

__global__ __launch_bounds__(S2_BLOCK) void bodyOriginsKernel(const s2amdBody* bodies, int n, float2* origins)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
	{
		return;
	}
	const s2amdBody* b = bodies + i;
	if (b->type == S2AMD_BODY_FREE || b->type == S2AMD_BODY_STATIC)
	{
		return;
	}
	Rot q;
	q.s = b->rot[0], q.c = b->rot[1];
	V2 o = sub(v2(b->position[0], b->position[1]), rotate(q, v2(b->localCenter[0], b->localCenter[1])));
	origins[i] = make_float2(o.x, o.y);
}

__global__ __launch_bounds__(S2_BLOCK) void refitShapesKernel(const s2amdBody* bodies, int nb, s2amdShape* shapes, int ns, const float2* origins)
{
	int si = blockIdx.x * blockDim.x + threadIdx.x;
	if (si >= ns)
	{
		return;
	}
	s2amdShape* sh = shapes + si;
	if (sh->type == S2AMD_SHAPE_FREE || sh->body < 0 || sh->body >= nb)
	{
		return;
	}
	const s2amdBody* b = bodies + sh->body;
	if (b->type == S2AMD_BODY_FREE || b->type == S2AMD_BODY_STATIC)
	{
		return;
	}
	float2 o = origins[sh->body];
	Rot qb;
	qb.s = b->rot[0], qb.c = b->rot[1];
	(void)refitShapeOne(qb, sh, v2(o.x, o.y));
}

// Stage 4 of the resident world in ONE launch (src/world.c:259-301): blocks [0, shapeBlocks) refit one shape per thread,
// with the body origin recomputed from the body (the same expression bodyOriginsKernel evaluates, so the same bits);
// the other blocks walk the bodies: origin written for the next stage 3, applied forces consumed (src/world.c:274-275).
__global__ __launch_bounds__(S2_BLOCK) void stage4Kernel(s2amdBody* bodies, int nb, s2amdShape* shapes, int ns, float2* origins, int shapeBlocks,
														  int* summary, const unsigned int* stepFailed)
{
	if (stepFailed != nullptr && *stepFailed != 0u)
	{
		return; // (a persistent step that lost a hand-off left the bodies alone and will be repeated: nothing moved, and the applied forces are still to be consumed)
	}
	if ((int)blockIdx.x >= shapeBlocks)
	{
		int i = ((int)blockIdx.x - shapeBlocks) * (int)blockDim.x + (int)threadIdx.x;
		if (i < nb)
		{
			stage4BodyOne(bodies, i, origins, nullptr);
		}
		return;
	}
	int si = blockIdx.x * blockDim.x + threadIdx.x;
	int enlarged = 0;
	if (si < ns)
	{
		enlarged = stage4ShapeOne(bodies, nb, shapes + si, nullptr);
	}
	unsigned long long m = __ballot(enlarged != 0);
	if ((threadIdx.x & 63) == 0 && m != 0ull)
	{
		atomicAdd(summary + 4, __popcll(m));
	}
}

// ---- pair discovery ----

// order-preserving map float -> uint32 (total order identical to '<' on non-NaN floats, -0 < +0 aside)
static inline uint32_t sortableFloat(float f)
{
	uint32_t u;
	memcpy(&u, &f, 4);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(S2_BLOCK) void gatherLowerXKernel(const s2amdShape* shapes, const int* sortedIdx, int n, float* lowerX)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		lowerX[i] = shapes[sortedIdx[i]].fatAABB[0];
	}
}

// the resident query's compact view of the proxies in sweep order: fat box and move flag (the pair kernels test a candidate on 17 bytes
// before they touch the 196-byte shape records of the few that pass)
__global__ __launch_bounds__(S2_BLOCK) void gatherBoxesKernel(const s2amdShape* shapes, const unsigned char* moved, const int* sortedIdx, int n, float4* box,
															  unsigned char* movedSorted)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		const int s = sortedIdx[i];
		const float* f = shapes[s].fatAABB;
		box[i] = make_float4(f[0], f[1], f[2], f[3]);
		movedSorted[i] = moved[s];
	}
}
S2_DEV bool boxesOverlap(float4 a, float4 b) // aabbOverlaps on the copies: the same comparisons on the same values
{
	float d1x = b.x - a.z, d1y = b.y - a.w;
	float d2x = a.x - b.z, d2y = a.y - b.w;
	if (d1x > 0.0f || d1y > 0.0f)
	{
		return false;
	}
	if (d2x > 0.0f || d2y > 0.0f)
	{
		return false;
	}
	return true;
}

// for sorted position i: first position whose lower x lies beyond this shape's upper x
__global__ __launch_bounds__(S2_BLOCK) void runLengthKernel(const s2amdShape* shapes, const int* sortedIdx, const float* sortedLowerX, int n,
															 unsigned int* runLength)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
	{
		return;
	}
	float ux = shapes[sortedIdx[i]].fatAABB[2];
	int lo = i + 1, hi = n; // first j in (i, n) with lowerX[j] - ux > 0
	while (lo < hi)
	{
		int mid = (lo + hi) >> 1;
		if (sortedLowerX[mid] - ux > 0.0f)
		{
			hi = mid;
		}
		else
		{
			lo = mid + 1;
		}
	}
	runLength[i] = (unsigned int)(lo - (i + 1));
}

S2_DEV bool aabbOverlaps(const float* a, const float* b) // s2AABB_Overlaps
{
	float d1x = b[0] - a[2], d1y = b[1] - a[3];
	float d2x = a[0] - b[2], d2y = a[1] - b[3];
	if (d1x > 0.0f || d1y > 0.0f)
	{
		return false;
	}
	if (d2x > 0.0f || d2y > 0.0f)
	{
		return false;
	}
	return true;
}

S2_DEV bool containsKey(const unsigned long long* keys, int n, unsigned long long key)
{
	int lo = 0, hi = n;
	while (lo < hi)
	{
		int mid = (lo + hi) >> 1;
		if (keys[mid] < key)
		{
			lo = mid + 1;
		}
		else
		{
			hi = mid;
		}
	}
	return lo < n && keys[lo] == key;
}

// ... and where (-1: nowhere)
S2_DEV int findKey(const unsigned long long* keys, int n, unsigned long long key)
{
	int lo = 0, hi = n;
	while (lo < hi)
	{
		int mid = (lo + hi) >> 1;
		if (keys[mid] < key)
		{
			lo = mid + 1;
		}
		else
		{
			hi = mid;
		}
	}
	return lo < n && keys[lo] == key ? lo : -1;
}

// does proxy P (moved) report the pair when its tree query meets Q?  src/broad_phase.c:166-181, :283-298
S2_DEV bool reports(const s2amdShape& P, bool movedP, const s2amdShape& Q, bool movedQ)
{
	if (!movedP)
	{
		return false;
	}
	int ptype = P.proxyKey & 0xF, qtype = Q.proxyKey & 0xF;
	if (ptype == S2AMD_BODY_STATIC)
	{
		return false;
	}
	if (ptype == S2AMD_BODY_KINEMATIC && qtype != S2AMD_BODY_DYNAMIC)
	{
		return false;
	}
	if (movedQ && Q.proxyKey > P.proxyKey)
	{
		return false;
	}
	return true;
}

// The resident world's pair set as the query sees it (world.hip): `existing` -- the keys of every pair slot as they stood at the last
// sort, ascending -- is only a directory now.  An entry names the SLOT it came from, and "the contact exists" means that slot holds that
// pair NOW: a pair stage 3 has destroyed since (this step's, read behind the same launch sequence, or an earlier one's) fails the
// test without anybody having told the query.  Pairs created since the sort wait in a second, small directory the host keeps (`log`:
// [0] = entries, then keys ascending; `logSlots` their slots).  The big directory is sorted again -- a radix sort over every pair
// slot -- only when the small one is full.  All pointers null: `existing` is a plain key set (the host-array route).
struct PairSetView
{
	const int* sortedSlots = nullptr;
	const unsigned long long* log = nullptr;
	const int* logSlots = nullptr;
	const s2amdPairState* pairs = nullptr;
	// (the sorted directory holds COMPACT keys, (min shape << shapeBits) | max shape: half the radix passes of the 64-bit form when it
	// is sorted again; 0: the 64-bit form (min << 32) | max)
	int shapeBits = 0;
};
#define S2_PAIR_LOG_CAPACITY 255
typedef PairSetView GoneKeys;
S2_DEV void pairTest(int xi, int yi, const s2amdShape* shapes, const unsigned char* moved, const unsigned long long* existing, int existingCount,
					 const unsigned long long* jointed, int jointedCount, unsigned long long* outKeys, unsigned int outCapacity, unsigned int* outCount,
					 GoneKeys gone = GoneKeys{});

// one candidate (work item t of the sweep-and-prune runs): which pair it is
S2_DEV void pairOne(unsigned int t, const s2amdShape* shapes, const unsigned char* moved, const int* sortedIdx, const unsigned int* runOffset, int n,
					const unsigned long long* existing, int existingCount, const unsigned long long* jointed, int jointedCount,
					unsigned long long* outKeys, unsigned int outCapacity, unsigned int* outCount)
{
	// which run does work item t belong to?  last i with runOffset[i] <= t
	int lo = 0, hi = n;
	while (lo + 1 < hi)
	{
		int mid = (lo + hi) >> 1;
		if (runOffset[mid] <= t)
		{
			lo = mid;
		}
		else
		{
			hi = mid;
		}
	}
	int i = lo;
	int j = i + 1 + (int)(t - runOffset[i]);
	pairTest(sortedIdx[i], sortedIdx[j], shapes, moved, existing, existingCount, jointed, jointedCount, outKeys, outCapacity, outCount);
}

// one candidate pair of shapes whose fat boxes overlap on the sweep axis: the reference's pair rules
S2_DEV void pairTest(int xi, int yi, const s2amdShape* shapes, const unsigned char* moved, const unsigned long long* existing, int existingCount,
					 const unsigned long long* jointed, int jointedCount, unsigned long long* outKeys, unsigned int outCapacity, unsigned int* outCount,
					 GoneKeys gone)
{
	const s2amdShape& X = shapes[xi];
	const s2amdShape& Y = shapes[yi];
	if (!aabbOverlaps(X.fatAABB, Y.fatAABB))
	{
		return;
	}
	bool mx = moved[xi] != 0, my = moved[yi] != 0;
	if (!(reports(X, mx, Y, my) || reports(Y, my, X, mx)))
	{
		return;
	}
	unsigned int slo = (unsigned int)(xi < yi ? xi : yi), shi = (unsigned int)(xi < yi ? yi : xi);
	const unsigned long long key64 = ((unsigned long long)slo << 32) | shi;
	auto holds = [&](int slot) {
		const int pa = gone.pairs[slot].shapeA, pb = gone.pairs[slot].shapeB;
		return pa >= 0 && (unsigned int)(pa < pb ? pa : pb) == slo && (unsigned int)(pa < pb ? pb : pa) == shi;
	};
	bool exists = false;
	{
		const int at = findKey(existing, existingCount, gone.shapeBits > 0 ? (((unsigned long long)slo << gone.shapeBits) | shi) : key64);
		exists = at >= 0 && (gone.sortedSlots == nullptr || holds(gone.sortedSlots[at]));
	}
	if (!exists && gone.log != nullptr)
	{
		const int at = findKey(gone.log + 1, (int)gone.log[0], key64);
		exists = at >= 0 && holds(gone.logSlots[at]);
	}
	if (exists)
	{
		return; // the contact exists (:183-188)
	}
	int ia = xi, ib = yi;
	if (Y.proxyKey < X.proxyKey) // shape A has the lower proxy key (:190-200)
	{
		ia = yi, ib = xi;
	}
	const s2amdShape& A = shapes[ia];
	const s2amdShape& B = shapes[ib];
	if (A.body == B.body)
	{
		return;
	}
	// s2ShouldShapesCollide: src/contact.h:68-78
	if (A.groupIndex == B.groupIndex && A.groupIndex != 0)
	{
		if (!(A.groupIndex > 0))
		{
			return;
		}
	}
	else if (!((A.maskBits & B.categoryBits) != 0 && (A.categoryBits & B.maskBits) != 0))
	{
		return;
	}
	// s2ShouldBodiesCollide: any joint between the bodies blocks the pair
	unsigned int blo = (unsigned int)(A.body < B.body ? A.body : B.body), bhi = (unsigned int)(A.body < B.body ? B.body : A.body);
	if (containsKey(jointed, jointedCount, ((unsigned long long)blo << 32) | bhi))
	{
		return;
	}
	// s2CreateContact: segment/segment has no manifold function; non-primary type orders are flipped
	int tA = A.type, tB = B.type;
	if (tA == S2AMD_SHAPE_SEGMENT && tB == S2AMD_SHAPE_SEGMENT)
	{
		return;
	}
	const unsigned int primaryBits = 0x7730u | 0x0002u | 0x0003u; // rows capsule{1,1,0,0} circle{0,1,0,0} polygon{1,1,1,0} segment{1,1,1,0}
	(void)primaryBits;
	bool primary;
	switch (tA)
	{
		case S2AMD_SHAPE_CAPSULE:
			primary = tB == S2AMD_SHAPE_CAPSULE || tB == S2AMD_SHAPE_CIRCLE;
			break;
		case S2AMD_SHAPE_CIRCLE:
			primary = tB == S2AMD_SHAPE_CIRCLE;
			break;
		default: // polygon, segment
			primary = tB != S2AMD_SHAPE_SEGMENT;
			break;
	}
	if (!primary)
	{
		int tmp = ia;
		ia = ib;
		ib = tmp;
	}
	unsigned int slot = atomicAdd(outCount, 1u);
	if (slot < outCapacity)
	{
		outKeys[slot] = ((unsigned long long)(unsigned int)ia << 32) | (unsigned int)ib;
	}
}

__global__ __launch_bounds__(S2_BLOCK) void pairKernel(const s2amdShape* shapes, const unsigned char* moved, const int* sortedIdx,
														const unsigned int* runOffset, int n, unsigned int work, const unsigned long long* existing,
														int existingCount, const unsigned long long* jointed, int jointedCount,
														unsigned long long* outKeys, unsigned int outCapacity, unsigned int* outCount)
{
	unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < work)
	{
		pairOne(t, shapes, moved, sortedIdx, runOffset, n, existing, existingCount, jointed, jointedCount, outKeys, outCapacity, outCount);
	}
}

// the same over a fixed grid: the number of work items is read from the scan's total on the device, so the host does not
// have to fetch it before the launch
__global__ __launch_bounds__(S2_BLOCK) void pairStrideKernel(const s2amdShape* shapes, const unsigned char* moved, const int* sortedIdx,
															  const unsigned int* runOffset, int n, const unsigned long long* existing, int existingCount,
															  const unsigned long long* jointed, int jointedCount, unsigned long long* outKeys,
															  unsigned int outCapacity, unsigned int* outCount)
{
	const unsigned int work = runOffset[n];
	for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < work; t += gridDim.x * blockDim.x)
	{
		pairOne(t, shapes, moved, sortedIdx, runOffset, n, existing, existingCount, jointed, jointedCount, outKeys, outCapacity, outCount);
	}
}

// The resident query: ONE WAVE PER PROXY walks its run of the sweep axis -- the proxies after it in lower-x order whose lower x
// is not beyond its upper x (runLengthKernel's predicate, evaluated by the lanes as they go: no run lengths, no scan, no
// work-item decoding) -- 64 candidates at a time.  Only a pair with a moved member can be reported (reports()), so for a
// proxy that did not move the lanes skip every candidate that did not move either after one byte read: in a settled pile
// 6 % of the proxies move and nine tenths of the sweep's candidates cost nothing more.  A run longer than S2_LONG_RUN (the
// ground under a pile: every proxy is in its run) is finished by the whole grid in pairLongKernel.
#define S2_LONG_RUN 512
__global__ __launch_bounds__(S2_BLOCK) void pairWaveKernel(const s2amdShape* shapes, const unsigned char* moved, const int* sortedIdx,
															const float4* sortedBox, const unsigned char* sortedMoved, int n, const unsigned long long* existing,
															int existingCount, const unsigned long long* jointed, int jointedCount, unsigned long long* outKeys,
															unsigned int outCapacity, unsigned int* outCount, int* longList, unsigned int* longCount, GoneKeys gone)
{
	const int lane = (int)threadIdx.x & 63;
	const int wavesPerBlock = (int)blockDim.x >> 6;
	for (int i = (int)blockIdx.x * wavesPerBlock + ((int)threadIdx.x >> 6); i < n; i += (int)gridDim.x * wavesPerBlock)
	{
		const float4 bi = sortedBox[i];
		const float ux = bi.z;
		const bool mi = sortedMoved[i] != 0;
		for (int base = i + 1;; base += 64)
		{
			if (base - (i + 1) >= S2_LONG_RUN)
			{
				if (lane == 0)
				{
					longList[atomicAdd(longCount, 1u)] = i; // (at most n entries)
				}
				break;
			}
			const int j = base + lane;
			float4 bj = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			if (j < n)
			{
				bj = sortedBox[j];
			}
			const bool in = j < n && !(bj.x - ux > 0.0f);
			if (in && (mi || sortedMoved[j] != 0) && boxesOverlap(bi, bj))
			{
				// (pairTest repeats the overlap test on the shape records: the same values, the same answer)
				pairTest(sortedIdx[i], sortedIdx[j], shapes, moved, existing, existingCount, jointed, jointedCount, outKeys, outCapacity, outCount, gone);
			}
			if (!__all(in))
			{
				break; // (lower x ascends: the run ended inside these 64)
			}
		}
	}
}

// the rest of the long runs: every thread of the grid strides over each of them
__global__ __launch_bounds__(S2_BLOCK) void pairLongKernel(const s2amdShape* shapes, const unsigned char* moved, const int* sortedIdx,
															const float4* sortedBox, const unsigned char* sortedMoved, int n, const unsigned long long* existing,
															int existingCount, const unsigned long long* jointed, int jointedCount, unsigned long long* outKeys,
															unsigned int outCapacity, unsigned int* outCount, const int* longList, const unsigned int* longCount, GoneKeys gone)
{
	const unsigned int count = *longCount;
	for (unsigned int e = 0; e < count; ++e)
	{
		const int i = longList[e];
		const float4 bi = sortedBox[i];
		const float ux = bi.z;
		const bool mi = sortedMoved[i] != 0;
		for (int j = i + 1 + S2_LONG_RUN + (int)(blockIdx.x * blockDim.x + threadIdx.x); j < n; j += (int)(gridDim.x * blockDim.x))
		{
			const float4 bj = sortedBox[j];
			if (bj.x - ux > 0.0f)
			{
				break; // (ascending: everything this thread would visit later lies beyond the run as well)
			}
			if ((mi || sortedMoved[j] != 0) && boxesOverlap(bi, bj))
			{
				pairTest(sortedIdx[i], sortedIdx[j], shapes, moved, existing, existingCount, jointed, jointedCount, outKeys, outCapacity, outCount, gone);
			}
		}
	}
}

namespace
{
struct Scratch
{
	void* p = nullptr;
	size_t bytes = 0;
	~Scratch()
	{
		if (p)
		{
			(void)hipFree(p);
		}
	}
	hipError_t ensure(size_t need)
	{
		if (need <= bytes)
		{
			return hipSuccess;
		}
		if (p)
		{
			(void)hipFree(p);
			p = nullptr;
			bytes = 0;
		}
		hipError_t e = hipMalloc(&p, need);
		if (e == hipSuccess)
		{
			bytes = need;
		}
		return e;
	}
};

#define BP_TRY(expr)                                                                                                             \
	do                                                                                                                           \
	{                                                                                                                            \
		hipError_t _e = (expr);                                                                                                  \
		if (_e != hipSuccess)                                                                                                    \
		{                                                                                                                        \
			return s2amdFail(S2AMD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
		}                                                                                                                        \
	} while (0)

dim3 gridFor(size_t n)
{
	return dim3((unsigned)((n + S2_BLOCK - 1) / S2_BLOCK));
}
} // namespace

// ---- stage 1 on resident arrays (world.hip: s2amd_world_find_pairs) ----
// sort keys of every shape slot (free slots sort last), moved[] from the refit's `enlarged` flags
__global__ __launch_bounds__(S2_BLOCK) void residentShapeKeysKernel(const s2amdShape* shapes, int ns, uint32_t* keys, int* idx, unsigned char* moved)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ns)
	{
		return;
	}
	uint32_t key = 0xffffffffu;
	if (shapes[i].type != S2AMD_SHAPE_FREE)
	{
		uint32_t u = __float_as_uint(shapes[i].fatAABB[0]);
		key = (u & 0x80000000u) ? ~u : (u | 0x80000000u); // sortableFloat
	}
	keys[i] = key;
	idx[i] = i;
	moved[i] = (shapes[i].type != S2AMD_SHAPE_FREE && shapes[i].enlarged != 0) ? 1 : 0;
}

// the resident query's results to the host: count[0] pairs found (count[1]: the long runs' counter), the first `first` keys; both counters
// zero again for the next query
__global__ __launch_bounds__(S2_BLOCK) void publishPairsKernel(unsigned int* count, const unsigned long long* keys, unsigned int first, unsigned int* hostFound,
															   unsigned long long* hostKeys, const TreeViews* trees)
{
	const unsigned int found = count[0];
	const unsigned int n = found < first ? found : first;
	for (unsigned int i = threadIdx.x; i < n; i += blockDim.x)
	{
		hostKeys[i] = keys[i];
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		hostFound[0] = found;
		// (the device trees' error words travel with the pairs they ordered: a tree that lost its shape must not order anything quietly)
		hostFound[1] = trees != nullptr ? (unsigned int)(trees->t[0].state[2] | trees->t[1].state[2] | trees->t[2].state[2]) : 0u;
		count[0] = 0u, count[1] = 0u;
	}
}

// (min shape, max shape) of every live pair slot; free slots sort last and match nothing
__global__ __launch_bounds__(S2_BLOCK) void residentPairKeysKernel(const s2amdPairState* pairs, int nc, unsigned long long* keys, int* slots, int shapeBits)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nc)
	{
		return;
	}
	unsigned long long key = ~0ull;
	if (pairs[i].shapeA >= 0 && pairs[i].shapeB >= 0)
	{
		unsigned int a = (unsigned int)pairs[i].shapeA, b = (unsigned int)pairs[i].shapeB;
		key = ((unsigned long long)(a < b ? a : b) << shapeBits) | (a < b ? b : a);
	}
	keys[i] = key;
	slots[i] = i;
}

int findPairsResident(hipStream_t st, const s2amdShape* dS, int ns, int liveShapes, const s2amdPairState* dPairs, int nc,
					  const unsigned long long* dJointed, int jointedCount, int32_t* outPairs, int32_t pairCapacity, int32_t* pairCount, void** scratch,
					  size_t* scratchBytes, unsigned long long* sortedPairKeys, bool* sortedPairKeysValid, PairQueryGraph* cache, int mode,
					  const unsigned long long* pairLog, const int* pairLogSlots, const TreeViews* trees, const PairQueryHook* hook)
{
	// mode: S2_PAIRS_FULL the whole query; S2_PAIRS_WARM buffers + captured graph, nothing runs (s2amd_world_upload); S2_PAIRS_ENQUEUE
	// the query enqueued behind the caller's work, no wait (s2amd_world_step); S2_PAIRS_COLLECT the results of such a query, after the
	// caller's wait (s2amd_world_find_pairs)
	const bool warmOnly = mode == S2_PAIRS_WARM;
	int shapeBits = 1;
	while ((1ll << shapeBits) < (long long)std::max(ns, 2))
	{
		shapeBits += 1;
	}
	int* sortedPairSlots = (int*)(sortedPairKeys + nc); // (the caller's buffer: nc keys, then nc slots)
	const PairSetView gone{sortedPairSlots, pairLog, pairLogSlots, dPairs, shapeBits};
	*pairCount = 0;
	const int n = liveShapes;
	if (n < 2)
	{
		return S2AMD_OK;
	}
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	size_t tmpSort = 0, tmpScan = 0, tmpKeys = 0;
	BP_TRY(rocprim::radix_sort_pairs(nullptr, tmpSort, (uint32_t*)nullptr, (uint32_t*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)ns, 0, 32, st));
	BP_TRY(rocprim::exclusive_scan(nullptr, tmpScan, (unsigned int*)nullptr, (unsigned int*)nullptr, 0u, (size_t)n + 1, rocprim::plus<unsigned int>(), st));
	// (the device-side pair buffer is sized by the pool, not by the caller's buffer: the captured graph below depends on it, and
	// s2amd_world_upload captures that graph before any caller has shown its buffer -- and by the largest query that did not fit)
	size_t outCap = std::max((size_t)std::max(nc, 1024), cache->outCapWanted);
	BP_TRY(rocprim::radix_sort_pairs(nullptr, tmpKeys, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int*)nullptr, (int*)nullptr, std::max((size_t)nc, outCap), 0, 64, st));
	size_t tmpBytes = std::max(std::max(tmpSort, tmpScan), tmpKeys);
	size_t layout[] = {al((size_t)ns), al((size_t)ns * 4), al((size_t)ns * 4), al((size_t)ns * 4), al((size_t)ns * 4), al((size_t)n * 16),
					   al(((size_t)n + 1) * 4), al(((size_t)n + 1) * 4), al((size_t)nc * 8 + 8), al((size_t)nc * 8 + 8), al(outCap * 8), al(outCap * 8), al(outCap * 8), al(256),
					   al(tmpBytes + 256)};
	size_t total = 0;
	for (size_t b : layout)
	{
		total += b;
	}
	if (*scratchBytes < total)
	{
		// owned by the caller (the solver keeps it between calls)
		if (*scratch)
		{
			(void)hipFree(*scratch);
			*scratch = nullptr;
			*scratchBytes = 0;
		}
		BP_TRY(hipMalloc(scratch, total + total / 4));
		*scratchBytes = total + total / 4;
		cache->countAt = nullptr;
	}
	char* p = (char*)*scratch;
	size_t li = 0;
	auto take = [&]() {
		char* r = p;
		p += layout[li++];
		return r;
	};
	unsigned char* dMoved = (unsigned char*)take();
	uint32_t* dKeysIn = (uint32_t*)take();
	uint32_t* dKeysOut = (uint32_t*)take();
	int* dIdxIn = (int*)take();
	int* dIdxOut = (int*)take();
	float4* dBox = (float4*)take(); // the proxies' fat boxes in sweep order (gatherBoxesKernel)
	unsigned int* dRun = (unsigned int*)take();
	unsigned int* dOff = (unsigned int*)take();
	unsigned char* dMovedSorted = (unsigned char*)dOff; // (... and their move flags: the run offsets of the host-array route are not used here)
	unsigned long long* dExistingIn = (unsigned long long*)take();
	unsigned long long* dExisting = (unsigned long long*)take();
	unsigned long long* dOutA = (unsigned long long*)take();
	unsigned long long* dOutB = (unsigned long long*)take(); // the pairs in creation order, when the device holds the reference's trees
	unsigned long long* dCreationKeys = (unsigned long long*)take();
	unsigned int* dCount = (unsigned int*)take();
	void* dTmp = take();
	if (cache->countAt != (void*)dCount)
	{
		// (the counters start at zero and publishPairsKernel leaves them there; a block the layout has just put them in may hold anything)
		BP_TRY(hipMemsetAsync(dCount, 0, 256, st));
		cache->countAt = (void*)dCount;
	}
	int* dSlotsIn = (int*)dExisting; // (the sorted pair keys -- and their slots -- live in the caller's buffer: this block holds the unsorted slots)
	const bool ordered = trees != nullptr;
	const unsigned long long* dResult = ordered ? dOutB : dOutA;

	// the sorted keys of the live pairs only change when a contact is created or destroyed: the caller keeps them
	size_t tmp = tmpBytes + 256;
	if (nc > 0 && !*sortedPairKeysValid && mode != S2_PAIRS_COLLECT)
	{
		// (compact keys: 2 x shapeBits significant bits -- 30 at 20k shapes: four radix passes where the 64-bit form took eight; a free
		// slot's key, all ones, sorts behind every live one on those bits too)
		residentPairKeysKernel<<<gridFor((size_t)nc), dim3(S2_BLOCK), 0, st>>>(dPairs, nc, dExistingIn, dSlotsIn, shapeBits);
		BP_TRY(rocprim::radix_sort_pairs(dTmp, tmp, dExistingIn, sortedPairKeys, dSlotsIn, sortedPairSlots, (size_t)nc, 0, (unsigned int)std::min(2 * shapeBits, 64), st));
		*sortedPairKeysValid = true;
	}
	constexpr unsigned int kFirst = 2048;
	// one read-back: the count and the first keys (a step rarely creates more than a few contacts), into pinned memory
	if (cache->host == nullptr)
	{
		BP_TRY(hipHostMalloc((void**)&cache->host, 16 + (size_t)kFirst * 8, hipHostMallocDefault));
		memset(cache->host, 0, 16);
	}
	unsigned int* hostFound = (unsigned int*)cache->host;
	unsigned long long* hostKeys = (unsigned long long*)(cache->host + 16);
	unsigned int* hostFoundDev = nullptr;
	unsigned long long* hostKeysDev = nullptr;
	{
		char* dev = nullptr;
		if (hipHostGetDevicePointer((void**)&dev, cache->host, 0) == hipSuccess && dev != nullptr)
		{
			hostFoundDev = (unsigned int*)dev, hostKeysDev = (unsigned long long*)(dev + 16);
		}
		else
		{
			(void)hipGetLastError();
		}
	}
	// The query proper -- key generation, the sort of the proxies, the sweep kernels, the read-back -- is the same dozen launches
	// every step: the second time a sequence (arrays, sizes) comes along it is captured into a hipGraph and replayed from then on
	// (with the reference's trees on the device the query has a tail that is not part of the graph: the refit's enlarge pass -- which
	// waits for the tree rebuild running beside the step, an event of another stream -- the ranking of the pairs, the read-back)
	auto enqueueSet = [&]() -> int {
		if (hostFoundDev == nullptr)
		{
			BP_TRY(hipMemsetAsync(dCount, 0, 256, st)); // (else publishPairsKernel left them at zero, and the scratch was zeroed when it was made)
		}
		residentShapeKeysKernel<<<gridFor((size_t)ns), dim3(S2_BLOCK), 0, st>>>(dS, ns, dKeysIn, dIdxIn, dMoved);
		size_t t2 = tmpBytes + 256;
		BP_TRY(rocprim::radix_sort_pairs(dTmp, t2, dKeysIn, dKeysOut, dIdxIn, dIdxOut, (size_t)ns, 0, 32, st));
		gatherBoxesKernel<<<gridFor((size_t)n), dim3(S2_BLOCK), 0, st>>>(dS, dMoved, dIdxOut, n, dBox, dMovedSorted);
		// (dCount[0] = pairs found, dCount[1] = long runs: both zeroed above; the run-length array of the host-array route holds the long list)
		pairWaveKernel<<<dim3((unsigned)std::min((n + 3) / 4, 16384)), dim3(S2_BLOCK), 0, st>>>(dS, dMoved, dIdxOut, dBox, dMovedSorted, n, sortedPairKeys, nc, dJointed,
																								jointedCount, dOutA, (unsigned int)outCap, dCount, (int*)dRun, dCount + 1, gone);
		pairLongKernel<<<dim3(256), dim3(S2_BLOCK), 0, st>>>(dS, dMoved, dIdxOut, dBox, dMovedSorted, n, sortedPairKeys, nc, dJointed, jointedCount, dOutA,
															  (unsigned int)outCap, dCount, (const int*)dRun, dCount + 1, gone);
		BP_TRY(hipGetLastError());
		return S2AMD_OK;
	};
	auto enqueueTail = [&]() -> int {
		if (ordered)
		{
			if (hook != nullptr && hook->fn != nullptr)
			{
				hook->fn(hook->arg, st);
			}
			// the reference's creation order (tree_mirror.hip): (move-buffer position of the asking proxy, tree, reversed traversal rank)
			launchOrderPairs(st, trees, dS, dMoved, dOutA, dCount, (unsigned int)outCap, dCreationKeys, dOutB);
			BP_TRY(hipGetLastError());
		}
		if (hostFoundDev != nullptr)
		{
			// the count and the first keys straight into the pinned page, the counters zeroed for the next query: one small kernel where
			// a memset and two copies were three blit kernels (~4.5 us each, serial on the stream)
			publishPairsKernel<<<dim3(1), dim3(S2_BLOCK), 0, st>>>(dCount, dResult, (unsigned int)std::min<size_t>(kFirst, outCap), hostFoundDev, hostKeysDev, trees);
			BP_TRY(hipGetLastError());
		}
		else
		{
			BP_TRY(hipMemcpyAsync(hostFound, dCount, 4, hipMemcpyDeviceToHost, st));
			BP_TRY(hipMemcpyAsync(hostKeys, dResult, (size_t)std::min<size_t>(kFirst, outCap) * 8, hipMemcpyDeviceToHost, st));
		}
		return S2AMD_OK;
	};
	auto enqueue = [&]() -> int { // what a captured graph holds
		const int rcSet = enqueueSet();
		return rcSet != S2AMD_OK || ordered ? rcSet : enqueueTail();
	};
	bool ran = false;
	unsigned long long key = 1469598103934665603ull;
	{
		const unsigned long long words[] = {(unsigned long long)(uintptr_t)dS, (unsigned long long)ns, (unsigned long long)n, (unsigned long long)nc,
											(unsigned long long)(uintptr_t)dJointed, (unsigned long long)jointedCount, (unsigned long long)(uintptr_t)*scratch,
											(unsigned long long)(uintptr_t)sortedPairKeys, (unsigned long long)outCap, (unsigned long long)tmpBytes,
											(unsigned long long)shapeBits, (unsigned long long)(uintptr_t)pairLog, (unsigned long long)(uintptr_t)pairLogSlots, (unsigned long long)(uintptr_t)dPairs,
											(unsigned long long)(uintptr_t)trees};
		for (unsigned long long w : words)
		{
			key = (key ^ w) * 1099511628211ull;
		}
		key |= 1ull;
	}
	if (warmOnly)
	{
		// s2amd_world_upload ("prebuild_solver"): scratch, pinned read-back buffer and the sorted pair keys are in place; the graph of the
		// query is captured and instantiated now -- nothing of it runs -- so that no step pays for it (4.5 ms in the second query of a
		// new world, measured r5: the slower of the two start-up steps)
		if (!cache->disabled && (key != cache->key || cache->exec == nullptr))
		{
			if (cache->exec)
			{
				(void)hipGraphExecDestroy(cache->exec);
				cache->exec = nullptr;
			}
			hipGraph_t g = nullptr;
			hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
			int rcE = ce == hipSuccess ? enqueue() : S2AMD_E_DEVICE;
			hipError_t ee = ce == hipSuccess ? hipStreamEndCapture(st, &g) : ce;
			if (rcE == S2AMD_OK && ee == hipSuccess && g != nullptr && hipGraphInstantiate(&cache->exec, g, nullptr, nullptr, 0) == hipSuccess)
			{
				cache->key = key;
			}
			else
			{
				(void)hipGetLastError();
				cache->exec = nullptr;
				cache->key = 0; // (the first two queries go the usual way)
			}
			if (g)
			{
				(void)hipGraphDestroy(g);
			}
		}
		BP_TRY(hipStreamSynchronize(st));
		return S2AMD_OK;
	}
	if (mode == S2_PAIRS_COLLECT)
	{
		// (enqueued by an earlier call in S2_PAIRS_ENQUEUE mode; the caller has waited for the stream since)
	}
	else if (cache->disabled || (key != cache->key && key != cache->keySeen))
	{
		cache->keySeen = key;
		int rcE = enqueue();
		if (rcE)
		{
			return rcE;
		}
		ran = true;
	}
	else
	{
		if (key != cache->key || cache->exec == nullptr)
		{
			if (cache->exec)
			{
				(void)hipGraphExecDestroy(cache->exec);
				cache->exec = nullptr;
			}
			hipGraph_t g = nullptr;
			hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
			int rcE = ce == hipSuccess ? enqueue() : S2AMD_E_DEVICE;
			hipError_t ee = ce == hipSuccess ? hipStreamEndCapture(st, &g) : ce;
			if (rcE == S2AMD_OK && ee == hipSuccess && g != nullptr && hipGraphInstantiate(&cache->exec, g, nullptr, nullptr, 0) == hipSuccess)
			{
				cache->key = key;
			}
			else
			{
				// (a runtime that cannot capture the library sort: enqueue directly from now on)
				(void)hipGetLastError();
				cache->exec = nullptr;
				cache->disabled = true;
				cache->key = 0;
				if ((rcE = enqueue()) != 0)
				{
					return rcE;
				}
				ran = true;
			}
			if (g)
			{
				(void)hipGraphDestroy(g);
			}
		}
		if (cache->exec != nullptr && key == cache->key)
		{
			BP_TRY(hipGraphLaunch(cache->exec, st));
			ran = true;
		}
	}
	if (ran && ordered)
	{
		const int rcTail = enqueueTail();
		if (rcTail)
		{
			return rcTail;
		}
	}
	if (mode == S2_PAIRS_ENQUEUE)
	{
		return S2AMD_OK;
	}
	if (mode != S2_PAIRS_COLLECT)
	{
		BP_TRY(hipStreamSynchronize(st));
	}
	const unsigned int found = *hostFound;
	*pairCount = (int32_t)found;
	if (ordered && hostFound[1] != 0u)
	{
		return s2amdFail(S2AMD_E_DEVICE, "the device's broad-phase trees reported error " + std::to_string(hostFound[1]) +
											 " (tree_mirror.hip): the order of the new pairs cannot be trusted -- upload the world and its trees again");
	}
	if ((size_t)found > outCap)
	{
		// more new pairs than the device-side buffer holds (a world with few contact slots and many bodies landing at once): the
		// buffer grows to fit and the query runs again, whole -- its inputs (move flags, pair slots) are what they were.  Only a
		// caller's buffer that is too small is the caller's to fix.
		cache->outCapWanted = (size_t)found + (size_t)found / 4;
		cache->key = 0, cache->keySeen = 0;
		return findPairsResident(st, dS, ns, liveShapes, dPairs, nc, dJointed, jointedCount, outPairs, pairCapacity, pairCount, scratch, scratchBytes, sortedPairKeys,
								 sortedPairKeysValid, cache, S2_PAIRS_FULL, pairLog, pairLogSlots, trees, nullptr);
	}
	if ((int64_t)found > (int64_t)pairCapacity)
	{
		return s2amdFail(S2AMD_E_CAPACITY, "pair buffer too small: " + std::to_string(found) + " pairs found");
	}
	if (found == 0)
	{
		return S2AMD_OK;
	}
	std::vector<unsigned long long> out(hostKeys, hostKeys + std::min<size_t>(std::min<size_t>(kFirst, outCap), found));
	out.resize(found);
	if (found > kFirst)
	{
		BP_TRY(hipMemcpyAsync(out.data() + kFirst, dResult + kFirst, (size_t)(found - kFirst) * 8, hipMemcpyDeviceToHost, st));
		BP_TRY(hipStreamSynchronize(st));
	}
	if (!ordered)
	{
		std::sort(out.begin(), out.end()); // deterministic output order: by (A, B)
	}
	for (unsigned int i = 0; i < found; ++i)
	{
		outPairs[2 * i] = (int32_t)(out[i] >> 32);
		outPairs[2 * i + 1] = (int32_t)(out[i] & 0xffffffffu);
	}
	return S2AMD_OK;
}

// resident arrays (world.hip)
void launchStage4(hipStream_t st, s2amdBody* bodies, int bodyCapacity, s2amdShape* shapes, int shapeCapacity, float* origins, int* summary, const unsigned int* stepFailed)
{
	if (bodyCapacity <= 0)
	{
		return;
	}
	int shapeBlocks = (shapeCapacity + S2_BLOCK - 1) / S2_BLOCK, bodyBlocks = (bodyCapacity + S2_BLOCK - 1) / S2_BLOCK;
	stage4Kernel<<<dim3((unsigned)(shapeBlocks + bodyBlocks)), dim3(S2_BLOCK), 0, st>>>(bodies, bodyCapacity, shapes, shapeCapacity, (float2*)origins,
																					shapeBlocks, summary, stepFailed);
}

#pragma GCC visibility push(default)
extern "C"
{

int s2amd_refit_shapes(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, s2amdShape* shapes, int32_t shapeCapacity, float* origins)
{
	if (!solver || bodyCapacity < 0 || shapeCapacity < 0 || (bodyCapacity > 0 && (!bodies || !origins)) || (shapeCapacity > 0 && !shapes))
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument");
	}
	if (bodyCapacity == 0 || shapeCapacity == 0)
	{
		return S2AMD_OK;
	}
	BP_TRY(hipSetDevice(s2amdDevice(solver)));
	hipStream_t st = s2amdStream(solver);
	Scratch buf;
	size_t bBytes = (size_t)bodyCapacity * sizeof(s2amdBody), sBytes = (size_t)shapeCapacity * sizeof(s2amdShape);
	size_t oBytes = (size_t)bodyCapacity * sizeof(float2);
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	BP_TRY(buf.ensure(al(bBytes) + al(sBytes) + al(oBytes)));
	char* base = (char*)buf.p;
	s2amdBody* dB = (s2amdBody*)base;
	s2amdShape* dS = (s2amdShape*)(base + al(bBytes));
	float2* dO = (float2*)(base + al(bBytes) + al(sBytes));
	BP_TRY(hipMemcpyAsync(dB, bodies, bBytes, hipMemcpyHostToDevice, st));
	BP_TRY(hipMemcpyAsync(dS, shapes, sBytes, hipMemcpyHostToDevice, st));
	BP_TRY(hipMemcpyAsync(dO, origins, oBytes, hipMemcpyHostToDevice, st));
	bodyOriginsKernel<<<gridFor((size_t)bodyCapacity), dim3(S2_BLOCK), 0, st>>>(dB, bodyCapacity, dO);
	refitShapesKernel<<<gridFor((size_t)shapeCapacity), dim3(S2_BLOCK), 0, st>>>(dB, bodyCapacity, dS, shapeCapacity, dO);
	BP_TRY(hipGetLastError());
	BP_TRY(hipMemcpyAsync(shapes, dS, sBytes, hipMemcpyDeviceToHost, st));
	BP_TRY(hipMemcpyAsync(origins, dO, oBytes, hipMemcpyDeviceToHost, st));
	BP_TRY(hipStreamSynchronize(st));
	return S2AMD_OK;
}

int s2amd_find_pairs(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdShape* shapes, int32_t shapeCapacity,
					 const uint8_t* moved, const int32_t* existingPairs, int32_t existingPairCount, const s2amdJoint* joints, int32_t jointCapacity,
					 int32_t* outPairs, int32_t pairCapacity, int32_t* pairCount)
{
	(void)bodies;
	(void)bodyCapacity;
	if (!solver || !pairCount || shapeCapacity < 0 || existingPairCount < 0 || jointCapacity < 0 || pairCapacity < 0 ||
		(shapeCapacity > 0 && (!shapes || !moved)) || (existingPairCount > 0 && !existingPairs) || (jointCapacity > 0 && !joints) ||
		(pairCapacity > 0 && !outPairs))
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument");
	}
	*pairCount = 0;
	// live shapes, keyed by the lower x of their fat AABB
	std::vector<uint32_t> keys;
	std::vector<int> idx;
	keys.reserve((size_t)shapeCapacity);
	idx.reserve((size_t)shapeCapacity);
	for (int i = 0; i < shapeCapacity; ++i)
	{
		if (shapes[i].type != S2AMD_SHAPE_FREE)
		{
			keys.push_back(sortableFloat(shapes[i].fatAABB[0]));
			idx.push_back(i);
		}
	}
	const int n = (int)idx.size();
	if (n < 2)
	{
		return S2AMD_OK;
	}
	std::vector<unsigned long long> existing((size_t)existingPairCount), jointed;
	for (int e = 0; e < existingPairCount; ++e)
	{
		unsigned int a = (unsigned int)existingPairs[2 * e], b = (unsigned int)existingPairs[2 * e + 1];
		existing[(size_t)e] = ((unsigned long long)std::min(a, b) << 32) | std::max(a, b);
	}
	std::sort(existing.begin(), existing.end());
	for (int j = 0; j < jointCapacity; ++j)
	{
		if (joints[j].type != S2AMD_JOINT_FREE && joints[j].bodyA >= 0 && joints[j].bodyB >= 0)
		{
			unsigned int a = (unsigned int)joints[j].bodyA, b = (unsigned int)joints[j].bodyB;
			jointed.push_back(((unsigned long long)std::min(a, b) << 32) | std::max(a, b));
		}
	}
	std::sort(jointed.begin(), jointed.end());

	BP_TRY(hipSetDevice(s2amdDevice(solver)));
	hipStream_t st = s2amdStream(solver);
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	size_t sBytes = (size_t)shapeCapacity * sizeof(s2amdShape);
	size_t tmpSort = 0, tmpScan = 0;
	BP_TRY(rocprim::radix_sort_pairs(nullptr, tmpSort, (uint32_t*)nullptr, (uint32_t*)nullptr, (int*)nullptr, (int*)nullptr, (size_t)n, 0, 32, st));
	BP_TRY(rocprim::exclusive_scan(nullptr, tmpScan, (unsigned int*)nullptr, (unsigned int*)nullptr, 0u, (size_t)n + 1, rocprim::plus<unsigned int>(), st));
	size_t tmpBytes = std::max(tmpSort, tmpScan);
	size_t outCap = (size_t)std::max(pairCapacity, 1024);
	size_t layout[] = {al(sBytes),
					   al((size_t)shapeCapacity),					  // moved
					   al((size_t)n * 4), al((size_t)n * 4),		  // keys in / out
					   al((size_t)n * 4), al((size_t)n * 4),		  // idx in / out
					   al((size_t)n * 4),							  // sorted lower x
					   al(((size_t)n + 1) * 4), al(((size_t)n + 1) * 4), // run length / offsets
					   al(existing.size() * 8 + 8), al(jointed.size() * 8 + 8),
					   al(outCap * 8), al(outCap * 8), al(256), al(tmpBytes + 256)};
	size_t total = 0;
	for (size_t b : layout)
	{
		total += b;
	}
	Scratch buf;
	BP_TRY(buf.ensure(total));
	char* p = (char*)buf.p;
	size_t li = 0;
	auto take = [&]() {
		char* r = p;
		p += layout[li++];
		return r;
	};
	s2amdShape* dS = (s2amdShape*)take();
	unsigned char* dMoved = (unsigned char*)take();
	uint32_t* dKeysIn = (uint32_t*)take();
	uint32_t* dKeysOut = (uint32_t*)take();
	int* dIdxIn = (int*)take();
	int* dIdxOut = (int*)take();
	float* dLowerX = (float*)take();
	unsigned int* dRun = (unsigned int*)take();
	unsigned int* dOff = (unsigned int*)take();
	unsigned long long* dExisting = (unsigned long long*)take();
	unsigned long long* dJointed = (unsigned long long*)take();
	unsigned long long* dOutA = (unsigned long long*)take();
	unsigned long long* dOutB = (unsigned long long*)take();
	unsigned int* dCount = (unsigned int*)take();
	void* dTmp = take();

	BP_TRY(hipMemcpyAsync(dS, shapes, sBytes, hipMemcpyHostToDevice, st));
	BP_TRY(hipMemcpyAsync(dMoved, moved, (size_t)shapeCapacity, hipMemcpyHostToDevice, st));
	BP_TRY(hipMemcpyAsync(dKeysIn, keys.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
	BP_TRY(hipMemcpyAsync(dIdxIn, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
	if (!existing.empty())
	{
		BP_TRY(hipMemcpyAsync(dExisting, existing.data(), existing.size() * 8, hipMemcpyHostToDevice, st));
	}
	if (!jointed.empty())
	{
		BP_TRY(hipMemcpyAsync(dJointed, jointed.data(), jointed.size() * 8, hipMemcpyHostToDevice, st));
	}
	BP_TRY(hipMemsetAsync(dCount, 0, 256, st));
	BP_TRY(hipMemsetAsync(dRun, 0, ((size_t)n + 1) * 4, st));

	size_t tmp = tmpBytes + 256;
	BP_TRY(rocprim::radix_sort_pairs(dTmp, tmp, dKeysIn, dKeysOut, dIdxIn, dIdxOut, (size_t)n, 0, 32, st));
	// sorted lower x as floats (gather)
	gatherLowerXKernel<<<gridFor((size_t)n), dim3(S2_BLOCK), 0, st>>>(dS, dIdxOut, n, dLowerX);
	runLengthKernel<<<gridFor((size_t)n), dim3(S2_BLOCK), 0, st>>>(dS, dIdxOut, dLowerX, n, dRun);
	tmp = tmpBytes + 256;
	BP_TRY(rocprim::exclusive_scan(dTmp, tmp, dRun, dOff, 0u, (size_t)n + 1, rocprim::plus<unsigned int>(), st));
	unsigned int work = 0;
	BP_TRY(hipMemcpyAsync(&work, dOff + n, 4, hipMemcpyDeviceToHost, st));
	BP_TRY(hipStreamSynchronize(st));
	unsigned int found = 0;
	if (work > 0)
	{
		pairKernel<<<gridFor((size_t)work), dim3(S2_BLOCK), 0, st>>>(dS, dMoved, dIdxOut, dOff, n, work, dExisting, (int)existing.size(), dJointed,
																	  (int)jointed.size(), dOutA, (unsigned int)outCap, dCount);
		BP_TRY(hipGetLastError());
		BP_TRY(hipMemcpyAsync(&found, dCount, 4, hipMemcpyDeviceToHost, st));
		BP_TRY(hipStreamSynchronize(st));
	}
	*pairCount = (int32_t)found;
	if ((int64_t)found > (int64_t)pairCapacity)
	{
		return s2amdFail(S2AMD_E_CAPACITY, "pair buffer too small: " + std::to_string(found) + " pairs found");
	}
	if (found == 0)
	{
		return S2AMD_OK;
	}
	// deterministic output: sort the (A << 32 | B) keys
	size_t tmpKeys = 0;
	BP_TRY(rocprim::radix_sort_keys(nullptr, tmpKeys, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)found, 0, 64, st));
	Scratch tmp2;
	BP_TRY(tmp2.ensure(tmpKeys + 256));
	BP_TRY(rocprim::radix_sort_keys(tmp2.p, tmpKeys, dOutA, dOutB, (size_t)found, 0, 64, st));
	std::vector<unsigned long long> out((size_t)found);
	BP_TRY(hipMemcpyAsync(out.data(), dOutB, (size_t)found * 8, hipMemcpyDeviceToHost, st));
	BP_TRY(hipStreamSynchronize(st));
	for (unsigned int i = 0; i < found; ++i)
	{
		outPairs[2 * i] = (int32_t)(out[i] >> 32);
		outPairs[2 * i + 1] = (int32_t)(out[i] & 0xffffffffu);
	}
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop

S2_DEFINE_WARM(broadphase)
