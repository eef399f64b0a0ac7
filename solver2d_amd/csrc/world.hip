// Resident world: stages 3 and 4 of s2World_Step (src/world.c:132-168, :259-301) chained with the solve in HBM.
//
// The stage entry points of include/solver2d_amd.h (s2amd_update_contacts, s2amd_refit_shapes) take host arrays in
// and out, which costs more than their kernels.  Here the shapes, the narrow-phase pair states and the body origins
// stay on the device beside the solver's wire bodies / contacts / joints, and one s2amd_world_step runs
//
//     update contacts + counters (narrowphase.hip)  ->  s2Solve_* (solver_step.cpp: doStep)  ->  refit + origins + force reset (broadphase.hip)
//
// on them.  What comes back per step is 32 bytes of counters.  The constraint graph structure (islands, colours,
// strips) is still built on the host, so the point counts of the new manifolds are fetched (one byte per contact
// slot) in the steps where the summary kernel saw one of them change.
#include "solver_internal.h"

#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#define S2_BLOCK 256

namespace
{

// int[8] on the device: the narrow-phase kernel fills [0..3], the stage-4 kernel [4] (launch.h)
struct WorldSummary
{
	int separated; // pairs whose fat AABBs parted this step
	int active;	   // manifolds with at least one point
	int flips;	   // manifolds that went between zero and non-zero points: the constraint graph changed
	int moves;	   // manifolds whose point count changed at all
	int enlarged;  // shapes whose fat AABB was re-inflated by the refit: the broad phase has to look at them
	int watchedFlips; // flips of manifolds on hub bodies: those are changes of the constraint graph
	int pad[2];
};

// s2amd_world_set_contacts: staged records into their slots
__global__ __launch_bounds__(S2_BLOCK) void scatterContactsKernel(const int32_t* slots, int n, const s2amdContact* newContacts, const s2amdPairState* newPairs,
																  s2amdContact* contacts, s2amdPairState* pairs, uint8_t* pointBytes, int32_t* status)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		int k = slots[i];
		contacts[k] = newContacts[i];
		pairs[k] = newPairs[i];
		int pc = newContacts[i].pointCount;
		pointBytes[k] = (uint8_t)(pc > 0 ? pc : 0);
		status[k] = newPairs[i].shapeA >= 0 ? S2AMD_PAIR_UPDATED : S2AMD_PAIR_FREE;
	}
}

dim3 gridFor(size_t n)
{
	return dim3((unsigned)((n + S2_BLOCK - 1) / S2_BLOCK));
}

// The pair query has consumed the move buffer (src/broad_phase.c: s2UpdateBroadPhasePairs ends by clearing moveArray and
// moveSet): every `enlarged` flag goes back to zero.  Stage 4 only rewrites the flags of non-static bodies' shapes, so
// without this a static shape uploaded "in the move buffer" (as after its creation) would stay there for good -- counted
// as moved every step, and, never querying itself, suppressing its pairs with moved proxies of a lower key.
__global__ __launch_bounds__(S2_BLOCK) void clearMovedKernel(s2amdShape* shapes, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		shapes[i].enlarged = 0;
	}
}

// one byte per pair slot for the host's structure build: 0xff = free (its pair separated or was never there), else the
// manifold's point count -- what syncDeadSlots and fetchPointCounts need, as ONE contiguous copy
__global__ __launch_bounds__(S2_BLOCK) void slotBytesKernel(const s2amdPairState* pairs, const uint8_t* pointBytes, int n, uint8_t* out)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		out[i] = pairs[i].shapeA < 0 ? (uint8_t)0xff : pointBytes[i];
	}
}

// the boxes of every shape slot, packed for the host (s2amd_world_download_boxes)
__global__ __launch_bounds__(S2_BLOCK) void shapeBoxesKernel(const s2amdShape* shapes, int n, s2amdShapeBox* out)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		s2amdShapeBox b;
		for (int k = 0; k < 4; ++k)
		{
			b.aabb[k] = shapes[i].aabb[k];
			b.fatAABB[k] = shapes[i].fatAABB[k];
		}
		b.enlarged = shapes[i].enlarged;
		out[i] = b;
	}
}

// s2amd_world_download_step: {origin, rot} of every body slot
__global__ __launch_bounds__(S2_BLOCK) void stepPosesKernel(const s2amdBody* bodies, const float* origins, int n, float4* out)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		out[i] = make_float4(origins[2 * i], origins[2 * i + 1], bodies[i].rot[0], bodies[i].rot[1]);
	}
}

// ... and the re-inflated shapes in the caller's refit order: an ordered compaction in two launches over tiles of 256 entries of
// the order array -- per-tile counts, then every tile adds up the counts before it (a few hundred at most) and writes its
// entries at their ranks.  out[0] = count, entries from out + 4 words on; counts: one int per tile.
S2_DEV bool movedAt(const s2amdShape* shapes, int shapeCapacity, const int* order, int n, int r, int& sh)
{
	if (r >= n)
	{
		return false;
	}
	sh = order ? order[r] : r;
	return sh >= 0 && sh < shapeCapacity && shapes[sh].type != S2AMD_SHAPE_FREE && shapes[sh].enlarged != 0;
}

__global__ __launch_bounds__(S2_BLOCK) void movedCountKernel(const s2amdShape* shapes, int shapeCapacity, const int* order, int n, int* counts)
{
	__shared__ int waves[S2_BLOCK / 64];
	int sh;
	const bool mine = movedAt(shapes, shapeCapacity, order, n, (int)(blockIdx.x * blockDim.x + threadIdx.x), sh);
	const unsigned long long mask = __ballot(mine);
	if ((threadIdx.x & 63) == 0)
	{
		waves[threadIdx.x >> 6] = __popcll(mask);
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		int total = 0;
		for (int w = 0; w < S2_BLOCK / 64; ++w)
		{
			total += waves[w];
		}
		counts[blockIdx.x] = total;
	}
}

// (blocks behind the list's tiles, if any, write the step's poses: stepPosesKernel's work without its launch)
__global__ __launch_bounds__(S2_BLOCK) void movedWriteKernel(const s2amdShape* shapes, int shapeCapacity, const int* order, int n, const int* counts, int32_t* out,
															int capacity, int tiles, const s2amdBody* bodies, const float* origins, int nb, float4* poses)
{
	if ((int)blockIdx.x >= tiles)
	{
		const int i = ((int)blockIdx.x - tiles) * S2_BLOCK + (int)threadIdx.x;
		if (i < nb)
		{
			poses[i] = make_float4(origins[2 * i], origins[2 * i + 1], bodies[i].rot[0], bodies[i].rot[1]);
		}
		return;
	}
	__shared__ int waves[S2_BLOCK / 64];
	__shared__ int base;
	int sh = 0;
	const bool mine = movedAt(shapes, shapeCapacity, order, n, (int)(blockIdx.x * blockDim.x + threadIdx.x), sh);
	const unsigned long long mask = __ballot(mine);
	const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
	if (lane == 0)
	{
		waves[wave] = __popcll(mask);
	}
	if (threadIdx.x < 64)
	{
		// the tiles before this one (and, in the last tile, the total)
		int partial = 0;
		for (int b = lane; b < (int)blockIdx.x; b += 64)
		{
			partial += counts[b];
		}
		for (int d = 32; d > 0; d >>= 1)
		{
			partial += __shfl_xor(partial, d);
		}
		if (lane == 0)
		{
			base = partial;
		}
	}
	__syncthreads();
	int at = base;
	for (int w = 0; w < wave; ++w)
	{
		at += waves[w];
	}
	at += __popcll(mask & ((1ull << lane) - 1ull));
	if ((int)blockIdx.x == tiles - 1 && threadIdx.x == blockDim.x - 1)
	{
		out[0] = at + (mine ? 1 : 0);
	}
	if (mine && at < capacity)
	{
		s2amdMovedBox e;
		e.shape = sh;
		for (int k = 0; k < 4; ++k)
		{
			e.fatAABB[k] = shapes[sh].fatAABB[k];
		}
		((s2amdMovedBox*)(out + 4))[at] = e;
	}
}

// manifold.constraintIndex from the resident point counts: exclusive scan of "has points" over the pool (the reference's
// gather order, e.g. src/solve_tgs_soft.c:162-179), -1 for the slots the gather skips
struct HasPoints
{
	const uint8_t* pointBytes;
	__host__ __device__ int operator()(int i) const { return pointBytes[i] > 0 ? 1 : 0; }
};

__global__ __launch_bounds__(S2_BLOCK) void writeConstraintIndexFromScanKernel(s2amdContact* contacts, int n, const uint8_t* pointBytes, const int* scanned)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		contacts[i].constraintIndex = pointBytes[i] > 0 ? scanned[i] : -1;
	}
}

// the step's counters to the host (a mapped page the kernels would store into and the host spin on was measured: same
// 0.332 ms per step, hipStreamSynchronize already spins); `reset`: the last read-back of a step clears them for the next
// The poses and the re-inflated boxes of the step that was just enqueued, on their way to pinned host memory behind stage 4: what
// s2amd_world_download_step hands out.  The number of boxes is not known yet: room for twice the last step's + 2,048 is copied, a
// step that re-inflates more falls back to the explicit read-back of that call.
int enqueueStepBack(s2amdSolver* s)
{
	const int nb = s->bodyCapacity, ns = s->shapeCapacity;
	const int guess = std::min(ns, 2 * s->lastMovedCount + 2048);
	const size_t headBytes = 16 + (size_t)std::max(ns, 1) * sizeof(s2amdMovedBox);
	const size_t poseOffset = (headBytes + 255) & ~size_t(255);
	int rc = s->dStepBack.ensure(poseOffset + (size_t)std::max(nb, 1) * sizeof(float4));
	if (rc)
	{
		return rc;
	}
	const size_t hostPose = ((16 + (size_t)guess * sizeof(s2amdMovedBox)) + 255) & ~size_t(255);
	const size_t hostBytes = hostPose + (size_t)nb * sizeof(float4);
	if (s->hostStepBackBytes < hostBytes)
	{
		if (s->hostStepBack)
		{
			(void)hipHostFree(s->hostStepBack);
			s->hostStepBack = nullptr, s->hostStepBackBytes = 0;
		}
		HIP_TRY(hipHostMalloc((void**)&s->hostStepBack, hostBytes + hostBytes / 2, hipHostMallocDefault));
		s->hostStepBackBytes = hostBytes + hostBytes / 2;
	}
	char* base = (char*)s->dStepBack.p;
	const int n = s->refitOrderCount;
	const int tiles = (n + S2_BLOCK - 1) / S2_BLOCK;
	if ((rc = s->dScanTmp.ensure((size_t)std::max(tiles, 1) * sizeof(int))) != 0)
	{
		return rc;
	}
	movedCountKernel<<<dim3((unsigned)tiles), dim3(S2_BLOCK), 0, s->stream>>>((const s2amdShape*)s->dShapes.p, ns, (const int*)s->dRefitOrder.p, n, (int*)s->dScanTmp.p);
	movedWriteKernel<<<dim3((unsigned)(tiles + (nb + S2_BLOCK - 1) / S2_BLOCK)), dim3(S2_BLOCK), 0, s->stream>>>(
		(const s2amdShape*)s->dShapes.p, ns, (const int*)s->dRefitOrder.p, n, (const int*)s->dScanTmp.p, (int32_t*)base, ns, tiles, (const s2amdBody*)s->dBodies.p,
		(const float*)s->dOrigins.p, nb, (float4*)(base + poseOffset));
	HIP_TRY(hipGetLastError());
	s->stepBackListFresh = true; // (launchTreeEnlarge: the re-inflated shapes are a list on the device from here on)
	HIP_TRY(hipMemcpyAsync(s->hostStepBack, base, 16 + (size_t)guess * sizeof(s2amdMovedBox), hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipMemcpyAsync(s->hostStepBack + hostPose, base + poseOffset, (size_t)nb * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
	s->stepBackBoxes = guess;
	s->stepBackPoseOffset = hostPose;
	return S2AMD_OK;
}

// The step's counters to the host: one small kernel writes them into the pinned page (mapped for the device) and, at the end of a
// step, zeroes them for the next -- a copy and a memset were two blit kernels of ~4.5 us each, serial on the stream like everything
// else a step enqueues (r5: the small copies were 38 us of a churn step's device time).
__global__ void publishSummaryKernel(int* summary, int* host, int reset)
{
	const int i = (int)threadIdx.x;
	if (i < 8)
	{
		host[i] = summary[i];
		if (reset)
		{
			summary[i] = 0;
		}
	}
}
int fetchSummary(s2amdSolver* s, int reset)
{
	int* hostDev = nullptr;
	if (s->hostWorldSummary && hipHostGetDevicePointer((void**)&hostDev, s->hostWorldSummary, 0) == hipSuccess && hostDev != nullptr)
	{
		publishSummaryKernel<<<dim3(1), dim3(64), 0, s->stream>>>((int*)s->dWorldSummary.p, hostDev, reset);
	}
	else
	{
		(void)hipGetLastError();
		HIP_TRY(hipMemcpyAsync(s->hostWorldSummary, s->dWorldSummary.p, 8 * sizeof(int), hipMemcpyDeviceToHost, s->stream));
		if (reset)
		{
			HIP_TRY(hipMemsetAsync(s->dWorldSummary.p, 0, 8 * sizeof(int), s->stream));
		}
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

} // namespace

int refreshConstraintIndexOnDevice(s2amdSolver* s)
{
	const int nc = s->contactCapacity;
	if (nc <= 0 || !s->worldResident)
	{
		return S2AMD_OK;
	}
	bool grew = false;
	int rc = s->dGatherIndex.ensure((size_t)nc * sizeof(int), &grew);
	if (rc)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
		s->gatherIndexDirty = true;
	}
	auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0), HasPoints{(const uint8_t*)s->dPointBytes.p});
	size_t tmp = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, flags, (int*)s->dGatherIndex.p, 0, (size_t)nc, rocprim::plus<int>(), s->stream));
	if ((rc = s->dScanTmp.ensure(std::max<size_t>(tmp, 256))) != 0)
	{
		return rc;
	}
	HIP_TRY(rocprim::exclusive_scan(s->dScanTmp.p, tmp, flags, (int*)s->dGatherIndex.p, 0, (size_t)nc, rocprim::plus<int>(), s->stream));
	writeConstraintIndexFromScanKernel<<<gridFor((size_t)nc), dim3(S2_BLOCK), 0, s->stream>>>((s2amdContact*)s->dContacts.p, nc, (const uint8_t*)s->dPointBytes.p,
																							   (const int*)s->dGatherIndex.p);
	HIP_TRY(hipGetLastError());
	s->gatherIndexDirty = true; // dGatherIndex no longer holds what the host would compute from its own (older) point counts
	return S2AMD_OK;
}

int syncDeadSlots(s2amdSolver* s)
{
	s->deadUnknown = false;
	const int nc = s->contactCapacity;
	if (!s->worldResident || nc <= 0)
	{
		return S2AMD_OK;
	}
	HIP_TRY(hipSetDevice(s->device));
	static const bool debugAsync = getenv("S2AMD_DEBUG_ASYNC") != nullptr;
	const double tq0 = debugAsync ? nowMs() : 0.0;
	int rc = s->dSlotBytes.ensure(std::max<size_t>((size_t)nc, 256));
	if (rc)
	{
		return rc;
	}
	s->hSlotBytes.resize((size_t)nc);
	const double tq1 = debugAsync ? nowMs() : 0.0;
	slotBytesKernel<<<gridFor((size_t)nc), dim3(S2_BLOCK), 0, s->stream>>>((const s2amdPairState*)s->dPairs.p, (const uint8_t*)s->dPointBytes.p, nc,
																			 (uint8_t*)s->dSlotBytes.p);
	HIP_TRY(hipGetLastError());
	if (s->hostSlotStage && s->hostSlotStageBytes >= (size_t)nc)
	{
		HIP_TRY(hipMemcpyAsync(s->hostSlotStage, s->dSlotBytes.p, (size_t)nc, hipMemcpyDeviceToHost, s->stream));
		HIP_TRY(hipStreamSynchronize(s->stream));
		memcpy(s->hSlotBytes.data(), s->hostSlotStage, (size_t)nc);
	}
	else
	{
		HIP_TRY(hipMemcpyAsync(s->hSlotBytes.data(), s->dSlotBytes.p, (size_t)nc, hipMemcpyDeviceToHost, s->stream));
		HIP_TRY(hipStreamSynchronize(s->stream));
	}
	const double tq2 = debugAsync ? nowMs() : 0.0;
	for (int i = 0; i < nc; ++i)
	{
		if (s->hContactEdge[(size_t)i] && s->hSlotBytes[(size_t)i] == 0xff)
		{
			s->hContactDead[(size_t)i] = 1;
		}
	}
	if (debugAsync)
	{
		fprintf(stderr, "[s2amd]   syncDeadSlots: buffers %.3f ms, kernel + copy + wait %.3f, scan %.3f\n", tq1 - tq0, tq2 - tq1, nowMs() - tq2);
	}
	s->slotBytesFresh = true; // (fetchPointCounts right behind this call, the hub rule's, needs no second copy)
	return S2AMD_OK;
}

int fetchPointCounts(s2amdSolver* s)
{
	const int nc = s->contactCapacity;
	if (!s->worldResident || nc <= 0)
	{
		return S2AMD_OK;
	}
	HIP_TRY(hipSetDevice(s->device));
	s->hPointBytes.resize((size_t)nc);
	if (s->slotBytesFresh && (int)s->hSlotBytes.size() == nc)
	{
		for (int i = 0; i < nc; ++i)
		{
			s->hPointBytes[(size_t)i] = s->hSlotBytes[(size_t)i] == 0xff ? 0 : s->hSlotBytes[(size_t)i];
		}
	}
	else if (s->hostSlotStage && s->hostSlotStageBytes >= (size_t)nc)
	{
		HIP_TRY(hipMemcpyAsync(s->hostSlotStage, s->dPointBytes.p, (size_t)nc, hipMemcpyDeviceToHost, s->stream));
		HIP_TRY(hipStreamSynchronize(s->stream));
		memcpy(s->hPointBytes.data(), s->hostSlotStage, (size_t)nc);
	}
	else
	{
		HIP_TRY(hipMemcpyAsync(s->hPointBytes.data(), s->dPointBytes.p, (size_t)nc, hipMemcpyDeviceToHost, s->stream));
		HIP_TRY(hipStreamSynchronize(s->stream));
	}
	for (int i = 0; i < nc; ++i)
	{
		s->hContactPoints[(size_t)i] = s->hPointBytes[(size_t)i];
	}
	s->slotBytesFresh = false;
	s->pointCountsFresh = true;
	return S2AMD_OK;
}

// the stage-1 pair query's scratch, sorted pair keys and captured graph, made ahead of the first query (s2amd_world_upload with
// "prebuild_solver"; again when the device gets the trees that order the query's pairs: s2amd_world_set_tree)
int worldWarmPairQuery(s2amdSolver* s)
{
	if (s->liveShapes < 2)
	{
		return S2AMD_OK;
	}
	int32_t none = 0;
	int rc = s->dPairKeys.ensure(std::max<size_t>((size_t)s->contactCapacity * 12, 256));
	if (rc)
	{
		return rc;
	}
	return findPairsResident(s->stream, (const s2amdShape*)s->dShapes.p, s->shapeCapacity, s->liveShapes, (const s2amdPairState*)s->dPairs.p, s->contactCapacity,
							 (const unsigned long long*)s->dJointedKeys.p, s->jointedCount, nullptr, 0, &none, &s->dPairScratch.p, &s->dPairScratch.bytes,
							 (unsigned long long*)s->dPairKeys.p, &s->pairKeysValid, &s->pairQuery, S2_PAIRS_WARM, (const unsigned long long*)s->dPairLog.p,
							 (const int*)((const unsigned long long*)s->dPairLog.p + S2_PAIR_LOG_ENTRIES + 1), treesViews(s));
}

#pragma GCC visibility push(default)
extern "C"
{

// The small directory of the pair set (broadphase.hip: PairSetView::log): the contacts s2amd_world_set_contacts has created since the big
// one was sorted -- [0] = entries, keys ascending, then (behind S2_PAIR_LOG_ENTRIES + 1 keys) their slots --, kept by the host, read by
// the pair query.  Destroyed pairs need no entry anywhere: the query checks what a directory entry's slot holds now.
static void pairLogReset(s2amdSolver* s)
{
	if (s->hostPairLog)
	{
		s->hostPairLog[0] = 0ull;
		s->pairLogDirty = true;
	}
}
static void pairKeysStale(s2amdSolver* s)
{
	s->pairKeysValid = false; // (the next query sorts the keys of the pair slots as they stand then: the small directory starts over)
	pairLogReset(s);
}
static void pairLogAppend(s2amdSolver* s, unsigned long long key, int slot)
{
	if (!s->pairKeysValid)
	{
		return; // (a sort is due anyway)
	}
	if (s->hostPairLog == nullptr || s->hostPairLog[0] >= (unsigned long long)S2_PAIR_LOG_ENTRIES)
	{
		pairKeysStale(s);
		return;
	}
	// one entry per key, ascending (the kernels search it): a key that is there -- the pair was created, destroyed and created again --
	// takes the new slot
	unsigned long long* e = s->hostPairLog + 1;
	int* slots = (int*)(s->hostPairLog + S2_PAIR_LOG_ENTRIES + 1);
	const int n = (int)s->hostPairLog[0];
	int at = 0;
	while (at < n && e[at] < key)
	{
		at += 1;
	}
	if (!(at < n && e[at] == key))
	{
		memmove(e + at + 1, e + at, (size_t)(n - at) * sizeof(unsigned long long));
		memmove(slots + at + 1, slots + at, (size_t)(n - at) * sizeof(int));
		s->hostPairLog[0] = (unsigned long long)(n + 1);
	}
	e[at] = key;
	slots[at] = slot;
	s->pairLogDirty = true;
}
static int pairLogFlush(s2amdSolver* s)
{
	if (s->pairLogDirty && s->hostPairLog && s->dPairLog.p)
	{
		HIP_TRY(hipMemcpyAsync(s->dPairLog.p, s->hostPairLog, (size_t)(S2_PAIR_LOG_ENTRIES + 1) * 12, hipMemcpyHostToDevice, s->stream));
		s->pairLogDirty = false;
	}
	return S2AMD_OK;
}

int s2amd_world_upload(s2amdSolver* s, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
					   const s2amdJoint* joints, int32_t jointCapacity, const s2amdShape* shapes, int32_t shapeCapacity, const s2amdPairState* pairs,
					   const float* origins)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	if (shapeCapacity < 0 || (shapeCapacity > 0 && !shapes) || (contactCapacity > 0 && !pairs) || (bodyCapacity > 0 && !origins))
	{
		return fail(S2AMD_E_INVALID, "null array with non-zero count");
	}
	s->stepBackValid = false;
	s->refitOrderCount = 0; // (the caller's order belongs to the world it was sent for)
	treesForget(s);			// (... and so do its trees)
	for (int i = 0; i < contactCapacity; ++i)
	{
		if (pairs[i].shapeA >= shapeCapacity || pairs[i].shapeB >= shapeCapacity)
		{
			return fail(S2AMD_E_INVALID, "pair " + std::to_string(i) + " names a shape outside the shape array");
		}
	}
	for (int i = 0; i < shapeCapacity; ++i)
	{
		if (shapes[i].type != S2AMD_SHAPE_FREE && (shapes[i].body < 0 || shapes[i].body >= bodyCapacity))
		{
			return fail(S2AMD_E_INVALID, "shape " + std::to_string(i) + " names a body outside the body array");
		}
	}
	s->worldResident = false;
	s->pairKeysValid = false;
	s->pairQueryUsed = false, s->pairCacheValid = false;
	s->gatherIndexDirty = true;
	int rc = doUpload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity, pairs);
	if (rc)
	{
		return rc;
	}
	if (!s->hostWorldSummary)
	{
		HIP_TRY(hipHostMalloc((void**)&s->hostWorldSummary, 64 * sizeof(int), hipHostMallocDefault));
		memset(s->hostWorldSummary, 0, 64 * sizeof(int));
	}
	if (!s->hostPairLog)
	{
		HIP_TRY(hipHostMalloc((void**)&s->hostPairLog, (size_t)(S2_PAIR_LOG_ENTRIES + 1) * 12, hipHostMallocDefault));
	}
	s->hostPairLog[0] = 0ull;
	s->pairLogDirty = true;
	const size_t sBytes = (size_t)shapeCapacity * sizeof(s2amdShape), pBytes = (size_t)contactCapacity * sizeof(s2amdPairState);
	const size_t oBytes = (size_t)bodyCapacity * 2 * sizeof(float);
	if ((rc = s->dShapes.ensure(std::max<size_t>(sBytes, 256))) != 0 || (rc = s->dPairs.ensure(std::max<size_t>(pBytes, 256))) != 0 ||
		(rc = s->dOrigins.ensure(std::max<size_t>(oBytes, 256))) != 0 || (rc = s->dStatus.ensure(std::max<size_t>((size_t)contactCapacity * 4, 256))) != 0 ||
		(rc = s->dPointBytes.ensure(std::max<size_t>((size_t)contactCapacity, 256))) != 0 || (rc = s->dWorldSummary.ensure(256)) != 0 ||
		(rc = s->dSeparated.ensure(std::max<size_t>((size_t)contactCapacity * 4, 256))) != 0 ||
		(rc = s->dPairLog.ensure((size_t)(S2_PAIR_LOG_ENTRIES + 1) * 12)) != 0)
	{
		return rc;
	}
	// bodies connected by a joint never collide (src/broad_phase.c:283-298 via s2ShouldShapesCollide's joint walk): sorted body-pair keys
	{
		std::vector<unsigned long long> jointed;
		for (int j = 0; j < jointCapacity; ++j)
		{
			if (joints[j].type != S2AMD_JOINT_FREE && joints[j].bodyA >= 0 && joints[j].bodyB >= 0)
			{
				unsigned int a = (unsigned int)joints[j].bodyA, b = (unsigned int)joints[j].bodyB;
				jointed.push_back(((unsigned long long)std::min(a, b) << 32) | std::max(a, b));
			}
		}
		std::sort(jointed.begin(), jointed.end());
		if ((rc = s->dJointedKeys.ensure(std::max<size_t>(jointed.size() * 8, 256))) != 0)
		{
			return rc;
		}
		if (!jointed.empty())
		{
			HIP_TRY(hipMemcpyAsync(s->dJointedKeys.p, jointed.data(), jointed.size() * 8, hipMemcpyHostToDevice, s->stream));
			HIP_TRY(hipStreamSynchronize(s->stream)); // `jointed` is a local
		}
		s->jointedCount = (int)jointed.size();
		s->liveShapes = 0;
		// the shapes stage 4 can re-inflate: live shapes of non-static bodies (src/world.c:259-297) -- what a refit order must cover
		s->hShapeMovable.assign((size_t)shapeCapacity, 0);
		s->movableShapes = 0;
		for (int i = 0; i < shapeCapacity; ++i)
		{
			s->liveShapes += shapes[i].type != S2AMD_SHAPE_FREE ? 1 : 0;
			if (shapes[i].type != S2AMD_SHAPE_FREE && bodies[shapes[i].body].type != S2AMD_BODY_FREE && bodies[shapes[i].body].type != S2AMD_BODY_STATIC)
			{
				s->hShapeMovable[(size_t)i] = 1;
				s->movableShapes += 1;
			}
		}
	}
	s->hPointBytes.assign((size_t)contactCapacity, 0);
	for (int i = 0; i < contactCapacity; ++i)
	{
		s->hPointBytes[(size_t)i] = (uint8_t)s->hContactPoints[(size_t)i];
	}
	if (sBytes)
	{
		HIP_TRY(hipMemcpyAsync(s->dShapes.p, shapes, sBytes, hipMemcpyHostToDevice, s->stream));
	}
	if (pBytes)
	{
		HIP_TRY(hipMemcpyAsync(s->dPairs.p, pairs, pBytes, hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipMemcpyAsync(s->dPointBytes.p, s->hPointBytes.data(), (size_t)contactCapacity, hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipMemsetAsync(s->dStatus.p, 0xff, (size_t)contactCapacity * 4, s->stream)); // S2AMD_PAIR_FREE
	}
	if (oBytes)
	{
		HIP_TRY(hipMemcpyAsync(s->dOrigins.p, origins, oBytes, hipMemcpyHostToDevice, s->stream));
	}
	HIP_TRY(hipMemsetAsync(s->dWorldSummary.p, 0, 256, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->shapeCapacity = shapeCapacity;
	s->worldResident = true;
	if (s->optAsyncBuild != 0)
	{
		// (a stream for the first worker-thread build, made now while nothing is stepping: creating one later stalls the step that asks;
		// and the staging block of syncDeadSlots, whose first hipMalloc cost the first request 3 ms)
		workerStreamGive(workerStreamTake());
		if (s->hostSlotStageBytes < (size_t)contactCapacity)
		{
			if (s->hostSlotStage)
			{
				(void)hipHostFree(s->hostSlotStage);
				s->hostSlotStage = nullptr, s->hostSlotStageBytes = 0;
			}
			const size_t want = std::max<size_t>((size_t)contactCapacity, 4096);
			HIP_TRY(hipHostMalloc((void**)&s->hostSlotStage, want, hipHostMallocDefault));
			s->hostSlotStageBytes = want;
		}
		// (and the patch list's buffers -- solver_incremental.cpp: incrementalFlush: the first placed contact used to make them, 0.8 ms)
		if (s->hostPatches == nullptr)
		{
			HIP_TRY(hipHostMalloc((void**)&s->hostPatches, 4096 * sizeof(uint4), hipHostMallocDefault));
			s->hostPatchCapacity = 4096;
		}
		{
			int rcPatch = s->dPatches.ensure(s->hostPatchCapacity * sizeof(uint4));
			if (rcPatch)
			{
				return rcPatch;
			}
		}
		int rcSlot = s->dSlotBytes.ensure(std::max<size_t>((size_t)contactCapacity, 256));
		if (rcSlot)
		{
			return rcSlot;
		}
		s->hSlotBytes.reserve((size_t)contactCapacity);
	}
	if (s->optAsyncBuild != 0 && contactCapacity > 0)
	{
		// (the first device-to-host copy of the per-slot bytes cost the step it fell into 7 ms -- the first flip, the first search
		// request --, every later one 0.1: made here once, with the values the upload has just written)
		int rcWarm = syncDeadSlots(s);
		if (rcWarm == S2AMD_OK)
		{
			rcWarm = fetchPointCounts(s);
		}
		if (rcWarm)
		{
			return rcWarm;
		}
	}
	if (s->optPrebuildSolver >= 0 && s->optPrebuildSolver < s2amd_solverTypeCount)
	{
		// the caller has said which solver will step this world (the drop-in knows it from s2WorldDef): its structure -- strips and all --
		// is built here, where the world arrives, instead of in the first two steps (2.3 + 5.1 ms at base 200)
		// (no waiting for the graph to settle: the patience stays at zero -- "as it stands", noteGraphChanged raises it again when strips
		// die young -- so that the first step finds this structure up to date)
		s->stripPatienceNow = 0;
		int rcBuild = buildStructure(s, s->optPrebuildSolver);
		if (rcBuild)
		{
			return rcBuild;
		}
		HIP_TRY(hipStreamSynchronize(s->stream));
		if ((rcBuild = asyncPrewarm(s, s->optPrebuildSolver)) != 0)
		{
			return rcBuild;
		}
		// ... and the stage-1 pair query (s2amd_world_find_pairs): its scratch, its sorted pair keys and its captured graph
		if ((rcBuild = worldWarmPairQuery(s)) != 0)
		{
			return rcBuild;
		}
	}
	return S2AMD_OK;
}

int s2amd_world_step(s2amdSolver* s, const s2amdStepParams* params, s2amdWorldStepInfo* info)
{
	if (!s || !params)
	{
		return fail(S2AMD_E_INVALID, "null argument");
	}
	if (!s->worldResident || !s->resident)
	{
		return fail(S2AMD_E_STATE, "s2amd_world_step called before s2amd_world_upload");
	}
	HIP_TRY(hipSetDevice(s->device));
	const int nc = s->contactCapacity, nb = s->bodyCapacity, ns = s->shapeCapacity;
	hipStream_t st = s->stream;
	s->stepBackValid = false;
	WorldSummary* dSum = (WorldSummary*)s->dWorldSummary.p;
	WorldSummary* hSum = (WorldSummary*)s->hostWorldSummary;
	WorldSummary firstSeen{};
	bool haveFirst = false;
	const double t0 = nowMs();

	// ---- stage 3 (update contacts), s2Solve_* and stage 4 (refit), enqueued back to back: ONE read-back per step ----
	// The structure the solve runs on covers every live pair slot whether its manifold has points or not, so nothing the
	// narrow phase finds this step -- manifolds gaining or losing their points, pairs separating -- has to reach the host
	// before the solve is enqueued.
	// When the persistent step kernel reports a dead hand-off its epilogue leaves the wire arrays untouched, so the refit
	// behind it saw the bodies of the previous step (same AABBs, nothing enlarged): the solve is repeated on the
	// multi-launch path, and so is the refit.
	s->slotBytesFresh = false;
	// stage 2 (src/world.c:130): the trees the last refit flagged are rebuilt -- after the pair query of this step's stage 1, which the
	// caller has run (or the last step enqueued behind its stage 4), before this step's refit flags them again
	launchTreeRebuild(s, st);
	{
		int rcWatched = uploadWatched(s);
		if (rcWatched)
		{
			return rcWatched;
		}
	}
	if (nc > 0)
	{
		// (the kernel also destroys separated pairs and accumulates the step's contact counters)
		launchUpdateContacts(st, (const s2amdBody*)s->dBodies.p, (const float*)s->dOrigins.p, (const s2amdShape*)s->dShapes.p,
							 (s2amdPairState*)s->dPairs.p, (s2amdContact*)s->dContacts.p, nc, (int32_t*)s->dStatus.p, (uint8_t*)s->dPointBytes.p,
							 (int*)dSum, (int*)s->dSeparated.p, s->watchedCount > 0 && !s->structureDirty ? (const uint8_t*)s->dWatched.p : nullptr);
	}
	s->pointsKnown = false; // the manifolds are the device's now
	s->pointCountsFresh = false;
	s->hSeparated.clear();
	if (s->watchedCount > 0 && !s->structureDirty)
	{
		// a world with hub bodies: a manifold on one of them that gained or lost its points changes the graph, and the
		// solve must not be enqueued on the old structure -- the one case that needs stage 3's counters before the solve
		int rcMid = fetchSummary(s, 0);
		if (rcMid)
		{
			return rcMid;
		}
		if (hSum->watchedFlips > 0)
		{
			// (a watched manifold gained or lost its points.  A build in flight that was asked for after the contact was created holds it as
			// an ordinary potential constraint; one from before only watches it and is refused at adoption -- solver_async.cpp: asyncAdopt.
			// So the build is dropped only when the flip cannot be placed and the structure is rebuilt here: noteGraphChanged.)
			// s2Solve_Jacobi has no colours to find: a watched manifold that gained its first points gets a position and its two
			// incidence-list entries like a created contact (one that lost its points stays where it is, a no-op); every other
			// solver's structure is rebuilt
			// ... and under the soft contact solvers a manifold between bodies of the strips takes a free position of a strip or
			// seam round (solver_internal.h: IncrementalStrips); every other flip rebuilds the structure
			bool handled = false;
			bool byGroups = false; // (a flipped manifold on a body an LDS group holds: SolverRest::groupPatienceNow)
			const bool jacobi = s->inc.valid && s->inc.ignoreColours && s->optIncremental != 0;
			const bool strips = s->inc.valid && s->stripInc.valid && s->optIncremental != 0;
			// ... and (r4) a manifold between bodies of the global part takes a free colour position or, a hub's, a free position of
			// the sequential tail (solver_internal.h: IncrementalGlobal::tailFree)
			const bool tail = s->inc.valid && !s->inc.ignoreColours && s->optIncremental != 0 && (s->optFlipColours != 0 || !s->inc.tailFree.empty()); // (r6: a free colour position will do: tailCanPlace)
			auto inGlobalPart = [&](int a, int b) { return tailCanPlace(s, a, b); };
			if (jacobi || strips || tail)
			{
				if ((rcMid = fetchPointCounts(s)) != 0)
				{
					return rcMid;
				}
				std::vector<ContactChange> flipped;
				bool placeable = true;
				for (int i = 0; i < nc; ++i)
				{
					if (s->hContactWatched[(size_t)i] && s->hContactPoints[(size_t)i] > 0 && s->hContactEdge[(size_t)i] && !s->hContactDead[(size_t)i] &&
						s->inc.positionOfSlot[(size_t)i] == -1)
					{
						flipped.push_back(ContactChange{i, s->hContactA[(size_t)i], s->hContactB[(size_t)i]});
						byGroups = byGroups || ownedByLdsGroup(s, s->hContactA[(size_t)i]) || ownedByLdsGroup(s, s->hContactB[(size_t)i]);
						// (r5: ... or, where it fits no strip round, an overflow position behind the strips: the steps run sliced until a
						// worker thread's structure is adopted -- solver_internal.h: IncrementalStrips)
						placeable = placeable && (jacobi || inGlobalPart(s->hContactA[(size_t)i], s->hContactB[(size_t)i]) ||
												  (strips && (stripCanPlace(s, s->hContactA[(size_t)i], s->hContactB[(size_t)i]) ||
															  overflowCanPlace(s, s->hContactA[(size_t)i], s->hContactB[(size_t)i]))));
					}
				}
				handled = placeable && (flipped.empty() || incrementalApply(s, flipped));
				if (handled && !flipped.empty())
				{
					if ((rcMid = incrementalFlush(s)) != 0)
					{
						return rcMid;
					}
					if (s->inc.placedInGlobalPart)
					{
						noteGraphTouched(s);
					}
					s->gatherIndexDirty = true;
				}
			}
			if (!handled)
			{
				if (!(s->inc.ignoreColours && s->optIncremental != 0))
				{
					s->dirtyReason = "watched manifold flipped";
					s->dirtyByWatched = true;
				}
				s->dirtyByGroups = s->dirtyByGroups || byGroups;
				static const bool debugPrep = getenv("S2AMD_DEBUG_PREP") != nullptr;
				if (debugPrep)
				{
					fprintf(stderr, "[s2amd] watched flips %d not placed: jacobi %d strips %d tail %d (free %zu, bodies %d of %d, tail groups %d), watched %d, global %d of %d positions\n",
							hSum->watchedFlips, jacobi ? 1 : 0, strips ? 1 : 0, tail ? 1 : 0, s->inc.tailFree.size(), s->inc.tailBodyCount, s->inc.tailBodyCapacity,
							s->dContactTail.view.groupCount, s->watchedCount, s->contacts.globalCount, (int)s->contacts.order.size());
				}
				noteGraphChanged(s);
			}
		}
	}
	const double t1 = nowMs();
	static const bool debugAsync = getenv("S2AMD_DEBUG_ASYNC") != nullptr;
	double tStep = 0.0, tSummary = 0.0;
	int rc = S2AMD_OK;
	int fallbacks = 0, nearRetries = 0;
	for (;;)
	{
		const int savedAsync = s->optAsync;
		s->optAsync = 1;
		const double td0 = debugAsync ? nowMs() : 0.0;
		// (stage 4 rides in the solve's epilogue launch where that launch writes the bodies back -- contact_kernels.hip: storeImpulsesKernel --;
		// a step without such a launch (a world of self-contained resident islands; a store that finalizes positions itself) gets the
		// launch of its own; a captured step graph knows which of the two it holds: solver_step.cpp)
		s->stage4 = Stage4Args{(s2amdShape*)s->dShapes.p, ns, (float2*)s->dOrigins.p, (int*)dSum};
		s->stage4Carried = false;
		rc = doStep(s, params);
		s->stage4 = Stage4Args{};
		tStep += debugAsync ? nowMs() - td0 : 0.0;
		s->optAsync = savedAsync;
		if (rc)
		{
			return rc;
		}
		if (!s->stage4Carried)
		{
			launchStage4(st, (s2amdBody*)s->dBodies.p, nb, (s2amdShape*)s->dShapes.p, ns, (float*)s->dOrigins.p, (int*)dSum,
						 s->persistValid ? s->persist.deviceError : nullptr);
		}
		// (... and the proxies of the shapes it re-inflated enlarge the device's trees, src/world.c:283-290 -- once the rebuild running
		// beside this step is through: behind the pair query's own kernels when one rides along, which do not read the trees)
		const bool pairsRide = s->optPairsInStep != 0 && s->pairQueryUsed && fallbacks == 0 && nearRetries == 0 && s->liveShapes >= 2;
		const bool stepBack = s->optStepReadback != 0 && s->refitOrderCount > 0 && nb > 0 && ns > 0;
		s->stepBackListFresh = false;
		if (stepBack && (rc = enqueueStepBack(s)) != 0)
		{
			return rc;
		}
		if (!(pairsRide && treesActive(s)))
		{
			launchTreeEnlarge(s, st, s->persistValid ? s->persist.deviceError : nullptr);
		}
		// Stage 1 of the NEXT step behind stage 4 of this one: the pair query reads what the refit just wrote (the re-inflated fat boxes,
		// the move flags) and its results come back with this step's counters -- s2amd_world_find_pairs then costs no device round trip.
		// Pairs this step's stage 3 destroyed are still in the sorted directory of the pair set: the kernels check what a directory entry's
		// slot holds now (broadphase.hip: PairSetView).  Not in a step that is being repeated.
		s->pairCacheValid = false;
		if (pairsRide)
		{
			int32_t none = 0;
			const PairQueryHook enlarge{[](void* arg, hipStream_t hst) {
											s2amdSolver* hs = (s2amdSolver*)arg;
											launchTreeEnlarge(hs, hst, hs->persistValid ? hs->persist.deviceError : nullptr);
										},
										s};
			if ((rc = s->dPairKeys.ensure(std::max<size_t>((size_t)s->contactCapacity * 12, 256))) != 0 || (rc = pairLogFlush(s)) != 0 ||
				(rc = findPairsResident(st, (const s2amdShape*)s->dShapes.p, s->shapeCapacity, s->liveShapes, (const s2amdPairState*)s->dPairs.p, s->contactCapacity,
										(const unsigned long long*)s->dJointedKeys.p, s->jointedCount, nullptr, 0, &none, &s->dPairScratch.p, &s->dPairScratch.bytes,
										(unsigned long long*)s->dPairKeys.p, &s->pairKeysValid, &s->pairQuery, S2_PAIRS_ENQUEUE,
										(const unsigned long long*)s->dPairLog.p, (const int*)((const unsigned long long*)s->dPairLog.p + S2_PAIR_LOG_ENTRIES + 1), treesViews(s),
										&enlarge)) != 0)
			{
				return rc;
			}
			s->pairCacheValid = true;
		}
		const double ts0 = debugAsync ? nowMs() : 0.0;
		if ((rc = fetchSummary(s, 1)) != 0)
		{
			return rc;
		}
		tSummary += debugAsync ? nowMs() - ts0 : 0.0;
		s->stepBackValid = stepBack;
		if (s->hostError && *s->hostError != 0u && s->nearHandoffNow != 0 && nearRetries == 0)
		{
			// (the same-XCD hand-off path was in use: once more on the same kernel, with agent-scope stores everywhere -- solver_step.cpp)
			if ((rc = resetPersistState(s, st)) != 0)
			{
				return rc;
			}
			s->nearHandoffNow = 0;
			s->nearHandoffTimeouts += 1;
			s->layoutGeneration += 1;
			nearRetries += 1;
			if (!haveFirst)
			{
				firstSeen = *hSum;
				haveFirst = true;
			}
			continue;
		}
		if (s->hostError && *s->hostError != 0u && fallbacks == 0)
		{
			// bodies, impulses and (because the solve left the bodies alone) the shapes are what they were before the solve;
			// the counters of stage 3 were read and reset above and are kept (contactsSeen below takes the first read)
			if ((rc = resetPersistState(s, st)) != 0)
			{
				return rc;
			}
			if (s->overflowKernelThisStep && !s->overflowKernelFailed)
			{
				s->overflowKernelFailed = true; // (the launch carried the overflow workgroup: the step again, sliced)
				s->layoutGeneration += 1;
			}
			else
			{
				s->persistFailed = true;
			}
			s->persistFallbacks += 1;
			fallbacks += 1;
			if (!haveFirst)
			{
				firstSeen = *hSum;
				haveFirst = true;
			}
			continue;
		}
		break;
	}
	if (debugAsync && nowMs() - t0 > 1.0)
	{
		fprintf(stderr, "[s2amd] step %ld took %.3f ms on the host: stage 3 + flips %.3f, doStep (enqueue) %.3f, wait for the device %.3f\n", s->stepCounter, nowMs() - t0, t1 - t0,
				tStep, tSummary);
	}
	if (!s->poolWarmed && asyncBuildsOn(s) && (rc = asyncPrewarm(s, params->solverType)) != 0)
	{
		return rc; // (a world uploaded without "prebuild_solver": stocked behind its first step)
	}
	WorldSummary contactsSeen = haveFirst ? firstSeen : *hSum;
	if (contactsSeen.separated > 0)
	{
		// (pair slots were freed on the device: the pair query sees that in the slots themselves -- broadphase.hip: PairSetView)
		// which ones: the caller needs them for its own s2DestroyContact (s2amd_world_separated), and their entries leave the
		// structure now where they can (solver_incremental.cpp), as a host that ran stage 3 itself would see them gone from
		// the arrays of its next upload
		s->hSeparated.resize((size_t)contactsSeen.separated);
		HIP_TRY(hipMemcpyAsync(s->hSeparated.data(), s->dSeparated.p, s->hSeparated.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		std::sort(s->hSeparated.begin(), s->hSeparated.end());
		for (int32_t slot : s->hSeparated)
		{
			s->hContactDead[(size_t)slot] = 1;
			unwatchSlot(s, slot);
			asyncLogDestroyed(s, slot);
		}
		incrementalRemove(s, s->hSeparated.data(), (int)s->hSeparated.size());
		if ((rc = incrementalFlush(s)) != 0)
		{
			return rc;
		}
	}
	s->activeContacts = contactsSeen.active;
	s->stats.constraintCount = contactsSeen.active;
	{
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, s->evBegin, s->evEnd) == hipSuccess)
		{
			s->stats.deviceMs = ms;
		}
		s->stats.persistFallbacks = s->persistFallbacks;
	}
	s->lastMovedCount = hSum->enlarged;
	if (info)
	{
		info->separatedCount = contactsSeen.separated;
		info->activeContacts = contactsSeen.active;
		info->graphChanged = contactsSeen.flips > 0 ? 1 : 0;
		info->movedCount = hSum->enlarged;
		info->contactsMs = (float)(t1 - t0);
		info->solveMs = s->stats.deviceMs;
		info->stepMs = (float)(nowMs() - t0);
	}
	return S2AMD_OK;
}

int s2amd_world_find_pairs(s2amdSolver* s, int32_t* outPairs, int32_t pairCapacity, int32_t* pairCount)
{
	if (!s || !pairCount || pairCapacity < 0 || (pairCapacity > 0 && !outPairs))
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (!s->worldResident || !s->resident)
	{
		return fail(S2AMD_E_STATE, "no resident world");
	}
	HIP_TRY(hipSetDevice(s->device));
	{
		int rc = s->dPairKeys.ensure(std::max<size_t>((size_t)s->contactCapacity * 12, 256));
		if (rc)
		{
			return rc;
		}
	}
	// (the last step has run this query behind its stage 4 -- s2amd_world_step -- unless this is the first call, a contact was set since,
	// or the step was repeated: then it runs now)
	const bool collect = s->pairCacheValid;
	s->pairQueryUsed = true;
	int rc = collect ? S2AMD_OK : pairLogFlush(s);
	if (rc)
	{
		return rc;
	}
	rc = findPairsResident(s->stream, (const s2amdShape*)s->dShapes.p, s->shapeCapacity, s->liveShapes, (const s2amdPairState*)s->dPairs.p, s->contactCapacity,
							   (const unsigned long long*)s->dJointedKeys.p, s->jointedCount, outPairs, pairCapacity, pairCount, &s->dPairScratch.p,
							   &s->dPairScratch.bytes, (unsigned long long*)s->dPairKeys.p, &s->pairKeysValid, &s->pairQuery, collect ? S2_PAIRS_COLLECT : S2_PAIRS_FULL,
							   (const unsigned long long*)s->dPairLog.p, (const int*)((const unsigned long long*)s->dPairLog.p + S2_PAIR_LOG_ENTRIES + 1), treesViews(s));
	if (rc == S2AMD_OK)
	{
		s->pairCacheValid = false; // (collected once: the move flags go below, as after a query of this call's own)
	}
	if (rc == S2AMD_OK && s->shapeCapacity > 0) // (S2AMD_E_CAPACITY: the caller asks again with a larger buffer)
	{
		clearMovedKernel<<<gridFor((size_t)s->shapeCapacity), dim3(S2_BLOCK), 0, s->stream>>>((s2amdShape*)s->dShapes.p, s->shapeCapacity);
		HIP_TRY(hipGetLastError());
	}
	return rc;
}

int s2amd_world_download_boxes(s2amdSolver* s, s2amdShapeBox* boxes, int32_t shapeCapacity)
{
	if (!s || !boxes || shapeCapacity < 0)
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (!s->worldResident || !s->resident)
	{
		return fail(S2AMD_E_STATE, "no resident world");
	}
	if (shapeCapacity < s->shapeCapacity)
	{
		return fail(S2AMD_E_CAPACITY, "box array smaller than the resident shape array");
	}
	if (s->shapeCapacity == 0)
	{
		return S2AMD_OK;
	}
	HIP_TRY(hipSetDevice(s->device));
	int rc = s->dShapeBoxes.ensure((size_t)s->shapeCapacity * sizeof(s2amdShapeBox));
	if (rc)
	{
		return rc;
	}
	shapeBoxesKernel<<<gridFor((size_t)s->shapeCapacity), dim3(S2_BLOCK), 0, s->stream>>>((const s2amdShape*)s->dShapes.p, s->shapeCapacity,
																						  (s2amdShapeBox*)s->dShapeBoxes.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(boxes, s->dShapeBoxes.p, (size_t)s->shapeCapacity * sizeof(s2amdShapeBox), hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

int s2amd_world_set_refit_order(s2amdSolver* s, const int32_t* shapeOrder, int32_t count)
{
	if (!s || count < 0 || (count > 0 && !shapeOrder))
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	HIP_TRY(hipSetDevice(s->device));
	s->refitOrderCount = 0;
	s->stepBackValid = false;
	if (count == 0)
	{
		return treesSyncRefitOrder(s);
	}
	// The order must name every shape stage 4 can re-inflate exactly once: s2amd_world_download_step trusts it (a stale order would
	// hand the caller a step's moved boxes with some of them missing).  Shapes only change with s2amd_world_upload, which drops the order.
	if (!s->worldResident || (int)s->hShapeMovable.size() != s->shapeCapacity)
	{
		return fail(S2AMD_E_STATE, "s2amd_world_set_refit_order called before s2amd_world_upload");
	}
	{
		std::vector<uint8_t> seen((size_t)s->shapeCapacity, 0);
		for (int i = 0; i < count; ++i)
		{
			const int k = shapeOrder[i];
			if (k < 0 || k >= s->shapeCapacity || seen[(size_t)k] || !s->hShapeMovable[(size_t)k])
			{
				return fail(S2AMD_E_INVALID, "refit order entry " + std::to_string(i) + " is not a live shape of a movable body, or is named twice");
			}
			seen[(size_t)k] = 1;
		}
		if (count != s->movableShapes)
		{
			return fail(S2AMD_E_INVALID, "refit order names " + std::to_string(count) + " of the world's " + std::to_string(s->movableShapes) + " movable shapes");
		}
	}
	int rc = s->dRefitOrder.ensure((size_t)count * sizeof(int32_t));
	if (rc)
	{
		return rc;
	}
	HIP_TRY(hipMemcpyAsync(s->dRefitOrder.p, shapeOrder, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->refitOrderCount = count;
	return treesSyncRefitOrder(s);
}

int s2amd_world_download_step(s2amdSolver* s, float* poses, int32_t bodyCapacity, s2amdMovedBox* moved, int32_t movedCapacity, int32_t* movedCount)
{
	if (!s || !movedCount || bodyCapacity < 0 || movedCapacity < 0 || (movedCapacity > 0 && !moved))
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (!s->worldResident || !s->resident)
	{
		return fail(S2AMD_E_STATE, "no resident world");
	}
	if (poses && bodyCapacity < s->bodyCapacity)
	{
		return fail(S2AMD_E_CAPACITY, "pose array smaller than the resident body array");
	}
	*movedCount = s->lastMovedCount;
	if (s->lastMovedCount > movedCapacity)
	{
		return fail(S2AMD_E_CAPACITY, "moved-box array too small");
	}
	if (s->stepBackValid && s->lastMovedCount <= s->stepBackBoxes && ((const int32_t*)s->hostStepBack)[0] == s->lastMovedCount)
	{
		// the step brought them along (enqueueStepBack)
		if (s->lastMovedCount > 0)
		{
			memcpy(moved, s->hostStepBack + 16, (size_t)s->lastMovedCount * sizeof(s2amdMovedBox));
		}
		if (poses && s->bodyCapacity > 0)
		{
			memcpy(poses, s->hostStepBack + s->stepBackPoseOffset, (size_t)s->bodyCapacity * sizeof(float4));
		}
		return S2AMD_OK;
	}
	HIP_TRY(hipSetDevice(s->device));
	const int nb = s->bodyCapacity, want = s->lastMovedCount;
	const size_t headBytes = 16 + (size_t)std::max(want, 1) * sizeof(s2amdMovedBox);
	const size_t poseOffset = (headBytes + 255) & ~size_t(255);
	int rc = s->dStepBack.ensure(poseOffset + (size_t)std::max(nb, 1) * sizeof(float4));
	if (rc)
	{
		return rc;
	}
	char* base = (char*)s->dStepBack.p;
	if (want > 0)
	{
		const bool ordered = s->refitOrderCount > 0;
		const int n = ordered ? s->refitOrderCount : s->shapeCapacity;
		const int tiles = (n + S2_BLOCK - 1) / S2_BLOCK;
		const int* order = ordered ? (const int*)s->dRefitOrder.p : nullptr;
		if ((rc = s->dScanTmp.ensure((size_t)std::max(tiles, 1) * sizeof(int))) != 0)
		{
			return rc;
		}
		movedCountKernel<<<dim3((unsigned)tiles), dim3(S2_BLOCK), 0, s->stream>>>((const s2amdShape*)s->dShapes.p, s->shapeCapacity, order, n, (int*)s->dScanTmp.p);
		movedWriteKernel<<<dim3((unsigned)tiles), dim3(S2_BLOCK), 0, s->stream>>>((const s2amdShape*)s->dShapes.p, s->shapeCapacity, order, n, (const int*)s->dScanTmp.p,
																				   (int32_t*)base, want, tiles, nullptr, nullptr, 0, nullptr);
		HIP_TRY(hipGetLastError());
	}
	if (poses && nb > 0)
	{
		stepPosesKernel<<<gridFor((size_t)nb), dim3(S2_BLOCK), 0, s->stream>>>((const s2amdBody*)s->dBodies.p, (const float*)s->dOrigins.p, nb, (float4*)(base + poseOffset));
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(poses, base + poseOffset, (size_t)nb * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
	}
	int32_t head[4] = {0, 0, 0, 0};
	if (want > 0)
	{
		HIP_TRY(hipMemcpyAsync(head, base, sizeof(head), hipMemcpyDeviceToHost, s->stream));
		HIP_TRY(hipMemcpyAsync(moved, base + 16, (size_t)want * sizeof(s2amdMovedBox), hipMemcpyDeviceToHost, s->stream));
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	if (want > 0 && head[0] != want)
	{
		// (the refit order does not cover every shape that moved: the caller's order is stale)
		return fail(S2AMD_E_STATE, "refit order misses " + std::to_string(want - head[0]) + " re-inflated shapes");
	}
	return S2AMD_OK;
}

int s2amd_world_separated(s2amdSolver* s, int32_t* slots, int32_t capacity, int32_t* count)
{
	if (!s || !count || capacity < 0 || (capacity > 0 && !slots))
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	*count = (int32_t)s->hSeparated.size();
	if (*count > capacity)
	{
		return fail(S2AMD_E_CAPACITY, "slot buffer too small");
	}
	std::copy(s->hSeparated.begin(), s->hSeparated.end(), slots);
	return S2AMD_OK;
}

int s2amd_world_set_contacts(s2amdSolver* s, const int32_t* slots, int32_t count, const s2amdContact* contacts, const s2amdPairState* pairs)
{
	if (!s || count < 0 || (count > 0 && (!slots || !contacts || !pairs)))
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (!s->worldResident || !s->resident)
	{
		return fail(S2AMD_E_STATE, "no resident world");
	}
	if (count == 0)
	{
		return S2AMD_OK;
	}
	std::vector<uint8_t> seen((size_t)s->contactCapacity, 0);
	for (int i = 0; i < count; ++i)
	{
		const int k = slots[i];
		if (k < 0 || k >= s->contactCapacity || seen[(size_t)k])
		{
			return fail(S2AMD_E_INVALID, "contact slot " + std::to_string(k) + " is outside the resident contact array or named twice");
		}
		seen[(size_t)k] = 1;
		const s2amdContact& c = contacts[i];
		const bool live = pairs[i].shapeA >= 0;
		if (pairs[i].shapeA >= s->shapeCapacity || pairs[i].shapeB >= s->shapeCapacity || (live && pairs[i].shapeB < 0) ||
			(live && (c.bodyA < 0 || c.bodyA >= s->bodyCapacity || c.bodyB < 0 || c.bodyB >= s->bodyCapacity)) || c.pointCount > 2)
		{
			return fail(S2AMD_E_INVALID, "contact for slot " + std::to_string(k) + " names a shape or body outside the resident arrays");
		}
	}
	HIP_TRY(hipSetDevice(s->device));
	hipStream_t st = s->stream;
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	const size_t b0 = al((size_t)count * sizeof(int32_t)), b1 = al((size_t)count * sizeof(s2amdContact)), b2 = al((size_t)count * sizeof(s2amdPairState));
	int rc = s->dContactStage.ensure(b0 + b1 + b2);
	if (rc)
	{
		return rc;
	}
	char* base = (char*)s->dContactStage.p;
	HIP_TRY(hipMemcpyAsync(base, slots, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(base + b0, contacts, (size_t)count * sizeof(s2amdContact), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(base + b0 + b1, pairs, (size_t)count * sizeof(s2amdPairState), hipMemcpyHostToDevice, st));
	scatterContactsKernel<<<gridFor((size_t)count), dim3(S2_BLOCK), 0, st>>>((const int32_t*)base, count, (const s2amdContact*)(base + b0),
																			  (const s2amdPairState*)(base + b0 + b1), (s2amdContact*)s->dContacts.p,
																			  (s2amdPairState*)s->dPairs.p, (uint8_t*)s->dPointBytes.p, (int32_t*)s->dStatus.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	// the pair set changed: a contact created in a slot that held no live pair goes into the pair log; anything else (a pair taken
	// away or replaced by the caller: its old key is the device's to know) has the key set sorted again
	for (int i = 0; i < count; ++i)
	{
		const int k = slots[i];
		const bool wasLive = k < (int)s->hContactEdge.size() && s->hContactEdge[(size_t)k] && !s->hContactDead[(size_t)k];
		if (pairs[i].shapeA >= 0 && pairs[i].shapeB >= 0 && !wasLive)
		{
			const unsigned long long a = (unsigned long long)std::min(pairs[i].shapeA, pairs[i].shapeB), b = (unsigned long long)std::max(pairs[i].shapeA, pairs[i].shapeB);
			pairLogAppend(s, (a << 32) | b, k);
		}
		else
		{
			pairKeysStale(s);
		}
	}
	s->pairCacheValid = false; // (a query enqueued behind the last step did not know this contact)
	// host shadows of the constraint graph (solver_step.cpp: refreshShadows): a slot is a potential constraint while its
	// pair is live, whatever its manifold holds.  Created contacts get a place in the existing structure when they fit
	// (solver_incremental.cpp), in pool order like the host-array route; else the structure is rebuilt.
	std::vector<int> byslot((size_t)count);
	for (int i = 0; i < count; ++i)
	{
		byslot[(size_t)i] = i;
	}
	std::sort(byslot.begin(), byslot.end(), [&](int x, int y) { return slots[x] < slots[y]; });
	std::vector<ContactChange> created;
	bool hubTouched = false; // a contact on a hub body came or went: which of its contacts are structural is decided by a rebuild
	auto onHub = [&](int a, int b) {
		return !s->hBodyHub.empty() && ((a >= 0 && a < (int)s->hBodyHub.size() && s->hBodyHub[(size_t)a]) || (b >= 0 && b < (int)s->hBodyHub.size() && s->hBodyHub[(size_t)b]));
	};
	for (int i : byslot)
	{
		const int k = slots[i];
		const s2amdContact& c = contacts[i];
		const int pc = c.pointCount > 0 ? c.pointCount : 0;
		const bool edge = pairs[i].shapeA >= 0 || pc > 0;
		if (s->pointsKnown)
		{
			s->activeContacts += (pc > 0 ? 1 : 0) - (s->hContactPoints[(size_t)k] > 0 ? 1 : 0);
			s->hContactPoints[(size_t)k] = pc;
			s->hPointBytes[(size_t)k] = (uint8_t)pc;
		}
		if (!edge)
		{
			if (s->hContactEdge[(size_t)k])
			{
				s->hContactDead[(size_t)k] = 1; // the caller's s2DestroyContact: the entry leaves the structure where it can, else lingers as a no-op
				unwatchSlot(s, k);
				const int32_t one = k;
				incrementalRemove(s, &one, 1);
				asyncLogDestroyed(s, k);
			}
			continue;
		}
		if (!s->hContactEdge[(size_t)k] || s->hContactA[(size_t)k] != c.bodyA || s->hContactB[(size_t)k] != c.bodyB)
		{
			if (pc > 0)
			{
				asyncDrop(s); // (created with points already: not what a build in flight replays)
			}
			asyncLogCreated(s, k, c.bodyA, c.bodyB);
			if (pc == 0 && canDeferCreated(s, k, c.bodyA, c.bodyB))
			{
				deferCreated(s, k, c.bodyA, c.bodyB); // watched until stage 3 finds its first manifold points
				continue;
			}
			created.push_back(ContactChange{k, c.bodyA, c.bodyB});
			hubTouched = hubTouched || onHub(c.bodyA, c.bodyB);
			continue;
		}
		s->hContactDead[(size_t)k] = 0; // the same pair again in its old slot
	}
	if (!created.empty())
	{
		const bool placed = !hubTouched && incrementalApply(s, created);
		for (const ContactChange& ch : created)
		{
			s->hContactA[(size_t)ch.slot] = ch.a;
			s->hContactB[(size_t)ch.slot] = ch.b;
			s->hContactEdge[(size_t)ch.slot] = 1;
			s->hContactDead[(size_t)ch.slot] = 0;
		}
		if (placed)
		{
			int rcFlush = incrementalFlush(s);
			if (rcFlush)
			{
				return rcFlush;
			}
			if (s->inc.placedInGlobalPart)
			{
				noteGraphTouched(s);
			}
		}
		else
		{
			noteGraphChanged(s);
		}
	}
	else
	{
		int rcFlush = incrementalFlush(s); // (removals)
		if (rcFlush)
		{
			return rcFlush;
		}
		if (s->persistValid)
		{
			s->persist.allTwoPoints = stripsAllTwoPoints(s) ? 1 : 0;
		}
		s->residentAllTwoPoints = residentAllTwoPoints(s) ? 1 : 0;
	}
	s->gatherIndexDirty = true;
	return S2AMD_OK;
}

int s2amd_world_download(s2amdSolver* s, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts, int32_t contactCapacity, s2amdJoint* joints,
						 int32_t jointCapacity, s2amdShape* shapes, int32_t shapeCapacity, s2amdPairState* pairs, float* origins, int32_t* status)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	if (!s->worldResident || !s->resident)
	{
		return fail(S2AMD_E_STATE, "no resident world");
	}
	if ((bodies && bodyCapacity < s->bodyCapacity) || (origins && bodyCapacity < s->bodyCapacity) || (contacts && contactCapacity < s->contactCapacity) ||
		(pairs && contactCapacity < s->contactCapacity) || (status && contactCapacity < s->contactCapacity) || (joints && jointCapacity < s->jointCapacity) ||
		(shapes && shapeCapacity < s->shapeCapacity))
	{
		return fail(S2AMD_E_CAPACITY, "output arrays smaller than the resident world");
	}
	HIP_TRY(hipSetDevice(s->device));
	hipStream_t st = s->stream;
	if (contacts && !s->pointsKnown && s->lastStepWroteIndex)
	{
		// manifold.constraintIndex is an output nobody on the device reads: written when somebody asks for the contacts
		int rc = refreshConstraintIndexOnDevice(s);
		if (rc)
		{
			return rc;
		}
	}
	if (bodies && s->bodyCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(bodies, s->dBodies.p, (size_t)s->bodyCapacity * sizeof(s2amdBody), hipMemcpyDeviceToHost, st));
	}
	if (origins && s->bodyCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(origins, s->dOrigins.p, (size_t)s->bodyCapacity * 2 * sizeof(float), hipMemcpyDeviceToHost, st));
	}
	if (contacts && s->contactCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(contacts, s->dContacts.p, (size_t)s->contactCapacity * sizeof(s2amdContact), hipMemcpyDeviceToHost, st));
	}
	if (pairs && s->contactCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(pairs, s->dPairs.p, (size_t)s->contactCapacity * sizeof(s2amdPairState), hipMemcpyDeviceToHost, st));
	}
	if (status && s->contactCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(status, s->dStatus.p, (size_t)s->contactCapacity * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	}
	if (joints && s->jointCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(joints, s->dJoints.p, (size_t)s->jointCapacity * sizeof(s2amdJoint), hipMemcpyDeviceToHost, st));
	}
	if (shapes && s->shapeCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(shapes, s->dShapes.p, (size_t)s->shapeCapacity * sizeof(s2amdShape), hipMemcpyDeviceToHost, st));
	}
	HIP_TRY(hipStreamSynchronize(st));
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop

S2_DEFINE_WARM(world)
