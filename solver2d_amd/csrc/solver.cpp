// Host side of the C-ABI (include/solver2d_amd.h): device memory, graph colouring, the ten solver
// drivers as kernel-launch sequences, hipGraph capture/replay, timing.
//
// One s2amdSolver owns one HIP stream and all device state of one world.  Each driver below
// enqueues exactly the stage sequence of the reference driver it names; a "sweep" over contacts
// or joints is one launch per colour batch.  There is NO CPU fallback: without a gfx950 device
// s2amd_create fails with S2AMD_E_NODEVICE.

#include <atomic>
#include "solver_executor.h"

namespace
{

thread_local std::string g_lastError;

} // namespace

// shared with the other translation units (solver_internal.h: fail) and the .hip stage files
int s2amdFail(int code, const std::string& msg)
{
	g_lastError = msg;
	return code;
}

hipStream_t s2amdStream(s2amdSolver* s)
{
	return s->stream;
}
// kernel time of the last stage call (narrowphase.hip, broadphase.hip report through s2amdStepStats.deviceMs)
void s2amdRecordDeviceMs(s2amdSolver* s, float ms)
{
	s->stats.deviceMs = ms;
}
const s2amdBody* s2amdResidentBodies(s2amdSolver* s) { return s != nullptr ? (const s2amdBody*)s->dBodies.p : nullptr; }
bool s2amdStepFailed(s2amdSolver* s) { return s != nullptr && s->hostError != nullptr && *s->hostError != 0u; }
int s2amdDevice(s2amdSolver* s)
{
	return s->device;
}

#pragma GCC visibility push(default)
extern "C"
{

int s2amd_api_version(void)
{
	return S2AMD_API_VERSION;
}

const char* s2amd_last_error(void)
{
	return g_lastError.c_str();
}

const char* s2amd_build_flags(void)
{
#if defined(S2AMD_FAST_BUILD) && S2AMD_FAST_BUILD
	return "fp-contract=fast";
#else
	return "fp-contract=off";
#endif
}

int s2amd_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
	{
		(void)hipGetLastError();
		return 0;
	}
	return n;
}

int s2amd_device_bus_id(int device, char* out, int32_t capacity)
{
	if (!out || capacity < 16)
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (device < 0 || device >= s2amd_device_count())
	{
		return fail(S2AMD_E_INVALID, "device ordinal out of range");
	}
	HIP_TRY(hipDeviceGetPCIBusId(out, capacity, device));
	return S2AMD_OK;
}

int s2amd_create(int device, s2amdSolver** out)
{
	if (!out)
	{
		return fail(S2AMD_E_INVALID, "null out pointer");
	}
	*out = nullptr;
	int n = s2amd_device_count();
	if (n <= 0)
	{
		return fail(S2AMD_E_NODEVICE, "no HIP device visible; this library has no CPU path");
	}
	if (device < 0 || device >= n)
	{
		return fail(S2AMD_E_INVALID, "device ordinal out of range");
	}
	HIP_TRY(hipSetDevice(device));
	s2amdSolver* s = new s2amdSolver();
	s->device = device;
	if (const char* v = getenv("S2AMD_PAIR_LANES")) // experiments: the defaults of options "pair_lanes" and "wide"
	{
		s->optPairLanes = atoi(v) != 0;
	}
	if (const char* v = getenv("S2AMD_GENERIC"))
	{
		s->optGeneric = atoi(v) != 0;
	}
	if (const char* v = getenv("S2AMD_WIDE"))
	{
		s->optWide = atoi(v) != 0;
	}
	hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
	if (e == hipSuccess)
	{
		e = hipEventCreate(&s->evBegin);
	}
	if (e == hipSuccess)
	{
		e = hipEventCreate(&s->evEnd);
	}
	for (int i = 0; i < 2 && e == hipSuccess; ++i)
	{
		e = hipStreamCreateWithFlags(&s->side[i], hipStreamNonBlocking);
		if (e == hipSuccess)
		{
			e = hipEventCreateWithFlags(&s->evFork[i], hipEventDisableTiming);
		}
	}
	for (int i = 0; i < 4 && e == hipSuccess; ++i)
	{
		e = hipEventCreateWithFlags(&s->evJoin[i], hipEventDisableTiming);
	}
	if (e != hipSuccess)
	{
		delete s;
		return fail(S2AMD_E_DEVICE, std::string("stream/event creation: ") + hipGetErrorString(e));
	}
	(void)hipDeviceGetAttribute(&s->cuCount, hipDeviceAttributeMultiprocessorCount, device);
	{
		// the hub rule's unit costs (solver_internal.h: HubCosts): one row per architecture this library has been measured on
		static const struct
		{
			const char* arch;
			HubCosts costs;
		} table[] = {{"gfx950", HubCosts{1.4f, 2.0f, 0.38f}}}; // MI355X: op-interpreter colour round, dependent launch in a graph, tail visit (profiles/r04_config3b_*)
		hipDeviceProp_t prop{};
		if (hipGetDeviceProperties(&prop, device) == hipSuccess)
		{
			for (const auto& row : table)
			{
				if (strncmp(prop.gcnArchName, row.arch, strlen(row.arch)) == 0)
				{
					s->hubCosts = row.costs;
				}
			}
		}
	}

	if (hipHostMalloc((void**)&s->hostError, sizeof(unsigned int), hipHostMallocMapped) == hipSuccess)
	{
		*s->hostError = 0u;
	}
	else
	{
		s->hostError = nullptr;
		(void)hipGetLastError();
	}
	if (groupKernelSetup() != 0 || stripKernelSetup() != 0 || pairKernelSetup() != 0 || wideKernelSetup() != 0 || genericKernelSetup() != 0 || jacobiKernelSetup() != 0)
	{
		(void)hipGetLastError(); // not fatal: groups are then limited to the default 64 KiB of LDS
		s->optMaxGroupBodies = 1536;
	}
	{
		// every code object of the library loaded now, not by whichever step first launches a kernel of it (launch.h: S2_DEFINE_WARM)
		// (two threads creating solvers at once -- one per device of a sharded solver -- must not both find the flag clear, nor race on it)
		static std::atomic<bool> warmed[64];
		if (device < 64 && !warmed[device].exchange(true))
		{
			s2Warm_contact_kernels(s->stream);
			s2Warm_body_kernels(s->stream);
			s2Warm_joint_kernels(s->stream);
			s2Warm_group_kernel(s->stream);
			s2Warm_strip_kernel(s->stream);
			s2Warm_pair_kernel(s->stream);
			s2Warm_wide_kernel(s->stream);
			s2WarmScratch(s->stream);
			s2Warm_generic_kernel(s->stream);
			s2Warm_broadphase(s->stream);
			s2Warm_narrowphase(s->stream);
			s2Warm_tree_mirror(s->stream);
			s2Warm_structure(s->stream);
			s2Warm_world(s->stream);
			s2Warm_sharded(s->stream);
			s2Warm_jacobi_kernel(s->stream);
			(void)hipStreamSynchronize(s->stream);
			(void)hipGetLastError();
		}
	}
	devPoolSolverCreated(device);
	*out = s;
	return S2AMD_OK;
}

void s2amd_destroy(s2amdSolver* s)
{
	if (!s)
	{
		return;
	}
	(void)hipSetDevice(s->device);
	asyncShutdown(s); // (a worker thread may still be building on a copy of this solver)
	if (devPoolSolverDestroyed(s->device))
	{
		devPoolDrain(); // (what the copies gave back on this device, once its last solver goes: solver_internal.h: DevBuf)
		if (devPoolNoSolverLeft())
		{
			spareClonesRelease(); // (... and the retired copies themselves -- megabytes of host vectors -- with the process's last solver)
		}
	}
	(void)hipStreamSynchronize(s->stream);
	destroyGraph(s);
	for (hipEvent_t e : s->sweepEvents)
	{
		(void)hipEventDestroy(e);
	}
	DevBuf* bufs[] = {&s->dBodies,		&s->dContacts,	  &s->dJoints,		 &s->dBodiesSaved,	 &s->dBodyFlags,	&s->soaBodies,
					  &s->soaContacts,	&s->soaJoints,	  &s->dContactIndex, &s->dJointIndex,	 &s->dContactLocal, &s->dJointLocal,
					  &s->dAdjOffsets,	&s->dAdjList,	  &s->dAdjHeavy,	  &s->dGatherIndex,	 &s->dOps,			 &s->dGroups.buf,	&s->dContactTail.buf,
					  &s->dJointTail.buf, &s->dMsg,			  &s->dStripA.buf,	 &s->dStripB.buf,	 &s->dStripLean,	&s->dPersist,
					  &s->dGranules,	&s->dOverflowBodies, &s->dPairLog, &s->dPersistOps, &s->dJacobi, &s->dJacobiGran,	  &s->dShapes,		 &s->dPairs,		 &s->dOrigins,		&s->dStatus,
					  &s->dPointBytes,	&s->dWorldSummary, &s->dJointedKeys,	 &s->dContactStage, &s->dPairScratch,	  &s->dPairKeys,		 &s->dPatches,		 &s->dScanTmp,		 &s->dResident.buf,	 &s->dResidentDesc, &s->dResidentOps, &s->dWatched, &s->dRefitOrder, &s->dStepBack,
					  &s->dSlotBytes,	  &s->dJointAdjRange, &s->dJointAdjList, &s->dShapeBoxes};
	for (DevBuf* b : bufs)
	{
		b->release();
	}
	treesFree(s);
	if (s->hostError)
	{
		(void)hipHostFree(s->hostError);
	}
	if (s->pairQuery.exec)
	{
		(void)hipGraphExecDestroy(s->pairQuery.exec);
	}
	if (s->pairQuery.host)
	{
		(void)hipHostFree(s->pairQuery.host);
	}
	if (s->hostStepBack)
	{
		(void)hipHostFree(s->hostStepBack);
	}
	if (s->hostPairLog)
	{
		(void)hipHostFree(s->hostPairLog);
		s->hostPairLog = nullptr;
	}
	if (s->hostSlotStage)
	{
		(void)hipHostFree(s->hostSlotStage);
		s->hostSlotStage = nullptr;
	}
	if (s->hostWorldSummary)
	{
		(void)hipHostFree(s->hostWorldSummary);
	}
	if (s->hostPatches)
	{
		(void)hipHostFree(s->hostPatches);
	}
	if (s->hostTimes)
	{
		int n = (int)s->hostTimes[255];
		if (s->hostTimes[254] == 1ull && n > 1)
		{
			// pair_kernel.hip's tagged stamps: time per phase of the last step, one workgroup
			static const char* names[16] = {"start", "load", "body stages", "warm starts", "interior rounds (last)", "hand-offs", "seam rounds (last)", "store",
											"interior 0", "interior 1", "interior 2", "interior 3", "commit wait (self-contained)", "interior 5", "seam 0", "seam 1 | warm-start terms (body-centric)"};
			double sum[16] = {0};
			int count[16] = {0};
			for (int i = 1; i < n && i < 250; ++i)
			{
				const unsigned tag = (unsigned)(s->hostTimes[i] & 15ull);
				sum[tag] += 0.01 * (double)((s->hostTimes[i] >> 4) - (s->hostTimes[i - 1] >> 4));
				count[tag] += 1;
			}
			fprintf(stderr, "[s2amd] pair-lane persistent step, one workgroup, us per phase (count):");
			for (int t = 1; t < 16; ++t)
			{
				if (count[t] > 0)
				{
					fprintf(stderr, " %s %.1f (%d);", names[t], sum[t], count[t]);
				}
			}
			fprintf(stderr, " total %.1f\n", 0.01 * (double)((s->hostTimes[n - 1] >> 4) - (s->hostTimes[0] >> 4)));
			n = 0;
		}
		fprintf(stderr, "[s2amd] persistent step, one workgroup, wall_clock64 ticks (10 ns) since kernel start:");
		for (int i = 1; i < n && i < 255; ++i)
		{
			fprintf(stderr, " %llu", s->hostTimes[i] - s->hostTimes[0]);
		}
		fprintf(stderr, "\n");
		(void)hipHostFree(s->hostTimes);
	}
	(void)hipEventDestroy(s->evBegin);
	(void)hipEventDestroy(s->evEnd);
	for (hipEvent_t e : s->evExport)
	{
		if (e)
		{
			(void)hipEventDestroy(e);
		}
	}
	for (int i = 0; i < 2; ++i)
	{
		if (s->evFork[i])
		{
			(void)hipEventDestroy(s->evFork[i]);
		}
		if (s->side[i])
		{
			(void)hipStreamDestroy(s->side[i]);
		}
	}
	for (int i = 0; i < 4; ++i)
	{
		if (s->evJoin[i])
		{
			(void)hipEventDestroy(s->evJoin[i]);
		}
	}
	(void)hipStreamDestroy(s->stream);
	delete s;
}

int s2amd_upload(s2amdSolver* s, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
				 const s2amdJoint* joints, int32_t jointCapacity)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	s->gatherIndexDirty = true;
	// the stage-3 / stage-4 arrays of a world chain (s2amd_world_upload) are sized for THAT upload's capacities
	s->worldResident = false;
	s->pairKeysValid = false;
	int rc = doUpload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
	if (rc == S2AMD_OK)
	{
		HIP_TRY(hipStreamSynchronize(s->stream));
	}
	return rc;
}

int s2amd_step_resident(s2amdSolver* s, const s2amdStepParams* params)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	return doStep(s, params);
}

int s2amd_download(s2amdSolver* s, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts, int32_t contactCapacity, s2amdJoint* joints,
				   int32_t jointCapacity)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	return doDownload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
}

int s2amd_solve(s2amdSolver* s, const s2amdStepParams* params, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts,
				int32_t contactCapacity, s2amdJoint* joints, int32_t jointCapacity)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	s->gatherIndexDirty = true;
	s->worldResident = false; // see s2amd_upload
	s->pairKeysValid = false;
	int rc = doUpload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
	if (rc)
	{
		return rc;
	}
	rc = doStep(s, params);
	if (rc)
	{
		return rc;
	}
	return doDownload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
}

int s2amd_save_bodies(s2amdSolver* s)
{
	if (!s || !s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident");
	}
	HIP_TRY(hipSetDevice(s->device));
	int rc = s->dBodiesSaved.ensure((size_t)std::max(s->bodyCapacity, 1) * sizeof(s2amdBody));
	if (rc)
	{
		return rc;
	}
	if (s->bodyCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dBodiesSaved.p, s->dBodies.p, (size_t)s->bodyCapacity * sizeof(s2amdBody), hipMemcpyDeviceToDevice, s->stream));
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->savedValid = true;
	s->savedBodyCapacity = s->bodyCapacity;
	return S2AMD_OK;
}

int s2amd_synchronize(s2amdSolver* s)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	if (s->hostError && *s->hostError != 0u)
	{
		// A persistent step enqueued under "async" lost a hand-off.  Its epilogue -- and that of every step enqueued behind
		// it, because the device-side word stays set until it is cleared here -- left the wire arrays untouched, so the
		// resident world stands where it stood before the first step that failed.  As doStep does on the synchronous path:
		// clear both error words and the hand-off buffers and keep this solver on the multi-launch strip path; the steps
		// that were dropped are the caller's to repeat (it knows how many it enqueued; stats.persistFallbacks counts).
		int rcReset = resetPersistState(s, s->stream);
		if (rcReset)
		{
			return rcReset;
		}
		HIP_TRY(hipStreamSynchronize(s->stream));
		if (s->nearHandoffNow != 0)
		{
			// (the same-XCD hand-off path was in use: the repeated steps stay on the one-launch kernel, with agent-scope stores everywhere)
			s->nearHandoffNow = 0;
			s->nearHandoffTimeouts += 1;
			s->stats.nearHandoffTimeouts = s->nearHandoffTimeouts;
			s->layoutGeneration += 1;
			return fail(S2AMD_E_DEVICE, "strip hand-off timed out inside the persistent step kernel while the same-XCD hand-off path was in use: the steps "
										"enqueued since the last s2amd_synchronize were dropped from the failing one on -- repeat them (the solver now "
										"hands off with agent-scope stores only)");
		}
		if (s->overflowKernelThisStep && !s->overflowKernelFailed)
		{
			// (the launch carried the overflow workgroup: the repeated steps run sliced, the kernel itself stays)
			s->overflowKernelFailed = true;
			s->layoutGeneration += 1;
			s->persistFallbacks += 1;
			s->stats.persistFallbacks = s->persistFallbacks;
			return fail(S2AMD_E_DEVICE, "a hand-off with the overflow workgroup of the persistent step kernel timed out: the steps enqueued since the last "
										"s2amd_synchronize were dropped from the failing one on -- repeat them (steps with overflow contacts now run sliced)");
		}
		s->persistFailed = true;
		s->persistFailedAge = 0;
		s->persistFallbacks += 1;
		s->stats.persistFallbacks = s->persistFallbacks;
		return fail(S2AMD_E_DEVICE, "strip hand-off timed out inside the persistent step kernel (workgroups not co-resident?): the steps enqueued "
									"since the last s2amd_synchronize were dropped from the failing one on -- repeat them (the solver now uses the "
									"multi-launch strip path)");
	}
	return S2AMD_OK;
}

int s2amd_restore_bodies(s2amdSolver* s)
{
	if (!s || !s->resident || !s->savedValid)
	{
		return fail(S2AMD_E_STATE, "no saved bodies");
	}
	if (s->savedBodyCapacity != s->bodyCapacity)
	{
		return fail(S2AMD_E_STATE, "the saved bodies belong to a world of another size (" + std::to_string(s->savedBodyCapacity) + " body slots, now " +
									   std::to_string(s->bodyCapacity) + ")");
	}
	HIP_TRY(hipSetDevice(s->device));
	if (s->bodyCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dBodies.p, s->dBodiesSaved.p, (size_t)s->bodyCapacity * sizeof(s2amdBody), hipMemcpyDeviceToDevice, s->stream));
	}
	return S2AMD_OK;
}

static int copyOrder(const std::vector<int>& order, const std::vector<int>& offsets, int32_t* outOrder, int32_t orderCapacity,
					 int32_t* outOffsets, int32_t colorCapacity, int32_t* count, int32_t* colorCount)
{
	int n = (int)order.size();
	int nc = offsets.empty() ? 0 : (int)offsets.size() - 1;
	if (count)
	{
		*count = n;
	}
	if (colorCount)
	{
		*colorCount = nc;
	}
	if (outOrder)
	{
		if (orderCapacity < n)
		{
			return fail(S2AMD_E_CAPACITY, "order buffer too small");
		}
		std::copy(order.begin(), order.end(), outOrder);
	}
	if (outOffsets)
	{
		if (colorCapacity < nc + 1)
		{
			return fail(S2AMD_E_CAPACITY, "colour offset buffer too small");
		}
		std::copy(offsets.begin(), offsets.end(), outOffsets);
		if (offsets.empty())
		{
			outOffsets[0] = 0;
		}
	}
	return S2AMD_OK;
}

int s2amd_get_contact_order(s2amdSolver* s, int32_t* order, int32_t orderCapacity, int32_t* colorOffsets, int32_t colorCapacity,
							int32_t* constraintCount, int32_t* colorCount)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	// the sweep order holds every potential constraint; what the step swept -- and what the reference would have gathered --
	// are the ones whose manifold had points
	if (!s->pointsKnown)
	{
		int rc = fetchPointCounts(s);
		if (rc)
		{
			return rc;
		}
	}
	const SweepSet& cs = s->contacts;
	std::vector<int> active, offsets(1, 0);
	active.reserve(cs.order.size());
	// (some launch batches are sequential tails -- a hub body's constraints, or a group's tiny colours, swept one after the other by one
	// wave: each of their constraints is a colour of its own in what is reported, so that a reported colour never holds two
	// constraints on one writable body)
	std::vector<uint8_t> sequential(cs.order.size(), 0);
	for (const HostGroupTable* t : {&s->hGroups, &s->hResident, &s->hStripA, &s->hStripB, &s->hContactTail})
	{
		for (const int4& batch : t->cBatches)
		{
			if (batch.z == 1)
			{
				for (int k2 = std::max(batch.x, 0); k2 < batch.y && k2 < (int)sequential.size(); ++k2)
				{
					sequential[(size_t)k2] = 1;
				}
			}
		}
	}
	size_t k = 0;
	for (size_t c = 0; c + 1 < cs.colorOffsets.size(); ++c)
	{
		for (; k < (size_t)cs.colorOffsets[c + 1]; ++k)
		{
			if (cs.order[k] >= 0 && s->hContactPoints[(size_t)cs.order[k]] > 0) // (-1: a free position of the slack layout)
			{
				active.push_back(cs.order[k]);
				if (sequential[k])
				{
					offsets.push_back((int)active.size());
				}
			}
		}
		if ((int)active.size() > offsets.back())
		{
			offsets.push_back((int)active.size());
		}
	}
	return copyOrder(active, offsets, order, orderCapacity, colorOffsets, colorCapacity, constraintCount, colorCount);
}

// The bodies the sweeps of the last structure write -- the nodes two constraints of one colour must not share.  Velocity-class sweeps
// (every solver's contact passes but the position passes below) write the bodies with mass; the position-class sweeps of PGS_NGS /
// PGS_NGS_Block / TGS_NGS also store rot = normalize(rot) of immovable bodies (src/solve_common.c:383-392), so there only a static body
// whose rotation is a fixed point of that normalisation is read-only (solver_step.cpp: the flags of the upload).
int s2amd_get_writable_bodies(s2amdSolver* s, uint8_t* writable, int32_t bodyCapacity, int32_t* solverClass)
{
	if (!s || bodyCapacity < 0 || (bodyCapacity > 0 && !writable))
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if ((int)s->hBodyFlags.size() > bodyCapacity)
	{
		return fail(S2AMD_E_CAPACITY, "the solver holds " + std::to_string(s->hBodyFlags.size()) + " bodies");
	}
	const int cls = s->orderSolverClass == 1 ? 1 : 0;
	const uint32_t bit = cls == 1 ? S2F_WRITE_POS : S2F_WRITE_VEL;
	for (size_t i = 0; i < s->hBodyFlags.size(); ++i)
	{
		writable[i] = (s->hBodyFlags[i] & bit) != 0 ? 1 : 0;
	}
	if (solverClass)
	{
		*solverClass = cls;
	}
	return S2AMD_OK;
}

int s2amd_get_joint_order(s2amdSolver* s, int32_t* order, int32_t orderCapacity, int32_t* colorOffsets, int32_t colorCapacity, int32_t* jointCount,
						  int32_t* colorCount)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	return copyOrder(s->joints.order, s->joints.colorOffsets, order, orderCapacity, colorOffsets, colorCapacity, jointCount, colorCount);
}

int s2amd_get_strip_owners(s2amdSolver* s, int32_t* ownerStrip, int32_t* onSeam, int32_t capacity, int32_t* stripCount)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	const IncrementalStrips& m = s->stripInc;
	const int nb = s->bodyCapacity;
	if ((ownerStrip || onSeam) && capacity < nb)
	{
		return fail(S2AMD_E_CAPACITY, "strip owner buffer too small");
	}
	if (stripCount)
	{
		*stripCount = m.valid ? (int)m.stripBodyCount.size() : 0;
	}
	for (int i = 0; i < nb; ++i)
	{
		if (ownerStrip)
		{
			ownerStrip[i] = (m.valid && i < (int)m.ownerStrip.size()) ? m.ownerStrip[(size_t)i] : -1;
		}
		if (onSeam)
		{
			onSeam[i] = -1;
		}
	}
	if (onSeam && m.valid)
	{
		for (size_t sm = 0; sm < m.seamGroupOf.size(); ++sm)
		{
			const int g = m.seamGroupOf[sm];
			if (g < 0)
			{
				continue;
			}
			for (const auto& kv : m.seamSlot[(size_t)g])
			{
				if (kv.first >= 0 && kv.first < nb && m.ownerStrip[(size_t)kv.first] >= 0)
				{
					onSeam[kv.first] = (int)sm;
				}
			}
		}
	}
	return S2AMD_OK;
}

int s2amd_get_stats(s2amdSolver* s, s2amdStepStats* stats)
{
	if (!s || !stats)
	{
		return fail(S2AMD_E_INVALID, "null argument");
	}
	*stats = s->stats;
	return S2AMD_OK;
}

int s2amd_export_poses(s2amdSolver* s, void* devicePoses, int32_t capacity)
{
	if (!s || !s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident");
	}
	if (!devicePoses || capacity < s->bodyCapacity)
	{
		return fail(S2AMD_E_CAPACITY, "pose buffer missing or too small");
	}
	HIP_TRY(hipSetDevice(s->device));
	launchExportPoses(s->stream, (const s2amdBody*)s->dBodies.p, s->bodyCapacity, devicePoses);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

int s2amd_export_poses_async(s2amdSolver* s, void* devicePoses, int32_t capacity, int32_t slot)
{
	if (!s || !s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident");
	}
	if (!devicePoses || capacity < s->bodyCapacity || slot < 0 || slot >= 4)
	{
		return fail(S2AMD_E_CAPACITY, "pose buffer missing or too small, or slot outside 0..3");
	}
	HIP_TRY(hipSetDevice(s->device));
	if (!s->evExport[slot])
	{
		HIP_TRY(hipEventCreateWithFlags(&s->evExport[slot], hipEventDisableTiming));
	}
	launchExportPoses(s->stream, (const s2amdBody*)s->dBodies.p, s->bodyCapacity, devicePoses);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(s->evExport[slot], s->stream));
	return S2AMD_OK;
}

int s2amd_export_bodies_async(s2amdSolver* s, void* deviceRecords, int32_t capacity, int32_t slot)
{
	if (!s || !s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident");
	}
	if (!deviceRecords || capacity < s->bodyCapacity || slot < 0 || slot >= 4)
	{
		return fail(S2AMD_E_CAPACITY, "record buffer missing or too small, or slot outside 0..3");
	}
	HIP_TRY(hipSetDevice(s->device));
	if (!s->evExport[slot])
	{
		HIP_TRY(hipEventCreateWithFlags(&s->evExport[slot], hipEventDisableTiming));
	}
	launchExportPoses(s->stream, (const s2amdBody*)s->dBodies.p, s->bodyCapacity, deviceRecords, 1);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(s->evExport[slot], s->stream));
	return S2AMD_OK;
}

int s2amd_export_wait(s2amdSolver* s, int32_t slot)
{
	if (!s || slot < 0 || slot >= 4 || !s->evExport[slot])
	{
		return fail(S2AMD_E_STATE, "no export recorded in this slot");
	}
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipEventSynchronize(s->evExport[slot]));
	return S2AMD_OK;
}

int s2amd_device_alloc(s2amdSolver* s, uint64_t bytes, void** devicePtr)
{
	if (!s || !devicePtr)
	{
		return fail(S2AMD_E_INVALID, "null argument");
	}
	*devicePtr = nullptr;
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipMalloc(devicePtr, bytes > 0 ? (size_t)bytes : 16));
	HIP_TRY(hipMemsetAsync(*devicePtr, 0, bytes > 0 ? (size_t)bytes : 16, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

int s2amd_device_free(s2amdSolver* s, void* devicePtr)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	if (devicePtr)
	{
		HIP_TRY(hipSetDevice(s->device));
		HIP_TRY(hipStreamSynchronize(s->stream));
		HIP_TRY(hipFree(devicePtr));
	}
	return S2AMD_OK;
}

int s2amd_device_read(s2amdSolver* s, void* hostDst, const void* deviceSrc, uint64_t bytes)
{
	if (!s || (bytes > 0 && (!hostDst || !deviceSrc)))
	{
		return fail(S2AMD_E_INVALID, "null argument");
	}
	HIP_TRY(hipSetDevice(s->device));
	if (bytes > 0)
	{
		HIP_TRY(hipMemcpyAsync(hostDst, deviceSrc, (size_t)bytes, hipMemcpyDeviceToHost, s->stream));
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

int s2amd_measure_dominant(s2amdSolver* s, const s2amdStepParams* params, int32_t repeats, float* usPerLaunch, int32_t* launchesPerSweep,
						   int32_t* constraintsPerLaunch)
{
	if (!s || !params || !usPerLaunch || repeats <= 0)
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (!s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident");
	}
	HIP_TRY(hipSetDevice(s->device));
	// make sure plan, structure and the device op list are current (one ordinary step)
	int rc = doStep(s, params);
	if (rc)
	{
		return rc;
	}
	const StepPlan& plan = s->plan;
	int dominant = -1;
	for (int i = 0; i < (int)plan.ops.size(); ++i)
	{
		if (Executor::isSolveSweep(plan.ops[(size_t)i].code))
		{
			dominant = i;
			break;
		}
	}
	Executor q{s, s->stream, plan, isPositionSolver(params->solverType) ? 1 : 0, false};
	q.msg = messageEligible(s, params->solverType);
	const bool global = s->contacts.globalCount > 0 && dominant >= 0;
	const bool strips = !global && s->contacts.stripCount > 0 && dominant >= 0;
	int pkind = -1, pwarm = -1;
	const bool persistent = strips && q.persistPlan(pkind, pwarm);
	hipGraph_t g = nullptr;
	hipGraphExec_t ge = nullptr;
	s->launchCounter = 0;
	HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
	for (int r = 0; r < repeats; ++r)
	{
		if (global)
		{
			q.runGlobalOp(dominant);
		}
		else if (persistent)
		{
			q.runPersistent(pkind, pwarm, true, q.selfContainedStrips()); // (the kernel the step launches: alone it IS the step, run after run)
		}
		else if (strips)
		{
			q.launchStripGroups(s->dStripA, dominant, 1, false);
			if (s->contacts.seamCount > 0)
			{
				q.launchStripGroups(s->dStripB, dominant, 1, false);
			}
		}
		else if (s->dResident.view.groupCount > 0)
		{
			// the whole step of the resident islands (wideIslandKernel / islandStepKernel, or the interpreter on the same table), in the
			// form the step launches it
			q.runResidentGroups(q.selfContainedIslands());
		}
		else if (s->dGroups.view.groupCount > 0)
		{
			launchGroupKernel(s->stream, s->cv, s->jv, s->bv, s->dGroups.view, q.deviceOps(), (int)plan.ops.size(), plan.sc,
							  (s2amdContact*)s->dContacts.p, s->dGroups.maxBodies, plan.usesDq0 ? 1 : 0);
			s->launchCounter += 1;
		}
	}
	hipError_t ce = hipStreamEndCapture(s->stream, &g);
	if (ce != hipSuccess)
	{
		return fail(S2AMD_E_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
	}
	int launches = s->launchCounter;
	if (launches == 0)
	{
		(void)hipGraphDestroy(g);
		*usPerLaunch = 0.0f;
		return S2AMD_OK;
	}
	HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	HIP_TRY(hipGraphLaunch(ge, s->stream)); // warm
	HIP_TRY(hipStreamSynchronize(s->stream));
	const int reps = 5;
	HIP_TRY(hipEventRecord(s->evBegin, s->stream));
	for (int r = 0; r < reps; ++r)
	{
		HIP_TRY(hipGraphLaunch(ge, s->stream));
	}
	HIP_TRY(hipEventRecord(s->evEnd, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	float ms = 0.0f;
	HIP_TRY(hipEventElapsedTime(&ms, s->evBegin, s->evEnd));
	(void)hipGraphExecDestroy(ge);
	(void)hipGraphDestroy(g);
	*usPerLaunch = 1e3f * ms / (float)(reps * launches);
	if (persistent)
	{
		// the whole step is one launch (+ the memset of the hand-off buffers in front of it)
		*usPerLaunch = 1e3f * ms / (float)(reps * repeats);
		launches = repeats;
	}
	if (launchesPerSweep)
	{
		*launchesPerSweep = launches / repeats;
	}
	if (constraintsPerLaunch)
	{
		// (constraints that are swept: the potential ones without manifold points and the slack positions do not count)
		*constraintsPerLaunch = global		 ? s->activeContacts / std::max(launches / repeats, 1)
								: persistent ? std::min(s->contacts.stripCount, s->activeContacts) * plan.solveSweeps // constraint-sweeps of the whole-step launch
								: strips	 ? std::min(s->contacts.stripCount, s->activeContacts) / std::max(launches / repeats, 1)
											 : s->activeContacts;
	}
	return S2AMD_OK;
}

int s2amd_set_option(s2amdSolver* s, const char* key, int32_t value)
{
	if (!s || !key)
	{
		return fail(S2AMD_E_INVALID, "null argument");
	}
	asyncDrop(s); // (a structure being built under the options as they were)
	if (strcmp(key, "async_build") == 0)
	{
		s->optAsyncBuild = std::max(0, std::min(value, 2));
	}
	else if (strcmp(key, "async_build_delay") == 0)
	{
		s->optAsyncBuildDelay = std::max(0, value);
	}
	else if (strcmp(key, "graph") == 0)
	{
		s->optGraph = value;
	}
	else if (strcmp(key, "graph_min_launches") == 0)
	{
		s->optGraphMinLaunches = std::max(0, value);
	}
	else if (strcmp(key, "profile") == 0)
	{
		s->optProfile = value;
	}
	else if (strcmp(key, "message") == 0)
	{
		s->optMessage = value;
		s->structureDirty = true; // the message tables are only built when the option is on
	}
	else if (strcmp(key, "body_warm") == 0)
	{
		s->optBodyWarm = value;
	}
	else if (strcmp(key, "pair_query_graph") == 0)
	{
		s->pairQuery.disabled = value == 0; // (0: the pair query's kernels are enqueued one by one instead of replayed from a captured graph: diagnostics)
	}
	else if (strcmp(key, "tail_tiny_colour") == 0)
	{
		s->optTailTinyColour = std::max(0, value);
		s->structureDirty = true;
	}
	else if (strcmp(key, "group_tiny_colour") == 0)
	{
		s->optGroupTinyColour = std::max(0, value);
		s->structureDirty = true;
	}
	else if (strcmp(key, "generic_place") == 0)
	{
		s->optGenericPlace = value;
		s->structureDirty = true;
	}
	else if (strcmp(key, "flip_colours") == 0)
	{
		s->optFlipColours = value;
	}
	else if (strcmp(key, "group_patience") == 0)
	{
		s->optGroupPatience = value;
		s->groupPatienceNow = 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "groups") == 0)
	{
		s->optGroups = value;
		s->structureDirty = true;
	}
	else if (strcmp(key, "max_group_bodies") == 0)
	{
		s->stripsRejected = false;
		s->optMaxGroupBodies = std::max(1, std::min(value, 2816)) /* 56 B of LDS per body with XPBD's dq0 */;
		s->maxGroupBodiesSet = true;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strips") == 0)
	{
		s->stripsRejected = false;
		s->optStrips = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "async") == 0)
	{
		s->optAsync = value != 0;
	}
	else if (strcmp(key, "fork") == 0)
	{
		s->optFork = value != 0;
	}
	else if (strcmp(key, "persist_spin_limit") == 0)
	{
		s->optPersistSpinLimit = std::max(256, value);
		s->stripsRejected = false;
		s->structureDirty = true;
	}
	else if (strcmp(key, "persist_debug") == 0)
	{
		s->stripsRejected = false;
		s->optPersistDebug = value;
		s->structureDirty = true;
	}
	else if (strcmp(key, "seam_regs") == 0)
	{
		s->optSeamRegs = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "wide") == 0)
	{
		s->optWide = value != 0;
		s->structureDirty = true; // (re-captures the step graph)
	}
	else if (strcmp(key, "pair_lanes") == 0)
	{
		s->optPairLanes = value != 0;
		s->structureDirty = true; // (re-captures the step graph)
	}
	else if (strcmp(key, "strip_slack") == 0)
	{
		s->optStripSlack = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_retry") == 0)
	{
		s->optStripRetry = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "persist") == 0)
	{
		s->stripsRejected = false;
		s->optPersist = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_patience") == 0)
	{
		s->optStripPatience = std::max(0, value);
		s->stripPatienceNow = s->optStripPatience;
		s->stripPatienceSet = true;
	}
	else if (strcmp(key, "near_handoff") == 0)
	{
		s->optNearHandoff = value != 0;
		s->nearHandoffNow = s->optNearHandoff;
	}
	else if (strcmp(key, "strip_adopt") == 0)
	{
		s->optStripAdopt = value != 0; // a constraint-free body moves to the strip of the body it first touches (IncrementalStrips)
	}
	else if (strcmp(key, "prebuild_solver") == 0)
	{
		s->optPrebuildSolver = value; // s2amd_world_upload builds the structure for this s2amdSolverType (-1: the first step does)
	}
	else if (strcmp(key, "jacobi_persist") == 0)
	{
		s->optJacobiPersist = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "jacobi_min_constraints") == 0)
	{
		s->optJacobiMinConstraints = std::max(0, value);
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_overflow") == 0)
	{
		s->optOverflow = value != 0; // a contact that fits nowhere in the strips: overflow position + sliced steps + worker-thread build (IncrementalStrips)
		s->structureDirty = true;
	}
	else if (strcmp(key, "cost_strip_round_ns") == 0 || strcmp(key, "cost_launch_ns") == 0 || strcmp(key, "cost_tail_visit_ns") == 0)
	{
		// the hub rule's unit costs in nanoseconds (solver_internal.h: HubCosts): measured elsewhere than on the table's architectures
		float& f = key[5] == 's' ? s->hubCosts.stripRoundUs : (key[5] == 'l' ? s->hubCosts.launchUs : s->hubCosts.tailVisitUs);
		f = 1e-3f * (float)std::max(value, 1);
		s->structureDirty = true;
	}
	else if (strcmp(key, "tree_stream") == 0)
	{
		s->optTreeStream = value != 0 ? 1 : 0;
	}
	else if (strcmp(key, "pairs_in_step") == 0)
	{
		s->optPairsInStep = value != 0; // the stage-1 pair query enqueued behind every world step (world.hip)
		s->pairCacheValid = false;
	}
	else if (strcmp(key, "overflow_kernel") == 0)
	{
		s->optOverflowKernel = value != 0; // overflow contacts inside the persistent launch (one more workgroup) instead of sliced steps
		s->overflowKernelFailed = false;
		s->layoutGeneration += 1;
	}
	else if (strcmp(key, "stage_joints") == 0)
	{
		s->optStageJoints = value != 0;
	}
	else if (strcmp(key, "step_readback") == 0)
	{
		s->optStepReadback = value != 0;
	}
	else if (strcmp(key, "self_contained") == 0)
	{
		s->optSelfContained = value != 0;
	}
	else if (strcmp(key, "strip_body_warm") == 0)
	{
		s->optWideBodyWarm = value != 0;
	}
	else if (strcmp(key, "self_contained_strips") == 0)
	{
		if (s->optSelfContainedStrips != (value != 0 ? 1 : 0))
		{
			s->structureDirty = true; // (the overflow region behind the strips is laid out for the multi-launch form of the step only)
		}
		s->optSelfContainedStrips = value != 0;
	}
	else if (strcmp(key, "free_body_groups") == 0)
	{
		s->optFreeBodyGroups = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "persist_retry") == 0)
	{
		s->optPersistRetry = std::max(0, value);
		s->persistRetryAfter = std::max(1, value);
	}
	else if (strcmp(key, "generic") == 0)
	{
		s->stripsRejected = false;
		s->optGeneric = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strips_any_solver") == 0)
	{
		s->stripsRejected = false;
		s->optStripsAnySolver = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "island_resident") == 0)
	{
		s->optIslandResident = value != 0;
		s->residentRejected = false;
		s->structureDirty = true;
	}
	else if (strcmp(key, "incremental") == 0)
	{
		s->optIncremental = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "defer") == 0)
	{
		s->optDefer = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_lean") == 0)
	{
		s->stripsRejected = false;
		s->optStripLean = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_bodies") == 0)
	{
		s->stripsRejected = false;
		s->optStripBodies = std::max(1, value);
		s->stripBodiesSet = true;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_min_bodies") == 0)
	{
		s->stripsRejected = false;
		s->optStripMinBodies = std::max(0, value);
		s->stripMinBodiesSet = true;
		s->structureDirty = true;
	}
	else if (strcmp(key, "pack_group_bodies") == 0)
	{
		s->optPackGroupBodies = std::max(1, value);
		s->packGroupBodiesSet = true;
		s->structureDirty = true;
	}
	else
	{
		return fail(S2AMD_E_INVALID, std::string("unknown option ") + key);
	}
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop

