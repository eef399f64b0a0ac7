// upload / step / download of a resident world: host shadows of the graph, hipGraph capture and replay, timing.
#include "solver_executor.h"

void destroyGraph(s2amdSolver* s)
{
	if (s->graphExec)
	{
		(void)hipGraphExecDestroy(s->graphExec);
		s->graphExec = nullptr;
	}
	if (s->graph)
	{
		(void)hipGraphDestroy(s->graph);
		s->graph = nullptr;
	}
	s->graphKey = 0;
}

int resetPersistState(s2amdSolver* s, hipStream_t st)
{
	if (s->hostError)
	{
		*s->hostError = 0u;
	}
	if (s->persist.deviceError)
	{
		HIP_TRY(hipMemsetAsync(s->persist.deviceError, 0, 256, st)); // the error word and the self-contained kernel's commit counter (PersistView::state)
	}
	if (s->dGranules.p && s->granuleBytes)
	{
		HIP_TRY(hipMemsetAsync(s->dGranules.p, 0, s->granuleBytes, st));
	}
	if (s->jacobi.deviceError)
	{
		HIP_TRY(hipMemsetAsync(s->jacobi.deviceError, 0, 256, st));
	}
	if (s->dJacobiGran.p && s->jacobiGranBytes)
	{
		HIP_TRY(hipMemsetAsync(s->dJacobiGran.p, 0, s->jacobiGranBytes, st));
	}
	s->selfStepsSinceReset = 0;
	return S2AMD_OK;
}

int refreshShadows(s2amdSolver* s, const s2amdBody* bodies, int nb, const s2amdContact* contacts, int nc, const s2amdJoint* joints, int nj,
				   const s2amdPairState* pairs)
{
	const bool newWorld = nb != (int)s->hBodyFlags.size() || nc != (int)s->hContactA.size() || nj != (int)s->hJointType.size();
	bool changed = s->structureDirty || newWorld;
	std::vector<uint32_t> flags((size_t)nb);
	s->hBodyLive.assign((size_t)nb, 0);
	s->hBodyStatic.assign((size_t)nb, 0);
	for (int i = 0; i < nb; ++i)
	{
		const s2amdBody& b = bodies[i];
		uint32_t f = 0;
		s->hBodyLive[i] = b.type != S2AMD_BODY_FREE;
		s->hBodyStatic[i] = b.type == S2AMD_BODY_STATIC;
		if (b.type != S2AMD_BODY_FREE)
		{
			bool massless = b.invMass == 0.0f && b.invI == 0.0f;
			if (!massless)
			{
				f |= S2F_WRITE_VEL;
			}
			// position sweeps store rot = normalize(rot) even for immovable bodies
			// (solve_common.c:383-392): only a static body whose rot is a fixed point of the
			// normalisation can be treated as read-only there
			if (!(massless && b.type == S2AMD_BODY_STATIC && rotIsFixedPoint(b.rot[0], b.rot[1])))
			{
				f |= S2F_WRITE_POS;
			}
		}
		flags[i] = f;
	}
	if (!changed && flags != s->hBodyFlags)
	{
		changed = true;
	}
	s->hBodyFlags.swap(flags);

	// joints first: whether the graph changed for a reason other than created contacts decides how those are handled
	if ((int)s->hJointType.size() != nj)
	{
		s->hJointType.assign(nj, S2AMD_JOINT_FREE);
		s->hJointA.assign(nj, -1);
		s->hJointB.assign(nj, -1);
	}
	for (int i = 0; i < nj; ++i)
	{
		const s2amdJoint& j = joints[i];
		if (!changed && (s->hJointType[i] != j.type || s->hJointA[i] != j.bodyA || s->hJointB[i] != j.bodyB))
		{
			changed = true;
		}
		s->hJointType[i] = j.type;
		s->hJointA[i] = j.bodyA;
		s->hJointB[i] = j.bodyB;
		if (j.type != S2AMD_JOINT_FREE)
		{
			if (j.type != S2AMD_JOINT_REVOLUTE && j.type != S2AMD_JOINT_MOUSE)
			{
				return fail(S2AMD_E_INVALID, "joint " + std::to_string(i) + " has an unknown type");
			}
			if (j.bodyB < 0 || j.bodyB >= nb || (j.type == S2AMD_JOINT_REVOLUTE && (j.bodyA < 0 || j.bodyA >= nb)))
			{
				return fail(S2AMD_E_INVALID, "joint " + std::to_string(i) + " has an invalid body index");
			}
		}
	}

	if ((int)s->hContactA.size() != nc)
	{
		s->hContactA.assign(nc, -1);
		s->hContactB.assign(nc, -1);
		s->hContactPoints.assign(nc, 0);
		s->hContactEdge.assign(nc, 0);
		s->hContactDead.assign(nc, 0);
	}
	bool pointCountsMoved = false;
	int active = 0;
	std::vector<ContactChange> created; // contacts that appeared (or whose slot now holds another pair), in pool order
	std::vector<int32_t> died;			// ... that were destroyed since the last upload
	std::vector<ContactChange> deferred; // created without points where nothing can be placed: watched, not structural (optDefer)
	std::vector<ContactChange> flipped;	 // watched manifolds with their first points, under a structure built for s2Solve_Jacobi
	const bool placeFlips = s->inc.valid && s->inc.ignoreColours && s->optIncremental != 0 && !s->structureDirty && !newWorld;
	// ... and under the soft contact solvers' strips a flipped manifold takes a free position of a strip or seam round where it fits
	const bool stripFlips = s->inc.valid && s->stripInc.valid && s->optIncremental != 0 && !s->structureDirty && !newWorld;
	bool hubTouched = false;			// something happened to a contact on a hub body: decided by a rebuild
	auto onHub = [&](int a, int b) {
		return !s->hBodyHub.empty() && (int)s->hBodyHub.size() == nb && a >= 0 && b >= 0 && a < nb && b < nb && (s->hBodyHub[(size_t)a] || s->hBodyHub[(size_t)b]);
	};
	for (int i = 0; i < nc; ++i)
	{
		const s2amdContact& c = contacts[i];
		int pc = c.pointCount > 0 ? c.pointCount : 0;
		pointCountsMoved = pointCountsMoved || s->hContactPoints[i] != pc;
		const bool valid = c.bodyA >= 0 && c.bodyA < nb && c.bodyB >= 0 && c.bodyB < nb;
		if (pc > 0 && (!valid || pc > 2))
		{
			return fail(S2AMD_E_INVALID, "contact " + std::to_string(i) + " has an invalid body index or point count");
		}
		// a potential constraint: the world chain says which pair slots are live; a bare solver input does not, there a slot
		// without points counts when it names two distinct live bodies (a free pool slot names none)
		const bool edge = pc > 0 || (pairs ? (pairs[i].shapeA >= 0 && valid)
										   : (valid && c.bodyA != c.bodyB && bodies[c.bodyA].type != S2AMD_BODY_FREE && bodies[c.bodyB].type != S2AMD_BODY_FREE));
		const int oldPoints = s->hContactPoints[i];
		s->hContactPoints[i] = pc;
		active += pc > 0 ? 1 : 0;
		if (!edge && s->hContactEdge[i] && !newWorld)
		{
			// The contact was destroyed (src/contact.c:205-229).  Its entry stays in the structure as a dead constraint
			// -- pointCount 0 is a no-op -- with the bodies it had, until the slot is used again or the structure is rebuilt
			// for another reason: the world chain, where pairs separate on the device, learns of a destruction no earlier,
			// and both routes must sweep in the same order to stay bit-identical (tests/test_gpu_dropin.py).
			if (!s->hContactDead[i])
			{
				died.push_back(i);
			}
			s->hContactDead[i] = 1;
			unwatchSlot(s, i);
			continue;
		}
		if (edge && pc == 0 && !newWorld && (!s->hContactEdge[i] || s->hContactA[i] != c.bodyA || s->hContactB[i] != c.bodyB) &&
			canDeferCreated(s, i, c.bodyA, c.bodyB))
		{
			deferred.push_back(ContactChange{i, c.bodyA, c.bodyB});
			continue;
		}
		if (edge && pc > 0 && (placeFlips || (stripFlips && (stripCanPlace(s, c.bodyA, c.bodyB) || overflowCanPlace(s, c.bodyA, c.bodyB))) || (!newWorld && onHub(c.bodyA, c.bodyB) && tailCanPlace(s, c.bodyA, c.bodyB))) &&
			(!s->hContactEdge[i] || s->hContactA[i] != c.bodyA || s->hContactB[i] != c.bodyB) && canDeferCreated(s, i, c.bodyA, c.bodyB))
		{
			// created AND touching at first sight (the caller ran stage 3 itself): the world chain, which sees the contact created
			// without points and its manifold gain them in its own stage 3, watches it first and places it with the flips --
			// same sequence here, so that both routes hand out the same positions
			deferred.push_back(ContactChange{i, c.bodyA, c.bodyB});
			flipped.push_back(ContactChange{i, c.bodyA, c.bodyB});
			continue;
		}
		if (edge && (!s->hContactEdge[i] || s->hContactA[i] != c.bodyA || s->hContactB[i] != c.bodyB))
		{
			created.push_back(ContactChange{i, c.bodyA, c.bodyB}); // (shadows of this slot are written below, after the old entry was found)
			hubTouched = hubTouched || (!s->hBodyHub.empty() && (int)s->hBodyHub.size() == nb && (s->hBodyHub[(size_t)c.bodyA] || s->hBodyHub[(size_t)c.bodyB]));
			continue;
		}
		if (edge && !s->hContactWatched.empty() && s->hContactWatched[(size_t)i] && (oldPoints > 0) != (pc > 0))
		{
			// a watched manifold (on a hub body, or deferred) gained or lost its points (solver_internal.h: hContactWatched)
			if (placeFlips || (stripFlips && (stripCanPlace(s, c.bodyA, c.bodyB) || overflowCanPlace(s, c.bodyA, c.bodyB))) || (!newWorld && tailCanPlace(s, c.bodyA, c.bodyB)))
			{
				if (pc > 0 && i < (int)s->inc.positionOfSlot.size() && s->inc.positionOfSlot[(size_t)i] == -1)
				{
					flipped.push_back(ContactChange{i, c.bodyA, c.bodyB}); // s2Solve_Jacobi: a position and two list entries, no colour to find
				}
			}
			else
			{
				hubTouched = true;
				s->dirtyReason = "watched manifold flipped";
				s->dirtyByWatched = true;
				s->dirtyByGroups = s->dirtyByGroups || ownedByLdsGroup(s, c.bodyA) || ownedByLdsGroup(s, c.bodyB);
			}
		}
		if (!edge && s->hContactEdge[i])
		{
			changed = true; // (newWorld)
		}
		s->hContactEdge[i] = edge ? 1 : 0;
		s->hContactDead[i] = 0;
	}
	s->deadUnknown = false;
	s->pointsKnown = true;
	s->activeContacts = active;
	if (!created.empty())
	{
		// created contacts: a place in the existing structure when that is all that happened and they fit, else a rebuild
		const bool placed = !changed && !hubTouched && incrementalApply(s, created);
		for (const ContactChange& ch : created)
		{
			s->hContactA[(size_t)ch.slot] = ch.a;
			s->hContactB[(size_t)ch.slot] = ch.b;
			s->hContactEdge[(size_t)ch.slot] = 1;
			s->hContactDead[(size_t)ch.slot] = 0;
		}
		if (placed)
		{
			if (s->inc.placedInGlobalPart)
			{
				noteGraphTouched(s);
			}
		}
		else
		{
			changed = true;
		}
	}
	changed = changed || hubTouched;
	if (!changed)
	{
		for (const ContactChange& ch : deferred)
		{
			deferCreated(s, ch.slot, ch.a, ch.b);
		}
		// (after this step's created contacts, before its destroyed ones: the order the world chain does it in)
		if (!flipped.empty())
		{
			if (incrementalApply(s, flipped))
			{
				if (s->inc.placedInGlobalPart)
				{
					noteGraphTouched(s); // (a place in the strips leaves the strips where they are)
				}
			}
			else
			{
				changed = true;
			}
		}
	}
	else
	{
		for (const ContactChange& ch : deferred)
		{
			// the structure is rebuilt anyway: an ordinary potential constraint of the new one
			s->hContactA[(size_t)ch.slot] = ch.a;
			s->hContactB[(size_t)ch.slot] = ch.b;
			s->hContactEdge[(size_t)ch.slot] = 1;
			s->hContactDead[(size_t)ch.slot] = 0;
		}
	}
	if (!changed)
	{
		// Destroyed contacts give their places back AFTER this step's created ones were placed: the world chain learns of the
		// pairs its stage 3 separates only after the step they separate in, and both routes must build the same structure.
		incrementalRemove(s, died.data(), (int)died.size());
		int rcFlush = incrementalFlush(s);
		if (rcFlush)
		{
			return rcFlush;
		}
	}
	if (!changed && pointCountsMoved)
	{
		s->residentAllTwoPoints = residentAllTwoPoints(s) ? 1 : 0;
	}
	if (!changed && pointCountsMoved && s->persistValid)
	{
		// same graph, but a manifold went from two points to one or back: the persistent kernel's two-point
		// fast path is a property of the point counts, not of the graph (the step graph is keyed on it)
		s->persist.allTwoPoints = stripsAllTwoPoints(s) ? 1 : 0;
	}
	if (changed)
	{
		noteGraphChanged(s, newWorld);
	}
	return S2AMD_OK;
}

int doUpload(s2amdSolver* s, const s2amdBody* bodies, int nb, const s2amdContact* contacts, int nc, const s2amdJoint* joints, int nj,
			 const s2amdPairState* pairs)
{
	if (nb < 0 || nc < 0 || nj < 0 || (nb > 0 && !bodies) || (nc > 0 && !contacts) || (nj > 0 && !joints))
	{
		return fail(S2AMD_E_INVALID, "null array with non-zero count");
	}
	HIP_TRY(hipSetDevice(s->device));
	asyncDrop(s); // (a structure being built for the arrays as they were)
	int rc = refreshShadows(s, bodies, nb, contacts, nc, joints, nj, pairs);
	if (rc)
	{
		return rc;
	}
	bool grew = false;
	if ((rc = s->dBodies.ensure((size_t)std::max(nb, 1) * sizeof(s2amdBody), &grew)) != 0)
	{
		return rc;
	}
	if ((rc = s->dContacts.ensure((size_t)std::max(nc, 1) * sizeof(s2amdContact), &grew)) != 0)
	{
		return rc;
	}
	if ((rc = s->dJoints.ensure((size_t)std::max(nj, 1) * sizeof(s2amdJoint), &grew)) != 0)
	{
		return rc;
	}
	if ((rc = s->dBodyFlags.ensure((size_t)std::max(nb, 1) * sizeof(uint32_t), &grew)) != 0)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
		s->savedValid = false;
		s->structureDirty = true; // dBodyFlags may have moved
	}
	if (nb != s->bodyCapacity)
	{
		s->savedValid = false; // s2amd_save_bodies took a snapshot of a world of another size
	}
	s->bodyCapacity = nb;
	s->contactCapacity = nc;
	s->jointCapacity = nj;
	if ((rc = carveBodies(s, nb)) != 0)
	{
		return rc;
	}
	if (nb > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dBodies.p, bodies, (size_t)nb * sizeof(s2amdBody), hipMemcpyHostToDevice, s->stream));
		// dBodyFlags is written by buildStructure (it adds the LDS-group ownership bits)
	}
	if (nc > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dContacts.p, contacts, (size_t)nc * sizeof(s2amdContact), hipMemcpyHostToDevice, s->stream));
	}
	if (nj > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dJoints.p, joints, (size_t)nj * sizeof(s2amdJoint), hipMemcpyHostToDevice, s->stream));
	}
	s->resident = true;
	return S2AMD_OK;
}

__global__ void writeConstraintIndexKernel(s2amdContact* wire, int n, const int* gatherIndex)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		wire[i].constraintIndex = gatherIndex[i];
	}
}

int doStep(s2amdSolver* s, const s2amdStepParams* params)
{
	if (!params)
	{
		return fail(S2AMD_E_INVALID, "null params");
	}
	if (params->solverType < 0 || params->solverType >= s2amd_solverTypeCount)
	{
		return fail(S2AMD_E_INVALID, "unknown solver type " + std::to_string(params->solverType));
	}
	if (!s->resident)
	{
		return fail(S2AMD_E_STATE, "s2amd_step_resident called before s2amd_upload");
	}
	HIP_TRY(hipSetDevice(s->device));
	s->stats = s2amdStepStats{};
	s->stepCounter += 1;
	buildPlan(s, params);
	{
		// a structure a worker thread has built for this world falls due at a fixed step after its request (solver_async.cpp)
		int rcAdopt = S2AMD_OK;
		(void)asyncAdopt(s, params->solverType, &rcAdopt);
		if (rcAdopt)
		{
			return rcAdopt;
		}
	}
	if (s->stripInc.valid && s->stripInc.overflowUsed > 0 && !s->structureDirty && s->overflowRefusals >= 2)
	{
		// two builds for the overflow contacts in a row could not be adopted (a burst of contacts the copies could not follow): the
		// structure is built here, in the step, as before round 5
		s->overflowRefusals = 0;
		s->dirtyReason = "overflow builds refused";
		noteGraphChanged(s);
	}
	if (s->stripInc.valid && s->stripInc.overflowUsed > 0 && !s->structureDirty && asyncBuildsOn(s) && (!asyncPending(s) || asyncPendingSearch(s)))
	{
		asyncDrop(s); // (a search over strip widths in flight: it was made without the contact and is a hundred steps from falling due)
		// contacts in the overflow region behind the strips: the steps run sliced until a worker thread's structure that holds them is
		// adopted (solver_internal.h: IncrementalStrips); asked for here when none is on its way (the first step, or after one was dropped)
		int rcAsync = asyncRequest(s, params->solverType, false, true);
		if (rcAsync)
		{
			return rcAsync;
		}
	}
	if (s->persistFailed && s->optPersistRetry > 0 && ++s->persistFailedAge > s->persistRetryAfter)
	{
		// a hand-off timed out a while ago (workgroups not co-resident: something else held part of the GPU): try again
		s->persistFailed = false;
		s->persistFailedAge = 0;
		s->persistRetryAfter = std::min(s->persistRetryAfter * 2, 1 << 20);
		s->stripsRejected = false;
		s->structureDirty = true; // (the structure may have dropped its strips for want of a kernel that could run them)
	}
	if (s->stripRetryPending && s->graphAge >= 32 && !s->structureDirty && asyncBuildsOn(s) && s->stepCounter < s->stripSearchNotBefore)
	{
		s->stripRetryPending = false; // (the last search found nothing better: SolverRest::stripSearchNotBefore)
	}
	if (s->stripRetryPending && s->graphAge >= 32 && !s->structureDirty)
	{
		// the postponed search for a better strip partition (solver_structure.cpp: buildStructure): seven more builds, tens of
		// milliseconds -- in the world chain a worker thread's, on a copy; the steps go on with the strips they have
		if (asyncBuildsOn(s))
		{
			if (!asyncPending(s))
			{
				int rcAsync = asyncRequest(s, params->solverType, true);
				if (rcAsync)
				{
					return rcAsync;
				}
				s->stripRetryPending = false;
			}
		}
		else
		{
			s->structureDirty = true;
			s->stripsRejected = false;
		}
	}
	int rc = buildStructure(s, params->solverType);
	if (rc)
	{
		return rc;
	}
	if (s->jacobiDeferred > 0 && !s->structureDirty && params->solverType == s2amd_solverJacobi)
	{
		// (a world chain's rebuilt structure -- solver_structure.cpp: finish -- has lived for a few steps: the persistent launch's tables
		// now, unless a contact placed without colours has already put the steps on the multi-launch path until the next build)
		if (s->inc.colourFreePlaced)
		{
			s->jacobiDeferred = 0;
		}
		else if (--s->jacobiDeferred == 0)
		{
			if ((rc = buildJacobiBlocks(s)) != 0)
			{
				return rc;
			}
			HIP_TRY(hipStreamSynchronize(s->stream));
			s->layoutGeneration += 1;
		}
	}
	const StepPlan& plan = s->plan;
	if (s->opsGeneration != s->planGeneration)
	{
		bool grew = false;
		if ((rc = s->dOps.ensure(std::max<size_t>(plan.ops.size(), 1) * sizeof(Op), &grew)) != 0)
		{
			return rc;
		}
		if (grew)
		{
			s->layoutGeneration += 1;
		}
		if (!plan.ops.empty())
		{
			HIP_TRY(hipMemcpyAsync(s->dOps.p, plan.ops.data(), plan.ops.size() * sizeof(Op), hipMemcpyHostToDevice, s->stream));
			HIP_TRY(hipStreamSynchronize(s->stream));
		}
		s->opsGeneration = s->planGeneration;
	}

	{
		// strips that only the one-launch kernel may run (solver_structure.cpp: stripsNeedOneLaunch), and a plan or a state
		// that kernel cannot take (another solver family, a hand-off that timed out earlier): colour batches instead
		Executor probe{s, s->stream, plan, isPositionSolver(params->solverType) ? 1 : 0, false};
		const bool soft = params->solverType == s2amd_solverTGS_Soft || params->solverType == s2amd_solverSoftStep || params->solverType == s2amd_solverPGS_Soft;
		const bool multiLaunch = !s->stripsNeedOneLaunch && (s->optStripsAnySolver != 0 || (soft && s->joints.stripCount == 0 && s->leanAValid && s->leanBValid));
		if (s->dStripA.view.groupCount > 0 && !multiLaunch && !probe.oneLaunchPlan())
		{
			s->stripsRejected = true;
			s->structureDirty = true;
			if ((rc = buildStructure(s, params->solverType)) != 0)
			{
				return rc;
			}
		}
	}
	{
		// contacts were placed into the strips' rounds (IncrementalStrips): only the persistent kernels read nothing else; the
		// multi-launch strip path (warm-start slot tables) needs the structure built again
		Executor probe{s, s->stream, plan, isPositionSolver(params->solverType) ? 1 : 0, false};
		int kind, warm;
		// (the op interpreter reads the group tables, which placement does not patch: it takes strips whose placed contacts sit in rounds the
		// build laid out and nothing else -- IncrementalStrips::takeOnly)
		if (s->stripInc.touched && s->dStripA.view.groupCount > 0 && !probe.persistPlan(kind, warm) && !(s->stripInc.takeOnly && probe.genericPlan()))
		{
			s->structureDirty = true;
			s->dirtyReason = "strips with placed contacts off the persistent kernel";
			s->graphAge = 0; // (the graph HAS just changed: no search over strip widths -- seven more builds, 40 ms at base 200 -- in this step)
			if ((rc = buildStructure(s, params->solverType)) != 0)
			{
				return rc;
			}
		}
	}
	{
		// overflow contacts are swept between the launches of the SLICED 512-thread kernel and by nothing else: a plan or a state that
		// kernel cannot take (another solver family, a hand-off that timed out, option "wide" off) gets its structure built now
		Executor probe{s, s->stream, plan, isPositionSolver(params->solverType) ? 1 : 0, false};
		int kind, warm;
		if (probe.slicedPlan() && !(s->dStripA.view.groupCount > 0 && probe.persistPlan(kind, warm) && probe.widePlan(kind, warm)))
		{
			s->structureDirty = true;
			s->dirtyReason = "overflow contacts off the sliced kernel";
			asyncDrop(s);
			if ((rc = buildStructure(s, params->solverType)) != 0)
			{
				return rc;
			}
		}
	}
	Executor q{s, s->stream, plan, isPositionSolver(params->solverType) ? 1 : 0, s->optProfile != 0};
	q.msg = messageEligible(s, params->solverType);
	s->stats.messagePassing = q.msg ? 1 : 0;
	{
		int kind, warm;
		if (s->dStripA.view.groupCount > 0 && (q.persistPlan(kind, warm) || q.genericPlan()) && q.uploadPersistOps() != 0)
		{
			return fail(S2AMD_E_DEVICE, "could not upload the persistent step plan");
		}
	}
	{
		int kind, warm;
		if (q.residentPlan(kind, warm) && q.uploadResidentOps() != 0)
		{
			return fail(S2AMD_E_DEVICE, "could not upload the resident-island step plan");
		}
	}
	s->launchCounter = 0;
	s->sweepEventsUsed = 0;
	s->slicedThisStep = false;

	const bool xpbdEarlyOut = plan.earlyOut;
	const bool writesConstraintIndex = !xpbdEarlyOut && params->solverType != s2amd_solverPGS_NGS_Block;

	// manifold.constraintIndex (pool-order gather index, -1 for skipped slots).  With manifolds recomputed on the device
	// (world chain) the host does not know the point counts: nothing on the device reads the field, so it is brought up to
	// date when somebody downloads the contacts (refreshConstraintIndexOnDevice).
	s->lastStepWroteIndex = writesConstraintIndex;
	if (writesConstraintIndex && s->contactCapacity > 0 && s->pointsKnown)
	{
		if (s->gatherIndexDirty || s->dGatherIndex.bytes < (size_t)s->contactCapacity * sizeof(int))
		{
			std::vector<int> gi((size_t)s->contactCapacity, -1);
			int k = 0;
			for (int i = 0; i < s->contactCapacity; ++i)
			{
				if (s->hContactPoints[i] > 0)
				{
					gi[i] = k++;
				}
			}
			bool grew = false;
			if ((rc = s->dGatherIndex.ensure(gi.size() * sizeof(int), &grew)) != 0)
			{
				return rc;
			}
			if (grew)
			{
				s->layoutGeneration += 1;
			}
			HIP_TRY(hipMemcpyAsync(s->dGatherIndex.p, gi.data(), gi.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
			HIP_TRY(hipStreamSynchronize(s->stream));
			s->gatherIndexDirty = false;
			s->indexInWire = false; // (every upload of contacts marks the gather index dirty: the wire array holds the caller's values again)
		}
	}

	// manifold.constraintIndex only changes with the gather index: a resident world that is stepped again keeps what the last
	// step wrote (1.2 M strided 4-byte stores per step at BASELINE config 5 otherwise)
	const bool indexNow = writesConstraintIndex && s->contactCapacity > 0 && s->pointsKnown && !s->indexInWire;
	auto enqueueAll = [&]() {
		bool indexBranch = false;
		q.gatherIndex = (indexNow && !q.fork) ? (const int*)s->dGatherIndex.p : nullptr;
		if (indexNow && q.fork)
		{
			// touches only manifold.constraintIndex, which no solver kernel reads: a parallel branch that joins at the end
			int n = s->contactCapacity;
			hipStream_t where = q.branch(0, 0);
			writeConstraintIndexKernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, where>>>((s2amdContact*)s->dContacts.p, n,
																									 (const int*)s->dGatherIndex.p);
			q.count();
			indexBranch = true;
		}
		q.run();
		if (indexBranch)
		{
			q.join(0, 0);
		}
	};

	if (q.selfContainedStrips() && ++s->selfStepsSinceReset >= (1 << 20))
	{
		// the self-contained strip kernel's commit counter runs on from step to step: back to zero long before it could wrap
		HIP_TRY(hipMemsetAsync(s->persist.state, 0, sizeof(unsigned int), s->stream));
		s->selfStepsSinceReset = 0;
	}
	bool useGraph = s->optGraph != 0 && !q.profile;
	q.fork = useGraph && s->optFork != 0;
	HIP_TRY(hipEventRecord(s->evBegin, s->stream));
	if (useGraph)
	{
		uint64_t key = 1469598103934665603ull;
		key = fnv(key, params, sizeof(*params));
		uint64_t gens[4] = {s->layoutGeneration, s->structureGeneration, s->planGeneration, (uint64_t)((q.msg ? 1 : 0) | (s->optBodyWarm ? 2 : 0) | (s->optStripLean ? 4 : 0) | (s->optPersist ? 8 : 0) | (s->optFork ? 16 : 0) | (s->persistFailed ? 32 : 0) | ((s->persistValid && s->persist.allTwoPoints) ? 64 : 0) | (s->pointsKnown ? 128 : 0) | (s->optPairLanes ? 256 : 0) | (s->optWide ? 512 : 0) | (s->optGeneric ? 1024 : 0) | (s->genericValid ? 2048 : 0) | (indexNow ? 4096 : 0) | (s->optSelfContained ? 8192 : 0) | ((s->residentAllTwoPoints && s->pointsKnown) ? 16384 : 0) | (s->optStageJoints ? 32768 : 0) | (s->optWideBodyWarm ? 65536 : 0) | (s->optSelfContainedStrips ? 131072 : 0) | (s->stage4.shapes != nullptr ? 262144 : 0))};
		key = fnv(key, gens, sizeof(gens));
		int sizes[3] = {s->bodyCapacity, s->contactCapacity, s->jointCapacity};
		key = fnv(key, sizes, sizeof(sizes));
		// (the world step's stage 4 rides in the captured epilogue launch with its arguments baked in: a shape array that moved or grew
		// under the same body / contact / joint capacities -- a shape added to an existing body -- is another graph)
		const uint64_t stage4Words[4] = {(uint64_t)(uintptr_t)s->stage4.shapes, (uint64_t)s->stage4.shapeCapacity, (uint64_t)(uintptr_t)s->stage4.origins,
										 (uint64_t)(uintptr_t)s->stage4.summary};
		key = fnv(key, stage4Words, sizeof(stage4Words));
		if (key == 0)
		{
			key = 1;
		}
		if ((key != s->graphKey || s->graphExec == nullptr) && (key != s->graphKeySeen || s->graphKeySeenLaunches < s->optGraphMinLaunches))
		{
			// A launch sequence seen for the first time is enqueued directly: capture + instantiate cost more than the
			// launches themselves and only pay off when the same sequence comes back (a changing contact graph never
			// brings one back).  And a step of only a few launches stays direct for good: replaying a graph of one to three
			// kernels measured 6-8 us SLOWER per step than enqueueing them (r4, LargePyramid base-200: 0.159 against 0.151 ms).
			s->graphKeySeen = key;
			q.fork = false; // the parallel branches only exist inside a captured graph
			enqueueAll();
			s->graphKeySeenLaunches = s->launchCounter;
		}
		else if (key != s->graphKey || s->graphExec == nullptr)
		{
			destroyGraph(s);
			HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
			enqueueAll();
			hipError_t ce = hipStreamEndCapture(s->stream, &s->graph);
			if (ce != hipSuccess)
			{
				s->graph = nullptr;
				return fail(S2AMD_E_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
			}
			HIP_TRY(hipGraphInstantiate(&s->graphExec, s->graph, nullptr, nullptr, 0));
			s->graphKey = key;
			s->graphLaunches = s->launchCounter;
			s->graphCarriesStage4 = s->stage4Carried;
		}
		else
		{
			s->launchCounter = s->graphLaunches;
			s->stats.graphReplayed = 1;
			s->stage4Carried = s->graphCarriesStage4; // (the captured epilogue launch holds the world step's stage-4 blocks, or not)
		}
		if (key == s->graphKey && s->graphExec != nullptr)
		{
			HIP_TRY(hipGraphLaunch(s->graphExec, s->stream));
		}
	}
	else
	{
		enqueueAll();
	}
	HIP_TRY(hipEventRecord(s->evEnd, s->stream));
	HIP_TRY(hipGetLastError());
	s->indexInWire = s->indexInWire || indexNow;
	const bool async = s->optAsync != 0 && !q.profile;
	if (!async)
	{
		HIP_TRY(hipStreamSynchronize(s->stream));
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, s->evBegin, s->evEnd));
		s->stats.deviceMs = ms;
	}
	s->stats.constraintCount = s->activeContacts; // (cv.count also holds the potential constraints whose manifold has no points now)
	s->stats.jointCount = s->jv.count;
	s->stats.contactColors = (int)s->contacts.colorOffsets.size() - 1;
	s->stats.jointColors = (int)s->joints.colorOffsets.size() - 1;
	s->stats.solveSweeps = plan.solveSweeps;
	s->stats.kernelLaunches = s->launchCounter;
	s->stats.groupCount = s->dGroups.view.groupCount + s->dResident.view.groupCount;
	s->stats.stripCount = s->dStripA.view.groupCount;
	s->stats.seamCount = s->dStripB.view.groupCount;
	{
		int kind, warm;
		s->stats.persistent = ((s->dStripA.view.groupCount > 0 && (q.persistPlan(kind, warm) || q.genericPlan())) || q.jacobiPlan()) ? 1 : 0;
	}
	s->stats.persistFallbacks = s->persistFallbacks;
	s->stats.asyncBuildsRequested = s->asyncRequested, s->stats.asyncBuildsAdopted = s->asyncAdopted, s->stats.asyncWaitMs = s->asyncWaitMs;
	s->stats.nearHandoffTimeouts = s->nearHandoffTimeouts;
	{
		int kind, warm;
		s->slicedThisStep = s->dStripA.view.groupCount > 0 && q.slicedPlan() && q.persistPlan(kind, warm) && q.widePlan(kind, warm) && !q.selfContainedStrips();
		s->overflowKernelThisStep = s->slicedThisStep && q.overflowKernelPlan();
		s->slicedSteps += s->slicedThisStep ? 1 : 0;
		s->stats.overflowContacts = s->stripInc.valid ? s->stripInc.overflowUsed : 0;
		s->stats.slicedStep = s->slicedThisStep ? (s->overflowKernelThisStep ? 2 : 1) : 0; // 1: one launch per sweep; 2: one launch, the overflow workgroup in it
		s->stats.slicedSteps = (int32_t)s->slicedSteps;
	}
	s->stats.bodiesAdopted = (int32_t)s->stripInc.adopted, s->stats.seamBodiesAdded = (int32_t)s->stripInc.seamBodiesAdded, s->stats.roundsOpened = (int32_t)s->stripInc.roundsOpened;
	s->stats.structureBuilds = (int32_t)s->structureGeneration;
	s->stats.placedContacts = (int32_t)s->placedTotal;
	{
		int kind = -1, warm = -1;
		const bool persistent = s->stats.persistent && q.persistPlan(kind, warm);
		const bool wide = persistent && q.widePlan(kind, warm);
		s->stats.pairLanes = wide ? 2 : (persistent && s->persist.pairLanes && s->optPairLanes) ? 1 : (s->stats.persistent && !persistent) ? 3 : 0;
	}
	s->stats.potentialConstraints = (int32_t)(s->contacts.order.size() - (size_t)s->slackPositions);
	if (!async && s->hostError && *s->hostError != 0u)
	{
		// The persistent kernel's workgroups were not all resident (something else occupies the GPU).  Its epilogue saw
		// the flag and left the wire arrays untouched, so the step is simply repeated on the multi-launch strip path,
		// which this solver keeps from now on.
		int rcReset = resetPersistState(s, s->stream);
		if (rcReset)
		{
			return rcReset;
		}
		if (s->nearHandoffNow != 0 && s->stats.persistent)
		{
			// ... unless the same-XCD hand-off path was in use: its stores are only promised to be seen inside one L2.  The step is
			// tried again on the same kernel with agent-scope stores everywhere, which this solver keeps from now on.
			s->nearHandoffNow = 0;
			s->nearHandoffTimeouts += 1;
			s->layoutGeneration += 1; // (a captured step graph has the old launch parameters)
			s->stepCounter -= 1;	  // (the same step again: the clock of the deferred adoptions must not run on time-outs)
			return doStep(s, params);
		}
		if (s->overflowKernelThisStep && !s->overflowKernelFailed)
		{
			// ... or the launch carried the overflow workgroup (wide_kernel.hip: wideOverflowWorker) and a hand-off with IT may be what
			// timed out: the step again sliced, which this solver keeps for its overflow steps
			s->overflowKernelFailed = true;
			s->layoutGeneration += 1;
			s->persistFallbacks += 1;
			s->stepCounter -= 1;
			return doStep(s, params);
		}
		s->persistFailed = true;
		s->persistFailedAge = 0;
		s->persistFallbacks += 1;
		s->stepCounter -= 1;
		return doStep(s, params);
	}
	s->graphAge += 1;
	if (q.profile)
	{
		float total = 0.0f;
		for (size_t i = 0; i + 1 < s->sweepEventsUsed; i += 2)
		{
			float t = 0.0f;
			if (hipEventElapsedTime(&t, s->sweepEvents[i], s->sweepEvents[i + 1]) == hipSuccess)
			{
				total += t;
			}
		}
		// calibrate: an empty event pair on the same stream measures the bracket's own cost
		float empty = 0.0f;
		int pairs = 0;
		if (s->sweepEvents.size() >= 2)
		{
			for (int r = 0; r < 32; ++r)
			{
				(void)hipEventRecord(s->sweepEvents[0], s->stream);
				(void)hipEventRecord(s->sweepEvents[1], s->stream);
				(void)hipStreamSynchronize(s->stream);
				float t = 0.0f;
				if (hipEventElapsedTime(&t, s->sweepEvents[0], s->sweepEvents[1]) == hipSuccess)
				{
					empty += t;
					pairs += 1;
				}
			}
		}
		s->stats.solveKernelMs = total;
		s->stats.solveLaunches = (int)(s->sweepEventsUsed / 2);
		s->stats.eventPairOverheadMs = pairs > 0 ? empty / pairs : 0.0f;
	}
	return S2AMD_OK;
}

int doDownload(s2amdSolver* s, s2amdBody* bodies, int nb, s2amdContact* contacts, int nc, s2amdJoint* joints, int nj)
{
	if (!s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident to download");
	}
	if (nb < s->bodyCapacity || nc < s->contactCapacity || nj < s->jointCapacity)
	{
		return fail(S2AMD_E_CAPACITY, "output arrays smaller than the resident world");
	}
	HIP_TRY(hipSetDevice(s->device));
	if (contacts && !s->pointsKnown && s->lastStepWroteIndex)
	{
		int rc = refreshConstraintIndexOnDevice(s);
		if (rc)
		{
			return rc;
		}
	}
	if (s->bodyCapacity > 0 && bodies)
	{
		HIP_TRY(hipMemcpyAsync(bodies, s->dBodies.p, (size_t)s->bodyCapacity * sizeof(s2amdBody), hipMemcpyDeviceToHost, s->stream));
	}
	if (s->contactCapacity > 0 && contacts)
	{
		HIP_TRY(hipMemcpyAsync(contacts, s->dContacts.p, (size_t)s->contactCapacity * sizeof(s2amdContact), hipMemcpyDeviceToHost, s->stream));
	}
	if (s->jointCapacity > 0 && joints)
	{
		HIP_TRY(hipMemcpyAsync(joints, s->dJoints.p, (size_t)s->jointCapacity * sizeof(s2amdJoint), hipMemcpyDeviceToHost, s->stream));
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

