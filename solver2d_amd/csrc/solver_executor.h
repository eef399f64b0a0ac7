// The launch sequence of one step: global colour batches, LDS groups, strips (lean launches or the persistent
// step kernel).  Header-only: used by solver_step.cpp (doStep) and solver.cpp (s2amd_measure_dominant).
#pragma once

#include "solver_internal.h"


// ------------------------------------------------------------------------------------------------
// execution of a plan
// ------------------------------------------------------------------------------------------------
struct Executor
{
	s2amdSolver* s;
	hipStream_t st;
	const StepPlan& p;
	int posSolver;
	bool profile;
	bool msg = false; // message-passing accessor for the global part
	bool fork = false; // graph capture: independent kernels go to side streams (parallel graph branches)
	const int* gatherIndex = nullptr; // manifold.constraintIndex rides in the unpack launch

	hipStream_t branch(int i, int forkEvent)
	{
		if (!fork)
		{
			return st;
		}
		(void)hipEventRecord(s->evFork[forkEvent], st);
		(void)hipStreamWaitEvent(s->side[i], s->evFork[forkEvent], 0);
		return s->side[i];
	}
	void join(int i, int joinEvent)
	{
		if (fork)
		{
			(void)hipEventRecord(s->evJoin[joinEvent], s->side[i]);
			(void)hipStreamWaitEvent(st, s->evJoin[joinEvent], 0);
		}
	}

	s2amdContact* wireContacts() const { return (s2amdContact*)s->dContacts.p; }
	s2amdBody* wireBodies() const { return (s2amdBody*)s->dBodies.p; }
	s2amdJoint* wireJoints() const { return (s2amdJoint*)s->dJoints.p; }
	const Op* deviceOps() const { return (const Op*)s->dOps.p; }

	void count(int n = 1) { s->launchCounter += n; }

	void recordEvent()
	{
		if (s->sweepEventsUsed == s->sweepEvents.size())
		{
			hipEvent_t e;
			if (hipEventCreate(&e) != hipSuccess)
			{
				return;
			}
			s->sweepEvents.push_back(e);
		}
		(void)hipEventRecord(s->sweepEvents[s->sweepEventsUsed++], st);
	}

	static bool isSolveSweep(int code)
	{
		return code == OP_SOLVE_SOFT || code == OP_SOLVE_RIGID || code == OP_SOLVE_STICKY || code == OP_SOLVE_NGS || code == OP_XPBD_POS ||
			   code == OP_XPBD_VEL || code == OP_BLOCK_VEL || code == OP_BLOCK_POS;
	}

	void launchContactBatch(const Op& o, int b, int e)
	{
		if (msg)
		{
			switch (o.code)
			{
				case OP_WARM:
					launchWarmStartContactsMsg(st, o.kind, s->cv, s->msg, b, e);
					return;
				case OP_SOLVE_SOFT:
					launchSolveContactsSoftMsg(st, o.kind, s->cv, s->msg, b, e, o.inv_h, o.useBias);
					return;
				case OP_SOLVE_RIGID:
					launchSolveContactsRigidMsg(st, o.kind, s->cv, s->msg, b, e, o.inv_h);
					return;
				case OP_SOLVE_STICKY:
					launchSolveContactsStickyMsg(st, s->cv, s->msg, wireContacts(), b, e, o.inv_h, o.useBias);
					return;
				default:
					return; // message mode is only enabled for plans made of the ops above
			}
		}
		switch (o.code)
		{
			case OP_WARM:
				launchWarmStartContacts(st, o.kind, s->cv, s->bv, b, e);
				break;
			case OP_SOLVE_SOFT:
				launchSolveContactsSoft(st, o.kind, s->cv, s->bv, b, e, o.inv_h, o.useBias);
				break;
			case OP_SOLVE_RIGID:
				launchSolveContactsRigid(st, o.kind, s->cv, s->bv, b, e, o.inv_h);
				break;
			case OP_SOLVE_STICKY:
				launchSolveContactsSticky(st, s->cv, s->bv, wireContacts(), b, e, o.inv_h, o.useBias);
				break;
			case OP_SOLVE_NGS:
				launchSolveContactsNGS(st, s->cv, s->bv, b, e);
				break;
			case OP_XPBD_POS:
				launchXpbdContactPositions(st, s->cv, s->bv, b, e, o.h);
				break;
			case OP_XPBD_VEL:
				launchXpbdContactVelocities(st, s->cv, s->bv, b, e, o.h);
				break;
			case OP_BLOCK_VEL:
				launchBlockSolveVelocity(st, s->cv, s->bv, b, e);
				break;
			case OP_BLOCK_POS:
				launchBlockSolvePosition(st, s->cv, s->bv, b, e);
				break;
		}
	}

	// one op of the plan over the GLOBAL part (bodies in HBM): one launch per colour batch
	void runGlobalOp(int index)
	{
		const Op& o = p.ops[(size_t)index];
		const SweepSet& cs = s->contacts;
		const SweepSet& js = s->joints;
		const bool bodies = s->looseBodies > 0;
		switch (o.code)
		{
			case OP_INTEGRATE_VEL:
				if (bodies)
				{
					if (msg)
					{
						launchIntegrateVelocitiesMsg(st, s->bv, s->msg);
					}
					else
					{
						launchIntegrateVelocities(st, s->bv);
					}
					count();
				}
				return;
			case OP_INTEGRATE_POS:
				if (bodies)
				{
					if (msg)
					{
						launchIntegratePositionsMsg(st, s->bv, s->msg, o.h);
					}
					else
					{
						launchIntegratePositions(st, s->bv, o.h);
					}
					count();
				}
				return;
			case OP_FINALIZE:
				if (bodies)
				{
					if (msg)
					{
						launchFinalizePositionsMsg(st, s->bv, s->msg, o.flag);
					}
					else
					{
						launchFinalizePositions(st, s->bv, o.flag);
					}
					count();
				}
				return;
			case OP_XPBD_INTEGRATE:
				if (bodies)
				{
					launchXpbdIntegrate(st, s->bv, o.h);
					count();
				}
				return;
			case OP_XPBD_PROJECT:
				if (bodies)
				{
					launchXpbdProject(st, s->bv, o.inv_h);
					count();
				}
				return;
			case OP_JACOBI_APPLY:
				launchJacobiApply(st, s->bv, s->cv, (const int2*)s->dAdjOffsets.p, (const int*)s->dAdjList.p, (const int*)s->dAdjHeavy.p, s->adjHeavyCapacity);
				count();
				return;
			case OP_JOINT_SWEEP:
			{
				int nb = (int)js.batchOffsets.size() - 1;
				for (int bi = 0; bi < nb; ++bi)
				{
					int b = js.batchOffsets[(size_t)bi], e = js.batchOffsets[(size_t)bi + 1];
					if (e <= b)
					{
						continue;
					}
					if (js.hasTail && bi == nb - 1)
					{
						launchGroupKernel(st, s->cv, s->jv, s->bv, s->dJointTail.view, deviceOps() + index, 1, p.sc, wireContacts(),
										  s->dJointTail.maxBodies, 0);
					}
					else
					{
						launchSolveJoints(st, o.kind, s->jv, s->bv, b, e, p.sc, o.h, o.inv_h, o.useBias);
					}
					count();
				}
				return;
			}
			default:
				break;
		}
		// contact sweeps
		if (o.code == OP_SOLVE_SOFT && o.kind == SOFT_JACOBI)
		{
			// the Jacobi pass writes per-constraint deltas, never a body: one launch for all colours
			if (cs.globalCount > 0)
			{
				if (profile)
				{
					recordEvent();
				}
				launchSolveContactsSoft(st, o.kind, s->cv, s->bv, 0, cs.globalCount, o.inv_h, o.useBias);
				if (profile)
				{
					recordEvent();
				}
				count();
			}
			return;
		}
		int nb = (int)cs.batchOffsets.size() - 1;
		for (int bi = 0; bi < nb; ++bi)
		{
			int b = cs.batchOffsets[(size_t)bi], e = cs.batchOffsets[(size_t)bi + 1];
			if (e <= b || emptyBatch(bi))
			{
				continue;
			}
			const bool timed = profile && isSolveSweep(o.code);
			if (timed)
			{
				recordEvent();
			}
			if (cs.hasTail && bi == nb - 1)
			{
				launchGroupKernel(st, s->cv, s->jv, s->bv, s->dContactTail.view, deviceOps() + index, 1, p.sc, wireContacts(),
								  s->dContactTail.maxBodies, 0);
			}
			else
			{
				launchContactBatch(o, b, e);
			}
			if (timed)
			{
				recordEvent();
			}
			count();
		}
	}

	// a colour batch of the global part that holds only free positions (the slack layout's spare colours; a batch whose contacts
	// have all gone): nothing to launch.  The first contact placed into it re-captures the step graph (solver_incremental.cpp).
	bool emptyBatch(int bi) const
	{
		const IncrementalGlobal& inc = s->inc;
		return inc.valid && bi < inc.parallelBatches && (int)inc.freePositions[(size_t)bi].size() == inc.batchEnd[(size_t)bi] - inc.batchBegin[(size_t)bi];
	}

	// constraints (not positions) in the global contact part
	bool anyGlobalContacts() const
	{
		const IncrementalGlobal& inc = s->inc;
		if (!inc.valid)
		{
			return s->contacts.globalCount > 0;
		}
		const SweepSet& cs = s->contacts;
		const int nb = (int)cs.batchOffsets.size() - 1;
		for (int bi = 0; bi < nb; ++bi)
		{
			if (cs.batchOffsets[(size_t)bi + 1] > cs.batchOffsets[(size_t)bi] && !emptyBatch(bi))
			{
				return true;
			}
		}
		return false;
	}

	static bool isBodyOp(int code)
	{
		return code == OP_INTEGRATE_VEL || code == OP_INTEGRATE_POS || code == OP_FINALIZE || code == OP_XPBD_INTEGRATE || code == OP_XPBD_PROJECT;
	}

	void launchStripGroups(const DeviceGroupTable& t, int first, int n, bool timed)
	{
		if (timed)
		{
			recordEvent();
		}
		launchStripKernel(st, s->cv, s->jv, s->bv, t.view, deviceOps() + first, n, p.sc, wireContacts(), t.maxBodies, p.usesDq0 ? 1 : 0);
		if (timed)
		{
			recordEvent();
		}
		count();
	}

	static bool leanSoftKind(const Op& o) { return o.code == OP_SOLVE_SOFT && (o.kind == SOFT_TGS || o.kind == SOFT_PGS || o.kind == SOFT_FIXED); }

	bool sweepsNothing(const Op& o) const
	{
		if (isBodyOp(o.code))
		{
			return false;
		}
		return (o.code == OP_JOINT_SWEEP ? s->joints.stripCount : s->contacts.stripCount) == 0;
	}

	// Can ops [first, sweep) ride in front of the soft sweep `sweep` inside ONE lean strip launch?  Allowed, in
	// this order: integrate positions, integrate velocities, contact warm start (body-centric).
	bool leanSegment(int first, int sweep, StripOps& out, int& warm) const
	{
		if (!s->leanAValid || !leanSoftKind(p.ops[(size_t)sweep]))
		{
			return false;
		}
		out = StripOps{};
		warm = -1;
		int stage = 0;
		for (int i = first; i < sweep; ++i)
		{
			const Op& o = p.ops[(size_t)i];
			if (o.code == OP_INTEGRATE_POS && stage < 1)
			{
				out.integratePos = 1, out.posH = o.h, stage = 1;
			}
			else if (o.code == OP_INTEGRATE_VEL && stage < 2)
			{
				out.integrateVel = 1, stage = 2;
			}
			else if (o.code == OP_WARM && stage < 3 && s->optBodyWarm && (o.kind == WARM_CURRENT || o.kind == WARM_FIXED))
			{
				warm = o.kind, stage = 3;
			}
			else if (!sweepsNothing(o))
			{
				return false;
			}
		}
		const Op& w = p.ops[(size_t)sweep];
		out.sweep = 1, out.useBias = w.useBias, out.inv_h = w.inv_h;
		return true;
	}

	// Can the whole plan run as ONE persistent launch over the strips (strip_kernel.hip: stripStepKernel)?
	bool persistPlan(int& kind, int& warm) const
	{
		if (!s->persistValid || s->persistFailed || p.ops.size() > 128 || p.solveSweeps > 63) // one hand-off epoch per sweep, 64 per step
		{
			return false;
		}
		kind = -1, warm = -1;
		for (const Op& o : p.ops)
		{
			if (o.code == OP_INTEGRATE_VEL || o.code == OP_INTEGRATE_POS || o.code == OP_FINALIZE)
			{
				continue;
			}
			if (o.code == OP_JOINT_SWEEP && s->joints.stripCount == 0)
			{
				continue;
			}
			if (o.code == OP_WARM && (o.kind == WARM_CURRENT || o.kind == WARM_FIXED) && (warm < 0 || warm == o.kind))
			{
				warm = o.kind;
				continue;
			}
			if (leanSoftKind(o) && (kind < 0 || kind == o.kind))
			{
				kind = o.kind;
				continue;
			}
			return false;
		}
		if (kind < 0)
		{
			return false;
		}
		if (warm < 0)
		{
			warm = kind == SOFT_FIXED ? WARM_FIXED : WARM_CURRENT;
		}
		if (widePlan(kind, warm))
		{
			return true; // (wide_kernel.hip keeps the seam constraints in registers: its own LDS budget, Executor::wideFits)
		}
		if (s->persist.wideOnly)
		{
			return false; // (rounds opened since the build that only that kernel's tables and budgets know: IncrementalStrips)
		}
		const bool narrow = kind == SOFT_TGS && warm == WARM_CURRENT;
		const int records = (narrow ? s->persist.ldsRecords : s->persistRecordsWide) + 2 * (int)p.ops.size();
		return records <= (160 * 1024) / 16;
	}

	// Can the whole plan run as ONE launch of the op interpreter over the strips (generic_kernel.hip: genericStepKernel)?  Every
	// Gauss-Seidel family and joints; not s2Solve_Jacobi (its body-centric apply needs the incidence lists).
	bool genericPlan() const
	{
		if (!s->genericValid || s->persistFailed || p.ops.size() > 128)
		{
			return false;
		}
		for (const Op& o : p.ops)
		{
			if (o.code == OP_JACOBI_APPLY || (o.code == OP_SOLVE_SOFT && o.kind == SOFT_JACOBI))
			{
				return false;
			}
		}
		return genericStepLds(s->genericBodies, s->genericSeamBodies, s->genericExports, (int)p.ops.size(), p.usesDq0 ? 1 : 0) <= 160 * 1024;
	}

	// one launch for the strips, whichever kernel: the register-resident soft kernels where they apply, else the op interpreter
	bool oneLaunchPlan() const
	{
		int kind, warm;
		return persistPlan(kind, warm) || genericPlan();
	}

	// ... and of the persistent kernels, the 512-thread one (wide_kernel.hip: wideStepKernel)?  TGS_Soft with the current-anchor
	// warm start on a partition with at most six interior colour batches per strip and two per seam.
	bool widePlan(int kind, int warm) const
	{
		const bool current = (kind == SOFT_TGS || kind == SOFT_PGS) && warm == WARM_CURRENT;
		const bool fixed = kind == SOFT_FIXED && warm == WARM_FIXED; // s2Solve_SoftStep: s2WarmStartContacts_Fixed
		return s->optWide && (current || fixed) && wideFits(false, false, kind);
	}

	// the kernel variant for this partition with these two features: does its dynamic LDS fit?
	bool wideFits(bool selfContained, bool bodyWarm, int kind = SOFT_TGS) const
	{
		const int extra = wideExtraRecords(s->persist, selfContained ? 1 : 0, bodyWarm ? 1 : 0, kind);
		return extra >= 0 && s->persist.bodyRecords + 3 + extra + 2 * s->persistOpCount <= (160 * 1024) / 16;
	}

	// ... as the step's ONLY launch (wide_kernel.hip: S2_WIDE_SELF): the strips are all there is -- every movable body is owned by one,
	// every constraint is one of theirs, no joints, no other group -- and manifold.constraintIndex is already in the wire array.  The
	// kernel then prepares its constraints from the wire contacts, stages its bodies from the wire bodies and writes both back.
	bool selfContainedStrips() const
	{
		int kind, warm;
		return s->optSelfContainedStrips && !slicedPlan() && s->dStripA.view.groupCount > 0 && persistPlan(kind, warm) && kind == SOFT_TGS && widePlan(kind, warm) && wideFits(true, false) && gatherIndex == nullptr && !msg &&
			   wireBodies() != nullptr && s->dGroups.view.groupCount == 0 && s->dResident.view.groupCount == 0 && s->looseBodies == 0 && !anyGlobalContacts() &&
			   s->joints.globalCount == 0 && s->jv.count == 0 && s->cv.count == s->persistK1 - s->persistK0 && p.prepContacts == PREP_SOFT && p.storeKind == STORE_PLAIN;
	}

	// s2WarmStartContacts body-centric inside that kernel: a variant without parked rounds whose term table fits LDS
	bool wideBodyWarm(bool selfContained) const
	{
		return s->optWideBodyWarm != 0 && wideBodyWarmVariant(s->persist) != 0 && wideFits(selfContained, true);
	}

	// Can the plan run on the resident-island kernel (strip_kernel.hip: islandStepKernel)?  The soft contact drivers: body
	// stages, a current- or fixed-anchor warm start, one soft sweep kind; joint sweeps have nothing to do there (the groups
	// it takes are contact-only).
	bool residentPlan(int& kind, int& warm) const
	{
		kind = -1, warm = -1;
		if (s->residentView.groupCount <= 0 || p.ops.size() > 128 || p.prepContacts != PREP_SOFT || p.storeKind != STORE_PLAIN)
		{
			return false;
		}
		// The island kernels write their impulses straight into the wire contacts.  In a step that also holds a persistent strip
		// launch -- which may lose a hand-off, whereupon the step is repeated from untouched wire arrays -- the groups run on the group
		// interpreter instead (SoA arrays; the epilogue, which stands down after a failure, carries the results over).  They must run
		// BEFORE the strips either way: a strip may own a kinematic body that an island reads, and writes it back at its end.
		if (s->dStripA.view.groupCount > 0 && oneLaunchPlan())
		{
			return false;
		}
		for (const Op& o : p.ops)
		{
			if (o.code == OP_INTEGRATE_VEL || o.code == OP_INTEGRATE_POS || o.code == OP_FINALIZE || o.code == OP_JOINT_SWEEP)
			{
				continue;
			}
			if (o.code == OP_WARM && (o.kind == WARM_CURRENT || o.kind == WARM_FIXED) && (warm < 0 || warm == o.kind))
			{
				warm = o.kind;
				continue;
			}
			if (leanSoftKind(o) && (kind < 0 || kind == o.kind))
			{
				kind = o.kind;
				continue;
			}
			return false;
		}
		if (kind < 0)
		{
			return false;
		}
		if (warm < 0)
		{
			warm = kind == SOFT_FIXED ? WARM_FIXED : WARM_CURRENT;
		}
		return true;
	}

	int uploadResidentOps()
	{
		if (s->residentOpsGeneration == s->planGeneration)
		{
			return 0;
		}
		std::vector<Op> kept;
		for (const Op& o : p.ops)
		{
			if (o.code != OP_JOINT_SWEEP)
			{
				kept.push_back(o);
			}
		}
		bool grew = false;
		int rc = s->dResidentOps.ensure(std::max<size_t>(kept.size(), 1) * sizeof(Op), &grew);
		if (rc)
		{
			return rc;
		}
		if (grew)
		{
			s->layoutGeneration += 1;
		}
		if (hipMemcpyAsync(s->dResidentOps.p, kept.data(), kept.size() * sizeof(Op), hipMemcpyHostToDevice, st) != hipSuccess ||
			hipStreamSynchronize(st) != hipSuccess)
		{
			return S2AMD_E_DEVICE;
		}
		s->residentOpCount = (int)kept.size();
		s->residentOpsGeneration = s->planGeneration;
		return 0;
	}

	// the groups that were laid out for the resident-island kernel: on it when the plan allows, else through the generic
	// group interpreter (the table is an ordinary group table too)
	// the resident islands on wide_kernel.hip: wideIslandKernel (TGS_Soft with the current-anchor warm start)
	bool wideIslandPlan() const
	{
		int kind, warm;
		return residentPlan(kind, warm) && s->optWide && kind == SOFT_TGS && warm == WARM_CURRENT &&
			   s->residentView.ldsRecords + 2 + wideIslandLocalRecords(s->residentRounds) + 2 * s->residentOpCount <= (160 * 1024) / 16;
	}

	// ... and nothing else in the world: every body is owned by a resident island, every constraint is one of theirs, no joints, and
	// manifold.constraintIndex is already in the wire array.  Then that kernel stages its bodies from the wire records and writes
	// them back itself -- the step is ONE launch (BASELINE config 5: 0.45 -> 0.39 ms per step without the two body passes).
	bool selfContainedIslands() const
	{
		return wideIslandPlan() && s->optSelfContained && gatherIndex == nullptr && !msg && wireBodies() != nullptr && s->dGroups.view.groupCount == 0 &&
			   s->dStripA.view.groupCount == 0 && s->looseBodies == 0 && !anyGlobalContacts() && s->joints.globalCount == 0 && s->jv.count == 0 &&
			   s->cv.count == s->residentK1 - s->residentK0;
	}

	void runResidentGroups(bool selfContained = false)
	{
		if (s->dResident.view.groupCount <= 0)
		{
			return;
		}
		const unsigned int* stepFailed = nullptr; // (see residentPlan: these kernels never share a step with a launch that can fail)
		int kind, warm;
		if (residentPlan(kind, warm))
		{
			float4 coef[2];
			for (int i = 0; i < 2; ++i)
			{
				coef[i] = make_float4(p.sc.softCoef[i][0], p.sc.softCoef[i][1], p.sc.softCoef[i][2], 0.0f);
			}
			// (the LDS budget of the resident groups leaves room for the two coefficient records: buildResidentTables)
			if (wideIslandPlan())
			{
				launchWideIsland(st, s->cv, s->bv, s->residentView, coef, (const Op*)s->dResidentOps.p, s->residentOpCount, s->residentRounds, wireContacts(),
								 wireBodies(), (const uint32_t*)s->dBodyFlags.p, p.sc.warmStart, p.sc, p.unpackH, selfContained ? 1 : 0, stepFailed,
								 (s->residentAllTwoPoints && s->pointsKnown) ? 1 : 0); // (manifolds recomputed on the device: any point count)
			}
			else
			{
				launchIslandStep(st, kind, warm, s->cv, s->bv, s->residentView, coef, (const Op*)s->dResidentOps.p, s->residentOpCount, s->residentRounds,
								 wireContacts(), wireBodies(), (const uint32_t*)s->dBodyFlags.p, p.sc.warmStart, stepFailed);
			}
		}
		else
		{
			launchGroupKernel(st, s->cv, s->jv, s->bv, s->dResident.view, deviceOps(), (int)p.ops.size(), p.sc, wireContacts(), s->dResident.maxBodies,
							  p.usesDq0 ? 1 : 0);
		}
		count();
	}

	// the plan without the sweeps that have nothing to sweep in the strips (joint sweeps of a contact-only island)
	int uploadPersistOps()
	{
		if (s->persistOpsGeneration == s->planGeneration && s->persistOpsStructure == s->structureGeneration)
		{
			return 0;
		}
		std::vector<Op> kept;
		for (const Op& o : p.ops)
		{
			if (!sweepsNothing(o))
			{
				kept.push_back(o);
			}
		}
		bool grew = false;
		int rc = s->dPersistOps.ensure(std::max<size_t>(kept.size(), 1) * sizeof(Op), &grew);
		if (rc)
		{
			return rc;
		}
		if (grew)
		{
			s->layoutGeneration += 1;
		}
		if (hipMemcpyAsync(s->dPersistOps.p, kept.data(), kept.size() * sizeof(Op), hipMemcpyHostToDevice, st) != hipSuccess ||
			hipStreamSynchronize(st) != hipSuccess)
		{
			return S2AMD_E_DEVICE;
		}
		s->persistOpCount = (int)kept.size();
		s->persistOpsGeneration = s->planGeneration;
		s->persistOpsStructure = s->structureGeneration;
		return 0;
	}

	// the ops the persistent kernels take, as uploadPersistOps lays them out on the device
	std::vector<Op> keptOps() const
	{
		std::vector<Op> kept;
		for (const Op& o : p.ops)
		{
			if (!sweepsNothing(o))
			{
				kept.push_back(o);
			}
		}
		return kept;
	}

	// contacts in the overflow region behind the strips (solver_internal.h: IncrementalStrips): the step runs sliced
	bool slicedPlan() const { return s->stripInc.valid && s->stripInc.overflowUsed > 0; }

	// The persistent step SLICED: one launch of wideStepKernel per sweep, with that sweep's ops (the body stages before it ride along) --
	// the kernel stages its bodies and records at every launch and writes bodies and impulses back at its end, so the state between
	// two launches is complete in the SoA arrays --, and behind each launch the overflow contacts' turn of the same sweep, one after
	// the other on the bodies in HBM (the colour-batch kernels of the global part; their positions come last in the sweep order).
	// Hand-off tags restart in every launch: each launch clears the buffers it read at its end (PersistView::clearOwn).
	void runPersistentSliced(int kind, int warm)
	{
		const std::vector<Op> kept = keptOps();
		const int n = (int)kept.size();
		PersistView pv = s->persist;
		pv.nearHandoff = s->nearHandoffNow;
		for (int i = 0; i < 2; ++i)
		{
			pv.softCoef[i] = make_float4(p.sc.softCoef[i][0], p.sc.softCoef[i][1], p.sc.softCoef[i][2], 0.0f);
		}
		if (!(kind == SOFT_TGS && warm == WARM_CURRENT))
		{
			pv.ldsRecords = s->persistRecordsWide;
		}
		pv.bodyWarm = 0;
		pv.clearOwn = 1; // (every launch leaves the buffers it read at zero tags for the next one: no memset between the slices)
		const IncrementalStrips& m = s->stripInc;
		int first = 0;
		for (int i = 0; i < n; ++i)
		{
			const Op& o = kept[(size_t)i];
			const bool sweep = o.code == OP_WARM || leanSoftKind(o);
			if (!sweep && i + 1 < n)
			{
				continue;
			}
			launchWideStep(st, kind, s->cv, s->bv, s->leanA, pv, (const Op*)s->dPersistOps.p + first, i + 1 - first, nullptr);
			count();
			if (sweep)
			{
				launchOverflowSweep(st, o, s->cv, s->bv, m.overflowBegin, m.overflowEnd); // (one launch: the contacts in it one after the other)
				count();
			}
			first = i + 1;
		}
	}

	void clearGranules(hipStream_t where)
	{
		(void)hipMemsetAsync(s->dGranules.p, 0, s->granuleBytes, where); // epochs restart at 1 every launch
		count();
	}

	// ... or in ONE launch that carries an overflow workgroup (wide_kernel.hip: wideOverflowWorker): the strips hand the bodies the
	// overflow contacts touch to it after every sweep and take them back.  One more workgroup has to be co-resident.
	bool overflowKernelPlan() const
	{
		return slicedPlan() && s->optOverflowKernel != 0 && !s->overflowKernelFailed && s->persist.overflowBodies != nullptr && s->dStripA.view.groupCount + 1 <= s->cuCount;
	}

	void runPersistent(int kind, int warm, bool clearFirst = false, bool selfContained = false, bool overflowKernel = false)
	{
		// hand-off tags are the exchange number; the step's epilogue kernel leaves the buffers zeroed for the next
		// step, so they only need clearing when this launch is replayed on its own (s2amd_measure_dominant)
		if (clearFirst)
		{
			clearGranules(st);
		}
		if (profile)
		{
			recordEvent();
		}
		PersistView pv = s->persist;
		pv.nearHandoff = s->nearHandoffNow;
		for (int i = 0; i < 2; ++i)
		{
			pv.softCoef[i] = make_float4(p.sc.softCoef[i][0], p.sc.softCoef[i][1], p.sc.softCoef[i][2], 0.0f);
		}
		if (!(kind == SOFT_TGS && warm == WARM_CURRENT))
		{
			pv.ldsRecords = s->persistRecordsWide;
		}
		if (widePlan(kind, warm))
		{
			WideSelf self{};
			if (selfContained)
			{
				self.wire = wireContacts(), self.wireBodies = wireBodies(), self.hostFlags = (const uint32_t*)s->dBodyFlags.p;
				self.warmStart = p.sc.warmStart, self.gravityX = p.sc.gravityX, self.gravityY = p.sc.gravityY, self.unpackH = p.unpackH;
			}
			pv.bodyWarm = (kind == SOFT_TGS && wideBodyWarm(selfContained)) ? 1 : 0;
			if (overflowKernel)
			{
				const IncrementalStrips& m = s->stripInc;
				pv.bodyWarm = 0;
				pv.overflowKernel = 1;
				pv.overflowBodyCount = (int)m.overflowBodyIds.size();
				pv.overflowBegin = m.overflowBegin, pv.overflowEnd = m.overflowEnd;
			}
			launchWideStep(st, kind, s->cv, s->bv, s->leanA, pv, (const Op*)s->dPersistOps.p, s->persistOpCount, selfContained ? &self : nullptr);
		}
		else if (pv.pairLanes && s->optPairLanes)
		{
			launchPairStep(st, kind, warm, s->cv, s->bv, s->leanA, pv, (const Op*)s->dPersistOps.p, s->persistOpCount);
		}
		else
		{
			launchStripStep(st, kind, warm, s->cv, s->bv, s->leanA, pv, (const Op*)s->dPersistOps.p, s->persistOpCount);
		}
		if (profile)
		{
			recordEvent();
		}
		count();
	}

	// the plan over the strips: body ops ride with the next sweep's phase A launch; every sweep is
	// phase A (interiors, all strips) then phase B (seams)
	void runGeneric()
	{
		if (profile)
		{
			recordEvent();
		}
		// the strips' joints resident in LDS when every strip's fit beside its bodies (JointGrid: 176 B x ~400 joints per strip)
		const size_t withJoints = genericStepLds(s->genericBodies, s->genericSeamBodies, s->genericExports, s->persistOpCount, p.usesDq0 ? 1 : 0, s->genericJoints);
		const bool stage = s->optStageJoints != 0 && s->genericJoints > 0 && withJoints <= 160 * 1024;
		const size_t lds = stage ? withJoints : genericStepLds(s->genericBodies, s->genericSeamBodies, s->genericExports, s->persistOpCount, p.usesDq0 ? 1 : 0);
		PersistView gpv = s->persist;
		gpv.nearHandoff = s->nearHandoffNow;
		launchGenericStep(st, s->cv, s->jv, s->bv, s->dStripA.view, s->dStripB.view, gpv, (const Op*)s->dPersistOps.p, s->persistOpCount, p.sc,
						  wireContacts(), p.usesDq0 ? 1 : 0, s->contacts.seamCount > 0 ? 1 : 0, s->joints.seamCount > 0 ? 1 : 0, lds, stage ? 1 : 0);
		if (profile)
		{
			recordEvent();
		}
		count();
	}

	void runStrips()
	{
		int kind, warm;
		if (persistPlan(kind, warm))
		{
			if (slicedPlan() && widePlan(kind, warm))
			{
				if (overflowKernelPlan())
				{
					runPersistent(kind, warm, false, false, true);
				}
				else
				{
					runPersistentSliced(kind, warm); // (doStep has made sure that overflow contacts only meet this kernel)
				}
			}
			else
			{
				runPersistent(kind, warm);
			}
			return;
		}
		if (genericPlan())
		{
			runGeneric();
			return;
		}
		const int n = (int)p.ops.size();
		int segStart = 0;
		for (int i = 0; i < n; ++i)
		{
			const Op& o = p.ops[(size_t)i];
			if (isBodyOp(o.code) || sweepsNothing(o))
			{
				continue; // a body op rides along; a sweep over nothing is a no-op wherever it lands
			}
			StripOps lean;
			int warm = -1;
			if (o.code == OP_WARM)
			{
				// folded into the lean launch of the next sweep when that launch can take it
				int j = i + 1;
				while (j < n && (isBodyOp(p.ops[(size_t)j].code) || sweepsNothing(p.ops[(size_t)j])))
				{
					j += 1;
				}
				if (j < n && leanSegment(segStart, j, lean, warm) && warm >= 0)
				{
					continue;
				}
			}
			const bool joint = o.code == OP_JOINT_SWEEP;
			const bool timed = profile && isSolveSweep(o.code);
			const bool seam = (joint ? s->joints.seamCount : s->contacts.seamCount) > 0;
			if (leanSegment(segStart, i, lean, warm))
			{
				if (timed)
				{
					recordEvent();
				}
				launchStripSoft(st, o.kind, warm, s->cv, s->bv, s->leanA, lean);
				if (timed)
				{
					recordEvent();
				}
				count();
			}
			else
			{
				launchStripGroups(s->dStripA, segStart, i + 1 - segStart, timed);
			}
			if (seam)
			{
				if (s->leanBValid && leanSoftKind(o))
				{
					StripOps only{};
					only.sweep = 1, only.useBias = o.useBias, only.inv_h = o.inv_h;
					if (timed)
					{
						recordEvent();
					}
					launchStripSoft(st, o.kind, -1, s->cv, s->bv, s->leanB, only);
					if (timed)
					{
						recordEvent();
					}
					count();
				}
				else
				{
					launchStripGroups(s->dStripB, i, 1, timed);
				}
			}
			segStart = i + 1;
		}
		if (segStart < n)
		{
			launchStripGroups(s->dStripA, segStart, n - segStart, false);
		}
	}

	// Can the whole plan run as ONE launch of jacobiStepKernel (jacobi_kernel.hip)?  s2Solve_Jacobi's plan on a structure whose block
	// tables were built (solver_jacobi.cpp): nothing in LDS groups or strips (a Jacobi structure has neither), no message passing.
	bool jacobiPlan() const
	{
		if (!s->jacobiValid || s->persistFailed || s->optJacobiPersist == 0 || msg || p.ops.size() > 128 || p.prepContacts != PREP_SOFT || p.storeKind != STORE_PLAIN ||
			s->dGroups.view.groupCount > 0 || s->dResident.view.groupCount > 0 || s->dStripA.view.groupCount > 0)
		{
			return false;
		}
		bool sweeps = false;
		for (const Op& o : p.ops)
		{
			const bool body = o.code == OP_INTEGRATE_VEL || o.code == OP_INTEGRATE_POS || o.code == OP_FINALIZE || o.code == OP_JACOBI_APPLY;
			const bool warm = o.code == OP_WARM && o.kind == WARM_CURRENT;
			const bool joint = o.code == OP_JOINT_SWEEP && (o.kind == JSOLVE_WARM || o.kind == JSOLVE_SOFT);
			const bool pass = o.code == OP_SOLVE_SOFT && o.kind == SOFT_JACOBI;
			sweeps = sweeps || pass;
			if (!(body || warm || joint || pass))
			{
				return false;
			}
		}
		return sweeps && jacobiStepLds(s->jacobiMaxOwned, s->jacobiMaxImports, s->jacobiMaxConstraints, (int)p.ops.size()) <= 160 * 1024;
	}

	void run()
	{
		if (p.earlyOut)
		{
			return;
		}
		// pre: wire -> SoA.  With contacts to prepare, ONE launch does the three independent prologue jobs (prepare
		// contacts, unpack bodies, manifold.constraintIndex); otherwise the unpack launch carries the index.
		const bool prepares = p.prepContacts >= 0 && s->cv.count > 0;
		ContactView cvIo = s->cv; // what the prologue / epilogue launches see
		{
			int kind, warm;
			if (residentPlan(kind, warm))
			{
				cvIo.skipBegin = s->residentK0, cvIo.skipEnd = s->residentK1; // the island kernel is its own prologue and epilogue
			}
		}
		const bool self = selfContainedIslands();
		if (self)
		{
			runResidentGroups(true);
			return;
		}
		if (selfContainedStrips())
		{
			int kind, warm;
			(void)persistPlan(kind, warm);
			runPersistent(kind, warm, false, true);
			return;
		}
		// the joints' preparation rides in the prologue launch (joint_prep.h: it reads the wire records, like the contacts')
		JointPrepArgs jp{};
		const bool joints = p.prepJoints >= 0 && s->jv.count > 0;
		if (joints)
		{
			jp.jv = s->jv, jp.wire = wireJoints(), jp.h = p.jprepH, jp.hertz = p.jprepHertz, jp.kind = p.prepJoints, jp.warmStart = p.jprepWarm;
			jp.blocks = (s->jv.count + 255) / 256;
		}
		bool carried;
		if (prepares)
		{
			carried = launchPrepareContacts(st, p.prepContacts, cvIo, s->bv, wireContacts(), wireBodies(), p.sc, p.prepH, p.prepHertz, posSolver,
											(const uint32_t*)s->dBodyFlags.p, true, p.unpackH, s->contactCapacity, gatherIndex, joints ? &jp : nullptr);
			count();
		}
		else
		{
			carried = launchUnpackBodies(st, s->bv, wireBodies(), (const uint32_t*)s->dBodyFlags.p, p.sc, p.unpackH, wireContacts(), s->contactCapacity, gatherIndex,
										 joints ? &jp : nullptr, posSolver);
			count();
		}
		if (joints && !carried)
		{
			launchPrepareJoints(st, p.prepJoints, s->jv, (const uint32_t*)s->dBodyFlags.p, wireJoints(), wireBodies(), p.sc, p.jprepH, p.jprepHertz, p.jprepWarm, posSolver);
			count();
		}
		if (msg)
		{
			launchFillMessageSlots(st, s->cv, s->bv, s->msg, s->contacts.globalCount);
			count();
		}
		if (jacobiPlan())
		{
			// s2Solve_Jacobi: prologue, ONE persistent launch over the blocks, epilogue (which clears the exchange buffers and stands down
			// when a hand-off timed out)
			JacobiView jview = s->jacobi;
			jview.debugSkip = s->optPersistDebug;
			launchJacobiStep(st, s->cv, s->jv, s->bv, jview, deviceOps(), (int)p.ops.size(), p.sc,
							 jacobiStepLds(s->jacobiMaxOwned, s->jacobiMaxImports, s->jacobiMaxConstraints, (int)p.ops.size()), s->jacobiMaxConstraints);
			count();
			s->stage4Carried = launchStoreImpulses(st, p.storeKind, cvIo, wireContacts(), p.storeScale, s->bv, wireBodies(), s->dJacobiGran.p, s->jacobiGranBytes,
													   s->jacobi.deviceError, -1, &s->jv, wireJoints(), &s->stage4);
			count();
			return;
		}
		// LDS groups: the whole op list in one launch
		if (s->dGroups.view.groupCount > 0)
		{
			launchGroupKernel(st, s->cv, s->jv, s->bv, s->dGroups.view, deviceOps(), (int)p.ops.size(), p.sc, wireContacts(), s->dGroups.maxBodies,
							  p.usesDq0 ? 1 : 0);
			count();
		}
		runResidentGroups();
		if (s->dStripA.view.groupCount > 0)
		{
			runStrips();
		}
		// global part: op by op
		int fusedFinalize = -1; // >= 0: the dynamicOnly flag of the s2FinalizePositions the epilogue launch performs
		const bool anyGlobal = s->looseBodies > 0 || anyGlobalContacts() || s->joints.globalCount > 0;
		if (anyGlobal)
		{
			const int n = (int)p.ops.size();
			std::vector<uint8_t> done((size_t)n, 0);
			const bool looseBodies = s->looseBodies > 0;
			const int lastBodyOp = n - 1; // (every plan ends with s2FinalizePositions; the store is the epilogue, not an op)
			for (int i = 0; i < n; ++i)
			{
				if (done[(size_t)i])
				{
					continue;
				}
				const Op& o = p.ops[(size_t)i];
				// the step's last body stage, s2FinalizePositions, rides in the epilogue launch (with the body write-back)
				if (o.code == OP_FINALIZE && !msg && looseBodies && i == lastBodyOp && wireBodies() != nullptr)
				{
					fusedFinalize = o.flag;
					done[(size_t)i] = 1;
					continue;
				}
				// joint warm start as ONE body-centric launch instead of one per joint colour (and the joints' sequential tail)
				if (o.code == OP_JOINT_SWEEP && o.kind == JSOLVE_WARM && s->optBodyWarm && s->jointAdjValid && s->joints.globalCount > 0)
				{
					launchWarmStartJointsBodies(st, s->jv, s->bv, (const int2*)s->dJointAdjRange.p, (const int*)s->dJointAdjList.p);
					count();
					done[(size_t)i] = 1;
					continue;
				}
				// contact warm start as ONE body-centric launch; an immediately preceding integrate-velocities
				// (joint sweeps in between only when there are no global joints) rides along in the same kernel
				// (a structure built for s2Solve_Jacobi has no colours: the body-centric form is the only one there)
				if (!msg && (s->optBodyWarm || s->orderColourless) && s->contacts.globalCount > 0 && (o.code == OP_WARM || o.code == OP_INTEGRATE_VEL))
				{
					int w = i;
					if (o.code == OP_INTEGRATE_VEL)
					{
						w = i + 1;
						while (w < n && p.ops[(size_t)w].code == OP_JOINT_SWEEP && s->joints.globalCount == 0)
						{
							w += 1;
						}
					}
					if (w < n && p.ops[(size_t)w].code == OP_WARM)
					{
						launchWarmStartBodies(st, p.ops[(size_t)w].kind, s->cv, s->bv, (const int2*)s->dAdjOffsets.p, (const int*)s->dAdjList.p,
											  o.code == OP_INTEGRATE_VEL ? 1 : 0, (const int*)s->dAdjHeavy.p, s->adjHeavyCapacity);
						count();
						for (int d = i; d <= w; ++d)
						{
							done[(size_t)d] = 1;
						}
						continue;
					}
				}
				runGlobalOp(i);
			}
		}
		if (msg)
		{
			launchGatherMessageSlots(st, s->bv, s->msg);
			count();
		}
		// post: SoA -> wire: impulses and bodies in one launch (+ the epoch base of the hand-off tags)
		const bool usedGranules = s->dStripA.view.groupCount > 0 && oneLaunchPlan();
		s->stage4Carried = launchStoreImpulses(st, p.storeKind, cvIo, wireContacts(), p.storeScale, s->bv, wireBodies(), usedGranules ? s->dGranules.p : nullptr,
												   usedGranules ? s->granuleBytes : 0, usedGranules ? s->persist.deviceError : nullptr, fusedFinalize, &s->jv, wireJoints(),
												   &s->stage4);
		count();
	}
};
