// Persistent strip step for EVERY solver family and for joints: genericStepKernel.
//
// strip_kernel.hip / wide_kernel.hip keep the soft contact drivers' constraints in registers; everything else that lives in a
// big island -- the rigid, sticky, NGS, block and XPBD contact sweeps, every joint sweep (src/joint.c:294-465) -- used to run
// one launch per colour batch (30-60 launches per step).  Here workgroup i owns strip i of the same partition for the whole
// step and walks the complete op list of the solver driver, as the LDS group kernel does for a small island
// (group_kernel.hip): bodies staged in LDS, constraint records streamed from their SoA arrays (they stay in the XCD's L2:
// ~120 KB per strip), `s_barrier` between colour batches.
//
// Seams are swept ONCE, by the left strip of the seam (the soft kernels sweep them redundantly on both sides, which needs
// the constraint state private to a workgroup; here it lives in the shared SoA arrays).  Per constraint op:
//
//   interior rounds
//   forward hand-off   strip i+1 publishes the bodies seam i touches on its side {v, w, dp, q}       (epoch E)
//   seam i rounds      in strip i, on its own last level + the imported copies
//   return hand-off    strip i writes the imported bodies back, strip i+1 takes them over             (epoch E+1)
//
// Seam i and seam i+1 share no writable body (solver_structure.cpp: partitionStrips), so strip i+1 sweeps seam i+1 while
// strip i sweeps seam i; what strip i+1 holds of its first level is stale in between and is not read.  The hand-offs are the
// 8-byte {epoch, value} granules of persist_handoff.h; forward and return strictly alternate between the two workgroups of a
// seam, so ONE buffer serves both (the two "parities" of the soft kernels' seam buffer hold the velocity and the pose here).
// The sequential-equivalent order is the multi-launch strip path's: per op, strip by strip colour-major, then seam by seam
// (s2amd_get_contact_order / _joint_order report it).

#include <type_traits>

#include "body_ops.h"
#include "group_ops.h"
#include "persist_handoff.h"

#define S2_GENERIC_THREADS 512
// 1: in-kernel time stamps (S2AMD_DEBUG_TIMES; wall_clock64 ticks of 10 ns, one workgroup in the middle) are compiled in: after the loads,
// and per constraint op after the interiors, the forward hand-off, the seam and the return hand-off (tools/generic_stamps.sh)
#ifndef S2_GENERIC_INSTRUMENTED
#define S2_GENERIC_INSTRUMENTED 0
#endif
#define S2_GENERIC_BATCH_RECORDS 64 // colour batches of a strip and its seam (contacts and joints) whose descriptors are kept in LDS

// seam-group-local body slots -> this workgroup's LDS slots (PersistView::remap, staged in LDS)
struct SeamBodies
{
	static constexpr int kMode = S2_IDX_LOCAL;
	static constexpr bool kLdsMass = false;
	float4* vel;
	float4* dq;
	const int* remap;
	S2_DEV float4 getVel(int i) const { return vel[remap[i]]; }
	S2_DEV void setVel(int i, float4 v) const { vel[remap[i]] = v; }
	S2_DEV float4 getDq(int i) const { return dq[remap[i]]; }
	S2_DEV void setDq(int i, float4 v) const { dq[remap[i]] = v; }
};

// a seam body's {v, w} and {dp, q} as seven tagged granules; `near`: the reader shares this workgroup's L2
S2_DEV void putBody(gu64* p, int poseOffset, unsigned epoch, float4 v, float4 d, bool near)
{
	gu64* q = p + poseOffset;
	if (near)
	{
		putGranuleNear(p + 0, epoch, v.x), putGranuleNear(p + 1, epoch, v.y), putGranuleNear(p + 2, epoch, v.z);
		putGranuleNear(q + 0, epoch, d.x), putGranuleNear(q + 1, epoch, d.y), putGranuleNear(q + 2, epoch, d.z), putGranuleNear(q + 3, epoch, d.w);
	}
	else
	{
		putGranule(p + 0, epoch, v.x), putGranule(p + 1, epoch, v.y), putGranule(p + 2, epoch, v.z);
		putGranule(q + 0, epoch, d.x), putGranule(q + 1, epoch, d.y), putGranule(q + 2, epoch, d.z), putGranule(q + 3, epoch, d.w);
	}
}

// one constraint op over a list of colour batches (the switch of groupKernel, one constraint per call)
template <class BA, class JV>
S2_DEV void sweepOp(const Op& op, const ContactView& c, const JV& jv, const BA& lb, const StepConsts& sc, s2amdContact* wire, const int4* cBatches,
					int cb0, int cb1, const int4* jBatches, int jb0, int jb1, int jBase)
{
	// jBase: the joint view's arrays start at sweep position jBase (a view of LDS-resident records: genericStepKernel), 0 for the global arrays
	auto pfC = [&](int k) { prefetchContact(c, k); };
	auto pfJ = [&](int k) { prefetchJoint(jv, k - jBase); };
	switch (op.code)
	{
		case OP_JOINT_SWEEP:
			switch (op.kind)
			{
				case JSOLVE_PLAIN:
					forBatches(jBatches, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_PLAIN>(jv, lb, sc, op.h, op.inv_h, op.useBias, k - jBase); });
					break;
				case JSOLVE_SOFT:
					forBatches(jBatches, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_SOFT>(jv, lb, sc, op.h, op.inv_h, op.useBias, k - jBase); });
					break;
				case JSOLVE_BAUMGARTE:
					forBatches(jBatches, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_BAUMGARTE>(jv, lb, sc, op.h, op.inv_h, op.useBias, k - jBase); });
					break;
				case JSOLVE_POSITION:
					forBatches(jBatches, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_POSITION>(jv, lb, sc, op.h, op.inv_h, op.useBias, k - jBase); });
					break;
				case JSOLVE_XPBD:
					forBatches(jBatches, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_XPBD>(jv, lb, sc, op.h, op.inv_h, op.useBias, k - jBase); });
					break;
				case JSOLVE_WARM:
					forBatches(jBatches, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_WARM>(jv, lb, sc, op.h, op.inv_h, op.useBias, k - jBase); });
					break;
			}
			break;
		case OP_WARM:
			switch (op.kind)
			{
				case WARM_CURRENT:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { warmStartContactsOne<WARM_CURRENT>(c, lb, k); });
					break;
				case WARM_FIXED:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { warmStartContactsOne<WARM_FIXED>(c, lb, k); });
					break;
				case WARM_BLOCK:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { warmStartContactsOne<WARM_BLOCK>(c, lb, k); });
					break;
			}
			break;
		case OP_SOLVE_SOFT:
			switch (op.kind)
			{
				case SOFT_TGS:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { solveContactsSoftOne<SOFT_TGS>(c, lb, op.inv_h, op.useBias, k); });
					break;
				case SOFT_PGS:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { solveContactsSoftOne<SOFT_PGS>(c, lb, op.inv_h, op.useBias, k); });
					break;
				case SOFT_FIXED:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { solveContactsSoftOne<SOFT_FIXED>(c, lb, op.inv_h, op.useBias, k); });
					break;
				default:
					break; // SOFT_JACOBI never runs here (Executor::genericPlan)
			}
			break;
		case OP_SOLVE_RIGID:
			switch (op.kind)
			{
				case RIGID_BAUMGARTE:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { solveContactsRigidOne<RIGID_BAUMGARTE>(c, lb, op.inv_h, k); });
					break;
				case RIGID_PGS:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { solveContactsRigidOne<RIGID_PGS>(c, lb, op.inv_h, k); });
					break;
				case RIGID_TGS:
					forBatches(cBatches, cb0, cb1, pfC, [&](int k) { solveContactsRigidOne<RIGID_TGS>(c, lb, op.inv_h, k); });
					break;
			}
			break;
		case OP_SOLVE_STICKY:
			forBatches(cBatches, cb0, cb1, pfC, [&](int k) { solveContactsStickyOne(c, lb, wire, op.inv_h, op.useBias, k); });
			break;
		case OP_SOLVE_NGS:
			forBatches(cBatches, cb0, cb1, pfC, [&](int k) { solveContactsNGSOne(c, lb, k); });
			break;
		case OP_XPBD_POS:
			forBatches(cBatches, cb0, cb1, pfC, [&](int k) { xpbdContactPositionsOne(c, lb, op.h, k); });
			break;
		case OP_XPBD_VEL:
			forBatches(cBatches, cb0, cb1, pfC, [&](int k) { xpbdContactVelocitiesOne(c, lb, op.h, k); });
			break;
		case OP_BLOCK_VEL:
			forBatches(cBatches, cb0, cb1, pfC, [&](int k) { blockSolveVelocityOne(c, lb, k); });
			break;
		case OP_BLOCK_POS:
			forBatches(cBatches, cb0, cb1, pfC, [&](int k) { blockSolvePositionOne(c, lb, k); });
			break;
		default:
			break;
	}
}

__global__ __launch_bounds__(S2_GENERIC_THREADS) void genericStepKernel(ContactView c, JointView jv, BodyView g, GroupTable ga, GroupTable gb, PersistView pv,
																		const Op* ops, int opCount, StepConsts sc, s2amdContact* wire, int useDq0,
																		int seamContacts, int seamJoints, int stageJoints)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	const int tid = (int)threadIdx.x;
	const bool stamp = S2_GENERIC_INSTRUMENTED && pv.debugTimes != nullptr && blockIdx.x == gridDim.x / 2 && tid == 0;
	int stamps = 0;
	auto stampAt = [&]() {
		if (stamp && stamps < 250)
		{
			pv.debugTimes[stamps++] = wall_clock64();
		}
	};
	stampAt(); // (kernel start)
	// Strip <-> workgroup: consecutive strips on ONE XCD, so that a seam's two workgroups share an L2 (the dispatcher is observed
	// to place block b on XCD b % 8: a speed assumption only).  The census (wide_kernel.hip): every workgroup publishes the XCD it
	// REALLY runs on, and a hand-off takes the L2 path (workgroup-scope stores, no write-through) only towards a neighbour that
	// was seen on the same XCD -- results never depend on placement.
	const int K = (int)gridDim.x;
	const int xcd = (int)blockIdx.x & 7, lane8 = (int)blockIdx.x >> 3;
	const int firstOfXcd = xcd * (K >> 3) + (xcd < (K & 7) ? xcd : (K & 7));
	const int countOfXcd = (K >> 3) + (xcd < (K & 7) ? 1 : 0);
	const int strip = firstOfXcd + lane8;
	unsigned myXcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(myXcc));
	gu64* census = (gu64*)pv.granules + pv.censusBase;
	if (tid == 0)
	{
		putGranule(census + strip, 1u, __uint_as_float(myXcc + 1u));
	}
	const PersistDesc* pd = pv.descs + strip;
	const int bodyBase = ga.bodyOffsets[strip];
	const int nb = ga.bodyOffsets[strip + 1] - bodyBase;
	const int nImp0 = pd->importCount[0], nImp1 = pd->importCount[1];
	const int nExp0 = pd->exportCount[0];
	const int nt = nb + nImp0 + nImp1;
	const int impBase = nb + nImp0; // the copies of the right neighbour's bodies (the left neighbour's slots stay unused here)
	const int seam = pd->seamGroup[1]; // the seam this workgroup sweeps: strip | strip + 1
	const int seamBodies = seam >= 0 ? gb.bodyOffsets[seam + 1] - gb.bodyOffsets[seam] : 0;
	gu64* gran = (gu64*)pv.granules;
	gu64* outLeft = gran + pd->outBase[0];	 // seam strip-1 | strip: written by me (forward), then by the left strip (return)
	gu64* inRight = gran + pd->inBase[1];	 // seam strip | strip+1: written by the right strip (forward), then by me (return)
	const int poseOffset = pv.parityStride; // the buffer's second half carries {dp, q}

	float4* lvel = lds;
	float4* ldq = lds + nt;
	float4* ldq0 = lds + 2 * nt; // only addressed when useDq0
	float2* lmass = (float2*)(lds + (useDq0 ? 3 : 2) * nt);
	int* lremap = (int*)(lmass + nt + (nt & 1));
	int* lexp = lremap + seamBodies;
	Op* lops = (Op*)(lds + (useDq0 ? 3 : 2) * nt + (nt + 1) / 2 + (seamBodies + nExp0 + 3) / 4);
	const int* ids = ga.bodyIds + bodyBase;

	for (int i = tid; i < nt; i += S2_GENERIC_THREADS)
	{
		const int gi = i < nb ? (int)((uint32_t)ids[i] & ~S2G_OWNED) : (i < impBase ? -1 : pv.importIds[pd->importIdBase[1] + i - impBase]);
		if (gi >= 0)
		{
			lvel[i] = g.vel[gi];
			ldq[i] = g.dq[gi];
			lmass[i] = g.massInv[gi];
			if (useDq0)
			{
				ldq0[i] = g.dq0[gi];
			}
		}
	}
	for (int i = tid; i < seamBodies; i += S2_GENERIC_THREADS)
	{
		lremap[i] = pv.remap[pd->remapBase[1] + i];
	}
	for (int i = tid; i < nExp0; i += S2_GENERIC_THREADS)
	{
		lexp[i] = pv.exportSrc[pd->exportSrcBase[0] + i];
	}
	for (int i = tid; i < opCount * 8; i += S2_GENERIC_THREADS)
	{
		((int*)lops)[i] = ((const int*)ops)[i];
	}
	// the neighbours' census entries have had the load phase to land; a missing one only costs the fast path
	int* lnear = (int*)(lops + opCount);
	if (tid < 2)
	{
		const bool hope = pv.nearHandoff != 0 && (tid == 0 ? strip > firstOfXcd : strip + 1 < firstOfXcd + countOfXcd);
		int near = 0;
		if (hope)
		{
			gu64* gc = census + (tid ? strip + 1 : strip - 1);
			for (int spins = 0; spins < 4096 && !near; ++spins)
			{
				const u64 x = __hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((unsigned)(x >> 32) == 1u)
				{
					near = (unsigned)x == myXcc + 1u ? 1 : -1;
				}
			}
		}
		lnear[tid] = near > 0 ? 1 : 0;
	}
	__syncthreads();
	const bool nearLeft = lnear[0] != 0, nearRight = lnear[1] != 0;

	// ---- joints resident in LDS (stageJoints: the host has found room for every strip's): the records of this strip's interior
	// joints and of the seam it sweeps -- one contiguous range of sweep positions each -- are read from HBM / L2 once per step
	// instead of once per sweep; the per-joint functions reach them through a view of typed LDS columns (constraint_ops.h: LdsJointView).  Sweeps write
	// `impulse` and `axial` (constraint_ops.h: solveJointsOne): those two go back at the end.
	const int jb0 = ga.jBatchOffsets[strip], jb1 = ga.jBatchOffsets[strip + 1];
	const int sjb0 = seam >= 0 ? gb.jBatchOffsets[seam] : 0, sjb1 = seam >= 0 ? gb.jBatchOffsets[seam + 1] : 0;
	const int jA0 = jb0 < jb1 ? ga.jBatches[jb0].x : 0, jA1 = jb0 < jb1 ? ga.jBatches[jb1 - 1].y : 0;
	const int jS0 = sjb0 < sjb1 ? gb.jBatches[sjb0].x : 0, jS1 = sjb0 < sjb1 ? gb.jBatches[sjb1 - 1].y : 0;
	const int njA = jA1 - jA0, nj = njA + (jS1 - jS0);
	LdsJointView ljA{}, ljS{};
	int jBaseA = 0, jBaseS = 0;
	const bool jointsInLds = stageJoints != 0 && nj > 0;
	LdsColumn<float2, f2v> blkImpulse{nullptr};
	LdsColumn<float4, f4v> blkAxial{nullptr};
	if (jointsInLds)
	{
		char* at = (char*)(lnear + 4) + S2_GENERIC_BATCH_RECORDS * sizeof(int4);
		auto kOf = [&](int i) { return i < njA ? jA0 + i : jS0 + (i - njA); };
		// one typed LDS column per array: the strip's interior joints first, the seam's behind them
		auto stage = [&](auto& colA, auto& colS, const auto* src) {
			typedef typename std::remove_reference_t<decltype(colA)>::Cell Cell;
			Cell* blk = (Cell*)at;
			at += (size_t)nj * sizeof(Cell);
			for (int i = tid; i < nj; i += S2_GENERIC_THREADS)
			{
				blk[i] = toLds(src[kOf(i)]);
			}
			colA.p = blk, colS.p = blk + njA;
		};
		stage(ljA.frame, ljS.frame, jv.frame), stage(ljA.mass, ljS.mass, jv.mass), stage(ljA.pivot, ljS.pivot, jv.pivot);
		stage(ljA.soft, ljS.soft, jv.soft), stage(ljA.axial, ljS.axial, jv.axial), stage(ljA.limits, ljS.limits, jv.limits);
		stage(ljA.misc, ljS.misc, jv.misc);
		stage(ljA.centerDiff0, ljS.centerDiff0, jv.centerDiff0), stage(ljA.impulse, ljS.impulse, jv.impulse);
		stage(ljA.localBodies, ljS.localBodies, jv.localBodies);
		blkImpulse = ljA.impulse, blkAxial = ljA.axial;
		// (what stays in global memory is addressed from the same origin: in-range pointers only)
		ljA.bodies = jv.bodies + jA0, ljS.bodies = jv.bodies + jS0;
		jBaseA = jA0, jBaseS = jS0;
		__syncthreads();
	}

	LdsMassBodies lb;
	lb.vel = lvel, lb.dq = ldq, lb.massInv = lmass;
	lb.softCoef[0] = make_float4(sc.softCoef[0][0], sc.softCoef[0][1], sc.softCoef[0][2], 0.0f);
	lb.softCoef[1] = make_float4(sc.softCoef[1][0], sc.softCoef[1][1], sc.softCoef[1][2], 0.0f);
	lb.softDiet = sc.softDiet;
	SeamBodies sb{lvel, ldq, lremap};
	const int cb0 = ga.cBatchOffsets[strip], cb1 = ga.cBatchOffsets[strip + 1];
	const int scb0 = seam >= 0 ? gb.cBatchOffsets[seam] : 0, scb1 = seam >= 0 ? gb.cBatchOffsets[seam + 1] : 0;

	// the colour batches' descriptors in LDS: a round starts by reading its {begin, end, tail} -- from global memory that is a
	// dependent L2 round trip on the critical path of every round
	const int4* batchC = ga.cBatches + cb0;
	const int4* batchJ = ga.jBatches + jb0;
	const int4* batchSC = gb.cBatches + scb0;
	const int4* batchSJ = gb.jBatches + sjb0;
	const int nC = cb1 - cb0, nJ = jb1 - jb0, nSC = scb1 - scb0, nSJ = sjb1 - sjb0;
	if (nC + nJ + nSC + nSJ <= S2_GENERIC_BATCH_RECORDS)
	{
		int4* lbatch = (int4*)(lnear + 4);
		for (int i = tid; i < nC + nJ + nSC + nSJ; i += S2_GENERIC_THREADS)
		{
			lbatch[i] = i < nC ? batchC[i] : (i < nC + nJ ? batchJ[i - nC] : (i < nC + nJ + nSC ? batchSC[i - nC - nJ] : batchSJ[i - nC - nJ - nSC]));
		}
		batchC = lbatch, batchJ = lbatch + nC, batchSC = lbatch + nC + nJ, batchSJ = lbatch + nC + nJ + nSC;
		__syncthreads();
	}

	stampAt(); // (loads done)
	unsigned epoch = 0; // the buffers are zero at launch (cleared by the previous step's epilogue)
	int bad = 0;
	for (int oi = 0; oi < opCount && !bad; ++oi)
	{
		const Op op = lops[oi];
		switch (op.code)
		{
			case OP_INTEGRATE_VEL:
				for (int i = tid; i < nb; i += S2_GENERIC_THREADS)
				{
					integrateVelocitiesOne(lb, i, g, (int)((uint32_t)ids[i] & ~S2G_OWNED));
				}
				__syncthreads();
				stampAt();
				continue;
			case OP_INTEGRATE_POS:
				for (int i = tid; i < nb; i += S2_GENERIC_THREADS)
				{
					integratePositionsOne(lb, i, g, (int)((uint32_t)ids[i] & ~S2G_OWNED), op.h);
				}
				__syncthreads();
				stampAt();
				continue;
			case OP_FINALIZE:
				for (int i = tid; i < nb; i += S2_GENERIC_THREADS)
				{
					const uint32_t id = (uint32_t)ids[i];
					finalizePositionsOne(lb, i, g, (int)(id & ~S2G_OWNED), op.flag, (id & S2G_OWNED) != 0);
				}
				__syncthreads();
				stampAt();
				continue;
			case OP_XPBD_INTEGRATE:
				for (int i = tid; i < nb; i += S2_GENERIC_THREADS)
				{
					xpbdIntegrateOne(lb, ldq0, i, g, (int)((uint32_t)ids[i] & ~S2G_OWNED), op.h);
				}
				__syncthreads();
				stampAt();
				continue;
			case OP_XPBD_PROJECT:
				for (int i = tid; i < nb; i += S2_GENERIC_THREADS)
				{
					xpbdProjectOne(lb, ldq0, i, g, (int)((uint32_t)ids[i] & ~S2G_OWNED), op.inv_h);
				}
				__syncthreads();
				stampAt();
				continue;
			default:
				break;
		}
		// ---- a constraint op: interiors ----
		if (jointsInLds)
		{
			sweepOp(op, c, ljA, lb, sc, wire, batchC, 0, nC, batchJ, 0, nJ, jBaseA);
		}
		else
		{
			sweepOp(op, c, jv, lb, sc, wire, batchC, 0, nC, batchJ, 0, nJ, 0);
		}
		stampAt();
		if ((op.code == OP_JOINT_SWEEP ? seamJoints : seamContacts) == 0)
		{
			continue; // nothing of this kind in any seam: every workgroup skips the hand-offs
		}
		// ---- forward: my bodies of the left seam to the left neighbour, the right neighbour's into my copies ----
		epoch += 1;
		const bool mute = (pv.debugSkip & 8) != 0 && blockIdx.x == 1; // fault injection: this workgroup stays silent
		for (int t = tid; t < nExp0 && !mute; t += S2_GENERIC_THREADS)
		{
			const float4 v = lvel[lexp[t]], d = ldq[lexp[t]];
			putBody(outLeft + 4 * t, poseOffset, epoch, v, d, nearLeft);
		}
		int fail = 0;
		for (int t = tid; t < nImp1; t += S2_GENERIC_THREADS)
		{
			float v[3], d[4];
			if (getGranules<3>(inRight + 4 * t, epoch, v, pv.error, pv.deviceError, pv.spinLimit) &&
				getGranules<4>(inRight + poseOffset + 4 * t, epoch, d, pv.error, pv.deviceError, pv.spinLimit))
			{
				lvel[impBase + t] = make_float4(v[0], v[1], v[2], 0.0f);
				ldq[impBase + t] = make_float4(d[0], d[1], d[2], d[3]);
			}
			else
			{
				fail = 1;
			}
		}
		bad = __syncthreads_or(fail);
		if (bad)
		{
			break;
		}
		stampAt();
		// ---- the seam to my right ----
		if (seam >= 0)
		{
			if (jointsInLds)
			{
				sweepOp(op, c, ljS, sb, sc, wire, batchSC, 0, nSC, batchSJ, 0, nSJ, jBaseS);
			}
			else
			{
				sweepOp(op, c, jv, sb, sc, wire, batchSC, 0, nSC, batchSJ, 0, nSJ, 0);
			}
		}
		stampAt();
		// ---- return: the right neighbour's bodies back to their owner, mine back from the left neighbour ----
		epoch += 1;
		for (int t = tid; t < nImp1; t += S2_GENERIC_THREADS)
		{
			const float4 v = lvel[impBase + t], d = ldq[impBase + t];
			putBody(inRight + 4 * t, poseOffset, epoch, v, d, nearRight);
		}
		for (int t = tid; t < nExp0; t += S2_GENERIC_THREADS)
		{
			float v[3], d[4];
			if (getGranules<3>(outLeft + 4 * t, epoch, v, pv.error, pv.deviceError, pv.spinLimit) &&
				getGranules<4>(outLeft + poseOffset + 4 * t, epoch, d, pv.error, pv.deviceError, pv.spinLimit))
			{
				lvel[lexp[t]] = make_float4(v[0], v[1], v[2], 0.0f);
				ldq[lexp[t]] = make_float4(d[0], d[1], d[2], d[3]);
			}
			else
			{
				fail = 1;
			}
		}
		bad = __syncthreads_or(fail);
		stampAt();
	}

	if (jointsInLds && !bad)
	{
		for (int i = tid; i < nj; i += S2_GENERIC_THREADS)
		{
			const int k = i < njA ? jA0 + i : jS0 + (i - njA);
			jv.impulse[k] = blkImpulse[i];
			jv.axial[k] = blkAxial[i];
		}
	}
	for (int i = tid; i < nb; i += S2_GENERIC_THREADS)
	{
		const uint32_t id = (uint32_t)ids[i];
		if (id & S2G_OWNED)
		{
			const int gi = (int)(id & ~S2G_OWNED);
			g.vel[gi] = lvel[i];
			g.dq[gi] = ldq[i];
			if (useDq0)
			{
				g.dq0[gi] = ldq0[i];
			}
		}
	}
	stampAt();
	if (stamp)
	{
		pv.debugTimes[254] = 0ull; // plain ticks (solver.cpp prints them at destroy)
		pv.debugTimes[255] = (unsigned long long)stamps;
	}
}

int genericKernelSetup()
{
	hipError_t e = hipFuncSetAttribute((const void*)genericStepKernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	return e == hipSuccess ? 0 : (int)e;
}

// bytes of dynamic LDS a strip with `bodies` staged bodies (own + both import ranges), `seamBodies` seam-group bodies, `exports`
// exported bodies and `opCount` ops needs (the host checks this against 160 KiB: Executor::genericPlan)
size_t genericStepLds(int bodies, int seamBodies, int exports, int opCount, int useDq0, int stagedJoints)
{
	const size_t records = (size_t)(useDq0 ? 3 : 2) * bodies + (size_t)(bodies + 1) / 2 + (size_t)(seamBodies + exports + 3) / 4 + 2 * (size_t)opCount + 1 + S2_GENERIC_BATCH_RECORDS; // (+ the census flags, the batch descriptors)
	// a staged joint: the seven 16-byte and two 8-byte arrays the sweeps read, and its local body pair (LdsJointView)
	return records * 16 + (size_t)stagedJoints * (7 * 16 + 2 * 8 + 8) + (stagedJoints ? 64 : 0);
}

void launchGenericStep(hipStream_t s, const ContactView& c, const JointView& j, const BodyView& g, const GroupTable& a, const GroupTable& b, const PersistView& pv,
					   const Op* ops, int opCount, const StepConsts& sc, s2amdContact* wire, int useDq0, int seamContacts, int seamJoints, size_t ldsBytes,
					   int stageJoints)
{
	if (a.groupCount <= 0 || opCount <= 0)
	{
		return;
	}
	genericStepKernel<<<dim3((unsigned)a.groupCount), dim3(S2_GENERIC_THREADS), ldsBytes, s>>>(c, j, g, a, b, pv, ops, opCount, sc, wire, useDq0, seamContacts,
																								 seamJoints, stageJoints);
}

S2_DEFINE_WARM(generic_kernel)
