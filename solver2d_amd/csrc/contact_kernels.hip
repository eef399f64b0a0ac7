// Contact-constraint kernels: one thread per constraint, constraints laid out SoA in sweep
// (colour-major) order so a wave reads 64 consecutive records of each array with 16-byte lanes.
// A sweep kernel is launched once per colour batch [begin, end): inside a batch no two
// constraints touch the same writable body, so the batch reproduces, bit for bit, a sequential
// Gauss-Seidel pass over the same constraints in the same order.  The per-constraint arithmetic
// lives in constraint_ops.h (shared with the LDS group kernel).

#include "body_ops.h"
#include "joint_prep.h"
#include "refit_ops.h"

#define S2_BLOCK 256

// ---------------------------------------------------------------------------------------------
// prepare: s2PrepareContacts_PGS / _Soft (solve_common.c:93-168, :188-274), s2PrepareContacts
// (solve_tgs_ngs.c:19-89), s2PrepareContacts_Sticky (solve_tgs_sticky.c:19-165),
// s2PrepareContacts_XPBD (solve_xpbd.c:18-86), and the first loop of s2CreateContactSolver
// (solve_pgs_ngs_block.c:151-277).  Reads the wire contact + two bodies, writes the SoA record.
// Embarrassingly parallel: no body is written.
// ---------------------------------------------------------------------------------------------
// The launch also carries the other two prologue jobs, which nothing here depends on (poses and write flags are read
// from the wire bodies and the host flag array, not from the SoA unpack fills): blocks [contactBlocks, +bodyBlocks)
// unpack the bodies (unpackBodyOne), the rest write manifold.constraintIndex for every contact slot.
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void prepareContactsKernel(ContactView c, BodyView b, s2amdContact* wire, const s2amdBody* wireBodies,
																  StepConsts sc, float h, float hertz, int posSolver, const uint32_t* hostFlags,
																  int contactBlocks, int bodyBlocks, float unpackH, int contactCapacity, const int* gatherIndex, JointPrepArgs jp)
{
	if ((int)blockIdx.x >= contactBlocks)
	{
		int rest = (int)blockIdx.x - contactBlocks;
		if (rest < bodyBlocks)
		{
			unpackBodyOne(b, wireBodies, hostFlags, sc, unpackH, rest * (int)blockDim.x + (int)threadIdx.x);
		}
		else if ((int)blockIdx.x >= (int)gridDim.x - jp.blocks)
		{
			// the joints' preparation (joint.c:297-447; joint_prep.h): from the wire records, like the contacts' above
			prepareJointsBlock(jp, hostFlags, wireBodies, sc, posSolver, (int)blockIdx.x - ((int)gridDim.x - jp.blocks));
		}
		else
		{
			int slot = (rest - bodyBlocks) * (int)blockDim.x + (int)threadIdx.x;
			if (slot < contactCapacity)
			{
				wire[slot].constraintIndex = gatherIndex[slot];
			}
		}
		return;
	}
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= c.skipBegin)
	{
		k += c.skipEnd - c.skipBegin; // the contact blocks cover [0, skipBegin) and [skipEnd, count) back to back
	}
	if (k >= c.count)
	{
		return;
	}
	const int slot = c.contactIndex[k];
	if (slot < 0)
	{
		// a free position of the sweep order (slack a colour batch keeps so that a created contact can be given a place
		// without rebuilding the structure, solver_incremental.cpp): an empty record, no bodies read or written
		const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		c.bodies[k] = make_int2(0, 0);
		c.mass[k] = zero;
		c.nf[k] = zero;
		for (int j = 0; j < 2; ++j)
		{
			c.anchor[j][k] = zero, c.r0[j][k] = zero, c.param[j][k] = zero, c.soft[j][k] = zero;
			c.impulse[j][k] = make_float2(0.0f, 0.0f);
		}
		if (KIND == PREP_BLOCK)
		{
			c.blockK[k] = zero, c.blockNM[k] = zero;
		}
		return;
	}
	s2amdContact* contact = wire + slot;
	int pointCount = contact->pointCount;
	int ia = contact->bodyA, ib = contact->bodyB;
	if (pointCount <= 0 && (ia < 0 || ib < 0 || ia >= b.capacity || ib >= b.capacity))
	{
		// a destroyed contact whose entry lingers in the structure until the next rebuild (solver_step.cpp: refreshShadows):
		// a free pool slot names no bodies; any body will do for a constraint that neither reads nor writes one
		pointCount = 0, ia = 0, ib = 0;
	}
	V2 normal = v2(contact->normal[0], contact->normal[1]);
	float friction = contact->friction;

	const s2amdBody* wa = wireBodies + ia;
	const s2amdBody* wb = wireBodies + ib;
	float mA = wa->invMass, iA = wa->invI;
	float mB = wb->invMass, iB = wb->invI;
	V2 lcA = v2(wa->localCenter[0], wa->localCenter[1]);
	V2 lcB = v2(wb->localCenter[0], wb->localCenter[1]);

	// rotation and write flags as unpackBodyOne would put them into the SoA (body_ops.h)
	Rot qA, qB;
	qA.s = wa->rot[0], qA.c = wa->rot[1];
	qB.s = wb->rot[0], qB.c = wb->rot[1];

	uint32_t fa = hostFlags[ia], fb = hostFlags[ib];
	uint32_t wbit = posSolver ? S2F_WRITE_POS : S2F_WRITE_VEL;
	// A manifold without points is a potential constraint only (solver_internal.h: hContactEdge): it keeps its place in the
	// sweep order but gets no write bits, so every sweep skips its point loops AND its body stores -- a no-op by
	// construction, which is what the reference does by never gathering it (e.g. solve_tgs_soft.c:162-179).
	uint32_t bits = (uint32_t)pointCount;
	if ((fa & wbit) && pointCount > 0)
	{
		bits |= S2C_WRITE_A;
	}
	if ((fb & wbit) && pointCount > 0)
	{
		bits |= S2C_WRITE_B;
	}

	c.bodies[k] = make_int2(ia, ib);
	c.mass[k] = make_float4(mA, iA, mB, iB);

	// solve_common.c:219
	float contactHertz = (mA == 0.0f || mB == 0.0f) ? 2.0f * hertz : hertz;
	V2 tangent = KIND == PREP_BLOCK ? crossVS(normal, 1.0f) : rightPerp(normal);

	float k11 = 0.0f, k22 = 0.0f, k12 = 0.0f;
	float rnA_[2] = {0.0f, 0.0f}, rnB_[2] = {0.0f, 0.0f};

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j >= pointCount)
		{
			// keep the unused slot defined (zeroed scratch in the reference: stack_allocator.c:84)
			c.anchor[j][k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			c.r0[j][k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			c.param[j][k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			c.soft[j][k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			c.impulse[j][k] = make_float2(0.0f, 0.0f);
			continue;
		}
		const s2amdManifoldPoint* mp = contact->points + j;
		float normalImpulse = 0.0f, tangentImpulse = 0.0f;
		bool copyImpulse = (KIND == PREP_PGS || KIND == PREP_SOFT || KIND == PREP_TGS || KIND == PREP_BLOCK) && sc.warmStart != 0;
		if (copyImpulse)
		{
			normalImpulse = mp->normalImpulse;
			tangentImpulse = mp->tangentImpulse;
		}

		V2 lA = sub(v2(mp->localAnchorA[0], mp->localAnchorA[1]), lcA);
		V2 lB = sub(v2(mp->localAnchorB[0], mp->localAnchorB[1]), lcB);
		V2 rA = rotate(qA, lA);
		V2 rB = rotate(qB, lB);

		float separation = mp->separation;
		float adjustedSeparation = separation - dot(sub(rB, rA), normal);

		float rtA = cross(rA, tangent);
		float rtB = cross(rB, tangent);
		float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
		float tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;

		float rnA = cross(rA, normal);
		float rnB = cross(rB, normal);
		float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
		float normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
		rnA_[j] = rnA, rnB_[j] = rnB;

		float biasCoefficient = 0.0f, massCoefficient = 0.0f, impulseCoefficient = 0.0f;
		if (KIND == PREP_PGS)
		{
			biasCoefficient = separation > 0.0f ? 1.0f : 0.0f;
		}
		else if (KIND == PREP_SOFT)
		{
			// solve_common.c:262-271
			const float zeta = 10.0f;
			float omega = 2.0f * S2_PI * contactHertz;
			float cc = h * omega * (2.0f * zeta + h * omega);
			biasCoefficient = omega / (2.0f * zeta + h * omega);
			impulseCoefficient = 1.0f / (1.0f + cc);
			massCoefficient = cc * impulseCoefficient;
		}
		else if (KIND == PREP_BLOCK)
		{
			// velocityBias lives in the biasCoefficient slot: solve_pgs_ngs_block.c:227
			biasCoefficient = -S2_MAXF(0.0f, separation * sc.inv_dt);
		}

		c.anchor[j][k] = make_float4(lA.x, lA.y, lB.x, lB.y);
		// TGS_NGS never sets rA0/rB0 (solve_tgs_ngs.c:19-89): zeroed scratch
		c.r0[j][k] = KIND == PREP_TGS ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(rA.x, rA.y, rB.x, rB.y);
		c.param[j][k] = make_float4(adjustedSeparation, normalMass, tangentMass, separation);
		c.soft[j][k] = make_float4(biasCoefficient, massCoefficient, impulseCoefficient, 0.0f);
		c.impulse[j][k] = make_float2(normalImpulse, tangentImpulse);
	}

	if (KIND == PREP_BLOCK)
	{
		// solve_pgs_ngs_block.c:245-276
		int reduced = pointCount;
		float4 K4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		float4 NM = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		if (pointCount == 2)
		{
			k11 = mA + mB + iA * rnA_[0] * rnA_[0] + iB * rnB_[0] * rnB_[0];
			k22 = mA + mB + iA * rnA_[1] * rnA_[1] + iB * rnB_[1] * rnB_[1];
			k12 = mA + mB + iA * rnA_[0] * rnA_[1] + iB * rnB_[0] * rnB_[1];
			const float k_maxConditionNumber = 1000.0f;
			if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12))
			{
				M22 K;
				K.cx = v2(k11, k12);
				K.cy = v2(k12, k22);
				M22 inv = inverse22(K);
				K4 = make_float4(k11, k12, k22, 0.0f);
				NM = make_float4(inv.cx.x, inv.cx.y, inv.cy.x, inv.cy.y);
			}
			else
			{
				reduced = 1;
			}
		}
		K4.w = fromBits((uint32_t)reduced);
		c.blockK[k] = K4;
		c.blockNM[k] = NM;
	}

	if (KIND == PREP_STICKY)
	{
		// friction anchor cache: solve_tgs_sticky.c:87-163 (reads and writes the manifold)
		V2 cA = v2(wa->position[0], wa->position[1]);
		V2 cB = v2(wb->position[0], wb->position[1]);
		bool frictionConfirmed = false;
		float tsep[2] = {0.0f, 0.0f};
		float tmass[2];
		V2 lfA[2], lfB[2];
		if (contact->frictionPersisted)
		{
			int confirmCount = 0;
			for (int j = 0; j < pointCount; ++j)
			{
				const s2amdManifoldPoint* mp = contact->points + j;
				V2 normalA = rotate(qA, v2(mp->frictionNormalA[0], mp->frictionNormalA[1]));
				V2 normalB = rotate(qB, v2(mp->frictionNormalB[0], mp->frictionNormalB[1]));
				float nn = dot(normalA, normalB);
				if (nn < 0.98f)
				{
					break;
				}
				lfA[j] = sub(v2(mp->frictionAnchorA[0], mp->frictionAnchorA[1]), lcA);
				lfB[j] = sub(v2(mp->frictionAnchorB[0], mp->frictionAnchorB[1]), lcB);
				V2 rAf = rotate(qA, lfA[j]);
				V2 rBf = rotate(qB, lfB[j]);
				V2 offset = add(sub(cB, cA), sub(rBf, rAf));
				float normalSeparation = dot(offset, normalA);
				if (S2_ABSF(normalSeparation) > 2.0f * S2_LINEAR_SLOP)
				{
					break;
				}
				tsep[j] = dot(sub(cB, cA), tangent);
				float rtA = cross(rAf, tangent);
				float rtB = cross(rBf, tangent);
				float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
				tmass[j] = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
				confirmCount += 1;
			}
			frictionConfirmed = confirmCount == pointCount;
			if (frictionConfirmed == false)
			{
				// points processed before the break keep their overwritten tangentMass /
				// tangentSeparation / friction anchors in the reference; the rebuild below overwrites
				// every point again, so nothing of the partial pass survives.
			}
		}
		if (frictionConfirmed == false)
		{
			for (int j = 0; j < pointCount; ++j)
			{
				s2amdManifoldPoint* mp = contact->points + j;
				float4 r0 = c.r0[j][k];
				V2 rA = v2(r0.x, r0.y), rB = v2(r0.z, r0.w);
				V2 fnA = invRotate(qA, normal);
				V2 fnB = invRotate(qB, normal);
				mp->frictionNormalA[0] = fnA.x, mp->frictionNormalA[1] = fnA.y;
				mp->frictionNormalB[0] = fnB.x, mp->frictionNormalB[1] = fnB.y;
				mp->frictionAnchorA[0] = mp->localAnchorA[0], mp->frictionAnchorA[1] = mp->localAnchorA[1];
				mp->frictionAnchorB[0] = mp->localAnchorB[0], mp->frictionAnchorB[1] = mp->localAnchorB[1];
				float4 an = c.anchor[j][k];
				lfA[j] = v2(an.x, an.y);
				lfB[j] = v2(an.z, an.w);
				tsep[j] = dot(sub(cB, cA), tangent);
				float rtA = cross(rA, tangent);
				float rtB = cross(rB, tangent);
				float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
				tmass[j] = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
			}
		}
		for (int j = 0; j < pointCount; ++j)
		{
			c.fanchor[j][k] = make_float4(lfA[j].x, lfA[j].y, lfB[j].x, lfB[j].y);
			float4 p = c.param[j][k];
			p.z = tmass[j];
			c.param[j][k] = p;
			float4 s = c.soft[j][k];
			s.w = tsep[j];
			c.soft[j][k] = s;
		}
		if (pointCount > 0) // the reference never visits a manifold without points (solve_tgs_sticky.c:331-348 gathers the others)
		{
			contact->frictionPersisted = 1;
		}
	}

	c.nf[k] = make_float4(normal.x, normal.y, friction, fromBits(bits));
}

// s2StoreContactImpulses (solve_common.c:396-410), the scaled XPBD variant (solve_xpbd.c:517-527)
// and s2ContactSolver_StoreImpulses (solve_pgs_ngs_block.c:660-677, reduced point count)
template <int KIND>
// blocks [0, contactBlocks): impulses -> wire contacts; then bodyBlocks blocks: SoA bodies -> wire bodies
// (packBodyOne); the remaining blocks zero the hand-off buffers of the persistent strip step (strip_kernel.hip: its
// tags restart at 1 every launch, so the buffers must be clean when the next step starts)
__global__ __launch_bounds__(S2_BLOCK) void storeImpulsesKernel(ContactView c, s2amdContact* wire, float scale, int contactBlocks, int bodyBlocks,
																BodyView bodies, s2amdBody* wireBodies, uint4* clear, int clearCount, const unsigned int* stepFailed,
																int finalizeMode, JointView jv, s2amdJoint* wireJoints, int jointBlocks, Stage4Args s4, int stage4Base,
																int stage4ShapeBlocks)
{
	// a persistent step whose hand-offs timed out leaves the wire arrays as they were: the host then repeats the step
	// on the multi-launch path (solver_step.cpp: doStep); the hand-off buffers are cleared either way
	const bool failed = stepFailed != nullptr && *stepFailed != 0u; // a device-memory word, written by the previous launch
	if ((int)blockIdx.x >= stage4Base)
	{
		// Stage 4 of the world step (src/world.c:259-301; refit_ops.h) in the same launch: the poses come from the SoA records the step
		// has finished (what the body blocks above copy into the wire bodies: the same values), so nothing here waits for them
		if (failed)
		{
			return; // (the step will be repeated: nothing moved, the applied forces are still to be consumed)
		}
		const int blk = (int)blockIdx.x - stage4Base;
		if (blk >= stage4ShapeBlocks)
		{
			const int i = (blk - stage4ShapeBlocks) * (int)blockDim.x + (int)threadIdx.x;
			if (i < bodies.capacity)
			{
				stage4BodyOne(wireBodies, i, s4.origins, &bodies);
			}
			return;
		}
		const int si = blk * (int)blockDim.x + (int)threadIdx.x;
		int enlarged = 0;
		if (si < s4.shapeCapacity)
		{
			enlarged = stage4ShapeOne(wireBodies, bodies.capacity, s4.shapes + si, &bodies);
		}
		const unsigned long long m = __ballot(enlarged != 0);
		if ((threadIdx.x & 63) == 0 && m != 0ull)
		{
			atomicAdd(s4.summary + 4, __popcll(m));
		}
		return;
	}
	if ((int)blockIdx.x >= contactBlocks + bodyBlocks && (int)blockIdx.x < contactBlocks + bodyBlocks + jointBlocks)
	{
		// joint impulses -> wire joints (joint.c: the persistent members of s2RevoluteJoint / s2MouseJoint), as storeJointsKernel
		const int k = ((int)blockIdx.x - contactBlocks - bodyBlocks) * (int)blockDim.x + (int)threadIdx.x;
		if (!failed && k < jv.count)
		{
			s2amdJoint* w = wireJoints + jv.jointIndex[k];
			float2 im = jv.impulse[k];
			float4 ax = jv.axial[k];
			w->impulse[0] = im.x, w->impulse[1] = im.y;
			w->motorImpulse = ax.x;
			if (w->type == S2AMD_JOINT_REVOLUTE)
			{
				w->lowerImpulse = ax.y;
				w->upperImpulse = ax.z;
			}
		}
		return;
	}
	if ((int)blockIdx.x >= contactBlocks + bodyBlocks + jointBlocks)
	{
		int i = ((int)blockIdx.x - contactBlocks - bodyBlocks - jointBlocks) * (int)blockDim.x + (int)threadIdx.x;
		if (i < clearCount)
		{
			clear[i] = make_uint4(0u, 0u, 0u, 0u);
		}
		return;
	}
	if (failed)
	{
		return;
	}
	if ((int)blockIdx.x >= contactBlocks)
	{
		const int i = ((int)blockIdx.x - contactBlocks) * (int)blockDim.x + (int)threadIdx.x;
		if (finalizeMode >= 0 && i < bodies.capacity && (bodies.flags[i] & S2F_IN_GROUP) == 0)
		{
			// s2FinalizePositions (solve_common.c:70) of the bodies no LDS group or strip owns, when it is the step's last
			// body stage: folded into the write-back instead of a launch of its own (finalizePositionsKernel)
			GlobalBodies gb{bodies.vel, bodies.dq};
			finalizePositionsOne(gb, i, bodies, i, finalizeMode, true);
		}
		packBodyOne(bodies, wireBodies, i);
		return;
	}
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= c.skipBegin)
	{
		k += c.skipEnd - c.skipBegin; // the contact blocks cover [0, skipBegin) and [skipEnd, count) back to back
	}
	if (k >= c.count)
	{
		return;
	}
	const int slot = c.contactIndex[k];
	if (slot < 0)
	{
		return; // a free position of the sweep order
	}
	s2amdContact* contact = wire + slot;
	int pointCount = KIND == STORE_BLOCK ? (int)asBits(c.blockK[k].w) : (int)(asBits(c.nf[k].w) & 0xffu);
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			float2 imp = c.impulse[j][k];
			if (KIND == STORE_SCALED)
			{
				contact->points[j].normalImpulse = imp.x * scale;
				contact->points[j].tangentImpulse = imp.y * scale;
			}
			else
			{
				contact->points[j].normalImpulse = imp.x;
				contact->points[j].tangentImpulse = imp.y;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------
// sweep kernels: one thread per constraint of one colour batch [begin, end), bodies in HBM/L2
// ---------------------------------------------------------------------------------------------
#define S2_SWEEP_BODY(CALL)                                                                                                      \
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;                                                                       \
	if (k < end)                                                                                                                 \
	{                                                                                                                            \
		GlobalBodies gb{b.vel, b.dq};                                                                                            \
		CALL;                                                                                                                    \
	}

// Body-centric warm start.  s2WarmStartContacts (solve_common.c:276-326) only ADDS terms that do not
// depend on any velocity -- P comes from the stored impulses, the lever arm from the body's own
// rotation -- so the value a body ends up with is v + t1 + t2 + ... over its incident contact points
// in sweep order.  One thread per body walks its incidence list (ascending sweep position) and
// performs exactly those additions in exactly that order: bit-identical to the coloured sweep, but
// ONE launch instead of one per colour, with no write conflicts at all.  Optionally preceded by
// s2IntegrateVelocities for the same body (the two stages are adjacent in every sub-stepping driver).
S2_DEV float laneValue(float v, int lane)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void warmStartBodiesKernel(ContactView c, BodyView b, const int2* adjRange, const int* adjList,
																   int integrateFirst, int bodyBlocks, const int* heavy)
{
	GlobalBodies gb{b.vel, b.dq};
	if ((int)blockIdx.x >= bodyBlocks)
	{
		// ---- heavy bodies (more than S2_HEAVY_DEGREE incident constraints: a drum, a platform): one wave each.  Every
		// lane loads ONE list entry and computes that entry's terms -- they depend on the stored impulses and on the
		// body's own rotation only --, then the additions run in list order on values broadcast from lane u, exactly
		// the additions of the one-thread walk below:  w +- iv*cross(r,P)  ==  w + (+-t),   v + (+-m)*P  ==  v + prod. ----
		const int lane = (int)threadIdx.x & 63;
		const int h = ((int)blockIdx.x - bodyBlocks) * (S2_BLOCK / 64) + ((int)threadIdx.x >> 6);
		if (h >= heavy[0]) // heavy[0] = count, heavy[1..] = body slots (see jacobiApplyKernel)
		{
			return;
		}
		const int i = heavy[1 + h];
		if ((b.flags[i] & S2F_IN_GROUP) != 0)
		{
			return;
		}
		if (integrateFirst)
		{
			integrateVelocitiesOne(gb, i, b, i); // every lane: the same loads, the same result, the same store; each lane reads back its own
		}
		const int2 range = adjRange[i];
		const int e0 = range.x, e1 = range.x + range.y;
		float4 v4 = b.vel[i];
		V2 v = v2(v4.x, v4.y);
		float w = v4.z;
		Rot q;
		q.s = 0.0f, q.c = 1.0f;
		if (KIND == WARM_CURRENT)
		{
			float4 d = b.dq[i];
			q.s = d.z, q.c = d.w;
		}
		for (int base = e0; base < e1; base += 64)
		{
			const int e = base + lane < e1 ? base + lane : e0;
			const int key = adjList[e];
			const int k = key >> 1;
			const bool sideB = (key & 1) != 0;
			const float4 nf = c.nf[k];
			const float4 ms = c.mass[k];
			V2 normal = v2(nf.x, nf.y);
			V2 tangent = KIND == WARM_BLOCK ? crossVS(normal, 1.0f) : rightPerp(normal);
			int pointCount = KIND == WARM_BLOCK ? (int)asBits(c.blockK[k].w) : (int)(asBits(nf.w) & 0xffu);
			const float m = sideB ? ms.z : ms.x;
			const float iv = sideB ? ms.w : ms.y;
			float tw[2], px[2], py[2];
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				float4 arm = KIND == WARM_CURRENT ? c.anchor[j][k] : c.r0[j][k];
				float2 imp = c.impulse[j][k];
				V2 l = sideB ? v2(arm.z, arm.w) : v2(arm.x, arm.y);
				V2 r = KIND == WARM_CURRENT ? rotate(q, l) : l;
				V2 P = add(mulSV(imp.x, normal), mulSV(imp.y, tangent));
				float t = iv * cross(r, P);
				tw[j] = sideB ? t : -t;
				V2 prod = mulSV(sideB ? m : -m, P);
				px[j] = prod.x, py[j] = prod.y;
			}
			const int left = __builtin_amdgcn_readfirstlane(e1 - base); // wave-uniform: a scalar loop, unrolled
			const int n = left < 64 ? left : 64;
			auto addEntry = [&](int u) {
				const int pc = __builtin_amdgcn_readlane(pointCount, u);
#pragma unroll
				for (int j = 0; j < 2; ++j)
				{
					const float wn = w + laneValue(tw[j], u);
					const V2 vn = add(v, v2(laneValue(px[j], u), laneValue(py[j], u)));
					w = j < pc ? wn : w;
					v = j < pc ? vn : v;
				}
			};
			int u = 0;
			for (; u + 4 <= n; u += 4)
			{
#pragma unroll
				for (int t = 0; t < 4; ++t)
				{
					addEntry(u + t);
				}
			}
			for (; u < n; ++u)
			{
				addEntry(u);
			}
		}
		if (lane == 0 && e0 != e1)
		{
			b.vel[i] = make_float4(v.x, v.y, w, 0.0f);
		}
		return;
	}
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= b.capacity || (b.flags[i] & S2F_IN_GROUP) != 0)
	{
		return;
	}
	const int2 range = adjRange[i];
	int e0 = range.x, e1 = range.x + range.y;
	if (e1 - e0 > S2_HEAVY_DEGREE)
	{
		return; // a wave of the heavy blocks walks this one (and integrates it first)
	}
	if (integrateFirst)
	{
		integrateVelocitiesOne(gb, i, b, i);
	}
	if (e0 == e1)
	{
		return;
	}
	float4 v4 = b.vel[i];
	V2 v = v2(v4.x, v4.y);
	float w = v4.z;
	Rot q;
	if (KIND == WARM_CURRENT)
	{
		float4 d = b.dq[i];
		q.s = d.z, q.c = d.w;
	}
	for (int e = e0; e < e1; ++e)
	{
		int key = adjList[e];
		int k = key >> 1;
		bool sideB = (key & 1) != 0;
		float4 nf = c.nf[k];
		float4 ms = c.mass[k];
		V2 normal = v2(nf.x, nf.y);
		V2 tangent = KIND == WARM_BLOCK ? crossVS(normal, 1.0f) : rightPerp(normal);
		int pointCount = KIND == WARM_BLOCK ? (int)asBits(c.blockK[k].w) : (int)(asBits(nf.w) & 0xffu);
		float m = sideB ? ms.z : ms.x;
		float iv = sideB ? ms.w : ms.y;
		for (int j = 0; j < pointCount; ++j)
		{
			float4 arm = KIND == WARM_CURRENT ? c.anchor[j][k] : c.r0[j][k];
			float2 imp = c.impulse[j][k];
			V2 l = sideB ? v2(arm.z, arm.w) : v2(arm.x, arm.y);
			V2 r = KIND == WARM_CURRENT ? rotate(q, l) : l;
			V2 P = add(mulSV(imp.x, normal), mulSV(imp.y, tangent));
			if (sideB)
			{
				w += iv * cross(r, P);
				v = mulAdd(v, m, P);
			}
			else
			{
				w -= iv * cross(r, P);
				v = mulAdd(v, -m, P);
			}
		}
	}
	b.vel[i] = make_float4(v.x, v.y, w, 0.0f);
}

// message-passing variants of the velocity-level sweeps (bodies read from per-constraint copies)
#define S2_SWEEP_BODY_MSG(CALL)                                                                                                  \
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;                                                                       \
	if (k < end)                                                                                                                 \
	{                                                                                                                            \
		MsgBodies gb{m.vel, m.dq, m.next};                                                                                       \
		CALL;                                                                                                                    \
	}

template <int KIND> __global__ __launch_bounds__(S2_BLOCK) void warmStartContactsMsgKernel(ContactView c, MsgView m, int begin, int end)
{
	S2_SWEEP_BODY_MSG(warmStartContactsOne<KIND>(c, gb, k))
}
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveContactsSoftMsgKernel(ContactView c, MsgView m, int begin, int end, float inv_h, int useBias)
{
	S2_SWEEP_BODY_MSG(solveContactsSoftOne<KIND>(c, gb, inv_h, useBias, k))
}
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveContactsRigidMsgKernel(ContactView c, MsgView m, int begin, int end, float inv_h)
{
	S2_SWEEP_BODY_MSG(solveContactsRigidOne<KIND>(c, gb, inv_h, k))
}
__global__ __launch_bounds__(S2_BLOCK) void solveContactsStickyMsgKernel(ContactView c, MsgView m, s2amdContact* wire, int begin, int end,
																		 float inv_h, int useBias)
{
	S2_SWEEP_BODY_MSG(solveContactsStickyOne(c, gb, wire, inv_h, useBias, k))
}

// fills both copies of every constraint of the global part from the body arrays (after prepare)
__global__ __launch_bounds__(S2_BLOCK) void fillMessageSlotsKernel(ContactView c, BodyView b, MsgView m, int count)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= count)
	{
		return;
	}
	int2 bd = c.bodies[k];
	m.vel[2 * k] = b.vel[bd.x];
	m.dq[2 * k] = b.dq[bd.x];
	m.vel[2 * k + 1] = b.vel[bd.y];
	m.dq[2 * k + 1] = b.dq[bd.y];
}

template <int KIND> __global__ __launch_bounds__(S2_BLOCK) void warmStartContactsKernel(ContactView c, BodyView b, int begin, int end)
{
	S2_SWEEP_BODY(warmStartContactsOne<KIND>(c, gb, k))
}
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveContactsSoftKernel(ContactView c, BodyView b, int begin, int end, float inv_h, int useBias)
{
	S2_SWEEP_BODY(solveContactsSoftOne<KIND>(c, gb, inv_h, useBias, k))
}
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveContactsRigidKernel(ContactView c, BodyView b, int begin, int end, float inv_h)
{
	S2_SWEEP_BODY(solveContactsRigidOne<KIND>(c, gb, inv_h, k))
}
__global__ __launch_bounds__(S2_BLOCK) void solveContactsStickyKernel(ContactView c, BodyView b, s2amdContact* wire, int begin, int end,
																	  float inv_h, int useBias)
{
	S2_SWEEP_BODY(solveContactsStickyOne(c, gb, wire, inv_h, useBias, k))
}
__global__ __launch_bounds__(S2_BLOCK) void solveContactsNGSKernel(ContactView c, BodyView b, int begin, int end)
{
	S2_SWEEP_BODY(solveContactsNGSOne(c, gb, k))
}
__global__ __launch_bounds__(S2_BLOCK) void xpbdContactPositionsKernel(ContactView c, BodyView b, int begin, int end, float hh)
{
	S2_SWEEP_BODY(xpbdContactPositionsOne(c, gb, hh, k))
}
__global__ __launch_bounds__(S2_BLOCK) void xpbdContactVelocitiesKernel(ContactView c, BodyView b, int begin, int end, float hh)
{
	S2_SWEEP_BODY(xpbdContactVelocitiesOne(c, gb, hh, k))
}
__global__ __launch_bounds__(S2_BLOCK) void blockSolveVelocityKernel(ContactView c, BodyView b, int begin, int end)
{
	S2_SWEEP_BODY(blockSolveVelocityOne(c, gb, k))
}
__global__ __launch_bounds__(S2_BLOCK) void blockSolvePositionKernel(ContactView c, BodyView b, int begin, int end)
{
	S2_SWEEP_BODY(blockSolvePositionOne(c, gb, k))
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline dim3 gridFor(int n)
{
	return dim3((unsigned)((n + S2_BLOCK - 1) / S2_BLOCK));
}

#define S2_LAUNCH_SWEEP(KERNEL, ...)                                                                                             \
	do                                                                                                                           \
	{                                                                                                                            \
		if (end > begin)                                                                                                         \
		{                                                                                                                        \
			KERNEL<<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(__VA_ARGS__);                                                 \
		}                                                                                                                        \
	} while (0)

bool launchPrepareContacts(hipStream_t s, int kind, const ContactView& c, const BodyView& b, s2amdContact* wire, const s2amdBody* wireBodies,
						   const StepConsts& sc, float h, float hertz, int posSolver, const uint32_t* hostFlags, bool unpackToo, float unpackH,
						   int contactCapacity, const int* gatherIndex, const JointPrepArgs* joints)
{
	if (c.count <= 0)
	{
		return false;
	}
	JointPrepArgs jp{};
	if (joints != nullptr)
	{
		jp = *joints;
	}
	const int contactBlocks = (c.count - (c.skipEnd - c.skipBegin) + S2_BLOCK - 1) / S2_BLOCK;
	const int bodyBlocks = unpackToo && b.capacity > 0 ? (b.capacity + S2_BLOCK - 1) / S2_BLOCK : 0;
	const int indexBlocks = unpackToo && gatherIndex && contactCapacity > 0 ? (contactCapacity + S2_BLOCK - 1) / S2_BLOCK : 0;
	if (contactBlocks + bodyBlocks + indexBlocks == 0)
	{
		return false; // every position is the resident-island kernel's own
	}
	dim3 g((unsigned)(contactBlocks + bodyBlocks + indexBlocks + jp.blocks)), t(S2_BLOCK);
	switch (kind)
	{
		case PREP_PGS:
			prepareContactsKernel<PREP_PGS><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver, hostFlags, contactBlocks, bodyBlocks, unpackH, contactCapacity, gatherIndex, jp);
			break;
		case PREP_SOFT:
			prepareContactsKernel<PREP_SOFT><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver, hostFlags, contactBlocks, bodyBlocks, unpackH, contactCapacity, gatherIndex, jp);
			break;
		case PREP_TGS:
			prepareContactsKernel<PREP_TGS><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver, hostFlags, contactBlocks, bodyBlocks, unpackH, contactCapacity, gatherIndex, jp);
			break;
		case PREP_STICKY:
			prepareContactsKernel<PREP_STICKY><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver, hostFlags, contactBlocks, bodyBlocks, unpackH, contactCapacity, gatherIndex, jp);
			break;
		case PREP_XPBD:
			prepareContactsKernel<PREP_XPBD><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver, hostFlags, contactBlocks, bodyBlocks, unpackH, contactCapacity, gatherIndex, jp);
			break;
		case PREP_BLOCK:
			prepareContactsKernel<PREP_BLOCK><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver, hostFlags, contactBlocks, bodyBlocks, unpackH, contactCapacity, gatherIndex, jp);
			break;
	}
	return true; // (launched: the joint blocks, if any were asked for, rode along)
}

// The overflow contacts of a sliced step (solver_executor.h: runPersistentSliced): positions [begin, end) swept ONE AFTER THE OTHER by one
// lane -- they may share a body (the ball that touches boxes of two strips) --, free positions skipped.  The same per-constraint
// functions as the colour batches, on the bodies in HBM.
template <int KIND> __global__ void overflowWarmKernel(ContactView c, BodyView b, int begin, int end)
{
	if (threadIdx.x == 0 && blockIdx.x == 0)
	{
		GlobalBodies gb{b.vel, b.dq};
		for (int k = begin; k < end; ++k)
		{
			if (c.contactIndex[k] >= 0)
			{
				warmStartContactsOne<KIND>(c, gb, k);
			}
		}
	}
}
template <int KIND> __global__ void overflowSoftKernel(ContactView c, BodyView b, int begin, int end, float inv_h, int useBias)
{
	if (threadIdx.x == 0 && blockIdx.x == 0)
	{
		GlobalBodies gb{b.vel, b.dq};
		for (int k = begin; k < end; ++k)
		{
			if (c.contactIndex[k] >= 0)
			{
				solveContactsSoftOne<KIND>(c, gb, inv_h, useBias, k);
			}
		}
	}
}

// one sweep op (s2WarmStartContacts / the soft solve of the three soft drivers) over the overflow positions, sequentially
void launchOverflowSweep(hipStream_t s, const Op& o, const ContactView& c, const BodyView& b, int begin, int end)
{
	const dim3 one(1), lanes(64);
	if (o.code == OP_WARM)
	{
		if (o.kind == WARM_FIXED)
		{
			overflowWarmKernel<WARM_FIXED><<<one, lanes, 0, s>>>(c, b, begin, end);
		}
		else
		{
			overflowWarmKernel<WARM_CURRENT><<<one, lanes, 0, s>>>(c, b, begin, end);
		}
	}
	else if (o.code == OP_SOLVE_SOFT)
	{
		switch (o.kind)
		{
			case SOFT_TGS:
				overflowSoftKernel<SOFT_TGS><<<one, lanes, 0, s>>>(c, b, begin, end, o.inv_h, o.useBias);
				break;
			case SOFT_PGS:
				overflowSoftKernel<SOFT_PGS><<<one, lanes, 0, s>>>(c, b, begin, end, o.inv_h, o.useBias);
				break;
			case SOFT_FIXED:
				overflowSoftKernel<SOFT_FIXED><<<one, lanes, 0, s>>>(c, b, begin, end, o.inv_h, o.useBias);
				break;
			default:
				break;
		}
	}
}

void launchWarmStartContacts(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end)
{
	switch (kind)
	{
		case WARM_CURRENT:
			S2_LAUNCH_SWEEP(warmStartContactsKernel<WARM_CURRENT>, c, b, begin, end);
			break;
		case WARM_FIXED:
			S2_LAUNCH_SWEEP(warmStartContactsKernel<WARM_FIXED>, c, b, begin, end);
			break;
		case WARM_BLOCK:
			S2_LAUNCH_SWEEP(warmStartContactsKernel<WARM_BLOCK>, c, b, begin, end);
			break;
	}
}

void launchSolveContactsSoft(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end, float inv_h, int useBias)
{
	switch (kind)
	{
		case SOFT_TGS:
			S2_LAUNCH_SWEEP(solveContactsSoftKernel<SOFT_TGS>, c, b, begin, end,
							inv_h, useBias);
			break;
		case SOFT_PGS:
			S2_LAUNCH_SWEEP(solveContactsSoftKernel<SOFT_PGS>, c, b, begin, end,
							inv_h, useBias);
			break;
		case SOFT_JACOBI:
			S2_LAUNCH_SWEEP(solveContactsSoftKernel<SOFT_JACOBI>, c, b, begin,
							end, inv_h, useBias);
			break;
		case SOFT_FIXED:
			S2_LAUNCH_SWEEP(solveContactsSoftKernel<SOFT_FIXED>, c, b, begin, end,
							inv_h, useBias);
			break;
	}
}

void launchSolveContactsRigid(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end, float inv_h)
{
	switch (kind)
	{
		case RIGID_BAUMGARTE:
			S2_LAUNCH_SWEEP(solveContactsRigidKernel<RIGID_BAUMGARTE>, c, b,
							begin, end, inv_h);
			break;
		case RIGID_PGS:
			S2_LAUNCH_SWEEP(solveContactsRigidKernel<RIGID_PGS>, c, b, begin, end,
							inv_h);
			break;
		case RIGID_TGS:
			S2_LAUNCH_SWEEP(solveContactsRigidKernel<RIGID_TGS>, c, b, begin, end,
							inv_h);
			break;
	}
}

void launchSolveContactsSticky(hipStream_t s, const ContactView& c, const BodyView& b, s2amdContact* wire, int begin, int end, float inv_h,
							   int useBias)
{
	S2_LAUNCH_SWEEP(solveContactsStickyKernel, c, b, wire, begin, end, inv_h, useBias);
}

void launchSolveContactsNGS(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end)
{
	S2_LAUNCH_SWEEP(solveContactsNGSKernel, c, b, begin, end);
}

void launchXpbdContactPositions(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end, float h)
{
	S2_LAUNCH_SWEEP(xpbdContactPositionsKernel, c, b, begin, end, h);
}

void launchXpbdContactVelocities(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end, float h)
{
	S2_LAUNCH_SWEEP(xpbdContactVelocitiesKernel, c, b, begin, end, h);
}

void launchBlockSolveVelocity(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end)
{
	S2_LAUNCH_SWEEP(blockSolveVelocityKernel, c, b, begin, end);
}

void launchBlockSolvePosition(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end)
{
	S2_LAUNCH_SWEEP(blockSolvePositionKernel, c, b, begin, end);
}

bool launchStoreImpulses(hipStream_t s, int kind, const ContactView& c, s2amdContact* wire, float scale, const BodyView& bodies, s2amdBody* wireBodies,
						 void* clear, size_t clearBytes, const unsigned int* stepFailed, int finalizeMode, const JointView* joints, s2amdJoint* wireJoints,
						 const Stage4Args* stage4)
{
	// bodies == nullptr-capacity: plain store; otherwise the body write-back rides in the same launch
	const int contactBlocks = c.count > 0 ? (c.count - (c.skipEnd - c.skipBegin) + S2_BLOCK - 1) / S2_BLOCK : 0;
	const int bodyBlocks = wireBodies && bodies.capacity > 0 ? (bodies.capacity + S2_BLOCK - 1) / S2_BLOCK : 0;
	const int clearCount = clear ? (int)(clearBytes / sizeof(uint4)) : 0; // buffers are allocated in multiples of 256 bytes
	const int clearBlocks = (clearCount + S2_BLOCK - 1) / S2_BLOCK;
	const int jointBlocks = joints && wireJoints && joints->count > 0 ? (joints->count + S2_BLOCK - 1) / S2_BLOCK : 0;
	const JointView jv = joints ? *joints : JointView{};
	// (stage 4 rides along when the body write-back does and no s2FinalizePositions is folded into it: the SoA poses are final)
	Stage4Args s4{};
	int s4Shapes = 0, s4Bodies = 0;
	if (stage4 != nullptr && stage4->shapes != nullptr && bodyBlocks > 0 && finalizeMode < 0)
	{
		s4 = *stage4;
		s4Shapes = (s4.shapeCapacity + S2_BLOCK - 1) / S2_BLOCK, s4Bodies = bodyBlocks;
	}
	if (contactBlocks + bodyBlocks + jointBlocks + clearBlocks == 0)
	{
		return false;
	}
	const int stage4Base = contactBlocks + bodyBlocks + jointBlocks + clearBlocks;
	dim3 g((unsigned)(stage4Base + s4Shapes + s4Bodies)), t(S2_BLOCK);
	switch (kind)
	{
		case STORE_SCALED:
			storeImpulsesKernel<STORE_SCALED><<<g, t, 0, s>>>(c, wire, scale, contactBlocks, bodyBlocks, bodies, wireBodies, (uint4*)clear, clearCount, stepFailed, finalizeMode, jv, wireJoints, jointBlocks, s4, s4Shapes + s4Bodies > 0 ? stage4Base : 0x7fffffff, s4Shapes);
			break;
		case STORE_BLOCK:
			storeImpulsesKernel<STORE_BLOCK><<<g, t, 0, s>>>(c, wire, scale, contactBlocks, bodyBlocks, bodies, wireBodies, (uint4*)clear, clearCount, stepFailed, finalizeMode, jv, wireJoints, jointBlocks, s4, s4Shapes + s4Bodies > 0 ? stage4Base : 0x7fffffff, s4Shapes);
			break;
		default:
			storeImpulsesKernel<STORE_PLAIN><<<g, t, 0, s>>>(c, wire, scale, contactBlocks, bodyBlocks, bodies, wireBodies, (uint4*)clear, clearCount, stepFailed, finalizeMode, jv, wireJoints, jointBlocks, s4, s4Shapes + s4Bodies > 0 ? stage4Base : 0x7fffffff, s4Shapes);
			break;
	}
	return s4Shapes + s4Bodies > 0; // (stage 4 was carried)
}

// ---- message-passing launchers ----
void launchFillMessageSlots(hipStream_t s, const ContactView& c, const BodyView& b, const MsgView& m, int count)
{
	if (count > 0)
	{
		fillMessageSlotsKernel<<<gridFor(count), dim3(S2_BLOCK), 0, s>>>(c, b, m, count);
	}
}

void launchWarmStartContactsMsg(hipStream_t s, int kind, const ContactView& c, const MsgView& m, int begin, int end)
{
	switch (kind)
	{
		case WARM_CURRENT:
			S2_LAUNCH_SWEEP(warmStartContactsMsgKernel<WARM_CURRENT>, c, m, begin, end);
			break;
		case WARM_FIXED:
			S2_LAUNCH_SWEEP(warmStartContactsMsgKernel<WARM_FIXED>, c, m, begin, end);
			break;
	}
}

void launchSolveContactsSoftMsg(hipStream_t s, int kind, const ContactView& c, const MsgView& m, int begin, int end, float inv_h, int useBias)
{
	switch (kind)
	{
		case SOFT_TGS:
			S2_LAUNCH_SWEEP(solveContactsSoftMsgKernel<SOFT_TGS>, c, m, begin, end, inv_h, useBias);
			break;
		case SOFT_PGS:
			S2_LAUNCH_SWEEP(solveContactsSoftMsgKernel<SOFT_PGS>, c, m, begin, end, inv_h, useBias);
			break;
		case SOFT_FIXED:
			S2_LAUNCH_SWEEP(solveContactsSoftMsgKernel<SOFT_FIXED>, c, m, begin, end, inv_h, useBias);
			break;
	}
}

void launchSolveContactsRigidMsg(hipStream_t s, int kind, const ContactView& c, const MsgView& m, int begin, int end, float inv_h)
{
	switch (kind)
	{
		case RIGID_BAUMGARTE:
			S2_LAUNCH_SWEEP(solveContactsRigidMsgKernel<RIGID_BAUMGARTE>, c, m, begin, end, inv_h);
			break;
	}
}

void launchSolveContactsStickyMsg(hipStream_t s, const ContactView& c, const MsgView& m, s2amdContact* wire, int begin, int end, float inv_h,
								  int useBias)
{
	S2_LAUNCH_SWEEP(solveContactsStickyMsgKernel, c, m, wire, begin, end, inv_h, useBias);
}

void launchWarmStartBodies(hipStream_t s, int kind, const ContactView& c, const BodyView& b, const int2* adjRange, const int* adjList,
						   int integrateFirst, const int* heavy, int heavyCapacity)
{
	if (b.capacity <= 0)
	{
		return;
	}
	const int bodyBlocks = (b.capacity + S2_BLOCK - 1) / S2_BLOCK, heavyBlocks = (heavyCapacity + S2_BLOCK / 64 - 1) / (S2_BLOCK / 64);
	dim3 g((unsigned)(bodyBlocks + heavyBlocks)), t(S2_BLOCK);
	switch (kind)
	{
		case WARM_CURRENT:
			warmStartBodiesKernel<WARM_CURRENT><<<g, t, 0, s>>>(c, b, adjRange, adjList, integrateFirst, bodyBlocks, heavy);
			break;
		case WARM_FIXED:
			warmStartBodiesKernel<WARM_FIXED><<<g, t, 0, s>>>(c, b, adjRange, adjList, integrateFirst, bodyBlocks, heavy);
			break;
		case WARM_BLOCK:
			warmStartBodiesKernel<WARM_BLOCK><<<g, t, 0, s>>>(c, b, adjRange, adjList, integrateFirst, bodyBlocks, heavy);
			break;
	}
}

S2_DEFINE_WARM(contact_kernels)
