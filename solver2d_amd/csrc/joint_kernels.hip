// Joint kernels: revolute (src/revolute_joint.c) and mouse (src/mouse_joint.c) joints, dispatched
// like src/joint.c:294-465.  One thread per joint; joints are stored SoA in sweep (colour-major)
// order and a sweep kernel is launched per colour batch, exactly like the contact kernels.  The
// per-joint arithmetic lives in constraint_ops.h (shared with the LDS group kernel).

#include "constraint_ops.h"

#define S2_BLOCK 256

// s2PrepareJoint (joint.c:297-312), s2PrepareJoint_Soft (:372-387), s2PrepareJoint_XPBD (:432-447)
//   revolute: s2PrepareRevolute revolute_joint.c:30-105, _Soft :421-506, _XPBD :792-823
//   mouse:    s2PrepareMouse mouse_joint.c:31-83 (all three dispatchers)
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void prepareJointsKernel(JointView jv, BodyView b, const s2amdJoint* wire, const s2amdBody* wireBodies,
																StepConsts sc, float h, float hertz, int warmStart, int posSolver)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= jv.count)
	{
		return;
	}
	const s2amdJoint* w = wire + jv.jointIndex[k];
	int ia = w->bodyA, ib = w->bodyB;
	const s2amdBody* wa = wireBodies + ia;
	const s2amdBody* wb = wireBodies + ib;
	uint32_t wbit = posSolver ? S2F_WRITE_POS : S2F_WRITE_VEL;
	uint32_t flags = 0;
	if (b.flags[ib] & wbit)
	{
		flags |= S2J_WRITE_B;
	}

	V2 impulse = v2(w->impulse[0], w->impulse[1]);
	float motorImpulse = w->motorImpulse, lowerImpulse = w->lowerImpulse, upperImpulse = w->upperImpulse;
	float bodyI = 0.0f;
	V2 lA = v2(0.0f, 0.0f), lB;
	float mA = 0.0f, iA = 0.0f, mB, iB;
	M22 pivotMass;
	float biasC = 0.0f, massC = 0.0f, impC = 0.0f, axialMass = 0.0f;
	V2 centerDiff0;

	if (w->type == S2AMD_JOINT_MOUSE)
	{
		flags |= S2J_MOUSE;
		mB = wb->invMass, iB = wb->invI;
		lB = sub(v2(w->localOriginAnchorB[0], w->localOriginAnchorB[1]), v2(wb->localCenter[0], wb->localCenter[1]));
		{
			float hh = sc.h;
			float zeta = w->dampingRatio;
			float omega = 2.0f * S2_PI * w->hertz;
			softCoefficients(hh, zeta, omega, biasC, massC, impC);
		}
		Rot qB = loadRotOnly(b, ib);
		V2 rB = rotate(qB, lB);
		M22 K;
		K.cx.x = mB + iB * rB.y * rB.y;
		K.cx.y = -iB * rB.x * rB.y;
		K.cy.x = K.cx.y;
		K.cy.y = mB + iB * rB.x * rB.x;
		pivotMass = inverse22(K);
		float2 pb = b.pos[ib];
		centerDiff0 = sub(v2(pb.x, pb.y), v2(w->targetA[0], w->targetA[1]));
		bodyI = wb->I;
	}
	else
	{
		if (b.flags[ia] & wbit)
		{
			flags |= S2J_WRITE_A;
		}
		if (w->enableMotor)
		{
			flags |= S2J_ENABLE_MOTOR;
		}
		if (w->enableLimit)
		{
			flags |= S2J_ENABLE_LIMIT;
		}
		const float inertiaScale = 1.0f;
		lA = sub(v2(w->localOriginAnchorA[0], w->localOriginAnchorA[1]), v2(wa->localCenter[0], wa->localCenter[1]));
		mA = wa->invMass;
		iA = KIND == JPREP_PLAIN ? inertiaScale * wa->invI : wa->invI;
		lB = sub(v2(w->localOriginAnchorB[0], w->localOriginAnchorB[1]), v2(wb->localCenter[0], wb->localCenter[1]));
		mB = wb->invMass;
		iB = KIND == JPREP_PLAIN ? inertiaScale * wb->invI : wb->invI;
		float2 pa = b.pos[ia], pb = b.pos[ib];
		centerDiff0 = sub(v2(pb.x, pb.y), v2(pa.x, pa.y));

		if (KIND == JPREP_XPBD)
		{
			pivotMass.cx = v2(0.0f, 0.0f);
			pivotMass.cy = v2(0.0f, 0.0f);
			axialMass = 0.0f;
			impulse = v2(0.0f, 0.0f);
			lowerImpulse = 0.0f;
			upperImpulse = 0.0f;
			motorImpulse = 0.0f;
		}
		else
		{
			Rot qA = loadRotOnly(b, ia), qB = loadRotOnly(b, ib);
			V2 rA = rotate(qA, lA);
			V2 rB = rotate(qB, lB);
			pivotMass = inverse22(revoluteK(mA, mB, iA, iB, rA, rB));
			if (KIND == JPREP_SOFT)
			{
				const float zeta = 10.0f;
				float omega = 2.0f * S2_PI * hertz;
				softCoefficients(h, zeta, omega, biasC, massC, impC);
			}
			axialMass = iA + iB;
			bool fixedRotation;
			if (axialMass > 0.0f)
			{
				axialMass = 1.0f / axialMass;
				fixedRotation = false;
			}
			else
			{
				fixedRotation = true;
			}
			bool enableLimit = w->enableLimit != 0, enableMotor = w->enableMotor != 0;
			if (enableLimit == false || fixedRotation || warmStart == 0)
			{
				lowerImpulse = 0.0f;
				upperImpulse = 0.0f;
			}
			if (enableMotor == false || fixedRotation || warmStart == 0)
			{
				motorImpulse = 0.0f;
			}
			if (warmStart == 0)
			{
				impulse = v2(0.0f, 0.0f);
			}
		}
	}

	jv.bodies[k] = make_int2(ia, ib);
	jv.frame[k] = make_float4(lA.x, lA.y, lB.x, lB.y);
	jv.mass[k] = make_float4(mA, iA, mB, iB);
	jv.pivot[k] = make_float4(pivotMass.cx.x, pivotMass.cx.y, pivotMass.cy.x, pivotMass.cy.y);
	jv.soft[k] = make_float4(biasC, massC, impC, axialMass);
	jv.centerDiff0[k] = make_float2(centerDiff0.x, centerDiff0.y);
	jv.impulse[k] = make_float2(impulse.x, impulse.y);
	jv.axial[k] = make_float4(motorImpulse, lowerImpulse, upperImpulse, bodyI);
	jv.limits[k] = make_float4(w->referenceAngle, w->lowerAngle, w->upperAngle, w->maxMotorTorque);
	jv.misc[k] = make_float4(w->motorSpeed, fromBits(flags), w->hertz, w->dampingRatio);
}

template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveJointsKernel(JointView jv, BodyView b, int begin, int end, StepConsts sc, float h, float inv_h,
															  int useBias)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k < end)
	{
		GlobalBodies gb{b.vel, b.dq};
		solveJointsOne<KIND>(jv, gb, sc, h, inv_h, useBias, k);
	}
}

// stepFailed: see storeImpulsesKernel (contact_kernels.hip) -- a persistent step whose hand-offs timed out leaves EVERY wire
// array as it was, so that the repeated step warm starts from the same joint impulses
__global__ __launch_bounds__(S2_BLOCK) void storeJointsKernel(JointView jv, s2amdJoint* wire, const unsigned int* stepFailed)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= jv.count || (stepFailed != nullptr && *stepFailed != 0u))
	{
		return;
	}
	s2amdJoint* w = wire + jv.jointIndex[k];
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

static inline dim3 gridFor(int n)
{
	return dim3((unsigned)((n + S2_BLOCK - 1) / S2_BLOCK));
}

void launchPrepareJoints(hipStream_t s, int kind, const JointView& j, const BodyView& b, const s2amdJoint* wire, const s2amdBody* wireBodies,
						 const StepConsts& sc, float h, float hertz, int warmStart, int posSolver)
{
	if (j.count <= 0)
	{
		return;
	}
	dim3 g = gridFor(j.count), t(S2_BLOCK);
	switch (kind)
	{
		case JPREP_PLAIN:
			prepareJointsKernel<JPREP_PLAIN><<<g, t, 0, s>>>(j, b, wire, wireBodies, sc, h, hertz, warmStart, posSolver);
			break;
		case JPREP_SOFT:
			prepareJointsKernel<JPREP_SOFT><<<g, t, 0, s>>>(j, b, wire, wireBodies, sc, h, hertz, warmStart, posSolver);
			break;
		case JPREP_XPBD:
			prepareJointsKernel<JPREP_XPBD><<<g, t, 0, s>>>(j, b, wire, wireBodies, sc, h, hertz, warmStart, posSolver);
			break;
	}
}

#define S2_JLAUNCH(K) solveJointsKernel<K><<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(j, b, begin, end, sc, h, inv_h, useBias);

void launchSolveJoints(hipStream_t s, int kind, const JointView& j, const BodyView& b, int begin, int end, const StepConsts& sc, float h,
					   float inv_h, int useBias)
{
	if (end <= begin)
	{
		return;
	}
	switch (kind)
	{
		case JSOLVE_PLAIN:
			S2_JLAUNCH(JSOLVE_PLAIN)
			break;
		case JSOLVE_SOFT:
			S2_JLAUNCH(JSOLVE_SOFT)
			break;
		case JSOLVE_BAUMGARTE:
			S2_JLAUNCH(JSOLVE_BAUMGARTE)
			break;
		case JSOLVE_POSITION:
			S2_JLAUNCH(JSOLVE_POSITION)
			break;
		case JSOLVE_XPBD:
			S2_JLAUNCH(JSOLVE_XPBD)
			break;
		case JSOLVE_WARM:
			S2_JLAUNCH(JSOLVE_WARM)
			break;
	}
}

// Joint warm start, body-centric (the counterpart of warmStartBodiesKernel for contacts): s2WarmStartRevolute / s2WarmStartMouse
// (revolute_joint.c:107-150, mouse_joint.c:85-107) only ADD velocity-independent terms, so each body adds the terms of its
// incident joints in sweep order -- the same additions in the same order as the coloured sweep -- in ONE launch instead of
// one per joint colour.  adjRange[i] = {first entry, count} in adjList; an entry is (k << 1) | side.
__global__ __launch_bounds__(S2_BLOCK) void warmStartJointsBodiesKernel(JointView jv, BodyView b, const int2* adjRange, const int* adjList)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= b.capacity)
	{
		return;
	}
	const int2 range = adjRange[i];
	if (range.y <= 0)
	{
		return;
	}
	const float4 v0 = b.vel[i], d = b.dq[i];
	Rot q;
	q.s = d.z, q.c = d.w;
	V2 v = v2(v0.x, v0.y);
	float w = v0.z;
	bool wrote = false;
	for (int e = range.x; e < range.x + range.y; ++e)
	{
		const int key = adjList[e];
		const JState s = loadJoint<S2_IDX_GLOBAL>(jv, key >> 1);
		if (s.flags & S2J_MOUSE)
		{
			if ((key & 1) && (s.flags & S2J_WRITE_B))
			{
				V2 rB = rotate(q, s.lB);
				v = mulAdd(v, s.mB, s.impulse);
				w += s.iB * (cross(rB, s.impulse) + s.motorImpulse);
				wrote = true;
			}
			continue;
		}
		const float axialImpulse = s.motorImpulse + s.lowerImpulse - s.upperImpulse;
		const V2 P = s.impulse;
		if ((key & 1) == 0)
		{
			if (s.flags & S2J_WRITE_A)
			{
				V2 rA = rotate(q, s.lA);
				v = mulSub(v, s.mA, P);
				w -= s.iA * (cross(rA, P) + axialImpulse);
				wrote = true;
			}
		}
		else if (s.flags & S2J_WRITE_B)
		{
			V2 rB = rotate(q, s.lB);
			v = mulAdd(v, s.mB, P);
			w += s.iB * (cross(rB, P) + axialImpulse);
			wrote = true;
		}
	}
	if (wrote)
	{
		b.vel[i] = make_float4(v.x, v.y, w, 0.0f);
	}
}

void launchWarmStartJointsBodies(hipStream_t s, const JointView& j, const BodyView& b, const int2* adjRange, const int* adjList)
{
	if (b.capacity > 0)
	{
		warmStartJointsBodiesKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(j, b, adjRange, adjList);
	}
}

void launchStoreJoints(hipStream_t s, const JointView& j, s2amdJoint* wire, const unsigned int* stepFailed)
{
	if (j.count > 0)
	{
		storeJointsKernel<<<gridFor(j.count), dim3(S2_BLOCK), 0, s>>>(j, wire, stepFailed);
	}
}

S2_DEFINE_WARM(joint_kernels)
