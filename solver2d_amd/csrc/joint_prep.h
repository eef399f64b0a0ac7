// The joints' preparation as a per-joint device function: joint_kernels.hip's own kernel and the joint blocks of the prologue launches
// (contact_kernels.hip, body_kernels.hip) call it.
#pragma once

#include "constraint_ops.h"

// s2PrepareJoint (joint.c:297-312), s2PrepareJoint_Soft (:372-387), s2PrepareJoint_XPBD (:432-447)
//   revolute: s2PrepareRevolute revolute_joint.c:30-105, _Soft :421-506, _XPBD :792-823
//   mouse:    s2PrepareMouse mouse_joint.c:31-83 (all three dispatchers)
// Everything a joint's preparation reads of a body comes from the WIRE records and the host's write flags -- what unpackBodyOne
// (body_ops.h) copies into the SoA arrays, bit for bit -- so that it can run in the same launch as that unpack (the prologue:
// contact_kernels.hip prepareContactsKernel, body_kernels.hip unpackBodiesKernel) instead of behind it.
template <int KIND>
S2_DEV void prepareJointOne(const JointView& jv, const uint32_t* hostFlags, const s2amdJoint* wire, const s2amdBody* wireBodies, const StepConsts& sc, float h,
							float hertz, int warmStart, int posSolver, int k)
{
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
	if (hostFlags[ib] & wbit)
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
		Rot qB;
		qB.s = wb->rot[0], qB.c = wb->rot[1];
		V2 rB = rotate(qB, lB);
		M22 K;
		K.cx.x = mB + iB * rB.y * rB.y;
		K.cx.y = -iB * rB.x * rB.y;
		K.cy.x = K.cx.y;
		K.cy.y = mB + iB * rB.x * rB.x;
		pivotMass = inverse22(K);
		centerDiff0 = sub(v2(wb->position[0], wb->position[1]), v2(w->targetA[0], w->targetA[1]));
		bodyI = wb->I;
	}
	else
	{
		if (hostFlags[ia] & wbit)
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
		centerDiff0 = sub(v2(wb->position[0], wb->position[1]), v2(wa->position[0], wa->position[1]));

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
			Rot qA, qB;
			qA.s = wa->rot[0], qA.c = wa->rot[1];
			qB.s = wb->rot[0], qB.c = wb->rot[1];
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


S2_DEV void prepareJointsBlock(const JointPrepArgs& a, const uint32_t* hostFlags, const s2amdBody* wireBodies, const StepConsts& sc, int posSolver, int block)
{
	const int k = block * (int)blockDim.x + (int)threadIdx.x;
	if (a.kind == JPREP_PLAIN)
	{
		prepareJointOne<JPREP_PLAIN>(a.jv, hostFlags, a.wire, wireBodies, sc, a.h, a.hertz, a.warmStart, posSolver, k);
	}
	else if (a.kind == JPREP_SOFT)
	{
		prepareJointOne<JPREP_SOFT>(a.jv, hostFlags, a.wire, wireBodies, sc, a.h, a.hertz, a.warmStart, posSolver, k);
	}
	else
	{
		prepareJointOne<JPREP_XPBD>(a.jv, hostFlags, a.wire, wireBodies, sc, a.h, a.hertz, a.warmStart, posSolver, k);
	}
}
