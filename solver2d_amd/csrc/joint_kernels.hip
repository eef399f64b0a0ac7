// Joint kernels: revolute (src/revolute_joint.c) and mouse (src/mouse_joint.c) joints, dispatched
// like src/joint.c:294-465.  One thread per joint; joints are stored SoA in sweep (colour-major)
// order and a sweep kernel is launched per colour batch, exactly like the contact kernels.

#include "launch.h"
#include "s2_device.h"

#include "solver2d_amd.h"

#define S2_BLOCK 256

struct JState
{
	int ia, ib;
	uint32_t flags;
	V2 lA, lB;
	float mA, iA, mB, iB;
	M22 pivotMass;
	float biasCoefficient, massCoefficient, impulseCoefficient, axialMass;
	V2 centerDiff0;
	V2 impulse;
	float motorImpulse, lowerImpulse, upperImpulse, bodyI;
	float referenceAngle, lowerAngle, upperAngle, maxMotorTorque, motorSpeed;
};

S2_DEV Rot loadRotOnly(const BodyView& b, int i)
{
	float4 d = b.dq[i];
	Rot q;
	q.s = d.z, q.c = d.w;
	return q;
}

S2_DEV JState loadJoint(const JointView& j, int k)
{
	JState s;
	int2 bd = j.bodies[k];
	float4 fr = j.frame[k], ms = j.mass[k], pv = j.pivot[k], sf = j.soft[k], ax = j.axial[k], lm = j.limits[k], mc = j.misc[k];
	float2 cd = j.centerDiff0[k], im = j.impulse[k];
	s.ia = bd.x, s.ib = bd.y;
	s.flags = asBits(mc.y);
	s.lA = v2(fr.x, fr.y), s.lB = v2(fr.z, fr.w);
	s.mA = ms.x, s.iA = ms.y, s.mB = ms.z, s.iB = ms.w;
	s.pivotMass.cx = v2(pv.x, pv.y), s.pivotMass.cy = v2(pv.z, pv.w);
	s.biasCoefficient = sf.x, s.massCoefficient = sf.y, s.impulseCoefficient = sf.z, s.axialMass = sf.w;
	s.centerDiff0 = v2(cd.x, cd.y);
	s.impulse = v2(im.x, im.y);
	s.motorImpulse = ax.x, s.lowerImpulse = ax.y, s.upperImpulse = ax.z, s.bodyI = ax.w;
	s.referenceAngle = lm.x, s.lowerAngle = lm.y, s.upperAngle = lm.z, s.maxMotorTorque = lm.w;
	s.motorSpeed = mc.x;
	return s;
}

S2_DEV void storeJointImpulses(const JointView& j, int k, const JState& s)
{
	j.impulse[k] = make_float2(s.impulse.x, s.impulse.y);
	j.axial[k] = make_float4(s.motorImpulse, s.lowerImpulse, s.upperImpulse, s.bodyI);
}

S2_DEV M22 revoluteK(float mA, float mB, float iA, float iB, V2 rA, V2 rB)
{
	// revolute_joint.c:70-74, :461-465, :631-636, :768-773
	M22 K;
	K.cx.x = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
	K.cy.x = -rA.y * rA.x * iA - rB.y * rB.x * iB;
	K.cx.y = K.cy.x;
	K.cy.y = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
	return K;
}

S2_DEV void softCoefficients(float h, float zeta, float omega, float& bias, float& mass, float& impulse)
{
	bias = omega / (2.0f * zeta + h * omega);
	float c = h * omega * (2.0f * zeta + h * omega);
	impulse = 1.0f / (1.0f + c);
	mass = c * impulse;
}

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

// s2SolveMouse: mouse_joint.c:109-167
S2_DEV void solveMouse(JState& s, const BodyView& b, float ctxH)
{
	float4 vb = b.vel[s.ib];
	float4 db = b.dq[s.ib];
	V2 vB = v2(vb.x, vb.y);
	float wB = vb.z;
	float mB = s.mB, iB = s.iB;
	{
		float h = ctxH;
		float zeta = 0.1f;
		float omega = 2.0f * S2_PI * 0.5f;
		float c = h * omega * (2.0f * zeta + h * omega);
		float impulseScale = 1.0f / (1.0f + c);
		float massScale = c * impulseScale;
		float impulse = -massScale * s.bodyI * wB - impulseScale * s.motorImpulse;
		s.motorImpulse += impulse;
		wB += iB * impulse;
	}
	{
		Rot qB;
		qB.s = db.z, qB.c = db.w;
		V2 rB = rotate(qB, s.lB);
		V2 Cdot = add(vB, crossSV(wB, rB));
		V2 dcB = v2(db.x, db.y);
		V2 separation = add(add(dcB, rB), s.centerDiff0);
		V2 bias = mulSV(s.biasCoefficient, separation);
		float massScale = s.massCoefficient;
		float impulseScale = s.impulseCoefficient;
		V2 bb = mulMV(s.pivotMass, add(Cdot, bias));
		V2 impulse;
		impulse.x = -massScale * bb.x - impulseScale * s.impulse.x;
		impulse.y = -massScale * bb.y - impulseScale * s.impulse.y;
		s.impulse.x += impulse.x;
		s.impulse.y += impulse.y;
		vB = mulAdd(vB, mB, impulse);
		wB += iB * cross(rB, impulse);
	}
	if (s.flags & S2J_WRITE_B)
	{
		b.vel[s.ib] = make_float4(vB.x, vB.y, wB, 0.0f);
	}
}

// motor row: revolute_joint.c:175-187, :526-538, :678-690
S2_DEV void revoluteMotor(JState& s, float h, float& wA, float& wB)
{
	float Cdot = wB - wA - s.motorSpeed;
	float impulse = -s.axialMass * Cdot;
	float oldImpulse = s.motorImpulse;
	float maxImpulse = h * s.maxMotorTorque;
	s.motorImpulse = S2_CLAMPF(s.motorImpulse + impulse, -maxImpulse, maxImpulse);
	impulse = s.motorImpulse - oldImpulse;
	wA -= s.iA * impulse;
	wB += s.iB * impulse;
}

template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveJointsKernel(JointView jv, BodyView b, int begin, int end, StepConsts sc, float h, float inv_h,
															  int useBias)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	JState s = loadJoint(jv, k);

	if (s.flags & S2J_MOUSE)
	{
		if (KIND == JSOLVE_WARM)
		{
			// s2WarmStartMouse: mouse_joint.c:85-107
			float4 vb = b.vel[s.ib];
			float4 db = b.dq[s.ib];
			Rot qB;
			qB.s = db.z, qB.c = db.w;
			V2 rB = rotate(qB, s.lB);
			V2 vB = v2(vb.x, vb.y);
			float wB = vb.z;
			vB = mulAdd(vB, s.mB, s.impulse);
			wB += s.iB * (cross(rB, s.impulse) + s.motorImpulse);
			if (s.flags & S2J_WRITE_B)
			{
				b.vel[s.ib] = make_float4(vB.x, vB.y, wB, 0.0f);
			}
		}
		else if (KIND == JSOLVE_PLAIN || KIND == JSOLVE_BAUMGARTE || KIND == JSOLVE_XPBD || (KIND == JSOLVE_SOFT && useBias))
		{
			// joint.c:342, :398-401, :418, :456
			solveMouse(s, b, sc.h);
			storeJointImpulses(jv, k, s);
		}
		return;
	}

	const bool writeA = (s.flags & S2J_WRITE_A) != 0, writeB = (s.flags & S2J_WRITE_B) != 0;
	const bool enableMotor = (s.flags & S2J_ENABLE_MOTOR) != 0, enableLimit = (s.flags & S2J_ENABLE_LIMIT) != 0;
	float mA = s.mA, iA = s.iA, mB = s.mB, iB = s.iB;

	if (KIND == JSOLVE_POSITION)
	{
		// s2SolveRevolutePosition: revolute_joint.c:305-419
		float4 da = b.dq[s.ia], db = b.dq[s.ib];
		V2 dcA = v2(da.x, da.y), dcB = v2(db.x, db.y);
		Rot qA, qB;
		qA.s = da.z, qA.c = da.w, qB.s = db.z, qB.c = db.w;
		bool fixedRotation = (iA + iB == 0.0f);
		if (enableLimit && fixedRotation == false)
		{
			float angle = relativeAngle(qB, qA) - s.referenceAngle;
			float C = 0.0f;
			if (S2_ABSF(s.upperAngle - s.lowerAngle) < 2.0f * S2_ANGULAR_SLOP)
			{
				C = S2_CLAMPF(angle - s.lowerAngle, -S2_MAX_ANGULAR_CORRECTION, S2_MAX_ANGULAR_CORRECTION);
			}
			else if (angle <= s.lowerAngle)
			{
				C = S2_CLAMPF(angle - s.lowerAngle + S2_ANGULAR_SLOP, -S2_MAX_ANGULAR_CORRECTION, 0.0f);
			}
			else if (angle >= s.upperAngle)
			{
				C = S2_CLAMPF(angle - s.upperAngle - S2_ANGULAR_SLOP, 0.0f, S2_MAX_ANGULAR_CORRECTION);
			}
			float limitImpulse = -s.axialMass * C;
			qA = integrateRot(qA, -iA * limitImpulse);
			qB = integrateRot(qB, iB * limitImpulse);
		}
		{
			V2 rA = rotate(qA, s.lA);
			V2 rB = rotate(qB, s.lB);
			V2 C = add(add(sub(dcB, dcA), sub(rB, rA)), s.centerDiff0);
			// fresh K with the operand order of revolute_joint.c:388-393
			M22 K;
			K.cx.x = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
			K.cx.y = -iA * rA.x * rA.y - iB * rB.x * rB.y;
			K.cy.x = K.cx.y;
			K.cy.y = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
			V2 impulse = solve22(K, neg(C));
			dcA = mulSub(dcA, mA, impulse);
			qA = integrateRot(qA, -iA * cross(rA, impulse));
			dcB = mulAdd(dcB, mB, impulse);
			qB = integrateRot(qB, iB * cross(rB, impulse));
		}
		if (writeA)
		{
			b.dq[s.ia] = make_float4(dcA.x, dcA.y, qA.s, qA.c);
		}
		if (writeB)
		{
			b.dq[s.ib] = make_float4(dcB.x, dcB.y, qB.s, qB.c);
		}
		return;
	}

	if (KIND == JSOLVE_XPBD)
	{
		// s2SolveRevolute_XPBD: revolute_joint.c:825-888
		const float compliance = 0.0f;
		float4 da = b.dq[s.ia], db = b.dq[s.ib];
		V2 dcA = v2(da.x, da.y), dcB = v2(db.x, db.y);
		Rot qA, qB;
		qA.s = da.z, qA.c = da.w, qB.s = db.z, qB.c = db.w;
		V2 rA = rotate(qA, s.lA);
		V2 rB = rotate(qB, s.lB);
		V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), s.centerDiff0);
		float c = length(separation);
		V2 n = normalize(separation);
		if (mA == 0.0f && mB == 0.0f)
		{
			return;
		}
		float rnA = cross(rA, n);
		float rnB = cross(rB, n);
		float kA = mA + iA * rnA * rnA;
		float kB = mB + iB * rnB * rnB;
		float lambda = -c / (kA + kB + compliance);
		V2 p = mulSV(lambda, n);
		dcA = mulSub(dcA, mA, p);
		qA = integrateRot(qA, -iA * cross(rA, p));
		dcB = mulAdd(dcB, mB, p);
		qB = integrateRot(qB, iB * cross(rB, p));
		if (writeA)
		{
			b.dq[s.ia] = make_float4(dcA.x, dcA.y, qA.s, qA.c);
		}
		if (writeB)
		{
			b.dq[s.ib] = make_float4(dcB.x, dcB.y, qB.s, qB.c);
		}
		return;
	}

	float4 va = b.vel[s.ia], vb = b.vel[s.ib];
	float4 da = b.dq[s.ia], db = b.dq[s.ib];
	V2 vA = v2(va.x, va.y), vB = v2(vb.x, vb.y);
	float wA = va.z, wB = vb.z;
	Rot qA, qB;
	qA.s = da.z, qA.c = da.w, qB.s = db.z, qB.c = db.w;

	if (KIND == JSOLVE_WARM)
	{
		// s2WarmStartRevolute: revolute_joint.c:107-150
		V2 rA = rotate(qA, s.lA);
		V2 rB = rotate(qB, s.lB);
		float axialImpulse = s.motorImpulse + s.lowerImpulse - s.upperImpulse;
		V2 P = s.impulse;
		vA = mulSub(vA, mA, P);
		wA -= iA * (cross(rA, P) + axialImpulse);
		vB = mulAdd(vB, mB, P);
		wB += iB * (cross(rB, P) + axialImpulse);
	}
	else
	{
		// s2SolveRevolute :152-303, s2SolveRevolute_Soft :508-657, s2SolveRevolute_Baumgarte :660-790
		bool fixedRotation = (iA + iB == 0.0f);
		if (enableMotor && fixedRotation == false)
		{
			revoluteMotor(s, h, wA, wB);
		}
		if (enableLimit && fixedRotation == false)
		{
			float jointAngle = relativeAngle(qB, qA) - s.referenceAngle;
			if (KIND == JSOLVE_PLAIN)
			{
				{
					float C = jointAngle - s.lowerAngle;
					float Cdot = wB - wA;
					float impulse = -s.axialMass * (Cdot + S2_MAXF(C, 0.0f) / h);
					float oldImpulse = s.lowerImpulse;
					s.lowerImpulse = S2_MAXF(s.lowerImpulse + impulse, 0.0f);
					impulse = s.lowerImpulse - oldImpulse;
					wA -= iA * impulse;
					wB += iB * impulse;
				}
				{
					float C = s.upperAngle - jointAngle;
					float Cdot = wA - wB;
					float impulse = -s.axialMass * (Cdot + S2_MAXF(C, 0.0f) / h);
					float oldImpulse = s.upperImpulse;
					s.upperImpulse = S2_MAXF(s.upperImpulse + impulse, 0.0f);
					impulse = s.upperImpulse - oldImpulse;
					wA += iA * impulse;
					wB -= iB * impulse;
				}
			}
			else
			{
				const bool soft = KIND == JSOLVE_SOFT;
				{
					float C = jointAngle - s.lowerAngle;
					float bias = 0.0f, massScale = 1.0f, impulseScale = 0.0f;
					if (C > 0.0f)
					{
						bias = C * inv_h;
					}
					else if (useBias)
					{
						if (soft)
						{
							bias = s.biasCoefficient * C;
							massScale = s.massCoefficient;
							impulseScale = s.impulseCoefficient;
						}
						else
						{
							bias = S2_BAUMGARTE * inv_h * C;
						}
					}
					float Cdot = wB - wA;
					float impulse = soft ? -s.axialMass * massScale * (Cdot + bias) - impulseScale * s.lowerImpulse : -s.axialMass * (Cdot + bias);
					float oldImpulse = s.lowerImpulse;
					s.lowerImpulse = S2_MAXF(s.lowerImpulse + impulse, 0.0f);
					impulse = s.lowerImpulse - oldImpulse;
					wA -= iA * impulse;
					wB += iB * impulse;
				}
				{
					float C = s.upperAngle - jointAngle;
					float bias = 0.0f, massScale = 1.0f, impulseScale = 0.0f;
					if (C > 0.0f)
					{
						bias = C * inv_h;
					}
					else if (useBias)
					{
						if (soft)
						{
							bias = s.biasCoefficient * C;
							massScale = s.massCoefficient;
							impulseScale = s.impulseCoefficient;
						}
						else
						{
							bias = S2_BAUMGARTE * inv_h * C;
						}
					}
					float Cdot = wA - wB;
					// the soft term reads lowerImpulse in the reference (revolute_joint.c:595); kept verbatim
					float impulse = soft ? -s.axialMass * massScale * (Cdot + bias) - impulseScale * s.lowerImpulse : -s.axialMass * (Cdot + bias);
					float oldImpulse = s.upperImpulse;
					s.upperImpulse = S2_MAXF(s.upperImpulse + impulse, 0.0f);
					impulse = s.upperImpulse - oldImpulse;
					wA += iA * impulse;
					wB -= iB * impulse;
				}
			}
		}

		{
			V2 rA = rotate(qA, s.lA);
			V2 rB = rotate(qB, s.lB);
			V2 Cdot = sub(add(vB, crossSV(wB, rB)), add(vA, crossSV(wA, rA)));
			V2 impulse;
			if (KIND == JSOLVE_PLAIN)
			{
				impulse = mulMV(s.pivotMass, neg(Cdot));
			}
			else
			{
				V2 bias = v2(0.0f, 0.0f);
				float massScale = 1.0f, impulseScale = 0.0f;
				V2 dcA = v2(da.x, da.y), dcB = v2(db.x, db.y);
				if (KIND == JSOLVE_SOFT)
				{
					if (useBias)
					{
						V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), s.centerDiff0);
						bias = mulSV(s.biasCoefficient, separation);
						massScale = s.massCoefficient;
						impulseScale = s.impulseCoefficient;
					}
				}
				else
				{
					V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), s.centerDiff0);
					bias = mulSV(S2_BAUMGARTE * inv_h, separation);
				}
				M22 K = revoluteK(mA, mB, iA, iB, rA, rB);
				V2 bb = solve22(K, add(Cdot, bias));
				if (KIND == JSOLVE_SOFT)
				{
					impulse.x = -massScale * bb.x - impulseScale * s.impulse.x;
					impulse.y = -massScale * bb.y - impulseScale * s.impulse.y;
				}
				else
				{
					impulse.x = -bb.x;
					impulse.y = -bb.y;
				}
			}
			s.impulse.x += impulse.x;
			s.impulse.y += impulse.y;
			vA = mulSub(vA, mA, impulse);
			wA -= iA * cross(rA, impulse);
			vB = mulAdd(vB, mB, impulse);
			wB += iB * cross(rB, impulse);
		}
		storeJointImpulses(jv, k, s);
	}

	if (writeA)
	{
		b.vel[s.ia] = make_float4(vA.x, vA.y, wA, 0.0f);
	}
	if (writeB)
	{
		b.vel[s.ib] = make_float4(vB.x, vB.y, wB, 0.0f);
	}
}

__global__ __launch_bounds__(S2_BLOCK) void storeJointsKernel(JointView jv, s2amdJoint* wire)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= jv.count)
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

void launchSolveJoints(hipStream_t s, int kind, const JointView& j, const BodyView& b, int begin, int end, const StepConsts& sc, float h,
					   float inv_h, int useBias)
{
	if (end <= begin)
	{
		return;
	}
	dim3 g = gridFor(end - begin), t(S2_BLOCK);
	switch (kind)
	{
		case JSOLVE_PLAIN:
			solveJointsKernel<JSOLVE_PLAIN><<<g, t, 0, s>>>(j, b, begin, end, sc, h, inv_h, useBias);
			break;
		case JSOLVE_SOFT:
			solveJointsKernel<JSOLVE_SOFT><<<g, t, 0, s>>>(j, b, begin, end, sc, h, inv_h, useBias);
			break;
		case JSOLVE_BAUMGARTE:
			solveJointsKernel<JSOLVE_BAUMGARTE><<<g, t, 0, s>>>(j, b, begin, end, sc, h, inv_h, useBias);
			break;
		case JSOLVE_POSITION:
			solveJointsKernel<JSOLVE_POSITION><<<g, t, 0, s>>>(j, b, begin, end, sc, h, inv_h, useBias);
			break;
		case JSOLVE_XPBD:
			solveJointsKernel<JSOLVE_XPBD><<<g, t, 0, s>>>(j, b, begin, end, sc, h, inv_h, useBias);
			break;
		case JSOLVE_WARM:
			solveJointsKernel<JSOLVE_WARM><<<g, t, 0, s>>>(j, b, begin, end, sc, h, inv_h, useBias);
			break;
	}
}

void launchStoreJoints(hipStream_t s, const JointView& j, s2amdJoint* wire)
{
	if (j.count > 0)
	{
		storeJointsKernel<<<gridFor(j.count), dim3(S2_BLOCK), 0, s>>>(j, wire);
	}
}
