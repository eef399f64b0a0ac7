// Contact-constraint kernels: one thread per constraint, constraints laid out SoA in sweep
// (colour-major) order so a wave reads 64 consecutive records of each array with 16-byte lanes.
// A sweep kernel is launched once per colour batch [begin, end): inside a batch no two
// constraints touch the same writable body, so the batch reproduces, bit for bit, a sequential
// Gauss-Seidel pass over the same constraints in the same order.
//
// Each kernel names the reference function it stands for (paths relative to /root/reference).

#include "launch.h"
#include "s2_device.h"

#include "solver2d_amd.h"

#define S2_BLOCK 256

struct CHeader
{
	int ia, ib;
	float mA, iA, mB, iB;
	V2 normal;
	float friction;
	int pointCount;
	bool writeA, writeB;
};

S2_DEV CHeader loadHeader(const ContactView& c, int k)
{
	CHeader h;
	int2 b = c.bodies[k];
	float4 m = c.mass[k];
	float4 nf = c.nf[k];
	h.ia = b.x, h.ib = b.y;
	h.mA = m.x, h.iA = m.y, h.mB = m.z, h.iB = m.w;
	h.normal = v2(nf.x, nf.y);
	h.friction = nf.z;
	uint32_t bits = asBits(nf.w);
	h.pointCount = (int)(bits & 0xffu);
	h.writeA = (bits & S2C_WRITE_A) != 0;
	h.writeB = (bits & S2C_WRITE_B) != 0;
	return h;
}

struct BodyVel
{
	V2 v;
	float w;
};
struct BodyPose
{
	V2 dc;
	Rot q;
};

S2_DEV BodyVel loadVel(const BodyView& b, int i)
{
	float4 t = b.vel[i];
	BodyVel r;
	r.v = v2(t.x, t.y);
	r.w = t.z;
	return r;
}
S2_DEV void storeVel(const BodyView& b, int i, V2 v, float w)
{
	b.vel[i] = make_float4(v.x, v.y, w, 0.0f);
}
S2_DEV BodyPose loadPose(const BodyView& b, int i)
{
	float4 t = b.dq[i];
	BodyPose r;
	r.dc = v2(t.x, t.y);
	r.q.s = t.z, r.q.c = t.w;
	return r;
}
S2_DEV void storePose(const BodyView& b, int i, V2 dc, Rot q)
{
	b.dq[i] = make_float4(dc.x, dc.y, q.s, q.c);
}

// ---------------------------------------------------------------------------------------------
// prepare: s2PrepareContacts_PGS / _Soft (solve_common.c:93-168, :188-274), s2PrepareContacts
// (solve_tgs_ngs.c:19-89), s2PrepareContacts_Sticky (solve_tgs_sticky.c:19-165),
// s2PrepareContacts_XPBD (solve_xpbd.c:18-86), and the first loop of s2CreateContactSolver
// (solve_pgs_ngs_block.c:151-277).  Reads the wire contact + two bodies, writes the SoA record.
// Embarrassingly parallel: no body is written.
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void prepareContactsKernel(ContactView c, BodyView b, s2amdContact* wire, const s2amdBody* wireBodies,
																  StepConsts sc, float h, float hertz, int posSolver)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= c.count)
	{
		return;
	}
	s2amdContact* contact = wire + c.contactIndex[k];
	int pointCount = contact->pointCount;
	int ia = contact->bodyA, ib = contact->bodyB;
	V2 normal = v2(contact->normal[0], contact->normal[1]);
	float friction = contact->friction;

	const s2amdBody* wa = wireBodies + ia;
	const s2amdBody* wb = wireBodies + ib;
	float mA = wa->invMass, iA = wa->invI;
	float mB = wb->invMass, iB = wb->invI;
	V2 lcA = v2(wa->localCenter[0], wa->localCenter[1]);
	V2 lcB = v2(wb->localCenter[0], wb->localCenter[1]);

	BodyPose pA = loadPose(b, ia);
	BodyPose pB = loadPose(b, ib);
	Rot qA = pA.q, qB = pB.q;

	uint32_t fa = b.flags[ia], fb = b.flags[ib];
	uint32_t wbit = posSolver ? S2F_WRITE_POS : S2F_WRITE_VEL;
	uint32_t bits = (uint32_t)pointCount;
	if (fa & wbit)
	{
		bits |= S2C_WRITE_A;
	}
	if (fb & wbit)
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
		contact->frictionPersisted = 1;
	}

	c.nf[k] = make_float4(normal.x, normal.y, friction, fromBits(bits));
}

// ---------------------------------------------------------------------------------------------
// warm start: s2WarmStartContacts (solve_common.c:276-326, current anchors),
// s2WarmStartContacts_Fixed (solve_soft_step.c:16-63), and the second loop of
// s2CreateContactSolver (solve_pgs_ngs_block.c:279-319, fixed anchors, reduced point count)
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void warmStartContactsKernel(ContactView c, BodyView b, int begin, int end)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	int pointCount = h.pointCount;
	V2 tangent = rightPerp(h.normal);
	if (KIND == WARM_BLOCK)
	{
		pointCount = (int)asBits(c.blockK[k].w);
		tangent = crossVS(h.normal, 1.0f);
	}
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	Rot qA, qB;
	if (KIND == WARM_CURRENT)
	{
		qA = loadPose(b, h.ia).q;
		qB = loadPose(b, h.ib).q;
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			V2 rA, rB;
			if (KIND == WARM_CURRENT)
			{
				float4 an = c.anchor[j][k];
				rA = rotate(qA, v2(an.x, an.y));
				rB = rotate(qB, v2(an.z, an.w));
			}
			else
			{
				float4 r0 = c.r0[j][k];
				rA = v2(r0.x, r0.y);
				rB = v2(r0.z, r0.w);
			}
			float2 imp = c.impulse[j][k];
			V2 P = add(mulSV(imp.x, h.normal), mulSV(imp.y, tangent));
			wA -= h.iA * cross(rA, P);
			vA = mulAdd(vA, -h.mA, P);
			wB += h.iB * cross(rB, P);
			vB = mulAdd(vB, h.mB, P);
		}
	}
	if (h.writeA)
	{
		storeVel(b, h.ia, vA, wA);
	}
	if (h.writeB)
	{
		storeVel(b, h.ib, vB, wB);
	}
}

// ---------------------------------------------------------------------------------------------
// soft velocity sweeps:
//   SOFT_TGS    s2SolveContacts_TGS_Soft     solve_tgs_soft.c:17-135
//   SOFT_PGS    s2SolveContacts_PGS_Soft     solve_pgs_soft.c:16-125
//   SOFT_JACOBI s2SolveContacts_Jacobi_Soft  solve_jacobi.c:21-132  (writes per-constraint deltas)
//   SOFT_FIXED  s2SolveContacts_TGS_Fixed    solve_soft_step.c:66-177
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveContactsSoftKernel(ContactView c, BodyView b, int begin, int end, float inv_h, int useBias)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	const float biasCap = (KIND == SOFT_TGS || KIND == SOFT_JACOBI) ? -S2_MAX_BAUMGARTE_VELOCITY : -0.5f * S2_MAX_BAUMGARTE_VELOCITY;

	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 dcA, dcB;
	Rot qA, qB;
	if (KIND == SOFT_TGS || KIND == SOFT_FIXED)
	{
		BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
		dcA = pA.dc, qA = pA.q, dcB = pB.dc, qB = pB.q;
	}
	V2 normal = h.normal;
	V2 tangent = rightPerp(normal);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;

	V2 rAj[2], rBj[2];
	float nImp[2], tImp[2];

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 par = c.param[j][k];
			float4 sf = c.soft[j][k];
			float2 imp = c.impulse[j][k];
			V2 rA, rB;
			float s;
			if (KIND == SOFT_TGS)
			{
				float4 an = c.anchor[j][k];
				rA = rotate(qA, v2(an.x, an.y));
				rB = rotate(qB, v2(an.z, an.w));
				V2 ds = add(sub(dcB, dcA), sub(rB, rA));
				s = dot(ds, normal) + par.x;
			}
			else if (KIND == SOFT_FIXED)
			{
				float4 an = c.anchor[j][k];
				float4 r0 = c.r0[j][k];
				V2 ds = add(sub(dcB, dcA), sub(rotate(qB, v2(an.z, an.w)), rotate(qA, v2(an.x, an.y))));
				s = dot(ds, normal) + par.x;
				rA = v2(r0.x, r0.y);
				rB = v2(r0.z, r0.w);
			}
			else
			{
				float4 r0 = c.r0[j][k];
				s = par.w;
				rA = v2(r0.x, r0.y);
				rB = v2(r0.z, r0.w);
			}
			rAj[j] = rA, rBj[j] = rB;

			float bias = 0.0f, massScale = 1.0f, impulseScale = 0.0f;
			if (s > 0.0f)
			{
				bias = s * inv_h;
			}
			else if (useBias)
			{
				bias = S2_MAXF(sf.x * s, biasCap);
				massScale = sf.y;
				impulseScale = sf.z;
			}

			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vn = dot(sub(vrB, vrA), normal);

			float impulse = -par.y * massScale * (vn + bias) - impulseScale * imp.x;
			float newImpulse = S2_MAXF(imp.x + impulse, 0.0f);
			impulse = newImpulse - imp.x;
			nImp[j] = newImpulse;
			tImp[j] = imp.y;

			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float tangentMass = c.param[j][k].z;
			V2 rA = rAj[j], rB = rBj[j];
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vt = dot(sub(vrB, vrA), tangent);
			float impulse = -tangentMass * vt;
			float maxFriction = h.friction * nImp[j];
			float newImpulse = S2_CLAMPF(tImp[j] + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - tImp[j];
			tImp[j] = newImpulse;
			V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
		}
	}

	if (KIND == SOFT_JACOBI)
	{
		// solve_jacobi.c:126-130: the body sums these in constraint order (jacobiApplyKernel)
		V2 dA = sub(vA, A.v), dB = sub(vB, B.v);
		c.deltaA[k] = make_float4(dA.x, dA.y, wA - A.w, 0.0f);
		c.deltaB[k] = make_float4(dB.x, dB.y, wB - B.w, 0.0f);
	}
	else
	{
		if (h.writeA)
		{
			storeVel(b, h.ia, vA, wA);
		}
		if (h.writeB)
		{
			storeVel(b, h.ib, vB, wB);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// rigid velocity sweeps:
//   RIGID_BAUMGARTE s2SolveContacts_PGS_Baumgarte solve_pgs.c:17-122      (normal first, fixed anchors)
//   RIGID_PGS       s2SolveContacts_PGS           solve_pgs_ngs.c:16-124  (friction first, no speculative)
//   RIGID_TGS       s2SolveContacts_TGS           solve_tgs_ngs.c:91-201  (current anchors, speculative)
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveContactsRigidKernel(ContactView c, BodyView b, int begin, int end, float inv_h)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 normal = h.normal;
	V2 tangent = KIND == RIGID_PGS ? crossVS(normal, 1.0f) : rightPerp(normal);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float friction = h.friction;

	V2 rAj[2], rBj[2];
	float nImp[2], tImp[2], sep[2], nMass[2], tMass[2], adj[2];
	V2 dcA, dcB;
	Rot qA, qB;
	if (KIND == RIGID_TGS)
	{
		BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
		dcA = pA.dc, qA = pA.q, dcB = pB.dc, qB = pB.q;
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 par = c.param[j][k];
			float2 imp = c.impulse[j][k];
			adj[j] = par.x, nMass[j] = par.y, tMass[j] = par.z, sep[j] = par.w;
			nImp[j] = imp.x, tImp[j] = imp.y;
			if (KIND == RIGID_TGS)
			{
				float4 an = c.anchor[j][k];
				rAj[j] = rotate(qA, v2(an.x, an.y));
				rBj[j] = rotate(qB, v2(an.z, an.w));
			}
			else
			{
				float4 r0 = c.r0[j][k];
				rAj[j] = v2(r0.x, r0.y);
				rBj[j] = v2(r0.z, r0.w);
			}
		}
	}

	if (KIND == RIGID_PGS)
	{
		// friction first: solve_pgs_ngs.c:42-80
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (j < h.pointCount)
			{
				if (sep[j] > 0.0f)
				{
					tImp[j] = 0.0f;
					continue;
				}
				V2 rA = rAj[j], rB = rBj[j];
				V2 vrB = add(vB, crossSV(wB, rB));
				V2 vrA = add(vA, crossSV(wA, rA));
				float vt = dot(sub(vrB, vrA), tangent);
				float lambda = tMass[j] * (-vt);
				float maxFriction = friction * nImp[j];
				float newImpulse = S2_CLAMPF(tImp[j] + lambda, -maxFriction, maxFriction);
				lambda = newImpulse - tImp[j];
				tImp[j] = newImpulse;
				V2 P = mulSV(lambda, tangent);
				vA = mulSub(vA, mA, P);
				wA -= iA * cross(rA, P);
				vB = mulAdd(vB, mB, P);
				wB += iB * cross(rB, P);
			}
		}
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (j < h.pointCount)
			{
				if (sep[j] > 0.0f)
				{
					nImp[j] = 0.0f;
					continue;
				}
				V2 rA = rAj[j], rB = rBj[j];
				V2 vrB = add(vB, crossSV(wB, rB));
				V2 vrA = add(vA, crossSV(wA, rA));
				float vn = dot(sub(vrB, vrA), normal);
				float impulse = -nMass[j] * vn;
				float newImpulse = S2_MAXF(nImp[j] + impulse, 0.0f);
				impulse = newImpulse - nImp[j];
				nImp[j] = newImpulse;
				V2 P = mulSV(impulse, normal);
				vA = mulSub(vA, mA, P);
				wA -= iA * cross(rA, P);
				vB = mulAdd(vB, mB, P);
				wB += iB * cross(rB, P);
			}
		}
	}
	else
	{
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (j < h.pointCount)
			{
				V2 rA = rAj[j], rB = rBj[j];
				float bias;
				if (KIND == RIGID_BAUMGARTE)
				{
					if (sep[j] > 0.0f)
					{
						bias = sep[j] * inv_h;
					}
					else
					{
						bias = S2_MAXF(S2_BAUMGARTE * inv_h * S2_MINF(0.0f, sep[j] + S2_LINEAR_SLOP), -S2_MAX_BAUMGARTE_VELOCITY);
					}
				}
				else
				{
					V2 d = add(sub(dcB, dcA), sub(rB, rA));
					float separation = dot(d, normal) + adj[j];
					bias = separation > 0.0f ? separation * inv_h : 0.0f;
				}
				V2 vrB = add(vB, crossSV(wB, rB));
				V2 vrA = add(vA, crossSV(wA, rA));
				float vn = dot(sub(vrB, vrA), normal);
				float impulse = -nMass[j] * (vn + bias);
				float newImpulse = S2_MAXF(nImp[j] + impulse, 0.0f);
				impulse = newImpulse - nImp[j];
				nImp[j] = newImpulse;
				V2 P = mulSV(impulse, normal);
				vA = mulSub(vA, mA, P);
				wA -= iA * cross(rA, P);
				vB = mulAdd(vB, mB, P);
				wB += iB * cross(rB, P);
			}
		}
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (j < h.pointCount)
			{
				V2 rA = rAj[j], rB = rBj[j];
				V2 vrB = add(vB, crossSV(wB, rB));
				V2 vrA = add(vA, crossSV(wA, rA));
				float vt = dot(sub(vrB, vrA), tangent);
				float lambda = KIND == RIGID_BAUMGARTE ? tMass[j] * (-vt) : -tMass[j] * vt;
				float maxFriction = friction * nImp[j];
				float newImpulse = S2_CLAMPF(tImp[j] + lambda, -maxFriction, maxFriction);
				lambda = newImpulse - tImp[j];
				tImp[j] = newImpulse;
				V2 P = mulSV(lambda, tangent);
				vA = mulSub(vA, mA, P);
				wA -= iA * cross(rA, P);
				vB = mulAdd(vB, mB, P);
				wB += iB * cross(rB, P);
			}
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
		}
	}
	if (h.writeA)
	{
		storeVel(b, h.ia, vA, wA);
	}
	if (h.writeB)
	{
		storeVel(b, h.ib, vB, wB);
	}
}

// s2SolveContacts_TGS_Sticky: solve_tgs_sticky.c:167-310
__global__ __launch_bounds__(S2_BLOCK) void solveContactsStickyKernel(ContactView c, BodyView b, s2amdContact* wire, int begin, int end,
																	  float inv_h, int useBias)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	const float contactBaumgarte = 0.8f;
	const float frictionBaumgarte = 0.5f;
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 dcA = pA.dc, dcB = pB.dc;
	Rot qA = pA.q, qB = pB.q;
	V2 normal = h.normal;
	V2 tangent = rightPerp(normal);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float totalNormalImpulse = 0.0f;
	float nImp[2], tImp[2];
	bool slipped = false;

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 an = c.anchor[j][k];
			float4 par = c.param[j][k];
			float2 imp = c.impulse[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + par.x;
			float bias = 0.0f;
			if (separation > 0.0f)
			{
				bias = separation * inv_h;
			}
			else if (useBias)
			{
				bias = S2_MAXF(-S2_MAX_BAUMGARTE_VELOCITY, contactBaumgarte * separation * inv_h);
			}
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 vrB = add(vB, crossSV(wB, rB));
			float vn = dot(sub(vrB, vrA), normal);
			float impulse = -par.y * (vn + bias);
			float newImpulse = S2_MAXF(imp.x + impulse, 0.0f);
			impulse = newImpulse - imp.x;
			nImp[j] = newImpulse;
			tImp[j] = imp.y;
			totalNormalImpulse += newImpulse;
			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 fa = c.fanchor[j][k];
			float tangentMass = c.param[j][k].z;
			float tangentSeparation = c.soft[j][k].w;
			V2 rAf = rotate(qA, v2(fa.x, fa.y));
			V2 rBf = rotate(qB, v2(fa.z, fa.w));
			V2 d = add(sub(dcB, dcA), sub(rBf, rAf));
			float separation = dot(d, tangent) + tangentSeparation;
			float bias = useBias ? frictionBaumgarte * separation * inv_h : 0.0f;
			V2 vrA = add(vA, crossSV(wA, rAf));
			V2 vrB = add(vB, crossSV(wB, rBf));
			float vt = dot(sub(vrB, vrA), tangent);
			float impulse = -tangentMass * (vt + bias);
			float maxFriction = 0.5f * h.friction * totalNormalImpulse;
			float newImpulse = tImp[j] + impulse;
			if (newImpulse < -maxFriction)
			{
				newImpulse = -maxFriction;
				slipped = true;
			}
			else if (newImpulse > maxFriction)
			{
				newImpulse = maxFriction;
				slipped = true;
			}
			impulse = newImpulse - tImp[j];
			tImp[j] = newImpulse;
			V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rAf, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rBf, P);
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
		}
	}
	if (slipped)
	{
		wire[c.contactIndex[k]].frictionPersisted = 0; // solve_tgs_sticky.c:284,289
	}
	if (h.writeA)
	{
		storeVel(b, h.ia, vA, wA);
	}
	if (h.writeB)
	{
		storeVel(b, h.ib, vB, wB);
	}
}

// s2SolveContact_NGS: solve_common.c:328-394
__global__ __launch_bounds__(S2_BLOCK) void solveContactsNGSKernel(ContactView c, BodyView b, int begin, int end)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
	V2 dcA = pA.dc, dcB = pB.dc;
	Rot qA = pA.q, qB = pB.q;
	V2 normal = h.normal;
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 par = c.param[j][k];
			if (par.w > 0.0f)
			{
				continue;
			}
			float4 an = c.anchor[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + par.x;
			float C = S2_CLAMPF(S2_BAUMGARTE * (separation + S2_LINEAR_SLOP), -S2_MAX_LINEAR_CORRECTION, 0.0f);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
			float impulse = K > 0.0f ? -C / K : 0.0f;
			V2 P = mulSV(impulse, normal);
			dcA = mulSub(dcA, mA, P);
			qA = integrateRot(qA, -iA * cross(rA, P));
			dcB = mulAdd(dcB, mB, P);
			qB = integrateRot(qB, iB * cross(rB, P));
		}
	}
	if (h.writeA)
	{
		storePose(b, h.ia, dcA, qA);
	}
	if (h.writeB)
	{
		storePose(b, h.ib, dcB, qB);
	}
}

// s2SolveContactPositions_XPBD: solve_xpbd.c:88-216
__global__ __launch_bounds__(S2_BLOCK) void xpbdContactPositionsKernel(ContactView c, BodyView b, int begin, int end, float hh)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	const float baseCompliance = 0.0f;
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float compliance = (mA == 0.0f || mB == 0.0f) ? 0.25f * baseCompliance : baseCompliance;
	BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
	V2 dcA = pA.dc, dcB = pB.dc;
	Rot qA = pA.q, qB = pB.q;
	V2 normal = h.normal;
	V2 tangent = crossVS(normal, 1.0f);
	float nImp[2] = {0.0f, 0.0f}, tImp[2] = {0.0f, 0.0f};

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 an = c.anchor[j][k];
			float4 r0 = c.r0[j][k];
			float2 imp = c.impulse[j][k];
			nImp[j] = imp.x, tImp[j] = imp.y;
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 drA = sub(rA, v2(r0.x, r0.y));
			V2 drB = sub(rB, v2(r0.z, r0.w));
			V2 ds = add(sub(dcB, dcA), sub(drB, drA));
			float C = dot(ds, normal) + c.param[j][k].w;
			if (C > 0)
			{
				nImp[j] = 0.0f;
				continue;
			}
			C = S2_MAXF(-S2_MAX_BAUMGARTE_VELOCITY * hh, C);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float kA = mA + iA * rnA * rnA;
			float kB = mB + iB * rnB * rnB;
			float lambda = -C / (kA + kB + compliance);
			nImp[j] = lambda;
			V2 P = mulSV(lambda, normal);
			dcA = mulSub(dcA, mA, P);
			qA = integrateRot(qA, -iA * cross(rA, P));
			dcB = mulAdd(dcB, mB, P);
			qB = integrateRot(qB, iB * cross(rB, P));
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 an = c.anchor[j][k];
			float4 r0 = c.r0[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 drA = sub(rA, v2(r0.x, r0.y));
			V2 drB = sub(rB, v2(r0.z, r0.w));
			V2 dp = add(sub(dcB, dcA), sub(drB, drA));
			float C = dot(dp, tangent);
			float rtA = cross(rA, tangent);
			float rtB = cross(rB, tangent);
			float kA = mA + iA * rtA * rtA;
			float kB = mB + iB * rtB * rtB;
			float lambda = -C / (kA + kB);
			float maxLambda = h.friction * nImp[j];
			if (lambda < -maxLambda || maxLambda < lambda)
			{
				tImp[j] = 0.0f;
			}
			else
			{
				tImp[j] = lambda;
				V2 P = mulSV(lambda, tangent);
				dcA = mulSub(dcA, mA, P);
				qA = integrateRot(qA, -iA * cross(rA, P));
				dcB = mulAdd(dcB, mB, P);
				qB = integrateRot(qB, iB * cross(rB, P));
			}
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
		}
	}
	if (h.writeA)
	{
		storePose(b, h.ia, dcA, qA);
	}
	if (h.writeB)
	{
		storePose(b, h.ib, dcB, qB);
	}
}

// s2SolveContactVelocities_XPBD: solve_xpbd.c:218-338
__global__ __launch_bounds__(S2_BLOCK) void xpbdContactVelocitiesKernel(ContactView c, BodyView b, int begin, int end, float hh)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	float inv_h = hh > 0.0f ? 1.0f / hh : 0.0f;
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	Rot qA = loadPose(b, h.ia).q, qB = loadPose(b, h.ib).q;
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 normal = h.normal;
	V2 tangent = crossVS(normal, 1.0f);
	float nImp[2] = {0.0f, 0.0f};

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float2 imp = c.impulse[j][k];
			nImp[j] = imp.x;
			if (imp.x == 0.0f)
			{
				continue;
			}
			float4 an = c.anchor[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 dv = sub(vrB, vrA);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float kA = mA + iA * rnA * rnA;
			float kB = mB + iB * rnB * rnB;
			float vn = dot(dv, normal);
			float lambda = -vn / (kA + kB);
			V2 P = mulSV(lambda, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 an = c.anchor[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 dv = sub(vrB, vrA);
			float vt = dot(dv, tangent);
			if (vt == 0.0f)
			{
				continue;
			}
			float rtA = cross(rA, tangent);
			float rtB = cross(rB, tangent);
			float kA = mA + iA * rtA * rtA;
			float kB = mB + iB * rtB * rtB;
			float maxFrictionImpulse = h.friction * nImp[j];
			float huf = (maxFrictionImpulse * inv_h) * (kA + kB);
			float abs_vt = S2_ABSF(vt);
			float Cdot = (vt / abs_vt) * S2_MINF(huf, abs_vt);
			float lambda = -Cdot / (kA + kB);
			c.impulse[j][k] = make_float2(nImp[j], lambda);
			V2 P = mulSV(lambda, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}
	if (h.writeA)
	{
		storeVel(b, h.ia, vA, wA);
	}
	if (h.writeB)
	{
		storeVel(b, h.ib, vB, wB);
	}
}

// ---------------------------------------------------------------------------------------------
// PGS_NGS_Block: s2BlockSolveVelocity (solve_pgs_ngs_block.c:329-658) and
// s2BlockSolvePosition (:679-890)
// ---------------------------------------------------------------------------------------------
#define BLOCK_APPLY_VELOCITY(d)                                                                                                  \
	{                                                                                                                            \
		V2 P1 = mulSV((d).x, normal);                                                                                            \
		V2 P2 = mulSV((d).y, normal);                                                                                            \
		vA = mulSub(vA, mA, add(P1, P2));                                                                                        \
		wA -= iA * (cross(rA1, P1) + cross(rA2, P2));                                                                            \
		vB = mulAdd(vB, mB, add(P1, P2));                                                                                        \
		wB += iB * (cross(rB1, P1) + cross(rB2, P2));                                                                            \
	}

__global__ __launch_bounds__(S2_BLOCK) void blockSolveVelocityKernel(ContactView c, BodyView b, int begin, int end)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	float4 K4 = c.blockK[k];
	int pointCount = (int)asBits(K4.w);
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 normal = h.normal;
	V2 tangent = crossVS(normal, 1.0f);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float friction = h.friction;

	V2 rAj[2], rBj[2];
	float nImp[2], tImp[2], nMass[2], tMass[2], vBias[2];
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			float4 r0 = c.r0[j][k];
			float4 par = c.param[j][k];
			float2 imp = c.impulse[j][k];
			rAj[j] = v2(r0.x, r0.y), rBj[j] = v2(r0.z, r0.w);
			nMass[j] = par.y, tMass[j] = par.z;
			nImp[j] = imp.x, tImp[j] = imp.y;
			vBias[j] = c.soft[j][k].x;
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			V2 vrB = add(vB, crossSV(wB, rBj[j]));
			V2 vrA = add(vA, crossSV(wA, rAj[j]));
			V2 dv = sub(vrB, vrA);
			float vt = dot(dv, tangent);
			float lambda = tMass[j] * (-vt);
			float maxFriction = friction * nImp[j];
			float newImpulse = S2_CLAMPF(tImp[j] + lambda, -maxFriction, maxFriction);
			lambda = newImpulse - tImp[j];
			tImp[j] = newImpulse;
			V2 P = mulSV(lambda, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rAj[j], P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rBj[j], P);
		}
	}

	if (pointCount == 1)
	{
		V2 vrB = add(vB, crossSV(wB, rBj[0]));
		V2 vrA = add(vA, crossSV(wA, rAj[0]));
		V2 dv = sub(vrB, vrA);
		float vn = dot(dv, normal);
		float lambda = -nMass[0] * (vn - vBias[0]);
		float newImpulse = S2_MAXF(nImp[0] + lambda, 0.0f);
		lambda = newImpulse - nImp[0];
		nImp[0] = newImpulse;
		V2 P = mulSV(lambda, normal);
		vA = mulSub(vA, mA, P);
		wA -= iA * cross(rAj[0], P);
		vB = mulAdd(vB, mB, P);
		wB += iB * cross(rBj[0], P);
	}
	else if (pointCount == 2)
	{
		V2 rA1 = rAj[0], rB1 = rBj[0], rA2 = rAj[1], rB2 = rBj[1];
		M22 K, NM;
		K.cx = v2(K4.x, K4.y);
		K.cy = v2(K4.y, K4.z);
		float4 nm = c.blockNM[k];
		NM.cx = v2(nm.x, nm.y);
		NM.cy = v2(nm.z, nm.w);
		V2 a = v2(nImp[0], nImp[1]);
		V2 vrA, vrB;
		vrA = add(vA, crossSV(wA, rA1));
		vrB = add(vB, crossSV(wB, rB1));
		V2 dv1 = sub(vrB, vrA);
		vrA = add(vA, crossSV(wA, rA2));
		vrB = add(vB, crossSV(wB, rB2));
		V2 dv2 = sub(vrB, vrA);
		float vn1 = dot(dv1, normal);
		float vn2 = dot(dv2, normal);
		V2 bb = v2(vn1 - vBias[0], vn2 - vBias[1]);
		bb = sub(bb, mulMV(K, a));

		for (;;)
		{
			V2 x = neg(mulMV(NM, bb));
			if (x.x >= 0.0f && x.y >= 0.0f)
			{
				V2 d = sub(x, a);
				BLOCK_APPLY_VELOCITY(d);
				nImp[0] = x.x, nImp[1] = x.y;
				break;
			}
			x.x = -nMass[0] * bb.x;
			x.y = 0.0f;
			vn1 = 0.0f;
			vn2 = K.cx.y * x.x + bb.y;
			if (x.x >= 0.0f && vn2 >= 0.0f)
			{
				V2 d = sub(x, a);
				BLOCK_APPLY_VELOCITY(d);
				nImp[0] = x.x, nImp[1] = x.y;
				break;
			}
			x.x = 0.0f;
			x.y = -nMass[1] * bb.y;
			vn1 = K.cy.x * x.y + bb.x;
			vn2 = 0.0f;
			if (x.y >= 0.0f && vn1 >= 0.0f)
			{
				V2 d = sub(x, a);
				BLOCK_APPLY_VELOCITY(d);
				nImp[0] = x.x, nImp[1] = x.y;
				break;
			}
			x.x = 0.0f;
			x.y = 0.0f;
			vn1 = bb.x;
			vn2 = bb.y;
			if (vn1 >= 0.0f && vn2 >= 0.0f)
			{
				V2 d = sub(x, a);
				BLOCK_APPLY_VELOCITY(d);
				nImp[0] = x.x, nImp[1] = x.y;
				break;
			}
			break;
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
		}
	}
	if (h.writeA)
	{
		storeVel(b, h.ia, vA, wA);
	}
	if (h.writeB)
	{
		storeVel(b, h.ib, vB, wB);
	}
}

#define BLOCK_APPLY_POSITION(d)                                                                                                  \
	{                                                                                                                            \
		V2 P1 = mulSV((d).x, normal);                                                                                            \
		V2 P2 = mulSV((d).y, normal);                                                                                            \
		dcA = mulSub(dcA, mA, add(P1, P2));                                                                                      \
		qA = integrateRot(qA, -iA * (cross(rA1, P1) + cross(rA2, P2)));                                                          \
		dcB = mulAdd(dcB, mB, add(P1, P2));                                                                                      \
		qB = integrateRot(qB, iB * (cross(rB1, P1) + cross(rB2, P2)));                                                           \
	}

__global__ __launch_bounds__(S2_BLOCK) void blockSolvePositionKernel(ContactView c, BodyView b, int begin, int end)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= end)
	{
		return;
	}
	CHeader h = loadHeader(c, k);
	int pointCount = (int)asBits(c.blockK[k].w);
	const float slop = S2_LINEAR_SLOP;
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
	V2 dcA = pA.dc, dcB = pB.dc;
	Rot qA = pA.q, qB = pB.q;
	V2 normal = h.normal;
	bool degenerate = pointCount != 2;

	if (pointCount == 2)
	{
		float4 an1 = c.anchor[0][k], an2 = c.anchor[1][k];
		float adj1 = c.param[0][k].x, adj2 = c.param[1][k].x;
		V2 rA1 = rotate(qA, v2(an1.x, an1.y));
		V2 rB1 = rotate(qB, v2(an1.z, an1.w));
		V2 rA2 = rotate(qA, v2(an2.x, an2.y));
		V2 rB2 = rotate(qB, v2(an2.z, an2.w));
		V2 dc = sub(dcB, dcA);
		V2 d1 = add(dc, sub(rB1, rA1));
		float separation1 = dot(d1, normal) + adj1;
		V2 d2 = add(dc, sub(rB2, rA2));
		float separation2 = dot(d2, normal) + adj2;
		float C1 = S2_CLAMPF(S2_BAUMGARTE * (separation1 + slop), -S2_MAX_LINEAR_CORRECTION, 0.0f);
		float C2 = S2_CLAMPF(S2_BAUMGARTE * (separation2 + slop), -S2_MAX_LINEAR_CORRECTION, 0.0f);
		V2 bb = v2(C1, C2);
		float rn1A = cross(rA1, normal);
		float rn1B = cross(rB1, normal);
		float rn2A = cross(rA2, normal);
		float rn2B = cross(rB2, normal);
		float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
		float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
		float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
		const float k_maxConditionNumber = 10000.0f;
		if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12))
		{
			M22 K;
			K.cx = v2(k11, k12);
			K.cy = v2(k12, k22);
			M22 invK = inverse22(K);
			for (;;)
			{
				V2 x = neg(mulMV(invK, bb));
				if (x.x >= 0.0f && x.y >= 0.0f)
				{
					BLOCK_APPLY_POSITION(x);
					break;
				}
				x.x = -bb.x / k11;
				x.y = 0.0f;
				float vn2 = K.cx.y * x.x + bb.y;
				if (x.x >= 0.0f && vn2 >= 0.0f)
				{
					BLOCK_APPLY_POSITION(x);
					break;
				}
				x.x = 0.0f;
				x.y = -bb.y / k22;
				float vn1 = K.cy.x * x.y + bb.x;
				if (x.y >= 0.0f && vn1 >= 0.0f)
				{
					BLOCK_APPLY_POSITION(x);
					break;
				}
				break;
			}
		}
		else
		{
			degenerate = true;
		}
	}

	if (degenerate)
	{
		for (int j = 0; j < pointCount; ++j)
		{
			float4 an = c.anchor[j][k];
			float adj = c.param[j][k].x;
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + adj;
			float C = S2_CLAMPF(S2_BAUMGARTE * (separation + slop), -S2_MAX_LINEAR_CORRECTION, 0.0f);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
			float impulse = K > 0.0f ? -C / K : 0.0f;
			V2 P = mulSV(impulse, normal);
			dcA = mulSub(dcA, mA, P);
			qA = integrateRot(qA, -iA * cross(rA, P));
			dcB = mulAdd(dcB, mB, P);
			qB = integrateRot(qB, iB * cross(rB, P));
		}
	}
	if (h.writeA)
	{
		storePose(b, h.ia, dcA, qA);
	}
	if (h.writeB)
	{
		storePose(b, h.ib, dcB, qB);
	}
}

// s2StoreContactImpulses (solve_common.c:396-410), the scaled XPBD variant (solve_xpbd.c:517-527)
// and s2ContactSolver_StoreImpulses (solve_pgs_ngs_block.c:660-677, reduced point count)
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void storeImpulsesKernel(ContactView c, s2amdContact* wire, float scale)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= c.count)
	{
		return;
	}
	s2amdContact* contact = wire + c.contactIndex[k];
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
// launchers
// ---------------------------------------------------------------------------------------------
static inline dim3 gridFor(int n)
{
	return dim3((unsigned)((n + S2_BLOCK - 1) / S2_BLOCK));
}

void launchPrepareContacts(hipStream_t s, int kind, const ContactView& c, const BodyView& b, s2amdContact* wire, const s2amdBody* wireBodies,
						   const StepConsts& sc, float h, float hertz, int posSolver)
{
	if (c.count <= 0)
	{
		return;
	}
	dim3 g = gridFor(c.count), t(S2_BLOCK);
	switch (kind)
	{
		case PREP_PGS:
			prepareContactsKernel<PREP_PGS><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver);
			break;
		case PREP_SOFT:
			prepareContactsKernel<PREP_SOFT><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver);
			break;
		case PREP_TGS:
			prepareContactsKernel<PREP_TGS><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver);
			break;
		case PREP_STICKY:
			prepareContactsKernel<PREP_STICKY><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver);
			break;
		case PREP_XPBD:
			prepareContactsKernel<PREP_XPBD><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver);
			break;
		case PREP_BLOCK:
			prepareContactsKernel<PREP_BLOCK><<<g, t, 0, s>>>(c, b, wire, wireBodies, sc, h, hertz, posSolver);
			break;
	}
}

void launchWarmStartContacts(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end)
{
	if (end <= begin)
	{
		return;
	}
	dim3 g = gridFor(end - begin), t(S2_BLOCK);
	switch (kind)
	{
		case WARM_CURRENT:
			warmStartContactsKernel<WARM_CURRENT><<<g, t, 0, s>>>(c, b, begin, end);
			break;
		case WARM_FIXED:
			warmStartContactsKernel<WARM_FIXED><<<g, t, 0, s>>>(c, b, begin, end);
			break;
		case WARM_BLOCK:
			warmStartContactsKernel<WARM_BLOCK><<<g, t, 0, s>>>(c, b, begin, end);
			break;
	}
}

void launchSolveContactsSoft(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end, float inv_h, int useBias)
{
	if (end <= begin)
	{
		return;
	}
	dim3 g = gridFor(end - begin), t(S2_BLOCK);
	switch (kind)
	{
		case SOFT_TGS:
			solveContactsSoftKernel<SOFT_TGS><<<g, t, 0, s>>>(c, b, begin, end, inv_h, useBias);
			break;
		case SOFT_PGS:
			solveContactsSoftKernel<SOFT_PGS><<<g, t, 0, s>>>(c, b, begin, end, inv_h, useBias);
			break;
		case SOFT_JACOBI:
			solveContactsSoftKernel<SOFT_JACOBI><<<g, t, 0, s>>>(c, b, begin, end, inv_h, useBias);
			break;
		case SOFT_FIXED:
			solveContactsSoftKernel<SOFT_FIXED><<<g, t, 0, s>>>(c, b, begin, end, inv_h, useBias);
			break;
	}
}

void launchSolveContactsRigid(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end, float inv_h)
{
	if (end <= begin)
	{
		return;
	}
	dim3 g = gridFor(end - begin), t(S2_BLOCK);
	switch (kind)
	{
		case RIGID_BAUMGARTE:
			solveContactsRigidKernel<RIGID_BAUMGARTE><<<g, t, 0, s>>>(c, b, begin, end, inv_h);
			break;
		case RIGID_PGS:
			solveContactsRigidKernel<RIGID_PGS><<<g, t, 0, s>>>(c, b, begin, end, inv_h);
			break;
		case RIGID_TGS:
			solveContactsRigidKernel<RIGID_TGS><<<g, t, 0, s>>>(c, b, begin, end, inv_h);
			break;
	}
}

void launchSolveContactsSticky(hipStream_t s, const ContactView& c, const BodyView& b, s2amdContact* wire, int begin, int end, float inv_h,
							   int useBias)
{
	if (end <= begin)
	{
		return;
	}
	solveContactsStickyKernel<<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(c, b, wire, begin, end, inv_h, useBias);
}

void launchSolveContactsNGS(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end)
{
	if (end <= begin)
	{
		return;
	}
	solveContactsNGSKernel<<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(c, b, begin, end);
}

void launchXpbdContactPositions(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end, float h)
{
	if (end <= begin)
	{
		return;
	}
	xpbdContactPositionsKernel<<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(c, b, begin, end, h);
}

void launchXpbdContactVelocities(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end, float h)
{
	if (end <= begin)
	{
		return;
	}
	xpbdContactVelocitiesKernel<<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(c, b, begin, end, h);
}

void launchBlockSolveVelocity(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end)
{
	if (end <= begin)
	{
		return;
	}
	blockSolveVelocityKernel<<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(c, b, begin, end);
}

void launchBlockSolvePosition(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end)
{
	if (end <= begin)
	{
		return;
	}
	blockSolvePositionKernel<<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(c, b, begin, end);
}

void launchStoreImpulses(hipStream_t s, int kind, const ContactView& c, s2amdContact* wire, float scale)
{
	if (c.count <= 0)
	{
		return;
	}
	dim3 g = gridFor(c.count), t(S2_BLOCK);
	switch (kind)
	{
		case STORE_PLAIN:
			storeImpulsesKernel<STORE_PLAIN><<<g, t, 0, s>>>(c, wire, scale);
			break;
		case STORE_SCALED:
			storeImpulsesKernel<STORE_SCALED><<<g, t, 0, s>>>(c, wire, scale);
			break;
		case STORE_BLOCK:
			storeImpulsesKernel<STORE_BLOCK><<<g, t, 0, s>>>(c, wire, scale);
			break;
	}
}
