// Shared by the resident-island kernels (strip_kernel.hip: islandStepKernel, wide_kernel.hip: wideIslandKernel).
#pragma once

#include "body_ops.h"

// s2PrepareContacts_Soft (solve_common.c:188-274) for ONE constraint straight from its wire record: the operations of
// prepareContactsKernel<PREP_SOFT> (contact_kernels.hip) in the same order, so the same bits -- but the prepared record
// goes into the caller's registers instead of through the SoA arrays.  Body data: rotation and inverse masses from the
// LDS copies (== the wire body's: body_ops.h unpackBodyOne), local centres from the wire bodies or from LDS (llc).
template <int KIND, class BA>
S2_DEV SoftRegs<KIND> prepareSoftFromWire(const s2amdContact* contact, const s2amdBody* wireBodies, const uint32_t* hostFlags, const BA& lb, const float2* lmass,
										  int2 local, int bodyCapacity, int warmStart, const float2* llc = nullptr)
{
	SoftRegs<KIND> r;
	int pointCount = contact->pointCount;
	int ia = contact->bodyA, ib = contact->bodyB;
	if (pointCount <= 0 && (ia < 0 || ib < 0 || ia >= bodyCapacity || ib >= bodyCapacity))
	{
		pointCount = 0, ia = 0, ib = 0; // a destroyed contact whose entry lingers (prepareContactsKernel has the same guard)
	}
	pointCount = pointCount > 0 ? pointCount : 0;
	const V2 normal = v2(contact->normal[0], contact->normal[1]);
	const V2 tangent = rightPerp(normal);
	// (llc: the local centres staged in LDS beside the other body records -- wide_kernel.hip: wideIslandKernel --: gathered from the wire
	// bodies per constraint they were two more cache lines each, and most of that kernel's HBM traffic)
	V2 lcA, lcB;
	if (llc != nullptr)
	{
		const float2 a = llc[local.x], b = llc[local.y];
		lcA = v2(a.x, a.y), lcB = v2(b.x, b.y);
	}
	else
	{
		const s2amdBody* wa = wireBodies + ia;
		const s2amdBody* wb = wireBodies + ib;
		lcA = v2(wa->localCenter[0], wa->localCenter[1]);
		lcB = v2(wb->localCenter[0], wb->localCenter[1]);
	}
	const float2 massA = lmass[local.x], massB = lmass[local.y];
	const float mA = massA.x, iA = massA.y, mB = massB.x, iB = massB.y;
	const Rot qA = loadPose(lb, local.x).q, qB = loadPose(lb, local.y).q;
	r.h.ia = local.x, r.h.ib = local.y;
	r.h.mA = mA, r.h.iA = iA, r.h.mB = mB, r.h.iB = iB;
	r.h.normal = normal;
	r.h.friction = contact->friction;
	r.h.pointCount = pointCount;
	r.h.writeA = pointCount > 0 && (hostFlags[ia] & S2F_WRITE_VEL) != 0; // (the soft solvers are velocity-class sweeps)
	r.h.writeB = pointCount > 0 && (hostFlags[ib] & S2F_WRITE_VEL) != 0;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		r.an[j] = zero, r.r0[j] = zero, r.par[j] = zero, r.sf[j] = zero;
		r.imp[j] = make_float2(0.0f, 0.0f);
		if (j < pointCount)
		{
			const s2amdManifoldPoint* mp = contact->points + j;
			if (warmStart)
			{
				r.imp[j] = make_float2(mp->normalImpulse, mp->tangentImpulse);
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
			r.an[j] = make_float4(lA.x, lA.y, lB.x, lB.y);
			r.r0[j] = make_float4(rA.x, rA.y, rB.x, rB.y);
			r.par[j] = make_float4(adjustedSeparation, normalMass, tangentMass, separation);
		}
	}
	return r;
}

// The same with the bodies' rotations and inverse masses read from the wire bodies themselves (== what a kernel stages of them:
// body_ops.h unpackBodyOne): for a kernel that prepares its constraints before it has staged anything (wide_kernel.hip, S2_WIDE_SELF).
// The record's body slots (h.ia, h.ib) are left 0: the caller knows them.
template <int KIND>
S2_DEV SoftRegs<KIND> prepareSoftFromWireBodies(const s2amdContact* contact, const s2amdBody* wireBodies, const uint32_t* hostFlags, int bodyCapacity, int warmStart)
{
	int ia = contact->bodyA, ib = contact->bodyB;
	if (ia < 0 || ib < 0 || ia >= bodyCapacity || ib >= bodyCapacity)
	{
		ia = 0, ib = 0; // (a destroyed contact whose entry lingers: prepareSoftFromWire drops its points)
	}
	// {invMass, invI} of the two bodies as a two-entry table, the poses through the accessor above
	const float2 mass[2] = {make_float2(wireBodies[ia].invMass, wireBodies[ia].invI), make_float2(wireBodies[ib].invMass, wireBodies[ib].invI)};
	struct Pair
	{
		const s2amdBody* w;
		int ia, ib;
		S2_DEV float4 getDq(int i) const
		{
			const s2amdBody* b = w + (i == 0 ? ia : ib);
			return make_float4(0.0f, 0.0f, b->rot[0], b->rot[1]);
		}
	};
	const Pair poses{wireBodies, ia, ib};
	SoftRegs<KIND> r = prepareSoftFromWire<KIND>(contact, wireBodies, hostFlags, poses, mass, make_int2(0, 1), bodyCapacity, warmStart);
	r.h.ia = 0, r.h.ib = 0;
	return r;
}

// ... and in three stages, for a kernel that prepares several constraints per lane and wants each stage's loads of ALL of them in
// flight at once (wide_kernel.hip, S2_WIDE_SELF): what s2PrepareContacts_Soft reads of a contact (every load unconditional; a
// manifold point beyond pointCount is memory of the record, read and ignored), of its two bodies, and the arithmetic of
// prepareSoftFromWire above on those values, operation for operation.
struct WireContactRaw
{
	int ia, ib, pointCount;
	V2 normal;
	float friction;
	V2 anchorA[2], anchorB[2];
	float separation[2], normalImpulse[2], tangentImpulse[2];
};
struct WireBodiesRaw
{
	V2 lcA, lcB;
	Rot qA, qB;
	float mA, iA, mB, iB;
	uint32_t flagsA, flagsB;
};
S2_DEV WireContactRaw loadWireContact(const s2amdContact* contact, int bodyCapacity)
{
	WireContactRaw r;
	r.pointCount = contact->pointCount;
	r.ia = contact->bodyA, r.ib = contact->bodyB;
	if (r.ia < 0 || r.ib < 0 || r.ia >= bodyCapacity || r.ib >= bodyCapacity)
	{
		// (a destroyed contact whose entry lingers: prepareSoftFromWire drops its points; any other contact with points names live bodies)
		r.pointCount = 0, r.ia = 0, r.ib = 0;
	}
	r.pointCount = r.pointCount > 0 ? r.pointCount : 0;
	r.normal = v2(contact->normal[0], contact->normal[1]);
	r.friction = contact->friction;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		const s2amdManifoldPoint* mp = contact->points + j;
		r.anchorA[j] = v2(mp->localAnchorA[0], mp->localAnchorA[1]);
		r.anchorB[j] = v2(mp->localAnchorB[0], mp->localAnchorB[1]);
		r.separation[j] = mp->separation;
		r.normalImpulse[j] = mp->normalImpulse;
		r.tangentImpulse[j] = mp->tangentImpulse;
	}
	return r;
}
S2_DEV WireBodiesRaw loadWireBodies(const s2amdBody* wireBodies, const uint32_t* hostFlags, const WireContactRaw& c)
{
	const s2amdBody* wa = wireBodies + c.ia;
	const s2amdBody* wb = wireBodies + c.ib;
	WireBodiesRaw r;
	r.lcA = v2(wa->localCenter[0], wa->localCenter[1]), r.lcB = v2(wb->localCenter[0], wb->localCenter[1]);
	r.qA.s = wa->rot[0], r.qA.c = wa->rot[1], r.qB.s = wb->rot[0], r.qB.c = wb->rot[1];
	r.mA = wa->invMass, r.iA = wa->invI, r.mB = wb->invMass, r.iB = wb->invI;
	r.flagsA = hostFlags[c.ia], r.flagsB = hostFlags[c.ib];
	return r;
}
// live == false: an empty record (a free position of the slack layout)
S2_DEV void prepareSoftFromRaw(const WireContactRaw& c, const WireBodiesRaw& b, int warmStart, bool live, float4& nf, float4 (&an)[2], float4 (&par)[2], float2 (&imp)[2])
{
	const int pointCount = live ? c.pointCount : 0;
	const V2 normal = c.normal;
	const V2 tangent = rightPerp(normal);
	const float mA = b.mA, iA = b.iA, mB = b.mB, iB = b.iB;
	const bool writeA = pointCount > 0 && (b.flagsA & S2F_WRITE_VEL) != 0; // (the soft solvers are velocity-class sweeps)
	const bool writeB = pointCount > 0 && (b.flagsB & S2F_WRITE_VEL) != 0;
	const uint32_t bits = ((uint32_t)pointCount & 3u) | (writeA ? S2C_WRITE_A : 0u) | (writeB ? S2C_WRITE_B : 0u);
	nf = live ? make_float4(normal.x, normal.y, c.friction, fromBits(bits)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		an[j] = zero, par[j] = zero;
		imp[j] = make_float2(0.0f, 0.0f);
		if (j < pointCount)
		{
			if (warmStart)
			{
				imp[j] = make_float2(c.normalImpulse[j], c.tangentImpulse[j]);
			}
			V2 lA = sub(c.anchorA[j], b.lcA);
			V2 lB = sub(c.anchorB[j], b.lcB);
			V2 rA = rotate(b.qA, lA);
			V2 rB = rotate(b.qB, lB);
			float separation = c.separation[j];
			float adjustedSeparation = separation - dot(sub(rB, rA), normal);
			float rtA = cross(rA, tangent);
			float rtB = cross(rB, tangent);
			float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
			float tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
			float normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
			an[j] = make_float4(lA.x, lA.y, lB.x, lB.y);
			par[j] = make_float4(adjustedSeparation, normalMass, tangentMass, separation);
		}
	}
}
