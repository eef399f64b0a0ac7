// Per-body device functions shared by the streaming body kernels (body_kernels.hip) and the LDS
// group kernel (group_kernel.hip).  `ba` addresses the two mutable 16-byte body records (HBM SoA or
// LDS), `li` is the index in that accessor, `g`/`gi` the global view and pool slot for the
// read-only per-step constants.
#pragma once

#include "constraint_ops.h"

// s2IntegrateVelocities: solve_common.c:10-45 (constants precomputed by unpackBodiesKernel)
template <class BA> S2_DEV void integrateVelocitiesOne(const BA& ba, int li, const BodyView& g, int gi)
{
	if ((g.flags[gi] & S2F_DYNAMIC) == 0)
	{
		return;
	}
	float4 v = ba.getVel(li);
	float4 k = g.integ[gi];
	float ad = g.angDamp[gi];
	V2 lv = add(v2(v.x, v.y), v2(k.x, k.y));
	float w = v.z + k.z;
	lv = mulSV(k.w, lv);
	w *= ad;
	ba.setVel(li, make_float4(lv.x, lv.y, w, 0.0f));
}

// s2IntegratePositions: solve_common.c:47-68
template <class BA> S2_DEV void integratePositionsOne(const BA& ba, int li, const BodyView& g, int gi, float h)
{
	if ((g.flags[gi] & S2F_MOVES) == 0)
	{
		return;
	}
	float4 v = ba.getVel(li);
	float4 d = ba.getDq(li);
	V2 dp = mulAdd(v2(d.x, d.y), h, v2(v.x, v.y));
	Rot q;
	q.s = d.z, q.c = d.w;
	q = integrateRot(q, h * v.z);
	ba.setDq(li, make_float4(dp.x, dp.y, q.s, q.c));
}

// s2FinalizePositions: solve_common.c:70-91; dynamicOnly = the XPBD variant, solve_xpbd.c:496-512.
// writePos == false for read-only replicas inside a group (only the owner updates g.pos).
template <class BA> S2_DEV void finalizePositionsOne(const BA& ba, int li, const BodyView& g, int gi, int dynamicOnly, bool writePos)
{
	uint32_t need = dynamicOnly ? S2F_DYNAMIC : S2F_MOVES;
	if ((g.flags[gi] & need) == 0)
	{
		return;
	}
	float4 d = ba.getDq(li);
	if (writePos)
	{
		float2 p = g.pos[gi];
		V2 np = add(v2(p.x, p.y), v2(d.x, d.y));
		g.pos[gi] = make_float2(np.x, np.y);
	}
	ba.setDq(li, make_float4(0.0f, 0.0f, d.z, d.w));
}

// XPBD sub-step head: solve_xpbd.c:411-449 (every non-static body, kinematic included).
// dq0 receives {deltaPosition0, rot0}.
template <class BA> S2_DEV void xpbdIntegrateOne(const BA& ba, float4* dq0, int li, const BodyView& g, int gi, float h)
{
	if ((g.flags[gi] & S2F_MOVES) == 0)
	{
		return;
	}
	float4 v = ba.getVel(li);
	float4 k = g.integ[gi];
	float ad = g.angDamp[gi];
	V2 lv = add(v2(v.x, v.y), v2(k.x, k.y));
	float w = v.z + k.z;
	lv = mulSV(k.w, lv);
	w *= ad;
	ba.setVel(li, make_float4(lv.x, lv.y, w, 0.0f));
	float4 d = ba.getDq(li);
	dq0[li] = d;
	V2 dp = mulAdd(v2(d.x, d.y), h, lv);
	Rot q;
	q.s = d.z, q.c = d.w;
	q = integrateRot(q, h * w);
	ba.setDq(li, make_float4(dp.x, dp.y, q.s, q.c));
}

// XPBD velocity projection: solve_xpbd.c:465-489 (dynamic bodies only)
template <class BA> S2_DEV void xpbdProjectOne(const BA& ba, const float4* dq0, int li, const BodyView& g, int gi, float inv_h)
{
	if ((g.flags[gi] & S2F_DYNAMIC) == 0)
	{
		return;
	}
	float4 d = ba.getDq(li);
	float4 d0 = dq0[li];
	V2 lv = mulSV(inv_h, sub(v2(d.x, d.y), v2(d0.x, d0.y)));
	Rot q0, q1;
	q0.s = d0.z, q0.c = d0.w;
	q1.s = d.z, q1.c = d.w;
	float w = computeAngularVelocity(q0, q1, inv_h);
	ba.setVel(li, make_float4(lv.x, lv.y, w, 0.0f));
}
