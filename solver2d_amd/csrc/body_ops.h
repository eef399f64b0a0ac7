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

// wire body -> SoA records + the per-step constants of the velocity integrator (solve_common.c:30-41)
S2_DEV void unpackBodyOne(const BodyView& b, const s2amdBody* wire, const uint32_t* hostFlags, const StepConsts& sc, float h, int i)
{
	if (i >= b.capacity)
	{
		return;
	}
	const s2amdBody* w = wire + i;
	int type = w->type;
	uint32_t flags = hostFlags[i] & (S2F_WRITE_VEL | S2F_WRITE_POS | S2F_IN_GROUP);
	if (type != S2AMD_BODY_FREE)
	{
		flags |= S2F_LIVE;
		if (type == S2AMD_BODY_DYNAMIC)
		{
			flags |= S2F_DYNAMIC;
		}
		if (type != S2AMD_BODY_STATIC)
		{
			flags |= S2F_MOVES;
		}
	}
	b.flags[i] = flags;
	b.vel[i] = make_float4(w->linearVelocity[0], w->linearVelocity[1], w->angularVelocity, 0.0f);
	b.dq[i] = make_float4(w->deltaPosition[0], w->deltaPosition[1], w->rot[0], w->rot[1]);
	b.pos[i] = make_float2(w->position[0], w->position[1]);
	b.massInv[i] = make_float2(w->invMass, w->invI);

	V2 gravity = v2(sc.gravityX, sc.gravityY);
	V2 force = v2(w->force[0], w->force[1]);
	V2 inner = mulAdd(force, w->mass * w->gravityScale, gravity);
	V2 a = mulSV(h * w->invMass, inner);
	float aw = h * w->invI * w->torque;
	float ld = 1.0f / (1.0f + h * w->linearDamping);
	float ad = 1.0f / (1.0f + h * w->angularDamping);
	b.integ[i] = make_float4(a.x, a.y, aw, ld);
	b.angDamp[i] = ad;
}

// SoA records -> wire body (the fields a solver writes: body.h:16-76)
S2_DEV void packBodyOne(const BodyView& b, s2amdBody* wire, int i)
{
	if (i >= b.capacity)
	{
		return;
	}
	if ((b.flags[i] & S2F_LIVE) == 0)
	{
		return;
	}
	s2amdBody* w = wire + i;
	float4 v = b.vel[i];
	float4 d = b.dq[i];
	float2 p = b.pos[i];
	w->position[0] = p.x, w->position[1] = p.y;
	w->rot[0] = d.z, w->rot[1] = d.w;
	w->linearVelocity[0] = v.x, w->linearVelocity[1] = v.y;
	w->angularVelocity = v.z;
	w->deltaPosition[0] = d.x, w->deltaPosition[1] = d.y;
}
