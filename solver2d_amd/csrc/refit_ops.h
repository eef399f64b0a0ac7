// Stage 4 of s2World_Step (src/world.c:259-301) per shape and per body, as device functions: broadphase.hip's kernels call them, and so do
// the stage-4 blocks that ride in a step's epilogue launch (contact_kernels.hip: storeImpulsesKernel).
#pragma once

#include "s2_device.h"

#include "solver2d_amd.h"

#define S2_SPECULATIVE_DISTANCE (4.0f * S2_LINEAR_SLOP) // constants.h:8
#define S2_AABB_MARGIN 0.1f								 // constants.h:9

struct Xf
{
	V2 p;
	Rot q;
};

S2_DEV V2 transformPoint(Xf xf, V2 p) // math.h:350-356
{
	float x = (xf.q.c * p.x - xf.q.s * p.y) + xf.p.x;
	float y = (xf.q.s * p.x + xf.q.c * p.y) + xf.p.y;
	return v2(x, y);
}
S2_DEV V2 vmin(V2 a, V2 b) { return v2(S2_MINF(a.x, b.x), S2_MINF(a.y, b.y)); }
S2_DEV V2 vmax(V2 a, V2 b) { return v2(S2_MAXF(a.x, b.x), S2_MAXF(a.y, b.y)); }

// s2Shape_ComputeAABB -> src/geometry.c:288-339, then src/world.c:283-296
// one shape of a non-static body: tight AABB + speculative margin, fat AABB re-inflated when it was left; returns `enlarged`
S2_DEV int refitShapeOne(Rot q, s2amdShape* sh, V2 origin)
{
	Xf xf;
	xf.p = origin;
	xf.q = q;
	V2 lower, upper;
	V2 v0 = v2(sh->vertices[0][0], sh->vertices[0][1]);
	V2 v1 = v2(sh->vertices[1][0], sh->vertices[1][1]);
	switch (sh->type)
	{
		case S2AMD_SHAPE_CIRCLE:
		{
			V2 p = transformPoint(xf, v0);
			float r = sh->radius;
			lower = v2(p.x - r, p.y - r);
			upper = v2(p.x + r, p.y + r);
			break;
		}
		case S2AMD_SHAPE_CAPSULE:
		{
			V2 a = transformPoint(xf, v0), c = transformPoint(xf, v1);
			V2 r = v2(sh->radius, sh->radius);
			lower = sub(vmin(a, c), r);
			upper = add(vmax(a, c), r);
			break;
		}
		case S2AMD_SHAPE_POLYGON:
		{
			lower = transformPoint(xf, v0);
			upper = lower;
			for (int i = 1; i < sh->count; ++i)
			{
				V2 v = transformPoint(xf, v2(sh->vertices[i][0], sh->vertices[i][1]));
				lower = vmin(lower, v);
				upper = vmax(upper, v);
			}
			V2 r = v2(sh->radius, sh->radius);
			lower = sub(lower, r);
			upper = add(upper, r);
			break;
		}
		case S2AMD_SHAPE_SEGMENT:
		{
			V2 a = transformPoint(xf, v0), c = transformPoint(xf, v1);
			lower = vmin(a, c);
			upper = vmax(a, c);
			break;
		}
		default:
			lower = xf.p;
			upper = xf.p;
			break;
	}
	float a0 = lower.x - S2_SPECULATIVE_DISTANCE, a1 = lower.y - S2_SPECULATIVE_DISTANCE;
	float a2 = upper.x + S2_SPECULATIVE_DISTANCE, a3 = upper.y + S2_SPECULATIVE_DISTANCE;
	sh->aabb[0] = a0, sh->aabb[1] = a1, sh->aabb[2] = a2, sh->aabb[3] = a3;
	bool contains = sh->fatAABB[0] <= a0 && sh->fatAABB[1] <= a1 && a2 <= sh->fatAABB[2] && a3 <= sh->fatAABB[3];
	int enlarged = 0;
	if (contains == false)
	{
		sh->fatAABB[0] = a0 - S2_AABB_MARGIN;
		sh->fatAABB[1] = a1 - S2_AABB_MARGIN;
		sh->fatAABB[2] = a2 + S2_AABB_MARGIN;
		sh->fatAABB[3] = a3 + S2_AABB_MARGIN;
		enlarged = 1;
	}
	sh->enlarged = enlarged;
	return enlarged;
}


// one shape of the resident world: refit (non-static body) from the body's pose `position, q` (the wire body's, or the SoA records the
// step has just finished: the same values, packBodyOne copies them); returns its `enlarged` flag
S2_DEV int stage4ShapeOne(const s2amdBody* bodies, int nb, s2amdShape* sh, const BodyView* soa)
{
	if (sh->type == S2AMD_SHAPE_FREE || sh->body < 0 || sh->body >= nb)
	{
		return 0;
	}
	const s2amdBody* b = bodies + sh->body;
	if (b->type == S2AMD_BODY_FREE || b->type == S2AMD_BODY_STATIC)
	{
		return sh->enlarged != 0 ? 1 : 0; // a static shape keeps the flag its creation gave it
	}
	Rot q;
	V2 p;
	if (soa != nullptr)
	{
		const float4 d = soa->dq[sh->body];
		const float2 pos = soa->pos[sh->body];
		q.s = d.z, q.c = d.w;
		p = v2(pos.x, pos.y);
	}
	else
	{
		q.s = b->rot[0], q.c = b->rot[1];
		p = v2(b->position[0], b->position[1]);
	}
	const V2 o = sub(p, rotate(q, v2(b->localCenter[0], b->localCenter[1])));
	return refitShapeOne(q, sh, o);
}
// one body: the origin for the next stage 3, applied forces consumed (src/world.c:274-275)
S2_DEV void stage4BodyOne(s2amdBody* bodies, int i, float2* origins, const BodyView* soa)
{
	s2amdBody* b = bodies + i;
	if (b->type != S2AMD_BODY_FREE && b->type != S2AMD_BODY_STATIC)
	{
		Rot q;
		V2 p;
		if (soa != nullptr)
		{
			const float4 d = soa->dq[i];
			const float2 pos = soa->pos[i];
			q.s = d.z, q.c = d.w;
			p = v2(pos.x, pos.y);
		}
		else
		{
			q.s = b->rot[0], q.c = b->rot[1];
			p = v2(b->position[0], b->position[1]);
		}
		const V2 o = sub(p, rotate(q, v2(b->localCenter[0], b->localCenter[1])));
		origins[i] = make_float2(o.x, o.y);
		b->force[0] = 0.0f;
		b->force[1] = 0.0f;
		b->torque = 0.0f;
	}
}
