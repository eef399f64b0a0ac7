// TEST INFRASTRUCTURE -- CPU restatement of the reference's narrow phase.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Restates, operation for operation (fp32, no contraction):
//   Stage 3 of s2World_Step "update contacts"           src/world.c:132-168
//   s2UpdateContact (manifold + id matching)             src/contact.c:296-358
//   manifold functions per shape-type pair               src/contact.c:66-151 (register), src/manifold.c
//     s2CollideCircles :16-49, s2CollideCapsuleAndCircle :51-110, s2CollidePolygonAndCircle :113-222,
//     s2ClipPolygons :248-399, s2FindMaxSeparation :402-438, s2PolygonSAT :441-506, s2CollidePolygons :509-650,
//     capsule / segment wrappers :224-246, :652-663 via s2MakeCapsule src/geometry.c:100-115
//   GJK distance with simplex cache                      src/distance.c:120-604 (s2ShapeDistance :485)
//   transforms                                           include/solver2d/math.h:103-106, :291-383, src/math.c:40-80
//
// Pinned against the unmodified reference by tests/test_narrowphase_oracle.py (captures at the end of Stage 2 and
// at solver entry, oracle/ref_hook.c) and by the committed fixtures tests/golden/np_*.npz.
#include "solver2d_amd.h"

#include <float.h>
#include <math.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

typedef struct V2
{
	float x, y;
} V2;
typedef struct Rot
{
	float s, c;
} Rot;
typedef struct Xf
{
	V2 p;
	Rot q;
} Xf;

#define LINEAR_SLOP 0.005f
#define SPECULATIVE_DISTANCE (4.0f * LINEAR_SLOP)
#define MAX_VERTS 8
#define MAKE_ID(A, B) ((uint16_t)(((uint8_t)(A) << 8) | (uint8_t)(B))) // manifold.h:17

static inline V2 v2(float x, float y) { V2 r = {x, y}; return r; }
static inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
static inline float cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
static inline V2 crossVS(V2 v, float s) { return v2(s * v.y, -s * v.x); }
static inline V2 crossSV(float s, V2 v) { return v2(-s * v.y, s * v.x); }
static inline V2 rightPerp(V2 v) { return v2(v.y, -v.x); }
static inline V2 add(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
static inline V2 sub(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
static inline V2 neg(V2 a) { return v2(-a.x, -a.y); }
static inline V2 mulAdd(V2 a, float s, V2 b) { return v2(a.x + s * b.x, a.y + s * b.y); }
static inline V2 mulSub(V2 a, float s, V2 b) { return v2(a.x - s * b.x, a.y - s * b.y); }
static inline V2 lerp(V2 a, V2 b, float t) { return v2(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y)); } // math.h:103
static inline float length(V2 v) { return sqrtf(v.x * v.x + v.y * v.y); }
static inline float distance(V2 a, V2 b) // math.h:180-185
{
	float dx = b.x - a.x, dy = b.y - a.y;
	return sqrtf(dx * dx + dy * dy);
}
static inline V2 normalize(V2 v) // src/math.c:40-51
{
	float len = length(v);
	if (len < 0.001f * FLT_EPSILON)
	{
		return v2(0.0f, 0.0f);
	}
	float inv = 1.0f / len;
	return v2(inv * v.x, inv * v.y);
}
static inline V2 normalizeChecked(V2 v) // src/math.c:54-66
{
	float len = length(v);
	if (len < FLT_EPSILON)
	{
		return v2(0.0f, 0.0f);
	}
	float inv = 1.0f / len;
	return v2(inv * v.x, inv * v.y);
}
static inline V2 lengthAndNormalize(float* len, V2 v) // src/math.c:68-80
{
	*len = length(v);
	if (*len < FLT_EPSILON)
	{
		return v2(0.0f, 0.0f);
	}
	float inv = 1.0f / *len;
	return v2(inv * v.x, inv * v.y);
}
static inline V2 rotate(Rot q, V2 v) { return v2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
static inline V2 invRotate(Rot q, V2 v) { return v2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
static inline V2 transformPoint(Xf xf, V2 p) // math.h:350-356
{
	float x = (xf.q.c * p.x - xf.q.s * p.y) + xf.p.x;
	float y = (xf.q.s * p.x + xf.q.c * p.y) + xf.p.y;
	return v2(x, y);
}
static inline V2 invTransformPoint(Xf xf, V2 p) // math.h:359-364
{
	float vx = p.x - xf.p.x, vy = p.y - xf.p.y;
	return v2(xf.q.c * vx + xf.q.s * vy, -xf.q.s * vx + xf.q.c * vy);
}
static inline Xf invMulTransforms(Xf A, Xf B) // math.h:378-384 with s2InvMulRot :307-317
{
	Xf C;
	C.q.s = A.q.c * B.q.s - A.q.s * B.q.c;
	C.q.c = A.q.c * B.q.c + A.q.s * B.q.s;
	C.p = invRotate(A.q, sub(B.p, A.p));
	return C;
}

typedef struct Poly
{
	V2 vertices[MAX_VERTS], normals[MAX_VERTS];
	float radius;
	int count;
} Poly;

typedef struct MPoint
{
	V2 localAnchorA, localAnchorB;
	float separation;
	uint16_t id;
} MPoint;
typedef struct Manifold
{
	MPoint points[2];
	V2 normal;
	int pointCount;
} Manifold;

typedef struct Cache
{
	float metric;
	int count;
	int indexA[3], indexB[3];
} Cache;

static Manifold emptyManifold(void)
{
	Manifold m;
	memset(&m, 0, sizeof(m));
	return m;
}

// ---- GJK (src/distance.c) ---------------------------------------------------------------------------------
typedef struct SVertex
{
	V2 wA, wB, w;
	float a;
	int indexA, indexB;
} SVertex;
typedef struct Simplex
{
	SVertex v[3];
	int count;
} Simplex;

static int findSupport(const V2* verts, int count, V2 d) // :120-135
{
	int best = 0;
	float bestValue = dot(verts[0], d);
	for (int i = 1; i < count; ++i)
	{
		float value = dot(verts[i], d);
		if (value > bestValue)
		{
			best = i;
			bestValue = value;
		}
	}
	return best;
}

static V2 weight2(float a1, V2 w1, float a2, V2 w2) { return v2(a1 * w1.x + a2 * w2.x, a1 * w1.y + a2 * w2.y); } // :110-113
static V2 weight3(float a1, V2 w1, float a2, V2 w2, float a3, V2 w3)												   // :115-118
{
	return v2(a1 * w1.x + a2 * w2.x + a3 * w3.x, a1 * w1.y + a2 * w2.y + a3 * w3.y);
}

static void solveSimplex2(Simplex* s) // :337-367
{
	V2 w1 = s->v[0].w, w2 = s->v[1].w;
	V2 e12 = sub(w2, w1);
	float d12_2 = -dot(w1, e12);
	if (d12_2 <= 0.0f)
	{
		s->v[0].a = 1.0f;
		s->count = 1;
		return;
	}
	float d12_1 = dot(w2, e12);
	if (d12_1 <= 0.0f)
	{
		s->v[1].a = 1.0f;
		s->count = 1;
		s->v[0] = s->v[1];
		return;
	}
	float inv_d12 = 1.0f / (d12_1 + d12_2);
	s->v[0].a = d12_1 * inv_d12;
	s->v[1].a = d12_2 * inv_d12;
	s->count = 2;
}

static void solveSimplex3(Simplex* s) // :369-473
{
	V2 w1 = s->v[0].w, w2 = s->v[1].w, w3 = s->v[2].w;
	V2 e12 = sub(w2, w1);
	float w1e12 = dot(w1, e12), w2e12 = dot(w2, e12);
	float d12_1 = w2e12, d12_2 = -w1e12;
	V2 e13 = sub(w3, w1);
	float w1e13 = dot(w1, e13), w3e13 = dot(w3, e13);
	float d13_1 = w3e13, d13_2 = -w1e13;
	V2 e23 = sub(w3, w2);
	float w2e23 = dot(w2, e23), w3e23 = dot(w3, e23);
	float d23_1 = w3e23, d23_2 = -w2e23;
	float n123 = cross(e12, e13);
	float d123_1 = n123 * cross(w2, w3);
	float d123_2 = n123 * cross(w3, w1);
	float d123_3 = n123 * cross(w1, w2);

	if (d12_2 <= 0.0f && d13_2 <= 0.0f)
	{
		s->v[0].a = 1.0f;
		s->count = 1;
		return;
	}
	if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f)
	{
		float inv = 1.0f / (d12_1 + d12_2);
		s->v[0].a = d12_1 * inv;
		s->v[1].a = d12_2 * inv;
		s->count = 2;
		return;
	}
	if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f)
	{
		float inv = 1.0f / (d13_1 + d13_2);
		s->v[0].a = d13_1 * inv;
		s->v[2].a = d13_2 * inv;
		s->count = 2;
		s->v[1] = s->v[2];
		return;
	}
	if (d12_1 <= 0.0f && d23_2 <= 0.0f)
	{
		s->v[1].a = 1.0f;
		s->count = 1;
		s->v[0] = s->v[1];
		return;
	}
	if (d13_1 <= 0.0f && d23_1 <= 0.0f)
	{
		s->v[2].a = 1.0f;
		s->count = 1;
		s->v[0] = s->v[2];
		return;
	}
	if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f)
	{
		float inv = 1.0f / (d23_1 + d23_2);
		s->v[1].a = d23_1 * inv;
		s->v[2].a = d23_2 * inv;
		s->count = 2;
		s->v[0] = s->v[2];
		return;
	}
	float inv = 1.0f / (d123_1 + d123_2 + d123_3);
	s->v[0].a = d123_1 * inv;
	s->v[1].a = d123_2 * inv;
	s->v[2].a = d123_3 * inv;
	s->count = 3;
}

typedef struct DistanceOutput
{
	V2 pointA, pointB;
	float distance;
} DistanceOutput;

// s2ShapeDistance with identity transforms and useRadii == false: the only form the manifold code calls (:523-530)
static DistanceOutput shapeDistance(Cache* cache, const V2* vertsA, int countA, const V2* vertsB, int countB)
{
	Simplex simplex;
	memset(&simplex, 0, sizeof(simplex));
	// s2MakeSimplexFromCache :172-214 (the identity transform adds +0: kept, it canonicalises -0.0)
	const Xf identity = {{0.0f, 0.0f}, {0.0f, 1.0f}};
	simplex.count = cache->count;
	for (int i = 0; i < simplex.count; ++i)
	{
		SVertex* v = simplex.v + i;
		v->indexA = cache->indexA[i];
		v->indexB = cache->indexB[i];
		v->wA = transformPoint(identity, vertsA[v->indexA]);
		v->wB = transformPoint(identity, vertsB[v->indexB]);
		v->w = sub(v->wB, v->wA);
		v->a = -1.0f;
	}
	if (simplex.count == 0)
	{
		SVertex* v = simplex.v;
		v->indexA = 0;
		v->indexB = 0;
		v->wA = transformPoint(identity, vertsA[0]);
		v->wB = transformPoint(identity, vertsB[0]);
		v->w = sub(v->wB, v->wA);
		v->a = 1.0f;
		simplex.count = 1;
	}

	int saveA[3], saveB[3];
	int iter = 0;
	while (iter < 20) // :511
	{
		int saveCount = simplex.count;
		for (int i = 0; i < saveCount; ++i)
		{
			saveA[i] = simplex.v[i].indexA;
			saveB[i] = simplex.v[i].indexB;
		}
		if (simplex.count == 2)
		{
			solveSimplex2(&simplex);
		}
		else if (simplex.count == 3)
		{
			solveSimplex3(&simplex);
		}
		if (simplex.count == 3)
		{
			break;
		}
		// s2ComputeSimplexSearchDirection :228-254
		V2 d;
		if (simplex.count == 1)
		{
			d = neg(simplex.v[0].w);
		}
		else
		{
			V2 e12 = sub(simplex.v[1].w, simplex.v[0].w);
			float sgn = cross(e12, neg(simplex.v[0].w));
			d = sgn > 0.0f ? crossSV(1.0f, e12) : crossVS(e12, 1.0f);
		}
		if (dot(d, d) < FLT_EPSILON * FLT_EPSILON)
		{
			break;
		}
		SVertex* vertex = simplex.v + simplex.count;
		vertex->indexA = findSupport(vertsA, countA, invRotate(identity.q, neg(d)));
		vertex->wA = transformPoint(identity, vertsA[vertex->indexA]);
		vertex->indexB = findSupport(vertsB, countB, invRotate(identity.q, d));
		vertex->wB = transformPoint(identity, vertsB[vertex->indexB]);
		vertex->w = sub(vertex->wB, vertex->wA);
		++iter;
		int duplicate = 0;
		for (int i = 0; i < saveCount; ++i)
		{
			if (vertex->indexA == saveA[i] && vertex->indexB == saveB[i])
			{
				duplicate = 1;
				break;
			}
		}
		if (duplicate)
		{
			break;
		}
		++simplex.count;
	}

	DistanceOutput out;
	// s2ComputeSimplexWitnessPoints :277-306
	if (simplex.count == 1)
	{
		out.pointA = simplex.v[0].wA;
		out.pointB = simplex.v[0].wB;
	}
	else if (simplex.count == 2)
	{
		out.pointA = weight2(simplex.v[0].a, simplex.v[0].wA, simplex.v[1].a, simplex.v[1].wA);
		out.pointB = weight2(simplex.v[0].a, simplex.v[0].wB, simplex.v[1].a, simplex.v[1].wB);
	}
	else
	{
		out.pointA = weight3(simplex.v[0].a, simplex.v[0].wA, simplex.v[1].a, simplex.v[1].wA, simplex.v[2].a, simplex.v[2].wA);
		out.pointB = out.pointA;
	}
	out.distance = distance(out.pointA, out.pointB);

	// s2MakeSimplexCache :216-226 with s2Simplex_Metric :149-170
	if (simplex.count == 1)
	{
		cache->metric = 0.0f;
	}
	else if (simplex.count == 2)
	{
		cache->metric = distance(simplex.v[0].w, simplex.v[1].w);
	}
	else
	{
		cache->metric = cross(sub(simplex.v[1].w, simplex.v[0].w), sub(simplex.v[2].w, simplex.v[0].w));
	}
	cache->count = simplex.count;
	for (int i = 0; i < simplex.count; ++i)
	{
		cache->indexA[i] = (uint8_t)simplex.v[i].indexA;
		cache->indexB[i] = (uint8_t)simplex.v[i].indexB;
	}
	return out;
}

// ---- manifolds (src/manifold.c) ------------------------------------------------------------------------------
static Manifold collideCircles(V2 pointA0, float radiusA, Xf xfA, V2 pointB0, float radiusB, Xf xfB) // :16-49
{
	Manifold m = emptyManifold();
	Xf xf = invMulTransforms(xfA, xfB);
	V2 pointA = pointA0;
	V2 pointB = transformPoint(xf, pointB0);
	float dist;
	V2 normal = lengthAndNormalize(&dist, sub(pointB, pointA));
	float separation = dist - radiusA - radiusB;
	if (separation > SPECULATIVE_DISTANCE)
	{
		return m;
	}
	V2 cA = mulAdd(pointA, radiusA, normal);
	V2 cB = mulAdd(pointB, -radiusB, normal);
	V2 contactPointA = lerp(cA, cB, 0.5f);
	m.normal = rotate(xfA.q, normal);
	m.points[0].localAnchorA = contactPointA;
	m.points[0].localAnchorB = invTransformPoint(xf, contactPointA);
	m.points[0].separation = separation;
	m.points[0].id = 0;
	m.pointCount = 1;
	return m;
}

static Manifold collideCapsuleAndCircle(V2 p1, V2 p2, float radiusA, Xf xfA, V2 pointB0, float radiusB, Xf xfB) // :51-110
{
	Manifold m = emptyManifold();
	Xf xf = invMulTransforms(xfA, xfB);
	V2 pB = transformPoint(xf, pointB0);
	V2 e = sub(p2, p1);
	V2 pA;
	float s1 = dot(sub(pB, p1), e);
	float s2 = dot(sub(p2, pB), e);
	if (s1 < 0.0f)
	{
		pA = p1;
	}
	else if (s2 < 0.0f)
	{
		pA = p2;
	}
	else
	{
		float s = s1 / dot(e, e);
		pA = mulAdd(p1, s, e);
	}
	float dist;
	V2 normal = lengthAndNormalize(&dist, sub(pB, pA));
	float separation = dist - radiusA - radiusB;
	if (separation > SPECULATIVE_DISTANCE)
	{
		return m;
	}
	V2 cA = mulAdd(pA, radiusA, normal);
	V2 cB = mulAdd(pB, -radiusB, normal);
	V2 contactPointA = lerp(cA, cB, 0.5f);
	m.normal = rotate(xfA.q, normal);
	m.points[0].localAnchorA = contactPointA;
	m.points[0].localAnchorB = invTransformPoint(xf, contactPointA);
	m.points[0].separation = separation;
	m.points[0].id = 0;
	m.pointCount = 1;
	return m;
}

static Manifold collidePolygonAndCircle(const Poly* polygonA, Xf xfA, V2 pointB0, float radiusB, Xf xfB) // :113-222
{
	Manifold m = emptyManifold();
	Xf xf = invMulTransforms(xfA, xfB);
	V2 c = transformPoint(xf, pointB0);
	float radiusA = polygonA->radius;
	float radius = radiusA + radiusB;
	int normalIndex = 0;
	float separation = -FLT_MAX;
	int vertexCount = polygonA->count;
	const V2* vertices = polygonA->vertices;
	const V2* normals = polygonA->normals;
	for (int i = 0; i < vertexCount; ++i)
	{
		float s = dot(normals[i], sub(c, vertices[i]));
		if (s > separation)
		{
			separation = s;
			normalIndex = i;
		}
	}
	if (separation > radius + SPECULATIVE_DISTANCE)
	{
		return m;
	}
	int vertIndex1 = normalIndex;
	int vertIndex2 = vertIndex1 + 1 < vertexCount ? vertIndex1 + 1 : 0;
	V2 v1 = vertices[vertIndex1], v2_ = vertices[vertIndex2];
	float u1 = dot(sub(c, v1), sub(v2_, v1));
	float u2 = dot(sub(c, v2_), sub(v1, v2_));
	if ((u1 < 0.0f && separation > FLT_EPSILON) || (u2 < 0.0f && separation > FLT_EPSILON))
	{
		// closest to a vertex and safely outside: v1 is tested first (:158, :179)
		V2 v = (u1 < 0.0f && separation > FLT_EPSILON) ? v1 : v2_;
		V2 normal = normalize(sub(c, v));
		separation = dot(sub(c, v), normal);
		if (separation > radius + SPECULATIVE_DISTANCE)
		{
			return m;
		}
		V2 cA = mulAdd(v, radiusA, normal);
		V2 cB = mulSub(c, radiusB, normal);
		V2 contactPointA = lerp(cA, cB, 0.5f);
		m.normal = rotate(xfA.q, normal);
		m.points[0].localAnchorA = contactPointA;
		m.points[0].localAnchorB = invTransformPoint(xf, contactPointA);
		m.points[0].separation = dot(sub(cB, cA), normal);
		m.points[0].id = 0;
		m.pointCount = 1;
	}
	else
	{
		V2 normal = normals[normalIndex];
		m.normal = rotate(xfA.q, normal);
		V2 cA = mulAdd(c, radiusA - dot(sub(c, v1), normal), normal);
		V2 cB = mulSub(c, radiusB, normal);
		V2 contactPointA = lerp(cA, cB, 0.5f);
		m.points[0].localAnchorA = contactPointA;
		m.points[0].localAnchorB = invTransformPoint(xf, contactPointA);
		m.points[0].separation = separation - radius;
		m.points[0].id = 0;
		m.pointCount = 1;
	}
	return m;
}

static Manifold clipPolygons(const Poly* polyA, const Poly* polyB, int edgeA, int edgeB, int flip) // :248-399
{
	Manifold m = emptyManifold();
	const Poly *poly1, *poly2;
	int i11, i12, i21, i22;
	if (flip)
	{
		poly1 = polyB, poly2 = polyA;
		i11 = edgeB, i12 = edgeB + 1 < polyB->count ? edgeB + 1 : 0;
		i21 = edgeA, i22 = edgeA + 1 < polyA->count ? edgeA + 1 : 0;
	}
	else
	{
		poly1 = polyA, poly2 = polyB;
		i11 = edgeA, i12 = edgeA + 1 < polyA->count ? edgeA + 1 : 0;
		i21 = edgeB, i22 = edgeB + 1 < polyB->count ? edgeB + 1 : 0;
	}
	V2 normal = poly1->normals[i11];
	V2 v11 = poly1->vertices[i11], v12 = poly1->vertices[i12];
	V2 v21 = poly2->vertices[i21], v22 = poly2->vertices[i22];
	V2 tangent = crossSV(1.0f, normal);
	float lower1 = 0.0f;
	float upper1 = dot(sub(v12, v11), tangent);
	float upper2 = dot(sub(v21, v11), tangent);
	float lower2 = dot(sub(v22, v11), tangent);

	V2 vLower = (lower2 < lower1 && upper2 - lower2 > FLT_EPSILON) ? lerp(v22, v21, (lower1 - lower2) / (upper2 - lower2)) : v22;
	V2 vUpper = (upper2 > upper1 && upper2 - lower2 > FLT_EPSILON) ? lerp(v22, v21, (upper1 - lower2) / (upper2 - lower2)) : v21;

	float separationLower = dot(sub(vLower, v11), normal);
	float separationUpper = dot(sub(vUpper, v11), normal);
	float r1 = poly1->radius, r2 = poly2->radius;
	vLower = mulAdd(vLower, 0.5f * (r1 - r2 - separationLower), normal);
	vUpper = mulAdd(vUpper, 0.5f * (r1 - r2 - separationUpper), normal);
	float radius = r1 + r2;
	// the reference's "if (separation < -0.5f) separation += 0.0f" (:347-350 ...) changes no bits and is left out
	if (!flip)
	{
		m.normal = normal;
		m.points[0].localAnchorA = vLower;
		m.points[0].separation = separationLower - radius;
		m.points[0].id = MAKE_ID(i11, i22);
		m.points[1].localAnchorA = vUpper;
		m.points[1].separation = separationUpper - radius;
		m.points[1].id = MAKE_ID(i12, i21);
	}
	else
	{
		m.normal = neg(normal);
		m.points[0].localAnchorA = vUpper;
		m.points[0].separation = separationUpper - radius;
		m.points[0].id = MAKE_ID(i21, i12);
		m.points[1].localAnchorA = vLower;
		m.points[1].separation = separationLower - radius;
		m.points[1].id = MAKE_ID(i22, i11);
	}
	m.pointCount = 2;
	return m;
}

static float findMaxSeparation(int* edgeIndex, const Poly* poly1, const Poly* poly2) // :402-438
{
	int bestIndex = 0;
	float maxSeparation = -FLT_MAX;
	for (int i = 0; i < poly1->count; ++i)
	{
		V2 n = poly1->normals[i], v1 = poly1->vertices[i];
		float si = FLT_MAX;
		for (int j = 0; j < poly2->count; ++j)
		{
			float sij = dot(n, sub(poly2->vertices[j], v1));
			if (sij < si)
			{
				si = sij;
			}
		}
		if (si > maxSeparation)
		{
			maxSeparation = si;
			bestIndex = i;
		}
	}
	*edgeIndex = bestIndex;
	return maxSeparation;
}

static int minDotEdge(V2 searchDirection, const Poly* poly)
{
	int edge = 0;
	float minDot = FLT_MAX;
	for (int i = 0; i < poly->count; ++i)
	{
		float d = dot(searchDirection, poly->normals[i]);
		if (d < minDot)
		{
			minDot = d;
			edge = i;
		}
	}
	return edge;
}

static Manifold polygonSAT(const Poly* polyA, const Poly* polyB) // :441-506
{
	int edgeA = 0, edgeB = 0;
	float separationA = findMaxSeparation(&edgeA, polyA, polyB);
	float separationB = findMaxSeparation(&edgeB, polyB, polyA);
	int flip;
	if (separationB > separationA)
	{
		flip = 1;
		edgeA = minDotEdge(polyB->normals[edgeB], polyA);
	}
	else
	{
		flip = 0;
		edgeB = minDotEdge(polyA->normals[edgeA], polyB);
	}
	return clipPolygons(polyA, polyB, edgeA, edgeB, flip);
}

static Manifold collidePolygons(const Poly* polyA, Xf xfA, const Poly* polyB, Xf xfB, Cache* cache) // :509-650
{
	Manifold m = emptyManifold();
	float radius = polyA->radius + polyB->radius;
	Xf xf = invMulTransforms(xfA, xfB);
	Poly localPolyB;
	localPolyB.count = polyB->count;
	localPolyB.radius = polyB->radius;
	for (int i = 0; i < localPolyB.count; ++i)
	{
		localPolyB.vertices[i] = transformPoint(xf, polyB->vertices[i]);
		localPolyB.normals[i] = rotate(xf.q, polyB->normals[i]);
	}
	DistanceOutput output = shapeDistance(cache, polyA->vertices, polyA->count, localPolyB.vertices, localPolyB.count);
	if (output.distance > radius + SPECULATIVE_DISTANCE)
	{
		return m;
	}
	if (output.distance < 0.1f * LINEAR_SLOP)
	{
		m = polygonSAT(polyA, &localPolyB);
	}
	else if (cache->count == 1)
	{
		V2 pA = output.pointA, pB = output.pointB;
		float dist = output.distance;
		V2 normal = normalize(sub(pB, pA));
		V2 contactPointA = mulAdd(pB, 0.5f * (polyA->radius - localPolyB.radius - dist), normal);
		m.normal = rotate(xfA.q, normal);
		m.points[0].localAnchorA = contactPointA;
		m.points[0].localAnchorB = invTransformPoint(xf, contactPointA);
		m.points[0].separation = dist - radius;
		m.points[0].id = MAKE_ID(cache->indexA[0], cache->indexB[0]);
		m.pointCount = 1;
		return m;
	}
	else
	{
		int flip, edgeA, edgeB;
		int countA = polyA->count, countB = localPolyB.count;
		int a1 = cache->indexA[0], a2 = cache->indexA[1];
		int b1 = cache->indexB[0], b2 = cache->indexB[1];
		if (a1 == a2)
		{
			V2 axis = sub(output.pointA, output.pointB);
			float dot1 = dot(axis, localPolyB.normals[b1]);
			float dot2 = dot(axis, localPolyB.normals[b2]);
			edgeB = dot1 > dot2 ? b1 : b2;
			flip = 1;
			axis = localPolyB.normals[edgeB];
			int edgeA1 = a1;
			int edgeA2 = edgeA1 == 0 ? countA - 1 : edgeA1 - 1;
			dot1 = dot(axis, polyA->normals[edgeA1]);
			dot2 = dot(axis, polyA->normals[edgeA2]);
			edgeA = dot1 < dot2 ? edgeA1 : edgeA2;
		}
		else
		{
			V2 axis = sub(output.pointB, output.pointA);
			float dot1 = dot(axis, polyA->normals[a1]);
			float dot2 = dot(axis, polyA->normals[a2]);
			edgeA = dot1 > dot2 ? a1 : a2;
			flip = 0;
			axis = polyA->normals[edgeA];
			int edgeB1 = b1;
			int edgeB2 = edgeB1 == 0 ? countB - 1 : edgeB1 - 1;
			dot1 = dot(axis, localPolyB.normals[edgeB1]);
			dot2 = dot(axis, localPolyB.normals[edgeB2]);
			edgeB = dot1 < dot2 ? edgeB1 : edgeB2;
		}
		m = clipPolygons(polyA, &localPolyB, edgeA, edgeB, flip);
	}
	if (m.pointCount > 0)
	{
		m.normal = rotate(xfA.q, m.normal);
		for (int i = 0; i < m.pointCount; ++i)
		{
			m.points[i].localAnchorB = invTransformPoint(xf, m.points[i].localAnchorA);
		}
	}
	return m;
}

static Poly makeCapsule(V2 p1, V2 p2, float radius) // src/geometry.c:100-115
{
	Poly shape;
	memset(&shape, 0, sizeof(shape));
	shape.vertices[0] = p1;
	shape.vertices[1] = p2;
	V2 axis = normalizeChecked(sub(p2, p1));
	V2 normal = rightPerp(axis);
	shape.normals[0] = normal;
	shape.normals[1] = neg(normal);
	shape.count = 2;
	shape.radius = radius;
	return shape;
}

static Poly polyOf(const s2amdShape* sh)
{
	Poly p;
	memset(&p, 0, sizeof(p));
	if (sh->type == S2AMD_SHAPE_POLYGON)
	{
		p.count = sh->count;
		p.radius = sh->radius;
		for (int i = 0; i < sh->count; ++i)
		{
			p.vertices[i] = v2(sh->vertices[i][0], sh->vertices[i][1]);
			p.normals[i] = v2(sh->normals[i][0], sh->normals[i][1]);
		}
		return p;
	}
	// capsule, or a segment as a zero-radius capsule (src/manifold.c:224-246, :658-663)
	return makeCapsule(v2(sh->vertices[0][0], sh->vertices[0][1]), v2(sh->vertices[1][0], sh->vertices[1][1]),
					   sh->type == S2AMD_SHAPE_CAPSULE ? sh->radius : 0.0f);
}

// The manifold function of an ORDERED shape-type pair (src/contact.c:139-151: only "primary" orders reach here)
static int collide(const s2amdShape* shapeA, Xf xfA, const s2amdShape* shapeB, Xf xfB, Cache* cache, Manifold* out)
{
	int ta = shapeA->type, tb = shapeB->type;
	V2 a0 = v2(shapeA->vertices[0][0], shapeA->vertices[0][1]), a1 = v2(shapeA->vertices[1][0], shapeA->vertices[1][1]);
	V2 b0 = v2(shapeB->vertices[0][0], shapeB->vertices[0][1]);
	if (tb == S2AMD_SHAPE_CIRCLE)
	{
		switch (ta)
		{
			case S2AMD_SHAPE_CIRCLE:
				*out = collideCircles(a0, shapeA->radius, xfA, b0, shapeB->radius, xfB);
				return 1;
			case S2AMD_SHAPE_CAPSULE:
				*out = collideCapsuleAndCircle(a0, a1, shapeA->radius, xfA, b0, shapeB->radius, xfB);
				return 1;
			case S2AMD_SHAPE_SEGMENT:
				*out = collideCapsuleAndCircle(a0, a1, 0.0f, xfA, b0, shapeB->radius, xfB);
				return 1;
			case S2AMD_SHAPE_POLYGON:
			{
				Poly pa = polyOf(shapeA);
				*out = collidePolygonAndCircle(&pa, xfA, b0, shapeB->radius, xfB);
				return 1;
			}
			default:
				return 0;
		}
	}
	// capsule-capsule, polygon-capsule, polygon-polygon, segment-capsule, segment-polygon: the polygon path
	int ok = (ta == S2AMD_SHAPE_CAPSULE && tb == S2AMD_SHAPE_CAPSULE) || (ta == S2AMD_SHAPE_POLYGON && tb == S2AMD_SHAPE_CAPSULE) ||
			 (ta == S2AMD_SHAPE_POLYGON && tb == S2AMD_SHAPE_POLYGON) || (ta == S2AMD_SHAPE_SEGMENT && tb == S2AMD_SHAPE_CAPSULE) ||
			 (ta == S2AMD_SHAPE_SEGMENT && tb == S2AMD_SHAPE_POLYGON);
	if (!ok)
	{
		return 0;
	}
	Poly pa = polyOf(shapeA), pb = polyOf(shapeB);
	*out = collidePolygons(&pa, xfA, &pb, xfB, cache);
	return 1;
}

static int fatOverlap(const s2amdShape* a, const s2amdShape* b) // aabb.h:111-123
{
	float d1x = b->fatAABB[0] - a->fatAABB[2], d1y = b->fatAABB[1] - a->fatAABB[3];
	float d2x = a->fatAABB[0] - b->fatAABB[2], d2y = a->fatAABB[1] - b->fatAABB[3];
	if (d1x > 0.0f || d1y > 0.0f)
	{
		return 0;
	}
	if (d2x > 0.0f || d2y > 0.0f)
	{
		return 0;
	}
	return 1;
}

ORACLE_API int s2oracle_update_contacts(const s2amdBody* bodies, int32_t bodyCapacity, const float* origins, const s2amdShape* shapes,
										 int32_t shapeCapacity, s2amdPairState* pairs, s2amdContact* contacts, int32_t contactCapacity, int32_t* status)
{
	(void)bodyCapacity, (void)shapeCapacity;
	for (int i = 0; i < contactCapacity; ++i)
	{
		s2amdPairState* ps = pairs + i;
		s2amdContact* ct = contacts + i;
		if (ps->shapeA < 0 || ps->shapeB < 0)
		{
			status[i] = S2AMD_PAIR_FREE;
			continue;
		}
		const s2amdShape* shapeA = shapes + ps->shapeA;
		const s2amdShape* shapeB = shapes + ps->shapeB;
		if (!fatOverlap(shapeA, shapeB))
		{
			status[i] = S2AMD_PAIR_SEPARATED;
			continue;
		}
		status[i] = S2AMD_PAIR_UPDATED;
		const s2amdBody* bodyA = bodies + shapeA->body;
		const s2amdBody* bodyB = bodies + shapeB->body;
		Xf xfA = {{origins[2 * shapeA->body], origins[2 * shapeA->body + 1]}, {bodyA->rot[0], bodyA->rot[1]}};
		Xf xfB = {{origins[2 * shapeB->body], origins[2 * shapeB->body + 1]}, {bodyB->rot[0], bodyB->rot[1]}};

		Cache cache;
		cache.metric = ps->cacheMetric;
		cache.count = ps->cacheCount;
		for (int k = 0; k < 3; ++k)
		{
			cache.indexA[k] = ps->cacheIndexA[k];
			cache.indexB[k] = ps->cacheIndexB[k];
		}
		Manifold m;
		if (!collide(shapeA, xfA, shapeB, xfB, &cache, &m))
		{
			m = emptyManifold();
		}

		// s2UpdateContact: src/contact.c:296-358
		s2amdContact old = *ct;
		uint16_t oldId[2] = {ps->id[0], ps->id[1]};
		int frictionPersisted = 1;
		if (m.pointCount != old.pointCount)
		{
			frictionPersisted = 0;
		}
		ct->pointCount = m.pointCount;
		ct->normal[0] = m.normal.x, ct->normal[1] = m.normal.y;
		for (int p = 0; p < 2; ++p)
		{
			s2amdManifoldPoint* q = ct->points + p;
			memset(q, 0, sizeof(*q));
			ps->id[p] = 0;
			ps->persisted[p] = 0;
			if (p >= m.pointCount)
			{
				continue; // a fresh manifold is zero-filled beyond pointCount (manifold = {0})
			}
			const MPoint* mp = m.points + p;
			q->localAnchorA[0] = mp->localAnchorA.x, q->localAnchorA[1] = mp->localAnchorA.y;
			q->localAnchorB[0] = mp->localAnchorB.x, q->localAnchorB[1] = mp->localAnchorB.y;
			q->separation = mp->separation;
			ps->id[p] = mp->id;
			for (int j = 0; j < old.pointCount; ++j)
			{
				if (oldId[j] == mp->id)
				{
					const s2amdManifoldPoint* o = old.points + j;
					memcpy(q->frictionNormalA, o->frictionNormalA, sizeof(q->frictionNormalA));
					memcpy(q->frictionNormalB, o->frictionNormalB, sizeof(q->frictionNormalB));
					memcpy(q->frictionAnchorA, o->frictionAnchorA, sizeof(q->frictionAnchorA));
					memcpy(q->frictionAnchorB, o->frictionAnchorB, sizeof(q->frictionAnchorB));
					q->normalImpulse = o->normalImpulse;
					q->tangentImpulse = o->tangentImpulse;
					ps->persisted[p] = 1;
					break;
				}
			}
			if (!ps->persisted[p])
			{
				frictionPersisted = 0;
			}
		}
		ct->frictionPersisted = frictionPersisted;
		ps->cacheMetric = cache.metric;
		ps->cacheCount = (uint16_t)cache.count;
		for (int k = 0; k < 3; ++k)
		{
			ps->cacheIndexA[k] = (uint8_t)cache.indexA[k];
			ps->cacheIndexB[k] = (uint8_t)cache.indexB[k];
		}
	}
	return 0;
}
