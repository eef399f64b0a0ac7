// Narrow phase, behind the same C-ABI:
//   s2amd_update_contacts == Stage 3 of s2World_Step, "update contacts" (src/world.c:132-168): per live contact
//   whose fat AABBs still overlap, s2UpdateContact (src/contact.c:296-358) -- the manifold function of the
//   shape-type pair, then the feature-id matching that carries impulses and the sticky-friction cache over.
//
// One thread per contact slot; the reference walks the pool on one core.  The two polygons of a pair (B
// already moved into A's frame, as s2CollidePolygons does) live in LDS because GJK addresses their vertices by
// computed index; everything else is registers.  All arithmetic is fp32 in the reference's operation order
// (-ffp-contract=off), so manifolds, feature ids and simplex caches are bit-identical to the reference's
// (tests/test_gpu_narrowphase.py against captures of the unmodified reference and the oracle).
//
// Reference map: src/manifold.c -- s2CollideCircles :16-49, s2CollideCapsuleAndCircle :51-110,
// s2CollidePolygonAndCircle :113-222, s2ClipPolygons :248-399, s2FindMaxSeparation :402-438, s2PolygonSAT
// :441-506, s2CollidePolygons :509-650; capsules and segments enter the polygon path through s2MakeCapsule
// (src/geometry.c:100-115) exactly as :224-246 and :652-663 do; GJK with its simplex cache: src/distance.c:120-604.

#include "launch.h"
#include "s2_device.h"

#include "solver2d_amd.h"

#include <cfloat>
#include <cstring>
#include <string>

#define S2_NP_BLOCK 128
#define S2_NP_SPECULATIVE (4.0f * S2_LINEAR_SLOP) // constants.h:8
#define S2_NP_MAX_VERTS 8

int s2amdFail(int code, const std::string& msg);
hipStream_t s2amdStream(s2amdSolver* s);
int s2amdDevice(s2amdSolver* s);
void s2amdRecordDeviceMs(s2amdSolver* s, float ms);

namespace
{

struct Xf
{
	V2 p;
	Rot q;
};

S2_DEV V2 xfPoint(Xf xf, V2 p) // math.h:350-356
{
	float x = (xf.q.c * p.x - xf.q.s * p.y) + xf.p.x;
	float y = (xf.q.s * p.x + xf.q.c * p.y) + xf.p.y;
	return v2(x, y);
}
S2_DEV V2 xfInvPoint(Xf xf, V2 p) // math.h:359-364
{
	float vx = p.x - xf.p.x, vy = p.y - xf.p.y;
	return v2(xf.q.c * vx + xf.q.s * vy, -xf.q.s * vx + xf.q.c * vy);
}
S2_DEV Xf xfInvMul(Xf A, Xf B) // math.h:378-384, s2InvMulRot :307-317
{
	Xf C;
	C.q.s = A.q.c * B.q.s - A.q.s * B.q.c;
	C.q.c = A.q.c * B.q.c + A.q.s * B.q.s;
	C.p = invRotate(A.q, sub(B.p, A.p));
	return C;
}
S2_DEV V2 lerp2(V2 a, V2 b, float t) { return v2(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y)); } // math.h:103-106
S2_DEV float distance2(V2 a, V2 b)																	 // math.h:180-185
{
	float dx = b.x - a.x, dy = b.y - a.y;
	return sqrtf(dx * dx + dy * dy);
}
S2_DEV V2 normalizeChecked(V2 v) // src/math.c:54-66
{
	float len = length(v);
	if (len < FLT_EPSILON)
	{
		return v2(0.0f, 0.0f);
	}
	float inv = 1.0f / len;
	return v2(inv * v.x, inv * v.y);
}
S2_DEV V2 lengthAndNormalize(float* len, V2 v) // src/math.c:68-80
{
	*len = length(v);
	if (*len < FLT_EPSILON)
	{
		return v2(0.0f, 0.0f);
	}
	float inv = 1.0f / *len;
	return v2(inv * v.x, inv * v.y);
}

// A polygon in LDS: vertices and normals interleaved per workgroup thread (stride = blockDim) so that lanes
// reading "their" vertex i hit distinct banks
struct PolyRef
{
	float2* verts; // [S2_NP_MAX_VERTS] at stride S2_NP_BLOCK
	float2* norms;
	float radius;
	int count;
	S2_DEV V2 v(int i) const
	{
		float2 t = verts[i * S2_NP_BLOCK];
		return v2(t.x, t.y);
	}
	S2_DEV V2 n(int i) const
	{
		float2 t = norms[i * S2_NP_BLOCK];
		return v2(t.x, t.y);
	}
	S2_DEV void set(int i, V2 vert, V2 norm) const
	{
		verts[i * S2_NP_BLOCK] = make_float2(vert.x, vert.y);
		norms[i * S2_NP_BLOCK] = make_float2(norm.x, norm.y);
	}
};

struct MPoint
{
	V2 localAnchorA, localAnchorB;
	float separation;
	uint32_t id;
};
struct Manifold
{
	MPoint points[2];
	V2 normal;
	int pointCount;
};
S2_DEV Manifold emptyManifold()
{
	Manifold m;
	m.points[0].localAnchorA = m.points[0].localAnchorB = v2(0.0f, 0.0f);
	m.points[1].localAnchorA = m.points[1].localAnchorB = v2(0.0f, 0.0f);
	m.points[0].separation = m.points[1].separation = 0.0f;
	m.points[0].id = m.points[1].id = 0u;
	m.normal = v2(0.0f, 0.0f);
	m.pointCount = 0;
	return m;
}
S2_DEV uint32_t makeId(int a, int b) { return (((uint32_t)a & 0xffu) << 8) | ((uint32_t)b & 0xffu); } // manifold.h:17

struct Cache
{
	float metric;
	int count;
	int indexA[3], indexB[3];
};

// ---- GJK (src/distance.c) ----
struct SVertex
{
	V2 wA, wB, w;
	float a;
	int indexA, indexB;
};

S2_DEV int findSupport(const PolyRef& p, V2 d) // :120-135
{
	int best = 0;
	float bestValue = dot(p.v(0), d);
	for (int i = 1; i < p.count; ++i)
	{
		float value = dot(p.v(i), d);
		if (value > bestValue)
		{
			best = i;
			bestValue = value;
		}
	}
	return best;
}

S2_DEV V2 weight2(float a1, V2 w1, float a2, V2 w2) { return v2(a1 * w1.x + a2 * w2.x, a1 * w1.y + a2 * w2.y); }
S2_DEV V2 weight3(float a1, V2 w1, float a2, V2 w2, float a3, V2 w3)
{
	return v2(a1 * w1.x + a2 * w2.x + a3 * w3.x, a1 * w1.y + a2 * w2.y + a3 * w3.y);
}

struct DistanceOutput
{
	V2 pointA, pointB;
	float distance;
};

// s2ShapeDistance in the only form the manifold code uses: identity transforms, no radii (src/manifold.c:523-530)
S2_DEV DistanceOutput shapeDistance(Cache& cache, const PolyRef& A, const PolyRef& B)
{
	const Xf identity = {v2(0.0f, 0.0f), {0.0f, 1.0f}};
	SVertex s0, s1, s2;
	s0.a = s1.a = s2.a = 0.0f;
	s0.indexA = s0.indexB = s1.indexA = s1.indexB = s2.indexA = s2.indexB = 0;
	s0.wA = s0.wB = s0.w = s1.wA = s1.wB = s1.w = s2.wA = s2.wB = s2.w = v2(0.0f, 0.0f);
	int count = cache.count;
	// s2MakeSimplexFromCache :172-214
	auto fromCache = [&](SVertex& v, int i) {
		v.indexA = cache.indexA[i];
		v.indexB = cache.indexB[i];
		v.wA = xfPoint(identity, A.v(v.indexA));
		v.wB = xfPoint(identity, B.v(v.indexB));
		v.w = sub(v.wB, v.wA);
		v.a = -1.0f;
	};
	if (count > 0)
	{
		fromCache(s0, 0);
	}
	if (count > 1)
	{
		fromCache(s1, 1);
	}
	if (count > 2)
	{
		fromCache(s2, 2);
	}
	if (count == 0)
	{
		s0.indexA = 0;
		s0.indexB = 0;
		s0.wA = xfPoint(identity, A.v(0));
		s0.wB = xfPoint(identity, B.v(0));
		s0.w = sub(s0.wB, s0.wA);
		s0.a = 1.0f;
		count = 1;
	}

	int saveA[3] = {0, 0, 0}, saveB[3] = {0, 0, 0};
	int iter = 0;
	while (iter < 20)
	{
		int saveCount = count;
		saveA[0] = s0.indexA, saveB[0] = s0.indexB;
		saveA[1] = s1.indexA, saveB[1] = s1.indexB;
		saveA[2] = s2.indexA, saveB[2] = s2.indexB;

		if (count == 2)
		{
			// s2SolveSimplex2 :337-367
			V2 w1 = s0.w, w2 = s1.w;
			V2 e12 = sub(w2, w1);
			float d12_2 = -dot(w1, e12);
			if (d12_2 <= 0.0f)
			{
				s0.a = 1.0f;
				count = 1;
			}
			else
			{
				float d12_1 = dot(w2, e12);
				if (d12_1 <= 0.0f)
				{
					s1.a = 1.0f;
					count = 1;
					s0 = s1;
				}
				else
				{
					float inv_d12 = 1.0f / (d12_1 + d12_2);
					s0.a = d12_1 * inv_d12;
					s1.a = d12_2 * inv_d12;
					count = 2;
				}
			}
		}
		else if (count == 3)
		{
			// s2SolveSimplex3 :369-473
			V2 w1 = s0.w, w2 = s1.w, w3 = s2.w;
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
				s0.a = 1.0f;
				count = 1;
			}
			else if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f)
			{
				float inv = 1.0f / (d12_1 + d12_2);
				s0.a = d12_1 * inv;
				s1.a = d12_2 * inv;
				count = 2;
			}
			else if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f)
			{
				float inv = 1.0f / (d13_1 + d13_2);
				s0.a = d13_1 * inv;
				s2.a = d13_2 * inv;
				count = 2;
				s1 = s2;
			}
			else if (d12_1 <= 0.0f && d23_2 <= 0.0f)
			{
				s1.a = 1.0f;
				count = 1;
				s0 = s1;
			}
			else if (d13_1 <= 0.0f && d23_1 <= 0.0f)
			{
				s2.a = 1.0f;
				count = 1;
				s0 = s2;
			}
			else if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f)
			{
				float inv = 1.0f / (d23_1 + d23_2);
				s1.a = d23_1 * inv;
				s2.a = d23_2 * inv;
				count = 2;
				s0 = s2;
			}
			else
			{
				float inv = 1.0f / (d123_1 + d123_2 + d123_3);
				s0.a = d123_1 * inv;
				s1.a = d123_2 * inv;
				s2.a = d123_3 * inv;
				count = 3;
			}
		}
		if (count == 3)
		{
			break;
		}
		// s2ComputeSimplexSearchDirection :228-254
		V2 d;
		if (count == 1)
		{
			d = neg(s0.w);
		}
		else
		{
			V2 e12 = sub(s1.w, s0.w);
			float sgn = cross(e12, neg(s0.w));
			d = sgn > 0.0f ? crossSV(1.0f, e12) : crossVS(e12, 1.0f);
		}
		if (dot(d, d) < FLT_EPSILON * FLT_EPSILON)
		{
			break;
		}
		SVertex nv;
		nv.a = count == 1 ? s1.a : s2.a; // the slot's barycentric coordinate is left as it was (:568-573 touch w*, index* only)
		nv.indexA = findSupport(A, invRotate(identity.q, neg(d)));
		nv.wA = xfPoint(identity, A.v(nv.indexA));
		nv.indexB = findSupport(B, invRotate(identity.q, d));
		nv.wB = xfPoint(identity, B.v(nv.indexB));
		nv.w = sub(nv.wB, nv.wA);
		if (count == 1)
		{
			s1 = nv;
		}
		else
		{
			s2 = nv;
		}
		++iter;
		bool duplicate = false;
		for (int i = 0; i < saveCount; ++i)
		{
			if (nv.indexA == saveA[i] && nv.indexB == saveB[i])
			{
				duplicate = true;
				break;
			}
		}
		if (duplicate)
		{
			break;
		}
		++count;
	}

	DistanceOutput out;
	if (count == 1)
	{
		out.pointA = s0.wA;
		out.pointB = s0.wB;
	}
	else if (count == 2)
	{
		out.pointA = weight2(s0.a, s0.wA, s1.a, s1.wA);
		out.pointB = weight2(s0.a, s0.wB, s1.a, s1.wB);
	}
	else
	{
		out.pointA = weight3(s0.a, s0.wA, s1.a, s1.wA, s2.a, s2.wA);
		out.pointB = out.pointA;
	}
	out.distance = distance2(out.pointA, out.pointB);

	// s2MakeSimplexCache :216-226 with s2Simplex_Metric :149-170 (slots beyond count keep their old indices)
	if (count == 1)
	{
		cache.metric = 0.0f;
	}
	else if (count == 2)
	{
		cache.metric = distance2(s0.w, s1.w);
	}
	else
	{
		cache.metric = cross(sub(s1.w, s0.w), sub(s2.w, s0.w));
	}
	cache.count = count;
	cache.indexA[0] = s0.indexA & 0xff, cache.indexB[0] = s0.indexB & 0xff;
	if (count > 1)
	{
		cache.indexA[1] = s1.indexA & 0xff, cache.indexB[1] = s1.indexB & 0xff;
	}
	if (count > 2)
	{
		cache.indexA[2] = s2.indexA & 0xff, cache.indexB[2] = s2.indexB & 0xff;
	}
	return out;
}

// ---- manifold functions (src/manifold.c) ----
S2_DEV Manifold collideCircles(V2 pointA, float radiusA, Xf xfA, V2 pointB0, float radiusB, Xf xfB)
{
	Manifold m = emptyManifold();
	Xf xf = xfInvMul(xfA, xfB);
	V2 pointB = xfPoint(xf, pointB0);
	float dist;
	V2 normal = lengthAndNormalize(&dist, sub(pointB, pointA));
	float separation = dist - radiusA - radiusB;
	if (separation > S2_NP_SPECULATIVE)
	{
		return m;
	}
	V2 cA = mulAdd(pointA, radiusA, normal);
	V2 cB = mulAdd(pointB, -radiusB, normal);
	V2 contactPointA = lerp2(cA, cB, 0.5f);
	m.normal = rotate(xfA.q, normal);
	m.points[0].localAnchorA = contactPointA;
	m.points[0].localAnchorB = xfInvPoint(xf, contactPointA);
	m.points[0].separation = separation;
	m.pointCount = 1;
	return m;
}

S2_DEV Manifold collideCapsuleAndCircle(V2 p1, V2 p2, float radiusA, Xf xfA, V2 pointB0, float radiusB, Xf xfB)
{
	Manifold m = emptyManifold();
	Xf xf = xfInvMul(xfA, xfB);
	V2 pB = xfPoint(xf, pointB0);
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
	if (separation > S2_NP_SPECULATIVE)
	{
		return m;
	}
	V2 cA = mulAdd(pA, radiusA, normal);
	V2 cB = mulAdd(pB, -radiusB, normal);
	V2 contactPointA = lerp2(cA, cB, 0.5f);
	m.normal = rotate(xfA.q, normal);
	m.points[0].localAnchorA = contactPointA;
	m.points[0].localAnchorB = xfInvPoint(xf, contactPointA);
	m.points[0].separation = separation;
	m.pointCount = 1;
	return m;
}

S2_DEV Manifold collidePolygonAndCircle(const PolyRef& polygonA, Xf xfA, V2 pointB0, float radiusB, Xf xfB)
{
	Manifold m = emptyManifold();
	Xf xf = xfInvMul(xfA, xfB);
	V2 c = xfPoint(xf, pointB0);
	float radiusA = polygonA.radius;
	float radius = radiusA + radiusB;
	int normalIndex = 0;
	float separation = -FLT_MAX;
	int vertexCount = polygonA.count;
	for (int i = 0; i < vertexCount; ++i)
	{
		float s = dot(polygonA.n(i), sub(c, polygonA.v(i)));
		if (s > separation)
		{
			separation = s;
			normalIndex = i;
		}
	}
	if (separation > radius + S2_NP_SPECULATIVE)
	{
		return m;
	}
	int vertIndex1 = normalIndex;
	int vertIndex2 = vertIndex1 + 1 < vertexCount ? vertIndex1 + 1 : 0;
	V2 v1 = polygonA.v(vertIndex1), v2_ = polygonA.v(vertIndex2);
	float u1 = dot(sub(c, v1), sub(v2_, v1));
	float u2 = dot(sub(c, v2_), sub(v1, v2_));
	const bool near1 = u1 < 0.0f && separation > FLT_EPSILON;
	const bool near2 = u2 < 0.0f && separation > FLT_EPSILON;
	if (near1 || near2)
	{
		V2 v = near1 ? v1 : v2_;
		V2 normal = normalize(sub(c, v));
		separation = dot(sub(c, v), normal);
		if (separation > radius + S2_NP_SPECULATIVE)
		{
			return m;
		}
		V2 cA = mulAdd(v, radiusA, normal);
		V2 cB = mulSub(c, radiusB, normal);
		V2 contactPointA = lerp2(cA, cB, 0.5f);
		m.normal = rotate(xfA.q, normal);
		m.points[0].localAnchorA = contactPointA;
		m.points[0].localAnchorB = xfInvPoint(xf, contactPointA);
		m.points[0].separation = dot(sub(cB, cA), normal);
		m.pointCount = 1;
	}
	else
	{
		V2 normal = polygonA.n(normalIndex);
		m.normal = rotate(xfA.q, normal);
		V2 cA = mulAdd(c, radiusA - dot(sub(c, v1), normal), normal);
		V2 cB = mulSub(c, radiusB, normal);
		V2 contactPointA = lerp2(cA, cB, 0.5f);
		m.points[0].localAnchorA = contactPointA;
		m.points[0].localAnchorB = xfInvPoint(xf, contactPointA);
		m.points[0].separation = separation - radius;
		m.pointCount = 1;
	}
	return m;
}

S2_DEV Manifold clipPolygons(const PolyRef& polyA, const PolyRef& polyB, int edgeA, int edgeB, bool flip)
{
	Manifold m = emptyManifold();
	const PolyRef& poly1 = flip ? polyB : polyA;
	const PolyRef& poly2 = flip ? polyA : polyB;
	int i11 = flip ? edgeB : edgeA;
	int i21 = flip ? edgeA : edgeB;
	int i12 = i11 + 1 < poly1.count ? i11 + 1 : 0;
	int i22 = i21 + 1 < poly2.count ? i21 + 1 : 0;
	V2 normal = poly1.n(i11);
	V2 v11 = poly1.v(i11), v12 = poly1.v(i12);
	V2 v21 = poly2.v(i21), v22 = poly2.v(i22);
	V2 tangent = crossSV(1.0f, normal);
	float lower1 = 0.0f;
	float upper1 = dot(sub(v12, v11), tangent);
	float upper2 = dot(sub(v21, v11), tangent);
	float lower2 = dot(sub(v22, v11), tangent);
	V2 vLower = (lower2 < lower1 && upper2 - lower2 > FLT_EPSILON) ? lerp2(v22, v21, (lower1 - lower2) / (upper2 - lower2)) : v22;
	V2 vUpper = (upper2 > upper1 && upper2 - lower2 > FLT_EPSILON) ? lerp2(v22, v21, (upper1 - lower2) / (upper2 - lower2)) : v21;
	float separationLower = dot(sub(vLower, v11), normal);
	float separationUpper = dot(sub(vUpper, v11), normal);
	float r1 = poly1.radius, r2 = poly2.radius;
	vLower = mulAdd(vLower, 0.5f * (r1 - r2 - separationLower), normal);
	vUpper = mulAdd(vUpper, 0.5f * (r1 - r2 - separationUpper), normal);
	float radius = r1 + r2;
	if (!flip)
	{
		m.normal = normal;
		m.points[0].localAnchorA = vLower;
		m.points[0].separation = separationLower - radius;
		m.points[0].id = makeId(i11, i22);
		m.points[1].localAnchorA = vUpper;
		m.points[1].separation = separationUpper - radius;
		m.points[1].id = makeId(i12, i21);
	}
	else
	{
		m.normal = neg(normal);
		m.points[0].localAnchorA = vUpper;
		m.points[0].separation = separationUpper - radius;
		m.points[0].id = makeId(i21, i12);
		m.points[1].localAnchorA = vLower;
		m.points[1].separation = separationLower - radius;
		m.points[1].id = makeId(i22, i11);
	}
	m.pointCount = 2;
	return m;
}

S2_DEV float findMaxSeparation(int* edgeIndex, const PolyRef& poly1, const PolyRef& poly2)
{
	int bestIndex = 0;
	float maxSeparation = -FLT_MAX;
	for (int i = 0; i < poly1.count; ++i)
	{
		V2 n = poly1.n(i), v1 = poly1.v(i);
		float si = FLT_MAX;
		for (int j = 0; j < poly2.count; ++j)
		{
			float sij = dot(n, sub(poly2.v(j), v1));
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

S2_DEV int minDotEdge(V2 searchDirection, const PolyRef& poly)
{
	int edge = 0;
	float minDot = FLT_MAX;
	for (int i = 0; i < poly.count; ++i)
	{
		float d = dot(searchDirection, poly.n(i));
		if (d < minDot)
		{
			minDot = d;
			edge = i;
		}
	}
	return edge;
}

// polyB must already be in polyA's frame (the caller built it in LDS)
S2_DEV Manifold collidePolygons(const PolyRef& polyA, Xf xfA, const PolyRef& localPolyB, Xf xf, Cache& cache)
{
	Manifold m = emptyManifold();
	float radius = polyA.radius + localPolyB.radius;
	DistanceOutput output = shapeDistance(cache, polyA, localPolyB);
	if (output.distance > radius + S2_NP_SPECULATIVE)
	{
		return m;
	}
	if (output.distance < 0.1f * S2_LINEAR_SLOP)
	{
		// s2PolygonSAT :441-506
		int edgeA = 0, edgeB = 0;
		float separationA = findMaxSeparation(&edgeA, polyA, localPolyB);
		float separationB = findMaxSeparation(&edgeB, localPolyB, polyA);
		bool flip;
		if (separationB > separationA)
		{
			flip = true;
			edgeA = minDotEdge(localPolyB.n(edgeB), polyA);
		}
		else
		{
			flip = false;
			edgeB = minDotEdge(polyA.n(edgeA), localPolyB);
		}
		m = clipPolygons(polyA, localPolyB, edgeA, edgeB, flip);
	}
	else if (cache.count == 1)
	{
		V2 pA = output.pointA, pB = output.pointB;
		float dist = output.distance;
		V2 normal = normalize(sub(pB, pA));
		V2 contactPointA = mulAdd(pB, 0.5f * (polyA.radius - localPolyB.radius - dist), normal);
		m.normal = rotate(xfA.q, normal);
		m.points[0].localAnchorA = contactPointA;
		m.points[0].localAnchorB = xfInvPoint(xf, contactPointA);
		m.points[0].separation = dist - radius;
		m.points[0].id = makeId(cache.indexA[0], cache.indexB[0]);
		m.pointCount = 1;
		return m;
	}
	else
	{
		bool flip;
		int edgeA, edgeB;
		int countA = polyA.count, countB = localPolyB.count;
		int a1 = cache.indexA[0], a2 = cache.indexA[1];
		int b1 = cache.indexB[0], b2 = cache.indexB[1];
		if (a1 == a2)
		{
			V2 axis = sub(output.pointA, output.pointB);
			float dot1 = dot(axis, localPolyB.n(b1));
			float dot2 = dot(axis, localPolyB.n(b2));
			edgeB = dot1 > dot2 ? b1 : b2;
			flip = true;
			axis = localPolyB.n(edgeB);
			int edgeA1 = a1;
			int edgeA2 = edgeA1 == 0 ? countA - 1 : edgeA1 - 1;
			dot1 = dot(axis, polyA.n(edgeA1));
			dot2 = dot(axis, polyA.n(edgeA2));
			edgeA = dot1 < dot2 ? edgeA1 : edgeA2;
		}
		else
		{
			V2 axis = sub(output.pointB, output.pointA);
			float dot1 = dot(axis, polyA.n(a1));
			float dot2 = dot(axis, polyA.n(a2));
			edgeA = dot1 > dot2 ? a1 : a2;
			flip = false;
			axis = polyA.n(edgeA);
			int edgeB1 = b1;
			int edgeB2 = edgeB1 == 0 ? countB - 1 : edgeB1 - 1;
			dot1 = dot(axis, localPolyB.n(edgeB1));
			dot2 = dot(axis, localPolyB.n(edgeB2));
			edgeB = dot1 < dot2 ? edgeB1 : edgeB2;
		}
		m = clipPolygons(polyA, localPolyB, edgeA, edgeB, flip);
	}
	if (m.pointCount > 0)
	{
		m.normal = rotate(xfA.q, m.normal);
		for (int i = 0; i < m.pointCount; ++i)
		{
			m.points[i].localAnchorB = xfInvPoint(xf, m.points[i].localAnchorA);
		}
	}
	return m;
}

// shape -> polygon in LDS; B is transformed into A's frame on the way (src/manifold.c:517-526).  Capsules and
// segments become two-vertex polygons as s2MakeCapsule builds them (src/geometry.c:100-115).
S2_DEV void stagePolygon(const s2amdShape* sh, const PolyRef& out, bool toFrame, Xf xf, float* radius, int* count)
{
	if (sh->type == S2AMD_SHAPE_POLYGON)
	{
		*count = sh->count;
		*radius = sh->radius;
		for (int i = 0; i < sh->count; ++i)
		{
			V2 vert = v2(sh->vertices[i][0], sh->vertices[i][1]);
			V2 norm = v2(sh->normals[i][0], sh->normals[i][1]);
			if (toFrame)
			{
				vert = xfPoint(xf, vert);
				norm = rotate(xf.q, norm);
			}
			out.set(i, vert, norm);
		}
		return;
	}
	V2 p1 = v2(sh->vertices[0][0], sh->vertices[0][1]), p2 = v2(sh->vertices[1][0], sh->vertices[1][1]);
	V2 axis = normalizeChecked(sub(p2, p1));
	V2 normal = rightPerp(axis);
	V2 n0 = normal, n1 = neg(normal);
	if (toFrame)
	{
		p1 = xfPoint(xf, p1), p2 = xfPoint(xf, p2);
		n0 = rotate(xf.q, n0), n1 = rotate(xf.q, n1);
	}
	out.set(0, p1, n0);
	out.set(1, p2, n1);
	*count = 2;
	*radius = sh->type == S2AMD_SHAPE_CAPSULE ? sh->radius : 0.0f;
}

// one contact slot of stage 3; returns S2AMD_PAIR_*.  lds: the block's 4 * S2_NP_MAX_VERTS * S2_NP_BLOCK float2 staging area
S2_DEV int updateContactOne(const s2amdBody* bodies, const float2* origins, const s2amdShape* shapes, s2amdPairState* ps, s2amdContact* ct, float2* lds)
{
	if (ps->shapeA < 0 || ps->shapeB < 0)
	{
		return S2AMD_PAIR_FREE;
	}
	const s2amdShape* shapeA = shapes + ps->shapeA;
	const s2amdShape* shapeB = shapes + ps->shapeB;
	{
		// s2AABB_Overlaps on the fat boxes: aabb.h:111-123, src/world.c:149-167
		float d1x = shapeB->fatAABB[0] - shapeA->fatAABB[2], d1y = shapeB->fatAABB[1] - shapeA->fatAABB[3];
		float d2x = shapeA->fatAABB[0] - shapeB->fatAABB[2], d2y = shapeA->fatAABB[1] - shapeB->fatAABB[3];
		if (d1x > 0.0f || d1y > 0.0f || d2x > 0.0f || d2y > 0.0f)
		{
			return S2AMD_PAIR_SEPARATED;
		}
	}
	const int bodyA = shapeA->body, bodyB = shapeB->body;
	Xf xfA, xfB;
	xfA.p = v2(origins[bodyA].x, origins[bodyA].y);
	xfA.q.s = bodies[bodyA].rot[0], xfA.q.c = bodies[bodyA].rot[1];
	xfB.p = v2(origins[bodyB].x, origins[bodyB].y);
	xfB.q.s = bodies[bodyB].rot[0], xfB.q.c = bodies[bodyB].rot[1];

	Cache cache;
	cache.metric = ps->cacheMetric;
	cache.count = ps->cacheCount;
	for (int k = 0; k < 3; ++k)
	{
		cache.indexA[k] = ps->cacheIndexA[k];
		cache.indexB[k] = ps->cacheIndexB[k];
	}

	// the manifold function of the ordered type pair (src/contact.c:139-151)
	const int ta = shapeA->type, tb = shapeB->type;
	Manifold m = emptyManifold();
	PolyRef pa, pb;
	pa.verts = lds + threadIdx.x;
	pa.norms = lds + S2_NP_MAX_VERTS * S2_NP_BLOCK + threadIdx.x;
	pb.verts = lds + 2 * S2_NP_MAX_VERTS * S2_NP_BLOCK + threadIdx.x;
	pb.norms = lds + 3 * S2_NP_MAX_VERTS * S2_NP_BLOCK + threadIdx.x;
	const V2 a0 = v2(shapeA->vertices[0][0], shapeA->vertices[0][1]), a1 = v2(shapeA->vertices[1][0], shapeA->vertices[1][1]);
	const V2 b0 = v2(shapeB->vertices[0][0], shapeB->vertices[0][1]);
	if (tb == S2AMD_SHAPE_CIRCLE)
	{
		if (ta == S2AMD_SHAPE_CIRCLE)
		{
			m = collideCircles(a0, shapeA->radius, xfA, b0, shapeB->radius, xfB);
		}
		else if (ta == S2AMD_SHAPE_CAPSULE || ta == S2AMD_SHAPE_SEGMENT)
		{
			m = collideCapsuleAndCircle(a0, a1, ta == S2AMD_SHAPE_CAPSULE ? shapeA->radius : 0.0f, xfA, b0, shapeB->radius, xfB);
		}
		else if (ta == S2AMD_SHAPE_POLYGON)
		{
			stagePolygon(shapeA, pa, false, xfA, &pa.radius, &pa.count);
			m = collidePolygonAndCircle(pa, xfA, b0, shapeB->radius, xfB);
		}
	}
	else if ((ta == S2AMD_SHAPE_CAPSULE && tb == S2AMD_SHAPE_CAPSULE) || (ta == S2AMD_SHAPE_POLYGON && tb == S2AMD_SHAPE_CAPSULE) ||
			 (ta == S2AMD_SHAPE_POLYGON && tb == S2AMD_SHAPE_POLYGON) || (ta == S2AMD_SHAPE_SEGMENT && tb == S2AMD_SHAPE_CAPSULE) ||
			 (ta == S2AMD_SHAPE_SEGMENT && tb == S2AMD_SHAPE_POLYGON))
	{
		Xf xf = xfInvMul(xfA, xfB);
		stagePolygon(shapeA, pa, false, xf, &pa.radius, &pa.count);
		stagePolygon(shapeB, pb, true, xf, &pb.radius, &pb.count);
		m = collidePolygons(pa, xfA, pb, xf, cache);
	}

	// s2UpdateContact: src/contact.c:296-358
	const int oldCount = ct->pointCount;
	const uint32_t oldId0 = ps->id[0], oldId1 = ps->id[1];
	s2amdManifoldPoint oldPoints[2] = {ct->points[0], ct->points[1]};
	int frictionPersisted = m.pointCount == oldCount ? 1 : 0;
	ct->pointCount = m.pointCount;
	ct->normal[0] = m.normal.x, ct->normal[1] = m.normal.y;
	for (int p = 0; p < 2; ++p)
	{
		s2amdManifoldPoint q;
		memset(&q, 0, sizeof(q));
		uint32_t id = 0u;
		int persisted = 0;
		if (p < m.pointCount)
		{
			const MPoint& mp = m.points[p];
			q.localAnchorA[0] = mp.localAnchorA.x, q.localAnchorA[1] = mp.localAnchorA.y;
			q.localAnchorB[0] = mp.localAnchorB.x, q.localAnchorB[1] = mp.localAnchorB.y;
			q.separation = mp.separation;
			id = mp.id;
			for (int j = 0; j < oldCount && j < 2; ++j)
			{
				if ((j == 0 ? oldId0 : oldId1) == id)
				{
					const s2amdManifoldPoint& o = oldPoints[j];
					q.frictionNormalA[0] = o.frictionNormalA[0], q.frictionNormalA[1] = o.frictionNormalA[1];
					q.frictionNormalB[0] = o.frictionNormalB[0], q.frictionNormalB[1] = o.frictionNormalB[1];
					q.frictionAnchorA[0] = o.frictionAnchorA[0], q.frictionAnchorA[1] = o.frictionAnchorA[1];
					q.frictionAnchorB[0] = o.frictionAnchorB[0], q.frictionAnchorB[1] = o.frictionAnchorB[1];
					q.normalImpulse = o.normalImpulse;
					q.tangentImpulse = o.tangentImpulse;
					persisted = 1;
					break;
				}
			}
			if (!persisted)
			{
				frictionPersisted = 0;
			}
		}
		ct->points[p] = q;
		ps->id[p] = (uint16_t)id;
		ps->persisted[p] = (uint8_t)persisted;
	}
	ct->frictionPersisted = frictionPersisted;
	ps->cacheMetric = cache.metric;
	ps->cacheCount = (uint16_t)cache.count;
	for (int k = 0; k < 3; ++k)
	{
		ps->cacheIndexA[k] = (uint8_t)cache.indexA[k];
		ps->cacheIndexB[k] = (uint8_t)cache.indexB[k];
	}
	return S2AMD_PAIR_UPDATED;
}

// SUMMARY (resident world, world.hip): a separated pair is destroyed the way src/world.c:149-167 destroys its contact (no
// manifold, free pair slot), the slot's point count is compared with the previous step's byte, and the step's counters
// {separated, active, flips, moves} are accumulated in summary[0..3].
template <bool SUMMARY>
__global__ __launch_bounds__(S2_NP_BLOCK) void updateContactsKernel(const s2amdBody* bodies, const float2* origins, const s2amdShape* shapes,
																	 s2amdPairState* pairs, s2amdContact* contacts, int contactCapacity, int32_t* status,
																	 uint8_t* pointBytes, int* summary, int* separatedSlots, const uint8_t* watched)
{
	__shared__ float2 lds[4 * S2_NP_MAX_VERTS * S2_NP_BLOCK]; // vertsA, normsA, vertsB, normsB: 32 KiB
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	int pc = 0;
	if (i < contactCapacity)
	{
		int st = updateContactOne(bodies, origins, shapes, pairs + i, contacts + i, lds);
		status[i] = st;
		if (SUMMARY)
		{
			if (st == S2AMD_PAIR_SEPARATED)
			{
				contacts[i].pointCount = 0;
				pairs[i].shapeA = -1;
				pairs[i].shapeB = -1;
				separatedSlots[atomicAdd(summary + 0, 1)] = i; // the caller's s2DestroyContact list (s2amd_world_separated)
			}
			else
			{
				pc = contacts[i].pointCount;
				pc = pc > 0 ? pc : 0;
			}
			int old = pointBytes[i];
			if (old != pc)
			{
				pointBytes[i] = (uint8_t)pc;
				atomicAdd(summary + 3, 1);
				if ((old > 0) != (pc > 0))
				{
					atomicAdd(summary + 2, 1);
					if (watched != nullptr && watched[i])
					{
						atomicAdd(summary + 5, 1); // on a hub body that is a change of the constraint graph (solver_internal.h: hContactWatched)
					}
				}
			}
		}
	}
	if (SUMMARY)
	{
		unsigned long long live = __ballot(pc > 0);
		if ((threadIdx.x & 63) == 0 && live != 0ull)
		{
			atomicAdd(summary + 1, __popcll(live));
		}
	}
}

struct Scratch
{
	void* p = nullptr;
	~Scratch()
	{
		if (p)
		{
			(void)hipFree(p);
		}
	}
};

} // namespace

#define NP_TRY(expr)                                                                                                             \
	do                                                                                                                           \
	{                                                                                                                            \
		hipError_t _e = (expr);                                                                                                  \
		if (_e != hipSuccess)                                                                                                    \
		{                                                                                                                        \
			return s2amdFail(S2AMD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
		}                                                                                                                        \
	} while (0)

// resident arrays (world.hip)
void launchUpdateContacts(hipStream_t st, const s2amdBody* bodies, const float* origins, const s2amdShape* shapes, s2amdPairState* pairs,
						  s2amdContact* contacts, int contactCapacity, int32_t* status, uint8_t* pointBytes, int* summary, int* separatedSlots, const uint8_t* watched)
{
	if (contactCapacity <= 0)
	{
		return;
	}
	dim3 grid((unsigned)((contactCapacity + S2_NP_BLOCK - 1) / S2_NP_BLOCK));
	updateContactsKernel<true><<<grid, dim3(S2_NP_BLOCK), 0, st>>>(bodies, (const float2*)origins, shapes, pairs, contacts, contactCapacity, status,
																   pointBytes, summary, separatedSlots, watched);
}

#pragma GCC visibility push(default)
extern "C"
{

int s2amd_update_contacts(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const float* origins, const s2amdShape* shapes,
						  int32_t shapeCapacity, s2amdPairState* pairs, s2amdContact* contacts, int32_t contactCapacity, int32_t* status)
{
	if (!solver || bodyCapacity < 0 || shapeCapacity < 0 || contactCapacity < 0 || (bodyCapacity > 0 && (!bodies || !origins)) ||
		(shapeCapacity > 0 && !shapes) || (contactCapacity > 0 && (!pairs || !contacts || !status)))
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument");
	}
	if (contactCapacity == 0)
	{
		return S2AMD_OK;
	}
	for (int i = 0; i < contactCapacity; ++i)
	{
		if (pairs[i].shapeA >= shapeCapacity || pairs[i].shapeB >= shapeCapacity)
		{
			return s2amdFail(S2AMD_E_INVALID, "pair " + std::to_string(i) + " names a shape outside the shape array");
		}
	}
	NP_TRY(hipSetDevice(s2amdDevice(solver)));
	hipStream_t st = s2amdStream(solver);
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	size_t bBytes = (size_t)bodyCapacity * sizeof(s2amdBody), oBytes = (size_t)bodyCapacity * sizeof(float2);
	size_t sBytes = (size_t)shapeCapacity * sizeof(s2amdShape), pBytes = (size_t)contactCapacity * sizeof(s2amdPairState);
	size_t cBytes = (size_t)contactCapacity * sizeof(s2amdContact), tBytes = (size_t)contactCapacity * sizeof(int32_t);
	Scratch buf;
	NP_TRY(hipMalloc(&buf.p, al(bBytes) + al(oBytes) + al(sBytes) + al(pBytes) + al(cBytes) + al(tBytes) + 256));
	char* base = (char*)buf.p;
	s2amdBody* dB = (s2amdBody*)base;
	float2* dO = (float2*)(base + al(bBytes));
	s2amdShape* dS = (s2amdShape*)((char*)dO + al(oBytes));
	s2amdPairState* dP = (s2amdPairState*)((char*)dS + al(sBytes));
	s2amdContact* dC = (s2amdContact*)((char*)dP + al(pBytes));
	int32_t* dT = (int32_t*)((char*)dC + al(cBytes));
	if (bodyCapacity > 0)
	{
		NP_TRY(hipMemcpyAsync(dB, bodies, bBytes, hipMemcpyHostToDevice, st));
		NP_TRY(hipMemcpyAsync(dO, origins, oBytes, hipMemcpyHostToDevice, st));
	}
	if (shapeCapacity > 0)
	{
		NP_TRY(hipMemcpyAsync(dS, shapes, sBytes, hipMemcpyHostToDevice, st));
	}
	NP_TRY(hipMemcpyAsync(dP, pairs, pBytes, hipMemcpyHostToDevice, st));
	NP_TRY(hipMemcpyAsync(dC, contacts, cBytes, hipMemcpyHostToDevice, st));
	dim3 grid((unsigned)((contactCapacity + S2_NP_BLOCK - 1) / S2_NP_BLOCK));
	hipEvent_t e0 = nullptr, e1 = nullptr;
	NP_TRY(hipEventCreate(&e0));
	NP_TRY(hipEventCreate(&e1));
	NP_TRY(hipEventRecord(e0, st));
	updateContactsKernel<false><<<grid, dim3(S2_NP_BLOCK), 0, st>>>(dB, dO, dS, dP, dC, contactCapacity, dT, nullptr, nullptr, nullptr, nullptr);
	NP_TRY(hipEventRecord(e1, st));
	NP_TRY(hipGetLastError());
	NP_TRY(hipMemcpyAsync(pairs, dP, pBytes, hipMemcpyDeviceToHost, st));
	NP_TRY(hipMemcpyAsync(contacts, dC, cBytes, hipMemcpyDeviceToHost, st));
	NP_TRY(hipMemcpyAsync(status, dT, tBytes, hipMemcpyDeviceToHost, st));
	NP_TRY(hipStreamSynchronize(st));
	float ms = 0.0f;
	(void)hipEventElapsedTime(&ms, e0, e1);
	s2amdRecordDeviceMs(solver, ms); // kernel only; the call's wall time is dominated by the host <-> device copies
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop

S2_DEFINE_WARM(narrowphase)
