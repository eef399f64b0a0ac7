// Headless scene constructors for the benchmark / parity corpus.
//
// Public-API-only C: everything here goes through include/solver2d/*.h (s2CreateWorld,
// s2CreateBody, s2Create*Shape, s2CreateRevoluteJoint, ...), so the same file builds against the
// reference library (oracle/_ref, test infrastructure) and against this repo's own host library.
// The scenes are the workloads BASELINE.json names; the recipes follow the reference's GUI
// samples where one exists (samples/collection/sample_contact.cpp:499-561 "Pyramid",
// samples/collection/sample_joints.cpp:365-457 "JointGrid") and SURVEY.md section 8(d) otherwise.
// Everything is deterministic: no rand(), a private LCG where jitter is wanted.

#include "solver2d/solver2d.h"
#include "solver2d/geometry.h"
#include "solver2d/hull.h"
#include "solver2d/joint_types.h"
#include "solver2d/math.h"
#include "solver2d/types.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__)
#define S2SCENE_API __attribute__((visibility("default")))
#else
#define S2SCENE_API
#endif

static uint32_t s_lcg = 12345u;
static void lcgSeed(uint32_t s) { s_lcg = s; }
static float lcgFloat(float lo, float hi)
{
	s_lcg = s_lcg * 1664525u + 1013904223u;
	float u = (float)(s_lcg >> 8) * (1.0f / 16777216.0f);
	return lo + (hi - lo) * u;
}

static s2BodyId makeStaticBox(s2WorldId w, float x, float y, float hx, float hy, float angle)
{
	s2BodyDef bd = s2_defaultBodyDef;
	bd.position = (s2Vec2){x, y};
	bd.angle = angle;
	s2BodyId id = s2CreateBody(w, &bd);
	s2Polygon box = s2MakeBox(hx, hy);
	s2ShapeDef sd = s2_defaultShapeDef;
	s2CreatePolygonShape(id, &sd, &box);
	return id;
}

// One box pyramid with its own static ground, bottom-left brick row starting at originX.
static void addPyramid(s2WorldId w, int base, float originX, float originY, float groundHalfWidth)
{
	makeStaticBox(w, originX, originY - 1.0f, groundHalfWidth, 1.0f, 0.0f);

	s2BodyDef bd = s2_defaultBodyDef;
	bd.type = s2_dynamicBody;
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.density = 1.0f;

	float h = 0.5f;
	s2Polygon box = s2MakeSquare(h);
	float shiftX = 1.0f * h;
	float shiftY = 1.0f * h;

	for (int i = 0; i < base; ++i)
	{
		float y = (2.0f * i + 1.0f) * shiftY;
		for (int j = i; j < base; ++j)
		{
			float x = (i + 1.0f) * shiftX + 2.0f * (j - i) * shiftX - h * base;
			bd.position = (s2Vec2){originX + x, originY + y};
			s2BodyId id = s2CreateBody(w, &bd);
			s2CreatePolygonShape(id, &sd, &box);
		}
	}
}

// BASELINE configs 1 and 2: "Pyramid"/"LargePyramid". p0 = base count.
static void scenePyramid(s2WorldId w, int base)
{
	float ground = base <= 100 ? 100.0f : (float)base;
	addPyramid(w, base, 0.0f, 0.0f, ground);
}

// BASELINE config 5: p0 pyramids of base p1 laid out on a lattice, one static ground each,
// so the world holds p0 disjoint simulation islands.
static void sceneMultiPyramid(s2WorldId w, int count, int base)
{
	int cols = 32;
	float pitchX = (float)base + 20.0f;
	float pitchY = (float)base + 20.0f;
	for (int k = 0; k < count; ++k)
	{
		int cx = k % cols;
		int cy = k / cols;
		addPyramid(w, base, cx * pitchX, cy * pitchY, 0.5f * base + 2.0f);
	}
}

// BASELINE config 4: n x n grid of circles pinned together by revolute joints; the circles do
// not collide with each other (category 2 / mask ~2).
static void sceneJointGrid(s2WorldId w, int numi, int numk)
{
	float rad = 0.4f;
	float shift = 1.0f;
	s2BodyId* bodies = (s2BodyId*)malloc((size_t)numi * numk * sizeof(s2BodyId));
	int index = 0;

	s2ShapeDef sd = s2_defaultShapeDef;
	sd.filter.categoryBits = 2;
	sd.filter.maskBits = ~2u;

	s2Circle circle = {{0.0f, 0.0f}, rad};

	s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
	jd.drawSize = 0.2f;

	for (int k = 0; k < numk; ++k)
	{
		for (int i = 0; i < numi; ++i)
		{
			s2BodyDef bd = s2_defaultBodyDef;
			if (k >= numk / 2 - 3 && k <= numk / 2 + 3 && i == 0)
			{
				bd.type = s2_staticBody;
			}
			else
			{
				bd.type = s2_dynamicBody;
			}
			bd.position = (s2Vec2){k * shift, -i * shift};
			bd.gravityScale = 2.0f;
			s2BodyId body = s2CreateBody(w, &bd);
			s2CreateCircleShape(body, &sd, &circle);

			if (i > 0)
			{
				jd.bodyIdA = bodies[index - 1];
				jd.bodyIdB = body;
				jd.localAnchorA = (s2Vec2){0.0f, -0.5f * shift};
				jd.localAnchorB = (s2Vec2){0.0f, 0.5f * shift};
				s2CreateRevoluteJoint(w, &jd);
			}
			if (k > 0)
			{
				jd.bodyIdA = bodies[index - numi];
				jd.bodyIdB = body;
				jd.localAnchorA = (s2Vec2){0.5f * shift, 0.0f};
				jd.localAnchorB = (s2Vec2){-0.5f * shift, 0.0f};
				s2CreateRevoluteJoint(w, &jd);
			}
			bodies[index++] = body;
		}
	}
	free(bodies);
}

// BASELINE config 3: a motor-driven hollow box full of small boxes. p0 = box count.
static void sceneTumbler(s2WorldId w, int count)
{
	s2BodyDef gd = s2_defaultBodyDef;
	s2BodyId ground = s2CreateBody(w, &gd);

	// side length chosen so the grid of boxes fills ~45% of the drum
	int side = 1;
	while (side * side < count)
	{
		side += 1;
	}
	float a = 0.125f;
	float inner = side * (2.0f * a) * 1.5f + 1.0f;
	float half = 0.5f * inner;
	float wall = 0.5f;

	s2BodyDef bd = s2_defaultBodyDef;
	bd.type = s2_dynamicBody;
	bd.position = (s2Vec2){0.0f, half + 2.0f};
	s2BodyId drum = s2CreateBody(w, &bd);

	s2ShapeDef sd = s2_defaultShapeDef;
	sd.density = 5.0f;
	s2Polygon p;
	p = s2MakeOffsetBox(wall, half + wall, (s2Vec2){half + wall, 0.0f}, 0.0f);
	s2CreatePolygonShape(drum, &sd, &p);
	p = s2MakeOffsetBox(wall, half + wall, (s2Vec2){-half - wall, 0.0f}, 0.0f);
	s2CreatePolygonShape(drum, &sd, &p);
	p = s2MakeOffsetBox(half + wall, wall, (s2Vec2){0.0f, half + wall}, 0.0f);
	s2CreatePolygonShape(drum, &sd, &p);
	p = s2MakeOffsetBox(half + wall, wall, (s2Vec2){0.0f, -half - wall}, 0.0f);
	s2CreatePolygonShape(drum, &sd, &p);

	s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
	jd.bodyIdA = ground;
	jd.bodyIdB = drum;
	jd.localAnchorA = (s2Vec2){0.0f, half + 2.0f};
	jd.localAnchorB = (s2Vec2){0.0f, 0.0f};
	jd.referenceAngle = 0.0f;
	jd.motorSpeed = 0.05f * s2_pi * 4.0f;
	jd.maxMotorTorque = 1e8f;
	jd.enableMotor = true;
	s2CreateRevoluteJoint(w, &jd);

	s2Polygon box = s2MakeSquare(a);
	s2ShapeDef bsd = s2_defaultShapeDef;
	bsd.density = 1.0f;
	float pitch = 2.0f * a * 1.25f;
	float x0 = -0.5f * (side - 1) * pitch;
	float y0 = half + 2.0f - half + a + 0.05f;
	int made = 0;
	for (int r = 0; r < side && made < count; ++r)
	{
		for (int c = 0; c < side && made < count; ++c)
		{
			s2BodyDef b = s2_defaultBodyDef;
			b.type = s2_dynamicBody;
			b.position = (s2Vec2){x0 + c * pitch, y0 + r * pitch};
			s2BodyId id = s2CreateBody(w, &b);
			s2CreatePolygonShape(id, &bsd, &box);
			made += 1;
		}
	}
}

// A small scene that touches every solver code path: a rotated static ramp (non-trivial static
// rotation), boxes at random angles, circles and capsules (1-point manifolds), a hanging chain
// of revolute joints with limits, a motorised arm, a kinematic platform and a mouse joint.
static void sceneMixed(s2WorldId w, int count)
{
	lcgSeed(777u);
	makeStaticBox(w, 0.0f, -1.0f, 40.0f, 1.0f, 0.0f);
	makeStaticBox(w, -12.0f, 3.0f, 6.0f, 0.25f, -0.35f);
	s2BodyId post = makeStaticBox(w, 12.0f, 6.0f, 0.25f, 0.25f, 0.2f);

	s2ShapeDef sd = s2_defaultShapeDef;
	sd.density = 1.0f;

	// kinematic platform sliding sideways
	{
		s2BodyDef bd = s2_defaultBodyDef;
		bd.type = s2_kinematicBody;
		bd.position = (s2Vec2){4.0f, 2.0f};
		bd.linearVelocity = (s2Vec2){-0.5f, 0.0f};
		s2BodyId id = s2CreateBody(w, &bd);
		s2Polygon box = s2MakeBox(2.0f, 0.2f);
		s2CreatePolygonShape(id, &sd, &box);
	}

	s2BodyId firstBox = s2_nullBodyId;
	for (int i = 0; i < count; ++i)
	{
		s2BodyDef bd = s2_defaultBodyDef;
		bd.type = s2_dynamicBody;
		bd.position = (s2Vec2){lcgFloat(-14.0f, 8.0f), 4.0f + 0.9f * (float)(i / 3) + lcgFloat(0.0f, 0.3f)};
		bd.angle = lcgFloat(-1.0f, 1.0f);
		bd.angularVelocity = lcgFloat(-1.0f, 1.0f);
		bd.linearDamping = (i % 5 == 0) ? 0.1f : 0.0f;
		bd.angularDamping = (i % 7 == 0) ? 0.2f : 0.0f;
		s2BodyId id = s2CreateBody(w, &bd);
		sd.friction = 0.2f + 0.1f * (float)(i % 6);
		int kind = i % 4;
		if (kind == 0 || kind == 3)
		{
			s2Polygon box = s2MakeBox(lcgFloat(0.25f, 0.6f), lcgFloat(0.25f, 0.6f));
			s2CreatePolygonShape(id, &sd, &box);
			if (S2_IS_NULL(firstBox))
			{
				firstBox = id;
			}
		}
		else if (kind == 1)
		{
			s2Circle c = {{0.0f, 0.0f}, lcgFloat(0.2f, 0.5f)};
			s2CreateCircleShape(id, &sd, &c);
		}
		else
		{
			s2Capsule c = {{-0.4f, 0.0f}, {0.4f, 0.0f}, lcgFloat(0.15f, 0.3f)};
			s2CreateCapsuleShape(id, &sd, &c);
		}
	}
	sd.friction = 0.6f;

	// hanging chain with limits
	{
		s2BodyId prev = post;
		s2Vec2 prevAnchorLocal = {0.0f, 0.0f};
		for (int i = 0; i < 8; ++i)
		{
			s2BodyDef bd = s2_defaultBodyDef;
			bd.type = s2_dynamicBody;
			bd.position = (s2Vec2){12.0f + 0.5f + 1.0f * i, 6.0f};
			s2BodyId id = s2CreateBody(w, &bd);
			s2Capsule c = {{-0.5f, 0.0f}, {0.5f, 0.0f}, 0.125f};
			s2CreateCapsuleShape(id, &sd, &c);

			s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
			jd.bodyIdA = prev;
			jd.bodyIdB = id;
			jd.localAnchorA = prevAnchorLocal;
			jd.localAnchorB = (s2Vec2){-0.5f, 0.0f};
			jd.enableLimit = (i % 2 == 1);
			jd.lowerAngle = -0.25f * s2_pi;
			jd.upperAngle = 0.1f * s2_pi;
			jd.drawSize = 0.1f;
			s2CreateRevoluteJoint(w, &jd);

			prev = id;
			prevAnchorLocal = (s2Vec2){0.5f, 0.0f};
		}
	}

	// motorised arm
	{
		s2BodyDef bd = s2_defaultBodyDef;
		bd.type = s2_dynamicBody;
		bd.position = (s2Vec2){-2.0f, 9.0f};
		s2BodyId arm = s2CreateBody(w, &bd);
		s2Polygon box = s2MakeBox(1.5f, 0.15f);
		s2CreatePolygonShape(arm, &sd, &box);

		s2BodyDef pd = s2_defaultBodyDef;
		pd.position = (s2Vec2){-2.0f, 9.0f};
		s2BodyId pivot = s2CreateBody(w, &pd);

		s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
		jd.bodyIdA = pivot;
		jd.bodyIdB = arm;
		jd.enableMotor = true;
		jd.motorSpeed = 2.0f;
		jd.maxMotorTorque = 50.0f;
		jd.enableLimit = true;
		jd.lowerAngle = -0.02f;
		jd.upperAngle = 0.02f;
		s2CreateRevoluteJoint(w, &jd);
	}

	// mouse joint dragging the first box
	if (S2_NON_NULL(firstBox))
	{
		s2BodyDef pd = s2_defaultBodyDef;
		s2BodyId anchor = s2CreateBody(w, &pd);
		s2MouseJointDef md = s2DefaultMouseJointDef();
		md.bodyIdA = anchor;
		md.bodyIdB = firstBox;
		md.target = (s2Vec2){0.0f, 8.0f};
		md.hertz = 5.0f;
		md.dampingRatio = 0.7f;
		s2CreateMouseJoint(w, &md);
	}
}

// Tall single stack: one long dependency chain, the worst case for colour-batched sweeps.
static void sceneVerticalStack(s2WorldId w, int count)
{
	makeStaticBox(w, 0.0f, -1.0f, 20.0f, 1.0f, 0.0f);
	s2ShapeDef sd = s2_defaultShapeDef;
	s2Polygon box = s2MakeSquare(0.5f);
	for (int i = 0; i < count; ++i)
	{
		s2BodyDef bd = s2_defaultBodyDef;
		bd.type = s2_dynamicBody;
		bd.position = (s2Vec2){(i & 1) ? 0.01f : -0.01f, 0.5f + 1.0f * i};
		s2BodyId id = s2CreateBody(w, &bd);
		s2CreatePolygonShape(id, &sd, &box);
	}
}

// Circles resting in a V of two rotated static planks: only 1-point manifolds, rolling.
static void sceneCirclePile(s2WorldId w, int count)
{
	makeStaticBox(w, -6.0f, 0.0f, 8.0f, 0.3f, -0.5f);
	makeStaticBox(w, 6.0f, 0.0f, 8.0f, 0.3f, 0.5f);
	s2ShapeDef sd = s2_defaultShapeDef;
	lcgSeed(99u);
	for (int i = 0; i < count; ++i)
	{
		s2BodyDef bd = s2_defaultBodyDef;
		bd.type = s2_dynamicBody;
		bd.position = (s2Vec2){lcgFloat(-3.0f, 3.0f), 0.2f + 0.7f * i};
		s2BodyId id = s2CreateBody(w, &bd);
		s2Circle c = {{0.0f, 0.0f}, lcgFloat(0.25f, 0.45f)};
		s2CreateCircleShape(id, &sd, &c);
	}
}

typedef struct SceneEntry
{
	const char* name;
} SceneEntry;

static const SceneEntry s_scenes[] = {{"pyramid"},		{"multi_pyramid"},	{"joint_grid"},		  {"tumbler"},	  {"mixed"},
									  {"vertical_stack"}, {"circle_pile"},	{"shapes_zoo"},		  {"arch"},		  {"high_mass_ratio"},
									  {"overlap_recovery"}, {"card_house"}, {"far_pyramid"},	  {"far_stack"},  {"far_recovery"},
									  {"far_ragdoll_pile"}, {"far_chain"},	{"ragdoll"},		  {"ball_and_chain"}, {"bridge"},
									  // (round 6) the rest of the reference's 26 samples
									  {"single_box"},	  {"warm_start_energy"}, {"friction_ramp"},  {"rush"},		  {"double_domino"},
									  {"confined"},		  {"circle_stack"},	{"ragdoll_stress"}, {"stretched_chain"}};

S2SCENE_API int s2scene_count(void)
{
	return (int)(sizeof(s_scenes) / sizeof(s_scenes[0]));
}

S2SCENE_API const char* s2scene_name(int index)
{
	if (index < 0 || index >= s2scene_count())
	{
		return NULL;
	}
	return s_scenes[index].name;
}

// Every shape type the narrow phase knows, tumbling into a bowl made of static segments: segments (ground chain),
// rounded polygons (radius > 0), hulls with 3..8 vertices, circles, capsules.  Public API only.
static void sceneShapesZoo(s2WorldId w, int count)
{
	lcgSeed(4242u);
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.density = 1.0f;
	{
		s2BodyDef bd = s2_defaultBodyDef;
		s2BodyId ground = s2CreateBody(w, &bd);
		const s2Vec2 chain[] = {{-12.0f, 6.0f}, {-8.0f, 1.0f}, {-3.0f, 0.0f}, {3.0f, 0.0f}, {8.0f, 1.5f}, {12.0f, 6.0f}};
		for (int i = 0; i + 1 < 6; ++i)
		{
			s2Segment seg = {chain[i], chain[i + 1]};
			s2CreateSegmentShape(ground, &sd, &seg);
		}
		s2Capsule rail = {{-2.0f, 2.5f}, {2.0f, 3.0f}, 0.2f};
		s2CreateCapsuleShape(ground, &sd, &rail);
	}
	for (int i = 0; i < count; ++i)
	{
		s2BodyDef bd = s2_defaultBodyDef;
		bd.type = s2_dynamicBody;
		bd.position = (s2Vec2){lcgFloat(-6.0f, 6.0f), 4.0f + 0.7f * (float)(i / 4) + lcgFloat(0.0f, 0.3f)};
		bd.angle = lcgFloat(-3.0f, 3.0f);
		bd.angularVelocity = lcgFloat(-2.0f, 2.0f);
		s2BodyId id = s2CreateBody(w, &bd);
		sd.friction = 0.1f + 0.15f * (float)(i % 5);
		switch (i % 5)
		{
			case 0:
			{
				s2Polygon box = s2MakeBox(lcgFloat(0.2f, 0.5f), lcgFloat(0.2f, 0.5f));
				box.radius = lcgFloat(0.02f, 0.15f); // rounded
				s2CreatePolygonShape(id, &sd, &box);
				break;
			}
			case 1:
			{
				int n = 3 + (i / 5) % 6; // 3..8 vertices on a squashed circle
				s2Vec2 pts[8];
				float rx = lcgFloat(0.3f, 0.6f), ry = lcgFloat(0.2f, 0.5f);
				for (int k = 0; k < n; ++k)
				{
					float a = 2.0f * s2_pi * (float)k / (float)n;
					pts[k] = (s2Vec2){rx * cosf(a), ry * sinf(a)};
				}
				s2Hull hull = s2ComputeHull(pts, n);
				s2Polygon poly = s2MakePolygon(&hull);
				s2CreatePolygonShape(id, &sd, &poly);
				break;
			}
			case 2:
			{
				s2Circle c = {{lcgFloat(-0.1f, 0.1f), 0.0f}, lcgFloat(0.15f, 0.4f)};
				s2CreateCircleShape(id, &sd, &c);
				break;
			}
			case 3:
			{
				s2Capsule c = {{-0.35f, 0.05f}, {0.35f, -0.05f}, lcgFloat(0.1f, 0.25f)};
				s2CreateCapsuleShape(id, &sd, &c);
				break;
			}
			default:
			{
				s2Polygon box = s2MakeOffsetBox(0.3f, 0.15f, (s2Vec2){0.1f, 0.0f}, 0.4f);
				s2CreatePolygonShape(id, &sd, &box);
				s2Circle c = {{-0.35f, 0.0f}, 0.2f}; // second shape on the same body
				s2CreateCircleShape(id, &sd, &c);
				break;
			}
		}
	}
}


// ---------------------------------------------------------------------------------------------
// The reference's own edge-case samples (the 30 GUI samples ARE its test corpus), restated headless
// through the public API.  Each recipe cites the sample whose construction it follows; numbers (sizes, densities,
// limits) are the sample's, the code is this file's own (table-driven where the sample repeats itself).
// ---------------------------------------------------------------------------------------------

static s2BodyId makeDynamic(s2WorldId w, float x, float y, float angle)
{
	s2BodyDef bd = s2_defaultBodyDef;
	bd.type = s2_dynamicBody;
	bd.position = (s2Vec2){x, y};
	bd.angle = angle;
	return s2CreateBody(w, &bd);
}

static s2BodyId makeGroundSegment(s2WorldId w, float ox, float oy, float halfLength, const s2ShapeDef* sd)
{
	s2BodyDef bd = s2_defaultBodyDef;
	bd.position = (s2Vec2){ox, oy};
	s2BodyId ground = s2CreateBody(w, &bd);
	s2Segment seg = {{-halfLength, 0.0f}, {halfLength, 0.0f}};
	s2CreateSegmentShape(ground, sd, &seg);
	return ground;
}

static void addQuad(s2WorldId w, const s2ShapeDef* sd, s2Vec2 a, s2Vec2 b, s2Vec2 c, s2Vec2 d)
{
	s2BodyId id = makeDynamic(w, 0.0f, 0.0f, 0.0f);
	s2Vec2 ps[4] = {a, b, c, d};
	s2Hull hull = s2ComputeHull(ps, 4);
	s2Polygon poly = s2MakePolygon(&hull);
	s2CreatePolygonShape(id, sd, &poly);
}

// "Arch" (samples/collection/sample_contact.cpp:666-759): 17 voussoirs cut from two catenary-like curves, four slabs on
// the keystone, segment ground.  Wedge-shaped polygons under compression: friction-dominated, all 2-point manifolds
// between non-box hulls.  The curve tables are the sample's data.
static void sceneArch(s2WorldId w)
{
	static const float inner[9][2] = {{16.0f, 0.0f},
									  {14.93803712795643f, 5.133601056842984f},
									  {13.79871746027416f, 10.24928069555078f},
									  {12.56252963284711f, 15.34107019122473f},
									  {11.20040987372525f, 20.39856541571217f},
									  {9.66521217819836f, 25.40369899225096f},
									  {7.87179930638133f, 30.3179337000085f},
									  {5.635199558196225f, 35.03820717801641f},
									  {2.405937953536585f, 39.09554102558315f}};
	static const float outer[9][2] = {{24.0f, 0.0f},
									  {22.33619528222415f, 6.02299846205841f},
									  {20.54936888969905f, 12.00964361211476f},
									  {18.60854610798073f, 17.9470321677465f},
									  {16.46769273811807f, 23.81367936585418f},
									  {14.05325025774858f, 29.57079353071012f},
									  {11.23551045834022f, 35.13775818285372f},
									  {7.752568160730571f, 40.30450679009583f},
									  {3.016931552701656f, 44.28891593799322f}};
	const float scale = 0.25f;
	s2Vec2 in[9], out[9];
	for (int i = 0; i < 9; ++i)
	{
		in[i] = s2MulSV(scale, (s2Vec2){inner[i][0], inner[i][1]});
		out[i] = s2MulSV(scale, (s2Vec2){outer[i][0], outer[i][1]});
	}
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.6f;
	makeGroundSegment(w, 0.0f, 0.0f, 100.0f, &sd);
	for (int i = 0; i < 8; ++i) // right leg, ground up
	{
		addQuad(w, &sd, in[i], out[i], out[i + 1], in[i + 1]);
	}
	for (int i = 0; i < 8; ++i) // left leg: the mirror image
	{
		addQuad(w, &sd, (s2Vec2){-out[i].x, out[i].y}, (s2Vec2){-in[i].x, in[i].y}, (s2Vec2){-in[i + 1].x, in[i + 1].y},
				(s2Vec2){-out[i + 1].x, out[i + 1].y});
	}
	addQuad(w, &sd, in[8], out[8], (s2Vec2){-out[8].x, out[8].y}, (s2Vec2){-in[8].x, in[8].y}); // keystone
	s2Polygon slab = s2MakeBox(2.0f, 0.5f);
	for (int i = 0; i < 4; ++i)
	{
		s2BodyId id = makeDynamic(w, 0.0f, 0.5f + out[8].y + 1.0f * i, 0.0f);
		s2CreatePolygonShape(id, &sd, &slab);
	}
}

// "High Mass Ratio 1/2/3" (sample_contact.cpp:122-299).  variant 1: three 10-base pyramids of 2 m boxes whose top box
// (dropped from 2 m above) weighs 100x, 200x, 300x a normal one; 2: a 20 m box falling onto two 1 m boxes on a
// segment; 3: the same on a thick box ground.  Mass ratios of 400:1 are where the ten solvers differ most.
static void sceneHighMassRatio(s2WorldId w, int variant)
{
	s2ShapeDef sd = s2_defaultShapeDef;
	if (variant <= 1)
	{
		const float e = 1.0f;
		sd.friction = 0.5f;
		makeGroundSegment(w, 0.0f, 0.0f, 66.0f * e, &sd);
		s2Polygon box = s2MakeBox(e, e);
		for (int pile = 0; pile < 3; ++pile)
		{
			const float offset = -20.0f * e + 2.0f * (10 + 1.0f) * e * pile;
			float y = e;
			for (int row = 10; row > 0; --row, y += 2.0f * e)
			{
				for (int i = 0; i < row; ++i)
				{
					s2BodyId id = makeDynamic(w, 2.0f * (i - 0.5f * row) * e + offset, row == 1 ? y + 2.0f : y, 0.0f);
					sd.density = row == 1 ? (pile + 1.0f) * 100.0f : 1.0f;
					s2CreatePolygonShape(id, &sd, &box);
				}
			}
		}
		return;
	}
	sd.density = 1.0f;
	if (variant == 2)
	{
		makeGroundSegment(w, 0.0f, 0.0f, 20.0f, &sd);
	}
	else
	{
		makeStaticBox(w, 0.0f, -2.0f, 40.0f, 2.0f, 0.0f);
	}
	const float e = 1.0f;
	s2Polygon small = s2MakeBox(0.5f * e, 0.5f * e), big = s2MakeBox(10.0f * e, 10.0f * e);
	s2CreatePolygonShape(makeDynamic(w, -9.0f * e, 0.5f * e, 0.0f), &sd, &small);
	s2CreatePolygonShape(makeDynamic(w, 9.0f * e, 0.5f * e, 0.0f), &sd, &small);
	s2CreatePolygonShape(makeDynamic(w, 0.0f, (10.0f + 16.0f) * e, 0.0f), &sd, &big);
}

// "Overlap Recovery" (sample_contact.cpp:368-418) and, with a far origin, "Far/Recovery" (sample_far.cpp:180-233): a
// 4-base pyramid of unit boxes created 25 % inside each other -- large negative separations, the push-out path of every
// solver (Baumgarte caps, soft-contact bias caps, NGS max correction).
static void sceneOverlapRecovery(s2WorldId w, float ox, float oy)
{
	const int baseCount = 4;
	const float overlap = 0.25f, extent = 0.5f;
	makeGroundSegment(w, ox, oy, 40.0f, &s2_defaultShapeDef);
	s2Polygon box = s2MakeSquare(extent);
	const float fraction = 1.0f - overlap;
	float y = extent;
	for (int i = 0; i < baseCount; ++i, y += 2.0f * fraction * extent)
	{
		float x = fraction * extent * (i - baseCount);
		for (int j = i; j < baseCount; ++j, x += 2.0f * fraction * extent)
		{
			s2CreatePolygonShape(makeDynamic(w, ox + x, oy + y, 0.0f), &s2_defaultShapeDef, &box);
		}
	}
}

// "Card House" (sample_contact.cpp:888-963, from PEEL): 2 mm thick cards leaning at +-25 degrees with horizontal cards
// across the peaks, five storeys.  Extreme aspect ratios, tiny inertia, friction 0.7.
static void sceneCardHouse(s2WorldId w)
{
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.7f;
	{
		s2BodyDef bd = s2_defaultBodyDef;
		bd.position = (s2Vec2){0.0f, -2.0f};
		s2BodyId ground = s2CreateBody(w, &bd);
		s2Polygon g = s2MakeBox(40.0f, 2.0f);
		s2CreatePolygonShape(ground, &sd, &g);
	}
	const float cardHeight = 0.2f, cardThickness = 0.001f;
	const float lean = 25.0f * s2_pi / 180.0f, flat = 0.5f * s2_pi;
	s2Polygon card = s2MakeBox(cardThickness, cardHeight);
	float z0 = 0.0f, y = cardHeight - 0.02f;
	for (int storey = 5; storey > 0; --storey)
	{
		float z = z0;
		for (int i = 0; i < storey; ++i)
		{
			if (i != storey - 1)
			{
				s2CreatePolygonShape(makeDynamic(w, z + 0.25f, y + cardHeight - 0.015f, flat), &sd, &card);
			}
			s2CreatePolygonShape(makeDynamic(w, z, y, -lean), &sd, &card);
			z += 0.175f;
			s2CreatePolygonShape(makeDynamic(w, z, y, lean), &sd, &card);
			z += 0.175f;
		}
		y += cardHeight * 2.0f - 0.03f;
		z0 += 0.175f;
	}
}

// "Far/Pyramid" (sample_far.cpp:15-83): a 10-base pyramid with a 25 % gap between boxes, 100 km from the origin, where
// one float ulp of a coordinate is 7.8 mm -- larger than the linear slop.  Exercises the centre-of-mass-relative
// arithmetic (delta positions, anchors relative to the body) that the TGS solvers rely on.
static void sceneFarPyramid(s2WorldId w, float ox, float oy)
{
	makeStaticBox(w, ox, oy - 1.0f, 100.0f, 1.0f, 0.0f);
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.density = 1.0f;
	const int baseCount = 10;
	const float h = 0.5f, shift = 1.25f * h;
	s2Polygon box = s2MakeSquare(h);
	for (int i = 0; i < baseCount; ++i)
	{
		float y = (2.0f * i + 1.0f) * shift + 0.5f;
		for (int j = i; j < baseCount; ++j)
		{
			float x = (i + 1.0f) * shift + 2.0f * (j - i) * shift - h * baseCount;
			s2CreatePolygonShape(makeDynamic(w, x + ox, y + oy, 0.0f), &sd, &box);
		}
	}
}

// "Far/Stack" (sample_far.cpp:85-164): a plank balanced on a small circle and a small box, two boxes on the plank;
// 47 km from the origin.
static void sceneFarStack(s2WorldId w, float ox, float oy)
{
	makeStaticBox(w, ox, oy - 1.0f, 10.0f, 1.0f, 0.0f);
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.density = 1.0f;
	s2Circle ball = {{0.0f, 0.0f}, 0.1f};
	s2CreateCircleShape(makeDynamic(w, ox + 1.875f, oy + 0.125f, 0.0f), &sd, &ball);
	static const float boxes[4][4] = {// x, y, hx, hy
									  {-1.875f, 0.15f, 0.1f, 0.125f},
									  {0.0f, 0.325f, 2.0f, 0.05f},
									  {-0.5f, 0.9f, 0.25f, 0.25f},
									  {-0.55f, 1.7f, 0.5f, 0.5f}};
	for (int i = 0; i < 4; ++i)
	{
		s2Polygon b = s2MakeBox(boxes[i][2], boxes[i][3]);
		s2CreatePolygonShape(makeDynamic(w, ox + boxes[i][0], oy + boxes[i][1], 0.0f), &sd, &b);
	}
}

// The ragdoll of samples/collection/human.cpp as a table: per bone the parent, the body height, one or two capsules
// (centre-line end points and radius, all times the scale), the pivot height, the limit angles in units of pi and the
// motor torque as a fraction of 0.025 * scale.  Every joint has limit AND motor enabled (motor speed 0: joint friction),
// all shapes share a negative group index so the bones of one ragdoll never collide with each other.
typedef struct BoneRow
{
	int parent;
	float bodyY;
	float cap[2][5]; // x1, y1, x2, y2, radius; radius 0 = no second capsule
	int footSecond;	 // the second capsule uses the low-friction foot material
	float pivotY, lower, upper, torque;
} BoneRow;

static const BoneRow s_human[11] = {
	{-1, 0.95f, {{0.0f, -0.02f, 0.0f, 0.025f, 0.095f}, {0}}, 0, 0.0f, 0.0f, 0.0f, 0.0f},						   // hip
	{0, 1.2f, {{0.0f, -0.135f, 0.0f, 0.135f, 0.09f}, {0}}, 0, 1.025f, -0.25f, 0.0f, 0.5f},						   // torso
	{1, 1.5f, {{0.0f, -0.0325f, 0.0f, 0.0325f, 0.08f}, {0.0f, -0.12f, 0.0f, -0.08f, 0.05f}}, 0, 1.4f, -0.3f, 0.1f, 0.25f}, // head + neck
	{0, 0.775f, {{0.0f, -0.125f, 0.0f, 0.125f, 0.055f}, {0}}, 0, 0.9f, -0.05f, 0.4f, 1.0f},					   // upper left leg
	{3, 0.475f, {{0.0f, -0.14f, 0.0f, 0.125f, 0.045f}, {-0.02f, -0.175f, 0.13f, -0.175f, 0.03f}}, 1, 0.625f, -0.5f, -0.02f, 0.5f}, // lower left leg + foot
	{0, 0.775f, {{0.0f, -0.125f, 0.0f, 0.125f, 0.055f}, {0}}, 0, 0.9f, -0.05f, 0.4f, 1.0f},					   // upper right leg
	{5, 0.475f, {{0.0f, -0.14f, 0.0f, 0.125f, 0.045f}, {-0.02f, -0.175f, 0.13f, -0.175f, 0.03f}}, 1, 0.625f, -0.5f, -0.02f, 0.5f}, // lower right leg + foot
	{1, 1.225f, {{0.0f, -0.125f, 0.0f, 0.125f, 0.035f}, {0}}, 0, 1.35f, -0.05f, 0.8f, 0.25f},					   // upper left arm
	{7, 0.975f, {{0.0f, -0.125f, 0.0f, 0.125f, 0.03f}, {0}}, 0, 1.1f, 0.01f, 0.5f, 0.1f},						   // lower left arm
	{1, 1.225f, {{0.0f, -0.125f, 0.0f, 0.125f, 0.035f}, {0}}, 0, 1.35f, -0.05f, 0.8f, 0.25f},					   // upper right arm
	{9, 0.975f, {{0.0f, -0.125f, 0.0f, 0.125f, 0.03f}, {0}}, 0, 1.1f, 0.01f, 0.5f, 0.1f},						   // lower right arm
};

static void spawnHuman(s2WorldId w, float px, float py, float scale, int groupIndex)
{
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.4f;
	sd.filter.groupIndex = -groupIndex;
	s2ShapeDef foot = sd;
	foot.friction = 0.1f;
	const float s = scale, maxTorque = 0.025f * s;
	s2BodyId bones[11];
	for (int i = 0; i < 11; ++i)
	{
		const BoneRow* r = s_human + i;
		bones[i] = makeDynamic(w, px + 0.0f, py + r->bodyY * s, 0.0f);
		for (int c = 0; c < 2; ++c)
		{
			if (r->cap[c][4] > 0.0f)
			{
				s2Capsule cap = {{r->cap[c][0] * s, r->cap[c][1] * s}, {r->cap[c][2] * s, r->cap[c][3] * s}, r->cap[c][4] * s};
				s2CreateCapsuleShape(bones[i], (c == 1 && r->footSecond) ? &foot : &sd, &cap);
			}
		}
		if (r->parent >= 0)
		{
			s2Vec2 pivot = {px + 0.0f, py + r->pivotY * s};
			s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
			jd.bodyIdA = bones[r->parent];
			jd.bodyIdB = bones[i];
			jd.localAnchorA = s2Body_GetLocalPoint(jd.bodyIdA, pivot);
			jd.localAnchorB = s2Body_GetLocalPoint(jd.bodyIdB, pivot);
			jd.enableLimit = true;
			jd.lowerAngle = r->lower * s2_pi;
			jd.upperAngle = r->upper * s2_pi;
			jd.enableMotor = true;
			jd.maxMotorTorque = r->torque * maxTorque;
			jd.drawSize = 0.025f;
			s2CreateRevoluteJoint(w, &jd);
		}
	}
}

// "Ragdoll" (sample_joints.cpp:208-240): one ragdoll dropped from 4 m onto a box ground: ten revolute joints with
// limits and motors -- the relative-angle (atan2f) path under load.
static void sceneRagdoll(s2WorldId w)
{
	makeStaticBox(w, 0.0f, -1.0f, 20.0f, 1.0f, 0.0f);
	spawnHuman(w, 0.0f, 4.0f, 1.0f, 1);
}

// "Far/Ragdoll Pile" (sample_far.cpp:235-281): six ragdolls dropped into a V of two tilted planks (two shapes on one
// static body), 6 km from the origin.
static void sceneFarRagdollPile(s2WorldId w, float ox, float oy)
{
	s2BodyDef bd = s2_defaultBodyDef;
	bd.position = (s2Vec2){ox, oy - 1.0f};
	s2BodyId ground = s2CreateBody(w, &bd);
	s2Polygon plank = s2MakeOffsetBox(10.0f, 0.5f, (s2Vec2){-5.0f, 2.0f}, -0.15f * s2_pi);
	s2CreatePolygonShape(ground, &s2_defaultShapeDef, &plank);
	plank = s2MakeOffsetBox(10.0f, 0.5f, (s2Vec2){5.0f, 2.0f}, 0.15f * s2_pi);
	s2CreatePolygonShape(ground, &s2_defaultShapeDef, &plank);
	static const float at[6][2] = {{0.0f, 0.5f}, {-0.2f, 1.0f}, {0.2f, 1.0f}, {-0.4f, 1.5f}, {0.4f, 1.5f}, {0.0f, 2.0f}};
	for (int i = 0; i < 6; ++i)
	{
		spawnHuman(w, ox + at[i][0], oy + at[i][1], 1.0f, i + 1);
	}
}

// A chain of `count` capsule links of half-length hx hanging from a static body at (ox, oy), optionally ending in a
// heavy ball: "Ball & Chain" (sample_joints.cpp:107-188; 40 links of 0.5 m, ball radius 8 m -- mass ratio ~1600:1 along a
// joint chain) and "Far Chain" (sample_far.cpp:283-342; 40 links of 0.1 m, no ball, 53 km from the origin, anchors
// given in local coordinates because s2Body_GetLocalPoint would lose them to rounding out there).
static void sceneChain(s2WorldId w, float ox, float oy, int count, float hx, float radius, float ballRadius)
{
	s2BodyDef gd = s2_defaultBodyDef;
	gd.position = (s2Vec2){ox, oy};
	s2BodyId prev = s2CreateBody(w, &gd);
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.density = 20.0f;
	s2Capsule link = {{-hx, 0.0f}, {hx, 0.0f}, radius};
	s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
	jd.drawSize = 0.1f;
	s2Vec2 prevLocal = {0.0f, count * hx};
	for (int i = 0; i <= count; ++i)
	{
		const int ball = i == count;
		if (ball && ballRadius <= 0.0f)
		{
			break;
		}
		s2BodyDef bd = s2_defaultBodyDef;
		bd.type = s2_dynamicBody;
		bd.linearDamping = 0.1f;
		bd.angularDamping = 0.1f;
		float localX = ball ? -(ballRadius - hx) - hx : -hx; // the pivot in the new body's frame
		bd.position = (s2Vec2){ox + (ball ? (1.0f + 2.0f * count) * hx + ballRadius - hx : (1.0f + 2.0f * i) * hx), oy + count * hx};
		s2BodyId id = s2CreateBody(w, &bd);
		if (ball)
		{
			s2Circle c = {{0.0f, 0.0f}, ballRadius};
			s2CreateCircleShape(id, &sd, &c);
		}
		else
		{
			s2CreateCapsuleShape(id, &sd, &link);
		}
		jd.bodyIdA = prev;
		jd.bodyIdB = id;
		jd.localAnchorA = prevLocal;
		jd.localAnchorB = (s2Vec2){localX, 0.0f};
		s2CreateRevoluteJoint(w, &jd);
		prevLocal = (s2Vec2){hx, 0.0f};
		prev = id;
	}
}

// "Bridge" (sample_joints.cpp:17-92): `count` planks (1 m x 0.25 m, density 20) hinged end to end between two points of
// one static body, 20 m up: a closed joint chain, the hardest case for sequential-impulse joint solvers.
static void sceneBridge(s2WorldId w, int count)
{
	s2BodyDef gd = s2_defaultBodyDef;
	s2BodyId ground = s2CreateBody(w, &gd);
	s2Polygon plank = s2MakeBox(0.5f, 0.125f);
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.density = 20.0f;
	s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
	jd.drawSize = 0.1f;
	const float xbase = -0.5f * count;
	s2BodyId prev = ground;
	for (int i = 0; i <= count; ++i)
	{
		s2BodyId id = ground;
		if (i < count)
		{
			s2BodyDef bd = s2_defaultBodyDef;
			bd.type = s2_dynamicBody;
			bd.position = (s2Vec2){xbase + 0.5f + 1.0f * i, 20.0f};
			bd.linearDamping = 0.1f;
			bd.angularDamping = 0.1f;
			id = s2CreateBody(w, &bd);
			s2CreatePolygonShape(id, &sd, &plank);
		}
		s2Vec2 pivot = {xbase + 1.0f * i, 20.0f};
		jd.bodyIdA = prev;
		jd.bodyIdB = id;
		jd.localAnchorA = s2Body_GetLocalPoint(prev, pivot);
		jd.localAnchorB = s2Body_GetLocalPoint(id, pivot);
		s2CreateRevoluteJoint(w, &jd);
		prev = id;
	}
}

// ---- the reference's remaining samples (round 6): every "Contact" and "Joints" sample has a constructor here ----
// Three of them do something in their Step override besides stepping the world: s2scene_pre_step / s2scene_post_step below are
// those overrides, so a headless loop  pre_step -> s2World_Step -> post_step  is the sample as the GUI runs it.

// what a sample keeps between steps, per world slot (the reference has s2_maxWorlds = 32 of them, constants.h:12)
#define SCENE_MAX_WORLDS 32
#define RUSH_COUNT 400		// sample_contact.cpp:572
#define STRESS_HUMANS 32	// sample_joints.cpp:212
#define HUMAN_BONES 11
typedef struct HumanIds
{
	s2BodyId bones[HUMAN_BONES];
	s2JointId joints[HUMAN_BONES];
	int spawned;
} HumanIds;
typedef struct SceneState
{
	int kind; // 0 nothing to do between steps, 1 warm_start_energy, 2 rush, 3 ragdoll_stress
	uint16_t revision;
	s2BodyId top;
	s2BodyId rush[RUSH_COUNT];
	HumanIds humans[STRESS_HUMANS];
	float wait, side;
} SceneState;
static SceneState s_state[SCENE_MAX_WORLDS];

static SceneState* stateOf(s2WorldId w, int create)
{
	if (w.index < 0 || w.index >= SCENE_MAX_WORLDS)
	{
		return NULL;
	}
	SceneState* st = s_state + w.index;
	if (create)
	{
		memset(st, 0, sizeof(*st));
		st->revision = w.revision;
	}
	return st->revision == w.revision ? st : NULL;
}

// "Single Box" (sample_contact.cpp:13-51): one 2 m box dropped from 4 m onto a segment
static void sceneSingleBox(s2WorldId w)
{
	const float extent = 1.0f;
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.5f;
	makeGroundSegment(w, 0.0f, 0.0f, 0.5f * 2.0f * 66.0f * extent, &sd);
	s2Polygon box = s2MakeBox(extent, extent);
	s2CreatePolygonShape(makeDynamic(w, 0.0f, 4.0f, 0.0f), &sd, &box);
}

// "Warm Start Energy" (sample_contact.cpp:53-118): three touching circles in a column, the top one a hundred times denser; after
// 120 steps the top one is destroyed (s2scene_pre_step) and what the warm start has stored pushes the other two apart
static void sceneWarmStartEnergy(s2WorldId w)
{
	SceneState* st = stateOf(w, 1);
	s2ShapeDef sd = s2_defaultShapeDef;
	makeGroundSegment(w, 0.0f, 0.0f, 10.0f, &sd);
	s2Circle circle = {{0.0f, 0.0f}, 0.5f};
	const float separation = 0.0f;
	for (int i = 0; i < 3; ++i)
	{
		s2BodyId id = makeDynamic(w, 0.0f, 0.5f + (float)i + separation, 0.0f);
		sd.density = i == 2 ? 100.0f : 1.0f;
		s2CreateCircleShape(id, &sd, &circle);
		if (i == 2 && st != NULL)
		{
			st->kind = 1;
			st->top = id;
		}
	}
}

// "Friction Ramp" (sample_contact.cpp:300-366): five boxes of friction 0.75 ... 0 sliding down three ramps of friction 0.2 --
// the one sample whose contacts mix two different shape frictions (src/contact.c:179)
static void sceneFrictionRamp(s2WorldId w)
{
	s2BodyId ground = s2CreateBody(w, &s2_defaultBodyDef);
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.2f;
	s2Segment segment = {{-40.0f, 0.0f}, {40.0f, 0.0f}};
	s2CreateSegmentShape(ground, &sd, &segment);
	const float ramps[5][5] = {{13.0f, 0.25f, -4.0f, 22.0f, -0.25f}, {0.25f, 1.0f, 10.5f, 19.0f, 0.0f}, {13.0f, 0.25f, 4.0f, 14.0f, 0.25f},
							   {0.25f, 1.0f, -10.5f, 11.0f, 0.0f},	 {13.0f, 0.25f, -4.0f, 6.0f, -0.25f}};
	for (int i = 0; i < 5; ++i)
	{
		s2Polygon box = s2MakeOffsetBox(ramps[i][0], ramps[i][1], (s2Vec2){ramps[i][2], ramps[i][3]}, ramps[i][4]);
		s2CreatePolygonShape(ground, &sd, &box);
	}
	s2Polygon box = s2MakeBox(0.5f, 0.5f);
	s2ShapeDef bd = s2_defaultShapeDef;
	bd.density = 25.0f;
	const float friction[5] = {0.75f, 0.5f, 0.35f, 0.1f, 0.0f};
	for (int i = 0; i < 5; ++i)
	{
		bd.friction = friction[i];
		s2CreatePolygonShape(makeDynamic(w, -15.0f + 4.0f * (float)i, 28.0f, 0.0f), &bd, &box);
	}
}

// "Rush" (sample_contact.cpp:562-661): 400 weightless circles on a spiral around a static one, pulled to the centre by a force of
// 1000 N applied before every step (s2scene_pre_step: s2Body_ApplyForceToCenter, the one sample that uses applied forces)
static void sceneRush(s2WorldId w, int count)
{
	SceneState* st = stateOf(w, 1);
	s2BodyDef bd = s2_defaultBodyDef;
	s2BodyId ground = s2CreateBody(w, &bd);
	s2Circle circle = {{0.0f, 0.0f}, 0.5f};
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.2f;
	sd.density = 100.0f;
	s2CreateCircleShape(ground, &sd, &circle);
	float distance = 5.0f, angle = 0.0f;
	const float deltaAngle = 1.0f / distance, deltaDistance = 0.05f;
	bd.type = s2_dynamicBody;
	bd.gravityScale = 0.0f;
	if (count > RUSH_COUNT)
	{
		count = RUSH_COUNT;
	}
	for (int i = 0; i < count; ++i)
	{
		bd.position = (s2Vec2){distance * cosf(angle), distance * sinf(angle)};
		s2BodyId id = s2CreateBody(w, &bd);
		s2CreateCircleShape(id, &sd, &circle);
		if (st != NULL)
		{
			st->rush[i] = id;
		}
		angle += deltaAngle;
		distance += deltaDistance;
	}
	if (st != NULL)
	{
		st->kind = 2;
		st->wait = (float)count; // (how many there are)
	}
}

// "Double Domino" (sample_contact.cpp:761-812): fifteen dominoes, the first one tipped by an impulse at creation
static void sceneDoubleDomino(s2WorldId w)
{
	makeStaticBox(w, 0.0f, -1.0f, 100.0f, 1.0f, 0.0f);
	s2Polygon box = s2MakeBox(0.125f, 0.5f);
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.6f;
	const int count = 15;
	float x = -0.5f * (float)count;
	for (int i = 0; i < count; ++i)
	{
		s2BodyId id = makeDynamic(w, x, 0.5f, 0.0f);
		s2CreatePolygonShape(id, &sd, &box);
		if (i == 0)
		{
			s2Body_ApplyLinearImpulse(id, (s2Vec2){0.2f, 0.0f}, (s2Vec2){x, 1.0f});
		}
		x += 1.0f;
	}
}

// "Confined" (sample_contact.cpp:814-886): 625 weightless circles of radius 0.5 on a lattice of pitch 0.72 inside a box of four
// static capsules -- every circle overlaps its neighbours at the start, nothing can get out
static void sceneConfined(s2WorldId w, int gridCount)
{
	s2BodyId ground = s2CreateBody(w, &s2_defaultBodyDef);
	const float walls[4][4] = {{-10.5f, 0.0f, 10.5f, 0.0f}, {-10.5f, 0.0f, -10.5f, 20.5f}, {10.5f, 0.0f, 10.5f, 20.5f}, {-10.5f, 20.5f, 10.5f, 20.5f}};
	for (int i = 0; i < 4; ++i)
	{
		s2Capsule capsule = {{walls[i][0], walls[i][1]}, {walls[i][2], walls[i][3]}, 0.5f};
		s2CreateCapsuleShape(ground, &s2_defaultShapeDef, &capsule);
	}
	s2BodyDef bd = s2_defaultBodyDef;
	bd.type = s2_dynamicBody;
	bd.gravityScale = 0.0f;
	s2Circle circle = {{0.0f, 0.0f}, 0.5f};
	for (int column = 0; column < gridCount; ++column)
	{
		for (int row = 0; row < gridCount; ++row)
		{
			bd.position = (s2Vec2){-8.75f + (float)column * 18.0f / (float)gridCount, 1.5f + (float)row * 18.0f / (float)gridCount};
			s2CreateCircleShape(s2CreateBody(w, &bd), &s2_defaultShapeDef, &circle);
		}
	}
}

// "Circle Stack" (sample_contact.cpp:971-1010): ten circles of radius 1 dropped in a column, 3 m apart
static void sceneCircleStack(s2WorldId w, int count)
{
	s2ShapeDef sd = s2_defaultShapeDef;
	makeGroundSegment(w, 0.0f, 0.0f, 40.0f, &sd);
	s2Circle circle = {{0.0f, 0.0f}, 1.0f};
	for (int i = 0; i < count; ++i)
	{
		s2CreateCircleShape(makeDynamic(w, 0.0f, 4.0f + 3.0f * (float)i, 0.0f), &sd, &circle);
	}
}

// the ragdoll of spawnHuman with its ids kept (samples/collection/human.cpp:24-347), so that it can be taken out of the world again
static void spawnHumanKept(s2WorldId w, float px, float py, float scale, int groupIndex, HumanIds* out)
{
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.4f;
	sd.filter.groupIndex = -groupIndex;
	s2ShapeDef foot = sd;
	foot.friction = 0.1f;
	const float s = scale, maxTorque = 0.025f * s;
	for (int i = 0; i < HUMAN_BONES; ++i)
	{
		const BoneRow* r = s_human + i;
		out->bones[i] = makeDynamic(w, px + 0.0f, py + r->bodyY * s, 0.0f);
		out->joints[i] = s2_nullJointId;
		for (int c = 0; c < 2; ++c)
		{
			if (r->cap[c][4] > 0.0f)
			{
				s2Capsule cap = {{r->cap[c][0] * s, r->cap[c][1] * s}, {r->cap[c][2] * s, r->cap[c][3] * s}, r->cap[c][4] * s};
				s2CreateCapsuleShape(out->bones[i], (c == 1 && r->footSecond) ? &foot : &sd, &cap);
			}
		}
		if (r->parent >= 0)
		{
			s2Vec2 pivot = {px + 0.0f, py + r->pivotY * s};
			s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
			jd.bodyIdA = out->bones[r->parent];
			jd.bodyIdB = out->bones[i];
			jd.localAnchorA = s2Body_GetLocalPoint(jd.bodyIdA, pivot);
			jd.localAnchorB = s2Body_GetLocalPoint(jd.bodyIdB, pivot);
			jd.enableLimit = true;
			jd.lowerAngle = r->lower * s2_pi;
			jd.upperAngle = r->upper * s2_pi;
			jd.enableMotor = true;
			jd.maxMotorTorque = r->torque * maxTorque;
			jd.drawSize = 0.025f;
			out->joints[i] = s2CreateRevoluteJoint(w, &jd);
		}
	}
	out->spawned = 1;
}

// Ragdoll Stress's CreateElement (sample_joints.cpp:293-318): the first free slot gets a ragdoll of scale 2, left and right in turn
static void stressCreateElement(s2WorldId w, SceneState* st)
{
	for (int i = 0; i < STRESS_HUMANS; ++i)
	{
		if (!st->humans[i].spawned)
		{
			spawnHumanKept(w, st->side, 28.0f, 2.0f, i + 1, st->humans + i);
			st->side = -st->side;
			return;
		}
	}
}

// "Ragdoll Stress" (sample_joints.cpp:207-362): a funnel of twenty static capsules with three motorised paddles in it; a ragdoll is
// dropped in every half second and taken out of the world when it has fallen through (s2scene_post_step): bodies, shapes and
// joints created and destroyed while the world runs
static void sceneRagdollStress(s2WorldId w)
{
	SceneState* st = stateOf(w, 1);
	s2BodyId ground = s2CreateBody(w, &s2_defaultBodyDef);
	const s2Vec2 points[20] = {
		{-16.8672504f, 31.088623f},	   {16.8672485f, 31.088623f},	 {16.8672485f, 17.1978741f}, {8.26824951f, 11.906374f},
		{16.8672485f, 11.906374f},	   {16.8672485f, -0.661376953f}, {8.26824951f, -5.953125f},	 {16.8672485f, -5.953125f},
		{16.8672485f, -13.229126f},	   {3.63799858f, -23.151123f},	 {3.63799858f, -31.088623f}, {-3.63800049f, -31.088623f},
		{-3.63800049f, -23.151123f},   {-16.8672504f, -13.229126f},	 {-16.8672504f, -5.953125f}, {-8.26825142f, -5.953125f},
		{-16.8672504f, -0.661376953f}, {-16.8672504f, 11.906374f},	 {-8.26825142f, 11.906374f}, {-16.8672504f, 17.1978741f},
	};
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.friction = 0.2f;
	for (int i = 0; i < 20; ++i)
	{
		s2Capsule capsule = {points[i], points[(i + 1) % 20], 0.5f};
		s2CreateCapsuleShape(ground, &sd, &capsule);
	}
	float sign = 1.0f, y = 14.0f;
	for (int i = 0; i < 3; ++i)
	{
		s2BodyId id = makeDynamic(w, 0.0f, y, 0.0f);
		s2Polygon box = s2MakeBox(6.0f, 0.5f);
		s2ShapeDef pd = s2_defaultShapeDef;
		pd.friction = 0.1f;
		pd.restitution = 1.0f;
		pd.density = 1.0f;
		s2CreatePolygonShape(id, &pd, &box);
		s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
		jd.bodyIdA = ground;
		jd.bodyIdB = id;
		jd.localAnchorA = (s2Vec2){0.0f, y};
		jd.localAnchorB = s2Vec2_zero;
		jd.maxMotorTorque = 200.0f;
		jd.motorSpeed = 5.0f * sign;
		jd.enableMotor = true;
		jd.drawSize = 0.2f;
		s2CreateRevoluteJoint(w, &jd);
		y -= 14.0f;
		sign = -sign;
	}
	if (st != NULL)
	{
		st->kind = 3;
		st->wait = 0.5f;
		st->side = -15.0f;
		stressCreateElement(w, st);
	}
}

// "Stretched Chain" (sample_joints.cpp:529-600): forty bodies created 2 m apart and jointed with anchors 1 m apart -- every joint
// starts a metre open, the whole error is the solver's to remove
static void sceneStretchedChain(s2WorldId w, int count)
{
	s2BodyId ground = s2CreateBody(w, &s2_defaultBodyDef);
	const float length = 1.0f, base = length * (float)count;
	s2ShapeDef sd = s2_defaultShapeDef;
	sd.filter.maskBits = 0;
	s2Circle circle = {{0.0f, 0.0f}, 0.2f};
	s2RevoluteJointDef jd = s2DefaultRevoluteJointDef();
	jd.drawSize = 0.2f;
	jd.bodyIdA = ground;
	jd.localAnchorA.y = base - 0.5f * length;
	jd.localAnchorB.y = 0.5f * length;
	float y = base - 2.0f * length;
	for (int i = 0; i < count; ++i)
	{
		s2BodyId id = makeDynamic(w, 0.0f, y, 0.0f);
		s2CreateCircleShape(id, &sd, &circle);
		jd.bodyIdB = id;
		s2CreateRevoluteJoint(w, &jd);
		jd.bodyIdA = id;
		jd.localAnchorA.y = -0.5f * length;
		y -= 2.0f * length;
	}
}

// The part of a sample's Step override that runs BEFORE Sample::Step (samples/sample.cpp:126-137).  stepIndex: steps taken so far.
S2SCENE_API void s2scene_pre_step(s2WorldId w, int stepIndex, float timeStep)
{
	SceneState* st = stateOf(w, 0);
	if (st == NULL)
	{
		return;
	}
	if (st->kind == 1 && stepIndex == 120 && st->top.index != s2_nullBodyId.index)
	{
		s2DestroyBody(st->top); // sample_contact.cpp:101-110
		st->top = s2_nullBodyId;
	}
	else if (st->kind == 2 && timeStep > 0.0f)
	{
		const float force = 1000.0f; // sample_contact.cpp:632-650
		const int count = (int)st->wait;
		for (int i = 0; i < count; ++i)
		{
			s2Vec2 p = s2Body_GetPosition(st->rush[i]);
			float distance = s2Length(p);
			if (distance < 0.1f)
			{
				continue;
			}
			float scale = force / distance;
			s2Body_ApplyForceToCenter(st->rush[i], (s2Vec2){-scale * p.x, -scale * p.y});
		}
	}
}

// ... and the part that runs AFTER it: Ragdoll Stress takes out the ragdolls that fell through and drops the next one
// (sample_joints.cpp:320-349)
S2SCENE_API void s2scene_post_step(s2WorldId w, float hertz)
{
	SceneState* st = stateOf(w, 0);
	if (st == NULL || st->kind != 3)
	{
		return;
	}
	for (int i = 0; i < STRESS_HUMANS; ++i)
	{
		HumanIds* h = st->humans + i;
		if (!h->spawned)
		{
			continue;
		}
		s2Vec2 p = s2Body_GetPosition(h->bones[1]); // Bone::e_torso
		if (p.y < -25.0f)
		{
			for (int b = 0; b < HUMAN_BONES; ++b) // Human::Despawn, samples/collection/human.cpp:350-377
			{
				if (h->joints[b].index != s2_nullJointId.index)
				{
					s2DestroyJoint(h->joints[b]);
					h->joints[b] = s2_nullJointId;
				}
			}
			for (int b = 0; b < HUMAN_BONES; ++b)
			{
				s2DestroyBody(h->bones[b]);
				h->bones[b] = s2_nullBodyId;
			}
			h->spawned = 0;
		}
	}
	if (hertz > 0.0f)
	{
		st->wait -= 1.0f / hertz;
		if (st->wait < 0.0f)
		{
			stressCreateElement(w, st);
			st->wait += 0.5f;
		}
	}
}

// Returns the new world (null id on unknown scene / no free world slot).
S2SCENE_API s2WorldId s2scene_create(const char* name, int solverType, int p0, int p1)
{
	s2WorldDef def = s2DefaultWorldDef();
	def.solverType = (s2SolverType)solverType;
	s2WorldId w = s2CreateWorld(&def);
	if (w.index == s2_nullWorldId.index)
	{
		return w;
	}
	(void)stateOf(w, 1); // (whatever sample lived in this world slot before has left)

	if (strcmp(name, "pyramid") == 0)
	{
		scenePyramid(w, p0 > 0 ? p0 : 10);
	}
	else if (strcmp(name, "multi_pyramid") == 0)
	{
		sceneMultiPyramid(w, p0 > 0 ? p0 : 4, p1 > 0 ? p1 : 10);
	}
	else if (strcmp(name, "joint_grid") == 0)
	{
		int n = p0 > 0 ? p0 : 10;
		sceneJointGrid(w, n, p1 > 0 ? p1 : n);
	}
	else if (strcmp(name, "tumbler") == 0)
	{
		sceneTumbler(w, p0 > 0 ? p0 : 100);
	}
	else if (strcmp(name, "mixed") == 0)
	{
		sceneMixed(w, p0 > 0 ? p0 : 24);
	}
	else if (strcmp(name, "vertical_stack") == 0)
	{
		sceneVerticalStack(w, p0 > 0 ? p0 : 10);
	}
	else if (strcmp(name, "circle_pile") == 0)
	{
		sceneCirclePile(w, p0 > 0 ? p0 : 20);
	}
	else if (strcmp(name, "shapes_zoo") == 0)
	{
		sceneShapesZoo(w, p0 > 0 ? p0 : 40);
	}
	else if (strcmp(name, "arch") == 0)
	{
		sceneArch(w);
	}
	else if (strcmp(name, "high_mass_ratio") == 0) // p0 = 1, 2, 3: the three samples
	{
		sceneHighMassRatio(w, p0 > 0 ? p0 : 1);
	}
	else if (strcmp(name, "overlap_recovery") == 0)
	{
		sceneOverlapRecovery(w, 0.0f, 0.0f);
	}
	else if (strcmp(name, "card_house") == 0)
	{
		sceneCardHouse(w);
	}
	else if (strcmp(name, "far_pyramid") == 0)
	{
		sceneFarPyramid(w, 100000.0f, -80000.0f);
	}
	else if (strcmp(name, "far_stack") == 0)
	{
		sceneFarStack(w, 40000.0f, -25000.0f);
	}
	else if (strcmp(name, "far_recovery") == 0)
	{
		sceneOverlapRecovery(w, 80000.0f, -70000.0f);
	}
	else if (strcmp(name, "far_ragdoll_pile") == 0)
	{
		sceneFarRagdollPile(w, 6000.0f, -1500.0f);
	}
	else if (strcmp(name, "far_chain") == 0)
	{
		sceneChain(w, 40000.0f, -35000.0f, 40, 0.1f, 0.025f, 0.0f);
	}
	else if (strcmp(name, "ragdoll") == 0)
	{
		sceneRagdoll(w);
	}
	else if (strcmp(name, "ball_and_chain") == 0)
	{
		sceneChain(w, 0.0f, 0.0f, p0 > 0 ? p0 : 40, 0.5f, 0.125f, 8.0f);
	}
	else if (strcmp(name, "bridge") == 0)
	{
		sceneBridge(w, p0 > 0 ? p0 : 160);
	}
	else if (strcmp(name, "single_box") == 0)
	{
		sceneSingleBox(w);
	}
	else if (strcmp(name, "warm_start_energy") == 0)
	{
		sceneWarmStartEnergy(w);
	}
	else if (strcmp(name, "friction_ramp") == 0)
	{
		sceneFrictionRamp(w);
	}
	else if (strcmp(name, "rush") == 0)
	{
		sceneRush(w, p0 > 0 ? p0 : RUSH_COUNT);
	}
	else if (strcmp(name, "double_domino") == 0)
	{
		sceneDoubleDomino(w);
	}
	else if (strcmp(name, "confined") == 0)
	{
		sceneConfined(w, p0 > 0 ? p0 : 25);
	}
	else if (strcmp(name, "circle_stack") == 0)
	{
		sceneCircleStack(w, p0 > 0 ? p0 : 10);
	}
	else if (strcmp(name, "ragdoll_stress") == 0)
	{
		sceneRagdollStress(w);
	}
	else if (strcmp(name, "stretched_chain") == 0)
	{
		sceneStretchedChain(w, p0 > 0 ? p0 : 40);
	}
	else
	{
		s2DestroyWorld(w);
		return s2_nullWorldId;
	}
	return w;
}
