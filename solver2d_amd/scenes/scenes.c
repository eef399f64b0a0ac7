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

static const SceneEntry s_scenes[] = {{"pyramid"},		   {"multi_pyramid"}, {"joint_grid"}, {"tumbler"},
									  {"mixed"},		   {"vertical_stack"}, {"circle_pile"}};

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
	else
	{
		s2DestroyWorld(w);
		return s2_nullWorldId;
	}
	return w;
}
