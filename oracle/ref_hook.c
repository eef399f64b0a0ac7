// TEST INFRASTRUCTURE -- not part of the shipped library.
//
// Hooks linked into oracle/_ref/libs2ref.so together with the UNMODIFIED reference sources
// (compiled where they lie under /root/reference by oracle/Makefile).  The linker option
// --wrap=s2Solve_<Variant> reroutes the ten calls in s2World_Step's switch
// (src/world.c:206-256) through hookSolve() below, which can
//   mode 0: pass straight through to the reference solver,
//   mode 1: capture the solver's inputs and outputs in the s2amd wire format
//           (include/solver2d_amd.h) -- this is how golden fixtures are generated and how the
//           oracle restatement is pinned bit-for-bit against the reference,
//   mode 2: replace the reference solver by a callback with the s2amd_solve signature -- the
//           reference's own broad phase / narrow phase / contact bookkeeping then drives the
//           HIP solver, which is the literal drop-in test.  s2ref_use_amd() installs such a callback in C: it loads
//           libs2amd.so and forwards to s2amd_solve, so that any program calling the PUBLIC s2World_Step of this
//           library (the samples' only entry point) runs on the GPU -- the binding of INTEGRATION.md, working.
//   whole step (s2ref_use_amd_world): this file's s2World_Step -- the library's exported one; oracle/Makefile renames
//           the reference's in the compiled world.o -- keeps stages 1 and 2 (trees, contact pool) and runs stage 3, the
//           solve and stage 4 as one s2amd_world_step on the resident world chain; optionally the pair query too.
// It reads the reference's internal structs through the reference's own headers; no reference
// source is copied into this repository.

#include "body.h"
#include "contact.h"
#include "core.h"
#include "joint.h"
#include "shape.h"
#include "solvers.h"
#include "stack_allocator.h"
#include "world.h"

#include "solver2d/solver2d.h"

#include "solver2d_amd.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define S2REF_API __attribute__((visibility("default")))

typedef void s2SolveFcn(s2World* world, s2StepContext* context);
typedef int s2refReplaceFcn(void* user, const s2amdStepParams* params, s2amdBody* bodies, int32_t bodyCapacity,
							s2amdContact* contacts, int32_t contactCapacity, s2amdJoint* joints, int32_t jointCapacity);

typedef struct Snapshot
{
	s2amdBody* bodies;
	s2amdContact* contacts;
	s2amdJoint* joints;
	int32_t bodyCapacity, contactCapacity, jointCapacity;
} Snapshot;

static int g_mode = 0;
static s2refReplaceFcn* g_replace = NULL;
static void* g_replaceUser = NULL;
static Snapshot g_pre = {0}, g_post = {0};
static s2amdStepParams g_params;
static int g_captureCount = 0;
static int g_replaceError = 0;
static double g_solveSeconds = 0.0;
static long g_solveCalls = 0;

static void packBodies(const s2World* world, s2amdBody* out)
{
	int n = world->bodyPool.capacity;
	for (int i = 0; i < n; ++i)
	{
		const s2Body* b = world->bodies + i;
		s2amdBody* o = out + i;
		memset(o, 0, sizeof(*o));
		if (s2IsFree(&b->object))
		{
			o->type = S2AMD_BODY_FREE;
			continue;
		}
		o->position[0] = b->position.x, o->position[1] = b->position.y;
		o->rot[0] = b->rot.s, o->rot[1] = b->rot.c;
		o->linearVelocity[0] = b->linearVelocity.x, o->linearVelocity[1] = b->linearVelocity.y;
		o->angularVelocity = b->angularVelocity;
		o->deltaPosition[0] = b->deltaPosition.x, o->deltaPosition[1] = b->deltaPosition.y;
		o->localCenter[0] = b->localCenter.x, o->localCenter[1] = b->localCenter.y;
		o->force[0] = b->force.x, o->force[1] = b->force.y;
		o->torque = b->torque;
		o->mass = b->mass, o->invMass = b->invMass;
		o->I = b->I, o->invI = b->invI;
		o->linearDamping = b->linearDamping;
		o->angularDamping = b->angularDamping;
		o->gravityScale = b->gravityScale;
		o->type = (int32_t)b->type;
	}
}

static void unpackBodies(s2World* world, const s2amdBody* in)
{
	int n = world->bodyPool.capacity;
	for (int i = 0; i < n; ++i)
	{
		s2Body* b = world->bodies + i;
		const s2amdBody* o = in + i;
		if (s2IsFree(&b->object))
		{
			continue;
		}
		b->position = (s2Vec2){o->position[0], o->position[1]};
		b->rot = (s2Rot){o->rot[0], o->rot[1]};
		b->linearVelocity = (s2Vec2){o->linearVelocity[0], o->linearVelocity[1]};
		b->angularVelocity = o->angularVelocity;
		b->deltaPosition = (s2Vec2){o->deltaPosition[0], o->deltaPosition[1]};
	}
}

static void packContacts(const s2World* world, s2amdContact* out)
{
	int n = world->contactPool.capacity;
	for (int i = 0; i < n; ++i)
	{
		const s2Contact* c = world->contacts + i;
		s2amdContact* o = out + i;
		memset(o, 0, sizeof(*o));
		o->constraintIndex = -1;
		if (s2IsFree(&c->object))
		{
			o->bodyA = -1, o->bodyB = -1;
			continue;
		}
		const s2Manifold* m = &c->manifold;
		o->bodyA = c->edges[0].bodyIndex;
		o->bodyB = c->edges[1].bodyIndex;
		o->pointCount = m->pointCount;
		o->frictionPersisted = m->frictionPersisted ? 1 : 0;
		o->normal[0] = m->normal.x, o->normal[1] = m->normal.y;
		o->friction = c->friction;
		o->constraintIndex = m->constraintIndex;
		for (int j = 0; j < 2; ++j)
		{
			const s2ManifoldPoint* p = m->points + j;
			s2amdManifoldPoint* q = o->points + j;
			q->localAnchorA[0] = p->localAnchorA.x, q->localAnchorA[1] = p->localAnchorA.y;
			q->localAnchorB[0] = p->localAnchorB.x, q->localAnchorB[1] = p->localAnchorB.y;
			q->frictionAnchorA[0] = p->frictionAnchorA.x, q->frictionAnchorA[1] = p->frictionAnchorA.y;
			q->frictionAnchorB[0] = p->frictionAnchorB.x, q->frictionAnchorB[1] = p->frictionAnchorB.y;
			q->frictionNormalA[0] = p->frictionNormalA.x, q->frictionNormalA[1] = p->frictionNormalA.y;
			q->frictionNormalB[0] = p->frictionNormalB.x, q->frictionNormalB[1] = p->frictionNormalB.y;
			q->separation = p->separation;
			q->normalImpulse = p->normalImpulse;
			q->tangentImpulse = p->tangentImpulse;
		}
	}
}

static void unpackContacts(s2World* world, const s2amdContact* in)
{
	int n = world->contactPool.capacity;
	for (int i = 0; i < n; ++i)
	{
		s2Contact* c = world->contacts + i;
		const s2amdContact* o = in + i;
		if (s2IsFree(&c->object))
		{
			continue;
		}
		s2Manifold* m = &c->manifold;
		m->frictionPersisted = o->frictionPersisted != 0;
		if (o->constraintIndex >= 0)
		{
			m->constraintIndex = o->constraintIndex;
		}
		for (int j = 0; j < 2; ++j)
		{
			s2ManifoldPoint* p = m->points + j;
			const s2amdManifoldPoint* q = o->points + j;
			p->frictionAnchorA = (s2Vec2){q->frictionAnchorA[0], q->frictionAnchorA[1]};
			p->frictionAnchorB = (s2Vec2){q->frictionAnchorB[0], q->frictionAnchorB[1]};
			p->frictionNormalA = (s2Vec2){q->frictionNormalA[0], q->frictionNormalA[1]};
			p->frictionNormalB = (s2Vec2){q->frictionNormalB[0], q->frictionNormalB[1]};
			p->normalImpulse = q->normalImpulse;
			p->tangentImpulse = q->tangentImpulse;
		}
	}
}

static void packJoints(const s2World* world, s2amdJoint* out)
{
	int n = world->jointPool.capacity;
	for (int i = 0; i < n; ++i)
	{
		const s2Joint* jn = world->joints + i;
		s2amdJoint* o = out + i;
		memset(o, 0, sizeof(*o));
		if (s2IsFree(&jn->object))
		{
			o->type = S2AMD_JOINT_FREE;
			o->bodyA = -1, o->bodyB = -1;
			continue;
		}
		o->bodyA = jn->edges[0].bodyIndex;
		o->bodyB = jn->edges[1].bodyIndex;
		o->localOriginAnchorA[0] = jn->localOriginAnchorA.x, o->localOriginAnchorA[1] = jn->localOriginAnchorA.y;
		o->localOriginAnchorB[0] = jn->localOriginAnchorB.x, o->localOriginAnchorB[1] = jn->localOriginAnchorB.y;
		if (jn->type == s2_revoluteJoint)
		{
			const s2RevoluteJoint* r = &jn->revoluteJoint;
			o->type = S2AMD_JOINT_REVOLUTE;
			o->enableMotor = r->enableMotor ? 1 : 0;
			o->enableLimit = r->enableLimit ? 1 : 0;
			o->impulse[0] = r->impulse.x, o->impulse[1] = r->impulse.y;
			o->motorImpulse = r->motorImpulse;
			o->lowerImpulse = r->lowerImpulse;
			o->upperImpulse = r->upperImpulse;
			o->maxMotorTorque = r->maxMotorTorque;
			o->motorSpeed = r->motorSpeed;
			o->referenceAngle = r->referenceAngle;
			o->lowerAngle = r->lowerAngle;
			o->upperAngle = r->upperAngle;
		}
		else
		{
			const s2MouseJoint* mj = &jn->mouseJoint;
			o->type = S2AMD_JOINT_MOUSE;
			o->impulse[0] = mj->impulse.x, o->impulse[1] = mj->impulse.y;
			o->motorImpulse = mj->motorImpulse;
			o->hertz = mj->hertz;
			o->dampingRatio = mj->dampingRatio;
			o->targetA[0] = mj->targetA.x, o->targetA[1] = mj->targetA.y;
		}
	}
}

static void unpackJoints(s2World* world, const s2amdJoint* in)
{
	int n = world->jointPool.capacity;
	for (int i = 0; i < n; ++i)
	{
		s2Joint* jn = world->joints + i;
		const s2amdJoint* o = in + i;
		if (s2IsFree(&jn->object))
		{
			continue;
		}
		if (jn->type == s2_revoluteJoint)
		{
			s2RevoluteJoint* r = &jn->revoluteJoint;
			r->impulse = (s2Vec2){o->impulse[0], o->impulse[1]};
			r->motorImpulse = o->motorImpulse;
			r->lowerImpulse = o->lowerImpulse;
			r->upperImpulse = o->upperImpulse;
		}
		else
		{
			s2MouseJoint* mj = &jn->mouseJoint;
			mj->impulse = (s2Vec2){o->impulse[0], o->impulse[1]};
			mj->motorImpulse = o->motorImpulse;
		}
	}
}

static void snapshotWorld(const s2World* world, Snapshot* s)
{
	int nb = world->bodyPool.capacity, nc = world->contactPool.capacity, nj = world->jointPool.capacity;
	s->bodies = (s2amdBody*)realloc(s->bodies, (size_t)(nb > 0 ? nb : 1) * sizeof(s2amdBody));
	s->contacts = (s2amdContact*)realloc(s->contacts, (size_t)(nc > 0 ? nc : 1) * sizeof(s2amdContact));
	s->joints = (s2amdJoint*)realloc(s->joints, (size_t)(nj > 0 ? nj : 1) * sizeof(s2amdJoint));
	s->bodyCapacity = nb, s->contactCapacity = nc, s->jointCapacity = nj;
	packBodies(world, s->bodies);
	packContacts(world, s->contacts);
	packJoints(world, s->joints);
}

static void fillParams(const s2World* world, const s2StepContext* context, int solverType)
{
	g_params.solverType = solverType;
	g_params.dt = context->dt;
	g_params.velIters = context->iterations;
	g_params.posIters = context->extraIterations;
	g_params.warmStart = context->warmStart ? 1 : 0;
	g_params.gravity[0] = world->gravity.x;
	g_params.gravity[1] = world->gravity.y;
}

static void narrowPhaseDone(const s2World* world);

static void hookSolve(s2World* world, s2StepContext* context, int solverType, s2SolveFcn* real)
{
	narrowPhaseDone(world);
	if (g_mode == 1)
	{
		fillParams(world, context, solverType);
		snapshotWorld(world, &g_pre);
		real(world, context);
		snapshotWorld(world, &g_post);
		g_captureCount += 1;
	}
	else if (g_mode == 2 && g_replace != NULL)
	{
		fillParams(world, context, solverType);
		snapshotWorld(world, &g_pre);
		int rc = g_replace(g_replaceUser, &g_params, g_pre.bodies, g_pre.bodyCapacity, g_pre.contacts, g_pre.contactCapacity,
						   g_pre.joints, g_pre.jointCapacity);
		if (rc != 0)
		{
			g_replaceError = rc;
		}
		unpackBodies(world, g_pre.bodies);
		unpackContacts(world, g_pre.contacts);
		unpackJoints(world, g_pre.joints);
		g_captureCount += 1;
	}
	else if (g_mode == 3)
	{
		// mode 3: time the reference solver alone (CPU baseline of bench.py)
		struct timespec t0, t1;
		clock_gettime(CLOCK_MONOTONIC, &t0);
		real(world, context);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		g_solveSeconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
		g_solveCalls += 1;
	}
	else
	{
		real(world, context);
	}
}

#define WRAP(NAME, TYPE)                                                                                                         \
	void __real_##NAME(s2World* world, s2StepContext* context);                                                                  \
	void __wrap_##NAME(s2World* world, s2StepContext* context)                                                                   \
	{                                                                                                                            \
		hookSolve(world, context, TYPE, __real_##NAME);                                                                          \
	}

WRAP(s2Solve_Jacobi, s2_solverJacobi)
WRAP(s2Solve_PGS, s2_solverPGS)
WRAP(s2Solve_PGS_NGS, s2_solverPGS_NGS)
WRAP(s2Solve_PGS_NGS_Block, s2_solverPGS_NGS_Block)
WRAP(s2Solve_PGS_Soft, s2_solverPGS_Soft)
WRAP(s2Solve_SoftStep, s2_solverSoftStep)
WRAP(s2Solve_TGS_Sticky, s2_solverTGS_Sticky)
WRAP(s2Solve_TGS_Soft, s2_solverTGS_Soft)
WRAP(s2Solve_TGS_NGS, s2_solverTGS_NGS)
WRAP(s2Solve_XPBD, s2_solverXPBD)

// ---- shapes and the broad phase ----

static void packShapes(const s2World* world, s2amdShape* out)
{
	int n = world->shapePool.capacity;
	for (int i = 0; i < n; ++i)
	{
		const s2Shape* sh = world->shapes + i;
		s2amdShape* o = out + i;
		memset(o, 0, sizeof(*o));
		if (s2IsFree(&sh->object))
		{
			o->body = -1;
			o->type = S2AMD_SHAPE_FREE;
			continue;
		}
		o->body = sh->bodyIndex;
		o->type = (int32_t)sh->type;
		o->categoryBits = sh->filter.categoryBits;
		o->maskBits = sh->filter.maskBits;
		o->groupIndex = sh->filter.groupIndex;
		o->proxyKey = sh->proxyKey;
		o->enlarged = sh->enlargedAABB ? 1 : 0;
		o->aabb[0] = sh->aabb.lowerBound.x, o->aabb[1] = sh->aabb.lowerBound.y;
		o->aabb[2] = sh->aabb.upperBound.x, o->aabb[3] = sh->aabb.upperBound.y;
		o->fatAABB[0] = sh->fatAABB.lowerBound.x, o->fatAABB[1] = sh->fatAABB.lowerBound.y;
		o->fatAABB[2] = sh->fatAABB.upperBound.x, o->fatAABB[3] = sh->fatAABB.upperBound.y;
		switch (sh->type)
		{
			case s2_polygonShape:
				o->count = sh->polygon.count;
				o->radius = sh->polygon.radius;
				for (int v = 0; v < sh->polygon.count; ++v)
				{
					o->vertices[v][0] = sh->polygon.vertices[v].x, o->vertices[v][1] = sh->polygon.vertices[v].y;
					o->normals[v][0] = sh->polygon.normals[v].x, o->normals[v][1] = sh->polygon.normals[v].y;
				}
				break;
			case s2_circleShape:
				o->radius = sh->circle.radius;
				o->vertices[0][0] = sh->circle.point.x, o->vertices[0][1] = sh->circle.point.y;
				break;
			case s2_capsuleShape:
				o->radius = sh->capsule.radius;
				o->vertices[0][0] = sh->capsule.point1.x, o->vertices[0][1] = sh->capsule.point1.y;
				o->vertices[1][0] = sh->capsule.point2.x, o->vertices[1][1] = sh->capsule.point2.y;
				break;
			case s2_segmentShape:
				o->vertices[0][0] = sh->segment.point1.x, o->vertices[0][1] = sh->segment.point1.y;
				o->vertices[1][0] = sh->segment.point2.x, o->vertices[1][1] = sh->segment.point2.y;
				break;
			default:
				break;
		}
	}
}

// broad-phase capture: state at s2UpdateBroadPhasePairs entry and the pairs it created
static s2amdShape* g_bpShapes = NULL;
static uint8_t* g_bpMoved = NULL;
static int32_t* g_bpExisting = NULL;
static int32_t* g_bpNew = NULL;
static int g_bpShapeCount = 0, g_bpExistingCount = 0, g_bpNewCount = 0;
static double g_bpSeconds = 0.0;

// ---- narrow phase: state at the end of Stage 2 (the input of the "update contacts" loop,
// src/world.c:132-168) and at solver entry (its output) ----
static s2World* g_stepWorld = NULL;
static s2amdShape* g_npShapes = NULL;
static s2amdBody* g_npBodies = NULL;
static float* g_npOrigins = NULL;
static s2amdPairState *g_npPairsPre = NULL, *g_npPairsPost = NULL;
static s2amdContact *g_npContactsPre = NULL, *g_npContactsPost = NULL;
static int g_npShapeCount = 0, g_npBodyCount = 0, g_npContactCount = 0;
static double g_npSeconds = 0.0;
static struct timespec g_npStart;
static int g_npTiming = 0;

static void packPairs(const s2World* world, s2amdPairState* out)
{
	int n = world->contactPool.capacity;
	for (int i = 0; i < n; ++i)
	{
		const s2Contact* c = world->contacts + i;
		s2amdPairState* o = out + i;
		memset(o, 0, sizeof(*o));
		if (s2IsFree(&c->object))
		{
			o->shapeA = -1, o->shapeB = -1;
			continue;
		}
		o->shapeA = c->shapeIndexA;
		o->shapeB = c->shapeIndexB;
		o->cacheMetric = c->cache.metric;
		o->cacheCount = c->cache.count;
		for (int k = 0; k < 3; ++k)
		{
			o->cacheIndexA[k] = c->cache.indexA[k];
			o->cacheIndexB[k] = c->cache.indexB[k];
		}
		for (int j = 0; j < 2; ++j)
		{
			o->id[j] = c->manifold.points[j].id;
			o->persisted[j] = c->manifold.points[j].persisted ? 1 : 0;
		}
	}
}

void __real_s2BroadPhase_RebuildTrees(s2BroadPhase* bp);
void __wrap_s2BroadPhase_RebuildTrees(s2BroadPhase* bp)
{
	__real_s2BroadPhase_RebuildTrees(bp);
	s2World* world = g_stepWorld;
	if (world == NULL || &world->broadPhase != bp)
	{
		return;
	}
	if (g_mode == 3)
	{
		clock_gettime(CLOCK_MONOTONIC, &g_npStart);
		g_npTiming = 1;
		return;
	}
	if (g_mode != 1)
	{
		return;
	}
	int ns = world->shapePool.capacity, nb = world->bodyPool.capacity, nc = world->contactPool.capacity;
	g_npShapes = (s2amdShape*)realloc(g_npShapes, (size_t)(ns > 0 ? ns : 1) * sizeof(s2amdShape));
	g_npBodies = (s2amdBody*)realloc(g_npBodies, (size_t)(nb > 0 ? nb : 1) * sizeof(s2amdBody));
	g_npOrigins = (float*)realloc(g_npOrigins, (size_t)(nb > 0 ? nb : 1) * 2 * sizeof(float));
	g_npPairsPre = (s2amdPairState*)realloc(g_npPairsPre, (size_t)(nc > 0 ? nc : 1) * sizeof(s2amdPairState));
	g_npPairsPost = (s2amdPairState*)realloc(g_npPairsPost, (size_t)(nc > 0 ? nc : 1) * sizeof(s2amdPairState));
	g_npContactsPre = (s2amdContact*)realloc(g_npContactsPre, (size_t)(nc > 0 ? nc : 1) * sizeof(s2amdContact));
	g_npContactsPost = (s2amdContact*)realloc(g_npContactsPost, (size_t)(nc > 0 ? nc : 1) * sizeof(s2amdContact));
	packShapes(world, g_npShapes);
	packBodies(world, g_npBodies);
	for (int i = 0; i < nb; ++i)
	{
		g_npOrigins[2 * i] = world->bodies[i].origin.x;
		g_npOrigins[2 * i + 1] = world->bodies[i].origin.y;
	}
	packPairs(world, g_npPairsPre);
	packContacts(world, g_npContactsPre);
	g_npShapeCount = ns, g_npBodyCount = nb, g_npContactCount = nc;
}

// called at solver entry: Stage 3's output
static void narrowPhaseDone(const s2World* world)
{
	if (g_mode == 3 && g_npTiming)
	{
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		g_npSeconds += (double)(t1.tv_sec - g_npStart.tv_sec) + 1e-9 * (double)(t1.tv_nsec - g_npStart.tv_nsec);
		g_npTiming = 0;
	}
	if (g_mode == 1 && g_npContactCount == world->contactPool.capacity && g_npPairsPost != NULL)
	{
		packPairs(world, g_npPairsPost);
		packContacts(world, g_npContactsPost);
	}
}

S2REF_API double s2ref_narrowphase_seconds(int reset)
{
	double v = g_npSeconds;
	if (reset)
	{
		g_npSeconds = 0.0;
	}
	return v;
}

S2REF_API int s2ref_narrowphase_capture(const s2amdShape** shapes, int32_t* shapeCount, const s2amdBody** bodies, int32_t* bodyCount,
										const float** origins, const s2amdPairState** pairsPre, const s2amdContact** contactsPre,
										const s2amdPairState** pairsPost, const s2amdContact** contactsPost, int32_t* contactCount)
{
	*shapes = g_npShapes, *shapeCount = g_npShapeCount, *bodies = g_npBodies, *bodyCount = g_npBodyCount, *origins = g_npOrigins;
	*pairsPre = g_npPairsPre, *contactsPre = g_npContactsPre, *pairsPost = g_npPairsPost, *contactsPost = g_npContactsPost;
	*contactCount = g_npContactCount;
	return 0;
}

void __real_s2UpdateBroadPhasePairs(s2World* world);
void __wrap_s2UpdateBroadPhasePairs(s2World* world)
{
	g_stepWorld = world;
	if (g_mode == 3)
	{
		struct timespec t0, t1;
		clock_gettime(CLOCK_MONOTONIC, &t0);
		__real_s2UpdateBroadPhasePairs(world);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		g_bpSeconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
		return;
	}
	if (g_mode != 1)
	{
		__real_s2UpdateBroadPhasePairs(world);
		return;
	}
	int ns = world->shapePool.capacity, nc = world->contactPool.capacity;
	g_bpShapes = (s2amdShape*)realloc(g_bpShapes, (size_t)(ns > 0 ? ns : 1) * sizeof(s2amdShape));
	g_bpMoved = (uint8_t*)realloc(g_bpMoved, (size_t)(ns > 0 ? ns : 1));
	g_bpExisting = (int32_t*)realloc(g_bpExisting, (size_t)(nc > 0 ? nc : 1) * 2 * sizeof(int32_t));
	packShapes(world, g_bpShapes);
	memset(g_bpMoved, 0, (size_t)(ns > 0 ? ns : 1));
	g_bpShapeCount = ns;
	s2BroadPhase* bp = &world->broadPhase;
	int moveCount = s2Array(bp->moveArray).count;
	for (int i = 0; i < moveCount; ++i)
	{
		int key = bp->moveArray[i];
		if (key == S2_NULL_INDEX)
		{
			continue;
		}
		int shapeIndex = s2BroadPhase_GetShapeIndex(bp, key);
		if (0 <= shapeIndex && shapeIndex < ns)
		{
			g_bpMoved[shapeIndex] = 1;
		}
	}
	g_bpExistingCount = 0;
	uint8_t* had = (uint8_t*)calloc((size_t)(nc > 0 ? nc : 1), 1);
	for (int i = 0; i < nc; ++i)
	{
		const s2Contact* c = world->contacts + i;
		if (s2IsFree(&c->object))
		{
			continue;
		}
		had[i] = 1;
		g_bpExisting[2 * g_bpExistingCount] = c->shapeIndexA;
		g_bpExisting[2 * g_bpExistingCount + 1] = c->shapeIndexB;
		g_bpExistingCount += 1;
	}
	int oldCapacity = nc;
	__real_s2UpdateBroadPhasePairs(world);
	int nc2 = world->contactPool.capacity;
	g_bpNew = (int32_t*)realloc(g_bpNew, (size_t)(nc2 > 0 ? nc2 : 1) * 2 * sizeof(int32_t));
	g_bpNewCount = 0;
	for (int i = 0; i < nc2; ++i)
	{
		const s2Contact* c = world->contacts + i;
		if (s2IsFree(&c->object) || (i < oldCapacity && had[i]))
		{
			continue;
		}
		g_bpNew[2 * g_bpNewCount] = c->shapeIndexA;
		g_bpNew[2 * g_bpNewCount + 1] = c->shapeIndexB;
		g_bpNewCount += 1;
	}
	free(had);
}

S2REF_API double s2ref_broadphase_seconds(int reset)
{
	double v = g_bpSeconds;
	if (reset)
	{
		g_bpSeconds = 0.0;
	}
	return v;
}

S2REF_API int s2ref_shape_capacity(s2WorldId id)
{
	return s2GetWorldFromId(id)->shapePool.capacity;
}

S2REF_API int s2ref_pack_shapes(s2WorldId id, s2amdShape* shapes, float* origins)
{
	s2World* world = s2GetWorldFromId(id);
	if (shapes)
	{
		packShapes(world, shapes);
	}
	if (origins)
	{
		for (int i = 0; i < world->bodyPool.capacity; ++i)
		{
			origins[2 * i] = world->bodies[i].origin.x;
			origins[2 * i + 1] = world->bodies[i].origin.y;
		}
	}
	return 0;
}

S2REF_API int s2ref_broadphase_capture(const s2amdShape** shapes, int32_t* shapeCount, const uint8_t** moved, const int32_t** existing,
									   int32_t* existingCount, const int32_t** created, int32_t* createdCount)
{
	*shapes = g_bpShapes, *shapeCount = g_bpShapeCount, *moved = g_bpMoved;
	*existing = g_bpExisting, *existingCount = g_bpExistingCount;
	*created = g_bpNew, *createdCount = g_bpNewCount;
	return 0;
}

// ---- control surface used by tests/ and tools/ through ctypes ----

S2REF_API void s2ref_set_mode(int mode)
{
	g_mode = mode;
	g_replaceError = 0;
}

S2REF_API double s2ref_solve_seconds(int reset)
{
	double v = g_solveSeconds;
	if (reset)
	{
		g_solveSeconds = 0.0;
		g_solveCalls = 0;
	}
	return v;
}

S2REF_API long s2ref_solve_calls(void)
{
	return g_solveCalls;
}

S2REF_API void s2ref_set_replace(s2refReplaceFcn* fcn, void* user)
{
	g_replace = fcn;
	g_replaceUser = user;
}

// ---- the native shim: s2Solve_* -> s2amd_solve, through dlopen (no link-time dependency on the HIP runtime) ----
static void* g_amdLib = NULL;
static s2amdSolver* g_amdSolver = NULL;
static int (*g_amdSolve)(s2amdSolver*, const s2amdStepParams*, s2amdBody*, int32_t, s2amdContact*, int32_t, s2amdJoint*, int32_t) = NULL;
static void (*g_amdDestroy)(s2amdSolver*) = NULL;

static int amdReplace(void* user, const s2amdStepParams* params, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts,
					  int32_t contactCapacity, s2amdJoint* joints, int32_t jointCapacity)
{
	(void)user;
	return g_amdSolve(g_amdSolver, params, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
}

static void mirrorSyncIfResident(void);
// path == NULL: back to the reference's own solvers.  Returns 0, or a negative number naming the step that failed.
S2REF_API int s2ref_use_amd(const char* path, int device)
{
	mirrorSyncIfResident();
	if (g_amdSolver != NULL && g_amdDestroy != NULL)
	{
		g_amdDestroy(g_amdSolver);
	}
	g_amdSolver = NULL;
	if (g_replace == amdReplace)
	{
		g_replace = NULL;
		g_mode = 0;
	}
	if (path == NULL)
	{
		return 0;
	}
	if (g_amdLib == NULL)
	{
		g_amdLib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
	}
	if (g_amdLib == NULL)
	{
		return -1;
	}
	int (*create)(int, s2amdSolver**) = (int (*)(int, s2amdSolver**))dlsym(g_amdLib, "s2amd_create");
	g_amdSolve = (int (*)(s2amdSolver*, const s2amdStepParams*, s2amdBody*, int32_t, s2amdContact*, int32_t, s2amdJoint*, int32_t))dlsym(g_amdLib, "s2amd_solve");
	g_amdDestroy = (void (*)(s2amdSolver*))dlsym(g_amdLib, "s2amd_destroy");
	if (create == NULL || g_amdSolve == NULL || g_amdDestroy == NULL)
	{
		return -2;
	}
	if (create(device, &g_amdSolver) != 0 || g_amdSolver == NULL)
	{
		return -3;
	}
	g_replace = amdReplace;
	g_replaceUser = NULL;
	g_mode = 2;
	g_replaceError = 0;
	return 0;
}

// ---- the native shim, whole step: stage 3, s2Solve_* and stage 4 of s2World_Step on the resident world chain
// (s2amd_world_upload / _set_contacts / _step / _download, include/solver2d_amd.h), stages 1 and 2 -- the dynamic
// trees and the contact pool, the control plane -- stay the reference's own code.  Per step the host receives the
// bodies, the origins, the stage-3 status words and (when a fat AABB was re-inflated) the shapes; the manifolds stay
// in HBM until somebody asks (s2ref_pack_world, s2ref_world_sync, a re-upload, leaving the mode).
typedef int s2amdWorldUploadFcn(s2amdSolver*, const s2amdBody*, int32_t, const s2amdContact*, int32_t, const s2amdJoint*, int32_t, const s2amdShape*,
								int32_t, const s2amdPairState*, const float*);
typedef int s2amdWorldStepFcn(s2amdSolver*, const s2amdStepParams*, s2amdWorldStepInfo*);
typedef int s2amdWorldSetContactsFcn(s2amdSolver*, const int32_t*, int32_t, const s2amdContact*, const s2amdPairState*);
typedef int s2amdWorldDownloadFcn(s2amdSolver*, s2amdBody*, int32_t, s2amdContact*, int32_t, s2amdJoint*, int32_t, s2amdShape*, int32_t,
								  s2amdPairState*, float*, int32_t*);
static s2amdWorldUploadFcn* g_amdWorldUpload = NULL;
static s2amdWorldStepFcn* g_amdWorldStep = NULL;
static s2amdWorldSetContactsFcn* g_amdWorldSetContacts = NULL;
static s2amdWorldDownloadFcn* g_amdWorldDownload = NULL;
static int g_wholeStep = 0;
typedef int s2amdWorldFindPairsFcn(s2amdSolver*, int32_t*, int32_t, int32_t*);
static s2amdWorldFindPairsFcn* g_amdWorldFindPairs = NULL;
static int g_devicePairs = 0;
static int32_t* g_newPairs = NULL;
static int g_newPairCapacity = 0;

typedef struct WorldMirror
{
	s2World* world; // whose state is resident on the device (NULL: nobody's)
	uint64_t stepId; // s2World.stepId after the last step taken here: worlds live in a static array, a new world reuses the
	                 // address of a destroyed one (and starts at 0 again); steps taken elsewhere show up too
	int bodyCapacity, bodyCount, shapeCapacity, shapeCount, jointCapacity, jointCount, contactCapacity;
	int contactsStale; // the device holds newer manifolds / impulses / joint impulses than the host pools
	s2amdBody* bodies;
	s2amdContact* contacts;
	s2amdJoint* joints;
	s2amdShape* shapes;
	s2amdPairState* pairs;
	float* origins;
	int32_t* status;
	int64_t* liveKey; // shapeIndexA << 32 | shapeIndexB of the slot as the device knows it, -1: free there
	int32_t* slots;
	s2amdContact* slotContacts;
	s2amdPairState* slotPairs;
	long uploads, steps;
} WorldMirror;
static WorldMirror g_mirror = {0};
static double g_wholeMs[6] = {0}; // stage 1+2, sync in, device step, download, apply, steps

static double wallMs(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

static void* growTo(void* p, size_t count, size_t size)
{
	return realloc(p, (count > 0 ? count : 1) * size);
}

static void unpackContactsWhole(s2World* world, const s2amdContact* in, const s2amdPairState* pairs, int count)
{
	int n = world->contactPool.capacity < count ? world->contactPool.capacity : count;
	for (int i = 0; i < n; ++i)
	{
		s2Contact* c = world->contacts + i;
		if (s2IsFree(&c->object) || pairs[i].shapeA != c->shapeIndexA || pairs[i].shapeB != c->shapeIndexB)
		{
			continue; // not the contact the device knows in this slot
		}
		const s2amdContact* o = in + i;
		const s2amdPairState* ps = pairs + i;
		s2Manifold* m = &c->manifold;
		m->pointCount = o->pointCount;
		m->frictionPersisted = o->frictionPersisted != 0;
		m->normal = (s2Vec2){o->normal[0], o->normal[1]};
		if (o->constraintIndex >= 0)
		{
			m->constraintIndex = o->constraintIndex; // (the wire's -1: not in a constraint array this step; the reference keeps the stale index)
		}
		for (int j = 0; j < 2; ++j)
		{
			s2ManifoldPoint* p = m->points + j;
			const s2amdManifoldPoint* q = o->points + j;
			p->localAnchorA = (s2Vec2){q->localAnchorA[0], q->localAnchorA[1]};
			p->localAnchorB = (s2Vec2){q->localAnchorB[0], q->localAnchorB[1]};
			p->frictionAnchorA = (s2Vec2){q->frictionAnchorA[0], q->frictionAnchorA[1]};
			p->frictionAnchorB = (s2Vec2){q->frictionAnchorB[0], q->frictionAnchorB[1]};
			p->frictionNormalA = (s2Vec2){q->frictionNormalA[0], q->frictionNormalA[1]};
			p->frictionNormalB = (s2Vec2){q->frictionNormalB[0], q->frictionNormalB[1]};
			p->separation = q->separation;
			p->normalImpulse = q->normalImpulse;
			p->tangentImpulse = q->tangentImpulse;
			p->id = ps->id[j];
			p->persisted = ps->persisted[j] != 0;
		}
		c->cache.metric = ps->cacheMetric;
		c->cache.count = ps->cacheCount;
		for (int k = 0; k < 3; ++k)
		{
			c->cache.indexA[k] = ps->cacheIndexA[k];
			c->cache.indexB[k] = ps->cacheIndexB[k];
		}
	}
}

// manifolds, GJK caches and joint impulses back into the reference's pools
static int mirrorSync(void)
{
	WorldMirror* m = &g_mirror;
	const uint64_t id = m->world ? (uint64_t)m->world->stepId : 0;
	if (m->world == NULL || (id != m->stepId && id != m->stepId + 1) || !m->contactsStale || g_amdSolver == NULL)
	{
		return 0;
	}
	int rc = g_amdWorldDownload(g_amdSolver, NULL, m->bodyCapacity, m->contacts, m->contactCapacity, m->joints, m->jointCapacity, NULL,
								m->shapeCapacity, m->pairs, NULL, NULL);
	if (rc != 0)
	{
		return rc;
	}
	unpackContactsWhole(m->world, m->contacts, m->pairs, m->contactCapacity);
	if (m->world->jointPool.capacity == m->jointCapacity && m->world->jointPool.count == m->jointCount)
	{
		unpackJoints(m->world, m->joints); // (joints were created or destroyed: their impulses start over)
	}
	m->contactsStale = 0;
	return 0;
}

static int mirrorMatches(const s2World* w)
{
	const WorldMirror* m = &g_mirror;
	return m->world == w && m->stepId + 1 == (uint64_t)w->stepId && m->bodyCapacity == w->bodyPool.capacity && m->bodyCount == w->bodyPool.count &&
		   m->shapeCapacity == w->shapePool.capacity && m->shapeCount == w->shapePool.count && m->jointCapacity == w->jointPool.capacity &&
		   m->jointCount == w->jointPool.count && m->contactCapacity == w->contactPool.capacity;
}

static int mirrorUpload(s2World* w)
{
	WorldMirror* m = &g_mirror;
	int rc = 0;
	if (m->world == w && (rc = mirrorSync()) != 0) // the pools changed under a resident world: its manifolds first
	{
		return rc;
	}
	int nb = w->bodyPool.capacity, ns = w->shapePool.capacity, nj = w->jointPool.capacity, nc = w->contactPool.capacity;
	m->bodies = (s2amdBody*)growTo(m->bodies, (size_t)nb, sizeof(s2amdBody));
	m->origins = (float*)growTo(m->origins, (size_t)nb * 2, sizeof(float));
	m->shapes = (s2amdShape*)growTo(m->shapes, (size_t)ns, sizeof(s2amdShape));
	m->joints = (s2amdJoint*)growTo(m->joints, (size_t)nj, sizeof(s2amdJoint));
	m->contacts = (s2amdContact*)growTo(m->contacts, (size_t)nc, sizeof(s2amdContact));
	m->pairs = (s2amdPairState*)growTo(m->pairs, (size_t)nc, sizeof(s2amdPairState));
	m->status = (int32_t*)growTo(m->status, (size_t)nc, sizeof(int32_t));
	m->liveKey = (int64_t*)growTo(m->liveKey, (size_t)nc, sizeof(int64_t));
	m->slots = (int32_t*)growTo(m->slots, (size_t)nc, sizeof(int32_t));
	m->slotContacts = (s2amdContact*)growTo(m->slotContacts, (size_t)nc, sizeof(s2amdContact));
	m->slotPairs = (s2amdPairState*)growTo(m->slotPairs, (size_t)nc, sizeof(s2amdPairState));
	packBodies(w, m->bodies);
	packShapes(w, m->shapes);
	packJoints(w, m->joints);
	packContacts(w, m->contacts);
	packPairs(w, m->pairs);
	for (int i = 0; i < nb; ++i)
	{
		m->origins[2 * i] = w->bodies[i].origin.x;
		m->origins[2 * i + 1] = w->bodies[i].origin.y;
	}
	for (int i = 0; i < nc; ++i)
	{
		m->liveKey[i] = m->pairs[i].shapeA < 0 ? -1 : ((int64_t)m->pairs[i].shapeA << 32) | (int64_t)m->pairs[i].shapeB;
	}
	rc = g_amdWorldUpload(g_amdSolver, m->bodies, nb, m->contacts, nc, m->joints, nj, m->shapes, ns, m->pairs, m->origins);
	if (rc != 0)
	{
		m->world = NULL;
		return rc;
	}
	m->world = w;
	m->bodyCapacity = nb, m->bodyCount = w->bodyPool.count;
	m->shapeCapacity = ns, m->shapeCount = w->shapePool.count;
	m->jointCapacity = nj, m->jointCount = w->jointPool.count;
	m->contactCapacity = nc;
	m->contactsStale = 0;
	m->uploads += 1;
	return 0;
}

// contacts stage 1 created since the device last saw the pool (the pool never frees a slot on its own between steps:
// stage 3's separations are applied to both sides below)
static int mirrorSendNewContacts(s2World* w)
{
	WorldMirror* m = &g_mirror;
	int count = 0;
	for (int i = 0; i < m->contactCapacity; ++i)
	{
		const s2Contact* c = w->contacts + i;
		int64_t live = s2IsFree(&c->object) ? -1 : ((int64_t)c->shapeIndexA << 32) | (int64_t)c->shapeIndexB;
		if (live >= 0 && m->liveKey[i] < 0)
		{
			s2amdContact* o = m->slotContacts + count;
			s2amdPairState* ps = m->slotPairs + count;
			memset(o, 0, sizeof(*o));
			memset(ps, 0, sizeof(*ps));
			o->bodyA = c->edges[0].bodyIndex;
			o->bodyB = c->edges[1].bodyIndex;
			o->friction = c->friction;
			o->constraintIndex = -1;
			ps->shapeA = c->shapeIndexA;
			ps->shapeB = c->shapeIndexB;
			m->slots[count++] = i;
			m->liveKey[i] = live;
		}
		else if (live != m->liveKey[i])
		{
			return 1; // somebody destroyed a contact behind our back (s2DestroyBody, s2CreateJoint ...): upload again
		}
	}
	return count > 0 ? g_amdWorldSetContacts(g_amdSolver, m->slots, count, m->slotContacts, m->slotPairs) : 0;
}

void __real_s2UpdateBroadPhasePairs(s2World* world);
// The reference's own s2World_Step: oracle/Makefile renames the symbol in the compiled world.o (objcopy, the source is
// untouched) so that THIS library's exported s2World_Step is the function below -- programs linked against the
// public API reach it without knowing.
void s2ref_World_Step_reference(s2WorldId worldId, float timeStep, int velIters, int posIters, bool warmStart);
S2REF_API void s2World_Step(s2WorldId worldId, float timeStep, int velIters, int posIters, bool warmStart)
{
	if (!g_wholeStep || g_amdSolver == NULL)
	{
		s2ref_World_Step_reference(worldId, timeStep, velIters, posIters, warmStart);
		return;
	}
	s2World* world = s2GetWorldFromId(worldId);
	WorldMirror* m = &g_mirror;
	world->stepId += 1;
	const double t0 = wallMs();
	int rc = 0;
	if (g_devicePairs && mirrorMatches(world))
	{
		// stage 1 with the pair discovery on the device (s2amd_world_find_pairs on the boxes the last refit re-inflated
		// -- the proxies in the reference's move buffer); the pool bookkeeping of each new pair is s2CreateContact as ever.
		// The pairs arrive sorted, not in the reference's tree-traversal order: contacts get other pool slots than with
		// the host's stage 1.  The trees keep following the fat boxes (below) for the reference's ray casts and queries.
		s2BroadPhase* bp = &world->broadPhase;
		if (s2Array(bp->moveArray).count > 0)
		{
			int32_t count = 0;
			rc = g_amdWorldFindPairs(g_amdSolver, g_newPairs, g_newPairCapacity, &count);
			if (rc == S2AMD_E_CAPACITY)
			{
				g_newPairCapacity = count + 1024;
				g_newPairs = (int32_t*)realloc(g_newPairs, (size_t)g_newPairCapacity * 2 * sizeof(int32_t));
				rc = g_amdWorldFindPairs(g_amdSolver, g_newPairs, g_newPairCapacity, &count);
			}
			for (int i = 0; rc == 0 && i < count; ++i)
			{
				s2CreateContact(world, world->shapes + g_newPairs[2 * i], world->shapes + g_newPairs[2 * i + 1]);
			}
			s2Array_Clear(bp->moveArray);
			s2ClearSet(&bp->moveSet);
		}
		if ((world->stepId & 63) == 0)
		{
			s2BroadPhase_RebuildTrees(bp); // stage 2 now and then: nobody queries the trees here, but the host's ray casts do
		}
	}
	else
	{
		// stages 1 and 2 (src/world.c:125-130): the reference's trees, the reference's contact pool
		__real_s2UpdateBroadPhasePairs(world);
		s2BroadPhase_RebuildTrees(&world->broadPhase);
	}
	const double t1 = wallMs();

	if (rc == 0 && (!mirrorMatches(world) || (rc = mirrorSendNewContacts(world)) == 1))
	{
		rc = mirrorUpload(world);
	}
	s2amdStepParams params;
	params.solverType = (int32_t)world->solverType;
	params.dt = timeStep;
	params.velIters = velIters;
	params.posIters = posIters;
	params.warmStart = warmStart ? 1 : 0;
	params.gravity[0] = world->gravity.x;
	params.gravity[1] = world->gravity.y;
	s2amdWorldStepInfo info = {0};
	const double t2 = wallMs();
	if (rc == 0)
	{
		rc = g_amdWorldStep(g_amdSolver, &params, &info);
	}
	const double t3 = wallMs();
	if (rc == 0)
	{
		rc = g_amdWorldDownload(g_amdSolver, m->bodies, m->bodyCapacity, NULL, m->contactCapacity, NULL, m->jointCapacity,
								info.movedCount > 0 ? m->shapes : NULL, m->shapeCapacity, NULL, m->origins,
								info.separatedCount > 0 ? m->status : NULL);
	}
	if (rc != 0)
	{
		const char* (*lastError)(void) = (const char* (*)(void))dlsym(g_amdLib, "s2amd_last_error");
		fprintf(stderr, "s2World_Step on the GPU failed (%d): %s\n", rc, lastError ? lastError() : "?");
		g_replaceError = rc;
		m->world = NULL;
		return;
	}
	const double t4 = wallMs();
	m->contactsStale = 1;
	m->steps += 1;
	m->stepId = (uint64_t)world->stepId;
	unpackBodies(world, m->bodies);
	for (int i = 0; i < m->bodyCapacity; ++i)
	{
		s2Body* b = world->bodies + i;
		if (s2IsFree(&b->object) || b->type == s2_staticBody)
		{
			continue;
		}
		b->origin = (s2Vec2){m->origins[2 * i], m->origins[2 * i + 1]};
		b->force = s2Vec2_zero;
		b->torque = 0.0f;
	}
	if (info.separatedCount > 0)
	{
		// src/world.c:163-167
		for (int i = 0; i < m->contactCapacity; ++i)
		{
			if (m->status[i] == S2AMD_PAIR_SEPARATED)
			{
				s2DestroyContact(world, world->contacts + i);
				m->liveKey[i] = -1;
			}
		}
	}
	if (info.movedCount > 0)
	{
		// src/world.c:259-297: the tight boxes of every shape, the tree only where the fat box was re-inflated -- in the
		// reference's order (bodies, then each body's shape list): the move buffer's order decides the pool slots of the
		// contacts stage 1 creates next step
		for (int b = 0; b < m->bodyCapacity; ++b)
		{
			const s2Body* body = world->bodies + b;
			if (s2IsFree(&body->object) || body->type == s2_staticBody)
			{
				continue;
			}
			for (int i = body->shapeList; i != S2_NULL_INDEX; i = world->shapes[i].nextShapeIndex)
			{
				s2Shape* sh = world->shapes + i;
				const s2amdShape* o = m->shapes + i;
				sh->aabb = (s2Box){{o->aabb[0], o->aabb[1]}, {o->aabb[2], o->aabb[3]}};
				if (o->enlarged)
				{
					sh->fatAABB = (s2Box){{o->fatAABB[0], o->fatAABB[1]}, {o->fatAABB[2], o->fatAABB[3]}};
					s2BroadPhase_EnlargeProxy(&world->broadPhase, sh->proxyKey, sh->fatAABB);
				}
			}
		}
	}
	s2GrowStack(world->stackAllocator);
	const double t5 = wallMs();
	g_wholeMs[0] += t1 - t0, g_wholeMs[1] += t2 - t1, g_wholeMs[2] += t3 - t2, g_wholeMs[3] += t4 - t3, g_wholeMs[4] += t5 - t4, g_wholeMs[5] += 1.0;
}

// Accumulated wall time of the whole-step shim's phases since the last call: stage 1 + 2 on the host, new contacts /
// upload, s2amd_world_step, download, applying the results to the pools and trees; [5] = steps.
S2REF_API void s2ref_world_timing(double out[6])
{
	for (int i = 0; i < 6; ++i)
	{
		out[i] = g_wholeMs[i];
		g_wholeMs[i] = 0.0;
	}
}

// The host pools of `id` brought up to date with the device (manifolds, caches, joint impulses).
S2REF_API int s2ref_world_sync(s2WorldId id)
{
	return g_mirror.world == s2GetWorldFromId(id) ? mirrorSync() : 0;
}

// After editing a resident world through the reference's API (velocities, forces, filters, joints' settings ...):
// the next step uploads it again.  Creating or destroying bodies, shapes and joints is noticed without this.
S2REF_API void s2ref_world_invalidate(void)
{
	mirrorSync();
	g_mirror.world = NULL;
}

// on: stage 1's pair discovery on the device as well (the host trees are still kept up to date, not queried)
S2REF_API void s2ref_world_device_pairs(int on)
{
	g_devicePairs = on;
}

S2REF_API long s2ref_world_uploads(void)
{
	return g_mirror.uploads;
}

static void mirrorSyncIfResident(void)
{
	if (g_wholeStep)
	{
		mirrorSync();
		g_mirror.world = NULL;
		g_wholeStep = 0;
	}
}

// As s2ref_use_amd, plus stage 3 and stage 4: the whole of s2World_Step but its tree and pool bookkeeping on the GPU.
S2REF_API int s2ref_use_amd_world(const char* path, int device)
{
	if (g_wholeStep)
	{
		mirrorSync();
	}
	g_wholeStep = 0;
	g_mirror.world = NULL;
	int rc = s2ref_use_amd(path, device);
	if (rc != 0 || path == NULL)
	{
		return rc;
	}
	g_amdWorldUpload = (s2amdWorldUploadFcn*)dlsym(g_amdLib, "s2amd_world_upload");
	g_amdWorldStep = (s2amdWorldStepFcn*)dlsym(g_amdLib, "s2amd_world_step");
	g_amdWorldSetContacts = (s2amdWorldSetContactsFcn*)dlsym(g_amdLib, "s2amd_world_set_contacts");
	g_amdWorldDownload = (s2amdWorldDownloadFcn*)dlsym(g_amdLib, "s2amd_world_download");
	g_amdWorldFindPairs = (s2amdWorldFindPairsFcn*)dlsym(g_amdLib, "s2amd_world_find_pairs");
	if (!g_amdWorldUpload || !g_amdWorldStep || !g_amdWorldSetContacts || !g_amdWorldDownload || !g_amdWorldFindPairs)
	{
		return -4;
	}
	g_wholeStep = 1;
	return 0;
}

S2REF_API int s2ref_replace_error(void)
{
	return g_replaceError;
}

S2REF_API int s2ref_capture_count(void)
{
	return g_captureCount;
}

S2REF_API const s2amdStepParams* s2ref_params(void)
{
	return &g_params;
}

// which: 0 = state at solver entry, 1 = state at solver exit
S2REF_API int s2ref_snapshot(int which, const s2amdBody** bodies, int32_t* bodyCapacity, const s2amdContact** contacts,
							 int32_t* contactCapacity, const s2amdJoint** joints, int32_t* jointCapacity)
{
	const Snapshot* s = which == 0 ? &g_pre : &g_post;
	*bodies = s->bodies, *bodyCapacity = s->bodyCapacity;
	*contacts = s->contacts, *contactCapacity = s->contactCapacity;
	*joints = s->joints, *jointCapacity = s->jointCapacity;
	return 0;
}

// Current world state in wire format (outside a step).
S2REF_API int s2ref_world_sizes(s2WorldId id, int32_t* bodyCapacity, int32_t* contactCapacity, int32_t* jointCapacity)
{
	s2World* world = s2GetWorldFromId(id);
	*bodyCapacity = world->bodyPool.capacity;
	*contactCapacity = world->contactPool.capacity;
	*jointCapacity = world->jointPool.capacity;
	return 0;
}

S2REF_API int s2ref_pack_world(s2WorldId id, s2amdBody* bodies, s2amdContact* contacts, s2amdJoint* joints)
{
	s2World* world = s2GetWorldFromId(id);
	if (g_mirror.world == world)
	{
		mirrorSync();
	}
	if (bodies)
		packBodies(world, bodies);
	if (contacts)
		packContacts(world, contacts);
	if (joints)
		packJoints(world, joints);
	return 0;
}

// Contact pair table: (shapeIndexA, shapeIndexB) per contact slot, -1 for free slots.
S2REF_API int s2ref_contact_pairs(s2WorldId id, int32_t* shapeA, int32_t* shapeB, int32_t capacity)
{
	s2World* world = s2GetWorldFromId(id);
	int n = world->contactPool.capacity;
	if (capacity < n)
	{
		return -1;
	}
	for (int i = 0; i < n; ++i)
	{
		const s2Contact* c = world->contacts + i;
		if (s2IsFree(&c->object))
		{
			shapeA[i] = -1, shapeB[i] = -1;
		}
		else
		{
			shapeA[i] = c->shapeIndexA, shapeB[i] = c->shapeIndexB;
		}
	}
	return n;
}

// Convenience for ctypes callers: step through the public entry point.
S2REF_API void s2ref_step(s2WorldId id, float dt, int velIters, int posIters, int warmStart)
{
	s2World_Step(id, dt, velIters, posIters, warmStart != 0);
}

S2REF_API void s2ref_destroy_world(s2WorldId id)
{
	s2DestroyWorld(id);
}

S2REF_API size_t s2ref_sizeof(int what)
{
	switch (what)
	{
		case 0:
			return sizeof(s2Body);
		case 1:
			return sizeof(s2Contact);
		case 2:
			return sizeof(s2Joint);
		case 3:
			return sizeof(s2ContactConstraint);
		default:
			return 0;
	}
}
