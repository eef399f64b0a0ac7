// s2_amd_binding.c -- the reference-side binding of the MI355X hot path (INTEGRATION.md shows where each piece hooks in).
//
// This is the file a maintainer of erincatto/solver2d adds to src/ to run the solvers -- or the whole of s2World_Step
// but its tree and pool bookkeeping -- on libs2amd.so.  It is compiled against the reference's own internal headers
// (body.h, contact.h, joint.h, shape.h, world.h) and against include/solver2d_amd.h, and it is all there is between the
// reference's pools and the C-ABI:
//
//   * field-for-field gather / scatter between the pools and the wire structs (s2amdBinding_Pack* / _Unpack*);
//   * s2amdBinding_Solve(world, context, solverType)  ==  s2Solve_<solverType>(world, context)   (src/solvers.h:70-79):
//     gather, s2amd_solve, scatter -- the drop-in for the switch in s2World_Step (src/world.c:206-256);
//   * s2amdBinding_WorldStep(world, dt, velIters, posIters, warmStart)  ==  s2World_Step (src/world.c:120-301) with stages 1
//     and 2 (dynamic trees, contact pool) left to the reference and stage 3 (update contacts), the solve and stage 4
//     (refit) on the resident world chain (s2amd_world_upload / _set_contacts / _step / _separated / _download);
//   * ONE device state per world: s2amdSolver handles and host mirrors live in a table indexed by s2World.index
//     (src/world.c:29, include/solver2d/constants.h:12: 32 worlds), created on a world's first step and released by
//     s2amdBinding_DestroyWorld, which s2DestroyWorld (src/world.c:105-118) calls first.  The samples' GUI steps up to
//     ten worlds per frame (samples/main.cpp:805-813): each keeps its own resident world, structure and step graph.
//
// libs2amd.so is loaded with dlopen: no link-time dependency on the HIP runtime, and no CPU fallback -- when the library
// or a GPU is missing s2amdBinding_Open fails and the reference keeps its own solvers.
#include "s2_amd_binding.h"

#include "body.h"
#include "contact.h"
#include "core.h"
#include "joint.h"
#include "shape.h"
#include "solvers.h"
#include "stack_allocator.h"
#include "world.h"

#include "solver2d/constants.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

// ---- gather / scatter between the reference's pools and the wire structs ----


void s2amdBinding_PackBodies(const s2World* world, s2amdBody* out)
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


void s2amdBinding_UnpackBodies(s2World* world, const s2amdBody* in)
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


void s2amdBinding_PackContacts(const s2World* world, s2amdContact* out)
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


void s2amdBinding_UnpackContacts(s2World* world, const s2amdContact* in)
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


void s2amdBinding_PackJoints(const s2World* world, s2amdJoint* out)
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


void s2amdBinding_UnpackJoints(s2World* world, const s2amdJoint* in)
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


void s2amdBinding_PackShapes(const s2World* world, s2amdShape* out)
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


void s2amdBinding_PackPairs(const s2World* world, s2amdPairState* out)
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


static void unpackManifolds(s2World* world, const s2amdContact* in, const s2amdPairState* pairs, int count)
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

// ---- the library ----

typedef struct AmdApi
{
	void* lib;
	int device;
	int (*create)(int, s2amdSolver**);
	void (*destroy)(s2amdSolver*);
	const char* (*lastError)(void);
	int (*solve)(s2amdSolver*, const s2amdStepParams*, s2amdBody*, int32_t, s2amdContact*, int32_t, s2amdJoint*, int32_t);
	int (*worldUpload)(s2amdSolver*, const s2amdBody*, int32_t, const s2amdContact*, int32_t, const s2amdJoint*, int32_t, const s2amdShape*, int32_t,
					   const s2amdPairState*, const float*);
	int (*worldStep)(s2amdSolver*, const s2amdStepParams*, s2amdWorldStepInfo*);
	int (*worldSetContacts)(s2amdSolver*, const int32_t*, int32_t, const s2amdContact*, const s2amdPairState*);
	int (*worldDownload)(s2amdSolver*, s2amdBody*, int32_t, s2amdContact*, int32_t, s2amdJoint*, int32_t, s2amdShape*, int32_t, s2amdPairState*, float*,
						 int32_t*);
	int (*worldFindPairs)(s2amdSolver*, int32_t*, int32_t, int32_t*);
	int (*worldSeparated)(s2amdSolver*, int32_t*, int32_t, int32_t*);
	int (*worldDownloadBoxes)(s2amdSolver*, s2amdShapeBox*, int32_t);
	int (*worldSetRefitOrder)(s2amdSolver*, const int32_t*, int32_t);
	int (*worldDownloadStep)(s2amdSolver*, float*, int32_t, s2amdMovedBox*, int32_t, int32_t*);
	int (*worldSetTree)(s2amdSolver*, int32_t, const s2amdTreeNode*, int32_t, int32_t);
	int (*worldGetTree)(s2amdSolver*, int32_t, s2amdTreeNode*, int32_t, int32_t*);
	int (*setOption)(s2amdSolver*, const char*, int32_t);
} AmdApi;
static AmdApi s_api = {0};

// the device state of ONE world
typedef struct WorldBinding
{
	s2amdSolver* solver;
	// solver-only route: the wire arrays of the last s2amdBinding_Solve
	s2amdBody* solveBodies;
	s2amdContact* solveContacts;
	s2amdJoint* solveJoints;
	// whole-step route: what of this world is resident on the device
	int resident;
	uint64_t stepId; // s2World.stepId after the last step taken here: a new world in a re-used slot starts at 0 again; steps taken elsewhere show up too
	int bodyCapacity, bodyCount, shapeCapacity, shapeCount, jointCapacity, jointCount, contactCapacity;
	int contactsStale; // the device holds newer manifolds / impulses / joint impulses than the host pools
	s2amdBody* bodies;
	s2amdContact* contacts;
	s2amdJoint* joints;
	s2amdShape* shapes;
	s2amdPairState* pairs;
	float* origins;
	int32_t* separated;
	int64_t* liveKey; // shapeIndexA << 32 | shapeIndexB of the slot as the device knows it, -1: free there
	int32_t* slots;
	s2amdContact* slotContacts;
	s2amdPairState* slotPairs;
	int32_t* newPairs;
	int newPairCapacity;
	s2amdShapeBox* boxes; // the shapes' boxes after the last step (stage 4's output)
	// lean read-back (device pairs): per step only what the public API reads -- body origins and rotations -- and the fat
	// boxes the refit re-inflated; velocities, centres of mass, tight boxes and the trees follow when somebody needs them
	float* poses;	 // {origin.x, origin.y, rot.s, rot.c} per body slot
	int32_t* refitOrder; // shape slots in the order the reference's refit visits them (bodies in pool order, each body's shape list)
	int bodiesStale; // the host bodies hold only the poses of the last step
	int boxesStale;	 // the host shapes' tight boxes are older than the device's
	// tree work not yet done: per step the boxes it re-inflated, in refit order.  Replaying a step = what the reference does
	// between two pair queries: clear the move buffer, rebuild the trees (stage 2), enlarge these proxies (stage 4).
	s2amdMovedBox* pendingBoxes;
	int pendingBoxCount, pendingBoxCapacity;
	int* pendingSteps; // boxes per pending step
	int pendingStepCount, pendingStepCapacity;
	int lastMoved;	  // boxes the last step re-inflated: the next pair query has something to look at
	// (round 6) the device holds the reference's trees (s2amd_world_set_tree) and maintains them itself: the pair query comes back in
	// creation order and nothing is replayed here.  While `treesStale` the host's trees, fat boxes and move buffer are those of the last
	// upload; pullTrees copies the device's back when somebody reads them.  The log then holds the LAST step's boxes only (the move buffer).
	int deviceTrees;
	int treesStale;
	int liveCount;	  // slots with liveKey >= 0
	int createdCount; // >= 0: this step's stage 1 ran here (device pairs) and created exactly the contacts in createdSlots
	int32_t* createdSlots;
	int createdCapacity;
} WorldBinding;
static WorldBinding s_bindings[s2_maxWorlds];
static long s_uploads = 0, s_steps = 0;
static int s_lastError = 0;
static int s_devicePairs = 0;
static int s_deviceTrees = -1;	// -1: not read yet (S2AMD_DEVICE_TREES, default 1); 0: round 5's host replay; 1: trees on the device
static int s_checkTrees = -1;	// S2AMD_CHECK_TREES=1: both at once -- the host replay (s2amdBinding_OrderPairs) checks the device's order and trees
static long s_treeCheck[4] = {0, 0, 0, 0}; // queries compared, queries whose order differed, tree comparisons, trees that differed
static double s_phaseMs[6] = {0}; // stage 1+2, sync in, device step, download, apply, steps

static double wallMs(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

// realloc that does not come back empty-handed: the mirrors are as large as the reference's own pools, whose allocator
// (src/allocate.c) does not survive an exhausted heap either -- fail loudly instead of writing through NULL
static void* growTo(void* p, size_t count, size_t size)
{
	void* q = realloc(p, (count > 0 ? count : 1) * size);
	if (q == NULL)
	{
		fprintf(stderr, "s2amd binding: out of memory (%zu x %zu bytes)\n", count, size);
		abort();
	}
	return q;
}

int s2amdBinding_Open(const char* libraryPath, int device)
{
	if (s_api.lib == NULL)
	{
		s_api.lib = dlopen(libraryPath, RTLD_NOW | RTLD_LOCAL);
		if (s_api.lib == NULL)
		{
			return -1;
		}
#define S2_BIND(member, symbol)                                                                                                  \
	*(void**)(&s_api.member) = dlsym(s_api.lib, symbol);                                                                         \
	if (s_api.member == NULL)                                                                                                    \
	{                                                                                                                            \
		return -2;                                                                                                               \
	}
		// the stats / wire structs are compiled into this file: refuse a library built from another header
		int (*apiVersion)(void) = NULL;
		*(void**)(&apiVersion) = dlsym(s_api.lib, "s2amd_api_version");
		if (apiVersion == NULL || apiVersion() != S2AMD_API_VERSION)
		{
			dlclose(s_api.lib);
			s_api.lib = NULL;
			return -4;
		}
		S2_BIND(create, "s2amd_create")
		S2_BIND(destroy, "s2amd_destroy")
		S2_BIND(lastError, "s2amd_last_error")
		S2_BIND(solve, "s2amd_solve")
		S2_BIND(worldUpload, "s2amd_world_upload")
		S2_BIND(worldStep, "s2amd_world_step")
		S2_BIND(worldSetContacts, "s2amd_world_set_contacts")
		S2_BIND(worldDownload, "s2amd_world_download")
		S2_BIND(worldFindPairs, "s2amd_world_find_pairs")
		S2_BIND(worldSeparated, "s2amd_world_separated")
		S2_BIND(worldDownloadBoxes, "s2amd_world_download_boxes")
		S2_BIND(worldSetRefitOrder, "s2amd_world_set_refit_order")
		S2_BIND(worldDownloadStep, "s2amd_world_download_step")
		S2_BIND(worldSetTree, "s2amd_world_set_tree")
		S2_BIND(worldGetTree, "s2amd_world_get_tree")
		S2_BIND(setOption, "s2amd_set_option")
#undef S2_BIND
	}
	s_api.device = device;
	// fail now, not at the first step, when there is no GPU (the library has no CPU path)
	s2amdSolver* probe = NULL;
	if (s_api.create(device, &probe) != 0 || probe == NULL)
	{
		return -3;
	}
	s_api.destroy(probe);
	s_lastError = 0;
	return 0;
}

int s2amdBinding_IsOpen(void)
{
	return s_api.lib != NULL;
}

int s2amdBinding_LastError(void)
{
	return s_lastError;
}

// The call sites of shim/call_sites.patch: once the binding is open there is no way back to the reference's CPU solver -- a device
// error must not silently run s2Solve_* on the host on the same world (the two would disagree from that step on).
void s2amdBinding_SolveOrDie(s2World* world, s2StepContext* context, int solverType)
{
	const int rc = s2amdBinding_Solve(world, context, solverType);
	if (rc != 0)
	{
		fprintf(stderr, "s2amd binding: s2Solve (solver type %d) failed on the device (error %d): %s\n", solverType, rc,
				s_api.lastError ? s_api.lastError() : "");
		abort();
	}
}

void s2amdBinding_WorldStepOrDie(s2World* world, float timeStep, int velIters, int posIters, bool warmStart, void (*updatePairs)(s2World*),
								 void (*rebuildTrees)(s2BroadPhase*))
{
	s_lastError = 0;
	s2amdBinding_WorldStep(world, timeStep, velIters, posIters, warmStart, updatePairs, rebuildTrees);
	if (s_lastError != 0)
	{
		fprintf(stderr, "s2amd binding: s2World_Step failed on the device (error %d): %s\n", s_lastError, s_api.lastError ? s_api.lastError() : "");
		abort();
	}
}

long s2amdBinding_Uploads(void)
{
	return s_uploads;
}

void s2amdBinding_DevicePairs(int on)
{
	s_devicePairs = on;
}

void s2amdBinding_Timing(double out[6])
{
	for (int i = 0; i < 6; ++i)
	{
		out[i] = s_phaseMs[i];
		s_phaseMs[i] = 0.0;
	}
}

static WorldBinding* bindingOf(const s2World* world)
{
	if (s_api.lib == NULL || world->index < 0 || world->index >= s2_maxWorlds)
	{
		return NULL;
	}
	WorldBinding* b = s_bindings + world->index;
	if (b->solver == NULL)
	{
		if (s_api.create(s_api.device, &b->solver) != 0 || b->solver == NULL)
		{
			b->solver = NULL;
			return NULL;
		}
		// S2AMD_OPTIONS="key=value,key=value": s2amd_set_option for every solver the binding makes (include/solver2d_amd.h lists the keys)
		const char* options = getenv("S2AMD_OPTIONS");
		if (options != NULL)
		{
			char buffer[512];
			strncpy(buffer, options, sizeof(buffer) - 1);
			buffer[sizeof(buffer) - 1] = 0;
			for (char* item = strtok(buffer, ","); item != NULL; item = strtok(NULL, ","))
			{
				char* eq = strchr(item, '=');
				if (eq != NULL)
				{
					*eq = 0;
					if (s_api.setOption(b->solver, item, (int32_t)atoi(eq + 1)) != 0)
					{
						fprintf(stderr, "s2amd binding: S2AMD_OPTIONS: option \"%s\" refused: %s\n", item, s_api.lastError());
					}
				}
			}
		}
	}
	return b;
}

// ---- deferred tree work (lean read-back) ----
// A world nobody queries (no s2World_QueryAABB, no s2World_Draw, no new pair) never needs its trees: the log is only replayed
// when something reads them, or when it holds this much (40 MB of boxes).
#define S2AMD_BINDING_MAX_PENDING_STEPS 4096
#define S2AMD_BINDING_MAX_PENDING_BOXES 2000000
static void (*s_rebuildTrees)(s2BroadPhase*) = NULL; // stage 2 as handed to s2amdBinding_WorldStep

// The steps whose re-inflated boxes were only logged, replayed into the reference's trees in order: per step the move buffer
// is cleared (as the pair query of that step would have left it, src/broad_phase.c end of s2UpdateBroadPhasePairs), the
// trees are rebuilt (world.c:130) and the step's proxies enlarged in refit order (world.c:283-290) -- the very calls the
// per-step path makes, later.  Afterwards trees, fat boxes and move buffer are what they would be had every step done it.
static void replayTrees(s2World* world, WorldBinding* b)
{
	if (b->pendingStepCount == 0 || s_rebuildTrees == NULL)
	{
		return;
	}
	s2BroadPhase* bp = &world->broadPhase;
	const s2amdMovedBox* e = b->pendingBoxes;
	for (int st = 0; st < b->pendingStepCount; ++st)
	{
		s2Array_Clear(bp->moveArray);
		s2ClearSet(&bp->moveSet);
		s_rebuildTrees(bp);
		for (int i = 0; i < b->pendingSteps[st]; ++i, ++e)
		{
			if (e->shape < 0 || e->shape >= world->shapePool.capacity || s2IsFree(&world->shapes[e->shape].object))
			{
				continue;
			}
			s2Shape* sh = world->shapes + e->shape;
			sh->fatAABB = (s2Box){{e->fatAABB[0], e->fatAABB[1]}, {e->fatAABB[2], e->fatAABB[3]}};
			s2BroadPhase_EnlargeProxy(bp, sh->proxyKey, sh->fatAABB);
		}
	}
	b->pendingStepCount = 0;
	b->pendingBoxCount = 0;
}

// ---- the trees on the device (round 6; include/solver2d_amd.h: s2amd_world_set_tree) ----
_Static_assert(sizeof(s2amdTreeNode) == sizeof(s2TreeNode), "s2amdTreeNode mirrors s2TreeNode byte for byte");

static int deviceTreesWanted(void)
{
	if (s_deviceTrees < 0)
	{
		const char* e = getenv("S2AMD_DEVICE_TREES");
		s_deviceTrees = e != NULL && atoi(e) == 0 ? 0 : 1;
		const char* c = getenv("S2AMD_CHECK_TREES");
		s_checkTrees = c != NULL && atoi(c) != 0 ? 1 : 0;
	}
	return s_deviceTrees;
}

// After s2amd_world_upload: the three trees as they stand -- stage 2 has run, so no internal node is flagged (the library refuses a
// tree that is: one a host edit left flags in, src/dynamic_tree.c:610-684 inserts unflagged parents between flagged nodes.  Such a
// world keeps round 5's host replay until its next upload).
static void sendTrees(s2World* world, WorldBinding* b)
{
	b->deviceTrees = 0;
	b->treesStale = 0;
	if (!s_devicePairs || !deviceTreesWanted())
	{
		return;
	}
	const s2BroadPhase* bp = &world->broadPhase;
	for (int type = 0; type < s2_bodyTypeCount; ++type)
	{
		const s2DynamicTree* tree = bp->trees + type;
		if (s_api.worldSetTree(b->solver, type, (const s2amdTreeNode*)tree->nodes, tree->nodeCapacity, tree->root) != 0)
		{
			return; // (the next s2amd_world_upload forgets what was sent)
		}
	}
	b->deviceTrees = 1;
}

// The device's trees into the reference's own (kinematic and dynamic: the static one never changes on the device), the fat boxes from
// their leaves, the move buffer as the last refit left it (world.c:283-290: the shapes it re-inflated, in its order).  What a replay of
// every step since the upload would have arrived at.
static int pullTrees(s2World* world, WorldBinding* b)
{
	s2BroadPhase* bp = &world->broadPhase;
	for (int type = s2_kinematicBody; type <= s2_dynamicBody; ++type)
	{
		s2DynamicTree* tree = bp->trees + type;
		int32_t root = S2_NULL_INDEX;
		const int rc = s_api.worldGetTree(b->solver, type, (s2amdTreeNode*)tree->nodes, tree->nodeCapacity, &root);
		if (rc != 0)
		{
			fprintf(stderr, "s2amd binding: the device's tree %d could not be read back (%d): %s\n", type, rc, s_api.lastError());
			s_lastError = rc;
			return rc;
		}
		tree->root = root;
	}
	const int ns = world->shapePool.capacity < b->shapeCapacity ? world->shapePool.capacity : b->shapeCapacity;
	for (int i = 0; i < ns; ++i)
	{
		s2Shape* sh = world->shapes + i;
		if (s2IsFree(&sh->object) || S2_PROXY_TYPE(sh->proxyKey) == s2_staticBody)
		{
			continue;
		}
		sh->fatAABB = bp->trees[S2_PROXY_TYPE(sh->proxyKey)].nodes[S2_PROXY_ID(sh->proxyKey)].aabb;
	}
	s2Array_Clear(bp->moveArray);
	s2ClearSet(&bp->moveSet);
	const s2amdMovedBox* e = b->pendingBoxes;
	for (int i = 0; i < b->pendingBoxCount; ++i, ++e)
	{
		if (e->shape >= 0 && e->shape < world->shapePool.capacity && !s2IsFree(&world->shapes[e->shape].object))
		{
			s2BufferMove(bp, world->shapes[e->shape].proxyKey);
		}
	}
	b->pendingStepCount = 0;
	b->pendingBoxCount = 0;
	b->treesStale = 0;
	return 0;
}

// S2AMD_CHECK_TREES: the reference's trees after the replay against the device's, node for node
static void checkTrees(s2World* world, WorldBinding* b)
{
	static s2amdTreeNode* got;
	static int gotCapacity;
	const s2BroadPhase* bp = &world->broadPhase;
	for (int type = s2_kinematicBody; type <= s2_dynamicBody; ++type)
	{
		const s2DynamicTree* tree = bp->trees + type;
		if (gotCapacity < tree->nodeCapacity)
		{
			gotCapacity = tree->nodeCapacity + 1024;
			got = (s2amdTreeNode*)growTo(got, (size_t)gotCapacity, sizeof(s2amdTreeNode));
		}
		int32_t root = S2_NULL_INDEX;
		s_treeCheck[2] += 1;
		if (s_api.worldGetTree(b->solver, type, got, tree->nodeCapacity, &root) != 0 || root != tree->root)
		{
			s_treeCheck[3] += 1;
			continue;
		}
		int same = 1;
		for (int i = 0; i < tree->nodeCapacity && same; ++i)
		{
			const s2TreeNode* r = tree->nodes + i;
			const s2amdTreeNode* g = got + i;
			same = r->parent == g->parent && r->height == g->height;
			if (same && r->height >= 0)
			{
				same = r->child1 == g->child1 && r->child2 == g->child2 && r->userData == g->userData && r->categoryBits == g->categoryBits &&
					   (r->enlarged ? 1 : 0) == (g->enlarged ? 1 : 0) && r->aabb.lowerBound.x == g->aabb[0] && r->aabb.lowerBound.y == g->aabb[1] &&
					   r->aabb.upperBound.x == g->aabb[2] && r->aabb.upperBound.y == g->aabb[3];
			}
			if (!same && getenv("S2AMD_CHECK_TREES_VERBOSE") != NULL)
			{
				fprintf(stderr, "tree %d node %d: reference {parent %d children %d %d height %d enlarged %d box %g %g %g %g} device {%d, %d %d, %d, %d, %g %g %g %g}\n", type,
						i, r->parent, r->child1, r->child2, r->height, r->enlarged, r->aabb.lowerBound.x, r->aabb.lowerBound.y, r->aabb.upperBound.x,
						r->aabb.upperBound.y, g->parent, g->child1, g->child2, g->height, g->enlarged, g->aabb[0], g->aabb[1], g->aabb[2], g->aabb[3]);
			}
		}
		s_treeCheck[3] += same ? 0 : 1;
	}
}

void s2amdBinding_DeviceTrees(int on, int check)
{
	s_deviceTrees = on ? 1 : 0;
	s_checkTrees = check ? 1 : 0;
}

void s2amdBinding_TreeCheck(long out[4])
{
	for (int i = 0; i < 4; ++i)
	{
		out[i] = s_treeCheck[i];
		s_treeCheck[i] = 0;
	}
}

// the host's trees, fat boxes and move buffer as the reference would hold them now
static void flushTrees(s2World* world, WorldBinding* b)
{
	if (b->deviceTrees && !s_checkTrees)
	{
		if (b->treesStale)
		{
			(void)pullTrees(world, b);
		}
		return;
	}
	replayTrees(world, b);
}

// room in the log for one more step of `boxes` re-inflated boxes
static int reservePending(WorldBinding* b, int boxes)
{
	if (b->pendingBoxCount + boxes > b->pendingBoxCapacity)
	{
		const int cap = 2 * (b->pendingBoxCount + boxes) + 1024;
		void* p = realloc(b->pendingBoxes, (size_t)cap * sizeof(s2amdMovedBox));
		if (p == NULL)
		{
			return S2AMD_E_DEVICE;
		}
		b->pendingBoxes = (s2amdMovedBox*)p, b->pendingBoxCapacity = cap;
	}
	if (b->pendingStepCount + 1 > b->pendingStepCapacity)
	{
		const int cap = 2 * b->pendingStepCount + 64;
		void* p = realloc(b->pendingSteps, (size_t)cap * sizeof(int));
		if (p == NULL)
		{
			return S2AMD_E_DEVICE;
		}
		b->pendingSteps = (int*)p, b->pendingStepCapacity = cap;
	}
	return 0;
}

// the order the reference's refit visits the shapes in (src/world.c:259-297): bodies in pool order, each body's shape list
static int sendRefitOrder(s2World* w, WorldBinding* b)
{
	void* p = realloc(b->refitOrder, (size_t)(w->shapePool.capacity > 0 ? w->shapePool.capacity : 1) * sizeof(int32_t));
	if (p == NULL)
	{
		return S2AMD_E_DEVICE;
	}
	b->refitOrder = (int32_t*)p;
	int n = 0;
	for (int bi = 0; bi < w->bodyPool.capacity; ++bi)
	{
		const s2Body* body = w->bodies + bi;
		if (s2IsFree(&body->object) || body->type == s2_staticBody)
		{
			continue;
		}
		for (int i = body->shapeList; i != S2_NULL_INDEX && n < w->shapePool.capacity; i = w->shapes[i].nextShapeIndex)
		{
			b->refitOrder[n++] = i;
		}
	}
	return s_api.worldSetRefitOrder(b->solver, b->refitOrder, n);
}

// everything the lean read-back left on the device: the bodies (velocities, centres of mass), the tight boxes, the trees
static int syncBodiesAndBoxes(s2World* world, WorldBinding* b)
{
	int rc = 0;
	if (b->bodiesStale)
	{
		if ((rc = s_api.worldDownload(b->solver, b->bodies, b->bodyCapacity, NULL, b->contactCapacity, NULL, b->jointCapacity, NULL, b->shapeCapacity, NULL,
									  b->origins, NULL)) != 0)
		{
			return rc;
		}
		// (the pool may have grown since: slots keep their index, src/pool.h:11)
		const int n = world->bodyPool.capacity < b->bodyCapacity ? world->bodyPool.capacity : b->bodyCapacity;
		for (int i = 0; i < n; ++i)
		{
			s2Body* body = world->bodies + i;
			const s2amdBody* o = b->bodies + i;
			// (a slot the device holds as FREE: the body in it was created after the upload -- its record there is all zeros, and the
			// next step uploads the pools again)
			if (s2IsFree(&body->object) || o->type == S2AMD_BODY_FREE)
			{
				continue;
			}
			body->position = (s2Vec2){o->position[0], o->position[1]};
			body->rot = (s2Rot){o->rot[0], o->rot[1]};
			body->linearVelocity = (s2Vec2){o->linearVelocity[0], o->linearVelocity[1]};
			body->angularVelocity = o->angularVelocity;
			body->deltaPosition = (s2Vec2){o->deltaPosition[0], o->deltaPosition[1]};
		}
		b->bodiesStale = 0;
	}
	if (b->boxesStale)
	{
		if ((rc = s_api.worldDownloadBoxes(b->solver, b->boxes, b->shapeCapacity)) != 0)
		{
			return rc;
		}
		const int ns = world->shapePool.capacity < b->shapeCapacity ? world->shapePool.capacity : b->shapeCapacity;
		for (int i = 0; i < ns; ++i)
		{
			s2Shape* sh = world->shapes + i;
			if (!s2IsFree(&sh->object))
			{
				const s2amdShapeBox* o = b->boxes + i;
				sh->aabb = (s2Box){{o->aabb[0], o->aabb[1]}, {o->aabb[2], o->aabb[3]}};
			}
		}
		b->boxesStale = 0;
	}
	flushTrees(world, b);
	return 0;
}

// manifolds, GJK caches and joint impulses back into the reference's pools
static int syncToPools(s2World* world, WorldBinding* b)
{
	const uint64_t id = (uint64_t)world->stepId;
	if (b->resident && (id == b->stepId || id == b->stepId + 1))
	{
		int rcLean = syncBodiesAndBoxes(world, b);
		if (rcLean != 0)
		{
			return rcLean;
		}
	}
	if (!b->resident || (id != b->stepId && id != b->stepId + 1) || !b->contactsStale)
	{
		return 0;
	}
	int rc = s_api.worldDownload(b->solver, NULL, b->bodyCapacity, b->contacts, b->contactCapacity, b->joints, b->jointCapacity, NULL, b->shapeCapacity,
								 b->pairs, NULL, NULL);
	if (rc != 0)
	{
		return rc;
	}
	unpackManifolds(world, b->contacts, b->pairs, b->contactCapacity);
	if (world->jointPool.capacity == b->jointCapacity && world->jointPool.count == b->jointCount)
	{
		s2amdBinding_UnpackJoints(world, b->joints); // (joints were created or destroyed: their impulses start over)
	}
	b->contactsStale = 0;
	return 0;
}

int s2amdBinding_Sync(s2World* world)
{
	WorldBinding* b = s_api.lib != NULL && world->index >= 0 && world->index < s2_maxWorlds ? s_bindings + world->index : NULL;
	return b != NULL && b->solver != NULL ? syncToPools(world, b) : 0;
}

void s2amdBinding_Invalidate(s2World* world)
{
	WorldBinding* b = s_api.lib != NULL && world->index >= 0 && world->index < s2_maxWorlds ? s_bindings + world->index : NULL;
	if (b != NULL && b->solver != NULL)
	{
		(void)syncToPools(world, b);
		b->resident = 0;
	}
}

// s2DestroyWorld (src/world.c:105-118) calls this before it frees the pools: the world's device state goes with it
void s2amdBinding_DestroyWorld(s2World* world)
{
	if (world->index < 0 || world->index >= s2_maxWorlds)
	{
		return;
	}
	WorldBinding* b = s_bindings + world->index;
	if (b->solver != NULL && s_api.destroy != NULL)
	{
		s_api.destroy(b->solver);
	}
	void* owned[] = {b->solveBodies, b->solveContacts, b->solveJoints, b->bodies,		b->contacts,	 b->joints,	   b->shapes,	b->pairs,
					 b->origins,	 b->separated,	   b->liveKey,	   b->slots,		b->slotContacts, b->slotPairs, b->newPairs,
					 b->createdSlots,	 b->boxes,	 b->poses,	  b->refitOrder,	b->pendingBoxes, b->pendingSteps};
	for (size_t i = 0; i < sizeof(owned) / sizeof(owned[0]); ++i)
	{
		free(owned[i]);
	}
	memset(b, 0, sizeof(*b));
}

// every world's manifolds back to its pools and every device state released (the library stays loaded)
void s2amdBinding_Close(void)
{
	for (int i = 0; i < s2_maxWorlds; ++i)
	{
		WorldBinding* b = s_bindings + i;
		if (b->solver == NULL)
		{
			continue;
		}
		s2World* world = s2GetWorldFromIndex((int16_t)i);
		if (world->blockAllocator != NULL)
		{
			(void)syncToPools(world, b);
			s2amdBinding_DestroyWorld(world);
		}
		else
		{
			s2World gone = {0};
			gone.index = (int16_t)i;
			s2amdBinding_DestroyWorld(&gone);
		}
	}
}

static void fillParams(const s2World* world, s2amdStepParams* p, int solverType, float dt, int velIters, int posIters, int warmStart)
{
	p->solverType = solverType;
	p->dt = dt;
	p->velIters = velIters;
	p->posIters = posIters;
	p->warmStart = warmStart;
	p->gravity[0] = world->gravity.x;
	p->gravity[1] = world->gravity.y;
}

// == s2Solve_<solverType>(world, context): the plug point of src/solvers.h:70-79
int s2amdBinding_Solve(s2World* world, s2StepContext* context, int solverType)
{
	WorldBinding* b = bindingOf(world);
	if (b == NULL)
	{
		return s_lastError = S2AMD_E_NODEVICE;
	}
	if (b->resident)
	{
		// this world was last stepped by the whole-step route: its manifolds are on the device
		(void)syncToPools(world, b);
		b->resident = 0;
	}
	const int nb = world->bodyPool.capacity, nc = world->contactPool.capacity, nj = world->jointPool.capacity;
	b->solveBodies = (s2amdBody*)growTo(b->solveBodies, (size_t)nb, sizeof(s2amdBody));
	b->solveContacts = (s2amdContact*)growTo(b->solveContacts, (size_t)nc, sizeof(s2amdContact));
	b->solveJoints = (s2amdJoint*)growTo(b->solveJoints, (size_t)nj, sizeof(s2amdJoint));
	s2amdBinding_PackBodies(world, b->solveBodies);
	s2amdBinding_PackContacts(world, b->solveContacts);
	s2amdBinding_PackJoints(world, b->solveJoints);
	s2amdStepParams params;
	fillParams(world, &params, solverType, context->dt, context->iterations, context->extraIterations, context->warmStart ? 1 : 0);
	int rc = s_api.solve(b->solver, &params, b->solveBodies, nb, b->solveContacts, nc, b->solveJoints, nj);
	if (rc != 0)
	{
		fprintf(stderr, "s2Solve on the GPU failed (%d): %s\n", rc, s_api.lastError());
		return s_lastError = rc;
	}
	s2amdBinding_UnpackBodies(world, b->solveBodies);
	s2amdBinding_UnpackContacts(world, b->solveContacts);
	s2amdBinding_UnpackJoints(world, b->solveJoints);
	return 0;
}

// ---- whole step ----

static int residentMatches(const s2World* w, const WorldBinding* b)
{
	return b->resident && b->stepId + 1 == (uint64_t)w->stepId && b->bodyCapacity == w->bodyPool.capacity && b->bodyCount == w->bodyPool.count &&
		   b->shapeCapacity == w->shapePool.capacity && b->shapeCount == w->shapePool.count && b->jointCapacity == w->jointPool.capacity &&
		   b->jointCount == w->jointPool.count && b->contactCapacity == w->contactPool.capacity;
}

static int uploadWorld(s2World* w, WorldBinding* b, int leanStep)
{
	int rc = 0;
	if (b->resident && (rc = syncToPools(w, b)) != 0) // the pools changed under a resident world: its manifolds first
	{
		return rc;
	}
	if (leanStep && s_devicePairs && deviceTreesWanted() && s_rebuildTrees != NULL)
	{
		// (this step's stage 1 ran on the device's pairs and its stage 2 has not run anywhere: the host's trees -- current after
		// syncToPools -- get it now, src/world.c:125-130, so that they can be sent along clean)
		if (b->pendingStepCount > 0)
		{
			replayTrees(w, b); // (a world on the host replay: its log first)
		}
		s2Array_Clear(w->broadPhase.moveArray);
		s2ClearSet(&w->broadPhase.moveSet);
		s_rebuildTrees(&w->broadPhase);
	}
	int nb = w->bodyPool.capacity, ns = w->shapePool.capacity, nj = w->jointPool.capacity, nc = w->contactPool.capacity;
	b->bodies = (s2amdBody*)growTo(b->bodies, (size_t)nb, sizeof(s2amdBody));
	b->origins = (float*)growTo(b->origins, (size_t)nb * 2, sizeof(float));
	b->shapes = (s2amdShape*)growTo(b->shapes, (size_t)ns, sizeof(s2amdShape));
	b->boxes = (s2amdShapeBox*)growTo(b->boxes, (size_t)ns, sizeof(s2amdShapeBox));
	b->joints = (s2amdJoint*)growTo(b->joints, (size_t)nj, sizeof(s2amdJoint));
	b->contacts = (s2amdContact*)growTo(b->contacts, (size_t)nc, sizeof(s2amdContact));
	b->pairs = (s2amdPairState*)growTo(b->pairs, (size_t)nc, sizeof(s2amdPairState));
	b->separated = (int32_t*)growTo(b->separated, (size_t)nc, sizeof(int32_t));
	b->liveKey = (int64_t*)growTo(b->liveKey, (size_t)nc, sizeof(int64_t));
	b->slots = (int32_t*)growTo(b->slots, (size_t)nc, sizeof(int32_t));
	b->slotContacts = (s2amdContact*)growTo(b->slotContacts, (size_t)nc, sizeof(s2amdContact));
	b->slotPairs = (s2amdPairState*)growTo(b->slotPairs, (size_t)nc, sizeof(s2amdPairState));
	s2amdBinding_PackBodies(w, b->bodies);
	s2amdBinding_PackShapes(w, b->shapes);
	s2amdBinding_PackJoints(w, b->joints);
	s2amdBinding_PackContacts(w, b->contacts);
	s2amdBinding_PackPairs(w, b->pairs);
	for (int i = 0; i < nb; ++i)
	{
		b->origins[2 * i] = w->bodies[i].origin.x;
		b->origins[2 * i + 1] = w->bodies[i].origin.y;
	}
	b->liveCount = 0;
	b->createdCount = -1;
	for (int i = 0; i < nc; ++i)
	{
		b->liveKey[i] = b->pairs[i].shapeA < 0 ? -1 : ((int64_t)b->pairs[i].shapeA << 32) | (int64_t)b->pairs[i].shapeB;
		b->liveCount += b->liveKey[i] >= 0 ? 1 : 0;
	}
	// (option "prebuild_solver" would build the structure with the upload instead of in the first steps.  Not set here: the whole-step
	// route must sweep in the order the solver-only route sweeps in -- their digests are compared bit for bit, tests/test_dropin_product.py --
	// and that route, which sees the world one s2Solve_* at a time, builds its strips a step after the colour batches.  S2AMD_PREBUILD=1
	// turns it on for a caller that runs one route only.)
	if (getenv("S2AMD_PREBUILD") != NULL && atoi(getenv("S2AMD_PREBUILD")) != 0)
	{
		(void)s_api.setOption(b->solver, "prebuild_solver", (int32_t)w->solverType);
	}
	rc = s_api.worldUpload(b->solver, b->bodies, nb, b->contacts, nc, b->joints, nj, b->shapes, ns, b->pairs, b->origins);
	if (rc != 0)
	{
		b->resident = 0;
		return rc;
	}
	b->poses = (float*)growTo(b->poses, (size_t)nb * 4, sizeof(float));
	b->bodiesStale = b->boxesStale = 0;
	b->lastMoved = 1; // (a fresh world: everything is in the move buffer)
	if (b->poses == NULL || (rc = sendRefitOrder(w, b)) != 0)
	{
		b->resident = 0;
		return rc != 0 ? rc : S2AMD_E_DEVICE;
	}
	b->resident = 1;
	sendTrees(w, b);
	b->bodyCapacity = nb, b->bodyCount = w->bodyPool.count;
	b->shapeCapacity = ns, b->shapeCount = w->shapePool.count;
	b->jointCapacity = nj, b->jointCount = w->jointPool.count;
	b->contactCapacity = nc;
	b->contactsStale = 0;
	s_uploads += 1;
	return 0;
}

// contacts stage 1 created since the device last saw the pool (the pool never frees a slot on its own between steps:
// stage 3's separations are applied to both sides below)
static void describeNewContact(const s2Contact* c, s2amdContact* o, s2amdPairState* ps)
{
	memset(o, 0, sizeof(*o));
	memset(ps, 0, sizeof(*ps));
	o->bodyA = c->edges[0].bodyIndex;
	o->bodyB = c->edges[1].bodyIndex;
	o->friction = c->friction;
	o->constraintIndex = -1;
	ps->shapeA = c->shapeIndexA;
	ps->shapeB = c->shapeIndexB;
}

static int sendNewContacts(s2World* w, WorldBinding* b)
{
	int count = 0;
	// When stage 1 ran HERE (device pairs) the binding made every s2CreateContact call of the step itself and knows the slots;
	// the pool's live count says whether anything else touched the pool since: no scan of the pool then.
	if (b->createdCount >= 0 && w->contactPool.count == b->liveCount + b->createdCount)
	{
		for (int n = 0; n < b->createdCount; ++n)
		{
			const int i = b->createdSlots[n];
			const s2Contact* c = w->contacts + i;
			if (i < 0 || i >= b->contactCapacity || s2IsFree(&c->object) || b->liveKey[i] >= 0)
			{
				return 1;
			}
			describeNewContact(c, b->slotContacts + count, b->slotPairs + count);
			b->slots[count++] = i;
			b->liveKey[i] = ((int64_t)c->shapeIndexA << 32) | (int64_t)c->shapeIndexB;
		}
		b->liveCount += count;
		b->createdCount = -1;
		// (s2amd_world_set_contacts wants ascending slots no more than the scan below delivers them: it sorts)
		return count > 0 ? s_api.worldSetContacts(b->solver, b->slots, count, b->slotContacts, b->slotPairs) : 0;
	}
	b->createdCount = -1;
	for (int i = 0; i < b->contactCapacity; ++i)
	{
		const s2Contact* c = w->contacts + i;
		int64_t live = s2IsFree(&c->object) ? -1 : ((int64_t)c->shapeIndexA << 32) | (int64_t)c->shapeIndexB;
		if (live >= 0 && b->liveKey[i] < 0)
		{
			describeNewContact(c, b->slotContacts + count, b->slotPairs + count);
			b->slots[count++] = i;
			b->liveKey[i] = live;
		}
		else if (live != b->liveKey[i])
		{
			return 1; // somebody destroyed a contact behind our back (s2DestroyBody, s2CreateJoint ...): upload again
		}
	}
	b->liveCount += count;
	return count > 0 ? s_api.worldSetContacts(b->solver, b->slots, count, b->slotContacts, b->slotPairs) : 0;
}

// -- the reference's creation order for the pairs the device found (src/broad_phase.c:253-254, :288-320, :332-357) -------------
// s2UpdateBroadPhasePairs creates contacts proxy by proxy in move-array order; within one querying proxy in REVERSE callback
// order (every callback pushes its pair at the front of the proxy's list); callbacks come tree by tree (dynamic, kinematic,
// static: broad_phase.c:300-311) and within a tree in s2DynamicTree_Query's traversal order (src/dynamic_tree.c:1171-1210:
// a stack that takes child2 before child1).  Pruned subtrees do not change the relative order of the leaves that are
// reported, so the traversal order of two leaves is decided at their lowest common ancestor: the one under child2 first.
// The trees are the reference's own (stage 4 below keeps enlarging them, stage 2 rebuilds them every step as world.c:130
// does), so sorting the device's pair SET on (move index, tree, traversal rank) gives s2CreateContact the reference's
// sequence and every contact the reference's pool slot.
typedef struct OrderedPair
{
	int32_t shapeA, shapeB;
	int32_t moveIndex; // of the querying proxy
	int32_t treeOrder; // 0 dynamic, 1 kinematic, 2 static: the order FindPairs queries them in
	int32_t leaf; // the other proxy's node
	int32_t depth; // of that node
} OrderedPair;

static const s2DynamicTree* s_orderTrees; // qsort has no context argument; the binding is single-threaded like the reference

static int nodeDepth(const s2DynamicTree* tree, int32_t node)
{
	int depth = 0;
	while (tree->nodes[node].parent != S2_NULL_INDEX)
	{
		node = tree->nodes[node].parent;
		depth += 1;
	}
	return depth;
}

// < 0 when s2DynamicTree_Query reports leaf x before leaf y
static int traversalOrder(const s2DynamicTree* tree, int32_t x, int dx, int32_t y, int dy)
{
	if (x == y)
	{
		return 0;
	}
	const s2TreeNode* nodes = tree->nodes;
	while (dx > dy)
	{
		x = nodes[x].parent, dx -= 1;
	}
	while (dy > dx)
	{
		y = nodes[y].parent, dy -= 1;
	}
	while (nodes[x].parent != nodes[y].parent)
	{
		x = nodes[x].parent, y = nodes[y].parent;
	}
	return nodes[nodes[x].parent].child2 == x ? -1 : 1;
}

static int creationOrder(const void* pa, const void* pb)
{
	const OrderedPair* a = (const OrderedPair*)pa;
	const OrderedPair* b = (const OrderedPair*)pb;
	if (a->moveIndex != b->moveIndex)
	{
		return a->moveIndex < b->moveIndex ? -1 : 1;
	}
	if (a->treeOrder != b->treeOrder)
	{
		return a->treeOrder > b->treeOrder ? -1 : 1; // reverse callback order
	}
	static const int typeOfOrder[3] = {s2_dynamicBody, s2_kinematicBody, s2_staticBody};
	return -traversalOrder(s_orderTrees + typeOfOrder[a->treeOrder], a->leaf, a->depth, b->leaf, b->depth);
}

// pairs[2*i], pairs[2*i+1] = (shapeA, shapeB) as s2amd_world_find_pairs returns them (oriented, any order), reordered in place
// into the sequence s2UpdateBroadPhasePairs would create them in.  moveArray = bp->moveArray as stage 1 would see it.
void s2amdBinding_OrderPairs(s2World* world, const int* moveArray, int moveCount, int32_t* pairs, int32_t count)
{
	static int32_t* moveIndexOfShape; // -1 everywhere between calls
	static int moveIndexCapacity;
	static OrderedPair* ordered;
	static int orderedCapacity;
	s2BroadPhase* bp = &world->broadPhase;
	if (moveIndexCapacity < world->shapePool.capacity)
	{
		moveIndexCapacity = world->shapePool.capacity;
		moveIndexOfShape = (int32_t*)growTo(moveIndexOfShape, (size_t)moveIndexCapacity, sizeof(int32_t));
		memset(moveIndexOfShape, 0xff, (size_t)moveIndexCapacity * sizeof(int32_t));
	}
	if (orderedCapacity < count)
	{
		orderedCapacity = count + 1024;
		ordered = (OrderedPair*)growTo(ordered, (size_t)orderedCapacity, sizeof(OrderedPair));
	}
	for (int i = 0; i < moveCount; ++i)
	{
		if (moveArray[i] != S2_NULL_INDEX)
		{
			moveIndexOfShape[s2BroadPhase_GetShapeIndex(bp, moveArray[i])] = i;
		}
	}
	static const int orderOfType[s2_bodyTypeCount] = {2, 1, 0}; // s2_staticBody, s2_kinematicBody, s2_dynamicBody
	for (int i = 0; i < count; ++i)
	{
		OrderedPair* o = ordered + i;
		o->shapeA = pairs[2 * i], o->shapeB = pairs[2 * i + 1];
		const int keyA = world->shapes[o->shapeA].proxyKey, keyB = world->shapes[o->shapeB].proxyKey;
		const int moveA = moveIndexOfShape[o->shapeA], moveB = moveIndexOfShape[o->shapeB];
		// who asked (broad_phase.c:196-201): the one that moved; when both did, the one with the larger key (the other's
		// query skipped the pair).  keyA < keyB by the callback's orientation rule (:210-219).
		const int queryIsB = moveB >= 0 && (moveA < 0 || keyB > keyA);
		const int other = queryIsB ? keyA : keyB;
		o->moveIndex = queryIsB ? moveB : moveA;
		o->treeOrder = orderOfType[S2_PROXY_TYPE(other)];
		o->leaf = S2_PROXY_ID(other);
		o->depth = nodeDepth(bp->trees + S2_PROXY_TYPE(other), o->leaf);
	}
	s_orderTrees = bp->trees;
	qsort(ordered, (size_t)count, sizeof(OrderedPair), creationOrder);
	for (int i = 0; i < count; ++i)
	{
		pairs[2 * i] = ordered[i].shapeA, pairs[2 * i + 1] = ordered[i].shapeB;
	}
	for (int i = 0; i < moveCount; ++i)
	{
		if (moveArray[i] != S2_NULL_INDEX)
		{
			moveIndexOfShape[s2BroadPhase_GetShapeIndex(bp, moveArray[i])] = -1;
		}
	}
}

// == s2World_Step(worldId, timeStep, velIters, posIters, warmStart) (src/world.c:120-301).  Stages 1 and 2 are passed in:
// the reference's own s2UpdateBroadPhasePairs and s2BroadPhase_RebuildTrees (world.c:125-130).
void s2amdBinding_WorldStep(s2World* world, float timeStep, int velIters, int posIters, bool warmStart, void (*updatePairs)(s2World*),
							void (*rebuildTrees)(s2BroadPhase*))
{
	WorldBinding* b = bindingOf(world);
	if (b == NULL)
	{
		s_lastError = S2AMD_E_NODEVICE;
		return;
	}
	world->stepId += 1;
	s_rebuildTrees = rebuildTrees;
	const double t0 = wallMs();
	int rc = 0;
	const int lean = s_devicePairs && residentMatches(world, b);
	if (!lean && b->resident)
	{
		// the host's stage 1 (or a re-upload) is about to read trees, boxes and bodies: whatever the lean steps left on the device
		rc = syncBodiesAndBoxes(world, b);
	}
	if (rc == 0 && lean)
	{
		// stage 1 with the pair discovery on the device (s2amd_world_find_pairs on the boxes the last refit re-inflated
		// -- the proxies in the reference's move buffer); the pool bookkeeping of each new pair is s2CreateContact as ever,
		// called in the order the reference's own stage 1 would have found the pairs in (s2amdBinding_OrderPairs), so every
		// contact lands in the pool slot the host route gives it.  The trees keep following the fat boxes (below).
		s2BroadPhase* bp = &world->broadPhase;
		b->createdCount = 0; // stage 1 runs here: every contact it creates is recorded below
		if (b->lastMoved > 0 || s2Array(bp->moveArray).count > 0)
		{
			int32_t count = 0;
			rc = s_api.worldFindPairs(b->solver, b->newPairs, b->newPairCapacity, &count);
			if (rc == S2AMD_E_CAPACITY)
			{
				b->newPairCapacity = count + 1024;
				b->newPairs = (int32_t*)growTo(b->newPairs, (size_t)b->newPairCapacity * 2, sizeof(int32_t));
				rc = s_api.worldFindPairs(b->solver, b->newPairs, b->newPairCapacity, &count);
			}
			// new pairs: the device returns them in the reference's creation order when it holds the reference's trees (round 6:
			// s2amd_world_set_tree); otherwise -- and as the checker, S2AMD_CHECK_TREES -- the order is read off the host's trees and
			// move buffer, which the steps since the last pair have only logged: replayed now (in nearly every step of a settled world
			// there is nothing to create, and the trees are not touched at all)
			const int ordered = b->deviceTrees;
			if (rc == 0 && count > 0 && (!ordered || s_checkTrees))
			{
				replayTrees(world, b);
				if (ordered)
				{
					checkTrees(world, b);
				}
			}
			if (rc == 0 && count > 1 && !ordered)
			{
				s2amdBinding_OrderPairs(world, bp->moveArray, s2Array(bp->moveArray).count, b->newPairs, count);
			}
			else if (rc == 0 && count > 1 && s_checkTrees)
			{
				static int32_t* want;
				static int wantCapacity;
				if (wantCapacity < count)
				{
					wantCapacity = count + 1024;
					want = (int32_t*)growTo(want, (size_t)wantCapacity * 2, sizeof(int32_t));
				}
				memcpy(want, b->newPairs, (size_t)count * 2 * sizeof(int32_t));
				s2amdBinding_OrderPairs(world, bp->moveArray, s2Array(bp->moveArray).count, want, count);
				s_treeCheck[0] += 1;
				if (memcmp(want, b->newPairs, (size_t)count * 2 * sizeof(int32_t)) != 0)
				{
					s_treeCheck[1] += 1;
					if (getenv("S2AMD_CHECK_TREES_VERBOSE") != NULL)
					{
						for (int i = 0; i < count; ++i)
						{
							fprintf(stderr, "pair %d: device (%d, %d) reference (%d, %d)\n", i, b->newPairs[2 * i], b->newPairs[2 * i + 1], want[2 * i], want[2 * i + 1]);
						}
					}
				}
			}
			if (b->createdCapacity < count)
			{
				b->createdCapacity = count + 1024;
				b->createdSlots = (int32_t*)growTo(b->createdSlots, (size_t)b->createdCapacity, sizeof(int32_t));
			}
			b->createdCount = 0;
			for (int i = 0; rc == 0 && i < count; ++i)
			{
				const s2Shape* shapeA = world->shapes + b->newPairs[2 * i];
				const int before = world->contactPool.count;
				s2CreateContact(world, world->shapes + b->newPairs[2 * i], world->shapes + b->newPairs[2 * i + 1]);
				if (world->contactPool.count == before + 1)
				{
					// the new contact heads the contact list of both its bodies (src/contact.c:189-221), flipped or not
					b->createdSlots[b->createdCount++] = world->bodies[shapeA->bodyIndex].contactList >> 1;
				}
			}
			// (the move buffer is cleared and the trees rebuilt -- stage 2, world.c:130 -- when this step's boxes are replayed:
			// flushTrees)
		}
	}
	else
	{
		// stages 1 and 2 (src/world.c:125-130): the reference's trees, the reference's contact pool
		b->createdCount = -1; // (which contacts it creates is found by scanning the pool)
		updatePairs(world);
		rebuildTrees(&world->broadPhase);
	}
	const double t1 = wallMs();

	if (rc == 0 && (!residentMatches(world, b) || (rc = sendNewContacts(world, b)) == 1))
	{
		rc = uploadWorld(world, b, lean);
	}
	s2amdStepParams params;
	fillParams(world, &params, (int32_t)world->solverType, timeStep, velIters, posIters, warmStart ? 1 : 0);
	s2amdWorldStepInfo info = {0};
	const double t2 = wallMs();
	if (rc == 0)
	{
		rc = s_api.worldStep(b->solver, &params, &info);
	}
	const double t3 = wallMs();
	int32_t separatedCount = 0;
	int32_t movedCount = 0;
	const int leanBack = lean && rc == 0; // (a step that had to upload the world reads everything back once, like the host-pairs route)
	if (leanBack)
	{
		if (b->deviceTrees && !s_checkTrees)
		{
			// (the device keeps the trees: the log holds the last step's boxes only -- the move buffer, should the host's trees be asked for)
			b->pendingBoxCount = 0;
			b->pendingStepCount = 0;
			b->treesStale = 1;
		}
		rc = reservePending(b, info.movedCount); // (the step's boxes are written straight into the log)
		if (rc == 0)
		{
			rc = s_api.worldDownloadStep(b->solver, b->poses, b->bodyCapacity, b->pendingBoxes + b->pendingBoxCount, b->pendingBoxCapacity - b->pendingBoxCount,
										 &movedCount);
		}
	}
	else
	{
		if (rc == 0)
		{
			rc = s_api.worldDownload(b->solver, b->bodies, b->bodyCapacity, NULL, b->contactCapacity, NULL, b->jointCapacity, NULL, b->shapeCapacity, NULL,
									 b->origins, NULL);
		}
		if (rc == 0 && info.movedCount > 0)
		{
			rc = s_api.worldDownloadBoxes(b->solver, b->boxes, b->shapeCapacity); // 36 bytes per shape instead of the 196-byte records
		}
	}
	if (rc == 0 && info.separatedCount > 0)
	{
		rc = s_api.worldSeparated(b->solver, b->separated, b->contactCapacity, &separatedCount);
	}
	if (rc != 0)
	{
		fprintf(stderr, "s2World_Step on the GPU failed (%d): %s\n", rc, s_api.lastError());
		s_lastError = rc;
		b->resident = 0;
		return;
	}
	const double t4 = wallMs();
	b->contactsStale = 1;
	s_steps += 1;
	b->stepId = (uint64_t)world->stepId;
	b->lastMoved = info.movedCount;
	if (leanBack)
	{
		// what the public API reads of a body (src/body.c:316-340: origin, rot) and what the step consumed (forces, world.c:274-275)
		for (int i = 0; i < b->bodyCapacity; ++i)
		{
			s2Body* body = world->bodies + i;
			if (s2IsFree(&body->object) || body->type == s2_staticBody)
			{
				continue;
			}
			const float* p = b->poses + 4 * i;
			body->origin = (s2Vec2){p[0], p[1]};
			body->rot = (s2Rot){p[2], p[3]};
			body->force = s2Vec2_zero;
			body->torque = 0.0f;
		}
		b->bodiesStale = 1;
		b->boxesStale = 1;
		// the step's boxes went straight into the log
		b->pendingBoxCount += movedCount;
		b->pendingSteps[b->pendingStepCount++] = movedCount;
	}
	else
	{
		s2amdBinding_UnpackBodies(world, b->bodies);
		for (int i = 0; i < b->bodyCapacity; ++i)
		{
			s2Body* body = world->bodies + i;
			if (s2IsFree(&body->object) || body->type == s2_staticBody)
			{
				continue;
			}
			body->origin = (s2Vec2){b->origins[2 * i], b->origins[2 * i + 1]};
			body->force = s2Vec2_zero;
			body->torque = 0.0f;
		}
	}
	// src/world.c:163-167: the pairs stage 3 found separated
	for (int i = 0; i < separatedCount; ++i)
	{
		const int slot = b->separated[i];
		s2DestroyContact(world, world->contacts + slot);
		b->liveCount -= b->liveKey[slot] >= 0 ? 1 : 0;
		b->liveKey[slot] = -1;
	}
	if (leanBack && (b->pendingStepCount >= S2AMD_BINDING_MAX_PENDING_STEPS || b->pendingBoxCount >= S2AMD_BINDING_MAX_PENDING_BOXES))
	{
		flushTrees(world, b); // (bounds the log; the work is what the per-step path would have done by now)
	}
	if (!leanBack && info.movedCount > 0)
	{
		// src/world.c:259-297: the tight boxes of every shape, the tree only where the fat box was re-inflated -- in the
		// reference's order (bodies, then each body's shape list): the move buffer's order decides the pool slots of the
		// contacts stage 1 creates next step
		for (int bi = 0; bi < b->bodyCapacity; ++bi)
		{
			const s2Body* body = world->bodies + bi;
			if (s2IsFree(&body->object) || body->type == s2_staticBody)
			{
				continue;
			}
			for (int i = body->shapeList; i != S2_NULL_INDEX; i = world->shapes[i].nextShapeIndex)
			{
				s2Shape* sh = world->shapes + i;
				const s2amdShapeBox* o = b->boxes + i;
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
	s_phaseMs[0] += t1 - t0, s_phaseMs[1] += t2 - t1, s_phaseMs[2] += t3 - t2, s_phaseMs[3] += t4 - t3, s_phaseMs[4] += t5 - t4, s_phaseMs[5] += 1.0;
}
