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
//           HIP solver, which is the literal drop-in test.  s2ref_use_amd() routes the ten solvers to the PRODUCT binding
//           instead (shim/s2_amd_binding.c: s2amdBinding_Solve -- the file a maintainer adds to the reference, compiled
//           into this library like the reference's own sources), so that any program calling the PUBLIC s2World_Step of
//           this library (the samples' only entry point) runs on the GPU.
//   whole step (s2ref_use_amd_world): this library's exported s2World_Step -- oracle/Makefile renames the reference's in
//           the compiled world.o -- calls the binding's s2amdBinding_WorldStep: stages 1 and 2 (trees, contact pool) stay
//           the reference's, stage 3, the solve and stage 4 run as one s2amd_world_step on the world's resident chain.
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
#include "s2_amd_binding.h"

// the gather / scatter between pools and wire structs is the binding's (shim/s2_amd_binding.c); the capture hooks use it too
#define packBodies s2amdBinding_PackBodies
#define unpackBodies s2amdBinding_UnpackBodies
#define packContacts s2amdBinding_PackContacts
#define unpackContacts s2amdBinding_UnpackContacts
#define packJoints s2amdBinding_PackJoints
#define unpackJoints s2amdBinding_UnpackJoints
#define packShapes s2amdBinding_PackShapes
#define packPairs s2amdBinding_PackPairs

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
static int g_useBinding = 0; // mode 2 without a callback: the ten s2Solve_* go to the product binding (s2amdBinding_Solve)
static int g_wholeStep = 0;	 // the exported s2World_Step goes to s2amdBinding_WorldStep
static s2refReplaceFcn* g_replace = NULL;
static void* g_replaceUser = NULL;
static Snapshot g_pre = {0}, g_post = {0};
static s2amdStepParams g_params;
static int g_captureCount = 0;
static int g_replaceError = 0;
static double g_solveSeconds = 0.0;
static long g_solveCalls = 0;
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

// a world the tests keep on the reference's own code while the binding drives the others (s2ref_plain_world): the checker next to
// the thing checked, stepped in turn
static int g_plainWorld = -1;

static void hookSolve(s2World* world, s2StepContext* context, int solverType, s2SolveFcn* real)
{
	narrowPhaseDone(world);
	if (world->index == g_plainWorld && g_mode == 2)
	{
		real(world, context);
		return;
	}
	if (g_mode == 1)
	{
		fillParams(world, context, solverType);
		snapshotWorld(world, &g_pre);
		real(world, context);
		snapshotWorld(world, &g_post);
		g_captureCount += 1;
	}
	else if (g_mode == 2 && g_replace == NULL && g_useBinding)
	{
		// the product binding: == s2Solve_<Variant>(world, context)
		int rc = s2amdBinding_Solve(world, context, solverType);
		if (rc != 0)
		{
			g_replaceError = rc;
		}
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

// -- creation-order check (CPU only): the sequence the reference's stage 1 really calls s2CreateContact in, against
// s2amdBinding_OrderPairs applied to the same pairs handed over as a sorted set (what s2amd_world_find_pairs returns)
static int g_orderCheck, g_orderRecording, g_orderCount, g_orderCapacity;
static int32_t* g_orderSeen;
static long g_orderChecked, g_orderMismatched, g_orderSteps;

void __real_s2CreateContact(s2World* world, s2Shape* shapeA, s2Shape* shapeB);
void __wrap_s2CreateContact(s2World* world, s2Shape* shapeA, s2Shape* shapeB)
{
	if (g_orderRecording)
	{
		if (g_orderCount == g_orderCapacity)
		{
			g_orderCapacity = 2 * g_orderCapacity + 256;
			g_orderSeen = (int32_t*)realloc(g_orderSeen, (size_t)g_orderCapacity * 2 * sizeof(int32_t));
		}
		g_orderSeen[2 * g_orderCount] = shapeA->object.index;
		g_orderSeen[2 * g_orderCount + 1] = shapeB->object.index;
		g_orderCount += 1;
	}
	__real_s2CreateContact(world, shapeA, shapeB);
}

static int pairAscending(const void* a, const void* b)
{
	const int32_t* x = (const int32_t*)a;
	const int32_t* y = (const int32_t*)b;
	return x[0] != y[0] ? (x[0] < y[0] ? -1 : 1) : (x[1] < y[1] ? -1 : (x[1] > y[1] ? 1 : 0));
}

void __real_s2UpdateBroadPhasePairs(s2World* world);
static void updatePairsChecked(s2World* world)
{
	s2BroadPhase* bp = &world->broadPhase;
	const int moveCount = s2Array(bp->moveArray).count;
	int* moves = (int*)malloc((size_t)(moveCount > 0 ? moveCount : 1) * sizeof(int));
	memcpy(moves, bp->moveArray, (size_t)moveCount * sizeof(int));
	g_orderCount = 0;
	g_orderRecording = 1;
	__real_s2UpdateBroadPhasePairs(world);
	g_orderRecording = 0;
	if (g_orderCount > 0)
	{
		int32_t* sorted = (int32_t*)malloc((size_t)g_orderCount * 2 * sizeof(int32_t));
		memcpy(sorted, g_orderSeen, (size_t)g_orderCount * 2 * sizeof(int32_t));
		qsort(sorted, (size_t)g_orderCount, 2 * sizeof(int32_t), pairAscending);
		s2amdBinding_OrderPairs(world, moves, moveCount, sorted, g_orderCount);
		g_orderMismatched += memcmp(sorted, g_orderSeen, (size_t)g_orderCount * 2 * sizeof(int32_t)) != 0;
		g_orderChecked += g_orderCount;
		g_orderSteps += 1;
		free(sorted);
	}
	free(moves);
}

void s2ref_order_check(int on)
{
	g_orderCheck = on;
	g_orderChecked = g_orderMismatched = g_orderSteps = 0;
}

// pairs whose order was checked, steps (with new pairs) whose sequence differed, steps with new pairs
void s2ref_order_check_result(long out[3])
{
	out[0] = g_orderChecked, out[1] = g_orderMismatched, out[2] = g_orderSteps;
}

void __wrap_s2UpdateBroadPhasePairs(s2World* world)
{
	g_stepWorld = world;
	if (g_orderCheck)
	{
		updatePairsChecked(world);
		return;
	}
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

// ---- the product binding (shim/s2_amd_binding.c), switched on and off for the tests ----
// The binding itself -- gather / scatter, s2Solve_* -> s2amd_solve, the whole-step s2World_Step, one device state per world
// -- is NOT test infrastructure and does not live here.  This file only decides when the reference's plug points call it.

// path == NULL: back to the reference's own solvers.  Returns 0, or a negative number naming the step that failed.
S2REF_API int s2ref_use_amd(const char* path, int device)
{
	s2amdBinding_Close(); // manifolds of resident worlds back into their pools, device state released
	g_useBinding = 0;
	g_wholeStep = 0;
	if (g_mode == 2 && g_replace == NULL)
	{
		g_mode = 0;
	}
	if (path == NULL)
	{
		return 0;
	}
	int rc = s2amdBinding_Open(path, device);
	if (rc != 0)
	{
		return rc;
	}
	g_replace = NULL;
	g_useBinding = 1;
	g_mode = 2;
	g_replaceError = 0;
	return 0;
}

// index of the world that stays on the reference's own s2World_Step and solvers whatever the switches above say (-1: none)
S2REF_API void s2ref_plain_world(int index)
{
	g_plainWorld = index;
}

// As s2ref_use_amd, plus stage 3 and stage 4: the whole of s2World_Step but its tree and pool bookkeeping on the GPU.
S2REF_API int s2ref_use_amd_world(const char* path, int device)
{
	int rc = s2ref_use_amd(path, device);
	if (rc != 0 || path == NULL)
	{
		return rc;
	}
	g_wholeStep = 1;
	return 0;
}

void __real_s2UpdateBroadPhasePairs(s2World* world);
// The reference's own s2World_Step: oracle/Makefile renames the symbol in the compiled world.o (objcopy, the source is
// untouched) so that THIS library's exported s2World_Step is the function below -- programs linked against the
// public API reach the binding without knowing.  (A maintainer would instead put the same three lines at the top of
// src/world.c: s2World_Step -- INTEGRATION.md.)
void s2ref_World_Step_reference(s2WorldId worldId, float timeStep, int velIters, int posIters, bool warmStart);
S2REF_API void s2World_Step(s2WorldId worldId, float timeStep, int velIters, int posIters, bool warmStart)
{
	if (!g_wholeStep || worldId.index == g_plainWorld)
	{
		s2ref_World_Step_reference(worldId, timeStep, velIters, posIters, warmStart);
		return;
	}
	s2amdBinding_WorldStep(s2GetWorldFromId(worldId), timeStep, velIters, posIters, warmStart, __real_s2UpdateBroadPhasePairs, s2BroadPhase_RebuildTrees);
	if (s2amdBinding_LastError() != 0)
	{
		g_replaceError = s2amdBinding_LastError();
	}
}

// the public calls that read or edit the trees and pools of a world the binding keeps resident: its deferred tree work first
// (shim/s2_amd_binding.c: lean read-back; the product's call sites are shim/s2_amd_dropin.c)
void __real_s2World_QueryAABB(s2WorldId worldId, s2Box aabb, s2QueryCallbackFcn* fcn, void* context);
void __wrap_s2World_QueryAABB(s2WorldId worldId, s2Box aabb, s2QueryCallbackFcn* fcn, void* context)
{
	(void)s2amdBinding_Sync(s2GetWorldFromId(worldId));
	__real_s2World_QueryAABB(worldId, aabb, fcn, context);
}
void __real_s2DestroyBody(s2BodyId bodyId);
void __wrap_s2DestroyBody(s2BodyId bodyId)
{
	(void)s2amdBinding_Sync(s2GetWorldFromIndex(bodyId.world));
	__real_s2DestroyBody(bodyId);
}
// (an applied force is an edit of the host's body the resident copy has to follow: the product drop-in wraps every setter,
// shim/s2_amd_dropin.c: S2_DROPIN_EDIT; here the one the Rush sample uses)
void __real_s2Body_ApplyForceToCenter(s2BodyId bodyId, s2Vec2 force);
void __wrap_s2Body_ApplyForceToCenter(s2BodyId bodyId, s2Vec2 force)
{
	s2amdBinding_Invalidate(s2GetWorldFromIndex(bodyId.world));
	__real_s2Body_ApplyForceToCenter(bodyId, force);
}
#define S2REF_SHAPE_WRAP(NAME, GEOM)                                                                                             \
	s2ShapeId __real_##NAME(s2BodyId bodyId, const s2ShapeDef* def, const GEOM* geometry);                                       \
	s2ShapeId __wrap_##NAME(s2BodyId bodyId, const s2ShapeDef* def, const GEOM* geometry)                                        \
	{                                                                                                                            \
		(void)s2amdBinding_Sync(s2GetWorldFromIndex(bodyId.world));                                                              \
		return __real_##NAME(bodyId, def, geometry);                                                                             \
	}
S2REF_SHAPE_WRAP(s2CreateCircleShape, s2Circle)
S2REF_SHAPE_WRAP(s2CreateSegmentShape, s2Segment)
S2REF_SHAPE_WRAP(s2CreateCapsuleShape, s2Capsule)
S2REF_SHAPE_WRAP(s2CreatePolygonShape, s2Polygon)

// s2DestroyWorld (src/world.c:105-118) frees the world's device state first (oracle/Makefile: --wrap)
void __real_s2DestroyWorld(s2WorldId id);
void __wrap_s2DestroyWorld(s2WorldId id)
{
	s2amdBinding_DestroyWorld(s2GetWorldFromId(id));
	__real_s2DestroyWorld(id);
}

// Accumulated wall time of the whole-step binding's phases since the last call: stage 1 + 2 on the host, new contacts /
// upload, s2amd_world_step, download, applying the results to the pools and trees; [5] = steps.
S2REF_API void s2ref_world_timing(double out[6])
{
	s2amdBinding_Timing(out);
}

// The host pools of `id` brought up to date with the device (manifolds, caches, joint impulses).
S2REF_API int s2ref_world_sync(s2WorldId id)
{
	return s2amdBinding_Sync(s2GetWorldFromId(id));
}

// After editing resident worlds through the reference's API (velocities, forces, filters, joints' settings ...):
// their next step uploads them again.  Creating or destroying bodies, shapes and joints is noticed without this.
S2REF_API void s2ref_world_invalidate(void)
{
	for (int16_t i = 0; i < s2_maxWorlds; ++i)
	{
		s2World* w = s2GetWorldFromIndex(i);
		if (w->blockAllocator != NULL)
		{
			s2amdBinding_Invalidate(w);
		}
	}
}

// on: stage 1's pair discovery on the device as well (the host trees are still kept up to date, not queried)
S2REF_API void s2ref_world_device_pairs(int on)
{
	s2amdBinding_DevicePairs(on);
}

S2REF_API long s2ref_world_uploads(void)
{
	return s2amdBinding_Uploads();
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
	(void)s2amdBinding_Sync(world);
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
