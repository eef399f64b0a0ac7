/* s2_amd_dropin.c -- the call sites of INTEGRATION.md section 1, supplied at LINK time: compiled together with an unmodified
 * checkout of erincatto/solver2d and shim/s2_amd_binding.c by shim/Makefile into libsolver2d_amd.so, a library that exports
 * the reference's whole public API (include/solver2d/solver2d.h:22-70 and the geometry / hull / distance / tree helpers) and
 * whose s2World_Step runs on the MI355X.  Programs written against the public headers link against it unchanged.
 *
 * No reference source file is edited or copied.  Two kinds of call site:
 *   s2Solve_<Variant>(world, context)   (src/solvers.h:70-79) is called by the reference itself (the switch in src/world.c:206-256):
 *                                       the linker's --wrap reroutes that call -> s2amdBinding_Solve, mode "solver";
 *   the PUBLIC functions a program calls -- s2World_Step, s2DestroyWorld, and every function that creates or destroys a body or
 *                                       joint, sets or reads state a resident world keeps in HBM (velocities, forces, joint settings;
 *                                       manifolds, joint impulses) -- are THIS file's: shim/Makefile renames the reference's
 *                                       definitions in the compiled objects (objcopy; each stays reachable as <name>_reference), exactly
 *                                       what a maintainer's edit of the function's first line would do.  (--wrap alone would not do:
 *                                       it reroutes references inside the link, and a program's calls come from outside it.)
 *                                       They bring the host pools up to date (s2amdBinding_Sync) and, when the call edits the pools,
 *                                       mark the resident copy stale (s2amdBinding_Invalidate) before the reference's function runs.
 *
 * Run-time switches (environment, read at the first step):
 *   S2AMD_DROPIN   step (default): stage 3, the solve and stage 4 on the device, stage 1's pair query too (S2AMD_DEVICE_PAIRS=0:
 *                  pair query on the host's trees); solver: only s2Solve_* on the device; off: the reference as it is.
 *   S2AMD_LIBRARY  path of libs2amd.so (default: "libs2amd.so" beside this library, then the loader's search path)
 *   S2AMD_DEVICE   HIP device ordinal (default 0)
 * There is no CPU fallback: when the library or the GPU is missing the first step says so on stderr and aborts. */
#define _GNU_SOURCE
#include "s2_amd_binding.h"

#include "solver2d/solver2d.h"
#include "solvers.h"
#include "world.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum
{
	MODE_UNSET = -1,
	MODE_OFF,
	MODE_SOLVER,
	MODE_STEP
};
static int s_mode = MODE_UNSET;

static void openOnce(void)
{
	if (s_mode != MODE_UNSET)
	{
		return;
	}
	const char* m = getenv("S2AMD_DROPIN");
	s_mode = (m && strcmp(m, "off") == 0) ? MODE_OFF : (m && strcmp(m, "solver") == 0) ? MODE_SOLVER : MODE_STEP;
	if (s_mode == MODE_OFF)
	{
		return;
	}
	char beside[4096];
	const char* path = getenv("S2AMD_LIBRARY");
	if (path == NULL)
	{
		path = "libs2amd.so";
		Dl_info info;
		if (dladdr((void*)&openOnce, &info) && info.dli_fname)
		{
			const char* slash = strrchr(info.dli_fname, '/');
			if (slash && (size_t)(slash - info.dli_fname) + 16 < sizeof(beside))
			{
				size_t n = (size_t)(slash - info.dli_fname) + 1;
				memcpy(beside, info.dli_fname, n);
				strcpy(beside + n, "libs2amd.so");
				FILE* f = fopen(beside, "rb");
				if (f)
				{
					fclose(f);
					path = beside;
				}
			}
		}
	}
	const char* dev = getenv("S2AMD_DEVICE");
	int rc = s2amdBinding_Open(path, dev ? atoi(dev) : 0);
	if (rc != 0)
	{
		fprintf(stderr, "libsolver2d_amd: cannot run on the GPU (s2amdBinding_Open(\"%s\") = %d: -1 library not found, -2 symbol missing, -3 no GPU, "
						"-4 API version).  There is no CPU path in this build; set S2AMD_DROPIN=off for the reference's own solvers.\n",
				path, rc);
		abort();
	}
	const char* pairs = getenv("S2AMD_DEVICE_PAIRS");
	s2amdBinding_DevicePairs(pairs ? atoi(pairs) != 0 : 1);
}

/* ---- src/world.c:120-301 ---- */
void s2World_Step_reference(s2WorldId worldId, float timeStep, int velIters, int posIters, bool warmStart);
void s2World_Step(s2WorldId worldId, float timeStep, int32_t velIters, int32_t posIters, bool warmStart)
{
	openOnce();
	if (s_mode != MODE_STEP)
	{
		s2World_Step_reference(worldId, timeStep, velIters, posIters, warmStart); /* (mode "solver": its switch reaches the wraps below) */
		return;
	}
	s2amdBinding_WorldStep(s2GetWorldFromId(worldId), timeStep, velIters, posIters, warmStart, s2UpdateBroadPhasePairs, s2BroadPhase_RebuildTrees);
	if (s2amdBinding_LastError() != 0)
	{
		fprintf(stderr, "libsolver2d_amd: s2World_Step failed on the device (error %d)\n", s2amdBinding_LastError());
		abort();
	}
}

/* ---- src/solvers.h:70-79 ---- */
#define S2_DROPIN_SOLVER(NAME, TYPE)                                                                                             \
	void __real_##NAME(s2World* world, s2StepContext* context);                                                                  \
	void __wrap_##NAME(s2World* world, s2StepContext* context)                                                                   \
	{                                                                                                                            \
		if (s_mode != MODE_SOLVER)                                                                                               \
		{                                                                                                                        \
			__real_##NAME(world, context);                                                                                       \
			return;                                                                                                              \
		}                                                                                                                        \
		int rc = s2amdBinding_Solve(world, context, TYPE);                                                                       \
		if (rc != 0)                                                                                                             \
		{                                                                                                                        \
			fprintf(stderr, "libsolver2d_amd: " #NAME " failed on the device (error %d)\n", rc);                                 \
			abort();                                                                                                             \
		}                                                                                                                        \
	}
S2_DROPIN_SOLVER(s2Solve_Jacobi, s2_solverJacobi)
S2_DROPIN_SOLVER(s2Solve_PGS, s2_solverPGS)
S2_DROPIN_SOLVER(s2Solve_PGS_NGS, s2_solverPGS_NGS)
S2_DROPIN_SOLVER(s2Solve_PGS_NGS_Block, s2_solverPGS_NGS_Block)
S2_DROPIN_SOLVER(s2Solve_PGS_Soft, s2_solverPGS_Soft)
S2_DROPIN_SOLVER(s2Solve_SoftStep, s2_solverSoftStep)
S2_DROPIN_SOLVER(s2Solve_TGS_Sticky, s2_solverTGS_Sticky)
S2_DROPIN_SOLVER(s2Solve_TGS_Soft, s2_solverTGS_Soft)
S2_DROPIN_SOLVER(s2Solve_TGS_NGS, s2_solverTGS_NGS)
S2_DROPIN_SOLVER(s2Solve_XPBD, s2_solverXPBD)

/* ---- src/world.c:105-118 ---- */
void s2DestroyWorld_reference(s2WorldId id);
void s2DestroyWorld(s2WorldId id)
{
	if (s2amdBinding_IsOpen())
	{
		s2amdBinding_DestroyWorld(s2GetWorldFromId(id));
	}
	s2DestroyWorld_reference(id);
}

/* ---- readers of what a resident world keeps on the device: manifolds (drawn by s2World_Draw, src/world.c:369-563), joint impulses ---- */
static void syncWorld(s2World* world)
{
	if (s_mode == MODE_STEP && s2amdBinding_IsOpen())
	{
		s2amdBinding_Sync(world);
	}
}
static void editWorld(s2World* world)
{
	if (s_mode == MODE_STEP && s2amdBinding_IsOpen())
	{
		s2amdBinding_Sync(world); /* the edit lands on current host pools ... */
		s2amdBinding_Invalidate(world); /* ... and the next step uploads them */
	}
}
void s2World_Draw_reference(s2WorldId worldId, s2DebugDraw* debugDraw);
void s2World_Draw(s2WorldId worldId, s2DebugDraw* debugDraw)
{
	syncWorld(s2GetWorldFromId(worldId));
	s2World_Draw_reference(worldId, debugDraw);
}
float s2RevoluteJoint_GetMotorTorque_reference(s2JointId jointId, float inverseTimeStep);
float s2RevoluteJoint_GetMotorTorque(s2JointId jointId, float inverseTimeStep)
{
	syncWorld(s2GetWorldFromIndex(jointId.world));
	return s2RevoluteJoint_GetMotorTorque_reference(jointId, inverseTimeStep);
}

/* ---- calls that read or edit the broad-phase trees and the pools: the steps since the last pair only logged their tree work
 * (s2_amd_binding.c: lean read-back), it is done now, before the reference's code touches the trees ---- */
void s2World_QueryAABB_reference(s2WorldId worldId, s2Box aabb, s2QueryCallbackFcn* fcn, void* context);
void s2World_QueryAABB(s2WorldId worldId, s2Box aabb, s2QueryCallbackFcn* fcn, void* context)
{
	syncWorld(s2GetWorldFromId(worldId));
	s2World_QueryAABB_reference(worldId, aabb, fcn, context);
}
/* ---- pool edits.  A destroy followed by a create leaves every pool count as it was (a mouse joint released and grabbed again, a
 * body replaced), so the counts residentMatches compares cannot see them: the edit itself marks the resident world stale.  The sync
 * comes first: a lean step leaves the bodies on the device, and a slot handed out now must not be overwritten later by the download
 * of what the device still holds there as a free slot. ---- */
void s2DestroyBody_reference(s2BodyId bodyId);
void s2DestroyBody(s2BodyId bodyId)
{
	editWorld(s2GetWorldFromIndex(bodyId.world));
	s2DestroyBody_reference(bodyId);
}
s2BodyId s2CreateBody_reference(s2WorldId worldId, const s2BodyDef* def);
s2BodyId s2CreateBody(s2WorldId worldId, const s2BodyDef* def)
{
	editWorld(s2GetWorldFromId(worldId));
	return s2CreateBody_reference(worldId, def);
}
void s2DestroyJoint_reference(s2JointId jointId);
void s2DestroyJoint(s2JointId jointId)
{
	editWorld(s2GetWorldFromIndex(jointId.world));
	s2DestroyJoint_reference(jointId);
}
s2JointId s2CreateMouseJoint_reference(s2WorldId worldId, const s2MouseJointDef* def);
s2JointId s2CreateMouseJoint(s2WorldId worldId, const s2MouseJointDef* def)
{
	editWorld(s2GetWorldFromId(worldId));
	return s2CreateMouseJoint_reference(worldId, def);
}
s2JointId s2CreateRevoluteJoint_reference(s2WorldId worldId, const s2RevoluteJointDef* def);
s2JointId s2CreateRevoluteJoint(s2WorldId worldId, const s2RevoluteJointDef* def)
{
	editWorld(s2GetWorldFromId(worldId));
	return s2CreateRevoluteJoint_reference(worldId, def);
}
#define S2_DROPIN_SHAPE(NAME, GEOM)                                                                                              \
	s2ShapeId NAME##_reference(s2BodyId bodyId, const s2ShapeDef* def, const GEOM* geometry);                                    \
	s2ShapeId NAME(s2BodyId bodyId, const s2ShapeDef* def, const GEOM* geometry)                                                 \
	{                                                                                                                            \
		syncWorld(s2GetWorldFromIndex(bodyId.world));                                                                            \
		return NAME##_reference(bodyId, def, geometry);                                                                          \
	}
S2_DROPIN_SHAPE(s2CreateCircleShape, s2Circle)
S2_DROPIN_SHAPE(s2CreateSegmentShape, s2Segment)
S2_DROPIN_SHAPE(s2CreateCapsuleShape, s2Capsule)
S2_DROPIN_SHAPE(s2CreatePolygonShape, s2Polygon)

/* ---- setters: the host copy changes, the resident copy has to follow ---- */
#define S2_DROPIN_EDIT(RET, NAME, ID_T, PARAMS, ARGS)                                                                            \
	RET NAME##_reference PARAMS;                                                                                                 \
	RET NAME PARAMS                                                                                                              \
	{                                                                                                                            \
		editWorld(s2GetWorldFromIndex(id.world));                                                                                \
		NAME##_reference ARGS;                                                                                                   \
	}
S2_DROPIN_EDIT(void, s2Body_SetLinearVelocity, s2BodyId, (s2BodyId id, s2Vec2 v), (id, v))
S2_DROPIN_EDIT(void, s2Body_SetAngularVelocity, s2BodyId, (s2BodyId id, float w), (id, w))
S2_DROPIN_EDIT(void, s2Body_ApplyForceToCenter, s2BodyId, (s2BodyId id, s2Vec2 f), (id, f))
S2_DROPIN_EDIT(void, s2Body_ApplyLinearImpulse, s2BodyId, (s2BodyId id, s2Vec2 impulse, s2Vec2 point), (id, impulse, point))
S2_DROPIN_EDIT(void, s2MouseJoint_SetTarget, s2JointId, (s2JointId id, s2Vec2 target), (id, target))
S2_DROPIN_EDIT(void, s2RevoluteJoint_EnableLimit, s2JointId, (s2JointId id, bool on), (id, on))
S2_DROPIN_EDIT(void, s2RevoluteJoint_EnableMotor, s2JointId, (s2JointId id, bool on), (id, on))
S2_DROPIN_EDIT(void, s2RevoluteJoint_SetMotorSpeed, s2JointId, (s2JointId id, float speed), (id, speed))

/* what the demo / a profiler may ask: accumulated ms per phase of the whole-step binding, see s2_amd_binding.h */
void s2amdDropin_Timing(double out[6])
{
	s2amdBinding_Timing(out);
}

/* a digest of every live body's position and rotation bits, pool order (tools/dropin_product_demo.c compares routes with it) */
#include "body.h"
#include "pool.h"
unsigned long long s2amdDropin_StateDigest(s2WorldId worldId)
{
	s2World* world = s2GetWorldFromId(worldId);
	syncWorld(world);
	unsigned long long h = 1469598103934665603ull;
	for (int i = 0; i < world->bodyPool.capacity; ++i)
	{
		const s2Body* b = world->bodies + i;
		if (s2ObjectValid(&b->object) == false)
		{
			continue;
		}
		const float v[4] = {b->position.x, b->position.y, b->rot.s, b->rot.c};
		unsigned int w[4];
		memcpy(w, v, sizeof(w));
		for (int k = 0; k < 4; ++k)
		{
			h = (h ^ w[k]) * 1099511628211ull;
		}
	}
	return h;
}
