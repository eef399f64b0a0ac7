/* s2_amd_binding.h -- what the reference's world.c / solvers see of the MI355X binding (shim/s2_amd_binding.c). */
#ifndef S2_AMD_BINDING_H
#define S2_AMD_BINDING_H

#include "solver2d_amd.h"

#include <stdbool.h>

typedef struct s2World s2World;
typedef struct s2StepContext s2StepContext;
typedef struct s2BroadPhase s2BroadPhase;

/* Loads libs2amd.so (dlopen) and checks that device `device` can run it.  0, or a negative code: -1 library not found,
 * -2 symbol missing, -3 no GPU.  Worlds get their device state on their first step. */
int s2amdBinding_Open(const char* libraryPath, int device);
int s2amdBinding_IsOpen(void);
/* Manifolds of every resident world back into its pools, all device state released. */
void s2amdBinding_Close(void);

/* == s2Solve_<solverType>(world, context) (src/solvers.h:70-79).  0 or an S2AMD_E_* code. */
int s2amdBinding_Solve(s2World* world, s2StepContext* context, int solverType);
/* == s2World_Step (src/world.c:120-301) with stage 3, the solve and stage 4 on the device; stages 1 and 2 are the two
 * reference functions handed in (s2UpdateBroadPhasePairs, s2BroadPhase_RebuildTrees). */
void s2amdBinding_WorldStep(s2World* world, float timeStep, int velIters, int posIters, bool warmStart, void (*updatePairs)(s2World*),
							void (*rebuildTrees)(s2BroadPhase*));
/* The same two for call sites that have no error path (shim/call_sites.patch): a device error is printed and abort()s -- it must
 * never fall through to the reference's CPU solver on the same world. */
void s2amdBinding_SolveOrDie(s2World* world, s2StepContext* context, int solverType);
void s2amdBinding_WorldStepOrDie(s2World* world, float timeStep, int velIters, int posIters, bool warmStart, void (*updatePairs)(s2World*),
								 void (*rebuildTrees)(s2BroadPhase*));
/* Called first thing by s2DestroyWorld (src/world.c:105-118). */
void s2amdBinding_DestroyWorld(s2World* world);
/* The host pools of `world` brought up to date with the device (manifolds, GJK caches, joint impulses): before anything
 * on the host reads them (s2World_Draw, a sensor, a save file). */
int s2amdBinding_Sync(s2World* world);
/* After editing a resident world through the reference's API (velocities, forces, filters, joint settings ...): its next
 * step uploads it again.  Creating or destroying bodies, shapes and joints is noticed without this. */
void s2amdBinding_Invalidate(s2World* world);
/* 1: stage 1's pair discovery on the device as well (the host trees are still kept up to date, not queried). */
void s2amdBinding_DevicePairs(int on);

/* The pair set s2amd_world_find_pairs returned, put into the order s2UpdateBroadPhasePairs creates contacts in
 * (src/broad_phase.c:253-254, :332-357): read off the reference's own trees and move array. */
void s2amdBinding_OrderPairs(s2World* world, const int* moveArray, int moveCount, int32_t* pairs, int32_t count);

/* S2AMD_CHECK_TREES=1: {pair queries compared, queries whose creation order differed from s2amdBinding_OrderPairs', tree comparisons, trees
 * that differed from the host replay's}; reading resets the counters */
void s2amdBinding_TreeCheck(long out[4]);
/* on: the device keeps the reference's trees and orders the new pairs itself (default; S2AMD_DEVICE_TREES=0 for round 5's host replay);
 * check: both, compared every query (S2AMD_CHECK_TREES=1).  Takes effect at a world's next upload. */
void s2amdBinding_DeviceTrees(int on, int check);
int s2amdBinding_LastError(void);
long s2amdBinding_Uploads(void);		   /* whole-world uploads so far: one per world, and one more whenever a pool grew */
void s2amdBinding_Timing(double out[6]); /* accumulated ms: stage 1+2, sync in, device step, download, apply; [5] = steps */

/* The field-for-field gather / scatter between the reference's pools and the wire structs (array index == pool index). */
void s2amdBinding_PackBodies(const s2World* world, s2amdBody* out);
void s2amdBinding_UnpackBodies(s2World* world, const s2amdBody* in);
void s2amdBinding_PackContacts(const s2World* world, s2amdContact* out);
void s2amdBinding_UnpackContacts(s2World* world, const s2amdContact* in);
void s2amdBinding_PackJoints(const s2World* world, s2amdJoint* out);
void s2amdBinding_UnpackJoints(s2World* world, const s2amdJoint* in);
void s2amdBinding_PackShapes(const s2World* world, s2amdShape* out);
void s2amdBinding_PackPairs(const s2World* world, s2amdPairState* out);

#endif
