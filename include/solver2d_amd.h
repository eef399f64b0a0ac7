/*
 * solver2d_amd.h -- C-ABI of the MI355X constraint-solve hot path.
 *
 * This is the drop-in boundary for the ten solver variants of erincatto/solver2d.  The
 * reference's plug point is
 *
 *     void s2Solve_<Variant>(s2World* world, s2StepContext* context);   (src/solvers.h:70-79)
 *
 * called from the switch in s2World_Step (src/world.c:206-256).  Each s2Solve_* reads and
 * mutates three pooled arrays -- world->bodies (src/body.h:16-76), world->contacts with their
 * embedded s2Manifold (src/contact.h:44-61, include/solver2d/manifold.h:19-46) and
 * world->joints (src/joint.h:28-103) -- and nothing else.  The structs below are plain-C
 * mirrors of exactly the fields those functions touch, indexed the same way (array index ==
 * pool index, free slots flagged), so a reference-side shim is a field-for-field gather
 * before the call and a scatter after it.
 *
 * Plain pointers and sizes only; no C++ / torch / HIP types cross this boundary.
 * All functions return 0 on success and a negative S2AMD_E_* code on failure;
 * s2amd_last_error() returns a human readable message for the calling thread.
 */
#ifndef SOLVER2D_AMD_H
#define SOLVER2D_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2AMD_API_VERSION 5

/* error codes */
#define S2AMD_OK 0
#define S2AMD_E_INVALID (-1)   /* bad argument (null pointer, negative size, unknown solver) */
#define S2AMD_E_DEVICE (-2)    /* HIP runtime failure (message has the hipError string) */
#define S2AMD_E_NODEVICE (-3)  /* no gfx950 device visible: the library never falls back to the CPU */
#define S2AMD_E_STATE (-4)     /* call sequence error (e.g. step_resident before upload) */
#define S2AMD_E_CAPACITY (-5)  /* output buffer too small */

/* Solver selector: values and order are the reference ABI (include/solver2d/types.h:75-88). */
typedef enum s2amdSolverType
{
	s2amd_solverJacobi = 0,
	s2amd_solverPGS = 1,
	s2amd_solverPGS_NGS = 2,
	s2amd_solverPGS_NGS_Block = 3,
	s2amd_solverPGS_Soft = 4,
	s2amd_solverSoftStep = 5,
	s2amd_solverTGS_Sticky = 6,
	s2amd_solverTGS_Soft = 7,
	s2amd_solverTGS_NGS = 8,
	s2amd_solverXPBD = 9,
	s2amd_solverTypeCount = 10
} s2amdSolverType;

/* Body type values are the reference's s2BodyType (types.h:99-105); -1 marks a free pool slot
 * (s2IsFree, src/pool.h). */
#define S2AMD_BODY_FREE (-1)
#define S2AMD_BODY_STATIC 0
#define S2AMD_BODY_KINEMATIC 1
#define S2AMD_BODY_DYNAMIC 2

/* Solver-visible part of s2Body (src/body.h:16-76).  In/out fields are marked. */
typedef struct s2amdBody
{
	float position[2];       /* in/out  center of mass, body.h:24 */
	float rot[2];            /* in/out  {s, c}, body.h:34 */
	float linearVelocity[2]; /* in/out  body.h:39 */
	float angularVelocity;   /* in/out  body.h:40 */
	float deltaPosition[2];  /* in/out  body.h:27 (zero between steps except kinematic bodies under XPBD) */
	float localCenter[2];    /* in      body.h:37 */
	float force[2];          /* in      body.h:49 */
	float torque;            /* in      body.h:50 */
	float mass, invMass;     /* in      body.h:61 */
	float I, invI;           /* in      body.h:64 */
	float linearDamping;     /* in      body.h:66 */
	float angularDamping;    /* in      body.h:67 */
	float gravityScale;      /* in      body.h:68 */
	int32_t type;            /* in      S2AMD_BODY_* */
} s2amdBody;

/* s2ManifoldPoint (include/solver2d/manifold.h:19-38) minus id/persisted, which no solver reads. */
typedef struct s2amdManifoldPoint
{
	float localAnchorA[2];    /* in  relative to body origin */
	float localAnchorB[2];    /* in */
	float frictionAnchorA[2]; /* in/out TGS_Sticky only (solve_tgs_sticky.c:87-163) */
	float frictionAnchorB[2]; /* in/out */
	float frictionNormalA[2]; /* in/out */
	float frictionNormalB[2]; /* in/out */
	float separation;         /* in */
	float normalImpulse;      /* in/out (warm start in, stored impulse out) */
	float tangentImpulse;     /* in/out */
} s2amdManifoldPoint;

/* s2Contact as the solvers see it (src/contact.h:44-61): body indices from edges[0/1].bodyIndex,
 * mixed friction, and the manifold.  pointCount == 0 marks a slot the gather loop skips
 * (free contact or no manifold points; e.g. src/solve_tgs_soft.c:162-179). */
typedef struct s2amdContact
{
	int32_t bodyA, bodyB;
	int32_t pointCount;        /* 0, 1 or 2 */
	int32_t frictionPersisted; /* in/out TGS_Sticky, manifold.h:45 */
	float normal[2];
	float friction;
	int32_t constraintIndex;   /* out  manifold.constraintIndex written by the gather loop; -1 if skipped */
	s2amdManifoldPoint points[2];
} s2amdContact;

#define S2AMD_JOINT_FREE (-1)
#define S2AMD_JOINT_REVOLUTE 0 /* s2_revoluteJoint, src/joint.h:17 */
#define S2AMD_JOINT_MOUSE 1    /* s2_mouseJoint,   src/joint.h:18 */

/* Persistent part of s2Joint + s2RevoluteJoint / s2MouseJoint (src/joint.h:28-103).  The
 * "solver temp" members are recomputed by every prepare function and never cross the boundary. */
typedef struct s2amdJoint
{
	int32_t type; /* S2AMD_JOINT_* */
	int32_t bodyA, bodyB;
	int32_t enableMotor, enableLimit; /* revolute */
	float localOriginAnchorA[2], localOriginAnchorB[2];
	float impulse[2];   /* in/out revolute + mouse */
	float motorImpulse; /* in/out revolute + mouse */
	float lowerImpulse; /* in/out revolute */
	float upperImpulse; /* in/out revolute */
	float maxMotorTorque, motorSpeed, referenceAngle, lowerAngle, upperAngle; /* revolute */
	float hertz, dampingRatio; /* mouse */
	float targetA[2];          /* mouse */
} s2amdJoint;

/* The arguments of s2World_Step (include/solver2d/solver2d.h:25) + world->gravity and
 * world->solverType; s2StepContext (src/solvers.h:13-24) is derived from these exactly as
 * src/world.c:170-202 does. */
typedef struct s2amdStepParams
{
	int32_t solverType; /* s2amdSolverType */
	float dt;
	int32_t velIters;
	int32_t posIters;
	int32_t warmStart;
	float gravity[2];
} s2amdStepParams;

/* Wall/device timing of the last step (filled by s2amd_solve / s2amd_step_resident). */
typedef struct s2amdStepStats
{
	int32_t constraintCount;   /* active contact constraints this step */
	int32_t jointCount;        /* active joints */
	int32_t contactColors;     /* colour batches of the contact graph (0 for Jacobi contacts) */
	int32_t jointColors;
	int32_t solveSweeps;       /* full passes of a s2SolveContacts_* kernel family this step */
	int32_t kernelLaunches;    /* launches enqueued for the step */
	float deviceMs;            /* HIP-event time of the whole device step */
	float solveKernelMs;       /* HIP-event time summed over the individual contact solve-sweep launches (0 unless profiling is on) */
	float hostPrepMs;          /* host time spent colouring/packing */
	int32_t graphReplayed;     /* 1 when the step ran as a hipGraph replay */
	int32_t solveLaunches;     /* contact solve-sweep kernel launches timed into solveKernelMs (profiling only) */
	float eventPairOverheadMs; /* elapsed time of an EMPTY HIP event pair on the stream (profiling only): subtract per launch */
	int32_t groupCount;        /* LDS groups (small islands advanced whole-step by one workgroup each) */
	int32_t messagePassing;    /* 1 when the big-island sweeps read bodies from per-constraint copies (no gather) */
	int32_t stripCount;        /* strips (BFS level ranges) a big island was cut into: phase A workgroups per sweep */
	int32_t seamCount;         /* seams between adjacent strips that carry constraints: phase B workgroups per sweep */
	int32_t persistent;        /* 1 when the strips ran as ONE persistent launch (constraints resident in registers all step) */
	int32_t persistFallbacks;  /* times a persistent step was abandoned (workgroups not co-resident) and repeated on the multi-launch path */
	int32_t structureBuilds;   /* full builds of the constraint-graph structure (islands, colours, tables) since s2amd_create */
	int32_t placedContacts;    /* created contacts that were given a place in the existing structure instead (no build) */
	int32_t potentialConstraints; /* contact slots the structure holds: constraintCount + manifolds without points + destroyed contacts not yet dropped */
	int32_t pairLanes;         /* (API 2) which persistent kernel ran: 2 = 512 threads per strip (wide_kernel.hip), 1 = two lanes per constraint (pair_kernel.hip), 0 = strip_kernel.hip */
	int32_t asyncBuildsRequested; /* (API 3) structure builds handed to the worker thread since s2amd_create (world chain: the strip structure, the strip-width search) */
	int32_t asyncBuildsAdopted;   /* (API 3) ... whose result replaced the live structure (the others were overtaken by the graph) */
	float asyncWaitMs;            /* (API 3) time the caller spent waiting for a worker that was not done when its build fell due, since s2amd_create */
	int32_t bodiesAdopted;        /* (API 3) bodies without constraints that moved to the strip of the body they first touched (no build), in the structure in use */
	int32_t seamBodiesAdded;      /* (API 3) bodies a seam between two strips came to carry after the build (one more export / import of its strips) */
	int32_t roundsOpened;         /* (API 3) spare colour rounds of strips and seams opened for created contacts */
	int32_t overflowContacts;     /* (API 4) contacts that fit nowhere in the strips and wait in the overflow positions behind them for a worker thread's structure (0: none) */
	int32_t slicedStep;           /* (API 4) 1: this step ran sliced -- the persistent strip kernel launched once per sweep, the overflow contacts swept behind each launch */
	int32_t slicedSteps;          /* (API 4) ... steps that did, since s2amd_create */
	int32_t nearHandoffTimeouts;  /* (API 3) hand-off time-outs while the same-XCD path (workgroup-scope stores between neighbours on one L2) was in use: the step was tried again with agent-scope stores, which the solver keeps */
} s2amdStepStats;

typedef struct s2amdSolver s2amdSolver;

/* ---- lifecycle ---- */
int s2amd_api_version(void);
int s2amd_device_count(void);
/* (API 2) PCI bus id of HIP device `device` ("0000:05:00.0"): what tells two ranks' GPUs apart (bench.py: ranks_seen / devices). */
int s2amd_device_bus_id(int device, char* out, int32_t capacity);
const char* s2amd_last_error(void);
/* The floating-point contract this library was compiled under: "fp-contract=off" (libs2amd.so: every result equal to the
 * reference's, bit for bit -- the reference's own build, gcc -O2 on x86-64 without -mfma, contracts nothing) or
 * "fp-contract=fast" (libs2amd_fast.so, the tolerance mode: the compiler may fuse a*b+c into one rounding; results within
 * the tolerances DESIGN.md section 2 states and tests/test_gpu_fast.py checks).  A static string. */
const char* s2amd_build_flags(void);
/* device: HIP ordinal.  Fails with S2AMD_E_NODEVICE when no GPU is visible. */
int s2amd_create(int device, s2amdSolver** out);
void s2amd_destroy(s2amdSolver* solver);

/* ---- drop-in entry point: == s2Solve_<params->solverType>(world, context) ----
 * Host arrays in, same arrays mutated on return (bodies: position/rot/velocities/deltaPosition;
 * contacts: impulses, constraintIndex, sticky cache; joints: impulses).  Any of the array
 * pointers may be NULL when its count is 0. */
int s2amd_solve(s2amdSolver* solver, const s2amdStepParams* params, s2amdBody* bodies, int32_t bodyCapacity,
				s2amdContact* contacts, int32_t contactCapacity, s2amdJoint* joints, int32_t jointCapacity);

/* ---- split phases, for callers that keep the world resident in HBM ---- */
int s2amd_upload(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts,
				 int32_t contactCapacity, const s2amdJoint* joints, int32_t jointCapacity);
/* One s2Solve_* on the resident arrays; results stay on the device (impulses are stored back
 * into the resident contact array, so consecutive calls warm start like consecutive steps). */
int s2amd_step_resident(s2amdSolver* solver, const s2amdStepParams* params);
int s2amd_download(s2amdSolver* solver, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts,
				   int32_t contactCapacity, s2amdJoint* joints, int32_t jointCapacity);
/* Snapshot / restore of the resident body array on the device (bench: re-solve one snapshot). */
int s2amd_save_bodies(s2amdSolver* solver);
/* With option "async" = 1, s2amd_step_resident returns once the step is enqueued on the solver's stream (no
 * device time in the stats); this call waits for everything enqueued so far and reports a deferred device error. */
int s2amd_synchronize(s2amdSolver* solver);
int s2amd_restore_bodies(s2amdSolver* solver);

/* ---- the stages either side of the solver ---- */

/* Shape type values are the reference's s2ShapeType (src/shape.h:14-21). */
#define S2AMD_SHAPE_FREE (-1)
#define S2AMD_SHAPE_CAPSULE 0
#define S2AMD_SHAPE_CIRCLE 1
#define S2AMD_SHAPE_POLYGON 2
#define S2AMD_SHAPE_SEGMENT 3

/* The part of s2Shape (src/shape.h:23-47) that Stage 4 of s2World_Step (AABB refit,
 * src/world.c:259-301) and the broad phase (src/broad_phase.c:166-307) read and write.
 * Geometry: polygon = vertices[0..count) + radius; circle = vertices[0] center, radius;
 * capsule = vertices[0], vertices[1], radius; segment = vertices[0], vertices[1]. */
typedef struct s2amdShape
{
	int32_t body;          /* s2Shape.bodyIndex; -1 with type S2AMD_SHAPE_FREE for a free pool slot */
	int32_t type;          /* S2AMD_SHAPE_* */
	uint32_t categoryBits; /* s2Filter, include/solver2d/types.h:141-146 */
	uint32_t maskBits;
	int32_t groupIndex;
	int32_t proxyKey;      /* (tree proxy id << 4) | body type, src/broad_phase.h:18-20: orders a pair's A/B */
	int32_t enlarged;      /* out of s2amd_refit_shapes: fat AABB grew, the proxy enters the move buffer */
	int32_t count;         /* polygon vertex count */
	float radius;
	float aabb[4];         /* in/out {lower.x, lower.y, upper.x, upper.y} */
	float fatAABB[4];      /* in/out */
	float vertices[8][2];
	float normals[8][2];   /* polygon edge normals (s2Polygon.normals, include/solver2d/geometry.h:44-50): read by the narrow phase */
} s2amdShape;

/* == Stage 4 of s2World_Step (src/world.c:259-301) for every non-static body: origin = position -
 * R * localCenter (written to origins[2 * body]), per shape the tight AABB + speculative margin, and
 * the fat AABB re-inflated where the tight one escaped it (`enlarged` = 1).  Host arrays in and out. */
int s2amd_refit_shapes(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, s2amdShape* shapes, int32_t shapeCapacity,
					   float* origins);

/* == the pair discovery of s2UpdateBroadPhasePairs (src/broad_phase.c:166-307): every moved proxy
 * (moved[shape] != 0) against every proxy whose fat AABB overlaps its own, with the reference's
 * rules (dynamic queries all three trees, kinematic only the dynamic one; no self pairs; when both
 * moved the lower proxy key reports; existing pairs, same-body pairs, filtered pairs and bodies
 * connected by a joint are skipped; shape A is the one with the lower proxy key), followed by
 * s2CreateContact's type rules (src/contact.c:137-175): segment/segment pairs are dropped, a pair
 * whose shape-type order has no primary manifold function is flipped.
 * existingPairs: shapeIndexA/B of the live contacts.  Output: the NEW pairs, sorted by (A, B) -- the
 * same SET the reference creates contacts for.  The reference's creation ORDER is a function of its own
 * trees (move-array order, then s2DynamicTree_Query's traversal order reversed, broad_phase.c:253-254,
 * :332-357): the caller, who owns those trees and the contact pool, sorts the set on that key before
 * calling s2CreateContact (shim/s2_amd_binding.c: s2amdBinding_OrderPairs), which gives every contact
 * the reference's pool slot.  Returns S2AMD_E_CAPACITY (with *pairCount = needed) when outPairs is too small. */
int s2amd_find_pairs(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdShape* shapes, int32_t shapeCapacity,
					 const uint8_t* moved, const int32_t* existingPairs, int32_t existingPairCount, const s2amdJoint* joints,
					 int32_t jointCapacity, int32_t* outPairs, int32_t pairCapacity, int32_t* pairCount);

/* Narrow-phase state of one contact slot that persists between steps: the shape pair, the GJK simplex cache
 * (s2DistanceCache, include/solver2d/distance.h:31-37, a member of s2Contact) and per manifold point the feature
 * id and `persisted` flag (s2ManifoldPoint, include/solver2d/manifold.h:19-38). */
typedef struct s2amdPairState
{
	int32_t shapeA, shapeB; /* s2Contact.shapeIndexA/B; -1: free contact slot */
	float cacheMetric;
	uint16_t cacheCount;
	uint16_t id[2];
	uint8_t cacheIndexA[3];
	uint8_t cacheIndexB[3];
	uint8_t persisted[2];
	uint8_t pad[2];
} s2amdPairState;

#define S2AMD_PAIR_UPDATED 0   /* manifold recomputed */
#define S2AMD_PAIR_SEPARATED 1 /* fat AABBs no longer overlap: the caller destroys the contact (src/world.c:149-167) */
#define S2AMD_PAIR_FREE (-1)

/* == Stage 3 of s2World_Step, "update contacts" (src/world.c:132-168): for every live contact slot whose shapes'
 * fat AABBs still overlap, s2UpdateContact (src/contact.c:296-358): the manifold function of the shape-type pair
 * (src/manifold.c: s2CollideCircles, s2CollideCapsuleAndCircle, s2CollidePolygonAndCircle, s2CollidePolygons with
 * its GJK distance query src/distance.c:485-604, SAT fallback and clipper; capsules and segments go through the
 * polygon path exactly as the reference's s2MakeCapsule does), then the id matching that carries impulses and the
 * sticky-friction cache from the old manifold to the new one.
 * In:  bodies (rot), origins[2 * body] (s2Body.origin: s2amd_refit_shapes writes them), shapes (geometry in the
 *      body frame, fatAABB), pairs, contacts (the old manifolds).
 * Out: contacts (pointCount, normal, points, frictionPersisted; bodyA/B, friction and constraintIndex are kept),
 *      pairs (ids, persisted, cache), status[contact] = S2AMD_PAIR_*.  Host arrays in and out. */
int s2amd_update_contacts(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const float* origins, const s2amdShape* shapes,
						  int32_t shapeCapacity, s2amdPairState* pairs, s2amdContact* contacts, int32_t contactCapacity, int32_t* status);

/* ---- resident world: stage 3, the solve and stage 4 of s2World_Step chained in HBM ----
 * The stage functions above take host arrays in and out.  These three keep the shapes, the pair states and the body
 * origins on the device beside the bodies / contacts / joints of s2amd_upload, so a step moves 32 bytes of counters, read
 * back once, after stage 4.  The constraint-graph structure (islands, colours, strips) covers every live pair slot
 * whether its manifold has points or not, so manifolds that gain or lose their points cost the host nothing; it is
 * touched when contact slots are written.  Pair creation (stage 1, src/world.c:125-130) stays with the caller: when
 * movedCount > 0 it downloads the shapes, runs s2amd_find_pairs, creates its contacts and uploads again, as the
 * reference does before the next step's stage 3. */
typedef struct s2amdWorldStepInfo
{
	int32_t separatedCount; /* pairs whose fat AABBs parted: destroyed on the device (pointCount 0, pair slot free,
	                           src/world.c:149-167); status[] of s2amd_world_download says which */
	int32_t activeContacts; /* manifolds with points after stage 3 */
	int32_t graphChanged;   /* a manifold went between zero and non-zero points this step.  Informational: the solve's structure
	                           covers every live pair slot, with or without points, and is only rebuilt when contact slots are
	                           written (s2amd_world_set_contacts) */
	int32_t movedCount;     /* shapes whose fat AABB the refit re-inflated (s2amdShape.enlarged) */
	float contactsMs;       /* host wall time of enqueueing stage 3 (its counters come back with the step's one read-back) */
	float solveMs;          /* device time of the s2Solve_* part (HIP events) */
	float stepMs;           /* host wall time of the whole call */
} s2amdWorldStepInfo;

/* bodies/contacts/joints as s2amd_upload; shapes (with valid aabb/fatAABB), pairs[contactCapacity] (the narrow-phase
 * state of every contact slot), origins[2 * bodyCapacity] (s2Body.origin). */
int s2amd_world_upload(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
					   const s2amdJoint* joints, int32_t jointCapacity, const s2amdShape* shapes, int32_t shapeCapacity, const s2amdPairState* pairs,
					   const float* origins);
/* == s2World_Step without stage 1: update contacts (src/world.c:132-168), s2Solve_* (src/world.c:206-256), refit
 * (src/world.c:259-301).  info may be NULL. */
int s2amd_world_step(s2amdSolver* solver, const s2amdStepParams* params, s2amdWorldStepInfo* info);
/* == the pair discovery of stage 1 (s2amd_find_pairs above) on the resident shapes: moved = the shapes the last refit
 * enlarged, existing pairs = the live pair slots, jointed bodies as uploaded.  New pairs sorted by (A, B) into the host
 * array outPairs.  Call it when info.movedCount > 0, before the next s2amd_world_step (the refit overwrites the flags).
 * A successful query consumes the move buffer as s2UpdateBroadPhasePairs does (src/broad_phase.c: moveArray and moveSet
 * are cleared): every shape's `enlarged` flag is zero afterwards, static shapes included. */
int s2amd_world_find_pairs(s2amdSolver* solver, int32_t* outPairs, int32_t pairCapacity, int32_t* pairCount);
/* The contact slots whose pairs the last s2amd_world_step found separated (fat AABBs apart) and destroyed on the device, in
 * ascending order: the slots the caller's own pool frees (s2DestroyContact, src/world.c:163-167).  info.separatedCount of them. */
int s2amd_world_separated(s2amdSolver* solver, int32_t* slots, int32_t capacity, int32_t* count);
/* What stage 4 (src/world.c:259-297) changed in a shape, without the rest of the 196-byte record: the tight box of every live
 * shape, its fat box, and whether the refit re-inflated it (== the shapes s2BroadPhase_EnlargeProxy is called for). */
typedef struct s2amdShapeBox
{
	float aabb[4];
	float fatAABB[4];
	int32_t enlarged;
} s2amdShapeBox;
/* boxes[shapeCapacity of the upload]: the boxes of the resident shapes after the last s2amd_world_step (36 bytes per shape
 * instead of s2amd_world_download's whole shape records). */
int s2amd_world_download_boxes(s2amdSolver* solver, s2amdShapeBox* boxes, int32_t shapeCapacity);
/* (API 2) The per-step read-back of a caller that keeps its own copy of the world and its own broad-phase trees (the binding,
 * shim/s2_amd_binding.c): after an s2amd_world_step, instead of the whole body and box arrays,
 *   poses[4 * body]  {origin.x, origin.y, rot.s, rot.c} of every body slot (what s2Body_GetPosition / GetAngle read), and
 *   moved[]          the shapes whose fat box the refit re-inflated (src/world.c:283-290), with the new fat box, IN THE ORDER the
 *                    reference's refit visits them -- `shapeOrder` of s2amd_world_set_refit_order: bodies in pool order, each
 *                    body's shape list -- which is the order s2BroadPhase_EnlargeProxy puts them into the move buffer in.
 * *movedCount == s2amdWorldStepInfo.movedCount of that step.  A new s2amd_world_upload forgets the order (send it again).  Velocities, manifolds, tight boxes stay in HBM until somebody
 * asks for them (s2amd_world_download / _download_boxes). */
typedef struct s2amdMovedBox
{
	int32_t shape;
	float fatAABB[4];
} s2amdMovedBox;
int s2amd_world_set_refit_order(s2amdSolver* solver, const int32_t* shapeOrder, int32_t count);
int s2amd_world_download_step(s2amdSolver* solver, float* poses, int32_t bodyCapacity, s2amdMovedBox* moved, int32_t movedCapacity, int32_t* movedCount);
/* (API 5) The reference's broad-phase trees on the device.  s2UpdateBroadPhasePairs creates a step's contacts in the order its tree
 * queries call back (src/broad_phase.c:253-254, :288-320, :332-357: move-buffer order of the asking proxies; per proxy the trees in
 * reverse query order and each tree's callbacks reversed), and that order decides every new contact's pool slot.  It is a function of
 * the topology of the reference's three trees (src/broad_phase.h:27), which is history: stage 2 rebuilds only what stage 4 flagged
 * (s2DynamicTree_Rebuild(tree, false), src/dynamic_tree.c:1764-1874).  A caller that hands the device those trees once --
 *   s2amd_world_set_tree(solver, bodyType, tree.nodes, tree.nodeCapacity, tree.root)   for s2_staticBody, s2_kinematicBody, s2_dynamicBody
 * after s2amd_world_upload, in the state stage 2 leaves them in (no internal node flagged `enlarged`; refused otherwise) -- gets
 *   - s2amd_world_step keeping them as the reference would: its stage 2 rebuilds them (same topology, same node ids, same boxes),
 *     its stage 4 enlarges the proxies of the shapes it re-inflates (src/world.c:283-290, src/dynamic_tree.c:803-839);
 *   - s2amd_world_find_pairs returning the new pairs IN THE REFERENCE'S CREATION ORDER instead of sorted by (A, B): calling
 *     s2CreateContact down the list gives every contact the reference's pool slot (the position of an asking proxy in the move buffer
 *     is its shape's position in the refit order of s2amd_world_set_refit_order; without one, its shape index);
 *   - s2amd_world_get_tree copying a tree back (node array as the reference would hold it now -- the free list's nodes are untouched
 *     by a rebuild, which takes exactly the nodes it frees, src/dynamic_tree.c:105-139 -- and the root), for the caller's own
 *     s2DynamicTree when it needs one: s2World_QueryAABB, s2World_Draw, a proxy created or destroyed.
 * A new s2amd_world_upload forgets the trees.  s2amdTreeNode is s2TreeNode byte for byte (include/solver2d/dynamic_tree.h:14-41). */
typedef struct s2amdTreeNode
{
	float aabb[4];
	uint32_t categoryBits;
	int32_t parent; /* `next` of a free node */
	int32_t child1, child2;
	int32_t userData;
	int16_t height; /* leaf 0, free node -1 */
	uint8_t enlarged;
	uint8_t pad[9];
} s2amdTreeNode;
int s2amd_world_set_tree(s2amdSolver* solver, int32_t bodyType, const s2amdTreeNode* nodes, int32_t nodeCapacity, int32_t root);
int s2amd_world_get_tree(s2amdSolver* solver, int32_t bodyType, s2amdTreeNode* nodes, int32_t nodeCapacity, int32_t* root);
/* Writes `count` contact slots of the resident world (slot indices < contactCapacity of the upload): the caller's
 * s2CreateContact (src/contact.c:137-203: pool slot, pair flip, mixed friction, empty manifold) or s2DestroyContact
 * (pairs[i].shapeA = -1, contacts[i].pointCount = 0).  A world that needs more slots, bodies or shapes is uploaded again. */
int s2amd_world_set_contacts(s2amdSolver* solver, const int32_t* slots, int32_t count, const s2amdContact* contacts, const s2amdPairState* pairs);
/* Any output pointer may be NULL.  status[contactCapacity]: S2AMD_PAIR_* of the last step's stage 3. */
int s2amd_world_download(s2amdSolver* solver, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts, int32_t contactCapacity,
						 s2amdJoint* joints, int32_t jointCapacity, s2amdShape* shapes, int32_t shapeCapacity, s2amdPairState* pairs, float* origins,
						 int32_t* status);

/* ---- constraint-graph structure on the device (the reference has neither islands nor colours) ----
 * Islands: connected components over the movable bodies (invMass != 0 or invI != 0) joined by active contacts
 * (pointCount > 0) and revolute joints; every other live non-static body is an island of its own; static and free
 * bodies get -1.  Islands are numbered by their lowest body index.  == solver2d_amd/islands.py: find_islands. */
int s2amd_find_islands(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
					   const s2amdJoint* joints, int32_t jointCapacity, int32_t* islandOfBody, int32_t* islandCount);
/* A proper colouring of the active contacts: no two contacts of one colour share a movable body, so one colour is one
 * race-free parallel batch of a Gauss-Seidel sweep.  Deterministic: the greedy colouring in descending order of the
 * fixed priority hash fmix32(contact index).  colorOfContact[c] = -1 for inactive slots.  *rounds: Jones-Plassmann
 * rounds the device needed. */
int s2amd_color_constraints(s2amdSolver* solver, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
							int32_t* colorOfContact, int32_t* colorCount, int32_t* rounds);

/* Multi-GPU exchange: writes one {position.x, position.y, rot.s, rot.c} record per body slot into
 * a DEVICE buffer owned by the caller (e.g. the send buffer of an RCCL all-gather of per-island
 * body arrays).  Returns after the copy has completed on the solver's stream. */
int s2amd_export_poses(s2amdSolver* solver, void* devicePoses, int32_t capacity);
/* The same copy, enqueued behind the steps already enqueued on the solver's stream (see "async") and followed by an event
 * in `slot` (0..3); s2amd_export_wait blocks the host until that event has fired.  Lets a host enqueue step s+1 before it
 * waits for the poses of step s and hands them to the collective: the device never idles between steps. */
int s2amd_export_poses_async(s2amdSolver* solver, void* devicePoses, int32_t capacity, int32_t slot);
/* (API 2) ... the per-island body arrays of a sharded world (28 bytes per body): TWO 16-byte records per body slot,
 * {position.x, position.y, rot.s, rot.c} and {linearVelocity.x, linearVelocity.y, angularVelocity, 0}; capacity in bodies. */
int s2amd_export_bodies_async(s2amdSolver* solver, void* deviceRecords, int32_t capacity, int32_t slot);
int s2amd_export_wait(s2amdSolver* solver, int32_t slot);

/* Device buffers for callers that have no device allocator of their own (a host language behind cgo / JNI / ctypes): the
 * destination of s2amd_export_poses and the source of s2amd_device_read.  Owned by the caller, freed with
 * s2amd_device_free (or never: s2amd_destroy does not track them). */
int s2amd_device_alloc(s2amdSolver* solver, uint64_t bytes, void** devicePtr);
int s2amd_device_free(s2amdSolver* solver, void* devicePtr);
/* Copies `bytes` from device memory to host memory behind everything enqueued on the solver's stream; returns when done. */
int s2amd_device_read(s2amdSolver* solver, void* hostDst, const void* deviceSrc, uint64_t bytes);

/* ---- introspection (tests, bench) ---- */
/* Execution order of the last step: order[k] = contact-array index of the k-th constraint in
 * sweep order; colorOffsets[c]..colorOffsets[c+1] delimit colour batch c.  A sequential
 * Gauss-Seidel sweep in this order is arithmetic-identical to the batched device sweep.  (s2Solve_Jacobi's contact pass
 * writes no body and needs no colours: there only the order -- the order of the per-body sums -- is meaningful.) */
int s2amd_get_contact_order(s2amdSolver* solver, int32_t* order, int32_t orderCapacity, int32_t* colorOffsets,
							int32_t colorCapacity, int32_t* constraintCount, int32_t* colorCount);
int s2amd_get_joint_order(s2amdSolver* solver, int32_t* order, int32_t orderCapacity, int32_t* colorOffsets,
						  int32_t colorCapacity, int32_t* jointCount, int32_t* colorCount);
/* (API 5) writable[b] = 1: the sweeps the contact order above was coloured for write body b -- no two constraints of one colour share
 * such a body (the check a caller can make of an order it is handed).  *solverClass: 0 the velocity sweeps (bodies with mass), 1 the
 * position sweeps of PGS_NGS / PGS_NGS_Block / TGS_NGS, which also rewrite the rotation of immovable bodies (src/solve_common.c:383-392:
 * only a static body whose rotation that normalisation leaves alone is read-only there). */
int s2amd_get_writable_bodies(s2amdSolver* solver, uint8_t* writable, int32_t bodyCapacity, int32_t* solverClass);
int s2amd_get_stats(s2amdSolver* solver, s2amdStepStats* stats);
/* (API 3) Which strip -- which workgroup of the persistent step kernel -- owns each body in the structure in use, and which bodies the
 * seam between strips i and i + 1 carries: ownerStrip[body] = strip or -1 (a static body, a body of an LDS group or of the global part;
 * every entry -1 when the structure has no strips or keeps no picture of them), onSeam[body] = i when the seam i | i + 1 exchanges the
 * body every sweep, else -1.  Either array may be NULL; capacity = entries each can hold (>= the body capacity of the world).
 * *stripCount = number of strips.  Tests use it to build contacts whose place in the structure they know. */
int s2amd_get_strip_owners(s2amdSolver* solver, int32_t* ownerStrip, int32_t* onSeam, int32_t capacity, int32_t* stripCount);
/* Live timing of the dominant kernel on the solver's own stream: the first contact solve sweep of
 * the step plan (all its colour-batch launches) is enqueued `repeats` times back to back in one
 * hipGraph and bracketed by a HIP event pair; *usPerLaunch = elapsed / launches, i.e. the time one
 * launch occupies in steady state (kernel + dependent-kernel boundary).  When every constraint lives
 * in an LDS group the whole-step group kernel is timed instead (launchesPerSweep = 1).  The resident
 * world is advanced by the extra sweeps: call it after the timed region. */
int s2amd_measure_dominant(s2amdSolver* solver, const s2amdStepParams* params, int32_t repeats, float* usPerLaunch,
						   int32_t* launchesPerSweep, int32_t* constraintsPerLaunch);
/* option keys: "graph" (0/1 hipGraph replay), "profile" (0/1 per-sweep HIP events),
 * "groups" (0/1 LDS group path for small islands), "message" (0/1 message-passing sweeps), "max_group_bodies", "pack_group_bodies",
 * "strips" (0/1 cut islands that fit no LDS group into strips of BFS levels: two launches per sweep), "strip_bodies" (target
 * bodies per strip, default 8 = strips of two BFS levels), "near_handoff" (0/1, default 1: hand-offs between workgroups the per-launch census finds on one XCD stay in its L2), "strip_retry" (0/1 rebuild the partition with other strip widths when the persistent kernel cannot take this one or it needs more than five interior colour rounds), "strip_min_bodies" (loose bodies below which the colour-batch path is kept), "strip_patience" (steps the constraint graph must
 * stay unchanged before the strip structure is built: its host build costs ~3 ms at 60k constraints, the colour-batch one ~1 ms), "async" (0/1, see s2amd_synchronize), "strip_lean" (0/1 dedicated strip
 * kernel for the soft sweeps), "persist" (0/1 whole step of the strips in one persistent launch), "wide" (0/1 TGS_Soft's persistent launch runs 512 threads per strip: wide_kernel.hip),
 * "generic" (0/1 every other Gauss-Seidel solver, and any big island with joints, runs its whole step as one launch of the op interpreter over the
 * same strips: generic_kernel.hip; 0 = colour batches for them), "persist_retry" (steps a solver whose persistent launch lost a hand-off stays on the
 * fallback path before the one-launch kernels get another chance -- the wait doubles with every further time-out; default 256, 0 = for ever),
 * "stage_joints" (0/1 the op interpreter keeps a strip's joint records in LDS for the whole step when they fit),
 * "step_readback" (0/1 s2amd_world_step of a world with a refit order brings the poses and the re-inflated boxes along in its own synchronisation: s2amd_world_download_step then
 * costs a host copy),
 * "free_body_groups" (0/1 bodies without any constraint form LDS groups of their own next to groups / strips instead of riding the global path's body launches),
 * "self_contained" (0/1 a world of resident islands only is stepped by their kernel alone: it stages its bodies from the wire records and writes them back),
 * "pair_lanes" (0/1 that launch solves a constraint with two lanes, one per body: pair_kernel.hip; measured no faster, off by default), "body_warm", "incremental" (0/1 created
 * contacts are placed into the existing structure when they fit; 0 = every created contact rebuilds it), "defer" (0/1 a created
 * contact that cannot be placed and has no manifold points yet is only watched until it gets its first points; 0 = it rebuilds the
 * structure when it is created), "island_resident" (0/1 small islands under the soft contact solvers keep their constraints in
 * registers for the whole step: one read of every record per step).  None of them changes a result beyond the sweep order the
 * library reports. */
int s2amd_set_option(s2amdSolver* solver, const char* key, int32_t value);

/* ---- (API 4) island-sharded worlds: ONE process, N devices (csrc/sharded.hip) ----
 * A world whose constraint graph falls into islands -- connected components over the movable bodies, as s2amd_find_islands labels them --
 * is partitioned over `deviceCount` shards, one s2amdSolver each on the HIP device named for it (the same ordinal may be named more
 * than once: logical shards on one GPU).  Islands share no movable body, so a step needs no collective: every shard runs the whole
 * s2Solve_* of its islands; the step's ONE exchange carries each shard's owned body records -- {position, rot}, {linearVelocity,
 * angularVelocity, 0}: 28 bytes per body in two 16-byte records -- into every device's copy of the WHOLE world's body records, on a
 * stream of its own per shard (the next step's solve does not wait for it).  How it travels is chosen at s2amd_sharded_create: ONE
 * kernel per shard storing into every copy when the shards share a device; ONE ncclAllGather per device plus a scatter kernel between
 * distinct devices (RCCL over xGMI, single process: librccl.so is looked up at run time and the library loads without it); peer
 * copies otherwise (S2AMD_SHARDED_EXCHANGE=stores|rccl|copies overrides).  The multi-PROCESS form of the same partition (one rank per
 * GPU, torch.distributed over RCCL) is solver2d_amd/distributed.py.
 *   s2amd_sharded_upload    islands found on the device, bin-packed by constraint count (2 per contact constraint + 1 per joint; longest
 *                           processing time first, ties by index: deterministic), every shard's sub-world uploaded -- its islands'
 *                           bodies and constraints in pool order, immovable bodies they touch as read-only replicas;
 *   s2amd_sharded_step      == s2Solve_<solverType> of the whole world, results resident on the shards' devices, + the exchange
 *                           (== s2amd_sharded_step_async + s2amd_sharded_wait);
 *   s2amd_sharded_step_async  (API 5) the same, enqueued only: O(shards) stream operations, no host wait;
 *   s2amd_sharded_wait      (API 5) the one host wait (one stream, which takes every shard's last event).  A shard whose persistent
 *                           launch lost a hand-off is noticed here: with ONE step outstanding its step is repeated and its rows exchanged
 *                           again; with more, the steps behind the failing one were dropped on that shard only, the call fails and the
 *                           world must be uploaded again.  Download, read_bodies, reshard and upload wait by themselves;
 *   s2amd_sharded_download  the world's arrays on the host, every body and constraint from the shard that owns it; constraintIndex as
 *                           the reference's gather loop over the whole pool writes it;
 *   s2amd_sharded_read_bodies  float[bodyCapacity][8] of the world's body records as of the last exchange, read from `shard`'s device;
 *   s2amd_sharded_reshard   the constraint graph changed (contacts / joints: the world's new arrays, pool capacities unchanged, or NULL
 *                           for "unchanged"): solver state comes down, a slot that kept its pair keeps its impulses, islands are found
 *                           again, an island stays on the shard that owned most of its bodies (ties: the lowest shard) -- a contact
 *                           created between two islands moves the smaller one --, islands nobody owned go to the least loaded shard,
 *                           and past 1.75 x the mean load the heaviest shard gives up its lightest islands;
 *   s2amd_sharded_solver    the s2amdSolver of a shard, for the queries of this header (orders, stats, options); owned by the sharded solver. */
typedef struct s2amdShardedSolver s2amdShardedSolver;
int s2amd_sharded_create(const int32_t* devices, int32_t deviceCount, s2amdShardedSolver** out);
void s2amd_sharded_destroy(s2amdShardedSolver* sharded);
int s2amd_sharded_shard_count(const s2amdShardedSolver* sharded);
s2amdSolver* s2amd_sharded_solver(s2amdShardedSolver* sharded, int32_t shard);
int s2amd_sharded_upload(s2amdShardedSolver* sharded, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
						 const s2amdJoint* joints, int32_t jointCapacity);
int s2amd_sharded_step(s2amdShardedSolver* sharded, const s2amdStepParams* params);
int s2amd_sharded_step_async(s2amdShardedSolver* sharded, const s2amdStepParams* params);
int s2amd_sharded_wait(s2amdShardedSolver* sharded);
/* what the last s2amd_sharded_step_async enqueued, counted where it is enqueued: stream operations (launches, copies, collectives,
 * event records and waits), host waits since (s2amd_sharded_wait: 1), and the form of the exchange (0 stores, 1 rccl, 2 peer copies) */
int s2amd_sharded_get_step_ops(s2amdShardedSolver* sharded, int32_t* streamOps, int32_t* hostWaits, int32_t* exchange);
/* the same count for `shards` shards in the given form, by the code that enqueues a step, enqueueing nothing (no device needed) */
int s2amd_sharded_count_ops(int32_t shards, int32_t exchange, int32_t* streamOps);
int s2amd_sharded_download(s2amdShardedSolver* sharded, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts, int32_t contactCapacity,
						   s2amdJoint* joints, int32_t jointCapacity);
int s2amd_sharded_read_bodies(s2amdShardedSolver* sharded, int32_t shard, float* records, int32_t bodyCapacity);
int s2amd_sharded_reshard(s2amdShardedSolver* sharded, const s2amdContact* contacts, int32_t contactCapacity, const s2amdJoint* joints, int32_t jointCapacity);
/* shardOfBody[b]: the shard that owns body b (-1: static or free -- replicated, owned by nobody); may be NULL */
int s2amd_sharded_get_partition(s2amdShardedSolver* sharded, int32_t* shardOfBody, int32_t capacity, int32_t* islandCount, int32_t* reshards);

#ifdef __cplusplus
}
#endif

#endif /* SOLVER2D_AMD_H */
