// Host-callable launchers of every kernel (defined in the *.hip files).
#pragma once

#include <hip/hip_runtime.h>

#include "solver2d_amd.h"

struct BodyView;
struct ContactView;
struct JointView;
struct JointPrepArgs;
struct Stage4Args;
struct StepConsts;
struct GroupTable;
struct Op;
struct MsgView;
struct StripTableView;
struct StripOps;
struct PersistView;
struct JacobiView;
struct s2amdBody;
struct s2amdContact;
struct s2amdJoint;
struct s2amdShape;
struct s2amdPairState;

// body flag bits that only the host sets (see s2_device.h for the rest)
#define S2F_WRITE_VEL 8u  // body is a conflict node for velocity sweeps (not read-only shareable)
#define S2F_WRITE_POS 16u // body is a conflict node for position sweeps
#define S2F_IN_GROUP 32u  // body is owned by an LDS group: the streaming body kernels skip it

enum PrepareKind
{
	PREP_PGS,
	PREP_SOFT,
	PREP_TGS,
	PREP_STICKY,
	PREP_XPBD,
	PREP_BLOCK
};
enum WarmKind
{
	WARM_CURRENT,
	WARM_FIXED,
	WARM_BLOCK
};
enum SoftKind
{
	SOFT_TGS,
	SOFT_PGS,
	SOFT_JACOBI,
	SOFT_FIXED
};
enum RigidKind
{
	RIGID_BAUMGARTE,
	RIGID_PGS,
	RIGID_TGS
};
enum StoreKind
{
	STORE_PLAIN,
	STORE_SCALED,
	STORE_BLOCK
};
enum JointPrepareKind
{
	JPREP_PLAIN,
	JPREP_SOFT,
	JPREP_XPBD
};
enum JointSolveKind
{
	JSOLVE_PLAIN,	  // s2SolveJoint
	JSOLVE_SOFT,	  // s2SolveJoint_Soft
	JSOLVE_BAUMGARTE, // s2SolveJoint_Baumgarte
	JSOLVE_POSITION,  // s2SolveJointPosition
	JSOLVE_XPBD,	  // s2SolveJoint_XPBD
	JSOLVE_WARM		  // s2WarmStartJoint
};

// contacts
bool launchPrepareContacts(hipStream_t s, int kind, const ContactView& c, const BodyView& b, s2amdContact* wire, const s2amdBody* wireBodies,
						   const StepConsts& sc, float h, float hertz, int posSolver, const uint32_t* hostFlags, bool unpackToo, float unpackH,
						   int contactCapacity, const int* gatherIndex, const JointPrepArgs* joints = nullptr);
void launchWarmStartContacts(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end);
void launchSolveContactsSoft(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end, float inv_h, int useBias);
void launchSolveContactsRigid(hipStream_t s, int kind, const ContactView& c, const BodyView& b, int begin, int end, float inv_h);
void launchSolveContactsSticky(hipStream_t s, const ContactView& c, const BodyView& b, s2amdContact* wire, int begin, int end, float inv_h,
							   int useBias);
void launchSolveContactsNGS(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end);
void launchXpbdContactPositions(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end, float h);
void launchXpbdContactVelocities(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end, float h);
void launchBlockSolveVelocity(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end);
void launchBlockSolvePosition(hipStream_t s, const ContactView& c, const BodyView& b, int begin, int end);
bool launchStoreImpulses(hipStream_t s, int kind, const ContactView& c, s2amdContact* wire, float scale, const BodyView& bodies, s2amdBody* wireBodies,
						 void* clear, size_t clearBytes, const unsigned int* stepFailed, int finalizeMode = -1, const JointView* joints = nullptr,
						 s2amdJoint* wireJoints = nullptr,
						 const Stage4Args* stage4 = nullptr); // (with joints: their impulses go back to the wire in the same launch)

// bodies
bool launchUnpackBodies(hipStream_t s, const BodyView& b, const s2amdBody* wire, const uint32_t* hostFlags, const StepConsts& sc, float h,
						s2amdContact* wireContacts, int contactCapacity, const int* gatherIndex, const JointPrepArgs* joints = nullptr, int posSolver = 0);
void launchPackBodies(hipStream_t s, const BodyView& b, s2amdBody* wire);
void launchIntegrateVelocities(hipStream_t s, const BodyView& b);
void launchIntegratePositions(hipStream_t s, const BodyView& b, float h);
void launchFinalizePositions(hipStream_t s, const BodyView& b, int dynamicOnly);
void launchJacobiApply(hipStream_t s, const BodyView& b, const ContactView& c, const int2* adjRange, const int* adjList, const int* heavy, int heavyCapacity);
void launchPatchWords(hipStream_t s, const void* devicePatches, int n);
void launchOverflowSweep(hipStream_t s, const Op& o, const ContactView& c, const BodyView& b, int begin, int end); // contact_kernels.hip
void launchXpbdIntegrate(hipStream_t s, const BodyView& b, float h);
void launchXpbdProject(hipStream_t s, const BodyView& b, float inv_h);
void launchExportPoses(hipStream_t s, const s2amdBody* wire, int n, void* out, int withVelocities = 0);

// joints
void launchPrepareJoints(hipStream_t s, int kind, const JointView& j, const uint32_t* hostFlags, const s2amdJoint* wire, const s2amdBody* wireBodies,
						 const StepConsts& sc, float h, float hertz, int warmStart, int posSolver);
void launchSolveJoints(hipStream_t s, int kind, const JointView& j, const BodyView& b, int begin, int end, const StepConsts& sc, float h,
					   float inv_h, int useBias);
void launchStoreJoints(hipStream_t s, const JointView& j, s2amdJoint* wire, const unsigned int* stepFailed);

// LDS groups
int groupKernelSetup();
void launchGroupKernel(hipStream_t s, const ContactView& c, const JointView& j, const BodyView& g, const GroupTable& gt, const Op* ops, int opCount,
					   const StepConsts& sc, s2amdContact* wire, int maxBodies, int useDq0);
void launchStripKernel(hipStream_t s, const ContactView& c, const JointView& j, const BodyView& g, const GroupTable& gt, const Op* ops, int opCount,
					   const StepConsts& sc, s2amdContact* wire, int maxBodies, int useDq0);

// message passing (big-island path)
void launchFillMessageSlots(hipStream_t s, const ContactView& c, const BodyView& b, const MsgView& m, int count);
void launchWarmStartContactsMsg(hipStream_t s, int kind, const ContactView& c, const MsgView& m, int begin, int end);
void launchSolveContactsSoftMsg(hipStream_t s, int kind, const ContactView& c, const MsgView& m, int begin, int end, float inv_h, int useBias);
void launchSolveContactsRigidMsg(hipStream_t s, int kind, const ContactView& c, const MsgView& m, int begin, int end, float inv_h);
void launchSolveContactsStickyMsg(hipStream_t s, const ContactView& c, const MsgView& m, s2amdContact* wire, int begin, int end, float inv_h,
								  int useBias);
void launchIntegrateVelocitiesMsg(hipStream_t s, const BodyView& b, const MsgView& m);
void launchIntegratePositionsMsg(hipStream_t s, const BodyView& b, const MsgView& m, float h);
void launchFinalizePositionsMsg(hipStream_t s, const BodyView& b, const MsgView& m, int dynamicOnly);
void launchGatherMessageSlots(hipStream_t s, const BodyView& b, const MsgView& m);

// body-centric warm start (one launch for all colours, optionally fused with integrate velocities)
// heavy: the bodies with more than S2_HEAVY_DEGREE list entries (a whole wave walks each of them)
void launchWarmStartJointsBodies(hipStream_t s, const JointView& j, const BodyView& b, const int2* adjRange, const int* adjList);
void launchWarmStartBodies(hipStream_t s, int kind, const ContactView& c, const BodyView& b, const int2* adjRange, const int* adjList,
						   int integrateFirst, const int* heavy, int heavyCount);

// strip_kernel.hip
int stripKernelSetup();
void launchIslandStep(hipStream_t s, int kind, int warm, const ContactView& c, const BodyView& g, const StripTableView& t, const float4* softCoef, const Op* ops,
					  int opCount, int maxRounds, s2amdContact* wire, const s2amdBody* wireBodies, const uint32_t* hostFlags, int warmStart,
					  const unsigned int* stepFailed);
void launchStripSoft(hipStream_t s, int kind, int warm, const ContactView& c, const BodyView& g, const StripTableView& t, const StripOps& ops);
void launchStripStep(hipStream_t s, int kind, int warm, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv,
					 const Op* ops, int opCount);

// pair_kernel.hip: the persistent strip step with two lanes per constraint (pv.pairLanes)
int pairKernelSetup();
void launchPairStep(hipStream_t s, int kind, int warm, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv, const Op* ops,
					int opCount);

// wide_kernel.hip: TGS_Soft's persistent strip step on 512 threads per strip
int wideKernelSetup();
// What the self-contained variant of the strip step needs beside the tables (wide_kernel.hip: S2_WIDE_SELF): the wire arrays it
// reads its bodies and constraints from and writes its results to, and the constants of the body prologue (body_ops.h: unpackBodyOne).
struct WideSelf
{
	s2amdContact* wire;
	s2amdBody* wireBodies;
	const uint32_t* hostFlags;
	int warmStart;
	float gravityX, gravityY, unpackH;
};
// float4 records of dynamic LDS the kernel variant for this partition needs beside the bodies, the ops and its three fixed records
// (parked rounds, staged positions, the warm start's term table); -1: no variant takes the partition
int wideExtraRecords(const PersistView& pv, int selfContained, int bodyWarm, int kind = SOFT_TGS);
int wideBodyWarmVariant(const PersistView& pv);
int wideIslandLocalRecords(int maxRounds); // LDS records the resident-island kernel's eight-round variant keeps local anchors in // the variant for this partition has the body-centric warm start
// ... and the resident islands' step (strip_kernel.hip: launchIslandStep) for TGS_Soft with the current-anchor warm start
// selfContained: the kernel also stages its bodies from the wire records and writes them back (no prologue / epilogue launch)
void launchWideIsland(hipStream_t s, const ContactView& c, const BodyView& g, const StripTableView& t, const float4* softCoef, const Op* ops, int opCount,
					  int maxRounds, s2amdContact* wire, s2amdBody* wireBodies, const uint32_t* hostFlags, int warmStart, const StepConsts& sc, float unpackH,
					  int selfContained, const unsigned int* stepFailed, int allTwoPoints);
// self != nullptr: the self-contained variant (the step is this one launch); pv.bodyWarm: the body-centric warm start
// kind: SOFT_TGS, SOFT_PGS (the two soft drivers whose constraint fits the 22-dword register record) or SOFT_FIXED (the same record plus
// rA0 / rB0 in LDS)
void launchWideStep(hipStream_t s, int kind, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv, const Op* ops, int opCount,
					const WideSelf* self);

// jacobi_kernel.hip: s2Solve_Jacobi as one persistent launch over blocks of bodies (tables: solver_jacobi.cpp)
int jacobiKernelSetup();
size_t jacobiStepLds(int owned, int imports, int constraints, int opCount);
void launchJacobiStep(hipStream_t s, const ContactView& c, const JointView& jv, const BodyView& g, const JacobiView& t, const Op* ops, int opCount, const StepConsts& sc,
					  size_t ldsBytes, int maxConstraints);

// generic_kernel.hip: the persistent strip step as an op interpreter -- every solver family, joints included
int genericKernelSetup();
size_t genericStepLds(int bodies, int seamBodies, int exports, int opCount, int useDq0, int stagedJoints = 0);
void launchGenericStep(hipStream_t s, const ContactView& c, const JointView& j, const BodyView& g, const GroupTable& a, const GroupTable& b, const PersistView& pv,
					   const Op* ops, int opCount, const StepConsts& sc, s2amdContact* wire, int useDq0, int seamContacts, int seamJoints, size_t ldsBytes,
					   int stageJoints);

// stage kernels on resident arrays (narrowphase.hip, broadphase.hip; called by world.hip)
// summary: int[5] {separated pairs, active manifolds, zero/non-zero flips, point-count moves, enlarged shapes} (world.hip: WorldSummary)
void launchUpdateContacts(hipStream_t st, const s2amdBody* bodies, const float* origins, const s2amdShape* shapes, s2amdPairState* pairs,
						  s2amdContact* contacts, int contactCapacity, int32_t* status, uint8_t* pointBytes, int* summary, int* separatedSlots, const uint8_t* watched);
// stage 4 in one launch: refit per shape (origin recomputed from the body), origins + force reset per body, summary[4] += enlarged shapes
void launchStage4(hipStream_t st, s2amdBody* bodies, int bodyCapacity, s2amdShape* shapes, int shapeCapacity, float* origins, int* summary,
				  const unsigned int* stepFailed = nullptr);
// stage 1 on resident arrays: new pairs sorted by (A, B) into the host array outPairs; dJointed: sorted (min body << 32 | max body) keys
// the reference's broad-phase trees on the device (tree_mirror.hip): one s2DynamicTree's node array, the flags and leaf counts beside it,
// and the rebuild's scratch
struct TreeView
{
	s2amdTreeNode* nodes;
	int* flag;	 // s2TreeNode.enlarged of the internal nodes, as a word the enlarge pass can exchange
	int* leaves; // real leaves below every node: the traversal rank of a leaf is a sum of these
	int* state;	 // [0] root, [1] flagged nodes of the rebuild in flight (0 between rebuilds), [2] error
	int capacity;
	// the rebuild's scratch: flagged nodes below a flagged node and its flagged children still to report; the flagged nodes in pre-order;
	// the gathered leaves in depth-first order with their centres; the exchange partners of a split; reports at a new node
	int *acc, *pending, *oldPre, *leafIdx, *partner, *arrive;
	float *cx, *cy;
	int *tasks, *ready, *qstate; // the build's queue: {start, end, pre, parent, side} per task; published flags; {next, queued, leaves left, done}
};
struct TreeViews
{
	TreeView t[3];		 // by s2BodyType: static, kinematic, dynamic (include/solver2d/types.h:99-105)
	const int* refitPos; // shape -> position in the refit order (the move buffer's order), or null: the shape index
};
struct DeviceTrees;
void treesFree(s2amdSolver* s);
void treesForget(s2amdSolver* s);
bool treesActive(const s2amdSolver* s);
const TreeViews* treesViews(const s2amdSolver* s);
int treesSyncRefitOrder(s2amdSolver* s);
int worldWarmPairQuery(s2amdSolver* s); // world.hip
void launchTreeEnlarge(s2amdSolver* s, hipStream_t st, const unsigned int* stepFailed);
void launchTreeRebuild(s2amdSolver* s, hipStream_t st);
void treesJoin(s2amdSolver* s, hipStream_t st);
void launchOrderPairs(hipStream_t st, const TreeViews* views, const s2amdShape* shapes, const unsigned char* moved, const unsigned long long* keys,
					  const unsigned int* count, unsigned int cap, unsigned long long* ckeys, unsigned long long* out);
// the captured launch sequence of the resident pair query and its pinned read-back buffer (owned by the solver)
struct PairQueryGraph
{
	hipGraphExec_t exec = nullptr;
	unsigned long long key = 0, keySeen = 0;
	bool disabled = false;
	char* host = nullptr;
	void* countAt = nullptr; // where the query's counters live in the scratch block (zeroed when that changes)
	size_t outCapWanted = 0; // a query found more pairs than the device-side buffer held: the next one makes it this large
};
// something the ordered query runs between finding its pairs and ranking them (world.hip: the refit's tree enlarge pass)
struct PairQueryHook
{
	void (*fn)(void* arg, hipStream_t st);
	void* arg;
};
int findPairsResident(hipStream_t st, const s2amdShape* shapes, int shapeCapacity, int liveShapes, const s2amdPairState* pairs, int contactCapacity,
					  const unsigned long long* jointed, int jointedCount, int32_t* outPairs, int32_t pairCapacity, int32_t* pairCount, void** scratch,
					  size_t* scratchBytes, unsigned long long* sortedPairKeys, bool* sortedPairKeysValid, PairQueryGraph* cache, int mode = 0,
					  const unsigned long long* pairLog = nullptr, const int* pairLogSlots = nullptr, const TreeViews* trees = nullptr,
					  const PairQueryHook* hook = nullptr);
#define S2_PAIR_LOG_ENTRIES 255 // broadphase.hip: S2_PAIR_LOG_CAPACITY
#define S2_PAIRS_FULL 0
#define S2_PAIRS_WARM 1
#define S2_PAIRS_ENQUEUE 2
#define S2_PAIRS_COLLECT 3

// One empty kernel per translation unit with kernels: the HIP runtime loads a TU's code object when its first kernel is launched -- 2 to 14 ms
// each at the sizes of this library, which the first steps of a new process, the first structure request, the first flip used to pay
// (r5: the first slotBytesKernel launch, 7 ms inside step 33 of the wrecking-ball loop).  s2amd_create launches them all once.
#define S2_DEFINE_WARM(name)                                                                                                     \
	__global__ void s2WarmKernel_##name() {}                                                                                    \
	void s2Warm_##name(hipStream_t st) { s2WarmKernel_##name<<<dim3(1), dim3(64), 0, st>>>(); }
void s2Warm_contact_kernels(hipStream_t st);
void s2Warm_body_kernels(hipStream_t st);
void s2Warm_joint_kernels(hipStream_t st);
void s2Warm_group_kernel(hipStream_t st);
void s2Warm_strip_kernel(hipStream_t st);
void s2Warm_pair_kernel(hipStream_t st);
void s2Warm_wide_kernel(hipStream_t st);
void s2Warm_generic_kernel(hipStream_t st);
void s2Warm_broadphase(hipStream_t st);
void s2Warm_narrowphase(hipStream_t st);
void s2Warm_tree_mirror(hipStream_t st);
void s2Warm_structure(hipStream_t st);
void s2Warm_world(hipStream_t st);
void s2Warm_sharded(hipStream_t st);
void s2Warm_jacobi_kernel(hipStream_t st);
void s2WarmScratch(hipStream_t st); // wide_kernel.hip: the queue's scratch memory allocated before a step needs it
