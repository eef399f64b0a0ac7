// Internal declarations shared by the host-side translation units of libs2amd.so:
//   solver.cpp            the C-ABI entry points (include/solver2d_amd.h)
//   solver_step.cpp       upload / step / download: shadows, hipGraph capture and replay, timing
//   solver_executor.h     the launch sequence of one step (global path, LDS groups, strips, persistent step)
//   solver_plan.cpp       the ten reference drivers as lists of Ops
//   solver_structure.cpp  islands, groups, strips, colour batches and their device tables
//   graph_coloring.cpp    greedy colouring, batch formation
#pragma once

#include "launch.h"
#include "s2_device.h"

#include "solver2d_amd.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

// sets the thread's last error text and returns `code` (solver.cpp)
int s2amdFail(int code, const std::string& msg);
inline int fail(int code, const std::string& msg) { return s2amdFail(code, msg); }

#define S2_HUB_DEGREE 12
// Hub bodies and strips (solver_structure.cpp: cutStrips).  A strip sweeps the constraints of one body in as many colour rounds as the
// body has constraints, and every strip waits on that strip's hand-offs: a sweep over the strips costs about
//     max(degree of any writable body inside the strips) x S2_COST_STRIP_ROUND_US        (1.4 us per round: the op interpreter, r3),
// the same sweep on the colour batches with the wave-walked tail about
//     S2_COST_BATCH_COLOURS x S2_COST_LAUNCH_US  +  (sum of the hubs' degrees) x S2_COST_TAIL_VISIT_US
// (a dozen colour launches of ~2 us in a graph; 0.38 us per tail visit: Tumbler 10k).  The graph keeps its strips
// when the first is the smaller -- for ONE hub that is a degree of 23 or less; the Tumbler's drum (238) is 333 us against 114.
// (Through round 3 this was a constant, 48.)  The unit costs are per device architecture (solver.cpp: hubCostsFor picks the row at
// s2amd_create from the device's gcnArchName; options "cost_strip_round_ns" / "cost_launch_ns" / "cost_tail_visit_ns" override them):
// the defaults below are the MI355X's (gfx950), measured on the Tumbler and the base-200 pyramid.
#define S2_COST_STRIP_ROUND_US 1.4f
#define S2_COST_LAUNCH_US 2.0f
#define S2_COST_BATCH_COLOURS 12
#define S2_COST_TAIL_VISIT_US 0.38f
struct HubCosts
{
	float stripRoundUs = S2_COST_STRIP_ROUND_US, launchUs = S2_COST_LAUNCH_US, tailVisitUs = S2_COST_TAIL_VISIT_US;
};

#define HIP_TRY(expr)                                                                                                            \
	do                                                                                                                           \
	{                                                                                                                            \
		hipError_t _e = (expr);                                                                                                  \
		if (_e != hipSuccess)                                                                                                    \
		{                                                                                                                        \
			return fail(S2AMD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                                     \
		}                                                                                                                        \
	} while (0)

double nowMs();

// Device memory of the worker threads (solver_async.cpp).  hipMalloc and hipFree synchronise the device and serialise with every
// other HIP call of the process: a worker that builds seven structures on a copy of the solver -- some twenty allocations each -- and
// the thread that frees the copy afterwards stalled the stepping thread for 4 ms at a time (measured: wreck-200, the step after a
// search was dropped).  So on those threads (devPoolThread(true)) a DevBuf takes its memory from a pool of blocks earlier copies
// have given back and returns it there; the stepping thread allocates and frees as ever.  The pool is emptied by s2amd_destroy.
// (the same for the copies' streams and pinned staging buffers: creating and destroying those stalls the stepping thread as well)
hipStream_t workerStreamTake();
void workerStreamGive(hipStream_t s);
void* pinnedPoolTake(size_t need, size_t* got);
void pinnedPoolGive(void* p, size_t bytes);
void devPoolThread(bool on);
bool devPoolOn();
void* devPoolTake(size_t need, size_t* got);
bool devPoolGive(void* p, size_t bytes);
void devPoolDrain(); // (the calling thread's current device's pool)
void devPoolSolverCreated(int device);
bool devPoolSolverDestroyed(int device);
bool devPoolNoSolverLeft();
void spareClonesRelease(); // solver_async.cpp: the retired copies of solvers kept for the next structure request

// growable raw device allocation
struct DevBuf
{
	void* p = nullptr;
	size_t bytes = 0;

	int ensure(size_t need, bool* grew = nullptr)
	{
		if (need <= bytes)
		{
			return S2AMD_OK;
		}
		size_t want = std::max(need, bytes + bytes / 2);
		want = (want + 255) & ~size_t(255);
		void* np = nullptr;
		if (devPoolOn())
		{
			size_t got = 0;
			np = devPoolTake(want, &got);
			if (np)
			{
				want = got;
			}
		}
		if (!np)
		{
			HIP_TRY(hipMalloc(&np, want));
		}
		if (p && !(devPoolOn() && devPoolGive(p, bytes)))
		{
			(void)hipFree(p);
		}
		p = np;
		bytes = want;
		if (grew)
		{
			*grew = true;
		}
		return S2AMD_OK;
	}
	void release()
	{
		if (p && !(devPoolOn() && devPoolGive(p, bytes)))
		{
			(void)hipFree(p);
		}
		p = nullptr;
		bytes = 0;
	}
};


bool isPositionSolver(int type);
bool rotIsFixedPoint(float s, float c);
int colorGraph(const std::vector<int>& ea, const std::vector<int>& eb, const std::vector<uint8_t>& conflict, int bodyCount,
			   std::vector<int>& color, int balanced = 0, std::vector<uint64_t>* bitsOut = nullptr);
void sortByColor(const std::vector<int>& ids, const std::vector<int>& color, int colorCount, std::vector<int>& order, std::vector<int>& offsets);
bool makeBatches(const std::vector<int>& colorOffsets, std::vector<int>& batchOffsets, bool allowTail = true, int tinyColor = 32);
uint64_t fnv(uint64_t h, const void* data, size_t n);

// One sweepable family (contacts or joints): order, colour batches, LDS groups
struct SweepSet
{
	std::vector<int> order;		   // k -> wire index (global part first, then group by group)
	std::vector<int> colorOffsets; // every (part, colour) batch as a range of k: API + validity tests
	// global part: launch batches (parallel colours, then optionally one sequential tail)
	std::vector<int> batchOffsets;
	bool hasTail = false;
	int globalCount = 0;
	int stripCount = 0;		 // constraints that live in strip groups (phase A interiors + phase B seams)
	int seamCount = 0;		 // ... of which seams
	std::vector<int2> local; // k -> group-local body slots (groups and the global tail)
};

struct HostGroupTable
{
	std::vector<int> bodyOffsets{0}, bodyIds, cBatchOffsets{0}, jBatchOffsets{0};
	std::vector<int4> cBatches, jBatches;
	int maxBodies = 0;
	int count() const { return (int)bodyOffsets.size() - 1; }
	void clear()
	{
		bodyOffsets.assign(1, 0);
		cBatchOffsets.assign(1, 0);
		jBatchOffsets.assign(1, 0);
		bodyIds.clear();
		cBatches.clear();
		jBatches.clear();
		maxBodies = 0;
	}
};

struct DeviceGroupTable
{
	DevBuf buf;
	GroupTable view{};
	int maxBodies = 0;
	int spareIdsBase = 0, spareIdsCount = 0; // entries of view.bodyIds behind the table's own (IncrementalStrips: relocated body lists)
};

// The launch sequence of one s2Solve_* driver, recorded once per parameter set
struct StepPlan
{
	bool valid = false;
	s2amdStepParams params{};
	StepConsts sc{};
	bool earlyOut = false;
	float unpackH = 0.0f;
	int prepContacts = -1;
	float prepH = 0.0f, prepHertz = 0.0f;
	int prepJoints = -1;
	float jprepH = 0.0f, jprepHertz = 0.0f;
	int jprepWarm = 0;
	std::vector<Op> ops;
	int storeKind = STORE_PLAIN;
	float storeScale = 0.0f;
	int solveSweeps = 0;
	bool usesDq0 = false;
};

// Host mirror of the GLOBAL part's tables (colour batches over HBM-resident bodies), kept between steps so that a created
// contact can be given a place in the sweep order WITHOUT rebuilding the structure (solver_incremental.cpp):
//   * every parallel colour batch is laid out with slack -- free positions whose contactIndex is -1 (an empty record, no
//     body touched); a new constraint takes a free position of the lowest colour unused on both of its writable bodies;
//   * the body -> incident constraints lists (body-centric warm start, Jacobi apply) carry per-body slack; an entry is
//     inserted at its place in sweep order;
//   * the launch sequence does not change (same batch ranges, same grid sizes): the captured hipGraph stays valid.  The few
//     words that do change travel as one patch list (address, value) applied by one small kernel.
// A contact that does not fit (no colour with a free position, a body owned by an LDS group or a strip, list capacity
// exhausted) falls back to the full rebuild.
struct IncrementalGlobal
{
	bool valid = false;
	int solverClass = -1;				// colouring class the bits were built for (0 velocity, 1 position)
	int parallelBatches = 0;			// colour batches that run as one launch each (the sequential tail, if any, comes after); the last
										// `spare` of them start out empty: colours for bodies whose ordinary colours are all taken
	bool ignoreColours = false;			// the structure was built for s2Solve_Jacobi: its contact pass writes no body, any free position will do
	bool colourFreePlaced = false;		// ... and a contact was placed that way: no other solver may run on this structure
	std::vector<int> colorIdOfBatch;	// bit index of a batch in colorBits (a spare batch gets an id above every colour of the tail)
	std::vector<int> batchBegin, batchEnd; // position range of every parallel batch, slack included
	std::vector<std::vector<int>> freePositions; // per batch, descending: pop_back() hands out the lowest free position
	std::vector<uint64_t> colorBits;	// 4 words per body: colours in use on the body (global part)
	std::vector<int> positionOfSlot;	// contact slot -> position k in the sweep order, -1
	std::vector<int> colorOfPosition;	// position -> colour batch (parallel batches only), -1
	// adjacency mirror
	std::vector<int2> adjRange;			// {begin, count} per body
	std::vector<int> adjCapacity;		// per body
	std::vector<int> adjList;			// device-sized mirror
	int adjUsed = 0;					// first never-used entry of adjList
	int adjFailure = 0;					// why adjInsert last said no: 1 no room to move a list to, 2 the list of heavy bodies is full
	std::vector<int> heavy;				// [0] = count, then body slots; size = capacity + 1
	// patch list of the current call: {address lo, address hi, value, 0}
	std::vector<uint4> patches;
	// The sequential tail (the hubs' constraints and the tiny colours: one workgroup stages the bodies they touch and walks them in
	// order -- group_kernel.hip: walkTail) takes created contacts too: it ends in S2_TAIL_SLACK free positions, and a contact for
	// which no parallel colour is free (a contact of the Tumbler's drum: the drum uses every one) takes the lowest of them -- any
	// position of a sequential sweep is a valid one.  A body the tail does not stage yet is appended to its body list
	// (S2_TAIL_BODY_SLACK more fit its LDS and its table).  Entries of the tail can be removed like any other.
	int tailBegin = 0, tailEnd = 0;				// positions of the tail, slack included (0, 0: no tail)
	std::vector<int> tailFree;					// descending: pop_back() hands out the lowest
	std::unordered_map<int, int> tailBodySlot;	// body -> local slot in the tail's body list
	int tailBodyCount = 0, tailBodyCapacity = 0;
	long tailPlaced = 0;
	long inserted = 0, removed = 0, fallbacks = 0;
	bool placedInGlobalPart = false; // the last incrementalApply put something into a colour batch of the global part (not only into strips)
};

// Created contacts in the STRIPS of a big island (the persistent step kernels' tables).  A colour round of a strip or seam is a
// range of positions k, one per lane; the ranges are laid out with slack (free positions: contactIndex -1, an empty record, a
// lane that does nothing), so a created contact between two bodies one strip owns -- or between the two sides of a seam whose
// body lists already hold both -- is given a free position of a round that is unused on both bodies: three patched words
// (contactIndex[k], localBodies[k]), no table rebuilt, the persistent kernel keeps running.  Anything else (a body that joins
// the island, bodies of non-adjacent strips, a round or colour short) is a rebuild.
struct IncrementalStrips
{
	bool valid = false;
	bool touched = false; // something was placed or removed since the build: the multi-launch strip tables (warm-start slots) are stale
	int base = 0, end = 0; // [base, end): the strips' positions in contacts.order
	struct Round
	{
		int table, group, round; // table 0: strip interiors (hStripA), 1: seams (hStripB)
		std::vector<int> freePositions; // descending: pop_back() hands out the lowest
	};
	std::vector<Round> rounds;
	std::vector<int> roundOfPosition;	// [end - base] -> index into rounds
	std::vector<std::vector<int>> roundsOf[2]; // per group: its open rounds in `rounds`, round 0 first (as built: contiguous; opened spare rounds follow)
	std::vector<std::vector<int>> spareOf[2];  // per group: its CLOSED spare rounds in `rounds` (no batch of a descriptor covers them yet), next round first
	std::vector<int> seamOfGroup;		// seam group -> the seam (between strips i and i + 1) it is
	std::vector<int> ownerStrip, ownerSlot; // per body: the strip that owns it and its local slot there, -1
	std::vector<std::unordered_map<int, int>> replicaSlot; // per strip: read-only body -> local slot
	std::vector<std::unordered_map<int, int>> seamSlot;	   // per seam group: body -> local slot
	std::vector<int> seamGroupOf;		// seam between strips i and i + 1 -> seam group, -1
	std::vector<uint32_t> roundMask[2]; // per (body offset of the group + local slot): rounds in use on a writable body
	std::vector<int> bodyOffset[2];		// per group: its first entry in roundMask
	std::vector<int> positionOfSlot;	// contact slot -> strip position, -1
	long placed = 0, roundsOpened = 0;
	// rounds a strip / a seam may have OPEN under the solver these strips were built for: s2Solve_SoftStep's resident kernel exists in
	// the <3, 2> layout only (wide_kernel.hip: wideExtraRecords) -- a seventh interior or third seam round opened for a created contact
	// left it without a kernel, and the step rebuilt the strips instead (wreck-200 under SoftStep, r6: 50 such steps of 200, 5-40 ms each)
	int roundLimit[2] = {S2_STRIP_ROUNDS_MAX, S2_PERSIST_B_ROUNDS};
	// strips the OP INTERPRETER sweeps (generic_kernel.hip: every solver family but the soft ones): it finds its rounds and body lists in the
	// group tables, which placement does not patch -- so a created contact takes a free position of a round the build laid out, between two
	// bodies the strip or seam already lists, or nothing: no adopted body, no extended seam, no opened round, no overflow position
	bool takeOnly = false;
	// A body that JOINS an island (a ball thrown into the pile).  A writable body without a single constraint in
	// the strips is owned by whichever strip the build put it in; when its first contact is with a body of another strip it MOVES there
	// -- the receiving strip's body list is written again behind the table (one more entry; the descriptor's two words follow it), the
	// old entry loses its OWNED flag (a read-only copy nobody looks at), the imports' LDS slots behind the own bodies shift by one
	// (the strip's two remap ranges) -- and the contact is an interior one like any other.  The budgets of a build (LDS records, body
	// chunks) include S2_STRIP_ADOPT_SLACK more bodies per strip for it.
	std::vector<int> stripBodyBase, stripBodyCount; // per strip: its list in the device table NOW (GroupTable::bodyIds of dStripA)
	std::vector<int> stripListCapacity;				// ... entries its place can hold (a relocated list: + the slack)
	std::vector<std::vector<int>> movedList;		// per strip: the host copy of a relocated list (empty: hStripA's CSR range)
	std::vector<int> adoptedBy;						// per strip: bodies it has adopted since the build
	int spareIdsNext = 0, spareIdsEnd = 0;			// free entries behind the table in that buffer
	long adopted = 0;
	// ... and a body that a SEAM has to carry from now on (the ball, now a body of strip i, touches a box of strip i + 1; or two boxes
	// of neighbouring strips that never faced each other): it becomes the seam's next local body -- one more export of its owner, one
	// more import of the neighbour (whose later imports move one LDS slot on), one more entry of both remap ranges.  The build leaves
	// room for S2_STRIP_ADOPT_SLACK more bodies in every such range and in the hand-off buffers.
	std::vector<int> seamBodyCount;		   // per seam group: local bodies it has now
	std::vector<int> seamExtra[2];		   // per seam group: bodies appended on its left / right side since the build
	long seamBodiesAdded = 0;
	// ... and a contact that fits NOWHERE in the strips (a ball that touches boxes two strips apart: a body on two seams is what the
	// partition cannot have).  Until round 5 that was a structure build in the step that found it -- 5 ms on the caller's thread.  Now
	// the sweep order ends in S2_OVERFLOW_SLACK free positions BEHIND the strips, each a colour batch of its own; such a contact takes
	// the lowest of them, and while any of them is in use the step runs SLICED (Executor::runPersistentSliced): the persistent kernel
	// is launched once per sweep with that sweep's ops -- it stages and writes back its bodies and impulses every time --, and behind
	// each launch the overflow contacts are swept one by one on the bodies in HBM by the colour-batch kernels (any position of a
	// sequential sweep is a valid one: the oracle sweeps in it).  ~75 launches and ~0.5 ms per step instead of 3 and 0.15 -- for the
	// dozen steps a worker thread needs to build a structure that holds the contact (solver_async.cpp), adopted at a step boundary.
	int overflowBegin = 0, overflowEnd = 0; // positions of the overflow region in contacts.order (0, 0: none)
	std::vector<int> overflowFree;			// descending: pop_back() hands out the lowest
	int overflowUsed = 0;
	long overflowPlaced = 0;
	// ... round 5, later: the overflow contacts swept INSIDE the persistent launch by one more workgroup (wide_kernel.hip:
	// wideOverflowWorker; PersistView::overflowBodies): the bodies the contacts in use touch -- pool slot, bit 30: nothing writes it, -1:
	// free entry --, at most S2_OVERFLOW_BODIES; an overflow contact's `local` pair indexes this list.  The device copy
	// (SolverStructure::dOverflowBodies) is kept up to date with patched words.
	std::vector<int> overflowBodyIds;
};
#define S2_OVERFLOW_SLACK 32
#define S2_STRIP_ADOPT_SLACK 8
#define S2_TAIL_SLACK 64
#define S2_GROUP_PATIENCE_MIN_POSITIONS 16384 // sweep positions (contacts + joints) from which on SolverRest::groupPatienceNow applies
#define S2_TAIL_BODY_SLACK 32

// What a STRUCTURE BUILD produces and the incremental placement keeps up to date: the host's picture of the constraint graph as the
// structure knows it, every table derived from it and their device copies, the SoA families (carved per structure), the captured step
// graph.  Kept apart from the rest of the solver so that a build can run on a copy of the solver in a worker thread while the steps go
// on with the structure they have, and its result be adopted by swapping this part (solver_async.cpp).
struct SolverStructure
{
	// host shadows of the graph structure (refreshed by every upload)
	// The structure (islands, colours, strips) is built over EVERY contact slot that can become a constraint -- a live pair of
	// the world chain, a slot with two distinct live bodies otherwise -- whether its manifold has points this step or not
	// (hContactEdge).  A manifold without points is a no-op in every sweep (no point loop iteration, no body store), so a
	// manifold that gains or loses its points changes NOTHING on the host: same tables, same launch sequence, same hipGraph.
	// Only a contact slot that appears, disappears or changes its bodies changes the graph (src/contact.c:137-229).
	std::vector<int> hContactA, hContactB;
	std::vector<uint8_t> hContactEdge;
	std::vector<uint8_t> hContactDead; // in the structure, but its contact has been destroyed since: dropped at the next rebuild
	// HUB bodies.  A long shape that rotates (the Tumbler's drum walls) has a world-space box that overlaps the fat boxes of
	// hundreds of bodies it never touches: hundreds of POTENTIAL constraints on one writable body, each of which would cost a
	// colour of its own.  So for a body with more than S2_HUB_DEGREE potential constraints only the manifolds WITH points are
	// structural (as the reference gathers them), and a manifold on such a body that gains or loses its points changes the
	// graph like a created or destroyed contact.  hContactWatched marks the live slots with a hub end; the device counts
	// their flips (stage 3), and a world with watched slots reads that counter back BEFORE its solve is enqueued.
	std::vector<uint8_t> hBodyHub, hContactWatched;
	int watchedCount = 0;
	DevBuf dWatched;
	std::vector<uint32_t> hBodyFlagsFinal; // ... + S2F_IN_GROUP as the last structure build uploaded them
	std::vector<uint8_t> hBodyLdsOwned; // per body: an LDS group or a resident island (not a strip) owns it
	DevBuf dBodyFlags;

	// working SoA
	DevBuf soaBodies, soaContacts, soaJoints, dContactIndex, dJointIndex, dContactLocal, dJointLocal, dAdjOffsets, dAdjList, dAdjHeavy;
	int adjHeavyCapacity = 0; // entries the heavy-body list (more than S2_HEAVY_DEGREE adjacency entries) can hold: sizes the launch
	IncrementalGlobal inc;
	uint4* hostPatches = nullptr; // pinned staging of inc.patches
	size_t hostPatchCapacity = 0;
	DevBuf dPatches;
	int spareColours = 0;	// empty colour batches a build adds to the global part: 0 until a created contact found every colour of its
							// bodies taken (a dense pile: a box inside a pyramid uses all six), then 2 -- two more launches per sweep
	DevBuf dJointAdjRange, dJointAdjList; // body -> incident global joints in sweep order (body-centric joint warm start)
	bool jointAdjValid = false;
	int slackPositions = 0; // free positions of the global part's slack layout
	bool slackBumped = false; // (since the last rebuild)
	int slackAtBuild = 0;	  // free positions the last rebuild laid out
	int slackShift = 0; // the structure's slack (free positions per colour batch, room for growing incidence lists) times 2^this: raised when a rebuild was forced by used-up slack
	const char* dirtyReason = ""; // S2AMD_DEBUG_PREP: what made the last rebuild necessary
	bool watchedDirty = false; // hContactWatched changed since it was last copied to dWatched
	BodyView bv{};
	ContactView cv{};
	JointView jv{};
	uint64_t layoutGeneration = 0;
	int bodySoaCap = 0, contactSoaCap = 0, jointSoaCap = 0;

	// structure of the last step
	SweepSet contacts, joints;
	HostGroupTable hGroups, hContactTail, hJointTail, hStripA, hStripB;
	std::vector<int> hPersistRemap;			  // host copies of PersistView::remap and ::descs (IncrementalStrips: a strip that adopts a body)
	std::vector<PersistDesc> hPersistDescs;
	DeviceGroupTable dGroups, dContactTail, dJointTail, dStripA, dStripB;
	// resident islands (strip_kernel.hip: islandStepKernel): LDS groups whose constraints stay in registers for the whole step
	HostGroupTable hResident;
	DeviceGroupTable dResident;
	DevBuf dResidentDesc, dResidentOps;
	StripTableView residentView{};
	int residentRounds = 0;
	int residentK0 = 0, residentK1 = 0; // their range in contacts.order
	bool residentRejected = false; // some group's colouring needs more rounds than the kernel holds: plain LDS groups for this graph
	uint64_t residentOpsGeneration = ~0ull;
	int residentOpCount = 0;
	// lean strip tables (strip_kernel.hip): descriptors of both phases, warm-start slots of phase A
	DevBuf dStripLean;
	StripTableView leanA{}, leanB{};
	bool leanAValid = false, leanBValid = false;
	// persistent strip step (strip_kernel.hip: stripStepKernel)
	DevBuf dPersist, dGranules, dOverflowBodies;
	PersistView persist{};
	bool persistValid = false;
	// ... as an op interpreter for every solver family and joints (generic_kernel.hip: genericStepKernel): the same partition,
	// import / export lists and hand-off buffers; seams swept once, by their left strip
	bool genericValid = false;
	int genericJoints = 0; // the most joints of a strip and the seam it sweeps
	int genericBodies = 0, genericSeamBodies = 0, genericExports = 0; // the most staged bodies / seam-group bodies / exported bodies of a strip (LDS)
	int residentAllTwoPoints = 0; // every constraint of the resident islands has two manifold points (wide_kernel.hip: the POINTS == 2 variants)
	int persistK0 = 0, persistK1 = 0; // the strip constraints' range in contacts.order (persist.allTwoPoints is recomputed over it)
	int persistRecordsWide = 0; // LDS records when a seam constraint takes 10 records (every kind but TGS_Soft's)
	DevBuf dPersistOps;
	int persistOpCount = 0;
	// s2Solve_Jacobi as one persistent launch over blocks of bodies (jacobi_kernel.hip; tables: solver_jacobi.cpp)
	DevBuf dJacobi, dJacobiGran;
	JacobiView jacobi{};
	int jacobiDeferred = 0; // steps until the persistent launch's tables are made for a structure a world chain rebuilt (solver_structure.cpp: finish)
	bool jacobiValid = false;
	size_t jacobiGranBytes = 0;
	int jacobiMaxOwned = 0, jacobiMaxImports = 0, jacobiMaxConstraints = 0;
	uint64_t persistOpsGeneration = ~0ull, persistOpsStructure = ~0ull;
	size_t granuleBytes = 0;
	long selfStepsSinceReset = 0; // self-contained strip steps since the commit counter was last zeroed (doStep: every 2^20)
	IncrementalStrips stripInc;
	float stripScaleFound = 0.0f;   // the strip width (x strip_bodies) the last search settled on, 0: none yet (a new world forgets it)
	bool stripRetryPending = false; // ... postponed until the graph has been quiet for 32 steps
	DevBuf dMsg;
	MsgView msg{};
	bool msgTablesValid = false; // the global part is contact-only and has no sequential tail
	int looseBodies = 0; // live non-static bodies that no LDS group owns
	int orderSolverClass = -1; // 0 velocity colouring, 1 position colouring
	bool orderGrouped = false;
	bool orderLdsGroups = false; // ... with small islands in LDS groups (SolverRest::groupPatienceNow)
	bool orderResident = false; // the groups were laid out for the resident-island kernel where they fit
	bool orderColourless = false; // built for s2Solve_Jacobi: the global contact part is one batch in pool order, no colours
	bool orderStrips = false;
	int orderStripBodies = 0; // the strip width the structure was cut with (StructureBuild::stripBodiesFor)
	bool stripsNeedOneLaunch = false; // a moving read-only body is shared between strips: persistent kernel or no strips at all
	bool stripsRejected = false; // this graph's strip partition fits no strip kernel: colour batches until the graph changes
	int stripsJudgedForClass = -1; // the colouring class (0 velocity, 1 position sweeps) the two verdicts were reached under: the other class's writable bodies differ
	bool stripsHopeless = false; // ... for a reason no other strip width would change (a body the sweeps write that no strip can own): no search
	bool adjValid = false;
	bool structureDirty = true;
	uint64_t structureGeneration = 0;
	int stripScaleFoundFor = 0;	 // the strip width stripScaleFound was searched with

	// graph cache
	hipGraph_t graph = nullptr;
	hipGraphExec_t graphExec = nullptr;
	uint64_t graphKey = 0;
	uint64_t graphKeySeen = 0; // the launch sequence of the last step that was enqueued directly
	int graphKeySeenLaunches = 0; // ... and how many launches it was
	int graphLaunches = 0;
};

struct AsyncBuild;

// ... and the rest: the device, the wire arrays and the world chain's arrays, what the host knows of them, the options, the plan.
struct SolverRest
{
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t evBegin = nullptr, evEnd = nullptr;
	hipEvent_t evExport[4] = {nullptr, nullptr, nullptr, nullptr}; // s2amd_export_poses_async / s2amd_export_wait
	// side streams: independent prologue / epilogue kernels become parallel branches of the captured graph
	hipStream_t side[2] = {nullptr, nullptr};
	hipEvent_t evFork[2] = {nullptr, nullptr}, evJoin[4] = {nullptr, nullptr, nullptr, nullptr};
	int optAsync = 0; // s2amd_step_resident returns after enqueueing; s2amd_synchronize collects errors
	bool constraintIndexInPrologue = false;
	int optFork = 0; // measured slower on MI355X (multi-branch graph replay costs more than the serial kernels): off

	// wire arrays resident on the device
	DevBuf dBodies, dContacts, dJoints, dBodiesSaved;
	int bodyCapacity = 0, contactCapacity = 0, jointCapacity = 0;
	bool resident = false;
	bool savedValid = false;
	int savedBodyCapacity = 0; // body slots of the snapshot in dBodiesSaved

	// resident world (world.hip): the arrays of stages 3 and 4 beside the solver's wire arrays
	DevBuf dShapes, dPairs, dOrigins, dStatus, dPointBytes, dWorldSummary, dJointedKeys, dContactStage, dPairScratch, dPairKeys;
	PairQueryGraph pairQuery; // the resident pair query's captured launch sequence (broadphase.hip: findPairsResident)
	bool pairKeysValid = false; // dPairKeys holds the sorted (shape, shape) keys of the live pair slots
	int shapeCapacity = 0, liveShapes = 0, jointedCount = 0;
	bool worldResident = false;
	int* hostWorldSummary = nullptr; // pinned: the per-step counters of both stages
	uint8_t* hostSlotStage = nullptr; // pinned, contactCapacity bytes: where the per-slot byte arrays land (syncDeadSlots, fetchPointCounts) -- a copy
									  // straight into a std::vector's pageable memory cost 0.8 ms per call at 140k slots, 7.5 ms the first time (r5)
	size_t hostSlotStageBytes = 0;
	// s2amd_world_step's speculative read-back for s2amd_world_download_step (a caller that set a refit order will ask for the poses
	// and the re-inflated boxes right after the step): enqueued behind stage 4, landed by the step's own synchronisation
	char* hostStepBack = nullptr; // pinned: {count, 0, 0, 0}, `stepBackBoxes` s2amdMovedBox records, then bodyCapacity poses at stepBackPoseOffset
	size_t hostStepBackBytes = 0, stepBackPoseOffset = 0;
	int stepBackBoxes = 0;
	bool stepBackValid = false;
	bool stepBackListFresh = false; // this step's list of re-inflated shapes is on the device (enqueueStepBack ran; launchTreeEnlarge reads it)
	int optStepReadback = 1;
	std::vector<uint8_t> hPointBytes;
	std::vector<uint8_t> hShapeMovable; // world chain: live shapes of non-static bodies (what s2amd_world_set_refit_order must cover)
	int movableShapes = 0;
	std::vector<int> hContactPoints; // manifold point counts as of the last host upload / the last fetchPointCounts (see pointsKnown)
	bool deadUnknown = false;		   // pairs separated on the device since the host last looked (syncDeadSlots, world.hip)
	bool pointsKnown = false; // hContactPoints is current: no stage 3 has recomputed manifolds on the device since the upload
	bool pointCountsFresh = false; // ... or fetchPointCounts has read them since this step's stage 3 (cleared when the next one is enqueued)
	int optPrebuildSolver = -1; // "prebuild_solver": s2amd_world_upload builds the structure for this solver type (strips at once) so that the first step has nothing to build; -1: the first step does
	int activeContacts = 0;	  // manifolds with points this step (host count, or the device's counter in the world chain)
	bool lastStepWroteIndex = false; // the last solve's driver writes manifold.constraintIndex (all but XPBD's early-out and Block)
	DevBuf dScanTmp;
	std::vector<int> hJointType, hJointA, hJointB;
	std::vector<uint32_t> hBodyFlags; // S2F_WRITE_VEL / S2F_WRITE_POS from the wire bodies
	std::vector<uint8_t> hBodyLive, hBodyStatic;
	DevBuf dOps; // the plan's op list (solver_step.cpp: doStep)
	long placedTotal = 0;	// created contacts placed without a rebuild, since s2amd_create
	DevBuf dSeparated;		// world chain: the pair slots stage 3 freed this step
	DevBuf dPairLog;		// the pairs created since the sorted directory of the pair set was made: [0] entries, keys ascending, then their slots (broadphase.hip: PairSetView)
	unsigned long long* hostPairLog = nullptr; // ... its pinned host copy (the host appends, the device reads)
	bool pairLogDirty = false;
	DevBuf dShapeBoxes;		// world chain: s2amd_world_download_boxes' staging
	DevBuf dRefitOrder, dStepBack; // s2amd_world_set_refit_order; staging of s2amd_world_download_step {count, moved boxes} and the poses
	int refitOrderCount = 0;
	int optTreeStream = 1;		  // "tree_stream": the device trees' rebuild on a stream of its own beside stage 3 and the solve (0: on the step's stream; tree_mirror.hip)
	DeviceTrees* trees = nullptr; // the reference's broad-phase trees on the device (tree_mirror.hip; s2amd_world_set_tree)
	int lastMovedCount = 0; // enlarged shapes of the last s2amd_world_step
	DevBuf dSlotBytes;		// world chain: one byte per pair slot for a structure build (world.hip: slotBytesKernel)
	std::vector<uint8_t> hSlotBytes;
	bool slotBytesFresh = false; // hSlotBytes is of the state the device is in right now (cleared by every world call that changes it)
	std::vector<int32_t> hSeparated;
	int optIncremental = 1; // created contacts are placed into the existing structure when they fit (0: always rebuild)
	// A created contact that cannot be placed (an LDS group or a strip owns one of its bodies, or one of them is a hub) and has
	// no manifold points yet is only WATCHED: no entry in the structure -- it would be a no-op there -- until stage 3 finds
	// its first points, which then counts as the change of the graph (option "defer", 0: rebuild when it is created)
	int optDefer = 1;
	int optIslandResident = 1;
	int optStripLean = 1;
	int optStageJoints = 1; // "stage_joints": the op interpreter keeps a strip's joint records in LDS when they fit
	int optSelfContained = 1; // "self_contained": a world of resident islands only is stepped by their kernel alone (no body prologue / epilogue launch)
	int optGeneric = 1;
	int optFreeBodyGroups = 1; // "free_body_groups": constraint-free bodies next to groups / strips form LDS groups instead of global launches
	int optPersist = 1;
	int optSeamRegs = 1;
	int optWide = 1;	  // TGS_Soft's persistent step on 512 threads per strip (wide_kernel.hip) where the partition fits
	// Two forms of that kernel that were built to the verdict of round 3 and measured no faster on the MI355X (the work
	// they remove from launches or barriers comes back as latency-bound work inside the persistent kernel); tested options, off by default:
	int optWideBodyWarm = 0;	  // "strip_body_warm": s2WarmStartContacts as one body-centric pass (wide_kernel.hip: S2_WIDE_BODYWARM) where its term table fits LDS
	int optSelfContainedStrips = 0; // "self_contained_strips": the kernel is the step's prologue and epilogue too -- ONE launch per step (S2_WIDE_SELF)
	int optStripSlack = 1; // strip and seam rounds are laid out with free positions for created contacts (solver_incremental.cpp)
	int optPairLanes = 0; // two lanes per constraint (pair_kernel.hip; measured no faster: kept as an option); 0: one lane per constraint
	int optStripRetry = 1; // try other strip widths when the partition needs the 8-round kernel variant
	int optPersistDebug = 0;
	int optPersistSpinLimit = 1 << 21;
	bool persistFailed = false; // a hand-off timed out once (workgroups not co-resident: a shared GPU): multi-launch strips from then on
	int persistFallbacks = 0;
	// ... but not for ever: after `persistRetryAfter` further steps the one-launch kernels get another chance (the GPU may have
	// been shared only for a while); a retry that times out again doubles the wait
	int optPersistRetry = 256; // "persist_retry": steps on the fallback path before the first retry, 0 = never
	int persistRetryAfter = 256, persistFailedAge = 0;
	int cuCount = 0;
	HubCosts hubCosts; // the hub rule's unit costs on this device (solver_structure.cpp: cutStrips)
	unsigned int* hostError = nullptr; // pinned, device-visible: a hand-off timed out
	unsigned long long* hostTimes = nullptr; // S2AMD_DEBUG_TIMES: pinned [256] phase time stamps of one workgroup
	int optMessage = 0;	 // measured slower than the plain gather on MI355X: off by default
	int optBodyWarm = 1; // body-centric contact warm start (one launch per sweep instead of one per colour)
	bool indexInWire = false; // the resident wire contacts hold the current gather index as manifold.constraintIndex
	int graphAge = 0;		  // steps solved since the constraint graph last changed
	int optStripPatience = 1; // steps of an unchanged graph before the (more expensive) strip structure is built
	// hand-offs inside one XCD's L2 (persist_handoff.h: putGranuleNear: workgroup-scope stores, found by the per-launch census): an
	// assumption about the cache hierarchy that the memory model does not promise, so it is an option ("near_handoff") and the first
	// thing to go when a hand-off times out -- the step is tried again with agent-scope stores before the one-launch kernels are given up
	int optNearHandoff = 1, nearHandoffNow = 1, nearHandoffTimeouts = 0;
	int optStripAdopt = 1;	  // a body without constraints moves to the strip of the body it first touches instead of forcing a rebuild
	int optJacobiPersist = 1;		  // "jacobi_persist": s2Solve_Jacobi of a world that qualifies (solver_jacobi.cpp) in ONE launch (jacobi_kernel.hip)
	int optJacobiMinConstraints = 1024; // "jacobi_min_constraints": ... from this many constraints on
	int optOverflow = 1;	  // "strip_overflow": a contact that fits nowhere in the strips takes an overflow position (sliced steps + a worker-thread build) instead of a rebuild in this step
	long slicedSteps = 0;	  // steps that ran sliced, since s2amd_create
	bool slicedThisStep = false;
	int stripPatienceNow = 1; // ... as it stands: doubled every time a strip structure died young (noteGraphChanged)
	// ... and the same patience for the LDS groups and resident islands: their tables take no created contact (a contact of one of their
	// bodies, a pool slot of theirs that is used again, a watched manifold that gains its points are each a rebuild on the stepping
	// thread), so a world that keeps doing that to them -- the debris of a pyramid the reference's default solver lets collapse: a 2.4 ms
	// build in EVERY step, measured r6 -- gets its next structures without them: small islands in the colour batches of the global part,
	// where every contact has a place.  Steps the graph must have been quiet for before groups are built again; 0: at once.
	int groupPatienceNow = 0;
	bool dirtyByGroups = false; // the rebuild that is due was forced by something an LDS group / resident island could not take
	// ... and the sequential tail's room for created contacts (S2_TAIL_SLACK positions, S2_TAIL_BODY_SLACK bodies) times 2^this: raised when
	// structures die young of a hub's manifold that gained its points and found the tail full
	int tailSlackShift = 0;
	bool dirtyByWatched = false;
	int optGroupPatience = 1;	// "group_patience" 0: groups whatever they cost (tests; round 5's behaviour)
	int optTailTinyColour = 32; // "tail_tiny_colour": ... of the global part (a colour there is a launch per sweep)
	int optGroupTinyColour = 3; // "group_tiny_colour": colours of an LDS group with at most this many constraints may form its sequential tail (32: round 5)
	int optGenericPlace = 1;	// "generic_place" 0: the op interpreter's strips take no created contact (round 5)
	int optFlipColours = 1;		// "flip_colours" 0: a hub's manifold that gains its points is placed only where a sequential tail has room (round 5)
	bool stripPatienceSet = false; // "strip_patience" was set by the caller (else a resident world builds its strips at once: stripPatienceBase)
	int optStripsAnySolver = 0; // tests: strips for every solver and with joints (through the generic group interpreter)

	StepPlan plan;
	uint64_t planGeneration = 0;

	// options
	int optGraph = 1;
	int optGraphMinLaunches = 6; // "graph_min_launches": steps of fewer launches are enqueued directly, never captured
	int optProfile = 0;
	int optGroups = 1;
	// (r6: 1,024, was 2,048 -- one workgroup sweeping an island of 1,275 bodies takes 0.34 ms per TGS_Soft step and 0.43 at 1,830, the
	// strips 0.136 whatever the size; below ~900 bodies the group is the faster one: profiles/r06_island_size_sweep.txt)
	int optMaxGroupBodies = 1024;
	bool maxGroupBodiesSet = false; // "max_group_bodies" was set by the caller: that limit and nothing else (StructureBuild::findIslands)
	int optPackGroupBodies = 1024;
	bool packGroupBodiesSet = false; // "pack_group_bodies" was set by the caller (else: spread over the CUs, StructureBuild::findIslands)
	int optStrips = 1;		   // cut islands that do not fit one LDS group into strips of BFS levels (2 launches per sweep)
	int optStripBodies = 8;  // target bodies per strip: small = strips of exactly two BFS levels, five interior colour rounds (r3: 133 us per step at base 200
	bool stripBodiesSet = false; // "strip_bodies" was set by the caller: every solver gets that width
	int optStripBodiesLds = 320; // ... of SoftStep / PGS_Soft, whose seam constraints live in LDS and are swept by both neighbours (strip_kernel.hip): few, wide strips
							 // against 154 us with the six rounds of three-level strips); strip_retry tries wider ones when there are more level pairs than CUs
	bool stripMinBodiesSet = false; // "strip_min_bodies" was set by the caller
	int optStripMinBodies = 768; // loose bodies below which the colour-batch path is kept (r6: was 4,096 -- base-60 to base-80 pyramids ran 114 launches, 0.37 ms, where strips take 0.136;
								 // below the smallest island that is too big for an LDS group)

	// profiling events for the contact solve sweeps
	std::vector<hipEvent_t> sweepEvents;
	size_t sweepEventsUsed = 0;

	s2amdStepStats stats{};
	int launchCounter = 0;
	DevBuf dGatherIndex;
	bool gatherIndexDirty = true;
	uint64_t opsGeneration = ~0ull;

	// structure builds off the caller's thread (solver_async.cpp)
	AsyncBuild* async = nullptr;
	int optAsyncBuild = 1;	   // "async_build": in the world chain, 1: the search over strip widths (tens of milliseconds) runs in a worker thread on a copy
							   // of the solver while the steps go on with the strips they have; 2: the strip structure itself too (the steps go on on the
							   // colour batches meanwhile); adopted a fixed number of steps after the request; 0: everything on the caller's thread
	int optAsyncBuildDelay = 12; // "async_build_delay": steps between the request and the adoption of a strip build (the search: 8 x); the caller waits if the worker is not done by then
	bool isClone = false;	   // a worker's copy: the wire and world buffers are the owner's
	bool poolWarmed = false;   // asyncPrewarm has stocked the workers' pool for this solver's world
	bool forcedBuild = false;  // (a worker's copy) the live structure runs sliced until this build is adopted: strips at once, and a partition the resident
							   // kernel can run AND take created contacts into (persistValid, stripInc.valid) is all it asks for -- the search over strip
							   // widths only when the first width gives neither
	Stage4Args stage4{};		 // (world.hip, around a step's doStep) stage 4 of the world step, for the epilogue launch to carry
	bool stage4Carried = false;	 // ... it did
	bool graphCarriesStage4 = false; // ... and so does the captured step graph's epilogue launch
	int optPairsInStep = 1;	   // "pairs_in_step": once the caller has asked for pairs (s2amd_world_find_pairs), every s2amd_world_step enqueues the next query behind its stage 4 and the call returns its results without a device round trip of its own
	bool pairQueryUsed = false;	 // ... it has
	bool pairCacheValid = false; // ... the last step's query is waiting to be collected
	int optOverflowKernel = 1; // "overflow_kernel": overflow contacts are swept inside the persistent launch by one more workgroup (wide_kernel.hip: wideOverflowWorker); 0: sliced steps
	bool overflowKernelFailed = false;	 // ... that launch lost a hand-off once: this solver's overflow steps run sliced from then on
	bool overflowKernelThisStep = false; // ... and it is what the step just enqueued used
	int overflowRefusals = 0;  // builds for overflow contacts that could not be adopted, in a row: the second one is followed by a build in the step
	long stepCounter = 0;	   // steps enqueued since s2amd_create (the clock of the deferred adoption)
	// the search over strip widths (seven more builds, a copy of the solver for the worker): after a request the next one waits
	// `stripSearchPause` steps, twice as long every time (a pile with a ball in it needs its seven rounds at any width, and a search
	// the graph overtakes has found nothing either); a new world or a search whose result was better than what ran starts over
	const void* cancelBuild = nullptr; // (a worker's copy) std::atomic<int>: set when the graph has overtaken the build -- the search over strip widths gives up
	long stripSearchNotBefore = 0;
	int stripSearchPause = 256;
	int asyncRequested = 0, asyncAdopted = 0;
	float asyncWaitMs = 0.0f;
};

struct s2amdSolver : SolverStructure, SolverRest
{
};

// The constraint graph changed (an upload, a manifold that gained or lost its points, a contact slot written): the
// structure is rebuilt at the next step.  The strip structure costs several milliseconds of host time (more with
// strip_retry), so a world whose graph keeps changing every few steps must not build it again and again: the patience
// doubles whenever strips were in use for fewer than 32 steps, and returns to the option's value after a quiet spell.
// (strip_patience 0 means "always build at once" and is left alone; a world of another size is a new world.)
void asyncDrop(s2amdSolver* s);
// Steps of an unchanged graph before the strip structure is built ("strip_patience", default 1: the step that finds the graph changed
// runs on colour batches -- a cheaper build --, the strips come when the change has stayed alone); the wait doubles when strips die young (below).
inline int stripPatienceBase(const s2amdSolver* s)
{
	// (r4: "a resident world builds its strips in the step that needs them" -- patience 0 with the back-off below -- was measured on the
	// wrecking-ball world and LOST: graph changes come in bursts, the strips built at the first one die with the second, the wait doubles:
	// 151 instead of 201 of 240 steps on the persistent kernel.  And both routes of the drop-in must wait alike to sweep alike.)
	return s->optStripPatience;
}

inline void noteGraphChanged(s2amdSolver* s, bool newWorld = false)
{
	asyncDrop(s); // (a build in flight was made for the graph as it was)
	const bool stripsInUse = s->dStripA.view.groupCount > 0;
	const int base = stripPatienceBase(s);
	if (newWorld)
	{
		s->stripScaleFound = 0.0f;
		s->stripSearchNotBefore = 0, s->stripSearchPause = 256;
		s->groupPatienceNow = 0;
	}
	else if (s->optGroupPatience != 0 && s->dirtyByGroups && s->graphAge < 8 && s->contacts.order.size() + s->joints.order.size() >= S2_GROUP_PATIENCE_MIN_POSITIONS)
	{
		// (groups that died young -- in a world whose build costs more than the launches the colour batches cost per step: a build is
		// ~40 ns per constraint, the multi-launch path 0.5-1 ms per step whatever the size; for a world of a few hundred constraints
		// building again is the cheap way -- mixed-60: 0.28 ms per step with its groups, 0.72 without them)
		s->groupPatienceNow = std::min(std::max(2 * s->groupPatienceNow, 4), 256);
	}
	else if (s->graphAge >= 256)
	{
		s->groupPatienceNow = 0;
	}
	s->dirtyByGroups = false;
	if (!newWorld && s->dirtyByWatched && s->graphAge < 8)
	{
		s->tailSlackShift = std::min(s->tailSlackShift + 1, 4);
	}
	else if (newWorld || s->graphAge >= 256)
	{
		s->tailSlackShift = 0;
	}
	s->dirtyByWatched = false;
	if (newWorld || s->optStripPatience == 0)
	{
		s->stripPatienceNow = base; // (strip_patience 0 as an OPTION means "always at once", no backing off: tests)
	}
	else if (stripsInUse && s->stripInc.valid && !s->stripInc.takeOnly)
	{
		// strips that take created contacts in place (IncrementalStrips) die only of a contact that fits nowhere: they are worth
		// building again AT ONCE -- in the step that found the contact, no colour-batch structure in between -- unless this one lived for
		// less than it cost (a strip build ~ 5 ms buys ~0.4 ms per step).  (r4, once joining bodies, seam bodies and spare rounds were
		// placed: the strips built in the middle of a burst of contact changes now live through the rest of it -- wreck-200: 226 -> 240 of
		// 240 steps on the persistent kernel, 9 -> 5 steps that build anything.)
		s->stripPatienceNow = s->graphAge < 8 ? std::min(std::max(2 * s->stripPatienceNow, 2), 32) : 0;
		(void)base;
	}
	else if (stripsInUse && s->graphAge < 32)
	{
		s->stripPatienceNow = std::min(std::max(2 * s->stripPatienceNow, 2), 512);
	}
	else if (s->graphAge >= 256)
	{
		s->stripPatienceNow = base;
	}
	s->graphAge = 0;
	s->stripsRejected = false;
	s->stripsHopeless = false;
	s->residentRejected = false;
	s->structureDirty = true;
}

// A created contact was placed into the existing structure (solver_incremental.cpp): nothing is rebuilt, but the graph has
// not settled either -- the strip structure (milliseconds of host time) keeps waiting as after any change.
inline void noteGraphTouched(s2amdSolver* s)
{
	s->graphAge = 0;
}

StepConsts makeConsts(const s2amdStepParams* p);
int carveBodies(s2amdSolver* s, int n); // (re)carves the body SoA family for n slots
bool stripsAllTwoPoints(const s2amdSolver* s);
bool residentAllTwoPoints(const s2amdSolver* s);
int buildStructure(s2amdSolver* s, int solverType);
int buildJacobiBlocks(s2amdSolver* s); // solver_jacobi.cpp
void buildPlan(s2amdSolver* s, const s2amdStepParams* params);
bool messageEligible(const s2amdSolver* s, int solverType);
void destroyGraph(s2amdSolver* s);
// pairs != nullptr (world chain): a slot is a potential constraint iff its pair is live; else iff it names two distinct live bodies
int doUpload(s2amdSolver* s, const s2amdBody* bodies, int nb, const s2amdContact* contacts, int nc, const s2amdJoint* joints, int nj,
			 const s2amdPairState* pairs = nullptr);
// manifold.constraintIndex of every resident contact slot from the resident point counts (world.hip; device scan)
int refreshConstraintIndexOnDevice(s2amdSolver* s);
// point counts of the resident manifolds -> hPointBytes / hContactPoints (world chain: one byte per slot)
int fetchPointCounts(s2amdSolver* s);
// world chain: which pair slots the device has freed (stage 3 separations) -> hContactDead, before a structure rebuild
int syncDeadSlots(s2amdSolver* s);

// solver_async.cpp: structure builds in a worker thread on a copy of the solver, adopted a fixed number of steps later
bool asyncBuildsOn(const s2amdSolver* s);
bool asyncPending(const s2amdSolver* s);
bool asyncPendingSearch(const s2amdSolver* s);
int asyncRequest(s2amdSolver* s, int solverType, bool search, bool forceStrips = false);
bool asyncAdopt(s2amdSolver* s, int solverType, int* rc);
int asyncPrewarm(s2amdSolver* s, int solverType);
void asyncDrop(s2amdSolver* s);
void asyncLogCreated(s2amdSolver* s, int slot, int a, int b);
void asyncLogDestroyed(s2amdSolver* s, int slot);
void asyncShutdown(s2amdSolver* s);

// solver_incremental.cpp
struct ContactChange
{
	int slot, a, b; // a < 0: the slot's old entry is only removed
};
// Gives every change a place in the existing structure (removing the slot's previous entry first); false: one of them
// does not fit -- the caller marks the graph changed (full rebuild; nothing has reached the device).
bool canDeferCreated(const s2amdSolver* s, int slot, int a, int b);
void deferCreated(s2amdSolver* s, int slot, int a, int b);
void unwatchSlot(s2amdSolver* s, int slot);
int uploadWatched(s2amdSolver* s);
bool stripCanPlace(const s2amdSolver* s, int a, int b);
bool ownedByLdsGroup(const s2amdSolver* s, int body); // an LDS group or a resident island (not a strip) holds the body: nothing can be placed on it
// ... or, failing that, a free position of the overflow region behind the strips (IncrementalStrips::overflowFree): the steps run sliced
// until a worker thread's structure that holds the contact is adopted
bool overflowCanPlace(const s2amdSolver* s, int a, int b);
// ... or, between two bodies of the global part, a colour position or a free position of the sequential tail (IncrementalGlobal::tailFree)?
bool tailCanPlace(const s2amdSolver* s, int a, int b);
bool incrementalApply(s2amdSolver* s, const std::vector<ContactChange>& changes);
// Destroyed contacts give their place back (colour, position, list entries) where the entry is in the global part's
// parallel batches; elsewhere (LDS group, strip, sequential tail) the entry lingers as a no-op until the next rebuild.
void incrementalRemove(s2amdSolver* s, const int32_t* slots, int count);
// enqueues the patch list built by incrementalApply on the solver's stream
int incrementalFlush(s2amdSolver* s);
int doStep(s2amdSolver* s, const s2amdStepParams* params);
// after a persistent step lost a hand-off: both error words, the commit counter and the hand-off buffers back to zero (enqueued on `st`)
int resetPersistState(s2amdSolver* s, hipStream_t st);
int doDownload(s2amdSolver* s, s2amdBody* bodies, int nb, s2amdContact* contacts, int nc, s2amdJoint* joints, int nj);
