// Structure builds off the caller's thread ("per-step host work O(1)").
//
// The reference changes its constraint graph every step (src/world.c:138-168, src/contact.c:156-229); most changes are placed into the
// existing structure (solver_incremental.cpp), but two things still cost milliseconds of host time when they come: building the strip
// structure of a big island (BFS levels, per-strip colouring, the persistent kernel's tables: ~4 ms at 60k constraints) and the search
// over other strip widths (seven more builds).  Both used to run inside the step that found them due.  Now, in the world chain, they run
// in a worker thread on a COPY of the solver -- same host shadows, same options, its own stream and device tables -- while the steps go
// on with the structure they have, which is valid (colour batches, or the strips the search wants to improve on).  The result is adopted
// a FIXED number of steps after the request (the caller waits if the worker is not done: the step at which the sweep order changes must
// not depend on thread timing), by swapping the SolverStructure part of the two solver objects.
//
// What happens to the graph between request and adoption is logged and replayed on the copy first: a destroyed contact leaves it as it
// leaves the live structure (incrementalRemove), a created one is placed or deferred exactly as s2amd_world_set_contacts does it on the
// live structure (incrementalApply / deferCreated).  Anything the copy cannot take -- a contact that fits nowhere, a watched manifold
// that gained its points, an upload, an option that changes the structure -- drops the build; the next step that finds one due asks again.
#include "solver_internal.h"

#include <atomic>
#include <mutex>
#include <thread>

// ---- the worker threads' device-memory pool (solver_internal.h: DevBuf) ----
namespace
{
struct DevPool
{
	std::mutex m;
	std::vector<std::pair<void*, size_t>> blocks;
	size_t bytes = 0;
	std::vector<hipStream_t> streams;
	std::vector<std::pair<void*, size_t>> pinned;
};
// One pool per device (the calling thread's current one -- every taker and giver has set it: workerMain, destroyClone, s2amd_destroy):
// a block, a stream or a pinned buffer made on device 0 must never reach a copy that builds for device 1.
constexpr int kMaxPoolDevices = 64;
DevPool& devPool()
{
	static DevPool pools[kMaxPoolDevices];
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxPoolDevices)
	{
		(void)hipGetLastError();
		device = 0;
	}
	return pools[device];
}
// solvers alive per device: the last one to go empties its device's pool (s2amd_create / s2amd_destroy)
std::mutex g_aliveMutex;
int g_alive[kMaxPoolDevices] = {};
thread_local bool tlsDevPool = false;
constexpr size_t kDevPoolLimit = size_t(1) << 30; // what a device's pool keeps at most; a block beyond that is freed as ever
} // namespace

void devPoolSolverCreated(int device)
{
	std::lock_guard<std::mutex> lock(g_aliveMutex);
	if (device >= 0 && device < kMaxPoolDevices)
	{
		g_alive[device] += 1;
	}
}
// true: that was the device's last solver (the caller, with the device current, drains its pool)
bool devPoolSolverDestroyed(int device)
{
	std::lock_guard<std::mutex> lock(g_aliveMutex);
	if (device < 0 || device >= kMaxPoolDevices)
	{
		return true;
	}
	g_alive[device] = std::max(g_alive[device] - 1, 0);
	return g_alive[device] == 0;
}
bool devPoolNoSolverLeft()
{
	std::lock_guard<std::mutex> lock(g_aliveMutex);
	for (int n : g_alive)
	{
		if (n != 0)
		{
			return false;
		}
	}
	return true;
}

void devPoolThread(bool on) { tlsDevPool = on; }
bool devPoolOn() { return tlsDevPool; }
void* devPoolTake(size_t need, size_t* got)
{
	DevPool& pool = devPool();
	std::lock_guard<std::mutex> lock(pool.m);
	int best = -1;
	for (int i = 0; i < (int)pool.blocks.size(); ++i)
	{
		const size_t b = pool.blocks[(size_t)i].second;
		if (b >= need && b <= 4 * need + (size_t(1) << 16) && (best < 0 || b < pool.blocks[(size_t)best].second))
		{
			best = i;
		}
	}
	if (best < 0)
	{
		return nullptr;
	}
	void* p = pool.blocks[(size_t)best].first;
	*got = pool.blocks[(size_t)best].second;
	pool.bytes -= *got;
	pool.blocks.erase(pool.blocks.begin() + best);
	return p;
}
bool devPoolGive(void* p, size_t bytes)
{
	DevPool& pool = devPool();
	std::lock_guard<std::mutex> lock(pool.m);
	if (pool.bytes + bytes > kDevPoolLimit)
	{
		return false;
	}
	pool.blocks.emplace_back(p, bytes);
	pool.bytes += bytes;
	return true;
}
void devPoolDrain()
{
	DevPool& pool = devPool();
	std::vector<std::pair<void*, size_t>> blocks, pinned;
	std::vector<hipStream_t> streams;
	{
		std::lock_guard<std::mutex> lock(pool.m);
		blocks.swap(pool.blocks);
		pinned.swap(pool.pinned);
		streams.swap(pool.streams);
		pool.bytes = 0;
	}
	for (const auto& b : blocks)
	{
		(void)hipFree(b.first);
	}
	for (const auto& b : pinned)
	{
		(void)hipHostFree(b.first);
	}
	for (hipStream_t st : streams)
	{
		(void)hipStreamDestroy(st);
	}
}
hipStream_t workerStreamTake()
{
	DevPool& pool = devPool();
	{
		std::lock_guard<std::mutex> lock(pool.m);
		if (!pool.streams.empty())
		{
			hipStream_t st = pool.streams.back();
			pool.streams.pop_back();
			return st;
		}
	}
	hipStream_t st = nullptr;
	if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
	{
		(void)hipGetLastError();
		return nullptr;
	}
	return st;
}
void workerStreamGive(hipStream_t st)
{
	if (st)
	{
		DevPool& pool = devPool();
		std::lock_guard<std::mutex> lock(pool.m);
		pool.streams.push_back(st);
	}
}
void* pinnedPoolTake(size_t need, size_t* got)
{
	DevPool& pool = devPool();
	std::lock_guard<std::mutex> lock(pool.m);
	for (size_t i = 0; i < pool.pinned.size(); ++i)
	{
		if (pool.pinned[i].second >= need)
		{
			void* p = pool.pinned[i].first;
			*got = pool.pinned[i].second;
			pool.pinned.erase(pool.pinned.begin() + (long)i);
			return p;
		}
	}
	return nullptr;
}
void pinnedPoolGive(void* p, size_t bytes)
{
	DevPool& pool = devPool();
	std::lock_guard<std::mutex> lock(pool.m);
	pool.pinned.emplace_back(p, bytes);
}

struct AsyncBuild
{
	std::thread worker;
	s2amdSolver* clone = nullptr;
	std::atomic<int> done{0};
	int rc = S2AMD_OK;
	int solverType = 0;
	bool search = false;   // the search over strip widths (buildStructure with the graph at rest), not only the strip structure
	bool forced = false;   // the live structure runs sliced until this build is adopted (overflow contacts)
	bool dropped = false;  // overtaken by the graph: the result is thrown away when the worker is done
	std::atomic<int> cancel{0}; // ... and the worker is told: a search stops after the build it is in (SolverRest::cancelBuild)
	long requestedAtStep = 0;
	int delay = 0; // steps between the request and the adoption
	struct Event
	{
		int kind; // 0 created (slot, a, b), 1 destroyed (slot)
		int slot, a, b;
	};
	std::vector<Event> log;
	AsyncBuild* older = nullptr; // dropped builds whose worker may still be running
};

namespace
{

// every device allocation, pinned buffer and captured graph of the structure part: a copy must not share them with its original
void forgetDeviceState(SolverStructure& t)
{
	DevBuf* bufs[] = {&t.dWatched, &t.dBodyFlags, &t.soaBodies, &t.soaContacts, &t.soaJoints, &t.dContactIndex, &t.dJointIndex, &t.dContactLocal, &t.dJointLocal,
					  &t.dAdjOffsets, &t.dAdjList, &t.dAdjHeavy, &t.dPatches, &t.dJointAdjRange, &t.dJointAdjList, &t.dResidentDesc, &t.dResidentOps, &t.dStripLean,
					  &t.dPersist, &t.dGranules, &t.dOverflowBodies, &t.dPersistOps, &t.dJacobi, &t.dJacobiGran, &t.dMsg, &t.dGroups.buf, &t.dContactTail.buf, &t.dJointTail.buf, &t.dStripA.buf, &t.dStripB.buf,
					  &t.dResident.buf};
	for (DevBuf* b : bufs)
	{
		b->p = nullptr, b->bytes = 0;
	}
	t.dGroups = DeviceGroupTable{}, t.dContactTail = DeviceGroupTable{}, t.dJointTail = DeviceGroupTable{}, t.dStripA = DeviceGroupTable{}, t.dStripB = DeviceGroupTable{};
	t.dResident = DeviceGroupTable{};
	t.hostPatches = nullptr, t.hostPatchCapacity = 0;
	t.graph = nullptr, t.graphExec = nullptr, t.graphKey = 0, t.graphKeySeen = 0, t.graphKeySeenLaunches = 0, t.graphLaunches = 0;
	t.bv = BodyView{}, t.cv = ContactView{}, t.jv = JointView{};
	t.bodySoaCap = t.contactSoaCap = t.jointSoaCap = 0;
	t.leanA = StripTableView{}, t.leanB = StripTableView{}, t.residentView = StripTableView{};
	t.leanAValid = t.leanBValid = t.persistValid = t.genericValid = false;
	t.persist = PersistView{};
	t.jacobi = JacobiView{};
	t.jacobiValid = false;
	t.msg = MsgView{};
	t.msgTablesValid = false;
	t.adjValid = false, t.jointAdjValid = false;
	t.granuleBytes = 0;
	t.residentOpsGeneration = ~0ull, t.persistOpsGeneration = ~0ull, t.persistOpsStructure = ~0ull;
	t.inc = IncrementalGlobal{};
	t.stripInc = IncrementalStrips{};
	t.structureDirty = true;
}

void releaseDeviceState(s2amdSolver* c)
{
	(void)hipSetDevice(c->device);
	if (c->stream)
	{
		(void)hipStreamSynchronize(c->stream);
	}
	destroyGraph(c);
	SolverStructure& t = *c;
	DevBuf* bufs[] = {&t.dWatched, &t.dBodyFlags, &t.soaBodies, &t.soaContacts, &t.soaJoints, &t.dContactIndex, &t.dJointIndex, &t.dContactLocal, &t.dJointLocal,
					  &t.dAdjOffsets, &t.dAdjList, &t.dAdjHeavy, &t.dPatches, &t.dJointAdjRange, &t.dJointAdjList, &t.dResidentDesc, &t.dResidentOps, &t.dStripLean,
					  &t.dPersist, &t.dGranules, &t.dOverflowBodies, &t.dPersistOps, &t.dJacobi, &t.dJacobiGran, &t.dMsg, &t.dGroups.buf, &t.dContactTail.buf, &t.dJointTail.buf, &t.dStripA.buf, &t.dStripB.buf,
					  &t.dResident.buf};
	for (DevBuf* b : bufs)
	{
		b->release();
	}
	if (t.hostPatches)
	{
		if (devPoolOn())
		{
			pinnedPoolGive(t.hostPatches, t.hostPatchCapacity * sizeof(uint4));
		}
		else
		{
			(void)hipHostFree(t.hostPatches);
		}
		t.hostPatches = nullptr;
		t.hostPatchCapacity = 0;
	}
}

// Retired copies, kept as OBJECTS for the next request: a copy is megabytes of host vectors (what the host knows of the wire arrays, the
// graph's endpoints), and filling fresh ones costs the requesting step 0.35 ms of page faults at 140k contact slots against 0.07 ms
// into vectors that have their capacity (measured, r5: the first overflow build's request was one of the two steps over 1 ms).
std::mutex gSpareMutex;
std::vector<s2amdSolver*> gSpareClones;

s2amdSolver* cloneTake()
{
	{
		std::lock_guard<std::mutex> lock(gSpareMutex);
		if (!gSpareClones.empty())
		{
			s2amdSolver* c = gSpareClones.back();
			gSpareClones.pop_back();
			return c;
		}
	}
	return new s2amdSolver();
}

// a worker's copy of the solver: frees what IT owns (the structure part's device state, its stream); everything else is the owner's
void destroyClone(s2amdSolver* c)
{
	if (!c)
	{
		return;
	}
	releaseDeviceState(c);
	workerStreamGive(c->stream); // (synchronised above; kept for the next copy: creating and destroying streams stalls every thread's HIP calls)
	c->stream = nullptr;
	c->async = nullptr;
	{
		// the structure part as a new object's would be -- but for the four vectors every request fills -- (here, on the thread that
		// takes the copy apart: giving back the tables' megabytes costs what filling them did)
		std::vector<int> a = std::move(c->hContactA), b = std::move(c->hContactB);
		auto edge = std::move(c->hContactEdge);
		auto dead = std::move(c->hContactDead);
		static_cast<SolverStructure&>(*c) = SolverStructure{};
		c->hContactA = std::move(a), c->hContactB = std::move(b), c->hContactEdge = std::move(edge), c->hContactDead = std::move(dead);
	}
	{
		std::lock_guard<std::mutex> lock(gSpareMutex);
		if (gSpareClones.size() < 2)
		{
			gSpareClones.push_back(c);
			return;
		}
	}
	delete c;
}

void workerMain(AsyncBuild* job)
{
	devPoolThread(true);
	s2amdSolver* c = job->clone;
	int rc = S2AMD_OK;
	if (hipSetDevice(c->device) != hipSuccess)
	{
		rc = S2AMD_E_DEVICE;
	}
	bool grew = false;
	if (rc == S2AMD_OK)
	{
		rc = c->dBodyFlags.ensure((size_t)std::max(c->bodyCapacity, 1) * sizeof(uint32_t), &grew);
	}
	if (rc == S2AMD_OK)
	{
		rc = carveBodies(c, c->bodyCapacity);
	}
	if (rc == S2AMD_OK)
	{
		rc = buildStructure(c, job->solverType);
	}
	if (rc == S2AMD_OK && hipStreamSynchronize(c->stream) != hipSuccess)
	{
		rc = S2AMD_E_DEVICE;
	}
	job->rc = rc;
	job->done.store(1, std::memory_order_release);
}

// dropped builds whose worker has finished: their copies go
void reap(AsyncBuild*& list, bool wait)
{
	AsyncBuild** at = &list;
	while (*at)
	{
		AsyncBuild* j = *at;
		if (!wait && j->dropped && j->done.load(std::memory_order_acquire) && j->clone != nullptr)
		{
			// a dropped build whose worker is done: its copy is taken apart by a thread of its own (stream, pinned buffer, graph,
			// megabytes of host vectors: 3 ms on the stepping thread, measured) and the job is reaped once that is over
			if (j->worker.joinable())
			{
				j->worker.join();
			}
			j->done.store(0, std::memory_order_release);
			j->worker = std::thread([j]() {
				devPoolThread(true);
				destroyClone(j->clone);
				j->clone = nullptr;
				j->done.store(1, std::memory_order_release);
			});
			at = &j->older;
			continue;
		}
		if (wait || (j->dropped && j->done.load(std::memory_order_acquire)))
		{
			if (j->worker.joinable())
			{
				j->worker.join();
			}
			{
				// (the caller's thread: the copy's device memory goes to the workers' pool, not through hipFree -- DevBuf)
				const bool was = devPoolOn();
				devPoolThread(true);
				destroyClone(j->clone);
				devPoolThread(was);
			}
			*at = j->older;
			delete j;
		}
		else
		{
			at = &j->older;
		}
	}
}

} // namespace

// the process's last solver is gone (s2amd_destroy): the retired copies go too, like the device pools
void spareClonesRelease()
{
	std::vector<s2amdSolver*> spare;
	{
		std::lock_guard<std::mutex> lock(gSpareMutex);
		spare.swap(gSpareClones);
	}
	for (s2amdSolver* c : spare)
	{
		delete c;
	}
}


bool asyncBuildsOn(const s2amdSolver* s)
{
	return s->optAsyncBuild != 0 && s->worldResident && !s->isClone;
}

bool asyncPending(const s2amdSolver* s)
{
	return s->async != nullptr && !s->async->dropped;
}

// ... and is it the search over strip widths (adopted only when it scores better, eight times the delay away)?
bool asyncPendingSearch(const s2amdSolver* s)
{
	return asyncPending(s) && s->async->search;
}

// the graph moved in a way the pending build cannot follow: its result will be thrown away
void asyncDrop(s2amdSolver* s)
{
	if (s->async && !s->async->dropped)
	{
		s->async->dropped = true;
		s->async->cancel.store(1, std::memory_order_relaxed);
		s->async->log.clear();
	}
}

void asyncLogCreated(s2amdSolver* s, int slot, int a, int b)
{
	if (asyncPending(s))
	{
		s->async->log.push_back(AsyncBuild::Event{0, slot, a, b});
	}
}

void asyncLogDestroyed(s2amdSolver* s, int slot)
{
	if (asyncPending(s))
	{
		s->async->log.push_back(AsyncBuild::Event{1, slot, -1, -1});
	}
}

// the owner is going away (or its world is replaced): nothing of a worker may outlive it
void asyncShutdown(s2amdSolver* s)
{
	asyncDrop(s); // (sets the worker's cancel flag: a search over strip widths stops after the build it is in instead of running its seven)
	reap(s->async, true);
}

// `search`: the graph has been at rest long enough for the search over strip widths (the copy sees the same age and runs it)
// `forceStrips`: the copy builds its strips whatever the graph's age (the live structure runs sliced until it is adopted)
// the copy a worker builds on: everything but the structure part as it is (what the host knows of the wire arrays, the options, the
// plan); of the structure part only what a build reads -- the graph as the structure knows it and the policies earlier builds have learnt
static void fillClone(s2amdSolver* c, const s2amdSolver* s)
{
	static_cast<SolverRest&>(*c) = static_cast<const SolverRest&>(*s);
	c->hContactA = s->hContactA, c->hContactB = s->hContactB, c->hContactEdge = s->hContactEdge, c->hContactDead = s->hContactDead;
	c->spareColours = s->spareColours, c->slackShift = s->slackShift, c->slackBumped = s->slackBumped, c->slackAtBuild = s->slackAtBuild;
	c->slackPositions = s->slackPositions;
	c->stripScaleFound = s->stripScaleFound, c->stripScaleFoundFor = s->stripScaleFoundFor, c->stripsJudgedForClass = s->stripsJudgedForClass;
	c->stripsRejected = false, c->stripsHopeless = false, c->residentRejected = s->residentRejected;
	c->layoutGeneration = s->layoutGeneration, c->structureGeneration = s->structureGeneration;
	c->isClone = true;
	c->async = nullptr;
	c->worldResident = false; // (no device reads of the world's arrays from the worker: the shadows above are current)
	c->pointsKnown = true;
	c->stream = nullptr;
	c->evBegin = c->evEnd = nullptr;
	for (hipEvent_t& e : c->evExport)
	{
		e = nullptr;
	}
	c->side[0] = c->side[1] = nullptr;
	c->hostTimes = nullptr;
	c->sweepEvents.clear();
}

int asyncRequest(s2amdSolver* s, int solverType, bool search, bool forceStrips)
{
	static const bool debugAsync = getenv("S2AMD_DEBUG_ASYNC") != nullptr;
	const double tr0 = debugAsync ? nowMs() : 0.0;
	double tr1 = 0.0, tr2 = 0.0, tr3 = 0.0;
	reap(s->async, false);
	if (asyncPending(s))
	{
		return S2AMD_OK;
	}
	HIP_TRY(hipSetDevice(s->device));
	// what a synchronous build would learn from the device first (buildStructureWith, gatherEdges): the pairs stage 3 has freed, and --
	// for the hub rule -- which manifolds have points right now.  The copy then builds from host state alone.
	// (a world without hub bodies -- solver_internal.h: hBodyHub -- asks nothing of the point counts, and the pairs its stage 3 frees reach
	// hContactDead with every step's read-back: the two device reads, 7 ms on the step that asks for a search, are the hub rule's)
	bool anyHub = false;
	for (uint8_t h : s->hBodyHub)
	{
		anyHub = anyHub || h != 0;
	}
	if (!s->pointsKnown && !(forceStrips && s->pointCountsFresh) && anyHub)
	{
		// (forceStrips: the flip that asked for this build has just read the point counts, and a pair stage 3 freed this step lingers in
		// the copy's structure as it does in the live one -- a no-op -- until it is read at the end of the step)
		int rc = syncDeadSlots(s);
		if (rc == S2AMD_OK)
		{
			rc = fetchPointCounts(s);
		}
		if (rc)
		{
			return rc;
		}
	}
	tr1 = debugAsync ? nowMs() : 0.0;
	AsyncBuild* job = new AsyncBuild();
	// the copy: everything but the structure part as it is (what the host knows of the wire arrays, the options, the plan); of the
	// structure part only what a build reads -- the graph as the structure knows it and the policies earlier builds have learnt.
	// (A copy of the whole object, tables and placement mirrors included, cost the requesting step 5 ms at 140k contact slots.)
	s2amdSolver* c = cloneTake();
	fillClone(c, s);
	forgetDeviceState(*c);
	tr2 = debugAsync ? nowMs() : 0.0;
	c->stream = workerStreamTake();
	if (c->stream == nullptr)
	{
		delete c;
		delete job;
		return fail(S2AMD_E_DEVICE, "could not create the build worker's stream");
	}
	if (search)
	{
		c->stripRetryPending = false;
	}
	if (forceStrips)
	{
		// (the live structure runs sliced until this one is adopted: its strips at once, and ONE build where that gives a partition the
		// resident kernel runs -- solver_structure.cpp: buildStructure; the search for the best strip width, tens of milliseconds, is asked
		// for again by the adopted structure when the graph has been quiet for a while)
		c->stripPatienceNow = 0;
		c->forcedBuild = true;
	}
	job->clone = c;
	c->cancelBuild = &job->cancel;
	job->solverType = solverType;
	job->search = search;
	job->forced = forceStrips;
	job->requestedAtStep = s->stepCounter;
	// (a build the live structure is waiting for -- it runs sliced meanwhile -- falls due sooner: one strip build is ~5 ms of the worker's time,
	// a sliced step ~0.8 ms of the caller's)
	job->delay = search ? 8 * s->optAsyncBuildDelay : (forceStrips ? std::max(2, s->optAsyncBuildDelay / 2) : s->optAsyncBuildDelay);
	if (search)
	{
		// (the request itself costs the caller a device synchronisation -- the point counts, the freed pairs -- and a copy of the
		// host's picture of the graph: 3-4 ms at 140k contact slots.  Until one pays off, each search waits twice as long as the last.)
		s->stripSearchNotBefore = s->stepCounter + s->stripSearchPause;
		s->stripSearchPause = std::min(2 * s->stripSearchPause, 1 << 14);
	}
	job->older = s->async;
	s->async = job;
	s->asyncRequested += 1;
	tr3 = debugAsync ? nowMs() : 0.0;
	job->worker = std::thread(workerMain, job);
	if (debugAsync)
	{
		fprintf(stderr, "[s2amd] step %ld: build requested (%s): reap + device reads %.3f ms, copy %.3f, stream + job %.3f, thread start %.3f\n", s->stepCounter,
				search ? "search" : (forceStrips ? "forced" : "build"), tr1 - tr0, tr2 - tr1, tr3 - tr2, nowMs() - tr3);
	}
	return S2AMD_OK;
}

// The first worker-thread build finds the pool empty and makes its twenty-odd device blocks with hipMalloc -- which serialises with every
// HIP call of the process: the steps beside that build stalled for 4-8 ms (measured, r5: the step that asks for the first search, the
// step that asks for the first overflow build).  So once a world has its structure, blocks of the sizes a copy will ask for -- the live
// structure's own -- and one pinned patch buffer go into the pool while nothing depends on the step's latency (the upload, or the first
// step's tail).  Twice the structure's device memory: tens of megabytes at 60k constraints.
int asyncPrewarm(s2amdSolver* s, int solverType)
{
	if (s->poolWarmed || s->optAsyncBuild == 0 || s->isClone || s->structureDirty)
	{
		return S2AMD_OK;
	}
	s->poolWarmed = true;
	HIP_TRY(hipSetDevice(s->device));
	SolverStructure& t = *s;
	const DevBuf* bufs[] = {&t.dWatched, &t.dBodyFlags, &t.soaBodies, &t.soaContacts, &t.soaJoints, &t.dContactIndex, &t.dJointIndex, &t.dContactLocal, &t.dJointLocal,
							&t.dAdjOffsets, &t.dAdjList, &t.dAdjHeavy, &t.dPatches, &t.dJointAdjRange, &t.dJointAdjList, &t.dResidentDesc, &t.dResidentOps, &t.dStripLean,
							&t.dPersist, &t.dGranules, &t.dOverflowBodies, &t.dPersistOps, &t.dJacobi, &t.dJacobiGran, &t.dMsg, &t.dGroups.buf, &t.dContactTail.buf, &t.dJointTail.buf, &t.dStripA.buf, &t.dStripB.buf,
							&t.dResident.buf};
	for (const DevBuf* b : bufs)
	{
		const size_t bytes = std::max<size_t>(b->bytes, 4096);
		void* p = nullptr;
		HIP_TRY(hipMalloc(&p, bytes));
		if (!devPoolGive(p, bytes))
		{
			(void)hipFree(p);
			break;
		}
	}
	void* pinned = nullptr;
	const size_t pinnedBytes = 8192 * sizeof(uint4);
	if (hipHostMalloc(&pinned, pinnedBytes, hipHostMallocDefault) == hipSuccess)
	{
		pinnedPoolGive(pinned, pinnedBytes);
	}
	else
	{
		(void)hipGetLastError();
	}
	{
		// three streams: a search over strip widths in flight, the build that takes its place, the one after an adoption (creating one
		// later costs the step that asks 3.7 ms, measured)
		hipStream_t st[3] = {workerStreamTake(), workerStreamTake(), workerStreamTake()};
		for (hipStream_t x : st)
		{
			workerStreamGive(x);
		}
	}
	// ... and one build by a worker thread, thrown away: the first thread that talks to the HIP runtime, the first launches of the kernels a
	// build uses and the first copy of the solver cost the steps beside them 4-8 ms each (r5: the requests at steps 32 and 87 of the
	// wrecking-ball loop) -- paid here instead
	if (asyncPending(s))
	{
		return S2AMD_OK;
	}
	const int requestedWas = s->asyncRequested;
	int rc = asyncRequest(s, solverType, false, true);
	if (rc)
	{
		return rc;
	}
	asyncDrop(s);
	reap(s->async, true);
	s->asyncRequested = requestedWas;
	// (the copy that build worked on waits as an OBJECT for the next request -- cloneTake --; a second one beside it, because a search over
	// strip widths holds its copy for a hundred steps and the overflow build that falls into them found none: 0.31 ms of its request)
	{
		s2amdSolver* extra = new s2amdSolver();
		fillClone(extra, s);
		forgetDeviceState(*extra);
		const bool was = devPoolOn();
		devPoolThread(true);
		destroyClone(extra);
		devPoolThread(was);
	}
	return S2AMD_OK;
}

// the logged changes of the graph, applied to the copy as s2amd_world_set_contacts / s2amd_world_step applied them to the live structure
static bool replay(s2amdSolver* c, const std::vector<AsyncBuild::Event>& log)
{
	for (const AsyncBuild::Event& e : log)
	{
		if (e.slot < 0 || e.slot >= c->contactCapacity)
		{
			return false;
		}
		if (e.kind == 1)
		{
			if (c->hContactEdge[(size_t)e.slot])
			{
				c->hContactDead[(size_t)e.slot] = 1;
				unwatchSlot(c, e.slot);
				const int32_t one = e.slot;
				incrementalRemove(c, &one, 1);
			}
			continue;
		}
		if (canDeferCreated(c, e.slot, e.a, e.b))
		{
			deferCreated(c, e.slot, e.a, e.b);
			continue;
		}
		const std::vector<ContactChange> one{ContactChange{e.slot, e.a, e.b}};
		if (!incrementalApply(c, one))
		{
			return false;
		}
		c->hContactA[(size_t)e.slot] = e.a;
		c->hContactB[(size_t)e.slot] = e.b;
		c->hContactEdge[(size_t)e.slot] = 1;
		c->hContactDead[(size_t)e.slot] = 0;
	}
	return incrementalFlush(c) == S2AMD_OK && hipStreamSynchronize(c->stream) == hipSuccess;
}

// Called by doStep before it looks at the structure: adopts a build that is due.  true: the structure was replaced.
bool asyncAdopt(s2amdSolver* s, int solverType, int* rcOut)
{
	*rcOut = S2AMD_OK;
	reap(s->async, false); // (dropped builds whose worker is done)
	AsyncBuild* job = s->async;
	if (!job)
	{
		return false;
	}
	if (job->dropped)
	{
		return false; // (its worker is still running: reaped at a later step)
	}
	const long due = job->requestedAtStep + (long)job->delay;
	if (s->stepCounter < due)
	{
		return false;
	}
	if (job->forced && !job->done.load(std::memory_order_acquire))
	{
		// a build the live structure waits for in sliced steps: one that needed the search over strip widths after all is not waited for
		// on the stepping thread (tens of milliseconds) -- it falls due a few steps later
		job->delay += 4;
		return false;
	}
	// due: the step at which the sweep order changes is fixed, so a worker that is not done yet is waited for
	const double t0 = nowMs();
	const double ta0 = t0;
	if (job->worker.joinable())
	{
		job->worker.join();
	}
	s->asyncWaitMs += (float)(nowMs() - t0);
	s2amdSolver* c = job->clone;
	bool ok = job->rc == S2AMD_OK && job->solverType == solverType && !c->structureDirty;
	// (a search that found nothing better than what runs now is not worth the swap -- nor another search soon)
	auto score = [](const s2amdSolver* x) {
		if (x->dStripA.view.groupCount == 0)
		{
			return 0;
		}
		if (x->persistValid && x->persist.maxRoundsA <= S2_STRIP_ROUNDS)
		{
			return (x->persist.maxRoundsA <= 5 && x->persist.maxSeamRounds <= 2) ? 4 : 3;
		}
		return x->persistValid ? 2 : 1;
	};
	if (ok && job->search)
	{
		if (score(c) <= score(s))
		{
			ok = false;
			// (what the search learnt stays: the width it settled on -- its own first width when none was better -- keeps
			// buildStructure from asking for another search while the partition is one the resident kernels take)
			if (c->stripScaleFound > 0.0f && c->stripScaleFoundFor == s->stripScaleFoundFor)
			{
				s->stripScaleFound = c->stripScaleFound;
			}
		}
		else
		{
			s->stripSearchPause = 256, s->stripSearchNotBefore = 0; // (it paid off: the next one may come as soon as it is asked for)
		}
	}
	// (replay and placement on the copy grow its patch buffers: from the workers' pool, not through hipHostMalloc / hipMalloc, which
	// stall this -- the stepping -- thread for milliseconds: 10.7 ms measured on the step a search fell due in, r5)
	const bool poolWas = devPoolOn();
	devPoolThread(true);
	ok = ok && replay(c, job->log);
	if (ok && c->watchedCount > 0)
	{
		// a slot the copy only watches (no entry in its structure) that has gained its manifold points meanwhile -- the live structure has
		// placed it: a strip round, an overflow position -- takes its place in the copy's strips now, as it would have had the copy been
		// live (the order the world chain places flips in: ascending slots); one that fits nowhere there refuses the adoption
		if (fetchPointCounts(s) != S2AMD_OK)
		{
			ok = false;
		}
		std::vector<ContactChange> flipped;
		for (int i = 0; ok && i < s->contactCapacity; ++i)
		{
			if (c->hContactWatched[(size_t)i] && s->hContactPoints[(size_t)i] > 0 && c->inc.positionOfSlot[(size_t)i] == -1)
			{
				if (c->hContactEdge[(size_t)i] && !c->hContactDead[(size_t)i] && c->stripInc.valid &&
					(stripCanPlace(c, c->hContactA[(size_t)i], c->hContactB[(size_t)i]) || overflowCanPlace(c, c->hContactA[(size_t)i], c->hContactB[(size_t)i])))
				{
					flipped.push_back(ContactChange{i, c->hContactA[(size_t)i], c->hContactB[(size_t)i]});
				}
				else
				{
					ok = false;
				}
			}
		}
		if (ok && !flipped.empty())
		{
			ok = incrementalApply(c, flipped) && incrementalFlush(c) == S2AMD_OK && hipStreamSynchronize(c->stream) == hipSuccess;
		}
	}
	devPoolThread(poolWas);
	static const bool debugAsync = getenv("S2AMD_DEBUG_ASYNC") != nullptr;
	if (debugAsync)
	{
		fprintf(stderr, "[s2amd]   adoption work before the swap: %.3f ms\n", nowMs() - ta0);
		fprintf(stderr, "[s2amd] step %ld: build requested at step %ld (%s) %s: rc %d, solver %d/%d, copy dirty %d, %zu logged events, copy watches %d, strips %d, overflow in use %d\n",
				s->stepCounter, job->requestedAtStep, job->search ? "search" : "build", ok ? "ADOPTED" : "refused", job->rc, job->solverType, solverType, c->structureDirty ? 1 : 0,
				job->log.size(), c->watchedCount, c->dStripA.view.groupCount, c->stripInc.valid ? c->stripInc.overflowUsed : -1);
	}
	if (job->forced)
	{
		s->overflowRefusals = ok ? 0 : s->overflowRefusals + 1;
	}
	if (ok)
	{
		std::swap(static_cast<SolverStructure&>(*s), static_cast<SolverStructure&>(*c));
		s->watchedDirty = true; // (the copy never uploaded its watched bytes: the world chain does, before its next stage 3)
		// (contacts placed into the old structure meanwhile reset the graph's age; they are part of the new one: it is as settled as
		// a structure built this step would be -- and must not be taken for one that is due for a rebuild without strips)
		s->graphAge = std::max(s->graphAge, s->stripPatienceNow);
		s->asyncAdopted += 1;
		s->stripRetryPending = job->search ? false : s->stripRetryPending;
	}
	else if (job->search)
	{
		s->stripRetryPending = false; // (tried; a graph that changes asks again through buildStructure)
	}
	// the copy (holding the old structure after a swap) is freed by a thread of its own: hipFree waits for the device
	job->dropped = true;
	job->done.store(0, std::memory_order_release);
	job->worker = std::thread([job]() {
		devPoolThread(true);
		destroyClone(job->clone);
		job->clone = nullptr;
		job->done.store(1, std::memory_order_release);
	});
	return ok;
}
