// Host side of the C-ABI (include/solver2d_amd.h): device memory, graph colouring, the ten solver
// drivers as kernel-launch sequences, hipGraph capture/replay, timing.
//
// One s2amdSolver owns one HIP stream and all device state of one world.  Each driver below
// enqueues exactly the stage sequence of the reference driver it names; a "sweep" over contacts
// or joints is one launch per colour batch.  There is NO CPU fallback: without a gfx950 device
// s2amd_create fails with S2AMD_E_NODEVICE.

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
#include <vector>

namespace
{

thread_local std::string g_lastError;

int fail(int code, const std::string& msg)
{
	g_lastError = msg;
	return code;
}

} // namespace

// shared with broadphase.hip
int s2amdFail(int code, const std::string& msg)
{
	return fail(code, msg);
}

namespace
{

#define HIP_TRY(expr)                                                                                                            \
	do                                                                                                                           \
	{                                                                                                                            \
		hipError_t _e = (expr);                                                                                                  \
		if (_e != hipSuccess)                                                                                                    \
		{                                                                                                                        \
			return fail(S2AMD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                                     \
		}                                                                                                                        \
	} while (0)

double nowMs()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

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
		HIP_TRY(hipMalloc(&np, want));
		if (p)
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
		if (p)
		{
			(void)hipFree(p);
		}
		p = nullptr;
		bytes = 0;
	}
};

bool isPositionSolver(int type)
{
	return type == s2amd_solverPGS_NGS || type == s2amd_solverPGS_NGS_Block || type == s2amd_solverTGS_NGS || type == s2amd_solverXPBD;
}

// host copy of math.h:201-207 (same fp32 operations as the device helper)
bool rotIsFixedPoint(float s, float c)
{
	float mag = sqrtf(s * s + c * c);
	float invMag = mag > 0.0f ? 1.0f / mag : 0.0f;
	float ns = s * invMag, nc = c * invMag;
	return memcmp(&ns, &s, 4) == 0 && memcmp(&nc, &c, 4) == 0;
}

// Greedy colouring of a constraint graph.  edges[k] = {a, b} (b may equal -1 for one-body
// constraints); a body takes part in conflicts only when conflict[body] is true.  Constraints are
// visited in the given order and receive the lowest colour unused on both bodies, so the result is
// deterministic.  Returns colour per constraint and the colour count.
struct ColorMasks
{
	enum
	{
		WORDS = 4
	};
	std::vector<uint64_t> bits; // WORDS per body
	std::vector<std::vector<int>> overflow; // colours >= 64*WORDS (rare: bodies with hundreds of constraints)
};

// balanced (strip groups: one constraint per thread and colour round): a repair pass after the greedy pass
// evens out colours wider than one workgroup.
int colorGraph(const std::vector<int>& ea, const std::vector<int>& eb, const std::vector<uint8_t>& conflict, int bodyCount,
			   std::vector<int>& color, bool balanced = false)
{
	const int W = ColorMasks::WORDS;
	size_t n = ea.size();
	color.assign(n, 0);
	std::vector<int> population;
	std::vector<uint64_t> bits((size_t)bodyCount * W, 0);
	std::vector<std::vector<int>> extra;
	std::vector<int> extraIndex; // body -> index in extra or -1
	int colorCount = 0;
	for (size_t k = 0; k < n; ++k)
	{
		int a = ea[k], b = eb[k];
		bool ca = a >= 0 && conflict[a], cb = b >= 0 && b != a && conflict[b];
		int chosen = -1;
		for (int w = 0; w < W && chosen < 0; ++w)
		{
			uint64_t used = (ca ? bits[(size_t)a * W + w] : 0) | (cb ? bits[(size_t)b * W + w] : 0);
			if (~used)
			{
				chosen = w * 64 + __builtin_ctzll(~used);
			}
		}
		if (chosen < 0)
		{
			// all 256 fast colours taken on these bodies: linear probe in the overflow sets
			if (extraIndex.empty())
			{
				extraIndex.assign(bodyCount, -1);
			}
			auto usedIn = [&](int body, int c) {
				int ei = extraIndex[body];
				if (ei < 0)
				{
					return false;
				}
				const std::vector<int>& v = extra[ei];
				return std::find(v.begin(), v.end(), c) != v.end();
			};
			int c = 64 * W;
			while ((ca && usedIn(a, c)) || (cb && usedIn(b, c)))
			{
				c += 1;
			}
			chosen = c;
			auto mark = [&](int body) {
				if (extraIndex[body] < 0)
				{
					extraIndex[body] = (int)extra.size();
					extra.emplace_back();
				}
				extra[extraIndex[body]].push_back(chosen);
			};
			if (ca)
			{
				mark(a);
			}
			if (cb)
			{
				mark(b);
			}
		}
		else
		{
			if (ca)
			{
				bits[(size_t)a * W + chosen / 64] |= 1ull << (chosen % 64);
			}
			if (cb)
			{
				bits[(size_t)b * W + chosen / 64] |= 1ull << (chosen % 64);
			}
		}
		color[k] = chosen;
		colorCount = std::max(colorCount, chosen + 1);
		if (balanced)
		{
			if ((int)population.size() <= chosen)
			{
				population.resize((size_t)chosen + 1, 0);
			}
			population[(size_t)chosen] += 1;
		}
	}
	if (balanced && colorCount <= 64)
	{
		// repair pass: greedy fills the low colours first; move constraints out of colours wider than one
		// workgroup into the least populated colour that is free on both bodies (never adds a colour)
		const int cap = 256;
		for (size_t kk = n; kk-- > 0;)
		{
			int c = color[kk];
			if (population[(size_t)c] <= cap)
			{
				continue;
			}
			int a = ea[kk], b = eb[kk];
			bool ca = a >= 0 && conflict[a], cb = b >= 0 && b != a && conflict[b];
			uint64_t used = (ca ? bits[(size_t)a * W] : 0) | (cb ? bits[(size_t)b * W] : 0);
			int best = -1;
			for (int c2 = 0; c2 < colorCount; ++c2)
			{
				if (c2 != c && ((used >> c2) & 1ull) == 0 && population[(size_t)c2] < cap && (best < 0 || population[(size_t)c2] < population[(size_t)best]))
				{
					best = c2;
				}
			}
			if (best < 0)
			{
				continue;
			}
			if (ca)
			{
				bits[(size_t)a * W] = (bits[(size_t)a * W] & ~(1ull << c)) | (1ull << best);
			}
			if (cb)
			{
				bits[(size_t)b * W] = (bits[(size_t)b * W] & ~(1ull << c)) | (1ull << best);
			}
			color[kk] = best;
			population[(size_t)c] -= 1;
			population[(size_t)best] += 1;
		}
	}
	return colorCount;
}

// stable counting sort of constraint ids by colour
void sortByColor(const std::vector<int>& ids, const std::vector<int>& color, int colorCount, std::vector<int>& order, std::vector<int>& offsets)
{
	offsets.assign((size_t)colorCount + 1, 0);
	for (size_t k = 0; k < ids.size(); ++k)
	{
		offsets[(size_t)color[k] + 1] += 1;
	}
	for (int c = 0; c < colorCount; ++c)
	{
		offsets[(size_t)c + 1] += offsets[c];
	}
	std::vector<int> cursor(offsets.begin(), offsets.end() - 1);
	order.resize(ids.size());
	for (size_t k = 0; k < ids.size(); ++k)
	{
		order[(size_t)cursor[color[k]]++] = ids[k];
	}
}

// Launch batches from colour offsets.  Colours are launched one kernel each; when the colouring
// has a long run of tiny high colours (a body with dozens of constraints forces one colour per
// constraint) that run becomes ONE sequential tail batch instead of dozens of launches.
bool makeBatches(const std::vector<int>& colorOffsets, std::vector<int>& batchOffsets, bool allowTail = true)
{
	int n = (int)colorOffsets.size() - 1;
	batchOffsets.clear();
	if (n <= 0)
	{
		batchOffsets.push_back(0);
		return false;
	}
	int total = colorOffsets[n];
	// the tail starts at the first colour from which on EVERY colour is tiny (a launch would cost more
	// than sweeping its few constraints serially); it must replace at least kMinTailColors launches
	const int kTinyColor = 32, kMinTailColors = 4;
	int tailColor = n;
	for (int c = n - 1; c >= 1; --c)
	{
		if (colorOffsets[(size_t)c + 1] - colorOffsets[c] > kTinyColor)
		{
			break;
		}
		tailColor = c;
	}
	if (n - tailColor < kMinTailColors || (!allowTail && n <= 8))
	{
		tailColor = n; // strip groups with a handful of colours run them as preloaded rounds, however small
	}
	for (int c = 0; c <= tailColor; ++c)
	{
		batchOffsets.push_back(colorOffsets[c]);
	}
	if (tailColor < n)
	{
		batchOffsets.push_back(total);
		return true;
	}
	return false;
}

uint64_t fnv(uint64_t h, const void* data, size_t n)
{
	const unsigned char* p = (const unsigned char*)data;
	for (size_t i = 0; i < n; ++i)
	{
		h ^= p[i];
		h *= 1099511628211ull;
	}
	return h;
}

} // namespace

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

struct s2amdSolver
{
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t evBegin = nullptr, evEnd = nullptr;
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

	// host shadows of the graph structure (refreshed by every upload)
	std::vector<int> hContactA, hContactB, hContactPoints;
	std::vector<int> hJointType, hJointA, hJointB;
	std::vector<uint32_t> hBodyFlags; // S2F_WRITE_VEL / S2F_WRITE_POS from the wire bodies
	std::vector<uint8_t> hBodyLive, hBodyStatic;
	DevBuf dBodyFlags;

	// working SoA
	DevBuf soaBodies, soaContacts, soaJoints, dContactIndex, dJointIndex, dContactLocal, dJointLocal, dAdjOffsets, dAdjList, dOps;
	BodyView bv{};
	ContactView cv{};
	JointView jv{};
	uint64_t layoutGeneration = 0;
	int bodySoaCap = 0, contactSoaCap = 0, jointSoaCap = 0;

	// structure of the last step
	SweepSet contacts, joints;
	HostGroupTable hGroups, hContactTail, hJointTail, hStripA, hStripB;
	DeviceGroupTable dGroups, dContactTail, dJointTail, dStripA, dStripB;
	// lean strip tables (strip_kernel.hip): descriptors of both phases, warm-start slots of phase A
	DevBuf dStripLean;
	StripTableView leanA{}, leanB{};
	bool leanAValid = false, leanBValid = false;
	int optStripLean = 1;
	// persistent strip step (strip_kernel.hip: stripStepKernel)
	DevBuf dPersist, dGranules;
	PersistView persist{};
	bool persistValid = false;
	int persistRecordsWide = 0; // LDS records when a seam constraint takes 10 records (every kind but TGS_Soft's)
	DevBuf dPersistOps;
	int persistOpCount = 0;
	uint64_t persistOpsGeneration = ~0ull, persistOpsStructure = ~0ull;
	size_t granuleBytes = 0;
	int optPersist = 1;
	int optPersistDebug = 0;
	int optPersistSpinLimit = 1 << 21;
	bool persistFailed = false; // a hand-off timed out once (workgroups not co-resident: a shared GPU): multi-launch strips from then on
	int persistFallbacks = 0;
	int cuCount = 0;
	unsigned int* hostError = nullptr; // pinned, device-visible: a hand-off timed out
	unsigned long long* hostTimes = nullptr; // S2AMD_DEBUG_TIMES: pinned [256] phase time stamps of one workgroup
	DevBuf dMsg;
	MsgView msg{};
	bool msgTablesValid = false; // the global part is contact-only and has no sequential tail
	int optMessage = 0;	 // measured slower than the plain gather on MI355X: off by default
	int optBodyWarm = 1; // body-centric contact warm start (one launch per sweep instead of one per colour)
	int looseBodies = 0; // live non-static bodies that no LDS group owns
	int orderSolverClass = -1; // 0 velocity colouring, 1 position colouring
	bool orderGrouped = false;
	bool orderStrips = false;
	bool stripsRejected = false; // this graph's strip partition fits no strip kernel: colour batches until the graph changes
	int graphAge = 0;		  // steps solved since the constraint graph last changed
	int optStripPatience = 1; // steps of an unchanged graph before the (more expensive) strip structure is built
	int optStripsAnySolver = 0; // tests: strips for every solver and with joints (through the generic group interpreter)
	bool adjValid = false;
	bool structureDirty = true;
	uint64_t structureGeneration = 0;

	StepPlan plan;
	uint64_t planGeneration = 0;

	// options
	int optGraph = 1;
	int optProfile = 0;
	int optGroups = 1;
	int optMaxGroupBodies = 2048;
	int optPackGroupBodies = 1024;
	int optStrips = 1;		   // cut islands that do not fit one LDS group into strips of BFS levels (2 launches per sweep)
	int optStripBodies = 320;  // target bodies per strip
	int optStripMinBodies = 4096; // loose bodies below which the colour-batch path is kept

	// graph cache
	hipGraph_t graph = nullptr;
	hipGraphExec_t graphExec = nullptr;
	uint64_t graphKey = 0;

	// profiling events for the contact solve sweeps
	std::vector<hipEvent_t> sweepEvents;
	size_t sweepEventsUsed = 0;

	s2amdStepStats stats{};
	int launchCounter = 0;
	int graphLaunches = 0;
	DevBuf dGatherIndex;
	bool gatherIndexDirty = true;
	uint64_t opsGeneration = ~0ull;
};

namespace
{

StepConsts makeConsts(const s2amdStepParams* p)
{
	// src/world.c:170-202
	StepConsts sc;
	sc.dt = p->dt;
	sc.iterations = p->velIters;
	sc.extraIterations = p->posIters;
	sc.warmStart = p->warmStart != 0 ? 1 : 0;
	sc.inv_dt = p->dt > 0.0f ? 1.0f / p->dt : 0.0f;
	int type = p->solverType;
	if (type == s2amd_solverXPBD || type == s2amd_solverTGS_Soft || type == s2amd_solverTGS_Sticky || type == s2amd_solverTGS_NGS ||
		type == s2amd_solverSoftStep)
	{
		sc.h = sc.dt / sc.iterations;
		sc.inv_h = sc.inv_dt * sc.iterations;
	}
	else
	{
		sc.h = sc.dt;
		sc.inv_h = sc.inv_dt;
	}
	sc.gravityX = p->gravity[0];
	sc.gravityY = p->gravity[1];
	return sc;
}

// SoA carving: one device allocation per family, arrays laid end to end at 256-byte boundaries.
// The element capacity only grows (x1.5), so device pointers -- and a captured hipGraph -- stay
// valid until a family actually has to grow (layoutGeneration is bumped then).
struct Carver
{
	char* p;
	char* end;
	template <class T> T* take(size_t count)
	{
		char* r = p;
		p += (count * sizeof(T) + 255) & ~size_t(255);
		return (T*)r;
	}
};

int growFamily(s2amdSolver* s, DevBuf& buf, int& cap, int need, size_t slotBytes, int arrays)
{
	if (need <= cap && buf.p != nullptr)
	{
		return S2AMD_OK;
	}
	int newCap = std::max(std::max(need, 64), cap + cap / 2);
	buf.release();
	bool grew = false;
	int rc = buf.ensure((size_t)newCap * slotBytes + (size_t)arrays * 256, &grew);
	if (rc)
	{
		cap = 0;
		return rc;
	}
	cap = newCap;
	s->layoutGeneration += 1;
	return S2AMD_OK;
}

constexpr size_t kBodySlotBytes = sizeof(float4) * 4 + sizeof(float2) + sizeof(float) + sizeof(uint32_t);
constexpr size_t kContactSlotBytes = sizeof(int2) + sizeof(float4) * 2 + 2 * (sizeof(float4) * 5 + sizeof(float2)) + sizeof(float4) * 4;
constexpr size_t kJointSlotBytes = sizeof(int2) + sizeof(float4) * 8 + sizeof(float2) * 3;

int carveBodies(s2amdSolver* s, int n)
{
	int rc = growFamily(s, s->soaBodies, s->bodySoaCap, n, kBodySlotBytes, 8);
	if (rc)
	{
		return rc;
	}
	size_t cap = (size_t)s->bodySoaCap;
	Carver c{(char*)s->soaBodies.p, (char*)s->soaBodies.p + s->soaBodies.bytes};
	s->bv.vel = c.take<float4>(cap);
	s->bv.dq = c.take<float4>(cap);
	s->bv.integ = c.take<float4>(cap);
	s->bv.dq0 = c.take<float4>(cap);
	s->bv.pos = c.take<float2>(cap);
	s->bv.angDamp = c.take<float>(cap);
	s->bv.flags = c.take<uint32_t>(cap);
	s->bv.capacity = n;
	return c.p <= c.end ? S2AMD_OK : fail(S2AMD_E_DEVICE, "internal: body SoA carve overflow");
}

int carveContacts(s2amdSolver* s, int n)
{
	int rc = growFamily(s, s->soaContacts, s->contactSoaCap, n, kContactSlotBytes, 20);
	if (rc)
	{
		return rc;
	}
	size_t cap = (size_t)s->contactSoaCap;
	Carver c{(char*)s->soaContacts.p, (char*)s->soaContacts.p + s->soaContacts.bytes};
	ContactView& v = s->cv;
	v.bodies = c.take<int2>(cap);
	v.mass = c.take<float4>(cap);
	v.nf = c.take<float4>(cap);
	for (int j = 0; j < 2; ++j)
	{
		v.anchor[j] = c.take<float4>(cap);
		v.r0[j] = c.take<float4>(cap);
		v.param[j] = c.take<float4>(cap);
		v.soft[j] = c.take<float4>(cap);
		v.fanchor[j] = c.take<float4>(cap);
		v.impulse[j] = c.take<float2>(cap);
	}
	v.blockK = c.take<float4>(cap);
	v.blockNM = c.take<float4>(cap);
	v.deltaA = c.take<float4>(cap);
	v.deltaB = c.take<float4>(cap);
	return c.p <= c.end ? S2AMD_OK : fail(S2AMD_E_DEVICE, "internal: contact SoA carve overflow");
}

int carveJoints(s2amdSolver* s, int n)
{
	int rc = growFamily(s, s->soaJoints, s->jointSoaCap, n, kJointSlotBytes, 14);
	if (rc)
	{
		return rc;
	}
	size_t cap = (size_t)s->jointSoaCap;
	Carver c{(char*)s->soaJoints.p, (char*)s->soaJoints.p + s->soaJoints.bytes};
	JointView& j = s->jv;
	j.bodies = c.take<int2>(cap);
	j.frame = c.take<float4>(cap);
	j.mass = c.take<float4>(cap);
	j.pivot = c.take<float4>(cap);
	j.soft = c.take<float4>(cap);
	j.axial = c.take<float4>(cap);
	j.limits = c.take<float4>(cap);
	j.misc = c.take<float4>(cap);
	j.origin = c.take<float4>(cap);
	j.centerDiff0 = c.take<float2>(cap);
	j.impulse = c.take<float2>(cap);
	j.target = c.take<float2>(cap);
	return c.p <= c.end ? S2AMD_OK : fail(S2AMD_E_DEVICE, "internal: joint SoA carve overflow");
}

// ------------------------------------------------------------------------------------------------
// structure: islands -> LDS groups, colouring, sweep order, index tables
// ------------------------------------------------------------------------------------------------
struct UnionFind
{
	std::vector<int> parent;
	explicit UnionFind(int n) : parent((size_t)n)
	{
		for (int i = 0; i < n; ++i)
		{
			parent[i] = i;
		}
	}
	int find(int x)
	{
		while (parent[x] != x)
		{
			parent[x] = parent[parent[x]];
			x = parent[x];
		}
		return x;
	}
	void unite(int a, int b)
	{
		a = find(a), b = find(b);
		if (a != b)
		{
			// the lower index becomes the root: labels are deterministic
			if (a < b)
			{
				parent[b] = a;
			}
			else
			{
				parent[a] = b;
			}
		}
	}
};

struct EdgeList
{
	std::vector<int> ids, a, b; // wire index and endpoints (a == -1: one-body constraint)
};

// Colours one part (the global part or one group), appends its sweep order to `set` and returns its
// launch batches as ranges of k.  Endpoints are indices into `conflict`.
void colourPart(const std::vector<int>& ids, const std::vector<int>& ea, const std::vector<int>& eb, const std::vector<uint8_t>& conflict,
				int bodyCount, SweepSet& set, std::vector<int>& batchOffsetsOut, bool& hasTailOut, std::vector<int>* positions, bool balanced = false)
{
	std::vector<int> color, partOrder, partOffsets;
	int cc = colorGraph(ea, eb, conflict, bodyCount, color, balanced);
	// stable counting sort of positions by colour
	std::vector<int> pos(ids.size());
	for (size_t i = 0; i < ids.size(); ++i)
	{
		pos[i] = (int)i;
	}
	sortByColor(pos, color, cc, partOrder, partOffsets);
	int base = (int)set.order.size();
	for (int p : partOrder)
	{
		set.order.push_back(ids[p]);
	}
	if (positions)
	{
		*positions = partOrder;
	}
	for (int c = 0; c < cc; ++c)
	{
		if (set.colorOffsets.empty())
		{
			set.colorOffsets.push_back(0);
		}
		if (partOffsets[(size_t)c + 1] > partOffsets[c])
		{
			set.colorOffsets.push_back(base + partOffsets[(size_t)c + 1]);
		}
	}
	std::vector<int> rel;
	hasTailOut = makeBatches(partOffsets, rel, !balanced);
	batchOffsetsOut.clear();
	for (int r : rel)
	{
		batchOffsetsOut.push_back(base + r);
	}
}

int uploadGroupTable(s2amdSolver* s, const HostGroupTable& h, DeviceGroupTable& d)
{
	auto pad4 = [](size_t n) { return (n + 3) & ~size_t(3); };
	size_t nBO = pad4(h.bodyOffsets.size()), nBI = pad4(std::max<size_t>(h.bodyIds.size(), 1));
	size_t nCO = pad4(h.cBatchOffsets.size()), nJO = pad4(h.jBatchOffsets.size());
	size_t nCB = std::max<size_t>(h.cBatches.size(), 1) * 4, nJB = std::max<size_t>(h.jBatches.size(), 1) * 4;
	std::vector<int> blob(nBO + nBI + nCO + nJO + nCB + nJB, 0);
	size_t o = 0;
	size_t oBO = o;
	std::copy(h.bodyOffsets.begin(), h.bodyOffsets.end(), blob.begin() + o);
	o += nBO;
	size_t oBI = o;
	std::copy(h.bodyIds.begin(), h.bodyIds.end(), blob.begin() + o);
	o += nBI;
	size_t oCO = o;
	std::copy(h.cBatchOffsets.begin(), h.cBatchOffsets.end(), blob.begin() + o);
	o += nCO;
	size_t oJO = o;
	std::copy(h.jBatchOffsets.begin(), h.jBatchOffsets.end(), blob.begin() + o);
	o += nJO;
	size_t oCB = o;
	if (!h.cBatches.empty())
	{
		memcpy(blob.data() + o, h.cBatches.data(), h.cBatches.size() * sizeof(int4));
	}
	o += nCB;
	size_t oJB = o;
	if (!h.jBatches.empty())
	{
		memcpy(blob.data() + o, h.jBatches.data(), h.jBatches.size() * sizeof(int4));
	}
	bool grew = false;
	int rc = d.buf.ensure(blob.size() * sizeof(int), &grew);
	if (rc)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
	}
	HIP_TRY(hipMemcpyAsync(d.buf.p, blob.data(), blob.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
	const int* base = (const int*)d.buf.p;
	d.view.bodyOffsets = base + oBO;
	d.view.bodyIds = base + oBI;
	d.view.cBatchOffsets = base + oCO;
	d.view.jBatchOffsets = base + oJO;
	d.view.cBatches = (const int4*)(base + oCB);
	d.view.jBatches = (const int4*)(base + oJB);
	d.view.groupCount = h.count();
	d.maxBodies = h.maxBodies;
	return S2AMD_OK;
}

// Local body slots of one group: bodies get slots in order of first use by the group's constraints.
struct LocalSlots
{
	std::vector<int> slot, stamp;
	int epoch = 0;
	explicit LocalSlots(int nb) : slot((size_t)nb, -1), stamp((size_t)nb, -1) {}
	void begin() { epoch += 1; }
	void seed(int body, std::vector<int>& ids, bool owned)
	{
		stamp[body] = epoch;
		slot[body] = (int)ids.size();
		ids.push_back((int)((uint32_t)body | (owned ? S2G_OWNED : 0u)));
	}
	int get(int body, std::vector<int>& ids, const std::vector<uint8_t>& conflict)
	{
		if (stamp[body] != epoch)
		{
			stamp[body] = epoch;
			slot[body] = (int)ids.size();
			ids.push_back((int)((uint32_t)body | (conflict[body] ? S2G_OWNED : 0u)));
		}
		return slot[body];
	}
};

// Strips.  An island too big for one LDS group is cut along the level sets of a breadth-first search
// over its writable bodies: a constraint joins bodies of the same or of adjacent levels, so with every
// strip spanning >= 2 levels
//   * "interior" constraints (both bodies in one strip) of different strips share no writable body,
//   * "seam" constraints between strips i and i+1 touch the last level of i and the first of i+1
//     only, so different seams share no writable body either.
// A Gauss-Seidel sweep over the island is then TWO launches -- all interiors (phase A, one workgroup
// per strip, colours separated by __syncthreads), all seams (phase B) -- instead of one launch per
// colour; its sequential-equivalent order is strip by strip colour-major, then seam by seam.
struct StripPartition
{
	bool active = false;
	std::vector<std::vector<int>> bodies;  // per strip: owned bodies, level by level
	std::vector<std::vector<int>> cA, jA;  // per strip: interior contacts / joints (indices into the edge lists)
	std::vector<std::vector<int>> cB, jB;  // per seam i | i+1
};

void partitionStrips(const EdgeList& ce, const EdgeList& je, const std::vector<int>& cGlobal, const std::vector<int>& jGlobal,
					 const std::vector<uint8_t>& conflict, const std::vector<uint8_t>& loose, int nb, int targetBodies, int maxBodies,
					 StripPartition& out)
{
	// adjacency of the loose writable bodies
	auto linked = [&](int a, int b) { return a >= 0 && b >= 0 && conflict[a] && conflict[b] && loose[a] && loose[b]; };
	std::vector<int> deg((size_t)nb + 1, 0);
	auto countEdges = [&](const EdgeList& e, const std::vector<int>& ks) {
		for (int k : ks)
		{
			if (linked(e.a[k], e.b[k]))
			{
				deg[(size_t)e.a[k] + 1] += 1;
				deg[(size_t)e.b[k] + 1] += 1;
			}
		}
	};
	countEdges(ce, cGlobal);
	countEdges(je, jGlobal);
	for (int i = 0; i < nb; ++i)
	{
		deg[(size_t)i + 1] += deg[i];
	}
	std::vector<int> adj((size_t)deg[nb]), cursor(deg.begin(), deg.end() - 1);
	auto fillEdges = [&](const EdgeList& e, const std::vector<int>& ks) {
		for (int k : ks)
		{
			if (linked(e.a[k], e.b[k]))
			{
				adj[(size_t)cursor[e.a[k]]++] = e.b[k];
				adj[(size_t)cursor[e.b[k]]++] = e.a[k];
			}
		}
	};
	fillEdges(ce, cGlobal);
	fillEdges(je, jGlobal);

	// levels: per component, BFS from a pseudo-peripheral body (the last body a first BFS reaches)
	std::vector<int> level((size_t)nb, -1), queue, levelOffsets{0}, levelBodies;
	std::vector<int> seen((size_t)nb, 0);
	int epoch = 0;
	auto bfs = [&](int root, bool record) {
		epoch += 1;
		queue.clear();
		queue.push_back(root);
		seen[root] = epoch;
		size_t head = 0, levelEnd = 1;
		while (head < queue.size())
		{
			if (head == levelEnd)
			{
				if (record)
				{
					levelOffsets.push_back((int)levelBodies.size());
				}
				levelEnd = queue.size();
			}
			int u = queue[head++];
			if (record)
			{
				level[u] = (int)levelOffsets.size() - 1;
				levelBodies.push_back(u);
			}
			for (int e = deg[u]; e < deg[(size_t)u + 1]; ++e)
			{
				int v = adj[(size_t)e];
				if (seen[v] != epoch)
				{
					seen[v] = epoch;
					queue.push_back(v);
				}
			}
		}
		if (record)
		{
			levelOffsets.push_back((int)levelBodies.size());
		}
		return queue.back();
	};
	for (int i = 0; i < nb; ++i)
	{
		if (!loose[i] || level[i] >= 0)
		{
			continue;
		}
		int far = deg[(size_t)i + 1] > deg[i] ? bfs(i, false) : i;
		bfs(far, true);
	}
	const int levels = (int)levelOffsets.size() - 1;
	if (levels < 4)
	{
		return;
	}

	// strips: consecutive levels, >= 2 levels and >= targetBodies bodies each
	std::vector<int> stripOf((size_t)nb, -1);
	int curLevels = 0;
	out.bodies.emplace_back();
	for (int l = 0; l < levels; ++l)
	{
		if (curLevels >= 2 && (int)out.bodies.back().size() >= targetBodies)
		{
			out.bodies.emplace_back();
			curLevels = 0;
		}
		for (int e = levelOffsets[l]; e < levelOffsets[(size_t)l + 1]; ++e)
		{
			out.bodies.back().push_back(levelBodies[(size_t)e]);
		}
		curLevels += 1;
	}
	// a last strip of a single level is merged into its predecessor (both of its seams would meet in it)
	if (curLevels < 2 && out.bodies.size() >= 2)
	{
		std::vector<int> lastStrip = std::move(out.bodies.back());
		out.bodies.pop_back();
		out.bodies.back().insert(out.bodies.back().end(), lastStrip.begin(), lastStrip.end());
	}
	const int K = (int)out.bodies.size();
	if (K < 2)
	{
		out = StripPartition();
		return;
	}
	for (int i = 0; i < K; ++i)
	{
		for (int body : out.bodies[(size_t)i])
		{
			stripOf[body] = i;
		}
	}

	// classification
	out.cA.assign((size_t)K, {}), out.jA.assign((size_t)K, {});
	out.cB.assign((size_t)K - 1, {}), out.jB.assign((size_t)K - 1, {});
	bool ok = true;
	auto classify = [&](const EdgeList& e, const std::vector<int>& ks, std::vector<std::vector<int>>& A, std::vector<std::vector<int>>& B) {
		for (int k : ks)
		{
			int a = e.a[k], b = e.b[k];
			int sa = (a >= 0 && conflict[a]) ? stripOf[a] : -1;
			int sb = (b >= 0 && conflict[b]) ? stripOf[b] : -1;
			if (sa < 0 && sb < 0)
			{
				// no writable body: any strip will do (the sweep writes nothing)
				int any = (a >= 0 && stripOf[a] >= 0) ? stripOf[a] : ((b >= 0 && stripOf[b] >= 0) ? stripOf[b] : 0);
				A[(size_t)any].push_back(k);
			}
			else if (sa < 0 || sb < 0 || sa == sb)
			{
				A[(size_t)std::max(sa, sb)].push_back(k);
			}
			else if (sa - sb == 1 || sb - sa == 1)
			{
				B[(size_t)std::min(sa, sb)].push_back(k);
			}
			else
			{
				ok = false;
			}
		}
	};
	classify(ce, cGlobal, out.cA, out.cB);
	classify(je, jGlobal, out.jA, out.jB);

	// every group must fit the LDS body budget (owned bodies + read-only replicas)
	std::vector<int> stamp((size_t)nb, -1);
	int tick = 0;
	auto groupBodies = [&](const std::vector<int>& seedBodies, const std::vector<int>& cKs, const std::vector<int>& jKs) {
		tick += 1;
		int n = 0;
		auto touch = [&](int body) {
			if (body >= 0 && stamp[body] != tick)
			{
				stamp[body] = tick;
				n += 1;
			}
		};
		for (int body : seedBodies)
		{
			touch(body);
		}
		for (int k : cKs)
		{
			touch(ce.a[k]), touch(ce.b[k]);
		}
		for (int k : jKs)
		{
			touch(je.a[k]), touch(je.b[k]);
		}
		return n;
	};
	const std::vector<int> none;
	for (int i = 0; i < K && ok; ++i)
	{
		ok = groupBodies(out.bodies[(size_t)i], out.cA[(size_t)i], out.jA[(size_t)i]) <= maxBodies;
		if (ok && i + 1 < K)
		{
			ok = groupBodies(none, out.cB[(size_t)i], out.jB[(size_t)i]) <= maxBodies;
		}
	}
	if (!ok)
	{
		out = StripPartition();
		return;
	}
	out.active = true;
}

int buildStructure(s2amdSolver* s, int solverType)
{
	const int cls = isPositionSolver(solverType) ? 1 : 0;
	const bool needAdj = solverType == s2amd_solverJacobi;
	const bool grouped = s->optGroups != 0 && !needAdj;
	// strips pay off through the lean / persistent strip kernels, which exist for the soft contact sweeps
	const bool wantStrips = grouped && s->optStrips != 0 && !s->stripsRejected && s->graphAge >= s->optStripPatience &&
							(s->optStripsAnySolver != 0 || solverType == s2amd_solverTGS_Soft || solverType == s2amd_solverSoftStep ||
							 solverType == s2amd_solverPGS_Soft);
	if (!s->structureDirty && cls == s->orderSolverClass && grouped == s->orderGrouped && wantStrips == s->orderStrips && s->adjValid)
	{
		return S2AMD_OK;
	}
	double t0 = nowMs();
	const int nb = s->bodyCapacity;
	std::vector<uint8_t> conflict((size_t)nb);
	for (int i = 0; i < nb; ++i)
	{
		conflict[i] = (s->hBodyFlags[i] & (cls == 1 ? S2F_WRITE_POS : S2F_WRITE_VEL)) != 0;
	}

	// active constraints in pool order (the reference's gather: e.g. solve_tgs_soft.c:162-179)
	EdgeList ce, je;
	for (int i = 0; i < s->contactCapacity; ++i)
	{
		if (s->hContactPoints[i] > 0)
		{
			ce.ids.push_back(i);
			ce.a.push_back(s->hContactA[i]);
			ce.b.push_back(s->hContactB[i]);
		}
	}
	for (int i = 0; i < s->jointCapacity; ++i)
	{
		if (s->hJointType[i] != S2AMD_JOINT_FREE)
		{
			je.ids.push_back(i);
			je.a.push_back(s->hJointType[i] == S2AMD_JOINT_MOUSE ? -1 : s->hJointA[i]); // a mouse joint only touches body B
			je.b.push_back(s->hJointB[i]);
		}
	}
	const int C = (int)ce.ids.size(), J = (int)je.ids.size();

	// ---- islands: connected components over the writable bodies ----
	std::vector<int> cPart((size_t)C, -1), jPart((size_t)J, -1); // -1 = global part, else group id
	int groupCount = 0;
	std::vector<uint32_t> flags(s->hBodyFlags);
	if (grouped && (C > 0 || J > 0))
	{
		UnionFind uf(nb);
		auto link = [&](int a, int b) {
			if (a >= 0 && b >= 0 && conflict[a] && conflict[b])
			{
				uf.unite(a, b);
			}
		};
		for (int k = 0; k < C; ++k)
		{
			link(ce.a[k], ce.b[k]);
		}
		for (int k = 0; k < J; ++k)
		{
			link(je.a[k], je.b[k]);
		}
		auto rootOf = [&](int a, int b) {
			if (a >= 0 && conflict[a])
			{
				return uf.find(a);
			}
			if (b >= 0 && conflict[b])
			{
				return uf.find(b);
			}
			return -1;
		};
		// bodies an island would stage in LDS: its members that carry constraints + read-only replicas
		std::vector<int> islandBodies((size_t)nb, 0), seenBy((size_t)nb, -1), cRoot((size_t)C), jRoot((size_t)J);
		auto touch = [&](int body, int root) {
			if (body < 0 || root < 0)
			{
				return;
			}
			int key = conflict[body] ? -2 - root : root; // members are unique per island; replicas per (body, island)
			if (conflict[body])
			{
				if (seenBy[body] != -2)
				{
					seenBy[body] = -2;
					islandBodies[root] += 1;
				}
			}
			else if (seenBy[body] != key)
			{
				seenBy[body] = key; // approximate distinct count (exact when an immovable body's uses by one island are contiguous)
				islandBodies[root] += 1;
			}
		};
		for (int k = 0; k < C; ++k)
		{
			cRoot[k] = rootOf(ce.a[k], ce.b[k]);
			touch(ce.a[k], cRoot[k]);
			touch(ce.b[k], cRoot[k]);
		}
		for (int k = 0; k < J; ++k)
		{
			jRoot[k] = rootOf(je.a[k], je.b[k]);
			touch(je.a[k], jRoot[k]);
			touch(je.b[k], jRoot[k]);
		}
		// pack eligible islands into groups in order of first appearance
		std::vector<int> groupOfRoot((size_t)nb, -2); // -2 unassigned, -1 global
		int curBodies = 0;
		auto assign = [&](int root) {
			if (root < 0)
			{
				return -1;
			}
			if (groupOfRoot[root] != -2)
			{
				return groupOfRoot[root];
			}
			int n = islandBodies[root];
			if (n > s->optMaxGroupBodies)
			{
				groupOfRoot[root] = -1;
				return -1;
			}
			if (groupCount == 0 || curBodies + n > s->optPackGroupBodies)
			{
				groupCount += 1;
				curBodies = 0;
			}
			curBodies += n;
			groupOfRoot[root] = groupCount - 1;
			return groupCount - 1;
		};
		for (int k = 0; k < C; ++k)
		{
			cPart[k] = assign(cRoot[k]);
		}
		for (int k = 0; k < J; ++k)
		{
			jPart[k] = assign(jRoot[k]);
		}
	}

	// ---- per part lists (pool order is preserved inside every part) ----
	std::vector<std::vector<int>> cOf((size_t)groupCount + 1), jOf((size_t)groupCount + 1); // index 0 = global, g + 1 = group g
	for (int k = 0; k < C; ++k)
	{
		cOf[(size_t)cPart[k] + 1].push_back(k);
	}
	for (int k = 0; k < J; ++k)
	{
		jOf[(size_t)jPart[k] + 1].push_back(k);
	}

	SweepSet& cs = s->contacts;
	SweepSet& js = s->joints;
	cs = SweepSet();
	js = SweepSet();
	cs.colorOffsets.push_back(0);
	js.colorOffsets.push_back(0);
	s->hGroups.clear();
	s->hContactTail.clear();
	s->hJointTail.clear();
	s->hStripA.clear();
	s->hStripB.clear();

	// ---- strips: the part that fits no LDS group, cut along BFS level sets ----
	StripPartition strips;
	if (wantStrips && (s->optStripsAnySolver != 0 || jOf[0].empty()))
	{
		std::vector<uint8_t> ownedByIsland((size_t)nb, 0);
		auto mark = [&](int body) {
			if (body >= 0 && conflict[body])
			{
				ownedByIsland[body] = 1;
			}
		};
		for (int k = 0; k < C; ++k)
		{
			if (cPart[k] >= 0)
			{
				mark(ce.a[k]), mark(ce.b[k]);
			}
		}
		for (int k = 0; k < J; ++k)
		{
			if (jPart[k] >= 0)
			{
				mark(je.a[k]), mark(je.b[k]);
			}
		}
		std::vector<uint8_t> loose((size_t)nb);
		int looseCount = 0;
		for (int i = 0; i < nb; ++i)
		{
			loose[i] = s->hBodyLive[i] && !s->hBodyStatic[i] && !ownedByIsland[i];
			looseCount += loose[i];
		}
		if (looseCount >= s->optStripMinBodies)
		{
			partitionStrips(ce, je, cOf[0], jOf[0], conflict, loose, nb, s->optStripBodies, s->optMaxGroupBodies, strips);
		}
		if (strips.active)
		{
			cOf[0].clear();
			jOf[0].clear();
		}
	}

	LocalSlots slots(nb);
	auto gather = [&](const EdgeList& e, const std::vector<int>& ks, std::vector<int>& ids, std::vector<int>& a, std::vector<int>& b) {
		ids.clear(), a.clear(), b.clear();
		for (int k : ks)
		{
			ids.push_back(e.ids[k]);
			a.push_back(e.a[k]);
			b.push_back(e.b[k]);
		}
	};

	// global part: colour batches over HBM-resident bodies (+ a sequential tail as a one-group LDS table)
	{
		std::vector<int> ids, a, b, pos;
		gather(ce, cOf[0], ids, a, b);
		colourPart(ids, a, b, conflict, nb, cs, cs.batchOffsets, cs.hasTail, &pos);
		cs.globalCount = (int)ids.size();
		cs.local.assign((size_t)cs.globalCount, make_int2(0, 0));
		if (cs.hasTail)
		{
			HostGroupTable& t = s->hContactTail;
			int begin = cs.batchOffsets[cs.batchOffsets.size() - 2], end = cs.batchOffsets.back();
			slots.begin();
			std::vector<int> bodies;
			for (int k = begin; k < end; ++k)
			{
				int p = pos[(size_t)k];
				cs.local[(size_t)k] = make_int2(slots.get(a[p], bodies, conflict), slots.get(b[p], bodies, conflict));
			}
			t.bodyIds = bodies;
			t.bodyOffsets = {0, (int)bodies.size()};
			t.cBatches.push_back(make_int4(begin, end, 1, 0));
			t.cBatchOffsets = {0, 1};
			t.jBatchOffsets = {0, 0};
			t.maxBodies = (int)bodies.size();
		}
		gather(je, jOf[0], ids, a, b);
		colourPart(ids, a, b, conflict, nb, js, js.batchOffsets, js.hasTail, &pos);
		js.globalCount = (int)ids.size();
		js.local.assign((size_t)js.globalCount, make_int2(0, 0));
		if (js.hasTail)
		{
			HostGroupTable& t = s->hJointTail;
			int begin = js.batchOffsets[js.batchOffsets.size() - 2], end = js.batchOffsets.back();
			slots.begin();
			std::vector<int> bodies;
			for (int k = begin; k < end; ++k)
			{
				int p = pos[(size_t)k];
				int la = a[p] >= 0 ? slots.get(a[p], bodies, conflict) : 0;
				js.local[(size_t)k] = make_int2(la, slots.get(b[p], bodies, conflict));
			}
			t.bodyIds = bodies;
			t.bodyOffsets = {0, (int)bodies.size()};
			t.jBatches.push_back(make_int4(begin, end, 1, 0));
			t.jBatchOffsets = {0, 1};
			t.cBatchOffsets = {0, 0};
			t.maxBodies = (int)bodies.size();
		}
	}

	// one LDS group: local body slots (seeded bodies first: owned, in the given order), colour batches of
	// its contacts and joints appended to the sweep sets, one row in table `t`
	auto emitGroup = [&](HostGroupTable& t, const std::vector<int>& cKs, const std::vector<int>& jKs, const std::vector<int>& seedBodies,
						 const std::vector<int>* replicaOf = nullptr, const std::vector<int>* replicaOf2 = nullptr) {
		std::vector<int> ids, a, b, bodies, la, lb, pos, batchOffsets;
		bool tail = false;
		slots.begin();
		for (int body : seedBodies)
		{
			slots.seed(body, bodies, true);
		}
		for (const std::vector<int>* list : {replicaOf, replicaOf2})
		{
			if (!list)
			{
				continue;
			}
			// read-only bodies of the seams this strip also sweeps in the persistent kernel (strip_kernel.hip)
			for (int k : *list)
			{
				if (ce.a[k] >= 0 && !conflict[ce.a[k]])
				{
					slots.get(ce.a[k], bodies, conflict);
				}
				if (ce.b[k] >= 0 && !conflict[ce.b[k]])
				{
					slots.get(ce.b[k], bodies, conflict);
				}
			}
		}
		// contacts
		gather(ce, cKs, ids, a, b);
		la.resize(ids.size()), lb.resize(ids.size());
		for (size_t i = 0; i < ids.size(); ++i)
		{
			la[i] = slots.get(a[i], bodies, conflict);
			lb[i] = slots.get(b[i], bodies, conflict);
		}
		// joints (slots first so both families share one body list)
		std::vector<int> jids, ja, jb, jla, jlb;
		gather(je, jKs, jids, ja, jb);
		jla.resize(jids.size()), jlb.resize(jids.size());
		for (size_t i = 0; i < jids.size(); ++i)
		{
			jla[i] = ja[i] >= 0 ? slots.get(ja[i], bodies, conflict) : -1;
			jlb[i] = slots.get(jb[i], bodies, conflict);
		}
		// colouring conflicts are the writable bodies (an owned kinematic body is shareable in velocity sweeps)
		std::vector<uint8_t> lconf(bodies.size());
		for (size_t i = 0; i < bodies.size(); ++i)
		{
			lconf[i] = conflict[(size_t)((uint32_t)bodies[i] & ~S2G_OWNED)];
		}
		colourPart(ids, la, lb, lconf, (int)bodies.size(), cs, batchOffsets, tail, &pos, &t != &s->hGroups);
		for (size_t i = 0; i < pos.size(); ++i)
		{
			cs.local.push_back(make_int2(la[(size_t)pos[i]], lb[(size_t)pos[i]]));
		}
		for (size_t bi = 0; bi + 1 < batchOffsets.size(); ++bi)
		{
			bool isTail = tail && bi + 2 == batchOffsets.size();
			if (batchOffsets[bi + 1] > batchOffsets[bi])
			{
				t.cBatches.push_back(make_int4(batchOffsets[bi], batchOffsets[bi + 1], isTail ? 1 : 0, 0));
			}
		}
		t.cBatchOffsets.push_back((int)t.cBatches.size());
		colourPart(jids, jla, jlb, lconf, (int)bodies.size(), js, batchOffsets, tail, &pos);
		for (size_t i = 0; i < pos.size(); ++i)
		{
			js.local.push_back(make_int2(std::max(jla[(size_t)pos[i]], 0), jlb[(size_t)pos[i]]));
		}
		for (size_t bi = 0; bi + 1 < batchOffsets.size(); ++bi)
		{
			bool isTail = tail && bi + 2 == batchOffsets.size();
			if (batchOffsets[bi + 1] > batchOffsets[bi])
			{
				t.jBatches.push_back(make_int4(batchOffsets[bi], batchOffsets[bi + 1], isTail ? 1 : 0, 0));
			}
		}
		t.jBatchOffsets.push_back((int)t.jBatches.size());
		for (int id : bodies)
		{
			t.bodyIds.push_back(id);
			if ((uint32_t)id & S2G_OWNED)
			{
				flags[(size_t)((uint32_t)id & ~S2G_OWNED)] |= S2F_IN_GROUP;
			}
		}
		t.bodyOffsets.push_back((int)t.bodyIds.size());
		t.maxBodies = std::max(t.maxBodies, (int)bodies.size());
	};

	// LDS groups: whole-step kernel, bodies in LDS
	const std::vector<int> noSeed;
	for (int g = 0; g < groupCount; ++g)
	{
		emitGroup(s->hGroups, cOf[(size_t)g + 1], jOf[(size_t)g + 1], noSeed);
	}

	// strips of the big islands: phase A = interiors (own every body of the strip), phase B = seams
	const int stripBaseC = (int)cs.order.size(), stripBaseJ = (int)js.order.size();
	for (size_t i = 0; i < strips.bodies.size(); ++i)
	{
		emitGroup(s->hStripA, strips.cA[i], strips.jA[i], strips.bodies[i], i < strips.cB.size() ? &strips.cB[i] : nullptr,
				  i > 0 ? &strips.cB[i - 1] : nullptr);
	}
	int stripInterior = (int)cs.order.size(), stripInteriorJ = (int)js.order.size();
	std::vector<int> seamGroup(strips.cB.size(), -1);
	for (size_t i = 0; i < strips.cB.size(); ++i)
	{
		if (!strips.cB[i].empty() || !strips.jB[i].empty())
		{
			seamGroup[i] = s->hStripB.count();
			emitGroup(s->hStripB, strips.cB[i], strips.jB[i], noSeed);
		}
	}
	if (strips.active)
	{
		cs.stripCount = (int)cs.order.size() - stripBaseC;
		js.stripCount = (int)js.order.size() - stripBaseJ;
		cs.seamCount = (int)cs.order.size() - stripInterior;
		js.seamCount = (int)js.order.size() - stripInteriorJ;
	}

	s->looseBodies = 0;
	for (int i = 0; i < nb; ++i)
	{
		if (s->hBodyLive[i] && !s->hBodyStatic[i] && (flags[i] & S2F_IN_GROUP) == 0)
		{
			s->looseBodies += 1;
		}
	}

	// ---- device tables ----
	int rc;
	if ((rc = carveContacts(s, C)) != 0 || (rc = carveJoints(s, J)) != 0)
	{
		return rc;
	}
	bool grew = false;
	if ((rc = s->dContactIndex.ensure((size_t)std::max(C, 1) * sizeof(int), &grew)) != 0 ||
		(rc = s->dJointIndex.ensure((size_t)std::max(J, 1) * sizeof(int), &grew)) != 0 ||
		(rc = s->dContactLocal.ensure((size_t)std::max(C, 1) * sizeof(int2), &grew)) != 0 ||
		(rc = s->dJointLocal.ensure((size_t)std::max(J, 1) * sizeof(int2), &grew)) != 0)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
	}
	if (C > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dContactIndex.p, cs.order.data(), (size_t)C * sizeof(int), hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipMemcpyAsync(s->dContactLocal.p, cs.local.data(), (size_t)C * sizeof(int2), hipMemcpyHostToDevice, s->stream));
	}
	if (J > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dJointIndex.p, js.order.data(), (size_t)J * sizeof(int), hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipMemcpyAsync(s->dJointLocal.p, js.local.data(), (size_t)J * sizeof(int2), hipMemcpyHostToDevice, s->stream));
	}
	if (nb > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dBodyFlags.p, flags.data(), (size_t)nb * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
	}
	s->cv.contactIndex = (int*)s->dContactIndex.p;
	s->cv.localBodies = (int2*)s->dContactLocal.p;
	s->cv.count = C;
	s->jv.jointIndex = (int*)s->dJointIndex.p;
	s->jv.localBodies = (int2*)s->dJointLocal.p;
	s->jv.count = J;
	if ((rc = uploadGroupTable(s, s->hGroups, s->dGroups)) != 0 || (rc = uploadGroupTable(s, s->hContactTail, s->dContactTail)) != 0 ||
		(rc = uploadGroupTable(s, s->hJointTail, s->dJointTail)) != 0 || (rc = uploadGroupTable(s, s->hStripA, s->dStripA)) != 0 ||
		(rc = uploadGroupTable(s, s->hStripB, s->dStripB)) != 0)
	{
		return rc;
	}

	// ---- lean strip tables: per-group descriptors + warm-start slots (strip_kernel.hip) ----
	s->leanAValid = s->leanBValid = false;
	s->persistValid = false;
	s->leanA = StripTableView{};
	s->leanB = StripTableView{};
	if (strips.active && s->optStripLean)
	{
		const int k0 = stripBaseC, k1 = stripBaseC + cs.stripCount;
		// body -> incident strip constraints in sweep order
		std::vector<int> off((size_t)nb + 1, 0), inc;
		for (int k = k0; k < k1; ++k)
		{
			int a = s->hContactA[cs.order[(size_t)k]], b = s->hContactB[cs.order[(size_t)k]];
			off[(size_t)a + 1] += conflict[a] ? 1 : 0;
			off[(size_t)b + 1] += conflict[b] ? 1 : 0;
		}
		for (int i = 0; i < nb; ++i)
		{
			off[(size_t)i + 1] += off[i];
		}
		inc.resize((size_t)off[nb]);
		{
			std::vector<int> cur(off.begin(), off.end() - 1);
			for (int k = k0; k < k1; ++k)
			{
				int a = s->hContactA[cs.order[(size_t)k]], b = s->hContactB[cs.order[(size_t)k]];
				if (conflict[a])
				{
					inc[(size_t)cur[a]++] = (k << 1) | 0;
				}
				if (conflict[b])
				{
					inc[(size_t)cur[b]++] = (k << 1) | 1;
				}
			}
		}
		std::vector<StripDesc> descA, descB;
		std::vector<int2> slotList;
		std::vector<int> slotOffsets;
		int maxRounds = 0;
		bool persistTablesOk = false;
		auto describe = [&](const HostGroupTable& t, std::vector<StripDesc>& out, bool withSlots, int& ldsRecords) {
			bool ok = true;
			ldsRecords = 0;
			maxRounds = 0;
			for (int g = 0; g < t.count() && ok; ++g)
			{
				StripDesc d{};
				d.bodyBase = t.bodyOffsets[(size_t)g];
				d.bodyCount = t.bodyOffsets[(size_t)g + 1] - d.bodyBase;
				int b0 = t.cBatchOffsets[(size_t)g], b1 = t.cBatchOffsets[(size_t)g + 1];
				d.batchCount = b1 - b0;
				ok = d.batchCount <= (withSlots ? S2_STRIP_ROUNDS_MAX : S2_STRIP_ROUNDS) && d.bodyCount <= S2_STRIP_BODY_CHUNKS * 256;
				maxRounds = std::max(maxRounds, d.batchCount);
				for (int b = b0; b < b1 && ok; ++b)
				{
					int4 bt = t.cBatches[(size_t)b];
					ok = bt.z == 0;
					d.batch[b - b0] = make_int4(bt.x, bt.y, 0, 0);
				}
				while (d.ownedCount < d.bodyCount && ((uint32_t)t.bodyIds[(size_t)d.bodyBase + d.ownedCount] & S2G_OWNED) != 0)
				{
					d.ownedCount += 1;
				}
				if (withSlots)
				{
					// phase A groups list their owned bodies first (seeded): slots in body order
					d.slotBase = (int)slotList.size();
					d.slotOffBase = (int)slotOffsets.size();
					for (int i = 0; i < d.ownedCount; ++i)
					{
						int body = (int)((uint32_t)t.bodyIds[(size_t)d.bodyBase + i] & ~S2G_OWNED);
						slotOffsets.push_back((int)slotList.size() - d.slotBase);
						for (int e = off[body]; e < off[(size_t)body + 1]; ++e)
						{
							slotList.push_back(make_int2(inc[(size_t)e], i));
						}
					}
					slotOffsets.push_back((int)slotList.size() - d.slotBase);
					d.slotCount = (int)slotList.size() - d.slotBase;
				}
				int records = 2 * d.bodyCount + 2 * d.slotCount;
				ok = ok && records <= (160 * 1024) / 16;
				ldsRecords = std::max(ldsRecords, records);
				out.push_back(d);
			}
			return ok;
		};
		int ldsA = 0, ldsB = 0;
		bool okA = describe(s->hStripA, descA, true, ldsA);
		const int maxRoundsA = maxRounds; // <= 8: the persistent kernel's wide variant; <= 6: also the lean launches
		bool okB = describe(s->hStripB, descB, false, ldsB);
		// owned bodies must be exactly the seeded prefix in phase A (replicas are never owned there)
		if (okA)
		{
			auto pad = [](size_t n) { return (n + 63) & ~size_t(63); };
			size_t bA = pad(descA.size() * sizeof(StripDesc)), bB = pad(std::max<size_t>(descB.size(), 1) * sizeof(StripDesc));
			size_t bS = pad(std::max<size_t>(slotList.size(), 1) * sizeof(int2)), bO = pad(std::max<size_t>(slotOffsets.size(), 1) * sizeof(int));
			std::vector<unsigned char> blob(bA + bB + bS + bO, 0);
			memcpy(blob.data(), descA.data(), descA.size() * sizeof(StripDesc));
			if (!descB.empty())
			{
				memcpy(blob.data() + bA, descB.data(), descB.size() * sizeof(StripDesc));
			}
			if (!slotList.empty())
			{
				memcpy(blob.data() + bA + bB, slotList.data(), slotList.size() * sizeof(int2));
			}
			if (!slotOffsets.empty())
			{
				memcpy(blob.data() + bA + bB + bS, slotOffsets.data(), slotOffsets.size() * sizeof(int));
			}
			bool grewLean = false;
			if ((rc = s->dStripLean.ensure(blob.size(), &grewLean)) != 0)
			{
				return rc;
			}
			if (grewLean)
			{
				s->layoutGeneration += 1;
			}
			HIP_TRY(hipMemcpyAsync(s->dStripLean.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s->stream));
			HIP_TRY(hipStreamSynchronize(s->stream)); // blob is a local
			const unsigned char* base = (const unsigned char*)s->dStripLean.p;
			s->leanA.descs = (const StripDesc*)base;
			s->leanA.bodyIds = s->dStripA.view.bodyIds;
			s->leanA.slots = (const int2*)(base + bA + bB);
			s->leanA.slotOffsets = (const int*)(base + bA + bB + bS);
			s->leanA.groupCount = (int)descA.size();
			s->leanA.ldsRecords = ldsA;
			s->leanAValid = maxRoundsA <= S2_STRIP_ROUNDS;
			persistTablesOk = okB;
			if (okB)
			{
				s->leanB.descs = (const StripDesc*)(base + bA);
				s->leanB.bodyIds = s->dStripB.view.bodyIds;
				s->leanB.slots = s->leanA.slots;
				s->leanB.slotOffsets = s->leanA.slotOffsets;
				s->leanB.groupCount = (int)descB.size();
				s->leanB.ldsRecords = ldsB;
				s->leanBValid = true;
			}
		}

		if (getenv("S2AMD_DEBUG"))
		{
			fprintf(stderr, "[s2amd] strips: %d strips, %d seams, lean A %d B %d, strip joints %d, CUs %d\n", s->hStripA.count(), s->hStripB.count(),
					(int)s->leanAValid, (int)s->leanBValid, js.stripCount, s->cuCount);
		}
		// ---- persistent strip step (strip_kernel.hip: stripStepKernel): per workgroup both seams' remaps, the
		// import / export lists of the symmetric exchange, warm-start term slots, granule buffers ----
		s->persistValid = false;
		if (persistTablesOk && js.stripCount == 0 && s->optPersist && s->hostError != nullptr && s->hStripA.count() <= s->cuCount)
		{
			const HostGroupTable& A = s->hStripA;
			const HostGroupTable& B = s->hStripB;
			const int K = A.count();
			bool ok = true;
			std::vector<int> ownerGroup((size_t)nb, -1), ownerSlot((size_t)nb, -1);
			for (int gi = 0; gi < K; ++gi)
			{
				for (int e = A.bodyOffsets[(size_t)gi]; e < A.bodyOffsets[(size_t)gi + 1]; ++e)
				{
					uint32_t id = (uint32_t)A.bodyIds[(size_t)e];
					if (id & S2G_OWNED)
					{
						ownerGroup[id & ~S2G_OWNED] = gi;
						ownerSlot[id & ~S2G_OWNED] = e - A.bodyOffsets[(size_t)gi];
					}
				}
				for (int bb = A.cBatchOffsets[(size_t)gi]; bb < A.cBatchOffsets[(size_t)gi + 1]; ++bb)
				{
					ok = ok && A.cBatches[(size_t)bb].y - A.cBatches[(size_t)bb].x <= 256; // one constraint per thread and round
				}
			}
			// seams: bodies on either side, in the order of the seam group's body list
			const int S = K - 1;
			std::vector<std::vector<int>> leftBodies((size_t)std::max(S, 0)), rightBodies((size_t)std::max(S, 0));
			std::vector<int> posInSeam((size_t)nb, -1);
			for (int sm = 0; sm < S && ok; ++sm)
			{
				int g = seamGroup[(size_t)sm];
				if (g < 0)
				{
					continue;
				}
				for (int e = B.bodyOffsets[(size_t)g]; e < B.bodyOffsets[(size_t)g + 1]; ++e)
				{
					int body = (int)((uint32_t)B.bodyIds[(size_t)e] & ~S2G_OWNED);
					if (!conflict[body])
					{
						continue;
					}
					if (ownerGroup[body] == sm)
					{
						posInSeam[body] = (int)leftBodies[(size_t)sm].size();
						leftBodies[(size_t)sm].push_back(body);
					}
					else if (ownerGroup[body] == sm + 1)
					{
						posInSeam[body] = (int)rightBodies[(size_t)sm].size();
						rightBodies[(size_t)sm].push_back(body);
					}
					else
					{
						ok = false;
					}
				}
				ok = ok && leftBodies[(size_t)sm].size() <= 256 && rightBodies[(size_t)sm].size() <= 256;
			}
			// granule buffers: per seam {toLeft: 4 per right body, toRight: 4 per left body}, two parities
			std::vector<int> seamBase((size_t)std::max(S, 0), 0);
			int granules = 0;
			for (int sm = 0; sm < S; ++sm)
			{
				seamBase[(size_t)sm] = granules;
				granules += 4 * (int)(leftBodies[(size_t)sm].size() + rightBodies[(size_t)sm].size());
			}
			const int parityStride = granules;
			std::vector<PersistDesc> descs((size_t)K);
			std::vector<int> remap, exportSrc, importIds;
			std::vector<int> replicaStamp((size_t)nb, -1), replicaSlot((size_t)nb, -1);
			int ldsRecords = 0, ldsRecordsWide = 0;
			for (int i = 0; i < K && ok; ++i)
			{
				PersistDesc& d = descs[(size_t)i];
				memset(&d, 0, sizeof(d));
				const int bodyBase = A.bodyOffsets[(size_t)i];
				const int nbA = A.bodyOffsets[(size_t)i + 1] - bodyBase;
				for (int e = bodyBase; e < bodyBase + nbA; ++e)
				{
					uint32_t id = (uint32_t)A.bodyIds[(size_t)e];
					if ((id & S2G_OWNED) == 0)
					{
						replicaStamp[id] = i;
						replicaSlot[id] = e - bodyBase;
					}
				}
				const int seamOf[2] = {i - 1, i};
				int importOffset = nbA, seamSlots = 0;
				for (int side = 0; side < 2; ++side)
				{
					const int sm = seamOf[side];
					const int g = (sm >= 0 && sm < S) ? seamGroup[(size_t)sm] : -1;
					d.importIdBase[side] = (int)importIds.size();
					d.exportSrcBase[side] = (int)exportSrc.size();
					d.remapBase[side] = (int)remap.size();
					if (g < 0)
					{
						continue;
					}
					// side 0: I am the RIGHT strip of seam i-1 (import its left bodies, export its right bodies);
					// side 1: I am the LEFT strip of seam i
					const std::vector<int>& imports = side == 0 ? leftBodies[(size_t)sm] : rightBodies[(size_t)sm];
					const std::vector<int>& exports = side == 0 ? rightBodies[(size_t)sm] : leftBodies[(size_t)sm];
					d.importCount[side] = (int)imports.size();
					d.exportCount[side] = (int)exports.size();
					importIds.insert(importIds.end(), imports.begin(), imports.end());
					for (int body : exports)
					{
						exportSrc.push_back(ownerSlot[body]);
					}
					const int nR = (int)rightBodies[(size_t)sm].size();
					const int toLeft = seamBase[(size_t)sm], toRight = seamBase[(size_t)sm] + 4 * nR;
					d.inBase[side] = side == 0 ? toRight : toLeft;
					d.outBase[side] = side == 0 ? toLeft : toRight;
					for (int e = B.bodyOffsets[(size_t)g]; e < B.bodyOffsets[(size_t)g + 1] && ok; ++e)
					{
						int body = (int)((uint32_t)B.bodyIds[(size_t)e] & ~S2G_OWNED);
						if (ownerGroup[body] == i)
						{
							remap.push_back(ownerSlot[body]);
						}
						else if (conflict[body])
						{
							remap.push_back(importOffset + posInSeam[body]);
						}
						else if (replicaStamp[body] == i)
						{
							remap.push_back(replicaSlot[body]);
						}
						else
						{
							ok = false;
						}
					}
					int b0 = B.cBatchOffsets[(size_t)g], b1 = B.cBatchOffsets[(size_t)g + 1];
					d.seamBatchCount[side] = b1 - b0;
					ok = ok && b1 - b0 <= S2_PERSIST_B_ROUNDS;
					for (int bb = b0; bb < b1 && ok; ++bb)
					{
						int4 bt = B.cBatches[(size_t)bb];
						ok = bt.z == 0;
						d.seamBatch[side][bb - b0] = make_int2(bt.x, bt.y);
						seamSlots += bt.y - bt.x;
					}
					importOffset += d.importCount[side];
				}
				for (int r = 0; r < S2_PERSIST_B_ROUNDS && ok; ++r)
				{
					int n0 = r < d.seamBatchCount[0] ? d.seamBatch[0][r].y - d.seamBatch[0][r].x : 0;
					int n1 = r < d.seamBatchCount[1] ? d.seamBatch[1][r].y - d.seamBatch[1][r].x : 0;
					ok = n0 + n1 <= 512; // both seams share a round: at most two constraints per thread
				}
				const int nt = importOffset;
				// bodies, seam constraints (8 records each for TGS_Soft, 10 for the other kinds)
				int fixedRecords = 3 * nt + (nt + 3) / 4; // velocity, pose, integrator constants, angular damping
				ok = ok && fixedRecords + 8 * seamSlots + 2 * 128 <= (160 * 1024) / 16 && nt < 16384;
				ldsRecords = std::max(ldsRecords, fixedRecords + 8 * seamSlots);
				ldsRecordsWide = std::max(ldsRecordsWide, fixedRecords + 10 * seamSlots);
			}
			if (getenv("S2AMD_DEBUG"))
			{
				fprintf(stderr, "[s2amd] persistent step: %s (K=%d, lds records %d, granules/parity %d)\n", ok ? "eligible" : "NOT eligible", K, ldsRecords,
						parityStride);
			}
			if (ok)
			{
				auto pad = [](size_t n) { return (n + 63) & ~size_t(63); };
				auto bytesOf = [&](size_t n, size_t elem) { return pad(std::max<size_t>(n, 1) * elem); };
				size_t o0 = 0, o1 = o0 + bytesOf(descs.size(), sizeof(PersistDesc)), o2 = o1 + bytesOf(remap.size(), sizeof(int));
				size_t o3 = o2 + bytesOf(exportSrc.size(), sizeof(int)), o4 = o3 + bytesOf(importIds.size(), sizeof(int));
				size_t o5 = o4 + 256; // the device-side "hand-off timed out" word
				std::vector<unsigned char> blob(o5, 0);
				auto put = [&](size_t at, const void* src, size_t bytes) {
					if (bytes)
					{
						memcpy(blob.data() + at, src, bytes);
					}
				};
				put(o0, descs.data(), descs.size() * sizeof(PersistDesc));
				put(o1, remap.data(), remap.size() * sizeof(int));
				put(o2, exportSrc.data(), exportSrc.size() * sizeof(int));
				put(o3, importIds.data(), importIds.size() * sizeof(int));
				bool grewP = false;
				s->granuleBytes = ((std::max<size_t>((size_t)2 * parityStride, 1) * sizeof(unsigned long long)) + 255) & ~size_t(255);
				if ((rc = s->dPersist.ensure(blob.size(), &grewP)) != 0 || (rc = s->dGranules.ensure(s->granuleBytes, &grewP)) != 0)
				{
					return rc;
				}
				if (grewP)
				{
					s->layoutGeneration += 1;
				}
				HIP_TRY(hipMemcpyAsync(s->dPersist.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s->stream));
				HIP_TRY(hipStreamSynchronize(s->stream));
				const unsigned char* base = (const unsigned char*)s->dPersist.p;
				PersistView& pv = s->persist;
				pv = PersistView{};
				pv.descs = (const PersistDesc*)(base + o0);
				pv.remap = (const int*)(base + o1);
				pv.exportSrc = (const int*)(base + o2);
				pv.importIds = (const int*)(base + o3);
				pv.granules = (unsigned long long*)s->dGranules.p;
				unsigned int* devError = nullptr;
				HIP_TRY(hipHostGetDevicePointer((void**)&devError, s->hostError, 0));
				pv.error = devError;
				pv.deviceError = (unsigned int*)(base + o4);
				pv.parityStride = parityStride;
				// fresh buffers start from zero tags
				HIP_TRY(hipMemsetAsync(s->dGranules.p, 0, s->granuleBytes, s->stream));
				pv.wideRounds = maxRoundsA > S2_STRIP_ROUNDS ? 1 : 0;
				pv.allTwoPoints = 1;
				for (int k = k0; k < k1; ++k)
				{
					if (s->hContactPoints[(size_t)cs.order[(size_t)k]] != 2)
					{
						pv.allTwoPoints = 0;
						break;
					}
				}
				pv.ldsRecords = ldsRecords;
				s->persistRecordsWide = ldsRecordsWide;
				pv.debugSkip = s->optPersistDebug;
				pv.spinLimit = (unsigned int)s->optPersistSpinLimit;
				pv.debugTimes = nullptr;
				if (getenv("S2AMD_DEBUG_TIMES"))
				{
					if (!s->hostTimes && hipHostMalloc((void**)&s->hostTimes, 256 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess)
					{
						s->hostTimes = nullptr;
						(void)hipGetLastError();
					}
					if (s->hostTimes)
					{
						memset(s->hostTimes, 0, 256 * sizeof(unsigned long long));
						unsigned long long* dev = nullptr;
						if (hipHostGetDevicePointer((void**)&dev, s->hostTimes, 0) == hipSuccess)
						{
							pv.debugTimes = dev;
						}
					}
				}
				s->persistValid = true;
			}
		}
	}

	// strips only pay through the strip kernels: when neither the persistent step nor the lean launches can take this
	// partition (too many colours, a hub body, LDS budget), fall back to the colour-batch structure for this graph
	if (strips.active && !s->optStripsAnySolver && !s->persistValid && !(s->leanAValid && s->leanBValid))
	{
		s->stripsRejected = true;
		s->structureDirty = true;
		return buildStructure(s, solverType);
	}

	// ---- message-passing tables of the global part (see MsgBodies) ----
	s->msgTablesValid = false;
	if (cs.globalCount > 0 && js.globalCount == 0 && !cs.hasTail && !needAdj)
	{
		const int G = cs.globalCount;
		std::vector<int> offsets((size_t)nb + 1, 0), list((size_t)2 * G), next((size_t)2 * G, 0), first((size_t)nb, -1);
		for (int k = 0; k < G; ++k)
		{
			offsets[(size_t)s->hContactA[cs.order[k]] + 1] += 1;
			offsets[(size_t)s->hContactB[cs.order[k]] + 1] += 1;
		}
		for (int i = 0; i < nb; ++i)
		{
			offsets[(size_t)i + 1] += offsets[i];
		}
		std::vector<int> cursor(offsets.begin(), offsets.end() - 1);
		for (int k = 0; k < G; ++k) // ascending k: every body's copies end up in sweep order
		{
			list[(size_t)cursor[s->hContactA[cs.order[k]]]++] = 2 * k;
			list[(size_t)cursor[s->hContactB[cs.order[k]]]++] = 2 * k + 1;
		}
		for (int i = 0; i < nb; ++i)
		{
			int b0 = offsets[i], b1 = offsets[(size_t)i + 1];
			if (b1 > b0)
			{
				first[i] = list[(size_t)b0];
				for (int e = b0; e < b1; ++e)
				{
					next[(size_t)list[(size_t)e]] = list[(size_t)(e + 1 < b1 ? e + 1 : b0)];
				}
			}
		}
		size_t bytes = (size_t)2 * G * (2 * sizeof(float4) + 2 * sizeof(int)) + ((size_t)2 * nb + 1) * sizeof(int) + 1024;
		grew = false;
		if ((rc = s->dMsg.ensure(bytes, &grew)) != 0)
		{
			return rc;
		}
		if (grew)
		{
			s->layoutGeneration += 1;
		}
		Carver cvr{(char*)s->dMsg.p, (char*)s->dMsg.p + s->dMsg.bytes};
		float4* dvel = cvr.take<float4>((size_t)2 * G);
		float4* ddq = cvr.take<float4>((size_t)2 * G);
		int* dnext = cvr.take<int>((size_t)2 * G);
		int* dlist = cvr.take<int>((size_t)2 * G);
		int* dfirst = cvr.take<int>((size_t)nb);
		int* doffsets = cvr.take<int>((size_t)nb + 1);
		if (cvr.p > cvr.end)
		{
			// alignment slack exceeded: grow once more
			if ((rc = s->dMsg.ensure(bytes + 8192, &grew)) != 0)
			{
				return rc;
			}
			s->layoutGeneration += 1;
			cvr = Carver{(char*)s->dMsg.p, (char*)s->dMsg.p + s->dMsg.bytes};
			dvel = cvr.take<float4>((size_t)2 * G);
			ddq = cvr.take<float4>((size_t)2 * G);
			dnext = cvr.take<int>((size_t)2 * G);
			dlist = cvr.take<int>((size_t)2 * G);
			dfirst = cvr.take<int>((size_t)nb);
			doffsets = cvr.take<int>((size_t)nb + 1);
		}
		HIP_TRY(hipMemcpyAsync(dnext, next.data(), next.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipMemcpyAsync(dlist, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipMemcpyAsync(dfirst, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipMemcpyAsync(doffsets, offsets.data(), offsets.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
		HIP_TRY(hipStreamSynchronize(s->stream));
		s->msg.vel = dvel, s->msg.dq = ddq, s->msg.next = dnext, s->msg.firstSlot = dfirst, s->msg.slotOffsets = doffsets, s->msg.slotList = dlist;
		s->msgTablesValid = true;
	}

	s->adjValid = false;
	{
		// body -> incident constraints in SWEEP order (ascending k), key = k<<1 | side, so the per-body
		// sums of jacobiApplyKernel add in exactly the order a sequential pass in sweep order would;
		// read-only shareable bodies are skipped (their deltas are exact zeros)
		std::vector<int> offsets((size_t)nb + 1, 0), list;
		const int GC = cs.globalCount; // LDS groups walk their own colours; only the global part is indexed
		for (int k = 0; k < GC; ++k)
		{
			int a = s->hContactA[cs.order[k]], b = s->hContactB[cs.order[k]];
			if (conflict[a])
			{
				offsets[(size_t)a + 1] += 1;
			}
			if (conflict[b])
			{
				offsets[(size_t)b + 1] += 1;
			}
		}
		for (int i = 0; i < nb; ++i)
		{
			offsets[(size_t)i + 1] += offsets[i];
		}
		list.resize((size_t)offsets[nb]);
		std::vector<int> cursor(offsets.begin(), offsets.end() - 1);
		for (int k = 0; k < GC; ++k)
		{
			int a = s->hContactA[cs.order[k]], b = s->hContactB[cs.order[k]];
			if (conflict[a])
			{
				list[(size_t)cursor[a]++] = (k << 1) | 0;
			}
			if (conflict[b])
			{
				list[(size_t)cursor[b]++] = (k << 1) | 1;
			}
		}
		grew = false;
		if ((rc = s->dAdjOffsets.ensure(((size_t)nb + 1) * sizeof(int), &grew)) != 0 ||
			(rc = s->dAdjList.ensure(std::max<size_t>(list.size(), 1) * sizeof(int), &grew)) != 0)
		{
			return rc;
		}
		if (grew)
		{
			s->layoutGeneration += 1;
		}
		HIP_TRY(hipMemcpyAsync(s->dAdjOffsets.p, offsets.data(), ((size_t)nb + 1) * sizeof(int), hipMemcpyHostToDevice, s->stream));
		if (!list.empty())
		{
			HIP_TRY(hipMemcpyAsync(s->dAdjList.p, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
		}
		s->adjValid = true;
	}
	// the staging vectors above die with this scope: hipMemcpyAsync from pageable host memory
	// copies through a staging buffer before it returns, so that is safe
	HIP_TRY(hipStreamSynchronize(s->stream));

	s->orderSolverClass = cls;
	s->orderGrouped = grouped;
	s->orderStrips = wantStrips;
	s->structureDirty = false;
	s->structureGeneration += 1;
	s->stats.hostPrepMs = (float)(nowMs() - t0);
	return S2AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// plans: each builder records the stage sequence of one reference s2Solve_* function
// ------------------------------------------------------------------------------------------------
struct PlanBuilder
{
	StepPlan& p;
	const StepConsts& sc;

	void op(int code, int kind = 0, float h = 0.0f, float inv_h = 0.0f, bool useBias = false, int flag = 0)
	{
		Op o;
		o.code = code, o.kind = kind, o.useBias = useBias ? 1 : 0, o.flag = flag;
		o.h = h, o.inv_h = inv_h, o.f0 = 0.0f, o.f1 = 0.0f;
		p.ops.push_back(o);
	}
	void integrateVelocities() { op(OP_INTEGRATE_VEL); }
	void integratePositions(float h) { op(OP_INTEGRATE_POS, 0, h); }
	void finalizePositions(int dynamicOnly = 0) { op(OP_FINALIZE, 0, 0.0f, 0.0f, false, dynamicOnly); }
	void jointSweep(int kind, float h, float inv_h, bool useBias) { op(OP_JOINT_SWEEP, kind, h, inv_h, useBias); }
	void warmStartContacts(int kind) { op(OP_WARM, kind); }
	void solveSoft(int kind, float inv_h, bool useBias)
	{
		op(OP_SOLVE_SOFT, kind, 0.0f, inv_h, useBias);
		p.solveSweeps += 1;
	}
	void solveRigid(int kind, float inv_h)
	{
		op(OP_SOLVE_RIGID, kind, 0.0f, inv_h);
		p.solveSweeps += 1;
	}
	void solveNGS()
	{
		op(OP_SOLVE_NGS);
		p.solveSweeps += 1;
	}
	void solveSticky(float inv_h, bool useBias)
	{
		op(OP_SOLVE_STICKY, 0, 0.0f, inv_h, useBias);
		p.solveSweeps += 1;
	}
	void prepareContacts(int kind, float h, float hertz) { p.prepContacts = kind, p.prepH = h, p.prepHertz = hertz; }
	void prepareJoints(int kind, float h, float hertz, bool warm) { p.prepJoints = kind, p.jprepH = h, p.jprepHertz = hertz, p.jprepWarm = warm ? 1 : 0; }
	void storeImpulses(int kind, float scale = 0.0f) { p.storeKind = kind, p.storeScale = scale; }

	// s2Solve_TGS_Soft (solve_tgs_soft.c:138-280) / s2Solve_SoftStep (solve_soft_step.c:182-311)
	void solveTgsSoft(bool fixedAnchors)
	{
		float h = sc.h, inv_h = sc.inv_h;
		float contactHertz = S2_MINF(S2_CONTACT_HERTZ, 0.25f * inv_h);
		float jointHertz = fixedAnchors ? S2_MINF(S2_JOINT_HERTZ, 0.25f * inv_h) : S2_MINF(S2_JOINT_HERTZ, 0.125f * inv_h);
		p.unpackH = h;
		prepareContacts(PREP_SOFT, h, contactHertz);
		prepareJoints(JPREP_SOFT, h, jointHertz, true);
		for (int substep = 0; substep < sc.iterations; ++substep)
		{
			integrateVelocities();
			if (sc.warmStart)
			{
				jointSweep(JSOLVE_WARM, h, inv_h, false);
				warmStartContacts(fixedAnchors ? WARM_FIXED : WARM_CURRENT);
			}
			jointSweep(JSOLVE_SOFT, h, inv_h, true);
			solveSoft(fixedAnchors ? SOFT_FIXED : SOFT_TGS, inv_h, true);
			integratePositions(h);
			if (sc.extraIterations > 0)
			{
				jointSweep(JSOLVE_SOFT, h, inv_h, false);
				solveSoft(fixedAnchors ? SOFT_FIXED : SOFT_TGS, inv_h, false);
			}
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_Jacobi (solve_jacobi.c:134-292) / s2Solve_PGS_Soft (solve_pgs_soft.c:127-242)
	void solveJacobiOrPgsSoft(bool jacobi)
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		float contactHertz = S2_MINF(S2_CONTACT_HERTZ, 0.333f * inv_h);
		float jointHertz = S2_MINF(S2_JOINT_HERTZ, 0.5f * inv_h);
		p.unpackH = h;
		integrateVelocities();
		prepareContacts(PREP_SOFT, h, contactHertz);
		if (sc.warmStart)
		{
			warmStartContacts(WARM_CURRENT);
		}
		// prepare reads only poses, warm start writes only velocities: "prepare all, then warm start
		// in order" is the reference's interleaved loop (solve_jacobi.c:193-206)
		prepareJoints(JPREP_SOFT, h, jointHertz, sc.warmStart != 0);
		if (sc.warmStart)
		{
			jointSweep(JSOLVE_WARM, h, inv_h, false);
		}
		for (int iter = 0; iter < sc.iterations; ++iter)
		{
			jointSweep(JSOLVE_SOFT, h, inv_h, true);
			solveSoft(jacobi ? SOFT_JACOBI : SOFT_PGS, inv_h, true);
			if (jacobi)
			{
				op(OP_JACOBI_APPLY);
			}
		}
		integratePositions(h);
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			jointSweep(JSOLVE_SOFT, h, inv_h, false);
			solveSoft(jacobi ? SOFT_JACOBI : SOFT_PGS, inv_h, false);
			if (jacobi)
			{
				op(OP_JACOBI_APPLY);
			}
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_PGS: solve_pgs.c:125-213
	void solvePgs()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		p.unpackH = h;
		integrateVelocities();
		prepareContacts(PREP_PGS, h, 0.0f);
		if (sc.warmStart)
		{
			warmStartContacts(WARM_CURRENT);
		}
		prepareJoints(JPREP_PLAIN, h, 0.0f, sc.warmStart != 0);
		if (sc.warmStart)
		{
			jointSweep(JSOLVE_WARM, h, inv_h, false);
		}
		for (int iter = 0; iter < sc.iterations; ++iter)
		{
			jointSweep(JSOLVE_BAUMGARTE, h, inv_h, true);
			solveRigid(RIGID_BAUMGARTE, inv_h);
		}
		integratePositions(h);
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_PGS_NGS: solve_pgs_ngs.c:149-255.  The reference stores the impulses before the NGS
	// sweeps (:232); the NGS sweeps never touch an impulse, so storing after them is the same.
	void solvePgsNgs()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		p.unpackH = h;
		integrateVelocities();
		prepareContacts(PREP_PGS, h, 0.0f);
		if (sc.warmStart)
		{
			warmStartContacts(WARM_CURRENT);
		}
		prepareJoints(JPREP_PLAIN, h, 0.0f, sc.warmStart != 0);
		if (sc.warmStart)
		{
			jointSweep(JSOLVE_WARM, h, inv_h, false);
		}
		for (int iter = 0; iter < sc.iterations; ++iter)
		{
			jointSweep(JSOLVE_PLAIN, h, inv_h, false);
			solveRigid(RIGID_PGS, inv_h);
		}
		integratePositions(h);
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			jointSweep(JSOLVE_POSITION, h, inv_h, false);
			solveNGS();
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_PGS_NGS_Block: solve_pgs_ngs_block.c:892-963
	void solveBlock()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		p.unpackH = h;
		integrateVelocities();
		prepareContacts(PREP_BLOCK, h, 0.0f);
		warmStartContacts(WARM_BLOCK); // always applied: solve_pgs_ngs_block.c:279-319
		prepareJoints(JPREP_PLAIN, h, 0.0f, sc.warmStart != 0);
		if (sc.warmStart)
		{
			jointSweep(JSOLVE_WARM, h, inv_h, false);
		}
		for (int iter = 0; iter < sc.iterations; ++iter)
		{
			jointSweep(JSOLVE_PLAIN, h, inv_h, false);
			op(OP_BLOCK_VEL);
			p.solveSweeps += 1;
		}
		integratePositions(h);
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			op(OP_BLOCK_POS); // contacts before joints here (:945-957)
			p.solveSweeps += 1;
			jointSweep(JSOLVE_POSITION, h, inv_h, false);
		}
		finalizePositions();
		storeImpulses(STORE_BLOCK);
	}

	// s2Solve_TGS_NGS: solve_tgs_ngs.c:207-317
	void solveTgsNgs()
	{
		float h = sc.h, inv_h = sc.inv_h;
		p.unpackH = h;
		prepareContacts(PREP_TGS, h, 0.0f);
		prepareJoints(JPREP_PLAIN, h, 0.0f, sc.warmStart != 0);
		for (int substep = 0; substep < sc.iterations; ++substep)
		{
			integrateVelocities();
			if (sc.warmStart)
			{
				jointSweep(JSOLVE_WARM, h, inv_h, false);
				warmStartContacts(WARM_CURRENT);
			}
			jointSweep(JSOLVE_PLAIN, h, inv_h, false);
			solveRigid(RIGID_TGS, inv_h);
			integratePositions(h);
			jointSweep(JSOLVE_POSITION, h, inv_h, false);
			solveNGS();
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_TGS_Sticky: solve_tgs_sticky.c:313-417
	void solveTgsSticky()
	{
		float h = sc.h, inv_h = sc.inv_h;
		p.unpackH = h;
		prepareJoints(JPREP_PLAIN, h, 0.0f, false);
		prepareContacts(PREP_STICKY, h, 0.0f);
		for (int substep = 0; substep < sc.iterations; ++substep)
		{
			integrateVelocities();
			jointSweep(JSOLVE_BAUMGARTE, h, inv_h, true);
			solveSticky(inv_h, true);
			integratePositions(h);
		}
		finalizePositions();
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			jointSweep(JSOLVE_BAUMGARTE, h, inv_h, false);
			solveSticky(inv_h, false);
		}
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_XPBD: solve_xpbd.c:342-530
	void solveXpbd()
	{
		int substepCount = sc.iterations;
		if (substepCount == 0 || sc.dt == 0.0f)
		{
			p.earlyOut = true;
			return;
		}
		float h = sc.dt / substepCount;
		float inv_h = 1.0f / h;
		p.unpackH = h;
		p.usesDq0 = true;
		prepareContacts(PREP_XPBD, h, 0.0f);
		prepareJoints(JPREP_XPBD, h, 0.0f, false);
		for (int substep = 0; substep < substepCount; ++substep)
		{
			op(OP_XPBD_INTEGRATE, 0, h);
			jointSweep(JSOLVE_XPBD, h, inv_h, false);
			op(OP_XPBD_POS, 0, h);
			op(OP_XPBD_PROJECT, 0, 0.0f, inv_h);
			op(OP_XPBD_VEL, 0, h);
			p.solveSweeps += 2;
		}
		finalizePositions(1);
		storeImpulses(STORE_SCALED, inv_h);
	}
};

void buildPlan(s2amdSolver* s, const s2amdStepParams* params)
{
	if (s->plan.valid && memcmp(&s->plan.params, params, sizeof(*params)) == 0)
	{
		return;
	}
	StepPlan& p = s->plan;
	p = StepPlan();
	p.params = *params;
	p.sc = makeConsts(params);
	PlanBuilder b{p, p.sc};
	switch (params->solverType)
	{
		case s2amd_solverJacobi:
			b.solveJacobiOrPgsSoft(true);
			break;
		case s2amd_solverPGS:
			b.solvePgs();
			break;
		case s2amd_solverPGS_NGS:
			b.solvePgsNgs();
			break;
		case s2amd_solverPGS_NGS_Block:
			b.solveBlock();
			break;
		case s2amd_solverPGS_Soft:
			b.solveJacobiOrPgsSoft(false);
			break;
		case s2amd_solverSoftStep:
			b.solveTgsSoft(true);
			break;
		case s2amd_solverTGS_Sticky:
			b.solveTgsSticky();
			break;
		case s2amd_solverTGS_Soft:
			b.solveTgsSoft(false);
			break;
		case s2amd_solverTGS_NGS:
			b.solveTgsNgs();
			break;
		case s2amd_solverXPBD:
			b.solveXpbd();
			break;
	}
	p.valid = true;
	s->planGeneration += 1;
}

// Message passing applies when the global part is contact-only without a sequential tail (tables
// valid) and the plan consists of velocity-level contact sweeps only (poses change in body kernels).
bool messageEligible(const s2amdSolver* s, int solverType)
{
	if (!s->optMessage || !s->msgTablesValid)
	{
		return false;
	}
	return solverType == s2amd_solverTGS_Soft || solverType == s2amd_solverSoftStep || solverType == s2amd_solverPGS ||
		   solverType == s2amd_solverPGS_Soft || solverType == s2amd_solverTGS_Sticky;
}

// ------------------------------------------------------------------------------------------------
// execution of a plan
// ------------------------------------------------------------------------------------------------
struct Executor
{
	s2amdSolver* s;
	hipStream_t st;
	const StepPlan& p;
	int posSolver;
	bool profile;
	bool msg = false; // message-passing accessor for the global part
	bool fork = false; // graph capture: independent kernels go to side streams (parallel graph branches)
	const int* gatherIndex = nullptr; // manifold.constraintIndex rides in the unpack launch

	hipStream_t branch(int i, int forkEvent)
	{
		if (!fork)
		{
			return st;
		}
		(void)hipEventRecord(s->evFork[forkEvent], st);
		(void)hipStreamWaitEvent(s->side[i], s->evFork[forkEvent], 0);
		return s->side[i];
	}
	void join(int i, int joinEvent)
	{
		if (fork)
		{
			(void)hipEventRecord(s->evJoin[joinEvent], s->side[i]);
			(void)hipStreamWaitEvent(st, s->evJoin[joinEvent], 0);
		}
	}

	s2amdContact* wireContacts() const { return (s2amdContact*)s->dContacts.p; }
	s2amdBody* wireBodies() const { return (s2amdBody*)s->dBodies.p; }
	s2amdJoint* wireJoints() const { return (s2amdJoint*)s->dJoints.p; }
	const Op* deviceOps() const { return (const Op*)s->dOps.p; }

	void count(int n = 1) { s->launchCounter += n; }

	void recordEvent()
	{
		if (s->sweepEventsUsed == s->sweepEvents.size())
		{
			hipEvent_t e;
			if (hipEventCreate(&e) != hipSuccess)
			{
				return;
			}
			s->sweepEvents.push_back(e);
		}
		(void)hipEventRecord(s->sweepEvents[s->sweepEventsUsed++], st);
	}

	static bool isSolveSweep(int code)
	{
		return code == OP_SOLVE_SOFT || code == OP_SOLVE_RIGID || code == OP_SOLVE_STICKY || code == OP_SOLVE_NGS || code == OP_XPBD_POS ||
			   code == OP_XPBD_VEL || code == OP_BLOCK_VEL || code == OP_BLOCK_POS;
	}

	void launchContactBatch(const Op& o, int b, int e)
	{
		if (msg)
		{
			switch (o.code)
			{
				case OP_WARM:
					launchWarmStartContactsMsg(st, o.kind, s->cv, s->msg, b, e);
					return;
				case OP_SOLVE_SOFT:
					launchSolveContactsSoftMsg(st, o.kind, s->cv, s->msg, b, e, o.inv_h, o.useBias);
					return;
				case OP_SOLVE_RIGID:
					launchSolveContactsRigidMsg(st, o.kind, s->cv, s->msg, b, e, o.inv_h);
					return;
				case OP_SOLVE_STICKY:
					launchSolveContactsStickyMsg(st, s->cv, s->msg, wireContacts(), b, e, o.inv_h, o.useBias);
					return;
				default:
					return; // message mode is only enabled for plans made of the ops above
			}
		}
		switch (o.code)
		{
			case OP_WARM:
				launchWarmStartContacts(st, o.kind, s->cv, s->bv, b, e);
				break;
			case OP_SOLVE_SOFT:
				launchSolveContactsSoft(st, o.kind, s->cv, s->bv, b, e, o.inv_h, o.useBias);
				break;
			case OP_SOLVE_RIGID:
				launchSolveContactsRigid(st, o.kind, s->cv, s->bv, b, e, o.inv_h);
				break;
			case OP_SOLVE_STICKY:
				launchSolveContactsSticky(st, s->cv, s->bv, wireContacts(), b, e, o.inv_h, o.useBias);
				break;
			case OP_SOLVE_NGS:
				launchSolveContactsNGS(st, s->cv, s->bv, b, e);
				break;
			case OP_XPBD_POS:
				launchXpbdContactPositions(st, s->cv, s->bv, b, e, o.h);
				break;
			case OP_XPBD_VEL:
				launchXpbdContactVelocities(st, s->cv, s->bv, b, e, o.h);
				break;
			case OP_BLOCK_VEL:
				launchBlockSolveVelocity(st, s->cv, s->bv, b, e);
				break;
			case OP_BLOCK_POS:
				launchBlockSolvePosition(st, s->cv, s->bv, b, e);
				break;
		}
	}

	// one op of the plan over the GLOBAL part (bodies in HBM): one launch per colour batch
	void runGlobalOp(int index)
	{
		const Op& o = p.ops[(size_t)index];
		const SweepSet& cs = s->contacts;
		const SweepSet& js = s->joints;
		const bool bodies = s->looseBodies > 0;
		switch (o.code)
		{
			case OP_INTEGRATE_VEL:
				if (bodies)
				{
					if (msg)
					{
						launchIntegrateVelocitiesMsg(st, s->bv, s->msg);
					}
					else
					{
						launchIntegrateVelocities(st, s->bv);
					}
					count();
				}
				return;
			case OP_INTEGRATE_POS:
				if (bodies)
				{
					if (msg)
					{
						launchIntegratePositionsMsg(st, s->bv, s->msg, o.h);
					}
					else
					{
						launchIntegratePositions(st, s->bv, o.h);
					}
					count();
				}
				return;
			case OP_FINALIZE:
				if (bodies)
				{
					if (msg)
					{
						launchFinalizePositionsMsg(st, s->bv, s->msg, o.flag);
					}
					else
					{
						launchFinalizePositions(st, s->bv, o.flag);
					}
					count();
				}
				return;
			case OP_XPBD_INTEGRATE:
				if (bodies)
				{
					launchXpbdIntegrate(st, s->bv, o.h);
					count();
				}
				return;
			case OP_XPBD_PROJECT:
				if (bodies)
				{
					launchXpbdProject(st, s->bv, o.inv_h);
					count();
				}
				return;
			case OP_JACOBI_APPLY:
				launchJacobiApply(st, s->bv, s->cv, (const int*)s->dAdjOffsets.p, (const int*)s->dAdjList.p);
				count();
				return;
			case OP_JOINT_SWEEP:
			{
				int nb = (int)js.batchOffsets.size() - 1;
				for (int bi = 0; bi < nb; ++bi)
				{
					int b = js.batchOffsets[(size_t)bi], e = js.batchOffsets[(size_t)bi + 1];
					if (e <= b)
					{
						continue;
					}
					if (js.hasTail && bi == nb - 1)
					{
						launchGroupKernel(st, s->cv, s->jv, s->bv, s->dJointTail.view, deviceOps() + index, 1, p.sc, wireContacts(),
										  s->dJointTail.maxBodies, 0);
					}
					else
					{
						launchSolveJoints(st, o.kind, s->jv, s->bv, b, e, p.sc, o.h, o.inv_h, o.useBias);
					}
					count();
				}
				return;
			}
			default:
				break;
		}
		// contact sweeps
		if (o.code == OP_SOLVE_SOFT && o.kind == SOFT_JACOBI)
		{
			// the Jacobi pass writes per-constraint deltas, never a body: one launch for all colours
			if (cs.globalCount > 0)
			{
				if (profile)
				{
					recordEvent();
				}
				launchSolveContactsSoft(st, o.kind, s->cv, s->bv, 0, cs.globalCount, o.inv_h, o.useBias);
				if (profile)
				{
					recordEvent();
				}
				count();
			}
			return;
		}
		int nb = (int)cs.batchOffsets.size() - 1;
		for (int bi = 0; bi < nb; ++bi)
		{
			int b = cs.batchOffsets[(size_t)bi], e = cs.batchOffsets[(size_t)bi + 1];
			if (e <= b)
			{
				continue;
			}
			const bool timed = profile && isSolveSweep(o.code);
			if (timed)
			{
				recordEvent();
			}
			if (cs.hasTail && bi == nb - 1)
			{
				launchGroupKernel(st, s->cv, s->jv, s->bv, s->dContactTail.view, deviceOps() + index, 1, p.sc, wireContacts(),
								  s->dContactTail.maxBodies, 0);
			}
			else
			{
				launchContactBatch(o, b, e);
			}
			if (timed)
			{
				recordEvent();
			}
			count();
		}
	}

	static bool isBodyOp(int code)
	{
		return code == OP_INTEGRATE_VEL || code == OP_INTEGRATE_POS || code == OP_FINALIZE || code == OP_XPBD_INTEGRATE || code == OP_XPBD_PROJECT;
	}

	void launchStripGroups(const DeviceGroupTable& t, int first, int n, bool timed)
	{
		if (timed)
		{
			recordEvent();
		}
		launchStripKernel(st, s->cv, s->jv, s->bv, t.view, deviceOps() + first, n, p.sc, wireContacts(), t.maxBodies, p.usesDq0 ? 1 : 0);
		if (timed)
		{
			recordEvent();
		}
		count();
	}

	static bool leanSoftKind(const Op& o) { return o.code == OP_SOLVE_SOFT && (o.kind == SOFT_TGS || o.kind == SOFT_PGS || o.kind == SOFT_FIXED); }

	bool sweepsNothing(const Op& o) const
	{
		if (isBodyOp(o.code))
		{
			return false;
		}
		return (o.code == OP_JOINT_SWEEP ? s->joints.stripCount : s->contacts.stripCount) == 0;
	}

	// Can ops [first, sweep) ride in front of the soft sweep `sweep` inside ONE lean strip launch?  Allowed, in
	// this order: integrate positions, integrate velocities, contact warm start (body-centric).
	bool leanSegment(int first, int sweep, StripOps& out, int& warm) const
	{
		if (!s->leanAValid || !leanSoftKind(p.ops[(size_t)sweep]))
		{
			return false;
		}
		out = StripOps{};
		warm = -1;
		int stage = 0;
		for (int i = first; i < sweep; ++i)
		{
			const Op& o = p.ops[(size_t)i];
			if (o.code == OP_INTEGRATE_POS && stage < 1)
			{
				out.integratePos = 1, out.posH = o.h, stage = 1;
			}
			else if (o.code == OP_INTEGRATE_VEL && stage < 2)
			{
				out.integrateVel = 1, stage = 2;
			}
			else if (o.code == OP_WARM && stage < 3 && s->optBodyWarm && (o.kind == WARM_CURRENT || o.kind == WARM_FIXED))
			{
				warm = o.kind, stage = 3;
			}
			else if (!sweepsNothing(o))
			{
				return false;
			}
		}
		const Op& w = p.ops[(size_t)sweep];
		out.sweep = 1, out.useBias = w.useBias, out.inv_h = w.inv_h;
		return true;
	}

	// Can the whole plan run as ONE persistent launch over the strips (strip_kernel.hip: stripStepKernel)?
	bool persistPlan(int& kind, int& warm) const
	{
		if (!s->persistValid || s->persistFailed || p.ops.size() > 128 || p.solveSweeps > 63) // one hand-off epoch per sweep, 64 per step
		{
			return false;
		}
		kind = -1, warm = -1;
		for (const Op& o : p.ops)
		{
			if (o.code == OP_INTEGRATE_VEL || o.code == OP_INTEGRATE_POS || o.code == OP_FINALIZE)
			{
				continue;
			}
			if (o.code == OP_JOINT_SWEEP && s->joints.stripCount == 0)
			{
				continue;
			}
			if (o.code == OP_WARM && (o.kind == WARM_CURRENT || o.kind == WARM_FIXED) && (warm < 0 || warm == o.kind))
			{
				warm = o.kind;
				continue;
			}
			if (leanSoftKind(o) && (kind < 0 || kind == o.kind))
			{
				kind = o.kind;
				continue;
			}
			return false;
		}
		if (kind < 0)
		{
			return false;
		}
		if (warm < 0)
		{
			warm = kind == SOFT_FIXED ? WARM_FIXED : WARM_CURRENT;
		}
		const bool narrow = kind == SOFT_TGS && warm == WARM_CURRENT;
		const int records = (narrow ? s->persist.ldsRecords : s->persistRecordsWide) + 2 * (int)p.ops.size();
		return records <= (160 * 1024) / 16;
	}

	// the plan without the sweeps that have nothing to sweep in the strips (joint sweeps of a contact-only island)
	int uploadPersistOps()
	{
		if (s->persistOpsGeneration == s->planGeneration && s->persistOpsStructure == s->structureGeneration)
		{
			return 0;
		}
		std::vector<Op> kept;
		for (const Op& o : p.ops)
		{
			if (!sweepsNothing(o))
			{
				kept.push_back(o);
			}
		}
		bool grew = false;
		int rc = s->dPersistOps.ensure(std::max<size_t>(kept.size(), 1) * sizeof(Op), &grew);
		if (rc)
		{
			return rc;
		}
		if (grew)
		{
			s->layoutGeneration += 1;
		}
		if (hipMemcpyAsync(s->dPersistOps.p, kept.data(), kept.size() * sizeof(Op), hipMemcpyHostToDevice, st) != hipSuccess ||
			hipStreamSynchronize(st) != hipSuccess)
		{
			return S2AMD_E_DEVICE;
		}
		s->persistOpCount = (int)kept.size();
		s->persistOpsGeneration = s->planGeneration;
		s->persistOpsStructure = s->structureGeneration;
		return 0;
	}

	void clearGranules(hipStream_t where)
	{
		(void)hipMemsetAsync(s->dGranules.p, 0, s->granuleBytes, where); // epochs restart at 1 every launch
		count();
	}

	void runPersistent(int kind, int warm, bool clearFirst = false)
	{
		// hand-off tags are the exchange number; the step's epilogue kernel leaves the buffers zeroed for the next
		// step, so they only need clearing when this launch is replayed on its own (s2amd_measure_dominant)
		if (clearFirst)
		{
			clearGranules(st);
		}
		if (profile)
		{
			recordEvent();
		}
		PersistView pv = s->persist;
		if (!(kind == SOFT_TGS && warm == WARM_CURRENT))
		{
			pv.ldsRecords = s->persistRecordsWide;
		}
		launchStripStep(st, kind, warm, s->cv, s->bv, s->leanA, pv, (const Op*)s->dPersistOps.p, s->persistOpCount);
		if (profile)
		{
			recordEvent();
		}
		count();
	}

	// the plan over the strips: body ops ride with the next sweep's phase A launch; every sweep is
	// phase A (interiors, all strips) then phase B (seams)
	void runStrips()
	{
		int kind, warm;
		if (persistPlan(kind, warm))
		{
			runPersistent(kind, warm);
			return;
		}
		const int n = (int)p.ops.size();
		int segStart = 0;
		for (int i = 0; i < n; ++i)
		{
			const Op& o = p.ops[(size_t)i];
			if (isBodyOp(o.code) || sweepsNothing(o))
			{
				continue; // a body op rides along; a sweep over nothing is a no-op wherever it lands
			}
			StripOps lean;
			int warm = -1;
			if (o.code == OP_WARM)
			{
				// folded into the lean launch of the next sweep when that launch can take it
				int j = i + 1;
				while (j < n && (isBodyOp(p.ops[(size_t)j].code) || sweepsNothing(p.ops[(size_t)j])))
				{
					j += 1;
				}
				if (j < n && leanSegment(segStart, j, lean, warm) && warm >= 0)
				{
					continue;
				}
			}
			const bool joint = o.code == OP_JOINT_SWEEP;
			const bool timed = profile && isSolveSweep(o.code);
			const bool seam = (joint ? s->joints.seamCount : s->contacts.seamCount) > 0;
			if (leanSegment(segStart, i, lean, warm))
			{
				if (timed)
				{
					recordEvent();
				}
				launchStripSoft(st, o.kind, warm, s->cv, s->bv, s->leanA, lean);
				if (timed)
				{
					recordEvent();
				}
				count();
			}
			else
			{
				launchStripGroups(s->dStripA, segStart, i + 1 - segStart, timed);
			}
			if (seam)
			{
				if (s->leanBValid && leanSoftKind(o))
				{
					StripOps only{};
					only.sweep = 1, only.useBias = o.useBias, only.inv_h = o.inv_h;
					if (timed)
					{
						recordEvent();
					}
					launchStripSoft(st, o.kind, -1, s->cv, s->bv, s->leanB, only);
					if (timed)
					{
						recordEvent();
					}
					count();
				}
				else
				{
					launchStripGroups(s->dStripB, i, 1, timed);
				}
			}
			segStart = i + 1;
		}
		if (segStart < n)
		{
			launchStripGroups(s->dStripA, segStart, n - segStart, false);
		}
	}

	void run()
	{
		if (p.earlyOut)
		{
			return;
		}
		// pre: wire -> SoA.  With contacts to prepare, ONE launch does the three independent prologue jobs (prepare
		// contacts, unpack bodies, manifold.constraintIndex); otherwise the unpack launch carries the index.
		const bool prepares = p.prepContacts >= 0 && s->cv.count > 0;
		if (prepares)
		{
			launchPrepareContacts(st, p.prepContacts, s->cv, s->bv, wireContacts(), wireBodies(), p.sc, p.prepH, p.prepHertz, posSolver,
								  (const uint32_t*)s->dBodyFlags.p, true, p.unpackH, s->contactCapacity, gatherIndex);
			count();
		}
		else
		{
			launchUnpackBodies(st, s->bv, wireBodies(), (const uint32_t*)s->dBodyFlags.p, p.sc, p.unpackH, wireContacts(), s->contactCapacity, gatherIndex);
			count();
		}
		if (p.prepJoints >= 0 && s->jv.count > 0)
		{
			launchPrepareJoints(st, p.prepJoints, s->jv, s->bv, wireJoints(), wireBodies(), p.sc, p.jprepH, p.jprepHertz, p.jprepWarm, posSolver);
			count();
		}
		if (msg)
		{
			launchFillMessageSlots(st, s->cv, s->bv, s->msg, s->contacts.globalCount);
			count();
		}
		// LDS groups: the whole op list in one launch
		if (s->dGroups.view.groupCount > 0)
		{
			launchGroupKernel(st, s->cv, s->jv, s->bv, s->dGroups.view, deviceOps(), (int)p.ops.size(), p.sc, wireContacts(), s->dGroups.maxBodies,
							  p.usesDq0 ? 1 : 0);
			count();
		}
		if (s->dStripA.view.groupCount > 0)
		{
			runStrips();
		}
		// global part: op by op
		const bool anyGlobal = s->looseBodies > 0 || s->contacts.globalCount > 0 || s->joints.globalCount > 0;
		if (anyGlobal)
		{
			const int n = (int)p.ops.size();
			std::vector<uint8_t> done((size_t)n, 0);
			for (int i = 0; i < n; ++i)
			{
				if (done[(size_t)i])
				{
					continue;
				}
				const Op& o = p.ops[(size_t)i];
				// contact warm start as ONE body-centric launch; an immediately preceding integrate-velocities
				// (joint sweeps in between only when there are no global joints) rides along in the same kernel
				if (!msg && s->optBodyWarm && s->contacts.globalCount > 0 && (o.code == OP_WARM || o.code == OP_INTEGRATE_VEL))
				{
					int w = i;
					if (o.code == OP_INTEGRATE_VEL)
					{
						w = i + 1;
						while (w < n && p.ops[(size_t)w].code == OP_JOINT_SWEEP && s->joints.globalCount == 0)
						{
							w += 1;
						}
					}
					if (w < n && p.ops[(size_t)w].code == OP_WARM)
					{
						launchWarmStartBodies(st, p.ops[(size_t)w].kind, s->cv, s->bv, (const int*)s->dAdjOffsets.p, (const int*)s->dAdjList.p,
											  o.code == OP_INTEGRATE_VEL ? 1 : 0);
						count();
						for (int d = i; d <= w; ++d)
						{
							done[(size_t)d] = 1;
						}
						continue;
					}
				}
				runGlobalOp(i);
			}
		}
		if (msg)
		{
			launchGatherMessageSlots(st, s->bv, s->msg);
			count();
		}
		// post: SoA -> wire: impulses and bodies in one launch (+ the epoch base of the hand-off tags)
		{
			int kind, warm;
			const bool usedGranules = s->dStripA.view.groupCount > 0 && persistPlan(kind, warm);
			launchStoreImpulses(st, p.storeKind, s->cv, wireContacts(), p.storeScale, s->bv, wireBodies(), usedGranules ? s->dGranules.p : nullptr,
								usedGranules ? s->granuleBytes : 0, usedGranules ? s->persist.deviceError : nullptr);
		}
		count();
		if (s->jv.count > 0)
		{
			launchStoreJoints(st, s->jv, wireJoints());
			count();
		}
	}
};

} // namespace

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
namespace
{

void destroyGraph(s2amdSolver* s)
{
	if (s->graphExec)
	{
		(void)hipGraphExecDestroy(s->graphExec);
		s->graphExec = nullptr;
	}
	if (s->graph)
	{
		(void)hipGraphDestroy(s->graph);
		s->graph = nullptr;
	}
	s->graphKey = 0;
}

int refreshShadows(s2amdSolver* s, const s2amdBody* bodies, int nb, const s2amdContact* contacts, int nc, const s2amdJoint* joints, int nj)
{
	bool changed = s->structureDirty || nb != (int)s->hBodyFlags.size() || nc != (int)s->hContactA.size() || nj != (int)s->hJointType.size();
	std::vector<uint32_t> flags((size_t)nb);
	s->hBodyLive.assign((size_t)nb, 0);
	s->hBodyStatic.assign((size_t)nb, 0);
	for (int i = 0; i < nb; ++i)
	{
		const s2amdBody& b = bodies[i];
		uint32_t f = 0;
		s->hBodyLive[i] = b.type != S2AMD_BODY_FREE;
		s->hBodyStatic[i] = b.type == S2AMD_BODY_STATIC;
		if (b.type != S2AMD_BODY_FREE)
		{
			bool massless = b.invMass == 0.0f && b.invI == 0.0f;
			if (!massless)
			{
				f |= S2F_WRITE_VEL;
			}
			// position sweeps store rot = normalize(rot) even for immovable bodies
			// (solve_common.c:383-392): only a static body whose rot is a fixed point of the
			// normalisation can be treated as read-only there
			if (!(massless && b.type == S2AMD_BODY_STATIC && rotIsFixedPoint(b.rot[0], b.rot[1])))
			{
				f |= S2F_WRITE_POS;
			}
		}
		flags[i] = f;
	}
	if (!changed && flags != s->hBodyFlags)
	{
		changed = true;
	}
	s->hBodyFlags.swap(flags);

	if ((int)s->hContactA.size() != nc)
	{
		s->hContactA.assign(nc, -1);
		s->hContactB.assign(nc, -1);
		s->hContactPoints.assign(nc, 0);
	}
	for (int i = 0; i < nc; ++i)
	{
		const s2amdContact& c = contacts[i];
		int pc = c.pointCount > 0 ? c.pointCount : 0;
		if (!changed && (s->hContactA[i] != c.bodyA || s->hContactB[i] != c.bodyB || (s->hContactPoints[i] > 0) != (pc > 0)))
		{
			changed = true;
		}
		s->hContactA[i] = c.bodyA;
		s->hContactB[i] = c.bodyB;
		s->hContactPoints[i] = pc;
		if (pc > 0 && (c.bodyA < 0 || c.bodyA >= nb || c.bodyB < 0 || c.bodyB >= nb || pc > 2))
		{
			return fail(S2AMD_E_INVALID, "contact " + std::to_string(i) + " has an invalid body index or point count");
		}
	}
	if ((int)s->hJointType.size() != nj)
	{
		s->hJointType.assign(nj, S2AMD_JOINT_FREE);
		s->hJointA.assign(nj, -1);
		s->hJointB.assign(nj, -1);
	}
	for (int i = 0; i < nj; ++i)
	{
		const s2amdJoint& j = joints[i];
		if (!changed && (s->hJointType[i] != j.type || s->hJointA[i] != j.bodyA || s->hJointB[i] != j.bodyB))
		{
			changed = true;
		}
		s->hJointType[i] = j.type;
		s->hJointA[i] = j.bodyA;
		s->hJointB[i] = j.bodyB;
		if (j.type != S2AMD_JOINT_FREE)
		{
			if (j.type != S2AMD_JOINT_REVOLUTE && j.type != S2AMD_JOINT_MOUSE)
			{
				return fail(S2AMD_E_INVALID, "joint " + std::to_string(i) + " has an unknown type");
			}
			if (j.bodyB < 0 || j.bodyB >= nb || (j.type == S2AMD_JOINT_REVOLUTE && (j.bodyA < 0 || j.bodyA >= nb)))
			{
				return fail(S2AMD_E_INVALID, "joint " + std::to_string(i) + " has an invalid body index");
			}
		}
	}
	if (changed)
	{
		s->graphAge = 0; // strips wait until the graph has stayed the same for optStripPatience steps
		s->stripsRejected = false;
		s->structureDirty = true;
	}
	return S2AMD_OK;
}

int doUpload(s2amdSolver* s, const s2amdBody* bodies, int nb, const s2amdContact* contacts, int nc, const s2amdJoint* joints, int nj)
{
	if (nb < 0 || nc < 0 || nj < 0 || (nb > 0 && !bodies) || (nc > 0 && !contacts) || (nj > 0 && !joints))
	{
		return fail(S2AMD_E_INVALID, "null array with non-zero count");
	}
	HIP_TRY(hipSetDevice(s->device));
	int rc = refreshShadows(s, bodies, nb, contacts, nc, joints, nj);
	if (rc)
	{
		return rc;
	}
	bool grew = false;
	if ((rc = s->dBodies.ensure((size_t)std::max(nb, 1) * sizeof(s2amdBody), &grew)) != 0)
	{
		return rc;
	}
	if ((rc = s->dContacts.ensure((size_t)std::max(nc, 1) * sizeof(s2amdContact), &grew)) != 0)
	{
		return rc;
	}
	if ((rc = s->dJoints.ensure((size_t)std::max(nj, 1) * sizeof(s2amdJoint), &grew)) != 0)
	{
		return rc;
	}
	if ((rc = s->dBodyFlags.ensure((size_t)std::max(nb, 1) * sizeof(uint32_t), &grew)) != 0)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
		s->savedValid = false;
		s->structureDirty = true; // dBodyFlags may have moved
	}
	s->bodyCapacity = nb;
	s->contactCapacity = nc;
	s->jointCapacity = nj;
	if ((rc = carveBodies(s, nb)) != 0)
	{
		return rc;
	}
	if (nb > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dBodies.p, bodies, (size_t)nb * sizeof(s2amdBody), hipMemcpyHostToDevice, s->stream));
		// dBodyFlags is written by buildStructure (it adds the LDS-group ownership bits)
	}
	if (nc > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dContacts.p, contacts, (size_t)nc * sizeof(s2amdContact), hipMemcpyHostToDevice, s->stream));
	}
	if (nj > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dJoints.p, joints, (size_t)nj * sizeof(s2amdJoint), hipMemcpyHostToDevice, s->stream));
	}
	s->resident = true;
	return S2AMD_OK;
}

__global__ void writeConstraintIndexKernel(s2amdContact* wire, int n, const int* gatherIndex)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		wire[i].constraintIndex = gatherIndex[i];
	}
}

int doStep(s2amdSolver* s, const s2amdStepParams* params)
{
	if (!params)
	{
		return fail(S2AMD_E_INVALID, "null params");
	}
	if (params->solverType < 0 || params->solverType >= s2amd_solverTypeCount)
	{
		return fail(S2AMD_E_INVALID, "unknown solver type " + std::to_string(params->solverType));
	}
	if (!s->resident)
	{
		return fail(S2AMD_E_STATE, "s2amd_step_resident called before s2amd_upload");
	}
	HIP_TRY(hipSetDevice(s->device));
	s->stats = s2amdStepStats{};
	buildPlan(s, params);
	int rc = buildStructure(s, params->solverType);
	if (rc)
	{
		return rc;
	}
	const StepPlan& plan = s->plan;
	if (s->opsGeneration != s->planGeneration)
	{
		bool grew = false;
		if ((rc = s->dOps.ensure(std::max<size_t>(plan.ops.size(), 1) * sizeof(Op), &grew)) != 0)
		{
			return rc;
		}
		if (grew)
		{
			s->layoutGeneration += 1;
		}
		if (!plan.ops.empty())
		{
			HIP_TRY(hipMemcpyAsync(s->dOps.p, plan.ops.data(), plan.ops.size() * sizeof(Op), hipMemcpyHostToDevice, s->stream));
			HIP_TRY(hipStreamSynchronize(s->stream));
		}
		s->opsGeneration = s->planGeneration;
	}

	Executor q{s, s->stream, plan, isPositionSolver(params->solverType) ? 1 : 0, s->optProfile != 0};
	q.msg = messageEligible(s, params->solverType);
	s->stats.messagePassing = q.msg ? 1 : 0;
	{
		int kind, warm;
		if (s->dStripA.view.groupCount > 0 && q.persistPlan(kind, warm) && q.uploadPersistOps() != 0)
		{
			return fail(S2AMD_E_DEVICE, "could not upload the persistent step plan");
		}
	}
	s->launchCounter = 0;
	s->sweepEventsUsed = 0;

	const bool xpbdEarlyOut = plan.earlyOut;
	const bool writesConstraintIndex = !xpbdEarlyOut && params->solverType != s2amd_solverPGS_NGS_Block;

	// manifold.constraintIndex (pool-order gather index, -1 for skipped slots)
	if (writesConstraintIndex && s->contactCapacity > 0)
	{
		if (s->gatherIndexDirty || s->dGatherIndex.bytes < (size_t)s->contactCapacity * sizeof(int))
		{
			std::vector<int> gi((size_t)s->contactCapacity, -1);
			int k = 0;
			for (int i = 0; i < s->contactCapacity; ++i)
			{
				if (s->hContactPoints[i] > 0)
				{
					gi[i] = k++;
				}
			}
			bool grew = false;
			if ((rc = s->dGatherIndex.ensure(gi.size() * sizeof(int), &grew)) != 0)
			{
				return rc;
			}
			if (grew)
			{
				s->layoutGeneration += 1;
			}
			HIP_TRY(hipMemcpyAsync(s->dGatherIndex.p, gi.data(), gi.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
			HIP_TRY(hipStreamSynchronize(s->stream));
			s->gatherIndexDirty = false;
		}
	}

	auto enqueueAll = [&]() {
		bool indexBranch = false;
		q.gatherIndex = (writesConstraintIndex && s->contactCapacity > 0 && !q.fork) ? (const int*)s->dGatherIndex.p : nullptr;
		if (writesConstraintIndex && s->contactCapacity > 0 && q.fork)
		{
			// touches only manifold.constraintIndex, which no solver kernel reads: a parallel branch that joins at the end
			int n = s->contactCapacity;
			hipStream_t where = q.branch(0, 0);
			writeConstraintIndexKernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, where>>>((s2amdContact*)s->dContacts.p, n,
																									 (const int*)s->dGatherIndex.p);
			q.count();
			indexBranch = true;
		}
		q.run();
		if (indexBranch)
		{
			q.join(0, 0);
		}
	};

	bool useGraph = s->optGraph != 0 && !q.profile;
	q.fork = useGraph && s->optFork != 0;
	HIP_TRY(hipEventRecord(s->evBegin, s->stream));
	if (useGraph)
	{
		uint64_t key = 1469598103934665603ull;
		key = fnv(key, params, sizeof(*params));
		uint64_t gens[4] = {s->layoutGeneration, s->structureGeneration, s->planGeneration, (uint64_t)((q.msg ? 1 : 0) | (s->optBodyWarm ? 2 : 0) | (s->optStripLean ? 4 : 0) | (s->optPersist ? 8 : 0) | (s->optFork ? 16 : 0) | (s->persistFailed ? 32 : 0))};
		key = fnv(key, gens, sizeof(gens));
		int sizes[3] = {s->bodyCapacity, s->contactCapacity, s->jointCapacity};
		key = fnv(key, sizes, sizeof(sizes));
		if (key == 0)
		{
			key = 1;
		}
		if (key != s->graphKey || s->graphExec == nullptr)
		{
			destroyGraph(s);
			HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
			enqueueAll();
			hipError_t ce = hipStreamEndCapture(s->stream, &s->graph);
			if (ce != hipSuccess)
			{
				s->graph = nullptr;
				return fail(S2AMD_E_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
			}
			HIP_TRY(hipGraphInstantiate(&s->graphExec, s->graph, nullptr, nullptr, 0));
			s->graphKey = key;
			s->graphLaunches = s->launchCounter;
		}
		else
		{
			s->launchCounter = s->graphLaunches;
			s->stats.graphReplayed = 1;
		}
		HIP_TRY(hipGraphLaunch(s->graphExec, s->stream));
	}
	else
	{
		enqueueAll();
	}
	HIP_TRY(hipEventRecord(s->evEnd, s->stream));
	HIP_TRY(hipGetLastError());
	const bool async = s->optAsync != 0 && !q.profile;
	if (!async)
	{
		HIP_TRY(hipStreamSynchronize(s->stream));
		float ms = 0.0f;
		HIP_TRY(hipEventElapsedTime(&ms, s->evBegin, s->evEnd));
		s->stats.deviceMs = ms;
	}
	s->stats.constraintCount = s->cv.count;
	s->stats.jointCount = s->jv.count;
	s->stats.contactColors = (int)s->contacts.colorOffsets.size() - 1;
	s->stats.jointColors = (int)s->joints.colorOffsets.size() - 1;
	s->stats.solveSweeps = plan.solveSweeps;
	s->stats.kernelLaunches = s->launchCounter;
	s->stats.groupCount = s->dGroups.view.groupCount;
	s->stats.stripCount = s->dStripA.view.groupCount;
	s->stats.seamCount = s->dStripB.view.groupCount;
	{
		int kind, warm;
		s->stats.persistent = (s->dStripA.view.groupCount > 0 && q.persistPlan(kind, warm)) ? 1 : 0;
	}
	s->stats.persistFallbacks = s->persistFallbacks;
	if (!async && s->hostError && *s->hostError != 0u)
	{
		// The persistent kernel's workgroups were not all resident (something else occupies the GPU).  Its epilogue saw
		// the flag and left the wire arrays untouched, so the step is simply repeated on the multi-launch strip path,
		// which this solver keeps from now on.
		*s->hostError = 0u;
		(void)hipMemsetAsync(s->persist.deviceError, 0, sizeof(unsigned int), s->stream);
		s->persistFailed = true;
		s->persistFallbacks += 1;
		return doStep(s, params);
	}
	s->graphAge += 1;
	if (q.profile)
	{
		float total = 0.0f;
		for (size_t i = 0; i + 1 < s->sweepEventsUsed; i += 2)
		{
			float t = 0.0f;
			if (hipEventElapsedTime(&t, s->sweepEvents[i], s->sweepEvents[i + 1]) == hipSuccess)
			{
				total += t;
			}
		}
		// calibrate: an empty event pair on the same stream measures the bracket's own cost
		float empty = 0.0f;
		int pairs = 0;
		if (s->sweepEvents.size() >= 2)
		{
			for (int r = 0; r < 32; ++r)
			{
				(void)hipEventRecord(s->sweepEvents[0], s->stream);
				(void)hipEventRecord(s->sweepEvents[1], s->stream);
				(void)hipStreamSynchronize(s->stream);
				float t = 0.0f;
				if (hipEventElapsedTime(&t, s->sweepEvents[0], s->sweepEvents[1]) == hipSuccess)
				{
					empty += t;
					pairs += 1;
				}
			}
		}
		s->stats.solveKernelMs = total;
		s->stats.solveLaunches = (int)(s->sweepEventsUsed / 2);
		s->stats.eventPairOverheadMs = pairs > 0 ? empty / pairs : 0.0f;
	}
	return S2AMD_OK;
}

int doDownload(s2amdSolver* s, s2amdBody* bodies, int nb, s2amdContact* contacts, int nc, s2amdJoint* joints, int nj)
{
	if (!s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident to download");
	}
	if (nb < s->bodyCapacity || nc < s->contactCapacity || nj < s->jointCapacity)
	{
		return fail(S2AMD_E_CAPACITY, "output arrays smaller than the resident world");
	}
	HIP_TRY(hipSetDevice(s->device));
	if (s->bodyCapacity > 0 && bodies)
	{
		HIP_TRY(hipMemcpyAsync(bodies, s->dBodies.p, (size_t)s->bodyCapacity * sizeof(s2amdBody), hipMemcpyDeviceToHost, s->stream));
	}
	if (s->contactCapacity > 0 && contacts)
	{
		HIP_TRY(hipMemcpyAsync(contacts, s->dContacts.p, (size_t)s->contactCapacity * sizeof(s2amdContact), hipMemcpyDeviceToHost, s->stream));
	}
	if (s->jointCapacity > 0 && joints)
	{
		HIP_TRY(hipMemcpyAsync(joints, s->dJoints.p, (size_t)s->jointCapacity * sizeof(s2amdJoint), hipMemcpyDeviceToHost, s->stream));
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

} // namespace

hipStream_t s2amdStream(s2amdSolver* s)
{
	return s->stream;
}
// kernel time of the last stage call (narrowphase.hip, broadphase.hip report through s2amdStepStats.deviceMs)
void s2amdRecordDeviceMs(s2amdSolver* s, float ms)
{
	s->stats.deviceMs = ms;
}
int s2amdDevice(s2amdSolver* s)
{
	return s->device;
}

#pragma GCC visibility push(default)
extern "C"
{

int s2amd_api_version(void)
{
	return S2AMD_API_VERSION;
}

const char* s2amd_last_error(void)
{
	return g_lastError.c_str();
}

int s2amd_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
	{
		(void)hipGetLastError();
		return 0;
	}
	return n;
}

int s2amd_create(int device, s2amdSolver** out)
{
	if (!out)
	{
		return fail(S2AMD_E_INVALID, "null out pointer");
	}
	*out = nullptr;
	int n = s2amd_device_count();
	if (n <= 0)
	{
		return fail(S2AMD_E_NODEVICE, "no HIP device visible; this library has no CPU path");
	}
	if (device < 0 || device >= n)
	{
		return fail(S2AMD_E_INVALID, "device ordinal out of range");
	}
	HIP_TRY(hipSetDevice(device));
	s2amdSolver* s = new s2amdSolver();
	s->device = device;
	hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
	if (e == hipSuccess)
	{
		e = hipEventCreate(&s->evBegin);
	}
	if (e == hipSuccess)
	{
		e = hipEventCreate(&s->evEnd);
	}
	for (int i = 0; i < 2 && e == hipSuccess; ++i)
	{
		e = hipStreamCreateWithFlags(&s->side[i], hipStreamNonBlocking);
		if (e == hipSuccess)
		{
			e = hipEventCreateWithFlags(&s->evFork[i], hipEventDisableTiming);
		}
	}
	for (int i = 0; i < 4 && e == hipSuccess; ++i)
	{
		e = hipEventCreateWithFlags(&s->evJoin[i], hipEventDisableTiming);
	}
	if (e != hipSuccess)
	{
		delete s;
		return fail(S2AMD_E_DEVICE, std::string("stream/event creation: ") + hipGetErrorString(e));
	}
	(void)hipDeviceGetAttribute(&s->cuCount, hipDeviceAttributeMultiprocessorCount, device);

	if (hipHostMalloc((void**)&s->hostError, sizeof(unsigned int), hipHostMallocMapped) == hipSuccess)
	{
		*s->hostError = 0u;
	}
	else
	{
		s->hostError = nullptr;
		(void)hipGetLastError();
	}
	if (groupKernelSetup() != 0 || stripKernelSetup() != 0)
	{
		(void)hipGetLastError(); // not fatal: groups are then limited to the default 64 KiB of LDS
		s->optMaxGroupBodies = 1536;
	}
	*out = s;
	return S2AMD_OK;
}

void s2amd_destroy(s2amdSolver* s)
{
	if (!s)
	{
		return;
	}
	(void)hipSetDevice(s->device);
	(void)hipStreamSynchronize(s->stream);
	destroyGraph(s);
	for (hipEvent_t e : s->sweepEvents)
	{
		(void)hipEventDestroy(e);
	}
	DevBuf* bufs[] = {&s->dBodies,		&s->dContacts,	  &s->dJoints,		 &s->dBodiesSaved,	 &s->dBodyFlags,	&s->soaBodies,
					  &s->soaContacts,	&s->soaJoints,	  &s->dContactIndex, &s->dJointIndex,	 &s->dContactLocal, &s->dJointLocal,
					  &s->dAdjOffsets,	&s->dAdjList,	  &s->dGatherIndex,	 &s->dOps,			 &s->dGroups.buf,	&s->dContactTail.buf,
					  &s->dJointTail.buf, &s->dMsg,			  &s->dStripA.buf,	 &s->dStripB.buf,	 &s->dStripLean,	&s->dPersist,
					  &s->dGranules,	&s->dPersistOps};
	for (DevBuf* b : bufs)
	{
		b->release();
	}
	if (s->hostError)
	{
		(void)hipHostFree(s->hostError);
	}
	if (s->hostTimes)
	{
		int n = (int)s->hostTimes[255];
		fprintf(stderr, "[s2amd] persistent step, one workgroup, wall_clock64 ticks (10 ns) since kernel start:");
		for (int i = 1; i < n && i < 255; ++i)
		{
			fprintf(stderr, " %llu", s->hostTimes[i] - s->hostTimes[0]);
		}
		fprintf(stderr, "\n");
		(void)hipHostFree(s->hostTimes);
	}
	(void)hipEventDestroy(s->evBegin);
	(void)hipEventDestroy(s->evEnd);
	for (int i = 0; i < 2; ++i)
	{
		if (s->evFork[i])
		{
			(void)hipEventDestroy(s->evFork[i]);
		}
		if (s->side[i])
		{
			(void)hipStreamDestroy(s->side[i]);
		}
	}
	for (int i = 0; i < 4; ++i)
	{
		if (s->evJoin[i])
		{
			(void)hipEventDestroy(s->evJoin[i]);
		}
	}
	(void)hipStreamDestroy(s->stream);
	delete s;
}

int s2amd_upload(s2amdSolver* s, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
				 const s2amdJoint* joints, int32_t jointCapacity)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	s->gatherIndexDirty = true;
	int rc = doUpload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
	if (rc == S2AMD_OK)
	{
		HIP_TRY(hipStreamSynchronize(s->stream));
	}
	return rc;
}

int s2amd_step_resident(s2amdSolver* s, const s2amdStepParams* params)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	return doStep(s, params);
}

int s2amd_download(s2amdSolver* s, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts, int32_t contactCapacity, s2amdJoint* joints,
				   int32_t jointCapacity)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	return doDownload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
}

int s2amd_solve(s2amdSolver* s, const s2amdStepParams* params, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts,
				int32_t contactCapacity, s2amdJoint* joints, int32_t jointCapacity)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	s->gatherIndexDirty = true;
	int rc = doUpload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
	if (rc)
	{
		return rc;
	}
	rc = doStep(s, params);
	if (rc)
	{
		return rc;
	}
	return doDownload(s, bodies, bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
}

int s2amd_save_bodies(s2amdSolver* s)
{
	if (!s || !s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident");
	}
	HIP_TRY(hipSetDevice(s->device));
	int rc = s->dBodiesSaved.ensure((size_t)std::max(s->bodyCapacity, 1) * sizeof(s2amdBody));
	if (rc)
	{
		return rc;
	}
	if (s->bodyCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dBodiesSaved.p, s->dBodies.p, (size_t)s->bodyCapacity * sizeof(s2amdBody), hipMemcpyDeviceToDevice, s->stream));
	}
	HIP_TRY(hipStreamSynchronize(s->stream));
	s->savedValid = true;
	return S2AMD_OK;
}

int s2amd_synchronize(s2amdSolver* s)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipStreamSynchronize(s->stream));
	if (s->hostError && *s->hostError != 0u)
	{
		*s->hostError = 0u;
		(void)hipMemsetAsync(s->dGranules.p, 0, s->granuleBytes, s->stream);
		return fail(S2AMD_E_DEVICE, "strip hand-off timed out inside the persistent step kernel (workgroups not co-resident?)");
	}
	return S2AMD_OK;
}

int s2amd_restore_bodies(s2amdSolver* s)
{
	if (!s || !s->resident || !s->savedValid)
	{
		return fail(S2AMD_E_STATE, "no saved bodies");
	}
	HIP_TRY(hipSetDevice(s->device));
	if (s->bodyCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dBodies.p, s->dBodiesSaved.p, (size_t)s->bodyCapacity * sizeof(s2amdBody), hipMemcpyDeviceToDevice, s->stream));
	}
	return S2AMD_OK;
}

static int copyOrder(const std::vector<int>& order, const std::vector<int>& offsets, int32_t* outOrder, int32_t orderCapacity,
					 int32_t* outOffsets, int32_t colorCapacity, int32_t* count, int32_t* colorCount)
{
	int n = (int)order.size();
	int nc = offsets.empty() ? 0 : (int)offsets.size() - 1;
	if (count)
	{
		*count = n;
	}
	if (colorCount)
	{
		*colorCount = nc;
	}
	if (outOrder)
	{
		if (orderCapacity < n)
		{
			return fail(S2AMD_E_CAPACITY, "order buffer too small");
		}
		std::copy(order.begin(), order.end(), outOrder);
	}
	if (outOffsets)
	{
		if (colorCapacity < nc + 1)
		{
			return fail(S2AMD_E_CAPACITY, "colour offset buffer too small");
		}
		std::copy(offsets.begin(), offsets.end(), outOffsets);
		if (offsets.empty())
		{
			outOffsets[0] = 0;
		}
	}
	return S2AMD_OK;
}

int s2amd_get_contact_order(s2amdSolver* s, int32_t* order, int32_t orderCapacity, int32_t* colorOffsets, int32_t colorCapacity,
							int32_t* constraintCount, int32_t* colorCount)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	return copyOrder(s->contacts.order, s->contacts.colorOffsets, order, orderCapacity, colorOffsets, colorCapacity, constraintCount, colorCount);
}

int s2amd_get_joint_order(s2amdSolver* s, int32_t* order, int32_t orderCapacity, int32_t* colorOffsets, int32_t colorCapacity, int32_t* jointCount,
						  int32_t* colorCount)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	return copyOrder(s->joints.order, s->joints.colorOffsets, order, orderCapacity, colorOffsets, colorCapacity, jointCount, colorCount);
}

int s2amd_get_stats(s2amdSolver* s, s2amdStepStats* stats)
{
	if (!s || !stats)
	{
		return fail(S2AMD_E_INVALID, "null argument");
	}
	*stats = s->stats;
	return S2AMD_OK;
}

int s2amd_export_poses(s2amdSolver* s, void* devicePoses, int32_t capacity)
{
	if (!s || !s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident");
	}
	if (!devicePoses || capacity < s->bodyCapacity)
	{
		return fail(S2AMD_E_CAPACITY, "pose buffer missing or too small");
	}
	HIP_TRY(hipSetDevice(s->device));
	launchExportPoses(s->stream, (const s2amdBody*)s->dBodies.p, s->bodyCapacity, devicePoses);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

int s2amd_measure_dominant(s2amdSolver* s, const s2amdStepParams* params, int32_t repeats, float* usPerLaunch, int32_t* launchesPerSweep,
						   int32_t* constraintsPerLaunch)
{
	if (!s || !params || !usPerLaunch || repeats <= 0)
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (!s->resident)
	{
		return fail(S2AMD_E_STATE, "nothing resident");
	}
	HIP_TRY(hipSetDevice(s->device));
	// make sure plan, structure and the device op list are current (one ordinary step)
	int rc = doStep(s, params);
	if (rc)
	{
		return rc;
	}
	const StepPlan& plan = s->plan;
	int dominant = -1;
	for (int i = 0; i < (int)plan.ops.size(); ++i)
	{
		if (Executor::isSolveSweep(plan.ops[(size_t)i].code))
		{
			dominant = i;
			break;
		}
	}
	Executor q{s, s->stream, plan, isPositionSolver(params->solverType) ? 1 : 0, false};
	q.msg = messageEligible(s, params->solverType);
	const bool global = s->contacts.globalCount > 0 && dominant >= 0;
	const bool strips = !global && s->contacts.stripCount > 0 && dominant >= 0;
	int pkind = -1, pwarm = -1;
	const bool persistent = strips && q.persistPlan(pkind, pwarm);
	hipGraph_t g = nullptr;
	hipGraphExec_t ge = nullptr;
	s->launchCounter = 0;
	HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
	for (int r = 0; r < repeats; ++r)
	{
		if (global)
		{
			q.runGlobalOp(dominant);
		}
		else if (persistent)
		{
			q.runPersistent(pkind, pwarm, true);
		}
		else if (strips)
		{
			q.launchStripGroups(s->dStripA, dominant, 1, false);
			if (s->contacts.seamCount > 0)
			{
				q.launchStripGroups(s->dStripB, dominant, 1, false);
			}
		}
		else if (s->dGroups.view.groupCount > 0)
		{
			launchGroupKernel(s->stream, s->cv, s->jv, s->bv, s->dGroups.view, q.deviceOps(), (int)plan.ops.size(), plan.sc,
							  (s2amdContact*)s->dContacts.p, s->dGroups.maxBodies, plan.usesDq0 ? 1 : 0);
			s->launchCounter += 1;
		}
	}
	hipError_t ce = hipStreamEndCapture(s->stream, &g);
	if (ce != hipSuccess)
	{
		return fail(S2AMD_E_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
	}
	int launches = s->launchCounter;
	if (launches == 0)
	{
		(void)hipGraphDestroy(g);
		*usPerLaunch = 0.0f;
		return S2AMD_OK;
	}
	HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	HIP_TRY(hipGraphLaunch(ge, s->stream)); // warm
	HIP_TRY(hipStreamSynchronize(s->stream));
	const int reps = 5;
	HIP_TRY(hipEventRecord(s->evBegin, s->stream));
	for (int r = 0; r < reps; ++r)
	{
		HIP_TRY(hipGraphLaunch(ge, s->stream));
	}
	HIP_TRY(hipEventRecord(s->evEnd, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	float ms = 0.0f;
	HIP_TRY(hipEventElapsedTime(&ms, s->evBegin, s->evEnd));
	(void)hipGraphExecDestroy(ge);
	(void)hipGraphDestroy(g);
	*usPerLaunch = 1e3f * ms / (float)(reps * launches);
	if (persistent)
	{
		// the whole step is one launch (+ the memset of the hand-off buffers in front of it)
		*usPerLaunch = 1e3f * ms / (float)(reps * repeats);
		launches = repeats;
	}
	if (launchesPerSweep)
	{
		*launchesPerSweep = launches / repeats;
	}
	if (constraintsPerLaunch)
	{
		*constraintsPerLaunch = global		 ? s->contacts.globalCount / std::max(launches / repeats, 1)
								: persistent ? s->contacts.stripCount * plan.solveSweeps // constraint-sweeps of the whole-step launch
								: strips	 ? s->contacts.stripCount / std::max(launches / repeats, 1)
											 : s->cv.count;
	}
	return S2AMD_OK;
}

int s2amd_set_option(s2amdSolver* s, const char* key, int32_t value)
{
	if (!s || !key)
	{
		return fail(S2AMD_E_INVALID, "null argument");
	}
	if (strcmp(key, "graph") == 0)
	{
		s->optGraph = value;
	}
	else if (strcmp(key, "profile") == 0)
	{
		s->optProfile = value;
	}
	else if (strcmp(key, "message") == 0)
	{
		s->optMessage = value;
	}
	else if (strcmp(key, "body_warm") == 0)
	{
		s->optBodyWarm = value;
	}
	else if (strcmp(key, "groups") == 0)
	{
		s->optGroups = value;
		s->structureDirty = true;
	}
	else if (strcmp(key, "max_group_bodies") == 0)
	{
		s->stripsRejected = false;
		s->optMaxGroupBodies = std::max(1, std::min(value, 3072));
		s->structureDirty = true;
	}
	else if (strcmp(key, "strips") == 0)
	{
		s->stripsRejected = false;
		s->optStrips = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "async") == 0)
	{
		s->optAsync = value != 0;
	}
	else if (strcmp(key, "fork") == 0)
	{
		s->optFork = value != 0;
	}
	else if (strcmp(key, "persist_spin_limit") == 0)
	{
		s->optPersistSpinLimit = std::max(256, value);
		s->stripsRejected = false;
		s->structureDirty = true;
	}
	else if (strcmp(key, "persist_debug") == 0)
	{
		s->stripsRejected = false;
		s->optPersistDebug = value;
		s->structureDirty = true;
	}
	else if (strcmp(key, "persist") == 0)
	{
		s->stripsRejected = false;
		s->optPersist = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_patience") == 0)
	{
		s->optStripPatience = std::max(0, value);
	}
	else if (strcmp(key, "strips_any_solver") == 0)
	{
		s->stripsRejected = false;
		s->optStripsAnySolver = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_lean") == 0)
	{
		s->stripsRejected = false;
		s->optStripLean = value != 0;
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_bodies") == 0)
	{
		s->stripsRejected = false;
		s->optStripBodies = std::max(1, value);
		s->structureDirty = true;
	}
	else if (strcmp(key, "strip_min_bodies") == 0)
	{
		s->stripsRejected = false;
		s->optStripMinBodies = std::max(0, value);
		s->structureDirty = true;
	}
	else if (strcmp(key, "pack_group_bodies") == 0)
	{
		s->optPackGroupBodies = std::max(1, value);
		s->structureDirty = true;
	}
	else
	{
		return fail(S2AMD_E_INVALID, std::string("unknown option ") + key);
	}
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop
