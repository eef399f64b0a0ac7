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

int colorGraph(const std::vector<int>& ea, const std::vector<int>& eb, const std::vector<uint8_t>& conflict, int bodyCount,
			   std::vector<int>& color)
{
	const int W = ColorMasks::WORDS;
	size_t n = ea.size();
	color.assign(n, 0);
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

struct s2amdSolver
{
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t evBegin = nullptr, evEnd = nullptr;

	// wire arrays resident on the device
	DevBuf dBodies, dContacts, dJoints, dBodiesSaved;
	int bodyCapacity = 0, contactCapacity = 0, jointCapacity = 0;
	bool resident = false;
	bool savedValid = false;

	// host shadows of the graph structure (refreshed by every upload)
	std::vector<int> hContactA, hContactB, hContactPoints;
	std::vector<int> hJointType, hJointA, hJointB;
	std::vector<uint32_t> hBodyFlags; // S2F_WRITE_VEL / S2F_WRITE_POS
	DevBuf dBodyFlags;

	// working SoA
	DevBuf soaBodies, soaContacts, soaJoints, dContactIndex, dJointIndex, dAdjOffsets, dAdjList;
	BodyView bv{};
	ContactView cv{};
	JointView jv{};
	uint64_t layoutGeneration = 0;
	int bodySoaCap = 0, contactSoaCap = 0, jointSoaCap = 0;

	// sweep order of the last step
	std::vector<int> contactOrder, contactColorOffsets, jointOrder, jointColorOffsets;
	int orderSolverClass = -1; // 0 velocity colouring, 1 position colouring
	bool adjValid = false;
	bool structureDirty = true;

	// options
	int optGraph = 1;
	int optProfile = 0;

	// graph cache
	hipGraph_t graph = nullptr;
	hipGraphExec_t graphExec = nullptr;
	uint64_t graphKey = 0;

	// profiling events for the contact solve sweeps
	std::vector<hipEvent_t> sweepEvents;
	size_t sweepEventsUsed = 0;

	s2amdStepStats stats{};
	int launchCounter = 0;
	int sweepCounter = 0;
	int graphLaunches = 0, graphSweeps = 0;
	DevBuf dGatherIndex;
	bool gatherIndexDirty = true;
};

namespace
{

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

// Build sweep order + colour batches for the given solver on the host, upload the index tables.
int buildOrder(s2amdSolver* s, int solverType)
{
	// class 0: velocity-level colouring, 1: position-level colouring.  Jacobi uses class 0 for its
	// warm start and joints and additionally needs the body -> constraint incidence lists.
	int cls = isPositionSolver(solverType) ? 1 : 0;
	const bool needAdj = solverType == s2amd_solverJacobi;
	if (!s->structureDirty && cls == s->orderSolverClass && (!needAdj || s->adjValid))
	{
		return S2AMD_OK;
	}
	double t0 = nowMs();
	const bool pos = cls == 1;
	int nb = s->bodyCapacity;
	std::vector<uint8_t> conflict((size_t)nb);
	for (int i = 0; i < nb; ++i)
	{
		conflict[i] = (s->hBodyFlags[i] & (pos ? S2F_WRITE_POS : S2F_WRITE_VEL)) != 0;
	}

	// contacts: gather in pool order (e.g. solve_tgs_soft.c:162-179)
	std::vector<int> ids, ea, eb;
	ids.reserve(s->contactCapacity);
	for (int i = 0; i < s->contactCapacity; ++i)
	{
		if (s->hContactPoints[i] > 0)
		{
			ids.push_back(i);
			ea.push_back(s->hContactA[i]);
			eb.push_back(s->hContactB[i]);
		}
	}
	{
		std::vector<int> color;
		int cc = colorGraph(ea, eb, conflict, nb, color);
		sortByColor(ids, color, cc, s->contactOrder, s->contactColorOffsets);
	}

	// joints: live joints in pool order; a mouse joint only touches body B
	std::vector<int> jids, ja, jb;
	for (int i = 0; i < s->jointCapacity; ++i)
	{
		if (s->hJointType[i] != S2AMD_JOINT_FREE)
		{
			jids.push_back(i);
			ja.push_back(s->hJointType[i] == S2AMD_JOINT_MOUSE ? -1 : s->hJointA[i]);
			jb.push_back(s->hJointB[i]);
		}
	}
	{
		std::vector<int> color;
		int jc = colorGraph(ja, jb, conflict, nb, color);
		sortByColor(jids, color, jc, s->jointOrder, s->jointColorOffsets);
	}

	int C = (int)s->contactOrder.size(), J = (int)s->jointOrder.size();
	int rc;
	if ((rc = carveContacts(s, C)) != 0 || (rc = carveJoints(s, J)) != 0)
	{
		return rc;
	}
	bool grew = false;
	if ((rc = s->dContactIndex.ensure((size_t)std::max(C, 1) * sizeof(int), &grew)) != 0)
	{
		return rc;
	}
	if ((rc = s->dJointIndex.ensure((size_t)std::max(J, 1) * sizeof(int), &grew)) != 0)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
	}
	if (C > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dContactIndex.p, s->contactOrder.data(), (size_t)C * sizeof(int), hipMemcpyHostToDevice, s->stream));
	}
	if (J > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dJointIndex.p, s->jointOrder.data(), (size_t)J * sizeof(int), hipMemcpyHostToDevice, s->stream));
	}
	s->cv.contactIndex = (int*)s->dContactIndex.p;
	s->cv.count = C;
	s->jv.jointIndex = (int*)s->dJointIndex.p;
	s->jv.count = J;

	s->adjValid = false;
	if (needAdj)
	{
		// body -> incident constraints in SWEEP order (ascending k), key = k<<1 | side, so the per-body
		// sums of jacobiApplyKernel add in exactly the order a sequential pass in sweep order would;
		// read-only shareable bodies are skipped (their deltas are exact zeros)
		std::vector<int> offsets((size_t)nb + 1, 0), list;
		for (int k = 0; k < C; ++k)
		{
			int a = s->hContactA[s->contactOrder[k]], b = s->hContactB[s->contactOrder[k]];
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
		for (int k = 0; k < C; ++k)
		{
			int a = s->hContactA[s->contactOrder[k]], b = s->hContactB[s->contactOrder[k]];
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
		if ((rc = s->dAdjOffsets.ensure(((size_t)nb + 1) * sizeof(int), &grew)) != 0)
		{
			return rc;
		}
		if ((rc = s->dAdjList.ensure(std::max<size_t>(list.size(), 1) * sizeof(int), &grew)) != 0)
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
		// the vectors go out of scope after an async copy from pageable memory: hipMemcpyAsync from
		// pageable host memory stages the data before returning, so this is safe.
		s->adjValid = true;
	}

	s->orderSolverClass = cls;
	s->structureDirty = false;
	s->stats.hostPrepMs = (float)(nowMs() - t0);
	return S2AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// drivers: each is the launch sequence of one reference s2Solve_* function
// ------------------------------------------------------------------------------------------------
struct Enqueue
{
	s2amdSolver* s;
	hipStream_t st;
	StepConsts sc;
	int posSolver;
	bool profile;

	s2amdContact* wireContacts() const { return (s2amdContact*)s->dContacts.p; }
	s2amdBody* wireBodies() const { return (s2amdBody*)s->dBodies.p; }
	s2amdJoint* wireJoints() const { return (s2amdJoint*)s->dJoints.p; }

	int contactColors() const { return (int)s->contactColorOffsets.size() - 1; }
	int jointColors() const { return (int)s->jointColorOffsets.size() - 1; }

	void count(int n = 1) { s->launchCounter += n; }

	bool inSweep = false;
	void markSweepBegin()
	{
		s->sweepCounter += 1;
		inSweep = true;
	}
	void markSweepEnd()
	{
		inSweep = false;
	}
	void recordSweepEvent()
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

	template <class F> void eachContactColor(F f)
	{
		for (int c = 0; c < contactColors(); ++c)
		{
			int b = s->contactColorOffsets[c], e = s->contactColorOffsets[(size_t)c + 1];
			if (e > b)
			{
				// profiling: a HIP event pair around every solve-sweep launch (on the launch stream)
				const bool timed = profile && inSweep;
				if (timed)
				{
					recordSweepEvent();
				}
				f(b, e);
				if (timed)
				{
					recordSweepEvent();
				}
				count();
			}
		}
	}
	template <class F> void eachJointColor(F f)
	{
		for (int c = 0; c < jointColors(); ++c)
		{
			int b = s->jointColorOffsets[c], e = s->jointColorOffsets[(size_t)c + 1];
			if (e > b)
			{
				f(b, e);
				count();
			}
		}
	}

	// stages
	void unpack(float h)
	{
		launchUnpackBodies(st, s->bv, wireBodies(), (const uint32_t*)s->dBodyFlags.p, sc, h);
		count();
	}
	void pack()
	{
		launchPackBodies(st, s->bv, wireBodies());
		count();
	}
	void integrateVelocities()
	{
		launchIntegrateVelocities(st, s->bv);
		count();
	}
	void integratePositions(float h)
	{
		launchIntegratePositions(st, s->bv, h);
		count();
	}
	void finalizePositions(int dynamicOnly = 0)
	{
		launchFinalizePositions(st, s->bv, dynamicOnly);
		count();
	}
	void prepareContacts(int kind, float h, float hertz)
	{
		if (s->cv.count > 0)
		{
			launchPrepareContacts(st, kind, s->cv, s->bv, wireContacts(), wireBodies(), sc, h, hertz, posSolver);
			count();
		}
	}
	void prepareJoints(int kind, float h, float hertz, bool warmStart)
	{
		if (s->jv.count > 0)
		{
			launchPrepareJoints(st, kind, s->jv, s->bv, wireJoints(), wireBodies(), sc, h, hertz, warmStart ? 1 : 0, posSolver);
			count();
		}
	}
	void warmStartContacts(int kind)
	{
		eachContactColor([&](int b, int e) { launchWarmStartContacts(st, kind, s->cv, s->bv, b, e); });
	}
	void jointSweep(int kind, float h, float inv_h, bool useBias)
	{
		eachJointColor([&](int b, int e) { launchSolveJoints(st, kind, s->jv, s->bv, b, e, sc, h, inv_h, useBias ? 1 : 0); });
	}
	void solveSoft(int kind, float inv_h, bool useBias)
	{
		markSweepBegin();
		if (kind == SOFT_JACOBI)
		{
			// the Jacobi pass writes per-constraint deltas, never a body: one launch for all colours
			if (s->cv.count > 0)
			{
				if (profile)
				{
					recordSweepEvent();
				}
				launchSolveContactsSoft(st, kind, s->cv, s->bv, 0, s->cv.count, inv_h, useBias ? 1 : 0);
				if (profile)
				{
					recordSweepEvent();
				}
				count();
			}
		}
		else
		{
			eachContactColor([&](int b, int e) { launchSolveContactsSoft(st, kind, s->cv, s->bv, b, e, inv_h, useBias ? 1 : 0); });
		}
		markSweepEnd();
	}
	void solveRigid(int kind, float inv_h)
	{
		markSweepBegin();
		eachContactColor([&](int b, int e) { launchSolveContactsRigid(st, kind, s->cv, s->bv, b, e, inv_h); });
		markSweepEnd();
	}
	void solveNGS()
	{
		markSweepBegin();
		eachContactColor([&](int b, int e) { launchSolveContactsNGS(st, s->cv, s->bv, b, e); });
		markSweepEnd();
	}
	void solveSticky(float inv_h, bool useBias)
	{
		markSweepBegin();
		eachContactColor([&](int b, int e) { launchSolveContactsSticky(st, s->cv, s->bv, wireContacts(), b, e, inv_h, useBias ? 1 : 0); });
		markSweepEnd();
	}
	void storeImpulses(int kind, float scale = 0.0f)
	{
		if (s->cv.count > 0)
		{
			launchStoreImpulses(st, kind, s->cv, wireContacts(), scale);
			count();
		}
	}
	void storeJoints()
	{
		if (s->jv.count > 0)
		{
			launchStoreJoints(st, s->jv, wireJoints());
			count();
		}
	}
	void jacobiApply()
	{
		launchJacobiApply(st, s->bv, s->cv, (const int*)s->dAdjOffsets.p, (const int*)s->dAdjList.p);
		count();
	}

	// s2Solve_TGS_Soft (solve_tgs_soft.c:138-280) / s2Solve_SoftStep (solve_soft_step.c:182-311)
	void solveTgsSoft(bool fixedAnchors)
	{
		int substepCount = sc.iterations;
		float h = sc.h, inv_h = sc.inv_h;
		float contactHertz = S2_MINF(S2_CONTACT_HERTZ, 0.25f * inv_h);
		float jointHertz = fixedAnchors ? S2_MINF(S2_JOINT_HERTZ, 0.25f * inv_h) : S2_MINF(S2_JOINT_HERTZ, 0.125f * inv_h);
		unpack(h);
		prepareContacts(PREP_SOFT, h, contactHertz);
		prepareJoints(JPREP_SOFT, h, jointHertz, true);
		for (int substep = 0; substep < substepCount; ++substep)
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
		unpack(h);
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
				jacobiApply();
			}
		}
		integratePositions(h);
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			jointSweep(JSOLVE_SOFT, h, inv_h, false);
			solveSoft(jacobi ? SOFT_JACOBI : SOFT_PGS, inv_h, false);
			if (jacobi)
			{
				jacobiApply();
			}
		}
		finalizePositions();
		storeImpulses(STORE_PLAIN);
	}

	// s2Solve_PGS: solve_pgs.c:125-213
	void solvePgs()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		unpack(h);
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

	// s2Solve_PGS_NGS: solve_pgs_ngs.c:149-255
	void solvePgsNgs()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		unpack(h);
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
		storeImpulses(STORE_PLAIN); // before the position sweeps: solve_pgs_ngs.c:232
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			jointSweep(JSOLVE_POSITION, h, inv_h, false);
			solveNGS();
		}
		finalizePositions();
	}

	// s2Solve_PGS_NGS_Block: solve_pgs_ngs_block.c:892-963
	void solveBlock()
	{
		float h = sc.dt, inv_h = sc.inv_dt;
		unpack(h);
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
			markSweepBegin();
			eachContactColor([&](int b, int e) { launchBlockSolveVelocity(st, s->cv, s->bv, b, e); });
			markSweepEnd();
		}
		storeImpulses(STORE_BLOCK);
		integratePositions(h);
		for (int iter = 0; iter < sc.extraIterations; ++iter)
		{
			markSweepBegin();
			eachContactColor([&](int b, int e) { launchBlockSolvePosition(st, s->cv, s->bv, b, e); });
			markSweepEnd();
			jointSweep(JSOLVE_POSITION, h, inv_h, false); // contacts before joints here (:945-957)
		}
		finalizePositions();
	}

	// s2Solve_TGS_NGS: solve_tgs_ngs.c:207-317
	void solveTgsNgs()
	{
		float h = sc.h, inv_h = sc.inv_h;
		unpack(h);
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
		unpack(h);
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
			return;
		}
		float h = sc.dt / substepCount;
		float inv_h = 1.0f / h;
		unpack(h);
		prepareContacts(PREP_XPBD, h, 0.0f);
		prepareJoints(JPREP_XPBD, h, 0.0f, false);
		for (int substep = 0; substep < substepCount; ++substep)
		{
			launchXpbdIntegrate(st, s->bv, h);
			count();
			jointSweep(JSOLVE_XPBD, h, inv_h, false);
			markSweepBegin();
			eachContactColor([&](int b, int e) { launchXpbdContactPositions(st, s->cv, s->bv, b, e, h); });
			markSweepEnd();
			launchXpbdProject(st, s->bv, inv_h);
			count();
			markSweepBegin();
			eachContactColor([&](int b, int e) { launchXpbdContactVelocities(st, s->cv, s->bv, b, e, h); });
			markSweepEnd();
		}
		finalizePositions(1);
		storeImpulses(STORE_SCALED, inv_h);
	}

	void run(int solverType)
	{
		switch (solverType)
		{
			case s2amd_solverJacobi:
				solveJacobiOrPgsSoft(true);
				break;
			case s2amd_solverPGS:
				solvePgs();
				break;
			case s2amd_solverPGS_NGS:
				solvePgsNgs();
				break;
			case s2amd_solverPGS_NGS_Block:
				solveBlock();
				break;
			case s2amd_solverPGS_Soft:
				solveJacobiOrPgsSoft(false);
				break;
			case s2amd_solverSoftStep:
				solveTgsSoft(true);
				break;
			case s2amd_solverTGS_Sticky:
				solveTgsSticky();
				break;
			case s2amd_solverTGS_Soft:
				solveTgsSoft(false);
				break;
			case s2amd_solverTGS_NGS:
				solveTgsNgs();
				break;
			case s2amd_solverXPBD:
				solveXpbd();
				break;
		}
		if (!(solverType == s2amd_solverXPBD && (sc.iterations == 0 || sc.dt == 0.0f)))
		{
			storeJoints();
			pack();
		}
	}
};

} // namespace


// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
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
	for (int i = 0; i < nb; ++i)
	{
		const s2amdBody& b = bodies[i];
		uint32_t f = 0;
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
		HIP_TRY(hipMemcpyAsync(s->dBodyFlags.p, s->hBodyFlags.data(), (size_t)nb * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
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
	int rc = buildOrder(s, params->solverType);
	if (rc)
	{
		return rc;
	}

	Enqueue q{s, s->stream, makeConsts(params), isPositionSolver(params->solverType) ? 1 : 0, s->optProfile != 0};
	s->launchCounter = 0;
	s->sweepCounter = 0;
	s->sweepEventsUsed = 0;

	const bool xpbdEarlyOut = params->solverType == s2amd_solverXPBD && (params->velIters == 0 || params->dt == 0.0f);
	const bool writesConstraintIndex = !xpbdEarlyOut && params->solverType != s2amd_solverPGS_NGS_Block;

	// manifold.constraintIndex (pool-order gather index, -1 for skipped slots)
	if (writesConstraintIndex && s->contactCapacity > 0)
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
		if (s->gatherIndexDirty || s->dGatherIndex.bytes < gi.size() * sizeof(int))
		{
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
			s->gatherIndexDirty = false;
		}
	}

	auto enqueueAll = [&]() {
		if (writesConstraintIndex && s->contactCapacity > 0)
		{
			int n = s->contactCapacity;
			writeConstraintIndexKernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream>>>((s2amdContact*)s->dContacts.p, n,
																										 (const int*)s->dGatherIndex.p);
			q.count();
		}
		q.run(params->solverType);
	};

	bool useGraph = s->optGraph != 0 && !q.profile;
	HIP_TRY(hipEventRecord(s->evBegin, s->stream));
	if (useGraph)
	{
		uint64_t key = 1469598103934665603ull;
		key = fnv(key, params, sizeof(*params));
		key = fnv(key, &s->layoutGeneration, sizeof(s->layoutGeneration));
		int sizes[5] = {s->bodyCapacity, s->contactCapacity, s->jointCapacity, s->cv.count, s->jv.count};
		key = fnv(key, sizes, sizeof(sizes));
		key = fnv(key, s->contactColorOffsets.data(), s->contactColorOffsets.size() * sizeof(int));
		key = fnv(key, s->jointColorOffsets.data(), s->jointColorOffsets.size() * sizeof(int));
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
			s->graphSweeps = s->sweepCounter;
		}
		else
		{
			s->launchCounter = s->graphLaunches;
			s->sweepCounter = s->graphSweeps;
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
	HIP_TRY(hipStreamSynchronize(s->stream));

	float ms = 0.0f;
	HIP_TRY(hipEventElapsedTime(&ms, s->evBegin, s->evEnd));
	s->stats.deviceMs = ms;
	s->stats.constraintCount = s->cv.count;
	s->stats.jointCount = s->jv.count;
	s->stats.contactColors = (int)s->contactColorOffsets.size() - 1;
	s->stats.jointColors = (int)s->jointColorOffsets.size() - 1;
	s->stats.solveSweeps = s->sweepCounter;
	s->stats.kernelLaunches = s->launchCounter;
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
	if (e != hipSuccess)
	{
		delete s;
		return fail(S2AMD_E_DEVICE, std::string("stream/event creation: ") + hipGetErrorString(e));
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
	DevBuf* bufs[] = {&s->dBodies,	   &s->dContacts,	 &s->dJoints,	   &s->dBodiesSaved, &s->dBodyFlags, &s->soaBodies,	 &s->soaContacts,
					  &s->soaJoints,   &s->dContactIndex, &s->dJointIndex, &s->dAdjOffsets,	 &s->dAdjList,	 &s->dGatherIndex};
	for (DevBuf* b : bufs)
	{
		b->release();
	}
	(void)hipEventDestroy(s->evBegin);
	(void)hipEventDestroy(s->evEnd);
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
	return copyOrder(s->contactOrder, s->contactColorOffsets, order, orderCapacity, colorOffsets, colorCapacity, constraintCount, colorCount);
}

int s2amd_get_joint_order(s2amdSolver* s, int32_t* order, int32_t orderCapacity, int32_t* colorOffsets, int32_t colorCapacity, int32_t* jointCount,
						  int32_t* colorCount)
{
	if (!s)
	{
		return fail(S2AMD_E_INVALID, "null solver");
	}
	return copyOrder(s->jointOrder, s->jointColorOffsets, order, orderCapacity, colorOffsets, colorCapacity, jointCount, colorCount);
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
	else
	{
		return fail(S2AMD_E_INVALID, std::string("unknown option ") + key);
	}
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop
