// Island-sharded worlds behind the C-ABI (include/solver2d_amd.h: s2amd_sharded_*): ONE process, N devices.
//
// The reference has no islands; an island here is a connected component of the graph whose nodes are the movable
// bodies and whose edges are the active contact constraints and the revolute joints (structure.hip: s2amd_find_islands; the CPU
// statement is solver2d_amd/islands.py, which this file follows function by function: constraint_islands, partition,
// sticky_partition, extract, merge_back).  Islands share no movable body, so solving them apart is arithmetic-identical to solving
// them together as long as every shard keeps the pool order of its own bodies and constraints -- what `extract` guarantees.
//
//   upload   islands found on the device, bin-packed onto the shards by constraint count (longest processing time first), every
//            shard's sub-world (its islands' bodies, the immovable bodies they touch as read-only replicas, its constraints)
//            uploaded to its device's solver;
//   step     every shard's s2Solve_* enqueued on its own device -- no collective inside a step --, then the step's ONE exchange of the
//            owned body records ({position, rot}, {linearVelocity, angularVelocity}: the 28 bytes per body of SURVEY.md 8e in two
//            16-byte records) into every device's copy of the WHOLE world's body records -- every device ends the step with every
//            body, what a device-side collision phase needs.  The exchange runs on a stream of its own per shard, behind an event of
//            the solve stream; the next step's solve does not wait for it (it reads only its own shard), only the next export does.
//            Three forms, O(n) stream operations per step in the first two (round 6; r5 enqueued n (n - 1) wait / copy / scatter
//            triples on the destinations' SOLVE streams and synchronised every shard in turn):
//              stores  shards that share one device: ONE kernel per shard writes its rows into every shard's world copy;
//              rccl    n distinct devices: one ncclAllGather per device (single process, ncclCommInitAll; librccl.so is dlopen'ed, the
//                      library loads without it) of the padded compact runs, one scatter kernel per device;
//              copies  whatever is left (some devices shared, some not, or no RCCL): r5's peer copies, on the exchange streams.
//            s2amd_sharded_step_async enqueues and returns; s2amd_sharded_wait is the one place the host waits (one stream) and
//            where a shard's lost hand-off is noticed and its step repeated;
//   reshard  the constraint graph changed: every shard's solver state comes down, islands are found again (device), an island
//            stays on the shard that owned most of its bodies (a created contact that joins two islands moves the smaller one),
//            the partition is rebalanced past REBALANCE_THRESHOLD, the new sub-worlds go up.
//
// The multi-PROCESS form of the same partition (one rank per GPU, torch.distributed over RCCL) is solver2d_amd/distributed.py; the
// tests hold the two against each other and against the unsharded solver, bit for bit.
#include "launch.h"
#include "s2_device.h"

#include "solver2d_amd.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

int s2amdFail(int code, const std::string& msg);
hipStream_t s2amdStream(s2amdSolver* s);
int s2amdDevice(s2amdSolver* s);
const s2amdBody* s2amdResidentBodies(s2amdSolver* s); // solver.cpp: the resident wire bodies (device pointer)
bool s2amdStepFailed(s2amdSolver* s); // solver.cpp: the host-visible error word of an enqueued step (no device call)
extern "C" int s2amd_sharded_wait(s2amdShardedSolver* w);

namespace
{

#define SH_TRY(expr)                                                                                                             \
	do                                                                                                                           \
	{                                                                                                                            \
		hipError_t _e = (expr);                                                                                                  \
		if (_e != hipSuccess)                                                                                                    \
		{                                                                                                                        \
			return s2amdFail(S2AMD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));                                \
		}                                                                                                                        \
	} while (0)

constexpr double kRebalanceThreshold = 1.75; // islands.py: REBALANCE_THRESHOLD

// owned rows of a shard's exported records (two float4 per body, shard-local order) -> a compact run of the same records
__global__ void compactOwnedKernel(const float4* records, const int* ownedLocal, int n, float4* compact)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		const int b = ownedLocal[i];
		compact[2 * i] = records[2 * b];
		compact[2 * i + 1] = records[2 * b + 1];
	}
}

// a compact run -> the rows of the world's body records its bodies have in the pool
__global__ void scatterOwnedKernel(const float4* compact, const int* ownedWorld, int n, float4* world)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		const int b = ownedWorld[i];
		world[2 * b] = compact[2 * i];
		world[2 * b + 1] = compact[2 * i + 1];
	}
}

// the owned rows of a shard into every world copy this device can store to (shards of one device),
// ... straight from the solver's resident wire bodies (the records exportBodiesKernel would make, body_kernels.hip: {position, rot},
// {linearVelocity, angularVelocity, 0}): shards of one device need no staging copy in between
__global__ void pushOwnedBodiesKernel(const s2amdBody* wire, const int* ownedLocal, const int* ownedWorld, int n, float4* const* worlds, int nWorlds)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		const s2amdBody* b = wire + ownedLocal[i];
		const int wi = ownedWorld[i];
		const float4 r0 = make_float4(b->position[0], b->position[1], b->rot[0], b->rot[1]);
		const float4 r1 = make_float4(b->linearVelocity[0], b->linearVelocity[1], b->angularVelocity, 0.0f);
		for (int d = 0; d < nWorlds; ++d)
		{
			worlds[d][2 * wi] = r0;
			worlds[d][2 * wi + 1] = r1;
		}
	}
}

// every rank's padded compact run as ncclAllGather delivered it -> the world's rows (ids[k] < 0: padding)
__global__ void scatterGatheredKernel(const float4* gathered, const int* ids, int total, float4* world)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total && ids[i] >= 0)
	{
		const int b = ids[i];
		world[2 * b] = gathered[2 * i];
		world[2 * b + 1] = gathered[2 * i + 1];
	}
}

// RCCL, looked up at run time: the library loads and runs without it (one GPU, or shards that share one)
struct RcclApi
{
	void* lib = nullptr;
	int (*commInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
	int (*commDestroy)(void* comm) = nullptr;
	int (*groupStart)() = nullptr;
	int (*groupEnd)() = nullptr;
	int (*allGather)(const void* send, void* recv, size_t sendcount, int datatype, void* comm, hipStream_t stream) = nullptr;
	const char* (*errorString)(int) = nullptr;
	bool tried = false;
	bool load()
	{
		if (tried)
		{
			return lib != nullptr;
		}
		tried = true;
		// (the RCCL that sits BESIDE the HIP runtime this process runs on comes first: RCCL opens "libhsa-runtime64.so" by its bare name,
		// which resolves through its own RUNPATH -- PyTorch ships private copies of all three libraries, and an RCCL of one set beside a
		// HIP runtime of the other finds an HSA runtime nobody initialised: "no ROCm-capable device is detected")
		Dl_info hipAt;
		if (dladdr((const void*)&hipGetDeviceCount, &hipAt) != 0 && hipAt.dli_fname != nullptr)
		{
			std::string dir(hipAt.dli_fname);
			const size_t slash = dir.rfind('/');
			if (slash != std::string::npos)
			{
				dir.resize(slash + 1);
				for (const char* leaf : {"librccl.so", "librccl.so.1"})
				{
					if (lib == nullptr)
					{
						lib = dlopen((dir + leaf).c_str(), RTLD_NOW | RTLD_LOCAL);
					}
				}
			}
		}
		const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
		for (const char* name : names)
		{
			if (lib != nullptr || (lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr)
			{
				break;
			}
		}
		if (lib == nullptr)
		{
			return false;
		}
		*(void**)(&commInitAll) = dlsym(lib, "ncclCommInitAll");
		*(void**)(&commDestroy) = dlsym(lib, "ncclCommDestroy");
		*(void**)(&groupStart) = dlsym(lib, "ncclGroupStart");
		*(void**)(&groupEnd) = dlsym(lib, "ncclGroupEnd");
		*(void**)(&allGather) = dlsym(lib, "ncclAllGather");
		*(void**)(&errorString) = dlsym(lib, "ncclGetErrorString");
		if (!commInitAll || !commDestroy || !groupStart || !groupEnd || !allGather)
		{
			dlclose(lib);
			lib = nullptr;
		}
		return lib != nullptr;
	}
};
RcclApi g_rccl;
constexpr int kNcclFloat = 7; // ncclFloat32 (nccl.h: ncclDataType_t)

enum Exchange
{
	EX_STORES = 0,
	EX_RCCL = 1,
	EX_COPIES = 2
};

struct DeviceBlock
{
	void* p = nullptr;
	size_t bytes = 0;
	int device = 0;
	int ensure(int dev, size_t need)
	{
		if (need <= bytes && dev == device && p)
		{
			return S2AMD_OK;
		}
		SH_TRY(hipSetDevice(dev));
		if (p)
		{
			(void)hipFree(p);
			p = nullptr, bytes = 0;
		}
		const size_t want = std::max<size_t>(need + need / 4, 256);
		SH_TRY(hipMalloc(&p, want));
		bytes = want, device = dev;
		return S2AMD_OK;
	}
	void release()
	{
		if (p)
		{
			(void)hipSetDevice(device);
			(void)hipFree(p);
		}
		p = nullptr, bytes = 0;
	}
};

struct Shard
{
	s2amdSolver* solver = nullptr;
	int device = 0;
	// the sub-world (islands.py: Shard) and its maps back into the world
	std::vector<s2amdBody> bodies;
	std::vector<s2amdContact> contacts;
	std::vector<s2amdJoint> joints;
	std::vector<int> bodyIds, contactIds, jointIds; // shard-local index -> pool index of the world
	std::vector<uint8_t> owned;						// per shard body: a body this shard integrates (the rest are read-only replicas)
	std::vector<int> ownedLocal, ownedWorld;		// the owned ones: shard-local index, pool index
	// device side of the exchange, all on this shard's device
	DeviceBlock dRecords, dOwnedLocal, dOwnedWorld, dCompact, dWorld;
	std::vector<DeviceBlock> inbox, inboxIds; // per source shard: its compact run and the pool indices of its rows
	hipEvent_t evCompact = nullptr;			  // this shard's compact run is complete
	// (round 6) the exchange's own stream: it starts behind `evStep` of the solve stream and ends in `evXchg`, which only the next
	// step's export (it rewrites dRecords) and the host's one wait look at
	hipStream_t xs = nullptr;
	hipEvent_t evStep = nullptr, evXchg = nullptr;
	DeviceBlock dPeers;				 // stores: every shard's dWorld, as this device addresses it
	DeviceBlock dGather, dGatherIds; // rccl: n padded compact runs as the all-gather delivers them; the pool index of every row (-1: padding)
	void* comm = nullptr;			 // rccl: this device's communicator
	bool uploaded = false;
};

} // namespace

struct s2amdShardedSolver
{
	std::vector<Shard> shards;
	// the world as of the last upload / reshard (solver state as of the last download)
	std::vector<s2amdBody> bodies;
	std::vector<s2amdContact> contacts;
	std::vector<s2amdJoint> joints;
	std::vector<int32_t> island;	 // per body, -1: static / free
	std::vector<int32_t> shardOfIsland;
	int islandCount = 0;
	int reshards = 0;
	bool resident = false;
	long steps = 0;
	int repeatedSteps = 0; // shard steps repeated after a persistent launch lost a hand-off
	int lastSolverType = -1; // of the last step (-1: none yet): whether its driver writes manifold.constraintIndex
	int exchange = EX_COPIES; // how the step's exchange travels (Exchange; chosen at s2amd_sharded_create)
	size_t maxOwned = 0;	  // rccl: rows per rank of the padded all-gather
	std::vector<s2amdStepParams> outstanding; // steps enqueued since the host last waited (s2amd_sharded_wait)
	// what the last s2amd_sharded_step_async enqueued, counted where it is enqueued: {stream operations (launches, copies, event
	// records and waits, collectives), host waits}
	int32_t opsLastStep = 0, hostWaitsLastStep = 0;
	bool dry = false; // s2amd_sharded_count_ops: count, enqueue nothing
};

namespace
{

bool movable(const s2amdBody& b) { return b.type != S2AMD_BODY_FREE && (b.invMass != 0.0f || b.invI != 0.0f); }

// islands.py: constraint_islands
void constraintIslands(const s2amdShardedSolver& w, std::vector<int>& ci, std::vector<int>& ji)
{
	const int nc = (int)w.contacts.size(), nj = (int)w.joints.size();
	ci.assign((size_t)nc, -1);
	ji.assign((size_t)nj, -1);
	for (int k = 0; k < nc; ++k)
	{
		const s2amdContact& c = w.contacts[(size_t)k];
		if (c.pointCount <= 0)
		{
			continue;
		}
		const int ia = w.island[(size_t)c.bodyA], ib = w.island[(size_t)c.bodyB];
		const int a = movable(w.bodies[(size_t)c.bodyA]) ? ia : -1;
		const int b = movable(w.bodies[(size_t)c.bodyB]) ? ib : -1;
		const int fallback = ia >= 0 ? ia : ib; // e.g. kinematic against static
		ci[(size_t)k] = a >= 0 ? a : (b >= 0 ? b : fallback);
	}
	for (int k = 0; k < nj; ++k)
	{
		const s2amdJoint& j = w.joints[(size_t)k];
		if (j.type < 0)
		{
			continue;
		}
		const bool rev = j.type == S2AMD_JOINT_REVOLUTE;
		const int jb = j.bodyB, ja = std::max(j.bodyA, 0);
		const int b = movable(w.bodies[(size_t)jb]) ? w.island[(size_t)jb] : -1;
		const int a = (rev && movable(w.bodies[(size_t)ja])) ? w.island[(size_t)ja] : -1;
		const int fallback = w.island[(size_t)jb] >= 0 ? w.island[(size_t)jb] : (rev ? w.island[(size_t)ja] : -1);
		ji[(size_t)k] = a >= 0 ? a : (b >= 0 ? b : fallback);
	}
}

// islands.py: partition -- longest-processing-time bin packing, deterministic (heaviest first, ties by index; least loaded shard, ties by index)
void partitionLpt(const std::vector<long long>& weights, const std::vector<int>& which, int nShards, std::vector<long long>& load, std::vector<int32_t>& shard)
{
	std::vector<int> order = which;
	std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return weights[(size_t)x] != weights[(size_t)y] ? weights[(size_t)x] > weights[(size_t)y] : x < y; });
	for (int i : order)
	{
		const int s = (int)(std::min_element(load.begin(), load.end()) - load.begin());
		shard[(size_t)i] = s;
		load[(size_t)s] += std::max<long long>(weights[(size_t)i], 1);
	}
	(void)nShards;
}

// islands.py: sticky_partition (previousOwner empty: the plain partition)
void stickyPartition(const s2amdShardedSolver& w, const std::vector<long long>& weights, const std::vector<int32_t>& previousOwner, int nShards,
					 std::vector<int32_t>& shard)
{
	const int n = w.islandCount;
	shard.assign((size_t)n, -1);
	std::vector<long long> load((size_t)nShards, 0);
	if (!previousOwner.empty())
	{
		std::vector<long long> votes((size_t)n * (size_t)nShards, 0);
		for (size_t b = 0; b < w.island.size(); ++b)
		{
			if (w.island[b] >= 0 && previousOwner[b] >= 0)
			{
				votes[(size_t)w.island[b] * (size_t)nShards + (size_t)previousOwner[b]] += 1;
			}
		}
		for (int i = 0; i < n; ++i)
		{
			long long best = 0;
			int at = -1;
			for (int s = 0; s < nShards; ++s)
			{
				if (votes[(size_t)i * (size_t)nShards + (size_t)s] > best) // (the lowest shard among equals)
				{
					best = votes[(size_t)i * (size_t)nShards + (size_t)s], at = s;
				}
			}
			if (at >= 0)
			{
				shard[(size_t)i] = at;
				load[(size_t)at] += std::max<long long>(weights[(size_t)i], 1);
			}
		}
	}
	std::vector<int> rest;
	for (int i = 0; i < n; ++i)
	{
		if (shard[(size_t)i] < 0)
		{
			rest.push_back(i);
		}
	}
	partitionLpt(weights, rest, nShards, load, shard);
	if (previousOwner.empty())
	{
		return;
	}
	// ... and past the threshold the heaviest shard's lightest islands that still help go to the least loaded shard
	for (int guard = 0; guard < n; ++guard)
	{
		const long long total = std::accumulate(load.begin(), load.end(), 0LL);
		const double mean = (double)total / (double)std::max(nShards, 1);
		const int heavy = (int)(std::max_element(load.begin(), load.end()) - load.begin());
		const int light = (int)(std::min_element(load.begin(), load.end()) - load.begin());
		if (nShards < 2 || mean <= 0.0 || (double)load[(size_t)heavy] <= kRebalanceThreshold * mean)
		{
			break;
		}
		int pick = -1;
		for (int i = 0; i < n; ++i)
		{
			const long long wi = std::max<long long>(weights[(size_t)i], 1);
			if (shard[(size_t)i] == heavy && load[(size_t)light] + wi < load[(size_t)heavy] &&
				(pick < 0 || wi < std::max<long long>(weights[(size_t)pick], 1)))
			{
				pick = i;
			}
		}
		if (pick < 0)
		{
			break;
		}
		const long long wp = std::max<long long>(weights[(size_t)pick], 1);
		shard[(size_t)pick] = light;
		load[(size_t)heavy] -= wp;
		load[(size_t)light] += wp;
	}
}

// islands.py: extract
void extract(const s2amdShardedSolver& w, const std::vector<int>& ci, const std::vector<int>& ji, int s, Shard& out)
{
	const int nb = (int)w.bodies.size();
	std::vector<uint8_t> own((size_t)nb, 0), used((size_t)nb, 0);
	for (int b = 0; b < nb; ++b)
	{
		own[(size_t)b] = w.island[(size_t)b] >= 0 && w.shardOfIsland[(size_t)w.island[(size_t)b]] == s;
		used[(size_t)b] = own[(size_t)b];
	}
	out.contactIds.clear(), out.jointIds.clear();
	for (int k = 0; k < (int)w.contacts.size(); ++k)
	{
		if (ci[(size_t)k] >= 0 && w.shardOfIsland[(size_t)ci[(size_t)k]] == s)
		{
			out.contactIds.push_back(k);
			used[(size_t)w.contacts[(size_t)k].bodyA] = 1;
			used[(size_t)w.contacts[(size_t)k].bodyB] = 1;
		}
	}
	for (int k = 0; k < (int)w.joints.size(); ++k)
	{
		if (ji[(size_t)k] >= 0 && w.shardOfIsland[(size_t)ji[(size_t)k]] == s)
		{
			out.jointIds.push_back(k);
			used[(size_t)w.joints[(size_t)k].bodyB] = 1;
			if (w.joints[(size_t)k].type == S2AMD_JOINT_REVOLUTE)
			{
				used[(size_t)w.joints[(size_t)k].bodyA] = 1;
			}
		}
	}
	std::vector<int> remap((size_t)nb, -1);
	out.bodyIds.clear(), out.bodies.clear(), out.owned.clear(), out.ownedLocal.clear(), out.ownedWorld.clear();
	for (int b = 0; b < nb; ++b) // ascending: pool order is preserved inside the shard
	{
		if (used[(size_t)b])
		{
			remap[(size_t)b] = (int)out.bodyIds.size();
			if (own[(size_t)b])
			{
				out.ownedLocal.push_back((int)out.bodyIds.size());
				out.ownedWorld.push_back(b);
			}
			out.bodyIds.push_back(b);
			out.bodies.push_back(w.bodies[(size_t)b]);
			out.owned.push_back(own[(size_t)b]);
		}
	}
	out.contacts.clear(), out.joints.clear();
	for (int k : out.contactIds)
	{
		s2amdContact c = w.contacts[(size_t)k];
		c.bodyA = remap[(size_t)c.bodyA], c.bodyB = remap[(size_t)c.bodyB];
		out.contacts.push_back(c);
	}
	for (int k : out.jointIds)
	{
		s2amdJoint j = w.joints[(size_t)k];
		j.bodyB = remap[(size_t)j.bodyB];
		if (j.bodyA >= 0)
		{
			j.bodyA = remap[(size_t)j.bodyA];
		}
		out.joints.push_back(j);
	}
}

int findIslands(s2amdShardedSolver* w)
{
	const int nb = (int)w->bodies.size();
	w->island.assign((size_t)nb, -1);
	int32_t count = 0;
	int rc = s2amd_find_islands(w->shards[0].solver, w->bodies.data(), nb, w->contacts.data(), (int32_t)w->contacts.size(), w->joints.data(),
								(int32_t)w->joints.size(), w->island.data(), &count);
	w->islandCount = count;
	for (int32_t x : w->island)
	{
		if (rc == S2AMD_OK && (x < -1 || x >= count))
		{
			rc = s2amdFail(S2AMD_E_DEVICE, "the device's island labels are out of range");
		}
	}
	return rc;
}

// body indices of the live constraints, before anything indexes w.bodies / w.island with them (s2amd_find_islands checks them too, but
// returns early on an empty body pool)
int checkConstraintBodies(int nb, const s2amdContact* contacts, int nc, const s2amdJoint* joints, int nj)
{
	for (int k = 0; k < nc; ++k)
	{
		const s2amdContact& c = contacts[k];
		if (c.pointCount > 0 && (c.bodyA < 0 || c.bodyA >= nb || c.bodyB < 0 || c.bodyB >= nb))
		{
			return s2amdFail(S2AMD_E_INVALID, "contact " + std::to_string(k) + " names a body outside the pool");
		}
	}
	for (int k = 0; k < nj; ++k)
	{
		const s2amdJoint& j = joints[k];
		if (j.type >= 0 && (j.bodyA < 0 || j.bodyA >= nb || j.bodyB < 0 || j.bodyB >= nb))
		{
			return s2amdFail(S2AMD_E_INVALID, "joint " + std::to_string(k) + " names a body outside the pool");
		}
	}
	return S2AMD_OK;
}

// what the chosen form of the exchange reads: every shard's table of world copies (stores), or the pool index of every row of the
// padded all-gather (rccl)
int buildExchangeTables(s2amdShardedSolver* w)
{
	const int nShards = (int)w->shards.size();
	int rc = S2AMD_OK;
	if (w->exchange == EX_STORES)
	{
		std::vector<float4*> worlds((size_t)nShards);
		for (int s = 0; s < nShards; ++s)
		{
			worlds[(size_t)s] = (float4*)w->shards[(size_t)s].dWorld.p;
		}
		for (Shard& sh : w->shards)
		{
			if ((rc = sh.dPeers.ensure(sh.device, (size_t)nShards * sizeof(float4*))) != 0)
			{
				return rc;
			}
			SH_TRY(hipSetDevice(sh.device));
			SH_TRY(hipMemcpy(sh.dPeers.p, worlds.data(), (size_t)nShards * sizeof(float4*), hipMemcpyHostToDevice));
		}
	}
	else if (w->exchange == EX_RCCL)
	{
		w->maxOwned = 1;
		for (const Shard& sh : w->shards)
		{
			w->maxOwned = std::max(w->maxOwned, sh.ownedWorld.size());
		}
		std::vector<int> ids((size_t)nShards * w->maxOwned, -1);
		for (int s = 0; s < nShards; ++s)
		{
			const Shard& src = w->shards[(size_t)s];
			std::copy(src.ownedWorld.begin(), src.ownedWorld.end(), ids.begin() + (size_t)s * w->maxOwned);
		}
		for (Shard& sh : w->shards)
		{
			if ((rc = sh.dGather.ensure(sh.device, ids.size() * 32)) != 0 || (rc = sh.dGatherIds.ensure(sh.device, ids.size() * sizeof(int))) != 0 ||
				(rc = sh.dCompact.ensure(sh.device, w->maxOwned * 32)) != 0)
			{
				return rc;
			}
			SH_TRY(hipSetDevice(sh.device));
			SH_TRY(hipMemcpy(sh.dGatherIds.p, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
			SH_TRY(hipMemset(sh.dCompact.p, 0, w->maxOwned * 32)); // (the padding rows travel: they are never scattered)
		}
	}
	return S2AMD_OK;
}

// partition (fresh, or sticky when previousOwner is given), extract, upload, exchange buffers
int partitionAndUpload(s2amdShardedSolver* w, const std::vector<int32_t>& previousOwner)
{
	int rc = findIslands(w);
	if (rc)
	{
		return rc;
	}
	const int nShards = (int)w->shards.size(), nb = (int)w->bodies.size();
	std::vector<int> ci, ji;
	constraintIslands(*w, ci, ji);
	std::vector<long long> weights((size_t)w->islandCount, 0);
	for (int x : ci)
	{
		if (x >= 0)
		{
			weights[(size_t)x] += 2;
		}
	}
	for (int x : ji)
	{
		if (x >= 0)
		{
			weights[(size_t)x] += 1;
		}
	}
	stickyPartition(*w, weights, previousOwner, nShards, w->shardOfIsland);
	for (int s = 0; s < nShards; ++s)
	{
		Shard& sh = w->shards[(size_t)s];
		extract(*w, ci, ji, s, sh);
		rc = s2amd_upload(sh.solver, sh.bodies.data(), (int32_t)sh.bodies.size(), sh.contacts.data(), (int32_t)sh.contacts.size(), sh.joints.data(),
						  (int32_t)sh.joints.size());
		if (rc)
		{
			return rc;
		}
		sh.uploaded = true;
		const size_t nOwned = sh.ownedLocal.size();
		if ((rc = sh.dRecords.ensure(sh.device, std::max<size_t>(sh.bodies.size(), 1) * 32)) != 0 ||
			(rc = sh.dOwnedLocal.ensure(sh.device, std::max<size_t>(nOwned, 1) * sizeof(int))) != 0 ||
			(rc = sh.dOwnedWorld.ensure(sh.device, std::max<size_t>(nOwned, 1) * sizeof(int))) != 0 ||
			(rc = sh.dCompact.ensure(sh.device, std::max<size_t>(nOwned, 1) * 32)) != 0 || (rc = sh.dWorld.ensure(sh.device, std::max<size_t>((size_t)nb, 1) * 32)) != 0)
		{
			return rc;
		}
		SH_TRY(hipSetDevice(sh.device));
		if (nOwned > 0)
		{
			SH_TRY(hipMemcpy(sh.dOwnedLocal.p, sh.ownedLocal.data(), nOwned * sizeof(int), hipMemcpyHostToDevice));
			SH_TRY(hipMemcpy(sh.dOwnedWorld.p, sh.ownedWorld.data(), nOwned * sizeof(int), hipMemcpyHostToDevice));
		}
		// the world's records as the upload has them (bodies nobody owns -- static ones -- keep these rows for good)
		std::vector<float> rows((size_t)nb * 8, 0.0f);
		for (int b = 0; b < nb; ++b)
		{
			const s2amdBody& x = w->bodies[(size_t)b];
			float* r = rows.data() + (size_t)b * 8;
			r[0] = x.position[0], r[1] = x.position[1], r[2] = x.rot[0], r[3] = x.rot[1];
			r[4] = x.linearVelocity[0], r[5] = x.linearVelocity[1], r[6] = x.angularVelocity, r[7] = 0.0f;
		}
		if (nb > 0)
		{
			SH_TRY(hipMemcpy(sh.dWorld.p, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
		}
	}
	// every shard's inbox for every other shard's compact run, and that run's pool indices, on the RECEIVER's device
	for (int t = 0; t < nShards; ++t)
	{
		Shard& dst = w->shards[(size_t)t];
		dst.inbox.resize((size_t)nShards), dst.inboxIds.resize((size_t)nShards);
		for (int s = 0; s < nShards; ++s)
		{
			if (s == t)
			{
				continue;
			}
			const Shard& src = w->shards[(size_t)s];
			const size_t n = src.ownedWorld.size();
			if ((rc = dst.inbox[(size_t)s].ensure(dst.device, std::max<size_t>(n, 1) * 32)) != 0 ||
				(rc = dst.inboxIds[(size_t)s].ensure(dst.device, std::max<size_t>(n, 1) * sizeof(int))) != 0)
			{
				return rc;
			}
			if (n > 0)
			{
				SH_TRY(hipSetDevice(dst.device));
				SH_TRY(hipMemcpy(dst.inboxIds[(size_t)s].p, src.ownedWorld.data(), n * sizeof(int), hipMemcpyHostToDevice));
			}
		}
	}
	if ((rc = buildExchangeTables(w)) != 0)
	{
		return rc;
	}
	w->outstanding.clear();
	w->resident = true;
	return S2AMD_OK;
}

// every shard's solver state back into its host sub-world, and from there into the world's arrays (islands.py: merge_back)
int pullShards(s2amdShardedSolver* w)
{
	for (Shard& sh : w->shards)
	{
		if (!sh.uploaded)
		{
			continue;
		}
		int rc = s2amd_synchronize(sh.solver);
		if (rc)
		{
			return rc;
		}
		rc = s2amd_download(sh.solver, sh.bodies.data(), (int32_t)sh.bodies.size(), sh.contacts.data(), (int32_t)sh.contacts.size(), sh.joints.data(),
							(int32_t)sh.joints.size());
		if (rc)
		{
			return rc;
		}
		for (size_t i = 0; i < sh.bodyIds.size(); ++i)
		{
			if (sh.owned[i])
			{
				w->bodies[(size_t)sh.bodyIds[i]] = sh.bodies[i];
			}
		}
		for (size_t i = 0; i < sh.contactIds.size(); ++i)
		{
			s2amdContact& dst = w->contacts[(size_t)sh.contactIds[i]];
			const int a = dst.bodyA, b = dst.bodyB, index = dst.constraintIndex;
			dst = sh.contacts[i];
			dst.bodyA = a, dst.bodyB = b, dst.constraintIndex = index; // (the gather index is a whole-world property: s2amd_sharded_download)
		}
		for (size_t i = 0; i < sh.jointIds.size(); ++i)
		{
			s2amdJoint& dst = w->joints[(size_t)sh.jointIds[i]];
			const int a = dst.bodyA, b = dst.bodyB;
			dst = sh.joints[i];
			dst.bodyA = a, dst.bodyB = b;
		}
	}
	return S2AMD_OK;
}

dim3 blocksFor(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

} // namespace

#pragma GCC visibility push(default)
extern "C"
{

int s2amd_sharded_create(const int32_t* devices, int32_t deviceCount, s2amdShardedSolver** out)
{
	if (!devices || deviceCount <= 0 || deviceCount > 64 || !out)
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument (1 to 64 shards)");
	}
	*out = nullptr;
	s2amdShardedSolver* w = new s2amdShardedSolver();
	w->shards.resize((size_t)deviceCount);
	for (int i = 0; i < deviceCount; ++i)
	{
		Shard& sh = w->shards[(size_t)i];
		sh.device = devices[i];
		int rc = s2amd_create(devices[i], &sh.solver);
		if (rc == S2AMD_OK)
		{
			// the steps of all shards are enqueued before any is waited for
			rc = s2amd_set_option(sh.solver, "async", 1);
		}
		if (rc == S2AMD_OK && (hipSetDevice(sh.device) != hipSuccess || hipEventCreateWithFlags(&sh.evCompact, hipEventDisableTiming) != hipSuccess ||
								hipEventCreateWithFlags(&sh.evStep, hipEventDisableTiming) != hipSuccess ||
								hipEventCreateWithFlags(&sh.evXchg, hipEventDisableTiming) != hipSuccess))
		{
			rc = s2amdFail(S2AMD_E_DEVICE, "could not create the shard's exchange stream and events");
		}
		if (rc)
		{
			s2amd_sharded_destroy(w);
			return rc;
		}
	}
	// peers that can reach each other directly (xGMI, PCIe P2P) are enabled once; a pair that cannot is still served by
	// hipMemcpyPeerAsync, through the host
	for (int i = 0; i < deviceCount; ++i)
	{
		for (int j = 0; j < deviceCount; ++j)
		{
			int can = 0;
			if (devices[i] != devices[j] && hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can)
			{
				(void)hipSetDevice(devices[i]);
				(void)hipDeviceEnablePeerAccess(devices[j], 0);
			}
			(void)hipGetLastError(); // (already enabled: not an error)
		}
	}
	// how the step's exchange travels: stores between shards of one device, RCCL between distinct devices, peer copies for the rest
	// (S2AMD_SHARDED_EXCHANGE=stores|rccl|copies overrides: tests, and a node whose RCCL misbehaves)
	{
		bool allSame = true, allDistinct = true;
		for (int i = 0; i < deviceCount; ++i)
		{
			for (int j = i + 1; j < deviceCount; ++j)
			{
				allSame = allSame && devices[i] == devices[j];
				allDistinct = allDistinct && devices[i] != devices[j];
			}
		}
		const char* forced = getenv("S2AMD_SHARDED_EXCHANGE");
		w->exchange = allSame ? EX_STORES : (allDistinct ? EX_RCCL : EX_COPIES);
		if (forced != nullptr && strcmp(forced, "copies") == 0)
		{
			w->exchange = EX_COPIES;
		}
		else if (forced != nullptr && strcmp(forced, "rccl") == 0 && allDistinct)
		{
			w->exchange = EX_RCCL; // (one rank per device: shards that share a device cannot be RCCL ranks)
		}
		else if (forced != nullptr && strcmp(forced, "stores") == 0 && allSame)
		{
			w->exchange = EX_STORES;
		}
		if (w->exchange == EX_RCCL)
		{
			std::vector<void*> comms((size_t)deviceCount, nullptr);
			std::vector<int> devs(devices, devices + deviceCount);
			const bool loaded = g_rccl.load();
			const int rcInit = loaded ? g_rccl.commInitAll(comms.data(), deviceCount, devs.data()) : -1;
			if (!loaded || rcInit != 0)
			{
				w->exchange = EX_COPIES; // (no librccl.so, or it would not initialise: the peer copies)
				if (getenv("S2AMD_DEBUG") != nullptr)
				{
					fprintf(stderr, "[s2amd] sharded exchange: RCCL %s (%s): peer copies instead\n", loaded ? "would not initialise" : "could not be loaded",
							loaded ? (g_rccl.errorString ? g_rccl.errorString(rcInit) : "?") : (dlerror() ? dlerror() : "dlopen"));
				}
			}
			else
			{
				for (int i = 0; i < deviceCount; ++i)
				{
					w->shards[(size_t)i].comm = comms[(size_t)i];
				}
			}
		}
	}
	// the exchange's own streams, where it has them (shards of one device send their rows on the solve stream -- and a stream more per
	// shard moved their solve streams onto ONE hardware queue of the four: 4 logical shards took 0.76 ms a step instead of 0.50)
	for (Shard& sh : w->shards)
	{
		if (w->exchange != EX_STORES && (hipSetDevice(sh.device) != hipSuccess || hipStreamCreateWithFlags(&sh.xs, hipStreamNonBlocking) != hipSuccess))
		{
			s2amd_sharded_destroy(w);
			return s2amdFail(S2AMD_E_DEVICE, "could not create the shard's exchange stream");
		}
	}
	*out = w;
	return S2AMD_OK;
}

void s2amd_sharded_destroy(s2amdShardedSolver* w)
{
	if (!w)
	{
		return;
	}
	for (Shard& sh : w->shards)
	{
		if (sh.solver)
		{
			(void)s2amd_synchronize(sh.solver);
		}
		if (sh.xs)
		{
			(void)hipSetDevice(sh.device);
			(void)hipStreamSynchronize(sh.xs);
		}
		if (sh.comm && g_rccl.commDestroy)
		{
			(void)g_rccl.commDestroy(sh.comm);
			sh.comm = nullptr;
		}
		DeviceBlock* blocks[] = {&sh.dRecords, &sh.dOwnedLocal, &sh.dOwnedWorld, &sh.dCompact, &sh.dWorld, &sh.dPeers, &sh.dGather, &sh.dGatherIds};
		for (DeviceBlock* b : blocks)
		{
			b->release();
		}
		for (DeviceBlock& b : sh.inbox)
		{
			b.release();
		}
		for (DeviceBlock& b : sh.inboxIds)
		{
			b.release();
		}
		if (sh.evCompact)
		{
			(void)hipSetDevice(sh.device);
			(void)hipEventDestroy(sh.evCompact);
		}
		if (sh.evStep)
		{
			(void)hipEventDestroy(sh.evStep);
		}
		if (sh.evXchg)
		{
			(void)hipEventDestroy(sh.evXchg);
		}
		if (sh.xs)
		{
			(void)hipStreamDestroy(sh.xs);
		}
		if (sh.solver)
		{
			s2amd_destroy(sh.solver);
		}
	}
	delete w;
}

int s2amd_sharded_shard_count(const s2amdShardedSolver* w) { return w ? (int)w->shards.size() : 0; }

s2amdSolver* s2amd_sharded_solver(s2amdShardedSolver* w, int32_t shard)
{
	return (w && shard >= 0 && shard < (int32_t)w->shards.size()) ? w->shards[(size_t)shard].solver : nullptr;
}

int s2amd_sharded_upload(s2amdShardedSolver* w, const s2amdBody* bodies, int32_t bodyCapacity, const s2amdContact* contacts, int32_t contactCapacity,
						 const s2amdJoint* joints, int32_t jointCapacity)
{
	if (!w || bodyCapacity < 0 || contactCapacity < 0 || jointCapacity < 0 || (bodyCapacity > 0 && !bodies) || (contactCapacity > 0 && !contacts) ||
		(jointCapacity > 0 && !joints))
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument");
	}
	int rc = checkConstraintBodies(bodyCapacity, contacts, contactCapacity, joints, jointCapacity);
	if (rc)
	{
		return rc;
	}
	if (w->resident && (rc = s2amd_sharded_wait(w)) != 0)
	{
		return rc;
	}
	w->bodies.assign(bodies, bodies + bodyCapacity);
	w->contacts.assign(contacts, contacts + contactCapacity);
	w->joints.assign(joints, joints + jointCapacity);
	w->resident = false;
	const std::vector<int32_t> none;
	return partitionAndUpload(w, none);
}

// one shard's step and its part of the exchange, enqueued: the solve on the solver's stream, the rows on the exchange stream behind it
static int enqueueShard(s2amdShardedSolver* w, int index, const s2amdStepParams* params, int32_t* ops)
{
	Shard& sh = w->shards[(size_t)index];
	const bool dry = w->dry;
	const int nShards = (int)w->shards.size();
	const size_t n = sh.ownedLocal.size();
	hipStream_t st = dry ? nullptr : s2amdStream(sh.solver);
	// Shards of one device gain nothing from a second stream each (the device is the same one, and more streams than hardware queues
	// serialise in ways nobody asked for: 4 shards took 0.78 ms that way against 0.50): their rows go out on the solve stream itself.
	const bool beside = w->exchange != EX_STORES;
	hipStream_t xs = dry ? nullptr : (beside ? sh.xs : st);
	if (!dry)
	{
		SH_TRY(hipSetDevice(sh.device));
		int rc = s2amd_step_resident(sh.solver, params);
		if (rc == S2AMD_OK && beside)
		{
			// (the last exchange has read the records this step's export rewrites; the solve itself did not have to wait for it)
			SH_TRY(hipStreamWaitEvent(st, sh.evXchg, 0));
		}
		if (rc == S2AMD_OK && beside)
		{
			rc = s2amd_export_bodies_async(sh.solver, sh.dRecords.p, (int32_t)sh.bodies.size(), 0);
		}
		if (rc)
		{
			return rc;
		}
		if (beside)
		{
			SH_TRY(hipEventRecord(sh.evStep, st));
			SH_TRY(hipStreamWaitEvent(sh.xs, sh.evStep, 0));
		}
	}
	*ops += beside ? 5 : 1; // step (+ wait, export, record, wait)
	if (w->exchange == EX_STORES)
	{
		if (!dry && n > 0)
		{
			pushOwnedBodiesKernel<<<blocksFor(n), dim3(256), 0, xs>>>(s2amdResidentBodies(sh.solver), (const int*)sh.dOwnedLocal.p, (const int*)sh.dOwnedWorld.p, (int)n,
																	  (float4* const*)sh.dPeers.p, nShards);
			SH_TRY(hipGetLastError());
		}
		*ops += 1;
	}
	else
	{
		if (!dry && n > 0)
		{
			compactOwnedKernel<<<blocksFor(n), dim3(256), 0, sh.xs>>>((const float4*)sh.dRecords.p, (const int*)sh.dOwnedLocal.p, (int)n, (float4*)sh.dCompact.p);
			SH_TRY(hipGetLastError());
		}
		*ops += 1;
		if (w->exchange == EX_COPIES)
		{
			if (!dry && n > 0)
			{
				scatterOwnedKernel<<<blocksFor(n), dim3(256), 0, sh.xs>>>((const float4*)sh.dCompact.p, (const int*)sh.dOwnedWorld.p, (int)n, (float4*)sh.dWorld.p);
				SH_TRY(hipGetLastError());
			}
			if (!dry)
			{
				SH_TRY(hipEventRecord(sh.evCompact, sh.xs));
			}
			*ops += 2;
		}
	}
	return S2AMD_OK;
}

// the rest of the exchange, which involves more than one shard: the collective, or the peer copies; every exchange stream ends in its event
static int enqueueExchange(s2amdShardedSolver* w, int only, int32_t* ops)
{
	const bool dry = w->dry;
	const int nShards = (int)w->shards.size();
	if (w->exchange == EX_RCCL)
	{
		// one all-gather per device in one group: every device receives every rank's padded run (only == -1: a repeated shard's rows
		// travel with everybody's, the others' being what they were)
		if (!dry)
		{
			if (g_rccl.groupStart() != 0)
			{
				return s2amdFail(S2AMD_E_DEVICE, "ncclGroupStart failed");
			}
			for (Shard& sh : w->shards)
			{
				const int rcN = g_rccl.allGather(sh.dCompact.p, sh.dGather.p, w->maxOwned * 8, kNcclFloat, sh.comm, sh.xs);
				if (rcN != 0)
				{
					(void)g_rccl.groupEnd();
					return s2amdFail(S2AMD_E_DEVICE, std::string("ncclAllGather failed: ") + (g_rccl.errorString ? g_rccl.errorString(rcN) : "?"));
				}
			}
			if (g_rccl.groupEnd() != 0)
			{
				return s2amdFail(S2AMD_E_DEVICE, "ncclGroupEnd failed");
			}
			for (Shard& sh : w->shards)
			{
				const size_t total = (size_t)nShards * w->maxOwned;
				SH_TRY(hipSetDevice(sh.device));
				scatterGatheredKernel<<<blocksFor(total), dim3(256), 0, sh.xs>>>((const float4*)sh.dGather.p, (const int*)sh.dGatherIds.p, (int)total, (float4*)sh.dWorld.p);
				SH_TRY(hipGetLastError());
			}
		}
		*ops += 2 * nShards;
	}
	else if (w->exchange == EX_COPIES)
	{
		for (int t = 0; t < nShards; ++t)
		{
			Shard& dst = w->shards[(size_t)t];
			if (!dry)
			{
				SH_TRY(hipSetDevice(dst.device));
			}
			for (int s = 0; s < nShards; ++s)
			{
				const Shard& src = w->shards[(size_t)s];
				const size_t n = src.ownedLocal.size();
				if (s == t || n == 0 || (only >= 0 && s != only))
				{
					continue;
				}
				if (!dry)
				{
					SH_TRY(hipStreamWaitEvent(dst.xs, src.evCompact, 0));
					SH_TRY(hipMemcpyPeerAsync(dst.inbox[(size_t)s].p, dst.device, src.dCompact.p, src.device, n * 32, dst.xs));
					scatterOwnedKernel<<<blocksFor(n), dim3(256), 0, dst.xs>>>((const float4*)dst.inbox[(size_t)s].p, (const int*)dst.inboxIds[(size_t)s].p, (int)n,
																			   (float4*)dst.dWorld.p);
					SH_TRY(hipGetLastError());
				}
				*ops += 3;
			}
		}
	}
	for (Shard& sh : w->shards)
	{
		if (!dry)
		{
			SH_TRY(hipSetDevice(sh.device));
			SH_TRY(hipEventRecord(sh.evXchg, w->exchange == EX_STORES ? s2amdStream(sh.solver) : sh.xs));
		}
		*ops += 1;
	}
	return S2AMD_OK;
}

int s2amd_sharded_step_async(s2amdShardedSolver* w, const s2amdStepParams* params)
{
	if (!w || !params)
	{
		return s2amdFail(S2AMD_E_INVALID, "null argument");
	}
	if (!w->resident && !w->dry)
	{
		return s2amdFail(S2AMD_E_STATE, "s2amd_sharded_step called before s2amd_sharded_upload");
	}
	const int nShards = (int)w->shards.size();
	int32_t ops = 0;
	for (int i = 0; i < nShards; ++i)
	{
		const int rc = enqueueShard(w, i, params, &ops);
		if (rc)
		{
			// (earlier shards have this step enqueued and this one has not: the world's shards disagree from here on)
			w->resident = false;
			return rc;
		}
	}
	const int rc = enqueueExchange(w, -1, &ops);
	if (rc)
	{
		w->resident = false;
		return rc;
	}
	w->opsLastStep = ops;
	w->hostWaitsLastStep = 0;
	w->outstanding.push_back(*params);
	w->steps += 1;
	w->lastSolverType = params->solverType;
	return S2AMD_OK;
}

// The one place the host waits: shard 0's exchange stream takes every other shard's last event, and the host takes that stream.  A shard
// whose persistent launch lost a hand-off (another process on its GPU) has left its arrays untouched and keeps to the multi-launch path
// from now on: its step is repeated and its rows exchanged again -- possible when one step is outstanding; with more, the steps
// behind the failing one were dropped too and the world is marked torn (upload it again).
int s2amd_sharded_wait(s2amdShardedSolver* w)
{
	if (!w)
	{
		return s2amdFail(S2AMD_E_INVALID, "null solver");
	}
	if (w->outstanding.empty() || w->shards.empty())
	{
		return S2AMD_OK;
	}
	const int nShards = (int)w->shards.size();
	Shard& first = w->shards[0];
	hipStream_t join = first.xs != nullptr ? first.xs : s2amdStream(first.solver);
	SH_TRY(hipSetDevice(first.device));
	for (int t = 0; t < nShards; ++t)
	{
		if (t > 0 || first.xs != nullptr)
		{
			SH_TRY(hipStreamWaitEvent(join, w->shards[(size_t)t].evXchg, 0));
		}
	}
	SH_TRY(hipStreamSynchronize(join));
	w->hostWaitsLastStep += 1;
	int result = S2AMD_OK;
	for (int t = 0; t < nShards && result == S2AMD_OK; ++t)
	{
		Shard& sh = w->shards[(size_t)t];
		if (!s2amdStepFailed(sh.solver))
		{
			continue; // (the error word is host-visible: nothing to ask the device)
		}
		int rc = s2amd_synchronize(sh.solver); // (clears the error words, drops the solver to its fall-back path)
		if (rc != S2AMD_E_DEVICE)
		{
			result = rc != S2AMD_OK ? rc : s2amdFail(S2AMD_E_DEVICE, "a shard reported a failed step and none on inspection");
			break;
		}
		if (w->outstanding.size() > 1)
		{
			w->resident = false;
			result = s2amdFail(S2AMD_E_DEVICE, "shard " + std::to_string(t) + " lost a hand-off with " + std::to_string(w->outstanding.size()) +
													" steps outstanding: the steps behind the failing one were dropped on that shard only; upload the world again "
													"(or wait after every step: s2amd_sharded_step)");
			break;
		}
		const s2amdStepParams params = w->outstanding.back();
		// (near hand-offs off, then the overflow workgroup off, then the persistent kernels off: three repeats can be legitimate)
		for (int attempt = 0; rc == S2AMD_E_DEVICE && attempt < 4; ++attempt)
		{
			w->repeatedSteps += 1;
			int32_t ops = 0;
			rc = enqueueShard(w, t, &params, &ops);
			if (rc == S2AMD_OK)
			{
				rc = enqueueExchange(w, t, &ops);
			}
			if (rc == S2AMD_OK)
			{
				for (Shard& other : w->shards)
				{
					SH_TRY(hipSetDevice(other.device));
					SH_TRY(hipStreamSynchronize(other.xs != nullptr ? other.xs : s2amdStream(other.solver)));
					w->hostWaitsLastStep += 1;
				}
				rc = s2amdStepFailed(sh.solver) ? s2amd_synchronize(sh.solver) : S2AMD_OK;
			}
		}
		if (rc)
		{
			w->resident = false; // (the shard could not complete the step the others have: torn)
			result = rc;
		}
	}
	w->outstanding.clear();
	return result;
}

int s2amd_sharded_step(s2amdShardedSolver* w, const s2amdStepParams* params)
{
	const int rc = s2amd_sharded_step_async(w, params);
	return rc != S2AMD_OK ? rc : s2amd_sharded_wait(w);
}

// {stream operations, host waits} of the last s2amd_sharded_step_async (+ the waits of the s2amd_sharded_wait behind it), and how the
// exchange travels (0 stores, 1 rccl, 2 peer copies)
int s2amd_sharded_get_step_ops(s2amdShardedSolver* w, int32_t* streamOps, int32_t* hostWaits, int32_t* exchange)
{
	if (!w)
	{
		return s2amdFail(S2AMD_E_INVALID, "null solver");
	}
	if (streamOps)
	{
		*streamOps = w->opsLastStep;
	}
	if (hostWaits)
	{
		*hostWaits = w->hostWaitsLastStep;
	}
	if (exchange)
	{
		*exchange = w->exchange;
	}
	return S2AMD_OK;
}

// what one step of `shards` shards enqueues in the given form of the exchange, counted by the code that enqueues it (nothing is
// enqueued: callable without a device)
int s2amd_sharded_count_ops(int32_t shards, int32_t exchange, int32_t* streamOps)
{
	if (shards <= 0 || shards > 64 || exchange < 0 || exchange > 2 || !streamOps)
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument");
	}
	s2amdShardedSolver w;
	w.shards.resize((size_t)shards);
	for (Shard& sh : w.shards)
	{
		sh.ownedLocal.assign(1, 0); // (every shard owns something)
	}
	w.exchange = exchange;
	w.dry = true;
	s2amdStepParams params{};
	const int rc = s2amd_sharded_step_async(&w, &params);
	*streamOps = w.opsLastStep;
	return rc;
}

int s2amd_sharded_read_bodies(s2amdShardedSolver* w, int32_t shard, float* out, int32_t bodyCapacity)
{
	if (!w || !out || shard < 0 || shard >= (int32_t)w->shards.size())
	{
		return s2amdFail(S2AMD_E_INVALID, "bad argument");
	}
	if (!w->resident)
	{
		return s2amdFail(S2AMD_E_STATE, "nothing resident");
	}
	if (bodyCapacity < (int32_t)w->bodies.size())
	{
		return s2amdFail(S2AMD_E_CAPACITY, "body record buffer too small");
	}
	{
		const int rcWait = s2amd_sharded_wait(w);
		if (rcWait)
		{
			return rcWait;
		}
	}
	Shard& sh = w->shards[(size_t)shard];
	SH_TRY(hipSetDevice(sh.device));
	SH_TRY(hipStreamSynchronize(sh.xs != nullptr ? sh.xs : s2amdStream(sh.solver)));
	if (!w->bodies.empty())
	{
		SH_TRY(hipMemcpy(out, sh.dWorld.p, w->bodies.size() * 32, hipMemcpyDeviceToHost));
	}
	return S2AMD_OK;
}

int s2amd_sharded_download(s2amdShardedSolver* w, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts, int32_t contactCapacity,
						   s2amdJoint* joints, int32_t jointCapacity)
{
	if (!w)
	{
		return s2amdFail(S2AMD_E_INVALID, "null solver");
	}
	if (!w->resident)
	{
		return s2amdFail(S2AMD_E_STATE, "nothing resident");
	}
	if ((bodies && bodyCapacity < (int32_t)w->bodies.size()) || (contacts && contactCapacity < (int32_t)w->contacts.size()) ||
		(joints && jointCapacity < (int32_t)w->joints.size()))
	{
		return s2amdFail(S2AMD_E_CAPACITY, "output arrays smaller than the resident world");
	}
	int rc = s2amd_sharded_wait(w);
	if (rc == S2AMD_OK)
	{
		rc = pullShards(w);
	}
	if (rc)
	{
		return rc;
	}
	// manifold.constraintIndex as the reference's gather loop over the WHOLE pool writes it (e.g. solve_tgs_soft.c:162-179): the rank
	// among the contacts with manifold points, -1 for the skipped ones
	// (every driver but s2Solve_PGS_NGS_Block, whose own constraint array has no such field: solve_pgs_ngs_block.c:892-963)
	if (w->lastSolverType >= 0 && w->lastSolverType != s2amd_solverPGS_NGS_Block)
	{
		int rank = 0;
		for (s2amdContact& c : w->contacts)
		{
			c.constraintIndex = c.pointCount > 0 ? rank++ : -1;
		}
	}
	if (bodies)
	{
		std::copy(w->bodies.begin(), w->bodies.end(), bodies);
	}
	if (contacts)
	{
		std::copy(w->contacts.begin(), w->contacts.end(), contacts);
	}
	if (joints)
	{
		std::copy(w->joints.begin(), w->joints.end(), joints);
	}
	return S2AMD_OK;
}

int s2amd_sharded_reshard(s2amdShardedSolver* w, const s2amdContact* contacts, int32_t contactCapacity, const s2amdJoint* joints, int32_t jointCapacity)
{
	if (!w)
	{
		return s2amdFail(S2AMD_E_INVALID, "null solver");
	}
	if (!w->resident)
	{
		return s2amdFail(S2AMD_E_STATE, "nothing resident");
	}
	if ((contacts && contactCapacity != (int32_t)w->contacts.size()) || (joints && jointCapacity != (int32_t)w->joints.size()))
	{
		return s2amdFail(S2AMD_E_INVALID, "the new constraint arrays must have the pool capacities of the uploaded world");
	}
	// (validated before anything is touched: a failed reshard leaves the solver as it was)
	int rc = checkConstraintBodies((int)w->bodies.size(), contacts ? contacts : w->contacts.data(), (int)w->contacts.size(), joints ? joints : w->joints.data(),
								   (int)w->joints.size());
	if (rc == S2AMD_OK)
	{
		rc = s2amd_sharded_wait(w);
	}
	if (rc == S2AMD_OK)
	{
		rc = pullShards(w); // bodies, impulses, TGS_Sticky's friction cache: the solver state lives with the owner
	}
	if (rc)
	{
		return rc;
	}
	std::vector<int32_t> previous(w->bodies.size(), -1);
	for (size_t s = 0; s < w->shards.size(); ++s)
	{
		for (int b : w->shards[s].ownedWorld)
		{
			previous[(size_t)b] = (int32_t)s;
		}
	}
	if (contacts)
	{
		// the caller's array decides which slots are live and what their manifolds are; a slot that kept its pair keeps its solver state
		for (int k = 0; k < contactCapacity; ++k)
		{
			s2amdContact c = contacts[k];
			const s2amdContact& old = w->contacts[(size_t)k];
			if (c.bodyA == old.bodyA && c.bodyB == old.bodyB && old.bodyA >= 0)
			{
				for (int j = 0; j < 2; ++j)
				{
					c.points[j].normalImpulse = old.points[j].normalImpulse, c.points[j].tangentImpulse = old.points[j].tangentImpulse;
					memcpy(c.points[j].frictionAnchorA, old.points[j].frictionAnchorA, sizeof(float) * 2);
					memcpy(c.points[j].frictionAnchorB, old.points[j].frictionAnchorB, sizeof(float) * 2);
					memcpy(c.points[j].frictionNormalA, old.points[j].frictionNormalA, sizeof(float) * 2);
					memcpy(c.points[j].frictionNormalB, old.points[j].frictionNormalB, sizeof(float) * 2);
				}
				c.frictionPersisted = old.frictionPersisted;
			}
			w->contacts[(size_t)k] = c;
		}
	}
	if (joints)
	{
		w->joints.assign(joints, joints + jointCapacity);
	}
	w->reshards += 1;
	return partitionAndUpload(w, previous);
}

int s2amd_sharded_get_partition(s2amdShardedSolver* w, int32_t* shardOfBody, int32_t capacity, int32_t* islandCount, int32_t* reshards)
{
	if (!w)
	{
		return s2amdFail(S2AMD_E_INVALID, "null solver");
	}
	if (shardOfBody && capacity < (int32_t)w->bodies.size())
	{
		return s2amdFail(S2AMD_E_CAPACITY, "partition buffer too small");
	}
	if (shardOfBody)
	{
		for (size_t b = 0; b < w->bodies.size(); ++b)
		{
			shardOfBody[b] = (b < w->island.size() && w->island[b] >= 0) ? w->shardOfIsland[(size_t)w->island[b]] : -1;
		}
	}
	if (islandCount)
	{
		*islandCount = w->islandCount;
	}
	if (reshards)
	{
		*reshards = w->reshards;
	}
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop

S2_DEFINE_WARM(sharded)
