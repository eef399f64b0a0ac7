// Host tables of s2Solve_Jacobi's persistent launch (jacobi_kernel.hip: jacobiStepKernel; BASELINE.json configs[2]).
//
// The bodies the step integrates -- every live non-static body -- are dealt to BLOCKS: chunks of a breadth-first order of the
// constraint graph (so that a block's neighbours in the graph are mostly its own bodies), at most S2_JACOBI_BLOCK_BODIES bodies and
// S2_JACOBI_BLOCK_CONSTRAINTS constraints each.  A block holds EVERY constraint that touches a body it owns; the other body of such a
// constraint is an import when another block owns it (or nobody does: a static body, loaded once).  Per block, for the kernel:
// owned and imported pool slots, the constraints with their local body slots, the owned bodies' incidence lists in pool order (the
// order the reference adds a body's deltas in, solve_jacobi.c:126-130), the bodies other blocks import, the bodies with long lists,
// the joints whose bodies the block owns.  A hub -- the Tumbler's drum -- is a body like any other here: its block holds all of
// its constraints and imports their boxes; BFS does not expand through it (else everything it touches would be one level).
//
// Built with the structure of a s2Solve_Jacobi step (solver_structure.cpp: finish) when the world qualifies: enough constraints to
// be worth a persistent launch, every joint inside one block, no more blocks than compute units (the workgroups wait for each
// other: they must be co-resident), no body with more constraints than a block takes.
#include "solver_internal.h"

#include <queue>

#define S2_JACOBI_BLOCK_BODIES 192		// at most; the build aims at one block per compute unit (below)
#define S2_JACOBI_BLOCK_CONSTRAINTS 960 // (the kernel holds two per lane: 1024)
#define S2_JACOBI_LANE_CONSTRAINTS 480	// ... and a block that stays below 512 runs the one-record kernel: the aim
#define S2_JACOBI_BLOCK_IMPORTS 1024
#define S2_JACOBI_HUB_DEGREE 64 // BFS does not expand through a body with more constraints

int buildJacobiBlocks(s2amdSolver* s)
{
	s->jacobiValid = false;
	s->jacobi = JacobiView{};
	const SweepSet& cs = s->contacts;
	const SweepSet& js = s->joints;
	const int nb = s->bodyCapacity, P = (int)cs.order.size();
	if (s->optJacobiPersist == 0 || s->hostError == nullptr || !s->orderColourless || nb <= 0)
	{
		return S2AMD_OK;
	}
	int live = 0;
	for (int k = 0; k < P; ++k)
	{
		live += cs.order[(size_t)k] >= 0 ? 1 : 0;
	}
	if (live < s->optJacobiMinConstraints)
	{
		return S2AMD_OK;
	}
	auto ownedBody = [&](int b) { return b >= 0 && b < nb && s->hBodyLive[(size_t)b] && !s->hBodyStatic[(size_t)b]; };

	// ---- incidence of every body: positions in ascending order ----
	std::vector<int> degree((size_t)nb + 1, 0);
	for (int k = 0; k < P; ++k)
	{
		const int slot = cs.order[(size_t)k];
		if (slot >= 0)
		{
			degree[(size_t)s->hContactA[(size_t)slot]] += 1;
			degree[(size_t)s->hContactB[(size_t)slot]] += 1;
		}
	}
	std::vector<int> first((size_t)nb + 1, 0);
	for (int b = 0; b < nb; ++b)
	{
		first[(size_t)b + 1] = first[(size_t)b] + degree[(size_t)b];
	}
	std::vector<int> incident((size_t)first[(size_t)nb]), cursor(first.begin(), first.end() - 1);
	for (int k = 0; k < P; ++k)
	{
		const int slot = cs.order[(size_t)k];
		if (slot >= 0)
		{
			incident[(size_t)cursor[(size_t)s->hContactA[(size_t)slot]]++] = k;
			incident[(size_t)cursor[(size_t)s->hContactB[(size_t)slot]]++] = k;
		}
	}
	for (int b = 0; b < nb; ++b)
	{
		if (ownedBody(b) && degree[(size_t)b] > S2_JACOBI_BLOCK_CONSTRAINTS)
		{
			return S2AMD_OK; // (a body with more constraints than a block takes: the multi-launch path)
		}
	}

	// ---- breadth-first order of the owned bodies (not expanding through hubs) ----
	std::vector<int> order;
	order.reserve((size_t)nb);
	std::vector<uint8_t> seen((size_t)nb, 0);
	std::vector<int> queue;
	for (int root = 0; root < nb; ++root)
	{
		if (!ownedBody(root) || seen[(size_t)root])
		{
			continue;
		}
		queue.clear();
		queue.push_back(root);
		seen[(size_t)root] = 1;
		for (size_t head = 0; head < queue.size(); ++head)
		{
			const int u = queue[head];
			order.push_back(u);
			if (degree[(size_t)u] > S2_JACOBI_HUB_DEGREE)
			{
				continue;
			}
			for (int e = first[(size_t)u]; e < first[(size_t)u + 1]; ++e)
			{
				const int slot = cs.order[(size_t)incident[(size_t)e]];
				const int v = s->hContactA[(size_t)slot] == u ? s->hContactB[(size_t)slot] : s->hContactA[(size_t)slot];
				if (ownedBody(v) && !seen[(size_t)v])
				{
					seen[(size_t)v] = 1;
					queue.push_back(v);
				}
			}
		}
	}

	// ---- chunks: as many blocks as the device has compute units to spare (an iteration's time is its slowest block's: one constraint
	// per lane where the world is small enough), a hub alone in its block (every block that touches it waits for its sums) ----
	std::vector<int> blockOf((size_t)nb, -1), slotOf((size_t)nb, -1);
	std::vector<std::vector<int>> owned;
	{
		const int target = std::max(s->cuCount - 8, 8);
		const int bodiesPerBlock = std::min(std::max(((int)order.size() + target - 1) / target, 24), S2_JACOBI_BLOCK_BODIES);
		const int constraintsPerBlock = bodiesPerBlock < S2_JACOBI_BLOCK_BODIES ? S2_JACOBI_LANE_CONSTRAINTS : S2_JACOBI_BLOCK_CONSTRAINTS;
		std::vector<int> stamp((size_t)P, -1);
		int constraints = 0;
		bool hubBlock = false;
		owned.emplace_back();
		for (int u : order)
		{
			int fresh = 0;
			for (int e = first[(size_t)u]; e < first[(size_t)u + 1]; ++e)
			{
				fresh += stamp[(size_t)incident[(size_t)e]] != (int)owned.size() - 1 ? 1 : 0;
			}
			const bool hub = degree[(size_t)u] > S2_JACOBI_HUB_DEGREE;
			if (!owned.back().empty() && (hub || hubBlock || (int)owned.back().size() >= bodiesPerBlock || constraints + fresh > constraintsPerBlock))
			{
				owned.emplace_back();
				constraints = 0;
				fresh = degree[(size_t)u];
			}
			for (int e = first[(size_t)u]; e < first[(size_t)u + 1]; ++e)
			{
				stamp[(size_t)incident[(size_t)e]] = (int)owned.size() - 1;
			}
			constraints += fresh;
			hubBlock = hub;
			blockOf[(size_t)u] = (int)owned.size() - 1;
			slotOf[(size_t)u] = (int)owned.back().size();
			owned.back().push_back(u);
		}
	}
	const int B = (int)owned.size();
	if (B == 0 || owned[0].empty() || B > std::max(s->cuCount, 1))
	{
		return S2AMD_OK; // (the workgroups wait for each other: one per compute unit at most)
	}

	// ---- joints: each inside one block ----
	std::vector<std::vector<int>> jointsOf((size_t)B);
	for (int p = 0; p < (int)js.order.size(); ++p)
	{
		const int j = js.order[(size_t)p];
		if (j < 0)
		{
			continue;
		}
		const int a = s->hJointA[(size_t)j], b = s->hJointB[(size_t)j];
		const int ba = ownedBody(a) ? blockOf[(size_t)a] : -1, bb = ownedBody(b) ? blockOf[(size_t)b] : -1;
		if (ba >= 0 && bb >= 0 && ba != bb)
		{
			return S2AMD_OK; // a joint across two blocks: joints are sequential (solve_jacobi.c:211-221), the multi-launch path sweeps them
		}
		const int home = ba >= 0 ? ba : bb;
		if (home < 0)
		{
			continue; // (between bodies nothing integrates: it moves nothing)
		}
		jointsOf[(size_t)home].push_back(p);
	}
	// (joints of one block run one after the other in sweep order; joints of DIFFERENT blocks share no body the sweeps write, so their
	// order among each other is free -- but a joint reads the velocity of a body another block may own only through its own block)

	// ---- per block tables ----
	std::vector<JacobiBlockDesc> descs((size_t)B);
	std::vector<int> ints;
	std::vector<uint8_t> exported((size_t)nb, 0);
	std::vector<std::vector<int>> imports((size_t)B), constraintsOf((size_t)B);
	std::vector<int> localOf((size_t)nb, -1), touched;
	int maxOwned = 0, maxImports = 0, maxConstraints = 0;
	for (int g = 0; g < B; ++g)
	{
		std::vector<int>& mine = constraintsOf[(size_t)g];
		for (int u : owned[(size_t)g])
		{
			for (int e = first[(size_t)u]; e < first[(size_t)u + 1]; ++e)
			{
				mine.push_back(incident[(size_t)e]);
			}
		}
		std::sort(mine.begin(), mine.end());
		mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
		for (int k : mine)
		{
			const int slot = cs.order[(size_t)k];
			for (int v : {s->hContactA[(size_t)slot], s->hContactB[(size_t)slot]})
			{
				if (!(ownedBody(v) && blockOf[(size_t)v] == g) && localOf[(size_t)v] < 0)
				{
					localOf[(size_t)v] = (int)owned[(size_t)g].size() + (int)imports[(size_t)g].size();
					imports[(size_t)g].push_back(v);
					touched.push_back(v);
					if (ownedBody(v))
					{
						exported[(size_t)v] = 1;
					}
				}
			}
		}
		for (int v : touched)
		{
			localOf[(size_t)v] = -1;
		}
		touched.clear();
		if ((int)mine.size() > 2 * 512 || (int)imports[(size_t)g].size() > S2_JACOBI_BLOCK_IMPORTS)
		{
			return S2AMD_OK;
		}
		maxOwned = std::max(maxOwned, (int)owned[(size_t)g].size());
		maxImports = std::max(maxImports, (int)imports[(size_t)g].size());
		maxConstraints = std::max(maxConstraints, (int)mine.size());
	}
	for (int g = 0; g < B; ++g)
	{
		JacobiBlockDesc& d = descs[(size_t)g];
		const std::vector<int>& own = owned[(size_t)g];
		const std::vector<int>& imp = imports[(size_t)g];
		const std::vector<int>& mine = constraintsOf[(size_t)g];
		const int nOwn = (int)own.size();
		auto local = [&](int v) {
			if (ownedBody(v) && blockOf[(size_t)v] == g)
			{
				return slotOf[(size_t)v];
			}
			return nOwn + (int)(std::find(imp.begin(), imp.end(), v) - imp.begin());
		};
		// (imports are few hundred at most; a map for the lookup)
		std::unordered_map<int, int> importSlot;
		for (size_t i = 0; i < imp.size(); ++i)
		{
			importSlot[imp[i]] = nOwn + (int)i;
		}
		auto localFast = [&](int v) { return (ownedBody(v) && blockOf[(size_t)v] == g) ? slotOf[(size_t)v] : importSlot[v]; };
		(void)local;
		d.ownedBase = (int)ints.size(), d.ownedCount = nOwn;
		ints.insert(ints.end(), own.begin(), own.end());
		d.importBase = (int)ints.size(), d.importCount = (int)imp.size();
		ints.insert(ints.end(), imp.begin(), imp.end());
		for (int v : imp)
		{
			ints.push_back(ownedBody(v) ? 1 : 0);
		}
		d.constraintBase = (int)ints.size(), d.constraintCount = (int)mine.size();
		std::vector<std::vector<int>> lists((size_t)nOwn);
		for (size_t e = 0; e < mine.size(); ++e)
		{
			const int k = mine[e];
			const int slot = cs.order[(size_t)k];
			const int a = s->hContactA[(size_t)slot], b = s->hContactB[(size_t)slot];
			// the block that owns body A stores the impulses; where nobody owns A, the one that owns B
			const int storer = ownedBody(a) ? blockOf[(size_t)a] : (ownedBody(b) ? blockOf[(size_t)b] : -1);
			ints.push_back(k | (storer == g ? 0x40000000 : 0));
			ints.push_back(localFast(a));
			ints.push_back(localFast(b));
			if (ownedBody(a) && blockOf[(size_t)a] == g)
			{
				lists[(size_t)slotOf[(size_t)a]].push_back((int)(e << 1));
			}
			if (ownedBody(b) && blockOf[(size_t)b] == g)
			{
				lists[(size_t)slotOf[(size_t)b]].push_back((int)(e << 1) | 1);
			}
		}
		d.listBase = (int)ints.size();
		std::vector<int> ranges;
		int at = 0;
		for (int i = 0; i < nOwn; ++i)
		{
			ranges.push_back(at), ranges.push_back((int)lists[(size_t)i].size());
			ints.insert(ints.end(), lists[(size_t)i].begin(), lists[(size_t)i].end());
			at += (int)lists[(size_t)i].size();
		}
		ints.resize((size_t)d.listBase + 2 * mine.size(), 0); // (the kernel copies 2 * constraintCount entries)
		d.rangeBase = (int)ints.size();
		ints.insert(ints.end(), ranges.begin(), ranges.end());
		d.exportBase = (int)ints.size(), d.exportCount = 0;
		for (int i = 0; i < nOwn; ++i)
		{
			if (exported[(size_t)own[(size_t)i]])
			{
				ints.push_back(i);
				d.exportCount += 1;
			}
		}
		d.heavyBase = (int)ints.size(), d.heavyCount = 0;
		for (int i = 0; i < nOwn; ++i)
		{
			if ((int)lists[(size_t)i].size() > S2_JACOBI_HEAVY)
			{
				ints.push_back(i);
				d.heavyCount += 1;
			}
		}
		d.jointBase = (int)ints.size(), d.jointCount = (int)jointsOf[(size_t)g].size();
		for (int p : jointsOf[(size_t)g])
		{
			const int j = js.order[(size_t)p];
			const int a = s->hJointA[(size_t)j], b = s->hJointB[(size_t)j];
			ints.push_back(p);
			ints.push_back((ownedBody(a) && blockOf[(size_t)a] == g) ? slotOf[(size_t)a] : -1);
			ints.push_back((ownedBody(b) && blockOf[(size_t)b] == g) ? slotOf[(size_t)b] : -1);
		}
	}
	if (jacobiStepLds(maxOwned, maxImports, maxConstraints, 64) > 160 * 1024)
	{
		return S2AMD_OK;
	}

	// ---- device: descriptors + ints + the error word; the exchange granules ----
	HIP_TRY(hipSetDevice(s->device));
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	const size_t o0 = 0, o1 = al(descs.size() * sizeof(JacobiBlockDesc)), o2 = o1 + al(ints.size() * sizeof(int));
	std::vector<unsigned char> blob(o2 + 256, 0);
	memcpy(blob.data() + o0, descs.data(), descs.size() * sizeof(JacobiBlockDesc));
	memcpy(blob.data() + o1, ints.data(), ints.size() * sizeof(int));
	bool grew = false;
	int rc = s->dJacobi.ensure(blob.size(), &grew);
	const size_t granBytes = (size_t)2 * 4 * (size_t)nb * sizeof(unsigned long long);
	if (rc == S2AMD_OK)
	{
		rc = s->dJacobiGran.ensure(std::max<size_t>(granBytes, 256), &grew);
	}
	if (rc)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
	}
	HIP_TRY(hipMemcpyAsync(s->dJacobi.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipMemsetAsync(s->dJacobiGran.p, 0, granBytes, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	JacobiView& v = s->jacobi;
	const unsigned char* base = (const unsigned char*)s->dJacobi.p;
	v.descs = (const JacobiBlockDesc*)(base + o0);
	v.ints = (const int*)(base + o1);
	v.granules = (unsigned long long*)s->dJacobiGran.p;
	v.parityStride = 4 * nb;
	unsigned int* devError = nullptr;
	HIP_TRY(hipHostGetDevicePointer((void**)&devError, s->hostError, 0));
	v.error = devError;
	v.deviceError = (unsigned int*)(base + o2);
	v.spinLimit = (unsigned int)s->optPersistSpinLimit;
	v.blockCount = B;
	s->jacobiGranBytes = granBytes;
	s->jacobiMaxOwned = maxOwned, s->jacobiMaxImports = maxImports, s->jacobiMaxConstraints = maxConstraints;
	s->jacobiValid = true;
	if (getenv("S2AMD_DEBUG"))
	{
		int dup = 0, exports = 0;
		for (const JacobiBlockDesc& d : descs)
		{
			dup += d.constraintCount, exports += d.exportCount;
		}
		fprintf(stderr, "[s2amd] s2Solve_Jacobi persistent: %d blocks (at most %d bodies, %d imports, %d constraints), %d constraints held %d times, %d exported bodies\n", B,
				maxOwned, maxImports, maxConstraints, live, dup, exports);
	}
	return S2AMD_OK;
}
