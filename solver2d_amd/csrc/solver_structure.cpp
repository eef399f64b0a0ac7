// Structure of a world for the sweeps: islands, LDS groups, strips of the big islands, colour batches, and the
// device tables the kernels read (GroupTable, StripDesc, PersistDesc).  Rebuilt when the constraint graph changes.
#include "solver_internal.h"

#include <atomic>

namespace
{

constexpr int ColorBitsWords = 4; // graph_coloring.cpp: ColorMasks::WORDS

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

constexpr size_t kBodySlotBytes = sizeof(float4) * 4 + sizeof(float2) * 2 + sizeof(float) + sizeof(uint32_t);
constexpr size_t kContactSlotBytes = sizeof(int2) + sizeof(float4) * 2 + 2 * (sizeof(float4) * 5 + sizeof(float2)) + sizeof(float4) * 4;
constexpr size_t kJointSlotBytes = sizeof(int2) + sizeof(float4) * 8 + sizeof(float2) * 3;

} // namespace

// every strip constraint has two manifold points: selects the persistent kernel's POINTS == 2 variant
bool stripsAllTwoPoints(const s2amdSolver* s)
{
	if (!s->pointsKnown)
	{
		return false; // manifolds are recomputed on the device (world chain): the per-point variant takes any point count
	}
	for (int k = s->persistK0; k < s->persistK1; ++k)
	{
		if (s->contacts.order[(size_t)k] >= 0 && s->hContactPoints[(size_t)s->contacts.order[(size_t)k]] != 2) // (-1: a free position)
		{
			return false;
		}
	}
	return true;
}

bool residentAllTwoPoints(const s2amdSolver* s)
{
	if (!s->pointsKnown || s->residentK1 <= s->residentK0)
	{
		return false;
	}
	for (int k = s->residentK0; k < s->residentK1; ++k)
	{
		if (s->contacts.order[(size_t)k] >= 0 && s->hContactPoints[(size_t)s->contacts.order[(size_t)k]] != 2)
		{
			return false;
		}
	}
	return true;
}

int carveBodies(s2amdSolver* s, int n)
{
	int rc = growFamily(s, s->soaBodies, s->bodySoaCap, n, kBodySlotBytes, 9);
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
	s->bv.massInv = c.take<float2>(cap);
	s->bv.angDamp = c.take<float>(cap);
	s->bv.flags = c.take<uint32_t>(cap);
	s->bv.capacity = n;
	return c.p <= c.end ? S2AMD_OK : fail(S2AMD_E_DEVICE, "internal: body SoA carve overflow");
}

namespace
{

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
// inc != nullptr (the global contact part): every parallel colour batch is laid out with SLACK -- free positions (order -1)
// behind its constraints -- and the colouring state is kept in *inc, so that created contacts can be placed without a
// rebuild (IncrementalGlobal, solver_incremental.cpp).  *positions then has -1 at the free positions.
void colourPart(const std::vector<int>& ids, const std::vector<int>& ea, const std::vector<int>& eb, const std::vector<uint8_t>& conflict,
				int bodyCount, SweepSet& set, std::vector<int>& batchOffsetsOut, bool& hasTailOut, std::vector<int>* positions, int balanced = 0,
				IncrementalGlobal* inc = nullptr, int spareColours = 0, int slackShift = 0, bool colourless = false, int roundSlack = 0, int tailSlack = S2_TAIL_SLACK,
				int tinyColour = 32)
{
	std::vector<int> color, partOrder, partOffsets;
	int cc;
	if (colourless)
	{
		// s2Solve_Jacobi: the contact pass writes no body, one batch in pool order is the whole sweep (and nothing ends up in
		// a sequential tail, whose entries a re-used slot could not give back)
		color.assign(ids.size(), 0);
		cc = ids.empty() ? 0 : 1;
		if (inc)
		{
			inc->colorBits.assign((size_t)bodyCount * ColorBitsWords, 0);
		}
	}
	else
	{
		cc = colorGraph(ea, eb, conflict, bodyCount, color, balanced, inc ? &inc->colorBits : nullptr);
	}
	// stable counting sort of positions by colour
	std::vector<int> pos(ids.size());
	for (size_t i = 0; i < ids.size(); ++i)
	{
		pos[i] = (int)i;
	}
	sortByColor(pos, color, cc, partOrder, partOffsets);
	std::vector<int> rel;
	hasTailOut = makeBatches(partOffsets, rel, balanced == 0, tinyColour);
	int base = (int)set.order.size();
	if (set.colorOffsets.empty())
	{
		set.colorOffsets.push_back(0);
	}
	if (inc)
	{
		// parallel batch i == colour i (makeBatches: one launch per colour below the tail)
		const int parallel = (int)rel.size() - 1 - (hasTailOut ? 1 : 0);
		// ... followed by `spare` EMPTY batches: colours for the contacts of bodies that have every ordinary colour taken (a box
		// inside a pyramid uses all six).  Their bit ids lie above the tail's colours, whose bits the bodies may carry.
		const int spare = std::max(0, std::min(spareColours, 64 * ColorBitsWords - cc));
		const int batches = parallel + spare;
		inc->parallelBatches = batches;
		inc->batchBegin.assign((size_t)batches, 0), inc->batchEnd.assign((size_t)batches, 0);
		inc->freePositions.assign((size_t)batches, {});
		inc->colorIdOfBatch.assign((size_t)batches, 0);
		inc->colorOfPosition.clear();
		std::vector<int> laidOut; // partOrder with the free positions (-1)
		batchOffsetsOut.clear();
		for (int bi = 0; bi < batches; ++bi)
		{
			const int n = bi < parallel ? partOffsets[(size_t)bi + 1] - partOffsets[bi] : 0;
			const int cap = bi < parallel ? ((n + std::max(32, (n / 8) << slackShift)) + 31) & ~31 : std::max(64, ((int)ids.size() / 64 + 31) & ~31);
			const int begin = (int)laidOut.size();
			batchOffsetsOut.push_back(base + begin);
			inc->batchBegin[(size_t)bi] = base + begin, inc->batchEnd[(size_t)bi] = base + begin + cap;
			inc->colorIdOfBatch[(size_t)bi] = bi < parallel ? bi : cc + (bi - parallel);
			if (n > 0)
			{
				laidOut.insert(laidOut.end(), partOrder.begin() + partOffsets[bi], partOrder.begin() + partOffsets[(size_t)bi + 1]);
			}
			laidOut.resize((size_t)begin + cap, -1);
			for (int k = begin + cap - 1; k >= begin + n; --k)
			{
				inc->freePositions[(size_t)bi].push_back(base + k); // descending: pop_back() = the lowest
			}
			inc->colorOfPosition.resize((size_t)base + begin + cap, bi);
			set.colorOffsets.push_back(base + begin + cap);
		}
		batchOffsetsOut.push_back(base + (int)laidOut.size());
		inc->tailBegin = inc->tailEnd = 0;
		inc->tailFree.clear();
		const int tailStart = (int)laidOut.size();
		for (int c = parallel; c < cc; ++c) // the sequential tail: colour by colour
		{
			laidOut.insert(laidOut.end(), partOrder.begin() + partOffsets[c], partOrder.begin() + partOffsets[(size_t)c + 1]);
			if (partOffsets[(size_t)c + 1] > partOffsets[c])
			{
				set.colorOffsets.push_back(base + (int)laidOut.size());
			}
		}
		if (hasTailOut)
		{
			// ... and behind them free positions for created contacts no parallel colour can take (IncrementalGlobal::tailFree)
			const int first = (int)laidOut.size();
			laidOut.resize(laidOut.size() + (size_t)tailSlack, -1);
			inc->tailBegin = base + tailStart, inc->tailEnd = base + (int)laidOut.size();
			for (int k = (int)laidOut.size() - 1; k >= first; --k)
			{
				inc->tailFree.push_back(base + k);
			}
			if (set.colorOffsets.back() > base + tailStart)
			{
				set.colorOffsets.back() = base + (int)laidOut.size();
			}
			else
			{
				set.colorOffsets.push_back(base + (int)laidOut.size());
			}
			batchOffsetsOut.push_back(base + (int)laidOut.size());
		}
		inc->colorOfPosition.resize((size_t)base + laidOut.size(), -1);
		for (int p : laidOut)
		{
			set.order.push_back(p >= 0 ? ids[(size_t)p] : -1);
		}
		if (positions)
		{
			*positions = laidOut;
		}
		return;
	}
	if (roundSlack > 0 && balanced > 0 && !hasTailOut)
	{
		// a strip's or seam's rounds with free positions behind the constraints of every colour (IncrementalStrips): a round
		// stays one pass of the workgroup, so its range never exceeds `balanced` positions
		std::vector<int> laidOut;
		batchOffsetsOut.clear();
		for (int c = 0; c < cc; ++c)
		{
			const int n = partOffsets[(size_t)c + 1] - partOffsets[c];
			if (n <= 0)
			{
				continue;
			}
			const int cap = std::min(balanced, (n + std::max(roundSlack, n / 4) + 7) & ~7);
			batchOffsetsOut.push_back(base + (int)laidOut.size());
			laidOut.insert(laidOut.end(), partOrder.begin() + partOffsets[c], partOrder.begin() + partOffsets[(size_t)c + 1]);
			laidOut.resize(laidOut.size() + (size_t)(std::max(cap, n) - n), -1);
			set.colorOffsets.push_back(base + (int)laidOut.size());
		}
		batchOffsetsOut.push_back(base + (int)laidOut.size());
		for (int p : laidOut)
		{
			set.order.push_back(p >= 0 ? ids[(size_t)p] : -1);
		}
		if (positions)
		{
			*positions = laidOut;
		}
		return;
	}
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
		if (partOffsets[(size_t)c + 1] > partOffsets[c])
		{
			set.colorOffsets.push_back(base + partOffsets[(size_t)c + 1]);
		}
	}
	batchOffsetsOut.clear();
	for (int r : rel)
	{
		batchOffsetsOut.push_back(base + r);
	}
}

int uploadGroupTable(s2amdSolver* s, const HostGroupTable& h, DeviceGroupTable& d, size_t spareIds = 0)
{
	auto pad4 = [](size_t n) { return (n + 3) & ~size_t(3); };
	size_t nBO = pad4(h.bodyOffsets.size()), nBI = pad4(std::max<size_t>(h.bodyIds.size(), 1) + spareIds);
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
	d.spareIdsBase = (int)std::max<size_t>(h.bodyIds.size(), 1), d.spareIdsCount = (int)spareIds;
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

} // namespace

// Message-passing tables of the global part (see MsgBodies): per body the slots of its incident constraints in sweep order.
static int buildMessageTables(s2amdSolver* s, int nb)
{
	const SweepSet& cs = s->contacts;
	int rc = 0;
	bool grew = false;
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
	return S2AMD_OK;
}

// body -> incident constraints of the global part (Jacobi apply, body-centric warm start) with the heavy-body list.
// Ranges {begin, count} with per-body slack and a list with room at its end, mirrored on the host (IncrementalGlobal) so
// that a created contact's two entries can be inserted in place.
static int buildAdjacency(s2amdSolver* s, const std::vector<uint8_t>& conflict, int nb)
{
	const SweepSet& cs = s->contacts;
	IncrementalGlobal& inc = s->inc;
	int rc = 0;
	bool grew = false;
	// body -> incident constraints in SWEEP order (ascending k), key = k<<1 | side, so the per-body
	// sums of jacobiApplyKernel add in exactly the order a sequential pass in sweep order would;
	// read-only shareable bodies are skipped (their deltas are exact zeros)
	std::vector<int> count((size_t)nb, 0);
	const int GC = cs.globalCount; // LDS groups walk their own colours; only the global part is indexed
	for (int k = 0; k < GC; ++k)
	{
		if (cs.order[(size_t)k] < 0)
		{
			continue;
		}
		int a = s->hContactA[cs.order[(size_t)k]], b = s->hContactB[cs.order[(size_t)k]];
		count[(size_t)a] += conflict[a] ? 1 : 0;
		count[(size_t)b] += conflict[b] ? 1 : 0;
	}
	inc.adjRange.assign((size_t)nb, make_int2(0, 0));
	inc.adjCapacity.assign((size_t)nb, 0);
	int total = 0;
	for (int i = 0; i < nb; ++i)
	{
		const int cap = s->hBodyLive[(size_t)i] && conflict[(size_t)i] ? ((count[(size_t)i] + std::max(4 << std::min(s->slackShift, 1), count[(size_t)i] / 4) + 3) & ~3) : 0;
		inc.adjRange[(size_t)i] = make_int2(total, 0);
		inc.adjCapacity[(size_t)i] = cap;
		total += cap;
	}
	inc.adjUsed = total;
	// room for lists that outgrow their slack and move to the end: as much again as the lists themselves, whatever the slack -- entries
	// nobody uses cost a memset, while lists that found no room cost a build each (r6: 30 of the Tumbler fill's 54, 25 of 90 under XPBD in
	// the wrecked pyramid) and the slack they then doubled made every later build dearer (8 entries more per body: 2.7 ms of upload)
	inc.adjList.assign((size_t)total + (size_t)std::max(4096, total), 0);
	for (int k = 0; k < GC; ++k)
	{
		if (cs.order[(size_t)k] < 0)
		{
			continue;
		}
		int a = s->hContactA[cs.order[(size_t)k]], b = s->hContactB[cs.order[(size_t)k]];
		if (conflict[a])
		{
			int2& r = inc.adjRange[(size_t)a];
			inc.adjList[(size_t)(r.x + r.y++)] = (k << 1) | 0;
		}
		if (conflict[b])
		{
			int2& r = inc.adjRange[(size_t)b];
			inc.adjList[(size_t)(r.x + r.y++)] = (k << 1) | 1;
		}
	}
	std::vector<int> heavyBodies;
	for (int i = 0; i < nb; ++i)
	{
		if (inc.adjRange[(size_t)i].y > S2_HEAVY_DEGREE)
		{
			heavyBodies.push_back(i);
		}
	}
	// (room for bodies that become heavy under placed contacts: a wave of the body-centric launches per entry, idle while it is free --
	// 16 were used up in a step or two of a pile that is being pressed together: 29 of the Tumbler fill's 54 builds, and with 64 still 27)
	const int heavyCap = (((int)heavyBodies.size() + std::max(1024, (int)heavyBodies.size())) + 15) & ~15;
	inc.heavy.assign((size_t)heavyCap + 1, 0);
	inc.heavy[0] = (int)heavyBodies.size();
	std::copy(heavyBodies.begin(), heavyBodies.end(), inc.heavy.begin() + 1);
	grew = false;
	if ((rc = s->dAdjOffsets.ensure(std::max<size_t>((size_t)nb, 1) * sizeof(int2), &grew)) != 0 ||
		(rc = s->dAdjList.ensure(std::max<size_t>(inc.adjList.size(), 1) * sizeof(int), &grew)) != 0 ||
		(rc = s->dAdjHeavy.ensure(inc.heavy.size() * sizeof(int), &grew)) != 0)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
	}
	// the list's capacity is the device buffer's (DevBuf grows by half: the mirror follows, so relocations can use it all)
	inc.adjList.resize(s->dAdjList.bytes / sizeof(int), 0);
	s->adjHeavyCapacity = heavyCap;
	HIP_TRY(hipMemcpyAsync(s->dAdjHeavy.p, inc.heavy.data(), inc.heavy.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
	if (nb > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dAdjOffsets.p, inc.adjRange.data(), (size_t)nb * sizeof(int2), hipMemcpyHostToDevice, s->stream));
	}
	if (inc.adjUsed > 0)
	{
		HIP_TRY(hipMemcpyAsync(s->dAdjList.p, inc.adjList.data(), (size_t)inc.adjUsed * sizeof(int), hipMemcpyHostToDevice, s->stream));
	}
	s->adjValid = true;
	return S2AMD_OK;
}

// body -> incident joints of the global part in SWEEP order, key = k << 1 | side (mouse joints only have a B side): the
// body-centric joint warm start (joint_kernels.hip: warmStartJointsBodiesKernel)
static int buildJointAdjacency(s2amdSolver* s, int nb)
{
	s->jointAdjValid = false;
	const SweepSet& js = s->joints;
	const int GJ = js.globalCount;
	if (GJ <= 0 || nb <= 0)
	{
		return S2AMD_OK;
	}
	std::vector<int2> range((size_t)nb, make_int2(0, 0));
	auto movable = [&](int body) { return body >= 0 && body < nb && (s->hBodyFlags[(size_t)body] & (S2F_WRITE_VEL | S2F_WRITE_POS)) != 0; };
	for (int k = 0; k < GJ; ++k)
	{
		const int slot = js.order[(size_t)k];
		const int a = s->hJointType[(size_t)slot] == S2AMD_JOINT_MOUSE ? -1 : s->hJointA[(size_t)slot], b = s->hJointB[(size_t)slot];
		range[(size_t)std::max(a, 0)].y += movable(a) ? 1 : 0;
		range[(size_t)std::max(b, 0)].y += movable(b) ? 1 : 0;
	}
	int total = 0;
	for (int i = 0; i < nb; ++i)
	{
		range[(size_t)i].x = total;
		total += range[(size_t)i].y;
		range[(size_t)i].y = 0;
	}
	std::vector<int> list((size_t)std::max(total, 1), 0);
	for (int k = 0; k < GJ; ++k)
	{
		const int slot = js.order[(size_t)k];
		const int a = s->hJointType[(size_t)slot] == S2AMD_JOINT_MOUSE ? -1 : s->hJointA[(size_t)slot], b = s->hJointB[(size_t)slot];
		if (movable(a))
		{
			list[(size_t)(range[(size_t)a].x + range[(size_t)a].y++)] = (k << 1) | 0;
		}
		if (movable(b))
		{
			list[(size_t)(range[(size_t)b].x + range[(size_t)b].y++)] = (k << 1) | 1;
		}
	}
	int rc;
	bool grew = false;
	if ((rc = s->dJointAdjRange.ensure((size_t)nb * sizeof(int2), &grew)) != 0 || (rc = s->dJointAdjList.ensure(list.size() * sizeof(int), &grew)) != 0)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
	}
	HIP_TRY(hipMemcpyAsync(s->dJointAdjRange.p, range.data(), (size_t)nb * sizeof(int2), hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipMemcpyAsync(s->dJointAdjList.p, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream)); // range and list are locals
	s->jointAdjValid = true;
	return S2AMD_OK;
}

// The lean / persistent strip tables of one strip partition (strip_kernel.hip): per-strip descriptors, warm-start slots,
// the persistent kernel's register / LDS plan.  Part of buildStructureWith; leaves s->leanAValid / leanBValid /
// persistValid false when this partition cannot use those kernels.
static int buildLeanStripTables(s2amdSolver* s, const StripPartition& strips, const std::vector<uint8_t>& conflict, const std::vector<int>& seamGroup,
								int stripBaseC, int nb)
{
	int rc = 0;
	SweepSet& cs = s->contacts;
	SweepSet& js = s->joints;
	const int k0 = stripBaseC, k1 = stripBaseC + cs.stripCount;
	// body -> incident strip constraints in sweep order
	std::vector<int> off((size_t)nb + 1, 0), inc;
	for (int k = k0; k < k1; ++k)
	{
		if (cs.order[(size_t)k] < 0)
		{
			continue; // a free position (IncrementalStrips)
		}
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
			if (cs.order[(size_t)k] < 0)
			{
				continue;
			}
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
	const char* leanWhy = "";
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
			if (!ok)
			{
				leanWhy = d.bodyCount > S2_STRIP_BODY_CHUNKS * 256 ? "more than 1024 bodies in a strip or seam" : (withSlots ? "more than 8 interior colours" : "more than 6 seam colours");
			}
			maxRounds = std::max(maxRounds, d.batchCount);
			for (int b = b0; b < b1 && ok; ++b)
			{
				int4 bt = t.cBatches[(size_t)b];
				ok = bt.z == 0;
				if (!ok)
				{
					leanWhy = "a sequential tail batch";
				}
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
			if (ok && records > (160 * 1024) / 16)
			{
				leanWhy = "LDS: bodies + warm-start slots";
			}
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
		fprintf(stderr, "[s2amd] strips: %d strips, %d seams, lean A %d B %d, strip joints %d, CUs %d%s%s\n", s->hStripA.count(), s->hStripB.count(),
				(int)s->leanAValid, (int)s->leanBValid, js.stripCount, s->cuCount, leanWhy[0] ? " -- lean tables: " : "", leanWhy);
	}
	// ---- persistent strip step (strip_kernel.hip: stripStepKernel): per workgroup both seams' remaps, the
	// import / export lists of the symmetric exchange, warm-start term slots, granule buffers ----
	// `ok`: what every persistent kernel needs (the partition's seam structure, co-resident workgroups); `okSoft`: what the
	// register-resident soft kernels need on top (lean tables, no joints, one constraint per thread and round, few seam rounds,
	// their LDS budget).  The op interpreter (generic_kernel.hip) runs on `ok` alone.
	s->persistValid = false;
	s->genericValid = false;
	if (s->optPersist && s->hostError != nullptr && s->hStripA.count() <= s->cuCount && (persistTablesOk || s->optGeneric))
	{
		bool okSoft = persistTablesOk && js.stripCount == 0;
		const char* whySoft = okSoft ? "" : (js.stripCount ? "joints in the strips" : "lean tables");
		const HostGroupTable& A = s->hStripA;
		const HostGroupTable& B = s->hStripB;
		const int K = A.count();
		bool ok = true;
		const char* why = "";
#define NEED(cond)                                                                                                                \
do                                                                                                                           \
{                                                                                                                            \
	if (ok && !(cond))                                                                                                       \
	{                                                                                                                        \
		ok = false;                                                                                                          \
		why = #cond;                                                                                                         \
	}                                                                                                                        \
} while (0)
#define NEEDSOFT(cond)                                                                                                            \
do                                                                                                                           \
{                                                                                                                            \
	if (okSoft && !(cond))                                                                                                   \
	{                                                                                                                        \
		okSoft = false;                                                                                                      \
		whySoft = #cond;                                                                                                     \
	}                                                                                                                        \
} while (0)
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
				NEEDSOFT(A.cBatches[(size_t)bb].y - A.cBatches[(size_t)bb].x <= 256); // one constraint per thread and round
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
					NEED(false);
				}
			}
			NEEDSOFT(leftBodies[(size_t)sm].size() <= 256 && rightBodies[(size_t)sm].size() <= 256);
		}
		// granule buffers: per seam {toLeft: 4 per right body, toRight: 4 per left body}, two parities
		std::vector<int> seamBase((size_t)std::max(S, 0), 0);
		int granules = 0;
		for (int sm = 0; sm < S; ++sm)
		{
			seamBase[(size_t)sm] = granules;
			// (+ room for the bodies a seam may come to carry later: IncrementalStrips)
			granules += 4 * ((int)(leftBodies[(size_t)sm].size() + rightBodies[(size_t)sm].size()) + 2 * S2_STRIP_ADOPT_SLACK);
		}
		const int parityStride = granules;
		// TGS_Soft keeps the seam constraints in registers when no seam has more than two colour batches and no interior
		// more than six (strip_kernel.hip: SEAMREG): then they cost no LDS at all
		bool seamRegs = s->optSeamRegs != 0 && maxRoundsA <= S2_STRIP_ROUNDS;
		for (int sm = 0; sm < S && seamRegs; ++sm)
		{
			const int g = seamGroup[(size_t)sm];
			seamRegs = g < 0 || B.cBatchOffsets[(size_t)g + 1] - B.cBatchOffsets[(size_t)g] <= 2;
		}
		// pair_kernel.hip: the same limits, whatever the seam_regs option says
		bool pairLanes = maxRoundsA <= S2_STRIP_ROUNDS;
		for (int sm = 0; sm < S && pairLanes; ++sm)
		{
			const int g = seamGroup[(size_t)sm];
			pairLanes = g < 0 || B.cBatchOffsets[(size_t)g + 1] - B.cBatchOffsets[(size_t)g] <= 2;
		}
		std::vector<PersistDesc> descs((size_t)K);
		int genericBodies = 0, genericSeamBodies = 0, genericExports = 0, genericJoints = 0;
		std::vector<int> remap, exportSrc, importIds;
		std::vector<int> replicaStamp((size_t)nb, -1), replicaSlot((size_t)nb, -1);
		int ldsRecords = 0, ldsRecordsWide = 0, bodyRecordsMax = 0, maxStaged = 0, maxStripBodies = 0;
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
			d.seamGroup[0] = d.seamGroup[1] = -1;
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
				d.seamGroup[side] = g;
				d.importCount[side] = (int)imports.size();
				d.exportCount[side] = (int)exports.size();
				importIds.insert(importIds.end(), imports.begin(), imports.end());
				importIds.insert(importIds.end(), (size_t)S2_STRIP_ADOPT_SLACK, 0); // (room: IncrementalStrips)
				for (int body : exports)
				{
					exportSrc.push_back(ownerSlot[body]);
				}
				exportSrc.insert(exportSrc.end(), (size_t)S2_STRIP_ADOPT_SLACK, 0);
				const int nR = (int)rightBodies[(size_t)sm].size();
				const int toLeft = seamBase[(size_t)sm], toRight = seamBase[(size_t)sm] + 4 * (nR + S2_STRIP_ADOPT_SLACK);
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
						NEED(false);
					}
				}
				remap.insert(remap.end(), (size_t)2 * S2_STRIP_ADOPT_SLACK, 0); // (room for the seam's later bodies, from either side)
				int b0 = B.cBatchOffsets[(size_t)g], b1 = B.cBatchOffsets[(size_t)g + 1];
				NEEDSOFT(b1 - b0 <= S2_PERSIST_B_ROUNDS);
				d.seamBatchCount[side] = std::min(b1 - b0, S2_PERSIST_B_ROUNDS);
				for (int bb = b0; bb < b0 + d.seamBatchCount[side] && ok; ++bb)
				{
					int4 bt = B.cBatches[(size_t)bb];
					NEEDSOFT(bt.z == 0);
					d.seamBatch[side][bb - b0] = make_int2(bt.x, bt.y);
					seamSlots += bt.y - bt.x;
				}
				importOffset += d.importCount[side];
				if (side == 1)
				{
					genericSeamBodies = std::max(genericSeamBodies, B.bodyOffsets[(size_t)g + 1] - B.bodyOffsets[(size_t)g]);
				}
				else
				{
					genericExports = std::max(genericExports, d.exportCount[side]);
				}
			}
			for (int r = 0; r < S2_PERSIST_B_ROUNDS && ok; ++r)
			{
				int n0 = r < d.seamBatchCount[0] ? d.seamBatch[0][r].y - d.seamBatch[0][r].x : 0;
				int n1 = r < d.seamBatchCount[1] ? d.seamBatch[1][r].y - d.seamBatch[1][r].x : 0;
				NEEDSOFT(n0 + n1 <= 512); // both seams share a round: at most two constraints per thread
			}
			// (the budgets include the bodies the strip may still adopt: IncrementalStrips)
			const int nt = importOffset + S2_STRIP_ADOPT_SLACK;
			maxStaged = std::max(maxStaged, nt);
			maxStripBodies = std::max(maxStripBodies, nbA + S2_STRIP_ADOPT_SLACK);
			genericBodies = std::max(genericBodies, nt);
			{
				auto rangeOf = [](const HostGroupTable& t, int g) {
					const int b0 = t.jBatchOffsets[(size_t)g], b1 = t.jBatchOffsets[(size_t)g + 1];
					return b0 < b1 ? t.jBatches[(size_t)b1 - 1].y - t.jBatches[(size_t)b0].x : 0;
				};
				const int gS = d.seamGroup[1];
				genericJoints = std::max(genericJoints, rangeOf(A, i) + (gS >= 0 ? rangeOf(B, gS) : 0));
			}
			// bodies, seam constraints (S2_PERSIST_Q_NARROW records each for TGS_Soft, S2_PERSIST_Q_WIDE for the other kinds)
			int fixedRecords = 3 * nt + (nt + 3) / 4 + (nt + 1) / 2; // velocity, pose, integrator constants, angular damping, inverse masses
			const int seamRecordsNarrow = seamRegs ? 0 : S2_PERSIST_Q_NARROW * seamSlots;
			NEEDSOFT(fixedRecords + seamRecordsNarrow + 2 * 16 <= (160 * 1024) / 16 && nt < 16384); // the plan's own records are checked when it is known (persistPlan)
			NEED(genericStepLds(nt, genericSeamBodies, genericExports, 16, 0) <= 160 * 1024); // (with the plan's ops and its XPBD history: Executor::genericPlan)
			ldsRecords = std::max(ldsRecords, fixedRecords + seamRecordsNarrow);
			bodyRecordsMax = std::max(bodyRecordsMax, fixedRecords);
			ldsRecordsWide = std::max(ldsRecordsWide, fixedRecords + S2_PERSIST_Q_WIDE * seamSlots);
		}
		if (getenv("S2AMD_DEBUG"))
		{
			fprintf(stderr, "[s2amd] persistent step: %s (K=%d, lds records %d, granules/parity %d)%s%s%s%s\n",
					ok ? (okSoft ? "eligible" : "op interpreter only") : "NOT eligible", K, ldsRecords, parityStride, ok ? "" : " -- failed: ", why,
					ok && !okSoft ? " -- resident soft kernels: " : "", ok && !okSoft ? whySoft : "");
			int histA[16] = {0}, histB[16] = {0};
			for (int i = 0; i < K; ++i)
			{
				int ra = A.cBatchOffsets[(size_t)i + 1] - A.cBatchOffsets[(size_t)i];
				int rb = std::max(descs[(size_t)i].seamBatchCount[0], descs[(size_t)i].seamBatchCount[1]);
				histA[std::min(ra, 15)] += 1;
				histB[std::min(rb, 15)] += 1;
			}
			for (int r = 0; r < 16; ++r)
			{
				if (histA[r] || histB[r])
				{
					fprintf(stderr, "[s2amd]   rounds %d: %d interiors, %d seam pairs\n", r, histA[r], histB[r]);
				}
			}
			if (getenv("S2AMD_DEBUG_ROUNDS")) // constraints (free positions included) per interior round of every strip
			{
				for (int i = 0; i < K; ++i)
				{
					fprintf(stderr, "[s2amd]   strip %d:", i);
					for (int bi = A.cBatchOffsets[(size_t)i]; bi < A.cBatchOffsets[(size_t)i + 1]; ++bi)
					{
						fprintf(stderr, " %d", A.cBatches[(size_t)bi].y - A.cBatches[(size_t)bi].x);
					}
					fprintf(stderr, "\n");
				}
			}
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
			// + one census granule per strip (wide_kernel.hip), + the exchange rings of the overflow workgroup (PersistView::overflowGranBase)
			const size_t overflowGranBase = ((size_t)2 * parityStride + (size_t)K + 31) & ~size_t(31);
			s->granuleBytes = (((overflowGranBase + S2_OVERFLOW_GRANULES) * sizeof(unsigned long long)) + 255) & ~size_t(255);
			if ((rc = s->dPersist.ensure(blob.size(), &grewP)) != 0 || (rc = s->dGranules.ensure(s->granuleBytes, &grewP)) != 0 ||
				(rc = s->dOverflowBodies.ensure(S2_OVERFLOW_BODIES * sizeof(int), &grewP)) != 0)
			{
				return rc;
			}
			HIP_TRY(hipMemsetAsync(s->dOverflowBodies.p, 0xff, S2_OVERFLOW_BODIES * sizeof(int), s->stream)); // (-1: free entries)
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
			pv.state = (unsigned int*)(base + o4 + 128); // (a cache line of its own in the zeroed 256 bytes: wide_kernel.hip's commit counter)
			pv.maxStaged = maxStaged;
			pv.maxStripBodies = (maxStripBodies + 31) & ~31;
			pv.parityStride = parityStride;
			pv.censusBase = 2 * parityStride;
			pv.overflowBodies = (const int*)s->dOverflowBodies.p;
			pv.overflowGranBase = (int)overflowGranBase;
			// fresh buffers start from zero tags
			HIP_TRY(hipMemsetAsync(s->dGranules.p, 0, s->granuleBytes, s->stream));
			pv.wideRounds = maxRoundsA > S2_STRIP_ROUNDS ? 1 : 0;
			pv.seamRegs = seamRegs ? 1 : 0;
			pv.pairLanes = pairLanes ? 1 : 0;
			pv.maxRoundsA = maxRoundsA;
			pv.maxSeamRounds = 0;
			pv.parkSeamWidth = pv.parkInteriorWidth = 64;
			for (const PersistDesc& d : descs)
			{
				pv.maxSeamRounds = std::max(pv.maxSeamRounds, std::max(d.seamBatchCount[0], d.seamBatchCount[1]));
				// wide_kernel.hip parks seam rounds 3 and 4 (both seams of a strip share a round: left batch, then right batch)
				for (int r = 2; r < S2_PERSIST_B_ROUNDS; ++r)
				{
					const int n0 = r < d.seamBatchCount[0] ? d.seamBatch[0][r].y - d.seamBatch[0][r].x : 0;
					const int n1 = r < d.seamBatchCount[1] ? d.seamBatch[1][r].y - d.seamBatch[1][r].x : 0;
					pv.parkSeamWidth = std::max(pv.parkSeamWidth, n0 + n1);
				}
			}
			// ... and interior rounds 7 and 8
			for (int i = 0; i < K; ++i)
			{
				for (int bb = A.cBatchOffsets[(size_t)i] + S2_STRIP_ROUNDS; bb < A.cBatchOffsets[(size_t)i + 1]; ++bb)
				{
					pv.parkInteriorWidth = std::max(pv.parkInteriorWidth, A.cBatches[(size_t)bb].y - A.cBatches[(size_t)bb].x);
				}
			}
			if ((s->optPersistDebug & 16) != 0)
			{
				pv.parkSeamWidth = 192, pv.parkInteriorWidth = 128; // (tests: the parked variants on partitions that do not need them; r5: beside
																		 // the local anchors in LDS -- wideLocalsInLds -- the full 512 / 256 columns no longer fit)
			}
			pv.parkSeamWidth = (pv.parkSeamWidth + 63) & ~63, pv.parkInteriorWidth = (pv.parkInteriorWidth + 63) & ~63;
			s->persistK0 = k0, s->persistK1 = k1;
			pv.allTwoPoints = stripsAllTwoPoints(s) ? 1 : 0;
			pv.ldsRecords = ldsRecords;
			pv.bodyRecords = bodyRecordsMax;
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
			s->hPersistRemap = remap;
			s->hPersistDescs = descs;
			s->persistValid = okSoft;
			s->genericValid = s->optGeneric != 0;
			s->genericBodies = genericBodies, s->genericSeamBodies = genericSeamBodies, s->genericExports = genericExports;
			s->genericJoints = genericJoints;
		}
	}
	return rc;
}

// Descriptors of the resident islands (strip_kernel.hip: islandStepKernel reads one StripDesc per workgroup).  Sets
// s->residentRejected when a group does not fit the kernel (rounds, round width, body chunks, LDS).
static int buildResidentTables(s2amdSolver* s)
{
	const HostGroupTable& t = s->hResident;
	s->residentView = StripTableView{};
	s->residentRounds = 0;
	if (t.count() == 0)
	{
		return S2AMD_OK;
	}
	std::vector<StripDesc> descs((size_t)t.count());
	int ldsRecords = 0;
	bool ok = true;
	for (int g = 0; g < t.count() && ok; ++g)
	{
		StripDesc& d = descs[(size_t)g];
		memset(&d, 0, sizeof(d));
		d.bodyBase = t.bodyOffsets[(size_t)g];
		d.bodyCount = t.bodyOffsets[(size_t)g + 1] - d.bodyBase;
		const int b0 = t.cBatchOffsets[(size_t)g], b1 = t.cBatchOffsets[(size_t)g + 1];
		d.batchCount = b1 - b0;
		ok = d.batchCount <= S2_STRIP_ROUNDS_MAX && d.bodyCount <= S2_STRIP_BODY_CHUNKS * 512;
		for (int b = b0; b < b1 && ok; ++b)
		{
			const int4 bt = t.cBatches[(size_t)b];
			ok = bt.z == 0 && bt.y - bt.x <= 512;
			d.batch[b - b0] = make_int4(bt.x, bt.y, 0, 0);
		}
		while (d.ownedCount < d.bodyCount && ((uint32_t)t.bodyIds[(size_t)d.bodyBase + d.ownedCount] & S2G_OWNED) != 0)
		{
			d.ownedCount += 1;
		}
		const int nbG = d.bodyCount;
		const int records = 3 * nbG + (nbG + 3) / 4 + 2 * ((nbG + 1) / 2); // (+ the local centres wide_kernel.hip: wideIslandKernel stages)
		ok = ok && (size_t)records * 16 + 128 * sizeof(Op) <= 160 * 1024;
		ldsRecords = std::max(ldsRecords, records);
		s->residentRounds = std::max(s->residentRounds, d.batchCount);
	}
	if (!ok)
	{
		s->residentRejected = true;
		return S2AMD_OK;
	}
	bool grew = false;
	int rc = s->dResidentDesc.ensure(descs.size() * sizeof(StripDesc), &grew);
	if (rc)
	{
		return rc;
	}
	if (grew)
	{
		s->layoutGeneration += 1;
	}
	HIP_TRY(hipMemcpyAsync(s->dResidentDesc.p, descs.data(), descs.size() * sizeof(StripDesc), hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream)); // descs is a local
	s->residentView.descs = (const StripDesc*)s->dResidentDesc.p;
	s->residentView.bodyIds = s->dResident.view.bodyIds;
	s->residentView.groupCount = t.count();
	s->residentView.ldsRecords = ldsRecords;
	return S2AMD_OK;
}

// One structure build, phase by phase.  The phases share the edge lists, the island / strip partition and the sweep sets
// they append to; each one is a method below, buildStructureWith() at the end of the block is the driver.
struct StructureBuild
{
	s2amdSolver* s;
	const int solverType;
	const float stripScale;
	const int cls; // 0: the sweeps write velocities, 1: positions too (conflict = what a colour must not share)
	const bool needAdj; // s2Solve_Jacobi: body-centric apply instead of colours
	const bool grouped;
	const bool ldsGroups; // small islands into LDS groups / resident islands (else: with everything else, into strips or the global part)
	const bool wantStrips;
	const bool residentWanted;
	const bool stripSlackWanted; // strip and seam rounds get free positions for created contacts (IncrementalStrips): the soft contact solvers' strips
	const int nb;
	SweepSet& cs;
	SweepSet& js;

	std::vector<uint8_t> conflict;
	EdgeList ce, je; // potential contact constraints and joints in pool order
	int C = 0, J = 0;
	std::vector<int> cPart, jPart; // per edge: -1 = global part, else LDS group id
	int groupCount = 0;
	std::vector<uint32_t> flags;
	std::vector<std::vector<int>> cOf, jOf; // index 0 = global, g + 1 = group g
	StripPartition strips;
	bool stripsNeedOneLaunch = false;
	LocalSlots slots;
	std::vector<int> seamGroup;
	struct SpareRound
	{
		int table, group, begin, count, round;
	};
	std::vector<SpareRound> spareRounds; // the closed spare rounds of the strips (table 0) and seams (table 1): emitGroup
	int stripBaseC = 0;
	int overflowBase = 0, overflowCount = 0; // the overflow region behind the strips (IncrementalStrips::overflowFree)
	bool rebuild = false; // the structure just built cannot run: build again with what was learnt (stripsRejected / residentRejected)

	double t0, tPhase;
	bool prepTimes; // S2AMD_DEBUG_PREP=1: where the host time of a structure build goes
	double tSlots = 0, tColour = 0, tAppend = 0; // ... inside emitGroup, summed over the groups

	StructureBuild(s2amdSolver* solver, int type, float scale)
		: s(solver), solverType(type), stripScale(scale), cls(isPositionSolver(type) ? 1 : 0), needAdj(type == s2amd_solverJacobi),
		  grouped(solver->optGroups != 0 && !needAdj),
		  ldsGroups(grouped && (solver->optGroupPatience == 0 || solver->graphAge >= solver->groupPatienceNow)),
		  // strips pay off through the lean / persistent strip kernels, which exist for the soft contact sweeps
		  wantStrips(grouped && solver->optStrips != 0 && !solver->stripsRejected && solver->graphAge >= solver->stripPatienceNow &&
					 (solver->optStripsAnySolver != 0 || isSoftFamily(type) || genericWanted(solver))),
		  residentWanted(grouped && isSoftFamily(type) && solver->optIslandResident != 0 && !solver->residentRejected),
		  // (slack positions in the strips' rounds: the soft solvers on the 512-thread kernel -- in the 256-thread kernels of SoftStep /
		  // PGS_Soft (`wide` off) a seam round that outgrows 256 positions is dealt in two passes, which cost them 0.26 -> 0.40 ms per
		  // SoftStep step at base 200)
		  stripSlackWanted(solver->optStripSlack != 0 && solver->optIncremental != 0 && solver->optPersist != 0 &&
						   (type == s2amd_solverTGS_Soft || (solver->optWide != 0 && (type == s2amd_solverSoftStep || type == s2amd_solverPGS_Soft)) ||
							(!isSoftFamily(type) && solver->optGenericPlace != 0 && genericWanted(solver)))), // (r6: IncrementalStrips::takeOnly)
		  nb(solver->bodyCapacity),
		  cs(solver->contacts), js(solver->joints), slots(solver->bodyCapacity)
	{
		static const bool fromEnv = getenv("S2AMD_DEBUG_PREP") != nullptr;
		prepTimes = fromEnv;
		t0 = tPhase = nowMs();
	}

	// the persistent op interpreter takes every family and joints: strips for all of them (it needs the persistent machinery)
	static bool genericWanted(const s2amdSolver* solver)
	{
		return solver->optGeneric != 0 && solver->optPersist != 0 && solver->optStripLean != 0 && !solver->persistFailed && solver->hostError != nullptr;
	}

	// Strip width by solver: two BFS levels per strip (five interior colour rounds) for the kernels that sweep a seam cheaply --
	// wide_kernel.hip (TGS_Soft: seams in registers) and the op interpreter (seams swept once) --, few wide strips for SoftStep
	// and PGS_Soft, whose seam constraints live in LDS and are swept by both neighbours (0.25 vs 0.38 ms per SoftStep step at
	// base 200).  A width set by the caller ("strip_bodies") goes for every solver.
	static int stripBodiesFor(const s2amdSolver* solver, int type)
	{
		// (r4: s2Solve_PGS_Soft and s2Solve_SoftStep run on the 512-thread kernel too -- the same 22-dword constraint in registers, SoftStep's
		// rA0 / rB0 beside it in LDS -- and take TGS_Soft's thin strips with them; with `wide` off, the 256-thread kernel and its few wide strips)
		const bool ldsSeams = (type == s2amd_solverSoftStep || type == s2amd_solverPGS_Soft) && solver->optWide == 0;
		if (solver->stripBodiesSet || !ldsSeams)
		{
			return solver->optStripBodies;
		}
		return solver->optStripBodiesLds;
	}

	// bodies from which on an island is swept faster by strips than by one workgroup, where it has the strips to itself (findIslands)
	static int groupLowerLimit(int type)
	{
		// (measured and NOT adopted, r6: s2Solve_SoftStep / s2Solve_PGS_Soft have no register-resident island kernel and their strips beat
		// one workgroup from 80 / 200 bodies on -- 0.136 against 0.17-0.25 ms, 0.060 against 0.07-0.09 --, but a lower limit for them
		// makes the two routes of the drop-in build different structures and moves PGS_Soft's unconverged pile outside the physical
		// tolerances the tests hold the group order to)
		return isSoftFamily(type) ? 896 : 1024;
	}

	static bool isSoftFamily(int type)
	{
		return type == s2amd_solverTGS_Soft || type == s2amd_solverSoftStep || type == s2amd_solverPGS_Soft;
	}

	// the structure the solver holds serves this solver type in everything but the strips that have fallen due (the graph has aged
	// enough): what a worker thread can build while the steps go on (solver_async.cpp)
	bool onlyStripsMissing() const
	{
		const bool colourFree = s->inc.colourFreePlaced && !needAdj;
		return !s->structureDirty && cls == s->orderSolverClass && grouped == s->orderGrouped && ldsGroups == s->orderLdsGroups && wantStrips && !s->orderStrips && s->adjValid && !colourFree &&
			   residentWanted == s->orderResident && needAdj == s->orderColourless;
	}

	// does the structure the solver holds already serve this solver type?
	bool upToDate() const
	{
		// (contacts placed into a structure built for s2Solve_Jacobi took any free position, whatever its colour: only Jacobi can run on that)
		const bool colourFree = s->inc.colourFreePlaced && !needAdj;
		return !s->structureDirty && cls == s->orderSolverClass && grouped == s->orderGrouped && ldsGroups == s->orderLdsGroups && wantStrips == s->orderStrips && (!wantStrips || stripBodiesFor(s, solverType) == s->orderStripBodies) && s->adjValid && !colourFree &&
			   residentWanted == s->orderResident && needAdj == s->orderColourless;
	}

	void phase(const char* name)
	{
		if (prepTimes)
		{
			double t = nowMs();
			fprintf(stderr, "[s2amd] prep %-22s %.3f ms\n", name, t - tPhase);
			tPhase = t;
		}
	}

	void gather(const EdgeList& e, const std::vector<int>& ks, std::vector<int>& ids, std::vector<int>& a, std::vector<int>& b) const
	{
		ids.clear(), a.clear(), b.clear();
		for (int k : ks)
		{
			ids.push_back(e.ids[k]);
			a.push_back(e.a[k]);
			b.push_back(e.b[k]);
		}
	}

	// ---- phase 1: the potential constraints (edges) of this structure, the hub rule ----
	int gatherEdges()
	{
		// slack that was laid out generously after an exhaustion and then sat unused (a rebuild for another reason finds
		// three quarters of it still free) shrinks again: free positions are threads that do nothing in every sweep
		if (s->slackShift > 0 && !s->slackBumped && s->slackAtBuild > 0 && (long long)s->slackPositions * 4 > (long long)s->slackAtBuild * 3)
		{
			s->slackShift -= 1;
		}
		s->slackBumped = false;
		conflict.resize((size_t)nb);
		for (int i = 0; i < nb; ++i)
		{
			conflict[i] = (s->hBodyFlags[i] & (cls == 1 ? S2F_WRITE_POS : S2F_WRITE_VEL)) != 0;
		}

		// potential constraints in pool order (the reference's gather, e.g. solve_tgs_soft.c:162-179, over the slots that CAN have
		// manifold points; the ones that have none this step are no-ops wherever the sweep order puts them)
		// hub bodies (solver_internal.h): potential constraints per writable body
		std::vector<int> degree((size_t)nb, 0);
		for (int i = 0; i < s->contactCapacity; ++i)
		{
			if (s->hContactEdge[i] && s->hContactDead[i])
			{
				s->hContactEdge[i] = 0; // a destroyed contact leaves the structure with this rebuild
				s->hContactDead[i] = 0;
			}
			if (s->hContactEdge[i])
			{
				degree[(size_t)s->hContactA[i]] += 1;
				degree[(size_t)s->hContactB[i]] += 1;
			}
		}
		s->hBodyHub.assign((size_t)nb, 0);
		bool anyHub = false;
		for (int i = 0; i < nb; ++i)
		{
			if (degree[(size_t)i] > S2_HUB_DEGREE && (s->hBodyFlags[(size_t)i] & (S2F_WRITE_VEL | S2F_WRITE_POS)) != 0)
			{
				s->hBodyHub[(size_t)i] = 1;
				anyHub = true;
			}
		}
		if (anyHub && s->worldResident && !s->pointsKnown)
		{
			int rcPoints = fetchPointCounts(s); // which manifolds on the hubs have points right now
			if (rcPoints)
			{
				return rcPoints;
			}
		}
		s->hContactWatched.assign((size_t)s->contactCapacity, 0);
		s->watchedCount = 0;
		for (int i = 0; i < s->contactCapacity; ++i)
		{
			if (s->hContactEdge[i] && anyHub && (s->hBodyHub[(size_t)s->hContactA[i]] || s->hBodyHub[(size_t)s->hContactB[i]]))
			{
				s->hContactWatched[(size_t)i] = 1;
				s->watchedCount += 1;
				if (s->hContactPoints[(size_t)i] <= 0)
				{
					continue; // a potential constraint on a hub body: structural only while its manifold has points
				}
			}
			if (s->hContactEdge[i])
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
		C = (int)ce.ids.size(), J = (int)je.ids.size();
		if (prepTimes)
		{
			uint64_t h = 1469598103934665603ull;
			h = fnv(h, ce.ids.data(), ce.ids.size() * sizeof(int));
			h = fnv(h, ce.a.data(), ce.a.size() * sizeof(int));
			h = fnv(h, ce.b.data(), ce.b.size() * sizeof(int));
			fprintf(stderr, "[s2amd] rebuild #%llu: %d potential contact constraints, %d joints, edge hash %016llx, reason: %s\n",
					(unsigned long long)s->structureGeneration, C, J, (unsigned long long)h, s->dirtyReason);
			s->dirtyReason = "";
		}

		phase("edge lists");
		return S2AMD_OK;
	}

	// ---- phase 2: islands = connected components over the writable bodies; the small ones are packed into LDS groups ----
	void findIslands()
	{
		cPart.assign((size_t)C, -1), jPart.assign((size_t)J, -1);
		groupCount = 0;
		flags = s->hBodyFlags;
		if (ldsGroups && (C > 0 || J > 0))
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
			// Islands of ~900 - 2,048 bodies fit one workgroup but are swept faster by strips (island_size_sweep, r6) -- as long as
			// there are few of them: strips need ALL their workgroups co-resident, and 64 pyramids of 1,830 bodies cut into 1,856 strips
			// fell off the persistent kernel (1.77 ms per TGS_Soft step; 0.43 as 64 groups side by side, which keep 64 CUs busy anyway).
			// So: the lower limit while the islands above it hold no more bodies than the strips can take.
			int groupLimitAll = s->optMaxGroupBodies;
			if (!s->maxGroupBodiesSet)
			{
				// (measured, tools/many_islands_table.py: strips hold up to 64 pyramids of 1,275 bodies -- 82k bodies in 240 strips,
				// 0.19 ms per TGS_Soft step against 0.36 as groups -- and 40 of 1,830; at 117k the partition no longer fits the kernel.
				// Under the op interpreter the groups catch up from ~32 such islands on: 0.17 against 0.22 ms.)
				const bool soft = isSoftFamily(solverType);
				const int lower = groupLowerLimit(solverType), upper = 2048;
				long long above = 0; // bodies of every island the lower limit sends to the strips
				for (int i = 0; i < nb; ++i)
				{
					if (islandBodies[(size_t)i] > lower)
					{
						above += islandBodies[(size_t)i];
					}
				}
				groupLimitAll = above <= (soft ? 81920 : 32768) ? lower : upper;
			}
			// How full a group is packed: a group is ONE workgroup, so the small islands of a world are spread over as many groups
			// as the GPU has CUs before any group gets a second helping (r6: 40 small pyramids packed 1,024 bodies to the group ran
			// on a dozen workgroups, and the fuller groups overflowed the resident kernel's 512 constraints per round: 0.54 ms per
			// TGS_Soft step, 0.27 at 640 bodies per group).  An island's sweep order does not depend on its group's company.
			int packBodies = s->optPackGroupBodies;
			if (!s->packGroupBodiesSet && s->cuCount > 0)
			{
				long long eligible = 0;
				for (int i = 0; i < nb; ++i)
				{
					if (islandBodies[(size_t)i] > 0 && islandBodies[(size_t)i] <= groupLimitAll)
					{
						eligible += islandBodies[(size_t)i];
					}
				}
				packBodies = (int)std::min<long long>(std::max<long long>(eligible / s->cuCount, 128), s->optPackGroupBodies);
			}
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
				// (the soft solvers' resident-island kernel takes 8 rounds of 512 constraints: a pyramid of 990 bodies falls to the
				// group interpreter, 0.28 ms per TGS_Soft step, where strips take 0.136 -- island_size_sweep, r6)
				const int groupLimit = groupLimitAll;
				if (n > groupLimit || !ldsGroups)
				{
					groupOfRoot[root] = -1;
					return -1;
				}
				if (groupCount == 0 || curBodies + n > packBodies)
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

		phase("islands and groups");
	}

	// ---- phase 3: per part lists (pool order is preserved inside every part), empty sweep sets ----
	void splitParts()
	{
		cOf.assign((size_t)groupCount + 1, {}), jOf.assign((size_t)groupCount + 1, {});
		for (int k = 0; k < C; ++k)
		{
			cOf[(size_t)cPart[k] + 1].push_back(k);
		}
		for (int k = 0; k < J; ++k)
		{
			jOf[(size_t)jPart[k] + 1].push_back(k);
		}

		cs = SweepSet();
		js = SweepSet();
		cs.colorOffsets.push_back(0);
		js.colorOffsets.push_back(0);
		s->hGroups.clear();
		s->hResident.clear();
		s->hContactTail.clear();
		s->hJointTail.clear();
		s->hStripA.clear();
		s->hStripB.clear();
	}

	// ---- phase 4: strips = the part that fits no LDS group, cut along BFS level sets ----
	void cutStrips()
	{
		strips = StripPartition();
		stripsNeedOneLaunch = false;
		if (wantStrips && (s->optStripsAnySolver != 0 || jOf[0].empty() || genericWanted(s)))
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
			// (an island the lower group limit sent here is worth its strips whatever its size: else it would fall to the colour batches)
			if (looseCount >= (s->stripMinBodiesSet || s->maxGroupBodiesSet ? s->optStripMinBodies : std::min(s->optStripMinBodies, groupLowerLimit(solverType))))
			{
				partitionStrips(ce, je, cOf[0], jOf[0], conflict, loose, nb, std::max(8, (int)((float)stripBodiesFor(s, solverType) * stripScale)), s->optMaxGroupBodies, strips);
			}
			if (strips.active)
			{
				// A body the sweeps WRITE that the partition did not place in any strip -- a static body whose rot is not a fixed
				// point of the normalisation is written by the position sweeps but is not one of the loose bodies the
				// breadth-first search walks -- would be written by every strip that touches it: no strips for this graph
				// (found by fuzzing: seed 259 under XPBD).
				std::vector<uint8_t> placed((size_t)nb, 0);
				for (const std::vector<int>& list : strips.bodies)
				{
					for (int b : list)
					{
						placed[(size_t)b] = 1;
					}
				}
				bool unfit = false; // an orphan body, or a hub
				for (const std::vector<std::vector<int>>* lists : {&strips.cA, &strips.cB})
				{
					for (const std::vector<int>& list : *lists)
					{
						for (int k : list)
						{
							unfit = unfit || (ce.a[k] >= 0 && conflict[(size_t)ce.a[k]] && !placed[(size_t)ce.a[k]]) ||
									 (ce.b[k] >= 0 && conflict[(size_t)ce.b[k]] && !placed[(size_t)ce.b[k]]);
						}
					}
				}
				for (const std::vector<std::vector<int>>* lists : {&strips.jA, &strips.jB})
				{
					for (const std::vector<int>& list : *lists)
					{
						for (int k : list)
						{
							unfit = unfit || (je.a[k] >= 0 && conflict[(size_t)je.a[k]] && !placed[(size_t)je.a[k]]) ||
									 (je.b[k] >= 0 && conflict[(size_t)je.b[k]] && !placed[(size_t)je.b[k]]);
						}
					}
				}
				// A strip sweeps the constraints of one body in as many colour rounds as the body has constraints: a hub (the
				// Tumbler's drum, 238 contacts) would hold its strip -- and every strip waiting on its hand-offs -- for hundreds
				// of rounds per sweep.  Such a graph stays on the colour batches and their wave-walked tail (group_kernel.hip:
				// walkTail; Tumbler 10k TGS_Soft: 3.2 ms there, 5.8 ms through the op interpreter).
				// Which way is the cheaper one is an estimate from measured unit costs (solver_internal.h: S2_COST_*), not a constant.
				std::vector<int> stripDegree((size_t)nb, 0);
				int maxDegree = 0;
				for (const std::vector<std::vector<int>>* lists : {&strips.cA, &strips.cB})
				{
					for (const std::vector<int>& list : *lists)
					{
						for (int k : list)
						{
							for (int b : {ce.a[k], ce.b[k]})
							{
								if (b >= 0 && conflict[(size_t)b])
								{
									maxDegree = std::max(maxDegree, ++stripDegree[(size_t)b]);
								}
							}
						}
					}
				}
				if (maxDegree > S2_HUB_DEGREE) // (up to there a body's constraints are colour rounds like any other's)
				{
					long tailVisits = 0;
					for (int d : stripDegree)
					{
						tailVisits += d > S2_HUB_DEGREE ? d : 0;
					}
					const float onStrips = (float)maxDegree * s->hubCosts.stripRoundUs;
					const float onBatches = (float)S2_COST_BATCH_COLOURS * s->hubCosts.launchUs + (float)tailVisits * s->hubCosts.tailVisitUs;
					unfit = unfit || onStrips > onBatches;
				}
				if (unfit)
				{
					strips = StripPartition();
					s->stripsHopeless = true; // (whatever the strip width: the search over widths is skipped, solver_structure.cpp: buildStructure)
					s->stripsRejected = true;
				}
			}
			if (strips.active)
			{
				// A body the sweeps do not write but the body stages MOVE (kinematic, massless) is a replica in every strip
				// that touches it.  One launch per step (the persistent kernel) keeps such a copy consistent from start to end;
				// with one launch per sweep every workgroup re-reads it from HBM while its owner is writing it in the same
				// launch -- a race.  Such partitions only run on the persistent kernel (found by fuzzing: seed 205).
				for (const std::vector<std::vector<int>>* lists : {&strips.cA, &strips.cB})
				{
					for (const std::vector<int>& list : *lists)
					{
						for (int k : list)
						{
							for (int b : {ce.a[k], ce.b[k]})
							{
								if (b >= 0 && !conflict[(size_t)b] && s->hBodyLive[(size_t)b] && !s->hBodyStatic[(size_t)b])
								{
									stripsNeedOneLaunch = true;
								}
							}
						}
					}
				}
			}
			if (strips.active && getenv("S2AMD_DEBUG_CHECK"))
			{
				// the invariants the strip kernels rely on (comment above StripPartition)
				std::vector<int> owner((size_t)nb, -1), seamOf((size_t)nb, -1);
				for (size_t i = 0; i < strips.bodies.size(); ++i)
				{
					for (int b : strips.bodies[i])
					{
						if (owner[(size_t)b] != -1)
						{
							fprintf(stderr, "[s2amd] CHECK: body %d owned by strips %d and %zu\n", b, owner[(size_t)b], i);
						}
						owner[(size_t)b] = (int)i;
					}
				}
				for (size_t i = 0; i < strips.cA.size(); ++i)
				{
					for (int k : strips.cA[i])
					{
						for (int b : {ce.a[k], ce.b[k]})
						{
							if (b >= 0 && conflict[(size_t)b] && owner[(size_t)b] != (int)i)
							{
								fprintf(stderr, "[s2amd] CHECK: interior constraint %d of strip %zu touches body %d of strip %d\n", k, i, b, owner[(size_t)b]);
							}
						}
					}
				}
				for (size_t i = 0; i < strips.cB.size(); ++i)
				{
					for (int k : strips.cB[i])
					{
						for (int b : {ce.a[k], ce.b[k]})
						{
							if (b < 0 || !conflict[(size_t)b])
							{
								continue;
							}
							if (owner[(size_t)b] != (int)i && owner[(size_t)b] != (int)i + 1)
							{
								fprintf(stderr, "[s2amd] CHECK: seam %zu constraint %d touches body %d of strip %d\n", i, k, b, owner[(size_t)b]);
							}
							if (seamOf[(size_t)b] != -1 && seamOf[(size_t)b] != (int)i)
							{
								fprintf(stderr, "[s2amd] CHECK: body %d (strip %d) is touched by seams %d and %zu\n", b, owner[(size_t)b], seamOf[(size_t)b], i);
							}
							seamOf[(size_t)b] = (int)i;
						}
					}
				}
				size_t total = 0;
				for (size_t i = 0; i < strips.cA.size(); ++i)
				{
					total += strips.cA[i].size();
				}
				for (size_t i = 0; i < strips.cB.size(); ++i)
				{
					total += strips.cB[i].size();
				}
				fprintf(stderr, "[s2amd] CHECK: %zu strips, %zu seams, %zu constraints of %zu in the strip part\n", strips.bodies.size(), strips.cB.size(), total, cOf[0].size());
			}
			if (strips.active)
			{
				cOf[0].clear();
				jOf[0].clear();
			}
		}
	}

	// ---- phase 5: the global part = colour batches over HBM-resident bodies (+ a sequential tail as a one-group LDS table) ----
	void colourGlobalPart()
	{
		std::vector<int> ids, a, b, pos;
		gather(ce, cOf[0], ids, a, b);
		s->inc = IncrementalGlobal();
		const bool slack = s->optIncremental != 0 && s->optMessage == 0;
		s->inc.ignoreColours = needAdj;
		colourPart(ids, a, b, conflict, nb, cs, cs.batchOffsets, cs.hasTail, &pos, 0, slack ? &s->inc : nullptr, needAdj ? 0 : s->spareColours, s->slackShift, needAdj, 0,
				   S2_TAIL_SLACK << s->tailSlackShift, s->optTailTinyColour);
		cs.globalCount = (int)cs.order.size(); // (with the free positions of the slack layout)
		cs.local.assign((size_t)cs.globalCount, make_int2(0, 0));
		if (slack)
		{
			s->inc.solverClass = cls;
			s->inc.positionOfSlot.assign((size_t)s->contactCapacity, -1);
			for (int k = 0; k < cs.globalCount; ++k)
			{
				if (cs.order[(size_t)k] >= 0)
				{
					s->inc.positionOfSlot[(size_t)cs.order[(size_t)k]] = k;
				}
			}
		}
		if (cs.hasTail)
		{
			HostGroupTable& t = s->hContactTail;
			int begin = cs.batchOffsets[cs.batchOffsets.size() - 2], end = cs.batchOffsets.back();
			slots.begin();
			std::vector<int> bodies;
			for (int k = begin; k < end; ++k)
			{
				int p = pos[(size_t)k];
				if (p < 0)
				{
					continue; // (a free position of the tail's slack: IncrementalGlobal::tailFree)
				}
				cs.local[(size_t)k] = make_int2(slots.get(a[p], bodies, conflict), slots.get(b[p], bodies, conflict));
			}
			s->inc.tailBodySlot.clear();
			s->inc.tailBodyCount = s->inc.tailBodyCapacity = 0;
			const int tailBodySlack = S2_TAIL_BODY_SLACK << s->tailSlackShift;
			if ((int)bodies.size() + tailBodySlack > 2800)
			{
				s->inc.tailBegin = s->inc.tailEnd = 0;
				s->inc.tailFree.clear();
				// The tail's bodies do not fit one workgroup's LDS (160 KiB at 40-56 B per body): no sequential tail for
				// this graph, its tiny colours are launched one by one like the others (a dense pool -- every pair whose fat
				// boxes overlap is a potential constraint -- can put thousands of constraints into the tiny colours)
				cs.hasTail = false;
				cs.batchOffsets.pop_back();
				for (size_t ci = 0; ci < cs.colorOffsets.size(); ++ci)
				{
					if (cs.colorOffsets[ci] > begin)
					{
						cs.batchOffsets.push_back(cs.colorOffsets[ci]);
					}
				}
				std::fill(cs.local.begin(), cs.local.end(), make_int2(0, 0));
			}
			else
			{
				t.bodyIds = bodies;
				t.bodyOffsets = {0, (int)bodies.size()};
				t.cBatches.push_back(make_int4(begin, end, 1, 0));
				t.cBatchOffsets = {0, 1};
				t.jBatchOffsets = {0, 0};
				const bool tailSlack = slack && !s->inc.tailFree.empty();
				t.maxBodies = (int)bodies.size() + (tailSlack ? tailBodySlack : 0); // (the launch's LDS: room for the bodies created contacts bring along)
				if (tailSlack)
				{
					for (size_t i = 0; i < bodies.size(); ++i)
					{
						s->inc.tailBodySlot[(int)((uint32_t)bodies[i] & ~S2G_OWNED)] = (int)i;
					}
					s->inc.tailBodyCount = (int)bodies.size(), s->inc.tailBodyCapacity = (int)bodies.size() + tailBodySlack;
				}
			}
		}
		else
		{
			s->inc.tailBegin = s->inc.tailEnd = 0;
			s->inc.tailFree.clear();
			s->inc.tailBodySlot.clear();
			s->inc.tailBodyCount = s->inc.tailBodyCapacity = 0;
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
	void emitGroup(HostGroupTable& t, const std::vector<int>& cKs, const std::vector<int>& jKs, const std::vector<int>& seedBodies,
				   const std::vector<int>* replicaOf = nullptr, const std::vector<int>* replicaOf2 = nullptr)
	{
		std::vector<int> ids, a, b, bodies, la, lb, pos, batchOffsets;
		bool tail = false;
		const double tg0 = prepTimes ? nowMs() : 0.0;
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
		// strips: a round is one constraint per thread of a 256-thread workgroup, wider colour classes are evened out and cut.
		// LDS groups and resident islands keep the plain greedy colouring: an island's sweep order must not depend on which
		// other islands share its group (a world sharded over several GPUs packs them differently and must sweep the same).
		const bool stripTable = &t == &s->hStripA || &t == &s->hStripB;
		const int roundSlack = (stripTable && stripSlackWanted && jKs.empty()) ? 16 : 0;
		const double tg1 = prepTimes ? nowMs() : 0.0;
		// (a colour of an LDS group or a strip is a barrier, not a launch: only colours of three constraints or fewer are worth a sequential tail)
		const bool ldsTable = &t == &s->hGroups || &t == &s->hResident;
		colourPart(ids, la, lb, lconf, (int)bodies.size(), cs, batchOffsets, tail, &pos, ldsTable ? 0 : 256, nullptr, 0, 0, false, roundSlack, S2_TAIL_SLACK,
				   s->optGroupTinyColour);
		const double tg2 = prepTimes ? nowMs() : 0.0;
		tSlots += tg1 - tg0, tColour += tg2 - tg1;
		for (size_t i = 0; i < pos.size(); ++i)
		{
			cs.local.push_back(pos[i] >= 0 ? make_int2(la[(size_t)pos[i]], lb[(size_t)pos[i]]) : make_int2(0, 0));
		}
		for (size_t bi = 0; bi + 1 < batchOffsets.size(); ++bi)
		{
			bool isTail = tail && bi + 2 == batchOffsets.size();
			if (batchOffsets[bi + 1] > batchOffsets[bi])
			{
				t.cBatches.push_back(make_int4(batchOffsets[bi], batchOffsets[bi + 1], isTail ? 1 : 0, 0));
			}
		}
		if (roundSlack > 0 && stripTable)
		{
			// a strip (a seam) with fewer rounds than the kernels take keeps positions for the rounds it does not have yet (eight each): a created contact whose
			// bodies have every round taken opens the next (solver_incremental.cpp: stripPlace patches the descriptors); until then no
			// round covers them and they cost nothing.  (A ball that comes to rest in the pile is a seventh and eighth constraint on the
			// boxes it touches, a third on a seam.)
			const int rounds = (int)t.cBatches.size() - t.cBatchOffsets.back();
			const int most = &t == &s->hStripA ? S2_STRIP_ROUNDS_MAX : S2_PERSIST_B_ROUNDS;
			for (int r = rounds; rounds > 0 && r < most; ++r)
			{
				const int begin = (int)cs.order.size(), cap = 8;
				cs.order.resize(cs.order.size() + (size_t)cap, -1);
				cs.local.resize(cs.local.size() + (size_t)cap, make_int2(0, 0));
				cs.colorOffsets.push_back(begin + cap);
				spareRounds.push_back(SpareRound{&t == &s->hStripA ? 0 : 1, t.count(), begin, cap, r}); // (t.count(): this group's index, its row is closed below)
			}
		}
		t.cBatchOffsets.push_back((int)t.cBatches.size());
		colourPart(jids, jla, jlb, lconf, (int)bodies.size(), js, batchOffsets, tail, &pos, 0, nullptr, 0, 0, false, 0, S2_TAIL_SLACK, s->optGroupTinyColour);
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
		tAppend += prepTimes ? nowMs() - tg2 : 0.0;
	}

	// does the greedy colouring of this group fit the resident kernel: at most 8 rounds of at most 512 constraints, no
	// sequential tail?  (The same colouring emitGroup will produce: a dry run on the pool indices.)
	bool fitsResident(const std::vector<int>& cKs) const
	{
		std::vector<int> a, b, color;
		a.reserve(cKs.size()), b.reserve(cKs.size());
		for (int k : cKs)
		{
			a.push_back(ce.a[k]), b.push_back(ce.b[k]);
		}
		const int cc = colorGraph(a, b, conflict, nb, color);
		if (cc > S2_STRIP_ROUNDS_MAX)
		{
			return false;
		}
		int population[S2_STRIP_ROUNDS_MAX] = {0};
		for (int c : color)
		{
			if (++population[c] > 512)
			{
				return false;
			}
		}
		return true;
	}

	// ---- phase 6: LDS groups (whole-step kernel, bodies in LDS) ... or, under the soft contact solvers, RESIDENT islands: the
	// same groups with their constraints in the registers of one 512-thread workgroup for the whole step (strip_kernel.hip:
	// islandStepKernel) -- contact-only groups whose colouring fits the kernel's rounds; the others stay plain LDS groups
	void emitIslandGroups()
	{
		const std::vector<int> noSeed;
		std::vector<uint8_t> residentGroup((size_t)groupCount, 0);
		for (int g = 0; g < groupCount; ++g)
		{
			const std::vector<int>& cKs = cOf[(size_t)g + 1];
			residentGroup[(size_t)g] = residentWanted && jOf[(size_t)g + 1].empty() && !cKs.empty() && cKs.size() <= (size_t)S2_STRIP_ROUNDS_MAX * 512 && fitsResident(cKs);
			if (!residentGroup[(size_t)g])
			{
				emitGroup(s->hGroups, cKs, jOf[(size_t)g + 1], noSeed);
			}
		}
		s->residentK0 = (int)cs.order.size(); // the resident islands' constraints are one range of the sweep order
		for (int g = 0; g < groupCount; ++g)
		{
			const std::vector<int>& cKs = cOf[(size_t)g + 1];
			if (residentGroup[(size_t)g])
			{
				std::vector<int> seed; // the bodies the group owns (its sweeps write them), in order of first use: they lead the body list
				slots.begin();
				for (int k : cKs)
				{
					for (int body : {ce.a[k], ce.b[k]})
					{
						if (body >= 0 && conflict[(size_t)body] && slots.stamp[(size_t)body] != slots.epoch)
						{
							slots.stamp[(size_t)body] = slots.epoch;
							seed.push_back(body);
						}
					}
				}
				emitGroup(s->hResident, cKs, jOf[(size_t)g + 1], seed);
			}
		}
		s->residentK1 = (int)cs.order.size();
	}

	// ---- phase 7: strips of the big islands: phase A = interiors (own every body of the strip), phase B = seams ----
	void emitStrips()
	{
		const std::vector<int> noSeed;
		stripBaseC = (int)cs.order.size();
		const int stripBaseJ = (int)js.order.size();
		for (size_t i = 0; i < strips.bodies.size(); ++i)
		{
			emitGroup(s->hStripA, strips.cA[i], strips.jA[i], strips.bodies[i], i < strips.cB.size() ? &strips.cB[i] : nullptr,
					  i > 0 ? &strips.cB[i - 1] : nullptr);
		}
		int stripInterior = (int)cs.order.size(), stripInteriorJ = (int)js.order.size();
		seamGroup.assign(strips.cB.size(), -1);
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

		if (!s->inc.positionOfSlot.empty())
		{
			for (size_t k = (size_t)cs.globalCount; k < cs.order.size(); ++k)
			{
				if (cs.order[k] >= 0)
				{
					s->inc.positionOfSlot[(size_t)cs.order[k]] = -2; // lives in an LDS group or a strip (IncrementalStrips knows where, for strips)
				}
			}
		}
		// Bodies that carry no constraint of this structure at all (a ball in flight) would be the ONLY reason for the global path's
		// launches -- one per body stage, ~25 per TGS_Soft step -- next to groups or strips that take one launch for everything:
		// they become LDS groups of their own (body stages only).  With constraints in the global part they ride along there.
		if (grouped && s->optFreeBodyGroups && cs.globalCount == 0 && js.globalCount == 0 && (s->hGroups.count() > 0 || s->hResident.count() > 0 || strips.active))
		{
			std::vector<int> freeBodies;
			for (int i = 0; i < nb; ++i)
			{
				// (only bodies the sweeps could write: a kinematic body may be a read-only replica in other groups of the same launch, which
				// must not see its owner's write-back -- it stays with the body kernels that run after that launch)
				if (s->hBodyLive[i] && !s->hBodyStatic[i] && conflict[(size_t)i] && (flags[i] & S2F_IN_GROUP) == 0)
				{
					freeBodies.push_back(i);
					if ((int)freeBodies.size() == s->optMaxGroupBodies)
					{
						emitGroup(s->hGroups, noSeed, noSeed, freeBodies);
						freeBodies.clear();
					}
				}
			}
			if (!freeBodies.empty())
			{
				emitGroup(s->hGroups, noSeed, noSeed, freeBodies);
			}
		}
		s->hBodyFlagsFinal = flags;
		// (which of the bodies with S2F_IN_GROUP an LDS group or a resident island holds -- their tables take no created contact --, as
		// against a strip: what SolverRest::groupPatienceNow counts)
		s->hBodyLdsOwned.assign((size_t)nb, 0);
		for (const HostGroupTable* t : {&s->hGroups, &s->hResident})
		{
			for (int id : t->bodyIds)
			{
				if ((uint32_t)id & S2G_OWNED)
				{
					s->hBodyLdsOwned[(size_t)((uint32_t)id & ~S2G_OWNED)] = 1;
				}
			}
		}
		s->looseBodies = 0;
		for (int i = 0; i < nb; ++i)
		{
			if (s->hBodyLive[i] && !s->hBodyStatic[i] && (flags[i] & S2F_IN_GROUP) == 0)
			{
				s->looseBodies += 1;
			}
		}
		// the overflow region (solver_internal.h: IncrementalStrips): free positions behind everything else, one colour batch each, for
		// created contacts that fit nowhere in the strips -- where a worker thread can build the structure that will hold them (the world
		// chain and its workers' copies) and the step is not the self-contained strip kernel's (which is its own prologue and epilogue)
		overflowBase = (int)cs.order.size(), overflowCount = 0;
		if (strips.active && stripSlackWanted && js.stripCount == 0 && s->optOverflow != 0 && s->optAsyncBuild != 0 && (s->worldResident || s->isClone) &&
			s->optSelfContainedStrips == 0 && isSoftFamily(solverType))
		{
			overflowCount = S2_OVERFLOW_SLACK;
			for (int i = 0; i < overflowCount; ++i)
			{
				cs.order.push_back(-1);
				cs.local.push_back(make_int2(0, 0));
				cs.colorOffsets.push_back((int)cs.order.size());
			}
		}

		if (prepTimes)
		{
			fprintf(stderr, "[s2amd]   inside the groups: body slots %.3f ms, colouring %.3f ms, tables %.3f ms\n", tSlots, tColour, tAppend);
		}
		phase("colours, batches");
	}

	// ---- phase 8: device tables ----
	int uploadTables()
	{
		int rc;
		const int CP = (int)cs.order.size(); // positions of the sweep order: the C potential constraints + the global part's free positions
		if ((rc = carveContacts(s, CP)) != 0 || (rc = carveJoints(s, J)) != 0)
		{
			return rc;
		}
		bool grew = false;
		if ((rc = s->dContactIndex.ensure((size_t)std::max(CP, 1) * sizeof(int), &grew)) != 0 ||
			(rc = s->dJointIndex.ensure((size_t)std::max(J, 1) * sizeof(int), &grew)) != 0 ||
			(rc = s->dContactLocal.ensure((size_t)std::max(CP, 1) * sizeof(int2), &grew)) != 0 ||
			(rc = s->dJointLocal.ensure((size_t)std::max(J, 1) * sizeof(int2), &grew)) != 0)
		{
			return rc;
		}
		if (grew)
		{
			s->layoutGeneration += 1;
		}
		if (CP > 0)
		{
			HIP_TRY(hipMemcpyAsync(s->dContactIndex.p, cs.order.data(), (size_t)CP * sizeof(int), hipMemcpyHostToDevice, s->stream));
			HIP_TRY(hipMemcpyAsync(s->dContactLocal.p, cs.local.data(), (size_t)CP * sizeof(int2), hipMemcpyHostToDevice, s->stream));
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
		if (s->worldResident)
		{
			if ((rc = s->dWatched.ensure(std::max<size_t>((size_t)s->contactCapacity, 256), &grew)) != 0)
			{
				return rc;
			}
			if (s->contactCapacity > 0)
			{
				HIP_TRY(hipMemcpyAsync(s->dWatched.p, s->hContactWatched.data(), (size_t)s->contactCapacity, hipMemcpyHostToDevice, s->stream));
				s->watchedDirty = false;
			}
		}
		s->cv.contactIndex = (int*)s->dContactIndex.p;
		s->cv.localBodies = (int2*)s->dContactLocal.p;
		s->cv.count = CP;
		s->cv.skipBegin = s->cv.skipEnd = 0; // (set per step by the executor when the resident-island kernel runs)
		s->jv.jointIndex = (int*)s->dJointIndex.p;
		s->jv.localBodies = (int2*)s->dJointLocal.p;
		s->jv.count = J;
		if ((rc = uploadGroupTable(s, s->hResident, s->dResident)) != 0 || (rc = buildResidentTables(s)) != 0)
		{
			return rc;
		}
		if (s->residentRejected && s->hResident.count() > 0)
		{
			// some island needs more colour rounds than the resident kernel holds: all of them as plain LDS groups for this graph
			s->structureDirty = true;
			rebuild = true;
			return S2AMD_OK;
		}
		if ((rc = uploadGroupTable(s, s->hGroups, s->dGroups)) != 0 || (rc = uploadGroupTable(s, s->hContactTail, s->dContactTail, s->hContactTail.count() > 0 ? (size_t)(S2_TAIL_BODY_SLACK << s->tailSlackShift) : 0)) != 0 ||
			(rc = uploadGroupTable(s, s->hJointTail, s->dJointTail)) != 0 ||
			(rc = uploadGroupTable(s, s->hStripA, s->dStripA, s->hStripA.count() > 0 ? (size_t)32 * (size_t)(s->hStripA.maxBodies + S2_STRIP_ADOPT_SLACK) : 0)) != 0 ||
			(rc = uploadGroupTable(s, s->hStripB, s->dStripB)) != 0)
		{
			return rc;
		}
		phase("index tables");
		return S2AMD_OK;
	}

	// ---- phase 9: lean strip tables: per-group descriptors + warm-start slots (strip_kernel.hip) ----
	int buildStripTables()
	{
		int rc;
		s->leanAValid = s->leanBValid = false;
		s->persistValid = false;
		s->leanA = StripTableView{};
		s->leanB = StripTableView{};
		if (strips.active && s->optStripLean)
		{
			if ((rc = buildLeanStripTables(s, strips, conflict, seamGroup, stripBaseC, nb)) != 0)
			{
				return rc;
			}
		}

		// strips only pay through the strip kernels: when neither the persistent step nor the lean launches can take this
		// partition (too many colours, a hub body, LDS budget), fall back to the colour-batch structure for this graph
		s->stripsNeedOneLaunch = strips.active && stripsNeedOneLaunch;
		if (stripsNeedOneLaunch)
		{
			s->leanAValid = s->leanBValid = false;
		}
		// what can run this partition: one launch (the soft drivers' resident kernels on a joint-free partition, the op
		// interpreter for everything else) or the multi-launch strip path (lean soft launches; the group interpreter in tests)
		const bool soft = isSoftFamily(solverType);
		const bool oneLaunch = (soft && s->persistValid) || s->genericValid;
		const bool multiLaunch = !stripsNeedOneLaunch && (s->optStripsAnySolver != 0 || (soft && js.stripCount == 0 && s->leanAValid && s->leanBValid));
		if (strips.active && !oneLaunch && !multiLaunch)
		{
			s->stripsRejected = true;
			s->structureDirty = true;
			rebuild = true;
			return S2AMD_OK;
		}
		phase("strip tables");
		buildStripMirror();
		phase("strip mirror");
		return S2AMD_OK;
	}

	// the host's picture of the strips' rounds for placing created contacts into them (solver_internal.h: IncrementalStrips)
	void buildStripMirror()
	{
		IncrementalStrips& m = s->stripInc;
		m = IncrementalStrips{};
		const bool interpreter = !isSoftFamily(solverType);
		if (!strips.active || !stripSlackWanted || !(interpreter ? s->genericValid : s->persistValid) || js.stripCount != 0)
		{
			return;
		}
		m.base = stripBaseC, m.end = stripBaseC + cs.stripCount;
		if (interpreter)
		{
			m.takeOnly = true;
			m.roundLimit[0] = m.roundLimit[1] = 0;
		}
		if (solverType == s2amd_solverSoftStep)
		{
			m.roundLimit[0] = S2_STRIP_ROUNDS, m.roundLimit[1] = 2;
		}
		m.roundOfPosition.assign((size_t)cs.stripCount, -1);
		m.ownerStrip.assign((size_t)nb, -1), m.ownerSlot.assign((size_t)nb, -1);
		m.positionOfSlot.assign((size_t)s->contactCapacity, -1);
		m.seamGroupOf = seamGroup;
		const HostGroupTable* tables[2] = {&s->hStripA, &s->hStripB};
		m.replicaSlot.resize((size_t)s->hStripA.count());
		m.seamSlot.resize((size_t)s->hStripB.count());
		for (int t = 0; t < 2; ++t)
		{
			const HostGroupTable& h = *tables[t];
			m.roundsOf[t].assign((size_t)h.count(), std::vector<int>()), m.spareOf[t].assign((size_t)h.count(), std::vector<int>());
			m.bodyOffset[t].assign(h.bodyOffsets.begin(), h.bodyOffsets.end());
			m.roundMask[t].assign(h.bodyIds.size(), 0u);
			for (int g = 0; g < h.count(); ++g)
			{
				const int b0 = h.bodyOffsets[(size_t)g], b1 = h.bodyOffsets[(size_t)g + 1];
				for (int e = b0; e < b1; ++e)
				{
					const int body = (int)((uint32_t)h.bodyIds[(size_t)e] & ~S2G_OWNED);
					if (t == 1)
					{
						m.seamSlot[(size_t)g][body] = e - b0;
					}
					else if (conflict[(size_t)body])
					{
						m.ownerStrip[(size_t)body] = g, m.ownerSlot[(size_t)body] = e - b0;
					}
					else
					{
						m.replicaSlot[(size_t)g][body] = e - b0;
					}
				}
				for (int bi = h.cBatchOffsets[(size_t)g]; bi < h.cBatchOffsets[(size_t)g + 1]; ++bi)
				{
					const int4 bt = h.cBatches[(size_t)bi];
					IncrementalStrips::Round r;
					r.table = t, r.group = g, r.round = bi - h.cBatchOffsets[(size_t)g];
					if (bt.x < m.base || bt.y > m.end || r.round >= 32)
					{
						m = IncrementalStrips{};
						return;
					}
					for (int k = bt.y - 1; k >= bt.x; --k)
					{
						m.roundOfPosition[(size_t)(k - m.base)] = (int)m.rounds.size();
						const int slot = cs.order[(size_t)k];
						if (slot < 0)
						{
							r.freePositions.push_back(k);
							continue;
						}
						m.positionOfSlot[(size_t)slot] = k;
						const int2 l = cs.local[(size_t)k];
						for (int side = 0; side < 2; ++side)
						{
							const int ls = side ? l.y : l.x;
							const int body = (int)((uint32_t)h.bodyIds[(size_t)(b0 + ls)] & ~S2G_OWNED);
							if (conflict[(size_t)body])
							{
								m.roundMask[t][(size_t)(b0 + ls)] |= 1u << r.round;
							}
						}
					}
					m.roundsOf[t][(size_t)g].push_back((int)m.rounds.size());
					m.rounds.push_back(std::move(r));
				}
			}
		}
		m.seamOfGroup.assign((size_t)s->hStripB.count(), -1);
		for (size_t sm = 0; sm < seamGroup.size(); ++sm)
		{
			if (seamGroup[sm] >= 0 && seamGroup[sm] < (int)m.seamOfGroup.size())
			{
				m.seamOfGroup[(size_t)seamGroup[sm]] = (int)sm;
			}
		}
		// where every strip's body list is, and the room behind the table for lists that move (a strip that adopts a body)
		{
			const HostGroupTable& h = s->hStripA;
			const int K = h.count();
			m.stripBodyBase.assign((size_t)K, 0), m.stripBodyCount.assign((size_t)K, 0), m.stripListCapacity.assign((size_t)K, 0);
			m.movedList.assign((size_t)K, std::vector<int>()), m.adoptedBy.assign((size_t)K, 0);
			for (int g = 0; g < K; ++g)
			{
				m.stripBodyBase[(size_t)g] = h.bodyOffsets[(size_t)g];
				m.stripBodyCount[(size_t)g] = m.stripListCapacity[(size_t)g] = h.bodyOffsets[(size_t)g + 1] - h.bodyOffsets[(size_t)g];
			}
			m.spareIdsNext = s->dStripA.spareIdsBase, m.spareIdsEnd = s->dStripA.spareIdsBase + s->dStripA.spareIdsCount;
			const HostGroupTable& hb = s->hStripB;
			m.seamBodyCount.assign((size_t)hb.count(), 0);
			m.seamExtra[0].assign((size_t)hb.count(), 0), m.seamExtra[1].assign((size_t)hb.count(), 0);
			for (int g = 0; g < hb.count(); ++g)
			{
				m.seamBodyCount[(size_t)g] = hb.bodyOffsets[(size_t)g + 1] - hb.bodyOffsets[(size_t)g];
			}
		}
		// the closed spare rounds: no batch covers their positions yet
		for (const SpareRound& sp : spareRounds) // (in emission order: a group's spare rounds by round index)
		{
			if (sp.group >= (int)m.spareOf[sp.table].size() || sp.begin < m.base || sp.begin + sp.count > m.end)
			{
				continue;
			}
			IncrementalStrips::Round r;
			r.table = sp.table, r.group = sp.group, r.round = sp.round;
			for (int k = sp.begin + sp.count - 1; k >= sp.begin; --k)
			{
				m.roundOfPosition[(size_t)(k - m.base)] = (int)m.rounds.size();
				r.freePositions.push_back(k);
			}
			m.spareOf[sp.table][(size_t)sp.group].push_back((int)m.rounds.size());
			m.rounds.push_back(std::move(r));
		}
		m.overflowBegin = overflowBase, m.overflowEnd = overflowBase + overflowCount;
		m.overflowBodyIds.clear();
		for (int k = m.overflowEnd - 1; k >= m.overflowBegin; --k)
		{
			m.overflowFree.push_back(k);
		}
		m.valid = true;
	}

	// ---- phase 10: message-passing tables of the global part (see MsgBodies), body adjacency, bookkeeping ----
	int finish()
	{
		int rc;
		s->msgTablesValid = false;
		if (s->optMessage != 0 && cs.globalCount > 0 && js.globalCount == 0 && !cs.hasTail && !needAdj) // 0.25 ms of host time at 60k constraints
		{
			if ((rc = buildMessageTables(s, nb)) != 0)
			{
				return rc;
			}
		}

		phase("message tables");
		s->adjValid = false;
		if ((rc = buildAdjacency(s, conflict, nb)) != 0 || (rc = buildJointAdjacency(s, nb)) != 0)
		{
			return rc;
		}
		// the staging vectors above die with this scope: hipMemcpyAsync from pageable host memory
		// copies through a staging buffer before it returns, so that is safe
		HIP_TRY(hipStreamSynchronize(s->stream));
		phase("adjacency + sync");
		s->orderColourless = needAdj;
		// s2Solve_Jacobi: the persistent launch's tables, where the world qualifies -- at once for a snapshot and for a world's first
		// structure; a world chain that has just rebuilt (a burst of created contacts: the Tumbler filling rebuilds every other step, and
		// the first contact placed without colours puts the step back on the multi-launch path anyway) waits until the structure has
		// lived for a few steps: the tables cost as much host time as the rest of the build (1.2-1.8 ms at 10k constraints)
		s->jacobiValid = false;
		s->jacobi = JacobiView{};
		s->jacobiDeferred = 0;
		if (needAdj && s->worldResident && !s->isClone && s->structureGeneration > 0 && s->optJacobiPersist != 0)
		{
			s->jacobiDeferred = 4;
		}
		else if ((rc = buildJacobiBlocks(s)) != 0)
		{
			return rc;
		}
		phase("jacobi blocks");

		// created contacts can be placed into this structure while its global part has the slack layout
		s->slackPositions = 0;
		for (size_t k = 0; k < cs.order.size(); ++k) // (the global part's slack and the strips')
		{
			s->slackPositions += cs.order[k] < 0 ? 1 : 0;
		}
		s->slackAtBuild = s->slackPositions;
		s->inc.valid = s->optIncremental != 0 && s->inc.solverClass == cls && !s->msgTablesValid && !s->inc.positionOfSlot.empty();
		s->inc.patches.clear();
		s->orderSolverClass = cls;
		s->orderResident = residentWanted;
		s->orderColourless = needAdj;
		s->orderGrouped = grouped;
		s->orderLdsGroups = ldsGroups;
		if (getenv("S2AMD_DEBUG"))
		{
			int real = 0;
			for (int k = 0; k < cs.globalCount; ++k)
			{
				real += cs.order[(size_t)k] >= 0 ? 1 : 0;
			}
			fprintf(stderr, "[s2amd] structure: global part %d contact positions (%d constraints, %d batches%s), %d joints; %d LDS groups, %d resident, %d strips; loose bodies %d\n",
					cs.globalCount, real, (int)cs.batchOffsets.size() - 1, cs.hasTail ? ", tail" : "", js.globalCount, s->hGroups.count(), s->hResident.count(),
					s->hStripA.count(), s->looseBodies);
		}
		s->residentAllTwoPoints = residentAllTwoPoints(s) ? 1 : 0;
		s->orderStrips = wantStrips;
		s->orderStripBodies = stripBodiesFor(s, solverType);
		s->structureDirty = false;
		s->structureGeneration += 1;
		s->stats.hostPrepMs = (float)(nowMs() - t0);
		return S2AMD_OK;
	}
};

static int buildStructureWith(s2amdSolver* s, int solverType, float stripScale)
{
	StructureBuild build(s, solverType, stripScale);
	if (build.upToDate())
	{
		return S2AMD_OK;
	}
	if (s->worldResident && !s->pointsKnown)
	{
		// world chain: the pair slots this step's (already enqueued) stage 3 and the earlier ones freed leave the structure
		// with this rebuild, exactly as a host that ran stage 3 itself would have dropped them from the arrays it uploads
		int rcDead = syncDeadSlots(s);
		if (rcDead)
		{
			return rcDead;
		}
	}
	int rc = build.gatherEdges();
	if (rc)
	{
		return rc;
	}
	build.findIslands();
	build.splitParts();
	build.cutStrips();
	build.phase("strip partition");
	build.colourGlobalPart();
	build.phase("global part");
	build.emitIslandGroups();
	build.emitStrips();
	if ((rc = build.uploadTables()) != 0 || build.rebuild || (rc = build.buildStripTables()) != 0 || build.rebuild)
	{
		return rc ? rc : buildStructureWith(s, solverType, stripScale);
	}
	return build.finish();
}

// The strip partition is a heuristic cut (BFS levels, `strip_bodies` per strip) and the colouring of what it cuts out
// decides which persistent kernel variant can run: one strip that needs a seventh or eighth interior colour puts every
// workgroup on the 8-round variant, which spills (0.30 vs 0.26 ms at base-200).  A build that ends there is repeated
// with a few other strip widths; the first partition that needs six rounds wins, else the original one stays.  This
// runs only in the (rare) steps that build the strip structure at all.
int buildStructure(s2amdSolver* s, int solverType)
{
	if (asyncBuildsOn(s) && s->optAsyncBuild >= 2 && !s->structureDirty)
	{
		StructureBuild probe(s, solverType, 1.0f);
		if (probe.onlyStripsMissing())
		{
			// the strip structure is due (milliseconds of host time): a worker thread builds it on a copy, this step and the next few
			// run on the colour batches there are.  (Option "async_build" 2; measured a LOSS on the wrecking-ball world at base 200 --
			// every build costs `async_build_delay` more steps on the colour batches, and half the builds are overtaken by the graph:
			// 164 instead of 207 of 240 steps on the persistent kernel, median 0.79 against 0.56 ms -- so by default only the search
			// over strip widths, which improves on strips that already run, goes to the worker.)
			return asyncRequest(s, solverType, false);
		}
	}
	const uint64_t before = s->structureGeneration;
	// (a world whose partition was searched for once keeps the width that won: a graph that changes is not searched again as
	// long as that width still gives the persistent kernel something it can run)
	{
		const int cls = isPositionSolver(solverType) ? 1 : 0;
		if (s->stripsJudgedForClass != cls)
		{
			s->stripsRejected = s->stripsHopeless = false; // (rejected under the other class's set of writable bodies)
			s->stripsJudgedForClass = cls;
		}
	}
	if (s->stripScaleFoundFor != StructureBuild::stripBodiesFor(s, solverType))
	{
		s->stripScaleFound = 0.0f; // (found for another strip width: SoftStep / PGS_Soft against the rest)
		s->stripScaleFoundFor = StructureBuild::stripBodiesFor(s, solverType);
	}
	const float firstScale = s->stripScaleFound > 0.0f ? s->stripScaleFound : 1.0f;
	int rc = buildStructureWith(s, solverType, firstScale);
	if (rc != S2AMD_OK || s->structureGeneration == before || s->optStripRetry == 0)
	{
		return rc;
	}
	// outcome of a build that tried strips: 3 = persistent step with at most five interior and two seam colour rounds per sweep
	// (strips of exactly two BFS levels: no body has all six neighbours inside its strip -- measured 133 against 154 us per
	// step at base 200, r3), 2 = persistent on the six-round variants, 1 = some strip kernel can run it, 0 = rejected (no strip
	// kernel takes this partition: colour batches)
	auto outcome = [&]() {
		if (s->persistValid && !s->persist.wideRounds)
		{
			return (s->persist.maxRoundsA <= 5 && s->persist.maxSeamRounds <= 2) ? 3 : 2;
		}
		return (s->persistValid || (s->leanAValid && s->leanBValid)) ? 1 : 0;
	};
	// ... and what this solver is after: TGS_Soft the five-round partition (wide_kernel.hip); SoftStep / PGS_Soft any partition
	// their resident kernel takes without spilling (their default strips are wide: stripBodiesFor); every other family any
	// partition the op interpreter can run (generic_kernel.hip: as many strips as CUs at most)
	auto satisfied = [&]() {
		if (!StructureBuild::isSoftFamily(solverType))
		{
			return s->dStripA.view.groupCount > 0 && (s->genericValid || outcome() >= 1);
		}
		if (s->joints.stripCount > 0)
		{
			return s->dStripA.view.groupCount > 0 && s->genericValid;
		}
		return outcome() >= (solverType == s2amd_solverTGS_Soft ? 3 : 2);
	};
	const bool triedStrips = s->stripsRejected || s->dStripA.view.groupCount > 0;
	s->stripRetryPending = false;
	// (a worker's build for overflow contacts -- SolverRest::forcedBuild: anything the resident kernel runs and the placement can work on)
	auto usable = [&]() { return s->dStripA.view.groupCount > 0 && s->persistValid && s->stripInc.valid; };
	if (s->forcedBuild && (usable() || !triedStrips || s->stripsHopeless))
	{
		return rc;
	}
	if (!s->forcedBuild && (!triedStrips || s->stripsHopeless || satisfied() || (s->stripScaleFound > 0.0f && outcome() >= 2)))
	{
		return rc;
	}
	// seven more partitions cost seven more builds (tens of milliseconds at 60k constraints): only for a graph that has been
	// quiet for a while (or when the caller asked for strips at once, strip_patience 0); doStep comes back for it
	if (!s->forcedBuild && s->optStripPatience != 0 && s->graphAge < 32)
	{
		s->stripRetryPending = true;
		return rc;
	}
	float bestScale = firstScale;
	int best = outcome();
	// (the default strip is as thin as the level structure allows -- strip_bodies 8 --: the other candidates are wider, for
	// graphs with more level pairs than the GPU has CUs)
	const float scales[] = {4.0f, 10.0f, 20.0f, 2.0f, 40.0f, 80.0f, 0.5f};
	for (float scale : scales)
	{
		if (s->cancelBuild && static_cast<const std::atomic<int>*>(s->cancelBuild)->load(std::memory_order_relaxed) != 0)
		{
			return rc; // (a worker's copy whose result will be thrown away: solver_async.cpp)
		}
		s->structureDirty = true;
		s->stripsRejected = false; // a width that fits no strip kernel says nothing about the next one
		if ((rc = buildStructureWith(s, solverType, scale)) != S2AMD_OK)
		{
			return rc;
		}
		const int o = outcome();
		if (s->forcedBuild ? usable() : satisfied())
		{
			s->stripScaleFound = scale;
			return rc;
		}
		if (o > best)
		{
			best = o, bestScale = scale;
		}
	}
	s->structureDirty = true;
	s->stripsRejected = false;
	s->stripScaleFound = best >= 2 ? bestScale : 0.0f;
	return buildStructureWith(s, solverType, bestScale);
}
