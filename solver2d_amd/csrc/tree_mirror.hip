// The reference's broad-phase trees on the device ("sort emitted pairs into the reference's creation order").
//
// s2UpdateBroadPhasePairs creates contacts in the order its tree queries call back (src/broad_phase.c:253-254, :288-320, :332-357),
// so a contact's pool slot is a function of the TOPOLOGY of the reference's three s2DynamicTrees (src/broad_phase.h:27) at the moment
// of the query -- and that topology is history: stage 2 of every step rebuilds only the part of a tree that stage 4 of the step before
// flagged (s2DynamicTree_Rebuild(tree, false), src/dynamic_tree.c:1764-1874), the rest is kept as it was built from the boxes of
// earlier steps.  There is no key of the current boxes that gives the order; the trees have to be maintained.  Rounds 3-5 did that on
// the host, replaying the reference's own functions (shim/s2_amd_binding.c: replayTrees, s2amdBinding_OrderPairs: 1.5 ms per step at
// base 200 whenever a pair is created).  This file keeps the node arrays in HBM instead and maintains them there:
//
//   treeEnlargeKernel   stage 4's s2DynamicTree_EnlargeProxy (src/dynamic_tree.c:803-839, called from src/world.c:283-290) for every shape
//                       the refit re-inflated: leaf box replaced, ancestors flagged; flags are only set, so any order gives the
//                       reference's result.  The ancestors' boxes only grow (min / max), equally order-free; they are made when the
//                       tree is read back (treeBoxesClimbKernel), the next rebuild frees those nodes anyway.
//   treeGather*Kernel   stage 2's rebuild, first half (:1800-1860): every flagged node takes its place in the order the reference frees
//                       them in (child1-first pre-order) and every gathered leaf -- a proxy, or an un-flagged subtree kept whole -- its
//                       depth-first position: one walk up each over the leaf counts the tree keeps, two prefix sums.
//   treeBuildTasksKernel  s2BuildTree's recursive median split (:1610-1761, s2PartitionMid :1317-1427) as TASKS, one open segment of the
//                       leaf array each, taken from a queue by 1024-thread workgroups: the Hoare loop of a segment is a fixed
//                       permutation that a prefix sum of the predicate gives (the j-th misplaced element from the left changes places with
//                       the j-th from the right); a segment longer than a workgroup is split once, in global memory, and its parts
//                       queued; a shorter one is finished in LDS down to its leaves, boxes / heights / category bits included.  Node IDS
//                       are the reference's too (the k-th new node in pre-order takes the (M-1-k)-th freed one, as the free list would
//                       hand them out, :105-139), so the arrays can be copied back into the host's s2DynamicTree and proxies created
//                       later get the ids the reference gives them.  No workgroup waits for another one's progress except for a task
//                       to appear, and a producer never waits: no co-residency is needed.
//   pairCreationKeysKernel / pairCreationScatterKernel
//                       the new pairs of a query sorted on (position of the querying proxy in the move buffer, tree, reversed traversal
//                       rank of the other proxy): the sequence s2CreateContact is called in.
//
// tests/tree_parallel.py states the same algorithm in numpy and is pinned node for node against the compiled reference
// (tests/test_tree_rebuild.py, CPU); tests/test_gpu_trees.py pins these kernels against the reference's own tree functions.
#include "solver_internal.h"

#define S2_BLOCK 256
#define TREE_THREADS 1024
#define TREE_NULL (-1)
#define TREE_SMALL TREE_THREADS // a segment this short is finished by one task
#define TREE_BATCH 8				 // loads a lane has in flight in the passes over a long segment
#define TREE_TASK_INTS 5		 // start, end, pre-order index of its node, the parent's node id, which child of it
#define TREE_POLL_LIMIT (1u << 22) // polls of a workgroup waiting for its task before the build is declared starved (about a second)

static_assert(sizeof(s2amdTreeNode) == 48, "s2TreeNode is 48 bytes (include/solver2d/dynamic_tree.h:14-41)");

namespace
{
S2_DEV bool treeFlagged(const TreeView& t, int n)
{
	return t.nodes[n].height > 0 && t.flag[n] != 0;
}

S2_DEV unsigned int sortable(float f)
{
	const unsigned int u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
S2_DEV float unsortable(unsigned int k)
{
	return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// a node as s2AllocateNode hands it out (s2_defaultTreeNode, src/dynamic_tree.c:19), hung under `parent`
S2_DEV void treeInitNode(TreeView& t, int id, int parent)
{
	s2amdTreeNode& n = t.nodes[id];
	n.aabb[0] = 0.0f, n.aabb[1] = 0.0f, n.aabb[2] = 0.0f, n.aabb[3] = 0.0f;
	n.categoryBits = 0u;
	n.parent = parent;
	n.child1 = TREE_NULL, n.child2 = TREE_NULL;
	n.userData = -1;
	n.height = -2;
	n.enlarged = 0;
	t.flag[id] = 0;
	t.arrive[id] = 0;
}

// box, height, category bits (src/dynamic_tree.c:1655-1657, :1742-1744) and leaf count of a new node from its finished children
S2_DEV void treeFinishNode(TreeView& t, int n)
{
	s2amdTreeNode& nn = t.nodes[n];
	const int a = nn.child1, b = nn.child2;
	const s2amdTreeNode& na = t.nodes[a];
	const s2amdTreeNode& nb = t.nodes[b];
	// s2AABB_Union, include/solver2d/aabb.h:42-50 (s2MinFloat / s2MaxFloat: a < b ? a : b)
	nn.aabb[0] = na.aabb[0] < nb.aabb[0] ? na.aabb[0] : nb.aabb[0];
	nn.aabb[1] = na.aabb[1] < nb.aabb[1] ? na.aabb[1] : nb.aabb[1];
	nn.aabb[2] = na.aabb[2] > nb.aabb[2] ? na.aabb[2] : nb.aabb[2];
	nn.aabb[3] = na.aabb[3] > nb.aabb[3] ? na.aabb[3] : nb.aabb[3];
	nn.height = (int16_t)(1 + (na.height > nb.height ? na.height : nb.height));
	nn.categoryBits = na.categoryBits | nb.categoryBits;
	t.leaves[n] = t.leaves[a] + t.leaves[b];
}

// `n` (a gathered leaf, or a new node that is finished) reports to its parent; whoever reports second finishes the parent and goes on.
// The new root's second report ends the rebuild: the tree's root and "nothing flagged" are published.
S2_DEV void treeReport(TreeView& t, int n, int newRoot)
{
	for (;;)
	{
		const int p = t.nodes[n].parent;
		__threadfence();
		if (atomicAdd(t.arrive + p, 1) == 0)
		{
			return;
		}
		__threadfence();
		treeFinishNode(t, p);
		if (p == newRoot)
		{
			__threadfence();
			t.state[0] = newRoot;
			t.state[1] = 0;
			return;
		}
		n = p;
	}
}

} // namespace

// ---- stage 4: the shapes the refit re-inflated enlarge their proxies ----
S2_DEV void treeEnlargeOne(const s2amdShape& sh, TreeViews* views)
{
	const int type = sh.proxyKey & 0xF; // S2_PROXY_TYPE, src/broad_phase.h:18
	if (type != 1 && type != 2)
	{
		return; // (a static shape "in the move buffer" since its creation: its tree was not touched, src/world.c:261-265)
	}
	TreeView& t = views->t[type];
	const int leaf = sh.proxyKey >> 4;
	if (leaf < 0 || leaf >= t.capacity)
	{
		atomicExch(t.state + 2, 1);
		return;
	}
	t.nodes[leaf].aabb[0] = sh.fatAABB[0], t.nodes[leaf].aabb[1] = sh.fatAABB[1], t.nodes[leaf].aabb[2] = sh.fatAABB[2], t.nodes[leaf].aabb[3] = sh.fatAABB[3];
	// Every ancestor is flagged; a node somebody has flagged already has its ancestors flagged, or will have: the walk ends there.
	// The ancestors' BOXES (the reference grows them on the way up, :818-829) are of no consequence to the next rebuild, which frees
	// every flagged node: they are made when somebody asks for the tree (s2amd_world_get_tree: treeBoxesClimbKernel).
	// Nor is a list of the flagged nodes kept (an append per node, 20,000 of them to one counter when every proxy of the base-200
	// pyramid moves, took 120 us): the rebuild looks at every node's flag.
	int p = t.nodes[leaf].parent;
	int guard = 0;
	while (p != TREE_NULL && guard++ < 4096)
	{
		if (__atomic_load_n(t.flag + p, __ATOMIC_RELAXED) != 0 || atomicExch(t.flag + p, 1) != 0)
		{
			break;
		}
		t.nodes[p].enlarged = 1;
		p = t.nodes[p].parent;
	}
}

__global__ __launch_bounds__(S2_BLOCK) void treeEnlargeKernel(const s2amdShape* shapes, int ns, TreeViews* views, const unsigned int* stepFailed)
{
	if (stepFailed != nullptr && *stepFailed != 0u)
	{
		return; // (the step will be repeated: stage 4 stood down, the flags are the previous step's)
	}
	const int si = blockIdx.x * blockDim.x + threadIdx.x;
	if (si >= ns)
	{
		return;
	}
	const s2amdShape& sh = shapes[si];
	if (sh.type == S2AMD_SHAPE_FREE || sh.enlarged == 0)
	{
		return;
	}
	treeEnlargeOne(sh, views);
}

// ... from the list of re-inflated shapes the step's own read-back has just made (world.hip: stepBackKernel; {count, -, -, -, entries}):
// a few hundred entries on a settled world instead of a look at every shape record (17 us at 27,000 shapes)
__global__ __launch_bounds__(S2_BLOCK) void treeEnlargeListKernel(const s2amdShape* shapes, int ns, const int32_t* list, TreeViews* views, const unsigned int* stepFailed)
{
	if (stepFailed != nullptr && *stepFailed != 0u)
	{
		return;
	}
	const int count = list[0] < ns ? list[0] : ns;
	const s2amdMovedBox* moved = (const s2amdMovedBox*)(list + 4);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
	{
		const int si = moved[i].shape;
		if (si >= 0 && si < ns)
		{
			treeEnlargeOne(shapes[si], views);
		}
	}
}

// ---- stage 2: s2DynamicTree_Rebuild(tree, false) ----
// A + B.  The order the flagged nodes are freed in (child1-first pre-order, :1838-1853) and the depth-first order of the gathered leaves,
// without counting anything bottom-up (a count per flagged node climbing to the root is a chain of device-scope atomics: 220 us at base
// 200).  With first(n) = the proxies left of n's subtree -- one walk up over the leaf counts the tree keeps, reads only --, gathered
// leaves in depth-first order are gathered leaves by `first` (their subtrees are disjoint: distinct values), and flagged nodes in
// pre-order are flagged nodes by `first`, ancestors before descendants: those that share a `first` lie on one leftmost path, and a
// node's place among them is the number of consecutive child1 steps above it.  Two histograms over `first`, one prefix sum each.
__global__ __launch_bounds__(S2_BLOCK) void treeGatherInitKernel(TreeViews* views, int which)
{
	TreeView& t = views->t[which];
	const int C = t.capacity;
	const int root = C > 0 ? t.state[0] : TREE_NULL;
	const int gid = blockIdx.x * blockDim.x + threadIdx.x, gn = gridDim.x * blockDim.x;
	if (gid == 0)
	{
		// (a flagged node's parent is flagged -- what s2amd_world_set_tree checked and the enlarge pass keeps --, so nothing is flagged
		// unless the root is)
		const bool go = root != TREE_NULL && treeFlagged(t, root);
		if (t.state[1] != 0)
		{
			atomicExch(t.state + 2, 4); // the last rebuild never reached its root
		}
		t.qstate[0] = 0;		  // next task to take
		t.qstate[1] = 1;		  // tasks queued
		t.qstate[2] = 0;		  // gathered leaves not yet hung under a new node (treeGatherScanKernel knows how many)
		t.qstate[3] = go ? 0 : 1; // done (nothing to do counts as done)
	}
	for (int n = gid; n <= C; n += gn)
	{
		t.ready[n] = 0;
		t.arrive[n] = 0;  // flagged nodes per `first` (the build uses the array per node afterwards, zeroed as it makes each node)
		t.partner[n] = 0; // gathered leaves per `first`
	}
}

__global__ __launch_bounds__(S2_BLOCK) void treeGatherWalkKernel(TreeViews* views, int which)
{
	TreeView& t = views->t[which];
	if (t.capacity <= 0 || t.qstate[3] != 0)
	{
		return;
	}
	const int root = t.state[0];
	const s2amdTreeNode* nodes = t.nodes;
	for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < t.capacity; n += gridDim.x * blockDim.x)
	{
		if (!treeFlagged(t, n))
		{
			continue;
		}
		int first = 0, chain = 0;
		bool leftmost = true;
		for (int x = n; x != root;)
		{
			const int p = nodes[x].parent;
			if (nodes[p].child2 == x)
			{
				first += t.leaves[nodes[p].child1];
				leftmost = false;
			}
			else if (leftmost)
			{
				chain += 1;
			}
			x = p;
		}
		t.acc[n] = first;
		t.pending[n] = chain;
		atomicAdd(t.arrive + first, 1);
		const int c1 = nodes[n].child1, c2 = nodes[n].child2;
		if (!treeFlagged(t, c1))
		{
			t.partner[first] = 1;
		}
		if (!treeFlagged(t, c2))
		{
			t.partner[first + t.leaves[c1]] = 1;
		}
	}
}

// exclusive prefix sums of the two histograms, in place; the counts (M flagged nodes, M + 1 gathered leaves) start the build's queue.
// Every wave owns a run of 64-entry tiles and loads them all before it adds anything (a load per tile in turn is a memory round trip
// per tile: 51 us for 2 x 20,000 entries).
#define TREE_SCAN_TILES 24 // tiles a wave holds in registers per batch
__global__ __launch_bounds__(TREE_THREADS) void treeGatherScanKernel(TreeViews* views, int which)
{
	__shared__ unsigned long long sWave[TREE_THREADS / 64];
	TreeView& t = views->t[which];
	if (t.capacity <= 0 || t.qstate[3] != 0)
	{
		return;
	}
	const int N = t.leaves[t.state[0]] + 1; // `first` runs over [0, proxies)
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int tiles = (N + 63) / 64, perWave = (tiles + TREE_THREADS / 64 - 1) / (TREE_THREADS / 64);
	const int tile0 = min(wave * perWave, tiles), tile1 = min(tile0 + perWave, tiles);
	// both histograms at once: leaves in the high word, flagged nodes in the low one (neither sum reaches 2^31)
	unsigned long long v[TREE_SCAN_TILES];
	unsigned long long sum = 0ull;
	for (int b0 = tile0; b0 < tile1; b0 += TREE_SCAN_TILES)
	{
#pragma unroll
		for (int k = 0; k < TREE_SCAN_TILES; ++k)
		{
			const int i = (b0 + k) * 64 + lane;
			const bool in = b0 + k < tile1 && i < N;
			v[k] = in ? (((unsigned long long)(unsigned int)t.partner[i] << 32) | (unsigned long long)(unsigned int)t.arrive[i]) : 0ull;
		}
#pragma unroll
		for (int k = 0; k < TREE_SCAN_TILES; ++k)
		{
			sum += v[k];
		}
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
	{
		sum += ((unsigned long long)(unsigned int)__shfl_xor((int)(sum >> 32), d, 64) << 32) | (unsigned long long)(unsigned int)__shfl_xor((int)(sum & 0xffffffffull), d, 64);
	}
	if (lane == 0)
	{
		sWave[wave] = sum;
	}
	__syncthreads();
	unsigned long long run = 0ull, total = 0ull;
	for (int w = 0; w < TREE_THREADS / 64; ++w)
	{
		run += w < wave ? sWave[w] : 0ull;
		total += sWave[w];
	}
	const bool kept = tile1 - tile0 <= TREE_SCAN_TILES; // (the wave's whole run is still in its registers)
	for (int b0 = tile0; b0 < tile1; b0 += TREE_SCAN_TILES)
	{
		if (!kept)
		{
#pragma unroll
			for (int k = 0; k < TREE_SCAN_TILES; ++k)
			{
				const int i = (b0 + k) * 64 + lane;
				const bool in = b0 + k < tile1 && i < N;
				v[k] = in ? (((unsigned long long)(unsigned int)t.partner[i] << 32) | (unsigned long long)(unsigned int)t.arrive[i]) : 0ull;
			}
		}
#pragma unroll
		for (int k = 0; k < TREE_SCAN_TILES; ++k)
		{
			unsigned long long x = v[k];
#pragma unroll
			for (int d = 1; d < 64; d <<= 1)
			{
				const unsigned long long y = ((unsigned long long)(unsigned int)__shfl_up((int)(x >> 32), d, 64) << 32) |
											 (unsigned long long)(unsigned int)__shfl_up((int)(x & 0xffffffffull), d, 64);
				if (lane >= d)
				{
					x += y;
				}
			}
			const int i = (b0 + k) * 64 + lane;
			if (b0 + k < tile1 && i < N)
			{
				const unsigned long long before = run + x - v[k];
				t.arrive[i] = (int)(before & 0xffffffffull);
				t.partner[i] = (int)(before >> 32);
			}
			run += ((unsigned long long)(unsigned int)__shfl((int)(x >> 32), 63, 64) << 32) | (unsigned long long)(unsigned int)__shfl((int)(x & 0xffffffffull), 63, 64);
		}
	}
	if (tid == 0)
	{
		const int M = (int)(total & 0xffffffffull), leavesFound = (int)(total >> 32);
		if (leavesFound != M + 1)
		{
			atomicExch(t.state + 2, 3); // a flagged node whose parent is not flagged
			t.qstate[3] = 1;
			return;
		}
		t.state[1] = M;
		t.qstate[2] = M + 1;
		t.tasks[0] = 0, t.tasks[1] = M + 1, t.tasks[2] = 0, t.tasks[3] = TREE_NULL, t.tasks[4] = 0;
		__threadfence();
		t.ready[0] = 1;
	}
}

__global__ __launch_bounds__(S2_BLOCK) void treeGatherPlaceKernel(TreeViews* views, int which)
{
	TreeView& t = views->t[which];
	if (t.capacity <= 0 || t.qstate[3] != 0)
	{
		return;
	}
	const s2amdTreeNode* nodes = t.nodes;
	for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < t.capacity; n += gridDim.x * blockDim.x)
	{
		if (!treeFlagged(t, n))
		{
			continue;
		}
		const int first = t.acc[n];
		t.oldPre[t.arrive[first] + t.pending[n]] = n;
		for (int side = 0; side < 2; ++side)
		{
			const int c = side == 0 ? nodes[n].child1 : nodes[n].child2;
			if (treeFlagged(t, c))
			{
				continue;
			}
			const int li = t.partner[first + (side == 0 ? 0 : t.leaves[nodes[n].child1])];
			t.leafIdx[li] = c;
			t.cx[li] = 0.5f * (nodes[c].aabb[0] + nodes[c].aabb[2]); // s2AABB_Center, include/solver2d/aabb.h:28-32
			t.cy[li] = 0.5f * (nodes[c].aabb[1] + nodes[c].aabb[3]);
		}
	}
}

// Reading a tree back between a refit and the next rebuild (s2amd_world_get_tree): every flagged node's box grown to hold its children's,
// bottom-up -- what the reference's enlarge walks leave in them (src/dynamic_tree.c:818-829: a flagged node's box = its box at the last
// rebuild united with the new boxes of the proxies below).
__global__ __launch_bounds__(S2_BLOCK) void treeBoxesInitKernel(TreeViews* views, int which)
{
	TreeView& t = views->t[which];
	for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < t.capacity; n += gridDim.x * blockDim.x)
	{
		if (treeFlagged(t, n))
		{
			t.pending[n] = (treeFlagged(t, t.nodes[n].child1) ? 1 : 0) + (treeFlagged(t, t.nodes[n].child2) ? 1 : 0);
		}
	}
}

__global__ __launch_bounds__(S2_BLOCK) void treeBoxesClimbKernel(TreeViews* views, int which)
{
	TreeView& t = views->t[which];
	if (t.capacity <= 0)
	{
		return;
	}
	const int root = t.state[0];
	s2amdTreeNode* nodes = t.nodes;
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < t.capacity; k += gridDim.x * blockDim.x)
	{
		int n = k;
		if (!treeFlagged(t, n) || treeFlagged(t, nodes[n].child1) || treeFlagged(t, nodes[n].child2))
		{
			continue;
		}
		for (;;)
		{
			s2amdTreeNode& nn = nodes[n];
			for (int side = 0; side < 2; ++side)
			{
				const s2amdTreeNode& c = nodes[side == 0 ? nn.child1 : nn.child2];
				// s2AABB_Enlarge, include/solver2d/aabb.h:62-90
				nn.aabb[0] = c.aabb[0] < nn.aabb[0] ? c.aabb[0] : nn.aabb[0];
				nn.aabb[1] = c.aabb[1] < nn.aabb[1] ? c.aabb[1] : nn.aabb[1];
				nn.aabb[2] = nn.aabb[2] < c.aabb[2] ? c.aabb[2] : nn.aabb[2];
				nn.aabb[3] = nn.aabb[3] < c.aabb[3] ? c.aabb[3] : nn.aabb[3];
			}
			if (n == root)
			{
				break;
			}
			const int p = nn.parent;
			__threadfence();
			if (atomicSub(t.pending + p, 1) != 1)
			{
				break;
			}
			__threadfence();
			n = p;
		}
	}
}

// C + D. the median splits as tasks; a task's node is finished (box, height, category bits, leaf count) when both its children are
__global__ __launch_bounds__(TREE_THREADS) void treeBuildTasksKernel(TreeViews* views, int which)
{
	__shared__ int sTask[TREE_TASK_INTS + 1];
	__shared__ int lds[2 * (TREE_THREADS / 64) + 2];
	__shared__ unsigned int sRed[4 * (TREE_THREADS / 64)];
	__shared__ int sLeaf[TREE_SMALL];
	__shared__ float sCx[TREE_SMALL], sCy[TREE_SMALL];
	__shared__ int sSeg[TREE_SMALL], sScan[TREE_SMALL + 1], sPartner[TREE_SMALL];
	__shared__ int gStart[TREE_SMALL], gEnd[TREE_SMALL], gSplit[TREE_SMALL];
	__shared__ int cLeft[TREE_SMALL], cRight[TREE_SMALL]; // a local node's children: a local node index, or ~(node id) of a gathered leaf
	__shared__ unsigned int gBounds[4 * TREE_SMALL];
	TreeView& t = views->t[which];
	if (t.capacity <= 0 || t.state[2] != 0)
	{
		return;
	}
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int M = t.state[1];
	if (M <= 0)
	{
		return;
	}
	s2amdTreeNode* nodes = t.nodes;
	auto nodeOf = [&](int pre) { return t.oldPre[M - 1 - pre]; };
	const int newRoot = nodeOf(0);
	for (;;)
	{
		// take the next task (one thread polls; a task that never comes means the build is done)
		if (tid == 0)
		{
			int got = 0;
			if (__atomic_load_n(t.qstate + 3, __ATOMIC_RELAXED) == 0)
			{
				const int slot = atomicAdd(t.qstate + 0, 1);
				if (slot < M)
				{
					// (bounded: a queue that starves -- a count that does not add up -- ends the build with error 5 instead of holding
					// the device; a second or so of polls, five thousand times the longest rebuild measured)
					for (unsigned int polls = 0;; ++polls)
					{
						if (__atomic_load_n(t.ready + slot, __ATOMIC_RELAXED) != 0)
						{
							got = 1;
							break;
						}
						if (__atomic_load_n(t.qstate + 3, __ATOMIC_RELAXED) != 0)
						{
							break;
						}
						if (polls >= TREE_POLL_LIMIT)
						{
							atomicExch(t.state + 2, 5);
							atomicExch(t.qstate + 3, 1);
							break;
						}
						__builtin_amdgcn_s_sleep(8);
					}
				}
				if (got)
				{
					__threadfence();
					for (int k = 0; k < TREE_TASK_INTS; ++k)
					{
						sTask[k] = __atomic_load_n(t.tasks + TREE_TASK_INTS * (size_t)slot + k, __ATOMIC_RELAXED);
					}
				}
			}
			sTask[TREE_TASK_INTS] = got;
		}
		__syncthreads();
		if (sTask[TREE_TASK_INTS] == 0)
		{
			return;
		}
		__threadfence(); // (acquire: what the producer of this task wrote -- the exchanged leaf array -- is read below)
		const int start = sTask[0], end = sTask[1], pre = sTask[2], parent = sTask[3], side = sTask[4];
		const int n = end - start;
		const int me = nodeOf(pre);
		if (tid == 0)
		{
			treeInitNode(t, me, parent);
			if (parent != TREE_NULL)
			{
				(side == 0 ? nodes[parent].child1 : nodes[parent].child2) = me;
			}
		}
		int hung = 0; // gathered leaves this task hangs under a node
		if (n > TREE_SMALL)
		{
			// ---- one split of a long segment, in global memory (s2PartitionMid, src/dynamic_tree.c:1317-1427) ----
			// Loads are issued TREE_BATCH at a time (a tile loaded when it is needed is a memory round trip per tile and pass: 100 us
			// for the top segment of the base-200 pyramid).
			unsigned int lx = 0xffffffffu, ly = 0xffffffffu, ux = 0u, uy = 0u;
			for (int i0 = start + tid; i0 < end; i0 += TREE_BATCH * TREE_THREADS)
			{
				float vx[TREE_BATCH], vy[TREE_BATCH];
#pragma unroll
				for (int k = 0; k < TREE_BATCH; ++k)
				{
					const int i = i0 + k * TREE_THREADS;
					vx[k] = i < end ? t.cx[i] : 0.0f, vy[k] = i < end ? t.cy[i] : 0.0f;
				}
#pragma unroll
				for (int k = 0; k < TREE_BATCH; ++k)
				{
					if (i0 + k * TREE_THREADS < end)
					{
						const unsigned int x = sortable(vx[k]), y = sortable(vy[k]);
						lx = min(lx, x), ly = min(ly, y), ux = max(ux, x), uy = max(uy, y);
					}
				}
			}
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1)
			{
				lx = min(lx, (unsigned int)__shfl_xor((int)lx, d, 64));
				ly = min(ly, (unsigned int)__shfl_xor((int)ly, d, 64));
				ux = max(ux, (unsigned int)__shfl_xor((int)ux, d, 64));
				uy = max(uy, (unsigned int)__shfl_xor((int)uy, d, 64));
			}
			if (lane == 0)
			{
				sRed[4 * wave + 0] = lx, sRed[4 * wave + 1] = ly, sRed[4 * wave + 2] = ux, sRed[4 * wave + 3] = uy;
			}
			__syncthreads();
			for (int w = 0; w < TREE_THREADS / 64; ++w)
			{
				lx = min(lx, sRed[4 * w + 0]), ly = min(ly, sRed[4 * w + 1]), ux = max(ux, sRed[4 * w + 2]), uy = max(uy, sRed[4 * w + 3]);
			}
			const float flx = unsortable(lx), fly = unsortable(ly), fux = unsortable(ux), fuy = unsortable(uy);
			const bool byX = (fux - flx) > (fuy - fly);
			const float pivot = byX ? 0.5f * (flx + fux) : 0.5f * (fly + fuy);
			const float* centre = byX ? t.cx : t.cy;
			// every wave owns a run of 64-element tiles: the predicate's prefix inside a tile is a ballot
			const int tiles = (n + 63) / 64, perWave = (tiles + TREE_THREADS / 64 - 1) / (TREE_THREADS / 64);
			const int tile0 = min(wave * perWave, tiles), tile1 = min(tile0 + perWave, tiles);
			int mine = 0;
#pragma unroll 1
			for (int b0 = tile0; b0 < tile1; b0 += TREE_BATCH)
			{
				float v[TREE_BATCH];
#pragma unroll
				for (int k = 0; k < TREE_BATCH; ++k)
				{
					const int i = start + (b0 + k) * 64 + lane;
					v[k] = (b0 + k < tile1 && i < end) ? centre[i] : pivot;
				}
#pragma unroll
				for (int k = 0; k < TREE_BATCH; ++k)
				{
					mine += __popcll(__ballot(v[k] < pivot));
				}
			}
			__syncthreads();
			if (lane == 0)
			{
				lds[wave] = mine;
			}
			__syncthreads();
			int base = 0, m = 0;
			for (int w = 0; w < TREE_THREADS / 64; ++w)
			{
				base += w < wave ? lds[w] : 0;
				m += lds[w];
			}
			int split = n / 2; // (:1422-1429: nothing on one side of the pivot)
			if (m > 0 && m < n)
			{
				// the Hoare loop's exchanges (:1357-1420): the j-th misplaced element from the left (position into `acc`) with the j-th
				// from the right (into `partner`); there are as many of the one as of the other
				split = m;
				int running = base, misplaced = 0;
#pragma unroll 1
				for (int b0 = tile0; b0 < tile1; b0 += TREE_BATCH)
				{
					float v[TREE_BATCH];
#pragma unroll
					for (int k = 0; k < TREE_BATCH; ++k)
					{
						const int i = start + (b0 + k) * 64 + lane;
						v[k] = (b0 + k < tile1 && i < end) ? centre[i] : pivot;
					}
#pragma unroll
					for (int k = 0; k < TREE_BATCH; ++k)
					{
						const int i = start + (b0 + k) * 64 + lane;
						const bool in = b0 + k < tile1 && i < end;
						const bool left = v[k] < pivot;
						const unsigned long long mask = __ballot(left);
						const int before = running + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
						if (in && left && i - start >= m)
						{
							t.partner[start + (m - before - 1)] = i;
						}
						else if (in && !left && i - start < m)
						{
							t.acc[start + (i - start - before)] = i;
						}
						misplaced += __popcll(__ballot(in && !left && i - start < m));
						running += __popcll(mask);
					}
				}
				__syncthreads();
				if (lane == 0)
				{
					lds[wave] = misplaced;
				}
				__threadfence_block();
				__syncthreads();
				int pairs = 0;
				for (int w = 0; w < TREE_THREADS / 64; ++w)
				{
					pairs += lds[w];
				}
				for (int j0 = tid; j0 < pairs; j0 += TREE_BATCH * TREE_THREADS)
				{
					int pi[TREE_BATCH], pp[TREE_BATCH];
#pragma unroll
					for (int k = 0; k < TREE_BATCH; ++k)
					{
						const int j = j0 + k * TREE_THREADS;
						pi[k] = j < pairs ? t.acc[start + j] : -1, pp[k] = j < pairs ? t.partner[start + j] : -1;
					}
#pragma unroll
					for (int k = 0; k < TREE_BATCH; ++k)
					{
						if (pi[k] >= 0)
						{
							const int i = pi[k], p = pp[k];
							const int li = t.leafIdx[i], lp = t.leafIdx[p];
							const float ox = t.cx[i], oy = t.cy[i], px = t.cx[p], py = t.cy[p];
							t.leafIdx[i] = lp, t.cx[i] = px, t.cy[i] = py;
							t.leafIdx[p] = li, t.cx[p] = ox, t.cy[p] = oy;
						}
					}
				}
			}
			__threadfence();
			__syncthreads();
			if (tid == 0)
			{
				for (int which2 = 0; which2 < 2; ++which2)
				{
					const int ps = which2 == 0 ? start : start + split, pe = which2 == 0 ? start + split : end;
					if (pe - ps == 1)
					{
						const int c = t.leafIdx[ps];
						nodes[c].parent = me;
						(which2 == 0 ? nodes[me].child1 : nodes[me].child2) = c;
						hung += 1;
						treeReport(t, c, newRoot);
					}
					else
					{
						const int slot = atomicAdd(t.qstate + 1, 1);
						int* q = t.tasks + TREE_TASK_INTS * (size_t)slot;
						q[0] = ps, q[1] = pe, q[2] = which2 == 0 ? pre + 1 : pre + split, q[3] = me, q[4] = which2;
						__threadfence();
						atomicExch(t.ready + slot, 1);
					}
				}
			}
		}
		else
		{
			// ---- a short segment down to its leaves, in LDS, with the segment's own numbering (element i - start, node pre + local
			// index): every level of all its open sub-segments at once ----
			const bool has = tid < n;
			// (every local segment's bounds start empty: a segment is named by its local pre-order index, used once)
			for (int i = tid; i < 4 * TREE_SMALL; i += TREE_THREADS)
			{
				gBounds[i] = (i & 2) ? 0u : 0xffffffffu;
			}
			if (has)
			{
				sLeaf[tid] = t.leafIdx[start + tid], sCx[tid] = t.cx[start + tid], sCy[tid] = t.cy[start + tid];
				sSeg[tid] = 0;
			}
			if (tid == 0)
			{
				gStart[0] = 0, gEnd[0] = n;
			}
			__syncthreads();
			if (has && n > 2)
			{
				atomicMin(gBounds + 0, sortable(sCx[tid])), atomicMin(gBounds + 1, sortable(sCy[tid]));
				atomicMax(gBounds + 2, sortable(sCx[tid])), atomicMax(gBounds + 3, sortable(sCy[tid]));
			}
			__syncthreads();
			for (int level = 0; level <= n; ++level)
			{
				const int s = has ? sSeg[tid] : -1;
				const int s0 = s >= 0 ? gStart[s] : 0, s1 = s >= 0 ? gEnd[s] : 0;
				const bool wide = s >= 0 && s1 - s0 > 2; // (:1320-1323: two elements or fewer are split in the middle, untouched)
				bool left = false;
				if (wide)
				{
					const float flx = unsortable(gBounds[4 * s + 0]), fly = unsortable(gBounds[4 * s + 1]);
					const float fux = unsortable(gBounds[4 * s + 2]), fuy = unsortable(gBounds[4 * s + 3]);
					left = (fux - flx) > (fuy - fly) ? sCx[tid] < 0.5f * (flx + fux) : sCy[tid] < 0.5f * (fly + fuy);
				}
				// prefix sum of the predicate over the task's elements: one barrier (every thread adds the waves before its own)
				int x = left ? 1 : 0;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1)
				{
					const int y = __shfl_up(x, d, 64);
					if (lane >= d)
					{
						x += y;
					}
				}
				int* waveSums = lds + (level & 1) * (TREE_THREADS / 64); // (two sets: the next level writes while a slow wave still adds)
				if (lane == 63)
				{
					waveSums[wave] = x;
				}
				__syncthreads();
				int before0 = x - (left ? 1 : 0), total = 0;
				for (int w = 0; w < TREE_THREADS / 64; ++w)
				{
					before0 += w < wave ? waveSums[w] : 0;
					total += waveSums[w];
				}
				sScan[tid] = before0;
				if (tid == 0)
				{
					sScan[TREE_SMALL] = total;
				}
				__syncthreads();
				int split = (s1 - s0) / 2, m = 0;
				if (wide)
				{
					m = sScan[s1] - sScan[s0];
					if (m > 0 && m < s1 - s0)
					{
						split = m;
						if (left && tid - s0 >= m)
						{
							sPartner[s0 + (m - (before0 - sScan[s0]) - 1)] = tid;
						}
					}
				}
				if (s >= 0 && tid == s0)
				{
					gSplit[s] = split;
				}
				__syncthreads();
				if (wide && m > 0 && m < s1 - s0 && !left && tid - s0 < m)
				{
					const int p = sPartner[s0 + (tid - s0 - (before0 - sScan[s0]))];
					const int li = sLeaf[tid];
					const float x2 = sCx[tid], y2 = sCy[tid];
					sLeaf[tid] = sLeaf[p], sCx[tid] = sCx[p], sCy[tid] = sCy[p];
					sLeaf[p] = li, sCx[p] = x2, sCy[p] = y2;
				}
				__syncthreads();
				int open = 0;
				if (s >= 0)
				{
					const int sp = gSplit[s];
					const bool inLeft = tid - s0 < sp;
					const int ps = inLeft ? s0 : s0 + sp, pe = inLeft ? s0 + sp : s1;
					if (pe - ps == 1)
					{
						(inLeft ? cLeft : cRight)[s] = ~sLeaf[tid];
						sSeg[tid] = -1;
					}
					else
					{
						const int cs = inLeft ? s + 1 : s + sp; // (allocated in pre-order: :1622, :1716)
						if (tid == ps)
						{
							(inLeft ? cLeft : cRight)[s] = cs;
							gStart[cs] = ps, gEnd[cs] = pe;
						}
						if (pe - ps > 2)
						{
							// (the part's centres into its bounds now: the next level starts with them)
							atomicMin(gBounds + 4 * cs + 0, sortable(sCx[tid])), atomicMin(gBounds + 4 * cs + 1, sortable(sCy[tid]));
							atomicMax(gBounds + 4 * cs + 2, sortable(sCx[tid])), atomicMax(gBounds + 4 * cs + 3, sortable(sCy[tid]));
						}
						sSeg[tid] = cs;
						open = 1;
					}
				}
				if (__syncthreads_or(open) == 0)
				{
					break;
				}
			}
			// the n - 1 nodes of the subtree, one thread each: links, then boxes / heights / category bits / leaf counts bottom-up
			// (a node is finished in the round after its children; the values travel through the arrays the splits are done with)
			float* fBox = (float*)gBounds;
			int* fHeight = gStart;
			unsigned int* fCat = (unsigned int*)gEnd;
			int* fLeaves = gSplit;
			const bool mineNode = tid < n - 1;
			const int refL = mineNode ? cLeft[tid] : 0, refR = mineNode ? cRight[tid] : 0;
			const int id = mineNode ? nodeOf(pre + tid) : TREE_NULL;
			float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f, q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
			int hL = 0, hR = 0, lvL = 0, lvR = 0;
			unsigned int catL = 0u, catR = 0u;
			__syncthreads();
			if (mineNode)
			{
				fHeight[tid] = -2;
				if (refL < 0)
				{
					const s2amdTreeNode& c = nodes[~refL];
					b0 = c.aabb[0], b1 = c.aabb[1], b2 = c.aabb[2], b3 = c.aabb[3], hL = c.height, catL = c.categoryBits, lvL = t.leaves[~refL];
				}
				if (refR < 0)
				{
					const s2amdTreeNode& c = nodes[~refR];
					q0 = c.aabb[0], q1 = c.aabb[1], q2 = c.aabb[2], q3 = c.aabb[3], hR = c.height, catR = c.categoryBits, lvR = t.leaves[~refR];
				}
			}
			__syncthreads();
			bool finished = !mineNode;
			for (int round = 0; round <= n; ++round)
			{
				const bool now = !finished && (refL < 0 || fHeight[refL] >= 0) && (refR < 0 || fHeight[refR] >= 0);
				__syncthreads();
				if (now)
				{
					if (refL >= 0)
					{
						b0 = fBox[4 * refL + 0], b1 = fBox[4 * refL + 1], b2 = fBox[4 * refL + 2], b3 = fBox[4 * refL + 3];
						hL = fHeight[refL], catL = fCat[refL], lvL = fLeaves[refL];
					}
					if (refR >= 0)
					{
						q0 = fBox[4 * refR + 0], q1 = fBox[4 * refR + 1], q2 = fBox[4 * refR + 2], q3 = fBox[4 * refR + 3];
						hR = fHeight[refR], catR = fCat[refR], lvR = fLeaves[refR];
					}
					// s2AABB_Union, include/solver2d/aabb.h:42-50
					b0 = b0 < q0 ? b0 : q0, b1 = b1 < q1 ? b1 : q1, b2 = b2 > q2 ? b2 : q2, b3 = b3 > q3 ? b3 : q3;
					hL = 1 + (hL > hR ? hL : hR), catL |= catR, lvL += lvR;
					fBox[4 * tid + 0] = b0, fBox[4 * tid + 1] = b1, fBox[4 * tid + 2] = b2, fBox[4 * tid + 3] = b3;
					fCat[tid] = catL, fLeaves[tid] = lvL;
					fHeight[tid] = hL; // (a parent looks at this in the NEXT round's first half, behind the barrier below)
					finished = true;
				}
				if (__syncthreads_and(finished ? 1 : 0) != 0)
				{
					break;
				}
			}
			if (mineNode)
			{
				s2amdTreeNode& nn = nodes[id];
				if (tid != 0)
				{
					treeInitNode(t, id, TREE_NULL); // (the subtree's own root was made when the task was taken)
				}
				nn.aabb[0] = b0, nn.aabb[1] = b1, nn.aabb[2] = b2, nn.aabb[3] = b3;
				nn.height = (int16_t)hL;
				nn.categoryBits = catL;
				t.leaves[id] = lvL;
				nn.child1 = refL < 0 ? ~refL : nodeOf(pre + refL), nn.child2 = refR < 0 ? ~refR : nodeOf(pre + refR);
			}
			__syncthreads(); // (a child's own thread has made the child: now its parent hangs it)
			if (mineNode)
			{
				nodes[nodes[id].child1].parent = id, nodes[nodes[id].child2].parent = id;
			}
			__threadfence();
			__syncthreads();
			hung = tid == 0 ? n : 0;
			if (tid == 0)
			{
				if (me == newRoot)
				{
					t.state[0] = newRoot;
					t.state[1] = 0;
				}
				else
				{
					treeReport(t, me, newRoot);
				}
			}
		}
		if (tid == 0 && hung > 0)
		{
			__threadfence();
			if (atomicSub(t.qstate + 2, hung) == hung)
			{
				atomicExch(t.qstate + 3, 1); // the last leaf hangs: every workgroup waiting for a task leaves
			}
		}
		__syncthreads();
	}
}

// ---- the creation order of a query's new pairs ----
// keys[i] = shapeA << 32 | shapeB of a new pair as the query kernels emitted it (any order); out = the same pairs in the order
// s2UpdateBroadPhasePairs creates them in: move-buffer position of the proxy that asked (src/broad_phase.c:332), then the trees in reverse
// query order and each tree's callbacks in reverse (the pair list is LIFO, :253-254; the trees are queried dynamic, kinematic, static,
// :300-311; s2DynamicTree_Query pops child2 before child1, src/dynamic_tree.c:1171-1210).
__global__ __launch_bounds__(S2_BLOCK) void pairCreationKeysKernel(const s2amdShape* shapes, const unsigned char* moved, const TreeViews* views,
																   const unsigned long long* keys, const unsigned int* count, unsigned int cap,
																   unsigned long long* ckeys)
{
	const unsigned int found = min(count[0], cap);
	for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < found; i += gridDim.x * blockDim.x)
	{
		const int a = (int)(keys[i] >> 32), b = (int)(keys[i] & 0xffffffffull);
		const int keyA = shapes[a].proxyKey, keyB = shapes[b].proxyKey;
		// (a static proxy never asks: it is not in the move buffer, src/broad_phase.c:94-104 -- whatever flag its shape was uploaded with)
		const bool movedA = moved[a] != 0 && (keyA & 0xF) != 0, movedB = moved[b] != 0 && (keyB & 0xF) != 0;
		// who asked (:196-201): the one that moved; when both did, the one with the larger key
		const bool queryIsB = movedB && (!movedA || keyB > keyA);
		const int other = queryIsB ? keyA : keyB;
		const int asking = queryIsB ? b : a;
		const unsigned long long moveIndex = views->refitPos != nullptr ? (unsigned long long)(unsigned int)views->refitPos[asking] : (unsigned long long)(unsigned int)asking;
		const int type = other & 0xF;
		const TreeView& t = views->t[type < 3 ? type : 0];
		int n = other >> 4;
		unsigned int rank = 0;
		if (n >= 0 && n < t.capacity)
		{
			int guard = 0;
			for (int p = t.nodes[n].parent; p != TREE_NULL && guard++ < 4096; p = t.nodes[n].parent)
			{
				if (t.nodes[p].child1 == n)
				{
					rank += (unsigned int)t.leaves[t.nodes[p].child2];
				}
				n = p;
			}
		}
		ckeys[i] = (moveIndex << 34) | ((unsigned long long)(unsigned int)type << 32) | (unsigned long long)(0xffffffffu - rank);
	}
}

// out[#keys below mine] = my pair: the keys are distinct (one querying proxy meets another proxy once)
__global__ __launch_bounds__(S2_BLOCK) void pairCreationScatterKernel(const unsigned long long* keys, const unsigned long long* ckeys, const unsigned int* count,
																	  unsigned int cap, unsigned long long* out)
{
	__shared__ unsigned long long tile[S2_BLOCK];
	const unsigned int found = min(count[0], cap);
	const unsigned int rounds = (found + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
	for (unsigned int r = 0; r < rounds; ++r)
	{
		const unsigned int i = (r * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
		const unsigned long long mine = i < found ? ckeys[i] : ~0ull;
		unsigned int below = 0;
		for (unsigned int base = 0; base < found; base += S2_BLOCK)
		{
			__syncthreads();
			tile[threadIdx.x] = base + threadIdx.x < found ? ckeys[base + threadIdx.x] : ~0ull;
			__syncthreads();
			const unsigned int n = min((unsigned int)S2_BLOCK, found - base);
			for (unsigned int j = 0; j < n; ++j)
			{
				below += tile[j] < mine ? 1u : 0u;
			}
		}
		if (i < found)
		{
			out[below] = keys[i];
		}
	}
}

__global__ __launch_bounds__(S2_BLOCK) void refitPositionsKernel(const int* order, int n, int* pos)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		pos[order[i]] = i;
	}
}

// ---- host side ----
struct DeviceTrees
{
	DevBuf block[3]; // per tree: nodes, flags, leaf counts, flagged list, state, rebuild scratch
	DevBuf dViews, dRefitPos;
	TreeViews h{};
	bool set[3] = {false, false, false};
	int leafCount[3] = {0, 0, 0};
	// the rebuild runs beside stage 3 and the solve, on a stream of its own: it starts when the step starts (behind whatever the step's
	// stream holds by then -- the last pair query's ranking) and the refit's enlarge pass waits for it
	hipStream_t stream = nullptr;
	hipEvent_t evStart = nullptr, evDone = nullptr;
	bool pending = false; // a rebuild is in flight on `stream`
	int refitPosFor = -1; // refit order (count) the positions were made for
};

void treesFree(s2amdSolver* s)
{
	if (s->trees)
	{
		for (DevBuf& b : s->trees->block)
		{
			b.release();
		}
		s->trees->dViews.release();
		s->trees->dRefitPos.release();
		if (s->trees->stream)
		{
			(void)hipStreamSynchronize(s->trees->stream);
			(void)hipStreamDestroy(s->trees->stream);
			(void)hipEventDestroy(s->trees->evStart);
			(void)hipEventDestroy(s->trees->evDone);
		}
		delete s->trees;
		s->trees = nullptr;
	}
}

void treesForget(s2amdSolver* s)
{
	if (s->trees)
	{
		s->trees->set[0] = s->trees->set[1] = s->trees->set[2] = false;
		s->trees->refitPosFor = -1;
	}
}

bool treesActive(const s2amdSolver* s)
{
	return s->trees != nullptr && s->trees->set[0] && s->trees->set[1] && s->trees->set[2];
}

const TreeViews* treesViews(const s2amdSolver* s)
{
	return treesActive(s) ? (const TreeViews*)s->trees->dViews.p : nullptr;
}

static int pushViews(s2amdSolver* s)
{
	DeviceTrees* d = s->trees;
	int rc = d->dViews.ensure(sizeof(TreeViews));
	if (rc)
	{
		return rc;
	}
	HIP_TRY(hipMemcpyAsync(d->dViews.p, &d->h, sizeof(TreeViews), hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	return S2AMD_OK;
}

// the refit order's inverse (the move buffer's order, src/world.c:259-297): made when the order or the trees change
int treesSyncRefitOrder(s2amdSolver* s)
{
	DeviceTrees* d = s->trees;
	if (d == nullptr)
	{
		return S2AMD_OK;
	}
	const int n = s->refitOrderCount;
	if (n <= 0)
	{
		if (d->h.refitPos != nullptr)
		{
			d->h.refitPos = nullptr;
			d->refitPosFor = -1;
			return pushViews(s);
		}
		return S2AMD_OK;
	}
	int rc = d->dRefitPos.ensure((size_t)std::max(s->shapeCapacity, 1) * sizeof(int));
	if (rc)
	{
		return rc;
	}
	HIP_TRY(hipMemsetAsync(d->dRefitPos.p, 0, (size_t)std::max(s->shapeCapacity, 1) * sizeof(int), s->stream));
	refitPositionsKernel<<<dim3((unsigned)((n + S2_BLOCK - 1) / S2_BLOCK)), dim3(S2_BLOCK), 0, s->stream>>>((const int*)s->dRefitOrder.p, n, (int*)d->dRefitPos.p);
	HIP_TRY(hipGetLastError());
	d->h.refitPos = (const int*)d->dRefitPos.p;
	d->refitPosFor = n;
	return pushViews(s);
}

void launchTreeEnlarge(s2amdSolver* s, hipStream_t st, const unsigned int* stepFailed)
{
	if (!treesActive(s) || s->shapeCapacity <= 0)
	{
		return;
	}
	treesJoin(s, st);
	if (s->stepBackListFresh && s->dStepBack.p != nullptr)
	{
		const int blocks = (s->shapeCapacity + S2_BLOCK - 1) / S2_BLOCK; // (everything may have moved; a block without an entry is gone at once)
		treeEnlargeListKernel<<<dim3((unsigned)blocks), dim3(S2_BLOCK), 0, st>>>((const s2amdShape*)s->dShapes.p, s->shapeCapacity, (const int32_t*)s->dStepBack.p,
																				 (TreeViews*)s->trees->dViews.p, stepFailed);
		return;
	}
	treeEnlargeKernel<<<dim3((unsigned)((s->shapeCapacity + S2_BLOCK - 1) / S2_BLOCK)), dim3(S2_BLOCK), 0, st>>>((const s2amdShape*)s->dShapes.p, s->shapeCapacity,
																											  (TreeViews*)s->trees->dViews.p, stepFailed);
}

void launchTreeRebuild(s2amdSolver* s, hipStream_t main)
{
	if (!treesActive(s))
	{
		return;
	}
	DeviceTrees* d = s->trees;
	TreeViews* v = (TreeViews*)d->dViews.p;
	hipStream_t st = main;
	const bool beside = d->stream != nullptr && s->optTreeStream != 0;
	if (beside)
	{
		(void)hipEventRecord(d->evStart, main);
		(void)hipStreamWaitEvent(d->stream, d->evStart, 0);
		st = d->stream;
	}
	for (int which = 1; which <= 2; ++which)
	{
		const int C = d->h.t[which].capacity;
		if (C <= 0 || d->leafCount[which] < 2)
		{
			continue; // (a tree of one leaf or none has no internal node to flag)
		}
		const unsigned blocks = (unsigned)std::min(std::max(C / S2_BLOCK, 1), 256);
		treeGatherInitKernel<<<dim3(blocks), dim3(S2_BLOCK), 0, st>>>(v, which);
		treeGatherWalkKernel<<<dim3(blocks), dim3(S2_BLOCK), 0, st>>>(v, which);
		treeGatherScanKernel<<<dim3(1), dim3(TREE_THREADS), 0, st>>>(v, which);
		treeGatherPlaceKernel<<<dim3(blocks), dim3(S2_BLOCK), 0, st>>>(v, which);
		treeBuildTasksKernel<<<dim3((unsigned)std::min(std::max(C / TREE_SMALL, 1), 64)), dim3(TREE_THREADS), 0, st>>>(v, which);
	}
	if (beside)
	{
		(void)hipEventRecord(d->evDone, st);
		d->pending = true;
	}
}

// what the step's own stream does next reads or writes the trees: the rebuild beside it has to be through
void treesJoin(s2amdSolver* s, hipStream_t main)
{
	if (s->trees != nullptr && s->trees->pending)
	{
		(void)hipStreamWaitEvent(main, s->trees->evDone, 0);
		s->trees->pending = false;
	}
}

void launchOrderPairs(hipStream_t st, const TreeViews* views, const s2amdShape* shapes, const unsigned char* moved, const unsigned long long* keys,
					  const unsigned int* count, unsigned int cap, unsigned long long* ckeys, unsigned long long* out)
{
	pairCreationKeysKernel<<<dim3(64), dim3(S2_BLOCK), 0, st>>>(shapes, moved, views, keys, count, cap, ckeys);
	pairCreationScatterKernel<<<dim3(128), dim3(S2_BLOCK), 0, st>>>(keys, ckeys, count, cap, out);
}

#pragma GCC visibility push(default)
extern "C"
{

int s2amd_world_set_tree(s2amdSolver* s, int32_t bodyType, const s2amdTreeNode* nodes, int32_t nodeCapacity, int32_t root)
{
	if (!s || bodyType < 0 || bodyType > 2 || nodeCapacity < 0 || (nodeCapacity > 0 && !nodes) || root < -1 || root >= std::max(nodeCapacity, 0))
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (!s->worldResident)
	{
		return fail(S2AMD_E_STATE, "s2amd_world_set_tree called before s2amd_world_upload");
	}
	HIP_TRY(hipSetDevice(s->device));
	// what the kernels rely on: links in range and mutual, no flagged internal node (the state stage 2 leaves a tree in)
	const int C = nodeCapacity;
	std::vector<int> leaves((size_t)std::max(C, 1), 0);
	if (root != TREE_NULL)
	{
		std::vector<int> order;
		order.reserve((size_t)C);
		std::vector<int> stack{root};
		if (nodes[root].parent != TREE_NULL)
		{
			return fail(S2AMD_E_INVALID, "tree root has a parent");
		}
		while (!stack.empty())
		{
			const int n = stack.back();
			stack.pop_back();
			if ((int)order.size() >= C)
			{
				return fail(S2AMD_E_INVALID, "tree has a cycle");
			}
			order.push_back(n);
			const s2amdTreeNode& nd = nodes[n];
			if (nd.height < 0)
			{
				return fail(S2AMD_E_INVALID, "tree reaches a free node");
			}
			if (nd.height > 0)
			{
				if (nd.enlarged != 0)
				{
					return fail(S2AMD_E_INVALID, "tree holds a flagged internal node: upload it as stage 2 leaves it (src/world.c:130)");
				}
				for (int c : {nd.child1, nd.child2})
				{
					if (c < 0 || c >= C || nodes[c].parent != n)
					{
						return fail(S2AMD_E_INVALID, "tree links are inconsistent at node " + std::to_string(n));
					}
					stack.push_back(c);
				}
			}
		}
		for (size_t k = order.size(); k-- > 0;)
		{
			const int n = order[k];
			leaves[(size_t)n] = nodes[n].height == 0 ? 1 : leaves[(size_t)nodes[n].child1] + leaves[(size_t)nodes[n].child2];
		}
	}
	if (s->trees == nullptr)
	{
		s->trees = new DeviceTrees();
	}
	DeviceTrees* d = s->trees;
	if (d->stream == nullptr)
	{
		HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&d->evStart, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&d->evDone, hipEventDisableTiming));
	}
	treesJoin(s, s->stream);
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	const size_t n4 = al((size_t)std::max(C, 1) * 4 + 8);
	const bool rebuilt = bodyType != 0;
	// nodes | flag | leaves | state | (marked acc pending oldPre leafIdx seg segStart segEnd segSplit scan partner arrive cx cy | bounds)
	const size_t total = al((size_t)std::max(C, 1) * sizeof(s2amdTreeNode)) + 2 * n4 + 256 + (rebuilt ? 8 * n4 + (TREE_TASK_INTS + 1) * n4 + 256 : 0);
	int rc = d->block[bodyType].ensure(total);
	if (rc)
	{
		return rc;
	}
	char* p = (char*)d->block[bodyType].p;
	TreeView& t = d->h.t[bodyType];
	t = TreeView{};
	t.capacity = C;
	t.nodes = (s2amdTreeNode*)p, p += al((size_t)std::max(C, 1) * sizeof(s2amdTreeNode));
	t.flag = (int*)p, p += n4;
	t.leaves = (int*)p, p += n4;
	t.state = (int*)p, p += 256;
	if (rebuilt)
	{
		int** ints[] = {&t.acc, &t.pending, &t.oldPre, &t.leafIdx, &t.partner, &t.arrive};
		for (int** q : ints)
		{
			*q = (int*)p, p += n4;
		}
		t.cx = (float*)p, p += n4;
		t.cy = (float*)p, p += n4;
		t.tasks = (int*)p, p += TREE_TASK_INTS * n4;
		t.ready = (int*)p, p += n4;
		t.qstate = (int*)p, p += 256;
	}
	hipStream_t st = s->stream;
	HIP_TRY(hipMemsetAsync(d->block[bodyType].p, 0, total, st));
	if (C > 0)
	{
		HIP_TRY(hipMemcpyAsync(t.nodes, nodes, (size_t)C * sizeof(s2amdTreeNode), hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(t.leaves, leaves.data(), (size_t)C * sizeof(int), hipMemcpyHostToDevice, st));
	}
	const int state[4] = {root, 0, 0, 0};
	HIP_TRY(hipMemcpyAsync(t.state, state, sizeof(state), hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	d->set[bodyType] = true;
	d->leafCount[bodyType] = root != TREE_NULL ? leaves[(size_t)root] : 0;
	s->pairQuery.key = 0; // (the captured pair query does or does not order its pairs)
	s->pairQuery.keySeen = 0;
	s->pairCacheValid = false;
	if ((rc = pushViews(s)) != 0 || (rc = treesSyncRefitOrder(s)) != 0)
	{
		return rc;
	}
	// (the third tree makes the pair query an ordered one: its graph is captured now, not in the first step that asks)
	return treesActive(s) && s->optPrebuildSolver >= 0 ? worldWarmPairQuery(s) : S2AMD_OK;
}

int s2amd_world_get_tree(s2amdSolver* s, int32_t bodyType, s2amdTreeNode* nodes, int32_t nodeCapacity, int32_t* root)
{
	if (!s || bodyType < 0 || bodyType > 2 || nodeCapacity < 0 || (nodeCapacity > 0 && !nodes) || !root)
	{
		return fail(S2AMD_E_INVALID, "bad argument");
	}
	if (s->trees == nullptr || !s->trees->set[bodyType])
	{
		return fail(S2AMD_E_STATE, "no tree of this body type on the device (s2amd_world_set_tree)");
	}
	HIP_TRY(hipSetDevice(s->device));
	treesJoin(s, s->stream);
	const TreeView& t = s->trees->h.t[bodyType];
	if (nodeCapacity != t.capacity)
	{
		return fail(S2AMD_E_INVALID, "the tree on the device has " + std::to_string(t.capacity) + " nodes");
	}
	int state[4] = {0, 0, 0, 0};
	if (bodyType != 0 && t.capacity > 0 && s->trees->leafCount[bodyType] >= 2)
	{
		// (the boxes of the flagged nodes: the enlarge pass leaves them to whoever reads the tree)
		const unsigned blocks = (unsigned)std::min(std::max(t.capacity / S2_BLOCK, 1), 256);
		treeBoxesInitKernel<<<dim3(blocks), dim3(S2_BLOCK), 0, s->stream>>>((TreeViews*)s->trees->dViews.p, bodyType);
		treeBoxesClimbKernel<<<dim3(blocks), dim3(S2_BLOCK), 0, s->stream>>>((TreeViews*)s->trees->dViews.p, bodyType);
		HIP_TRY(hipGetLastError());
	}
	if (nodeCapacity > 0)
	{
		HIP_TRY(hipMemcpyAsync(nodes, t.nodes, (size_t)nodeCapacity * sizeof(s2amdTreeNode), hipMemcpyDeviceToHost, s->stream));
	}
	HIP_TRY(hipMemcpyAsync(state, t.state, sizeof(state), hipMemcpyDeviceToHost, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream));
	if (state[2] != 0)
	{
		return fail(S2AMD_E_DEVICE, "the device tree of body type " + std::to_string(bodyType) + " reported error " + std::to_string(state[2]));
	}
	*root = state[0];
	return S2AMD_OK;
}

} // extern "C"
#pragma GCC visibility pop

S2_DEFINE_WARM(tree_mirror)
