// The reference's broad-phase trees on the device ("sort emitted pairs into the reference's creation order").
//
// s2UpdateBroadPhasePairs creates contacts in the order its tree queries call back (src/broad_phase.c:253-254, :288-320, :332-357),
// so a contact's pool slot is a function of the TOPOLOGY of the reference's three s2DynamicTrees (src/broad_phase.h:27) at the moment
// of the query -- and that topology is history: stage 2 of every step rebuilds only the part of a tree that stage 4 of the step before
// flagged (s2DynamicTree_Rebuild(tree, false), src/dynamic_tree.c:1764-1874), the rest is kept as it was built from the boxes of
// earlier steps.  There is no key of the current boxes that gives the order; the trees have to be maintained.  Rounds 3-5 did that on
// the host, replaying the reference's own functions (shim/s2_amd_binding.c: flushTrees, s2amdBinding_OrderPairs: 1.5 ms per step at
// base 200 whenever a pair is created).  This file keeps the node arrays in HBM instead and maintains them with three kernels:
//
//   treeEnlargeKernel   stage 4's s2DynamicTree_EnlargeProxy (src/dynamic_tree.c:803-839, called from src/world.c:283-290) for every shape
//                       the refit re-inflated: leaf box replaced, ancestors' boxes grown (CAS on '<', the reference's comparison) and
//                       flagged; boxes only grow and flags are only set, so any order gives the reference's result
//   treeRebuildKernel   stage 2's rebuild, one 1024-thread workgroup per tree: the flagged region is measured bottom-up, the gathered
//                       leaves take their depth-first positions by a walk up each, s2BuildTree's recursive median split (:1610-1761,
//                       s2PartitionMid :1317-1427) runs level by level over all open segments at once -- the Hoare loop of a segment is
//                       a fixed permutation that a prefix sum of the predicate gives -- and boxes / heights / category bits are
//                       finished bottom-up.  Node IDS are the reference's too (the k-th new node in pre-order takes the (M-1-k)-th freed
//                       one, as the free list would hand them out), so the arrays can be copied back into the host's s2DynamicTree
//                       and proxies created later get the ids the reference gives them.
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

static_assert(sizeof(s2amdTreeNode) == 48, "s2TreeNode is 48 bytes (include/solver2d/dynamic_tree.h:14-41)");

namespace
{
S2_DEV bool treeFlagged(const TreeView& t, int n)
{
	return t.nodes[n].height > 0 && t.flag[n] != 0;
}

// *addr = min(*addr, v) by the reference's comparison (s2AABB_Enlarge, include/solver2d/aabb.h:62-90: `b < a`); returns whether it changed
S2_DEV bool casMin(float* addr, float v)
{
	int* ia = (int*)addr;
	int old = __atomic_load_n(ia, __ATOMIC_RELAXED);
	while (v < __int_as_float(old))
	{
		const int seen = atomicCAS(ia, old, __float_as_int(v));
		if (seen == old)
		{
			return true;
		}
		old = seen;
	}
	return false;
}
S2_DEV bool casMax(float* addr, float v)
{
	int* ia = (int*)addr;
	int old = __atomic_load_n(ia, __ATOMIC_RELAXED);
	while (__int_as_float(old) < v)
	{
		const int seen = atomicCAS(ia, old, __float_as_int(v));
		if (seen == old)
		{
			return true;
		}
		old = seen;
	}
	return false;
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
} // namespace

// ---- stage 4: the shapes the refit re-inflated enlarge their proxies ----
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
	const float b0 = sh.fatAABB[0], b1 = sh.fatAABB[1], b2 = sh.fatAABB[2], b3 = sh.fatAABB[3];
	t.nodes[leaf].aabb[0] = b0, t.nodes[leaf].aabb[1] = b1, t.nodes[leaf].aabb[2] = b2, t.nodes[leaf].aabb[3] = b3;
	int p = t.nodes[leaf].parent;
	int guard = 0;
	while (p != TREE_NULL && guard++ < 4096)
	{
		s2amdTreeNode& n = t.nodes[p];
		bool changed = casMin(&n.aabb[0], b0);
		changed = casMin(&n.aabb[1], b1) || changed;
		changed = casMax(&n.aabb[2], b2) || changed;
		changed = casMax(&n.aabb[3], b3) || changed;
		const bool first = atomicExch(t.flag + p, 1) == 0;
		if (first)
		{
			n.enlarged = 1;
			t.marked[atomicAdd(t.state + 1, 1)] = p;
		}
		if (!changed && !first)
		{
			// whoever flagged this node is on its way to the root, and a box that holds this one already is held by its ancestors
			// -- or will be, by the box that grew it
			break;
		}
		p = n.parent;
	}
}

// ---- stage 2: s2DynamicTree_Rebuild(tree, false) ----
namespace
{
// exclusive prefix sum of one int per thread over the workgroup; *total = the sum
S2_DEV int blockExclusive(int v, int* lds, int* total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1)
	{
		const int y = __shfl_up(x, d, 64);
		if (lane >= d)
		{
			x += y;
		}
	}
	if (lane == 63)
	{
		lds[wave] = x;
	}
	__syncthreads();
	if (wave == 0)
	{
		const int w = lane < TREE_THREADS / 64 ? lds[lane] : 0;
		int xs = w;
#pragma unroll
		for (int d = 1; d < TREE_THREADS / 64; d <<= 1)
		{
			const int y = __shfl_up(xs, d, 64);
			if (lane >= d)
			{
				xs += y;
			}
		}
		if (lane < TREE_THREADS / 64)
		{
			lds[lane] = xs - w;
		}
		if (lane == TREE_THREADS / 64 - 1)
		{
			lds[TREE_THREADS / 64] = xs;
		}
	}
	__syncthreads();
	*total = lds[TREE_THREADS / 64];
	const int r = lds[wave] + x - v;
	__syncthreads();
	return r;
}

// does element i of an open segment go left of the pivot?  (s2PartitionMid, src/dynamic_tree.c:1326-1352: bounds of the centres,
// the longer axis, pivot = the middle of the bounds)
S2_DEV bool goesLeft(const TreeView& t, int s, int i)
{
	const unsigned int* b = t.bounds + 4 * (size_t)s;
	const float lx = unsortable(b[0]), ly = unsortable(b[1]), ux = unsortable(b[2]), uy = unsortable(b[3]);
	const float dx = ux - lx, dy = uy - ly;
	if (dx > dy)
	{
		const float pivot = 0.5f * (lx + ux);
		return t.cx[i] < pivot;
	}
	const float pivot = 0.5f * (ly + uy);
	return t.cy[i] < pivot;
}
} // namespace

__global__ __launch_bounds__(TREE_THREADS) void treeRebuildKernel(TreeViews* views)
{
	__shared__ int lds[TREE_THREADS / 64 + 2];
	const int which = 1 + (int)blockIdx.x; // kinematic, dynamic (src/broad_phase.c:349-353 rebuilds those two)
	TreeView& t = views->t[which];
	const int tid = threadIdx.x;
	const int M = t.capacity > 0 ? t.state[1] : 0;
	const int root = t.capacity > 0 ? t.state[0] : TREE_NULL;
	if (M <= 0 || root == TREE_NULL || !treeFlagged(t, root))
	{
		if (tid == 0 && M > 0)
		{
			atomicExch(t.state + 2, 2); // flags without a flagged root: the closure the upload checked is gone
		}
		return;
	}
	const int L = M + 1;
	s2amdTreeNode* nodes = t.nodes;

	// A. flagged nodes below every flagged node, bottom-up: a node is finished by whichever child arrives second
	for (int k = tid; k < M; k += TREE_THREADS)
	{
		const int n = t.marked[k];
		t.acc[n] = 1;
		t.pending[n] = (treeFlagged(t, nodes[n].child1) ? 1 : 0) + (treeFlagged(t, nodes[n].child2) ? 1 : 0);
	}
	__syncthreads();
	for (int k = tid; k < M; k += TREE_THREADS)
	{
		int n = t.marked[k];
		if (treeFlagged(t, nodes[n].child1) || treeFlagged(t, nodes[n].child2))
		{
			continue;
		}
		for (;;)
		{
			if (n == root)
			{
				break;
			}
			const int p = nodes[n].parent;
			atomicAdd(t.acc + p, __atomic_load_n(t.acc + n, __ATOMIC_RELAXED));
			__threadfence();
			if (atomicSub(t.pending + p, 1) != 1)
			{
				break;
			}
			__threadfence();
			n = p;
		}
	}
	__syncthreads();
	if (__atomic_load_n(t.acc + root, __ATOMIC_RELAXED) != M)
	{
		if (tid == 0)
		{
			atomicExch(t.state + 2, 3); // a flagged node whose parent is not flagged
		}
		return;
	}
	// B. pre-order index of every flagged node (the order they are freed in, :1838-1853), depth-first index of every gathered leaf
	for (int k = tid; k < M; k += TREE_THREADS)
	{
		const int n = t.marked[k];
		int idx = 0;
		for (int x = n; x != root;)
		{
			const int p = nodes[x].parent;
			const int c1 = nodes[p].child1;
			idx += 1 + ((nodes[p].child2 == x && treeFlagged(t, c1)) ? t.acc[c1] : 0);
			x = p;
		}
		t.oldPre[idx] = n;
		for (int side = 0; side < 2; ++side)
		{
			const int c = side == 0 ? nodes[n].child1 : nodes[n].child2;
			if (treeFlagged(t, c))
			{
				continue;
			}
			int li = 0;
			for (int x = c, p = n;;)
			{
				if (nodes[p].child2 == x)
				{
					const int c1 = nodes[p].child1;
					li += 1 + (treeFlagged(t, c1) ? t.acc[c1] : 0);
				}
				if (p == root)
				{
					break;
				}
				x = p;
				p = nodes[p].parent;
			}
			t.leafIdx[li] = c;
			t.cx[li] = 0.5f * (nodes[c].aabb[0] + nodes[c].aabb[2]); // s2AABB_Center, include/solver2d/aabb.h:28-32
			t.cy[li] = 0.5f * (nodes[c].aabb[1] + nodes[c].aabb[3]);
			t.seg[li] = 0;
		}
	}
	__syncthreads();
	// (from here on the flagged nodes are the free list's: node_of(pre) hands them out as s2AllocateNode would, :105-139)
	auto nodeOf = [&](int pre) { return t.oldPre[M - 1 - pre]; };
	auto makeNode = [&](int pre, int parent, int start, int end) {
		const int id = nodeOf(pre);
		s2amdTreeNode& n = nodes[id];
		n.aabb[0] = 0.0f, n.aabb[1] = 0.0f, n.aabb[2] = 0.0f, n.aabb[3] = 0.0f;
		n.categoryBits = 0u;
		n.parent = parent;
		n.child1 = TREE_NULL, n.child2 = TREE_NULL;
		n.userData = -1;
		n.height = -2;
		n.enlarged = 0;
		t.flag[id] = 0;
		t.arrive[id] = 0;
		t.segStart[pre] = start, t.segEnd[pre] = end;
		t.bounds[4 * (size_t)pre + 0] = 0xffffffffu, t.bounds[4 * (size_t)pre + 1] = 0xffffffffu;
		t.bounds[4 * (size_t)pre + 2] = 0u, t.bounds[4 * (size_t)pre + 3] = 0u;
		return id;
	};
	if (tid == 0)
	{
		makeNode(0, TREE_NULL, 0, L);
	}
	__syncthreads();

	// C. s2BuildTree, one level of all open segments per turn.  seg[i] = pre-order index of the node whose segment element i is in.
	const int chunk = (L + TREE_THREADS - 1) / TREE_THREADS;
	for (int level = 0; level <= L; ++level)
	{
		// bounds of the centres of every segment with more than two elements
		for (int i = tid; i < L; i += TREE_THREADS)
		{
			const int s = t.seg[i];
			if (s < 0 || t.segEnd[s] - t.segStart[s] <= 2)
			{
				continue;
			}
			unsigned int lx = sortable(t.cx[i]), ly = sortable(t.cy[i]), ux = lx, uy = ly;
			// a wave whose lanes are all in one segment reduces first
			const int s0 = __shfl(s, __ffsll((long long)__ballot(1)) - 1, 64);
			if (__all(s == s0) && __popcll(__ballot(1)) == 64)
			{
#pragma unroll
				for (int d = 32; d >= 1; d >>= 1)
				{
					lx = min(lx, (unsigned int)__shfl_xor((int)lx, d, 64));
					ly = min(ly, (unsigned int)__shfl_xor((int)ly, d, 64));
					ux = max(ux, (unsigned int)__shfl_xor((int)ux, d, 64));
					uy = max(uy, (unsigned int)__shfl_xor((int)uy, d, 64));
				}
				if ((tid & 63) != 0)
				{
					continue;
				}
			}
			unsigned int* b = t.bounds + 4 * (size_t)s;
			atomicMin(b + 0, lx), atomicMin(b + 1, ly), atomicMax(b + 2, ux), atomicMax(b + 3, uy);
		}
		__syncthreads();
		// prefix sum of the predicate over the whole leaf array (scan[i] = elements left of their pivot before i; scan[L] = all)
		{
			const int lo = min(tid * chunk, L), hi = min(lo + chunk, L);
			int mine = 0;
			for (int i = lo; i < hi; ++i)
			{
				const int s = t.seg[i];
				mine += (s >= 0 && t.segEnd[s] - t.segStart[s] > 2 && goesLeft(t, s, i)) ? 1 : 0;
			}
			int total = 0;
			int run = blockExclusive(mine, lds, &total);
			for (int i = lo; i < hi; ++i)
			{
				const int s = t.seg[i];
				t.scan[i] = run;
				run += (s >= 0 && t.segEnd[s] - t.segStart[s] > 2 && goesLeft(t, s, i)) ? 1 : 0;
			}
			if (tid == 0)
			{
				t.scan[L] = total;
			}
		}
		__syncthreads();
		// the Hoare loop's exchanges (:1357-1420): the j-th misplaced element from the left with the j-th from the right
		for (int i = tid; i < L; i += TREE_THREADS)
		{
			const int s = t.seg[i];
			if (s < 0)
			{
				continue;
			}
			const int start = t.segStart[s], end = t.segEnd[s], n = end - start;
			int split = n / 2; // (:1320-1323 two elements or fewer; :1422-1429 nothing on one side)
			if (n > 2)
			{
				const int m = t.scan[end] - t.scan[start];
				if (m > 0 && m < n)
				{
					split = m;
					const int before = t.scan[i] - t.scan[start];
					if (i - start >= m && goesLeft(t, s, i))
					{
						t.partner[start + (m - before - 1)] = i;
					}
				}
			}
			if (i == start)
			{
				t.segSplit[s] = split;
			}
		}
		__syncthreads();
		for (int i = tid; i < L; i += TREE_THREADS)
		{
			const int s = t.seg[i];
			if (s < 0)
			{
				continue;
			}
			const int start = t.segStart[s], end = t.segEnd[s], n = end - start;
			const int m = t.scan[end] - t.scan[start];
			if (n > 2 && m > 0 && m < n && i - start < m && !goesLeft(t, s, i))
			{
				const int before = t.scan[i] - t.scan[start];
				const int p = t.partner[start + (i - start - before)];
				const int li = t.leafIdx[i];
				const float x = t.cx[i], y = t.cy[i];
				t.leafIdx[i] = t.leafIdx[p], t.cx[i] = t.cx[p], t.cy[i] = t.cy[p];
				t.leafIdx[p] = li, t.cx[p] = x, t.cy[p] = y;
			}
		}
		__syncthreads();
		// children: a part of one element is that leaf, a longer one a new node (allocated in pre-order: :1622, :1716) with its own segment
		int open = 0;
		for (int i = tid; i < L; i += TREE_THREADS)
		{
			const int s = t.seg[i];
			if (s < 0)
			{
				continue;
			}
			const int start = t.segStart[s], end = t.segEnd[s], split = t.segSplit[s];
			const bool left = i - start < split;
			const int ps = left ? start : start + split, pe = left ? start + split : end;
			const int me = nodeOf(s);
			if (pe - ps == 1)
			{
				const int c = t.leafIdx[i];
				nodes[c].parent = me;
				(left ? nodes[me].child1 : nodes[me].child2) = c;
				t.seg[i] = -1;
			}
			else
			{
				const int cs = left ? s + 1 : s + split;
				if (i == ps)
				{
					const int c = makeNode(cs, me, ps, pe);
					(left ? nodes[me].child1 : nodes[me].child2) = c;
				}
				t.seg[i] = cs;
				open = 1;
			}
		}
		// (segStart / segEnd of the parents are read above and those of the children written: different entries, a child's index is new)
		if (__syncthreads_or(open) == 0)
		{
			break;
		}
	}
	__threadfence();
	__syncthreads();
	// D. boxes, heights, category bits and leaf counts of the new nodes, bottom-up (:1655-1657, :1742-1744)
	const int newRoot = nodeOf(0);
	for (int i = tid; i < L; i += TREE_THREADS)
	{
		int n = nodes[t.leafIdx[i]].parent;
		for (;;)
		{
			__threadfence();
			if (atomicAdd(t.arrive + n, 1) == 0)
			{
				break;
			}
			__threadfence();
			const int a = nodes[n].child1, b = nodes[n].child2;
			const s2amdTreeNode& na = nodes[a];
			const s2amdTreeNode& nb = nodes[b];
			s2amdTreeNode& nn = nodes[n];
			// s2AABB_Union, include/solver2d/aabb.h:42-50 (s2MinFloat / s2MaxFloat: a < b ? a : b)
			nn.aabb[0] = na.aabb[0] < nb.aabb[0] ? na.aabb[0] : nb.aabb[0];
			nn.aabb[1] = na.aabb[1] < nb.aabb[1] ? na.aabb[1] : nb.aabb[1];
			nn.aabb[2] = na.aabb[2] > nb.aabb[2] ? na.aabb[2] : nb.aabb[2];
			nn.aabb[3] = na.aabb[3] > nb.aabb[3] ? na.aabb[3] : nb.aabb[3];
			nn.height = (int16_t)(1 + (na.height > nb.height ? na.height : nb.height));
			nn.categoryBits = na.categoryBits | nb.categoryBits;
			t.leaves[n] = t.leaves[a] + t.leaves[b];
			if (n == newRoot)
			{
				break;
			}
			n = nn.parent;
		}
	}
	__syncthreads();
	if (tid == 0)
	{
		t.state[0] = newRoot;
		t.state[1] = 0;
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
	treeEnlargeKernel<<<dim3((unsigned)((s->shapeCapacity + S2_BLOCK - 1) / S2_BLOCK)), dim3(S2_BLOCK), 0, st>>>((const s2amdShape*)s->dShapes.p, s->shapeCapacity,
																											  (TreeViews*)s->trees->dViews.p, stepFailed);
}

void launchTreeRebuild(s2amdSolver* s, hipStream_t st)
{
	if (!treesActive(s))
	{
		return;
	}
	treeRebuildKernel<<<dim3(2), dim3(TREE_THREADS), 0, st>>>((TreeViews*)s->trees->dViews.p);
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
	auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
	const size_t n4 = al((size_t)std::max(C, 1) * 4 + 8);
	const bool rebuilt = bodyType != 0;
	// nodes | flag | leaves | state | (marked acc pending oldPre leafIdx seg segStart segEnd segSplit scan partner arrive cx cy | bounds)
	const size_t total = al((size_t)std::max(C, 1) * sizeof(s2amdTreeNode)) + 2 * n4 + 256 + (rebuilt ? 14 * n4 + 4 * n4 : 0);
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
		int** ints[] = {&t.marked, &t.acc, &t.pending, &t.oldPre, &t.leafIdx, &t.seg, &t.segStart, &t.segEnd, &t.segSplit, &t.scan, &t.partner, &t.arrive};
		for (int** q : ints)
		{
			*q = (int*)p, p += n4;
		}
		t.cx = (float*)p, p += n4;
		t.cy = (float*)p, p += n4;
		t.bounds = (unsigned int*)p, p += 4 * n4;
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
	const TreeView& t = s->trees->h.t[bodyType];
	if (nodeCapacity != t.capacity)
	{
		return fail(S2AMD_E_INVALID, "the tree on the device has " + std::to_string(t.capacity) + " nodes");
	}
	int state[4] = {0, 0, 0, 0};
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
