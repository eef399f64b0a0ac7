// LDS group kernel: one workgroup advances one group through a list of step Ops with the group's
// body velocities and poses staged in LDS (32 B per body), constraints streamed from their SoA
// arrays, `__syncthreads()` between colour batches instead of kernel boundaries.
//
//  * whole-step mode: the group is a set of small simulation islands; the op list is the complete
//    driver of the solver (all sub-steps), so the island costs ONE launch per step.  512 base-40
//    pyramids = 512 workgroups, two per CU.
//  * single-op mode: the group is the "sequential tail" of a big island -- the constraints that
//    greedy colouring left in dozens of tiny colours because one body touches dozens of others
//    (tumbler drum).  They are swept by one lane in sweep order with their bodies in LDS.
//
// Arithmetic is constraint_ops.h / body_ops.h, i.e. identical to the global kernels; sweep order is
// (batch-major, index within batch), reported to the host like any other order.

#include "body_ops.h"
#include "group_ops.h"

#define S2_GROUP_THREADS 512
#define S2_GROUP_STAGED_BATCHES 192 // colour batches (contacts + joints) of a group and ...
#define S2_GROUP_STAGED_OPS 128		// ... ops of a step whose records the kernel keeps in LDS, where they fit beside the bodies

// The sequential tail, one WAVE instead of one lane: every lane pulls its own constraint into registers (the misses of 64
// constraints in flight at once, no dependent L2 round trip left in the walk), then the lanes take turns in sweep order.
// Bodies go through LDS, which one wave reads and writes in program order; the wavefront fence keeps the compiler from
// moving a lane's LDS reads across the turn before it.  Same per-constraint arithmetic, same order as forBatches.
// (r4) Only the lanes whose constraint HAS manifold points take a turn: a constraint without points writes no body (its record has the
// write bits cleared), so leaving its turn out changes no bit -- and the tail now ends in free positions for created contacts
// (IncrementalGlobal::tailFree: 64 empty records that cost a turn each before this) and holds the hubs' potential constraints.
template <class R> S2_DEV auto tailLive(const R& r, int) -> decltype(r.h.pointCount, bool()) { return r.h.pointCount > 0; }
template <class R> S2_DEV auto tailLive(const R& r, long) -> decltype(r.r.h.pointCount, bool()) { return r.r.h.pointCount > 0; }
template <class R> S2_DEV bool tailLive(const R&, ...) { return true; }
template <class R, class L, class F, class S> S2_DEV void walkTail(int begin, int end, L load, F compute, S store)
{
	if (threadIdx.x < 64)
	{
		const int lane = (int)threadIdx.x;
		for (int base = begin; base < end; base += 64)
		{
			const int k = base + lane;
			const int n = min(64, end - base);
			R r = load(min(k, end - 1));
			unsigned long long turns = __ballot(lane < n && tailLive(r, 0));
			while (turns != 0ull)
			{
				const int j = __ffsll((long long)turns) - 1;
				turns &= turns - 1ull;
				if (lane == j)
				{
					compute(r, k);
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			}
			if (lane < n)
			{
				store(r, k);
			}
		}
	}
}

// a tail constraint of the soft sweeps between its two parts (constraint_ops.h: prepSoft / chainSoft)
template <int KIND> struct TailSoft
{
	SoftRegs<KIND> r;
	SoftPre pre;
	SoftPerp perp;
};

// forBatches with the tail batches walked by walkTail
template <class R, class L, class C, class S, class F> S2_DEV void forBatchesSplit(const int4* batches, int b0, int b1, L load, C compute, S store, F f)
{
	for (int bi = b0; bi < b1; ++bi)
	{
		int4 bt = batches[bi];
		if (bt.z)
		{
			walkTail<R>(bt.x, bt.y, load, compute, store);
		}
		else
		{
			for (int k = bt.x + (int)threadIdx.x; k < bt.y; k += (int)blockDim.x)
			{
				f(k);
			}
		}
		__syncthreads();
	}
}

// Soft sweep with PRELOADED rounds: every thread first issues the loads of its constraint in each of the
// next MAXR colour batches (all in flight at once: one memory round trip for the whole chunk instead of
// one per colour), then the batches are swept in order with only LDS traffic between barriers.  Same
// per-constraint arithmetic and the same order as forBatches.
template <int KIND, int MAXR, class LB>
S2_DEV void sweepSoftPreloaded(const ContactView& c, const LB& lb, const int4* batches, int b0, int b1, float inv_h, int useBias)
{
	for (int base = b0; base < b1; base += MAXR)
	{
		SoftRegs<KIND> r[MAXR];
		int kk[MAXR];
#pragma unroll
		for (int i = 0; i < MAXR; ++i)
		{
			kk[i] = -1;
			if (base + i < b1)
			{
				int4 bt = batches[base + i];
				int k = bt.x + (int)threadIdx.x;
				if (bt.z == 0 && k < bt.y)
				{
					kk[i] = k;
					r[i] = loadSoftB<KIND>(c, lb, k);
				}
			}
		}
#pragma unroll
		for (int i = 0; i < MAXR; ++i)
		{
			if (base + i < b1)
			{
				int4 bt = batches[base + i];
				if (bt.z)
				{
					for (int k = bt.x + (int)threadIdx.x; k < bt.y; k += (int)blockDim.x)
					{
						prefetchContact(c, k);
					}
					__syncthreads();
					if (threadIdx.x == 0)
					{
						for (int k = bt.x; k < bt.y; ++k)
						{
							solveContactsSoftOne<KIND>(c, lb, inv_h, useBias, k);
						}
					}
				}
				else
				{
					if (kk[i] >= 0)
					{
						solveSoftRegs<KIND>(r[i], c, lb, inv_h, useBias, kk[i]);
						storeSoft<KIND>(c, r[i], kk[i]);
					}
					// a batch wider than the workgroup: the rest streams
					for (int k = bt.x + (int)threadIdx.x + (int)blockDim.x; k < bt.y; k += (int)blockDim.x)
					{
						solveContactsSoftOne<KIND>(c, lb, inv_h, useBias, k);
					}
				}
				__syncthreads();
			}
		}
	}
}

template <int THREADS, int PRELOAD>
__global__ __launch_bounds__(THREADS) void groupKernel(ContactView c, JointView jv, BodyView g, GroupTable gt, const Op* opsGlobal, int opCount, StepConsts sc,
													   s2amdContact* wire, int useDq0, int stageTables)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	const int grp = blockIdx.x;
	const int bodyBase = gt.bodyOffsets[grp];
	const int nb = gt.bodyOffsets[grp + 1] - bodyBase;
	float4* lvel = lds;
	float4* ldq = lds + nb;
	float4* ldq0 = lds + 2 * nb; // only addressed when useDq0
	float2* lmass = (float2*)(lds + (useDq0 ? 3 : 2) * nb);
	const int* ids = gt.bodyIds + bodyBase;

	for (int i = threadIdx.x; i < nb; i += blockDim.x)
	{
		int gi = (int)((uint32_t)ids[i] & ~S2G_OWNED);
		lvel[i] = g.vel[gi];
		ldq[i] = g.dq[gi];
		lmass[i] = g.massInv[gi];
		if (useDq0)
		{
			ldq0[i] = g.dq0[gi];
		}
	}
	__syncthreads();

	LdsMassBodies lb;
	lb.vel = lvel, lb.dq = ldq, lb.massInv = lmass;
	lb.softCoef[0] = make_float4(sc.softCoef[0][0], sc.softCoef[0][1], sc.softCoef[0][2], 0.0f);
	lb.softCoef[1] = make_float4(sc.softCoef[1][0], sc.softCoef[1][1], sc.softCoef[1][2], 0.0f);
	lb.softDiet = sc.softDiet;
	auto pfC = [&](int k) { prefetchContact(c, k); };
	auto pfJ = [&](int k) { prefetchJoint(jv, k); };
	// The colour batches' descriptors and the op list in LDS (r6): a round starts by reading its {begin, end, tail} and an op by reading
	// its record -- from global memory each is a dependent L2 round trip on the critical path, ~900 of them in a TGS_Soft step of a card
	// house whose 161 constraints take 1.35 ms (the op interpreter of the strips has done this since r4).  Where the host found room.
	const int cbG = gt.cBatchOffsets[grp], jbG = gt.jBatchOffsets[grp];
	const int nCB = gt.cBatchOffsets[grp + 1] - cbG, nJB = gt.jBatchOffsets[grp + 1] - jbG;
	const int4* batchC = gt.cBatches + cbG;
	const int4* batchJ = gt.jBatches + jbG;
	const Op* ops = opsGlobal;
	if (stageTables != 0 && nCB + nJB <= S2_GROUP_STAGED_BATCHES && opCount <= S2_GROUP_STAGED_OPS)
	{
		int4* lbatch = (int4*)(lds + (useDq0 ? 3 : 2) * nb + (nb + 1) / 2);
		Op* lops = (Op*)(lbatch + S2_GROUP_STAGED_BATCHES);
		for (int i = threadIdx.x; i < nCB + nJB; i += blockDim.x)
		{
			lbatch[i] = i < nCB ? batchC[i] : batchJ[i - nCB];
		}
		for (int i = threadIdx.x; i < opCount * 8; i += blockDim.x)
		{
			((int*)lops)[i] = ((const int*)opsGlobal)[i];
		}
		batchC = lbatch, batchJ = lbatch + nCB, ops = lops;
		__syncthreads();
	}
	const int cb0 = 0, cb1 = nCB, jb0 = 0, jb1 = nJB;

	for (int oi = 0; oi < opCount; ++oi)
	{
		const Op op = ops[oi];
		switch (op.code)
		{
			case OP_INTEGRATE_VEL:
				for (int i = threadIdx.x; i < nb; i += blockDim.x)
				{
					integrateVelocitiesOne(lb, i, g, (int)((uint32_t)ids[i] & ~S2G_OWNED));
				}
				__syncthreads();
				break;
			case OP_INTEGRATE_POS:
				for (int i = threadIdx.x; i < nb; i += blockDim.x)
				{
					integratePositionsOne(lb, i, g, (int)((uint32_t)ids[i] & ~S2G_OWNED), op.h);
				}
				__syncthreads();
				break;
			case OP_FINALIZE:
				for (int i = threadIdx.x; i < nb; i += blockDim.x)
				{
					uint32_t id = (uint32_t)ids[i];
					finalizePositionsOne(lb, i, g, (int)(id & ~S2G_OWNED), op.flag, (id & S2G_OWNED) != 0);
				}
				__syncthreads();
				break;
			case OP_XPBD_INTEGRATE:
				for (int i = threadIdx.x; i < nb; i += blockDim.x)
				{
					xpbdIntegrateOne(lb, ldq0, i, g, (int)((uint32_t)ids[i] & ~S2G_OWNED), op.h);
				}
				__syncthreads();
				break;
			case OP_XPBD_PROJECT:
				for (int i = threadIdx.x; i < nb; i += blockDim.x)
				{
					xpbdProjectOne(lb, ldq0, i, g, (int)((uint32_t)ids[i] & ~S2G_OWNED), op.inv_h);
				}
				__syncthreads();
				break;
			case OP_JOINT_SWEEP:
				switch (op.kind)
				{
					case JSOLVE_PLAIN:
						forBatches(batchJ, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_PLAIN>(jv, lb, sc, op.h, op.inv_h, op.useBias, k); });
						break;
					case JSOLVE_SOFT:
						forBatches(batchJ, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_SOFT>(jv, lb, sc, op.h, op.inv_h, op.useBias, k); });
						break;
					case JSOLVE_BAUMGARTE:
						forBatches(batchJ, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_BAUMGARTE>(jv, lb, sc, op.h, op.inv_h, op.useBias, k); });
						break;
					case JSOLVE_POSITION:
						forBatches(batchJ, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_POSITION>(jv, lb, sc, op.h, op.inv_h, op.useBias, k); });
						break;
					case JSOLVE_XPBD:
						forBatches(batchJ, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_XPBD>(jv, lb, sc, op.h, op.inv_h, op.useBias, k); });
						break;
					case JSOLVE_WARM:
						forBatches(batchJ, jb0, jb1, pfJ, [&](int k) { solveJointsOne<JSOLVE_WARM>(jv, lb, sc, op.h, op.inv_h, op.useBias, k); });
						break;
				}
				break;
			case OP_WARM:
				switch (op.kind)
				{
					case WARM_CURRENT:
						forBatchesSplit<WarmRegs>(
							batchC, cb0, cb1, [&](int k) { return loadWarm<WARM_CURRENT>(c, lb, k); }, [&](WarmRegs& r, int) { applyWarm<WARM_CURRENT>(r, lb); },
							[](const WarmRegs&, int) {}, [&](int k) { warmStartContactsOne<WARM_CURRENT>(c, lb, k); });
						break;
					case WARM_FIXED:
						forBatchesSplit<WarmRegs>(
							batchC, cb0, cb1, [&](int k) { return loadWarm<WARM_FIXED>(c, lb, k); }, [&](WarmRegs& r, int) { applyWarm<WARM_FIXED>(r, lb); },
							[](const WarmRegs&, int) {}, [&](int k) { warmStartContactsOne<WARM_FIXED>(c, lb, k); });
						break;
					case WARM_BLOCK:
						forBatchesSplit<WarmRegs>(
							batchC, cb0, cb1, [&](int k) { return loadWarm<WARM_BLOCK>(c, lb, k); }, [&](WarmRegs& r, int) { applyWarm<WARM_BLOCK>(r, lb); },
							[](const WarmRegs&, int) {}, [&](int k) { warmStartContactsOne<WARM_BLOCK>(c, lb, k); });
						break;
				}
				break;
			case OP_SOLVE_SOFT:
				switch (op.kind)
				{
					case SOFT_TGS:
						if constexpr (PRELOAD > 0)
						{
							sweepSoftPreloaded<SOFT_TGS, PRELOAD>(c, lb, batchC, cb0, cb1, op.inv_h, op.useBias);
						}
						else
						{
							forBatchesSplit<TailSoft<SOFT_TGS>>(
								batchC, cb0, cb1,
								[&](int k) {
									TailSoft<SOFT_TGS> t;
									t.r = loadSoftB<SOFT_TGS>(c, lb, k);
									t.pre = prepSoft<SOFT_TGS>(t.r, lb, op.inv_h, op.useBias); // poses only: every lane at once
									t.perp = perpOf(t.pre);
									return t;
								},
								[&](TailSoft<SOFT_TGS>& t, int) { chainSoftPacked<SOFT_TGS>(t.r, t.pre, t.perp, lb); }, // velocities: lane after lane
								[&](const TailSoft<SOFT_TGS>& t, int k) { storeSoft<SOFT_TGS>(c, t.r, k); },
								[&](int k) { solveContactsSoftOne<SOFT_TGS>(c, lb, op.inv_h, op.useBias, k); });
						}
						break;
					case SOFT_PGS:
						if constexpr (PRELOAD > 0)
						{
							sweepSoftPreloaded<SOFT_PGS, PRELOAD>(c, lb, batchC, cb0, cb1, op.inv_h, op.useBias);
						}
						else
						{
							forBatchesSplit<TailSoft<SOFT_PGS>>(
								batchC, cb0, cb1,
								[&](int k) {
									TailSoft<SOFT_PGS> t;
									t.r = loadSoftB<SOFT_PGS>(c, lb, k);
									t.pre = prepSoft<SOFT_PGS>(t.r, lb, op.inv_h, op.useBias); // poses only: every lane at once
									t.perp = perpOf(t.pre);
									return t;
								},
								[&](TailSoft<SOFT_PGS>& t, int) { chainSoftPacked<SOFT_PGS>(t.r, t.pre, t.perp, lb); }, // velocities: lane after lane
								[&](const TailSoft<SOFT_PGS>& t, int k) { storeSoft<SOFT_PGS>(c, t.r, k); },
								[&](int k) { solveContactsSoftOne<SOFT_PGS>(c, lb, op.inv_h, op.useBias, k); });
						}
						break;
					case SOFT_FIXED:
						if constexpr (PRELOAD > 0)
						{
							sweepSoftPreloaded<SOFT_FIXED, PRELOAD>(c, lb, batchC, cb0, cb1, op.inv_h, op.useBias);
						}
						else
						{
							forBatchesSplit<TailSoft<SOFT_FIXED>>(
								batchC, cb0, cb1,
								[&](int k) {
									TailSoft<SOFT_FIXED> t;
									t.r = loadSoftB<SOFT_FIXED>(c, lb, k);
									t.pre = prepSoft<SOFT_FIXED>(t.r, lb, op.inv_h, op.useBias); // poses only: every lane at once
									t.perp = perpOf(t.pre);
									return t;
								},
								[&](TailSoft<SOFT_FIXED>& t, int) { chainSoftPacked<SOFT_FIXED>(t.r, t.pre, t.perp, lb); }, // velocities: lane after lane
								[&](const TailSoft<SOFT_FIXED>& t, int k) { storeSoft<SOFT_FIXED>(c, t.r, k); },
								[&](int k) { solveContactsSoftOne<SOFT_FIXED>(c, lb, op.inv_h, op.useBias, k); });
						}
						break;
					default:
						break; // SOFT_JACOBI never runs in a group (needs the per-body incidence sums)
				}
				break;
			case OP_SOLVE_RIGID:
				switch (op.kind)
				{
					case RIGID_BAUMGARTE:
						forBatches(batchC, cb0, cb1, pfC, [&](int k) { solveContactsRigidOne<RIGID_BAUMGARTE>(c, lb, op.inv_h, k); });
						break;
					case RIGID_PGS:
						forBatches(batchC, cb0, cb1, pfC, [&](int k) { solveContactsRigidOne<RIGID_PGS>(c, lb, op.inv_h, k); });
						break;
					case RIGID_TGS:
						forBatches(batchC, cb0, cb1, pfC, [&](int k) { solveContactsRigidOne<RIGID_TGS>(c, lb, op.inv_h, k); });
						break;
				}
				break;
			case OP_SOLVE_STICKY:
				forBatches(batchC, cb0, cb1, pfC, [&](int k) { solveContactsStickyOne(c, lb, wire, op.inv_h, op.useBias, k); });
				break;
			case OP_SOLVE_NGS:
				forBatches(batchC, cb0, cb1, pfC, [&](int k) { solveContactsNGSOne(c, lb, k); });
				break;
			case OP_XPBD_POS:
				forBatches(batchC, cb0, cb1, pfC, [&](int k) { xpbdContactPositionsOne(c, lb, op.h, k); });
				break;
			case OP_XPBD_VEL:
				forBatches(batchC, cb0, cb1, pfC, [&](int k) { xpbdContactVelocitiesOne(c, lb, op.h, k); });
				break;
			case OP_BLOCK_VEL:
				forBatches(batchC, cb0, cb1, pfC, [&](int k) { blockSolveVelocityOne(c, lb, k); });
				break;
			case OP_BLOCK_POS:
				forBatches(batchC, cb0, cb1, pfC, [&](int k) { blockSolvePositionOne(c, lb, k); });
				break;
			default:
				break;
		}
	}

	for (int i = threadIdx.x; i < nb; i += blockDim.x)
	{
		uint32_t id = (uint32_t)ids[i];
		if (id & S2G_OWNED)
		{
			int gi = (int)(id & ~S2G_OWNED);
			g.vel[gi] = lvel[i];
			g.dq[gi] = ldq[i];
			if (useDq0)
			{
				g.dq0[gi] = ldq0[i];
			}
		}
	}
}

#define S2_STRIP_THREADS 256
#define S2_STRIP_PRELOAD 6

int groupKernelSetup()
{
	// allow a group to use the full 160 KiB of LDS
	hipError_t e = hipFuncSetAttribute((const void*)groupKernel<S2_GROUP_THREADS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	if (e == hipSuccess)
	{
		e = hipFuncSetAttribute((const void*)groupKernel<S2_STRIP_THREADS, S2_STRIP_PRELOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	}
	return e == hipSuccess ? 0 : (int)e;
}

// strips: 256 threads (one wave per SIMD, the whole register file for preloaded rounds), one sweep op per launch
void launchStripKernel(hipStream_t s, const ContactView& c, const JointView& j, const BodyView& g, const GroupTable& gt, const Op* ops, int opCount,
					   const StepConsts& sc, s2amdContact* wire, int maxBodies, int useDq0)
{
	if (gt.groupCount <= 0 || opCount <= 0)
	{
		return;
	}
	size_t lds = (size_t)maxBodies * (useDq0 ? 56 : 40);
	groupKernel<S2_STRIP_THREADS, S2_STRIP_PRELOAD>
		<<<dim3((unsigned)gt.groupCount), dim3(S2_STRIP_THREADS), lds, s>>>(c, j, g, gt, ops, opCount, sc, wire, useDq0, 0);
}

void launchGroupKernel(hipStream_t s, const ContactView& c, const JointView& j, const BodyView& g, const GroupTable& gt, const Op* ops, int opCount,
					   const StepConsts& sc, s2amdContact* wire, int maxBodies, int useDq0)
{
	if (gt.groupCount <= 0 || opCount <= 0)
	{
		return;
	}
	size_t lds = (size_t)maxBodies * (useDq0 ? 56 : 40);
	// (the staged tables behind the bodies of the largest group: 16-byte aligned, and only where 160 KiB hold them too -- a whole-step
	// launch; the one-op launches of a sequential tail have one batch and one op to read)
	const size_t staged = (((size_t)maxBodies + 1) / 2 * 16 + (size_t)(useDq0 ? 3 : 2) * maxBodies * 16) + (size_t)S2_GROUP_STAGED_BATCHES * 16 + (size_t)S2_GROUP_STAGED_OPS * sizeof(Op);
	const int stage = opCount > 1 && staged <= 160 * 1024 ? 1 : 0;
	lds = stage ? std::max(lds, staged) : lds;
	groupKernel<S2_GROUP_THREADS, 0><<<dim3((unsigned)gt.groupCount), dim3(S2_GROUP_THREADS), lds, s>>>(c, j, g, gt, ops, opCount, sc, wire, useDq0, stage);
}

S2_DEFINE_WARM(group_kernel)
