// s2Solve_Jacobi (src/solve_jacobi.c:134-292) as ONE persistent launch: BASELINE.json configs[2] (Tumbler, 10k boxes).
//
// The contact pass of s2SolveContacts_Jacobi_Soft (solve_jacobi.c:21-132) writes no body: every constraint reads its two bodies'
// velocities as the last apply left them and adds ONE delta per body to body->dv / dw, in constraint order; the apply loop
// (:233-245) adds the sums to the velocities.  So the sweep has no colours and no order among constraints -- only the order of
// the additions per body, which is the pool order of that body's constraints.  Through round 4 this ran as three launches per
// iteration (joint sweep, contact pass, body-centric apply: 24 launches per step, 0.132 ms on the Tumbler).
//
// Here the bodies are dealt to BLOCKS (solver_jacobi.cpp: chunks of a breadth-first order of the constraint graph, ~190 bodies
// each), one 512-thread workgroup per block for the whole step:
//   * OWNER COMPUTES, with redundancy: a block holds EVERY constraint that touches a body it owns -- a constraint between bodies of
//     two blocks is held by both, computed by both from the same bits, and each block uses the delta of its own body only.  So an
//     iteration needs ONE exchange -- every block publishes the velocities of its bodies other blocks read, and reads theirs
//     (tagged 8-byte granules in global memory, persist_handoff.h; two parities) -- and no reduction across blocks, not even for a
//     hub: the Tumbler's drum is one body of one block, which holds all 238 of its constraints and imports their boxes.
//   * constraints are resident in registers for the whole step (SoftRegs<SOFT_JACOBI>, up to two per lane), velocities of the
//     block's bodies in LDS, the per-constraint deltas in LDS; a body adds the deltas of its incidence list in list order (pool
//     order: the reference's order), a body with a long list (the drum) is walked by one wave, 64 entries loaded at once, the
//     additions sequential on values broadcast lane by lane -- body_kernels.hip: jacobiApplyKernel's walk;
//   * s2WarmStartContacts (solve_common.c:276-326) the same way: per-point terms into LDS, added per body in list order;
//   * joints (sequential in the reference, solve_jacobi.c:211-221) by the lane 0 of the block that owns their bodies, on the
//     velocities in LDS (constraint_ops.h: solveJointsOne through JacobiJointBodies): the host admits a world whose every joint lies inside
//     one block (the Tumbler's motor joint: the drum and the static ground).
// Integer tables from the host (JacobiView); results equal the multi-launch path's and the oracle's bit for bit (same operations on
// the same operands in the same order).  Like the strip kernels: all workgroups must be co-resident, every poll loop is bounded,
// a hand-off that times out sets the error words, the epilogue launch then leaves the wire arrays alone and the host repeats the
// step on the multi-launch path.
#include "body_ops.h"
#include "persist_handoff.h"

#define S2_JACOBI_THREADS 512
#define S2_JACOBI_RECORDS 2 // constraints a lane holds: a block has at most 1024

namespace
{

#define S2_JACOBI_STAGE 128 // terms one wave stages per batch of a long list: 64 incidence entries, two points each

S2_DEV float laneOf(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// The sequential part of a long list's sum: acc += comp[0], += comp[1], ... += comp[n - 1], in that order (the reference's order: fp
// addition does not reassociate).  Lane c of the wave carries component c (x, y, w), so a term costs ONE dependent add and a quarter of
// a 16-byte LDS read; `n` is wave-uniform, `comp` 16-byte aligned.
S2_DEV float addInOrder(float acc, const float* comp, int n)
{
	int k = 0;
	for (; k + 8 <= n; k += 8)
	{
		const float4 a = *(const float4*)(comp + k), b = *(const float4*)(comp + k + 4);
		acc = acc + a.x, acc = acc + a.y, acc = acc + a.z, acc = acc + a.w;
		acc = acc + b.x, acc = acc + b.y, acc = acc + b.z, acc = acc + b.w;
	}
	for (; k < n; ++k)
	{
		acc = acc + comp[k];
	}
	return acc;
}

S2_DEV void waveLdsOrder()
{
	// LDS operations of one wave execute in order: this only keeps the compiler from moving them across
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The body accessor of a joint solved inside a block (constraint_ops.h: solveJointsOne): indices are pool slots (the joint SoA's); the
// velocities of the joint's two bodies live in the block's LDS where the block owns them, everything else -- the poses, a static body's
// velocity -- in the global arrays (poses only change in s2IntegratePositions, which this kernel performs on the global arrays)
struct JacobiJointBodies
{
	static constexpr int kMode = S2_IDX_GLOBAL;
	float4* lvel;
	float4* gvel;
	float4* gdq;
	int ga, la, gb, lb; // pool slot and local slot of the joint's owned bodies (-1: none)
	S2_DEV float4 getVel(int i) const { return i == ga ? lvel[la] : (i == gb ? lvel[lb] : gvel[i]); }
	S2_DEV void setVel(int i, float4 v) const
	{
		if (i == ga)
		{
			lvel[la] = v;
		}
		else if (i == gb)
		{
			lvel[lb] = v;
		}
		else
		{
			gvel[i] = v;
		}
	}
	S2_DEV float4 getDq(int i) const { return gdq[i]; }
	S2_DEV void setDq(int i, float4 v) const { gdq[i] = v; }
};

template <int RECORDS> __global__ __launch_bounds__(S2_JACOBI_THREADS) void jacobiStepKernel(ContactView c, JointView jv, BodyView g, JacobiView t, const Op* ops,
																							  int opCount, StepConsts sc)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const JacobiBlockDesc d = t.descs[blockIdx.x];
	const int nOwn = d.ownedCount, nImp = d.importCount, nC = d.constraintCount;
	float4* lvel = lds;							   // [nOwn + nImp]
	float4* lterm = lds + nOwn + nImp;			   // [4 * nC]: warm start {A p0, A p1, B p0, B p1}; iterations: [2 * nC] {deltaA, deltaB}
	int* llist = (int*)(lterm + 4 * nC);		   // [2 * nC] incidence entries (local constraint << 1 | side), per body in pool order
	int2* lrange = (int2*)(llist + 2 * nC + (nC & 1 ? 2 : 0)); // [nOwn] {first entry, entries}
	Op* lops = (Op*)(lrange + nOwn + (nOwn & 1));
	float* stage = (float*)(lops + opCount) + wave * 3 * S2_JACOBI_STAGE; // this wave's staging rows {x, y, w} of a long list's terms
	const int cLane = lane < 3 ? lane : 0;

	// ---- loads: tables, bodies, constraint records ----
	for (int i = tid; i < 2 * nC; i += S2_JACOBI_THREADS)
	{
		llist[i] = t.ints[d.listBase + i];
	}
	for (int i = tid; i < nOwn; i += S2_JACOBI_THREADS)
	{
		lrange[i] = make_int2(t.ints[d.rangeBase + 2 * i], t.ints[d.rangeBase + 2 * i + 1]);
	}
	for (int i = tid; i < opCount * 8; i += S2_JACOBI_THREADS)
	{
		((int*)lops)[i] = ((const int*)ops)[i];
	}
	for (int i = tid; i < nOwn + nImp; i += S2_JACOBI_THREADS)
	{
		const int gi = t.ints[(i < nOwn ? d.ownedBase : d.importBase - nOwn) + i];
		lvel[i] = g.vel[gi];
	}
	SoftRegs<SOFT_JACOBI> reg[RECORDS];
	int kOf[RECORDS]; // position in the sweep order (the SoA index); -1: none; bit 30: this block stores the impulses
#pragma unroll
	for (int r = 0; r < RECORDS; ++r)
	{
		const int e = tid + r * S2_JACOBI_THREADS;
		kOf[r] = -1;
		if (e < nC)
		{
			const int packed = t.ints[d.constraintBase + 3 * e];
			const int k = packed & 0x3fffffff;
			kOf[r] = packed;
			reg[r] = loadSoft<SOFT_JACOBI, S2_IDX_GLOBAL>(c, k);
			reg[r].h.ia = t.ints[d.constraintBase + 3 * e + 1];
			reg[r].h.ib = t.ints[d.constraintBase + 3 * e + 2];
		}
	}
	__syncthreads();

	LdsBodies lb{lvel, nullptr};
	ContactView cl = c; // the pass writes its deltas through c.deltaA / c.deltaB: here those are the LDS table
	cl.deltaA = lterm;
	cl.deltaB = lterm + nC;
	gu64* gran = (gu64*)t.granules;
	unsigned epoch = 0u;
	int bad = 0;

	// a body's sum over its incidence list, in list order; `heavy`: the bodies one wave walks (the host lists them)
	auto ownedGlobal = [&](int i) { return t.ints[d.ownedBase + i]; };

	for (int oi = 0; oi < opCount && !bad; ++oi)
	{
		const Op op = lops[oi];
		if (op.code == OP_INTEGRATE_VEL)
		{
			for (int i = tid; i < nOwn; i += S2_JACOBI_THREADS)
			{
				integrateVelocitiesOne(lb, i, g, ownedGlobal(i));
			}
			__syncthreads();
		}
		else if (op.code == OP_WARM)
		{
			// s2WarmStartContacts: per point and side {dv.x, dv.y, dw, valid}; solve_jacobi.c prepares and warm-starts at the poses the
			// anchors rA0 / rB0 were made at, so rotate(q, localAnchor) IS rA0 (the same operation on the same operands)
#pragma unroll
			for (int r = 0; r < RECORDS; ++r)
			{
				const int e = tid + r * S2_JACOBI_THREADS;
				if (e < nC)
				{
					const CHeader& h = reg[r].h;
					const V2 tangent = rightPerp(h.normal);
#pragma unroll
					for (int j = 0; j < 2; ++j)
					{
						const float4 arm = reg[r].r0[j];
						const V2 rA = v2(arm.x, arm.y), rB = v2(arm.z, arm.w);
						const V2 P = add(mulSV(reg[r].imp[j].x, h.normal), mulSV(reg[r].imp[j].y, tangent));
						const float valid = j < h.pointCount ? 1.0f : 0.0f;
						// wA -= iA * cross(rA, P); vA = mulAdd(vA, -mA, P); wB += iB * cross(rB, P); vB = mulAdd(vB, mB, P)
						const V2 pa = mulSV(-h.mA, P), pb = mulSV(h.mB, P);
						lterm[4 * e + j] = make_float4(pa.x, pa.y, -(h.iA * cross(rA, P)), valid);
						lterm[4 * e + 2 + j] = make_float4(pb.x, pb.y, h.iB * cross(rB, P), valid);
					}
				}
			}
			__syncthreads();
			for (int i = tid; i < nOwn; i += S2_JACOBI_THREADS)
			{
				const int2 range = lrange[i];
				if (range.y > 0 && range.y <= S2_JACOBI_HEAVY)
				{
					float4 v = lvel[i];
					for (int x = 0; x < range.y; ++x)
					{
						const int key = llist[range.x + x];
						const float4* term = lterm + 4 * (key >> 1) + 2 * (key & 1);
#pragma unroll
						for (int j = 0; j < 2; ++j)
						{
							const float4 q = term[j];
							if (q.w != 0.0f)
							{
								v.z = v.z + q.z;
								v.x = v.x + q.x, v.y = v.y + q.y;
							}
						}
					}
					lvel[i] = v;
				}
			}
			for (int hb = wave; hb < ((t.debugSkip & 1) ? 0 : d.heavyCount); hb += S2_JACOBI_THREADS / 64)
			{
				const int i = t.ints[d.heavyBase + hb];
				const int2 range = lrange[i];
				const float4 v = lvel[i];
				// 64 entries gathered at once, their valid points compacted into the wave's staging rows in list order (point 0 before
				// point 1), then added in that order
				float acc = cLane == 0 ? v.x : (cLane == 1 ? v.y : v.z);
				for (int base = 0; base < range.y; base += 64)
				{
					const bool in = base + lane < range.y;
					const int key = llist[range.x + (in ? base + lane : 0)];
					const float4 q0 = lterm[4 * (key >> 1) + 2 * (key & 1)], q1 = lterm[4 * (key >> 1) + 2 * (key & 1) + 1];
					const bool v0 = in && q0.w != 0.0f, v1 = in && q1.w != 0.0f;
					const unsigned long long b0 = __ballot(v0), b1 = __ballot(v1), lower = (1ull << lane) - 1ull;
					const int p0 = __popcll(b0 & lower) + __popcll(b1 & lower), p1 = p0 + (v0 ? 1 : 0);
					if (v0)
					{
						stage[p0] = q0.x, stage[S2_JACOBI_STAGE + p0] = q0.y, stage[2 * S2_JACOBI_STAGE + p0] = q0.z;
					}
					if (v1)
					{
						stage[p1] = q1.x, stage[S2_JACOBI_STAGE + p1] = q1.y, stage[2 * S2_JACOBI_STAGE + p1] = q1.z;
					}
					const int n = __builtin_amdgcn_readfirstlane(__popcll(b0) + __popcll(b1));
					waveLdsOrder();
					acc = addInOrder(acc, stage + cLane * S2_JACOBI_STAGE, n);
					waveLdsOrder();
				}
				const float ax = laneOf(acc, 0), ay = laneOf(acc, 1), az = laneOf(acc, 2);
				if (lane == 0)
				{
					lvel[i] = make_float4(ax, ay, az, v.w);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_JOINT_SWEEP)
		{
			// the block's joints, one after the other in sweep order, by one lane: velocities in LDS, poses in the global arrays
			if (d.jointCount > 0 && tid == 0 && (t.debugSkip & 2) == 0)
			{
				for (int x = 0; x < d.jointCount; ++x)
				{
					const int k = t.ints[d.jointBase + 3 * x];
					const int la = t.ints[d.jointBase + 3 * x + 1], lbx = t.ints[d.jointBase + 3 * x + 2]; // local slots of its owned bodies, -1: none
					const JacobiJointBodies jb{lvel, g.vel, g.dq, la >= 0 ? ownedGlobal(la) : -1, la, lbx >= 0 ? ownedGlobal(lbx) : -1, lbx};
					if (op.kind == JSOLVE_WARM)
					{
						solveJointsOne<JSOLVE_WARM>(jv, jb, sc, op.h, op.inv_h, op.useBias, k);
					}
					else
					{
						solveJointsOne<JSOLVE_SOFT>(jv, jb, sc, op.h, op.inv_h, op.useBias, k);
					}
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_SOLVE_SOFT)
		{
			// ---- the iteration's exchange: the velocities other blocks read ----
			epoch += 1;
			const int par = (int)(epoch & 1u) * t.parityStride;
			for (int x = tid; x < ((t.debugSkip & 4) ? 0 : d.exportCount); x += S2_JACOBI_THREADS)
			{
				const int i = t.ints[d.exportBase + x];
				const float4 v = lvel[i];
				gu64* p = gran + par + 4 * (size_t)ownedGlobal(i);
				putGranule(p + 0, epoch, v.x), putGranule(p + 1, epoch, v.y), putGranule(p + 2, epoch, v.z);
			}
			int fail = 0;
			for (int x = tid; x < ((t.debugSkip & 4) ? 0 : nImp); x += S2_JACOBI_THREADS)
			{
				const int gi = t.ints[d.importBase + x];
				if (gi >= 0 && (t.ints[d.importBase + nImp + x] & 1) != 0) // (a body somebody owns: the others never change)
				{
					float v[3];
					if (getGranules<3>(gran + par + 4 * (size_t)gi, epoch, v, t.error, t.deviceError, t.spinLimit))
					{
						lvel[nOwn + x] = make_float4(v[0], v[1], v[2], 0.0f);
					}
					else
					{
						fail = 1;
					}
				}
			}
			bad = __syncthreads_or(fail);
			if (bad)
			{
				break;
			}
			// ---- s2SolveContacts_Jacobi_Soft: every constraint from the velocities as they stand, deltas into LDS ----
#pragma unroll
			for (int r = 0; r < RECORDS; ++r)
			{
				const int e = tid + r * S2_JACOBI_THREADS;
				if (e < nC && (t.debugSkip & 8) == 0)
				{
					solveSoftRegs<SOFT_JACOBI, LdsBodies, false>(reg[r], cl, lb, op.inv_h, op.useBias, e);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_JACOBI_APPLY)
		{
			// solve_jacobi.c:233-245: v += dv, dv the sum of the body's deltas in constraint order (from zero, like body->dv)
			for (int i = tid; i < nOwn; i += S2_JACOBI_THREADS)
			{
				const int2 range = lrange[i];
				if (range.y > 0 && range.y <= S2_JACOBI_HEAVY)
				{
					V2 dv = v2(0.0f, 0.0f);
					float dw = 0.0f;
					for (int x = 0; x < range.y; ++x)
					{
						const int key = llist[range.x + x];
						const float4 q = lterm[(key & 1) * nC + (key >> 1)];
						dv = add(dv, v2(q.x, q.y));
						dw += q.z;
					}
					const float4 v = lvel[i];
					const V2 lv = add(v2(v.x, v.y), dv);
					lvel[i] = make_float4(lv.x, lv.y, v.z + dw, 0.0f);
				}
			}
			for (int hb = wave; hb < ((t.debugSkip & 1) ? 0 : d.heavyCount); hb += S2_JACOBI_THREADS / 64)
			{
				const int i = t.ints[d.heavyBase + hb];
				const int2 range = lrange[i];
				float acc = 0.0f;
				for (int base = 0; base < range.y; base += 64)
				{
					const bool in = base + lane < range.y;
					const int key = llist[range.x + (in ? base + lane : 0)];
					const float4 q = lterm[(key & 1) * nC + (key >> 1)];
					if (in)
					{
						stage[lane] = q.x, stage[S2_JACOBI_STAGE + lane] = q.y, stage[2 * S2_JACOBI_STAGE + lane] = q.z;
					}
					const int left = __builtin_amdgcn_readfirstlane(range.y - base);
					waveLdsOrder();
					acc = addInOrder(acc, stage + cLane * S2_JACOBI_STAGE, left < 64 ? left : 64);
					waveLdsOrder();
				}
				const float dx = laneOf(acc, 0), dy = laneOf(acc, 1), dw = laneOf(acc, 2);
				if (lane == 0)
				{
					const float4 v = lvel[i];
					const V2 lv = add(v2(v.x, v.y), v2(dx, dy));
					lvel[i] = make_float4(lv.x, lv.y, v.z + dw, 0.0f);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_INTEGRATE_POS)
		{
			// s2IntegratePositions (solve_common.c:47-68): the poses stay in the global arrays (no sweep of this driver reads them)
			for (int i = tid; i < nOwn; i += S2_JACOBI_THREADS)
			{
				const int gi = ownedGlobal(i);
				if ((g.flags[gi] & S2F_MOVES) != 0)
				{
					const float4 v = lvel[i], q4 = g.dq[gi];
					const V2 dp = mulAdd(v2(q4.x, q4.y), op.h, v2(v.x, v.y));
					Rot q;
					q.s = q4.z, q.c = q4.w;
					q = integrateRot(q, op.h * v.z);
					g.dq[gi] = make_float4(dp.x, dp.y, q.s, q.c);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_FINALIZE)
		{
			const GlobalBodies gb{g.vel, g.dq};
			for (int i = tid; i < nOwn; i += S2_JACOBI_THREADS)
			{
				const int gi = ownedGlobal(i);
				finalizePositionsOne(gb, gi, g, gi, op.flag, true);
			}
			__syncthreads();
		}
	}

	// ---- results: owned velocities, impulses (a constraint two blocks hold is stored by one of them: the same bits either way) ----
	for (int i = tid; i < nOwn; i += S2_JACOBI_THREADS)
	{
		g.vel[ownedGlobal(i)] = lvel[i];
	}
#pragma unroll
	for (int r = 0; r < RECORDS; ++r)
	{
		const int e = tid + r * S2_JACOBI_THREADS;
		if (e < nC && (kOf[r] & 0x40000000) != 0)
		{
			const int k = kOf[r] & 0x3fffffff;
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				if (j < reg[r].h.pointCount)
				{
					c.impulse[j][k] = reg[r].imp[j];
				}
			}
		}
	}
}

} // namespace

size_t jacobiStepLds(int owned, int imports, int constraints, int opCount)
{
	size_t bytes = (size_t)(owned + imports) * sizeof(float4) + (size_t)4 * constraints * sizeof(float4);
	bytes += (size_t)(2 * constraints + 2) * sizeof(int) + (size_t)(owned + 1) * sizeof(int2) + (size_t)opCount * sizeof(Op) + 64;
	bytes += (size_t)(S2_JACOBI_THREADS / 64) * 3 * S2_JACOBI_STAGE * sizeof(float);
	return bytes;
}

int jacobiKernelSetup()
{
	for (const void* f : {(const void*)jacobiStepKernel<1>, (const void*)jacobiStepKernel<2>})
	{
		if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
		{
			return 1;
		}
	}
	return 0;
}

void launchJacobiStep(hipStream_t s, const ContactView& c, const JointView& jv, const BodyView& g, const JacobiView& t, const Op* ops, int opCount, const StepConsts& sc,
					  size_t ldsBytes, int maxConstraints)
{
	const dim3 grid((unsigned)t.blockCount), block(S2_JACOBI_THREADS);
	if (maxConstraints <= S2_JACOBI_THREADS)
	{
		jacobiStepKernel<1><<<grid, block, ldsBytes, s>>>(c, jv, g, t, ops, opCount, sc);
	}
	else
	{
		jacobiStepKernel<2><<<grid, block, ldsBytes, s>>>(c, jv, g, t, ops, opCount, sc);
	}
}

S2_DEFINE_WARM(jacobi_kernel)
