// Strip kernel: the lean launch of the strip path (solver_structure.cpp: partitionStrips) for the soft contact
// sweeps -- s2SolveContacts_TGS_Soft / _PGS_Soft / _TGS_Fixed, plus the body stages and the contact warm
// start that sit between two sweeps in their drivers.  One workgroup (256 threads = one wave per SIMD,
// the whole register file) per strip (phase A) or seam (phase B).
//
// Why a dedicated kernel and not group_kernel.hip's interpreter: a strip launch lives for a few
// microseconds, so what matters is the number of DEPENDENT memory round trips and the instruction
// footprint.  Here there are three trips -- (1) the group's 128-byte descriptor (scalar), (2) body ids +
// the constraint records of all colour rounds of this thread (everything in flight at once), (3) the body
// records -- and the colour rounds then run from registers and LDS only, separated by s_barrier.
//
// Arithmetic: constraint_ops.h / body_ops.h, i.e. the same functions as every other path; the sweep
// order (round-major inside the group) is the one reported by s2amd_get_contact_order.

#include "body_ops.h"

#define S2_STRIP_THREADS 256

template <int KIND, int WARM>
__global__ __launch_bounds__(S2_STRIP_THREADS) void stripSoftKernel(ContactView c, BodyView g, const StripDesc* descs, const int* bodyIds,
																	 const int2* slots, const int* slotOffsets, StripOps ops)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	const StripDesc* dp = descs + blockIdx.x;
	const int bodyBase = dp->bodyBase, nb = dp->bodyCount, rounds = dp->batchCount;
	const int slotBase = dp->slotBase, slotCount = dp->slotCount, slotOffBase = dp->slotOffBase;
	int4 batch[S2_STRIP_ROUNDS];
#pragma unroll
	for (int i = 0; i < S2_STRIP_ROUNDS; ++i)
	{
		batch[i] = dp->batch[i];
	}
	float4* lvel = lds;
	float4* ldq = lds + nb;
	float4* lterm = lds + 2 * nb; // warm start: two records per incident (constraint, side) slot
	const int tid = (int)threadIdx.x;

	// ---- trip 2: body ids, then every constraint record this thread will need ----
	uint32_t id[S2_STRIP_BODY_CHUNKS];
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		int i = tid + ch * S2_STRIP_THREADS;
		id[ch] = i < nb ? (uint32_t)bodyIds[bodyBase + i] : 0u;
	}
	SoftRegs<KIND> r[S2_STRIP_ROUNDS];
	int kk[S2_STRIP_ROUNDS];
#pragma unroll
	for (int i = 0; i < S2_STRIP_ROUNDS; ++i)
	{
		kk[i] = -1;
		if (ops.sweep && i < rounds)
		{
			int k = batch[i].x + tid;
			if (k < batch[i].y)
			{
				kk[i] = k;
				r[i] = loadSoft<KIND, S2_IDX_LOCAL>(c, k);
			}
		}
	}

	// ---- trip 3: body records (+ the integrator constants when a body stage rides along) ----
	float4 vel[S2_STRIP_BODY_CHUNKS], dq[S2_STRIP_BODY_CHUNKS], integ[S2_STRIP_BODY_CHUNKS];
	float angDamp[S2_STRIP_BODY_CHUNKS];
	uint32_t flags[S2_STRIP_BODY_CHUNKS];
	const bool bodyStage = ops.integrateVel || ops.integratePos;
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		int i = tid + ch * S2_STRIP_THREADS;
		if (i < nb)
		{
			int gi = (int)(id[ch] & ~S2G_OWNED);
			vel[ch] = g.vel[gi];
			dq[ch] = g.dq[gi];
			if (bodyStage)
			{
				flags[ch] = g.flags[gi];
				if (ops.integrateVel)
				{
					integ[ch] = g.integ[gi];
					angDamp[ch] = g.angDamp[gi];
				}
			}
		}
	}
	// body stages in registers, exactly integratePositionsOne / integrateVelocitiesOne (body_ops.h)
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		int i = tid + ch * S2_STRIP_THREADS;
		if (i < nb)
		{
			if (ops.integratePos && (flags[ch] & S2F_MOVES) != 0)
			{
				float4 v = vel[ch], d = dq[ch];
				V2 dpos = mulAdd(v2(d.x, d.y), ops.posH, v2(v.x, v.y));
				Rot q;
				q.s = d.z, q.c = d.w;
				q = integrateRot(q, ops.posH * v.z);
				dq[ch] = make_float4(dpos.x, dpos.y, q.s, q.c);
			}
			if (ops.integrateVel && (flags[ch] & S2F_DYNAMIC) != 0)
			{
				float4 v = vel[ch], k = integ[ch];
				V2 lv = add(v2(v.x, v.y), v2(k.x, k.y));
				float w = v.z + k.z;
				lv = mulSV(k.w, lv);
				w *= angDamp[ch];
				vel[ch] = make_float4(lv.x, lv.y, w, 0.0f);
			}
			ldq[i] = dq[ch];
			if (WARM < 0)
			{
				lvel[i] = vel[ch];
			}
		}
	}

	if (WARM >= 0)
	{
		// s2WarmStartContacts (solve_common.c:276), body-centric: the warm start adds velocity-independent
		// terms, so every (constraint, side) slot computes its two terms in parallel and each body then adds
		// its own in sweep order -- the same bits as the coloured sweep (see warmStartBodiesKernel)
		__syncthreads(); // poses in LDS
		for (int s = tid; s < slotCount; s += S2_STRIP_THREADS)
		{
			int2 sl = slots[slotBase + s];
			int k = sl.x >> 1;
			bool sideB = (sl.x & 1) != 0;
			float4 nf = c.nf[k];
			float4 ms = c.mass[k];
			float4 arm[2];
			float2 imp[2];
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				arm[j] = WARM == WARM_CURRENT ? c.anchor[j][k] : c.r0[j][k];
				imp[j] = c.impulse[j][k];
			}
			V2 normal = v2(nf.x, nf.y);
			V2 tangent = rightPerp(normal);
			int pointCount = (int)(asBits(nf.w) & 0xffu);
			float m = sideB ? ms.z : ms.x;
			float iv = sideB ? ms.w : ms.y;
			Rot q;
			if (WARM == WARM_CURRENT)
			{
				float4 d = ldq[sl.y];
				q.s = d.z, q.c = d.w;
			}
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				V2 l = sideB ? v2(arm[j].z, arm[j].w) : v2(arm[j].x, arm[j].y);
				V2 rr = WARM == WARM_CURRENT ? rotate(q, l) : l;
				V2 P = add(mulSV(imp[j].x, normal), mulSV(imp[j].y, tangent));
				float tw = iv * cross(rr, P);
				float sm = sideB ? m : -m;
				lterm[2 * s + j] = make_float4(sm * P.x, sm * P.y, sideB ? tw : -tw, j < pointCount ? 1.0f : 0.0f);
			}
		}
		__syncthreads();
#pragma unroll
		for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
		{
			int i = tid + ch * S2_STRIP_THREADS;
			if (i < nb)
			{
				if (i < dp->ownedCount)
				{
					int e0 = slotOffsets[slotOffBase + i], e1 = slotOffsets[slotOffBase + i + 1];
					V2 v = v2(vel[ch].x, vel[ch].y);
					float w = vel[ch].z;
					for (int e = e0; e < e1; ++e)
					{
#pragma unroll
						for (int j = 0; j < 2; ++j)
						{
							float4 t = lterm[2 * e + j];
							if (t.w != 0.0f)
							{
								v = v2(v.x + t.x, v.y + t.y);
								w = w + t.z;
							}
						}
					}
					if (e1 > e0)
					{
						vel[ch] = make_float4(v.x, v.y, w, 0.0f);
					}
				}
				lvel[i] = vel[ch];
			}
		}
	}
	__syncthreads();

	// ---- colour rounds: registers + LDS ----
	if (ops.sweep)
	{
		LdsBodies lb{lvel, ldq};
#pragma unroll
		for (int i = 0; i < S2_STRIP_ROUNDS; ++i)
		{
			if (i < rounds)
			{
				if (kk[i] >= 0)
				{
					solveSoftRegs<KIND>(r[i], c, lb, ops.inv_h, ops.useBias, kk[i]);
					storeSoft<KIND>(c, r[i], kk[i]);
				}
				// a batch wider than the workgroup: the rest streams
				for (int k = batch[i].x + tid + S2_STRIP_THREADS; k < batch[i].y; k += S2_STRIP_THREADS)
				{
					solveContactsSoftOne<KIND>(c, lb, ops.inv_h, ops.useBias, k);
				}
				__syncthreads();
			}
		}
	}

	// ---- write back what this group owns ----
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		int i = tid + ch * S2_STRIP_THREADS;
		if (i < nb && (id[ch] & S2G_OWNED) != 0)
		{
			int gi = (int)(id[ch] & ~S2G_OWNED);
			g.vel[gi] = lvel[i];
			if (ops.integratePos)
			{
				g.dq[gi] = ldq[i];
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Persistent strip step: ONE launch per s2Solve_* call.  Workgroup i keeps strip i's interior constraints
// AND the constraints of seam i | i+1 in registers for the whole step, its bodies in LDS, and walks the
// step plan.  What a colour-batch launch costs the other paths (reloading every constraint record
// through one CU's memory pipe, ~25-60 GB/s) is paid once per step here.
//
// Per sweep:  A rounds (own bodies)  ->  the bodies of strip i+1 that seam i touches travel i+1 -> i
//             B rounds (seam i)      ->  they travel back i -> i+1
// as 8-byte {epoch, value} granules written with ONE agent-scope (sc1, write-through) store each and
// polled with agent-scope loads: the data is the flag, no fence (cdna_hip_programming.md G16, form R2).
// Every buffer is used strictly ping-pong (a sender can only reach epoch e+1 after it has consumed what
// the receiver produced from epoch e), so one slot per granule suffices.  All K <= CU-count workgroups
// are resident (one per CU); every poll loop is bounded and reports through pv.error.
// ------------------------------------------------------------------------------------------------
// 1: the timing-only switches of `persist_debug` (bits 1, 2, 4) and the S2AMD_DEBUG_TIMES stamps are compiled in
#ifndef S2_PERSIST_INSTRUMENTED
#define S2_PERSIST_INSTRUMENTED 0
#endif
#include "persist_handoff.h"

// s2WarmStartContacts (solve_common.c:276-330) for one constraint held in registers: warmStartContactsOne's
// arithmetic (constraint_ops.h) on SoftRegs
template <int WARM, int KIND, class BA> S2_DEV void warmSoftRegs(const SoftRegs<KIND>& r, const BA& b)
{
	const CHeader& h = r.h;
	V2 tangent = rightPerp(h.normal);
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	Rot qA, qB;
	if (WARM == WARM_CURRENT)
	{
		qA = loadPose(b, h.ia).q;
		qB = loadPose(b, h.ib).q;
	}
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 arm = WARM == WARM_CURRENT ? r.an[j] : r.r0[j];
			V2 rA, rB;
			if (WARM == WARM_CURRENT)
			{
				rA = rotate(qA, v2(arm.x, arm.y));
				rB = rotate(qB, v2(arm.z, arm.w));
			}
			else
			{
				rA = v2(arm.x, arm.y);
				rB = v2(arm.z, arm.w);
			}
			V2 P = add(mulSV(r.imp[j].x, h.normal), mulSV(r.imp[j].y, tangent));
			wA -= h.iA * cross(rA, P);
			vA = mulAdd(vA, -h.mA, P);
			wB += h.iB * cross(rB, P);
			vB = mulAdd(vB, h.mB, P);
		}
	}
	if (h.writeA)
	{
		storeVel(b, h.ia, vA, wA);
	}
	if (h.writeB)
	{
		storeVel(b, h.ib, vB, wB);
	}
}

// A constraint as the persistent kernel keeps it for the whole step: only what the sweeps of this solver read and what is
// not the same for many constraints -- 22 registers for TGS_Soft.  Not kept: the inverse masses (they sit once per body
// in LDS beside the body records) and the three soft coefficients (contact_kernels.hip prepareContactsKernel<PREP_SOFT>;
// solve_common.c:219, 262-271: one of two step-wide triples, chosen by whether a side is static; PersistView.softCoef).
struct ImpulsePart
{
	float2 imp[2]; // first: one 16-byte record of the LDS copy (the only part a sweep changes)
};
template <int TAG, bool ON> struct ArmsPart
{
	float4 v[2];
};
template <int TAG> struct ArmsPart<TAG, false>
{
};
template <bool ON> struct SeparationPart
{
	float p3[2]; // par.w (the prepare-time separation) is read by the PGS_Soft sweep only
};
template <> struct SeparationPart<false>
{
};
template <int KIND, int WARM>
struct PersistRegs : ImpulsePart,
					 ArmsPart<0, KIND == SOFT_TGS || KIND == SOFT_FIXED || WARM == WARM_CURRENT>, // local anchors
					 ArmsPart<1, KIND != SOFT_TGS || WARM == WARM_FIXED>,						  // prepare-time arms rA0 / rB0
					 SeparationPart<KIND == SOFT_PGS>
{
	static constexpr bool kAnchors = KIND == SOFT_TGS || KIND == SOFT_FIXED || WARM == WARM_CURRENT;
	static constexpr bool kArms0 = KIND != SOFT_TGS || WARM == WARM_FIXED;
	uint32_t idx; // ia | ib << 14 | pointCount << 28 | writeA << 30 | writeB << 31
	float nx, ny, friction;
	float p0[2], p1[2], p2[2];
};
static_assert(sizeof(PersistRegs<SOFT_TGS, WARM_CURRENT>) <= 96, "6 LDS records per seam constraint (solver_structure.cpp: S2_PERSIST_Q_NARROW)");
static_assert(sizeof(PersistRegs<SOFT_PGS, WARM_CURRENT>) <= 128 && sizeof(PersistRegs<SOFT_FIXED, WARM_FIXED>) <= 128 &&
				  sizeof(PersistRegs<SOFT_TGS, WARM_FIXED>) <= 128 && sizeof(PersistRegs<SOFT_PGS, WARM_FIXED>) <= 128 &&
				  sizeof(PersistRegs<SOFT_FIXED, WARM_CURRENT>) <= 128,
			  "8 LDS records per seam constraint for every other kind (S2_PERSIST_Q_WIDE)");

// the per-body inverse masses and the two coefficient triples a resident constraint is completed from
struct PersistShared
{
	const float2* massInv; // LDS, one per staged body
	float4 softCoef[2];	   // [0] dynamic-dynamic, [1] one side static
};

template <int KIND, int WARM> S2_DEV PersistRegs<KIND, WARM> packPersist(const SoftRegs<KIND>& r, int ia, int ib)
{
	PersistRegs<KIND, WARM> p;
	p.idx = (uint32_t)ia | ((uint32_t)ib << 14) | ((uint32_t)r.h.pointCount << 28) | (r.h.writeA ? 1u << 30 : 0u) | (r.h.writeB ? 1u << 31 : 0u);
	p.nx = r.h.normal.x, p.ny = r.h.normal.y, p.friction = r.h.friction;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if constexpr (PersistRegs<KIND, WARM>::kAnchors)
		{
			static_cast<ArmsPart<0, true>&>(p).v[j] = r.an[j];
		}
		if constexpr (PersistRegs<KIND, WARM>::kArms0)
		{
			static_cast<ArmsPart<1, true>&>(p).v[j] = r.r0[j];
		}
		p.p0[j] = r.par[j].x, p.p1[j] = r.par[j].y, p.p2[j] = r.par[j].z;
		if constexpr (KIND == SOFT_PGS)
		{
			static_cast<SeparationPart<true>&>(p).p3[j] = r.par[j].w;
		}
		p.imp[j] = r.imp[j];
	}
	return p;
}

template <int KIND, int WARM> S2_DEV SoftRegs<KIND> unpackPersist(const PersistRegs<KIND, WARM>& p, const PersistShared& sh, uint32_t salt = 0u)
{
	SoftRegs<KIND> r;
	// `salt` is an opaque zero produced inside the step loop: without it the compiler hoists the decoding of every
	// round's indices (and the LDS addresses made from them) out of that loop and pays for it in scratch spills
	const uint32_t idx = p.idx ^ salt;
	r.h.ia = (int)(idx & 0x3fffu), r.h.ib = (int)((idx >> 14) & 0x3fffu);
	r.h.pointCount = (int)((idx >> 28) & 3u);
	r.h.writeA = (idx & (1u << 30)) != 0, r.h.writeB = (idx & (1u << 31)) != 0;
	const float2 massA = sh.massInv[r.h.ia], massB = sh.massInv[r.h.ib];
	r.h.mA = massA.x, r.h.iA = massA.y, r.h.mB = massB.x, r.h.iB = massB.y;
	const float4 coef = (massA.x == 0.0f || massB.x == 0.0f) ? sh.softCoef[1] : sh.softCoef[0];
	// the same for the values whose negations the sweep uses (tangent = (ny, -nx), -normalMass, -tangentMass)
	r.h.normal = v2(fromBits(asBits(p.nx) ^ salt), p.ny), r.h.friction = p.friction;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if constexpr (PersistRegs<KIND, WARM>::kAnchors)
		{
			r.an[j] = static_cast<const ArmsPart<0, true>&>(p).v[j];
		}
		if constexpr (PersistRegs<KIND, WARM>::kArms0)
		{
			r.r0[j] = static_cast<const ArmsPart<1, true>&>(p).v[j];
		}
		float separation = 0.0f;
		if constexpr (KIND == SOFT_PGS)
		{
			separation = static_cast<const SeparationPart<true>&>(p).p3[j];
		}
		r.par[j] = make_float4(p.p0[j], fromBits(asBits(p.p1[j]) ^ salt), fromBits(asBits(p.p2[j]) ^ salt), separation);
		r.sf[j] = coef;
		r.imp[j] = p.imp[j];
	}
	return r;
}

// one constraint of a sweep, from its resident registers: warm start or soft solve
template <int KIND, int WARM, int POINTS, class BA>
S2_DEV void sweepPersist(PersistRegs<KIND, WARM>& p, const PersistShared& sh, const ContactView& c, const BA& lb, float inv_h, int useBias, int k,
						 uint32_t salt = 0u)
{
	SoftRegs<KIND> r = unpackPersist<KIND, WARM>(p, sh, salt);
	solveSoftRegs<KIND, BA, false, POINTS>(r, c, lb, inv_h, useBias, k);
	p.imp[0] = r.imp[0], p.imp[1] = r.imp[1];
}

template <int KIND, int WARM> S2_DEV SoftRegs<KIND> loadPersist(const ContactView& c, int k)
{
	SoftRegs<KIND> r = loadSoft<KIND, S2_IDX_LOCAL>(c, k);
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (WARM == WARM_CURRENT && KIND == SOFT_PGS)
		{
			r.an[j] = c.anchor[j][k];
		}
		if (WARM == WARM_FIXED && KIND == SOFT_TGS)
		{
			r.r0[j] = c.r0[j][k];
		}
	}
	return r;
}

// POINTS == 2: the host has checked that every constraint of the strips has two manifold points (box stacks): the
// sweeps then run without per-point exec masking; POINTS == 0: general.
// ROUNDS: interior colour batches kept in registers (6, or 8 for strips whose colouring needs more).
// SEAMREG: no seam has more than two colour batches, and the seam constraints (two rounds x two passes = four per
// thread) stay in registers like the interior ones instead of in LDS records.
template <int KIND, int WARM, int POINTS, int ROUNDS, int SEAMREG>
__global__ __launch_bounds__(S2_STRIP_THREADS) void stripStepKernel(ContactView c, BodyView g, StripTableView ta, PersistView pv, const Op* ops, int opCount)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	const int tid = (int)threadIdx.x;
	const bool stamp = S2_PERSIST_INSTRUMENTED && pv.debugTimes != nullptr && blockIdx.x == gridDim.x / 2 && tid == 0;
	int stamps = 0;
	if (stamp)
	{
		pv.debugTimes[stamps++] = wall_clock64();
	}
	const StripDesc* da = ta.descs + blockIdx.x;
	const PersistDesc* pd = pv.descs + blockIdx.x;
	const int bodyBase = da->bodyBase, nb = da->bodyCount, roundsA = da->batchCount;
	int4 batchA[ROUNDS];
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		batchA[i] = da->batch[i];
	}
	const int nImp0 = pd->importCount[0], nImp1 = pd->importCount[1];
	const int nExp0 = pd->exportCount[0], nExp1 = pd->exportCount[1];
	const int roundsB0 = pd->seamBatchCount[0], roundsB1 = pd->seamBatchCount[1];
	int2 batchB0[S2_PERSIST_B_ROUNDS], batchB1[S2_PERSIST_B_ROUNDS];
	int seamSlots0 = 0, seamSlots1 = 0;
#pragma unroll
	for (int i = 0; i < S2_PERSIST_B_ROUNDS; ++i)
	{
		batchB0[i] = pd->seamBatch[0][i];
		batchB1[i] = pd->seamBatch[1][i];
		seamSlots0 += i < roundsB0 ? batchB0[i].y - batchB0[i].x : 0;
		seamSlots1 += i < roundsB1 ? batchB1[i].y - batchB1[i].x : 0;
	}
	const int roundsB = roundsB0 > roundsB1 ? roundsB0 : roundsB1;
	const int seamSlots = seamSlots0 + seamSlots1;
	const int firstK0 = batchB0[0].x, firstK1 = batchB1[0].x;
	const int nt = nb + nImp0 + nImp1;
	gu64* gran = (gu64*)pv.granules;
	const int in0 = pd->inBase[0], in1 = pd->inBase[1], out0 = pd->outBase[0], out1 = pd->outBase[1];

	constexpr int Q = (int)((sizeof(PersistRegs<KIND, WARM>) + 15) / 16);
	float4* lvel = lds;
	float4* ldq = lds + nt;
	float4* linteg = lds + 2 * nt;				  // velocity-integrator constants of every staged body (body_ops.h)
	float* langDamp = (float*)(lds + 3 * nt);	  // nt floats, padded to records
	float2* lmass = (float2*)(lds + 3 * nt + (nt + 3) / 4); // {invMass, invI} of every staged body, padded to records
	const int bodyRecords = 3 * nt + (nt + 3) / 4 + (nt + 1) / 2;
	Op* lops = (Op*)(lds + bodyRecords);		  // 2 records per op
	float4* lseam = lds + bodyRecords + 2 * opCount; // field-major: Q records per seam constraint

	// ---- loads ----
	uint32_t id[S2_STRIP_BODY_CHUNKS];
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		int i = tid + ch * S2_STRIP_THREADS;
		id[ch] = i < nb ? (uint32_t)ta.bodyIds[bodyBase + i] : 0u;
	}
	// imported bodies: thread t keeps left import t and right import t (at most 256 per side)
	int impId[2], expIdx[2];
	impId[0] = tid < nImp0 ? pv.importIds[pd->importIdBase[0] + tid] : -1;
	impId[1] = tid < nImp1 ? pv.importIds[pd->importIdBase[1] + tid] : -1;
	expIdx[0] = tid < nExp0 ? pv.exportSrc[pd->exportSrcBase[0] + tid] : 0;
	expIdx[1] = tid < nExp1 ? pv.exportSrc[pd->exportSrcBase[1] + tid] : 0;
	for (int i = tid; i < opCount * 8; i += S2_STRIP_THREADS)
	{
		((int*)lops)[i] = ((const int*)ops)[i];
	}
	PersistRegs<KIND, WARM> rA[ROUNDS];
	// the constraint a thread holds in round i is recomputed where needed (batch ranges sit in scalar registers)
	auto kOfRound = [&](int i) {
		int k = batchA[i].x + tid;
		return (i < roundsA && k < batchA[i].y) ? k : -1;
	};
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		if (i < roundsA)
		{
			int k = batchA[i].x + tid;
			if (k < batchA[i].y)
			{
				SoftRegs<KIND> t = loadPersist<KIND, WARM>(c, k);
				rA[i] = packPersist<KIND, WARM>(t, t.h.ia, t.h.ib);
			}
		}
	}
	// seam constraints: round r = left seam's batch r followed by right seam's batch r, dealt to the threads in
	// two passes (a round holds at most 512 constraints); where an item lives is recomputed, not stored
	// `salt`: the opaque zero of the step loop (see unpackPersist) -- keeps these few integer operations inside the loop
	// instead of 24 hoisted registers
	auto seamItem = [&](int r, int pass, int& side, int& k, int& slot, uint32_t salt = 0u) {
		const int n0 = r < roundsB0 ? batchB0[r].y - batchB0[r].x : 0;
		const int n1 = r < roundsB1 ? batchB1[r].y - batchB1[r].x : 0;
		const int idx = (int)((uint32_t)(tid + pass * S2_STRIP_THREADS) ^ salt);
		if (idx < n0)
		{
			side = 0, k = batchB0[r].x + idx, slot = k - firstK0;
			return true;
		}
		if (idx - n0 < n1)
		{
			side = 1, k = batchB1[r].x + idx - n0, slot = seamSlots0 + k - firstK1;
			return true;
		}
		return false;
	};
	constexpr int SEAM_REG_ROUNDS = 2;
	PersistRegs<KIND, WARM> rB[SEAMREG ? 2 * SEAM_REG_ROUNDS : 1];
	uint32_t seamMask = 0u; // SEAMREG: bit 2 i + pass set when this thread holds a seam constraint in that round and pass
#pragma unroll
	for (int i = 0; i < (SEAMREG ? SEAM_REG_ROUNDS : S2_PERSIST_B_ROUNDS); ++i)
	{
#pragma unroll
		for (int pass = 0; pass < 2; ++pass)
		{
			int side, k, slot;
			if (i < roundsB && seamItem(i, pass, side, k, slot))
			{
				SoftRegs<KIND> t = loadPersist<KIND, WARM>(c, k);
				PersistRegs<KIND, WARM> pb = packPersist<KIND, WARM>(t, pv.remap[pd->remapBase[side] + t.h.ia], pv.remap[pd->remapBase[side] + t.h.ib]);
				if constexpr (SEAMREG)
				{
					rB[2 * i + pass] = pb;
					seamMask |= 1u << (2 * i + pass);
				}
				else
				{
					float4 q[Q];
					__builtin_memcpy(q, &pb, sizeof(pb));
#pragma unroll
					for (int f = 0; f < Q; ++f)
					{
						lseam[f * seamSlots + slot] = q[f];
					}
				}
			}
		}
	}
	// bodies (+ their integrator constants) into LDS: own list, then both imports
	uint32_t flags[S2_STRIP_BODY_CHUNKS + 2];
	int ldsIdx[S2_STRIP_BODY_CHUNKS + 2];
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS + 2; ++ch)
	{
		int gi = -1;
		if (ch < S2_STRIP_BODY_CHUNKS)
		{
			int i = tid + ch * S2_STRIP_THREADS;
			ldsIdx[ch] = i;
			gi = i < nb ? (int)(id[ch] & ~S2G_OWNED) : -1;
		}
		else
		{
			const int side = ch - S2_STRIP_BODY_CHUNKS;
			ldsIdx[ch] = nb + (side ? nImp0 : 0) + tid;
			gi = impId[side];
		}
		flags[ch] = 0u;
		if (gi >= 0)
		{
			lvel[ldsIdx[ch]] = g.vel[gi];
			ldq[ldsIdx[ch]] = g.dq[gi];
			flags[ch] = g.flags[gi] | 0x80000000u; // bit 31: slot in use
			linteg[ldsIdx[ch]] = g.integ[gi];
			langDamp[ldsIdx[ch]] = g.angDamp[gi];
			lmass[ldsIdx[ch]] = g.massInv[gi];
		}
	}
	__syncthreads();
	if (stamp)
	{
		pv.debugTimes[stamps++] = wall_clock64();
	}

	LdsBodies lb{lvel, ldq};
	PersistShared shared;
	shared.massInv = lmass;
	shared.softCoef[0] = pv.softCoef[0], shared.softCoef[1] = pv.softCoef[1];
	unsigned epoch = 0; // tags are the exchange number: the buffers are zero at launch (cleared by the previous step's epilogue)
	int bad = 0;
	for (int oi = 0; oi < opCount && !bad; ++oi)
	{
		const Op op = lops[oi];
		uint32_t salt;
		asm volatile("s_mov_b32 %0, 0" : "=s"(salt));
		if (op.code == OP_INTEGRATE_VEL)
		{
#pragma unroll
			for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS + 2; ++ch)
			{
				if ((flags[ch] & S2F_DYNAMIC) != 0)
				{
					const int i = ldsIdx[ch];
					float4 v = lvel[i], k = linteg[i];
					V2 lv = add(v2(v.x, v.y), v2(k.x, k.y));
					float w = v.z + k.z;
					lv = mulSV(k.w, lv);
					w *= langDamp[i];
					lvel[i] = make_float4(lv.x, lv.y, w, 0.0f);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_INTEGRATE_POS)
		{
#pragma unroll
			for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS + 2; ++ch)
			{
				if ((flags[ch] & S2F_MOVES) != 0)
				{
					const int i = ldsIdx[ch];
					float4 v = lvel[i], d = ldq[i];
					V2 dpos = mulAdd(v2(d.x, d.y), op.h, v2(v.x, v.y));
					Rot q;
					q.s = d.z, q.c = d.w;
					q = integrateRot(q, op.h * v.z);
					ldq[i] = make_float4(dpos.x, dpos.y, q.s, q.c);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_FINALIZE)
		{
#pragma unroll
			for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS + 2; ++ch)
			{
				if (flags[ch] != 0u)
				{
					const int i = ldsIdx[ch];
					if (ch < S2_STRIP_BODY_CHUNKS)
					{
						finalizePositionsOne(lb, i, g, (int)(id[ch] & ~S2G_OWNED), op.flag, (id[ch] & S2G_OWNED) != 0);
					}
					else if ((flags[ch] & (op.flag ? S2F_DYNAMIC : S2F_MOVES)) != 0)
					{
						float4 d = ldq[i]; // the copy of a neighbour's body: same reset, its owner writes the position
						ldq[i] = make_float4(0.0f, 0.0f, d.z, d.w);
					}
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_WARM)
		{
			// s2WarmStartContacts as a coloured sweep WITHOUT an exchange: a side's warm-start term depends on the
			// impulses, the anchors and that body's own pose only, so every body this workgroup owns ends up with
			// the right bits; the copies of the neighbours' bodies are refreshed by the next sweep's exchange
			// before anything reads them
#pragma unroll
			for (int i = 0; i < ROUNDS; ++i)
			{
				if (i < roundsA)
				{
					if (kOfRound(i) >= 0)
					{
						warmSoftRegs<WARM>(unpackPersist<KIND, WARM>(rA[i], shared, salt), lb);
					}
					__syncthreads();
				}
			}
			if constexpr (SEAMREG)
			{
#pragma unroll
				for (int i = 0; i < SEAM_REG_ROUNDS; ++i)
				{
					if (i < roundsB)
					{
#pragma unroll
						for (int pass = 0; pass < 2; ++pass)
						{
							if ((seamMask >> (2 * i + pass)) & 1u)
							{
								warmSoftRegs<WARM>(unpackPersist<KIND, WARM>(rB[2 * i + pass], shared, salt), lb);
							}
						}
						__syncthreads();
					}
				}
			}
			else
			{
#pragma unroll 1
				for (int i = 0; i < roundsB; ++i)
				{
#pragma unroll
					for (int pass = 0; pass < 2; ++pass)
					{
						int side, k, slot;
						if (seamItem(i, pass, side, k, slot, salt))
						{
							float4 q[Q];
#pragma unroll
							for (int f = 0; f < Q; ++f)
							{
								q[f] = lseam[f * seamSlots + slot];
							}
							PersistRegs<KIND, WARM> pb;
							__builtin_memcpy(&pb, q, sizeof(pb));
							warmSoftRegs<WARM>(unpackPersist<KIND, WARM>(pb, shared), lb);
						}
					}
					__syncthreads();
				}
			}
		}
		else if (op.code == OP_SOLVE_SOFT)
		{
			// ---- interiors ----
#pragma unroll
			for (int i = 0; i < ROUNDS; ++i)
			{
				if (i < roundsA && (!S2_PERSIST_INSTRUMENTED || (pv.debugSkip & 4) == 0))
				{
					if (kOfRound(i) >= 0)
					{
						sweepPersist<KIND, WARM, POINTS>(rA[i], shared, c, lb, op.inv_h, op.useBias, kOfRound(i), salt);
					}
					__syncthreads();
				}
			}
			// ---- symmetric exchange of the seam bodies' velocities (poses are replicated by the body stages) ----
			epoch += 1;
			const int par = (int)(epoch & 1u) * pv.parityStride;
			const bool mute = (pv.debugSkip & 8) != 0 && blockIdx.x == 1; // fault injection: this workgroup stays silent
			if (tid < nExp0 && !mute)
			{
				float4 v = lvel[expIdx[0]];
				gu64* p = gran + par + out0 + 4 * tid;
				putGranule(p + 0, epoch, v.x), putGranule(p + 1, epoch, v.y), putGranule(p + 2, epoch, v.z);
			}
			if (tid < nExp1 && !mute)
			{
				float4 v = lvel[expIdx[1]];
				gu64* p = gran + par + out1 + 4 * tid;
				putGranule(p + 0, epoch, v.x), putGranule(p + 1, epoch, v.y), putGranule(p + 2, epoch, v.z);
			}
			int fail = 0;
			if (!S2_PERSIST_INSTRUMENTED || (pv.debugSkip & 1) == 0)
			{
				float v[3];
				if (tid < nImp0)
				{
					if (getGranules<3>(gran + par + in0 + 4 * tid, epoch, v, pv.error, pv.deviceError, pv.spinLimit))
					{
						lvel[nb + tid] = make_float4(v[0], v[1], v[2], 0.0f);
					}
					else
					{
						fail = 1;
					}
				}
				if (tid < nImp1)
				{
					if (getGranules<3>(gran + par + in1 + 4 * tid, epoch, v, pv.error, pv.deviceError, pv.spinLimit))
					{
						lvel[nb + nImp0 + tid] = make_float4(v[0], v[1], v[2], 0.0f);
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
			// ---- both seams (the neighbours compute the same bits on their side) ----
			if constexpr (SEAMREG)
			{
#pragma unroll
				for (int i = 0; i < SEAM_REG_ROUNDS; ++i)
				{
					if (i < roundsB && (!S2_PERSIST_INSTRUMENTED || (pv.debugSkip & 2) == 0))
					{
#pragma unroll
						for (int pass = 0; pass < 2; ++pass)
						{
							if ((seamMask >> (2 * i + pass)) & 1u)
							{
								// (the constraint's sweep position is only used by the Jacobi kind, which never runs here)
								sweepPersist<KIND, WARM, POINTS>(rB[2 * i + pass], shared, c, lb, op.inv_h, op.useBias, 0, salt);
							}
						}
						__syncthreads();
					}
				}
			}
			else
			{
#pragma unroll 1
				for (int i = 0; i < roundsB && (!S2_PERSIST_INSTRUMENTED || (pv.debugSkip & 2) == 0); ++i)
				{
#pragma unroll
					for (int pass = 0; pass < 2; ++pass)
					{
						int side, k, slot;
						if (seamItem(i, pass, side, k, slot, salt))
						{
							float4 q[Q];
#pragma unroll
							for (int f = 0; f < Q; ++f)
							{
								q[f] = lseam[f * seamSlots + slot];
							}
							PersistRegs<KIND, WARM> pb;
							__builtin_memcpy(&pb, q, sizeof(pb));
							sweepPersist<KIND, WARM, POINTS>(pb, shared, c, lb, op.inv_h, op.useBias, k);
							// only the impulses changed: PersistRegs starts with them, record 0
							__builtin_memcpy(&q[0], &pb.imp[0], 16);
							lseam[slot] = q[0];
						}
					}
					__syncthreads();
				}
			}
		}
		if (stamp)
		{
			pv.debugTimes[stamps++] = wall_clock64();
		}
	}

	// ---- results: owned bodies, impulses ----
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		int i = tid + ch * S2_STRIP_THREADS;
		if (i < nb && (id[ch] & S2G_OWNED) != 0)
		{
			int gi = (int)(id[ch] & ~S2G_OWNED);
			g.vel[gi] = lvel[i];
			g.dq[gi] = ldq[i];
		}
	}
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		if (kOfRound(i) >= 0)
		{
			storeSoft<KIND>(c, unpackPersist<KIND, WARM>(rA[i], shared), kOfRound(i));
		}
	}
	// the right seam's impulses are stored by this workgroup (its left neighbour of that seam), nobody stores twice
	if constexpr (SEAMREG)
	{
#pragma unroll
		for (int i = 0; i < SEAM_REG_ROUNDS; ++i)
		{
#pragma unroll
			for (int pass = 0; pass < 2; ++pass)
			{
				int side, k, slot;
				if (i < roundsB && seamItem(i, pass, side, k, slot) && side == 1)
				{
					storeSoft<KIND>(c, unpackPersist<KIND, WARM>(rB[2 * i + pass], shared), k);
				}
			}
		}
	}
	else
#pragma unroll 1
	for (int i = 0; i < roundsB; ++i)
	{
#pragma unroll 1
		for (int pass = 0; pass < 2; ++pass)
		{
			int side, k, slot;
			if (seamItem(i, pass, side, k, slot) && side == 1)
			{
				float4 q[Q];
#pragma unroll
				for (int f = 0; f < Q; ++f)
				{
					q[f] = lseam[f * seamSlots + slot];
				}
				PersistRegs<KIND, WARM> pb;
				__builtin_memcpy(&pb, q, sizeof(pb));
				storeSoft<KIND>(c, unpackPersist<KIND, WARM>(pb, shared), k);
			}
		}
	}
	if (stamp)
	{
		pv.debugTimes[stamps++] = wall_clock64();
		pv.debugTimes[255] = (unsigned long long)stamps;
	}
}

#include "soft_from_wire.h"

// ------------------------------------------------------------------------------------------------
// Resident island step: the persistent strip step without seams.  An LDS group -- one or several small simulation
// islands (a base-40 pyramid: 820 boxes, 2,380 constraints) -- is advanced through the WHOLE s2Solve_* by one workgroup
// that keeps the group's constraints in REGISTERS (PersistRegs, one per colour round and thread) and its bodies in LDS.
// Islands exchange nothing, so there are no hand-offs, no co-residency requirement and no limit on the number of
// workgroups.  Against groupKernel (group_kernel.hip), which streams every constraint record from its SoA arrays in
// every sweep (120 B per constraint-sweep, 24 sweeps per TGS_Soft 8/4 step: bound by the Infinity Cache at 512 groups),
// a record is read ONCE per step.  THREADS = 512: two waves per SIMD, a colour round holds up to 512 constraints.
// ------------------------------------------------------------------------------------------------
// The kernel also is its own prologue and epilogue for these constraints: records are prepared from the wire contacts and
// the impulses go back into them (s2StoreContactImpulses, solve_common.c:396-410) -- no SoA round trip at all.
template <int KIND, int WARM, int ROUNDS, int THREADS>
__global__ __launch_bounds__(THREADS) void islandStepKernel(ContactView c, BodyView g, StripTableView ta, float4 softCoef0, float4 softCoef1, const Op* ops,
															 int opCount, s2amdContact* wire, const s2amdBody* wireBodies, const uint32_t* hostFlags, int warmStart,
															 const unsigned int* stepFailed)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	if (stepFailed != nullptr && *stepFailed != 0u)
	{
		return; // a persistent strip kernel of this step lost a hand-off: the step will be repeated, nothing of it may reach the wire arrays
	}
	const int tid = (int)threadIdx.x;
	const StripDesc* da = ta.descs + blockIdx.x;
	const int bodyBase = da->bodyBase, nb = da->bodyCount, roundsA = da->batchCount;
	int4 batchA[ROUNDS];
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		batchA[i] = da->batch[i];
	}
	float4* lvel = lds;
	float4* ldq = lds + nb;
	float4* linteg = lds + 2 * nb;
	float* langDamp = (float*)(lds + 3 * nb);
	float2* lmass = (float2*)(lds + 3 * nb + (nb + 3) / 4);
	float2* llc = (float2*)(lds + 3 * nb + (nb + 3) / 4 + (nb + 1) / 2); // the bodies' local centres (soft_from_wire.h: prepareSoftFromWire)
	const int bodyRecords = 3 * nb + (nb + 3) / 4 + 2 * ((nb + 1) / 2);
	Op* lops = (Op*)(lds + bodyRecords);

	uint32_t id[S2_STRIP_BODY_CHUNKS];
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		int i = tid + ch * THREADS;
		id[ch] = i < nb ? (uint32_t)ta.bodyIds[bodyBase + i] : 0u;
	}
	for (int i = tid; i < opCount * 8; i += THREADS)
	{
		((int*)lops)[i] = ((const int*)ops)[i];
	}
	PersistRegs<KIND, WARM> rA[ROUNDS];
	auto kOfRound = [&](int i) {
		int k = batchA[i].x + tid;
		return (i < roundsA && k < batchA[i].y) ? k : -1;
	};
	// this thread's constraints: pool slot and group-local body slots (the wire records follow once the bodies are staged)
	int slotOf[ROUNDS];
	int2 localOf[ROUNDS];
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		slotOf[i] = -1;
		localOf[i] = make_int2(0, 0);
		if (kOfRound(i) >= 0)
		{
			slotOf[i] = c.contactIndex[kOfRound(i)];
			localOf[i] = c.localBodies[kOfRound(i)];
		}
	}
	uint32_t flags[S2_STRIP_BODY_CHUNKS];
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		const int i = tid + ch * THREADS;
		flags[ch] = 0u;
		if (i < nb)
		{
			const int gi = (int)(id[ch] & ~S2G_OWNED);
			lvel[i] = g.vel[gi];
			ldq[i] = g.dq[gi];
			flags[ch] = g.flags[gi] | 0x80000000u;
			linteg[i] = g.integ[gi];
			langDamp[i] = g.angDamp[gi];
			lmass[i] = g.massInv[gi];
			llc[i] = make_float2(wireBodies[gi].localCenter[0], wireBodies[gi].localCenter[1]);
		}
	}
	__syncthreads();

	LdsBodies lb{lvel, ldq};
	PersistShared shared;
	shared.massInv = lmass;
	shared.softCoef[0] = softCoef0, shared.softCoef[1] = softCoef1;
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		if (slotOf[i] >= 0)
		{
			SoftRegs<KIND> t = prepareSoftFromWire<KIND>(wire + slotOf[i], wireBodies, hostFlags, lb, lmass, localOf[i], g.capacity, warmStart, llc);
			rA[i] = packPersist<KIND, WARM>(t, localOf[i].x, localOf[i].y);
		}
	}
	for (int oi = 0; oi < opCount; ++oi)
	{
		const Op op = lops[oi];
		uint32_t salt;
		asm volatile("s_mov_b32 %0, 0" : "=s"(salt));
		if (op.code == OP_INTEGRATE_VEL)
		{
#pragma unroll
			for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
			{
				if ((flags[ch] & S2F_DYNAMIC) != 0)
				{
					const int i = tid + ch * THREADS;
					float4 v = lvel[i], k = linteg[i];
					V2 lv = add(v2(v.x, v.y), v2(k.x, k.y));
					float w = v.z + k.z;
					lv = mulSV(k.w, lv);
					w *= langDamp[i];
					lvel[i] = make_float4(lv.x, lv.y, w, 0.0f);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_INTEGRATE_POS)
		{
#pragma unroll
			for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
			{
				if ((flags[ch] & S2F_MOVES) != 0)
				{
					const int i = tid + ch * THREADS;
					float4 v = lvel[i], d = ldq[i];
					V2 dpos = mulAdd(v2(d.x, d.y), op.h, v2(v.x, v.y));
					Rot q;
					q.s = d.z, q.c = d.w;
					q = integrateRot(q, op.h * v.z);
					ldq[i] = make_float4(dpos.x, dpos.y, q.s, q.c);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_FINALIZE)
		{
#pragma unroll
			for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
			{
				if (flags[ch] != 0u)
				{
					finalizePositionsOne(lb, tid + ch * THREADS, g, (int)(id[ch] & ~S2G_OWNED), op.flag, (id[ch] & S2G_OWNED) != 0);
				}
			}
			__syncthreads();
		}
		else if (op.code == OP_WARM)
		{
#pragma unroll
			for (int i = 0; i < ROUNDS; ++i)
			{
				if (i < roundsA)
				{
					if (kOfRound(i) >= 0)
					{
						warmSoftRegs<WARM>(unpackPersist<KIND, WARM>(rA[i], shared, salt), lb);
					}
					__syncthreads();
				}
			}
		}
		else if (op.code == OP_SOLVE_SOFT)
		{
#pragma unroll
			for (int i = 0; i < ROUNDS; ++i)
			{
				if (i < roundsA)
				{
					if (kOfRound(i) >= 0)
					{
						sweepPersist<KIND, WARM, 0>(rA[i], shared, c, lb, op.inv_h, op.useBias, kOfRound(i), salt);
					}
					__syncthreads();
				}
			}
		}
	}
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		int i = tid + ch * THREADS;
		if (i < nb && (id[ch] & S2G_OWNED) != 0)
		{
			int gi = (int)(id[ch] & ~S2G_OWNED);
			g.vel[gi] = lvel[i];
			g.dq[gi] = ldq[i];
		}
	}
	// s2StoreContactImpulses (solve_common.c:396-410): straight into the manifolds
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		if (slotOf[i] >= 0)
		{
			const int pointCount = (int)((rA[i].idx >> 28) & 3u);
			s2amdContact* contact = wire + slotOf[i];
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				if (j < pointCount)
				{
					contact->points[j].normalImpulse = rA[i].imp[j].x;
					contact->points[j].tangentImpulse = rA[i].imp[j].y;
				}
			}
		}
	}
}

#define S2_ISLAND_THREADS 512
template <int KIND, int WARM>
static void launchIsland(hipStream_t s, dim3 grid, size_t lds, const ContactView& c, const BodyView& g, const StripTableView& t, const float4* coef, const Op* ops,
						 int opCount, int rounds, s2amdContact* wire, const s2amdBody* wireBodies, const uint32_t* hostFlags, int warmStart, const unsigned int* stepFailed)
{
	if (rounds <= S2_STRIP_ROUNDS)
	{
		islandStepKernel<KIND, WARM, S2_STRIP_ROUNDS, S2_ISLAND_THREADS>
			<<<grid, dim3(S2_ISLAND_THREADS), lds, s>>>(c, g, t, coef[0], coef[1], ops, opCount, wire, wireBodies, hostFlags, warmStart, stepFailed);
	}
	else
	{
		islandStepKernel<KIND, WARM, S2_STRIP_ROUNDS_MAX, S2_ISLAND_THREADS>
			<<<grid, dim3(S2_ISLAND_THREADS), lds, s>>>(c, g, t, coef[0], coef[1], ops, opCount, wire, wireBodies, hostFlags, warmStart, stepFailed);
	}
}

// t.ldsRecords: body records of the largest group; maxRounds: colour rounds of the group with the most
void launchIslandStep(hipStream_t s, int kind, int warm, const ContactView& c, const BodyView& g, const StripTableView& t, const float4* softCoef, const Op* ops,
					  int opCount, int maxRounds, s2amdContact* wire, const s2amdBody* wireBodies, const uint32_t* hostFlags, int warmStart, const unsigned int* stepFailed)
{
	if (t.groupCount <= 0)
	{
		return;
	}
	dim3 grid((unsigned)t.groupCount);
	size_t lds = (size_t)t.ldsRecords * sizeof(float4) + (size_t)opCount * sizeof(Op);
	if (kind == SOFT_TGS)
	{
		warm == WARM_FIXED ? launchIsland<SOFT_TGS, WARM_FIXED>(s, grid, lds, c, g, t, softCoef, ops, opCount, maxRounds, wire, wireBodies, hostFlags, warmStart, stepFailed)
						   : launchIsland<SOFT_TGS, WARM_CURRENT>(s, grid, lds, c, g, t, softCoef, ops, opCount, maxRounds, wire, wireBodies, hostFlags, warmStart, stepFailed);
	}
	else if (kind == SOFT_PGS)
	{
		warm == WARM_FIXED ? launchIsland<SOFT_PGS, WARM_FIXED>(s, grid, lds, c, g, t, softCoef, ops, opCount, maxRounds, wire, wireBodies, hostFlags, warmStart, stepFailed)
						   : launchIsland<SOFT_PGS, WARM_CURRENT>(s, grid, lds, c, g, t, softCoef, ops, opCount, maxRounds, wire, wireBodies, hostFlags, warmStart, stepFailed);
	}
	else
	{
		warm == WARM_FIXED ? launchIsland<SOFT_FIXED, WARM_FIXED>(s, grid, lds, c, g, t, softCoef, ops, opCount, maxRounds, wire, wireBodies, hostFlags, warmStart, stepFailed)
						   : launchIsland<SOFT_FIXED, WARM_CURRENT>(s, grid, lds, c, g, t, softCoef, ops, opCount, maxRounds, wire, wireBodies, hostFlags, warmStart, stepFailed);
	}
}

template <int KIND, int WARM>
static void launchStep(hipStream_t s, dim3 grid, size_t lds, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv,
					   const Op* ops, int opCount)
{
	dim3 block(S2_STRIP_THREADS);
	if (pv.wideRounds)
	{
		if (pv.allTwoPoints)
		{
			stripStepKernel<KIND, WARM, 2, S2_STRIP_ROUNDS_MAX, 0><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount);
		}
		else
		{
			stripStepKernel<KIND, WARM, 0, S2_STRIP_ROUNDS_MAX, 0><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount);
		}
	}
	else if (pv.seamRegs && KIND == SOFT_TGS && WARM == WARM_CURRENT)
	{
		// TGS_Soft only: the other kinds' records are 30-32 dwords and four more of them spill; and the two-point fast path
		// does not fit the register file together with them either (249 spilled registers), so this is the per-point variant
		if constexpr (KIND == SOFT_TGS && WARM == WARM_CURRENT)
		{
			stripStepKernel<KIND, WARM, 0, S2_STRIP_ROUNDS, 1><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount);
		}
	}
	else if (pv.allTwoPoints)
	{
		stripStepKernel<KIND, WARM, 2, S2_STRIP_ROUNDS, 0><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount);
	}
	else
	{
		stripStepKernel<KIND, WARM, 0, S2_STRIP_ROUNDS, 0><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount);
	}
}

void launchStripStep(hipStream_t s, int kind, int warm, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv,
					 const Op* ops, int opCount)
{
	dim3 grid((unsigned)a.groupCount);
	size_t lds = (size_t)pv.ldsRecords * sizeof(float4) + (size_t)opCount * sizeof(Op); // ldsRecords: bodies + seam constraint records
	if (kind == SOFT_TGS)
	{
		warm == WARM_FIXED ? launchStep<SOFT_TGS, WARM_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount)
						   : launchStep<SOFT_TGS, WARM_CURRENT>(s, grid, lds, c, g, a, pv, ops, opCount);
	}
	else if (kind == SOFT_PGS)
	{
		warm == WARM_FIXED ? launchStep<SOFT_PGS, WARM_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount)
						   : launchStep<SOFT_PGS, WARM_CURRENT>(s, grid, lds, c, g, a, pv, ops, opCount);
	}
	else
	{
		warm == WARM_FIXED ? launchStep<SOFT_FIXED, WARM_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount)
						   : launchStep<SOFT_FIXED, WARM_CURRENT>(s, grid, lds, c, g, a, pv, ops, opCount);
	}
}

template <int KIND>
static void launchKind(hipStream_t s, int warm, dim3 grid, size_t lds, const ContactView& c, const BodyView& g, const StripTableView& t, const StripOps& ops)
{
	switch (warm)
	{
		case WARM_CURRENT:
			stripSoftKernel<KIND, WARM_CURRENT><<<grid, dim3(S2_STRIP_THREADS), lds, s>>>(c, g, t.descs, t.bodyIds, t.slots, t.slotOffsets, ops);
			break;
		case WARM_FIXED:
			stripSoftKernel<KIND, WARM_FIXED><<<grid, dim3(S2_STRIP_THREADS), lds, s>>>(c, g, t.descs, t.bodyIds, t.slots, t.slotOffsets, ops);
			break;
		default:
			stripSoftKernel<KIND, -1><<<grid, dim3(S2_STRIP_THREADS), lds, s>>>(c, g, t.descs, t.bodyIds, t.slots, t.slotOffsets, ops);
			break;
	}
}

void launchStripSoft(hipStream_t s, int kind, int warm, const ContactView& c, const BodyView& g, const StripTableView& t, const StripOps& ops)
{
	if (t.groupCount <= 0)
	{
		return;
	}
	dim3 grid((unsigned)t.groupCount);
	size_t lds = (size_t)t.ldsRecords * sizeof(float4);
	switch (kind)
	{
		case SOFT_TGS:
			launchKind<SOFT_TGS>(s, warm, grid, lds, c, g, t, ops);
			break;
		case SOFT_PGS:
			launchKind<SOFT_PGS>(s, warm, grid, lds, c, g, t, ops);
			break;
		case SOFT_FIXED:
			launchKind<SOFT_FIXED>(s, warm, grid, lds, c, g, t, ops);
			break;
	}
}

int stripKernelSetup()
{
	const void* fns[] = {
		(const void*)stripSoftKernel<SOFT_TGS, WARM_CURRENT>,	(const void*)stripSoftKernel<SOFT_TGS, WARM_FIXED>,	  (const void*)stripSoftKernel<SOFT_TGS, -1>,
		(const void*)stripSoftKernel<SOFT_PGS, WARM_CURRENT>,	(const void*)stripSoftKernel<SOFT_PGS, WARM_FIXED>,	  (const void*)stripSoftKernel<SOFT_PGS, -1>,
		(const void*)stripSoftKernel<SOFT_FIXED, WARM_CURRENT>, (const void*)stripSoftKernel<SOFT_FIXED, WARM_FIXED>, (const void*)stripSoftKernel<SOFT_FIXED, -1>,
	};
	const void* steps[] = {
		(const void*)stripStepKernel<SOFT_TGS, WARM_CURRENT, 0, S2_STRIP_ROUNDS, 1>,
#define S2_STEP_VARIANTS(K, W)                                                                                                   \
	(const void*)stripStepKernel<K, W, 0, S2_STRIP_ROUNDS, 0>, (const void*)stripStepKernel<K, W, 2, S2_STRIP_ROUNDS, 0>,        \
		(const void*)stripStepKernel<K, W, 0, S2_STRIP_ROUNDS_MAX, 0>, (const void*)stripStepKernel<K, W, 2, S2_STRIP_ROUNDS_MAX, 0>
		S2_STEP_VARIANTS(SOFT_TGS, WARM_CURRENT),	S2_STEP_VARIANTS(SOFT_TGS, WARM_FIXED),	  S2_STEP_VARIANTS(SOFT_PGS, WARM_CURRENT),
		S2_STEP_VARIANTS(SOFT_PGS, WARM_FIXED),		S2_STEP_VARIANTS(SOFT_FIXED, WARM_CURRENT), S2_STEP_VARIANTS(SOFT_FIXED, WARM_FIXED),
#undef S2_STEP_VARIANTS
	};
	const void* islands[] = {
#define S2_ISLAND_VARIANTS(K, W)                                                                                                 \
	(const void*)islandStepKernel<K, W, S2_STRIP_ROUNDS, S2_ISLAND_THREADS>, (const void*)islandStepKernel<K, W, S2_STRIP_ROUNDS_MAX, S2_ISLAND_THREADS>
		S2_ISLAND_VARIANTS(SOFT_TGS, WARM_CURRENT),	 S2_ISLAND_VARIANTS(SOFT_TGS, WARM_FIXED),	 S2_ISLAND_VARIANTS(SOFT_PGS, WARM_CURRENT),
		S2_ISLAND_VARIANTS(SOFT_PGS, WARM_FIXED),	 S2_ISLAND_VARIANTS(SOFT_FIXED, WARM_CURRENT), S2_ISLAND_VARIANTS(SOFT_FIXED, WARM_FIXED),
#undef S2_ISLAND_VARIANTS
	};
	for (const void* f : islands)
	{
		hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		if (e != hipSuccess)
		{
			return (int)e;
		}
	}
	for (const void* f : steps)
	{
		hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		if (e != hipSuccess)
		{
			return (int)e;
		}
	}
	for (const void* f : fns)
	{
		hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		if (e != hipSuccess)
		{
			return (int)e;
		}
	}
	return 0;
}

S2_DEFINE_WARM(strip_kernel)
