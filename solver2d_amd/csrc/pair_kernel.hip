// Persistent strip step, two lanes per constraint ("pair lanes").
//
// The same step as strip_kernel.hip: stripStepKernel -- workgroup i owns strip i of a big island for the WHOLE s2Solve_*
// call, its interior and seam constraints resident in registers, its bodies in LDS, seam bodies exchanged with the two
// neighbouring workgroups once per sweep as tagged granules -- and the same tables (StripDesc, PersistDesc), the same sweep
// order, the same bits.  What differs is how a colour round is mapped onto the lanes:
//
//   stripStepKernel: 256 threads (one wave per SIMD), lane t solves constraint t of the round: ~300 VALU instructions per
//                    round issued by ONE wave per SIMD (that kernel is instruction-issue bound inside a wave);
//   pairStepKernel:  512 threads (two waves per SIMD), lanes 2c and 2c+1 solve constraint c of the round TOGETHER: the even
//                    lane is body A's side, the odd lane body B's side.  Each lane loads, rotates and updates ONE body; the
//                    only quantities that cross the pair are the three differences the reference forms between the sides --
//                    dcB - dcA, rB - rA, and the anchor velocities vrB - vrA (solve_tgs_soft.c:17-135) -- and they cross as
//                    a quad-permute DPP operand (no LDS, no extra round trip).  The impulse chain (separation, bias,
//                    relative velocity -> impulse -> clamp) is computed by both lanes on identical operands, so both hold
//                    identical impulses without talking again.  A lane's resident record is one side of the constraint.
//
// Bit-exactness of the split.  Every per-side value (r = rotate(q, l), vr = v + w x r, the applied impulse) is computed with
// the reference's operations on the reference's operands.  A difference tB - tA is formed as (tB) + (-tA) in BOTH lanes:
// each lane flips the sign bit of its own term when it is the A side (an exact negation, one v_xor), then adds its partner's
// flipped term (IEEE addition is commutative, and x - y == x + (-y) bit for bit, zeros and infinities included).  Negation
// is never moved across an addition (-(a + b) and (-a) + (-b) differ in the sign of an exact zero).  The sides' applications
// `vA -= mA P, wA -= iA (rA x P)` and `vB += mB P, wB += iB (rB x P)` become one form, v += sm P, w += si (r x P), with the
// sign of the A side carried by the inverse mass and inertia (sm = -mA: (-m) p == -(m p) and v + (-t) == v - t exactly).
//
// Arithmetic reference: constraint_ops.h solveSoftRegs / strip_kernel.hip warmSoftRegs, which the parity tests compare with
// (the oracle sweeps in the order s2amd_get_contact_order reports; that order does not depend on the kernel).

#include "body_ops.h"
#include "persist_handoff.h"

// 1: in-kernel time stamps (S2AMD_DEBUG_TIMES) are compiled in -- `make variant NAME=stamps EXTRA=-DS2_PERSIST_INSTRUMENTED=1`
#ifndef S2_PERSIST_INSTRUMENTED
#define S2_PERSIST_INSTRUMENTED 0
#endif
#define S2_PAIR_THREADS 512
#define S2_PAIR_SLOTS 256		  // constraints per round and pass
#define S2_PAIR_BODY_CHUNKS 2	  // own bodies per thread: a strip stages at most 2 * 512 = S2_STRIP_BODY_CHUNKS * 256
#define S2_PAIR_SEAM_ROUNDS 2	  // seam colour batches kept in registers (two passes each)

// quad_perm [1, 0, 3, 2]: every lane reads its pair partner
S2_DEV float swapPair(float x)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}
S2_DEV uint32_t swapPairBits(uint32_t x)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
}
S2_DEV float flipSign(float x, uint32_t su)
{
	return fromBits(asBits(x) ^ su);
}
// own-side term t of a difference tB - tA: both lanes of the pair end up with the bits of sub(tB, tA)
S2_DEV float pairDiff(float t, uint32_t su)
{
	const float s = flipSign(t, su);
	return s + swapPair(s);
}

// One side of one constraint as a lane keeps it for the whole step.  Fields a (KIND, WARM) combination never reads are dead
// and take no register.
struct HalfRegs
{
	uint32_t idx; // own body's LDS slot | pointCount << 28 | write << 30 | (a side is static: the doubled contact hertz) << 31
	float nx, ny, friction;
	float ax[2], ay[2]; // own local anchor, relative to the centre of mass
	float rx[2], ry[2]; // own prepare-time arm (rA0 / rB0)
	float p0[2], p1[2], p2[2], p3[2]; // adjustedSeparation, normalMass, tangentMass, prepare-time separation
	float in[2], it[2]; // normal / tangent impulses (identical in both lanes of the pair)
};

template <int KIND, int WARM> S2_DEV HalfRegs loadHalf(const ContactView& c, int k, int side, int slot)
{
	constexpr bool kAnchors = KIND == SOFT_TGS || KIND == SOFT_FIXED || WARM == WARM_CURRENT;
	constexpr bool kArms0 = KIND != SOFT_TGS || WARM == WARM_FIXED;
	HalfRegs p;
	const float4 nf = c.nf[k];
	const uint32_t bits = asBits(nf.w);
	const bool write = (bits & (side ? S2C_WRITE_B : S2C_WRITE_A)) != 0;
	p.idx = (uint32_t)slot | ((bits & 3u) << 28) | (write ? 1u << 30 : 0u);
	p.nx = nf.x, p.ny = nf.y, p.friction = nf.z;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		p.ax[j] = p.ay[j] = p.rx[j] = p.ry[j] = 0.0f;
		if (kAnchors)
		{
			const float4 a = c.anchor[j][k];
			p.ax[j] = side ? a.z : a.x, p.ay[j] = side ? a.w : a.y;
		}
		if (kArms0)
		{
			const float4 a = c.r0[j][k];
			p.rx[j] = side ? a.z : a.x, p.ry[j] = side ? a.w : a.y;
		}
		const float4 par = c.param[j][k];
		p.p0[j] = par.x, p.p1[j] = par.y, p.p2[j] = par.z, p.p3[j] = par.w;
		const float2 imp = c.impulse[j][k];
		p.in[j] = imp.x, p.it[j] = imp.y;
	}
	return p;
}

// the doubled contact hertz of a constraint with a static side (prepareContactsKernel<PREP_SOFT>; solve_common.c:219): the
// same test as strip_kernel.hip unpackPersist, made once -- the masses do not change during a step
S2_DEV void markStaticSide(HalfRegs& p, const float2* lmass)
{
	const uint32_t own = lmass[p.idx & 0x3fffu].x == 0.0f ? 1u : 0u;
	const uint32_t other = swapPairBits(own);
	p.idx |= (own | other) << 31;
}

// s2WarmStartContacts (solve_common.c:276-330) / s2WarmStartContacts_Fixed (solve_soft_step.c:16-63): this lane's side
template <int WARM, int POINTS> S2_DEV void warmPair(const HalfRegs& p, float4* lvel, const float4* ldq, const float2* lmass, uint32_t su, uint32_t salt)
{
	const uint32_t idx = p.idx ^ salt;
	const int slot = (int)(idx & 0x3fffu);
	const int pointCount = (int)((idx >> 28) & 3u);
	const float4 vel = lvel[slot];
	const float2 mi = lmass[slot];
	Rot q;
	if (WARM == WARM_CURRENT)
	{
		const float4 d = ldq[slot];
		q.s = d.z, q.c = d.w;
	}
	const V2 normal = v2(fromBits(asBits(p.nx) ^ salt), p.ny);
	const V2 tangent = rightPerp(normal);
	const float sm = flipSign(mi.x, su), si = flipSign(mi.y, su);
	V2 v = v2(vel.x, vel.y);
	float w = vel.z;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			const V2 r = WARM == WARM_CURRENT ? rotate(q, v2(p.ax[j], p.ay[j])) : v2(p.rx[j], p.ry[j]);
			const V2 P = add(mulSV(p.in[j], normal), mulSV(p.it[j], tangent));
			w += si * cross(r, P);
			v = mulAdd(v, sm, P);
		}
	}
	if ((idx & (1u << 30)) != 0)
	{
		lvel[slot] = make_float4(v.x, v.y, w, 0.0f);
	}
}

// s2SolveContacts_TGS_Soft (solve_tgs_soft.c:17-135), _PGS_Soft (solve_pgs_soft.c:16-125), _TGS_Fixed
// (solve_soft_step.c:66-177): this lane's side of one constraint; see the header for what crosses the pair
template <int KIND, int POINTS>
S2_DEV void solvePair(HalfRegs& p, float4* lvel, const float4* ldq, const float2* lmass, const float4& coef0, const float4& coef1, float inv_h, int useBias,
					  uint32_t su, uint32_t salt)
{
	const uint32_t idx = p.idx ^ salt;
	const int slot = (int)(idx & 0x3fffu);
	const int pointCount = (int)((idx >> 28) & 3u);
	const float biasCap = KIND == SOFT_TGS ? -S2_MAX_BAUMGARTE_VELOCITY : -0.5f * S2_MAX_BAUMGARTE_VELOCITY;
	const float4 vel = lvel[slot];
	const float2 mi = lmass[slot];
	Rot q;
	V2 dd;
	if (KIND == SOFT_TGS || KIND == SOFT_FIXED)
	{
		const float4 d = ldq[slot];
		q.s = d.z, q.c = d.w;
		dd = v2(pairDiff(d.x, su), pairDiff(d.y, su)); // sub(dcB, dcA)
	}
	const float4 sf = (idx & 0x80000000u) != 0 ? coef1 : coef0;
	const V2 normal = v2(fromBits(asBits(p.nx) ^ salt), p.ny);
	const V2 tangent = rightPerp(normal);
	const float sm = flipSign(mi.x, su), si = flipSign(mi.y, su);
	V2 v = v2(vel.x, vel.y);
	float w = vel.z;
	V2 rj[2];
	float nImp[2], tImp[2];

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			V2 r;
			float s;
			if (KIND == SOFT_TGS)
			{
				r = rotate(q, v2(p.ax[j], p.ay[j]));
				const V2 ds = add(dd, v2(pairDiff(r.x, su), pairDiff(r.y, su)));
				s = dot(ds, normal) + p.p0[j];
			}
			else if (KIND == SOFT_FIXED)
			{
				const V2 rc = rotate(q, v2(p.ax[j], p.ay[j]));
				const V2 ds = add(dd, v2(pairDiff(rc.x, su), pairDiff(rc.y, su)));
				s = dot(ds, normal) + p.p0[j];
				r = v2(p.rx[j], p.ry[j]);
			}
			else
			{
				s = p.p3[j];
				r = v2(p.rx[j], p.ry[j]);
			}
			rj[j] = r;

			// select form of: if (s > 0) bias = s * inv_h; else if (useBias) {bias = max(biasCoefficient * s, cap); ...}
			const bool speculative = s > 0.0f;
			const bool soft = !speculative && useBias != 0;
			const float softBias = S2_MAXF(sf.x * s, biasCap);
			const float bias = speculative ? s * inv_h : (soft ? softBias : 0.0f);
			const float massScale = soft ? sf.y : 1.0f;
			const float impulseScale = soft ? sf.z : 0.0f;

			const V2 vr = add(v, crossSV(w, r));
			const V2 dv = v2(pairDiff(vr.x, su), pairDiff(vr.y, su)); // sub(vrB, vrA)
			const float vn = dot(dv, normal);

			const float normalMass = fromBits(asBits(p.p1[j]) ^ salt);
			float impulse = -normalMass * massScale * (vn + bias) - impulseScale * p.in[j];
			const float newImpulse = S2_MAXF(p.in[j] + impulse, 0.0f);
			impulse = newImpulse - p.in[j];
			nImp[j] = newImpulse;
			tImp[j] = p.it[j];

			const V2 P = mulSV(impulse, normal);
			v = mulAdd(v, sm, P);
			w += si * cross(r, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			const float tangentMass = fromBits(asBits(p.p2[j]) ^ salt);
			const V2 r = rj[j];
			const V2 vr = add(v, crossSV(w, r));
			const V2 dv = v2(pairDiff(vr.x, su), pairDiff(vr.y, su));
			const float vt = dot(dv, tangent);
			float impulse = -tangentMass * vt;
			const float maxFriction = p.friction * nImp[j];
			const float newImpulse = S2_CLAMPF(tImp[j] + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - tImp[j];
			const V2 P = mulSV(impulse, tangent);
			v = mulAdd(v, sm, P);
			w += si * cross(r, P);
			p.in[j] = nImp[j], p.it[j] = newImpulse;
		}
	}

	if ((idx & (1u << 30)) != 0)
	{
		lvel[slot] = make_float4(v.x, v.y, w, 0.0f);
	}
}

S2_DEV void storeHalf(const ContactView& c, const HalfRegs& p, int k)
{
	const int pointCount = (int)((p.idx >> 28) & 3u);
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			c.impulse[j][k] = make_float2(p.in[j], p.it[j]);
		}
	}
}

// POINTS == 2: the host has checked that every constraint of the strips has two manifold points: no per-point masking.
// ROUNDS: interior colour batches kept in registers.
template <int KIND, int WARM, int POINTS, int ROUNDS>
__global__ __launch_bounds__(S2_PAIR_THREADS) void pairStepKernel(ContactView c, BodyView g, StripTableView ta, PersistView pv, const Op* ops, int opCount)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	const int tid = (int)threadIdx.x;
	const int ct = tid >> 1;						   // this lane's constraint of a round
	const int side = tid & 1;						   // 0: body A's lane, 1: body B's lane
	const uint32_t su = side ? 0u : 0x80000000u;	   // the A side enters every B - A difference negated
	const int half = tid >> 8, ht = tid & 255;		   // hand-offs: waves 0-3 serve the left neighbour, waves 4-7 the right
	// stamps: (wall_clock64 << 4) | tag; tags: 0 start, 1 loaded, 2 body stage, 3 warm start, 4 interior rounds, 5 hand-off, 6 seam rounds, 7 end
	const bool stamp = S2_PERSIST_INSTRUMENTED && pv.debugTimes != nullptr && blockIdx.x == gridDim.x / 2 && tid == 0;
	int stamps = 0;
	auto stampAt = [&](unsigned tag) {
		if (stamp && stamps < 250)
		{
			pv.debugTimes[stamps++] = (wall_clock64() << 4) | tag;
		}
	};
	stampAt(0);
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
	const int roundsB0 = pd->seamBatchCount[0], roundsB1 = pd->seamBatchCount[1];
	int2 batchB0[S2_PAIR_SEAM_ROUNDS], batchB1[S2_PAIR_SEAM_ROUNDS];
#pragma unroll
	for (int i = 0; i < S2_PAIR_SEAM_ROUNDS; ++i)
	{
		batchB0[i] = pd->seamBatch[0][i];
		batchB1[i] = pd->seamBatch[1][i];
	}
	const int roundsB = roundsB0 > roundsB1 ? roundsB0 : roundsB1;
	const int nt = nb + nImp0 + nImp1;
	gu64* gran = (gu64*)pv.granules;
	// this half's side of the exchange
	const int nImpH = half ? nImp1 : nImp0, nExpH = pd->exportCount[half];
	const int inH = pd->inBase[half], outH = pd->outBase[half];
	const int impSlotH = nb + (half ? nImp0 : 0) + ht; // LDS slot of the import this thread receives

	float4* lvel = lds;
	float4* ldq = lds + nt;
	float4* linteg = lds + 2 * nt;							// velocity-integrator constants of every staged body (body_ops.h)
	float* langDamp = (float*)(lds + 3 * nt);				// nt floats, padded to records
	float2* lmass = (float2*)(lds + 3 * nt + (nt + 3) / 4); // {invMass, invI} of every staged body, padded to records
	const int bodyRecords = 3 * nt + (nt + 3) / 4 + (nt + 1) / 2;
	Op* lops = (Op*)(lds + bodyRecords); // 2 records per op

	// ---- loads ----
	uint32_t id[S2_PAIR_BODY_CHUNKS];
#pragma unroll
	for (int ch = 0; ch < S2_PAIR_BODY_CHUNKS; ++ch)
	{
		const int i = tid + ch * S2_PAIR_THREADS;
		id[ch] = i < nb ? (uint32_t)ta.bodyIds[bodyBase + i] : 0u;
	}
	const int impId = ht < nImpH ? pv.importIds[pd->importIdBase[half] + ht] : -1;
	const int expIdx = ht < nExpH ? pv.exportSrc[pd->exportSrcBase[half] + ht] : 0;
	for (int i = tid; i < opCount * 8; i += S2_PAIR_THREADS)
	{
		((int*)lops)[i] = ((const int*)ops)[i];
	}
	// the constraint this lane pair holds in interior round i (batch ranges sit in scalar registers)
	auto kOfRound = [&](int i) {
		const int k = batchA[i].x + ct;
		return (i < roundsA && k < batchA[i].y) ? k : -1;
	};
	HalfRegs rA[ROUNDS];
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		const int k = kOfRound(i);
		if (k >= 0)
		{
			const int2 lb = c.localBodies[k];
			rA[i] = loadHalf<KIND, WARM>(c, k, side, side ? lb.y : lb.x);
		}
	}
	// seam constraints: round r = left seam's batch r followed by right seam's batch r, dealt to the lane pairs in two
	// passes (a round holds at most 512 constraints); where an item lives is recomputed, not stored
	auto seamItem = [&](int r, int pass, int& seam, int& k, uint32_t salt = 0u) {
		const int n0 = r < roundsB0 ? batchB0[r].y - batchB0[r].x : 0;
		const int n1 = r < roundsB1 ? batchB1[r].y - batchB1[r].x : 0;
		const int idx = (int)((uint32_t)(ct + pass * S2_PAIR_SLOTS) ^ salt);
		if (idx < n0)
		{
			seam = 0, k = batchB0[r].x + idx;
			return true;
		}
		if (idx - n0 < n1)
		{
			seam = 1, k = batchB1[r].x + idx - n0;
			return true;
		}
		return false;
	};
	HalfRegs rB[2 * S2_PAIR_SEAM_ROUNDS];
	uint32_t seamMask = 0u; // bit 2 i + pass: this lane pair holds a seam constraint in that round and pass
#pragma unroll
	for (int i = 0; i < S2_PAIR_SEAM_ROUNDS; ++i)
	{
#pragma unroll
		for (int pass = 0; pass < 2; ++pass)
		{
			int seam, k;
			if (i < roundsB && seamItem(i, pass, seam, k))
			{
				const int2 lb = c.localBodies[k];
				rB[2 * i + pass] = loadHalf<KIND, WARM>(c, k, side, pv.remap[pd->remapBase[seam] + (side ? lb.y : lb.x)]);
				seamMask |= 1u << (2 * i + pass);
			}
		}
	}
	// bodies (+ their integrator constants) into LDS: own list, then this half's imports
	uint32_t flags[S2_PAIR_BODY_CHUNKS + 1];
	int ldsIdx[S2_PAIR_BODY_CHUNKS + 1];
#pragma unroll
	for (int ch = 0; ch < S2_PAIR_BODY_CHUNKS + 1; ++ch)
	{
		int gi = -1;
		if (ch < S2_PAIR_BODY_CHUNKS)
		{
			const int i = tid + ch * S2_PAIR_THREADS;
			ldsIdx[ch] = i;
			gi = i < nb ? (int)(id[ch] & ~S2G_OWNED) : -1;
		}
		else
		{
			ldsIdx[ch] = impSlotH;
			gi = impId;
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
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		if (kOfRound(i) >= 0)
		{
			markStaticSide(rA[i], lmass);
		}
	}
#pragma unroll
	for (int i = 0; i < 2 * S2_PAIR_SEAM_ROUNDS; ++i)
	{
		if ((seamMask >> i) & 1u)
		{
			markStaticSide(rB[i], lmass);
		}
	}

	stampAt(1);
	const float4 coef0 = pv.softCoef[0], coef1 = pv.softCoef[1];
	unsigned epoch = 0; // tags are the exchange number: the buffers are zero at launch (cleared by the previous step's epilogue)
	int bad = 0;
	for (int oi = 0; oi < opCount && !bad; ++oi)
	{
		const Op op = lops[oi];
		// an opaque zero produced inside the step loop: without it the compiler hoists the decoding of every round's indices
		// (and the LDS addresses made from them) out of that loop and pays for it in scratch spills
		uint32_t salt;
		asm volatile("s_mov_b32 %0, 0" : "=s"(salt));
		if (op.code == OP_INTEGRATE_VEL)
		{
#pragma unroll
			for (int ch = 0; ch < S2_PAIR_BODY_CHUNKS + 1; ++ch)
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
			stampAt(2);
		}
		else if (op.code == OP_INTEGRATE_POS)
		{
#pragma unroll
			for (int ch = 0; ch < S2_PAIR_BODY_CHUNKS + 1; ++ch)
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
			stampAt(2);
		}
		else if (op.code == OP_FINALIZE)
		{
			// s2FinalizePositions (solve_common.c:70-91; body_ops.h finalizePositionsOne): the owner writes the position,
			// every copy resets its deltaPosition
			const uint32_t need = op.flag ? S2F_DYNAMIC : S2F_MOVES;
#pragma unroll
			for (int ch = 0; ch < S2_PAIR_BODY_CHUNKS + 1; ++ch)
			{
				if ((flags[ch] & need) != 0)
				{
					const int i = ldsIdx[ch];
					const float4 d = ldq[i];
					if (ch < S2_PAIR_BODY_CHUNKS && (id[ch] & S2G_OWNED) != 0)
					{
						const int gi = (int)(id[ch] & ~S2G_OWNED);
						const float2 pos = g.pos[gi];
						const V2 np = add(v2(pos.x, pos.y), v2(d.x, d.y));
						g.pos[gi] = make_float2(np.x, np.y);
					}
					ldq[i] = make_float4(0.0f, 0.0f, d.z, d.w);
				}
			}
			__syncthreads();
			stampAt(2);
		}
		else if (op.code == OP_WARM)
		{
			// s2WarmStartContacts as a coloured sweep WITHOUT an exchange: a side's warm-start term depends on the impulses,
			// the anchors and that body's own pose only, so every body this workgroup owns ends up with the right bits; the
			// copies of the neighbours' bodies are refreshed by the next sweep's exchange before anything reads them
#pragma unroll
			for (int i = 0; i < ROUNDS; ++i)
			{
				if (i < roundsA)
				{
					if (kOfRound(i) >= 0)
					{
						warmPair<WARM, POINTS>(rA[i], lvel, ldq, lmass, su, salt);
					}
					__syncthreads();
				}
			}
#pragma unroll
			for (int i = 0; i < S2_PAIR_SEAM_ROUNDS; ++i)
			{
				if (i < roundsB)
				{
#pragma unroll
					for (int pass = 0; pass < 2; ++pass)
					{
						if ((seamMask >> (2 * i + pass)) & 1u)
						{
							warmPair<WARM, POINTS>(rB[2 * i + pass], lvel, ldq, lmass, su, salt);
						}
					}
					__syncthreads();
				}
			}
			stampAt(3);
		}
		else if (op.code == OP_SOLVE_SOFT)
		{
			// ---- interiors ----
#pragma unroll
			for (int i = 0; i < ROUNDS; ++i)
			{
				if (i < roundsA)
				{
					if (kOfRound(i) >= 0)
					{
						solvePair<KIND, POINTS>(rA[i], lvel, ldq, lmass, coef0, coef1, op.inv_h, op.useBias, su, salt);
					}
					__syncthreads();
				}
			}
			stampAt(4);
			// ---- symmetric exchange of the seam bodies' velocities (poses are replicated by the body stages) ----
			epoch += 1;
			const int par = (int)(epoch & 1u) * pv.parityStride;
			const bool mute = (pv.debugSkip & 8) != 0 && blockIdx.x == 1; // fault injection: this workgroup stays silent
			if (ht < nExpH && !mute)
			{
				const float4 v = lvel[expIdx];
				gu64* p = gran + par + outH + 4 * ht;
				putGranule(p + 0, epoch, v.x), putGranule(p + 1, epoch, v.y), putGranule(p + 2, epoch, v.z);
			}
			int fail = 0;
			if (ht < nImpH)
			{
				float v[3];
				if (getGranules<3>(gran + par + inH + 4 * ht, epoch, v, pv.error, pv.deviceError, pv.spinLimit))
				{
					lvel[impSlotH] = make_float4(v[0], v[1], v[2], 0.0f);
				}
				else
				{
					fail = 1;
				}
			}
			bad = __syncthreads_or(fail);
			if (bad)
			{
				break;
			}
			stampAt(5);
			// ---- both seams (the neighbours compute the same bits on their side) ----
#pragma unroll
			for (int i = 0; i < S2_PAIR_SEAM_ROUNDS; ++i)
			{
				if (i < roundsB)
				{
#pragma unroll
					for (int pass = 0; pass < 2; ++pass)
					{
						if ((seamMask >> (2 * i + pass)) & 1u)
						{
							solvePair<KIND, POINTS>(rB[2 * i + pass], lvel, ldq, lmass, coef0, coef1, op.inv_h, op.useBias, su, salt);
						}
					}
					__syncthreads();
				}
			}
			stampAt(6);
		}
	}

	// ---- results: owned bodies, impulses ----
#pragma unroll
	for (int ch = 0; ch < S2_PAIR_BODY_CHUNKS; ++ch)
	{
		const int i = tid + ch * S2_PAIR_THREADS;
		if (i < nb && (id[ch] & S2G_OWNED) != 0)
		{
			const int gi = (int)(id[ch] & ~S2G_OWNED);
			g.vel[gi] = lvel[i];
			g.dq[gi] = ldq[i];
		}
	}
	if (side == 0)
	{
#pragma unroll
		for (int i = 0; i < ROUNDS; ++i)
		{
			if (kOfRound(i) >= 0)
			{
				storeHalf(c, rA[i], kOfRound(i));
			}
		}
		// the right seam's impulses are stored by this workgroup (its left neighbour of that seam), nobody stores twice
#pragma unroll
		for (int i = 0; i < S2_PAIR_SEAM_ROUNDS; ++i)
		{
#pragma unroll
			for (int pass = 0; pass < 2; ++pass)
			{
				int seam, k;
				if (i < roundsB && seamItem(i, pass, seam, k) && seam == 1)
				{
					storeHalf(c, rB[2 * i + pass], k);
				}
			}
		}
	}
	stampAt(7);
	if (stamp)
	{
		pv.debugTimes[254] = 1ull; // tagged format
		pv.debugTimes[255] = (unsigned long long)stamps;
	}
}

template <int KIND, int WARM>
static void launchPair(hipStream_t s, dim3 grid, size_t lds, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv, const Op* ops,
					   int opCount)
{
	const dim3 block(S2_PAIR_THREADS);
	if (pv.allTwoPoints)
	{
		pairStepKernel<KIND, WARM, 2, S2_STRIP_ROUNDS><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount);
	}
	else
	{
		pairStepKernel<KIND, WARM, 0, S2_STRIP_ROUNDS><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount);
	}
}

// Eligibility (checked by the caller, solver_executor.h runPersistent): pv.pairLanes -- no strip has more than
// S2_STRIP_ROUNDS interior colour batches and no seam more than S2_PAIR_SEAM_ROUNDS.
void launchPairStep(hipStream_t s, int kind, int warm, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv, const Op* ops,
					int opCount)
{
	const dim3 grid((unsigned)a.groupCount);
	const size_t lds = (size_t)pv.ldsRecords * sizeof(float4) + (size_t)opCount * sizeof(Op);
	if (kind == SOFT_TGS)
	{
		warm == WARM_FIXED ? launchPair<SOFT_TGS, WARM_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount)
						   : launchPair<SOFT_TGS, WARM_CURRENT>(s, grid, lds, c, g, a, pv, ops, opCount);
	}
	else if (kind == SOFT_PGS)
	{
		warm == WARM_FIXED ? launchPair<SOFT_PGS, WARM_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount)
						   : launchPair<SOFT_PGS, WARM_CURRENT>(s, grid, lds, c, g, a, pv, ops, opCount);
	}
	else
	{
		warm == WARM_FIXED ? launchPair<SOFT_FIXED, WARM_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount)
						   : launchPair<SOFT_FIXED, WARM_CURRENT>(s, grid, lds, c, g, a, pv, ops, opCount);
	}
}

int pairKernelSetup()
{
	const void* steps[] = {
#define S2_PAIR_VARIANTS(K, W) (const void*)pairStepKernel<K, W, 0, S2_STRIP_ROUNDS>, (const void*)pairStepKernel<K, W, 2, S2_STRIP_ROUNDS>
		S2_PAIR_VARIANTS(SOFT_TGS, WARM_CURRENT),	S2_PAIR_VARIANTS(SOFT_TGS, WARM_FIXED),	  S2_PAIR_VARIANTS(SOFT_PGS, WARM_CURRENT),
		S2_PAIR_VARIANTS(SOFT_PGS, WARM_FIXED),		S2_PAIR_VARIANTS(SOFT_FIXED, WARM_CURRENT), S2_PAIR_VARIANTS(SOFT_FIXED, WARM_FIXED),
#undef S2_PAIR_VARIANTS
	};
	for (const void* f : steps)
	{
		hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		if (e != hipSuccess)
		{
			return (int)e;
		}
	}
	return 0;
}

S2_DEFINE_WARM(pair_kernel)
