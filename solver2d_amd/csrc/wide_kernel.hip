// Persistent strip step of s2Solve_TGS_Soft, "wide" form: 512 threads per strip, one lane per constraint.
//
// The same step as strip_kernel.hip: stripStepKernel<SOFT_TGS, WARM_CURRENT, ., ., SEAMREG> -- workgroup i owns strip i of a
// big island for the WHOLE s2Solve_TGS_Soft (src/solve_tgs_soft.c:138-280), interior and seam constraints resident in
// registers, bodies in LDS, seam bodies exchanged with the two neighbouring workgroups once per sweep as tagged granules --
// on the same tables (StripDesc, PersistDesc), in the same sweep order, to the same bits.  What it does with what round 3
// measured on this part (tools/valu_bench.hip):
//
//  * a wave that is alone on its SIMD issues ONE instruction per 4 cycles whatever the instruction is, and a packed fp32
//    instruction (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations, each rounded separately) costs that wave about what a
//    scalar one does.  A colour round is one such wave's instruction stream, so the round's arithmetic is written as
//    2-vector operations on (x, y) pairs -- rotate, anchor velocity, impulse application are 2-vector algebra -- with the
//    operands laid out so that the pairs come out of the LDS records and the resident constraint record as aligned
//    register pairs (no packing moves): `solveWide` / `warmWide` below.  Same operations on the same operands as
//    constraint_ops.h solveSoftRegs (s2SolveContacts_TGS_Soft, solve_tgs_soft.c:17-135), which the parity tests pin.
//  * a second wave on a SIMD runs beside the first at nearly full speed.  With 512 threads a seam round -- both seams of the
//    strip, up to 512 constraints -- is ONE pass (two waves per SIMD) instead of two passes of one wave per SIMD, the body
//    stages touch every staged body in one go, and the two halves of the workgroup serve one neighbour each in the hand-off.
//    Interior round i runs on half i & 1 of the workgroup, so a lane holds three interior and two seam records of 22 dwords:
//    110 of its 256 registers, no spills, no AGPR copies.
//  * the three soft coefficients (one of two step-wide triples) are looked up in LDS by the constraint's "a side is static"
//    bit instead of being selected from six scalar registers.
//
// Two lanes per constraint (pair_kernel.hip) was built and measured first: bit-exact, and no faster than one lane (a DPP
// operand costs two issue slots on this part and the impulse chain is duplicated in both lanes).

#include "body_ops.h"
#include "persist_handoff.h"
#include "soft_from_wire.h"

#ifndef S2_PERSIST_INSTRUMENTED
#define S2_PERSIST_INSTRUMENTED 0
#endif
// 1: strips are dealt to the workgroups so that neighbours share an XCD, and a seam inside an XCD hands its bodies over through
// that XCD's L2 (experiment switch)
#ifndef S2_WIDE_XCD_AFFINE
#define S2_WIDE_XCD_AFFINE 1
#endif
#define S2_WIDE_THREADS 512
#define S2_WIDE_INTERIOR 256  // a colour batch of a strip has at most 256 constraints: interior round i runs on half i & 1 of the workgroup
#define S2_WIDE_ROUNDS_PER_HALF (S2_STRIP_ROUNDS / 2) // ... so a lane holds three interior records
#define S2_WIDE_BODY_CHUNKS 2 // own bodies per thread: a strip stages at most 2 * 512 = S2_STRIP_BODY_CHUNKS * 256
#define S2_WIDE_SEAM_ROUNDS 2

typedef float f2 __attribute__((ext_vector_type(2)));

S2_DEV f2 lo2(float4 v) { return f2{v.x, v.y}; }
S2_DEV f2 hi2(float4 v) { return f2{v.z, v.w}; }
// A constraint as a lane keeps it for the whole step: 22 dwords.
struct WideRegs
{
	uint32_t idx; // ia | ib << 13 | pointCount << 26 | writeA << 28 | writeB << 29 | (a side is static) << 30
	f2 n;		  // normal
	float friction;
	f2 lA[2], lB[2]; // local anchors relative to the centres of mass
	float p0[2], p1[2], p2[2]; // adjustedSeparation, normalMass, tangentMass
	f2 imp[2];				   // {normal, tangent} impulse
};

// KIND: SOFT_TGS (s2SolveContacts_TGS_Soft, solve_tgs_soft.c:17-135: anchors in the bodies' frames, turned and re-measured every sweep) or
// SOFT_PGS (s2SolveContacts_PGS_Soft, solve_pgs_soft.c:16-125: the anchors rA0 / rB0 and the separation as s2PrepareContacts_Soft left them).
// The record has the same 22 dwords either way: for SOFT_PGS lA / lB hold rA0 / rB0 (world orientation) and p0 the separation.
// (inside the kernel) SOFT_PGS on a variant without parked rounds: rA0 / rB0 wait in LDS as for SOFT_FIXED -- the record is 14 dwords
#define S2_WIDE_PGS_ARMS 100
template <int KIND> constexpr bool wideIsPgs = KIND == SOFT_PGS || KIND == S2_WIDE_PGS_ARMS;
template <int KIND> constexpr bool wideLdsArms = KIND == SOFT_FIXED || KIND == S2_WIDE_PGS_ARMS;
template <int KIND> S2_DEV WideRegs loadWide(const ContactView& c, int k, int ia, int ib)
{
	WideRegs p;
	const float4 nf = c.nf[k];
	const uint32_t bits = asBits(nf.w);
	p.idx = (uint32_t)ia | ((uint32_t)ib << 13) | ((bits & 3u) << 26) | ((bits & S2C_WRITE_A) ? 1u << 28 : 0u) | ((bits & S2C_WRITE_B) ? 1u << 29 : 0u);
	p.n = lo2(nf), p.friction = nf.z;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if constexpr (KIND == S2_WIDE_PGS_ARMS)
		{
			p.lA[j] = p.lB[j] = f2{0.0f, 0.0f}; // (never read: the anchors are in LDS)
		}
		else
		{
			const float4 a = KIND == SOFT_PGS ? c.r0[j][k] : c.anchor[j][k];
			// (SOFT_PGS keeps perp(rA0) = (-y, x), the form the sweep multiplies with: nothing about the anchors is left to compute per sweep)
			p.lA[j] = KIND == SOFT_PGS ? f2{-a.y, a.x} : lo2(a), p.lB[j] = KIND == SOFT_PGS ? f2{-a.w, a.z} : hi2(a);
		}
		const float4 par = c.param[j][k];
		p.p0[j] = wideIsPgs<KIND> ? par.w : par.x, p.p1[j] = par.y, p.p2[j] = par.z;
		const float2 imp = c.impulse[j][k];
		p.imp[j] = f2{imp.x, imp.y};
	}
	return p;
}

S2_DEV V2 asV2(f2 v) { return v2(v.x, v.y); }
// rotate (math.h:330-341) on 2-vectors: q = {s, c}, qn = {-s, c}: (c x + (-s) y, s x + c y) -- (-s) y == -(s y) and
// a - b == a + (-b), so these are the reference's bits
S2_DEV f2 rot2(f2 q, f2 qn, f2 l)
{
	return q.yx * l.xx + qn * l.yy;
}

// s2WarmStartContacts (solve_common.c:276-330): strip_kernel.hip warmSoftRegs
// The warm start and the prep on 2-vectors both measured SLOWER (178 / 171 vs 131 us per launch: their even-aligned
// register pairs push resident constraint registers into scratch on the hand-off path -- 84 spilled VGPRs instead of ~30);
// only the chain (chainWide) is packed.  Kept for the A/B (tools/kernel_ab.sh).
#ifndef S2_WIDE_PACKED_WARM
#define S2_WIDE_PACKED_WARM 0
#endif
#ifndef S2_WIDE_PACKED_PREP
#define S2_WIDE_PACKED_PREP 0
#endif
#if S2_WIDE_PACKED_WARM
template <int KIND, int POINTS> S2_DEV void warmWide(const WideRegs& p, float4* lvel, const float4* ldq, const float2* lmass, uint32_t salt)
{
	const uint32_t idx = p.idx ^ salt;
	const int ia = (int)(idx & 0x1fffu), ib = (int)((idx >> 13) & 0x1fffu);
	const int pointCount = (int)((idx >> 26) & 3u);
	const float4 velA = lvel[ia], velB = lvel[ib];
	const f2 qA = hi2(ldq[ia]), qB = hi2(ldq[ib]);
	const float2 mA = lmass[ia], mB = lmass[ib];
	const f2 n = f2{fromBits(asBits(p.n.x) ^ salt), p.n.y};
	const f2 t = f2{n.y, -n.x};
	const f2 qnA = f2{-qA.x, qA.y}, qnB = f2{-qB.x, qB.y};
	const f2 nmA2 = f2{-mA.x, -mA.x}, mB2 = f2{mB.x, mB.x};
	f2 vA = lo2(velA), vB = lo2(velB);
	float wA = velA.z, wB = velB.z;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			const f2 rA = rot2(qA, qnA, p.lA[j]), rB = rot2(qB, qnB, p.lB[j]);
			const f2 P = p.imp[j].xx * n + p.imp[j].yy * t; // add(mulSV(normalImpulse, normal), mulSV(tangentImpulse, tangent))
			const f2 cA = rA * P.yx, cB = rB * P.yx;		// cross(r, P) = r.x P.y - r.y P.x
			wA -= mA.y * (cA.x - cA.y);
			vA = vA + nmA2 * P; // mulAdd(vA, -mA, P)
			wB += mB.y * (cB.x - cB.y);
			vB = vB + mB2 * P;
		}
	}
	if ((idx & (1u << 28)) != 0)
	{
		lvel[ia] = make_float4(vA.x, vA.y, wA, 0.0f);
	}
	if ((idx & (1u << 29)) != 0)
	{
		lvel[ib] = make_float4(vB.x, vB.y, wB, 0.0f);
	}
}
#else
// LL: the record's local anchors wait in LDS ({lA, lB} per manifold point and lane: `locals`), not in p.lA / p.lB -- the variants that hold
// a sixth resident record, or resident records beside parked rounds (wideLocalsInLds): the registers the record gives up are what
// those variants used to spill
template <int KIND, int POINTS, bool LL = false>
S2_DEV void warmWide(const WideRegs& p, float4* lvel, const float4* ldq, const float2* lmass, uint32_t salt, const float4* arms = nullptr, const float4* locals = nullptr)
{
	const uint32_t idx = p.idx ^ salt;
	const int ia = (int)(idx & 0x1fffu), ib = (int)((idx >> 13) & 0x1fffu);
	const int pointCount = (int)((idx >> 26) & 3u);
	const float4 velA = lvel[ia], velB = lvel[ib];
	const float4 dqA = ldq[ia], dqB = ldq[ib];
	const float2 mA = lmass[ia], mB = lmass[ib];
	const V2 normal = v2(fromBits(asBits(p.n.x) ^ salt), p.n.y);
	const V2 tangent = rightPerp(normal);
	Rot qA, qB;
	qA.s = dqA.z, qA.c = dqA.w, qB.s = dqB.z, qB.c = dqB.w;
	V2 vA = v2(velA.x, velA.y), vB = v2(velB.x, velB.y);
	float wA = velA.z, wB = velB.z;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			// (SOFT_PGS: the anchors in world orientation as s2PrepareContacts_Soft computed them -- rotate(q, local anchor) on the poses the
			// warm start sees too, the same operation on the same operands: rA0 IS the current anchor)
			V2 rA, rB;
			if constexpr (wideLdsArms<KIND>)
			{
				// s2WarmStartContacts_Fixed (solve_soft_step.c:20-64): rA0 / rB0, kept as perp in LDS (wideStepKernel: larms)
				const float4 a = arms[j * S2_WIDE_THREADS];
				rA = v2(a.y, -a.x), rB = v2(a.w, -a.z);
			}
			else if constexpr (KIND == SOFT_PGS)
			{
				rA = v2(p.lA[j].y, -p.lA[j].x), rB = v2(p.lB[j].y, -p.lB[j].x);
			}
			else if constexpr (LL)
			{
				const float4 a = locals[j * S2_WIDE_THREADS];
				rA = rotate(qA, v2(a.x, a.y)), rB = rotate(qB, v2(a.z, a.w));
			}
			else
			{
				rA = rotate(qA, asV2(p.lA[j])), rB = rotate(qB, asV2(p.lB[j]));
			}
			const V2 P = add(mulSV(p.imp[j].x, normal), mulSV(p.imp[j].y, tangent));
			wA -= mA.y * cross(rA, P);
			vA = mulAdd(vA, -mA.x, P);
			wB += mB.y * cross(rB, P);
			vB = mulAdd(vB, mB.x, P);
		}
	}
	if ((idx & (1u << 28)) != 0)
	{
		lvel[ia] = make_float4(vA.x, vA.y, wA, 0.0f);
	}
	if ((idx & (1u << 29)) != 0)
	{
		lvel[ib] = make_float4(vB.x, vB.y, wB, 0.0f);
	}
}
#endif

// s2SolveContacts_TGS_Soft (solve_tgs_soft.c:17-135) = constraint_ops.h solveSoftRegs<SOFT_TGS>, operation for operation, in two
// parts (the split of constraint_ops.h prepSoft / chainSoft):
//   prepWide  what depends on the POSES only, which no sweep changes: the anchors in world orientation, the current separation,
//             the bias and whether the soft coefficients apply.  It can run any time after the last s2IntegratePositions --
//             the kernel runs it in lanes that would otherwise wait (the idle half of the workgroup during an interior round,
//             every lane while the seam bodies are in flight);
//   chainWide what depends on the VELOCITIES: relative velocity -> impulse -> clamp -> apply, point after point, normal then
//             friction.  This is what a colour round has to wait for.
// 1: the chain on explicit 2-vectors (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per issue slot, each rounded on its own)
#ifndef S2_WIDE_PACKED
#define S2_WIDE_PACKED 1
#endif
struct WidePrep
{
#if S2_WIDE_PACKED
	// the anchors as the chain wants them: perp(r) = (-r.y, r.x).  crossSV(w, r) = w perp(r) and cross(r, P) = perp(r).x P.x +
	// perp(r).y P.y, term for term the reference's products ((-a) b == -(a b), x - y == x + (-y)), so the chain needs no
	// per-component sign and its 2-vector algebra packs without moves
	f2 pA[2], pB[2];
#else
	V2 rA[2], rB[2];
#endif
	float bias[2];
	uint32_t soft; // bit j: point j takes the soft mass / impulse scales
};

template <int KIND, int POINTS, bool LL = false>
S2_DEV WidePrep prepWide(const WideRegs& p, const float4* ldq, const float4* lcoef, float inv_h, int useBias, uint32_t salt, const float4* arms = nullptr, const float4* locals = nullptr)
{
	const uint32_t idx = p.idx ^ salt;
	const int ia = (int)(idx & 0x1fffu), ib = (int)((idx >> 13) & 0x1fffu);
	const int pointCount = (int)((idx >> 26) & 3u);
	const float4 dqA = ldq[ia], dqB = ldq[ib];
	const float biasCoefficient = lcoef[(idx >> 30) & 1u].x;
	const V2 normal = v2(fromBits(asBits(p.n.x) ^ salt), p.n.y);
	const V2 dcA = v2(dqA.x, dqA.y), dcB = v2(dqB.x, dqB.y);
	Rot qA, qB;
	qA.s = dqA.z, qA.c = dqA.w, qB.s = dqB.z, qB.c = dqB.w;
#if S2_WIDE_PACKED && S2_WIDE_PACKED_PREP
	const f2 q2A = hi2(dqA), q2B = hi2(dqB);
	const f2 qnA = f2{-q2A.x, q2A.y}, qnB = f2{-q2B.x, q2B.y};
	const f2 dd2 = lo2(dqB) - lo2(dqA);
	const f2 n2 = f2{normal.x, normal.y};
#endif
	WidePrep pre;
	pre.soft = 0u;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
#if S2_WIDE_PACKED
		pre.pA[j] = pre.pB[j] = f2{0.0f, 0.0f};
#else
		pre.rA[j] = pre.rB[j] = v2(0.0f, 0.0f);
#endif
		pre.bias[j] = 0.0f;
		if (POINTS == 2 || j < pointCount)
		{
#if S2_WIDE_PACKED && S2_WIDE_PACKED_PREP
			const f2 rA = rot2(q2A, qnA, p.lA[j]), rB = rot2(q2B, qnB, p.lB[j]);
			pre.pA[j] = f2{-rA.y, rA.x}, pre.pB[j] = f2{-rB.y, rB.x};
			const f2 ds2 = dd2 + (rB - rA); // add(sub(dcB, dcA), sub(rB, rA))
			const f2 sn = ds2 * n2;
			const float s = (sn.x + sn.y) + p.p0[j];
#else
			V2 lAj, lBj;
			if constexpr (LL)
			{
				const float4 a = locals[j * S2_WIDE_THREADS];
				lAj = v2(a.x, a.y), lBj = v2(a.z, a.w);
			}
			else
			{
				lAj = asV2(p.lA[j]), lBj = asV2(p.lB[j]);
			}
			const V2 rA = wideIsPgs<KIND> ? v2(0.0f, 0.0f) : rotate(qA, lAj), rB = wideIsPgs<KIND> ? v2(0.0f, 0.0f) : rotate(qB, lBj);
#if S2_WIDE_PACKED
			if constexpr (wideLdsArms<KIND>)
			{
				// s2SolveContacts_TGS_Fixed (solve_soft_step.c:66-177): the separation from the anchors as the bodies stand now (below),
				// the impulses along rA0 / rB0 -- perp(rA0), perp(rB0) wait in LDS, the form the chain multiplies with
				const float4 a = arms[j * S2_WIDE_THREADS];
				pre.pA[j] = lo2(a), pre.pB[j] = hi2(a);
			}
			else
			{
				pre.pA[j] = KIND == SOFT_PGS ? p.lA[j] : f2{-rA.y, rA.x}, pre.pB[j] = KIND == SOFT_PGS ? p.lB[j] : f2{-rB.y, rB.x};
			}
#else
			static_assert(KIND == SOFT_TGS, "SOFT_PGS / SOFT_FIXED keep the packed chain's anchors");
			pre.rA[j] = rA, pre.rB[j] = rB;
#endif
			const V2 ds = add(sub(dcB, dcA), sub(rB, rA));
			const float s = wideIsPgs<KIND> ? p.p0[j] : dot(ds, normal) + p.p0[j];
#endif
			// select form of: if (s > 0) bias = s * inv_h; else if (useBias) {bias = max(biasCoefficient * s, cap); ...}
			const bool speculative = s > 0.0f;
			const bool soft = !speculative && useBias != 0;
			const float softBias = S2_MAXF(biasCoefficient * s, KIND == SOFT_TGS ? -S2_MAX_BAUMGARTE_VELOCITY : -0.5f * S2_MAX_BAUMGARTE_VELOCITY);
			pre.bias[j] = speculative ? s * inv_h : (soft ? softBias : 0.0f);
			pre.soft |= soft ? 1u << j : 0u;
		}
	}
	return pre;
}

#if S2_WIDE_PACKED
S2_DEV float dot2(f2 a, f2 b) // a.x b.x + a.y b.y
{
	const f2 m = a * b;
	return m.x + m.y;
}
S2_DEV float crossP(f2 perpR, f2 P) // cross(r, P) = r.x P.y - r.y P.x = perp(r).y P.y + perp(r).x P.x
{
	const f2 m = perpR * P;
	return m.y + m.x;
}

template <int POINTS> S2_DEV void chainWide(WideRegs& p, const WidePrep& pre, float4* lvel, const float2* lmass, const float4* lcoef, uint32_t salt)
{
	const uint32_t idx = p.idx ^ salt;
	const int ia = (int)(idx & 0x1fffu), ib = (int)((idx >> 13) & 0x1fffu);
	const int pointCount = (int)((idx >> 26) & 3u);
	const float4 velA = lvel[ia], velB = lvel[ib];
	const float2 massA = lmass[ia], massB = lmass[ib];
	const float4 sf = lcoef[(idx >> 30) & 1u];
	const f2 n = f2{fromBits(asBits(p.n.x) ^ salt), p.n.y};
	const f2 t = f2{n.y, -n.x}; // rightPerp
	const f2 mA2 = f2{massA.x, massA.x}, mB2 = f2{massB.x, massB.x};
	const float iA = massA.y, iB = massB.y;
	f2 vA = lo2(velA), vB = lo2(velB);
	float wA = velA.z, wB = velB.z;
	float nImp[2], tImp[2];

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			const f2 pA = pre.pA[j], pB = pre.pB[j];
			const bool soft = (pre.soft >> j) & 1u;
			const float massScale = soft ? sf.y : 1.0f;
			const float impulseScale = soft ? sf.z : 0.0f;

			// add(v, crossSV(w, r)) = v + w perp(r); sub(vrB, vrA); dot with the normal
			const f2 vrB = vB + f2{wB, wB} * pB;
			const f2 vrA = vA + f2{wA, wA} * pA;
			const float vn = dot2(vrB - vrA, n);

			const float normalMass = fromBits(asBits(p.p1[j]) ^ salt);
			const float old = p.imp[j].x;
			float impulse = -normalMass * massScale * (vn + pre.bias[j]) - impulseScale * old;
			const float newImpulse = S2_MAXF(old + impulse, 0.0f);
			impulse = newImpulse - old;
			nImp[j] = newImpulse;
			tImp[j] = p.imp[j].y;

			const f2 P = f2{impulse, impulse} * n; // mulSV
			vA = vA - mA2 * P;					   // mulSub
			wA -= iA * crossP(pA, P);
			vB = vB + mB2 * P; // mulAdd
			wB += iB * crossP(pB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			const float tangentMass = fromBits(asBits(p.p2[j]) ^ salt);
			const f2 pA = pre.pA[j], pB = pre.pB[j];
			const f2 vrB = vB + f2{wB, wB} * pB;
			const f2 vrA = vA + f2{wA, wA} * pA;
			const float vt = dot2(vrB - vrA, t);
			float impulse = -tangentMass * vt;
			const float maxFriction = p.friction * nImp[j];
			const float newImpulse = S2_CLAMPF(tImp[j] + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - tImp[j];
			const f2 P = f2{impulse, impulse} * t;
			vA = vA - mA2 * P;
			wA -= iA * crossP(pA, P);
			vB = vB + mB2 * P;
			wB += iB * crossP(pB, P);
			p.imp[j] = f2{nImp[j], newImpulse};
		}
	}

	if ((idx & (1u << 28)) != 0)
	{
		lvel[ia] = make_float4(vA.x, vA.y, wA, 0.0f);
	}
	if ((idx & (1u << 29)) != 0)
	{
		lvel[ib] = make_float4(vB.x, vB.y, wB, 0.0f);
	}
}
#else
template <int POINTS> S2_DEV void chainWide(WideRegs& p, const WidePrep& pre, float4* lvel, const float2* lmass, const float4* lcoef, uint32_t salt)
{
	const uint32_t idx = p.idx ^ salt;
	const int ia = (int)(idx & 0x1fffu), ib = (int)((idx >> 13) & 0x1fffu);
	const int pointCount = (int)((idx >> 26) & 3u);
	const float4 velA = lvel[ia], velB = lvel[ib];
	const float2 massA = lmass[ia], massB = lmass[ib];
	const float4 sf = lcoef[(idx >> 30) & 1u];
	const V2 normal = v2(fromBits(asBits(p.n.x) ^ salt), p.n.y);
	const V2 tangent = rightPerp(normal);
	const float mA = massA.x, iA = massA.y, mB = massB.x, iB = massB.y;
	V2 vA = v2(velA.x, velA.y), vB = v2(velB.x, velB.y);
	float wA = velA.z, wB = velB.z;
	float nImp[2], tImp[2];

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			const V2 rA = pre.rA[j], rB = pre.rB[j];
			const bool soft = (pre.soft >> j) & 1u;
			const float massScale = soft ? sf.y : 1.0f;
			const float impulseScale = soft ? sf.z : 0.0f;

			const V2 vrB = add(vB, crossSV(wB, rB));
			const V2 vrA = add(vA, crossSV(wA, rA));
			const float vn = dot(sub(vrB, vrA), normal);

			const float normalMass = fromBits(asBits(p.p1[j]) ^ salt);
			const float old = p.imp[j].x;
			float impulse = -normalMass * massScale * (vn + pre.bias[j]) - impulseScale * old;
			const float newImpulse = S2_MAXF(old + impulse, 0.0f);
			impulse = newImpulse - old;
			nImp[j] = newImpulse;
			tImp[j] = p.imp[j].y;

			const V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < pointCount)
		{
			const float tangentMass = fromBits(asBits(p.p2[j]) ^ salt);
			const V2 rA = pre.rA[j], rB = pre.rB[j];
			const V2 vrB = add(vB, crossSV(wB, rB));
			const V2 vrA = add(vA, crossSV(wA, rA));
			const float vt = dot(sub(vrB, vrA), tangent);
			float impulse = -tangentMass * vt;
			const float maxFriction = p.friction * nImp[j];
			const float newImpulse = S2_CLAMPF(tImp[j] + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - tImp[j];
			const V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
			p.imp[j] = f2{nImp[j], newImpulse};
		}
	}

	if ((idx & (1u << 28)) != 0)
	{
		lvel[ia] = make_float4(vA.x, vA.y, wA, 0.0f);
	}
	if ((idx & (1u << 29)) != 0)
	{
		lvel[ib] = make_float4(vB.x, vB.y, wB, 0.0f);
	}
}
#endif

S2_DEV void storeWide(const ContactView& c, const WideRegs& p, int k)
{
	const int pointCount = (int)((p.idx >> 26) & 3u);
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			c.impulse[j][k] = make_float2(p.imp[j].x, p.imp[j].y);
		}
	}
}

// A seam record parked in LDS (seam rounds beyond the SR a lane keeps in registers: a partition whose seams need a third or fourth
// colour next to seven or eight interior ones): six 16-byte records per lane and round, field-major so that a wave reads
// consecutive addresses; the impulses are a record of their own -- the only part a sweep writes back.
#define S2_WIDE_PARKED_RECORDS 6
// 1: the preps of the parked seam rounds run while the hand-off is in flight, like those of the rounds in registers.  Measured slower
// on the wreck world (0.289 against 0.271 ms per churn step: 22 more live registers, 148-168 bytes of scratch instead of ~110): off
#ifndef S2_WIDE_PARKED_PREP_EARLY
#define S2_WIDE_PARKED_PREP_EARLY 0
#endif
// (`stride`: the lanes of a parked round -- as many columns as its widest instance in any strip holds constraints: PersistView::parkSeamWidth / parkInteriorWidth)
S2_DEV void parkWide(float4* slot, int stride, const WideRegs& p)
{
	slot[0 * stride] = make_float4(fromBits(p.idx), p.n.x, p.n.y, p.friction);
	slot[1 * stride] = make_float4(p.lA[0].x, p.lA[0].y, p.lA[1].x, p.lA[1].y);
	slot[2 * stride] = make_float4(p.lB[0].x, p.lB[0].y, p.lB[1].x, p.lB[1].y);
	slot[3 * stride] = make_float4(p.p0[0], p.p0[1], p.p1[0], p.p1[1]);
	slot[4 * stride] = make_float4(p.p2[0], p.p2[1], 0.0f, 0.0f);
	slot[5 * stride] = make_float4(p.imp[0].x, p.imp[0].y, p.imp[1].x, p.imp[1].y);
}
S2_DEV WideRegs unparkWide(const float4* slot, int stride)
{
	const float4 q0 = slot[0 * stride], q1 = slot[1 * stride], q2 = slot[2 * stride], q3 = slot[3 * stride], q4 = slot[4 * stride], q5 = slot[5 * stride];
	WideRegs p;
	p.idx = asBits(q0.x), p.n = f2{q0.y, q0.z}, p.friction = q0.w;
	p.lA[0] = lo2(q1), p.lA[1] = hi2(q1), p.lB[0] = lo2(q2), p.lB[1] = hi2(q2);
	p.p0[0] = q3.x, p.p0[1] = q3.y, p.p1[0] = q3.z, p.p1[1] = q3.w, p.p2[0] = q4.x, p.p2[1] = q4.y;
	p.imp[0] = lo2(q5), p.imp[1] = hi2(q5);
	return p;
}

S2_DEV WideRegs wideFromSoft(const SoftRegs<SOFT_TGS>& t)
{
	WideRegs p;
	p.idx = (uint32_t)t.h.ia | ((uint32_t)t.h.ib << 13) | (((uint32_t)t.h.pointCount & 3u) << 26) | (t.h.writeA ? 1u << 28 : 0u) | (t.h.writeB ? 1u << 29 : 0u);
	p.n = f2{t.h.normal.x, t.h.normal.y}, p.friction = t.h.friction;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		p.lA[j] = lo2(t.an[j]), p.lB[j] = hi2(t.an[j]);
		p.p0[j] = t.par[j].x, p.p1[j] = t.par[j].y, p.p2[j] = t.par[j].z;
		p.imp[j] = f2{t.imp[j].x, t.imp[j].y};
	}
	return p;
}

// MODE bits
#define S2_WIDE_SELF 1 // the kernel is the step's prologue and epilogue too (Executor::selfContainedStrips): ONE launch per step
#define S2_WIDE_BODYWARM 2 // s2WarmStartContacts as a body-centric pass over per-constraint terms in LDS (no parked rounds)
#define S2_WIDE_SLICED 4 // a launch of a SLICED step (Executor::runPersistentSliced: one launch per sweep): the kernel zeroes the hand-off buffers it read at its
						 // end.  A variant of its own so that the one-launch step's kernel is the code it was (as a run-time branch the five more live values
						 // cost the headline 1.5 us per launch: 130.5 -> 132.2, measured)

// s2WarmStartContacts (solve_common.c:276-330), body-centric.  The warm start adds, per constraint and manifold point, a term to each
// of its bodies that depends on the impulses, the anchors and that body's own pose only -- not on any velocity.  So instead of one
// colour round per colour (seven barriers of ~0.4 us), every lane writes the terms of the constraints it holds into an LDS table
// indexed [component][round][body] and every body then adds its terms in round order: the same additions in the same order as the
// coloured sweep (v + (-mA) P.x is rounded as the reference's mulAdd; w - x == w + (-x)), two barriers.  Terms per side and point:
// {dv.x, dv.y, dw}; a round's two points are three float2 records {dv0}, {dw0, dv1.x}, {dv1.y, dw1}.
template <int POINTS, bool LL = false>
S2_DEV void warmTermsWide(const WideRegs& p, const float4* ldq, const float2* lmass, float2* lt, int tw, int R, int round, int nOwn, uint32_t salt, const float4* locals = nullptr)
{
	const uint32_t idx = p.idx ^ salt;
	const int ia = (int)(idx & 0x1fffu), ib = (int)((idx >> 13) & 0x1fffu);
	const int pointCount = (int)((idx >> 26) & 3u);
	const bool wa = (idx & (1u << 28)) != 0 && ia < nOwn, wb = (idx & (1u << 29)) != 0 && ib < nOwn;
	const float4 dqA = ldq[ia], dqB = ldq[ib];
	const float2 mA = lmass[ia], mB = lmass[ib];
	const V2 normal = v2(fromBits(asBits(p.n.x) ^ salt), p.n.y);
	const V2 tangent = rightPerp(normal);
	Rot qA, qB;
	qA.s = dqA.z, qA.c = dqA.w, qB.s = dqB.z, qB.c = dqB.w;
	float tA[6], tB[6];
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		tA[3 * j] = tA[3 * j + 1] = tA[3 * j + 2] = 0.0f;
		tB[3 * j] = tB[3 * j + 1] = tB[3 * j + 2] = 0.0f;
		if (POINTS == 2 || j < pointCount)
		{
			V2 lAj, lBj;
			if constexpr (LL)
			{
				const float4 a = locals[j * S2_WIDE_THREADS];
				lAj = v2(a.x, a.y), lBj = v2(a.z, a.w);
			}
			else
			{
				lAj = asV2(p.lA[j]), lBj = asV2(p.lB[j]);
			}
			const V2 rA = rotate(qA, lAj), rB = rotate(qB, lBj);
			const V2 P = add(mulSV(p.imp[j].x, normal), mulSV(p.imp[j].y, tangent));
			// wA -= iA * cross(rA, P); vA = mulAdd(vA, -mA, P); wB += iB * cross(rB, P); vB = mulAdd(vB, mB, P)
			tA[3 * j] = -mA.x * P.x, tA[3 * j + 1] = -mA.x * P.y, tA[3 * j + 2] = -(mA.y * cross(rA, P));
			tB[3 * j] = mB.x * P.x, tB[3 * j + 1] = mB.x * P.y, tB[3 * j + 2] = mB.y * cross(rB, P);
		}
	}
	float2* row = lt + (size_t)round * tw;
	const int plane = R * tw;
	if (wa)
	{
		row[ia] = make_float2(tA[0], tA[1]);
		row[plane + ia] = make_float2(tA[2], tA[3]);
		row[2 * plane + ia] = make_float2(tA[4], tA[5]);
	}
	if (wb)
	{
		row[ib] = make_float2(tB[0], tB[1]);
		row[plane + ib] = make_float2(tB[2], tB[3]);
		row[2 * plane + ib] = make_float2(tB[4], tB[5]);
	}
}

// Resident records of a lane whose local anchors wait in LDS instead of registers (s2Solve_TGS_Soft's variants; warmWide / prepWide: LL).
// The <3, 2> layout -- five records of 22 dwords -- fits the 256 registers of a lane beside a round's working set; a sixth record
// (<3, 3>, <4, 2>), or a parked round's record coming through the registers beside the five, does not: those variants spilled 40-520
// bytes per lane to scratch through round 4.  Eight dwords per record move to LDS (16 KB per record and workgroup) -- read once per
// sweep by the prep, in lanes that wait anyway.  How many: what took each variant below the register file (make resources).
constexpr int wideLocalsInLds(int RPH, int SR, int SL, int IL, bool sliced)
{
	const int plain = (SL > 0 || IL > 0) ? (IL > 0 ? 4 : 3) : ((RPH + SR > 5) ? 3 : 0);
	return plain > 0 && sliced ? plain + 1 : plain; // (a sliced step's launch keeps its inbox addresses live to its end: one more record's worth)
}

// POINTS == 2: the host has checked that every constraint of the strips has two manifold points: no per-point masking.
// RPH: interior records a lane keeps (colour batches / 2), SR: seam records a lane keeps, SL: seam rounds parked in LDS.
// IL: interior rounds parked in LDS behind the 2 RPH a lane keeps (a strip that needs a seventh or eighth colour: a hub body inside it).
// MODE: S2_WIDE_SELF | S2_WIDE_BODYWARM.
// A strip's side of the exchange with the overflow workgroup behind a sweep op (every lane has passed the sweep's last barrier): lovf[m]
// says where this workgroup stages overflow body m -- (priority << 24) | LDS slot, 0: not here; priority 3: it owns the body and
// publishes it, every staged copy takes the result.  Returns non-zero when a hand-off timed out.
S2_DEV int wideOverflowExchange(const PersistView& pv, const int* lovf, float4* lvel, int ovCount, unsigned sweep)
{
	const int tid = (int)threadIdx.x;
	int fail = 0;
	if (tid < ovCount)
	{
		const int e = lovf[tid];
		if (e != 0)
		{
			const int slot = e & 0xffffff;
			gu64* in = (gu64*)pv.granules + pv.overflowGranBase + (int)(sweep & (S2_OVERFLOW_RING - 1)) * S2_OVERFLOW_BODIES * 4 + 4 * tid;
			gu64* out = in + S2_OVERFLOW_GRANULES / 2;
			if ((e >> 24) == 3)
			{
				const float4 v = lvel[slot];
				putGranule(in + 0, sweep, v.x), putGranule(in + 1, sweep, v.y), putGranule(in + 2, sweep, v.z);
			}
			float v[3];
			if (getGranules<3>(out, sweep, v, pv.error, pv.deviceError, pv.spinLimit))
			{
				lvel[slot] = make_float4(v[0], v[1], v[2], 0.0f);
			}
			else
			{
				fail = 1;
			}
		}
	}
	return __syncthreads_or(fail);
}

#define S2_WIDE_OVERFLOW 8 // the launch carries one more workgroup, which sweeps the contacts of the overflow positions (PersistView::overflowBodies) after every
						   // sweep of the strips: the step stays ONE launch while contacts wait there for the worker thread's next structure

// The overflow workgroup of such a launch (the last of the grid).  It stages the bodies the overflow contacts touch, repeats the body
// stages on its copies as every strip does on its imports (poses never travel), and per sweep op: takes {v, w} of every body some strip
// writes from that strip (which has finished its own rounds of the sweep: the overflow positions come last in the sweep order), sweeps
// the contacts ONE AFTER THE OTHER in lane 0 -- they may share a body, the ball that touches boxes of two strips -- with the
// per-constraint functions of the colour batches on the copies in LDS (impulses in the SoA arrays), and hands {v, w} back to every
// strip that stages the body.  The same operations on the same operands as Executor::runPersistentSliced's launches.
template <int KIND> S2_DEV void wideOverflowWorker(const ContactView& c, const BodyView& g, const PersistView& pv, const Op* ops, int opCount, float4* lds)
{
	const int tid = (int)threadIdx.x;
	const int M = pv.overflowBodyCount;
	// lane m keeps body m's records in its registers between the sweeps; for a sweep {v, w} and the pose go to LDS, where wave 0's lanes
	// take their turns on them
	// (These launches carry a 36-byte private frame the code never touches -- no scratch instruction in the kernel: tools/kernel_resources.py
	// counts them -- whichever memory the two arrays live in once the records are pre-loaded per lane; s2amd_create has the queue's scratch
	// allocated by then: s2WarmScratch below.)
	__shared__ float4 lvel[S2_OVERFLOW_BODIES], ldq[S2_OVERFLOW_BODIES];
	Op* lops = (Op*)lds;
	uint32_t flags = 0u;
	bool exchanged = false;
	float4 vel = make_float4(0.0f, 0.0f, 0.0f, 0.0f), dq = vel, integ = vel;
	float angDamp = 0.0f;
	if (tid < M)
	{
		const int e = pv.overflowBodies[tid];
		if (e >= 0)
		{
			const int gi = e & 0x3fffffff;
			exchanged = (e & 0x40000000) == 0;
			vel = g.vel[gi], dq = g.dq[gi], integ = g.integ[gi], angDamp = g.angDamp[gi];
			flags = g.flags[gi];
		}
	}
	for (int i = tid; i < opCount * 8; i += S2_WIDE_THREADS)
	{
		((int*)lops)[i] = ((const int*)ops)[i];
	}
	// lane j of wave 0 takes the contact of overflow position j (S2_OVERFLOW_SLACK <= 64 positions)
	const int myK = pv.overflowBegin + tid;
	const bool mine = tid < 64 && myK < pv.overflowEnd && c.contactIndex[myK] >= 0;
	__syncthreads();
	const LdsBodies lb{lvel, ldq};
	gu64* in = (gu64*)pv.granules + pv.overflowGranBase + 4 * tid;
	gu64* out = in + S2_OVERFLOW_GRANULES / 2;
	unsigned sweep = 0u;
	for (int oi = 0; oi < opCount; ++oi)
	{
		const Op op = lops[oi];
		if (op.code == OP_INTEGRATE_VEL)
		{
			if ((flags & S2F_DYNAMIC) != 0)
			{
				V2 lv = add(v2(vel.x, vel.y), v2(integ.x, integ.y));
				float w = vel.z + integ.z;
				lv = mulSV(integ.w, lv);
				w *= angDamp;
				vel = make_float4(lv.x, lv.y, w, 0.0f);
			}
		}
		else if (op.code == OP_INTEGRATE_POS)
		{
			if ((flags & S2F_MOVES) != 0)
			{
				V2 dpos = mulAdd(v2(dq.x, dq.y), op.h, v2(vel.x, vel.y));
				Rot q;
				q.s = dq.z, q.c = dq.w;
				q = integrateRot(q, op.h * vel.z);
				dq = make_float4(dpos.x, dpos.y, q.s, q.c);
			}
		}
		else if (op.code == OP_FINALIZE)
		{
			// (a copy: the owner strip writes the position)
			if ((flags & (op.flag ? S2F_DYNAMIC : S2F_MOVES)) != 0)
			{
				dq = make_float4(0.0f, 0.0f, dq.z, dq.w);
			}
		}
		else if (op.code == OP_SOLVE_SOFT)
		{
			// the contact's record travels while the strips finish their sweep (the impulses in it are this lane's own of the sweep before)
			SoftRegs<KIND> r;
			if (mine)
			{
				r = loadSoft<KIND, S2_IDX_LOCAL>(c, myK);
			}
			sweep += 1u;
			const int ring = (int)(sweep & (S2_OVERFLOW_RING - 1)) * S2_OVERFLOW_BODIES * 4;
			int fail = 0;
			if (exchanged)
			{
				float v[3];
				if (getGranules<3>(in + ring, sweep, v, pv.error, pv.deviceError, pv.spinLimit))
				{
					vel = make_float4(v[0], v[1], v[2], 0.0f);
				}
				else
				{
					fail = 1;
				}
			}
			if (tid < M)
			{
				lvel[tid] = vel, ldq[tid] = dq;
			}
			if (__syncthreads_or(fail))
			{
				return;
			}
			if (tid < 64)
			{
				// one after the other, in position order (they may share a body); LDS operations of one wave execute in program order
				for (unsigned long long turns = __ballot(mine); turns != 0ull; turns &= turns - 1ull)
				{
					if (tid == __builtin_ctzll(turns))
					{
						solveSoftRegs<KIND>(r, c, lb, op.inv_h, op.useBias, myK);
						storeSoft<KIND>(c, r, myK);
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				}
			}
			__syncthreads();
			if (exchanged)
			{
				vel = lvel[tid];
				putGranule(out + ring + 0, sweep, vel.x), putGranule(out + ring + 1, sweep, vel.y), putGranule(out + ring + 2, sweep, vel.z);
			}
		}
		else if (op.code == OP_WARM)
		{
			WarmRegs w;
			if (mine)
			{
				w = op.kind == WARM_FIXED ? loadWarm<WARM_FIXED>(c, lb, myK) : loadWarm<WARM_CURRENT>(c, lb, myK);
			}
			sweep += 1u;
			const int ring = (int)(sweep & (S2_OVERFLOW_RING - 1)) * S2_OVERFLOW_BODIES * 4;
			int fail = 0;
			if (exchanged)
			{
				float v[3];
				if (getGranules<3>(in + ring, sweep, v, pv.error, pv.deviceError, pv.spinLimit))
				{
					vel = make_float4(v[0], v[1], v[2], 0.0f);
				}
				else
				{
					fail = 1;
				}
			}
			if (tid < M)
			{
				lvel[tid] = vel, ldq[tid] = dq;
			}
			if (__syncthreads_or(fail))
			{
				return;
			}
			if (tid < 64)
			{
				for (unsigned long long turns = __ballot(mine); turns != 0ull; turns &= turns - 1ull)
				{
					if (tid == __builtin_ctzll(turns))
					{
						if (op.kind == WARM_FIXED)
						{
							applyWarm<WARM_FIXED>(w, lb);
						}
						else
						{
							applyWarm<WARM_CURRENT>(w, lb);
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				}
			}
			__syncthreads();
			if (exchanged)
			{
				vel = lvel[tid];
				putGranule(out + ring + 0, sweep, vel.x), putGranule(out + ring + 1, sweep, vel.y), putGranule(out + ring + 2, sweep, vel.z);
			}
		}
	}
}

template <int POINTS, int RPH, int SR, int SL = 0, int IL = 0, int MODE = 0, int KIND = SOFT_TGS> __global__ __launch_bounds__(S2_WIDE_THREADS) void wideStepKernel(ContactView c, BodyView g, StripTableView ta, PersistView pv, const Op* ops, int opCount, WideSelf self)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	constexpr bool SELF = (MODE & S2_WIDE_SELF) != 0;
	constexpr bool BODYWARM = (MODE & S2_WIDE_BODYWARM) != 0;
	static_assert(!BODYWARM || (SL == 0 && IL == 0), "the body-centric warm start keeps no terms for parked rounds");
	constexpr bool SLICED = (MODE & S2_WIDE_SLICED) != 0;
	constexpr bool OVERFLOW = (MODE & S2_WIDE_OVERFLOW) != 0;
	static_assert(KIND == SOFT_TGS || (MODE & ~(S2_WIDE_SLICED | S2_WIDE_OVERFLOW)) == 0, "the self-contained form and the body-centric warm start are s2Solve_TGS_Soft's");
	static_assert(!SLICED || MODE == S2_WIDE_SLICED, "a sliced step's launches are the plain form");
	static_assert(!OVERFLOW || MODE == S2_WIDE_OVERFLOW, "the overflow workgroup rides with the plain form");
	if constexpr (OVERFLOW)
	{
		if (blockIdx.x + 1u == gridDim.x)
		{
			wideOverflowWorker<KIND>(c, g, pv, ops, opCount, lds);
			return;
		}
	}
	static_assert(KIND != SOFT_FIXED || (SL == 0 && IL == 0), "s2Solve_SoftStep: the variants without parked rounds");
	// the record form the constraint functions take: s2Solve_PGS_Soft on a variant without parked rounds keeps its anchors in LDS too
	// (RK: the records a lane keeps -- s2Solve_PGS_Soft's with their anchors in LDS; RKP: the records of parked rounds, whole in LDS)
	constexpr int RK = KIND == SOFT_PGS ? S2_WIDE_PGS_ARMS : KIND;
	constexpr int RKP = KIND;
	// resident records whose local anchors wait in LDS instead of registers (warmWide / prepWide: LL): the last LA of the RPH interior
	// + SR seam records a lane holds -- s2Solve_TGS_Soft's variants beyond the <3, 2> layout
	constexpr int NRES = RPH + SR;
	constexpr int LA = KIND == SOFT_TGS ? wideLocalsInLds(RPH, SR, SL, IL, SLICED || OVERFLOW) : 0;
	constexpr int LL0 = NRES - LA; // the first such record (interior records 0 .. RPH - 1, then the seam records)
	static_assert(LA >= 0 && LA <= NRES, "wideLocalsInLds");
	const int tid = (int)threadIdx.x;
#ifndef S2_WIDE_SPREAD_HALVES
#define S2_WIDE_SPREAD_HALVES 0
#endif
#if S2_WIDE_SPREAD_HALVES
	// (experiment, r4: the halves as waves {0, 1, 4, 5} / {2, 3, 6, 7}.  A CU deals its waves round-robin onto its four SIMDs and a
	// round of a two-level strip holds ~100 constraints = the first two waves of a half, so with waves 0-3 / 4-7 the chain (waves 0, 1)
	// and the prep (waves 4, 5) share SIMDs 0 and 1 while SIMDs 2 and 3 idle; this mapping gives each its own pair of SIMDs.  Measured
	// at base 200: 129.7 against 130.3 us per launch -- nothing: a round is ONE wave's dependent instruction stream, 4 cycles an
	// instruction, and a second wave on the same SIMD fills its stalls rather than lengthening it.  Bit-exact either way.)
	const int half = (tid >> 7) & 1, ht = (tid & 127) | ((tid >> 8) << 7);
#else
	const int half = tid >> 8, ht = tid & 255; // hand-offs: waves 0-3 serve the left neighbour, waves 4-7 the right
#endif
	// stamps: (wall_clock64 << 4) | tag; tags: 0 start, 1 loaded, 2 body stage, 3 warm start, 4 interior rounds, 5 hand-off, 6 seam rounds, 7 end
	// (the workgroup that stamps: persist_debug bits 8-15 + 1, default one in the middle of the island)
	const bool stamp = S2_PERSIST_INSTRUMENTED && pv.debugTimes != nullptr && tid == 0 &&
					   blockIdx.x == (((pv.debugSkip >> 8) & 0xff) != 0 ? (unsigned)(((pv.debugSkip >> 8) & 0xff) - 1) : 8 * (gridDim.x / 16) + 3);
	int stamps = 0;
	auto stampAt = [&](unsigned tag) {
		if (stamp && stamps < 250)
		{
			pv.debugTimes[stamps++] = (wall_clock64() << 4) | tag;
		}
	};
	stampAt(0);
	// Strip <-> workgroup: consecutive strips on ONE XCD, so that a seam's two workgroups share an L2 (the dispatcher is observed
	// to place block b on XCD b % 8: a speed assumption only, checked below).  XCD x runs blocks x, x + 8, ...; it takes the
	// strips [start(x), start(x) + count(x)).
	const int K = (int)gridDim.x - (OVERFLOW ? 1 : 0);
	const int xcd = (int)blockIdx.x & 7, lane8 = (int)blockIdx.x >> 3;
	const int strip = S2_WIDE_XCD_AFFINE ? xcd * (K >> 3) + (xcd < (K & 7) ? xcd : (K & 7)) + lane8 : (int)blockIdx.x;
	// does this strip's left / right neighbour run on my XCD (by the same map)?
	const int firstOfXcd = xcd * (K >> 3) + (xcd < (K & 7) ? xcd : (K & 7));
	const int countOfXcd = (K >> 3) + (xcd < (K & 7) ? 1 : 0);
	const bool nearAllowed = S2_WIDE_XCD_AFFINE && pv.nearHandoff != 0;
	const bool hopeLeft = nearAllowed && strip > firstOfXcd, hopeRight = nearAllowed && strip + 1 < firstOfXcd + countOfXcd;
	// The census: every workgroup publishes the XCD it REALLY runs on (write-through, once per launch), and a seam takes the
	// L2 path only when both of its workgroups have read the same id from each other -- results never depend on placement.
	unsigned myXcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(myXcc));
	// SELF: a step enqueued behind one that lost a hand-off must not run at all -- the resident world has to stand where it stood
	// before the first failure (s2amd_synchronize).  (The hand-off tags start from zero in every launch: this kernel clears the
	// buffers it reads at its end, as the epilogue launch does for the multi-launch form.  A tag base read from memory instead -- one
	// more scalar that lives through the kernel -- costs the step loop its register allocation: 32 spilled registers, measured.)
	constexpr unsigned epoch0 = 0u;
	if constexpr (SELF)
	{
		if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(pv.deviceError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u)
		{
			return;
		}
	}
	gu64* census = (gu64*)pv.granules + pv.censusBase;
	if (S2_WIDE_XCD_AFFINE && tid == 0)
	{
		putGranule(census + strip, epoch0 + 1u, __uint_as_float(myXcc + 1u));
	}
	const StripDesc* da = ta.descs + strip;
	const PersistDesc* pd = pv.descs + strip;
	const int bodyBase = da->bodyBase, nb = da->bodyCount, roundsA = da->batchCount;
	constexpr int ROUNDS = 2 * RPH; // interior colour batches a lane keeps records of (RPH per half of the workgroup)
	constexpr int RA = ROUNDS + IL; // ... and this variant takes
	static_assert(RA <= S2_STRIP_ROUNDS_MAX, "StripDesc::batch");
	int2 batchA[RA + 1]; // (+ 1: never read; keeps the index expressions of the IL == 0 variants in range)
	batchA[RA] = make_int2(0, 0);
#pragma unroll
	for (int i = 0; i < RA; ++i)
	{
		batchA[i] = make_int2(da->batch[i].x, da->batch[i].y);
	}
	const int nImp0 = pd->importCount[0], nImp1 = pd->importCount[1];
	const int roundsB0 = pd->seamBatchCount[0], roundsB1 = pd->seamBatchCount[1];
	constexpr int ST = SR + SL; // seam colour batches this variant takes
	static_assert(ST <= S2_PERSIST_B_ROUNDS, "PersistDesc::seamBatch");
	int2 batchB0[ST], batchB1[ST];
#pragma unroll
	for (int i = 0; i < ST; ++i)
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
	const bool hopeH = half ? hopeRight : hopeLeft;

	float4* lvel = lds;
	float4* ldq = lds + nt;
	float4* linteg = lds + 2 * nt;							// velocity-integrator constants of every staged body (body_ops.h)
	float* langDamp = (float*)(lds + 3 * nt);				// nt floats, padded to records
	float2* lmass = (float2*)(lds + 3 * nt + (nt + 3) / 4); // {invMass, invI} of every staged body, padded to records
	const int bodyRecords = 3 * nt + (nt + 3) / 4 + (nt + 1) / 2;
	Op* lops = (Op*)(lds + bodyRecords); // 2 records per op
	float4* lcoef = lds + bodyRecords + 2 * opCount; // 2 records + 1 for the census flags (the launch adds them to the size)
	// parked records: SL seam rounds of 6 x parkSeamWidth records (column = tid), then IL interior rounds of 6 x parkInteriorWidth (column = ht)
	const int sw = pv.parkSeamWidth, iw = pv.parkInteriorWidth;
	float4* lparked = lcoef + 3 + tid;
	float4* lparkedI = lcoef + 3 + SL * S2_WIDE_PARKED_RECORDS * sw + ht;
	// SELF: the positions of the staged bodies (s2FinalizePositions adds to them; the SoA array g.pos is not used); BODYWARM: per body
	// the rounds that hold a constraint writing it (bits 0-15; bits 16-31: ... with a second manifold point) and the term table
	float4* lextra = lcoef + 3 + S2_WIDE_PARKED_RECORDS * (SL * sw + IL * iw);
	// ... then the local anchors of the LA records that keep them here: [record - LL0][point][lane] {lA, lB}
	float4* llocals = lextra + tid;
	lextra += 2 * LA * S2_WIDE_THREADS;
	auto localsOf = [&](int r) { return llocals + 2 * (r - LL0 > 0 ? r - LL0 : 0) * S2_WIDE_THREADS; };
	float2* lpos = (float2*)lextra;
	lextra += SELF ? (pv.maxStaged + 1) / 2 : 0;
	constexpr int TR = 2 * RPH + SR; // rounds of the term table: the interior rounds, then the seam rounds
	const int tw = pv.maxStripBodies; // (a multiple of 32: PersistView)
	uint32_t* lmask = (uint32_t*)lextra;
	float2* lt = (float2*)(lextra + (tw + 3) / 4);
	// SOFT_FIXED: rA0 / rB0 of the constraints this lane holds, as {perp(rA0), perp(rB0)} per manifold point: [record][point][lane].
	// (In registers they would be 8 more dwords for each of the five resident records: the 30-dword constraint that kept
	// s2Solve_SoftStep on the 256-thread kernel through round 3.  They are read once per sweep, by the prep -- in lanes that wait.)
	const float4* larms = lextra + tid;

	// ---- loads ----
	uint32_t id[S2_WIDE_BODY_CHUNKS];
#pragma unroll
	for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS; ++ch)
	{
		const int i = tid + ch * S2_WIDE_THREADS;
		id[ch] = i < nb ? (uint32_t)ta.bodyIds[bodyBase + i] : 0u;
	}
	const int impId = ht < nImpH ? pv.importIds[pd->importIdBase[half] + ht] : -1;
	const int expIdx = ht < nExpH ? pv.exportSrc[pd->exportSrcBase[half] + ht] : 0;
	for (int i = tid; i < opCount * 8; i += S2_WIDE_THREADS)
	{
		((int*)lops)[i] = ((const int*)ops)[i];
	}
	if (tid < 2)
	{
		lcoef[tid] = pv.softCoef[tid];
	}
	// interior round i runs on the lanes of half i & 1 (waves 0-3 take the even rounds, waves 4-7 the odd ones): a lane holds
	// the records of rounds 2 s + half, s = 0..2, and the constraint it holds there is recomputed, not stored
	auto kOfSlot = [&](int s) {
		const int i = 2 * s + half; // (wave-uniform)
		const int k = (half ? batchA[2 * s + 1].x : batchA[2 * s].x) + ht;
		return (i < roundsA && k < (half ? batchA[2 * s + 1].y : batchA[2 * s].y)) ? k : -1;
	};
	WideRegs rA[RPH];
	// seam constraints: round r = left seam's batch r followed by right seam's batch r, one per lane (a round holds at
	// most 512 constraints); where an item lives is recomputed, not stored
	auto seamItem = [&](int r, int& seam, int& k, uint32_t salt = 0u) {
		const int n0 = r < roundsB0 ? batchB0[r].y - batchB0[r].x : 0;
		const int n1 = r < roundsB1 ? batchB1[r].y - batchB1[r].x : 0;
		const int idx = (int)((uint32_t)tid ^ salt);
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
	WideRegs rB[SR];
	uint32_t seamMask = 0u; // bit i: this lane holds a seam constraint in seam round i
	// the constraint this lane holds in parked interior round j (round ROUNDS + j runs on half (ROUNDS + j) & 1 like every interior round)
	auto kOfParked = [&](int j) {
		const int i = ROUNDS + j;
		const int k = batchA[i].x + ht;
		return (i < roundsA && (i & 1) == half && k < batchA[i].y) ? k : -1;
	};
	if constexpr (SELF)
	{
		// s2PrepareContacts_Soft (solve_common.c:188-274) for the constraints this lane holds, straight from the wire contacts and
		// bodies (soft_from_wire.h: the operations of prepareContactsKernel<PREP_SOFT>) and out to the SoA arrays the prologue launch
		// would have filled; from there they come back as plain loads, exactly as in the multi-launch form.  (Handed on in registers,
		// the five records' live ranges span this phase's own peak: measured 26 spilled registers that the step loop reloads at every
		// use.)  Three dependent memory round trips -- position -> pool slot -> contact -> its two bodies --, each issued for ALL of the
		// lane's constraints at once (unconditional loads from clamped indices: one constraint after the other cost the launch 20 us).
		// A seam's constraints are prepared by both of its workgroups: the same bits to the same addresses.  A free position of
		// the slack layout becomes an empty record.
		constexpr int NP = RPH + ST + IL;
		int kk[NP];
#pragma unroll
		for (int s = 0; s < RPH; ++s)
		{
			kk[s] = kOfSlot(s);
		}
#pragma unroll
		for (int i = 0; i < ST; ++i)
		{
			int seam, k;
			kk[RPH + i] = (i < roundsB && seamItem(i, seam, k)) ? k : -1;
		}
#pragma unroll
		for (int j = 0; j < IL; ++j)
		{
			kk[RPH + ST + j] = kOfParked(j);
		}
		int slot[NP];
#pragma unroll
		for (int i = 0; i < NP; ++i)
		{
			slot[i] = c.contactIndex[kk[i] >= 0 ? kk[i] : 0];
		}
		WireContactRaw raw[NP];
#pragma unroll
		for (int i = 0; i < NP; ++i)
		{
			slot[i] = kk[i] >= 0 ? slot[i] : -1;
			raw[i] = loadWireContact(self.wire + (slot[i] >= 0 ? slot[i] : 0), g.capacity);
		}
		WireBodiesRaw rawBodies[NP];
#pragma unroll
		for (int i = 0; i < NP; ++i)
		{
			rawBodies[i] = loadWireBodies(self.wireBodies, self.hostFlags, raw[i]);
		}
#pragma unroll
		for (int i = 0; i < NP; ++i)
		{
			float4 nf, an[2], par[2];
			float2 imp[2];
			prepareSoftFromRaw(raw[i], rawBodies[i], self.warmStart, slot[i] >= 0, nf, an, par, imp);
			if (kk[i] >= 0)
			{
				const int k = kk[i];
				c.nf[k] = nf;
#pragma unroll
				for (int j = 0; j < 2; ++j)
				{
					c.anchor[j][k] = an[j], c.param[j][k] = par[j], c.impulse[j][k] = imp[j];
				}
			}
		}
		asm volatile("" ::: "memory"); // (the loads below are not to be forwarded from the stores above)
	}
	// the prepared records of the constraints this lane holds, out of the SoA arrays: into registers, the parked rounds into LDS.
	// (The register-resident ones with unconditional loads from clamped positions, so that all five records' loads are in flight
	// together -- a lane without a constraint in a round reads position 0 and never looks at the record.)
#pragma unroll
	for (int s = 0; s < RPH; ++s)
	{
		const int k = kOfSlot(s);
		const int kc = k >= 0 ? k : 0;
		const int2 lb = c.localBodies[kc];
		rA[s] = loadWide<RK>(c, kc, lb.x, lb.y);
		if (s >= LL0)
		{
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				localsOf(s)[j * S2_WIDE_THREADS] = make_float4(rA[s].lA[j].x, rA[s].lA[j].y, rA[s].lB[j].x, rA[s].lB[j].y);
			}
		}
		if constexpr (wideLdsArms<RK>)
		{
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				const float4 a = c.r0[j][kc];
				lextra[(2 * s + j) * S2_WIDE_THREADS + tid] = make_float4(-a.y, a.x, -a.w, a.z);
			}
		}
	}
#pragma unroll
	for (int i = 0; i < SR; ++i)
	{
		int seam = 0, k = 0;
		const bool mine = i < roundsB && seamItem(i, seam, k);
		const int kc = mine ? k : 0;
		const int2 lb = c.localBodies[kc];
		const int base = mine ? pd->remapBase[seam] : 0; // (a valid entry of the remap table either way)
		rB[i] = loadWide<RK>(c, kc, pv.remap[base + (mine ? lb.x : 0)], pv.remap[base + (mine ? lb.y : 0)]);
		if (RPH + i >= LL0)
		{
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				localsOf(RPH + i)[j * S2_WIDE_THREADS] = make_float4(rB[i].lA[j].x, rB[i].lA[j].y, rB[i].lB[j].x, rB[i].lB[j].y);
			}
		}
		if constexpr (wideLdsArms<RK>)
		{
#pragma unroll
			for (int j = 0; j < 2; ++j)
			{
				const float4 a = c.r0[j][kc];
				lextra[(2 * (RPH + i) + j) * S2_WIDE_THREADS + tid] = make_float4(-a.y, a.x, -a.w, a.z);
			}
		}
		seamMask |= mine ? 1u << i : 0u;
	}
#pragma unroll
	for (int i = SR; i < ST; ++i)
	{
		int seam, k;
		if (i < roundsB && seamItem(i, seam, k))
		{
			const int2 lb = c.localBodies[k];
			parkWide(lparked + (i - SR) * S2_WIDE_PARKED_RECORDS * sw, sw,
					 loadWide<RKP>(c, k, pv.remap[pd->remapBase[seam] + lb.x], pv.remap[pd->remapBase[seam] + lb.y]));
			seamMask |= 1u << i;
		}
	}
#pragma unroll
	for (int j = 0; j < IL; ++j)
	{
		const int k = kOfParked(j);
		if (k >= 0)
		{
			const int2 lb = c.localBodies[k];
			parkWide(lparkedI + j * S2_WIDE_PARKED_RECORDS * iw, iw, loadWide<RKP>(c, k, lb.x, lb.y));
		}
	}
	// bodies (+ their integrator constants) into LDS: own list, then this half's imports
	uint32_t flags[S2_WIDE_BODY_CHUNKS + 1];
	int ldsIdx[S2_WIDE_BODY_CHUNKS + 1];
#pragma unroll
	for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS + 1; ++ch)
	{
		int gi = -1;
		if (ch < S2_WIDE_BODY_CHUNKS)
		{
			const int i = tid + ch * S2_WIDE_THREADS;
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
			if constexpr (SELF)
			{
				// body_ops.h: unpackBodyOne, into LDS instead of the SoA arrays.  (Asking for these records before the constraints are
				// prepared, so that they travel meanwhile, was measured: the 66 registers they hold spill, 161 against 151 us per launch.)
				const s2amdBody* w = self.wireBodies + gi;
				const int type = w->type;
				uint32_t f = 0x80000000u;
				if (type != S2AMD_BODY_FREE)
				{
					f |= S2F_LIVE | (type == S2AMD_BODY_DYNAMIC ? S2F_DYNAMIC : 0u) | (type != S2AMD_BODY_STATIC ? S2F_MOVES : 0u);
				}
				flags[ch] = f;
				const int i = ldsIdx[ch];
				lvel[i] = make_float4(w->linearVelocity[0], w->linearVelocity[1], w->angularVelocity, 0.0f);
				ldq[i] = make_float4(w->deltaPosition[0], w->deltaPosition[1], w->rot[0], w->rot[1]);
				if (ch < S2_WIDE_BODY_CHUNKS)
				{
					lpos[i] = make_float2(w->position[0], w->position[1]);
				}
				lmass[i] = make_float2(w->invMass, w->invI);
				const V2 gravity = v2(self.gravityX, self.gravityY);
				const V2 force = v2(w->force[0], w->force[1]);
				const V2 inner = mulAdd(force, w->mass * w->gravityScale, gravity);
				const V2 a = mulSV(self.unpackH * w->invMass, inner);
				const float aw = self.unpackH * w->invI * w->torque;
				const float ld = 1.0f / (1.0f + self.unpackH * w->linearDamping);
				const float ad = 1.0f / (1.0f + self.unpackH * w->angularDamping);
				linteg[i] = make_float4(a.x, a.y, aw, ld);
				langDamp[i] = ad;
			}
			else
			{
				lvel[ldsIdx[ch]] = g.vel[gi];
				ldq[ldsIdx[ch]] = g.dq[gi];
				flags[ch] = g.flags[gi] | 0x80000000u; // bit 31: slot in use
				linteg[ldsIdx[ch]] = g.integ[gi];
				langDamp[ldsIdx[ch]] = g.angDamp[gi];
				lmass[ldsIdx[ch]] = g.massInv[gi];
			}
		}
	}
	if constexpr (BODYWARM)
	{
		for (int i = tid; i < tw; i += S2_WIDE_THREADS)
		{
			lmask[i] = 0u;
		}
	}
	// OVERFLOW: where this workgroup stages the bodies the overflow contacts touch -- (priority << 24) | LDS slot, 0: not here.  The
	// owner (priority 3) publishes the body after every sweep; every copy that takes part in sweeps -- the owner's, the importing
	// neighbour's (2) -- takes the overflow workgroup's result (a copy an adoption left behind in a body list, 1, only if no other is here).
	__shared__ int lovf[S2_OVERFLOW_BODIES];
	const int ovCount = OVERFLOW ? pv.overflowBodyCount : 0;
	if constexpr (OVERFLOW)
	{
		if (tid < S2_OVERFLOW_BODIES)
		{
			lovf[tid] = 0;
		}
		__syncthreads();
#pragma unroll
		for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS + 1; ++ch)
		{
			if (flags[ch] != 0u)
			{
				const int gi = ch < S2_WIDE_BODY_CHUNKS ? (int)(id[ch] & ~S2G_OWNED) : impId;
				const int prio = ch < S2_WIDE_BODY_CHUNKS ? ((id[ch] & S2G_OWNED) != 0 ? 3 : 1) : 2;
				for (int m = 0; m < ovCount; ++m)
				{
					if (pv.overflowBodies[m] == gi) // (an entry nothing writes carries bit 30 and matches no pool slot: no exchange)
					{
						atomicMax(&lovf[m], (prio << 24) | ldsIdx[ch]);
					}
				}
			}
		}
	}
	unsigned ovSweep = 0u;
	// the neighbours' census entries have had the whole load phase to land; a missing one only costs the fast path
	int* lnear = (int*)(lcoef + 2);
	if (ht == 0)
	{
		int near = 0;
		if (hopeH)
		{
			gu64* g = census + (half ? strip + 1 : strip - 1);
			for (int spins = 0; spins < 4096 && !near; ++spins)
			{
				const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((unsigned)(x >> 32) == epoch0 + 1u)
				{
					near = (unsigned)x == myXcc + 1u ? 1 : -1;
				}
			}
		}
		lnear[half] = near > 0 ? 1 : 0;
	}
	__syncthreads();
	const bool nearH = lnear[half] != 0;
	// the doubled contact hertz of a constraint with a static side (prepareContactsKernel<PREP_SOFT>; solve_common.c:219): the
	// test of strip_kernel.hip unpackPersist, made once -- the masses do not change during a step
	auto markStatic = [&](WideRegs& p) {
		const bool st = lmass[p.idx & 0x1fffu].x == 0.0f || lmass[(p.idx >> 13) & 0x1fffu].x == 0.0f;
		p.idx |= st ? 1u << 30 : 0u;
	};
#pragma unroll
	for (int s = 0; s < RPH; ++s)
	{
		if (kOfSlot(s) >= 0)
		{
			markStatic(rA[s]);
		}
	}
#pragma unroll
	for (int i = 0; i < SR; ++i)
	{
		if ((seamMask >> i) & 1u)
		{
			markStatic(rB[i]);
		}
	}
#pragma unroll
	for (int i = SR; i < ST; ++i)
	{
		if ((seamMask >> i) & 1u)
		{
			float4* slot = lparked + (i - SR) * S2_WIDE_PARKED_RECORDS * sw;
			WideRegs p = unparkWide(slot, sw);
			markStatic(p);
			slot[0].x = fromBits(p.idx);
		}
	}
#pragma unroll
	for (int j = 0; j < IL; ++j)
	{
		if (kOfParked(j) >= 0)
		{
			float4* slot = lparkedI + j * S2_WIDE_PARKED_RECORDS * iw;
			WideRegs p = unparkWide(slot, iw);
			markStatic(p);
			slot[0].x = fromBits(p.idx);
		}
	}
	if constexpr (BODYWARM)
	{
		// which rounds write which of my bodies (constant during the step: write bits and point counts are the prepared record's)
		auto note = [&](const WideRegs& p, int round) {
			const int ia = (int)(p.idx & 0x1fffu), ib = (int)((p.idx >> 13) & 0x1fffu);
			const uint32_t bit = (1u << round) | (((p.idx >> 26) & 3u) == 2u ? 1u << (16 + round) : 0u);
			if ((p.idx & (1u << 28)) != 0 && ia < nb)
			{
				atomicOr(&lmask[ia], bit);
			}
			if ((p.idx & (1u << 29)) != 0 && ib < nb)
			{
				atomicOr(&lmask[ib], bit);
			}
		};
#pragma unroll
		for (int s = 0; s < RPH; ++s)
		{
			if (kOfSlot(s) >= 0)
			{
				note(rA[s], 2 * s + half);
			}
		}
#pragma unroll
		for (int i = 0; i < SR; ++i)
		{
			if ((seamMask >> i) & 1u)
			{
				note(rB[i], ROUNDS + i);
			}
		}
		__syncthreads();
	}
	stampAt(1);

	unsigned epoch = epoch0; // tags are the exchange number: the buffers are zero at launch (cleared by the previous step's epilogue), or (SELF) hold older tags only
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
			if (BODYWARM && oi + 1 < opCount && lops[oi + 1].code == OP_WARM)
			{
				continue; // the warm start's body pass integrates the velocities first, in the same lane
			}
#pragma unroll
			for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS + 1; ++ch)
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
			for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS + 1; ++ch)
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
			for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS + 1; ++ch)
			{
				if ((flags[ch] & need) != 0)
				{
					const int i = ldsIdx[ch];
					const float4 d = ldq[i];
					if (ch < S2_WIDE_BODY_CHUNKS && (id[ch] & S2G_OWNED) != 0)
					{
						if constexpr (SELF)
						{
							const float2 pos = lpos[i];
							const V2 np = add(v2(pos.x, pos.y), v2(d.x, d.y));
							lpos[i] = make_float2(np.x, np.y);
						}
						else
						{
							const int gi = (int)(id[ch] & ~S2G_OWNED);
							const float2 pos = g.pos[gi];
							const V2 np = add(v2(pos.x, pos.y), v2(d.x, d.y));
							g.pos[gi] = make_float2(np.x, np.y);
						}
					}
					ldq[i] = make_float4(0.0f, 0.0f, d.z, d.w);
				}
			}
			__syncthreads();
			stampAt(2);
		}
		else if (BODYWARM && op.code == OP_WARM)
		{
			// body-centric (warmTermsWide): every lane writes the terms of the constraints it holds, every body of the strip's own
			// list adds the terms of its rounds in round order -- behind s2IntegrateVelocities when that is the op before (the
			// imported copies need neither: the next sweep's exchange overwrites them before anything reads them)
#pragma unroll
			for (int s = 0; s < RPH; ++s)
			{
				if (kOfSlot(s) >= 0)
				{
					if (s >= LL0)
					{
						warmTermsWide<POINTS, true>(rA[s], ldq, lmass, lt, tw, TR, 2 * s + half, nb, salt, localsOf(s));
					}
					else
					{
						warmTermsWide<POINTS>(rA[s], ldq, lmass, lt, tw, TR, 2 * s + half, nb, salt);
					}
				}
			}
#pragma unroll
			for (int i = 0; i < SR; ++i)
			{
				if ((seamMask >> i) & 1u)
				{
					if (RPH + i >= LL0)
					{
						warmTermsWide<POINTS, true>(rB[i], ldq, lmass, lt, tw, TR, ROUNDS + i, nb, salt, localsOf(RPH + i));
					}
					else
					{
						warmTermsWide<POINTS>(rB[i], ldq, lmass, lt, tw, TR, ROUNDS + i, nb, salt);
					}
				}
			}
			__syncthreads();
			stampAt(15);
			const bool integrate = oi > 0 && lops[oi - 1].code == OP_INTEGRATE_VEL;
#pragma unroll
			for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS; ++ch)
			{
				const int i = ldsIdx[ch];
				if (flags[ch] != 0u && i < nb)
				{
					const uint32_t m = lmask[i];
					const bool dynamic = integrate && (flags[ch] & S2F_DYNAMIC) != 0;
					if (m != 0u || dynamic)
					{
						const float4 v = lvel[i];
						V2 lv = v2(v.x, v.y);
						float w = v.z;
						if (dynamic)
						{
							const float4 k = linteg[i];
							lv = add(lv, v2(k.x, k.y));
							w = w + k.z;
							lv = mulSV(k.w, lv);
							w *= langDamp[i];
						}
						const int plane = TR * tw;
#pragma unroll
						for (int r = 0; r < TR; ++r)
						{
							if ((m >> r) & 1u)
							{
								const float2 t0 = lt[r * tw + i], t1 = lt[plane + r * tw + i], t2 = lt[2 * plane + r * tw + i];
								lv.x = lv.x + t0.x, lv.y = lv.y + t0.y, w = w + t1.x;
								if (POINTS == 2 || ((m >> (16 + r)) & 1u))
								{
									lv.x = lv.x + t1.y, lv.y = lv.y + t2.x, w = w + t2.y;
								}
							}
						}
						lvel[i] = make_float4(lv.x, lv.y, w, 0.0f);
					}
				}
			}
			__syncthreads();
			stampAt(3);
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
					if ((i & 1) == half && kOfSlot(i >> 1) >= 0)
					{
						if ((i >> 1) >= LL0)
						{
							warmWide<RK, POINTS, true>(rA[i >> 1], lvel, ldq, lmass, salt, nullptr, localsOf(i >> 1));
						}
						else
						{
							warmWide<RK, POINTS>(rA[i >> 1], lvel, ldq, lmass, salt, larms + 2 * (i >> 1) * S2_WIDE_THREADS);
						}
					}
					__syncthreads();
				}
			}
#pragma unroll
			for (int j = 0; j < IL; ++j)
			{
				if (ROUNDS + j < roundsA)
				{
					if (kOfParked(j) >= 0)
					{
						warmWide<RKP, POINTS>(unparkWide(lparkedI + j * S2_WIDE_PARKED_RECORDS * iw, iw), lvel, ldq, lmass, salt);
					}
					__syncthreads();
				}
			}
#pragma unroll
			for (int i = 0; i < SR; ++i)
			{
				if (i < roundsB)
				{
					if ((seamMask >> i) & 1u)
					{
						if (RPH + i >= LL0)
						{
							warmWide<RK, POINTS, true>(rB[i], lvel, ldq, lmass, salt, nullptr, localsOf(RPH + i));
						}
						else
						{
							warmWide<RK, POINTS>(rB[i], lvel, ldq, lmass, salt, larms + 2 * (RPH + i) * S2_WIDE_THREADS);
						}
					}
					__syncthreads();
				}
			}
#pragma unroll
			for (int i = SR; i < ST; ++i)
			{
				if (i < roundsB)
				{
					if ((seamMask >> i) & 1u)
					{
						warmWide<RKP, POINTS>(unparkWide(lparked + (i - SR) * S2_WIDE_PARKED_RECORDS * sw, sw), lvel, ldq, lmass, salt);
					}
					__syncthreads();
				}
			}
			if constexpr (OVERFLOW)
			{
				bad = wideOverflowExchange(pv, lovf, lvel, ovCount, ++ovSweep);
			}
			stampAt(3);
		}
		else if (op.code == OP_SOLVE_SOFT)
		{
			// ---- interiors: round i's chain on half i & 1, while the other half prepares its round i + 1 (measured: 150 us per
			// launch against 159 us with prep and chain back to back in the same lanes) ----
			WidePrep pre;
			if (half == 0 && kOfSlot(0) >= 0)
			{
				if (0 >= LL0)
				{
					pre = prepWide<RK, POINTS, true>(rA[0], ldq, lcoef, op.inv_h, op.useBias, salt, nullptr, localsOf(0));
				}
				else
				{
					pre = prepWide<RK, POINTS>(rA[0], ldq, lcoef, op.inv_h, op.useBias, salt, larms);
				}
			}
			// (the parked rounds ROUNDS .. RA-1 take part in the same schedule: their records come out of LDS for the prep and again for the chain)
#pragma unroll
			for (int i = 0; i < RA; ++i)
			{
				if (i < roundsA)
				{
					if ((i & 1) == half)
					{
						if (i < ROUNDS)
						{
							if (kOfSlot(i >> 1) >= 0)
							{
								chainWide<POINTS>(rA[i >> 1 < RPH ? i >> 1 : 0], pre, lvel, lmass, lcoef, salt);
							}
						}
						else if (kOfParked(i - ROUNDS < IL ? i - ROUNDS : 0) >= 0)
						{
							float4* slot = lparkedI + (i - ROUNDS) * S2_WIDE_PARKED_RECORDS * iw;
							WideRegs p = unparkWide(slot, iw);
							chainWide<POINTS>(p, pre, lvel, lmass, lcoef, salt);
							slot[5 * iw] = make_float4(p.imp[0].x, p.imp[0].y, p.imp[1].x, p.imp[1].y);
						}
					}
					else if (i + 1 < RA) // (my next round is i + 1)
					{
						if (i + 1 < ROUNDS)
						{
							if (kOfSlot((i + 1) >> 1) >= 0)
							{
								if (((i + 1) >> 1) >= LL0)
								{
									pre = prepWide<RK, POINTS, true>(rA[(i + 1) >> 1 < RPH ? (i + 1) >> 1 : 0], ldq, lcoef, op.inv_h, op.useBias, salt, nullptr, localsOf((i + 1) >> 1));
								}
								else
								{
									pre = prepWide<RK, POINTS>(rA[(i + 1) >> 1 < RPH ? (i + 1) >> 1 : 0], ldq, lcoef, op.inv_h, op.useBias, salt, larms + 2 * ((i + 1) >> 1) * S2_WIDE_THREADS);
								}
							}
						}
						else if (kOfParked(i + 1 - ROUNDS < IL ? i + 1 - ROUNDS : 0) >= 0)
						{
							pre = prepWide<RKP, POINTS>(unparkWide(lparkedI + (i + 1 - ROUNDS) * S2_WIDE_PARKED_RECORDS * iw, iw), ldq, lcoef, op.inv_h, op.useBias, salt);
						}
					}
					__syncthreads();
					if (S2_PERSIST_INSTRUMENTED && i + 1 < roundsA && i < 6)
					{
						stampAt(8 + i);
					}
				}
			}
			stampAt(4);
			// ---- symmetric exchange of the seam bodies' velocities (poses are replicated by the body stages) ----
			epoch += 1;
			const int par = (int)(epoch & 1u) * pv.parityStride;
			const bool mute = (pv.debugSkip & 8) != 0 && strip == 1; // fault injection: this workgroup stays silent
			if (ht < nExpH && !mute)
			{
				const float4 v = lvel[expIdx];
				gu64* p = gran + par + outH + 4 * ht;
				if (nearH)
				{
					putGranuleNear(p + 0, epoch, v.x), putGranuleNear(p + 1, epoch, v.y), putGranuleNear(p + 2, epoch, v.z);
				}
				else
				{
					putGranule(p + 0, epoch, v.x), putGranule(p + 1, epoch, v.y), putGranule(p + 2, epoch, v.z);
				}
			}
			// the seam constraints' pose-dependent part, while the neighbours' bodies are in flight (poses never travel)
			constexpr int SRP = SR < 2 ? SR : 2; // (a third and fourth seam round prepare just before their chain: registers)
			WidePrep preB[SRP];
#pragma unroll
			for (int i = 0; i < SRP; ++i)
			{
				if ((seamMask >> i) & 1u)
				{
					if (RPH + i >= LL0)
					{
						preB[i] = prepWide<RK, POINTS, true>(rB[i], ldq, lcoef, op.inv_h, op.useBias, salt, nullptr, localsOf(RPH + i));
					}
					else
					{
						preB[i] = prepWide<RK, POINTS>(rB[i], ldq, lcoef, op.inv_h, op.useBias, salt, larms + 2 * (RPH + i) * S2_WIDE_THREADS);
					}
				}
			}
			// (... and of the parked seam rounds: their records come out of LDS for it, and again for the chain)
			WidePrep preP[SL > 0 ? SL : 1];
#pragma unroll
			for (int i = SR; i < ST; ++i)
			{
				if (S2_WIDE_PARKED_PREP_EARLY && ((seamMask >> i) & 1u))
				{
					preP[i - SR] = prepWide<RKP, POINTS>(unparkWide(lparked + (i - SR) * S2_WIDE_PARKED_RECORDS * sw, sw), ldq, lcoef, op.inv_h, op.useBias, salt);
				}
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
			for (int i = 0; i < SR; ++i)
			{
				if (i < roundsB)
				{
					if ((seamMask >> i) & 1u)
					{
						if constexpr (SR > 2)
						{
							if (i >= 2)
							{
								const WidePrep late = (RPH + i >= LL0) ? prepWide<RK, POINTS, true>(rB[i], ldq, lcoef, op.inv_h, op.useBias, salt, nullptr, localsOf(RPH + i))
																	   : prepWide<RK, POINTS>(rB[i], ldq, lcoef, op.inv_h, op.useBias, salt, larms + 2 * (RPH + i) * S2_WIDE_THREADS);
								chainWide<POINTS>(rB[i], late, lvel, lmass, lcoef, salt);
							}
							else
							{
								chainWide<POINTS>(rB[i], preB[i < 2 ? i : 0], lvel, lmass, lcoef, salt);
							}
						}
						else
						{
							chainWide<POINTS>(rB[i], preB[i], lvel, lmass, lcoef, salt);
						}
					}
					__syncthreads();
					if (S2_PERSIST_INSTRUMENTED && i + 1 < roundsB)
					{
						stampAt(14 + i);
					}
				}
			}
#pragma unroll
			for (int i = SR; i < ST; ++i)
			{
				if (i < roundsB)
				{
					if ((seamMask >> i) & 1u)
					{
						float4* slot = lparked + (i - SR) * S2_WIDE_PARKED_RECORDS * sw;
						WideRegs p = unparkWide(slot, sw);
						if (S2_WIDE_PARKED_PREP_EARLY)
						{
							chainWide<POINTS>(p, preP[i - SR < SL ? i - SR : 0], lvel, lmass, lcoef, salt);
						}
						else
						{
							const WidePrep late = prepWide<RKP, POINTS>(p, ldq, lcoef, op.inv_h, op.useBias, salt);
							chainWide<POINTS>(p, late, lvel, lmass, lcoef, salt);
						}
						slot[5 * sw] = make_float4(p.imp[0].x, p.imp[0].y, p.imp[1].x, p.imp[1].y);
					}
					__syncthreads();
				}
			}
			if constexpr (OVERFLOW)
			{
				bad = wideOverflowExchange(pv, lovf, lvel, ovCount, ++ovSweep);
			}
			stampAt(6);
		}
	}

	// ---- results: owned bodies, impulses ----
	if constexpr (SELF)
	{
		// Nothing of a step that lost a hand-off may reach the wire arrays (the host repeats it on the multi-launch path from
		// untouched inputs), and there is no epilogue launch to stand down: every workgroup that came through arrives at a counter
		// and writes only once all K have -- a workgroup that saw a dead hand-off never arrives, so nobody writes.  The counter
		// runs on from step to step (K arrivals each: the step an arrival belongs to is its ticket / K); the host zeroes it after a
		// failure and every 2^20 steps.
		__shared__ int lcommit;
		if (tid == 0)
		{
			int go = 0;
			if (S2_PERSIST_INSTRUMENTED && (pv.debugSkip & 32) != 0)
			{
				go = 1; // (timing experiment, instrumented build only: no commit wait)
			}
			else if (!bad)
			{
				const unsigned ticket = __hip_atomic_fetch_add(pv.state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const unsigned target = (ticket / (unsigned)K + 1u) * (unsigned)K;
				for (unsigned spins = 0; spins < pv.spinLimit; ++spins)
				{
					if (ticket + 1u == target || __hip_atomic_load(pv.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) // (the last to arrive knows at once)
					{
						go = 1;
						break;
					}
					if ((spins & 63u) == 63u && __hip_atomic_load(pv.deviceError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
					{
						break;
					}
					__builtin_amdgcn_s_sleep(1);
				}
				if (!go)
				{
					__hip_atomic_store(pv.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					__hip_atomic_store(pv.deviceError, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			lcommit = go;
		}
		__syncthreads();
		stampAt(12);
		if (!lcommit)
		{
			return;
		}
		// every workgroup is past its last hand-off: the buffers this one reads (both parities) and its census entry go back to
		// zero tags for the next launch
		if (ht < nImpH)
		{
#pragma unroll
			for (int par = 0; par < 2; ++par)
			{
				gu64* p = gran + par * pv.parityStride + inH + 4 * ht;
				p[0] = 0ull, p[1] = 0ull, p[2] = 0ull, p[3] = 0ull;
			}
		}
		if (S2_WIDE_XCD_AFFINE && tid == 0)
		{
			census[strip] = 0ull;
		}
		// body_ops.h: packBodyOne from the LDS copies
#pragma unroll
		for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS; ++ch)
		{
			const int i = tid + ch * S2_WIDE_THREADS;
			if (i < nb && (id[ch] & S2G_OWNED) != 0 && (flags[ch] & S2F_LIVE) != 0)
			{
				s2amdBody* w = self.wireBodies + (int)(id[ch] & ~S2G_OWNED);
				const float4 v = lvel[i], d = ldq[i];
				const float2 pos = lpos[i];
				w->position[0] = pos.x, w->position[1] = pos.y;
				w->rot[0] = d.z, w->rot[1] = d.w;
				w->linearVelocity[0] = v.x, w->linearVelocity[1] = v.y;
				w->angularVelocity = v.z;
				w->deltaPosition[0] = d.x, w->deltaPosition[1] = d.y;
			}
		}
		// s2StoreContactImpulses (solve_common.c:396-410): straight into the manifolds
		auto storeWire = [&](const WideRegs& p, int k) {
			const int slot = c.contactIndex[k];
			const int pointCount = (int)((p.idx >> 26) & 3u);
			if (slot >= 0)
			{
				s2amdContact* contact = self.wire + slot;
#pragma unroll
				for (int j = 0; j < 2; ++j)
				{
					if (j < pointCount)
					{
						contact->points[j].normalImpulse = p.imp[j].x;
						contact->points[j].tangentImpulse = p.imp[j].y;
					}
				}
			}
		};
#pragma unroll
		for (int s = 0; s < RPH; ++s)
		{
			if (kOfSlot(s) >= 0)
			{
				storeWire(rA[s], kOfSlot(s));
			}
		}
#pragma unroll
		for (int j = 0; j < IL; ++j)
		{
			if (kOfParked(j) >= 0)
			{
				storeWire(unparkWide(lparkedI + j * S2_WIDE_PARKED_RECORDS * iw, iw), kOfParked(j));
			}
		}
		// (the right seam's impulses are stored by this workgroup -- its left neighbour of that seam --, nobody stores twice)
#pragma unroll
		for (int i = 0; i < ST; ++i)
		{
			int seam, k;
			if (i < roundsB && seamItem(i, seam, k) && seam == 1)
			{
				if (i < SR)
				{
					storeWire(rB[i < SR ? i : 0], k);
				}
				else
				{
					storeWire(unparkWide(lparked + (i - SR) * S2_WIDE_PARKED_RECORDS * sw, sw), k);
				}
			}
		}
	}
	else
	{
#pragma unroll
		for (int ch = 0; ch < S2_WIDE_BODY_CHUNKS; ++ch)
		{
			const int i = tid + ch * S2_WIDE_THREADS;
			if (i < nb && (id[ch] & S2G_OWNED) != 0)
			{
				const int gi = (int)(id[ch] & ~S2G_OWNED);
				g.vel[gi] = lvel[i];
				g.dq[gi] = ldq[i];
			}
		}
#pragma unroll
		for (int s = 0; s < RPH; ++s)
		{
			if (kOfSlot(s) >= 0)
			{
				storeWide(c, rA[s], kOfSlot(s));
			}
		}
#pragma unroll
		for (int j = 0; j < IL; ++j)
		{
			if (kOfParked(j) >= 0)
			{
				storeWide(c, unparkWide(lparkedI + j * S2_WIDE_PARKED_RECORDS * iw, iw), kOfParked(j));
			}
		}
		// the right seam's impulses are stored by this workgroup (its left neighbour of that seam), nobody stores twice
#pragma unroll
		for (int i = 0; i < SR; ++i)
		{
			int seam, k;
			if (i < roundsB && seamItem(i, seam, k) && seam == 1)
			{
				storeWide(c, rB[i], k);
			}
		}
#pragma unroll
		for (int i = SR; i < ST; ++i)
		{
			int seam, k;
			if (i < roundsB && seamItem(i, seam, k) && seam == 1)
			{
				storeWide(c, unparkWide(lparked + (i - SR) * S2_WIDE_PARKED_RECORDS * sw, sw), k);
			}
		}
	}
	if constexpr (SLICED)
	{
		// sliced step (Executor::runPersistentSliced): the next launch of this step starts its tags from zero again.  Every granule
		// this workgroup reads has been written for the last time -- a neighbour writes one granule set per sweep and this workgroup
		// has read the last sweep's -- and launches of one stream do not overlap, so the reader clears its own inboxes (both
		// parities), as the self-contained form does behind its commit.  (The census entries stay: the next launch's workgroups
		// write the same XCD ids again -- and were the placement ever to differ, a hand-off on the same-L2 path would time out and the
		// step be repeated with agent-scope stores: solver_step.cpp.)
		if (!bad)
		{
			if (ht < nImpH)
			{
#pragma unroll
				for (int par = 0; par < 2; ++par)
				{
					gu64* p = gran + par * pv.parityStride + inH + 4 * ht;
					p[0] = 0ull, p[1] = 0ull, p[2] = 0ull, p[3] = 0ull;
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

// (experiments: compile the headline variant's four modes only -- tools/kernel_ab.sh, register counts)
#ifndef S2_WIDE_ONLY_MAIN
#define S2_WIDE_ONLY_MAIN 0
#endif
#if S2_WIDE_ONLY_MAIN
template __global__ void wideStepKernel<2, 3, 2, 0, 0, 0>(ContactView, BodyView, StripTableView, PersistView, const Op*, int, WideSelf);
template __global__ void wideStepKernel<2, 3, 2, 0, 0, 1>(ContactView, BodyView, StripTableView, PersistView, const Op*, int, WideSelf);
template __global__ void wideStepKernel<2, 3, 2, 0, 0, 2>(ContactView, BodyView, StripTableView, PersistView, const Op*, int, WideSelf);
template __global__ void wideStepKernel<2, 3, 2, 0, 0, 3>(ContactView, BodyView, StripTableView, PersistView, const Op*, int, WideSelf);
template __global__ void wideStepKernel<2, 3, 2, 0, 0, 8>(ContactView, BodyView, StripTableView, PersistView, const Op*, int, WideSelf);
#else
// Eligibility (checked by the caller, solver_executor.h widePlan): TGS_Soft with the current-anchor warm start on a partition with
// at most 6 interior colour batches per strip and 3 per seam, or 8 and 2 (pv.maxRoundsA, pv.maxSeamRounds): five or six resident
// records per lane fit its 256 registers beside the round's working set, seven do not (measured: 160 spilled registers).
template <int RPH, int SR, int SL, int IL, int MODE, int KIND = SOFT_TGS>
static void launchWideMode(hipStream_t s, dim3 grid, size_t lds, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv, const Op* ops, int opCount,
						   const WideSelf& self)
{
	const dim3 block(S2_WIDE_THREADS);
	if (pv.allTwoPoints)
	{
		wideStepKernel<2, RPH, SR, SL, IL, MODE, KIND><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount, self);
	}
	else
	{
		wideStepKernel<0, RPH, SR, SL, IL, MODE, KIND><<<grid, block, lds, s>>>(c, g, a, pv, ops, opCount, self);
	}
}

// dynamic LDS beside the bodies, the ops and the three fixed records: parked rounds, the staged positions (self-contained), the
// round masks and the term table of the body-centric warm start
static size_t wideExtraLds(const PersistView& pv, int RPH, int SR, int SL, int IL, bool selfContained, bool bodyWarm, bool fixedArms = false, bool tgsLocals = false)
{
	size_t records = (size_t)S2_WIDE_PARKED_RECORDS * ((size_t)SL * pv.parkSeamWidth + (size_t)IL * pv.parkInteriorWidth);
	records += fixedArms ? (size_t)2 * (RPH + SR) * S2_WIDE_THREADS : 0; // SOFT_FIXED: {perp(rA0), perp(rB0)} per record, point and lane
	records += tgsLocals ? (size_t)2 * wideLocalsInLds(RPH, SR, SL, IL, true) * S2_WIDE_THREADS : 0; // s2Solve_TGS_Soft: {lA, lB} of the records that keep them in LDS
	records += selfContained ? (size_t)(pv.maxStaged + 1) / 2 : 0;
	if (bodyWarm)
	{
		const size_t tw = (size_t)pv.maxStripBodies, rounds = (size_t)(2 * RPH + SR);
		records += (tw + 3) / 4 + (3 * rounds * tw + 1) / 2;
	}
	return records * sizeof(float4);
}

template <int RPH, int SR, int SL = 0, int IL = 0>
static void launchWide(hipStream_t s, dim3 grid, size_t lds, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv, const Op* ops, int opCount,
					   const WideSelf* self, int kind)
{
	if (kind == SOFT_FIXED)
	{
		// s2Solve_SoftStep: the <3, 2> layout (wideVariant) -- its record keeps rA0 / rB0 in LDS beside the TGS record in registers, and a
		// sixth such record fits neither
		if constexpr (RPH == 3 && SR == 2 && SL == 0 && IL == 0)
		{
			const WideSelf none{};
			lds += wideExtraLds(pv, RPH, SR, SL, IL, false, false, true);
			if (pv.overflowKernel != 0)
			{
				launchWideMode<RPH, SR, SL, IL, S2_WIDE_OVERFLOW, SOFT_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount, none);
			}
			else if (pv.clearOwn != 0)
			{
				launchWideMode<RPH, SR, SL, IL, S2_WIDE_SLICED, SOFT_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount, none);
			}
			else
			{
				launchWideMode<RPH, SR, SL, IL, 0, SOFT_FIXED>(s, grid, lds, c, g, a, pv, ops, opCount, none);
			}
		}
		return;
	}
	if (kind == SOFT_PGS)
	{
		// s2Solve_PGS_Soft: the plain form only (prologue and epilogue launches, the coloured warm start); rA0 / rB0 in LDS where no round is parked
		const WideSelf none{};
		lds += wideExtraLds(pv, RPH, SR, SL, IL, false, false, true);
		if (pv.overflowKernel != 0)
		{
			launchWideMode<RPH, SR, SL, IL, S2_WIDE_OVERFLOW, SOFT_PGS>(s, grid, lds, c, g, a, pv, ops, opCount, none);
		}
		else if (pv.clearOwn != 0)
		{
			launchWideMode<RPH, SR, SL, IL, S2_WIDE_SLICED, SOFT_PGS>(s, grid, lds, c, g, a, pv, ops, opCount, none);
		}
		else
		{
			launchWideMode<RPH, SR, SL, IL, 0, SOFT_PGS>(s, grid, lds, c, g, a, pv, ops, opCount, none);
		}
		return;
	}
	// (the self-contained form and the body-centric warm start -- options, measured no faster -- exist for the <3, 2> layout: beside a
	// sixth record or parked rounds they spilled up to 520 bytes per lane: wideExtraRecords says so to the caller)
	constexpr bool OPTIONAL_MODES = RPH == 3 && SR == 2 && SL == 0 && IL == 0;
	const bool selfContained = OPTIONAL_MODES && self != nullptr;
	const bool bodyWarm = OPTIONAL_MODES && pv.bodyWarm != 0;
	lds += wideExtraLds(pv, RPH, SR, SL, IL, selfContained, bodyWarm, false, true);
	const WideSelf none{};
	if constexpr (OPTIONAL_MODES)
	{
		if (selfContained && bodyWarm && pv.allTwoPoints)
		{
			// (both at once for manifolds of two points only: with per-point masking the combination spilled 24 bytes per lane; mixed
			// point counts take the self-contained form with the coloured warm start)
			wideStepKernel<2, RPH, SR, SL, IL, S2_WIDE_SELF | S2_WIDE_BODYWARM><<<grid, dim3(S2_WIDE_THREADS), lds, s>>>(c, g, a, pv, ops, opCount, *self);
			return;
		}
		if (bodyWarm && !selfContained)
		{
			launchWideMode<RPH, SR, SL, IL, S2_WIDE_BODYWARM>(s, grid, lds, c, g, a, pv, ops, opCount, none);
			return;
		}
	}
	if constexpr (OPTIONAL_MODES)
	{
		if (selfContained)
		{
			launchWideMode<RPH, SR, SL, IL, S2_WIDE_SELF>(s, grid, lds, c, g, a, pv, ops, opCount, *self);
			return;
		}
	}
	{
		if (pv.overflowKernel != 0)
		{
			launchWideMode<RPH, SR, SL, IL, S2_WIDE_OVERFLOW>(s, grid, lds, c, g, a, pv, ops, opCount, none);
		}
		else if (pv.clearOwn != 0)
		{
			launchWideMode<RPH, SR, SL, IL, S2_WIDE_SLICED>(s, grid, lds, c, g, a, pv, ops, opCount, none);
		}
		else
		{
			launchWideMode<RPH, SR, SL, IL, 0>(s, grid, lds, c, g, a, pv, ops, opCount, none);
		}
	}
}

// which variant takes the partition: RPH/SR/SL/IL as an index 0..4, -1: none
static int wideVariant(const PersistView& pv)
{
	if (pv.maxRoundsA > 8 || pv.maxSeamRounds > 4)
	{
		return -1;
	}
	const bool park = (pv.debugSkip & 16) != 0; // tests: the variants with parked seam rounds whatever the partition needs
	if (park)
	{
		return pv.maxRoundsA <= 6 ? 3 : 4;
	}
	if (pv.maxRoundsA <= 6 && pv.maxSeamRounds <= 2)
	{
		return 0; // <3, 2>
	}
	if (pv.maxRoundsA <= 6 && pv.maxSeamRounds <= 3)
	{
		return 1; // <3, 3>
	}
	if (pv.maxSeamRounds <= 2)
	{
		return 2; // <4, 2>
	}
	// (seven or eight interior colours AND three or four seam colours -- a pile after an impact: six interior and two seam rounds in
	// registers, the rest parked -- the <4, 2, 2> layout of eight interior records per lane pair spilled 41 registers)
	return pv.maxRoundsA <= 6 ? 3 : 4; // <3, 2, 2>, <3, 2, 2, 2>
}

// records of dynamic LDS the variant for this partition needs beside the bodies, the ops and the three fixed records; -1: no variant
// takes that partition (Executor::widePlan)
int wideExtraRecords(const PersistView& pv, int selfContained, int bodyWarm, int kind)
{
	static const int shape[5][4] = {{3, 2, 0, 0}, {3, 3, 0, 0}, {4, 2, 0, 0}, {3, 2, 2, 0}, {3, 2, 2, 2}};
	const int v = wideVariant(pv);
	if (v < 0 || (kind == SOFT_FIXED && v > 0) || ((selfContained != 0 || bodyWarm != 0) && v > 0))
	{
		return -1; // (s2Solve_SoftStep, the self-contained form and the body-centric warm start: the <3, 2> layout only)
	}
	if (kind == SOFT_FIXED)
	{
		return (int)(wideExtraLds(pv, shape[v][0], shape[v][1], 0, 0, false, false, true) / sizeof(float4));
	}
	if (kind == SOFT_PGS)
	{
		return (int)(wideExtraLds(pv, shape[v][0], shape[v][1], shape[v][2], shape[v][3], false, false, true) / sizeof(float4));
	}
	const bool warm = bodyWarm != 0 && shape[v][2] == 0 && shape[v][3] == 0;
	return (int)(wideExtraLds(pv, shape[v][0], shape[v][1], shape[v][2], shape[v][3], selfContained != 0, warm, false, true) / sizeof(float4));
}

// ... and whether that variant has the body-centric warm start at all (the parked ones keep the coloured sweep)
int wideBodyWarmVariant(const PersistView& pv)
{
	const int v = wideVariant(pv);
	return v == 0 ? 1 : 0;
}

void launchWideStep(hipStream_t s, int kind, const ContactView& c, const BodyView& g, const StripTableView& a, const PersistView& pv, const Op* ops, int opCount, const WideSelf* self)
{
	const dim3 grid((unsigned)a.groupCount + (pv.overflowKernel != 0 ? 1u : 0u)); // (+ the overflow workgroup: wideOverflowWorker)
	const size_t lds = (size_t)(pv.bodyRecords + 3) * sizeof(float4) + (size_t)opCount * sizeof(Op);
	switch (wideVariant(pv))
	{
		case 0:
			launchWide<3, 2>(s, grid, lds, c, g, a, pv, ops, opCount, self, kind);
			break;
		case 1:
			launchWide<3, 3>(s, grid, lds, c, g, a, pv, ops, opCount, self, kind);
			break;
		case 2:
			launchWide<4, 2>(s, grid, lds, c, g, a, pv, ops, opCount, self, kind);
			break;
		case 3:
			launchWide<3, 2, 2>(s, grid, lds, c, g, a, pv, ops, opCount, self, kind);
			break;
		case 4:
			launchWide<3, 2, 2, 2>(s, grid, lds, c, g, a, pv, ops, opCount, self, kind);
			break;
		default:
			break;
	}
}

// ------------------------------------------------------------------------------------------------
// Resident islands on the same arithmetic (strip_kernel.hip: islandStepKernel<SOFT_TGS, WARM_CURRENT> is the general form): a
// group of small islands advanced through the whole s2Solve_TGS_Soft by one 512-thread workgroup, bodies in LDS, one WideRegs
// record per lane and colour round, records prepared from and impulses stored to the wire contacts.  Every lane of a round is
// busy here (a round holds up to 512 constraints), so prep and chain run back to back in the same lane; what this kernel takes
// from the strip kernel above is the shorter instruction stream -- the 22-dword record without an unpack step, the coefficient
// table, the chain on 2-vectors: the island kernel is VALU-issue bound, instructions are its time.
// ------------------------------------------------------------------------------------------------
// SELF: the kernel is also the step's body prologue and epilogue -- it stages its bodies straight from the wire records (the
// operations of body_ops.h: unpackBodyOne) and writes the owned ones back (packBodyOne) --, for a world that consists of
// resident islands only (BASELINE config 5): the step is this one launch.
// POINTS == 2: the host has checked that every constraint of the islands has two manifold points (no per-point masking: 356 against 383 us at config 5).
// (the eight-round variant keeps the local anchors of its last six records in LDS -- wideStepKernel: wideLocalsInLds --: eight
// 22-dword records spilled 140-172 bytes per lane)
constexpr int wideIslandLocalsInLds(int ROUNDS) { return ROUNDS > S2_STRIP_ROUNDS ? 6 : 0; }
int wideIslandLocalRecords(int maxRounds) { return 2 * wideIslandLocalsInLds(maxRounds > S2_STRIP_ROUNDS ? S2_STRIP_ROUNDS_MAX : S2_STRIP_ROUNDS) * S2_WIDE_THREADS; }

template <int ROUNDS, bool SELF, int POINTS>
__global__ __launch_bounds__(S2_WIDE_THREADS) void wideIslandKernel(ContactView c, BodyView g, StripTableView ta, float4 softCoef0, float4 softCoef1, const Op* ops,
																	 int opCount, s2amdContact* wire, s2amdBody* wireBodies, const uint32_t* hostFlags, int warmStart,
																	 StepConsts sc, float unpackH, const unsigned int* stepFailed)
{
	extern __shared__ __attribute__((aligned(16))) float4 lds[];
	if (stepFailed != nullptr && *stepFailed != 0u)
	{
		return; // a persistent strip kernel of this step lost a hand-off: the step will be repeated, nothing of it may reach the wire arrays
	}
	const int tid = (int)threadIdx.x;
	const StripDesc* da = ta.descs + blockIdx.x;
	const int bodyBase = da->bodyBase, nb = da->bodyCount, roundsA = da->batchCount;
	int2 batchA[ROUNDS];
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		batchA[i] = make_int2(da->batch[i].x, da->batch[i].y);
	}
	float4* lvel = lds;
	float4* ldq = lds + nb;
	float4* linteg = lds + 2 * nb;
	float* langDamp = (float*)(lds + 3 * nb);
	float2* lmass = (float2*)(lds + 3 * nb + (nb + 3) / 4);
	float2* llc = (float2*)(lds + 3 * nb + (nb + 3) / 4 + (nb + 1) / 2); // the bodies' local centres (soft_from_wire.h: prepareSoftFromWire)
	const int bodyRecords = 3 * nb + (nb + 3) / 4 + 2 * ((nb + 1) / 2);
	Op* lops = (Op*)(lds + bodyRecords);
	float4* lcoef = lds + bodyRecords + 2 * opCount; // 2 records (the launch adds them to the size)
	constexpr int LA = wideIslandLocalsInLds(ROUNDS), LL0 = ROUNDS - LA;
	float4* llocals = lcoef + 2 + tid; // [record - LL0][point][lane] {lA, lB}
	auto localsOf = [&](int r) { return llocals + 2 * (r - LL0 > 0 ? r - LL0 : 0) * S2_WIDE_THREADS; };

	uint32_t id[S2_STRIP_BODY_CHUNKS];
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		const int i = tid + ch * S2_WIDE_THREADS;
		id[ch] = i < nb ? (uint32_t)ta.bodyIds[bodyBase + i] : 0u;
	}
	for (int i = tid; i < opCount * 8; i += S2_WIDE_THREADS)
	{
		((int*)lops)[i] = ((const int*)ops)[i];
	}
	if (tid < 2)
	{
		lcoef[tid] = tid ? softCoef1 : softCoef0;
	}
	auto kOfRound = [&](int i) {
		const int k = batchA[i].x + tid;
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
	float2 pos[S2_STRIP_BODY_CHUNKS]; // SELF: the positions of the bodies this lane stages (s2FinalizePositions adds to them)
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		const int i = tid + ch * S2_WIDE_THREADS;
		flags[ch] = 0u;
		pos[ch] = make_float2(0.0f, 0.0f);
		if (i < nb)
		{
			const int gi = (int)(id[ch] & ~S2G_OWNED);
			if constexpr (SELF)
			{
				// body_ops.h: unpackBodyOne, into LDS instead of the SoA arrays
				const s2amdBody* w = wireBodies + gi;
				const int type = w->type;
				uint32_t f = 0x80000000u;
				if (type != S2AMD_BODY_FREE)
				{
					f |= S2F_LIVE | (type == S2AMD_BODY_DYNAMIC ? S2F_DYNAMIC : 0u) | (type != S2AMD_BODY_STATIC ? S2F_MOVES : 0u);
				}
				flags[ch] = f;
				lvel[i] = make_float4(w->linearVelocity[0], w->linearVelocity[1], w->angularVelocity, 0.0f);
				ldq[i] = make_float4(w->deltaPosition[0], w->deltaPosition[1], w->rot[0], w->rot[1]);
				pos[ch] = make_float2(w->position[0], w->position[1]);
				lmass[i] = make_float2(w->invMass, w->invI);
				llc[i] = make_float2(w->localCenter[0], w->localCenter[1]);
				const V2 gravity = v2(sc.gravityX, sc.gravityY);
				const V2 force = v2(w->force[0], w->force[1]);
				const V2 inner = mulAdd(force, w->mass * w->gravityScale, gravity);
				const V2 a = mulSV(unpackH * w->invMass, inner);
				const float aw = unpackH * w->invI * w->torque;
				const float ld = 1.0f / (1.0f + unpackH * w->linearDamping);
				const float ad = 1.0f / (1.0f + unpackH * w->angularDamping);
				linteg[i] = make_float4(a.x, a.y, aw, ld);
				langDamp[i] = ad;
			}
			else
			{
				lvel[i] = g.vel[gi];
				ldq[i] = g.dq[gi];
				flags[ch] = g.flags[gi] | 0x80000000u;
				linteg[i] = g.integ[gi];
				langDamp[i] = g.angDamp[gi];
				lmass[i] = g.massInv[gi];
				llc[i] = make_float2(wireBodies[gi].localCenter[0], wireBodies[gi].localCenter[1]);
			}
		}
	}
	__syncthreads();

	LdsBodies lb{lvel, ldq};
	WideRegs rA[ROUNDS];
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		if (slotOf[i] >= 0)
		{
			rA[i] = wideFromSoft(prepareSoftFromWire<SOFT_TGS>(wire + slotOf[i], wireBodies, hostFlags, lb, lmass, localOf[i], g.capacity, warmStart, llc));
			const bool st = lmass[localOf[i].x].x == 0.0f || lmass[localOf[i].y].x == 0.0f; // the doubled contact hertz of a static side
			rA[i].idx |= st ? 1u << 30 : 0u;
			if (i >= LL0)
			{
#pragma unroll
				for (int j = 0; j < 2; ++j)
				{
					localsOf(i)[j * S2_WIDE_THREADS] = make_float4(rA[i].lA[j].x, rA[i].lA[j].y, rA[i].lB[j].x, rA[i].lB[j].y);
				}
			}
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
					const int i = tid + ch * S2_WIDE_THREADS;
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
					const int i = tid + ch * S2_WIDE_THREADS;
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
				if constexpr (SELF)
				{
					// s2FinalizePositions (solve_common.c:70-91; body_ops.h: finalizePositionsOne) on the lane's own copy of the position
					if ((flags[ch] & (op.flag ? S2F_DYNAMIC : S2F_MOVES)) != 0)
					{
						const int i = tid + ch * S2_WIDE_THREADS;
						const float4 d = ldq[i];
						const V2 np = add(v2(pos[ch].x, pos[ch].y), v2(d.x, d.y));
						pos[ch] = make_float2(np.x, np.y);
						ldq[i] = make_float4(0.0f, 0.0f, d.z, d.w);
					}
				}
				else if (flags[ch] != 0u)
				{
					finalizePositionsOne(lb, tid + ch * S2_WIDE_THREADS, g, (int)(id[ch] & ~S2G_OWNED), op.flag, (id[ch] & S2G_OWNED) != 0);
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
						if (i >= LL0)
						{
							warmWide<SOFT_TGS, POINTS, true>(rA[i], lvel, ldq, lmass, salt, nullptr, localsOf(i));
						}
						else
						{
							warmWide<SOFT_TGS, POINTS>(rA[i], lvel, ldq, lmass, salt);
						}
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
						const WidePrep pre = (i >= LL0) ? prepWide<SOFT_TGS, POINTS, true>(rA[i], ldq, lcoef, op.inv_h, op.useBias, salt, nullptr, localsOf(i))
													  : prepWide<SOFT_TGS, POINTS>(rA[i], ldq, lcoef, op.inv_h, op.useBias, salt);
						chainWide<POINTS>(rA[i], pre, lvel, lmass, lcoef, salt);
					}
					__syncthreads();
				}
			}
		}
	}
#pragma unroll
	for (int ch = 0; ch < S2_STRIP_BODY_CHUNKS; ++ch)
	{
		const int i = tid + ch * S2_WIDE_THREADS;
		if (i < nb && (id[ch] & S2G_OWNED) != 0)
		{
			const int gi = (int)(id[ch] & ~S2G_OWNED);
			if constexpr (SELF)
			{
				if ((flags[ch] & S2F_LIVE) != 0) // body_ops.h: packBodyOne
				{
					s2amdBody* w = wireBodies + gi;
					const float4 v = lvel[i], d = ldq[i];
					w->position[0] = pos[ch].x, w->position[1] = pos[ch].y;
					w->rot[0] = d.z, w->rot[1] = d.w;
					w->linearVelocity[0] = v.x, w->linearVelocity[1] = v.y;
					w->angularVelocity = v.z;
					w->deltaPosition[0] = d.x, w->deltaPosition[1] = d.y;
				}
			}
			else
			{
				g.vel[gi] = lvel[i];
				g.dq[gi] = ldq[i];
			}
		}
	}
	// s2StoreContactImpulses (solve_common.c:396-410): straight into the manifolds
#pragma unroll
	for (int i = 0; i < ROUNDS; ++i)
	{
		if (slotOf[i] >= 0)
		{
			const int pointCount = (int)((rA[i].idx >> 26) & 3u);
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

// t.ldsRecords: body records of the largest group; maxRounds: colour rounds of the group with the most
template <int ROUNDS>
static void launchWideIslandRounds(hipStream_t s, dim3 grid, size_t lds, const ContactView& c, const BodyView& g, const StripTableView& t, const float4* softCoef,
								   const Op* ops, int opCount, s2amdContact* wire, s2amdBody* wireBodies, const uint32_t* hostFlags, int warmStart, const StepConsts& sc,
								   float unpackH, int selfContained, const unsigned int* stepFailed, int allTwoPoints)
{
#define S2_LAUNCH_ISLAND(SELF, POINTS)                                                                                                                        \
	wideIslandKernel<ROUNDS, SELF, POINTS>                                                                                                                    \
		<<<grid, dim3(S2_WIDE_THREADS), lds, s>>>(c, g, t, softCoef[0], softCoef[1], ops, opCount, wire, wireBodies, hostFlags, warmStart, sc, unpackH, stepFailed)
	if (selfContained)
	{
		if (allTwoPoints)
		{
			S2_LAUNCH_ISLAND(true, 2);
		}
		else
		{
			S2_LAUNCH_ISLAND(true, 0);
		}
	}
	else if (allTwoPoints)
	{
		S2_LAUNCH_ISLAND(false, 2);
	}
	else
	{
		S2_LAUNCH_ISLAND(false, 0);
	}
#undef S2_LAUNCH_ISLAND
}

void launchWideIsland(hipStream_t s, const ContactView& c, const BodyView& g, const StripTableView& t, const float4* softCoef, const Op* ops, int opCount,
					  int maxRounds, s2amdContact* wire, s2amdBody* wireBodies, const uint32_t* hostFlags, int warmStart, const StepConsts& sc, float unpackH,
					  int selfContained, const unsigned int* stepFailed, int allTwoPoints)
{
	const dim3 grid((unsigned)t.groupCount);
	const size_t lds = (size_t)(t.ldsRecords + 2 + wideIslandLocalRecords(maxRounds)) * sizeof(float4) + (size_t)opCount * sizeof(Op);
	if (maxRounds <= S2_STRIP_ROUNDS)
	{
		launchWideIslandRounds<S2_STRIP_ROUNDS>(s, grid, lds, c, g, t, softCoef, ops, opCount, wire, wireBodies, hostFlags, warmStart, sc, unpackH, selfContained, stepFailed, allTwoPoints);
	}
	else
	{
		launchWideIslandRounds<S2_STRIP_ROUNDS_MAX>(s, grid, lds, c, g, t, softCoef, ops, opCount, wire, wireBodies, hostFlags, warmStart, sc, unpackH, selfContained, stepFailed, allTwoPoints);
	}
}

int wideKernelSetup()
{
	for (const void* f : {(const void*)wideIslandKernel<S2_STRIP_ROUNDS, false, 0>, (const void*)wideIslandKernel<S2_STRIP_ROUNDS_MAX, false, 0>,
						  (const void*)wideIslandKernel<S2_STRIP_ROUNDS, true, 0>, (const void*)wideIslandKernel<S2_STRIP_ROUNDS_MAX, true, 0>,
						  (const void*)wideIslandKernel<S2_STRIP_ROUNDS, false, 2>, (const void*)wideIslandKernel<S2_STRIP_ROUNDS_MAX, false, 2>,
						  (const void*)wideIslandKernel<S2_STRIP_ROUNDS, true, 2>, (const void*)wideIslandKernel<S2_STRIP_ROUNDS_MAX, true, 2>})
	{
		if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
		{
			return 1;
		}
	}
	// every variant the launch can pick (launchWide): the five layouts in the plain, the sliced and the overflow form for s2Solve_TGS_Soft and
	// s2Solve_PGS_Soft, the <3, 2> layout alone for s2Solve_SoftStep and for the two optional modes
#define S2_WIDE_LAYOUTS(P, MODE, KIND)                                                                                            \
	(const void*)wideStepKernel<P, 3, 2, 0, 0, MODE, KIND>, (const void*)wideStepKernel<P, 3, 3, 0, 0, MODE, KIND>, (const void*)wideStepKernel<P, 4, 2, 0, 0, MODE, KIND>, \
		(const void*)wideStepKernel<P, 3, 2, 2, 0, MODE, KIND>, (const void*)wideStepKernel<P, 3, 2, 2, 2, MODE, KIND>
	const void* steps[] = {S2_WIDE_LAYOUTS(0, 0, SOFT_TGS), S2_WIDE_LAYOUTS(2, 0, SOFT_TGS), S2_WIDE_LAYOUTS(0, S2_WIDE_SLICED, SOFT_TGS), S2_WIDE_LAYOUTS(2, S2_WIDE_SLICED, SOFT_TGS),
						   S2_WIDE_LAYOUTS(0, S2_WIDE_OVERFLOW, SOFT_TGS), S2_WIDE_LAYOUTS(2, S2_WIDE_OVERFLOW, SOFT_TGS),
						   (const void*)wideStepKernel<0, 3, 2, 0, 0, 1>, (const void*)wideStepKernel<0, 3, 2, 0, 0, 2>,
						   (const void*)wideStepKernel<2, 3, 2, 0, 0, 1>, (const void*)wideStepKernel<2, 3, 2, 0, 0, 2>, (const void*)wideStepKernel<2, 3, 2, 0, 0, 3>};
	const void* pgs[] = {S2_WIDE_LAYOUTS(0, 0, SOFT_PGS), S2_WIDE_LAYOUTS(2, 0, SOFT_PGS), S2_WIDE_LAYOUTS(0, S2_WIDE_SLICED, SOFT_PGS), S2_WIDE_LAYOUTS(2, S2_WIDE_SLICED, SOFT_PGS),
						 S2_WIDE_LAYOUTS(0, S2_WIDE_OVERFLOW, SOFT_PGS), S2_WIDE_LAYOUTS(2, S2_WIDE_OVERFLOW, SOFT_PGS)};
#undef S2_WIDE_LAYOUTS
	const void* fixed[] = {(const void*)wideStepKernel<0, 3, 2, 0, 0, 0, SOFT_FIXED>, (const void*)wideStepKernel<2, 3, 2, 0, 0, 0, SOFT_FIXED>,
						   (const void*)wideStepKernel<0, 3, 2, 0, 0, S2_WIDE_SLICED, SOFT_FIXED>, (const void*)wideStepKernel<2, 3, 2, 0, 0, S2_WIDE_SLICED, SOFT_FIXED>,
						   (const void*)wideStepKernel<0, 3, 2, 0, 0, S2_WIDE_OVERFLOW, SOFT_FIXED>, (const void*)wideStepKernel<2, 3, 2, 0, 0, S2_WIDE_OVERFLOW, SOFT_FIXED>};
	for (const void* f : fixed)
	{
		if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
		{
			return 1;
		}
	}
	for (const void* f : pgs)
	{
		if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
		{
			return 1;
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
	return 0;
}
#endif // S2_WIDE_ONLY_MAIN

S2_DEFINE_WARM(wide_kernel)

// A queue's scratch memory is allocated by the runtime when the first kernel with a private frame is dispatched on it -- a stall of the
// kind S2_DEFINE_WARM exists for.  The overflow variants above declare a (never touched) 36-byte frame and the 256-thread fall-back
// kernels spill for real: s2amd_create dispatches this kernel once, with a frame of 64 bytes per lane on a grid of the persistent
// kernels' shape, so that the allocation happens there.
__global__ __launch_bounds__(S2_WIDE_THREADS) void s2WarmScratchKernel(int* sink, int n)
{
	volatile int a[16];
	for (int i = 0; i < 16; ++i)
	{
		a[i] = (int)threadIdx.x + i;
	}
	int sum = 0;
	for (int i = 0; i < n; ++i)
	{
		sum += a[(i * 7) & 15];
	}
	if (sink != nullptr && sum == 0x7fffffff)
	{
		*sink = sum;
	}
}
void s2WarmScratch(hipStream_t st) { s2WarmScratchKernel<<<dim3(256), dim3(S2_WIDE_THREADS), 0, st>>>(nullptr, 4); }
