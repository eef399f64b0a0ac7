// Per-constraint device functions ("One" = one constraint of one sweep), shared by the global
// colour-batch kernels (contact_kernels.hip, joint_kernels.hip: bodies gathered from HBM/L2) and by
// the group kernel (group_kernel.hip: bodies of a small island, or of the sequential tail, staged in
// LDS).  B is the body accessor: BA::kMode selects how body indices are obtained, getVel/setVel
// and getDq/setDq read and write the two 16-byte body records.
#pragma once

#include "launch.h"
#include "s2_device.h"

#include "solver2d_amd.h"

// index modes of a body accessor
#define S2_IDX_GLOBAL 0	 // body-pool slots, gathered through c.bodies[k]
#define S2_IDX_LOCAL 1	 // group-local slots (LDS), gathered through c.localBodies[k]
#define S2_IDX_MESSAGE 2 // per-constraint copies: side A of constraint k is record 2k, side B is 2k+1

template <bool LOCAL> struct BodiesT
{
	static constexpr int kMode = LOCAL ? S2_IDX_LOCAL : S2_IDX_GLOBAL;
	static constexpr bool kLdsMass = false;
	float4* vel;
	float4* dq;
	S2_DEV float4 getVel(int i) const { return vel[i]; }
	S2_DEV void setVel(int i, float4 v) const { vel[i] = v; }
	S2_DEV float4 getDq(int i) const { return dq[i]; }
	S2_DEV void setDq(int i, float4 v) const { dq[i] = v; }
};
// Keeps already-issued loads from being sunk into later branches by the compiler: the value must be
// in its registers here.  Placed after the LAST load of a group so all of them share one wait.
S2_DEV void pin(float4& v)
{
	asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}
S2_DEV void pin(float2& v)
{
	asm volatile("" : "+v"(v.x), "+v"(v.y));
}

typedef BodiesT<false> GlobalBodies; // indices are body-pool slots, arrays are the HBM SoA
typedef BodiesT<true> LdsBodies;	 // indices are group-local slots, arrays live in LDS

// The group kernel's accessor: besides the two body records, the inverse mass and inertia of every staged body
// sit in LDS, so a constraint sweep does not stream its own copy of them (c.mass, 16 B per constraint and
// sweep) from HBM/L2.  When the plan prepared the contacts with PREP_SOFT, the three soft coefficients are one
// of two step-wide triples (solve_common.c:219, 262-271: they depend only on h, the contact hertz and on
// whether a side is static), so c.soft (32 B) is not streamed either: softCoef[0] is the triple of a
// dynamic-dynamic constraint, softCoef[1] of one with a static side; softDiet == 0 keeps the loads.
struct LdsMassBodies : BodiesT<true>
{
	static constexpr bool kLdsMass = true;
	const float2* massInv; // {invMass, invI} per local slot
	float4 softCoef[2];
	int softDiet;
};

// "Message passing" accessor for the big-island path.  Every constraint side owns a private copy of
// its body's two records, so a sweep kernel reads them at a computed address (no index load, no
// dependent gather: one memory round trip instead of two) and writes the updated velocity into the
// copy of the NEXT constraint that will touch that body (next[] is cyclic over the body's touches
// in sweep order, so the chain carries over from sweep to sweep).  Poses only change in the body
// kernels, which write them to every copy of the body.
struct MsgBodies
{
	static constexpr int kMode = S2_IDX_MESSAGE;
	static constexpr bool kLdsMass = false;
	float4* vel;	 // [2C]
	float4* dq;		 // [2C]
	const int* next; // [2C]
	S2_DEV float4 getVel(int i) const { return vel[i]; }
	S2_DEV void setVel(int i, float4 v) const { vel[next[i]] = v; }
	S2_DEV float4 getDq(int i) const { return dq[i]; }
	S2_DEV void setDq(int i, float4 v) const { dq[next[i]] = v; } // not used: position sweeps never run in this mode
};

struct CHeader
{
	int ia, ib;
	float mA, iA, mB, iB;
	V2 normal;
	float friction;
	int pointCount;
	bool writeA, writeB;
};

template <int MODE> S2_DEV CHeader loadHeader(const ContactView& c, int k)
{
	CHeader h;
	int2 b = MODE == S2_IDX_MESSAGE ? make_int2(2 * k, 2 * k + 1) : (MODE == S2_IDX_LOCAL ? c.localBodies[k] : c.bodies[k]);
	float4 m = c.mass[k];
	float4 nf = c.nf[k];
	h.ia = b.x, h.ib = b.y;
	h.mA = m.x, h.iA = m.y, h.mB = m.z, h.iB = m.w;
	h.normal = v2(nf.x, nf.y);
	h.friction = nf.z;
	uint32_t bits = asBits(nf.w);
	h.pointCount = (int)(bits & 0xffu);
	h.writeA = (bits & S2C_WRITE_A) != 0;
	h.writeB = (bits & S2C_WRITE_B) != 0;
	return h;
}

// header through the accessor: the LDS-mass accessor looks the masses up by local slot
template <class BA> S2_DEV CHeader loadHeaderB(const ContactView& c, const BA& b, int k)
{
	if constexpr (BA::kLdsMass)
	{
		CHeader h;
		int2 ib = c.localBodies[k];
		float4 nf = c.nf[k];
		h.ia = ib.x, h.ib = ib.y;
		float2 a = b.massInv[ib.x], bb = b.massInv[ib.y];
		h.mA = a.x, h.iA = a.y, h.mB = bb.x, h.iB = bb.y;
		h.normal = v2(nf.x, nf.y);
		h.friction = nf.z;
		uint32_t bits = asBits(nf.w);
		h.pointCount = (int)(bits & 0xffu);
		h.writeA = (bits & S2C_WRITE_A) != 0;
		h.writeB = (bits & S2C_WRITE_B) != 0;
		return h;
	}
	else
	{
		return loadHeader<BA::kMode>(c, k);
	}
}

struct BodyVel
{
	V2 v;
	float w;
};
struct BodyPose
{
	V2 dc;
	Rot q;
};

template <class BA> S2_DEV BodyVel loadVel(const BA& b, int i)
{
	float4 t = b.getVel(i);
	BodyVel r;
	r.v = v2(t.x, t.y);
	r.w = t.z;
	return r;
}
template <class BA> S2_DEV void storeVel(const BA& b, int i, V2 v, float w)
{
	b.setVel(i, make_float4(v.x, v.y, w, 0.0f));
}
template <class BA> S2_DEV BodyPose loadPose(const BA& b, int i)
{
	float4 t = b.getDq(i);
	BodyPose r;
	r.dc = v2(t.x, t.y);
	r.q.s = t.z, r.q.c = t.w;
	return r;
}
template <class BA> S2_DEV void storePose(const BA& b, int i, V2 dc, Rot q)
{
	b.setDq(i, make_float4(dc.x, dc.y, q.s, q.c));
}

// ---------------------------------------------------------------------------------------------
// warm start: s2WarmStartContacts (solve_common.c:276-326, current anchors),
// s2WarmStartContacts_Fixed (solve_soft_step.c:16-63), and the second loop of
// s2CreateContactSolver (solve_pgs_ngs_block.c:279-319, fixed anchors, reduced point count)
// ---------------------------------------------------------------------------------------------
// one constraint's warm-start data in registers: loading is separate from the arithmetic so the sequential tail
// (group_kernel.hip: walkTail) can have a whole wave issue its loads at once
struct WarmRegs
{
	CHeader h;
	float4 arm[2];
	float2 imp[2];
	V2 tangent;
	int pointCount;
};

template <int KIND, class BA> S2_DEV WarmRegs loadWarm(const ContactView& c, const BA& b, int k)
{
	WarmRegs r;
	r.h = loadHeaderB(c, b, k);
	r.pointCount = r.h.pointCount;
	r.tangent = rightPerp(r.h.normal);
	if (KIND == WARM_BLOCK)
	{
		r.pointCount = (int)asBits(c.blockK[k].w);
		r.tangent = crossVS(r.h.normal, 1.0f);
	}
	// every load of the constraint is issued before the first use: slot 1 of a one-point constraint is
	// a valid, zero-filled record (prepareContactsKernel), so the loads need no guard
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		r.arm[j] = KIND == WARM_CURRENT ? c.anchor[j][k] : c.r0[j][k];
		r.imp[j] = c.impulse[j][k];
	}
	return r;
}

template <int KIND, class BA> S2_DEV void applyWarm(WarmRegs& r, const BA& b)
{
	const CHeader& h = r.h;
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	Rot qA, qB;
	if (KIND == WARM_CURRENT)
	{
		qA = loadPose(b, h.ia).q;
		qB = loadPose(b, h.ib).q;
	}
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		pin(r.arm[j]);
		pin(r.imp[j]);
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < r.pointCount)
		{
			V2 rA, rB;
			if (KIND == WARM_CURRENT)
			{
				rA = rotate(qA, v2(r.arm[j].x, r.arm[j].y));
				rB = rotate(qB, v2(r.arm[j].z, r.arm[j].w));
			}
			else
			{
				rA = v2(r.arm[j].x, r.arm[j].y);
				rB = v2(r.arm[j].z, r.arm[j].w);
			}
			V2 P = add(mulSV(r.imp[j].x, h.normal), mulSV(r.imp[j].y, r.tangent));
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

template <int KIND, class BA>
S2_DEV void warmStartContactsOne(const ContactView& c, const BA& b, int k)
{
	WarmRegs r = loadWarm<KIND>(c, b, k);
	applyWarm<KIND>(r, b);
}

// ---------------------------------------------------------------------------------------------
// soft velocity sweeps:
//   SOFT_TGS    s2SolveContacts_TGS_Soft     solve_tgs_soft.c:17-135
//   SOFT_PGS    s2SolveContacts_PGS_Soft     solve_pgs_soft.c:16-125
//   SOFT_JACOBI s2SolveContacts_Jacobi_Soft  solve_jacobi.c:21-132  (writes per-constraint deltas)
//   SOFT_FIXED  s2SolveContacts_TGS_Fixed    solve_soft_step.c:66-177
// ---------------------------------------------------------------------------------------------
// One constraint's sweep data in registers.  Loading is separate from the arithmetic so a workgroup that
// owns several colour batches (group_kernel.hip: preloaded rounds) can issue all its loads at once.
template <int KIND> struct SoftRegs
{
	CHeader h;
	float4 an[2], r0[2], par[2], sf[2];
	float2 imp[2];
};

template <int KIND, int MODE> S2_DEV SoftRegs<KIND> loadSoft(const ContactView& c, int k)
{
	SoftRegs<KIND> r;
	r.h = loadHeader<MODE>(c, k);
	// slot 1 of a one-point constraint is a valid zero record, so no guard is needed
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (KIND == SOFT_TGS || KIND == SOFT_FIXED)
		{
			r.an[j] = c.anchor[j][k];
		}
		if (KIND != SOFT_TGS)
		{
			r.r0[j] = c.r0[j][k];
		}
		r.par[j] = c.param[j][k];
		r.sf[j] = c.soft[j][k];
		r.imp[j] = c.impulse[j][k];
	}
	return r;
}

// Loads through the accessor.  With the LDS-mass accessor the header's masses and (softDiet) the soft
// coefficients are NOT loaded here: completeSoft fills them at solve time, next to the body reads, so a
// preloaded chunk of rounds has no LDS lookup hanging on its first global load.
template <int KIND, class BA> S2_DEV SoftRegs<KIND> loadSoftB(const ContactView& c, const BA& b, int k)
{
	if constexpr (BA::kLdsMass)
	{
		SoftRegs<KIND> r;
		int2 ib = c.localBodies[k];
		float4 nf = c.nf[k];
		r.h.ia = ib.x, r.h.ib = ib.y;
		r.h.mA = r.h.iA = r.h.mB = r.h.iB = 0.0f;
		r.h.normal = v2(nf.x, nf.y);
		r.h.friction = nf.z;
		uint32_t bits = asBits(nf.w);
		r.h.pointCount = (int)(bits & 0xffu);
		r.h.writeA = (bits & S2C_WRITE_A) != 0;
		r.h.writeB = (bits & S2C_WRITE_B) != 0;
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (KIND == SOFT_TGS || KIND == SOFT_FIXED)
			{
				r.an[j] = c.anchor[j][k];
			}
			if (KIND != SOFT_TGS)
			{
				r.r0[j] = c.r0[j][k];
			}
			r.par[j] = c.param[j][k];
			if (b.softDiet == 0)
			{
				r.sf[j] = c.soft[j][k];
			}
			r.imp[j] = c.impulse[j][k];
		}
		return r;
	}
	else
	{
		return loadSoft<KIND, BA::kMode>(c, k);
	}
}

template <int KIND, class BA> S2_DEV void completeSoft(SoftRegs<KIND>& r, const BA& b)
{
	if constexpr (BA::kLdsMass)
	{
		float2 a = b.massInv[r.h.ia], bb = b.massInv[r.h.ib];
		r.h.mA = a.x, r.h.iA = a.y, r.h.mB = bb.x, r.h.iB = bb.y;
		if (b.softDiet)
		{
			// contact_kernels.hip prepareContactsKernel<PREP_SOFT>: contactHertz doubles when a side is static
			float4 sf = (a.x == 0.0f || bb.x == 0.0f) ? b.softCoef[1] : b.softCoef[0];
			r.sf[0] = sf;
			r.sf[1] = sf;
		}
	}
}

template <int KIND> S2_DEV void pinSoft(SoftRegs<KIND>& r)
{
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (KIND == SOFT_TGS || KIND == SOFT_FIXED)
		{
			pin(r.an[j]);
		}
		if (KIND != SOFT_TGS)
		{
			pin(r.r0[j]);
		}
		pin(r.par[j]);
		pin(r.sf[j]);
		pin(r.imp[j]);
	}
}

// solveSoftRegs (below) in two parts, for the sequential tail (group_kernel.hip: walkTail).  prepSoft: everything that does not
// depend on the bodies' VELOCITIES -- the anchors in world orientation, the current separation (poses only change between
// sweeps), the bias / mass scale / impulse scale selected from it: all lanes of the tail's wave at once.  chainSoft: the
// dependent chain -- relative velocity, impulse, clamp, apply, point after point, normal then friction: lane after lane.
// The same operations on the same operands as solveSoftRegs, which stays one function: the resident kernels' register
// allocation is tuned around it (splitting it there cost the headline kernel 17 us).
struct SoftPre
{
	V2 rA[2], rB[2];
	float bias[2], massScale[2], impulseScale[2];
};

// POINTS == 2: the caller has checked that the constraint has two points (a wave-uniform fast path without
// per-point exec masking); POINTS == 0: per-point guards on h.pointCount.
template <int KIND, class BA, bool PIN = true, int POINTS = 0>
S2_DEV SoftPre prepSoft(SoftRegs<KIND>& r, const BA& b, float inv_h, int useBias)
{
	completeSoft(r, b);
	const CHeader& h = r.h;
	const float biasCap = (KIND == SOFT_TGS || KIND == SOFT_JACOBI) ? -S2_MAX_BAUMGARTE_VELOCITY : -0.5f * S2_MAX_BAUMGARTE_VELOCITY;
	float4* an = r.an;
	float4* r0 = r.r0;
	float4* par = r.par;
	float4* sf = r.sf;
	V2 dcA, dcB;
	Rot qA, qB;
	if (KIND == SOFT_TGS || KIND == SOFT_FIXED)
	{
		BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
		dcA = pA.dc, qA = pA.q, dcB = pB.dc, qB = pB.q;
	}
	if (PIN)
	{
		pinSoft(r);
	}
	V2 normal = h.normal;
	SoftPre pre;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < h.pointCount)
		{
			V2 rA, rB;
			float s;
			if (KIND == SOFT_TGS)
			{
				rA = rotate(qA, v2(an[j].x, an[j].y));
				rB = rotate(qB, v2(an[j].z, an[j].w));
				V2 ds = add(sub(dcB, dcA), sub(rB, rA));
				s = dot(ds, normal) + par[j].x;
			}
			else if (KIND == SOFT_FIXED)
			{
				V2 ds = add(sub(dcB, dcA), sub(rotate(qB, v2(an[j].z, an[j].w)), rotate(qA, v2(an[j].x, an[j].y))));
				s = dot(ds, normal) + par[j].x;
				rA = v2(r0[j].x, r0[j].y);
				rB = v2(r0[j].z, r0[j].w);
			}
			else
			{
				s = par[j].w;
				rA = v2(r0[j].x, r0[j].y);
				rB = v2(r0[j].z, r0[j].w);
			}
			pre.rA[j] = rA, pre.rB[j] = rB;

			// select form of: if (s > 0) bias = s * inv_h; else if (useBias) {bias = max(biasCoefficient * s, cap); ...}
			const bool speculative = s > 0.0f;
			const bool soft = !speculative && useBias != 0;
			float softBias = S2_MAXF(sf[j].x * s, biasCap);
			pre.bias[j] = speculative ? s * inv_h : (soft ? softBias : 0.0f);
			pre.massScale[j] = soft ? sf[j].y : 1.0f;
			pre.impulseScale[j] = soft ? sf[j].z : 0.0f;
		}
	}
	return pre;
}

template <int KIND, class BA, int POINTS = 0>
S2_DEV void chainSoft(SoftRegs<KIND>& r, const SoftPre& pre, const ContactView& c, const BA& b, int k)
{
	const CHeader& h = r.h;
	float4* par = r.par;
	float2* imp = r.imp;
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 normal = h.normal;
	V2 tangent = rightPerp(normal);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float nImp[2], tImp[2];

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < h.pointCount)
		{
			V2 rA = pre.rA[j], rB = pre.rB[j];
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vn = dot(sub(vrB, vrA), normal);

			float impulse = -par[j].y * pre.massScale[j] * (vn + pre.bias[j]) - pre.impulseScale[j] * imp[j].x;
			float newImpulse = S2_MAXF(imp[j].x + impulse, 0.0f);
			impulse = newImpulse - imp[j].x;
			nImp[j] = newImpulse;
			tImp[j] = imp[j].y;

			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < h.pointCount)
		{
			float tangentMass = par[j].z;
			V2 rA = pre.rA[j], rB = pre.rB[j];
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vt = dot(sub(vrB, vrA), tangent);
			float impulse = -tangentMass * vt;
			float maxFriction = h.friction * nImp[j];
			float newImpulse = S2_CLAMPF(tImp[j] + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - tImp[j];
			tImp[j] = newImpulse;
			V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
			imp[j] = make_float2(nImp[j], tImp[j]);
		}
	}

	if (KIND == SOFT_JACOBI)
	{
		// solve_jacobi.c:126-130: the body sums these in constraint order (jacobiApplyKernel)
		V2 dA = sub(vA, A.v), dB = sub(vB, B.v);
		c.deltaA[k] = make_float4(dA.x, dA.y, wA - A.w, 0.0f);
		c.deltaB[k] = make_float4(dB.x, dB.y, wB - B.w, 0.0f);
	}
	else
	{
		if (h.writeA)
		{
			storeVel(b, h.ia, vA, wA);
		}
		if (h.writeB)
		{
			storeVel(b, h.ib, vB, wB);
		}
	}
}

// chainSoft on explicit 2-vectors over the perpendicular anchors (wide_kernel.hip: chainWide's formulation -- crossSV(w, r) = w perp(r),
// cross(r, P) = perp(r).x P.x + perp(r).y P.y, term for term the reference's products: (-a) b == -(a b), x - y == x + (-y) -- so that the
// vector algebra is v_pk_mul_f32 / v_pk_add_f32 without moves).  For the sequential tail (group_kernel.hip: walkTail), where a turn is
// ONE lane's instruction stream and nothing else runs: every instruction saved is four cycles of the hub's 238 x 24 visits per step.
typedef float pk2 __attribute__((ext_vector_type(2)));
S2_DEV float pkDot(pk2 a, pk2 b)
{
	const pk2 m = a * b;
	return m.x + m.y;
}
S2_DEV float pkCross(pk2 perpR, pk2 P) // cross(r, P) = r.x P.y - r.y P.x = perp(r).y P.y + perp(r).x P.x
{
	const pk2 m = perpR * P;
	return m.y + m.x;
}
// perp(rA), perp(rB) of a prepared constraint: what chainSoftPacked multiplies with (poses only: made with the prep, in every lane at once)
struct SoftPerp
{
	pk2 pA[2], pB[2];
};
S2_DEV SoftPerp perpOf(const SoftPre& pre)
{
	SoftPerp q;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		q.pA[j] = pk2{-pre.rA[j].y, pre.rA[j].x}, q.pB[j] = pk2{-pre.rB[j].y, pre.rB[j].x};
	}
	return q;
}
template <int KIND, class BA, int POINTS = 0> S2_DEV void chainSoftPacked(SoftRegs<KIND>& r, const SoftPre& pre, const SoftPerp& q, const BA& b)
{
	static_assert(KIND != SOFT_JACOBI, "the Jacobi pass writes per-constraint deltas: chainSoft");
	const CHeader& h = r.h;
	float4* par = r.par;
	float2* imp = r.imp;
	const float4 velA = b.getVel(h.ia), velB = b.getVel(h.ib);
	pk2 vA = pk2{velA.x, velA.y}, vB = pk2{velB.x, velB.y};
	float wA = velA.z, wB = velB.z;
	const pk2 n = pk2{h.normal.x, h.normal.y};
	const pk2 t = pk2{n.y, -n.x}; // rightPerp
	const pk2 mA2 = pk2{h.mA, h.mA}, mB2 = pk2{h.mB, h.mB};
	const float iA = h.iA, iB = h.iB;
	float nImp[2], tImp[2];
	const pk2* pA = q.pA;
	const pk2* pB = q.pB;
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < h.pointCount)
		{
			const pk2 vrB = vB + pk2{wB, wB} * pB[j];
			const pk2 vrA = vA + pk2{wA, wA} * pA[j];
			const float vn = pkDot(vrB - vrA, n);
			const float old = imp[j].x;
			float impulse = -par[j].y * pre.massScale[j] * (vn + pre.bias[j]) - pre.impulseScale[j] * old;
			const float newImpulse = S2_MAXF(old + impulse, 0.0f);
			impulse = newImpulse - old;
			nImp[j] = newImpulse;
			tImp[j] = imp[j].y;
			const pk2 P = pk2{impulse, impulse} * n;
			vA = vA - mA2 * P;
			wA -= iA * pkCross(pA[j], P);
			vB = vB + mB2 * P;
			wB += iB * pkCross(pB[j], P);
		}
	}
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < h.pointCount)
		{
			const float tangentMass = par[j].z;
			const pk2 vrB = vB + pk2{wB, wB} * pB[j];
			const pk2 vrA = vA + pk2{wA, wA} * pA[j];
			const float vt = pkDot(vrB - vrA, t);
			float impulse = -tangentMass * vt;
			const float maxFriction = h.friction * nImp[j];
			const float newImpulse = S2_CLAMPF(tImp[j] + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - tImp[j];
			const pk2 P = pk2{impulse, impulse} * t;
			vA = vA - mA2 * P;
			wA -= iA * pkCross(pA[j], P);
			vB = vB + mB2 * P;
			wB += iB * pkCross(pB[j], P);
			imp[j] = make_float2(nImp[j], newImpulse);
		}
	}
	if (h.writeA)
	{
		b.setVel(h.ia, make_float4(vA.x, vA.y, wA, 0.0f));
	}
	if (h.writeB)
	{
		b.setVel(h.ib, make_float4(vB.x, vB.y, wB, 0.0f));
	}
}

// the arithmetic of one constraint: bodies read and written through `b`, impulses updated in `r`
// POINTS == 2: the caller has checked that the constraint has two points (a wave-uniform fast path without
// per-point exec masking); POINTS == 0: per-point guards on h.pointCount.
template <int KIND, class BA, bool PIN = true, int POINTS = 0>
S2_DEV void solveSoftRegs(SoftRegs<KIND>& r, const ContactView& c, const BA& b, float inv_h, int useBias, int k)
{
	completeSoft(r, b);
	const CHeader& h = r.h;
	const float biasCap = (KIND == SOFT_TGS || KIND == SOFT_JACOBI) ? -S2_MAX_BAUMGARTE_VELOCITY : -0.5f * S2_MAX_BAUMGARTE_VELOCITY;
	float4* an = r.an;
	float4* r0 = r.r0;
	float4* par = r.par;
	float4* sf = r.sf;
	float2* imp = r.imp;

	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 dcA, dcB;
	Rot qA, qB;
	if (KIND == SOFT_TGS || KIND == SOFT_FIXED)
	{
		BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
		dcA = pA.dc, qA = pA.q, dcB = pB.dc, qB = pB.q;
	}
	if (PIN)
	{
		pinSoft(r);
	}
	V2 normal = h.normal;
	V2 tangent = rightPerp(normal);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;

	V2 rAj[2], rBj[2];
	float nImp[2], tImp[2];

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < h.pointCount)
		{
			V2 rA, rB;
			float s;
			if (KIND == SOFT_TGS)
			{
				rA = rotate(qA, v2(an[j].x, an[j].y));
				rB = rotate(qB, v2(an[j].z, an[j].w));
				V2 ds = add(sub(dcB, dcA), sub(rB, rA));
				s = dot(ds, normal) + par[j].x;
			}
			else if (KIND == SOFT_FIXED)
			{
				V2 ds = add(sub(dcB, dcA), sub(rotate(qB, v2(an[j].z, an[j].w)), rotate(qA, v2(an[j].x, an[j].y))));
				s = dot(ds, normal) + par[j].x;
				rA = v2(r0[j].x, r0[j].y);
				rB = v2(r0[j].z, r0[j].w);
			}
			else
			{
				s = par[j].w;
				rA = v2(r0[j].x, r0[j].y);
				rB = v2(r0[j].z, r0[j].w);
			}
			rAj[j] = rA, rBj[j] = rB;

			// select form of: if (s > 0) bias = s * inv_h; else if (useBias) {bias = max(biasCoefficient * s, cap); ...}
			const bool speculative = s > 0.0f;
			const bool soft = !speculative && useBias != 0;
			float softBias = S2_MAXF(sf[j].x * s, biasCap);
			float bias = speculative ? s * inv_h : (soft ? softBias : 0.0f);
			float massScale = soft ? sf[j].y : 1.0f;
			float impulseScale = soft ? sf[j].z : 0.0f;

			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vn = dot(sub(vrB, vrA), normal);

			float impulse = -par[j].y * massScale * (vn + bias) - impulseScale * imp[j].x;
			float newImpulse = S2_MAXF(imp[j].x + impulse, 0.0f);
			impulse = newImpulse - imp[j].x;
			nImp[j] = newImpulse;
			tImp[j] = imp[j].y;

			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (POINTS == 2 || j < h.pointCount)
		{
			float tangentMass = par[j].z;
			V2 rA = rAj[j], rB = rBj[j];
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vt = dot(sub(vrB, vrA), tangent);
			float impulse = -tangentMass * vt;
			float maxFriction = h.friction * nImp[j];
			float newImpulse = S2_CLAMPF(tImp[j] + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - tImp[j];
			tImp[j] = newImpulse;
			V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
			imp[j] = make_float2(nImp[j], tImp[j]);
		}
	}

	if (KIND == SOFT_JACOBI)
	{
		// solve_jacobi.c:126-130: the body sums these in constraint order (jacobiApplyKernel)
		V2 dA = sub(vA, A.v), dB = sub(vB, B.v);
		c.deltaA[k] = make_float4(dA.x, dA.y, wA - A.w, 0.0f);
		c.deltaB[k] = make_float4(dB.x, dB.y, wB - B.w, 0.0f);
	}
	else
	{
		if (h.writeA)
		{
			storeVel(b, h.ia, vA, wA);
		}
		if (h.writeB)
		{
			storeVel(b, h.ib, vB, wB);
		}
	}
}

template <int KIND> S2_DEV void storeSoft(const ContactView& c, const SoftRegs<KIND>& r, int k)
{
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < r.h.pointCount)
		{
			c.impulse[j][k] = r.imp[j];
		}
	}
}

template <int KIND, class BA>
S2_DEV void solveContactsSoftOne(const ContactView& c, const BA& b, float inv_h, int useBias, int k)
{
	// all loads first (one memory round trip for the constraint, one for the bodies)
	SoftRegs<KIND> r = loadSoftB<KIND>(c, b, k);
	solveSoftRegs<KIND>(r, c, b, inv_h, useBias, k);
	storeSoft<KIND>(c, r, k);
}

// ---------------------------------------------------------------------------------------------
// rigid velocity sweeps:
//   RIGID_BAUMGARTE s2SolveContacts_PGS_Baumgarte solve_pgs.c:17-122      (normal first, fixed anchors)
//   RIGID_PGS       s2SolveContacts_PGS           solve_pgs_ngs.c:16-124  (friction first, no speculative)
//   RIGID_TGS       s2SolveContacts_TGS           solve_tgs_ngs.c:91-201  (current anchors, speculative)
// ---------------------------------------------------------------------------------------------
template <int KIND, class BA>
S2_DEV void solveContactsRigidOne(const ContactView& c, const BA& b, float inv_h, int k)
{
	CHeader h = loadHeaderB(c, b, k);
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 normal = h.normal;
	V2 tangent = KIND == RIGID_PGS ? crossVS(normal, 1.0f) : rightPerp(normal);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float friction = h.friction;

	V2 rAj[2], rBj[2];
	float nImp[2], tImp[2], sep[2], nMass[2], tMass[2], adj[2];
	V2 dcA, dcB;
	Rot qA, qB;
	if (KIND == RIGID_TGS)
	{
		BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
		dcA = pA.dc, qA = pA.q, dcB = pB.dc, qB = pB.q;
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 par = c.param[j][k];
			float2 imp = c.impulse[j][k];
			adj[j] = par.x, nMass[j] = par.y, tMass[j] = par.z, sep[j] = par.w;
			nImp[j] = imp.x, tImp[j] = imp.y;
			if (KIND == RIGID_TGS)
			{
				float4 an = c.anchor[j][k];
				rAj[j] = rotate(qA, v2(an.x, an.y));
				rBj[j] = rotate(qB, v2(an.z, an.w));
			}
			else
			{
				float4 r0 = c.r0[j][k];
				rAj[j] = v2(r0.x, r0.y);
				rBj[j] = v2(r0.z, r0.w);
			}
		}
	}

	if (KIND == RIGID_PGS)
	{
		// friction first: solve_pgs_ngs.c:42-80
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (j < h.pointCount)
			{
				if (sep[j] > 0.0f)
				{
					tImp[j] = 0.0f;
					continue;
				}
				V2 rA = rAj[j], rB = rBj[j];
				V2 vrB = add(vB, crossSV(wB, rB));
				V2 vrA = add(vA, crossSV(wA, rA));
				float vt = dot(sub(vrB, vrA), tangent);
				float lambda = tMass[j] * (-vt);
				float maxFriction = friction * nImp[j];
				float newImpulse = S2_CLAMPF(tImp[j] + lambda, -maxFriction, maxFriction);
				lambda = newImpulse - tImp[j];
				tImp[j] = newImpulse;
				V2 P = mulSV(lambda, tangent);
				vA = mulSub(vA, mA, P);
				wA -= iA * cross(rA, P);
				vB = mulAdd(vB, mB, P);
				wB += iB * cross(rB, P);
			}
		}
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (j < h.pointCount)
			{
				if (sep[j] > 0.0f)
				{
					nImp[j] = 0.0f;
					continue;
				}
				V2 rA = rAj[j], rB = rBj[j];
				V2 vrB = add(vB, crossSV(wB, rB));
				V2 vrA = add(vA, crossSV(wA, rA));
				float vn = dot(sub(vrB, vrA), normal);
				float impulse = -nMass[j] * vn;
				float newImpulse = S2_MAXF(nImp[j] + impulse, 0.0f);
				impulse = newImpulse - nImp[j];
				nImp[j] = newImpulse;
				V2 P = mulSV(impulse, normal);
				vA = mulSub(vA, mA, P);
				wA -= iA * cross(rA, P);
				vB = mulAdd(vB, mB, P);
				wB += iB * cross(rB, P);
			}
		}
	}
	else
	{
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (j < h.pointCount)
			{
				V2 rA = rAj[j], rB = rBj[j];
				float bias;
				if (KIND == RIGID_BAUMGARTE)
				{
					if (sep[j] > 0.0f)
					{
						bias = sep[j] * inv_h;
					}
					else
					{
						bias = S2_MAXF(S2_BAUMGARTE * inv_h * S2_MINF(0.0f, sep[j] + S2_LINEAR_SLOP), -S2_MAX_BAUMGARTE_VELOCITY);
					}
				}
				else
				{
					V2 d = add(sub(dcB, dcA), sub(rB, rA));
					float separation = dot(d, normal) + adj[j];
					bias = separation > 0.0f ? separation * inv_h : 0.0f;
				}
				V2 vrB = add(vB, crossSV(wB, rB));
				V2 vrA = add(vA, crossSV(wA, rA));
				float vn = dot(sub(vrB, vrA), normal);
				float impulse = -nMass[j] * (vn + bias);
				float newImpulse = S2_MAXF(nImp[j] + impulse, 0.0f);
				impulse = newImpulse - nImp[j];
				nImp[j] = newImpulse;
				V2 P = mulSV(impulse, normal);
				vA = mulSub(vA, mA, P);
				wA -= iA * cross(rA, P);
				vB = mulAdd(vB, mB, P);
				wB += iB * cross(rB, P);
			}
		}
#pragma unroll
		for (int j = 0; j < 2; ++j)
		{
			if (j < h.pointCount)
			{
				V2 rA = rAj[j], rB = rBj[j];
				V2 vrB = add(vB, crossSV(wB, rB));
				V2 vrA = add(vA, crossSV(wA, rA));
				float vt = dot(sub(vrB, vrA), tangent);
				float lambda = KIND == RIGID_BAUMGARTE ? tMass[j] * (-vt) : -tMass[j] * vt;
				float maxFriction = friction * nImp[j];
				float newImpulse = S2_CLAMPF(tImp[j] + lambda, -maxFriction, maxFriction);
				lambda = newImpulse - tImp[j];
				tImp[j] = newImpulse;
				V2 P = mulSV(lambda, tangent);
				vA = mulSub(vA, mA, P);
				wA -= iA * cross(rA, P);
				vB = mulAdd(vB, mB, P);
				wB += iB * cross(rB, P);
			}
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
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

// s2SolveContacts_TGS_Sticky: solve_tgs_sticky.c:167-310
template <class BA>
S2_DEV void solveContactsStickyOne(const ContactView& c, const BA& b, s2amdContact* wire, float inv_h, int useBias, int k)
{
	CHeader h = loadHeaderB(c, b, k);
	const float contactBaumgarte = 0.8f;
	const float frictionBaumgarte = 0.5f;
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 dcA = pA.dc, dcB = pB.dc;
	Rot qA = pA.q, qB = pB.q;
	V2 normal = h.normal;
	V2 tangent = rightPerp(normal);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float totalNormalImpulse = 0.0f;
	float nImp[2], tImp[2];
	bool slipped = false;

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 an = c.anchor[j][k];
			float4 par = c.param[j][k];
			float2 imp = c.impulse[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + par.x;
			float bias = 0.0f;
			if (separation > 0.0f)
			{
				bias = separation * inv_h;
			}
			else if (useBias)
			{
				bias = S2_MAXF(-S2_MAX_BAUMGARTE_VELOCITY, contactBaumgarte * separation * inv_h);
			}
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 vrB = add(vB, crossSV(wB, rB));
			float vn = dot(sub(vrB, vrA), normal);
			float impulse = -par.y * (vn + bias);
			float newImpulse = S2_MAXF(imp.x + impulse, 0.0f);
			impulse = newImpulse - imp.x;
			nImp[j] = newImpulse;
			tImp[j] = imp.y;
			totalNormalImpulse += newImpulse;
			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 fa = c.fanchor[j][k];
			float tangentMass = c.param[j][k].z;
			float tangentSeparation = c.soft[j][k].w;
			V2 rAf = rotate(qA, v2(fa.x, fa.y));
			V2 rBf = rotate(qB, v2(fa.z, fa.w));
			V2 d = add(sub(dcB, dcA), sub(rBf, rAf));
			float separation = dot(d, tangent) + tangentSeparation;
			float bias = useBias ? frictionBaumgarte * separation * inv_h : 0.0f;
			V2 vrA = add(vA, crossSV(wA, rAf));
			V2 vrB = add(vB, crossSV(wB, rBf));
			float vt = dot(sub(vrB, vrA), tangent);
			float impulse = -tangentMass * (vt + bias);
			float maxFriction = 0.5f * h.friction * totalNormalImpulse;
			float newImpulse = tImp[j] + impulse;
			if (newImpulse < -maxFriction)
			{
				newImpulse = -maxFriction;
				slipped = true;
			}
			else if (newImpulse > maxFriction)
			{
				newImpulse = maxFriction;
				slipped = true;
			}
			impulse = newImpulse - tImp[j];
			tImp[j] = newImpulse;
			V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rAf, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rBf, P);
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
		}
	}
	if (slipped)
	{
		wire[c.contactIndex[k]].frictionPersisted = 0; // solve_tgs_sticky.c:284,289
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

// s2SolveContact_NGS: solve_common.c:328-394
template <class BA>
S2_DEV void solveContactsNGSOne(const ContactView& c, const BA& b, int k)
{
	CHeader h = loadHeaderB(c, b, k);
	BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
	V2 dcA = pA.dc, dcB = pB.dc;
	Rot qA = pA.q, qB = pB.q;
	V2 normal = h.normal;
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 par = c.param[j][k];
			if (par.w > 0.0f)
			{
				continue;
			}
			float4 an = c.anchor[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + par.x;
			float C = S2_CLAMPF(S2_BAUMGARTE * (separation + S2_LINEAR_SLOP), -S2_MAX_LINEAR_CORRECTION, 0.0f);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
			float impulse = K > 0.0f ? -C / K : 0.0f;
			V2 P = mulSV(impulse, normal);
			dcA = mulSub(dcA, mA, P);
			qA = integrateRot(qA, -iA * cross(rA, P));
			dcB = mulAdd(dcB, mB, P);
			qB = integrateRot(qB, iB * cross(rB, P));
		}
	}
	if (h.writeA)
	{
		storePose(b, h.ia, dcA, qA);
	}
	if (h.writeB)
	{
		storePose(b, h.ib, dcB, qB);
	}
}

// s2SolveContactPositions_XPBD: solve_xpbd.c:88-216
template <class BA>
S2_DEV void xpbdContactPositionsOne(const ContactView& c, const BA& b, float hh, int k)
{
	CHeader h = loadHeaderB(c, b, k);
	const float baseCompliance = 0.0f;
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float compliance = (mA == 0.0f || mB == 0.0f) ? 0.25f * baseCompliance : baseCompliance;
	BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
	V2 dcA = pA.dc, dcB = pB.dc;
	Rot qA = pA.q, qB = pB.q;
	V2 normal = h.normal;
	V2 tangent = crossVS(normal, 1.0f);
	float nImp[2] = {0.0f, 0.0f}, tImp[2] = {0.0f, 0.0f};

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 an = c.anchor[j][k];
			float4 r0 = c.r0[j][k];
			float2 imp = c.impulse[j][k];
			nImp[j] = imp.x, tImp[j] = imp.y;
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 drA = sub(rA, v2(r0.x, r0.y));
			V2 drB = sub(rB, v2(r0.z, r0.w));
			V2 ds = add(sub(dcB, dcA), sub(drB, drA));
			float C = dot(ds, normal) + c.param[j][k].w;
			if (C > 0)
			{
				nImp[j] = 0.0f;
				continue;
			}
			C = S2_MAXF(-S2_MAX_BAUMGARTE_VELOCITY * hh, C);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float kA = mA + iA * rnA * rnA;
			float kB = mB + iB * rnB * rnB;
			float lambda = -C / (kA + kB + compliance);
			nImp[j] = lambda;
			V2 P = mulSV(lambda, normal);
			dcA = mulSub(dcA, mA, P);
			qA = integrateRot(qA, -iA * cross(rA, P));
			dcB = mulAdd(dcB, mB, P);
			qB = integrateRot(qB, iB * cross(rB, P));
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 an = c.anchor[j][k];
			float4 r0 = c.r0[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 drA = sub(rA, v2(r0.x, r0.y));
			V2 drB = sub(rB, v2(r0.z, r0.w));
			V2 dp = add(sub(dcB, dcA), sub(drB, drA));
			float C = dot(dp, tangent);
			float rtA = cross(rA, tangent);
			float rtB = cross(rB, tangent);
			float kA = mA + iA * rtA * rtA;
			float kB = mB + iB * rtB * rtB;
			float lambda = -C / (kA + kB);
			float maxLambda = h.friction * nImp[j];
			if (lambda < -maxLambda || maxLambda < lambda)
			{
				tImp[j] = 0.0f;
			}
			else
			{
				tImp[j] = lambda;
				V2 P = mulSV(lambda, tangent);
				dcA = mulSub(dcA, mA, P);
				qA = integrateRot(qA, -iA * cross(rA, P));
				dcB = mulAdd(dcB, mB, P);
				qB = integrateRot(qB, iB * cross(rB, P));
			}
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
		}
	}
	if (h.writeA)
	{
		storePose(b, h.ia, dcA, qA);
	}
	if (h.writeB)
	{
		storePose(b, h.ib, dcB, qB);
	}
}

// s2SolveContactVelocities_XPBD: solve_xpbd.c:218-338
template <class BA>
S2_DEV void xpbdContactVelocitiesOne(const ContactView& c, const BA& b, float hh, int k)
{
	CHeader h = loadHeaderB(c, b, k);
	float inv_h = hh > 0.0f ? 1.0f / hh : 0.0f;
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	Rot qA = loadPose(b, h.ia).q, qB = loadPose(b, h.ib).q;
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 normal = h.normal;
	V2 tangent = crossVS(normal, 1.0f);
	float nImp[2] = {0.0f, 0.0f};

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float2 imp = c.impulse[j][k];
			nImp[j] = imp.x;
			if (imp.x == 0.0f)
			{
				continue;
			}
			float4 an = c.anchor[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 dv = sub(vrB, vrA);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float kA = mA + iA * rnA * rnA;
			float kB = mB + iB * rnB * rnB;
			float vn = dot(dv, normal);
			float lambda = -vn / (kA + kB);
			V2 P = mulSV(lambda, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < h.pointCount)
		{
			float4 an = c.anchor[j][k];
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 dv = sub(vrB, vrA);
			float vt = dot(dv, tangent);
			if (vt == 0.0f)
			{
				continue;
			}
			float rtA = cross(rA, tangent);
			float rtB = cross(rB, tangent);
			float kA = mA + iA * rtA * rtA;
			float kB = mB + iB * rtB * rtB;
			float maxFrictionImpulse = h.friction * nImp[j];
			float huf = (maxFrictionImpulse * inv_h) * (kA + kB);
			float abs_vt = S2_ABSF(vt);
			float Cdot = (vt / abs_vt) * S2_MINF(huf, abs_vt);
			float lambda = -Cdot / (kA + kB);
			c.impulse[j][k] = make_float2(nImp[j], lambda);
			V2 P = mulSV(lambda, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
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

// ---------------------------------------------------------------------------------------------
// PGS_NGS_Block: s2BlockSolveVelocity (solve_pgs_ngs_block.c:329-658) and
// s2BlockSolvePosition (:679-890)
// ---------------------------------------------------------------------------------------------
#define BLOCK_APPLY_VELOCITY(d)                                                                                                  \
	{                                                                                                                            \
		V2 P1 = mulSV((d).x, normal);                                                                                            \
		V2 P2 = mulSV((d).y, normal);                                                                                            \
		vA = mulSub(vA, mA, add(P1, P2));                                                                                        \
		wA -= iA * (cross(rA1, P1) + cross(rA2, P2));                                                                            \
		vB = mulAdd(vB, mB, add(P1, P2));                                                                                        \
		wB += iB * (cross(rB1, P1) + cross(rB2, P2));                                                                            \
	}

template <class BA>
S2_DEV void blockSolveVelocityOne(const ContactView& c, const BA& b, int k)
{
	CHeader h = loadHeaderB(c, b, k);
	float4 K4 = c.blockK[k];
	int pointCount = (int)asBits(K4.w);
	BodyVel A = loadVel(b, h.ia), B = loadVel(b, h.ib);
	V2 vA = A.v, vB = B.v;
	float wA = A.w, wB = B.w;
	V2 normal = h.normal;
	V2 tangent = crossVS(normal, 1.0f);
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	float friction = h.friction;

	V2 rAj[2], rBj[2];
	float nImp[2], tImp[2], nMass[2], tMass[2], vBias[2];
#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			float4 r0 = c.r0[j][k];
			float4 par = c.param[j][k];
			float2 imp = c.impulse[j][k];
			rAj[j] = v2(r0.x, r0.y), rBj[j] = v2(r0.z, r0.w);
			nMass[j] = par.y, tMass[j] = par.z;
			nImp[j] = imp.x, tImp[j] = imp.y;
			vBias[j] = c.soft[j][k].x;
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			V2 vrB = add(vB, crossSV(wB, rBj[j]));
			V2 vrA = add(vA, crossSV(wA, rAj[j]));
			V2 dv = sub(vrB, vrA);
			float vt = dot(dv, tangent);
			float lambda = tMass[j] * (-vt);
			float maxFriction = friction * nImp[j];
			float newImpulse = S2_CLAMPF(tImp[j] + lambda, -maxFriction, maxFriction);
			lambda = newImpulse - tImp[j];
			tImp[j] = newImpulse;
			V2 P = mulSV(lambda, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rAj[j], P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rBj[j], P);
		}
	}

	if (pointCount == 1)
	{
		V2 vrB = add(vB, crossSV(wB, rBj[0]));
		V2 vrA = add(vA, crossSV(wA, rAj[0]));
		V2 dv = sub(vrB, vrA);
		float vn = dot(dv, normal);
		float lambda = -nMass[0] * (vn - vBias[0]);
		float newImpulse = S2_MAXF(nImp[0] + lambda, 0.0f);
		lambda = newImpulse - nImp[0];
		nImp[0] = newImpulse;
		V2 P = mulSV(lambda, normal);
		vA = mulSub(vA, mA, P);
		wA -= iA * cross(rAj[0], P);
		vB = mulAdd(vB, mB, P);
		wB += iB * cross(rBj[0], P);
	}
	else if (pointCount == 2)
	{
		V2 rA1 = rAj[0], rB1 = rBj[0], rA2 = rAj[1], rB2 = rBj[1];
		M22 K, NM;
		K.cx = v2(K4.x, K4.y);
		K.cy = v2(K4.y, K4.z);
		float4 nm = c.blockNM[k];
		NM.cx = v2(nm.x, nm.y);
		NM.cy = v2(nm.z, nm.w);
		V2 a = v2(nImp[0], nImp[1]);
		V2 vrA, vrB;
		vrA = add(vA, crossSV(wA, rA1));
		vrB = add(vB, crossSV(wB, rB1));
		V2 dv1 = sub(vrB, vrA);
		vrA = add(vA, crossSV(wA, rA2));
		vrB = add(vB, crossSV(wB, rB2));
		V2 dv2 = sub(vrB, vrA);
		float vn1 = dot(dv1, normal);
		float vn2 = dot(dv2, normal);
		V2 bb = v2(vn1 - vBias[0], vn2 - vBias[1]);
		bb = sub(bb, mulMV(K, a));

		for (;;)
		{
			V2 x = neg(mulMV(NM, bb));
			if (x.x >= 0.0f && x.y >= 0.0f)
			{
				V2 d = sub(x, a);
				BLOCK_APPLY_VELOCITY(d);
				nImp[0] = x.x, nImp[1] = x.y;
				break;
			}
			x.x = -nMass[0] * bb.x;
			x.y = 0.0f;
			vn1 = 0.0f;
			vn2 = K.cx.y * x.x + bb.y;
			if (x.x >= 0.0f && vn2 >= 0.0f)
			{
				V2 d = sub(x, a);
				BLOCK_APPLY_VELOCITY(d);
				nImp[0] = x.x, nImp[1] = x.y;
				break;
			}
			x.x = 0.0f;
			x.y = -nMass[1] * bb.y;
			vn1 = K.cy.x * x.y + bb.x;
			vn2 = 0.0f;
			if (x.y >= 0.0f && vn1 >= 0.0f)
			{
				V2 d = sub(x, a);
				BLOCK_APPLY_VELOCITY(d);
				nImp[0] = x.x, nImp[1] = x.y;
				break;
			}
			x.x = 0.0f;
			x.y = 0.0f;
			vn1 = bb.x;
			vn2 = bb.y;
			if (vn1 >= 0.0f && vn2 >= 0.0f)
			{
				V2 d = sub(x, a);
				BLOCK_APPLY_VELOCITY(d);
				nImp[0] = x.x, nImp[1] = x.y;
				break;
			}
			break;
		}
	}

#pragma unroll
	for (int j = 0; j < 2; ++j)
	{
		if (j < pointCount)
		{
			c.impulse[j][k] = make_float2(nImp[j], tImp[j]);
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

#define BLOCK_APPLY_POSITION(d)                                                                                                  \
	{                                                                                                                            \
		V2 P1 = mulSV((d).x, normal);                                                                                            \
		V2 P2 = mulSV((d).y, normal);                                                                                            \
		dcA = mulSub(dcA, mA, add(P1, P2));                                                                                      \
		qA = integrateRot(qA, -iA * (cross(rA1, P1) + cross(rA2, P2)));                                                          \
		dcB = mulAdd(dcB, mB, add(P1, P2));                                                                                      \
		qB = integrateRot(qB, iB * (cross(rB1, P1) + cross(rB2, P2)));                                                           \
	}

template <class BA>
S2_DEV void blockSolvePositionOne(const ContactView& c, const BA& b, int k)
{
	CHeader h = loadHeaderB(c, b, k);
	int pointCount = (int)asBits(c.blockK[k].w);
	const float slop = S2_LINEAR_SLOP;
	float mA = h.mA, iA = h.iA, mB = h.mB, iB = h.iB;
	BodyPose pA = loadPose(b, h.ia), pB = loadPose(b, h.ib);
	V2 dcA = pA.dc, dcB = pB.dc;
	Rot qA = pA.q, qB = pB.q;
	V2 normal = h.normal;
	bool degenerate = pointCount != 2;

	if (pointCount == 2)
	{
		float4 an1 = c.anchor[0][k], an2 = c.anchor[1][k];
		float adj1 = c.param[0][k].x, adj2 = c.param[1][k].x;
		V2 rA1 = rotate(qA, v2(an1.x, an1.y));
		V2 rB1 = rotate(qB, v2(an1.z, an1.w));
		V2 rA2 = rotate(qA, v2(an2.x, an2.y));
		V2 rB2 = rotate(qB, v2(an2.z, an2.w));
		V2 dc = sub(dcB, dcA);
		V2 d1 = add(dc, sub(rB1, rA1));
		float separation1 = dot(d1, normal) + adj1;
		V2 d2 = add(dc, sub(rB2, rA2));
		float separation2 = dot(d2, normal) + adj2;
		float C1 = S2_CLAMPF(S2_BAUMGARTE * (separation1 + slop), -S2_MAX_LINEAR_CORRECTION, 0.0f);
		float C2 = S2_CLAMPF(S2_BAUMGARTE * (separation2 + slop), -S2_MAX_LINEAR_CORRECTION, 0.0f);
		V2 bb = v2(C1, C2);
		float rn1A = cross(rA1, normal);
		float rn1B = cross(rB1, normal);
		float rn2A = cross(rA2, normal);
		float rn2B = cross(rB2, normal);
		float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
		float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
		float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
		const float k_maxConditionNumber = 10000.0f;
		if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12))
		{
			M22 K;
			K.cx = v2(k11, k12);
			K.cy = v2(k12, k22);
			M22 invK = inverse22(K);
			for (;;)
			{
				V2 x = neg(mulMV(invK, bb));
				if (x.x >= 0.0f && x.y >= 0.0f)
				{
					BLOCK_APPLY_POSITION(x);
					break;
				}
				x.x = -bb.x / k11;
				x.y = 0.0f;
				float vn2 = K.cx.y * x.x + bb.y;
				if (x.x >= 0.0f && vn2 >= 0.0f)
				{
					BLOCK_APPLY_POSITION(x);
					break;
				}
				x.x = 0.0f;
				x.y = -bb.y / k22;
				float vn1 = K.cy.x * x.y + bb.x;
				if (x.y >= 0.0f && vn1 >= 0.0f)
				{
					BLOCK_APPLY_POSITION(x);
					break;
				}
				break;
			}
		}
		else
		{
			degenerate = true;
		}
	}

	if (degenerate)
	{
		for (int j = 0; j < pointCount; ++j)
		{
			float4 an = c.anchor[j][k];
			float adj = c.param[j][k].x;
			V2 rA = rotate(qA, v2(an.x, an.y));
			V2 rB = rotate(qB, v2(an.z, an.w));
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + adj;
			float C = S2_CLAMPF(S2_BAUMGARTE * (separation + slop), -S2_MAX_LINEAR_CORRECTION, 0.0f);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
			float impulse = K > 0.0f ? -C / K : 0.0f;
			V2 P = mulSV(impulse, normal);
			dcA = mulSub(dcA, mA, P);
			qA = integrateRot(qA, -iA * cross(rA, P));
			dcB = mulAdd(dcB, mB, P);
			qB = integrateRot(qB, iB * cross(rB, P));
		}
	}
	if (h.writeA)
	{
		storePose(b, h.ia, dcA, qA);
	}
	if (h.writeB)
	{
		storePose(b, h.ib, dcB, qB);
	}
}


// ---------------------------------------------------------------------------------------------
// joints: src/revolute_joint.c, src/mouse_joint.c, dispatch src/joint.c:294-465
// ---------------------------------------------------------------------------------------------
struct JState
{
	int ia, ib;
	uint32_t flags;
	V2 lA, lB;
	float mA, iA, mB, iB;
	M22 pivotMass;
	float biasCoefficient, massCoefficient, impulseCoefficient, axialMass;
	V2 centerDiff0;
	V2 impulse;
	float motorImpulse, lowerImpulse, upperImpulse, bodyI;
	float referenceAngle, lowerAngle, upperAngle, maxMotorTorque, motorSpeed;
};

S2_DEV Rot loadRotOnly(const BodyView& b, int i)
{
	float4 d = b.dq[i];
	Rot q;
	q.s = d.z, q.c = d.w;
	return q;
}

// A joint view whose arrays live in LDS (generic_kernel.hip stages a strip's joint records there for the whole step): the same
// member names as JointView, each array a typed LDS column, so that the functions below -- templated on the view -- read them with
// ds_read instead of global (or, through a generic pointer, flat) loads.  Index 0 is the first staged joint.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));
S2_DEV float4 toHip(f4v v) { return make_float4(v.x, v.y, v.z, v.w); }
S2_DEV float2 toHip(f2v v) { return make_float2(v.x, v.y); }
S2_DEV int2 toHip(i2v v) { return make_int2(v.x, v.y); }
S2_DEV f4v toLds(float4 v) { return f4v{v.x, v.y, v.z, v.w}; }
S2_DEV f2v toLds(float2 v) { return f2v{v.x, v.y}; }
S2_DEV i2v toLds(int2 v) { return i2v{v.x, v.y}; }
template <class T, class V> struct LdsColumn
{
	typedef __attribute__((address_space(3))) V Cell;
	Cell* p;
	struct Ref
	{
		Cell* q;
		S2_DEV operator T() const { return toHip(*q); }
		S2_DEV void operator=(T v) const { *q = toLds(v); }
	};
	S2_DEV Ref operator[](int k) const { return Ref{p + k}; }
};
struct LdsJointView
{
	int2* bodies; // (not staged: pool slots are only read by the global-index accessors)
	LdsColumn<int2, i2v> localBodies;
	LdsColumn<float4, f4v> frame, mass, pivot, soft, axial, limits, misc;
	LdsColumn<float2, f2v> centerDiff0, impulse;
};

template <int MODE, class JV> S2_DEV JState loadJoint(const JV& j, int k)
{
	JState s;
	int2 bd;
	if (MODE == S2_IDX_LOCAL)
	{
		bd = j.localBodies[k];
	}
	else
	{
		bd = j.bodies[k];
	}
	float4 fr = j.frame[k], ms = j.mass[k], pv = j.pivot[k], sf = j.soft[k], ax = j.axial[k], lm = j.limits[k], mc = j.misc[k];
	float2 cd = j.centerDiff0[k], im = j.impulse[k];
	s.ia = bd.x, s.ib = bd.y;
	s.flags = asBits(mc.y);
	s.lA = v2(fr.x, fr.y), s.lB = v2(fr.z, fr.w);
	s.mA = ms.x, s.iA = ms.y, s.mB = ms.z, s.iB = ms.w;
	s.pivotMass.cx = v2(pv.x, pv.y), s.pivotMass.cy = v2(pv.z, pv.w);
	s.biasCoefficient = sf.x, s.massCoefficient = sf.y, s.impulseCoefficient = sf.z, s.axialMass = sf.w;
	s.centerDiff0 = v2(cd.x, cd.y);
	s.impulse = v2(im.x, im.y);
	s.motorImpulse = ax.x, s.lowerImpulse = ax.y, s.upperImpulse = ax.z, s.bodyI = ax.w;
	s.referenceAngle = lm.x, s.lowerAngle = lm.y, s.upperAngle = lm.z, s.maxMotorTorque = lm.w;
	s.motorSpeed = mc.x;
	return s;
}

template <class JV> S2_DEV void storeJointImpulses(const JV& j, int k, const JState& s)
{
	j.impulse[k] = make_float2(s.impulse.x, s.impulse.y);
	j.axial[k] = make_float4(s.motorImpulse, s.lowerImpulse, s.upperImpulse, s.bodyI);
}

S2_DEV M22 revoluteK(float mA, float mB, float iA, float iB, V2 rA, V2 rB)
{
	// revolute_joint.c:70-74, :461-465, :631-636, :768-773
	M22 K;
	K.cx.x = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
	K.cy.x = -rA.y * rA.x * iA - rB.y * rB.x * iB;
	K.cx.y = K.cy.x;
	K.cy.y = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
	return K;
}

S2_DEV void softCoefficients(float h, float zeta, float omega, float& bias, float& mass, float& impulse)
{
	bias = omega / (2.0f * zeta + h * omega);
	float c = h * omega * (2.0f * zeta + h * omega);
	impulse = 1.0f / (1.0f + c);
	mass = c * impulse;
}

// s2SolveMouse: mouse_joint.c:109-167
template <class BA> S2_DEV void solveMouse(JState& s, const BA& b, float ctxH)
{
	float4 vb = b.getVel(s.ib);
	float4 db = b.getDq(s.ib);
	V2 vB = v2(vb.x, vb.y);
	float wB = vb.z;
	float mB = s.mB, iB = s.iB;
	{
		float h = ctxH;
		float zeta = 0.1f;
		float omega = 2.0f * S2_PI * 0.5f;
		float c = h * omega * (2.0f * zeta + h * omega);
		float impulseScale = 1.0f / (1.0f + c);
		float massScale = c * impulseScale;
		float impulse = -massScale * s.bodyI * wB - impulseScale * s.motorImpulse;
		s.motorImpulse += impulse;
		wB += iB * impulse;
	}
	{
		Rot qB;
		qB.s = db.z, qB.c = db.w;
		V2 rB = rotate(qB, s.lB);
		V2 Cdot = add(vB, crossSV(wB, rB));
		V2 dcB = v2(db.x, db.y);
		V2 separation = add(add(dcB, rB), s.centerDiff0);
		V2 bias = mulSV(s.biasCoefficient, separation);
		float massScale = s.massCoefficient;
		float impulseScale = s.impulseCoefficient;
		V2 bb = mulMV(s.pivotMass, add(Cdot, bias));
		V2 impulse;
		impulse.x = -massScale * bb.x - impulseScale * s.impulse.x;
		impulse.y = -massScale * bb.y - impulseScale * s.impulse.y;
		s.impulse.x += impulse.x;
		s.impulse.y += impulse.y;
		vB = mulAdd(vB, mB, impulse);
		wB += iB * cross(rB, impulse);
	}
	if (s.flags & S2J_WRITE_B)
	{
		b.setVel(s.ib, make_float4(vB.x, vB.y, wB, 0.0f));
	}
}

// motor row: revolute_joint.c:175-187, :526-538, :678-690
S2_DEV void revoluteMotor(JState& s, float h, float& wA, float& wB)
{
	float Cdot = wB - wA - s.motorSpeed;
	float impulse = -s.axialMass * Cdot;
	float oldImpulse = s.motorImpulse;
	float maxImpulse = h * s.maxMotorTorque;
	s.motorImpulse = S2_CLAMPF(s.motorImpulse + impulse, -maxImpulse, maxImpulse);
	impulse = s.motorImpulse - oldImpulse;
	wA -= s.iA * impulse;
	wB += s.iB * impulse;
}

template <int KIND, class BA, class JV>
S2_DEV void solveJointsOne(const JV& jv, const BA& b, const StepConsts& sc, float h, float inv_h, int useBias, int k)
{
	JState s = loadJoint<BA::kMode>(jv, k);

	if (s.flags & S2J_MOUSE)
	{
		if (KIND == JSOLVE_WARM)
		{
			// s2WarmStartMouse: mouse_joint.c:85-107
			float4 vb = b.getVel(s.ib);
			float4 db = b.getDq(s.ib);
			Rot qB;
			qB.s = db.z, qB.c = db.w;
			V2 rB = rotate(qB, s.lB);
			V2 vB = v2(vb.x, vb.y);
			float wB = vb.z;
			vB = mulAdd(vB, s.mB, s.impulse);
			wB += s.iB * (cross(rB, s.impulse) + s.motorImpulse);
			if (s.flags & S2J_WRITE_B)
			{
				b.setVel(s.ib, make_float4(vB.x, vB.y, wB, 0.0f));
			}
		}
		else if (KIND == JSOLVE_PLAIN || KIND == JSOLVE_BAUMGARTE || KIND == JSOLVE_XPBD || (KIND == JSOLVE_SOFT && useBias))
		{
			// joint.c:342, :398-401, :418, :456
			solveMouse(s, b, sc.h);
			storeJointImpulses(jv, k, s);
		}
		return;
	}

	const bool writeA = (s.flags & S2J_WRITE_A) != 0, writeB = (s.flags & S2J_WRITE_B) != 0;
	const bool enableMotor = (s.flags & S2J_ENABLE_MOTOR) != 0, enableLimit = (s.flags & S2J_ENABLE_LIMIT) != 0;
	float mA = s.mA, iA = s.iA, mB = s.mB, iB = s.iB;

	if (KIND == JSOLVE_POSITION)
	{
		// s2SolveRevolutePosition: revolute_joint.c:305-419
		float4 da = b.getDq(s.ia), db = b.getDq(s.ib);
		V2 dcA = v2(da.x, da.y), dcB = v2(db.x, db.y);
		Rot qA, qB;
		qA.s = da.z, qA.c = da.w, qB.s = db.z, qB.c = db.w;
		bool fixedRotation = (iA + iB == 0.0f);
		if (enableLimit && fixedRotation == false)
		{
			float angle = relativeAngle(qB, qA) - s.referenceAngle;
			float C = 0.0f;
			if (S2_ABSF(s.upperAngle - s.lowerAngle) < 2.0f * S2_ANGULAR_SLOP)
			{
				C = S2_CLAMPF(angle - s.lowerAngle, -S2_MAX_ANGULAR_CORRECTION, S2_MAX_ANGULAR_CORRECTION);
			}
			else if (angle <= s.lowerAngle)
			{
				C = S2_CLAMPF(angle - s.lowerAngle + S2_ANGULAR_SLOP, -S2_MAX_ANGULAR_CORRECTION, 0.0f);
			}
			else if (angle >= s.upperAngle)
			{
				C = S2_CLAMPF(angle - s.upperAngle - S2_ANGULAR_SLOP, 0.0f, S2_MAX_ANGULAR_CORRECTION);
			}
			float limitImpulse = -s.axialMass * C;
			qA = integrateRot(qA, -iA * limitImpulse);
			qB = integrateRot(qB, iB * limitImpulse);
		}
		{
			V2 rA = rotate(qA, s.lA);
			V2 rB = rotate(qB, s.lB);
			V2 C = add(add(sub(dcB, dcA), sub(rB, rA)), s.centerDiff0);
			// fresh K with the operand order of revolute_joint.c:388-393
			M22 K;
			K.cx.x = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
			K.cx.y = -iA * rA.x * rA.y - iB * rB.x * rB.y;
			K.cy.x = K.cx.y;
			K.cy.y = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
			V2 impulse = solve22(K, neg(C));
			dcA = mulSub(dcA, mA, impulse);
			qA = integrateRot(qA, -iA * cross(rA, impulse));
			dcB = mulAdd(dcB, mB, impulse);
			qB = integrateRot(qB, iB * cross(rB, impulse));
		}
		if (writeA)
		{
			b.setDq(s.ia, make_float4(dcA.x, dcA.y, qA.s, qA.c));
		}
		if (writeB)
		{
			b.setDq(s.ib, make_float4(dcB.x, dcB.y, qB.s, qB.c));
		}
		return;
	}

	if (KIND == JSOLVE_XPBD)
	{
		// s2SolveRevolute_XPBD: revolute_joint.c:825-888
		const float compliance = 0.0f;
		float4 da = b.getDq(s.ia), db = b.getDq(s.ib);
		V2 dcA = v2(da.x, da.y), dcB = v2(db.x, db.y);
		Rot qA, qB;
		qA.s = da.z, qA.c = da.w, qB.s = db.z, qB.c = db.w;
		V2 rA = rotate(qA, s.lA);
		V2 rB = rotate(qB, s.lB);
		V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), s.centerDiff0);
		float c = length(separation);
		V2 n = normalize(separation);
		if (mA == 0.0f && mB == 0.0f)
		{
			return;
		}
		float rnA = cross(rA, n);
		float rnB = cross(rB, n);
		float kA = mA + iA * rnA * rnA;
		float kB = mB + iB * rnB * rnB;
		float lambda = -c / (kA + kB + compliance);
		V2 p = mulSV(lambda, n);
		dcA = mulSub(dcA, mA, p);
		qA = integrateRot(qA, -iA * cross(rA, p));
		dcB = mulAdd(dcB, mB, p);
		qB = integrateRot(qB, iB * cross(rB, p));
		if (writeA)
		{
			b.setDq(s.ia, make_float4(dcA.x, dcA.y, qA.s, qA.c));
		}
		if (writeB)
		{
			b.setDq(s.ib, make_float4(dcB.x, dcB.y, qB.s, qB.c));
		}
		return;
	}

	float4 va = b.getVel(s.ia), vb = b.getVel(s.ib);
	float4 da = b.getDq(s.ia), db = b.getDq(s.ib);
	V2 vA = v2(va.x, va.y), vB = v2(vb.x, vb.y);
	float wA = va.z, wB = vb.z;
	Rot qA, qB;
	qA.s = da.z, qA.c = da.w, qB.s = db.z, qB.c = db.w;

	if (KIND == JSOLVE_WARM)
	{
		// s2WarmStartRevolute: revolute_joint.c:107-150
		V2 rA = rotate(qA, s.lA);
		V2 rB = rotate(qB, s.lB);
		float axialImpulse = s.motorImpulse + s.lowerImpulse - s.upperImpulse;
		V2 P = s.impulse;
		vA = mulSub(vA, mA, P);
		wA -= iA * (cross(rA, P) + axialImpulse);
		vB = mulAdd(vB, mB, P);
		wB += iB * (cross(rB, P) + axialImpulse);
	}
	else
	{
		// s2SolveRevolute :152-303, s2SolveRevolute_Soft :508-657, s2SolveRevolute_Baumgarte :660-790
		bool fixedRotation = (iA + iB == 0.0f);
		if (enableMotor && fixedRotation == false)
		{
			revoluteMotor(s, h, wA, wB);
		}
		if (enableLimit && fixedRotation == false)
		{
			float jointAngle = relativeAngle(qB, qA) - s.referenceAngle;
			if (KIND == JSOLVE_PLAIN)
			{
				{
					float C = jointAngle - s.lowerAngle;
					float Cdot = wB - wA;
					float impulse = -s.axialMass * (Cdot + S2_MAXF(C, 0.0f) / h);
					float oldImpulse = s.lowerImpulse;
					s.lowerImpulse = S2_MAXF(s.lowerImpulse + impulse, 0.0f);
					impulse = s.lowerImpulse - oldImpulse;
					wA -= iA * impulse;
					wB += iB * impulse;
				}
				{
					float C = s.upperAngle - jointAngle;
					float Cdot = wA - wB;
					float impulse = -s.axialMass * (Cdot + S2_MAXF(C, 0.0f) / h);
					float oldImpulse = s.upperImpulse;
					s.upperImpulse = S2_MAXF(s.upperImpulse + impulse, 0.0f);
					impulse = s.upperImpulse - oldImpulse;
					wA += iA * impulse;
					wB -= iB * impulse;
				}
			}
			else
			{
				const bool soft = KIND == JSOLVE_SOFT;
				{
					float C = jointAngle - s.lowerAngle;
					float bias = 0.0f, massScale = 1.0f, impulseScale = 0.0f;
					if (C > 0.0f)
					{
						bias = C * inv_h;
					}
					else if (useBias)
					{
						if (soft)
						{
							bias = s.biasCoefficient * C;
							massScale = s.massCoefficient;
							impulseScale = s.impulseCoefficient;
						}
						else
						{
							bias = S2_BAUMGARTE * inv_h * C;
						}
					}
					float Cdot = wB - wA;
					float impulse = soft ? -s.axialMass * massScale * (Cdot + bias) - impulseScale * s.lowerImpulse : -s.axialMass * (Cdot + bias);
					float oldImpulse = s.lowerImpulse;
					s.lowerImpulse = S2_MAXF(s.lowerImpulse + impulse, 0.0f);
					impulse = s.lowerImpulse - oldImpulse;
					wA -= iA * impulse;
					wB += iB * impulse;
				}
				{
					float C = s.upperAngle - jointAngle;
					float bias = 0.0f, massScale = 1.0f, impulseScale = 0.0f;
					if (C > 0.0f)
					{
						bias = C * inv_h;
					}
					else if (useBias)
					{
						if (soft)
						{
							bias = s.biasCoefficient * C;
							massScale = s.massCoefficient;
							impulseScale = s.impulseCoefficient;
						}
						else
						{
							bias = S2_BAUMGARTE * inv_h * C;
						}
					}
					float Cdot = wA - wB;
					// the soft term reads lowerImpulse in the reference (revolute_joint.c:595); kept verbatim
					float impulse = soft ? -s.axialMass * massScale * (Cdot + bias) - impulseScale * s.lowerImpulse : -s.axialMass * (Cdot + bias);
					float oldImpulse = s.upperImpulse;
					s.upperImpulse = S2_MAXF(s.upperImpulse + impulse, 0.0f);
					impulse = s.upperImpulse - oldImpulse;
					wA += iA * impulse;
					wB -= iB * impulse;
				}
			}
		}

		{
			V2 rA = rotate(qA, s.lA);
			V2 rB = rotate(qB, s.lB);
			V2 Cdot = sub(add(vB, crossSV(wB, rB)), add(vA, crossSV(wA, rA)));
			V2 impulse;
			if (KIND == JSOLVE_PLAIN)
			{
				impulse = mulMV(s.pivotMass, neg(Cdot));
			}
			else
			{
				V2 bias = v2(0.0f, 0.0f);
				float massScale = 1.0f, impulseScale = 0.0f;
				V2 dcA = v2(da.x, da.y), dcB = v2(db.x, db.y);
				if (KIND == JSOLVE_SOFT)
				{
					if (useBias)
					{
						V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), s.centerDiff0);
						bias = mulSV(s.biasCoefficient, separation);
						massScale = s.massCoefficient;
						impulseScale = s.impulseCoefficient;
					}
				}
				else
				{
					V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), s.centerDiff0);
					bias = mulSV(S2_BAUMGARTE * inv_h, separation);
				}
				M22 K = revoluteK(mA, mB, iA, iB, rA, rB);
				V2 bb = solve22(K, add(Cdot, bias));
				if (KIND == JSOLVE_SOFT)
				{
					impulse.x = -massScale * bb.x - impulseScale * s.impulse.x;
					impulse.y = -massScale * bb.y - impulseScale * s.impulse.y;
				}
				else
				{
					impulse.x = -bb.x;
					impulse.y = -bb.y;
				}
			}
			s.impulse.x += impulse.x;
			s.impulse.y += impulse.y;
			vA = mulSub(vA, mA, impulse);
			wA -= iA * cross(rA, impulse);
			vB = mulAdd(vB, mB, impulse);
			wB += iB * cross(rB, impulse);
		}
		storeJointImpulses(jv, k, s);
	}

	if (writeA)
	{
		b.setVel(s.ia, make_float4(vA.x, vA.y, wA, 0.0f));
	}
	if (writeB)
	{
		b.setVel(s.ib, make_float4(vB.x, vB.y, wB, 0.0f));
	}
}

