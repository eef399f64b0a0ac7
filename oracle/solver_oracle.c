// TEST INFRASTRUCTURE -- CPU oracle for the constraint-solve hot path.  NOT shipped, NOT a fallback.
//
// A plain-C, single-threaded restatement of the ten s2Solve_* variants of erincatto/solver2d on
// the s2amd wire arrays (include/solver2d_amd.h).  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this file's library; the product (solver2d_amd/csrc)
// never does and fails loudly without its HIP extension.
//
// PARITY PIN: tests/test_oracle_vs_reference.py runs the unmodified reference (oracle/_ref) and
// this restatement on identical captured inputs and requires BIT-IDENTICAL outputs for all ten
// solvers on every corpus scene; tests/golden/*.npz holds reference-generated vectors so the
// same pin is checked where /root/reference is absent.
//
// Two things the reference does not have:
//   * an explicit constraint ORDER.  contactOrder[k] / jointOrder[k] name the k-th constraint the
//     sweeps visit.  NULL = pool-index order = the reference's order (src/solve_tgs_soft.c:162-179).
//     A colour-sorted order makes this file the exact arithmetic twin of the batched GPU sweeps.
//   * for Jacobi, nothing: its contact pass is order-free by construction.
//
// Each function cites the reference lines it restates (paths relative to /root/reference).

#include "solver2d_amd.h"

#include <math.h>
#include <stdbool.h>
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__)
#define ORACLE_API __attribute__((visibility("default")))
#else
#define ORACLE_API
#endif

// ---- constants: include/solver2d/constants.h:6-22 ----
#define K_PI 3.14159265359f
#define K_LINEAR_SLOP 0.005f
#define K_ANGULAR_SLOP (2.0f / 180.0f * K_PI)
#define K_MAX_LINEAR_CORRECTION 0.2f
#define K_MAX_ANGULAR_CORRECTION (8.0f / 180.0f * K_PI)
#define K_BAUMGARTE 0.2f
#define K_MAX_BAUMGARTE_VELOCITY 4.0f
#define K_CONTACT_HERTZ 30.0f
#define K_JOINT_HERTZ 60.0f

// ---- math: include/solver2d/math.h ----
#define MIN_(A, B) ((A) < (B) ? (A) : (B))	   // math.h:10
#define MAX_(A, B) ((A) > (B) ? (A) : (B))	   // math.h:11
#define ABS_(A) ((A) > 0.0f ? (A) : -(A))	   // math.h:12
#define CLAMP_(A, B, C) MIN_(MAX_(A, B), C)   // math.h:13

typedef struct V2
{
	float x, y;
} V2;
typedef struct Rot
{
	float s, c;
} Rot;
typedef struct M22
{
	V2 cx, cy;
} M22;

static inline V2 v2(float x, float y) { V2 r = {x, y}; return r; }
static inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }				   // math.h:47
static inline float cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }				   // math.h:53
static inline V2 crossVS(V2 v, float s) { return v2(s * v.y, -s * v.x); }			   // math.h:60
static inline V2 crossSV(float s, V2 v) { return v2(-s * v.y, s * v.x); }			   // math.h:67
static inline V2 rightPerp(V2 v) { return v2(v.y, -v.x); }							   // math.h:73
static inline V2 add(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }				   // math.h:85
static inline V2 sub(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }				   // math.h:91
static inline V2 neg(V2 a) { return v2(-a.x, -a.y); }									   // math.h:97
static inline V2 mulSV(float s, V2 v) { return v2(s * v.x, s * v.y); }				   // math.h:115
static inline V2 mulAdd(V2 a, float s, V2 b) { return v2(a.x + s * b.x, a.y + s * b.y); } // math.h:121
static inline V2 mulSub(V2 a, float s, V2 b) { return v2(a.x - s * b.x, a.y - s * b.y); } // math.h:127
static inline float length(V2 v) { return sqrtf(v.x * v.x + v.y * v.y); }			   // math.h:171

// src/math.c:40-51 s2Normalize: zero vector for length < 0.001f * FLT_EPSILON
static inline V2 normalize(V2 v)
{
	float len = length(v);
	if (len < 0.001f * 1.19209290e-07f)
	{
		return v2(0.0f, 0.0f);
	}
	float inv = 1.0f / len;
	return v2(inv * v.x, inv * v.y);
}

static inline Rot normalizeRot(Rot q) // math.h:201-207 (note the double-typed 0.0 compare)
{
	float mag = sqrtf(q.s * q.s + q.c * q.c);
	float invMag = mag > 0.0 ? 1.0f / mag : 0.0f;
	Rot qn = {q.s * invMag, q.c * invMag};
	return qn;
}

static inline Rot integrateRot(Rot q1, float omegah) // math.h:209-223
{
	Rot q2 = {q1.s + omegah * q1.c, q1.c - omegah * q1.s};
	return normalizeRot(q2);
}

static inline float computeAngularVelocity(Rot q1, Rot q2, float inv_h) // math.h:238-252
{
	return inv_h * (q2.s * q1.c - q2.c * q1.s);
}

static inline float relativeAngle(Rot b, Rot a) // math.h:320-327
{
	float s = b.s * a.c - b.c * a.s;
	float c = b.c * a.c + b.s * a.s;
	return atan2f(s, c);
}

static inline V2 rotate(Rot q, V2 v) { return v2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }	   // math.h:330-341
static inline V2 invRotate(Rot q, V2 v) { return v2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); } // math.h:344-347

static inline V2 mulMV(M22 A, V2 v) // math.h:386-390
{
	return v2(A.cx.x * v.x + A.cy.x * v.y, A.cx.y * v.x + A.cy.y * v.y);
}

static inline M22 inverse22(M22 A) // math.h:392-406
{
	float a = A.cx.x, b = A.cy.x, c = A.cx.y, d = A.cy.y;
	M22 B;
	float det = a * d - b * c;
	if (det != 0.0f)
	{
		det = 1.0f / det;
	}
	B.cx.x = det * d;
	B.cy.x = -det * b;
	B.cx.y = -det * c;
	B.cy.y = det * a;
	return B;
}

static inline V2 solve22(M22 A, V2 b) // math.h:410-420
{
	float a11 = A.cx.x, a12 = A.cy.x, a21 = A.cx.y, a22 = A.cy.y;
	float det = a11 * a22 - a12 * a21;
	if (det != 0.0f)
	{
		det = 1.0f / det;
	}
	return v2(det * (a22 * b.x - a12 * b.y), det * (a11 * b.y - a21 * b.x));
}

// ---- working state ----

// src/body.h:16-76, solver-visible members
typedef struct Body
{
	V2 position, deltaPosition, deltaPosition0;
	Rot rot0, rot;
	V2 localCenter, linearVelocity;
	float angularVelocity;
	V2 dv;
	float dw;
	V2 force;
	float torque;
	float mass, invMass, I, invI;
	float linearDamping, angularDamping, gravityScale;
	int type;
} Body;

// src/solvers.h:26-56 (zeroed per step like the reference's stack allocator, stack_allocator.c:84)
typedef struct CPoint
{
	V2 rA0, rB0;
	V2 localAnchorA, localAnchorB;
	V2 localFrictionAnchorA, localFrictionAnchorB;
	float tangentSeparation, separation, adjustedSeparation;
	float normalImpulse, tangentImpulse;
	float normalMass, tangentMass;
	float massCoefficient, biasCoefficient, impulseCoefficient;
} CPoint;

typedef struct Constraint
{
	int contact; // index into the wire contact array
	int indexA, indexB;
	CPoint points[2];
	V2 normal;
	float friction;
	int pointCount;
} Constraint;

// src/joint.h:28-103
typedef struct Joint
{
	int wire; // index into the wire joint array
	int type;
	int indexA, indexB;
	V2 localOriginAnchorA, localOriginAnchorB;
	// shared
	V2 impulse;
	float motorImpulse, lowerImpulse, upperImpulse;
	bool enableMotor, enableLimit;
	float maxMotorTorque, motorSpeed, referenceAngle, lowerAngle, upperAngle;
	float hertz, dampingRatio;
	V2 targetA;
	// temp
	V2 localAnchorA, localAnchorB, centerDiff0;
	float invMassA, invMassB, invIA, invIB;
	M22 pivotMass;
	float biasCoefficient, massCoefficient, impulseCoefficient, axialMass;
} Joint;

// src/solvers.h:13-24
typedef struct Context
{
	float dt, inv_dt, h, inv_h;
	int iterations, extraIterations;
	bool warmStart;
	V2 gravity;
} Context;

typedef struct World
{
	Body* bodies;
	int bodyCapacity;
	s2amdContact* contacts; // wire (manifolds are read and written in place, like the reference)
	int contactCapacity;
	Constraint* constraints;
	int constraintCount;
	Joint* joints;
	int jointCount;
} World;

// ---------------------------------------------------------------------------------------------
// common stages: src/solve_common.c
// ---------------------------------------------------------------------------------------------

static void integrateVelocities(World* w, const Context* ctx, float h) // solve_common.c:10-45
{
	V2 gravity = ctx->gravity;
	for (int i = 0; i < w->bodyCapacity; ++i)
	{
		Body* body = w->bodies + i;
		if (body->type != S2AMD_BODY_DYNAMIC)
		{
			continue; // free slots have type -1
		}
		float invMass = body->invMass;
		float invI = body->invI;
		V2 v = body->linearVelocity;
		float wv = body->angularVelocity;
		v = add(v, mulSV(h * invMass, mulAdd(body->force, body->mass * body->gravityScale, gravity)));
		wv = wv + h * invI * body->torque;
		v = mulSV(1.0f / (1.0f + h * body->linearDamping), v);
		wv *= 1.0f / (1.0f + h * body->angularDamping);
		body->linearVelocity = v;
		body->angularVelocity = wv;
	}
}

static void integratePositions(World* w, float h) // solve_common.c:47-68
{
	for (int i = 0; i < w->bodyCapacity; ++i)
	{
		Body* body = w->bodies + i;
		if (body->type == S2AMD_BODY_FREE || body->type == S2AMD_BODY_STATIC)
		{
			continue;
		}
		body->deltaPosition = mulAdd(body->deltaPosition, h, body->linearVelocity);
		body->rot = integrateRot(body->rot, h * body->angularVelocity);
	}
}

static void finalizePositions(World* w) // solve_common.c:70-91
{
	for (int i = 0; i < w->bodyCapacity; ++i)
	{
		Body* body = w->bodies + i;
		if (body->type == S2AMD_BODY_FREE || body->type == S2AMD_BODY_STATIC)
		{
			continue;
		}
		body->position = add(body->position, body->deltaPosition);
		body->deltaPosition = v2(0.0f, 0.0f);
	}
}

enum PrepareKind
{
	PREP_PGS,	 // solve_common.c:93-168
	PREP_SOFT,	 // solve_common.c:188-274
	PREP_TGS,	 // solve_tgs_ngs.c:19-89
	PREP_STICKY, // solve_tgs_sticky.c:19-85 (first half)
	PREP_XPBD,	 // solve_xpbd.c:18-86
};

static void prepareContacts(World* w, enum PrepareKind kind, bool warmStart, float h, float hertz)
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		const s2amdContact* contact = w->contacts + constraint->contact;
		int pointCount = contact->pointCount;
		int indexA = contact->bodyA;
		int indexB = contact->bodyB;

		constraint->indexA = indexA;
		constraint->indexB = indexB;
		constraint->normal = v2(contact->normal[0], contact->normal[1]);
		constraint->friction = contact->friction;
		constraint->pointCount = pointCount;

		Body* bodyA = bodies + indexA;
		Body* bodyB = bodies + indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;

		// solve_common.c:219
		float contactHertz = (mA == 0.0f || mB == 0.0f) ? 2.0f * hertz : hertz;

		Rot qA = bodyA->rot, qB = bodyB->rot;
		V2 normal = constraint->normal;
		V2 tangent = rightPerp(normal);

		for (int j = 0; j < pointCount; ++j)
		{
			const s2amdManifoldPoint* mp = contact->points + j;
			CPoint* cp = constraint->points + j;

			bool copyImpulse;
			switch (kind)
			{
				case PREP_PGS:
					// cp->separation is still zero here (zeroed scratch), so the test is always true
					// when warm starting: solve_common.c:133 vs :152
					copyImpulse = warmStart && cp->separation <= 0.0f;
					break;
				case PREP_SOFT:
				case PREP_TGS:
					copyImpulse = warmStart;
					break;
				default:
					copyImpulse = false; // sticky :53-55, xpbd :52-53
					break;
			}
			if (copyImpulse)
			{
				cp->normalImpulse = mp->normalImpulse;
				cp->tangentImpulse = mp->tangentImpulse;
			}
			else
			{
				cp->normalImpulse = 0.0f;
				cp->tangentImpulse = 0.0f;
			}

			cp->localAnchorA = sub(v2(mp->localAnchorA[0], mp->localAnchorA[1]), bodyA->localCenter);
			cp->localAnchorB = sub(v2(mp->localAnchorB[0], mp->localAnchorB[1]), bodyB->localCenter);
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
			if (kind != PREP_TGS)
			{
				cp->rA0 = rA;
				cp->rB0 = rB;
			}

			cp->separation = mp->separation;
			cp->adjustedSeparation = mp->separation - dot(sub(rB, rA), normal);

			if (kind == PREP_PGS)
			{
				cp->biasCoefficient = mp->separation > 0.0f ? 1.0f : 0.0f;
			}
			else if (kind == PREP_XPBD)
			{
				cp->biasCoefficient = 0.0f;
			}

			float rtA = cross(rA, tangent);
			float rtB = cross(rB, tangent);
			float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
			cp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;

			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
			cp->normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;

			if (kind == PREP_SOFT)
			{
				// solve_common.c:262-271
				const float zeta = 10.0f;
				float omega = 2.0f * K_PI * contactHertz;
				float c = h * omega * (2.0f * zeta + h * omega);
				cp->biasCoefficient = omega / (2.0f * zeta + h * omega);
				cp->impulseCoefficient = 1.0f / (1.0f + c);
				cp->massCoefficient = c * cp->impulseCoefficient;
			}
		}
	}
}

// Second half of s2PrepareContacts_Sticky: friction anchor cache, solve_tgs_sticky.c:87-163.
// Reads AND writes the manifold.
static void prepareStickyFriction(World* w)
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		s2amdContact* manifold = w->contacts + constraint->contact;
		int pointCount = constraint->pointCount;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		Rot qA = bodyA->rot, qB = bodyB->rot;
		V2 normal = constraint->normal;
		V2 tangent = rightPerp(normal);
		V2 cA = bodyA->position;
		V2 cB = bodyB->position;

		bool frictionConfirmed = false;
		if (manifold->frictionPersisted)
		{
			int confirmCount = 0;
			for (int j = 0; j < pointCount; ++j)
			{
				const s2amdManifoldPoint* mp = manifold->points + j;
				CPoint* cp = constraint->points + j;

				V2 normalA = rotate(qA, v2(mp->frictionNormalA[0], mp->frictionNormalA[1]));
				V2 normalB = rotate(qB, v2(mp->frictionNormalB[0], mp->frictionNormalB[1]));
				float nn = dot(normalA, normalB);
				if (nn < 0.98f)
				{
					break;
				}

				cp->localFrictionAnchorA = sub(v2(mp->frictionAnchorA[0], mp->frictionAnchorA[1]), bodyA->localCenter);
				cp->localFrictionAnchorB = sub(v2(mp->frictionAnchorB[0], mp->frictionAnchorB[1]), bodyB->localCenter);
				V2 rAf = rotate(qA, cp->localFrictionAnchorA);
				V2 rBf = rotate(qB, cp->localFrictionAnchorB);
				V2 offset = add(sub(cB, cA), sub(rBf, rAf));
				float normalSeparation = dot(offset, normalA);
				if (ABS_(normalSeparation) > 2.0f * K_LINEAR_SLOP)
				{
					break;
				}

				cp->tangentSeparation = dot(sub(cB, cA), tangent);

				float rtA = cross(rAf, tangent);
				float rtB = cross(rBf, tangent);
				float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
				cp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;

				confirmCount += 1;
			}
			if (confirmCount == pointCount)
			{
				frictionConfirmed = true;
			}
		}

		if (frictionConfirmed == false)
		{
			for (int j = 0; j < pointCount; ++j)
			{
				s2amdManifoldPoint* mp = manifold->points + j;
				CPoint* cp = constraint->points + j;
				V2 rA = cp->rA0;
				V2 rB = cp->rB0;
				V2 fnA = invRotate(qA, normal);
				V2 fnB = invRotate(qB, normal);
				mp->frictionNormalA[0] = fnA.x, mp->frictionNormalA[1] = fnA.y;
				mp->frictionNormalB[0] = fnB.x, mp->frictionNormalB[1] = fnB.y;
				mp->frictionAnchorA[0] = mp->localAnchorA[0], mp->frictionAnchorA[1] = mp->localAnchorA[1];
				mp->frictionAnchorB[0] = mp->localAnchorB[0], mp->frictionAnchorB[1] = mp->localAnchorB[1];
				cp->localFrictionAnchorA = cp->localAnchorA;
				cp->localFrictionAnchorB = cp->localAnchorB;
				cp->tangentSeparation = dot(sub(cB, cA), tangent);
				float rtA = cross(rA, tangent);
				float rtB = cross(rB, tangent);
				float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
				cp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
			}
		}
		manifold->frictionPersisted = 1;
	}
}

// s2WarmStartContacts (solve_common.c:276-326, current anchors) and
// s2WarmStartContacts_Fixed (solve_soft_step.c:16-63, anchors rA0/rB0)
static void warmStartContacts(World* w, bool fixedAnchors)
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		int pointCount = constraint->pointCount;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;
		Rot qA = bodyA->rot, qB = bodyB->rot;
		V2 normal = constraint->normal;
		V2 tangent = rightPerp(normal);

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA = fixedAnchors ? cp->rA0 : rotate(qA, cp->localAnchorA);
			V2 rB = fixedAnchors ? cp->rB0 : rotate(qB, cp->localAnchorB);
			V2 P = add(mulSV(cp->normalImpulse, normal), mulSV(cp->tangentImpulse, tangent));
			wA -= iA * cross(rA, P);
			vA = mulAdd(vA, -mA, P);
			wB += iB * cross(rB, P);
			vB = mulAdd(vB, mB, P);
		}

		bodyA->linearVelocity = vA;
		bodyA->angularVelocity = wA;
		bodyB->linearVelocity = vB;
		bodyB->angularVelocity = wB;
	}
}

static void solveContact_NGS(World* w) // solve_common.c:328-394
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		V2 dcA = bodyA->deltaPosition;
		Rot qA = bodyA->rot;
		V2 dcB = bodyB->deltaPosition;
		Rot qB = bodyB->rot;
		V2 normal = constraint->normal;

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			if (cp->separation > 0.0f)
			{
				continue;
			}
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + cp->adjustedSeparation;
			float C = CLAMP_(K_BAUMGARTE * (separation + K_LINEAR_SLOP), -K_MAX_LINEAR_CORRECTION, 0.0f);
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

		bodyA->deltaPosition = dcA;
		bodyA->rot = qA;
		bodyB->deltaPosition = dcB;
		bodyB->rot = qB;
	}
}

static void storeContactImpulses(World* w, float scale, bool scaled) // solve_common.c:396-410, solve_xpbd.c:517-527
{
	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		s2amdContact* manifold = w->contacts + constraint->contact;
		for (int j = 0; j < constraint->pointCount; ++j)
		{
			if (scaled)
			{
				manifold->points[j].normalImpulse = constraint->points[j].normalImpulse * scale;
				manifold->points[j].tangentImpulse = constraint->points[j].tangentImpulse * scale;
			}
			else
			{
				manifold->points[j].normalImpulse = constraint->points[j].normalImpulse;
				manifold->points[j].tangentImpulse = constraint->points[j].tangentImpulse;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------
// velocity sweeps of the soft family.  One body of code, four reference functions:
//   SOFT_TGS    s2SolveContacts_TGS_Soft     solve_tgs_soft.c:17-135   current anchors everywhere
//   SOFT_PGS    s2SolveContacts_PGS_Soft     solve_pgs_soft.c:16-125   fixed anchors, bias cap -2
//   SOFT_JACOBI s2SolveContacts_Jacobi_Soft  solve_jacobi.c:21-132     fixed anchors, writes dv/dw
//   SOFT_FIXED  s2SolveContacts_TGS_Fixed    solve_soft_step.c:66-177  separation from current
//                                                                       anchors, Jacobians fixed, cap -2
// ---------------------------------------------------------------------------------------------
enum SoftKind
{
	SOFT_TGS,
	SOFT_PGS,
	SOFT_JACOBI,
	SOFT_FIXED
};

static void solveContactsSoft(World* w, enum SoftKind kind, float inv_h, bool useBias)
{
	Body* bodies = w->bodies;
	const float biasCap = (kind == SOFT_TGS || kind == SOFT_JACOBI) ? -K_MAX_BAUMGARTE_VELOCITY : -0.5f * K_MAX_BAUMGARTE_VELOCITY;

	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;

		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;

		V2 dcA = bodyA->deltaPosition;
		Rot qA = bodyA->rot;
		V2 dcB = bodyB->deltaPosition;
		Rot qB = bodyB->rot;

		V2 normal = constraint->normal;
		V2 tangent = rightPerp(normal);
		float friction = constraint->friction;

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;

			V2 rA, rB;
			float s;
			if (kind == SOFT_TGS)
			{
				rA = rotate(qA, cp->localAnchorA);
				rB = rotate(qB, cp->localAnchorB);
				V2 ds = add(sub(dcB, dcA), sub(rB, rA));
				s = dot(ds, normal) + cp->adjustedSeparation;
			}
			else if (kind == SOFT_FIXED)
			{
				V2 ds = add(sub(dcB, dcA), sub(rotate(qB, cp->localAnchorB), rotate(qA, cp->localAnchorA)));
				s = dot(ds, normal) + cp->adjustedSeparation;
				rA = cp->rA0;
				rB = cp->rB0;
			}
			else
			{
				s = cp->separation;
				rA = cp->rA0;
				rB = cp->rB0;
			}

			float bias = 0.0f;
			float massScale = 1.0f;
			float impulseScale = 0.0f;
			if (s > 0.0f)
			{
				bias = s * inv_h;
			}
			else if (useBias)
			{
				bias = MAX_(cp->biasCoefficient * s, biasCap);
				massScale = cp->massCoefficient;
				impulseScale = cp->impulseCoefficient;
			}

			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vn = dot(sub(vrB, vrA), normal);

			float impulse = -cp->normalMass * massScale * (vn + bias) - impulseScale * cp->normalImpulse;
			float newImpulse = MAX_(cp->normalImpulse + impulse, 0.0f);
			impulse = newImpulse - cp->normalImpulse;
			cp->normalImpulse = newImpulse;

			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA, rB;
			if (kind == SOFT_TGS)
			{
				rA = rotate(qA, cp->localAnchorA);
				rB = rotate(qB, cp->localAnchorB);
			}
			else
			{
				rA = cp->rA0;
				rB = cp->rB0;
			}

			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vt = dot(sub(vrB, vrA), tangent);

			float impulse = -cp->tangentMass * vt;
			float maxFriction = friction * cp->normalImpulse;
			float newImpulse = CLAMP_(cp->tangentImpulse + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - cp->tangentImpulse;
			cp->tangentImpulse = newImpulse;

			V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		if (kind == SOFT_JACOBI)
		{
			// solve_jacobi.c:126-130
			bodyA->dv = add(bodyA->dv, sub(vA, bodyA->linearVelocity));
			bodyA->dw += wA - bodyA->angularVelocity;
			bodyB->dv = add(bodyB->dv, sub(vB, bodyB->linearVelocity));
			bodyB->dw += wB - bodyB->angularVelocity;
		}
		else
		{
			bodyA->linearVelocity = vA;
			bodyA->angularVelocity = wA;
			bodyB->linearVelocity = vB;
			bodyB->angularVelocity = wB;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// rigid velocity sweeps
//   RIGID_BAUMGARTE  s2SolveContacts_PGS_Baumgarte  solve_pgs.c:17-122
//   RIGID_PGS        s2SolveContacts_PGS            solve_pgs_ngs.c:16-124  (friction first, no speculative)
//   RIGID_TGS        s2SolveContacts_TGS            solve_tgs_ngs.c:91-201
// ---------------------------------------------------------------------------------------------
static void solveContacts_PGS_Baumgarte(World* w, float inv_h)
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;
		V2 normal = constraint->normal;
		V2 tangent = rightPerp(normal);
		float friction = constraint->friction;

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			float bias = 0.0f;
			if (cp->separation > 0.0f)
			{
				bias = cp->separation * inv_h;
			}
			else
			{
				bias = MAX_(K_BAUMGARTE * inv_h * MIN_(0.0f, cp->separation + K_LINEAR_SLOP), -K_MAX_BAUMGARTE_VELOCITY);
			}
			V2 rA = cp->rA0, rB = cp->rB0;
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vn = dot(sub(vrB, vrA), normal);
			float impulse = -cp->normalMass * (vn + bias);
			float newImpulse = MAX_(cp->normalImpulse + impulse, 0.0f);
			impulse = newImpulse - cp->normalImpulse;
			cp->normalImpulse = newImpulse;
			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA = cp->rA0, rB = cp->rB0;
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 dv = sub(vrB, vrA);
			float vt = dot(dv, tangent);
			float lambda = cp->tangentMass * (-vt);
			float maxFriction = friction * cp->normalImpulse;
			float newImpulse = CLAMP_(cp->tangentImpulse + lambda, -maxFriction, maxFriction);
			lambda = newImpulse - cp->tangentImpulse;
			cp->tangentImpulse = newImpulse;
			V2 P = mulSV(lambda, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		bodyA->linearVelocity = vA;
		bodyA->angularVelocity = wA;
		bodyB->linearVelocity = vB;
		bodyB->angularVelocity = wB;
	}
}

static void solveContacts_PGS(World* w) // solve_pgs_ngs.c:16-124
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;
		V2 normal = constraint->normal;
		V2 tangent = crossVS(normal, 1.0f);
		float friction = constraint->friction;

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			if (cp->separation > 0.0f)
			{
				cp->tangentImpulse = 0.0f;
				continue;
			}
			V2 rA = cp->rA0, rB = cp->rB0;
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vt = dot(sub(vrB, vrA), tangent);
			float lambda = cp->tangentMass * (-vt);
			float maxFriction = friction * cp->normalImpulse;
			float newImpulse = CLAMP_(cp->tangentImpulse + lambda, -maxFriction, maxFriction);
			lambda = newImpulse - cp->tangentImpulse;
			cp->tangentImpulse = newImpulse;
			V2 P = mulSV(lambda, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(cp->rA0, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(cp->rB0, P);
		}

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			if (cp->separation > 0.0f)
			{
				cp->normalImpulse = 0.0f;
				continue;
			}
			V2 rA = cp->rA0, rB = cp->rB0;
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vn = dot(sub(vrB, vrA), normal);
			float impulse = -cp->normalMass * vn;
			float newImpulse = MAX_(cp->normalImpulse + impulse, 0.0f);
			impulse = newImpulse - cp->normalImpulse;
			cp->normalImpulse = newImpulse;
			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		bodyA->linearVelocity = vA;
		bodyA->angularVelocity = wA;
		bodyB->linearVelocity = vB;
		bodyB->angularVelocity = wB;
	}
}

static void solveContacts_TGS(World* w, float inv_h) // solve_tgs_ngs.c:91-201
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;
		V2 dcA = bodyA->deltaPosition, dcB = bodyB->deltaPosition;
		Rot qA = bodyA->rot, qB = bodyB->rot;
		V2 normal = constraint->normal;
		V2 tangent = rightPerp(normal);
		float friction = constraint->friction;

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + cp->adjustedSeparation;
			float bias = separation > 0.0f ? separation * inv_h : 0.0f;
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vn = dot(sub(vrB, vrA), normal);
			float impulse = -cp->normalMass * (vn + bias);
			float newImpulse = MAX_(cp->normalImpulse + impulse, 0.0f);
			impulse = newImpulse - cp->normalImpulse;
			cp->normalImpulse = newImpulse;
			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			float vt = dot(sub(vrB, vrA), tangent);
			float impulse = -cp->tangentMass * vt;
			float maxFriction = friction * cp->normalImpulse;
			float newImpulse = CLAMP_(cp->tangentImpulse + impulse, -maxFriction, maxFriction);
			impulse = newImpulse - cp->tangentImpulse;
			cp->tangentImpulse = newImpulse;
			V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		bodyA->linearVelocity = vA;
		bodyA->angularVelocity = wA;
		bodyB->linearVelocity = vB;
		bodyB->angularVelocity = wB;
	}
}

static void solveContacts_TGS_Sticky(World* w, float inv_h, bool useBias) // solve_tgs_sticky.c:167-310
{
	Body* bodies = w->bodies;
	float contactBaumgarte = 0.8f;
	float frictionBaumgarte = 0.5f;

	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;
		V2 dcA = bodyA->deltaPosition, dcB = bodyB->deltaPosition;
		Rot qA = bodyA->rot, qB = bodyB->rot;
		V2 normal = constraint->normal;
		V2 tangent = rightPerp(normal);
		float friction = constraint->friction;
		float totalNormalImpulse = 0.0f;

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
			V2 d = add(sub(dcB, dcA), sub(rB, rA));
			float separation = dot(d, normal) + cp->adjustedSeparation;
			float bias = 0.0f;
			if (separation > 0.0f)
			{
				bias = separation * inv_h;
			}
			else if (useBias)
			{
				bias = MAX_(-K_MAX_BAUMGARTE_VELOCITY, contactBaumgarte * separation * inv_h);
			}
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 vrB = add(vB, crossSV(wB, rB));
			float vn = dot(sub(vrB, vrA), normal);
			float impulse = -cp->normalMass * (vn + bias);
			float newImpulse = MAX_(cp->normalImpulse + impulse, 0.0f);
			impulse = newImpulse - cp->normalImpulse;
			cp->normalImpulse = newImpulse;
			totalNormalImpulse += cp->normalImpulse;
			V2 P = mulSV(impulse, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rAf = rotate(qA, cp->localFrictionAnchorA);
			V2 rBf = rotate(qB, cp->localFrictionAnchorB);
			V2 d = add(sub(dcB, dcA), sub(rBf, rAf));
			float separation = dot(d, tangent) + cp->tangentSeparation;
			float bias = useBias ? frictionBaumgarte * separation * inv_h : 0.0f;
			V2 vrA = add(vA, crossSV(wA, rAf));
			V2 vrB = add(vB, crossSV(wB, rBf));
			float vt = dot(sub(vrB, vrA), tangent);
			float impulse = -cp->tangentMass * (vt + bias);
			float maxFriction = 0.5f * friction * totalNormalImpulse;
			float newImpulse = cp->tangentImpulse + impulse;
			if (newImpulse < -maxFriction)
			{
				newImpulse = -maxFriction;
				w->contacts[constraint->contact].frictionPersisted = 0;
			}
			else if (newImpulse > maxFriction)
			{
				newImpulse = maxFriction;
				w->contacts[constraint->contact].frictionPersisted = 0;
			}
			impulse = newImpulse - cp->tangentImpulse;
			cp->tangentImpulse = newImpulse;
			V2 P = mulSV(impulse, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rAf, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rBf, P);
		}

		bodyA->linearVelocity = vA;
		bodyA->angularVelocity = wA;
		bodyB->linearVelocity = vB;
		bodyB->angularVelocity = wB;
	}
}

// ---------------------------------------------------------------------------------------------
// XPBD: solve_xpbd.c
// ---------------------------------------------------------------------------------------------
static void solveContactPositions_XPBD(World* w, float h) // solve_xpbd.c:88-216
{
	Body* bodies = w->bodies;
	float baseCompliance = 0.0f;

	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		float compliance = (mA == 0.0f || mB == 0.0f) ? 0.25f * baseCompliance : baseCompliance;
		V2 dcA = bodyA->deltaPosition;
		Rot qA = bodyA->rot;
		V2 dcB = bodyB->deltaPosition;
		Rot qB = bodyB->rot;
		V2 normal = constraint->normal;
		V2 tangent = crossVS(normal, 1.0f);

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
			V2 drA = sub(rA, cp->rA0);
			V2 drB = sub(rB, cp->rB0);
			V2 ds = add(sub(dcB, dcA), sub(drB, drA));
			float C = dot(ds, normal) + cp->separation;
			if (C > 0)
			{
				cp->normalImpulse = 0.0f;
				continue;
			}
			C = MAX_(-K_MAX_BAUMGARTE_VELOCITY * h, C);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float kA = mA + iA * rnA * rnA;
			float kB = mB + iB * rnB * rnB;
			float lambda = -C / (kA + kB + compliance);
			cp->normalImpulse = lambda;
			V2 P = mulSV(lambda, normal);
			dcA = mulSub(dcA, mA, P);
			qA = integrateRot(qA, -iA * cross(rA, P));
			dcB = mulAdd(dcB, mB, P);
			qB = integrateRot(qB, iB * cross(rB, P));
		}

		float friction = constraint->friction;
		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
			V2 drA = sub(rA, cp->rA0);
			V2 drB = sub(rB, cp->rB0);
			V2 dp = add(sub(dcB, dcA), sub(drB, drA));
			float C = dot(dp, tangent);
			float rtA = cross(rA, tangent);
			float rtB = cross(rB, tangent);
			float kA = mA + iA * rtA * rtA;
			float kB = mB + iB * rtB * rtB;
			float lambda = -C / (kA + kB);
			float maxLambda = friction * cp->normalImpulse;
			if (lambda < -maxLambda || maxLambda < lambda)
			{
				cp->tangentImpulse = 0.0f;
				continue;
			}
			cp->tangentImpulse = lambda;
			V2 P = mulSV(lambda, tangent);
			dcA = mulSub(dcA, mA, P);
			qA = integrateRot(qA, -iA * cross(rA, P));
			dcB = mulAdd(dcB, mB, P);
			qB = integrateRot(qB, iB * cross(rB, P));
		}

		bodyA->deltaPosition = dcA;
		bodyA->rot = qA;
		bodyB->deltaPosition = dcB;
		bodyB->rot = qB;
	}
}

static void solveContactVelocities_XPBD(World* w, float h) // solve_xpbd.c:218-338
{
	Body* bodies = w->bodies;
	float inv_h = h > 0.0f ? 1.0f / h : 0.0f;

	for (int i = 0; i < w->constraintCount; ++i)
	{
		Constraint* constraint = w->constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		Rot qA = bodyA->rot, qB = bodyB->rot;
		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;
		V2 normal = constraint->normal;
		V2 tangent = crossVS(normal, 1.0f);
		float friction = constraint->friction;

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			if (cp->normalImpulse == 0.0f)
			{
				continue;
			}
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
			V2 vrB = add(vB, crossSV(wB, rB));
			V2 vrA = add(vA, crossSV(wA, rA));
			V2 dv = sub(vrB, vrA);
			float rnA = cross(rA, normal);
			float rnB = cross(rB, normal);
			float kA = mA + iA * rnA * rnA;
			float kB = mB + iB * rnB * rnB;
			float vn = dot(dv, normal);
			float Cdot = vn;
			float lambda = -Cdot / (kA + kB);
			V2 P = mulSV(lambda, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		for (int j = 0; j < pointCount; ++j)
		{
			CPoint* cp = constraint->points + j;
			V2 rA = rotate(qA, cp->localAnchorA);
			V2 rB = rotate(qB, cp->localAnchorB);
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
			float maxFrictionImpulse = friction * cp->normalImpulse;
			float huf = (maxFrictionImpulse * inv_h) * (kA + kB);
			float abs_vt = ABS_(vt);
			float Cdot = (vt / abs_vt) * MIN_(huf, abs_vt);
			float lambda = -Cdot / (kA + kB);
			cp->tangentImpulse = lambda;
			V2 P = mulSV(lambda, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(rB, P);
		}

		bodyA->linearVelocity = vA;
		bodyA->angularVelocity = wA;
		bodyB->linearVelocity = vB;
		bodyB->angularVelocity = wB;
	}
}

// ---------------------------------------------------------------------------------------------
// joints: src/revolute_joint.c, src/mouse_joint.c, dispatch src/joint.c:294-465
// ---------------------------------------------------------------------------------------------

static void softCoefficients(float h, float zeta, float omega, float* bias, float* mass, float* impulse)
{
	// revolute_joint.c:469-475, mouse_joint.c:50-57
	*bias = omega / (2.0f * zeta + h * omega);
	float c = h * omega * (2.0f * zeta + h * omega);
	*impulse = 1.0f / (1.0f + c);
	*mass = c * (*impulse);
}

static void prepareMouse(World* w, Joint* joint, const Context* ctx) // mouse_joint.c:31-83
{
	Body* bodyB = w->bodies + joint->indexB;
	float mB = bodyB->invMass, iB = bodyB->invI;
	joint->localAnchorB = sub(joint->localOriginAnchorB, bodyB->localCenter);
	joint->invMassB = mB;
	joint->invIB = iB;
	{
		float h = ctx->h;
		float zeta = joint->dampingRatio;
		float omega = 2.0f * K_PI * joint->hertz;
		softCoefficients(h, zeta, omega, &joint->biasCoefficient, &joint->massCoefficient, &joint->impulseCoefficient);
	}
	Rot qB = bodyB->rot;
	V2 rB = rotate(qB, joint->localAnchorB);
	M22 K;
	K.cx.x = mB + iB * rB.y * rB.y;
	K.cx.y = -iB * rB.x * rB.y;
	K.cy.x = K.cx.y;
	K.cy.y = mB + iB * rB.x * rB.x;
	joint->pivotMass = inverse22(K);
	V2 cB = bodyB->position;
	joint->centerDiff0 = sub(cB, joint->targetA);
}

static void warmStartMouse(World* w, Joint* joint) // mouse_joint.c:85-107
{
	Body* bodyB = w->bodies + joint->indexB;
	Rot qB = bodyB->rot;
	V2 rB = rotate(qB, joint->localAnchorB);
	V2 vB = bodyB->linearVelocity;
	float wB = bodyB->angularVelocity;
	vB = mulAdd(vB, joint->invMassB, joint->impulse);
	wB += joint->invIB * (cross(rB, joint->impulse) + joint->motorImpulse);
	bodyB->linearVelocity = vB;
	bodyB->angularVelocity = wB;
}

static void solveMouse(World* w, Joint* joint, const Context* ctx) // mouse_joint.c:109-167
{
	Body* bodyB = w->bodies + joint->indexB;
	V2 vB = bodyB->linearVelocity;
	float wB = bodyB->angularVelocity;
	float mB = joint->invMassB;
	float iB = joint->invIB;
	{
		float h = ctx->h;
		float zeta = 0.1f;
		float omega = 2.0f * K_PI * 0.5f;
		float c = h * omega * (2.0f * zeta + h * omega);
		float impulseScale = 1.0f / (1.0f + c);
		float massScale = c * impulseScale;
		float impulse = -massScale * bodyB->I * wB - impulseScale * joint->motorImpulse;
		joint->motorImpulse += impulse;
		wB += iB * impulse;
	}
	{
		Rot qB = bodyB->rot;
		V2 rB = rotate(qB, joint->localAnchorB);
		V2 Cdot = add(vB, crossSV(wB, rB));
		V2 dcB = bodyB->deltaPosition;
		V2 separation = add(add(dcB, rB), joint->centerDiff0);
		V2 bias = mulSV(joint->biasCoefficient, separation);
		float massScale = joint->massCoefficient;
		float impulseScale = joint->impulseCoefficient;
		V2 b = mulMV(joint->pivotMass, add(Cdot, bias));
		V2 impulse;
		impulse.x = -massScale * b.x - impulseScale * joint->impulse.x;
		impulse.y = -massScale * b.y - impulseScale * joint->impulse.y;
		joint->impulse.x += impulse.x;
		joint->impulse.y += impulse.y;
		vB = mulAdd(vB, mB, impulse);
		wB += iB * cross(rB, impulse);
	}
	bodyB->linearVelocity = vB;
	bodyB->angularVelocity = wB;
}

// Shared tail of s2PrepareRevolute (revolute_joint.c:77-105) and s2PrepareRevolute_Soft (:477-505)
static void revoluteResetImpulses(Joint* joint, bool warmStart)
{
	float iA = joint->invIA, iB = joint->invIB;
	joint->axialMass = iA + iB;
	bool fixedRotation;
	if (joint->axialMass > 0.0f)
	{
		joint->axialMass = 1.0f / joint->axialMass;
		fixedRotation = false;
	}
	else
	{
		fixedRotation = true;
	}
	if (joint->enableLimit == false || fixedRotation || warmStart == false)
	{
		joint->lowerImpulse = 0.0f;
		joint->upperImpulse = 0.0f;
	}
	if (joint->enableMotor == false || fixedRotation || warmStart == false)
	{
		joint->motorImpulse = 0.0f;
	}
	if (warmStart == false)
	{
		joint->impulse = v2(0.0f, 0.0f);
	}
}

static M22 revoluteK(float mA, float mB, float iA, float iB, V2 rA, V2 rB)
{
	// revolute_joint.c:70-74, :461-465, :631-636, :768-773
	M22 K;
	K.cx.x = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
	K.cy.x = -rA.y * rA.x * iA - rB.y * rB.x * iB;
	K.cx.y = K.cy.x;
	K.cy.y = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
	return K;
}

// s2PrepareRevolute (revolute_joint.c:30-105) when soft == false,
// s2PrepareRevolute_Soft (revolute_joint.c:421-506) when soft == true
static void prepareRevolute(World* w, Joint* joint, bool soft, float h, float hertz, bool warmStart)
{
	Body* bodyA = w->bodies + joint->indexA;
	Body* bodyB = w->bodies + joint->indexB;
	const float inertiaScale = 1.0f; // joint.c:216, revolute_joint.c:46/50
	joint->localAnchorA = sub(joint->localOriginAnchorA, bodyA->localCenter);
	joint->invMassA = bodyA->invMass;
	joint->invIA = soft ? bodyA->invI : inertiaScale * bodyA->invI;
	joint->localAnchorB = sub(joint->localOriginAnchorB, bodyB->localCenter);
	joint->invMassB = bodyB->invMass;
	joint->invIB = soft ? bodyB->invI : inertiaScale * bodyB->invI;
	joint->centerDiff0 = sub(bodyB->position, bodyA->position);

	Rot qA = bodyA->rot, qB = bodyB->rot;
	V2 rA = rotate(qA, joint->localAnchorA);
	V2 rB = rotate(qB, joint->localAnchorB);
	float mA = joint->invMassA, mB = joint->invMassB;
	float iA = joint->invIA, iB = joint->invIB;
	joint->pivotMass = inverse22(revoluteK(mA, mB, iA, iB, rA, rB));

	if (soft)
	{
		const float zeta = 10.0f;
		float omega = 2.0f * K_PI * hertz;
		softCoefficients(h, zeta, omega, &joint->biasCoefficient, &joint->massCoefficient, &joint->impulseCoefficient);
	}
	revoluteResetImpulses(joint, warmStart);
}

static void prepareRevolute_XPBD(World* w, Joint* joint) // revolute_joint.c:792-823
{
	Body* bodyA = w->bodies + joint->indexA;
	Body* bodyB = w->bodies + joint->indexB;
	joint->localAnchorA = sub(joint->localOriginAnchorA, bodyA->localCenter);
	joint->invMassA = bodyA->invMass;
	joint->invIA = bodyA->invI;
	joint->localAnchorB = sub(joint->localOriginAnchorB, bodyB->localCenter);
	joint->invMassB = bodyB->invMass;
	joint->invIB = bodyB->invI;
	joint->centerDiff0 = sub(bodyB->position, bodyA->position);
	memset(&joint->pivotMass, 0, sizeof(joint->pivotMass));
	joint->axialMass = 0.0f;
	joint->impulse = v2(0.0f, 0.0f);
	joint->lowerImpulse = 0.0f;
	joint->upperImpulse = 0.0f;
	joint->motorImpulse = 0.0f;
}

static void warmStartRevolute(World* w, Joint* joint) // revolute_joint.c:107-150
{
	Body* bodyA = w->bodies + joint->indexA;
	Body* bodyB = w->bodies + joint->indexB;
	Rot qA = bodyA->rot;
	V2 vA = bodyA->linearVelocity;
	float wA = bodyA->angularVelocity;
	Rot qB = bodyB->rot;
	V2 vB = bodyB->linearVelocity;
	float wB = bodyB->angularVelocity;
	V2 rA = rotate(qA, joint->localAnchorA);
	V2 rB = rotate(qB, joint->localAnchorB);
	float mA = joint->invMassA, mB = joint->invMassB;
	float iA = joint->invIA, iB = joint->invIB;
	float axialImpulse = joint->motorImpulse + joint->lowerImpulse - joint->upperImpulse;
	V2 P = {joint->impulse.x, joint->impulse.y};
	vA = mulSub(vA, mA, P);
	wA -= iA * (cross(rA, P) + axialImpulse);
	vB = mulAdd(vB, mB, P);
	wB += iB * (cross(rB, P) + axialImpulse);
	bodyA->linearVelocity = vA;
	bodyA->angularVelocity = wA;
	bodyB->linearVelocity = vB;
	bodyB->angularVelocity = wB;
}

// motor row shared by all three velocity variants: revolute_joint.c:175-187, :526-538, :678-690
static void revoluteMotor(Joint* joint, float h, float* wA, float* wB, float iA, float iB)
{
	float Cdot = *wB - *wA - joint->motorSpeed;
	float impulse = -joint->axialMass * Cdot;
	float oldImpulse = joint->motorImpulse;
	float maxImpulse = h * joint->maxMotorTorque;
	joint->motorImpulse = CLAMP_(joint->motorImpulse + impulse, -maxImpulse, maxImpulse);
	impulse = joint->motorImpulse - oldImpulse;
	*wA -= iA * impulse;
	*wB += iB * impulse;
}

static void solveRevolute(World* w, Joint* joint, float h) // revolute_joint.c:152-303
{
	Body* bodyA = w->bodies + joint->indexA;
	Body* bodyB = w->bodies + joint->indexB;
	Rot qA = bodyA->rot, qB = bodyB->rot;
	V2 vA = bodyA->linearVelocity;
	float wA = bodyA->angularVelocity;
	V2 vB = bodyB->linearVelocity;
	float wB = bodyB->angularVelocity;
	float mA = joint->invMassA, mB = joint->invMassB;
	float iA = joint->invIA, iB = joint->invIB;
	bool fixedRotation = (iA + iB == 0.0f);

	if (joint->enableMotor && fixedRotation == false)
	{
		revoluteMotor(joint, h, &wA, &wB, iA, iB);
	}

	if (joint->enableLimit && fixedRotation == false)
	{
		float angle = relativeAngle(qB, qA) - joint->referenceAngle;
		{
			float C = angle - joint->lowerAngle;
			float Cdot = wB - wA;
			float impulse = -joint->axialMass * (Cdot + MAX_(C, 0.0f) / h);
			float oldImpulse = joint->lowerImpulse;
			joint->lowerImpulse = MAX_(joint->lowerImpulse + impulse, 0.0f);
			impulse = joint->lowerImpulse - oldImpulse;
			wA -= iA * impulse;
			wB += iB * impulse;
		}
		{
			float C = joint->upperAngle - angle;
			float Cdot = wA - wB;
			float impulse = -joint->axialMass * (Cdot + MAX_(C, 0.0f) / h);
			float oldImpulse = joint->upperImpulse;
			joint->upperImpulse = MAX_(joint->upperImpulse + impulse, 0.0f);
			impulse = joint->upperImpulse - oldImpulse;
			wA += iA * impulse;
			wB -= iB * impulse;
		}
	}

	{
		V2 rA = rotate(qA, joint->localAnchorA);
		V2 rB = rotate(qB, joint->localAnchorB);
		V2 Cdot = sub(add(vB, crossSV(wB, rB)), add(vA, crossSV(wA, rA)));
		V2 impulse = mulMV(joint->pivotMass, neg(Cdot));
		joint->impulse.x += impulse.x;
		joint->impulse.y += impulse.y;
		vA = mulSub(vA, mA, impulse);
		wA -= iA * cross(rA, impulse);
		vB = mulAdd(vB, mB, impulse);
		wB += iB * cross(rB, impulse);
	}

	bodyA->linearVelocity = vA;
	bodyA->angularVelocity = wA;
	bodyB->linearVelocity = vB;
	bodyB->angularVelocity = wB;
}

static void solveRevolutePosition(World* w, Joint* joint) // revolute_joint.c:305-419
{
	Body* bodyA = w->bodies + joint->indexA;
	Body* bodyB = w->bodies + joint->indexB;
	V2 dcA = bodyA->deltaPosition;
	Rot qA = bodyA->rot;
	V2 dcB = bodyB->deltaPosition;
	Rot qB = bodyB->rot;
	bool fixedRotation = (joint->invIA + joint->invIB == 0.0f);

	if (joint->enableLimit && fixedRotation == false)
	{
		float angle = relativeAngle(qB, qA) - joint->referenceAngle;
		float C = 0.0f;
		if (ABS_(joint->upperAngle - joint->lowerAngle) < 2.0f * K_ANGULAR_SLOP)
		{
			C = CLAMP_(angle - joint->lowerAngle, -K_MAX_ANGULAR_CORRECTION, K_MAX_ANGULAR_CORRECTION);
		}
		else if (angle <= joint->lowerAngle)
		{
			C = CLAMP_(angle - joint->lowerAngle + K_ANGULAR_SLOP, -K_MAX_ANGULAR_CORRECTION, 0.0f);
		}
		else if (angle >= joint->upperAngle)
		{
			C = CLAMP_(angle - joint->upperAngle - K_ANGULAR_SLOP, 0.0f, K_MAX_ANGULAR_CORRECTION);
		}
		float limitImpulse = -joint->axialMass * C;
		qA = integrateRot(qA, -joint->invIA * limitImpulse);
		qB = integrateRot(qB, joint->invIB * limitImpulse);
	}

	{
		V2 rA = rotate(qA, joint->localAnchorA);
		V2 rB = rotate(qB, joint->localAnchorB);
		V2 C = add(add(sub(dcB, dcA), sub(rB, rA)), joint->centerDiff0);
		float mA = joint->invMassA, mB = joint->invMassB;
		float iA = joint->invIA, iB = joint->invIB;
		// S2_FRESH_PIVOT_MASS == 1 (revolute_joint.c:15, :388-395) -- note the operand order differs
		// from revoluteK(): iA * rA.y * rA.y, not rA.y * rA.y * iA
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

	bodyA->deltaPosition = dcA;
	bodyA->rot = qA;
	bodyB->deltaPosition = dcB;
	bodyB->rot = qB;
}

// s2SolveRevolute_Soft (revolute_joint.c:508-657) when soft == true,
// s2SolveRevolute_Baumgarte (revolute_joint.c:660-790) when soft == false
static void solveRevoluteBiased(World* w, Joint* joint, bool soft, float h, float inv_h, bool useBias)
{
	Body* bodyA = w->bodies + joint->indexA;
	Body* bodyB = w->bodies + joint->indexB;
	V2 vA = bodyA->linearVelocity;
	float wA = bodyA->angularVelocity;
	V2 vB = bodyB->linearVelocity;
	float wB = bodyB->angularVelocity;
	float mA = joint->invMassA, mB = joint->invMassB;
	float iA = joint->invIA, iB = joint->invIB;
	bool fixedRotation = (iA + iB == 0.0f);

	if (joint->enableMotor && fixedRotation == false)
	{
		revoluteMotor(joint, h, &wA, &wB, iA, iB);
	}

	if (joint->enableLimit && fixedRotation == false)
	{
		float jointAngle = relativeAngle(bodyB->rot, bodyA->rot) - joint->referenceAngle;
		{
			float C = jointAngle - joint->lowerAngle;
			float bias = 0.0f, massScale = 1.0f, impulseScale = 0.0f;
			if (C > 0.0f)
			{
				bias = C * inv_h;
			}
			else if (useBias)
			{
				if (soft)
				{
					bias = joint->biasCoefficient * C;
					massScale = joint->massCoefficient;
					impulseScale = joint->impulseCoefficient;
				}
				else
				{
					bias = K_BAUMGARTE * inv_h * C;
				}
			}
			float Cdot = wB - wA;
			float impulse = soft ? -joint->axialMass * massScale * (Cdot + bias) - impulseScale * joint->lowerImpulse
								 : -joint->axialMass * (Cdot + bias);
			float oldImpulse = joint->lowerImpulse;
			joint->lowerImpulse = MAX_(joint->lowerImpulse + impulse, 0.0f);
			impulse = joint->lowerImpulse - oldImpulse;
			wA -= iA * impulse;
			wB += iB * impulse;
		}
		{
			float C = joint->upperAngle - jointAngle;
			float bias = 0.0f, massScale = 1.0f, impulseScale = 0.0f;
			if (C > 0.0f)
			{
				bias = C * inv_h;
			}
			else if (useBias)
			{
				if (soft)
				{
					bias = joint->biasCoefficient * C;
					massScale = joint->massCoefficient;
					impulseScale = joint->impulseCoefficient;
				}
				else
				{
					bias = K_BAUMGARTE * inv_h * C;
				}
			}
			float Cdot = wA - wB;
			// reference quirk kept verbatim: the soft term uses lowerImpulse (revolute_joint.c:595)
			float impulse = soft ? -joint->axialMass * massScale * (Cdot + bias) - impulseScale * joint->lowerImpulse
								 : -joint->axialMass * (Cdot + bias);
			float oldImpulse = joint->upperImpulse;
			joint->upperImpulse = MAX_(joint->upperImpulse + impulse, 0.0f);
			impulse = joint->upperImpulse - oldImpulse;
			wA += iA * impulse;
			wB -= iB * impulse;
		}
	}

	{
		Rot qA = bodyA->rot, qB = bodyB->rot;
		V2 rA = rotate(qA, joint->localAnchorA);
		V2 rB = rotate(qB, joint->localAnchorB);
		V2 Cdot = sub(add(vB, crossSV(wB, rB)), add(vA, crossSV(wA, rA)));
		V2 bias = v2(0.0f, 0.0f);
		float massScale = 1.0f, impulseScale = 0.0f;
		if (soft)
		{
			if (useBias)
			{
				V2 dcA = bodyA->deltaPosition, dcB = bodyB->deltaPosition;
				V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), joint->centerDiff0);
				bias = mulSV(joint->biasCoefficient, separation);
				massScale = joint->massCoefficient;
				impulseScale = joint->impulseCoefficient;
			}
		}
		else
		{
			// Baumgarte variant ignores useBias for the point constraint (revolute_joint.c:764-765)
			V2 dcA = bodyA->deltaPosition, dcB = bodyB->deltaPosition;
			V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), joint->centerDiff0);
			bias = mulSV(K_BAUMGARTE * inv_h, separation);
		}
		M22 K = revoluteK(mA, mB, iA, iB, rA, rB);
		V2 b = solve22(K, add(Cdot, bias));
		V2 impulse;
		if (soft)
		{
			impulse.x = -massScale * b.x - impulseScale * joint->impulse.x;
			impulse.y = -massScale * b.y - impulseScale * joint->impulse.y;
		}
		else
		{
			impulse.x = -b.x;
			impulse.y = -b.y;
		}
		joint->impulse.x += impulse.x;
		joint->impulse.y += impulse.y;
		vA = mulSub(vA, mA, impulse);
		wA -= iA * cross(rA, impulse);
		vB = mulAdd(vB, mB, impulse);
		wB += iB * cross(rB, impulse);
	}

	bodyA->linearVelocity = vA;
	bodyA->angularVelocity = wA;
	bodyB->linearVelocity = vB;
	bodyB->angularVelocity = wB;
}

static void solveRevolute_XPBD(World* w, Joint* joint) // revolute_joint.c:825-888
{
	float compliance = 0.0f;
	Body* bodyA = w->bodies + joint->indexA;
	Body* bodyB = w->bodies + joint->indexB;
	V2 dcA = bodyA->deltaPosition;
	Rot qA = bodyA->rot;
	V2 dcB = bodyB->deltaPosition;
	Rot qB = bodyB->rot;
	{
		V2 rA = rotate(qA, joint->localAnchorA);
		V2 rB = rotate(qB, joint->localAnchorB);
		V2 separation = add(add(sub(dcB, dcA), sub(rB, rA)), joint->centerDiff0);
		float c = length(separation);
		V2 n = normalize(separation);
		float mA = joint->invMassA, mB = joint->invMassB;
		if (mA == 0.0f && mB == 0.0f)
		{
			return;
		}
		float iA = joint->invIA, iB = joint->invIB;
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
	}
	bodyA->deltaPosition = dcA;
	bodyA->rot = qA;
	bodyB->deltaPosition = dcB;
	bodyB->rot = qB;
}

// dispatchers: src/joint.c:294-465
static void prepareJoint(World* w, Joint* j, const Context* ctx, bool warmStart) // joint.c:297-312
{
	if (j->type == S2AMD_JOINT_MOUSE)
		prepareMouse(w, j, ctx);
	else
		prepareRevolute(w, j, false, 0.0f, 0.0f, warmStart);
}
static void prepareJoint_Soft(World* w, Joint* j, const Context* ctx, float h, float hertz, bool warmStart) // joint.c:372-387
{
	if (j->type == S2AMD_JOINT_MOUSE)
		prepareMouse(w, j, ctx);
	else
		prepareRevolute(w, j, true, h, hertz, warmStart);
}
static void prepareJoint_XPBD(World* w, Joint* j, const Context* ctx) // joint.c:432-447
{
	if (j->type == S2AMD_JOINT_MOUSE)
		prepareMouse(w, j, ctx);
	else
		prepareRevolute_XPBD(w, j);
}
static void warmStartJoint(World* w, Joint* j) // joint.c:317-332
{
	if (j->type == S2AMD_JOINT_MOUSE)
		warmStartMouse(w, j);
	else
		warmStartRevolute(w, j);
}
static void solveJoint(World* w, Joint* j, const Context* ctx, float h) // joint.c:337-352
{
	if (j->type == S2AMD_JOINT_MOUSE)
		solveMouse(w, j, ctx);
	else
		solveRevolute(w, j, h);
}
static void solveJointPosition(World* w, Joint* j) // joint.c:356-367
{
	if (j->type == S2AMD_JOINT_REVOLUTE)
		solveRevolutePosition(w, j);
}
static void solveJoint_Soft(World* w, Joint* j, const Context* ctx, float h, float inv_h, bool useBias) // joint.c:391-409
{
	if (j->type == S2AMD_JOINT_MOUSE)
	{
		if (useBias)
			solveMouse(w, j, ctx);
	}
	else
		solveRevoluteBiased(w, j, true, h, inv_h, useBias);
}
static void solveJoint_Baumgarte(World* w, Joint* j, const Context* ctx, float h, float inv_h, bool useBias) // joint.c:413-428
{
	if (j->type == S2AMD_JOINT_MOUSE)
		solveMouse(w, j, ctx);
	else
		solveRevoluteBiased(w, j, false, h, inv_h, useBias);
}
static void solveJoint_XPBD(World* w, Joint* j, const Context* ctx) // joint.c:451-465
{
	if (j->type == S2AMD_JOINT_MOUSE)
		solveMouse(w, j, ctx);
	else
		solveRevolute_XPBD(w, j);
}

#define FOR_JOINTS(W, J) for (Joint* J = (W)->joints; J < (W)->joints + (W)->jointCount; ++J)

// ---------------------------------------------------------------------------------------------
// drivers
// ---------------------------------------------------------------------------------------------

static void solve_TGS_Soft(World* w, const Context* ctx, bool fixedAnchors)
{
	// s2Solve_TGS_Soft solve_tgs_soft.c:138-280 ; s2Solve_SoftStep solve_soft_step.c:182-311
	int substepCount = ctx->iterations;
	float h = ctx->h, inv_h = ctx->inv_h;
	float contactHertz = MIN_(K_CONTACT_HERTZ, 0.25f * inv_h);
	float jointHertz = fixedAnchors ? MIN_(K_JOINT_HERTZ, 0.25f * inv_h) : MIN_(K_JOINT_HERTZ, 0.125f * inv_h);

	prepareContacts(w, PREP_SOFT, ctx->warmStart, h, contactHertz);
	FOR_JOINTS(w, j) { prepareJoint_Soft(w, j, ctx, h, jointHertz, true); }

	for (int substep = 0; substep < substepCount; ++substep)
	{
		integrateVelocities(w, ctx, h);
		if (ctx->warmStart)
		{
			FOR_JOINTS(w, j) { warmStartJoint(w, j); }
			warmStartContacts(w, fixedAnchors);
		}
		FOR_JOINTS(w, j) { solveJoint_Soft(w, j, ctx, h, inv_h, true); }
		solveContactsSoft(w, fixedAnchors ? SOFT_FIXED : SOFT_TGS, inv_h, true);
		integratePositions(w, h);
		if (ctx->extraIterations > 0)
		{
			FOR_JOINTS(w, j) { solveJoint_Soft(w, j, ctx, h, inv_h, false); }
			solveContactsSoft(w, fixedAnchors ? SOFT_FIXED : SOFT_TGS, inv_h, false);
		}
	}
	finalizePositions(w);
	storeContactImpulses(w, 0.0f, false);
}

static void jacobiApply(World* w) // solve_jacobi.c:233-245
{
	for (int i = 0; i < w->bodyCapacity; ++i)
	{
		Body* body = w->bodies + i;
		if (body->type == S2AMD_BODY_FREE)
		{
			continue;
		}
		body->linearVelocity = add(body->linearVelocity, body->dv);
		body->angularVelocity += body->dw;
		body->dv = v2(0.0f, 0.0f);
		body->dw = 0.0f;
	}
}

static void solve_Jacobi_or_PGS_Soft(World* w, const Context* ctx, bool jacobi)
{
	// s2Solve_Jacobi solve_jacobi.c:134-292 ; s2Solve_PGS_Soft solve_pgs_soft.c:127-242
	int velocityIterations = ctx->iterations;
	int positionIterations = ctx->extraIterations;
	float h = ctx->dt, inv_h = ctx->inv_dt;
	float contactHertz = MIN_(K_CONTACT_HERTZ, 0.333f * inv_h);
	float jointHertz = MIN_(K_JOINT_HERTZ, 0.5f * inv_h);

	if (jacobi)
	{
		for (int i = 0; i < w->bodyCapacity; ++i)
		{
			w->bodies[i].dv = v2(0.0f, 0.0f);
			w->bodies[i].dw = 0.0f;
		}
	}

	integrateVelocities(w, ctx, h);
	prepareContacts(w, PREP_SOFT, ctx->warmStart, h, contactHertz);
	if (ctx->warmStart)
	{
		warmStartContacts(w, false);
	}
	FOR_JOINTS(w, j)
	{
		prepareJoint_Soft(w, j, ctx, h, jointHertz, ctx->warmStart);
		if (ctx->warmStart)
		{
			warmStartJoint(w, j);
		}
	}

	for (int iter = 0; iter < velocityIterations; ++iter)
	{
		FOR_JOINTS(w, j) { solveJoint_Soft(w, j, ctx, h, inv_h, true); }
		solveContactsSoft(w, jacobi ? SOFT_JACOBI : SOFT_PGS, inv_h, true);
		if (jacobi)
		{
			jacobiApply(w);
		}
	}
	integratePositions(w, h);
	for (int iter = 0; iter < positionIterations; ++iter)
	{
		FOR_JOINTS(w, j) { solveJoint_Soft(w, j, ctx, h, inv_h, false); }
		solveContactsSoft(w, jacobi ? SOFT_JACOBI : SOFT_PGS, inv_h, false);
		if (jacobi)
		{
			jacobiApply(w);
		}
	}
	finalizePositions(w);
	storeContactImpulses(w, 0.0f, false);
}

static void solve_PGS(World* w, const Context* ctx) // solve_pgs.c:125-213
{
	int iterations = ctx->iterations;
	float h = ctx->dt, inv_h = ctx->inv_dt;
	integrateVelocities(w, ctx, h);
	prepareContacts(w, PREP_PGS, ctx->warmStart, 0.0f, 0.0f);
	if (ctx->warmStart)
	{
		warmStartContacts(w, false);
	}
	FOR_JOINTS(w, j)
	{
		prepareJoint(w, j, ctx, ctx->warmStart);
		if (ctx->warmStart)
		{
			warmStartJoint(w, j);
		}
	}
	for (int iter = 0; iter < iterations; ++iter)
	{
		FOR_JOINTS(w, j) { solveJoint_Baumgarte(w, j, ctx, h, inv_h, true); }
		solveContacts_PGS_Baumgarte(w, inv_h);
	}
	integratePositions(w, h);
	finalizePositions(w);
	storeContactImpulses(w, 0.0f, false);
}

static void solve_PGS_NGS(World* w, const Context* ctx) // solve_pgs_ngs.c:149-255
{
	int velocityIterations = ctx->iterations;
	int positionIterations = ctx->extraIterations;
	float h = ctx->dt;
	integrateVelocities(w, ctx, h);
	prepareContacts(w, PREP_PGS, ctx->warmStart, 0.0f, 0.0f);
	if (ctx->warmStart)
	{
		warmStartContacts(w, false);
	}
	FOR_JOINTS(w, j)
	{
		prepareJoint(w, j, ctx, ctx->warmStart);
		if (ctx->warmStart)
		{
			warmStartJoint(w, j);
		}
	}
	for (int iter = 0; iter < velocityIterations; ++iter)
	{
		FOR_JOINTS(w, j) { solveJoint(w, j, ctx, h); }
		solveContacts_PGS(w);
	}
	integratePositions(w, h);
	storeContactImpulses(w, 0.0f, false); // before the position sweeps: solve_pgs_ngs.c:232
	for (int iter = 0; iter < positionIterations; ++iter)
	{
		FOR_JOINTS(w, j) { solveJointPosition(w, j); }
		solveContact_NGS(w);
	}
	finalizePositions(w);
}

static void solve_TGS_NGS(World* w, const Context* ctx) // solve_tgs_ngs.c:207-317
{
	prepareContacts(w, PREP_TGS, ctx->warmStart, 0.0f, 0.0f);
	FOR_JOINTS(w, j) { prepareJoint(w, j, ctx, ctx->warmStart); }
	int substepCount = ctx->iterations;
	float h = ctx->h, inv_h = ctx->inv_h;
	for (int substep = 0; substep < substepCount; ++substep)
	{
		integrateVelocities(w, ctx, h);
		if (ctx->warmStart)
		{
			FOR_JOINTS(w, j) { warmStartJoint(w, j); }
			warmStartContacts(w, false);
		}
		FOR_JOINTS(w, j) { solveJoint(w, j, ctx, h); }
		solveContacts_TGS(w, inv_h);
		integratePositions(w, h);
		FOR_JOINTS(w, j) { solveJointPosition(w, j); }
		solveContact_NGS(w);
	}
	finalizePositions(w);
	storeContactImpulses(w, 0.0f, false);
}

static void solve_TGS_Sticky(World* w, const Context* ctx) // solve_tgs_sticky.c:313-417
{
	FOR_JOINTS(w, j) { prepareJoint(w, j, ctx, false); }
	prepareContacts(w, PREP_STICKY, false, 0.0f, 0.0f);
	prepareStickyFriction(w);
	int substepCount = ctx->iterations;
	float h = ctx->h, inv_h = ctx->inv_h;
	for (int substep = 0; substep < substepCount; ++substep)
	{
		integrateVelocities(w, ctx, h);
		FOR_JOINTS(w, j) { solveJoint_Baumgarte(w, j, ctx, h, inv_h, true); }
		solveContacts_TGS_Sticky(w, inv_h, true);
		integratePositions(w, h);
	}
	finalizePositions(w);
	int relaxCount = ctx->extraIterations;
	for (int iter = 0; iter < relaxCount; ++iter)
	{
		FOR_JOINTS(w, j) { solveJoint_Baumgarte(w, j, ctx, h, inv_h, false); }
		solveContacts_TGS_Sticky(w, inv_h, false);
	}
	storeContactImpulses(w, 0.0f, false);
}

static void solve_XPBD(World* w, const Context* ctx) // solve_xpbd.c:342-530
{
	int substepCount = ctx->iterations;
	if (substepCount == 0 || ctx->dt == 0.0f)
	{
		// early out BEFORE the gather loop: constraintIndex is not written (solve_xpbd.c:344-353)
		return;
	}
	prepareContacts(w, PREP_XPBD, false, 0.0f, 0.0f);
	FOR_JOINTS(w, j) { prepareJoint_XPBD(w, j, ctx); }

	float h = ctx->dt / substepCount;
	float inv_h = 1.0f / h;
	V2 gravity = ctx->gravity;

	for (int substep = 0; substep < substepCount; ++substep)
	{
		for (int i = 0; i < w->bodyCapacity; ++i) // solve_xpbd.c:411-449
		{
			Body* body = w->bodies + i;
			if (body->type == S2AMD_BODY_FREE || body->type == S2AMD_BODY_STATIC)
			{
				continue;
			}
			float invMass = body->invMass, invI = body->invI;
			V2 v = body->linearVelocity;
			float wv = body->angularVelocity;
			v = add(v, mulSV(h * invMass, mulAdd(body->force, body->mass * body->gravityScale, gravity)));
			wv = wv + h * invI * body->torque;
			v = mulSV(1.0f / (1.0f + h * body->linearDamping), v);
			wv *= 1.0f / (1.0f + h * body->angularDamping);
			body->linearVelocity = v;
			body->angularVelocity = wv;
			body->rot0 = body->rot;
			body->deltaPosition0 = body->deltaPosition;
			body->deltaPosition = mulAdd(body->deltaPosition, h, v);
			body->rot = integrateRot(body->rot, h * wv);
		}

		FOR_JOINTS(w, j) { solveJoint_XPBD(w, j, ctx); }
		solveContactPositions_XPBD(w, h);

		for (int i = 0; i < w->bodyCapacity; ++i) // solve_xpbd.c:465-489
		{
			Body* body = w->bodies + i;
			if (body->type != S2AMD_BODY_DYNAMIC)
			{
				continue;
			}
			body->linearVelocity = mulSV(inv_h, sub(body->deltaPosition, body->deltaPosition0));
			body->angularVelocity = computeAngularVelocity(body->rot0, body->rot, inv_h);
		}

		solveContactVelocities_XPBD(w, h);
	}

	for (int i = 0; i < w->bodyCapacity; ++i) // solve_xpbd.c:496-512 (dynamic bodies only)
	{
		Body* body = w->bodies + i;
		if (body->type != S2AMD_BODY_DYNAMIC)
		{
			continue;
		}
		body->position = add(body->position, body->deltaPosition);
		body->deltaPosition = v2(0.0f, 0.0f);
	}
	storeContactImpulses(w, inv_h, true);
}

// ---------------------------------------------------------------------------------------------
// PGS_NGS_Block: src/solve_pgs_ngs_block.c
// ---------------------------------------------------------------------------------------------
typedef struct BPoint // :87-99
{
	V2 rA, rB, localAnchorA, localAnchorB;
	float separation, adjustedSeparation, normalImpulse, tangentImpulse, normalMass, tangentMass, velocityBias;
} BPoint;

typedef struct BConstraint // :101-110
{
	int contact, indexA, indexB;
	BPoint points[2];
	V2 normal;
	M22 normalMass, K;
	float friction;
	int pointCount;
} BConstraint;

static void blockCreate(World* w, const Context* ctx, BConstraint* constraints) // s2CreateContactSolver :135-322
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		BConstraint* constraint = constraints + i;
		constraint->contact = w->constraints[i].contact;
		const s2amdContact* contact = w->contacts + constraint->contact;
		int pointCount = contact->pointCount;
		int indexA = contact->bodyA, indexB = contact->bodyB;
		Body* bodyA = bodies + indexA;
		Body* bodyB = bodies + indexB;
		constraint->indexA = indexA;
		constraint->indexB = indexB;
		constraint->normal = v2(contact->normal[0], contact->normal[1]);
		constraint->friction = contact->friction;
		constraint->pointCount = pointCount;
		memset(&constraint->K, 0, sizeof(M22));
		memset(&constraint->normalMass, 0, sizeof(M22));

		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		Rot qA = bodyA->rot, qB = bodyB->rot;
		V2 normal = constraint->normal;

		for (int j = 0; j < pointCount; ++j)
		{
			const s2amdManifoldPoint* mp = contact->points + j;
			BPoint* cp = constraint->points + j;
			if (ctx->warmStart)
			{
				cp->normalImpulse = mp->normalImpulse;
				cp->tangentImpulse = mp->tangentImpulse;
			}
			else
			{
				cp->normalImpulse = 0.0f;
				cp->tangentImpulse = 0.0f;
			}
			V2 localAnchorA = sub(v2(mp->localAnchorA[0], mp->localAnchorA[1]), bodyA->localCenter);
			V2 localAnchorB = sub(v2(mp->localAnchorB[0], mp->localAnchorB[1]), bodyB->localCenter);
			V2 rA = rotate(qA, localAnchorA);
			V2 rB = rotate(qB, localAnchorB);
			cp->rA = rA;
			cp->rB = rB;
			float rnA = cross(cp->rA, normal);
			float rnB = cross(cp->rB, normal);
			float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
			cp->normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
			V2 tangent = crossVS(normal, 1.0f);
			float rtA = cross(cp->rA, tangent);
			float rtB = cross(cp->rB, tangent);
			float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
			cp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
			cp->velocityBias = -MAX_(0.0f, mp->separation * ctx->inv_dt);
			cp->localAnchorA = localAnchorA;
			cp->localAnchorB = localAnchorB;
			cp->separation = mp->separation;
			cp->adjustedSeparation = mp->separation - dot(sub(rB, rA), normal);
		}

		if (constraint->pointCount == 2)
		{
			BPoint* cp1 = constraint->points + 0;
			BPoint* cp2 = constraint->points + 1;
			float rn1A = cross(cp1->rA, normal);
			float rn1B = cross(cp1->rB, normal);
			float rn2A = cross(cp2->rA, normal);
			float rn2B = cross(cp2->rB, normal);
			float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
			float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
			float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
			const float k_maxConditionNumber = 1000.0f;
			if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12))
			{
				constraint->K.cx = v2(k11, k12);
				constraint->K.cy = v2(k12, k22);
				constraint->normalMass = inverse22(constraint->K);
			}
			else
			{
				constraint->pointCount = 1;
			}
		}
	}

	// warm start, always applied (zero impulses when !warmStart): :279-319
	for (int i = 0; i < w->constraintCount; ++i)
	{
		BConstraint* constraint = constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;
		V2 normal = constraint->normal;
		V2 tangent = crossVS(normal, 1.0f);
		for (int j = 0; j < pointCount; ++j)
		{
			BPoint* cp = constraint->points + j;
			V2 P = add(mulSV(cp->normalImpulse, normal), mulSV(cp->tangentImpulse, tangent));
			wA -= iA * cross(cp->rA, P);
			vA = mulAdd(vA, -mA, P);
			wB += iB * cross(cp->rB, P);
			vB = mulAdd(vB, mB, P);
		}
		bodyA->linearVelocity = vA;
		bodyA->angularVelocity = wA;
		bodyB->linearVelocity = vB;
		bodyB->angularVelocity = wB;
	}
}

// applies the pair of normal impulses d to both bodies: the block repeated four times at :520-530 etc.
#define BLOCK_APPLY_VELOCITY(d)                                                                                                  \
	{                                                                                                                            \
		V2 P1 = mulSV((d).x, normal);                                                                                            \
		V2 P2 = mulSV((d).y, normal);                                                                                            \
		vA = mulSub(vA, mA, add(P1, P2));                                                                                        \
		wA -= iA * (cross(cp1->rA, P1) + cross(cp2->rA, P2));                                                                    \
		vB = mulAdd(vB, mB, add(P1, P2));                                                                                        \
		wB += iB * (cross(cp1->rB, P1) + cross(cp2->rB, P2));                                                                    \
	}

static void blockSolveVelocity(World* w, BConstraint* constraints) // s2BlockSolveVelocity :329-658
{
	Body* bodies = w->bodies;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		BConstraint* constraint = constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		V2 vA = bodyA->linearVelocity;
		float wA = bodyA->angularVelocity;
		V2 vB = bodyB->linearVelocity;
		float wB = bodyB->angularVelocity;
		V2 normal = constraint->normal;
		V2 tangent = crossVS(normal, 1.0f);
		float friction = constraint->friction;

		for (int j = 0; j < pointCount; ++j)
		{
			BPoint* cp = constraint->points + j;
			V2 vrB = add(vB, crossSV(wB, cp->rB));
			V2 vrA = add(vA, crossSV(wA, cp->rA));
			V2 dv = sub(vrB, vrA);
			float vt = dot(dv, tangent);
			float lambda = cp->tangentMass * (-vt);
			float maxFriction = friction * cp->normalImpulse;
			float newImpulse = CLAMP_(cp->tangentImpulse + lambda, -maxFriction, maxFriction);
			lambda = newImpulse - cp->tangentImpulse;
			cp->tangentImpulse = newImpulse;
			V2 P = mulSV(lambda, tangent);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(cp->rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(cp->rB, P);
		}

		if (pointCount == 1)
		{
			BPoint* cp = constraint->points + 0;
			V2 vrB = add(vB, crossSV(wB, cp->rB));
			V2 vrA = add(vA, crossSV(wA, cp->rA));
			V2 dv = sub(vrB, vrA);
			float vn = dot(dv, normal);
			float lambda = -cp->normalMass * (vn - cp->velocityBias);
			float newImpulse = MAX_(cp->normalImpulse + lambda, 0.0f);
			lambda = newImpulse - cp->normalImpulse;
			cp->normalImpulse = newImpulse;
			V2 P = mulSV(lambda, normal);
			vA = mulSub(vA, mA, P);
			wA -= iA * cross(cp->rA, P);
			vB = mulAdd(vB, mB, P);
			wB += iB * cross(cp->rB, P);
		}
		else
		{
			BPoint* cp1 = constraint->points + 0;
			BPoint* cp2 = constraint->points + 1;
			V2 a = {cp1->normalImpulse, cp2->normalImpulse};
			V2 vrA, vrB;
			vrA = add(vA, crossSV(wA, cp1->rA));
			vrB = add(vB, crossSV(wB, cp1->rB));
			V2 dv1 = sub(vrB, vrA);
			vrA = add(vA, crossSV(wA, cp2->rA));
			vrB = add(vB, crossSV(wB, cp2->rB));
			V2 dv2 = sub(vrB, vrA);
			float vn1 = dot(dv1, normal);
			float vn2 = dot(dv2, normal);
			V2 b = {vn1 - cp1->velocityBias, vn2 - cp2->velocityBias};
			b = sub(b, mulMV(constraint->K, a));

			for (;;)
			{
				// case 1: both active (:493-531)
				V2 x = neg(mulMV(constraint->normalMass, b));
				if (x.x >= 0.0f && x.y >= 0.0f)
				{
					V2 d = sub(x, a);
					BLOCK_APPLY_VELOCITY(d);
					cp1->normalImpulse = x.x;
					cp2->normalImpulse = x.y;
					break;
				}
				// case 2: x2 = 0 (:546-577)
				x.x = -cp1->normalMass * b.x;
				x.y = 0.0f;
				vn1 = 0.0f;
				vn2 = constraint->K.cx.y * x.x + b.y;
				if (x.x >= 0.0f && vn2 >= 0.0f)
				{
					V2 d = sub(x, a);
					BLOCK_APPLY_VELOCITY(d);
					cp1->normalImpulse = x.x;
					cp2->normalImpulse = x.y;
					break;
				}
				// case 3: x1 = 0 (:585-616)
				x.x = 0.0f;
				x.y = -cp2->normalMass * b.y;
				vn1 = constraint->K.cy.x * x.y + b.x;
				vn2 = 0.0f;
				if (x.y >= 0.0f && vn1 >= 0.0f)
				{
					V2 d = sub(x, a);
					BLOCK_APPLY_VELOCITY(d);
					cp1->normalImpulse = x.x;
					cp2->normalImpulse = x.y;
					break;
				}
				// case 4: both inactive (:622-647)
				x.x = 0.0f;
				x.y = 0.0f;
				vn1 = b.x;
				vn2 = b.y;
				if (vn1 >= 0.0f && vn2 >= 0.0f)
				{
					V2 d = sub(x, a);
					BLOCK_APPLY_VELOCITY(d);
					cp1->normalImpulse = x.x;
					cp2->normalImpulse = x.y;
					break;
				}
				break; // no solution, give up (:649-650)
			}
		}

		bodyA->linearVelocity = vA;
		bodyA->angularVelocity = wA;
		bodyB->linearVelocity = vB;
		bodyB->angularVelocity = wB;
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

static void blockSolvePosition(World* w, BConstraint* constraints) // s2BlockSolvePosition :679-890
{
	Body* bodies = w->bodies;
	float slop = K_LINEAR_SLOP;
	for (int i = 0; i < w->constraintCount; ++i)
	{
		BConstraint* constraint = constraints + i;
		Body* bodyA = bodies + constraint->indexA;
		Body* bodyB = bodies + constraint->indexB;
		float mA = bodyA->invMass, iA = bodyA->invI;
		float mB = bodyB->invMass, iB = bodyB->invI;
		int pointCount = constraint->pointCount;
		V2 dcA = bodyA->deltaPosition;
		Rot qA = bodyA->rot;
		V2 dcB = bodyB->deltaPosition;
		Rot qB = bodyB->rot;
		V2 normal = constraint->normal;
		bool degenerate = pointCount != 2;

		if (pointCount == 2)
		{
			BPoint* cp1 = constraint->points + 0;
			BPoint* cp2 = constraint->points + 1;
			V2 rA1 = rotate(qA, cp1->localAnchorA);
			V2 rB1 = rotate(qB, cp1->localAnchorB);
			V2 rA2 = rotate(qA, cp2->localAnchorA);
			V2 rB2 = rotate(qB, cp2->localAnchorB);
			V2 dc = sub(dcB, dcA);
			V2 d1 = add(dc, sub(rB1, rA1));
			float separation1 = dot(d1, normal) + cp1->adjustedSeparation;
			V2 d2 = add(dc, sub(rB2, rA2));
			float separation2 = dot(d2, normal) + cp2->adjustedSeparation;
			float C1 = CLAMP_(K_BAUMGARTE * (separation1 + slop), -K_MAX_LINEAR_CORRECTION, 0.0f);
			float C2 = CLAMP_(K_BAUMGARTE * (separation2 + slop), -K_MAX_LINEAR_CORRECTION, 0.0f);
			V2 b = {C1, C2};
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
				M22 K, invK;
				K.cx = v2(k11, k12);
				K.cy = v2(k12, k22);
				invK = inverse22(K);
				for (;;)
				{
					V2 x = neg(mulMV(invK, b));
					if (x.x >= 0.0f && x.y >= 0.0f)
					{
						BLOCK_APPLY_POSITION(x);
						break;
					}
					x.x = -b.x / k11;
					x.y = 0.0f;
					float vn2 = K.cx.y * x.x + b.y;
					if (x.x >= 0.0f && vn2 >= 0.0f)
					{
						BLOCK_APPLY_POSITION(x);
						break;
					}
					x.x = 0.0f;
					x.y = -b.y / k22;
					float vn1 = K.cy.x * x.y + b.x;
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
				degenerate = true; // goto manifold_degenerate (:747-751)
			}
		}

		if (degenerate)
		{
			for (int j = 0; j < pointCount; ++j) // :851-876
			{
				BPoint* cp = constraint->points + j;
				V2 rA = rotate(qA, cp->localAnchorA);
				V2 rB = rotate(qB, cp->localAnchorB);
				V2 d = add(sub(dcB, dcA), sub(rB, rA));
				float separation = dot(d, normal) + cp->adjustedSeparation;
				float C = CLAMP_(K_BAUMGARTE * (separation + slop), -K_MAX_LINEAR_CORRECTION, 0.0f);
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

		bodyA->deltaPosition = dcA;
		bodyA->rot = qA;
		bodyB->deltaPosition = dcB;
		bodyB->rot = qB;
	}
}

static void solve_PGS_NGS_Block(World* w, const Context* ctx) // s2Solve_PGS_NGS_Block :892-963
{
	float h = ctx->dt;
	integrateVelocities(w, ctx, h);
	BConstraint* constraints = (BConstraint*)calloc((size_t)(w->constraintCount > 0 ? w->constraintCount : 1), sizeof(BConstraint));
	blockCreate(w, ctx, constraints);
	FOR_JOINTS(w, j)
	{
		prepareJoint(w, j, ctx, ctx->warmStart);
		if (ctx->warmStart)
		{
			warmStartJoint(w, j);
		}
	}
	for (int i = 0; i < ctx->iterations; ++i)
	{
		FOR_JOINTS(w, j) { solveJoint(w, j, ctx, h); }
		blockSolveVelocity(w, constraints);
	}
	for (int i = 0; i < w->constraintCount; ++i) // s2ContactSolver_StoreImpulses :660-677
	{
		BConstraint* constraint = constraints + i;
		s2amdContact* manifold = w->contacts + constraint->contact;
		for (int j = 0; j < constraint->pointCount; ++j)
		{
			manifold->points[j].normalImpulse = constraint->points[j].normalImpulse;
			manifold->points[j].tangentImpulse = constraint->points[j].tangentImpulse;
		}
	}
	integratePositions(w, h);
	for (int i = 0; i < ctx->extraIterations; ++i)
	{
		blockSolvePosition(w, constraints); // contacts BEFORE joints here (:945-957)
		FOR_JOINTS(w, j) { solveJointPosition(w, j); }
	}
	finalizePositions(w);
	free(constraints);
}

// ---------------------------------------------------------------------------------------------
// entry point
// ---------------------------------------------------------------------------------------------

ORACLE_API int s2oracle_api_version(void)
{
	return S2AMD_API_VERSION;
}

// == s2Solve_<params->solverType>(world, context) on wire arrays.
// contactOrder: NULL, or the contact-array indices of the active constraints in sweep order
//               (must be a permutation of the slots with pointCount > 0; count given).
// jointOrder:   NULL, or the joint-array indices of the live joints in sweep order.
ORACLE_API int s2oracle_solve(const s2amdStepParams* params, s2amdBody* bodies, int32_t bodyCapacity, s2amdContact* contacts,
							  int32_t contactCapacity, s2amdJoint* joints, int32_t jointCapacity, const int32_t* contactOrder,
							  int32_t contactOrderCount, const int32_t* jointOrder, int32_t jointOrderCount)
{
	if (params == NULL || bodyCapacity < 0 || contactCapacity < 0 || jointCapacity < 0)
	{
		return S2AMD_E_INVALID;
	}
	if (params->solverType < 0 || params->solverType >= s2amd_solverTypeCount)
	{
		return S2AMD_E_INVALID;
	}

	// step context: src/world.c:170-202
	Context ctx;
	memset(&ctx, 0, sizeof(ctx));
	ctx.dt = params->dt;
	ctx.iterations = params->velIters;
	ctx.extraIterations = params->posIters;
	ctx.warmStart = params->warmStart != 0;
	ctx.inv_dt = params->dt > 0.0f ? 1.0f / params->dt : 0.0f;
	int type = params->solverType;
	if (type == s2amd_solverXPBD || type == s2amd_solverTGS_Soft || type == s2amd_solverTGS_Sticky || type == s2amd_solverTGS_NGS ||
		type == s2amd_solverSoftStep)
	{
		ctx.h = ctx.dt / ctx.iterations;
		ctx.inv_h = ctx.inv_dt * ctx.iterations;
	}
	else
	{
		ctx.h = ctx.dt;
		ctx.inv_h = ctx.inv_dt;
	}
	ctx.gravity = v2(params->gravity[0], params->gravity[1]);

	World w;
	memset(&w, 0, sizeof(w));
	w.bodyCapacity = bodyCapacity;
	w.bodies = (Body*)calloc((size_t)(bodyCapacity > 0 ? bodyCapacity : 1), sizeof(Body));
	for (int i = 0; i < bodyCapacity; ++i)
	{
		const s2amdBody* s = bodies + i;
		Body* b = w.bodies + i;
		b->type = s->type;
		b->position = v2(s->position[0], s->position[1]);
		b->rot.s = s->rot[0], b->rot.c = s->rot[1];
		b->rot0 = b->rot;
		b->linearVelocity = v2(s->linearVelocity[0], s->linearVelocity[1]);
		b->angularVelocity = s->angularVelocity;
		b->deltaPosition = v2(s->deltaPosition[0], s->deltaPosition[1]);
		b->localCenter = v2(s->localCenter[0], s->localCenter[1]);
		b->force = v2(s->force[0], s->force[1]);
		b->torque = s->torque;
		b->mass = s->mass, b->invMass = s->invMass, b->I = s->I, b->invI = s->invI;
		b->linearDamping = s->linearDamping, b->angularDamping = s->angularDamping, b->gravityScale = s->gravityScale;
	}

	w.contacts = contacts;
	w.contactCapacity = contactCapacity;

	// constraintIndex is written by the gather loop of nine drivers; s2Solve_PGS_NGS_Block has no such
	// loop (s2CreateContactSolver, solve_pgs_ngs_block.c:135-277 never touches it) and XPBD returns
	// before it when there is nothing to do (solve_xpbd.c:344-353).
	bool xpbdEarlyOut = (type == s2amd_solverXPBD && (ctx.iterations == 0 || ctx.dt == 0.0f)) || type == s2amd_solverPGS_NGS_Block;

	// gather: e.g. solve_tgs_soft.c:162-179.  constraintIndex is always the pool-order index.
	w.constraints = (Constraint*)calloc((size_t)(contactCapacity > 0 ? contactCapacity : 1), sizeof(Constraint));
	int gatherCount = 0;
	for (int i = 0; i < contactCapacity; ++i)
	{
		if (contacts[i].pointCount == 0)
		{
			if (!xpbdEarlyOut)
			{
				contacts[i].constraintIndex = -1;
			}
			continue;
		}
		if (!xpbdEarlyOut)
		{
			contacts[i].constraintIndex = gatherCount;
		}
		w.constraints[gatherCount].contact = i;
		gatherCount += 1;
	}
	w.constraintCount = gatherCount;
	if (contactOrder != NULL)
	{
		if (contactOrderCount != gatherCount)
		{
			free(w.bodies);
			free(w.constraints);
			return S2AMD_E_INVALID;
		}
		for (int k = 0; k < gatherCount; ++k)
		{
			int ci = contactOrder[k];
			if (ci < 0 || ci >= contactCapacity || contacts[ci].pointCount == 0)
			{
				free(w.bodies);
				free(w.constraints);
				return S2AMD_E_INVALID;
			}
			w.constraints[k].contact = ci;
		}
	}

	// live joints in pool order (or the given order)
	w.joints = (Joint*)calloc((size_t)(jointCapacity > 0 ? jointCapacity : 1), sizeof(Joint));
	int jc = 0;
	for (int k = 0; k < (jointOrder ? jointOrderCount : jointCapacity); ++k)
	{
		int i = jointOrder ? jointOrder[k] : k;
		if (i < 0 || i >= jointCapacity || joints[i].type == S2AMD_JOINT_FREE)
		{
			if (jointOrder)
			{
				free(w.bodies);
				free(w.constraints);
				free(w.joints);
				return S2AMD_E_INVALID;
			}
			continue;
		}
		const s2amdJoint* s = joints + i;
		Joint* j = w.joints + jc++;
		j->wire = i;
		j->type = s->type;
		j->indexA = s->bodyA, j->indexB = s->bodyB;
		j->localOriginAnchorA = v2(s->localOriginAnchorA[0], s->localOriginAnchorA[1]);
		j->localOriginAnchorB = v2(s->localOriginAnchorB[0], s->localOriginAnchorB[1]);
		j->impulse = v2(s->impulse[0], s->impulse[1]);
		j->motorImpulse = s->motorImpulse, j->lowerImpulse = s->lowerImpulse, j->upperImpulse = s->upperImpulse;
		j->enableMotor = s->enableMotor != 0, j->enableLimit = s->enableLimit != 0;
		j->maxMotorTorque = s->maxMotorTorque, j->motorSpeed = s->motorSpeed;
		j->referenceAngle = s->referenceAngle, j->lowerAngle = s->lowerAngle, j->upperAngle = s->upperAngle;
		j->hertz = s->hertz, j->dampingRatio = s->dampingRatio;
		j->targetA = v2(s->targetA[0], s->targetA[1]);
	}
	w.jointCount = jc;

	switch (type)
	{
		case s2amd_solverJacobi:
			solve_Jacobi_or_PGS_Soft(&w, &ctx, true);
			break;
		case s2amd_solverPGS:
			solve_PGS(&w, &ctx);
			break;
		case s2amd_solverPGS_NGS:
			solve_PGS_NGS(&w, &ctx);
			break;
		case s2amd_solverPGS_NGS_Block:
			solve_PGS_NGS_Block(&w, &ctx);
			break;
		case s2amd_solverPGS_Soft:
			solve_Jacobi_or_PGS_Soft(&w, &ctx, false);
			break;
		case s2amd_solverSoftStep:
			solve_TGS_Soft(&w, &ctx, true);
			break;
		case s2amd_solverTGS_Sticky:
			solve_TGS_Sticky(&w, &ctx);
			break;
		case s2amd_solverTGS_Soft:
			solve_TGS_Soft(&w, &ctx, false);
			break;
		case s2amd_solverTGS_NGS:
			solve_TGS_NGS(&w, &ctx);
			break;
		case s2amd_solverXPBD:
			solve_XPBD(&w, &ctx);
			break;
		default:
			break;
	}

	for (int i = 0; i < bodyCapacity; ++i)
	{
		s2amdBody* s = bodies + i;
		const Body* b = w.bodies + i;
		if (b->type == S2AMD_BODY_FREE)
		{
			continue;
		}
		s->position[0] = b->position.x, s->position[1] = b->position.y;
		s->rot[0] = b->rot.s, s->rot[1] = b->rot.c;
		s->linearVelocity[0] = b->linearVelocity.x, s->linearVelocity[1] = b->linearVelocity.y;
		s->angularVelocity = b->angularVelocity;
		s->deltaPosition[0] = b->deltaPosition.x, s->deltaPosition[1] = b->deltaPosition.y;
	}
	for (int k = 0; k < w.jointCount; ++k)
	{
		const Joint* j = w.joints + k;
		s2amdJoint* s = joints + j->wire;
		s->impulse[0] = j->impulse.x, s->impulse[1] = j->impulse.y;
		s->motorImpulse = j->motorImpulse;
		if (j->type == S2AMD_JOINT_REVOLUTE)
		{
			s->lowerImpulse = j->lowerImpulse;
			s->upperImpulse = j->upperImpulse;
		}
	}

	free(w.bodies);
	free(w.constraints);
	free(w.joints);
	return S2AMD_OK;
}

// ---------------------------------------------------------------------------------------------
// Stages either side of the solver: Stage 4 AABB refit and broad-phase pair discovery
// ---------------------------------------------------------------------------------------------
#define K_SPECULATIVE_DISTANCE (4.0f * K_LINEAR_SLOP) // constants.h:8
#define K_AABB_MARGIN 0.1f							   // constants.h:9

typedef struct Xf
{
	V2 p;
	Rot q;
} Xf;

static inline V2 transformPoint(Xf xf, V2 p) // math.h:350-356
{
	float x = (xf.q.c * p.x - xf.q.s * p.y) + xf.p.x;
	float y = (xf.q.s * p.x + xf.q.c * p.y) + xf.p.y;
	return v2(x, y);
}
static inline V2 vmin(V2 a, V2 b) { return v2(MIN_(a.x, b.x), MIN_(a.y, b.y)); } // math.h:144-150
static inline V2 vmax(V2 a, V2 b) { return v2(MAX_(a.x, b.x), MAX_(a.y, b.y)); } // math.h:153-159

// s2Shape_ComputeAABB (src/shape.c) -> s2Compute{Capsule,Circle,Polygon,Segment}AABB (src/geometry.c:288-339)
static void shapeAABB(const s2amdShape* sh, Xf xf, float out[4])
{
	V2 lower, upper;
	V2 v0 = v2(sh->vertices[0][0], sh->vertices[0][1]);
	V2 v1 = v2(sh->vertices[1][0], sh->vertices[1][1]);
	switch (sh->type)
	{
		case S2AMD_SHAPE_CIRCLE:
		{
			V2 p = transformPoint(xf, v0);
			float r = sh->radius;
			lower = v2(p.x - r, p.y - r);
			upper = v2(p.x + r, p.y + r);
			break;
		}
		case S2AMD_SHAPE_CAPSULE:
		{
			V2 a = transformPoint(xf, v0), b = transformPoint(xf, v1);
			V2 r = v2(sh->radius, sh->radius);
			lower = sub(vmin(a, b), r);
			upper = add(vmax(a, b), r);
			break;
		}
		case S2AMD_SHAPE_POLYGON:
		{
			lower = transformPoint(xf, v0);
			upper = lower;
			for (int i = 1; i < sh->count; ++i)
			{
				V2 v = transformPoint(xf, v2(sh->vertices[i][0], sh->vertices[i][1]));
				lower = vmin(lower, v);
				upper = vmax(upper, v);
			}
			V2 r = v2(sh->radius, sh->radius);
			lower = sub(lower, r);
			upper = add(upper, r);
			break;
		}
		case S2AMD_SHAPE_SEGMENT:
		{
			V2 a = transformPoint(xf, v0), b = transformPoint(xf, v1);
			lower = vmin(a, b);
			upper = vmax(a, b);
			break;
		}
		default:
			lower = xf.p;
			upper = xf.p;
			break;
	}
	out[0] = lower.x, out[1] = lower.y, out[2] = upper.x, out[3] = upper.y;
}

// Stage 4 of s2World_Step: src/world.c:259-301
ORACLE_API int s2oracle_refit_shapes(const s2amdBody* bodies, int32_t bodyCapacity, s2amdShape* shapes, int32_t shapeCapacity, float* origins)
{
	for (int i = 0; i < bodyCapacity; ++i)
	{
		const s2amdBody* b = bodies + i;
		if (b->type == S2AMD_BODY_FREE || b->type == S2AMD_BODY_STATIC)
		{
			continue;
		}
		Rot q = {b->rot[0], b->rot[1]};
		V2 o = sub(v2(b->position[0], b->position[1]), rotate(q, v2(b->localCenter[0], b->localCenter[1])));
		origins[2 * i] = o.x, origins[2 * i + 1] = o.y;
	}
	for (int si = 0; si < shapeCapacity; ++si)
	{
		s2amdShape* sh = shapes + si;
		if (sh->type == S2AMD_SHAPE_FREE || sh->body < 0 || sh->body >= bodyCapacity)
		{
			continue;
		}
		const s2amdBody* b = bodies + sh->body;
		if (b->type == S2AMD_BODY_FREE || b->type == S2AMD_BODY_STATIC)
		{
			continue;
		}
		Xf xf;
		xf.p = v2(origins[2 * sh->body], origins[2 * sh->body + 1]);
		xf.q.s = b->rot[0], xf.q.c = b->rot[1];
		shapeAABB(sh, xf, sh->aabb);
		sh->aabb[0] -= K_SPECULATIVE_DISTANCE;
		sh->aabb[1] -= K_SPECULATIVE_DISTANCE;
		sh->aabb[2] += K_SPECULATIVE_DISTANCE;
		sh->aabb[3] += K_SPECULATIVE_DISTANCE;
		// s2AABB_Contains(fatAABB, aabb): include/solver2d/aabb.h
		bool contains = sh->fatAABB[0] <= sh->aabb[0] && sh->fatAABB[1] <= sh->aabb[1] && sh->aabb[2] <= sh->fatAABB[2] &&
						sh->aabb[3] <= sh->fatAABB[3];
		sh->enlarged = 0;
		if (contains == false)
		{
			sh->fatAABB[0] = sh->aabb[0] - K_AABB_MARGIN;
			sh->fatAABB[1] = sh->aabb[1] - K_AABB_MARGIN;
			sh->fatAABB[2] = sh->aabb[2] + K_AABB_MARGIN;
			sh->fatAABB[3] = sh->aabb[3] + K_AABB_MARGIN;
			sh->enlarged = 1;
		}
	}
	return S2AMD_OK;
}

static bool aabbOverlaps(const float a[4], const float b[4]) // s2AABB_Overlaps, include/solver2d/aabb.h
{
	float d1x = b[0] - a[2], d1y = b[1] - a[3];
	float d2x = a[0] - b[2], d2y = a[1] - b[3];
	if (d1x > 0.0f || d1y > 0.0f)
	{
		return false;
	}
	if (d2x > 0.0f || d2y > 0.0f)
	{
		return false;
	}
	return true;
}

static bool shouldShapesCollide(const s2amdShape* a, const s2amdShape* b) // src/contact.h:68-78
{
	if (a->groupIndex == b->groupIndex && a->groupIndex != 0)
	{
		return a->groupIndex > 0;
	}
	return (a->maskBits & b->categoryBits) != 0 && (a->categoryBits & b->maskBits) != 0;
}

static int cmpPair(const void* x, const void* y)
{
	const int32_t* a = (const int32_t*)x;
	const int32_t* b = (const int32_t*)y;
	if (a[0] != b[0])
	{
		return a[0] < b[0] ? -1 : 1;
	}
	return a[1] < b[1] ? -1 : (a[1] > b[1] ? 1 : 0);
}

// The pair discovery of s2FindPairs / s2PairQueryCallback (src/broad_phase.c:166-307) by brute force:
// the tree query is "every proxy of that tree whose fat AABB overlaps the query's fat AABB".
ORACLE_API int s2oracle_find_pairs(const s2amdBody* bodies, int32_t bodyCapacity, const s2amdShape* shapes, int32_t shapeCapacity,
								   const uint8_t* moved, const int32_t* existingPairs, int32_t existingPairCount, const s2amdJoint* joints,
								   int32_t jointCapacity, int32_t* outPairs, int32_t pairCapacity, int32_t* pairCount)
{
	(void)bodyCapacity;
	int n = 0;
	for (int pi = 0; pi < shapeCapacity; ++pi)
	{
		const s2amdShape* P = shapes + pi;
		if (P->type == S2AMD_SHAPE_FREE || !moved[pi])
		{
			continue;
		}
		int ptype = P->proxyKey & 0xF; // S2_PROXY_TYPE
		if (ptype == S2AMD_BODY_STATIC)
		{
			continue;
		}
		for (int qi = 0; qi < shapeCapacity; ++qi)
		{
			const s2amdShape* Q = shapes + qi;
			if (Q->type == S2AMD_SHAPE_FREE || qi == pi)
			{
				continue; // a proxy cannot pair with itself (:170-174)
			}
			int qtype = Q->proxyKey & 0xF;
			if (ptype == S2AMD_BODY_KINEMATIC && qtype != S2AMD_BODY_DYNAMIC)
			{
				continue; // a kinematic proxy only queries the dynamic tree (:293-297)
			}
			if (!aabbOverlaps(P->fatAABB, Q->fatAABB))
			{
				continue;
			}
			if (moved[qi] && Q->proxyKey > P->proxyKey)
			{
				continue; // both moving: the lower key reports the pair (:176-181)
			}
			int lo = pi < qi ? pi : qi, hi = pi < qi ? qi : pi;
			bool exists = false;
			for (int e = 0; e < existingPairCount; ++e)
			{
				int ea = existingPairs[2 * e], eb = existingPairs[2 * e + 1];
				if ((ea == lo && eb == hi) || (ea == hi && eb == lo))
				{
					exists = true;
					break;
				}
			}
			if (exists)
			{
				continue; // :183-188
			}
			int ia, ib;
			if (Q->proxyKey < P->proxyKey) // :190-200
			{
				ia = qi, ib = pi;
			}
			else
			{
				ia = pi, ib = qi;
			}
			const s2amdShape* A = shapes + ia;
			const s2amdShape* B = shapes + ib;
			if (A->body == B->body)
			{
				continue;
			}
			if (!shouldShapesCollide(A, B))
			{
				continue;
			}
			// s2ShouldBodiesCollide (src/body.c): any joint between the two bodies blocks the pair
			bool jointed = false;
			for (int j = 0; j < jointCapacity; ++j)
			{
				if (joints[j].type == S2AMD_JOINT_FREE)
				{
					continue;
				}
				if ((joints[j].bodyA == A->body && joints[j].bodyB == B->body) || (joints[j].bodyA == B->body && joints[j].bodyB == A->body))
				{
					jointed = true;
					break;
				}
			}
			if (jointed)
			{
				continue;
			}
			// s2CreateContact (src/contact.c:156-175) + the register table (:137-153): no manifold function
			// for segment vs segment => no contact; a non-primary type order is flipped
			{
				int tA = A->type, tB = B->type;
				if (tA == S2AMD_SHAPE_SEGMENT && tB == S2AMD_SHAPE_SEGMENT)
				{
					continue;
				}
				// primary orders: (circle,circle) (capsule,circle) (capsule,capsule) (polygon,circle)
				// (polygon,capsule) (polygon,polygon) (segment,circle) (segment,capsule) (segment,polygon)
				static const unsigned char primary[4][4] = {
					/* capsule */ {1, 1, 0, 0},
					/* circle  */ {0, 1, 0, 0},
					/* polygon */ {1, 1, 1, 0},
					/* segment */ {1, 1, 1, 0},
				};
				if (!primary[tA][tB])
				{
					int t = ia;
					ia = ib;
					ib = t;
				}
			}
			if (n < pairCapacity)
			{
				outPairs[2 * n] = ia;
				outPairs[2 * n + 1] = ib;
			}
			n += 1;
		}
	}
	*pairCount = n;
	if (n > pairCapacity)
	{
		return S2AMD_E_CAPACITY;
	}
	qsort(outPairs, (size_t)n, 2 * sizeof(int32_t), cmpPair);
	return S2AMD_OK;
}
