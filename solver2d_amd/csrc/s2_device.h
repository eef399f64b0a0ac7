// Device-side vocabulary shared by every kernel file: fp32 vector math with the reference's exact
// operation order, and the SoA "views" the kernels address.
//
// Arithmetic contract: every translation unit is compiled with -ffp-contract=off, IEEE divide and
// sqrt (hipcc's default for HIP), fp32 denormals on.  Each helper below is written operation for
// operation like the inline function it mirrors in include/solver2d/math.h of the reference, so
// a batched sweep on the GPU is bit-identical to a sequential sweep in the same constraint order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define S2_DEV __device__ __forceinline__

// include/solver2d/constants.h:6-22
#define S2_PI 3.14159265359f
#define S2_LINEAR_SLOP 0.005f
#define S2_ANGULAR_SLOP (2.0f / 180.0f * S2_PI)
#define S2_MAX_LINEAR_CORRECTION 0.2f
#define S2_MAX_ANGULAR_CORRECTION (8.0f / 180.0f * S2_PI)
#define S2_BAUMGARTE 0.2f
#define S2_MAX_BAUMGARTE_VELOCITY 4.0f
#define S2_CONTACT_HERTZ 30.0f
#define S2_JOINT_HERTZ 60.0f

// math.h:10-13 (macros with the reference's NaN behaviour)
#define S2_MINF(A, B) ((A) < (B) ? (A) : (B))
#define S2_MAXF(A, B) ((A) > (B) ? (A) : (B))
#define S2_ABSF(A) ((A) > 0.0f ? (A) : -(A))
#define S2_CLAMPF(A, B, C) S2_MINF(S2_MAXF(A, B), C)

struct V2
{
	float x, y;
};
struct Rot
{
	float s, c;
};
struct M22
{
	V2 cx, cy;
};

S2_DEV V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
S2_DEV float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }					  // math.h:47
S2_DEV float cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }				  // math.h:53
S2_DEV V2 crossVS(V2 v, float s) { return v2(s * v.y, -s * v.x); }				  // math.h:60
S2_DEV V2 crossSV(float s, V2 v) { return v2(-s * v.y, s * v.x); }				  // math.h:67
S2_DEV V2 rightPerp(V2 v) { return v2(v.y, -v.x); }								  // math.h:73
S2_DEV V2 add(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }					  // math.h:85
S2_DEV V2 sub(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }					  // math.h:91
S2_DEV V2 neg(V2 a) { return v2(-a.x, -a.y); }									  // math.h:97
S2_DEV V2 mulSV(float s, V2 v) { return v2(s * v.x, s * v.y); }					  // math.h:115
S2_DEV V2 mulAdd(V2 a, float s, V2 b) { return v2(a.x + s * b.x, a.y + s * b.y); } // math.h:121
S2_DEV V2 mulSub(V2 a, float s, V2 b) { return v2(a.x - s * b.x, a.y - s * b.y); } // math.h:127
S2_DEV float length(V2 v) { return sqrtf(v.x * v.x + v.y * v.y); }				  // math.h:171

S2_DEV V2 normalize(V2 v) // src/math.c:40-51
{
	float len = length(v);
	if (len < 0.001f * 1.19209290e-07f)
	{
		return v2(0.0f, 0.0f);
	}
	float inv = 1.0f / len;
	return v2(inv * v.x, inv * v.y);
}

S2_DEV Rot normalizeRot(Rot q) // math.h:201-207
{
	float mag = sqrtf(q.s * q.s + q.c * q.c);
	float invMag = mag > 0.0f ? 1.0f / mag : 0.0f;
	Rot qn;
	qn.s = q.s * invMag;
	qn.c = q.c * invMag;
	return qn;
}

S2_DEV Rot integrateRot(Rot q1, float omegah) // math.h:209-223
{
	Rot q2;
	q2.s = q1.s + omegah * q1.c;
	q2.c = q1.c - omegah * q1.s;
	return normalizeRot(q2);
}

S2_DEV float computeAngularVelocity(Rot q1, Rot q2, float inv_h) // math.h:238-252
{
	return inv_h * (q2.s * q1.c - q2.c * q1.s);
}

// fdlibm-lineage single precision atan / atan2 (the algorithm glibc's flt-32 e_atan2f.c / s_atanf.c
// implement): restated here from the published algorithm so that joint-limit angles do not depend on
// the device libm.  Pure fp32 IEEE operations.
S2_DEV float s2_atanf(float x)
{
	const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
	const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
	const float aT[11] = {3.3333334327e-01f,  -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
						  9.0908870101e-02f,  -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
						  4.9768779427e-02f,  -3.6531571299e-02f, 1.6285819933e-02f};
	int32_t hx = __float_as_int(x);
	int32_t ix = hx & 0x7fffffff;
	int id;
	if (ix >= 0x4c000000) // |x| >= 2^25
	{
		if (ix > 0x7f800000)
		{
			return x + x;
		}
		return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
	}
	if (ix < 0x3ee00000) // |x| < 0.4375
	{
		if (ix < 0x31000000) // |x| < 2^-29
		{
			return x;
		}
		id = -1;
	}
	else
	{
		x = fabsf(x);
		if (ix < 0x3f980000) // |x| < 1.1875
		{
			if (ix < 0x3f300000) // 7/16 <= |x| < 11/16
			{
				id = 0;
				x = (2.0f * x - 1.0f) / (2.0f + x);
			}
			else // 11/16 <= |x| < 19/16
			{
				id = 1;
				x = (x - 1.0f) / (x + 1.0f);
			}
		}
		else
		{
			if (ix < 0x401c0000) // |x| < 2.4375
			{
				id = 2;
				x = (x - 1.5f) / (1.0f + 1.5f * x);
			}
			else // 2.4375 <= |x| < 2^34
			{
				id = 3;
				x = -1.0f / x;
			}
		}
	}
	float z = x * x;
	float w = z * z;
	float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
	float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
	if (id < 0)
	{
		return x - x * (s1 + s2);
	}
	z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
	return (hx < 0) ? -z : z;
}

S2_DEV float s2_atan2f(float y, float x)
{
	const float tiny = 1.0e-30f;
	const float pi_o_4 = 7.8539818525e-01f;
	const float pi_o_2 = 1.5707963705e+00f;
	const float pi = 3.1415927410e+00f;
	const float pi_lo = -8.7422776573e-08f;
	int32_t hx = __float_as_int(x);
	int32_t ix = hx & 0x7fffffff;
	int32_t hy = __float_as_int(y);
	int32_t iy = hy & 0x7fffffff;
	if (ix > 0x7f800000 || iy > 0x7f800000)
	{
		return x + y;
	}
	if (hx == 0x3f800000)
	{
		return s2_atanf(y);
	}
	int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
	if (iy == 0)
	{
		switch (m)
		{
			case 0:
			case 1:
				return y;
			case 2:
				return pi + tiny;
			case 3:
				return -pi - tiny;
		}
	}
	if (ix == 0)
	{
		return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	}
	if (ix == 0x7f800000)
	{
		if (iy == 0x7f800000)
		{
			switch (m)
			{
				case 0:
					return pi_o_4 + tiny;
				case 1:
					return -pi_o_4 - tiny;
				case 2:
					return 3.0f * pi_o_4 + tiny;
				case 3:
					return -3.0f * pi_o_4 - tiny;
			}
		}
		else
		{
			switch (m)
			{
				case 0:
					return 0.0f;
				case 1:
					return -0.0f;
				case 2:
					return pi + tiny;
				case 3:
					return -pi - tiny;
			}
		}
	}
	if (iy == 0x7f800000)
	{
		return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
	}
	int32_t k = (iy - ix) >> 23;
	float z;
	if (k > 60)
	{
		z = pi_o_2 + 0.5f * pi_lo;
	}
	else if (hx < 0 && k < -60)
	{
		z = 0.0f;
	}
	else
	{
		z = s2_atanf(fabsf(y / x));
	}
	switch (m)
	{
		case 0:
			return z;
		case 1:
			return -z;
		case 2:
			return pi - (z - pi_lo);
		default:
			return (z - pi_lo) - pi;
	}
}

S2_DEV float relativeAngle(Rot b, Rot a) // math.h:320-327
{
	float s = b.s * a.c - b.c * a.s;
	float c = b.c * a.c + b.s * a.s;
	return s2_atan2f(s, c);
}

S2_DEV V2 rotate(Rot q, V2 v) { return v2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }	  // math.h:330-341
S2_DEV V2 invRotate(Rot q, V2 v) { return v2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); } // math.h:344-347

S2_DEV V2 mulMV(M22 A, V2 v) // math.h:386-390
{
	return v2(A.cx.x * v.x + A.cy.x * v.y, A.cx.y * v.x + A.cy.y * v.y);
}

S2_DEV M22 inverse22(M22 A) // math.h:392-406
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

S2_DEV V2 solve22(M22 A, V2 b) // math.h:410-420
{
	float a11 = A.cx.x, a12 = A.cy.x, a21 = A.cx.y, a22 = A.cy.y;
	float det = a11 * a22 - a12 * a21;
	if (det != 0.0f)
	{
		det = 1.0f / det;
	}
	return v2(det * (a22 * b.x - a12 * b.y), det * (a11 * b.y - a21 * b.x));
}

// ---------------------------------------------------------------------------------------------
// SoA views (device pointers; passed to kernels by value)
// ---------------------------------------------------------------------------------------------

// Body flags
#define S2F_LIVE 1u	   // not a free pool slot
#define S2F_DYNAMIC 2u // type == dynamic
#define S2F_MOVES 4u   // live && type != static  (s2IntegratePositions / s2FinalizePositions)

// Body state that the sweeps gather by index.  Two 16-byte records per body so a constraint reads
// a body with two dwordx4 loads and writes its velocity with one.
struct BodyView
{
	float4* vel; // {vx, vy, w, unused}
	float4* dq;	 // {deltaPosition.x, deltaPosition.y, rot.s, rot.c}
	float2* pos; // center of mass
	// per-step constants of the velocity integrator, precomputed once per step by unpackBodies
	float4* integ;	// {h*invMass*(f + m*g*gs).x, ....y, h*invI*torque, 1/(1+h*linearDamping)}
	float* angDamp; // 1/(1+h*angularDamping)
	float2* massInv; // {invMass, invI}: staged into LDS by the group kernel (constraint_ops.h LdsMassBodies)
	uint32_t* flags;
	// Jacobi accumulation and XPBD history
	float4* dq0; // XPBD: {deltaPosition0, rot0}
	int capacity;
};

// Contact constraints in sweep order (colour-major).  k = position in sweep order.
struct ContactView
{
	int2* bodies;	  // {indexA, indexB} (body-pool slots)
	int2* localBodies; // {indexA, indexB} as group-local slots (constraints that belong to an LDS group)
	float4* mass;	  // {mA, iA, mB, iB}
	float4* nf;		  // {normal.x, normal.y, friction, bits(pointCount | writeA<<8 | writeB<<9)}
	float4* anchor[2]; // {localAnchorA, localAnchorB} relative to the centers of mass
	float4* r0[2];	  // {rA0, rB0}
	float4* param[2]; // {adjustedSeparation, normalMass, tangentMass, separation}
	float4* soft[2];  // {biasCoefficient, massCoefficient, impulseCoefficient, tangentSeparation}
	float2* impulse[2]; // {normalImpulse, tangentImpulse}
	float4* fanchor[2]; // TGS_Sticky: {localFrictionAnchorA, localFrictionAnchorB}
	// PGS_NGS_Block extras
	float4* blockK;	 // {k11, k12, k22, bits(reduced pointCount)}
	float4* blockNM; // inverse of K: {cx.x, cx.y, cy.x, cy.y}
	// Jacobi: per-constraint velocity deltas, summed per body in constraint order
	float4* deltaA; // {dvA.x, dvA.y, dwA, 0}
	float4* deltaB;
	int* contactIndex; // position k -> index into the wire contact array
	int count;
	// positions [skipBegin, skipEnd) are prepared and stored by the resident-island kernel itself, straight from and to the
	// wire contacts (strip_kernel.hip: islandStepKernel): the prologue / epilogue launches pass them by
	int skipBegin, skipEnd;
};

#define S2C_WRITE_A 0x100u
#define S2C_WRITE_B 0x200u

struct JointView
{
	int2* bodies;
	int2* localBodies;
	float4* frame;	  // {localAnchorA, localAnchorB} relative to centers of mass
	float4* mass;	  // {mA, iA, mB, iB}
	float4* pivot;	  // pivotMass {cx.x, cx.y, cy.x, cy.y}
	float4* soft;	  // {biasCoefficient, massCoefficient, impulseCoefficient, axialMass}
	float2* centerDiff0;
	float2* impulse;  // in/out
	float4* axial;	  // {motorImpulse, lowerImpulse, upperImpulse, mouse: body I of B}
	float4* limits;	  // {referenceAngle, lowerAngle, upperAngle, maxMotorTorque}
	float4* misc;	  // {motorSpeed, bits(flags), mouse hertz, mouse dampingRatio}
	float2* target;	  // mouse targetA
	float4* origin;	  // {localOriginAnchorA, localOriginAnchorB}
	int* jointIndex;  // position -> index into the wire joint array
	int count;
};

// the joint blocks of a prologue launch (joint_prep.h: prepareJointsBlock): what launchPrepareJoints would have been given; blocks 0: none
struct JointPrepArgs
{
	JointView jv;
	const s2amdJoint* wire;
	float h, hertz;
	int kind, warmStart, blocks;
};

// the stage-4 blocks of a world step's epilogue launch (contact_kernels.hip: storeImpulsesKernel; refit_ops.h): what launchStage4 would
// have been given; shapes == nullptr: none
struct Stage4Args
{
	s2amdShape* shapes;
	int shapeCapacity;
	float2* origins;
	int* summary;
};

#define S2J_MOUSE 1u
#define S2J_ENABLE_MOTOR 2u
#define S2J_ENABLE_LIMIT 4u
#define S2J_WRITE_A 0x100u
#define S2J_WRITE_B 0x200u

// body-centric kernels (jacobiApplyKernel, warmStartBodiesKernel): a body with more incident constraints than this is
// walked by a whole wave instead of one thread (the host lists those bodies beside the adjacency)
#define S2_HEAVY_DEGREE 12

struct StepConsts
{
	float dt, inv_dt, h, inv_h;
	float gravityX, gravityY;
	int iterations, extraIterations;
	int warmStart;
	// PREP_SOFT plans: the soft-contact coefficient triples {bias, mass, impulse} of a dynamic-dynamic
	// constraint [0] and of one with a static side [1] (solve_common.c:219, 262-271), computed on the host with the
	// operations of prepareContactsKernel; softDiet == 0 for every other plan
	float softCoef[2][3];
	int softDiet;
};

S2_DEV uint32_t asBits(float f) { return __float_as_uint(f); }
S2_DEV float fromBits(uint32_t u) { return __uint_as_float(u); }

// ---------------------------------------------------------------------------------------------
// Step programs and LDS groups
// ---------------------------------------------------------------------------------------------
// A solver driver (one s2Solve_* of the reference) is recorded once per step as a list of Ops.
// The same list is executed two ways: by the host as one kernel launch per op and colour batch
// (bodies in HBM), and by group_kernel.hip, where one workgroup walks the whole list for a
// "group" -- a set of small islands (or the sequential tail of a big one) whose bodies fit in LDS.
enum OpCode
{
	OP_INTEGRATE_VEL,
	OP_INTEGRATE_POS,  // h
	OP_FINALIZE,	   // flag = dynamicOnly
	OP_XPBD_INTEGRATE, // h
	OP_XPBD_PROJECT,   // inv_h
	OP_JOINT_SWEEP,	   // kind, h, inv_h, useBias
	OP_WARM,		   // kind
	OP_SOLVE_SOFT,	   // kind, inv_h, useBias
	OP_SOLVE_RIGID,	   // kind, inv_h
	OP_SOLVE_STICKY,   // inv_h, useBias
	OP_SOLVE_NGS,
	OP_XPBD_POS, // h
	OP_XPBD_VEL, // h
	OP_BLOCK_VEL,
	OP_BLOCK_POS,
	OP_JACOBI_APPLY
};

struct Op
{
	int code, kind, useBias, flag;
	float h, inv_h, f0, f1;
};

// batches: {begin, end, sequential?, 0} ranges into the constraint / joint SoA
struct GroupTable
{
	const int* bodyOffsets; // [groupCount + 1] into bodyIds
	const int* bodyIds;		// pool slot of each local body; bit 31 set = owned (written back)
	const int* cBatchOffsets;
	const int4* cBatches;
	const int* jBatchOffsets;
	const int4* jBatches;
	int groupCount;
};
#define S2G_OWNED 0x80000000u

// Strip groups as the lean strip kernel reads them (strip_kernel.hip): one 128-byte descriptor per group
#define S2_STRIP_ROUNDS 6	   // colour batches a thread preloads (lean launches; the persistent kernel's narrow variant)
#define S2_STRIP_ROUNDS_MAX 8  // ... its wide variant, for strips whose greedy colouring needs a 7th or 8th colour
#define S2_STRIP_BODY_CHUNKS 4 // bodies per thread: a group stages at most 4 * 256 bodies
struct StripDesc
{
	int bodyBase, bodyCount, ownedCount; // range of GroupTable::bodyIds; the owned bodies come first
	int batchCount;
	int slotBase, slotCount, slotOffBase; // warm start: incident (constraint, side) slots of the owned bodies
	int pad;
	int4 batch[S2_STRIP_ROUNDS_MAX]; // {begin, end, 0, 0} ranges of k
};
struct StripOps
{
	int integratePos, integrateVel, sweep, useBias; // stages, in this order: positions, velocities, [warm start], sweep
	float posH, inv_h;
};
struct StripTableView
{
	const StripDesc* descs;
	const int* bodyIds;
	const int2* slots;		// {k << 1 | side, local body}
	const int* slotOffsets; // per group: ownedCount + 1 offsets into the group's slots
	int groupCount;
	int ldsRecords; // float4 records of dynamic LDS a launch needs
};

// Persistent strip step (strip_kernel.hip: stripStepKernel): workgroup i owns strip i for the whole step and
// sweeps BOTH of its seams (i-1 | i and i | i+1); the two workgroups of a seam compute it redundantly from
// identical inputs, so one symmetric exchange of seam bodies per sweep (after the interior rounds) suffices.
#define S2_PERSIST_B_ROUNDS 4
#define S2_PERSIST_Q_NARROW 6 // LDS records per seam constraint: TGS_Soft with current-anchor warm start (strip_kernel.hip: PersistRegs)
#define S2_PERSIST_Q_WIDE 8   // ... every other kind // colour batches of a seam
struct PersistDesc
{
	int importCount[2];	  // [0] from the left neighbour (its bodies that seam i-1 touches), [1] from the right
	int importIdBase[2];  // into importIds[]: body-pool slot of every imported body (initial load)
	int exportCount[2];	  // [0] to the left (my bodies that seam i-1 touches), [1] to the right
	int exportSrcBase[2]; // into exportSrc[]: own LDS index of every exported body
	int inBase[2];		  // granule offset of the buffer this workgroup READS (per parity: + parityStride)
	int outBase[2];		  // granule offset of the buffer this workgroup WRITES
	int remapBase[2];	  // seam-group-local body -> LDS index of this workgroup, per seam side
	int seamBatchCount[2];
	int2 seamBatch[2][S2_PERSIST_B_ROUNDS]; // {begin, end} ranges of k
	int seamGroup[2]; // the seam's group in the phase-B group table (generic_kernel.hip walks its batch lists), -1: no such seam
	int pad[6];
};
struct PersistView
{
	const PersistDesc* descs;
	const int* remap;
	const int* exportSrc;
	const int* importIds;
	unsigned long long* granules;
	unsigned int* error;	   // host-visible (pinned): set when a hand-off timed out
	unsigned int* deviceError; // the same flag in device memory: what the step's epilogue launch checks
	int parityStride; // granules between the two parities of a buffer
	int censusBase;	  // granule of strip 0 in the per-launch XCD census (wide_kernel.hip): one granule per strip behind the two parities
	int allTwoPoints; // every strip constraint has two manifold points
	float4 softCoef[2]; // the step's two soft-coefficient triples (StepConsts.softCoef), set at launch
	int wideRounds;	  // some strip has 7 or 8 interior colour batches: the ROUNDS == 8 kernel variant
	int seamRegs;	  // no seam has more than two colour batches: the seam constraints stay in registers (SEAMREG variant)
	int maxRoundsA, maxSeamRounds; // the most interior colour batches of a strip / colour batches of a seam
	int pairLanes;	  // the partition fits pair_kernel.hip: pairStepKernel (<= 6 interior batches per strip, <= 2 per seam)
	int nearHandoff;  // hand-offs between two workgroups that the census finds on one XCD may use workgroup-scope stores (they stay in that L2);
					  // 0: agent-scope stores everywhere (option near_handoff, and after a hand-off time-out while this was on)
	int wideOnly;	  // (host) a spare round beyond those opened since the build: only wide_kernel.hip's budgets were kept up to date (IncrementalStrips)
	int ldsRecords;
	int parkSeamWidth, parkInteriorWidth; // wide_kernel.hip: lanes that hold a record in a parked seam round (rounds 3-4) / interior round (7-8), >= 64
	int bodyRecords; // ... of them the staged bodies alone (wide_kernel.hip keeps no seam constraint in the kernel-independent LDS budget)
	int debugSkip; // timing experiments only (results are wrong; compiled in with -DS2_PERSIST_INSTRUMENTED=1): 1 = no hand-offs, 2 = no seam rounds, 4 = no interior rounds;
				   // 8 = fault injection for the fallback test: workgroup 1 never publishes its seam bodies
	unsigned int spinLimit; // polls before a hand-off is declared dead
	// wide_kernel.hip, self-contained variant (the step is that one launch: no epilogue clears the hand-off buffers or stands down
	// after a failure): state[0] = epoch base of this step's hand-off tags (the kernel adds 64 per step, so no tag ever comes back),
	// state[16] = commit counter -- every workgroup arrives once per step and writes its results only when all have
	unsigned int* state;
	int maxStaged;		// the most bodies a strip stages (own list + both imports)
	int maxStripBodies; // ... of them in its own list (owned + read-only replicas)
	int bodyWarm;		// wide_kernel.hip: s2WarmStartContacts as one body-centric pass (set at launch when the term table fits LDS)
	int clearOwn;		// wide_kernel.hip, sliced step (one launch per sweep): the kernel zeroes the hand-off buffers it reads and its census entry at its
						// end, as the step's epilogue launch does after the last slice -- every launch starts from zero tags without a memset in between
	// wide_kernel.hip, S2_WIDE_OVERFLOW: the contacts in the overflow positions behind the strips swept INSIDE the persistent launch by one
	// more workgroup (the last of the grid).  overflowBodies: the bodies those contacts touch (pool slot; bit 30: nothing writes it -- a
	// static or kinematic body, no exchange; -1: free entry), index = the contacts' c.localBodies entries.  After every sweep the strip
	// that owns such a body hands {v, w} to that workgroup and every strip that stages it (the owner, the neighbour that imports it)
	// takes the result back: granules at overflowGranBase, [in | out][S2_OVERFLOW_RING][S2_OVERFLOW_BODIES][4], tag = the sweep's number.
	const int* overflowBodies;
	int overflowBodyCount;
	int overflowBegin, overflowEnd; // sweep positions of the overflow region
	int overflowGranBase;
	int overflowKernel; // (set at launch) this launch carries that workgroup: grid = strips + 1
	unsigned long long* debugTimes; // S2AMD_DEBUG_TIMES: wall_clock64() of one workgroup at kernel start, after the loads, after every op, at the end
};
#define S2_OVERFLOW_BODIES 64
#define S2_OVERFLOW_RING 4
#define S2_OVERFLOW_GRANULES (2 * S2_OVERFLOW_RING * S2_OVERFLOW_BODIES * 4)

// s2Solve_Jacobi as one persistent launch (jacobi_kernel.hip; tables: solver_jacobi.cpp).  One descriptor per block of bodies; every
// list lives in ONE int array (`ints`).
#define S2_JACOBI_HEAVY 32 // incidence entries beyond which a body is walked by a wave (the block's heavy list) instead of a lane
struct JacobiBlockDesc
{
	int ownedBase, ownedCount;			 // pool slots of the bodies this block owns (integrates, applies, publishes)
	int importBase, importCount;		 // pool slots of the bodies of other blocks its constraints read, then importCount flags (bit 0: somebody owns it)
	int constraintBase, constraintCount; // per constraint {position in the sweep order | bit 30: this block stores its impulses, local slot A, local slot B}
	int listBase, rangeBase;			 // incidence entries (local constraint << 1 | side) of the owned bodies in pool order; per owned body {first, count}
	int exportBase, exportCount;		 // local slots of the owned bodies other blocks import
	int heavyBase, heavyCount;			 // local slots of the owned bodies with more than S2_JACOBI_HEAVY entries
	int jointBase, jointCount;			 // per joint {position in the joint SoA, local slot of body A or -1, local slot of body B or -1}, in sweep order
};
struct JacobiView
{
	const JacobiBlockDesc* descs;
	const int* ints;
	unsigned long long* granules; // [2 parities][bodyCapacity][4]: {epoch, value} of v.x, v.y, w (persist_handoff.h)
	int parityStride;			  // granules between the parities
	unsigned int* error;		  // host-visible: a hand-off timed out
	unsigned int* deviceError;
	unsigned int spinLimit;
	int blockCount;
	int debugSkip; // timing experiments only (results are wrong): 1 = no wave walks of long lists, 2 = no joints, 4 = no exchange, 8 = no contact pass (option persist_debug)
};

// Message-passing tables of the global part (see MsgBodies in constraint_ops.h)
struct MsgView
{
	float4* vel;		   // [2 * globalCount] per-constraint-side velocity copies
	float4* dq;			   // [2 * globalCount] per-constraint-side pose copies
	const int* next;	   // [2 * globalCount] copy that receives the updated velocity
	const int* firstSlot;  // [bodyCapacity] copy holding a body's current velocity between sweeps (-1: body has no copies)
	const int* slotOffsets; // [bodyCapacity + 1] CSR of all copies of a body
	const int* slotList;
};
