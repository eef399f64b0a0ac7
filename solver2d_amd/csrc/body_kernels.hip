// Per-body kernels: one thread per body-pool slot, plain streaming passes over the body SoA.
// Reference: src/solve_common.c:10-91 (integrate velocities / positions, finalize),
// src/solve_jacobi.c:233-245 (apply accumulated deltas), src/solve_xpbd.c:411-449, :465-489.

#include "body_ops.h"
#include "joint_prep.h"

#define S2_BLOCK 256

// Wire AoS -> working SoA, plus the per-step constants of s2IntegrateVelocities
// (solve_common.c:30-41): with a = (h*invMass) * (force + (mass*gravityScale)*gravity),
// aw = (h*invI)*torque, ld = 1/(1 + h*linearDamping), ad = 1/(1 + h*angularDamping) the update is
// v = ld * (v + a), w = (w + aw) * ad -- the same fp32 operations in the same order.
// blocks [0, bodyBlocks): wire bodies -> SoA; further blocks: manifold.constraintIndex of every contact slot
// (solver_step.cpp: the pool-order gather index; a field no solver kernel reads)
__global__ __launch_bounds__(S2_BLOCK) void unpackBodiesKernel(BodyView b, const s2amdBody* wire, const uint32_t* hostFlags, StepConsts sc, float h,
															   int bodyBlocks, s2amdContact* wireContacts, int contactCapacity, const int* gatherIndex, JointPrepArgs jp, int posSolver)
{
	if ((int)blockIdx.x >= (int)gridDim.x - jp.blocks)
	{
		// the joints' preparation (joint.c:297-447; joint_prep.h), from the wire records
		prepareJointsBlock(jp, hostFlags, wire, sc, posSolver, (int)blockIdx.x - ((int)gridDim.x - jp.blocks));
		return;
	}
	if ((int)blockIdx.x >= bodyBlocks)
	{
		int c = ((int)blockIdx.x - bodyBlocks) * (int)blockDim.x + (int)threadIdx.x;
		if (c < contactCapacity)
		{
			wireContacts[c].constraintIndex = gatherIndex[c];
		}
		return;
	}
	unpackBodyOne(b, wire, hostFlags, sc, h, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

__global__ __launch_bounds__(S2_BLOCK) void packBodiesKernel(BodyView b, s2amdBody* wire)
{
	packBodyOne(b, wire, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

#define S2_BODY_KERNEL_HEAD                                                                                                      \
	int i = blockIdx.x * blockDim.x + threadIdx.x;                                                                               \
	if (i >= b.capacity || (b.flags[i] & S2F_IN_GROUP) != 0)                                                                     \
	{                                                                                                                            \
		return;                                                                                                                  \
	}                                                                                                                            \
	GlobalBodies gb{b.vel, b.dq};

// Streaming passes over the bodies that are NOT owned by an LDS group (those are advanced inside
// group_kernel.hip).  Arithmetic: body_ops.h.
__global__ __launch_bounds__(S2_BLOCK) void integrateVelocitiesKernel(BodyView b)
{
	S2_BODY_KERNEL_HEAD
	integrateVelocitiesOne(gb, i, b, i);
}
__global__ __launch_bounds__(S2_BLOCK) void integratePositionsKernel(BodyView b, float h)
{
	S2_BODY_KERNEL_HEAD
	integratePositionsOne(gb, i, b, i, h);
}
__global__ __launch_bounds__(S2_BLOCK) void finalizePositionsKernel(BodyView b, int dynamicOnly)
{
	S2_BODY_KERNEL_HEAD
	finalizePositionsOne(gb, i, b, i, dynamicOnly, true);
}
__global__ __launch_bounds__(S2_BLOCK) void xpbdIntegrateKernel(BodyView b, float h)
{
	S2_BODY_KERNEL_HEAD
	xpbdIntegrateOne(gb, b.dq0, i, b, i, h);
}
__global__ __launch_bounds__(S2_BLOCK) void xpbdProjectKernel(BodyView b, float inv_h)
{
	S2_BODY_KERNEL_HEAD
	xpbdProjectOne(gb, b.dq0, i, b, i, inv_h);
}

// ---- message-passing variants: a body that has per-constraint copies keeps its CURRENT velocity in
// copy firstSlot[i] (the last toucher of a sweep writes there) and its pose in every copy ----
S2_DEV void scatterPose(const MsgView& m, int i, float4 d)
{
	for (int e = m.slotOffsets[i]; e < m.slotOffsets[i + 1]; ++e)
	{
		m.dq[m.slotList[e]] = d;
	}
}

__global__ __launch_bounds__(S2_BLOCK) void integrateVelocitiesMsgKernel(BodyView b, MsgView m)
{
	S2_BODY_KERNEL_HEAD
	int f = m.firstSlot[i];
	if (f < 0)
	{
		integrateVelocitiesOne(gb, i, b, i);
		return;
	}
	GlobalBodies sb{m.vel, m.dq};
	integrateVelocitiesOne(sb, f, b, i);
}

__global__ __launch_bounds__(S2_BLOCK) void integratePositionsMsgKernel(BodyView b, MsgView m, float h)
{
	S2_BODY_KERNEL_HEAD
	int f = m.firstSlot[i];
	if (f < 0)
	{
		integratePositionsOne(gb, i, b, i, h);
		return;
	}
	if ((b.flags[i] & S2F_MOVES) == 0)
	{
		return;
	}
	GlobalBodies sb{m.vel, m.dq};
	integratePositionsOne(sb, f, b, i, h);
	scatterPose(m, i, m.dq[f]);
}

__global__ __launch_bounds__(S2_BLOCK) void finalizePositionsMsgKernel(BodyView b, MsgView m, int dynamicOnly)
{
	S2_BODY_KERNEL_HEAD
	int f = m.firstSlot[i];
	if (f < 0)
	{
		finalizePositionsOne(gb, i, b, i, dynamicOnly, true);
		return;
	}
	uint32_t need = dynamicOnly ? S2F_DYNAMIC : S2F_MOVES;
	if ((b.flags[i] & need) == 0)
	{
		return;
	}
	GlobalBodies sb{m.vel, m.dq};
	finalizePositionsOne(sb, f, b, i, dynamicOnly, true);
	scatterPose(m, i, m.dq[f]);
}

// copies -> body arrays at the end of the step
__global__ __launch_bounds__(S2_BLOCK) void gatherMessageSlotsKernel(BodyView b, MsgView m)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= b.capacity || (b.flags[i] & S2F_IN_GROUP) != 0)
	{
		return;
	}
	int f = m.firstSlot[i];
	if (f >= 0)
	{
		b.vel[i] = m.vel[f];
		b.dq[i] = m.dq[f];
	}
}

// Jacobi: the reference adds every constraint's velocity delta into body->dv / dw in constraint
// order and applies the sum afterwards (solve_jacobi.c:126-130, :233-245).  Here each body walks
// its incidence list (ascending constraint index) and performs the same additions in the same
// order -- no atomics, deterministic, bit-identical to the sequential reference.
S2_DEV float laneValue(float v, int lane)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// adjRange[i] = {first entry, entry count} of body i in adjList (ranges carry slack so that the host can add an entry in
// place); heavy[0] = number of heavy bodies, heavy[1..] their slots (the launch is sized for the list's capacity)
__global__ __launch_bounds__(S2_BLOCK) void jacobiApplyKernel(BodyView b, ContactView c, const int2* adjRange, const int* adjList, int bodyBlocks,
															   const int* heavy)
{
	if ((int)blockIdx.x >= bodyBlocks)
	{
		// ---- heavy bodies: one wave each.  64 list entries are loaded at once, one per lane; the additions stay
		// sequential in list order, every lane performing the same ones on values broadcast from lane u. ----
		const int lane = (int)threadIdx.x & 63;
		const int h = ((int)blockIdx.x - bodyBlocks) * (S2_BLOCK / 64) + ((int)threadIdx.x >> 6);
		if (h >= heavy[0])
		{
			return;
		}
		const int i = heavy[1 + h];
		const int2 range = adjRange[i];
		const int begin = range.x, end = range.x + range.y;
		V2 dv = v2(0.0f, 0.0f);
		float dw = 0.0f;
		constexpr int WIDE = 4; // 4 x 64 list entries are loaded before the first addition: two memory round trips per 256 entries
		for (int base = begin; base < end; base += 64 * WIDE)
		{
			int key[WIDE];
			float4 d[WIDE];
#pragma unroll
			for (int g = 0; g < WIDE; ++g)
			{
				const int e = base + 64 * g + lane;
				key[g] = adjList[e < end ? e : begin];
			}
#pragma unroll
			for (int g = 0; g < WIDE; ++g)
			{
				d[g] = (key[g] & 1) ? c.deltaB[key[g] >> 1] : c.deltaA[key[g] >> 1];
			}
#pragma unroll
			for (int g = 0; g < WIDE; ++g)
			{
				const int left = __builtin_amdgcn_readfirstlane(end - (base + 64 * g)); // wave-uniform: a scalar loop, unrolled
				const int n = left < 64 ? left : 64;
				int u = 0;
				for (; u + 8 <= n; u += 8)
				{
#pragma unroll
					for (int t = 0; t < 8; ++t)
					{
						dv = add(dv, v2(laneValue(d[g].x, u + t), laneValue(d[g].y, u + t)));
						dw += laneValue(d[g].z, u + t);
					}
				}
				for (; u < n; ++u)
				{
					dv = add(dv, v2(laneValue(d[g].x, u), laneValue(d[g].y, u)));
					dw += laneValue(d[g].z, u);
				}
			}
		}
		if (lane == 0)
		{
			float4 v = b.vel[i];
			V2 lv = add(v2(v.x, v.y), dv);
			float w = v.z;
			w += dw;
			b.vel[i] = make_float4(lv.x, lv.y, w, 0.0f);
		}
		return;
	}
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= b.capacity)
	{
		return;
	}
	if ((b.flags[i] & S2F_LIVE) == 0)
	{
		return;
	}
	const int2 range = adjRange[i];
	int begin = range.x, end = range.x + range.y;
	if (end - begin > S2_HEAVY_DEGREE)
	{
		return; // a wave of the heavy blocks walks this one
	}
	V2 dv = v2(0.0f, 0.0f);
	float dw = 0.0f;
	for (int e = begin; e < end; ++e)
	{
		int key = adjList[e];
		float4 d = (key & 1) ? c.deltaB[key >> 1] : c.deltaA[key >> 1];
		dv = add(dv, v2(d.x, d.y));
		dw += d.z;
	}
	float4 v = b.vel[i];
	V2 lv = add(v2(v.x, v.y), dv);
	float w = v.z;
	w += dw;
	b.vel[i] = make_float4(lv.x, lv.y, w, 0.0f);
}

// per-island body poses {position, rot} for the inter-GPU exchange (one float4 per body)
__global__ __launch_bounds__(S2_BLOCK) void exportPosesKernel(const s2amdBody* wire, int n, float4* out)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
	{
		return;
	}
	const s2amdBody* w = wire + i;
	out[i] = make_float4(w->position[0], w->position[1], w->rot[0], w->rot[1]);
}

// ... or the per-island body arrays of SURVEY.md 8e: {position, rot} and {linearVelocity, angularVelocity} (28 bytes in two 16-byte records)
__global__ __launch_bounds__(S2_BLOCK) void exportBodiesKernel(const s2amdBody* wire, int n, float4* out)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
	{
		return;
	}
	const s2amdBody* w = wire + i;
	out[2 * i] = make_float4(w->position[0], w->position[1], w->rot[0], w->rot[1]);
	out[2 * i + 1] = make_float4(w->linearVelocity[0], w->linearVelocity[1], w->angularVelocity, 0.0f);
}

static inline dim3 gridFor(int n)
{
	return dim3((unsigned)((n + S2_BLOCK - 1) / S2_BLOCK));
}

bool launchUnpackBodies(hipStream_t s, const BodyView& b, const s2amdBody* wire, const uint32_t* hostFlags, const StepConsts& sc, float h,
						s2amdContact* wireContacts, int contactCapacity, const int* gatherIndex, const JointPrepArgs* joints, int posSolver)
{
	JointPrepArgs jp{};
	if (joints != nullptr)
	{
		jp = *joints;
	}
	if (b.capacity > 0)
	{
		dim3 bodyGrid = gridFor(b.capacity), contactGrid = gatherIndex && contactCapacity > 0 ? gridFor(contactCapacity) : dim3(0);
		unpackBodiesKernel<<<dim3(bodyGrid.x + contactGrid.x + (unsigned)jp.blocks), dim3(S2_BLOCK), 0, s>>>(b, wire, hostFlags, sc, h, (int)bodyGrid.x, wireContacts,
																												contactCapacity, gatherIndex, jp, posSolver);
		return true;
	}
	return false;
}
void launchPackBodies(hipStream_t s, const BodyView& b, s2amdBody* wire)
{
	if (b.capacity > 0)
	{
		packBodiesKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, wire);
	}
}
void launchIntegrateVelocities(hipStream_t s, const BodyView& b)
{
	if (b.capacity > 0)
	{
		integrateVelocitiesKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b);
	}
}
void launchIntegratePositions(hipStream_t s, const BodyView& b, float h)
{
	if (b.capacity > 0)
	{
		integratePositionsKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, h);
	}
}
void launchFinalizePositions(hipStream_t s, const BodyView& b, int dynamicOnly)
{
	if (b.capacity > 0)
	{
		finalizePositionsKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, dynamicOnly);
	}
}
void launchJacobiApply(hipStream_t s, const BodyView& b, const ContactView& c, const int2* adjRange, const int* adjList, const int* heavy, int heavyCapacity)
{
	if (b.capacity > 0)
	{
		const int bodyBlocks = (b.capacity + S2_BLOCK - 1) / S2_BLOCK, heavyBlocks = (heavyCapacity + S2_BLOCK / 64 - 1) / (S2_BLOCK / 64);
		jacobiApplyKernel<<<dim3((unsigned)(bodyBlocks + heavyBlocks)), dim3(S2_BLOCK), 0, s>>>(b, c, adjRange, adjList, bodyBlocks, heavy);
	}
}

// Incremental structure updates (solver_incremental.cpp): 32-bit words written into the device tables the kernels read
__global__ __launch_bounds__(S2_BLOCK) void patchWordsKernel(const uint4* patches, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
	{
		uint4 p = patches[i]; // {address lo, address hi, value, 0}
		*(uint32_t*)(((unsigned long long)p.y << 32) | (unsigned long long)p.x) = p.z;
	}
}
void launchPatchWords(hipStream_t s, const void* devicePatches, int n)
{
	if (n > 0)
	{
		patchWordsKernel<<<gridFor(n), dim3(S2_BLOCK), 0, s>>>((const uint4*)devicePatches, n);
	}
}
void launchXpbdIntegrate(hipStream_t s, const BodyView& b, float h)
{
	if (b.capacity > 0)
	{
		xpbdIntegrateKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, h);
	}
}
void launchXpbdProject(hipStream_t s, const BodyView& b, float inv_h)
{
	if (b.capacity > 0)
	{
		xpbdProjectKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, inv_h);
	}
}

void launchExportPoses(hipStream_t s, const s2amdBody* wire, int n, void* out, int withVelocities)
{
	if (n > 0)
	{
		if (withVelocities)
		{
			exportBodiesKernel<<<gridFor(n), dim3(S2_BLOCK), 0, s>>>(wire, n, (float4*)out);
		}
		else
		{
			exportPosesKernel<<<gridFor(n), dim3(S2_BLOCK), 0, s>>>(wire, n, (float4*)out);
		}
	}
}

void launchIntegrateVelocitiesMsg(hipStream_t s, const BodyView& b, const MsgView& m)
{
	if (b.capacity > 0)
	{
		integrateVelocitiesMsgKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, m);
	}
}
void launchIntegratePositionsMsg(hipStream_t s, const BodyView& b, const MsgView& m, float h)
{
	if (b.capacity > 0)
	{
		integratePositionsMsgKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, m, h);
	}
}
void launchFinalizePositionsMsg(hipStream_t s, const BodyView& b, const MsgView& m, int dynamicOnly)
{
	if (b.capacity > 0)
	{
		finalizePositionsMsgKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, m, dynamicOnly);
	}
}
void launchGatherMessageSlots(hipStream_t s, const BodyView& b, const MsgView& m)
{
	if (b.capacity > 0)
	{
		gatherMessageSlotsKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(b, m);
	}
}

S2_DEFINE_WARM(body_kernels)
