// Joint kernels: revolute (src/revolute_joint.c) and mouse (src/mouse_joint.c) joints, dispatched
// like src/joint.c:294-465.  One thread per joint; joints are stored SoA in sweep (colour-major)
// order and a sweep kernel is launched per colour batch, exactly like the contact kernels.  The
// per-joint arithmetic lives in constraint_ops.h (shared with the LDS group kernel).

#include "constraint_ops.h"
#include "joint_prep.h"

#define S2_BLOCK 256

// s2PrepareJoint (joint.c:297-312), s2PrepareJoint_Soft (:372-387), s2PrepareJoint_XPBD (:432-447): joint_prep.h.  The step's prologue
// launch carries these blocks itself (Executor::run); this kernel is the stand-alone form.
template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void prepareJointsKernel(JointView jv, const uint32_t* hostFlags, const s2amdJoint* wire, const s2amdBody* wireBodies,
																StepConsts sc, float h, float hertz, int warmStart, int posSolver)
{
	prepareJointOne<KIND>(jv, hostFlags, wire, wireBodies, sc, h, hertz, warmStart, posSolver, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

template <int KIND>
__global__ __launch_bounds__(S2_BLOCK) void solveJointsKernel(JointView jv, BodyView b, int begin, int end, StepConsts sc, float h, float inv_h,
															  int useBias)
{
	int k = begin + blockIdx.x * blockDim.x + threadIdx.x;
	if (k < end)
	{
		GlobalBodies gb{b.vel, b.dq};
		solveJointsOne<KIND>(jv, gb, sc, h, inv_h, useBias, k);
	}
}

// stepFailed: see storeImpulsesKernel (contact_kernels.hip) -- a persistent step whose hand-offs timed out leaves EVERY wire
// array as it was, so that the repeated step warm starts from the same joint impulses
__global__ __launch_bounds__(S2_BLOCK) void storeJointsKernel(JointView jv, s2amdJoint* wire, const unsigned int* stepFailed)
{
	int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= jv.count || (stepFailed != nullptr && *stepFailed != 0u))
	{
		return;
	}
	s2amdJoint* w = wire + jv.jointIndex[k];
	float2 im = jv.impulse[k];
	float4 ax = jv.axial[k];
	w->impulse[0] = im.x, w->impulse[1] = im.y;
	w->motorImpulse = ax.x;
	if (w->type == S2AMD_JOINT_REVOLUTE)
	{
		w->lowerImpulse = ax.y;
		w->upperImpulse = ax.z;
	}
}

static inline dim3 gridFor(int n)
{
	return dim3((unsigned)((n + S2_BLOCK - 1) / S2_BLOCK));
}

void launchPrepareJoints(hipStream_t s, int kind, const JointView& j, const uint32_t* hostFlags, const s2amdJoint* wire, const s2amdBody* wireBodies,
						 const StepConsts& sc, float h, float hertz, int warmStart, int posSolver)
{
	if (j.count <= 0)
	{
		return;
	}
	dim3 g = gridFor(j.count), t(S2_BLOCK);
	switch (kind)
	{
		case JPREP_PLAIN:
			prepareJointsKernel<JPREP_PLAIN><<<g, t, 0, s>>>(j, hostFlags, wire, wireBodies, sc, h, hertz, warmStart, posSolver);
			break;
		case JPREP_SOFT:
			prepareJointsKernel<JPREP_SOFT><<<g, t, 0, s>>>(j, hostFlags, wire, wireBodies, sc, h, hertz, warmStart, posSolver);
			break;
		case JPREP_XPBD:
			prepareJointsKernel<JPREP_XPBD><<<g, t, 0, s>>>(j, hostFlags, wire, wireBodies, sc, h, hertz, warmStart, posSolver);
			break;
	}
}

#define S2_JLAUNCH(K) solveJointsKernel<K><<<gridFor(end - begin), dim3(S2_BLOCK), 0, s>>>(j, b, begin, end, sc, h, inv_h, useBias);

void launchSolveJoints(hipStream_t s, int kind, const JointView& j, const BodyView& b, int begin, int end, const StepConsts& sc, float h,
					   float inv_h, int useBias)
{
	if (end <= begin)
	{
		return;
	}
	switch (kind)
	{
		case JSOLVE_PLAIN:
			S2_JLAUNCH(JSOLVE_PLAIN)
			break;
		case JSOLVE_SOFT:
			S2_JLAUNCH(JSOLVE_SOFT)
			break;
		case JSOLVE_BAUMGARTE:
			S2_JLAUNCH(JSOLVE_BAUMGARTE)
			break;
		case JSOLVE_POSITION:
			S2_JLAUNCH(JSOLVE_POSITION)
			break;
		case JSOLVE_XPBD:
			S2_JLAUNCH(JSOLVE_XPBD)
			break;
		case JSOLVE_WARM:
			S2_JLAUNCH(JSOLVE_WARM)
			break;
	}
}

// Joint warm start, body-centric (the counterpart of warmStartBodiesKernel for contacts): s2WarmStartRevolute / s2WarmStartMouse
// (revolute_joint.c:107-150, mouse_joint.c:85-107) only ADD velocity-independent terms, so each body adds the terms of its
// incident joints in sweep order -- the same additions in the same order as the coloured sweep -- in ONE launch instead of
// one per joint colour.  adjRange[i] = {first entry, count} in adjList; an entry is (k << 1) | side.
__global__ __launch_bounds__(S2_BLOCK) void warmStartJointsBodiesKernel(JointView jv, BodyView b, const int2* adjRange, const int* adjList)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= b.capacity)
	{
		return;
	}
	const int2 range = adjRange[i];
	if (range.y <= 0)
	{
		return;
	}
	const float4 v0 = b.vel[i], d = b.dq[i];
	Rot q;
	q.s = d.z, q.c = d.w;
	V2 v = v2(v0.x, v0.y);
	float w = v0.z;
	bool wrote = false;
	for (int e = range.x; e < range.x + range.y; ++e)
	{
		const int key = adjList[e];
		const JState s = loadJoint<S2_IDX_GLOBAL>(jv, key >> 1);
		if (s.flags & S2J_MOUSE)
		{
			if ((key & 1) && (s.flags & S2J_WRITE_B))
			{
				V2 rB = rotate(q, s.lB);
				v = mulAdd(v, s.mB, s.impulse);
				w += s.iB * (cross(rB, s.impulse) + s.motorImpulse);
				wrote = true;
			}
			continue;
		}
		const float axialImpulse = s.motorImpulse + s.lowerImpulse - s.upperImpulse;
		const V2 P = s.impulse;
		if ((key & 1) == 0)
		{
			if (s.flags & S2J_WRITE_A)
			{
				V2 rA = rotate(q, s.lA);
				v = mulSub(v, s.mA, P);
				w -= s.iA * (cross(rA, P) + axialImpulse);
				wrote = true;
			}
		}
		else if (s.flags & S2J_WRITE_B)
		{
			V2 rB = rotate(q, s.lB);
			v = mulAdd(v, s.mB, P);
			w += s.iB * (cross(rB, P) + axialImpulse);
			wrote = true;
		}
	}
	if (wrote)
	{
		b.vel[i] = make_float4(v.x, v.y, w, 0.0f);
	}
}

void launchWarmStartJointsBodies(hipStream_t s, const JointView& j, const BodyView& b, const int2* adjRange, const int* adjList)
{
	if (b.capacity > 0)
	{
		warmStartJointsBodiesKernel<<<gridFor(b.capacity), dim3(S2_BLOCK), 0, s>>>(j, b, adjRange, adjList);
	}
}

void launchStoreJoints(hipStream_t s, const JointView& j, s2amdJoint* wire, const unsigned int* stepFailed)
{
	if (j.count > 0)
	{
		storeJointsKernel<<<gridFor(j.count), dim3(S2_BLOCK), 0, s>>>(j, wire, stepFailed);
	}
}

S2_DEFINE_WARM(joint_kernels)
