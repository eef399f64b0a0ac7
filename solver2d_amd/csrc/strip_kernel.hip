// Strip kernel: the lean launch of the strip path (solver.cpp: partitionStrips) for the soft contact
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
