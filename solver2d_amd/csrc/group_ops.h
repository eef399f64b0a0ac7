// Helpers shared by the op interpreters that walk colour batches with a workgroup (group_kernel.hip: groupKernel,
// generic_kernel.hip: genericStepKernel): cache warm-up of a sequential batch, the batch walk itself.
#pragma once

#include "body_ops.h"

// Pull one constraint's SoA records into the cache hierarchy.  The sequential tail is walked by one
// lane, so every miss would be paid serially; this pass lets all lanes of the workgroup issue the
// misses at once, the walk afterwards hits L1/L2.
S2_DEV void touch(float4 v)
{
	asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
}
S2_DEV void prefetchContact(const ContactView& c, int k)
{
	int2 b = c.localBodies[k];
	asm volatile("" ::"v"(b.x), "v"(b.y));
	touch(c.nf[k]);
	touch(c.blockK[k]);
	touch(c.blockNM[k]);
	for (int j = 0; j < 2; ++j)
	{
		touch(c.anchor[j][k]);
		touch(c.r0[j][k]);
		touch(c.param[j][k]);
		touch(c.soft[j][k]);
		touch(c.fanchor[j][k]);
		float2 i = c.impulse[j][k];
		asm volatile("" ::"v"(i.x), "v"(i.y));
	}
}
template <class JV> S2_DEV void prefetchJoint(const JV& j, int k)
{
	int2 b = j.localBodies[k];
	asm volatile("" ::"v"(b.x), "v"(b.y));
	touch(float4(j.frame[k]));
	touch(float4(j.mass[k]));
	touch(float4(j.pivot[k]));
	touch(float4(j.soft[k]));
	touch(float4(j.axial[k]));
	touch(float4(j.limits[k]));
	touch(float4(j.misc[k]));
	float2 a = j.centerDiff0[k], i = j.impulse[k];
	asm volatile("" ::"v"(a.x), "v"(a.y), "v"(i.x), "v"(i.y));
}

template <class P, class F> S2_DEV void forBatches(const int4* batches, int b0, int b1, P prefetch, F f)
{
	for (int bi = b0; bi < b1; ++bi)
	{
		int4 bt = batches[bi];
		if (bt.z)
		{
			for (int k = bt.x + (int)threadIdx.x; k < bt.y; k += (int)blockDim.x)
			{
				prefetch(k);
			}
			__syncthreads();
			if (threadIdx.x == 0)
			{
				for (int k = bt.x; k < bt.y; ++k)
				{
					f(k);
				}
			}
		}
		else
		{
			for (int k = bt.x + (int)threadIdx.x; k < bt.y; k += (int)blockDim.x)
			{
				f(k);
			}
		}
		__syncthreads();
	}
}

