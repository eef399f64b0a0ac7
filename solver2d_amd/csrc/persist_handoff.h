// Hand-off primitives of the persistent step kernels (strip_kernel.hip: stripStepKernel, pair_kernel.hip: pairStepKernel):
// 8-byte {epoch, value} granules written with ONE agent-scope (sc1, write-through) store each and polled with agent-scope
// loads -- the data is the flag, no fence (cdna_hip_programming.md G16, form R2).  Every poll loop is bounded and reports
// through a host-visible error word (never a hang).
#pragma once

#include "s2_device.h"

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

S2_DEV void putGranule(gu64* g, unsigned epoch, float v)
{
	__hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same granule for a reader on the writer's own XCD: the store stays in that XCD's L2 (no write-through to the fabric),
// where the reader's agent-scope (L1-bypassing) poll finds it.  Only valid when writer and reader share the L2 -- the caller
// has established that (wide_kernel.hip: the XCC id handshake).
S2_DEV void putGranuleNear(gu64* g, unsigned epoch, float v)
{
	__hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#define S2_PERSIST_SPIN_LIMIT (1u << 21)

template <int N> S2_DEV bool getGranules(gu64* g, unsigned epoch, float (&v)[N], unsigned int* error, unsigned int* deviceError, unsigned int spinLimit)
{
	for (unsigned spins = 0;; ++spins)
	{
		bool ok = true;
#pragma unroll
		for (int k = 0; k < N; ++k)
		{
			u64 x = __hip_atomic_load(g + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			v[k] = __uint_as_float((unsigned)x);
			ok = ok && (unsigned)(x >> 32) == epoch;
		}
		if (ok)
		{
			return true;
		}
		if ((spins & 255u) == 255u)
		{
			if (spins >= spinLimit)
			{
				__hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(deviceError, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return false;
			}
			if (__hip_atomic_load(deviceError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
			{
				return false;
			}
		}
		__builtin_amdgcn_s_sleep(1);
	}
}
