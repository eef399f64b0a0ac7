// Issue-rate probes behind the pair-lane kernel's design (pair_kernel.hip): what ONE wave per SIMD pays per VALU
// instruction (dependent / independent, scalar / packed fp32, DPP), what a second wave on the SIMD changes, the LDS
// round trip and s_barrier at 4 and 8 waves per workgroup.  hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
#define STR2(x) #x
#define STR(x) STR2(x)

template <int MODE> __global__ void probe(unsigned long long* out, float* sink, int iters)
{
	float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 2.0f, f = 3.0f, g = 5.0f, h = 7.0f;
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p = {a, b}, q = {c, d}, r = {e, f}, s = {g, h};
	__shared__ float lds[2048];
	lds[threadIdx.x] = a;
	__syncthreads();
	unsigned idx = threadIdx.x * 4;
	unsigned long long t0 = clock64();
	for (int it = 0; it < iters; ++it)
	{
		if (MODE == 0) // dependent v_mul_f32 chain
		{
			asm volatile(".rept " STR(REP) "\n v_mul_f32 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
		}
		else if (MODE == 1) // 4 independent chains
		{
			asm volatile(".rept " STR(REP) "\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n .endr"
						 : "+v"(a), "+v"(c), "+v"(d), "+v"(e)
						 : "v"(b));
		}
		else if (MODE == 2) // dependent v_pk_mul_f32 chain
		{
			asm volatile(".rept " STR(REP) "\n v_pk_mul_f32 %0, %0, %1\n .endr" : "+v"(p) : "v"(q));
		}
		else if (MODE == 3) // 4 independent v_pk_mul_f32 chains
		{
			asm volatile(".rept " STR(REP) "\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n .endr"
						 : "+v"(p), "+v"(r), "+v"(s), "+v"(q)
						 : "v"(f2{1.0001f, 0.9999f}));
		}
		else if (MODE == 4) // xor + dpp mov + add (the pair exchange as the compiler emits it), dependent
		{
			asm volatile(".rept " STR(REP) "\n v_xor_b32 %1, %2, %0\n s_nop 1\n v_mov_b32_dpp %3, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 %0, %1, %3\n .endr"
						 : "+v"(a), "+v"(c), "+v"(d), "+v"(e)
						 :);
		}
		else if (MODE == 5) // xor + add with a DPP operand, dependent
		{
			asm volatile(".rept " STR(REP) "\n v_xor_b32 %1, %2, %0\n s_nop 1\n v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n .endr"
						 : "+v"(a), "+v"(c)
						 : "v"(d));
		}
		else if (MODE == 6) // LDS round trip: dependent ds_read_b32 chain (address = value read)
		{
			asm volatile(".rept 64\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(idx));
		}
		else if (MODE == 7) // s_barrier back to back
		{
			asm volatile(".rept 64\n s_barrier\n .endr");
		}
		else if (MODE == 8) // ds_read_b128 + dependent use + ds_write_b128 + barrier (a round's fixed cost)
		{
			asm volatile(".rept 64\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %0\n ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n s_barrier\n .endr"
						 : "+v"(a)
						 : "v"(idx));
		}
		else if (MODE == 9) // v_cmp -> v_cndmask dependent pair
		{
			asm volatile(".rept " STR(REP) "\n v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc\n .endr" : "+v"(a) : "v"(b) : "vcc");
		}
		else if (MODE == 10) // v_max_f32 / v_min_f32 dependent pair
		{
			asm volatile(".rept " STR(REP) "\n v_max_f32 %0, %0, %1\n v_min_f32 %0, %0, %2\n .endr" : "+v"(a) : "v"(b), "v"(c));
		}
	}
	unsigned long long t1 = clock64();
	if (threadIdx.x % 64 == 0)
	{
		out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
	}
	sink[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e + p.x + p.y + r.x + s.x + q.x + __uint_as_float(idx);
}

template <int MODE> static void run(const char* name, int threads, int perIter)
{
	unsigned long long* out;
	float* sink;
	hipMalloc(&out, 64 * 16 * sizeof(unsigned long long));
	hipMalloc(&sink, 64 * 1024 * sizeof(float));
	const int iters = 20;
	probe<MODE><<<64, threads>>>(out, sink, 2);
	probe<MODE><<<64, threads>>>(out, sink, iters);
	hipDeviceSynchronize();
	std::vector<unsigned long long> h(64 * 16);
	hipMemcpy(h.data(), out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
	double sum = 0;
	int n = 0;
	for (int b = 0; b < 64; ++b)
	{
		for (int w = 0; w < threads / 64; ++w)
		{
			sum += (double)h[(size_t)b * 16 + w];
			n += 1;
		}
	}
	// clock64 = s_memtime: 100 MHz constant clock on gfx9?  report raw ticks per instruction AND ns via wall_clock
	printf("%-58s threads %4d  ticks/instr %.3f\n", name, threads, sum / n / ((double)iters * perIter));
	hipFree(out);
	hipFree(sink);
}

int main()
{
	int rate = 0;
	hipDeviceGetAttribute(&rate, hipDeviceAttributeClockRate, 0);
	int wall = 0;
	hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
	printf("clock rate %d kHz, wall clock rate %d kHz (clock64 ticks are shader cycles)\n", rate, wall);
	for (int threads : {256, 512, 1024})
	{
		run<0>("v_mul_f32 dependent chain", threads, REP);
		run<1>("v_mul_f32 4 independent chains", threads, REP * 4);
		run<2>("v_pk_mul_f32 dependent chain", threads, REP);
		run<3>("v_pk_mul_f32 4 independent chains", threads, REP * 4);
		run<4>("xor + s_nop1 + mov_dpp + add (per group of 4)", threads, REP);
		run<5>("xor + s_nop1 + add_dpp (per group of 3)", threads, REP);
		run<9>("v_cmp + v_cndmask dependent (per pair)", threads, REP);
		run<10>("v_max + v_min dependent (per pair)", threads, REP);
		run<6>("ds_read_b32 dependent round trip", threads, 64);
		run<7>("s_barrier", threads, 64);
		run<8>("ds_read + add + ds_write + barrier", threads, 64);
	}
	return 0;
}
