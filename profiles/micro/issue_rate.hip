// issue_rate.hip -- how fast does ONE wavefront issue instructions on gfx950?  (hipcc --offload-arch=gfx950 -O2 issue_rate.hip -o issue_rate)
// Each test is a loop of 64 instructions repeated 4096 times by a single wave on an otherwise idle GPU; s_memtime around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__global__ void k(unsigned long long *out, float seed)
{
	float a = seed, b = seed + 1, c = seed + 2, d = seed + 3;
	int si;
	asm volatile("s_mov_b32 %0, 0" : "=s"(si));
	unsigned long long t0, t1;
	// (0) dependent v_add_f32 chain
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP64(asm volatile("v_add_f32 %0, %0, %0" : "+v"(a));) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[0] = t1 - t0;
	// (1) four independent v_add_f32 chains interleaved
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP16(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[1] = t1 - t0;
	// (2) dependent s_add_u32 chain
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP64(asm volatile("s_add_u32 %0, %0, 1" : "+s"(si) : : "scc");) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[2] = t1 - t0;
	// (3) v_readlane -> v_mov (VALU -> SGPR -> VALU ping-pong, dependent)
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP16(asm volatile("v_readlane_b32 %1, %0, 3\n s_nop 3\n v_mov_b32 %0, %1\n v_add_f32 %0, %0, %0" : "+v"(a), "+s"(si));) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[3] = t1 - t0;      // 16 x 4 = 64 instructions incl. s_nop
	// (4) dependent DPP row_shr adds
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP16(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[4] = t1 - t0;      // 16 x 4 = 64 instructions incl. s_nop
	// (5) dependent v_sin_f32 chain
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP64(asm volatile("v_sin_f32 %0, %0" : "+v"(a));) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[5] = t1 - t0;
	// (6) dependent LDS round trip: ds_write + ds_read + wait
	__shared__ float sm[64];
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP16(sm[threadIdx.x] = a; __builtin_amdgcn_s_waitcnt(0xc07f); a = ((volatile float *)sm)[threadIdx.x ^ 1]; asm volatile("" : "+v"(a));) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[6] = t1 - t0;      // 16 round trips per iteration
	// (7) alternating SALU / VALU independent
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP16(asm volatile("s_add_u32 %1, %1, 1\n v_add_f32 %0, %0, %0\n s_add_u32 %1, %1, 1\n v_add_f32 %2, %2, %2" : "+v"(a), "+s"(si), "+v"(b) : : "scc");) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[7] = t1 - t0;
	// (8) taken forward scalar branch over one instruction: s_cmp + s_cbranch_scc1 (2 instructions issued per group)
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP16(asm volatile("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n v_add_f32 %1, %1, %1\n1:\n v_add_f32 %2, %2, %2\n s_add_u32 %0, %0, 1" : "+s"(si), "+v"(a), "+v"(b) : : "scc");) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[9] = t1 - t0;      // 16 groups of 4 issued instructions
	// (9) the same with the branch NOT taken (5 instructions issued per group)
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP16(asm volatile("s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1f\n v_add_f32 %1, %1, %1\n1:\n v_add_f32 %2, %2, %2\n s_add_u32 %0, %0, 1" : "+s"(si), "+v"(a), "+v"(b) : : "scc");) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[10] = t1 - t0;
	// (10) the branch-free form of the same choice: v_cmp + v_cndmask on the value (4 instructions)
	t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < 4096; i++) { REP16(asm volatile("v_add_f32 %3, %1, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %3, vcc\n v_add_f32 %2, %2, %2" : "+s"(si), "+v"(a), "+v"(b), "+v"(c) : : "vcc");) }
	t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[11] = t1 - t0;
	if (threadIdx.x == 0) { int sv; asm volatile("v_mov_b32 %0, %1" : "=v"(sv) : "s"(si)); out[8] = (unsigned long long)(a + b + c + d) + (unsigned)sv; }
}
int main()
{
	unsigned long long *d, h[12];
	(void)hipMalloc(&d, sizeof(h));
	for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.0f); (void)hipDeviceSynchronize(); }
	(void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	const char *names[8] = { "dependent v_add_f32", "4 independent v_add_f32 chains", "dependent s_add_u32", "v_readlane -> s_nop 3 -> v_mov -> v_add (4 instr group)",
		"dependent DPP adds with s_nop 1 (per pair of instr)", "dependent v_sin_f32", "LDS write->read round trip (per trip, /16)", "alternating independent SALU / VALU" };
	for (int i = 0; i < 8; i++) printf("%-62s %8.2f cycles per instruction (s_memtime ticks)\n", names[i], (double)h[i] / (4096.0 * (i == 6 ? 16 : 64)));
	printf("%-62s %8.2f cycles per group (cmp, branch taken over 1 instr, v_add, s_add)\n", "taken forward s_cbranch_scc1", (double)h[9] / (4096.0 * 16));
	printf("%-62s %8.2f cycles per group (cmp, branch not taken, 2 v_add, s_add)\n", "not-taken s_cbranch_scc1", (double)h[10] / (4096.0 * 16));
	printf("%-62s %8.2f cycles per group (v_add, v_cmp, v_cndmask, v_add)\n", "branch-free select", (double)h[11] / (4096.0 * 16));
	return 0;
}
