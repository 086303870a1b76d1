// neighbour.hip -- which of the fold's resources slows the demodulator's waves down?  (laboratory; profiles/neighbour_probe.py)
//
// The demodulator kernel executes up to twice the shader cycles when its three waves share a CU with the fold's four (profiles/r06/
// clock_probe_cfg3.md: cycles counted inside the kernel, the clock is NOT what moves).  This file holds a synthetic neighbour that uses
// ONE of the fold's resources at a time, at about the fold's rate -- one "quad" of work per ~2048 cycles and wave, four waves per
// workgroup, one workgroup per CU (84 KiB of LDS keeps a second one out and leaves the demodulator its 59 KiB) -- so that the demodulator
// can be timed beside each:
//   bit 0  matrix pipe   64 v_mfma_f32_16x16x4_f32 per quad on 16 independent accumulators (the fold's 32-column form: the pipe full)
//   bit 1  LDS reads     16 ds_read_b128 per quad and wave (operand B)
//   bit 2  LDS writes    4 ds_write_b128 per quad and wave, the fold's own addresses (pitch of 68 / 132 entries: parts 4-way on a bank)
//   bit 3  barrier       one s_barrier per quad
//   bit 4  HBM reads     5 KiB per quad and wave of non-temporal 16-byte loads (4 KiB of taps + the spectrum share)
//   bit 5  vector ALU    16 DPP moves per quad (the rotated operand)
//   3000           v_mfma_f32_16x16x32_bf16 back to back (the bf16 matrix pipe: what an exact multi-term split of the fp32 product would run on)
//   2000           the matrix pipe alone on v_mfma_f32_4x4x1_16B_f32: the same multiply-accumulates in instructions of 2 passes instead of 8
//   1000 + n       the matrix pipe alone with n idle cycles (s_nop) of the issuing wave behind every instruction
//   bit 6  no pacing     without bit 0 a quad is otherwise padded to ~2048 cycles by s_sleep; with bit 6 it runs flat out
// A workgroup lives for `ticks` of the 100 MHz s_memrealtime counter, so a launch ends by itself.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC neighbour.hip -o libneighbour.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float v4f __attribute__((ext_vector_type(4)));

// a 16-byte LDS read the compiler can neither narrow nor drop
__device__ __forceinline__ v4f lds_read128(const v4f *p)
{
	v4f v;
	asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)p) : "memory");
	return v;
}

// one matrix instruction followed by NOPS idle cycles of THIS wave (s_nop: the sequencer waits, no issue port is held)
template <int NOPS>
__device__ __forceinline__ void mfma_paced(v4f &acc, float a, float b)
{
	if constexpr (NOPS == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
	else if constexpr (NOPS == 8) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n s_nop 7" : "+v"(acc) : "v"(a), "v"(b));
	else if constexpr (NOPS == 16) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n s_nop 15" : "+v"(acc) : "v"(a), "v"(b));
	else if constexpr (NOPS == 20) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n s_nop 15\n s_nop 3" : "+v"(acc) : "v"(a), "v"(b));
	else if constexpr (NOPS == 24) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n s_nop 15\n s_nop 7" : "+v"(acc) : "v"(a), "v"(b));
	else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n s_nop 15\n s_nop 11" : "+v"(acc) : "v"(a), "v"(b));
}

template <int MODE, int NOPS = -1>
__global__ __launch_bounds__(256) void neighbour_kernel(const v4f *__restrict__ src, size_t src_items, unsigned long long ticks, float *sink, unsigned *quads_done)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	v4f *xt = (v4f *)lds_raw;                                   // [2][8][132]
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	v4f acc[16];
#pragma unroll
	for (int j = 0; j < 16; j++) acc[j] = v4f{ 0.f, 0.f, 0.f, 0.f };
	v4f x[2] = { v4f{ 1.f, 2.f, 3.f, 4.f }, v4f{ 0.5f, 0.25f, 0.125f, 1.f } };
	v4f h = v4f{ (float)lane, 1.f, 2.f, 3.f };
	// this wave's window of the source: 5 loads of 1 KiB per quad, wave-contiguous, walking a private stripe
	const size_t stripe = src_items / ((size_t)gridDim.x * 4);
	const v4f *p = src + ((size_t)blockIdx.x * 4 + wave) * stripe + lane;
	size_t off = 0;
	float vsum = 0.f;
	int stage = 0;
	unsigned it = 0;
	for (;; it++) {
		if ((it & 15) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > ticks) break;
		if constexpr (MODE & 16) {
			v4f ld[5];
#pragma unroll
			for (int i = 0; i < 5; i++) ld[i] = __builtin_nontemporal_load(p + off + 64 * i);
			off += 320;
			if (off + 320 > stripe) off = 0;
#pragma unroll
			for (int i = 0; i < 5; i++) h += ld[i];
		}
		if constexpr (MODE & 4) {
			// the fold's stash: item = (wave * 4 + i) * 64 + lane -> segment item >> 3, part item & 7 at part * 132 + segment
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int item = (wave * 4 + i) * 64 + lane, seg = item >> 3, part = item & 7;
				xt[stage * 8 * 132 + part * 132 + seg] = h;
			}
		}
		if constexpr (MODE & 8) __syncthreads();
		if constexpr (MODE & 2) {
#pragma unroll
			for (int b = 0; b < 8; b++) {
				x[0] = lds_read128(xt + stage * 8 * 132 + b * 132 + lane);
				x[1] = lds_read128(xt + stage * 8 * 132 + b * 132 + 64 + lane);
				asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]) : : "memory");
				if constexpr (!(MODE & 1)) { vsum += x[0][0] + x[0][3] + x[1][1] + x[1][2]; }
				else {
#pragma unroll
					for (int q = 0; q < 2; q++) {
						acc[2 * b + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[q], x[0][q], acc[2 * b + q], 0, 0, 0);
						acc[(2 * b + q + 8) & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[q + 2], x[1][q], acc[(2 * b + q + 8) & 15], 0, 0, 0);
						acc[2 * b + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[q], x[0][q + 2], acc[2 * b + q], 0, 0, 0);
						acc[(2 * b + q + 8) & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[q + 2], x[1][q + 2], acc[(2 * b + q + 8) & 15], 0, 0, 0);
					}
				}
			}
		} else if constexpr (MODE & 1) {
#pragma unroll
			for (int r = 0; r < 4; r++)
#pragma unroll
				for (int j = 0; j < 16; j++) {
					if constexpr (NOPS == -3) {        // the bf16 matrix instruction (16 x the multiply-accumulates per instruction): does IT leave the vector issue alone?
						typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
						v8bf a8, b8;
#pragma unroll
						for (int e = 0; e < 8; e++) { a8[e] = (__bf16)h[e & 3]; b8[e] = (__bf16)x[j & 1][e & 3]; }
						acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[j], 0, 0, 0);
					} else if constexpr (NOPS == -2) {        // the same multiply-accumulates in the short form: four v_mfma_f32_4x4x1_16B_f32 (2 passes each) per 16x16x4 (8 passes)
#pragma unroll
						for (int u = 0; u < 4; u++) acc[(j + 4 * u) & 15] = __builtin_amdgcn_mfma_f32_4x4x1f32(h[(j + u) & 3], x[j & 1][r], acc[(j + 4 * u) & 15], 0, 0, 0);
					} else if constexpr (NOPS >= 0) mfma_paced<NOPS>(acc[j], h[j & 3], x[j & 1][r]);
					else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[j & 3], x[j & 1][r], acc[j], 0, 0, 0);
				}
		}
		if constexpr (MODE & 32) {
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const int v = __builtin_amdgcn_update_dpp(0, __float_as_int(h[i & 3]), 0xB1, 0xF, 0xF, true);
				h[i & 3] = __int_as_float(v ^ (int)0x80000000);
			}
		}
		if constexpr (!(MODE & 1) && !(MODE & 64)) __builtin_amdgcn_s_sleep(30);
		stage ^= 1;
	}
	float s = vsum;
#pragma unroll
	for (int j = 0; j < 16; j++) s += acc[j][0] + acc[j][3];
	s += h[0] + h[1] + h[2] + h[3];
	if (s == 1.2345e33f) *sink = s;
	if (threadIdx.x == 0) quads_done[blockIdx.x] = it;         // quads of work this workgroup's waves got through: the neighbour's own rate
}

static hipStream_t nb_stream;
static float *nb_sink;
static unsigned *nb_quads;          // [4096]
static int nb_groups;

template <int MODE, int NOPS = -1> static int go(const void *src, size_t bytes, double ms, int groups)
{
	constexpr unsigned LDS = 84 * 1024;      // two do not fit a CU (160 KiB); 76 KiB left for the demodulator (59) or a pass of the pipeline
	if (hipFuncSetAttribute((const void *)neighbour_kernel<MODE, NOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) != hipSuccess) return -2;
	nb_groups = groups;
	hipLaunchKernelGGL((neighbour_kernel<MODE, NOPS>), dim3((unsigned)groups), dim3(256), LDS, nb_stream, (const v4f *)src, bytes / 16, (unsigned long long)(ms * 1e5), nb_sink, nb_quads);
	return hipGetLastError() == hipSuccess ? 0 : -3;
}

// launch a neighbour of `mode` for `ms` milliseconds on a stream of its own: `groups` workgroups (256 = one per CU), reading `src`
extern "C" int neighbour_start(int mode, const void *src, size_t bytes, double ms, int groups)
{
	if (!nb_stream) {
		if (hipStreamCreateWithFlags(&nb_stream, hipStreamNonBlocking) != hipSuccess) return -1;
		if (hipMalloc(&nb_sink, 64) != hipSuccess) return -1;
		if (hipMalloc(&nb_quads, 4096 * sizeof(unsigned)) != hipSuccess) return -1;
	}
	if (groups < 1 || groups > 4096) return -5;
	if (ms > 2000.0) ms = 2000.0;
	switch (mode) {
#define M(v) case v: return go<v>(src, bytes, ms, groups);
	case 1000: return go<1, 0>(src, bytes, ms, groups);         // 1000 + n: the matrix pipe alone, n idle cycles behind every instruction
	case 1008: return go<1, 8>(src, bytes, ms, groups);
	case 1016: return go<1, 16>(src, bytes, ms, groups);
	case 1020: return go<1, 20>(src, bytes, ms, groups);
	case 1024: return go<1, 24>(src, bytes, ms, groups);
	case 1028: return go<1, 28>(src, bytes, ms, groups);
	case 2000: return go<1, -2>(src, bytes, ms, groups);
	case 3000: return go<1, -3>(src, bytes, ms, groups);        // v_mfma_f32_16x16x32_bf16 back to back        // the matrix pipe alone on the 4x4x1_16B form: 256 instructions of 2 passes per quad
	M(1) M(2) M(4) M(6) M(14) M(16) M(32) M(33) M(15) M(31) M(63) M(66) M(68) M(80) M(78) M(96)
#undef M
	}
	return -4;
}

// wait for the neighbour to end; returns the quads of work an average workgroup got through (-1: error)
extern "C" double neighbour_wait(void)
{
	if (!nb_stream) return 0.0;
	if (hipStreamSynchronize(nb_stream) != hipSuccess) return -1.0;
	static unsigned host[4096];
	if (hipMemcpy(host, nb_quads, sizeof(unsigned) * (size_t)nb_groups, hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
	double sum = 0;
	for (int i = 0; i < nb_groups; i++) sum += host[i];
	return sum / nb_groups;
}
