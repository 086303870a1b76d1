// mfma_layout.hip -- which lane holds what in v_mfma_f32_4x4x1_16B_f32 on gfx950, and is one instruction an exact fmaf?
//
// The batched fold (dumphfdl_amd/csrc/fold_kernels.hip, fold_mfma_kernel) assumes:
//   A (4x1 per block, 16 blocks): lane 4*blk + i holds A[i]        B (1x4): lane 4*blk + j holds B[j]
//   D (4x4):                      VGPR i of lane 4*blk + j holds D[i][j]
// and that D = fma(A, B, C) with a single rounding (the FMA chain of the plain-VALU reference kernel, bit for bit).
// Prints "layout ok" / "fma exact" or the mapping it found.      hipcc --offload-arch=gfx950 -O2 mfma_layout.hip -o mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const float *a, const float *b, const float *c, float *d)
{
	const int l = threadIdx.x;
	v4f acc = { c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3] };
	acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
	for (int i = 0; i < 4; i++) d[l * 4 + i] = acc[i];
}

typedef float v16f __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x1_4B_f32: four independent 16x16 outer products per instruction
__global__ void probe16(const float *a, const float *b, float *d)
{
	const int l = threadIdx.x;
	v16f acc;
	for (int i = 0; i < 16; i++) acc[i] = 0.f;
	acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[l], b[l], acc, 0, 0, 0);
	for (int i = 0; i < 16; i++) d[l * 16 + i] = acc[i];
}

__global__ void rate16(float *sink, int n)
{
	v16f acc[4];
	for (int i = 0; i < 4; i++) for (int k = 0; k < 16; k++) acc[i][k] = 0.f;
	float a = (float)threadIdx.x, b = 1.0f / (1 + threadIdx.x);
	for (int k = 0; k < n; k++) {
#pragma unroll
		for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[i], 0, 0, 0);
	}
	float s = 0;
	for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][15];
	if (s == 1.2345f) *sink = s;
}

// throughput: `n` dependent-free instructions per wave on 16 accumulators
__global__ void rate(float *sink, int n)
{
	v4f acc[16];
	for (int i = 0; i < 16; i++) acc[i] = (v4f)(0.f);
	float a = (float)threadIdx.x, b = 1.0f / (1 + threadIdx.x);
	for (int k = 0; k < n; k++) {
#pragma unroll
		for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
	}
	float s = 0;
	for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
	if (s == 1.2345f) *sink = s;
}

int main()
{
	float ha[64], hb[64], hc[256], hd[256];
	// distinct, rounding-sensitive values
	for (int l = 0; l < 64; l++) { ha[l] = 1.0f + (float)l * 0.37f + 1e-4f * l * l; hb[l] = 3.0f - (float)l * 0.113f; }
	for (int i = 0; i < 256; i++) hc[i] = 0.001f * (float)i + 1e3f;
	float *da, *db, *dc, *dd;
	hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(hc)); hipMalloc(&dd, sizeof(hd));
	hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice); hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
	hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
	int bad = 0, inexact = 0;
	for (int blk = 0; blk < 16; blk++)
		for (int i = 0; i < 4; i++)
			for (int j = 0; j < 4; j++) {
				const float got = hd[(4 * blk + j) * 4 + i];
				const float want = fmaf(ha[4 * blk + i], hb[4 * blk + j], hc[(4 * blk + j) * 4 + i]);
				if (got != want) {
					bad++;
					const float unfused = ha[4 * blk + i] * hb[4 * blk + j] + hc[(4 * blk + j) * 4 + i];
					if (std::fabs(got - want) < 1e-3f * std::fabs(want)) inexact++;
					if (bad <= 8) printf("blk %d i %d j %d: got %.9g want(fma) %.9g unfused %.9g\n", blk, i, j, got, want, unfused);
				}
			}
	if (!bad) printf("layout ok: A lane 4*blk+i, B lane 4*blk+j, D vgpr i lane 4*blk+j; fma exact (256 / 256 bit-identical to fmaf)\n");
	else printf("MISMATCH: %d of 256 differ (%d of them close: rounding, not layout)\n", bad, inexact);
	// denormal behaviour: a product that is subnormal
	{
		float ta[64], tb[64], tc[256];
		for (int l = 0; l < 64; l++) { ta[l] = 1e-20f; tb[l] = 1e-20f; }
		for (int i = 0; i < 256; i++) tc[i] = 2e-39f;
		hipMemcpy(da, ta, sizeof(ta), hipMemcpyHostToDevice); hipMemcpy(db, tb, sizeof(tb), hipMemcpyHostToDevice); hipMemcpy(dc, tc, sizeof(tc), hipMemcpyHostToDevice);
		hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
		hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
		printf("subnormal accumulate: mfma gives %.9g, fmaf gives %.9g (%s)\n", hd[0], fmaf(1e-20f, 1e-20f, 2e-39f), hd[0] == fmaf(1e-20f, 1e-20f, 2e-39f) ? "same" : "MFMA flushes");
	}
	// 16x16x1 4B: hypothesis A lane i + 16 blk, B lane j + 16 blk, D[blk][i][j] in vgpr 4 blk + (i & 3) of lane 16 (i >> 2) + j
	{
		float ta[64], tb[64], td[1024];
		for (int l = 0; l < 64; l++) { ta[l] = 1.0f + (float)l; tb[l] = 100.0f + 3.0f * (float)l; }
		float *dd16;
		hipMalloc(&dd16, sizeof(td));
		hipMemcpy(da, ta, sizeof(ta), hipMemcpyHostToDevice); hipMemcpy(db, tb, sizeof(tb), hipMemcpyHostToDevice);
		hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, da, db, dd16);
		hipMemcpy(td, dd16, sizeof(td), hipMemcpyDeviceToHost);
		int bad16 = 0;
		for (int blk = 0; blk < 4; blk++)
			for (int i = 0; i < 16; i++)
				for (int j = 0; j < 16; j++) {
					const float got = td[(16 * (i >> 2) + j) * 16 + 4 * blk + (i & 3)], want = ta[i + 16 * blk] * tb[j + 16 * blk];
					if (got != want) { bad16++; if (bad16 <= 6) printf("16x16x1 blk %d i %d j %d: got %.1f want %.1f\n", blk, i, j, got, want); }
				}
		if (!bad16) printf("16x16x1_4B layout ok: A lane i + 16 blk, B lane j + 16 blk, D[blk][i][j] = vgpr 4 blk + (i & 3) of lane 16 (i >> 2) + j\n");
		else {
			printf("16x16x1_4B MISMATCH in %d of 1024; decoding the real layout:\n", bad16);
			for (int v = 0; v < 16; v += 5)
				for (int l = 0; l < 64; l += 21) {
					const float got = td[l * 16 + v];
					for (int blk = 0; blk < 4; blk++) for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++)
						if (got == ta[i + 16 * blk] * tb[j + 16 * blk]) printf("  vgpr %d lane %d holds blk %d i %d j %d (if A lane i+16blk, B lane j+16blk)\n", v, l, blk, i, j);
				}
		}
	}
	{
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		const int n = 4096;
		for (int waves = 1; waves <= 2; waves++) {
			hipLaunchKernelGGL(rate16, dim3(256 * waves), dim3(256), 0, 0, dd, 16);
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(rate16, dim3(256 * waves), dim3(256), 0, 0, dd, n);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			const double inst = (double)n * 4 * 4 * 256 * waves;
			printf("16x16x1_4B rate, %d wave(s)/SIMD: %.1f G wave-inst/s = %.1f TFLOP/s (%.2f ms); at 2.4 GHz that is %.1f cycles per instruction and SIMD\n",
					waves, inst / ms / 1e6, inst * 2048 / ms / 1e9, ms, 2.4e9 * 1024 * (ms * 1e-3) / inst);
		}
	}
	// issue rate: one workgroup of 4 waves per CU
	{
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		const int n = 4096;
		for (int waves = 1; waves <= 2; waves++) {
			hipLaunchKernelGGL(rate, dim3(256 * waves), dim3(256), 0, 0, dd, 16);
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(rate, dim3(256 * waves), dim3(256), 0, 0, dd, n);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			const double inst = (double)n * 16 * 4 * 256 * waves;            // wave-instructions
			printf("4x4x1 rate, %d wave(s)/SIMD: %.1f G wave-inst/s = %.1f TFLOP/s (%.2f ms); at 2.4 GHz that is %.1f cycles per instruction and SIMD\n",
					waves, inst / ms / 1e6, inst * 512 / ms / 1e9, ms, 2.4e9 * 1024 * (ms * 1e-3) / inst);
		}
	}
	return bad ? 1 : 0;
}
