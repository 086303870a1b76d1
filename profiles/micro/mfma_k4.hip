// mfma_k4.hip -- v_mfma_f32_16x16x4_f32 on gfx950: which lane holds what, and in which order (with which roundings) does one
// instruction add its four products to the accumulator?  A K = 4 fold (four alias rows per instruction, a quarter of the accumulator
// traffic of the 16x16x1_4B form) is only usable if a plain-VALU kernel and the 4x4x1 form (K = 1) can reproduce its sums bit for bit.
//   hypothesis: A lane i + 16 k = A[i][k], B lane j + 16 k = B[k][j], D[i][j] = VGPR (i & 3) of lane 16 (i >> 2) + j
//   candidates for the arithmetic: fma chain k = 0..3, fma chain k = 3..0, pairwise, exact sum rounded once
//   hipcc --offload-arch=gfx950 -O2 mfma_k4.hip -o mfma_k4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const float *a, const float *b, const float *c, float *d)
{
	const int l = threadIdx.x;
	v4f acc = { c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3] };
	acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
	for (int i = 0; i < 4; i++) d[l * 4 + i] = acc[i];
}

__global__ void rate(float *sink, int n)
{
	v4f acc[16];
	for (int i = 0; i < 16; i++) acc[i] = (v4f)(0.f);
	float a = (float)threadIdx.x, b = 1.0f / (1 + threadIdx.x);
	for (int k = 0; k < n; k++) {
#pragma unroll
		for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
	}
	float s = 0;
	for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
	if (s == 1.2345f) *sink = s;
}

static float frand(int spread)
{
	const float m = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
	return ldexpf(m, rand() % (2 * spread + 1) - spread);
}

int main()
{
	float ha[64], hb[64], hc[256], hd[256];
	float *da, *db, *dc, *dd;
	hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(hc)); hipMalloc(&dd, sizeof(hd));
	auto run = [&]() {
		hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice); hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice);
		hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
		hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
	};
	// 1. layout with small integers (every sum exact): A[i][k] = 1 + i + 16 k, B[k][j] = 1000^k-ish distinct weights
	for (int l = 0; l < 64; l++) { ha[l] = (float)(1 + l); hb[l] = (float)((l % 16 + 1) * ((l / 16) == 0 ? 1 : (l / 16) == 1 ? 64 : (l / 16) == 2 ? 4096 : 262144) % 8191 + l); }
	for (int i = 0; i < 256; i++) hc[i] = 0.f;
	run();
	int bad = 0;
	for (int i = 0; i < 16; i++)
		for (int j = 0; j < 16; j++) {
			float want = 0;
			for (int k = 0; k < 4; k++) want += ha[i + 16 * k] * hb[j + 16 * k];
			const float got = hd[(16 * (i >> 2) + j) * 4 + (i & 3)];
			if (got != want) { bad++; if (bad <= 6) printf("layout: i %d j %d got %.1f want %.1f\n", i, j, got, want); }
		}
	printf(bad ? "16x16x4 layout MISMATCH in %d of 256\n" : "16x16x4 layout ok: A lane i + 16 k, B lane j + 16 k, D[i][j] = vgpr (i & 3) of lane 16 (i >> 2) + j\n", bad);
	// 2. the arithmetic: random operands over a wide range of exponents, a non-zero accumulator
	const char *names[] = { "fma chain k = 0, 1, 2, 3", "fma chain k = 3, 2, 1, 0", "pairwise fma((0,1) + (2,3))", "exact sum rounded once", "products rounded, added in order (no fma)" };
	long match[5] = { 0, 0, 0, 0, 0 }, total = 0;
	srand(7);
	for (int trial = 0; trial < 200; trial++) {
		const int spread = trial < 100 ? 2 : 12;
		for (int l = 0; l < 64; l++) { ha[l] = frand(spread); hb[l] = frand(spread); }
		for (int i = 0; i < 256; i++) hc[i] = frand(spread);
		run();
		for (int i = 0; i < 16; i++)
			for (int j = 0; j < 16; j++) {
				const int slot = (16 * (i >> 2) + j) * 4 + (i & 3);
				const float c = hc[slot], got = hd[slot];
				float a[4], b[4];
				for (int k = 0; k < 4; k++) { a[k] = ha[i + 16 * k]; b[k] = hb[j + 16 * k]; }
				const float fwd = fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], fmaf(a[0], b[0], c))));
				const float rev = fmaf(a[0], b[0], fmaf(a[1], b[1], fmaf(a[2], b[2], fmaf(a[3], b[3], c))));
				const float pw = fmaf(a[1], b[1], a[0] * b[0]) + fmaf(a[3], b[3], a[2] * b[2]) + c;
				__float128 ex = (__float128)c;
				for (int k = 0; k < 4; k++) ex += (__float128)a[k] * (__float128)b[k];
				const float once = (float)ex;
				float plain = c;
				for (int k = 0; k < 4; k++) { volatile float p = a[k] * b[k]; plain = plain + p; }
				const float cand[5] = { fwd, rev, pw, once, plain };
				for (int m = 0; m < 5; m++) match[m] += cand[m] == got;
				total++;
				if (trial == 150 && i == 3 && j < 3) printf("  sample: got %.9g | fwd %.9g rev %.9g pairwise %.9g exact %.9g plain %.9g\n", got, fwd, rev, pw, once, plain);
			}
	}
	for (int m = 0; m < 5; m++) printf("%-44s: %ld of %ld bit-identical\n", names[m], match[m], total);
	// 3. subnormal products
	for (int l = 0; l < 64; l++) { ha[l] = 1e-20f; hb[l] = 1e-20f; }
	for (int i = 0; i < 256; i++) hc[i] = 2e-39f;
	run();
	{
		float w = 2e-39f;
		for (int k = 0; k < 4; k++) w = fmaf(1e-20f, 1e-20f, w);
		printf("subnormal accumulate: mfma gives %.9g, the fma chain %.9g (%s)\n", hd[0], w, hd[0] == w ? "same" : "DIFFERENT");
	}
	// 4. issue rate
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
			const double inst = (double)n * 16 * 4 * 256 * waves;
			printf("16x16x4 rate, %d wave(s)/SIMD: %.1f G wave-inst/s = %.1f TFLOP/s (%.2f ms); at 2.4 GHz that is %.1f cycles per instruction and SIMD\n",
					waves, inst / ms / 1e6, inst * 2048 / ms / 1e9, ms, 2.4e9 * 1024 * (ms * 1e-3) / inst);
		}
	}
	return 0;
}
