// fft_regs.h -- register-resident radix-16 x radix-B column FFT for the wideband forward passes (gfx950).
//
// A pass transforms tiles of 16 adjacent columns x R = 16 B points (B = 4, 8, 16: R = 64, 128, 256).  Round 1-4's passes ran four
// radix-4 stages through LDS with a barrier each and an exact sincospif per output element: ~130 vector instructions per element,
// which is what bounded them (0.44 - 0.54 of the HBM peak, VALU-bound beside the demodulators).  Here a thread keeps 16 points of one
// column in registers:
//   stage A  thread (n2, col): x[n1] = in[n2 + B n1], 16-point DFT over n1 in registers, times W_R^(n2 k1), to LDS
//   stage B  thread (k1, col): z[n2] from LDS, B-point DFT over n2 in registers: X[k1 + 16 k2]
// one LDS exchange, one barrier, no bit reversal; the inter-pass twiddles W^(k c) of a thread's outputs k = k1 + 16 k2 come from
// three exact sincospif (base, step, step^4) and a short chain of complex multiplies (depth <= 6).
#pragma once
#include "kernels.h"
#include "fft_core.h"

namespace hfdl {

constexpr int FR_PITCH = 16 * 16 + 16;         // cf32 per n2 plane of the exchange buffer: [k1][col] + 128 B, so that two planes written by one
                                               // 64-bit LDS instruction (32 lanes per clock) fall on disjoint bank halves

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mj(float2 a) { return make_float2(a.y, -a.x); }       // a * (-j)

// forward 4-point DFT in place: (a, b, c, d) -> (X0, X1, X2, X3)
__device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d)
{
	const float2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = mul_mj(csub(b, d));
	a = cadd(t0, t2); c = csub(t0, t2);
	b = cadd(t1, t3); d = csub(t1, t3);
}

// forward 16-point DFT in place; X[k] ends up in slot 4 (k & 3) + (k >> 2)
__device__ __forceinline__ void dft16(float2 (&x)[16])
{
	constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;      // cos, sin of pi/8; sqrt(1/2)
#pragma unroll
	for (int b = 0; b < 4; b++) dft4(x[b], x[4 + b], x[8 + b], x[12 + b]);          // over a of x[4 a + b] -> t[b][c] in x[4 c + b]
	// t[b][c] *= W16^(b c)
	x[5] = cmul(x[5], make_float2(C1, -S1));  x[6] = cmul(x[6], make_float2(H, -H));    x[7] = cmul(x[7], make_float2(S1, -C1));
	x[9] = cmul(x[9], make_float2(H, -H));    x[10] = mul_mj(x[10]);                     x[11] = cmul(x[11], make_float2(-H, -H));
	x[13] = cmul(x[13], make_float2(S1, -C1)); x[14] = cmul(x[14], make_float2(-H, -H)); x[15] = cmul(x[15], make_float2(-C1, S1));
#pragma unroll
	for (int c = 0; c < 4; c++) dft4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);     // over b -> X[c + 4 d] in x[4 c + d]
}
__device__ __forceinline__ constexpr int slot16(int k) { return 4 * (k & 3) + (k >> 2); }

// forward 8-point DFT in place; X[k] ends up in slot 4 (k & 1) + (k >> 1)
__device__ __forceinline__ void dft8(float2 (&x)[8])
{
	constexpr float H = 0.70710678118654752f;
#pragma unroll
	for (int b = 0; b < 4; b++) { const float2 s = cadd(x[b], x[4 + b]), d = csub(x[b], x[4 + b]); x[b] = s; x[4 + b] = d; }
	x[5] = cmul(x[5], make_float2(H, -H)); x[6] = mul_mj(x[6]); x[7] = cmul(x[7], make_float2(-H, -H));
	dft4(x[0], x[1], x[2], x[3]);
	dft4(x[4], x[5], x[6], x[7]);
}
__device__ __forceinline__ constexpr int slot8(int k) { return 4 * (k & 1) + (k >> 1); }

template <int B> __device__ __forceinline__ void dft_small(float2 (&z)[B])
{
	if constexpr (B == 16) dft16(z);
	else if constexpr (B == 8) dft8(z);
	else { static_assert(B == 4, "B = 4, 8 or 16"); dft4(z[0], z[1], z[2], z[3]); }
}
template <int B> __device__ __forceinline__ constexpr int slot_small(int k) { return B == 16 ? slot16(k) : B == 8 ? slot8(k) : k; }

// stage A + exchange: x[n1] (n1 = 0..15) of column `col`, sub-transform n2 -> LDS plane n2, row k1, column col, times W_R^(n2 k1)
__device__ __forceinline__ void stage_a(float2 (&x)[16], float2 *ex, const float2 *ltw, int n2, int col)
{
	dft16(x);
	float2 *dst = ex + n2 * FR_PITCH + col;
	dst[0] = x[0];
#pragma unroll
	for (int k1 = 1; k1 < 16; k1++) dst[k1 * 16] = cmul(x[slot16(k1)], ltw[n2 * k1]);
}

// stage B of task (k1, col): B points from the exchange buffer, B-point DFT; X[k1 + 16 k2] is z[slot_small<B>(k2)]
template <int B> __device__ __forceinline__ void stage_b(float2 (&z)[B], const float2 *ex, int k1, int col)
{
	const float2 *src = ex + k1 * 16 + col;
#pragma unroll
	for (int n2 = 0; n2 < B; n2++) z[n2] = src[n2 * FR_PITCH];
	dft_small<B>(z);
}

// w[k2] = exp(-2 pi i (e0 + k2 es) / 2^logn), k2 < B: e0, es already reduced mod 2^logn.  Three exact evaluations (base, step, step^4)
// and products of at most three factors of each: the phase error stays below ~7 roundings
template <int B> __device__ __forceinline__ void twiddle_run(float2 (&w)[B], unsigned e0, unsigned es, int logn)
{
	const unsigned mask = (1u << logn) - 1u;
	const float2 base = unit_twiddle(e0 & mask, logn), s1 = unit_twiddle(es & mask, logn);
	const float2 s2 = cmul(s1, s1), s3 = cmul(s2, s1);
	float2 s4 = make_float2(1.f, 0.f);
	if constexpr (B > 4) s4 = unit_twiddle((4u * es) & mask, logn);
	float2 q = base;
#pragma unroll
	for (int g = 0; g < B / 4; g++) {
		if (g > 0) q = cmul(q, s4);
		w[4 * g] = q; w[4 * g + 1] = cmul(q, s1); w[4 * g + 2] = cmul(q, s2); w[4 * g + 3] = cmul(q, s3);
	}
}

}  // namespace hfdl
