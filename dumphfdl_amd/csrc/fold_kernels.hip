// fold_kernels.hip -- per-channel spectrum x filter fold, inverse FFT, overlap scrap, NCO + decimate (gfx950).
//
// Replaces fastddc_inv_cc (reference src/fastddc.c:152-215): multiply_and_shift (:123-150), fft_swap_sides,
// the M-point backward FFT + normalisation (:190-197) and decimating_shift_addition_cc (src/libcsdr_gpl.c:41-74).
//
// The fold is THE roofline kernel: per channel it streams N cf32 filter taps (distinct per channel, 8*N bytes) against the shared
// N-bin spectrum and accumulates the N/M alias rows onto M bins:
//      Y_c[(h0 + j) mod M] = sum_a  H_c[a*M + j] * X[a*M + j]
// The taps are the same for every block, so one launch multiplies them into the spectra of up to 16 queued blocks -- and with more
// than one block the fold IS a small dense contraction per bin j:  Y[j] (channels x blocks) = H[j] (channels x alias rows) . X[j]
// (alias rows x blocks).  It runs on the fp32 matrix pipe -- v_mfma_f32_16x16x4_f32: per instruction one bin x eight channels' Re / Im
// rows x FOUR alias rows x sixteen blocks -- which takes the multiply-accumulates off the vector ALUs the demodulator kernel next door
// lives on.  Each product is one exact fmaf, applied in a fixed order (cmac_quad below).
#include <hip/hip_ext.h>
#include <type_traits>
#include "kernels.h"
#include "fft_core.h"

namespace hfdl {

constexpr int FOLD_THREADS = 256;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// ---- tap layouts: kernels.h (TAPL_*, tap_index_f) ----

__device__ __forceinline__ float2 tap_at(const float *taps, size_t row_stride_f, int m, int layout, int c, int row, int j)
{
	const float *p = taps + tap_index_f(layout, m, row_stride_f, c, row, j, 0);
	return make_float2(p[0], p[layout == TAPL_PLAIN ? 1 : 4]);
}

// The sum of a (block, channel, bin) is a FIXED chain of fused multiply-adds -- the order the matrix instruction applies them in:
// v_mfma_f32_16x16x4_f32 adds its four products to the accumulator one after the other, k = 0, 1, 2, 3, each an exact fmaf
// (profiles/micro/mfma_k4.hip: 51200 of 51200 bit-identical, subnormals included), and a group of four alias rows takes two
// instructions: first the four Re(X) products, then the four Im(X) products.  The plain-VALU reference spells exactly that out, so
// that it and every tiling round a sum alike: a block folded alone and the same block folded beside fifteen others give the same
// 32 bits (tests/test_gpu_parity.py::test_fold_batching_changes_nothing, ::test_fold_mfma_equals_fma_chain).
__device__ __forceinline__ void cmac_quad(float2 &a, const float2 *h, const float2 *x, int n)
{
	for (int k = 0; k < n; k++) { a.x = __builtin_fmaf(h[k].x, x[k].x, a.x); a.y = __builtin_fmaf(h[k].y, x[k].x, a.y); }
	for (int k = 0; k < n; k++) { a.x = __builtin_fmaf(-h[k].y, x[k].y, a.x); a.y = __builtin_fmaf(h[k].x, x[k].y, a.y); }
}

// reference / fallback: one thread per (channel, slice, bin), either tap layout, any geometry, `nb` blocks one after the other
__global__ __launch_bounds__(FOLD_THREADS) void fold_ref_kernel(const float *__restrict__ taps, const float2 *__restrict__ spec, float2 *__restrict__ partial,
		size_t row_stride_f, size_t spec_stride, size_t partial_stride, int m, int slices, int rows, int layout, int nb)
{
	const int s = blockIdx.x % slices, c = blockIdx.x / slices;
	for (int b0 = 0; b0 < nb; b0 += 4)              // four blocks per pass over the taps
		for (int j = threadIdx.x; j < m; j += FOLD_THREADS) {
			const float2 *sp = spec + (size_t)b0 * spec_stride + (size_t)s * rows * (size_t)m + j;
			float2 acc[4] = { make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f) };
			for (int r = 0; r < rows; r += 4) {
				const int n = rows - r < 4 ? rows - r : 4;
				float2 h[4], x[4];
				for (int k = 0; k < n; k++) h[k] = tap_at(taps, row_stride_f, m, layout, c, s * rows + r + k, j);
#pragma unroll
				for (int b = 0; b < 4; b++)
					if (b0 + b < nb) {
						for (int k = 0; k < n; k++) x[k] = sp[(size_t)b * spec_stride + (size_t)(r + k) * m];
						cmac_quad(acc[b], h, x, n);
					}
			}
#pragma unroll
			for (int k = 0; k < 4; k++)
				if (b0 + k < nb) partial[(size_t)(b0 + k) * partial_stride + ((size_t)c * slices + s) * (size_t)m + j] = acc[k];
		}
}

// lanes 4 b + i: i <-> i ^ 1 inside every quad, then the sign of the even lanes flipped: (Re c0, Im c0, Re c1, Im c1) -> (-Im c0, Re c0, -Im c1, Re c1)
__device__ __forceinline__ float rot90(float a, int sign_mask)
{
	const int v = __builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
	return __int_as_float(v ^ sign_mask);
}

#ifdef HFDL_LAB
// Laboratory build: the shader clock a fold launch ran at, from inside it (demod_kernels.hip has the demodulator's twin).  The workgroup
// in the middle of the grid notes s_memtime and s_memrealtime (100 MHz) around its whole life (1/16 of the launch): cycles / ticks x 100 MHz.
__device__ unsigned long long hfdl_fold_clk_probe[1024 * 4];
__device__ unsigned hfdl_fold_clk_probe_n;
#endif

// registers the tile asks for -> waves per SIMD told to the compiler (512 per lane and SIMD): left to itself it aims at 8 waves,
// squeezes the loop into 64 registers and gets there by loading, waiting, multiplying, loading again
constexpr int fold16_waves(int p, int w, int d, bool small = false, int cg = 1)
{
	const int loads = 8 * cg, mine = w >= loads ? 1 : loads / w;
	const int regs = (small ? 16 : 64) * p * cg + 16 * p * d + 4 * mine * d + 4 * mine + 8 + 28;
	// the four-column form never takes more than two waves of a SIMD: it is bound by the HBM reads, and a third wave would take the
	// registers the demodulator's waves need beside it (a 2-block demodulator launch beside a 3-wave 4-block fold: 1.7 ms instead of 0.5)
	if (small) return regs <= 168 ? 2 : 1;
	if (cg > 1) return 1;                  // 128 accumulation registers per octet: one wave per SIMD, its software pipeline keeps the matrix pipe fed (two would spill)
	return regs <= 96 ? 5 : regs <= 128 ? 4 : regs <= 168 ? 3 : regs <= 256 ? 2 : 1;
}

// THE fold: v_mfma_f32_16x16x4_f32 on TAPL_OCTET taps.  One instruction = one bin x (8 channels' Re / Im rows) x FOUR alias rows x 16
// blocks: 1024 multiply-accumulates with a quarter of the accumulator traffic of the one-row form (16x16x1_4B) of the first builds
// (profiles/r05_experiments.md: the same loop with this instruction in the place of the other ran 22 % faster before it was right).
// The fp32 matrix instruction runs at the vector ALUs' own rate (64 FLOP per clock and SIMD) and holds the SIMD's vector issue while it
// executes: a neighbour wave on the SIMD -- the demodulator's -- gets one instruction in per matrix instruction (measured with a
// synthetic neighbour, profiles/r06/neighbour_probe_cfg3.md: the demodulator takes 3.4 - 4.3 x its cycles beside back-to-back matrix
// instructions, 1.0 x beside LDS or HBM traffic alone).
// P channel OCTETS per wave, W waves per workgroup, D groups of four alias rows ("quads") of loads in flight.  A workgroup = one group
// of 16 bins x one slice of alias rows x 8 P W channels; its W waves cover the SAME bins and different channels, so each quad's spectrum
// tile (16 blocks x 4 rows x 16 bins = 8 KiB) is fetched ONCE per workgroup -- every wave a share of it, in whole 128-byte segments --
// written to LDS in operand-B order (lane n + 16 k <- block n, row k) and read from there by all waves.  Per quad and wave: P tap loads
// of 4 KiB (four consecutive KiB: the four bin quads of the tile; non-temporal), 8 LDS reads, 16 P vector instructions (the rotated
// operand) and 32 P matrix instructions: per bin the Re(X) product, then the Im(X) product.  Loads run D quads ahead, the spectrum
// tile one quad ahead through two LDS stages, one barrier per quad.  A launch takes any block count up to 16: columns past the last
// block repeat it and are never stored (their multiplies are done all the same: 1 block costs what 16 cost).
// WIN (the pruned fold, hfdl_gpu.h HFDL_GPU_FOLD_PRUNE): a workgroup folds only the window of quads `win[group]` = (first quad, count)
// around its channels' pass bands -- circular, one slice, rows = all alias rows -- instead of a slice of all of them.
// SMALL (launches of at most FOUR blocks: the ragged end of a run, a drained pipeline, a live receiver's block at a time): the sixteen
// columns of the instruction cost their time whatever the block count.  v_mfma_f32_4x4x1_16B_f32 -- sixteen 4 x 4 products, K = 1 --
// does four blocks at a quarter of the matrix time and leaves the launch to the HBM reads of the taps.  Its operand A wants the four
// lanes of a product to hold Re / Im of two channels at ONE (bin, alias row); the taps hold the four alias rows of a quad in the four
// 16-lane groups and the four bins in the four registers: two v_permlane32_swap + two v_permlane16_swap per KiB transpose groups
// against registers, after which register k = alias row k and the lane group = the bin.  Then k = 0 .. 3 with Re(X), k = 0 .. 3 with
// Im(X), one instruction each, on the same accumulator: the chain of the sixteen-column form, the same bits.
// CG (round 6: launches of 17 .. 32 blocks): column groups of sixteen blocks per pass over the taps.  A launch costs about its matrix
// time PLUS its memory time, not the larger of the two (alone: four columns 2.8 ms, sixteen 3.9, thirty-two 5.7 .. 6.3 for the same
// 16 GiB of taps): one wave per SIMD issues a matrix instruction every 40 cycles at best (32 of execution + 8 of issue,
// profiles/micro/neighbour.hip) and every wait at a quad's barrier for the slowest wave's taps is idle pipe -- so what shortens a
// block's share is more blocks per byte of taps.  With CG = 2 a loaded (and rotated) tap operand multiplies TWO spectrum operands,
// blocks 0 .. 15 and 16 .. 31: twice the matrix instructions per KiB of taps and per vector instruction, the same chain of FMAs per
// (block, channel, bin).
template <int P, int W, int D, bool WIN = false, bool SMALL = false, int CG = 1>
__device__ __forceinline__ void fold_mfma16_body(
		const float *__restrict__ taps, const float2 *__restrict__ spec, float2 *__restrict__ partial,
		size_t row_stride_f, size_t spec_stride, size_t partial_stride, int m, int slices, int rows, int octet_base, int nch, int nb,
		const int2 *__restrict__ win)
{
	static_assert(!WIN || W == 1, "windows are per wave: no spectrum tile is shared");
	static_assert(D == 2 || D == 4, "the LDS stage of a trip is a compile-time constant for even D");
	static_assert(CG == 1 || (CG == 2 && !SMALL && !WIN), "two column groups: the sixteen-column instruction, every row");
	// a quad's spectrum tile = 64 segments (block n, row k) of 128 contiguous bytes (16 bins) = 512 items of 16 bytes.  A load
	// instruction takes 64 consecutive items -- eight whole segments: eight cache lines, like a load of taps; fetched the way the
	// matrix operand wants them (lane n + 16 k <- its own 32 bytes) an instruction touched 64 lines and the address unit, not the matrix
	// pipe, set the pace.  EVERY wave fetches MINE instructions' worth -- with more than eight waves the upper ones would fetch (and
	// store) what the lower ones do -- so that all waves issue the same loads and no branch sits in the loop: behind a branch the
	// compiler's s_waitcnt count assumes the path with the most loads, and the waves on the other path wait for all but one quad
#ifdef HFDL_LAB
	unsigned long long clk_c0 = 0, clk_r0 = 0;
	const bool clk_me = blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
	if (clk_me) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
	constexpr int NLOADS = 8 * CG;                            // load instructions per quad's tile: 64 (CG = 2: 128) segments of eight 16-byte items
	constexpr int MINE = W >= NLOADS ? 1 : NLOADS / W;
	// v4f per bin pair: the segments + 1.  A stash instruction writes, per group of eight lanes, the eight 16-byte parts of ONE segment,
	// a pitch apart; LDS WRITES bank on a 32-dword modulus (MI355X_MICROARCH.md, LDS table), so the pitch in dwords must be 4 mod 32 for
	// the eight parts to fall on eight different bank quads.  (Rounds 5 - 6a had + 4 -- 16 mod 32: parts p and p + 2 on one bank, the
	// four-way conflict behind round 5's unexplained 1.0e8 SQ_LDS_BANK_CONFLICT cycles per launch; + 1: fold -1 .. 2 % alone, -2 % in
	// the pipeline, same bits, profiles/r06_experiments.md.)  The reads take 64 consecutive entries of a row: any pitch serves them.
	constexpr int XPITCH = 64 * CG + 1;
	__shared__ v4f xt[2][8][XPITCH];                          // [stage][bin pair][64 cg + segment n + 16 k] = (Re, Im) of bins 2 b, 2 b + 1 of block 16 cg + n, row k
	const int ngrp = m >> 4;
	// blockIdx -> (tile = bin group x slice, channel group).  The workgroups that share a spectrum tile must sit on ONE XCD (the
	// dispatcher puts block b on XCD b mod 8) and be resident TOGETHER, so that the tile comes out of HBM once and out of that XCD's
	// L2 for every other group: per XCD the channel groups vary fastest.
	const int ntile = ngrp * slices, groups = (int)gridDim.x / ntile;
	int tile_id, grp;
	if ((ntile & 7) == 0) {
		const int xcd = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
		grp = i % groups;
		tile_id = (i / groups) * 8 + xcd;
	} else {
		tile_id = (int)blockIdx.x % ntile;
		grp = (int)blockIdx.x / ntile;
	}
	const int g = tile_id % ngrp, s = tile_id / ngrp;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
	const int n = lane & 15, k = lane >> 4;
	const int octet0 = octet_base + (grp * W + wave) * P;
	const int sign_mask = (lane & 1) ? 0 : (int)0x80000000;
	const int quads = rows >> 2;
	int next_quad = 0, trips = quads;                         // WIN: the next quad to ask for (circular), quads in the window
	if constexpr (WIN) {
		const int2 wn = win[(octet0 - octet_base) / P];
		next_quad = __builtin_amdgcn_readfirstlane(wn.x);
		trips = __builtin_amdgcn_readfirstlane(wn.y);
	}
	// wave-uniform bases stepped by scalar adds + one byte offset per lane
	const char *tb = (const char *)(taps + (size_t)s * rows * row_stride_f + (size_t)octet0 * 64 * m + (size_t)g * 1024) + lane * 16;
	const size_t qs_b = row_stride_f * 16, os_b = (size_t)m * 256, xquad_b = (size_t)m * 32;       // bytes per quad of rows / per octet / per quad of spectrum rows
	const char *xp[MINE], *xp0[MINE];                         // item (wave * MINE + i) * 64 + lane = (segment, 16-byte part)
	int xseg[MINE];
#pragma unroll
	for (int i = 0; i < MINE; i++) {
		const int item = (((wave * MINE) % NLOADS) + i) * 64 + lane, seg = item >> 3, part = item & 7;
		const int sn = (seg & 15) + 16 * (seg >> 6), sk = (seg >> 4) & 3;
		const int bi = sn < nb ? sn : nb - 1;                 // columns past the last block repeat it; they are never stored
		xp[i] = xp0[i] = (const char *)(spec + (size_t)bi * spec_stride + ((size_t)s * rows + sk) * (size_t)m + g * 16 + 2 * part);
		xseg[i] = part * XPITCH + seg;
	}
	const char *const tb0 = tb;
	constexpr int NACC = SMALL ? 4 : 16;                      // accumulators per octet: one per bin quad (4 x 4 products) / one per bin
	v4f acc[P * CG][NACC];                                    // [octet p, column group cg] at p * CG + cg
#pragma unroll
	for (int p = 0; p < P * CG; p++)
#pragma unroll
		for (int j = 0; j < NACC; j++) acc[p][j] = v4f{ 0.f, 0.f, 0.f, 0.f };
	v4f h[D][P][4];
	v4f xs[D][MINE];
	auto issue = [&](int slot) {               // the loads of the next quad not yet asked for: spectrum share first, then the taps
		if constexpr (WIN) {                   // no branch: the quad index wraps by a scalar select
			tb = tb0 + (size_t)next_quad * qs_b;
#pragma unroll
			for (int i = 0; i < MINE; i++) xp[i] = xp0[i] + (size_t)next_quad * xquad_b;
			next_quad = next_quad + 1 == quads ? 0 : next_quad + 1;
		}
#pragma unroll
		for (int i = 0; i < MINE; i++) xs[slot][i] = *(const v4f *)xp[i];
#pragma unroll
		for (int p = 0; p < P; p++)
#pragma unroll
			for (int q = 0; q < 4; q++) h[slot][p][q] = __builtin_nontemporal_load((const v4f *)(tb + (size_t)p * os_b + 1024 * q));
		if constexpr (!WIN) {
#pragma unroll
			for (int i = 0; i < MINE; i++) xp[i] += xquad_b;
			tb += qs_b;
		}
	};
	auto stash = [&](int slot, int stage) {
#pragma unroll
		for (int i = 0; i < MINE; i++) (&xt[stage][0][0])[xseg[i]] = xs[slot][i];
	};
	auto multiply = [&](int slot, int stage) {
		if constexpr (SMALL) {
			// operand B of (bin quad q, alias row kk): lane 16 v + 4 cp + j <- block j of bin 4 q + v, row kk: half a tile entry
			const float *xl = (const float *)&xt[stage][0][0] + (size_t)((k >> 1) * XPITCH + (lane & 3)) * 4 + 2 * (k & 1);
#pragma unroll
			for (int q = 0; q < 4; q++) {
				v2f x[4];
#pragma unroll
				for (int kk = 0; kk < 4; kk++) x[kk] = *(const v2f *)(xl + (size_t)(2 * q * XPITCH + 16 * kk) * 4);
#pragma unroll
				for (int p = 0; p < P; p++) {
					// registers <-> 16-lane groups: (register v, group kk) -> (register kk, group v)
					typedef unsigned v2u __attribute__((ext_vector_type(2)));
					const v4f hv = h[slot][p][q];
					const v2u a02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(hv[0]), __float_as_uint(hv[2]), false, false);
					const v2u a13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(hv[1]), __float_as_uint(hv[3]), false, false);
					const v2u t01 = __builtin_amdgcn_permlane16_swap(a02[0], a13[0], false, false);
					const v2u t23 = __builtin_amdgcn_permlane16_swap(a02[1], a13[1], false, false);
					const float t[4] = { __uint_as_float(t01[0]), __uint_as_float(t01[1]), __uint_as_float(t23[0]), __uint_as_float(t23[1]) };
#pragma unroll
					for (int kk = 0; kk < 4; kk++) acc[p][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(t[kk], x[kk].x, acc[p][q], 0, 0, 0);
#pragma unroll
					for (int kk = 0; kk < 4; kk++) acc[p][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(rot90(t[kk], sign_mask), x[kk].y, acc[p][q], 0, 0, 0);
				}
			}
			return;
		}
#pragma unroll
		for (int b = 0; b < 8; b++) {
			v4f x[CG];                                        // (Re, Im) of bins 2 b and 2 b + 1 for (block 16 cg + n, row k)
#pragma unroll
			for (int cg = 0; cg < CG; cg++) x[cg] = xt[stage][b][64 * cg + lane];
#pragma unroll
			for (int p = 0; p < P; p++)
#pragma unroll
				for (int cg = 0; cg < CG; cg++) {
					v4f *a = acc[p * CG + cg];
					a[2 * b] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[slot][p][b >> 1][(2 * b) & 3], x[cg][0], a[2 * b], 0, 0, 0);
					a[2 * b + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[slot][p][b >> 1][(2 * b + 1) & 3], x[cg][2], a[2 * b + 1], 0, 0, 0);
				}
#pragma unroll
			for (int p = 0; p < P; p++) {
				const float r0 = rot90(h[slot][p][b >> 1][(2 * b) & 3], sign_mask), r1 = rot90(h[slot][p][b >> 1][(2 * b + 1) & 3], sign_mask);
#pragma unroll
				for (int cg = 0; cg < CG; cg++) {
					v4f *a = acc[p * CG + cg];
					a[2 * b] = __builtin_amdgcn_mfma_f32_16x16x4f32(r0, x[cg][1], a[2 * b], 0, 0, 0);
					a[2 * b + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(r1, x[cg][3], a[2 * b + 1], 0, 0, 0);
				}
			}
		}
	};
	// loads of different quads must stay in program order: s_waitcnt counts loads in issue order, and a prologue that asks for quad
	// 0's taps last costs every iteration its quads in flight -- scheduling barriers between the quads
#pragma unroll
	for (int d = 0; d < D; d++) {
		issue(d);
		__builtin_amdgcn_sched_barrier(0);
	}
	stash(0, 0);
	__syncthreads();
	for (int r = 0; r < trips - D; r += D) {
#pragma unroll
		for (int d = 0; d < D; d++) {
			multiply(d, d & 1);
			issue(d);
			stash((d + 1) % D, (d + 1) & 1);
			__builtin_amdgcn_sched_barrier(0);
			__syncthreads();
		}
	}
#pragma unroll
	for (int d = 0; d < D; d++) {
		multiply(d, d & 1);
		if (d + 1 < D) {
			stash(d + 1, (d + 1) & 1);
			__builtin_amdgcn_sched_barrier(0);
			__syncthreads();
		}
	}
	if constexpr (SMALL) {
		// D[q][row i][block j] of a 4 x 4 product sits in register i of lane 4 q + j: this lane holds block lane & 3, the channel pair
		// (lane >> 2) & 3 of each octet (Re, Im of its two channels in the four registers) and bin 4 q + (lane >> 4) of accumulator q
		const int j = lane & 3, cpq = (lane >> 2) & 3;
		if (j < nb) {
#pragma unroll
			for (int p = 0; p < P; p++)
#pragma unroll
				for (int cp = 0; cp < 2; cp++) {
					const int c = 8 * (octet0 + p) + 2 * cpq + cp;
					if (c >= nch) continue;
					float2 *po = partial + (size_t)j * partial_stride + ((size_t)c * slices + s) * (size_t)m + g * 16 + k;
#pragma unroll
					for (int q = 0; q < 4; q++) po[4 * q] = make_float2(acc[p][q][2 * cp], acc[p][q][2 * cp + 1]);
				}
		}
	} else
	// D[row i][block n] sits in register i & 3 of lane 16 (i >> 2) + n: this lane holds block n and the channels 2 k, 2 k + 1 of each
	// octet (Re, Im, Re, Im in its four registers), one accumulator per bin: 128 contiguous bytes per channel
	{
#pragma unroll
		for (int cg = 0; cg < CG; cg++) {
			const int blk = 16 * cg + n;
			if (blk >= nb) continue;
#pragma unroll
			for (int p = 0; p < P; p++)
#pragma unroll
				for (int cp = 0; cp < 2; cp++) {
					const int c = 8 * (octet0 + p) + 2 * k + cp;
					if (c >= nch) continue;
					const v4f *a = acc[p * CG + cg];
					float2 *po = partial + (size_t)blk * partial_stride + ((size_t)c * slices + s) * (size_t)m + g * 16;
#pragma unroll
					for (int j = 0; j < 16; j += 2)
						*(v4f *)(po + j) = v4f{ a[j][2 * cp], a[j][2 * cp + 1], a[j + 1][2 * cp], a[j + 1][2 * cp + 1] };
				}
		}
	}
#ifdef HFDL_LAB
	if (clk_me) {
		const unsigned slot = atomicAdd(&hfdl_fold_clk_probe_n, 1u) & 1023u;
		hfdl_fold_clk_probe[4 * slot] = (SMALL ? 4ull : 16ull * CG) * 100ull + (unsigned long long)nb;      // tag: columns x 100 + blocks
		hfdl_fold_clk_probe[4 * slot + 1] = __builtin_amdgcn_s_memtime() - clk_c0;
		hfdl_fold_clk_probe[4 * slot + 2] = __builtin_amdgcn_s_memrealtime() - clk_r0;
		hfdl_fold_clk_probe[4 * slot + 3] = clk_r0;
	}
#endif
}

template <int P, int W, int D, bool WIN = false, bool SMALL = false, int CG = 1>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(fold16_waves(P, W, D, SMALL, CG), fold16_waves(P, W, D, SMALL, CG)))) void fold_mfma16_kernel(
		const float *__restrict__ taps, const float2 *__restrict__ spec, float2 *__restrict__ partial,
		size_t row_stride_f, size_t spec_stride, size_t partial_stride, int m, int slices, int rows, int octet_base, int nch, int nb,
		const int2 *__restrict__ win)
{
	fold_mfma16_body<P, W, D, WIN, SMALL, CG>(taps, spec, partial, row_stride_f, spec_stride, partial_stride, m, slices, rows, octet_base, nch, nb, win);
}

#ifdef HFDL_LAB
// Laboratory: the thirty-two-column (1, 8, 2) tiling held to 208 registers, so that TWO of its waves and a demodulator wave (80) share a
// SIMD's 512.  The attribute wants a literal, hence a kernel of its own around the same body; it counts each half of the unified file
// (104 + 104); the compiler keeps the loop free of scratch at that (one 8-byte spill before the loop, reloaded after it).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) __attribute__((amdgpu_num_vgpr(104))) void fold32_two_waves_kernel(
		const float *__restrict__ taps, const float2 *__restrict__ spec, float2 *__restrict__ partial,
		size_t row_stride_f, size_t spec_stride, size_t partial_stride, int m, int slices, int rows, int octet_base, int nch, int nb,
		const int2 *__restrict__ win)
{
	fold_mfma16_body<1, 8, 2, false, false, 2>(taps, spec, partial, row_stride_f, spec_stride, partial_stride, m, slices, rows, octet_base, nch, nb, win);
}
#endif

#ifdef HFDL_LAB
// ---- what the memory system gives a kernel that only reads (bench.py's stream-read probe, profiles/fold_traffic.py's calibration) ----
template <int L, bool SPAN>
__global__ __launch_bounds__(FOLD_THREADS) void stream_read_kernel(const float4 *__restrict__ src, size_t chunks, float *__restrict__ sink)
{
	float acc = 0.f;
	const size_t per_block = chunks / gridDim.x;
	const size_t first = SPAN ? (size_t)blockIdx.x * per_block : blockIdx.x, step = SPAN ? 1 : gridDim.x;
	const size_t last = SPAN ? first + per_block : chunks;
	for (size_t ch = first; ch < last; ch += step) {
		const v4f *p = (const v4f *)src + ch * (L * FOLD_THREADS) + threadIdx.x;
		v4f v[L];
#pragma unroll
		for (int i = 0; i < L; i++) v[i] = __builtin_nontemporal_load(p + i * FOLD_THREADS);
#pragma unroll
		for (int i = 0; i < L; i++) acc += v[i].x + v[i].w;
	}
	if (acc == 1.2345e33f) *sink = acc;
}

int fold_clock_probe_read(unsigned long long *out, int max, int *n)
{
	unsigned cnt = 0;
	if (hipDeviceSynchronize() != hipSuccess) return -1;
	if (hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(hfdl_fold_clk_probe_n), sizeof(cnt)) != hipSuccess) return -1;
	unsigned long long all[1024 * 4];
	if (hipMemcpyFromSymbol(all, HIP_SYMBOL(hfdl_fold_clk_probe), sizeof(all)) != hipSuccess) return -1;
	const unsigned have = cnt < 1024u ? cnt : 1024u;
	int k = 0;
	for (unsigned i = 0; i < have && k < max; i++) {
		const unsigned slot = (cnt - have + i) & 1023u;
		for (int j = 0; j < 4; j++) out[4 * k + j] = all[4 * slot + j];
		k++;
	}
	*n = k;
	cnt = 0;
	return hipMemcpyToSymbol(HIP_SYMBOL(hfdl_fold_clk_probe_n), &cnt, sizeof(cnt)) == hipSuccess ? 0 : -1;
}

int stream_read_variants() { return 4; }

void launch_stream_read(int variant, const float2 *src, size_t bytes, float *sink, hipStream_t st)
{
	const dim3 block(FOLD_THREADS);
	switch (variant) {
	case 0: hipLaunchKernelGGL((stream_read_kernel<4, false>), dim3(2048), block, 0, st, (const float4 *)src, bytes / (64 * FOLD_THREADS), sink); break;
	case 1: hipLaunchKernelGGL((stream_read_kernel<8, false>), dim3(2048), block, 0, st, (const float4 *)src, bytes / (128 * FOLD_THREADS), sink); break;
	case 2: hipLaunchKernelGGL((stream_read_kernel<4, true>), dim3((unsigned)(bytes >> 22)), block, 0, st, (const float4 *)src, bytes / (64 * FOLD_THREADS), sink); break;
	default: hipLaunchKernelGGL((stream_read_kernel<8, true>), dim3((unsigned)(bytes >> 22)), block, 0, st, (const float4 *)src, bytes / (128 * FOLD_THREADS), sink); break;
	}
}
#endif

// ---- the compiled tilings ----
struct FoldArgs {
	const float *taps;
	const float2 *spec;
	float2 *partial;
	size_t rs_f, ss, ps;
	int m, slices, rows, nch, ngroups, nb;            // ngroups: channel groups of the tap layout (octets / pairs) the buffer holds
	hipStream_t st;
	hipEvent_t start, stop;
};

// workgroups of 8 P W channels first; the octets left over get single-wave workgroups in a launch of their own
template <int P, int W, int D, bool SMALL = false, int CG = 1>
static int fold16_go(const FoldArgs &a)
{
	const int ntile = (a.m >> 4) * a.slices;
	const int groups = a.ngroups / (P * W), rest = a.ngroups - groups * P * W;
	int launches = 0;
	if (groups > 0) {
		hipExtLaunchKernelGGL((fold_mfma16_kernel<P, W, D, false, SMALL, CG>), dim3((unsigned)(groups * ntile)), dim3(64 * W), 0, a.st, a.start, rest ? nullptr : a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, a.slices, a.rows, 0, a.nch, a.nb, (const int2 *)nullptr);
		launches++;
	}
	if (rest > 0) {
		// (a single wave fetches the whole spectrum tile itself: two quads in flight keep that within its registers when the tile is 32 blocks wide)
		constexpr int DR = CG > 1 ? 2 : D;
		hipExtLaunchKernelGGL((fold_mfma16_kernel<1, 1, DR, false, SMALL, CG>), dim3((unsigned)(rest * ntile)), dim3(64), 0, a.st, groups > 0 ? nullptr : a.start, a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, a.slices, a.rows, groups * P * W, a.nch, a.nb, (const int2 *)nullptr);
		launches++;
	}
	return launches;
}

#ifdef HFDL_LAB
static int fold32_two_waves_go(const FoldArgs &a)
{
	const int ntile = (a.m >> 4) * a.slices;
	const int groups = a.ngroups / 8, rest = a.ngroups - groups * 8;
	int launches = 0;
	if (groups > 0) {
		hipExtLaunchKernelGGL(fold32_two_waves_kernel, dim3((unsigned)(groups * ntile)), dim3(512), 0, a.st, a.start, rest ? nullptr : a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, a.slices, a.rows, 0, a.nch, a.nb, (const int2 *)nullptr);
		launches++;
	}
	if (rest > 0) {
		hipExtLaunchKernelGGL((fold_mfma16_kernel<1, 1, 2, false, false, 2>), dim3((unsigned)(rest * ntile)), dim3(64), 0, a.st, groups > 0 ? nullptr : a.start, a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, a.slices, a.rows, groups * 8, a.nch, a.nb, (const int2 *)nullptr);
		launches++;
	}
	return launches;
}
#endif

// the pruned fold: a single-wave workgroup per octet and group of 16 bins, each octet with its own window of quads; one slice
template <int D>
static int fold16_go_win(const FoldArgs &a, const int2 *win)
{
	const int ntile = a.m >> 4;
	hipExtLaunchKernelGGL((fold_mfma16_kernel<1, 1, D, true>), dim3((unsigned)(a.ngroups * ntile)), dim3(64), 0, a.st, a.start, a.stop, 0,
		a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, 1, a.rows, 0, a.nch, a.nb, win);
	return 1;
}

// P octets per wave, W waves per workgroup, D quads of loads in flight; a tiling takes any block count up to 16
struct FoldVariant { int layout, p, q, w, d; int (*go)(const FoldArgs &); };
#define F16(P, W, D) { TAPL_OCTET, P, 4, W, D, fold16_go<P, W, D> }
#define F4(P, W, D) { TAPL_OCTET, P, 1, W, D, fold16_go<P, W, D, true> }        // the four-column form: at most four blocks
#define F32(P, W, D) { TAPL_OCTET, P, 8, W, D, fold16_go<P, W, D, false, 2> }   // two column groups: 17 .. 32 blocks
// The first entry whose look-ahead D divides the slice's quads is the one used.
// Measured on cfg3 (M = 4096) with profiles/fold_variants.py: profiles/r05/fold_variants_cfg3_k4.md (512 rows per slice),
// profiles/r06/run10_fold_variants_slices1.md (one slice of 2048 rows, the default since round 6).
static const FoldVariant fold_variants[] = {
	// (2, 4, 4): 372 registers -- ONE wave per SIMD, four quads of taps in flight -- and 128 left on the SIMD: the demodulator's waves
	// (73 used, 80 allocated) fit beside it.  (2, 4, 2) at 256 registers runs two waves per SIMD, is faster alone and leaves no room: the
	// demodulator takes turns with it and the pipeline loses 10 % (profiles/r05/k4_tilings_in_pipeline.txt).  The thirty-two-column
	// (1, 4, 4) takes 420 registers; two waves of it held to 208 beside a demodulator wave (laboratory tiling 25) gain nothing in the
	// pipeline: the denser the matrix instructions, the less the demodulator's waves get to issue (profiles/r06_experiments.md)
	F16(2, 4, 4), F16(2, 4, 2),
	F4(2, 4, 2),
	F32(1, 4, 4), F32(1, 4, 2),
#ifdef HFDL_LAB
	F32(1, 8, 2), F32(1, 8, 4), F32(1, 2, 2), F32(2, 4, 2),
	F4(2, 4, 4), F4(4, 4, 2), F4(1, 8, 4), F4(2, 8, 2), F4(2, 8, 4), F4(4, 2, 2), F4(1, 4, 4),
	F16(1, 4, 2), F16(1, 4, 4), F16(1, 8, 2), F16(1, 8, 4), F16(2, 2, 2), F16(2, 8, 2), F16(1, 2, 2), F16(1, 2, 4), F16(3, 4, 2),
	{ TAPL_OCTET, 1, 8, 8, 2, fold32_two_waves_go },          // index 25: F32(1, 8, 2) held to 208 registers -- two waves per SIMD beside a demodulator wave
#endif
};
#undef F32
#undef F4
#undef F16
constexpr int N_FOLD_VARIANTS = (int)(sizeof(fold_variants) / sizeof(fold_variants[0]));

int fold_variant_count() { return N_FOLD_VARIANTS; }

int fold_variant_describe(int v, int desc[6])
{
	if (v < 0 || v >= N_FOLD_VARIANTS) return -1;
	const FoldVariant &f = fold_variants[v];
	desc[0] = f.p; desc[1] = f.q; desc[2] = f.w; desc[3] = f.d; desc[4] = 4 * f.q; desc[5] = f.layout;
	return 0;
}

static bool variant_fits(const FoldVariant &f, const Geometry &g)
{
	return g.tap_layout == f.layout && (g.rows_per_slice & 3) == 0 && (g.rows_per_slice >> 2) % f.d == 0;
}

// the tiling used for this geometry: the first entry of the list that fits (null: the plain-VALU kernel folds)
static const FoldVariant *pick_variant(const Geometry &g, int nb)
{
	if (g.fold_tile >= 0 && g.fold_tile < N_FOLD_VARIANTS && 4 * fold_variants[g.fold_tile].q >= nb && variant_fits(fold_variants[g.fold_tile], g)) return &fold_variants[g.fold_tile];
	for (const FoldVariant &f : fold_variants)             // up to four blocks: the four-column form
		if (nb <= 4 && f.q == 1 && variant_fits(f, g)) return &f;
	for (const FoldVariant &f : fold_variants)             // the narrowest form that holds the blocks: sixteen columns up to 16 blocks, thirty-two beyond
		if (f.q == (nb <= 16 ? 4 : 8) && variant_fits(f, g)) return &f;
	return nullptr;
}

static FoldArgs fold_args(const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial, size_t partial_stride,
		int nb, hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
	FoldArgs a;
	a.taps = (const float *)taps; a.spec = spectrum; a.partial = partial;
	a.rs_f = (size_t)g.tap_row_stride * 2; a.ss = spec_stride; a.ps = partial_stride;
	a.m = g.m; a.slices = g.slices; a.rows = g.rows_per_slice; a.nch = g.nch; a.ngroups = g.nch_pad / tap_layout_group(g.tap_layout); a.nb = nb;
	a.st = st; a.start = start; a.stop = stop;
	return a;
}

static void launch_fold_ref(const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial, size_t partial_stride,
		int nb, hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
	hipExtLaunchKernelGGL(fold_ref_kernel, dim3((unsigned)(g.nch * g.slices)), dim3(FOLD_THREADS), 0, st, start, stop, 0,
		(const float *)taps, spectrum, partial, (size_t)g.tap_row_stride * 2, spec_stride, partial_stride, g.m, g.slices, g.rows_per_slice, g.tap_layout, nb);
}

int launch_fold_variant(int v, const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial,
		size_t partial_stride, int nb, hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
	if (v == -1) { launch_fold_ref(g, taps, spectrum, spec_stride, partial, partial_stride, nb, st, start, stop); return 1; }    // the FMA-chain reference
	if (v < 0 || v >= N_FOLD_VARIANTS || !variant_fits(fold_variants[v], g) || nb < 1 || nb > 4 * fold_variants[v].q) return -1;
	return fold_variants[v].go(fold_args(g, taps, spectrum, spec_stride, partial, partial_stride, nb, st, start, stop));
}

int launch_fold(const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial, size_t partial_stride,
		int nb, int nb_max, hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
	int launches = 0;
	if (nb_max > FOLD_MAX_BLOCKS) nb_max = FOLD_MAX_BLOCKS;
	// one launch per nb_max blocks: a launch takes ANY block count up to 16 (columns past the last block are computed and dropped)
	if (g.fold_win && g.tap_layout == TAPL_OCTET && nb_max > 16) nb_max = 16;      // the pruned fold has the sixteen-column form only
	for (int done = 0; done < nb;) {
		int take = nb - done < nb_max ? nb - done : nb_max;
		if (take > 16 && !pick_variant(g, take)) take = 16;                       // no thirty-two-column tiling fits this geometry
		const bool first = done == 0, last = done + take >= nb;
		const float2 *sp = spectrum + (size_t)done * spec_stride;
		float2 *pp = partial + (size_t)done * partial_stride;
		const FoldVariant *f = pick_variant(g, take);
		if (g.fold_win && g.tap_layout == TAPL_OCTET) launches += fold16_go_win<2>(fold_args(g, taps, sp, spec_stride, pp, partial_stride, take, st, first ? start : nullptr, last ? stop : nullptr), g.fold_win);
		else if (f) launches += f->go(fold_args(g, taps, sp, spec_stride, pp, partial_stride, take, st, first ? start : nullptr, last ? stop : nullptr));
		else { launch_fold_ref(g, taps, sp, spec_stride, pp, partial_stride, take, st, first ? start : nullptr, last ? stop : nullptr); launches++; }
		done += take;
	}
	return launches;
}

// energy of every (alias row, channel) of the octet-interleaved taps: one workgroup per (quad of rows, octet) walks its M / 4 chunks of
// 1 KiB (lane = 16 (row % 4) + 2 (c % 8) + comp: the two lanes of a (row, channel) add into one cell)
__global__ __launch_bounds__(FOLD_THREADS) void tap_row_energy_kernel(const float *__restrict__ taps, float *__restrict__ energy, size_t row_stride_f, int m, int noct, int nch_pad)
{
	const int oct = blockIdx.x % noct, quad = blockIdx.x / noct;
	const v4f *base = (const v4f *)(taps + (size_t)quad * 4 * row_stride_f + (size_t)oct * 64 * m);
	float acc = 0.f;
	for (int e = threadIdx.x; e < (m >> 2) * 64; e += FOLD_THREADS) {       // float4 index: chunk * 64 + lane
		const v4f v = base[e];
		acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
	}
	const int lane = threadIdx.x & 63;            // FOLD_THREADS is a multiple of 64: a thread keeps its lane position over the chunks
	atomicAdd(energy + (size_t)(4 * quad + (lane >> 4)) * nch_pad + oct * 8 + ((lane & 15) >> 1), acc);
}

void launch_tap_row_energy(const float2 *taps, const Geometry &g, float *energy, hipStream_t st)
{
	const int noct = g.nch_pad / 8;
	hipLaunchKernelGGL(tap_row_energy_kernel, dim3((unsigned)((g.pre >> 2) * noct)), dim3(FOLD_THREADS), 0, st, (const float *)taps, energy, (size_t)g.tap_row_stride * 2, g.m, noct, g.nch_pad);
}

// filter taps of one channel back in plain order (HFDL_GPU_TAP_FILTER): dst[N] cf32
__global__ __launch_bounds__(FOLD_THREADS) void tap_extract_kernel(const float *__restrict__ taps, float2 *__restrict__ dst, size_t row_stride_f, int m, int pre, int layout, int c)
{
	const size_t n = (size_t)m * pre;
	for (size_t e = (size_t)blockIdx.x * FOLD_THREADS + threadIdx.x; e < n; e += (size_t)gridDim.x * FOLD_THREADS)
		dst[e] = tap_at(taps, row_stride_f, m, layout, c, (int)(e / m), (int)(e % m));
}

void launch_tap_extract(const float2 *taps, const Geometry &g, int channel, float2 *dst, hipStream_t st)
{
	hipLaunchKernelGGL(tap_extract_kernel, dim3(1024), dim3(FOLD_THREADS), 0, st, (const float *)taps, dst, (size_t)g.tap_row_stride * 2, g.m, g.pre, g.tap_layout, channel);
}

// ---- inverse FFT + scrap + NCO/decimate : one workgroup per channel, M bins in LDS ----
//
// inv_in[(h0 + j) mod M] = Y[j], h0 = (N - offsetbin + M/2) mod M      (src/fastddc.c:130)
// fft_swap_sides(inv_in)  ->  x[u] = Y[(u - h0 - M/2) mod M]            (:190)
// y = IFFT_M(x) / (pre * M); drop `scrap`; out[k] = y[scrap + rem + q k] * e^{j phi_k}   (:193-211)
// phi_k follows the reference's fp32 phasor recurrence exactly: it is serial (1792 dependent steps at cfg3, ~16 us), so
// wave 0 runs it while the other 15 waves sum the fold slices out of HBM; all 16 then share the inverse FFT.
constexpr int IFFT_THREADS = 1024;

// ph[i] = the phasor that multiplies output i (recurrence: fft_core.h); one thread runs it -- it is a serial chain
__device__ __forceinline__ void nco_phasor_run(float2 *ph, int cnt, float starting_phase, float cd, float sd)
{
	float2 p = nco_phasor_seed(starting_phase);
	for (int i = 0; i < cnt; i++) {
		ph[i] = p;
		nco_phasor_step(p.x, p.y, cd, sd);
	}
}

// output[k] = input[i] * e^{j phi_k}, the reference's expression term for term (:54-57)
__device__ __forceinline__ float2 nco_rotate(float2 p, float2 v)
{
#pragma clang fp contract(off)
	const float re = p.x * v.x - p.y * v.y, im = p.y * v.x + p.x * v.y;
	return make_float2(re, im);
}

// grid = channels x blocks of the batch; every block has its own partial sums, carried-state snapshot and phasor table (both made
// by the riders of that block's forward FFT), so the blocks of a batch are independent here
__global__ __launch_bounds__(IFFT_THREADS) void ifft_nco_kernel(const float2 *__restrict__ partial, size_t partial_stride, const ChanConst *__restrict__ cc,
		const NcoState *__restrict__ snap, const float2 *__restrict__ ph, size_t ph_stride, const float2 *__restrict__ tw, float2 *__restrict__ chan_out,
		int *__restrict__ out_count, Geometry g, int logm)
{
	extern __shared__ float2 sm[];          // m bins
	const int c = blockIdx.x % g.nch, b = blockIdx.x / g.nch;
	const ChanConst k = cc[c];
	const int m = g.m, mask = m - 1;
	const NcoState st = snap[(size_t)b * g.nch + c];
	const int q = g.post;
	const int cnt = nco_output_count(st, g.post_input_size, q);
	{
		const int h0 = (int)(((long long)g.n - k.offsetbin + m / 2) % m);
		const float2 *pc = partial + (size_t)b * partial_stride + (size_t)c * g.slices * (size_t)m;
		for (int u = threadIdx.x; u < m; u += IFFT_THREADS) {
			const int j = (u - h0 - m / 2) & mask;
			float2 acc = make_float2(0.f, 0.f);
			for (int s = 0; s < g.slices; s++) {
				float2 v = pc[(size_t)s * m + j];
				acc.x += v.x; acc.y += v.y;
			}
			sm[u] = acc;
		}
	}
	__syncthreads();
	lds_fft_columns<+1>(sm, m, logm, 1, 0, tw);

	const float norm = (float)g.pre * (float)m;
	float2 *o = chan_out + ((size_t)b * g.nch + c) * g.outs;
	const float2 *phb = ph + (size_t)b * ph_stride;
	for (int i = threadIdx.x; i < cnt; i += IFFT_THREADS) {
		const int idx = g.scrap + st.decimation_remain + q * i;
		float2 v = sm[(int)(__brev((unsigned)idx) >> (32 - logm))];
		v.x = __fdiv_rn(v.x, norm); v.y = __fdiv_rn(v.y, norm);
		o[i] = nco_rotate(phb[(size_t)i * g.nch + c], v);       // the block's phasor table, made beside its forward FFT (kernels.h NcoJob)
	}
	if (threadIdx.x == 0) out_count[(size_t)b * g.nch + c] = cnt;      // the carried state itself is advanced by the riders (fft_core.h)
}

// the NCO / decimator stage on its own (stage entry point hfdl_gpu_nco_decimate): the same device functions the channelizer
// kernel above runs after its inverse FFT, one workgroup, phasors through a global scratch buffer
__global__ __launch_bounds__(IFFT_THREADS) void nco_decimate_kernel(const float2 *__restrict__ in, int input_size, float cd, float sd, float rate,
		int q, NcoState *__restrict__ state, float2 *__restrict__ ph, float2 *__restrict__ out)
{
	NcoState st = *state;
	const int cnt = nco_output_count(st, input_size, q);
	if (threadIdx.x == 0) nco_phasor_run(ph, cnt, st.starting_phase, cd, sd);
	__syncthreads();
	for (int i = threadIdx.x; i < cnt; i += IFFT_THREADS) out[i] = nco_rotate(ph[i], in[st.decimation_remain + q * i]);
	__syncthreads();
	if (threadIdx.x == 0) {
		nco_advance(st, cnt, q, input_size, rate);
		*state = st;
	}
}

void launch_nco_decimate(const float2 *in, int input_size, float cd, float sd, float rate, int q, NcoState *state, float2 *ph, float2 *out, hipStream_t st)
{
	hipLaunchKernelGGL(nco_decimate_kernel, dim3(1), dim3(IFFT_THREADS), 0, st, in, input_size, cd, sd, rate, q, state, ph, out);
}

// once per front end: an inverse FFT of 8192 points needs more than the default 64 KiB of dynamic LDS
hipError_t prepare_ifft_nco(int m)
{
	const size_t lds = sizeof(float2) * ((size_t)m + 1);
	if (lds <= 64 * 1024) return hipSuccess;
	return hipFuncSetAttribute((const void *)ifft_nco_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

void launch_ifft_nco(const Geometry &g, const float2 *partial, size_t partial_stride, const ChanConst *cc, const NcoState *snap, const float2 *ph,
		size_t ph_stride, const float2 *tw_m, float2 *chan_out, int *out_count, int nb, hipStream_t st, hipEvent_t done, hipEvent_t start)
{
	int logm = 0;
	while ((1 << logm) < g.m) logm++;
	size_t lds = sizeof(float2) * ((size_t)g.m + 1);
	hipExtLaunchKernelGGL(ifft_nco_kernel, dim3((unsigned)(g.nch * nb)), dim3(IFFT_THREADS), (unsigned)lds, st, start, done, 0, partial, partial_stride, cc, snap, ph, ph_stride,
			tw_m, chan_out, out_count, g, logm);
}

}  // namespace hfdl
