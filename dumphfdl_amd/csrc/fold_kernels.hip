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
// (alias rows x blocks).  It runs on the fp32 matrix pipe -- v_mfma_f32_16x16x1_4B_f32: four independent 16 x 16 outer products per
// instruction = four bins x eight channels' Re / Im rows x sixteen blocks -- which takes the multiply-accumulates off the vector ALUs
// the demodulator kernel next door lives on.  Each product is one exact fmaf, applied in a fixed order (cmac_chain below).
#include <hip/hip_ext.h>
#include <type_traits>
#include "kernels.h"
#include "fft_core.h"

namespace hfdl {

constexpr int FOLD_THREADS = 256;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// ---- tap layouts: kernels.h (TAPL_*, tap_offset_f) ----

__device__ __forceinline__ float2 tap_at(const float *taps, size_t row_stride_f, int m, int layout, int c, int row, int j)
{
	const float *p = taps + (size_t)row * row_stride_f + tap_offset_f(layout, m, c, j, 0);
	return make_float2(p[0], p[layout == TAPL_PLAIN ? 1 : 4]);
}

// One complex multiply-accumulate per bin as a FIXED chain of four fused multiply-adds -- the order the matrix instructions below
// apply them in (real part of X first, then the imaginary part), so that the plain-VALU reference and every MFMA tiling round a
// (block, channel, bin) sum exactly alike: a block folded alone and the same block folded beside fifteen others give the same 32 bits
// (tests/test_gpu_parity.py::test_fold_batching_changes_nothing, ::test_fold_mfma_equals_fma_chain).
__device__ __forceinline__ void cmac_chain(float2 &a, const float2 h, const float2 x)
{
	a.x = __builtin_fmaf(h.x, x.x, a.x); a.y = __builtin_fmaf(h.y, x.x, a.y);
	a.x = __builtin_fmaf(-h.y, x.y, a.x); a.y = __builtin_fmaf(h.x, x.y, a.y);
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
			for (int r = 0; r < rows; r++) {
				const float2 h = tap_at(taps, row_stride_f, m, layout, c, s * rows + r, j);
#pragma unroll
				for (int k = 0; k < 4; k++)
					if (b0 + k < nb) cmac_chain(acc[k], h, sp[(size_t)k * spec_stride + (size_t)r * m]);
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

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int fold16_waves(int p, int w, int d, bool small = false)
{
	const int mine = (2 + w - 1) / w;
	const int regs = (small ? 16 : 64) * p + 4 * p * d + 4 * mine * d + 8 + 4 * p + 28;
	return regs <= 96 ? 5 : regs <= 128 ? 4 : regs <= 168 ? 3 : 2;
}

// THE fold: v_mfma_f32_16x16x1_4B_f32, TAPL_OCTET taps.  One instruction = four bins x (8 channels' Re / Im rows) x 16 blocks: 1024
// multiply-accumulates, a quarter of the register-file traffic per product of the 4x4x1 form (whose launches the board ran at
// 1.3 - 1.5 GHz: profiles/r05_experiments.md), and the whole batch of up to 16 blocks in its sixteen columns.
// P channel OCTETS per wave, W waves per workgroup, D rows of loads in flight.  A workgroup = one group of 16 bins x one slice of alias
// rows x 8 P W channels; its W waves cover the SAME bins and different channels, so each alias row's spectrum tile (16 blocks x 16
// bins = 2 KiB) is fetched ONCE per workgroup (two waves load half of it each, already in operand-B order: lane n + 16 blk = bin blk
// of block n), written to LDS and read from there by all waves.  Per alias row and wave: P tap loads of 1 KiB (non-temporal), two LDS
// reads, 4 P vector instructions (the rotated operand) and 8 P matrix instructions: first every accumulator's Re(X) product, then every
// Im(X) product.  Loads run D rows ahead, the spectrum tile one row ahead through two LDS stages, one barrier per row.
// WIN (the pruned fold, hfdl_gpu.h HFDL_GPU_FOLD_PRUNE): a workgroup folds only the window of alias rows `win[group]` = (first row,
// count) around its channels' pass bands -- circular, one slice, rows = all alias rows -- instead of a slice of all of them.
// SMALL (launches of at most FOUR blocks: the ragged end of a run, a live receiver's block at a time): the same taps, the same loop,
// v_mfma_f32_4x4x1_16B_f32 instead -- sixteen 4 x 4 products per instruction: lane 4 q + i of operand A = row i of (bin-set register v,
// bin 4 v + (lane >> 4), channel pair (lane >> 2) & 3) -- the octet layout as it lies -- and lane 4 q + j of operand B = block j of
// that bin.  The sixteen-column form computes all sixteen columns whatever the block count (4.0 ms per cfg3 launch at 1 ... 16 blocks);
// this one leaves the matrix pipe three quarters idle and the launch to the HBM reads of the taps.  Same FMA chain per sum: same bits.
template <int P, int W, int D, bool WIN = false, int FORM = 0>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(fold16_waves(P, W, D, FORM != 0), fold16_waves(P, W, D, FORM != 0)))) void fold_mfma16_kernel(
		const float *__restrict__ taps, const float2 *__restrict__ spec, float2 *__restrict__ partial,
		size_t row_stride_f, size_t spec_stride, size_t partial_stride, int m, int slices, int rows, int octet_base, int nch, int nb,
		const int2 *__restrict__ win = nullptr)
{
	static_assert(!WIN || W == 1, "windows are per wave: no spectrum tile is shared");
	constexpr bool SMALL = FORM == 1;          // FORM 0: sixteen columns (16x16x1_4B); 1: four columns (4x4x1_16B); 2 (laboratory): a TIMING probe, see fold_variants[]
	constexpr bool K4PROBE = FORM == 2;
	static_assert(D == 2 || D == 4, "the LDS stage of a trip is a compile-time constant for even D");
	// a row's spectrum tile = 4 pieces of 512 B (bin-set v = 0 .. 3: lane n + 16 blk <- bin 4 v + blk of block n).  EVERY wave fetches
	// MINE of them -- with more than four waves the upper ones fetch (and store) what the lower ones do -- so that all waves issue the
	// same loads and no branch sits in the loop: behind a branch the compiler's s_waitcnt count assumes the path with the most loads,
	// and the waves on the other path wait for all but one row of theirs
	constexpr int MINE = W >= 4 ? 1 : 4 / W;
	__shared__ v2f xt[2][4][64];                              // [stage][bin-set][lane] = (Re, Im)
	const int ngrp = m >> 4;
	// blockIdx -> (tile = bin group x slice, channel group), XCD-aware as in fold_mfma_kernel
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
	const int n = FORM == 1 ? lane & 3 : lane & 15, blk = lane >> 4;
	const int octet0 = octet_base + (grp * W + wave) * P;
	const int sign_mask = (lane & 1) ? 0 : (int)0x80000000;
	int next_row = 0, trips = rows;                           // WIN: the next row to ask for (circular), rows in the window
	if constexpr (WIN) {
		const int2 wn = win[(octet0 - octet_base) / P];
		next_row = __builtin_amdgcn_readfirstlane(wn.x);
		trips = __builtin_amdgcn_readfirstlane(wn.y);
	}
	const char *tb = (const char *)(taps + (size_t)s * rows * row_stride_f + (size_t)octet0 * 16 * m + (size_t)g * 256) + lane * 16;
	const size_t rs_b = row_stride_f * 4, os_b = (size_t)m * 64, xrow_b = (size_t)m * 8;
	const int v0 = (wave * MINE) & 3;                         // this wave's first piece
	const char *xp;
	{
		const int bi = n < nb ? n : nb - 1;                   // columns past the last block repeat it; they are never stored
		xp = (const char *)(spec + (size_t)bi * spec_stride + (size_t)s * rows * (size_t)m + g * 16 + 4 * v0 + blk);
	}
	const char *const tb0 = tb, *const xp0 = xp;
	typedef typename std::conditional<FORM != 0, v4f, v16f>::type Acc;
	Acc acc[P][4];
#pragma unroll
	for (int p = 0; p < P; p++)
#pragma unroll
		for (int v = 0; v < 4; v++)
#pragma unroll
			for (int e = 0; e < (FORM != 0 ? 4 : 16); e++) acc[p][v][e] = 0.f;
	v4f h[D][P];
	v2f xs[D][MINE];
	auto issue = [&](int slot) {               // the loads of the next row not yet asked for: spectrum share first, then the taps
		if constexpr (WIN) {                   // no branch: the row index wraps by a scalar select
			tb = tb0 + (size_t)next_row * rs_b;
			xp = xp0 + (size_t)next_row * xrow_b;
			next_row = next_row + 1 == rows ? 0 : next_row + 1;
		}
#pragma unroll
		for (int i = 0; i < MINE; i++) xs[slot][i] = *(const v2f *)(xp + 32 * i);       // consecutive bin-sets: four bins apart
#pragma unroll
		for (int p = 0; p < P; p++) h[slot][p] = __builtin_nontemporal_load((const v4f *)(tb + (size_t)p * os_b));
		if constexpr (!WIN) {
			xp += xrow_b;
			tb += rs_b;
		}
	};
	auto stash = [&](int slot, int stage) {
#pragma unroll
		for (int i = 0; i < MINE; i++) xt[stage][v0 + i][lane] = xs[slot][i];
	};
	auto multiply = [&](int slot, int stage) {
		v2f x[4];
#pragma unroll
		for (int v = 0; v < 4; v++) x[v] = xt[stage][v][lane];
#pragma unroll
		for (int p = 0; p < P; p++)
#pragma unroll
			for (int v = 0; v < 4; v++) {
				if constexpr (SMALL) acc[p][v] = __builtin_amdgcn_mfma_f32_4x4x1f32(h[slot][p][v], x[v].x, acc[p][v], 0, 0, 0);
				else if constexpr (K4PROBE) acc[p][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(h[slot][p][v], x[v].x, acc[p][v], 0, 0, 0);
				else acc[p][v] = __builtin_amdgcn_mfma_f32_16x16x1f32(h[slot][p][v], x[v].x, acc[p][v], 0, 0, 0);
			}
#pragma unroll
		for (int p = 0; p < P; p++)
#pragma unroll
			for (int v = 0; v < 4; v++) {
				if constexpr (SMALL) acc[p][v] = __builtin_amdgcn_mfma_f32_4x4x1f32(rot90(h[slot][p][v], sign_mask), x[v].y, acc[p][v], 0, 0, 0);
				else if constexpr (K4PROBE) acc[p][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(rot90(h[slot][p][v], sign_mask), x[v].y, acc[p][v], 0, 0, 0);
				else acc[p][v] = __builtin_amdgcn_mfma_f32_16x16x1f32(rot90(h[slot][p][v], sign_mask), x[v].y, acc[p][v], 0, 0, 0);
			}
	};
	// loads of different rows must stay in program order (see fold_mfma_kernel): scheduling barriers between the rows
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
	if constexpr (FORM != 0) {
		// (the probe form stores through the same code: its numbers mean nothing)
		// D[q][row i][block j] sits in register i of lane 4 q + j: this lane holds block n = lane & 3, the channel pair (lane >> 2) & 3 of
		// each octet (Re, Im of its two channels in registers 0 .. 3) and bin 4 v + (lane >> 4) of bin-set register v
		if (n < nb) {
			const int cpq = (lane >> 2) & 3;
#pragma unroll
			for (int p = 0; p < P; p++)
#pragma unroll
				for (int cp = 0; cp < 2; cp++) {
					const int c = 8 * (octet0 + p) + 2 * cpq + cp;
					if (c >= nch) continue;
					float2 *po = partial + (size_t)n * partial_stride + ((size_t)c * slices + s) * (size_t)m + g * 16 + blk;
#pragma unroll
					for (int v = 0; v < 4; v++) po[4 * v] = make_float2(acc[p][v][2 * cp], acc[p][v][2 * cp + 1]);
				}
		}
	} else
	// D[bin blk][row i][block n] sits in register 4 blk + (i & 3) of lane 16 (i >> 2) + n: this lane holds block n, channels
	// 2 (lane >> 4) and + 1 of each octet (Re, Im in adjacent registers), bins 4 v + (0 .. 3): 32 contiguous bytes per channel and v
	if (n < nb) {
#pragma unroll
		for (int p = 0; p < P; p++)
#pragma unroll
			for (int cp = 0; cp < 2; cp++) {
				const int c = 8 * (octet0 + p) + 2 * blk + cp;
				if (c >= nch) continue;
				float2 *po = partial + (size_t)n * partial_stride + ((size_t)c * slices + s) * (size_t)m + g * 16;
#pragma unroll
				for (int v = 0; v < 4; v++) {
					const Acc a = acc[p][v];
					((v4f *)(po + 4 * v))[0] = v4f{ a[2 * cp], a[2 * cp + 1], a[4 + 2 * cp], a[4 + 2 * cp + 1] };
					((v4f *)(po + 4 * v))[1] = v4f{ a[8 + 2 * cp], a[8 + 2 * cp + 1], a[12 + 2 * cp], a[12 + 2 * cp + 1] };
				}
			}
	}
}

#ifdef HFDL_LAB
// registers the tile asks for -> waves per SIMD told to the compiler (512 per lane and SIMD): left to itself it aims at 8 waves,
// squeezes the loop into 64 registers and gets there by loading, waiting, multiplying, loading again
constexpr int fold_mfma_waves(int p, int q, int w, int d)
{
	const int mine = (2 * q + w - 1) / w;
	const int regs = 16 * p * q + 4 * p * d + 4 * mine * d + 8 * q + 4 * p + 28;
	return regs <= 128 ? 4 : regs <= 168 ? 3 : 2;
}

// P channel PAIRS per wave, Q groups of four blocks (a launch folds nb <= 4 Q blocks), W waves per workgroup, D row trips of loads in
// flight.  A workgroup = one group of 64 bins x one slice of alias rows x 2 P W channels; its W waves cover the SAME bins and
// different channels, so each alias row's spectrum tile (4 Q blocks x 64 bins) is fetched ONCE per workgroup -- every wave loads
// 1 / W of it, already in operand-B order (lane 4 b + n = bin b of block n) -- written to LDS and read from there by all of them.
// Per alias row and wave: P tap loads of 1 KiB (non-temporal), 2 Q LDS reads, 4 P vector instructions (the rotated operand), and
// 8 P Q matrix instructions of 256 multiply-accumulates: first every accumulator's Re(X) product, then every Im(X) product, so two
// instructions on the same accumulator are 4 P Q instructions apart.  Loads run D rows ahead of the multiplies (registers are
// cheap here: the accumulators are the only large tile), the spectrum tile one row ahead through two LDS stages, one barrier per row.
template <int P, int Q, int W, int D>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(fold_mfma_waves(P, Q, W, D), fold_mfma_waves(P, Q, W, D)))) void fold_mfma_kernel(
		const float *__restrict__ taps, const float2 *__restrict__ spec, float2 *__restrict__ partial,
		size_t row_stride_f, size_t spec_stride, size_t partial_stride, int m, int slices, int rows, int pair_base, int nch, int nb)
{
	static_assert(D == 2 || D == 4, "the LDS stage of a trip is a compile-time constant for even D");
	constexpr int COMBOS = 2 * Q;                             // (block group, bin-set pair) pieces of a row's spectrum tile
	constexpr int MINE = (COMBOS + W - 1) / W;                // ... and this wave's share
	__shared__ v4f xt[2][Q][2][64];                           // [stage][block group][bin-set pair][lane] = (Re, Im) of bin-sets 2 vp, 2 vp + 1
	const int ngrp = m >> 6;
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
	const int b = lane >> 2, n = lane & 3;
	const int pair0 = pair_base + (grp * W + wave) * P;
	const int sign_mask = (lane & 1) ? 0 : (int)0x80000000;
	// wave-uniform bases stepped by scalar adds + one byte offset per lane
	const char *tb = (const char *)(taps + (size_t)s * rows * row_stride_f + (size_t)pair0 * 4 * m + (size_t)g * 256) + lane * 16;
	const size_t rs_b = row_stride_f * 4, ps_b = (size_t)m * 16, xrow_b = (size_t)m * 8;
	const char *xp[MINE];
#pragma unroll
	for (int i = 0; i < MINE; i++) {
		const int k = wave + i * W, q = k >> 1, vp = k & 1;
		int blk = 4 * q + n;
		blk = blk < nb ? blk : nb - 1;                        // columns past the last block repeat it; they are never stored
		xp[i] = (const char *)(spec + (size_t)blk * spec_stride + (size_t)s * rows * (size_t)m + g * 64 + 32 * vp + b);
	}
	v4f acc[P][4][Q];
#pragma unroll
	for (int p = 0; p < P; p++)
#pragma unroll
		for (int v = 0; v < 4; v++)
#pragma unroll
			for (int q = 0; q < Q; q++) acc[p][v][q] = (v4f)(0.f);
	v4f h[D][P], xs[D][MINE];
	auto issue = [&](int slot) {               // the loads of the next row not yet asked for: spectrum share first, then the taps
#pragma unroll
		for (int i = 0; i < MINE; i++)
			if (COMBOS % W == 0 || wave + i * W < COMBOS) {
				const v2f lo = *(const v2f *)xp[i], hi = *(const v2f *)(xp[i] + 128);
				xs[slot][i] = v4f{ lo.x, lo.y, hi.x, hi.y };
				xp[i] += xrow_b;
			}
#pragma unroll
		for (int p = 0; p < P; p++) h[slot][p] = __builtin_nontemporal_load((const v4f *)(tb + (size_t)p * ps_b));
		tb += rs_b;
	};
	auto stash = [&](int slot, int stage) {
#pragma unroll
		for (int i = 0; i < MINE; i++)
			if (COMBOS % W == 0 || wave + i * W < COMBOS) {
				const int k = wave + i * W;
				xt[stage][k >> 1][k & 1][lane] = xs[slot][i];
			}
	};
	auto multiply = [&](int slot, int stage) {
		v4f x[Q][2];
#pragma unroll
		for (int q = 0; q < Q; q++)
#pragma unroll
			for (int vp = 0; vp < 2; vp++) x[q][vp] = xt[stage][q][vp][lane];
#pragma unroll
		for (int p = 0; p < P; p++)
#pragma unroll
			for (int v = 0; v < 4; v++)
#pragma unroll
				for (int q = 0; q < Q; q++)
					acc[p][v][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(h[slot][p][v], x[q][v >> 1][(v & 1) * 2], acc[p][v][q], 0, 0, 0);
#pragma unroll
		for (int p = 0; p < P; p++)
#pragma unroll
			for (int v = 0; v < 4; v++) {
				const float hr = rot90(h[slot][p][v], sign_mask);
#pragma unroll
				for (int q = 0; q < Q; q++)
					acc[p][v][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(hr, x[q][v >> 1][(v & 1) * 2 + 1], acc[p][v][q], 0, 0, 0);
			}
	};
	// The scheduler must not reorder the loads of different rows: s_waitcnt counts loads in ISSUE order, and the count it is given at
	// the loop header is the smaller of what the prologue and the back edge allow -- a prologue that asks for row 0's taps last makes
	// every iteration wait for all but the last three loads, and the rows in flight are gone (measured: time = memory + multiplies).
#pragma unroll
	for (int d = 0; d < D; d++) {                      // rows 0 .. D-1 (rows is a multiple of D)
		issue(d);
		__builtin_amdgcn_sched_barrier(0);
	}
	stash(0, 0);
	__syncthreads();
	for (int r = 0; r < rows - D; r += D) {
#pragma unroll
		for (int d = 0; d < D; d++) {
			multiply(d, d & 1);
			issue(d);                                  // row r + d + D into the registers row r + d has just left
			stash((d + 1) % D, (d + 1) & 1);           // row r + d + 1 (asked for D - 1 trips ago) into the other stage
			__builtin_amdgcn_sched_barrier(0);
			__syncthreads();
		}
	}
#pragma unroll
	for (int d = 0; d < D; d++) {                      // the last D rows: nothing left to ask for
		multiply(d, d & 1);
		if (d + 1 < D) {
			stash(d + 1, (d + 1) & 1);
			__builtin_amdgcn_sched_barrier(0);
			__syncthreads();
		}
	}
#pragma unroll
	for (int p = 0; p < P; p++) {
		const int c0 = 2 * (pair0 + p);
#pragma unroll
		for (int q = 0; q < Q; q++) {
			const int blk = 4 * q + n;
			if (blk >= nb) continue;
#pragma unroll
			for (int v = 0; v < 4; v++) {
				float2 *po = partial + (size_t)blk * partial_stride + ((size_t)c0 * slices + s) * (size_t)m + g * 64 + 16 * v + b;
				const v4f a = acc[p][v][q];
				if (c0 < nch) *po = make_float2(a.x, a.y);
				if (c0 + 1 < nch) po[(size_t)slices * m] = make_float2(a.z, a.w);
			}
		}
	}
}

// Read-only streaming probe: what this board's HBM delivers to a bare kernel doing nothing but non-temporal 16-byte loads
// (1 KiB per wave instruction).  L loads in flight per thread; SPAN: every workgroup walks its own contiguous 4 MiB span,
// otherwise the whole grid sweeps one moving window (workgroup b reads chunks b, b + grid, ...).  bench.py prints the best of the
// variants as roofline.stream_read_GBs (SURVEY.md 8(d) asks for it next to the spec peak).  Laboratory build only.
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
template <int P, int W, int D, int FORM = 0>
static int fold16_go(const FoldArgs &a)
{
	const int ntile = (a.m >> 4) * a.slices;
	const int groups = a.ngroups / (P * W), rest = a.ngroups - groups * P * W;
	int launches = 0;
	if (groups > 0) {
		hipExtLaunchKernelGGL((fold_mfma16_kernel<P, W, D, false, FORM>), dim3((unsigned)(groups * ntile)), dim3(64 * W), 0, a.st, a.start, rest ? nullptr : a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, a.slices, a.rows, 0, a.nch, a.nb, (const int2 *)nullptr);
		launches++;
	}
	if (rest > 0) {
		hipExtLaunchKernelGGL((fold_mfma16_kernel<1, 1, D, false, FORM>), dim3((unsigned)(rest * ntile)), dim3(64), 0, a.st, groups > 0 ? nullptr : a.start, a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, a.slices, a.rows, groups * P * W, a.nch, a.nb, (const int2 *)nullptr);
		launches++;
	}
	return launches;
}

// the pruned fold: single-wave workgroups of two octets (their row window: win2), a left-over octet with its own (win1); one slice
template <int D>
static int fold16_go_win(const FoldArgs &a, const int2 *win2, const int2 *win1)
{
	const int ntile = a.m >> 4;
	const int groups = a.ngroups / 2, rest = a.ngroups - groups * 2;
	int launches = 0;
	if (groups > 0) {
		hipExtLaunchKernelGGL((fold_mfma16_kernel<2, 1, D, true>), dim3((unsigned)(groups * ntile)), dim3(64), 0, a.st, a.start, rest ? nullptr : a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, 1, a.rows, 0, a.nch, a.nb, win2);
		launches++;
	}
	if (rest > 0) {
		hipExtLaunchKernelGGL((fold_mfma16_kernel<1, 1, D, true>), dim3((unsigned)(rest * ntile)), dim3(64), 0, a.st, groups > 0 ? nullptr : a.start, a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, 1, a.rows, groups * 2, a.nch, a.nb, win1);
		launches++;
	}
	return launches;
}

#ifdef HFDL_LAB
// the 4x4x1 family (TAPL_PAIR): channel groups of 2 P W channels first; the pairs left over get single-wave workgroups
template <int P, int Q, int W, int D>
static int fold_go(const FoldArgs &a)
{
	const int ntile = (a.m >> 6) * a.slices;
	const int groups = a.ngroups / (P * W), rest = a.ngroups - groups * P * W;
	int launches = 0;
	if (groups > 0) {
		hipExtLaunchKernelGGL((fold_mfma_kernel<P, Q, W, D>), dim3((unsigned)(groups * ntile)), dim3(64 * W), 0, a.st, a.start, rest ? nullptr : a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, a.slices, a.rows, 0, a.nch, a.nb);
		launches++;
	}
	if (rest > 0) {
		hipExtLaunchKernelGGL((fold_mfma_kernel<1, Q, 1, 2>), dim3((unsigned)(rest * ntile)), dim3(64), 0, a.st, groups > 0 ? nullptr : a.start, a.stop, 0,
			a.taps, a.spec, a.partial, a.rs_f, a.ss, a.ps, a.m, a.slices, a.rows, groups * P * W, a.nch, a.nb);
		launches++;
	}
	return launches;
}
#endif

// layout: the tap layout the tiling reads; q: groups of four blocks it takes (4x4x1 family; 4 = any count up to 16 for the 16x16x1 family)
struct FoldVariant { int layout, p, q, w, d; int (*go)(const FoldArgs &); bool probe; };       // probe: times, does not fold (never picked)
#define F16(P, W, D) { TAPL_OCTET, P, 4, W, D, fold16_go<P, W, D> }
#define F4(P, W, D) { TAPL_OCTET, P, 1, W, D, fold16_go<P, W, D, 1> }
// laboratory, TIMING ONLY (the sums are wrong): the sixteen-column loop with v_mfma_f32_16x16x4_f32 in the place of 16x16x1_4B -- the same
// loads, LDS traffic and instruction count, a quarter of the accumulator traffic: what a K = 4 fold could gain
#define FK4(P, W, D) { TAPL_OCTET, P, 4, W, D, fold16_go<P, W, D, 2>, true }
#define FM(P, Q, W, D) { TAPL_PAIR, P, Q, W, D, fold_go<P, Q, W, D> }
// The first entry whose layout is the geometry's and whose row look-ahead D divides the slice is the one used.
// Measured on cfg3 (M = 4096, 512 rows per slice) with profiles/fold_variants.py: profiles/r05/fold_variants_cfg3.md.
static const FoldVariant fold_variants[] = {
	F16(2, 4, 4), F16(2, 4, 2),
	F4(4, 4, 4), F4(4, 4, 2),                  // launches of at most four blocks
#ifdef HFDL_LAB
	F4(2, 4, 4), F4(2, 4, 2), F4(2, 8, 4), F4(1, 4, 4), F4(4, 2, 4), F4(1, 8, 4),
	FK4(2, 4, 4), FK4(4, 4, 4), FK4(4, 4, 2), FK4(2, 8, 4),
	F16(1, 4, 4), F16(1, 4, 2), F16(1, 8, 4), F16(1, 8, 2), F16(2, 2, 4), F16(1, 2, 4), F16(2, 8, 2),
	// the 4x4x1 family of the first matrix-pipe build (HFDL_GPU_FOLD_MFMA=4 at create): measured, kept for the record
	FM(2, 1, 4, 4), FM(2, 1, 4, 2), FM(4, 1, 4, 4),
	FM(2, 2, 4, 4), FM(2, 2, 4, 2), FM(4, 2, 4, 4), FM(4, 2, 4, 2), FM(2, 2, 8, 2),
	FM(2, 4, 4, 2), FM(2, 4, 8, 2), FM(2, 4, 8, 4),
#endif
};
#undef FM
#undef FK4
#undef F4
#undef F16
constexpr int N_FOLD_VARIANTS = (int)(sizeof(fold_variants) / sizeof(fold_variants[0]));

int fold_variant_count() { return N_FOLD_VARIANTS; }

int fold_variant_describe(int v, int desc[6])
{
	if (v < 0 || v >= N_FOLD_VARIANTS) return -1;
	const FoldVariant &f = fold_variants[v];
	desc[0] = f.p; desc[1] = f.q; desc[2] = f.w; desc[3] = f.d; desc[4] = 4 * f.q; desc[5] = f.probe ? -f.layout : f.layout;
	return 0;
}

static bool variant_fits(const FoldVariant &f, const Geometry &g)
{
	return g.tap_layout == f.layout && g.rows_per_slice % f.d == 0;
}

// the tiling used for `nb` blocks of this geometry: the first entry of the list that fits
static const FoldVariant *pick_variant(const Geometry &g, int nb)
{
	// octet layout: the four-column form up to four blocks, the sixteen-column form beyond; pair layout (laboratory): 4, 8 or 16 columns
	const int q = nb <= 4 ? 1 : (nb <= 8 && g.tap_layout != TAPL_OCTET) ? 2 : 4;
	for (const FoldVariant &f : fold_variants)
		if (f.q == q && !f.probe && variant_fits(f, g)) return &f;
	for (const FoldVariant &f : fold_variants)
		if (4 * f.q >= nb && !f.probe && variant_fits(f, g)) return &f;
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
	for (int done = 0; done < nb;) {
		const int take = nb - done < nb_max ? nb - done : nb_max;
		const bool first = done == 0, last = done + take >= nb;
		const float2 *sp = spectrum + (size_t)done * spec_stride;
		float2 *pp = partial + (size_t)done * partial_stride;
		const FoldVariant *f = pick_variant(g, take);
		if (g.fold_win2 && g.tap_layout == TAPL_OCTET) launches += fold16_go_win<4>(fold_args(g, taps, sp, spec_stride, pp, partial_stride, take, st, first ? start : nullptr, last ? stop : nullptr), g.fold_win2, g.fold_win1);
		else if (f) launches += f->go(fold_args(g, taps, sp, spec_stride, pp, partial_stride, take, st, first ? start : nullptr, last ? stop : nullptr));
		else { launch_fold_ref(g, taps, sp, spec_stride, pp, partial_stride, take, st, first ? start : nullptr, last ? stop : nullptr); launches++; }
		done += take;
	}
	return launches;
}

// energy of every (alias row, channel) of the octet-interleaved taps: one workgroup per (row, octet) walks the row's M / 16 chunks of
// 1 KiB (lane = 2 (c % 8) + comp + 16 blk: the eight lanes of a channel add into one cell)
__global__ __launch_bounds__(FOLD_THREADS) void tap_row_energy_kernel(const float *__restrict__ taps, float *__restrict__ energy, size_t row_stride_f, int m, int noct, int nch_pad)
{
	const int oct = blockIdx.x % noct, row = blockIdx.x / noct;
	const v4f *base = (const v4f *)(taps + (size_t)row * row_stride_f + (size_t)oct * 16 * m);
	float acc = 0.f;
	for (int e = threadIdx.x; e < (m >> 4) * 64; e += FOLD_THREADS) {       // float4 index: chunk * 64 + lane
		const v4f v = base[e];
		acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
	}
	const int lane = threadIdx.x & 63;            // FOLD_THREADS is a multiple of 64: a thread keeps its lane position over the chunks
	atomicAdd(energy + (size_t)row * nch_pad + oct * 8 + ((lane & 15) >> 1), acc);
}

void launch_tap_row_energy(const float2 *taps, const Geometry &g, float *energy, hipStream_t st)
{
	const int noct = g.nch_pad / 8;
	hipLaunchKernelGGL(tap_row_energy_kernel, dim3((unsigned)(g.pre * noct)), dim3(FOLD_THREADS), 0, st, (const float *)taps, energy, (size_t)g.tap_row_stride * 2, g.m, noct, g.nch_pad);
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
