// fold_kernels.hip -- per-channel spectrum x filter fold, inverse FFT, overlap scrap, NCO + decimate (gfx950).
//
// Replaces fastddc_inv_cc (reference src/fastddc.c:152-215): multiply_and_shift (:123-150), fft_swap_sides,
// the M-point backward FFT + normalisation (:190-197) and decimating_shift_addition_cc (src/libcsdr_gpl.c:41-74).
//
// fold_kernel is THE roofline kernel: per channel it streams N cf32 filter taps (distinct per channel, read once,
// 8*N bytes) against the shared N-bin spectrum and accumulates the N/M alias rows onto M bins:
//      Y_c[(h0 + j) mod M] = sum_a  H_c[a*M + j] * X[a*M + j]
// Workgroup = (channel pair, slice of alias rows, half of the M columns); a row is M contiguous cf32, so every wave
// issues 1 KiB dwordx4 runs; each spectrum value is loaded once and multiplied into both channels' tap streams.
// blockIdx -> (column half, slice, pair): the dispatcher puts block b on XCD b mod 8, so every XCD's L2 only ever
// sees 1/8 of the spectrum (two slices x one column half), shared by all channels.
#include <hip/hip_ext.h>
#include "kernels.h"
#include "fft_core.h"

namespace hfdl {

constexpr int FOLD_THREADS = 256;

typedef float v4f __attribute__((ext_vector_type(4)));

// streamed-once filter taps: non-temporal 16-byte load, so they do not evict the shared spectrum from L2
__device__ __forceinline__ float4 load_stream(const float4 *p)
{
	v4f v = __builtin_nontemporal_load((const v4f *)p);
	return make_float4(v.x, v.y, v.z, v.w);
}

// One complex multiply-accumulate per bin, spelled out as FMAs in a FIXED order so that every instantiation below -- any
// (U, CS, NC, NB) -- rounds a (block, channel, bin) sum exactly alike: a block folded alone and the same block folded beside three
// others give the same 32 bits (tests/test_gpu_parity.py::test_fold_batching_changes_nothing).
__device__ __forceinline__ void cmac2(v4f &a, const v4f h, const v4f x)
{
	a.x = __builtin_fmaf(h.x, x.x, a.x); a.x = __builtin_fmaf(-h.y, x.y, a.x);
	a.y = __builtin_fmaf(h.x, x.y, a.y); a.y = __builtin_fmaf(h.y, x.x, a.y);
	a.z = __builtin_fmaf(h.z, x.z, a.z); a.z = __builtin_fmaf(-h.w, x.w, a.z);
	a.w = __builtin_fmaf(h.z, x.w, a.w); a.w = __builtin_fmaf(h.w, x.z, a.w);
}

// U float4 (= 2U bins) per thread per alias row; R = alias rows per loop trip; CS = column split: a workgroup covers 1/CS of a
// row; NC = channels per thread sharing every spectrum load; NB = BLOCKS per launch sharing every tap load: the spectra of NB
// consecutive blocks (`spec_stride4` apart) are folded against ONE pass over the taps -- the taps are 99.9 % of a block's bytes and
// identical from block to block, so when blocks are queued (replay, catch-up, the bench) a launch serves NB of them for the HBM
// traffic of one (src/fastddc.c:123-150 run NB times).  Register tile: acc[NB][NC][U], an NB x NC outer product per column.
// WV = waves over channels: the workgroup's four wavefronts cover the SAME 64 * U columns and four different groups of NC channels,
// so one wave's spectrum loads fill the CU's L1 and the other three hit it: with NB blocks per launch the spectrum traffic out of
// L2 is NB / NC times the tap traffic (measured: what bounds the launch, profiles/r04_fold_variants.md), NB / (4 NC) this way.
// waves per SIMD the register tile allows (512 VGPRs per lane and SIMD).  Told to the compiler: with a bare __launch_bounds__(256) it
// aims at 8 waves per SIMD, squeezes the kernel into 64 VGPRs and gets there by issuing a load or two, waiting, multiplying, loading
// again -- one or two kilobytes in flight per wave where the trip has eight or more to ask for at once.
constexpr int fold_waves(int u, int r, int nc, int nb, bool lds = false)
{
	const int regs = 4 * (nb * nc * u + nc * r * u + nb * r * u) + (lds ? 16 : 24);
	return regs <= 128 ? 4 : regs <= 168 ? 3 : 2;
}

template <int U, int R, int CS, int NC, int NB, bool WV>
__global__ __launch_bounds__(FOLD_THREADS) __attribute__((amdgpu_waves_per_eu(fold_waves(U, R, NC, NB), fold_waves(U, R, NC, NB)))) void fold_kernel(
		const float4 *__restrict__ taps, const float4 *__restrict__ spec,
		float4 *__restrict__ partial, size_t chan_stride4, size_t row_stride4, size_t spec_stride4, size_t partial_stride4,
		int m, int slices, int rows, int c_base)
{
	constexpr int LANES = WV ? 64 : FOLD_THREADS;             // threads side by side along a row
	// blockIdx -> (tile = column part x slice, channel group), XCD-aware: block b runs on XCD b mod 8; a tile's workgroups all land on
	// one XCD and, there, the channel groups vary fastest, so the groups that share a tile of the spectra are resident together and the
	// tile crosses the fabric once (see fold_kernel_lds; with 8 tiles -- cfg3 at one block per launch -- this is the round-1 layout)
	const int ntile = CS * slices, groups = (int)gridDim.x / ntile;
	int tile_id, grp;
	if ((ntile & 7) == 0) {
		const int xcd = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
		grp = i % groups;
		tile_id = (i / groups) * 8 + xcd;
	} else {
		tile_id = (int)blockIdx.x % ntile;
		grp = (int)blockIdx.x / ntile;
	}
	const int cpart = tile_id % CS;
	const int wave = WV ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0, lane = WV ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
	const int s = tile_id / CS, c0 = c_base + (grp * (WV ? 4 : 1) + wave) * NC;
	const int row4 = m >> 1;                                  // float4 (= 2 bins) per alias row of the spectrum
	const size_t col = (size_t)cpart * U * LANES + lane;
	// Addresses = a wave-uniform base per stream (scalar registers, stepped by scalar adds) + ONE 32-bit byte offset per thread + an
	// immediate: a load costs no vector arithmetic, and the whole trip's loads go out back to back.
	const unsigned voff = (unsigned)lane * 16u;
	const char *tb = (const char *)(taps + (size_t)c0 * chan_stride4 + (size_t)s * rows * row_stride4 + (size_t)cpart * U * LANES);
	const char *sb = (const char *)(spec + (((size_t)s * rows * (size_t)m) >> 1) + (size_t)cpart * U * LANES);
	const size_t cs_b = chan_stride4 * 16, rs_b = row_stride4 * 16, ss_b = spec_stride4 * 16, r4_b = (size_t)row4 * 16;
	v4f acc[NB][NC][U];
#pragma unroll
	for (int b = 0; b < NB; b++)
#pragma unroll
		for (int k = 0; k < NC; k++)
#pragma unroll
			for (int u = 0; u < U; u++) acc[b][k][u] = (v4f)(0.f);
	const bool live = WV || (U * CS > 1) || ((int)threadIdx.x < row4);
	if (live) {
		for (int r = 0; r < rows; r += R) {
			v4f h[NC][R][U], x[NB][R][U];
#pragma unroll
			for (int q = 0; q < R; q++) {
#pragma unroll
				for (int u = 0; u < U; u++) {
#pragma unroll
					for (int k = 0; k < NC; k++)
						h[k][q][u] = __builtin_nontemporal_load((const v4f *)(tb + (size_t)k * cs_b + (size_t)q * rs_b + (size_t)voff + (size_t)(u * LANES * 16)));
#pragma unroll
					for (int b = 0; b < NB; b++)
						x[b][q][u] = *(const v4f *)(sb + (size_t)b * ss_b + (size_t)q * r4_b + (size_t)voff + (size_t)(u * LANES * 16));
				}
			}
#pragma unroll
			for (int q = 0; q < R; q++)       // rows strictly in order: the sum over a slice's rows is the same chain in every variant
#pragma unroll
				for (int b = 0; b < NB; b++)
#pragma unroll
					for (int k = 0; k < NC; k++)
#pragma unroll
						for (int u = 0; u < U; u++) cmac2(acc[b][k][u], h[k][q][u], x[b][q][u]);
			// the trip as the scheduler is to lay it out: every load first, then the multiplies as their operands arrive
			__builtin_amdgcn_sched_group_barrier(0x020, (NC + NB) * R * U, 0);
			__builtin_amdgcn_sched_group_barrier(0x002, NB * NC * U * R * 8, 0);
			tb += (size_t)R * rs_b;
			sb += (size_t)R * r4_b;
		}
#pragma unroll
		for (int b = 0; b < NB; b++)
#pragma unroll
			for (int k = 0; k < NC; k++) {
				v4f *po = (v4f *)partial + (size_t)b * partial_stride4 + (((size_t)(c0 + k) * slices + s) * (size_t)m >> 1) + col;
#pragma unroll
				for (int u = 0; u < U; u++) po[u * LANES] = acc[b][k][u];
			}
	}
}

// The same fold with the block spectra staged through LDS.  With NB blocks per launch every tap byte out of HBM wants NB / NC
// spectrum bytes out of L2, and the vector memory path (one 64-byte lane group per clock and CU, whatever level answers) is what
// bounds the launch from NB / NC = 1 up (profiles/r04_fold_variants.md: 2.5 ms at 0.5, 3.1 at 1, 4.3 at 2, 6 at 4).  Here the WPW
// wavefronts of a workgroup cover the SAME 64 * U columns and WPW different groups of NC channels: each alias row's spectrum tile
// (NB blocks x 64 * U columns) is fetched ONCE per workgroup -- every wave loads 1 / WPW of it -- written to LDS and read from
// there by all of them (ds_read_b128: 256 B per clock and CU, four times the vector memory path), so the spectrum costs
// NB / (WPW * NC) of the tap traffic.  Two LDS stages: the tile of trip t + 1 is fetched into registers while trip t is multiplied,
// stored to the other stage at the end of the trip, one workgroup barrier per trip.  Same FMA chain per bin as fold_kernel.
template <int U, int R, int NC, int NB, int WPW>
__global__ __launch_bounds__(64 * WPW) __attribute__((amdgpu_waves_per_eu(fold_waves(U, R, NC, NB, true), fold_waves(U, R, NC, NB, true)))) void fold_kernel_lds(
		const float4 *__restrict__ taps, const float4 *__restrict__ spec,
		float4 *__restrict__ partial, size_t chan_stride4, size_t row_stride4, size_t spec_stride4, size_t partial_stride4,
		int m, int slices, int rows, int c_base)
{
	constexpr int PIECES = NB * R * U;                         // 1 KiB wave-loads per spectrum tile
	constexpr int MINE = (PIECES + WPW - 1) / WPW;              // ... and this wave's share
	__shared__ v4f tile[2][PIECES][64];
	const int row4 = m >> 1;                                  // float4 (= 2 bins) per alias row of the spectrum
	const int cs = row4 / (64 * U);                           // column parts per row
	// blockIdx -> (tile = column part x slice, channel group).  The workgroups that share a spectrum tile must sit on ONE XCD (the
	// dispatcher puts block b on XCD b mod 8) and be resident TOGETHER, so that the tile comes out of HBM once and out of that XCD's
	// L2 for every other group: per XCD the channel groups vary fastest.  (Column part fastest -- the first layout -- kept only 4 of
	// the 16 groups of a tile resident at a time and the spectra crossed the fabric four times: 1.11 x the algorithmic bytes.)
	const int ntile = cs * slices, groups = (int)gridDim.x / ntile;
	int tile_id, grp;
	if ((ntile & 7) == 0) {
		const int xcd = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
		grp = i % groups;
		tile_id = (i / groups) * 8 + xcd;
	} else {
		tile_id = (int)blockIdx.x % ntile;
		grp = (int)blockIdx.x / ntile;
	}
	const int cpart = tile_id % cs;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
	const int s = tile_id / cs, c0 = c_base + (grp * WPW + wave) * NC;
	const size_t col = (size_t)cpart * U * 64 + lane;
	const unsigned voff = (unsigned)lane * 16u;
	const char *tb = (const char *)(taps + (size_t)c0 * chan_stride4 + (size_t)s * rows * row_stride4 + (size_t)cpart * U * 64);
	const char *sb = (const char *)(spec + (((size_t)s * rows * (size_t)m) >> 1) + (size_t)cpart * U * 64);
	const size_t cs_b = chan_stride4 * 16, rs_b = row_stride4 * 16, ss_b = spec_stride4 * 16, r4_b = (size_t)row4 * 16;
	v4f acc[NB][NC][U];
#pragma unroll
	for (int b = 0; b < NB; b++)
#pragma unroll
		for (int k = 0; k < NC; k++)
#pragma unroll
			for (int u = 0; u < U; u++) acc[b][k][u] = (v4f)(0.f);
	// piece p = (b, q, u) of a trip's tile; this wave fetches pieces wave, wave + WPW, ...
	v4f xs[MINE];
	auto fetch = [&](const char *base) {
#pragma unroll
		for (int i = 0; i < MINE; i++) {
			const int p = wave + i * WPW;
			if (PIECES % WPW == 0 || p < PIECES) {
				const int b = p / (R * U), q = (p / U) % R, u = p % U;
				xs[i] = *(const v4f *)(base + (size_t)b * ss_b + (size_t)q * r4_b + (size_t)voff + (size_t)(u * 64 * 16));
			}
		}
	};
	auto stash = [&](int stage) {
#pragma unroll
		for (int i = 0; i < MINE; i++) {
			const int p = wave + i * WPW;
			if (PIECES % WPW == 0 || p < PIECES) tile[stage][p][lane] = xs[i];
		}
	};
	auto load_taps = [&](v4f (&h)[NC][R][U], const char *base) {
#pragma unroll
		for (int q = 0; q < R; q++)
#pragma unroll
			for (int u = 0; u < U; u++)
#pragma unroll
				for (int k = 0; k < NC; k++)
					h[k][q][u] = __builtin_nontemporal_load((const v4f *)(base + (size_t)k * cs_b + (size_t)q * rs_b + (size_t)voff + (size_t)(u * 64 * 16)));
	};
	// One trip: ask for this wave's share of the NEXT trip's spectrum tile, then for this trip's taps (the tile answers out of L2,
	// ahead of the taps: loads return in order); read this trip's tile from LDS; store the fetched share to the other stage; multiply
	// as the taps arrive; barrier.  The fences keep the compiler from rotating the trip (it would issue the taps after the wait for
	// the tile share, one load latency after the other).
	fetch(sb);
	stash(0);
	__syncthreads();
	int stage = 0;
	for (int r = 0; r < rows; r += R) {
		const bool more = r + R < rows;
		v4f h[NC][R][U];
		sb += (size_t)R * r4_b;
		if (more) fetch(sb);
		load_taps(h, tb);
		asm volatile("" ::: "memory");
		v4f x[NB][R][U];
#pragma unroll
		for (int q = 0; q < R; q++)
#pragma unroll
			for (int b = 0; b < NB; b++)
#pragma unroll
				for (int u = 0; u < U; u++) x[b][q][u] = tile[stage][(b * R + q) * U + u][lane];
		if (more) stash(stage ^ 1);
		asm volatile("" ::: "memory");
#pragma unroll
		for (int q = 0; q < R; q++)           // rows strictly in order: the sum over a slice's rows is the same chain in every variant
#pragma unroll
			for (int b = 0; b < NB; b++)
#pragma unroll
				for (int u = 0; u < U; u++)
#pragma unroll
					for (int k = 0; k < NC; k++) cmac2(acc[b][k][u], h[k][q][u], x[b][q][u]);
		__syncthreads();
		stage ^= 1;
		tb += (size_t)R * rs_b;
	}
#pragma unroll
	for (int b = 0; b < NB; b++)
#pragma unroll
		for (int k = 0; k < NC; k++) {
			v4f *po = (v4f *)partial + (size_t)b * partial_stride4 + (((size_t)(c0 + k) * slices + s) * (size_t)m >> 1) + col;
#pragma unroll
			for (int u = 0; u < U; u++) po[u * 64] = acc[b][k][u];
		}
}

// Read-only streaming probe: what this board's HBM delivers to a bare kernel doing nothing but non-temporal 16-byte loads
// (1 KiB per wave instruction).  L loads in flight per thread; SPAN: every workgroup walks its own contiguous 4 MiB span,
// otherwise the whole grid sweeps one moving window (workgroup b reads chunks b, b + grid, ...).  The front end reports the
// best of the variants: SURVEY.md 8(d) asks for it next to the spec peak; bench.py prints it as roofline.stream_read_GBs.
template <int L, bool SPAN>
__global__ __launch_bounds__(FOLD_THREADS) void stream_read_kernel(const float4 *__restrict__ src, size_t chunks, float *__restrict__ sink)
{
	float acc = 0.f;
	const size_t per_block = chunks / gridDim.x;
	const size_t first = SPAN ? (size_t)blockIdx.x * per_block : blockIdx.x, step = SPAN ? 1 : gridDim.x;
	const size_t last = SPAN ? first + per_block : chunks;
	for (size_t ch = first; ch < last; ch += step) {
		const float4 *p = src + ch * (L * FOLD_THREADS) + threadIdx.x;
		float4 v[L];
#pragma unroll
		for (int i = 0; i < L; i++) v[i] = load_stream(p + i * FOLD_THREADS);
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

// generic fallback for row sizes that are not 512*2^k bins (same FMA chain per bin; one block per launch)
__global__ __launch_bounds__(FOLD_THREADS) void fold_kernel_generic(const float2 *__restrict__ taps, const float2 *__restrict__ spec,
		float2 *__restrict__ partial, size_t chan_stride, size_t row_stride, int m, int slices, int rows)
{
	const int s = blockIdx.x % slices, c = blockIdx.x / slices;
	for (int j = threadIdx.x; j < m; j += FOLD_THREADS) {
		const float2 *tp = taps + (size_t)c * chan_stride + (size_t)s * rows * row_stride + j;
		const float2 *sp = spec + (size_t)s * rows * (size_t)m + j;
		float2 acc = make_float2(0.f, 0.f);
		for (int r = 0; r < rows; r++) {
			float2 h = tp[(size_t)r * row_stride], x = sp[(size_t)r * m];
			acc.x = __builtin_fmaf(h.x, x.x, acc.x); acc.x = __builtin_fmaf(-h.y, x.y, acc.x);
			acc.y = __builtin_fmaf(h.x, x.y, acc.y); acc.y = __builtin_fmaf(h.y, x.x, acc.y);
		}
		partial[((size_t)c * slices + s) * (size_t)m + j] = acc;
	}
}

// ---- the compiled register tilings ----
struct FoldArgs {
	const float4 *taps, *spec;
	float4 *partial;
	size_t cs4, rs4, ss4, ps4;
	int m, slices, rows, nch;
	hipStream_t st;
	hipEvent_t start, stop;
};

// channel groups first (NC channels, or 4 * NC with the waves over channels); the channels left over get single-channel workgroups
// (one column range per workgroup) in a launch of their own
template <int U, int R, int CS, int NC, int NB, bool WV>
static int fold_go(const FoldArgs &a)
{
	const dim3 block(FOLD_THREADS);
	constexpr int GC = WV ? 4 * NC : NC;                  // channels per workgroup
	constexpr int RU = WV ? (U >= 4 ? U / 4 : 1) : U, RCS = WV ? (U >= 4 ? CS : CS * U / 4) : CS;      // the same row as RU * RCS * 256 columns
	const int groups = a.nch / GC, rest = a.nch - groups * GC;
	int launches = 0;
	if (groups > 0) {
		hipExtLaunchKernelGGL((fold_kernel<U, R, CS, NC, NB, WV>), dim3((unsigned)(groups * a.slices * CS)), block, 0, a.st, a.start, rest ? nullptr : a.stop, 0,
			a.taps, a.spec, a.partial, a.cs4, a.rs4, a.ss4, a.ps4, a.m, a.slices, a.rows, 0);
		launches++;
	}
	if (rest > 0) {
		hipExtLaunchKernelGGL((fold_kernel<RU, 1, RCS, 1, NB, false>), dim3((unsigned)(rest * a.slices * RCS)), block, 0, a.st, groups > 0 ? nullptr : a.start, a.stop, 0,
			a.taps, a.spec, a.partial, a.cs4, a.rs4, a.ss4, a.ps4, a.m, a.slices, a.rows, groups * GC);
		launches++;
	}
	return launches;
}

template <int U, int R, int NC, int NB, int WPW>
static int fold_go_lds(const FoldArgs &a)
{
	constexpr int GC = WPW * NC;                          // channels per workgroup
	const int cs = (a.m >> 1) / (64 * U);
	constexpr int RU = U >= 4 ? U / 4 : 1;                  // the channels left over: single-channel workgroups of the plain kernel
	const int groups = a.nch / GC, rest = a.nch - groups * GC;
	int launches = 0;
	if (groups > 0) {
		hipExtLaunchKernelGGL((fold_kernel_lds<U, R, NC, NB, WPW>), dim3((unsigned)(groups * a.slices * cs)), dim3(64 * WPW), 0, a.st, a.start, rest ? nullptr : a.stop, 0,
			a.taps, a.spec, a.partial, a.cs4, a.rs4, a.ss4, a.ps4, a.m, a.slices, a.rows, 0);
		launches++;
	}
	if (rest > 0) {
		const int rcs = a.m / (2 * FOLD_THREADS * RU);
		auto go = [&](auto kern, int cs_) {
			hipExtLaunchKernelGGL(kern, dim3((unsigned)(rest * a.slices * cs_)), dim3(FOLD_THREADS), 0, a.st, groups > 0 ? nullptr : a.start, a.stop, 0,
				a.taps, a.spec, a.partial, a.cs4, a.rs4, a.ss4, a.ps4, a.m, a.slices, a.rows, groups * GC);
		};
		switch (rcs) {          // CS is a template argument of the plain kernel
		case 1: go(fold_kernel<RU, 1, 1, 1, NB, false>, 1); break;
		case 2: go(fold_kernel<RU, 1, 2, 1, NB, false>, 2); break;
		case 4: go(fold_kernel<RU, 1, 4, 1, NB, false>, 4); break;
		case 8: go(fold_kernel<RU, 1, 8, 1, NB, false>, 8); break;
		default: go(fold_kernel<RU, 1, 16, 1, NB, false>, 16); break;
		}
		launches++;
	}
	return launches;
}

struct FoldVariant { int u, r, cs, nc, nb, wv; int (*go)(const FoldArgs &); };
#define FV(U, R, CS, NC, NB) { U, R, CS, NC, NB, 0, fold_go<U, R, CS, NC, NB, false> }
#define FW(U, R, CS, NC, NB) { U, R, CS, NC, NB, 1, fold_go<U, R, CS, NC, NB, true> }
// LDS-staged spectra: wv = 2 + waves per workgroup; `cs` is left 0 (the column split follows from M: M / (128 U) parts)
#define FL(U, R, NC, NB, WPW) { U, R, 0, NC, NB, 2 + WPW, fold_go_lds<U, R, NC, NB, WPW> }
// A row of M bins = U * CS * 512 (FV) or U * CS * 128 (FW: waves over channels).  Measured on cfg3 (M = 4096) with
// profiles/fold_variants.py: profiles/r04_fold_variants.md.
// What paid at one block per launch (profiles/r01_experiments.md): non-temporal tap loads (+7 %) and TWO channels per workgroup
// sharing every spectrum load (+14 %: halves the L2 -> L1 spectrum traffic).  With NB blocks per launch the spectrum traffic is
// NB / NC times the tap traffic, so the tile trades registers between the two (acc = 4 * NB * NC * U VGPRs).
static const FoldVariant fold_variants[] = {
	// The first entry of a block count that fits the geometry is the one used; the rest are kept for profiles/fold_variants.py
	// (measured on cfg3, M = 4096: profiles/r04_fold_variants.md).
	// one block per launch, M = 512 .. 8192: the round-1 tilings
	FV(1, 1, 1, 2, 1), FV(1, 1, 2, 2, 1), FV(2, 1, 2, 2, 1), FV(4, 1, 2, 2, 1), FV(8, 1, 2, 2, 1),
	FV(4, 2, 2, 2, 1), FW(4, 1, 8, 2, 1), FL(4, 1, 4, 1, 4), FL(4, 1, 2, 1, 8),
	// two blocks
	FV(1, 1, 1, 2, 2), FV(1, 1, 2, 2, 2), FV(1, 1, 4, 4, 2), FV(2, 1, 4, 4, 2), FV(4, 1, 4, 2, 2),
	FV(4, 1, 2, 2, 2), FW(2, 1, 16, 4, 2), FV(1, 1, 8, 8, 2), FL(4, 1, 2, 2, 4), FL(2, 2, 2, 2, 8),
	// four blocks: spectra through LDS where the slices are long enough (16 channels per workgroup), register tiles otherwise
	FL(2, 1, 4, 4, 4),
	FV(1, 1, 1, 2, 4), FV(1, 1, 2, 4, 4), FV(1, 1, 4, 8, 4), FV(1, 1, 8, 8, 4), FV(2, 1, 8, 4, 4),
	FL(1, 1, 4, 4, 8), FL(2, 1, 2, 4, 8), FL(1, 1, 4, 4, 4), FL(2, 1, 4, 4, 8), FV(2, 1, 4, 4, 4), FW(1, 1, 32, 8, 4), FW(2, 1, 16, 4, 4),
	// eight blocks
	FL(1, 1, 4, 8, 4),
	FV(1, 1, 1, 2, 8), FV(1, 1, 2, 4, 8), FV(1, 1, 4, 4, 8), FW(1, 1, 32, 4, 8), FV(1, 1, 16, 4, 8),
	FL(1, 1, 4, 8, 8), FL(1, 1, 2, 8, 8), FL(1, 2, 4, 8, 8), FL(1, 1, 2, 8, 4), FV(1, 1, 8, 4, 8), FV(1, 1, 8, 2, 8),
};
#undef FV
#undef FW
#undef FL
constexpr int N_FOLD_VARIANTS = (int)(sizeof(fold_variants) / sizeof(fold_variants[0]));

int fold_variant_count() { return N_FOLD_VARIANTS; }

int fold_variant_describe(int v, int desc[6])
{
	if (v < 0 || v >= N_FOLD_VARIANTS) return -1;
	const FoldVariant &f = fold_variants[v];
	desc[0] = f.u; desc[1] = f.r; desc[2] = f.cs; desc[3] = f.nc; desc[4] = f.nb; desc[5] = f.wv;
	return 0;
}

static bool variant_fits(const FoldVariant &f, const Geometry &g)
{
	// LDS-staged spectra: any M that is a multiple of the workgroup's 128 U bins, at least one full group of channels, and slices long
	// enough for the two-stage pipeline to matter (the small geometries' 16-row slices keep the register tiles)
	if (f.wv >= 2) return g.m % (128 * f.u) == 0 && g.rows_per_slice % f.r == 0 && g.rows_per_slice >= 64 * f.r && g.nch >= (f.wv - 2) * f.nc;
	return g.m == 2 * (f.wv ? 64 : FOLD_THREADS) * f.u * f.cs && g.rows_per_slice % f.r == 0;
}

// the tiling used for `nb` blocks of this geometry: the first entry of the preference list that fits (HFDL_GPU_FOLD_TILE =
// "U,R,CS,NC[,W]" overrides it for A/B measurements when such a variant is compiled)
static const FoldVariant *pick_variant(const Geometry &g, int nb)
{
	static int want[5] = { 0, 0, 0, 0, 0 };
	static bool parsed = false;
	if (!parsed) {
		parsed = true;
		if (const char *e = getenv("HFDL_GPU_FOLD_TILE"))
			if (sscanf(e, "%d,%d,%d,%d,%d", &want[0], &want[1], &want[2], &want[3], &want[4]) < 4) want[0] = 0;
	}
	if (want[0])
		for (const FoldVariant &f : fold_variants)
			if (f.nb == nb && f.u == want[0] && f.r == want[1] && f.cs == want[2] && f.nc == want[3] && f.wv == want[4] && variant_fits(f, g)) return &f;
	for (const FoldVariant &f : fold_variants)
		if (f.nb == nb && variant_fits(f, g)) return &f;
	return nullptr;
}

static FoldArgs fold_args(const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial, size_t partial_stride,
		hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
	FoldArgs a;
	a.taps = (const float4 *)taps; a.spec = (const float4 *)spectrum; a.partial = (float4 *)partial;
	a.cs4 = (size_t)g.tap_chan_stride >> 1; a.rs4 = (size_t)g.tap_row_stride >> 1; a.ss4 = spec_stride >> 1; a.ps4 = partial_stride >> 1;
	a.m = g.m; a.slices = g.slices; a.rows = g.rows_per_slice; a.nch = g.nch;
	a.st = st; a.start = start; a.stop = stop;
	return a;
}

int launch_fold_variant(int v, const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial,
		size_t partial_stride, hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
	if (v < 0 || v >= N_FOLD_VARIANTS || !variant_fits(fold_variants[v], g)) return -1;
	return fold_variants[v].go(fold_args(g, taps, spectrum, spec_stride, partial, partial_stride, st, start, stop));
}

int launch_fold(const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial, size_t partial_stride,
		int nb, int nb_max, hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
	int launches = 0;
	// greedy split into launches of 8 / 4 / 2 / 1 blocks (a ragged batch of 3 = 2 + 1): every block's sums are the same whatever its company
	for (int done = 0; done < nb;) {
		int take = 1;
		const FoldVariant *f = nullptr;
		for (int t = 8; t >= 1; t >>= 1)
			if (t <= nb - done && t <= nb_max && (f = pick_variant(g, t)) != nullptr) { take = t; break; }
		const bool first = done == 0, last = done + take >= nb;
		const float2 *sp = spectrum + (size_t)done * spec_stride;
		float2 *pp = partial + (size_t)done * partial_stride;
		if (f) {
			launches += f->go(fold_args(g, taps, sp, spec_stride, pp, partial_stride, st, first ? start : nullptr, last ? stop : nullptr));
		} else {
			hipExtLaunchKernelGGL(fold_kernel_generic, dim3((unsigned)(g.nch * g.slices)), dim3(FOLD_THREADS), 0, st, first ? start : nullptr, last ? stop : nullptr, 0,
				taps, sp, pp, (size_t)g.tap_chan_stride, (size_t)g.tap_row_stride, g.m, g.slices, g.rows_per_slice);
			launches++;
		}
		done += take;
	}
	return launches;
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
		size_t ph_stride, const float2 *tw_m, float2 *chan_out, int *out_count, int nb, hipStream_t st, hipEvent_t done)
{
	int logm = 0;
	while ((1 << logm) < g.m) logm++;
	size_t lds = sizeof(float2) * ((size_t)g.m + 1);
	hipExtLaunchKernelGGL(ifft_nco_kernel, dim3((unsigned)(g.nch * nb)), dim3(IFFT_THREADS), (unsigned)lds, st, nullptr, done, 0, partial, partial_stride, cc, snap, ph, ph_stride,
			tw_m, chan_out, out_count, g, logm);
}

}  // namespace hfdl
