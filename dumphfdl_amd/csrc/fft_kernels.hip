// fft_kernels.hip -- wideband forward FFT for the overlap-and-scrap channelizer (gfx950).
//
// Replaces fft_thread's overlap assembly + csdr_fft_execute(fwd) + fft_swap_sides
// (reference src/fft.c:49-59, src/fft_fftw.c:39-41, src/fastddc.c:102-112).
//
// N = R1*R2*R3 (each <= 256).  Three launches, each workgroup holds a tile of 16 columns x R points in
// LDS, runs an in-place radix-4 DIF (result bit-reversed, undone on the way out) and applies the
// inter-pass twiddle while storing.  Global accesses are 128-byte runs (16 adjacent columns of cf32);
// the overlap history and the fftshift are index remaps on the first load / last store, never a pass.
#include <hip/hip_ext.h>
#include "kernels.h"
#include "fft_core.h"
#include "fft_regs.h"

namespace hfdl {

// raw sample -> float complex with the reference's scaling: convert_cf32 / convert_cs16 / convert_cu8
// (src/input-helpers.c:10-78; full scale 1.0 / 32767.5 / 127, cu8 offset = full_scale / 2 as written there)
template <int FMT>
__device__ __forceinline__ float2 load_sample(const void *__restrict__ raw, int i)
{
	if (FMT == SFMT_CS16) {
		const short2 v = ((const short2 *)raw)[i];
		return make_float2((float)v.x / 32767.5f, (float)v.y / 32767.5f);
	} else if (FMT == SFMT_CU8) {
		const uchar2 v = ((const uchar2 *)raw)[i];
		return make_float2(((float)v.x - 63.5f) / 127.0f, ((float)v.y - 63.5f) / 127.0f);
	}
	return ((const float2 *)raw)[i];
}

// last-pass store: plain index, or (the transform is a channel's filter) straight into the matrix-operand tap layout
__device__ __forceinline__ void store_out(float2 *out, const FftOutLayout &lay, unsigned i, size_t at, float2 v)
{
	if (lay.kind == TAPL_PLAIN) { out[at] = v; return; }
	float *o = (float *)out + tap_index_f(lay.kind, 1 << lay.row_log, (size_t)(2 * lay.row_stride), lay.chan, (int)(i >> lay.row_log), (int)(i & ((1u << lay.row_log) - 1u)), 0);
	o[0] = v.x;
	o[4] = v.y;              // Im sits one lane on: four floats
}

// pass 1: columns c = n2*R3+n3 (stride R2*R3 between the R1 samples of a column).  Input is the
// virtual concatenation [hist(split) , fresh(n-split)] -- the overlap assembly of src/fft.c:49-54.
// The last `split` samples of the block are the next block's history: they are written to `hist_next` (a second buffer,
// never the one being read) as they pass through, so no separate copy runs.
template <int FMT>
__global__ __launch_bounds__(FFT_THREADS) void fft_pass1(const float2 *__restrict__ hist, const void *__restrict__ fresh,
		int split, float2 *__restrict__ hist_next, float2 *__restrict__ out, FftPlan p, NcoJob job, int riders)
{
	// rider workgroups come FIRST in the grid: dispatched at once, they run beside the whole pass (at the end of the grid they would start when the pass is nearly over)
	if ((int)blockIdx.x < riders) { nco_table_segment(job, (int)blockIdx.x * FFT_THREADS + (int)threadIdx.x); return; }
	const int bid = (int)blockIdx.x - riders;
	extern __shared__ float2 sm[];
	const int cs = p.n >> p.l1;            // R2*R3 columns
	const int c0 = bid * FFT_TILE;
	const int total = p.r1 * FFT_TILE;
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int r = e >> FFT_TILE_LOG, col = e & (FFT_TILE - 1);
		const int c = c0 + col;
		float2 v = make_float2(0.f, 0.f);
		if (c < cs) {
			const int idx = r * cs + c;
			v = idx < split ? hist[idx] : load_sample<FMT>(fresh, idx - split);
			if (hist_next != nullptr && idx >= p.n - split) hist_next[idx - (p.n - split)] = v;
		}
		sm[e] = v;
	}
	float2 *ltw = sm + total;                  // the radix's twiddle table next to the tile: no global load inside the stages
	for (int t = threadIdx.x; t < p.r1; t += FFT_THREADS) ltw[t] = p.tw1[t];
	__syncthreads();
	lds_fft_columns<-1>(sm, p.r1, p.l1, FFT_TILE, FFT_TILE_LOG, ltw);
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int k1 = e >> FFT_TILE_LOG, col = e & (FFT_TILE - 1);
		const int c = c0 + col;
		if (c >= cs) continue;
		float2 v = sm[bitrev(k1, p.l1) * FFT_TILE + col];
		const unsigned ex = ((unsigned)k1 * (unsigned)c) & (unsigned)(p.n - 1);
		out[(size_t)k1 * cs + c] = cmul(v, unit_twiddle(ex, p.logn));
	}
}

// pass 2: for each (k1, n3): R2-point FFT over n2 (stride R3), twiddle W_{R2 R3}^{k2 n3}.  In place.
__global__ __launch_bounds__(FFT_THREADS) void fft_pass2(float2 *__restrict__ buf, FftPlan p, NcoJob job, int riders)
{
	// rider workgroups come FIRST in the grid: dispatched at once, they run beside the whole pass (at the end of the grid they would start when the pass is nearly over)
	if ((int)blockIdx.x < riders) { nco_table_segment(job, (int)blockIdx.x * FFT_THREADS + (int)threadIdx.x); return; }
	const int bid = (int)blockIdx.x - riders;
	extern __shared__ float2 sm[];
	const int r23 = p.n >> p.l1, ncol = p.r1 * p.r3;
	const int cc0 = bid * FFT_TILE;
	const int total = p.r2 * FFT_TILE;
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int r = e >> FFT_TILE_LOG, col = e & (FFT_TILE - 1);
		const int cc = cc0 + col;
		float2 v = make_float2(0.f, 0.f);
		if (cc < ncol) {
			const int k1 = cc >> p.l3, n3 = cc & (p.r3 - 1);
			v = buf[(size_t)k1 * r23 + (size_t)r * p.r3 + n3];
		}
		sm[e] = v;
	}
	float2 *ltw = sm + total;
	for (int t = threadIdx.x; t < p.r2; t += FFT_THREADS) ltw[t] = p.tw2[t];
	__syncthreads();
	lds_fft_columns<-1>(sm, p.r2, p.l2, FFT_TILE, FFT_TILE_LOG, ltw);
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int k2 = e >> FFT_TILE_LOG, col = e & (FFT_TILE - 1);
		const int cc = cc0 + col;
		if (cc >= ncol) continue;
		const int k1 = cc >> p.l3, n3 = cc & (p.r3 - 1);
		float2 v = sm[bitrev(k2, p.l2) * FFT_TILE + col];
		const unsigned ex = ((unsigned)k2 * (unsigned)n3) & (unsigned)(r23 - 1);
		buf[(size_t)k1 * r23 + (size_t)k2 * p.r3 + n3] = cmul(v, unit_twiddle(ex, p.l2 + p.l3));
	}
}

// pass 3: for each (k1,k2): contiguous R3-point FFT; X[k1 + R1 k2 + R1 R2 k3] stored at (k + n/2) mod n
// when `shifted` (fft_swap_sides as an index remap).  A tile is 16 adjacent k1 so stores stay 128-byte runs.
__global__ __launch_bounds__(FFT_THREADS) void fft_pass3(const float2 *__restrict__ in, float2 *__restrict__ out, FftPlan p, int shifted,
		FftOutLayout lay, NcoJob job, int riders)
{
	// rider workgroups come FIRST in the grid: dispatched at once, they run beside the whole pass (at the end of the grid they would start when the pass is nearly over)
	if ((int)blockIdx.x < riders) { nco_table_segment(job, (int)blockIdx.x * FFT_THREADS + (int)threadIdx.x); return; }
	const int bid = (int)blockIdx.x - riders;
	extern __shared__ float2 sm[];
	const int r23 = p.n >> p.l1, ncol = p.r1 * p.r2;
	const int cc0 = bid * FFT_TILE;
	const int total = p.r3 * FFT_TILE;
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int col = e >> p.l3, n3 = e & (p.r3 - 1);     // fast index walks a contiguous row
		const int cc = cc0 + col;
		float2 v = make_float2(0.f, 0.f);
		if (cc < ncol) {
			const int k2 = cc >> p.l1, k1 = cc & (p.r1 - 1);
			v = in[(size_t)k1 * r23 + (size_t)k2 * p.r3 + n3];
		}
		sm[n3 * FFT_TILE + ((col + n3) & (FFT_TILE - 1))] = v;      // row-skewed tile (fft_core.h): this transposing fill is conflict-free
	}
	float2 *ltw = sm + total;
	for (int t = threadIdx.x; t < p.r3; t += FFT_THREADS) ltw[t] = p.tw3[t];
	__syncthreads();
	lds_fft_columns<-1, true>(sm, p.r3, p.l3, FFT_TILE, FFT_TILE_LOG, ltw);
	const unsigned half = shifted ? (unsigned)(p.n >> 1) : 0u;
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int k3 = e >> FFT_TILE_LOG, col = e & (FFT_TILE - 1);
		const int cc = cc0 + col;
		if (cc >= ncol) continue;
		const int k2 = cc >> p.l1, k1 = cc & (p.r1 - 1);
		const unsigned k = (unsigned)k1 + ((unsigned)k2 << p.l1) + ((unsigned)k3 << (p.l1 + p.l2));
		const unsigned i = (k + half) & (unsigned)(p.n - 1);
		const size_t at = lay.row_log ? (size_t)(i >> lay.row_log) * (size_t)lay.row_stride + (i & ((1u << lay.row_log) - 1u)) : (size_t)i;
		const int r = bitrev(k3, p.l3);
		store_out(out, lay, i, at, sm[r * FFT_TILE + ((col + r) & (FFT_TILE - 1))]);
	}
}

// ---- the register-resident passes (fft_regs.h): R = 16 B points x 16 columns per workgroup of 16 B threads ----

// pass 1, fast form: same indexing as fft_pass1
// Every pass: a workgroup takes `tpw` consecutive tiles and asks for the NEXT tile's sixteen points per thread (registers) before it
// transforms the current one: beside a resident demodulator workgroup (118 KiB of a CU's LDS) only one or two of these workgroups fit
// a CU, and without the look-ahead their load -> transform -> store phases ran one after the other (pass 3: 97 us in the pipeline
// against 35 us alone, profiles/r05_experiments.md).
template <int B, int FMT, bool AHEAD>
__global__ __launch_bounds__(16 * B) void fft_rpass1(const float2 *__restrict__ hist, const void *__restrict__ fresh,
		int split, float2 *__restrict__ hist_next, float2 *__restrict__ out, FftPlan p, NcoJob job, int riders, int tpw)
{
	constexpr int NT = 16 * B;
	if ((int)blockIdx.x < riders) { nco_table_segment(job, (int)blockIdx.x * NT + (int)threadIdx.x); return; }
	__shared__ float2 ex[B * FR_PITCH];
	__shared__ float2 ltw[16 * B];
	const int bid = ((int)blockIdx.x - riders) * tpw, tid = (int)threadIdx.x;
	const int cs = p.n >> p.l1;            // R2*R3 columns
	const int col = tid & 15, n2 = tid >> 4;
	float2 x[16], xn[16];
	auto load = [&](float2 (&v)[16], int tile) {
		const int c = tile * 16 + col;
#pragma unroll
		for (int n1 = 0; n1 < 16; n1++) {
			const int idx = (n2 + B * n1) * cs + c;
			v[n1] = idx < split ? hist[idx] : load_sample<FMT>(fresh, idx - split);
		}
	};
	load(x, bid);
	ltw[tid] = p.tw1[tid];
	for (int t = 0; t < (AHEAD ? tpw : 1); t++) {
		const int tile = bid + t;
		if (AHEAD && t + 1 < tpw) load(xn, tile + 1);
		if (hist_next != nullptr) {
			const int c = tile * 16 + col;
#pragma unroll
			for (int n1 = 0; n1 < 16; n1++) {
				const int idx = (n2 + B * n1) * cs + c;
				if (idx >= p.n - split) hist_next[idx - (p.n - split)] = x[n1];
			}
		}
		__syncthreads();                               // the twiddle table (first tile); the exchange buffer free again (later ones)
		stage_a(x, ex, ltw, n2, col);
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 256 / NT; i++) {
			const int tt = tid + i * NT, tcol = tt & 15, k1 = tt >> 4, tc = tile * 16 + tcol;
			float2 z[B], w[B];
			stage_b<B>(z, ex, k1, tcol);
			twiddle_run<B>(w, (unsigned)k1 * (unsigned)tc, 16u * (unsigned)tc, p.logn);
#pragma unroll
			for (int k2 = 0; k2 < B; k2++) out[(size_t)(k1 + 16 * k2) * cs + tc] = cmul(z[slot_small<B>(k2)], w[k2]);
		}
		if (AHEAD) {
#pragma unroll
			for (int n1 = 0; n1 < 16; n1++) x[n1] = xn[n1];
		}
	}
}

// pass 2, fast form: in place on buf; for each (k1, n3): R2-point FFT over n2 (stride R3), twiddle W_{R2 R3}^{k2 n3}
template <int B, bool AHEAD>
__global__ __launch_bounds__(16 * B) void fft_rpass2(float2 *__restrict__ buf, FftPlan p, NcoJob job, int riders, int tpw)
{
	constexpr int NT = 16 * B;
	if ((int)blockIdx.x < riders) { nco_table_segment(job, (int)blockIdx.x * NT + (int)threadIdx.x); return; }
	__shared__ float2 ex[B * FR_PITCH];
	__shared__ float2 ltw[16 * B];
	const int bid = ((int)blockIdx.x - riders) * tpw, tid = (int)threadIdx.x;
	const int r23 = p.n >> p.l1;
	const int col = tid & 15, n2 = tid >> 4;
	auto base_of = [&](int tile) {                            // 16 adjacent n3 of one k1 (R3 is a multiple of 16)
		const int cc0 = tile * 16;
		return buf + (size_t)(cc0 >> p.l3) * r23 + (cc0 & (p.r3 - 1));
	};
	float2 x[16], xn[16];
	auto load = [&](float2 (&v)[16], int tile) {
		const float2 *base = base_of(tile);
#pragma unroll
		for (int n1 = 0; n1 < 16; n1++) v[n1] = base[(size_t)(n2 + B * n1) * p.r3 + col];
	};
	load(x, bid);
	ltw[tid] = p.tw2[tid];
	for (int t = 0; t < (AHEAD ? tpw : 1); t++) {
		const int tile = bid + t;
		if (AHEAD && t + 1 < tpw) load(xn, tile + 1);
		float2 *base = base_of(tile);
		const int n30 = (tile * 16) & (p.r3 - 1);
		__syncthreads();
		stage_a(x, ex, ltw, n2, col);
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 256 / NT; i++) {
			const int tt = tid + i * NT, tcol = tt & 15, k1 = tt >> 4, n3 = n30 + tcol;
			float2 z[B], w[B];
			stage_b<B>(z, ex, k1, tcol);
			twiddle_run<B>(w, (unsigned)k1 * (unsigned)n3, 16u * (unsigned)n3, p.l2 + p.l3);
#pragma unroll
			for (int k2 = 0; k2 < B; k2++) base[(size_t)(k1 + 16 * k2) * p.r3 + tcol] = cmul(z[slot_small<B>(k2)], w[k2]);
		}
		if (AHEAD) {
#pragma unroll
			for (int n1 = 0; n1 < 16; n1++) x[n1] = xn[n1];
		}
	}
}

// pass 3, fast form: contiguous R3-point FFTs of 16 adjacent k1; the tile is filled row-skewed from 1 KiB runs (fft_pass3), the
// sixteen points of a thread come out of LDS, and X[k1 + R1 k2 + R1 R2 k3] goes to (k + n/2) mod n when `shifted`
template <int B>
__global__ __launch_bounds__(16 * B) void fft_rpass3(const float2 *__restrict__ in, float2 *__restrict__ out, FftPlan p, int shifted,
		FftOutLayout lay, NcoJob job, int riders, int tpw)
{
	constexpr int NT = 16 * B, R = 16 * B;
	if ((int)blockIdx.x < riders) { nco_table_segment(job, (int)blockIdx.x * NT + (int)threadIdx.x); return; }
	__shared__ float2 ex[B * FR_PITCH];                      // first the skewed input tile (R x 16 <= B x 272), then the exchange buffer
	__shared__ float2 ltw[16 * B];
	const int bid = ((int)blockIdx.x - riders) * tpw, tid = (int)threadIdx.x;
	const int r23 = p.n >> p.l1;
	const int col = tid & 15, n2 = tid >> 4;
	float2 pre[16], pn[16];                                  // the tile as it comes out of HBM: element tid + i NT of 16 rows x R points
	auto load = [&](float2 (&v)[16], int tile) {
		const int cc0 = tile * 16;                            // 16 adjacent k1 of one k2 (R1 is a multiple of 16)
		const float2 *src = in + (size_t)(cc0 >> p.l1) * R + (size_t)(cc0 & (p.r1 - 1)) * r23;
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const int e = tid + i * NT;                       // fast index walks a contiguous row (16 R elements, NT = R threads)
			v[i] = src[(size_t)(e / R) * r23 + (e % R)];
		}
	};
	load(pre, bid);
	ltw[tid] = p.tw3[tid];
	const unsigned half = shifted ? (unsigned)(p.n >> 1) : 0u;
	for (int t = 0; t < tpw; t++) {
		const int tile = bid + t;
		const int k2g = (tile * 16) >> p.l1, k10 = (tile * 16) & (p.r1 - 1);
		__syncthreads();                               // the exchange buffer of the tile before has been read
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const int e = tid + i * NT, tcol = e / R, n3 = e % R;
			ex[n3 * 16 + ((tcol + n3) & 15)] = pre[i];
		}
		if (t + 1 < tpw) load(pn, tile + 1);
		__syncthreads();
		float2 x[16];
#pragma unroll
		for (int n1 = 0; n1 < 16; n1++) {
			const int r = n2 + B * n1;
			x[n1] = ex[r * 16 + ((col + r) & 15)];
		}
		__syncthreads();
		stage_a(x, ex, ltw, n2, col);
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 256 / NT; i++) {
			const int tt = tid + i * NT, tcol = tt & 15, ka = tt >> 4;
			float2 z[B];
			stage_b<B>(z, ex, ka, tcol);
			const unsigned kbase = (unsigned)(k10 + tcol) + ((unsigned)k2g << p.l1);
#pragma unroll
			for (int kb = 0; kb < B; kb++) {
				const unsigned k = kbase + ((unsigned)(ka + 16 * kb) << (p.l1 + p.l2));
				const unsigned i2 = (k + half) & (unsigned)(p.n - 1);
				const size_t at = lay.row_log ? (size_t)(i2 >> lay.row_log) * (size_t)lay.row_stride + (i2 & ((1u << lay.row_log) - 1u)) : (size_t)i2;
				store_out(out, lay, i2, at, z[slot_small<B>(kb)]);
			}
		}
#pragma unroll
		for (int i = 0; i < 16; i++) pre[i] = pn[i];
	}
}

template <int B>
static void launch_rpass1(int fmt, int grid, hipStream_t st, hipEvent_t start, hipEvent_t input_read, const float2 *hist, const void *fresh, int split, float2 *hist_next,
		float2 *work, const FftPlan &p, const NcoJob &nco, int riders, int tpw)
{
	const dim3 blk(16 * B);
	if (fmt == SFMT_CS16) hipExtLaunchKernelGGL((fft_rpass1<B, SFMT_CS16, false>), dim3(grid), blk, 0, st, start, input_read, 0, hist, fresh, split, hist_next, work, p, nco, riders, tpw);
	else if (fmt == SFMT_CU8) hipExtLaunchKernelGGL((fft_rpass1<B, SFMT_CU8, false>), dim3(grid), blk, 0, st, start, input_read, 0, hist, fresh, split, hist_next, work, p, nco, riders, tpw);
	else hipExtLaunchKernelGGL((fft_rpass1<B, SFMT_CF32, false>), dim3(grid), blk, 0, st, start, input_read, 0, hist, fresh, split, hist_next, work, p, nco, riders, tpw);
}

// the register-resident passes take radices 64 / 128 / 256 (N = 2^18 .. 2^24); smaller transforms keep the LDS radix-4 passes
static bool fast_plan(const FftPlan &p) { return p.l1 >= 6 && p.l1 <= 8 && p.l2 >= 6 && p.l2 <= 8 && p.l3 >= 6 && p.l3 <= 8; }

void launch_fft_forward(const FftPlan &p, const float2 *hist, const void *fresh, int fmt, int split, float2 *hist_next,
		float2 *work, float2 *out, bool shifted, hipStream_t st, FftOutLayout lay, hipEvent_t done, NcoJob nco, hipEvent_t input_read, hipEvent_t start)
{
	const int c1 = (p.n >> p.l1), c2 = p.r1 * p.r3, c3 = p.r1 * p.r2;
	const int g1 = (c1 + FFT_TILE - 1) / FFT_TILE, g2 = (c2 + FFT_TILE - 1) / FFT_TILE, g3 = (c3 + FFT_TILE - 1) / FFT_TILE;
	nco.nseg = 3;
	if (fast_plan(p)) {
		// one rider workgroup per 16 B channels in front of each pass (a third of the NCO phasor table each)
		auto riders_of = [&](int l) { const int nt = 1 << l; return nco.cc ? (nco.nch + nt - 1) / nt : 0; };
		const int rd1 = riders_of(p.l1), rd2 = riders_of(p.l2), rd3 = riders_of(p.l3);
		// tiles per workgroup: 4 where that still leaves four workgroups per CU to hand out, else 2, else 1
		auto tpw_of = [](int tiles) { return tiles % 4 == 0 && tiles >= 4096 ? 4 : tiles % 2 == 0 && tiles >= 1024 ? 2 : 1; };
		// (passes 1 and 2 keep one tile per workgroup: their sixteen loads per thread go straight into the transform's registers, a second
		// set costs 90 - 140 VGPRs with the index arithmetic of the overlap assembly, and they lose little beside the demodulators)
		const int t1 = 1, t2 = 1, t3 = tpw_of(g3);
		nco.seg = 0;
		switch (p.l1) {
		case 6: launch_rpass1<4>(fmt, g1 / t1 + rd1, st, start, input_read, hist, fresh, split, hist_next, work, p, nco, rd1, t1); break;
		case 7: launch_rpass1<8>(fmt, g1 / t1 + rd1, st, start, input_read, hist, fresh, split, hist_next, work, p, nco, rd1, t1); break;
		default: launch_rpass1<16>(fmt, g1 / t1 + rd1, st, start, input_read, hist, fresh, split, hist_next, work, p, nco, rd1, t1); break;
		}
		nco.seg = 1;
		switch (p.l2) {
		case 6: hipLaunchKernelGGL((fft_rpass2<4, false>), dim3(g2 / t2 + rd2), dim3(64), 0, st, work, p, nco, rd2, t2); break;
		case 7: hipLaunchKernelGGL((fft_rpass2<8, false>), dim3(g2 / t2 + rd2), dim3(128), 0, st, work, p, nco, rd2, t2); break;
		default: hipLaunchKernelGGL((fft_rpass2<16, false>), dim3(g2 / t2 + rd2), dim3(256), 0, st, work, p, nco, rd2, t2); break;
		}
		nco.seg = 2;
		switch (p.l3) {
		case 6: hipExtLaunchKernelGGL(fft_rpass3<4>, dim3(g3 / t3 + rd3), dim3(64), 0, st, nullptr, done, 0, (const float2 *)work, out, p, shifted ? 1 : 0, lay, nco, rd3, t3); break;
		case 7: hipExtLaunchKernelGGL(fft_rpass3<8>, dim3(g3 / t3 + rd3), dim3(128), 0, st, nullptr, done, 0, (const float2 *)work, out, p, shifted ? 1 : 0, lay, nco, rd3, t3); break;
		default: hipExtLaunchKernelGGL(fft_rpass3<16>, dim3(g3 / t3 + rd3), dim3(256), 0, st, nullptr, done, 0, (const float2 *)work, out, p, shifted ? 1 : 0, lay, nco, rd3, t3); break;
		}
		return;
	}
	const int riders = nco.cc ? (nco.nch + FFT_THREADS - 1) / FFT_THREADS : 0;       // workgroups that run a third of the NCO phasor table each pass
	const dim3 blk(FFT_THREADS);
	// tile + the radix's twiddle table
	const size_t l1 = (size_t)p.r1 * (FFT_TILE + 1) * sizeof(float2), l2 = (size_t)p.r2 * (FFT_TILE + 1) * sizeof(float2);
	nco.seg = 0;
	if (fmt == SFMT_CS16) hipExtLaunchKernelGGL(fft_pass1<SFMT_CS16>, dim3(g1 + riders), blk, l1, st, start, input_read, 0, hist, fresh, split, hist_next, work, p, nco, riders);
	else if (fmt == SFMT_CU8) hipExtLaunchKernelGGL(fft_pass1<SFMT_CU8>, dim3(g1 + riders), blk, l1, st, start, input_read, 0, hist, fresh, split, hist_next, work, p, nco, riders);
	else hipExtLaunchKernelGGL(fft_pass1<SFMT_CF32>, dim3(g1 + riders), blk, l1, st, start, input_read, 0, hist, fresh, split, hist_next, work, p, nco, riders);
	nco.seg = 1;
	hipLaunchKernelGGL(fft_pass2, dim3(g2 + riders), blk, l2, st, work, p, nco, riders);
	nco.seg = 2;
	hipExtLaunchKernelGGL(fft_pass3, dim3(g3 + riders), blk, (size_t)p.r3 * (FFT_TILE + 1) * sizeof(float2), st, nullptr, done, 0,
			(const float2 *)work, out, p, shifted ? 1 : 0, lay, nco, riders);
}

}  // namespace hfdl
