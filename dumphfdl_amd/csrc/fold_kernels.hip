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

// U float4 (= 2U bins) per thread per alias row; NT = non-temporal tap loads; R = alias rows per loop trip;
// CS = column split: a workgroup covers 1/CS of a row; NC = channels per workgroup sharing every spectrum load
template <int U, bool NT, int R, int CS, int NC>
__global__ __launch_bounds__(FOLD_THREADS) void fold_kernel(const float4 *__restrict__ taps, const float4 *__restrict__ spec,
		float4 *__restrict__ partial, size_t chan_stride4, size_t row_stride4, int m, int slices, int rows, int c_base)
{
	const int cpart = blockIdx.x % CS;
	const int bs = blockIdx.x / CS;
	const int s = bs % slices, c0 = c_base + (bs / slices) * NC;
	const int row4 = m >> 1;                                  // float4 (= 2 bins) per alias row of the spectrum
	const size_t col = (size_t)cpart * U * FOLD_THREADS + threadIdx.x;
	const float4 *tp = taps + (size_t)c0 * chan_stride4 + (size_t)s * rows * row_stride4 + col;
	const float4 *sp = spec + (((size_t)s * rows * (size_t)m) >> 1) + col;
	const size_t cstride = chan_stride4;                      // float4 between consecutive channels' taps
	float4 acc[NC][U];
#pragma unroll
	for (int k = 0; k < NC; k++)
#pragma unroll
		for (int u = 0; u < U; u++) acc[k][u] = make_float4(0.f, 0.f, 0.f, 0.f);
	const bool live = (U * CS > 1) || ((int)threadIdx.x < row4);
	if (live) {
		for (int r = 0; r < rows; r += R) {
			float4 h[NC][R][U], x[R][U];
#pragma unroll
			for (int q = 0; q < R; q++) {
#pragma unroll
				for (int u = 0; u < U; u++) {
#pragma unroll
					for (int k = 0; k < NC; k++) {
						const float4 *p = tp + (size_t)k * cstride + (size_t)q * row_stride4 + u * FOLD_THREADS;
						h[k][q][u] = NT ? load_stream(p) : *p;
					}
					x[q][u] = sp[(size_t)q * row4 + u * FOLD_THREADS];
				}
			}
#pragma unroll
			for (int k = 0; k < NC; k++) {
#pragma unroll
				for (int q = 0; q < R; q++) {
#pragma unroll
					for (int u = 0; u < U; u++) {
						acc[k][u].x += h[k][q][u].x * x[q][u].x - h[k][q][u].y * x[q][u].y;
						acc[k][u].y += h[k][q][u].x * x[q][u].y + h[k][q][u].y * x[q][u].x;
						acc[k][u].z += h[k][q][u].z * x[q][u].z - h[k][q][u].w * x[q][u].w;
						acc[k][u].w += h[k][q][u].z * x[q][u].w + h[k][q][u].w * x[q][u].z;
					}
				}
			}
			tp += (size_t)R * row_stride4;
			sp += (size_t)R * row4;
		}
#pragma unroll
		for (int k = 0; k < NC; k++) {
			float4 *po = partial + (((size_t)(c0 + k) * slices + s) * (size_t)m >> 1) + (size_t)cpart * U * FOLD_THREADS + threadIdx.x;
#pragma unroll
			for (int u = 0; u < U; u++) po[u * FOLD_THREADS] = acc[k][u];
		}
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

// generic fallback for row sizes that are not 512*2^k bins
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
			acc.x += h.x * x.x - h.y * x.y;
			acc.y += h.x * x.y + h.y * x.x;
		}
		partial[((size_t)c * slices + s) * (size_t)m + j] = acc;
	}
}

void launch_fold(const Geometry &g, const float2 *taps, const float2 *spectrum, float2 *partial, hipStream_t st,
		hipEvent_t start, hipEvent_t stop)
{
	const dim3 block(FOLD_THREADS);
	const size_t cs4 = (size_t)g.tap_chan_stride >> 1, rs4 = (size_t)g.tap_row_stride >> 1;     // in float4
	const int u = g.m / (2 * FOLD_THREADS);
	const int pairs = g.nch / 2, odd = g.nch & 1;
	// channel pairs first; an odd last channel gets its own single-channel launch
#define FOLD_LAUNCH(U, CS, NC, GROUPS) do { \
	if ((GROUPS) > 0) hipExtLaunchKernelGGL((fold_kernel<U, true, 1, CS, NC>), dim3((unsigned)((GROUPS) * g.slices * CS)), block, 0, st, \
		start, odd ? nullptr : stop, 0, \
		(const float4 *)taps, (const float4 *)spectrum, (float4 *)partial, cs4, rs4, g.m, g.slices, g.rows_per_slice, 0); \
	if (odd) hipExtLaunchKernelGGL((fold_kernel<U, true, 1, CS, 1>), dim3((unsigned)(g.slices * CS)), block, 0, st, \
		(GROUPS) > 0 ? nullptr : start, stop, 0, \
		(const float4 *)taps, (const float4 *)spectrum, (float4 *)partial, cs4, rs4, g.m, g.slices, g.rows_per_slice, g.nch - 1); } while (0)
	// Variants measured on cfg3 (profiles/r01_experiments.md).  What pays: non-temporal tap loads (+7 %) and TWO channels per
	// workgroup sharing every spectrum load (+14 %: halves the L2->L1 spectrum traffic, which equals the HBM tap traffic
	// when each channel re-reads the spectrum).  A workgroup = (channel pair, slice, half of the columns).
	if (g.m == 2 * FOLD_THREADS * u && u >= 1) {
		switch (u) {
		case 1: FOLD_LAUNCH(1, 1, 2, pairs); return;
		case 2: FOLD_LAUNCH(1, 2, 2, pairs); return;
		case 4: FOLD_LAUNCH(2, 2, 2, pairs); return;
		case 8: FOLD_LAUNCH(4, 2, 2, pairs); return;
		case 16: FOLD_LAUNCH(8, 2, 2, pairs); return;
		default: break;
		}
	}
	hipExtLaunchKernelGGL(fold_kernel_generic, dim3((unsigned)(g.nch * g.slices)), block, 0, st, start, stop, 0, taps, spectrum, partial, (size_t)g.tap_chan_stride, (size_t)g.tap_row_stride, g.m, g.slices, g.rows_per_slice);
#undef FOLD_LAUNCH
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

// carried state after a block of `cnt` outputs (:67-72): remainder of the decimation stride, phase advanced in double and
// wrapped to (-pi, pi], stored as float
__device__ __forceinline__ void nco_advance(NcoState &st, int cnt, int q, int input_size, float rate)
{
	const int last = st.decimation_remain + q * cnt;
	st.decimation_remain = last - input_size;
	const double phase = (double)st.starting_phase + (double)rate * M_PI * (double)cnt;
	float fp = (float)phase;
	while ((double)fp > M_PI) fp = (float)((double)fp - 2 * M_PI);
	while ((double)fp < -M_PI) fp = (float)((double)fp + 2 * M_PI);
	st.starting_phase = fp;
	st.output_size = cnt;
}


__global__ __launch_bounds__(IFFT_THREADS) void ifft_nco_kernel(const float2 *__restrict__ partial, const ChanConst *__restrict__ cc,
		NcoState *__restrict__ nco, const float2 *__restrict__ ph, const float2 *__restrict__ tw, float2 *__restrict__ chan_out, int *__restrict__ out_count,
		Geometry g, int logm)
{
	extern __shared__ float2 sm[];          // m bins
	const int c = blockIdx.x;
	const ChanConst k = cc[c];
	const int m = g.m, mask = m - 1;
	NcoState st = nco[c];
	const int q = g.post;
	const int cnt = nco_output_count(st, g.post_input_size, q);
	{
		const int h0 = (int)(((long long)g.n - k.offsetbin + m / 2) % m);
		const float2 *pc = partial + (size_t)c * g.slices * (size_t)m;
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
	float2 *o = chan_out + (size_t)c * g.outs;
	for (int i = threadIdx.x; i < cnt; i += IFFT_THREADS) {
		const int idx = g.scrap + st.decimation_remain + q * i;
		float2 v = sm[(int)(__brev((unsigned)idx) >> (32 - logm))];
		v.x = __fdiv_rn(v.x, norm); v.y = __fdiv_rn(v.y, norm);
		o[i] = nco_rotate(ph[(size_t)i * g.nch + c], v);       // the block's phasor table, made beside its forward FFT (kernels.h NcoJob)
	}
	if (threadIdx.x == 0) {
		nco_advance(st, cnt, q, g.post_input_size, k.nco_rate);
		nco[c] = st;
		out_count[c] = cnt;      // per-buffer copy: the demodulator of this block may run while the next block updates nco[]
	}
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

void launch_ifft_nco(const Geometry &g, const float2 *partial, const ChanConst *cc, NcoState *nco, const float2 *ph,
		const float2 *tw_m, float2 *chan_out, int *out_count, hipStream_t st, hipEvent_t done)
{
	int logm = 0;
	while ((1 << logm) < g.m) logm++;
	size_t lds = sizeof(float2) * ((size_t)g.m + 1);
	hipExtLaunchKernelGGL(ifft_nco_kernel, dim3((unsigned)g.nch), dim3(IFFT_THREADS), (unsigned)lds, st, nullptr, done, 0, partial, cc, nco, ph, tw_m, chan_out, out_count, g, logm);
}

}  // namespace hfdl
