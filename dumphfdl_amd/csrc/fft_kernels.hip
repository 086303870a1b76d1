// fft_kernels.hip -- wideband forward FFT for the overlap-and-scrap channelizer (gfx950).
//
// Replaces fft_thread's overlap assembly + csdr_fft_execute(fwd) + fft_swap_sides
// (reference src/fft.c:49-59, src/fft_fftw.c:39-41, src/fastddc.c:102-112).
//
// N = R1*R2*R3 (each <= 256).  Three launches, each workgroup holds a tile of 16 columns x R points in
// LDS, runs an in-place radix-4 DIF (result bit-reversed, undone on the way out) and applies the
// inter-pass twiddle while storing.  Global accesses are 128-byte runs (16 adjacent columns of cf32);
// the overlap history and the fftshift are index remaps on the first load / last store, never a pass.
#include "kernels.h"
#include "fft_core.h"

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

// pass 1: columns c = n2*R3+n3 (stride R2*R3 between the R1 samples of a column).  Input is the
// virtual concatenation [hist(split) , fresh(n-split)] -- the overlap assembly of src/fft.c:49-54.
template <int FMT>
__global__ __launch_bounds__(FFT_THREADS) void fft_pass1(const float2 *__restrict__ hist, const void *__restrict__ fresh,
		int split, float2 *__restrict__ out, FftPlan p)
{
	extern __shared__ float2 sm[];
	const int cs = p.n >> p.l1;            // R2*R3 columns
	const int c0 = blockIdx.x * FFT_TILE;
	const int total = p.r1 * FFT_TILE;
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int r = e >> 4, col = e & 15;
		const int c = c0 + col;
		float2 v = make_float2(0.f, 0.f);
		if (c < cs) {
			const int idx = r * cs + c;
			v = idx < split ? hist[idx] : load_sample<FMT>(fresh, idx - split);
		}
		sm[e] = v;
	}
	__syncthreads();
	lds_fft_columns<-1>(sm, p.r1, p.l1, FFT_TILE, 4, p.tw1);
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int k1 = e >> 4, col = e & 15;
		const int c = c0 + col;
		if (c >= cs) continue;
		float2 v = sm[bitrev(k1, p.l1) * FFT_TILE + col];
		const unsigned ex = ((unsigned)k1 * (unsigned)c) & (unsigned)(p.n - 1);
		out[(size_t)k1 * cs + c] = cmul(v, unit_twiddle(ex, p.logn));
	}
}

// pass 2: for each (k1, n3): R2-point FFT over n2 (stride R3), twiddle W_{R2 R3}^{k2 n3}.  In place.
__global__ __launch_bounds__(FFT_THREADS) void fft_pass2(float2 *__restrict__ buf, FftPlan p)
{
	extern __shared__ float2 sm[];
	const int r23 = p.n >> p.l1, ncol = p.r1 * p.r3;
	const int cc0 = blockIdx.x * FFT_TILE;
	const int total = p.r2 * FFT_TILE;
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int r = e >> 4, col = e & 15;
		const int cc = cc0 + col;
		float2 v = make_float2(0.f, 0.f);
		if (cc < ncol) {
			const int k1 = cc >> p.l3, n3 = cc & (p.r3 - 1);
			v = buf[(size_t)k1 * r23 + (size_t)r * p.r3 + n3];
		}
		sm[e] = v;
	}
	__syncthreads();
	lds_fft_columns<-1>(sm, p.r2, p.l2, FFT_TILE, 4, p.tw2);
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int k2 = e >> 4, col = e & 15;
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
__global__ __launch_bounds__(FFT_THREADS) void fft_pass3(const float2 *__restrict__ in, float2 *__restrict__ out, FftPlan p, int shifted)
{
	extern __shared__ float2 sm[];
	const int r23 = p.n >> p.l1, ncol = p.r1 * p.r2;
	const int cc0 = blockIdx.x * FFT_TILE;
	const int total = p.r3 * FFT_TILE;
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int col = e >> p.l3, n3 = e & (p.r3 - 1);     // fast index walks a contiguous row
		const int cc = cc0 + col;
		float2 v = make_float2(0.f, 0.f);
		if (cc < ncol) {
			const int k2 = cc >> p.l1, k1 = cc & (p.r1 - 1);
			v = in[(size_t)k1 * r23 + (size_t)k2 * p.r3 + n3];
		}
		sm[n3 * FFT_TILE + col] = v;
	}
	__syncthreads();
	lds_fft_columns<-1>(sm, p.r3, p.l3, FFT_TILE, 4, p.tw3);
	const unsigned half = shifted ? (unsigned)(p.n >> 1) : 0u;
	for (int e = threadIdx.x; e < total; e += FFT_THREADS) {
		const int k3 = e >> 4, col = e & 15;
		const int cc = cc0 + col;
		if (cc >= ncol) continue;
		const int k2 = cc >> p.l1, k1 = cc & (p.r1 - 1);
		const unsigned k = (unsigned)k1 + ((unsigned)k2 << p.l1) + ((unsigned)k3 << (p.l1 + p.l2));
		out[(k + half) & (unsigned)(p.n - 1)] = sm[bitrev(k3, p.l3) * FFT_TILE + col];
	}
}

__global__ void copy_tail_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

void launch_fft_forward(const FftPlan &p, const float2 *hist, const void *fresh, int fmt, int split,
		float2 *work, float2 *out, bool shifted, hipStream_t st)
{
	const int c1 = (p.n >> p.l1), c2 = p.r1 * p.r3, c3 = p.r1 * p.r2;
	const dim3 g1((c1 + FFT_TILE - 1) / FFT_TILE), blk(FFT_THREADS);
	const size_t l1 = p.r1 * FFT_TILE * sizeof(float2);
	if (fmt == SFMT_CS16) hipLaunchKernelGGL(fft_pass1<SFMT_CS16>, g1, blk, l1, st, hist, fresh, split, work, p);
	else if (fmt == SFMT_CU8) hipLaunchKernelGGL(fft_pass1<SFMT_CU8>, g1, blk, l1, st, hist, fresh, split, work, p);
	else hipLaunchKernelGGL(fft_pass1<SFMT_CF32>, g1, blk, l1, st, hist, fresh, split, work, p);
	hipLaunchKernelGGL(fft_pass2, dim3((c2 + FFT_TILE - 1) / FFT_TILE), dim3(FFT_THREADS), p.r2 * FFT_TILE * sizeof(float2), st,
			work, p);
	hipLaunchKernelGGL(fft_pass3, dim3((c3 + FFT_TILE - 1) / FFT_TILE), dim3(FFT_THREADS), p.r3 * FFT_TILE * sizeof(float2), st,
			(const float2 *)work, out, p, shifted ? 1 : 0);
}

template <int FMT>
__global__ void convert_tail_kernel(const void *__restrict__ raw, float2 *__restrict__ dst, int first, int n)
{
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = load_sample<FMT>(raw, first + i);
}

// hist <- last `overlap` samples of this block's input (input_size >= overlap always: N >= 4*taps_length)
void launch_copy_tail(const void *fresh, int fmt, float2 *hist, int input_size, int overlap, hipStream_t st)
{
	const int first = input_size - overlap;
	if (fmt == SFMT_CS16) { hipLaunchKernelGGL(convert_tail_kernel<SFMT_CS16>, dim3(512), dim3(256), 0, st, fresh, hist, first, overlap); return; }
	if (fmt == SFMT_CU8) { hipLaunchKernelGGL(convert_tail_kernel<SFMT_CU8>, dim3(512), dim3(256), 0, st, fresh, hist, first, overlap); return; }
	const float2 *src = (const float2 *)fresh + first;
	if ((((uintptr_t)src | (uintptr_t)hist) & 15) == 0 && (overlap & 1) == 0) {
		size_t n4 = (size_t)overlap / 2;
		hipLaunchKernelGGL(copy_tail_kernel, dim3(512), dim3(256), 0, st, (const float4 *)src, (float4 *)hist, n4);
	} else {
		(void)hipMemcpyAsync(hist, src, sizeof(float2) * (size_t)overlap, hipMemcpyDeviceToDevice, st);
	}
}

}  // namespace hfdl
