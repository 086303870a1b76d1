// fft_core.h -- LDS-resident radix-4 FFT core shared by the wideband passes and the per-channel inverse FFT (gfx950).
#pragma once
#include "kernels.h"

namespace hfdl {

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-place radix-4 (+ final radix-2) DIF over `ncols` columns held as s[r * pitch + col].
// On exit position p of a column holds X[bitrev(p)].  DIR = -1 forward, +1 backward (unnormalised).
// tw[t] = exp(-2 pi i t / R).
// SKEW: the columns of row r are rotated by r inside the row (element (r, col) at r * pitch + ((col + r) & cols_mask), pitch = number
// of columns).  A row still fills one aligned run, so the stages' accesses (16 adjacent columns of a row per lane group) stay
// conflict-free, and a tile FILLED row-major along r (pass 3: consecutive lanes = consecutive r of one column, 128 bytes apart ->
// every ds_write_b64 lane group on one bank pair, a 16-way conflict) spreads over all banks.
template <int DIR, bool SKEW = false>
__device__ void lds_fft_columns(float2 *s, int R, int logR, int pitch, int log_cols, const float2 *tw)
{
	const int cols_mask = (1 << log_cols) - 1;
	int len = R, loglen = logR;
	while (len >= 4) {
		const int q = len >> 2, logq = loglen - 2;
		const int nb = (R >> 2) << log_cols;
		const int tstep = R >> loglen;       // W_len^j = tw[j * R/len]
		for (int t = threadIdx.x; t < nb; t += blockDim.x) {
			const int col = t & cols_mask, b = t >> log_cols;
			const int j = b & (q - 1), blk = b >> logq;
			const int r0 = (blk << loglen) + j;
			float2 *p = s + (size_t)r0 * pitch + (SKEW ? 0 : col);
			const int qs = q * pitch;
			// SKEW: the four rows r0 + k q hold the column at (col + r0 + k q) & mask
			const int c0 = SKEW ? ((col + r0) & cols_mask) : 0, c1 = SKEW ? ((col + r0 + q) & cols_mask) : qs;
			const int c2 = SKEW ? (2 * qs + ((col + r0 + 2 * q) & cols_mask)) : 2 * qs, c3 = SKEW ? (3 * qs + ((col + r0 + 3 * q) & cols_mask)) : 3 * qs;
			const int c1s = SKEW ? qs + c1 : c1;
			float2 a = p[c0], bb = p[c1s], c = p[c2], d = p[c3];
			float2 w1 = tw[j * tstep], w2 = tw[2 * j * tstep], w3 = tw[3 * j * tstep];
			if (DIR > 0) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
			float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
			float2 bpd = make_float2(bb.x + d.x, bb.y + d.y), bmd = make_float2(bb.x - d.x, bb.y - d.y);
			// DIR * j * (b - d)
			float2 jb = (DIR < 0) ? make_float2(bmd.y, -bmd.x) : make_float2(-bmd.y, bmd.x);
			float2 y0 = make_float2(apc.x + bpd.x, apc.y + bpd.y);
			float2 y2 = cmul(make_float2(apc.x - bpd.x, apc.y - bpd.y), w2);
			float2 y1 = cmul(make_float2(amc.x + jb.x, amc.y + jb.y), w1);
			float2 y3 = cmul(make_float2(amc.x - jb.x, amc.y - jb.y), w3);
			// order y0,y2,y1,y3 == two radix-2 DIF stages, so the final permutation is a plain bit reversal
			p[c0] = y0; p[c1s] = y2; p[c2] = y1; p[c3] = y3;
		}
		__syncthreads();
		len = q; loglen = logq;
	}
	if (len == 2) {
		const int nb = (R >> 1) << log_cols;
		for (int t = threadIdx.x; t < nb; t += blockDim.x) {
			const int col = t & cols_mask, b = t >> log_cols;
			float2 *p = s + (size_t)(2 * b) * pitch + (SKEW ? 0 : col);
			const int c0 = SKEW ? ((col + 2 * b) & cols_mask) : 0, c1 = pitch + (SKEW ? ((col + 2 * b + 1) & cols_mask) : 0);
			float2 a = p[c0], bb = p[c1];
			p[c0] = make_float2(a.x + bb.x, a.y + bb.y);
			p[c1] = make_float2(a.x - bb.x, a.y - bb.y);
		}
		__syncthreads();
	}
}

__device__ __forceinline__ int bitrev(int x, int bits) { return bits ? (int)(__brev((unsigned)x) >> (32 - bits)) : 0; }

// exp(-2 pi i e / n) for 0 <= e < n = 2^logn <= 2^24 : e/n is exact in fp32
__device__ __forceinline__ float2 unit_twiddle(unsigned e, int logn)
{
	float frac = (float)e * __builtin_ldexpf(1.0f, -logn + 1);      // 2e/n in [0,2)
	float sn, cs;
	sincospif(-frac, &sn, &cs);
	return make_float2(cs, sn);
}

// ---- decimating_shift_addition_cc's phasor recurrence (src/libcsdr_gpl.c:48-66), shared by every kernel that runs it ----
// fp32, products rounded separately: HIP's __fmul_rn / __fadd_rn are plain operators inside the headers and hipcc contracts
// a * b + c into an FMA by default, so they do NOT keep the products apart.  The pragma on plain operators written HERE does, as the
// reference's x86-64 build rounds them (found by feeding tests/golden/nco_ref.npz straight to the device).
__device__ __forceinline__ void nco_phasor_step(float &cphi, float &sphi, float cd, float sd)
{
#pragma clang fp contract(off)
	const float c0 = cphi, s0 = sphi;
	cphi = c0 * cd - s0 * sd;
	sphi = s0 * cd + c0 * sd;
}

// the reference seeds the recurrence with `float cosphi = cos(s.starting_phase)`: double-precision cos / sin, rounded to float
__device__ __forceinline__ float2 nco_phasor_seed(float starting_phase)
{
	return make_float2((float)cos((double)starting_phase), (float)sin((double)starting_phase));
}

__device__ __forceinline__ int nco_output_count(const NcoState &st, int input_size, int q)
{
	return st.decimation_remain < input_size ? (input_size - st.decimation_remain + q - 1) / q : 0;
}

// carried state after a block of `cnt` outputs (src/libcsdr_gpl.c:67-72): remainder of the decimation stride, phase advanced in
// double and wrapped to (-pi, pi], stored as float
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

// segment `job.seg` of the phasor table, lanes over channels: thread = channel.  Segment 0 snapshots the carried state for the
// block's inverse-FFT kernel, the last segment advances it for the next block (kernels.h NcoJob).
__device__ __forceinline__ void nco_table_segment(const NcoJob &job, int c)
{
	if (c >= job.nch) return;
	const ChanConst k = job.cc[c];
	NcoState st = job.chain[c];
	if (job.seg == 0) job.snap[c] = st;
	const int cnt = nco_output_count(st, job.post_input_size, job.post);
	const int per = (job.outs + job.nseg - 1) / job.nseg;
	const int i0 = job.seg * per, i1 = (job.seg + 1) * per < cnt ? (job.seg + 1) * per : cnt;
	float2 p = job.seg == 0 ? nco_phasor_seed(st.starting_phase) : job.cont[c];
	float2 *dst = job.ph + (size_t)i0 * job.nch + c;
	for (int i = i0; i < i1; i++, dst += job.nch) {
		*dst = p;
		nco_phasor_step(p.x, p.y, k.nco_cosdelta, k.nco_sindelta);
	}
	job.cont[c] = p;
	if (job.seg == job.nseg - 1) {
		nco_advance(st, cnt, job.post, job.post_input_size, k.nco_rate);
		job.chain[c] = st;
	}
}

}  // namespace hfdl
