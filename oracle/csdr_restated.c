/*
 * csdr_restated.c -- ORACLE (test infrastructure only; see hfdl_oracle.h).
 *
 * Plain-C restatement of the reference's libcsdr / fastddc channelizer arithmetic:
 * geometry planner, windowed-sinc tap design, forward/backward DFT contract, the
 * spectrum x taps fold, and the NCO + decimator.  Every function cites the reference
 * lines it follows.  Float/double mixing follows the reference expression by expression,
 * because several results (startbin, tap phase accumulation, NCO recurrence) depend on it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "hfdl_oracle.h"

/* ---- small helpers (src/libcsdr.c:35-51,135-144) ---- */

int32_t orc_next_pow2(int32_t x)
{
	/* smallest power of two strictly greater than x */
	for (int32_t sh = 0; sh < 31; sh++) {
		int32_t p = (int32_t)1 << sh;
		if (x < p) return p;
	}
	return -1;
}

int32_t orc_firdes_filter_len(float transition_bw)
{
	int32_t len = (int32_t)(4.0 / transition_bw);
	return (len % 2 == 0) ? len + 1 : len;
}

float orc_transition_bw(int32_t fs, int32_t bw_hz)
{
	return (float)bw_hz / (float)fs;
}

int32_t orc_compute_fft_decimation_rate(int32_t fs, int32_t target)
{
	int32_t whole = (int32_t)floorf((float)fs / (float)target);
	return orc_next_pow2(whole) / 2;
}

/* ---- geometry planner (src/fastddc.c:46-80, src/libcsdr_gpl.c:26-39) ---- */

int orc_fastddc_init(orc_ddc *d, float transition_bw, int32_t decimation, float shift_rate)
{
	int32_t pre = 1, post = decimation;
	/* push factors of two into the frequency domain until post would become 1 */
	for (;;) {
		float half = (float)post / 2;
		if (floorf(half) != half || post / 2 == 1) break;
		post /= 2;
		pre *= 2;
	}
	d->pre_decimation = pre;
	d->post_decimation = post;
	d->taps_min_length = orc_firdes_filter_len(transition_bw);
	d->taps_length = orc_next_pow2((int32_t)(ceil(d->taps_min_length / (float)pre) * pre)) + 1;
	d->fft_size = orc_next_pow2(d->taps_length * 4);
	while (d->fft_size < pre) d->fft_size *= 2;
	d->overlap_length = d->taps_length - 1;
	d->input_size = d->fft_size - d->overlap_length;
	d->fft_inv_size = d->fft_size / pre;

	d->v = d->fft_size / d->overlap_length;
	int32_t middle = d->fft_size / 2;
	/* float arithmetic, truncated on assignment -- src/fastddc.c:66 */
	float sb = (float)middle + (float)middle * (-shift_rate) * 2;
	d->startbin = (int32_t)sb;
	d->startbin = (int32_t)(d->v * round(d->startbin / (float)d->v));
	d->offsetbin = d->startbin - middle;
	d->post_shift = pre * (shift_rate + ((float)d->offsetbin / d->fft_size));
	d->pre_shift = d->offsetbin / (float)d->fft_size;

	/* decimating_shift_addition_init(post_shift, post_decimation) */
	float rate = d->post_shift * post;
	rate *= 2;
	d->nco_sindelta = (float)sin(rate * M_PI);
	d->nco_cosdelta = (float)cos(rate * M_PI);
	d->nco_rate = rate;

	d->scrap = d->overlap_length / pre;
	d->post_input_size = d->fft_inv_size - d->scrap;
	return d->fft_size <= 2;
}

/* ---- tap design (src/libcsdr.c:62-68,83-133) ---- */

static float hamming_kernel(float rate)
{
	rate = 0.5 + rate / 2;
	return 0.54 - 0.46 * cos(2 * M_PI * rate);
}

void orc_firdes_lowpass_f(float *out, int32_t length, float cutoff)
{
	int32_t mid = length / 2;
	out[mid] = 2 * M_PI * cutoff * hamming_kernel(0);
	for (int32_t i = 1; i <= mid; i++) {
		float t = (sin(2 * M_PI * cutoff * i) / i) * hamming_kernel((float)i / mid);
		out[mid - i] = t;
		out[mid + i] = t;
	}
	float sum = 0;
	for (int32_t i = 0; i < length; i++) sum += out[i];
	for (int32_t i = 0; i < length; i++) out[i] = out[i] / sum;
}

void orc_firdes_bandpass_c(orc_cf *out, int32_t length, float lowcut, float highcut)
{
	float *lp = malloc(sizeof(float) * (size_t)length);
	orc_firdes_lowpass_f(lp, length, (highcut - lowcut) / 2);
	float center = (highcut + lowcut) / 2;
	float phase = 0;
	for (int32_t i = 0; i < length; i++) {
		float c = cos(phase), s = sin(phase);
		phase += 2 * M_PI * center;
		while (phase > 2 * M_PI) phase -= 2 * M_PI;
		while (phase < 0) phase += 2 * M_PI;
		out[i].re = c * lp[i];
		out[i].im = s * lp[i];
	}
	free(lp);
}

void orc_fft_swap_sides(orc_cf *io, int32_t n)
{
	int32_t half = n / 2;
	for (int32_t i = 0; i < half; i++) {
		orc_cf t = io[i];
		io[i] = io[i + half];
		io[i + half] = t;
	}
}

/* ---- DFT (contract of src/fft_fftw.c:22-41: unnormalised, FFTW_FORWARD = e^-, BACKWARD = e^+) ----
 * Stockham autosort radix-4 (+ one radix-2 pass when log2 n is odd), twiddles from a
 * double-precision table.  Written for this oracle; FFTW itself is not available here. */

#define TW_CACHE 8
static struct { int32_t n; double *tw; } tw_cache[TW_CACHE];
static pthread_mutex_t tw_lock = PTHREAD_MUTEX_INITIALIZER;

static const double *twiddles(int32_t n)
{
	/* fast path without the lock (entries are only ever added; tw is published before n): the threaded transform asks per column */
	for (int i = 0; i < TW_CACHE; i++)
		if (__atomic_load_n(&tw_cache[i].n, __ATOMIC_ACQUIRE) == n) return tw_cache[i].tw;
	pthread_mutex_lock(&tw_lock);
	int slot = -1;
	for (int i = 0; i < TW_CACHE; i++) {
		if (tw_cache[i].n == n) { pthread_mutex_unlock(&tw_lock); return tw_cache[i].tw; }
		if (tw_cache[i].n == 0 && slot < 0) slot = i;
	}
	double *tw = malloc(sizeof(double) * 2 * (size_t)n);
	for (int32_t k = 0; k < n; k++) {
		double a = -2.0 * M_PI * (double)k / (double)n;
		tw[2 * k] = cos(a);
		tw[2 * k + 1] = sin(a);
	}
	if (slot >= 0) { tw_cache[slot].tw = tw; __atomic_store_n(&tw_cache[slot].n, n, __ATOMIC_RELEASE); }   /* else: leaked, small */
	pthread_mutex_unlock(&tw_lock);
	return tw;
}

#define FFT_BODY(REAL, NAME)                                                                   \
static void NAME(REAL *a, REAL *b, int32_t n, int sign, REAL *out)                             \
{                                                                                              \
	const double *tw = twiddles(n);                                                            \
	REAL *x = a, *y = b;                                                                       \
	int32_t len = n, s = 1;                                                                    \
	const REAL sg = (REAL)sign; /* -1 forward, +1 backward */                                  \
	while (len >= 4) {                                                                         \
		int32_t q4 = len / 4, tstep = n / len;                                                 \
		for (int32_t p = 0; p < q4; p++) {                                                     \
			REAL w1r = (REAL)tw[2 * (size_t)(p * tstep)], w1i = -sg * (REAL)tw[2 * (size_t)(p * tstep) + 1];         \
			REAL w2r = (REAL)tw[2 * (size_t)(2 * p * tstep)], w2i = -sg * (REAL)tw[2 * (size_t)(2 * p * tstep) + 1]; \
			REAL w3r = (REAL)tw[2 * (size_t)(3 * p * tstep)], w3i = -sg * (REAL)tw[2 * (size_t)(3 * p * tstep) + 1]; \
			for (int32_t q = 0; q < s; q++) {                                                  \
				size_t i0 = 2 * ((size_t)q + (size_t)s * (size_t)p);                           \
				size_t st = 2 * (size_t)s * (size_t)q4;                                        \
				REAL ar = x[i0], ai = x[i0 + 1];                                               \
				REAL br = x[i0 + st], bi = x[i0 + st + 1];                                     \
				REAL cr = x[i0 + 2 * st], ci = x[i0 + 2 * st + 1];                             \
				REAL dr = x[i0 + 3 * st], di = x[i0 + 3 * st + 1];                             \
				REAL apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;           \
				REAL bpdr = br + dr, bpdi = bi + di;                                           \
				/* sign * j * (b - d) */                                                       \
				REAL jr = -sg * (bi - di), ji = sg * (br - dr);                                \
				size_t o0 = 2 * ((size_t)q + (size_t)s * (size_t)(4 * p));                     \
				size_t os = 2 * (size_t)s;                                                     \
				y[o0] = apcr + bpdr; y[o0 + 1] = apci + bpdi;                                  \
				REAL t1r = amcr + jr, t1i = amci + ji;                                         \
				y[o0 + os] = t1r * w1r - t1i * w1i; y[o0 + os + 1] = t1r * w1i + t1i * w1r;    \
				REAL t2r = apcr - bpdr, t2i = apci - bpdi;                                     \
				y[o0 + 2 * os] = t2r * w2r - t2i * w2i; y[o0 + 2 * os + 1] = t2r * w2i + t2i * w2r; \
				REAL t3r = amcr - jr, t3i = amci - ji;                                         \
				y[o0 + 3 * os] = t3r * w3r - t3i * w3i; y[o0 + 3 * os + 1] = t3r * w3i + t3i * w3r; \
			}                                                                                  \
		}                                                                                      \
		REAL *t = x; x = y; y = t;                                                             \
		len /= 4; s *= 4;                                                                      \
	}                                                                                          \
	if (len == 2) {                                                                            \
		for (int32_t q = 0; q < s; q++) {                                                      \
			size_t i0 = 2 * (size_t)q, i1 = 2 * ((size_t)q + (size_t)s);                       \
			REAL ar = x[i0], ai = x[i0 + 1], br = x[i1], bi = x[i1 + 1];                       \
			y[i0] = ar + br; y[i0 + 1] = ai + bi;                                              \
			y[i1] = ar - br; y[i1 + 1] = ai - bi;                                              \
		}                                                                                      \
		REAL *t = x; x = y; y = t;                                                             \
	}                                                                                          \
	if (x != out) memcpy(out, x, sizeof(REAL) * 2 * (size_t)n);                                \
}

FFT_BODY(float, fft_core_f32)
FFT_BODY(double, fft_core_f64)

void orc_fft_f32(const orc_cf *in, orc_cf *out, int32_t n, int sign)
{
	float *a = malloc(sizeof(float) * 2 * (size_t)n), *b = malloc(sizeof(float) * 2 * (size_t)n);
	memcpy(a, in, sizeof(float) * 2 * (size_t)n);
	if (n == 1) { out[0] = in[0]; free(a); free(b); return; }
	fft_core_f32(a, b, n, sign, (float *)out);
	free(a); free(b);
}

void orc_fft_f64(const double *in, double *out, int32_t n, int sign)
{
	double *a = malloc(sizeof(double) * 2 * (size_t)n), *b = malloc(sizeof(double) * 2 * (size_t)n);
	memcpy(a, in, sizeof(double) * 2 * (size_t)n);
	if (n == 1) { out[0] = in[0]; out[1] = in[1]; free(a); free(b); return; }
	fft_core_f64(a, b, n, sign, out);
	free(a); free(b);
}

/* ---- the forward FFT on several threads, TIMING ONLY (bench.py's cpu_baseline) ----
 * dumphfdl plans its forward FFT with fftwf_plan_with_nthreads(FFT_THREAD_CNT_DEFAULT = 4) (src/fft.h:15, src/fft_fftw.c:9-20).
 * FFTW is not available here; to give the CPU baseline the reference's threading shape the transform is split the six-step way,
 * n = n1 * n2:  X[k1 + n1 k2] = sum_n2 W_n2^(n2 k2) W_n^(n2 k1) sum_n1 x[n2 + n2_total n1] W_n1^(n1 k1),
 * columns and rows done by the single-thread kernel above on `threads` pthreads.  Same mathematics, another order of rounding:
 * the parity checks never use it (orc_fft_threads stays 1 unless the timing leg raises it). */
static int orc_fft_threads = 1;
void orc_set_fft_threads(int n) { orc_fft_threads = n < 1 ? 1 : (n > 64 ? 64 : n); }
int  orc_get_fft_threads(void) { return orc_fft_threads; }

struct mt_job { const float *in; float *mid, *out; int32_t n, n1, n2; int sign, phase, t, nt; };

#define MT_TILE 16      /* columns / rows handled together, so that every pass over the big arrays moves 128-byte runs */

static void *mt_worker(void *ctx)
{
	struct mt_job *j = ctx;
	const int32_t n1 = j->n1, n2 = j->n2;
	const double *tw = twiddles(j->n);
	const float sg = (float)j->sign;
	const int32_t len = j->phase == 0 ? n1 : n2;
	float *tile = malloc(sizeof(float) * 2 * (size_t)len * MT_TILE), *b = malloc(sizeof(float) * 2 * (size_t)len), *o = malloc(sizeof(float) * 2 * (size_t)len);
	if (j->phase == 0) {
		/* columns: for every n2, an n1-point transform over n1 (stride n2), times W_n^(n2 k1); result stored [k1][n2] */
		for (int32_t c0 = j->t * MT_TILE; c0 < n2; c0 += j->nt * MT_TILE) {
			for (int32_t r = 0; r < n1; r++)
				for (int32_t c = 0; c < MT_TILE; c++) {
					const size_t at = 2 * ((size_t)c0 + c + (size_t)n2 * r);
					tile[2 * ((size_t)c * n1 + r)] = j->in[at]; tile[2 * ((size_t)c * n1 + r) + 1] = j->in[at + 1];
				}
			for (int32_t c = 0; c < MT_TILE; c++) {
				float *col = tile + 2 * (size_t)c * n1;
				if (n1 > 1) { fft_core_f32(col, b, n1, j->sign, o); memcpy(col, o, sizeof(float) * 2 * (size_t)n1); }
			}
			for (int32_t k1 = 0; k1 < n1; k1++)
				for (int32_t c = 0; c < MT_TILE; c++) {
					/* W_n^e, e = n2 k1 (n is a power of two), as W_n^(e_hi * n2) * W_n^(e_lo): the two factors come from n1 + n2
					 * table entries that stay in cache, instead of one access scattered over the whole table per element */
					const size_t e = ((size_t)(c0 + c) * (size_t)k1) & (size_t)(j->n - 1);
					const size_t eh = (e / (size_t)n2) * (size_t)n2, el = e % (size_t)n2;
					const double ar = tw[2 * eh], ai = tw[2 * eh + 1], br = tw[2 * el], bi = tw[2 * el + 1];
					const float wr = (float)(ar * br - ai * bi), wi = -sg * (float)(ar * bi + ai * br);
					const float xr = tile[2 * ((size_t)c * n1 + k1)], xi = tile[2 * ((size_t)c * n1 + k1) + 1];
					j->mid[2 * ((size_t)k1 * n2 + c0 + c)] = xr * wr - xi * wi;
					j->mid[2 * ((size_t)k1 * n2 + c0 + c) + 1] = xr * wi + xi * wr;
				}
		}
	} else {
		/* rows: for every k1, an n2-point transform over n2 (contiguous); X[k1 + n1 k2] */
		for (int32_t r0 = j->t * MT_TILE; r0 < n1; r0 += j->nt * MT_TILE) {
			for (int32_t r = 0; r < MT_TILE; r++) {
				float *row = j->mid + 2 * (size_t)(r0 + r) * n2;
				if (n2 > 1) fft_core_f32(row, b, n2, j->sign, tile + 2 * (size_t)r * n2); else memcpy(tile + 2 * (size_t)r * n2, row, sizeof(float) * 2);
			}
			for (int32_t k2 = 0; k2 < n2; k2++)
				for (int32_t r = 0; r < MT_TILE; r++) {
					j->out[2 * ((size_t)r0 + r + (size_t)n1 * k2)] = tile[2 * ((size_t)r * n2 + k2)];
					j->out[2 * ((size_t)r0 + r + (size_t)n1 * k2) + 1] = tile[2 * ((size_t)r * n2 + k2) + 1];
				}
		}
	}
	free(tile); free(b); free(o);
	return NULL;
}

void orc_fft_f32_mt(const orc_cf *in, orc_cf *out, int32_t n, int sign, int threads)
{
	if (threads <= 1 || n < 4096) { orc_fft_f32(in, out, n, sign); return; }
	int lg = 0;
	while ((1 << lg) < n) lg++;
	const int32_t n1 = 1 << (lg / 2), n2 = n / n1;
	float *mid = malloc(sizeof(float) * 2 * (size_t)n);
	(void)twiddles(n); (void)twiddles(n1); (void)twiddles(n2);              /* built once, before the threads ask for them */
	pthread_t th[64];
	struct mt_job jobs[64];
	for (int phase = 0; phase < 2; phase++) {
		for (int t = 0; t < threads; t++) {
			jobs[t] = (struct mt_job){ (const float *)in, mid, (float *)out, n, n1, n2, sign, phase, t, threads };
			pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
		}
		for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
	}
	free(mid);
}

/* ---- per-channel frequency-domain taps (src/fastddc.c:217-252) ---- */

int orc_channelizer_taps(const orc_ddc *d, int32_t decimation, float freq_shift, orc_cf *taps_fft, int f64_fft)
{
	int32_t n = d->fft_size;
	orc_cf *taps = calloc((size_t)n, sizeof(orc_cf));
	if (!taps) return -1;
	float half_bw = 0.5f / decimation;
	orc_firdes_bandpass_c(taps, d->taps_length, (-freq_shift) - half_bw, (-freq_shift) + half_bw);
	if (f64_fft) {
		double *tin = malloc(sizeof(double) * 2 * (size_t)n), *tout = malloc(sizeof(double) * 2 * (size_t)n);
		for (int32_t i = 0; i < n; i++) { tin[2 * i] = taps[i].re; tin[2 * i + 1] = taps[i].im; }
		orc_fft_f64(tin, tout, n, -1);
		for (int32_t i = 0; i < n; i++) { taps_fft[i].re = (float)tout[2 * i]; taps_fft[i].im = (float)tout[2 * i + 1]; }
		free(tin); free(tout);
	} else {
		orc_fft_f32(taps, taps_fft, n, -1);
	}
	orc_fft_swap_sides(taps_fft, n);
	free(taps);
	return 0;
}

/* ---- fold (src/fastddc.c:114-150): out[(h0+i) mod m] += taps[i]*spectrum[i], ascending i ---- */

void orc_fold(const orc_cf *spectrum, const orc_cf *taps, int32_t n, orc_cf *out, int32_t m, int32_t offsetbin)
{
	int32_t o = (n - offsetbin + m / 2) % m;
	memset(out, 0, sizeof(orc_cf) * (size_t)m);
	int32_t i = 0;
	while (i < n) {
		int32_t run = m - o;
		if (run > n - i) run = n - i;
		const orc_cf *x = spectrum + i, *h = taps + i;
		orc_cf *y = out + o;
		for (int32_t k = 0; k < run; k++) {
			y[k].re += h[k].re * x[k].re - h[k].im * x[k].im;
			y[k].im += h[k].re * x[k].im + h[k].im * x[k].re;
		}
		i += run;
		o = 0;
	}
}

/* ---- NCO + decimator (src/libcsdr_gpl.c:41-74) ---- */

orc_nco_state orc_shift_decimate(const orc_cf *in, orc_cf *out, int32_t n, const orc_ddc *d, orc_nco_state s)
{
	float cphi = cos(s.starting_phase), sphi = sin(s.starting_phase);
	int32_t i, k = 0;
	for (i = s.decimation_remain; i < n; i += d->post_decimation) {
		out[k].re = cphi * in[i].re - sphi * in[i].im;
		out[k].im = sphi * in[i].re + cphi * in[i].im;
		k++;
		float c0 = cphi, s0 = sphi;
		cphi = c0 * d->nco_cosdelta - s0 * d->nco_sindelta;
		sphi = s0 * d->nco_cosdelta + c0 * d->nco_sindelta;
	}
	s.decimation_remain = i - n;
	s.starting_phase += d->nco_rate * M_PI * k;
	s.output_size = k;
	while (s.starting_phase > M_PI) s.starting_phase -= 2 * M_PI;
	while (s.starting_phase < -M_PI) s.starting_phase += 2 * M_PI;
	return s;
}

/* ---- fastddc_inv_cc (src/fastddc.c:152-215) ---- */

orc_nco_state orc_fastddc_inv(const orc_cf *spectrum, orc_cf *out, const orc_ddc *d, const orc_cf *taps_fft,
		orc_nco_state s, orc_cf *scratch)
{
	int32_t m = d->fft_inv_size;
	orc_cf *inv_in = scratch, *inv_out = scratch + m;
	orc_fold(spectrum, taps_fft, d->fft_size, inv_in, m, d->offsetbin);
	orc_fft_swap_sides(inv_in, m);
	orc_fft_f32(inv_in, inv_out, m, +1);
	float norm = (float)d->pre_decimation * (float)m;
	for (int32_t i = 0; i < m; i++) { inv_out[i].re /= norm; inv_out[i].im /= norm; }
	return orc_shift_decimate(inv_out + d->scrap, out, d->post_input_size, d, s);
}

/* ---- fft_thread body (src/fft.c:49-59) ---- */

void orc_forward_block(orc_cf *buf, const orc_cf *new_samples, const orc_ddc *d, orc_cf *spectrum)
{
	memmove(buf, buf + d->input_size, sizeof(orc_cf) * (size_t)d->overlap_length);
	memcpy(buf + d->overlap_length, new_samples, sizeof(orc_cf) * (size_t)d->input_size);
	orc_fft_f32_mt(buf, spectrum, d->fft_size, -1, orc_fft_threads);      /* 1 thread = orc_fft_f32: every parity check */
	orc_fft_swap_sides(spectrum, d->fft_size);
}
