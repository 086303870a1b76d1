/*
 * channel_restated.c -- ORACLE (test infrastructure only; see hfdl_oracle.h).
 *
 * Restatement of the per-channel HFDL demodulator thread (src/hfdl.c:593-935) and of the
 * liquid-dsp objects it drives.  liquid-dsp (>=1.3.0,<2.0.0) is a third-party dependency that
 * is neither vendored in /root/reference nor installed here: its published algorithms are
 * restated below with the constructor arguments dumphfdl uses (src/hfdl.c:468-534).
 * PARITY UNPINNED for every liquid object; the framer / Costas / sampler logic follows the
 * reference source line by line (cited inline).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "hfdl_oracle.h"

/* ======================= variant switches (hfdl_oracle.h) ======================= */

#include "../tests/hostsim/shared_math.h"     /* orc_variant.shared_math: libm replaced by fixed fp32 sequences (test infrastructure, like this file) */

/* the numeric constants of src/hfdl.c used below, by name, so that orc_constants() can hand them to the test that compares them with
 * the reference's text (tests/golden/hfdl_constants.json, tests/test_constants_cpu.py) */
#define CORR_THRESHOLD_A1 0.36f            /* src/hfdl.c:42-44 */
#define CORR_THRESHOLD_A2 0.3f
#define CORR_THRESHOLD_M1 0.3f
#define MAX_SEARCH_RETRIES 3               /* :45 */
#define NO_FRAME_TIMEOUT_FRAMES 13         /* :613 */
#define HFDL_SYMBOL_RATE 1800              /* src/hfdl.h:6-7 */
#define HFDL_SPS 3
#define COSTAS_ALPHA 0.1f                  /* :252 */
#define COSTAS_BETA_K 0.047f               /* :253: beta = 0.047 alpha^2 */
#define COSTAS_ERR_LIMIT 1.0f              /* :276 */
#define COSTAS_RUNAWAY_DPHI 0.25f          /* :709 */
#define AGC_BANDWIDTH 0.01f                /* :487 */
#define EQ_STEP 0.1f                       /* :496 */
#define SYMSYNC_LF_BW 0.001f               /* :504 */
#define NF_INIT 1.0f                       /* :490 */
#define NF_KEEP 0.65f                      /* :700-701 */
#define NF_TAKE 0.35f
#define NF_BIAS 1e-6f
#define NF_CLK_MASK 0xFFu                  /* :699 */
#define T_SEQ_BITS 0x9AFu                  /* T_seq[0] as bits, MSB (bit 14) first, 0 -> +1: :157-160 */
orc_variant orc_v = { .soft_dmin_init = 4.0f };
void orc_variant_default(orc_variant *v) { memset(v, 0, sizeof(*v)); v->soft_dmin_init = 4.0f; }
void orc_variant_set(const orc_variant *v) { orc_v = *v; }
void orc_variant_get(orc_variant *v) { *v = orc_v; }

/* ======================= liquid-dsp filter design ======================= */

static double bessel_i0(double z)
{
	/* I0(z) = sum_k ((z/2)^k / k!)^2 */
	double term = 1.0, sum = 1.0, hz = 0.5 * z;
	for (int k = 1; k < 64; k++) {
		term *= hz / k;
		double t2 = term * term;
		sum += t2;
		if (t2 < 1e-18 * sum) break;
	}
	return sum;
}

static double kaiser_beta_from_As(double As)
{
	As = fabs(As);
	if (As > 50.0) return 0.1102 * (As - 8.7);
	if (As > 21.0) return 0.5842 * pow(As - 21.0, 0.4) + 0.07886 * (As - 21.0);
	return 0.0;
}

static double sinc_pi(double x)
{
	if (fabs(x) < 1e-9) return 1.0;
	return sin(M_PI * x) / (M_PI * x);
}

/* variant design_float: the same design in single precision with liquid's own series -- liquid_besseli0f (32 terms through
 * lngammaf), sincf (product of three cosines below |x| = 0.01), kaiser_beta_As in float */
static float besseli0f_series(float z)
{
	if (z == 0.0f) return 1.0f;
	float y = 0.0f;
	for (int k = 0; k < 32; k++) {
		float t = (float)k * logf(0.5f * z) - lgammaf((float)k + 1.0f);
		y += expf(2 * t);
	}
	return y;
}

static float sincf_liquid(float x)
{
	if (fabsf(x) < 0.01f) return cosf((float)M_PI * x / 2.0f) * cosf((float)M_PI * x / 4.0f) * cosf((float)M_PI * x / 8.0f);
	return sinf((float)M_PI * x) / ((float)M_PI * x);
}

/* liquid_firdes_kaiser(n, fc, As, mu=0): h[i] = sinc(2 fc t) * kaiser(i, n, beta) */
static void firdes_kaiser(int n, double fc, double As, float *h)
{
	const double den = orc_v.kaiser_arg ? (double)(n - 1) : (double)n;
	if (orc_v.design_float) {
		float a = fabsf((float)As), beta;
		if (a > 50.0f) beta = 0.1102f * (a - 8.7f);
		else if (a > 21.0f) beta = 0.5842f * powf(a - 21.0f, 0.4f) + 0.07886f * (a - 21.0f);
		else beta = 0.0f;
		for (int i = 0; i < n; i++) {
			float t = (float)i - (float)(n - 1) / 2;
			float r = 2.0f * t / (float)den;
			float w = besseli0f_series(beta * sqrtf(1 - r * r)) / besseli0f_series(beta);
			h[i] = sincf_liquid(2.0f * (float)fc * t) * w;
		}
		return;
	}
	double beta = kaiser_beta_from_As(As), i0b = bessel_i0(beta);
	for (int i = 0; i < n; i++) {
		double t = (double)i - (double)(n - 1) / 2.0;
		double r = 2.0 * t / den;
		double w = bessel_i0(beta * sqrt(1.0 - r * r)) / i0b;
		h[i] = (float)(sinc_pi(2.0 * fc * t) * w);
	}
}

/* sum_k h[k] * w[k] (real taps, complex window): sequential, or (variant dot_order) even / odd partial sums as a 4-lane SIMD
 * dot product accumulates them */
static inline orc_cf dot_rc(const float *h, const orc_cf *w, int n)
{
	orc_cf y;
	if (!orc_v.dot_order) {
		float ar = 0, ai = 0;
		for (int k = 0; k < n; k++) { ar += h[k] * w[k].re; ai += h[k] * w[k].im; }
		y.re = ar; y.im = ai;
		return y;
	}
	float er = 0, ei = 0, or_ = 0, oi = 0;
	int k = 0;
	for (; k + 1 < n; k += 2) { er += h[k] * w[k].re; ei += h[k] * w[k].im; or_ += h[k + 1] * w[k + 1].re; oi += h[k + 1] * w[k + 1].im; }
	float ar = er + or_, ai = ei + oi;
	for (; k < n; k++) { ar += h[k] * w[k].re; ai += h[k] * w[k].im; }
	y.re = ar; y.im = ai;
	return y;
}

/* ======================= msresamp_crcf (a10) =======================
 * rate in (0.5,1): no half-band stage; one arbitrary resamp_crcf(rate, m=7,
 * fc=min(0.515 rate, 0.49), As=60, npfb=256) with 24-bit fixed-point phase. */

#define RS_NPFB 256
#define RS_TAPS 14

/* the prototype split into `npfb` branches of 14 taps; fc relative to the input rate */
static void resamp_design(int npfb, double fc, float *h)
{
	const int n = 2 * 7 * npfb + 1;
	float *hf = malloc(sizeof(float) * (size_t)n);
	firdes_kaiser(n, (float)fc / (float)npfb, 60.0, hf);
	float gain = 0.0f;
	for (int i = 0; i < n; i++) gain += hf[i];
	gain = (float)npfb / gain;
	/* polyphase split: branch b, tap k  <-  prototype[b + k*npfb] (prototype[n-1] unused) */
	for (int b = 0; b < npfb; b++)
		for (int k = 0; k < RS_TAPS; k++)
			h[b * RS_TAPS + k] = hf[b + k * npfb] * gain;
	free(hf);
}

void orc_resamp_filter(float rate, float *h, uint32_t *step)
{
	double fc = 0.515 * rate;
	if (fc > 0.49) fc = 0.49;
	resamp_design(RS_NPFB, fc, h);
	*step = (uint32_t)lround((double)(1u << 24) / (double)rate);
}

typedef struct {
	float h[RS_NPFB * RS_TAPS];
	orc_cf win[RS_TAPS];       /* win[0] newest */
	uint32_t step, phase;
	/* variant resamp_kind != 0 */
	int kind, npfb, bits;      /* bits: phase >> (24 - bits) = branch (fixed-phase kinds) */
	float del, tau, bf, mu;    /* float-phase kinds (liquid <= 1.3.1 resamp_crcf): tau in samples, bf = tau * npfb = b + mu */
	int b, boundary;
	orc_cf y0, y1;
} resamp_t;

static void resamp_init(resamp_t *r, float rate)
{
	memset(r, 0, sizeof(*r));
	r->kind = orc_v.resamp_kind;
	r->npfb = (r->kind == 1 || r->kind == 3) ? 64 : RS_NPFB;
	r->bits = r->npfb == 64 ? 6 : 8;
	double fc = 0.515 * rate;
	if (fc > 0.49) fc = 0.49;
	if (r->kind == 1) fc = 0.4;
	resamp_design(r->npfb, fc, r->h);
	r->step = (uint32_t)lround((double)(1u << 24) / (double)rate);
	r->del = 1.0f / rate;
}

/* liquid <= 1.3.1 resamp_crcf_execute: outputs interpolated linearly between the two branches around the float phase; the pair
 * (last branch, branch 0 of the next input sample) is finished when that sample arrives (RESAMP_STATE_BOUNDARY) */
static int32_t resamp_push_float(resamp_t *r, orc_cf x, orc_cf *y)
{
	int32_t n = 0;
	while (r->b < r->npfb) {
		if (r->boundary) {
			r->y1 = dot_rc(r->h, r->win, RS_TAPS);
			r->boundary = 0;
		} else {
			r->y0 = dot_rc(r->h + r->b * RS_TAPS, r->win, RS_TAPS);
			if (r->b == r->npfb - 1) { r->boundary = 1; r->b = r->npfb; break; }
			r->y1 = dot_rc(r->h + (r->b + 1) * RS_TAPS, r->win, RS_TAPS);
		}
		y[n].re = (1.0f - r->mu) * r->y0.re + r->mu * r->y1.re;
		y[n].im = (1.0f - r->mu) * r->y0.im + r->mu * r->y1.im;
		n++;
		r->tau += r->del;
		r->bf = r->tau * (float)r->npfb;
		r->b = (int)floorf(r->bf);
		r->mu = r->bf - (float)r->b;
	}
	r->tau -= 1.0f; r->bf -= (float)r->npfb; r->b -= r->npfb;
	(void)x;
	return n;
}

static int32_t resamp_push(resamp_t *r, orc_cf x, orc_cf *y)
{
	memmove(r->win + 1, r->win, sizeof(orc_cf) * (RS_TAPS - 1));
	r->win[0] = x;
	if (r->kind == 1 || r->kind == 2) return resamp_push_float(r, x, y);
	int32_t n = 0;
	while (r->phase < (1u << 24)) {
		const float *hb = r->h + (r->phase >> (24 - r->bits)) * RS_TAPS;
		y[n] = dot_rc(hb, r->win, RS_TAPS);
		n++;
		r->phase += r->step;
	}
	r->phase -= (1u << 24);
	return n;
}

int32_t orc_resamp_run(float rate, const orc_cf *x, int32_t n, orc_cf *y, uint32_t *phase_io, orc_cf *hist14)
{
	resamp_t *r = malloc(sizeof(*r));
	resamp_init(r, rate);
	r->phase = *phase_io;
	memcpy(r->win, hist14, sizeof(r->win));
	int32_t total = 0;
	for (int32_t i = 0; i < n; i++) total += resamp_push(r, x[i], y + total);
	*phase_io = r->phase;
	memcpy(hist14, r->win, sizeof(r->win));
	free(r);
	return total;
}

/* ======================= agc_crcf (a11) ======================= */

typedef struct { float g, y2, alpha; } agc_t;

static orc_cf agc_step(agc_t *a, orc_cf x)
{
	orc_cf y = { x.re * a->g, x.im * a->g };
	float e = y.re * y.re + y.im * y.im;
	if (orc_v.agc_double) a->y2 = (float)((1.0 - a->alpha) * a->y2 + a->alpha * e);      /* liquid writes (1.0 - alpha): a double expression */
	else a->y2 = (1.0f - a->alpha) * a->y2 + a->alpha * e;
	if (a->y2 > 1e-6f) a->g *= orc_v.shared_math ? sm_expf(-0.5f * a->alpha * sm_logf(a->y2)) : expf(-0.5f * a->alpha * logf(a->y2));
	if (a->g > 1e6f) a->g = 1e6f;
	return y;
}

/* ======================= symsync_crcf (a13) =======================
 * create_kaiser(k=3, m=3, beta=0.9, npfb=16): prototype of 2*16*3*3+1 = 289 taps,
 * fc = 0.75/(k*npfb), As = 40, scaled by 2*0.75; derivative filter by central differences
 * scaled to 0.06/max|h*dh|; 16 branches x 18 taps; loop filter from set_lf_bw(0.001);
 * output rate 2 samples/symbol. */

#define SS_NPFB 16
#define SS_TAPS 18
#define SS_K 3
#define SS_KOUT 2

void orc_symsync_filters(float *mf, float *dmf)
{
	enum { HL = 2 * SS_NPFB * SS_K * 3 + 1 };
	float hf[HL], H[HL], dH[HL];
	const float fc = 0.75f;
	firdes_kaiser(HL, fc / (float)(SS_K * SS_NPFB), 40.0, hf);
	for (int i = 0; i < HL; i++) H[i] = hf[i] * 2.0f * fc;
	float hdh_max = 0.0f;
	for (int i = 0; i < HL; i++) {
		if (i == 0) dH[i] = H[i + 1] - H[HL - 1];
		else if (i == HL - 1) dH[i] = H[0] - H[i - 1];
		else dH[i] = H[i + 1] - H[i - 1];
		float v = fabsf(H[i] * dH[i]);
		if (v > hdh_max || i == 0) hdh_max = v;
	}
	const float dscale = orc_v.symsync_dmf_scale > 0.f ? orc_v.symsync_dmf_scale : 1.0f;
	for (int i = 0; i < HL; i++) dH[i] *= dscale * 0.06f / hdh_max;
	for (int b = 0; b < SS_NPFB; b++)
		for (int k = 0; k < SS_TAPS; k++) {
			mf[b * SS_TAPS + k] = H[b + k * SS_NPFB];
			dmf[b * SS_TAPS + k] = dH[b + k * SS_NPFB];
		}
}

typedef struct {
	float mf[SS_NPFB * SS_TAPS], dmf[SS_NPFB * SS_TAPS];
	orc_cf win_mf[SS_TAPS], win_dmf[SS_TAPS];    /* [0] newest; reset() clears only the mf window */
	float rate, del, tau, bf, q, q_hat, rate_adjustment;
	int b;
	uint32_t decim_counter;
	float lf_b0, lf_a1, lf_a2, lf_v1, lf_v2;   /* iirfiltsos_rrrf, normalised by a0 */
	float lf_b1, lf_b2;
} symsync_t;

static void symsync_reset(symsync_t *s)
{
	memset(s->win_mf, 0, sizeof(s->win_mf));
	if (orc_v.symsync_reset_both) memset(s->win_dmf, 0, sizeof(s->win_dmf));
	s->rate = (float)SS_K / (float)SS_KOUT;
	s->del = s->rate;
	s->b = 0; s->bf = 0.0f; s->tau = 0.0f; s->q = 0.0f; s->q_hat = 0.0f;
	s->decim_counter = 0;
	s->lf_v1 = s->lf_v2 = 0.0f;
}

static void symsync_init(symsync_t *s, float lf_bw)
{
	memset(s, 0, sizeof(*s));
	orc_symsync_filters(s->mf, s->dmf);
	float alpha = 1.000f - lf_bw, beta = 0.220f * lf_bw, a = 0.500f, b = orc_v.symsync_lf_b > 0.f ? orc_v.symsync_lf_b : 0.495f;
	float B0 = beta, A0 = 1.00f - a * alpha, A1 = -b * alpha;
	s->lf_b0 = B0 / A0; s->lf_b1 = 0.0f; s->lf_b2 = 0.0f;
	s->lf_a1 = A1 / A0; s->lf_a2 = 0.0f;
	s->rate_adjustment = 0.5f * lf_bw;
	symsync_reset(s);
}

static orc_cf bank_dot(const float *h, const orc_cf *w)
{
	return dot_rc(h, w, SS_TAPS);
}

/* one input sample -> 0..2 outputs */
static int32_t symsync_step(symsync_t *s, orc_cf x, orc_cf *y)
{
	memmove(s->win_mf + 1, s->win_mf, sizeof(orc_cf) * (SS_TAPS - 1));
	s->win_mf[0] = x;
	memmove(s->win_dmf + 1, s->win_dmf, sizeof(orc_cf) * (SS_TAPS - 1));
	s->win_dmf[0] = x;
	int32_t n = 0;
	while (s->b < SS_NPFB) {
		orc_cf mf = bank_dot(s->mf + s->b * SS_TAPS, s->win_mf);
		y[n].re = mf.re / (float)SS_K;
		y[n].im = mf.im / (float)SS_K;
		if (s->decim_counter == SS_KOUT) {
			s->decim_counter = 0;
			orc_cf dmf = bank_dot(s->dmf + s->b * SS_TAPS, s->win_dmf);
			/* timing error Re{conj(mf) dmf}, clipped to +-1, through the loop filter (DF-II) */
			float q = mf.re * dmf.re + mf.im * dmf.im;
			if (q > 1.0f) q = 1.0f; else if (q < -1.0f) q = -1.0f;
			s->q = q;
			float v0 = q - s->lf_a1 * s->lf_v1 - s->lf_a2 * s->lf_v2;
			s->q_hat = s->lf_b0 * v0 + s->lf_b1 * s->lf_v1 + s->lf_b2 * s->lf_v2;
			s->lf_v2 = s->lf_v1; s->lf_v1 = v0;
			s->rate += s->rate_adjustment * s->q_hat;
			s->del = s->rate + s->q_hat;
		}
		s->decim_counter++;
		s->tau += s->del;
		s->bf = s->tau * (float)SS_NPFB;
		s->b = orc_v.symsync_bank_floor ? (int)floorf(s->bf) : (int)roundf(s->bf);
		n++;
	}
	s->tau -= 1.0f;
	s->bf -= (float)SS_NPFB;
	s->b -= SS_NPFB;
	return n;
}

/* ======================= eqlms_cccf (a15) ======================= */

#define EQ_LEN 15

void orc_eq_initial_taps(float *w)
{
	float h[EQ_LEN];
	firdes_kaiser(EQ_LEN, 0.45, 40.0, h);
	for (int i = 0; i < EQ_LEN; i++) w[i] = h[i] * 2.0f * 0.45f;
}

typedef struct {
	orc_cf w[EQ_LEN], h0[EQ_LEN];
	orc_cf buf[EQ_LEN];        /* [0] oldest */
	float x2[EQ_LEN];          /* |x|^2 of the same samples, [0] oldest */
	float x2_sum, mu;
	uint32_t count;
	int buf_full;
} eqlms_t;

static void eqlms_reset(eqlms_t *e)
{
	memcpy(e->w, e->h0, sizeof(e->w));
	memset(e->buf, 0, sizeof(e->buf));
	memset(e->x2, 0, sizeof(e->x2));
	e->x2_sum = 0; e->count = 0; e->buf_full = 0;
}

static void eqlms_init(eqlms_t *e)
{
	float h[EQ_LEN];
	orc_eq_initial_taps(h);
	for (int i = 0; i < EQ_LEN; i++) { e->h0[i].re = h[i]; e->h0[i].im = 0; }
	e->mu = EQ_STEP;              /* eqlms_cccf_set_bw(0.1), src/hfdl.c:496 */
	eqlms_reset(e);
}

static void eqlms_push(eqlms_t *e, orc_cf x)
{
	float x2n = x.re * x.re + x.im * x.im, x2o = e->x2[0];
	memmove(e->buf, e->buf + 1, sizeof(orc_cf) * (EQ_LEN - 1));
	memmove(e->x2, e->x2 + 1, sizeof(float) * (EQ_LEN - 1));
	e->buf[EQ_LEN - 1] = x;
	e->x2[EQ_LEN - 1] = x2n;
	e->x2_sum = e->x2_sum + x2n - x2o;
	e->count++;
}

static orc_cf eqlms_execute(const eqlms_t *e)
{
	float ar = 0, ai = 0;
	for (int i = 0; i < EQ_LEN; i++) {
		/* conj(w) * x */
		ar += e->w[i].re * e->buf[i].re + e->w[i].im * e->buf[i].im;
		ai += e->w[i].re * e->buf[i].im - e->w[i].im * e->buf[i].re;
	}
	orc_cf y = { ar, ai };
	return y;
}

static void eqlms_step(eqlms_t *e, orc_cf d, orc_cf d_hat)
{
	if (!e->buf_full) {
		if (e->count < EQ_LEN) return;
		e->buf_full = 1;
	}
	/* w += mu * conj(d - d_hat) * x / sum|x|^2 */
	float er = d.re - d_hat.re, ei = -(d.im - d_hat.im);
	float norm = e->x2_sum;
	if (orc_v.eqlms_norm == 1) { norm = 0.f; for (int i = 0; i < EQ_LEN; i++) norm += e->x2[i]; }
	else if (orc_v.eqlms_norm == 2) norm = 1.0f;
	for (int i = 0; i < EQ_LEN; i++) {
		float pr = er * e->buf[i].re - ei * e->buf[i].im;
		float pi = er * e->buf[i].im + ei * e->buf[i].re;
		e->w[i].re = e->w[i].re + e->mu * pr / norm;
		e->w[i].im = e->w[i].im + e->mu * pi / norm;
	}
}

/* ======================= 127-bit sequences (bsequence, a17) ======================= */

typedef struct { uint64_t hi, lo; } bits127;           /* bit 126 = oldest */
#define HI_MASK 0x7FFFFFFFFFFFFFFFull

static inline void bits_push(bits127 *b, uint32_t bit)
{
	b->hi = ((b->hi << 1) | (b->lo >> 63)) & HI_MASK;
	b->lo = (b->lo << 1) | (bit & 1u);
}

static inline int bits_correlate(const bits127 *a, const bits127 *b)
{
	return 127 - __builtin_popcountll((a->hi ^ b->hi) & HI_MASK) - __builtin_popcountll(a->lo ^ b->lo);
}

static bits127 seq_A, seq_M1[ORC_MODE_CNT];
static pthread_once_t seq_once = PTHREAD_ONCE_INIT;

static void seq_init(void)
{
	uint8_t t[127];
	orc_preamble_A(t);
	memset(&seq_A, 0, sizeof(seq_A));
	for (int i = 0; i < 127; i++) bits_push(&seq_A, t[i]);
	for (int m = 0; m < ORC_MODE_CNT; m++) {
		orc_preamble_M1(m, t);
		memset(&seq_M1[m], 0, sizeof(bits127));
		for (int i = 0; i < 127; i++) bits_push(&seq_M1[m], t[i]);
	}
}

/* ======================= the channel (src/hfdl.c:183-230, 468-534) ======================= */

enum { SAMPLER_BITS = 1, SAMPLER_SYMBOLS = 2, SAMPLER_SKIP = 3 };
enum { FR_A1 = 1, FR_A2, FR_M1, FR_M2_SKIP, FR_EQ_TRAIN, FR_DATA_1, FR_DATA_2 };

#define PREKEY_LEN 448
#define A_LEN 127
#define M1_LEN 127
#define M2_LEN 15
#define T_LEN 15
#define DATA_FRAME_LEN 30
#define PREAMBLE_LEN (2 * A_LEN + M1_LEN + M2_LEN + 9 * T_LEN)
#define SINGLE_SLOT_FRAME_LEN (PREKEY_LEN + PREAMBLE_LEN + 72 * (DATA_FRAME_LEN + T_LEN))
#define MAX_DATA_SYMBOLS (168 * DATA_FRAME_LEN)

static const float MF_TAPS[19] = {   /* protocol pulse-shape table, src/hfdl.c:147-154 */
	-0.0170974647427123f, 0.01148231492068473f, 0.03138375667422348f, 0.009454398851680437f,
	-0.04161644170893816f, -0.06451564801420356f, -0.005495792933327306f, 0.1316404671361545f,
	0.2759693160697777f, 0.3375901874933208f, 0.2759693160697777f, 0.1316404671361545f,
	-0.005495792933327306f, -0.06451564801420356f, -0.04161644170893816f, 0.009454398851680437f,
	0.03138375667422348f, 0.01148231492068473f, -0.0170974647427123f
};

struct orc_channel {
	int32_t chan_freq;
	/* channelizer */
	int has_channelizer;
	orc_ddc ddc;
	orc_cf *taps_fft, *scratch, *chan_out;
	orc_nco_state nco;
	/* DSP objects */
	float resamp_rate;
	resamp_t rs;
	agc_t agc;
	orc_cf mf_win[19];          /* [0] newest */
	symsync_t ss;
	struct { float alpha, beta, phi, dphi, err; } loop;     /* src/hfdl.c:236-294 */
	eqlms_t eq;
	/* framer */
	bits127 bits;
	orc_cf training[T_LEN]; int32_t training_n;
	orc_cf data[MAX_DATA_SYMBOLS]; int32_t data_n;
	int use_data_buffer;
	uint64_t symbol_cnt, sample_cnt;
	int s_state, fr_state, data_arity, cur_arity;
	int32_t symbols_wanted, search_retries, eq_train_seq_cnt, data_segment_cnt;
	int32_t train_bits_total, train_bits_bad, T_idx, M1;
	uint32_t bitmask, symsync_out_idx, nf_clk;
	float frame_symbol_cnt;
	uint64_t pdu_sample_index;
	float freq_err_hz, signal_level, noise_floor;
	uint32_t cnt_a2_found, cnt_m1_found, cnt_m1_not_found, cnt_frames;   /* statsd increments, src/hfdl.c:818,828,840 */
	uint32_t cnt_a1_found, cum_train_bad, cum_train_total;               /* S.A1_found, S.train_bits_*, :786, :962-963 */
	float corr_total[3];                                                  /* S.A1_corr_total, S.A2_corr_total, S.M1_corr_total */
	/* stage taps */
	orc_cf *resampled; int32_t resampled_cap, resampled_n;
	orc_cf *mf_out; float *agc_level;
	orc_cf *symbols; int32_t symbols_cap, symbols_n;
	int32_t chan_out_n;
};

static void sampler_reset(orc_channel *c)           /* src/hfdl.c:968-972 */
{
	symsync_reset(&c->ss);
	c->s_state = SAMPLER_BITS;
	c->bitmask = 0;
}

static void framer_reset(orc_channel *c)            /* src/hfdl.c:974-991 */
{
	c->fr_state = FR_A1;
	c->symbols_wanted = 1;
	c->search_retries = 0;
	c->cur_arity = 1;
	c->train_bits_total = c->train_bits_bad = 0;
	c->T_idx = 0;
	c->use_data_buffer = 0;
	eqlms_reset(&c->eq);
	c->data_n = 0;
	c->training_n = 0;
	sampler_reset(c);
}

orc_channel *orc_channel_create(int32_t sample_rate, int32_t decimation, float transition_bw,
		int32_t centerfreq, int32_t frequency, int want_channelizer)
{
	pthread_once(&seq_once, seq_init);
	orc_channel *c = calloc(1, sizeof(*c));
	c->chan_freq = frequency;
	c->resamp_rate = (float)(HFDL_SYMBOL_RATE * HFDL_SPS) / ((float)sample_rate / (float)decimation);
	resamp_init(&c->rs, c->resamp_rate);
	float freq_shift = (float)(centerfreq - (frequency + 1440)) / (float)sample_rate;
	if (orc_fastddc_init(&c->ddc, transition_bw, decimation, freq_shift)) { free(c); return NULL; }
	c->has_channelizer = want_channelizer;
	if (want_channelizer) {
		c->taps_fft = malloc(sizeof(orc_cf) * (size_t)c->ddc.fft_size);
		c->scratch = malloc(sizeof(orc_cf) * 2 * (size_t)c->ddc.fft_inv_size);
		orc_channelizer_taps(&c->ddc, decimation, freq_shift, c->taps_fft, 0);
	}
	c->chan_out = malloc(sizeof(orc_cf) * (size_t)c->ddc.post_input_size);
	c->agc.g = 1.0f; c->agc.y2 = orc_v.agc_y2_init > 0.f ? orc_v.agc_y2_init : 1.0f; c->agc.alpha = AGC_BANDWIDTH;   /* src/hfdl.c:485-487 */
	c->noise_floor = NF_INIT;                                  /* :490 */
	c->loop.alpha = COSTAS_ALPHA;
	c->loop.beta = COSTAS_BETA_K * c->loop.alpha * c->loop.alpha;     /* :240-245 */
	eqlms_init(&c->eq);
	symsync_init(&c->ss, SYMSYNC_LF_BW);                              /* :503-505 */
	framer_reset(c);
	c->resampled_cap = c->ddc.post_input_size + 64;
	c->resampled = malloc(sizeof(orc_cf) * (size_t)c->resampled_cap);
	c->mf_out = malloc(sizeof(orc_cf) * (size_t)c->resampled_cap);
	c->agc_level = malloc(sizeof(float) * (size_t)c->resampled_cap);
	c->symbols_cap = c->resampled_cap;
	c->symbols = malloc(sizeof(orc_cf) * (size_t)c->symbols_cap);
	return c;
}

void orc_channel_destroy(orc_channel *c)
{
	if (!c) return;
	free(c->taps_fft); free(c->scratch); free(c->chan_out);
	free(c->resampled); free(c->mf_out); free(c->agc_level); free(c->symbols);
	free(c);
}

const orc_ddc *orc_channel_ddc(const orc_channel *c) { return &c->ddc; }

void orc_channel_counters(const orc_channel *c, uint32_t out[4], float *noise_floor, int *framer_state)
{
	out[0] = c->cnt_a2_found; out[1] = c->cnt_m1_found; out[2] = c->cnt_m1_not_found; out[3] = c->cnt_frames;
	*noise_floor = c->noise_floor;
	*framer_state = c->fr_state;
}
const orc_cf *orc_channel_taps(const orc_channel *c) { return c->taps_fft; }

void orc_channel_summary(const orc_channel *c, uint32_t out[6], float corr[3])
{
	out[0] = c->cnt_a1_found; out[1] = c->cnt_a2_found; out[2] = c->cnt_m1_found; out[3] = c->cnt_m1_not_found;
	out[4] = c->cum_train_bad; out[5] = c->cum_train_total;
	for (int i = 0; i < 3; i++) corr[i] = c->corr_total[i];
}

void orc_channel_taps_view(const orc_channel *c, orc_taps_view *v)
{
	v->chan_out = c->chan_out; v->chan_out_n = c->chan_out_n;
	v->resampled = c->resampled; v->resampled_n = c->resampled_n;
	v->mf_out = c->mf_out; v->mf_out_n = c->resampled_n;
	v->symbols = c->symbols; v->symbols_n = c->symbols_n;
	v->agc_level = c->agc_level;
}

static void emit_pdu(orc_channel *c, orc_pdu_sink sink, void *ctx)   /* src/hfdl.c:993-1080 */
{
	orc_pdu p;
	memset(&p, 0, sizeof(p));
	p.len = orc_decode_user_data(c->M1, c->data, (int)(c->bitmask & 1), p.octets);
	p.freq = c->chan_freq;
	p.mode = c->M1;
	p.freq_err_hz = c->freq_err_hz;
	p.rssi_db = 20.0f * log10f(c->signal_level);
	p.noise_floor_db = 20.0f * log10f(c->noise_floor);
	const orc_mode_params *m = &orc_modes[c->M1];
	p.bit_rate = HFDL_SYMBOL_RATE * m->arity / m->code_rate * DATA_FRAME_LEN / (DATA_FRAME_LEN + T_LEN);
	p.slot = m->segments == 72 ? 'S' : 'D';
	p.sample_index = c->pdu_sample_index;
	p.train_bits_bad = c->train_bits_bad;
	p.train_bits_total = c->train_bits_total;
	if (sink) sink(ctx, &p);
}

/* compute_train_bit_error_cnt, src/hfdl.c:952-966 */
static void count_train_errors(orc_channel *c)
{
	uint32_t seq = 0;
	for (int i = 0; i < T_LEN; i++) {
		uint32_t bit = (c->training[i].re > 0) ? 0 : 1;
		bit ^= (c->bitmask & 1);
		seq = (seq << 1) | bit;
	}
	int err = __builtin_popcount(T_SEQ_BITS ^ seq);
	c->train_bits_total += T_LEN;
	c->train_bits_bad += err;
	c->cum_train_total += T_LEN;
	c->cum_train_bad += (uint32_t)err;
}

/* everything after the equaliser for one on-time symbol: src/hfdl.c:737-891 */
static void on_symbol(orc_channel *c, orc_cf s, orc_pdu_sink sink, void *ctx)
{
	float perr;
	uint32_t bits = orc_modem_demod_hard(c->cur_arity, s, &perr);
	/* costas_cccf_adjust, :276-281 */
	{
		float e = perr;
		e = 0.5f * (fabsf(e + COSTAS_ERR_LIMIT) - fabsf(e - COSTAS_ERR_LIMIT));
		c->loop.err = e;
		c->loop.phi += c->loop.alpha * e;
		c->loop.dphi += c->loop.beta * e;
	}
	c->symbol_cnt++;
	if (c->symbol_cnt >= (uint64_t)(NO_FRAME_TIMEOUT_FRAMES * SINGLE_SLOT_FRAME_LEN) && c->fr_state == FR_A1) {
		c->symbol_cnt = 0;
		c->loop.dphi = c->loop.phi = 0.0f;
		symsync_reset(&c->ss);
	}
	if (c->s_state == SAMPLER_BITS) {
		bits ^= c->bitmask;
		for (int b = 0; b < c->cur_arity; b++, bits >>= 1) bits_push(&c->bits, bits);
	} else if (c->s_state == SAMPLER_SYMBOLS) {
		if (c->use_data_buffer) { if (c->data_n < MAX_DATA_SYMBOLS) c->data[c->data_n++] = s; }
		else if (c->training_n < T_LEN) c->training[c->training_n++] = s;
	}
	if (c->fr_state > FR_A1) {
		float lvl = 1.0f / c->agc.g;
		c->signal_level = (c->signal_level * c->frame_symbol_cnt + lvl) / (c->frame_symbol_cnt + 1.0f);
		c->frame_symbol_cnt += 1.0f;
	}
	if (c->symbols_wanted > 1) { c->symbols_wanted--; return; }

	switch (c->fr_state) {
	case FR_A1: {
		float corr = 2.0f * (float)bits_correlate(&seq_A, &c->bits) / (float)A_LEN - 1.0f;
		if (fabsf(corr) > CORR_THRESHOLD_A1) {
			c->cnt_a1_found++;
			c->corr_total[0] += fabsf(corr);
			c->bitmask = corr > 0.f ? 0 : ~0u;
			c->signal_level = 1.0f / c->agc.g;
			c->frame_symbol_cnt = 1.0f;
			c->symbols_wanted = A_LEN;
			c->search_retries = 0;
			c->fr_state = FR_A2;
		}
		break; }
	case FR_A2: {
		float corr = 2.0f * (float)bits_correlate(&seq_A, &c->bits) / (float)A_LEN - 1.0f;
		if (fabsf(corr) > CORR_THRESHOLD_A2) {
			c->cnt_a2_found++;
			c->corr_total[1] += fabsf(corr);
			c->pdu_sample_index = c->sample_cnt;   /* reference: wall clock, :808-809 */
			c->freq_err_hz = (float)(c->loop.dphi * HFDL_SYMBOL_RATE / (2.0 * M_PI));
			c->symbols_wanted = M1_LEN;
			c->search_retries = 0;
			c->fr_state = FR_M1;
		} else if (++c->search_retries >= MAX_SEARCH_RETRIES) {
			framer_reset(c);
		}
		break; }
	case FR_M1: {
		float best = 0.f; int best_idx = -1;
		for (int m = 0; m < ORC_MODE_CNT; m++) {
			float corr = fabsf(2.0f * (float)bits_correlate(&seq_M1[m], &c->bits) / 127.0f - 1.0f);
			if (corr > best) { best = corr; best_idx = m; }
		}
		if (fabsf(best) > CORR_THRESHOLD_M1) {
			c->cnt_m1_found++;
			c->corr_total[2] += fabsf(best);
			c->data_segment_cnt = orc_modes[best_idx].segments;
			c->data_arity = orc_modes[best_idx].arity;
			c->M1 = best_idx;
			c->symbols_wanted = M2_LEN;
			c->search_retries = 0;
			c->fr_state = FR_M2_SKIP;
			c->s_state = SAMPLER_SKIP;
		} else {
			c->cnt_m1_not_found++;
			framer_reset(c);
		}
		break; }
	case FR_M2_SKIP:
		c->training_n = 0;
		c->symbols_wanted = T_LEN;
		c->eq_train_seq_cnt = 9;
		c->fr_state = FR_EQ_TRAIN;
		c->s_state = SAMPLER_SYMBOLS;
		break;
	case FR_EQ_TRAIN:
		count_train_errors(c);
		c->training_n = 0;
		if (c->eq_train_seq_cnt > 1) {
			c->eq_train_seq_cnt--;
			c->symbols_wanted = T_LEN;
			c->T_idx = 0;
		} else if (c->data_segment_cnt > 0) {
			c->symbols_wanted = DATA_FRAME_LEN / 2;
			c->fr_state = FR_DATA_1;
			c->cur_arity = c->data_arity;
			c->use_data_buffer = 1;
		} else {
			emit_pdu(c, sink, ctx);
			c->cnt_frames++;
			framer_reset(c);
			c->symbol_cnt = 0;
		}
		break;
	case FR_DATA_1:
		c->symbols_wanted = DATA_FRAME_LEN / 2;
		c->fr_state = FR_DATA_2;
		break;
	case FR_DATA_2:
		c->data_segment_cnt--;
		c->cur_arity = 1;
		c->use_data_buffer = 0;
		c->fr_state = FR_EQ_TRAIN;
		c->eq_train_seq_cnt = 1;
		c->symbols_wanted = T_LEN;
		c->T_idx = 0;
		break;
	}
}

static const float T_BPSK[15] = { 1, 1, 1, -1, 1, 1, -1, -1, 1, -1, 1, -1, -1, -1, -1 };  /* src/hfdl.c:157-160 */

/* the 5400-sps loop body: src/hfdl.c:685-892 */
static void process_resampled(orc_channel *c, orc_pdu_sink sink, void *ctx)
{
	c->symbols_n = 0;
	for (int32_t k = 0; k < c->resampled_n; k++, c->sample_cnt++) {
		orc_cf r = agc_step(&c->agc, c->resampled[k]);
		c->agc_level[k] = 1.0f / c->agc.g;
		memmove(c->mf_win + 1, c->mf_win, sizeof(orc_cf) * 18);
		c->mf_win[0] = r;
		orc_cf s = dot_rc(MF_TAPS, c->mf_win, 19);
		c->mf_out[k] = s;
		if (c->fr_state == FR_A1 && (++c->nf_clk & NF_CLK_MASK) == NF_CLK_MASK) {
			float lvl = 1.0f / c->agc.g;
			c->noise_floor = NF_KEEP * c->noise_floor + NF_TAKE * fminf(c->noise_floor, lvl) + NF_BIAS;
		}
		orc_cf sym[4];
		int32_t produced = symsync_step(&c->ss, s, sym);
		for (int32_t i = 0; i < produced; i++, c->symsync_out_idx++) {
			/* costas step + execute, :284-292, :256-258 */
			c->loop.phi += c->loop.dphi;
			if (c->loop.phi > (float)M_PI) c->loop.phi -= (float)(2.0 * M_PI);
			else if (c->loop.phi < -(float)M_PI) c->loop.phi += (float)(2.0 * M_PI);
			float cp, sp;
			if (orc_v.shared_math) sm_sincosf(c->loop.phi, &sp, &cp);
			else { cp = cosf(c->loop.phi); sp = sinf(c->loop.phi); }
			r.re = sym[i].re * cp + sym[i].im * sp;
			r.im = sym[i].im * cp - sym[i].re * sp;
			if (fabsf(c->loop.dphi) > COSTAS_RUNAWAY_DPHI && c->fr_state == FR_A1) {
				c->loop.dphi = c->loop.phi = 0.f;
				symsync_reset(&c->ss);
			}
			eqlms_push(&c->eq, r);
			if (!(c->symsync_out_idx & 1)) continue;
			s = eqlms_execute(&c->eq);
			if (c->fr_state == FR_EQ_TRAIN) {
				float tv = T_BPSK[c->T_idx] * ((c->bitmask & 1) ? -1.0f : 1.0f);
				orc_cf d = { tv, 0 };
				eqlms_step(&c->eq, d, s);
				c->T_idx++;
			}
			if (c->symbols_n < c->symbols_cap) c->symbols[c->symbols_n++] = s;
			on_symbol(c, s, sink, ctx);
		}
	}
}

void orc_channel_process_baseband(orc_channel *c, const orc_cf *x, int32_t n, orc_pdu_sink sink, void *ctx)
{
	if (n + 64 > c->resampled_cap) {
		c->resampled_cap = n + 64; c->symbols_cap = n + 64;
		c->resampled = realloc(c->resampled, sizeof(orc_cf) * (size_t)c->resampled_cap);
		c->mf_out = realloc(c->mf_out, sizeof(orc_cf) * (size_t)c->resampled_cap);
		c->agc_level = realloc(c->agc_level, sizeof(float) * (size_t)c->resampled_cap);
		c->symbols = realloc(c->symbols, sizeof(orc_cf) * (size_t)c->symbols_cap);
	}
	int32_t total = 0;
	for (int32_t i = 0; i < n; i++) total += resamp_push(&c->rs, x[i], c->resampled + total);
	c->resampled_n = total;
	if (total < 1) return;
	process_resampled(c, sink, ctx);
}

void orc_channel_process_spectrum(orc_channel *c, const orc_cf *spectrum, orc_pdu_sink sink, void *ctx)
{
	c->nco = orc_fastddc_inv(spectrum, c->chan_out, &c->ddc, c->taps_fft, c->nco, c->scratch);
	c->chan_out_n = c->nco.output_size;
	orc_channel_process_baseband(c, c->chan_out, c->chan_out_n, sink, ctx);
}

/* ======================= whole front end (src/main.c:699-774) ======================= */

struct orc_frontend {
	int32_t nch;
	orc_ddc ddc;                 /* shift = 0 geometry, src/fft.c:70-86 */
	orc_cf *buf, *spectrum;
	orc_channel **ch;
};

struct creator { orc_frontend *f; int32_t sample_rate, centerfreq, decim; float tbw; const int32_t *freqs; int first, step; };

static void *creator_main(void *arg)
{
	struct creator *w = arg;
	for (int32_t i = w->first; i < w->f->nch; i += w->step)
		w->f->ch[i] = orc_channel_create(w->sample_rate, w->decim, w->tbw, w->centerfreq, w->freqs[i], 1);
	return NULL;
}

orc_frontend *orc_frontend_create_mt(int32_t sample_rate, int32_t centerfreq, const int32_t *freqs, int32_t nch, int nthreads)
{
	orc_frontend *f = calloc(1, sizeof(*f));
	int32_t decim = orc_compute_fft_decimation_rate(sample_rate, HFDL_SYMBOL_RATE * HFDL_SPS);
	float tbw = orc_transition_bw(sample_rate, 250);
	if (orc_fastddc_init(&f->ddc, tbw, decim, 0)) { free(f); return NULL; }
	f->buf = calloc((size_t)f->ddc.fft_size, sizeof(orc_cf));
	f->spectrum = calloc((size_t)f->ddc.fft_size, sizeof(orc_cf));
	f->nch = nch;
	f->ch = calloc((size_t)nch, sizeof(*f->ch));
	if (nthreads < 1) nthreads = 1;
	if (nthreads > nch) nthreads = nch;
	struct creator *w = calloc((size_t)nthreads, sizeof(*w));
	pthread_t *th = calloc((size_t)nthreads, sizeof(*th));
	for (int t = 0; t < nthreads; t++) {
		w[t] = (struct creator){ f, sample_rate, centerfreq, decim, tbw, freqs, t, nthreads };
		if (nthreads == 1) creator_main(&w[t]); else pthread_create(&th[t], NULL, creator_main, &w[t]);
	}
	for (int t = 0; t < nthreads && nthreads > 1; t++) pthread_join(th[t], NULL);
	free(w); free(th);
	return f;
}

orc_frontend *orc_frontend_create(int32_t sample_rate, int32_t centerfreq, const int32_t *freqs, int32_t nch)
{
	return orc_frontend_create_mt(sample_rate, centerfreq, freqs, nch, 1);
}

void orc_frontend_destroy(orc_frontend *f)
{
	if (!f) return;
	for (int32_t i = 0; i < f->nch; i++) orc_channel_destroy(f->ch[i]);
	free(f->ch); free(f->buf); free(f->spectrum); free(f);
}

const orc_ddc *orc_frontend_ddc(const orc_frontend *f) { return &f->ddc; }
const orc_cf *orc_frontend_spectrum(const orc_frontend *f) { return f->spectrum; }
orc_channel *orc_frontend_channel(orc_frontend *f, int32_t i) { return f->ch[i]; }

struct worker { orc_frontend *f; int first, step; orc_pdu *out; int32_t n_out, cap; };

static void collect(void *ctx, const orc_pdu *p)
{
	struct worker *w = ctx;
	if (w->n_out == w->cap) { w->cap = w->cap ? 2 * w->cap : 4; w->out = realloc(w->out, sizeof(orc_pdu) * (size_t)w->cap); }
	w->out[w->n_out++] = *p;
}

static void *worker_main(void *arg)
{
	struct worker *w = arg;
	for (int32_t i = w->first; i < w->f->nch; i += w->step)
		orc_channel_process_spectrum(w->f->ch[i], w->f->spectrum, collect, w);
	return NULL;
}

void orc_frontend_push_block(orc_frontend *f, const orc_cf *samples, int nthreads, orc_pdu_sink sink, void *ctx)
{
	orc_forward_block(f->buf, samples, &f->ddc, f->spectrum);
	if (nthreads < 1) nthreads = 1;
	if (nthreads > f->nch) nthreads = f->nch;
	struct worker *w = calloc((size_t)nthreads, sizeof(*w));
	pthread_t *th = calloc((size_t)nthreads, sizeof(*th));
	for (int t = 0; t < nthreads; t++) {
		w[t].f = f; w[t].first = t; w[t].step = nthreads;
		if (nthreads == 1) worker_main(&w[t]); else pthread_create(&th[t], NULL, worker_main, &w[t]);
	}
	for (int t = 0; t < nthreads; t++) {
		if (nthreads > 1) pthread_join(th[t], NULL);
		for (int32_t i = 0; i < w[t].n_out; i++) if (sink) sink(ctx, &w[t].out[i]);
		free(w[t].out);
	}
	free(w); free(th);
}

/* the constants and static tables of this restatement, for the comparison with the reference's text (tests/test_constants_cpu.py):
 * ints[0..16], floats[0..12], mf[19], t_seq[15]; the orders are those of the test */
void orc_constants(int32_t *ints, float *floats, float *mf, float *t_seq)
{
	const int32_t iv[] = { PREKEY_LEN, A_LEN, M1_LEN, M2_LEN, T_LEN, DATA_FRAME_LEN, PREAMBLE_LEN, SINGLE_SLOT_FRAME_LEN, MAX_DATA_SYMBOLS,
		MAX_SEARCH_RETRIES, NO_FRAME_TIMEOUT_FRAMES, HFDL_SYMBOL_RATE, HFDL_SPS, EQ_LEN, 19, SS_NPFB, (int32_t)NF_CLK_MASK,
		SAMPLER_BITS, SAMPLER_SYMBOLS, SAMPLER_SKIP, FR_A1, FR_A2, FR_M1, FR_M2_SKIP, FR_EQ_TRAIN, FR_DATA_1, FR_DATA_2, SS_K, SS_KOUT };
	const float fv[] = { CORR_THRESHOLD_A1, CORR_THRESHOLD_A2, CORR_THRESHOLD_M1, COSTAS_ALPHA, COSTAS_BETA_K * COSTAS_ALPHA * COSTAS_ALPHA, COSTAS_ERR_LIMIT,
		COSTAS_RUNAWAY_DPHI, AGC_BANDWIDTH, EQ_STEP, NF_KEEP, NF_TAKE, NF_BIAS, NF_INIT, SYMSYNC_LF_BW };
	for (size_t i = 0; i < sizeof(iv) / sizeof(iv[0]); i++) ints[i] = iv[i];
	for (size_t i = 0; i < sizeof(fv) / sizeof(fv[0]); i++) floats[i] = fv[i];
	for (int i = 0; i < 19; i++) mf[i] = MF_TAPS[i];
	for (int i = 0; i < 15; i++) t_seq[i] = T_BPSK[i];
}
