// demod_tables.h -- constant tables of the per-channel HFDL demodulator, designed on the host at create time.
//
// Product code.  The filter designs restate the published algorithms of the liquid-dsp objects dumphfdl
// constructs (liquid-dsp >=1.3.0,<2.0.0 is an external dependency of the reference, src/CMakeLists.txt:71-73):
//   msresamp_crcf_create(rate, 60 dB)                     src/hfdl.c:472-473
//   firfilt_crcf_create(hfdl_matched_filter, 19)          src/hfdl.c:147-154,494
//   symsync_crcf_create_kaiser(3, 3, 0.9, 16), lf_bw .001 src/hfdl.c:503-505
//   eqlms_cccf_create_lowpass(15, 0.45), bw 0.1           src/hfdl.c:495-496
//   bsequence A / M1[8], descrambler LFSR                 src/hfdl.c:300-347,419-459
#pragma once
#include "demod_logic.h"       // the named constants of src/hfdl.c (thresholds, lengths); every user includes it first anyway
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace hfdl {

constexpr int RS_NPFB = 256, RS_TAPS = 14;
constexpr int SS_NPFB = 16, SS_TAPS = 18;
constexpr int EQ_TAPS = 15, MF_TAPS = 19;

constexpr int DEC_PSK_OFFSET = 128, DEC_CONST_BYTES = 256;      // burst decoder constants: scrambler bits, then psk_pts

struct DemodTables {
	float rs_h[RS_NPFB * RS_TAPS];
	uint32_t rs_step;
	float mf[MF_TAPS];
	float ss_mf[SS_NPFB * SS_TAPS], ss_dmf[SS_NPFB * SS_TAPS];
	float lf_b0, lf_a1, ss_rate_adj;
	float eq_h0[EQ_TAPS];
	uint64_t a_hi, a_lo, m1_hi[8], m1_lo[8];
	// scrambler and constellations are adjacent on purpose: the burst decoder takes ONE pointer (scrambler) and finds the
	// constellation table DEC_PSK_OFFSET bytes after it
	uint8_t scrambler[120], scr_pad[8];
	// the PSK constellations as modem_modulate_psk makes them, cexpjf(s * 2 * pi / M) for the LINEAR (Gray-decoded) index s, entry
	// (1 << arity) - 2 + s = {re, im}: arity 1 at [0..1], 2 at [2..5], 3 at [6..13].  The slicer re-modulates its decision once
	// per symbol on the carrier loop's critical path; two v_readlane instead of a sinf + a cosf.
	float psk_pts[16][2];
	float corr_tab[128];
	// the preamble thresholds of src/hfdl.c:42-44 as match counts: |corr_tab[m]| > 0.36 <=> m <= a1_lo or m >= a1_hi (the
	// table is monotonic in m), same for 0.30 (A2); corr_tab[m] > 0 <=> m >= pos_min.  Derived FROM the fp32 table, so the
	// integer tests decide exactly as the reference's float comparisons do.
	int32_t a1_lo, a1_hi, a2_lo, a2_hi, pos_min, thr_pad;
};

static_assert(offsetof(DemodTables, psk_pts) - offsetof(DemodTables, scrambler) == DEC_PSK_OFFSET, "decoder constants layout");

namespace tables_detail {

inline double bessel_i0(double z)
{
	double t = 1.0, s = 1.0;
	for (int k = 1; k < 64; k++) { t *= (0.5 * z) / k; s += t * t; if (t * t < 1e-18 * s) break; }
	return s;
}

inline double kaiser_beta(double as)
{
	as = std::fabs(as);
	if (as > 50.0) return 0.1102 * (as - 8.7);
	if (as > 21.0) return 0.5842 * std::pow(as - 21.0, 0.4) + 0.07886 * (as - 21.0);
	return 0.0;
}

// liquid_firdes_kaiser(n, fc, As, mu = 0)
inline void kaiser_lowpass(int n, double fc, double as, float *h)
{
	const double beta = kaiser_beta(as), denom = bessel_i0(beta);
	for (int i = 0; i < n; i++) {
		double t = i - (n - 1) / 2.0, r = 2.0 * t / n;
		double x = 2.0 * fc * t;
		double sinc = std::fabs(x) < 1e-9 ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
		h[i] = (float)(sinc * bessel_i0(beta * std::sqrt(1.0 - r * r)) / denom);
	}
}

struct Bits127 { uint64_t hi = 0, lo = 0; void push(unsigned b) { hi = ((hi << 1) | (lo >> 63)) & 0x7FFFFFFFFFFFFFFFull; lo = (lo << 1) | (b & 1u); } };

}  // namespace tables_detail

inline void build_demod_tables(DemodTables &t, float resamp_rate)
{
	using namespace tables_detail;
	std::memset(&t, 0, sizeof(t));
	// --- arbitrary resampler: m = 7, fc = min(0.515 r, 0.49), As = 60, 256 branches, unity DC gain per branch
	{
		const int n = 2 * 7 * RS_NPFB + 1;
		std::vector<float> hf((size_t)n);
		double fc = 0.515 * resamp_rate;
		if (fc > 0.49) fc = 0.49;
		kaiser_lowpass(n, (float)fc / (float)RS_NPFB, 60.0, hf.data());
		float gain = 0.f;
		for (int i = 0; i < n; i++) gain += hf[i];
		gain = (float)RS_NPFB / gain;
		for (int b = 0; b < RS_NPFB; b++)
			for (int k = 0; k < RS_TAPS; k++) t.rs_h[b * RS_TAPS + k] = hf[b + k * RS_NPFB] * gain;
		t.rs_step = (uint32_t)std::lround((double)(1u << 24) / (double)resamp_rate);
	}
	// --- matched filter: the protocol's pulse table
	static const float mf[MF_TAPS] = {
		-0.0170974647427123f, 0.01148231492068473f, 0.03138375667422348f, 0.009454398851680437f,
		-0.04161644170893816f, -0.06451564801420356f, -0.005495792933327306f, 0.1316404671361545f,
		0.2759693160697777f, 0.3375901874933208f, 0.2759693160697777f, 0.1316404671361545f,
		-0.005495792933327306f, -0.06451564801420356f, -0.04161644170893816f, 0.009454398851680437f,
		0.03138375667422348f, 0.01148231492068473f, -0.0170974647427123f };
	std::memcpy(t.mf, mf, sizeof(mf));
	// --- symbol synchroniser filter banks
	{
		constexpr int HL = 2 * SS_NPFB * 3 * 3 + 1;
		float hf[HL], H[HL], dH[HL];
		const float fc = 0.75f;
		kaiser_lowpass(HL, fc / (float)(3 * SS_NPFB), 40.0, hf);
		for (int i = 0; i < HL; i++) H[i] = hf[i] * 2.0f * fc;
		float peak = 0.f;
		for (int i = 0; i < HL; i++) {
			dH[i] = (i == 0) ? H[1] - H[HL - 1] : (i == HL - 1) ? H[0] - H[i - 1] : H[i + 1] - H[i - 1];
			float v = std::fabs(H[i] * dH[i]);
			if (v > peak || i == 0) peak = v;
		}
		for (int i = 0; i < HL; i++) dH[i] *= 0.06f / peak;
		for (int b = 0; b < SS_NPFB; b++)
			for (int k = 0; k < SS_TAPS; k++) {
				t.ss_mf[b * SS_TAPS + k] = H[b + k * SS_NPFB];
				t.ss_dmf[b * SS_TAPS + k] = dH[b + k * SS_NPFB];
			}
		const float bw = 0.001f;
		const float alpha = 1.000f - bw, beta = 0.220f * bw, a = 0.500f, b = 0.495f;
		const float A0 = 1.00f - a * alpha, A1 = -b * alpha;
		t.lf_b0 = beta / A0;
		t.lf_a1 = A1 / A0;
		t.ss_rate_adj = 0.5f * bw;
	}
	// --- equaliser start taps
	{
		float h[EQ_TAPS];
		kaiser_lowpass(EQ_TAPS, 0.45, 40.0, h);
		for (int i = 0; i < EQ_TAPS; i++) t.eq_h0[i] = h[i] * 2.0f * 0.45f;
	}
	// --- preamble sequences (oldest bit in bit 126)
	{
		static const uint8_t a_oct[16] = { 0x5B, 0xBC, 0x74, 0x57, 0x03, 0xD9, 0x89, 0x39, 0xF2, 0x08, 0xD5, 0x36, 0x94, 0x2C, 0x32, 0xFE };
		static const char *m1 = "01110110111101000101100" "10111110001000000110011011" "00011100111010111000010011"
				"00000101010110100100101001" "11100100011010100001111111";
		static const int shifts[8] = { 72, 82, 113, 123, 61, 103, 93, 9 };
		Bits127 a;
		for (int i = 0; i < 127; i++) a.push((a_oct[i / 8] >> (7 - i % 8)) & 1);
		t.a_hi = a.hi; t.a_lo = a.lo;
		for (int m = 0; m < 8; m++) {
			Bits127 s;
			for (int j = 0; j < 127; j++) s.push((unsigned)(m1[(shifts[m] + j) % 127] - '0'));
			t.m1_hi[m] = s.hi; t.m1_lo[m] = s.lo;
		}
	}
	// --- preamble correlation value for m matching bits out of 127, in the reference's fp32 expression (src/hfdl.c:781)
	for (int m = 0; m < 128; m++) {
		volatile float v = 2.0f * (float)m;
		v = v / (float)127;
		v = v - 1.0f;
		t.corr_tab[m] = v;
	}
	t.a1_lo = t.a2_lo = -1; t.a1_hi = t.a2_hi = t.pos_min = 128; t.thr_pad = 0;
	for (int m = 0; m < 128; m++) {
		const float c = t.corr_tab[m];
		if (std::fabs(c) > CORR_THRESHOLD_A1) { if (c < 0.f) t.a1_lo = m; else if (m < t.a1_hi) t.a1_hi = m; }
		if (std::fabs(c) > CORR_THRESHOLD_A2) { if (c < 0.f) t.a2_lo = m; else if (m < t.a2_hi) t.a2_hi = m; }
		if (c > 0.f && m < t.pos_min) t.pos_min = m;
	}
	// --- PSK constellations, in the modem's own fp32 expressions (alpha = pi / M as float, angle = s * 2 * alpha)
	t.psk_pts[0][0] = 1.0f; t.psk_pts[1][0] = -1.0f;
	for (int arity = 2; arity <= 3; arity++) {
		const uint32_t M = 1u << arity;
		const float alpha = (float)M_PI / (float)M;
		for (uint32_t k = 0; k < M; k++) {
			const float ang = (float)k * 2 * alpha;
			t.psk_pts[M - 2 + k][0] = cosf(ang);
			t.psk_pts[M - 2 + k][1] = sinf(ang);
		}
	}
	// --- descrambler: x^15 + x + 1 LFSR, fill 0x4d4b, 120-symbol period
	{
		uint32_t v = 0x4d4b;
		for (int i = 0; i < 120; i++) {
			uint32_t b = (uint32_t)__builtin_parity(v & 0x4001);
			v = ((v << 1) | b) & 0x7fff;
			t.scrambler[i] = (uint8_t)b;
		}
	}
}

}  // namespace hfdl
