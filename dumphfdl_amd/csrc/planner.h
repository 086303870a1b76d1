// planner.h -- host-side (C++) block-geometry planner and filter design of the GPU front end.
//
// Product code (not the oracle): mirrors, expression by expression, the float/double mixing of the reference's
// init-time arithmetic, because bin shifts and tap phases depend on it:
//   fastddc_init                      src/fastddc.c:46-80
//   decimating_shift_addition_init    src/libcsdr_gpl.c:26-39
//   firdes_lowpass_f / firdes_bandpass_c / Hamming kernel   src/libcsdr.c:62-68,83-133
//   compute_fft_decimation_rate, compute_filter_relative_transition_bw   src/libcsdr.c:135-144
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include <complex>

namespace hfdl {

struct Plan {
	int32_t pre, post, taps_min_length, taps_length, overlap, n, m, input_size, post_input_size, scrap, v;
	int32_t startbin, offsetbin;
	float pre_shift, post_shift, sindelta, cosdelta, rate;
};

inline int32_t pow2_above(int32_t x)
{
	int32_t p = 1;
	for (int i = 0; i < 31; i++, p <<= 1) if (x < p) return p;
	return -1;
}

inline int32_t fft_decimation_rate(int32_t fs, int32_t target)
{
	return pow2_above((int32_t)std::floor((float)fs / (float)target)) / 2;
}

inline float relative_transition_bw(int32_t fs, int32_t hz) { return (float)hz / (float)fs; }

inline bool plan_block(Plan &p, float transition_bw, int32_t decimation, float shift_rate)
{
	p.pre = 1; p.post = decimation;
	while (true) {
		float h = (float)p.post / 2;
		if (std::floor(h) != h || p.post / 2 == 1) break;
		p.post /= 2; p.pre *= 2;
	}
	int32_t tl = (int32_t)(4.0 / transition_bw);
	p.taps_min_length = (tl % 2 == 0) ? tl + 1 : tl;
	p.taps_length = pow2_above((int32_t)(std::ceil((double)(p.taps_min_length / (float)p.pre)) * p.pre)) + 1;
	p.n = pow2_above(p.taps_length * 4);
	while (p.n < p.pre) p.n *= 2;
	p.overlap = p.taps_length - 1;
	p.input_size = p.n - p.overlap;
	p.m = p.n / p.pre;
	p.v = p.n / p.overlap;
	const int32_t mid = p.n / 2;
	float sb = (float)mid + (float)mid * (-shift_rate) * 2;
	p.startbin = (int32_t)sb;
	p.startbin = (int32_t)(p.v * std::round((double)(p.startbin / (float)p.v)));
	p.offsetbin = p.startbin - mid;
	p.post_shift = p.pre * (shift_rate + ((float)p.offsetbin / p.n));
	p.pre_shift = p.offsetbin / (float)p.n;
	float r = p.post_shift * p.post;
	r *= 2;
	p.sindelta = (float)std::sin(r * M_PI);
	p.cosdelta = (float)std::cos(r * M_PI);
	p.rate = r;
	p.scrap = p.overlap / p.pre;
	p.post_input_size = p.m - p.scrap;
	return p.n > 2;
}

inline float hamming_w(float rate)
{
	rate = (float)(0.5 + rate / 2);
	return (float)(0.54 - 0.46 * std::cos(2 * M_PI * rate));
}

// windowed-sinc low-pass, sum-normalised (fp32 running sum as in the reference)
inline void design_lowpass(std::vector<float> &h, int32_t length, float cutoff)
{
	h.assign((size_t)length, 0.f);
	const int32_t mid = length / 2;
	h[mid] = (float)(2 * M_PI * cutoff * hamming_w(0));
	for (int32_t i = 1; i <= mid; i++) {
		float t = (float)((std::sin(2 * M_PI * cutoff * i) / i) * hamming_w((float)i / mid));
		h[mid - i] = t; h[mid + i] = t;
	}
	float sum = 0;
	for (int32_t i = 0; i < length; i++) sum += h[i];
	for (int32_t i = 0; i < length; i++) h[i] = h[i] / sum;
}

// complex band-pass taps of one channel: low-pass x e^{j theta_n}, theta accumulated in fp32 and wrapped to [0, 2 pi]
inline void design_bandpass(std::complex<float> *out, int32_t length, float lowcut, float highcut, std::vector<float> &lp_cache,
		float &lp_cache_cutoff)
{
	const float cutoff = (highcut - lowcut) / 2;
	if (lp_cache.size() != (size_t)length || lp_cache_cutoff != cutoff) {
		design_lowpass(lp_cache, length, cutoff);
		lp_cache_cutoff = cutoff;
	}
	const float center = (highcut + lowcut) / 2;
	float phase = 0;
	for (int32_t i = 0; i < length; i++) {
		float c = (float)std::cos((double)phase), s = (float)std::sin((double)phase);
		phase = (float)(phase + 2 * M_PI * center);
		while (phase > 2 * M_PI) phase = (float)(phase - 2 * M_PI);
		while (phase < 0) phase = (float)(phase + 2 * M_PI);
		out[i] = std::complex<float>(c * lp_cache[i], s * lp_cache[i]);
	}
}

}  // namespace hfdl
