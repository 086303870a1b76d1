// serial_demod.h -- TEST HARNESS ONLY: a one-lane, plain-C++ form of the demodulator's block loop, written against the
// product's own state, tables, modem and framer FSM (dumphfdl_amd/csrc/demod_logic.h, compiled for the host behind the
// shims in hostsim.cpp).  The shipped block loop (dumphfdl_amd/csrc/demod_core.h) is device code -- register-resident
// windows, DPP reductions, hardware transcendentals -- and is gated on the GPU; this serial form exists so that the
// framer / sampler / slicer logic the device calls can be checked bit for bit against the oracle on a machine without a GPU.
#pragma once
#include "../../dumphfdl_amd/csrc/demod_logic.h"

// The second user (round 4): the device's test-only build -DHFDL_DM_STRICT runs THIS loop, one lane per channel, in place of the
// three-wave pipeline (dumphfdl_amd/csrc/demod_kernels.hip), with the elementary functions of tests/hostsim/shared_math.h on both
// sides (the oracle through orc_variant.shared_math): device and oracle then run the same chain of fp32 operations and must agree
// bit for bit.  HFDL_DM_STRICT_FAST re-enables, one bit at a time, the forms in which the shipped pipeline differs from this loop --
// each emulated here in the exact operation order of demod_core.h -- so that the frames that differ from the oracle at low SNR can be
// charged to the form that causes them (profiles/strict_study.py); with all bits set this loop reproduces the shipped kernel.
//   1  SUM     the timing loop's and the equaliser's dot products summed as the DPP row scan sums them (a balanced pairwise tree
//              over the 16 lanes of a row; taps t and t + 16 of the 18-tap windows share lane t) instead of tap after tap
//   2  AGC     gain update exp2(-alpha/2 log2 y2) on the hardware v_log_f32 / v_exp_f32, level 1/g by v_rcp_f32
//   4  TRIG    carrier NCO on the hardware v_sin_f32 / v_cos_f32 (argument in revolutions)
//   8  SLICER  the carrier loop's nearest-point slicer (largest Re x conj p) instead of arg() + reference ladder
#ifndef HFDL_DM_STRICT_FAST
#define HFDL_DM_STRICT_FAST 0
#endif
#ifdef HFDL_DM_STRICT
#define SD_EXPF sm_expf
#define SD_LOGF sm_logf
#define SD_SINCOS(x, s, c) sm_sincosf((x), (s), (c))
#else
#define SD_EXPF expf
#define SD_LOGF logf
#define SD_SINCOS(x, s, c) (*(c) = cosf(x), *(s) = sinf(x))
#endif

namespace hfdl {

// row_scan_sum's order of additions (demod_core.h): lane 15 of a row ends up with ((v15+v14)+(v13+v12)) + ... -- at every level the
// higher-lane partial sum is the left operand
HFDL_FN float tree16(const float *v)
{
	float t[16];
	for (int i = 0; i < 16; i++) t[i] = v[i];
	for (int n = 16; n > 1; n >>= 1)
		for (int j = 0; j < n / 2; j++) t[j] = t[2 * j + 1] + t[2 * j];
	return t[0];
}

// sum_t h[t] * win[(head - t) mod 18] : polyphase branch output, newest sample first
HFDL_FN cf bank_dot(const float *h, const cf *win, int head)
{
	if (HFDL_DM_STRICT_FAST & 1) {
		// the device: lane t holds h[t] * w[t] + h[t + 16] * w[t + 16] (the second product only for t < 2; elsewhere the tap is 0), rows
		// reduced by the scan
		float vr[16], vi[16];
		for (int t = 0; t < 16; t++) {
			int i0 = head - t; if (i0 < 0) i0 += D_SS_TAPS;
			int i1 = head - t - 16; while (i1 < 0) i1 += D_SS_TAPS;
			const float h1 = t + 16 < D_SS_TAPS ? h[t + 16] : 0.f;
			vr[t] = h[t] * win[i0].x + h1 * win[i1].x;
			vi[t] = h[t] * win[i0].y + h1 * win[i1].y;
		}
		cf y; y.x = tree16(vr); y.y = tree16(vi);
		return y;
	}
	float ar = 0, ai = 0;
	int idx = head;
	for (int t = 0; t < D_SS_TAPS; t++) {
		ar += h[t] * win[idx].x;
		ai += h[t] * win[idx].y;
		idx = idx == 0 ? D_SS_TAPS - 1 : idx - 1;
	}
	cf y; y.x = ar; y.y = ai;
	return y;
}

// the carrier wave's slicer (demod_core.h LaneSlicer) as a loop over the table: the point with the largest Re(x conj p), the lowest
// index among equals, the first point if none compares equal
struct NearestSlicer {
	const float *p;
	HFDL_FN uint32_t operator()(int arity, cf x, float *phase_error) const
	{
		uint32_t sym;
		cf xh;
		if (arity == 1) {
			sym = (x.x > 0) ? 0 : 1;
			xh.x = sym ? -1.0f : 1.0f; xh.y = 0.0f;
		} else {
			const int M = 1 << arity, base = M - 2;
			float best = -1.0f;
			for (int i = 0; i < M; i++) {
				const float d = x.x * p[2 * (base + i)] + x.y * p[2 * (base + i) + 1];
				best = d > best ? d : best;
			}
			int win = base;
			for (int i = M - 1; i >= 0; i--) {
				const float d = x.x * p[2 * (base + i)] + x.y * p[2 * (base + i) + 1];
				if (d == best) win = base + i;
			}
			const uint32_t lin = (uint32_t)(win - base);
			sym = lin ^ (lin >> 1);
			xh.x = p[2 * win]; xh.y = p[2 * win + 1];
		}
		if (phase_error) *phase_error = x.y * xh.x - x.x * xh.y;
		return sym;
	}
};

// returns the number of 5400-sps samples produced
HFDL_FN int demod_block_serial(ChanScalars &s, ChanArrays &a, const DemodConst &T, const BlockIo &io, const cf *in, int n_in)
{
	// ---- R: arbitrary resampler, 24-bit fixed-point phase (msresamp_crcf_execute, src/hfdl.c:676)
	const uint64_t total = (uint64_t)n_in << 24;
	int n_out = 0;
	if ((uint64_t)s.rs_phase < total) n_out = (int)((total - s.rs_phase + T.rs_step - 1) / T.rs_step);
	if (n_out > io.cap - 4) n_out = io.cap - 4;
	for (int k = 0; k < n_out; k++) {
		const uint64_t t = (uint64_t)s.rs_phase + (uint64_t)k * T.rs_step;
		const int i = (int)(t >> 24);
		const float *h = T.rs_h + ((t & 0xFFFFFFu) >> 16) * D_RS_TAPS;
		float ar = 0, ai = 0;
		for (int j = 0; j < D_RS_TAPS; j++) {
			const int idx = i - j;
			const cf x = idx >= 0 ? in[idx] : a.rs_hist[-idx - 1];
			ar += h[j] * x.x;
			ai += h[j] * x.y;
		}
		io.rs[k].x = ar; io.rs[k].y = ai;
	}
	{
		cf tmp[D_RS_TAPS - 1];
		for (int q = 0; q < D_RS_TAPS - 1; q++) tmp[q] = (n_in - 1 - q >= 0) ? in[n_in - 1 - q] : a.rs_hist[q - n_in];
		for (int q = 0; q < D_RS_TAPS - 1; q++) a.rs_hist[q] = tmp[q];
	}
	s.rs_phase = (uint32_t)((uint64_t)s.rs_phase + (uint64_t)n_out * T.rs_step - total);
	if (io.tap_counts) io.tap_counts[0] = n_out;
	if (n_out < 1) return 0;

	// ---- A: AGC (agc_crcf_execute, src/hfdl.c:686)
	{
		float g = s.agc_g, y2 = s.agc_y2;
		const float alpha = AGC_BANDWIDTH;
		for (int k = 0; k < n_out; k++) {
			const cf x = io.rs[k];
			cf y; y.x = x.x * g; y.y = x.y * g;
			const float e = y.x * y.x + y.y * y.y;
			y2 = (1.0f - alpha) * y2 + alpha * e;
#if (HFDL_DM_STRICT_FAST & 2) && defined(__HIP_DEVICE_COMPILE__)
			if (y2 > 1e-6f) g *= __builtin_amdgcn_exp2f(-0.5f * alpha * __builtin_amdgcn_logf(y2));
			if (g > 1e6f) g = 1e6f;
			io.agc[k] = y;
			io.lvl[k] = __builtin_amdgcn_rcpf(g);
#else
			if (y2 > 1e-6f) g *= SD_EXPF(-0.5f * alpha * SD_LOGF(y2));
			if (g > 1e6f) g = 1e6f;
			io.agc[k] = y;
			io.lvl[k] = 1.0f / g;
#endif
		}
		s.agc_g = g; s.agc_y2 = y2;
	}

	// ---- M: 19-tap matched filter (firfilt_crcf, src/hfdl.c:694-695)
	for (int k = 0; k < n_out; k++) {
		float ar = 0, ai = 0;
		for (int t = 0; t < D_MF; t++) {
			const int idx = k - t;
			const cf x = idx >= 0 ? io.agc[idx] : a.mf_hist[-idx - 1];
			ar += T.mf[t] * x.x;
			ai += T.mf[t] * x.y;
		}
		io.mf[k].x = ar; io.mf[k].y = ai;
	}
	{
		cf tmp[D_MF - 1];
		for (int q = 0; q < D_MF - 1; q++) tmp[q] = (n_out - 1 - q >= 0) ? io.agc[n_out - 1 - q] : a.mf_hist[q - n_out];
		for (int q = 0; q < D_MF - 1; q++) a.mf_hist[q] = tmp[q];
	}
	if (io.tap_resampled) {
		for (int k = 0; k < n_out; k++) {
			io.tap_resampled[k] = io.rs[k];
			io.tap_mf[k] = io.mf[k];
			io.tap_level[k] = io.lvl[k];
		}
	}

	// ---- S: timing recovery, carrier loop, equaliser, slicer, framer (src/hfdl.c:696-891)
	int nsym = 0;
	for (int k = 0; k < n_out; k++, s.sample_cnt++) {
		const cf mfo = io.mf[k];
		const float level = io.lvl[k];
		if (s.fr_state == FR_A1 && (++s.nf_clk & NF_CLK_MASK) == NF_CLK_MASK)
			s.noise_floor = NF_KEEP * s.noise_floor + NF_TAKE * fminf(s.noise_floor, level) + NF_BIAS;

		// symsync_crcf_execute, one input sample
		s.ss_head = s.ss_head + 1 == D_SS_TAPS ? 0 : s.ss_head + 1;
		a.ss_mf[s.ss_head] = mfo;
		a.ss_dmf[s.ss_head] = mfo;
		cf out[4];
		int produced = 0;
		while (s.ss_b < D_SS_NPFB && produced < 4) {
			const cf m = bank_dot(T.ss_mf + s.ss_b * D_SS_TAPS, a.ss_mf, s.ss_head);
			out[produced].x = m.x / 3.0f;
			out[produced].y = m.y / 3.0f;
			if (s.ss_decim == 2) {
				s.ss_decim = 0;
				const cf d = bank_dot(T.ss_dmf + s.ss_b * D_SS_TAPS, a.ss_dmf, s.ss_head);
				float q = m.x * d.x + m.y * d.y;
				q = q > 1.0f ? 1.0f : (q < -1.0f ? -1.0f : q);
				s.ss_q = q;
				const float v0 = q - T.lf_a1 * s.ss_v1;
				s.ss_qhat = T.lf_b0 * v0;
				s.ss_v1 = v0;
				s.ss_rate += T.ss_rate_adj * s.ss_qhat;
				s.ss_del = s.ss_rate + s.ss_qhat;
			}
			s.ss_decim++;
			s.ss_tau += s.ss_del;
			s.ss_bf = s.ss_tau * (float)D_SS_NPFB;
			s.ss_b = (int)roundf(s.ss_bf);
			produced++;
		}
		s.ss_tau -= 1.0f;
		s.ss_bf -= (float)D_SS_NPFB;
		s.ss_b -= D_SS_NPFB;

		for (int i = 0; i < produced; i++, s.symsync_out_idx++) {
			// costas_cccf_step + execute, :256-258, :284-292
			s.phi += s.dphi;
			if (s.phi > (float)M_PI) s.phi -= (float)(2.0 * M_PI);
			else if (s.phi < -(float)M_PI) s.phi += (float)(2.0 * M_PI);
			float cp, sp;
#if (HFDL_DM_STRICT_FAST & 4) && defined(__HIP_DEVICE_COMPILE__)
			{ const float rev = s.phi * 0.15915494309189535f; sp = __builtin_amdgcn_sinf(rev); cp = __builtin_amdgcn_cosf(rev); }
#else
			SD_SINCOS(s.phi, &sp, &cp);
#endif
			cf r;
			r.x = out[i].x * cp + out[i].y * sp;
			r.y = out[i].y * cp - out[i].x * sp;
			if (fabsf(s.dphi) > COSTAS_RUNAWAY_DPHI && s.fr_state == FR_A1) {
				s.dphi = s.phi = 0.f;
				symsync_reset(s, a);
			}
			// eqlms_cccf_push
			{
				const float x2n = r.x * r.x + r.y * r.y, x2o = a.eq_x2[s.eq_head];
				a.eq_buf[s.eq_head] = r;
				a.eq_x2[s.eq_head] = x2n;
				s.eq_head = s.eq_head + 1 == D_EQ ? 0 : s.eq_head + 1;
				s.eq_x2sum = s.eq_x2sum + x2n - x2o;
				s.eq_count++;
			}
			if (!(s.symsync_out_idx & 1u)) continue;
			// eqlms_cccf_execute: sum conj(w_i) x_i, x_0 oldest
			cf y; y.x = 0.f; y.y = 0.f;
			if (HFDL_DM_STRICT_FAST & 1) {
				// the device: tap t in lane t + 1 of a row (lane 0 holds 0), row 0 the real part, row 1 the imaginary part
				float vr[16], vi[16];
				vr[0] = 0.f; vi[0] = 0.f;
				int idx = s.eq_head;
				for (int t = 0; t < D_EQ; t++) {
					const cf w = a.eq_w[t], x = a.eq_buf[idx];
					vr[t + 1] = w.x * x.x + w.y * x.y;
					vi[t + 1] = w.x * x.y + w.y * (-x.x);
					idx = idx + 1 == D_EQ ? 0 : idx + 1;
				}
				y.x = tree16(vr); y.y = tree16(vi);
			} else {
				int idx = s.eq_head;
				for (int t = 0; t < D_EQ; t++) {
					const cf w = a.eq_w[t], x = a.eq_buf[idx];
					y.x += w.x * x.x + w.y * x.y;
					y.y += w.x * x.y - w.y * x.x;
					idx = idx + 1 == D_EQ ? 0 : idx + 1;
				}
			}
			if (s.fr_state == FR_EQ_TRAIN) {
				// eqlms_cccf_step(d = known T symbol, d_hat = y)
				bool run = true;
				if (!s.eq_full) { if (s.eq_count < (uint32_t)D_EQ) run = false; else s.eq_full = 1; }
				if (run) {
					const float tv = t_symbol(s.T_idx) * ((s.bitmask & 1u) ? -1.0f : 1.0f);
					const float er = tv - y.x, ei = -(0.0f - y.y);
					int idx = s.eq_head;
					for (int t = 0; t < D_EQ; t++) {
						const cf x = a.eq_buf[idx];
						const float pr = er * x.x - ei * x.y, pi = er * x.y + ei * x.x;
						a.eq_w[t].x = a.eq_w[t].x + EQ_STEP * pr / s.eq_x2sum;
						a.eq_w[t].y = a.eq_w[t].y + EQ_STEP * pi / s.eq_x2sum;
						idx = idx + 1 == D_EQ ? 0 : idx + 1;
					}
				}
				s.T_idx++;
			}
			if (io.tap_symbols) io.tap_symbols[nsym] = y;
			nsym++;
			if (HFDL_DM_STRICT_FAST & 8) on_symbol(s, s, a, T, io, y, level, NearestSlicer{T.psk_pts});
			else on_symbol(s, s, a, T, io, y, level, TableSlicer{T.psk_pts});
		}
	}
	if (io.tap_counts) io.tap_counts[1] = nsym;
	return n_out;
}

}  // namespace hfdl
