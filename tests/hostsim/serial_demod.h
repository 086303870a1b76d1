// serial_demod.h -- TEST HARNESS ONLY: a one-lane, plain-C++ form of the demodulator's block loop, written against the
// product's own state, tables, modem and framer FSM (dumphfdl_amd/csrc/demod_logic.h, compiled for the host behind the
// shims in hostsim.cpp).  The shipped block loop (dumphfdl_amd/csrc/demod_core.h) is device code -- register-resident
// windows, DPP reductions, hardware transcendentals -- and is gated on the GPU; this serial form exists so that the
// framer / sampler / slicer logic the device calls can be checked bit for bit against the oracle on a machine without a GPU.
#pragma once
#include "../../dumphfdl_amd/csrc/demod_logic.h"

namespace hfdl {

// sum_t h[t] * win[(head - t) mod 18] : polyphase branch output, newest sample first
static inline cf bank_dot(const float *h, const cf *win, int head)
{
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

// returns the number of 5400-sps samples produced
static inline int demod_block_serial(ChanScalars &s, ChanArrays &a, const DemodConst &T, const BlockIo &io, const cf *in, int n_in)
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
		const float alpha = 0.01f;
		for (int k = 0; k < n_out; k++) {
			const cf x = io.rs[k];
			cf y; y.x = x.x * g; y.y = x.y * g;
			const float e = y.x * y.x + y.y * y.y;
			y2 = (1.0f - alpha) * y2 + alpha * e;
			if (y2 > 1e-6f) g *= expf(-0.5f * alpha * logf(y2));
			if (g > 1e6f) g = 1e6f;
			io.agc[k] = y;
			io.lvl[k] = 1.0f / g;
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
		if (s.fr_state == FR_A1 && (++s.nf_clk & 0xFFu) == 0xFFu)
			s.noise_floor = 0.65f * s.noise_floor + 0.35f * fminf(s.noise_floor, level) + 1e-6f;

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
			const float cp = cosf(s.phi), sp = sinf(s.phi);
			cf r;
			r.x = out[i].x * cp + out[i].y * sp;
			r.y = out[i].y * cp - out[i].x * sp;
			if (fabsf(s.dphi) > 0.25f && s.fr_state == FR_A1) {
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
			{
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
						a.eq_w[t].x = a.eq_w[t].x + 0.1f * pr / s.eq_x2sum;
						a.eq_w[t].y = a.eq_w[t].y + 0.1f * pi / s.eq_x2sum;
						idx = idx + 1 == D_EQ ? 0 : idx + 1;
					}
				}
				s.T_idx++;
			}
			if (io.tap_symbols) io.tap_symbols[nsym] = y;
			nsym++;
			on_symbol(s, s, a, T, io, y, level, TableSlicer{T.psk_pts});
		}
	}
	if (io.tap_counts) io.tap_counts[1] = nsym;
	return n_out;
}

}  // namespace hfdl
