// demod_core.h -- per-channel HFDL demodulator, one block of channelizer output per call (gfx950, device only).
//
// Replaces the body of hfdl_decoder_thread after fastddc_inv_cc (reference src/hfdl.c:676-892):
// msresamp -> AGC -> matched filter -> symsync -> Costas -> LMS equaliser -> M-PSK slicer -> preamble
// correlator / framer.  Mapping: ONE WAVEFRONT PER CHANNEL.  The feed-forward stages (resampler, matched
// filter) run with lanes over output samples; the feedback stages (AGC, timing/carrier loops, equaliser,
// framer FSM) are one data-dependent recurrence per channel and run wave-uniformly, so every branch of the
// framer is a scalar branch and no lane ever diverges.  The symsync / equaliser windows live in registers, one tap per lane.
// State, modem and the framer FSM are in demod_logic.h.
#pragma once
#include <hip/hip_runtime.h>
#include "demod_logic.h"

namespace hfdl {

// lane i <- src of lane i-1; lane 0 <- fill          (DPP wave_shr:1, GFX9)
__device__ inline float wave_shr1(float fill, float src)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(src), 0x138, 0xf, 0xf, false));
}
// lane i <- src of lane i+1; lane 63 <- fill         (DPP wave_shl:1, GFX9)
__device__ inline float wave_shl1(float fill, float src)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(src), 0x130, 0xf, 0xf, false));
}
// Sum of v over lanes 0..31 (two DPP rows), returned wave-uniform.
__device__ inline float row32_sum(float v)
{
	int x = __float_as_int(v);
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true)));
	return __int_as_float(__builtin_amdgcn_readlane(x, 15)) + __int_as_float(__builtin_amdgcn_readlane(x, 31));
}
// Sum of v over lanes 0..15, returned wave-uniform.  Four DPP row_shr steps build an inclusive scan inside the
// 16-lane row (lanes shifted in from outside the row read 0), lane 15 then holds the total.
__device__ inline float row16_sum(float v)
{
	int x = __float_as_int(v);
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true)));
	return __int_as_float(__builtin_amdgcn_readlane(x, 15));
}

// ---------------- one block ----------------

// returns the number of 5400-sps samples produced
__device__ inline int demod_block(ChanScalars &s, ChanArrays &a, const DemodConst &T, const BlockIo &io, const cf *in, int n_in)
{
	const int lane = (int)threadIdx.x;
	unsigned long long tR0 = __builtin_amdgcn_s_memtime();

	// ---- R: arbitrary resampler, 24-bit fixed-point phase, lanes over outputs (msresamp_crcf_execute, src/hfdl.c:676)
	const uint64_t total = (uint64_t)n_in << 24;
	int n_out = 0;
	if ((uint64_t)s.rs_phase < total) n_out = (int)((total - s.rs_phase + T.rs_step - 1) / T.rs_step);
	if (n_out > io.cap - 4) n_out = io.cap - 4;      // cannot happen: cap is sized from the geometry (+8); the last 4 level slots carry phase cycles
	for (int k = lane; k < n_out; k += 64) {
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
	__syncthreads();
	{
		cf nh;
		const int j = lane;
		if (j < D_RS_TAPS - 1) nh = (n_in - 1 - j >= 0) ? in[n_in - 1 - j] : a.rs_hist[j - n_in];
		__syncthreads();
		if (j < D_RS_TAPS - 1) a.rs_hist[j] = nh;
	}
	s.rs_phase = (uint32_t)((uint64_t)s.rs_phase + (uint64_t)n_out * T.rs_step - total);
	if (io.tap_counts && lane == 0) io.tap_counts[0] = n_out;
	if (n_out < 1) return 0;

	unsigned long long tA0 = __builtin_amdgcn_s_memtime();
	// ---- A: AGC, a per-sample gain recurrence (agc_crcf_execute, src/hfdl.c:686)
	{
		float g = s.agc_g, y2 = s.agc_y2;
		const float alpha = 0.01f;
		for (int k = 0; k < n_out; k++) {
			const cf x = io.rs[k];
			cf y; y.x = x.x * g; y.y = x.y * g;
			const float e = y.x * y.x + y.y * y.y;
			y2 = (1.0f - alpha) * y2 + alpha * e;
			// exp(a ln y2) == 2^(a log2 y2): one v_log_f32 + one v_exp_f32 on the gain recurrence's critical path
			if (y2 > 1e-6f) g *= __builtin_amdgcn_exp2f(-0.5f * alpha * __builtin_amdgcn_logf(y2));
			if (g > 1e6f) g = 1e6f;
			io.agc[k] = y;
			io.lvl[k] = __builtin_amdgcn_rcpf(g);
		}
		s.agc_g = g; s.agc_y2 = y2;
	}
	__syncthreads();

	unsigned long long tM0 = __builtin_amdgcn_s_memtime();
	// ---- M: 19-tap matched filter, lanes over outputs (firfilt_crcf, src/hfdl.c:694-695)
	for (int k = lane; k < n_out; k += 64) {
		float ar = 0, ai = 0;
		for (int t = 0; t < D_MF; t++) {
			const int idx = k - t;
			const cf x = idx >= 0 ? io.agc[idx] : a.mf_hist[-idx - 1];
			ar += T.mf[t] * x.x;
			ai += T.mf[t] * x.y;
		}
		io.mf[k].x = ar; io.mf[k].y = ai;
	}
	__syncthreads();
	{
		cf nh;
		const int j = lane;
		if (j < D_MF - 1) nh = (n_out - 1 - j >= 0) ? io.agc[n_out - 1 - j] : a.mf_hist[j - n_out];
		__syncthreads();
		if (j < D_MF - 1) a.mf_hist[j] = nh;
	}
	if (io.tap_resampled) {
		for (int k = lane; k < n_out; k += 64) {
			io.tap_resampled[k] = io.rs[k];
			io.tap_mf[k] = io.mf[k];
			io.tap_level[k] = io.lvl[k];
		}
	}
	__syncthreads();

	unsigned long long tS0 = __builtin_amdgcn_s_memtime();
	// ---- S: timing recovery, carrier loop, equaliser, slicer, framer -- wave-uniform (src/hfdl.c:696-891)
	// Device form: the symsync / equaliser windows live in REGISTERS, one tap per lane (pushed with a DPP wave shift),
	// every FIR is one multiply per lane plus a DPP row reduction, the matched and derivative-matched branch outputs are
	// reduced together, and the branch taps of the next output are fetched as soon as its bank index is known.
	int nsym = 0;
	{
		cf wmf, wdmf, ebuf, ew;
		float ex2;
		{
			int i = s.ss_head - lane; if (i < 0) i += D_SS_TAPS;
			const bool in = lane < D_SS_TAPS;
			wmf = in ? a.ss_mf[i] : cf{0.f, 0.f};
			wdmf = in ? a.ss_dmf[i] : cf{0.f, 0.f};
			int j = s.eq_head + lane; if (j >= D_EQ) j -= D_EQ;
			const bool ine = lane < D_EQ;
			ebuf = ine ? a.eq_buf[j] : cf{0.f, 0.f};
			ex2 = ine ? a.eq_x2[j] : 0.f;
			ew = ine ? a.eq_w[lane] : cf{0.f, 0.f};
		}
		s.ev_flags = 0;
		const int tapl = lane < D_SS_TAPS ? lane : 0;
		const float tapm = lane < D_SS_TAPS ? 1.0f : 0.0f;
		int bi = s.ss_b < 0 ? 0 : (s.ss_b >= D_SS_NPFB ? D_SS_NPFB - 1 : s.ss_b);
		float hmf = T.ss_mf[bi * D_SS_TAPS + tapl] * tapm, hdm = T.ss_dmf[bi * D_SS_TAPS + tapl] * tapm;
		for (int k = 0; k < n_out; k++, s.sample_cnt++) {
			const cf mfo = io.mf[k];
			const float level = io.lvl[k];
			if (s.fr_state == FR_A1 && (++s.nf_clk & 0xFFu) == 0xFFu)
				s.noise_floor = 0.65f * s.noise_floor + 0.35f * fminf(s.noise_floor, level) + 1e-6f;
			// symsync_crcf_execute: push one sample into both windows (lane 0 = newest)
			wmf.x = wave_shr1(mfo.x, wmf.x); wmf.y = wave_shr1(mfo.y, wmf.y);
			wdmf.x = wave_shr1(mfo.x, wdmf.x); wdmf.y = wave_shr1(mfo.y, wdmf.y);
			// at most 4 outputs per input sample, kept in named registers: an array indexed by `produced` would live in
			// scratch memory, and beside the HBM-saturating fold kernel every scratch access is a multi-microsecond stall
			cf out0 = cf{0.f, 0.f}, out1 = out0, out2 = out0, out3 = out0;
			int produced = 0;
			while (s.ss_b < D_SS_NPFB && produced < 4) {
				cf m, d;
				m.x = row32_sum(hmf * wmf.x); m.y = row32_sum(hmf * wmf.y);
				d.x = row32_sum(hdm * wdmf.x); d.y = row32_sum(hdm * wdmf.y);
				{
					cf o; o.x = m.x / 3.0f; o.y = m.y / 3.0f;
					if (produced == 0) out0 = o; else if (produced == 1) out1 = o; else if (produced == 2) out2 = o; else out3 = o;
				}
				if (s.ss_decim == 2) {
					s.ss_decim = 0;
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
				if (s.ss_b < D_SS_NPFB) {       // another output from this input sample: its branch taps are needed now
					bi = s.ss_b < 0 ? 0 : s.ss_b;
					hmf = T.ss_mf[bi * D_SS_TAPS + tapl] * tapm; hdm = T.ss_dmf[bi * D_SS_TAPS + tapl] * tapm;
				}
			}
			s.ss_tau -= 1.0f;
			s.ss_bf -= (float)D_SS_NPFB;
			s.ss_b -= D_SS_NPFB;
			// branch of the next input sample is known now: fetch its taps while the carrier loop / equaliser run
			bi = s.ss_b < 0 ? 0 : (s.ss_b >= D_SS_NPFB ? D_SS_NPFB - 1 : s.ss_b);
			hmf = T.ss_mf[bi * D_SS_TAPS + tapl] * tapm; hdm = T.ss_dmf[bi * D_SS_TAPS + tapl] * tapm;

			for (int i = 0; i < produced; i++, s.symsync_out_idx++) {
				s.phi += s.dphi;
				if (s.phi > (float)M_PI) s.phi -= (float)(2.0 * M_PI);
				else if (s.phi < -(float)M_PI) s.phi += (float)(2.0 * M_PI);
				// |phi| <= pi: the hardware sin/cos (argument in revolutions) needs no range reduction
				const float rev = s.phi * 0.15915494309189535f;
				const float sp = __builtin_amdgcn_sinf(rev), cp = __builtin_amdgcn_cosf(rev);
				const cf oi = i == 0 ? out0 : (i == 1 ? out1 : (i == 2 ? out2 : out3));
				cf r;
				r.x = oi.x * cp + oi.y * sp;
				r.y = oi.y * cp - oi.x * sp;
				if (fabsf(s.dphi) > 0.25f && s.fr_state == FR_A1) {
					s.dphi = s.phi = 0.f;
					symsync_reset(s, a);
				}
				// eqlms_cccf_push: lane 0 = oldest ... lane 14 = newest
				{
					const float x2n = r.x * r.x + r.y * r.y;
					const float x2o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex2), 0));
					const bool ins = lane == D_EQ;          // park the new sample in lane 15, then shift everything down one lane
					ebuf.x = wave_shl1(0.f, ins ? r.x : ebuf.x); ebuf.y = wave_shl1(0.f, ins ? r.y : ebuf.y);
					ex2 = wave_shl1(0.f, ins ? x2n : ex2);
					s.eq_x2sum = s.eq_x2sum + x2n - x2o;
					s.eq_count++;
				}
				if (s.symsync_out_idx & 1u) {
					cf y;
					{
						const bool act = lane < D_EQ;
						y.x = row16_sum(act ? ew.x * ebuf.x + ew.y * ebuf.y : 0.f);
						y.y = row16_sum(act ? ew.x * ebuf.y - ew.y * ebuf.x : 0.f);
					}
					if (s.fr_state == FR_EQ_TRAIN) {
						bool run = true;
						if (!s.eq_full) { if (s.eq_count < (uint32_t)D_EQ) run = false; else s.eq_full = 1; }
						if (run) {
							const float tv = t_symbol(s.T_idx) * ((s.bitmask & 1u) ? -1.0f : 1.0f);
							const float er = tv - y.x, ei = -(0.0f - y.y);
							const float pr = er * ebuf.x - ei * ebuf.y, pi = er * ebuf.y + ei * ebuf.x;
							ew.x = ew.x + 0.1f * pr / s.eq_x2sum;
							ew.y = ew.y + 0.1f * pi / s.eq_x2sum;
						}
						s.T_idx++;
					}
					if (io.tap_symbols && lane == 0) io.tap_symbols[nsym] = y;
					nsym++;
					on_symbol(s, a, T, io, y, level);
				}
				if (s.ev_flags) {       // a reset ran (carrier runaway, framer reset, timeout): mirror it in the register windows
					if (s.ev_flags & EV_SS_RESET) { wmf.x = 0.f; wmf.y = 0.f; }
					if (s.ev_flags & EV_EQ_RESET) {
						ebuf.x = 0.f; ebuf.y = 0.f; ex2 = 0.f;
						ew.x = lane < D_EQ ? T.eq_h0[lane < D_EQ ? lane : 0] : 0.f; ew.y = 0.f;
					}
					s.ev_flags = 0;
					bi = s.ss_b < 0 ? 0 : (s.ss_b >= D_SS_NPFB ? D_SS_NPFB - 1 : s.ss_b);
					hmf = T.ss_mf[bi * D_SS_TAPS + tapl] * tapm; hdm = T.ss_dmf[bi * D_SS_TAPS + tapl] * tapm;
				}
			}
		}
		// back to the canonical array form (newest symsync sample at index 0, oldest equaliser sample at index 0)
		__syncthreads();
		if (lane < D_SS_TAPS) {
			const int i = lane == 0 ? 0 : D_SS_TAPS - lane;
			a.ss_mf[i] = wmf; a.ss_dmf[i] = wdmf;
		}
		if (lane < D_EQ) { a.eq_buf[lane] = ebuf; a.eq_x2[lane] = ex2; a.eq_w[lane] = ew; }
		s.ss_head = 0;
		s.eq_head = 0;
		__syncthreads();
	}
	if (io.tap_counts && lane == 0) {
		io.tap_counts[1] = nsym;
		unsigned long long tE = __builtin_amdgcn_s_memtime();
		io.tap_level[io.cap - 4] = (float)(tA0 - tR0); io.tap_level[io.cap - 3] = (float)(tM0 - tA0);
		io.tap_level[io.cap - 2] = (float)(tS0 - tM0); io.tap_level[io.cap - 1] = (float)(tE - tS0);
	}
	return n_out;
}

}  // namespace hfdl
