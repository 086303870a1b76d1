// demod_core.h -- per-channel HFDL demodulator, one block of channelizer output per call (gfx950).
//
// Replaces the body of hfdl_decoder_thread after fastddc_inv_cc (reference src/hfdl.c:676-892):
// msresamp -> AGC -> matched filter -> symsync -> Costas -> LMS equaliser -> M-PSK slicer -> preamble
// correlator / framer.  Mapping: ONE WAVEFRONT PER CHANNEL.  The feed-forward stages (resampler, matched
// filter) run with lanes over output samples; the feedback stages (AGC, timing/carrier loops, equaliser,
// framer FSM) are one data-dependent recurrence per channel and run wave-uniformly, so every branch of the
// framer is a scalar branch and no lane ever diverges.  Filter windows live in LDS, loop scalars in registers.
//
// The file is plain C++ on purpose: tests/hostsim compiles it with g++ (LANES=1) to check the control logic
// against the oracle on a machine without a GPU.  Nothing in the shipped library takes that path.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HFDL_FN __device__ inline
#define HFDL_HD __host__ __device__ inline
#define HFDL_LANE ((int)threadIdx.x)
#define HFDL_LANES 64
#define HFDL_SYNC() __syncthreads()
#define HFDL_POPC64(x) __popcll(x)
#define HFDL_POPC32(x) __popc(x)
#define HFDL_ATOMIC_INC(p) atomicAdd((p), 1)
#else
#define HFDL_FN static inline
#define HFDL_HD static inline
#define HFDL_LANE 0
#define HFDL_LANES 1
#define HFDL_SYNC() do {} while (0)
#define HFDL_POPC64(x) __builtin_popcountll(x)
#define HFDL_POPC32(x) __builtin_popcount(x)
#define HFDL_ATOMIC_INC(p) ((*(p))++)
#endif

namespace hfdl {

struct cf { float x, y; };

constexpr int D_RS_NPFB = 256, D_RS_TAPS = 14, D_SS_NPFB = 16, D_SS_TAPS = 18, D_EQ = 15, D_MF = 19;
constexpr int MAX_DATA_SYMBOLS = 168 * 30;

// framer constants, src/hfdl.c:29-46
constexpr int A_LEN = 127, M1_LEN = 127, M2_LEN = 15, T_LEN = 15, DATA_FRAME_LEN = 30;
constexpr int SINGLE_SLOT_FRAME_LEN = 448 + (2 * 127 + 127 + 15 + 9 * 15) + 72 * 45;
enum { SAMPLER_BITS = 1, SAMPLER_SYMBOLS = 2, SAMPLER_SKIP = 3 };
enum { FR_A1 = 1, FR_A2, FR_M1, FR_M2_SKIP, FR_EQ_TRAIN, FR_DATA_1, FR_DATA_2 };

// mode table, src/hfdl.c:81-138: {bits/symbol, data segments, code-rate denominator, interleaver column shift}
struct ModeParams { int arity, segments, code_rate, col_shift; };
HFDL_HD ModeParams mode_params(int m)
{
	ModeParams p;
	p.arity = (m & 3) == 0 ? 1 : (m & 3);
	p.segments = (m & 4) ? 168 : 72;
	p.code_rate = (m & 3) == 0 ? 4 : 2;
	p.col_shift = (m & 4) ? 23 : 17;
	return p;
}

struct ChanScalars {
	uint32_t rs_phase;
	float agc_g, agc_y2;
	float ss_rate, ss_del, ss_tau, ss_bf, ss_q, ss_qhat, ss_v1;
	int32_t ss_b, ss_head;
	uint32_t ss_decim;
	float phi, dphi, err;
	float eq_x2sum;
	uint32_t eq_count;
	int32_t eq_full, eq_head;
	uint64_t bits_hi, bits_lo;
	int32_t training_n, data_n, use_data, data_slot;
	uint64_t symbol_cnt, sample_cnt, pdu_sample_index;
	int32_t s_state, fr_state, data_arity, cur_arity, symbols_wanted, search_retries;
	int32_t eq_train_seq_cnt, data_segment_cnt, train_total, train_bad, T_idx, M1;
	uint32_t bitmask, symsync_out_idx, nf_clk;
	float frame_symbol_cnt, freq_err_hz, signal_level, noise_floor;
	// observability: the per-channel StatsD counters of the reference's hot path (src/hfdl.c:818,828,840; doc/STATSD_METRICS.md)
	uint32_t cnt_a2_found, cnt_m1_found, cnt_m1_not_found, cnt_frames;
	uint32_t ev_flags;             // EV_*: resets that happened inside on_symbol(), for the register-resident device windows
};
enum { EV_SS_RESET = 1, EV_EQ_RESET = 2 };

struct ChanArrays {
	cf rs_hist[D_RS_TAPS - 1];     // [0] = most recent channelizer sample of the previous block
	cf mf_hist[D_MF - 1];          // [0] = most recent AGC output of the previous block
	cf ss_mf[D_SS_TAPS], ss_dmf[D_SS_TAPS];   // circular, ss_head = newest
	cf eq_w[D_EQ], eq_buf[D_EQ];   // eq_buf circular, eq_head = oldest
	float eq_x2[D_EQ];
	cf training[T_LEN];
};

struct ChanState { ChanScalars s; ChanArrays a; };

// what a finished frame hands to the burst decoder (K5)
struct FrameRec {
	int32_t channel, slot, mode, bitmask_lsb;
	float freq_err_hz, signal_level, noise_floor;
	int32_t train_bad, train_total, pad;
	uint64_t sample_index;
};

// constant tables as the kernels see them
struct DemodConst {
	const float *rs_h;             // [256][14]
	uint32_t rs_step;
	const float *mf;               // [19]
	const float *ss_mf, *ss_dmf;   // [16][18]  (the kernel stages these in LDS)
	float lf_b0, lf_a1, ss_rate_adj;
	const float *eq_h0;            // [15]
	uint64_t a_hi, a_lo;
	const uint64_t *m1_hi, *m1_lo; // [8]
	const float *corr_tab;         // [128]: 2.0f * m / 127.0f - 1.0f for m matching bits (the reference's expression, src/hfdl.c:781)
};

// per-block scratch (LDS) and outputs (global)
struct BlockIo {
	cf *rs, *agc, *mf;             // scratch [cap]
	float *lvl;                    // scratch [cap]
	int cap;
	cf *data;                      // global [2][MAX_DATA_SYMBOLS] of this channel
	FrameRec *frames;              // global queue
	int *frame_count;
	int frame_cap;
	// optional stage taps (global, this channel), null when disabled
	cf *tap_resampled, *tap_mf, *tap_symbols;
	float *tap_level;
	int *tap_counts;               // [2]: resampled count, symbol count
	int channel;
};

HFDL_HD void chan_state_init(ChanState &st, const float *eq_h0)
{
	ChanScalars &s = st.s;
	ChanArrays &a = st.a;
	char *p = (char *)&st;
	for (unsigned i = 0; i < sizeof(ChanState); i++) p[i] = 0;
	s.agc_g = 1.0f; s.agc_y2 = 1.0f;          // agc_crcf_create + reset
	s.noise_floor = 1.0f;                     // src/hfdl.c:490
	s.ss_rate = 1.5f; s.ss_del = 1.5f;        // k / k_out = 3 / 2
	for (int i = 0; i < D_EQ; i++) { a.eq_w[i].x = eq_h0[i]; a.eq_w[i].y = 0.f; }
	// framer_reset, src/hfdl.c:974-991
	s.fr_state = FR_A1; s.symbols_wanted = 1; s.cur_arity = 1; s.s_state = SAMPLER_BITS;
}

// ---------------- modem (liquid-dsp semantics, see oracle/fec_restated.c for the cited restatement) ----------------

HFDL_FN uint32_t gray_enc(uint32_t b) { return b ^ (b >> 1); }
HFDL_FN uint32_t gray_dec(uint32_t g) { uint32_t b = g; while (g >>= 1) b ^= g; return b; }

HFDL_FN cf psk_point(int arity, uint32_t sym)
{
	cf y;
	if (arity == 1) { y.x = sym ? -1.0f : 1.0f; y.y = 0.0f; return y; }
	const uint32_t M = 1u << arity;
	const float alpha = (float)M_PI / (float)M;
	const float ang = (float)gray_dec(sym) * 2 * alpha;
	y.x = cosf(ang); y.y = sinf(ang);
	return y;
}

HFDL_FN uint32_t psk_slice(int arity, cf x, float *phase_error)
{
	uint32_t sym;
	cf xh;
	if (arity == 1) {
		sym = (x.x > 0) ? 0 : 1;
		xh.x = sym ? -1.0f : 1.0f; xh.y = 0.0f;
	} else {
		const uint32_t M = 1u << arity;
		const float alpha = (float)M_PI / (float)M;
		float theta = atan2f(x.y, x.x);
		theta -= (float)M_PI * (1.0f - 1.0f / (float)M);
		if (theta < -(float)M_PI) theta += 2 * (float)M_PI;
		uint32_t s = 0;
		float v = theta;
		for (int k = arity - 1; k >= 0; k--) {
			const float ref = (float)(1u << k) * alpha;
			s <<= 1;
			if (v > 0) { s |= 1; v -= ref; } else { v += ref; }
		}
		sym = gray_enc(s);
		xh = psk_point(arity, sym);
	}
	if (phase_error) *phase_error = x.y * xh.x - x.x * xh.y;
	return sym;
}

HFDL_FN uint8_t soft_clamp(float v)
{
	int s = (int)(v + 127);
	return (uint8_t)(s > 255 ? 255 : (s < 0 ? 0 : s));
}

// modem_demodulate_soft: 255 = confident '1', soft[0] = MSB of the symbol
HFDL_FN void psk_soft(int arity, cf x, uint8_t *soft)
{
	if (arity == 1) {
		const float llr = -2.0f * x.x * 4.0f;
		soft[0] = soft_clamp(llr * 16);
		return;
	}
	const uint32_t sym = psk_slice(arity, x, nullptr);
	if (arity == 2) { soft[0] = (sym & 2) ? 255 : 0; soft[1] = (sym & 1) ? 255 : 0; return; }
	const uint32_t M = 1u << arity;
	const float gamma = 1.2f * (float)M;
	float d0[3], d1[3];
	cf c = psk_point(arity, sym);
	float er = x.x - c.x, ei = x.y - c.y;
	float d = er * er + ei * ei;
	for (int k = 0; k < arity; k++) {
		if ((sym >> (arity - k - 1)) & 1) { d0[k] = 4.0f; d1[k] = d; } else { d0[k] = d; d1[k] = 4.0f; }
	}
	const uint32_t lin = gray_dec(sym);
	for (int nb = 0; nb < 2; nb++) {
		const uint32_t ns = gray_enc((lin + (nb ? 1 : M - 1)) % M);
		c = psk_point(arity, ns);
		er = x.x - c.x; ei = x.y - c.y;
		d = er * er + ei * ei;
		for (int k = 0; k < arity; k++) {
			if ((ns >> (arity - k - 1)) & 1) { if (d < d1[k]) d1[k] = d; } else { if (d < d0[k]) d0[k] = d; }
		}
	}
	for (int k = 0; k < arity; k++) soft[k] = soft_clamp(((d0[k] - d1[k]) * gamma) * 16);
}

// ---------------- PDU header triage: FCS = CRC-16/X-25 over the header, stored low octet first ----------------
// hfdl_pdu_fcs_check (src/pdu.c:68-79), header length rules of mpdu_parse (src/mpdu.c:56-79) and spdu_parse (src/spdu.c:12,55-62)

HFDL_FN uint16_t crc16_x25(const uint8_t *p, uint32_t len)
{
	uint32_t crc = 0xFFFFu;
	for (uint32_t i = 0; i < len; i++) {
		crc ^= p[i];
		for (int b = 0; b < 8; b++) crc = (crc & 1u) ? (crc >> 1) ^ 0x8408u : crc >> 1;
	}
	return (uint16_t)(crc ^ 0xFFFFu);
}

// returns fcs status (0 good, 1 bad, 2 too short); kind: 0 SPDU, 1 MPDU downlink, 2 MPDU uplink
HFDL_FN int pdu_triage(const uint8_t *buf, uint32_t len, int *kind, uint32_t *hdr_len_out)
{
	uint32_t hdr_len;
	if ((buf[0] & 1u) == 0) {
		*kind = 0;
		hdr_len = 64;
		*hdr_len_out = hdr_len;
		if (len < 66) return 2;
	} else if (buf[0] & 0x2u) {
		*kind = 1;
		hdr_len = 6 + ((buf[0] >> 2) & 0xFu);
	} else {
		*kind = 2;
		const uint32_t aircraft_cnt = ((buf[0] & 0x70u) >> 4) + 1;
		hdr_len = 2;
		for (uint32_t i = 0; i < aircraft_cnt; i++) {
			if (len < hdr_len + 2) { *hdr_len_out = hdr_len; return 2; }
			hdr_len += 2 + (buf[hdr_len + 1] >> 4);
		}
	}
	*hdr_len_out = hdr_len;
	if (len < hdr_len + 2) return 2;
	const uint16_t rx = (uint16_t)(buf[hdr_len] | (buf[hdr_len + 1] << 8));
	return rx == crc16_x25(buf, hdr_len) ? 0 : 1;
}

// ---------------- helpers of the sequential stage ----------------

#if HFDL_LANES > 1
// lane i <- src of lane i-1; lane 0 <- fill          (DPP wave_shr:1, GFX9)
HFDL_FN float wave_shr1(float fill, float src)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(src), 0x138, 0xf, 0xf, false));
}
// lane i <- src of lane i+1; lane 63 <- fill         (DPP wave_shl:1, GFX9)
HFDL_FN float wave_shl1(float fill, float src)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(src), 0x130, 0xf, 0xf, false));
}
// Sum of v over lanes 0..31 (two DPP rows), returned wave-uniform.
HFDL_FN float row32_sum(float v)
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
HFDL_FN float row16_sum(float v)
{
	int x = __float_as_int(v);
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true)));
	return __int_as_float(__builtin_amdgcn_readlane(x, 15));
}
#endif

// sum_t h[t] * win[(head - t) mod 18] : polyphase branch output, newest sample first
HFDL_FN cf bank_dot(const float *h, const cf *win, int head)
{
#if HFDL_LANES > 1
	// lanes 0..15 take taps t and t+16 (18 taps), then one row reduction per component
	const int t = HFDL_LANE & 15;
	int i0 = head - t; if (i0 < 0) i0 += D_SS_TAPS;
	float pr = h[t] * win[i0].x, pi = h[t] * win[i0].y;
	if (t + 16 < D_SS_TAPS) {
		int i1 = head - t - 16; if (i1 < 0) i1 += D_SS_TAPS;
		pr += h[t + 16] * win[i1].x; pi += h[t + 16] * win[i1].y;
	}
	if (HFDL_LANE >= 16) { pr = 0.f; pi = 0.f; }
	cf r; r.x = row16_sum(pr); r.y = row16_sum(pi);
	return r;
#else
	float ar = 0, ai = 0;
	int idx = head;
	for (int t = 0; t < D_SS_TAPS; t++) {
		ar += h[t] * win[idx].x;
		ai += h[t] * win[idx].y;
		idx = idx == 0 ? D_SS_TAPS - 1 : idx - 1;
	}
	cf y; y.x = ar; y.y = ai;
	return y;
#endif
}

HFDL_FN void symsync_reset(ChanScalars &s, ChanArrays &a)
{
	// symsync_crcf_reset clears the matched-filter window only
	for (int i = 0; i < D_SS_TAPS; i++) { a.ss_mf[i].x = 0.f; a.ss_mf[i].y = 0.f; }
	s.ss_rate = 1.5f; s.ss_del = 1.5f;
	s.ss_b = 0; s.ss_bf = 0.f; s.ss_tau = 0.f; s.ss_q = 0.f; s.ss_qhat = 0.f;
	s.ss_decim = 0; s.ss_v1 = 0.f;
	s.ev_flags |= EV_SS_RESET;
}

HFDL_FN void eq_reset(ChanScalars &s, ChanArrays &a, const float *eq_h0)
{
	for (int i = 0; i < D_EQ; i++) {
		a.eq_w[i].x = eq_h0[i]; a.eq_w[i].y = 0.f;
		a.eq_buf[i].x = 0.f; a.eq_buf[i].y = 0.f;
		a.eq_x2[i] = 0.f;
	}
	s.eq_x2sum = 0.f; s.eq_count = 0; s.eq_full = 0; s.eq_head = 0;
	s.ev_flags |= EV_EQ_RESET;
}

HFDL_FN void framer_reset(ChanScalars &s, ChanArrays &a, const float *eq_h0)      // src/hfdl.c:968-991
{
	s.fr_state = FR_A1;
	s.symbols_wanted = 1;
	s.search_retries = 0;
	s.cur_arity = 1;
	s.train_total = s.train_bad = 0;
	s.T_idx = 0;
	s.use_data = 0;
	eq_reset(s, a, eq_h0);
	s.data_n = 0;
	s.training_n = 0;
	symsync_reset(s, a);
	s.s_state = SAMPLER_BITS;
	s.bitmask = 0;
}

HFDL_FN int bits_correlate(uint64_t hi, uint64_t lo, uint64_t thi, uint64_t tlo)
{
	return 127 - HFDL_POPC64((hi ^ thi) & 0x7FFFFFFFFFFFFFFFull) - HFDL_POPC64(lo ^ tlo);
}

HFDL_FN float t_symbol(int idx)          // T = 0x9AF, bit 14 first; BPSK 0 -> +1
{
	if (idx > 14) idx = 14;
	return ((0x9AFu >> (14 - idx)) & 1u) ? -1.0f : 1.0f;
}

// everything after the equaliser for one on-time symbol: src/hfdl.c:737-891
HFDL_FN void on_symbol(ChanScalars &s, ChanArrays &a, const DemodConst &T, const BlockIo &io, cf sym, float level)
{
	float perr;
	uint32_t bits = psk_slice(s.cur_arity, sym, &perr);
	{   // costas_cccf_adjust, :276-281
		const float e = 0.5f * (fabsf(perr + 1.0f) - fabsf(perr - 1.0f));
		s.err = e;
		s.phi += 0.1f * e;
		s.dphi += (0.047f * 0.1f * 0.1f) * e;
	}
	s.symbol_cnt++;
	if (s.symbol_cnt >= (uint64_t)(13 * SINGLE_SLOT_FRAME_LEN) && s.fr_state == FR_A1) {
		s.symbol_cnt = 0;
		s.dphi = s.phi = 0.0f;
		symsync_reset(s, a);
	}
	if (s.s_state == SAMPLER_BITS) {
		bits ^= s.bitmask;
		for (int b = 0; b < s.cur_arity; b++, bits >>= 1) {
			s.bits_hi = ((s.bits_hi << 1) | (s.bits_lo >> 63)) & 0x7FFFFFFFFFFFFFFFull;
			s.bits_lo = (s.bits_lo << 1) | (bits & 1u);
		}
	} else if (s.s_state == SAMPLER_SYMBOLS) {
		if (s.use_data) {
			if (s.data_n < MAX_DATA_SYMBOLS) {
				if (HFDL_LANE == 0) io.data[s.data_slot * MAX_DATA_SYMBOLS + s.data_n] = sym;
				s.data_n++;
			}
		} else if (s.training_n < T_LEN) {
			a.training[s.training_n] = sym;
			s.training_n++;
		}
	}
	if (s.fr_state > FR_A1) {
		s.signal_level = (s.signal_level * s.frame_symbol_cnt + level) / (s.frame_symbol_cnt + 1.0f);
		s.frame_symbol_cnt += 1.0f;
	}
	if (s.symbols_wanted > 1) { s.symbols_wanted--; return; }

	switch (s.fr_state) {
	case FR_A1: {
		const float corr = T.corr_tab[bits_correlate(s.bits_hi, s.bits_lo, T.a_hi, T.a_lo)];
		if (fabsf(corr) > 0.36f) {
			s.bitmask = corr > 0.f ? 0u : ~0u;
			s.signal_level = level;
			s.frame_symbol_cnt = 1.0f;
			s.symbols_wanted = A_LEN;
			s.search_retries = 0;
			s.fr_state = FR_A2;
		}
		break; }
	case FR_A2: {
		const float corr = T.corr_tab[bits_correlate(s.bits_hi, s.bits_lo, T.a_hi, T.a_lo)];
		if (fabsf(corr) > 0.3f) {
			s.cnt_a2_found++;                    // statsd "demod.preamble.A2_found"
			s.pdu_sample_index = s.sample_cnt;
			s.freq_err_hz = (float)((double)(s.dphi * 1800) / (2.0 * M_PI));
			s.symbols_wanted = M1_LEN;
			s.search_retries = 0;
			s.fr_state = FR_M1;
		} else if (++s.search_retries >= 3) {
			framer_reset(s, a, T.eq_h0);
		}
		break; }
	case FR_M1: {
		float best = 0.f;
		int best_idx = -1;
		for (int m = 0; m < 8; m++) {
			const float corr = fabsf(T.corr_tab[bits_correlate(s.bits_hi, s.bits_lo, T.m1_hi[m], T.m1_lo[m])]);
			if (corr > best) { best = corr; best_idx = m; }
		}
		if (fabsf(best) > 0.3f) {
			const ModeParams mp = mode_params(best_idx);
			s.cnt_m1_found++;                    // "demod.preamble.M1_found"
			s.data_segment_cnt = mp.segments;
			s.data_arity = mp.arity;
			s.M1 = best_idx;
			s.symbols_wanted = M2_LEN;
			s.search_retries = 0;
			s.fr_state = FR_M2_SKIP;
			s.s_state = SAMPLER_SKIP;
		} else {
			s.cnt_m1_not_found++;                // "demod.preamble.errors.M1_not_found"
			framer_reset(s, a, T.eq_h0);
		}
		break; }
	case FR_M2_SKIP:
		s.training_n = 0;
		s.symbols_wanted = T_LEN;
		s.eq_train_seq_cnt = 9;
		s.fr_state = FR_EQ_TRAIN;
		s.s_state = SAMPLER_SYMBOLS;
		break;
	case FR_EQ_TRAIN: {
		// compute_train_bit_error_cnt, :952-966
		uint32_t seq = 0;
		for (int i = 0; i < T_LEN; i++) {
			uint32_t bit = (a.training[i].x > 0) ? 0u : 1u;
			bit ^= (s.bitmask & 1u);
			seq = (seq << 1) | bit;
		}
		s.train_total += T_LEN;
		s.train_bad += HFDL_POPC32(0x9AFu ^ seq);
		s.training_n = 0;
		if (s.eq_train_seq_cnt > 1) {
			s.eq_train_seq_cnt--;
			s.symbols_wanted = T_LEN;
			s.T_idx = 0;
		} else if (s.data_segment_cnt > 0) {
			s.symbols_wanted = DATA_FRAME_LEN / 2;
			s.fr_state = FR_DATA_1;
			s.cur_arity = s.data_arity;
			s.use_data = 1;
		} else {
			// end of frame: queue it for the burst decoder (decode_user_data + dispatch_pdu, :993-1080)
			if (HFDL_LANE == 0) {
				const int slot = HFDL_ATOMIC_INC(io.frame_count);
				if (slot < io.frame_cap) {
					FrameRec fr;
					fr.channel = io.channel; fr.slot = s.data_slot; fr.mode = s.M1; fr.bitmask_lsb = (int32_t)(s.bitmask & 1u);
					fr.freq_err_hz = s.freq_err_hz; fr.signal_level = s.signal_level; fr.noise_floor = s.noise_floor;
					fr.train_bad = s.train_bad; fr.train_total = s.train_total; fr.pad = 0;
					fr.sample_index = s.pdu_sample_index;
					io.frames[slot] = fr;
				}
			}
			s.data_slot ^= 1;
			s.cnt_frames++;
			framer_reset(s, a, T.eq_h0);
			s.symbol_cnt = 0;
		}
		break; }
	case FR_DATA_1:
		s.symbols_wanted = DATA_FRAME_LEN / 2;
		s.fr_state = FR_DATA_2;
		break;
	case FR_DATA_2:
		s.data_segment_cnt--;
		s.cur_arity = 1;
		s.use_data = 0;
		s.fr_state = FR_EQ_TRAIN;
		s.eq_train_seq_cnt = 1;
		s.symbols_wanted = T_LEN;
		s.T_idx = 0;
		break;
	}
}

// ---------------- one block ----------------

// returns the number of 5400-sps samples produced
HFDL_FN int demod_block(ChanScalars &s, ChanArrays &a, const DemodConst &T, const BlockIo &io, const cf *in, int n_in)
{
	const int lane = HFDL_LANE;
#if HFDL_LANES > 1
	unsigned long long tR0 = __builtin_amdgcn_s_memtime();
#endif

	// ---- R: arbitrary resampler, 24-bit fixed-point phase, lanes over outputs (msresamp_crcf_execute, src/hfdl.c:676)
	const uint64_t total = (uint64_t)n_in << 24;
	int n_out = 0;
	if ((uint64_t)s.rs_phase < total) n_out = (int)((total - s.rs_phase + T.rs_step - 1) / T.rs_step);
	if (n_out > io.cap - 4) n_out = io.cap - 4;      // cannot happen: cap is sized from the geometry (+8); the last 4 level slots carry phase cycles
	for (int k = lane; k < n_out; k += HFDL_LANES) {
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
	HFDL_SYNC();
	{
		cf nh;
		const int j = lane;
		if (HFDL_LANES > 1) {
			if (j < D_RS_TAPS - 1) nh = (n_in - 1 - j >= 0) ? in[n_in - 1 - j] : a.rs_hist[j - n_in];
			HFDL_SYNC();
			if (j < D_RS_TAPS - 1) a.rs_hist[j] = nh;
		} else {
			cf tmp[D_RS_TAPS - 1];
			for (int q = 0; q < D_RS_TAPS - 1; q++) tmp[q] = (n_in - 1 - q >= 0) ? in[n_in - 1 - q] : a.rs_hist[q - n_in];
			for (int q = 0; q < D_RS_TAPS - 1; q++) a.rs_hist[q] = tmp[q];
		}
	}
	s.rs_phase = (uint32_t)((uint64_t)s.rs_phase + (uint64_t)n_out * T.rs_step - total);
	if (io.tap_counts && lane == 0) io.tap_counts[0] = n_out;
	if (n_out < 1) return 0;

#if HFDL_LANES > 1
	unsigned long long tA0 = __builtin_amdgcn_s_memtime();
#endif
	// ---- A: AGC, a per-sample gain recurrence (agc_crcf_execute, src/hfdl.c:686)
	{
		float g = s.agc_g, y2 = s.agc_y2;
		const float alpha = 0.01f;
		for (int k = 0; k < n_out; k++) {
			const cf x = io.rs[k];
			cf y; y.x = x.x * g; y.y = x.y * g;
			const float e = y.x * y.x + y.y * y.y;
			y2 = (1.0f - alpha) * y2 + alpha * e;
#if HFDL_LANES > 1
			// exp(a ln y2) == 2^(a log2 y2): one v_log_f32 + one v_exp_f32 on the gain recurrence's critical path
			if (y2 > 1e-6f) g *= __builtin_amdgcn_exp2f(-0.5f * alpha * __builtin_amdgcn_logf(y2));
			if (g > 1e6f) g = 1e6f;
			io.agc[k] = y;
			io.lvl[k] = __builtin_amdgcn_rcpf(g);
#else
			if (y2 > 1e-6f) g *= expf(-0.5f * alpha * logf(y2));
			if (g > 1e6f) g = 1e6f;
			io.agc[k] = y;
			io.lvl[k] = 1.0f / g;
#endif
		}
		s.agc_g = g; s.agc_y2 = y2;
	}
	HFDL_SYNC();

#if HFDL_LANES > 1
	unsigned long long tM0 = __builtin_amdgcn_s_memtime();
#endif
	// ---- M: 19-tap matched filter, lanes over outputs (firfilt_crcf, src/hfdl.c:694-695)
	for (int k = lane; k < n_out; k += HFDL_LANES) {
		float ar = 0, ai = 0;
		for (int t = 0; t < D_MF; t++) {
			const int idx = k - t;
			const cf x = idx >= 0 ? io.agc[idx] : a.mf_hist[-idx - 1];
			ar += T.mf[t] * x.x;
			ai += T.mf[t] * x.y;
		}
		io.mf[k].x = ar; io.mf[k].y = ai;
	}
	HFDL_SYNC();
	{
		cf nh;
		const int j = lane;
		if (HFDL_LANES > 1) {
			if (j < D_MF - 1) nh = (n_out - 1 - j >= 0) ? io.agc[n_out - 1 - j] : a.mf_hist[j - n_out];
			HFDL_SYNC();
			if (j < D_MF - 1) a.mf_hist[j] = nh;
		} else {
			cf tmp[D_MF - 1];
			for (int q = 0; q < D_MF - 1; q++) tmp[q] = (n_out - 1 - q >= 0) ? io.agc[n_out - 1 - q] : a.mf_hist[q - n_out];
			for (int q = 0; q < D_MF - 1; q++) a.mf_hist[q] = tmp[q];
		}
	}
	if (io.tap_resampled) {
		for (int k = lane; k < n_out; k += HFDL_LANES) {
			io.tap_resampled[k] = io.rs[k];
			io.tap_mf[k] = io.mf[k];
			io.tap_level[k] = io.lvl[k];
		}
	}
	HFDL_SYNC();

#if HFDL_LANES > 1
	unsigned long long tS0 = __builtin_amdgcn_s_memtime();
#endif
#if HFDL_LANES > 1
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
		HFDL_SYNC();
		if (lane < D_SS_TAPS) {
			const int i = lane == 0 ? 0 : D_SS_TAPS - lane;
			a.ss_mf[i] = wmf; a.ss_dmf[i] = wdmf;
		}
		if (lane < D_EQ) { a.eq_buf[lane] = ebuf; a.eq_x2[lane] = ex2; a.eq_w[lane] = ew; }
		s.ss_head = 0;
		s.eq_head = 0;
		HFDL_SYNC();
	}
#else
	// ---- S: timing recovery, carrier loop, equaliser, slicer, framer -- wave-uniform (src/hfdl.c:696-891)
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
#if HFDL_LANES > 1
			float sp, cp;
			sincosf(s.phi, &sp, &cp);            // one shared range reduction
#else
			const float cp = cosf(s.phi), sp = sinf(s.phi);
#endif
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
#if HFDL_LANES > 1
			{   // lane t < 15 owns tap t
				const int t = lane & 15;
				int idx = s.eq_head + t; if (idx >= D_EQ) idx -= D_EQ;
				float pr = 0.f, pi = 0.f;
				if (lane < D_EQ) {
					const cf w = a.eq_w[t], x = a.eq_buf[idx];
					pr = w.x * x.x + w.y * x.y;
					pi = w.x * x.y - w.y * x.x;
				}
				y.x = row16_sum(pr); y.y = row16_sum(pi);
			}
#else
			{
				int idx = s.eq_head;
				for (int t = 0; t < D_EQ; t++) {
					const cf w = a.eq_w[t], x = a.eq_buf[idx];
					y.x += w.x * x.x + w.y * x.y;
					y.y += w.x * x.y - w.y * x.x;
					idx = idx + 1 == D_EQ ? 0 : idx + 1;
				}
			}
#endif
			if (s.fr_state == FR_EQ_TRAIN) {
				// eqlms_cccf_step(d = known T symbol, d_hat = y)
				bool run = true;
				if (!s.eq_full) { if (s.eq_count < (uint32_t)D_EQ) run = false; else s.eq_full = 1; }
				if (run) {
					const float tv = t_symbol(s.T_idx) * ((s.bitmask & 1u) ? -1.0f : 1.0f);
					const float er = tv - y.x, ei = -(0.0f - y.y);
#if HFDL_LANES > 1
					if (lane < D_EQ) {
						int idx = s.eq_head + lane; if (idx >= D_EQ) idx -= D_EQ;
						const cf x = a.eq_buf[idx];
						const float pr = er * x.x - ei * x.y, pi = er * x.y + ei * x.x;
						a.eq_w[lane].x = a.eq_w[lane].x + 0.1f * pr / s.eq_x2sum;
						a.eq_w[lane].y = a.eq_w[lane].y + 0.1f * pi / s.eq_x2sum;
					}
#else
					int idx = s.eq_head;
					for (int t = 0; t < D_EQ; t++) {
						const cf x = a.eq_buf[idx];
						const float pr = er * x.x - ei * x.y, pi = er * x.y + ei * x.x;
						a.eq_w[t].x = a.eq_w[t].x + 0.1f * pr / s.eq_x2sum;
						a.eq_w[t].y = a.eq_w[t].y + 0.1f * pi / s.eq_x2sum;
						idx = idx + 1 == D_EQ ? 0 : idx + 1;
					}
#endif
				}
				s.T_idx++;
			}
			if (io.tap_symbols && lane == 0) io.tap_symbols[nsym] = y;
			nsym++;
			on_symbol(s, a, T, io, y, level);
		}
	}
#endif
	if (io.tap_counts && lane == 0) io.tap_counts[1] = nsym;
#if HFDL_LANES > 1
	if (io.tap_counts && lane == 0) {
		unsigned long long tE = __builtin_amdgcn_s_memtime();
		io.tap_level[io.cap - 4] = (float)(tA0 - tR0); io.tap_level[io.cap - 3] = (float)(tM0 - tA0);
		io.tap_level[io.cap - 2] = (float)(tS0 - tM0); io.tap_level[io.cap - 1] = (float)(tE - tS0);
	}
#endif
	return n_out;
}

}  // namespace hfdl
