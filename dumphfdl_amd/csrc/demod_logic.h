// demod_logic.h -- per-channel HFDL demodulator: state, modem, PDU triage and the framer FSM (gfx950 device code).
//
// The lane-independent part of the body of hfdl_decoder_thread after fastddc_inv_cc (reference src/hfdl.c:676-892):
// everything here is one wave-uniform recurrence per channel -- M-PSK slicer / soft de-mapper, Costas adjust, preamble
// correlator, framer / sampler state machine, resets, FCS / header triage.  The lane-parallel block loop that drives it is
// demod_core.h.  Plain C++ expressions only (no DPP, no LDS): written once, for the device; the two host-side users are
// Demod::init (initial channel state) and the burst-decode batch entry point (mode table).
// tests/hostsim compiles this file with g++ behind a few shims it defines itself (test harness, never shipped).
#pragma once
#include <stdint.h>
#include <math.h>

#define HFDL_FN __device__ inline
#define HFDL_HD __host__ __device__ inline
#ifndef HFDL_ATAN2F                 // the test-only build -DHFDL_DM_STRICT substitutes a fixed fp32 sequence shared with the oracle (demod_kernels.hip)
#define HFDL_ATAN2F atan2f
#endif

namespace hfdl {

struct cf { float x, y; };

constexpr int D_RS_NPFB = 256, D_RS_TAPS = 14, D_SS_NPFB = 16, D_SS_TAPS = 18, D_EQ = 15, D_MF = 19;
constexpr int MAX_DATA_SYMBOLS = 168 * 30;

// framer constants, src/hfdl.c:29-46
constexpr int PREKEY_LEN = 448, A_LEN = 127, M1_LEN = 127, M2_LEN = 15, T_LEN = 15, DATA_FRAME_LEN = 30, PREAMBLE_T_SEQS = 9;
constexpr int SINGLE_SLOT_FRAME_LEN = PREKEY_LEN + (2 * A_LEN + M1_LEN + M2_LEN + PREAMBLE_T_SEQS * T_LEN) + 72 * (DATA_FRAME_LEN + T_LEN);
// The numeric constants of the reference's hot path this file and demod_core.h use, BY NAME, so that they can be read back and compared
// with the reference's own text: tests/golden/hfdl_constants.json (parsed from src/hfdl.c by tests/golden/make_constants.py) meets them
// through tests/hostsim on the CPU and through a read-back kernel on the device (hfdl_constants() below).
constexpr float CORR_THRESHOLD_A1 = 0.36f, CORR_THRESHOLD_A2 = 0.3f, CORR_THRESHOLD_M1 = 0.3f;      // src/hfdl.c:42-44
constexpr int MAX_SEARCH_RETRIES = 3, NO_FRAME_TIMEOUT_FRAMES = 13;                                  // :45, :613
constexpr int HFDL_SYMBOL_RATE = 1800, HFDL_SPS = 3;                                                 // src/hfdl.h:6-7
constexpr float COSTAS_ALPHA = 0.1f, COSTAS_BETA = 0.047f * COSTAS_ALPHA * COSTAS_ALPHA;            // costas_cccf_create, :252-253
constexpr float COSTAS_ERR_LIMIT = 1.0f, COSTAS_RUNAWAY_DPHI = 0.25f;                                // :276, :709
constexpr float AGC_BANDWIDTH = 0.01f, EQ_STEP = 0.1f;                                               // agc_crcf_set_bandwidth :487, eqlms_cccf_set_bw :496
constexpr float NF_KEEP = 0.65f, NF_TAKE = 0.35f, NF_BIAS = 1e-6f;                                   // noise-floor estimator, :700-701
constexpr uint32_t NF_CLK_MASK = 0xFFu;                                                              // ... sampled when the low byte of its clock reads 0xFF, :699
constexpr uint32_t T_SEQ_BITS = 0x9AFu;                                                              // T_seq[0], :157-160: bit 14 first, 0 -> +1
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
	int32_t ss_b, ss_head;         // ss_head on the device: first input sample (relative to the next block, <= 0) whose matched-filter window entries count
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
	// the reference's debug summary (hfdl_print_summary, src/hfdl.c:563-573): A1 detections, the correlation magnitudes at the three
	// detections as sums of |2 m - 127| over match counts m (|corr| = that / 127), training bits over all frames
	uint32_t cnt_a1_found, sum_a1_dev, sum_a2_dev, sum_m1_dev, cum_train_bad, cum_train_total;
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
	const float *psk_pts;          // [16][2]: the PSK constellations by linear index (demod_tables.h)
	int32_t a1_lo, a1_hi, a2_lo, a2_hi, pos_min;      // the A1 / A2 thresholds on corr_tab as match counts (demod_tables.h)
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

// The constellation point of LINEAR index lin (modem_modulate_psk of gray_enc(lin)) comes from a 16-entry table made on the
// host (demod_tables.h psk_pts); `pts(i)` returns entry i -- two v_readlane in the carrier wave, a memory read elsewhere.
HFDL_FN int psk_entry(int arity, uint32_t lin) { return (1 << arity) - 2 + (int)lin; }

struct PskTable {                  // the table in memory (burst decoder, stage entry points, the CPU harness)
	const float *p;
	HFDL_FN cf operator()(int i) const { cf y; y.x = p[2 * i]; y.y = p[2 * i + 1]; return y; }
};

template <class Pts>
HFDL_FN uint32_t psk_slice(int arity, cf x, float *phase_error, const Pts &pts)
{
	uint32_t sym;
	cf xh;
	if (arity == 1) {
		sym = (x.x > 0) ? 0 : 1;
		xh.x = sym ? -1.0f : 1.0f; xh.y = 0.0f;
	} else {
		const uint32_t M = 1u << arity;
		const float alpha = (float)M_PI / (float)M;
		float theta = HFDL_ATAN2F(x.y, x.x);
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
		xh = pts(psk_entry(arity, s));             // = modem_modulate_psk(sym): gray_dec(gray_enc(s)) == s
	}
	if (phase_error) *phase_error = x.y * xh.x - x.x * xh.y;
	return sym;
}

HFDL_FN uint8_t soft_clamp(float v)
{
	int s = (int)(v + 127);
	return (uint8_t)(s > 255 ? 255 : (s < 0 ? 0 : s));
}

struct TableSlicer {               // psk_slice() with the table in memory, as on_symbol()'s slicer
	const float *p;
	HFDL_FN uint32_t operator()(int arity, cf x, float *phase_error) const { return psk_slice(arity, x, phase_error, PskTable{p}); }
};

// modem_demodulate_soft: 255 = confident '1', soft[0] = MSB of the symbol
template <class Pts>
HFDL_FN void psk_soft(int arity, cf x, uint8_t *soft, const Pts &pts)
{
	if (arity == 1) {
		const float llr = -2.0f * x.x * 4.0f;
		soft[0] = soft_clamp(llr * 16);
		return;
	}
	const uint32_t sym = psk_slice(arity, x, nullptr, pts);
	if (arity == 2) { soft[0] = (sym & 2) ? 255 : 0; soft[1] = (sym & 1) ? 255 : 0; return; }
	const uint32_t M = 1u << arity;
	const float gamma = 1.2f * (float)M;
	float d0[3], d1[3];
	const uint32_t lin = gray_dec(sym);
	cf c = pts(psk_entry(arity, lin));
	float er = x.x - c.x, ei = x.y - c.y;
	float d = er * er + ei * ei;
	for (int k = 0; k < arity; k++) {
		if ((sym >> (arity - k - 1)) & 1) { d0[k] = 4.0f; d1[k] = d; } else { d0[k] = d; d1[k] = 4.0f; }
	}
	for (int nb = 0; nb < 2; nb++) {
		const uint32_t nl = (lin + (nb ? 1 : M - 1)) % M;
		const uint32_t ns = gray_enc(nl);
		c = pts(psk_entry(arity, nl));
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

// crc16_ccitt(data, len, crc_init) of src/crc.c:4-47 (reflected polynomial 0x8408).  The reference walks a 256-entry table; one
// table entry is the eight shift-and-conditional-xor steps of the low octet, which for this polynomial (x^16 + x^12 + x^5 + 1)
// collapse to three shifted copies of d = low ^ (low << 4): ~8 integer operations per octet on the one lane that runs it, no
// table to stage.  Pinned against the reference's compiled crc.c by tests/golden/crc_ref.json.
HFDL_FN uint16_t crc16_ccitt_step(const uint8_t *p, uint32_t len, uint16_t crc_init)
{
	uint32_t crc = crc_init;
	for (uint32_t i = 0; i < len; i++) {
		uint32_t d = (p[i] ^ crc) & 0xFFu;
		d = (d ^ (d << 4)) & 0xFFu;
		crc = ((d << 8) | (crc >> 8)) ^ (d >> 4) ^ (d << 3);
	}
	return (uint16_t)crc;
}

// the FCS as hfdl_pdu_fcs_check computes it (src/pdu.c:69): init 0xFFFF, final XOR 0xFFFF = CRC-16/X-25
HFDL_FN uint16_t crc16_x25(const uint8_t *p, uint32_t len)
{
	return (uint16_t)(crc16_ccitt_step(p, len, 0xFFFFu) ^ 0xFFFFu);
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

// ---------------- LPDU list walk: what mpdu_parse does next (src/mpdu.c:92-158) with lpdu_parse's FCS check (src/lpdu.c:136-149) ----------------
// For an MPDU whose header FCS is good: every LPDU the header announces is located (one size octet per LPDU, size - 1) and its own
// FCS -- the same CRC, over all but its last two octets -- is checked.  The counts are the reference's StatsD events of this stage:
// lpdus.processed, lpdus.good, lpdu.errors.bad_fcs, lpdu.errors.too_short; `truncated` = an announced LPDU runs past the PDU
// (parse_lpdu_list returns -1 and the MPDU's remaining LPDUs are not looked at).
struct LpduCounts { uint8_t processed, good, bad_fcs, too_short, truncated; };

// one aircraft's / the downlink's list: size octets at `lens`, LPDUs from `data`; returns octets consumed or -1 (src/mpdu.c:136-158)
HFDL_FN int lpdu_list_walk(const uint8_t *lens, const uint8_t *data, const uint8_t *end, uint32_t lpdu_cnt, LpduCounts &c)
{
	int consumed = 0;
	for (uint32_t j = 0; j < lpdu_cnt; j++) {
		const uint32_t lpdu_len = (uint32_t)lens[j] + 1;
		if (data + lpdu_len > end) { c.truncated = 1; return -1; }
		c.processed++;
		if (lpdu_len < 3) c.too_short++;                                        // lpdu_parse: "need at least LPDU type + FCS"
		else {
			const uint16_t rx = (uint16_t)(data[lpdu_len - 2] | (data[lpdu_len - 1] << 8));
			if (rx == crc16_x25(data, lpdu_len - 2)) c.good++; else c.bad_fcs++;
		}
		data += lpdu_len;
		consumed += (int)lpdu_len;
	}
	return consumed;
}

// buf / len: the PDU; kind / hdr_len: from pdu_triage(), which must have returned 0 (header FCS good)
HFDL_FN LpduCounts lpdu_walk(const uint8_t *buf, uint32_t len, int kind, uint32_t hdr_len)
{
	LpduCounts c;
	c.processed = c.good = c.bad_fcs = c.too_short = c.truncated = 0;
	const uint8_t *end = buf + len;
	const uint8_t *data = buf + hdr_len + 2;                                    // first data octet of the first LPDU, src/mpdu.c:91
	if (kind == 1) {                                                            // downlink: size octets from octet 6
		lpdu_list_walk(buf + 6, data, end, (buf[0] >> 2) & 0xFu, c);
	} else if (kind == 2) {                                                     // uplink: per aircraft {id, NLP/DDR/P, size octets}, src/mpdu.c:104-123
		const uint32_t aircraft_cnt = ((buf[0] & 0x70u) >> 4) + 1;
		const uint8_t *hdr = buf + 2;
		for (uint32_t i = 0; i < aircraft_cnt; i++) {
			hdr++;                                                              // destination aircraft id
			const uint32_t lpdu_cnt = (*hdr++ >> 4) & 0xFu;
			const int used = lpdu_list_walk(hdr, data, end, lpdu_cnt, c);
			if (used < 0) break;
			hdr += lpdu_cnt;
			data += used;
		}
	}
	return c;
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

// `c`: where the fields only the framer's rare transitions touch live (see on_symbol)
HFDL_FN void framer_reset(ChanScalars &s, ChanScalars &c, ChanArrays &a, const float *eq_h0)      // src/hfdl.c:968-991
{
	s.fr_state = FR_A1;
	s.symbols_wanted = 1;
	c.search_retries = 0;
	s.cur_arity = 1;
	c.train_total = c.train_bad = 0;
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
	return 127 - __popcll((hi ^ thi) & 0x7FFFFFFFFFFFFFFFull) - __popcll(lo ^ tlo);
}

HFDL_HD float t_symbol(int idx)          // T = 0x9AF, bit 14 first; BPSK 0 -> +1
{
	if (idx > T_LEN - 1) idx = T_LEN - 1;
	return ((T_SEQ_BITS >> (T_LEN - 1 - idx)) & 1u) ? -1.0f : 1.0f;
}

// everything after the equaliser for one on-time symbol: src/hfdl.c:737-891
// `s` / `c`: the channel's scalars in two places.  The carrier wave keeps `s` in registers; the dozen fields that only the framer's rare
// transitions read or write (search_retries, the preamble / frame counters, M1, the training tallies, ...) are reached through `c`,
// which there is the copy in LDS -- the wave is short of scalar registers, and every live value it does not need in its loop is a
// spill inside it.  Everywhere else `c` is `s` itself.
// `slice(arity, x, &phase_error)` = modem_demodulate + the demodulator phase error (psk_slice() above, or the carrier wave's
// lane-parallel form of the same decision)
template <class Slicer>
HFDL_FN void on_symbol(ChanScalars &s, ChanScalars &c, ChanArrays &a, const DemodConst &T, const BlockIo &io, cf sym, float level, const Slicer &slice)
{
	float perr;
	uint32_t bits = slice(s.cur_arity, sym, &perr);
	{   // costas_cccf_adjust, :276-281
		const float e = 0.5f * (fabsf(perr + COSTAS_ERR_LIMIT) - fabsf(perr - COSTAS_ERR_LIMIT));
		s.err = e;
		s.phi += COSTAS_ALPHA * e;
		s.dphi += COSTAS_BETA * e;
	}
	s.symbol_cnt++;
	if (s.symbol_cnt >= (uint64_t)(NO_FRAME_TIMEOUT_FRAMES * SINGLE_SLOT_FRAME_LEN) && s.fr_state == FR_A1) {
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
				if ((threadIdx.x & 63u) == 0) io.data[s.data_slot * MAX_DATA_SYMBOLS + s.data_n] = sym;
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
		// |corr| > 0.36 on the fp32 table == a match count outside (a1_lo, a1_hi): no table look-up on the every-symbol path
		const int m = bits_correlate(s.bits_hi, s.bits_lo, T.a_hi, T.a_lo);
		if (m <= T.a1_lo || m >= T.a1_hi) {
			c.cnt_a1_found++;                    // S.A1_found / S.A1_corr_total, :786-787
			c.sum_a1_dev += (uint32_t)(2 * m > 127 ? 2 * m - 127 : 127 - 2 * m);
			s.bitmask = m >= T.pos_min ? 0u : ~0u;
			s.signal_level = level;
			s.frame_symbol_cnt = 1.0f;
			s.symbols_wanted = A_LEN;
			c.search_retries = 0;
			s.fr_state = FR_A2;
		}
		break; }
	case FR_A2: {
		const int m = bits_correlate(s.bits_hi, s.bits_lo, T.a_hi, T.a_lo);
		if (m <= T.a2_lo || m >= T.a2_hi) {
			c.cnt_a2_found++;                    // statsd "demod.preamble.A2_found"
			c.sum_a2_dev += (uint32_t)(2 * m > 127 ? 2 * m - 127 : 127 - 2 * m);
			c.pdu_sample_index = s.sample_cnt;
			c.freq_err_hz = (float)((double)(s.dphi * HFDL_SYMBOL_RATE) / (2.0 * M_PI));
			s.symbols_wanted = M1_LEN;
			c.search_retries = 0;
			s.fr_state = FR_M1;
		} else if (++c.search_retries >= MAX_SEARCH_RETRIES) {
			framer_reset(s, c, a, T.eq_h0);
		}
		break; }
	case FR_M1: {
		float best = 0.f;
		int best_idx = -1, best_cnt = 0;
		for (int m = 0; m < 8; m++) {
			const int cnt = bits_correlate(s.bits_hi, s.bits_lo, T.m1_hi[m], T.m1_lo[m]);
			const float corr = fabsf(T.corr_tab[cnt]);
			if (corr > best) { best = corr; best_idx = m; best_cnt = cnt; }
		}
		if (fabsf(best) > CORR_THRESHOLD_M1) {
			const ModeParams mp = mode_params(best_idx);
			c.cnt_m1_found++;                    // "demod.preamble.M1_found"
			c.sum_m1_dev += (uint32_t)(2 * best_cnt > 127 ? 2 * best_cnt - 127 : 127 - 2 * best_cnt);
			c.data_segment_cnt = mp.segments;
			c.data_arity = mp.arity;
			c.M1 = best_idx;
			s.symbols_wanted = M2_LEN;
			c.search_retries = 0;
			s.fr_state = FR_M2_SKIP;
			s.s_state = SAMPLER_SKIP;
		} else {
			c.cnt_m1_not_found++;                // "demod.preamble.errors.M1_not_found"
			framer_reset(s, c, a, T.eq_h0);
		}
		break; }
	case FR_M2_SKIP:
		s.training_n = 0;
		s.symbols_wanted = T_LEN;
		c.eq_train_seq_cnt = PREAMBLE_T_SEQS;
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
		const int train_err = __popc(T_SEQ_BITS ^ seq);
		c.train_total += T_LEN;
		c.train_bad += train_err;
		c.cum_train_total += T_LEN;              // S.train_bits_total / S.train_bits_bad, :962-963
		c.cum_train_bad += (uint32_t)train_err;
		s.training_n = 0;
		if (c.eq_train_seq_cnt > 1) {
			c.eq_train_seq_cnt--;
			s.symbols_wanted = T_LEN;
			s.T_idx = 0;
		} else if (c.data_segment_cnt > 0) {
			s.symbols_wanted = DATA_FRAME_LEN / 2;
			s.fr_state = FR_DATA_1;
			s.cur_arity = c.data_arity;
			s.use_data = 1;
		} else {
			// end of frame: queue it for the burst decoder (decode_user_data + dispatch_pdu, :993-1080)
			if ((threadIdx.x & 63u) == 0) {
				const int slot = atomicAdd(io.frame_count, 1);
				if (slot < io.frame_cap) {
					FrameRec fr;
					fr.channel = io.channel; fr.slot = s.data_slot; fr.mode = c.M1; fr.bitmask_lsb = (int32_t)(s.bitmask & 1u);
					fr.freq_err_hz = c.freq_err_hz; fr.signal_level = s.signal_level; fr.noise_floor = s.noise_floor;
					fr.train_bad = c.train_bad; fr.train_total = c.train_total; fr.pad = 0;
					fr.sample_index = c.pdu_sample_index;
					io.frames[slot] = fr;
				}
			}
			s.data_slot ^= 1;
			c.cnt_frames++;
			framer_reset(s, c, a, T.eq_h0);
			s.symbol_cnt = 0;
		}
		break; }
	case FR_DATA_1:
		s.symbols_wanted = DATA_FRAME_LEN / 2;
		s.fr_state = FR_DATA_2;
		break;
	case FR_DATA_2:
		c.data_segment_cnt--;
		s.cur_arity = 1;
		s.use_data = 0;
		s.fr_state = FR_EQ_TRAIN;
		c.eq_train_seq_cnt = 1;
		s.symbols_wanted = T_LEN;
		s.T_idx = 0;
		break;
	}
}

// The constants above and the tables this file computes from them (mode table, training sequence), gathered for a read-back: the same
// function runs on the host (tests/hostsim) and on the device (constants_kernel, laboratory entry point hfdl_gpu_lab_read_constants),
// and both are compared with the reference's text (tests/golden/hfdl_constants.json).
struct HfdlConstants {
	int32_t prekey_len, a_len, m1_len, m2_len, t_len, data_frame_len, preamble_t_seqs, single_slot_frame_len, max_data_symbols;
	int32_t max_search_retries, no_frame_timeout_frames, symbol_rate, sps, eq_len, mf_taps, ss_npfb, nf_clk_mask;
	int32_t sampler_states[3], framer_states[7];
	int32_t modes[8][4];                   // mode_params(m): bits per symbol, data segments, code rate, interleaver column shift
	float corr_a1, corr_a2, corr_m1, costas_alpha, costas_beta, costas_err_limit, costas_runaway_dphi, agc_bandwidth, eq_step;
	float nf_keep, nf_take, nf_bias;
	float t_seq[15];                       // t_symbol(i): T_seq[0]
};
HFDL_HD void hfdl_constants(HfdlConstants &k)
{
	k.prekey_len = PREKEY_LEN; k.a_len = A_LEN; k.m1_len = M1_LEN; k.m2_len = M2_LEN; k.t_len = T_LEN; k.data_frame_len = DATA_FRAME_LEN;
	k.preamble_t_seqs = PREAMBLE_T_SEQS; k.single_slot_frame_len = SINGLE_SLOT_FRAME_LEN; k.max_data_symbols = MAX_DATA_SYMBOLS;
	k.max_search_retries = MAX_SEARCH_RETRIES; k.no_frame_timeout_frames = NO_FRAME_TIMEOUT_FRAMES; k.symbol_rate = HFDL_SYMBOL_RATE; k.sps = HFDL_SPS;
	k.eq_len = D_EQ; k.mf_taps = D_MF; k.ss_npfb = D_SS_NPFB; k.nf_clk_mask = (int32_t)NF_CLK_MASK;
	k.sampler_states[0] = SAMPLER_BITS; k.sampler_states[1] = SAMPLER_SYMBOLS; k.sampler_states[2] = SAMPLER_SKIP;
	k.framer_states[0] = FR_A1; k.framer_states[1] = FR_A2; k.framer_states[2] = FR_M1; k.framer_states[3] = FR_M2_SKIP;
	k.framer_states[4] = FR_EQ_TRAIN; k.framer_states[5] = FR_DATA_1; k.framer_states[6] = FR_DATA_2;
	for (int m = 0; m < 8; m++) {
		const ModeParams p = mode_params(m);
		k.modes[m][0] = p.arity; k.modes[m][1] = p.segments; k.modes[m][2] = p.code_rate; k.modes[m][3] = p.col_shift;
	}
	k.corr_a1 = CORR_THRESHOLD_A1; k.corr_a2 = CORR_THRESHOLD_A2; k.corr_m1 = CORR_THRESHOLD_M1;
	k.costas_alpha = COSTAS_ALPHA; k.costas_beta = COSTAS_BETA; k.costas_err_limit = COSTAS_ERR_LIMIT; k.costas_runaway_dphi = COSTAS_RUNAWAY_DPHI;
	k.agc_bandwidth = AGC_BANDWIDTH; k.eq_step = EQ_STEP;
	k.nf_keep = NF_KEEP; k.nf_take = NF_TAKE; k.nf_bias = NF_BIAS;
	for (int i = 0; i < 15; i++) k.t_seq[i] = t_symbol(i);
}

}  // namespace hfdl
