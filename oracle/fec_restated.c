/*
 * fec_restated.c -- ORACLE (test infrastructure only; see hfdl_oracle.h).
 *
 * Integer / bit-exact pieces of the HFDL burst decoder restated in plain C:
 * K=7 r=1/2 soft Viterbi (libfec port as dumphfdl calls it), CRC-16, the 40-row
 * de-interleaver, the 15-stage descrambler, PSK soft de-mapping (liquid-dsp modem
 * semantics, unpinned) and decode_user_data().
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "hfdl_oracle.h"

extern orc_variant orc_v;       /* channel_restated.c */
#include "../tests/hostsim/shared_math.h"

static inline int parity32(uint32_t x)
{
	x ^= x >> 16; x ^= x >> 8; x ^= x >> 4; x ^= x >> 2; x ^= x >> 1;
	return (int)(x & 1u);
}

/* ---- convolutional code, polynomials 0x6d / 0x4f (src/libfec/fec.h:13-14) ---- */

void orc_conv27_encode(const uint8_t *bits, int32_t nbits, uint8_t *coded)
{
	uint32_t sr = 0;
	for (int32_t i = 0; i < nbits; i++) {
		sr = (sr << 1) | (bits[i] & 1u);
		coded[2 * i] = (uint8_t)parity32(sr & 0x6d);
		coded[2 * i + 1] = (uint8_t)parity32(sr & 0x4f);
	}
}

/* ---- Viterbi, following src/libfec/viterbi27_port.c as driven by src/hfdl.c:1045-1047 ----
 *  - metrics are uint32, start at 63 except state 0 = 0            (viterbi27_port.c:65-79)
 *  - branch metric (tab0^sym0)+(tab1^sym1), 510-metric for the complementary branch,
 *    decision = (int)(m0-m1) > 0 picks m1                            (:147-160)
 *  - one 64-bit decision word per decoded bit, bit s = decision of new state s
 *  - traceback starts in state 0 and reads the decision word 6 steps AHEAD of the bit it
 *    emits (:122 "d += 6"); dumphfdl runs update for nbits steps only, so the last 6 words
 *    it reads are the never-written (zero) tail of the calloc'ed array (:97) */
void orc_viterbi27_decode(const uint8_t *soft, int32_t nbits, uint8_t *out)
{
	uint8_t tab0[32], tab1[32];
	for (uint32_t st = 0; st < 32; st++) {
		tab0[st] = parity32((2 * st) & 0x6d) ? 255 : 0;
		tab1[st] = parity32((2 * st) & 0x4f) ? 255 : 0;
	}
	uint64_t *dec = calloc((size_t)nbits + 6, sizeof(uint64_t));
	uint32_t ma[64], mb[64], *cur = ma, *nxt = mb;
	for (int i = 0; i < 64; i++) cur[i] = 63;
	cur[0] = 0;
	for (int32_t t = 0; t < nbits; t++) {
		uint32_t s0 = soft[2 * t], s1 = soft[2 * t + 1];
		uint64_t word = 0;
		for (uint32_t i = 0; i < 32; i++) {
			uint32_t bm = (tab0[i] ^ s0) + (tab1[i] ^ s1);
			uint32_t m0 = cur[i] + bm, m1 = cur[i + 32] + (510 - bm);
			uint32_t pick = (int32_t)(m0 - m1) > 0;
			nxt[2 * i] = pick ? m1 : m0;
			word |= (uint64_t)pick << (2 * i);
			m0 -= (bm + bm - 510);
			m1 += (bm + bm - 510);
			pick = (int32_t)(m0 - m1) > 0;
			nxt[2 * i + 1] = pick ? m1 : m0;
			word |= (uint64_t)pick << (2 * i + 1);
		}
		dec[t] = word;
		uint32_t *sw = cur; cur = nxt; nxt = sw;
	}
	uint32_t reg = 0;            /* encoder state in bits 7..2 */
	memset(out, 0, (size_t)((nbits + 7) / 8));
	for (int32_t idx = nbits - 1; idx >= 0; idx--) {
		uint32_t k = (uint32_t)(dec[idx + 6] >> (reg >> 2)) & 1u;
		reg = (reg >> 1) | (k << 7);
		out[idx >> 3] = (uint8_t)reg;
	}
	free(dec);
}

/* ---- CRC-16 (src/crc.c:4-47): reflected 0x1021 (=0x8408), caller supplies init ---- */

uint16_t orc_crc16_ccitt(const uint8_t *data, uint32_t len, uint16_t init)
{
	uint16_t crc = init;
	for (uint32_t i = 0; i < len; i++) {
		crc ^= data[i];
		for (int b = 0; b < 8; b++) crc = (crc & 1) ? (uint16_t)((crc >> 1) ^ 0x8408) : (uint16_t)(crc >> 1);
	}
	return crc;
}

int orc_fcs_check(const uint8_t *buf, uint32_t hdr_len)
{
	uint16_t rx = (uint16_t)(buf[hdr_len] | (buf[hdr_len + 1] << 8));
	uint16_t calc = orc_crc16_ccitt(buf, hdr_len, 0xFFFFu) ^ 0xFFFFu;
	return rx == calc;
}

int orc_pdu_triage(const uint8_t *buf, uint32_t len, int *kind, uint32_t *hdr_len)
{
	if (!(buf[0] & 1)) {                                   /* IS_MPDU(buf) is bit 0, src/pdu.c:102 */
		*kind = 0;
		*hdr_len = 64;                                     /* SPDU_LEN 66 = 64 + FCS, src/spdu.c:12 */
		if (len < 66) return 2;
		return orc_fcs_check(buf, 64) ? 0 : 1;
	}
	uint32_t hl;
	if (buf[0] & 0x2) {                                    /* downlink, src/mpdu.c:62-65 */
		*kind = 1;
		hl = 6 + ((buf[0] >> 2) & 0xF);
	} else {                                               /* uplink, :66-79 */
		*kind = 2;
		uint32_t ac = ((buf[0] & 0x70) >> 4) + 1;
		hl = 2;
		for (uint32_t i = 0; i < ac; i++) {
			if (len < hl + 2) { *hdr_len = hl; return 2; }
			hl += 2 + (buf[hl + 1] >> 4);
		}
	}
	*hdr_len = hl;
	if (len < hl + 2) return 2;
	return orc_fcs_check(buf, hl) ? 0 : 1;
}

/* ---- LPDU list walk: parse_lpdu_list (src/mpdu.c:136-158) + lpdu_parse's length / FCS checks (src/lpdu.c:136-149) ----
 * counts[0..4] = lpdus.processed, lpdus.good, lpdu.errors.bad_fcs, lpdu.errors.too_short, truncated flag. */
static int lpdu_list_counts(const uint8_t *lpdu_len_ptr, const uint8_t *data_ptr, const uint8_t *endptr, uint32_t lpdu_cnt, uint8_t *counts)
{
	int consumed_octets = 0;
	for (uint32_t j = 0; j < lpdu_cnt; j++) {
		uint32_t lpdu_len = (uint32_t)*lpdu_len_ptr + 1;                       /* src/mpdu.c:142 */
		if (data_ptr + lpdu_len <= endptr) {                                   /* :143 */
			counts[0]++;                                                       /* lpdu_parse: "lpdus.processed", src/lpdu.c:127 */
			if (lpdu_len < 3) counts[3]++;                                     /* :136-140 */
			else if (orc_fcs_check(data_ptr, lpdu_len - 2)) counts[1]++;       /* :143-150 */
			else counts[2]++;
			data_ptr += lpdu_len;
			consumed_octets += (int)lpdu_len;
			lpdu_len_ptr++;
		} else {
			counts[4] = 1;                                                     /* :152-155: return -1 */
			return -1;
		}
	}
	return consumed_octets;
}

void orc_lpdu_walk(const uint8_t *buf, uint32_t len, uint8_t *counts)
{
	int kind = 0;
	uint32_t hdr_len = 0;
	for (int i = 0; i < 5; i++) counts[i] = 0;
	if (orc_pdu_triage(buf, len, &kind, &hdr_len) != 0 || kind == 0) return;  /* bad header FCS: "goto end" before any LPDU, src/mpdu.c:83-89 */
	const uint8_t *dataptr = buf + hdr_len + 2;                                /* src/mpdu.c:91 */
	if (kind == 1) {
		lpdu_list_counts(buf + 6, dataptr, buf + len, (buf[0] >> 2) & 0xF, counts);        /* :96-100 */
	} else {
		uint32_t aircraft_cnt = ((buf[0] & 0x70) >> 4) + 1;
		const uint8_t *hdrptr = buf + 2;                                       /* :106 */
		uint32_t lpdu_cnt = 0;
		int consumed = 0;
		for (uint32_t i = 0; i < aircraft_cnt; i++, hdrptr += lpdu_cnt, dataptr += consumed) {   /* :108 */
			hdrptr++;                                                          /* dst_id */
			lpdu_cnt = (*hdrptr++ >> 4) & 0xF;
			if ((consumed = lpdu_list_counts(hdrptr, dataptr, buf + len, lpdu_cnt, counts)) < 0) return;
		}
	}
}

uint8_t orc_reverse_byte(uint8_t x)
{
	x = (uint8_t)((x >> 4) | (x << 4));
	x = (uint8_t)(((x & 0xCC) >> 2) | ((x & 0x33) << 2));
	x = (uint8_t)(((x & 0xAA) >> 1) | ((x & 0x55) << 1));
	return x;
}

/* ---- frame parameters (src/hfdl.c:81-138) ---- */

const orc_mode_params orc_modes[ORC_MODE_CNT] = {
	{ 1, 72, 4, 17 }, { 1, 72, 2, 17 }, { 2, 72, 2, 17 }, { 3, 72, 2, 17 },
	{ 1, 168, 4, 23 }, { 1, 168, 2, 23 }, { 2, 168, 2, 23 }, { 3, 168, 2, 23 },
};

int32_t orc_mode_num_symbols(int mode) { return orc_modes[mode].segments * 30; }
int32_t orc_mode_coded_bits(int mode) { return orc_mode_num_symbols(mode) * orc_modes[mode].arity; }
int32_t orc_mode_viterbi_bits(int mode)
{
	int32_t vin = orc_mode_coded_bits(mode);
	if (orc_modes[mode].code_rate == 4) vin /= 2;
	return vin / 2;
}
int32_t orc_mode_pdu_octets(int mode) { return (orc_mode_viterbi_bits(mode) + 7) / 8; }

/* preamble sequences: protocol constants (src/hfdl.c:420-459) */
static const uint8_t A_OCTETS[16] = {
	0x5B, 0xBC, 0x74, 0x57, 0x03, 0xD9, 0x89, 0x39, 0xF2, 0x08, 0xD5, 0x36, 0x94, 0x2C, 0x32, 0xFE
};
static const char M1_BASE[128] =
	"01110110111101000101100" "10111110001000000110011011" "00011100111010111000010011"
	"00000101010110100100101001" "11100100011010100001111111";
static const int M1_SHIFT[ORC_MODE_CNT] = { 72, 82, 113, 123, 61, 103, 93, 9 };

void orc_preamble_A(uint8_t bits[127])
{
	for (int i = 0; i < 127; i++) bits[i] = (A_OCTETS[i / 8] >> (7 - (i % 8))) & 1;
}

void orc_preamble_M1(int mode, uint8_t bits[127])
{
	for (int j = 0; j < 127; j++) bits[j] = (uint8_t)(M1_BASE[(M1_SHIFT[mode] + j) % 127] - '0');
}

void orc_training_T(uint8_t bits[15])
{
	const uint32_t T = 0x9AF;       /* src/hfdl.c:181; MSB (bit 14) is sent first */
	for (int i = 0; i < 15; i++) bits[i] = (T >> (14 - i)) & 1;
}

/* descrambler: 15-stage LFSR, liquid msequence semantics (UNPINNED), restart every 120 symbols
 * (src/hfdl.c:300-347): state 0x4d4b, taps 0x4001, b = parity(state & taps), state = (state<<1 | b) */
void orc_scrambler_bits(uint8_t *bits, int32_t n)
{
	if (orc_v.lfsr_kind == 1) {
		/* the pre-1.6 msequence API as src/hfdl.c:331-333 feeds it (genpoly 0x8002, init 0x6959), restated literally:
		 * msequence_create keeps g = genpoly >> 1 and a = the m-bit reversal of init; advance: b = parity(v & g), v = (v << 1 | b) & (2^m - 1) */
		const uint32_t m = 15, g = 0x8002u >> 1;
		uint32_t a = 0, init = 0x6959u;
		for (uint32_t i = 0; i < m; i++) { a = (a << 1) | (init & 1u); init >>= 1; }
		uint32_t v = a;
		for (int32_t i = 0; i < n; i++) {
			if (i % 120 == 0) v = a;
			uint32_t b = (uint32_t)parity32(v & g);
			v = ((v << 1) | b) & ((1u << m) - 1);
			bits[i] = (uint8_t)b;
		}
		return;
	}
	if (orc_v.lfsr_kind == 2) {
		/* the other way to read "msequence_create(15, 0x4001, 0x4d4b)": a register shifting RIGHT, feedback into the top bit */
		uint32_t v = 0x4d4b;
		for (int32_t i = 0; i < n; i++) {
			if (i % 120 == 0) v = 0x4d4b;
			uint32_t b = (uint32_t)parity32(v & 0x4001);
			v = (v >> 1) | (b << 14);
			bits[i] = (uint8_t)b;
		}
		return;
	}
	uint32_t v = 0x4d4b;
	for (int32_t i = 0; i < n; i++) {
		if (i % 120 == 0) v = 0x4d4b;
		uint32_t b = (uint32_t)parity32(v & 0x4001);
		v = ((v << 1) | b) & 0x7fff;
		bits[i] = (uint8_t)b;
	}
}

/* de-interleaver (src/hfdl.c:353-413): 40 rows x C columns; position = row*C + col */
void orc_deinterleave_maps(int mode, int32_t *push_pos, int32_t *pop_pos)
{
	int32_t total = orc_mode_coded_bits(mode), cols = total / 40, shift = orc_modes[mode].col_shift;
	int32_t row = 0, col = 0;
	for (int32_t k = 0; k < total; k++) {
		push_pos[k] = row * cols + col;
		if (++row == 40) { row = 0; col++; }
		col -= shift;
		if (col < 0) col += cols;
	}
	/* after a whole frame the push cursor is back at (0,0): pops start there */
	row = 0; col = 0;
	for (int32_t k = 0; k < total; k++) {
		pop_pos[k] = row * cols + col;
		row = (row + 9) % 40;
		if (row == 0) col++;
	}
}

/* ---- liquid-dsp modem semantics (a16), UNPINNED restatement ---- */

static inline uint32_t gray_enc(uint32_t b) { return b ^ (b >> 1); }
static inline uint32_t gray_dec(uint32_t g) { uint32_t b = g; while (g >>= 1) b ^= g; return b; }

orc_cf orc_modem_modulate(int arity, uint32_t sym)
{
	orc_cf y;
	if (arity == 1) { y.re = sym ? -1.0f : 1.0f; y.im = 0.0f; return y; }
	uint32_t M = 1u << arity;
	float alpha = (float)M_PI / (float)M;
	float ang = (float)gray_dec(sym) * 2 * alpha;
	y.re = cosf(ang); y.im = sinf(ang);
	return y;
}

/* hard decision + phase error Im{r conj(x_hat)} */
uint32_t orc_modem_demod_hard(int arity, orc_cf x, float *phase_error)
{
	uint32_t sym;
	orc_cf xh;
	if (arity == 1) {
		sym = (x.re > 0) ? 0 : 1;
		xh.re = sym ? -1.0f : 1.0f; xh.im = 0.0f;
	} else {
		uint32_t M = 1u << arity;
		float alpha = (float)M_PI / (float)M;
		float theta = orc_v.shared_math ? sm_atan2f(x.im, x.re) : atan2f(x.im, x.re);
		theta -= (float)M_PI * (1.0f - 1.0f / (float)M);
		if (theta < -(float)M_PI) theta += 2 * (float)M_PI;
		/* successive-approximation slicer on ref[k] = 2^k * alpha */
		uint32_t s = 0;
		float v = theta;
		for (int k = arity - 1; k >= 0; k--) {
			float ref = (float)(1u << k) * alpha;
			s <<= 1;
			if (v > 0) { s |= 1; v -= ref; } else { v += ref; }
		}
		sym = gray_enc(s);
		xh = orc_modem_modulate(arity, sym);
	}
	if (phase_error) {
		*phase_error = x.im * xh.re - x.re * xh.im;
		if (orc_v.perr_kind == 1) *phase_error = atan2f(x.im * xh.re - x.re * xh.im, x.re * xh.re + x.im * xh.im);
	}
	return sym;
}

static inline uint8_t clamp_soft(float llr16)
{
	int v = orc_v.soft_floor ? (int)floorf(llr16 + 127) : (int)(llr16 + 127);
	if (v > 255) v = 255;
	if (v < 0) v = 0;
	return (uint8_t)v;
}

void orc_modem_demod_soft(int arity, orc_cf x, uint8_t *soft)
{
	if (arity == 1) {
		/* gamma = 4: LLR = -2*re*gamma, soft = LLR*16 + 127 */
		float llr = -2.0f * x.re * 4.0f;
		soft[0] = clamp_soft(llr * 16);
		return;
	}
	uint32_t sym = orc_modem_demod_hard(arity, x, NULL);
	if (arity == 2) {
		/* no neighbour table for m<3: hard bits expanded to 0/255, MSB first */
		soft[0] = (sym & 2) ? 255 : 0;
		soft[1] = (sym & 1) ? 255 : 0;
		return;
	}
	/* m>=3: nearest-neighbour approximation with p=2 neighbours, gamma = 1.2*M */
	const uint32_t M = 1u << arity;
	const float gamma = (orc_v.soft_gamma_scale > 0.f ? orc_v.soft_gamma_scale : 1.0f) * 1.2f * (float)M;
	float d0[3], d1[3];
	orc_cf xh = orc_modem_modulate(arity, sym);
	float er = x.re - xh.re, ei = x.im - xh.im;
	float d = er * er + ei * ei;
	for (int k = 0; k < arity; k++) {
		const float far = orc_v.soft_dmin_init;
		if ((sym >> (arity - k - 1)) & 1) { d0[k] = far; d1[k] = d; } else { d0[k] = d; d1[k] = far; }
	}
	uint32_t lin = gray_dec(sym);
	for (int nb = 0; nb < 2; nb++) {
		uint32_t ns = gray_enc((lin + (nb ? 1 : M - 1)) % M);
		orc_cf c = orc_modem_modulate(arity, ns);
		er = x.re - c.re; ei = x.im - c.im;
		d = er * er + ei * ei;
		for (int k = 0; k < arity; k++) {
			if ((ns >> (arity - k - 1)) & 1) { if (d < d1[k]) d1[k] = d; } else { if (d < d0[k]) d0[k] = d; }
		}
	}
	for (int k = 0; k < arity; k++) soft[k] = clamp_soft(((d0[k] - d1[k]) * gamma) * 16);
}

/* ---- decode_user_data (src/hfdl.c:993-1056) ---- */

int32_t orc_decode_user_data(int mode, const orc_cf *symbols, int bitmask_lsb, uint8_t *octets)
{
	const orc_mode_params *p = &orc_modes[mode];
	int32_t nsym = orc_mode_num_symbols(mode), ncoded = orc_mode_coded_bits(mode);
	uint8_t *scr = malloc((size_t)nsym), *table = malloc((size_t)ncoded);
	int32_t *push_pos = malloc(sizeof(int32_t) * (size_t)ncoded), *pop_pos = malloc(sizeof(int32_t) * (size_t)ncoded);
	orc_scrambler_bits(scr, nsym);
	orc_deinterleave_maps(mode, push_pos, pop_pos);
	int32_t k = 0;
	for (int32_t i = 0; i < nsym; i++) {
		float flip = (scr[i] ? -1.0f : 1.0f) * (bitmask_lsb ? -1.0f : 1.0f);
		orc_cf x = { symbols[i].re * flip, symbols[i].im * flip };
		uint8_t soft[3];
		orc_modem_demod_soft(p->arity, x, soft);
		for (int j = 0; j < p->arity; j++) table[push_pos[k++]] = soft[j];
	}
	int32_t vin_len = (p->code_rate == 4) ? ncoded / 2 : ncoded;
	uint8_t *vin = malloc((size_t)vin_len);
	if (p->code_rate == 4) {
		for (int32_t i = 0; i < vin_len; i++) {
			uint8_t a = table[pop_pos[2 * i]], b = table[pop_pos[2 * i + 1]];
			vin[i] = (uint8_t)((a & b) + ((a ^ b) >> 1));
		}
	} else {
		for (int32_t i = 0; i < vin_len; i++) vin[i] = table[pop_pos[i]];
	}
	int32_t nbits = vin_len / 2, noct = (nbits + 7) / 8;
	orc_viterbi27_decode(vin, nbits, octets);
	for (int32_t i = 0; i < noct; i++) octets[i] = orc_reverse_byte(octets[i]);
	free(scr); free(table); free(push_pos); free(pop_pos); free(vin);
	return noct;
}
