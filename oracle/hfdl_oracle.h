/*
 * hfdl_oracle.h -- CPU restatement of dumphfdl's channelizer + HFDL demod/FEC hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's parity / cpu_baseline legs (the checker and the
 * reported CPU baseline, never the thing measured) may load or call anything under oracle/.  The shipped path (dumphfdl_amd/) never links or dlopens it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - Viterbi K=7 r=1/2, CRC-16, NCO/decimator: PINNED against the reference's own C files
 *     compiled unmodified into oracle/_ref/libhfdl_ref.so (oracle/Makefile).
 *   - fastddc geometry / tap design / fold / framer / deinterleaver / descrambler: restated from
 *     the reference source (file:line cited per function); the reference holds no tests or golden
 *     vectors and these files need generated/external headers -> "parity unpinned" against a
 *     reference binary; validated against float64 numpy direct-form math and round trips.
 *   - liquid-dsp objects (msresamp/agc/firfilt/symsync/eqlms/modem/bsequence/msequence): the
 *     library (pin >=1.3.0,<2.0.0, src/CMakeLists.txt:71-73) is absent from /root/reference and
 *     from this image; its published algorithms are restated -> PARITY UNPINNED.
 */
#ifndef HFDL_ORACLE_H
#define HFDL_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cf;

/* ---------------- libcsdr / fastddc restatement ---------------- */

/* reference: struct fastddc_s, src/fastddc.h:8-27 */
typedef struct {
	int32_t pre_decimation, post_decimation;
	int32_t taps_length, taps_min_length, overlap_length;
	int32_t fft_size, fft_inv_size, input_size, post_input_size;
	int32_t startbin, v, offsetbin, scrap;
	float pre_shift, post_shift;
	float nco_sindelta, nco_cosdelta, nco_rate;   /* shift_addition_data_t, src/libcsdr_gpl.h:26-31 */
} orc_ddc;

/* reference: decimating_shift_addition_status_t, src/libcsdr_gpl.h:35-40 */
typedef struct {
	int32_t decimation_remain;
	float starting_phase;
	int32_t output_size;
} orc_nco_state;

int32_t orc_next_pow2(int32_t x);                                   /* src/libcsdr.c:35-44 */
int32_t orc_firdes_filter_len(float transition_bw);                 /* src/libcsdr.c:46-51 */
int32_t orc_compute_fft_decimation_rate(int32_t fs, int32_t target);/* src/libcsdr.c:140-144 */
float   orc_transition_bw(int32_t fs, int32_t bw_hz);               /* src/libcsdr.c:135-138 */
int     orc_fastddc_init(orc_ddc *d, float transition_bw, int32_t decimation, float shift_rate); /* src/fastddc.c:46-80 */
void    orc_firdes_lowpass_f(float *out, int32_t length, float cutoff);        /* src/libcsdr.c:94-108 (Hamming) */
void    orc_firdes_bandpass_c(orc_cf *out, int32_t length, float lowcut, float highcut); /* src/libcsdr.c:110-133 */
void    orc_fft_swap_sides(orc_cf *io, int32_t n);                  /* src/fastddc.c:102-112 */
/* forward (sign=-1) / backward (sign=+1) unnormalised DFT, out of place; src/fft_fftw.c:22-41 contract */
void    orc_fft_f32(const orc_cf *in, orc_cf *out, int32_t n, int sign);
/* the forward transform of orc_forward_block on `threads` pthreads (six-step split): cpu_baseline timing only, see csdr_restated.c */
void    orc_fft_f32_mt(const orc_cf *in, orc_cf *out, int32_t n, int sign, int threads);
void    orc_set_fft_threads(int n);
int     orc_get_fft_threads(void);
void    orc_fft_f64(const double *in_ri, double *out_ri, int32_t n, int sign);
/* channel taps in the frequency domain, fftshifted: src/fastddc.c:217-252. f64_fft!=0 -> double FFT */
int     orc_channelizer_taps(const orc_ddc *d, int32_t decimation, float freq_shift, orc_cf *taps_fft, int f64_fft);
/* spectrum*taps fold: src/fastddc.c:123-150 */
void    orc_fold(const orc_cf *spectrum, const orc_cf *taps, int32_t n, orc_cf *out, int32_t m, int32_t offsetbin);
/* NCO + decimate with carried state: src/libcsdr_gpl.c:41-74 */
orc_nco_state orc_shift_decimate(const orc_cf *in, orc_cf *out, int32_t n, const orc_ddc *d, orc_nco_state s);
/* whole per-channel inverse step: src/fastddc.c:152-215. scratch: 2*M orc_cf */
orc_nco_state orc_fastddc_inv(const orc_cf *spectrum, orc_cf *out, const orc_ddc *d, const orc_cf *taps_fft,
		orc_nco_state s, orc_cf *scratch);
/* overlap assembly + forward FFT + swap: src/fft.c:49-59. buf holds N samples of history. */
void    orc_forward_block(orc_cf *buf, const orc_cf *new_samples, const orc_ddc *d, orc_cf *spectrum);

/* ---------------- libfec / crc restatement ---------------- */
/* src/libfec/viterbi27_port.c:65-79,105-135,166-221 as used by src/hfdl.c:1045-1047.
 * soft: 2*nbits bytes (0=strong 0, 255=strong 1); out: ceil(nbits/8) bytes, MSB-first, not yet bit-reversed */
void     orc_viterbi27_decode(const uint8_t *soft, int32_t nbits, uint8_t *out);
void     orc_conv27_encode(const uint8_t *bits, int32_t nbits, uint8_t *coded /* 2*nbits, values 0/1 */);
uint16_t orc_crc16_ccitt(const uint8_t *data, uint32_t len, uint16_t init);   /* src/crc.c:4-47 */
int      orc_fcs_check(const uint8_t *buf, uint32_t hdr_len);                  /* src/pdu.c:68-79 */
/* header triage as mpdu_parse / spdu_parse begin: src/mpdu.c:56-89, src/spdu.c:12,55-62, src/pdu.c:124-128.
 * returns 0 good FCS, 1 bad FCS, 2 too short; kind 0 SPDU, 1 MPDU downlink, 2 MPDU uplink */
int      orc_pdu_triage(const uint8_t *buf, uint32_t len, int *kind, uint32_t *hdr_len);
/* parse_lpdu_list + lpdu_parse's checks (src/mpdu.c:92-158, src/lpdu.c:127-150): counts[5] = processed, good, bad FCS, too short, truncated */
void     orc_lpdu_walk(const uint8_t *buf, uint32_t len, uint8_t *counts);
uint8_t  orc_reverse_byte(uint8_t x);                                          /* src/util.h:109 */

/* ---------------- HFDL frame constants (src/hfdl.c:29-46,81-138) ---------------- */
#define ORC_MODE_CNT 8
typedef struct { int32_t arity, segments, code_rate, col_shift; } orc_mode_params;
extern const orc_mode_params orc_modes[ORC_MODE_CNT];
int32_t orc_mode_num_symbols(int mode);       /* segments*30 */
int32_t orc_mode_coded_bits(int mode);        /* num_symbols*arity */
int32_t orc_mode_viterbi_bits(int mode);      /* decoded bits */
int32_t orc_mode_pdu_octets(int mode);
void    orc_preamble_A(uint8_t bits[127]);                 /* src/hfdl.c:419-438 */
void    orc_preamble_M1(int mode, uint8_t bits[127]);      /* src/hfdl.c:440-459 */
void    orc_training_T(uint8_t bits[15]);                  /* src/hfdl.c:157-160,181 */
void    orc_scrambler_bits(uint8_t *bits, int32_t n);      /* src/hfdl.c:300-347 (+liquid msequence) */
/* table position written by the k-th push / read by the k-th pop: src/hfdl.c:378-403 */
void    orc_deinterleave_maps(int mode, int32_t *push_pos, int32_t *pop_pos);
/* soft symbols (after eq) -> octets: src/hfdl.c:993-1056. bitmask_lsb = c->bitmask&1 */
int32_t orc_decode_user_data(int mode, const orc_cf *symbols, int bitmask_lsb, uint8_t *octets);
/* liquid modem soft demod restatement (a16) */
void    orc_modem_demod_soft(int arity, orc_cf x, uint8_t *soft);
orc_cf  orc_modem_modulate(int arity, uint32_t sym);                 /* modem_modulate_psk: cexpjf(gray_decode(sym) * 2 * pi / M) */
uint32_t orc_modem_demod_hard(int arity, orc_cf x, float *phase_error);
uint32_t orc_modem_demod_hard(int arity, orc_cf x, float *phase_error);
orc_cf  orc_modem_modulate(int arity, uint32_t sym);

/* the numeric constants and static tables of the restatement (src/hfdl.c:29-46,147-160,252-253,485-505,613,699-709), for the comparison
 * with the reference's text: ints[29], floats[14], mf[19], t_seq[15] in the order tests/test_constants_cpu.py names */
void    orc_constants(int32_t *ints, float *floats, float *mf, float *t_seq);

/* ---------------- variant switches: sensitivity of the result to the UNPINNED readings ----------------
 * liquid-dsp is absent here, so every choice below is a recollection or differs between liquid releases.  The defaults (all zero,
 * dmin 4.0) are the restatement every parity test uses and the GPU implements; profiles/variant_study.py decodes the same traffic
 * under each alternative and tabulates which ones change a decoded octet (profiles/r03_variant_sensitivity.md).  A variant is
 * process-global: set it, THEN create channels (filter tables are designed at create time). */
typedef struct {
	int32_t symsync_reset_both;  /* symsync_crcf_reset: 0 = clears the matched-filter bank only (default), 1 = both banks */
	int32_t resamp_kind;         /* arbitrary resampler of msresamp_crcf: 0 = 24-bit fixed-point phase, 256 branches, fc = min(0.515 r, 0.49)
	                                (liquid >= 1.3.2); 1 = float phase + linear interpolation between adjacent branches, 64 branches,
	                                fc = 0.4 (liquid <= 1.3.1 as recollected); 2 = float phase + interpolation, 256 branches, fc as 0;
	                                3 = fixed phase, 64 branches, fc as 0 */
	int32_t kaiser_arg;          /* Kaiser window argument: 0 = 2t/N (default), 1 = 2t/(N-1) */
	float   soft_dmin_init;      /* 8-PSK soft de-mapper: initial "nearest 0 / 1" distance (default 4.0) */
	int32_t lfsr_kind;           /* scrambler: 0 = v = (v<<1 | b), taps 0x4001, fill 0x4d4b (default = msequence API of liquid >= 1.6);
	                                1 = the pre-1.6 API restated literally: genpoly 0x8002 >> 1, fill = bit-reversed 0x6959 (must equal 0);
	                                2 = a right-shifting register with the same numbers (the other way to read the new API) */
	int32_t eqlms_norm;          /* eqlms step normalisation: 0 = running sum of |x|^2 (default), 1 = sum recomputed every step, 2 = none */
	int32_t agc_double;          /* agc y2 recursion: 0 = fp32 (default), 1 = evaluated in double as liquid's (1.0 - alpha) literal implies */
	int32_t design_float;        /* filter design: 0 = double precision (default), 1 = single precision with liquid's series (besseli0f, sincf) */
	int32_t perr_kind;           /* demodulator phase error: 0 = Im(r conj(x_hat)) (default), 1 = angle of r conj(x_hat) */
	int32_t dot_order;           /* dot products: 0 = sequential (default), 1 = even / odd partial sums (a 4-lane SIMD dotprod) */
	int32_t symsync_bank_floor;  /* filter-bank index: 0 = roundf(bf) (default), 1 = floorf(bf) */
	/* recollected constants (0 = the value used, a factor otherwise): what a mis-remembered number would do */
	float   symsync_dmf_scale;   /* derivative filter normalised to this * 0.06 / max|h dh| (0 -> 1.0) */
	float   symsync_lf_b;        /* loop filter feed-back coefficient b (0 -> 0.495) */
	float   soft_gamma_scale;    /* 8-PSK soft de-mapper gamma = this * 1.2 M (0 -> 1.0) */
	int32_t soft_floor;          /* soft bit conversion: 0 = C cast of (llr * 16 + 127) (truncation, default), 1 = floor */
	float   agc_y2_init;         /* AGC energy estimate at create (0 -> 1.0) */
	int32_t shared_math;         /* 1: expf / logf (AGC), sinf / cosf (carrier NCO) and atan2f (PSK slicer) come from tests/hostsim/shared_math.h
	                                -- fixed sequences of fp32 operations -- instead of glibc's libm: the arithmetic the device's test-only
	                                build -DHFDL_DM_STRICT runs, so that the two can be compared bit for bit (tests/test_strict_cpu.py,
	                                profiles/strict_study.py).  Rounding-level change only: ~1 ulp functions either way */
} orc_variant;
void orc_variant_default(orc_variant *v);
void orc_variant_set(const orc_variant *v);
void orc_variant_get(orc_variant *v);

/* ---------------- per-channel demodulator (src/hfdl.c:593-935) ---------------- */
typedef struct {
	int32_t freq;           /* channel frequency Hz */
	int32_t mode;           /* M1 index */
	int32_t len;
	uint8_t octets[960];
	float freq_err_hz, rssi_db, noise_floor_db;
	int32_t bit_rate;
	char slot;
	uint64_t sample_index;  /* 5400-sps sample counter at A2 detection */
	int32_t train_bits_bad, train_bits_total;
} orc_pdu;

typedef struct orc_channel orc_channel;
typedef void (*orc_pdu_sink)(void *ctx, const orc_pdu *pdu);

orc_channel *orc_channel_create(int32_t sample_rate, int32_t decimation, float transition_bw,
		int32_t centerfreq, int32_t frequency, int want_channelizer);  /* src/hfdl.c:468-534 */
void orc_channel_destroy(orc_channel *c);
const orc_ddc *orc_channel_ddc(const orc_channel *c);
const orc_cf *orc_channel_taps(const orc_channel *c);
/* StatsD counters of the hot path (src/hfdl.c:818,828,840) + noise floor (linear) + framer state */
void orc_channel_counters(const orc_channel *c, uint32_t out[4], float *noise_floor, int *framer_state);
/* the debug summary of src/hfdl.c:563-573 for one channel: out[0..5] = A1_found, A2_found, M1_found, M1_not_found, train_bits_bad,
 * train_bits_total (all frames); corr[0..2] = sums of |corr| at the A1 / A2 / M1 detections */
void orc_channel_summary(const orc_channel *c, uint32_t out[6], float corr[3]);
/* one block of the shared spectrum -> PDUs (src/hfdl.c:662-891) */
void orc_channel_process_spectrum(orc_channel *c, const orc_cf *spectrum, orc_pdu_sink sink, void *ctx);
/* enter after the channelizer: n samples at fs/decimation */
void orc_channel_process_baseband(orc_channel *c, const orc_cf *x, int32_t n, orc_pdu_sink sink, void *ctx);
/* stage taps (DATADUMPS analogue, src/hfdl.c:616-644): pointers valid until next process call */
typedef struct {
	const orc_cf *chan_out; int32_t chan_out_n;       /* channelizer output */
	const orc_cf *resampled; int32_t resampled_n;     /* after msresamp */
	const orc_cf *mf_out; int32_t mf_out_n;           /* after AGC + matched filter */
	const orc_cf *symbols; int32_t symbols_n;         /* equalised on-time symbols */
	const float *agc_level;                           /* 1/g per resampled sample */
} orc_taps_view;
void orc_channel_taps_view(const orc_channel *c, orc_taps_view *v);
/* liquid resampler restatement alone, for stage tests */
int32_t orc_resamp_run(float rate, const orc_cf *x, int32_t n, orc_cf *y, uint32_t *phase_io, orc_cf *hist14);
void orc_resamp_filter(float rate, float *h /* 256*14 */, uint32_t *step);
void orc_symsync_filters(float *mf /*16*18*/, float *dmf /*16*18*/);
void orc_eq_initial_taps(float *w15);

/* ---------------- whole front end (src/main.c:699-774 wiring) ---------------- */
typedef struct orc_frontend orc_frontend;
orc_frontend *orc_frontend_create(int32_t sample_rate, int32_t centerfreq, const int32_t *freqs, int32_t nch);
orc_frontend *orc_frontend_create_mt(int32_t sample_rate, int32_t centerfreq, const int32_t *freqs, int32_t nch, int nthreads);
void orc_frontend_destroy(orc_frontend *f);
const orc_ddc *orc_frontend_ddc(const orc_frontend *f);
/* push exactly input_size new samples; nthreads worker threads over channels (reference: 1 thread/channel) */
void orc_frontend_push_block(orc_frontend *f, const orc_cf *samples, int nthreads, orc_pdu_sink sink, void *ctx);
const orc_cf *orc_frontend_spectrum(const orc_frontend *f);
orc_channel *orc_frontend_channel(orc_frontend *f, int32_t i);

#ifdef __cplusplus
}
#endif
#endif
