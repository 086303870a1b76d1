// hostsim.cpp -- TEST HARNESS ONLY: compiles the product's demod_logic.h (device code: modem, framer FSM, header triage),
// demod_tables.h and planner.h with g++ so the demodulator's control logic and the host-side planner can be checked
// against the oracle on a machine without a GPU.  The device qualifiers and the four HIP names demod_logic.h uses are
// defined HERE (the product headers carry no host branch); the block loop around the logic is tests/hostsim/serial_demod.h.
// Never linked into libhfdl_gpu.so; nothing in dumphfdl_amd/ calls it.
#include <cstdlib>
#include <cstring>
#include <vector>
#include <complex>
// ---- shims for compiling device code on the host (one "lane")
#define __device__
#define __host__
static const struct { unsigned x; } threadIdx = { 0 };
static inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
#ifdef HFDL_DM_STRICT              // the arithmetic of the device's test-only strict build, on the host: fixed-sequence elementary functions
#define SM_FN static inline
#include "shared_math.h"
#define HFDL_ATAN2F sm_atan2f
#endif
#include "serial_demod.h"
#include "../../dumphfdl_amd/csrc/demod_tables.h"
#include "../../dumphfdl_amd/csrc/planner.h"

using namespace hfdl;

struct Sim {
	DemodTables tab;
	DemodConst K;
	ChanState st;
	std::vector<cf> rs, agc, mf, data, tap_rs, tap_mf, tap_sym;
	std::vector<float> lvl, tap_lvl;
	std::vector<FrameRec> frames;
	int frame_count = 0;
	int tap_counts[2] = { 0, 0 };
	int cap;
};

extern "C" {

Sim *sim_create(float resamp_rate, int cap)
{
	Sim *s = new Sim();
	build_demod_tables(s->tab, resamp_rate);
	s->K.rs_h = s->tab.rs_h; s->K.rs_step = s->tab.rs_step; s->K.mf = s->tab.mf;
	s->K.ss_mf = s->tab.ss_mf; s->K.ss_dmf = s->tab.ss_dmf;
	s->K.lf_b0 = s->tab.lf_b0; s->K.lf_a1 = s->tab.lf_a1; s->K.ss_rate_adj = s->tab.ss_rate_adj;
	s->K.eq_h0 = s->tab.eq_h0; s->K.a_hi = s->tab.a_hi; s->K.a_lo = s->tab.a_lo;
	s->K.m1_hi = s->tab.m1_hi; s->K.m1_lo = s->tab.m1_lo; s->K.corr_tab = s->tab.corr_tab; s->K.psk_pts = &s->tab.psk_pts[0][0];
	s->K.a1_lo = s->tab.a1_lo; s->K.a1_hi = s->tab.a1_hi; s->K.a2_lo = s->tab.a2_lo; s->K.a2_hi = s->tab.a2_hi; s->K.pos_min = s->tab.pos_min;
	chan_state_init(s->st, s->tab.eq_h0);
	s->cap = cap;
	s->rs.resize(cap); s->agc.resize(cap); s->mf.resize(cap); s->lvl.resize(cap);
	s->tap_rs.resize(cap); s->tap_mf.resize(cap); s->tap_sym.resize(cap); s->tap_lvl.resize(cap);
	s->data.resize(2 * MAX_DATA_SYMBOLS);
	s->frames.resize(16);
	return s;
}

void sim_destroy(Sim *s) { delete s; }

// returns number of frames finished in this block; frame records + their data symbols are copied out
int sim_block(Sim *s, const float *in, int n_in, FrameRec *frames_out, float *symbols_out /* per frame 2*5040 floats */)
{
	BlockIo io;
	io.rs = s->rs.data(); io.agc = s->agc.data(); io.mf = s->mf.data(); io.lvl = s->lvl.data(); io.cap = s->cap;
	io.data = s->data.data(); io.frames = s->frames.data(); io.frame_count = &s->frame_count; io.frame_cap = 16;
	io.tap_resampled = s->tap_rs.data(); io.tap_mf = s->tap_mf.data(); io.tap_symbols = s->tap_sym.data();
	io.tap_level = s->tap_lvl.data(); io.tap_counts = s->tap_counts; io.channel = 0;
	s->frame_count = 0;
	demod_block_serial(s->st.s, s->st.a, s->K, io, (const cf *)in, n_in);
	for (int i = 0; i < s->frame_count; i++) {
		frames_out[i] = s->frames[i];
		std::memcpy(symbols_out + (size_t)i * 2 * MAX_DATA_SYMBOLS, s->data.data() + (size_t)s->frames[i].slot * MAX_DATA_SYMBOLS,
				sizeof(cf) * MAX_DATA_SYMBOLS);
	}
	return s->frame_count;
}

int sim_taps(Sim *s, float *rs, float *mf, float *sym, float *lvl, int *counts)
{
	counts[0] = s->tap_counts[0]; counts[1] = s->tap_counts[1];
	std::memcpy(rs, s->tap_rs.data(), sizeof(cf) * counts[0]);
	std::memcpy(mf, s->tap_mf.data(), sizeof(cf) * counts[0]);
	std::memcpy(lvl, s->tap_lvl.data(), sizeof(float) * counts[0]);
	std::memcpy(sym, s->tap_sym.data(), sizeof(cf) * counts[1]);
	return 0;
}

void sim_psk_soft(int arity, float re, float im, uint8_t *soft)
{
	static DemodTables tab;
	static bool made = false;
	if (!made) { build_demod_tables(tab, 0.6912f); made = true; }
	cf x; x.x = re; x.y = im;
	psk_soft(arity, x, soft, PskTable{&tab.psk_pts[0][0]});
}

int sim_pdu_triage(const uint8_t *buf, uint32_t len, int *kind, uint32_t *hdr_len) { return pdu_triage(buf, len, kind, hdr_len); }

void sim_lpdu_walk(const uint8_t *buf, uint32_t len, uint8_t *counts)
{
	int kind = 0; uint32_t hl = 0;
	for (int i = 0; i < 5; i++) counts[i] = 0;
	if (pdu_triage(buf, len, &kind, &hl) != 0 || kind == 0) return;
	const LpduCounts c = lpdu_walk(buf, len, kind, hl);
	counts[0] = c.processed; counts[1] = c.good; counts[2] = c.bad_fcs; counts[3] = c.too_short; counts[4] = c.truncated;
}

void sim_tables(float resamp_rate, DemodTables *out) { build_demod_tables(*out, resamp_rate); }

// planner: geometry + channel constants + time-domain taps as the GPU shim computes them
int sim_plan(float tbw, int decimation, float shift, Plan *out) { return plan_block(*out, tbw, decimation, shift) ? 0 : -1; }
int sim_fft_decimation_rate(int fs, int target) { return fft_decimation_rate(fs, target); }
float sim_transition_bw(int fs, int hz) { return relative_transition_bw(fs, hz); }
void sim_bandpass(float *out, int length, float lowcut, float highcut)
{
	std::vector<float> lp; float cut = -1.f;
	design_bandpass((std::complex<float> *)out, length, lowcut, highcut, lp, cut);
}
size_t sim_sizeof_tables(void) { return sizeof(DemodTables); }
size_t sim_sizeof_constants(void) { return sizeof(HfdlConstants); }
void sim_constants(HfdlConstants *out) { hfdl_constants(*out); }
size_t sim_sizeof_framerec(void) { return sizeof(FrameRec); }

}
