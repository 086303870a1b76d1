// demod_core.h -- per-channel HFDL demodulator, one block of channelizer output per call (gfx950, device only).
//
// Replaces the body of hfdl_decoder_thread after fastddc_inv_cc (reference src/hfdl.c:676-892):
// msresamp -> AGC -> matched filter -> symsync -> Costas -> LMS equaliser -> M-PSK slicer -> preamble
// correlator / framer.
//
// Mapping: ONE WORKGROUP OF THREE WAVEFRONTS PER CHANNEL, the waves forming a software pipeline over chunks of DM_CHUNK
// 5400-sps samples.  The reference runs these stages as one serial loop; three of them are genuine recurrences (the AGC gain,
// the symbol-timing loop, the carrier loop + equaliser + framer FSM) and each costs a lone wavefront a few hundred cycles per
// sample in dependent-instruction latency.  They only feed FORWARD -- AGC -> timing recovery -> carrier/equaliser/framer --
// with one rare exception, so they run concurrently on three SIMDs of the CU:
//
//   wave 0   AGC recurrence (agc_crcf_execute) + matched filter (lanes over outputs) of chunk s
//   wave 1   symsync_crcf of chunk s-1: every output gathers its 18-tap windows from LDS, four dot products per DPP row reduction
//   wave 2   Costas loop, equaliser, slicer, framer FSM of chunk s-2 (demod_logic.h on_symbol)
//
// The exception: the framer resets the timing loop (symsync_crcf_reset at the end of a frame, on a failed preamble search,
// on carrier run-away: src/hfdl.c:714,751,969).  Wave 1 therefore runs AHEAD speculatively; when wave 2 hits a reset at
// sample k it publishes k, and wave 1 restarts at k+1 from the reset state, which is fully determined (empty matched-filter
// window, the last 18 matched-filter samples in the derivative window, initial loop scalars).  Results are those of the
// serial order, sample for sample; resets are rare (once per frame), so the re-run costs nothing measurable.
// The waves meet at one workgroup barrier per chunk and exchange progress through a double-buffered LDS mailbox.
#pragma once
#include <hip/hip_runtime.h>
#include "demod_logic.h"

namespace hfdl {

#ifndef HFDL_DM_CHUNK
#define HFDL_DM_CHUNK 32
#endif
constexpr int DM_WAVES = 3, DM_THREADS = 64 * DM_WAVES, DM_CHUNK = HFDL_DM_CHUNK;      // chunk: measured, profiles/r02_experiments.md
// The timing-recovery outputs travel from wave 1 to wave 2 through a RING of this many entries (round 6; until then a buffer of twice the
// launch's samples: 16 of the 46 bytes of LDS a sample cost).  Wave 1 never runs more than two chunks ahead of what wave 2 has finished
// (demod_block), a sample yields at most four outputs and wave 2 looks at 64 entries at a time: at most 2 x 32 x 4 + 64 entries are live.
constexpr int OUTQ_RING = 512;
static_assert((OUTQ_RING & (OUTQ_RING - 1)) == 0 && OUTQ_RING >= 2 * DM_CHUNK * 4 + 64 + 64, "ring: a power of two that holds what can be live");

// ---- DPP helpers (GFX9 encodings): all row-local, a row = 16 lanes ----
__device__ __forceinline__ float dpp_row_shr1(float old, float src)      // lane i <- src[i-1]; lane 0 of every row <- old
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shl1(float old, float src)      // lane i <- src[i+1]; lane 15 of every row <- old
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), 0x101, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shl2(float old, float src)      // lane i <- src[i+2]; lanes 14, 15 of every row <- old
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), 0x102, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_ror1(float src)                 // lane i <- src[i-1], lane 0 <- src[15] (inside the row)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x121, 0xf, 0xf, false));
}
// inclusive scan-sum inside every 16-lane row (lanes shifted in from outside the row read 0): lane 15 of a row ends up
// with the row's total.  One call reduces FOUR independent 16-term sums, one per row.
__device__ __forceinline__ float row_scan_sum(float v)
{
	int x = __float_as_int(v);
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true)));
	x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true)));
	return __int_as_float(x);
}
// inclusive scan-max inside every 16-lane row (lanes shifted in from outside the row read 0, so the result is max(0, ...)).
// v_max_f32 with the DPP operand, one instruction per step: written through the compiler's fmaxf each step is a DPP move, a
// canonicalising v_max x, x and the max itself.  (s_nop 1: the two wait states between a VALU write and a DPP read of the register.)
__device__ __forceinline__ float row_scan_max(float v)
{
	asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
			"s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
			"s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
			"s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 0"
			: "+v"(v));
	return v;
}
// x / 3.0f, correctly rounded, without the IEEE division sequence (12 instructions on the timing loop's output path): reciprocal
// multiply plus one FMA residual correction (Markstein); checked against true division over 6e4 random floats
__device__ __forceinline__ float div3(float x)
{
	const float c = 0x1.555556p-2f;
	const float q = x * c;
	return __builtin_fmaf(__builtin_fmaf(-3.0f, q, x), c, q);
}
__device__ __forceinline__ float lane_value(float v, int lane)           // wave-uniform copy of one lane (lane is uniform)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// what the three waves share through LDS besides the sample buffers
struct DemodShared {
	cf *outq;                      // symsync outputs of the launch, in order: output j at outq[j & (OUTQ_RING - 1)]
	int outq_cap;                  // outputs a launch may produce at most (the 16-bit counts of cum[]); beyond it they are dropped
	uint16_t *cum;                 // cum[k] = outputs produced up to and including input sample k
	const float2 *sstab;           // [16 banks][64 lanes] {tap t, tap t+16} of the lane's row: rows 0,1 matched filter, rows 2,3 derivative
	ChanScalars *S;                // the channel's scalars; every wave owns a disjoint set of fields
	int *mbox;                     // [2][4] progress mailbox: {mf_ready, ss_to, s3_done, s3_reset}
	float *sink;                   // [64] write-only scratch, one word per lane
	cf *stage;                     // [64] the carrier wave's data symbols of one chunk (a chunk has at most 64 outputs)
};

// progress of the three stages as every wave sees it after a step's barrier
struct PipeProgress {
	int mf_ready = 0, ss_ready = 0, s3_done = 0;
	bool restart = false;
	__device__ __forceinline__ void read(const int *mb)
	{
		mf_ready = mb[0];
		s3_done = mb[2];
		restart = mb[3] != 0;
		ss_ready = restart ? s3_done : mb[1];          // a reset discards what wave 1 computed beyond the reset sample
	}
};

// ---------------- wave 0: AGC + matched filter of samples [a0, a1) ----------------

__device__ __forceinline__ void agc_mf_chunk(float &g, float &y2, const ChanArrays &a, const DemodConst &T, const BlockIo &io, int a0, int a1, int lane)
{
	// agc_crcf_execute (src/hfdl.c:686): a per-sample gain recurrence, wave-uniform
	const float alpha = AGC_BANDWIDTH;
	const cf xin = (a0 + lane < a1 && lane < DM_CHUNK) ? io.rs[a0 + lane] : cf{0.f, 0.f};      // the chunk, one sample per lane
	for (int k = a0; k < a1; k++) {
		cf x; x.x = lane_value(xin.x, k - a0); x.y = lane_value(xin.y, k - a0);
		cf y; y.x = x.x * g; y.y = x.y * g;
		const float e = y.x * y.x + y.y * y.y;
		y2 = (1.0f - alpha) * y2 + alpha * e;
		// exp(a ln y2) == 2^(a log2 y2): one v_log_f32 + one v_exp_f32 on the gain recurrence's critical path
		if (y2 > 1e-6f) g *= __builtin_amdgcn_exp2f(-0.5f * alpha * __builtin_amdgcn_logf(y2));
		if (g > 1e6f) g = 1e6f;
		io.agc[k] = y;
		io.lvl[k] = __builtin_amdgcn_rcpf(g);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	// 19-tap matched filter, lanes over the chunk's outputs (firfilt_crcf, src/hfdl.c:694-695)
	const int k = a0 + lane;
	if (k < a1) {
		float ar = 0, ai = 0;
		for (int t = 0; t < D_MF; t++) {
			const int idx = k - t;
			const cf x = idx >= 0 ? io.agc[idx] : a.mf_hist[-idx - 1];
			ar += T.mf[t] * x.x;
			ai += T.mf[t] * x.y;
		}
		io.mf[k].x = ar; io.mf[k].y = ai;
	}
}

// ---------------- wave 1: symbol timing recovery (symsync_crcf_execute, src/hfdl.c:696) ----------------

// The filter windows are not kept anywhere: window sample `age` of input sample k is matched-filter output k - age, and the
// block's matched-filter outputs sit in LDS behind a prefix holding the previous block's last 18 (SS_HIST entries before
// mf[0], padded so that the unused "tap t + 16" reads of lanes t >= 2 stay in bounds).  Every output GATHERS its taps with
// one LDS read per lane, issued one input sample ahead; nothing is shifted per input sample.  liquid's symsync_crcf_reset
// clears the matched-filter window only: `valid_from` is the first input sample whose matched-filter window entries count
// (older ones read as zero in rows 0, 1); ChanScalars.ss_head carries it from block to block (<= 0 at block start).
constexpr int SS_HIST = 34;
struct SymsyncRegs {
	float rate, del, tau, bf, q, qhat, v1;
	int b, decim, j;               // filter-bank index, decimation counter, running output index of the block
	int valid_from;
};

__device__ __forceinline__ void symsync_load(SymsyncRegs &r, const ChanScalars &s, const ChanArrays &a, const BlockIo &io, int lane)
{
	// history prefix: mf[-1 - age] = the window the previous block left (canonical array form: index 0 newest, 18 - age older)
	if (lane < SS_HIST) {
		cf v; v.x = 0.f; v.y = 0.f;
		if (lane < D_SS_TAPS) v = a.ss_dmf[lane == 0 ? 0 : D_SS_TAPS - lane];
		io.mf[-1 - lane] = v;
	}
	r.rate = s.ss_rate; r.del = s.ss_del; r.tau = s.ss_tau; r.bf = s.ss_bf; r.q = s.ss_q; r.qhat = s.ss_qhat; r.v1 = s.ss_v1;
	r.b = s.ss_b; r.decim = (int)s.ss_decim; r.j = 0;
	r.valid_from = s.ss_head;
}

// symsync_crcf_reset as the framer left it after input sample k0-1: loop scalars initial, matched-filter window empty
__device__ __forceinline__ void symsync_restart(SymsyncRegs &r, const DemodShared &sh, int k0)
{
	r.rate = 1.5f; r.del = 1.5f; r.tau = 0.f; r.bf = 0.f; r.q = 0.f; r.qhat = 0.f; r.v1 = 0.f;
	r.b = 0; r.decim = 0;
	r.j = k0 > 0 ? (int)sh.cum[k0 - 1] : 0;
	r.valid_from = k0;
}

__device__ __forceinline__ void symsync_store(const SymsyncRegs &r, ChanScalars &s, ChanArrays &a, const BlockIo &io, int n_out, int lane)
{
	if (lane < D_SS_TAPS) a.ss_dmf[lane == 0 ? 0 : D_SS_TAPS - lane] = io.mf[n_out - 1 - lane];      // reaches into the prefix when n_out < 18
	if (lane == 0) {
		s.ss_rate = r.rate; s.ss_del = r.del; s.ss_tau = r.tau; s.ss_bf = r.bf; s.ss_q = r.q; s.ss_qhat = r.qhat; s.ss_v1 = r.v1;
		s.ss_b = r.b; s.ss_decim = (uint32_t)r.decim;
		const int vf = r.valid_from - n_out;
		s.ss_head = vf < -64 ? -64 : vf;
	}
}

// MASKED: some window entries of the chunk's first samples date from before the last timing-loop reset (only in the 18 samples after
// one); the common chunk runs the variant without the per-sample test and selects (14 instructions of this wave's ~75 per sample).
template <bool MASKED>
__device__ __forceinline__ void symsync_chunk(SymsyncRegs &r, const DemodConst &T, const BlockIo &io, const DemodShared &sh, int k0, int k1, int lane)
{
	const int row = lane >> 4, t = lane & 15;
	// &mf[k - t].component of this lane's row for k = 0; tap t + 16 is 16 samples further back
	const float *base = (const float *)(io.mf - t) + (row & 1);
	int bi = r.b < 0 ? 0 : (r.b >= D_SS_NPFB ? D_SS_NPFB - 1 : r.b);
	float2 h = sh.sstab[bi * 64 + lane];
	float w_lo = base[2 * k0], w_hi = base[2 * (k0 - 16)];
	// An output's two components are the row totals in lanes 15 (re) and 31 (im): those two lanes store them themselves with an ALL-lane
	// LDS write whose other lanes aim at a scratch word of their own -- no lane reads, no exec masking on the loop's path.  The
	// per-sample output counts collect in a register, one lane per input sample of the chunk, and go to LDS once per chunk.
	const bool out_lane = lane == 15 || lane == 31;
	float *const out_base = out_lane ? (float *)sh.outq + (lane == 31 ? 1 : 0) : sh.sink + lane;
	const int out_stride = out_lane ? 2 : 0;
	int cum_v = 0;
	for (int k = k0; k < k1; k++) {
		// the next input sample's window entries are fetched now, a whole iteration before they can be needed (sample k1 belongs to
		// the next chunk and may still be in the making: that value is never used)
		const float n_lo = base[2 * (k + 1)], n_hi = base[2 * (k + 1 - 16)];
		if (r.b < D_SS_NPFB) {
			float wl = w_lo, wh = w_hi;
			if (MASKED && k - (D_SS_TAPS - 1) < r.valid_from && row < 2) {          // matched-filter window entries from before the last reset are empty
				if (k - t < r.valid_from) wl = 0.f;
				if (k - t - 16 < r.valid_from) wh = 0.f;
			}
			int produced = 0;
			do {
				// four 18-tap dot products at once: row 0/1 = matched filter re/im, row 2/3 = derivative filter re/im
				const float p = row_scan_sum(h.x * wl + h.y * wh);
				if (__builtin_expect(r.j < sh.outq_cap, 1)) out_base[out_stride * (r.j & (OUTQ_RING - 1))] = div3(p);
				r.j++;
				if (r.decim == 2) {
					r.decim = 0;
					const float mx = lane_value(p, 15), my = lane_value(p, 31);
					const float dx = lane_value(p, 47), dy = lane_value(p, 63);
					float q = mx * dx + my * dy;
					q = q > 1.0f ? 1.0f : (q < -1.0f ? -1.0f : q);
					r.q = q;
					const float v0 = q - T.lf_a1 * r.v1;
					r.qhat = T.lf_b0 * v0;
					r.v1 = v0;
					r.rate += T.ss_rate_adj * r.qhat;
					r.del = r.rate + r.qhat;
				}
				r.decim++;
				r.tau += r.del;
				r.bf = r.tau * (float)D_SS_NPFB;
				r.b = (int)roundf(r.bf);
				produced++;
				if (r.b < D_SS_NPFB) h = sh.sstab[(r.b < 0 ? 0 : r.b) * 64 + lane];      // another output from this input sample
			} while (r.b < D_SS_NPFB && produced < 4);
		}
		r.tau -= 1.0f;
		r.bf -= (float)D_SS_NPFB;
		r.b -= D_SS_NPFB;
		bi = r.b < 0 ? 0 : (r.b >= D_SS_NPFB ? D_SS_NPFB - 1 : r.b);
		h = sh.sstab[bi * 64 + lane];                   // branch of the next input sample: fetched while the stores drain
		cum_v = (lane == k - k0) ? (r.j < 65535 ? r.j : 65535) : cum_v;
		w_lo = n_lo; w_hi = n_hi;
	}
	if (lane < k1 - k0) sh.cum[k0 + lane] = (uint16_t)cum_v;
}

// ---------------- wave 2: carrier loop, equaliser, slicer, framer (src/hfdl.c:709-891) ----------------

struct CarrierRegs {
	float eu, ev, ex2, ewx, ewy;   // equaliser: lane 1 + t (t < 15) of rows 0 and 1 = tap t (0 oldest); row 0 holds (x, y), row 1 (y, -x)
	float px, py;                  // lane i < 16: PSK constellation entry i (demod_tables.h psk_pts)
#ifdef HFDL_DM_PROBE               // experiment builds (profiles/probe_states.py): = 2 carrier-wave cycles per framer state, = 3 outputs per state,
                                   // reported in the phase-cycle slots of the level tap
	unsigned long long pa = 0, pb = 0, pc = 0;
#endif
};

// The carrier wave's slicer.  modem_demodulate_psk takes arg(x), subtracts pi (1 - 1/M) and walks a binary reference ladder:
// that IS the nearest constellation point by angle, i.e. the point with the largest Re(x conj(p)).  Here every lane of row 0
// holds one point of the table (demod_tables.h psk_pts): one multiply-add per lane, a row max, a compare -- instead of atan2f
// and the ladder (~45 instructions of the carrier loop's per-symbol critical path).  Decisions can differ from the atan2f form
// only when x is within rounding of a decision boundary, exactly as two atan2f implementations differ from each other; what
// leaves this function into the loop is the phase error against the chosen point.  (The decoded bits of data symbols come from
// the burst decoder's own soft de-mapper, which keeps the reference's arg() form.)
struct LaneSlicer {
	float px, py;                  // lane i < 16: table entry i
	int lane;
	__device__ __forceinline__ uint32_t operator()(int arity, cf x, float *phase_error) const
	{
		uint32_t sym;
		cf xh;
		if (arity == 1) {
			sym = (x.x > 0) ? 0 : 1;
			xh.x = sym ? -1.0f : 1.0f; xh.y = 0.0f;
		} else {
			const int M = 1 << arity, base = M - 2;
			const bool mine = lane >= base && lane < base + M;
			const float d = mine ? x.x * px + x.y * py : -1.0f;
			const float best = lane_value(row_scan_max(d), 15);        // >= 0: some point is within 90 degrees of x
			const unsigned long long hit = __builtin_amdgcn_fcmpf(d, best, 1 /* FCMP_OEQ */);       // lanes without a point hold -1: never equal
			int win = hit ? (int)__builtin_ctzll(hit) : base;           // NaN input: no lane compares equal
			const uint32_t lin = (uint32_t)(win - base);
			sym = lin ^ (lin >> 1);
			xh.x = lane_value(px, win); xh.y = lane_value(py, win);
		}
		if (phase_error) *phase_error = x.y * xh.x - x.x * xh.y;
		return sym;
	}
};

__device__ __forceinline__ void carrier_load(CarrierRegs &c, const ChanScalars &s, const ChanArrays &a, const DemodConst &T, int lane)
{
	c.px = lane < 16 ? T.psk_pts[2 * lane] : 0.f;
	c.py = lane < 16 ? T.psk_pts[2 * lane + 1] : 0.f;
	const int t = (lane & 15) - 1;                 // the window sits in lanes 1..15 of a row: a push is ONE row shift whose vacated lane 15 takes the new sample
	const bool act = t >= 0 && lane < 32, row1 = (lane >> 4) & 1;
	int j = s.eq_head + t; if (j >= D_EQ) j -= D_EQ;
	const cf e = act ? a.eq_buf[j] : cf{0.f, 0.f};
	c.eu = row1 ? e.y : e.x;
	c.ev = row1 ? -e.x : e.y;
	c.ex2 = act ? a.eq_x2[j] : 0.f;
	const cf w = act ? a.eq_w[t] : cf{0.f, 0.f};
	c.ewx = w.x; c.ewy = w.y;
}

__device__ __forceinline__ void carrier_store(const CarrierRegs &c, ChanArrays &a, int lane)
{
	if (lane >= 1 && lane <= D_EQ) {
		cf e; e.x = c.eu; e.y = c.ev;
		a.eq_buf[lane - 1] = e; a.eq_x2[lane - 1] = c.ex2;
		cf w; w.x = c.ewx; w.y = c.ewy;
		a.eq_w[lane - 1] = w;
	}
}

// processes samples [k0, k1); returns the index of the sample during which the framer reset the timing loop (the wave stops
// after that sample), or -1.
// The loop runs over the timing-recovery OUTPUTS of the chunk, not over its input samples: two of three input samples yield an
// output, and a loop nest "per sample: level, noise-floor clock, output count; per output: ..." spends ~30 instructions and seven
// branches per input sample on bookkeeping -- a quarter of this wave's time.  The input sample an output belongs to is the first one
// whose cumulative output count exceeds the output's index (one compare across the lanes that hold the chunk's counts + s_ff1);
// what the reference does per input sample (the noise-floor estimator's clock while searching, src/hfdl.c:700-702; the sample
// counter) is caught up in closed form in front of every symbol that can change the framer state (those that go through on_symbol())
// and at the end of the chunk -- the state is the same for all the samples in between, it only changes on such symbols.
template <bool TAPS>
__device__ __forceinline__ int carrier_chunk(CarrierRegs &c, ChanScalars &s, ChanArrays &a, const DemodConst &T, const BlockIo &io, const DemodShared &sh,
		int k0, int &k1, int &nsym, int lane)
{
	const int t = (lane & 15) - 1;
	const bool row1 = (lane >> 4) & 1, act = t >= 0 && lane < 32;
	int n = k1 - k0;                             // <= DM_CHUNK <= 64
	// the chunk's levels, output counts and (up to 64) timing-recovery outputs, one per lane: the loop below takes them with
	// v_readlane instead of a dependent LDS round trip per sample
	const float lv_l = (lane < n) ? io.lvl[k0 + lane] : 0.f;
	const int cum_l = (lane < n) ? (int)sh.cum[k0 + lane] : 0x7fffffff;
	const int jbase = k0 > 0 ? (int)sh.cum[k0 - 1] : 0;
	const cf oq_l = (jbase + lane < sh.outq_cap) ? sh.outq[(jbase + lane) & (OUTQ_RING - 1)] : cf{0.f, 0.f};
	{   // the chunk's outputs are taken from the 64 lanes of oq_l: a chunk that produced more (a timing loop far off its rate: up to four
		// outputs per input sample) is cut where they end, and k1 tells the caller
#ifndef HFDL_DM_OUT_LANES            // (a smaller value makes every chunk take the cut: how that path was tested, profiles/r03_experiments.md)
#define HFDL_DM_OUT_LANES 64
#endif
		const unsigned long long over = __ballot(lane < n && cum_l - jbase > HFDL_DM_OUT_LANES);
		if (__builtin_expect(over != 0, 0)) { n = (int)__builtin_ctzll(over); k1 = k0 + n; }      // n >= 1: one sample yields at most 4 outputs
	}
	int jstop = __builtin_amdgcn_readlane(cum_l, n - 1);
	if (jstop > sh.outq_cap) jstop = sh.outq_cap;
	const uint64_t cnt0 = s.sample_cnt;
	int kdone = 0;                               // input samples of the chunk whose per-sample bookkeeping is done: k0 .. k0 + kdone - 1
	int reset_at = -1;
	// Data symbols of a frame in progress go to HBM (the frame buffer the burst decoder reads) a chunk at a time: every lane writes the
	// (wave-uniform) symbol to the next slot of a 64-entry LDS stage, and one coalesced store per chunk moves them -- instead of a lane-0
	// store under an exec mask with its 64-bit address arithmetic per symbol.  A chunk's staged symbols are consecutive in one frame buffer
	// (a frame's data ends with the frame; the next frame's starts a preamble later).
	int staged = 0, stage_at = 0;                // symbols in the stage, frame-buffer index of the first
	auto flush_stage = [&]() {
		if (lane < staged) io.data[stage_at + lane] = sh.stage[lane];
		staged = 0;
	};
	// noise-floor estimator clock of input samples [kdone, upto] of the chunk (src/hfdl.c:700-702): it ticks while the framer searches, and
	// every 256th tick takes that sample's level.  The framer state is the same for all of them (no output in between).
	// Called in front of every symbol handed to on_symbol() (the only place the framer state changes) and at the end of the chunk, so
	// every sample is counted in the state the reference saw it in.
	auto catch_up = [&](int upto) {
		if (s.fr_state == FR_A1) {
			const uint32_t d = (uint32_t)(upto + 1 - kdone), c0 = s.nf_clk;
			uint32_t first = (NF_CLK_MASK - c0) & NF_CLK_MASK;          // ticks until the low byte reads 0xFF (0: a whole turn)
			if (first == 0) first = NF_CLK_MASK + 1;
			s.nf_clk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c0 + d));
			if (first <= d) {                                // at most once: d <= 64
				const float level = lane_value(lv_l, kdone + (int)first - 1);
				s.noise_floor = NF_KEEP * s.noise_floor + NF_TAKE * fminf(s.noise_floor, level) + NF_BIAS;
			}
		}
		kdone = upto + 1;
	};
	bool runaway = fabsf(s.dphi) > COSTAS_RUNAWAY_DPHI && s.fr_state == FR_A1;
	const int pair_sel = (lane & 1) << 2;        // byte offset of the pair's second output in a lane permute
	for (int j = jbase; j < jstop;) {
#ifdef HFDL_DM_PROBE
		const unsigned long long tA0 = __builtin_amdgcn_s_memtime();
		const int st0 = s.fr_state;
#endif
		int jo;                                   // the output the rest of the iteration is about
		bool on_time;
		if (!(s.symsync_out_idx & 1u) && j + 1 < jstop && !runaway) {
			// A symbol's two timing-recovery outputs at once, the off-time one in the even lanes and the on-time one in the odd lanes:
			// one rotation (sin, cos, four products) and one two-lane move of the equaliser window serve both -- the window's newest
			// entries are lanes 14 (even: the off-time sample) and 15 (odd: the on-time sample) of its row.  Same arithmetic per
			// output as the single step below; only the carrier phase of each has to be stepped and wrapped on its own.
			const int sel = ((j - jbase) << 2) + pair_sel;
			cf oi;
			oi.x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(oq_l.x)));
			oi.y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(oq_l.y)));
			float ph1, ph2;
			{
				const float ph = s.phi + s.dphi;
				const float dn = ph - (float)(2.0 * M_PI), up = ph + (float)(2.0 * M_PI);
				ph1 = ph > (float)M_PI ? dn : (ph < -(float)M_PI ? up : ph);
			}
			{
				const float ph = ph1 + s.dphi;
				const float dn = ph - (float)(2.0 * M_PI), up = ph + (float)(2.0 * M_PI);
				ph2 = ph > (float)M_PI ? dn : (ph < -(float)M_PI ? up : ph);
			}
			s.phi = ph2;
#ifdef HFDL_DM_LIBM_TRIG
			float sp, cp;
			sincosf((lane & 1) ? ph2 : ph1, &sp, &cp);
#else
			const float rev = ((lane & 1) ? ph2 : ph1) * 0.15915494309189535f;
			const float sp = __builtin_amdgcn_sinf(rev), cp = __builtin_amdgcn_cosf(rev);
#endif
			cf r;
			r.x = oi.x * cp + oi.y * sp;
			r.y = oi.y * cp - oi.x * sp;
			const float x2n = r.x * r.x + r.y * r.y;
			const float x2n1 = lane_value(x2n, 0), x2n2 = lane_value(x2n, 1);
			const float x2o1 = lane_value(c.ex2, 1), x2o2 = lane_value(c.ex2, 2);
			const float nu = row1 ? r.y : r.x, nv = row1 ? -r.x : r.y;
			c.eu = dpp_row_shl2(nu, c.eu);
			c.ev = dpp_row_shl2(nv, c.ev);
			c.ex2 = dpp_row_shl2(x2n, c.ex2);
			s.eq_x2sum = s.eq_x2sum + x2n1 - x2o1;
			s.eq_x2sum = s.eq_x2sum + x2n2 - x2o2;
			s.eq_count += 2;
			jo = j + 1; j += 2; s.symsync_out_idx += 2;
			on_time = true;
		} else {
			cf oi;
			oi.x = lane_value(oq_l.x, j - jbase); oi.y = lane_value(oq_l.y, j - jbase);
			// costas_cccf_step + execute, :256-258, :284-292
			{   // selects, not branches: a taken branch costs a lone wave ~4 instruction slots (profiles/micro)
				const float ph = s.phi + s.dphi;
				const float dn = ph - (float)(2.0 * M_PI), up = ph + (float)(2.0 * M_PI);
				s.phi = ph > (float)M_PI ? dn : (ph < -(float)M_PI ? up : ph);
			}
			// |phi| <= pi: the hardware sin/cos (argument in revolutions) needs no range reduction
#ifdef HFDL_DM_LIBM_TRIG              // experiment (profiles/r03_experiments.md): the library's accurate sincosf instead of v_sin / v_cos
			float sp, cp;
			sincosf(s.phi, &sp, &cp);
#else
			const float rev = s.phi * 0.15915494309189535f;
			const float sp = __builtin_amdgcn_sinf(rev), cp = __builtin_amdgcn_cosf(rev);
#endif
			cf r;
			r.x = oi.x * cp + oi.y * sp;
			r.y = oi.y * cp - oi.x * sp;
			if (__builtin_expect(runaway, 0)) {  // costas run-away while searching (src/hfdl.c:709-716); dphi only moves in on_symbol()
				s.dphi = s.phi = 0.f;
				symsync_reset(s, a);
				runaway = false;
			}
			// eqlms_cccf_push: the row moves down one lane and lane 15, which has no source inside the row, takes the new sample
			{
				const float x2n = r.x * r.x + r.y * r.y;
				const float x2o = lane_value(c.ex2, 1);
				const float nu = row1 ? r.y : r.x, nv = row1 ? -r.x : r.y;
				c.eu = dpp_row_shl1(nu, c.eu);
				c.ev = dpp_row_shl1(nv, c.ev);
				c.ex2 = dpp_row_shl1(x2n, c.ex2);
				s.eq_x2sum = s.eq_x2sum + x2n - x2o;
				s.eq_count++;
			}
			on_time = (s.symsync_out_idx & 1u) != 0;
			jo = j; j++; s.symsync_out_idx++;
		}
		if (on_time) {
			// eqlms_cccf_execute, sum conj(w_i) x_i: real part reduced in row 0, imaginary part in row 1, one scan
			const float p = row_scan_sum(act ? c.ewx * c.eu + c.ewy * c.ev : 0.f);
			cf y; y.x = lane_value(p, 15); y.y = lane_value(p, 31);
			if (s.fr_state == FR_EQ_TRAIN) {
				// eqlms_cccf_step(d = known T symbol, d_hat = y)
				bool run = true;
				if (!s.eq_full) { if (s.eq_count < (uint32_t)D_EQ) run = false; else s.eq_full = 1; }
				if (run) {
					const float tv = t_symbol(s.T_idx) * ((s.bitmask & 1u) ? -1.0f : 1.0f);
					const float er = tv - y.x, ei = -(0.0f - y.y);
					const float bx = row1 ? -c.ev : c.eu, by = row1 ? c.eu : c.ev;       // the window sample (x, y) in either row
					const float pr = er * bx - ei * by, pi = er * by + ei * bx;
					c.ewx = c.ewx + EQ_STEP * pr / s.eq_x2sum;
					c.ewy = c.ewy + EQ_STEP * pi / s.eq_x2sum;
				}
				s.T_idx++;
			}
			if (TAPS && lane == 0) io.tap_symbols[nsym] = y;
			nsym++;
			// (The two straight-line copies of on_symbol()'s commonest paths below are guarded against drifting from it: the test-only strict
			// build runs EVERY symbol through on_symbol() in a serial loop with this pipeline's arithmetic forms emulated, and its PDUs must
			// equal this kernel's in every SNR bin -- tests/test_gpu_strict.py, strict_15 == shipped.)
			// The searching framer's symbol -- by far the commonest: a BPSK decision into the 127-bit window, the carrier loop's update, the
			// correlation against the A sequence, and nothing found.  Decided BEFORE anything is changed: such a symbol is finished here in
			// straight-line code; every other one (a detection, a frame in progress, the 13-frame re-centring) goes through on_symbol().
			bool plain = false;
			if (s.fr_state == FR_A1) {
				const bool neg = !(y.x > 0);
				const uint32_t bit = (neg ? 1u : 0u) ^ (s.bitmask & 1u);
				const uint64_t nhi = ((s.bits_hi << 1) | (s.bits_lo >> 63)) & 0x7FFFFFFFFFFFFFFFull, nlo = (s.bits_lo << 1) | bit;
				const int m = bits_correlate(nhi, nlo, T.a_hi, T.a_lo);
				if (__builtin_expect(m > T.a1_lo && m < T.a1_hi && s.s_state == SAMPLER_BITS && s.cur_arity == 1 && s.symbols_wanted <= 1
						&& s.symbol_cnt + 1 < (uint64_t)(NO_FRAME_TIMEOUT_FRAMES * SINGLE_SLOT_FRAME_LEN), 1)) {
					const float perr = y.y * (neg ? -1.0f : 1.0f) - y.x * 0.0f;      // modem phase error against (+-1, 0), as LaneSlicer writes it
					const float e = 0.5f * (fabsf(perr + COSTAS_ERR_LIMIT) - fabsf(perr - COSTAS_ERR_LIMIT));      // costas_cccf_adjust, :276-281
					s.err = e;
					s.phi += COSTAS_ALPHA * e;
					s.dphi += COSTAS_BETA * e;
					s.symbol_cnt++;
					s.bits_hi = nhi; s.bits_lo = nlo;
					runaway = fabsf(s.dphi) > COSTAS_RUNAWAY_DPHI;
					plain = true;
				}
			}
			else if (s.symbols_wanted > 1) {
				// A frame in progress, between two framer transitions (on_symbol() returns in front of its state switch): decision and
				// carrier loop update, the symbol to where the sampler wants it, the running signal level, the countdown -- the same
				// statements in the same order, laid out as one straight line instead of on_symbol()'s general control flow (which costs
				// this lone wave ten taken branches per symbol).  The framer state is not FR_A1 here and does not change.
				const int ki = (int)__builtin_ctzll(__ballot(cum_l > jo));
				if (ki >= kdone) kdone = ki + 1;      // the noise-floor clock stands still outside the search: nothing to catch up with
				const float level = lane_value(lv_l, ki);
				float perr;
				uint32_t bits = LaneSlicer{c.px, c.py, lane}(s.cur_arity, y, &perr);
				const float e = 0.5f * (fabsf(perr + COSTAS_ERR_LIMIT) - fabsf(perr - COSTAS_ERR_LIMIT));      // costas_cccf_adjust, :276-281
				s.err = e;
				s.phi += COSTAS_ALPHA * e;
				s.dphi += COSTAS_BETA * e;
				s.symbol_cnt++;
				if (s.s_state == SAMPLER_SYMBOLS) {
					if (s.use_data) {
						if (s.data_n < MAX_DATA_SYMBOLS) {
							if (staged == 0) stage_at = s.data_slot * MAX_DATA_SYMBOLS + s.data_n;
							sh.stage[staged++] = y;
							s.data_n++;
						}
					} else if (s.training_n < T_LEN) {
						a.training[s.training_n] = y;
						s.training_n++;
					}
				} else if (s.s_state == SAMPLER_BITS) {
					bits ^= s.bitmask;
					for (int b = 0; b < s.cur_arity; b++, bits >>= 1) {
						s.bits_hi = ((s.bits_hi << 1) | (s.bits_lo >> 63)) & 0x7FFFFFFFFFFFFFFFull;
						s.bits_lo = (s.bits_lo << 1) | (bits & 1u);
					}
				}
				s.signal_level = (s.signal_level * s.frame_symbol_cnt + level) / (s.frame_symbol_cnt + 1.0f);
				s.frame_symbol_cnt += 1.0f;
				s.symbols_wanted--;
				runaway = false;                     // it only counts while searching
				plain = true;
			}
			if (!plain) {
				if (staged) flush_stage();           // on_symbol() stores a data symbol itself: the stage holds consecutive symbols only
				// the input sample that produced this output: the first whose cumulative count exceeds j (lanes beyond the chunk hold INT_MAX)
				const int ki = (int)__builtin_ctzll(__ballot(cum_l > jo));
				if (ki >= kdone) catch_up(ki);       // before the framer state can change
				const float level = lane_value(lv_l, ki);
				s.sample_cnt = cnt0 + (uint64_t)ki;
				on_symbol(s, *sh.S, a, T, io, y, level, LaneSlicer{c.px, c.py, lane});
				runaway = fabsf(s.dphi) > COSTAS_RUNAWAY_DPHI && s.fr_state == FR_A1;
			}
		}
#if defined(HFDL_DM_PROBE) && HFDL_DM_PROBE >= 2
		{
			const unsigned long long dt = HFDL_DM_PROBE == 2 ? __builtin_amdgcn_s_memtime() - tA0 : 1ull;
			if (st0 == FR_A1) c.pa += dt; else if (st0 == FR_EQ_TRAIN) c.pb += dt; else if (st0 == FR_DATA_1 || st0 == FR_DATA_2) c.pc += dt;
		}
#endif
		if (__builtin_expect(s.ev_flags != 0, 0)) {
			if (s.ev_flags & EV_EQ_RESET) {      // eqlms_cccf_reset ran (framer reset): mirror it in the register window
				c.eu = 0.f; c.ev = 0.f; c.ex2 = 0.f;
				c.ewx = act ? T.eq_h0[t >= 0 ? t : 0] : 0.f; c.ewy = 0.f;
				s.ev_flags &= ~(uint32_t)EV_EQ_RESET;
			}
			if ((s.ev_flags & EV_SS_RESET) != 0 && reset_at < 0) {
				// the timing loop was reset during this output's input sample: the outputs it had already produced from that sample still
				// go through (symsync_crcf_execute had returned them, src/hfdl.c:707-708), then the wave stops and wave 1 restarts after it
				const int kr = (int)__builtin_ctzll(__ballot(cum_l > jo));
				if (kr >= kdone) catch_up(kr);       // only after an off-time output (carrier run-away), which leaves the framer state alone
				reset_at = kr;
				int j1 = __builtin_amdgcn_readlane(cum_l, kr);
				if (j1 < jstop) jstop = j1;
			}
		}
	}
	if (staged) flush_stage();
	if (reset_at >= 0) {
		s.ev_flags = 0;
		s.sample_cnt = cnt0 + (uint64_t)reset_at + 1u;
		return k0 + reset_at;
	}
	if (kdone < n) catch_up(n - 1);              // input samples after the last output
	s.sample_cnt = cnt0 + (uint64_t)n;
	return -1;
}

// ---------------- one block ----------------

// All DM_THREADS threads of the channel's workgroup call this.  On return the channel's arrays (LDS, `a`) and scalars (*sh.S)
// hold the state after the block.  Returns the number of 5400-sps samples produced.
// TAPS: the per-stage debug taps (DATADUMPS analogue) and the phase cycle counters are compiled in; the production launch uses the
// variant without them (fewer live pointers in the carrier wave, which is short of SGPRs as it is).
template <bool TAPS>
__device__ inline int demod_block(ChanArrays &a, const DemodConst &T, const BlockIo &io, const DemodShared &sh, const cf *in, int n_in)
{
	const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
	ChanScalars &S = *sh.S;
	unsigned long long tR0 = __builtin_amdgcn_s_memtime();

	// ---- R: arbitrary resampler, 24-bit fixed-point phase, all threads over the outputs (msresamp_crcf_execute, src/hfdl.c:676)
	const uint32_t rs_phase = S.rs_phase;
	const uint64_t total = (uint64_t)n_in << 24;
	int n_out = 0;
	if ((uint64_t)rs_phase < total) n_out = (int)((total - rs_phase + T.rs_step - 1) / T.rs_step);
	if (n_out > io.cap - 4) n_out = io.cap - 4;      // cannot happen: cap is sized from the geometry (+8); the last 4 level slots carry phase cycles
	for (int k = tid; k < n_out; k += DM_THREADS) {
		const uint64_t t = (uint64_t)rs_phase + (uint64_t)k * T.rs_step;
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
		if (tid < D_RS_TAPS - 1) nh = (n_in - 1 - tid >= 0) ? in[n_in - 1 - tid] : a.rs_hist[tid - n_in];
		__syncthreads();
		if (tid < D_RS_TAPS - 1) a.rs_hist[tid] = nh;
		if (tid == 0) {
			S.rs_phase = (uint32_t)((uint64_t)rs_phase + (uint64_t)n_out * T.rs_step - total);
			if (TAPS) { io.tap_counts[0] = n_out; io.tap_counts[1] = 0; }
		}
	}
	__syncthreads();             // `in` (staged in the space of agc + mf) is dead from here on
	if (n_out < 1) return 0;

	unsigned long long tP0 = __builtin_amdgcn_s_memtime();
	// ---- the three-wave pipeline over chunks of DM_CHUNK samples.  Every wave runs its OWN copy of the step loop (same number
	// of barriers, same mailbox reads), so that only its own stage's state is live in its registers.
	unsigned long long busy = 0;
	int nsym = 0;
#ifdef HFDL_DM_PROBE
	unsigned long long cr_probe[3] = { 0, 0, 0 };
#endif
	if (wave == 0) {
		float agc_g = S.agc_g, agc_y2 = S.agc_y2;
		PipeProgress pp;
		for (int step = 0; pp.s3_done < n_out; step++) {
			int *mb = sh.mbox + 4 * (step & 1);
			int to = pp.mf_ready;
			if (pp.mf_ready < n_out) {
				to = pp.mf_ready + DM_CHUNK < n_out ? pp.mf_ready + DM_CHUNK : n_out;
				agc_mf_chunk(agc_g, agc_y2, a, T, io, pp.mf_ready, to, lane);
			}
			if (lane == 0) mb[0] = to;
			__syncthreads();
			pp.read(mb);
		}
		cf nh;
		if (lane < D_MF - 1) nh = (n_out - 1 - lane >= 0) ? io.agc[n_out - 1 - lane] : a.mf_hist[lane - n_out];
		__builtin_amdgcn_wave_barrier();
		if (lane < D_MF - 1) a.mf_hist[lane] = nh;
		if (lane == 0) { S.agc_g = agc_g; S.agc_y2 = agc_y2; }
	} else if (wave == 1) {
		SymsyncRegs ss;
		symsync_load(ss, S, a, io, lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		PipeProgress pp;
		for (int step = 0; pp.s3_done < n_out; step++) {
			int *mb = sh.mbox + 4 * (step & 1);
			const unsigned long long tb = TAPS ? __builtin_amdgcn_s_memtime() : 0ull;
			if (pp.restart) symsync_restart(ss, sh, pp.ss_ready);
			int to = pp.ss_ready + DM_CHUNK < pp.mf_ready ? pp.ss_ready + DM_CHUNK : pp.mf_ready;
			// never more than two chunks ahead of what wave 2 has finished: the output ring holds that much (in step, wave 2 is exactly one
			// chunk behind and this never binds; it does when a timing loop far off its rate makes wave 2 cut its chunks short)
			if (to > pp.s3_done + 2 * DM_CHUNK) to = pp.s3_done + 2 * DM_CHUNK;
			if (to > pp.ss_ready) {
				if (__builtin_expect(pp.ss_ready - (D_SS_TAPS - 1) < ss.valid_from, 0)) symsync_chunk<true>(ss, T, io, sh, pp.ss_ready, to, lane);
				else symsync_chunk<false>(ss, T, io, sh, pp.ss_ready, to, lane);
			} else to = pp.ss_ready;
			if (lane == 0) mb[1] = to;
			if (TAPS) busy += __builtin_amdgcn_s_memtime() - tb;
			__syncthreads();
			pp.read(mb);
		}
		if (pp.restart) symsync_restart(ss, sh, n_out);      // a reset during the block's last sample
		symsync_store(ss, S, a, io, n_out, lane);
	} else {
		CarrierRegs cr;
		ChanScalars s3 = S;               // wave 2's working copy: it owns every field but the resampler / AGC / timing-loop ones
		s3.ev_flags = 0;
		carrier_load(cr, s3, a, T, lane);
		PipeProgress pp;
		for (int step = 0; pp.s3_done < n_out; step++) {
			int *mb = sh.mbox + 4 * (step & 1);
			const unsigned long long tb = TAPS ? __builtin_amdgcn_s_memtime() : 0ull;
			int to = pp.s3_done + DM_CHUNK < pp.ss_ready ? pp.s3_done + DM_CHUNK : pp.ss_ready;
			int reset_at = -1;
			if (to > pp.s3_done) reset_at = carrier_chunk<TAPS>(cr, s3, a, T, io, sh, pp.s3_done, to, nsym, lane); else to = pp.s3_done;
			if (lane == 0) { mb[2] = reset_at >= 0 ? reset_at + 1 : to; mb[3] = reset_at >= 0; }
			if (TAPS) busy += __builtin_amdgcn_s_memtime() - tb;
			__syncthreads();
			pp.read(mb);
		}
		carrier_store(cr, a, lane);
#ifdef HFDL_DM_PROBE
		cr_probe[0] = cr.pa; cr_probe[1] = cr.pb; cr_probe[2] = cr.pc;
#endif
		if (lane == 0) {
			// every field wave 2 owns (the timing-loop scalars it touched through symsync_reset() are wave 1's)
			S.phi = s3.phi; S.dphi = s3.dphi; S.err = s3.err;
			S.eq_x2sum = s3.eq_x2sum; S.eq_count = s3.eq_count; S.eq_full = s3.eq_full; S.eq_head = 0;
			S.bits_hi = s3.bits_hi; S.bits_lo = s3.bits_lo;
			S.training_n = s3.training_n; S.data_n = s3.data_n; S.use_data = s3.use_data; S.data_slot = s3.data_slot;
			S.symbol_cnt = s3.symbol_cnt; S.sample_cnt = s3.sample_cnt;
			S.s_state = s3.s_state; S.fr_state = s3.fr_state; S.cur_arity = s3.cur_arity;
			S.symbols_wanted = s3.symbols_wanted; S.T_idx = s3.T_idx;
			S.bitmask = s3.bitmask; S.symsync_out_idx = s3.symsync_out_idx; S.nf_clk = s3.nf_clk;
			S.frame_symbol_cnt = s3.frame_symbol_cnt; S.signal_level = s3.signal_level;
			S.noise_floor = s3.noise_floor;
			// (the fields only the framer's rare transitions touch were updated in place: on_symbol()'s `c`)
			S.ev_flags = 0;
			if (TAPS) io.tap_counts[1] = nsym;
		}
	}
	unsigned long long tP1 = __builtin_amdgcn_s_memtime();
	if (TAPS) {
		__syncthreads();
		for (int k = tid; k < n_out; k += DM_THREADS) {
			io.tap_resampled[k] = io.rs[k];
			io.tap_mf[k] = io.mf[k];
			io.tap_level[k] = io.lvl[k];
		}
		// phase cycles: resampler, the whole pipelined phase (wall), wave 1 busy, wave 2 busy
		if (tid == 0) { io.tap_level[io.cap - 4] = (float)(tP0 - tR0); io.tap_level[io.cap - 3] = (float)(tP1 - tP0); }
		if (wave == 1 && lane == 0) io.tap_level[io.cap - 2] = (float)busy;
		if (wave == 2 && lane == 0) io.tap_level[io.cap - 1] = (float)busy;
#ifdef HFDL_DM_PROBE
		__syncthreads();
		if (wave == 2 && lane == 0) { io.tap_level[io.cap - 4] = (float)cr_probe[0]; io.tap_level[io.cap - 3] = (float)cr_probe[1]; io.tap_level[io.cap - 2] = (float)cr_probe[2]; }
#endif
	}
	return n_out;
}

}  // namespace hfdl
