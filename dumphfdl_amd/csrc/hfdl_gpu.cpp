// hfdl_gpu.cpp -- C-ABI shim of libhfdl_gpu.so: owns device memory, plans, streams; launches the gfx950 kernels.
// The only C++ translation unit a C host ever sees is through include/hfdl_gpu.h (extern "C", plain pointers).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/hfdl_gpu.h"
#include "kernels.h"
#include "planner.h"
#include "demod.h"

using namespace hfdl;

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	return fail(HFDL_GPU_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

extern "C" const char *hfdl_gpu_last_error(void) { return g_err; }

// kernel time of the last stage-level entry point called by this thread (HIP events around its launches, copies excluded)
static thread_local double g_stage_ms = 0.0;
extern "C" double hfdl_gpu_last_stage_ms(void) { return g_stage_ms; }

// brackets the launches of a stage entry point with events on the null stream
struct StageTimer {
	hipEvent_t e0 = nullptr, e1 = nullptr;
	StageTimer() { g_stage_ms = 0.0; if (hipEventCreate(&e0) != hipSuccess) e0 = nullptr; if (hipEventCreate(&e1) != hipSuccess) e1 = nullptr; if (e0) (void)hipEventRecord(e0, nullptr); }
	void stop() { if (e0 && e1) { (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1); float ms = 0; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) g_stage_ms = ms; } }
	~StageTimer() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

extern "C" int hfdl_gpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

static int select_device(int device)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return fail(HFDL_GPU_ENODEV, "no HIP device visible: the HFDL front end has no CPU fallback");
	if (device < 0 || device >= n) return fail(HFDL_GPU_EINVAL, "device %d out of range (%d visible)", device, n);
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(HFDL_GPU_ENODEV, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
	return 0;
}

// ---------------------------------------------------------------- FFT plan

static int ilog2(int x) { int l = 0; while ((1 << l) < x) l++; return l; }

struct HostFftPlan {
	FftPlan p{};
	float2 *d_tw[3] = { nullptr, nullptr, nullptr };
	int build(int n);
	void release() { for (auto &t : d_tw) { if (t) (void)hipFree(t); t = nullptr; } }
};

static int upload_twiddles(int r, float2 **out)
{
	std::vector<float2> h((size_t)r);
	for (int t = 0; t < r; t++) {
		double a = -2.0 * M_PI * (double)t / (double)r;
		h[t] = make_float2((float)std::cos(a), (float)std::sin(a));
	}
	HIP_TRY(hipMalloc(out, sizeof(float2) * (size_t)r));
	HIP_TRY(hipMemcpy(*out, h.data(), sizeof(float2) * (size_t)r, hipMemcpyHostToDevice));
	return 0;
}

int HostFftPlan::build(int n)
{
	int logn = ilog2(n);
	if ((1 << logn) != n || logn < 9 || logn > 24) return fail(HFDL_GPU_ERANGE, "fft size %d unsupported (need 2^9..2^24)", n);
	// balanced split, largest radix last-but-one; every radix <= 256 so a 16-column tile fits 32 KiB of LDS
	int l1 = (logn + 2) / 3, l2 = (logn - l1 + 1) / 2, l3 = logn - l1 - l2;
	p.n = n; p.logn = logn;
	p.l1 = l1; p.l2 = l2; p.l3 = l3;
	p.r1 = 1 << l1; p.r2 = 1 << l2; p.r3 = 1 << l3;
	int rc;
	if ((rc = upload_twiddles(p.r1, &d_tw[0]))) return rc;
	if ((rc = upload_twiddles(p.r2, &d_tw[1]))) return rc;
	if ((rc = upload_twiddles(p.r3, &d_tw[2]))) return rc;
	p.tw1 = d_tw[0]; p.tw2 = d_tw[1]; p.tw3 = d_tw[2];
	return 0;
}

// ---------------------------------------------------------------- front end

struct hfdl_gpu_frontend {
	int device = 0;
	hipStream_t stream = nullptr;       // A: forward FFTs of the half being filled, then ONE fold and ONE inverse FFT / NCO launch per half
	hipStream_t stream_b = nullptr;     // B: demodulator launches of half k-1, beside the forward FFTs and the fold of half k
	hipStream_t stream_d = nullptr;     // D: burst decoders + PDU snapshots, off the demodulators' critical path (== stream_b only in a laboratory A/B run)
	bool own_decode_stream = false;
	static constexpr int MAX_HALF = FOLD_MAX_BLOCKS;      // blocks per half at most: what one fold launch can take (32)
	static constexpr int MAX_STAGE = HFDL_GPU_PREFETCH_MAX + 1;      // staging buffers for host input at most: uploads run at most 17 blocks ahead
	hipEvent_t ev_dm[2][MAX_HALF] = {};              // demodulator launch j of the half in buffer 0 / 1 done (the decoder may start)
	hipEvent_t ev_dm_cur[2] = { nullptr, nullptr };  // LAST demodulator launch of that half done (chan_out free): an ev_dm, or (timing on) the stop event of a timed pair
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_dmt;      // timed demodulator launches not yet read
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_fftt, ev_ifftt, ev_dect;      // ... forward FFTs (first pass start -> last pass stop), inverse FFT / NCO launches, burst decoders
	double fft_ms = 0, ifft_ms = 0, decode_ms = 0;
	int64_t fft_timed = 0, ifft_timed = 0, decode_timed = 0;
	hipEvent_t ev_chan_cur[2] = { nullptr, nullptr };       // "channelizer output of this half ready": ev_chan, or (timing on) the stop event of the timed inverse FFT
	hipEvent_t ev_fft_cur = nullptr;                        // what the held-back demodulators wait for: ev_fft, or the stop event of a timed forward FFT
	double demod_ms = 0;
	int64_t demod_launches = 0, demod_timed_blocks = 0;
	hipStream_t stream_c = nullptr;     // C: host -> device copies into the staging ring, up to a half ahead of the blocks that compute
	hipStream_t stream_f = nullptr;     // F: forward FFTs of the half being filled, beside the fold of the half before (== stream when HFDL_GPU_FFT_STREAM=0)
	bool fft_own_stream = false;
	hipEvent_t ev_spec[2] = { nullptr, nullptr };    // newest forward FFT of the half in spectrum set 0 / 1 done (rides on its last pass)

	hipEvent_t ev_chan[2] = { nullptr, nullptr }, ev_demod[2] = { nullptr, nullptr };
	// Host input goes through a RING of n_stage = half_blocks + 2 staging buffers in HBM: host block j is copied (stream C) into buffer
	// j % n_stage, which the forward FFT's first pass of block j - n_stage has finished reading -- that pass runs BEFORE the fold of its
	// half, so uploads run a whole half ahead and never sit behind the 3 ms fold (with two buffers, upload k+2 waited for FFT k, which
	// waited for the fold of the half before: the link idled a third of the time).
	int n_stage = 0;
	hipEvent_t ev_stage_ready[MAX_STAGE] = {};  // copy of the host block in this buffer done: what its forward FFT and input_done_upto() wait for
	hipEvent_t ev_stage_free[MAX_STAGE] = {};   // pass 1 of the forward FFT that read this buffer done (rides on that dispatch): the copy stream may refill it
	uint64_t host_blocks = 0;           // host blocks whose copy has been queued (pushed or prefetched)
	uint64_t host_pushed = 0;           // ... of which this many have been pushed (or cancelled): the rest wait in the prefetch queue, oldest first
	const void *pf_ptr[MAX_STAGE] = {}; // prefetch queue entry of host block j at [j % n_stage]: the host pointer ...
	int pf_fmt[MAX_STAGE] = {};         // ... and its sample format
	int32_t sample_rate = 0, centerfreq = 0, decimation = 0;
	float tbw = 0;
	Plan plan{};                       // shift = 0 geometry (src/fft.c:70-86)
	Geometry geo{};
	HostFftPlan fft;
	std::vector<int32_t> freqs;
	std::vector<ChanConst> cc;
	float2 *d_hist[2] = { nullptr, nullptr }, *d_work = nullptr, *d_spec = nullptr, *d_taps = nullptr, *d_partial = nullptr;
	float2 *d_tw_m = nullptr, *d_stage[MAX_STAGE] = {};
	// Channelizer output, double-buffered between stream A and stream B in two HALVES of `half_blocks` blocks each:
	// [2][half_blocks][nch][outs].  The forward FFT of a block is queued when it is pushed; the fold and the inverse FFTs run when a
	// half is closed (full, or a sync / poll found it part-filled): ONE pass over the filter taps serves up to `fold_nb` blocks.
	// The demodulator then takes the half `batch` blocks per launch while the channelizer fills the other half.
	float2 *d_chan_all = nullptr;
	int *d_cnt_all = nullptr;           // [2][half_blocks][nch] outputs per channel of each block
	int batch = 1;                      // blocks per demodulator launch (see pick_demod_batch)
	int fold_nb = 1;                    // blocks per fold launch (see pick_fold_batch)
	int half_blocks = 1;                // slots per half: a multiple of fold_nb, at least `batch`
	// How many blocks close the half being filled.  A pipeline that starts empty closes its first half at `half_first` blocks (16 where a
	// half holds 32): the first fold launch is the sixteen-column form and the demodulators start 3 ms earlier; once a half has been
	// closed BY FILLING -- the caller pushes faster than it collects -- the next ones take all `half_blocks`.  Any sync / poll that closes
	// a half early (a drain) starts over.  Results do not depend on where the halves are cut (test_fold_batching_changes_nothing).
	int half_first = 1, half_target = 1;
	int cur_half = 0, batch_fill = 0;   // the half being filled and the blocks already in it (forward FFT queued, fold not yet)
	int last_slot = 0;                  // slot (half * half_blocks + index) of the newest channelized block: what HFDL_GPU_TAP_CHAN_OUT reads
	int last_index = 0;                 // its index inside the half: spectrum / phasor-table slot of the newest block
	float2 *chan_slot(int slot) const { return d_chan_all + (size_t)slot * (size_t)geo.nch * (size_t)geo.outs; }
	int *cnt_slot(int slot) const { return d_cnt_all + (size_t)slot * (size_t)geo.nch; }
	// spectra, NCO phasor tables and carried-state snapshots of a half: two sets (the forward FFTs of half k+1 fill one while the fold
	// and inverse FFTs of half k read the other); `set` = cur_half of the half they belong to
	int last_set = 0;                   // set of the newest channelized half: what the taps read
	float2 *spec_slot(int set, int i) const { return d_spec + ((size_t)set * (size_t)half_blocks + (size_t)i) * (size_t)geo.n; }
	float2 *ph_slot(int set, int i) const { return d_ph + ((size_t)set * (size_t)half_blocks + (size_t)i) * ph_stride(); }
	NcoState *snap_slot(int set, int i) const { return d_nco_snap + ((size_t)set * (size_t)half_blocks + (size_t)i) * (size_t)geo.nch; }
	size_t partial_stride() const { return (size_t)geo.nch * (size_t)geo.slices * (size_t)geo.m; }
	size_t ph_stride() const { return (size_t)geo.nch * (size_t)geo.outs; }
	ChanConst *d_cc = nullptr;
	int2 *d_win = nullptr;              // pruned fold: window of quads of alias rows per octet (kernels.h Geometry::fold_win)
	double prune_tol = 0.0;             // HFDL_GPU_FOLD_PRUNE: share of a filter's energy (as an amplitude ratio) the skipped alias rows may hold; 0 = fold every row
	int fold_rows_max = 0;              // the longest row window (0: every row is folded)
	NcoState *d_nco = nullptr;          // [nch] carried NCO state, owned by the forward FFT's rider workgroups (kernels.h NcoJob)
	NcoState *d_nco_snap = nullptr;     // [half_blocks][nch] the state each block of the half starts from
	float2 *d_ph = nullptr, *d_ph_cont = nullptr;      // [half_blocks] NCO phasor tables [outs][nch] and the riders' segment hand-over [nch]
	size_t stage_cap[MAX_STAGE] = {};
	bool fold_bound = false;            // many channels: the fold bounds the block and the demodulator launches of a half are placed under the NEXT half's fold
	Demod demod;
	// fold timing
	bool timing = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;      // timing events made by reset_timers(), outside any timed region
	std::vector<int> ev_blocks;         // blocks folded between each pair of `ev`
	double fold_ms = 0;
	int64_t fold_launches = 0, fold_timed_blocks = 0, fold_last_blocks = 0;
	int64_t fold_shapes[FOLD_MAX_BLOCKS + 1] = {};   // timed fold launches by block count
	double fold_shape_ms[FOLD_MAX_BLOCKS + 1] = {};  // ... and their kernel time
	hipEvent_t ev_first_fold = nullptr;  // start of the first timed fold since reset_timers: anchor of the steady-state step period
	double span_ms = 0;                 // first timed fold start -> last timed fold start
	uint64_t blocks = 0;
	FftOutLayout tap_layout;
	int pending_demod_buf = -1;         // half whose demodulator launches are held back until the next half's forward FFTs are queued ...
	int pending_demod_nblk = 0;         // ... and the blocks in it
	bool frames_wait_on_a = false;      // stream A has waited for the frame queue the next demodulator launch reuses
	hipEvent_t ev_fft = nullptr;
	int demod_buf = -1;                 // half / snapshot slot of the newest demodulator launch
	int prev_demod_buf = -1;            // ... and of the one before it
};

// a start / stop event pair for a timed launch: from the pool reset_timers() filled (no event creation between the timed launches of a
// bench run), made on demand when a long run without a draining call has used the pool up -- a timed launch is never silently untimed
static bool take_timer_pair(hfdl_gpu_frontend *fe, std::pair<hipEvent_t, hipEvent_t> &e)
{
	if (!fe->ev_pool.empty()) {
		e = fe->ev_pool.back();
		fe->ev_pool.pop_back();
		return true;
	}
	if (hipEventCreate(&e.first) != hipSuccess) return false;
	if (hipEventCreate(&e.second) != hipSuccess) { (void)hipEventDestroy(e.first); return false; }
	return true;
}

static void frontend_free(hfdl_gpu_frontend *fe)
{
	if (!fe) return;
	(void)hipSetDevice(fe->device);
	if (fe->stream) (void)hipStreamSynchronize(fe->stream);
	if (fe->stream_b) (void)hipStreamSynchronize(fe->stream_b);
	if (fe->own_decode_stream && fe->stream_d) (void)hipStreamSynchronize(fe->stream_d);
	if (fe->stream_c) (void)hipStreamSynchronize(fe->stream_c);
	if (fe->fft_own_stream && fe->stream_f) (void)hipStreamSynchronize(fe->stream_f);
	for (hipEvent_t e : fe->ev_spec) if (e) (void)hipEventDestroy(e);
	for (int i = 0; i < 2; i++)
		for (hipEvent_t e : { fe->ev_chan[i], fe->ev_demod[i] }) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : fe->ev_stage_ready) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : fe->ev_stage_free) if (e) (void)hipEventDestroy(e);
	for (auto &h : fe->ev_dm) for (hipEvent_t e : h) if (e) (void)hipEventDestroy(e);
	for (auto &e : fe->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
	for (auto &e : fe->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
	for (auto &e : fe->ev_dmt) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
	for (auto *v : { &fe->ev_fftt, &fe->ev_ifftt, &fe->ev_dect }) for (auto &e : *v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
	if (fe->ev_fft) (void)hipEventDestroy(fe->ev_fft);
	if (fe->ev_first_fold) (void)hipEventDestroy(fe->ev_first_fold);
	fe->demod.release();
	fe->fft.release();
	void *ptrs[] = { fe->d_hist[0], fe->d_hist[1], fe->d_work, fe->d_spec, fe->d_taps, fe->d_partial, fe->d_chan_all, fe->d_tw_m,
		fe->d_cc, fe->d_nco, fe->d_nco_snap, fe->d_ph, fe->d_ph_cont, fe->d_cnt_all, fe->d_win };
	for (void *p : ptrs) if (p) (void)hipFree(p);
	for (float2 *p : fe->d_stage) if (p) (void)hipFree(p);
	if (fe->stream) (void)hipStreamDestroy(fe->stream);
	if (fe->own_decode_stream && fe->stream_d) (void)hipStreamDestroy(fe->stream_d);
	if (fe->stream_b) (void)hipStreamDestroy(fe->stream_b);
	if (fe->stream_c) (void)hipStreamDestroy(fe->stream_c);
	if (fe->fft_own_stream && fe->stream_f) (void)hipStreamDestroy(fe->stream_f);
	delete fe;
}

extern "C" void hfdl_gpu_frontend_destroy(hfdl_gpu_frontend *fe) { frontend_free(fe); }

static int pick_slices(int nch, int rows)
{
	// Slices of alias rows per (channel group, bin group): each slice is a workgroup of its own and leaves a partial sum that the inverse
	// FFT adds up.  A CU holds ONE matrix-pipe fold workgroup at a time (one 384 / 420-register wave per SIMD), so every workgroup
	// generation pays its dispatch, its first loads and its stores with an idle matrix pipe: as few and as long-lived workgroups as fill
	// the chip.  256 channels need no slicing (cfg3: 2048 workgroups of 512 quads at one slice; against round 1's rule of
	// channels x slices >= 1024 -- four slices -- the 32-block fold takes 5.7 instead of 6.8 ms alone, 0.203 instead of 0.22 ms per block
	// in the pipeline, and a quarter of the partial sums are written and read back: profiles/r06_experiments.md); fewer channels are
	// sliced until channels x slices >= 256, a slice keeping at least 8 alias rows.
	int s = 1;
	while (s * 2 <= rows / 8 && nch * s < 256) s *= 2;
	return s;
}

// Create-time configuration from the environment (include/hfdl_gpu.h documents every name).  Read at every create and never cached:
// there is no function-local static to race on when front ends are created from several threads.  The A/B switches of the
// measurement scripts exist in the laboratory build only (-DHFDL_LAB, libhfdl_gpu_lab.so).
static long env_long(const char *name, long lo, long hi, long otherwise)
{
	const char *e = getenv(name);
	if (!e || !*e) return otherwise;
	char *end = nullptr;
	const long v = strtol(e, &end, 10);
	return (end != e && v >= lo && v <= hi) ? v : otherwise;
}

// Blocks per demodulator launch.  Every launch pays fixed costs: the barrier packet in front of it (~11 us), ~25 KiB of tables and
// state staged into LDS and written back, and two chunks of pipeline fill and drain -- ~40 us against ~210 us of recurrence per
// cfg2 block.  When blocks arrive faster than they are demodulated (file replay, catching up) consecutive blocks are therefore handed
// to ONE launch, which treats them as one longer stretch of samples -- the per-channel state is carried sample by sample, so the
// result is that of block-by-block processing.  A caller that waits for its PDUs after every block (live input: poll / sync) still
// gets a launch per block: a partial batch is launched by any call that needs the results.  Bounds: the LDS (Demod::init keeps what
// fits: 30 B per sample, three cfg3 blocks), and one second of signal -- less than half the shortest frame (2.34 s), so that a channel
// finishes at most one frame per launch (frame queue: one entry per channel; two data slots).
static int pick_demod_batch(const hfdl_gpu_frontend *fe)
{
	const double block_s = (double)fe->plan.input_size / (double)fe->sample_rate;
	int want = (int)std::floor(1.0 / block_s);
	want = std::max(1, std::min(8, want));
	// Where the fold bounds the block the demodulator workgroups (one per channel, ~one per CU) must stay CO-RESIDENT with the fold's
	// (34 KiB of LDS per workgroup in the thirty-two-column form) and a forward-FFT tile: three cfg3 blocks per launch take 117 KiB of a
	// CU's 160 KiB since the timing-recovery outputs go through a ring (round 6; round 5: two blocks, 118 KiB).  One block more and the
	// kernels take turns (measured in round 5 at 159 KiB: a demodulator launch beside a 4.8 ms fold took 5.5 ms, profiles/r05_experiments.md).
	if (fe->fold_bound) want = std::min(want, 3);
	return (int)env_long("HFDL_GPU_DEMOD_BATCH", 1, 8, want);       // 1 = a launch per block
}

// Blocks per fold launch.  The filter taps are 99.9 % of a block's bytes on the fold-bound geometries (cfg3: 16 GiB of taps against
// a 64 MiB spectrum) and they are the same for every block: when blocks are pushed faster than they are collected (file replay,
// catching up, the bench) the spectra of up to `fold_nb` consecutive blocks are folded in ONE pass over the taps on the matrix pipe
// (fold_kernels.hip).  Every block's sums are bit-identical to a launch of its own (fixed FMA chain per bin); a caller that polls
// or syncs after every block (live input) still gets one launch per block: a sync / poll closes the half as it is.
static_assert(HFDL_GPU_FOLD_BATCH_MAX == FOLD_MAX_BLOCKS, "include/hfdl_gpu.h and kernels.h name the same limit");
static int pick_fold_batch(const hfdl_gpu_frontend *fe)
{
	// 32 where the fold bounds the block (128 channels and more): two column groups of the sixteen-column matrix instruction per loaded tap
	// operand.  A launch costs about its matrix time plus its memory time (fold_kernels.hip, profiles/r06_experiments.md), so a block's share
	// shrinks with the blocks per byte of taps: 0.20 ms per block at 32 against 0.245 at 16 in the pipeline.  The
	// first half after a drain closes at 16 (half_first).  Where the demodulator bounds the block (fewer than 128 channels: the taps are a
	// few hundred MiB and a fold launch takes 0.2 ms whatever it folds) a long half only adds fill, drain and latency: 8, as in round 4
	// (cfg2: 0.1545 against 0.1595 ms per block over 256 blocks)
	return (int)env_long("HFDL_GPU_FOLD_BATCH", 1, hfdl_gpu_frontend::MAX_HALF, fe->fold_bound ? 32 : 8);       // 1 = a pass over the taps per block
}

static double env_double(const char *name, double lo, double hi, double otherwise)
{
	const char *e = getenv(name);
	if (!e || !*e) return otherwise;
	char *end = nullptr;
	const double v = strtod(e, &end);
	return (end != e && v >= lo && v <= hi) ? v : otherwise;
}

// The pruned fold (HFDL_GPU_FOLD_PRUNE = tolerance).  A channel's filter is a band-pass M / 2 bins wide with a Hamming-window stop band:
// of the p = N / M alias rows the fold adds up, all but the few around the pass band hold taps below fp32 resolution of the sum (cfg3:
// rows 32 or more from the pass band hold 3.7e-8 of the filter's energy as an amplitude ratio -- less than half an ulp; DESIGN.md section
// 9).  From the taps themselves: per channel the smallest window of rows around the pass band outside which less than tol^2 of the
// filter's energy lies; the workgroup of an octet folds the circular hull of its eight channels' windows in quads of rows (what one
// matrix instruction takes), rounded up to whole look-ahead groups of two quads.
static int build_fold_windows(hfdl_gpu_frontend *fe)
{
	Geometry &g = fe->geo;
	const int p = g.pre, npad = g.nch_pad, nch = g.nch;
	DevBuf d_en;
	HIP_TRY(d_en.alloc(sizeof(float) * (size_t)p * (size_t)npad));
	HIP_TRY(hipMemsetAsync(d_en.p, 0, sizeof(float) * (size_t)p * (size_t)npad, fe->stream));
	launch_tap_row_energy(fe->d_taps, g, d_en.as<float>(), fe->stream);
	std::vector<float> en((size_t)p * (size_t)npad);
	HIP_TRY(hipMemcpyAsync(en.data(), d_en.p, sizeof(float) * en.size(), hipMemcpyDeviceToHost, fe->stream));
	HIP_TRY(hipStreamSynchronize(fe->stream));
	// per channel: the window grows from the row that holds the most energy, towards the richer neighbour, until the rows outside
	// hold less than tol^2 of the total (rows picked by energy alone would scatter: the fp32 transform that made the taps left its
	// rounding noise in every row, and the largest noise rows lie anywhere)
	std::vector<std::vector<char>> keep((size_t)npad, std::vector<char>((size_t)p, 0));
	for (int c = 0; c < nch; c++) {
		double tot = 0;
		int peak = 0;
		for (int r = 0; r < p; r++) { tot += en[(size_t)r * npad + c]; if (en[(size_t)r * npad + c] > en[(size_t)peak * npad + c]) peak = r; }
		auto e_at = [&](int r) { return (double)en[(size_t)((r % p + p) % p) * npad + c]; };
		int lo = peak, hi = peak;                         // window [lo, hi], indices unwrapped
		double left = tot - e_at(peak);
		while (hi - lo + 1 < p && left > fe->prune_tol * fe->prune_tol * tot) {
			if (e_at(lo - 1) > e_at(hi + 1)) left -= e_at(--lo); else left -= e_at(++hi);
		}
		for (int r = lo; r <= hi; r++) keep[(size_t)c][(size_t)((r % p + p) % p)] = 1;
	}
	auto hull = [&](int c0, int c1) {                   // circular hull of the rows kept by channels [c0, c1), in QUADS of rows (first quad, count)
		const int nq = p / 4;
		std::vector<char> any((size_t)nq, 0);
		int kept = 0;
		for (int c = c0; c < c1; c++) for (int r = 0; r < p; r++) if (keep[(size_t)c][(size_t)r] && !any[(size_t)(r >> 2)]) { any[(size_t)(r >> 2)] = 1; kept++; }
		if (kept == 0) return make_int2(0, 2);           // channels that only fill the octet up: zero taps, any two quads
		int best_len = 0, best_end = 0;                  // the longest circular run of quads nobody keeps
		for (int q = 0; q < nq; q++) {
			if (any[(size_t)q] || !any[(size_t)((q + nq - 1) % nq)]) continue;      // q = first quad of a gap
			int len = 0;
			while (len < nq && !any[(size_t)((q + len) % nq)]) len++;
			if (len > best_len) { best_len = len; best_end = (q + len) % nq; }
		}
		int count = nq - best_len;
		count = std::min(nq, (count + 1) & ~1);          // whole look-ahead groups of two quads
		return make_int2(best_len ? best_end : 0, count);
	};
	const int noct = npad / 8;
	std::vector<int2> w((size_t)noct);
	fe->fold_rows_max = 0;
	for (int i = 0; i < noct; i++) { w[(size_t)i] = hull(8 * i, 8 * i + 8); fe->fold_rows_max = std::max(fe->fold_rows_max, 4 * w[(size_t)i].y); }
	HIP_TRY(hipMalloc(&fe->d_win, sizeof(int2) * w.size()));
	HIP_TRY(hipMemcpy(fe->d_win, w.data(), sizeof(int2) * w.size(), hipMemcpyHostToDevice));
	g.fold_win = fe->d_win;
	return 0;
}

static int build_taps(hfdl_gpu_frontend *fe)
{
	const Plan &pl = fe->plan;
	const int nch = (int)fe->freqs.size();
	const size_t n = (size_t)pl.n;
	// time-domain taps on the host (exact reference arithmetic), one worker per hardware thread
	std::vector<std::complex<float>> host((size_t)nch * (size_t)pl.taps_length);
	fe->cc.resize((size_t)nch);
	unsigned nthreads = std::max(1u, std::min((unsigned)nch, std::thread::hardware_concurrency()));
	// several front ends created at once on one host (one process per GPU): share the cores
	nthreads = std::min(nthreads, (unsigned)env_long("HFDL_GPU_HOST_THREADS", 1, 1 << 16, (long)nthreads));
	std::atomic<int> next{0};
	std::atomic<int> bad{0};
	auto work = [&]() {
		std::vector<float> lp;
		float lp_cut = -1.f;
		for (;;) {
			int c = next.fetch_add(1);
			if (c >= nch) break;
			// src/hfdl.c:476: shift relative to the SSB carrier 1440 Hz above the channel frequency
			float shift = (float)(fe->centerfreq - (fe->freqs[c] + 1440)) / (float)fe->sample_rate;
			Plan cp;
			if (!plan_block(cp, fe->tbw, fe->decimation, shift)) { bad++; continue; }
			ChanConst k{};
			k.offsetbin = cp.offsetbin;
			k.nco_sindelta = cp.sindelta; k.nco_cosdelta = cp.cosdelta; k.nco_rate = cp.rate;
			k.frequency = fe->freqs[c];
			fe->cc[c] = k;
			float half_bw = 0.5f / fe->decimation;
			design_bandpass(host.data() + (size_t)c * pl.taps_length, pl.taps_length, (-shift) - half_bw, (-shift) + half_bw, lp, lp_cut);
		}
	};
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(work);
	work();
	for (auto &t : pool) t.join();
	if (bad) return fail(HFDL_GPU_EINVAL, "fastddc planning failed for %d channel(s)", (int)bad);

	// frequency-domain taps on the device: zero-pad to N, forward FFT, fftshift (src/fastddc.c:231-240)
	DevBuf pad;
	HIP_TRY(pad.alloc(sizeof(float2) * n));
	float2 *d_pad = pad.as<float2>();
	HIP_TRY(hipMemsetAsync(d_pad, 0, sizeof(float2) * n, fe->stream));
	if (fe->geo.nch_pad > nch)          // the channels that fill the last group of the interleaved layout up: all-zero taps
		HIP_TRY(hipMemsetAsync(fe->d_taps, 0, sizeof(float2) * n * (size_t)fe->geo.nch_pad, fe->stream));
	for (int c = 0; c < nch; c++) {
		HIP_TRY(hipMemcpyAsync(d_pad, host.data() + (size_t)c * pl.taps_length, sizeof(float2) * (size_t)pl.taps_length,
				hipMemcpyHostToDevice, fe->stream));
		// the last pass writes the channel's filter straight into the matrix-operand layout (kernels.h tap_index_f)
		FftOutLayout lay = fe->tap_layout;
		lay.chan = c;
		float2 *dst = lay.kind == TAPL_PLAIN ? fe->d_taps + (size_t)c * (size_t)fe->geo.tap_chan_stride : fe->d_taps;
		launch_fft_forward(fe->fft.p, nullptr, d_pad, SFMT_CF32, 0, nullptr, fe->d_work, dst, true, fe->stream, lay);
	}
	HIP_TRY(hipStreamSynchronize(fe->stream));
	HIP_TRY(hipGetLastError());
	return 0;
}

extern "C" int hfdl_gpu_frontend_create(hfdl_gpu_frontend **out, int device, int32_t sample_rate, int32_t centerfreq,
		const int32_t *freqs, int32_t nch)
{
	if (!out || !freqs || nch <= 0) return fail(HFDL_GPU_EINVAL, "bad arguments");
	*out = nullptr;
	if (sample_rate < 5400) return fail(HFDL_GPU_EINVAL, "sample rate must be >= 5400 (src/main.c:638-641)");
	int rc = select_device(device);
	if (rc) return rc;
	auto *fe = new hfdl_gpu_frontend();
	fe->device = device;
	fe->sample_rate = sample_rate; fe->centerfreq = centerfreq;
	fe->decimation = fft_decimation_rate(sample_rate, 1800 * 3);
	fe->tbw = relative_transition_bw(sample_rate, 250);
	fe->freqs.assign(freqs, freqs + nch);
	for (int32_t f : fe->freqs) {
		// span check of src/main.c:214-226
		if (std::abs((int64_t)centerfreq - f) >= sample_rate / 2) {
			delete fe;
			return fail(HFDL_GPU_EINVAL, "channel %d Hz outside +-fs/2 of centre %d", f, centerfreq);
		}
	}
	if (!plan_block(fe->plan, fe->tbw, fe->decimation, 0.f)) { delete fe; return fail(HFDL_GPU_EINVAL, "fastddc planning failed"); }
	const Plan &pl = fe->plan;
	Geometry &g = fe->geo;
	g.n = pl.n; g.m = pl.m; g.pre = pl.pre; g.post = pl.post; g.scrap = pl.scrap; g.post_input_size = pl.post_input_size;
	g.overlap = pl.overlap; g.input_size = pl.input_size; g.outs = (pl.post_input_size + pl.post - 1) / pl.post + 1;
	g.nch = nch;
	// Filter taps row-major over channels: alias row r of every channel sits in one nch*M run, so the workgroups of all
	// channels, which walk the rows together, stream through a few moving windows of HBM instead of nch windows 8N bytes
	// apart (fold kernel 2.58 -> 2.48 ms on cfg3 and a tighter run-to-run spread, profiles/r01_experiments.md)
	g.tap_layout = (pl.m % 16) == 0 && (pl.pre % 8) == 0 ? TAPL_OCTET : TAPL_PLAIN;       // the matrix-pipe fold walks the alias rows four at a time, two such quads in flight
	{
		const int grp = tap_layout_group(g.tap_layout);
		g.nch_pad = (nch + grp - 1) / grp * grp;
	}
	g.tap_chan_stride = pl.m; g.tap_row_stride = (int64_t)g.nch_pad * pl.m;
	fe->tap_layout.kind = g.tap_layout;
	fe->tap_layout.row_log = ilog2(pl.m); fe->tap_layout.row_stride = g.tap_row_stride;
	g.slices = pick_slices(nch, pl.pre);
	// HFDL_GPU_FOLD_PRUNE=tol (0 < tol <= 1e-3; unset: every alias row is folded, the reference's sum term for term): fold only the
	// rows around each channel's pass band (build_fold_windows) -- one slice, the windows are the parallelism
#ifdef HFDL_LAB
	g.fold_tile = (int32_t)env_long("HFDL_GPU_FOLD_TILE", 0, 63, -1);
	{	// laboratory A/B: slices of alias rows per channel and bin (1, 2, 4 ...; a slice keeps at least 16 rows)
		const int sl = (int)env_long("HFDL_GPU_FOLD_SLICES", 1, 64, 0);
		if (sl > 0 && (sl & (sl - 1)) == 0 && pl.pre % sl == 0 && pl.pre / sl >= 16) g.slices = sl;
	}
#endif
	fe->prune_tol = g.tap_layout == TAPL_OCTET ? env_double("HFDL_GPU_FOLD_PRUNE", 1e-12, 1e-3, 0.0) : 0.0;
	if (fe->prune_tol > 0) g.slices = 1;
	g.rows_per_slice = pl.pre / g.slices;
	if (pl.m > 8192 || pl.m < 16) { delete fe; return fail(HFDL_GPU_ERANGE, "inverse FFT size %d unsupported", pl.m); }

#define FE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	int rc_ = fail(e_ == hipErrorOutOfMemory ? HFDL_GPU_ENOMEM : HFDL_GPU_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
	frontend_free(fe); return rc_; } } while (0)
#ifdef HFDL_LAB
	// Laboratory A/B (HFDL_GPU_CU_SPLIT=k, k = 2 / 4 / 8): the demodulator's stream on every k-th CU, the channelizer's stream on the others --
	// does a demodulator launch still execute twice the cycles while a fold runs, when no fold wave shares its SIMD?  CU i belongs to the
	// demodulator iff ((i >> 3) + i) % k == 0: an equal share of every XCD whether the mask counts XCD-major or XCD-interleaved.
	const int cu_split = (int)env_long("HFDL_GPU_CU_SPLIT", 2, 8, 0);
	uint32_t mask_a[8] = {}, mask_b[8] = {};
	for (int i = 0; i < 256; i++) {
		const bool demod_cu = cu_split && (((i >> 3) + i) % cu_split) == 0;
		(demod_cu ? mask_b : mask_a)[i >> 5] |= 1u << (i & 31);
	}
	if (cu_split) FE_TRY(hipExtStreamCreateWithCUMask(&fe->stream, 8, mask_a));
	else
#endif
	FE_TRY(hipStreamCreateWithFlags(&fe->stream, hipStreamNonBlocking));
	{
		// measured on MI355X: stream priority (hi/lo) and CU-masking of this stream change nothing beyond run-to-run
		// noise (profiles/r01_experiments.md), so a plain non-blocking stream is used
#ifdef HFDL_LAB
		if (cu_split) FE_TRY(hipExtStreamCreateWithCUMask(&fe->stream_b, 8, mask_b));
		else
#endif
		FE_TRY(hipStreamCreateWithFlags(&fe->stream_b, hipStreamNonBlocking));
	}
	FE_TRY(hipStreamCreateWithFlags(&fe->stream_c, hipStreamNonBlocking));
	{
		// Forward FFTs of the half being filled on a stream of their own, beside the fold of the half before (two sets of spectra / phasor
		// tables / state snapshots): measured on cfg3 in round 4 (profiles/r04_experiments.md) the passes then take their HBM share out of
		// the fold and the demodulators and the step gets slower, so the FFTs stay in front of the fold on stream A.  The switch lives in the
		// laboratory build.  (More than four busy streams also need GPU_MAX_HW_QUEUES > 4: two streams on one hardware queue run in turn.)
		fe->fft_own_stream = false;
#ifdef HFDL_LAB
		fe->fft_own_stream = env_long("HFDL_GPU_FFT_STREAM", 0, 1, 0) != 0;
#endif
		if (fe->fft_own_stream) FE_TRY(hipStreamCreateWithFlags(&fe->stream_f, hipStreamNonBlocking));
		else fe->stream_f = fe->stream;
		for (auto &e : fe->ev_spec) FE_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	{
		// The burst decoder of launch k only hands PDUs to the host; the demodulator of launch k+1 does not need it, and a long frame
		// ending in a block puts 0.3 - 1.2 ms of Viterbi in front of it: the decoder has its own stream (cfg2: +25 % in round 2; at 256
		// channels the round-4 timeline shows a 1.18 ms decoder launch serially ahead of a half's first demodulator).
		// fold_bound: with many channels the fold bounds the block; the demodulator launches of a half are then held back until the next
		// half's forward FFTs are queued (launch_demod) instead of starting at once.
		fe->fold_bound = nch >= 128;
		fe->own_decode_stream = true;
#ifdef HFDL_LAB
		fe->fold_bound = env_long("HFDL_GPU_FOLD_BOUND", 0, 1, fe->fold_bound ? 1 : 0) != 0;
		fe->own_decode_stream = env_long("HFDL_GPU_DECODE_STREAM", 0, 1, 1) != 0;
#endif
		if (fe->own_decode_stream) FE_TRY(hipStreamCreateWithFlags(&fe->stream_d, hipStreamNonBlocking));
		else fe->stream_d = fe->stream_b;
		fe->demod.separate_decode = fe->own_decode_stream;
	}
	for (int i = 0; i < 2; i++) {
		FE_TRY(hipEventCreateWithFlags(&fe->ev_chan[i], hipEventDisableTiming));
		FE_TRY(hipEventCreateWithFlags(&fe->ev_demod[i], hipEventDisableTiming));
		for (auto &e : fe->ev_dm[i]) FE_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	if ((rc = fe->fft.build(pl.n))) { frontend_free(fe); return rc; }
	const size_t n = (size_t)pl.n;
	for (int i = 0; i < 2; i++) {      // overlap history, ping-pong: block k reads [k&1] and leaves the next one in [(k+1)&1]
		FE_TRY(hipMalloc(&fe->d_hist[i], sizeof(float2) * (size_t)pl.overlap));
		FE_TRY(hipMemsetAsync(fe->d_hist[i], 0, sizeof(float2) * (size_t)pl.overlap, fe->stream));   // calloc'ed history, src/fft.c:79
	}
	FE_TRY(hipMalloc(&fe->d_work, sizeof(float2) * n));
	FE_TRY(hipMalloc(&fe->d_taps, sizeof(float2) * n * (size_t)g.nch_pad));
	FE_TRY(hipMalloc(&fe->d_nco, sizeof(NcoState) * (size_t)nch));
	FE_TRY(hipMemsetAsync(fe->d_nco, 0, sizeof(NcoState) * (size_t)nch, fe->stream));
	FE_TRY(hipMalloc(&fe->d_ph_cont, sizeof(float2) * (size_t)nch));
	FE_TRY(hipMalloc(&fe->d_cc, sizeof(ChanConst) * (size_t)nch));
	{
		float2 *tw = nullptr;
		if ((rc = upload_twiddles(pl.m, &tw))) { frontend_free(fe); return rc; }
		fe->d_tw_m = tw;
	}
	if (prepare_ifft_nco(pl.m) != hipSuccess) {
		rc = fail(HFDL_GPU_EHIP, "inverse FFT of %d points: LDS attribute refused: %s", pl.m, hipGetErrorString(hipGetLastError()));
		frontend_free(fe);
		return rc;
	}
	if ((rc = build_taps(fe))) { frontend_free(fe); return rc; }
	if (fe->prune_tol > 0 && g.tap_layout == TAPL_OCTET && (rc = build_fold_windows(fe))) { frontend_free(fe); return rc; }
	FE_TRY(hipMemcpy(fe->d_cc, fe->cc.data(), sizeof(ChanConst) * (size_t)nch, hipMemcpyHostToDevice));
	float resamp_rate = (float)(1800 * 3) / ((float)sample_rate / (float)fe->decimation);
	fe->fold_nb = pick_fold_batch(fe);
	int want_batch = Demod::fit_batch(g.outs, resamp_rate, pick_demod_batch(fe));       // what fits the demodulator's LDS
	if (!getenv("HFDL_GPU_DEMOD_BATCH") && want_batch < fe->fold_nb) {
		// even launches: a half of 8 blocks at up to 7 per launch is two launches of 4, not 7 + 1 (cfg2: 5.5 against 5.8 Gsamples/s); an
		// explicit HFDL_GPU_DEMOD_BATCH is taken as it is
		const int launches = (fe->fold_nb + want_batch - 1) / want_batch;
		want_batch = (fe->fold_nb + launches - 1) / launches;
	}
	if ((rc = fe->demod.init(nch, g.outs, resamp_rate, fe->freqs.data(), fe->stream, want_batch))) { frontend_free(fe); return rc; }
	fe->batch = fe->demod.batch;
	fe->half_blocks = std::min((int)hfdl_gpu_frontend::MAX_HALF, ((std::max(fe->fold_nb, fe->batch) + fe->fold_nb - 1) / fe->fold_nb) * fe->fold_nb);
	fe->half_first = (fe->fold_bound && fe->half_blocks > 16) ? 16 : fe->half_blocks;
#ifdef HFDL_LAB
	if (env_long("HFDL_GPU_FOLD_RAMP", 0, 1, 1) == 0) fe->half_first = fe->half_blocks;      // A/B: every half the full size from the start
#endif
	fe->half_target = fe->half_first;
	fe->n_stage = std::min(fe->half_blocks + 2, (int)hfdl_gpu_frontend::MAX_STAGE);      // a 32-block half is not uploaded a whole half ahead: 17 blocks of link time cover a 6 ms fold five times over
	for (int i = 0; i < fe->n_stage; i++) {
		FE_TRY(hipEventCreateWithFlags(&fe->ev_stage_ready[i], hipEventDisableTiming));
		FE_TRY(hipEventCreateWithFlags(&fe->ev_stage_free[i], hipEventDisableTiming));
	}
	const size_t hb = (size_t)fe->half_blocks;
	FE_TRY(hipMalloc(&fe->d_spec, sizeof(float2) * n * 2 * hb));
	FE_TRY(hipMalloc(&fe->d_partial, sizeof(float2) * fe->partial_stride() * hb));
	FE_TRY(hipMalloc(&fe->d_ph, sizeof(float2) * fe->ph_stride() * 2 * hb));
	FE_TRY(hipMalloc(&fe->d_nco_snap, sizeof(NcoState) * (size_t)nch * 2 * hb));
	FE_TRY(hipMalloc(&fe->d_chan_all, sizeof(float2) * 2 * hb * (size_t)nch * g.outs));
	FE_TRY(hipMalloc(&fe->d_cnt_all, sizeof(int) * 2 * hb * (size_t)nch));
	FE_TRY(hipMemsetAsync(fe->d_cnt_all, 0, sizeof(int) * 2 * hb * (size_t)nch, fe->stream));
	FE_TRY(hipStreamSynchronize(fe->stream));
#undef FE_TRY
	*out = fe;
	return 0;
}

extern "C" int hfdl_gpu_frontend_geometry(const hfdl_gpu_frontend *fe, hfdl_gpu_geometry *g)
{
	if (!fe || !g) return fail(HFDL_GPU_EINVAL, "null argument");
	const Plan &p = fe->plan;
	g->sample_rate = fe->sample_rate; g->decimation = fe->decimation;
	g->pre_decimation = p.pre; g->post_decimation = p.post;
	g->taps_length = p.taps_length; g->overlap_length = p.overlap;
	g->fft_size = p.n; g->fft_inv_size = p.m; g->input_size = p.input_size;
	g->post_input_size = p.post_input_size; g->scrap = p.scrap;
	g->outputs_per_block = p.post_input_size / p.post;
	g->max_outputs_per_block = (p.post_input_size + p.post - 1) / p.post;
	g->channels = fe->geo.nch; g->fold_slices = fe->geo.slices;
	g->demod_batch = fe->batch;
	g->fold_batch = fe->fold_nb;
	g->prefetch_depth = fe->n_stage - 1;
	g->fold_rows = fe->fold_rows_max ? fe->fold_rows_max : p.pre;
	g->transition_bw = fe->tbw;
	g->resamp_rate = (float)(1800 * 3) / ((float)fe->sample_rate / (float)fe->decimation);
	return 0;
}

extern "C" int hfdl_gpu_plan_geometry(int32_t decimation, float transition_bw, hfdl_gpu_geometry *g)
{
	if (!g || decimation < 1 || !(transition_bw > 0.f)) return fail(HFDL_GPU_EINVAL, "bad arguments");
	Plan p;
	if (!plan_block(p, transition_bw, decimation, 0.f)) return fail(HFDL_GPU_EINVAL, "fastddc planning failed");
	memset(g, 0, sizeof(*g));
	g->decimation = decimation;
	g->pre_decimation = p.pre; g->post_decimation = p.post;
	g->taps_length = p.taps_length; g->overlap_length = p.overlap;
	g->fft_size = p.n; g->fft_inv_size = p.m; g->input_size = p.input_size;
	g->post_input_size = p.post_input_size; g->scrap = p.scrap;
	g->outputs_per_block = p.post_input_size / p.post;
	g->max_outputs_per_block = (p.post_input_size + p.post - 1) / p.post;
	g->transition_bw = transition_bw;
	return 0;
}

// page-locked ranges handed out by hfdl_gpu_host_alloc(): only these are left to the DMA engine after push_block returns
static std::mutex g_pinned_lock;
static std::vector<std::pair<const char *, size_t>> g_pinned;

static bool is_library_pinned(const void *p, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_pinned_lock);
	for (auto &r : g_pinned)
		if ((const char *)p >= r.first && (const char *)p + bytes <= r.first + r.second) return true;
	return false;
}

extern "C" int hfdl_gpu_host_alloc(void **ptr, size_t bytes)
{
	if (!ptr || !bytes) return fail(HFDL_GPU_EINVAL, "bad arguments");
	HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
	std::lock_guard<std::mutex> lk(g_pinned_lock);
	g_pinned.emplace_back((const char *)*ptr, bytes);
	return 0;
}

extern "C" void hfdl_gpu_host_free(void *ptr)
{
	if (!ptr) return;
	{
		std::lock_guard<std::mutex> lk(g_pinned_lock);
		for (size_t i = 0; i < g_pinned.size(); i++)
			if (g_pinned[i].first == (const char *)ptr) { g_pinned.erase(g_pinned.begin() + (long)i); break; }
	}
	(void)hipHostFree(ptr);
}

extern "C" void *hfdl_gpu_frontend_stream(hfdl_gpu_frontend *fe) { return fe ? (void *)fe->stream_f : nullptr; }      // the stream that reads the input

static size_t sample_bytes(int fmt) { return fmt == SFMT_CS16 ? 4 : fmt == SFMT_CU8 ? 2 : 8; }

// queue the host -> device copy of the next host block on stream C into staging buffer host_blocks % n_stage
static int queue_input_copy(hfdl_gpu_frontend *fe, const void *iq, size_t nsamples, int fmt, int *sb_out)
{
	// buffer j % n_stage is freed by the forward FFT of host block j - n_stage: that block must have been pushed
	if (fe->host_blocks - fe->host_pushed >= (uint64_t)fe->n_stage)
		return fail(HFDL_GPU_ERANGE, "%d uploads are queued ahead of their blocks: push the oldest first", fe->n_stage);
	const uint64_t j = fe->host_blocks;
	const int sb = (int)(j % (uint64_t)fe->n_stage);
	if (fe->stage_cap[sb] < nsamples) {
		HIP_TRY(hipStreamSynchronize(fe->stream_c));
		HIP_TRY(hipStreamSynchronize(fe->stream_f));
		if (fe->d_stage[sb]) (void)hipFree(fe->d_stage[sb]);
		fe->d_stage[sb] = nullptr; fe->stage_cap[sb] = 0;
		HIP_TRY(hipMalloc(&fe->d_stage[sb], sizeof(float2) * nsamples));
		fe->stage_cap[sb] = nsamples;
	}
	HIP_TRY(hipStreamWaitEvent(fe->stream_c, fe->ev_stage_free[sb], 0));     // the forward FFT that read this buffer last is past its first pass
	// (one hipMemcpyAsync per block: cutting a block in 2 or 4 pieces on as many streams was measured and is slower, profiles/r03_experiments.md)
	HIP_TRY(hipMemcpyAsync(fe->d_stage[sb], iq, sample_bytes(fmt) * nsamples, hipMemcpyHostToDevice, fe->stream_c));
	HIP_TRY(hipEventRecord(fe->ev_stage_ready[sb], fe->stream_c));
	fe->host_blocks = j + 1;
	// a buffer this library did not allocate may be reused by the caller as soon as we return (include/hfdl_gpu.h): do not
	// rely on the runtime staging pageable memory synchronously -- wait for the copy (the kernels of the previous block keep running)
	if (!is_library_pinned(iq, sample_bytes(fmt) * nsamples)) HIP_TRY(hipStreamSynchronize(fe->stream_c));
	*sb_out = sb;
	return 0;
}

// Host input is staged in HBM: the copies (stream C) run up to a whole half ahead of the blocks that compute (stream A).
// *stage_idx = staging buffer used (-1 for device input): the forward FFT's first pass signals ev_stage_free when it has read it.
static int stage_input(hfdl_gpu_frontend *fe, const void *iq, size_t nsamples, int fmt, int on_device, const void **dev, int *stage_idx)
{
	*stage_idx = -1;
	if (!fe || !iq) return fail(HFDL_GPU_EINVAL, "null argument");
	if (fmt != SFMT_CF32 && fmt != SFMT_CS16 && fmt != SFMT_CU8) return fail(HFDL_GPU_EINVAL, "unknown sample format %d", fmt);
	if (nsamples != (size_t)fe->plan.input_size)
		return fail(HFDL_GPU_EINVAL, "a block is exactly %d samples (got %zu)", fe->plan.input_size, nsamples);
	HIP_TRY(hipSetDevice(fe->device));
	const bool queued = fe->host_pushed < fe->host_blocks;      // prefetched blocks are waiting
	if (on_device) {
		// the prefetched host blocks are numbered and staged: a device block slipped in front of them would be processed out of order
		if (queued) return fail(HFDL_GPU_EINVAL, "a prefetched block is pending: push it or call hfdl_gpu_frontend_prefetch_cancel()");
		*dev = iq;
		return 0;
	}
	int sb;
	if (queued) {
		// the copy of this block was queued ahead by hfdl_gpu_frontend_prefetch_block_raw(): blocks are pushed in the order they were prefetched
		sb = (int)(fe->host_pushed % (uint64_t)fe->n_stage);
		if (fe->pf_ptr[sb] != iq || fe->pf_fmt[sb] != fmt) return fail(HFDL_GPU_EINVAL, "the block pushed after a prefetch must be the (oldest) prefetched one");
	} else {
		int rc = queue_input_copy(fe, iq, nsamples, fmt, &sb);
		if (rc) return rc;
	}
	fe->host_pushed++;
	HIP_TRY(hipStreamWaitEvent(fe->stream_f, fe->ev_stage_ready[sb], 0));
	*dev = fe->d_stage[sb];
	*stage_idx = sb;
	return 0;
}

// Where the demodulator's 256 single-wave workgroups land decides how much they disturb the fold kernel they run beside.
// Launched the moment the channelizer of a half is done, they race the next blocks' forward-FFT workgroups for LDS and
// the outcome depends on details as small as the FFT's LDS footprint: measured on MI355X, the same fold kernel took
// 2.56 ms or 2.87 ms per launch (profiles/r01_experiments.md).  So the demodulator launches of a half are held back until the
// forward FFTs of the NEXT half have been queued: the workgroups then arrive while only the LDS-free fold kernel is resident, spread
// evenly, and the fold time is the good one every time.  A sync / poll launches held-back demodulators at once.
// A half of `nblk` blocks is demodulated `batch` blocks per launch, each launch followed by its burst decoder.
static int launch_demod(hfdl_gpu_frontend *fe, int buf, int nblk, bool after_fft)
{
	// the forward FFTs of the next half follow this half's inverse FFT on stream A, so their event covers ev_chan too
	if (after_fft) HIP_TRY(hipStreamWaitEvent(fe->stream_b, fe->ev_fft_cur, 0));
	else HIP_TRY(hipStreamWaitEvent(fe->stream_b, fe->ev_chan_cur[buf] ? fe->ev_chan_cur[buf] : fe->ev_chan[buf], 0));
	for (int j0 = 0, l = 0; j0 < nblk; j0 += fe->batch, l++) {
		const int take = std::min(fe->batch, nblk - j0);
		// the done event rides on the kernel's dispatch; with the decoder on its own stream the channelizer has already waited for the
		// frame queue (close_half), so on the demodulator-bound geometries ONE barrier packet separates consecutive demodulators
		hipEvent_t t_start = nullptr, done = fe->ev_dm[buf][l];
		std::pair<hipEvent_t, hipEvent_t> e;
		if (fe->timing && take_timer_pair(fe, e)) {
			// the kernel's own start / stop events (no extra packet): the stop event doubles as this launch's "done" event
			t_start = e.first; done = e.second;
			fe->ev_dmt.push_back(e);
			fe->demod_timed_blocks += take;
		}
		fe->ev_dm_cur[buf] = done;           // after the loop: the LAST launch of the half
		const int slot = buf * fe->half_blocks + j0;
		int rc = fe->demod.enqueue_demod(fe->chan_slot(slot), fe->cnt_slot(slot), take, fe->stream_b, done, l == 0 && fe->frames_wait_on_a, t_start);
		if (rc) return fail(rc, "demod enqueue failed: %s", hipGetErrorString(hipGetLastError()));
		if (fe->own_decode_stream) HIP_TRY(hipStreamWaitEvent(fe->stream_d, done, 0));
		hipEvent_t k5_start = nullptr, k5_stop = nullptr;
		if (fe->timing && take_timer_pair(fe, e)) {
			k5_start = e.first; k5_stop = e.second;
			fe->ev_dect.push_back(e);
		}
		rc = fe->demod.enqueue_decode(buf, fe->stream_d, k5_start, k5_stop);
		if (rc) return fail(rc, "burst decoder enqueue failed: %s", hipGetErrorString(hipGetLastError()));
	}
	fe->frames_wait_on_a = false;
	HIP_TRY(hipEventRecord(fe->ev_demod[buf], fe->stream_d));
	return 0;
}

static int flush_pending_demod(hfdl_gpu_frontend *fe, bool after_fft)
{
	if (fe->pending_demod_buf < 0) return 0;
	const int buf = fe->pending_demod_buf, nblk = fe->pending_demod_nblk;
	fe->pending_demod_buf = -1;
	return launch_demod(fe, buf, nblk, after_fft);
}

// Stream A, when a block is pushed: its forward FFT into the next spectrum slot of the half being filled (the block's NCO phasor
// table and carried state ride on the three pass launches).  Nothing else happens until the half is closed.
static int enqueue_fft(hfdl_gpu_frontend *fe, const void *fresh, int fmt, int stage_idx)
{
	const Geometry &g = fe->geo;
	const int i = fe->batch_fill, set = fe->cur_half;
	// The events other streams (and the bench's fold timer) wait for ride on the kernel dispatches themselves
	// (hipExtLaunchKernelGGL start / stop events): a separate hipEventRecord is one more barrier packet in the queue, ~5 us
	// of idle machine each (profiles/r01_experiments.md).
	// FFT on stream A: the held-back demodulators of the half before follow the LAST forward FFT of this half (launch_demod)
	const bool pend = !fe->fft_own_stream && fe->pending_demod_buf >= 0 && i + 1 == fe->half_target;
	if (pend && !fe->ev_fft) HIP_TRY(hipEventCreateWithFlags(&fe->ev_fft, hipEventDisableTiming));
	// FFT on its own stream: this set of spectra / phasor tables / snapshots was last read by the fold and inverse FFT two halves ago
	if (fe->fft_own_stream && i == 0) HIP_TRY(hipStreamWaitEvent(fe->stream_f, fe->ev_chan_cur[set] ? fe->ev_chan_cur[set] : fe->ev_chan[set], 0));
	NcoJob job;
	job.cc = fe->d_cc; job.chain = fe->d_nco; job.snap = fe->snap_slot(set, i);
	job.ph = fe->ph_slot(set, i); job.cont = fe->d_ph_cont;
	job.nch = g.nch; job.outs = g.outs; job.post_input_size = g.post_input_size; job.post = g.post;
	hipEvent_t fft_done = fe->fft_own_stream ? fe->ev_spec[set] : (pend ? fe->ev_fft : nullptr), fft_start = nullptr;
	std::pair<hipEvent_t, hipEvent_t> e;
	if (fe->timing && !fe->fft_own_stream && take_timer_pair(fe, e)) {
		// the first pass' start and the last pass' stop ride on their dispatches; the stop event doubles as "this forward FFT is done"
		fft_start = e.first; fft_done = e.second;
		fe->ev_fftt.push_back(e);
	}
	if (pend) fe->ev_fft_cur = fft_done;
	launch_fft_forward(fe->fft.p, fe->d_hist[fe->blocks & 1], fresh, fmt, g.overlap, fe->d_hist[(fe->blocks + 1) & 1], fe->d_work, fe->spec_slot(set, i), true, fe->stream_f,
			FftOutLayout(), fft_done, job,
			stage_idx >= 0 ? fe->ev_stage_free[stage_idx] : nullptr,        // input consumed once pass 1 is done: the copy stream may refill the buffer
			fft_start);
	if (pend) {
		int rc = flush_pending_demod(fe, true);
		if (rc) return rc;
	}
	HIP_TRY(hipGetLastError());
	fe->blocks++;
	fe->batch_fill++;
	return 0;
}

// Stream A, when a half is closed (full, or as a sync / poll finds it): ONE pass over the filter taps per `fold_nb` spectra, then the
// inverse FFT / NCO of every block of the half in one launch.  A's inverse FFT may not overwrite a half before the demodulator
// launches that read it last (two halves ago) are done; B may not start before A has filled what it is given.
// with_demod: hand the half to the demodulator (now, or held back until the next half's forward FFTs are queued).
static int close_half(hfdl_gpu_frontend *fe, bool launch_now, bool with_demod = true)
{
	const int nblk = fe->batch_fill;
	if (nblk == 0) return 0;
	const Geometry &g = fe->geo;
	const int half = fe->cur_half;
	if (fe->fft_own_stream) HIP_TRY(hipStreamWaitEvent(fe->stream, fe->ev_spec[half], 0));      // the newest forward FFT of this half (stream F)
	// one launch per `fold_nb` blocks (a half holds a whole number of them only when it is full), each timed and counted AS LAUNCHED: the
	// shape the bench prices is a launch that happened
	for (int done = 0, take = 0; done < nblk; done += take) {
		take = std::min(fe->fold_nb, nblk - done);
		// a ragged rest of 17 .. 20 blocks: sixteen columns, then the four-column form (3.9 + 2.7 ms) -- thirty-two columns cost their 6.8 ms
		// whatever the block count
		if (take > 16 && take <= 20) take = 16;
		float2 *pp = fe->d_partial + (size_t)done * fe->partial_stride();
		if (fe->timing) {
			std::pair<hipEvent_t, hipEvent_t> e;
			if (!take_timer_pair(fe, e)) return fail(HFDL_GPU_EHIP, "hipEventCreate: %s", hipGetErrorString(hipGetLastError()));
			launch_fold(g, fe->d_taps, fe->spec_slot(half, done), (size_t)g.n, pp, fe->partial_stride(), take, fe->fold_nb, fe->stream, e.first, e.second);
			fe->ev.push_back(e);
			fe->ev_blocks.push_back(take);
		} else {
			launch_fold(g, fe->d_taps, fe->spec_slot(half, done), (size_t)g.n, pp, fe->partial_stride(), take, fe->fold_nb, fe->stream);
		}
	}
	// this half is free once the demodulator launches that read it last (two halves ago) are done
	if (fe->ev_dm_cur[half]) HIP_TRY(hipStreamWaitEvent(fe->stream, fe->ev_dm_cur[half], 0));
	if (with_demod && fe->own_decode_stream && !fe->fold_bound) {
		// this half's first demodulator (launched right after this kernel, on stream B) reuses the frame queue the decoder of two
		// launches ago read: on the demodulator-bound geometries wait for it HERE, where the stream has slack, instead of in front of
		// the demodulator.  (Where the fold bounds the block stream A is the critical one and must not wait for a burst decoder:
		// measured, a 4.3 ms hole in front of every inverse FFT.  There the demodulator waits itself, enqueue_demod.)
		hipEvent_t e = fe->demod.frames_free_event();
		if (e) HIP_TRY(hipStreamWaitEvent(fe->stream, e, 0));
		fe->frames_wait_on_a = true;
	}
	const int slot0 = half * fe->half_blocks;
	hipEvent_t ifft_start = nullptr, ifft_done = fe->ev_chan[half];
	std::pair<hipEvent_t, hipEvent_t> e;
	if (fe->timing && take_timer_pair(fe, e)) {
		ifft_start = e.first; ifft_done = e.second;
		fe->ev_ifftt.push_back(e);
	}
	fe->ev_chan_cur[half] = ifft_done;
	launch_ifft_nco(g, fe->d_partial, fe->partial_stride(), fe->d_cc, fe->snap_slot(half, 0), fe->ph_slot(half, 0), fe->ph_stride(), fe->d_tw_m,
			fe->chan_slot(slot0), fe->cnt_slot(slot0), nblk, fe->stream, ifft_done, ifft_start);
	HIP_TRY(hipGetLastError());
	fe->last_slot = slot0 + nblk - 1;
	fe->last_index = nblk - 1;
	fe->last_set = half;
	fe->cur_half ^= 1;
	fe->batch_fill = 0;
	if (!with_demod) return 0;                   // channelize-only: the half is simply left behind
	fe->prev_demod_buf = fe->demod_buf;          // what poll_pdus_ready(.., 1) waits for: the half before the newest one
	fe->demod_buf = half;
	// (forward FFTs on their own stream run beside everything anyway: there is no quiet moment to hold the demodulators back for)
	if (launch_now || fe->fft_own_stream) return launch_demod(fe, half, nblk, false);
	fe->pending_demod_buf = half;
	fe->pending_demod_nblk = nblk;
	return 0;
}

extern "C" int hfdl_gpu_frontend_channelize_block(hfdl_gpu_frontend *fe, const float *iq, size_t nsamples, int on_device)
{
	const void *fresh = nullptr;
	int sidx = -1;
	int rc = stage_input(fe, iq, nsamples, SFMT_CF32, on_device, &fresh, &sidx);
	if (rc) return rc;
	if ((rc = flush_pending_demod(fe, false))) return rc;
	if ((rc = close_half(fe, true))) return rc;             // blocks pushed for the demodulator and not yet handed to it
	if ((rc = enqueue_fft(fe, fresh, SFMT_CF32, sidx))) return rc;
	return close_half(fe, false, false);                    // this block is never demodulated
}

// The demodulator / burst-decoder stage alone: one block of channelizer OUTPUT from the host (what fastddc_inv_cc hands to
// hfdl_decoder_thread's loop body, src/hfdl.c:676) goes through K4 + K5 with the channel state carried as usual.  Stage parity:
// fed with the oracle's own channelizer output, the device demodulator is compared with the oracle's without the two channelizers'
// different fp32 roundings in between (tests/test_gpu_parity.py, profiles/strict_study.py).
extern "C" int hfdl_gpu_frontend_push_baseband(hfdl_gpu_frontend *fe, const float *chan_out, const int32_t *counts)
{
	if (!fe || !chan_out || !counts) return fail(HFDL_GPU_EINVAL, "null argument");
	const Geometry &g = fe->geo;
	for (int c = 0; c < g.nch; c++)
		if (counts[c] < 0 || counts[c] > g.outs) return fail(HFDL_GPU_ERANGE, "channel %d: %d samples, a block holds at most %d", c, counts[c], g.outs);
	int rc = hfdl_gpu_frontend_sync(fe);                    // a stage-test entry point: everything pushed before is finished first
	if (rc) return rc;
	const int half = fe->cur_half, slot0 = half * fe->half_blocks;
	HIP_TRY(hipMemcpy(fe->chan_slot(slot0), chan_out, sizeof(float2) * (size_t)g.nch * (size_t)g.outs, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(fe->cnt_slot(slot0), counts, sizeof(int32_t) * (size_t)g.nch, hipMemcpyHostToDevice));
	fe->last_slot = slot0;
	fe->last_index = 0;                                     // one block in this half: nothing further back to read (read_tap_block)
	fe->cur_half ^= 1;
	fe->prev_demod_buf = fe->demod_buf;
	fe->demod_buf = half;
	return launch_demod(fe, half, 1, false);
}

static int push_any(hfdl_gpu_frontend *fe, const void *raw, size_t nsamples, int fmt, int on_device)
{
	const void *fresh = nullptr;
	int sidx = -1;
	int rc = stage_input(fe, raw, nsamples, fmt, on_device, &fresh, &sidx);
	if (rc) return rc;
	if ((rc = enqueue_fft(fe, fresh, fmt, sidx))) return rc;
	if (fe->batch_fill < fe->half_target) return 0;         // the half is still filling
	// demodulator-bound geometry (few channels): the fold is short, there is nothing to place the demodulator under, and
	// holding it back until the NEXT half's forward FFTs would put those blocks' host -> device copies on the demodulator's
	// critical path (cfg2 fed from host memory: 0.56 -> 0.33 ms per block): launched at once.  Otherwise held back (launch_demod).
	rc = close_half(fe, !fe->fold_bound);
	fe->half_target = fe->half_blocks;                      // closed by filling: the caller runs ahead of the collection, the next halves take every slot
	return rc;
}

extern "C" int hfdl_gpu_frontend_push_block(hfdl_gpu_frontend *fe, const float *iq, size_t nsamples, int on_device)
{
	return push_any(fe, iq, nsamples, SFMT_CF32, on_device);
}

extern "C" int hfdl_gpu_frontend_push_block_raw(hfdl_gpu_frontend *fe, const void *raw, size_t nsamples, int sample_format, int on_device)
{
	return push_any(fe, raw, nsamples, sample_format, on_device);
}

static int drain_events(hfdl_gpu_frontend *fe)
{
	for (size_t i = 0; i < fe->ev.size(); i++) {
		auto &e = fe->ev[i];
		float ms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
		fe->fold_ms += ms;
		fe->fold_launches++;
		fe->fold_timed_blocks += fe->ev_blocks[i];
		fe->fold_last_blocks = fe->ev_blocks[i];
		if (fe->ev_blocks[i] >= 1 && fe->ev_blocks[i] <= FOLD_MAX_BLOCKS) { fe->fold_shapes[fe->ev_blocks[i]]++; fe->fold_shape_ms[fe->ev_blocks[i]] += ms; }
		if (!fe->ev_first_fold) {
			fe->ev_first_fold = e.first;            // kept until the next reset
			HIP_TRY(hipEventCreate(&e.first));
		} else {
			HIP_TRY(hipEventElapsedTime(&ms, fe->ev_first_fold, e.first));
			fe->span_ms = ms;
		}
		fe->ev_pool.push_back(e);                   // both events are complete: reused by later launches
	}
	fe->ev.clear();
	fe->ev_blocks.clear();
	for (auto &e : fe->ev_dmt) {
		float ms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
		fe->demod_ms += ms;
		fe->demod_launches++;
		fe->ev_pool.push_back(e);
	}
	fe->ev_dmt.clear();
	struct { std::vector<std::pair<hipEvent_t, hipEvent_t>> *v; double *ms; int64_t *n; } more[] = {
		{ &fe->ev_fftt, &fe->fft_ms, &fe->fft_timed }, { &fe->ev_ifftt, &fe->ifft_ms, &fe->ifft_timed }, { &fe->ev_dect, &fe->decode_ms, &fe->decode_timed } };
	for (auto &m : more) {
		for (auto &e : *m.v) {
			float ms = 0;
			HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
			*m.ms += ms;
			(*m.n)++;
			fe->ev_pool.push_back(e);
		}
		m.v->clear();
	}
	// everything is complete: the pooled events may be reused (a completed event stands for "done" as well as the half's own)
	for (int i = 0; i < 2; i++) if (fe->ev_dm_cur[i]) fe->ev_dm_cur[i] = fe->ev_dm[i][0];
	for (int i = 0; i < 2; i++) fe->ev_chan_cur[i] = nullptr;
	fe->ev_fft_cur = fe->ev_fft;
	return 0;
}

extern "C" int hfdl_gpu_frontend_sync(hfdl_gpu_frontend *fe)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(fe->device));
	{ int rc = flush_pending_demod(fe, false); if (rc) return rc; }
	{ int rc = close_half(fe, true); if (rc) return rc; }           // blocks waiting for their half to fill: folded and demodulated now
	fe->half_target = fe->half_first;                               // a drain: the pipeline starts over with a short first half
	HIP_TRY(hipStreamSynchronize(fe->stream_c));
	if (fe->fft_own_stream) HIP_TRY(hipStreamSynchronize(fe->stream_f));
	HIP_TRY(hipStreamSynchronize(fe->stream));
	HIP_TRY(hipStreamSynchronize(fe->stream_b));
	if (fe->own_decode_stream) HIP_TRY(hipStreamSynchronize(fe->stream_d));
	HIP_TRY(hipGetLastError());
	return drain_events(fe);
}

extern "C" int hfdl_gpu_frontend_input_done(hfdl_gpu_frontend *fe)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	HIP_TRY(hipSetDevice(fe->device));
	HIP_TRY(hipStreamSynchronize(fe->stream_c));
	return 0;
}

extern "C" int hfdl_gpu_frontend_input_done_upto(hfdl_gpu_frontend *fe, uint64_t host_block)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	if (host_block >= fe->host_blocks) return fail(HFDL_GPU_EINVAL, "host block %llu has not been pushed (%llu so far)", (unsigned long long)host_block, (unsigned long long)fe->host_blocks);
	// The copy of host block j signals ev_stage_ready[j % n_stage].  Copies run in order on one stream, so for a block more than
	// n_stage - 1 behind the newest (its event has been re-recorded since) the oldest event still its own block's implies it.
	const uint64_t newest = fe->host_blocks - 1, span = (uint64_t)fe->n_stage - 1;
	const uint64_t j = newest - host_block <= span ? host_block : newest - span;
	HIP_TRY(hipSetDevice(fe->device));
	HIP_TRY(hipEventSynchronize(fe->ev_stage_ready[j % (uint64_t)fe->n_stage]));
	return 0;
}

extern "C" int hfdl_gpu_frontend_input_copied(hfdl_gpu_frontend *fe, uint64_t host_block)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	if (host_block >= fe->host_blocks) return fail(HFDL_GPU_EINVAL, "host block %llu has not been pushed (%llu so far)", (unsigned long long)host_block, (unsigned long long)fe->host_blocks);
	const uint64_t newest = fe->host_blocks - 1, span = (uint64_t)fe->n_stage - 1;
	const uint64_t j = newest - host_block <= span ? host_block : newest - span;
	HIP_TRY(hipSetDevice(fe->device));
	const hipError_t e = hipEventQuery(fe->ev_stage_ready[j % (uint64_t)fe->n_stage]);
	if (e == hipSuccess) return 1;
	if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }       // "not yet" is an answer, not an error to be found by a later check
	return fail(HFDL_GPU_EHIP, "hipEventQuery: %s", hipGetErrorString(e));
}

extern "C" int hfdl_gpu_frontend_prefetch_block_raw(hfdl_gpu_frontend *fe, const void *raw, size_t nsamples, int sample_format)
{
	if (!fe || !raw) return fail(HFDL_GPU_EINVAL, "null argument");
	if (sample_format != SFMT_CF32 && sample_format != SFMT_CS16 && sample_format != SFMT_CU8) return fail(HFDL_GPU_EINVAL, "unknown sample format %d", sample_format);
	if (nsamples != (size_t)fe->plan.input_size)
		return fail(HFDL_GPU_EINVAL, "a block is exactly %d samples (got %zu)", fe->plan.input_size, nsamples);
	if (fe->host_blocks - fe->host_pushed >= (uint64_t)(fe->n_stage - 1))
		return fail(HFDL_GPU_ERANGE, "%d blocks are prefetched already (geometry.prefetch_depth): push the oldest first", fe->n_stage - 1);
	if (!is_library_pinned(raw, sample_bytes(sample_format) * nsamples)) return fail(HFDL_GPU_EINVAL, "only buffers from hfdl_gpu_host_alloc() can be prefetched");
	HIP_TRY(hipSetDevice(fe->device));
	int sb = -1;
	int rc = queue_input_copy(fe, raw, nsamples, sample_format, &sb);
	if (rc) return rc;
	fe->pf_ptr[sb] = raw; fe->pf_fmt[sb] = sample_format;
	return 0;
}

extern "C" int hfdl_gpu_frontend_prefetch_cancel(hfdl_gpu_frontend *fe)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	if (fe->host_pushed == fe->host_blocks) return 0;
	HIP_TRY(hipSetDevice(fe->device));
	// the copies are in flight on stream C: let them finish (the caller gets its buffers back), then forget the blocks.  They keep
	// their host block numbers -- input_done_upto() of those numbers returns at once.  Their staging buffers are refilled by later
	// copies, which wait for ev_stage_free as last recorded by the blocks that used the buffers BEFORE the cancelled ones: long done.
	HIP_TRY(hipStreamSynchronize(fe->stream_c));
	fe->host_pushed = fe->host_blocks;
	return 0;
}

extern "C" int hfdl_gpu_frontend_reset_timers(hfdl_gpu_frontend *fe, int enable)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	fe->fold_ms = 0; fe->fold_launches = 0; fe->fold_timed_blocks = 0; fe->fold_last_blocks = 0; fe->timing = enable != 0;
	for (auto &c : fe->fold_shapes) c = 0;
	for (auto &c : fe->fold_shape_ms) c = 0;
	fe->demod_ms = 0; fe->demod_launches = 0; fe->demod_timed_blocks = 0;
	fe->fft_ms = fe->ifft_ms = fe->decode_ms = 0; fe->fft_timed = fe->ifft_timed = fe->decode_timed = 0;
	if (fe->ev_first_fold) { (void)hipEventDestroy(fe->ev_first_fold); fe->ev_first_fold = nullptr; }
	fe->span_ms = 0;
	// enough event pairs for the launches between two drains (a sync / poll recycles them): created here, not in the timed loop
	while (enable && fe->ev_pool.size() < 1280) {
		std::pair<hipEvent_t, hipEvent_t> e;
		HIP_TRY(hipEventCreate(&e.first));
		HIP_TRY(hipEventCreate(&e.second));
		fe->ev_pool.push_back(e);
	}
	return 0;
}

extern "C" int hfdl_gpu_frontend_fold_time_ms(hfdl_gpu_frontend *fe, double *total_ms, int64_t *launches)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	if (total_ms) *total_ms = fe->fold_ms;
	if (launches) *launches = fe->fold_launches;
	return 0;
}

extern "C" int hfdl_gpu_frontend_fold_launch_shapes(hfdl_gpu_frontend *fe, int64_t counts[HFDL_GPU_FOLD_BATCH_MAX + 1], double ms[HFDL_GPU_FOLD_BATCH_MAX + 1])
{
	if (!fe || !counts) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	for (int i = 0; i <= FOLD_MAX_BLOCKS; i++) counts[i] = fe->fold_shapes[i];
	if (ms) for (int i = 0; i <= FOLD_MAX_BLOCKS; i++) ms[i] = fe->fold_shape_ms[i];
	return 0;
}

extern "C" int hfdl_gpu_frontend_fold_blocks(hfdl_gpu_frontend *fe, int64_t *blocks)
{
	if (!fe || !blocks) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	*blocks = fe->fold_timed_blocks;
	return 0;
}

extern "C" int hfdl_gpu_frontend_demod_time_ms(hfdl_gpu_frontend *fe, double *total_ms, int64_t *launches, int64_t *blocks)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	if (total_ms) *total_ms = fe->demod_ms;
	if (launches) *launches = fe->demod_launches;
	if (blocks) *blocks = fe->demod_timed_blocks;
	return 0;
}

extern "C" int hfdl_gpu_frontend_stage_times(hfdl_gpu_frontend *fe, double ms[5], int64_t launches[5])
{
	if (!fe || !ms || !launches) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	ms[0] = fe->fft_ms; ms[1] = fe->fold_ms; ms[2] = fe->ifft_ms; ms[3] = fe->demod_ms; ms[4] = fe->decode_ms;
	launches[0] = fe->fft_timed; launches[1] = fe->fold_launches; launches[2] = fe->ifft_timed; launches[3] = fe->demod_launches; launches[4] = fe->decode_timed;
	return 0;
}

extern "C" int hfdl_gpu_frontend_step_period_ms(hfdl_gpu_frontend *fe, double *period_ms)
{
	if (!fe || !period_ms) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	// first timed fold start -> last timed fold start covers every timed block but the last launch's; per BLOCK
	const int64_t covered = fe->fold_timed_blocks - fe->fold_last_blocks;
	*period_ms = (fe->fold_launches > 1 && covered > 0) ? fe->span_ms / (double)covered : 0.0;
	return 0;
}

extern "C" int hfdl_gpu_frontend_poll_pdus(hfdl_gpu_frontend *fe, hfdl_gpu_pdu *out, int32_t max, int32_t *n)
{
	if (!fe || !n) return fail(HFDL_GPU_EINVAL, "null argument");
	if (!out && max > 0) return fail(HFDL_GPU_EINVAL, "null PDU buffer with max = %d", max);
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	rc = fe->demod.collect(out, max, n, fe->stream_d);
	if (rc) return fail(rc, "pdu collection failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

extern "C" int hfdl_gpu_frontend_poll_pdus_ready(hfdl_gpu_frontend *fe, hfdl_gpu_pdu *out, int32_t max, int32_t *n, int32_t max_in_flight)
{
	if (!fe || !n) return fail(HFDL_GPU_EINVAL, "null argument");
	if (!out && max > 0) return fail(HFDL_GPU_EINVAL, "null PDU buffer with max = %d", max);
	if (max_in_flight <= 0) return hfdl_gpu_frontend_poll_pdus(fe, out, max, n);
	*n = 0;
	// Leave the newest launch running: wait for the DEMODULATOR of the one before it -- that is the flow control: the caller may
	// queue the next blocks, the demodulator stream will not run dry -- and take what the PDU ring is known to hold: the snapshot
	// written after that launch's burst decoder if it has finished too, else the one written two launches earlier (a stale snapshot
	// yields nothing new; those PDUs come with the next call).  Waiting for the burst decoder here (0.3 ms when a long frame ended)
	// left the demodulator idle for as long on the demodulator-bound geometries (profiles/r03_experiments.md).
	const int buf = fe->prev_demod_buf;
	if (buf < 0) return 0;                              // fewer than two launches: nothing is known to be done
	HIP_TRY(hipSetDevice(fe->device));
	{
		hipEvent_t ev = fe->ev_dm_cur[buf] ? fe->ev_dm_cur[buf] : fe->ev_demod[buf];
		if (max_in_flight >= 2) {
			// no waiting at all: a caller whose flow control is elsewhere (the C host: the page-locked ring slots it leases to the
			// uploads) only takes what is complete, and comes back
			const hipError_t q = hipEventQuery(ev);
			if (q == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
			if (q != hipSuccess) return fail(HFDL_GPU_EHIP, "hipEventQuery: %s", hipGetErrorString(q));
		} else {
			HIP_TRY(hipEventSynchronize(ev));
		}
	}
	// The snapshot slot of that half is written by its burst decoders' 16-byte copies: read it only once the last of them is known to be
	// done (ev_demod is recorded behind it); until then the OTHER slot is the stable one -- it was written two halves ago, and the
	// newest half's decoders, which write it next, sit behind this half's on their stream.
	const bool decoded = hipEventQuery(fe->ev_demod[buf]) == hipSuccess;
	if (!decoded) (void)hipGetLastError();            // "not ready" is an answer, not an error to be found by a later check
	int rc = fe->demod.collect_snapshot(decoded ? buf : buf ^ 1, out, max, n, fe->stream_d);
	if (rc) return fail(rc, "pdu collection failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

extern "C" int hfdl_gpu_frontend_all_channel_stats(hfdl_gpu_frontend *fe, hfdl_gpu_channel_stats *out, int32_t cap, int32_t *n)
{
	if (!fe || !out || !n) return fail(HFDL_GPU_EINVAL, "null argument");
	if (cap < fe->geo.nch) return fail(HFDL_GPU_ERANGE, "%d channels, buffer holds %d", fe->geo.nch, cap);
	HIP_TRY(hipSetDevice(fe->device));
	memset(out, 0, sizeof(*out) * (size_t)fe->geo.nch);
	int rc = fe->demod.stats_all(out, fe->geo.nch);
	if (rc) return fail(rc, "stats read failed: %s", hipGetErrorString(hipGetLastError()));
	for (int i = 0; i < fe->geo.nch; i++) out[i].freq = fe->freqs[(size_t)i];
	*n = fe->geo.nch;
	return 0;
}

extern "C" int hfdl_gpu_frontend_counters(hfdl_gpu_frontend *fe, hfdl_gpu_frontend_counters_t *out)
{
	if (!fe || !out) return fail(HFDL_GPU_EINVAL, "null argument");
	memset(out, 0, sizeof(*out));
	out->blocks = fe->blocks;
	out->pdus_taken = fe->demod.taken;
	out->pdus_dropped = fe->demod.dropped;
	out->pdu_ring_capacity = (uint32_t)fe->demod.pdu_cap;
	return 0;
}

extern "C" int hfdl_gpu_frontend_enable_taps(hfdl_gpu_frontend *fe, int enable)
{
	if (!fe) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	fe->demod.taps_enabled = enable != 0;
	return 0;
}

extern "C" int hfdl_gpu_frontend_channel_stats(hfdl_gpu_frontend *fe, int32_t channel, hfdl_gpu_channel_stats *out)
{
	if (!fe || !out) return fail(HFDL_GPU_EINVAL, "null argument");
	if (channel < 0 || channel >= fe->geo.nch) return fail(HFDL_GPU_EINVAL, "channel out of range");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	memset(out, 0, sizeof(*out));
	out->freq = fe->freqs[(size_t)channel];
	rc = fe->demod.stats(channel, out);
	if (rc) return fail(rc, "stats read failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

extern "C" int hfdl_gpu_frontend_read_tap_block(hfdl_gpu_frontend *fe, int what, int32_t channel, int32_t back, float *dst, size_t cap, size_t *n_floats)
{
	if (!fe || !dst || !n_floats) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	const Geometry &g = fe->geo;
	// the channelizer's own buffers hold every block of the newest half: `back` blocks before the newest one
	if (back < 0 || back > fe->last_index) return fail(HFDL_GPU_ERANGE, "block %d back is not held any more (%d are)", back, fe->last_index);
	if (back && what != HFDL_GPU_TAP_SPECTRUM && what != HFDL_GPU_TAP_CHAN_OUT && what != HFDL_GPU_TAP_NCO_PHASORS)
		return fail(HFDL_GPU_EINVAL, "tap %d holds the last launch only", what);
	const int slot = fe->last_slot - back, index = fe->last_index - back;
	if (what != HFDL_GPU_TAP_SPECTRUM && (channel < 0 || channel >= g.nch)) return fail(HFDL_GPU_EINVAL, "channel out of range");
	const void *src = nullptr;
	size_t nf = 0;
	switch (what) {
	case HFDL_GPU_TAP_SPECTRUM: src = fe->spec_slot(fe->last_set, index); nf = 2 * (size_t)g.n; break;
	case HFDL_GPU_TAP_FILTER: {
		// the taps lie in matrix-operand order (kernels.h tap_index_f): a kernel gathers the channel into plain cf32[N]
		if (2 * (size_t)g.n > cap) return fail(HFDL_GPU_ERANGE, "tap needs %zu floats, buffer holds %zu", 2 * (size_t)g.n, cap);
		DevBuf plain;
		HIP_TRY(plain.alloc(sizeof(float2) * (size_t)g.n));
		launch_tap_extract(fe->d_taps, g, channel, plain.as<float2>(), fe->stream);
		HIP_TRY(hipStreamSynchronize(fe->stream));
		HIP_TRY(hipMemcpy(dst, plain.p, sizeof(float2) * (size_t)g.n, hipMemcpyDeviceToHost));
		*n_floats = 2 * (size_t)g.n;
		return 0; }
	case HFDL_GPU_TAP_CHAN_OUT: {
		int cnt = 0;
		HIP_TRY(hipMemcpy(&cnt, fe->cnt_slot(slot) + channel, sizeof(cnt), hipMemcpyDeviceToHost));
		src = fe->chan_slot(slot) + (size_t)channel * g.outs; nf = 2 * (size_t)cnt; break; }
	case HFDL_GPU_TAP_NCO_PHASORS: {
		int cnt = 0;
		HIP_TRY(hipMemcpy(&cnt, fe->cnt_slot(slot) + channel, sizeof(cnt), hipMemcpyDeviceToHost));
		if (2 * (size_t)cnt > cap) return fail(HFDL_GPU_ERANGE, "tap needs %zu floats, buffer holds %zu", 2 * (size_t)cnt, cap);
		// column `channel` of the [outs][nch] table
		if (cnt) HIP_TRY(hipMemcpy2D(dst, sizeof(float2), fe->ph_slot(fe->last_set, index) + channel, sizeof(float2) * (size_t)g.nch, sizeof(float2), (size_t)cnt, hipMemcpyDeviceToHost));
		*n_floats = 2 * (size_t)cnt;
		return 0; }
	case HFDL_GPU_TAP_PHASE_CYCLES: src = fe->demod.d_tap_lvl + (size_t)channel * fe->demod.cap + fe->demod.cap - 4; nf = 4; break;
	default:
		rc = fe->demod.tap(what, channel, &src, &nf);
		if (rc) return fail(rc, "unknown tap %d", what);
	}
	if (nf > cap) return fail(HFDL_GPU_ERANGE, "tap needs %zu floats, buffer holds %zu", nf, cap);
	if (nf) HIP_TRY(hipMemcpy(dst, src, sizeof(float) * nf, hipMemcpyDeviceToHost));
	*n_floats = nf;
	return 0;
}

extern "C" int hfdl_gpu_frontend_read_tap(hfdl_gpu_frontend *fe, int what, int32_t channel, float *dst, size_t cap, size_t *n_floats)
{
	return hfdl_gpu_frontend_read_tap_block(fe, what, channel, 0, dst, cap, n_floats);
}

// ---------------------------------------------------------------- stage-level entry points

extern "C" int hfdl_gpu_fft_forward(int device, const float *in, float *out, int32_t n, int shifted)
{
	if (!in || !out) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = select_device(device);
	if (rc) return rc;
	struct PlanGuard { HostFftPlan p; ~PlanGuard() { p.release(); } } plan;
	if ((rc = plan.p.build(n))) return rc;
	DevBuf d_in, d_work, d_out;
	const size_t bytes = sizeof(float2) * (size_t)n;
	HIP_TRY(d_in.alloc(bytes));
	HIP_TRY(d_work.alloc(bytes));
	HIP_TRY(d_out.alloc(bytes));
	HIP_TRY(hipMemcpy(d_in.p, in, bytes, hipMemcpyHostToDevice));
	StageTimer tm;
	launch_fft_forward(plan.p.p, nullptr, d_in.p, SFMT_CF32, 0, nullptr, d_work.as<float2>(), d_out.as<float2>(), shifted != 0, nullptr);
	tm.stop();
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpy(out, d_out.p, bytes, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int hfdl_gpu_viterbi27(int device, const uint8_t *soft, int32_t nbits, int32_t nframes, uint8_t *out)
{
	if (!soft || !out || nbits <= 0 || nframes <= 0) return fail(HFDL_GPU_EINVAL, "bad arguments");
	int rc = select_device(device);
	if (rc) return rc;
	g_stage_ms = 0.0;
	rc = demod_viterbi_batch(soft, nbits, nframes, out, &g_stage_ms);
	if (rc) return fail(rc, "viterbi batch failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

extern "C" int hfdl_gpu_burst_decode(int device, const float *symbols, const int32_t *modes, const int32_t *bitmask_lsb,
		int32_t nframes, uint8_t *octets, int32_t *lens)
{
	if (!symbols || !modes || !bitmask_lsb || !octets || !lens || nframes <= 0) return fail(HFDL_GPU_EINVAL, "bad arguments");
	for (int i = 0; i < nframes; i++) if (modes[i] < 0 || modes[i] > 7) return fail(HFDL_GPU_EINVAL, "mode out of range");
	int rc = select_device(device);
	if (rc) return rc;
	g_stage_ms = 0.0;
	rc = demod_burst_decode_batch(symbols, modes, bitmask_lsb, nframes, octets, lens, &g_stage_ms);
	if (rc) return fail(rc, "burst decode failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

// decimating_shift_addition_init + decimating_shift_addition_cc (src/libcsdr_gpl.c:26-74) on the device: the NCO / decimator
// tail of the channelizer kernel as a stage of its own, state carried by the caller exactly like the reference's status struct
extern "C" int hfdl_gpu_nco_decimate(int device, const float *in, int32_t input_size, float rate, int32_t decimation,
		int32_t *decimation_remain, float *starting_phase, float *out, int32_t *output_size)
{
	if (!in || !decimation_remain || !starting_phase || !out || !output_size || input_size <= 0 || decimation <= 0 || *decimation_remain < 0)
		return fail(HFDL_GPU_EINVAL, "bad arguments");
	int rc = select_device(device);
	if (rc) return rc;
	float r = rate * (float)decimation;        // decimating_shift_addition_init -> shift_addition_init, fp32 products as written there
	r *= 2;
	const float sd = (float)std::sin(r * M_PI), cd = (float)std::cos(r * M_PI);
	NcoState st{};
	st.decimation_remain = *decimation_remain; st.starting_phase = *starting_phase;
	const size_t max_out = ((size_t)input_size + (size_t)decimation - 1) / (size_t)decimation;
	DevBuf d_in, d_out, d_ph, d_st;
	HIP_TRY(d_in.alloc(sizeof(float2) * (size_t)input_size));
	HIP_TRY(d_out.alloc(sizeof(float2) * max_out));
	HIP_TRY(d_ph.alloc(sizeof(float2) * max_out));
	HIP_TRY(d_st.alloc(sizeof(NcoState)));
	HIP_TRY(hipMemcpy(d_in.p, in, sizeof(float2) * (size_t)input_size, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_st.p, &st, sizeof(st), hipMemcpyHostToDevice));
	launch_nco_decimate(d_in.as<const float2>(), input_size, cd, sd, r, decimation, d_st.as<NcoState>(), d_ph.as<float2>(), d_out.as<float2>(), nullptr);
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpy(&st, d_st.p, sizeof(st), hipMemcpyDeviceToHost));
	if (st.output_size > 0) HIP_TRY(hipMemcpy(out, d_out.p, sizeof(float2) * (size_t)st.output_size, hipMemcpyDeviceToHost));
	*decimation_remain = st.decimation_remain; *starting_phase = st.starting_phase; *output_size = st.output_size;
	return 0;
}

extern "C" int hfdl_gpu_crc16_ccitt(int device, const uint8_t *data, uint32_t len, uint16_t crc_init, uint16_t *crc)
{
	if (!crc || (!data && len)) return fail(HFDL_GPU_EINVAL, "bad arguments");
	int rc = select_device(device);
	if (rc) return rc;
	rc = demod_crc16(data, len, crc_init, crc);
	if (rc) return fail(rc, "crc16 failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

extern "C" int hfdl_gpu_pdu_triage(int device, const uint8_t *octets, const int32_t *lens, int32_t npdus, int32_t stride,
		uint8_t *fcs_status, uint8_t *pdu_kind, uint16_t *hdr_len)
{
	if (!octets || !lens || !fcs_status || !pdu_kind || !hdr_len || npdus <= 0 || stride <= 0) return fail(HFDL_GPU_EINVAL, "bad arguments");
	for (int i = 0; i < npdus; i++) if (lens[i] < 1 || lens[i] > stride) return fail(HFDL_GPU_EINVAL, "PDU %d: length %d outside 1..%d", i, lens[i], stride);
	int rc = select_device(device);
	if (rc) return rc;
	rc = demod_pdu_triage_batch(octets, lens, npdus, stride, fcs_status, pdu_kind, hdr_len);
	if (rc) return fail(rc, "pdu triage failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

extern "C" int hfdl_gpu_lpdu_walk(int device, const uint8_t *octets, const int32_t *lens, int32_t npdus, int32_t stride, uint8_t *counts)
{
	if (!octets || !lens || !counts || npdus <= 0 || stride <= 0) return fail(HFDL_GPU_EINVAL, "bad arguments");
	for (int i = 0; i < npdus; i++) if (lens[i] < 1 || lens[i] > stride) return fail(HFDL_GPU_EINVAL, "PDU %d: length %d outside 1..%d", i, lens[i], stride);
	int rc = select_device(device);
	if (rc) return rc;
	rc = demod_lpdu_walk_batch(octets, lens, npdus, stride, counts);
	if (rc) return fail(rc, "lpdu walk failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

extern "C" int hfdl_gpu_psk_slice(int device, int32_t arity, const float *xy, int32_t n, uint32_t *sym, float *phase_error)
{
	if (!xy || !sym || !phase_error || n <= 0) return fail(HFDL_GPU_EINVAL, "bad arguments");
	if (arity < 1 || arity > 3) return fail(HFDL_GPU_EINVAL, "arity %d: HFDL uses BPSK, QPSK and 8-PSK (1..3 bits per symbol)", arity);
	int rc = select_device(device);
	if (rc) return rc;
	rc = demod_psk_slice_batch(arity, xy, n, sym, phase_error);
	if (rc) return fail(rc, "psk slice failed: %s", hipGetErrorString(hipGetLastError()));
	return 0;
}

// ---------------------------------------------------------------- laboratory build only (libhfdl_gpu_lab.so, include/hfdl_gpu_lab.h)
#ifdef HFDL_LAB
#include "../../include/hfdl_gpu_lab.h"

extern "C" int hfdl_gpu_lab_stream_read_probe(hfdl_gpu_frontend *fe, double *gb_per_s)
{
	if (!fe || !gb_per_s) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	// read the resident filter taps themselves (whole multiples of 4 MiB, at most 16 GiB): best launch of every variant
	size_t bytes = sizeof(float2) * (size_t)fe->geo.n * (size_t)fe->geo.nch_pad;
	bytes -= bytes % ((size_t)4 << 20);
	if (bytes > ((size_t)16 << 30)) bytes = (size_t)16 << 30;
	if (bytes == 0) return fail(HFDL_GPU_ERANGE, "front end too small for the probe");
	DevBuf sink;
	HIP_TRY(sink.alloc(sizeof(float)));
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	double best = 0;
	for (int variant = 0; variant < stream_read_variants(); variant++)
		for (int it = 0; it < 3; it++) {
			HIP_TRY(hipEventRecord(e0, fe->stream));
			launch_stream_read(variant, fe->d_taps, bytes, sink.as<float>(), fe->stream);
			HIP_TRY(hipEventRecord(e1, fe->stream));
			HIP_TRY(hipEventSynchronize(e1));
			float ms = 0;
			HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
			if (getenv("HFDL_GPU_PROBE_VERBOSE")) fprintf(stderr, "stream read variant %d: %.1f GB/s\n", variant, (double)bytes / (ms * 1e-3) / 1e9);
			if (it > 0 && ms > 0) best = std::max(best, (double)bytes / (ms * 1e-3) / 1e9);
		}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	*gb_per_s = best;
	return 0;
}

extern "C" int hfdl_gpu_lab_read_constants(hfdl_gpu_frontend *fe, void *tables, size_t tables_bytes, void *constants, size_t constants_bytes)
{
	if (!fe || !tables || !constants) return fail(HFDL_GPU_EINVAL, "null argument");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	rc = fe->demod.read_constants(tables, tables_bytes, constants, constants_bytes);
	if (rc) return fail(rc, "constants read-back failed (sizes %zu / %zu): %s", tables_bytes, constants_bytes, hipGetErrorString(hipGetLastError()));
	return 0;
}

extern "C" int hfdl_gpu_lab_clock_probe_read(int which, uint64_t *records, int32_t max, int32_t *n)
{
	if (!records || !n || max < 1) return fail(HFDL_GPU_EINVAL, "bad arguments");
	int k = 0;
	const int rc = which == 0 ? fold_clock_probe_read((unsigned long long *)records, max, &k) : demod_clock_probe_read((unsigned long long *)records, max, &k);
	if (rc) return fail(HFDL_GPU_EHIP, "clock probe read failed: %s", hipGetErrorString(hipGetLastError()));
	*n = k;
	return 0;
}

extern "C" int hfdl_gpu_lab_fold_variant_count(void) { return fold_variant_count(); }

extern "C" int hfdl_gpu_lab_fold_variant_describe(int variant, int32_t desc[6])
{
	int d[6];
	if (!desc || fold_variant_describe(variant, d)) return fail(HFDL_GPU_EINVAL, "no fold variant %d", variant);
	for (int i = 0; i < 6; i++) desc[i] = d[i];
	return 0;
}

// `reps` launches of one compiled tiling (variant -1: the plain-VALU FMA-chain reference kernel) over the front end's own taps and the
// spectra / partial sums of the newest half (whatever the last blocks left there), `nb` blocks per launch, timed by the kernels' own
// events; *checksum = a 64-bit sum over the partial sums' bit patterns, equal across kernels when they are bit-identical.
extern "C" int hfdl_gpu_lab_fold_variant_probe(hfdl_gpu_frontend *fe, int variant, int nb, int reps, double *avg_ms, double *best_ms, uint64_t *checksum)
{
	if (!fe || !avg_ms || reps < 1 || nb < 1) return fail(HFDL_GPU_EINVAL, "bad arguments");
	int rc = hfdl_gpu_frontend_sync(fe);
	if (rc) return rc;
	if (nb > fe->half_blocks) return fail(HFDL_GPU_ERANGE, "%d blocks asked for, a half holds %d", nb, fe->half_blocks);
	const Geometry &g = fe->geo;
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipMemsetAsync(fe->d_partial, 0xff, sizeof(float2) * fe->partial_stride() * (size_t)nb, fe->stream));    // nothing left over from another kernel counts
	double sum = 0, best = 1e30;
	for (int i = 0; i < reps + 1; i++) {
		if (launch_fold_variant(variant, g, fe->d_taps, fe->spec_slot(fe->last_set, 0), (size_t)g.n, fe->d_partial, fe->partial_stride(), nb, fe->stream, e0, e1) < 0) {
			(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
			return fail(HFDL_GPU_ERANGE, "fold variant %d does not fit this geometry (M = %d, %d rows per slice) or block count %d", variant, g.m, g.rows_per_slice, nb);
		}
		HIP_TRY(hipEventSynchronize(e1));
		float ms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
		if (i > 0) { sum += ms; best = std::min(best, (double)ms); }       // first launch: code load
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	HIP_TRY(hipGetLastError());
	*avg_ms = sum / reps;
	if (best_ms) *best_ms = best;
	if (checksum) {
		const size_t words = 2 * fe->partial_stride() * (size_t)nb;
		std::vector<uint32_t> h(words);
		HIP_TRY(hipMemcpy(h.data(), fe->d_partial, sizeof(uint32_t) * words, hipMemcpyDeviceToHost));
		uint64_t acc = 0;
		for (size_t i = 0; i < words; i++) acc += (uint64_t)h[i] * (uint64_t)(2 * (i % 65521) + 1);
		*checksum = acc;
	}
	return 0;
}
#endif
