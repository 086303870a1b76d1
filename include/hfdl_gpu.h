/*
 * hfdl_gpu.h -- C ABI of the MI355X (gfx950) HFDL front end: libhfdl_gpu.so
 *
 * Drop-in boundary for dumphfdl's hot path.  Plain C types only: the host
 * program stays C (block / input-common / hfdl_channel API in include/hfdl_host.h) and binds these
 * entry points; they replace, for ALL channels of one receiver at once:
 *
 *   reference interface (file:line)                               -> entry point here
 *   ---------------------------------------------------------------------------------------------
 *   fft_create(decimation, transition_bw)          src/fft.h:31,  src/fft.c:70-86      \
 *   hfdl_channel_create(fs, decim, tbw, cf, freq)  src/hfdl.h:11, src/hfdl.c:468-534    } hfdl_gpu_frontend_create
 *   fft_channelizer_create(...)                    src/fastddc.h:42, src/fastddc.c:217 /
 *   fft_thread loop body: overlap + csdr_fft_execute + fft_swap_sides   src/fft.c:49-59 \
 *   hfdl_decoder_thread loop body: fastddc_inv_cc .. decode_user_data   src/hfdl.c:662-891 } hfdl_gpu_frontend_push_block
 *   dispatch_pdu -> pdu_decoder_queue_push(metadata, octet_string, 0)   src/hfdl.c:1058-1080 -> hfdl_gpu_frontend_poll_pdus
 *   fft_destroy / hfdl_channel_destroy             src/fft.h:32, src/hfdl.h:13          -> hfdl_gpu_frontend_destroy
 *   csdr_fft_execute(fwd)+fft_swap_sides           src/fft_fftw.c:39-41, src/fastddc.c:102 -> hfdl_gpu_fft_forward
 *   update_viterbi27_blk + chainback_viterbi27     src/libfec/fec.h:19-20               -> hfdl_gpu_burst_decode / hfdl_gpu_viterbi27
 *   decimating_shift_addition_init / _cc           src/libcsdr_gpl.h:43-44, src/libcsdr_gpl.c:26-74 -> hfdl_gpu_nco_decimate
 *   crc16_ccitt                                    src/crc.h, src/crc.c:4-47            -> hfdl_gpu_crc16_ccitt
 *   hfdl_pdu_fcs_check + header length rules       src/pdu.c:68-79, src/mpdu.c:56-79, src/spdu.c:55-62 -> hfdl_gpu_pdu_triage
 *   modem_demodulate + phase error (liquid PSK)    src/hfdl.c:737-741                   -> hfdl_gpu_psk_slice (the carrier loop's slicer)
 *   parse_lpdu_list + lpdu_parse's FCS check       src/mpdu.c:92-158, src/lpdu.c:136-149 -> hfdl_gpu_lpdu_walk (and hfdl_gpu_pdu.lpdus_*)
 *
 * All functions return 0 on success or a negative HFDL_GPU_E* code (the reference's constructors
 * return NULL / -1 and xcalloc failure _exit()s: src/util.c:25-33); hfdl_gpu_last_error() gives text.
 * There is no CPU fallback: if no gfx950 device is usable every call fails with HFDL_GPU_ENODEV.
 *
 * Threading: a front end is driven by ONE thread at a time (the reference's fft_thread; src/fft.c:30-66); different front
 * ends -- one per GPU -- are independent.  hfdl_gpu_last_error() is per calling thread.
 */
#ifndef HFDL_GPU_H
#define HFDL_GPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HFDL_GPU_EINVAL   (-1)
#define HFDL_GPU_ENODEV   (-2)
#define HFDL_GPU_ENOMEM   (-3)
#define HFDL_GPU_EHIP     (-4)
#define HFDL_GPU_ERANGE   (-5)

#define HFDL_GPU_PDU_MAX_OCTETS 960

/* on-device header triage of a decoded PDU (what mpdu_parse / spdu_parse do first: src/mpdu.c:56-89, src/spdu.c:55-62) */
#define HFDL_GPU_FCS_GOOD      0
#define HFDL_GPU_FCS_BAD       1
#define HFDL_GPU_FCS_TOO_SHORT 2
#define HFDL_GPU_KIND_SPDU          0
#define HFDL_GPU_KIND_MPDU_DOWNLINK 1
#define HFDL_GPU_KIND_MPDU_UPLINK   2

typedef struct hfdl_gpu_frontend hfdl_gpu_frontend;

/* block geometry: the fields of the reference's fastddc_t (src/fastddc.h:8-27) for shift = 0 */
typedef struct {
	int32_t sample_rate, decimation;
	int32_t pre_decimation, post_decimation;
	int32_t taps_length, overlap_length;
	int32_t fft_size, fft_inv_size, input_size, post_input_size, scrap;
	int32_t outputs_per_block;       /* post_input_size / post_decimation */
	int32_t channels;
	int32_t fold_slices;             /* alias-row slices per channel in the fold kernel */
	float   transition_bw;
	float   resamp_rate;             /* 5400 / (fs / decimation) */
	int32_t max_outputs_per_block;   /* ceil(post_input_size / post_decimation): what a block can emit when the carried
	                                    decimation remainder is non-zero (post_input_size not a multiple of post_decimation);
	                                    size HFDL_GPU_TAP_CHAN_OUT buffers from this */
	int32_t demod_batch;             /* blocks one demodulator launch takes when they are pushed faster than they are collected (up to one second
	                                    of signal, cut down to what fits the LDS: 3 at 40 Msps, up to 8 on the small geometries).  Results do not
	                                    depend on it; a poll / sync always demodulates what has been pushed.  HFDL_GPU_DEMOD_BATCH=1..8 overrides
	                                    the default at create time.  0 from hfdl_gpu_plan_geometry() */
	int32_t fold_batch;              /* blocks whose spectra one fold launch multiplies against ONE pass over the per-channel filter taps when
	                                    they are pushed faster than they are collected (src/fastddc.c:123-150 run for that many blocks; the taps are
	                                    > 99 % of a block's bytes on the 256-channel geometries).  Every block's result is bit-identical to a launch
	                                    of its own; a poll / sync always folds what has been pushed.  HFDL_GPU_FOLD_BATCH=1..32 overrides the default
	                                    (32 from 128 channels up, where the fold bounds the block -- the first half after a drain closes at 16 --; 8 below) at create time.  0 from hfdl_gpu_plan_geometry() */
	int32_t prefetch_depth;          /* host blocks whose upload hfdl_gpu_frontend_prefetch_block_raw() may queue ahead of their push
	                                    (half + 1, for a staging ring of half + 2 buffers in HBM, where a half = the blocks between two fold / inverse-FFT
	                                    phases = fold_batch rounded up to hold at least demod_batch blocks); at most HFDL_GPU_PREFETCH_MAX (a 32-block
	                                    half is not uploaded a whole half ahead: 17 blocks of link time cover its fold several times over).
	                                    0 from hfdl_gpu_plan_geometry() */
	int32_t fold_rows;               /* alias rows a fold workgroup adds up at most: pre_decimation (all of them, the reference's sum term for
	                                    term) unless HFDL_GPU_FOLD_PRUNE is set.  0 from hfdl_gpu_plan_geometry() */
} hfdl_gpu_geometry;
#define HFDL_GPU_PREFETCH_MAX 17
#define HFDL_GPU_FOLD_BATCH_MAX 32   /* blocks one fold launch takes at most: two column groups of the sixteen-column matrix instruction */

/* Create-time configuration read from the environment by hfdl_gpu_frontend_create() (every create reads it afresh; nothing is cached):
 *   HFDL_GPU_FOLD_BATCH   1..32  blocks per fold launch (geometry.fold_batch)
 *   HFDL_GPU_DEMOD_BATCH  1..8   blocks per demodulator launch (geometry.demod_batch)
 *   HFDL_GPU_HOST_THREADS >= 1   host threads that design the filter taps (default: one per core; set to cores / processes when several
 *                                front ends are created at once on one host)
 *   HFDL_GPU_PDU_RING     >= 1   capacity of the device PDU ring (default max(4096, 64 per channel))
 *   HFDL_GPU_FOLD_PRUNE   tol    (default unset: off) the pruned fold: a channel's filter is a band-pass with a windowed-sinc stop band, and
 *                                of the N / M alias rows the reference adds up (src/fastddc.c:123-150) all but the few around the pass band hold
 *                                taps at the level of their own rounding noise.  With tol set (1e-12 .. 1e-3) a workgroup folds only
 *                                the rows outside which its channels' filters hold less than tol^2 of their energy (geometry.fold_rows of
 *                                pre_decimation): the channelizer output moves by ~tol relative RMS and the fold's time by the ratio of the
 *                                rows.  The stored taps carry the rounding noise of the fp32 transform that made them (2.1e-7 of their energy
 *                                as an amplitude ratio at N = 2^23) in EVERY row: a tolerance below that keeps every row.  Off, every row is
 *                                folded -- the reference's sum, term for term
 * The A/B switches of the measurement scripts (stream placement, tiling sweeps, probes) are in the laboratory build only:
 * include/hfdl_gpu_lab.h, libhfdl_gpu_lab.so. */

/* one decoded PDU: what dispatch_pdu() hands to pdu_decoder_queue_push (src/hfdl.c:1058-1080,
 * struct hfdl_pdu_metadata src/pdu.h:8-17).  The wall-clock rx_timestamp of the reference is
 * replaced by the 5400-sps sample index at A2 detection (reproducible on file input). */
typedef struct {
	int32_t  channel;                /* index into the freqs[] given at create */
	int32_t  freq;                   /* Hz */
	int32_t  mode;                   /* M1 index 0..7 */
	int32_t  bit_rate;
	int32_t  len;                    /* octets */
	float    freq_err_hz, rssi_db, noise_floor_db;
	char     slot;                   /* 'S' / 'D' */
	uint8_t  fcs_status;             /* HFDL_GPU_FCS_*: header FCS checked on the device (src/pdu.c:68-79) */
	uint8_t  pdu_kind;               /* HFDL_GPU_KIND_*: SPDU / MPDU direction triage (src/pdu.c:124-128, src/mpdu.c:56-74) */
	uint16_t hdr_len;                /* octets covered by the FCS */
	uint64_t sample_index;
	int32_t  train_bits_bad, train_bits_total;
	/* MPDUs with a good header FCS: the LPDU list walked on the device (parse_lpdu_list, src/mpdu.c:136-158) with every LPDU's own
	 * FCS checked (lpdu_parse, src/lpdu.c:136-149) -- the reference's lpdus.processed / lpdus.good / lpdu.errors.bad_fcs /
	 * lpdu.errors.too_short events of this PDU; lpdus_truncated = an announced LPDU runs past the PDU.  All 0 otherwise. */
	uint8_t  lpdus_processed, lpdus_good, lpdus_bad_fcs, lpdus_too_short, lpdus_truncated, lpdu_pad[3];
	uint8_t  octets[HFDL_GPU_PDU_MAX_OCTETS];
} hfdl_gpu_pdu;

/* ---- planning (host only, no device needed) ---- */

/* fastddc_init(.., shift = 0) as fft_create() runs it (src/fft.c:70-86, src/fastddc.c:46-80): block geometry for a
 * decimation / relative transition bandwidth pair.  channels / fold_slices / sample_rate / resamp_rate are left 0. */
int  hfdl_gpu_plan_geometry(int32_t decimation, float transition_bw, hfdl_gpu_geometry *g);

/* page-locked host memory for block staging (the role of the reference's ring read pointer, src/fft.c:50-53) */
int  hfdl_gpu_host_alloc(void **ptr, size_t bytes);
void hfdl_gpu_host_free(void *ptr);

/* ---- whole front end ---- */

/* frequencies in Hz as in hfdl_channel_create(); decimation / transition_bw are derived exactly as
 * main() does (src/main.c:699-704): compute_fft_decimation_rate(fs, 5400), 250 Hz / fs. */
int  hfdl_gpu_frontend_create(hfdl_gpu_frontend **out, int device, int32_t sample_rate, int32_t centerfreq,
		const int32_t *freqs, int32_t nch);
void hfdl_gpu_frontend_destroy(hfdl_gpu_frontend *fe);
int  hfdl_gpu_frontend_geometry(const hfdl_gpu_frontend *fe, hfdl_gpu_geometry *g);

/* Enqueue one block: exactly geometry.input_size new complex samples (interleaved I,Q float32).  Asynchronous: the block's forward
 * FFT is queued at once; fold, inverse FFT, demodulator and burst decoder follow when geometry.fold_batch blocks are waiting or the
 * caller syncs / polls, whichever comes first (the results do not depend on which).
 * on_device != 0: `iq` is a device pointer that stays valid until the next sync.
 * on_device == 0: the host -> device copy runs on its own stream into a ring of geometry.prefetch_depth + 1 staging buffers, so the copies
 *   run up to a whole half (fold_batch blocks, or more where demod_batch is larger) ahead of the kernels.  A buffer from hfdl_gpu_host_alloc() (page-locked) is read by DMA after the call
 *   returns: reuse it only after hfdl_gpu_frontend_input_done() / _sync() / _poll_pdus().  Any other host buffer (pageable,
 *   or registered by the caller) is waited for inside the call and may be reused as soon as it returns. */
int  hfdl_gpu_frontend_push_block(hfdl_gpu_frontend *fe, const float *iq, size_t nsamples, int on_device);
/* wait until every host -> device input copy enqueued so far has finished (the kernels keep running) */
int  hfdl_gpu_frontend_input_done(hfdl_gpu_frontend *fe);
/* wait until the copy of host block number `host_block` (0 = the first block pushed with on_device == 0) has finished, WITHOUT
 * waiting for the blocks pushed after it: a caller that leases two page-locked buffers pushes block k+1, then waits for block k and
 * reuses its buffer -- the copy engine never idles on the host thread */
int  hfdl_gpu_frontend_input_done_upto(hfdl_gpu_frontend *fe, uint64_t host_block);
/* the same question without waiting: 1 = the copy of that host block has finished, 0 = not yet, negative = error */
int  hfdl_gpu_frontend_input_copied(hfdl_gpu_frontend *fe, uint64_t host_block);
/* Queue the host -> device copy of a block that will be pushed LATER (a buffer from hfdl_gpu_host_alloc(), host pointer) without
 * pushing it: the copy then runs beside the blocks still computing -- the fold of a 40 Msps x 256-channel half takes 3 ms, five blocks
 * of PCIe time -- and the following hfdl_gpu_frontend_push_block_raw() of the same pointer and format only queues kernels.  Up to
 * geometry.prefetch_depth blocks may wait this way (HFDL_GPU_ERANGE beyond); they must be pushed in the order they were prefetched.
 * A prefetched block counts as a host block for hfdl_gpu_frontend_input_done_upto() from this call on. */
int  hfdl_gpu_frontend_prefetch_block_raw(hfdl_gpu_frontend *fe, const void *raw, size_t nsamples, int sample_format);
/* Forget the prefetched blocks (a caller that hit an error between a prefetch and its push): waits for the copies, after which the
 * buffers are the caller's again and any block may be pushed next.  The blocks keep their host block numbers.  No-op without a prefetch.
 * While a prefetch is pending, a push of anything but the oldest prefetched block -- another pointer, another format, a device block --
 * is HFDL_GPU_EINVAL and leaves the queue in place. */
int  hfdl_gpu_frontend_prefetch_cancel(hfdl_gpu_frontend *fe);
/* Same, for raw recorder / SDR samples converted on the device inside the overlap-assembly load of the forward FFT
 * (convert_cs16 / convert_cu8 / convert_cf32, src/input-helpers.c:10-78): interleaved I,Q int16 (full scale 32767.5),
 * uint8 (offset 63.5, full scale 127) or float32.  Halves / quarters the host->device bytes per sample. */
#define HFDL_GPU_SFMT_CF32 0
#define HFDL_GPU_SFMT_CS16 1
#define HFDL_GPU_SFMT_CU8  2
int  hfdl_gpu_frontend_push_block_raw(hfdl_gpu_frontend *fe, const void *raw, size_t nsamples, int sample_format, int on_device);
/* the demodulator + burst decoder stage alone (for stage parity): one block of channelizer OUTPUT from host memory -- chan_out[channels]
 * [max_outputs_per_block + 1] complex samples (interleaved re, im; rows of that length whatever counts[] says), counts[c] of them valid
 * for channel c -- through the same kernels and carried channel state as a pushed block's (hfdl_decoder_thread's loop body after
 * fastddc_inv_cc, src/hfdl.c:676-892).  Syncs first; collect with hfdl_gpu_frontend_poll_pdus(). */
int  hfdl_gpu_frontend_push_baseband(hfdl_gpu_frontend *fe, const float *chan_out, const int32_t *counts);
/* run only the channelizer part of a block (forward FFT + fold + inverse FFT + NCO); for stage parity/bench */
int  hfdl_gpu_frontend_channelize_block(hfdl_gpu_frontend *fe, const float *iq, size_t nsamples, int on_device);
int  hfdl_gpu_frontend_sync(hfdl_gpu_frontend *fe);
/* Collect PDUs produced by all blocks enqueued so far (implies a sync). Returns count in *n.  `out` must hold `max`
 * entries; out == NULL with max > 0 is HFDL_GPU_EINVAL (nothing is discarded), max == 0 just syncs. */
int  hfdl_gpu_frontend_poll_pdus(hfdl_gpu_frontend *fe, hfdl_gpu_pdu *out, int32_t max, int32_t *n);
/* Same without draining the pipeline: with max_in_flight = 1 whatever was pushed since the last half filled keeps filling, the newest
 * CLOSED half (geometry.fold_batch blocks, or max(fold_batch, demod_batch)) keeps running; the call waits only for the demodulators of the
 * half before it and returns the PDUs known to be complete at that moment -- those of that half if its burst decoders have finished too,
 * otherwise they come with the next call (nothing until two halves were closed).  max_in_flight = 0 is hfdl_gpu_frontend_poll_pdus();
 * max_in_flight >= 2 does not wait at all: if those demodulators are still running the call returns no PDUs (for a caller that bounds
 * what it queues by other means -- the C host program by the ring slots it leases to the uploads).
 * A file replay pushes block k+1, then collects this way, so copies, channelizer, demodulator and burst decoder of consecutive halves
 * overlap and the fold shares its pass over the filter taps between the blocks of a half; a live receiver that has no further input
 * queued uses 0 and gets its PDUs at once. */
int  hfdl_gpu_frontend_poll_pdus_ready(hfdl_gpu_frontend *fe, hfdl_gpu_pdu *out, int32_t max, int32_t *n, int32_t max_in_flight);
/* PDUs wait in a device ring of pdu_ring_capacity entries; one that finds the ring full is dropped and counted
 * (the reference's GAsyncQueue is unbounded, src/pdu.c:37-43: poll at least once per few seconds of signal).
 * Capacity: max(4096, 64 per channel); the environment variable HFDL_GPU_PDU_RING overrides it at create time. */
typedef struct {
	uint64_t blocks;                 /* blocks pushed so far */
	uint32_t pdus_taken;             /* PDUs handed to the host so far (mod 2^32) */
	uint32_t pdus_dropped;           /* as of the last poll */
	uint32_t pdu_ring_capacity;
} hfdl_gpu_frontend_counters_t;
int  hfdl_gpu_frontend_counters(hfdl_gpu_frontend *fe, hfdl_gpu_frontend_counters_t *out);
/* the HIP stream the channelizer kernels (forward FFT, fold, inverse FFT) are launched on (hipStream_t as void*); device
 * input handed to push_block must be complete on, or synchronised with, this stream */
void *hfdl_gpu_frontend_stream(hfdl_gpu_frontend *fe);

/* per-channel observability: the hot-path StatsD counters and the noise-floor gauge of the reference
 * (src/hfdl.c:818-840, 1082-1105; doc/STATSD_METRICS.md), read from the device-resident channel state */
typedef struct {
	int32_t  freq;
	uint32_t a2_found, m1_found, m1_not_found, frames;
	float    noise_floor_db;         /* 20 log10(noise_floor), what noise_floor_stats_thread reports */
	float    agc_level, costas_dphi;
	int32_t  framer_state;           /* 1 = A1 search ... 7 = DATA_2 (src/hfdl.c:54-62) */
	uint64_t sample_cnt, symbol_cnt;
	/* the reference's debug summary (hfdl_print_summary, src/hfdl.c:563-573), per channel: A1 detections, mean |correlation| at the
	 * A1 / A2 / M1 detections (0 when there were none), training bits of all frames received */
	uint32_t a1_found;
	float    a1_corr_avg, a2_corr_avg, m1_corr_avg;
	uint32_t train_bits_bad, train_bits_total;
} hfdl_gpu_channel_stats;
int  hfdl_gpu_frontend_channel_stats(hfdl_gpu_frontend *fe, int32_t channel, hfdl_gpu_channel_stats *out);
/* every channel in one strided device read, WITHOUT waiting for blocks in flight: each field is read whole, the set may
 * straddle a block boundary (what a periodic gauge needs: noise_floor_stats_thread, src/hfdl.c:1082-1105) */
int  hfdl_gpu_frontend_all_channel_stats(hfdl_gpu_frontend *fe, hfdl_gpu_channel_stats *out, int32_t cap, int32_t *n);

/* stage taps -- the DATADUMPS analogue (src/hfdl.c:616-644): copy an intermediate buffer to host */
enum {
	HFDL_GPU_TAP_SPECTRUM = 1,       /* cf32[fft_size], fftshifted forward FFT (shared.buf after src/fft.c:59) */
	HFDL_GPU_TAP_FILTER = 2,         /* cf32[fft_size], filtertaps_fft of `channel` */
	HFDL_GPU_TAP_CHAN_OUT = 3,       /* cf32[<= max_outputs_per_block], fastddc_inv_cc output of `channel` */
	HFDL_GPU_TAP_RESAMPLED = 4,      /* cf32[n], msresamp output of the last block */
	HFDL_GPU_TAP_MF_OUT = 5,         /* cf32[n], AGC + matched filter output of the last block */
	HFDL_GPU_TAP_SYMBOLS = 6,        /* cf32[n], equalised on-time symbols of the last block */
	HFDL_GPU_TAP_AGC_LEVEL = 7,      /* f32[n], agc signal level per 5400-sps sample */
	HFDL_GPU_TAP_NCO_PHASORS = 9,    /* cf32[n]: the NCO phasors (cos phi_k, sin phi_k) that multiplied the last block's outputs of `channel`
	                                    (decimating_shift_addition_cc's recurrence, src/libcsdr_gpl.c:48-66; bit-exact vs the reference) */
	HFDL_GPU_TAP_PHASE_CYCLES = 8    /* f32[4]: shader cycles of the last demod launch: resampler phase, the whole three-wave pipelined phase (wall),
	                                    busy cycles of the timing-recovery wave, busy cycles of the carrier / equaliser / framer wave */
};
/* Stage taps 4..8 make the demodulator write its intermediate samples to HBM every launch; on by default (tests), a
 * production caller / the bench turns them off.  Taps 1..3 are always available (they are the kernels' own buffers).  Taps 4..8 hold
 * the LAST demodulator launch: one block when the caller syncs / polls after every push, up to geometry.demod_batch blocks otherwise
 * (size buffers for demod_batch * (post_input_size + 64) complex samples); tap 3 and tap 9 hold the last block. */
int  hfdl_gpu_frontend_enable_taps(hfdl_gpu_frontend *fe, int enable);
/* dst holds `cap` floats; *n_floats receives the number written */
int  hfdl_gpu_frontend_read_tap(hfdl_gpu_frontend *fe, int what, int32_t channel, float *dst, size_t cap, size_t *n_floats);
/* taps 1, 3 and 9 of the block `back` blocks before the newest one (0 = read_tap): the channelizer keeps the blocks of the newest half
 * -- what was pushed since the half before it filled (geometry.fold_batch blocks, or max(fold_batch, demod_batch)) or since the last
 * sync / poll; HFDL_GPU_ERANGE beyond that */
int  hfdl_gpu_frontend_read_tap_block(hfdl_gpu_frontend *fe, int what, int32_t channel, int32_t back, float *dst, size_t cap, size_t *n_floats);

/* timing of the dominant kernel (fold) measured with HIP events on the front end's stream; a launch folds up to geometry.fold_batch
 * blocks: hfdl_gpu_frontend_fold_blocks() = the blocks the timed launches covered */
int  hfdl_gpu_frontend_fold_time_ms(hfdl_gpu_frontend *fe, double *total_ms, int64_t *launches);
int  hfdl_gpu_frontend_fold_blocks(hfdl_gpu_frontend *fe, int64_t *blocks);
/* timed fold launches by block count: counts[nb] = launches that folded nb blocks (nb = 1 .. HFDL_GPU_FOLD_BATCH_MAX; counts[0] unused)
 * since the timers were reset, ms[nb] (may be NULL) = their kernel time -- what a caller needs to price the launches it timed (a launch
 * of up to 4 blocks runs the four-column form of the kernel and is bound by the HBM reads of the taps, one of 5 .. 16 the sixteen-column
 * form, one of 17 .. 32 the thirty-two-column form: every loaded tap operand multiplies two spectrum operands) */
int  hfdl_gpu_frontend_fold_launch_shapes(hfdl_gpu_frontend *fe, int64_t counts[HFDL_GPU_FOLD_BATCH_MAX + 1], double ms[HFDL_GPU_FOLD_BATCH_MAX + 1]);
int  hfdl_gpu_frontend_reset_timers(hfdl_gpu_frontend *fe, int enable);
/* the same for the demodulator kernel (the kernel that bounds the small geometries), from its dispatch's own start / stop events;
 * *blocks = the blocks those launches covered (a launch takes up to geometry.demod_batch blocks) */
int  hfdl_gpu_frontend_demod_time_ms(hfdl_gpu_frontend *fe, double *total_ms, int64_t *launches, int64_t *blocks);
/* kernel time by stage since the timers were reset, from start / stop events that ride on the dispatches themselves:
 * ms / launches [0] forward FFTs (first pass start -> last pass stop, one per block), [1] fold launches, [2] inverse FFT / NCO launches
 * (one per half) -- together the channelizer's stream; [3] demodulator launches -- their own stream; [4] burst decoder launches -- theirs.
 * Kernels of different streams overlap: the sums say which stream bounds a half, not what a block costs. */
int  hfdl_gpu_frontend_stage_times(hfdl_gpu_frontend *fe, double ms[5], int64_t launches[5]);
/* steady-state period of one block: (start of the last timed fold launch - start of the first) / (blocks folded by all timed launches
 * but the last), free of the pipeline fill before the first block and the demodulator / burst-decoder drain after the last */
int  hfdl_gpu_frontend_step_period_ms(hfdl_gpu_frontend *fe, double *period_ms);
/* ---- stage-level entry points (host pointers; allocate / copy / free internally) ---- */

/* out[(k + n/2) mod n] = sum_t in[t] e^{-2 pi i k t / n} when shifted != 0 (plain order otherwise); n = power of two >= 512 */
int  hfdl_gpu_fft_forward(int device, const float *in, float *out, int32_t n, int shifted);
/* K=7 r=1/2 Viterbi on `nframes` frames of equal size: soft = nframes * 2*nbits bytes; out = nframes * ceil(nbits/8) */
int  hfdl_gpu_viterbi27(int device, const uint8_t *soft, int32_t nbits, int32_t nframes, uint8_t *out);
/* decode_user_data for a batch: symbols = nframes * (segments*30) equalised data symbols (cf32), one mode and
 * bitmask bit per frame; octets = nframes * HFDL_GPU_PDU_MAX_OCTETS, lens[nframes] */
int  hfdl_gpu_burst_decode(int device, const float *symbols, const int32_t *modes, const int32_t *bitmask_lsb,
		int32_t nframes, uint8_t *octets, int32_t *lens);

/* decimating_shift_addition_cc(input, output, input_size, decimating_shift_addition_init(rate, decimation), decimation, status)
 * (src/libcsdr_gpl.c:26-74): out[k] = in[remain + decimation k] e^{j phi_k} with the reference's fp32 phasor recurrence; the
 * status fields (*decimation_remain, *starting_phase) are read and updated, *output_size receives k.  `out` holds
 * ceil(input_size / decimation) complex samples.  This is the device code the channelizer runs after its inverse FFT. */
int  hfdl_gpu_nco_decimate(int device, const float *in, int32_t input_size, float rate, int32_t decimation,
		int32_t *decimation_remain, float *starting_phase, float *out, int32_t *output_size);
/* crc16_ccitt(data, len, crc_init) of src/crc.c:4-47, computed by the device function the burst decoder checks every FCS with */
int  hfdl_gpu_crc16_ccitt(int device, const uint8_t *data, uint32_t len, uint16_t crc_init, uint16_t *crc);
/* header triage of `npdus` decoded PDUs, PDU i = octets[i * stride .. + lens[i]): fcs_status[i] = HFDL_GPU_FCS_*,
 * pdu_kind[i] = HFDL_GPU_KIND_*, hdr_len[i] = octets covered by the FCS (what burst decoding fills into hfdl_gpu_pdu) */
int  hfdl_gpu_pdu_triage(int device, const uint8_t *octets, const int32_t *lens, int32_t npdus, int32_t stride,
		uint8_t *fcs_status, uint8_t *pdu_kind, uint16_t *hdr_len);

/* the LPDU list walk of `npdus` PDUs (must be MPDUs or SPDUs as they come out of the decoder): counts[i * 5 + 0..4] = lpdus processed,
 * good, bad FCS, too short, truncated flag -- zeros when the header triage of PDU i is not "FCS good" */
int  hfdl_gpu_lpdu_walk(int device, const uint8_t *octets, const int32_t *lens, int32_t npdus, int32_t stride, uint8_t *counts);

/* The carrier loop's slicer on `n` equalised symbols (interleaved re, im): modem_demodulate of liquid's PSK modem (arity 1..3 bits
 * per symbol: the Gray-coded symbol) and its demodulator phase error Im(x conj(x_hat)), src/hfdl.c:737-741.  The device decides by
 * the nearest constellation point (largest Re(x conj(p))), which is what arg() + the reference ladder computes; the two can differ
 * only for x within rounding of a decision boundary. */
int  hfdl_gpu_psk_slice(int device, int32_t arity, const float *xy, int32_t n, uint32_t *sym, float *phase_error);

/* kernel time in ms of the last hfdl_gpu_fft_forward / _viterbi27 / _burst_decode call made by this thread (HIP events
 * around the launch; allocation and host <-> device copies excluded) */
double hfdl_gpu_last_stage_ms(void);

const char *hfdl_gpu_last_error(void);
int  hfdl_gpu_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
