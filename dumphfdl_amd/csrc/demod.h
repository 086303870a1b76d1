// demod.h -- host-side owner of the per-channel demodulator / burst decoder device state (K4 + K5).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include "../../include/hfdl_gpu.h"
#include "kernels.h"

namespace hfdl {

struct ChanState;
struct FrameRec;

struct Demod {
	int nch = 0, outs = 0, cap = 0;         // cap = max 5400-sps samples per demodulator launch (`batch` blocks)
	int batch = 1;                          // blocks a launch can take: what was asked for, cut down to what fits the LDS
	float *d_tables = nullptr;              // packed DemodTables image
	ChanState *d_states = nullptr;
	float2 *d_data = nullptr;               // [nch][2][5040] equalised data symbols
	FrameRec *d_frames = nullptr;           // [2][nch]: frames finished by the demodulator of an even / odd block
	int *d_counts = nullptr;                // [1] pdus produced, [2] pdus dropped, [3] pdus taken by the host, [4..7] frames queued, one counter per block mod 4
	int *h_snap = nullptr;                  // pinned [2][4]: d_counts as of the end of the block that used buffer 0 / 1
	uint32_t taken = 0, dropped = 0;
	uint64_t launches = 0, decodes = 0;
	bool separate_decode = false;           // the burst decoder runs on another stream than the demodulator
	hipEvent_t ev_dec[2] = { nullptr, nullptr };   // decoder of an even / odd launch done: its frame queue and counter may be reused
	hfdl_gpu_pdu *d_pdus = nullptr;
	// Collection runs beside the kernels: device -> host copies go through page-locked bounce buffers on a stream of their own.
	// (A synchronous hipMemcpy waits for the kernels in flight -- up to a whole demodulator launch, ~1 ms on the small geometries,
	// every time a PDU is collected: profiles/r03_experiments.md.)
	hipStream_t st_collect = nullptr;
	hfdl_gpu_pdu *h_pdu_bounce = nullptr;   // pinned [bounce_cap]
	int bounce_cap = 0;
	void *h_stats_bounce = nullptr;         // pinned [nch] ChanScalars
	int32_t *d_freqs = nullptr;
	int pdu_cap = 0;
	// stage taps
	bool taps_on = true;                    // buffers allocated
	bool taps_enabled = true;               // written by the kernel this block
	float2 *d_tap_rs = nullptr, *d_tap_mf = nullptr, *d_tap_sym = nullptr;
	float *d_tap_lvl = nullptr;
	int *d_tap_counts = nullptr;            // [nch][2]
	size_t lds_bytes = 0;
	void *priv = nullptr;                   // DemodPriv (host image of the tables + resolved device pointers)

	static int fit_batch(int outs, float resamp_rate, int want);     // blocks a launch can take at most, `want` or fewer (LDS, 16-bit output counts, < 1 s of signal)
	int init(int nch, int outs, float resamp_rate, const int32_t *freqs, hipStream_t st, int batch_want = 1);
	// K4 of a block.  `done` (optional) is signalled by the kernel's own dispatch packet.  frames_free: the caller has already
	// ordered this launch after the decoder of launch i-2 (frames_free_event()), so no wait is queued in front of the kernel.
	// chan_out / out_count: `nblk` consecutive blocks, [nblk][nch][outs] and [nblk][nch]
	int enqueue_demod(const float2 *chan_out, const int *out_count, int nblk, hipStream_t st, hipEvent_t done = nullptr, bool frames_free = false, hipEvent_t start = nullptr);
	hipEvent_t frames_free_event() const { return separate_decode ? ev_dec[launches & 1] : nullptr; }   // of the NEXT launch; may be null
	int enqueue_decode(int buf, hipStream_t st, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);   // K5 + PDU-ring snapshot of the same block; the events ride on the kernel's dispatch (timing)
	int collect(hfdl_gpu_pdu *out, int32_t max, int32_t *n, hipStream_t st);                    // stream idle: everything produced
	int collect_snapshot(int buf, hfdl_gpu_pdu *out, int32_t max, int32_t *n, hipStream_t st);  // up to the end of that buffer's block
	int take(unsigned produced, hfdl_gpu_pdu *out, int32_t max, int32_t *n, hipStream_t st);
	int stats_all(hfdl_gpu_channel_stats *out, int n);
	int tap(int what, int channel, const void **src, size_t *nfloats);
	int stats(int channel, hfdl_gpu_channel_stats *out);
	int read_constants(void *tables, size_t tables_bytes, void *constants, size_t constants_bytes);   // laboratory read-back: sizeof(DemodTables), sizeof(HfdlConstants)
	void release();
};

#ifdef HFDL_LAB
int demod_clock_probe_read(unsigned long long *out, int max, int *n);      // laboratory: {tag, shader cycles, 100 MHz ticks, start tick} per probed launch
#endif
// kernel_ms (optional): time of the kernel launch alone, HIP events on the null stream
int demod_viterbi_batch(const uint8_t *soft, int32_t nbits, int32_t nframes, uint8_t *out, double *kernel_ms = nullptr);
int demod_crc16(const uint8_t *data, uint32_t len, uint16_t crc_init, uint16_t *crc);
int demod_pdu_triage_batch(const uint8_t *octets, const int32_t *lens, int32_t npdus, int32_t stride, uint8_t *fcs_status, uint8_t *kind, uint16_t *hdr_len);
int demod_psk_slice_batch(int arity, const float *xy, int32_t n, uint32_t *sym, float *phase_error);
int demod_lpdu_walk_batch(const uint8_t *octets, const int32_t *lens, int32_t npdus, int32_t stride, uint8_t *counts);
int demod_burst_decode_batch(const float *symbols, const int32_t *modes, const int32_t *bitmask_lsb, int32_t nframes,
		uint8_t *octets, int32_t *lens, double *kernel_ms = nullptr);

}  // namespace hfdl
