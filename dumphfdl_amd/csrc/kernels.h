// kernels.h -- device-side contracts shared by the HIP translation units of libhfdl_gpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hfdl {

constexpr int FFT_TILE_LOG = 4;
constexpr int FFT_TILE = 1 << FFT_TILE_LOG;   // columns per workgroup in the strided FFT passes (8-byte samples: 128-byte runs at 16)
constexpr int FFT_THREADS = 512;            // measured: 512 threads per 256x16 tile beat 256 and 1024 (profiles/r01_experiments.md)

// N = R1*R2*R3 three-pass plan for the wideband forward FFT (all radices powers of two <= 256)
struct FftPlan {
	int n, logn;
	int r1, r2, r3, l1, l2, l3;
	const float2 *tw1, *tw2, *tw3;     // W_R^t = exp(-2 pi i t / R), t < R, computed on the host in double
};

// per-channel channelizer constants (what fastddc_t holds for one channel, src/fastddc.h:8-27)
struct ChanConst {
	int32_t offsetbin;                 // startbin - N/2
	float nco_sindelta, nco_cosdelta;  // shift_addition_data_t
	float nco_rate;
	int32_t frequency;
};

// carried NCO state, decimating_shift_addition_status_t (src/libcsdr_gpl.h:35-40)
struct NcoState {
	int32_t decimation_remain;
	float starting_phase;
	int32_t output_size;
	int32_t pad;
};

struct Geometry {
	int32_t n, m, pre, post, scrap, post_input_size, overlap, input_size, outs;
	int32_t slices, rows_per_slice, nch;
	int32_t nch_pad;                   // channels the tap buffer holds: nch rounded up to a whole group of the tap layout (the extra ones all zero)
	int32_t tap_layout;                // TAPL_*: how an alias row of the filter taps lies in HBM (fold_kernels.hip)
	// an alias row of ALL channels is tap_row_stride cf32 long (nch_pad * M); TAPL_PLAIN: channel c at c*tap_chan_stride inside it,
	// bins in order; TAPL_OCTET: tap_index_f()
	int64_t tap_chan_stride, tap_row_stride;
	// The pruned fold (include/hfdl_gpu.h, HFDL_GPU_FOLD_PRUNE): per channel octet the window of alias rows, in quads of rows (first
	// quad, count; circular), outside which the filters of those channels hold less than the tolerated share of their energy.
	// Null: every row is folded.  With windows the geometry has one slice.
	const int2 *fold_win = nullptr;
	int32_t fold_tile = -1;            // laboratory (HFDL_GPU_FOLD_TILE): index of the tiling to use instead of the first that fits
};

// Filter-tap layouts.  The fold runs on the fp32 matrix pipe -- v_mfma_f32_16x16x4_f32: D (16 x 16) += A (16 x 4) . B (4 x 16) with
// the 16 rows = Re / Im of eight channels, the 4 inner indices = four consecutive alias rows, the 16 columns = sixteen blocks, for ONE
// bin -- and one wave load of 1 KiB (16 bytes per lane) must yield, register by register, operand A of an instruction: the taps are
// stored in operand order (the forward FFT that makes them writes it directly).
//   TAPL_OCTET: 1 KiB = 4 alias rows x 8 channels x 4 bins: lane = 16 (row % 4) + 2 (c % 8) + comp, register j % 4; the bin quads of an
//               octet follow each other (M / 4 KiB), then the octets, then the next four rows
//   TAPL_PLAIN: rows of M cf32 per channel (geometries whose M is no multiple of 16 or whose row count is no multiple of 4)
enum { TAPL_PLAIN = 0, TAPL_OCTET = 2 };
// float index of (alias row, channel c, bin j, comp 0 = Re / 1 = Im) in the tap buffer; rs_f = floats per alias row of ALL channels
// (2 * tap_row_stride), chan_stride_f = floats between channels of a row (TAPL_PLAIN)
__host__ __device__ inline size_t tap_index_f(int layout, int m, size_t rs_f, int c, int row, int j, int comp)
{
	if (layout == TAPL_OCTET)
		return (size_t)(row >> 2) * 4 * rs_f + (size_t)(c >> 3) * 64 * m + (size_t)(j >> 2) * 256 + (size_t)(16 * (row & 3) + 2 * (c & 7) + comp) * 4 + (j & 3);
	return (size_t)row * rs_f + ((size_t)c * m + j) * 2 + comp;
}
inline int tap_layout_group(int layout) { return layout == TAPL_OCTET ? 8 : 1; }

constexpr int FOLD_MAX_BLOCKS = 32;         // blocks one fold launch can take: two column groups of the sixteen-column matrix instruction

// The NCO phasor table of a block, made while that block's forward FFT runs.  decimating_shift_addition_cc's phasor recurrence
// (src/libcsdr_gpl.c:48-66) is 1792 strictly serial fp32 steps per channel at cfg3 -- 47 us for a lone lane, which used to be the
// whole duration of the inverse-FFT kernel.  It depends on the carried NcoState only: one extra workgroup per FFT pass launch runs
// a third of it, LANES OVER CHANNELS (256 channels = 4 wavefronts), hidden inside the pass.  Table layout [output index][channel]
// so the lanes' stores coalesce.
// The riders also OWN the carried state (decimating_shift_addition_status_t): the forward FFTs of several blocks are queued before
// the first of their inverse FFTs runs (fold batching, hfdl_gpu.cpp), so the state cannot wait for that kernel.  Segment 0 leaves the
// state the block starts from in `snap` (what the block's inverse-FFT kernel reads), the last segment advances `chain` (:67-72).
struct NcoJob {
	const ChanConst *cc = nullptr;     // null: no rider workgroup on this launch
	NcoState *chain = nullptr;         // [nch] carried state: before this block on entry, after it once the last segment has run
	NcoState *snap = nullptr;          // [nch] copy of the state before this block, for its inverse-FFT / NCO kernel
	float2 *ph = nullptr;              // [outs][nch] phasor table of the block
	float2 *cont = nullptr;            // [nch] phasor at the start of the next segment
	int32_t nch = 0, outs = 0, post_input_size = 0, post = 0;
	int32_t seg = 0, nseg = 3;
};

// owning device allocation for the one-shot stage entry points (freed on every exit path)
struct DevBuf {
	void *p = nullptr;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	~DevBuf() { if (p) (void)hipFree(p); }
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
	template <typename T> T *as() const { return (T *)p; }
};

// ---- launchers (host side, defined next to their kernels) ----
// sample formats of the raw ingest path (reference: src/input-helpers.c:10-78,108-125)
enum { SFMT_CF32 = 0, SFMT_CS16 = 1, SFMT_CU8 = 2 };
// output index i of the transform is stored at (i >> row_log) * row_stride + (i & (2^row_log - 1)); row_log = 0: contiguous.
// kind != TAPL_PLAIN: the transform is the filter of channel `chan`, `out` is the tap buffer, and element (row i >> row_log, bin) goes
// to tap_index_f(kind, 2^row_log, 2 * row_stride, chan, row, bin, comp) floats
struct FftOutLayout { int row_log = 0; int64_t row_stride = 0; int kind = 0; int chan = 0; };
void launch_fft_forward(const FftPlan &p, const float2 *hist, const void *fresh, int fmt, int split, float2 *hist_next,
		float2 *work, float2 *out, bool shifted, hipStream_t st, FftOutLayout lay = FftOutLayout(), hipEvent_t done = nullptr,
		NcoJob nco = NcoJob(), hipEvent_t input_read = nullptr, hipEvent_t start = nullptr);
		// input_read: signalled when the first pass has consumed `fresh`; start: rides on the first pass' dispatch (timing)
// optional events ride on the kernel dispatches themselves (hipExtLaunchKernelGGL): no separate barrier packets in the queue
// `nb` consecutive blocks: spectra `spec_stride` cf32 apart, partial sums `partial_stride` apart; launches of at most `nb_max` blocks
// sharing one pass over the taps.  Returns the number of kernel launches made.
int launch_fold(const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial, size_t partial_stride,
		int nb, int nb_max, hipStream_t st, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// the compiled tilings (the laboratory build carries the sweep set, profiles/fold_variants.py); launch one of them on `nb` spectra.
// variant -1 = the plain-VALU FMA-chain reference kernel every tiling must equal bit for bit
int fold_variant_count();
int fold_variant_describe(int variant, int desc[6]);            // P, Q, W, D, max blocks, 0
int launch_fold_variant(int variant, const Geometry &g, const float2 *taps, const float2 *spectrum, size_t spec_stride, float2 *partial,
		size_t partial_stride, int nb, hipStream_t st, hipEvent_t start, hipEvent_t stop);
void launch_tap_extract(const float2 *taps, const Geometry &g, int channel, float2 *dst, hipStream_t st);      // one channel's taps back in plain order
// energy[row * nch_pad + c] += sum over the row's M bins of |H_c|^2 (TAPL_OCTET; `energy` zeroed by the caller): what the pruned fold's
// row windows are chosen from
void launch_tap_row_energy(const float2 *taps, const Geometry &g, float *energy, hipStream_t st);
hipError_t prepare_ifft_nco(int m);     // LDS attribute of the inverse-FFT kernel for this size (checked at create time)
// `nb` blocks in one launch (grid nch x nb): partial sums, carried-state snapshots, phasor tables [outs][nch], outputs and counts of
// consecutive blocks lie `partial_stride` / nch / `ph_stride` / nch * outs / nch apart
void launch_ifft_nco(const Geometry &g, const float2 *partial, size_t partial_stride, const ChanConst *cc, const NcoState *snap, const float2 *ph,
		size_t ph_stride, const float2 *tw_m, float2 *chan_out, int *out_count, int nb, hipStream_t st, hipEvent_t done = nullptr, hipEvent_t start = nullptr);
void launch_nco_decimate(const float2 *in, int input_size, float cosdelta, float sindelta, float rate, int decimation,
		NcoState *state, float2 *phasor_scratch, float2 *out, hipStream_t st);
#ifdef HFDL_LAB
int fold_clock_probe_read(unsigned long long *out, int max, int *n);      // {columns x 100 + blocks, shader cycles, 100 MHz ticks, start tick} per fold launch
int stream_read_variants();
void launch_stream_read(int variant, const float2 *src, size_t bytes, float *sink, hipStream_t st);
#endif

}  // namespace hfdl
