// demod_kernels.hip -- K4 (per-channel streaming demodulator) and K5 (batched burst decoder) for gfx950.
//
// K4  demod_kernel        one workgroup of three wavefronts per channel, a software pipeline over chunks of samples;
//                         body in demod_core.h / demod_logic.h (reference src/hfdl.c:676-892)
// K5  burst_decode_kernel one wavefront per finished frame: descramble + soft de-map + 40-row de-interleave
//                         + rate-1/4 combine + K=7 Viterbi + octet bit reversal + PDU metadata
//                         (reference decode_user_data / dispatch_pdu, src/hfdl.c:993-1080;
//                          libfec update_viterbi27_blk / chainback_viterbi27, src/libfec/viterbi27_port.c:105-221).
//     Viterbi mapping: lane = trellis state (64 = one wavefront); the two predecessor metrics arrive by
//     cross-lane shuffle, the 64 decisions of a trellis step are one __ballot word kept in LDS.
#include <hip/hip_ext.h>
#include <cstdlib>
#include <utility>
#include <vector>
#include <cstring>
#include "demod_core.h"
#ifdef HFDL_DM_STRICT
// Hook of the TEST-ONLY strict builds (tests/hostsim/strict_demod_kernels.hip includes this file after defining HFDL_DM_STRICT, the
// fixed-sequence elementary functions and HFDL_DM_SERIAL_LOOP = the header with the one-lane serial loop it wants run instead of the
// three-wave pipeline).  The product build defines none of it and this file names nothing under tests/.
#include HFDL_DM_SERIAL_LOOP
#endif
#include "demod_tables.h"
#include "demod.h"

namespace hfdl {

static_assert(sizeof(cf) == sizeof(float2), "cf must alias float2");

struct DevTables {            // device image, pointers resolved on the host
	DemodConst c;
	const uint8_t *scrambler;
};

struct DemodBuffers {
	ChanState *states;
	cf *data;
	FrameRec *frames;
	int *counts;
	int *frame_count;           // this launch's frames-queued counter (four, rotating: see Demod::enqueue_demod)
	int frame_cap;
	cf *tap_rs, *tap_mf, *tap_sym;
	float *tap_lvl;
	int *tap_counts;
	int cap;
};

// global -> LDS copy by NT threads, 8 independent loads per thread issued before the first is used
template <int NT, typename T>
__device__ __forceinline__ void stage_copy(T *__restrict__ dst, const T *__restrict__ src, int n, int tid)
{
	int i = tid;
	for (; i + 7 * NT < n; i += 8 * NT) {
		T v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) v[u] = src[i + u * NT];
#pragma unroll
		for (int u = 0; u < 8; u++) dst[i + u * NT] = v[u];
	}
	for (; i < n; i += NT) dst[i] = src[i];
}

// LDS carve-up of a channel's workgroup, shared by the kernel and the host-side size computation
struct DemodLds {
	size_t arrays, scalars, sstab, mf, eq, m1, corr, mbox, sink, stage, rs, agc, mfo, lvl, outq, cum, rs_h, total;
	__host__ __device__ explicit DemodLds(int cap)
	{
		size_t o = 0;
		auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 15) & ~(size_t)15; return at; };
		arrays = take(sizeof(ChanArrays));
		scalars = take(sizeof(ChanScalars));
		sstab = take(sizeof(float2) * D_SS_NPFB * 64);
		mf = take(sizeof(float) * 32);
		eq = take(sizeof(float) * 16);
		m1 = take(sizeof(uint64_t) * 16);
		corr = take(sizeof(float) * 128);
		mbox = take(sizeof(int) * 8);
		sink = take(sizeof(float) * 64);               // where the lanes of an all-lane LDS write that have nothing to say put it
		stage = take(sizeof(cf) * 64);                 // data symbols of the carrier wave's current chunk, on their way to HBM
		rs = take(sizeof(cf) * (size_t)cap);
		agc = take(sizeof(cf) * (size_t)cap);          // agc and mfo are adjacent: together they stage the block's input
		mfo = take(sizeof(cf) * ((size_t)cap + SS_HIST)) + sizeof(cf) * SS_HIST;     // SS_HIST history entries sit right before mf[0]
		lvl = take(sizeof(float) * (size_t)cap);
		outq = take(sizeof(cf) * (size_t)OUTQ_RING);
		cum = take(sizeof(uint16_t) * (size_t)cap);
		rs_h = take(sizeof(float) * D_RS_NPFB * D_RS_TAPS);
		total = o;
	}
};

#ifdef HFDL_LAB
// Laboratory build: the shader clock a launch ran at, measured from inside it.  Workgroup 0 notes s_memtime (shader cycles) and
// s_memrealtime (the constant 100 MHz reference) when it starts and when it ends: cycles / reference ticks x 100 MHz = the average clock
// of THIS launch -- what rocm-smi's quarter-second samples cannot resolve (a fold launch lasts 4 - 7 ms, a demodulator launch 1 ms).
// Ring of 4096 records {launch tag, cycles, reference ticks, 0}; hfdl_gpu_lab_clock_probe_read().
__device__ unsigned long long hfdl_clk_probe[4096 * 4];
__device__ unsigned hfdl_clk_probe_n;
#define HFDL_CLK_PROBE_BEGIN(tag) unsigned long long clk_c0 = 0, clk_r0 = 0; unsigned clk_slot = 0; \
	if (blockIdx.x == 0 && threadIdx.x == 0) { clk_slot = atomicAdd(&hfdl_clk_probe_n, 1u) & 4095u; clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#define HFDL_CLK_PROBE_END(tag) if (blockIdx.x == 0 && threadIdx.x == 0) { hfdl_clk_probe[4 * clk_slot] = (tag); hfdl_clk_probe[4 * clk_slot + 1] = __builtin_amdgcn_s_memtime() - clk_c0; \
	hfdl_clk_probe[4 * clk_slot + 2] = __builtin_amdgcn_s_memrealtime() - clk_r0; hfdl_clk_probe[4 * clk_slot + 3] = clk_r0; }
#else
#define HFDL_CLK_PROBE_BEGIN(tag)
#define HFDL_CLK_PROBE_END(tag)
#endif

// __launch_bounds__(192, 5): at most 96 VGPRs per wave (the code object says 73 used, 80 allocated).  The demodulator workgroups (three
// waves each, on three SIMDs of a CU) are co-resident with the fold kernel's workgroups (stream A), and the two are budgeted against each
// other: the fold's tiling leaves a SIMD at least these registers of its 512 (one 372 / 420-register wave per SIMD since round 5; four waves
// of 104 in round 1) -- a fold tiling that does not (two waves of 256) makes the two kernels take turns and costs the pipeline 10 %
// (fold_kernels.hip, profiles/r05/k4_tilings_in_pipeline.txt).  Staying inside the budget also keeps the loops free of scratch spills:
// scratch is memory, and beside a kernel that reads HBM at 4.6 TB/s every spill reload is a multi-microsecond stall.
template <bool TAPS>
__global__ __launch_bounds__(DM_THREADS, 5) void demod_kernel(DevTables T, DemodBuffers B, const cf *__restrict__ chan_out,
		const int *__restrict__ n_in, int outs_stride, int nblk, int nch)
{
	// These workgroups are a handful of wavefronts that run serial recurrences beside thousands of throughput-bound ones (fold, forward
	// FFT): whenever one of them has an instruction ready it goes first (the instruction arbiter otherwise treats all waves of a SIMD alike)
	__builtin_amdgcn_s_setprio(3);
	HFDL_CLK_PROBE_BEGIN(1)
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	const int c = blockIdx.x, tid = threadIdx.x;
	const DemodLds L(B.cap);
	ChanArrays *A = (ChanArrays *)(lds + L.arrays);
	ChanScalars *S = (ChanScalars *)(lds + L.scalars);
	float2 *l_sstab = (float2 *)(lds + L.sstab);
	float *l_mf = (float *)(lds + L.mf), *l_eq = (float *)(lds + L.eq), *l_corr = (float *)(lds + L.corr);
	uint64_t *l_m1 = (uint64_t *)(lds + L.m1);
	float *l_rs_h = (float *)(lds + L.rs_h);
	// The channelizer output of this block is staged in LDS (in the space of agc + mf, which are written only after the
	// resampler has consumed it: 16 cap >= 8 n_in because the resampling rate is > 0.5), and so is the resampler's filter
	// bank: all loads of a lane are in flight together, instead of one dependent memory round trip per filter tap.
	cf *l_in = (cf *)(lds + L.agc);

	ChanState *gs = B.states + c;
	// prologue copies with 8 loads of a lane in flight at a time: beside the fold kernel a dependent load costs microseconds
	stage_copy<DM_THREADS>((uint32_t *)A, (const uint32_t *)&gs->a, (int)(sizeof(ChanArrays) / 4), tid);
	stage_copy<DM_THREADS>((uint32_t *)S, (const uint32_t *)&gs->s, (int)(sizeof(ChanScalars) / 4), tid);
	// the launch's blocks ([nblk][nch][outs]; one block unless the host batches) end to end: one stretch of channelizer output
	int n_block = 0;
	for (int b = 0; b < nblk; b++) {
		const int nb = n_in[b * nch + c];
		stage_copy<DM_THREADS>(l_in + n_block, chan_out + ((size_t)b * nch + c) * outs_stride, nb, tid);
		n_block += nb;
	}
	stage_copy<DM_THREADS>((float4 *)l_rs_h, (const float4 *)T.c.rs_h, D_RS_NPFB * D_RS_TAPS / 4, tid);
	// symsync taps in the lane order of the timing-recovery wave: entry [bank][lane] = {tap t, tap t + 16} of the lane's row
	for (int e = tid; e < D_SS_NPFB * 64; e += DM_THREADS) {
		const int bank = e >> 6, row = (e >> 4) & 3, t = e & 15;
		const float *src = (row < 2 ? T.c.ss_mf : T.c.ss_dmf) + bank * D_SS_TAPS;
		l_sstab[e] = make_float2(src[t], t + 16 < D_SS_TAPS ? src[t + 16] : 0.f);
	}
	if (tid < D_MF) l_mf[tid] = T.c.mf[tid];
	if (tid < D_EQ) l_eq[tid] = T.c.eq_h0[tid];
	if (tid < 8) { l_m1[tid] = T.c.m1_hi[tid]; l_m1[8 + tid] = T.c.m1_lo[tid]; }
	if (tid < 128) l_corr[tid] = T.c.corr_tab[tid];
	if (tid < 8) ((int *)(lds + L.mbox))[tid] = 0;
	__syncthreads();

	DemodConst K = T.c;
	K.rs_h = l_rs_h; K.ss_mf = nullptr; K.ss_dmf = nullptr; K.mf = l_mf; K.eq_h0 = l_eq; K.m1_hi = l_m1; K.m1_lo = l_m1 + 8; K.corr_tab = l_corr;
	BlockIo io;
	io.rs = (cf *)(lds + L.rs); io.agc = (cf *)(lds + L.agc); io.mf = (cf *)(lds + L.mfo); io.lvl = (float *)(lds + L.lvl); io.cap = B.cap;
	io.data = B.data + (size_t)c * 2 * MAX_DATA_SYMBOLS;
	io.frames = B.frames; io.frame_count = B.frame_count; io.frame_cap = B.frame_cap;
	io.channel = c;
	if (TAPS) {
		io.tap_resampled = B.tap_rs + (size_t)c * B.cap; io.tap_mf = B.tap_mf + (size_t)c * B.cap;
		io.tap_symbols = B.tap_sym + (size_t)c * B.cap; io.tap_level = B.tap_lvl + (size_t)c * B.cap;
		io.tap_counts = B.tap_counts + 2 * c;
	} else {
		io.tap_resampled = nullptr; io.tap_mf = nullptr; io.tap_symbols = nullptr; io.tap_level = nullptr; io.tap_counts = nullptr;
	}
	DemodShared sh;
	sh.outq = (cf *)(lds + L.outq); sh.outq_cap = 2 * B.cap;
	sh.cum = (uint16_t *)(lds + L.cum);
	sh.sstab = l_sstab;
	sh.S = S;
	sh.mbox = (int *)(lds + L.mbox);
	sh.sink = (float *)(lds + L.sink);
	sh.stage = (cf *)(lds + L.stage);
#ifdef HFDL_DM_STRICT
	if (tid == 0) {
		DemodConst Ks = K;
		Ks.ss_mf = T.c.ss_mf; Ks.ss_dmf = T.c.ss_dmf;      // the serial loop reads the filter banks where they lie
		demod_block_serial(*S, *A, Ks, io, l_in, n_block);
	}
#else
	demod_block<TAPS>(*A, K, io, sh, l_in, n_block);
#endif
	__syncthreads();
	{
		uint32_t *dst = (uint32_t *)&gs->a;
		const uint32_t *src = (const uint32_t *)A;
		for (unsigned i = tid; i < sizeof(ChanArrays) / 4; i += DM_THREADS) dst[i] = src[i];
		dst = (uint32_t *)&gs->s;
		src = (const uint32_t *)S;
		for (unsigned i = tid; i < sizeof(ChanScalars) / 4; i += DM_THREADS) dst[i] = src[i];
	}
	HFDL_CLK_PROBE_END(1000ull + (unsigned long long)nblk)           // tag: 1000 + blocks of the launch = a demodulator launch
}

// ---------------------------------------------------------------- K5

__device__ inline int parity_u32(uint32_t x) { return __popc(x) & 1; }

// ---- K=7 r=1/2 Viterbi, one wave per frame, arithmetic of update_viterbi27_blk / chainback_viterbi27
// (src/libfec/viterbi27_port.c:105-135, 147-221).
//
// Both loops are serial chains, so their length in instructions is the decoder's latency (measured with s_memtime: the
// first version spent 136 cycles per step going forward and 290 per step coming back).
//  * State metrics stay IN PLACE with rotating labels: at step t the metric of state s lives in lane rotr6(s, t mod 6).  The
//    predecessors i and i+32 of the states 2i and 2i+1 then sit in the two lanes that differ in bit 5-(t mod 6), and the
//    same two lanes hold 2i and 2i+1 afterwards: ONE exchange with lane ^ (32 >> t mod 6) per step -- a DPP move on the VALU
//    for four of the six residues -- instead of two arbitrary lane gathers.
//  * In a lane the "own" predecessor costs bm and the "other" 510 - bm whatever the parity of the new state, and the libfec
//    tie rule (decision = path via i+32 strictly cheaper) becomes take_other = (own - other + odd) > 0; decision = take ^ odd.
//  * The 64 decisions of a step are the compare's own mask; v_writelane parks it in lane (t mod 60) of a register pair and
//    60 words go to LDS lane-parallel: no exec-masked store in the chain.
//  * The chainback is scalar code: 64 words per trip are fetched lane-parallel and pulled into SGPRs by v_readlane with a
//    loop-constant lane, the state register never leaves the SALU, finished octets are parked with v_writelane and stored
//    64 at a time.
template <int X> __device__ __forceinline__ uint32_t lane_xor(uint32_t v)
{
	if (X == 32) return (uint32_t)__shfl_xor((int)v, 32);
	if (X == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                         // bit mode: and 0x1f, xor 0x10
	if (X == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);      // row_ror:8
	if (X == 4) {
		const int h = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);                 // row_half_mirror: ^7
		return (uint32_t)__builtin_amdgcn_update_dpp(0, h, 0x1B, 0xf, 0xf, false);                     // quad_perm [3,2,1,0]: ^3
	}
	if (X == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);                   // quad_perm [1,0,3,2]
}

// v_writelane_b32 (no clang builtin in this toolchain): lane SLOT of `old` takes the wave-uniform `value`.  On gfx9 a VALU
// instruction reads one SGPR over the constant bus, so the lane select has to be an inline constant.
template <int SLOT> __device__ __forceinline__ uint32_t write_lane(uint32_t old, uint32_t value)
{
	asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(value), "n"(SLOT));
	return old;
}

struct VitLane {                                  // per-lane constants of the six residues
	uint32_t t0[6], t1[6];                        // branch-table bytes (0 / 255) of the butterfly this lane belongs to
	int odd[6];                                   // 1 if the lane's new state is odd (its own predecessor is i+32)
};

// lanes whose bit 5-R is set: the lanes that hold an odd new state after a residue-R step
template <int R> __device__ __forceinline__ constexpr uint64_t odd_lanes()
{
	return R == 0 ? 0xFFFFFFFF00000000ull : R == 1 ? 0xFFFF0000FFFF0000ull : R == 2 ? 0xFF00FF00FF00FF00ull
		: R == 3 ? 0xF0F0F0F0F0F0F0F0ull : R == 4 ? 0xCCCCCCCCCCCCCCCCull : 0xAAAAAAAAAAAAAAAAull;
}

// one trellis step of residue R; returns the step's 64 decisions (bit = lane)
template <int R> __device__ __forceinline__ uint64_t acs_step(uint32_t &metric, uint32_t sj, const VitLane &c)
{
	const uint32_t s0 = sj & 255u, s1 = sj >> 8;
	const uint32_t bm = (c.t0[R] ^ s0) + (c.t1[R] ^ s1);
	const uint32_t other = lane_xor<(32 >> R)>(metric);
	const uint32_t cown = metric + bm, coth = other + (510u - bm);
	const bool take_other = (int32_t)(cown - coth) + c.odd[R] > 0;
	metric = take_other ? coth : cown;
	return __ballot(take_other) ^ odd_lanes<R>();
}

// step J of a 60-step chunk: decisions parked in lane J of (wlo, whi)
template <int J> __device__ __forceinline__ void chunk_step(uint32_t &metric, uint32_t &wlo, uint32_t &whi, uint32_t pair, const VitLane &c)
{
	const uint64_t word = acs_step<J % 6>(metric, (uint32_t)__builtin_amdgcn_readlane((int)pair, J), c);
	wlo = write_lane<J>(wlo, (uint32_t)word);
	whi = write_lane<J>(whi, (uint32_t)(word >> 32));
}

template <int... J> __device__ __forceinline__ void chunk_steps(uint32_t &metric, uint32_t &wlo, uint32_t &whi, uint32_t pair, const VitLane &c,
		std::integer_sequence<int, J...>)
{
	(chunk_step<J>(metric, wlo, whi, pair, c), ...);
}

// chainback step J of a 48-step trip (wave-uniform values throughout): word of step = lane J of (wlo, whi)
template <int J> __device__ __forceinline__ void chain_step(uint32_t wlo, uint32_t whi, uint32_t &pos, uint32_t &hi, uint32_t &lo)
{
	const uint64_t wj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wlo, J) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)whi, J) << 32);
	const uint32_t k = (uint32_t)(wj >> pos) & 1u;
	constexpr uint32_t B = 1u << (J % 6);
	pos = k ? (pos | B) : (pos & ~B);
	if (J < 16) hi = (hi << 1) | k; else lo = (lo << 1) | k;
}

template <int... J> __device__ __forceinline__ void chain_steps(uint32_t wlo, uint32_t whi, uint32_t &pos, uint32_t &hi, uint32_t &lo,
		std::integer_sequence<int, J...>)
{
	(chain_step<J>(wlo, whi, pos, hi, lo), ...);
}

__device__ __forceinline__ uint32_t rotl6(uint32_t x, int k) { return k ? ((x << k) | (x >> (6 - k))) & 63u : x; }
__device__ __forceinline__ uint32_t rotr6(uint32_t x, int k) { return k ? ((x >> k) | (x << (6 - k))) & 63u : x; }

__host__ __device__ constexpr size_t viterbi_lds_bytes(int nbits) { return sizeof(uint64_t) * ((size_t)nbits + 8); }

// vin = 2*nbits soft bytes (LDS or global); decw = viterbi_lds_bytes(nbits) of LDS.  out receives ceil(nbits/8) octets:
// MSB-first as libfec leaves them, or bit-reversed (what dumphfdl dispatches).
__device__ __forceinline__ void viterbi27_wave(const uint8_t *vin, int nbits, uint64_t *decw, uint8_t *out, bool reverse_bits)
{
	const int lane = threadIdx.x;
	VitLane c;
#pragma unroll
	for (int r = 0; r < 6; r++) {
		const uint32_t n = rotl6((uint32_t)lane, (r + 1) % 6), i = n >> 1;      // new state held by this lane after a residue-r step
		c.t0[r] = parity_u32((2u * i) & 0x6d) ? 255u : 0u;
		c.t1[r] = parity_u32((2u * i) & 0x4f) ? 255u : 0u;
		c.odd[r] = (lane >> (5 - r)) & 1;
	}
	uint32_t metric = lane == 0 ? 0u : 63u;           // init_viterbi27(vp, 0)
#ifdef HFDL_VIT_DEBUG
	const unsigned long long dbg0 = __builtin_amdgcn_s_memtime();
#endif
	// 60 trellis steps per trip: every lane fetches the soft pair of one step, the serial loop takes them with v_readlane
	for (int base = 0; base < nbits; base += 60) {
		const int tt = base + lane;
		const uint32_t pair = (lane < 60 && tt < nbits) ? ((uint32_t)vin[2 * tt] | ((uint32_t)vin[2 * tt + 1] << 8)) : 0u;
		const int lim = nbits - base < 60 ? nbits - base : 60;
		uint32_t wlo = 0, whi = 0;
		if (lim == 60) {                                 // every HFDL frame size is a multiple of 60
			chunk_steps(metric, wlo, whi, pair, c, std::make_integer_sequence<int, 60>());
		} else {                                         // ragged end of an arbitrary nbits: same steps, generic slot select
#define HFDL_ACS(R) if (j + R < lim) { const uint64_t wd = acs_step<R>(metric, (uint32_t)__builtin_amdgcn_readlane((int)pair, j + R), c); \
			if (lane == j + R) { wlo = (uint32_t)wd; whi = (uint32_t)(wd >> 32); } }
			for (int j = 0; j < lim; j += 6) { HFDL_ACS(0) HFDL_ACS(1) HFDL_ACS(2) HFDL_ACS(3) HFDL_ACS(4) HFDL_ACS(5) }
#undef HFDL_ACS
		}
		if (lane < lim) decw[base + lane] = ((uint64_t)whi << 32) | wlo;
	}
#ifdef HFDL_VIT_DEBUG
	const unsigned long long dbg1 = __builtin_amdgcn_s_memtime();
#endif
	__syncthreads();
	// chainback with the "d += 6" offset: bit idx comes from the word of step idx+6 (words past the end read as 0).  The
	// decision of state s at step t sits in bit rotr6(s, (t+1) mod 6); that rotated state register `pos` changes in ONE bit per
	// step (bit (6 - (t+1) mod 6) mod 6 takes the decoded bit), so it is never shifted or rotated.
	const int noct = (nbits + 7) >> 3;
	uint32_t pos = 0;
	int idx = nbits - 1;
	{	// head: single steps until idx = 47 (mod 48); covers a ragged top octet and any nbits
		uint32_t reg = 0;
		for (; idx >= 0 && idx % 48 != 47; idx--) {
			const int tt = idx + 6;
			uint64_t w = 0;
			if (tt < nbits) w = decw[tt];
			const uint32_t wl = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w), wh = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32));
			const uint32_t k = (uint32_t)((((uint64_t)wh << 32) | wl) >> pos) & 1u;
			const int b = (6 - (tt + 1) % 6) % 6;
			pos = (pos & ~(1u << b)) | (k << b);
			reg = (reg >> 1) | (k << 7);
			if ((idx & 7) == 0 && lane == 0) out[idx >> 3] = reverse_bits ? (uint8_t)(__brev(reg) >> 24) : (uint8_t)reg;
		}
	}
	// body: 48 steps per trip (a multiple of 6 and of 8: bit index and octet boundaries are compile-time), scalar code
	for (; idx >= 47; idx -= 48) {
		const int mine = idx - lane + 6;
		const uint64_t w = (lane < 48 && mine < nbits) ? decw[mine] : 0ull;
		uint32_t hi = 0, lo = 0;
		chain_steps((uint32_t)w, (uint32_t)(w >> 32), pos, hi, lo, std::make_integer_sequence<int, 48>());
		// bit n of (hi:lo) = decoded bit idx-47+n: octet L of the trip is byte L, already in dumphfdl's (bit-reversed) order
		if (lane < 6) {
			uint32_t v = (lane < 4 ? lo >> (8 * lane) : hi >> (8 * (lane - 4))) & 255u;
			if (!reverse_bits) v = __brev(v) >> 24;
			const int o = ((idx - 47) >> 3) + lane;
			if (o < noct) out[o] = (uint8_t)v;
		}
	}
#ifdef HFDL_VIT_DEBUG
	if (lane == 0 && blockIdx.x == 0) printf("viterbi nbits %d forward %llu chainback %llu cycles\n", nbits, dbg1 - dbg0, (unsigned long long)__builtin_amdgcn_s_memtime() - dbg1);
#endif
}

constexpr int K5_TABLE_BYTES = 15360;      // >= 168*30*3 coded bits, 256-aligned

__global__ __launch_bounds__(64) void burst_decode_kernel(const FrameRec *__restrict__ frames, int *counts, const int *nframes_ptr, int *stale_count, int frame_cap,
		const cf *__restrict__ data_all, const uint8_t *__restrict__ scrambler, const int32_t *__restrict__ freqs,
		hfdl_gpu_pdu *__restrict__ pdus, int pdu_cap)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	__builtin_amdgcn_s_setprio(2);          // one wavefront per frame beside the fold: first in line when it has something to issue (see demod_kernel)
	const int f = blockIdx.x, lane = threadIdx.x;
	int nframes = *nframes_ptr;
	if (f == 0 && lane == 0 && stale_count) *stale_count = 0;      // the counter of the launch after next: its last users finished before this launch began
	if (nframes > frame_cap) nframes = frame_cap;
	if (f >= nframes) return;
	uint8_t *table = lds;
	uint8_t *vin = lds + K5_TABLE_BYTES;
	uint64_t *decw = (uint64_t *)(lds + 2 * K5_TABLE_BYTES);
	uint8_t *l_scr = lds + 2 * K5_TABLE_BYTES + viterbi_lds_bytes(7560);             // 128 bytes, then the constellation table
	float *l_psk = (float *)(l_scr + DEC_PSK_OFFSET);

	const FrameRec fr = frames[f];
	const ModeParams mp = mode_params(fr.mode);
	const int nsym = mp.segments * 30, ncoded = nsym * mp.arity, cols = ncoded / 40;
	const cf *sym = data_all + ((size_t)fr.channel * 2 + fr.slot) * MAX_DATA_SYMBOLS;
	const float mask_flip = fr.bitmask_lsb ? -1.0f : 1.0f;
	l_scr[lane] = lane < 120 ? scrambler[lane] : 0;
	if (lane + 64 < 120) l_scr[lane + 64] = scrambler[lane + 64];
	if (lane < 32) l_psk[lane] = ((const float *)(scrambler + DEC_PSK_OFFSET))[lane];
	__syncthreads();
	// descramble + soft de-map + de-interleaver push (src/hfdl.c:1008-1019, 378-392); four symbol loads per lane in flight
	// (this kernel runs beside the fold, where a dependent global load costs microseconds)
	for (int base = 0; base < nsym; base += 256) {
		cf xs[4];
#pragma unroll
		for (int u = 0; u < 4; u++) { const int i = base + u * 64 + lane; xs[u] = i < nsym ? sym[i] : cf{0.f, 0.f}; }
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int i = base + u * 64 + lane;
			if (i >= nsym) continue;
			const float flip = (l_scr[i % 120] ? -1.0f : 1.0f) * mask_flip;
			cf x = xs[u];
			x.x *= flip; x.y *= flip;
			uint8_t soft[3];
			psk_soft(mp.arity, x, soft, PskTable{l_psk});
			for (int j = 0; j < mp.arity; j++) {
				const int k = i * mp.arity + j;
				const int row = k % 40;
				int col = (k / 40 - mp.col_shift * k) % cols;
				if (col < 0) col += cols;
				table[row * cols + col] = soft[j];
			}
		}
	}
	__syncthreads();
	// de-interleaver pop (+ rate-1/4 chip averaging), src/hfdl.c:394-403, 1022-1038
	const int vin_len = mp.code_rate == 4 ? ncoded / 2 : ncoded;
	for (int j = lane; j < vin_len; j += 64) {
		if (mp.code_rate == 4) {
			const int p0 = 2 * j, p1 = 2 * j + 1;
			const uint8_t a = table[((9 * p0) % 40) * cols + p0 / 40], b = table[((9 * p1) % 40) * cols + p1 / 40];
			vin[j] = (uint8_t)((a & b) + ((a ^ b) >> 1));
		} else {
			vin[j] = table[((9 * j) % 40) * cols + j / 40];
		}
	}
	// claim a slot of the PDU ring: counts[1] = PDUs ever produced, counts[3] = PDUs the host has taken (both mod 2^32);
	// a full ring drops the PDU (counts[2]) without leaving a hole
	int slot = -1;
	if (lane == 0) {
		unsigned *produced = (unsigned *)&counts[1];
		const unsigned taken = *(volatile unsigned *)&counts[3];
		unsigned t = *(volatile unsigned *)produced;
		for (;;) {
			if (t - taken >= (unsigned)pdu_cap) { atomicAdd(&counts[2], 1); break; }
			const unsigned seen = atomicCAS(produced, t, t + 1u);
			if (seen == t) { slot = (int)(t % (unsigned)pdu_cap); break; }
			t = seen;
		}
	}
	slot = __shfl(slot, 0);
	__syncthreads();
	if (slot < 0) return;
	const int nbits = vin_len / 2, noct = (nbits + 7) / 8;
	hfdl_gpu_pdu *out = pdus + slot;
	uint8_t *l_oct = table;                    // the de-interleaver table is dead: decoded octets go to LDS first
	viterbi27_wave(vin, nbits, decw, l_oct, true);
	__syncthreads();
	for (int i = lane; i < noct; i += 64) out->octets[i] = l_oct[i];
	if (lane == 0) {
		// dispatch_pdu metadata, src/hfdl.c:1058-1074
		out->channel = fr.channel;
		out->freq = freqs[fr.channel];
		out->mode = fr.mode;
		out->bit_rate = 1800 * mp.arity / mp.code_rate * DATA_FRAME_LEN / (DATA_FRAME_LEN + T_LEN);
		out->len = noct;
		out->freq_err_hz = fr.freq_err_hz;
		out->rssi_db = 20.0f * log10f(fr.signal_level);
		out->noise_floor_db = 20.0f * log10f(fr.noise_floor);
		out->slot = mp.segments == 72 ? 'S' : 'D';
		out->sample_index = fr.sample_index;
		out->train_bits_bad = fr.train_bad;
		out->train_bits_total = fr.train_total;
		int kind = 0;
		uint32_t hdr_len = 0;
		out->fcs_status = (uint8_t)pdu_triage(l_oct, (uint32_t)noct, &kind, &hdr_len);
		out->pdu_kind = (uint8_t)kind;
		out->hdr_len = (uint16_t)hdr_len;
		LpduCounts lc;
		lc.processed = lc.good = lc.bad_fcs = lc.too_short = lc.truncated = 0;
		if (out->fcs_status == 0 && kind != 0) lc = lpdu_walk(l_oct, (uint32_t)noct, kind, hdr_len);
		out->lpdus_processed = lc.processed; out->lpdus_good = lc.good; out->lpdus_bad_fcs = lc.bad_fcs;
		out->lpdus_too_short = lc.too_short; out->lpdus_truncated = lc.truncated;
		out->lpdu_pad[0] = out->lpdu_pad[1] = out->lpdu_pad[2] = 0;
	}
}

__global__ __launch_bounds__(64) void viterbi_batch_kernel(const uint8_t *__restrict__ soft, int nbits, uint8_t *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	const int f = blockIdx.x;
	viterbi27_wave(soft + (size_t)f * 2 * nbits, nbits, (uint64_t *)lds, out + (size_t)f * ((nbits + 7) / 8), false);
}

// stage kernels behind hfdl_gpu_crc16_ccitt / hfdl_gpu_pdu_triage: the device functions burst_decode_kernel runs on every PDU
__global__ void crc16_kernel(const uint8_t *__restrict__ data, uint32_t len, uint32_t crc_init, uint32_t *__restrict__ out)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) *out = crc16_ccitt_step(data, len, (uint16_t)crc_init);
}

__global__ void pdu_triage_kernel(const uint8_t *__restrict__ octets, const int32_t *__restrict__ lens, int npdus, int stride,
		uint8_t *__restrict__ fcs_status, uint8_t *__restrict__ kind, uint16_t *__restrict__ hdr_len)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= npdus) return;
	int k = 0;
	uint32_t hl = 0;
	fcs_status[i] = (uint8_t)pdu_triage(octets + (size_t)i * stride, (uint32_t)lens[i], &k, &hl);
	kind[i] = (uint8_t)k;
	hdr_len[i] = (uint16_t)hl;
}

// the carrier wave's slicer (demod_core.h LaneSlicer) on its own: every wave walks its share of the symbols one at a time, the
// way the carrier loop meets them (the symbol is wave-uniform, the constellation sits one point per lane)
__global__ __launch_bounds__(64) void psk_slice_kernel(int arity, const cf *__restrict__ x, int n, const float *__restrict__ pts, uint32_t *__restrict__ sym, float *__restrict__ perr)
{
	const int lane = (int)threadIdx.x;
	const LaneSlicer slice{lane < 16 ? pts[2 * lane] : 0.f, lane < 16 ? pts[2 * lane + 1] : 0.f, lane};
	const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x, i0 = (int)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
	for (int base = i0; base < i1; base += 64) {
		const cf mine = base + lane < i1 ? x[base + lane] : cf{0.f, 0.f};
		uint32_t s_l = 0;
		float e_l = 0.f;
		for (int k = 0; k < 64 && base + k < i1; k++) {
			cf v; v.x = lane_value(mine.x, k); v.y = lane_value(mine.y, k);
			float e;
			const uint32_t sy = slice(arity, v, &e);
			if (lane == k) { s_l = sy; e_l = e; }
		}
		if (base + lane < i1) { sym[base + lane] = s_l; perr[base + lane] = e_l; }
	}
}

__global__ void lpdu_walk_kernel(const uint8_t *__restrict__ octets, const int32_t *__restrict__ lens, int npdus, int stride, uint8_t *__restrict__ counts)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= npdus) return;
	const uint8_t *buf = octets + (size_t)i * stride;
	int k = 0;
	uint32_t hl = 0;
	LpduCounts c;
	c.processed = c.good = c.bad_fcs = c.too_short = c.truncated = 0;
	if (pdu_triage(buf, (uint32_t)lens[i], &k, &hl) == 0 && k != 0) c = lpdu_walk(buf, (uint32_t)lens[i], k, hl);
	uint8_t *o = counts + (size_t)i * 5;
	o[0] = c.processed; o[1] = c.good; o[2] = c.bad_fcs; o[3] = c.too_short; o[4] = c.truncated;
}

// read-back of the named constants and the tables computed from them AS THE DEVICE SEES THEM (laboratory entry point
// hfdl_gpu_lab_read_constants; compared with the reference's text in tests/test_gpu_constants.py)
__global__ void constants_kernel(HfdlConstants *out)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		HfdlConstants k;
		hfdl_constants(k);
		*out = k;
	}
}

// ---------------------------------------------------------------- host side

#define D_TRY(expr) do { if ((expr) != hipSuccess) return HFDL_GPU_EHIP; } while (0)

// HIP events on the null stream around a stage entry point's launch (kernel time without the copies)
struct KernelTimer {
	hipEvent_t e0 = nullptr, e1 = nullptr;
	explicit KernelTimer(bool on) { if (on && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, nullptr); }
	double stop() { float ms = 0; if (e0 && e1) { (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); } return ms; }
	~KernelTimer() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

static size_t demod_lds_bytes(int cap) { return DemodLds(cap).total; }

static size_t k5_lds_bytes() { return 2 * (size_t)K5_TABLE_BYTES + viterbi_lds_bytes(7560) + DEC_CONST_BYTES; }

static DevTables resolve_tables(const float *d_img, const DemodTables &h)
{
	const unsigned char *base = (const unsigned char *)d_img;
	auto at = [&](const void *field) { return base + ((const unsigned char *)field - (const unsigned char *)&h); };
	DevTables t;
	t.c.rs_h = (const float *)at(h.rs_h);
	t.c.rs_step = h.rs_step;
	t.c.mf = (const float *)at(h.mf);
	t.c.ss_mf = (const float *)at(h.ss_mf);
	t.c.ss_dmf = (const float *)at(h.ss_dmf);
	t.c.lf_b0 = h.lf_b0; t.c.lf_a1 = h.lf_a1; t.c.ss_rate_adj = h.ss_rate_adj;
	t.c.eq_h0 = (const float *)at(h.eq_h0);
	t.c.a_hi = h.a_hi; t.c.a_lo = h.a_lo;
	t.c.m1_hi = (const uint64_t *)at(h.m1_hi);
	t.c.m1_lo = (const uint64_t *)at(h.m1_lo);
	t.scrambler = (const uint8_t *)at(h.scrambler);
	t.c.corr_tab = (const float *)at(h.corr_tab);
	t.c.psk_pts = (const float *)at(h.psk_pts);
	t.c.a1_lo = h.a1_lo; t.c.a1_hi = h.a1_hi; t.c.a2_lo = h.a2_lo; t.c.a2_hi = h.a2_hi; t.c.pos_min = h.pos_min;
	return t;
}

struct DemodPriv { DemodTables h; DevTables t; };
static DemodPriv *priv_of(const Demod *d) { return (DemodPriv *)d->priv; }

static int set_big_lds(const void *fn, size_t bytes)
{
	if (bytes > 64 * 1024) D_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
	return 0;
}

// blocks a launch can take at most, `want` or fewer: what fits the LDS ...
int Demod::fit_batch(int outs, float resamp_rate, int want)
{
	int batch = want < 1 ? 1 : want;
	for (;; batch--) {
		const int cap = (int)((double)outs * (double)batch * (double)resamp_rate + 8);
		// ... and less than one second of signal per launch WHATEVER asked for the batch (cap samples at 5400 sps): a channel then finishes
		// at most one frame per launch -- the frame queue has one entry per channel and the frame buffers two slots (hfdl_gpu.cpp
		// pick_demod_batch states the same bound; an override or a larger LDS must not get past it)
		if (batch == 1 || (demod_lds_bytes(cap) <= 160 * 1024 && 2 * cap <= 65535 && (double)cap / 5400.0 < 1.0)) break;      // cum[] counts outputs in 16 bits
	}
	return batch;
}

int Demod::init(int nch_, int outs_, float resamp_rate, const int32_t *freqs, hipStream_t st, int batch_want)
{
	nch = nch_; outs = outs_;
	if (resamp_rate <= 0.5f || resamp_rate > 1.0f) return HFDL_GPU_ERANGE;   // one arbitrary stage, no half-band stages
	// blocks per launch: the per-launch sample buffers live in LDS (30 bytes per 5400-sps sample next to ~29 KiB of tables, state and the output ring)
	batch = fit_batch(outs, resamp_rate, batch_want);
	cap = (int)((double)outs * (double)batch * (double)resamp_rate + 8);
	auto *pv = new DemodPriv();
	build_demod_tables(pv->h, resamp_rate);
	priv = pv;
	D_TRY(hipMalloc(&d_tables, sizeof(DemodTables)));
	D_TRY(hipMemcpyAsync(d_tables, &pv->h, sizeof(DemodTables), hipMemcpyHostToDevice, st));
	pv->t = resolve_tables(d_tables, pv->h);

	std::vector<ChanState> init((size_t)nch);
	for (auto &s : init) chan_state_init(s, pv->h.eq_h0);
	D_TRY(hipMalloc(&d_states, sizeof(ChanState) * (size_t)nch));
	D_TRY(hipMemcpy(d_states, init.data(), sizeof(ChanState) * (size_t)nch, hipMemcpyHostToDevice));
	D_TRY(hipMalloc(&d_data, sizeof(float2) * (size_t)nch * 2 * MAX_DATA_SYMBOLS));
	D_TRY(hipMemsetAsync(d_data, 0, sizeof(float2) * (size_t)nch * 2 * MAX_DATA_SYMBOLS, st));
	D_TRY(hipMalloc(&d_frames, sizeof(FrameRec) * 2 * (size_t)nch));
	D_TRY(hipMalloc(&d_counts, sizeof(int) * 8));
	D_TRY(hipMemsetAsync(d_counts, 0, sizeof(int) * 8, st));
	D_TRY(hipHostMalloc((void **)&h_snap, sizeof(int) * 8, hipHostMallocDefault));
	std::memset(h_snap, 0, sizeof(int) * 8);
	taken = 0; dropped = 0;
	pdu_cap = std::max(4096, 64 * nch);       // ~1 KiB each; polled by the host at least once per few seconds of signal
	if (const char *e = getenv("HFDL_GPU_PDU_RING")) {       // test / tuning knob (include/hfdl_gpu.h)
		const long v = strtol(e, nullptr, 10);
		if (v >= 1 && v <= (1 << 20)) pdu_cap = (int)v;
	}
	D_TRY(hipMalloc(&d_pdus, sizeof(hfdl_gpu_pdu) * (size_t)pdu_cap));
	D_TRY(hipStreamCreateWithFlags(&st_collect, hipStreamNonBlocking));
	bounce_cap = pdu_cap < 512 ? pdu_cap : 512;
	D_TRY(hipHostMalloc((void **)&h_pdu_bounce, sizeof(hfdl_gpu_pdu) * (size_t)bounce_cap, hipHostMallocDefault));
	D_TRY(hipHostMalloc(&h_stats_bounce, sizeof(ChanScalars) * (size_t)nch, hipHostMallocDefault));
	D_TRY(hipMalloc(&d_freqs, sizeof(int32_t) * (size_t)nch));
	D_TRY(hipMemcpyAsync(d_freqs, freqs, sizeof(int32_t) * (size_t)nch, hipMemcpyHostToDevice, st));
	if (taps_on) {
		D_TRY(hipMalloc(&d_tap_rs, sizeof(float2) * (size_t)nch * cap));
		D_TRY(hipMalloc(&d_tap_mf, sizeof(float2) * (size_t)nch * cap));
		D_TRY(hipMalloc(&d_tap_sym, sizeof(float2) * (size_t)nch * cap));
		D_TRY(hipMalloc(&d_tap_lvl, sizeof(float) * (size_t)nch * cap));
		D_TRY(hipMalloc(&d_tap_counts, sizeof(int) * 2 * (size_t)nch));
		D_TRY(hipMemsetAsync(d_tap_counts, 0, sizeof(int) * 2 * (size_t)nch, st));
	}
	lds_bytes = demod_lds_bytes(cap);
	if (lds_bytes > 160 * 1024) return HFDL_GPU_ERANGE;
	int rc;
	if ((rc = set_big_lds((const void *)demod_kernel<true>, lds_bytes))) return rc;
	if ((rc = set_big_lds((const void *)demod_kernel<false>, lds_bytes))) return rc;
	if ((rc = set_big_lds((const void *)burst_decode_kernel, k5_lds_bytes()))) return rc;
	return 0;
}

// Frames finished by the demodulator of launch i are queued in d_frames[i & 1] and counted in d_counts[4 + (i & 3)].  The
// burst decoder of launch i may run on another stream than the demodulator of launch i+1; it zeroes the counter of launch i+2,
// whose previous users (launch i-2) are done and whose next user (the demodulator of launch i+2) is made to wait for this
// decoder by the caller -- no memset launch per block, no counter shared by two kernels that can overlap.
int Demod::enqueue_demod(const float2 *chan_out, const int *out_count, int nblk, hipStream_t st, hipEvent_t done, bool frames_free, hipEvent_t start)
{
	DemodPriv *pv = priv_of(this);
	if (!pv || nblk < 1 || nblk > batch) return HFDL_GPU_EINVAL;
	DemodBuffers B;
	const uint64_t i = launches++;          // per demodulator launch (not per block: channelize-only blocks launch none)
	// the decoder of launch i-2 has read this frame queue.  Every wait or record is a barrier packet of its own in the queue
	// (~5 us of idle stream each): on the demodulator-bound geometries the caller moves this one to the channelizer's stream
	if (separate_decode && !frames_free && ev_dec[i & 1]) D_TRY(hipStreamWaitEvent(st, ev_dec[i & 1], 0));
	B.states = d_states; B.data = (cf *)d_data; B.frames = d_frames + (size_t)(i & 1) * nch; B.counts = d_counts;
	B.frame_count = d_counts + 4 + (int)(i & 3); B.frame_cap = nch;
	const bool tw = taps_on && taps_enabled;
	B.tap_rs = tw ? (cf *)d_tap_rs : nullptr; B.tap_mf = (cf *)d_tap_mf; B.tap_sym = (cf *)d_tap_sym; B.tap_lvl = d_tap_lvl; B.tap_counts = d_tap_counts;
	B.cap = cap;
	if (tw) hipExtLaunchKernelGGL(demod_kernel<true>, dim3((unsigned)nch), dim3(DM_THREADS), (unsigned)lds_bytes, st, start, done, 0, pv->t, B, (const cf *)chan_out, out_count, outs, nblk, nch);
	else hipExtLaunchKernelGGL(demod_kernel<false>, dim3((unsigned)nch), dim3(DM_THREADS), (unsigned)lds_bytes, st, start, done, 0, pv->t, B, (const cf *)chan_out, out_count, outs, nblk, nch);
	D_TRY(hipGetLastError());
	return 0;
}

int Demod::enqueue_decode(int buf, hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
	DemodPriv *pv = priv_of(this);
	if (!pv) return HFDL_GPU_EINVAL;
	const uint64_t i = decodes++;
	if (i + 1 != launches) return HFDL_GPU_EINVAL;      // one decoder launch per demodulator launch, in order
	hipExtLaunchKernelGGL(burst_decode_kernel, dim3((unsigned)nch), dim3(64), (unsigned)k5_lds_bytes(), st, start, stop, 0, (const FrameRec *)(d_frames + (size_t)(i & 1) * nch), d_counts,
			d_counts + 4 + (int)(i & 3), d_counts + 4 + (int)((i + 2) & 3), nch,
			(const cf *)d_data, pv->t.scrambler, (const int32_t *)d_freqs, d_pdus, pdu_cap);
	// what the ring holds once this block is done, for a host that collects without draining the pipeline
	D_TRY(hipMemcpyAsync(h_snap + 4 * (buf & 1), d_counts, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
	if (separate_decode) {
		if (!ev_dec[i & 1]) D_TRY(hipEventCreateWithFlags(&ev_dec[i & 1], hipEventDisableTiming));
		D_TRY(hipEventRecord(ev_dec[i & 1], st));
	}
	D_TRY(hipGetLastError());
	return 0;
}

// copy ring entries [taken, produced) to the host, at most `max`; every entry below `produced` is complete.
// `produced` may be an OLDER snapshot than `taken` (a draining poll followed by a snapshot poll with no push in between):
// the difference is taken as signed, so a stale snapshot yields nothing instead of wrapping.
int Demod::take(unsigned produced, hfdl_gpu_pdu *out, int32_t max, int32_t *n, hipStream_t st)
{
	*n = 0;
	const int32_t avail = (int32_t)(produced - taken);
	if (avail <= 0 || max <= 0) return 0;
	if (!out) return HFDL_GPU_EINVAL;              // a NULL buffer never discards PDUs
	unsigned have = (unsigned)avail;
	if (have > (unsigned)pdu_cap) have = (unsigned)pdu_cap;
	const unsigned cnt = have < (unsigned)max ? have : (unsigned)max;
	// ring entries [taken, taken + cnt) in pieces that neither wrap nor exceed the bounce buffer; each piece is copied on the
	// collection stream (beside whatever kernels are running) and waited for on that stream alone
	for (unsigned done = 0; done < cnt;) {
		const unsigned first = (taken + done) % (unsigned)pdu_cap;
		unsigned n1 = cnt - done;
		if (n1 > (unsigned)pdu_cap - first) n1 = (unsigned)pdu_cap - first;
		if (n1 > (unsigned)bounce_cap) n1 = (unsigned)bounce_cap;
		D_TRY(hipMemcpyAsync(h_pdu_bounce, d_pdus + first, sizeof(hfdl_gpu_pdu) * n1, hipMemcpyDeviceToHost, st_collect));
		D_TRY(hipStreamSynchronize(st_collect));
		std::memcpy(out + done, h_pdu_bounce, sizeof(hfdl_gpu_pdu) * n1);
		done += n1;
	}
	taken += cnt;
	D_TRY(hipMemsetD32Async((hipDeviceptr_t)(d_counts + 3), (int)taken, 1, st));    // ordered after the blocks already queued
	*n = (int32_t)cnt;
	return 0;
}

int Demod::collect(hfdl_gpu_pdu *out, int32_t max, int32_t *n, hipStream_t st)
{
	int counts[4];
	D_TRY(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, st));
	D_TRY(hipStreamSynchronize(st));
	dropped = (uint32_t)counts[2];
	return take((unsigned)counts[1], out, max, n, st);
}

int Demod::collect_snapshot(int buf, hfdl_gpu_pdu *out, int32_t max, int32_t *n, hipStream_t st)
{
	const volatile int *snap = h_snap + 4 * (buf & 1);
	const uint32_t d = (uint32_t)snap[2];
	if ((int32_t)(d - dropped) > 0) dropped = d;       // monotone: an older snapshot never takes the count back
	return take((unsigned)snap[1], out, max, n, st);
}

int Demod::tap(int what, int channel, const void **src, size_t *nfloats)
{
	if (!taps_on || !taps_enabled) return HFDL_GPU_EINVAL;
	int counts[2];
	D_TRY(hipMemcpy(counts, d_tap_counts + 2 * channel, sizeof(counts), hipMemcpyDeviceToHost));
	switch (what) {
	case HFDL_GPU_TAP_RESAMPLED: *src = d_tap_rs + (size_t)channel * cap; *nfloats = 2 * (size_t)counts[0]; return 0;
	case HFDL_GPU_TAP_MF_OUT: *src = d_tap_mf + (size_t)channel * cap; *nfloats = 2 * (size_t)counts[0]; return 0;
	case HFDL_GPU_TAP_SYMBOLS: *src = d_tap_sym + (size_t)channel * cap; *nfloats = 2 * (size_t)counts[1]; return 0;
	case HFDL_GPU_TAP_AGC_LEVEL: *src = d_tap_lvl + (size_t)channel * cap; *nfloats = (size_t)counts[0]; return 0;
	default: return HFDL_GPU_EINVAL;
	}
}

static void fill_stats(const ChanScalars &sc, hfdl_gpu_channel_stats *out)
{
	out->a2_found = sc.cnt_a2_found; out->m1_found = sc.cnt_m1_found; out->m1_not_found = sc.cnt_m1_not_found; out->frames = sc.cnt_frames;
	out->noise_floor_db = 20.0f * log10f(sc.noise_floor);
	out->agc_level = 1.0f / sc.agc_g;
	out->costas_dphi = sc.dphi;
	out->framer_state = sc.fr_state;
	out->sample_cnt = sc.sample_cnt; out->symbol_cnt = sc.symbol_cnt;
	out->a1_found = sc.cnt_a1_found;
	out->a1_corr_avg = sc.cnt_a1_found ? (float)sc.sum_a1_dev / 127.0f / (float)sc.cnt_a1_found : 0.f;
	out->a2_corr_avg = sc.cnt_a2_found ? (float)sc.sum_a2_dev / 127.0f / (float)sc.cnt_a2_found : 0.f;
	out->m1_corr_avg = sc.cnt_m1_found ? (float)sc.sum_m1_dev / 127.0f / (float)sc.cnt_m1_found : 0.f;
	out->train_bits_bad = sc.cum_train_bad; out->train_bits_total = sc.cum_train_total;
}

int Demod::stats(int channel, hfdl_gpu_channel_stats *out)
{
	ChanScalars sc;
	D_TRY(hipMemcpy(&sc, &d_states[channel].s, sizeof(sc), hipMemcpyDeviceToHost));
	fill_stats(sc, out);
	return 0;
}

// all channels in one strided copy; does not wait for blocks in flight (each field is read whole, the set may straddle a block)
int Demod::stats_all(hfdl_gpu_channel_stats *out, int n)
{
	if (n > nch) return HFDL_GPU_EINVAL;
	ChanScalars *sc = (ChanScalars *)h_stats_bounce;
	D_TRY(hipMemcpy2DAsync(sc, sizeof(ChanScalars), &d_states[0].s, sizeof(ChanState), sizeof(ChanScalars), (size_t)n, hipMemcpyDeviceToHost, st_collect));
	D_TRY(hipStreamSynchronize(st_collect));
	for (int i = 0; i < n; i++) fill_stats(sc[i], out + i);
	return 0;
}

void Demod::release()
{
	void *ptrs[] = { d_tables, d_states, d_data, d_frames, d_counts, d_pdus, d_freqs, d_tap_rs, d_tap_mf, d_tap_sym, d_tap_lvl, d_tap_counts };
	for (void *p : ptrs) if (p) (void)hipFree(p);
	if (h_snap) (void)hipHostFree(h_snap);
	h_snap = nullptr;
	if (st_collect) { (void)hipStreamSynchronize(st_collect); (void)hipStreamDestroy(st_collect); st_collect = nullptr; }
	if (h_pdu_bounce) (void)hipHostFree(h_pdu_bounce);
	if (h_stats_bounce) (void)hipHostFree(h_stats_bounce);
	h_pdu_bounce = nullptr; h_stats_bounce = nullptr;
	for (auto &e : ev_dec) { if (e) (void)hipEventDestroy(e); e = nullptr; }
	d_tables = nullptr; d_states = nullptr; d_data = nullptr; d_frames = nullptr; d_counts = nullptr; d_pdus = nullptr; d_freqs = nullptr;
	d_tap_rs = d_tap_mf = d_tap_sym = nullptr; d_tap_lvl = nullptr; d_tap_counts = nullptr;
	delete (DemodPriv *)priv;
	priv = nullptr;
}

#ifdef HFDL_LAB
// the clock-probe ring of this translation unit's kernels: records made since the last read, oldest first (at most `max`)
int demod_clock_probe_read(unsigned long long *out, int max, int *n)
{
	unsigned cnt = 0;
	D_TRY(hipDeviceSynchronize());
	D_TRY(hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(hfdl_clk_probe_n), sizeof(cnt)));
	std::vector<unsigned long long> all(4096 * 4);
	D_TRY(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(hfdl_clk_probe), sizeof(unsigned long long) * all.size()));
	const unsigned have = cnt < 4096u ? cnt : 4096u;
	int k = 0;
	for (unsigned i = 0; i < have && k < max; i++) {
		const unsigned slot = (cnt - have + i) & 4095u;
		for (int j = 0; j < 4; j++) out[4 * k + j] = all[4 * slot + j];
		k++;
	}
	*n = k;
	cnt = 0;
	D_TRY(hipMemcpyToSymbol(HIP_SYMBOL(hfdl_clk_probe_n), &cnt, sizeof(cnt)));
	return 0;
}
#endif

// the DemodTables image resident on the device and the device's own evaluation of hfdl_constants()
int Demod::read_constants(void *tables, size_t tables_bytes, void *constants, size_t constants_bytes)
{
	if (tables_bytes != sizeof(DemodTables) || constants_bytes != sizeof(HfdlConstants) || !d_tables) return HFDL_GPU_EINVAL;
	DevBuf d_k;
	D_TRY(d_k.alloc(sizeof(HfdlConstants)));
	D_TRY(hipMemset(d_k.p, 0xff, sizeof(HfdlConstants)));
	hipLaunchKernelGGL(constants_kernel, dim3(1), dim3(64), 0, nullptr, d_k.as<HfdlConstants>());
	D_TRY(hipDeviceSynchronize());
	D_TRY(hipGetLastError());
	D_TRY(hipMemcpy(constants, d_k.p, sizeof(HfdlConstants), hipMemcpyDeviceToHost));
	D_TRY(hipMemcpy(tables, d_tables, sizeof(DemodTables), hipMemcpyDeviceToHost));
	return 0;
}

int demod_viterbi_batch(const uint8_t *soft, int32_t nbits, int32_t nframes, uint8_t *out, double *kernel_ms)
{
	const size_t in_bytes = (size_t)nframes * 2 * nbits, out_bytes = (size_t)nframes * ((nbits + 7) / 8);
	const size_t lds = viterbi_lds_bytes(nbits);
	if (lds > 160 * 1024) return HFDL_GPU_ERANGE;
	DevBuf d_in, d_out;
	D_TRY(d_in.alloc(in_bytes));
	D_TRY(d_out.alloc(out_bytes));
	D_TRY(hipMemcpy(d_in.p, soft, in_bytes, hipMemcpyHostToDevice));
	D_TRY(hipMemset(d_out.p, 0, out_bytes));
	int rc = set_big_lds((const void *)viterbi_batch_kernel, lds);
	if (rc) return rc;
	KernelTimer tm(kernel_ms != nullptr);
	hipLaunchKernelGGL(viterbi_batch_kernel, dim3((unsigned)nframes), dim3(64), lds, nullptr, d_in.as<const uint8_t>(), nbits, d_out.as<uint8_t>());
	if (kernel_ms) *kernel_ms = tm.stop();
	D_TRY(hipDeviceSynchronize());
	D_TRY(hipGetLastError());
	D_TRY(hipMemcpy(out, d_out.p, out_bytes, hipMemcpyDeviceToHost));
	return 0;
}

int demod_crc16(const uint8_t *data, uint32_t len, uint16_t crc_init, uint16_t *crc)
{
	DevBuf d_in, d_out;
	D_TRY(d_in.alloc(len));
	D_TRY(d_out.alloc(sizeof(uint32_t)));
	if (len) D_TRY(hipMemcpy(d_in.p, data, len, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(crc16_kernel, dim3(1), dim3(64), 0, nullptr, d_in.as<const uint8_t>(), len, (uint32_t)crc_init, d_out.as<uint32_t>());
	D_TRY(hipDeviceSynchronize());
	D_TRY(hipGetLastError());
	uint32_t v = 0;
	D_TRY(hipMemcpy(&v, d_out.p, sizeof(v), hipMemcpyDeviceToHost));
	*crc = (uint16_t)v;
	return 0;
}

int demod_pdu_triage_batch(const uint8_t *octets, const int32_t *lens, int32_t npdus, int32_t stride, uint8_t *fcs_status, uint8_t *kind, uint16_t *hdr_len)
{
	DevBuf d_oct, d_lens, d_fcs, d_kind, d_hl;
	const size_t n = (size_t)npdus;
	D_TRY(d_oct.alloc(n * (size_t)stride));
	D_TRY(d_lens.alloc(n * sizeof(int32_t)));
	D_TRY(d_fcs.alloc(n)); D_TRY(d_kind.alloc(n)); D_TRY(d_hl.alloc(n * sizeof(uint16_t)));
	D_TRY(hipMemcpy(d_oct.p, octets, n * (size_t)stride, hipMemcpyHostToDevice));
	D_TRY(hipMemcpy(d_lens.p, lens, n * sizeof(int32_t), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(pdu_triage_kernel, dim3((unsigned)((npdus + 63) / 64)), dim3(64), 0, nullptr, d_oct.as<const uint8_t>(), d_lens.as<const int32_t>(),
			npdus, stride, d_fcs.as<uint8_t>(), d_kind.as<uint8_t>(), d_hl.as<uint16_t>());
	D_TRY(hipDeviceSynchronize());
	D_TRY(hipGetLastError());
	D_TRY(hipMemcpy(fcs_status, d_fcs.p, n, hipMemcpyDeviceToHost));
	D_TRY(hipMemcpy(kind, d_kind.p, n, hipMemcpyDeviceToHost));
	D_TRY(hipMemcpy(hdr_len, d_hl.p, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
	return 0;
}

int demod_psk_slice_batch(int arity, const float *xy, int32_t n, uint32_t *sym, float *phase_error)
{
	DemodTables h;
	build_demod_tables(h, 0.6912f);
	DevBuf d_x, d_pts, d_sym, d_err;
	D_TRY(d_x.alloc(sizeof(cf) * (size_t)n));
	D_TRY(d_pts.alloc(sizeof(h.psk_pts)));
	D_TRY(d_sym.alloc(sizeof(uint32_t) * (size_t)n));
	D_TRY(d_err.alloc(sizeof(float) * (size_t)n));
	D_TRY(hipMemcpy(d_x.p, xy, sizeof(cf) * (size_t)n, hipMemcpyHostToDevice));
	D_TRY(hipMemcpy(d_pts.p, h.psk_pts, sizeof(h.psk_pts), hipMemcpyHostToDevice));
	const int waves = n < 64 * 256 ? (n + 63) / 64 : 256;
	hipLaunchKernelGGL(psk_slice_kernel, dim3((unsigned)waves), dim3(64), 0, nullptr, arity, d_x.as<const cf>(), n, d_pts.as<const float>(), d_sym.as<uint32_t>(), d_err.as<float>());
	D_TRY(hipDeviceSynchronize());
	D_TRY(hipGetLastError());
	D_TRY(hipMemcpy(sym, d_sym.p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
	D_TRY(hipMemcpy(phase_error, d_err.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
	return 0;
}

int demod_lpdu_walk_batch(const uint8_t *octets, const int32_t *lens, int32_t npdus, int32_t stride, uint8_t *counts)
{
	DevBuf d_oct, d_lens, d_cnt;
	const size_t n = (size_t)npdus;
	D_TRY(d_oct.alloc(n * (size_t)stride));
	D_TRY(d_lens.alloc(n * sizeof(int32_t)));
	D_TRY(d_cnt.alloc(n * 5));
	D_TRY(hipMemcpy(d_oct.p, octets, n * (size_t)stride, hipMemcpyHostToDevice));
	D_TRY(hipMemcpy(d_lens.p, lens, n * sizeof(int32_t), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(lpdu_walk_kernel, dim3((unsigned)((npdus + 63) / 64)), dim3(64), 0, nullptr, d_oct.as<const uint8_t>(), d_lens.as<const int32_t>(), npdus, stride, d_cnt.as<uint8_t>());
	D_TRY(hipDeviceSynchronize());
	D_TRY(hipGetLastError());
	D_TRY(hipMemcpy(counts, d_cnt.p, n * 5, hipMemcpyDeviceToHost));
	return 0;
}

int demod_burst_decode_batch(const float *symbols, const int32_t *modes, const int32_t *bitmask_lsb, int32_t nframes,
		uint8_t *octets, int32_t *lens, double *kernel_ms)
{
	DemodTables h;
	build_demod_tables(h, 0.6912f);
	std::vector<FrameRec> fr((size_t)nframes);
	std::vector<cf> data((size_t)nframes * 2 * MAX_DATA_SYMBOLS);
	std::vector<int32_t> freqs((size_t)nframes, 0);
	size_t off = 0;
	for (int i = 0; i < nframes; i++) {
		const ModeParams mp = mode_params(modes[i]);
		const size_t nsym = (size_t)mp.segments * 30;
		std::memcpy(&data[(size_t)i * 2 * MAX_DATA_SYMBOLS], symbols + 2 * off, sizeof(cf) * nsym);
		off += nsym;
		FrameRec &f = fr[(size_t)i];
		std::memset(&f, 0, sizeof(f));
		f.channel = i; f.slot = 0; f.mode = modes[i]; f.bitmask_lsb = bitmask_lsb[i] & 1;
		f.signal_level = 1.0f; f.noise_floor = 1.0f;
		f.sample_index = (uint64_t)i;
	}
	DevBuf d_scr, d_fr, d_data, d_counts, d_freqs, d_pdus;
	int counts[8] = { 0, 0, 0, 0, nframes, 0, 0, 0 };
	D_TRY(d_scr.alloc(DEC_CONST_BYTES));
	D_TRY(hipMemcpy(d_scr.p, h.scrambler, DEC_CONST_BYTES, hipMemcpyHostToDevice));
	D_TRY(d_fr.alloc(sizeof(FrameRec) * fr.size()));
	D_TRY(hipMemcpy(d_fr.p, fr.data(), sizeof(FrameRec) * fr.size(), hipMemcpyHostToDevice));
	D_TRY(d_data.alloc(sizeof(cf) * data.size()));
	D_TRY(hipMemcpy(d_data.p, data.data(), sizeof(cf) * data.size(), hipMemcpyHostToDevice));
	D_TRY(d_counts.alloc(sizeof(counts)));
	D_TRY(hipMemcpy(d_counts.p, counts, sizeof(counts), hipMemcpyHostToDevice));
	D_TRY(d_freqs.alloc(sizeof(int32_t) * freqs.size()));
	D_TRY(hipMemcpy(d_freqs.p, freqs.data(), sizeof(int32_t) * freqs.size(), hipMemcpyHostToDevice));
	D_TRY(d_pdus.alloc(sizeof(hfdl_gpu_pdu) * (size_t)nframes));
	int rc = set_big_lds((const void *)burst_decode_kernel, k5_lds_bytes());
	if (rc) return rc;
	KernelTimer tm(kernel_ms != nullptr);
	hipLaunchKernelGGL(burst_decode_kernel, dim3((unsigned)nframes), dim3(64), k5_lds_bytes(), nullptr, d_fr.as<const FrameRec>(), d_counts.as<int>(), d_counts.as<int>() + 4, (int *)nullptr,
			nframes, d_data.as<const cf>(), d_scr.as<const uint8_t>(), d_freqs.as<const int32_t>(), d_pdus.as<hfdl_gpu_pdu>(), nframes);
	if (kernel_ms) *kernel_ms = tm.stop();
	D_TRY(hipDeviceSynchronize());
	D_TRY(hipGetLastError());
	std::vector<hfdl_gpu_pdu> out((size_t)nframes);
	D_TRY(hipMemcpy(out.data(), d_pdus.p, sizeof(hfdl_gpu_pdu) * out.size(), hipMemcpyDeviceToHost));
	for (int i = 0; i < nframes; i++) lens[i] = 0;
	for (auto &p : out) {       // PDU slots are claimed in completion order: route by channel (= frame index)
		if (p.channel < 0 || p.channel >= nframes) continue;
		lens[p.channel] = p.len;
		std::memcpy(octets + (size_t)p.channel * HFDL_GPU_PDU_MAX_OCTETS, p.octets, (size_t)p.len);
	}
	return 0;
}

}  // namespace hfdl
