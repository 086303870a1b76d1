/*
 * hfdl_gpu_lab.h -- measurement aids of the LABORATORY build of the HFDL front end: libhfdl_gpu_lab.so
 *
 * The same sources as libhfdl_gpu.so compiled with -DHFDL_LAB (dumphfdl_amd/csrc/build.sh lab): every entry point of
 * include/hfdl_gpu.h, plus the fold tiling sweep, the probes below and the A/B environment switches the scripts under profiles/
 * use.  Nothing here is part of the drop-in boundary; the product library exports none of it (tests/test_host_lib_cpu.py checks).
 *
 *   HFDL_GPU_FFT_STREAM=1     forward FFTs of the half being filled on a stream of their own (measured slower in round 4)
 *   HFDL_GPU_DECODE_STREAM=0  burst decoders back on the demodulators' stream
 *   HFDL_GPU_FOLD_TILE=i      the i-th entry of fold_kernels.hip fold_variants[] instead of the first that fits (25: the thirty-two-column
 *                             tiling with two waves of 208 registers per SIMD)
 *   HFDL_GPU_FOLD_SLICES=s    slices of alias rows per fold workgroup column (a power of two; default: channels x slices >= 256)
 *   HFDL_GPU_FOLD_RAMP=0      the first half after a drain as long as the others (default: 16 blocks where halves are 32)
 *   HFDL_GPU_FOLD_BOUND=0|1   override "the fold bounds the block" (128 channels and more): halves of 8 / 32, demodulator batch
 *   HFDL_GPU_CU_SPLIT=k       k = 2 .. 8: the demodulators' stream on every k-th CU, the channelizer's on the others (hipExtStreamCreateWithCUMask)
 *   HFDL_GPU_PROBE_VERBOSE=1  the stream-read probe prints every variant
 */
#ifndef HFDL_GPU_LAB_H
#define HFDL_GPU_LAB_H
#include "hfdl_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the compiled tilings of the matrix-pipe fold kernel: desc = { channel octets per wave, groups of four blocks (4: the sixteen-column
 * form, 1: the four-column form), waves per workgroup, quads of alias rows of loads in flight, max blocks per launch, tap layout } */
int  hfdl_gpu_lab_fold_variant_count(void);
int  hfdl_gpu_lab_fold_variant_describe(int variant, int32_t desc[6]);
/* `reps` timed launches of one tiling (variant -1: the plain-VALU FMA-chain reference kernel) folding `nb` blocks over the front end's
 * resident taps and the spectra of its newest half; *checksum sums the bit patterns of the partial sums (equal for bit-identical kernels
 * at the same nb) */
int  hfdl_gpu_lab_fold_variant_probe(hfdl_gpu_frontend *fe, int variant, int nb, int reps, double *avg_ms, double *best_ms, uint64_t *checksum);
/* The demodulator's constant tables as they lie in device memory (`tables`: struct DemodTables of csrc/demod_tables.h) and the named
 * constants of the reference's hot path as the DEVICE evaluates them (`constants`: struct HfdlConstants of csrc/demod_logic.h): both are
 * compared with the reference's own text, tests/golden/hfdl_constants.json.  The byte counts must equal the structs' sizes
 * (tests/hostsim reports them). */
int  hfdl_gpu_lab_read_constants(hfdl_gpu_frontend *fe, void *tables, size_t tables_bytes, void *constants, size_t constants_bytes);
/* The shader clock the probed launches ran at, measured from inside them (s_memtime against the constant 100 MHz s_memrealtime): records of
 * four 64-bit words {tag, shader cycles, reference ticks, reference tick at the start}, oldest first, made since the last read.
 * which = 0: fold launches (tag = columns x 100 + blocks; the workgroup in the middle of the grid), 1: demodulator launches (tag = 1000 +
 * blocks of the launch; workgroup 0, the whole launch).  At most 1024 / 4096 records are kept. */
int  hfdl_gpu_lab_clock_probe_read(int which, uint64_t *records, int32_t max, int32_t *n);
/* what the board's HBM delivers to a read-only streaming kernel with the fold's access pattern (reads the resident taps) */
int  hfdl_gpu_lab_stream_read_probe(hfdl_gpu_frontend *fe, double *gb_per_s);

#ifdef __cplusplus
}
#endif
#endif
