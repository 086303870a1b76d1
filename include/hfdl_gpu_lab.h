/*
 * hfdl_gpu_lab.h -- measurement aids of the LABORATORY build of the HFDL front end: libhfdl_gpu_lab.so
 *
 * The same sources as libhfdl_gpu.so compiled with -DHFDL_LAB (dumphfdl_amd/csrc/build.sh lab): every entry point of
 * include/hfdl_gpu.h, plus the fold tiling sweep, the probes below and the A/B environment switches the scripts under profiles/
 * use.  Nothing here is part of the drop-in boundary; the product library exports none of it (tests/test_host_lib_cpu.py checks).
 *
 *   HFDL_GPU_FFT_STREAM=1     forward FFTs of the half being filled on a stream of their own (measured slower in round 4)
 *   HFDL_GPU_DECODE_STREAM=0  burst decoders back on the demodulators' stream
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
/* what the board's HBM delivers to a read-only streaming kernel with the fold's access pattern (reads the resident taps) */
int  hfdl_gpu_lab_stream_read_probe(hfdl_gpu_frontend *fe, double *gb_per_s);

#ifdef __cplusplus
}
#endif
#endif
