// strict_demod_kernels.hip -- TEST-ONLY translation unit (dumphfdl_amd/csrc/build_strict.sh, never part of libhfdl_gpu.so): the product's
// demodulator / burst-decoder kernels compiled with the demodulator running the one-lane serial loop of serial_demod.h on the
// fixed-sequence elementary functions of shared_math.h -- the arithmetic the oracle runs under orc_variant.shared_math -- so that device
// and oracle can be compared BIT FOR BIT, and the pipeline's fast forms can be switched back on one at a time (-DHFDL_DM_STRICT_FAST=
// 1 sums | 2 AGC | 4 trig | 8 slicer).  profiles/strict_study.py, tests/test_gpu_strict.py, DESIGN.md section 5.1.
#define HFDL_DM_STRICT 1
#include <hip/hip_runtime.h>
#define SM_FN __host__ __device__ static inline
#include "shared_math.h"
#define HFDL_ATAN2F sm_atan2f
#define HFDL_DM_SERIAL_LOOP "../../tests/hostsim/serial_demod.h"
#include "../../dumphfdl_amd/csrc/demod_kernels.hip"
