"""GPU test: the reference's static data as the DEVICE holds it.

The demodulator's table image is read back from device memory and the named constants are evaluated by a kernel
(hfdl_gpu_lab_read_constants, laboratory build: the same sources as the product library); both are compared with
tests/golden/hfdl_constants.json -- the numbers parsed from the text of the reference's src/hfdl.c (tests/golden/make_constants.py)."""
import ctypes as C
import json
import os
import pytest

from test_constants_cpu import DemodTables, HfdlConstants, check_constants_struct, check_tables_struct, GOLD

pytestmark = pytest.mark.gpu


def test_device_tables_and_constants_match_the_reference_text(gpu):
    from dumphfdl_amd import frontend as F
    K = json.load(open(os.path.join(GOLD, "hfdl_constants.json")))
    lab = F.load_lab()
    fs, cf = 250000, 10_000_000
    fe = gpu.Frontend(fs, cf, [9_958_000, 10_061_000], lib=lab)
    try:
        t, k = DemodTables(), HfdlConstants()
        F._check(lab.hfdl_gpu_lab_read_constants(fe._h, C.byref(t), C.sizeof(t), C.byref(k), C.sizeof(k)), lab)
        check_constants_struct(k, K)
        check_tables_struct(t, K)
        # a wrong size is refused, not copied
        assert lab.hfdl_gpu_lab_read_constants(fe._h, C.byref(t), C.sizeof(t) - 4, C.byref(k), C.sizeof(k)) != 0
    finally:
        fe.close()
