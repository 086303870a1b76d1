"""What separates the device demodulator from the oracle, SHOWN (profiles/strict_study.py; the full table is profiles/r04/strict_study.md).

Test-only builds of the library (dumphfdl_amd/csrc/build_strict.sh -> build/strict/, made by this test when missing): `strict_0` runs the demodulator as the one-lane serial loop of
tests/hostsim/serial_demod.h on the fixed fp32 sequences of tests/hostsim/shared_math.h -- the arithmetic the oracle runs under
orc_variant.shared_math; `strict_15` is the same loop with the shipped pipeline's four fast forms (DPP-order sums, hardware log / exp /
rcp in the AGC, hardware sin / cos, nearest-point slicer) emulated operation for operation.

  * strict_0 fed with the ORACLE's channelizer output: every symbol (bit pattern) and every PDU of every SNR bin equals the oracle's.
    The device's control flow, state carrying, framer, burst decoder -- everything that is not a rounding -- is the oracle's.
  * strict_15 == the shipped library, feed for feed, bin for bin: the shipped three-wave pipeline differs from that serial loop in
    those four forms and in nothing else.
  * what is left between the shipped library and the oracle below +2 dB is therefore rounding: of the device's own channelizer (a
    different FFT factorisation), of the AGC's hardware transcendentals and of the scan-order sums (strict_study.md splits it up)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
pytestmark = pytest.mark.gpu


def test_strict_build_is_the_oracle_and_the_shipped_pipeline_is_strict_plus_four_fast_forms(gpu):
    import subprocess
    import strict_study as S
    libs = [os.path.join(ROOT, "build", "strict", "libhfdl_gpu_strict_%d.so" % f) for f in (0, 15)]
    product = os.path.join(ROOT, "dumphfdl_amd", "libhfdl_gpu.so")
    if not all(os.path.exists(l) and os.path.getmtime(l) >= os.path.getmtime(product) for l in libs):
        # test-only libraries, built where the test runs (hipcc cross-compiles gfx950 anywhere): never part of the product build.  They
        # link the product build's channelizer objects: one that is older than the product library carries another channelizer
        subprocess.check_call(["bash", os.path.join(ROOT, "dumphfdl_amd", "csrc", "build_strict.sh"), "0", "15"], stdout=subprocess.DEVNULL)
    out = S.run_study([-6, -2, 2], bursts_per_channel=2, builds=("shipped", "strict_0", "strict_15"))
    rows = out["rows"]
    pick = lambda build, feed, against: [r for r in rows if (r["build"], r["feed"], r["against"]) == (build, feed, against)]
    # 1. same arithmetic, same input -> same bits: symbols and PDUs
    t = out["taps"]["base"]
    assert t["symbols_compared"] > 50_000 and t["symbols_with_other_bits"] == 0 and t["pdus_identical"] and t["pdus_gpu"] >= 8, t
    base0 = pick("strict_0", "base", "oracle(shared_math)")
    assert len(base0) == 3 and all(r["identical"] and r["pdus"] >= 60 for r in base0), base0
    # 2. the shipped pipeline = that loop + the four fast forms
    for feed in ("base", "wide"):
        same = pick("strict_15", feed, "shipped")
        assert len(same) == 3 and all(r["identical"] for r in same), same
    # 3. and the device's own channelizer in front changes what rounding changes: nothing at +2 dB, a frame or two per hundred below,
    #    every correctly decoded frame common
    for r in pick("strict_0", "wide", "oracle(shared_math)") + pick("shipped", "wide", "oracle(libm)") + pick("shipped", "base", "oracle(libm)"):
        assert r["recovered"] == r["other_recovered"], r
        if r["snr_db"] >= 2:
            assert r["identical"], r
        else:
            assert r["only_here"] + r["only_there"] <= 0.1 * (r["pdus"] + r["other_pdus"]), r
