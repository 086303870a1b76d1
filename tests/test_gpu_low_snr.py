"""Where "the decoded-frame set is identical to the CPU reference's" holds: GPU vs strict oracle on 256 bursts per 2 dB bin of in-channel
SNR, -8 .. +10 dB, all eight modes (profiles/low_snr_parity.py).  The device demodulator differs from the oracle in rounding only
(v_sin / v_cos / v_log / v_exp, DPP tree sums, the nearest-point slicer); profiles/r03_variant_sensitivity.md shows what ANY rounding-level
change does to the oracle itself: nothing from -2 dB up, a frame or two per 200 below."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
pytestmark = pytest.mark.gpu


def test_frame_sets_identical_down_to_marginal_snr(gpu, oracle):
    import low_snr_parity as L
    bins = list(range(-8, 11, 2))
    rows = L.sweep(gpu, oracle, bins, bursts_per_channel=4)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(rows=rows), open(os.path.join(ROOT, "gpurun_out", "low_snr_sweep.json"), "w"), indent=1)
    for r in rows:
        assert r["bursts"] == 256
        differing = r["gpu_only"] + r["oracle_only"]
        if r["snr_db"] >= 2:
            # every frame either side dispatches, same detection sample, same octets -- including the frames with bit errors
            assert r["identical"] and r["gpu_pdus"] >= 250, r
        elif r["snr_db"] >= -4:
            assert differing <= 0.02 * (r["gpu_pdus"] + r["oracle_pdus"]), r          # a frame or two per 200 whose fate hangs on one soft decision
            assert abs(r["gpu_recovered"] - r["oracle_recovered"]) <= 2, r
        else:
            assert differing <= 0.05 * (r["gpu_pdus"] + r["oracle_pdus"]), r
            assert abs(r["gpu_recovered"] - r["oracle_recovered"]) <= 4, r
        assert r["same_place_other_octets"] <= max(1, differing // 2), r
