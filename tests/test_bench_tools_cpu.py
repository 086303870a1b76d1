"""bench.py's roofline bookkeeping (no GPU): the per-launch byte model, the launch shapes of a timed region, and the guard that keeps
a traffic record measured on other device code -- or on other launch shapes -- out of the line."""
import json
import os

import pytest

import bench


class G:          # the cfg3 geometry (hfdl_gpu_geometry fields the model reads)
    input_size, channels, fft_size, post_input_size, post_decimation = 7340032, 256, 8388608, 3584, 2


def test_algorithmic_bytes_per_launch_restates_the_per_block_model():
    assert bench.alg_bytes_per_launch(G, 1) == bench.alg_bytes_per_block(G) == 17_242_259_456          # 2349.1 B/sample
    b8 = bench.alg_bytes_per_launch(G, 8)
    assert b8 == 8 * 8 * G.input_size + 256 * 8 * G.fft_size + 256 * 8 * 8 * 1792 == 17_678_991_360   # the taps ONCE for 8 blocks
    assert b8 < 8 * bench.alg_bytes_per_block(G) / 7.7


def test_algorithmic_bytes_at_sixteen_blocks_per_launch():
    b16 = bench.alg_bytes_per_launch(G, 16)
    assert b16 == 8 * 16 * G.input_size + 256 * 8 * G.fft_size + 256 * 8 * 16 * 1792 == 18_178_113_536   # the taps ONCE for 16 blocks
    assert b16 < 16 * bench.alg_bytes_per_block(G) / 15.1


def test_stale_traffic_is_withheld(tmp_path, monkeypatch):
    rec = dict(kernel="k", blocks_per_launch=8, measured_at_commit="abc", csrc_sha16="1111", hbm_bytes_per_launch=100,
               per_shape={"8": dict(hbm_bytes_per_launch=100, traffic_over_algorithmic=1.02), "4": dict(hbm_bytes_per_launch=60, traffic_over_algorithmic=1.01)})
    os.makedirs(tmp_path / "profiles")
    json.dump(rec, open(tmp_path / "profiles" / "fold_traffic_cfgX.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_hash", lambda: "1111")
    t, src = bench.traffic_record("cfgX", [8, 8, 4], 3)
    assert t == pytest.approx((100 + 100 + 60) / 3) and src["csrc_matches_head"] and "stale" not in src
    t, src = bench.traffic_record("cfgX", [8, 2], 2)                  # a shape the record does not hold
    assert t is None and "stale" in src
    t, src = bench.traffic_record("cfgX", [8, 8, 4], 4)               # the run launched something else than reconstructed
    assert t is None and "stale" in src
    monkeypatch.setattr(bench, "csrc_hash", lambda: "2222")           # other kernels than the ones measured
    t, src = bench.traffic_record("cfgX", [8, 8, 4], 3)
    assert t is None and src["csrc_matches_head"] is False and "stale" in src


def test_committed_traffic_records_match_the_committed_kernels():
    """The records under profiles/ were measured on THIS tree's dumphfdl_amd/csrc: a kernel edit without a new PMC pass fails here."""
    for wl in ("cfg3", "cfg2", "cfg4"):
        rec = json.load(open(os.path.join(bench.ROOT, "profiles", "fold_traffic_%s.json" % wl)))
        assert rec["csrc_sha16"] == bench.csrc_hash(), \
            "%s: dumphfdl_amd/csrc changed since profiles/fold_traffic_%s.json was measured -- commit, run profiles/stamp.sh, then " \
            "`gpurun -- bash profiles/pmc_passes.sh %s gpurun_out/final <commit>` and copy the record into profiles/" % (wl, wl, wl)
        assert set(rec["per_shape"]) >= ({"8", "4"} if wl == "cfg2" else {"16", "4"})            # the driver's --steps 20 = 16 + 4 (cfg2: 8 + 8 + 4)
    r3 = json.load(open(os.path.join(bench.ROOT, "profiles", "fold_traffic_cfg3.json")))
    # no wasted re-reads on the roofline kernel: reads = the algorithmic bytes; the excess is the partial sums of 16 blocks (0.54 GB,
    # not in the model, written in 32-byte pieces: 1.1 GB measured) -- 3 - 7 % of a launch whose bound is the matrix pipe, not HBM
    assert r3["per_shape"]["16"]["traffic_over_algorithmic"] <= 1.10
    assert r3["per_shape"]["16"]["hbm_read_bytes_per_launch"] <= 1.02 * r3["per_shape"]["16"]["algorithmic_bytes_per_launch"]


def test_xcd_aware_tile_placement_is_a_bijection():
    """fold_kernels.hip maps blockIdx -> (tile, channel group) so that a tile's workgroups land on one XCD (block b runs on XCD b mod 8)
    with the groups varying fastest there; the same arithmetic here: every (tile, group) exactly once, a tile on one XCD, and the
    workgroups of one tile consecutive in that XCD's dispatch order."""
    def place(b, ntile, groups):
        if ntile % 8 == 0:
            xcd, i = b & 7, b >> 3
            return (i // groups) * 8 + xcd, i % groups
        return b % ntile, b // ntile

    for ntile, groups in ((8, 128), (128, 16), (64, 8), (128, 8), (32, 3), (12, 5), (1, 4)):
        seen, xcd_of = set(), {}
        for b in range(ntile * groups):
            t, g = place(b, ntile, groups)
            assert 0 <= t < ntile and 0 <= g < groups and (t, g) not in seen
            seen.add((t, g))
            if ntile % 8 == 0:
                assert xcd_of.setdefault(t, b & 7) == (b & 7)
        assert len(seen) == ntile * groups
        if ntile % 8 == 0:
            for x in range(8):
                order = [place(b, ntile, groups)[0] for b in range(x, ntile * groups, 8)]         # this XCD's blocks in dispatch order
                assert order == sorted(order) and all(order[i:i + groups] == [order[i]] * groups for i in range(0, len(order), groups))


def test_roofline_is_priced_on_the_shape_that_dominates():
    """bench.dominant_shape: the driver's 20 steps = one 16-block launch (sixteen-column form, ~4 ms) + one 4-block launch (four-column
    form, ~2.7 ms): priced on the 16-block shape; HFDL_GPU_FOLD_BATCH=1 (256 one-block launches): on that one; nothing timed: nothing."""
    assert bench.dominant_shape({16: (1, 3.99), 4: (1, 2.72)}) == (16, 1, 3.99)
    assert bench.dominant_shape({1: (256, 256 * 2.6)})[:2] == (1, 256)
    nb, n, avg = bench.dominant_shape({16: (16, 64.0), 8: (1, 3.5)})
    assert (nb, n) == (16, 16) and abs(avg - 4.0) < 1e-12
    assert bench.dominant_shape({4: (2, 5.0), 8: (1, 5.0)})[0] == 8          # a tie goes to the larger shape
    assert bench.dominant_shape({}) == (0, 0, None)
