"""CPU tests: the reference's STATIC DATA pinned by text.

tests/golden/hfdl_constants.json holds the numbers parsed from the initialisers and literals of /root/reference/src/hfdl.c
(tests/golden/make_constants.py, container only; numbers, no source text).  Here the oracle's copies (oracle/*.c) and the product's
copies (dumphfdl_amd/csrc/demod_logic.h, demod_tables.h, through tests/hostsim) are compared with that file -- until round 6 the two
hand-typed copies were only ever compared with each other.  The device-resident copies: tests/test_gpu_constants.py."""
import ctypes as C
import json
import os
import numpy as np
import pytest

from test_host_logic_cpu import build_sim

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = "/root/reference/src/hfdl.c"


@pytest.fixture(scope="module")
def K():
    return json.load(open(os.path.join(GOLD, "hfdl_constants.json")))


@pytest.fixture(scope="module")
def sim():
    return build_sim("libhostsim.so", [])


# ---- expected values, computed from the JSON alone ----

def bits127(bits):
    """a 127-bit sequence as (hi, lo) with the OLDEST bit in bit 126: bsequence_push order (src/hfdl.c:436-437, 449-456)"""
    v = 0
    for b in bits:
        v = ((v << 1) | (int(b) & 1)) & ((1 << 127) - 1)
    return v >> 64, v & ((1 << 64) - 1)


def a_bits(K):
    """bsequence_init(A_bs, A_octets) with A_LEN = 127: liquid pushes num_bits bits, octet by octet, MSB first -- the FIRST 127 of the
    128 bits (the last octet's LSB is padding).  (liquid-dsp is absent here: this is the published routine as the oracle, the product
    and the synthetic transmitter all read it; the numbers are the reference's.)"""
    allbits = [(o >> (7 - i)) & 1 for o in K["A_octets"] for i in range(8)]
    return allbits[:K["defines"]["A_LEN"]]


def m1_bits(K, mode):
    return [K["M1_bits"][(K["M_shifts"][mode] + j) % K["defines"]["M1_LEN"]] for j in range(K["defines"]["M1_LEN"])]


class HfdlConstants(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("prekey_len", "a_len", "m1_len", "m2_len", "t_len", "data_frame_len", "preamble_t_seqs", "single_slot_frame_len",
                                         "max_data_symbols", "max_search_retries", "no_frame_timeout_frames", "symbol_rate", "sps", "eq_len", "mf_taps",
                                         "ss_npfb", "nf_clk_mask")] + \
        [("sampler_states", C.c_int32 * 3), ("framer_states", C.c_int32 * 7), ("modes", (C.c_int32 * 4) * 8)] + \
        [(n, C.c_float) for n in ("corr_a1", "corr_a2", "corr_m1", "costas_alpha", "costas_beta", "costas_err_limit", "costas_runaway_dphi",
                                  "agc_bandwidth", "eq_step", "nf_keep", "nf_take", "nf_bias")] + \
        [("t_seq", C.c_float * 15)]


class DemodTables(C.Structure):
    _fields_ = [("rs_h", C.c_float * (256 * 14)), ("rs_step", C.c_uint32), ("mf", C.c_float * 19),
                ("ss_mf", C.c_float * 288), ("ss_dmf", C.c_float * 288), ("lf_b0", C.c_float), ("lf_a1", C.c_float),
                ("ss_rate_adj", C.c_float), ("eq_h0", C.c_float * 15), ("a_hi", C.c_uint64), ("a_lo", C.c_uint64),
                ("m1_hi", C.c_uint64 * 8), ("m1_lo", C.c_uint64 * 8), ("scrambler", C.c_uint8 * 120), ("scr_pad", C.c_uint8 * 8), ("psk_pts", C.c_float * 32),
                ("corr_tab", C.c_float * 128), ("a1_lo", C.c_int32), ("a1_hi", C.c_int32), ("a2_lo", C.c_int32),
                ("a2_hi", C.c_int32), ("pos_min", C.c_int32), ("thr_pad", C.c_int32)]


def f32(x):
    return np.float32(x)


def check_constants_struct(k, K):
    """an HfdlConstants (host or device evaluation of hfdl_constants()) against the reference's numbers"""
    D, E = K["defines"], K["enums"]
    assert (k.prekey_len, k.a_len, k.m1_len, k.m2_len, k.t_len, k.data_frame_len) == \
        (D["PREKEY_LEN"], D["A_LEN"], D["M1_LEN"], D["M2_LEN"], D["T_LEN"], D["DATA_FRAME_LEN"])
    # PREAMBLE_LEN = 2 A + M1 + M2 + 9 T: the nine training sequences of the preamble
    assert 2 * k.a_len + k.m1_len + k.m2_len + k.preamble_t_seqs * k.t_len == D["PREAMBLE_LEN"]
    assert k.single_slot_frame_len == D["SINGLE_SLOT_FRAME_LEN"] and k.max_data_symbols == D["DATA_SYMBOLS_CNT_MAX"]
    assert k.max_search_retries == D["MAX_SEARCH_RETRIES"]
    assert k.no_frame_timeout_frames == K["decoder_thread"]["max_frames_without_frame"]
    assert (k.symbol_rate, k.sps) == (D["HFDL_SYMBOL_RATE"], D["SPS"])
    assert (k.eq_len, k.mf_taps, k.ss_npfb) == (D["EQ_LEN"], D["HFDL_MF_TAPS_CNT"], D["SYMSYNC_PFB_CNT"])
    assert k.nf_clk_mask == K["decoder_thread"]["noise_floor_clk_mask"] == K["decoder_thread"]["noise_floor_clk_match"]
    assert list(k.sampler_states) == [E["SAMPLER_EMIT_BITS"], E["SAMPLER_EMIT_SYMBOLS"], E["SAMPLER_SKIP"]]
    assert list(k.framer_states) == [E[n] for n in ("FRAMER_A1_SEARCH", "FRAMER_A2_SEARCH", "FRAMER_M1_SEARCH", "FRAMER_M2_SKIP", "FRAMER_EQ_TRAIN",
                                                    "FRAMER_DATA_1", "FRAMER_DATA_2")]
    assert [list(m) for m in k.modes] == K["frame_params"]["modes"]
    assert (f32(k.corr_a1), f32(k.corr_a2), f32(k.corr_m1)) == (f32(D["CORR_THRESHOLD_A1"]), f32(D["CORR_THRESHOLD_A2"]), f32(D["CORR_THRESHOLD_M1"]))
    co = K["costas"]
    alpha = f32(co["alpha"])
    assert f32(k.costas_alpha) == alpha
    assert f32(k.costas_beta) == f32(co["beta_over_alpha_squared"]) * alpha * alpha        # c->beta = 0.047f * c->alpha * c->alpha, in float
    assert f32(k.costas_err_limit) == f32(co["limit"]) and f32(k.costas_runaway_dphi) == f32(co["runaway_dphi"])
    assert f32(k.agc_bandwidth) == f32(K["constructors"]["agc_bandwidth"]) and f32(k.eq_step) == f32(K["constructors"]["eqlms_bw"])
    dt = K["decoder_thread"]
    assert (f32(k.nf_keep), f32(k.nf_take), f32(k.nf_bias)) == (f32(dt["noise_floor_keep"]), f32(dt["noise_floor_take"]), f32(dt["noise_floor_bias"]))
    assert [float(v) for v in k.t_seq] == K["T_seq"][0]
    assert [-float(v) for v in k.t_seq] == K["T_seq"][1]        # the reference's second row is the negation the code applies through the bitmask


def check_tables_struct(t, K):
    """a DemodTables image (host-built or read back from the device) against the reference's numbers"""
    assert np.array_equal(np.frombuffer(t.mf, np.float32), np.array(K["matched_filter"], np.float64).astype(np.float32))
    hi, lo = bits127(a_bits(K))
    assert (t.a_hi, t.a_lo) == (hi, lo)
    for m in range(8):
        hi, lo = bits127(m1_bits(K, m))
        assert (t.m1_hi[m], t.m1_lo[m]) == (hi, lo), m
    # thresholds as match counts: decided exactly like fabsf(corr) > CORR_THRESHOLD_x on corr = 2 m / 127 - 1 in fp32 (src/hfdl.c:781-804)
    m = np.arange(128, dtype=np.float32)
    corr = np.float32(2.0) * m / np.float32(K["defines"]["A_LEN"]) - np.float32(1.0)
    assert np.array_equal(np.frombuffer(t.corr_tab, np.float32), corr)
    for k in range(128):
        assert (abs(corr[k]) > f32(K["defines"]["CORR_THRESHOLD_A1"])) == (k <= t.a1_lo or k >= t.a1_hi)
        assert (abs(corr[k]) > f32(K["defines"]["CORR_THRESHOLD_A2"])) == (k <= t.a2_lo or k >= t.a2_hi)
    # descrambler: the (liquid >= 1.6) branch's numbers under the register the oracle states for that API -- v = (v << 1 | parity(v & taps)),
    # output = the fed-back bit -- 120 symbols, then restart
    d = K["descrambler"]
    v, taps, out = d["liquid_1_6_and_later"]["init"], d["liquid_1_6_and_later"]["genpoly"], []
    for _ in range(d["seq_len"]):
        b = bin(v & taps).count("1") & 1
        v = ((v << 1) | b) & ((1 << d["numbits"]) - 1)
        out.append(b)
    assert bytes(t.scrambler) == bytes(out)
    # the pre-1.6 branch's numbers are the same register written the other way round: genpoly 0x8002 = (0x4001 << 1), init = bit-reversed
    old = d["liquid_before_1_6"]
    assert old["genpoly"] >> 1 == taps and int(format(old["init"], "015b")[::-1], 2) == d["liquid_1_6_and_later"]["init"]


# ---- the tests ----

def test_golden_file_is_what_the_reference_says(K):
    """container only: re-parse the reference and compare with the committed file (the GPU box has no reference: skipped there)"""
    if not os.path.exists(REF):
        pytest.skip("no /root/reference here")
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("make_constants", os.path.join(GOLD, "make_constants.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as d:
        mod.HERE = d
        mod.main()
        assert json.load(open(os.path.join(d, "hfdl_constants.json"))) == K
    # sanity of the parse itself: lengths and a few values no parser bug could fake
    assert len(K["matched_filter"]) == 19 and K["matched_filter"] == K["matched_filter"][::-1] and abs(sum(K["matched_filter"]) - 1.0) < 0.05
    assert sum(K["M1_bits"]) == 64 and len(K["M1_bits"]) == 127          # an m-sequence of length 127 has 64 ones
    assert K["defines"]["SINGLE_SLOT_FRAME_LEN"] == 448 + 531 + 72 * 45


def test_product_constants_match_the_reference_text(sim, K):
    sim.sim_sizeof_constants.restype = C.c_size_t
    assert sim.sim_sizeof_constants() == C.sizeof(HfdlConstants)
    k = HfdlConstants()
    sim.sim_constants.argtypes = [C.c_void_p]
    sim.sim_constants(C.byref(k))
    check_constants_struct(k, K)


def test_product_tables_match_the_reference_text(sim, K):
    sim.sim_sizeof_tables.restype = C.c_size_t
    assert sim.sim_sizeof_tables() == C.sizeof(DemodTables)
    t = DemodTables()
    sim.sim_tables.argtypes = [C.c_float, C.c_void_p]
    sim.sim_tables(0.55296, C.byref(t))
    check_tables_struct(t, K)


def test_oracle_constants_match_the_reference_text(oracle, K):
    L = oracle.lib()
    D, E = K["defines"], K["enums"]
    ints = (C.c_int32 * 29)(); floats = (C.c_float * 14)(); mf = (C.c_float * 19)(); tseq = (C.c_float * 15)()
    L.orc_constants(ints, floats, mf, tseq)
    want_i = [D["PREKEY_LEN"], D["A_LEN"], D["M1_LEN"], D["M2_LEN"], D["T_LEN"], D["DATA_FRAME_LEN"], D["PREAMBLE_LEN"], D["SINGLE_SLOT_FRAME_LEN"],
              D["DATA_SYMBOLS_CNT_MAX"], D["MAX_SEARCH_RETRIES"], K["decoder_thread"]["max_frames_without_frame"], D["HFDL_SYMBOL_RATE"], D["SPS"],
              D["EQ_LEN"], D["HFDL_MF_TAPS_CNT"], D["SYMSYNC_PFB_CNT"], K["decoder_thread"]["noise_floor_clk_mask"],
              E["SAMPLER_EMIT_BITS"], E["SAMPLER_EMIT_SYMBOLS"], E["SAMPLER_SKIP"],
              E["FRAMER_A1_SEARCH"], E["FRAMER_A2_SEARCH"], E["FRAMER_M1_SEARCH"], E["FRAMER_M2_SKIP"], E["FRAMER_EQ_TRAIN"], E["FRAMER_DATA_1"], E["FRAMER_DATA_2"],
              K["constructors"]["symsync_create_kaiser"][0], K["constructors"]["symsync_output_rate"]]
    assert list(ints) == want_i
    co, dt, cs = K["costas"], K["decoder_thread"], K["constructors"]
    alpha = f32(co["alpha"])
    want_f = [f32(D["CORR_THRESHOLD_A1"]), f32(D["CORR_THRESHOLD_A2"]), f32(D["CORR_THRESHOLD_M1"]), alpha, f32(co["beta_over_alpha_squared"]) * alpha * alpha,
              f32(co["limit"]), f32(co["runaway_dphi"]), f32(cs["agc_bandwidth"]), f32(cs["eqlms_bw"]), f32(dt["noise_floor_keep"]), f32(dt["noise_floor_take"]),
              f32(dt["noise_floor_bias"]), f32(cs["noise_floor_init"]), f32(cs["symsync_lf_bw"])]
    assert [f32(v) for v in floats] == want_f
    assert np.array_equal(np.array(mf, np.float32), np.array(K["matched_filter"], np.float64).astype(np.float32))
    assert [float(v) for v in tseq] == K["T_seq"][0]
    # the mode table, the preamble sequences and the training sequence as the oracle's accessors hand them out
    class MP(C.Structure):
        _fields_ = [("arity", C.c_int32), ("segments", C.c_int32), ("code_rate", C.c_int32), ("col_shift", C.c_int32)]
    modes = (MP * 8).in_dll(L, "orc_modes")
    assert [[m.arity, m.segments, m.code_rate, m.col_shift] for m in modes] == K["frame_params"]["modes"]
    b = (C.c_uint8 * 127)()
    L.orc_preamble_A(b)
    assert list(b) == a_bits(K)
    for m in range(8):
        L.orc_preamble_M1(m, b)
        assert list(b) == m1_bits(K, m), m
    t = (C.c_uint8 * 15)()
    L.orc_training_T(t)
    assert [1.0 - 2.0 * v for v in t] == K["T_seq"][0]
    # the descrambler sequence under both API readings of the reference's two branches agrees with the oracle's default
    sb = (C.c_uint8 * 120)()
    L.orc_scrambler_bits(sb, 120)
    d = K["descrambler"]
    v, taps, out = d["liquid_1_6_and_later"]["init"], d["liquid_1_6_and_later"]["genpoly"], []
    for _ in range(d["seq_len"]):
        bit = bin(v & taps).count("1") & 1
        v = ((v << 1) | bit) & ((1 << d["numbits"]) - 1)
        out.append(bit)
    assert list(sb) == out
