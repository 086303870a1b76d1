"""CPU tests: the oracle against the committed golden vectors (made from the reference's own compiled sources),
against the live oracle/_ref library when it exists (container only), and against float64 numpy math."""
import ctypes as C
import json
import os
import numpy as np
import pytest

import hfdl_synth as synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_rms(a, b):
    a = np.asarray(a, np.complex128); b = np.asarray(b, np.complex128)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / max(np.mean(np.abs(b) ** 2), 1e-300)))


# ---------------------------------------------------------------- bit-exact pieces pinned to the reference

def test_viterbi_golden_vectors(oracle):
    g = np.load(os.path.join(GOLD, "viterbi_ref.npz"))
    n = 0
    for mode in range(8):
        nbits = synth.mode_sizes(mode)["nbits"]
        for c in range(3):
            k = "m%d_c%d_soft" % (mode, c)
            if k not in g:
                continue
            got = oracle.viterbi27(g[k], nbits)
            assert bytes(got) == bytes(g["m%d_c%d_out" % (mode, c)]), (mode, c)
            n += 1
        # clean and noisy cases decode to the transmitted bits (tail included: last 6 bits are zero)
        assert bytes(oracle.viterbi27(g["m%d_c0_soft" % mode], nbits)) == bytes(g["m%d_bits" % mode])
        assert bytes(oracle.viterbi27(g["m%d_c1_soft" % mode], nbits)) == bytes(g["m%d_bits" % mode])
    assert n == 18


def test_viterbi_tail_quirk(oracle):
    """dumphfdl runs update for nbits steps and chains back nbits: the last 6 decoded bits are always 0."""
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2, 540).astype(np.uint8)
    bits[-6:] = 1
    soft = synth.conv_encode(bits) * 255
    out = np.unpackbits(oracle.viterbi27(soft, 540))[:540]
    assert (out[:534] == bits[:534]).all() and (out[534:] == 0).all()


def test_viterbi_live_reference(oracle):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built (reference sources only exist in the build container)")
    rng = np.random.default_rng(2)
    for nbits in (540, 1260, 3240, 7560):
        for _ in range(3):
            soft = rng.integers(0, 256, 2 * nbits).astype(np.uint8)
            assert bytes(oracle.viterbi27(soft, nbits)) == bytes(oracle.ref_viterbi27(soft, nbits))


def test_crc_golden_vectors(oracle):
    vec = json.load(open(os.path.join(GOLD, "crc_ref.json")))
    assert len(vec) == 24
    for v in vec:
        d = np.frombuffer(bytes.fromhex(v["data"]), np.uint8)
        assert oracle.crc16(d if len(d) else np.zeros(0, np.uint8), v["init"]) == v["crc"]
    assert oracle.crc16(np.frombuffer(b"123456789", np.uint8), 0xFFFF) ^ 0xFFFF == 0x906E     # CRC-16/X-25
    for v in vec:
        if v["init"] == 0xFFFF:
            assert synth.crc16_x25(bytes.fromhex(v["data"])) == v["crc"] ^ 0xFFFF


def test_fcs_check(oracle):
    rng = np.random.default_rng(3)
    spdu = np.frombuffer(synth.make_spdu(rng), np.uint8).copy()
    assert oracle.lib().orc_fcs_check(spdu.ctypes.data, 64) == 1
    spdu[10] ^= 1
    assert oracle.lib().orc_fcs_check(spdu.ctypes.data, 64) == 0


def test_nco_golden_vectors(oracle):
    g = np.load(os.path.join(GOLD, "nco_ref.npz"))
    L = oracle.lib()
    for case in range(4):
        rate, dec, n, sd, cd, r2 = g["c%d_params" % case]
        d = oracle.Ddc()
        d.post_decimation = int(dec)
        d.nco_sindelta, d.nco_cosdelta, d.nco_rate = float(sd), float(cd), float(r2)
        st = oracle.NcoState(0, 0.0, 0)
        ys = []
        for blk in range(3):
            x = np.ascontiguousarray(g["c%d_x" % case][blk])
            y = np.zeros(int(n), np.complex64)
            st = L.orc_shift_decimate(x.ctypes.data, y.ctypes.data, int(n), C.byref(d), st)
            ys.append(y[:st.output_size])
            ref = g["c%d_state" % case][blk]
            assert st.decimation_remain == int(ref[0]) and st.output_size == int(ref[2])
            assert np.float32(st.starting_phase) == np.float32(ref[1])
        got = np.concatenate(ys)
        want = g["c%d_y" % case]
        # the reference is built -O2 without -ffast-math here: identical fp32 recurrence, bit for bit
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), case


# ---------------------------------------------------------------- geometry

@pytest.mark.parametrize("fs,exp", [
    (250000, dict(dec=32, pre=16, post=2, taps=4097, n=32768, inp=28672, m=2048, scrap=256, pis=1792)),
    (8000000, dict(dec=1024, pre=512, post=2, taps=131073, n=1048576, inp=917504, m=2048, scrap=256, pis=1792)),
    (40000000, dict(dec=4096, pre=2048, post=2, taps=1048577, n=8388608, inp=7340032, m=4096, scrap=512, pis=3584)),
])
def test_geometry_table(oracle, fs, exp):
    dec, tbw, g = oracle.geometry(fs)
    assert dec == exp["dec"]
    assert (g.pre_decimation, g.post_decimation, g.taps_length, g.fft_size, g.input_size, g.fft_inv_size, g.scrap,
            g.post_input_size) == (exp["pre"], exp["post"], exp["taps"], exp["n"], exp["inp"], exp["m"], exp["scrap"], exp["pis"])
    assert g.v == 8 and g.overlap_length == g.taps_length - 1


def test_bin_shift_is_multiple_of_v(oracle):
    for fs, off in [(250000, 37000), (8000000, -3100000), (40000000, 19000000)]:
        dec, tbw, _ = oracle.geometry(fs)
        shift = np.float32(-(off + 1440)) / np.float32(fs)
        d = oracle.fastddc_init(tbw, dec, float(shift))
        assert d.offsetbin % d.v == 0
        assert abs(d.offsetbin / d.fft_size - (off + 1440) / fs) < d.v / d.fft_size
        # residual shift left to the NCO is below one bin step of the inverse FFT
        assert abs(d.post_shift) <= d.pre_decimation * d.v / d.fft_size


# ---------------------------------------------------------------- float stages against float64 math

@pytest.mark.parametrize("n", [2, 8, 64, 2048, 4096, 32768])
def test_fft_vs_numpy(oracle, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    assert rel_rms(oracle.fft(x, -1), np.fft.fft(x.astype(np.complex128))) < 1e-6
    assert rel_rms(oracle.fft(x, +1), np.fft.ifft(x.astype(np.complex128)) * n) < 1e-6
    assert rel_rms(oracle.fft(x, -1, f64=True), np.fft.fft(x.astype(np.complex128))) < 1e-13


def test_channelizer_is_a_ddc(oracle):
    """The overlap-and-scrap channelizer equals direct-form mix -> FIR -> decimate in float64 (unity gain, exact tone placement)."""
    fs, cf, freq = 250000, 10_000_000, 10_037_000
    ch = oracle.Channel(fs, cf, freq)
    fe = oracle.Frontend(fs, cf, [freq])
    g = fe.ddc
    n = g.input_size
    nblk = 4
    t = np.arange(nblk * n)
    f0 = (freq + 1440 - cf) / fs
    x = (0.3 * np.exp(2j * np.pi * (f0 + 400 / fs) * t) + 0.2 * np.exp(2j * np.pi * (f0 - 700 / fs) * t + 1j)
         + 0.5 * np.exp(2j * np.pi * (f0 + 30000 / fs) * t)).astype(np.complex64)     # last tone is out of band
    outs = []
    for b in range(nblk):
        fe.push_block(x[b * n:(b + 1) * n])
        outs.append(fe.channel_view(0)["chan_out"])
    y = np.concatenate(outs)
    rate = fs / (g.pre_decimation * g.post_decimation)
    # direct form: time-domain taps (same design), convolve, take every 32nd sample, mix down
    taps = np.zeros(g.taps_length, np.complex64)
    dec = oracle.lib().orc_compute_fft_decimation_rate(fs, 5400)
    shift = np.float32(cf - (freq + 1440)) / np.float32(fs)
    hb = np.float32(0.5) / np.float32(dec)
    oracle.lib().orc_firdes_bandpass_c(taps.ctypes.data, g.taps_length, float(-shift - hb), float(-shift + hb))
    full = np.convolve(x.astype(np.complex128), taps.astype(np.complex128))[:len(x)]
    # block k output j corresponds to input sample k*n - overlap + pre*(scrap + 2j) (first block: zeros history)
    idx = (np.arange(len(y)) * g.pre_decimation * g.post_decimation) + g.pre_decimation * g.scrap - g.overlap_length
    want = full[idx[idx >= 0]] * np.exp(-2j * np.pi * f0 * idx[idx >= 0])
    got = y[idx >= 0]
    # compare magnitudes of the two in-band tones via projection (phase reference of the NCO is arbitrary but constant)
    k = 2000           # skip the start-up transient (the first block sees a zero history)
    tt = np.arange(len(got))[k:] / rate
    for df, amp in ((400, 0.3), (-700, 0.2)):
        pg = np.abs(np.vdot(np.exp(2j * np.pi * df * tt), got[k:])) / len(tt)
        pw = np.abs(np.vdot(np.exp(2j * np.pi * df * tt), want[k:])) / len(tt)
        assert abs(pg - amp) < 2e-3 and abs(pg - pw) < 1e-3, (df, pg, pw)
    # rotate by the constant phase offset and compare sample by sample
    rot = np.vdot(want[k:], got[k:]) / np.vdot(want[k:], want[k:])
    assert abs(abs(rot) - 1) < 1e-3
    assert rel_rms(got[k:], want[k:] * rot) < 2e-3
    # out-of-band tone suppressed by > 50 dB
    assert np.sqrt(np.mean(np.abs(got[k:]) ** 2)) < 0.4


def test_taps_f32_vs_f64_fft(oracle):
    dec, tbw, _ = oracle.geometry(250000)
    d = oracle.fastddc_init(tbw, dec, 0.123)
    a = np.zeros(d.fft_size, np.complex64); b = np.zeros(d.fft_size, np.complex64)
    oracle.lib().orc_channelizer_taps(C.byref(d), dec, 0.123, a.ctypes.data, 0)
    oracle.lib().orc_channelizer_taps(C.byref(d), dec, 0.123, b.ctypes.data, 1)
    assert rel_rms(a, b) < 1e-6
    # unity pass-band gain at the channel centre
    centre = d.fft_size // 2 + int(round(-0.123 * d.fft_size))
    assert abs(abs(b[centre]) - 1.0) < 1e-3


# ---------------------------------------------------------------- frame layer

def test_deinterleaver_closed_form(oracle):
    for mode in range(8):
        n = oracle.lib().orc_mode_coded_bits(mode)
        push = np.zeros(n, np.int32); pop = np.zeros(n, np.int32)
        oracle.lib().orc_deinterleave_maps(mode, push.ctypes.data, pop.ctypes.data)
        sp, so = synth.interleave_maps(mode)
        assert np.array_equal(push, sp) and np.array_equal(pop, so)
        assert sorted(push) == list(range(n)) and sorted(pop) == list(range(n))     # both are permutations


def test_scrambler_period_and_balance(oracle):
    b = np.zeros(240, np.uint8)
    oracle.lib().orc_scrambler_bits(b.ctypes.data, 240)
    assert np.array_equal(b[:120], b[120:])
    assert np.array_equal(b[:120], synth.scrambler_bits(120))
    # first 15 outputs of x^15+x+1 from fill 110100101011001 (ARINC 635 scrambler)
    assert 40 < b[:120].sum() < 80


@pytest.mark.parametrize("mode", range(8))
def test_decode_user_data_round_trip(oracle, mode):
    rng = np.random.default_rng(mode)
    pdu = synth.make_pdu(rng, mode)
    sym = synth.encode_data_symbols(pdu, mode)
    sz = synth.mode_sizes(mode)
    assert len(sym) == sz["nsym"] == oracle.lib().orc_mode_num_symbols(mode)
    for mask in (0, 1):
        s = sym * (1 - 2 * mask) + 0.15 * (rng.standard_normal(len(sym)) + 1j * rng.standard_normal(len(sym)))
        out = oracle.decode_user_data(mode, s.astype(np.complex64), mask)
        assert len(out) == sz["octets"] == oracle.lib().orc_mode_pdu_octets(mode)
        assert bytes(out[:len(pdu)]) == pdu and not out[len(pdu):].any()


def test_pdu_sizes():
    assert [synth.mode_sizes(m)["octets"] for m in range(8)] == [68, 135, 270, 405, 158, 315, 630, 945]
    assert [synth.mode_sizes(m)["nbits"] for m in range(8)] == [540, 1080, 2160, 3240, 1260, 2520, 5040, 7560]


def test_soft_demod_properties(oracle):
    soft = np.zeros(3, np.uint8)
    L = oracle.lib()
    f2 = oracle.Cf
    L.orc_modem_demod_soft(1, f2(1.0, 0.0), soft.ctypes.data); assert soft[0] == 0
    L.orc_modem_demod_soft(1, f2(-1.0, 0.0), soft.ctypes.data); assert soft[0] == 255
    L.orc_modem_demod_soft(1, f2(0.0, 0.3), soft.ctypes.data); assert soft[0] == 127
    for arity in (2, 3):
        M = 1 << arity
        for sym in range(M):
            lin = sym
            g = sym ^ (sym >> 1)            # gray_encode(lin) sits at phase index lin
            p = np.exp(2j * np.pi * lin / M)
            L.orc_modem_demod_soft(arity, f2(p.real, p.imag), soft.ctypes.data)
            bits = [(g >> (arity - 1 - k)) & 1 for k in range(arity)]
            assert [int(s > 127) for s in soft[:arity]] == bits


# ---------------------------------------------------------------- whole receive chain on synthetic traffic

def test_oracle_decodes_all_modes_baseband(oracle):
    rng = np.random.default_rng(5)
    ch = oracle.Channel(250000, 10_000_000, 10_030_000, want_channelizer=False)
    rate = 7812.5
    t, bursts = 0.2, []
    for mode in range(8):
        bursts.append(dict(mode=mode, octets=synth.make_pdu(rng, mode), t0=t, amp=0.1, cfo=float(rng.uniform(-15, 15))))
        t += synth.burst_symbols_len(mode) / 1800 + 0.25
    x = synth.synth_channel_baseband(rate, int((t + 0.3) * rate), bursts, noise_sigma=0.006, seed=9)
    for i in range(0, len(x), 896):
        ch.process_baseband(x[i:i + 896])
    assert [p["mode"] for p in ch.pdus] == list(range(8))
    for p, b in zip(ch.pdus, bursts):
        assert p["octets"][:len(b["octets"])] == b["octets"]
        assert p["train_bits_bad"] <= 0.01 * p["train_bits_total"]
        assert abs(p["rssi_db"] - (-20.0)) < 0.5
        assert abs(p["freq_err_hz"] - b["cfo"] / 2) < 1.0       # dphi is per half-symbol step: the reference reports cfo/2
        assert p["slot"] == ("S" if b["mode"] < 4 else "D")
        assert p["bit_rate"] == [300, 600, 1200, 1800][b["mode"] % 4]


def test_oracle_frontend_end_to_end(oracle):
    fs, cf = 250000, 10_000_000
    freqs = [9_930_000, 10_037_000, 10_081_500]
    bursts = synth.plan_traffic(freqs, 6.0, seed=3, dense=True)
    x = synth.synth_wideband(fs, cf, int(6.0 * fs), bursts, noise_sigma=0.01, seed=1)
    fe = oracle.Frontend(fs, cf, freqs)
    n = fe.ddc.input_size
    for b in range(len(x) // n):
        fe.push_block(x[b * n:(b + 1) * n], nthreads=3)
    assert len(fe.pdus) == len(bursts)
    for p in fe.pdus:
        m = [b for b in bursts if b["freq"] == p["freq"] and p["octets"][:len(b["octets"])] == b["octets"]]
        assert len(m) == 1
        assert abs(p["rssi_db"] - 20 * np.log10(m[0]["amp"])) < 0.3      # unity gain through channelizer + resampler
    # polarity-inverted burst still decodes (bitmask path)
    inv = [dict(next(b for b in bursts if b["freq"] == f), t0=0.2) for f in freqs[:2]]
    inv = [dict(b, amp=-b["amp"]) for b in inv]
    x2 = synth.synth_wideband(fs, cf, int(3.0 * fs), inv, noise_sigma=0.01, seed=2)
    fe2 = oracle.Frontend(fs, cf, freqs)
    for b in range(len(x2) // n):
        fe2.push_block(x2[b * n:(b + 1) * n])
    assert len(fe2.pdus) == 2


def test_threaded_forward_fft_is_the_same_transform(oracle):
    """orc_fft_f32_mt (six-step split on pthreads; used by bench.py's cpu_baseline only, where dumphfdl runs FFTW on 4 threads)
    computes the DFT of orc_fft_f32 -- another order of rounding, same result to fp32 accuracy -- and the parity oracle stays on
    the single-thread transform unless told otherwise."""
    import ctypes as C
    L = oracle.lib()
    assert L.orc_get_fft_threads() == 1
    rng = np.random.default_rng(5)
    for n in (4096, 1 << 15, 1 << 18):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128))
        one = np.zeros(n, np.complex64)
        L.orc_fft_f32(x.ctypes.data_as(C.c_void_p), one.ctypes.data_as(C.c_void_p), n, -1)
        for threads in (2, 4, 7):
            y = np.zeros(n, np.complex64)
            L.orc_fft_f32_mt(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), n, -1, threads)
            assert np.linalg.norm(y - ref) / np.linalg.norm(ref) < 5e-7
            assert np.linalg.norm(y - one) / np.linalg.norm(ref) < 5e-7
        back = np.zeros(n, np.complex64)
        L.orc_fft_f32_mt(one.ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p), n, +1, 4)
        assert np.linalg.norm(back / n - x) / np.linalg.norm(x) < 1e-6
