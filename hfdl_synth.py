"""Synthetic HFDL traffic generator (the TX side -- not part of the reference, which is receive-only).

Builds valid HFDL bursts (prekey + preamble + interleaved/FEC-coded/scrambled M-PSK data, SURVEY.md
Appendix B) and mixes any number of channels into one wideband cf32 stream by FFT interpolation, so
that the bench and the parity tests have a seeded, reproducible input of the shape
BASELINE.json's configs name.  numpy only; TEST AND BENCH INFRASTRUCTURE (used by tests/, bench.py, __graft_entry__.smoke()); not part of the product package.
"""
import numpy as np

try:                                   # scipy's pocketfft keeps complex64 in single precision
    from scipy import fft as _fft
except Exception:                      # pragma: no cover
    _fft = np.fft

SYMBOL_RATE = 1800
CARRIER_OFFSET_HZ = 1440               # src/hfdl.c:46
PREKEY_LEN = 448

# mode -> (bits/symbol, data segments, code rate denominator, interleaver column shift): src/hfdl.c:81-138
MODES = [(1, 72, 4, 17), (1, 72, 2, 17), (2, 72, 2, 17), (3, 72, 2, 17),
         (1, 168, 4, 23), (1, 168, 2, 23), (2, 168, 2, 23), (3, 168, 2, 23)]

_A_OCTETS = [0x5B, 0xBC, 0x74, 0x57, 0x03, 0xD9, 0x89, 0x39, 0xF2, 0x08, 0xD5, 0x36, 0x94, 0x2C, 0x32, 0xFE]
_M1_BASE = ("01110110111101000101100" "10111110001000000110011011" "00011100111010111000010011"
            "00000101010110100100101001" "11100100011010100001111111")
_M1_SHIFT = [72, 82, 113, 123, 61, 103, 93, 9]
_T_BITS = [(0x9AF >> (14 - i)) & 1 for i in range(15)]


def seq_A():
    return np.array([(_A_OCTETS[i // 8] >> (7 - i % 8)) & 1 for i in range(127)], np.uint8)


def seq_M1(mode):
    return np.array([int(_M1_BASE[(_M1_SHIFT[mode] + j) % 127]) for j in range(127)], np.uint8)


def mode_sizes(mode):
    arity, segs, rate, _ = MODES[mode]
    nsym = segs * 30
    coded = nsym * arity
    vin = coded // 2 if rate == 4 else coded
    nbits = vin // 2
    return dict(arity=arity, segments=segs, code_rate=rate, nsym=nsym, coded=coded, vin=vin, nbits=nbits,
                octets=(nbits + 7) // 8, max_payload=(nbits - 6) // 8)


def burst_symbols_len(mode):
    return PREKEY_LEN + 2 * 127 + 127 + 15 + 9 * 15 + MODES[mode][1] * 45


def crc16_x25(data):
    crc = 0xFFFF
    for b in bytes(data):
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x8408 if crc & 1 else crc >> 1
    return crc ^ 0xFFFF


def make_spdu(rng):
    """66-octet squitter: bit0 of octet0 = 0, FCS over the first 64 octets (src/spdu.c:12,55-62)."""
    body = bytearray(rng.integers(0, 256, 64, dtype=np.uint8).tobytes())
    body[0] &= 0xFE
    fcs = crc16_x25(body)
    return bytes(body) + bytes([fcs & 0xFF, fcs >> 8])


def make_mpdu(rng, total_len):
    """Downlink MPDU: bit0=1, bit1=1, lpdu_cnt in bits 2..5; header 6+lpdu_cnt octets then FCS (src/mpdu.c:56-89)."""
    lpdu_cnt = int(rng.integers(0, 8))
    hdr_len = 6 + lpdu_cnt
    total_len = max(total_len, hdr_len + 2)
    buf = bytearray(rng.integers(0, 256, total_len, dtype=np.uint8).tobytes())
    buf[0] = (buf[0] & 0xC0) | (lpdu_cnt << 2) | 0x3
    fcs = crc16_x25(buf[:hdr_len])
    buf[hdr_len] = fcs & 0xFF
    buf[hdr_len + 1] = fcs >> 8
    return bytes(buf)


def make_mpdu_with_lpdus(rng, total_len, uplink=False, spoil=()):
    """An MPDU whose LPDU list is real (src/mpdu.c:56-158): `total_len` octets, size octets = LPDU length - 1, every LPDU ends in
    its own FCS (src/lpdu.c:143-144).  `spoil`: indices of LPDUs whose FCS is broken.  Returns (octets, lpdu_count)."""
    def lpdu(n, bad):
        body = bytearray(rng.integers(0, 256, n - 2, dtype=np.uint8).tobytes())
        fcs = crc16_x25(body) ^ (0x0100 if bad else 0)
        return bytes(body) + bytes([fcs & 0xFF, fcs >> 8])

    if not uplink:
        cnt = int(rng.integers(1, 8))
        hdr_len = 6 + cnt
        room = total_len - hdr_len - 2
        lens = [int(v) for v in rng.integers(3, min(257, max(4, room // cnt)), cnt)]
        hdr = bytearray(rng.integers(0, 256, hdr_len, dtype=np.uint8).tobytes())
        hdr[0] = (hdr[0] & 0xC0) | (cnt << 2) | 0x3
        for j, n in enumerate(lens):
            hdr[6 + j] = n - 1
        groups = [lens]
    else:
        ac = int(rng.integers(1, 4))
        per_ac = [int(rng.integers(1, 4)) for _ in range(ac)]
        cnt = sum(per_ac)
        room = total_len - (2 + 2 * ac + cnt) - 2
        groups = [[int(v) for v in rng.integers(3, min(257, max(4, room // cnt)), k)] for k in per_ac]
        hdr = bytearray([0x01 | ((ac - 1) << 4), int(rng.integers(0, 128))])
        for g in groups:
            hdr += bytes([int(rng.integers(0, 256)), (len(g) << 4) | int(rng.integers(0, 16))]) + bytes(n - 1 for n in g)
    fcs = crc16_x25(hdr)
    out = bytes(hdr) + bytes([fcs & 0xFF, fcs >> 8])
    k = 0
    for g in groups:
        for n in g:
            out += lpdu(n, k in spoil)
            k += 1
    assert len(out) <= total_len
    return out + bytes(total_len - len(out)), cnt


def make_pdu(rng, mode):
    sz = mode_sizes(mode)
    if sz["max_payload"] == 66 and rng.random() < 0.5:
        return make_spdu(rng)
    return make_mpdu(rng, sz["max_payload"])


def conv_encode(bits):
    """K=7 r=1/2, polynomials 0x6d / 0x4f, newest bit in the LSB of the register."""
    b = np.concatenate([np.zeros(6, np.uint8), bits.astype(np.uint8)])
    n = len(bits)
    sl = lambda d: b[6 - d:6 - d + n]
    c0 = sl(0) ^ sl(2) ^ sl(3) ^ sl(5) ^ sl(6)
    c1 = sl(0) ^ sl(1) ^ sl(2) ^ sl(3) ^ sl(6)
    out = np.empty(2 * n, np.uint8)
    out[0::2] = c0
    out[1::2] = c1
    return out


def scrambler_bits(n):
    out = np.empty(n, np.uint8)
    v = 0
    for i in range(n):
        if i % 120 == 0:
            v = 0x4D4B
        b = bin(v & 0x4001).count("1") & 1
        v = ((v << 1) | b) & 0x7FFF
        out[i] = b
    return out


_SCR = scrambler_bits(120)


def interleave_maps(mode):
    arity, segs, _, shift = MODES[mode]
    total = segs * 30 * arity
    cols = total // 40
    k = np.arange(total)
    push = (k % 40) * cols + ((k // 40 - shift * k) % cols)
    pop = ((9 * k) % 40) * cols + (k // 40)
    return push, pop


def _gray_decode(g):
    b = g
    s = g >> 1
    while s:
        b ^= s
        s >>= 1
    return b


def encode_data_symbols(octets, mode):
    """PDU octets -> the scrambled data symbols of one burst (inverse of src/hfdl.c:993-1056)."""
    sz = mode_sizes(mode)
    assert len(octets) <= sz["max_payload"], "payload does not fit this mode"
    bits = np.zeros(sz["nbits"], np.uint8)
    ob = np.unpackbits(np.frombuffer(bytes(octets), np.uint8), bitorder="little")
    bits[:len(ob)] = ob
    coded = conv_encode(bits)
    if sz["code_rate"] == 4:
        coded = np.repeat(coded, 2)
    push, pop = interleave_maps(mode)
    inv_pop = np.empty(len(pop), np.int64)
    inv_pop[pop] = np.arange(len(pop))
    tx = coded[inv_pop[push]]
    a = sz["arity"]
    grp = tx.reshape(-1, a)
    sym = np.zeros(len(grp), np.int64)
    for j in range(a):
        sym = (sym << 1) | grp[:, j]
    if a == 1:
        pts = 1.0 - 2.0 * sym
    else:
        M = 1 << a
        lin = np.array([_gray_decode(int(s)) for s in range(M)])
        pts = np.exp(2j * np.pi * lin[sym] / M)
    scr = np.tile(_SCR, sz["nsym"] // 120)
    return (pts * (1.0 - 2.0 * scr)).astype(np.complex64)


def burst_symbols(octets, mode):
    """All symbols of one burst at 1 sample/symbol."""
    bp = lambda bits: (1.0 - 2.0 * np.asarray(bits, float)).astype(np.complex64)
    A = bp(seq_A())
    m1 = seq_M1(mode)
    T = bp(_T_BITS)
    parts = [np.ones(PREKEY_LEN, np.complex64), A, A, bp(m1), bp(m1[:15])] + [T] * 9
    data = encode_data_symbols(octets, mode)
    for s in range(MODES[mode][1]):
        parts.append(data[30 * s:30 * s + 30])
        parts.append(T)
    return np.concatenate(parts)


def rrc_pulse(t, beta=0.2):
    """Root-raised-cosine pulse, t in symbols, peak-normalised so that MF(RRC*RRC) ~ 1 at the sampling instant."""
    t = np.asarray(t, np.float64)
    out = np.empty_like(t)
    z = np.abs(t) < 1e-9
    s = np.abs(np.abs(4 * beta * t) - 1) < 1e-9
    g = ~(z | s)
    out[z] = 1 - beta + 4 * beta / np.pi
    out[s] = beta / np.sqrt(2) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * beta)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * beta)))
    tg = t[g]
    out[g] = (np.sin(np.pi * tg * (1 - beta)) + 4 * beta * tg * np.cos(np.pi * tg * (1 + beta))) / (np.pi * tg * (1 - (4 * beta * tg) ** 2))
    return out


# the receiver's 19-tap matched filter at 3 samples per symbol (protocol pulse-shape table, src/hfdl.c:147-154)
_MF_TABLE = np.array([-0.0170974647427123, 0.01148231492068473, 0.03138375667422348, 0.009454398851680437, -0.04161644170893816,
                      -0.06451564801420356, -0.005495792933327306, 0.1316404671361545, 0.2759693160697777, 0.3375901874933208])
_MF_TABLE = np.concatenate([_MF_TABLE, _MF_TABLE[-2::-1]])

# transmit pulse of every burst shaped from here on: ("rrc", beta) -- a textbook root-raised cosine, the default (0.2) -- or
# ("mf_table",): the receiver's own table, band-limited interpolation between its 3-per-symbol points, peak-normalised.
# A module-level switch (set_tx_pulse) so that the sensitivity study can vary the transmitter without threading it through every caller.
_TX_PULSE = ("rrc", 0.2)


def set_tx_pulse(kind="rrc", beta=0.2):
    global _TX_PULSE
    _TX_PULSE = (kind, beta)


def tx_pulse(t):
    """The transmit pulse at t (in symbols)."""
    if _TX_PULSE[0] == "mf_table":
        t = np.asarray(t, np.float64)
        k = np.arange(19) - 9
        return (np.sinc(3.0 * t[..., None] - k) * _MF_TABLE).sum(-1) / _MF_TABLE[9]
    return rrc_pulse(t, _TX_PULSE[1])


def shape_burst(symbols, rate, t0, nsamples, span=6):
    """Pulse-shape `symbols` (1800 baud) onto a grid of `nsamples` samples at `rate` Hz, first symbol at t0 seconds."""
    out = np.zeros(nsamples, np.complex128)
    n = np.arange(nsamples)
    ts = (n / rate - t0) * SYMBOL_RATE           # time in symbols
    lo = int(np.searchsorted(ts, -span))
    hi = int(np.searchsorted(ts, len(symbols) + span))
    if hi <= lo:
        return out
    tt = ts[lo:hi]
    k0 = np.floor(tt).astype(np.int64)
    acc = np.zeros(hi - lo, np.complex128)
    for d in range(-span + 1, span + 1):
        k = k0 + d
        ok = (k >= 0) & (k < len(symbols))
        a = np.where(ok, symbols[np.clip(k, 0, len(symbols) - 1)], 0)
        acc += a * tx_pulse(tt - k)
    out[lo:hi] = acc
    return out


def baseband_rate(fs):
    """Per-channel synthesis rate fs/P: P = largest power of two keeping >= 7 kHz."""
    P = 1
    while fs / (2 * P) >= 7000:
        P *= 2
    return P


def synth_wideband(fs, centerfreq, nsamples, bursts, noise_sigma=0.0, seed=0, nlo=4096):
    """Mix bursts into a wideband complex64 stream of `nsamples` samples at `fs`.

    bursts: iterable of dicts {freq (Hz, channel frequency as passed to hfdl_channel_create),
            mode, octets, t0 (s, start of prekey), amp, cfo (Hz, optional carrier error)}.
    Carrier of a channel = freq + 1440 Hz (src/hfdl.c:46,476).
    """
    rng = np.random.default_rng(seed)
    P = baseband_rate(fs)
    nbig = nlo * P
    hop_lo = nlo // 2
    nseg = int(np.ceil(nsamples / (hop_lo * P))) + 1
    llo = nseg * hop_lo + nlo
    fs_lo = fs / P
    by_freq = {}
    for b in bursts:
        by_freq.setdefault(int(b["freq"]), []).append(b)
    # frequency-domain interpolation window: flat to 0.3 fs_lo, raised-cosine to 0 at 0.45 fs_lo
    f = np.abs(np.fft.fftfreq(nlo))
    win = np.where(f < 0.3, 1.0, np.where(f < 0.45, 0.5 * (1 + np.cos(np.pi * (f - 0.3) / 0.15)), 0.0)).astype(np.float32)
    chans = []
    for freq, bl in by_freq.items():
        f_off = freq + CARRIER_OFFSET_HZ - centerfreq
        kc = int(np.round(f_off / (fs / nbig)))
        resid = f_off - kc * fs / nbig
        lo = np.zeros(llo, np.complex128)
        for b in bl:
            sym = burst_symbols(b["octets"], b["mode"])
            # low-rate sample m corresponds to absolute time (m - nlo/4)/fs_lo
            lo += b.get("amp", 0.1) * shape_burst(sym, fs_lo, b["t0"] + (nlo // 4) / fs_lo, llo) * \
                np.exp(2j * np.pi * b.get("cfo", 0.0) * (np.arange(llo) - nlo // 4) / fs_lo)
        lo *= np.exp(2j * np.pi * resid * (np.arange(llo) - nlo // 4) / fs_lo)
        chans.append((kc, lo.astype(np.complex64)))
    out = np.empty(nseg * hop_lo * P, np.complex64)
    big = np.zeros(nbig, np.complex64)
    for h in range(nseg):
        big[:] = 0
        for kc, lo in chans:
            seg = lo[h * hop_lo:h * hop_lo + nlo]
            if not np.any(seg):
                continue
            X = _fft.fft(seg) * win
            X = X * np.complex64(np.exp(1j * np.pi * kc * (h - 0.5)))
            idx = (np.fft.fftfreq(nlo, 1.0 / nlo).astype(np.int64) + kc) % nbig
            big[idx] += X.astype(np.complex64)
        seg_t = _fft.ifft(big) * np.float32(P)
        out[h * hop_lo * P:(h + 1) * hop_lo * P] = seg_t[nbig // 4:nbig // 4 + nbig // 2]
    out = out[:nsamples]
    if noise_sigma > 0:
        out = out + (noise_sigma * (rng.standard_normal(nsamples, dtype=np.float32)
                                    + 1j * rng.standard_normal(nsamples, dtype=np.float32))).astype(np.complex64)
    return np.ascontiguousarray(out, dtype=np.complex64)


def synth_channel_baseband(rate, nsamples, bursts, noise_sigma=0.0, seed=0):
    """Single-channel complex baseband at `rate` Hz (carrier at DC): feeds the demodulator without a channelizer."""
    rng = np.random.default_rng(seed)
    x = np.zeros(nsamples, np.complex128)
    n = np.arange(nsamples)
    for b in bursts:
        sym = burst_symbols(b["octets"], b["mode"])
        x += b.get("amp", 0.1) * shape_burst(sym, rate, b["t0"], nsamples) * np.exp(2j * np.pi * b.get("cfo", 0.0) * n / rate)
    if noise_sigma > 0:
        x = x + noise_sigma * (rng.standard_normal(nsamples) + 1j * rng.standard_normal(nsamples))
    return x.astype(np.complex64)


def plan_traffic(freqs, duration_s, seed, modes=None, gap_s=0.3, amp=(0.05, 0.2), cfo_hz=20.0, dense=True):
    """Schedule bursts on every channel: back-to-back (dense) or one burst per channel (sparse)."""
    rng = np.random.default_rng(seed)
    bursts = []
    for f in freqs:
        t = float(rng.uniform(0.05, 0.6))
        i = int(rng.integers(0, 8))
        while True:
            mode = int(modes[i % len(modes)]) if modes is not None else i % 8
            dur = burst_symbols_len(mode) / SYMBOL_RATE
            if t + dur > duration_s - 0.05:
                break
            bursts.append(dict(freq=int(f), mode=mode, octets=make_pdu(rng, mode), t0=t,
                               amp=float(rng.uniform(*amp)), cfo=float(rng.uniform(-cfo_hz, cfo_hz))))
            t += dur + gap_s
            i += 1
            if not dense:
                break
    return bursts
