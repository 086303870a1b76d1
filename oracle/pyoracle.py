"""ctypes binding of oracle/liboracle.so (and oracle/_ref/libhfdl_ref.so when present).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from dumphfdl_amd/ (the product path).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Ddc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "pre_decimation", "post_decimation", "taps_length", "taps_min_length", "overlap_length",
        "fft_size", "fft_inv_size", "input_size", "post_input_size", "startbin", "v", "offsetbin", "scrap")] + \
        [(n, C.c_float) for n in ("pre_shift", "post_shift", "nco_sindelta", "nco_cosdelta", "nco_rate")]


class Cf(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class NcoState(C.Structure):
    _fields_ = [("decimation_remain", C.c_int32), ("starting_phase", C.c_float), ("output_size", C.c_int32)]


class Pdu(C.Structure):
    _fields_ = [("freq", C.c_int32), ("mode", C.c_int32), ("len", C.c_int32), ("octets", C.c_uint8 * 960),
                ("freq_err_hz", C.c_float), ("rssi_db", C.c_float), ("noise_floor_db", C.c_float),
                ("bit_rate", C.c_int32), ("slot", C.c_char), ("sample_index", C.c_uint64),
                ("train_bits_bad", C.c_int32), ("train_bits_total", C.c_int32)]


class TapsView(C.Structure):
    _fields_ = [("chan_out", C.c_void_p), ("chan_out_n", C.c_int32),
                ("resampled", C.c_void_p), ("resampled_n", C.c_int32),
                ("mf_out", C.c_void_p), ("mf_out_n", C.c_int32),
                ("symbols", C.c_void_p), ("symbols_n", C.c_int32),
                ("agc_level", C.c_void_p)]


class Variant(C.Structure):
    """orc_variant (hfdl_oracle.h): switches for the unpinned readings; all zero / dmin 4.0 = the restatement the parity tests use."""
    _fields_ = [("symsync_reset_both", C.c_int32), ("resamp_kind", C.c_int32), ("kaiser_arg", C.c_int32), ("soft_dmin_init", C.c_float),
                ("lfsr_kind", C.c_int32), ("eqlms_norm", C.c_int32), ("agc_double", C.c_int32), ("design_float", C.c_int32),
                ("perr_kind", C.c_int32), ("dot_order", C.c_int32), ("symsync_bank_floor", C.c_int32),
                ("symsync_dmf_scale", C.c_float), ("symsync_lf_b", C.c_float), ("soft_gamma_scale", C.c_float), ("soft_floor", C.c_int32),
                ("agc_y2_init", C.c_float), ("shared_math", C.c_int32)]


SINK = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Pdu))


_build_kind = "strict"


def select_build(kind):
    """"strict" (default: -ffp-contract=off, no fast-math -- the build every parity check uses) or "fast" (the reference's
    release flags, -O3 -ffast-math: bench.py's cpu_baseline timing only).  Close every Frontend / Channel made with the
    previous build before switching: their handles belong to the library that made them."""
    global _build_kind, _lib
    assert kind in ("strict", "fast")
    if kind != _build_kind:
        _build_kind, _lib = kind, None


def build(force=False):
    name = "liboracle.so" if _build_kind == "strict" else "liboracle_fast.so"
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ("csdr_restated.c", "fec_restated.c", "channel_restated.c", "hfdl_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src"):
        ref = os.path.join(_HERE, "_ref", "libhfdl_ref.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.orc_fastddc_init.argtypes = [C.POINTER(Ddc), C.c_float, C.c_int32, C.c_float]
        L.orc_fastddc_init.restype = C.c_int
        L.orc_transition_bw.restype = C.c_float
        L.orc_transition_bw.argtypes = [C.c_int32, C.c_int32]
        L.orc_compute_fft_decimation_rate.argtypes = [C.c_int32, C.c_int32]
        L.orc_crc16_ccitt.restype = C.c_uint16
        L.orc_crc16_ccitt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint16]
        L.orc_fcs_check.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_pdu_triage.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
        L.orc_lpdu_walk.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_set_fft_threads.argtypes = [C.c_int]
        L.orc_fft_f32_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int]
        L.orc_viterbi27_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_conv27_encode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_fft_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int]
        L.orc_fft_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int]
        L.orc_channelizer_taps.argtypes = [C.POINTER(Ddc), C.c_int32, C.c_float, C.c_void_p, C.c_int]
        L.orc_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
        L.orc_shift_decimate.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(Ddc), NcoState]
        L.orc_shift_decimate.restype = NcoState
        L.orc_fastddc_inv.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Ddc), C.c_void_p, NcoState, C.c_void_p]
        L.orc_fastddc_inv.restype = NcoState
        L.orc_forward_block.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Ddc), C.c_void_p]
        L.orc_firdes_bandpass_c.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float]
        L.orc_decode_user_data.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_decode_user_data.restype = C.c_int32
        L.orc_deinterleave_maps.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_scrambler_bits.argtypes = [C.c_void_p, C.c_int32]
        L.orc_modem_demod_soft.argtypes = [C.c_int, Cf, C.c_void_p]
        L.orc_channel_create.restype = C.c_void_p
        L.orc_channel_create.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int]
        L.orc_channel_destroy.argtypes = [C.c_void_p]
        L.orc_channel_ddc.restype = C.POINTER(Ddc)
        L.orc_channel_ddc.argtypes = [C.c_void_p]
        L.orc_channel_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.orc_channel_summary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_variant_default.argtypes = [C.POINTER(Variant)]
        L.orc_variant_set.argtypes = [C.POINTER(Variant)]
        L.orc_variant_get.argtypes = [C.POINTER(Variant)]
        L.orc_channel_taps.restype = C.c_void_p
        L.orc_channel_taps.argtypes = [C.c_void_p]
        L.orc_channel_process_baseband.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, SINK, C.c_void_p]
        L.orc_channel_process_spectrum.argtypes = [C.c_void_p, C.c_void_p, SINK, C.c_void_p]
        L.orc_channel_taps_view.argtypes = [C.c_void_p, C.POINTER(TapsView)]
        L.orc_resamp_run.restype = C.c_int32
        L.orc_resamp_run.argtypes = [C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
        L.orc_resamp_filter.argtypes = [C.c_float, C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_symsync_filters.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_eq_initial_taps.argtypes = [C.c_void_p]
        L.orc_frontend_create.restype = C.c_void_p
        L.orc_frontend_create.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        L.orc_frontend_create_mt.restype = C.c_void_p
        L.orc_frontend_create_mt.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int]
        L.orc_frontend_destroy.argtypes = [C.c_void_p]
        L.orc_frontend_ddc.restype = C.POINTER(Ddc)
        L.orc_frontend_ddc.argtypes = [C.c_void_p]
        L.orc_frontend_push_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, SINK, C.c_void_p]
        L.orc_frontend_spectrum.restype = C.c_void_p
        L.orc_frontend_spectrum.argtypes = [C.c_void_p]
        L.orc_frontend_channel.restype = C.c_void_p
        L.orc_frontend_channel.argtypes = [C.c_void_p, C.c_int32]
        for fn in ("orc_mode_num_symbols", "orc_mode_coded_bits", "orc_mode_viterbi_bits", "orc_mode_pdu_octets"):
            getattr(L, fn).argtypes = [C.c_int]
    return _lib


def ref():
    """The reference's own viterbi27_port.c / crc.c / libcsdr_gpl.c, compiled unmodified (or None)."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libhfdl_ref.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
        _ref.create_viterbi27.restype = C.c_void_p
        _ref.create_viterbi27.argtypes = [C.c_int]
        _ref.init_viterbi27.argtypes = [C.c_void_p, C.c_int]
        _ref.update_viterbi27_blk.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _ref.chainback_viterbi27.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
        _ref.delete_viterbi27.argtypes = [C.c_void_p]
        _ref.crc16_ccitt.restype = C.c_uint16
        _ref.crc16_ccitt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint16]
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_variant(**kw):
    """Process-global: applies to channels created afterwards (and to decode_user_data / the scrambler at call time).
    set_variant() with no arguments restores the default restatement."""
    v = Variant()
    lib().orc_variant_default(C.byref(v))
    for k, val in kw.items():
        assert hasattr(v, k), k
        setattr(v, k, val)
    lib().orc_variant_set(C.byref(v))


def get_variant():
    v = Variant()
    lib().orc_variant_get(C.byref(v))
    return {n: getattr(v, n) for n, _ in Variant._fields_}


def scrambler_bits(n):
    out = np.zeros(n, np.uint8)
    lib().orc_scrambler_bits(_p(out), n)
    return out


def cf(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


# ---------------------------------------------------------------- convenience wrappers

def fastddc_init(transition_bw, decimation, shift_rate):
    d = Ddc()
    rc = lib().orc_fastddc_init(C.byref(d), transition_bw, decimation, shift_rate)
    assert rc == 0
    return d


def geometry(sample_rate):
    L = lib()
    dec = L.orc_compute_fft_decimation_rate(sample_rate, 5400)
    tbw = L.orc_transition_bw(sample_rate, 250)
    return dec, tbw, fastddc_init(tbw, dec, 0.0)


def fft(x, sign=-1, f64=False):
    n = len(x)
    if f64:
        xi = np.ascontiguousarray(x, dtype=np.complex128)
        out = np.empty(n, np.complex128)
        lib().orc_fft_f64(_p(xi), _p(out), n, sign)
        return out
    xi = cf(x)
    out = np.empty(n, np.complex64)
    lib().orc_fft_f32(_p(xi), _p(out), n, sign)
    return out


def viterbi27(soft, nbits):
    soft = np.ascontiguousarray(soft, dtype=np.uint8)
    out = np.zeros((nbits + 7) // 8, np.uint8)
    lib().orc_viterbi27_decode(_p(soft), nbits, _p(out))
    return out


def ref_viterbi27(soft, nbits):
    R = ref()
    soft = np.ascontiguousarray(soft, dtype=np.uint8)
    out = np.zeros((nbits + 7) // 8, np.uint8)
    v = R.create_viterbi27(nbits)
    R.init_viterbi27(v, 0)
    R.update_viterbi27_blk(v, _p(soft), nbits)
    R.chainback_viterbi27(v, _p(out), nbits, 0)
    R.delete_viterbi27(v)
    return out


def conv_encode(bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    out = np.zeros(2 * len(bits), np.uint8)
    lib().orc_conv27_encode(_p(bits), len(bits), _p(out))
    return out


def crc16(data, init=0xFFFF):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return lib().orc_crc16_ccitt(_p(data), len(data), init)


def pdu_triage(octets):
    a = np.frombuffer(bytes(octets), np.uint8).copy()
    kind, hl = C.c_int(0), C.c_uint32(0)
    st = lib().orc_pdu_triage(_p(a), len(a), C.byref(kind), C.byref(hl))
    return st, kind.value, hl.value


def modem_demod_hard(arity, symbols):
    """liquid's modem_demodulate + get_demodulator_phase_error as restated (fec_restated.c): (symbols uint32, phase errors float32)."""
    L = lib()
    L.orc_modem_demod_hard.restype = C.c_uint32
    L.orc_modem_demod_hard.argtypes = [C.c_int, Cf, C.POINTER(C.c_float)]
    x = np.ascontiguousarray(symbols, dtype=np.complex64)
    sym = np.zeros(len(x), np.uint32)
    err = np.zeros(len(x), np.float32)
    e = C.c_float(0)
    for i, v in enumerate(x):
        sym[i] = L.orc_modem_demod_hard(arity, Cf(float(v.real), float(v.imag)), C.byref(e))
        err[i] = e.value
    return sym, err


def lpdu_walk(octets):
    """(processed, good, bad_fcs, too_short, truncated) of the PDU's LPDU list."""
    a = np.frombuffer(bytes(octets), np.uint8).copy()
    counts = np.zeros(5, np.uint8)
    lib().orc_lpdu_walk(_p(a), len(a), _p(counts))
    return tuple(int(v) for v in counts)


def decode_user_data(mode, symbols, bitmask_lsb=0):
    symbols = cf(symbols)
    out = np.zeros(960, np.uint8)
    n = lib().orc_decode_user_data(mode, _p(symbols), bitmask_lsb, _p(out))
    return out[:n].copy()


def pdu_to_dict(p):
    return dict(freq=p.freq, mode=p.mode, octets=bytes(p.octets[:p.len]), freq_err_hz=p.freq_err_hz,
                rssi_db=p.rssi_db, noise_floor_db=p.noise_floor_db, bit_rate=p.bit_rate,
                slot=p.slot.decode(), sample_index=p.sample_index,
                train_bits_bad=p.train_bits_bad, train_bits_total=p.train_bits_total)


class Channel:
    """orc_channel: one HFDL channel (optionally with its fastddc channelizer)."""

    def __init__(self, sample_rate, centerfreq, frequency, want_channelizer=True):
        L = lib()
        self.dec = L.orc_compute_fft_decimation_rate(sample_rate, 5400)
        self.tbw = L.orc_transition_bw(sample_rate, 250)
        self.h = L.orc_channel_create(sample_rate, self.dec, self.tbw, centerfreq, frequency, int(want_channelizer))
        assert self.h
        self.ddc = L.orc_channel_ddc(self.h).contents
        self.pdus = []
        self._sink = SINK(lambda ctx, p: self.pdus.append(pdu_to_dict(p.contents)))

    def taps_fft(self):
        n = self.ddc.fft_size
        ptr = lib().orc_channel_taps(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (2 * n,)).view(np.complex64).copy()

    def process_baseband(self, x):
        x = cf(x)
        lib().orc_channel_process_baseband(self.h, _p(x), len(x), self._sink, None)

    def process_spectrum(self, spec):
        spec = cf(spec)
        lib().orc_channel_process_spectrum(self.h, _p(spec), self._sink, None)

    def view(self):
        v = TapsView()
        lib().orc_channel_taps_view(self.h, C.byref(v))

        def arr(ptr, n, dt=np.complex64):
            if not ptr or n <= 0:
                return np.zeros(0, dt)
            k = 2 * n if dt == np.complex64 else n
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (k,)).view(dt).copy()
        return dict(chan_out=arr(v.chan_out, v.chan_out_n), resampled=arr(v.resampled, v.resampled_n),
                    mf_out=arr(v.mf_out, v.mf_out_n), symbols=arr(v.symbols, v.symbols_n),
                    agc_level=arr(v.agc_level, v.resampled_n, np.float32))

    def summary(self):
        """hfdl_print_summary's figures for this channel (src/hfdl.c:563-573)."""
        cnt = (C.c_uint32 * 6)()
        corr = (C.c_float * 3)()
        lib().orc_channel_summary(self.h, cnt, corr)
        return dict(a1_found=cnt[0], a2_found=cnt[1], m1_found=cnt[2], m1_not_found=cnt[3], train_bits_bad=cnt[4], train_bits_total=cnt[5],
                    a1_corr_total=corr[0], a2_corr_total=corr[1], m1_corr_total=corr[2])

    def close(self):
        if self.h:
            lib().orc_channel_destroy(self.h)
            self.h = None

    __del__ = close


class Frontend:
    """orc_frontend: forward FFT block + N channels, the reference's main.c wiring."""

    def __init__(self, sample_rate, centerfreq, freqs, nthreads=1):
        fr = np.ascontiguousarray(freqs, dtype=np.int32)
        self.h = lib().orc_frontend_create_mt(sample_rate, centerfreq, _p(fr), len(fr), nthreads)
        assert self.h
        self.nch = len(fr)
        self.ddc = lib().orc_frontend_ddc(self.h).contents
        self.pdus = []
        self._sink = SINK(lambda ctx, p: self.pdus.append(pdu_to_dict(p.contents)))

    def push_block(self, samples, nthreads=1):
        s = cf(samples)
        assert len(s) == self.ddc.input_size
        lib().orc_frontend_push_block(self.h, _p(s), nthreads, self._sink, None)

    def spectrum(self):
        n = self.ddc.fft_size
        ptr = lib().orc_frontend_spectrum(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (2 * n,)).view(np.complex64).copy()

    def channel_counters(self, i):
        cnt = (C.c_uint32 * 4)()
        nf, st = C.c_float(0), C.c_int(0)
        lib().orc_channel_counters(lib().orc_frontend_channel(self.h, i), cnt, C.byref(nf), C.byref(st))
        return dict(a2_found=cnt[0], m1_found=cnt[1], m1_not_found=cnt[2], frames=cnt[3], noise_floor=nf.value, framer_state=st.value)

    def channel_summary(self, i):
        ch = Channel.__new__(Channel)
        ch.h = lib().orc_frontend_channel(self.h, i)
        out = Channel.summary(ch)
        ch.h = None
        return out

    def channel_view(self, i):
        ch = Channel.__new__(Channel)
        ch.h = lib().orc_frontend_channel(self.h, i)
        v = Channel.view(ch)
        ch.h = None
        return v

    def close(self):
        if self.h:
            lib().orc_frontend_destroy(self.h)
            self.h = None

    __del__ = close
