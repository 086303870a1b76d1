"""ctypes mirror of include/hfdl_gpu.h (libhfdl_gpu.so).  No compute happens in Python."""
import ctypes as C
import os
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PDU_MAX_OCTETS = 960

TAP_SPECTRUM, TAP_FILTER, TAP_CHAN_OUT, TAP_RESAMPLED, TAP_MF_OUT, TAP_SYMBOLS, TAP_AGC_LEVEL, TAP_PHASE_CYCLES, TAP_NCO_PHASORS = range(1, 10)


SFMT_CF32, SFMT_CS16, SFMT_CU8 = 0, 1, 2
FCS_GOOD, FCS_BAD, FCS_TOO_SHORT = 0, 1, 2
KIND_SPDU, KIND_MPDU_DOWNLINK, KIND_MPDU_UPLINK = 0, 1, 2


class GpuError(RuntimeError):
    pass


class Geometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "sample_rate", "decimation", "pre_decimation", "post_decimation", "taps_length", "overlap_length",
        "fft_size", "fft_inv_size", "input_size", "post_input_size", "scrap", "outputs_per_block",
        "channels", "fold_slices")] + [("transition_bw", C.c_float), ("resamp_rate", C.c_float), ("max_outputs_per_block", C.c_int32), ("demod_batch", C.c_int32), ("fold_batch", C.c_int32),
                                       ("prefetch_depth", C.c_int32), ("fold_rows", C.c_int32)]


class Pdu(C.Structure):
    _fields_ = [("channel", C.c_int32), ("freq", C.c_int32), ("mode", C.c_int32), ("bit_rate", C.c_int32),
                ("len", C.c_int32), ("freq_err_hz", C.c_float), ("rssi_db", C.c_float), ("noise_floor_db", C.c_float),
                ("slot", C.c_char), ("fcs_status", C.c_uint8), ("pdu_kind", C.c_uint8), ("hdr_len", C.c_uint16),
                ("sample_index", C.c_uint64),
                ("train_bits_bad", C.c_int32), ("train_bits_total", C.c_int32),
                ("lpdus_processed", C.c_uint8), ("lpdus_good", C.c_uint8), ("lpdus_bad_fcs", C.c_uint8), ("lpdus_too_short", C.c_uint8),
                ("lpdus_truncated", C.c_uint8), ("lpdu_pad", C.c_uint8 * 3),
                ("octets", C.c_uint8 * PDU_MAX_OCTETS)]


class ChannelStats(C.Structure):
    _fields_ = [("freq", C.c_int32), ("a2_found", C.c_uint32), ("m1_found", C.c_uint32), ("m1_not_found", C.c_uint32),
                ("frames", C.c_uint32), ("noise_floor_db", C.c_float), ("agc_level", C.c_float), ("costas_dphi", C.c_float),
                ("framer_state", C.c_int32), ("sample_cnt", C.c_uint64), ("symbol_cnt", C.c_uint64),
                ("a1_found", C.c_uint32), ("a1_corr_avg", C.c_float), ("a2_corr_avg", C.c_float), ("m1_corr_avg", C.c_float),
                ("train_bits_bad", C.c_uint32), ("train_bits_total", C.c_uint32)]


class FrontendCounters(C.Structure):
    _fields_ = [("blocks", C.c_uint64), ("pdus_taken", C.c_uint32), ("pdus_dropped", C.c_uint32), ("pdu_ring_capacity", C.c_uint32)]


def lib_path():
    # HFDL_GPU_LIB: load a specific build of the library (A/B measurements of two builds in one gpurun)
    return os.environ.get("HFDL_GPU_LIB") or os.path.join(_HERE, "libhfdl_gpu.so")


def lab_lib_path():
    return os.environ.get("HFDL_GPU_LAB_LIB") or os.path.join(_HERE, "libhfdl_gpu_lab.so")


_lib = None
_lab = None

# every symbol include/hfdl_gpu.h declares (tests/test_host_lib_cpu.py compares this list with the header and with `nm -D`)
EXPORTS = [
    "hfdl_gpu_plan_geometry", "hfdl_gpu_host_alloc", "hfdl_gpu_host_free",
    "hfdl_gpu_frontend_create", "hfdl_gpu_frontend_destroy", "hfdl_gpu_frontend_geometry",
    "hfdl_gpu_frontend_push_block", "hfdl_gpu_frontend_push_block_raw", "hfdl_gpu_frontend_input_done", "hfdl_gpu_frontend_input_done_upto", "hfdl_gpu_frontend_channelize_block", "hfdl_gpu_frontend_sync",
    "hfdl_gpu_frontend_poll_pdus", "hfdl_gpu_frontend_poll_pdus_ready", "hfdl_gpu_frontend_counters", "hfdl_gpu_frontend_all_channel_stats", "hfdl_gpu_frontend_stream", "hfdl_gpu_frontend_read_tap",
    "hfdl_gpu_frontend_channel_stats", "hfdl_gpu_frontend_enable_taps", "hfdl_gpu_frontend_fold_time_ms", "hfdl_gpu_frontend_demod_time_ms", "hfdl_gpu_frontend_reset_timers", "hfdl_gpu_frontend_step_period_ms", "hfdl_gpu_last_stage_ms",
    "hfdl_gpu_frontend_fold_blocks", "hfdl_gpu_frontend_fold_launch_shapes", "hfdl_gpu_frontend_stage_times", "hfdl_gpu_frontend_read_tap_block", "hfdl_gpu_frontend_push_baseband", "hfdl_gpu_frontend_input_copied",
    "hfdl_gpu_fft_forward", "hfdl_gpu_viterbi27", "hfdl_gpu_burst_decode", "hfdl_gpu_nco_decimate", "hfdl_gpu_crc16_ccitt", "hfdl_gpu_pdu_triage", "hfdl_gpu_lpdu_walk", "hfdl_gpu_frontend_prefetch_block_raw", "hfdl_gpu_frontend_prefetch_cancel", "hfdl_gpu_psk_slice",
    "hfdl_gpu_last_error", "hfdl_gpu_device_count",
]


FOLD_BATCH_MAX = 32        # HFDL_GPU_FOLD_BATCH_MAX of include/hfdl_gpu.h

# what include/hfdl_gpu_lab.h adds in the laboratory build (libhfdl_gpu_lab.so)
LAB_EXPORTS = ["hfdl_gpu_lab_fold_variant_count", "hfdl_gpu_lab_fold_variant_describe", "hfdl_gpu_lab_fold_variant_probe", "hfdl_gpu_lab_stream_read_probe",
               "hfdl_gpu_lab_read_constants", "hfdl_gpu_lab_clock_probe_read"]


def fold_variants():
    """The compiled tilings of the matrix-pipe fold kernel in the laboratory build: [(P, Q, W, D, max blocks, tap layout)]."""
    L = load_lab()
    out = []
    for v in range(L.hfdl_gpu_lab_fold_variant_count()):
        d = (C.c_int32 * 6)()
        _check(L.hfdl_gpu_lab_fold_variant_describe(v, C.byref(d)), L)
        out.append(tuple(d))
    return out


def load():
    """Load libhfdl_gpu.so.  If torch is going to be used in this process it must own the HIP runtime:
    torch bundles its own libamdhip64 with the same SONAME, so it is imported first when available in sys.modules."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise GpuError("libhfdl_gpu.so is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(dumphfdl_amd/csrc/build.sh). There is no CPU fallback.")
    if "torch" in sys.modules:
        import torch  # noqa: F401  (already imported: make sure its HIP runtime is the one resolved)
    _lib = _bind(C.CDLL(p, mode=C.RTLD_GLOBAL))
    return _lib


def load_lab():
    """Load the laboratory build (libhfdl_gpu_lab.so: include/hfdl_gpu_lab.h on top of include/hfdl_gpu.h).  It lives beside the
    product library in one process; Frontend(..., lib=load_lab()) binds a front end to it."""
    global _lab
    if _lab is not None:
        return _lab
    p = lab_lib_path()
    if not os.path.exists(p):
        raise GpuError("libhfdl_gpu_lab.so is missing: build it with dumphfdl_amd/csrc/build.sh lab")
    if "torch" in sys.modules:
        import torch  # noqa: F401
    L = _bind(C.CDLL(p, mode=C.RTLD_LOCAL))
    L.hfdl_gpu_lab_fold_variant_describe.argtypes = [C.c_int, C.POINTER(C.c_int32 * 6)]
    L.hfdl_gpu_lab_fold_variant_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.hfdl_gpu_lab_stream_read_probe.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.hfdl_gpu_lab_read_constants.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.hfdl_gpu_lab_clock_probe_read.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    _lab = L
    return L


def _bind(L):
    L.hfdl_gpu_last_error.restype = C.c_char_p
    L.hfdl_gpu_frontend_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    L.hfdl_gpu_frontend_destroy.argtypes = [C.c_void_p]
    L.hfdl_gpu_plan_geometry.argtypes = [C.c_int32, C.c_float, C.POINTER(Geometry)]
    L.hfdl_gpu_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.hfdl_gpu_host_free.argtypes = [C.c_void_p]
    L.hfdl_gpu_host_free.restype = None
    L.hfdl_gpu_frontend_destroy.restype = None
    L.hfdl_gpu_frontend_geometry.argtypes = [C.c_void_p, C.POINTER(Geometry)]
    L.hfdl_gpu_frontend_push_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.hfdl_gpu_frontend_channelize_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.hfdl_gpu_frontend_push_block_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.hfdl_gpu_frontend_push_baseband.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hfdl_gpu_frontend_sync.argtypes = [C.c_void_p]
    L.hfdl_gpu_frontend_input_done.argtypes = [C.c_void_p]
    L.hfdl_gpu_frontend_input_done_upto.argtypes = [C.c_void_p, C.c_uint64]
    L.hfdl_gpu_frontend_poll_pdus.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.hfdl_gpu_frontend_poll_pdus_ready.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    L.hfdl_gpu_frontend_counters.argtypes = [C.c_void_p, C.POINTER(FrontendCounters)]
    L.hfdl_gpu_frontend_all_channel_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.hfdl_gpu_frontend_stream.argtypes = [C.c_void_p]
    L.hfdl_gpu_frontend_stream.restype = C.c_void_p
    L.hfdl_gpu_frontend_read_tap.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.hfdl_gpu_frontend_read_tap_block.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.hfdl_gpu_frontend_enable_taps.argtypes = [C.c_void_p, C.c_int]
    L.hfdl_gpu_frontend_channel_stats.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ChannelStats)]
    L.hfdl_gpu_frontend_fold_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.hfdl_gpu_frontend_demod_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.hfdl_gpu_frontend_fold_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.hfdl_gpu_frontend_fold_launch_shapes.argtypes = [C.c_void_p, C.POINTER(C.c_int64 * (FOLD_BATCH_MAX + 1)), C.POINTER(C.c_double * (FOLD_BATCH_MAX + 1))]
    L.hfdl_gpu_frontend_input_copied.argtypes = [C.c_void_p, C.c_uint64]
    L.hfdl_gpu_frontend_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_double * 5), C.POINTER(C.c_int64 * 5)]
    L.hfdl_gpu_frontend_reset_timers.argtypes = [C.c_void_p, C.c_int]
    L.hfdl_gpu_fft_forward.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int]
    L.hfdl_gpu_viterbi27.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.hfdl_gpu_burst_decode.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.hfdl_gpu_frontend_step_period_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.hfdl_gpu_last_stage_ms.restype = C.c_double
    L.hfdl_gpu_nco_decimate.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                        C.c_void_p, C.POINTER(C.c_int32)]
    L.hfdl_gpu_crc16_ccitt.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint16, C.POINTER(C.c_uint16)]
    L.hfdl_gpu_pdu_triage.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hfdl_gpu_lpdu_walk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.hfdl_gpu_psk_slice.argtypes = [C.c_int, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.hfdl_gpu_frontend_prefetch_block_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.hfdl_gpu_frontend_prefetch_cancel.argtypes = [C.c_void_p]
    return L


def _check(rc, L=None):
    if rc != 0:
        raise GpuError("hfdl_gpu error %d: %s" % (rc, (L or load()).hfdl_gpu_last_error().decode(errors="replace")))


def plan_geometry(decimation, transition_bw):
    g = Geometry()
    _check(load().hfdl_gpu_plan_geometry(decimation, transition_bw, C.byref(g)))
    return g


def host_alloc(nbytes):
    p = C.c_void_p()
    _check(load().hfdl_gpu_host_alloc(C.byref(p), nbytes))
    return p.value


def host_free(ptr):
    load().hfdl_gpu_host_free(C.c_void_p(ptr))


def device_count():
    return load().hfdl_gpu_device_count()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Frontend:
    """All HFDL channels of one wideband receiver on one GPU (hfdl_gpu_frontend_*).

    Mirrors the reference wiring fft_create + N x hfdl_channel_create (src/main.c:699-755): one block of
    `geometry.input_size` samples in, PDUs out."""

    def __init__(self, sample_rate, centerfreq, freqs, device=0, lib=None):
        L = self._L = lib or load()
        fr = np.ascontiguousarray(freqs, dtype=np.int32)
        self._h = C.c_void_p()
        _check(L.hfdl_gpu_frontend_create(C.byref(self._h), device, sample_rate, centerfreq, _p(fr), len(fr)), self._L)
        self.geometry = Geometry()
        _check(L.hfdl_gpu_frontend_geometry(self._h, C.byref(self.geometry)), self._L)
        self.freqs = [int(f) for f in fr]

    @property
    def input_size(self):
        return self.geometry.input_size

    def push_block(self, samples):
        """samples: complex64 numpy array of input_size samples (host), or an int device pointer."""
        L = self._L
        if isinstance(samples, int):
            _check(L.hfdl_gpu_frontend_push_block(self._h, C.c_void_p(samples), self.geometry.input_size, 1), self._L)
        else:
            s = np.ascontiguousarray(samples, dtype=np.complex64)
            _check(L.hfdl_gpu_frontend_push_block(self._h, _p(s), len(s), 0), self._L)
            # the copy is enqueued asynchronously from pageable memory: HIP stages it before returning

    def push_block_raw(self, raw, sample_format):
        """raw: int16 (SFMT_CS16) / uint8 (SFMT_CU8) / float32 (SFMT_CF32) numpy array of 2*input_size interleaved I,Q values."""
        dt = {SFMT_CF32: np.float32, SFMT_CS16: np.int16, SFMT_CU8: np.uint8}[sample_format]
        r = np.ascontiguousarray(raw, dtype=dt)
        _check(self._L.hfdl_gpu_frontend_push_block_raw(self._h, _p(r), len(r) // 2, sample_format, 0), self._L)

    def channelize_block(self, samples):
        L = self._L
        if isinstance(samples, int):
            _check(L.hfdl_gpu_frontend_channelize_block(self._h, C.c_void_p(samples), self.geometry.input_size, 1), self._L)
        else:
            s = np.ascontiguousarray(samples, dtype=np.complex64)
            _check(L.hfdl_gpu_frontend_channelize_block(self._h, _p(s), len(s), 0), self._L)

    def push_baseband(self, per_channel):
        """per_channel: one complex64 array of channelizer OUTPUT per channel (<= max_outputs_per_block + 1 samples each): the demodulator
        and burst decoder stage alone, fed from the host (stage parity)."""
        g = self.geometry
        row = g.max_outputs_per_block + 1
        assert len(per_channel) == g.channels
        buf = np.zeros((g.channels, row), np.complex64)
        cnt = np.zeros(g.channels, np.int32)
        for c, x in enumerate(per_channel):
            x = np.asarray(x, np.complex64)
            buf[c, :len(x)] = x
            cnt[c] = len(x)
        _check(self._L.hfdl_gpu_frontend_push_baseband(self._h, _p(buf), _p(cnt)), self._L)

    def push_host_ptr(self, ptr, sample_format=SFMT_CF32):
        """ptr: address of a (page-locked) host buffer holding one block; valid until input_done() / sync()."""
        _check(self._L.hfdl_gpu_frontend_push_block_raw(self._h, C.c_void_p(ptr), self.geometry.input_size, sample_format, 0), self._L)

    def prefetch_host_ptr(self, ptr, sample_format=SFMT_CF32):
        """Queue the copy of a block that push_host_ptr(ptr, sample_format) will push later (up to geometry.prefetch_depth wait this
        way; they are pushed in the order they were prefetched)."""
        _check(self._L.hfdl_gpu_frontend_prefetch_block_raw(self._h, C.c_void_p(ptr), self.geometry.input_size, sample_format), self._L)

    def prefetch_cancel(self):
        _check(self._L.hfdl_gpu_frontend_prefetch_cancel(self._h), self._L)

    def input_done(self):
        _check(self._L.hfdl_gpu_frontend_input_done(self._h), self._L)

    def input_copied(self, host_block):
        """True once the upload of that host block has finished (no waiting)."""
        rc = self._L.hfdl_gpu_frontend_input_copied(self._h, C.c_uint64(host_block))
        if rc < 0:
            _check(rc, self._L)
        return bool(rc)

    def input_done_upto(self, host_block):
        _check(self._L.hfdl_gpu_frontend_input_done_upto(self._h, C.c_uint64(host_block)), self._L)

    def sync(self):
        _check(self._L.hfdl_gpu_frontend_sync(self._h), self._L)

    def poll_pdus_raw(self, max_pdus=4096, max_in_flight=0):
        """The C structs as the library hands them over: (ctypes array of hfdl_gpu_pdu, count)."""
        buf = (Pdu * max_pdus)()
        n = C.c_int32(0)
        if max_in_flight:
            _check(self._L.hfdl_gpu_frontend_poll_pdus_ready(self._h, buf, max_pdus, C.byref(n), max_in_flight), self._L)
        else:
            _check(self._L.hfdl_gpu_frontend_poll_pdus(self._h, buf, max_pdus, C.byref(n)), self._L)
        return buf, n.value

    @staticmethod
    def pdus_to_dicts(buf, n):
        out = []
        for i in range(n):
            p = buf[i]
            out.append(dict(channel=p.channel, freq=p.freq, mode=p.mode, bit_rate=p.bit_rate,
                            octets=bytes(p.octets[:p.len]), freq_err_hz=p.freq_err_hz, rssi_db=p.rssi_db,
                            noise_floor_db=p.noise_floor_db, slot=p.slot.decode(), sample_index=p.sample_index,
                            fcs_status=p.fcs_status, pdu_kind=p.pdu_kind, hdr_len=p.hdr_len,
                            train_bits_bad=p.train_bits_bad, train_bits_total=p.train_bits_total,
                            lpdus=(p.lpdus_processed, p.lpdus_good, p.lpdus_bad_fcs, p.lpdus_too_short, p.lpdus_truncated)))
        return out

    def poll_pdus(self, max_pdus=4096, max_in_flight=0):
        """max_in_flight=0: everything decoded so far (drains the pipeline); 1: nothing is drained -- the half being filled keeps
        filling, the newest closed half keeps running, and what is known complete (the half before it) is returned: nothing until two
        halves of geometry.fold_batch blocks were closed, and the tail only comes with a draining poll (include/hfdl_gpu.h)."""
        return self.pdus_to_dicts(*self.poll_pdus_raw(max_pdus, max_in_flight))

    def counters(self):
        c = FrontendCounters()
        _check(self._L.hfdl_gpu_frontend_counters(self._h, C.byref(c)), self._L)
        return {n: getattr(c, n) for n, _ in FrontendCounters._fields_}

    def all_channel_stats(self):
        nch = self.geometry.channels
        buf = (ChannelStats * nch)()
        n = C.c_int32(0)
        _check(self._L.hfdl_gpu_frontend_all_channel_stats(self._h, buf, nch, C.byref(n)), self._L)
        return [{k: getattr(buf[i], k) for k, _ in ChannelStats._fields_} for i in range(n.value)]

    def read_tap(self, what, channel=0, back=0):
        """back: TAP_SPECTRUM / TAP_CHAN_OUT / TAP_NCO_PHASORS of the block `back` blocks before the newest (within the newest half)."""
        g = self.geometry
        cap = 2 * g.fft_size if what in (TAP_SPECTRUM, TAP_FILTER) else 2 * (g.post_input_size + 64) * max(1, g.demod_batch)   # demodulator taps cover a launch
        buf = np.empty(cap, np.float32)
        n = C.c_size_t(0)
        _check(self._L.hfdl_gpu_frontend_read_tap_block(self._h, what, channel, back, _p(buf), cap, C.byref(n)), self._L)
        out = buf[:n.value].copy()
        return out if what in (TAP_AGC_LEVEL, TAP_PHASE_CYCLES) else out.view(np.complex64)

    def enable_taps(self, enable=True):
        _check(self._L.hfdl_gpu_frontend_enable_taps(self._h, int(enable)), self._L)

    def channel_stats(self, channel):
        st = ChannelStats()
        _check(self._L.hfdl_gpu_frontend_channel_stats(self._h, channel, C.byref(st)), self._L)
        return {n: getattr(st, n) for n, _ in ChannelStats._fields_}

    def reset_timers(self, enable=True):
        _check(self._L.hfdl_gpu_frontend_reset_timers(self._h, int(enable)), self._L)

    def fold_time_ms(self):
        ms = C.c_double(0)
        n = C.c_int64(0)
        _check(self._L.hfdl_gpu_frontend_fold_time_ms(self._h, C.byref(ms), C.byref(n)), self._L)
        return ms.value, n.value

    def fold_blocks(self):
        """Blocks covered by the timed fold launches (a launch folds up to geometry.fold_batch blocks)."""
        n = C.c_int64(0)
        _check(self._L.hfdl_gpu_frontend_fold_blocks(self._h, C.byref(n)), self._L)
        return n.value

    def fold_launch_shapes(self):
        """{blocks per launch: timed fold launches of that size} since reset_timers(True)."""
        c = (C.c_int64 * (FOLD_BATCH_MAX + 1))()
        _check(self._L.hfdl_gpu_frontend_fold_launch_shapes(self._h, C.byref(c), None), self._L)
        return {nb: int(c[nb]) for nb in range(1, FOLD_BATCH_MAX + 1) if c[nb]}

    def fold_launch_times(self):
        """{blocks per launch: (timed fold launches of that size, their kernel time in ms)} since reset_timers(True)."""
        c, ms = (C.c_int64 * (FOLD_BATCH_MAX + 1))(), (C.c_double * (FOLD_BATCH_MAX + 1))()
        _check(self._L.hfdl_gpu_frontend_fold_launch_shapes(self._h, C.byref(c), C.byref(ms)), self._L)
        return {nb: (int(c[nb]), float(ms[nb])) for nb in range(1, FOLD_BATCH_MAX + 1) if c[nb]}

    def stage_times(self):
        """{stage: (total ms, launches)} of the timed kernels since reset_timers(True): fft (per block), fold, ifft, demod, decode."""
        ms, n = (C.c_double * 5)(), (C.c_int64 * 5)()
        _check(self._L.hfdl_gpu_frontend_stage_times(self._h, C.byref(ms), C.byref(n)), self._L)
        return {k: (ms[i], int(n[i])) for i, k in enumerate(("fft", "fold", "ifft", "demod", "decode"))}

    def fold_variant_probe(self, variant, nb, reps=3):
        """Laboratory build: (avg ms, best ms, checksum of the partial sums) of `reps` launches of one compiled fold tiling on `nb` blocks
        (variant -1: the plain-VALU FMA-chain reference kernel)."""
        avg, best, chk = C.c_double(0), C.c_double(0), C.c_uint64(0)
        _check(self._L.hfdl_gpu_lab_fold_variant_probe(self._h, variant, nb, reps, C.byref(avg), C.byref(best), C.byref(chk)), self._L)
        return avg.value, best.value, chk.value

    def demod_time_ms(self):
        ms = C.c_double(0)
        n = C.c_int64(0)
        nb = C.c_int64(0)
        _check(self._L.hfdl_gpu_frontend_demod_time_ms(self._h, C.byref(ms), C.byref(n), C.byref(nb)), self._L)
        return ms.value, n.value, nb.value

    def step_period_ms(self):
        v = C.c_double(0)
        _check(self._L.hfdl_gpu_frontend_step_period_ms(self._h, C.byref(v)), self._L)
        return v.value

    def stream_read_probe(self):
        """Laboratory build: GB/s a bare read-only kernel reaches over the resident filter taps."""
        v = C.c_double(0)
        _check(self._L.hfdl_gpu_lab_stream_read_probe(self._h, C.byref(v)), self._L)
        return v.value

    def stream(self):
        return self._L.hfdl_gpu_frontend_stream(self._h)

    def close(self):
        if self._h:
            self._L.hfdl_gpu_frontend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fft_forward(x, shifted=False, device=0):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty_like(x)
    _check(load().hfdl_gpu_fft_forward(device, _p(x), _p(out), len(x), int(shifted)))
    return out


def viterbi27(soft, nbits, device=0):
    """soft: uint8 array [nframes, 2*nbits] -> uint8 [nframes, ceil(nbits/8)] (libfec bit order)."""
    soft = np.ascontiguousarray(soft, dtype=np.uint8).reshape(-1, 2 * nbits)
    out = np.zeros((soft.shape[0], (nbits + 7) // 8), np.uint8)
    _check(load().hfdl_gpu_viterbi27(device, _p(soft), nbits, soft.shape[0], _p(out)))
    return out


def burst_decode(symbol_list, modes, bitmask_lsb=None, device=0):
    """decode_user_data for a batch of frames; symbol_list[i] = the segments*30 equalised data symbols of frame i."""
    n = len(symbol_list)
    modes = np.ascontiguousarray(modes, dtype=np.int32)
    bm = np.zeros(n, np.int32) if bitmask_lsb is None else np.ascontiguousarray(bitmask_lsb, dtype=np.int32)
    sym = np.ascontiguousarray(np.concatenate([np.asarray(s, np.complex64) for s in symbol_list]), dtype=np.complex64)
    octets = np.zeros((n, PDU_MAX_OCTETS), np.uint8)
    lens = np.zeros(n, np.int32)
    _check(load().hfdl_gpu_burst_decode(device, _p(sym), _p(modes), _p(bm), n, _p(octets), _p(lens)))
    return [bytes(octets[i, :lens[i]]) for i in range(n)]


def last_stage_ms():
    return load().hfdl_gpu_last_stage_ms()


def nco_decimate(x, rate, decimation, decimation_remain=0, starting_phase=0.0, device=0):
    """decimating_shift_addition_cc on the device; returns (out, decimation_remain, starting_phase) -- feed the state back in."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.zeros((len(x) + decimation - 1) // decimation, np.complex64)
    rem, ph, n = C.c_int32(decimation_remain), C.c_float(starting_phase), C.c_int32(0)
    _check(load().hfdl_gpu_nco_decimate(device, _p(x), len(x), rate, decimation, C.byref(rem), C.byref(ph), _p(out), C.byref(n)))
    return out[:n.value].copy(), rem.value, ph.value


def crc16_ccitt(data, crc_init=0xFFFF, device=0):
    a = np.frombuffer(bytes(data), np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
    crc = C.c_uint16(0)
    _check(load().hfdl_gpu_crc16_ccitt(device, _p(a), len(data), crc_init, C.byref(crc)))
    return crc.value


def pdu_triage(pdus, device=0):
    """pdus: list of bytes -> list of (fcs_status, pdu_kind, hdr_len) computed on the device."""
    n = len(pdus)
    stride = max(len(p) for p in pdus)
    octets = np.zeros((n, stride), np.uint8)
    for i, p in enumerate(pdus):
        octets[i, :len(p)] = np.frombuffer(bytes(p), np.uint8)
    lens = np.array([len(p) for p in pdus], np.int32)
    fcs, kind, hl = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.uint16)
    _check(load().hfdl_gpu_pdu_triage(device, _p(octets), _p(lens), n, stride, _p(fcs), _p(kind), _p(hl)))
    return [(int(fcs[i]), int(kind[i]), int(hl[i])) for i in range(n)]


def lpdu_walk(pdus, device=0):
    """pdus: list of bytes -> list of (processed, good, bad_fcs, too_short, truncated) computed on the device."""
    n = len(pdus)
    stride = max(len(p) for p in pdus)
    octets = np.zeros((n, stride), np.uint8)
    for i, p in enumerate(pdus):
        octets[i, :len(p)] = np.frombuffer(bytes(p), np.uint8)
    lens = np.array([len(p) for p in pdus], np.int32)
    counts = np.zeros((n, 5), np.uint8)
    _check(load().hfdl_gpu_lpdu_walk(device, _p(octets), _p(lens), n, stride, _p(counts)))
    return [tuple(int(v) for v in counts[i]) for i in range(n)]


def psk_slice(arity, symbols, device=0):
    """symbols: complex64 array -> (Gray-coded decisions uint32, phase errors float32) from the carrier loop's slicer."""
    x = np.ascontiguousarray(symbols, dtype=np.complex64)
    sym = np.zeros(len(x), np.uint32)
    err = np.zeros(len(x), np.float32)
    _check(load().hfdl_gpu_psk_slice(device, arity, _p(x), len(x), _p(sym), _p(err)))
    return sym, err
