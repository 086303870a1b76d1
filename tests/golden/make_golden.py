"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code (container only).

Runs the reference's viterbi27_port.c, crc.c and libcsdr_gpl.c -- compiled unmodified into oracle/_ref/libhfdl_ref.so
by oracle/Makefile -- on seeded inputs and stores inputs + outputs.  The fixtures are data; no reference source
is copied.  Re-run with:  python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O  # noqa: E402
import hfdl_synth as synth  # noqa: E402

R = O.ref()
assert R is not None, "needs /root/reference (oracle/_ref/libhfdl_ref.so)"


def viterbi_vectors():
    rng = np.random.default_rng(20260927)
    out = {}
    for mode in range(8):
        nbits = synth.mode_sizes(mode)["nbits"]
        bits = rng.integers(0, 2, nbits).astype(np.uint8)
        bits[-6:] = 0
        coded = synth.conv_encode(bits).astype(float) * 255
        cases = [np.clip(coded, 0, 255).astype(np.uint8),
                 np.clip(coded + rng.normal(0, 70, len(coded)), 0, 255).astype(np.uint8)]
        if mode in (0, 3):
            cases.append(rng.integers(0, 256, 2 * nbits).astype(np.uint8))
        for i, soft in enumerate(cases):
            out["m%d_c%d_soft" % (mode, i)] = soft
            out["m%d_c%d_out" % (mode, i)] = O.ref_viterbi27(soft, nbits)
        out["m%d_bits" % mode] = np.packbits(bits)
    np.savez_compressed(os.path.join(HERE, "viterbi_ref.npz"), **out)


def crc_vectors():
    rng = np.random.default_rng(7)
    vec = []
    for n in [0, 1, 2, 9, 64, 66, 255, 945]:
        d = b"123456789" if n == 9 else rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        a = np.frombuffer(d, np.uint8).copy() if n else np.zeros(1, np.uint8)
        for init in (0xFFFF, 0x0000, 0x1D0F):
            vec.append(dict(data=d.hex(), init=init, crc=int(R.crc16_ccitt(a.ctypes.data, n, init))))
    json.dump(vec, open(os.path.join(HERE, "crc_ref.json"), "w"), indent=0)


class SA(C.Structure):
    _fields_ = [("sindelta", C.c_float), ("cosdelta", C.c_float), ("rate", C.c_float)]


class ST(C.Structure):
    _fields_ = [("decimation_remain", C.c_int32), ("starting_phase", C.c_float), ("output_size", C.c_int32)]


def nco_vectors():
    R.decimating_shift_addition_init.restype = SA
    R.decimating_shift_addition_init.argtypes = [C.c_float, C.c_int]
    R.decimating_shift_addition_cc.restype = ST
    R.decimating_shift_addition_cc.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, SA, C.c_int32, ST]
    rng = np.random.default_rng(5)
    out = {}
    for case, (rate, dec, n) in enumerate([(0.0123, 2, 1792), (-0.21, 2, 3584), (0.4031, 3, 1001), (0.0, 2, 64)]):
        d = R.decimating_shift_addition_init(rate, dec)
        st = ST(0, 0.0, 0)
        xs, ys, sts = [], [], []
        for blk in range(3):
            x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
            y = np.zeros(n, np.complex64)
            st = R.decimating_shift_addition_cc(x.ctypes.data, y.ctypes.data, n, d, dec, st)
            xs.append(x); ys.append(y[:st.output_size].copy())
            sts.append([st.decimation_remain, st.starting_phase, st.output_size])
        out["c%d_params" % case] = np.array([rate, dec, n, d.sindelta, d.cosdelta, d.rate], np.float64)
        out["c%d_x" % case] = np.stack(xs)
        out["c%d_y" % case] = np.concatenate(ys)
        out["c%d_state" % case] = np.array(sts, np.float64)
    np.savez_compressed(os.path.join(HERE, "nco_ref.npz"), **out)


if __name__ == "__main__":
    viterbi_vectors()
    crc_vectors()
    nco_vectors()
    print("golden vectors written to", HERE)
