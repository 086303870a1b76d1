"""CPU tests of libhfdl_host.so (plain-C block runtime / inputs keeping the reference's interface), driven from a small C
program the way dumphfdl's main() drives the reference.  No GPU work: the front-end thread is never started here."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dumphfdl_amd")


@pytest.fixture(scope="module")
def host_check(tmp_path_factory):
    subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "host")])
    exe = str(tmp_path_factory.mktemp("hc") / "host_check")
    # built as dumphfdl is WITH_SOAPYSDR: the radio slot exists (tests/test_host_abi_cpu.py runs both numberings)
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-D_GNU_SOURCE", "-DWITH_SOAPYSDR", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "host"),
                           os.path.join(ROOT, "tests", "hostsim", "host_check.c"), "-o", exe,
                           "-L", PKG, "-lhfdl_host", "-lhfdl_gpu", "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-lm"])
    return exe


def test_ring_and_overrun(host_check):
    out = subprocess.run([host_check, "ring"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Sample buffer overrun (4/6 samples lost)" in out.stderr       # the reference's message, src/input-helpers.c:85
    assert "ring ok" in out.stdout


def test_input_plugin_slot(host_check):
    """input_vtable_register(): a host-provided input (the SoapySDR slot) runs through input_create / input_init / block_start."""
    out = subprocess.run([host_check, "plugin"], capture_output=True, text=True)
    assert out.returncode == 0 and "plugin ok" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_block_graph_contract(host_check):
    out = subprocess.run([host_check, "graph"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


@pytest.mark.parametrize("fmt,bufsize", [("CF32", 320000), ("CS16", 4096), ("CU8", 1000), ("CF32", 8)])
def test_file_input_conversion(host_check, tmp_path, fmt, bufsize):
    rng = np.random.default_rng(3)
    n = 30011
    if fmt == "CF32":
        raw = rng.standard_normal(2 * n).astype(np.float32)
        want = raw.view(np.complex64)
    elif fmt == "CS16":
        raw = rng.integers(-32768, 32768, 2 * n).astype(np.int16)
        want = (raw.astype(np.float32) / np.float32(32767.5)).view(np.complex64)      # src/input-helpers.c:116
    else:
        raw = rng.integers(0, 256, 2 * n).astype(np.uint8)
        want = ((raw.astype(np.float32) - np.float32(63.5)) / np.float32(127.0)).view(np.complex64)   # :108-113, :71
    src, dst = tmp_path / "in.bin", tmp_path / "out.cf32"
    raw.tofile(src)
    out = subprocess.run([host_check, "file", str(src), fmt, str(bufsize), str(dst)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(dst, np.complex64)
    assert len(got) == n and np.array_equal(got, want)
    assert ("max_tu %d " % (bufsize // {"CF32": 8, "CS16": 4, "CU8": 2}[fmt])) in out.stdout


@pytest.mark.parametrize("n", [5 * 1024, 30011])
def test_file_input_converting_loop_replays(host_check, tmp_path, n):
    """--loop through the converting reader (a cs16 file in front of a cf32 ring): the file is replayed from the start, also
    when its length is a whole number of read buffers (the read that hits end of file then returns nothing)."""
    raw = np.random.default_rng(4).integers(-32768, 32768, 2 * n).astype(np.int16)
    src, dst = tmp_path / "in.bin", tmp_path / "out.cf32"
    raw.tofile(src)
    out = subprocess.run([host_check, "file", str(src), "CS16", "4096", str(dst), "3"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(dst, np.complex64)
    want = np.tile((raw.astype(np.float32) / np.float32(32767.5)).view(np.complex64), 3)
    assert len(got) == len(want) and np.array_equal(got, want)


@pytest.mark.parametrize("fmt,loops", [("CS16", 1), ("CU8", 1), ("CF32", 1), ("CS16", 3)])
def test_file_input_direct_into_frontend_ring(host_check, tmp_path, fmt, loops):
    """In front of the GPU front-end block the ring carries the file's RAW samples (the device converts them), is a whole
    number of front-end blocks long, and the file is read straight into it: what the consumer sees in place is the file,
    byte for byte (x loops), with an odd trailing fragment of a sample dropped at end of file (whole_samples())."""
    rng = np.random.default_rng(9)
    bps = {"CF32": 8, "CS16": 4, "CU8": 2}[fmt]
    n = 3 * 28672 + 12345
    raw = rng.integers(0, 256, n * bps + (3 if bps > 2 else 1), dtype=np.uint8)       # trailing fragment of a sample
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    raw.tofile(src)
    out = subprocess.run([host_check, "direct", str(src), fmt, str(dst), str(loops)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, (out.returncode, out.stderr)
    got = np.fromfile(dst, np.uint8)
    want = np.tile(raw[:n * bps], loops)
    assert len(got) == len(want) and np.array_equal(got, want)
    assert ("samples %d blocks %d elem %d " % (n * loops, n * loops // 28672, bps)) in out.stdout


def test_file_input_rejects_bad_config(host_check, tmp_path):
    src = tmp_path / "x.bin"
    np.zeros(16, np.float32).tofile(src)
    assert subprocess.run([host_check, "file", str(src), "CF32", "12", str(tmp_path / "o")], capture_output=True).returncode == 2
    assert subprocess.run([host_check, "file", str(tmp_path / "missing"), "CF32", "80", str(tmp_path / "o")], capture_output=True).returncode == 2
    assert subprocess.run([host_check, "file", str(src), "XX", "80", str(tmp_path / "o")], capture_output=True).returncode == 2


def test_planner_abi_without_device():
    import dumphfdl_amd as hf
    g = hf.plan_geometry(4096, 250 / 40e6)
    assert (g.fft_size, g.fft_inv_size, g.input_size, g.post_input_size, g.scrap, g.outputs_per_block) == \
        (8388608, 4096, 7340032, 3584, 512, 1792)
    with pytest.raises(hf.GpuError):
        hf.plan_geometry(0, 0.001)


def test_ring_against_a_model(host_check):
    """The sample ring (classic cf32 and raw-sample forms) driven through random sequences of its operations -- copying
    writes / reads, in-place producer (acquire / commit with partial samples carried over), in-place consumer (peek / drop) --
    against a byte-FIFO model: sizes, contents and the contiguity contract of peek() hold after every step."""
    import ctypes as C
    from hypothesis import given, settings, strategies as st
    L = C.CDLL(os.path.join(PKG, "libhfdl_host.so"))
    L.hfdl_ring_create_ex.restype = C.c_void_p
    L.hfdl_ring_create_ex.argtypes = [C.c_size_t, C.c_int, C.c_int]
    L.hfdl_ring_destroy.argtypes = [C.c_void_p]
    for name in ("hfdl_ring_size", "hfdl_ring_space_available", "hfdl_ring_capacity", "hfdl_ring_elem_size"):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [C.c_void_p]
    L.hfdl_ring_write.restype = C.c_size_t
    L.hfdl_ring_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.hfdl_ring_read.restype = C.c_size_t
    L.hfdl_ring_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.hfdl_ring_write_acquire.restype = C.c_size_t
    L.hfdl_ring_write_acquire.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.hfdl_ring_write_commit.restype = C.c_size_t
    L.hfdl_ring_write_commit.argtypes = [C.c_void_p, C.c_size_t]
    L.hfdl_ring_discard_partial.argtypes = [C.c_void_p]
    L.hfdl_ring_peek.restype = C.c_void_p
    L.hfdl_ring_peek.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    L.hfdl_ring_drop.restype = C.c_size_t
    L.hfdl_ring_drop.argtypes = [C.c_void_p, C.c_size_t]
    SFMT = {2: 1, 4: 2, 8: 3}                                    # element bytes -> sample_format (CU8, CS16, CF32)
    op = st.tuples(st.sampled_from(["write", "read", "fill", "peek", "drop", "discard"]), st.integers(0, 40), st.integers(0, 7))

    @settings(max_examples=150, deadline=None)
    @given(st.sampled_from([2, 4, 8]), st.integers(0, 24), st.lists(op, max_size=60), st.randoms(use_true_random=False))
    def run(elem, cap, ops, rnd):
        r = L.hfdl_ring_create_ex(cap, SFMT[elem], 0)
        assert L.hfdl_ring_capacity(r) == cap and L.hfdl_ring_elem_size(r) == elem
        model = bytearray()                                    # whole samples readable
        carry = bytearray()                                    # bytes of an incomplete sample behind them
        fresh = lambda n: bytes(rnd.getrandbits(8) for _ in range(n))
        for kind, n, frag in ops:
            if kind == "write":                                # copying producer: cf32 rings only, a raw ring refuses it
                if carry:
                    continue                                   # (mixing it with a pending in-place fragment is not a supported use)
                data = fresh(n * elem)
                took = L.hfdl_ring_write(r, data, n)
                assert took == (min(n, cap - len(model) // elem) if elem == 8 else 0)
                model += data[:took * elem]
            elif kind == "read":
                buf = C.create_string_buffer(max(1, n * elem))
                got = L.hfdl_ring_read(r, buf, n)
                if elem == 8:
                    assert got == min(n, len(model) // elem) and buf.raw[:got * elem] == bytes(model[:got * elem])
                    del model[:got * elem]
                else:
                    assert got == 0
            elif kind == "fill":                               # in-place producer: some bytes, possibly ending mid-sample
                ptr = C.c_void_p()
                room = L.hfdl_ring_write_acquire(r, C.byref(ptr))
                free_samples = cap - len(model) // elem
                assert (room == 0) == (free_samples == 0)
                if room:
                    assert 0 < room <= free_samples * elem - len(carry)
                    nbytes = min(room, n * elem + frag % elem)
                    data = fresh(nbytes)
                    C.memmove(ptr.value, data, nbytes)
                    made = L.hfdl_ring_write_commit(r, nbytes)
                    carry += data
                    whole = len(carry) // elem
                    assert made == whole
                    model += carry[:whole * elem]
                    del carry[:whole * elem]
            elif kind == "peek":
                avail = len(model) // elem
                off = n % (avail + 1)
                cnt = frag % (avail - off + 1)
                p = L.hfdl_ring_peek(r, off, cnt)
                if p:                                            # contiguous: exactly the model's bytes
                    assert C.string_at(p, cnt * elem) == bytes(model[off * elem:(off + cnt) * elem])
                assert L.hfdl_ring_peek(r, avail, 1) is None     # never beyond what is readable
            elif kind == "drop":
                got = L.hfdl_ring_drop(r, n)
                assert got == min(n, len(model) // elem)
                del model[:got * elem]
            else:
                L.hfdl_ring_discard_partial(r)
                carry.clear()
            assert L.hfdl_ring_size(r) == len(model) // elem
            assert L.hfdl_ring_space_available(r) == cap - len(model) // elem
        L.hfdl_ring_destroy(r)

    run()


def test_host_library_under_sanitizers(tmp_path):
    """The host library and its harness built with -fsanitize=address,undefined: ring, block graph, plug-in slot, the converting
    and the direct file readers (with loops) run clean -- no invalid access, no undefined behaviour, nothing left allocated."""
    probe = tmp_path / "p.c"
    probe.write_text("int main(void){return 0;}\n")
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    if subprocess.run(["gcc"] + san + [str(probe), "-o", str(tmp_path / "p")], capture_output=True).returncode != 0:
        pytest.skip("this gcc has no sanitizer runtime")
    host = os.path.join(PKG, "host")
    exe = str(tmp_path / "host_check_san")
    subprocess.check_call(["gcc", "-g", "-O1", "-std=c11", "-D_GNU_SOURCE"] + san + ["-I", os.path.join(ROOT, "include"), "-I", host] +
                          [os.path.join(host, f) for f in ("ring.c", "blocks.c", "input.c", "sink.c", "frontend.c")] +
                          [os.path.join(ROOT, "tests", "hostsim", "host_check.c"), "-o", exe,
                           "-L", PKG, "-lhfdl_gpu", "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-lm"])
    rng = np.random.default_rng(3)
    a, b = tmp_path / "a.cs16", tmp_path / "b.bin"
    rng.integers(-32768, 32768, 2 * 30011).astype(np.int16).tofile(a)
    rng.integers(0, 256, (3 * 28672 + 12345) * 4 + 3, dtype=np.uint8).tofile(b)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for args in (["ring"], ["graph"], ["plugin"], ["file", str(a), "CS16", "4096", str(tmp_path / "o1"), "3"],
                 ["file", str(a), "CF32", "320000", str(tmp_path / "o2")], ["direct", str(b), "CS16", str(tmp_path / "o3"), "3"],
                 ["direct", str(b), "CU8", str(tmp_path / "o4"), "1"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120, env=env)
        assert out.returncode == 0 and "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, (args, out.stderr[-1500:])


@pytest.mark.timeout(90)
def test_pipe_input_fragment_across_the_ring_wrap(host_check, tmp_path):
    """Samples arriving through a pipe in pieces that are not whole samples (nc / ssh do that): the fragment of a sample is
    carried to the next read.  When it sits in the LAST slot of the ring's storage the free run is shorter than a sample --
    the producer must still read the bytes that complete it (it used to wait for a whole sample of room: forever)."""
    bps, blk = 4, 28672
    probe = tmp_path / "probe.bin"
    np.zeros(2 * blk * bps, np.uint8).tofile(probe)
    out = subprocess.run([host_check, "direct", str(probe), "CS16", str(tmp_path / "o0"), "1"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    cap = int(out.stdout.split("capacity")[1].split()[0])
    n = cap + 2 * blk + 77                                        # more than the storage holds: the tail wraps once
    raw = np.random.default_rng(11).integers(0, 256, n * bps, dtype=np.uint8).tobytes()
    stop_at = (cap - 1) * bps + 1                                 # one byte into the last sample slot of the storage
    dst = tmp_path / "out.bin"
    proc = subprocess.Popen([host_check, "direct", "-", "CS16", str(dst), "1"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    import time
    pos = 0
    for size in (4099, 12345, 1):                                 # odd-sized first writes: reads return fragments
        proc.stdin.write(raw[pos:pos + size]); proc.stdin.flush(); pos += size
        time.sleep(0.02)
    proc.stdin.write(raw[pos:stop_at]); proc.stdin.flush(); pos = stop_at
    time.sleep(0.3)                                               # the reader drains the pipe: fragment pending at the last slot
    proc.stdin.write(raw[pos:pos + 2]); proc.stdin.flush(); pos += 2
    time.sleep(0.1)
    proc.stdin.write(raw[pos:]); proc.stdin.close()
    try:
        rc = proc.wait(timeout=30)
    except subprocess.TimeoutExpired:
        proc.kill()
        pytest.fail("the pipe reader hung with a sample fragment pending at the ring's wrap point")
    stdout = proc.stdout.read().decode()
    assert rc == 0, proc.stderr.read().decode()
    got = np.fromfile(dst, np.uint8).tobytes()
    assert got == raw and ("samples %d " % n) in stdout
