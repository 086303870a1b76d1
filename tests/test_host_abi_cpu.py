"""include/hfdl_host.h against the reference's declarations: struct layouts, enumerator values and entry-point signatures are pinned
by _Static_asserts (tests/abi/hfdl_host_abi.c, every number derived from the reference line it cites) in BOTH builds of the host
program -- default and WITH_SOAPYSDR (src/input-common.h:8-15: INPUT_TYPE_FILE is 1 or 2) -- and the one library binary serves both
numberings at run time."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dumphfdl_amd")
ABI_C = os.path.join(ROOT, "tests", "abi", "hfdl_host_abi.c")


@pytest.mark.parametrize("flags", [[], ["-DWITH_SOAPYSDR"]], ids=["default", "with_soapysdr"])
def test_header_layouts_and_enums_match_the_reference(flags):
    out = subprocess.run(["gcc", "-std=c11", "-D_GNU_SOURCE", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), ABI_C] + flags,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_static_asserts_bite(tmp_path):
    """The pin file is not vacuous: a header whose enum lists INPUT_TYPE_SOAPYSDR unconditionally (round 3's) fails it."""
    hdr = open(os.path.join(ROOT, "include", "hfdl_host.h")).read()
    broken = hdr.replace("#ifdef WITH_SOAPYSDR\n\tINPUT_TYPE_SOAPYSDR,\n#endif", "\tINPUT_TYPE_SOAPYSDR,")
    assert broken != hdr
    (tmp_path / "hfdl_host.h").write_text(broken)
    out = subprocess.run(["gcc", "-std=c11", "-D_GNU_SOURCE", "-fsyntax-only", "-I", str(tmp_path), ABI_C], capture_output=True, text=True)
    assert out.returncode != 0 and "INPUT_TYPE_FILE != 1" in out.stderr


def test_reference_enum_text_if_present():
    """Where the reference tree is available (the build container), its own text says what the pins say."""
    path = "/root/reference/src/input-common.h"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    txt = open(path).read()
    m = re.search(r"typedef enum \{\s*INPUT_TYPE_UNDEF,\s*#ifdef WITH_SOAPYSDR\s*INPUT_TYPE_SOAPYSDR,\s*#endif\s*INPUT_TYPE_FILE,\s*INPUT_TYPE_MAX\s*\} input_type;", txt)
    assert m, "the reference's input_type enum changed shape"


@pytest.fixture(scope="module")
def host_checks(tmp_path_factory):
    subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "host")])
    d = tmp_path_factory.mktemp("abi")
    exes = {}
    for name, flags in (("default", []), ("with_soapysdr", ["-DWITH_SOAPYSDR"])):
        exe = str(d / ("host_check_" + name))
        subprocess.check_call(["gcc", "-O1", "-std=c11", "-D_GNU_SOURCE", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "host"),
                               os.path.join(ROOT, "tests", "hostsim", "host_check.c"), "-o", exe] + flags +
                              ["-L", PKG, "-lhfdl_host", "-lhfdl_gpu", "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-lm"])
        exes[name] = exe
    return exes


@pytest.mark.parametrize("build", ["default", "with_soapysdr"])
def test_one_library_serves_both_numberings(host_checks, tmp_path, build):
    """A host program compiled either way gets the file input from cfg->type = INPUT_TYPE_FILE (1 or 2), and only the WITH_SOAPYSDR
    build has a radio slot to register: the same libhfdl_host.so."""
    import numpy as np
    raw = np.random.default_rng(1).standard_normal(2 * 5000).astype(np.float32)
    src, dst = tmp_path / "in.bin", tmp_path / "out.cf32"
    raw.tofile(src)
    out = subprocess.run([host_checks[build], "file", str(src), "CF32", "4096", str(dst)], capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert np.array_equal(np.fromfile(dst, np.float32), raw)
    out = subprocess.run([host_checks[build], "plugin"], capture_output=True, text=True)
    assert out.returncode == 0 and "plugin ok" in out.stdout, (out.returncode, out.stdout, out.stderr)
    assert ("no slot" in out.stdout) == (build == "default")
