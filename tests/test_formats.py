"""On-disk formats of the Java driver (ldagroupedgibbssampler_amd/formats.py).

The binary and integer formats are byte-exact by construction and are checked against hand-packed bytes (what
ByteBuffer.putDouble / putInt and DataInputStream do is fixed by the JDK specification: big-endian).  The
double-to-text rules are restated from the JDK documentation; the reference holds no output file and the image no
JVM, so those cases are the documentation's own examples -- parity unpinned.
"""
import os
import struct

import numpy as np
import pytest

from ldagroupedgibbssampler_amd import formats as F


def test_binary_double_matrix_bytes(tmp_path):
    m = np.array([[1.0, -2.5, 3.0e-310], [np.inf, 0.1, -0.0]])
    fn = os.path.join(str(tmp_path), F.binary_matrix_name("phi", 2, 3, 7))
    assert fn.endswith("phi_2_3_00007.BINARY")                            # LDAUtils.java:1127-1129
    F.write_binary_double_matrix(m, fn)
    raw = open(fn, "rb").read()
    assert raw == b"".join(struct.pack(">d", v) for v in m.ravel())       # ByteBuffer.putDouble: big-endian
    assert np.array_equal(F.read_binary_double_matrix(2, 3, fn).view(np.int64), m.view(np.int64))


def test_binary_int_matrix_keeps_the_reference_file_size(tmp_path):
    m = np.array([[1, -2, 3], [2**31 - 1, -2**31, 0]], np.int32)
    fn = os.path.join(str(tmp_path), F.binary_matrix_name("N", 2, 3, 12345))
    assert fn.endswith("N_2_3_12345.BINARY")
    F.write_binary_int_matrix(m, fn)
    raw = open(fn, "rb").read()
    assert len(raw) == 8 * m.size                                         # mapped at 8*rows*cols (LDAUtils.java:1164)
    assert raw[:4 * m.size] == b"".join(struct.pack(">i", int(v)) for v in m.ravel())
    assert raw[4 * m.size:] == b"\0" * (4 * m.size)
    assert np.array_equal(F.read_binary_int_matrix(2, 3, fn), m)          # readInt x rows*cols, the rest ignored


def test_subset_writers_follow_the_reference_tests(tmp_path):
    """LDAUtilsTest.testWriteDoubleMatrixRows / IntMatrixRows / DobuleMatrixIndices(Explicit) / DobuleMatrixCols / IntMatrixCols
    (LDAUtilsTest.java:37-216): a 3x3 matrix 1..9, a selection written, the file read back value by value from its head; the
    file itself is named and sized for the whole matrix (8*rows*cols bytes: FileChannel.map), zeros behind the selection."""
    m = np.arange(1.0, 10.0).reshape(3, 3)
    im = np.arange(1, 10, dtype=np.int32).reshape(3, 3)
    fn = F.write_binary_double_matrix_rows(m, 1, 3, 3, str(tmp_path / "a"), [0, 2])
    assert fn.endswith("a_3_3_00001.BINARY") and os.path.getsize(fn) == 72
    assert np.fromfile(fn, ">f8").tolist() == [1, 2, 3, 7, 8, 9, 0, 0, 0]
    fn = F.write_binary_int_matrix_rows(im, 1, 3, 3, str(tmp_path / "b"), [0, 2])
    assert os.path.getsize(fn) == 72 and np.fromfile(fn, ">i4").tolist() == [1, 2, 3, 7, 8, 9] + [0] * 12
    fn = F.write_binary_double_matrix_cols(m, 1, 3, 3, str(tmp_path / "c"), [0, 2])
    assert np.fromfile(fn, ">f8").tolist() == [1, 3, 4, 6, 7, 9, 0, 0, 0]
    fn = F.write_binary_int_matrix_cols(im, 1, 3, 3, str(tmp_path / "d"), [0, 2])
    assert np.fromfile(fn, ">i4").tolist() == [1, 3, 4, 6, 7, 9] + [0] * 12
    idx = [[1, 2], [0, 1], [0, 2]]                                     # testWriteDobuleMatrixIndices: per-row column lists
    fn = F.write_binary_double_matrix_indices(m, 1, str(tmp_path / "e"), idx)
    assert fn.endswith("e_3_2_00001.BINARY") and os.path.getsize(fn) == 3 * 2 * 8          # assertEquals(indices.length*indices[0].length*8, ttmp.length())
    assert np.fromfile(fn, ">f8").tolist() == [2, 3, 4, 5, 7, 9]
    fn = F.write_binary_double_matrix_indices(m, 1, str(tmp_path / "f"), idx, rows=3, cols=2)      # ...Explicit: the caller passes the same dimensions
    assert fn.endswith("f_3_2_00001.BINARY") and os.path.getsize(fn) == 48 and np.fromfile(fn, ">f8").tolist() == [2, 3, 4, 5, 7, 9]


def test_topic_indicator_and_int_csv(tmp_path):
    doc_ptr = np.array([0, 3, 3, 5], np.int64)
    z = np.array([4, 0, 11, 2, 2], np.int32)
    fn = F.write_topic_indicators(doc_ptr, z, str(tmp_path), 42)
    assert os.path.basename(fn) == "z_42.csv"
    assert open(fn, newline="").read() == "4,0,11" + os.linesep + os.linesep + "2,2" + os.linesep
    fn2 = os.path.join(str(tmp_path), "m.csv")
    F.write_ascii_int_matrix(np.array([[1, 2], [3, -4]]), fn2, ";")
    assert open(fn2, newline="").read() == "1;2" + os.linesep + "3;-4" + os.linesep


def test_double_to_string_rules():
    """java.lang.Double.toString as documented: at least one digit on either side of the point, plain notation for
    1e-3 <= |d| < 1e7, computerized scientific notation otherwise."""
    cases = [(1.0, "1.0"), (100.0, "100.0"), (0.001, "0.001"), (9999999.0, "9999999.0"), (1.0e7, "1.0E7"), (1.0e-4, "1.0E-4"),
             (123456789.0, "1.23456789E8"), (-13123222.510316258, "-1.3123222510316258E7"), (0.5, "0.5"), (1.0e-5, "1.0E-5"),
             (4.9e-324, "4.9E-324"), (1.7976931348623157e308, "1.7976931348623157E308"), (-0.0, "-0.0"), (0.0, "0.0"),
             (float("nan"), "NaN"), (float("inf"), "Infinity"), (float("-inf"), "-Infinity"), (1234.5678, "1234.5678"),
             (1e22, "1.0E22"), (2e-3, "0.002")]
    for d, want in cases:
        assert F.java_double_to_string(d) == want, (d, F.java_double_to_string(d), want)


def test_format_double_rules():
    """LDAUtils.formatDouble: "%.4f" (HALF_UP on the shortest decimal digits) from 1e-4 up, DecimalFormat("00.###E0")
    (two integer digits, up to three fraction digits, HALF_EVEN) below."""
    assert F.format_double(0.5) == "0.5000"
    assert F.format_double(0.0) == "0.0000"
    assert F.format_double(0.0001) == "0.0001"
    assert F.format_double(0.00015) == "0.0002"                           # shortest digits "1.5E-4": a tie, HALF_UP
    assert F.format_double(0.12345) == "0.1235"                           # the binary value is below the tie; Java rounds the digits
    assert F.format_double(1.0) == "1.0000"
    assert F.format_double(-2.00005) == "-2.0001"
    assert F.format_double(0.00005) == "50E-6"
    assert F.format_double(1.2345e-5) == "12.345E-6"
    assert F.format_double(1.23456e-5) == "12.346E-6"
    assert F.format_double(-9.9e-5) == "-99E-6"
    assert F.format_double(9.99996e-5) == "10E-5"                         # 99.9996 rounds up to three integer digits: renormalised
    assert F.format_double(3e-300) == "30E-301"
    assert F.java_format_fixed(-13123893.66552341, 6) == "-13123893.665523"


def test_text_logs_and_double_csv(tmp_path):
    d = str(tmp_path)
    F.append_log_likelihood(d, 10, -13123222.510316258)
    F.append_log_likelihood(d, 20, -0.25)
    assert open(os.path.join(d, "log-likelihood.txt"), newline="").read() == \
        "10\t-1.3123222510316258E7" + os.linesep + "20\t-0.25" + os.linesep
    F.append_heldout_log_likelihood(d, 10, -8544.048647764415)
    assert open(os.path.join(d, "test_held_out_log_likelihood.txt"), newline="").read() == "10\t-8544.048647764415" + os.linesep
    F.append_log_posterior(d, 3, -1234.5678915, 1700000000000)
    assert open(os.path.join(d, "log-posterior.txt"), newline="").read() == "3\t-1234.567892\t1700000000000" + os.linesep
    fn = F.ascii_matrix_name(d, "Phi_KxV", 2, 3, 5)
    assert os.path.basename(fn) == "Phi_KxV_2_3_00005.csv"
    F.write_ascii_double_matrix(np.array([[0.5, 0.25, 0.25], [0.99995, 2e-5, 3e-5]]), fn)
    assert open(fn, newline="").read() == "0.5000,0.2500,0.2500" + os.linesep + "1.0000,20E-6,30E-6" + os.linesep


# ---------------------------------------------------------------- the C++ writers (include/ggs_formats.hpp) against the Python ones
@pytest.fixture(scope="module")
def formats_demo(tmp_path_factory):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path_factory.mktemp("fmt") / "ggs_formats_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "ggs_formats_demo.cpp"), "-o", exe])
    return exe


def test_cpp_text_rules_equal_the_python_ones(formats_demo):
    """Double.toString, %.4f, %.6f and LDAUtils.formatDouble from C++ on 6 000 values: log-likelihood-sized numbers, probabilities,
    magnitudes around the 1e-4 / 1e-3 / 1e7 switches, exact decimal ties, denormals, infinities, both zeros."""
    import subprocess
    rng = np.random.default_rng(5)
    vals = np.concatenate([
        -np.exp(rng.uniform(0, 25, 1500)), rng.random(1500), np.exp(rng.uniform(-30, -5, 1000)), rng.normal(0, 1e7, 500),
        np.exp(rng.uniform(np.log(5e-5), np.log(2e-4), 300)), np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), 300)),
        np.exp(rng.uniform(np.log(5e6), np.log(2e7), 300)), (rng.integers(0, 100000, 300) + 0.5) / 10000.0, -(rng.integers(0, 100000, 200) + 0.5) / 1e6,
        [0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 0.00005, 0.99995, 9.99995, 99.99995, 1e-4, 1e-3, 1e7, 9999999.999999998, 1e22, 1e23, 4.9e-324, 2.2250738585072014e-308,
         1.7976931348623157e308, float("inf"), float("-inf"), float("nan"), 123456789.0, 0.001, 0.0009999999999999998, -7.0e-5, 12345.678949999999]])
    inp = "\n".join("%016x" % int(b) for b in vals.view(np.uint64)) + "\n"
    out = subprocess.run([formats_demo, "text"], input=inp, capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == vals.size
    for v, line in zip(vals.tolist(), out):
        ts, f4, f6, fd = line.split("\t")
        assert ts == F.java_double_to_string(v), (v, ts)
        assert f4 == F.java_format_fixed(v, 4), (v, f4)
        assert f6 == F.java_format_fixed(v, 6), (v, f6)
        if v == v and v not in (0.0, float("inf"), float("-inf")):
            assert fd == F.format_double(v), (v, fd)


def test_cpp_files_equal_the_python_files(formats_demo, tmp_path):
    import subprocess
    cdir, pdir = tmp_path / "c", tmp_path / "p"
    cdir.mkdir()
    pdir.mkdir()
    subprocess.check_call([formats_demo, "files", str(cdir)])
    im = np.array([[1, -2, 300000], [0, 2147483647, -2147483648]], np.int32)
    dm = np.array([[0.25, -1.5e-7], [3.14159265358979, 0.0], [1e300, 4.9e-324]])
    F.write_binary_int_matrix(im, str(pdir / F.binary_matrix_name("N", 2, 3, 12)))
    F.write_binary_double_matrix(dm, str(pdir / F.binary_matrix_name("phi", 3, 2, 12)))
    F.write_ascii_int_matrix(im, str(pdir / "ints.csv"))
    F.write_ascii_double_matrix(dm, F.ascii_matrix_name(str(pdir), "Phi_KxV", 3, 2, 12))
    F.write_topic_indicators(np.array([0, 2, 2, 5]), np.array([4, 0, 1, 1, 9]), str(pdir), 7)
    F.append_log_likelihood(str(pdir), 3, -123456.789)
    F.append_heldout_log_likelihood(str(pdir), 3, -1.0e-5)
    F.append_log_posterior(str(pdir), 3, -98765.4321987, 1700000000000)
    big, ibig = np.arange(1.0, 10.0).reshape(3, 3), np.arange(1, 10, dtype=np.int32).reshape(3, 3)
    F.write_binary_double_matrix_rows(big, 1, 3, 3, str(pdir / "drows"), [0, 2])
    F.write_binary_int_matrix_rows(ibig, 1, 3, 3, str(pdir / "irows"), [0, 2])
    F.write_binary_double_matrix_cols(big, 1, 3, 3, str(pdir / "dcols"), [0, 2])
    F.write_binary_int_matrix_cols(ibig, 1, 3, 3, str(pdir / "icols"), [2, 1])
    F.write_binary_double_matrix_indices(big, 1, str(pdir / "dsel"), [[0, 2], [1, 2], [0, 1]])
    names = sorted(os.listdir(pdir))
    assert names == sorted(os.listdir(cdir)) and len(names) == 13
    for n in names:
        assert (cdir / n).read_bytes() == (pdir / n).read_bytes(), n
