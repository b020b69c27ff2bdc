"""On-disk formats the Java driver writes, so a run driven from Python can leave the same files.

Byte-exact by construction (integers and raw IEEE bits, nothing to round):
  * ``*_<rows>_<cols>_%05d.BINARY`` matrices        util/LDAUtils.java:1120-1173  (ByteBuffer.putDouble / putInt: big-endian)
  * ``readBinaryIntMatrix`` / ``readBinaryDoubleMatrix``        LDAUtils.java:1255-1267,1330-  (DataInputStream: big-endian)
  * ASCII integer matrices                           LDAUtils.java:1175-1197
  * ``z_<iteration>.csv``                            topics/UncollapsedParallelLDA.java:945-968

Restated from the JDK's documented behaviour and NOT checkable here (no JVM in the image; the reference holds no
output file) -- parity unpinned:
  * ``Double.toString`` (the ``iteration \\t logLik`` lines, LDAUtils.java:928-940,971-979): shortest digits that
    round-trip, plain notation for 1e-3 <= |d| < 1e7, otherwise ``d.dddE[-]n``.  (JDKs before 19 print a longer digit
    string for a few values, JDK-4511638.)
  * ``String.format("%.nf")`` (LDAUtils.formatDouble, LDAUtils.java:1199-1207; log-posterior.txt, :955-968): HALF_UP on
    the shortest decimal digits (java.util.Formatter works from FormattedFloatingDecimal, not from the binary value).
  * ``new DecimalFormat("00.###E0")`` for 0 < |d| < 1e-4 (LDAUtils.java:1200-1201,1225): exactly two integer digits, at
    most three fraction digits, HALF_EVEN on the exact binary value, exponent without a plus sign.
"""
import os
from decimal import ROUND_HALF_EVEN, ROUND_HALF_UP, Decimal, localcontext

import numpy as np


# ---------------------------------------------------------------- binary matrices
def binary_matrix_name(prefix, rows, cols, iteration):
    """LDAUtils.java:1127-1129,1154-1156: ``String.format(filename + "_" + rows + "_" + columns + "_%05d.BINARY", iteration)``."""
    return "%s_%d_%d_%05d.BINARY" % (prefix, rows, cols, iteration)


def write_binary_double_matrix(matrix, fn):
    """LDAUtils.java:1132-1143: rows*cols big-endian doubles, row-major, in a file of 8*rows*cols bytes."""
    m = np.ascontiguousarray(matrix, np.float64)
    with open(fn, "wb") as f:
        f.write(m.astype(">f8").tobytes())


def write_binary_int_matrix(matrix, fn):
    """LDAUtils.java:1161-1173: rows*cols big-endian ints -- in a file MAPPED at 8*rows*cols bytes (``bufferSize =
    8*columns*rows`` is the double writer's size), so the second half of the file is zeros.  Reproduced as is."""
    m = np.ascontiguousarray(matrix, np.int32)
    with open(fn, "wb") as f:
        f.write(m.astype(">i4").tobytes())
        f.write(b"\0" * (4 * m.size))


def _write_mapped(fn, payload, mapped_bytes):
    """RandomAccessFile + FileChannel.map(READ_WRITE, 0, bufferSize): the file is bufferSize bytes long whatever is put."""
    with open(fn, "wb") as f:
        f.write(payload)
        if mapped_bytes > len(payload):
            f.write(b"\0" * (mapped_bytes - len(payload)))


def write_binary_double_matrix_rows(matrix, iteration, rows, cols, prefix, row_indices):
    """LDAUtils.writeBinaryDoubleMatrixRows (LDAUtils.java:1037-1051): the listed rows, in the listed order, at the head of
    a file named and MAPPED for the whole rows x cols matrix (the tail stays zero).  Returns the file name."""
    m = np.asarray(matrix, np.float64)
    fn = binary_matrix_name(prefix, rows, cols, iteration)
    _write_mapped(fn, np.ascontiguousarray(m[np.asarray(row_indices, np.int64), :cols]).astype(">f8").tobytes(), 8 * rows * cols)
    return fn


def write_binary_int_matrix_rows(matrix, iteration, rows, cols, prefix, row_indices):
    """LDAUtils.writeBinaryIntMatrixRows (LDAUtils.java:1053-1067); mapped at 8*rows*cols bytes like every writer here."""
    m = np.asarray(matrix, np.int32)
    fn = binary_matrix_name(prefix, rows, cols, iteration)
    _write_mapped(fn, np.ascontiguousarray(m[np.asarray(row_indices, np.int64), :cols]).astype(">i4").tobytes(), 8 * rows * cols)
    return fn


def write_binary_double_matrix_cols(matrix, iteration, rows, cols, prefix, col_indices):
    """LDAUtils.writeBinaryDoubleMatrixCols (LDAUtils.java:1069-1083): every row, the listed columns in the listed order."""
    m = np.asarray(matrix, np.float64)
    fn = binary_matrix_name(prefix, rows, cols, iteration)
    _write_mapped(fn, np.ascontiguousarray(m[:rows][:, np.asarray(col_indices, np.int64)]).astype(">f8").tobytes(), 8 * rows * cols)
    return fn


def write_binary_int_matrix_cols(matrix, iteration, rows, cols, prefix, col_indices):
    """LDAUtils.writeBinaryIntMatrixCols (LDAUtils.java:1108-1122)."""
    m = np.asarray(matrix, np.int32)
    fn = binary_matrix_name(prefix, rows, cols, iteration)
    _write_mapped(fn, np.ascontiguousarray(m[:rows][:, np.asarray(col_indices, np.int64)]).astype(">i4").tobytes(), 8 * rows * cols)
    return fn


def write_binary_double_matrix_indices(matrix, iteration, prefix, indices, rows=None, cols=None):
    """LDAUtils.writeBinaryDoubleMatrixIndices (LDAUtils.java:1085-1106): row r contributes matrix[r][indices[r][j]] for
    every j -- the driver's Selected_Phi_KxV file of each topic's top words (UPLDA:888).  The name carries
    indices.length x indices[0].length unless the caller gives the dimensions; returns the file name."""
    m = np.asarray(matrix, np.float64)
    rows = len(indices) if rows is None else rows
    cols = len(indices[0]) if cols is None else cols
    fn = binary_matrix_name(prefix, rows, cols, iteration)
    payload = b"".join(np.ascontiguousarray(m[r, np.asarray(idx, np.int64)]).astype(">f8").tobytes() for r, idx in enumerate(indices))
    _write_mapped(fn, payload, 8 * rows * cols)
    return fn


def read_binary_int_matrix(rows, cols, fn):
    """LDAUtils.java:1255-1267 (DataInputStream.readInt: big-endian; trailing bytes ignored)."""
    return np.fromfile(fn, dtype=">i4", count=rows * cols).astype(np.int32).reshape(rows, cols)


def read_binary_double_matrix(rows, cols, fn):
    return np.fromfile(fn, dtype=">f8", count=rows * cols).astype(np.float64).reshape(rows, cols)


# ---------------------------------------------------------------- integer text
def write_ascii_int_matrix(matrix, fn, sep=","):
    """LDAUtils.java:1175-1197: values joined by `sep`, one row per line (PrintWriter.println: the platform separator)."""
    with open(fn, "w", newline="") as f:
        for row in np.asarray(matrix):
            f.write(sep.join(str(int(v)) for v in row) + os.linesep)


def write_topic_indicators(doc_ptr, z, log_dir, iteration):
    """UPLDA:945-968 logTopicIndicators: ``z_<iteration>.csv``, one document per line, an empty line for an empty one."""
    fn = os.path.join(log_dir, "z_%d.csv" % iteration)
    z = np.asarray(z)
    with open(fn, "w", newline="") as f:
        for d in range(len(doc_ptr) - 1):
            f.write(",".join(str(int(t)) for t in z[doc_ptr[d]:doc_ptr[d + 1]]) + os.linesep)
    return fn


# ---------------------------------------------------------------- doubles as text (parity unpinned, see the module docstring)
def _shortest_digits(d):
    """(sign, digits, exponent) with |d| = 0.d1d2... x 10^exponent, digits = the shortest that round-trip (repr)."""
    sign, digits, exp = Decimal(repr(float(d))).as_tuple()
    digits = list(digits)
    while len(digits) > 1 and digits[-1] == 0:
        digits.pop()
        exp += 1
    if len(digits) == 1:
        # Java renders at least two digits and, among the two-digit decimals that round to d, takes the one closest to
        # d (Double.MIN_VALUE prints as 4.9E-324, not 5.0E-324)
        exact = Decimal(float(d))
        two = exact.quantize(Decimal(1).scaleb(exact.adjusted() - 1), rounding=ROUND_HALF_EVEN)
        if float(two) == float(d):
            _, digits2, exp2 = two.as_tuple()
            digits2 = list(digits2)
            while len(digits2) > 1 and digits2[-1] == 0:
                digits2.pop()
                exp2 += 1
            digits, exp = digits2, exp2
    return sign, digits, exp + len(digits)


def java_double_to_string(d):
    """java.lang.Double.toString."""
    d = float(d)
    if d != d:
        return "NaN"
    if d in (float("inf"), float("-inf")):
        return "Infinity" if d > 0 else "-Infinity"
    if d == 0:
        return "-0.0" if str(d).startswith("-") else "0.0"
    sign, digits, e10 = _shortest_digits(d)            # 0.DIGITS x 10^e10
    s = "".join(map(str, digits))
    if 1e-3 <= abs(d) < 1e7:
        if e10 <= 0:
            body = "0." + "0" * (-e10) + s
        elif e10 >= len(s):
            body = s + "0" * (e10 - len(s)) + ".0"
        else:
            body = s[:e10] + "." + s[e10:]
    else:
        body = s[0] + "." + (s[1:] or "0") + "E" + str(e10 - 1)
    return ("-" if sign else "") + body


def java_format_fixed(d, digits):
    """String.format("%.<digits>f", d): HALF_UP on the shortest decimal digits."""
    d = float(d)
    if d != d:
        return "NaN"
    if d in (float("inf"), float("-inf")):
        return "Infinity" if d > 0 else "-Infinity"
    with localcontext() as ctx:
        ctx.prec = 400                                 # 1e300 with six decimals has 307 digits
        q = Decimal(repr(d)).quantize(Decimal(1).scaleb(-digits), rounding=ROUND_HALF_UP)
    out = format(q, "f")
    if q == 0 and str(d).startswith("-"):
        out = "-" + out.lstrip("-")                    # Java keeps the sign of a negative value that rounds to zero
    return out


def java_decimal_format_00_3e0(d):
    """new DecimalFormat("00.###E0").format(d) for a finite non-zero d."""
    x = Decimal(float(d))                              # the exact binary value
    sign = "-" if x < 0 else ""
    x = abs(x)
    e = x.adjusted() - 1                               # mantissa with two integer digits
    m = (x.scaleb(-e)).quantize(Decimal("0.001"), rounding=ROUND_HALF_EVEN)
    if m >= 100:
        e += 1
        m = (x.scaleb(-e)).quantize(Decimal("0.001"), rounding=ROUND_HALF_EVEN)
    text = format(m, "f")
    if "." in text:
        text = text.rstrip("0").rstrip(".")
    return "%s%sE%d" % (sign, text, e)


def format_double(d, digits=4):
    """LDAUtils.formatDouble (LDAUtils.java:1199-1207): scientific below 1e-4 in magnitude (and not zero), else %.<digits>f."""
    d = float(d)
    if (0 < d < 0.0001) or (-0.0001 < d < 0):
        return java_decimal_format_00_3e0(d)
    return java_format_fixed(d, digits)


def write_ascii_double_matrix(matrix, fn, sep=",", digits=4):
    """LDAUtils.java:1222-1249 (Phi_KxV_*.csv, Theta_DxK_*.csv; UPLDA:757-817)."""
    with open(fn, "w", newline="") as f:
        for row in np.asarray(matrix, np.float64):
            f.write(sep.join(format_double(v, digits) for v in row) + os.linesep)


def ascii_matrix_name(out_dir, kind, rows, cols, iteration):
    """UPLDA:757-762,805-811: ``Theta_DxK_<n>_<K>_%05d.csv`` / ``Phi_KxV_<K>_<V>_%05d.csv``."""
    return os.path.join(out_dir, "%s_%d_%d_%05d.csv" % (kind, rows, cols, iteration))


def append_log_likelihood(log_dir, iteration, log_lik):
    """LDAUtils.logLikelihoodToFile(LogState) (LDAUtils.java:971-979): ``iteration \\t logLik``, appended."""
    with open(os.path.join(log_dir, "log-likelihood.txt"), "a", newline="") as f:
        f.write("%d\t%s%s" % (iteration, java_double_to_string(log_lik), os.linesep))


def append_heldout_log_likelihood(log_dir, iteration, value):
    """LDAUtils.heldOutLLToFile (LDAUtils.java:928-940)."""
    with open(os.path.join(log_dir, "test_held_out_log_likelihood.txt"), "a", newline="") as f:
        f.write("%d\t%s%s" % (iteration, java_double_to_string(value), os.linesep))


def append_log_posterior(log_dir, iteration, value, millis):
    """LDAUtils.logPosteriorToFile (LDAUtils.java:955-968): ``%d\\t%.6f\\t%d%n`` with System.currentTimeMillis()."""
    with open(os.path.join(log_dir, "log-posterior.txt"), "a", newline="") as f:
        f.write("%d\t%s\t%d%s" % (iteration, java_format_fixed(value, 6), millis, os.linesep))
