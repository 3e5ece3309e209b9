"""The number conversions of the text ingest (minnow_amd/csrc/mnw_text_num.h: strconv.Atoi, strconv.ParseFloat restated for
the device) compiled for the HOST and compared with Python's int() / float() -- exact and correctly rounded, like the Go
functions, hence the same values.  Covers the general byte-wise forms and the register forms the kernel's fast path uses:
whenever a register form returns a value it must be the right one, and whenever it gives up the general form decides."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "text_num_harness.cpp")
HDR = os.path.join(ROOT, "minnow_amd", "csrc", "mnw_text_num.h")
LIB = os.path.join(ROOT, "tests", "cpp", "libtext_num_harness.so")


@pytest.fixture(scope="module")
def tn():
    if os.environ.get("MNW_TEXT_NUM_LIB"):                  # another build of the same source (tests/test_sanitizers_cpu.py)
        L = C.CDLL(os.environ["MNW_TEXT_NUM_LIB"])
    else:
        if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
            subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", SRC, "-o", LIB])
        L = C.CDLL(LIB)
    for f in ("tn_atoi", "tn_fast_atoi"):
        getattr(L, f).argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int64)]
    for f in ("tn_atof", "tn_fast_atof"):
        getattr(L, f).argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_double)]
    return L


def _go_float_ok(s):
    """does strconv.ParseFloat accept s as a plain decimal literal (the subset the device decides itself)"""
    try:
        float(s)
    except ValueError:
        return False
    return "_" not in s and s.strip() == s and not any(c in s.lower() for c in "xinfa")


def _check_float(tn, s):
    b = s.encode()
    x = C.c_double()
    rc = tn.tn_atof(b, len(b), C.byref(x))
    if rc == 0:
        assert _go_float_ok(s), s
        want = float(s)
        assert x.value == want and math.copysign(1, x.value) == math.copysign(1, want), (s, x.value, want)
    elif rc == 2:
        assert not _go_float_ok(s), s                      # a syntax error only where Go reports one
    if len(b) <= 16:
        y = C.c_double()
        frc = tn.tn_fast_atof(b, len(b), C.byref(y))
        if frc == 0:
            assert rc == 0 and y.value == x.value and math.copysign(1, y.value) == math.copysign(1, x.value), (s, y.value, x.value)
    return rc


def _check_int(tn, s):
    b = s.encode()
    x = C.c_int64()
    ok = tn.tn_atoi(b, len(b), C.byref(x))
    try:
        want = int(s) if ("_" not in s and s.strip() == s) else None
    except ValueError:
        want = None
    if want is not None and not -2**63 <= want < 2**63:
        want = None
    assert bool(ok) == (want is not None), s
    if ok:
        assert x.value == want, s
    if len(b) <= 16:
        y = C.c_int64()
        if tn.tn_fast_atoi(b, len(b), C.byref(y)):
            assert ok and y.value == x.value, s


def test_float_literals(tn):
    fixed = ["0", "-0", "0.0", "1", "-1.5", "+2.5", "1e22", "1e23", "9007199254740993", "9007199254740992", "0.1", "0.3", ".5", "5.", "1E5", "1e+5",
             "1e-5", "123456.789e3", "0.00001e-20", "4.929148e+11", "190.24664", "-493172645126", "1.7976931348623157e308", "4.9e-324", "1e-45",
             "1e39", "16777217", "0.000000000000001", "1234567890123456", "12345678901234567", "00000000000001.5", "1e0000000000005",
             "", "+", "-", ".", "e5", ".e5", "1e", "1e+", "1e+-5", "1..2", "1.2.3", "--1", "1-", "1e5.5", "abc", "1x", "0x10", "inf", "nan", "1_000", " 1", "1 "]
    for s in fixed:
        _check_float(tn, s)
    rng = np.random.default_rng(17)
    n_value = 0
    for fmt, scale in (("%.4e", 1e12), ("%.5f", 250.0), ("%.2f", 3000.0), ("%g", 300.0), ("%.17g", 1.0), ("%.10e", 1e-10), ("%.6e", 1e30), ("%.3f", 1e9)):
        for v in rng.standard_normal(4000) * scale:
            n_value += _check_float(tn, fmt % v) == 0
    for _ in range(4000):                                  # every mantissa / exponent the fast path covers, and its edges
        m = int(rng.integers(0, 2**53 + 5))
        e = int(rng.integers(-24, 25))
        n_value += _check_float(tn, "%de%d" % (m, e)) == 0
        s = str(m)
        k = int(rng.integers(0, len(s) + 1))
        n_value += _check_float(tn, s[:k] + "." + s[k:]) == 0
    assert n_value > 30000
    # the register form must take the tokens catalogues are made of itself (giving up is correct but slow)
    for fmt, scale in (("%.4e", 1e12), ("%.5f", 250.0), ("%.2f", 3000.0), ("%.6e", 1e-5), ("%d", 1e6), ("%.3E", 1e8)):
        for v in rng.standard_normal(500) * scale:
            tok = (fmt % v).encode()
            y = C.c_double()
            assert len(tok) <= 16 and tn.tn_fast_atof(tok, len(tok), C.byref(y)) == 0 and y.value == float(tok), tok
    for tok in (b".5", b"5.", b"-.5e-3", b"+1E+05", b"0", b"-0.0", b"00012.50", b"1e22", b"9007199254740991"):
        y = C.c_double()
        assert tn.tn_fast_atof(tok, len(tok), C.byref(y)) == 0 and y.value == float(tok), tok


def test_int_literals(tn):
    for s in ["0", "-0", "+0", "1", "-1", "+17", "9223372036854775807", "9223372036854775808", "-9223372036854775808", "-9223372036854775809",
              "99999999999999999999", "0000000000000000000001", "1234567890123456", "12345678901234567", "-493172645126",
              "", "+", "-", "1.5", "1e5", "--1", "1-", "12a", "a12", " 1", "1 ", "1_000", "0x10"]:
        _check_int(tn, s)
    rng = np.random.default_rng(3)
    for bits in range(1, 64):
        for v in rng.integers(-2**bits, 2**bits, 50):
            _check_int(tn, str(int(v)))


def test_float_literals_beyond_clinger(tn):
    """17-19 significant digits and exponents beyond 10^+-22 -- numpy.savetxt's default '%.18e', repr() of a float64, physical
    constants -- are converted by the 128-bit power-of-ten table (text_scale10): when the function returns a value it is the
    correctly rounded one (Python's float() is); it may only give up (rc 1: left to the host) on a literal whose value lies
    within 2^-64 (relative) of a rounding boundary, which a random literal practically never does."""
    rng = np.random.default_rng(29)
    decided = total = 0

    def chk(s):
        nonlocal decided, total
        total += 1
        rc = _check_float(tn, s)
        assert rc in (0, 1), s
        decided += rc == 0

    for s in ("1.000000000000000000e+00", "2.500000000000000000e+02", "0.30000000000000004", "1.7976931348623157e308", "2.2250738585072014e-308",
              "9007199254740993", "9007199254740993e0", "9.007199254740993e15", "18446744073709551615", "1e23", "8.5e-310", "1e-320", "1e309",
              "6.02214076e23", "6.62607015e-34", "1.380649e-23", "9999999999999999999", "1000000000000000000e-18", "123456789012345678e4",
              "4.35e300", "7.2e-290", "1e308", "1e-307", "0.000000000000000000000000000001", "3.141592653589793238", "2.718281828459045235e0"):
        chk(s)
    # numpy.savetxt's default format, and repr()
    for scale in (1.0, 1e12, 1e-9, 1e150, 1e-150, 250.0):
        for v in rng.standard_normal(6000) * scale:
            chk("%.18e" % v)
            chk(repr(float(v)))
            chk("%.16e" % v)
    # random 17..19-digit mantissas against every exponent the table covers
    for _ in range(60000):
        nd = int(rng.integers(17, 20))
        m = int(rng.integers(10 ** (nd - 1), min(10 ** nd, 2 ** 64 - 1), dtype=np.uint64))
        e = int(rng.integers(-307 - nd + 1, 308 - nd + 1))          # value in [1e-307, 1e308): float64's normal range
        chk("%de%d" % (m, e))
    # short mantissas with large exponents
    for _ in range(20000):
        m = int(rng.integers(1, 10 ** 6))
        e = int(rng.integers(-307, 302))
        chk("%de%d" % (m, e))
    # as close to a rounding boundary as 19 digits get: the decimal expansion of the midpoint of two neighbouring
    # float64 values, cut to 19 digits (just below the midpoint) and the same plus one unit in the last digit (just above)
    from decimal import Decimal, getcontext
    getcontext().prec = 800
    for v in np.abs(rng.standard_normal(4000)) * 10.0 ** rng.integers(-300, 300, size=4000):
        v = float(v)
        if not (1e-300 < v < 1e300):
            continue
        mid = (Decimal(v) + Decimal(float(np.nextafter(v, np.inf)))) / 2
        sign, digits, exp = mid.as_tuple()
        m19 = int("".join(map(str, digits[:19])))
        e19 = exp + len(digits) - 19
        chk("%de%d" % (m19, e19))
        chk("%de%d" % (m19 + 1, e19))
    assert decided > 0.9995 * total, (decided, total)
    # subnormal results, zero and overflow are left to the host (rc 1), never decided wrongly
    for s in ("1e-320", "8.5e-310", "4.9e-324", "1e309", "2e308", "123456789012345678e-340", "1e-400", "1e400"):
        b = s.encode()
        x = C.c_double()
        assert tn.tn_atof(b, len(b), C.byref(x)) == 1, s
