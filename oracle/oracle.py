"""ctypes front end of the CPU oracle (oracle/mnw_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package (minnow_amd) never imports it.

Each wrapper names the reference function it restates (see mnw_oracle.h for file:line).
"""
import ctypes as C
import os
import subprocess

import math

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmnw_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mnw_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libmnw_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        # MNW_ORACLE_LIB: another build of the same source (tests/test_sanitizers_cpu.py runs the golden tests against an
        # AddressSanitizer + UBSan build)
        path = os.environ.get("MNW_ORACLE_LIB") or _LIB_PATH
        if path == _LIB_PATH and not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(path)
        i64, u64, p, f32, dbl, ci = C.c_int64, C.c_uint64, C.c_void_p, C.c_float, C.c_double, C.c_int
        sigs = {
            "mnwo_precision_needed": (i64, [u64]),
            "mnwo_go_log2": (dbl, [dbl]),
            "mnwo_array_bytes": (i64, [i64, i64]),
            "mnwo_pack": (ci, [ci, p, i64, p, i64]),
            "mnwo_unpack": (ci, [ci, p, i64, p]),
            "mnwo_pack_fast": (ci, [ci, p, i64, p, i64]),
            "mnwo_unpack_fast": (ci, [ci, p, i64, p]),
            "mnwo_bits": (i64, [p, i64]),
            "mnwo_int64_min": (i64, [p, i64]),
            "mnwo_bound": (None, [p, i64, i64, i64]),
            "mnwo_periodic_min": (i64, [p, i64, i64]),
            "mnwo_periodic_distance": (i64, [i64, i64, i64]),
            "mnwo_float_group_pixels": (i64, [f32, f32, f32]),
            "mnwo_int_encode": (ci, [p, i64, p, p, p, p, p, ci]),
            "mnwo_int_decode": (ci, [p, i64, i64, i64, p, p, ci]),
            "mnwo_float_encode": (ci, [p, i64, f32, f32, i64, ci, p, p, p, p, p, p, ci]),
            "mnwo_quantize": (None, [p, i64, f32, f32, i64, p]),
            "mnwo_float_decode": (ci, [p, i64, i64, f32, f32, i64, ci, p, i64, p, p, p, ci]),
            "mnwo_float_decode_q": (ci, [p, i64, i64, i64, ci, i64, p, p, ci]),
            "mnwo_dither_u01": (dbl, [u64, u64, u64]),
            "mnwo_dither_fill": (None, [u64, u64, i64, p]),
            "mnwo_tail_array_encode": (i64, [p, i64, p]),
            "mnwo_tail_array_decode": (i64, [p, i64, p]),
            "mnwo_get_sub_cell": (None, [p, p, p, p, i64, i64, i64]),
            "mnwo_set_sub_cell": (None, [p, p, p, p, i64, i64, i64]),
            "mnwo_bounds3": (None, [p, i64, p, p]),
            "mnwo_periodic_fix": (None, [p, i64, f32]),
            "mnwo_nextafter32": (f32, [f32, f32]),
            "mnwo_process_float_group": (None, [p, i64, ci, f32, f32]),
            "mnwo_unlog": (None, [p, i64]),
            "mnwo_go_log10": (dbl, [dbl]),
            "mnwo_go_pow": (dbl, [dbl, dbl]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def precision_needed(mx):
    """bit.PrecisionNeeded"""
    return int(lib().mnwo_precision_needed(C.c_uint64(int(mx) & (2**64 - 1))))


def go_log2(x):
    return float(lib().mnwo_go_log2(float(x)))


def array_bytes(bits, length):
    """bit.ArrayBytes"""
    return int(lib().mnwo_array_bytes(int(bits), int(length)))


def pack(bits, x, literal=True):
    """bit.NewArray(bits, x).Data"""
    x = _c(x, np.uint64)
    n = x.size
    out = np.empty(array_bytes(bits, n), dtype=np.uint8)
    fn = lib().mnwo_pack if literal else lib().mnwo_pack_fast
    rc = fn(int(bits), _ptr(x), n, _ptr(out), out.size)
    if rc:
        raise ValueError("oracle pack failed rc=%d" % rc)
    return out


def unpack(bits, data, n, literal=True):
    """(*bit.Array).Slice"""
    data = _c(data, np.uint8)
    out = np.empty(n, dtype=np.uint64)
    # the literal decoder reads one byte past a 64-bit element's last byte only when that
    # byte exists in Go's array; pad so the C side never reads outside the buffer
    padded = np.concatenate([data, np.zeros(16, np.uint8)])
    fn = lib().mnwo_unpack if literal else lib().mnwo_unpack_fast
    if not literal:
        padded = data if data.size else np.zeros(1, np.uint8)
    rc = fn(int(bits), _ptr(padded), int(n), _ptr(out))
    if rc:
        raise ValueError("oracle unpack failed rc=%d" % rc)
    return out


def bits_needed(x):
    """bit.ArrayBuffer.Bits"""
    x = _c(x, np.uint64)
    return int(lib().mnwo_bits(_ptr(x), x.size))


def int64_min(x):
    x = _c(x, np.int64)
    return int(lib().mnwo_int64_min(_ptr(x), x.size))


def bound(x, mn, pixels):
    x = np.array(x, dtype=np.int64)
    lib().mnwo_bound(_ptr(x), x.size, int(mn), int(pixels))
    return x


def periodic_min(x, pixels):
    x = _c(x, np.int64)
    return int(lib().mnwo_periodic_min(_ptr(x), x.size, int(pixels)))


def float_group_pixels(low, high, dx):
    """Writer.FloatGroup's pixels derivation (writer.go:73)"""
    return int(lib().mnwo_float_group_pixels(float(np.float32(low)), float(np.float32(high)),
                                             float(np.float32(dx))))


def int_encode(x, literal=True):
    """intGroup.writeData -> (min, bits, stream bytes)"""
    x = _c(x, np.int64)
    n = x.size
    out = np.zeros(8 * n + 8, dtype=np.uint8)
    scratch = np.empty(max(n, 1), dtype=np.uint64)
    mn, bits, olen = C.c_int64(), C.c_int64(), C.c_int64()
    rc = lib().mnwo_int_encode(_ptr(x), n, C.byref(mn), C.byref(bits), _ptr(out),
                               C.byref(olen), _ptr(scratch), int(literal))
    if rc:
        raise ValueError("oracle int_encode failed rc=%d (bits=%d)" % (rc, bits.value))
    return mn.value, bits.value, out[: olen.value].copy()


def int_decode(data, mn, bits, n, literal=True):
    """intGroup.readData"""
    data = _c(data, np.uint8)
    padded = np.concatenate([data, np.zeros(16, np.uint8)])
    out = np.empty(n, dtype=np.int64)
    scratch = np.empty(max(n, 1), dtype=np.uint64)
    rc = lib().mnwo_int_decode(_ptr(padded), int(mn), int(bits), int(n), _ptr(out),
                               _ptr(scratch), int(literal))
    if rc:
        raise ValueError("oracle int_decode failed rc=%d" % rc)
    return out


def quantize(x, low, high, pixels):
    """floatGroup.writeData's quantize loop (group.go:316-320)"""
    x = _c(x, np.float32)
    q = np.empty(x.size, dtype=np.int64)
    lib().mnwo_quantize(_ptr(x), x.size, float(np.float32(low)), float(np.float32(high)),
                        int(pixels), _ptr(q))
    return q


def float_encode(x, low, high, pixels, periodic=True, literal=True):
    """floatGroup.writeData -> (min, bits, stream bytes)"""
    x = _c(x, np.float32)
    n = x.size
    out = np.zeros(8 * n + 8, dtype=np.uint8)
    qbuf = np.empty(max(n, 1), dtype=np.int64)
    scratch = np.empty(max(n, 1), dtype=np.uint64)
    mn, bits, olen = C.c_int64(), C.c_int64(), C.c_int64()
    rc = lib().mnwo_float_encode(_ptr(x), n, float(np.float32(low)), float(np.float32(high)),
                                 int(pixels), int(bool(periodic)), C.byref(mn), C.byref(bits),
                                 _ptr(out), C.byref(olen), _ptr(qbuf), _ptr(scratch),
                                 int(literal))
    if rc:
        raise ValueError("oracle float_encode failed rc=%d (bits=%d)" % (rc, bits.value))
    return mn.value, bits.value, out[: olen.value].copy()


def float_decode_q(data, mn, bits, pixels, periodic, n, literal=True):
    """integer stream of floatGroup.readData before dequantize (group.go:301-303)"""
    data = _c(data, np.uint8)
    padded = np.concatenate([data, np.zeros(16, np.uint8)])
    q = np.empty(n, dtype=np.int64)
    scratch = np.empty(max(n, 1), dtype=np.uint64)
    rc = lib().mnwo_float_decode_q(_ptr(padded), int(mn), int(bits), int(pixels),
                                   int(bool(periodic)), int(n), _ptr(q), _ptr(scratch),
                                   int(literal))
    if rc:
        raise ValueError("oracle float_decode_q failed rc=%d" % rc)
    return q


def dither(seed, block, n):
    u = np.empty(n, dtype=np.float64)
    lib().mnwo_dither_fill(C.c_uint64(seed), C.c_uint64(block), int(n), _ptr(u))
    return u


def float_decode(data, mn, bits, low, high, pixels, periodic, n, u=None, literal=True):
    """floatGroup.readData with explicit dither deviates u (None -> 0.0)"""
    data = _c(data, np.uint8)
    padded = np.concatenate([data, np.zeros(16, np.uint8)])
    out = np.empty(n, dtype=np.float32)
    qbuf = np.empty(max(n, 1), dtype=np.int64)
    scratch = np.empty(max(n, 1), dtype=np.uint64)
    up = None
    if u is not None:
        u = _c(u, np.float64)
        up = _ptr(u)
    rc = lib().mnwo_float_decode(_ptr(padded), int(mn), int(bits), float(np.float32(low)),
                                 float(np.float32(high)), int(pixels), int(bool(periodic)),
                                 up, int(n), _ptr(out), _ptr(qbuf), _ptr(scratch), int(literal))
    if rc:
        raise ValueError("oracle float_decode failed rc=%d" % rc)
    return out


def tail_array_encode(x):
    """write(x) inside intGroup.writeTail (group.go:216-225)"""
    x = _c(x, np.int64)
    out = np.zeros(16 + 8 * x.size + 8, dtype=np.uint8)
    n = lib().mnwo_tail_array_encode(_ptr(x), x.size, _ptr(out))
    if n < 0:
        raise ValueError("oracle tail encode failed")
    return out[:n].copy()


def tail_array_decode(data, blocks):
    """read() inside newIntGroupFromTail (group.go:190-198) -> (values, bytes consumed)"""
    data = _c(data, np.uint8)
    padded = np.concatenate([data, np.zeros(16, np.uint8)])
    out = np.empty(blocks, dtype=np.int64)
    n = lib().mnwo_tail_array_decode(_ptr(padded), int(blocks), _ptr(out))
    if n < 0:
        raise ValueError("oracle tail decode failed")
    return out, int(n)


# ---- minp sub-cells (go/minp/minp.go) --------------------------------------------------------------
def get_sub_cell(vec, sc, sub_cells, n_sub):
    """getSubCell (minp.go:246-266): vec is [n_file^3, 3] float32 -> three arrays of n_sub^3"""
    vec = _c(vec, np.float32)
    bufs = [np.empty(n_sub ** 3, dtype=np.float32) for _ in range(3)]
    lib().mnwo_get_sub_cell(_ptr(vec), _ptr(bufs[0]), _ptr(bufs[1]), _ptr(bufs[2]), sc, sub_cells, n_sub)
    return bufs


def set_sub_cell(vec, bufs, sc, sub_cells, n_sub):
    """setSubCell (minp.go:272-288), in place on the C-contiguous float32 array vec"""
    assert vec.dtype == np.float32 and vec.flags.c_contiguous
    b = [_c(x, np.float32) for x in bufs]
    lib().mnwo_set_sub_cell(_ptr(vec), _ptr(b[0]), _ptr(b[1]), _ptr(b[2]), sc, sub_cells, n_sub)


def bounds3(vec):
    vec = _c(vec, np.float32)
    mn, mx = np.empty(3, dtype=np.float32), np.empty(3, dtype=np.float32)
    lib().mnwo_bounds3(_ptr(vec), vec.size // 3, _ptr(mn), _ptr(mx))
    return mn, mx


def nextafter32(x, y):
    return np.float32(lib().mnwo_nextafter32(float(np.float32(x)), float(np.float32(y))))


def minp_limits(vec, periodic, L):
    """the (min, max) per component that minp.Writer.Vectors hands to FloatGroup (minp.go:86-96)"""
    if periodic:
        return np.zeros(3, dtype=np.float32), np.full(3, np.float32(L), dtype=np.float32)
    mn, mx = bounds3(vec)
    mx = np.array([nextafter32(m, np.float32(2) * m) for m in mx], dtype=np.float32)
    return mn, mx


def minp_vectors_encode(vec, n_file, sub_cells, dx, periodic, L):
    """minp.Writer.Vectors (minp.go:85-118) up to the file writes: [(min, bits, bytes)] in file order
    (k-major), plus the limits and pixels of the three groups."""
    n_sub = n_file // sub_cells
    mn, mx = minp_limits(vec, periodic, L)
    blocks, pix = [], []
    for k in range(3):
        pixels = float_group_pixels(mn[k], mx[k], dx)
        pix.append(pixels)
        for sc in range(sub_cells ** 3):
            buf = get_sub_cell(vec, sc, sub_cells, n_sub)[k]
            blocks.append(float_encode(buf, mn[k], mx[k], pixels, True, literal=False))
    return blocks, mn, mx, pix


def minp_vectors_decode(blocks, n_file, sub_cells, mn, mx, pix, periodic, L, seed, first_block=0):
    """minp.Reader.Vectors (minp.go:170-198) after the file reads, with the build's dither"""
    n_sub = n_file // sub_cells
    s3, n3 = sub_cells ** 3, n_sub ** 3
    out = np.zeros((n_file ** 3, 3), dtype=np.float32)
    for sc in range(s3):
        bufs = []
        for k in range(3):
            b = k * s3 + sc
            m, bits, data = blocks[b]
            x = float_decode(data, m, bits, mn[k], mx[k], pix[k], True, n3, u=dither(seed, first_block + b, n3),
                             literal=False)
            if periodic:
                x = np.ascontiguousarray(x, dtype=np.float32)
                lib().mnwo_periodic_fix(_ptr(x), n3, float(np.float32(L)))
            bufs.append(x)
        set_sub_cell(out, bufs, sc, sub_cells, n_sub)
    return out


# ---- minh column transforms (go/minh/minh.go:139-149, 315-319) --------------------------------------
def process_float_group(x, log, low, high):
    buf = np.array(x, dtype=np.float32, copy=True, order="C")
    lib().mnwo_process_float_group(_ptr(buf), buf.size, int(bool(log)), float(np.float32(low)), float(np.float32(high)))
    return buf


def unlog(x):
    buf = np.array(x, dtype=np.float32, copy=True, order="C")
    lib().mnwo_unlog(_ptr(buf), buf.size)
    return buf


def go_log10(x):
    return float(lib().mnwo_go_log10(float(x)))


def go_pow(x, y):
    return float(lib().mnwo_go_pow(float(x), float(y)))


# ---- text catalogue ingest (go/text/text.go:176-197, go/text/parse.go) ---------------------------------
_GO_FLOAT = None


def go_parse_float(tok):
    """strconv.ParseFloat(tok, 64) (Go 1.16, strconv/atof.go restated: special(), readFloat's grammar, underscoreOK); the value
    of a well-formed literal is the correctly rounded float64, which Python's float() / float.fromhex() return as well.
    Raises ValueError where ParseFloat returns an error (the reference panics on it).  Parity unpinned beyond the reference's
    own TestReader vector: there is no Go toolchain here."""
    import re
    global _GO_FLOAT
    if _GO_FLOAT is None:
        _GO_FLOAT = (re.compile(r"[+-]?(inf|infinity)$", re.I), re.compile(r"nan$", re.I),
                     re.compile(r"[+-]?(?=[0-9_]*[0-9]|[0-9_]*\.[0-9_]*[0-9])[0-9_]*\.?[0-9_]*([eE][+-]?[0-9][0-9_]*)?$"),
                     re.compile(r"[+-]?0[xX](?=[0-9a-fA-F_]*[0-9a-fA-F]|[0-9a-fA-F_]*\.[0-9a-fA-F_]*[0-9a-fA-F])[0-9a-fA-F_]*\.?[0-9a-fA-F_]*[pP][+-]?[0-9][0-9_]*$"))
    inf_re, nan_re, dec_re, hex_re = _GO_FLOAT
    bad = ValueError("strconv.ParseFloat: parsing %r: invalid syntax" % tok)
    if inf_re.match(tok):
        return -math.inf if tok[0] == "-" else math.inf
    if nan_re.match(tok):                                   # no sign in front of nan (special() only lets one through for inf)
        return math.nan
    is_hex = bool(hex_re.match(tok))
    if not is_hex and not dec_re.match(tok):
        raise bad
    if "_" in tok:
        # underscoreOK: an underscore only between digits, or between the base prefix and a digit
        t = tok[1:] if tok[0] in "+-" else tok
        saw, i = "^", 0
        if len(t) >= 2 and t[0] == "0" and t[1].lower() in "box":
            i, saw = 2, "0"
        digits = "0123456789abcdefABCDEF" if is_hex else "0123456789"
        for ch in t[i:]:
            if ch in digits:
                saw = "0"
            elif ch == "_":
                if saw != "0":
                    raise bad
                saw = "_"
            else:
                if saw == "_":
                    raise bad
                saw = "!"
        if saw == "_":
            raise bad
        tok = tok.replace("_", "")
    try:
        x = float.fromhex(tok) if is_hex else float(tok)
    except OverflowError:
        x = math.inf
    if math.isinf(x):                                        # ErrRange, which the reference panics on
        raise ValueError("strconv.ParseFloat: parsing %r: value out of range" % tok)
    return x


def text_block_parse(data, sep, comm, float_cols, int_cols):
    """text.Reader.Block after the read, restated: split at '\\n' (parse.go:15-36), uncomment (43-63), trim (66-78),
    fields (172-211), strconv.Atoi / float32(strconv.ParseFloat(s, 64)).  Python's int() and float() stand in for Atoi
    and ParseFloat: both conversions are exact / correctly rounded, hence unique, for well-formed decimal literals.
    Returns (float columns, int columns) as lists of numpy arrays.  Raises ValueError where the reference panics."""
    sep_b, comm_b = bytes([sep]), bytes([comm])
    lines = bytes(data).split(b"\n")
    lines = [ln.split(comm_b, 1)[0] for ln in lines]
    lines = [ln for ln in lines if ln.strip(sep_b) != b""]
    fout = [np.empty(len(lines), dtype=np.float32) for _ in float_cols]
    iout = [np.empty(len(lines), dtype=np.int64) for _ in int_cols]
    if not lines:
        return fout, iout
    buf_len = len(lines[0].split())                     # bytes.Fields: white space
    want = list(float_cols) + list(int_cols)
    if want and max(want) >= buf_len:
        raise ValueError("Data has %d columns, but column %d was requested." % (buf_len, max(want)))
    if not want:                                        # parseInt64s / parseFloat32s return at once without columns (parse.go:83, 132)
        return fout, iout
    for i, ln in enumerate(lines):
        words = [w for w in ln.split(sep_b) if w != b""]
        if len(words) != buf_len:
            raise ValueError("Data on line %d has %d columns, not %d." % (i + 1, len(words), buf_len))
        for k, c in enumerate(float_cols):
            tok = words[c].decode("ascii")
            x = go_parse_float(tok)
            with np.errstate(over="ignore"):                       # float32(x) beyond float32's range is +-Inf in Go too
                fout[k][i] = np.float32(x)
        for k, c in enumerate(int_cols):
            tok = words[c].decode("ascii")
            if "_" in tok or tok != tok.strip():                  # as above: Atoi takes neither
                raise ValueError("strconv.Atoi: parsing %r: invalid syntax" % tok)
            v = int(tok)
            if not -2**63 <= v < 2**63:
                raise ValueError("strconv.Atoi: value out of range")
            iout[k][i] = v
    return fout, iout


def generic_cut(cutoff, mass, cols):
    """GenericCut of scripts/text_to_minh.go:271-312: ok[i] = m[i] > cutoff (float32 compare, false for NaN), every
    column keeps its ok rows in order.  Returns (n kept, filtered columns)."""
    m = np.asarray(mass, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        ok = m > np.float32(cutoff)
    return int(ok.sum()), [np.asarray(x)[ok] for x in cols]
