"""Text catalogue ingest (go/text/text.go:176-197, go/text/parse.go) against the oracle's restatement:
the reference's own TestReader text (go/text/text_test.go:66-127), random Rockstar-like catalogues, decimal literals
that need the correctly rounded slow path, comments and blank lines, and the inputs the reference panics on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mnw():
    import minnow_amd
    from minnow_amd import build as b
    b.build()
    if minnow_amd.device_count() < 1:
        pytest.fail("no HIP device: GPU tests need a real MI355X (no CPU fallback exists)")
    return minnow_amd


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _parse(mnw, torch, ctx, data, sep, comm, fcols, icols, cap=None):
    dev = torch.device("cuda:0")
    cap = cap if cap is not None else data.count(b"\n") + 2
    fo = [torch.full((cap,), float("nan"), dtype=torch.float32, device=dev) for _ in fcols]
    io = [torch.full((cap,), -7, dtype=torch.int64, device=dev) for _ in icols]
    torch.cuda.synchronize()
    n = ctx.text_parse(data, sep, comm, fcols, [t.data_ptr() for t in fo], icols, [t.data_ptr() for t in io], cap)
    torch.cuda.synchronize()
    return n, [t[:n].cpu().numpy() for t in fo], [t[:n].cpu().numpy() for t in io]


def _check(mnw, torch, ctx, oracle, data, sep, comm, fcols, icols):
    n, fo, io = _parse(mnw, torch, ctx, data, sep, comm, fcols, icols)
    rf, ri = oracle.text_block_parse(data, sep, comm, fcols, icols)
    assert n == (rf[0].size if rf else ri[0].size)
    for a, b in zip(fo, rf):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True)
    for a, b in zip(io, ri):
        assert np.array_equal(a, b)
    return n


def test_reference_reader_text(mnw, oracle, torch_mod):
    """TestReader of go/text/text_test.go: columns "4", "1", "2", "3" as float32, int64, float32, int64; the two
    blocks of the reference test (block size 120) and the whole text at once"""
    text = b"""#123456789012345678
#123456789012345678
1    2     3      5
11  12    13     15
21  22    23     25
31  32    33     35
41  42    43     45
"""
    with mnw.Context(0) as c:
        n, fo, io = _parse(mnw, torch_mod, c, text, ord(" "), ord("#"), [3, 1], [0, 2])
        assert n == 5
        assert list(fo[0]) == [5, 15, 25, 35, 45] and list(fo[1]) == [2, 12, 22, 32, 42]
        assert list(io[0]) == [1, 11, 21, 31, 41] and list(io[1]) == [3, 13, 23, 33, 43]
        # the reference's blocks: findBlocks cuts at the first newline at or after byte 120 - 50 (text.go:101-127)
        cut = text.index(b"\n", 70) + 1
        n1, fo1, io1 = _parse(mnw, torch_mod, c, text[:cut], ord(" "), ord("#"), [3, 1], [0, 2])
        n2, fo2, io2 = _parse(mnw, torch_mod, c, text[cut:], ord(" "), ord("#"), [3, 1], [0, 2])
        assert (n1, n2) == (2, 3)
        assert list(fo1[0]) == [5, 15] and list(io1[0]) == [1, 11] and list(fo2[0]) == [25, 35, 45] and list(io2[1]) == [23, 33, 43]


def test_random_catalogues(mnw, oracle, torch_mod):
    rng = np.random.default_rng(71)
    with mnw.Context(0) as c:
        for trial, (rows, ncol, sep) in enumerate([(1, 3, " "), (257, 5, " "), (5000, 44, " "), (20000, 12, " "), (70000, 8, "\t")]):
            lines = ["#id mass x y z flag" + " pad" * 3, "# second header line"]
            for r in range(rows):
                w = []
                for k in range(ncol):
                    kind = k % 4
                    if kind == 0:
                        w.append(str(int(rng.integers(-10**12, 10**12))))
                    elif kind == 1:
                        w.append("%.6e" % (10 ** rng.uniform(8, 15)))
                    elif kind == 2:
                        w.append("%.5f" % rng.uniform(0, 250))
                    else:
                        w.append("%g" % rng.normal(0, 300))
                pad = sep * int(rng.integers(1, 4)) if sep == " " else sep
                ln = (sep * int(rng.integers(0, 3)) if sep == " " else "") + pad.join(w)
                if rng.random() < 0.02:
                    ln += " # trailing comment"
                lines.append(ln)
                if rng.random() < 0.01:
                    lines.append(sep * int(rng.integers(0, 5)))           # blank or separator-only line: dropped
            data = ("\n".join(lines) + ("\n" if trial % 2 == 0 else "")).encode()
            if sep != " ":
                data = data.replace(b" # trailing comment", b"# trailing comment")
            fcols = [k for k in range(ncol) if k % 4 != 0][:40]
            icols = [k for k in range(ncol) if k % 4 == 0][:40]
            n = _check(mnw, torch_mod, c, oracle, data, ord(sep), ord("#"), fcols, icols)
            assert n == rows
        # a separator that is not white space: the reference counts the words of the first line with bytes.Fields
        # (parse.go:96) and panics on any request beyond that single "word"; so do the oracle and the library
        with pytest.raises(ValueError):
            oracle.text_block_parse(b"1,2,3\n4,5,6\n", ord(","), ord("#"), [1], [0])
        with pytest.raises(mnw.MnwError):
            _parse(mnw, torch_mod, c, b"1,2,3\n4,5,6\n", ord(","), ord("#"), [1], [0])


def test_tile_geometry(mnw, oracle, torch_mod):
    """lines of every awkward length against the 12,288-byte tiles and the 16,384 bytes a workgroup stages: newlines on
    and around tile boundaries, lines longer than the staged overhang (4,080 bytes), longer than a tile, longer than
    the staged text; blank lines; with and without a final newline; the fast path must have taken the plain ones"""
    rng = np.random.default_rng(11)
    with mnw.Context(0) as c:
        for L in [1, 2, 3, 5, 11, 12, 47, 48, 63, 64, 65, 127, 128, 1023, 4079, 4080, 4081, 4095, 4096, 4097, 12287, 12288, 12289, 16384, 40001]:
            nlines = max(3 * 12288 // L + 7, 9)
            nlines = min(nlines, 20000)
            out = []
            for i in range(nlines):
                if L < 4:
                    tok = str(i % 10)
                    ln = " " * (L - 1 - len(tok)) + tok if L - 1 >= len(tok) else ""
                else:
                    a, b = str(i % 7), "%.3g" % rng.normal(0, 50)
                    body = a + " " + b
                    ln = (body + " " * (L - 1 - len(body))) if L - 1 >= len(body) else a + " 1"
                    if i % 3 == 0 and L - 1 > len(body) + 2:
                        pad = L - 1 - len(body)
                        ln = " " * (pad // 2) + a + " " * (pad - pad // 2 - 0) + b
                        ln = ln[:L - 1] if len(ln) > L - 1 else ln + " " * (L - 1 - len(ln))
                out.append(ln)
            data = "\n".join(out).encode() + (b"\n" if L % 2 else b"")
            if L < 4:
                n = _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [], [0])
            else:
                n = _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1], [0])
            if n == 0:
                continue
            lines, kept, general = c.text_stats()
            assert lines == data.count(b"\n") + 1 and kept == n
            if 11 <= L <= 4080:
                assert general == 0, (L, general)
            if L > 16368:                          # cannot end inside the staged text wherever it starts
                assert general > 0
        # random line lengths, blank lines and separator-only lines in between
        for trial in range(3):
            out = []
            for i in range(4000):
                r = rng.random()
                if r < 0.05:
                    out.append("")
                elif r < 0.1:
                    out.append(" " * int(rng.integers(1, 300)))
                else:
                    pad = int(rng.integers(1, 40)) if r < 0.95 else int(rng.integers(1, 6000))
                    out.append(" " * int(rng.integers(0, 4)) + str(int(rng.integers(-99, 10**9))) + " " * pad + "%.4f" % rng.uniform(0, 250) + " " * int(rng.integers(0, 3)))
            data = "\n".join(out).encode()
            _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1], [0])
        # comments in the middle of the data: those tiles take the general path, the rest does not
        out = ["%d %.3f" % (i, i * 0.25) + (" # note" if i == 5000 else "") for i in range(20000)]
        data = ("# header\n" + "\n".join(out) + "\n").encode()
        n = _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1], [0])
        lines, kept, general = c.text_stats()
        assert n == 20000 and 1 <= general <= 4 and lines == 20002
        # capacity: one row short
        with pytest.raises(mnw.MnwError):
            _parse(mnw, torch_mod, c, data, ord(" "), ord("#"), [1], [0], cap=19999)


def test_hard_float_literals(mnw, oracle, torch_mod):
    """literals outside the device's exact fast path (more than 15-19 digits, exponents beyond 10^+-22, subnormals,
    halfway cases of the float64 and of the float32 rounding, inf/nan spellings): converted on the host with strtod"""
    toks = ["0", "-0", "0.0", "1", "-1.5", "+2.5", "1e22", "1e23", "9007199254740993", "9007199254740992", "123456789012345678901234567890",
            "0.1", "0.3", "1.7976931348623157e308", "4.9e-324", "2.2250738585072014e-308", "1e-45", "1.401298464324817e-45",
            "7.006492321624085e-46", "3.4028234663852886e38", "3.4028235677973366e38", "3.4028235677973367e38", "1e39",
            "16777217", "16777217.0000000001", "0.000000000000000000000000000000000000000000001", "1.00000017881393432617187500001",
            "1.000000178813934326171875", "8.5", ".5", "5.", "1E5", "1e+5", "1e-5", "inf", "-Inf", "+infinity", "nan", "NaN",
            "123456.789e3", "0.00001e-20", "179769313486231570000000000000000000000e270"]
    rng = np.random.default_rng(5)
    toks += ["%.17g" % v for v in rng.standard_normal(300) * 10.0 ** rng.integers(-30, 30, 300)]
    toks += ["%d.%018d" % (rng.integers(0, 10**6), rng.integers(0, 10**18)) for _ in range(200)]
    data = "\n".join("%d %s" % (i, t) for i, t in enumerate(toks)).encode()
    with mnw.Context(0) as c:
        n = _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1], [0])
        assert n == len(toks)


def test_reference_panics_become_errors(mnw, torch_mod):
    cases = {
        b"1 2 3\n4 5\n": "columns",                      # a line with another number of fields (parse.go:114-118)
        b"1 2 3\n4 x 6\n": "Atoi",                       # strconv.Atoi fails
        b"1 2 3\n4 5 6.5.1\n": "ParseFloat",             # strconv.ParseFloat fails
        b"1 2 3\n4 99999999999999999999 6\n": "Atoi",    # out of int64's range
        b"1 2 3\n4 5 1e999\n": "range",                  # ParseFloat's ErrRange
    }
    with mnw.Context(0) as c:
        for data, what in cases.items():
            with pytest.raises(mnw.MnwError) as e:
                _parse(mnw, torch_mod, c, data, ord(" "), ord("#"), [2], [1])
            assert what in str(e.value), (data, str(e.value))
        with pytest.raises(mnw.MnwError) as e:             # a requested column the data does not have (parse.go:98-103)
            _parse(mnw, torch_mod, c, b"1 2 3\n", ord(" "), ord("#"), [5], [])
        assert "column 5 was requested" in str(e.value)
        n, fo, io = _parse(mnw, torch_mod, c, b"# only comments\n\n   \n", ord(" "), ord("#"), [0], [])
        assert n == 0


def test_wide_catalogue_all_columns(mnw, oracle, torch_mod):
    """120 columns, all of them requested in one call (the C ABI takes up to 128 columns per call); 129 is an argument error"""
    rng = np.random.default_rng(8)
    ncol = 120
    rows = []
    for r in range(700):
        rows.append(" ".join(str(int(rng.integers(-10**9, 10**9))) if k % 2 else "%.4e" % (10 ** rng.uniform(-3, 12)) for k in range(ncol)))
    data = ("\n".join(rows) + "\n").encode()
    with mnw.Context(0) as c:
        fcols, icols = list(range(0, ncol, 2)), list(range(1, ncol, 2))
        assert _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), fcols, icols) == 700
        assert c.text_stats()[2] == 0
        with pytest.raises(mnw.MnwError):
            _parse(mnw, torch_mod, c, b"1 2\n", ord(" "), ord("#"), list(range(65)), list(range(65, 129)))


def test_text_rows_then_parse(mnw, oracle, torch_mod):
    """mnw_text_rows = the first half of the parse (lines, lines kept: what the reference sizes its outputs with, text.go:192);
    the parse that follows reuses its upload, a parse of ANOTHER buffer of the same length does not"""
    rng = np.random.default_rng(5)
    with mnw.Context(0) as c:
        def make(seed):
            r = np.random.default_rng(seed)
            rows = ["%07d %9.4f" % (int(r.integers(0, 10**7)), r.uniform(0, 9999)) for _ in range(3000)]
            rows[100] = "# a comment line of the same length"[:17]
            rows[200] = " " * 17
            return ("\n".join(rows) + "\n").encode()
        a, b = make(1), make(2)
        assert len(a) == len(b) and a != b
        for data in (a, b, b"# nothing\n\n", b"1 2\n"):
            lines, kept = c.text_rows(data, ord(" "), ord("#"))
            rf, ri = oracle.text_block_parse(data, ord(" "), ord("#"), [1], [0])
            assert lines == data.count(b"\n") + 1 and kept == ri[0].size
            n = _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1], [0])
            assert n == kept and c.text_stats()[:2] == (lines, kept)
        c.text_rows(a, ord(" "), ord("#"))                       # rows of a, then a parse of b (same length, other bytes, other object)
        _check(mnw, torch_mod, c, oracle, b, ord(" "), ord("#"), [1], [0])
        _check(mnw, torch_mod, c, oracle, a, ord(" "), ord("#"), [1], [0])


def test_randomised_differential_short(mnw, oracle, torch_mod):
    """a short seeded leg of tests/fuzz_text.py: random well-formed and malformed catalogues, library against oracle"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_text
    n, errors, general = fuzz_text.run(cases=250, seed=20261004)
    assert n == 250 and errors > 10 and general > 10


def test_block_beyond_4_gib(mnw, oracle, torch_mod):
    """a block of more than 2^32 bytes (the reference's blocks go up to 5 GiB, text.go:35): offsets, line numbers and rows
    past the 32-bit range.  The text is one chunk repeated, so the oracle only has to parse the chunk; a literal that only
    the host path converts sits behind the 4 GiB mark."""
    rng = np.random.default_rng(12)
    lines = ["%d %.5f" % (int(rng.integers(0, 10**9)), rng.uniform(0, 250)) for _ in range(20000)]
    chunk = ("\n".join(lines) + "\n").encode()
    reps = (2**32 + 2**27) // len(chunk) + 1
    special = b"7 0.12345678901234567890123\n"                      # more than 19 digits: the host's strtod converts it
    data = chunk * (reps - 1) + special + chunk
    assert len(data) > 2**32 + 2**26
    rf, ri = oracle.text_block_parse(chunk, ord(" "), ord("#"), [1], [0])
    nrows = 20000 * reps + 1
    dev = torch_mod.device("cuda:0")
    with mnw.Context(0) as c:
        lines_n, kept = c.text_rows(data, ord(" "), ord("#"))
        assert kept == nrows and lines_n == nrows + 1
        fo = torch_mod.empty(nrows + 8, dtype=torch_mod.float32, device=dev)
        io = torch_mod.empty(nrows + 8, dtype=torch_mod.int64, device=dev)
        torch_mod.cuda.synchronize()
        n = c.text_parse(data, ord(" "), ord("#"), [1], [fo.data_ptr()], [0], [io.data_ptr()], nrows + 8)
        torch_mod.cuda.synchronize()
        assert n == nrows and c.text_stats()[2] == 0
        f, i = fo[:n].cpu().numpy(), io[:n].cpu().numpy()
    at = 20000 * (reps - 1)
    assert np.array_equal(f[:at].reshape(reps - 1, 20000).view(np.uint32), np.broadcast_to(rf[0].view(np.uint32), (reps - 1, 20000)))
    assert np.array_equal(i[:at].reshape(reps - 1, 20000), np.broadcast_to(ri[0], (reps - 1, 20000)))
    assert i[at] == 7 and f[at] == np.float32(0.12345678901234567890123)
    assert np.array_equal(f[at + 1:].view(np.uint32), rf[0].view(np.uint32)) and np.array_equal(i[at + 1:], ri[0])


def test_savetxt_default_format_catalogue(mnw, oracle, torch_mod):
    """numpy.savetxt's default '%.18e' (19 significant digits, many of them trailing zeros) and repr()-style 17-digit literals:
    converted on the device by the 128-bit power-of-ten table (mnw_text_num.h: text_scale10), bit-equal to ParseFloat's values"""
    rng = np.random.default_rng(31)
    n = 30000
    a = rng.standard_normal(n) * 10.0 ** rng.integers(-20, 20, n)
    b = np.round(rng.standard_normal(n) * 250.0, 2)              # short values printed long: "2.500000000000000000e+02"
    lines = ["%d %.18e %.18e %r" % (i, a[i], b[i], float(a[i] * 3.0)) for i in range(n)]
    data = ("\n".join(lines) + "\n").encode()
    with mnw.Context(0) as c:
        assert _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1, 2, 3], [0]) == n


def test_more_host_literals_than_the_first_list_holds(mnw, oracle, torch_mod):
    """more than 65,536 literals that only the host converts (25 digits each): the list grows and the pass is repeated"""
    rng = np.random.default_rng(32)
    n = 70000
    lines = ["%d 0.%025d" % (i, int(rng.integers(0, 10**18)) * 10**7 + int(rng.integers(1, 10**7))) for i in range(n)]
    data = ("\n".join(lines) + "\n").encode()
    with mnw.Context(0) as c:
        assert _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1], [0]) == n
        assert _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1], [0]) == n      # and again with the grown list


def test_field_requested_twice(mnw, oracle, torch_mod):
    """text.Reader.Block fills every output it is given, also when a column is requested twice or both as []int64 and as
    []float32 (go/text/text.go:202-221 parses the two lists independently)"""
    rng = np.random.default_rng(33)
    lines = ["%d %d %.4e %d" % (i, rng.integers(-10**6, 10**6), rng.standard_normal() * 1e9, rng.integers(0, 100)) for i in range(5000)]
    data = ("\n".join(lines) + "\n").encode()
    with mnw.Context(0) as c:
        n, fo, io = _parse(mnw, torch_mod, c, data, ord(" "), ord("#"), [2, 1, 2, 2], [1, 3, 1])
        rf, ri = oracle.text_block_parse(data, ord(" "), ord("#"), [2, 1, 2, 2], [1, 3, 1])
        assert n == 5000
        for a, b in zip(fo, rf):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for a, b in zip(io, ri):
            assert np.array_equal(a, b)


def test_go_literal_rules_on_the_host_path(mnw, oracle, torch_mod):
    """what strconv.ParseFloat takes and what it refuses among the literals the device hands to the host: a sign in front of
    inf but not of nan, digit separators between digits only, hexadecimal floats with their mandatory p exponent"""
    good = ["inf", "-Inf", "+INFINITY", "nan", "NaN", "1_000", "1_23.50_0_0e+1_2", "0x1p-2", "0x1.8p3", "0X_1FFFP-16", "-0x.8p1"]
    data = "\n".join("%d %s" % (i, t) for i, t in enumerate(good)).encode()
    with mnw.Context(0) as c:
        assert _check(mnw, torch_mod, c, oracle, data, ord(" "), ord("#"), [1], [0]) == len(good)
        for t in ["+nan", "-nan", "nan(1)", "infinit", "_1", "1_", "1__0", "1_.5", "0x1.8", "1e_5", "0x", "1e999", "-1e999"]:
            with pytest.raises(mnw.MnwError) as e:
                _parse(mnw, torch_mod, c, ("0 1.5\n1 %s\n" % t).encode(), ord(" "), ord("#"), [1], [0])
            assert "ParseFloat" in str(e.value), (t, str(e.value))
            with pytest.raises(ValueError):
                oracle.go_parse_float(t)


def test_number_conversions_on_the_device(mnw):
    """minnow_amd/csrc/mnw_text_num.h compiled for the device, every token of tests/golden/text_num_tokens.txt read from global
    memory and from LDS, against the host build of the same source (which tests/test_text_num_cpu.py pins to the correctly
    rounded values).  Guards against the miscompilation DESIGN.md section 3.10 records: an earlier form of the conversion
    loops rejected every literal with an exponent when the token was read through a global pointer."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "cpp", "text_num_dev.hip")
    exe = os.path.join(root, "tests", "cpp", "text_num_dev")
    deps = [src, os.path.join(root, "minnow_amd", "csrc", "mnw_text_num.h"), os.path.join(root, "minnow_amd", "csrc", "mnw_pow10_table.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", src, "-o", exe])
    r = subprocess.run([exe, os.path.join(root, "tests", "golden", "text_num_tokens.txt")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
