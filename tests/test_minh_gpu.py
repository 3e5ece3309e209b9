"""minh column transforms (go/minh/minh.go:139-149, 315-319): processFloatGroup and the reader's
10^x on device columns against the oracle's restatement, and the whole logarithmic-column path on the vectors of the
reference's own test (go/minh/minh_test.go:10-117).  Log10 / Pow follow Go's portable algorithms: parity unpinned
(DESIGN.md section 1); the GPU and the oracle restate them independently and must agree bit for bit."""
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


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_process_float_group_matches_oracle(mnw, oracle, torch_mod):
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(41)
    n = 200003                                      # not a multiple of four: scalar tail
    mass = (10 ** rng.uniform(7.5, 16.5, n)).astype(np.float32)          # some below 10^8, some above 10^16: both clamps
    mass[:8] = [0.0, -1.0, np.inf, np.nan, 1e8, 1e16, np.float32(1e16) * np.float32(0.9999999), 1.0]
    lin = rng.uniform(90, 210, n).astype(np.float32)
    lin[:4] = [100.0, 200.0, np.nextafter(np.float32(200), np.float32(0)), np.nan]
    with mnw.Context(0) as c:
        tm, tl = torch.from_numpy(mass).to(dev), torch.from_numpy(lin).to(dev)
        om, ol = torch.empty_like(tm), torch.empty_like(tl)
        torch.cuda.synchronize()
        c.minh_process([(tm.data_ptr(), om.data_ptr(), n, 8.0, 16.0, 1), (tl.data_ptr(), ol.data_ptr(), n, 100.0, 200.0, 0)])
        torch.cuda.synchronize()
        rm, rl = oracle.process_float_group(mass, True, 8.0, 16.0), oracle.process_float_group(lin, False, 100.0, 200.0)
        gm, gl = om.cpu().numpy(), ol.cpu().numpy()
        nan = np.isnan(rm)
        assert np.array_equal(np.isnan(gm), nan) and np.array_equal(_bits(gm)[~nan], _bits(rm)[~nan])
        nan = np.isnan(rl)
        assert np.array_equal(np.isnan(gl), nan) and np.array_equal(_bits(gl)[~nan], _bits(rl)[~nan])
        assert gm[0] == 8.0 and np.isnan(gm[1]) and gm[2] == np.nextafter(np.float32(16), np.float32(0))   # log10(0) = -Inf -> low
        # in place, unaligned pointer (element offset 1)
        c.minh_process([(tm.data_ptr() + 4, tm.data_ptr() + 4, n - 1, 8.0, 16.0, 1)])
        torch.cuda.synchronize()
        assert np.array_equal(_bits(tm.cpu().numpy()[1:])[~np.isnan(rm[1:])], _bits(rm[1:])[~np.isnan(rm[1:])])
        # reader: 10^x
        logs = rng.uniform(-44, 39, n).astype(np.float32)
        logs[:6] = [0.0, 1.0, 0.5, -0.5, np.inf, -np.inf]
        t = torch.from_numpy(logs).to(dev)
        torch.cuda.synchronize()
        c.minh_unlog([(t.data_ptr(), n)])
        torch.cuda.synchronize()
        assert np.array_equal(_bits(t.cpu().numpy()), _bits(oracle.unlog(logs)))


def test_log_column_of_reference_test(mnw, oracle, torch_mod):
    """the "log" and "float" columns of TestReaderWriter (minh_test.go:21-45): Column{FloatGroup, Log: 1, Low: 10,
    High: 14, Dx: 0.01} on {1e10, 1e11, 1e11, 1e14, 3e13} and {1e14, 1e14, 1e14}; the reference checks the round trip
    with log32sEq, |log10(x) - log10(x')| <= dx"""
    torch = torch_mod
    dev = torch.device("cuda:0")
    cases = [(1, 10.0, 14.0, 0.01, [1e10, 1e11, 1e11, 1e14, 3e13]), (1, 10.0, 14.0, 0.01, [1e14, 1e14, 1e14]),
             (0, 100.0, 200.0, 1.0, [100, 200, 125, 150, 100]), (0, 100.0, 200.0, 1.0, [100, 100, 100])]
    with mnw.Context(0) as c:
        for blk, (log, low, high, dx, vals) in enumerate(cases):
            x = np.array(vals, dtype=np.float32)
            n = x.size
            pixels = mnw.float_group_pixels(low, high, dx)
            tx = torch.from_numpy(x).to(dev)
            buf = torch.empty_like(tx)
            out = torch.zeros(8 * n + 16, dtype=torch.uint8, device=dev)
            y = torch.empty_like(tx)
            torch.cuda.synchronize()
            c.minh_process([(tx.data_ptr(), buf.data_ptr(), n, low, high, log)])
            (m,) = c.encode_blocks([mnw.EncJob(buf.data_ptr(), out.data_ptr(), n, out.numel(), mnw.FLOAT_GROUP, low, high, pixels, 1)])
            c.decode_blocks([mnw.DecJob(out.data_ptr(), y.data_ptr(), n, m.nbytes, m.min, m.bits, mnw.FLOAT_GROUP, low, high,
                                        pixels, 1, seed=2, block=blk)])
            if log:
                c.minh_unlog([(y.data_ptr(), n)])
            torch.cuda.synchronize()
            # oracle: the same pipeline
            pb = oracle.process_float_group(x, log, low, high)
            omin, obits, odata = oracle.float_encode(pb, low, high, pixels, True, literal=False)
            assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0)
            assert out[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes()
            ref = oracle.float_decode(odata, omin, obits, low, high, pixels, True, n, u=oracle.dither(2, blk, n), literal=False)
            if log:
                ref = oracle.unlog(ref)
            got = y.cpu().numpy()
            assert np.array_equal(_bits(got), _bits(ref))
            if log:   # log32sEq (minh_test.go)
                assert np.all(np.abs(np.log10(got.astype(np.float64)) - np.log10(x.astype(np.float64))) <= dx * 1.001 + 1e-6)
            else:     # float32sEq with eps = dx; x == high is clamped to just below it
                assert np.all(np.abs(got.astype(np.float64) - np.minimum(x, np.nextafter(np.float32(high), np.float32(0)))) <= dx * 1.001 + 1e-4)


@pytest.mark.parametrize("which,name", [(0, "log10"), (1, "pow10")])
def test_fast_paths_equal_exact_for_every_float32(mnw, which, name):
    """The kernels compute float32(Log10(float64(x))) / float32(Pow(10, float64(y))) through a table + polynomial
    with a rounding-ambiguity check and fall back to the exact restatement otherwise.  Exhaustive: all 2^32 float32
    bit patterns, zero mismatches allowed."""
    import ctypes as C
    from minnow_amd import _lib
    L = _lib.lib()
    with mnw.Context(0) as c:
        out = (C.c_uint64 * 2)()
        bad = slow = 0
        for part in range(4):
            assert L.mnw_minh_selftest(c.handle, which, part << 30, 1 << 30, out) == 0
            bad += out[0]; slow += out[1]
        assert bad == 0, "%s: %d float32 inputs where the fast path differs from the exact one" % (name, bad)
        # the fast path's domain: positive normal x (log10: half of all patterns), |y| in (1e-6, 37.5) (pow10: a tenth of all patterns, all that 10^y maps into float32's range)
        assert slow < (1 << 32) * (0.51 if which == 0 else 0.92)


@pytest.mark.parametrize("mode", ["single_read", "two_pass", "fused"])
def test_transform_inside_the_codec_jobs(mnw, oracle, torch_mod, mode):
    """mnw_enc_job.xform / mnw_dec_job.xform: processFloatGroup applied while the block is encoded (at load time in the
    single-read kernel, through a staging copy in the other structures and in the exact path) and 10^x while it is
    decoded.  Against the oracle's processFloatGroup -> floatGroup.writeData and readData -> 10^x."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(61)
    cols = []
    for n in (1, 4097, 8192, 100001, 2**20 + 3):
        cols.append((mnw.XF_LOG10 | mnw.XF_CLAMP, 8.0, 16.0, 0.01, (10 ** rng.uniform(7.5, 16.5, n)).astype(np.float32)))
        cols.append((mnw.XF_CLAMP, 100.0, 200.0, 1.0, rng.uniform(90, 210, n).astype(np.float32)))
    zero = (10 ** rng.uniform(9, 12, 30000)).astype(np.float32); zero[17] = 0.0          # log10(0) = -Inf -> Low
    cols.append((mnw.XF_LOG10 | mnw.XF_CLAMP, 8.0, 16.0, 0.01, zero))
    neg = (10 ** rng.uniform(9, 12, 30000)).astype(np.float32); neg[5] = -3.0            # log10(<0) = NaN -> exact path
    cols.append((mnw.XF_LOG10 | mnw.XF_CLAMP, 8.0, 16.0, 0.01, neg))
    cols.append((0, 0.0, 250.0, 0.001, np.mod(125 + 20 * rng.standard_normal(50000), 250).astype(np.float32)))   # plain
    # a ring of more than 2^31 pixels: the 64-bit decode class (and the exact encode path), with the reader's 10^x
    cols.append((mnw.XF_LOG10 | mnw.XF_CLAMP, 8.0, 16.0, 1e-9, (10 ** rng.uniform(8.5, 15.5, 7001)).astype(np.float32)))
    with mnw.Context(0) as c:
        if mode == "two_pass":
            c.set_option("single_read", 0)
        elif mode == "fused":
            c.set_option("fused", 1)
        ts = [torch.from_numpy(x).to(dev) for *_, x in cols]
        outs = [torch.zeros(8 * x.size + 16, dtype=torch.uint8, device=dev) for *_, x in cols]
        ys = [torch.empty_like(t) for t in ts]
        pix = [mnw.float_group_pixels(lo, hi, dx) for _, lo, hi, dx, _ in cols]
        jobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.FLOAT_GROUP, lo, hi, p, 1, xform=xf)
                for (xf, lo, hi, dx, x), t, o, p in zip(cols, ts, outs, pix)]
        torch.cuda.synchronize()
        c.timing_enable(True)
        for rep in range(2):
            metas = c.encode_blocks(jobs)
        torch.cuda.synchronize()
        ran = {k for k, v in c.timing_get().items() if v["launches"]}
        assert {"single_read": "k_encode_reg_f32", "two_pass": "k_pack_f32", "fused": "k_encode_f32"}[mode] in ran
        assert ("k_minh_process" in ran)                       # the staging copy (always for the exact-path block)
        djobs = []
        for b, ((xf, lo, hi, dx, x), m, o, y, p) in enumerate(zip(cols, metas, outs, ys, pix)):
            pb = oracle.process_float_group(x, xf & mnw.XF_LOG10, lo, hi) if xf else x
            omin, obits, odata = oracle.float_encode(pb, lo, hi, p, True, literal=False)
            assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), b
            assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes(), b
            djobs.append(mnw.DecJob(o.data_ptr(), y.data_ptr(), x.size, m.nbytes, m.min, m.bits, mnw.FLOAT_GROUP, lo, hi, p, 1,
                                    seed=3, block=b, xform=xf & mnw.XF_LOG10))
        c.timing_reset()
        c.decode_blocks(djobs)
        torch.cuda.synchronize()
        # the reader's 10^x (go/minh/minh.go:315-319) happens inside the unpack kernel; as a pass of its own (unlog_fused = 0) it
        # gives the same floats
        kt = c.timing_get()
        assert kt["k_unpack_f32"]["launches"] == 2 and kt["k_minh_unlog"]["launches"] == 0      # narrow and wide decode classes
        first = [y.clone() for y in ys]
        c.set_option("unlog_fused", 0)
        for y in ys:
            y.zero_()
        torch.cuda.synchronize()
        c.decode_blocks(djobs)
        torch.cuda.synchronize()
        assert c.timing_get()["k_minh_unlog"]["launches"] >= 1
        for a_, b_ in zip(first, ys):
            assert torch.equal(a_.view(torch.int32), b_.view(torch.int32))
        c.set_option("unlog_fused", 1)
        for b, ((xf, lo, hi, dx, x), m, o, y, p) in enumerate(zip(cols, metas, outs, ys, pix)):
            odata = o[: m.nbytes].cpu().numpy()
            ref = oracle.float_decode(odata, m.min, m.bits, lo, hi, p, True, x.size, u=oracle.dither(3, b, x.size), literal=False)
            if xf & mnw.XF_LOG10:
                ref = oracle.unlog(ref)
            got = y.cpu().numpy()
            assert np.array_equal(_bits(got), _bits(ref)) or np.array_equal(got, ref, equal_nan=True), b
