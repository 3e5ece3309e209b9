"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the golden
fixtures.  Bit-exact for every integer/byte result (streams, mins, bits, decoded integers); decoded
floats are compared bit-for-bit with the oracle fed the same dither deviates, and with the
reference's own criterion |x - x'| <= dx (go/minnow_test.go:328-334).

Nothing here reads /root/reference: inputs are seeded or come from tests/golden/.
"""
import os
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
def ctx(mnw):
    c = mnw.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


# ------------------------------------------------------------------------------------------------
# L0: bit.Array pack / unpack
# ------------------------------------------------------------------------------------------------
def test_pack_unpack_golden_all_widths(mnw, ctx, golden):
    """Reference KATs (cy_bit.array / from_array), widths 1..64 x ragged lengths incl. empty."""
    _, kat = golden
    X, idx, blob = kat["pack_X"], kat["pack_index"], kat["pack_blob"]
    for b, L, pos, nbytes in idx:
        b, L, pos, nbytes = int(b), int(L), int(pos), int(nbytes)
        want = blob[pos:pos + nbytes]
        arr = mnw.BitArray.new(ctx, b, X[:L])          # un-masked input: masked by the kernel
        assert arr.Data.size == nbytes
        assert np.array_equal(arr.Data, want), (b, L)
        back = arr.slice(ctx)
        assert np.array_equal(back, X[:L] & np.uint64((1 << b) - 1)), (b, L)


def test_pack_known_answer(mnw, ctx):
    assert mnw.BitArray.new(ctx, 3, [1, 2, 3, 4, 5]).Data.tobytes().hex() == "d158"


def test_go_testarray_shape(mnw, ctx, oracle):
    """go/bit/bit_test.go:9-31: 123 random 63-bit values, every width 1..64."""
    rng = np.random.default_rng(11)
    data = rng.integers(0, 2**63, size=123, dtype=np.uint64)
    for bits in range(1, 65):
        arr = mnw.BitArray.new(ctx, bits, data)
        assert np.array_equal(arr.Data, oracle.pack(bits, data))
        assert np.array_equal(arr.slice(ctx), data & np.uint64((1 << bits) - 1))


def test_go_benchmark_shape(mnw, ctx, oracle):
    """go/bit/bit_test.go:71-112: 1e5 elements i%100 at widths 8/11/16/21/32/45/64."""
    x = (np.arange(100000) % 100).astype(np.uint64)
    for bits in (8, 11, 16, 21, 32, 45, 64):
        arr = mnw.BitArray.new(ctx, bits, x)
        assert np.array_equal(arr.Data, oracle.pack(bits, x, literal=False))
        assert np.array_equal(arr.slice(ctx), x)


def test_pack_multi_tile_ragged(mnw, ctx, oracle):
    rng = np.random.default_rng(12)
    for n in (2047, 2048, 2049, 4096 + 63, 3 * 2048 + 1, 100003):
        for bits in (1, 5, 13, 18, 31, 32, 33, 47, 63, 64):
            x = rng.integers(0, 2**63, size=n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            arr = mnw.BitArray.new(ctx, bits, x)
            assert np.array_equal(arr.Data, oracle.pack(bits, x, literal=False)), (n, bits)
            assert np.array_equal(arr.slice(ctx), x & np.uint64((1 << bits) - 1) if bits < 64 else x)


def test_pack_errors(mnw, ctx):
    x = np.zeros(4, np.uint64)
    with pytest.raises(mnw.MnwError) as e:
        mnw.BitArray.new(ctx, 65, x)                       # bit.go:85-87
    assert e.value.status == -1
    with pytest.raises(mnw.MnwError) as e:
        mnw.BitArray.buffered(ctx, 8, x, np.zeros(5, np.uint8))   # bit.go:88-92
    assert e.value.status == -2
    arr = mnw.BitArray.new(ctx, 8, x)
    with pytest.raises(mnw.MnwError):
        arr.slice(ctx, out=np.zeros(3, np.uint64))          # bit.go:30-33


# ------------------------------------------------------------------------------------------------
# L1: intGroup
# ------------------------------------------------------------------------------------------------
def _int_roundtrip(mnw, ctx, oracle, x, lossless=True):
    """lossless=False: the reference itself loses data there -- PrecisionNeeded's float64 rounding
    (bit.go:19-21) returns a width one too small for some max >= 2^49, and BufferedArray masks the
    top bit away (bit.go:104-107).  Parity means reproducing that, so decode is compared with the
    oracle's decode of the same stream."""
    x = np.asarray(x, dtype=np.int64)
    g = mnw.IntGroupCodec(ctx, x.size)
    data = g.write_data(x)
    omin, obits, odata = oracle.int_encode(x, literal=x.size <= 5000)
    assert (g.mins[0], g.bits[0]) == (omin, obits)
    assert data.tobytes() == odata.tobytes()
    back = g.read_data(data, 0)
    assert np.array_equal(back, oracle.int_decode(odata, omin, obits, x.size, literal=x.size <= 5000))
    if lossless:
        assert np.array_equal(back, x)
    return g


def test_int_group_golden(mnw, ctx, golden):
    small, _ = golden
    for case in small["int_group"]:
        x = np.array(case["x"], dtype=np.int64)
        g = mnw.IntGroupCodec(ctx, x.size)
        data = g.write_data(x)
        assert (g.mins[0], g.bits[0]) == (case["min"], case["bits"])
        assert data.tobytes() == bytes.fromhex(case["bytes"])
        assert np.array_equal(g.read_data(data, 0), x)


def test_int_group_reference_vectors_and_edges(mnw, ctx, oracle):
    # go/minnow_test.go:242-268 TestBitIntRecord
    for v in ([100, 101, 102, 104], [1024, 1024, 1024], [0, 1023, 500], [-1000000, -500000]):
        _int_roundtrip(mnw, ctx, oracle, v)
    g = _int_roundtrip(mnw, ctx, oracle, [1024, 1024, 1024])
    assert g.bits[0] == 0 and g.sizes[0] == 0                    # constant block: 0 bytes
    g = _int_roundtrip(mnw, ctx, oracle, np.zeros(0, np.int64))  # empty: min 0, bits 0
    assert (g.mins[0], g.bits[0]) == (0, 0)
    _int_roundtrip(mnw, ctx, oracle, [-(2**63), 2**63 - 2])       # 64-bit range
    _int_roundtrip(mnw, ctx, oracle, [-(2**63), 0, 2**63 - 2, 5])
    g = _int_roundtrip(mnw, ctx, oracle, [2**62, -(2**62)], lossless=False)   # range 2^63 -> 63 bits (!)
    assert g.bits[0] == 63
    g = _int_roundtrip(mnw, ctx, oracle, [0, 2**49], lossless=False)          # 49 bits, not 50
    assert g.bits[0] == 49
    _int_roundtrip(mnw, ctx, oracle, [0, 2**50 + 3], lossless=False)
    _int_roundtrip(mnw, ctx, oracle, [0, 2**49 - 1])
    _int_roundtrip(mnw, ctx, oracle, [5, 5 + 2**48])
    _int_roundtrip(mnw, ctx, oracle, [7])


def test_int_group_domain_error(mnw, ctx):
    """max-min == 2^64-1: Go's PrecisionNeeded returns int(-Inf); reported as MNW_E_DOMAIN."""
    g = mnw.IntGroupCodec(ctx, 2)
    with pytest.raises(mnw.MnwError) as e:
        g.write_data(np.array([-(2**63), 2**63 - 1], dtype=np.int64))
    assert e.value.status == -6


def test_int_group_random_widths(mnw, ctx, oracle):
    rng = np.random.default_rng(13)
    for b in (1, 2, 7, 8, 11, 16, 21, 27, 32, 33, 40, 45, 48, 52, 60, 63):
        for n in (1, 63, 64, 65, 2048, 5000, 70001):
            off = int(rng.integers(-10**12, 10**12))
            x = off + rng.integers(0, 2**b, size=n, dtype=np.uint64).astype(np.int64)
            _int_roundtrip(mnw, ctx, oracle, x)


def test_int_group_c1_config(mnw, ctx, oracle):
    """BASELINE config 1: int64 column, 1e6 rows, x = 1000 + U{0..2^b-1}."""
    rng = np.random.Generator(np.random.PCG64(1))
    for b in (8, 11, 16, 21, 32, 45):
        x = 1000 + rng.integers(0, 2**b, size=10**6, dtype=np.int64)
        g = mnw.IntGroupCodec(ctx, x.size)
        data = g.write_data(x)
        omin, obits, odata = oracle.int_encode(x, literal=False)
        assert (g.mins[0], g.bits[0]) == (omin, obits)
        assert data.tobytes() == odata.tobytes()
        assert np.array_equal(g.read_data(data, 0), x)


# ------------------------------------------------------------------------------------------------
# L1: floatGroup
# ------------------------------------------------------------------------------------------------
def _float_check(mnw, ctx, oracle, x, low, high, pixels, periodic=True, seed=5, block=3):
    x = np.asarray(x, dtype=np.float32)
    g = mnw.FloatGroupCodec(ctx, x.size, low, high, pixels, periodic)
    data = g.write_data(x)
    omin, obits, odata = oracle.float_encode(x, low, high, pixels, periodic, literal=x.size <= 5000)
    assert (g.ig.mins[0], g.ig.bits[0]) == (omin, obits), (g.ig.mins[0], g.ig.bits[0], omin, obits)
    assert data.tobytes() == odata.tobytes()
    # decode: same dither deviates -> bit-identical floats
    g.ig.mins[0], g.ig.bits[0] = omin, obits
    c = ctx
    import ctypes as C
    out = np.empty(x.size, dtype=np.float32)
    rc = c._L.mnw_float_decode(c.handle, data.ctypes.data_as(C.c_void_p), data.size, omin, obits,
                               g.low, g.high, g.pixels, g.periodic, seed, block, x.size,
                               out.ctypes.data_as(C.c_void_p))
    assert rc == 0, c._L.mnw_last_error(c.handle)
    u = oracle.dither(seed, block, x.size)
    want = oracle.float_decode(odata, omin, obits, low, high, pixels, periodic, x.size, u=u,
                               literal=x.size <= 5000)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    return g, out


def test_float_group_golden(mnw, ctx, golden):
    small, _ = golden
    for case in small["float_group"]:
        x = np.frombuffer(bytes.fromhex(case["x_f32_hex"]), dtype="<f4")
        assert mnw.float_group_pixels(case["low"], case["high"], case["dx"]) == case["pixels"]
        g = mnw.FloatGroupCodec(ctx, x.size, case["low"], case["high"], case["pixels"], True)
        data = g.write_data(x)
        assert (g.ig.mins[0], g.ig.bits[0]) == (case["min"], case["bits"])
        assert data.tobytes() == bytes.fromhex(case["bytes"])


def test_float_group_reference_test_vectors(mnw, ctx, oracle):
    """go/minnow_test.go:270-310 TestQFloatRecord: lim [-50,100], dx 1 and 10, tol dx."""
    low, high = -50.0, 100.0
    for dx, blocks in ((1.0, [[-50, 0, 50, 49], [25, 25, 25, 25]]),
                       (10.0, [[-50, 0, 50, 49, 0], [1, 2, 3, 4, 5], [0, 20, 0, 20, 0]])):
        g = mnw.FloatGroupCodec.from_limits(ctx, 4, (low, high), dx)
        assert g.pixels == (150 if dx == 1.0 else 15)
        for blk in blocks:
            x = np.array(blk, dtype=np.float32)
            _, y = _float_check(mnw, ctx, oracle, x, low, high, g.pixels)
            assert np.all(np.abs(y - x) <= np.float32(dx))        # float32sEq
    g, _ = _float_check(mnw, ctx, oracle, [25, 25, 25, 25], low, high, 150)
    assert g.ig.bits[0] == 0                                      # constant block costs 0 bytes


def test_float_group_gaussian_and_wrapping(mnw, ctx, oracle):
    """SURVEY 8d C2 shape at small N, plus arcs that straddle 0 (periodic re-basing)."""
    rng = np.random.default_rng(14)
    low, high, dx = 0.0, 250.0, 0.001
    pixels = mnw.float_group_pixels(low, high, dx)
    assert pixels == 250000
    for n in (1, 2, 63, 64, 65, 2047, 2048, 2049, 6000, 100001):
        for centre, sigma in ((125.0, 20.0), (0.3, 0.5), (249.9, 1.0), (60.0, 0.01), (125.0, 80.0)):
            x = centre + sigma * rng.standard_normal(n)
            x = np.mod(x, 250.0).astype(np.float32)
            x[x >= np.float32(250.0)] = 0.0
            g, y = _float_check(mnw, ctx, oracle, x, low, high, pixels)
            d = np.abs(y.astype(np.float64) - x.astype(np.float64))
            d = np.minimum(d, 250.0 - d)                           # periodic distance
            assert np.all(d <= dx * 1.01 + 1e-4), d.max()


def test_float_group_more_parameter_sets(mnw, ctx, oracle):
    rng = np.random.default_rng(15)
    for low, high, dx in ((-5000.0, 5000.0, 1.0), (8.0, 16.0, 0.01), (10.0, 14.0, 0.01), (0.0, 1000.0, 5.0),
                          (-1.0, 1.0, 1e-6), (0.0, 1.0, 0.5), (0.0, 62.5, 0.0003)):
        pixels = mnw.float_group_pixels(low, high, dx)
        assert pixels == oracle.float_group_pixels(low, high, dx)
        for n in (5, 3000, 40000):
            x = (low + (high - low) * rng.random(n)).astype(np.float32)
            q = oracle.quantize(x, low, high, pixels)
            x = x[(q >= 0) & (q < pixels)]                        # keep the in-range case here
            _float_check(mnw, ctx, oracle, x, low, high, pixels)
            c = low + (high - low) * rng.random()
            x = (c + (high - low) * 0.01 * rng.standard_normal(n)).astype(np.float32)
            q = oracle.quantize(x, low, high, pixels)
            x = x[(q >= 0) & (q < pixels)]
            _float_check(mnw, ctx, oracle, x, low, high, pixels)


def test_float_group_out_of_range_exact_path(mnw, ctx, oracle):
    """x < low, x >= high (q == pixels reachable by rounding), NaN, Inf: periodicMin depends on the
    element order there (group.go:384-409); the GPU's sequential path must match the restatement."""
    rng = np.random.default_rng(16)
    low, high, pixels = 0.0, 250.0, 250000
    base = np.mod(125 + 5 * rng.standard_normal(5000), 250).astype(np.float32)
    cases = []
    for pos in (0, 1, 77, 4999):
        for bad in (250.0, 251.5, -0.5, -300.0, 700.0, np.nextafter(np.float32(250), np.float32(0))):
            x = base.copy(); x[pos] = bad
            cases.append(x)
    x = base.copy(); x[10] = np.nan; cases.append(x)
    x = base.copy(); x[0] = np.inf; cases.append(x)
    x = base.copy(); x[4000] = -np.inf; x[17] = 1e30; cases.append(x)
    cases.append(np.array([250.0], dtype=np.float32))
    cases.append(np.array([-1.0, 260.0, 3.0], dtype=np.float32))
    cases.append(rng.uniform(-400, 600, size=3001).astype(np.float32))
    cases.append(np.sort(rng.uniform(-10, 260, size=2500)).astype(np.float32))
    for x in cases:
        g = mnw.FloatGroupCodec(ctx, x.size, low, high, pixels, True)
        data = g.write_data(x)
        omin, obits, odata = oracle.float_encode(x, low, high, pixels, True, literal=False)
        assert (g.ig.mins[0], g.ig.bits[0]) == (omin, obits)
        assert data.tobytes() == odata.tobytes()


def test_float_group_non_periodic_and_big_ring(mnw, ctx, oracle):
    rng = np.random.default_rng(17)
    x = (300 * rng.standard_normal(10000)).astype(np.float32)
    _float_check(mnw, ctx, oracle, x, -5000.0, 5000.0, 10000, periodic=False)
    _float_check(mnw, ctx, oracle, x * 100, -5000.0, 5000.0, 10000, periodic=False)   # q outside [0,pixels)
    # ring wider than 2^31 pixels: 64-bit path
    pixels = 2**33
    x = (rng.random(5000) * 1000).astype(np.float32)
    _float_check(mnw, ctx, oracle, x, 0.0, 1000.0, pixels, periodic=True)


# ------------------------------------------------------------------------------------------------
# batched device-pointer API (per-column blocks as the unit of dispatch)
# ------------------------------------------------------------------------------------------------
def test_batched_mixed_blocks_device_api(mnw, ctx, oracle, torch_mod):
    torch = torch_mod
    rng = np.random.default_rng(18)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    low, high, pixels = 0.0, 250.0, 250000
    host_in, jobs, outs, keep = [], [], [], []
    sizes = [0, 1, 100, 2048, 5000, 70000, 300000]
    for k, n in enumerate(sizes):
        if k % 2 == 0:
            x = np.mod(60 + 10 * rng.standard_normal(n), 250).astype(np.float32)
            t = torch.from_numpy(x).to(dev)
            o = torch.zeros(4 * n + 16, dtype=torch.uint8, device=dev)
            jobs.append(mnw.EncJob(t.data_ptr() if n else 0, o.data_ptr(), n, o.numel(), mnw.FLOAT_GROUP, low, high, pixels, 1))
        else:
            x = (10**9 + rng.integers(0, 2**(5 + 6 * k), size=n)).astype(np.int64)
            t = torch.from_numpy(x).to(dev)
            o = torch.zeros(8 * n + 16, dtype=torch.uint8, device=dev)
            jobs.append(mnw.EncJob(t.data_ptr(), o.data_ptr(), n, o.numel(), mnw.INT_GROUP))
        host_in.append(x); outs.append(o); keep.append(t)
    torch.cuda.synchronize()      # inputs/outputs were produced on torch's stream
    metas = ctx.encode_blocks(jobs, stream=stream)
    torch.cuda.synchronize()
    # one contiguous "file image" with byte-granular offsets, as blockIndex lays blocks out
    image = bytearray(b"\x07\x07\x07")            # odd start so streams are misaligned
    offs = []
    for k, (x, m, o) in enumerate(zip(host_in, metas, outs)):
        if k % 2 == 0:
            omin, obits, odata = oracle.float_encode(x, low, high, pixels, True, literal=False)
        else:
            omin, obits, odata = oracle.int_encode(x, literal=False)
        assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), k
        got = o[: m.nbytes].cpu().numpy()
        assert got.tobytes() == odata.tobytes(), k
        offs.append(len(image)); image += got.tobytes()
    image += b"\0" * 16
    img = torch.from_numpy(np.frombuffer(bytes(image), dtype=np.uint8).copy()).to(dev)
    djobs, douts = [], []
    for k, (x, m) in enumerate(zip(host_in, metas)):
        n = x.size
        if k % 2 == 0:
            o = torch.empty(n, dtype=torch.float32, device=dev)
            djobs.append(mnw.DecJob(img.data_ptr() + offs[k], o.data_ptr() if n else 0, n, m.nbytes, m.min, m.bits,
                                    mnw.FLOAT_GROUP, low, high, pixels, 1, seed=99, block=k))
        else:
            o = torch.empty(n, dtype=torch.int64, device=dev)
            djobs.append(mnw.DecJob(img.data_ptr() + offs[k], o.data_ptr(), n, m.nbytes, m.min, m.bits, mnw.INT_GROUP))
        douts.append(o)
    torch.cuda.synchronize()
    ctx.decode_blocks(djobs, stream=stream)
    torch.cuda.synchronize()
    for k, (x, m, o) in enumerate(zip(host_in, metas, douts)):
        got = o.cpu().numpy()
        if k % 2 == 0:
            u = oracle.dither(99, k, x.size)
            data = np.frombuffer(bytes(image[offs[k]: offs[k] + m.nbytes]), dtype=np.uint8)
            want = oracle.float_decode(data, m.min, m.bits, low, high, pixels, True, x.size, u=u, literal=False)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
        else:
            assert np.array_equal(got, x), k


def test_capacity_error_reported_per_block(mnw, ctx, torch_mod):
    torch = torch_mod
    dev = torch.device("cuda:0")
    x = torch.arange(0, 100000, dtype=torch.int64, device=dev)
    o = torch.zeros(64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    metas = ctx.encode_blocks([mnw.EncJob(x.data_ptr(), o.data_ptr(), x.numel(), 64, mnw.INT_GROUP)], check=False)
    assert metas[0].status == -5 and metas[0].bits == 17
    assert int(o.sum().item()) == 0                                # nothing written


def test_async_split_form(mnw, ctx, oracle, torch_mod):
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(19)
    xs = [np.mod(30 + rng.standard_normal(50000), 250).astype(np.float32) for _ in range(4)]
    ts = [torch.from_numpy(x).to(dev) for x in xs]
    os_ = [torch.zeros(4 * x.size, dtype=torch.uint8, device=dev) for x in xs]
    jobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1)
            for t, o in zip(ts, os_)]
    torch.cuda.synchronize()
    ctx.encode_blocks_async(jobs)
    with pytest.raises(mnw.MnwError):
        ctx.encode_blocks_async(jobs)            # one batch at a time (block order == file order)
    metas = ctx.encode_blocks_finish()
    for x, m, o in zip(xs, metas, os_):
        omin, obits, odata = oracle.float_encode(x, 0.0, 250.0, 250000, True, literal=False)
        assert (m.min, m.bits, m.nbytes) == (omin, obits, odata.size)
        assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes()


# ------------------------------------------------------------------------------------------------
# BASELINE-size checks through size-independent properties
# ------------------------------------------------------------------------------------------------
def test_c2_full_size_properties(mnw, ctx, oracle, torch_mod):
    """BASELINE config 2: float32 position column, 1e8 particles, blocks of 2^24.
    Stream bytes of the first block are compared with the oracle; all blocks are checked through
    the round trip (decoded integers re-encode to the same stream; |x - x'| <= dx)."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    N, B = 10**8, 2**24
    low, high, dx, pixels = 0.0, 250.0, 0.001, 250000
    g = torch.Generator(device=dev); g.manual_seed(0xACEDAD)
    x = torch.randn(N, device=dev, dtype=torch.float32, generator=g) * 20 + 125
    x = torch.remainder(x, 250.0)
    x[x >= 250.0] = 0.0
    nblk = (N + B - 1) // B
    outs = [torch.zeros(4 * min(B, N - k * B), dtype=torch.uint8, device=dev) for k in range(nblk)]
    jobs = [mnw.EncJob(x.data_ptr() + 4 * k * B, outs[k].data_ptr(), min(B, N - k * B), outs[k].numel(),
                       mnw.FLOAT_GROUP, low, high, pixels, 1) for k in range(nblk)]
    torch.cuda.synchronize()
    metas = ctx.encode_blocks(jobs)
    torch.cuda.synchronize()
    assert all(m.status == 0 for m in metas)
    # sigma 20 of 250: the arc exceeds pixels/2, so periodicMin returns 0 and min is the raw minimum
    assert all(m.bits == 18 and 0 <= m.min < 60000 for m in metas), [(m.min, m.bits) for m in metas]
    # oracle bytes for block 0 and the ragged last block
    for k in (0, nblk - 1):
        n = jobs[k].n
        xb = x[k * B: k * B + n].cpu().numpy()
        omin, obits, odata = oracle.float_encode(xb, low, high, pixels, True, literal=False)
        assert (metas[k].min, metas[k].bits, metas[k].nbytes) == (omin, obits, odata.size)
        assert outs[k][: odata.size].cpu().numpy().tobytes() == odata.tobytes()
    # decode everything and check the reference's tolerance
    y = torch.empty_like(x)
    djobs = [mnw.DecJob(outs[k].data_ptr(), y.data_ptr() + 4 * k * B, jobs[k].n, metas[k].nbytes, metas[k].min,
                        metas[k].bits, mnw.FLOAT_GROUP, low, high, pixels, 1, seed=1, block=k) for k in range(nblk)]
    torch.cuda.synchronize()
    ctx.decode_blocks(djobs)
    torch.cuda.synchronize()
    d = (y.double() - x.double()).abs()
    d = torch.minimum(d, 250.0 - d)
    assert float(d.max()) <= dx * 1.01 + 1e-4
    # idempotence: re-encoding the decoded column reproduces every stream byte for byte
    # (dithered values stay inside their pixel except where float32 rounding lands on the edge)
    q1 = torch.floor((x - low) / 0.001)
    q2 = torch.floor((y - low) / 0.001)
    frac_moved = float((q1 != q2).double().mean())
    assert frac_moved < 0.02


# ------------------------------------------------------------------------------------------------
# every float-encode structure must produce the reference's bytes: the single-read kernel (default for
# batches whose blocks fit on chip), the reduce + pack pair (large blocks, or single_read=0) and the
# opt-in fused launch
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["single_read", "two_pass", "fused"])
def test_encode_structures_bit_exact(mnw, oracle, torch_mod, mode):
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(21)
    low, high, pixels = 0.0, 250.0, 250000
    sizes = [1, 63, 8191, 8192, 8193, 16384, 16385, 40000, 100001, 300000, 2**20 + 17, 2**21 + 5]
    if mode == "single_read":
        sizes.append(2**24)                     # SURVEY 8d C2's default block: still read once (two-level record sweep)
    else:
        sizes.append(5 * 2**20)
    if mode == "two_pass":
        sizes.append(10**8)                     # a whole column as ONE block: more than the chip holds, two-pass in any mode
    xs = []
    for k, n in enumerate(sizes):
        c, sg = [(125.0, 20.0), (0.2, 0.4), (249.8, 0.5), (60.0, 0.01)][k % 4]
        x = np.mod(c + sg * rng.standard_normal(n), 250.0).astype(np.float32)
        x[x >= np.float32(250.0)] = 0.0        # exact zeros: quotient 0 must not take the slow path wrongly
        xs.append(x)
    xs.append(np.full(70000, 25.0, dtype=np.float32))                      # constant block: bits == 0
    bad = np.mod(125 + 5 * rng.standard_normal(50000), 250).astype(np.float32); bad[777] = np.nan
    xs.append(bad)                                                         # exact path inside a batch
    tiny = np.mod(125 + 5 * rng.standard_normal(20000), 250).astype(np.float32); tiny[5] = np.float32(1e-42)
    xs.append(tiny)                                                        # subnormal quotient: IEEE-division redo
    with mnw.Context(0) as c2:
        if mode == "fused":
            c2.set_option("fused", 1)
        elif mode == "two_pass":
            c2.set_option("single_read", 0)
        c2.set_option("group_bytes", 3 << 20)      # several groups for the fused form
        ts = [torch.from_numpy(x).to(dev) for x in xs]
        outs = [torch.zeros((8 if x.size < 10**6 else 4) * x.size + 16, dtype=torch.uint8, device=dev) for x in xs]   # the NaN block needs 64 bits
        jobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.FLOAT_GROUP, low, high, pixels, 1)
                for t, o in zip(ts, outs)]
        torch.cuda.synchronize()
        c2.timing_enable(True)
        c2.timing_reset()
        wants = [oracle.float_encode(x, low, high, pixels, True, literal=False) for x in xs]
        for rep in range(3):                        # slots are reused: stale sync words would show up here
            metas = c2.encode_blocks(jobs)
            torch.cuda.synchronize()
            for (omin, obits, odata), m, o in zip(wants, metas, outs):
                assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0)
                assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes()
        # the structure under test is the one that ran
        kt = c2.timing_get()
        ran = {k for k, v in kt.items() if v["launches"]}
        want = {"single_read": "k_encode_reg_f32", "two_pass": "k_pack_f32", "fused": "k_encode_f32"}[mode]
        assert want in ran, (mode, sorted(ran))
        if mode == "single_read":
            assert "k_reduce_f32" not in ran or kt["k_reduce_f32"]["launches"] <= 3   # only the exact-path reruns


def test_single_read_abort_and_host_fallback(mnw, oracle, torch_mod):
    """If the workgroups of a block cannot all make progress (another kernel holding part of the GPU), some record
    is never published.  Simulated with the test hook that drops one item's record: the kernel must give up after
    its bounded wait, raise the pinned abort word, and the host must rerun the batch through the reduce + pack
    launches.  Same bytes."""
    import time
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(24)
    low, high, pixels = 0.0, 250.0, 250000
    xs = [np.mod(125 + 20 * rng.standard_normal(2**20), 250.0).astype(np.float32) for _ in range(6)]
    with mnw.Context(0) as c2:
        ts = [torch.from_numpy(x).to(dev) for x in xs]
        outs = [torch.zeros(4 * x.size + 16, dtype=torch.uint8, device=dev) for x in xs]
        jobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.FLOAT_GROUP, low, high, pixels, 1)
                for t, o in zip(ts, outs)]
        torch.cuda.synchronize()
        c2.set_option("single_read_backoff", 0)           # (this test wants every batch to try the single-read kernel)
        c2.set_option("reg_test_drop_item", 300)          # an item of the third block
        c2.timing_enable(True)
        t0 = time.perf_counter()
        metas = c2.encode_blocks(jobs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kt = c2.timing_get()
        assert kt["k_encode_reg_f32"]["launches"] == 1 and kt["k_pack_f32"]["launches"] >= 1     # gave up, then the fallback ran
        assert 0.04 < dt < 5.0
        for x, m, o in zip(xs, metas, outs):
            omin, obits, odata = oracle.float_encode(x, low, high, pixels, True, literal=False)
            assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0)
            assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes()
        # and the context is healthy afterwards
        c2.set_option("reg_test_drop_item", -1)
        c2.timing_reset()
        metas2 = c2.encode_blocks(jobs)
        torch.cuda.synchronize()
        assert [(m.min, m.bits, m.nbytes) for m in metas2] == [(m.min, m.bits, m.nbytes) for m in metas]
        assert c2.timing_get()["k_pack_f32"]["launches"] == 0


def _int_blocks(rng):
    """int64 blocks of every width class and ragged lengths, as one batch: (values, ...)."""
    xs = []
    for b, n in ((0, 5000), (1, 4096), (7, 4097), (21, 2**20), (21, 2**20 - 1), (32, 70001), (33, 4095), (37, 2**18 + 13),
                 (48, 8192), (52, 12289), (63, 31), (64, 100000), (11, 2**21), (16, 1), (45, 2**21 - 4097), (20, 2**19)):
        off = int(rng.integers(-10**12, 10**12))
        if b == 64:
            x = rng.integers(-(2**63), 2**63 - 2, size=n, dtype=np.int64)
            x[:2] = (-(2**63), 2**63 - 2)
        elif b == 0:
            x = np.full(n, off, dtype=np.int64)
        else:
            x = off + rng.integers(0, 2**b, size=n, dtype=np.uint64).astype(np.int64)
        xs.append(x)
    return xs


def test_int_single_read_batch_bit_exact(mnw, oracle, torch_mod):
    """k_encode_reg_i64 (int64 blocks read once): a batch of blocks of every width class (0, <= 32, 33..64), ragged
    lengths, up to the kernel's 2^21-element limit, alone and next to float blocks; the same batch with the option off
    (k_reduce_i64 + k_pack_i64) gives the same bytes; the oracle (go/group.go:242-255, bit.go:79-159) checks both."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    xs = _int_blocks(rng)
    want = [oracle.int_encode(x, literal=False) for x in xs]
    fl = [np.mod(125 + 20 * rng.standard_normal(n), 250.0).astype(np.float32) for n in (2**20, 12345)]
    fwant = [oracle.float_encode(x, 0.0, 250.0, 250000, True, literal=False) for x in fl]
    with mnw.Context(0) as c2:
        ts = [torch.from_numpy(x).to(dev) for x in xs]
        outs = [torch.zeros(8 * x.size + 16, dtype=torch.uint8, device=dev) for x in xs]
        fts = [torch.from_numpy(x).to(dev) for x in fl]
        fouts = [torch.zeros(4 * x.size + 16, dtype=torch.uint8, device=dev) for x in fl]
        ijobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.INT_GROUP) for t, o in zip(ts, outs)]
        fjobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1)
                 for t, o in zip(fts, fouts)]
        torch.cuda.synchronize()
        for mode, with_floats in (("reg", False), ("reg", True), ("two_pass", True), ("reg_float_two_pass", True), ("reg", True)):
            c2.set_option("single_read_int", 0 if mode == "two_pass" else 1)
            c2.set_option("single_read", 0 if mode == "reg_float_two_pass" else 1)
            for o in outs + fouts:
                o.zero_()
            torch.cuda.synchronize()
            c2.timing_enable(True); c2.timing_reset(); c2.counters_reset()
            # int and float jobs interleaved in caller order
            jobs = ijobs[:5] + (fjobs[:1] if with_floats else []) + ijobs[5:] + (fjobs[1:] if with_floats else [])
            metas = c2.encode_blocks(jobs)
            torch.cuda.synchronize()
            kt = c2.timing_get()
            ran = {k for k, v in kt.items() if v["launches"]}
            cnt = c2.counters()
            if mode == "two_pass":
                assert "k_encode_reg_i64" not in ran and "k_pack_i64" in ran and cnt["int_single_read_batches"] == 0
            else:
                assert "k_encode_reg_i64" in ran and "k_pack_i64" not in ran and cnt["int_single_read_batches"] == 1, (mode, sorted(ran), cnt)
            assert cnt["single_read_aborts"] == 0
            if with_floats:
                assert ("k_pack_f32" if mode == "reg_float_two_pass" else "k_encode_reg_f32") in ran
            im = [m for m, j in zip(metas, jobs) if j.kind == mnw.INT_GROUP]
            fm = [m for m, j in zip(metas, jobs) if j.kind == mnw.FLOAT_GROUP]
            for x, m, o, (omin, obits, odata) in zip(xs, im, outs, want):
                assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), (mode, x.size, m.min, m.bits, omin, obits)
                assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes(), (mode, x.size, obits)
                assert int(o[m.nbytes:].max()) == 0                                   # nothing written past the stream
            for x, m, o, (omin, obits, odata) in zip(fl, fm, fouts, fwant):
                assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0)
                assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes()
        # blocks beyond the flat sweep's reach (more than 512 items of 4,096): read once too, their descriptors come from the
        # sweeper workgroups (round 3); next to a small block
        big = rng.integers(5, 2**27, size=2**21 + 4097, dtype=np.int64)
        tb = torch.from_numpy(big).to(dev)
        ob = torch.zeros(8 * big.size + 16, dtype=torch.uint8, device=dev)
        c2.timing_reset(); c2.counters_reset()
        m = c2.encode_blocks([mnw.EncJob(tb.data_ptr(), ob.data_ptr(), big.size, ob.numel(), mnw.INT_GROUP), ijobs[3]])
        torch.cuda.synchronize()
        assert c2.counters()["int_single_read_batches"] == 1 and c2.timing_get()["k_pack_i64"]["launches"] == 0
        omin, obits, odata = oracle.int_encode(big, literal=False)
        assert (m[0].min, m[0].bits, m[0].nbytes) == (omin, obits, odata.size)
        assert ob[: m[0].nbytes].cpu().numpy().tobytes() == odata.tobytes()
        assert (m[1].min, m[1].bits, m[1].nbytes) == (want[3][0], want[3][1], want[3][2].size)
        # the domain error (go/bit.go:19-21 with max - min = 2^64 - 1) and a capacity error, per block, next to good blocks
        bad = torch.tensor([-(2**63), 2**63 - 1, 5], dtype=torch.int64, device=dev)
        obad = torch.zeros(64, dtype=torch.uint8, device=dev)
        small = torch.zeros(16, dtype=torch.uint8, device=dev)
        c2.counters_reset()
        m = c2.encode_blocks([ijobs[1], mnw.EncJob(bad.data_ptr(), obad.data_ptr(), 3, 64, mnw.INT_GROUP),
                              mnw.EncJob(ts[3].data_ptr(), small.data_ptr(), ts[3].numel(), 16, mnw.INT_GROUP), ijobs[2]], check=False)
        torch.cuda.synchronize()
        assert c2.counters()["int_single_read_batches"] == 1
        assert [x.status for x in m] == [0, -6, -5, 0]
        assert int(small.max()) == 0 and int(obad.max()) == 0
        for k, mm in ((1, m[0]), (2, m[3])):
            assert outs[k][: mm.nbytes].cpu().numpy().tobytes() == want[k][2].tobytes()


def test_int_large_blocks_read_once(mnw, oracle, torch_mod):
    """int64 blocks of 2^21 .. 2^23 elements through k_encode_reg_i64 ONLY (descriptor records from its sweeper workgroups): a
    batch of the largest block the chip holds (two items of 4,096 in each of the 1,024 - 8 item workgroups of an MI355X: 8,323,072
    elements), a ragged 2^22 + 12,345 one of more than 32 bits,
    a constant one, small blocks in between; bytes equal to the oracle's (go/group.go:242-255) and to the two-launch path's.
    One element more than 2 x (item workgroups) x 4,096 keeps reduce + pack; so does big_blocks = 0; a record that never
    appears makes the launch give up and the call is redone by the two launches."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2023)
    nmax = 2 * (1024 - 8) * 4096
    xs = [rng.integers(-3, 2**19, size=nmax, dtype=np.int64),
          rng.integers(7, 100, size=5000, dtype=np.int64),
          rng.integers(-2**40, 2**41, size=2**22 + 12345, dtype=np.int64),
          np.full(2**21 + 1, 123456789012, dtype=np.int64),
          rng.integers(0, 2**31, size=2**20, dtype=np.int64)]
    xs[0][-1] = 2**20 + 5          # the maximum in the block's last item, the minimum in its first
    xs[0][0] = -77
    want = [oracle.int_encode(x, literal=False) for x in xs]
    with mnw.Context(0) as c2:
        ts = [torch.from_numpy(x).to(dev) for x in xs]
        outs = [torch.zeros(8 * x.size + 16, dtype=torch.uint8, device=dev) for x in xs]
        jobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.INT_GROUP) for t, o in zip(ts, outs)]
        torch.cuda.synchronize()
        for mode in ("reg", "two_pass", "no_big", "reg", "drop"):
            c2.set_option("single_read_int", 0 if mode == "two_pass" else 1)
            c2.set_option("big_blocks", 0 if mode == "no_big" else 1)
            c2.set_option("single_read_backoff", 0)
            c2.set_option("reg_test_drop_item", 2032 + 2 + 700 if mode == "drop" else -1)      # an item of the third block
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            c2.timing_enable(True); c2.timing_reset(); c2.counters_reset()
            metas = c2.encode_blocks(jobs)
            torch.cuda.synchronize()
            ran = {k for k, v in c2.timing_get().items() if v["launches"]}
            cnt = c2.counters()
            if mode == "reg":
                assert "k_encode_reg_i64" in ran and "k_pack_i64" not in ran and "k_reduce_i64" not in ran, (mode, sorted(ran))
                assert cnt["int_single_read_batches"] == 1 and cnt["single_read_aborts"] == 0
            elif mode == "drop":
                assert "k_encode_reg_i64" in ran and "k_pack_i64" in ran and cnt["single_read_aborts"] == 1, (sorted(ran), cnt)
            else:
                assert "k_encode_reg_i64" not in ran and "k_pack_i64" in ran, (mode, sorted(ran))
            for x, m, o, (omin, obits, odata) in zip(xs, metas, outs, want):
                assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), (mode, x.size, m.min, m.bits, omin, obits)
                assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes(), (mode, x.size, obits)
                assert int(o[m.nbytes:].max()) == 0
        c2.set_option("reg_test_drop_item", -1)
        # beyond what two items per workgroup hold: the two launches
        del ts, outs
        n = nmax + 1
        huge = rng.integers(0, 1000, size=n, dtype=np.int64)
        th = torch.from_numpy(huge).to(dev)
        oh = torch.zeros(8 * n + 16, dtype=torch.uint8, device=dev)
        c2.timing_reset(); c2.counters_reset()
        m = c2.encode_blocks([mnw.EncJob(th.data_ptr(), oh.data_ptr(), n, oh.numel(), mnw.INT_GROUP)])
        torch.cuda.synchronize()
        assert c2.counters()["int_single_read_batches"] == 0 and c2.timing_get()["k_pack_i64"]["launches"] >= 1
        omin, obits, odata = oracle.int_encode(huge, literal=False)
        assert (m[0].min, m[0].bits, m[0].nbytes) == (omin, obits, odata.size)
        assert oh[: m[0].nbytes].cpu().numpy().tobytes() == odata.tobytes()


def test_int_single_read_abort_and_host_fallback(mnw, oracle, torch_mod):
    """The int kernel's give-up path (see test_single_read_abort_and_host_fallback): one item's record is never
    published, the launch must end after its bounded wait and the host must rerun the int blocks -- and the float
    blocks of the same batch -- through the reduce + pack launches.  Same bytes."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(78)
    xs = [1000 + rng.integers(0, 2**b, size=n, dtype=np.int64) for b, n in ((21, 2**20), (37, 2**20), (8, 2**19 + 5))]
    fx = np.mod(125 + 20 * rng.standard_normal(2**20), 250.0).astype(np.float32)
    with mnw.Context(0) as c2:
        ts = [torch.from_numpy(x).to(dev) for x in xs]
        outs = [torch.zeros(8 * x.size + 16, dtype=torch.uint8, device=dev) for x in xs]
        ft = torch.from_numpy(fx).to(dev)
        fo = torch.zeros(4 * fx.size + 16, dtype=torch.uint8, device=dev)
        jobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.INT_GROUP) for t, o in zip(ts, outs)]
        jobs.append(mnw.EncJob(ft.data_ptr(), fo.data_ptr(), ft.numel(), fo.numel(), mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1))
        torch.cuda.synchronize()
        c2.set_option("single_read_backoff", 0)           # (this test wants every batch to try the single-read kernel)
        c2.set_option("reg_test_drop_item", 300)          # an item of the second int block; beyond the float block's 128 items
        c2.timing_enable(True)
        metas = c2.encode_blocks(jobs)
        torch.cuda.synchronize()
        kt = c2.timing_get()
        assert kt["k_encode_reg_i64"]["launches"] == 1 and kt["k_pack_i64"]["launches"] >= 1 and kt["k_pack_f32"]["launches"] >= 1
        assert c2.counters()["single_read_aborts"] == 1
        for x, m, o in zip(xs, metas, outs):
            omin, obits, odata = oracle.int_encode(x, literal=False)
            assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0)
            assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes()
        omin, obits, odata = oracle.float_encode(fx, 0.0, 250.0, 250000, True, literal=False)
        assert (metas[3].min, metas[3].bits, metas[3].nbytes) == (omin, obits, odata.size)
        assert fo[: odata.size].cpu().numpy().tobytes() == odata.tobytes()
        c2.set_option("reg_test_drop_item", -1)
        c2.timing_reset()
        metas2 = c2.encode_blocks(jobs)
        torch.cuda.synchronize()
        assert [(m.min, m.bits, m.nbytes) for m in metas2] == [(m.min, m.bits, m.nbytes) for m in metas]
        assert c2.timing_get()["k_pack_i64"]["launches"] == 0 and c2.counters()["single_read_aborts"] == 1


def _encode_and_compare(mnw, oracle, torch, c2, xs, low=0.0, high=250.0, pixels=250000, reps=1):
    dev = torch.device("cuda:0")
    ts = [torch.from_numpy(x).to(dev) for x in xs]
    # (a block with a NaN quantizes to MinInt64 somewhere and needs 64 bits per element: go/group.go:318-320 on amd64)
    outs = [torch.zeros((4 if np.isfinite(x).all() else 8) * x.size + 16, dtype=torch.uint8, device=dev) for x in xs]
    jobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.FLOAT_GROUP, low, high, pixels, 1)
            for t, o in zip(ts, outs)]
    want = [oracle.float_encode(x, low, high, pixels, True, literal=False) for x in xs]
    torch.cuda.synchronize()
    for _ in range(reps):
        metas = c2.encode_blocks(jobs)
        torch.cuda.synchronize()
        for k, (m, o, (omin, obits, odata)) in enumerate(zip(metas, outs, want)):
            assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), k
            assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes(), k
    return metas


def test_single_read_falls_back_for_blocks_beyond_the_chip(mnw, oracle, torch_mod):
    """A batch with a block of more items than the chip can hold (twice the grid of the single-read kernel: 2^24
    elements on a whole MI355X) takes the two-pass path as a whole -- same bytes either way.  With big_blocks = 0 that
    already happens above 2^22 elements."""
    rng = np.random.default_rng(22)
    xs = [np.mod(125 + 20 * rng.standard_normal(n), 250.0).astype(np.float32) for n in (2**24 + 8192 + 3, 5000)]
    with mnw.Context(0) as c2:
        c2.timing_enable(True)
        _encode_and_compare(mnw, oracle, torch_mod, c2, xs)
        kt = c2.timing_get()
        assert kt["k_pack_f32"]["launches"] >= 1 and not kt.get("k_encode_reg_f32", {"launches": 0})["launches"]
        assert c2.counters()["two_pass_batches"] == 1
    with mnw.Context(0) as c2:
        c2.set_option("big_blocks", 0)
        c2.timing_enable(True)
        _encode_and_compare(mnw, oracle, torch_mod, c2, [xs[0][: 2**22 + 8192 + 3], xs[1]])
        kt = c2.timing_get()
        assert kt["k_pack_f32"]["launches"] >= 1 and not kt.get("k_encode_reg_f32", {"launches": 0})["launches"]


@pytest.mark.parametrize("shape", ["2p24", "mixed", "one_and_a_half_grids", "wrapped_and_constant", "sweepers", "sweepers_from_257_items"])
def test_single_read_large_blocks_bit_exact(mnw, oracle, torch_mod, shape):
    """Blocks of more than 2^22 elements, up to twice the grid's items, are read once too (k_encode_reg_f32<.., BIG>: descriptor
    records from sweeper workgroups when every block has at most 2 x (grid - 8) items -- the last three shapes --, the two-level
    record sweep otherwise): same (min, bits) and bytes as the oracle; the launch that ran is the single-read one."""
    rng = np.random.default_rng(26)

    def col(n, c=125.0, sg=20.0):
        x = np.mod(c + sg * rng.standard_normal(n), 250.0).astype(np.float32)
        x[x >= np.float32(250.0)] = 0.0
        return x

    if shape == "2p24":
        xs = [col(2**24), col(2**24, 60.0, 3.0), col(2**24 - 5)]                    # SURVEY 8d C2's default block size, three of them
    elif shape == "mixed":
        xs = [col(2**20), col(2**23 + 12345), col(77), col(2**22 + 8192 + 3, 249.9, 0.5), col(2**20 + 1), col(2**24), col(3 * 8192)]
    elif shape == "one_and_a_half_grids":
        xs = [col(1536 * 8192 + 17), col(1025 * 8192), col(513 * 8192 - 1), col(2047 * 8192 + 1)]
    elif shape in ("sweepers", "sweepers_from_257_items"):
        # the largest block the sweepers take on an MI355X (2 x 1,016 items), one and a half grids, just above the flat sweep's 512
        # items, small blocks in between, a 300-item block (flat, or a sweeper's with float_flat_items = 256)
        xs = [col(2032 * 8192), col(1000), col(1536 * 8192 + 17), col(513 * 8192 - 1, 249.9, 0.5), col(2**20), col(300 * 8192 - 7), col(1017 * 8192)]
    else:
        a = col(2**23 + 999, 0.1, 0.3)                                              # ring wrapped: pmin != 0
        b = np.full(2**23 + 4096, 25.0, dtype=np.float32)                           # constant: bits == 0
        c = col(2**23); c[123456] = np.nan                                          # exact path behind a large block
        xs = [a, b, c]
    with mnw.Context(0) as c2:
        c2.timing_enable(True)
        if shape == "sweepers_from_257_items":
            c2.set_option("float_flat_items", 256)
        _encode_and_compare(mnw, oracle, torch_mod, c2, xs, reps=2)
        kt = c2.timing_get()
        assert kt["k_encode_reg_f32"]["launches"] == 2, kt
        if shape != "wrapped_and_constant":
            assert not kt.get("k_pack_f32", {"launches": 0})["launches"]
        cn = c2.counters()
        assert cn["single_read_batches"] == 2 and cn["single_read_aborts"] == 0 and cn["two_pass_batches"] == 0


@pytest.mark.parametrize("n", [2**24, 2**23 + 5])
def test_single_read_large_block_abort_and_fallback(mnw, oracle, torch_mod, n):
    """The bounded wait of the two-level sweep (2^24 elements) and of the sweeper workgroups (2^23 + 5): a dropped item record
    makes the launch give up and the host rerun the batch through reduce + pack."""
    rng = np.random.default_rng(27)
    xs = [np.mod(125 + 20 * rng.standard_normal(n), 250.0).astype(np.float32)]
    with mnw.Context(0) as c2:
        c2.set_option("single_read_backoff", 0)           # (this test wants every batch to try the single-read kernel)
        c2.set_option("reg_test_drop_item", 1500 if n == 2**24 else 700)
        c2.timing_enable(True)
        _encode_and_compare(mnw, oracle, torch_mod, c2, xs)
        kt = c2.timing_get()
        assert kt["k_encode_reg_f32"]["launches"] == 1 and kt["k_pack_f32"]["launches"] >= 1
        assert c2.counters()["single_read_aborts"] == 1


def test_repeated_batches_reuse_device_tables(mnw, oracle, torch_mod):
    """A batch whose job table is byte-identical to the one its ring slot uploaded last skips the upload
    launch.  The data behind unchanged pointers may change, and a changed table must be uploaded again."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(23)
    low, high, pixels = 0.0, 250.0, 250000
    n = 50000

    def column(c):
        return np.mod(c + 3.0 * rng.standard_normal(n), 250.0).astype(np.float32)

    with mnw.Context(0) as c2:
        xa, xb = torch.from_numpy(column(40.0)).to(dev), torch.from_numpy(column(200.0)).to(dev)
        out = torch.zeros(4 * n + 16, dtype=torch.uint8, device=dev)
        y = torch.empty(n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()

        def check(xt):
            x = xt.cpu().numpy()
            omin, obits, odata = oracle.float_encode(x, low, high, pixels, True, literal=False)
            jobs = [mnw.EncJob(xt.data_ptr(), out.data_ptr(), n, out.numel(), mnw.FLOAT_GROUP, low, high, pixels, 1)]
            (m,) = c2.encode_blocks(jobs)
            torch.cuda.synchronize()
            assert (m.min, m.bits, m.nbytes) == (omin, obits, odata.size)
            assert out[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes()
            c2.decode_blocks([mnw.DecJob(out.data_ptr(), y.data_ptr(), n, m.nbytes, m.min, m.bits, mnw.FLOAT_GROUP,
                                         low, high, pixels, 1, seed=5, block=2)])
            torch.cuda.synchronize()
            ref = oracle.float_decode(odata, omin, obits, low, high, pixels, True, n, u=oracle.dither(5, 2, n))
            assert np.array_equal(y.cpu().numpy(), ref)

        for _ in range(7):                     # the ring has three slots: from the fourth call on, tables repeat
            check(xa)
        xa.copy_(torch.from_numpy(column(120.0)).to(dev))      # same pointer, new contents (also a new width/min)
        torch.cuda.synchronize()
        for _ in range(4):
            check(xa)
        for _ in range(4):                     # another source pointer: the table differs and is uploaded again
            check(xb)
            check(xa)


# ------------------------------------------------------------------------------------------------
# BASELINE configs 3 and 4 at reduced row counts (parity-test cases, not bench lines)
# ------------------------------------------------------------------------------------------------
def test_c3_mixed_halo_catalogue_columns(mnw, ctx, oracle, torch_mod):
    """SURVEY 8d C3 shape: 32 mixed columns (3 position, 3 velocity, 18 log/linear float, 4 int,
    the 4 fixed-size ones are raw copies and not codec work) x 1e5 rows, one batch per block of rows,
    exactly how minh.Writer.Block walks its columns (go/minh/minh.go:115-137)."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(31)
    rows = 100000
    cols = []
    for c in range(3):
        cols.append(("f", np.mod(125 + 20 * rng.standard_normal(rows), 250).astype(np.float32), (0.0, 250.0, 0.001)))
    for c in range(3):
        v = np.clip(300 * rng.standard_normal(rows), -4999, 4999).astype(np.float32)
        cols.append(("f", v, (-5000.0, 5000.0, 1.0)))
    for c in range(18):
        v = np.clip(10 + c * 0.2 + 0.8 * rng.standard_normal(rows), 8.001, 15.99).astype(np.float32)
        cols.append(("f", v, (8.0, 16.0, 0.01)))
    cols.append(("i", (5 * 10**9 + np.arange(rows)).astype(np.int64), None))
    cols.append(("i", rng.poisson(40, rows).astype(np.int64), None))
    cols.append(("i", rng.integers(-1, 2**20, rows).astype(np.int64), None))
    cols.append(("i", np.full(rows, 7, dtype=np.int64), None))
    jobs, outs, keep, params = [], [], [], []
    for kind, v, lim in cols:
        t = torch.from_numpy(v).to(dev)
        o = torch.zeros(8 * rows + 16, dtype=torch.uint8, device=dev)
        if kind == "f":
            low, high, dx = lim
            pixels = mnw.float_group_pixels(low, high, dx)
            assert pixels == oracle.float_group_pixels(low, high, dx)
            jobs.append(mnw.EncJob(t.data_ptr(), o.data_ptr(), rows, o.numel(), mnw.FLOAT_GROUP, low, high, pixels, 1))
            params.append((low, high, pixels))
        else:
            jobs.append(mnw.EncJob(t.data_ptr(), o.data_ptr(), rows, o.numel(), mnw.INT_GROUP))
            params.append(None)
        outs.append(o); keep.append(t)
    torch.cuda.synchronize()
    metas = ctx.encode_blocks(jobs)
    torch.cuda.synchronize()
    djobs, douts = [], []
    for k, ((kind, v, lim), m, o, pr) in enumerate(zip(cols, metas, outs, params)):
        if kind == "f":
            omin, obits, odata = oracle.float_encode(v, pr[0], pr[1], pr[2], True, literal=False)
            y = torch.empty(rows, dtype=torch.float32, device=dev)
            djobs.append(mnw.DecJob(o.data_ptr(), y.data_ptr(), rows, m.nbytes, m.min, m.bits, mnw.FLOAT_GROUP, pr[0], pr[1], pr[2], 1, 3, k))
        else:
            omin, obits, odata = oracle.int_encode(v, literal=False)
            y = torch.empty(rows, dtype=torch.int64, device=dev)
            djobs.append(mnw.DecJob(o.data_ptr(), y.data_ptr(), rows, m.nbytes, m.min, m.bits, mnw.INT_GROUP))
        assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), k
        assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes(), k
        douts.append(y)
    torch.cuda.synchronize()
    ctx.decode_blocks(djobs)
    torch.cuda.synchronize()
    for (kind, v, lim), y in zip(cols, douts):
        got = y.cpu().numpy()
        if kind == "f":
            assert np.all(np.abs(got.astype(np.float64) - v.astype(np.float64)) <= lim[2] * 1.01 + 1e-3)
        else:
            assert np.array_equal(got, v)


def test_c4_particle_snapshot_blocks(mnw, ctx, oracle, torch_mod):
    """SURVEY 8d C4 shape, scaled: (x,y,z) Lagrangian grid + displacement wrapped into [0,250), periodic,
    and (vx,vy,vz) non-periodic with data-derived limits (go/minp/minp.go:87-96), 27 sub-cell blocks of
    64^3 particles per component -- one minnow group of many blocks per component (minp.go:112-118)."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(41)
    nsub, nblk = 64**3, 27
    jobs, outs, keep, spec = [], [], [], []
    for comp in range(6):
        for b in range(nblk):
            if comp < 3:
                grid = (np.arange(nsub) % 64) * (250.0 / 192) + (b % 3) * (250.0 / 3)
                v = np.mod(grid + 2 * rng.standard_normal(nsub), 250.0).astype(np.float32)
                v[v >= np.float32(250.0)] = 0.0
                low, high, pixels, per = 0.0, 250.0, 250000, 1
            else:
                v = (300 * rng.standard_normal(nsub)).astype(np.float32)
                low, high = float(v.min()), float(np.nextafter(v.max(), np.float32(np.inf)))
                pixels, per = mnw.float_group_pixels(low, high, 1.0), 0
            t = torch.from_numpy(v).to(dev)
            o = torch.zeros(4 * nsub + 16, dtype=torch.uint8, device=dev)
            jobs.append(mnw.EncJob(t.data_ptr(), o.data_ptr(), nsub, o.numel(), mnw.FLOAT_GROUP, low, high, pixels, per))
            outs.append(o); keep.append(t); spec.append((v, low, high, pixels, per))
    torch.cuda.synchronize()
    metas = ctx.encode_blocks(jobs)
    torch.cuda.synchronize()
    for k in range(0, len(jobs), 7):           # every 7th block against the oracle, byte for byte
        v, low, high, pixels, per = spec[k]
        omin, obits, odata = oracle.float_encode(v, low, high, pixels, bool(per), literal=False)
        m = metas[k]
        assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), k
        assert outs[k][: m.nbytes].cpu().numpy().tobytes() == odata.tobytes(), k
    ys = [torch.empty(nsub, dtype=torch.float32, device=dev) for _ in jobs]
    djobs = [mnw.DecJob(outs[k].data_ptr(), ys[k].data_ptr(), nsub, metas[k].nbytes, metas[k].min, metas[k].bits,
                        mnw.FLOAT_GROUP, spec[k][1], spec[k][2], spec[k][3], spec[k][4], 9, k) for k in range(len(jobs))]
    torch.cuda.synchronize()
    ctx.decode_blocks(djobs)
    torch.cuda.synchronize()
    for k in range(len(jobs)):
        v, low, high, pixels, per = spec[k]
        dxq = (np.float32(high) - np.float32(low)) / np.float32(pixels)
        d = np.abs(ys[k].cpu().numpy().astype(np.float64) - v.astype(np.float64))
        if per:
            d = np.minimum(d, 250.0 - d)
        assert d.max() <= float(dxq) * 1.01 + 1e-3, (k, d.max())


def test_randomised_differential_short(mnw):
    """a few seconds of tests/fuzz_parity.py (random blocks, parameters, batch shapes and encode structures against the
    oracle, minh value transforms included; the long runs are done by hand: `python tests/fuzz_parity.py 150 1` =
    112,627 blocks and `... 120 7` = 67,821 blocks without a mismatch in round 1)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_parity
    cases, blocks = fuzz_parity.run(6.0, 2024)
    assert cases > 50 and blocks > 200


def test_many_small_blocks_in_one_batch(mnw, ctx, oracle, torch_mod):
    """3,000 blocks of 1..20,000 elements (and a few empty ones) in one call: partial items only or mostly, several
    items of different blocks per workgroup trip, int and float mixed"""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(314)
    low, high, pixels = 0.0, 250.0, 250000
    sizes = rng.integers(1, 20000, 3000)
    sizes[::97] = 0
    sizes[1::211] = 8192
    xs, jobs, keep = [], [], []
    total = int(sizes.sum()) + 4 * len(sizes) + 16            # blocks start on multiples of four elements
    big_f = torch.from_numpy(np.mod(125 + 30 * rng.standard_normal(total), 250.0).astype(np.float32)).to(dev)
    big_i = torch.from_numpy(rng.integers(-5000, 5000, total).astype(np.int64)).to(dev)
    hf, hi = big_f.cpu().numpy(), big_i.cpu().numpy()
    out = torch.zeros(int(8 * sizes.sum() + 32 * len(sizes)), dtype=torch.uint8, device=dev)
    off = ooff = 0
    for k, n in enumerate(sizes):
        n = int(n)
        a0 = (off + 3) // 4 * 4                      # float blocks start 16-byte aligned
        cap = (8 * n + 31) // 16 * 16
        if k % 5 == 4:
            jobs.append(mnw.EncJob(big_i.data_ptr() + 8 * a0, out.data_ptr() + ooff, n, cap, mnw.INT_GROUP))
            xs.append(("i", hi[a0:a0 + n]))
        else:
            jobs.append(mnw.EncJob(big_f.data_ptr() + 4 * a0, out.data_ptr() + ooff, n, cap, mnw.FLOAT_GROUP, low, high, pixels, 1))
            xs.append(("f", hf[a0:a0 + n]))
        keep.append((ooff, cap))
        off = a0 + n
        ooff += cap
    torch.cuda.synchronize()
    metas = ctx.encode_blocks(jobs)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for (kind, x), m, (o0, cap) in zip(xs, metas, keep):
        if x.size == 0:
            assert (m.bits, m.nbytes, m.status) == (0, 0, 0)
            continue
        omin, obits, odata = oracle.float_encode(x, low, high, pixels, True, literal=False) if kind == "f" else oracle.int_encode(x, literal=False)
        assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0)
        assert host[o0: o0 + m.nbytes].tobytes() == odata.tobytes()


# ------------------------------------------------------------------------------------------------
# the headline launch itself, block by block against the oracle: bench.py's column, bench.py's block size,
# one batch = one k_encode_reg_f32 launch (asserted), then the decode of the same batch
# ------------------------------------------------------------------------------------------------
def _headline_column(torch, n, seed=0xACEDAD, chunk=1 << 28):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.empty(n, device=dev, dtype=torch.float32)
    for c0 in range(0, n, chunk):
        v = x[c0: c0 + chunk]
        v.normal_(0.0, 1.0, generator=g)
        v.mul_(20.0).add_(125.0)
        torch.remainder(v, 250.0, out=v)
        v[v >= 250.0] = 0.0
    return x


def _headline_check(mnw, oracle, torch, N, B, check_blocks, expect_kernel="k_encode_reg_f32"):
    """Encode the whole column as ONE batch, decode it as one batch, and compare the blocks in `check_blocks` -- their
    (min, bits, nbytes), every stream byte and every decoded float -- with the oracle."""
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.device("cuda:0")
    low, high, pixels = 0.0, 250.0, 250000
    x = _headline_column(torch, N)
    y = torch.empty_like(x)
    nblk = (N + B - 1) // B
    slot = (4 * B + 15) // 16 * 16
    packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    ctx = mnw.Context(0)
    try:
        jobs = [mnw.EncJob(x.data_ptr() + 4 * k * B, packed.data_ptr() + k * slot, min(B, N - k * B), slot,
                           mnw.FLOAT_GROUP, low, high, pixels, 1) for k in range(nblk)]
        torch.cuda.synchronize()
        ctx.timing_enable(True)
        ctx.timing_reset()
        metas = ctx.encode_blocks(jobs)
        kt = ctx.timing_get()
        ran = {k: v["launches"] for k, v in kt.items() if v["launches"]}
        assert ran == {expect_kernel: 1} if isinstance(expect_kernel, str) else ran == expect_kernel, ran
        assert ctx.counters()["single_read_aborts"] == 0
        assert all(m.status == 0 for m in metas)
        djobs = [mnw.DecJob(packed.data_ptr() + k * slot, y.data_ptr() + 4 * k * B, jobs[k].n, metas[k].nbytes, metas[k].min,
                            metas[k].bits, mnw.FLOAT_GROUP, low, high, pixels, 1, seed=7, block=k) for k in range(nblk)]
        ctx.decode_blocks(djobs)
        torch.cuda.synchronize()
        assert ctx.timing_get()["k_unpack_f32"]["launches"] == 1

        def one(k):
            n = jobs[k].n
            xb = x[k * B: k * B + n].cpu().numpy()
            omin, obits, odata = oracle.float_encode(xb, low, high, pixels, True, literal=False)
            if (metas[k].min, metas[k].bits, metas[k].nbytes) != (omin, obits, odata.size):
                return "block %d: (min, bits, nbytes) %r, oracle %r" % (k, (metas[k].min, metas[k].bits, metas[k].nbytes), (omin, obits, odata.size))
            got = packed[k * slot: k * slot + odata.size].cpu().numpy()
            if got.tobytes() != odata.tobytes():
                return "block %d: stream differs from the oracle" % k
            want = oracle.float_decode(odata, omin, obits, low, high, pixels, True, n, u=oracle.dither(7, k, n), literal=False)
            yb = y[k * B: k * B + n].cpu().numpy()
            if not np.array_equal(yb.view(np.uint32), want.view(np.uint32)):
                return "block %d: decoded floats differ from the oracle" % k
            d = np.abs(yb.astype(np.float64) - xb.astype(np.float64))
            d = np.minimum(d, 250.0 - d)
            if d.max() > 0.001 * 1.01 + 1e-4:
                return "block %d: |x - x'| > dx" % k
            return None

        with ThreadPoolExecutor(max_workers=8) as ex:          # the oracle's C calls release the GIL
            bad = [r for r in ex.map(one, check_blocks) if r]
        assert not bad, bad[:5]
    finally:
        ctx.close()
    return nblk


def test_headline_1e8_every_block_vs_oracle(mnw, oracle, torch_mod):
    """BASELINE configs[1] as bench.py runs it (1e8 particles, 96 blocks of 2^20): EVERY block of the single-read launch."""
    N, B = 10**8, 2**20
    nblk = (N + B - 1) // B
    assert _headline_check(mnw, oracle, torch_mod, N, B, range(nblk)) == 96


def test_headline_1e9_sampled_blocks_vs_oracle(mnw, oracle, torch_mod):
    """The metric's configuration (1e9 particles, 954 blocks of 2^20, one launch): every 16th block, the first and the
    (ragged) last one."""
    N, B = 10**9, 2**20
    nblk = (N + B - 1) // B
    blocks = sorted(set(range(0, nblk, 16)) | {0, nblk - 1})
    assert _headline_check(mnw, oracle, torch_mod, N, B, blocks) == 954


# ------------------------------------------------------------------------------------------------
# BASELINE configurations at FULL size (round 2): the launches bench.py and the tools time, checked against the oracle
# ------------------------------------------------------------------------------------------------
def test_c3_full_size_catalogue_sampled_blocks(mnw, oracle, torch_mod):
    """configs[2] at its full row count: 28 codec columns x 1e8 rows (3 positions, 3 velocities, 18 logarithm-like float columns,
    4 int columns; the 4 fixed-size columns of the shape are raw copies), one group per (2^20-row block, column) as
    minh.Writer.Block lays them out, 16 row blocks per call.  Every 131st (block, column) pair -- and the last, ragged row
    block of every column -- against the oracle: (min, bits, nbytes), stream bytes, decoded values."""
    from concurrent.futures import ThreadPoolExecutor
    torch = torch_mod
    dev = torch.device("cuda:0")
    rows, B = 10**8, 2**20
    nblk = (rows + B - 1) // B
    g = torch.Generator(device=dev); g.manual_seed(0xC3)
    cols = []                                           # (kind, tensor, (low, high, dx) or None)
    for c in range(3):
        v = torch.remainder(torch.randn(rows, device=dev, generator=g) * 20 + 125, 250.0)
        v[v >= 250.0] = 0.0
        cols.append(("f", v, (0.0, 250.0, 0.001)))
    for c in range(3):
        cols.append(("f", torch.clamp(torch.randn(rows, device=dev, generator=g) * 300, -4999, 4999), (-5000.0, 5000.0, 1.0)))
    for c in range(18):
        cols.append(("f", torch.clamp(10 + c * 0.2 + 0.8 * torch.randn(rows, device=dev, generator=g), 8.001, 15.99), (8.0, 16.0, 0.01)))
    cols.append(("i", torch.arange(rows, device=dev, dtype=torch.int64) + 5 * 10**9, None))
    cols.append(("i", torch.poisson(torch.full((rows,), 40.0, device=dev), generator=g).to(torch.int64), None))
    cols.append(("i", torch.randint(-1, 2**20, (rows,), device=dev, generator=g, dtype=torch.int64), None))
    cols.append(("i", torch.full((rows,), 7, device=dev, dtype=torch.int64), None))
    ncol = len(cols)
    # stream slots: 4 B per element for the float columns (pixels < 2^32), 8 B for the int columns
    slot = [(4 if k == "f" else 8) * B + 16 for k, _, _ in cols]
    packed = [torch.zeros(nblk * s, dtype=torch.uint8, device=dev) for s in slot]
    outs = [torch.empty_like(t) for _, t, _ in cols]
    pix = [mnw.float_group_pixels(*lim) if lim else 0 for _, _, lim in cols]
    metas = {}
    torch.cuda.synchronize()                             # the context's stream is not ordered with torch's: the columns must be complete
    with mnw.Context(0) as c2:
        for b0 in range(0, nblk, 16):                    # 16 row blocks x 28 columns = one batch
            jobs, keys = [], []
            for b in range(b0, min(b0 + 16, nblk)):
                n = min(B, rows - b * B)
                for k, (kind, t, lim) in enumerate(cols):
                    esz = 4 if kind == "f" else 8
                    if kind == "f":
                        jobs.append(mnw.EncJob(t.data_ptr() + esz * b * B, packed[k].data_ptr() + b * slot[k], n, slot[k], mnw.FLOAT_GROUP,
                                               lim[0], lim[1], pix[k], 1))
                    else:
                        jobs.append(mnw.EncJob(t.data_ptr() + esz * b * B, packed[k].data_ptr() + b * slot[k], n, slot[k], mnw.INT_GROUP))
                    keys.append((b, k))
            for key, m in zip(keys, c2.encode_blocks(jobs)):
                metas[key] = m
        assert all(m.status == 0 for m in metas.values())
        assert c2.counters()["single_read_aborts"] == 0
        for b0 in range(0, nblk, 16):
            djobs = []
            for b in range(b0, min(b0 + 16, nblk)):
                n = min(B, rows - b * B)
                for k, (kind, t, lim) in enumerate(cols):
                    m = metas[(b, k)]
                    esz = 4 if kind == "f" else 8
                    if kind == "f":
                        djobs.append(mnw.DecJob(packed[k].data_ptr() + b * slot[k], outs[k].data_ptr() + esz * b * B, n, m.nbytes, m.min, m.bits,
                                                mnw.FLOAT_GROUP, lim[0], lim[1], pix[k], 1, seed=11, block=b * ncol + k))
                    else:
                        djobs.append(mnw.DecJob(packed[k].data_ptr() + b * slot[k], outs[k].data_ptr() + esz * b * B, n, m.nbytes, m.min, m.bits,
                                                mnw.INT_GROUP))
            c2.decode_blocks(djobs)
        torch.cuda.synchronize()
    pairs = sorted(set([(b, k) for i, (b, k) in enumerate((b, k) for b in range(nblk) for k in range(ncol)) if i % 131 == 0] +
                       [(nblk - 1, k) for k in range(ncol)]))

    def one(bk):
        b, k = bk
        kind, t, lim = cols[k]
        n = min(B, rows - b * B)
        x = t[b * B: b * B + n].cpu().numpy()
        m = metas[bk]
        if kind == "f":
            omin, obits, odata = oracle.float_encode(x, lim[0], lim[1], pix[k], True, literal=False)
        else:
            omin, obits, odata = oracle.int_encode(x, literal=False)
        if (m.min, m.bits, m.nbytes) != (omin, obits, odata.size):
            return "%r: (min, bits, nbytes) %r, oracle %r" % (bk, (m.min, m.bits, m.nbytes), (omin, obits, odata.size))
        if packed[k][b * slot[k]: b * slot[k] + odata.size].cpu().numpy().tobytes() != odata.tobytes():
            return "%r: stream differs" % (bk,)
        y = outs[k][b * B: b * B + n].cpu().numpy()
        if kind == "f":
            want = oracle.float_decode(odata, omin, obits, lim[0], lim[1], pix[k], True, n, u=oracle.dither(11, b * ncol + k, n), literal=False)
            if not np.array_equal(y.view(np.uint32), want.view(np.uint32)):
                return "%r: decoded floats differ" % (bk,)
        elif not np.array_equal(y, x):
            return "%r: decoded ints differ" % (bk,)
        return None

    with ThreadPoolExecutor(max_workers=8) as ex:
        bad = [r for r in ex.map(one, pairs) if r]
    assert not bad, bad[:5]
    # and every float column as a whole: the reference's criterion (go/minnow_test.go:328-334)
    for k, (kind, t, lim) in enumerate(cols):
        if kind == "f":
            d = (outs[k].double() - t.double()).abs()
            if k < 3:
                d = torch.minimum(d, 250.0 - d)
            assert float(d.max()) <= lim[2] * 1.01 + 1e-3, k
        else:
            assert torch.equal(outs[k], t), k


def test_1e9_column_in_2p24_blocks_sampled(mnw, oracle, torch_mod):
    """The metric's column in SURVEY 8d C2's default block size: 60 blocks of 2^24 elements, read once by the large-block form of
    the single-read kernel (one launch, asserted); every 7th block, the first and the ragged last one against the oracle."""
    N, B = 10**9, 2**24
    nblk = (N + B - 1) // B
    blocks = sorted(set(range(0, nblk, 7)) | {0, nblk - 1})
    assert _headline_check(mnw, oracle, torch_mod, N, B, blocks) == 60


def test_copy_file_to_device(mnw, ctx, torch_mod, tmp_path):
    """mnw_copy_file_to_device: byte ranges of a file (what a reader fetches block by block, reader.go:121-126) into a device
    buffer, packed and at given offsets; ranges larger than a pinned chunk, empty ranges, a range past the end of the file."""
    import ctypes as C
    torch = torch_mod
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, size=45_000_001, dtype=np.uint8)
    fn = tmp_path / "ranges.bin"
    data.tofile(fn)
    fd = os.open(fn, os.O_RDONLY)
    try:
        offs = [0, 17, 1_000_003, 20_000_000, 44_999_990, 5]
        lens = [1, 4_000_001, 0, 24_999_990, 11, 19_000_000]
        dev = torch.zeros(sum(lens) + 64, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        ctx.copy_file_to_device(dev.data_ptr(), fd, offs, lens)
        want = np.concatenate([data[o:o + n] for o, n in zip(offs, lens)])
        got = dev.cpu().numpy()
        assert np.array_equal(got[: want.size], want) and not got[want.size:].any()
        # destination offsets: gaps stay untouched
        doff, at = [], 100
        for n in lens:
            doff.append(at); at += n + 37
        dev2 = torch.full((at + 64,), 255, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        ctx.copy_file_to_device(dev2.data_ptr(), fd, offs, lens, doff)
        got = dev2.cpu().numpy()
        ref = np.full(at + 64, 255, dtype=np.uint8)
        for o, n, d in zip(offs, lens, doff):
            ref[d:d + n] = data[o:o + n]
        assert np.array_equal(got, ref)
        # past the end of the file: an error, not a short copy
        with pytest.raises(mnw.MnwError) as ei:
            ctx.copy_file_to_device(dev.data_ptr(), fd, [44_999_000], [5000])
        assert ei.value.status == -8                      # MNW_E_IO: an I/O error, not a bad argument
        # ... also in the middle of a transfer of several pinned chunks (earlier chunks are already on their way when the read
        # fails): the call must have settled the ring, so the next one works and delivers the right bytes
        with pytest.raises(mnw.MnwError):
            ctx.copy_file_to_device(dev.data_ptr(), fd, [0, 44_000_000], [40_000_000, 2_000_000])
        dev.zero_()
        torch.cuda.synchronize()
        ctx.copy_file_to_device(dev.data_ptr(), fd, offs, lens)
        got = dev.cpu().numpy()
        assert np.array_equal(got[: want.size], want)
    finally:
        os.close(fd)


def test_single_read_backs_off_after_a_give_up(mnw, oracle, torch_mod):
    """A launch that gave up says something else holds part of the GPU: the next batch must not pay the 50 ms bound again
    but go to the two-launch path at once; a give-up right after the pause doubles it; a launch that runs to its end resets
    it.  Same bytes all along."""
    import time
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(31)
    xs = [np.mod(125 + 20 * rng.standard_normal(2**19), 250.0).astype(np.float32) for _ in range(3)]
    want = [oracle.float_encode(x, 0.0, 250.0, 250000, True, literal=False) for x in xs]
    with mnw.Context(0) as c2:
        ts = [torch.from_numpy(x).to(dev) for x in xs]
        outs = [torch.zeros(4 * x.size + 16, dtype=torch.uint8, device=dev) for x in xs]
        jobs = [mnw.EncJob(t.data_ptr(), o.data_ptr(), t.numel(), o.numel(), mnw.FLOAT_GROUP, 0.0, 250.0, 250000, 1) for t, o in zip(ts, outs)]
        torch.cuda.synchronize()

        def run():
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            metas = c2.encode_blocks(jobs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            for m, o, (omin, obits, odata) in zip(metas, outs, want):
                assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0)
                assert o[: m.nbytes].cpu().numpy().tobytes() == odata.tobytes()
            return dt

        run()
        assert c2.counters()["single_read_batches"] == 1
        c2.set_option("reg_test_drop_item", 70)
        assert run() > 0.04                                   # gives up after its bounded wait
        cnt = c2.counters()
        assert (cnt["single_read_aborts"], cnt["backoff_batches"]) == (1, 0)
        assert run() < 0.03                                   # one batch of pause: no attempt, no wait
        cnt = c2.counters()
        assert (cnt["single_read_aborts"], cnt["backoff_batches"], cnt["single_read_batches"]) == (1, 1, 2)
        assert run() > 0.04                                   # tries again, gives up again: the pause doubles
        assert c2.counters()["single_read_aborts"] == 2
        assert run() < 0.03 and run() < 0.03
        assert c2.counters()["backoff_batches"] == 3
        c2.set_option("reg_test_drop_item", -1)               # the co-tenant has gone
        assert run() < 0.03
        cnt = c2.counters()
        assert (cnt["single_read_aborts"], cnt["backoff_batches"], cnt["single_read_batches"]) == (2, 3, 4)
        c2.set_option("reg_test_drop_item", 70)               # and the next give-up starts from a pause of one again
        run(); run()
        c2.set_option("reg_test_drop_item", -1)
        run()
        cnt = c2.counters()
        assert (cnt["single_read_aborts"], cnt["backoff_batches"], cnt["single_read_batches"]) == (3, 4, 6)
