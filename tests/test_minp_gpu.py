"""minp particle vectors (go/minp/minp.go): the sub-cell gather / scatter / bounds kernels and
the Vectors write/read paths against the oracle's literal restatement, on the shapes of the reference's own
tests (go/minp/minp_test.go:7-73, 168-207) and on random grids."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_vectors(offset, L, n_file):
    """makeVectors of minp_test.go:139-166 (float32 arithmetic)"""
    n3 = n_file ** 3
    i = np.arange(n3)
    ix, iy, iz = i % n_file, (i // n_file) % n_file, i // (n_file * n_file)
    dx = np.float32(L) / np.float32(n_file)
    vec = np.stack([np.float32(offset[k]) + c.astype(np.float32) * dx for k, c in enumerate((ix, iy, iz))], axis=1)
    vec = vec.astype(np.float32)
    vec[vec > np.float32(L)] -= np.float32(L)
    return np.ascontiguousarray(vec)


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


@pytest.fixture()
def ctx2(mnw):
    with mnw.Context(0) as c:
        yield c


@pytest.mark.parametrize("n_sub,sub_cells", [(2, 3), (1, 1), (4, 2), (5, 2), (8, 1), (12, 3)])
def test_get_set_sub_cell(mnw, ctx2, oracle, torch_mod, n_sub, sub_cells):
    """TestGetSubCell (minp_test.go:168-207) generalised: gather == getSubCell for every sub-cell, scatter inverts it"""
    torch = torch_mod
    dev = torch.device("cuda:0")
    nf = n_sub * sub_cells
    rng = np.random.default_rng(nf)
    vec = rng.standard_normal((nf ** 3, 3)).astype(np.float32)
    t = torch.from_numpy(vec).to(dev)
    soa = torch.empty(3 * nf ** 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx2.subcell_gather(t.data_ptr(), nf, sub_cells, soa.data_ptr())
    torch.cuda.synchronize()
    got = soa.cpu().numpy().reshape(3, sub_cells ** 3, n_sub ** 3)
    for sc in range(sub_cells ** 3):
        ref = oracle.get_sub_cell(vec, sc, sub_cells, n_sub)
        for k in range(3):
            assert np.array_equal(got[k, sc], ref[k]), (sc, k)
    back = torch.zeros_like(t)
    torch.cuda.synchronize()
    ctx2.subcell_scatter(soa.data_ptr(), nf, sub_cells, False, 0.0, back.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(back.cpu().numpy(), vec)


def test_bounds_and_limits(mnw, ctx2, oracle, torch_mod):
    from minnow_amd import minp
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for n in (1, 3, 4, 5, 1000, 4099):
        vec = (300 * rng.standard_normal((n, 3))).astype(np.float32)
        vec[:, 2] -= 2000.0                    # an all-negative component: Nextafter32(max, 2*max) steps DOWN (minp.go:94)
        t = torch.from_numpy(vec).to(dev)
        torch.cuda.synchronize()
        mn, mx = ctx2.vectors_bounds(t.data_ptr(), n)
        omn, omx = oracle.bounds3(vec)
        assert np.array_equal(np.float32(mn), omn) and np.array_equal(np.float32(mx), omx)
        lo, hi = minp.limits(ctx2, t.data_ptr(), n, False, 0.0)
        olo, ohi = oracle.minp_limits(vec, False, 0.0)
        assert np.array_equal(np.float32(lo), olo) and np.array_equal(np.float32(hi), ohi)


@pytest.mark.parametrize("n_side,sub_cells", [(1, 1), (2, 1), (8, 1), (10, 1), (2, 2), (10, 2), (10, 5)])
@pytest.mark.parametrize("periodic", [False, True])
def test_vec_reader_writer_shapes(mnw, ctx2, oracle, torch_mod, n_side, sub_cells, periodic):
    """TestVecReaderWriter (minp_test.go:7-73): same grids, dx and box; blocks bit-exact with the oracle's Vectors,
    decoded vectors bit-exact with the oracle's reader (same dither) and within dx of the input as the reference asserts"""
    from minnow_amd import minp
    torch = torch_mod
    dev = torch.device("cuda:0")
    dx, L = 0.1, 100.0
    nf = n_side
    vec = make_vectors((0, 0, 0), L, nf)
    _roundtrip(mnw, ctx2, oracle, torch, dev, vec, nf, sub_cells, dx, periodic, L)


def test_vectors_random_grid(mnw, ctx2, oracle, torch_mod):
    """positions in a periodic box (Lagrangian grid + displacement, wrapped) and velocities with data-derived limits"""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    nf, sub_cells, L = 24, 3, 250.0
    g = make_vectors((0, 0, 0), L, nf)
    pos = np.mod(g + 2.0 * rng.standard_normal(g.shape), L).astype(np.float32)
    pos[pos >= np.float32(L)] = 0.0
    _roundtrip(mnw, ctx2, oracle, torch, dev, np.ascontiguousarray(pos), nf, sub_cells, 0.001, True, L)
    vel = (300.0 * rng.standard_normal(g.shape)).astype(np.float32)
    _roundtrip(mnw, ctx2, oracle, torch, dev, np.ascontiguousarray(vel), nf, sub_cells, 1.0, False, L)


def _roundtrip(mnw, ctx, oracle, torch, dev, vec, nf, sub_cells, dx, periodic, L):
    from minnow_amd import minp
    n_sub = nf // sub_cells
    nb = 3 * sub_cells ** 3
    slot = (8 * n_sub ** 3 + 16 + 15) // 16 * 16
    t = torch.from_numpy(vec).to(dev)
    packed = torch.zeros(nb * slot, dtype=torch.uint8, device=dev)
    out = torch.zeros_like(t)
    torch.cuda.synchronize()
    metas, low, high, pixels = minp.vectors_encode(ctx, t.data_ptr(), nf, sub_cells, dx, periodic, L, packed.data_ptr(), slot)
    torch.cuda.synchronize()
    blocks, omn, omx, opix = oracle.minp_vectors_encode(vec, nf, sub_cells, dx, periodic, L)
    assert np.array_equal(np.float32(low), omn) and np.array_equal(np.float32(high), omx) and list(pixels) == list(opix)
    host = packed.cpu().numpy()
    for b, (m, (omin, obits, odata)) in enumerate(zip(metas, blocks)):
        assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), b
        assert host[b * slot: b * slot + m.nbytes].tobytes() == odata.tobytes(), b
    minp.vectors_decode(ctx, packed.data_ptr(), slot, metas, nf, sub_cells, low, high, pixels, periodic, L, out.data_ptr(),
                        seed=9, first_block=4)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = oracle.minp_vectors_decode(blocks, nf, sub_cells, omn, omx, opix, periodic, L, seed=9, first_block=4)
    # (nSide = 1 without a box gives limits [0, 0], pixels = 0 and dx = NaN in the reference too: NaN out, and its
    # vectorsEq -- two comparisons that are false for NaN -- still passes; minp_test.go:14, 127-137)
    assert np.array_equal(got, ref, equal_nan=True)
    # vectorsEq of the reference's test (minp_test.go:127-137); periodic coordinates compare on the ring
    d = np.abs(got.astype(np.float64) - vec.astype(np.float64))
    if periodic:
        d = np.minimum(d, L - d)
    viol = np.argwhere(d > float(np.float32(dx)) * 1.01 + 1e-4)        # dx plus float32 rounding at |x| ~ L
    if periodic:
        assert len(viol) == 0
    else:
        # Reference behaviour, reproduced: with data-derived limits the component's maximum sits one ulp under
        # `high` (minp.go:94) and its float32 quotient can round up to `pixels`, which bound() wraps to the
        # bottom of the range (group.go:374-382).  Nothing else may deviate.
        for i, k in viol:
            assert vec[i, k] == vec[:, k].max()


def test_c4_shaped_file_cell(mnw, ctx2, torch_mod):
    """BASELINE configs[3] per-GPU shape at reduced size: one file cell of n_file = 200 (8e6 particles),
    sub_cells = 5 -> 375 blocks of 64,000; positions and velocities; properties only (round trip within dx,
    every block of a component has the group's limits, deterministic re-encode)."""
    from minnow_amd import minp
    torch = torch_mod
    dev = torch.device("cuda:0")
    nf, sub_cells, L, dxp, dxv = 200, 5, 250.0, 0.001, 1.0
    g = torch.Generator(device=dev); g.manual_seed(123)
    idx = torch.arange(nf ** 3, device=dev)
    grid = torch.stack([idx % nf, (idx // nf) % nf, idx // (nf * nf)], dim=1).to(torch.float32) * (L / nf)
    pos = torch.remainder(grid + 2.0 * torch.randn(grid.shape, device=dev, generator=g), L).contiguous()
    pos[pos >= L] = 0.0
    vel = (300.0 * torch.randn(grid.shape, device=dev, generator=g)).contiguous()
    n_sub = nf // sub_cells
    nb = 3 * sub_cells ** 3
    slot = (4 * n_sub ** 3 + 15) // 16 * 16
    packed = torch.zeros(nb * slot, dtype=torch.uint8, device=dev)
    out = torch.empty_like(pos)
    torch.cuda.synchronize()
    for vec, dx, periodic in ((pos, dxp, True), (vel, dxv, False)):
        metas, low, high, pixels = minp.vectors_encode(ctx2, vec.data_ptr(), nf, sub_cells, dx, periodic, L, packed.data_ptr(), slot)
        torch.cuda.synchronize()
        assert all(m.status == 0 for m in metas)
        first = packed.clone()
        minp.vectors_decode(ctx2, packed.data_ptr(), slot, metas, nf, sub_cells, low, high, pixels, periodic, L, out.data_ptr(), seed=3)
        torch.cuda.synchronize()
        d = (out.double() - vec.double()).abs()
        if periodic:
            d = torch.minimum(d, L - d)
        else:
            d[vec == vec.max(dim=0).values] = 0.0          # the component maxima may wrap (see _roundtrip)
        assert float(d.max()) <= dx * 1.01 + 1e-4
        metas2, *_ = minp.vectors_encode(ctx2, vec.data_ptr(), nf, sub_cells, dx, periodic, L, packed.data_ptr(), slot)
        torch.cuda.synchronize()
        assert [(m.min, m.bits, m.nbytes) for m in metas] == [(m.min, m.bits, m.nbytes) for m in metas2]
        assert torch.equal(first, packed)


@pytest.mark.parametrize("fused,single", [(1, 1), (1, 0), (0, 0)])
def test_vectors_fused_and_staged_paths_agree_with_oracle(mnw, oracle, torch_mod, fused, single):
    """n_sub and n_file multiples of four: mnw_vectors_encode reads the grid directly -- once (k_encode_reg_vec3), or twice
    with vec3_single_read = 0 (k_reduce_vec3 + k_pack_vec3) -- unless vec3_fused = 0; all must give the oracle's blocks.
    Includes a partial last vtile (n_sub^3 = 13,824 = 6 * 2,048 + 1,536), a constant component (bits == 0) and a wrapped ring."""
    from minnow_amd import minp
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(99)
    nf, sub_cells, L = 48, 2, 250.0
    g = make_vectors((0, 0, 0), L, nf)
    pos = np.mod(g + 3.0 * rng.standard_normal(g.shape), L).astype(np.float32)
    pos[pos >= np.float32(L)] = 0.0
    pos[:, 1] = np.mod(1.0 + 0.5 * rng.standard_normal(nf ** 3), L).astype(np.float32)      # hugs the seam: periodicMin != 0
    pos[:, 2] = np.float32(17.25)                                                            # constant: every block is 0 bits
    with mnw.Context(0) as c:
        c.set_option("vec3_fused", fused)
        c.set_option("vec3_single_read", single)
        c.timing_enable(True)
        _roundtrip(mnw, c, oracle, torch, dev, np.ascontiguousarray(pos), nf, sub_cells, 0.001, True, L)
        ran = {k for k, v in c.timing_get().items() if v["launches"]}
        assert ("k_pack_vec3" in ran) == bool(fused and not single) and ("k_subcell_gather" in ran) == (not fused)
        assert ("k_encode_reg_vec3" in ran) == bool(fused and single)
        assert c.counters()["vec3_single_read_calls"] == (1 if fused and single else 0) and c.counters()["single_read_aborts"] == 0
        assert ("k_unpack_vec3" in ran) == bool(fused) and ("k_subcell_scatter" in ran) == (not fused)
        # a NaN sends its block to the exact path: the fused kernels hand the whole call to the staging path
        bad = pos.copy(); bad[12345, 0] = np.nan
        c.timing_reset()
        t = torch.from_numpy(bad).to(dev)
        n_sub = nf // sub_cells
        slot = (8 * n_sub ** 3 + 16 + 15) // 16 * 16
        packed = torch.zeros(3 * sub_cells ** 3 * slot, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        metas, low, high, pixels = minp.vectors_encode(c, t.data_ptr(), nf, sub_cells, 0.001, True, L, packed.data_ptr(), slot)
        torch.cuda.synchronize()
        blocks, *_ = oracle.minp_vectors_encode(bad, nf, sub_cells, 0.001, True, L)
        host = packed.cpu().numpy()
        for b, (m, (omin, obits, odata)) in enumerate(zip(metas, blocks)):
            assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), b
            assert host[b * slot: b * slot + m.nbytes].tobytes() == odata.tobytes(), b


@pytest.mark.parametrize("x_order,tiles", [(0, 8), (1, 1), (1, 4), (2, 1), (2, 3), (2, 8), (2, 64)])
@pytest.mark.parametrize("nf,sub_cells", [(48, 2), (60, 3), (80, 5)])
def test_vectors_decode_orderings_agree_with_oracle(mnw, oracle, torch_mod, x_order, tiles, nf, sub_cells):
    """k_unpack_vec3's work distribution (vec3_x_order 0: contiguous ranges; 1: x-neighbours adjacent in dispatch order; 2, the
    default: x-neighbours on one XCD, grid rounded up to groups of 8 * sub_cells workgroups of which the surplus exits at once)
    and pieces of 1 .. more-than-a-sub-cell vtiles: every combination decodes the oracle's floats, sub-cell counts that are not
    multiples of eight and partial last vtiles included (13,824 / 8,000 / 4,096 particles per sub-cell)."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5 + nf)
    L = 250.0
    g = make_vectors((0, 0, 0), L, nf)
    pos = np.mod(g + 3.0 * rng.standard_normal(g.shape), L).astype(np.float32)
    pos[pos >= np.float32(L)] = 0.0
    with mnw.Context(0) as c:
        c.set_option("vec3_x_order", x_order)
        c.set_option("vec3_unpack_tiles", tiles)
        c.timing_enable(True)
        _roundtrip(mnw, c, oracle, torch, dev, np.ascontiguousarray(pos), nf, sub_cells, 0.001, True, L)
        assert c.timing_get()["k_unpack_vec3"]["launches"] >= 1


def test_vectors_host_pointer_forms(mnw, oracle):
    """mnw_vectors_encode_host / _decode_host: numpy arrays standing in for the Go slices of minp.Writer.Vectors"""
    import ctypes as C
    from minnow_amd import _lib
    L_ = _lib.lib()
    rng = np.random.default_rng(55)
    nf, sub_cells, L, dx = 16, 2, 100.0, 0.1
    vec = np.ascontiguousarray(np.mod(make_vectors((0, 0, 0), L, nf) + rng.standard_normal((nf ** 3, 3)), L).astype(np.float32))
    vec[vec >= np.float32(L)] = 0.0
    n_sub = nf // sub_cells
    nb = 3 * sub_cells ** 3
    slot = (4 * n_sub ** 3 + 15) // 16 * 16
    out = np.zeros(nb * slot, dtype=np.uint8)
    blocks, omn, omx, opix = oracle.minp_vectors_encode(vec, nf, sub_cells, dx, True, L)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    i3 = lambda v: (C.c_int64 * 3)(*[int(x) for x in v])
    meta = (_lib.BlockMetaC * nb)()
    with mnw.Context(0) as c:
        rc = L_.mnw_vectors_encode_host(c.handle, vec.ctypes.data, nf, sub_cells, f3(omn), f3(omx), i3(opix), out.ctypes.data, slot, meta)
        assert rc == 0
        for b, (omin, obits, odata) in enumerate(blocks):
            assert (meta[b].min, meta[b].bits, meta[b].nbytes, meta[b].status) == (omin, obits, odata.size, 0)
            assert out[b * slot: b * slot + odata.size].tobytes() == odata.tobytes()
        back = np.zeros_like(vec)
        rc = L_.mnw_vectors_decode_host(c.handle, out.ctypes.data, slot, meta, nf, sub_cells, f3(omn), f3(omx), i3(opix), 1, L, 9, 4,
                                        back.ctypes.data)
        assert rc == 0
    ref = oracle.minp_vectors_decode(blocks, nf, sub_cells, omn, omx, opix, True, L, seed=9, first_block=4)
    assert np.array_equal(back, ref)


def test_c4_full_size_file_cell_vs_oracle(mnw, oracle, torch_mod):
    """configs[3], one GPU's share at FULL size: one minp file cell of 500^3 = 1.25e8 particles, sub_cells = 5 -> 375 blocks of
    1e6 per component (go/minp/minp.go:98-118), positions in a periodic box.  EVERY block against the oracle's Vectors (limits,
    pixels, (min, bits, nbytes), stream bytes) and every decoded particle against the oracle's reader with the same dither."""
    from minnow_amd import minp
    torch = torch_mod
    dev = torch.device("cuda:0")
    nf, S, L, dx = 500, 5, 250.0, 0.001
    n = nf ** 3
    g = torch.Generator(device=dev); g.manual_seed(0xC4)
    idx = torch.arange(n, device=dev)
    pos = torch.stack([idx % nf, (idx // nf) % nf, idx // (nf * nf)], dim=1).to(torch.float32) * (L / nf)
    del idx
    pos.add_(torch.randn(pos.shape, device=dev, generator=g).mul_(2.0))
    pos = torch.remainder(pos, L).contiguous()
    pos[pos >= L] = 0.0
    n_sub = nf // S
    nb = 3 * S ** 3
    slot = (4 * n_sub ** 3 + 15) // 16 * 16
    packed = torch.zeros(nb * slot, dtype=torch.uint8, device=dev)
    out = torch.empty_like(pos)
    torch.cuda.synchronize()
    with mnw.Context(0) as c:
        c.timing_enable(True)
        metas, low, high, pixels = minp.vectors_encode(c, pos.data_ptr(), nf, S, dx, True, L, packed.data_ptr(), slot)
        minp.vectors_decode(c, packed.data_ptr(), slot, metas, nf, S, low, high, pixels, True, L, out.data_ptr(), seed=5, first_block=2)
        torch.cuda.synchronize()
        ran = {k for k, v in c.timing_get().items() if v["launches"]}
        # default (round 3): the grid is read once; nothing gave up, so neither of the two-launch kernels ran
        assert {"k_encode_reg_vec3", "k_unpack_vec3"} <= ran and not ({"k_reduce_vec3", "k_pack_vec3"} & ran), ran
        assert c.counters()["single_read_aborts"] == 0 and c.counters()["vec3_single_read_calls"] == 1
        # the two-launch form (the fallback of a give-up) gives the same file
        first = packed.clone()
        c.set_option("vec3_single_read", 0)
        packed.zero_()
        torch.cuda.synchronize()
        metas2, *_ = minp.vectors_encode(c, pos.data_ptr(), nf, S, dx, True, L, packed.data_ptr(), slot)
        torch.cuda.synchronize()
        assert c.timing_get()["k_encode_reg_vec3"]["launches"] == 1 and c.timing_get()["k_pack_vec3"]["launches"] == 1
        assert [(m.min, m.bits, m.nbytes, m.status) for m in metas2] == [(m.min, m.bits, m.nbytes, m.status) for m in metas]
        assert torch.equal(first, packed)
        del first
    vec = pos.cpu().numpy()
    blocks, omn, omx, opix = oracle.minp_vectors_encode(vec, nf, S, dx, True, L)
    assert np.array_equal(np.float32(low), omn) and np.array_equal(np.float32(high), omx) and list(pixels) == list(opix)
    host = packed.cpu().numpy()
    for b, (m, (omin, obits, odata)) in enumerate(zip(metas, blocks)):
        assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), b
        assert host[b * slot: b * slot + m.nbytes].tobytes() == odata.tobytes(), b
    ref = oracle.minp_vectors_decode(blocks, nf, S, omn, omx, opix, True, L, seed=5, first_block=2)
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    d = np.abs(got.astype(np.float64) - vec.astype(np.float64))
    d = np.minimum(d, L - d)
    assert d.max() <= dx * 1.01 + 1e-4


def test_c4_full_size_velocity_cell_vs_oracle(mnw, oracle, torch_mod):
    """configs[3], velocities at FULL size: one minp file cell of 500^3 particles, v = N(0, 300) per component, dv = 1, NOT
    periodic -- the limits come from bounds() with Nextafter32 on the maxima (go/minp/minp.go:87-96, 290-300;
    go/minp/snapshot/minp.go:111-122).  EVERY block against the oracle's Vectors (limits, pixels, (min, bits, nbytes), stream
    bytes) and every decoded particle against the oracle's reader with the same dither; the grid is read once."""
    from minnow_amd import minp
    torch = torch_mod
    dev = torch.device("cuda:0")
    nf, S, L, dv = 500, 5, 250.0, 1.0
    n = nf ** 3
    g = torch.Generator(device=dev); g.manual_seed(0xC4 + 1)
    vel = torch.randn((n, 3), device=dev, generator=g, dtype=torch.float32).mul_(300.0).contiguous()
    n_sub = nf // S
    nb = 3 * S ** 3
    slot = (4 * n_sub ** 3 + 15) // 16 * 16
    packed = torch.zeros(nb * slot, dtype=torch.uint8, device=dev)
    out = torch.empty_like(vel)
    torch.cuda.synchronize()
    with mnw.Context(0) as c:
        c.timing_enable(True)
        metas, low, high, pixels = minp.vectors_encode(c, vel.data_ptr(), nf, S, dv, False, L, packed.data_ptr(), slot)
        minp.vectors_decode(c, packed.data_ptr(), slot, metas, nf, S, low, high, pixels, False, L, out.data_ptr(), seed=6, first_block=3)
        torch.cuda.synchronize()
        ran = {k for k, v in c.timing_get().items() if v["launches"]}
        assert {"k_encode_reg_vec3", "k_unpack_vec3"} <= ran and not ({"k_reduce_vec3", "k_pack_vec3", "k_subcell_gather"} & ran), ran
        assert c.counters()["single_read_aborts"] == 0
    vec = vel.cpu().numpy()
    del vel
    blocks, omn, omx, opix = oracle.minp_vectors_encode(vec, nf, S, dv, False, L)
    assert np.array_equal(np.float32(low), omn) and np.array_equal(np.float32(high), omx) and list(pixels) == list(opix)
    host = packed.cpu().numpy()
    for b, (m, (omin, obits, odata)) in enumerate(zip(metas, blocks)):
        assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), b
        assert host[b * slot: b * slot + m.nbytes].tobytes() == odata.tobytes(), b
    ref = oracle.minp_vectors_decode(blocks, nf, S, omn, omx, opix, False, L, seed=6, first_block=3)
    got = out.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # the reference's tolerance -- except where its own quirk strikes: a component's maximum can quantize to `pixels` (the limits
    # come from the data: high = Nextafter32(max)) and is then read back at the BOTTOM of the range (bound() wraps it); the build
    # reproduces that bit for bit (asserted above against the oracle), so those elements, and only those, are far off
    far = np.abs(got.astype(np.float64) - vec.astype(np.float64)) > dv * 1.01 + 1e-3
    assert far.sum() <= 3
    for i, k in zip(*np.nonzero(far)):
        assert vec[i, k] == vec[:, k].max() and got[i, k] < np.float32(low[k]) + np.float32(dv)


@pytest.mark.parametrize("nf,sub_cells", [(96, 1), (120, 2), (160, 5)])
def test_vectors_single_read_shapes_and_give_up(mnw, oracle, torch_mod, nf, sub_cells):
    """k_encode_reg_vec3 on geometries with several record groups per block (n_sub^3 / 2,048 = 432, 105.5, 16 vtiles: 14, 4 and
    1 groups of 32; a short last group; a partial last vtile) against the oracle, positions hugging the box seam in one
    component; then the give-up path: a vtile's records are never published (test hook), the launch must end after its
    bounded wait and the call must be redone by k_reduce_vec3 + k_pack_vec3 -- same blocks."""
    from minnow_amd import minp
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 + nf)
    L, dx = 250.0, 0.001
    g = make_vectors((0, 0, 0), L, nf)
    pos = np.mod(g + 3.0 * rng.standard_normal(g.shape), L).astype(np.float32)
    pos[:, 1] = np.mod(0.5 + 0.4 * rng.standard_normal(nf ** 3), L).astype(np.float32)
    pos[pos >= np.float32(L)] = 0.0
    pos = np.ascontiguousarray(pos)
    n_sub = nf // sub_cells
    nb = 3 * sub_cells ** 3
    slot = (4 * n_sub ** 3 + 15) // 16 * 16
    blocks, omn, omx, opix = oracle.minp_vectors_encode(pos, nf, sub_cells, dx, True, L)
    t = torch.from_numpy(pos).to(dev)
    packed = torch.zeros(nb * slot, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def check(metas):
        host = packed.cpu().numpy()
        for b, (m, (omin, obits, odata)) in enumerate(zip(metas, blocks)):
            assert (m.min, m.bits, m.nbytes, m.status) == (omin, obits, odata.size, 0), b
            assert host[b * slot: b * slot + m.nbytes].tobytes() == odata.tobytes(), b
            assert not host[b * slot + m.nbytes: (b + 1) * slot].any()

    with mnw.Context(0) as c:
        c.set_option("vec3_single_read", 1)
        c.set_option("single_read_backoff", 0)    # (every call is to try the single-read kernel)
        c.timing_enable(True)
        for rep in range(3):                      # (generations: the records of an earlier call must not satisfy a later one)
            packed.zero_()
            torch.cuda.synchronize()
            metas, low, high, pixels = minp.vectors_encode(c, t.data_ptr(), nf, sub_cells, dx, True, L, packed.data_ptr(), slot)
            torch.cuda.synchronize()
            check(metas)
        kt = c.timing_get()
        assert kt["k_encode_reg_vec3"]["launches"] == 3 and kt["k_pack_vec3"]["launches"] == 0
        assert c.counters()["vec3_single_read_calls"] == 3 and c.counters()["single_read_aborts"] == 0
        c.set_option("reg_test_drop_item", (n_sub ** 3 + 2047) // 2048 // 2)        # a vtile in the middle of the first sub-cell
        packed.zero_()
        torch.cuda.synchronize()
        metas, *_ = minp.vectors_encode(c, t.data_ptr(), nf, sub_cells, dx, True, L, packed.data_ptr(), slot)
        torch.cuda.synchronize()
        check(metas)
        assert c.counters()["single_read_aborts"] == 1 and c.timing_get()["k_pack_vec3"]["launches"] == 1
        c.set_option("reg_test_drop_item", -1)
        packed.zero_()
        torch.cuda.synchronize()
        metas, *_ = minp.vectors_encode(c, t.data_ptr(), nf, sub_cells, dx, True, L, packed.data_ptr(), slot)
        torch.cuda.synchronize()
        check(metas)
        assert c.counters()["single_read_aborts"] == 1 and c.timing_get()["k_encode_reg_vec3"]["launches"] == 5


def test_fuzz_short(mnw, oracle):
    """a few seconds of tests/fuzz_minp.py: random file-cell geometries, value distributions, limits and library options (every depth
    and hand-out table of the single-read kernel, the two-launch form, the staging path, every decode ordering) against the oracle;
    the long runs are done by hand (`python tests/fuzz_minp.py 600 1`: profiles/r03/fuzz_minp.log)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import fuzz_minp
    cases, blocks, single, staged = fuzz_minp.run(8.0, 2026)
    assert cases >= 3 and single >= 1
