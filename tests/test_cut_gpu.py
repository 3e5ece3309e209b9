"""GenericCut of the reference's text_to_minh (scripts/text_to_minh.go:271-312) on device columns against the oracle's
restatement: ragged lengths around the 2,048-row tiles, all / none / alternating rows kept, NaN masses, unaligned columns,
more columns than one launch carries."""
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


def _run(mnw, torch, ctx, oracle, mass, cutoff, cols, offset=0):
    dev = torch.device("cuda:0")
    n = mass.size
    d_src, d_dst, spec = [], [], []
    for x in cols:
        t = torch.zeros(n + offset + 8, dtype=torch.float32 if x.dtype == np.float32 else torch.int64, device=dev)
        t[offset:offset + n] = torch.from_numpy(x).to(dev)
        o = torch.full((n + offset + 8,), -3, dtype=t.dtype, device=dev)
        d_src.append(t); d_dst.append(o)
        spec.append((t.data_ptr() + offset * x.itemsize, o.data_ptr() + offset * x.itemsize, x.itemsize))
    mass_idx = next(i for i, x in enumerate(cols) if x is mass)
    torch.cuda.synchronize()
    kept = ctx.generic_cut(spec[mass_idx][0], cutoff, n, spec)
    torch.cuda.synchronize()
    rk, rcols = oracle.generic_cut(cutoff, mass, cols)
    assert kept == rk
    for o, r in zip(d_dst, rcols):
        got = o[offset:offset + kept].cpu().numpy()
        assert np.array_equal(got.view(np.uint8), r.view(np.uint8))
        assert (o[offset + kept:offset + kept + 4].cpu().numpy() == -3).all()          # nothing written behind the kept rows
    return kept


def test_generic_cut(mnw, oracle):
    import torch
    rng = np.random.default_rng(9)
    with mnw.Context(0) as c:
        for n in [1, 7, 8, 9, 2047, 2048, 2049, 4096, 100003, 1 << 20]:
            for mode in ("random", "all", "none", "alternate"):
                mass = (10 ** rng.uniform(8, 14, n)).astype(np.float32)
                if mode == "all":
                    cutoff = 0.0
                elif mode == "none":
                    cutoff = 1e30
                elif mode == "alternate":
                    mass[::2] = 1.0
                    cutoff = 2.0
                else:
                    cutoff = 1e11
                    mass[rng.integers(0, n, max(n // 50, 1))] = np.nan
                cols = [rng.integers(-2**62, 2**62, n).astype(np.int64), mass, rng.normal(0, 1, n).astype(np.float32),
                        np.arange(n, dtype=np.int64)]
                _run(mnw, torch, c, oracle, mass, cutoff, cols, offset=0 if n % 2 == 0 else 1)
        # more columns than one launch carries (64)
        n = 5000
        mass = rng.uniform(0, 1, n).astype(np.float32)
        cols = [mass] + [rng.normal(0, 1, n).astype(np.float32) if k % 3 else rng.integers(0, 10**9, n).astype(np.int64) for k in range(99)]
        assert _run(mnw, torch, c, oracle, mass, 0.5, cols) == int((mass > 0.5).sum())
        # argument errors: in place, odd element size
        t = torch.zeros(16, dtype=torch.float32, device="cuda:0")
        with pytest.raises(mnw.MnwError):
            c.generic_cut(t.data_ptr(), 0.5, 16, [(t.data_ptr(), t.data_ptr(), 4)])
        with pytest.raises(mnw.MnwError):
            c.generic_cut(t.data_ptr(), 0.5, 16, [(t.data_ptr(), t.data_ptr() + 64, 2)])
