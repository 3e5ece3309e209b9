"""Context + batched block jobs over the C ABI (device pointers in, device pointers out).

torch is used only as plumbing by callers (device memory, streams); this module takes raw device
pointers (``tensor.data_ptr()``) and stream handles (``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes as C
from dataclasses import dataclass

from . import _lib
from ._lib import BlockMetaC, DecJobC, EncJobC, KernelTimeC, MinhColC, CutColC, MnwError

INT_GROUP = 10     # go/group.go:22
FLOAT_GROUP = 11   # go/group.go:23
XF_LOG10 = 1       # minh Column.Log (go/minh/minh.go:141, 316)
XF_CLAMP = 2       # minh's clamp into [Low, High) (go/minh/minh.go:142-147)


_META_DTYPE = [("min", "<i8"), ("bits", "<i8"), ("nbytes", "<i8"), ("status", "<i8")]
_DEC_DTYPE = [("src", "<u8"), ("dst", "<u8"), ("n", "<i8"), ("src_len", "<i8"), ("min", "<i8"), ("bits", "<i8"),
              ("pixels", "<i8"), ("low", "<f4"), ("high", "<f4"), ("kind", "<i4"), ("periodic", "<i4"),
              ("seed", "<u8"), ("block", "<u8"), ("xform", "<i4"), ("reserved", "<i4")]


@dataclass
class EncJob:
    """One block to encode: what Writer.Data hands to group.writeData (writer.go:90-103)."""
    src: int               # device pointer: float32[n] or int64[n]
    dst: int               # device pointer, 16-byte aligned
    n: int
    cap: int
    kind: int = FLOAT_GROUP
    low: float = 0.0
    high: float = 0.0
    pixels: int = 0
    periodic: int = 1      # Writer.FloatGroup hard-codes periodic = true (writer.go:74)
    xform: int = 0         # XF_LOG10 | XF_CLAMP: minh.processFloatGroup on the way in (go/minh/minh.go:139-149)


@dataclass
class DecJob:
    """One block to decode: what Reader.Data hands to group.readData (reader.go:114-127)."""
    src: int               # device pointer to the packed stream (any byte alignment)
    dst: int               # device pointer: float32[n] or int64[n]
    n: int
    src_len: int
    min: int
    bits: int
    kind: int = FLOAT_GROUP
    low: float = 0.0
    high: float = 0.0
    pixels: int = 0
    periodic: int = 1
    seed: int = 0
    block: int = 0
    xform: int = 0         # XF_LOG10: 10^x on the way out (go/minh/minh.go:315-319)


@dataclass
class BlockMeta:
    min: int       # appended to intGroup.mins (group.go:251)
    bits: int      # appended to intGroup.bits (group.go:252)
    nbytes: int    # argument of blockIndex.addBlock (group.go:254); 0 when bits == 0
    status: int


class Context:
    """One per Writer/Reader; not thread-safe, like the reference's Writer (writer.go:10-22)."""

    def __init__(self, device=0):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.mnw_ctx_create(int(device), C.byref(h))
        if rc:
            raise MnwError(rc, self._L.mnw_strerror(rc).decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.mnw_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- helpers ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc:
            raise MnwError(rc, self._L.mnw_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def set_option(self, name, value):
        self._check(self._L.mnw_set_option(self._h, name.encode(), int(value)))

    # -- batched device-pointer API -----------------------------------------------------------
    @staticmethod
    def _enc_array(jobs):
        """EncJob list -> C array.  A caller that encodes the same batch repeatedly can build the array
        once with `enc_batch()` and pass that instead of the list."""
        if isinstance(jobs, C.Array):
            return jobs
        arr = (EncJobC * max(len(jobs), 1))()
        for i, j in enumerate(jobs):
            arr[i] = EncJobC(j.src, j.dst, j.n, j.cap, j.pixels, j.low, j.high, j.kind, j.periodic, getattr(j, 'xform', 0), 0)
        arr._n = len(jobs)
        return arr

    enc_batch = _enc_array

    @staticmethod
    def dec_batch(jobs):
        if isinstance(jobs, C.Array):
            return jobs
        arr = (DecJobC * max(len(jobs), 1))()
        for i, j in enumerate(jobs):
            arr[i] = DecJobC(j.src, j.dst, j.n, j.src_len, j.min, j.bits, j.pixels, j.low, j.high,
                             j.kind, j.periodic, j.seed, j.block, getattr(j, 'xform', 0), 0)
        arr._n = len(jobs)
        return arr

    @staticmethod
    def _count(jobs):
        return jobs._n if isinstance(jobs, C.Array) else len(jobs)

    def encode_blocks(self, jobs, stream=None, check=True, raw=False, meta_out=None):
        """raw=True returns the C array of mnw_block_meta itself (see `metas_of`, `dec_batch_update`): a
        caller that runs many small blocks per call avoids one Python object per block that way.  meta_out: a raw
        array returned by an earlier call, filled again instead of allocating a new one."""
        arr = self._enc_array(jobs)
        n = self._count(arr)
        meta = meta_out if meta_out is not None else (BlockMetaC * max(n, 1))()
        meta._n = n
        rc = self._L.mnw_encode_blocks(self._h, stream, arr, n, meta)
        if check:
            self._check(rc)
        return meta if raw else self.metas_of(meta)

    @staticmethod
    def metas_of(meta):
        return [BlockMeta(m.min, m.bits, m.nbytes, m.status) for m in meta[:meta._n]]

    @staticmethod
    def dec_batch_update(arr, meta):
        """Copy (min, bits, nbytes) of a raw meta array into the matching jobs of a prebuilt decode batch:
        what a reader does with intGroup.mins/bits and the block index (group.go:257-263), vectorised."""
        import numpy as np
        n = arr._n
        d = getattr(arr, "_np", None)
        if d is None:
            d = arr._np = np.frombuffer(arr, dtype=_DEC_DTYPE, count=n)
        m = getattr(meta, "_np", None)
        if m is None:
            m = meta._np = np.frombuffer(meta, dtype=_META_DTYPE, count=n)
        d["min"] = m["min"]
        d["bits"] = m["bits"]
        d["src_len"] = m["nbytes"]

    def encode_blocks_async(self, jobs, stream=None):
        arr = self._enc_array(jobs)
        n = self._count(arr)
        self._check(self._L.mnw_encode_blocks_async(self._h, stream, arr, n))
        self._pending_n = n

    def encode_blocks_finish(self, stream=None, check=True):
        n = self._pending_n
        meta = (BlockMetaC * max(n, 1))()
        rc = self._L.mnw_encode_blocks_finish(self._h, stream, meta)
        if check:
            self._check(rc)
        return [BlockMeta(m.min, m.bits, m.nbytes, m.status) for m in meta[:n]]

    def decode_blocks(self, jobs, stream=None):
        arr = self.dec_batch(jobs)
        self._check(self._L.mnw_decode_blocks(self._h, stream, arr, self._count(arr)))

    # -- minp vector blocks (go/minp/minp.go) ----------------------------------------------------
    @staticmethod
    def _f3(v):
        return (C.c_float * 3)(*[float(x) for x in v])

    @staticmethod
    def _i3(v):
        return (C.c_int64 * 3)(*[int(x) for x in v])

    def vectors_bounds(self, vec_ptr, n, stream=None):
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        self._check(self._L.mnw_vectors_bounds(self._h, stream, vec_ptr, n, mn, mx))
        return list(mn), list(mx)

    def vectors_encode(self, vec_ptr, n_file, sub_cells, low, high, pixels, dst_ptr, slot_bytes, stream=None, raw=False):
        nb = 3 * sub_cells ** 3
        meta = (BlockMetaC * nb)()
        meta._n = nb
        self._check(self._L.mnw_vectors_encode(self._h, stream, vec_ptr, n_file, sub_cells, self._f3(low), self._f3(high),
                                               self._i3(pixels), dst_ptr, slot_bytes, meta))
        return meta if raw else self.metas_of(meta)

    def vectors_decode(self, src_ptr, slot_bytes, metas, n_file, sub_cells, low, high, pixels, periodic, L, seed, first_block,
                       vec_out_ptr, stream=None):
        if isinstance(metas, C.Array):
            meta = metas
        else:
            meta = (BlockMetaC * len(metas))(*[BlockMetaC(m.min, m.bits, m.nbytes, m.status) for m in metas])
        self._check(self._L.mnw_vectors_decode(self._h, stream, src_ptr, slot_bytes, meta, n_file, sub_cells, self._f3(low),
                                               self._f3(high), self._i3(pixels), int(bool(periodic)), float(L), seed,
                                               first_block, vec_out_ptr))

    def subcell_gather(self, vec_ptr, n_file, sub_cells, soa_ptr, stream=None):
        self._check(self._L.mnw_subcell_gather(self._h, stream, vec_ptr, n_file, sub_cells, soa_ptr))

    def subcell_scatter(self, soa_ptr, n_file, sub_cells, periodic, L, vec_ptr, stream=None):
        self._check(self._L.mnw_subcell_scatter(self._h, stream, soa_ptr, n_file, sub_cells, int(bool(periodic)), float(L), vec_ptr))

    # -- minh column transforms (go/minh/minh.go:139-149, 315-319) -----------------------------
    def minh_process(self, cols, stream=None):
        """cols: (src_ptr, dst_ptr, n, low, high, log) per column: copy + processFloatGroup on the device"""
        arr = (MinhColC * max(len(cols), 1))(*[MinhColC(s, d, n, lo, hi, int(bool(lg)), 0) for s, d, n, lo, hi, lg in cols])
        self._check(self._L.mnw_minh_process(self._h, stream, arr, len(cols)))

    def minh_unlog(self, cols, stream=None):
        """cols: (ptr, n) per logarithmic column: x <- float32(10 ** float64(x)) in place"""
        arr = (MinhColC * max(len(cols), 1))(*[MinhColC(None, d, n, 0.0, 0.0, 1, 0) for d, n in cols])
        self._check(self._L.mnw_minh_unlog(self._h, stream, arr, len(cols)))

    # -- a reader's f.Read per block (go/reader.go:121-126), for many blocks at once ---------------------
    def copy_file_to_device(self, dev_ptr, fd, file_offsets, nbytes, dst_offsets=None):
        """Byte ranges of the open file `fd` into the device buffer at dev_ptr: packed back to back in the order given, or range i at
        dev_ptr + dst_offsets[i]."""
        n = len(file_offsets)
        A = lambda v: (C.c_int64 * max(n, 1))(*[int(x) for x in v])
        self._check(self._L.mnw_copy_file_to_device(self._h, C.c_void_p(int(dev_ptr)), int(fd), n, A(file_offsets), A(nbytes),
                                                    A(dst_offsets) if dst_offsets is not None else None))

    # -- text catalogue ingest (go/text) ------------------------------------------------------------
    def text_parse(self, data, sep, comment, float_cols, float_out_ptrs, int_cols, int_out_ptrs, capacity_rows):
        """data: bytes-like host text; *_out_ptrs: device pointers of capacity_rows elements each -> rows kept"""
        buf = data if isinstance(data, bytes) else bytes(data)          # passed as a pointer to the bytes object's buffer: no copy
        fc = (C.c_int32 * max(len(float_cols), 1))(*float_cols)
        ic = (C.c_int32 * max(len(int_cols), 1))(*int_cols)
        fo = (C.c_void_p * max(len(float_out_ptrs), 1))(*float_out_ptrs)
        io = (C.c_void_p * max(len(int_out_ptrs), 1))(*int_out_ptrs)
        n = C.c_int64(0)
        self._check(self._L.mnw_text_parse(self._h, buf, len(data), sep, comment, fc, len(float_cols), fo, ic, len(int_cols), io,
                                           capacity_rows, C.byref(n)))
        return n.value

    def text_rows(self, data, sep, comment):
        """(lines, lines kept) of a block of host text; the text_parse that directly follows on the same bytes object reuses the upload"""
        nl, nr = C.c_int64(0), C.c_int64(0)
        self._check(self._L.mnw_text_rows(self._h, data, len(data), sep, comment, C.byref(nl), C.byref(nr)))
        return nl.value, nr.value

    def generic_cut(self, mass_ptr, cutoff, n, cols, stream=None):
        """GenericCut (scripts/text_to_minh.go:271-312): cols = [(src_ptr, dst_ptr, elem_bytes)] device columns -> rows kept"""
        arr = (CutColC * max(len(cols), 1))(*[CutColC(s, d, e, 0) for s, d, e in cols])
        kept = C.c_int64(0)
        self._check(self._L.mnw_generic_cut(self._h, stream, mass_ptr, cutoff, n, arr, len(cols), C.byref(kept)))
        return kept.value

    def text_stats(self):
        """of the last text_parse: (lines, lines kept, tiles on the general path)"""
        out = (C.c_int64 * 3)()
        self._check(self._L.mnw_text_stats(self._h, out))
        return tuple(out)

    def pack_u64_dev(self, x_ptr, n, bits, out_ptr, cap, stream=None):
        self._check(self._L.mnw_pack_u64_dev(self._h, stream, x_ptr, n, bits, out_ptr, cap))

    def unpack_u64_dev(self, in_ptr, in_len, bits, n, out_ptr, stream=None):
        self._check(self._L.mnw_unpack_u64_dev(self._h, stream, in_ptr, in_len, bits, n, out_ptr))

    # -- counters -----------------------------------------------------------------------------
    def counters(self):
        """which encode structure the float batches took and how often the single-read kernel gave up (mnw_counters_get)"""
        out = (C.c_int64 * _lib.C_COUNT)()
        self._check(self._L.mnw_counters_get(self._h, out))
        return {name: int(out[k]) for k, name in enumerate(_lib.COUNTER_NAMES)}

    def counters_reset(self):
        self._check(self._L.mnw_counters_reset(self._h))

    # -- timing ------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._check(self._L.mnw_timing_enable(self._h, int(on)))

    def timing_reset(self):
        self._check(self._L.mnw_timing_reset(self._h))

    def timing_get(self):
        arr = (KernelTimeC * _lib.K_COUNT)()
        self._check(self._L.mnw_timing_get(self._h, arr))
        out = {}
        for k in range(_lib.K_COUNT):
            name = self._L.mnw_kernel_name(k).decode()
            if name:
                out[name] = {"total_ms": arr[k].total_ms, "launches": arr[k].launches}
        return out
