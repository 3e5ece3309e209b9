"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI's
host-pointer entry points (numpy arrays in, numpy arrays out -- what the cgo shim does with Go
slices).  Names and error behaviour follow the reference:

  BitArray          go/bit/bit.go:13-17, NewArray 138-142, Slice 29-82
  IntGroupCodec     intGroup.writeData / readData, go/group.go:242-263
  FloatGroupCodec   floatGroup.writeData / readData, go/group.go:299-327
                    (+ Writer.FloatGroup's parameter derivation, go/writer.go:72-75)

Where the reference panics these raise MnwError.  File I/O, mins/bits bookkeeping and
blockIndex.addBlock stay with the caller, as they stay in Go behind the cgo shim.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MnwError


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class BitArray:
    """bit.Array{Length, Bits, Data} (go/bit/bit.go:13-17)."""

    def __init__(self, length, bits, data):
        self.Length, self.Bits, self.Data = int(length), int(bits), data

    @classmethod
    def new(cls, ctx, bits, x):
        """bit.NewArray(bits, x), bit.go:138-142"""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        if bits > 64:
            raise MnwError(-1, "Cannot pack more than 64 bits per element into a bit.Array")
        data = np.zeros(_lib.array_bytes(bits, x.size), dtype=np.uint8)
        ctx._check(ctx._L.mnw_pack_u64(ctx.handle, _ptr(x), x.size, int(bits), _ptr(data), data.size))
        return cls(x.size, bits, data)

    @classmethod
    def buffered(cls, ctx, bits, x, b):
        """bit.BufferedArray(bits, x, b), bit.go:84-134: len(b) must equal ArrayBytes."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        ctx._check(ctx._L.mnw_pack_u64(ctx.handle, _ptr(x), x.size, int(bits), _ptr(b), b.size))
        return cls(x.size, bits, b)

    def slice(self, ctx, out=None):
        """(*Array).Slice(out), bit.go:29-82: len(out) must be >= Length."""
        if out is None:
            out = np.empty(self.Length, dtype=np.uint64)
        if out.size < self.Length:
            raise MnwError(-2, "Array has length %d, but out buffer has length %d." % (self.Length, out.size))
        data = np.ascontiguousarray(self.Data, dtype=np.uint8)
        ctx._check(ctx._L.mnw_unpack_u64(ctx.handle, _ptr(data), data.size, self.Bits, self.Length, _ptr(out)))
        return out


class IntGroupCodec:
    """intGroup (go/group.go:172-263) without the file: keeps mins/bits like the Go struct."""

    def __init__(self, ctx, N):
        self.ctx, self.N = ctx, int(N)
        self.mins, self.bits, self.sizes = [], [], []

    def write_data(self, x):
        """intGroup.writeData (group.go:242-255) -> the bytes f.Write would receive."""
        x = np.ascontiguousarray(x, dtype=np.int64)
        out = np.zeros(8 * x.size + 8, dtype=np.uint8)
        mn, bits, olen = C.c_int64(), C.c_int64(), C.c_int64()
        c = self.ctx
        c._check(c._L.mnw_int_encode(c.handle, _ptr(x), x.size, C.byref(mn), C.byref(bits), _ptr(out),
                                     out.size, C.byref(olen)))
        self.mins.append(mn.value)
        self.bits.append(bits.value)
        self.sizes.append(_lib.array_bytes(bits.value, self.N))      # addBlock, group.go:254
        return out[: olen.value].copy()

    def read_data(self, data, b, n=None):
        """intGroup.readData (group.go:257-263) on the bytes f.Read produced."""
        n = self.N if n is None else n
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.empty(n, dtype=np.int64)
        c = self.ctx
        c._check(c._L.mnw_int_decode(c.handle, _ptr(data), data.size, int(self.mins[b]), int(self.bits[b]),
                                     n, _ptr(out)))
        return out


class FloatGroupCodec:
    """floatGroup (go/group.go:269-344) without the file."""

    def __init__(self, ctx, N, low, high, pixels, periodic=True):
        self.ig = IntGroupCodec(ctx, N)
        self.ctx = ctx
        self.low, self.high = float(np.float32(low)), float(np.float32(high))
        self.pixels, self.periodic = int(pixels), 1 if periodic else 0

    @classmethod
    def from_limits(cls, ctx, N, lim, dx):
        """Writer.FloatGroup(N, lim, dx), writer.go:72-75 (periodic is always true there)."""
        pixels = _lib.float_group_pixels(np.float32(lim[0]), np.float32(lim[1]), np.float32(dx))
        return cls(ctx, N, lim[0], lim[1], pixels, True)

    def write_data(self, x):
        """floatGroup.writeData (group.go:312-327)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(8 * x.size + 8, dtype=np.uint8)
        mn, bits, olen = C.c_int64(), C.c_int64(), C.c_int64()
        c = self.ctx
        c._check(c._L.mnw_float_encode(c.handle, _ptr(x), x.size, self.low, self.high, self.pixels,
                                       self.periodic, C.byref(mn), C.byref(bits), _ptr(out), out.size,
                                       C.byref(olen)))
        self.ig.mins.append(mn.value)
        self.ig.bits.append(bits.value)
        self.ig.sizes.append(_lib.array_bytes(bits.value, self.ig.N))
        return out[: olen.value].copy()

    def read_data(self, data, b, n=None, seed=0):
        """floatGroup.readData (group.go:299-310); dither keyed by (seed, b)."""
        n = self.ig.N if n is None else n
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.empty(n, dtype=np.float32)
        c = self.ctx
        c._check(c._L.mnw_float_decode(c.handle, _ptr(data), data.size, int(self.ig.mins[b]),
                                       int(self.ig.bits[b]), self.low, self.high, self.pixels,
                                       self.periodic, int(seed), int(b), n, _ptr(out)))
        return out
