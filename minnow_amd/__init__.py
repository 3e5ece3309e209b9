"""minnow_amd -- MI355X-native quantize + bit-pack hot path of minnow's IntGroup / FloatGroup.

The product is the C-ABI library libminnow_hip.so (include/minnow_hip.h, minnow_amd/csrc); this
package is the thin host-side mirror of the reference's interface for that path (group.go's
intGroup/floatGroup writeData/readData and go/bit's Array), written in Python because the Go
toolchain is absent from the build image.  There is no CPU fallback: without the HIP library or a
GPU every compute call raises.
"""
from ._lib import (  # noqa: F401
    MnwError,
    array_bytes,
    device_count,
    float_group_pixels,
    lib,
    lib_path,
    precision_needed,
)
from .codec import (  # noqa: F401
    FLOAT_GROUP,
    XF_CLAMP,
    XF_LOG10,
    INT_GROUP,
    BlockMeta,
    Context,
    DecJob,
    EncJob,
)
from .groups import BitArray, FloatGroupCodec, IntGroupCodec  # noqa: F401

__all__ = [
    "MnwError", "array_bytes", "device_count", "float_group_pixels", "lib", "lib_path",
    "precision_needed", "Context", "EncJob", "DecJob", "BlockMeta", "INT_GROUP", "FLOAT_GROUP", "XF_CLAMP", "XF_LOG10",
    "BitArray", "IntGroupCodec", "FloatGroupCodec",
]
