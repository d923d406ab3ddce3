"""Container shared by make_ref_inputs.py, dump_ref_golden.cpp and collect_ref_golden.py: a flat sequence of named arrays.
record = u32 name length | name | u8 dtype (0 u8, 1 i32, 2 f32, 3 f64, 4 u32) | u8 ndim | u32 dims[ndim] | raw little-endian data"""
import struct

import numpy as np

DT = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.float64, 4: np.uint32}
CODE = {np.dtype(v): k for k, v in DT.items()}


def write(path, arrays):
    with open(path, "wb") as fh:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            nb = name.encode()
            fh.write(struct.pack("<I", len(nb)) + nb + struct.pack("<BB", CODE[a.dtype], a.ndim) + struct.pack("<%dI" % a.ndim, *a.shape))
            fh.write(a.tobytes())


def read(path):
    out, buf, o = {}, open(path, "rb").read(), 0
    while o < len(buf):
        n, = struct.unpack_from("<I", buf, o); o += 4
        name = buf[o:o + n].decode(); o += n
        code, nd = struct.unpack_from("<BB", buf, o); o += 2
        dims = struct.unpack_from("<%dI" % nd, buf, o); o += 4 * nd
        dt = np.dtype(DT[code])
        cnt = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(buf, dtype=dt, count=cnt, offset=o).reshape(dims).copy(); o += cnt * dt.itemsize
    return out
