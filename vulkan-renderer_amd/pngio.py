"""PNG reader/writer for the Python-side tools (RGBA out).  Reading goes through the cube
tool's own C++ reader (host/png_io.cpp, built as libibl_png.so) when that library is present --
strips written by the reference's tool are full of average and Paeth rows, which a Python loop
unfilters a byte at a time -- and falls back to a small numpy/zlib reader (8-bit,
non-interlaced RGB/RGBA) otherwise.  The reference goes through stb_image for the same job
(cube.cpp:141,322,458)."""
import ctypes
import os
import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libibl_png.so")
_lib = None


def _native():
    global _lib
    if _lib is None and os.path.exists(_LIB_PATH):
        lib = ctypes.CDLL(_LIB_PATH)
        lib.ibl_png_read_rgba8.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                           ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_char_p, ctypes.c_size_t]
        lib.ibl_png_read_rgba8.restype = ctypes.c_int
        lib.ibl_image_read_rgba8.argtypes = lib.ibl_png_read_rgba8.argtypes
        lib.ibl_image_read_rgba8.restype = ctypes.c_int
        lib.ibl_png_free.argtypes = [ctypes.POINTER(ctypes.c_uint8)]
        lib.ibl_png_free.restype = None
        _lib = lib
    return _lib


def read_image(path):
    """Any input format the cube tool reads (PNG, BMP, binary PGM/PPM, Radiance HDR, TGA; recognised by
    content and converted to RGBA8 by stb_image's rules, host/image_formats.cpp) -> (H, W, 4) uint8.
    Needs the native library."""
    lib = _native()
    if lib is None:
        raise RuntimeError("libibl_png.so is not built")
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    px = ctypes.POINTER(ctypes.c_uint8)()
    err = ctypes.create_string_buffer(256)
    if lib.ibl_image_read_rgba8(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), ctypes.byref(px), err, 256) != 0:
        raise ValueError(err.value.decode() or "cannot read %s" % path)
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, 4)).copy()
    finally:
        lib.ibl_png_free(px)


def read_rgba8(path, native=True):
    """-> (H, W, 4) uint8."""
    lib = _native() if native else None
    if lib is not None:
        w, h = ctypes.c_uint32(), ctypes.c_uint32()
        px = ctypes.POINTER(ctypes.c_uint8)()
        err = ctypes.create_string_buffer(256)
        if lib.ibl_png_read_rgba8(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), ctypes.byref(px), err, 256) != 0:
            raise ValueError(err.value.decode() or "cannot read %s" % path)
        try:
            return np.ctypeslib.as_array(px, shape=(h.value, w.value, 4)).copy()
        finally:
            lib.ibl_png_free(px)
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _SIG:
        raise ValueError("not a PNG file: %s" % path)
    pos, idat, w, h, ct = 8, [], 0, 0, 6
    while pos + 12 <= len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if t == b"IHDR":
            w, h, depth, ct, _, _, il = struct.unpack(">IIBBBBB", body)
            if depth != 8 or il != 0 or ct not in (2, 6):
                raise ValueError("only 8-bit non-interlaced RGB/RGBA PNGs are supported")
        elif t == b"IDAT":
            idat.append(body)
        elif t == b"IEND":
            break
        pos += 12 + n
    ch = 4 if ct == 6 else 3
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, w * ch + 1)
    out = np.zeros((h, w * ch), np.uint8)
    zero = np.zeros(w * ch, np.uint8)
    for y in range(h):
        ft, row = int(raw[y, 0]), raw[y, 1:]
        up = out[y - 1] if y else zero
        if ft == 0:
            out[y] = row
        elif ft == 2:
            out[y] = row + up                      # uint8 arithmetic wraps, as the format wants
        elif ft == 1:
            cur = row.reshape(w, ch).astype(np.uint32)
            out[y] = (np.cumsum(cur, axis=0) & 255).astype(np.uint8).reshape(-1)
        else:                                      # average and Paeth need the left neighbour: per pixel
            cur = np.zeros(w * ch, np.int32)
            r, u = row.astype(np.int32), up.astype(np.int32)
            for i in range(w * ch):
                a = cur[i - ch] if i >= ch else 0
                b = u[i]
                if ft == 3:
                    pred = (a + b) // 2
                else:
                    c = u[i - ch] if i >= ch else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (r[i] + pred) & 255
            out[y] = cur.astype(np.uint8)
    img = out.reshape(h, w, ch)
    if ch == 3:
        img = np.concatenate([img, np.full((h, w, 1), 255, np.uint8)], axis=2)
    return img


def write_rgba8(path, rgba, level=4):
    """rgba: (H, W, 4) uint8; rows filtered with "up" like the C++ writer."""
    img = np.ascontiguousarray(rgba, np.uint8)
    h, w, ch = img.shape
    assert ch == 4
    flat = img.reshape(h, w * 4)
    rows = np.empty((h, w * 4 + 1), np.uint8)
    rows[0, 0] = 0
    rows[0, 1:] = flat[0]
    rows[1:, 0] = 2
    rows[1:, 1:] = flat[1:] - flat[:-1]

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(rows.tobytes(), level)) + chunk(b"IEND", b""))
