"""Shared comparison helpers for the parity tests."""
import numpy as np


def rel_err(got, want):
    """Per-texel, per-channel relative error on RGB, guarded for exact zeros."""
    got = np.asarray(got, np.float64)[..., :3]
    want = np.asarray(want, np.float64)[..., :3]
    den = np.maximum(np.abs(want), 1e-30)
    err = np.abs(got - want) / den
    err[(want == 0) & (got == 0)] = 0.0
    return err


def byte_identical_fraction(a_rgbe, b_rgbe):
    a = np.asarray(a_rgbe).reshape(-1, 4)
    b = np.asarray(b_rgbe).reshape(-1, 4)
    return float((a == b).all(axis=1).mean())


def decode_strip(orc, strip):
    """(6W, W, 4) uint8 -> (6, W, W, 4) float32 through the oracle's decoder."""
    w = strip.shape[1]
    return orc.rgbe_decode(strip).reshape(6, w, w, 4)


# ---- a tiny independent PNG codec for the CLI tests (zlib only) -----------------------

def filter_row(cur, prev, ch, ft):
    """One PNG scanline (int32 byte values, `ch` bytes per pixel) under filter type ft -> uint8 bytes."""
    left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
    upleft = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
    if ft == 0:
        out = cur
    elif ft == 1:
        out = cur - left
    elif ft == 2:
        out = cur - prev
    elif ft == 3:
        out = cur - (left + prev) // 2
    else:
        p = left + prev - upleft
        pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
        pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        out = cur - pred
    return (out & 255).astype(np.uint8)


def write_png(path, rgba, color_type=6, filters=(0,)):
    """rgba: (H, W, 4) uint8.  color_type 6 (RGBA) or 2 (RGB); `filters` cycles per row."""
    import struct
    import zlib
    img = np.asarray(rgba, np.uint8)
    if color_type == 2:
        img = img[..., :3]
    h, w, ch = img.shape
    rows = []
    prev = np.zeros((w * ch,), np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        ft = filters[y % len(filters)]
        rows.append(bytes([ft]) + filter_row(cur, prev, ch, ft).tobytes())
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
    z = zlib.compress(b"".join(rows), 6)
    half = len(z) // 2
    data += chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)


ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))


def write_png_general(path, samples, depth, color_type, interlace=False, palette=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """Any PNG the format allows: samples (H, W, C) of raw sample values (C = 1, 2, 3 or 4 by colour
    type), `depth` bits each, optionally Adam7-interlaced, with PLTE / tRNS chunks as given."""
    import struct
    import zlib
    smp = np.asarray(samples, np.uint16)
    h, w, ch = smp.shape
    bpp = max(1, depth * ch // 8)

    def pack_rows(sub):
        """(ph, pw, C) samples -> filtered scanlines of one (sub-)image."""
        ph, pw, _ = sub.shape
        out, prev = [], None
        for y in range(ph):
            flat = sub[y].reshape(-1)
            if depth == 16:
                row = np.stack([flat >> 8, flat & 255], axis=1).reshape(-1).astype(np.int32)
            elif depth == 8:
                row = flat.astype(np.int32)
            else:
                bits = np.zeros(((len(flat) * depth + 7) // 8) * 8, np.uint8)
                for b in range(depth):
                    bits[np.arange(len(flat)) * depth + b] = (flat >> (depth - 1 - b)) & 1
                row = np.packbits(bits).astype(np.int32)
            if prev is None:
                prev = np.zeros_like(row)
            ft = filters[y % len(filters)]
            out.append(bytes([ft]) + filter_row(row, prev, bpp, ft).tobytes())
            prev = row
        return b"".join(out)

    if interlace:
        raw = b"".join(pack_rows(smp[y0::dy, x0::dx]) for x0, y0, dx, dy in ADAM7
                       if smp[y0::dy, x0::dx].shape[0] and smp[y0::dy, x0::dx].shape[1])
    else:
        raw = pack_rows(smp)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    data += chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)


def read_png(path):
    """-> (H, W, 4) uint8 for the 8-bit RGBA files png_io writes."""
    import struct
    import zlib
    with open(path, "rb") as f:
        data = f.read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h, ct = 8, b"", 0, 0, 0
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(t + body) & 0xffffffff
        if t == b"IHDR":
            w, h, depth, ct, _, _, il = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and ct == 6 and il == 0
        elif t == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
    return unfilter_rows(raw)


def unfilter_rows(rows):
    """(H, 4W+1) uint8 PNG scanlines of an RGBA8 image (filter type first) -> (H, W, 4) uint8."""
    rows = np.asarray(rows, np.uint8)
    h, w = rows.shape[0], (rows.shape[1] - 1) // 4
    out = np.zeros((h, w, 4), np.int32)
    zero = np.zeros((w, 4), np.int32)
    for y in range(h):
        ft, row = int(rows[y, 0]), rows[y, 1:].astype(np.int32).reshape(w, 4)
        up = out[y - 1] if y else zero
        if ft == 0:
            out[y] = row
        elif ft == 1:
            out[y] = np.cumsum(row, axis=0) & 255
        elif ft == 2:
            out[y] = (row + up) & 255
        elif ft in (3, 4):
            left = np.zeros(4, np.int32)
            upleft = np.zeros(4, np.int32)
            for x in range(w):
                b = up[x]
                if ft == 3:
                    pred = (left + b) >> 1
                else:
                    p = left + b - upleft
                    pa, pb, pc = np.abs(p - left), np.abs(p - b), np.abs(p - upleft)
                    pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, b, upleft))
                left = (row[x] + pred) & 255
                out[y, x] = left
                upleft = b
        else:
            raise AssertionError("bad filter type %d" % ft)
    return out.astype(np.uint8)
