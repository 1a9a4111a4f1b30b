"""Image writers for the display step (no dependencies): PNG (8-bit sRGB, as the back buffer shows it), PFM and OpenEXR
(the linear RGBA32F resultTexture).  Rows arrive bottom-up (Unity texture convention) and are flipped for PNG."""
import struct
import zlib

import numpy as np


def write_png(path: str, rgba8: np.ndarray):
    """rgba8: (rows, W, 4) uint8 from Tracer.read_display(), row 0 = bottom."""
    a = np.ascontiguousarray(rgba8[::-1, :, :3], dtype=np.uint8)
    h, w, _ = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_pfm(path: str, rgba: np.ndarray):
    """Linear float image, RGB channels; PFM stores rows bottom-up, like the tracer."""
    a = np.ascontiguousarray(rgba[:, :, :3], dtype="<f4")
    with open(path, "wb") as f:
        f.write(f"PF\n{a.shape[1]} {a.shape[0]}\n-1.0\n".encode())
        f.write(a.tobytes())


def write_exr(path: str, rgba: np.ndarray, half: bool = False):
    """OpenEXR 2.0 single-part scanline file, no compression, channels A B G R as FLOAT (or HALF): the linear resultTexture
    with alpha.  EXR rows run top-down, the tracer's bottom-up, so rows are flipped."""
    a = np.ascontiguousarray(rgba[::-1], dtype=np.float32)
    h, w, c = a.shape
    assert c == 4
    ptype, dt = (1, "<f2") if half else (2, "<f4")

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chlist = b"".join(n + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for n in (b"A", b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (struct.pack("<ii", 20000630, 2)
              + attr("channels", "chlist", chlist) + attr("compression", "compression", b"\0")
              + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
              + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
              + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
              + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    row_bytes = 4 * w * np.dtype(dt).itemsize
    first = len(header) + 8 * h
    offsets = first + np.arange(h, dtype="<u8") * (8 + row_bytes)
    planar = np.ascontiguousarray(a[:, :, [3, 2, 1, 0]].transpose(0, 2, 1)).astype(dt)      # [row][A,B,G,R][x]
    with open(path, "wb") as f:
        f.write(header)
        f.write(offsets.tobytes())
        for y in range(h):
            f.write(struct.pack("<ii", y, row_bytes))
            f.write(planar[y].tobytes())


def read_exr(path: str) -> np.ndarray:
    """Reads back what write_exr wrote (uncompressed scanline A B G R, FLOAT or HALF); returns (rows, W, 4) float32 bottom-up."""
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != 20000630 or (version & 0xFF) != 2 or (version & ~0xFF):
        raise ValueError("not a single-part scanline OpenEXR 2.0 file")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        e = buf.index(b"\0", pos); name = buf[pos:e].decode(); pos = e + 1
        e = buf.index(b"\0", pos); typ = buf[pos:e].decode(); pos = e + 1
        (n,) = struct.unpack_from("<i", buf, pos); pos += 4
        attrs[name] = (typ, buf[pos:pos + n]); pos += n
    pos += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("compressed EXR not supported")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, cp, ch = [], 0, attrs["channels"][1]
    while ch[cp] != 0:
        e = ch.index(b"\0", cp); names.append(ch[cp:e].decode()); cp = e + 1
        (ptype,) = struct.unpack_from("<i", ch, cp); cp += 16
    dt = {1: "<f2", 2: "<f4"}[ptype]
    offsets = np.frombuffer(buf, "<u8", h, pos)
    out = np.empty((h, w, 4), np.float32)
    for y in range(h):
        o = int(offsets[y])
        yy, nbytes = struct.unpack_from("<ii", buf, o)
        rows = np.frombuffer(buf, dt, len(names) * w, o + 8).reshape(len(names), w)
        for i, nme in enumerate(names):
            out[yy - y0, :, "RGBA".index(nme)] = rows[i]
    return out[::-1].copy()
