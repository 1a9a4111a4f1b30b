"""Image writers for the display step (no dependencies): PNG (8-bit sRGB, as the back buffer shows it) and PFM (the
linear RGBA32F resultTexture).  Rows arrive bottom-up (Unity texture convention) and are flipped for PNG."""
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
