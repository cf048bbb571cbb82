"""Minimal PNG codec on zlib (no imageio / cv2 in this image): 8-bit gray / RGB / RGBA, non-interlaced,
all five scanline filters on read; RGB or RGBA, filter 0 on write.  Enough for DONeRF dataset images
(reference loader: src/datasets.py:275-287, imageio.imread(...)[:, :, :3] / 255)."""
import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def read_png(path: str) -> np.ndarray:
    """-> uint8 [h, w, c] (c = 1, 3 or 4)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _SIG:
        raise ValueError("%s: not a PNG file" % path)
    pos = 8
    idat = bytearray()
    w = h = depth = ctype = interlace = None
    while pos < len(data):
        (length,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat += body
        elif kind == b"IEND":
            break
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 6):
        raise ValueError("%s: unsupported PNG (bit depth %s, colour type %s, interlace %s)" % (path, depth, ctype, interlace))
    c = {0: 1, 2: 3, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(bytes(idat)), dtype=np.uint8)
    stride = w * c
    raw = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = int(raw[y, 0])
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):
            cur = np.zeros(stride, dtype=np.int32)
            for x in range(stride):        # left-dependent filters: sequential per byte
                a = cur[x - c] if x >= c else 0
                b = prev[x]
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    cc = prev[x - c] if x >= c else 0
                    pa, pb, pc = abs(b - cc), abs(a - cc), abs(a + b - 2 * cc)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                cur[x] = (line[x] + p) & 255
        else:
            raise ValueError("%s: bad filter type %d" % (path, ft))
        out[y] = cur
        prev = cur
    return out.reshape(h, w, c)


def write_png(path: str, img: np.ndarray) -> None:
    """uint8 [h, w, 3|4] -> PNG (filter 0, zlib level 6)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    if c not in (3, 4):
        raise ValueError("write_png expects RGB or RGBA")
    raw = np.zeros((h, w * c + 1), dtype=np.uint8)
    raw[:, 1:] = img.reshape(h, w * c)

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_SIG)
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))
