"""Image comparison for frame dumps: PPM (P6) or PNG (8-bit RGB/RGBA, non-interlaced) in, numbers out.

  python tools/imgdiff.py a.png b.ppm [--out diff.png] [--max-codes 1] [--max-fraction 0.02]

Prints size, the fraction of differing samples, the largest difference in 8-bit codes, RMSE and PSNR; exits 1 when the
images differ by more than --max-codes anywhere or in more than --max-fraction of the samples (defaults: the back-buffer
bar of the parity tests), 2 when the sizes differ.  --out writes |a - b| scaled to the full range.  Standard library +
numpy only; `load` / `save_png` are importable (tests use them to read what the C++ host wrote)."""
import struct
import sys
import zlib

import numpy as np


def load(path):
    """-> uint8 array [H, W, 3 or 4]"""
    data = open(path, "rb").read()
    if data[:2] == b"P6":
        fields, pos = [], 2
        while len(fields) < 3:                       # width, height, maxval, separated by whitespace / # comments
            while data[pos:pos + 1].isspace():
                pos += 1
            if data[pos:pos + 1] == b"#":
                pos = data.index(b"\n", pos) + 1
                continue
            end = pos
            while not data[end:end + 1].isspace():
                end += 1
            fields.append(int(data[pos:end])); pos = end
        w, h, maxval = fields
        if maxval != 255:
            raise ValueError("%s: only 8-bit PPM" % path)
        return np.frombuffer(data, np.uint8, w * h * 3, pos + 1).reshape(h, w, 3)
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s: neither P6 nor PNG" % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if zlib.crc32(typ + body) != struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]:
            raise ValueError("%s: bad CRC in %s" % (path, typ))
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in (2, 6) or interlace:
        raise ValueError("%s: only 8-bit RGB/RGBA, non-interlaced" % path)
    c = 3 if ctype == 2 else 4
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, w * c + 1)
    out = np.zeros((h, w * c), np.uint8)
    prev = np.zeros(w * c, np.int32)
    for y in range(h):                               # undo the scanline filters (PNG spec 9.2)
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(w * c, np.int32)
            for i in range(w * c):
                a = cur[i - c] if i >= c else 0
                b = prev[i]
                cc = prev[i - c] if i >= c else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - cc), abs(a - cc), abs(a + b - 2 * cc)
                    p = a if pa <= pb and pa <= pc else (b if pb <= pc else cc)
                cur[i] = (line[i] + p) & 255
        out[y] = cur; prev = cur
    return out.reshape(h, w, c)


def save_png(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    raw = np.zeros((h, w * c + 1), np.uint8); raw[:, 1:] = img.reshape(h, w * c)
    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body))
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + chunk(b"IEND", b""))


def compare(a, b):
    a, b = a[..., :3].astype(np.int32), b[..., :3].astype(np.int32)
    d = np.abs(a - b)
    mse = float((d.astype(np.float64) ** 2).mean())
    return {"differing": float((d != 0).mean()), "max_codes": int(d.max()), "rmse": mse ** 0.5,
            "psnr_db": float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse), "diff": d}


def main(argv):
    args = [x for x in argv if not x.startswith("--")]
    opt = {argv[i][2:]: argv[i + 1] for i in range(len(argv) - 1) if argv[i].startswith("--")}
    args = [x for x in args if x not in opt.values()]
    if len(args) != 2:
        print(__doc__); return 2
    a, b = load(args[0]), load(args[1])
    if a.shape[:2] != b.shape[:2]:
        print("sizes differ: %s vs %s" % (a.shape[:2], b.shape[:2])); return 2
    r = compare(a, b)
    print("%dx%d  differing samples %.4f %%  max difference %d codes  RMSE %.4f  PSNR %.2f dB" % (a.shape[1], a.shape[0], 100 * r["differing"], r["max_codes"], r["rmse"], r["psnr_db"]))
    if "out" in opt:
        d = r["diff"]; save_png(opt["out"], (d * (255 // max(int(d.max()), 1))).astype(np.uint8))
    ok = r["max_codes"] <= int(opt.get("max-codes", 1)) and r["differing"] <= float(opt.get("max-fraction", 0.02))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
