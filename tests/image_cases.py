"""Image files for the texture-loader tests, made with Pillow (and, for Adam7 PNGs, by hand): every flavour the
loader of raytracer_amd/csrc/rtx_image.cpp + rtx_jpeg.cpp claims to decode like the reference's stb_image."""
import io
import itertools
import struct
import zlib

import numpy as np


def picture(rng, w, h, kind):
    y, x = np.mgrid[0:h, 0:w]
    if kind == 0:
        a = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 90 * np.cos(x / 11.0 + y / 3.0), (x * 3 + y * 5) % 256], -1)
    elif kind == 1:
        a = rng.integers(0, 256, (h, w, 3))
    else:
        a = np.stack([(x // 8 + y // 8) % 2 * 255, (x // 3) % 2 * 255, np.full((h, w), 40)], -1)
    return np.clip(a, 0, 255).astype(np.uint8)


def _save(im, fmt, **kw):
    buf = io.BytesIO()
    im.save(buf, fmt, **kw)
    return buf.getvalue()


def adam7_png(a, depth16=False):
    """An interlaced PNG (Pillow cannot write one): filter type 0 rows, seven passes."""
    h, w, ch = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = a[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            data = row.astype(">u2").tobytes() if depth16 else row.astype(np.uint8).tobytes()
            raw += bytes([0]) + data

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16 if depth16 else 8, ctype, 0, 0, 1)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def jpeg_cases(rng, keep=0.08):
    from PIL import Image
    sizes = [(64, 48), (33, 17), (1, 1), (7, 100), (130, 131), (16, 16)]
    for (w, h), kind, q, sub, prog, opt, gray, rst in itertools.product(sizes, [0, 1, 2], [30, 75, 95, 100], [0, 1, 2], [False, True],
                                                                       [False, True], [False, True], [0, 3]):
        if rng.random() > keep:
            continue
        a = picture(rng, w, h, kind)
        im = Image.fromarray(a[:, :, 0] if gray else a)
        kw = dict(quality=q, progressive=prog, optimize=opt)
        if not gray:
            kw["subsampling"] = sub
        if rst:
            kw["restart_marker_blocks"] = rst
        name = "jpeg_%dx%d_k%d_q%d_s%d_%s%s%s%s.jpg" % (w, h, kind, q, sub, "p" if prog else "b", "o" if opt else "", "g" if gray else "", "r" if rst else "")
        yield name, _save(im, "JPEG", **kw)
    a = picture(rng, 40, 24, 0)
    yield "jpeg_cmyk.jpg", _save(Image.fromarray(np.concatenate([a, a[:, :, :1]], -1), "CMYK"), "JPEG", quality=90)


def lossless_cases(rng):
    from PIL import Image
    for (w, h) in [(37, 23), (1, 1), (8, 8), (5, 9)]:
        a = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
        rgb = Image.fromarray(a[:, :, :3])
        tag = "%dx%d" % (w, h)
        yield "rgb_%s.png" % tag, _save(rgb, "PNG")
        yield "rgba_%s.png" % tag, _save(Image.fromarray(a, "RGBA"), "PNG")
        yield "l_%s.png" % tag, _save(Image.fromarray(a[:, :, 0]), "PNG")
        yield "la_%s.png" % tag, _save(Image.fromarray(a[:, :, :2], "LA"), "PNG")
        for colors in (64, 16, 4, 2):
            yield "p%d_%s.png" % (colors, tag), _save(rgb.convert("P", palette=Image.ADAPTIVE, colors=colors), "PNG")
        for bits in (1, 2, 4):
            yield "l%d_%s.png" % (bits, tag), _save(Image.fromarray(a[:, :, 0] >> (8 - bits) << (8 - bits)), "PNG", bits=bits)
        yield "i16_%s.png" % tag, _save(Image.fromarray(a[:, :, 0].astype(np.uint16) * 257 + 3), "PNG")
        yield "adam7_rgb_%s.png" % tag, adam7_png(a[:, :, :3])
        yield "adam7_rgba_%s.png" % tag, adam7_png(a)
        yield "adam7_l16_%s.png" % tag, adam7_png(a[:, :, :1].astype(np.uint16) * 251, depth16=True)
        yield "rgb_%s.bmp" % tag, _save(rgb, "BMP")
        yield "rgb_%s.tga" % tag, _save(rgb, "TGA")
        yield "rgba_%s.tga" % tag, _save(Image.fromarray(a, "RGBA"), "TGA")
        yield "rle_%s.tga" % tag, _save(rgb, "TGA", compression="tga_rle")
        yield "l_%s.tga" % tag, _save(Image.fromarray(a[:, :, 0]), "TGA")
        yield "rgb_%s.ppm" % tag, _save(rgb, "PPM")
        yield "l_%s.pgm" % tag, _save(Image.fromarray(a[:, :, 0]), "PPM")
