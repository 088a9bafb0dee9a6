"""The C++ host mirror's PNG reader (toy-cpu-pathtracing_amd/host/renderer.hpp load_png: own inflate + unfiltering) against
PNGs written here with every filter type, colour type and bit depth the reference's `image::open(..).to_rgb8()` path
(scene/src/texture/loader.rs:44-64) would meet.  No GPU: the tool only links the C ABI library."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "toy-cpu-pathtracing_amd", "host", "png2ppm")


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _write_png(path, rows, w, h, depth, ctype, bpp, plte=None, level=6):
    """rows: list of bytes per scanline (unfiltered); filter type cycles 0..4 over the rows."""
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, line in enumerate(rows):
        ft = y % 5
        out = bytearray(len(line))
        for i, x in enumerate(line):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, _paeth(a, b, c)][ft]
            out[i] = (x - pred) & 0xff
        raw.append(ft); raw += out
        prev = line

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    comp = zlib.compress(bytes(raw), level)
    body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if plte is not None:
        body += chunk(b"PLTE", plte)
    half = len(comp) // 2                      # two IDAT chunks: the stream must be concatenated
    body += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")
    open(path, "wb").write(body)


def _decode(tmp_path, name):
    out = str(tmp_path / (name + ".ppm"))
    r = subprocess.run([TOOL, str(tmp_path / (name + ".png")), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = open(out, "rb").read()
    hdr, rest = d.split(b"\n255\n", 1)
    w, h = map(int, hdr.split()[1:3])
    return np.frombuffer(rest, np.uint8).reshape(h, w, 3)


@pytest.mark.skipif(not os.path.exists(TOOL), reason="host tools not built (python __graft_entry__.py)")
@pytest.mark.parametrize("level", [0, 1, 9])
def test_png_reader_colour_types(tmp_path, level):
    rng = np.random.default_rng(level)
    w, h = 37, 23
    smooth = (np.add.outer(np.arange(h) * 3, np.arange(w) * 5) % 256).astype(np.uint8)
    rgb = np.stack([smooth, rng.integers(0, 256, (h, w)).astype(np.uint8), 255 - smooth], -1)
    # RGB8
    _write_png(tmp_path / "rgb8.png", [rgb[y].tobytes() for y in range(h)], w, h, 8, 2, 3, level=level)
    assert np.array_equal(_decode(tmp_path, "rgb8"), rgb)
    # RGBA8 (alpha dropped)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1)).astype(np.uint8)], -1)
    _write_png(tmp_path / "rgba8.png", [rgba[y].tobytes() for y in range(h)], w, h, 8, 6, 4, level=level)
    assert np.array_equal(_decode(tmp_path, "rgba8"), rgb)
    # grey8 -> replicated
    _write_png(tmp_path / "g8.png", [smooth[y].tobytes() for y in range(h)], w, h, 8, 0, 1, level=level)
    assert np.array_equal(_decode(tmp_path, "g8"), np.repeat(smooth[..., None], 3, -1))
    # RGB16: high byte kept
    rgb16 = (rgb.astype(np.uint16) << 8) | rng.integers(0, 256, (h, w, 3)).astype(np.uint16)
    _write_png(tmp_path / "rgb16.png", [rgb16[y].astype(">u2").tobytes() for y in range(h)], w, h, 16, 2, 6, level=level)
    assert np.array_equal(_decode(tmp_path, "rgb16"), rgb)
    # palette, 8 bit
    pal = rng.integers(0, 256, (256, 3)).astype(np.uint8)
    idx = rng.integers(0, 256, (h, w)).astype(np.uint8)
    _write_png(tmp_path / "pal8.png", [idx[y].tobytes() for y in range(h)], w, h, 8, 3, 1, plte=pal.tobytes(), level=level)
    assert np.array_equal(_decode(tmp_path, "pal8"), pal[idx])
    # grey, 1 bit
    bits = rng.integers(0, 2, (h, w)).astype(np.uint8)
    _write_png(tmp_path / "g1.png", [np.packbits(bits[y]).tobytes() for y in range(h)], w, h, 1, 0, 1, level=level)
    assert np.array_equal(_decode(tmp_path, "g1"), np.repeat((bits * 255)[..., None], 3, -1))


@pytest.mark.skipif(not os.path.exists(TOOL), reason="host tools not built (python __graft_entry__.py)")
def test_png_reader_rejects_lfs_stub(tmp_path):
    (tmp_path / "stub.png").write_text("version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    r = subprocess.run([TOOL, str(tmp_path / "stub.png"), str(tmp_path / "o.ppm")], capture_output=True, text=True)
    assert r.returncode != 0 and "not a PNG" in r.stderr
