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


# ------------------------------------------------------------------------------------------------ OpenEXR (environment maps)
EXR_TOOL = os.path.join(ROOT, "toy-cpu-pathtracing_amd", "host", "exr2pfm")


def _write_exr(path, chans, w, h, compression, line_order=0, y_min=0):
    """chans: {name: 2-D array}, dtype float16 (HALF) or float32 (FLOAT).  Scanline EXR, compression 0 NONE / 2 ZIPS / 3 ZIP."""
    names = sorted(chans)                                               # channels are stored in alphabetical order

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 1 if chans[n].dtype == np.float16 else 2, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, y_min, w - 1, y_min + h - 1)
    hdr = (struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([compression])) +
           attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([line_order])) +
           attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) +
           attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lines = 16 if compression == 3 else 1
    blocks = []
    order = range(0, h, lines) if line_order == 0 else reversed(range(0, h, lines))
    for y0 in order:
        raw = b"".join(chans[n][y].tobytes() for y in range(y0, min(y0 + lines, h)) for n in names)
        data = raw
        if compression:
            half = (len(raw) + 1) // 2
            t = bytearray(raw[0::2] + raw[1::2])                        # interleave: even bytes, then odd bytes
            assert len(raw[0::2]) == half
            p = t[0]
            for i in range(1, len(t)):                                  # byte predictor
                d = (t[i] - p + 128) & 0xff
                p = t[i]; t[i] = d
            z = zlib.compress(bytes(t), 6)
            data = z if len(z) < len(raw) else raw                      # OpenEXR stores the block raw when deflate does not shrink it
        blocks.append(struct.pack("<ii", y_min + y0, len(data)) + data)
    off = len(hdr) + 8 * len(blocks)
    table = b""
    for b in blocks:
        table += struct.pack("<Q", off); off += len(b)
    with open(path, "wb") as f:
        f.write(hdr + table + b"".join(blocks))


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split()); scale = float(f.readline())
        a = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return a[::-1]                                                      # PFM rows run bottom to top


@pytest.mark.parametrize("compression", [0, 2, 3])
@pytest.mark.parametrize("dtype,extra,line_order", [(np.float16, False, 0), (np.float32, True, 0), (np.float16, True, 1), (np.float32, False, 1)])
def test_exr_reader_matches_written_pixels(tmp_path, compression, dtype, extra, line_order):
    """The host mirror's OpenEXR reader (renderer.hpp load_exr, what EnvironmentLightPrimitive loads: environment_light.rs:30-41 reads the
    sky through image::open(..).to_rgb32f()) against files written here: NONE / ZIPS / ZIP scanline blocks (37 rows: a ragged last ZIP block),
    HALF and FLOAT channels, an extra A channel to skip, both line orders, a data window that does not start at 0."""
    if not os.path.exists(EXR_TOOL):
        subprocess.check_call(["make", "-C", os.path.dirname(EXR_TOOL)])
    rng = np.random.default_rng(compression * 10 + line_order)
    w, h = 53, 37
    rgb = (rng.random((h, w, 3)) ** 4 * 60.0).astype(dtype)             # HDR range, sky-like
    rgb[3, 5] = [0.0, 6.0e-8 if dtype == np.float16 else 1e-30, 65504.0 if dtype == np.float16 else 3e38]   # zero, a subnormal half, the largest half
    chans = {"R": rgb[..., 0].copy(), "G": rgb[..., 1].copy(), "B": rgb[..., 2].copy()}
    if extra:
        chans["A"] = np.ones((h, w), dtype)
    src, out = str(tmp_path / "sky.exr"), str(tmp_path / "sky.pfm")
    _write_exr(src, chans, w, h, compression, line_order, y_min=7)
    r = subprocess.run([EXR_TOOL, src, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(_read_pfm(out), rgb.astype(np.float32))


def test_exr_reader_refuses_what_it_does_not_decode(tmp_path):
    rgb = np.ones((4, 4), np.float16)
    src = str(tmp_path / "piz.exr")
    _write_exr(src, {"R": rgb, "G": rgb, "B": rgb}, 4, 4, 0)
    data = bytearray(open(src, "rb").read())
    i = data.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    data[i] = 4                                                         # PIZ
    open(src, "wb").write(bytes(data))
    r = subprocess.run([EXR_TOOL, src, str(tmp_path / "o.pfm")], capture_output=True, text=True)
    assert r.returncode == 1 and "not supported" in r.stderr


EXR_ASAN = EXR_TOOL + "_asan"


def _run_asan(src, out):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")
    return subprocess.run([EXR_ASAN, src, out], capture_output=True, text=True, env=env, timeout=60)


def test_exr_reader_rejects_truncated_and_corrupt_headers_under_asan(tmp_path):
    """The reader trusts nothing in the header: every truncation of a valid file and a set of targeted corruptions — a channel
    list without its terminator, zero-sized compression / lineOrder / dataWindow attributes, an inverted, an overflowing and an absurdly
    large data window, a NONE chunk whose size field lies — must end in the reader's own error (exit code 1), never in a sanitizer report
    (AddressSanitizer + UBSan build of the same tool, CPU only) and never in a multi-gigabyte allocation."""
    subprocess.check_call(["make", "-C", os.path.dirname(EXR_TOOL), "exr2pfm_asan"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(5)
    w, h = 9, 5
    rgb = rng.random((h, w, 3)).astype(np.float16)
    chans = {"R": rgb[..., 0].copy(), "G": rgb[..., 1].copy(), "B": rgb[..., 2].copy()}
    good = str(tmp_path / "good.exr")
    _write_exr(good, chans, w, h, 0)
    blob = open(good, "rb").read()
    out = str(tmp_path / "o.pfm")
    r = _run_asan(good, out)
    assert r.returncode == 0, r.stderr

    def check(data, what):
        p = str(tmp_path / "bad.exr")
        open(p, "wb").write(bytes(data))
        r = _run_asan(p, out)
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (what, r.stderr[-2000:])
        assert r.returncode == 1, (what, r.returncode, r.stderr[-500:])

    for n in list(range(0, 420, 1)) + [len(blob) - 1, len(blob) - 40]:      # every truncation through the header, and inside the pixel data
        if n < len(blob):
            check(blob[:n], f"truncated at {n}")

    def at(name, typ):       # offset of the attribute's size field
        key = name.encode() + b"\0" + typ.encode() + b"\0"
        return blob.index(key) + len(key)

    i = at("channels", "chlist")
    size = struct.unpack_from("<i", blob, i)[0]
    d = bytearray(blob); d[i + 4 + size - 1] = ord("Z")                        # the list's terminating NUL becomes a name byte
    check(d, "channel list without terminator")
    d = bytearray(blob); d[i + 4:i + 4 + size] = b"R" * size                   # one endless name
    check(d, "endless channel name")
    for name, typ, payload in (("compression", "compression", 1), ("lineOrder", "lineOrder", 1), ("dataWindow", "box2i", 16)):
        i = at(name, typ)
        d = bytearray(blob)
        d[i:i + 4 + payload] = struct.pack("<i", 0)                            # size 0, payload removed
        check(d, f"zero-sized {name}")
    i = at("dataWindow", "box2i") + 4
    for box, what in (((5, 0, 2, 4), "xMax < xMin"), ((0, 3, 8, 1), "yMax < yMin"), ((-2147483648, 0, 2147483647, 4), "overflowing width"),
                      ((0, 0, 60000, 60000), "3.6 G pixels"), ((0, 0, 70000, 4), "too wide")):
        d = bytearray(blob); d[i:i + 16] = struct.pack("<4i", *box)
        check(d, what)
    j = blob.index(struct.pack("<ii", 0, w * 3 * 2))                           # first chunk: y = 0, size = one line of three HALF channels
    d = bytearray(blob); d[j + 4:j + 8] = struct.pack("<i", w * 3 * 2 - 6)     # NONE chunk shorter than a line
    check(d, "short NONE chunk")
    d = bytearray(blob); d[j:j + 4] = struct.pack("<i", 77)                    # chunk outside the window
    check(d, "chunk outside the data window")
