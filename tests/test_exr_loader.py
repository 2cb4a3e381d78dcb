"""The OpenEXR skybox reader (rt_exr_read_rgba32f, Connection.load_image — connection.rs:154-162) against
files written by tests/exr_writer.py: every supported compression, HALF / FLOAT / UINT channels, both PIZ
wavelet modes, tiled and multi-part files, windows, line orders, raw-stored chunks, alpha handling and the reference's error texts.
Host code only — runs without a GPU.  PARITY UNPINNED for files of other EXR writers (none exists here)."""
import os

import numpy as np
import pytest

import exr_writer as X
from rsraytracingv3_amd import Connection
from rsraytracingv3_amd import _native as N


def _image(h, w, seed, smooth=True):
    rs = np.random.RandomState(seed)
    if smooth:  # gradients + a few bright spots, like a panorama: compressible, many distinct halfs
        y, x = np.mgrid[0:h, 0:w].astype(np.float32)
        base = np.stack([0.2 + x / w, 0.1 + y / h, 0.5 + 0.25 * np.sin(x / 7.0) * np.cos(y / 5.0)], -1)
        base[rs.randint(0, h, 5), rs.randint(0, w, 5)] *= 900.0
        return base.astype(np.float32)
    return np.exp(rs.uniform(-8, 8, (h, w, 3))).astype(np.float32)


def _expected(rgb, typ, alpha=None):
    h, w, _ = rgb.shape
    out = np.ones((h, w, 4), np.float32)
    for k in range(3):
        out[..., k] = rgb[..., k].astype(X.NP_TYPE[typ]).astype(np.float32)
    if alpha is not None:
        out[..., 3] = alpha.astype(np.float16).astype(np.float32)
    return out


@pytest.mark.parametrize("compression", [X.NONE, X.RLE, X.ZIPS, X.ZIP, X.PIZ])
@pytest.mark.parametrize("typ", ["half", "float"])
def test_round_trip_every_compression(tmp_path, compression, typ):
    rgb = _image(45, 70, 1)  # 45 rows: a ragged last chunk for ZIP (16) and PIZ (32)
    path = str(tmp_path / "a.exr")
    X.write_exr(path, {"R": (rgb[..., 0], typ), "G": (rgb[..., 1], typ), "B": (rgb[..., 2], typ)}, compression)
    (w, h), got = Connection.load_image(path)
    assert (w, h) == (70, 45) and got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), _expected(rgb, typ).view(np.uint32))


def test_piz_16_bit_wavelet_mode_and_alpha(tmp_path):
    """> 16384 distinct half values in a chunk switch PIZ to its 16-bit wavelet (wdec16); alpha is kept."""
    rgb = _image(64, 400, 2, smooth=False)
    alpha = np.random.RandomState(3).uniform(0, 1, (64, 400)).astype(np.float32)
    path = str(tmp_path / "b.exr")
    chans = {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "half"), "B": (rgb[..., 2], "half"), "A": (alpha, "half")}
    X.write_exr(path, chans, X.PIZ)
    block = np.stack([rgb[:32, :, k] for k in range(3)] + [alpha[:32]]).astype(np.float16).view(np.uint16)
    assert len(np.unique(block)) > 16384  # the case is what it claims to be
    _, got = Connection.load_image(path)
    assert np.array_equal(got.view(np.uint32), _expected(rgb, "half", alpha).view(np.uint32))


def test_piz_flat_and_tiny_images(tmp_path):
    """Runs (the Huffman run-length escape), a constant image (empty value bitmap) and sizes below the wavelet's
    smallest step."""
    for i, (h, w, fill) in enumerate([(40, 50, 0.5), (33, 20, 0.0), (1, 1, 2.0), (2, 3, 1.5), (5, 1, 0.25)]):
        rgb = np.full((h, w, 3), fill, np.float32)
        if h > 8:
            rgb[3:6, 4:9, 1] = 7.0
        path = str(tmp_path / ("c%d.exr" % i))
        X.write_exr(path, {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "half")}, X.PIZ)
        _, got = Connection.load_image(path)
        want = np.ones((h, w, 4), np.float32)
        want[..., :3] = rgb
        assert np.array_equal(got, want), (h, w)


def test_uint_channels_extra_channels_and_line_order(tmp_path):
    rs = np.random.RandomState(5)
    rgb = rs.randint(0, 1 << 20, (37, 29, 3)).astype(np.uint32)
    z = rs.uniform(0, 100, (37, 29)).astype(np.float32)
    path = str(tmp_path / "d.exr")
    X.write_exr(path, {"R": (rgb[..., 0], "uint"), "G": (rgb[..., 1], "uint"), "B": (rgb[..., 2], "uint"), "Z": (z, "float")},
                X.ZIP, decreasing_y=True)
    _, got = Connection.load_image(path)
    want = np.ones((37, 29, 4), np.float32)
    want[..., :3] = rgb.astype(np.float32)
    assert np.array_equal(got, want)


def test_data_window_inside_display_window(tmp_path):
    rgb = _image(20, 30, 7)
    path = str(tmp_path / "e.exr")
    X.write_exr(path, {"R": (rgb[..., 0], "float"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "float")}, X.ZIPS,
                display_window=(0, 0, 39, 31), data_origin=(6, 9))
    (w, h), got = Connection.load_image(path)
    assert (w, h) == (40, 32)
    want = np.zeros((32, 40, 4), np.float32)
    want[9:29, 6:36, :3] = rgb
    want[9:29, 6:36, 3] = 1.0
    assert np.array_equal(got, want)


def test_chunks_stored_raw_when_compression_does_not_help(tmp_path):
    rgb = _image(35, 33, 9, smooth=False)
    for comp in (X.RLE, X.ZIP, X.PIZ):
        path = str(tmp_path / ("f%d.exr" % comp))
        X.write_exr(path, {"R": (rgb[..., 0], "float"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "float")}, comp,
                    force_raw_chunks=True)
        _, got = Connection.load_image(path)
        assert np.array_equal(got[..., :3], rgb)


def test_error_texts_follow_the_reference(tmp_path):
    with pytest.raises(N.RtError, match="Can't find image"):  # connection.rs:156
        Connection.load_image(str(tmp_path / "missing.exr"))
    bad = tmp_path / "bad.exr"
    bad.write_bytes(b"not an exr file at all")
    with pytest.raises(N.RtError, match="Image invalid format"):  # connection.rs:159
        Connection.load_image(str(bad))
    rgb = _image(8, 8, 1)
    path = str(tmp_path / "g.exr")
    X.write_exr(path, {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "half")}, X.ZIP)
    with pytest.raises(N.RtError, match="no R, G, B"):
        Connection.load_image(path)
    X.write_exr(path, {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "half"), "B": (rgb[..., 2], "half")}, X.ZIP)
    data = bytearray(open(path, "rb").read())
    k = data.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    data[k] = 6  # B44
    open(path, "wb").write(bytes(data))
    with pytest.raises(N.RtError, match="B44 is not supported"):
        Connection.load_image(path)
    truncated = bytes(data[:len(data) // 2])
    data[k] = 3
    open(path, "wb").write(bytes(data[:len(data) - 40]))
    with pytest.raises(N.RtError, match="Image invalid format"):
        Connection.load_image(path)
    assert truncated


def _float24_expected(rgb):
    out = np.ones(rgb.shape[:2] + (4,), np.float32)
    out[..., :3] = (X.float_to_float24(rgb.view(np.uint32)) << 8).astype(np.uint32).view(np.float32)
    return out


@pytest.mark.filterwarnings("ignore:overflow encountered in cast")  # the largest float becomes a half infinity
@pytest.mark.parametrize("typ", ["half", "float", "uint"])
def test_pxr24(tmp_path, typ):
    """PXR24: HALF and UINT come back exactly, a FLOAT with its upper 24 bits (rounded by the writer: the codec is
    lossy for 32-bit floats by design); 45 rows = a ragged last block of the 16-line blocks; specials survive."""
    rgb = _image(45, 70, 21)
    rgb[3, 5] = [np.inf, -np.inf, np.nan]
    rgb[4, 6] = [3.4028235e38, -0.0, 1e-42]  # the largest float must not round into infinity
    path = str(tmp_path / "p.exr")
    if typ == "uint":
        u = np.random.RandomState(5).randint(0, 2 ** 32, (45, 70, 3), dtype=np.uint64).astype(np.uint32)
        u[:, :35] = (np.arange(35) * 3)[None, :, None]  # something deflate can shrink, so the chunks are not stored raw
        X.write_exr(path, {"R": (u[..., 0], typ), "G": (u[..., 1], typ), "B": (u[..., 2], typ)}, X.PXR24)
        _, got = Connection.load_image(path)
        assert np.array_equal(got[..., :3], u.astype(np.float32)) and np.all(got[..., 3] == 1.0)
        return
    X.write_exr(path, {"R": (rgb[..., 0], typ), "G": (rgb[..., 1], typ), "B": (rgb[..., 2], typ)}, X.PXR24)
    (w, h), got = Connection.load_image(path)
    assert (w, h) == (70, 45)
    want = _expected(rgb, typ) if typ == "half" else _float24_expected(rgb)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if typ == "float":
        assert np.isnan(got[3, 5, 2]) and got[3, 5, 0] == np.inf and np.isfinite(got[4, 6, 0])
        assert not np.array_equal(got[..., :3], rgb)  # the case is what it claims to be: 8 bits were dropped
        mixed = {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "half"),
                 "A": (rgb[..., 1], "float"), "Z": (rgb[..., 2], "float")}
        X.write_exr(path, mixed, X.PXR24, decreasing_y=True)
        _, got = Connection.load_image(path)
        assert np.array_equal(got[..., 0], rgb[..., 0].astype(np.float16).astype(np.float32))
        assert np.array_equal(got[..., 1].view(np.uint32), _float24_expected(rgb)[..., 1].view(np.uint32))
        assert np.array_equal(got[..., 3].view(np.uint32), _float24_expected(rgb)[..., 1].view(np.uint32))


@pytest.mark.parametrize("compression", [X.NONE, X.RLE, X.ZIPS, X.ZIP, X.PIZ, X.PXR24])
def test_tiled_files(tmp_path, compression):
    """Tiled parts: ragged right and bottom tiles (70 x 45 in tiles of 32 x 16), one tile larger than the image,
    tiles written bottom-up, and the data window inside / beyond the display window."""
    rgb = _image(45, 70, 31)
    chans = {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "half"), "B": (rgb[..., 2], "half")}
    path = str(tmp_path / "t.exr")
    X.write_exr(path, chans, compression, tiles=(32, 16), decreasing_y=True)
    (w, h), got = Connection.load_image(path)
    assert (w, h) == (70, 45)
    assert np.array_equal(got.view(np.uint32), _expected(rgb, "half").view(np.uint32))
    X.write_exr(path, chans, compression, tiles=(128, 64))
    _, got = Connection.load_image(path)
    assert np.array_equal(got.view(np.uint32), _expected(rgb, "half").view(np.uint32))
    X.write_exr(path, chans, compression, tiles=(16, 16), data_origin=(-6, 4), display_window=(0, 0, 59, 59))
    (w, h), got = Connection.load_image(path)
    assert (w, h) == (60, 60)
    want = np.zeros((60, 60, 4), np.float32)
    want[4:49, 0:60] = _expected(rgb, "half")[:, 6:66]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_tiled_mipmap_reads_the_full_resolution_level(tmp_path):
    """The reference asks its reader for `largest_resolution_level`: of a MIPMAP file level 0, whose tiles lead the
    offset table; the reduced levels behind them are never touched."""
    rgb = _image(37, 53, 32)
    chans = {"R": (rgb[..., 0], "float"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "float")}
    path = str(tmp_path / "m.exr")
    X.write_exr(path, chans, X.ZIP, tiles=(16, 8), mipmap=True)
    (w, h), got = Connection.load_image(path)
    assert (w, h) == (53, 37) and np.array_equal(got[..., :3], rgb)
    # a table whose first entry names a reduced-level tile is refused, not mistaken for pixels
    data = bytearray(open(path, "rb").read())
    import struct
    table = data.index(b"tiledesc\0") + len(b"tiledesc\0") + 4 + 9 + 1
    n_level0 = 4 * 5
    offs = list(struct.unpack_from("<%dQ" % (n_level0 + 1), data, table))
    struct.pack_into("<Q", data, table, offs[n_level0])
    open(path, "wb").write(bytes(data))
    with pytest.raises(N.RtError, match="reduced level"):
        Connection.load_image(path)


def test_multipart_files_first_part_with_rgb(tmp_path):
    """The reference's reader takes `first_valid_layer`: the first part that is not deep and has R, G and B."""
    rgb, other = _image(20, 30, 41), _image(24, 24, 42)
    beauty = {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "half"), "B": (rgb[..., 2], "half")}
    later = {"R": (other[..., 0], "half"), "G": (other[..., 1], "half"), "B": (other[..., 2], "half")}
    path = str(tmp_path / "mp.exr")
    X.write_exr_multipart(path, [
        dict(channels={"Z": (rgb[..., 0], "float")}, compression=X.ZIPS, name="depth"),
        dict(channels=later, compression=X.ZIPS, name="deep", type="deepscanline"),
        dict(channels=beauty, compression=X.PIZ, tiles=(16, 16), name="beauty"),
        dict(channels=later, compression=X.ZIP, name="later")])
    (w, h), got = Connection.load_image(path)
    assert (w, h) == (30, 20)
    assert np.array_equal(got.view(np.uint32), _expected(rgb, "half").view(np.uint32))
    X.write_exr_multipart(path, [dict(channels=later, compression=X.PXR24, name="only", decreasing_y=True)])
    (w, h), got = Connection.load_image(path)
    assert (w, h) == (24, 24) and np.array_equal(got.view(np.uint32), _expected(other, "half").view(np.uint32))
    X.write_exr_multipart(path, [dict(channels={"Z": (rgb[..., 0], "float")}, compression=X.ZIPS, name="depth"),
                                 dict(channels=later, compression=X.ZIPS, name="deep", type="deeptile")])
    with pytest.raises(N.RtError, match="no part with R, G, B"):
        Connection.load_image(path)


def test_corrupted_files_are_rejected_or_read_never_crash(tmp_path):
    """Byte flips and truncations of valid files (every compression, tiled with reduced levels, multi-part): the reader returns pixels or an error.
    Runs in a child process so that a memory fault would fail this test instead of killing the test run."""
    import subprocess
    import sys
    rgb = _image(40, 36, 4)
    paths = []
    chans = {"R": (rgb[..., 0], "half"), "G": (rgb[..., 1], "float"), "B": (rgb[..., 2], "half")}
    for comp in (X.NONE, X.RLE, X.ZIPS, X.ZIP, X.PIZ):
        p = str(tmp_path / ("h%d.exr" % comp))
        X.write_exr(p, chans, comp)
        paths.append(p)
    paths.append(str(tmp_path / "h_pxr24.exr"))
    X.write_exr(paths[-1], dict(chans, A=(rgb[..., 0].astype(np.uint32), "uint")), X.PXR24)
    paths.append(str(tmp_path / "h_tiled.exr"))
    X.write_exr(paths[-1], chans, X.PIZ, tiles=(16, 12), mipmap=True)
    paths.append(str(tmp_path / "h_parts.exr"))
    X.write_exr_multipart(paths[-1], [dict(channels={"Z": (rgb[..., 0], "float")}, compression=X.ZIPS, name="depth"),
                                      dict(channels=chans, compression=X.PXR24, tiles=(20, 20), name="beauty")])
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from rsraytracingv3_amd import Connection
from rsraytracingv3_amd import _native as N
rs = np.random.RandomState(11)
done = 0
for path in %r:
    good = bytearray(open(path, 'rb').read())
    for trial in range(120):
        data = bytearray(good)
        if trial %% 3 == 0:
            data = data[:rs.randint(1, len(data))]
        else:
            for _ in range(rs.randint(1, 6)):
                data[rs.randint(0, len(data))] = rs.randint(0, 256)
        open(path + '.bad', 'wb').write(bytes(data))
        try:
            Connection.load_image(path + '.bad')
        except N.RtError:
            pass
        done += 1
print('survived', done)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), paths)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "survived 960" in out.stdout


def _raw_header(channels, data_window, compression):
    """A hand-built single-part scan-line header (no writer involved): magic, version 2, attributes."""
    import struct

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iIii", t, 0, 1, 1) for n, t in channels) + b"\0"
    box = struct.pack("<4i", *data_window)
    return (struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chlist) +
            attr("compression", "compression", bytes([compression])) + attr("dataWindow", "box2i", box) +
            attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") +
            attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
            attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) +
            attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")


def test_hostile_headers_fail_with_a_status_not_an_abort(tmp_path):
    """A small file whose header declares 200 000 FLOAT channels over a 65536-pixel line (52 GB per scan line)
    used to end the process in std::bad_alloc / the OOM killer.  Headers with more than 64 channels, and
    blocks above 1 GiB, are rejected before anything is allocated; no exception leaves the C-ABI.  Child
    process with an address-space limit, so that a regression fails this test instead of the box."""
    import subprocess
    import sys
    many = [("c%06d" % i, 2) for i in range(200000)] + [("R", 2), ("G", 2), ("B", 2)]
    p1 = str(tmp_path / "many_channels.exr")
    open(p1, "wb").write(_raw_header(many, (0, 0, 65535, 0), X.NONE) + b"\0" * 64)
    wide = [("c%02d" % i, 2) for i in range(61)] + [("R", 2), ("G", 2), ("B", 2)]  # 64 channels: passes the count
    p2 = str(tmp_path / "huge_block.exr")  # 64 ch x 65536 px x 4 B x 32 lines (PIZ) = 512 MiB x 2 buffers: allowed
    open(p2, "wb").write(_raw_header(wide, (0, 0, 65535, 65535), X.PIZ) + b"\0" * 64)
    p3 = str(tmp_path / "sixty_five.exr")
    open(p3, "wb").write(_raw_header([("c%02d" % i, 1) for i in range(62)] + [("R", 1), ("G", 1), ("B", 1)],
                                     (0, 0, 7, 7), X.NONE) + b"\0" * 4096)
    code = r"""
import os, resource, sys
if os.environ.get("RT355_HOST_ONLY") != "1":  # (AddressSanitizer reserves terabytes of address space: tests/test_sanitized_host.py)
    resource.setrlimit(resource.RLIMIT_AS, (6 << 30, 6 << 30))
sys.path.insert(0, %r)
from rsraytracingv3_amd import Connection
from rsraytracingv3_amd import _native as N
for path in %r:
    try:
        Connection.load_image(path)
        print('read', path)
    except N.RtError as e:
        print('rejected', e.code, str(e)[-60:])
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), [p1, p2, p3])
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 3 and all(l.startswith("rejected") for l in lines), out.stdout
    assert "more than 64 channels" in lines[0] and "more than 64 channels" in lines[2]
