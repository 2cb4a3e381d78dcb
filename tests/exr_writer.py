"""A small OpenEXR WRITER (scan-line, tiled, multi-part), test infrastructure for the reader in rsraytracingv3_amd/csrc/rt_exr.cpp.

Written separately from the reader (Python, encoder side of every step) after the published OpenEXR file
layout: magic/version, attribute list, chunk offset table, chunks of 1 (NONE, RLE, ZIPS), 16 (ZIP, PXR24) or 32
(PIZ) scan lines, or tiles.  No file made by another EXR implementation exists in this environment, so reader and
writer pin each other only — "parity unpinned" for real-world files."""
import heapq
import struct
import zlib

import numpy as np

NONE, RLE, ZIPS, ZIP, PIZ, PXR24 = 0, 1, 2, 3, 4, 5
LINES = {NONE: 1, RLE: 1, ZIPS: 1, ZIP: 16, PIZ: 32, PXR24: 16}
TYPE_CODE = {"uint": 0, "half": 1, "float": 2}
NP_TYPE = {"uint": np.uint32, "half": np.float16, "float": np.float32}


def _attr(name, typ, value):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value


def _predict_and_split(raw: bytes) -> bytes:
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
    return d.astype(np.uint8).tobytes()


def _rle(data: bytes) -> bytes:
    """Runs of 3..128 equal bytes -> (count-1, byte); everything else -> (-count, literals)."""
    out = bytearray()
    n, i = len(data), 0
    while i < n:
        j = i + 1
        while j < n and data[j] == data[i] and j - i < 128:
            j += 1
        if j - i >= 3:
            out += struct.pack("b", j - i - 1) + data[i:i + 1]
            i = j
            continue
        j = i
        while j < n and j - i < 127:
            if j + 2 < n and data[j] == data[j + 1] == data[j + 2]:
                break
            j += 1
        out += struct.pack("b", -(j - i)) + data[i:j]
        i = j
    return bytes(out)


# ---- PIZ ------------------------------------------------------------------------------------------
def _wenc14(a, b):
    a = a - 65536 if a >= 32768 else a
    b = b - 65536 if b >= 32768 else b
    return ((a + b) >> 1) & 0xffff, (a - b) & 0xffff


def _wenc16(a, b):
    ao = (a + 32768) & 0xffff
    m = (ao + b) >> 1
    d = ao - b
    if d < 0:
        m = (m + 32768) & 0xffff
    return m, d & 0xffff


def _wav2_encode(buf, start, nx, ox, ny, oy, mx):
    enc = _wenc14 if mx < (1 << 14) else _wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py, ey = start, start + oy * (ny - p2)
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i01 = enc(buf[px], buf[p01])
                i10, i11 = enc(buf[p10], buf[p11])
                buf[px], buf[p10] = enc(i00, i10)
                buf[p01], buf[p11] = enc(i01, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                buf[px], buf[p10] = enc(buf[px], buf[p10])
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                buf[px], buf[p01] = enc(buf[px], buf[p01])
                px += ox2
        p, p2 = p2, p2 << 1


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, nbits, value):
        self.acc = (self.acc << nbits) | value
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xff)
        self.acc &= (1 << self.n) - 1

    def finish(self):
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 0xff)
            self.acc, self.n = 0, 0
        return bytes(self.out)


def _huf_compress(symbols) -> bytes:
    if len(symbols) == 0:
        return b""
    freq = {}
    for s in symbols:
        freq[s] = freq.get(s, 0) + 1
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1  # the run-length escape
    # plain Huffman code lengths
    heap = [(f, s, (s,)) for s, f in freq.items()]
    heapq.heapify(heap)
    length = {s: 0 for s in freq}
    tie = 1 << 20
    while len(heap) > 1:
        f1, _, m1 = heapq.heappop(heap)
        f2, _, m2 = heapq.heappop(heap)
        for s in m1 + m2:
            length[s] += 1
        tie += 1
        heapq.heappush(heap, (f1 + f2, tie, m1 + m2))
    assert max(length.values()) <= 58
    # canonical codes
    count = [0] * 59
    for ln in length.values():
        count[ln] += 1
    count[0] = 0
    c, first = 0, [0] * 59
    for i in range(58, 0, -1):
        first[i] = c
        c = (c + count[i]) >> 1
    code = {}
    for s in sorted(length):
        code[s] = first[length[s]]
        first[length[s]] += 1
    # packed table
    tb = _Bits()
    s = im
    while s <= iM:
        ln = length.get(s, 0)
        if ln == 0:
            zerun = 1
            while s < iM and zerun < 255 + 6 and length.get(s + 1, 0) == 0:
                s += 1
                zerun += 1
            if zerun >= 2:
                if zerun >= 6:
                    tb.put(6, 63)
                    tb.put(8, zerun - 6)
                else:
                    tb.put(6, 59 + zerun - 2)
                s += 1
                continue
        tb.put(6, ln)
        s += 1
    table = tb.finish()
    # code stream with run-length escapes
    db = _Bits()
    nbits = 0

    def send(sym, run):
        nonlocal nbits
        ls, lr = length[sym], length[iM]
        if ls + lr + 8 < ls * run:
            db.put(ls, code[sym])
            db.put(lr, code[iM])
            db.put(8, run)
            nbits += ls + lr + 8
        else:
            for _ in range(run + 1):
                db.put(ls, code[sym])
            nbits += ls * (run + 1)

    prev, run = symbols[0], 0
    for s in symbols[1:]:
        if s == prev and run < 255:
            run += 1
            continue
        send(prev, run)
        prev, run = s, 0
    send(prev, run)
    data = db.finish()
    return struct.pack("<IIIII", im, iM, len(table), nbits, 0) + table + data


def _piz(planes, width, lines) -> bytes:
    """planes: per channel (array lines x width, words per sample)."""
    parts, shapes = [], []
    for arr, size in planes:
        parts.append(np.ascontiguousarray(arr).view(np.uint16).reshape(-1))
        shapes.append(size)
    buf = [int(v) for v in np.concatenate(parts)]
    present = set(buf)
    present.discard(0)
    bitmap = bytearray(8192)
    for v in present:
        bitmap[v >> 3] |= 1 << (v & 7)
    nz = [i for i, b in enumerate(bitmap) if b]
    min_nz, max_nz = (nz[0], nz[-1]) if nz else (8191, 0)
    lut, k = {}, 0
    for v in sorted(present | {0}):
        lut[v] = k
        k += 1
    max_value = k - 1
    buf = [lut[v] for v in buf]
    start = 0
    for size in shapes:
        for j in range(size):
            _wav2_encode(buf, start + j, width, size, lines, width * size, max_value)
        start += width * lines * size
    huf = _huf_compress(buf)
    head = struct.pack("<HH", min_nz, max_nz)
    if min_nz <= max_nz:
        head += bytes(bitmap[min_nz:max_nz + 1])
    return head + struct.pack("<i", len(huf)) + huf


# ---- PXR24 ------------------------------------------------------------------------------------------
def float_to_float24(bits):
    """The upper 24 bits of a float, rounded to nearest (infinities kept, NaNs kept NaN, no rounding into infinity)."""
    bits = np.asarray(bits, np.uint32).astype(np.int64)
    s, e, m = bits & 0x80000000, bits & 0x7f800000, bits & 0x007fffff
    rounded = ((e | m) + (m & 0x80)) >> 8
    rounded = np.where(rounded >= 0x7f8000, (e | m) >> 8, rounded)
    nan_m = m >> 8
    special = np.where(m != 0, (e >> 8) | nan_m | (nan_m == 0), e >> 8)
    return ((s >> 8) | np.where(e == 0x7f800000, special, rounded)).astype(np.uint32)


def _pxr24(planes, width, lines) -> bytes:
    out = []
    for y in range(lines):
        for arr, _ in planes:
            row = arr[y]
            if row.dtype == np.float16:
                v, nbytes = row.view(np.uint16).astype(np.int64), 2
            elif row.dtype == np.float32:
                v, nbytes = float_to_float24(row.view(np.uint32)).astype(np.int64), 3
            else:
                v, nbytes = row.astype(np.int64), 4
            d = v.copy()
            d[1:] = v[1:] - v[:-1]
            d &= (1 << (8 * nbytes)) - 1
            for k in range(nbytes - 1, -1, -1):  # most significant plane first
                out.append(((d >> (8 * k)) & 255).astype(np.uint8).tobytes())
    return zlib.compress(b"".join(out))


# ---- file --------------------------------------------------------------------------------------------
def _encode_chunk(job):
    """One chunk's payload (module level so that a process pool can run it)."""
    compression, force_raw, w, lines, planes = job
    raw = b"".join(arr[y].tobytes() for y in range(lines) for arr, _ in planes)
    if compression == NONE or force_raw:
        data = raw
    elif compression == RLE:
        data = _rle(_predict_and_split(raw))
    elif compression in (ZIPS, ZIP):
        data = zlib.compress(_predict_and_split(raw))
    elif compression == PXR24:
        data = _pxr24(planes, w, lines)
    else:
        data = _piz(planes, w, lines)
    return raw if len(data) >= len(raw) else data


def _build_part(channels, compression=ZIP, display_window=None, data_origin=(0, 0), decreasing_y=False,
                force_raw_chunks=False, workers=1, tiles=None, mipmap=False, extra_attrs=b""):
    """-> (attribute bytes without the terminating null, chunks in table order, file order of the chunks).
    A chunk is its coordinates (y, or tile x, tile y, level x, level y) + payload size + payload."""
    names = sorted(channels)
    h, w = channels[names[0]][0].shape
    x0, y0 = data_origin
    data_win = (x0, y0, x0 + w - 1, y0 + h - 1)
    disp = display_window or data_win
    chlist = b""
    arrays = []
    for n in names:
        arr, typ = channels[n]
        assert arr.shape == (h, w)
        chlist += n.encode() + b"\0" + struct.pack("<iB3xii", TYPE_CODE[typ], 0, 1, 1)
        arrays.append((np.ascontiguousarray(arr.astype(NP_TYPE[typ])), 1 if typ == "half" else 2))
    chlist += b"\0"
    header = _attr("channels", "chlist", chlist)
    header += _attr("compression", "compression", struct.pack("B", compression))
    header += _attr("dataWindow", "box2i", struct.pack("<4i", *data_win))
    header += _attr("displayWindow", "box2i", struct.pack("<4i", *disp))
    header += _attr("lineOrder", "lineOrder", struct.pack("B", 1 if decreasing_y else 0))
    header += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    header += _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    header += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    if tiles:
        header += _attr("tiles", "tiledesc", struct.pack("<IIB", tiles[0], tiles[1], 1 if mipmap else 0))
    header += extra_attrs
    jobs, coords = [], []
    if tiles:
        tx, ty = tiles
        levels = [arrays]
        while mipmap and max(levels[-1][0][0].shape) > 1:  # ROUND_DOWN: each level halves, never below one
            levels.append([(np.ascontiguousarray(a[::2, ::2][:max(a.shape[0] // 2, 1), :max(a.shape[1] // 2, 1)]), s)
                           for a, s in levels[-1]])
        for lv, arrs in enumerate(levels):
            lh, lw = arrs[0][0].shape
            for j in range(0, lh, ty):
                for i in range(0, lw, tx):
                    tw, th = min(tx, lw - i), min(ty, lh - j)
                    jobs.append((compression, force_raw_chunks, tw, th,
                                 [(np.ascontiguousarray(a[j:j + th, i:i + tw]), s) for a, s in arrs]))
                    coords.append(struct.pack("<4i", i // tx, j // ty, lv, lv))
    else:
        step = LINES[compression]
        for r0 in range(0, h, step):
            n = min(r0 + step, h) - r0
            jobs.append((compression, force_raw_chunks, w, n, [(arr[r0:r0 + n], size) for arr, size in arrays]))
            coords.append(struct.pack("<i", y0 + r0))
    if workers > 1:
        import multiprocessing
        with multiprocessing.get_context("fork").Pool(workers) as pool:
            payloads = pool.map(_encode_chunk, jobs, chunksize=1)
    else:
        payloads = [_encode_chunk(j) for j in jobs]
    chunks = [c + struct.pack("<i", len(data)) + data for c, data in zip(coords, payloads)]
    order = list(range(len(chunks) - 1, -1, -1)) if decreasing_y else list(range(len(chunks)))
    return header, chunks, order


def write_exr(path, channels, compression=ZIP, display_window=None, data_origin=(0, 0), decreasing_y=False,
              force_raw_chunks=False, workers=1, tiles=None, mipmap=False):
    """channels: {name: (2-D array, "half" | "float" | "uint")}, all of one shape (the data window).
    display_window: (xmin, ymin, xmax, ymax), default = the data window placed at data_origin.
    tiles = (width, height) writes a tiled file; mipmap adds the reduced levels after the full-resolution one.
    workers > 1 encodes the chunks in a process pool (the pure-Python PIZ encoder needs it at 8k)."""
    attrs, chunks, order = _build_part(channels, compression, display_window, data_origin, decreasing_y,
                                       force_raw_chunks, workers, tiles, mipmap)
    header = struct.pack("<II", 20000630, 2 | (0x200 if tiles else 0)) + attrs + b"\0"
    table_at = len(header)
    offsets, pos = [0] * len(chunks), table_at + 8 * len(chunks)
    body = b""
    for i in order:
        offsets[i] = pos + len(body)
        body += chunks[i]
    with open(path, "wb") as f:
        f.write(header + struct.pack("<%dQ" % len(chunks), *offsets) + body)


def write_exr_multipart(path, parts):
    """parts: list of dicts — the keyword arguments of write_exr (channels, compression, tiles, ...) plus "name" and
    optionally "type" (default by tiles; "deepscanline" writes a part whose chunks are opaque bytes a reader must skip)."""
    built = []
    for k, p in enumerate(parts):
        p = dict(p)
        name = p.pop("name", "part%d" % k)
        typ = p.pop("type", "tiledimage" if p.get("tiles") else "scanlineimage")
        attrs, chunks, order = _build_part(**p)
        if typ.startswith("deep"):
            chunks = [struct.pack("<i", 0) + b"\xde\xad" * 20 for _ in chunks]
        attrs += _attr("name", "string", name.encode()) + _attr("type", "string", typ.encode())
        attrs += _attr("chunkCount", "int", struct.pack("<i", len(chunks)))
        built.append((attrs, chunks, order))
    deep = any(p.get("type", "").startswith("deep") for p in parts)
    header = struct.pack("<II", 20000630, 2 | 0x1000 | (0x800 if deep else 0))
    header += b"".join(attrs + b"\0" for attrs, _, _ in built) + b"\0"
    pos = len(header) + 8 * sum(len(chunks) for _, chunks, _ in built)
    tables, body = b"", b""
    for k, (_, chunks, order) in enumerate(built):
        offsets = [0] * len(chunks)
        for i in order:
            offsets[i] = pos + len(body)
            body += struct.pack("<i", k) + chunks[i]
        tables += struct.pack("<%dQ" % len(chunks), *offsets)
    with open(path, "wb") as f:
        f.write(header + tables + body)


if __name__ == "__main__":
    # python tests/exr_writer.py skybox WIDTH PATH [piz|zip|zips|rle|none] [workers]
    # Writes rsraytracingv3_amd.synth.procedural_skybox(WIDTH, WIDTH/2) as a HALF R,G,B file.  Run as a child
    # process by the config-5 GPU test: this process never touches the GPU, so its fork pool is safe.
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from rsraytracingv3_amd import synth
    if len(sys.argv) < 4 or sys.argv[1] != "skybox":
        raise SystemExit(__doc__)
    width, out_path = int(sys.argv[2]), sys.argv[3]
    comp = {"piz": PIZ, "zip": ZIP, "zips": ZIPS, "rle": RLE, "none": NONE, "pxr24": PXR24}[sys.argv[4] if len(sys.argv) > 4 else "piz"]
    n_workers = int(sys.argv[5]) if len(sys.argv) > 5 else (os.cpu_count() or 1)
    img = synth.procedural_skybox(width, width // 2)
    write_exr(out_path, {"R": (img[..., 0], "half"), "G": (img[..., 1], "half"), "B": (img[..., 2], "half")}, comp,
              workers=n_workers)
    print("wrote %s: %dx%d, %d bytes" % (out_path, width, width // 2, os.path.getsize(out_path)))
