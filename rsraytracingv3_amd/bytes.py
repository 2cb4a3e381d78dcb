"""HLSL-layout byte packing — host-side mirror of the reference's ``src/bytes.rs``.

``bytes_concat`` (bytes.rs:31-52) joins variable-length pieces into an N-byte
array, truncating past N and zero-padding short input; ``bytes_concat_owned``
(bytes.rs:64-83) joins fixed-size M-byte records, keeping only the first N/M
(bytes.rs:75) and asserting N % M == 0 (bytes.rs:71).  These define the wire
format that crosses the C-ABI; the kernel side consumes it.
"""
from __future__ import annotations

import struct
from typing import Iterable, Sequence


def f32_bytes(*values: float) -> bytes:
    """Little-endian IEEE binary32 of each value (``to_le_bytes`` / the
    ``transmute`` of nalgebra storage, bytes.rs:13-23)."""
    return struct.pack("<%df" % len(values), *values)


def u32_bytes(value: int) -> bytes:
    return struct.pack("<I", value)


def bytes_concat(n: int, pieces: Iterable[bytes]) -> bytes:
    """bytes.rs:31-52."""
    out = bytearray(n)
    offset = 0
    for piece in pieces:
        length = min(len(piece), max(n - offset, 0))
        if length == 0:
            return bytes(out)
        out[offset:offset + length] = piece[:length]
        offset += length
    return bytes(out)


def bytes_concat_owned(m: int, n: int, records: Iterable[bytes]) -> bytes:
    """bytes.rs:64-83 — records beyond N/M are silently dropped (bytes.rs:75)."""
    if n % m != 0:
        raise AssertionError("N must be a multiple of M (bytes.rs:71)")
    out = bytearray(n)
    for slot, record in enumerate(records):
        if slot >= n // m:
            break
        if len(record) != m:
            raise ValueError("record of %d bytes, expected %d" % (len(record), m))
        out[slot * m:(slot + 1) * m] = record
    return bytes(out)


def show_bytes(data: Sequence[int]) -> str:
    """bytes.rs:85-102 — hex dump helper, 4-byte groups, 16 per line."""
    s = []
    for i, byte in enumerate(data):
        if i % 4 == 0:
            s.append(" ")
        if i % 16 == 0:
            s.append("\n")
        s.append("%02x " % byte)
    return "".join(s)
