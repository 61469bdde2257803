"""Scalar logging in the reference's style: ObjTracker/run.py:127 opens a tensorboardX SummaryWriter under
exps/<seq>/<exp>/board and jointopt.py:151-153 calls add_scalar(key, value, step) once per loss key per step.

make_writer(logdir) returns torch.utils.tensorboard's SummaryWriter when that package imports, and otherwise EventFileWriter --
a dependency-free writer of the same TensorBoard event-file format (TFRecord framing with masked CRC-32C, Event / Summary
protobuf messages restricted to simple_value scalars), so the Runner leaves `events.out.tfevents.*` files TensorBoard can open
on machines where neither tensorboard nor tensorboardX is installed (this build image is one).  read_scalars() parses such a
file back (tests)."""
from __future__ import annotations

import os
import socket
import struct
import time

_CRC_TABLE = None


def _crc32c(data: bytes) -> int:
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _len_field(tag: int, payload: bytes) -> bytes:
    return bytes([tag]) + _varint(len(payload)) + payload


def _event(wall_time: float, step: int, file_version: str | None = None, scalar=None) -> bytes:
    ev = b"\x09" + struct.pack("<d", wall_time) + b"\x10" + _varint(step)
    if file_version is not None:
        ev += _len_field(0x1A, file_version.encode())
    if scalar is not None:
        tag, value = scalar
        val = _len_field(0x0A, tag.encode()) + b"\x15" + struct.pack("<f", float(value))
        ev += _len_field(0x2A, _len_field(0x0A, val))
    return ev


class EventFileWriter:
    """add_scalar(tag, value, global_step) / flush() / close(): the part of SummaryWriter the reference uses."""

    def __init__(self, logdir: str):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s.%d" % (int(time.time()), socket.gethostname(), os.getpid()))
        self._f = open(self.path, "ab")
        self._record(_event(time.time(), 0, file_version="brain.Event:2"))

    def _record(self, data: bytes):
        head = struct.pack("<Q", len(data))
        self._f.write(head + struct.pack("<I", _masked_crc(head)) + data + struct.pack("<I", _masked_crc(data)))

    def add_scalar(self, tag: str, scalar_value, global_step: int = 0):
        self._record(_event(time.time(), int(global_step), scalar=(tag, scalar_value)))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def make_writer(logdir: str):
    """torch's SummaryWriter when the tensorboard package is importable, else the EventFileWriter above.  Only a failing IMPORT
    selects the fallback (a missing package, or a broken / incompatible tensorboard, protobuf or setuptools install -- those raise
    AttributeError / TypeError rather than ImportError); a bad logdir or a permission error surfaces from whichever writer is used."""
    try:
        from torch.utils.tensorboard import SummaryWriter       # needs the tensorboard package
    except Exception:                                           # noqa: BLE001 -- the import statement only
        return EventFileWriter(logdir)
    return SummaryWriter(logdir)


def _read_varint(buf: bytes, i: int):
    n, shift = 0, 0
    while True:
        b = buf[i]; i += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, i


def read_scalars(path: str):
    """[(step, tag, value)] of an event file; verifies both CRCs of every record."""
    out = []
    data = open(path, "rb").read()
    i = 0
    while i < len(data):
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        assert struct.unpack("<I", data[i + 8:i + 12])[0] == _masked_crc(head), "length CRC"
        rec = data[i + 12:i + 12 + n]
        assert struct.unpack("<I", data[i + 12 + n:i + 16 + n])[0] == _masked_crc(rec), "data CRC"
        i += 16 + n
        j, step, summ = 0, 0, None
        while j < len(rec):
            tag = rec[j]; j += 1
            if tag == 0x09:
                j += 8
            elif tag == 0x10:
                step, j = _read_varint(rec, j)
            else:
                ln, j = _read_varint(rec, j)
                if tag == 0x2A:
                    summ = rec[j:j + ln]
                j += ln
        if summ is not None:
            assert summ[0] == 0x0A
            ln, k = _read_varint(summ, 1)
            val = summ[k:k + ln]
            assert val[0] == 0x0A
            tl, k2 = _read_varint(val, 1)
            name = val[k2:k2 + tl].decode()
            assert val[k2 + tl] == 0x15
            out.append((step, name, struct.unpack("<f", val[k2 + tl + 1:k2 + tl + 5])[0]))
    return out
