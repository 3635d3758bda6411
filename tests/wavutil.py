"""WAV writers for the command line tool's tests (test infrastructure)."""
import struct

import numpy as np


def write_wav(path, frames: np.ndarray, rate: int, kind: str = "pcm16", extensible: bool = False):
    """frames [n, channels] float in [-1, 1) (float kinds) or already-quantised integers (pcm kinds)."""
    a = np.asarray(frames)
    n, ch = a.shape
    if kind == "pcm16":
        fmt, bits, data = 1, 16, a.astype("<i2").tobytes()
    elif kind == "pcm8":
        fmt, bits, data = 1, 8, a.astype("u1").tobytes()
    elif kind == "pcm24":
        v = a.astype("<i4")
        b = np.stack([(v >> 0) & 255, (v >> 8) & 255, (v >> 16) & 255], axis=-1).astype("u1")
        fmt, bits, data = 1, 24, b.tobytes()
    elif kind == "pcm32":
        fmt, bits, data = 1, 32, a.astype("<i4").tobytes()
    elif kind == "float32":
        fmt, bits, data = 3, 32, a.astype("<f4").tobytes()
    elif kind == "float64":
        fmt, bits, data = 3, 64, a.astype("<f8").tobytes()
    else:
        raise ValueError(kind)
    align = ch * bits // 8
    if extensible:
        guid = struct.pack("<H", fmt) + bytes.fromhex("000000001000800000aa00389b71")
        fmt_chunk = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * align, align, bits, 22, bits, 0) + guid
    else:
        fmt_chunk = struct.pack("<HHIIHH", fmt, ch, rate, rate * align, align, bits)
    junk = b"LIST" + struct.pack("<I", 5) + b"hello" + b"\0"          # an odd-sized chunk before fmt: readers must skip + pad
    body = b"WAVE" + junk + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk + b"data" + struct.pack("<I", len(data)) + data
    if len(data) & 1:
        body += b"\0"
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def to_float(frames: np.ndarray, kind: str) -> np.ndarray:
    """What the tool's reader turns the stored samples into (fp32)."""
    a = np.asarray(frames)
    if kind == "pcm16":
        return (a.astype(np.float32) * np.float32(1.0 / 32768.0)).astype(np.float32)
    if kind == "pcm8":
        return ((a.astype(np.float32) - 128.0) * np.float32(1.0 / 128.0)).astype(np.float32)
    if kind == "pcm24":
        return (a.astype(np.float32) * np.float32(1.0 / 8388608.0)).astype(np.float32)
    if kind == "pcm32":
        return (a.astype(np.float64) / 2147483648.0).astype(np.float32)
    return a.astype(np.float32)
