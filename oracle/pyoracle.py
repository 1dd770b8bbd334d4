"""ctypes wrapper of oracle/libfqs_oracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (fqsqueezer_amd) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfqs_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fqs_oracle.cpp")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "restate"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.fqo_create.restype = C.c_void_p
        _lib.fqo_create.argtypes = [C.c_char_p]
        _lib.fqo_destroy.argtypes = [C.c_void_p]
        _lib.fqo_encode_block.restype = C.c_int
        _lib.fqo_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        _lib.fqo_stream.restype = C.POINTER(C.c_uint8)
        _lib.fqo_stream.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
        _lib.fqo_decode_block.restype = C.c_int
        _lib.fqo_decode_block.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.fqo_qual_create.restype = C.c_void_p
        _lib.fqo_qual_create.argtypes = [C.c_char_p]
        _lib.fqo_qual_destroy.argtypes = [C.c_void_p]
        _lib.fqo_qual_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _lib.fqo_qual_stream.restype = C.POINTER(C.c_uint8)
        _lib.fqo_qual_stream.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
        _lib.fqo_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _lib.fqo_levels.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _lib.fqo_kat_mt19937.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.fqo_kat_cinc.argtypes = [C.c_uint32] * 4 + [C.c_void_p] * 3
        _lib.fqo_kat_rc.restype = C.c_uint64
        _lib.fqo_kat_rc.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    return _lib


class OracleCodec:
    """Same call shape as fqsqueezer_amd.DnaCodec (create / encode_block / streams)."""

    def __init__(self, header: bytes):
        self.T = header[4]
        self._h = lib().fqo_create(bytes(header))
        if not self._h:
            raise ValueError("oracle: unsupported header")

    def encode_block(self, bases: np.ndarray, read_off: np.ndarray, generation: int):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        rc = lib().fqo_encode_block(self._h, bases.ctypes.data, read_off.ctypes.data, len(read_off) - 1, generation)
        if rc:
            raise RuntimeError("oracle encode failed")
        out = []
        n = C.c_uint64()
        for w in range(self.T):
            p = lib().fqo_stream(self._h, w, C.byref(n))
            out.append(bytes(C.cast(p, C.POINTER(C.c_uint8 * n.value)).contents) if n.value else b"")
        return out

    def decode_block(self, streams, read_off: np.ndarray, generation: int) -> np.ndarray:
        """Inverse of encode_block: T streams + read offsets -> concatenated base bytes."""
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        arr = (C.c_char_p * self.T)(*[bytes(x) for x in streams])
        lens = np.array([len(x) for x in streams], dtype=np.uint64)
        out = np.zeros(int(read_off[-1]), dtype=np.uint8)
        rc = lib().fqo_decode_block(self._h, arr, lens.ctypes.data, read_off.ctypes.data, len(read_off) - 1, generation, out.ctypes.data)
        if rc:
            raise RuntimeError("oracle decode failed")
        return out

    def counters(self):
        a = (C.c_uint64 * 8)()
        lib().fqo_counters(self._h, a)
        return dict(zip(["probes", "slots", "inserts", "siv_words", "ctx", "coded", "lprobes", "linserts"], list(a)))

    def levels(self):
        a = (C.c_uint64 * 10)()
        lib().fqo_levels(self._h, a)
        return dict(zip(["none", "pmer", "smer", "bmer", "mixed", "bmer_unc", "draws_b", "draws_s", "draws_lb", "draws_ls"], list(a)))

    def close(self):
        if self._h:
            lib().fqo_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class OracleQual:
    """Quality-stream restatement (same call shape as fqsqueezer_amd.codec.QualCodec)."""

    def __init__(self, header: bytes):
        self.T = header[4]
        self._h = lib().fqo_qual_create(bytes(header))
        if not self._h:
            raise ValueError("oracle: unsupported header")

    def encode_block(self, quals: np.ndarray, read_off: np.ndarray):
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        lib().fqo_qual_encode_block(self._h, quals.ctypes.data, read_off.ctypes.data, len(read_off) - 1)
        out, n = [], C.c_uint64()
        for w in range(self.T):
            p = lib().fqo_qual_stream(self._h, w, C.byref(n))
            out.append(bytes(C.cast(p, C.POINTER(C.c_uint8 * n.value)).contents) if n.value else b"")
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fqo_qual_destroy(self._h)
            self._h = None
