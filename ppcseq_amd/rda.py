"""Minimal reader for R ``save()`` files (RDX2/RDX3, XDR serialisation).

Purpose: ingest ppcseq's bundled ``data/counts.rda`` (reference: man/counts.Rd:8,
README.md:32-45) without an R installation, so the `identify_outliers` mirror can
be driven with the reference's own example data. Only the SEXP types that occur
in data frames are decoded (pairlists, symbols, character / integer / logical /
real vectors, generic vectors, with attributes); everything else raises.

This is host-side data ingestion, not part of the GPU hot path.
"""
from __future__ import annotations

import bz2
import gzip
import lzma
import struct
from typing import Any, Dict, List

import numpy as np

NILVALUE_SXP = 254
GLOBALENV_SXP = 253
EMPTYENV_SXP = 242
BASEENV_SXP = 241
REFSXP = 255
ALTREP_SXP = 238

SYMSXP, LISTSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 1, 2, 9, 10, 13, 14, 16, 19
LANGSXP = 6

NA_INTEGER = -2147483648


class RObject:
    """A decoded R vector with its attributes."""

    __slots__ = ("value", "attrs")

    def __init__(self, value: Any, attrs: Dict[str, Any] | None = None):
        self.value = value
        self.attrs = attrs or {}

    def __repr__(self) -> str:  # pragma: no cover - debugging aid
        return f"RObject({type(self.value).__name__}, attrs={list(self.attrs)})"


class _Reader:
    def __init__(self, buf: bytes):
        self.buf = buf
        self.pos = 0
        self.refs: List[Any] = []

    def i32(self) -> int:
        v = struct.unpack_from(">i", self.buf, self.pos)[0]
        self.pos += 4
        return v

    def raw(self, n: int) -> bytes:
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b

    def length(self) -> int:
        n = self.i32()
        if n == -1:  # long vector
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) + (lo & 0xFFFFFFFF)
        return n

    def item(self) -> Any:
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if t == NILVALUE_SXP:
            return None
        if t in (GLOBALENV_SXP, EMPTYENV_SXP, BASEENV_SXP):
            return None
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == SYMSXP:
            name = self.item()
            self.refs.append(name)
            return name
        if t in (LISTSXP, LANGSXP):
            # pairlist -> ordered list of (tag, value)
            out = []
            while True:
                attrs = self.item() if has_attr else None  # noqa: F841
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                nflags = self.i32()
                nt = nflags & 0xFF
                if nt == NILVALUE_SXP:
                    break
                if nt not in (LISTSXP, LANGSXP):
                    raise ValueError(f"unexpected CDR type {nt} in pairlist")
                has_attr = bool(nflags & (1 << 9))
                has_tag = bool(nflags & (1 << 10))
            return out
        if t == CHARSXP:
            n = self.i32()
            if n == -1:
                return None  # NA_character_
            return self.raw(n).decode("utf-8", errors="replace")
        if t == ALTREP_SXP:
            raise ValueError("ALTREP objects are not supported; re-save with version=2")
        if t in (LGLSXP, INTSXP):
            n = self.length()
            arr = np.frombuffer(self.buf, dtype=">i4", count=n, offset=self.pos).astype(np.int32)
            self.pos += 4 * n
            val: Any = arr
        elif t == REALSXP:
            n = self.length()
            arr = np.frombuffer(self.buf, dtype=">f8", count=n, offset=self.pos).astype(np.float64)
            self.pos += 8 * n
            val = arr
        elif t == STRSXP:
            n = self.length()
            val = [self.item() for _ in range(n)]
        elif t == VECSXP:
            n = self.length()
            val = [self.item() for _ in range(n)]
        else:
            raise ValueError(f"unsupported SEXP type {t} at offset {self.pos}")
        attrs = {}
        if has_attr:
            pl = self.item()
            attrs = {k: v for k, v in (pl or [])}
        return RObject(val, attrs)


def _decompress(raw: bytes) -> bytes:
    if raw[:3] == b"BZh":
        return bz2.decompress(raw)
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    if raw[:6] == b"\xfd7zXZ\x00":
        return lzma.decompress(raw)
    return raw


def read_rda(path: str) -> Dict[str, Any]:
    """Return ``{object_name: decoded object}`` for an R ``save()`` file."""
    with open(path, "rb") as fh:
        buf = _decompress(fh.read())
    if buf[:5] not in (b"RDX2\n", b"RDX3\n"):
        raise ValueError("not an RDX2/RDX3 file")
    r = _Reader(buf)
    r.pos = 5
    fmt = r.raw(2)
    if fmt != b"X\n":
        raise ValueError("only XDR serialisation is supported")
    version = r.i32()
    r.i32()  # writer R version
    r.i32()  # min reader version
    if version == 3:
        n = r.i32()
        r.raw(n)  # native encoding
    top = r.item()
    return {k: v for k, v in top}


def _factor_or_plain(col: RObject):
    levels = col.attrs.get("levels")
    if levels is not None and isinstance(col.value, np.ndarray):
        lv = np.array(levels.value, dtype=object)
        return lv[col.value - 1]
    if isinstance(col.value, list):
        return np.array(col.value, dtype=object)
    return col.value


def data_frame_columns(obj: RObject) -> Dict[str, np.ndarray]:
    """Columns of a decoded data.frame/tibble as ``{name: ndarray}`` (factors expanded)."""
    names = obj.attrs["names"].value
    return {n: _factor_or_plain(c) for n, c in zip(names, obj.value)}
