"""Minimal reader for R's XDR serialisation (`.RData` / `.rda`, RDX2 and RDX3).

Only what the reference's bundled fixtures need:
pairlists, symbols, character/integer/logical/real vectors, generic vectors,
S4 objects whose slots arrive as attributes (dgCMatrix), external pointers
(data.table's `.internal.selfref`), reference table look-ups and the
environment sentinels.  Everything is returned as plain Python objects:

    {"type": ..., "value": ..., "attr": {...}}

No R is required.  Used by `tools/make_golden.py`; not part of the product.
"""
import bz2
import gzip
import lzma
import struct

import numpy as np


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.p = 0
        self.refs = []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def raw(self, n):
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) + lo
        return n

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if t == 254:  # NILVALUE_SXP
            return None
        if t in (253, 242, 241, 250, 249, 251):  # global/base/empty env, missing, unbound
            return {"type": "env_sentinel", "value": t}
        if t == 255:  # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == 1:  # SYMSXP
            name = self.item()
            sym = {"type": "sym", "value": name["value"]}
            self.refs.append(sym)
            return sym
        if t in (2, 6):  # LISTSXP / LANGSXP
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag["value"] if tag else None, car))
                nflags = self.i32()
                nt = nflags & 0xFF
                if nt == 254:
                    break
                if nt not in (2, 6):
                    self.p -= 4
                    cdr = self.item()
                    out.append((None, cdr))
                    break
                has_attr = bool(nflags & (1 << 9))
                has_tag = bool(nflags & (1 << 10))
            return {"type": "pairlist", "value": out}
        if t == 9:  # CHARSXP
            n = self.i32()
            if n == -1:
                return {"type": "char", "value": None}
            return {"type": "char", "value": self.raw(n).decode("utf-8", "replace")}
        if t in (10, 13):  # LGLSXP / INTSXP
            n = self.length()
            v = np.frombuffer(self.raw(4 * n), dtype=">i4").astype(np.int32)
            return self._fin({"type": "int" if t == 13 else "lgl", "value": v}, has_attr)
        if t == 14:  # REALSXP
            n = self.length()
            v = np.frombuffer(self.raw(8 * n), dtype=">f8").astype(np.float64)
            return self._fin({"type": "real", "value": v}, has_attr)
        if t == 16:  # STRSXP
            n = self.length()
            v = [self.item()["value"] for _ in range(n)]
            return self._fin({"type": "str", "value": v}, has_attr)
        if t == 19:  # VECSXP
            n = self.length()
            v = [self.item() for _ in range(n)]
            return self._fin({"type": "list", "value": v}, has_attr)
        if t == 22:  # EXTPTRSXP
            obj = {"type": "extptr", "value": None}
            self.refs.append(obj)
            self.item()  # prot
            self.item()  # tag
            return self._fin(obj, has_attr)
        if t == 25:  # S4SXP
            return self._fin({"type": "S4", "value": None}, has_attr)
        if t == 238:  # ALTREP
            info = self.item()
            state = self.item()
            self.item()  # attributes
            cls = info["value"][0][1]["value"]
            if cls == "compact_intseq":
                n, start, step = (int(x) for x in state["value"])
                return {"type": "int", "value": (start + step * np.arange(n)).astype(np.int32), "attr": {}}
            if cls == "wrap_integer" or cls == "wrap_real" or cls == "wrap_string":
                return state["value"][0][1]
            raise NotImplementedError("ALTREP class %s" % cls)
        raise NotImplementedError("SEXP type %d at byte %d" % (t, self.p))

    def _fin(self, obj, has_attr):
        obj["attr"] = {}
        if has_attr:
            a = self.item()
            if a is not None:
                obj["attr"] = {k: v for k, v in a["value"]}
        return obj


def read_rdata(path):
    """Return {name: object} for every top-level binding in an .RData/.rda file."""
    with open(path, "rb") as f:
        head = f.read(6)
    if head[:2] == b"\x1f\x8b":
        buf = gzip.open(path).read()
    elif head[:3] == b"BZh":
        buf = bz2.open(path).read()
    elif head[:6] == b"\xfd7zXZ\x00":
        buf = lzma.open(path).read()
    else:
        buf = open(path, "rb").read()
    magic = buf[:5]
    assert magic in (b"RDX2\n", b"RDX3\n"), magic
    r = _Reader(buf)
    r.p = 5
    assert r.raw(2) == b"X\n"
    version = r.i32()
    r.i32()
    r.i32()
    if version == 3:
        n = r.i32()
        r.raw(n)
    top = r.item()
    return {k: v for k, v in top["value"]}


def data_frame_columns(obj):
    """data.frame / data.table (a VECSXP with a `names` attribute) -> {col: ndarray | list}."""
    names = obj["attr"]["names"]["value"]
    out = {}
    for n, col in zip(names, obj["value"]):
        v = col["value"]
        if col["type"] == "int" and "levels" in col["attr"]:
            lev = col["attr"]["levels"]["value"]
            v = [lev[i - 1] for i in v]
        out[n] = v
    return out
