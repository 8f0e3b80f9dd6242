"""Minimal reader for R workspace files (``save(...)``; RDX2/RDX3, XDR encoding).

The reference ships its synthetic inputs as ``data/*.RData`` (written by
generate_data.R:229-241) and the Senate cube as ``data/senate_data_90.RData``
(senatedata.R:71).  This reader decodes numeric / integer / logical / string
vectors with their ``dim`` attribute, generic lists and ALTREP compact sequences,
which covers every object in those files.  It only parses bytes: nothing in the file
is evaluated.
"""
import gzip
import struct

import numpy as np

NA_INT = -2147483648


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.p = 0
        self.refs = []

    def take(self, k):
        v = self.b[self.p:self.p + k]
        if len(v) != k:
            raise ValueError("truncated RData stream")
        self.p += k
        return v

    def i32(self):
        return struct.unpack(">i", self.take(4))[0]

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            return (hi << 32) + (lo & 0xFFFFFFFF)
        return n

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if t == 254:  # NILVALUE
            return None
        if t in (253, 242, 241, 250, 251):  # global/empty/base env, missing arg, unbound
            return None
        if t == 255:  # reference
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == 1:  # symbol
            name = self.item()
            self.refs.append(name)
            return name
        if t in (2, 6):  # pairlist / language
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                del attr
                flags = self.i32()
                t2 = flags & 0xFF
                if t2 == 254:
                    break
                if t2 not in (2, 6):
                    raise ValueError("unexpected cdr type %d" % t2)
                has_attr = bool(flags & (1 << 9))
                has_tag = bool(flags & (1 << 10))
            return out
        if t == 9:  # CHARSXP
            n = self.i32()
            if n == -1:
                return None
            return self.take(n).decode("utf-8", "replace")
        if t == 238:  # ALTREP
            info = self.item()
            state = self.item()
            self.item()  # attributes
            cls = info[0][1] if isinstance(info, list) else None
            if cls in ("compact_intseq", "compact_realseq"):
                st = np.asarray(state, dtype=np.float64)
                n, first, inc = int(st[0]), st[1], st[2]
                v = first + inc * np.arange(n)
                return v.astype(np.int32) if cls == "compact_intseq" else v
            if cls in ("wrap_real", "wrap_integer", "wrap_logical", "wrap_string"):
                return state[0] if isinstance(state, list) and not isinstance(state[0], tuple) else state[0][1]
            raise ValueError("unsupported ALTREP class %r" % (cls,))
        if t in (10, 13):  # logical, integer
            n = self.length()
            v = np.frombuffer(self.take(4 * n), dtype=">i4").astype(np.int32)
        elif t == 14:  # real
            n = self.length()
            v = np.frombuffer(self.take(8 * n), dtype=">f8").astype(np.float64)
        elif t == 16:  # string vector
            n = self.length()
            v = [self.item() for _ in range(n)]
        elif t in (19, 20):  # generic vector / expression
            n = self.length()
            v = [self.item() for _ in range(n)]
        else:
            raise ValueError("unsupported SEXP type %d" % t)
        if has_attr:
            attrs = self.item()
            ad = {k: a for k, a in attrs} if attrs else {}
            if "dim" in ad and isinstance(v, np.ndarray):
                v = v.reshape(tuple(int(d) for d in ad["dim"]), order="F")
            elif "names" in ad and isinstance(v, list):
                v = dict(zip(ad["names"], v))
        return v


def read_rdata(path):
    """Return ``{name: value}`` for every object saved in the workspace file.
    Integer/logical NA is ``-2147483648``; real NA is NaN."""
    with open(path, "rb") as fh:
        raw = fh.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    if raw[:5] not in (b"RDX3\n", b"RDX2\n"):
        raise ValueError("not an XDR R workspace: %r" % raw[:5])
    r = _Reader(raw)
    r.p = 5
    if r.take(2) != b"X\n":
        raise ValueError("only XDR serialisation is supported")
    version = r.i32()
    r.i32()
    r.i32()
    if version == 3:
        n = r.i32()
        r.take(n)  # native encoding name
    top = r.item()
    return {k: v for k, v in top}


def responses_to_int8(data):
    """R response cube (doubles/ints with NA) -> int8 with 0 = missing."""
    a = np.asarray(data)
    if a.dtype.kind == "f":
        out = np.where(np.isnan(a), 0, a)
    else:
        out = np.where(a == NA_INT, 0, a)
    return np.asfortranarray(out.astype(np.int8))
