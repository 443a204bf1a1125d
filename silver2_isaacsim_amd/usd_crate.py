"""Minimal reader of binary USD ("crate", `PXR-USDC` 0.8-0.10) files - enough to recover the
per-prim hydrodynamics parameter table of a scene without `pxr`.

What it decodes: the table of contents, TOKENS / FIELDS / FIELDSETS / PATHS / SPECS sections
(LZ4 "fast compression" blocks + USD's delta/2-bit-code integer compression) and inlined scalar
values (float, double, int, bool, token) plus out-of-line doubles / float or double 3-vectors.
That covers `exposedVar:hydrodynamicsBehavior:*` (float), `physics:mass` (float),
`physxScene:timeStepsPerSecond` and `xformOp:translate` on the scenes the reference ships
(src/scenes/*.usd).  Everything else (arrays, dictionaries, time samples, payloads...) is skipped.

Format notes (from the published crate layout): header = "PXR-USDC", 8 version bytes, int64 TOC
offset; every section is a count followed by compressed columns; a 64-bit value rep carries the type
in bits 48-55, flags in bits 61-63 (compressed, inlined, array) and a 48-bit payload (the value
itself when inlined, else a file offset).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

from . import config as cfg

# value-rep type codes used here
_T_BOOL, _T_INT, _T_UINT, _T_INT64, _T_UINT64, _T_HALF, _T_FLOAT, _T_DOUBLE = 1, 3, 4, 5, 6, 7, 8, 9
_T_STRING, _T_TOKEN, _T_VEC3D, _T_VEC3F = 10, 11, 23, 24
_ARRAY, _INLINED, _COMPRESSED = 1 << 63, 1 << 62, 1 << 61


class CrateError(ValueError):
    pass


# --------------------------------------------------------------------------
# LZ4 block format + USD's chunk framing
# --------------------------------------------------------------------------
def lz4_block_decompress(src: bytes, expected: int) -> bytes:
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        token = src[i]; i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1
                lit += b
                if b != 255:
                    break
        out += src[i:i + lit]; i += lit
        if i >= n:
            break                                  # last sequence has literals only
        offset = src[i] | (src[i + 1] << 8); i += 2
        if offset == 0:
            raise CrateError("LZ4: zero offset")
        mlen = token & 15
        if mlen == 15:
            while True:
                b = src[i]; i += 1
                mlen += b
                if b != 255:
                    break
        mlen += 4
        start = len(out) - offset
        if start < 0:
            raise CrateError("LZ4: offset before start of output")
        if offset >= mlen:
            out += out[start:start + mlen]
        else:                                       # overlapping copy
            for k in range(mlen):
                out.append(out[start + k])
    if expected and len(out) != expected:
        raise CrateError(f"LZ4: got {len(out)} bytes, expected {expected}")
    return bytes(out)


def fast_decompress(buf: bytes, expected: int) -> bytes:
    """TfFastCompression framing: first byte = number of chunks (0 = one LZ4 block follows)."""
    n_chunks = buf[0]
    if n_chunks == 0:
        return lz4_block_decompress(buf[1:], expected)
    out, pos = bytearray(), 1
    for _ in range(n_chunks):
        (size,) = struct.unpack_from("<i", buf, pos); pos += 4
        out += lz4_block_decompress(buf[pos:pos + size], 0); pos += size
    if expected and len(out) != expected:
        raise CrateError("chunked LZ4: size mismatch")
    return bytes(out)


def decode_ints(buf: bytes, count: int) -> list[int]:
    """USD integer compression (32-bit): int32 common delta, 2-bit codes (0 common, 1 int8,
    2 int16, 3 int32), then the variable-width deltas; values are running sums."""
    if count == 0:
        return []
    (common,) = struct.unpack_from("<i", buf, 0)
    n_code_bytes = (count * 2 + 7) // 8
    codes = buf[4:4 + n_code_bytes]
    pos = 4 + n_code_bytes
    out, prev = [], 0
    for i in range(count):
        code = (codes[i >> 2] >> ((i & 3) * 2)) & 3
        if code == 0:
            delta = common
        elif code == 1:
            (delta,) = struct.unpack_from("<b", buf, pos); pos += 1
        elif code == 2:
            (delta,) = struct.unpack_from("<h", buf, pos); pos += 2
        else:
            (delta,) = struct.unpack_from("<i", buf, pos); pos += 4
        prev = (prev + delta) & 0xFFFFFFFF
        out.append(prev)
    return out


def _signed32(x: int) -> int:
    return x - (1 << 32) if x & 0x80000000 else x


@dataclass
class Spec:
    path: str
    spec_type: int
    fields: dict


class CrateFile:
    def __init__(self, path: str):
        with open(path, "rb") as f:
            self.data = f.read()
        d = self.data
        if d[:8] != b"PXR-USDC":
            raise CrateError("not a USD crate file")
        self.version = tuple(d[8:11])
        (toc,) = struct.unpack_from("<q", d, 16)
        (n_sections,) = struct.unpack_from("<Q", d, toc)
        self.sections = {}
        for i in range(n_sections):
            off = toc + 8 + 32 * i
            name = d[off:off + 16].split(b"\0")[0].decode()
            start, size = struct.unpack_from("<qq", d, off + 16)
            self.sections[name] = (start, size)
        self._read_tokens()
        self._read_fields()
        self._read_fieldsets()
        self._read_paths()
        self._read_specs()

    # -- low level ---------------------------------------------------------
    def _compressed_ints(self, pos: int, count: int):
        (csize,) = struct.unpack_from("<Q", self.data, pos); pos += 8
        raw = fast_decompress(self.data[pos:pos + csize], 0)
        return decode_ints(raw, count), pos + csize

    def _read_tokens(self):
        pos, _ = self.sections["TOKENS"]
        n, usize, csize = struct.unpack_from("<QQQ", self.data, pos); pos += 24
        raw = fast_decompress(self.data[pos:pos + csize], usize)
        toks = raw.split(b"\0")
        self.tokens = [t.decode("utf-8", "replace") for t in toks[:n]]

    def _read_fields(self):
        pos, _ = self.sections["FIELDS"]
        (n,) = struct.unpack_from("<Q", self.data, pos); pos += 8
        tok_idx, pos = self._compressed_ints(pos, n)
        (csize,) = struct.unpack_from("<Q", self.data, pos); pos += 8
        raw = fast_decompress(self.data[pos:pos + csize], n * 8)
        reps = struct.unpack(f"<{n}Q", raw)
        self.fields = [(self.tokens[t], r) for t, r in zip(tok_idx, reps)]

    def _read_fieldsets(self):
        pos, _ = self.sections["FIELDSETS"]
        (n,) = struct.unpack_from("<Q", self.data, pos); pos += 8
        self.fieldsets, _ = self._compressed_ints(pos, n)

    def _read_paths(self):
        pos, _ = self.sections["PATHS"]
        (n_paths,) = struct.unpack_from("<Q", self.data, pos); pos += 8
        (n_enc,) = struct.unpack_from("<Q", self.data, pos); pos += 8
        path_idx, pos = self._compressed_ints(pos, n_enc)
        elem_tok, pos = self._compressed_ints(pos, n_enc)
        jumps, pos = self._compressed_ints(pos, n_enc)
        elem_tok = [_signed32(x) for x in elem_tok]
        jumps = [_signed32(x) for x in jumps]
        self.paths = [""] * n_paths

        # iterative walk of the encoded tree: jump > 0: sibling at i + jump, child at i + 1;
        # jump == -1: only a child; jump == 0: only a sibling (at i + 1); jump == -2: leaf
        stack = [(0, "")]
        while stack:
            i, parent = stack.pop()
            while True:
                if i == 0 and parent == "":
                    path = "/"
                else:
                    tok = elem_tok[i]
                    name = self.tokens[abs(tok)]
                    if tok < 0:
                        path = f"{parent}.{name}"
                    else:
                        path = f"/{name}" if parent == "/" else f"{parent}/{name}"
                self.paths[path_idx[i]] = path
                j = jumps[i]
                has_child = j > 0 or j == -1
                has_sibling = j >= 0
                if has_child and has_sibling:
                    stack.append((i + j, parent))
                    parent, i = path, i + 1
                elif has_child:
                    parent, i = path, i + 1
                elif has_sibling:
                    i = i + 1
                else:
                    break

    def _read_specs(self):
        pos, _ = self.sections["SPECS"]
        (n,) = struct.unpack_from("<Q", self.data, pos); pos += 8
        pidx, pos = self._compressed_ints(pos, n)
        fsidx, pos = self._compressed_ints(pos, n)
        stype, pos = self._compressed_ints(pos, n)
        self.specs = []
        for p, fs, t in zip(pidx, fsidx, stype):
            fields = {}
            k = fs
            while k < len(self.fieldsets) and self.fieldsets[k] != 0xFFFFFFFF:
                name, rep = self.fields[self.fieldsets[k]]
                fields[name] = rep
                k += 1
            self.specs.append(Spec(self.paths[p], t, fields))

    # -- values ------------------------------------------------------------
    def value(self, rep: int):
        """Decode a value rep; returns None for types this reader does not cover."""
        t = (rep >> 48) & 0xFF
        payload = rep & ((1 << 48) - 1)
        if rep & _ARRAY:
            return None
        if rep & _INLINED:
            lo = payload & 0xFFFFFFFF
            if t == _T_FLOAT:
                return struct.unpack("<f", struct.pack("<I", lo))[0]
            if t == _T_DOUBLE:                     # inlined doubles are stored as the float they equal
                return float(struct.unpack("<f", struct.pack("<I", lo))[0])
            if t in (_T_INT, _T_INT64):
                return _signed32(lo)
            if t in (_T_UINT, _T_UINT64):
                return lo
            if t == _T_BOOL:
                return bool(lo)
            if t == _T_TOKEN:
                return self.tokens[lo]
            if t in (_T_VEC3D, _T_VEC3F):          # inlined vectors hold three int8 components
                return tuple(float(struct.unpack("<b", bytes([(payload >> (8 * k)) & 0xFF]))[0]) for k in range(3))
            return None
        if t == _T_DOUBLE:
            return struct.unpack_from("<d", self.data, payload)[0]
        if t == _T_VEC3D:
            return struct.unpack_from("<3d", self.data, payload)
        if t == _T_VEC3F:
            return struct.unpack_from("<3f", self.data, payload)
        if t in (_T_INT64, _T_UINT64):
            return struct.unpack_from("<q", self.data, payload)[0]
        return None

    def attributes(self) -> dict[str, object]:
        """property path -> default value, for every attribute spec with a decodable `default`."""
        out = {}
        for s in self.specs:
            if "." in s.path and "default" in s.fields:
                v = self.value(s.fields["default"])
                if v is not None:
                    out[s.path] = v
        return out


def hydrodynamics_table(path: str) -> dict:
    """Per-prim hydrodynamics parameters of a scene: {prim path: {schema name: value, 'mass': ...}}
    for every prim that carries `exposedVar:hydrodynamicsBehavior:*` attributes, plus the physics
    rate under key '__scene__'."""
    crate = CrateFile(path)
    attrs = crate.attributes()
    prefix = f"{cfg.EXPOSED_ATTR_NS}:{cfg.BEHAVIOR_NS}:"
    table: dict = {}
    for full, val in attrs.items():
        prim, prop = full.rsplit(".", 1)
        if prop.startswith(prefix) and prop[len(prefix):] in cfg.SCHEMA_NAMES:
            table.setdefault(prim, {})[prop[len(prefix):]] = float(val)
    for prim in list(table):
        m = attrs.get(f"{prim}.physics:mass")
        if m is not None:
            table[prim]["mass"] = float(m)
        t = attrs.get(f"{prim}.xformOp:translate")
        if t is not None:
            table[prim]["translate"] = tuple(float(x) for x in t)
    scene = {}
    for full, val in attrs.items():
        if full.endswith(".physxScene:timeStepsPerSecond"):
            scene["timeStepsPerSecond"] = val
    table["__scene__"] = scene
    return table


def params_rows(table: dict, default_mass: float = 1.0):
    """(prim paths, (N,11) float32 params in engine order, rho, g) from `hydrodynamics_table`."""
    import numpy as np
    prims = [p for p in table if p != "__scene__"]
    rows = []
    for p in prims:
        t = {**cfg.SCHEMA_DEFAULTS, **table[p]}
        rows.append([t["xDimension"], t["yDimension"], t["zDimension"], t["linearDragCoefficient"],
                     t["angularDragCoefficient"], t["linearDamping"], t["angularDamping"], t["liftCoefficient"],
                     t["linearAddedMassCoefficient"], t["angularAddedMassCoefficient"], t.get("mass", default_mass)])
    first = {**cfg.SCHEMA_DEFAULTS, **(table[prims[0]] if prims else {})}
    return prims, np.asarray(rows, dtype=np.float32), float(first["waterDensity"]), float(first["gravity"])


if __name__ == "__main__":                       # python -m silver2_isaacsim_amd.usd_crate scene.usd
    import json
    import sys
    print(json.dumps(hydrodynamics_table(sys.argv[1]), indent=1, sort_keys=True))
