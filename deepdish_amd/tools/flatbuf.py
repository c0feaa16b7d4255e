"""FlatBuffers and FlexBuffers, the subset a .tflite file needs -- written from the public format descriptions (the image has
no `flatbuffers` package).

Reader: `Table(buf, pos)` with typed field accessors by vtable slot; vectors, strings, nested tables, unions.
Builder: back-to-front like the official one (children first, so every uoffset points forward), tables with explicit slots.
FlexBuffers: `flex_map` reads a root map of scalars (TFLite custom-op options), `flex_build_map` writes one.
"""
import struct

# ------------------------------------------------------------------------------------------- reader
_SCALAR = {'bool': ('<?', 1), 'i8': ('<b', 1), 'u8': ('<B', 1), 'i16': ('<h', 2), 'u16': ('<H', 2), 'i32': ('<i', 4), 'u32': ('<I', 4),
           'i64': ('<q', 8), 'u64': ('<Q', 8), 'f32': ('<f', 4), 'f64': ('<d', 8)}


class Table:
    def __init__(self, buf, pos):
        self.buf, self.pos = buf, pos
        self.vt = pos - struct.unpack_from('<i', buf, pos)[0]
        self.vt_size = struct.unpack_from('<H', buf, self.vt)[0]

    def _field(self, slot):
        o = 4 + 2 * slot
        if o + 2 > self.vt_size:
            return 0
        off = struct.unpack_from('<H', self.buf, self.vt + o)[0]
        return self.pos + off if off else 0

    def scalar(self, slot, kind, default=0):
        p = self._field(slot)
        return struct.unpack_from(_SCALAR[kind][0], self.buf, p)[0] if p else default

    def _indirect(self, slot):
        p = self._field(slot)
        return p + struct.unpack_from('<I', self.buf, p)[0] if p else 0

    def table(self, slot):
        p = self._indirect(slot)
        return Table(self.buf, p) if p else None

    def string(self, slot, default=''):
        p = self._indirect(slot)
        if not p:
            return default
        n = struct.unpack_from('<I', self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n]).decode('utf-8')

    def vector_len(self, slot):
        p = self._indirect(slot)
        return struct.unpack_from('<I', self.buf, p)[0] if p else 0

    def scalars(self, slot, kind):
        """Vector of scalars as a list (bytes for 'u8')."""
        p = self._indirect(slot)
        if not p:
            return b'' if kind == 'u8' else []
        n = struct.unpack_from('<I', self.buf, p)[0]
        fmt, size = _SCALAR[kind]
        if kind == 'u8':
            return memoryview(self.buf)[p + 4:p + 4 + n]
        return list(struct.unpack_from('<%d%s' % (n, fmt[1]), self.buf, p + 4))

    def tables(self, slot):
        p = self._indirect(slot)
        if not p:
            return []
        n = struct.unpack_from('<I', self.buf, p)[0]
        out = []
        for i in range(n):
            e = p + 4 + 4 * i
            out.append(Table(self.buf, e + struct.unpack_from('<I', self.buf, e)[0]))
        return out


def root(buf, identifier=None):
    if identifier is not None and bytes(buf[4:8]) != identifier:
        raise ValueError('not a %s flatbuffer (file identifier %r)' % (identifier.decode(), bytes(buf[4:8])))
    return Table(buf, struct.unpack_from('<I', buf, 0)[0])


# ------------------------------------------------------------------------------------------- builder
class Builder:
    """Positions are distances from the END of the finished buffer (the buffer grows at the front)."""

    def __init__(self):
        self.chunks, self.n, self.minalign = [], 0, 1

    def _push(self, b):
        self.chunks.append(bytes(b))
        self.n += len(b)

    def prep(self, align, upcoming):
        """Pad so that after `upcoming` more bytes the position is a multiple of `align`."""
        self.minalign = max(self.minalign, align)
        pad = (-(self.n + upcoming)) % align
        if pad:
            self._push(bytes(pad))

    def vector(self, data, elem_size, align=None):
        """data: bytes of the elements (already little-endian) -> position of the vector."""
        n = len(data) // elem_size
        self.prep(max(4, align or elem_size), len(data))
        self._push(data)
        self.prep(4, 4)                              # (no-op: the data start is aligned to >= 4)
        self._push(struct.pack('<I', n))
        return self.n

    def scalars(self, values, kind, align=None):
        fmt, size = _SCALAR[kind]
        return self.vector(struct.pack('<%d%s' % (len(values), fmt[1]), *values), size, align)

    def string(self, s):
        b = s.encode('utf-8')
        self.prep(4, len(b) + 1)
        self._push(b + b'\0')
        self._push(struct.pack('<I', len(b)))
        return self.n

    def offsets(self, positions):
        """Vector of uoffsets to already-written objects."""
        self.prep(4, 4 * len(positions))
        for i in range(len(positions) - 1, -1, -1):            # element i will sit at distance (n after push) from the end
            self._push(struct.pack('<I', (self.n + 4) - positions[i]))
        self._push(struct.pack('<I', len(positions)))
        return self.n

    def table(self, fields):
        """fields: {slot: (kind, value)}; kind in _SCALAR, or 'offset' with value = position of a written object (0 = absent)."""
        items = [(slot, k, v) for slot, (k, v) in sorted(fields.items()) if not (k == 'offset' and not v)]
        start = self.n
        where = {}
        for slot, k, v in sorted(items, key=lambda it: -(4 if it[1] == 'offset' else _SCALAR[it[1]][1])):     # big fields first: no padding holes
            size = 4 if k == 'offset' else _SCALAR[k][1]
            self.prep(size, size)
            if k == 'offset':
                self._push(struct.pack('<I', (self.n + 4) - v))
            else:
                self._push(struct.pack(_SCALAR[k][0], v))
            where[slot] = self.n
        self.prep(4, 4)
        n_slots = (max(where) + 1) if where else 0
        vt_len = 4 + 2 * n_slots
        self._push(struct.pack('<i', vt_len))                   # soffset: the vtable follows directly below the table
        pos = self.n
        vt = struct.pack('<HH', vt_len, pos - start)
        for sl in range(n_slots):
            vt += struct.pack('<H', pos - where[sl] if sl in where else 0)
        self._push(vt)
        if self.n % 4:                                          # keep the next object 4-aligned
            self._push(bytes(4 - self.n % 4))
        return pos

    def finish(self, root_pos, identifier=b'TFL3'):
        self.prep(max(self.minalign, 8), 8)
        self._push(identifier)
        self._push(struct.pack('<I', (self.n + 4) - root_pos))
        return b''.join(reversed(self.chunks))


# ------------------------------------------------------------------------------------------- flexbuffers (root map of scalars)
FBT_INT, FBT_UINT, FBT_FLOAT, FBT_KEY, FBT_MAP, FBT_BOOL = 1, 2, 3, 4, 9, 26


def _fx_uint(buf, pos, width):
    return int.from_bytes(buf[pos:pos + width], 'little')


def _fx_scalar(buf, pos, width, typ):
    if typ == FBT_FLOAT:
        return struct.unpack_from('<f' if width == 4 else '<d', buf, pos)[0]
    if typ == FBT_INT:
        return int.from_bytes(buf[pos:pos + width], 'little', signed=True)
    if typ == FBT_BOOL:
        return bool(_fx_uint(buf, pos, width))
    return _fx_uint(buf, pos, width)


def flex_map(buf):
    """Root map {key: int | float | bool} of a FlexBuffer (nested values are skipped)."""
    buf = bytes(buf)
    if len(buf) < 3:
        return {}
    root_width = buf[-1]
    packed = buf[-2]
    typ, bw = packed >> 2, 1 << (packed & 3)
    if typ != FBT_MAP:
        raise ValueError('flexbuffer root is not a map (type %d)' % typ)
    ref = len(buf) - 2 - root_width
    mp = ref - _fx_uint(buf, ref, root_width)                 # the map's values
    n = _fx_uint(buf, mp - bw, bw)
    keys_pos = (mp - 3 * bw) - _fx_uint(buf, mp - 3 * bw, bw)
    kbw = _fx_uint(buf, mp - 2 * bw, bw)
    out = {}
    for i in range(n):
        kp = keys_pos + i * kbw
        ks = kp - _fx_uint(buf, kp, kbw)
        key = buf[ks:buf.index(b'\0', ks)].decode()
        t = buf[mp + n * bw + i]
        vt, vw = t >> 2, 1 << (t & 3)
        if vt in (FBT_INT, FBT_UINT, FBT_FLOAT, FBT_BOOL):
            out[key] = _fx_scalar(buf, mp + i * bw, bw, vt)
        elif vt in (6, 7, 8):                                 # indirect int / uint / float
            p = (mp + i * bw) - _fx_uint(buf, mp + i * bw, bw)
            out[key] = _fx_scalar(buf, p, vw, vt - 5)
    return out


def flex_build_map(d):
    """{key: int | float | bool} -> FlexBuffer with a root map, every slot 4 bytes wide (keys sorted, as the format requires)."""
    keys = sorted(d)
    buf = bytearray()
    key_pos = []
    for k in keys:
        key_pos.append(len(buf))
        buf += k.encode() + b'\0'
    while len(buf) % 4:
        buf += b'\0'
    buf += struct.pack('<I', len(keys))                       # keys vector: length, then offsets
    keys_vec = len(buf)
    for i, kp in enumerate(key_pos):
        buf += struct.pack('<I', len(buf) - kp)
    buf += struct.pack('<I', len(buf) - keys_vec)             # map prefix: offset to the keys vector, its byte width, length
    buf += struct.pack('<I', 4)
    buf += struct.pack('<I', len(keys))
    mp = len(buf)
    types = bytearray()
    for k in keys:
        v = d[k]
        if isinstance(v, bool):
            buf += struct.pack('<I', int(v)); types.append(FBT_BOOL << 2 | 2)
        elif isinstance(v, int):
            buf += struct.pack('<i', v); types.append(FBT_INT << 2 | 2)
        else:
            buf += struct.pack('<f', float(v)); types.append(FBT_FLOAT << 2 | 2)
    buf += types
    while len(buf) % 4:
        buf += b'\0'
    buf += struct.pack('<I', len(buf) - mp)                   # root: offset to the map, packed type, root byte width
    buf += bytes([FBT_MAP << 2 | 2, 4])
    return bytes(buf)
