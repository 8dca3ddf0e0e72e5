"""Restatement of fpzip's 1-D float32 stream in plain Python integers.

TEST INFRASTRUCTURE ONLY (checker of csrc/weights_codec.hip; nothing under lbdrn-msic_amd/ imports it).
fpzip==1.2.4 (requirements.txt:2; call sites encode.py:129, decode.py:113) is absent from /root/reference and from
this image: this file restates the published algorithm (Lindstrom & Isenburg, IEEE TVCG 2006, and the public
description of its range coder and quasi-static model) independently of the C++ -- different language, different
data structures -- so that the two can be compared byte for byte.  PARITY UNPINNED against fpzip itself: no
fpzip-written stream exists here to check either of them against.
"""
import struct

MODEL_BITS, MODEL_PERIOD = 16, 0x400
FORMAT_MAJOR, FORMAT_MINOR = 0x0110, 1
M32 = 0xFFFFFFFF


def f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def u2f(u):
    return struct.unpack("<f", struct.pack("<I", u & M32))[0]


def map_forward(bits, prec):
    """float32 bit pattern -> order-preserving unsigned integer of `prec` bits"""
    shift = 32 - prec
    r = (~bits & M32) >> shift
    r ^= ((-(r >> (prec - 1))) & M32) >> (shift + 1)
    return r


def map_inverse(r, prec):
    shift = 32 - prec
    r ^= ((-(r >> (prec - 1))) & M32) >> (shift + 1)
    return ((~r & M32) << shift) & M32


def truncate_bits(bits, prec):
    """the lossy value map of a stream of `prec` bits: inverse(forward(x))"""
    return map_inverse(map_forward(bits, prec), prec)


class Model:
    def __init__(self, n):
        self.n = n
        total = 1 << MODEL_BITS
        self.freq = [total // n + (1 if i < total % n else 0) for i in range(n)]
        self.cum = [0] * (n + 1)
        self.cum[n] = total
        self.rescale = (n >> 4) | 2
        self.nextleft = 0
        self.left = self.incr = 0
        self._update()

    def _update(self):
        if self.nextleft:
            self.incr += 1
            self.left, self.nextleft = self.nextleft, 0
            return
        if self.rescale < MODEL_PERIOD:
            self.rescale = min(self.rescale << 1, MODEL_PERIOD)
        cf = missing = self.cum[self.n]
        for i in range(self.n - 1, -1, -1):
            t = self.freq[i]
            cf -= t
            self.cum[i] = cf
            t = (t >> 1) | 1
            missing -= t
            self.freq[i] = t
        assert cf == 0
        self.incr, self.nextleft = divmod(missing, self.rescale)
        self.left = self.rescale - self.nextleft

    def use(self, s):
        lo, width = self.cum[s], self.cum[s + 1] - self.cum[s]
        if not self.left:
            self._update()
        self.left -= 1
        self.freq[s] += self.incr
        return lo, width

    def find(self, target):
        s = 0
        while s + 1 < self.n and self.cum[s + 1] <= target:
            s += 1
        return (s,) + self.use(s)


class Encoder:
    def __init__(self):
        self.low, self.range, self.out = 0, M32, bytearray()

    def _put(self):
        self.out.append(self.low >> 24)
        self.low = (self.low << 8) & M32

    def _normalize(self):
        while not ((self.low ^ ((self.low + self.range) & M32)) >> 24):
            self._put()
            self.range = (self.range << 8) & M32
        if not (self.range >> 16):
            self._put()
            self._put()
            self.range = (-self.low) & M32

    def raw(self, s, bits):
        if bits > 16:
            self.raw(s & 0xFFFF, 16)
            s >>= 16
            bits -= 16
        self.range >>= bits
        self.low = (self.low + self.range * s) & M32
        self._normalize()

    def symbol(self, s, model):
        lo, width = model.use(s)
        self.range >>= MODEL_BITS
        self.low = (self.low + self.range * lo) & M32
        self.range = (self.range * width) & M32
        self._normalize()

    def finish(self):
        for _ in range(4):
            self._put()
        return bytes(self.out)


class Decoder:
    def __init__(self, data):
        self.data, self.pos = data, 0
        self.low, self.range, self.code = 0, M32, 0
        for _ in range(4):
            self.code = ((self.code << 8) | self._get()) & M32

    def _get(self):
        b = self.data[self.pos] if self.pos < len(self.data) else 0
        self.pos += 1
        return b

    def _normalize(self):
        while not ((self.low ^ ((self.low + self.range) & M32)) >> 24):
            self.code = ((self.code << 8) | self._get()) & M32
            self.low = (self.low << 8) & M32
            self.range = (self.range << 8) & M32
        if not (self.range >> 16):
            self.code = ((self.code << 8) | self._get()) & M32
            self.code = ((self.code << 8) | self._get()) & M32
            self.low = (self.low << 16) & M32
            self.range = (-self.low) & M32

    def raw(self, bits):
        if bits > 16:
            lo = self.raw(16)
            return lo | (self.raw(bits - 16) << 16)
        self.range >>= bits
        s = ((self.code - self.low) & M32) // self.range
        self.low = (self.low + self.range * s) & M32
        self._normalize()
        return s

    def symbol(self, model):
        self.range >>= MODEL_BITS
        target = min(((self.code - self.low) & M32) // self.range, (1 << MODEL_BITS) - 1)
        s, lo, width = model.find(target)
        self.low = (self.low + self.range * lo) & M32
        self.range = (self.range * width) & M32
        self._normalize()
        return s


def _predict_bits(prev_bits):
    """1-D Lorenzo prediction: the previous reconstructed value through fpzip's seven-term float sum with six zero
    terms (x - 0 + 0 - 0 + 0 - 0 + 0): the identity except that -0.0 becomes +0.0"""
    return 0 if prev_bits == 0x80000000 else prev_bits


def compress(values, precision=16):
    """values: iterable of float32 bit patterns (uint32) -> stream bytes"""
    prec = 32 if precision == 0 else precision
    vals = [int(v) for v in values]
    e = Encoder()
    for ch in (ord("f"), ord("p"), ord("z"), 0):
        e.raw(ch, 8)
    e.raw(FORMAT_MAJOR, 16)
    e.raw(FORMAT_MINOR, 8)
    e.raw(0, 1)
    e.raw(prec, 7)
    for dim in (len(vals), 1, 1, 1):
        e.raw(dim, 32)
    model, bias, prev = Model(2 * prec + 1), prec, 0
    for bits in vals:
        a, p = map_forward(bits, prec), map_forward(_predict_bits(prev), prec)
        if p < a:
            d = a - p
            k = d.bit_length() - 1
            e.symbol(bias + 1 + k, model)
            e.raw(d - (1 << k), k)
        elif p > a:
            d = p - a
            k = d.bit_length() - 1
            e.symbol(bias - 1 - k, model)
            e.raw(d - (1 << k), k)
        else:
            e.symbol(bias, model)
        prev = map_inverse(a, prec)
    return e.finish()


def decompress(data):
    """stream bytes -> (list of float32 bit patterns, precision)"""
    d = Decoder(bytes(data))
    if [d.raw(8) for _ in range(4)] != [ord("f"), ord("p"), ord("z"), 0]:
        raise ValueError("not an fpzip stream")
    d.raw(16)
    d.raw(8)
    if d.raw(1) != 0:
        raise ValueError("double-precision stream")
    prec = d.raw(7)
    n, ny, nz, nf = d.raw(32), d.raw(32), d.raw(32), d.raw(32)
    if (ny, nz, nf) != (1, 1, 1):
        raise ValueError("not a 1-D stream")
    model, bias, prev, out = Model(2 * prec + 1), prec, 0, []
    mask = (1 << prec) - 1
    for _ in range(n):
        p = map_forward(_predict_bits(prev), prec)
        s = d.symbol(model)
        if s > bias:
            k = s - bias - 1
            a = p + (1 << k) + d.raw(k)
        elif s < bias:
            k = bias - 1 - s
            a = p - ((1 << k) + d.raw(k))
        else:
            a = p
        prev = map_inverse(a & mask, prec)
        out.append(prev)
    return out, prec
