// Weight payload of the .bin container: the float32 parameter vector as an fpzip stream
// (ref encode.py:129 `fpzip.compress(params, precision=args.precision, order='C')`, decode.py:113
// `fpzip.decompress(compressed_bytes, order='C')[0][0][0]`).
//
// fpzip (P. Lindstrom, LLNL; the reference pins fpzip==1.2.4, requirements.txt:2) is absent from /root/reference and
// from this image, so this file RESTATES its published algorithm for the one case the reference uses -- a 1-D
// float32 array, nx = N, ny = nz = nf = 1, precision 2..32 bits -- from the description of the method (Lindstrom &
// Isenburg, "Fast and efficient compression of floating-point data", IEEE TVCG 2006) and of its building blocks:
//   * the order-preserving map float32 -> unsigned integer of `precision` bits: complement the bit pattern, drop the
//     32-precision low bits, flip the remaining bits of negative numbers (PCmap<float, width>::forward); its inverse
//     re-expands with the dropped bits zero.  The lossy value map is therefore "keep the top `precision` bits of the
//     IEEE bit pattern" for every input: negative, zero (both signs), denormal, infinity, NaN (a NaN whose payload
//     lives only in the dropped bits becomes an infinity);
//   * the Lorenzo predictor, which for a 1-D array is the previous reconstructed value (the seven-term float sum
//     with six zero terms is evaluated in fpzip's order, so a predecessor of -0.0 predicts +0.0);
//   * the residual coder: symbol bias +- (1 + position of the residual's top bit), `bias` = precision, from an
//     adaptive quasi-static frequency model (M. Schindler's range-coder model: 16-bit total, rescale period 1024),
//     followed by the residual's lower bits as equiprobable raw bits;
//   * the byte-oriented carry-less range coder (32-bit low / range, a byte out whenever the top byte is settled, a
//     16-bit flush when the range underflows);
//   * the stream header, itself range-coded as raw bits: 'f' 'p' 'z' '\0', format version (16 + 8 bits), type (1 bit),
//     precision (7 bits), nx, ny, nz, nf (32 bits each).
// PARITY UNPINNED: no fpzip-written stream and no fpzip source are available here, so byte compatibility with the
// reference's payload is NOT verified (the header's version constants and the model constants are recalled from the
// published implementation, not checked).  What is tested: the value map on every class of float, the exact
// round trip, and byte equality with an independent Python restatement (oracle/fpz_port.py).
//
// A 17,544-value payload is a serial entropy code of 35 KB: it runs on the host (the one place where this C ABI
// takes HOST pointers; stated in include/lbdrn_hip.h).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "common.hpp"

namespace lbdrn {
namespace {

constexpr unsigned MODEL_BITS = 16, MODEL_PERIOD = 0x400;
constexpr unsigned FPZ_FORMAT_MAJOR = 0x0110, FPZ_FORMAT_MINOR = 1;   // recalled, unverified (see header comment)

struct QsModel {   // adaptive frequency model: frequencies halved every `rescale` symbols, increments in between
    unsigned n, left, nextleft, incr, rescale;
    std::vector<unsigned> symf, cumf;
    explicit QsModel(unsigned symbols) : n(symbols), symf(symbols + 1), cumf(symbols + 1)
    {
        cumf[0] = 0;
        cumf[n] = 1u << MODEL_BITS;
        rescale = (n >> 4) | 2;
        nextleft = 0;
        const unsigned initval = cumf[n] / n, end = cumf[n] % n;
        for (unsigned i = 0; i < n; ++i) symf[i] = initval + (i < end ? 1 : 0);
        update();
    }
    void update()
    {
        if (nextleft) {   // some more symbols at the next larger increment before the real rescaling
            incr++;
            left = nextleft;
            nextleft = 0;
            return;
        }
        if (rescale < MODEL_PERIOD) {
            rescale <<= 1;
            if (rescale > MODEL_PERIOD) rescale = MODEL_PERIOD;
        }
        unsigned cf = cumf[n], missing = cumf[n];
        for (unsigned i = n; i--;) {
            unsigned tmp = symf[i];
            cf -= tmp;
            cumf[i] = cf;
            tmp = (tmp >> 1) | 1;
            missing -= tmp;
            symf[i] = tmp;
        }
        incr = missing / rescale;
        nextleft = missing % rescale;
        left = rescale - nextleft;
    }
    void bump(unsigned s)
    {
        if (!left) update();
        left--;
        symf[s] += incr;
    }
    void interval(unsigned s, unsigned* l, unsigned* r)
    {
        *l = cumf[s];
        *r = cumf[s + 1] - cumf[s];
        bump(s);
    }
    unsigned find(unsigned target, unsigned* l, unsigned* r)   // symbol whose interval holds `target`
    {
        unsigned s = 0;
        while (s + 1 < n && cumf[s + 1] <= target) ++s;
        *l = cumf[s];
        *r = cumf[s + 1] - cumf[s];
        bump(s);
        return s;
    }
};

struct RangeEncoder {
    std::vector<uint8_t>& out;
    uint32_t low = 0, range = 0xFFFFFFFFu;
    explicit RangeEncoder(std::vector<uint8_t>& o) : out(o) {}
    void put() { out.push_back((uint8_t)(low >> 24)); low <<= 8; }
    void normalize()
    {
        while (!((low ^ (low + range)) >> 24)) {   // top byte settled
            put();
            range <<= 8;
        }
        if (!(range >> 16)) {   // range too small and the top byte still open: give up the carry, flush 16 bits
            put();
            put();
            range = 0u - low;
        }
    }
    void raw(uint32_t s, unsigned bits)   // equiprobable symbol of `bits` <= 16 bits
    {
        range >>= bits;
        low += range * s;
        normalize();
    }
    void raw_wide(uint32_t s, unsigned bits)
    {
        if (bits > 16) {
            raw(s & 0xFFFFu, 16);
            s >>= 16;
            bits -= 16;
        }
        raw(s, bits);
    }
    void symbol(unsigned s, QsModel& m)
    {
        unsigned l, r;
        m.interval(s, &l, &r);
        range >>= MODEL_BITS;
        low += range * l;
        range *= r;
        normalize();
    }
    void finish() { for (int k = 0; k < 4; ++k) put(); }
};

struct RangeDecoder {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t low = 0, range = 0xFFFFFFFFu, code = 0;
    bool overrun = false;
    RangeDecoder(const uint8_t* b, size_t n) : p(b), end(b + n)
    {
        for (int k = 0; k < 4; ++k) code = (code << 8) | get();
    }
    uint32_t get()
    {
        if (p < end) return *p++;
        overrun = true;   // (a well-formed stream is never read past its four flush bytes)
        return 0;
    }
    void normalize()
    {
        while (!((low ^ (low + range)) >> 24)) {
            code = (code << 8) | get();
            low <<= 8;
            range <<= 8;
        }
        if (!(range >> 16)) {
            code = (code << 8) | get();
            code = (code << 8) | get();
            low <<= 16;
            range = 0u - low;
        }
    }
    uint32_t raw(unsigned bits)
    {
        range >>= bits;
        if (!range) { overrun = true; range = 1; }   // (only a corrupt stream gets here)
        const uint32_t s = (code - low) / range;
        low += range * s;
        normalize();
        return s;
    }
    uint32_t raw_wide(unsigned bits)
    {
        if (bits > 16) {
            const uint32_t lo = raw(16);
            return lo | (raw(bits - 16) << 16);
        }
        return raw(bits);
    }
    unsigned symbol(QsModel& m)
    {
        range >>= MODEL_BITS;
        if (!range) { overrun = true; range = 1; }
        uint32_t target = (code - low) / range;
        if (target >= (1u << MODEL_BITS)) target = (1u << MODEL_BITS) - 1;   // (only a corrupt stream gets here)
        unsigned l, r;
        const unsigned s = m.find(target, &l, &r);
        low += range * l;
        range *= r;
        normalize();
        return s;
    }
};

inline uint32_t map_forward(uint32_t bits, unsigned prec)
{
    const unsigned shift = 32 - prec;
    uint32_t r = ~bits;
    r >>= shift;
    r ^= (0u - (r >> (prec - 1))) >> (shift + 1);
    return r;
}
inline uint32_t map_inverse(uint32_t r, unsigned prec)
{
    const unsigned shift = 32 - prec;
    r ^= (0u - (r >> (prec - 1))) >> (shift + 1);
    r = ~r;
    r <<= shift;
    return r;
}
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned top_bit(uint32_t d) { unsigned k = 0; while (d >>= 1) ++k; return k; }

// the 1-D Lorenzo prediction from the previous reconstructed value: fpzip's seven-term float sum with six zero
// terms (x - 0 + 0 - 0 + 0 - 0 + 0) is the identity except that -0.0 comes out as +0.0.  Stated on the bit pattern,
// so that the stream does not depend on how a platform's float unit treats signalling NaNs.
inline float predict(float prev)
{
    const uint32_t u = f2u(prev);
    return u2f(u == 0x80000000u ? 0u : u);
}

}  // namespace

size_t weights_bound(int64_t n) { return 64 + (size_t)n * 6; }

// fpzip switches to a narrow residual coder (one symbol per residual, no raw bits) at precisions of 8 bits and
// below; this restatement has only the wide one, so those precisions are refused on both sides instead of writing
// or reading a stream in the wrong syntax
static constexpr unsigned FPZ_NARROW_MAX = 8;

int weights_encode(const float* values, int64_t n, int precision, uint8_t* out, size_t cap, size_t* nbytes)
{
    const unsigned prec = precision == 0 ? 32 : (unsigned)precision;
    if (prec <= FPZ_NARROW_MAX || prec > 32) {
        set_error("weight payload: precision %d outside (8, 32] (fpzip's narrow coder for <= 8 bits is not implemented)", precision);
        return LBDRN_E_UNSUPPORTED;
    }
    if (n < 0 || n > 0xFFFFFFFFll) { set_error("weight payload: %lld values", (long long)n); return LBDRN_E_ARG; }
    std::vector<uint8_t> buf;
    buf.reserve((size_t)n * 3 + 64);
    RangeEncoder re(buf);
    re.raw('f', 8); re.raw('p', 8); re.raw('z', 8); re.raw(0, 8);
    re.raw(FPZ_FORMAT_MAJOR, 16);
    re.raw(FPZ_FORMAT_MINOR, 8);
    re.raw(0, 1);            // type: float
    re.raw(prec, 7);
    re.raw_wide((uint32_t)n, 32); re.raw_wide(1, 32); re.raw_wide(1, 32); re.raw_wide(1, 32);
    QsModel model(2 * prec + 1);
    const unsigned bias = prec;
    float prev = 0.0f;
    for (int64_t k = 0; k < n; ++k) {
        const uint32_t a = map_forward(f2u(values[k]), prec), p = map_forward(f2u(predict(prev)), prec);
        if (p < a) {
            const uint32_t d = a - p;
            const unsigned t = top_bit(d);
            re.symbol(bias + 1 + t, model);
            re.raw_wide(d - (1u << t), t);
        } else if (p > a) {
            const uint32_t d = p - a;
            const unsigned t = top_bit(d);
            re.symbol(bias - 1 - t, model);
            re.raw_wide(d - (1u << t), t);
        } else {
            re.symbol(bias, model);
        }
        prev = u2f(map_inverse(a, prec));
    }
    re.finish();
    *nbytes = buf.size();
    if (buf.size() > cap) {
        set_error("weight payload needs %zu bytes, the buffer holds %zu", buf.size(), cap);
        return LBDRN_E_WORKSPACE;
    }
    memcpy(out, buf.data(), buf.size());
    return 0;
}

// header only: count and precision of a stream (0 on success)
int weights_info(const uint8_t* in, size_t nbytes, int64_t* n, int* precision)
{
    if (nbytes < 8) { set_error("weight payload too short"); return LBDRN_E_ARG; }
    RangeDecoder rd(in, nbytes);
    if (rd.raw(8) != 'f' || rd.raw(8) != 'p' || rd.raw(8) != 'z' || rd.raw(8) != 0) {
        set_error("weight payload is not an fpzip stream (bad magic)");
        return LBDRN_E_ARG;
    }
    (void)rd.raw(16);
    (void)rd.raw(8);
    const uint32_t type = rd.raw(1), prec = rd.raw(7);
    const uint32_t nx = rd.raw_wide(32), ny = rd.raw_wide(32), nz = rd.raw_wide(32), nf = rd.raw_wide(32);
    if (type != 0 || prec <= FPZ_NARROW_MAX || prec > 32 || rd.overrun) {
        set_error("weight payload: unsupported fpzip stream (type %u, precision %u)", type, prec);
        return LBDRN_E_UNSUPPORTED;
    }
    if (ny != 1 || nz != 1 || nf != 1) {
        set_error("weight payload: only 1-D fpzip streams are supported (got %u x %u x %u x %u)", nx, ny, nz, nf);
        return LBDRN_E_UNSUPPORTED;
    }
    *n = nx;
    *precision = (int)prec;
    return 0;
}

int weights_decode(const uint8_t* in, size_t nbytes, float* values, int64_t cap)
{
    int64_t n = 0;
    int precision = 0;
    if (int rc = weights_info(in, nbytes, &n, &precision)) return rc;
    if (n > cap) { set_error("weight payload holds %lld values, the buffer %lld", (long long)n, (long long)cap); return LBDRN_E_WORKSPACE; }
    const unsigned prec = (unsigned)precision;
    RangeDecoder rd(in, nbytes);
    for (int k = 0; k < 4; ++k) (void)rd.raw(8);
    (void)rd.raw(16); (void)rd.raw(8); (void)rd.raw(1); (void)rd.raw(7);
    for (int k = 0; k < 4; ++k) (void)rd.raw_wide(32);
    QsModel model(2 * prec + 1);
    const unsigned bias = prec;
    const uint32_t vmask = prec == 32 ? 0xFFFFFFFFu : ((1u << prec) - 1);
    float prev = 0.0f;
    for (int64_t k = 0; k < n; ++k) {
        const uint32_t p = map_forward(f2u(predict(prev)), prec);
        const unsigned s = rd.symbol(model);
        uint32_t a;
        if (s > bias) {
            const unsigned t = s - bias - 1;
            a = p + ((1u << t) + rd.raw_wide(t));
        } else if (s < bias) {
            const unsigned t = bias - 1 - s;
            a = p - ((1u << t) + rd.raw_wide(t));
        } else {
            a = p;
        }
        a &= vmask;
        prev = u2f(map_inverse(a, prec));
        values[k] = prev;
    }
    if (rd.overrun) { set_error("weight payload is truncated"); return LBDRN_E_ARG; }
    // a well-formed stream ends with the coder's four flush bytes, all of them already pulled into `code`: a decoder
    // that stops anywhere else was fed a damaged stream (or trailing bytes that are not part of it)
    if (rd.p != rd.end) {
        set_error("weight payload: %zu bytes left behind the last value", (size_t)(rd.end - rd.p));
        return LBDRN_E_ARG;
    }
    return 0;
}

}  // namespace lbdrn
