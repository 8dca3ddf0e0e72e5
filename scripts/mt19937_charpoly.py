"""Where csrc/mt_jump.inc's MT_PHI comes from: the characteristic polynomial of MT19937's state transition, as the
minimal polynomial (Berlekamp-Massey over GF(2)) of one bit of the raw word sequence -- the period 2^19937 - 1 is prime, so
every non-zero bit sequence of the generator has the same one.  Prints its 135 exponents, then checks a jump polynomial
g = t^J mod phi the way the library uses it: x[J + k] = XOR over the set bits i of g of x[i + k].

    python scripts/mt19937_charpoly.py          (about three seconds; polynomials are Python integers)
"""
N, M = 624, 397


def mt_words(seed, total):
    x = [0] * total
    x[0] = seed & 0xFFFFFFFF
    for k in range(1, N):
        x[k] = (1812433253 * (x[k - 1] ^ (x[k - 1] >> 30)) + k) & 0xFFFFFFFF
    for k in range(total - N):
        y = (x[k] & 0x80000000) | (x[k + 1] & 0x7FFFFFFF)
        x[k + N] = x[k + M] ^ (y >> 1) ^ (0x9908B0DF if x[k + 1] & 1 else 0)
    return x


def berlekamp_massey(bits):
    """connection polynomial C (bit i = c_i, c_0 = 1) with s[n] = XOR_{i=1..L} c_i s[n-i], and L"""
    n = len(bits)
    rev = 0
    for k, b in enumerate(bits):
        if b:
            rev |= 1 << (n - 1 - k)   # bit j of rev = s[n-1-j]: (rev >> (n-1-nn)) has s[nn-i] at bit i
    C, B, L, m = 1, 1, 0, 1
    for nn in range(n):
        if bin(C & (rev >> (n - 1 - nn))).count("1") & 1:
            if 2 * L <= nn:
                C, B, L, m = C ^ (B << m), C, nn + 1 - L, 1
            else:
                C ^= B << m
                m += 1
        else:
            m += 1
    return C, L


def main():
    J = 624 * 256
    x = mt_words(5489, J + 2 * 19937 + 2000)
    C, L = berlekamp_massey([(x[1 + k] >> 5) & 1 for k in range(2 * 19937 + 64)])
    assert L == 19937
    phi = sum(1 << (L - i) for i in range(L + 1) if (C >> i) & 1)   # phi(t) = t^L C(1/t)
    exps = [i for i in range(L + 1) if (phi >> i) & 1]
    print(len(exps), "terms:", exps)

    def mulmod(a, b):
        r = 0
        while a:
            low = a & -a
            r ^= b << (low.bit_length() - 1)
            a ^= low
        while r.bit_length() - 1 >= L:
            r ^= phi << (r.bit_length() - 1 - L)
        return r

    g, sq, e = 1, 2, J
    while e:
        if e & 1:
            g = mulmod(g, sq)
        sq = mulmod(sq, sq)
        e >>= 1
    idx = [i for i in range(L) if (g >> i) & 1]
    for k in (1, 2, 623):
        v = 0
        for i in idx:
            v ^= x[i + k]
        assert v == x[J + k]
    print("t^%d mod phi: %d terms; x[J + k] = XOR x[i + k] holds" % (J, len(idx)))


if __name__ == "__main__":
    main()
