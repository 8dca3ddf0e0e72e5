// Canonical float32 arithmetic of the LBDRN hot path on gfx950.
//
// Every value the codec's integer output depends on is produced by an explicit sequence of
// IEEE-754 float32 operations (mul, add, fma, div, round-to-nearest-even), never by a library
// sin/exp whose result may change between ROCm releases.  The library is compiled with
// -ffp-contract=off, so the only fused operations are the fmaf() calls written here and the
// MFMA instructions (an f32 MFMA is a k-ordered fmaf chain, MI355X guide "FP32-input MFMA").
// DESIGN.md "Canonical arithmetic" states the sequences; oracle/ restates them independently in C.
//
// sin/cos: Cody-Waite reduction by pi/2 in three float32 pieces carried by fma, then degree-9 /
// degree-10 polynomials on [-pi/4, pi/4] (max error 1.5 ulp for |x| <= 2^16, ~1e-6 absolute up
// to 2^23; NaN/Inf give NaN).  Replaces torch.sin in Sine.forward (ref LBDRNmodel.py:12-13).
// sigmoid: 1/(1+e^-|z|) or e^-|z|/(1+e^-|z|) with a degree-6 exp kernel; replaces nn.Sigmoid
// (ref LBDRNmodel.py:75).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lbdrn {

#define LBDRN_DEV __device__ __forceinline__

constexpr float kTwoOverPi = 0x1.45f306p-1f;
constexpr float kPio2A = 0x1.921fb6p+0f, kPio2B = -0x1.777a5cp-25f, kPio2C = -0x1.ee59dap-50f;
constexpr float kSin0 = -0x1.555556p-3f, kSin1 = 0x1.11110ep-7f, kSin2 = -0x1.a0133p-13f,
                kSin3 = 0x1.6d96dp-19f;
constexpr float kCos0 = -0x1.0p-1f, kCos1 = 0x1.555556p-5f, kCos2 = -0x1.6c16bap-10f,
                kCos3 = 0x1.a0122ep-16f, kCos4 = -0x1.245a26p-22f;
constexpr float kLog2e = 0x1.715476p+0f, kLn2A = 0x1.62e43p-1f, kLn2B = -0x1.05c61p-29f;
constexpr float kExp0 = 0x1.0p-1f, kExp1 = 0x1.5554d4p-3f, kExp2 = 0x1.5554ep-5f,
                kExp3 = 0x1.121588p-7f, kExp4 = 0x1.6d4f48p-10f;

LBDRN_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// sin polynomial of the reduced argument
LBDRN_DEV float poly_sin(float r, float s)
{
    float p = fma_(fma_(fma_(kSin3, s, kSin2), s, kSin1), s, kSin0);
    return fma_(r * s, p, r);
}
LBDRN_DEV float poly_cos(float s)
{
    float p = fma_(fma_(fma_(fma_(kCos4, s, kCos3), s, kCos2), s, kCos1), s, kCos0);
    return fma_(s, p, 1.0f);
}

LBDRN_DEV float select_quadrant(float sn, float cs, int quad)
{
    float v = (quad & 1) ? cs : sn;
    return (quad & 2) ? -v : v;
}

// Reduction by pi/2, one branch-free path for every finite x.  The quadrant is k mod 4 taken in
// float arithmetic (exact for any integer-valued float), so no float->int conversion overflows.
LBDRN_DEV void reduce_pio2(float x, float& r, int& q)
{
    float k = __builtin_rintf(x * kTwoOverPi);
    r = fma_(-k, kPio2A, x);
    r = fma_(-k, kPio2B, r);
    r = fma_(-k, kPio2C, r);
    float qf = fma_(-4.0f, __builtin_floorf(k * 0.25f), k);
    // qf is 0, 1, 2 or 3 for finite k and NaN otherwise; v_cvt_i32_f32 maps NaN to 0, which is the
    // oracle's explicit "NaN -> 0" (four VALU instructions fewer per activation than spelling the guard)
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(q) : "v"(qf));
}

LBDRN_DEV float canon_sin(float x)
{
    float r;
    int q;
    reduce_pio2(x, r, q);
    float s = r * r;
    return select_quadrant(poly_sin(r, s), poly_cos(s), q);
}

LBDRN_DEV float canon_cos(float x)
{
    float r;
    int q;
    reduce_pio2(x, r, q);
    float s = r * r;
    return select_quadrant(poly_sin(r, s), poly_cos(s), q + 1);
}

// both at once (training: activation and its derivative share the reduction)
LBDRN_DEV void canon_sincos(float x, float& sn_out, float& cs_out)
{
    float r;
    int q;
    reduce_pio2(x, r, q);
    float s = r * r;
    float sn = poly_sin(r, s), cs = poly_cos(s);
    sn_out = select_quadrant(sn, cs, q);
    cs_out = select_quadrant(sn, cs, q + 1);
}

// e^(-a), 0 <= a <= 86
LBDRN_DEV float canon_exp_neg(float a)
{
    float x = -a;
    float n = __builtin_rintf(x * kLog2e);
    float r = fma_(-n, kLn2A, x);
    r = fma_(-n, kLn2B, r);
    float e = fma_(fma_(fma_(fma_(kExp4, r, kExp3), r, kExp2), r, kExp1), r, kExp0);
    float p = fma_(r * r, e, r) + 1.0f;
    int bits = __float_as_int(p) + (int)n * (1 << 23);   // (n <= 0: a multiplication, not a shift of a negative value)
    return __int_as_float(bits);
}

LBDRN_DEV float canon_sigmoid(float z)
{
    float a = __builtin_fabsf(z);
    a = (a > 86.0f) ? 86.0f : a;
    float t = canon_exp_neg(a);
    float d = 1.0f + t;
    return (z >= 0.0f) ? 1.0f / d : t / d;  // IEEE division (-fhip-fp32-correctly-rounded-divide-sqrt)
}

// Sine(w0=30): ref LBDRNmodel.py:13 -- the product 30*z is rounded to float32 first
LBDRN_DEV float siren_act(float z) { return canon_sin(30.0f * z); }

// ---- arithmetic of the passes that are held to a tolerance (north_star: encode-time float loss within 1e-5 relative),
// not to a bit pattern: the training step and the per-epoch evaluation pass that picks the best epoch.
// sin / cos from the hardware's v_sin_f32 / v_cos_f32 (arguments in revolutions, quarter rate) behind a compensated
// reduction: x / (2 pi) as a float32 product, plus that product's exact residual and x times the low half of 1 / (2 pi).
// 4.5e-7 absolute over the range the activations see (scripts/sin_probe.hip: the intrinsic error of the instructions).
constexpr float kRevHi = 0x1.45f306p-3f, kRevLo = 0x1.b93910p-28f;   // 1 / (2 pi) = kRevHi + kRevLo
LBDRN_DEV float fast_rev(float x)
{
    const float rev = x * kRevHi;
    const float res = fma_(x, kRevHi, -rev);                 // exact
    return __builtin_amdgcn_fractf(rev) + fma_(x, kRevLo, res);
}
LBDRN_DEV float fast_sin(float x) { return __builtin_amdgcn_sinf(fast_rev(x)); }
LBDRN_DEV void fast_sincos(float x, float& sn, float& cs)
{
    const float f = fast_rev(x);
    sn = __builtin_amdgcn_sinf(f);
    cs = __builtin_amdgcn_cosf(f);
}
// sigmoid by v_exp + v_rcp, ~1e-7 relative
LBDRN_DEV float fast_sigmoid(float z)
{
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * __builtin_fabsf(z));   // e^-|z|
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    return z >= 0.0f ? r : e * r;
}

// numpy.pad(mode="reflect") index map (ref LBDRNdataset.py:120-123)
LBDRN_DEV int reflect_idx(int i, int n)
{
    if (n == 1) return 0;
    int period = 2 * (n - 1);
    i %= period;
    if (i < 0) i += period;
    return (i < n) ? i : period - i;
}

// same map without the integer modulo when one reflection suffices (|offset| < n, i.e. D < size)
LBDRN_DEV int reflect_fast(int i, int n)
{
    int r = i < 0 ? -i : i;
    r = r >= n ? 2 * (n - 1) - r : r;
    if (r < 0 || r >= n) r = reflect_idx(i, n);
    return r;
}

}  // namespace lbdrn
