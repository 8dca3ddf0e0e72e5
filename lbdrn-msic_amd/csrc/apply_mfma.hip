// Fused apply kernel for gfx950: neighbourhood gather + normalise + nl x (Linear -> sin(30 z)) +
// Linear -> sigmoid, then either integer reconstruction (decode.py:122-134) or the squared error
// against the labels (LBDRNperformance.py:18-21), in ONE pass over the image with no feature
// matrix in HBM (the reference materialises [N,F] float32 = 3.36 GB at 2048^2x8, D=2).
//
// Mapping to CDNA4
//  * one wave = one 32-pixel row segment; every layer is computed transposed,
//    H^T[neuron][pixel] = W[neuron][k] * X^T[k][pixel], on v_mfma_f32_32x32x2_f32: A = weights
//    (one VGPR), B = inputs (one VGPR), accumulator tile = 32 neurons x 32 pixels (16 VGPRs).
//  * an f32 MFMA is a k-ordered fmaf chain seeded by the C operand, so seeding C with the bias
//    and walking k = 0,1,2,... reproduces the generic kernels / the oracle bit for bit.
//  * the accumulator layout (row = (r&3)+8(r>>2)+4(lane>>5), col = lane&31) is turned into the
//    next layer's B operand without any data movement: tile row i is assigned neuron
//    pi(i) = 2*((i&3)+4*(i>>3)) + ((i>>2)&1), so register r of lane-half h holds neuron 2r+h --
//    exactly the k = 2s+h the next MFMA step s consumes from that lane.  The permutation is
//    applied once when the weights are packed into "fragment order".
//  * weights (76 KB for F=200, bc=64) live in LDS in fragment order, so every A operand is one
//    conflict-free ds_read_b64; the D-ring of normalised MSB values is staged per 16x64 tile in
//    channel-planar LDS (conflict-free for 32 consecutive pixels) with reflect padding applied.
//  * 8 waves per CU (2 per SIMD): one wave's sin() VALU work overlaps the other's MFMAs.
#include <atomic>
#include <cstdlib>

#include "common.hpp"
#include "lbdrn_math.hpp"

namespace lbdrn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifndef LBDRN_APPLY_CHUNK
#define LBDRN_APPLY_CHUNK 20   // layer-0 steps whose B operands are made in one go, a multiple of 4 (0: four at a time, round
#endif                         // 2's loop); eval pass on the 8 x 2048^2 tile: 1.845 ms at 0, 1.79-1.82 at 20, 1.87 at 52
constexpr int APPLY_WAVES = 8;
constexpr int APPLY_THREADS = APPLY_WAVES * 64;
constexpr int TILE_W = 64;

// tile row i of a hidden accumulator tile carries neuron pi(i) (+32*tile)
__host__ __device__ inline int tile_row_to_neuron(int i)
{
    return 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1);
}
// accumulator register r of lane-half h is tile row:
__host__ __device__ inline int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct ApplyPlan {
    int NT;          // hidden tiles = bc/32
    int S0;          // MFMA steps of layer 0: P positional + colour steps padded to a multiple of 4
                     // (pad steps carry zero weights: fmaf(0, finite, acc) == acc)
    int TH;          // tile rows
    int SH, SW;      // staged rows / pitch = TH+2D, TILE_W+2D
    int ncolor;      // colour features = F - 2P
    int pair;        // 1: the tolerance-arithmetic pass walks the colour features in CHANNEL PAIRS (k_apply_mfma, layer0_pair):
                     // step (c, r) multiplies window position r of channel c in lane half 0 and of channel c + C/2 in lane
                     // half 1; with RELATIVE the window centre -- an exact zero -- is not a step
    int RS;          // pair: steps per channel pair = (2D+1)^2, minus the centre with RELATIVE
    int x16;         // 1 (opt-in, LBDRN_EVAL_X16): the colour features of layer 0 on the f16 matrix pipe with EXACT operands
                     // (layer0_pair_x16): integer window differences x W_0 in three fp16 pieces, f32 accumulation
    int M16;         // x16: MFMAs per channel pair and weight piece = RS / 8
    int off_scale;   // x16: one float, 2^-s / msb_max (what turns the scaled integer accumulator into the pre-activation)
    // float offsets inside the packed buffer == inside LDS
    int off_w0, off_wh, off_wl, off_bh, off_bl, pack_floats;
    // further LDS regions (float offsets)
    int off_ktab, off_rowt, off_colt, off_tile, lds_floats;
    int tiles_x, tiles_y;
};

constexpr int X16_PIECES = 3;   // fp16 pieces of a float32 weight: 3 x 11 bits hold all 24 of its mantissa

static bool make_plan(const lbdrn_geom& g, const lbdrn_net& net, ApplyPlan* p, bool fast = false, bool x16 = false)
{
    // hidden activation: Sine(30) (LBDRNmodel.py:37) or nn.ReLU (the alternative of encode.py:75 / decode.py:108): a template
    // parameter of k_apply_mfma
    if (net.act != LBDRN_ACT_SINE && net.act != LBDRN_ACT_RELU) return false;
    if (net.bc % 32 != 0 || net.bc < 32 || net.bc > 128) return false;
    if (net.C > 32 || net.nl < 1 || net.nl > 15) return false;
    ApplyPlan q;
    q.NT = net.bc / 32;
    q.ncolor = net.F - 2 * g.P;
    if (q.ncolor < 0) return false;
    q.S0 = g.P + ((q.ncolor + 1) / 2 + 3) / 4 * 4;
    // the evaluation pass in the tolerance arithmetic may take the features in any order (the sum is held to 1e-6, not to
    // a bit pattern): channel pairs, and no step for the window centres, which are exact zeros with RELATIVE
    // (LBDRNdataset.py:126-128) -- 96 MFMA steps instead of 100 at the headline shape.  (-DLBDRN_EXP_EVAL_NOPAIR: A/B build)
#ifdef LBDRN_EXP_EVAL_NOPAIR
    constexpr bool nopair = true;
#else
    constexpr bool nopair = false;
#endif
    const int side = 2 * g.D + 1;
    q.pair = fast && !nopair && q.NT <= 2 && g.use_colors && g.D >= 1 && g.D <= 3 && (g.C % 2) == 0 && q.ncolor == g.C * side * side;
    // (NT = 4, bc = 128: the pair's 24 operands on top of 128 accumulator / activation registers would spill at two waves per SIMD)
    q.RS = side * side - ((g.relative && g.D > 0) ? 1 : 0);
    if (q.pair) q.S0 = g.P + (g.C / 2) * q.RS;
    // the exact-operand f16 layer 0: relative colours only (the differences of two MSB values are integers; (2D+1)^2 - 1 is a
    // multiple of 8), no positional features, MSB values that fp16 holds exactly (<= 2047: any 16-bit image at K >= 5)
    q.x16 = x16 && q.pair && g.relative && g.P == 0 && g.msb_max >= 1 && g.msb_max <= 2047 && (q.RS % 8) == 0;
    q.M16 = q.RS / 8;
    const int half = net.bc / 2;
    int o = 0;
    q.off_w0 = o; o += q.x16 ? (g.C / 2) * q.M16 * X16_PIECES * 64 * q.NT * 4 : q.S0 * 64 * q.NT;
    q.off_wh = o; o += (net.nl - 1) * half * 64 * q.NT;
    q.off_wl = o; o += half * 64;
    q.off_bh = o; o += net.nl * q.NT * 32;
    q.off_bl = o; o += 32;
    q.off_scale = o; o += 4;
    q.pack_floats = o;
    q.off_ktab = o; o += 2 * 2 * (q.S0 - g.P) + 2;
    q.SW = TILE_W + 2 * g.D;
    const int fixed = o;
    int th = 16;
    for (;; th >>= 1) {
        if (th == 0) return false;
        q.TH = th;
        q.SH = th + 2 * g.D;
        o = fixed;
        q.off_rowt = o; o += th * g.P;
        q.off_colt = o; o += TILE_W * g.P;
        q.off_tile = o; o += (g.use_colors ? g.C * q.SH * q.SW : 0);
        q.lds_floats = std::max(o, q.off_rowt + 2 * APPLY_THREADS + 4);   // the per-tile regions double as the SSE reduction's doubles
        if ((size_t)q.lds_floats * 4 <= 160 * 1024) break;
    }
    q.tiles_x = (g.W + TILE_W - 1) / TILE_W;
    q.tiles_y = (g.H + q.TH - 1) / q.TH;
    *p = q;
    return true;
}

struct WApplyPlan;
static bool wapply_supported(const lbdrn_geom& g, const lbdrn_net& net);
static size_t wapply_workspace(const lbdrn_geom& g, const lbdrn_net& net);

bool mfma_apply_supported(const lbdrn_geom& g, const lbdrn_net& net)
{
    ApplyPlan p;
    return make_plan(g, net, &p) || wapply_supported(g, net);
}

constexpr int MFMA_PARTIALS = 1024;

// the packed weights of the largest of a shape's plans | the SSE partials | one float: max |W_0| (x16)
static size_t apply_pack_bytes(const lbdrn_geom& g, const lbdrn_net& net)
{
    ApplyPlan p{}, q{};
    if (!make_plan(g, net, &p)) return 0;
    lbdrn_geom gx = g;
    gx.msb_max = 1;                     // (the workspace is sized before the image's maximum is known to the caller's struct)
    const bool x = make_plan(gx, net, &q, true, true) && q.x16;
    return align_up((size_t)std::max(p.pack_floats, x ? q.pack_floats : 0) * 4, 256);
}

size_t mfma_apply_workspace(const lbdrn_geom& g, const lbdrn_net& net)
{
    ApplyPlan p;
    if (!make_plan(g, net, &p)) return wapply_workspace(g, net);
    return apply_pack_bytes(g, net) + align_up(MFMA_PARTIALS * sizeof(double), 256) + 256;
}

// ------------------------------------------------------------------ weight packing

// params (state_dict order) -> fragment order; one thread per packed float
// max |W_0| (x16: fixes the power of two that brings the weights into fp16's range); one block
__global__ void __launch_bounds__(256) k_w0_absmax(const float* __restrict__ params, int n, float* __restrict__ out)
{
    __shared__ float red[256];
    float m = 0.0f;
    for (int e = threadIdx.x; e < n; e += 256) m = fmaxf(m, fabsf(params[e]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// x16: the power of two s with max |W_0| 2^s < 2^15 (fp16 holds every scaled weight; 0 for an all-zero layer)
__device__ __forceinline__ int x16_shift(float amax)
{
    if (!(amax > 0.0f) || !(amax < 3.0e38f)) return 0;
    int ex;
    (void)frexpf(amax, &ex);      // amax = m 2^ex, 0.5 <= m < 1
    return 15 - ex;
}

__global__ void __launch_bounds__(256)
    k_pack_apply(const float* __restrict__ params, lbdrn_net net, ApplyPlan p, int P, int C, int msb_max, const float* __restrict__ amax,
                 float* __restrict__ out)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.pack_floats) return;
    const int NT = p.NT, half = net.bc / 2;
    float v = 0.0f;
    if (p.x16 && e < p.off_wh) {
        // layer 0, exact-operand f16 (layer0_pair_x16): [channel pair][MFMA m][piece][lane][tile][8 halfs]; this thread: two
        // halfs.  Lane (row i = lane & 31, half h = lane >> 5) of MFMA m holds W_0[neuron(i, tile)][channel cp + h C/2][window
        // positions 8 m .. 8 m + 7, the centre left out], scaled by 2^s and cut into three fp16 pieces whose sum IS the float32
        // weight (3 x 11 bits of mantissa)
        const int j = e & 3, q4 = e >> 2;
        const int t = q4 % NT, lane = (q4 / NT) % 64, rest = q4 / (NT * 64);
        const int piece = rest % X16_PIECES, m = (rest / X16_PIECES) % p.M16, cp = rest / (X16_PIECES * p.M16);
        const int s2 = p.ncolor / C, cen = s2 / 2, sh = x16_shift(*amax);
        const int neuron = tile_row_to_neuron(lane & 31) + 32 * t, ch = cp + (lane >> 5) * (C / 2);
        _Float16 hv[2];
        for (int u = 0; u < 2; ++u) {
            const int pos = 8 * m + 2 * j + u, r = pos + (pos >= cen ? 1 : 0);
            const float w = ldexpf(params[(int64_t)neuron * net.F + ch * s2 + r], sh);
            const _Float16 hi = (_Float16)w;
            const float r1 = w - (float)hi;
            const _Float16 mid = (_Float16)r1;
            const float r2 = r1 - (float)mid;
            hv[u] = piece == 0 ? hi : piece == 1 ? mid : (_Float16)r2;
        }
        f16x2 pk = {hv[0], hv[1]};
        out[e] = __builtin_bit_cast(float, pk);
        return;
    }
    if (e >= p.off_scale) {   // x16: acc * this = the pre-activation's sum (acc = sum of (w 2^s) x integer difference)
        out[e] = (p.x16 && e == p.off_scale) ? ldexpf(1.0f, -x16_shift(*amax)) / (float)msb_max : 0.0f;
        return;
    }
    if (e < p.off_wh) {  // layer 0: [S0][64][NT]
        int t = e % NT, lane = (e / NT) % 64, s = e / (NT * 64);
        int k = 2 * s + (lane >> 5);
        if (p.pair && s >= P) {   // step (c, r'): channel c + (C/2) * lane half, window position r' (the centre left out)
            const int s2 = (p.ncolor / C), sp = s - P, c = sp / p.RS, rr = sp - c * p.RS;
            const int r = rr + ((p.RS < s2 && rr >= s2 / 2) ? 1 : 0);
            k = 2 * P + (c + (lane >> 5) * (C / 2)) * s2 + r;
        }
        int neuron = tile_row_to_neuron(lane & 31) + 32 * t;
        if (k < net.F) v = params[(int64_t)neuron * net.F + k];
    } else if (e < p.off_wl) {  // hidden layers 1..nl-1: [l-1][half][64][NT]
        int q = e - p.off_wh;
        int t = q % NT, lane = (q / NT) % 64, s = (q / (NT * 64)) % half, l = 1 + q / (NT * 64 * half);
        int k = 2 * s + (lane >> 5);
        int neuron = tile_row_to_neuron(lane & 31) + 32 * t;
        int64_t base = (int64_t)net.bc * net.F + net.bc + (int64_t)(l - 1) * ((int64_t)net.bc * net.bc + net.bc);
        v = params[base + (int64_t)neuron * net.bc + k];
    } else if (e < p.off_bh) {  // last layer: [half][64], rows = channels (identity order)
        int q = e - p.off_wl;
        int lane = q % 64, s = q / 64;
        int k = 2 * s + (lane >> 5), ch = lane & 31;
        int64_t base = (int64_t)net.bc * net.F + net.bc + (int64_t)(net.nl - 1) * ((int64_t)net.bc * net.bc + net.bc);
        if (ch < net.C) v = params[base + (int64_t)ch * net.bc + k];
    } else if (e < p.off_bl) {  // hidden biases: [l][t][h][16]
        int q = e - p.off_bh;
        int r = q % 16, h = (q / 16) % 2, t = (q / 32) % NT, l = q / (32 * NT);
        int neuron = 2 * r + h + 32 * t;
        int64_t wbase = l == 0 ? 0 : (int64_t)net.bc * net.F + net.bc + (int64_t)(l - 1) * ((int64_t)net.bc * net.bc + net.bc);
        int64_t nin = l == 0 ? net.F : net.bc;
        v = params[wbase + (int64_t)net.bc * nin + neuron];
    } else {  // last bias: [h][16]
        int q = e - p.off_bl;
        int r = q % 16, h = q / 16;
        int ch = acc_row(r, h);
        int64_t base = (int64_t)net.bc * net.F + net.bc + (int64_t)(net.nl - 1) * ((int64_t)net.bc * net.bc + net.bc);
        if (ch < net.C) v = params[base + (int64_t)net.C * net.bc + ch];
    }
    out[e] = v;
}

// ------------------------------------------------------------------ the fused kernel

enum ApplyMode { MODE_DECODE = 0, MODE_EVAL = 1, MODE_EVAL_FAST = 2,   // _FAST: the evaluation pass in the tolerance arithmetic (lbdrn_math.hpp)
                 MODE_EVAL_X16 = 3 };   // _FAST with the colour features of layer 0 on the f16 matrix pipe, exact operands (opt-in)
__host__ __device__ constexpr bool mode_fast(int mode) { return mode == MODE_EVAL_FAST || mode == MODE_EVAL_X16; }

struct ApplyArgs {
    lbdrn_geom g;
    lbdrn_net net;
    ApplyPlan p;
    const float* packed;
    const uint16_t* msb;
    const uint16_t* img;  // EVAL: original image (labels)
    uint16_t* out;        // DECODE
    float* y_out;         // DECODE, optional
    double* partial;      // EVAL: [nvirt]
    int nvirt;            // virtual workgroups: virtual workgroup v takes tiles v, v + nvirt, .. and owns partial[v]; a
                          // launch of fewer real workgroups walks them in turn -- same sums bit for bit on any grid
    unsigned long long* stamps;  // diagnostic build (-DLBDRN_APPLY_STAMPS): [gridDim.x][8] cycle sums of wave 0
};

#ifdef LBDRN_APPLY_STAMPS
#define ASTAMP(k)                                                                     \
    do {                                                                              \
        unsigned long long t_;                                                        \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                            \
        acc_t[k] += t_ - last_t;                                                      \
        last_t = t_;                                                                  \
    } while (0)
#else
#define ASTAMP(k)
#endif

template <int NT>
__device__ __forceinline__ void load_a(const float* w, int idx, float (&a)[NT])
{
    if constexpr (NT == 1) {
        a[0] = w[idx];
    } else if constexpr (NT == 2) {
        float2 v = *reinterpret_cast<const float2*>(w + (size_t)idx * 2);
        a[0] = v.x; a[1] = v.y;
    } else {
        float4 v = *reinterpret_cast<const float4*>(w + (size_t)idx * 4);
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    }
}

// Layer 0 of the tolerance-arithmetic evaluation pass over the colour features, in channel pairs (ApplyPlan::pair): lane
// half h walks the window of channel c + h C/2, so one per-lane base address serves every step of a pair and what
// changes from step to step is a compile-time offset: per step ONE LDS read of the window (the canonical loop reads an
// offset pair out of a table, the neighbour AND the centre: three reads), the centre of the pair's channel is read once
// and held in a register, and the window centre itself -- centre minus centre, an exact zero -- is not a step.
// wc: the pair's first fragment (w0 + ((P + c RS) 64 + lane) NT); tp: tile + c plane + (h C/2) plane + ly SW + lx + j.
template <int NT, int D, bool REL>
__device__ __forceinline__ void layer0_pair(f32x16 (&acc)[NT], const float* wc, const float* tp)
{
    constexpr int side = 2 * D + 1, S2 = side * side, CEN = D * side + D, SW = TILE_W + 2 * D, RS = S2 - (REL ? 1 : 0);
    const float cen = tp[D * SW + D];
    float bq[RS];
#pragma unroll
    for (int r = 0; r < S2; ++r) {
        if (REL && r == CEN) continue;
        const float nb = tp[(r / side) * SW + (r % side)];
        bq[r - ((REL && r > CEN) ? 1 : 0)] = REL ? nb - cen : nb;   // minus centre, LBDRNdataset.py:126-128
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < RS; ++k) {
        float a[NT];
        load_a<NT>(wc, k * 64, a);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt], bq[k], acc[tt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Layer 0 over the colour features on the f16 matrix pipe, operands EXACT (MODE_EVAL_X16; ApplyPlan::x16).  A relative colour
// feature is (msb_nbr - msb_ctr) / max: the numerator an integer of at most 11 bits -- exactly an fp16 --, and a float32
// weight is exactly the sum of three fp16 pieces once a power of two has brought it into range (k_pack_apply).  So
//     sum_k W[n][k] x[k]  =  (2^-s / max)  sum_pieces sum_k piece[n][k] * int[k]
// with every product exact (11 x 11 bits) and the sums taken in float32 by v_mfma_f32_32x32x16_f16: the same real numbers as
// the float32 MFMA of layer0_pair, with fewer roundings (the reference rounds nbr / max, ctr / max and their difference; this
// rounds none of them), at 3/16 of its matrix-pipe cycles.  Lane half h walks channel c + h C/2 as in layer0_pair; one MFMA
// takes eight window positions of both channels (k = 8 h + i).  tp: the STAGED INTEGERS of this lane's pixel (tile holds
// float(msb), not msb / max, in this mode); wc: the pair's first fragment + lane NT 4 floats.
template <int NT, int D>
__device__ __forceinline__ void layer0_pair_x16(f32x16 (&acc)[NT], const float* wc, const float* tp)
{
    constexpr int side = 2 * D + 1, S2 = side * side, CEN = D * side + D, SW = TILE_W + 2 * D, RS = S2 - 1, M = RS / 8;
    static_assert(RS % 8 == 0, "(2D+1)^2 - 1 = 4 D (D+1)");
    const float cen = tp[D * SW + D];
    float bq[RS];
#pragma unroll
    for (int r = 0; r < S2; ++r) {
        if (r == CEN) continue;
        bq[r - (r > CEN ? 1 : 0)] = tp[(r / side) * SW + (r % side)] - cen;     // an integer in [-2047, 2047]: exact
    }
    f16x8 b[M];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x2 pk = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(bq[8 * m + 2 * j], bq[8 * m + 2 * j + 1]));   // (exact: integers)
            b[m][2 * j] = pk[0];
            b[m][2 * j + 1] = pk[1];
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int pc = 0; pc < X16_PIECES; ++pc) {
            const float4* ap = reinterpret_cast<const float4*>(wc + (size_t)((m * X16_PIECES + pc) * 64) * NT * 4);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ap[tt]), b[m], acc[tt], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
}

// hidden activation of one accumulator entry.  RELU: z where z > 0, else 0 (torch.nn.ReLU: the same in every arithmetic)
template <int MODE, bool RELU>
__device__ __forceinline__ float apply_hidden_act(float z)
{
    if constexpr (RELU) return z > 0.0f ? z : 0.0f;
    else return mode_fast(MODE) ? fast_sin(30.0f * z) : siren_act(z);
}

template <int NT, int MODE, bool RELU>
__global__ void __launch_bounds__(APPLY_THREADS) k_apply_mfma(ApplyArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const lbdrn_geom& g = A.g;
    const ApplyPlan& p = A.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int half = A.net.bc / 2;
    const int P = g.P, D = g.D, C = g.C;

    // ---- once per workgroup: weights (fragment order) and the colour offset table into LDS
    for (int e = tid * 4; e < p.pack_floats; e += APPLY_THREADS * 4) {
        if (e + 3 < p.pack_floats) {
            *reinterpret_cast<float4*>(lds + e) = *reinterpret_cast<const float4*>(A.packed + e);
        } else {
            for (int q = e; q < p.pack_floats; ++q) lds[q] = A.packed[q];
        }
    }
    int* ktab = reinterpret_cast<int*>(lds + p.off_ktab);
    {
        const int side = 2 * D + 1, plane = p.SH * p.SW;
        for (int k = tid; k < 2 * (p.S0 - P) + 1; k += APPLY_THREADS) {
            int kk = min(k, p.ncolor - 1);  // pad entries repeat the last feature (their weight is 0)
            kk = max(kk, 0);
            int c = kk / (side * side), r = kk - c * side * side;
            int dy = r / side, dx = r - dy * side;
            ktab[2 * k] = c * plane + (dy - D) * p.SW + (dx - D);
            ktab[2 * k + 1] = c * plane;
        }
    }
    const float* w0 = lds + p.off_w0;
    const float* wh = lds + p.off_wh;
    const float* wl = lds + p.off_wl;
    const float* bh = lds + p.off_bh;
    const float* bl = lds + p.off_bl;
    float* rowt = lds + p.off_rowt;
    float* colt = lds + p.off_colt;
    float* tile = lds + p.off_tile;
    const bool rel = g.relative && D > 0;
    const float maxf = (float)g.msb_max;
    const int64_t HW = (int64_t)g.H * g.W;
    const float scale = (float)((1 << g.K) - 1);
    const int lmask = (1 << g.K) - 1;
#ifdef LBDRN_APPLY_STAMPS
    unsigned long long acc_t[8] = {}, last_t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_t)::"memory");
#endif
    const int ntiles = p.tiles_x * p.tiles_y;
    for (int v = blockIdx.x; v < A.nvirt; v += gridDim.x) {
    double sse = 0.0;
    for (int t = v; t < ntiles; t += A.nvirt) {
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int y0 = ty * p.TH, x0 = tx * TILE_W;
        __syncthreads();  // previous tile fully consumed (and the weight copy landed)
        // ---- stage the (TH+2D) x (64+2D) ring: p = float(msb)/max, reflect-padded (LBDRNdataset.py:120-123)
        if (g.use_colors) {
            // element e = (c, sy, sx) in LDS order, consecutive lanes = consecutive columns of one row, so a
            // wave load is one or two full 128-byte lines of the uint16 plane.  Eight loads are issued
            // before the first division (a load -> divide -> store loop paid the memory latency once per
            // element: 27 % of the kernel).  Index split by multiply-high (exact for e < 2^16, SW,SH < 2^8).
            const int plane = p.SH * p.SW, total = C * plane;
            const unsigned msw = 0xFFFFFFFFu / (unsigned)p.SW + 1u, msh = 0xFFFFFFFFu / (unsigned)p.SH + 1u;
            // (SH == 1 would need the multiplier 2^32: such single-row tiles take the dividing path)
            const bool simple = D < g.H && D < g.W && total < 65536 && p.SH > 1;
            constexpr int UB = 8;
            for (int base = 0; base < total; base += APPLY_THREADS * UB) {
                unsigned short raw[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int e = min(base + u * APPLY_THREADS + tid, total - 1);
                    int row, sx, c, sy, yy, xx;
                    if (simple) {
                        row = (int)__umulhi((unsigned)e, msw); sx = e - row * p.SW;
                        c = (int)__umulhi((unsigned)row, msh); sy = row - c * p.SH;
                        yy = y0 + sy - D; xx = x0 + sx - D;
                        yy = yy < 0 ? -yy : yy; yy = yy >= g.H ? 2 * (g.H - 1) - yy : yy;
                        xx = xx < 0 ? -xx : xx; xx = xx >= g.W ? 2 * (g.W - 1) - xx : xx;
                        yy = max(0, min(yy, g.H - 1));  // rows / columns past a ragged last tile feed only
                        xx = max(0, min(xx, g.W - 1));  // masked outputs: any valid pixel will do
                    } else {
                        row = e / p.SW; sx = e - row * p.SW; c = row / p.SH; sy = row - c * p.SH;
                        yy = reflect_idx(y0 + sy - D, g.H); xx = reflect_idx(x0 + sx - D, g.W);
                    }
                    // (the fast evaluation pass is internal to a fit, whose MSB plane IS img >> K: one plane read per pass
                    //  instead of two; the API's canonical pass and the decode pass read the plane they are given)
                    raw[u] = mode_fast(MODE) ? (unsigned short)(A.img[(int64_t)c * HW + (int64_t)yy * g.W + xx] >> g.K)
                                             : A.msb[(int64_t)c * HW + (int64_t)yy * g.W + xx];
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int e = base + u * APPLY_THREADS + tid;
                    if (e < total) tile[e] = MODE == MODE_EVAL_X16 ? (float)raw[u] : (float)raw[u] / maxf;   // (x16: the integers themselves)
                }
            }
        }
        for (int e = tid; e < p.TH * P; e += APPLY_THREADS) {
            int yy = min(y0 + e / P, g.H - 1);
            rowt[e] = g.rowtab[(int64_t)yy * P + e % P];
        }
        for (int e = tid; e < TILE_W * P; e += APPLY_THREADS) {
            int xx = min(x0 + e / P, g.W - 1);
            colt[e] = g.coltab[(int64_t)xx * P + e % P];
        }
        __syncthreads();

        ASTAMP(0);  // staging + barriers
        const int nseg = p.TH * (TILE_W / 32);
        for (int seg = wave; seg < nseg; seg += APPLY_WAVES) {
            const int ly = seg / (TILE_W / 32), lx = (seg % (TILE_W / 32)) * 32;
            const int pixbase = (ly + D) * p.SW + (lx + j + D);
            // what the epilogue needs from HBM (the label source or the MSB to rebuild) is requested now, so
            // that its latency hides behind the layer-0 MFMAs instead of ending the block
            const int yy = y0 + ly, xx = x0 + lx + j;
            const bool inside = yy < g.H && xx < g.W;
            const int64_t pix = (int64_t)min(yy, g.H - 1) * g.W + min(xx, g.W - 1);
            const int nreg = 4 * ((C + 7) / 8);  // accumulator registers that hold real channels
            unsigned short pre[16];
            {
                const uint16_t* src = (MODE == MODE_DECODE) ? A.msb : A.img;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pre[r] = 0;
                    if (r < nreg) pre[r] = src[(int64_t)min(acc_row(r, h), C - 1) * HW + pix];
                }
            }
            f32x16 acc[NT];
            f32x16 hid[NT];
            // ---- layer 0: C operand = bias, then k = 0..F-1 in order
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    float4 b4 = *reinterpret_cast<const float4*>(bh + (tt * 2 + h) * 16 + q4 * 4);
                    acc[tt][q4 * 4 + 0] = b4.x; acc[tt][q4 * 4 + 1] = b4.y;
                    acc[tt][q4 * 4 + 2] = b4.z; acc[tt][q4 * 4 + 3] = b4.w;
                }
            if constexpr (MODE == MODE_EVAL_X16) {
                // exact-operand f16 layer 0 (the plan guarantees P == 0, relative colours, D in 1..3): the accumulator starts at
                // zero, collects sum (w 2^s) x integer, and becomes the pre-activation as fma(acc, 2^-s / max, bias)
                f32x16 bias0[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    bias0[tt] = acc[tt];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
                }
                const int plane = p.SH * p.SW;
                const float* tp = tile + h * (C / 2) * plane + ly * p.SW + lx + j;
                const float* wc = w0 + (size_t)lane * NT * 4;
                const size_t wstep = (size_t)p.M16 * X16_PIECES * 64 * NT * 4;
                for (int c = 0; c < C / 2; ++c, tp += plane, wc += wstep) {
                    if (D == 1) layer0_pair_x16<NT, 1>(acc, wc, tp);
                    else if (D == 2) layer0_pair_x16<NT, 2>(acc, wc, tp);
                    else layer0_pair_x16<NT, 3>(acc, wc, tp);
                }
                const float cs = lds[p.off_scale];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tt][r] = fmaf(acc[tt][r], cs, bias0[tt][r]);
            }
            for (int s = 0; s < (MODE == MODE_EVAL_X16 ? 0 : P); ++s) {  // positional features, k = 2s+h < 2P
                int k = 2 * s + h;
                float b = (k < P) ? rowt[ly * P + k] : colt[(lx + j) * P + (k - P)];
                float a[NT];
                load_a<NT>(w0, s * 64 + lane, a);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt], b, acc[tt], 0, 0, 0);
            }
            bool paired = MODE == MODE_EVAL_X16;
            if constexpr (MODE == MODE_EVAL_FAST) {
                if (p.pair) {
                    paired = true;
                    const int plane = p.SH * p.SW;
                    const float* tp = tile + h * (C / 2) * plane + ly * p.SW + lx + j;
                    const float* wc = w0 + ((size_t)P * 64 + lane) * NT;
                    for (int c = 0; c < C / 2; ++c, tp += plane, wc += (size_t)p.RS * 64 * NT) {
                        if (rel) {
                            if (D == 1) layer0_pair<NT, 1, true>(acc, wc, tp);
                            else if (D == 2) layer0_pair<NT, 2, true>(acc, wc, tp);
                            else layer0_pair<NT, 3, true>(acc, wc, tp);
                        } else {
                            if (D == 1) layer0_pair<NT, 1, false>(acc, wc, tp);
                            else if (D == 2) layer0_pair<NT, 2, false>(acc, wc, tp);
                            else layer0_pair<NT, 3, false>(acc, wc, tp);
                        }
                    }
                }
            }
            if (!paired) {
#if LBDRN_APPLY_CHUNK > 0
            // colour features in chunks of CH steps: the chunk's B operands (window gather, minus centre) are all made
            // first -- vector and LDS work only --, then its 2 CH MFMAs run with nothing but their A-operand reads
            // between them: an f32 MFMA and vector work of the SAME wave never overlap, those of the two waves
            // sharing a SIMD do once their streams are not both a fine mix of the two (scripts/pipe_probe.hip)
            constexpr int CH = LBDRN_APPLY_CHUNK;
            static_assert(CH % 4 == 0, "the steps left behind the chunks are taken four at a time");
            int s4 = P;
            for (; s4 + CH <= p.S0; s4 += CH) {
                float bq[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int kk = 2 * (s4 + u - P) + h;
                    const int2 e = *reinterpret_cast<const int2*>(ktab + 2 * kk);
                    const float nb = tile[pixbase + e.x];
                    const float ct = tile[pixbase + e.y];
                    bq[u] = rel ? nb - ct : nb;  // minus centre, LBDRNdataset.py:126-128
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    float a[NT];
                    load_a<NT>(w0, (s4 + u) * 64 + lane, a);
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt], bq[u], acc[tt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            for (; s4 < p.S0; s4 += 4) {
#else
            for (int s4 = P; s4 < p.S0; s4 += 4) {  // colour features, four MFMA steps per trip
#endif
                float bq[4];
                float aq[4][NT];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int kk = 2 * (s4 + u - P) + h;
                    int2 e = *reinterpret_cast<const int2*>(ktab + 2 * kk);
                    float nb = tile[pixbase + e.x];
                    float ct = tile[pixbase + e.y];
                    bq[u] = rel ? nb - ct : nb;  // minus centre, LBDRNdataset.py:126-128
                    load_a<NT>(w0, (s4 + u) * 64 + lane, aq[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u][tt], bq[u], acc[tt], 0, 0, 0);
            }
            }   // !paired
            ASTAMP(1);  // layer 0
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) hid[tt][r] = apply_hidden_act<MODE, RELU>(acc[tt][r]);
            ASTAMP(2);  // sin
            // ---- hidden layers 1..nl-1: B operand = previous activations, straight from registers
            for (int l = 1; l < A.net.nl; ++l) {
                const float* wlayer = wh + (size_t)(l - 1) * half * 64 * NT;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        float4 b4 = *reinterpret_cast<const float4*>(bh + ((l * NT + tt) * 2 + h) * 16 + q4 * 4);
                        acc[tt][q4 * 4 + 0] = b4.x; acc[tt][q4 * 4 + 1] = b4.y;
                        acc[tt][q4 * 4 + 2] = b4.z; acc[tt][q4 * 4 + 3] = b4.w;
                    }
#pragma unroll
                for (int s = 0; s < NT * 16; ++s) {
                    float b = hid[s / 16][s % 16];
                    float a[NT];
                    load_a<NT>(wlayer, s * 64 + lane, a);
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt], b, acc[tt], 0, 0, 0);
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hid[tt][r] = apply_hidden_act<MODE, RELU>(acc[tt][r]);
            }
            ASTAMP(3);  // hidden layers incl. their sin
            // ---- last layer: rows = channels
            f32x16 o;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                float4 b4 = *reinterpret_cast<const float4*>(bl + h * 16 + q4 * 4);
                o[q4 * 4 + 0] = b4.x; o[q4 * 4 + 1] = b4.y; o[q4 * 4 + 2] = b4.z; o[q4 * 4 + 3] = b4.w;
            }
#pragma unroll
            for (int s = 0; s < NT * 16; ++s) {
                float b = hid[s / 16][s % 16];
                float a = wl[s * 64 + lane];
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, o, 0, 0, 0);
            }
            ASTAMP(4);  // last layer
            // ---- epilogue
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = acc_row(r, h);
                if (r < nreg && ch < C && inside) {
                    float yv = mode_fast(MODE) ? fast_sigmoid(o[r]) : canon_sigmoid(o[r]);
                    if (MODE == MODE_DECODE) {
                        float rr = __builtin_rintf(yv * scale);  // torch.round, decode.py:131
                        int base = (int)pre[r] << g.K;             // decode.py:134
                        A.out[(int64_t)ch * HW + pix] = (uint16_t)(base + (int)rr);
                        if (A.y_out) A.y_out[pix * C + ch] = yv;
                    } else {
                        float lab = (float)((int)pre[r] & lmask) / scale;
                        float d = yv - lab;
                        sse += (double)(d * d);
                    }
                }
            }
            ASTAMP(5);  // epilogue
        }
    }
    if (MODE != MODE_DECODE) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds + ((p.off_rowt + 3) & ~3));  // the per-tile regions: consumed
        red[tid] = sse;
        __syncthreads();
        for (int o2 = APPLY_THREADS / 2; o2 > 0; o2 >>= 1) {
            if (tid < o2) red[tid] += red[tid + o2];
            __syncthreads();
        }
        if (tid == 0) A.partial[v] = red[0];
    }
    }
#ifdef LBDRN_APPLY_STAMPS
    if (tid == 0 && A.stamps)
        for (int k = 0; k < 8; ++k) A.stamps[(size_t)blockIdx.x * 8 + k] = acc_t[k];
#endif
}

// one wave: lane l sums partial[l], partial[l + 64], .. in index order, then the 64 lane sums pairwise in a fixed tree
// (the same value every run; a single thread walking 256 partials took 13 us)
__global__ void k_sum_partials_mfma(const double* __restrict__ partial, int n, double* dst)
{
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) *dst = s;
}

template <int NT, int MODE, bool RELU>
static int launch_apply_act(const ApplyArgs& A, int grid, hipStream_t s, bool whole_lds)
{
    // whole_lds (a fit's background evaluation pass): the workgroup asks for 125 of the CU's 128 LDS granules of 1,280
    // bytes whatever it needs, so that the k_reduce_adam launches of the training chain beside it (4 granules) are
    // placed on the CUs the chain's own training step has just left, not next to this pass's waves -- there they took
    // 14.4 instead of 4.9 us (kernel trace of a fit alone, scripts/lone_timeline.py)
    const size_t lds_bytes = std::max((size_t)A.p.lds_floats * 4, whole_lds ? (size_t)125 * 1280 : (size_t)0);
    auto kern = k_apply_mfma<NT, MODE, RELU>;
    // the kernel may use the whole 160 KB of a CU's LDS: told to the runtime once per device and kernel (a cache of an
    // idempotent setting, not state a caller can observe)
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(configured.load(std::memory_order_relaxed) & bit)) {
        LBDRN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    kern<<<grid, APPLY_THREADS, lds_bytes, s>>>(A);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

template <int NT, int MODE>
static int launch_apply(const ApplyArgs& A, int grid, hipStream_t s, bool whole_lds = false)
{
    return A.net.act == LBDRN_ACT_RELU ? launch_apply_act<NT, MODE, true>(A, grid, s, whole_lds)
                                       : launch_apply_act<NT, MODE, false>(A, grid, s, whole_lds);
}

#include "apply_wide.inc"

static bool wapply_supported(const lbdrn_geom& g, const lbdrn_net& net)
{
    WApplyPlan p;
    return make_wapply_plan(g, net, &p);
}
static size_t wapply_workspace(const lbdrn_geom& g, const lbdrn_net& net)
{
    WApplyPlan p;
    if (!make_wapply_plan(g, net, &p)) return 0;
    return align_up((size_t)p.pack_floats * 4, 256) + align_up(MFMA_PARTIALS * sizeof(double), 256);
}

template <int NT, int NL, int MODE>
static int launch_wapply(const WApplyArgs& A, int grid, hipStream_t s, bool whole_lds)
{
    // whole_lds: a fit's background pass claims its CU's LDS like k_apply_mfma's does (launch_apply), so that the reduce
    // launches of the training chain beside it land on the CUs the chain's own step has just left
    const size_t lds_bytes = std::max((size_t)A.p.lds_floats * 4, whole_lds ? (size_t)125 * 1280 : (size_t)0);
    auto kern = k_apply_wide<NT, NL, MODE>;
    if (whole_lds) {
        static std::atomic<unsigned long long> configured{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(configured.load(std::memory_order_relaxed) & bit)) {
            LBDRN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            configured.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    kern<<<grid, WA_THREADS, lds_bytes, s>>>(A);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

template <int MODE>
static int dispatch_wapply(const WApplyArgs& A, int grid, hipStream_t s, bool whole_lds)
{
    const bool one = A.net.nl == 1;
    if (A.p.NT == 8) return one ? launch_wapply<8, 1, MODE>(A, grid, s, whole_lds) : launch_wapply<8, 2, MODE>(A, grid, s, whole_lds);
    return one ? launch_wapply<16, 1, MODE>(A, grid, s, whole_lds) : launch_wapply<16, 2, MODE>(A, grid, s, whole_lds);
}

static int run_wapply(const lbdrn_geom& g, const lbdrn_net& net, int mode, const uint16_t* img,
                      const uint16_t* msb, const float* params, uint16_t* out, float* y_out,
                      double* sse, void* ws, size_t ws_bytes, bool background, hipStream_t s)
{
    WApplyArgs A;
    if (mode == MODE_EVAL_X16) mode = MODE_EVAL_FAST;     // (the streaming kernel has no exact-operand layer 0)
    if (!make_wapply_plan(g, net, &A.p, mode == MODE_EVAL_FAST)) {
        set_error("shape not supported by the MFMA apply kernels");
        return LBDRN_E_UNSUPPORTED;
    }
    if (!ws || ws_bytes < wapply_workspace(g, net)) {
        set_error("apply workspace too small: %zu < %zu", ws_bytes, wapply_workspace(g, net));
        return LBDRN_E_WORKSPACE;
    }
    float* packed = (float*)ws;
    double* partial = (double*)((char*)ws + align_up((size_t)A.p.pack_floats * 4, 256));
    k_pack_apply_wide<<<(A.p.pack_floats + 255) / 256, 256, 0, s>>>(params, net, A.p, 2 * g.P, packed);
    LBDRN_LAUNCH_CHECK();
    A.g = g; A.net = net; A.packed = packed; A.msb = msb; A.img = img; A.out = out; A.y_out = y_out;
    A.partial = partial;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    A.nvirt = std::min(std::min(A.p.tiles_x * A.p.tiles_y, cus), MFMA_PARTIALS);
    const int grid = background ? std::max(1, A.nvirt / 2) : A.nvirt;
    A.stamps = nullptr;
#ifdef LBDRN_APPLY_STAMPS
    static unsigned long long* wstamp_buf = nullptr;
    if (!wstamp_buf) LBDRN_HIP_TRY(hipMalloc(&wstamp_buf, 1024 * 8 * sizeof(unsigned long long)));
    A.stamps = wstamp_buf;
#endif
    const bool claim = background;   // (bc = 256, one tile alone: 569.5 -> 558.5 ms)
    int rc = mode == MODE_DECODE ? dispatch_wapply<MODE_DECODE>(A, grid, s, false)
           : mode == MODE_EVAL_FAST ? dispatch_wapply<MODE_EVAL_FAST>(A, grid, s, claim) : dispatch_wapply<MODE_EVAL>(A, grid, s, claim);
    if (rc) return rc;
    if (mode != MODE_DECODE) {
        k_sum_partials_mfma<<<1, 64, 0, s>>>(partial, A.nvirt, sse);
        LBDRN_LAUNCH_CHECK();
    }
#ifdef LBDRN_APPLY_STAMPS
    {
        LBDRN_HIP_TRY(hipStreamSynchronize(s));
        unsigned long long h[1024 * 8];
        LBDRN_HIP_TRY(hipMemcpy(h, A.stamps, (size_t)grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum[8] = {};
        for (int b = 0; b < grid; ++b) for (int k = 0; k < 8; ++k) sum[k] += (double)h[b * 8 + k];
        const double segs = (double)A.p.tiles_x * A.p.tiles_y * A.p.TH * (TILE_W / 16) / 4 / grid;
        double rtmin = 1e30, rtmax = 0;
        for (int b = 0; b < grid; ++b) { rtmin = std::min(rtmin, (double)h[b * 8 + 6]); rtmax = std::max(rtmax, (double)h[b * 8 + 6]); }
        fprintf(stderr, "[lbdrn wide apply stamps] wave lifetime %.2f .. %.2f ms (mean %.2f); per wave: tiles-staging %.0f total; per 16-px segment: L0 %.0f sin %.0f L1 %.0f last %.0f epilogue %.0f\n",
                rtmin / 1e5, rtmax / 1e5, sum[6] / grid / 1e5, sum[0] / grid, sum[1] / grid / segs, sum[2] / grid / segs, sum[3] / grid / segs, sum[4] / grid / segs, sum[5] / grid / segs);
    }
#endif
    return 0;
}

static int run_apply(const lbdrn_geom& g, const lbdrn_net& net, int mode, const uint16_t* img,
                     const uint16_t* msb, const float* params, uint16_t* out, float* y_out,
                     double* sse, void* ws, size_t ws_bytes, bool background, hipStream_t s)
{
    ApplyArgs A;
    if (!make_plan(g, net, &A.p, mode_fast(mode), mode == MODE_EVAL_X16))   // too wide for LDS-resident weights: the streaming kernel (apply_wide.inc)
        return run_wapply(g, net, mode, img, msb, params, out, y_out, sse, ws, ws_bytes, background, s);
    if (mode == MODE_EVAL_X16 && !A.p.x16) mode = MODE_EVAL_FAST;    // (a hint: the shape or the image does not qualify, the pass is the fast one)
    if (!ws || ws_bytes < mfma_apply_workspace(g, net)) {
        set_error("apply workspace too small: %zu < %zu", ws_bytes, mfma_apply_workspace(g, net));
        return LBDRN_E_WORKSPACE;
    }
    float* packed = (float*)ws;
    const size_t pack_bytes = apply_pack_bytes(g, net);
    double* partial = (double*)((char*)ws + pack_bytes);
    float* amax = (float*)((char*)ws + pack_bytes + align_up(MFMA_PARTIALS * sizeof(double), 256));
    if (A.p.x16) {
        k_w0_absmax<<<1, 256, 0, s>>>(params, net.bc * net.F, amax);
        LBDRN_LAUNCH_CHECK();
    }
    k_pack_apply<<<(A.p.pack_floats + 255) / 256, 256, 0, s>>>(params, net, A.p, g.P, g.C, g.msb_max, amax, packed);
    LBDRN_LAUNCH_CHECK();
    A.g = g; A.net = net; A.packed = packed; A.msb = msb; A.img = img; A.out = out; A.y_out = y_out;
    A.partial = partial;
    A.stamps = nullptr;
#ifdef LBDRN_APPLY_STAMPS
    static unsigned long long* stamp_buf = nullptr;
    if (!stamp_buf) LBDRN_HIP_TRY(hipMalloc(&stamp_buf, 1024 * 8 * sizeof(unsigned long long)));
    A.stamps = stamp_buf;
#endif
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    A.nvirt = std::min(std::min(A.p.tiles_x * A.p.tiles_y, cus), MFMA_PARTIALS);
    // a background pass (beside a training stream that holds the other half of the CUs) launches half as many
    // workgroups, each walking two virtual ones
    const int grid = background ? std::max(1, A.nvirt / 2) : A.nvirt;   // (a quarter / a sixth: 3 / 10 ms per tile slower)
    int rc;
    if (mode == MODE_DECODE) {
        rc = A.p.NT == 1 ? launch_apply<1, MODE_DECODE>(A, grid, s)
           : A.p.NT == 2 ? launch_apply<2, MODE_DECODE>(A, grid, s)
                         : launch_apply<4, MODE_DECODE>(A, grid, s);
    } else {
        rc = mode == MODE_EVAL_X16
                 ? (A.p.NT == 1 ? launch_apply<1, MODE_EVAL_X16>(A, grid, s, background) : launch_apply<2, MODE_EVAL_X16>(A, grid, s, background))
             : mode == MODE_EVAL_FAST
                 ? (A.p.NT == 1 ? launch_apply<1, MODE_EVAL_FAST>(A, grid, s, background)
                    : A.p.NT == 2 ? launch_apply<2, MODE_EVAL_FAST>(A, grid, s, background)
                                  // bc = 128: the tolerance arithmetic's extra operands spill at two waves per SIMD (88 registers,
                                  // 356 B of scratch when it was instantiated); the canonical pass IS within the 1e-6 the flag promises
                                  : launch_apply<4, MODE_EVAL>(A, grid, s, background))
                 : (A.p.NT == 1 ? launch_apply<1, MODE_EVAL>(A, grid, s, background)
                    : A.p.NT == 2 ? launch_apply<2, MODE_EVAL>(A, grid, s, background)
                                  : launch_apply<4, MODE_EVAL>(A, grid, s, background));
        if (rc) return rc;
        k_sum_partials_mfma<<<1, 64, 0, s>>>(partial, A.nvirt, sse);
        LBDRN_LAUNCH_CHECK();
    }
#ifdef LBDRN_APPLY_STAMPS
    if (rc == 0) {
        LBDRN_HIP_TRY(hipStreamSynchronize(s));
        unsigned long long h[1024 * 8];
        LBDRN_HIP_TRY(hipMemcpy(h, A.stamps, (size_t)grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum[8] = {};
        for (int b = 0; b < grid; ++b) for (int k = 0; k < 8; ++k) sum[k] += (double)h[b * 8 + k];
        const double blocks = (double)A.p.tiles_x * A.p.tiles_y * A.p.TH * (TILE_W / 32) / APPLY_WAVES / grid;
        fprintf(stderr, "[lbdrn apply stamps] per wave: tiles-staging %.0f total; per 32-px block: L0 %.0f sin %.0f hidden %.0f last %.0f epilogue %.0f\n",
                sum[0] / grid, sum[1] / grid / blocks, sum[2] / grid / blocks, sum[3] / grid / blocks, sum[4] / grid / blocks, sum[5] / grid / blocks);
    }
#endif
    return rc;
}

int mfma_decode(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* msb, const float* params,
                uint16_t* out, float* y_out, void* ws, size_t ws_bytes, hipStream_t s)
{
    return run_apply(g, net, MODE_DECODE, nullptr, msb, params, out, y_out, nullptr, ws, ws_bytes, false, s);
}

int mfma_eval_sse(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                  const uint16_t* msb, const float* params, double* sse, void* ws, size_t ws_bytes,
                  bool background, bool fast, bool x16, hipStream_t s)
{
    return run_apply(g, net, fast ? (x16 ? MODE_EVAL_X16 : MODE_EVAL_FAST) : MODE_EVAL, img, msb, params, nullptr, nullptr, sse, ws, ws_bytes,
                     background, s);
}

}  // namespace lbdrn
