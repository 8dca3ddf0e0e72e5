// Fused training step for gfx950: one launch per minibatch does gather -> forward -> loss ->
// backward -> per-workgroup gradient slab, a second launch reduces the slabs in a fixed order,
// applies torch's Adam update and refreshes the MFMA-ordered copy of the weights
// (ref modified_ignite_engine.py:18-27, encode.py:84; LBDRNdataset.py:136-155 for the gather).
//
// Why it looks like this on MI355X
//  * a minibatch is 8192 rows and 646 MFLOP: 4.1 us of the whole chip at the f32 MFMA peak.  The
//    step is a serial chain (the next step needs the updated weights), so it is latency-bound: the
//    batch is cut into 256 workgroups of 32 samples -- one per CU -- and every layer of those 32
//    samples is split over the CU's 4 SIMDs: wave w owns hidden neurons 16w..16w+15 for both
//    16-sample column tiles (v_mfma_f32_16x16x4_f32, two independent accumulators per wave).
//  * activations and their gradients sit in LDS as [sample][unit] with a row pitch = 2 (mod 32)
//    floats, which makes the forward B-operand reads conflict-free and the weight-gradient reads
//    2-way; weights are never staged in LDS: every A operand comes straight from L2 into VGPRs,
//    prefetched at kernel start (layer 0 in double-buffered chunks) in "fragment order" so that a
//    wave-wide load is one 256-byte line.  The kernel runs one wave per SIMD with up to 512 VGPRs.
//  * cos(30 z) needed by the backward pass is kept in registers: the wave that produced a tile of
//    z is the wave that later receives the matching tile of dL/dh, in the same lane/register slots.
//  * weight gradients are sums over the minibatch: each workgroup writes its partial (70 KB) to a
//    slab; k_reduce_adam adds the 256 slabs in index order (bitwise reproducible, no atomics).
//  * the minibatch gather reads whole rows of a [N][F+C] float32 matrix (features | labels) that
//    lbdrn_train_prepare materialises once per image: 3.5 GB for a 2048^2x8 tile, nothing on a
//    288 GB part, and a random 832-byte row is 7 full cache lines, where gathering the 5x5xC window
//    from a pixel plane moves 2-4x the bytes in partial lines (measured: 5.9 us of a 22 us step).
//    The evaluation / decode passes never use this matrix (they stage windows through LDS).
//  * each workgroup's gradient is assembled in LDS in parameter order and leaves as full 1 KB
//    wave-stores; per-tile strided dword stores cost 6.9 us of the step before.
//
// Training parity is a tolerance contract (1e-5 relative on the loss), not a bit
// pattern: the batch sum order differs from the generic kernels and from torch.
#include <cmath>
#include <vector>

#include "common.hpp"
#include "lbdrn_math.hpp"

namespace lbdrn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 32;       // samples per workgroup
constexpr int TBC = 64;      // hidden width this kernel is built for
constexpr int HPITCH = 66;   // [sample][64] activations, pitch = 2 (mod 32)
constexpr int OPITCH = 18;   // [sample][16] output-layer gradient
constexpr int RED_SLICES = 16;

struct TrainPlan {
    int CH;                    // layer-0 prefetch chunk (MFMA steps)
    int S0;                    // layer-0 MFMA steps, padded to a multiple of CH
    int FP;                    // X row pitch (floats), = 2 (mod 32), >= 4*S0 and >= 16*NT0
    int NT0;                   // 16-wide feature tiles of dW0 = ceil(F/16)
    int RP;                    // row pitch of the materialised [N][RP] matrix: F features, C labels, pad to x4
    int NPP;                   // slab pitch: NP padded to a multiple of 64 floats
    int side, ncolor;
    int64_t NP;
    int64_t offW[5], offB[5];  // canonical parameter offsets per layer (index nl = last layer)
    int pk_w0, pk_wh, pk_wl, pack_floats;  // fragment-order buffer (floats)
    int lds_x, lds_h, lds_z, lds_zo, lds_yx, lds_red, lds_g, lds_floats;
};

static bool make_train_plan(const lbdrn_geom& g, const lbdrn_net& net, TrainPlan* out)
{
    if (net.bc != TBC || net.nl < 1 || net.nl > 3 || net.C > 16 || net.F < 1) return false;
    TrainPlan p;
    const int s0 = (net.F + 3) / 4;
    int best_ch = 8, best_pad = 1 << 30;
    for (int ch : {10, 8}) {
        int pad = (s0 + ch - 1) / ch * ch;
        if (pad < best_pad) { best_pad = pad; best_ch = ch; }
    }
    p.CH = best_ch;
    p.S0 = best_pad;
    p.NT0 = (net.F + 15) / 16;
    p.RP = (net.F + net.C + 3) / 4 * 4;
    int need = std::max(std::max(4 * p.S0, 16 * p.NT0), p.RP);
    p.FP = (need + 29) / 32 * 32 + 2;
    if (p.FP < need) p.FP += 32;
    p.side = 2 * g.D + 1;
    p.ncolor = net.F - 2 * g.P;
    if (p.ncolor < 0) return false;
    p.NP = param_count(net);
    p.NPP = (int)((p.NP + 63) / 64 * 64);
    int64_t o = 0;
    for (int l = 0; l < net.nl; ++l) {
        int nin = l ? TBC : net.F;
        p.offW[l] = o; o += (int64_t)TBC * nin;
        p.offB[l] = o; o += TBC;
    }
    p.offW[net.nl] = o; o += (int64_t)net.C * TBC;
    p.offB[net.nl] = o;
    int k = 0;
    p.pk_w0 = k; k += 4 * p.S0 * 64;
    p.pk_wh = k; k += (net.nl - 1) * 4 * 16 * 64;
    p.pk_wl = k; k += 16 * 64;
    p.pack_floats = k;
    int f = 0;
    p.lds_x = f; f += TB * p.FP;
    p.lds_h = f; f += net.nl * TB * HPITCH;
    p.lds_z = f; f += net.nl * TB * HPITCH;
    p.lds_zo = f; f += TB * OPITCH;
    p.lds_yx = f; f += 2 * TB;
    f = (f + 1) & ~1;
    p.lds_red = f; f += 16;
    f = (f + 3) & ~3;
    p.lds_g = f; f += p.NPP;
    p.lds_floats = f;
    if ((size_t)f * 4 > 160 * 1024) return false;
    *out = p;
    return true;
}

bool mfma_train_supported(const lbdrn_geom& g, const lbdrn_net& net)
{
    TrainPlan p;
    return make_train_plan(g, net, &p);
}

struct TrainWsLayout {
    size_t off_rows, off_pack, off_slab, off_loss, total;
};

static TrainWsLayout train_ws_layout(const lbdrn_geom& g, const lbdrn_net& net, const TrainPlan& p, int bs)
{
    TrainWsLayout L;
    size_t o = 0;
    L.off_rows = o; o += align_up((size_t)g.H * g.W * p.RP * sizeof(float), 256);
    L.off_pack = o; o += align_up((size_t)p.pack_floats * sizeof(float), 256);
    const size_t nwg = (size_t)(bs + TB - 1) / TB;
    L.off_slab = o; o += align_up(nwg * (size_t)p.NPP * sizeof(float), 256);
    L.off_loss = o; o += align_up(nwg * sizeof(double), 256);
    L.total = o;
    return L;
}

size_t mfma_train_workspace(const lbdrn_geom& g, const lbdrn_net& net, int bs)
{
    TrainPlan p;
    if (!make_train_plan(g, net, &p)) return 0;
    return train_ws_layout(g, net, p, bs).total;
}

// ------------------------------------------------------------------ helper kernels

// rows[n][0..F) = features of pixel n, rows[n][F..F+C) = labels, rest 0
// (ref LBDRNdataset.py:95-97, 104-131 -- the reference's own [N,F] / [N,C] matrices, side by side)
__global__ void __launch_bounds__(256)
    k_build_rows(lbdrn_geom g, int F, int RP, const uint16_t* __restrict__ msb,
                 const uint16_t* __restrict__ img, float* __restrict__ rows)
{
    const int64_t HW = (int64_t)g.H * g.W;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= HW * RP) return;
    const int64_t pix = e / RP;
    const int f = (int)(e - pix * RP);
    const int y = (int)(pix / g.W), x = (int)(pix - (int64_t)y * g.W);
    float v = 0.0f;
    if (f < g.P) {
        v = g.rowtab[(int64_t)y * g.P + f];
    } else if (f < 2 * g.P) {
        v = g.coltab[(int64_t)x * g.P + (f - g.P)];
    } else if (f < F) {
        const int side = 2 * g.D + 1;
        int cf = f - 2 * g.P;
        int c = cf / (side * side);
        int r = cf - c * side * side;
        int dy = r / side, dx = r - dy * side;
        const uint16_t* pl = msb + (int64_t)c * HW;
        const float mx = (float)g.msb_max;
        int yy = reflect_fast(y + dy - g.D, g.H), xx = reflect_fast(x + dx - g.D, g.W);
        v = (float)pl[(int64_t)yy * g.W + xx] / mx;
        if (g.relative && g.D > 0) v = v - (float)pl[pix] / mx;
    } else if (f < F + g.C) {
        const int mask = (1 << g.K) - 1;
        v = (float)((int)img[(int64_t)(f - F) * HW + pix] & mask) / (float)mask;
    }
    rows[e] = v;
}

// canonical parameter index -> position in the fragment-order buffer (or -1: not packed)
__device__ __forceinline__ int frag_pos(int64_t idx, const TrainPlan& p, int F, int nl, int C)
{
    if (idx < p.offB[0]) {  // W0[n][k]
        int n = (int)(idx / F), k = (int)(idx - (int64_t)n * F);
        return p.pk_w0 + (((n >> 4) * p.S0 + (k >> 2)) * 64 + (k & 3) * 16 + (n & 15));
    }
    for (int l = 1; l < nl; ++l) {
        if (idx >= p.offW[l] && idx < p.offB[l]) {  // W_l[n][k]
            int e = (int)(idx - p.offW[l]);
            int n = e >> 6, k = e & 63;
            return p.pk_wh + ((((l - 1) * 4 + (n >> 4)) * 16 + (k >> 2)) * 64 + (k & 3) * 16 + (n & 15));
        }
    }
    if (idx >= p.offW[nl] && idx < p.offB[nl]) {  // W_last[c][k]
        int e = (int)(idx - p.offW[nl]);
        int c = e >> 6, k = e & 63;
        return p.pk_wl + ((k >> 2) * 64 + (k & 3) * 16 + c);
    }
    return -1;
}

__global__ void __launch_bounds__(256)
    k_pack_train(const float* __restrict__ params, TrainPlan p, int F, int nl, int C, float* __restrict__ packed)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.NP) return;
    int pos = frag_pos(idx, p, F, nl, C);
    if (pos >= 0) packed[pos] = params[idx];
}

// g[idx] = sum over workgroups (index order within a slice, slices in order) of slab[wg][idx];
// torch Adam (torch/optim/adam.py single-tensor path: lerp_, mul_/addcmul_, addcdiv_); refresh the
// fragment copy.  Block = 16 float4 lanes (64 parameters) x 16 workgroup slices.
__global__ void __launch_bounds__(256)
    k_reduce_adam(const float* __restrict__ slabs, int nwg, TrainPlan p, int F, int nl, int C,
                  float* __restrict__ params, float* __restrict__ m, float* __restrict__ v,
                  float* __restrict__ packed, float step_size, float bc2_sqrt,
                  const double* __restrict__ loss_part, double loss_count, float* loss_out)
{
    __shared__ float4 part[RED_SLICES][16];
    const int l16 = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int64_t base = (int64_t)blockIdx.x * 64 + 4 * l16;
    const int per = (nwg + RED_SLICES - 1) / RED_SLICES;
    const int w0 = slice * per, w1 = min(nwg, w0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = slabs + base;
    int w = w0;
    for (; w + 8 <= w1; w += 8) {  // eight loads in flight, added in index order
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(src + (size_t)(w + u) * p.NPP);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
    }
    for (; w < w1; ++w) {
        float4 t = *reinterpret_cast<const float4*>(src + (size_t)w * p.NPP);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    part[slice][l16] = acc;
    __syncthreads();
    if (slice == 0) {
        float4 gsum = part[0][l16];
#pragma unroll
        for (int s = 1; s < RED_SLICES; ++s) {
            float4 t = part[s][l16];
            gsum.x += t.x; gsum.y += t.y; gsum.z += t.z; gsum.w += t.w;
        }
        const float gv[4] = {gsum.x, gsum.y, gsum.z, gsum.w};
        const float w1c = (float)(1.0 - 0.9), b2 = 0.999f, w2c = (float)(1.0 - 0.999), eps = 1e-8f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t idx = base + u;
            if (idx < p.NP) {
                const float g = gv[u];
                float mi = m[idx] + w1c * (g - m[idx]);
                float vi = v[idx] * b2 + w2c * (g * g);
                float denom = __builtin_sqrtf(vi) / bc2_sqrt + eps;
                float pi = params[idx] + (-step_size) * (mi / denom);
                m[idx] = mi;
                v[idx] = vi;
                params[idx] = pi;
                int pos = frag_pos(idx, p, F, nl, C);
                if (pos >= 0) packed[pos] = pi;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) {
        double s = 0.0;
        for (int k = 0; k < nwg; ++k) s += loss_part[k];
        *loss_out = (float)(s / loss_count);
    }
}

// ------------------------------------------------------------------ the fused step

struct TrainArgs {
    lbdrn_net net;
    TrainPlan p;
    const float* rows;      // [N][RP] features | labels
    int64_t npix;
    const int64_t* perm;    // this minibatch's pixel indices
    int batch_n;            // rows in this minibatch
    const float* params;    // canonical
    const float* packed;    // fragment order
    float* slabs;           // [nwg][NPP]
    double* loss_part;      // [nwg]
    float inv;              // 1 / (batch_n * C)
    unsigned long long* stamps;  // diagnostic build only (-DLBDRN_TRAIN_STAMPS): [nwg][12] s_memtime
};

#ifdef LBDRN_TRAIN_STAMPS
#define STAMP(k)                                                                        \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[k])::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#else
#define STAMP(k)
#endif

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TRAIN_THREADS = 512;  // 8 waves: (neuron tile w = 0..3) x (sample tile st = 0..1)

// Two waves share each SIMD.  A wave issues in order, and the ~130 non-MFMA instructions of a tile
// step (operand reads, address math, accumulator write-out) cannot hide behind its own MFMAs when
// it is alone on the SIMD (measured: 86 cycles per 32-cycle MFMA in the weight-gradient phase with
// 4 waves).  With the sample tiles split over two waves per SIMD, one wave's MFMAs cover the
// other's bookkeeping.
template <int CH, int NL>
__global__ void __launch_bounds__(TRAIN_THREADS, 2) k_train_mfma(TrainArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TrainPlan& p = A.p;
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6;
    const int w = w8 & 3, st = w8 >> 2;
    const int i = lane & 15, q = lane >> 4;
    const int C = A.net.C, F = A.net.F, FP = p.FP;
    float* Xs = lds + p.lds_x;
    float* Hs = lds + p.lds_h;
    float* Zs = lds + p.lds_z;
    float* Zo = lds + p.lds_zo;
    int* pixs = reinterpret_cast<int*>(lds + p.lds_yx);
    double* red = reinterpret_cast<double*>(lds + p.lds_red);
    float* Gs = lds + p.lds_g;
    const int wg = blockIdx.x;
    const int first = wg * TB;
    const int nvalid = min(TB, A.batch_n - first);
    const int srow = i + 16 * st;  // the sample (row of X / H / dZ) this lane's B operands come from
#ifdef LBDRN_TRAIN_STAMPS
    unsigned long long stamp[16] = {};
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[12])::"memory");
#endif
    STAMP(0);

    // the one load everything else waits for goes first, so that the wait for it leaves the weight
    // prefetch below in flight (vmcnt retires in issue order)
    int64_t mypix = 0;
    if (tid < TB) mypix = A.perm[first + min(tid, nvalid - 1)];

    // ---- weight prefetch: everything this wave will multiply by, L2 -> VGPR, before the gather
    const float* wf0 = A.packed + p.pk_w0 + (size_t)w * p.S0 * 64 + lane;
    float a0[2][CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) a0[0][u] = wf0[u * 64];
    float ah[NL > 1 ? NL - 1 : 1][16];
#pragma unroll
    for (int l = 1; l < NL; ++l)
#pragma unroll
        for (int s = 0; s < 16; ++s) ah[l - 1][s] = A.packed[p.pk_wh + (((l - 1) * 4 + w) * 16 + s) * 64 + lane];
    float al[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) al[s] = A.packed[p.pk_wl + s * 64 + lane];
    float atl[4];  // W_last^T: A[i = hidden 16w+i][k = channel 4s+q]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int ch = 4 * s + q;
        atl[s] = ch < C ? A.params[p.offW[NL] + (int64_t)ch * TBC + 16 * w + i] : 0.0f;
    }
    float ath[NL > 1 ? NL - 1 : 1][16];  // W_l^T: A[i = in 16w+i][k = out 4s+q]
#pragma unroll
    for (int l = 1; l < NL; ++l)
#pragma unroll
        for (int s = 0; s < 16; ++s) ath[l - 1][s] = A.params[p.offW[l] + (int64_t)(4 * s + q) * TBC + 16 * w + i];
    f32x4 bias[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        float4 b4 = *reinterpret_cast<const float4*>(A.params + p.offB[l] + 16 * w + 4 * q);
        bias[l][0] = b4.x; bias[l][1] = b4.y; bias[l][2] = b4.z; bias[l][3] = b4.w;
    }
    f32x4 bias_last;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_last[r] = (4 * q + r) < C ? A.params[p.offB[NL] + 4 * q + r] : 0.0f;

    // ---- phase 0: which rows
    if (tid < TB) {
        int64_t pix = mypix < 0 ? 0 : (mypix >= A.npix ? A.npix - 1 : mypix);
        pixs[tid] = (int)pix;
    }
    __syncthreads();
    STAMP(1);
    // ---- phase 1: copy the 32 rows (features | labels) into LDS: 16 threads per row, 16 B per load,
    //      all loads of a batch issued before the first store (a4: ref LBDRNdataset.py:151-155)
    {
        const int s = tid >> 4, sub = tid & 15;
        const float* src = A.rows + (size_t)pixs[s] * p.RP;
        float* xr = Xs + s * FP;
        const int rp4 = p.RP >> 2;
        for (int c0 = 0; c0 < rp4; c0 += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int c4 = min(c0 + sub + 16 * u, rp4 - 1);  // clamped: never a load behind a branch
                v[u] = *reinterpret_cast<const float4*>(src + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int c4 = c0 + sub + 16 * u;
                if (c4 < rp4) {
                    *reinterpret_cast<float2*>(xr + 4 * c4) = make_float2(v[u].x, v[u].y);
                    *reinterpret_cast<float2*>(xr + 4 * c4 + 2) = make_float2(v[u].z, v[u].w);
                }
            }
        }
        for (int f = p.RP + sub; f < FP; f += 16) xr[f] = 0.0f;  // pad columns read by MFMA steps
    }
    __syncthreads();
    STAMP(2);

    // ---- phase 2: layer 0, z^T[16w..][16 samples of tile st] = b + W0 X^T
    f32x4 acc = bias[0];
    const float* xb = Xs + srow * FP + q;  // B operand: X[sample][4s + q]
    const int nch = p.S0 / CH;
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) {
#pragma unroll
            for (int u = 0; u < CH; ++u) a0[1][u] = wf0[((c + 1) * CH + u) * 64];
        }
        {
            float bx[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) bx[u] = xb[4 * (c * CH + u)];
#pragma unroll
            for (int u = 0; u < CH; ++u) acc = MFMA16(a0[0][u], bx[u], acc);
        }
        if (c + 1 < nch) {
            if (c + 2 < nch) {
#pragma unroll
                for (int u = 0; u < CH; ++u) a0[0][u] = wf0[((c + 2) * CH + u) * 64];
            }
            float bx[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) bx[u] = xb[4 * ((c + 1) * CH + u)];
#pragma unroll
            for (int u = 0; u < CH; ++u) acc = MFMA16(a0[1][u], bx[u], acc);
        }
    }
    STAMP(3);
    // activation; keep cos(30 z) in registers for the backward pass
    f32x4 cs[NL];
    auto activate_store = [&](int l) {
        f32x4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sn, co;
            canon_sincos(30.0f * acc[r], sn, co);
            hv[r] = sn;
            cs[l][r] = co;
        }
        float* dst = Hs + (size_t)l * TB * HPITCH + srow * HPITCH + 16 * w + 4 * q;
        *reinterpret_cast<float2*>(dst) = make_float2(hv[0], hv[1]);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(hv[2], hv[3]);
    };
    activate_store(0);
    __syncthreads();
    STAMP(4);

    // ---- phase 3: hidden layers 1..NL-1
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        acc = bias[l];
        const float* hb = Hs + (size_t)(l - 1) * TB * HPITCH + srow * HPITCH + q;
        float bh[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) bh[s] = hb[4 * s];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = MFMA16(ah[l - 1][s], bh[s], acc);
        activate_store(l);
        __syncthreads();
    }
    STAMP(5);

    // ---- phase 4: output layer + loss + d(loss)/dz_out on the w == 0 wave of each sample tile
    double lsum = 0.0;
    if (w == 0) {
        f32x4 oe = bias_last, oo = {0.f, 0.f, 0.f, 0.f};
        const float* hb = Hs + (size_t)(NL - 1) * TB * HPITCH + srow * HPITCH + q;
        float bo[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) bo[s] = hb[4 * s];
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            oe = MFMA16(al[s], bo[s], oe);
            oo = MFMA16(al[s + 1], bo[s + 1], oo);
        }
        const bool live = srow < nvalid;
        const float* labrow = Xs + srow * FP + F + 4 * q;  // labels ride behind the features
        float dzo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = live && (4 * q + r) < C;
            float y = canon_sigmoid(oe[r] + oo[r]);
            float d = y - (ok ? labrow[r] : 0.0f);
            lsum += ok ? (double)(d * d) : 0.0;                           // ref LBDRNloss.py:9
            dzo[r] = ok ? ((2.0f * d) * A.inv) * (y * (1.0f - y)) : 0.0f;  // mse + sigmoid backward
        }
        float* dst = Zo + srow * OPITCH + 4 * q;
        *reinterpret_cast<float2*>(dst) = make_float2(dzo[0], dzo[1]);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(dzo[2], dzo[3]);
        for (int o = 32; o > 0; o >>= 1) lsum += __shfl_down(lsum, o);
        if (lane == 0) red[st] = lsum;
    }
    __syncthreads();
    if (tid == 0) A.loss_part[wg] = red[0] + red[1];
    STAMP(6);

    auto backprop_store = [&](int l) {  // acc = dL/dh_l  ->  dz_l = (dh * cos(30 z)) * 30
        f32x4 dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = (acc[r] * cs[l][r]) * 30.0f;
        float* dst = Zs + (size_t)l * TB * HPITCH + srow * HPITCH + 16 * w + 4 * q;
        *reinterpret_cast<float2*>(dst) = make_float2(dz[0], dz[1]);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(dz[2], dz[3]);
    };

    // ---- phase 5: dh_{NL-1} = W_last^T dz_out  (K = 16 channel slots)
    {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        acc = zero;
        const float* zb = Zo + srow * OPITCH + q;
        float bz[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) bz[s] = zb[4 * s];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = MFMA16(atl[s], bz[s], acc);
        backprop_store(NL - 1);
    }
    __syncthreads();
    // ---- phase 6: dh_{l-1} = W_l^T dz_l
#pragma unroll
    for (int l = NL - 1; l >= 1; --l) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        acc = zero;
        const float* zb = Zs + (size_t)l * TB * HPITCH + srow * HPITCH + q;
        float bz[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) bz[s] = zb[4 * s];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = MFMA16(ath[l - 1][s], bz[s], acc);
        backprop_store(l - 1);
        __syncthreads();
    }
    STAMP(7);

    // ---- phase 7: weight gradients, K = 32 samples (8 MFMA steps), assembled in LDS (Gs) in
    //      parameter order.  dW_l[row0 + row][col] = sum_s dz_l[s][row0 + row] * in_l[s][col];
    //      this wave takes the 16-column tiles nt = tile0, tile0 + tstep, ...
    auto grad_tiles = [&](const float* zsrc, int zpitch, const float* bsrc, int bpitch, int ntiles,
                          int tile0, int tstep, int ncols, int64_t off, int ld, int row0) {
        float az[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) az[s] = zsrc[(4 * s + q) * zpitch + row0 + i];
        // two tiles per trip (two independent accumulator chains); tiles past the end recompute the
        // last one and are not stored
        for (int nt = tile0; nt < ntiles; nt += 2 * tstep) {
            const int nt1 = nt + tstep;
            const bool two = nt1 < ntiles;
            const float* b0 = bsrc + q * bpitch + 16 * nt + i;
            const float* b1 = bsrc + q * bpitch + 16 * (two ? nt1 : nt) + i;
            float bv[16];
#pragma unroll
            for (int s = 0; s < 8; ++s) { bv[s] = b0[4 * s * bpitch]; bv[8 + s] = b1[4 * s * bpitch]; }
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 g0 = zero, g1 = zero;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                g0 = MFMA16(az[s], bv[s], g0);
                g1 = MFMA16(az[s], bv[8 + s], g1);
            }
            float* dst0 = Gs + off + (int64_t)(row0 + 4 * q) * ld + 16 * nt + i;
            float* dst1 = Gs + off + (int64_t)(row0 + 4 * q) * ld + 16 * nt1 + i;
            const bool ok0 = 16 * nt + i < ncols, ok1 = two && 16 * nt1 + i < ncols;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (ok0) dst0[r * ld] = g0[r];
                if (ok1) dst1[r * ld] = g1[r];
            }
        }
    };
    grad_tiles(Zs, HPITCH, Xs, FP, p.NT0, st, 2, F, p.offW[0], F, 16 * w);
    STAMP(10);
#pragma unroll
    for (int l = 1; l < NL; ++l)
        grad_tiles(Zs + (size_t)l * TB * HPITCH, HPITCH, Hs + (size_t)(l - 1) * TB * HPITCH, HPITCH, 4, st, 2,
                   TBC, p.offW[l], TBC, 16 * w);
    if (st == 0) {  // output layer: rows = channels, wave w takes hidden columns 16w..16w+15
        float az[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) az[s] = Zo[(4 * s + q) * OPITCH + i];
        const float* b0 = Hs + (size_t)(NL - 1) * TB * HPITCH + q * HPITCH + 16 * w + i;
        float bv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) bv[s] = b0[4 * s * HPITCH];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 g0 = zero, g1 = zero;
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            g0 = MFMA16(az[s], bv[s], g0);
            g1 = MFMA16(az[s + 1], bv[s + 1], g1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = 4 * q + r;
            if (ch < C) Gs[p.offW[NL] + (int64_t)ch * TBC + 16 * w + i] = g0[r] + g1[r];
        }
    }
    STAMP(11);
    // bias gradients: column sums over the 32 samples, one thread per unit, samples in order
    for (int u = tid; u < NL * TBC + C; u += TRAIN_THREADS) {
        float v = 0.0f;
        if (u < NL * TBC) {
            const int l = u >> 6, n = u & 63;
            const float* z = Zs + (size_t)l * TB * HPITCH + n;
            float zz[TB];
#pragma unroll
            for (int s = 0; s < TB; ++s) zz[s] = z[s * HPITCH];
#pragma unroll
            for (int s = 0; s < TB; ++s) v += zz[s];
            Gs[p.offB[l] + n] = v;
        } else {
            const int ch = u - NL * TBC;
            for (int s = 0; s < TB; ++s) v += Zo[s * OPITCH + ch];
            Gs[p.offB[NL] + ch] = v;
        }
    }
    for (int u = (int)p.NP + tid; u < p.NPP; u += TRAIN_THREADS) Gs[u] = 0.0f;
    __syncthreads();
    STAMP(8);
    // ---- phase 8: the slab leaves as full-width stores (16 B per lane, 1 KB per wave instruction)
    {
        float4* dst = reinterpret_cast<float4*>(A.slabs + (size_t)wg * p.NPP);
        const float4* srcg = reinterpret_cast<const float4*>(Gs);
        for (int u = tid; u < (p.NPP >> 2); u += TRAIN_THREADS) dst[u] = srcg[u];
    }
#ifdef LBDRN_TRAIN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(9);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[13])::"memory");
    if (tid == 0 && A.stamps)
        for (int k = 0; k < 16; ++k) A.stamps[(size_t)wg * 16 + k] = stamp[k];
#endif
}

// ------------------------------------------------------------------ host driver

template <int CH, int NL>
static int launch_train(const TrainArgs& A, int nwg, hipStream_t s)
{
    auto kern = k_train_mfma<CH, NL>;
    const size_t lds_bytes = (size_t)A.p.lds_floats * 4;
    static thread_local size_t configured = 0;
    if (configured < lds_bytes) {
        LBDRN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        configured = lds_bytes;
    }
    kern<<<nwg, TRAIN_THREADS, lds_bytes, s>>>(A);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

static int dispatch_train(const TrainArgs& A, int nwg, hipStream_t s)
{
    const int nl = A.net.nl, ch = A.p.CH;
    if (ch == 10) {
        if (nl == 1) return launch_train<10, 1>(A, nwg, s);
        if (nl == 2) return launch_train<10, 2>(A, nwg, s);
        return launch_train<10, 3>(A, nwg, s);
    }
    if (nl == 1) return launch_train<8, 1>(A, nwg, s);
    if (nl == 2) return launch_train<8, 2>(A, nwg, s);
    return launch_train<8, 3>(A, nwg, s);
}

// Build the per-image state of the fused training path in the caller's workspace: the
// [N][F+C] row matrix.  Must run once per image before mfma_train_epoch.
int mfma_train_prepare(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                       const uint16_t* msb, int bs, void* ws, size_t ws_bytes, hipStream_t s)
{
    TrainPlan p;
    if (!make_train_plan(g, net, &p)) {
        set_error("shape not supported by the MFMA train kernel");
        return LBDRN_E_UNSUPPORTED;
    }
    const TrainWsLayout L = train_ws_layout(g, net, p, bs);
    if (!ws || ws_bytes < L.total) {
        set_error("train workspace too small: %zu < %zu", ws_bytes, L.total);
        return LBDRN_E_WORKSPACE;
    }
    float* rows = (float*)((char*)ws + L.off_rows);
    const int64_t total = (int64_t)g.H * g.W * p.RP;
    LBDRN_REQUIRE((total + 255) / 256 < ((int64_t)1 << 31), "image too large for one launch");
    k_build_rows<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(g, net.F, p.RP, msb, img, rows);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

int mfma_train_epoch(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                     const uint16_t* msb, const int64_t* perm, int64_t n, int bs, float* params,
                     float* m, float* v, int64_t step0, double lr, float* losses, void* ws,
                     size_t ws_bytes, hipStream_t s)
{
    TrainArgs A;
    if (!make_train_plan(g, net, &A.p)) {
        set_error("shape not supported by the MFMA train kernel");
        return LBDRN_E_UNSUPPORTED;
    }
    const TrainWsLayout L = train_ws_layout(g, net, A.p, bs);
    if (!ws || ws_bytes < L.total) {
        set_error("train workspace too small: %zu < %zu", ws_bytes, L.total);
        return LBDRN_E_WORKSPACE;
    }
    float* rows = (float*)((char*)ws + L.off_rows);
    float* packed = (float*)((char*)ws + L.off_pack);
    float* slabs = (float*)((char*)ws + L.off_slab);
    double* loss_part = (double*)((char*)ws + L.off_loss);
    LBDRN_HIP_TRY(hipMemsetAsync(packed, 0, (size_t)A.p.pack_floats * sizeof(float), s));
    k_pack_train<<<(unsigned)((A.p.NP + 255) / 256), 256, 0, s>>>(params, A.p, net.F, net.nl, net.C, packed);
    LBDRN_LAUNCH_CHECK();
    A.net = net; A.rows = rows; A.npix = (int64_t)g.H * g.W; A.params = params; A.packed = packed;
    A.slabs = slabs; A.loss_part = loss_part;
    A.stamps = nullptr;
#ifdef LBDRN_TRAIN_STAMPS
    const int max_wg = (bs + TB - 1) / TB;
    LBDRN_HIP_TRY(hipMalloc(&A.stamps, (size_t)max_wg * 16 * sizeof(unsigned long long)));
#endif
    int64_t step = step0;
    int si = 0;
    for (int64_t first = 0; first < n; first += bs, ++si) {
        const int B = (int)std::min<int64_t>(bs, n - first);
        const int nwg = (B + TB - 1) / TB;
        A.perm = perm + first;
        A.batch_n = B;
        A.inv = 1.0f / ((float)B * (float)net.C);
        if (int rc = dispatch_train(A, nwg, s)) return rc;
        ++step;
        const double bc1 = 1.0 - std::pow(0.9, (double)step), bc2 = 1.0 - std::pow(0.999, (double)step);
        k_reduce_adam<<<(unsigned)(A.p.NPP / 64), 256, 0, s>>>(
            slabs, nwg, A.p, net.F, net.nl, net.C, params, m, v, packed, (float)(lr / bc1),
            (float)std::sqrt(bc2), loss_part, (double)B * net.C, losses ? losses + si : nullptr);
        LBDRN_LAUNCH_CHECK();
    }
#ifdef LBDRN_TRAIN_STAMPS
    {   // diagnostic: mean cycles per phase over the workgroups of the last step
        LBDRN_HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)max_wg * 16);
        LBDRN_HIP_TRY(hipMemcpy(h.data(), A.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(A.stamps);
        double sum[12] = {};
        for (int wq = 0; wq < max_wg; ++wq)
            for (int k = 1; k < 10; ++k) sum[k] += (double)(h[wq * 16 + k] - h[wq * 16 + k - 1]);
        double d10 = 0, d11 = 0, d8 = 0, clk = 0;
        for (int wq = 0; wq < max_wg; ++wq) {
            d10 += (double)(h[wq * 16 + 10] - h[wq * 16 + 7]);
            d11 += (double)(h[wq * 16 + 11] - h[wq * 16 + 10]);
            d8 += (double)(h[wq * 16 + 8] - h[wq * 16 + 11]);
            clk += (double)(h[wq * 16 + 9] - h[wq * 16 + 0]) / (double)(h[wq * 16 + 13] - h[wq * 16 + 12]) * 100.0;
        }
        fprintf(stderr, "[lbdrn stamps] mean in-kernel clock %.0f MHz\n", clk / max_wg);
        fprintf(stderr, "[lbdrn stamps] dW0 %.0f dWhid+out %.0f bias+pad+barrier %.0f\n", d10 / max_wg, d11 / max_wg, d8 / max_wg);
        fprintf(stderr, "[lbdrn stamps] mean cycles/phase: perm %.0f rows %.0f L0mfma %.0f act0 %.0f hidden %.0f "
                        "out+loss %.0f backprop %.0f dW->LDS %.0f slab-out %.0f\n",
                sum[1] / max_wg, sum[2] / max_wg, sum[3] / max_wg, sum[4] / max_wg, sum[5] / max_wg,
                sum[6] / max_wg, sum[7] / max_wg, sum[8] / max_wg, sum[9] / max_wg);
    }
#endif
    return 0;
}

}  // namespace lbdrn
