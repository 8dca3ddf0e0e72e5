// Fused training step for gfx950: one launch per minibatch does gather -> forward -> loss ->
// backward -> per-workgroup gradient slab, a second launch reduces the slabs in a fixed order,
// applies torch's Adam update and refreshes the MFMA-ordered copy of the weights
// (ref modified_ignite_engine.py:18-27, encode.py:84; LBDRNdataset.py:136-155 for the gather).
//
// Why it looks like this on MI355X
//  * a minibatch is 8192 rows and 646 MFLOP: 4.1 us of the whole chip at the f32 MFMA peak.  The
//    step is a serial chain (the next step needs the updated weights), so it is latency-bound: the
//    batch is cut into 256 workgroups of 32 samples -- one per CU -- and every layer of those 32
//    samples is split over 8 waves: wave (w, st) owns hidden neurons 16w..16w+15 of the 16-sample
//    column tile st (v_mfma_f32_16x16x4_f32).  Two waves share each SIMD so that one wave's
//    MFMAs cover the other's operand reads and bookkeeping (a wave issues in order).
//  * "quarter-K" operand order: in one 16x16x4 MFMA the four k slots belong to the four 16-lane
//    quarters of the wave; any assignment of k indices to slots is legal as long as A and B
//    agree.  Quarter q takes the contiguous range k = q*L .. q*L+L-1 (L = K/4), so a lane's B
//    operands for ALL steps of a layer are L contiguous floats of one LDS row: a handful of
//    ds_read_b128 instead of one ds_read_b32 per MFMA (measured: with b32 operand reads the step
//    spent 55-86 cycles per 32-cycle MFMA).  The weights are packed to match.
//  * activations and their gradients sit in LDS twice, [sample][unit] for the forward / backward
//    products and [unit][sample] for the weight-gradient products, all rows 16-byte aligned with an
//    odd number of 16-byte chunks per row.  Weights never touch LDS: every A operand is prefetched
//    L2 -> VGPR at kernel start, in fragment order (one 256-byte line per wave load).
//  * cos(30 z) needed by the backward pass is kept in registers: the wave that produced a tile of
//    z is the wave that later receives the matching tile of dL/dh, in the same lane/register slots.
//  * weight gradients are sums over the minibatch: each workgroup writes its partial in TILE order
//    (an accumulator tile is 64 lanes x 16 contiguous bytes = one 1 KB wave store, no LDS staging);
//    k_reduce_adam adds the 256 slabs in index order (bitwise reproducible, no atomics) and maps
//    tile order back to state_dict order for the Adam update.
//  * the minibatch gather reads whole rows of a [N][F+C] float32 matrix (features | labels) that
//    lbdrn_train_prepare materialises once per image: 3.5 GB for a 2048^2x8 tile, nothing on a
//    288 GB part, and a random 832-byte row is 7 full cache lines, where gathering the 5x5xC window
//    from a pixel plane moves 2-4x the bytes in partial lines (measured: 5.9 us of a 22 us step).
//    The evaluation / decode passes never use this matrix (they stage windows through LDS).
//
// Training parity is a tolerance contract (1e-5 relative on the loss), not a bit
// pattern: the summation order differs from the generic kernels and from torch.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <utility>
#include <chrono>
#include <vector>

#include "common.hpp"
#include "lbdrn_math.hpp"

namespace lbdrn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_GROUP = 4;  // independent fits of one shape that may step side by side in one launch (blockIdx.y = fit)
constexpr int TB = 32;       // samples per workgroup
constexpr int TBC = 64;      // hidden width this kernel is built for
constexpr int HP = 68;       // [sample][64] rows: 17 chunks of 16 B
constexpr int TP = 36;       // [unit][32 samples] rows: 9 chunks of 16 B
constexpr int OP = 20;       // [sample][16 channel slots] rows: 5 chunks
#ifndef LBDRN_RED_SLICES
#define LBDRN_RED_SLICES 8
#endif
constexpr int RED_SLICES = LBDRN_RED_SLICES;   // workgroup slices per reduce block: 8 x 32 float4 lanes = 512 B per slab and block.
                                // A/B with the round-2 kernels (128 slabs; scripts/ab_lib.sh, ms per tile for one fit alone /
                                // four in flight): 8 slices 136 / 76-80, 4: 144 / 76, 16: 144 / 77-78, 32: 140 / 77
#ifndef LBDRN_RED_LANES
#define LBDRN_RED_LANES 32
#endif
constexpr int RED_LANES = LBDRN_RED_LANES;   // float4 lanes per reduce block (x RED_SLICES threads)
#ifndef LBDRN_SLAB_ROT
#define LBDRN_SLAB_ROT 4
#endif
// Slab buffers a fit's steps rotate through.  The XCDs' L2s are kept coherent by hardware (a snoop filter at the Infinity
// Cache): a store to a line that another XCD's L2 holds -- or held: the filter keeps the entry -- waits for a probe.  The
// reduce / Adam launch reads every slab line into some L2, and with ONE buffer the next training launch's write-through
// stores then paid for it: 5.7 us of a lone k_train_split step (profiles/r05_slab_store_ab.txt: 22.4 us a step; 15.2 with
// the reduce launch reading nothing, 16.6 with the training launch storing nothing).  With four buffers a line is
// rewritten three steps after it was read and the stores run free: 22.3 -> 17.1 us a step (two buffers: 17.7).  Reading
// the slabs with non-temporal loads instead (LBDRN_RED_AUX=2) also frees the stores, but the reads themselves take 5.0
// instead of 3.7 us.
constexpr int SLAB_ROT = LBDRN_SLAB_ROT;           // ... of the launch over k_train_split's slab pairs: 64 threads, 584 blocks at the headline shape
constexpr int TRAIN_THREADS = 512;  // 8 waves: (neuron tile w = 0..3) x (sample tile st = 0..1)

// ---- the wave-local steps (k_train_stream, k_train_wide): 64 samples per workgroup, 16 samples per compute wave
constexpr int WB = 64;            // samples per workgroup
constexpr int WAVE_THREADS = 256;
constexpr int WOP = 16;           // k_train_wide: [sample][16 channel slots] pitch of dz_out
__host__ __device__ constexpr int wave_xp(int LQ) { return LQ == 16 ? 80 : LQ == 32 ? 144 : LQ == 48 || LQ == 52 ? 208 : 272; }   // k_train_wide: row pitch, matrix = LDS
constexpr int TBC_W = 64;
constexpr int WPT = 68;   // [unit][64 samples] pitch of the transposed dz copies (16-byte rows)

// The features the streamed step multiplies.  With RELATIVE and D > 0 the window's centre minus itself is an exact
// 0.0f for every band (LBDRNdataset.py:126-128: feature 2P + c (2D+1)^2 + D (2D+1) + D).  Its products add nothing to a
// pre-activation, the gradient of its weights is a sum of exact zeros, and torch's Adam leaves a parameter whose
// gradient has always been 0 where it is (exp_avg = exp_avg_sq = 0: the step is 0 / (0 + eps); encode.py:84 sets no
// weight decay).  So those C columns of W_0 take no part in a fit: the step runs over the Fe = F - C features that
// can differ from zero (slot j of its order = feature feat_of_slot(j)); F = 200 -> 192 = twelve full MFMA groups instead
// of thirteen, and the columns keep their initial values in the parameter vector, as they do in the reference.
struct FeatMap {
    int Fe;        // features the step multiplies
    int zP, zs2, zc;   // zs2 > 0: features zP + c zs2 + zc, c = 0, 1, .. are skipped
};
__host__ __device__ __forceinline__ int feat_of_slot(int j, const FeatMap& m)
{
    if (m.zs2 == 0 || j < m.zP) return j;
    const int cj = j - m.zP, c = cj / (m.zs2 - 1), r = cj - c * (m.zs2 - 1);
    return m.zP + c * m.zs2 + r + (r >= m.zc ? 1 : 0);
}
__host__ __device__ __forceinline__ int slot_of_feat(int k, const FeatMap& m)   // -1: a skipped feature
{
    if (m.zs2 == 0 || k < m.zP) return k;
    const int ck = k - m.zP, c = ck / m.zs2, r = ck - c * m.zs2;
    if (r == m.zc) return -1;
    return m.zP + c * (m.zs2 - 1) + r - (r > m.zc ? 1 : 0);
}

// the map of a shape whose fused step skips the window centres (identity otherwise)
static FeatMap centre_skipping_map(const lbdrn_geom& g, const lbdrn_net& net)
{
#ifdef LBDRN_EXP_KEEP_CENTRE   // (A/B build: the step multiplies the zeros as well)
    constexpr bool keep_zero = true;
#else
    constexpr bool keep_zero = false;
#endif
    const int side = 2 * g.D + 1;
    if (!keep_zero && g.use_colors && g.relative && g.D > 0 && net.F == 2 * g.P + g.C * side * side)
        return FeatMap{net.F - g.C, 2 * g.P, side * side, g.D * side + g.D};
    return FeatMap{net.F, 0, 0, 0};
}

struct TrainPlan {
    FeatMap fm;
    int LQ;                    // layer-0 quarter length = MFMA steps of layer 0 (fm.Fe <= 4*LQ)
    int XP;                    // X row pitch = 4*LQ + 4 floats
    int NT0;                   // 16-wide feature tiles of dW0 = ceil(fm.Fe/16)
    int RP;                    // row pitch of the materialised [N][RP] matrix: F features, C labels, pad to x4
    int64_t NP;
    int64_t offW[5], offB[5];  // canonical parameter offsets per layer (index nl = last layer)
    int pk_w0, pk_wh, pk_wl, pack_floats;  // fragment-order buffer (floats)
    int wave;                  // 2: k_train_stream, 0: the tile kernel (fragment orders differ, see frag_pos)
    int pk_wht, pk_wlt;        // wave-local kernel: W_l^T and W_last^T fragments for the backward products
    int w_dp, w_df;            // 64 / (RP/4) and 64 % (RP/4): chunk walk of the row copy
    int wave_lds_floats;
    int sl_hid, sl_out, sl_bias, slab_floats;  // slab (tile order) section starts, in floats
    int lds_x, lds_xt, lds_h, lds_ht, lds_z, lds_zt, lds_zo, lds_zot, lds_pix, lds_red, lds_floats;
};

// which fused step runs a shape this file supports: 2 = k_train_stream, 0 = the 8-wave tile kernel k_train_mfma (nl = 3).
// A function of the shape only (the choice fixes the row-matrix layout that lbdrn_train_prepare builds and
// lbdrn_train_epoch reads).  -DLBDRN_EXP_TILE_KERNEL: A/B build that runs every shape on the tile kernel.
static int train_kernel_choice()
{
#ifdef LBDRN_EXP_TILE_KERNEL
    return 0;
#else
    return 2;
#endif
}

__host__ __device__ constexpr int stream_rp(int LQ);
struct StreamLds;
static int stream_lds_total(int LQ, int NL);

static bool make_train_plan(const lbdrn_geom& g, const lbdrn_net& net, TrainPlan* out)
{
    if ((net.act != LBDRN_ACT_SINE && net.act != LBDRN_ACT_RELU) || net.bc != TBC || net.nl < 1 || net.nl > 3 || net.C > 16 || net.F < 1) return false;
    if (net.act == LBDRN_ACT_RELU && (net.nl > 2 || train_kernel_choice() != 2)) return false;   // (ReLU: the streamed step and k_train_split; the nl = 3 tile kernel is the Sine network's)
    TrainPlan p;
    p.RP = (net.F + net.C + 3) / 4 * 4;
    p.LQ = 0;
    p.fm = FeatMap{net.F, 0, 0, 0};
    int kind = net.nl <= 2 ? train_kernel_choice() : 0;
    if (kind == 2) {   // the streamed step: features in 4 LQ slots, labels in a group of their own
        p.fm = centre_skipping_map(g, net);
        for (int lq : {16, 24, 32, 48, 52, 64})   // (24: the reference's 4-band shape -- F = 100, 96 features that can differ from zero)
            if (p.fm.Fe <= 4 * lq) { p.LQ = lq; break; }
        if (!p.LQ || (size_t)stream_lds_total(p.LQ, net.nl) * 4 > 160 * 1024) { kind = 0; p.LQ = 0; p.fm = FeatMap{net.F, 0, 0, 0}; }
    }
    if (kind != 2 && net.act != LBDRN_ACT_SINE) return false;
    if (kind != 2)
        for (int lq : {16, 32, 52, 64})
            if (net.F <= 4 * lq && p.RP <= 4 * lq + 4) { p.LQ = lq; break; }
    if (!p.LQ) return false;
    p.XP = 4 * p.LQ + 4;
    p.NT0 = (p.fm.Fe + 15) / 16;
    if (kind != 2 && 16 * p.NT0 > p.XP) return false;
    p.NP = param_count(net);
    int64_t o = 0;
    for (int l = 0; l < net.nl; ++l) {
        int nin = l ? TBC : net.F;
        p.offW[l] = o; o += (int64_t)TBC * nin;
        p.offB[l] = o; o += TBC;
    }
    p.offW[net.nl] = o; o += (int64_t)net.C * TBC;
    p.offB[net.nl] = o;
    int k = 0;
    p.pk_w0 = k; k += 4 * p.LQ * 64;
    p.pk_wh = k; k += (net.nl - 1) * 4 * 16 * 64;
    p.pk_wl = k; k += 16 * 64;
    p.pk_wht = k; k += (net.nl - 1) * 4 * 16 * 64;
    p.pk_wlt = k; k += 16 * 64;
    p.pack_floats = k;
    p.w_dp = p.w_df = 0;
    p.wave_lds_floats = 0;
    if (kind == 2) {
        p.RP = stream_rp(p.LQ);   // the matrix in the order layer 0 eats it (train_stream.inc)
        p.wave_lds_floats = stream_lds_total(p.LQ, net.nl);
    }
    p.wave = kind;
    int s = 4 * p.NT0 * 256;
    p.sl_hid = s; s += (net.nl - 1) * 16 * 256;
    p.sl_out = s; s += 4 * 256;
    p.sl_bias = s; s += net.nl * TBC + 16;
    p.slab_floats = (s + 127) / 128 * 128;  // multiple of 4*RED_LANES
    int f = 0;
    p.lds_x = f; f += TB * p.XP;
    p.lds_xt = f; f += 16 * p.NT0 * TP;
    p.lds_h = f; f += net.nl * TB * HP;
    p.lds_ht = f; f += net.nl * TBC * TP;
    p.lds_z = f; f += net.nl * TB * HP;
    p.lds_zt = f; f += net.nl * TBC * TP;
    p.lds_zo = f; f += TB * OP;
    p.lds_zot = f; f += 16 * TP;
    p.lds_pix = f; f += TB;
    p.lds_red = f; f += 16;
    p.lds_floats = f;
    if (kind == 0 && (size_t)f * 4 > 160 * 1024) return false;
    *out = p;
    return true;
}

struct WidePlan;
static bool make_wide_plan(const lbdrn_geom& g, const lbdrn_net& net, WidePlan* out);
static bool wide_supported(const lbdrn_geom& g, const lbdrn_net& net);
static size_t wide_workspace(const lbdrn_geom& g, const lbdrn_net& net, int bs);

static bool slabs_addressable(const TrainPlan& p, int bs);

// bs > 0: ... at that minibatch size (k_reduce_adam addresses one step's slabs through a buffer resource: 32-bit offsets)
bool mfma_train_supported(const lbdrn_geom& g, const lbdrn_net& net, int bs)
{
    TrainPlan p;
    if (make_train_plan(g, net, &p)) return bs <= 0 || slabs_addressable(p, bs);
    return wide_supported(g, net);
}

bool mfma_train_takes_groups(const lbdrn_geom& g, const lbdrn_net& net)
{
    TrainPlan p;
    return make_train_plan(g, net, &p) && p.wave == 2;
}


struct TrainWsLayout {
    size_t off_rows, off_pack, off_slab, slab_bytes, off_loss, off_stage, stage_bytes, off_map, total;
};

static TrainWsLayout train_ws_layout(const lbdrn_geom& g, const lbdrn_net& net, const TrainPlan& p, int bs)
{
    TrainWsLayout L;
    size_t o = 0;
    L.off_rows = o; o += align_up((size_t)g.H * g.W * p.RP * sizeof(float), 256);
    L.off_pack = o; o += align_up((size_t)p.pack_floats * sizeof(float), 256);
    // slabs / loss partials: one per workgroup of whichever kernel steps -- 32-row workgroups (the tile kernel), 64-row ones
    // (k_train_stream), or two per 64-row group (k_train_split)
    const size_t nwg = std::max((size_t)(bs + TB - 1) / TB, 2 * ((size_t)(bs + WB - 1) / WB));
    L.slab_bytes = align_up(nwg * (size_t)p.slab_floats * sizeof(float), 256);
    L.off_slab = o; o += SLAB_ROT * L.slab_bytes;
    L.off_loss = o; o += align_up(nwg * sizeof(double), 256);
    // one minibatch of rows, contiguous, rounded up to whole workgroups of either kernel (32 / 64 rows)
    L.stage_bytes = align_up(align_up((size_t)bs, WB) * p.RP * sizeof(float), 256);
    L.off_stage = o; o += 2 * L.stage_bytes;
    L.off_map = o; o += align_up((size_t)p.slab_floats * 4 * sizeof(int), 256);  // slab element -> (param, fragment slots)
    L.total = o;
    return L;
}

// One step's set of slabs is read through ONE buffer resource with 32-bit byte offsets (k_reduce_adam): it must stay below
// 2 GiB -- ~0.92 M rows per minibatch at the headline shape (a slab is 73 KB per 32 rows).  Larger minibatches take the
// generic step (LBDRN_PATH_AUTO) or are refused (LBDRN_PATH_MFMA): ADVICE round 5.
static bool slabs_addressable(const TrainPlan& p, int bs)
{
    const size_t nwg = std::max((size_t)(bs + TB - 1) / TB, 2 * ((size_t)(bs + WB - 1) / WB));
    return nwg * (size_t)p.slab_floats * sizeof(float) < ((size_t)1 << 31);
}

size_t mfma_train_workspace(const lbdrn_geom& g, const lbdrn_net& net, int bs)
{
    TrainPlan p;
    if (!make_train_plan(g, net, &p)) return wide_workspace(g, net, bs);
    return train_ws_layout(g, net, p, bs).total;
}

// ------------------------------------------------------------------ helper kernels

// what sits at position `pos` of a row of the streamed step's matrix (LQs = its quarter length; 0: rows are
// the step's features (FeatMap) in slot order | labels): a row is LQs/4 groups of 16 floats, group g = [quarter kq][e] = feature
// kq LQs + 4 g + e -- what MFMA group g of layer 0 multiplies -- then one group of 16 label slots.
// (slot kq LQs + 4 g + e of the step's feature order, FeatMap.)  Returns the feature index, F + channel for a label, or
// -1 (a zero).
__device__ __forceinline__ int row_source(int pos, int LQs, int F, int C, const FeatMap& fm)
{
    if (LQs == 0) return pos < fm.Fe ? feat_of_slot(pos, fm) : pos < fm.Fe + C ? F + (pos - fm.Fe) : -1;
    const int g = pos >> 4;
    if (g < (LQs >> 2)) {
        const int j = ((pos >> 2) & 3) * LQs + 4 * g + (pos & 3);
        return j < fm.Fe ? feat_of_slot(j, fm) : -1;
    }
    const int ch = pos - 4 * LQs;
    return ch < C ? F + ch : -1;
}

// rows[n][0..F) = features of pixel n, rows[n][F..F+C) = labels, rest 0 (LQs > 0: the same values in the streamed
// step's order, row_source)
// (ref LBDRNdataset.py:95-97, 104-131 -- the reference's own [N,F] / [N,C] matrices, side by side)
__global__ void __launch_bounds__(256)
    k_build_rows(lbdrn_geom g, int F, int RP, int LQs, FeatMap fm, const uint16_t* __restrict__ msb,
                 const uint16_t* __restrict__ img, float* __restrict__ rows)
{
    const int64_t HW = (int64_t)g.H * g.W;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= HW * RP) return;
    const int64_t pix = e / RP;
    const int f = row_source((int)(e - pix * RP), LQs, F, g.C, fm);
    const int y = (int)(pix / g.W), x = (int)(pix - (int64_t)y * g.W);
    float v = 0.0f;
    if (f < 0) {
    } else if (f < g.P) {
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

// The same matrix, built tile-wise: one block = 32 consecutive pixels of one image row (9.0 KB of LDS at the headline
// shape: the launch fits beside the training workgroups of other fits, which leave 16.6 KB of a CU's LDS free, and
// its 3.5 GB of stores ride under their MFMAs instead of holding the chip).  The normalised window p = msb/max of the
// tile (C x (2D+1) rows x (32+2D) columns, reflect-padded) is staged in LDS once -- one IEEE division per staged value
// instead of two per output --, what sits at each of the row's positions is tabulated once per block, and the outputs
// leave in row-major order as 16-byte stores (the per-element kernel above spends its time in 64-bit index divisions
// and scattered uint16 gathers: 0.7 TB/s; this one: 1.8 ms for the 3.49 GB of a 2048^2 x 8 tile, 2.9 ms before the
// table and the wide stores).
#ifndef LBDRN_BR_TW
#define LBDRN_BR_TW 32
#endif
constexpr int BR_TW = LBDRN_BR_TW;   // pixels per block.  A/B (round 4, alone on the device): 32: 1.80 ms, 64: 1.61, 128: 1.66 -- but at 64 the
                                     // launch needs 16 KB of LDS and no longer fits beside the training workgroups of other fits (13.5 KB free)
__global__ void __launch_bounds__(256)
    k_build_rows_tiled(lbdrn_geom g, int F, int RP, int LQs, FeatMap fm, const uint16_t* __restrict__ msb,
                       const uint16_t* __restrict__ img, float* __restrict__ rows)
{
    extern __shared__ float br_lds[];
    const int side = 2 * g.D + 1, SW = BR_TW + 2 * g.D;
    const int ncolor = F - 2 * g.P;
    float* tile = br_lds;                                   // [C][side][SW]
    int* nbo = reinterpret_cast<int*>(br_lds + g.C * side * SW);   // [ncolor] neighbour offset
    int* cto = nbo + ncolor;                                // [ncolor] centre offset
    const int64_t HW = (int64_t)g.H * g.W;
    const int tiles_x = (g.W + BR_TW - 1) / BR_TW;
    const int y = blockIdx.x / tiles_x, x0 = (blockIdx.x - y * tiles_x) * BR_TW;
    const int tw = min(BR_TW, g.W - x0);
    const int tid = threadIdx.x;
    if (g.use_colors) {
        const float mx = (float)g.msb_max;
        const int total = g.C * side * SW;
        for (int e = tid; e < total; e += 256) {
            const int c = e / (side * SW), r = e - c * side * SW;
            const int sy = r / SW, sx = r - sy * SW;
            const int yy = reflect_fast(y + sy - g.D, g.H), xx = reflect_fast(min(x0 + sx - g.D, g.W - 1 + g.D), g.W);
            tile[e] = (float)msb[(int64_t)c * HW + (int64_t)yy * g.W + xx] / mx;
        }
        for (int k = tid; k < ncolor; k += 256) {
            const int c = k / (side * side), r = k - c * side * side;
            const int dy = r / side, dx = r - dy * side;
            nbo[k] = (c * side + dy) * SW + dx;
            cto[k] = (c * side + g.D) * SW + g.D;
        }
    }
    // what sits at every position of a row, worked out ONCE per block (row_source and the feature map cost ~40
    // instructions per element, and there are 872 M elements in a 2048^2 tile: the launch was instruction-bound at
    // 1.2 TB/s of stores): .x >= 0: colour, tile[.x + pixel] (minus tile[.y + pixel] when .y >= 0); -1: zero; -2: row
    // table entry .y; -3: column table entry .y; -4: label of channel .y
    __syncthreads();   // window and offset tables are in LDS
    const bool rel = g.relative && g.D > 0;
    int2* ptab = reinterpret_cast<int2*>(cto + ncolor);   // [RP]; 8-byte aligned as it stands: C side SW (SW = 32 + 2 D) and 2 ncolor are both even
    // the block's labels, staged like the window (round 4): the store loop below used to fetch each label with a 2-byte
    // global load of its own -- consecutive threads walk the positions of ONE pixel, so eight planes per pixel, no two
    // loads in a line -- in the middle of the stores
    float* labs = reinterpret_cast<float*>(ptab + RP);     // [C][BR_TW]
    {
        const int lmask = (1 << g.K) - 1;
        const float lmaskf = (float)lmask;
        for (int e = tid; e < g.C * BR_TW; e += 256) {
            const int c = e / BR_TW, px = e - c * BR_TW;
            labs[e] = (float)((int)img[(int64_t)c * HW + (int64_t)y * g.W + min(x0 + px, g.W - 1)] & lmask) / lmaskf;
        }
    }
    for (int pos = tid; pos < RP; pos += 256) {
        const int f = row_source(pos, LQs, F, g.C, fm);
        int2 t = make_int2(-1, 0);
        if (f < 0) {
        } else if (f < g.P) t = make_int2(-2, f);
        else if (f < 2 * g.P) t = make_int2(-3, f - g.P);
        else if (f < F) t = make_int2(nbo[f - 2 * g.P], rel ? cto[f - 2 * g.P] : -1);
        else if (f < F + g.C) t = make_int2(-4, f - F);
        ptab[pos] = t;
    }
    __syncthreads();
    float* out = rows + ((int64_t)y * g.W + x0) * RP;
    auto value = [&](int pos, int pix) -> float {
        const int2 t = ptab[pos];
        if (t.x >= 0) {
            const float nb = tile[t.x + pix];
            return t.y >= 0 ? nb - tile[t.y + pix] : nb;
        }
        if (t.x == -1) return 0.0f;
        if (t.x == -2) return g.rowtab[(int64_t)y * g.P + t.y];
        if (t.x == -3) return g.coltab[(int64_t)(x0 + pix) * g.P + t.y];
        return labs[t.y * BR_TW + pix];
    };
    if ((RP & 3) == 0) {
        // four consecutive positions of one pixel per thread and trip: one 16-byte store (rows are 16-byte aligned: RP % 4 == 0)
        const int RQ = RP >> 2;
        int pix = tid / RQ, pq = tid - pix * RQ;
        const int dp = 256 / RQ, dq = 256 - dp * RQ;
        while (pix < tw) {
            const int pos = 4 * pq;
            const float4 v = make_float4(value(pos, pix), value(pos + 1, pix), value(pos + 2, pix), value(pos + 3, pix));
            *reinterpret_cast<float4*>(out + (int64_t)pix * RP + pos) = v;
            pix += dp;
            pq += dq;
            if (pq >= RQ) { pq -= RQ; pix += 1; }
        }
    } else {
        int pix = tid / RP, pos = tid - pix * RP;
        const int dp = 256 / RP, df = 256 - dp * RP;
        for (int e = tid; pix < tw; e += 256) {
            out[e] = value(pos, pix);
            pix += dp;
            pos += df;
            if (pos >= RP) { pos -= RP; pix += 1; }
        }
    }
}

// canonical parameter index -> positions in the fragment-order buffer (or -1: not packed): .x = the copy the forward
// products read, .y = the transposed copy the wave-local kernel's backward products read.
// Quarter-K order: MFMA step s of lane quarter q multiplies k = q*L + s.  Four consecutive steps of a
// lane are adjacent ([step/4][lane][step%4]) so that a fragment fetch is one 16-byte load per four steps.
// Tile kernel (p.wave == 0): hidden and output layers walk k = 16q + s as well.
// Wave-local kernel (p.wave == 1): the B operand of a hidden / output product is the previous layer's accumulator
// as it stands in registers (lane quarter q, tile t, register r holds unit 16t + 4q + r), so step (t, r) multiplies
// k = 16t + 4q + r; the backward products take W^T the same way (k = output unit of the layer).
__device__ __forceinline__ int2 frag_pos(int64_t idx, const TrainPlan& p, int F, int nl, int C)
{
    if (idx < p.offB[0]) {  // W0[n][k]
        int n = (int)(idx / F), k = slot_of_feat((int)(idx - (int64_t)n * F), p.fm);
        if (k < 0) return make_int2(-1, -1);   // a column the step skips (FeatMap)
        int q = k / p.LQ, s = k - q * p.LQ;
        return make_int2(p.pk_w0 + ((((n >> 4) * (p.LQ >> 2) + (s >> 2)) * 64 + q * 16 + (n & 15)) * 4 + (s & 3)), -1);
    }
    for (int l = 1; l < nl; ++l) {
        if (idx >= p.offW[l] && idx < p.offB[l]) {  // W_l[n][k]
            int e = (int)(idx - p.offW[l]);
            int n = e >> 6, k = e & 63;
            if (!p.wave) {
                int sq = k & 15;
                return make_int2(p.pk_wh + (((((l - 1) * 4 + (n >> 4)) * 4 + (sq >> 2)) * 64 + (k >> 4) * 16 + (n & 15)) * 4 + (sq & 3)), -1);
            }
            const int fwd = p.pk_wh + (((((l - 1) * 4 + (n >> 4)) * 4 + (k >> 4)) * 64 + ((k >> 2) & 3) * 16 + (n & 15)) * 4 + (k & 3));
            const int bwd = p.pk_wht + (((((l - 1) * 4 + (k >> 4)) * 4 + (n >> 4)) * 64 + ((n >> 2) & 3) * 16 + (k & 15)) * 4 + (n & 3));
            return make_int2(fwd, bwd);
        }
    }
    if (idx >= p.offW[nl] && idx < p.offB[nl]) {  // W_last[c][k]
        int e = (int)(idx - p.offW[nl]);
        int c = e >> 6, k = e & 63;
        if (!p.wave) {
            int sq = k & 15;
            return make_int2(p.pk_wl + (((sq >> 2) * 64 + (k >> 4) * 16 + c) * 4 + (sq & 3)), -1);
        }
        const int fwd = p.pk_wl + (((k >> 4) * 64 + ((k >> 2) & 3) * 16 + c) * 4 + (k & 3));
        const int bwd = p.pk_wlt + (((k >> 4) * 64 + (c >> 2) * 16 + (k & 15)) * 4 + (c & 3));
        return make_int2(fwd, bwd);
    }
    return make_int2(-1, -1);
}

// slab element (tile order) -> canonical parameter index (or -1: padding)
__device__ __forceinline__ int64_t slab_to_param(int e, const TrainPlan& p, int F, int nl, int C)
{
    if (e >= p.sl_bias) {
        int b = e - p.sl_bias;
        if (b < nl * TBC) return p.offB[b >> 6] + (b & 63);
        int ch = b - nl * TBC;
        return ch < C ? p.offB[nl] + ch : -1;
    }
    const int tile = e >> 8, lane = (e >> 2) & 63, r = e & 3;
    const int row = 4 * (lane >> 4) + r, col = lane & 15;
    if (e < p.sl_hid) {
        int w = tile / p.NT0, nt = tile - w * p.NT0;
        int k = 16 * nt + col;
        return k < p.fm.Fe ? p.offW[0] + (int64_t)(16 * w + row) * F + feat_of_slot(k, p.fm) : -1;
    }
    if (e < p.sl_out) {
        int t2 = tile - 4 * p.NT0;
        int l = 1 + (t2 >> 4), w = (t2 >> 2) & 3, nt = t2 & 3;
        return p.offW[l] + (int64_t)(16 * w + row) * TBC + 16 * nt + col;
    }
    int w = tile - (p.sl_out >> 8);
    return row < C ? p.offW[nl] + (int64_t)row * TBC + 16 * w + col : -1;
}

__global__ void __launch_bounds__(256)
    k_pack_train(const float* __restrict__ params, TrainPlan p, int F, int nl, int C, float* __restrict__ packed)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.NP) return;
    const int2 pos = frag_pos(idx, p, F, nl, C);
    if (pos.x >= 0) packed[pos.x] = params[idx];
    if (pos.y >= 0) packed[pos.y] = params[idx];
}

// map[e] = (canonical parameter index or -1, fragment-order slots or -1) of slab element e: the index
// arithmetic (integer divisions by F and LQ) is done once per epoch call, not in every reduce launch
__global__ void __launch_bounds__(256)
    k_build_map(TrainPlan p, int F, int nl, int C, int4* __restrict__ map)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.slab_floats) return;
    int64_t idx = e < p.sl_bias + nl * TBC + 16 ? slab_to_param(e, p, F, nl, C) : -1;
    const int2 pos = idx >= 0 ? frag_pos(idx, p, F, nl, C) : make_int2(-1, -1);
    map[e] = make_int4((int)idx, pos.x, pos.y, 0);
}

// g[e] = sum over workgroups of slab[wg][e], in a fixed order (slices of nwg/RED_SLICES slabs summed in
// index order, then the slices in index order: bitwise reproducible, no atomics); torch Adam
// (torch/optim/adam.py single-tensor path: lerp_, mul_/addcmul_, addcdiv_); refresh the fragment copy.
// Block = RED_LANES float4 lanes (4*RED_LANES slab elements) x RED_SLICES workgroup slices; the final sum over
// slices and the update are spread over 4*RED_LANES threads, one slab element each.
#ifndef LBDRN_RED_AUX
#define LBDRN_RED_AUX 0x00   // cache policy of k_reduce_adam's slab loads (gfx950: bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
#ifdef LBDRN_TIMELINE
// plain stores to distinct addresses (no atomics: they would serialise and distort what is measured)
constexpr int TL_SLOTS = 2048;   // per step: [0] train start (workgroup 0), [1] reduce start (block 0), [2 .. 2+512) train wave ends, [514 ..) reduce block ends
__device__ __forceinline__ void timeline_store(unsigned long long* tl, int slot, bool drain)
{
    if (!tl || slot >= TL_SLOTS) return;
    unsigned long long t;
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    tl[slot] = t;
}
#endif
struct ReduceFit {
    const float* slabs;
    float *params, *m, *v, *packed;
    const double* loss_part;
    float* loss_out;
};
struct ReduceArgs {
    ReduceFit fit[MAX_GROUP];
    unsigned long long* tl;
    int nloss;   // loss partials per fit (one per training workgroup; the slab count is the kernel's nwg argument)
};

// pairs != 0: the slabs come from k_train_split (train_split.inc) -- two per 64-row group, the chains of its even and odd
// rows: nwg counts GROUPS, slab 2 G plus slab 2 G + 1 is what k_train_stream's workgroup G writes as one slab (it joins
// the same two chains in registers), and the groups are then added exactly as before.  The same bits either way.
// (Fewer lanes per block = more blocks on more CUs was measured for the 256 slabs of k_train_split: 32 / 16 / 8 lanes -- 146 /
// 292 / 584 blocks -- 3.69 / 3.66 / 4.37 us: the launch is not bound by the CUs that share its reads.)
__global__ void __launch_bounds__(RED_SLICES * RED_LANES)
    k_reduce_adam(ReduceArgs R, int nwg, int slab_floats, const int4* __restrict__ map, float step_size, float bc2_sqrt,
                  double loss_count, int pairs)
{
    // 4 KB of LDS, not a byte more: a CU that holds a training workgroup of another fit (159,744 of its 163,840 bytes)
    // has exactly this much left, so the reduce launch of one chain runs BESIDE the training step of another instead
    // of waiting for a CU to come free (with four fits in flight the two launches took turns: 13 ms per tile)
    __shared__ __attribute__((aligned(16))) float part[RED_SLICES][4 * RED_LANES];
    // the chain waits for this launch and it is all latency: where its waves share a SIMD with another kernel's (a
    // fit's background evaluation pass, other fits' small launches) they go first
    __builtin_amdgcn_s_setprio(3);
#ifdef LBDRN_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) timeline_store(R.tl, 1, false);
#endif
    const ReduceFit& F = R.fit[blockIdx.y];   // the fits of a group: same shape, own state
    const float* __restrict__ slabs = F.slabs;
    float* __restrict__ params = F.params;
    float* __restrict__ m = F.m;
    float* __restrict__ v = F.v;
    float* __restrict__ packed = F.packed;
    const int l16 = threadIdx.x % RED_LANES, slice = threadIdx.x / RED_LANES;
    const int base = blockIdx.x * (4 * RED_LANES);
    // the 32 updating threads fetch their parameter's state while the slab reads fly
    int4 me = make_int4(-1, -1, -1, 0);
    float pm = 0.f, pv = 0.f, pp = 0.f;
    if (threadIdx.x < 4 * RED_LANES) {
        me = map[base + threadIdx.x];
        const int j = me.x < 0 ? 0 : me.x;
        pm = m[j]; pv = v[j]; pp = params[j];
    }
    const int per = (nwg + RED_SLICES - 1) / RED_SLICES;
    const int w0 = slice * per, w1 = min(nwg, w0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // (buffer loads: the cache policy of the slab reads is an A/B switch, LBDRN_RED_AUX -- see SLAB_ROT)
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(slabs), (short)0,
                                                                         (int)((size_t)nwg * (pairs ? 2 : 1) * slab_floats * 4), 0x00020000);
    const int voff = (base + 4 * l16) * 4;
    auto slab4 = [&](int slab) -> float4 {
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i x = __builtin_amdgcn_raw_buffer_load_b128(srs, voff + slab * slab_floats * 4, 0, LBDRN_RED_AUX);   // (the slab index differs within a wave: vector offset)
        return make_float4(__int_as_float(x[0]), __int_as_float(x[1]), __int_as_float(x[2]), __int_as_float(x[3]));
    };
    int w = w0;
#ifdef LBDRN_EXP_REDUCE_NOREAD   // (timing only: the reduce launch reads no slab -- what do its reads do to the NEXT training launch's stores?)
    w = w1;
#endif
    if (pairs) {
        for (; w + 8 <= w1; w += 8) {  // eight groups = sixteen loads in flight; a group's two chains first, then the groups in index order
            float4 t[8], o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t[u] = slab4(2 * (w + u));
                o[u] = slab4(2 * (w + u) + 1);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += t[u].x + o[u].x; acc.y += t[u].y + o[u].y; acc.z += t[u].z + o[u].z; acc.w += t[u].w + o[u].w; }
        }
        for (; w < w1; ++w) {
            const float4 t = slab4(2 * w), o = slab4(2 * w + 1);
            acc.x += t.x + o.x; acc.y += t.y + o.y; acc.z += t.z + o.z; acc.w += t.w + o.w;
        }
    }
    for (; w + 16 <= w1; w += 16) {  // sixteen loads in flight, added in index order (8: +0.9 ms per tile, 32: +5 ms)
        float4 t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = slab4(w + u);
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
    }
    for (; w < w1; ++w) {
        const float4 t = slab4(w);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    *reinterpret_cast<float4*>(&part[slice][4 * l16]) = acc;
    __syncthreads();
    if (threadIdx.x < 4 * RED_LANES && me.x >= 0) {
        float g = part[0][threadIdx.x];
#pragma unroll
        for (int s = 1; s < RED_SLICES; ++s) g += part[s][threadIdx.x];
        const float w1c = (float)(1.0 - 0.9), b2 = 0.999f, w2c = (float)(1.0 - 0.999), eps = 1e-8f;
        float mi = pm + w1c * (g - pm);
        float vi = pv * b2 + w2c * (g * g);
        float denom = __builtin_sqrtf(vi) / bc2_sqrt + eps;
        float pi = pp + (-step_size) * (mi / denom);
        m[me.x] = mi;
        v[me.x] = vi;
        params[me.x] = pi;
        if (me.y >= 0) packed[me.y] = pi;
        if (me.z >= 0) packed[me.z] = pi;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && F.loss_out) {
        double s = 0.0;
        if (pairs) for (int k = 0; k + 1 < R.nloss; k += 2) s += F.loss_part[k] + F.loss_part[k + 1];
        else for (int k = 0; k < R.nloss; ++k) s += F.loss_part[k];
        *F.loss_out = (float)(s / loss_count);
    }
#ifdef LBDRN_TIMELINE
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.y == 0) timeline_store(R.tl, 514 + blockIdx.x, true);
#endif
}

// the reduce / Adam launch over one slab per workgroup, or -- pairs -- over k_train_split's two slabs per group
static void launch_reduce(const ReduceArgs& R, int blocks, int count, bool pairs, int nwg, int slab_floats, const int4* map,
                          float step_size, float bc2_sqrt, double loss_count, hipStream_t s)
{
    k_reduce_adam<<<dim3((unsigned)blocks, (unsigned)count), RED_SLICES * RED_LANES, 0, s>>>(R, nwg, slab_floats, map, step_size, bc2_sqrt,
                                                                                            loss_count, pairs ? 1 : 0);
}

// ------------------------------------------------------------------ the fused step

// what differs between the fits of a group (independent fits of one shape stepping side by side in ONE launch:
// blockIdx.y = fit; lbdrn_train_epoch_group)
struct TrainFit {
    const float* rows;      // [N][RP] features | labels
    const int64_t* perm;    // this minibatch's pixel indices
    const float* params;    // canonical
    const float* packed;    // fragment order
    float* slabs;           // [nwg][slab_floats]
    double* loss_part;      // [nwg]
};
struct TrainArgs {
    lbdrn_net net;
    TrainPlan p;
    int64_t npix;
    int batch_n;            // rows in this minibatch
    float inv;              // 1 / (batch_n * C)
    TrainFit fit[MAX_GROUP];
    const float* stage_in;  // (tile kernel) [nwg][32][RP] rows of THIS minibatch, staged by the previous launch (or null)
    float* stage_out;       // (tile kernel) where to stage the NEXT minibatch's rows (or null)
    const int64_t* perm_next;
    int next_n;
    int touch_row_bytes;    // (k_train_stream) bytes at the head of each of the next minibatch's rows the loader wave touches before it ends
    unsigned long long* stamps;  // diagnostic build only (-DLBDRN_TRAIN_STAMPS): [nwg][16] s_memtime
    unsigned long long* tl;      // diagnostic build only (-DLBDRN_TIMELINE): [steps][4] first start / last end of the step's two launches, 100 MHz
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

// 16-byte store that writes through to memory (sc1): the gradient slab and the staged rows are consumed
// by the NEXT launch, never by this one, so they should not sit dirty in this XCD's L2 until the
// end-of-kernel write-back (26 MB per step, measured as ~4 us between the last wave's exit and the kernel's
// completion); written through, they drain while the remaining phases compute.
typedef int v4i32 __attribute__((ext_vector_type(4)));
struct WtBuf {  // buffer descriptor of a wave-uniform region + 16-byte write-through stores into it
    __amdgpu_buffer_rsrc_t r;
    float* base;
    __device__ __forceinline__ WtBuf(float* p, size_t bytes) : base(p)
    {
        r = __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)bytes, 0x00020000);
    }
    __device__ __forceinline__ void store_plain(int float_off, float a, float b, float c, float d) const
    {
        *reinterpret_cast<float4*>(base + float_off) = make_float4(a, b, c, d);
    }
    __device__ __forceinline__ void store(int float_off, float a, float b, float c, float d) const
    {
#ifdef LBDRN_PLAIN_SLAB_STORES
        *reinterpret_cast<float4*>(base + float_off) = make_float4(a, b, c, d);
#else
        v4i32 v = {__float_as_int(a), __float_as_int(b), __float_as_int(c), __float_as_int(d)};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, float_off * 4, 0, 16 /* sc1 */);
#endif
    }
};

// N contiguous floats (N % 4 == 0) from a 16-byte aligned LDS address into registers
template <int N>
__device__ __forceinline__ void lds_load(const float* src, float (&dst)[N])
{
#pragma unroll
    for (int g = 0; g < N / 4; ++g) {
        float4 v = *reinterpret_cast<const float4*>(src + 4 * g);
        dst[4 * g] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
    }
}

template <int LQ, int NL>
__global__ void __launch_bounds__(TRAIN_THREADS, 2) k_train_mfma(TrainArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TrainPlan& p = A.p;
    const TrainFit& Ft = A.fit[0];
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6;
    const int w = w8 & 3, st = w8 >> 2;
    const int i = lane & 15, q = lane >> 4;
    const int C = A.net.C, F = A.net.F;
    constexpr int XP = 4 * LQ + 4;
    float* Xs = lds + p.lds_x;
    float* XT = lds + p.lds_xt;
    float* Hs = lds + p.lds_h;
    float* HT = lds + p.lds_ht;
    float* Zs = lds + p.lds_z;
    float* ZT = lds + p.lds_zt;
    float* Zo = lds + p.lds_zo;
    float* ZoT = lds + p.lds_zot;
    int* pixs = reinterpret_cast<int*>(lds + p.lds_pix);
    double* red = reinterpret_cast<double*>(lds + p.lds_red);
    const int wg = blockIdx.x;
    const int first = wg * TB;
    const int nvalid = min(TB, A.batch_n - first);
    const int srow = i + 16 * st;  // the sample (row of X / H / dZ) this lane's B operands come from
    float* slab = Ft.slabs + (size_t)wg * p.slab_floats;
    const WtBuf slabw(slab, (size_t)p.slab_floats * 4);
#ifdef LBDRN_TRAIN_STAMPS
    unsigned long long stamp[16] = {};
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[12])::"memory");
#endif
    STAMP(0);

    // Where the rows come from.  The random-access gather of a minibatch (8192 x 832 B from a 3.5 GB
    // matrix, ~5 us of exposed latency when done at the top of the step that needs it) is taken off
    // the critical path: launch s loads the rows of minibatch s+1 into registers right at its start,
    // lets the loads fly during its own compute and parks them, contiguous, in a staging buffer at its
    // end; launch s+1 then starts from a coalesced 26 KB read.  Only the first step of an epoch
    // gathers for itself.
    const bool staged = A.stage_in != nullptr;
    int64_t mypix = 0;
    if (!staged && tid < TB) mypix = Ft.perm[first + min(tid, nvalid - 1)];
    const int rs = tid >> 4, rsub = tid & 15;  // row copy: 16 threads per row
    constexpr int NLD = (XP / 4 + 15) / 16;
    const int rp4 = p.RP >> 2;
    // the staged rows are what the first barrier waits for: request them before the (TA-bound, ~5 k cycle)
    // weight prefetch so that they arrive while it is still being issued
    float4 v[NLD];
    if (staged) {
        const float* src = A.stage_in + ((size_t)wg * TB + rs) * p.RP;
#pragma unroll
        for (int u = 0; u < NLD; ++u) v[u] = *reinterpret_cast<const float4*>(src + 4 * min(rsub + 16 * u, rp4 - 1));
    }
    int64_t nextpix = -1;
    // (rows past the end of a short minibatch repeat its last row, like the direct gather does: they
    //  are masked out of the loss but must hold finite values)
    if (A.stage_out && first < A.next_n) nextpix = A.perm_next[min(first + rs, A.next_n - 1)];

    // ---- weight prefetch: everything this wave will multiply by, L2 -> VGPR, before the gather
    float a0[LQ];
    {
        const float4* wf0 = reinterpret_cast<const float4*>(Ft.packed + p.pk_w0) + (size_t)w * (LQ / 4) * 64 + lane;
#pragma unroll
        for (int g = 0; g < LQ / 4; ++g) {
            const float4 t = wf0[g * 64];
            a0[4 * g] = t.x; a0[4 * g + 1] = t.y; a0[4 * g + 2] = t.z; a0[4 * g + 3] = t.w;
        }
    }
    float ah[NL > 1 ? NL - 1 : 1][16];
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        const float4* wfh = reinterpret_cast<const float4*>(Ft.packed + p.pk_wh) + (size_t)((l - 1) * 4 + w) * 4 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = wfh[g * 64];
            ah[l - 1][4 * g] = t.x; ah[l - 1][4 * g + 1] = t.y; ah[l - 1][4 * g + 2] = t.z; ah[l - 1][4 * g + 3] = t.w;
        }
    }
    float al[16];
    {
        const float4* wfl = reinterpret_cast<const float4*>(Ft.packed + p.pk_wl) + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = wfl[g * 64];
            al[4 * g] = t.x; al[4 * g + 1] = t.y; al[4 * g + 2] = t.z; al[4 * g + 3] = t.w;
        }
    }
    float atl[4];  // W_last^T: A[i = hidden 16w+i][k = channel 4q+s]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int ch = 4 * q + s;
        atl[s] = ch < C ? Ft.params[p.offW[NL] + (int64_t)ch * TBC + 16 * w + i] : 0.0f;
    }
    float ath[NL > 1 ? NL - 1 : 1][16];  // W_l^T: A[i = in 16w+i][k = out 16q+s]
#pragma unroll
    for (int l = 1; l < NL; ++l)
#pragma unroll
        for (int s = 0; s < 16; ++s) ath[l - 1][s] = Ft.params[p.offW[l] + (int64_t)(16 * q + s) * TBC + 16 * w + i];
    f32x4 bias[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        float4 b4 = *reinterpret_cast<const float4*>(Ft.params + p.offB[l] + 16 * w + 4 * q);
        bias[l][0] = b4.x; bias[l][1] = b4.y; bias[l][2] = b4.z; bias[l][3] = b4.w;
    }
    f32x4 bias_last;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_last[r] = (4 * q + r) < C ? Ft.params[p.offB[NL] + 4 * q + r] : 0.0f;

    // ---- phase 0: which rows
    if (!staged) {
        if (tid < TB) {
            int64_t pix = mypix < 0 ? 0 : (mypix >= A.npix ? A.npix - 1 : mypix);
            pixs[tid] = (int)pix;
        }
        __syncthreads();
    }
    STAMP(1);
    // ---- phase 1: copy the 32 rows (features | labels) into LDS, row-major (X) and transposed (XT):
    //      16 threads per row, 16 B per load, all loads issued before the first store
    //      (a4: ref LBDRNdataset.py:151-155)
    float4 vnext[NLD];
    {
        const int s = rs, sub = rsub;
        float* xr = Xs + s * XP;
        const int nt16 = 16 * p.NT0;
        if (!staged) {  // first step of an epoch: gather for itself
            const float* src = Ft.rows + (size_t)pixs[s] * p.RP;
#pragma unroll
            for (int u = 0; u < NLD; ++u) v[u] = *reinterpret_cast<const float4*>(src + 4 * min(sub + 16 * u, rp4 - 1));
        }
        // the next minibatch's rows: issued now, consumed at the very end of the kernel
        {
            const int64_t np = nextpix < 0 ? 0 : (nextpix >= A.npix ? A.npix - 1 : nextpix);
            const float* nsrc = Ft.rows + (size_t)np * p.RP;
#pragma unroll
            for (int u = 0; u < NLD; ++u) vnext[u] = *reinterpret_cast<const float4*>(nsrc + 4 * min(sub + 16 * u, rp4 - 1));
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int c4 = sub + 16 * u;
            if (c4 < rp4) {
                *reinterpret_cast<float4*>(xr + 4 * c4) = v[u];
                const float e4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * c4 + e < nt16) XT[(4 * c4 + e) * TP + s] = e4[e];
            }
        }
        for (int f = p.RP + sub; f < XP; f += 16) xr[f] = 0.0f;          // pad columns read by MFMA steps
        for (int f = p.RP + sub; f < nt16; f += 16) XT[f * TP + s] = 0.0f;  // pad rows of the transposed copy
    }
    __syncthreads();
    STAMP(2);

    // ---- phase 2: layer 0, z^T[16w..][16 samples of tile st] = b + W0 X^T, quarter q walks
    //      features q*LQ .. q*LQ+LQ-1
    f32x4 acc = bias[0];
    {
        float bx[LQ];
        lds_load<LQ>(Xs + srow * XP + q * LQ, bx);
#pragma unroll
        for (int s = 0; s < LQ; ++s) acc = MFMA16(a0[s], bx[s], acc);
    }
    STAMP(3);
    // activation; keep cos(30 z) in registers for the backward pass
    f32x4 cs[NL];
    auto activate_store = [&](int l) {
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sn, co;
            canon_sincos(30.0f * acc[r], sn, co);
            hv[r] = sn;
            cs[l][r] = co;
        }
        *reinterpret_cast<float4*>(Hs + (size_t)l * TB * HP + srow * HP + 16 * w + 4 * q) =
            make_float4(hv[0], hv[1], hv[2], hv[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) HT[(size_t)l * TBC * TP + (16 * w + 4 * q + r) * TP + srow] = hv[r];
    };
    activate_store(0);
    __syncthreads();
    STAMP(4);

    // ---- phase 3: hidden layers 1..NL-1 (quarter q walks inputs 16q..16q+15)
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        acc = bias[l];
        float bh[16];
        lds_load<16>(Hs + (size_t)(l - 1) * TB * HP + srow * HP + 16 * q, bh);
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
        float bo[16];
        lds_load<16>(Hs + (size_t)(NL - 1) * TB * HP + srow * HP + 16 * q, bo);
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            oe = MFMA16(al[s], bo[s], oe);
            oo = MFMA16(al[s + 1], bo[s + 1], oo);
        }
        const bool live = srow < nvalid;
        const float* labrow = Xs + srow * XP + F + 4 * q;  // labels ride behind the features
        float dzo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = live && (4 * q + r) < C;
            float y = canon_sigmoid(oe[r] + oo[r]);
            float d = y - (ok ? labrow[r] : 0.0f);
            lsum += ok ? (double)(d * d) : 0.0;                           // ref LBDRNloss.py:9
            dzo[r] = ok ? ((2.0f * d) * A.inv) * (y * (1.0f - y)) : 0.0f;  // mse + sigmoid backward
        }
        *reinterpret_cast<float4*>(Zo + srow * OP + 4 * q) = make_float4(dzo[0], dzo[1], dzo[2], dzo[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) ZoT[(4 * q + r) * TP + srow] = dzo[r];
        for (int o = 32; o > 0; o >>= 1) lsum += __shfl_down(lsum, o);
        if (lane == 0) red[st] = lsum;
    }
    __syncthreads();
    if (tid == 0) Ft.loss_part[wg] = red[0] + red[1];
    STAMP(6);
    // (the loads were issued ~8 k cycles ago; storing here, write-through, lets them drain under the backward pass)
    if (nextpix >= 0) {  // park the next minibatch's rows (loaded at kernel start) in the staging buffer
        const WtBuf stw(A.stage_out + (size_t)wg * TB * p.RP, (size_t)TB * p.RP * 4);
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int c4 = rsub + 16 * u;
            if (c4 < rp4) stw.store(rs * p.RP + 4 * c4, vnext[u].x, vnext[u].y, vnext[u].z, vnext[u].w);
        }
    }

    auto backprop_store = [&](int l) {  // acc = dL/dh_l  ->  dz_l = (dh * cos(30 z)) * 30
        float dz[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[r] = (acc[r] * cs[l][r]) * 30.0f;
        *reinterpret_cast<float4*>(Zs + (size_t)l * TB * HP + srow * HP + 16 * w + 4 * q) =
            make_float4(dz[0], dz[1], dz[2], dz[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) ZT[(size_t)l * TBC * TP + (16 * w + 4 * q + r) * TP + srow] = dz[r];
    };

    // ---- phase 5: dh_{NL-1} = W_last^T dz_out  (K = 16 channel slots, quarter q walks 4q..4q+3)
    {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        acc = zero;
        float bz[4];
        lds_load<4>(Zo + srow * OP + 4 * q, bz);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = MFMA16(atl[s], bz[s], acc);
        backprop_store(NL - 1);
    }
    __syncthreads();
    // ---- phase 6: dh_{l-1} = W_l^T dz_l  (quarter q walks outputs 16q..16q+15)
#pragma unroll
    for (int l = NL - 1; l >= 1; --l) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        acc = zero;
        float bz[16];
        lds_load<16>(Zs + (size_t)l * TB * HP + srow * HP + 16 * q, bz);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = MFMA16(ath[l - 1][s], bz[s], acc);
        backprop_store(l - 1);
        __syncthreads();
    }
    STAMP(7);

    // ---- phase 7: weight gradients, K = 32 samples (8 MFMA steps, quarter q walks samples 8q..8q+7):
    //      dW[16w + row][16nt + col] = sum_s dz[s][16w + row] * in[s][16nt + col]; wave (w, st) takes the
    //      column tiles nt = st, st+2, ...; each finished tile leaves as one 1 KB store in tile order.
    // early = written through (drains under the remaining phases); late tiles use plain stores: a write-
    // through store issued just before the kernel ends only adds its full latency to the tail
    auto grad_tiles = [&](const float* zt, const float* bt, int ntiles, int out, bool early) {
        float az[8];
        lds_load<8>(zt + (16 * w + i) * TP + 8 * q, az);
        for (int nt = st; nt < ntiles; nt += 4) {
            const int nt1 = nt + 2;
            const bool two = nt1 < ntiles;
            float b0[8], b1[8];
            lds_load<8>(bt + (16 * nt + i) * TP + 8 * q, b0);
            lds_load<8>(bt + (16 * (two ? nt1 : nt) + i) * TP + 8 * q, b1);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 g0 = zero, g1 = zero;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                g0 = MFMA16(az[s], b0[s], g0);
                g1 = MFMA16(az[s], b1[s], g1);
            }
            if (early) {
                slabw.store(out + ((w * ntiles + nt) * 64 + lane) * 4, g0[0], g0[1], g0[2], g0[3]);
                if (two) slabw.store(out + ((w * ntiles + nt1) * 64 + lane) * 4, g1[0], g1[1], g1[2], g1[3]);
            } else {
                slabw.store_plain(out + ((w * ntiles + nt) * 64 + lane) * 4, g0[0], g0[1], g0[2], g0[3]);
                if (two) slabw.store_plain(out + ((w * ntiles + nt1) * 64 + lane) * 4, g1[0], g1[1], g1[2], g1[3]);
            }
        }
    };
    grad_tiles(ZT, XT, p.NT0, 0, true);
    STAMP(10);
#pragma unroll
    for (int l = 1; l < NL; ++l)
        grad_tiles(ZT + (size_t)l * TBC * TP, HT + (size_t)(l - 1) * TBC * TP, 4, p.sl_hid + (l - 1) * 16 * 256, false);
    if (st == 0) {  // output layer: rows = channel slots, wave w takes hidden columns 16w..16w+15
        float az[8], bv[8];
        lds_load<8>(ZoT + i * TP + 8 * q, az);
        lds_load<8>(HT + (size_t)(NL - 1) * TBC * TP + (16 * w + i) * TP + 8 * q, bv);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 g0 = zero, g1 = zero;
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            g0 = MFMA16(az[s], bv[s], g0);
            g1 = MFMA16(az[s + 1], bv[s + 1], g1);
        }
        slabw.store_plain(p.sl_out + (w * 64 + lane) * 4, g0[0] + g1[0], g0[1] + g1[1], g0[2] + g1[2], g0[3] + g1[3]);
    }
    STAMP(11);
    // bias gradients: sums over the 32 samples of one unit = one row of the transposed copies
    for (int u = tid; u < NL * TBC + 16; u += TRAIN_THREADS) {
        float zz[TB];
        lds_load<TB>(u < NL * TBC ? ZT + (size_t)u * TP : ZoT + (size_t)(u - NL * TBC) * TP, zz);
        float v = 0.0f;
#pragma unroll
        for (int s = 0; s < TB; ++s) v += zz[s];
        slab[p.sl_bias + u] = v;
    }
    for (int u = p.sl_bias + NL * TBC + 16 + tid; u < p.slab_floats; u += TRAIN_THREADS) slab[u] = 0.0f;
    STAMP(8);
#ifdef LBDRN_TRAIN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(9);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[13])::"memory");
    if (tid == 0 && A.stamps)
        for (int k = 0; k < 16; ++k) A.stamps[(size_t)wg * 16 + k] = stamp[k];
#endif
}

// ------------------------------------------------------------------ the wave-local steps
//
// k_train_stream (train_stream.inc; bc = 64) and k_train_wide (train_wide.inc; bc = 128 / 256): 64 samples per workgroup,
// 4 compute waves (one per SIMD), each wave owns 16 samples from the row copy to dL/dz of the first layer WITHOUT
// exchanging data with another wave: a layer's accumulators (v_mfma_f32_16x16x4_f32: lane (i, q), tile t, register r =
// unit 16t+4q+r of sample i) are exactly the B operand of the next product when that product walks k = 16t+4q+r
// (frag_pos, p.wave), forward and backward alike.  So the whole forward + backpropagation is one dependency-free stream
// per wave -- four independent accumulator tiles per layer (the 40-cycle dependent latency of the 32-cycle MFMA never
// shows), no LDS round trip between layers.  Weight gradients sum over the 64 samples of the workgroup: dz is parked
// transposed in LDS, each wave takes every fourth 16-column strip of the inputs against all four unit tiles.  A
// minibatch of 8192 rows is 128 workgroups, one per CU on half of the chip: two fits in flight run side by side.
// (Round 2's first kernel of this family, k_train_wave, waited for all of its rows and layer-0 fragments before its
// first MFMA; k_train_stream replaced it in round 3 -- the history keeps it.)
// sigmoid, sin and cos of the training step: a tolerance contract (1e-5 relative on the loss), so the hardware's
// transcendentals behind a compensated reduction (lbdrn_math.hpp: fast_sigmoid, fast_sincos; 4.5e-7 absolute against 1e-7
// for the canonical polynomials, 7 instructions against 31); the decode kernels keep the canonical arithmetic.
// -DLBDRN_TRAIN_CANON_SINCOS puts the canonical pair back (A/B).
__device__ __forceinline__ float train_sigmoid(float z) { return fast_sigmoid(z); }
// The hidden activation of the fused steps and its derivative, as a template parameter of k_train_stream / k_train_split:
// ACT = LBDRN_ACT_SINE: h = sin(30 z), dh/dz = 30 cos(30 z) (d holds the cosine);  LBDRN_ACT_RELU (torch.nn.ReLU, the
// alternative the reference names at encode.py:75 / decode.py:108): h = z where z > 0, else 0; d = 1 / 0 in the slot of the
// cosine and dz = dh where d != 0 (threshold_backward: what generic.hip's BackDx does)
template <int ACT> __device__ __forceinline__ void train_act(float z, float& h, float& d);
template <int ACT> __device__ __forceinline__ float train_act_back(float dh, float d)
{
    if constexpr (ACT == LBDRN_ACT_RELU) return d != 0.0f ? dh : 0.0f;
    else return (dh * d) * 30.0f;
}
#ifdef LBDRN_TRAIN_CANON_SINCOS
__device__ __forceinline__ void train_sincos(float x, float& sn, float& cs) { canon_sincos(x, sn, cs); }
#else
__device__ __forceinline__ void train_sincos(float x, float& sn, float& cs) { fast_sincos(x, sn, cs); }
#endif
template <int ACT> __device__ __forceinline__ void train_act(float z, float& h, float& d)
{
    if constexpr (ACT == LBDRN_ACT_RELU) { h = z > 0.0f ? z : 0.0f; d = z > 0.0f ? 1.0f : 0.0f; }
    else train_sincos(30.0f * z, h, d);
}

// sum over the 64 lanes, same order every time, no LDS: rows of 16 by DPP shifts, then the four row totals
// sum over each row of 16 lanes (DPP shifts, zero fill): lane 15 of a row holds the row's total
__device__ __forceinline__ float row_sum16(float x)
{
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
    return x;
}

// sum over the 64 lanes, same order every time, no LDS: the four row totals, then (0 + 1) + (2 + 3)
__device__ __forceinline__ float wave_sum(float x)
{
    const int xi = __float_as_int(row_sum16(x));
    const float a = __int_as_float(__builtin_amdgcn_readlane(xi, 15)), b = __int_as_float(__builtin_amdgcn_readlane(xi, 31));
    const float c = __int_as_float(__builtin_amdgcn_readlane(xi, 47)), d = __int_as_float(__builtin_amdgcn_readlane(xi, 63));
    return (a + b) + (c + d);
}

// the dynamic-LDS ceiling of a kernel is told to the runtime once per device and kernel (a cache of an idempotent
// setting; several host threads may get here together)
template <class K>
static int configure_lds_once(K kern, int bytes, std::atomic<unsigned long long>& configured)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(configured.load(std::memory_order_relaxed) & bit)) {
        LBDRN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    return 0;
}

// measurement aid (see lbdrn_hip.h): mode 1 doubles the reduce/Adam launch of every step, mode 2 the training launch,
// mode 3 leaves the reduce/Adam launch out (an epoch of training launches back to back, each on its own rows); the split
// wide step (train_wide.inc) has two launches in front of the reduce: mode 3 runs both, mode 4 the forward/backward launch
// alone, mode 5 doubles the weight-gradient launch
static thread_local int g_prof_mode = 0;
int train_profile_mode(int mode)
{
    g_prof_mode = mode;
    return 0;
}

#include "train_stream.inc"

static int stream_lds_total(int LQ, int NL) { return stream_lds(LQ, NL).total; }

#include "train_split.inc"

#include "train_wide.inc"

int mfma_train_step_features(const lbdrn_geom& g, const lbdrn_net& net)
{
    TrainPlan p;
    if (make_train_plan(g, net, &p)) return p.fm.Fe;
    WidePlan wp;
    return make_wide_plan(g, net, &wp) ? wide_plan_features(wp) : net.F;
}

static bool wide_supported(const lbdrn_geom& g, const lbdrn_net& net)
{
    WidePlan p;
    return make_wide_plan(g, net, &p);
}
static size_t wide_workspace(const lbdrn_geom& g, const lbdrn_net& net, int bs)
{
    WidePlan p;
    if (!make_wide_plan(g, net, &p)) return 0;
    return wide_ws_layout(g, net, p, bs).total;
}

// ------------------------------------------------------------------ host driver

template <int LQ, int NL>
static int launch_train(const TrainArgs& A, int nwg, hipStream_t s)
{
    auto kern = k_train_mfma<LQ, NL>;
    const size_t lds_bytes = (size_t)A.p.lds_floats * 4;
    static std::atomic<unsigned long long> configured{0};
    if (int rc = configure_lds_once(kern, 160 * 1024, configured)) return rc;
    kern<<<nwg, TRAIN_THREADS, lds_bytes, s>>>(A);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

template <int LQ>
static int dispatch_nl(const TrainArgs& A, int nwg, hipStream_t s)
{
    if (A.net.nl == 1) return launch_train<LQ, 1>(A, nwg, s);
    if (A.net.nl == 2) return launch_train<LQ, 2>(A, nwg, s);
    return launch_train<LQ, 3>(A, nwg, s);
}

template <int LQ, int NL, int NT0C, int ACT>
static int launch_stream_act(const TrainArgs& A, int nwg, int count, hipStream_t s)
{
    auto kern = k_train_stream<LQ, NL, LBDRN_STREAM_PD, NT0C, ACT>;
    static std::atomic<unsigned long long> configured{0};
    if (int rc = configure_lds_once(kern, A.p.wave_lds_floats * 4, configured)) return rc;
    kern<<<dim3((unsigned)nwg, (unsigned)count), STREAM_THREADS, (size_t)A.p.wave_lds_floats * 4, s>>>(A);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

template <int LQ, int NL, int NT0C>
static int launch_stream(const TrainArgs& A, int nwg, int count, hipStream_t s)
{
    return A.net.act == LBDRN_ACT_RELU ? launch_stream_act<LQ, NL, NT0C, LBDRN_ACT_RELU>(A, nwg, count, s)
                                       : launch_stream_act<LQ, NL, NT0C, LBDRN_ACT_SINE>(A, nwg, count, s);
}

static int dispatch_stream(const TrainArgs& A, int nwg, int count, hipStream_t s)
{
    const bool one = A.net.nl == 1;
    // the shapes BASELINE.json names (F = 200: 12 strips of features that can differ from zero, F = 250: 16) and the
    // reference's 4-band shape (F = 100: 6) run the straight-line weight-gradient schedule (-DLBDRN_EXP_STREAM_LOOP: A/B
    // build that keeps them on the loop)
#ifdef LBDRN_EXP_STREAM_LOOP
    constexpr bool loop_only = true;
#else
    constexpr bool loop_only = false;
#endif
    switch (A.p.LQ) {
        case 16: return one ? launch_stream<16, 1, 0>(A, nwg, count, s) : launch_stream<16, 2, 0>(A, nwg, count, s);
        case 24:
            if (!one && A.p.NT0 == 6 && !loop_only) return launch_stream<24, 2, 6>(A, nwg, count, s);
            return one ? launch_stream<24, 1, 0>(A, nwg, count, s) : launch_stream<24, 2, 0>(A, nwg, count, s);
        case 32: return one ? launch_stream<32, 1, 0>(A, nwg, count, s) : launch_stream<32, 2, 0>(A, nwg, count, s);
        case 48:
            if (!one && A.p.NT0 == 12 && !loop_only) return launch_stream<48, 2, 12>(A, nwg, count, s);
            return one ? launch_stream<48, 1, 0>(A, nwg, count, s) : launch_stream<48, 2, 0>(A, nwg, count, s);
        case 52:
            if (!one && A.p.NT0 == 13 && !loop_only) return launch_stream<52, 2, 13>(A, nwg, count, s);
            return one ? launch_stream<52, 1, 0>(A, nwg, count, s) : launch_stream<52, 2, 0>(A, nwg, count, s);
        default:
            if (!one && A.p.NT0 == 16 && !loop_only) return launch_stream<64, 2, 16>(A, nwg, count, s);
            return one ? launch_stream<64, 1, 0>(A, nwg, count, s) : launch_stream<64, 2, 0>(A, nwg, count, s);
    }
}

// the shapes that have k_train_split beside k_train_stream (same bits: train_split.inc) -- the two BASELINE.json names and
// the 4-band shape of the reference's own image list (run.sh:14-28)
static bool split_available(const TrainPlan& p, const lbdrn_net& net)
{
    return p.wave == 2 && net.nl == 2 && ((p.LQ == 48 && p.NT0 == 12) || (p.LQ == 64 && p.NT0 == 16) || (p.LQ == 24 && p.NT0 == 6));
}

template <int LQ, int NT0C, int ACT>
static int launch_split_act(const TrainArgs& A, int nwg, int count, hipStream_t s)
{
    auto kern = k_train_split<LQ, NT0C, ACT>;
    constexpr int bytes = split_lds(LQ).total * 4;
    static std::atomic<unsigned long long> configured{0};
    if (int rc = configure_lds_once(kern, bytes, configured)) return rc;
    kern<<<dim3((unsigned)nwg, (unsigned)count), WAVE_THREADS, (size_t)bytes, s>>>(A);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

template <int LQ, int NT0C>
static int launch_split(const TrainArgs& A, int nwg, int count, hipStream_t s)
{
    return A.net.act == LBDRN_ACT_RELU ? launch_split_act<LQ, NT0C, LBDRN_ACT_RELU>(A, nwg, count, s)
                                       : launch_split_act<LQ, NT0C, LBDRN_ACT_SINE>(A, nwg, count, s);
}

static int dispatch_split(const TrainArgs& A, int nwg, int count, hipStream_t s)
{
    return A.p.LQ == 48 ? launch_split<48, 12>(A, nwg, count, s) : A.p.LQ == 24 ? launch_split<24, 6>(A, nwg, count, s) : launch_split<64, 16>(A, nwg, count, s);
}

static int dispatch_train(const TrainArgs& A, int nwg, int count, hipStream_t s)
{
    if (A.p.wave == 2) return dispatch_stream(A, nwg, count, s);
    switch (A.p.LQ) {
        case 16: return dispatch_nl<16>(A, nwg, s);
        case 32: return dispatch_nl<32>(A, nwg, s);
        case 52: return dispatch_nl<52>(A, nwg, s);
        default: return dispatch_nl<64>(A, nwg, s);
    }
}

// Build the per-image state of the fused training path in the caller's workspace: the
// [N][F+C] row matrix.  Must run once per image before mfma_train_epoch.
int mfma_train_prepare(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                       const uint16_t* msb, int bs, void* ws, size_t ws_bytes, hipStream_t s)
{
    TrainPlan p;
    size_t need = 0, off_rows = 0;
    int LQs = 0;   // > 0: the streamed step's row order (row_source)
    if (make_train_plan(g, net, &p)) {
        const TrainWsLayout L = train_ws_layout(g, net, p, bs);
        need = L.total; off_rows = L.off_rows;
        if (p.wave == 2) LQs = p.LQ;
    } else {
        WidePlan wp;
        if (!make_wide_plan(g, net, &wp)) {
            set_error("shape not supported by the MFMA train kernels");
            return LBDRN_E_UNSUPPORTED;
        }
        const WideWsLayout L = wide_ws_layout(g, net, wp, bs);
        need = L.total; off_rows = L.off_rows;
        p.RP = wp.RP;
        p.fm = wp.fm;
    }
    if (!ws || ws_bytes < need) {
        set_error("train workspace too small: %zu < %zu", ws_bytes, need);
        return LBDRN_E_WORKSPACE;
    }
    float* rows = (float*)((char*)ws + off_rows);
    const int64_t total = (int64_t)g.H * g.W * p.RP;
    LBDRN_REQUIRE((total + 255) / 256 < ((int64_t)1 << 31), "image too large for one launch");
    const int side = 2 * g.D + 1;
    const size_t ncol = (size_t)std::max(net.F - 2 * g.P, 0);
    const size_t tile_lds = ((size_t)g.C * side * (BR_TW + 2 * g.D) + 2 * ncol + 2 * (size_t)p.RP + (size_t)g.C * BR_TW) * 4;   // window + offsets + position table + labels
    const int64_t nblk = (int64_t)g.H * ((g.W + BR_TW - 1) / BR_TW);
#ifdef LBDRN_EXP_PREPARE_DIAG   // (timing-only build, never the shipped library: no row matrix is built, the fit trains on whatever the buffer holds)
    return 0;
#endif
    if (tile_lds <= 48 * 1024 && g.D < g.H && g.D < g.W && nblk < ((int64_t)1 << 31)) {
        k_build_rows_tiled<<<(unsigned)nblk, 256, tile_lds, s>>>(g, net.F, p.RP, LQs, p.fm, msb, img, rows);
    } else {
        k_build_rows<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(g, net.F, p.RP, LQs, p.fm, msb, img, rows);
    }
    LBDRN_LAUNCH_CHECK();
    return 0;
}


// One epoch of `count` fits of one shape, stepping side by side (count == 1: the plain call).  Groups of more than
// one fit run on the streamed step only: minibatch s of every fit is ONE launch of count x nwg workgroups
// (blockIdx.y = fit) and one reduce launch -- two fits fill the chip's 256 CUs exactly, instead of two 128-workgroup
// launches of independent chains meeting each other by chance.  Every fit keeps its own workspace, state and slabs:
// the numbers are those of `count` separate calls, bit for bit.
int mfma_train_epoch_group(int count, const lbdrn_geom& g, const lbdrn_net& net, const int64_t* const* perm, int64_t n,
                           int bs, float* const* params, float* const* m, float* const* v, int64_t step0, double lr,
                           float* const* losses, void* const* ws, size_t ws_bytes, hipStream_t s, bool alone)
{
    TrainArgs A;
    if (!make_train_plan(g, net, &A.p)) {
        if (count != 1) { set_error("groups of fits run on the bc = 64 fused step only"); return LBDRN_E_UNSUPPORTED; }
        return wide_train_epoch(g, net, perm[0], n, bs, params[0], m[0], v[0], step0, lr, losses ? losses[0] : nullptr,
                                ws[0], ws_bytes, s);
    }
    if (count < 1 || count > MAX_GROUP || (count > 1 && A.p.wave != 2)) {
        set_error("a group of %d fits is not supported by this shape's train kernel", count);
        return LBDRN_E_UNSUPPORTED;
    }
    if (!slabs_addressable(A.p, bs)) {
        set_error("minibatch of %d rows: one step's gradient slabs pass 2 GiB, the fused step does not address them", bs);
        return LBDRN_E_UNSUPPORTED;
    }
    const TrainWsLayout L = train_ws_layout(g, net, A.p, bs);
    if (ws_bytes < L.total) {
        set_error("train workspace too small: %zu < %zu", ws_bytes, L.total);
        return LBDRN_E_WORKSPACE;
    }
    A.net = net; A.npix = (int64_t)g.H * g.W;
    ReduceArgs R;
    for (int f = 0; f < MAX_GROUP; ++f) {
        const int k = f < count ? f : 0;   // (unused slots repeat fit 0: never read)
        if (!ws[k]) { set_error("null train workspace"); return LBDRN_E_WORKSPACE; }
        float* packed = (float*)((char*)ws[k] + L.off_pack);
        A.fit[f].rows = (float*)((char*)ws[k] + L.off_rows);
        A.fit[f].params = params[k];
        A.fit[f].packed = packed;
        A.fit[f].slabs = (float*)((char*)ws[k] + L.off_slab);
        A.fit[f].loss_part = (double*)((char*)ws[k] + L.off_loss);
        A.fit[f].perm = perm[k];
        R.fit[f].slabs = A.fit[f].slabs; R.fit[f].params = params[k]; R.fit[f].m = m[k]; R.fit[f].v = v[k];
        R.fit[f].packed = packed; R.fit[f].loss_part = A.fit[f].loss_part; R.fit[f].loss_out = nullptr;
        if (f < count) {
            LBDRN_HIP_TRY(hipMemsetAsync(packed, 0, (size_t)A.p.pack_floats * sizeof(float), s));
            k_pack_train<<<(unsigned)((A.p.NP + 255) / 256), 256, 0, s>>>(params[k], A.p, net.F, net.nl, net.C, packed);
            LBDRN_LAUNCH_CHECK();
        }
    }
    int4* map = (int4*)((char*)ws[0] + L.off_map);   // slab element -> parameter: a function of the shape only
    k_build_map<<<(unsigned)((A.p.slab_floats + 255) / 256), 256, 0, s>>>(A.p, net.F, net.nl, net.C, map);
    LBDRN_LAUNCH_CHECK();
    float* stage[2] = {(float*)((char*)ws[0] + L.off_stage), (float*)((char*)ws[0] + L.off_stage + L.stage_bytes)};
    A.stamps = nullptr;
#ifdef LBDRN_TRAIN_STAMPS
    const int max_wg = (bs + TB - 1) / TB;
    LBDRN_HIP_TRY(hipMalloc(&A.stamps, (size_t)max_wg * 8 * 16 * sizeof(unsigned long long)));
#endif
    A.tl = nullptr; R.tl = nullptr;
#ifdef LBDRN_TIMELINE
    const int tl_steps = (int)((n + bs - 1) / bs);
    unsigned long long* tl_buf = nullptr;
    {
        LBDRN_HIP_TRY(hipMalloc(&tl_buf, (size_t)tl_steps * TL_SLOTS * sizeof(unsigned long long)));
        LBDRN_HIP_TRY(hipMemset(tl_buf, 0, (size_t)tl_steps * TL_SLOTS * sizeof(unsigned long long)));
    }
#endif
#ifndef LBDRN_TOUCH_ROW_BYTES
#define LBDRN_TOUCH_ROW_BYTES 256   // (A/B builds: 0 .. the row's bytes, a multiple of 4; performance only, the numbers do not depend on it)
#endif
    constexpr int touch_alone = LBDRN_TOUCH_ROW_BYTES;
    const int rows_per_wg = A.p.wave ? WB : TB;
    // A fit that has the device to itself (LBDRN_TRAIN_ALONE) steps on k_train_split where the shape has it: 256 workgroups
    // of 32 rows, every CU, two slabs per 64-row group.  The numbers are k_train_stream's bit for bit (train_split.inc).
#ifdef LBDRN_EXP_NO_SPLIT   // (A/B build: the lone fit stays on k_train_stream)
    const bool split = false;
#else
#ifdef LBDRN_EXP_SPLIT_ALIAS_LDS   // (timing only: every launch on k_train_split, groups included -- two workgroups per CU)
    const bool split = split_available(A.p, net);
#else
    const bool split = alone && count == 1 && split_available(A.p, net);
#endif
#endif
    const int red_blocks = A.p.slab_floats / (4 * RED_LANES);
    int64_t step = step0;
    int si = 0;
    // diagnostic (-DLBDRN_HOST_TRACE build): how long the host spends in each iteration of the launch loop
#ifdef LBDRN_HOST_TRACE
    constexpr bool host_trace = true;
#else
    constexpr bool host_trace = false;
#endif
    std::vector<std::pair<int, double>> slow;
    double host_total = 0.0;
    auto tprev = std::chrono::steady_clock::now();
    const auto tbegin = tprev;
    for (int64_t first = 0; first < n; first += bs, ++si) {
        if (host_trace) {
            const auto tn = std::chrono::steady_clock::now();
            const double us = std::chrono::duration<double, std::micro>(tn - tprev).count();
            host_total += us;
            if (us > 40.0) slow.emplace_back(si, us);
            tprev = tn;
        }
        const int B = (int)std::min<int64_t>(bs, n - first);
        const int nwg = (B + rows_per_wg - 1) / rows_per_wg;
        for (int f = 0; f < MAX_GROUP; ++f) {
            const int k = f < count ? f : 0;
            A.fit[f].perm = perm[k] + first;
            R.fit[f].loss_out = (f < count && losses && losses[k]) ? losses[k] + si : nullptr;
        }
        A.batch_n = B;
        for (int f = 0; f < MAX_GROUP; ++f) {   // this step's slab buffer
            const int k = f < count ? f : 0;
            A.fit[f].slabs = (float*)((char*)ws[k] + L.off_slab + (size_t)(si % SLAB_ROT) * L.slab_bytes);
            R.fit[f].slabs = A.fit[f].slabs;
        }
#ifdef LBDRN_TIMELINE
        A.tl = tl_buf + (size_t)TL_SLOTS * si; R.tl = A.tl;
#endif
        A.inv = 1.0f / ((float)B * (float)net.C);
        const int64_t nextB = std::max<int64_t>(0, std::min<int64_t>(bs, n - first - bs));
        A.stage_in = si > 0 ? stage[si & 1] : nullptr;          // staged by the previous launch
        A.stage_out = nextB > 0 ? stage[(si + 1) & 1] : nullptr;
        if (A.p.wave) A.stage_in = nullptr, A.stage_out = nullptr;   // the wave-local kernels gather for themselves
        A.perm_next = perm[0] + first + bs;
        A.next_n = (int)nextB;
        A.touch_row_bytes = (alone && count == 1) ? touch_alone : 0;   // (LBDRN_TRAIN_ALONE; see k_train_stream's loader wave)
        // (split: nwg counts the 64-row GROUPS -- the unit k_reduce_adam adds in its fixed order --, two workgroups each)
        const bool split_now = split && B >= 2;   // (k_train_split fetches the pixel indices two at a time: a minibatch of ONE row -- the tail of an
                                                  // epoch of n = 1 (mod batch size) rows -- steps on k_train_stream; the same bits)
        auto train_launch = [&]() { return split_now ? dispatch_split(A, 2 * nwg, count, s) : dispatch_train(A, nwg, count, s); };
        if (int rc = train_launch()) return rc;
        if (g_prof_mode == 2)   // measurement only: the same launch again (it writes the same slabs and loss partials)
            if (int rc = train_launch()) return rc;
        ++step;
        const double bc1 = 1.0 - std::pow(0.9, (double)step), bc2 = 1.0 - std::pow(0.999, (double)step);
        R.nloss = split_now ? 2 * nwg : nwg;
        if (g_prof_mode != 3 && g_prof_mode != 4)   // (modes 3 / 4, measurement only: the training launches alone, every one on its own slice of rows)
                launch_reduce(R, red_blocks, count, split_now, nwg, A.p.slab_floats, map, (float)(lr / bc1), (float)std::sqrt(bc2), (double)B * net.C, s);
        if (g_prof_mode == 1) {  // measurement only: the same launch again with a zero step
            ReduceArgs R0 = R;
            for (int f = 0; f < MAX_GROUP; ++f) R0.fit[f].loss_out = nullptr;
            launch_reduce(R0, red_blocks, count, split_now, nwg, A.p.slab_floats, map, 0.0f, (float)std::sqrt(bc2), (double)B * net.C, s);
        }
        LBDRN_LAUNCH_CHECK();
    }
#ifdef LBDRN_TIMELINE
    {
        LBDRN_HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)tl_steps * TL_SLOTS);
        LBDRN_HIP_TRY(hipMemcpy(h.data(), tl_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(tl_buf);
        double tr = 0, g1 = 0, rd = 0, g2 = 0; int cnt = 0;
        auto last = [&](int k, int lo, int hi) { unsigned long long m = 0; for (int j = lo; j < hi; ++j) m = std::max(m, h[(size_t)k * TL_SLOTS + j]); return m; };
        for (int k = 8; k + 1 < tl_steps; ++k) {   // (the first steps of an epoch call start behind its packing launches)
            const unsigned long long ts = h[(size_t)k * TL_SLOTS], rs = h[(size_t)k * TL_SLOTS + 1], te = last(k, 2, 514), re = last(k, 514, TL_SLOTS);
            const unsigned long long ts2 = h[(size_t)(k + 1) * TL_SLOTS];
            tr += (double)(te - ts); g1 += (double)((long long)(rs - te)); rd += (double)(re - rs); g2 += (double)((long long)(ts2 - re));
            ++cnt;
        }
        const double u = 0.01 / std::max(cnt, 1);   // 100 MHz ticks -> us
        fprintf(stderr, "[lbdrn timeline] per step over %d steps: train first start -> last end %.2f us | -> reduce starts %.2f | reduce %.2f | -> next train starts %.2f | step %.2f us\n",
                cnt, tr * u, g1 * u, rd * u, g2 * u, (tr + g1 + rd + g2) * u);
    }
#endif
    if (host_trace) {
        fprintf(stderr, "[lbdrn host trace] %d iterations in %.0f us (%.2f us each); slower than 40 us:", si,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tbegin).count(), host_total / std::max(si, 1));
        for (auto& e : slow) fprintf(stderr, " %d:%.0f", e.first, e.second);
        fprintf(stderr, "\n");
    }
#ifdef LBDRN_TRAIN_STAMPS
    if (A.p.wave) {   // diagnostic: mean cycles per phase over the waves of the last step
        LBDRN_HIP_TRY(hipStreamSynchronize(s));
        const int nw = ((int)std::min<int64_t>(bs, n) + rows_per_wg - 1) / rows_per_wg * 4 * (A.p.wave == 2 ? count : 1) * (split ? 2 : 1);
        std::vector<unsigned long long> h((size_t)nw * 16);
        LBDRN_HIP_TRY(hipMemcpy(h.data(), A.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(A.stamps);
        double d[14] = {}, clk = 0, span = 0;
        unsigned long long t0min = ~0ull, t1max = 0;
        for (int k = 0; k < nw; ++k) {
            for (int j = 1; j <= 13; ++j) d[j] += (double)(h[k * 16 + j] - h[k * 16 + j - 1]);
            clk += (double)(h[k * 16 + 13] - h[k * 16 + 0]) / (double)(h[k * 16 + 15] - h[k * 16 + 14]) * 100.0;
            span += (double)(h[k * 16 + 13] - h[k * 16 + 0]);
            t0min = std::min(t0min, h[k * 16 + 14]);
            t1max = std::max(t1max, h[k * 16 + 15]);
        }
        if (split)
            fprintf(stderr, "[lbdrn stamps, split kernel] clock %.0f MHz; wave lifetime %.0f cycles; first start -> last end %.2f us; "
                            "mean cycles: requests out %.0f | layer 0 (waits for W0 and the rows in it) %.0f | act0 %.0f | barrier, h0 read %.0f | hidden + act1 %.0f | "
                            "barrier, h1 read %.0f | out + loss %.0f | dh1, dz1 parked %.0f | barrier, dh0, dz0 parked %.0f | barrier, gradient operands read %.0f | "
                            "gradient products, stores, bias sums %.0f | drain %.0f\n",
                    clk / nw, span / nw, (double)(t1max - t0min) / 100.0, d[1] / nw, d[2] / nw, d[3] / nw, d[4] / nw, d[5] / nw, d[6] / nw,
                    d[7] / nw, d[8] / nw, d[9] / nw, d[10] / nw, (d[11] + d[12]) / nw, d[13] / nw);
        else
        fprintf(stderr, A.p.wave == 2 ?
                        "[lbdrn stamps, stream kernel] clock %.0f MHz; wave lifetime %.0f cycles; first start -> last end %.2f us; "
                        "mean cycles: indices + first requests out %.0f | first stage landed %.0f | layer 0 under the stream %.0f | all landed, barrier %.0f | "
                        "act0 %.0f | hidden+act %.0f | out+loss %.0f | backward %.0f | barrier 3 %.0f | dW0 strips %.0f | dW tail+hidden %.0f | "
                        "bias sums %.0f | drain %.0f\n" :
                        "[lbdrn stamps, wave kernel] clock %.0f MHz; wave lifetime %.0f cycles; first start -> last end %.2f us; "
                        "mean cycles: W0 DMA + rows->LDS %.0f | barrier 1 %.0f | small-matrix requests + layer0 %.0f | barrier 2 %.0f | "
                        "act0 %.0f | hidden+act %.0f | out+loss %.0f | backward %.0f | barrier 3 %.0f | dW0 strips %.0f | dW tail+hidden %.0f | "
                        "bias sums %.0f | drain %.0f\n",
                clk / nw, span / nw, (double)(t1max - t0min) / 100.0, d[1] / nw, d[2] / nw, d[3] / nw, d[4] / nw, d[5] / nw, d[6] / nw,
                d[7] / nw, d[8] / nw, d[9] / nw, d[10] / nw, d[11] / nw, d[12] / nw, d[13] / nw);
    } else {   // diagnostic: mean cycles per phase over the workgroups of the last step
        LBDRN_HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)max_wg * 16);
        LBDRN_HIP_TRY(hipMemcpy(h.data(), A.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(A.stamps);
        double sum[12] = {}, d10 = 0, d11 = 0, d8 = 0, clk = 0;
        for (int wq = 0; wq < max_wg; ++wq) {
            for (int k = 1; k < 10; ++k)
                if (k != 8) sum[k] += (double)(h[wq * 16 + k] - h[wq * 16 + (k == 9 ? 8 : k - 1)]);
            d10 += (double)(h[wq * 16 + 10] - h[wq * 16 + 7]);
            d11 += (double)(h[wq * 16 + 11] - h[wq * 16 + 10]);
            d8 += (double)(h[wq * 16 + 8] - h[wq * 16 + 11]);
            clk += (double)(h[wq * 16 + 9] - h[wq * 16 + 0]) / (double)(h[wq * 16 + 13] - h[wq * 16 + 12]) * 100.0;
        }
        fprintf(stderr, "[lbdrn stamps] in-kernel clock %.0f MHz; mean cycles/phase: perm %.0f rows %.0f L0 %.0f act0 %.0f "
                        "hidden %.0f out+loss %.0f backprop %.0f dW0 %.0f dWhid+out %.0f bias %.0f drain %.0f\n",
                clk / max_wg, sum[1] / max_wg, sum[2] / max_wg, sum[3] / max_wg, sum[4] / max_wg, sum[5] / max_wg,
                sum[6] / max_wg, sum[7] / max_wg, d10 / max_wg, d11 / max_wg, d8 / max_wg, sum[9] / max_wg);
    }
#endif
    return 0;
}

int mfma_train_epoch(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                     const uint16_t* msb, const int64_t* perm, int64_t n, int bs, float* params,
                     float* m, float* v, int64_t step0, double lr, float* losses, void* ws,
                     size_t ws_bytes, hipStream_t s, bool alone)
{
    return mfma_train_epoch_group(1, g, net, &perm, n, bs, &params, &m, &v, step0, lr, &losses, &ws, ws_bytes, s, alone);
}

}  // namespace lbdrn
