// Generic (any-shape) HIP kernels of the LBDRN hot path: bit split, label / feature matrices,
// a k-ordered MFMA-tiled f32 GEMM with fused epilogues for forward / backward, loss, Adam, whole-image
// squared error and integer reconstruction.  These serve every configuration the reference
// accepts (any bc, nl, C, D, constants.py switches); the fused MFMA kernels in apply_mfma.hip /
// train_mfma.hip take over for the shapes they support and must reproduce these bit for bit
// (forward) or within 1e-5 (training).
//
// Summation order of every forward dot product: accumulator starts at the bias, k ascending,
// one fmaf per term -- identical to an f32 MFMA chain and to oracle/lbdrn_oracle.c.
#include <stdarg.h>

#include <cmath>

#include "common.hpp"
#include "lbdrn_math.hpp"

namespace lbdrn {

// ------------------------------------------------------------------ error string

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

int check_geom(const lbdrn_geom* g)
{
    LBDRN_REQUIRE(g != nullptr, "geom is null");
    LBDRN_REQUIRE(g->C >= 1 && g->H >= 1 && g->W >= 1, "bad image shape C=%d H=%d W=%d", g->C,
                  g->H, g->W);
    LBDRN_REQUIRE((int64_t)g->H * g->W < (int64_t)1 << 31, "image too large");
    LBDRN_REQUIRE(g->K >= 1 && g->K <= 15, "K=%d outside [1,15] (header nibble, encode.py:55)", g->K);
    LBDRN_REQUIRE(g->D >= 0 && g->D <= 15, "D=%d outside [0,15] (header nibble, encode.py:55)", g->D);
    LBDRN_REQUIRE(g->P >= 0, "P=%d negative", g->P);
    LBDRN_REQUIRE(g->P == 0 || (g->rowtab && g->coltab), "P>0 needs rowtab and coltab");
    LBDRN_REQUIRE(feature_dim(*g) >= 1, "feature dimension is zero (USE_COLORS and USE_COORDINATES both off)");
    return 0;
}
int check_net(const lbdrn_net* n)
{
    LBDRN_REQUIRE(n != nullptr, "net is null");
    LBDRN_REQUIRE(n->F >= 1 && n->bc >= 1 && n->C >= 1 && n->nl >= 1,
                  "bad net F=%d bc=%d C=%d nl=%d", n->F, n->bc, n->C, n->nl);
    LBDRN_REQUIRE(n->act == LBDRN_ACT_SINE || n->act == LBDRN_ACT_RELU, "unknown hidden activation %d", n->act);
    return 0;
}

// ------------------------------------------------------------------ a1: bit split / labels

__global__ void __launch_bounds__(256) k_split_bits(const uint16_t* __restrict__ img, int64_t total,
                                                    int K, uint16_t* __restrict__ msb, int* mx)
{
    int local = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int hi = (int)img[i] >> K;  // ref LBDRNdataset.py:95
        if (msb) msb[i] = (uint16_t)hi;
        local = max(local, hi);
    }
    for (int o = 32; o > 0; o >>= 1) local = max(local, __shfl_down(local, o));
    __shared__ int red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = max(max(red[0], red[1]), max(red[2], red[3]));
        atomicMax(mx, m);  // integer max: order-independent, so still deterministic
    }
}

int generic_split_bits(const uint16_t* img, int C, int H, int W, int K, uint16_t* msb, int32_t* mx,
                       hipStream_t s)
{
    int64_t total = (int64_t)C * H * W;
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
    k_split_bits<<<blocks, 256, 0, s>>>(img, total, K, msb, mx);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ int64_t clamp_pix(int64_t p, int64_t hw)
{
    return p < 0 ? 0 : (p >= hw ? hw - 1 : p);
}

__global__ void __launch_bounds__(256) k_labels(const uint16_t* __restrict__ img, int C, int64_t HW,
                                                int K, const int64_t* __restrict__ idx, int64_t n,
                                                float* __restrict__ labels)
{
    const float denom = (float)((1 << K) - 1);
    const int mask = (1 << K) - 1;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * C) return;
    int64_t i = e / C;
    int c = (int)(e - i * C);
    int64_t pix = clamp_pix(idx ? idx[i] : i, HW);
    int lo = (int)img[(int64_t)c * HW + pix] & mask;  // img - (msb<<K), ref LBDRNdataset.py:96
    labels[e] = (float)lo / denom;                    // :97
}

int generic_labels(const uint16_t* img, int C, int H, int W, int K, const int64_t* idx, int64_t n,
                   float* labels, hipStream_t s)
{
    if (n == 0) return 0;
    int64_t total = n * C;
    k_labels<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(img, C, (int64_t)H * W, K, idx, n, labels);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ a2/a3: feature matrix

__global__ void __launch_bounds__(256)
    k_features(lbdrn_geom g, int F, const uint16_t* __restrict__ msb, const int64_t* __restrict__ idx,
               int64_t n, int64_t first, float* __restrict__ out)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * F) return;
    int64_t i = e / F;
    int f = (int)(e - i * F);
    const int64_t HW = (int64_t)g.H * g.W;
    int64_t pix = clamp_pix(idx ? idx[i] : first + i, HW);
    int y = (int)(pix / g.W), x = (int)(pix - (int64_t)y * g.W);
    float v;
    if (f < g.P) {
        v = g.rowtab[(int64_t)y * g.P + f];
    } else if (f < 2 * g.P) {
        v = g.coltab[(int64_t)x * g.P + (f - g.P)];
    } else {
        const int side = 2 * g.D + 1;
        int cf = f - 2 * g.P;
        int c = cf / (side * side);
        int r = cf - c * side * side;
        int dy = r / side, dx = r - dy * side;
        const uint16_t* pl = msb + (int64_t)c * HW;
        const float mx = (float)g.msb_max;
        int yy = reflect_idx(y + dy - g.D, g.H), xx = reflect_idx(x + dx - g.D, g.W);
        v = (float)pl[(int64_t)yy * g.W + xx] / mx;  // ref LBDRNdataset.py:120
        if (g.relative && g.D > 0) v = v - (float)pl[pix] / mx;  // :126-128
    }
    out[e] = v;
}

int generic_features(const lbdrn_geom& g, const uint16_t* msb, const int64_t* idx, int64_t n,
                     int64_t first, float* out, hipStream_t s)
{
    if (n == 0) return 0;
    const int F = feature_dim(g);
    int64_t total = n * F;
    LBDRN_REQUIRE((total + 255) / 256 < ((int64_t)1 << 31), "feature matrix too large for one launch");
    k_features<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(g, F, msb, idx, n, first, out);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ tiled k-ordered GEMM

// C[m][n] = store( init(m,n) + sum_k a(m,k)*b(k,n) ), k ascending, one fma per term -- on the matrix
// cores: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain seeded by its C operand, so tiling the sum
// over MFMA steps of four consecutive k keeps exactly the order of a scalar loop (and of the oracle).
// 64x64 output tile per 256-thread block: wave (wm, wn) owns a 32x32 quarter as 2x2 MFMA tiles; BK = 16
// (four MFMA steps).  Both operand tiles sit in LDS as [row][16 k] with the k positions interleaved by
// quarter (slot (k%4)*4 + k/4): lane quarter q needs k = 4s+q for the steps s = 0..3 of a k-tile, which
// are then 16 contiguous bytes -- one ds_read_b128 per operand tile per k-tile.  Row pitch 20 floats (an
// odd number of 16-byte chunks).  This is the MFMA-tiled GEMM path of every shape the fused kernels do
// not cover (bc = 256, nl > 3, C > 16 ...).
typedef float gemm_f32x4 __attribute__((ext_vector_type(4)));

template <class Prob>
__global__ void __launch_bounds__(256) k_gemm_mfma(Prob p)
{
    constexpr int BM = 64, BN = 64, BK = 16, PITCH = 20;
    __shared__ __attribute__((aligned(16))) float As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[BN * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    gemm_f32x4 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[mt][nt][r] = p.init(m0 + 32 * wm + 16 * mt + 4 * q + r, n0 + 32 * wn + 16 * nt + i);
    // optional split over k (blockIdx.z): slices of p.kchunk terms, each k-ordered
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.Kd, kbeg + p.kchunk);
    // operand tiles go global -> registers -> LDS; the registers of k-tile t+1 are loaded while the MFMAs of
    // k-tile t run (the loads are issued before the MFMAs and written to LDS after the second barrier)
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
        const int kmax = min(BK, kend - k0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + r * 256;
            int kk, mm;
            if (Prob::a_k_contig) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
            ra[r] = (kk < kmax && m0 + mm < p.M) ? p.a(m0 + mm, k0 + kk) : 0.0f;
            int kb, nn;
            if (Prob::b_k_contig) { kb = e & 15; nn = e >> 4; } else { nn = e & 63; kb = e >> 6; }
            rb[r] = (kb < kmax && n0 + nn < p.N) ? p.b(k0 + kb, n0 + nn) : 0.0f;
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + r * 256;
            int kk, mm;
            if (Prob::a_k_contig) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
            As[mm * PITCH + (kk & 3) * 4 + (kk >> 2)] = ra[r];
            int kb, nn;
            if (Prob::b_k_contig) { kb = e & 15; nn = e >> 4; } else { nn = e & 63; kb = e >> 6; }
            Bs[nn * PITCH + (kb & 3) * 4 + (kb >> 2)] = rb[r];
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        park();
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);
        float4 a4[2], b4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a4[t] = *reinterpret_cast<const float4*>(&As[(32 * wm + 16 * t + i) * PITCH + 4 * q]);
            b4[t] = *reinterpret_cast<const float4*>(&Bs[(32 * wn + 16 * t + i) * PITCH + 4 * q]);
        }
        const float av[2][4] = {{a4[0].x, a4[0].y, a4[0].z, a4[0].w}, {a4[1].x, a4[1].y, a4[1].z, a4[1].w}};
        const float bv[2][4] = {{b4[0].x, b4[0].y, b4[0].z, b4[0].w}, {b4[1].x, b4[1].y, b4[1].z, b4[1].w}};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)  // k = k0 + 4*s4 + q: ascending over steps, ascending within a step
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s4], bv[nt][s4], acc[mt][nt], 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wm + 16 * mt + 4 * q + r, n = n0 + 32 * wn + 16 * nt + i;
                if (m < p.M && n < p.N) p.store(m, n, acc[mt][nt][r], (int)blockIdx.z);
            }
}

template <class Prob>
static int launch_gemm(const Prob& p, hipStream_t s)
{
    if (p.M <= 0 || p.N <= 0) return 0;
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, (p.Kd + p.kchunk - 1) / p.kchunk);
    k_gemm_mfma<Prob><<<grid, 256, 0, s>>>(p);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

enum Act { ACT_SIN = 0, ACT_SIGMOID = 1, ACT_RELU = 2 };

// y[b][j] = act(bias[j] + sum_k x[b][k] W[j][k])   (nn.Linear + activation, LBDRNmodel.py:39-43)
template <int ACT, bool KEEP_COS>
struct LinearFwd {
    static constexpr bool a_k_contig = true, b_k_contig = true;
    int M, N, Kd, kchunk;
    const float* x;     // [M][Kd]
    const float* W;     // [N][Kd]
    const float* bias;  // [N]
    float* out;         // [M][N]
    float* cosout;      // [M][N] cos(30 z), training only
    __device__ float a(int m, int k) const { return x[(int64_t)m * Kd + k]; }
    __device__ float b(int k, int n) const { return W[(int64_t)n * Kd + k]; }
    __device__ float init(int m, int n) const { return n < N ? bias[n] : 0.0f; }
    __device__ void store(int m, int n, float z, int) const
    {
        if (ACT == ACT_SIN) {
            if (KEEP_COS) {
                float sn, cs;
                canon_sincos(30.0f * z, sn, cs);
                out[(int64_t)m * N + n] = sn;
                cosout[(int64_t)m * N + n] = cs;
            } else {
                out[(int64_t)m * N + n] = siren_act(z);
            }
        } else if (ACT == ACT_RELU) {   // nn.ReLU: z where z > 0, else 0; its derivative (1 / 0) in the slot of the cosine
            out[(int64_t)m * N + n] = z > 0.0f ? z : 0.0f;
            if (KEEP_COS) cosout[(int64_t)m * N + n] = z > 0.0f ? 1.0f : 0.0f;
        } else {
            out[(int64_t)m * N + n] = canon_sigmoid(z);
        }
    }
};

// one hidden layer of the net's activation
template <bool KEEP_COS>
static int launch_hidden(const lbdrn_net& net, int B, int nin, const float* in, const float* W, float* out, float* dact,
                         hipStream_t s)
{
    if (net.act == LBDRN_ACT_RELU) {
        LinearFwd<ACT_RELU, KEEP_COS> g{B, net.bc, nin, nin, in, W, W + (int64_t)net.bc * nin, out, dact};
        return launch_gemm(g, s);
    }
    LinearFwd<ACT_SIN, KEEP_COS> g{B, net.bc, nin, nin, in, W, W + (int64_t)net.bc * nin, out, dact};
    return launch_gemm(g, s);
}

// dzprev[b][k] = ((sum_j dz[b][j] W[j][k]) * cos(30 zprev[b][k])) * 30   (autograd of Linear + Sine); with nn.ReLU the sum
// where zprev > 0 and zero elsewhere (threshold_backward)
struct BackDx {
    static constexpr bool a_k_contig = true, b_k_contig = false;
    int M, N, Kd, kchunk;
    const float* dz;    // [M][Kd]
    const float* W;     // [Kd][N]
    const float* cosv;  // [M][N]
    float* out;         // [M][N]
    int relu;
    __device__ float a(int m, int k) const { return dz[(int64_t)m * Kd + k]; }
    __device__ float b(int k, int n) const { return W[(int64_t)k * N + n]; }
    __device__ float init(int, int) const { return 0.0f; }
    __device__ void store(int m, int n, float v, int) const
    {
        const float d = cosv[(int64_t)m * N + n];
        out[(int64_t)m * N + n] = relu ? (d != 0.0f ? v : 0.0f) : (v * d) * 30.0f;
    }
};

// dW[j][k] = sum_b dz[b][j] in[b][k] and, as column N-1 (an input of ones), db[j] = sum_b dz[b][j]
// (autograd of nn.Linear).  The batch sum is cut into slices of kchunk rows (blockIdx.z); each slice
// is a b-ordered fmaf chain into part[z][M][N]; k_reduce_slices then adds the slices in z order.
struct GradW {
    static constexpr bool a_k_contig = false, b_k_contig = false;
    int M, N, Kd, kchunk;  // N = inputs + 1
    const float* dz;  // [Kd][M]
    const float* in;  // [Kd][N-1]
    float* part;      // [slices][M][N]
    __device__ float a(int m, int k) const { return dz[(int64_t)k * M + m]; }
    __device__ float b(int k, int n) const { return n < N - 1 ? in[(int64_t)k * (N - 1) + n] : 1.0f; }
    __device__ float init(int, int) const { return 0.0f; }
    __device__ void store(int m, int n, float v, int z) const
    {
        part[((int64_t)z * M + m) * N + n] = v;
    }
};

// gW[m][n] = sum_z part[z][m][n] (n < N-1), gb[m] = sum_z part[z][m][N-1]; z ascending
__global__ void __launch_bounds__(256)
    k_reduce_slices(const float* __restrict__ part, int slices, int M, int N, float* __restrict__ gW,
                    float* __restrict__ gb)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * N) return;
    // (sixteen loads in flight, added in z order: one load per add made the launch a chain of memory latencies -- 9.4 us for the
    //  32 slices of an 8192-row minibatch, a fifth of the generic step)
    float acc = 0.0f;
    const int64_t MN = (int64_t)M * N;
    int z = 0;
    for (; z + 16 <= slices; z += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = part[(int64_t)(z + u) * MN + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += t[u];
    }
    for (; z < slices; ++z) acc += part[(int64_t)z * MN + e];
    int m = e / N, n = e - m * N;
    if (n < N - 1) gW[(int64_t)m * (N - 1) + n] = acc;
    else gb[m] = acc;
}

constexpr int GRAD_SLICE = 256;  // batch rows per slice

// ------------------------------------------------------------------ a5: forward

size_t generic_forward_workspace(const lbdrn_net& net, int64_t B)
{
    return 2 * align_up((size_t)B * net.bc * sizeof(float), 256);
}

int generic_forward(const lbdrn_net& net, const float* params, const float* x, int64_t B, float* y,
                    void* ws, size_t ws_bytes, hipStream_t s)
{
    if (B == 0) return 0;
    LBDRN_REQUIRE(B < ((int64_t)1 << 31) / std::max(net.bc, net.F), "batch too large for one call");
    if (ws_bytes < generic_forward_workspace(net, B) || !ws) {
        set_error("forward workspace too small: %zu < %zu", ws_bytes, generic_forward_workspace(net, B));
        return LBDRN_E_WORKSPACE;
    }
    float* h[2] = {(float*)ws, (float*)((char*)ws + align_up((size_t)B * net.bc * sizeof(float), 256))};
    const float* in = x;
    int nin = net.F;
    const float* p = params;
    for (int l = 0; l < net.nl; ++l) {
        if (int rc = launch_hidden<false>(net, (int)B, nin, in, p, h[l & 1], nullptr, s)) return rc;
        p += (int64_t)net.bc * nin + net.bc;
        in = h[l & 1];
        nin = net.bc;
    }
    LinearFwd<ACT_SIGMOID, false> g{(int)B, net.C, nin, nin, in, p, p + (int64_t)net.C * nin, y, nullptr};
    return launch_gemm(g, s);
}

// ------------------------------------------------------------------ a11: reconstruct, a9: SSE

__global__ void __launch_bounds__(256)
    k_reconstruct(const float* __restrict__ y, const uint16_t* __restrict__ msb, int C, int64_t HW,
                  int64_t first, int64_t n, int K, uint16_t* __restrict__ out)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * C) return;
    // thread e -> (c, i) with i fastest so that the planar store is coalesced
    int c = (int)(e / n);
    int64_t i = e - (int64_t)c * n;
    const float scale = (float)((1 << K) - 1);
    float r = __builtin_rintf(y[i * C + c] * scale);  // torch.round, ref decode.py:131
    int64_t pix = first + i;
    int base = (int)msb[(int64_t)c * HW + pix] << K;  // ref decode.py:134
    out[(int64_t)c * HW + pix] = (uint16_t)(base + (int)r);
}

constexpr int SSE_BLOCKS = 256;

// partial[b] = sum over this block's strided elements of (y - label)^2 in float64
__global__ void __launch_bounds__(256)
    k_sse_partial(const float* __restrict__ y, const uint16_t* __restrict__ img, int C, int64_t HW,
                  int64_t first, int64_t n, int K, double* __restrict__ partial)
{
    const float denom = (float)((1 << K) - 1);
    const int mask = (1 << K) - 1;
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * C;
         e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = e / C;
        int c = (int)(e - i * C);
        float lab = (float)((int)img[(int64_t)c * HW + first + i] & mask) / denom;
        float d = y[e] - lab;
        acc += (double)(d * d);
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// *dst (+)= sum of partial[0..n) in index order (single thread: the order is the contract)
__global__ void k_sum_partials(const double* __restrict__ partial, int n, int accumulate, double* dst)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = accumulate ? *dst : 0.0;
    for (int i = 0; i < n; ++i) s += partial[i];
    *dst = s;
}

constexpr int64_t APPLY_CHUNK = 65536;

size_t generic_apply_workspace(const lbdrn_geom& g, const lbdrn_net& net)
{
    int64_t chunk = std::min<int64_t>(APPLY_CHUNK, (int64_t)g.H * g.W);
    return align_up((size_t)chunk * net.F * sizeof(float), 256) +
           align_up((size_t)chunk * net.C * sizeof(float), 256) +
           generic_forward_workspace(net, chunk) + align_up(SSE_BLOCKS * sizeof(double), 256);
}

struct ApplyWs {
    float* x;
    float* y;
    void* fwd;
    size_t fwd_bytes;
    double* partial;
    int64_t chunk;
};

static int carve_apply(const lbdrn_geom& g, const lbdrn_net& net, void* ws, size_t ws_bytes, ApplyWs* a)
{
    if (!ws || ws_bytes < generic_apply_workspace(g, net)) {
        set_error("apply workspace too small: %zu < %zu", ws_bytes, generic_apply_workspace(g, net));
        return LBDRN_E_WORKSPACE;
    }
    a->chunk = std::min<int64_t>(APPLY_CHUNK, (int64_t)g.H * g.W);
    char* p = (char*)ws;
    a->x = (float*)p;
    p += align_up((size_t)a->chunk * net.F * sizeof(float), 256);
    a->y = (float*)p;
    p += align_up((size_t)a->chunk * net.C * sizeof(float), 256);
    a->fwd = p;
    a->fwd_bytes = generic_forward_workspace(net, a->chunk);
    p += a->fwd_bytes;
    a->partial = (double*)p;
    return 0;
}

int generic_decode(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* msb,
                   const float* params, uint16_t* out, float* y_out, void* ws, size_t ws_bytes,
                   hipStream_t s)
{
    ApplyWs a;
    if (int rc = carve_apply(g, net, ws, ws_bytes, &a)) return rc;
    const int64_t HW = (int64_t)g.H * g.W;
    for (int64_t first = 0; first < HW; first += a.chunk) {
        int64_t n = std::min(a.chunk, HW - first);
        float* y = y_out ? y_out + first * net.C : a.y;
        if (int rc = generic_features(g, msb, nullptr, n, first, a.x, s)) return rc;
        if (int rc = generic_forward(net, params, a.x, n, y, a.fwd, a.fwd_bytes, s)) return rc;
        int64_t total = n * net.C;
        k_reconstruct<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(y, msb, net.C, HW, first, n, g.K, out);
        LBDRN_LAUNCH_CHECK();
    }
    return 0;
}

int generic_eval_sse(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                     const uint16_t* msb, const float* params, double* sse, void* ws,
                     size_t ws_bytes, hipStream_t s)
{
    ApplyWs a;
    if (int rc = carve_apply(g, net, ws, ws_bytes, &a)) return rc;
    const int64_t HW = (int64_t)g.H * g.W;
    for (int64_t first = 0; first < HW; first += a.chunk) {
        int64_t n = std::min(a.chunk, HW - first);
        if (int rc = generic_features(g, msb, nullptr, n, first, a.x, s)) return rc;
        if (int rc = generic_forward(net, params, a.x, n, a.y, a.fwd, a.fwd_bytes, s)) return rc;
        k_sse_partial<<<SSE_BLOCKS, 256, 0, s>>>(a.y, img, net.C, HW, first, n, g.K, a.partial);
        LBDRN_LAUNCH_CHECK();
        k_sum_partials<<<1, 64, 0, s>>>(a.partial, SSE_BLOCKS, first != 0, sse);
        LBDRN_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------ a7/a8: loss, backward, Adam

constexpr int LOSS_BLOCKS = 64;

// dz[b][c] = (2 (y-t)/(B*C)) * (y (1-y)); partial[b] = block sum of (y-t)^2 (float64)
__global__ void __launch_bounds__(256)
    k_loss_grad(const float* __restrict__ y, const float* __restrict__ t, int64_t total, float inv,
                float* __restrict__ dz, double* __restrict__ partial)
{
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        float yy = y[e];
        float d = yy - t[e];
        acc += (double)(d * d);                  // F.mse_loss numerator, ref LBDRNloss.py:9
        float dy = (2.0f * d) * inv;             // d mean((y-t)^2) / dy
        dz[e] = dy * (yy * (1.0f - yy));         // sigmoid backward
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void k_finish_loss(const double* __restrict__ partial, int n, double count, float* loss)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[i];
    *loss = (float)(s / count);
}

// torch.optim.Adam single-tensor update (torch/optim/adam.py: lerp_, mul_/addcmul_, addcdiv_),
// defaults beta=(0.9,0.999), eps=1e-8, no weight decay (ref encode.py:84).
__global__ void __launch_bounds__(256)
    k_adam(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
           const float* __restrict__ g, int64_t n, float step_size, float bc2_sqrt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float w1 = (float)(1.0 - 0.9), b2 = 0.999f, w2 = (float)(1.0 - 0.999), eps = 1e-8f;
    float gi = g[i];
    float mi = m[i] + w1 * (gi - m[i]);
    float vi = v[i] * b2 + w2 * (gi * gi);
    float denom = __builtin_sqrtf(vi) / bc2_sqrt + eps;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] + (-step_size) * (mi / denom);
}

int launch_adam(float* p, float* m, float* v, const float* g, int64_t n, int64_t adam_step, double lr,
                hipStream_t s)
{
    const double bc1 = 1.0 - std::pow(0.9, (double)adam_step);
    const double bc2 = 1.0 - std::pow(0.999, (double)adam_step);
    const float step_size = (float)(lr / bc1);
    const float bc2s = (float)std::sqrt(bc2);
    k_adam<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p, m, v, g, n, step_size, bc2s);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

// workspace layout of one generic training step
struct TrainWs {
    float* h;      // [nl][B][bc] activations
    float* cs;     // [nl][B][bc] cos(30 z)
    float* y;      // [B][C]
    float* dzl;    // [B][C]
    float* dz[2];  // [B][bc] ping-pong
    float* grads;  // [NP]
    float* slices; // [ceil(B/GRAD_SLICE)][max layer rows][max layer cols + 1]
    double* partial;
    float* x;      // [B][F]  (epoch driver)
    float* t;      // [B][C]
};

static size_t carve_train(const lbdrn_net& net, int B, void* ws, TrainWs* w)
{
    char* p = (char*)ws;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += align_up(bytes, 256);
        return r;
    };
    size_t act = (size_t)net.nl * B * net.bc * sizeof(float);
    char* h = take(act);
    char* cs = take(act);
    char* y = take((size_t)B * net.C * sizeof(float));
    char* dzl = take((size_t)B * net.C * sizeof(float));
    char* dz0 = take((size_t)B * net.bc * sizeof(float));
    char* dz1 = take((size_t)B * net.bc * sizeof(float));
    char* gr = take((size_t)param_count(net) * sizeof(float));
    const size_t nsl = (size_t)(B + GRAD_SLICE - 1) / GRAD_SLICE;
    const size_t rows = (size_t)std::max(net.bc, net.C), cols = (size_t)std::max(net.bc, net.F) + 1;
    char* sl = take(nsl * rows * cols * sizeof(float));
    char* pa = take(LOSS_BLOCKS * sizeof(double));
    char* x = take((size_t)B * net.F * sizeof(float));
    char* t = take((size_t)B * net.C * sizeof(float));
    if (w) {
        w->h = (float*)h; w->cs = (float*)cs; w->y = (float*)y; w->dzl = (float*)dzl;
        w->dz[0] = (float*)dz0; w->dz[1] = (float*)dz1; w->grads = (float*)gr; w->slices = (float*)sl;
        w->partial = (double*)pa; w->x = (float*)x; w->t = (float*)t;
    }
    return (size_t)(p - (char*)ws);
}

size_t generic_train_workspace(const lbdrn_net& net, int B) { return carve_train(net, B, nullptr, nullptr); }

// gradients of one Linear: gW[rows][cols] and, directly behind it, gb[rows]
static int grad_layer(const TrainWs& w, int rows, int cols, int B, const float* dz, const float* in,
                      float* gW, hipStream_t s)
{
    GradW g{rows, cols + 1, B, GRAD_SLICE, dz, in, w.slices};
    if (int rc = launch_gemm(g, s)) return rc;
    const int slices = (B + GRAD_SLICE - 1) / GRAD_SLICE;
    const int total = rows * (cols + 1);
    k_reduce_slices<<<(total + 255) / 256, 256, 0, s>>>(w.slices, slices, rows, cols + 1, gW,
                                                        gW + (int64_t)rows * cols);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

int generic_train_step(const lbdrn_net& net, const float* x, const float* t, int B, float* params,
                       float* m, float* v, int64_t adam_step, double lr, int apply_adam,
                       float* loss, float* grads_out, void* ws, size_t ws_bytes, hipStream_t s)
{
    LBDRN_REQUIRE(B >= 1, "empty minibatch");
    if (!ws || ws_bytes < generic_train_workspace(net, B)) {
        set_error("train workspace too small: %zu < %zu", ws_bytes, generic_train_workspace(net, B));
        return LBDRN_E_WORKSPACE;
    }
    TrainWs w;
    carve_train(net, B, ws, &w);
    const int64_t NP = param_count(net);
    const size_t act = (size_t)B * net.bc;
    // forward, keeping activations and cos(30 z)
    const float* in = x;
    int nin = net.F;
    for (int l = 0; l < net.nl; ++l) {
        const float* W = params + layer_offset(net, l);
        if (int rc = launch_hidden<true>(net, B, nin, in, W, w.h + l * act, w.cs + l * act, s)) return rc;
        in = w.h + l * act;
        nin = net.bc;
    }
    const float* Wl = params + layer_offset(net, net.nl);
    {
        LinearFwd<ACT_SIGMOID, false> g{B, net.C, net.bc, net.bc, in, Wl, Wl + (int64_t)net.C * net.bc, w.y, nullptr};
        if (int rc = launch_gemm(g, s)) return rc;
    }
    // loss + d/dz of the last layer
    const int64_t total = (int64_t)B * net.C;
    const float inv = 1.0f / ((float)B * (float)net.C);
    k_loss_grad<<<LOSS_BLOCKS, 256, 0, s>>>(w.y, t, total, inv, w.dzl, w.partial);
    LBDRN_LAUNCH_CHECK();
    if (loss) {
        k_finish_loss<<<1, 64, 0, s>>>(w.partial, LOSS_BLOCKS, (double)total, loss);
        LBDRN_LAUNCH_CHECK();
    }
    // last layer gradients
    float* gl = w.grads + layer_offset(net, net.nl);
    if (int rc = grad_layer(w, net.C, net.bc, B, w.dzl, w.h + (net.nl - 1) * act, gl, s)) return rc;
    // back through the hidden layers
    const float* dz_up = w.dzl;
    int n_up = net.C;
    const float* W_up = Wl;
    for (int l = net.nl - 1; l >= 0; --l) {
        float* dz = w.dz[l & 1];
        BackDx bd{B, net.bc, n_up, n_up, dz_up, W_up, w.cs + l * act, dz, net.act == LBDRN_ACT_RELU};
        if (int rc = launch_gemm(bd, s)) return rc;
        const int lin = l ? net.bc : net.F;
        const float* lin_act = l ? w.h + (l - 1) * act : x;
        float* gW = w.grads + layer_offset(net, l);
        if (int rc = grad_layer(w, net.bc, lin, B, dz, lin_act, gW, s)) return rc;
        dz_up = dz;
        n_up = net.bc;
        W_up = params + layer_offset(net, l);
    }
    if (grads_out) LBDRN_HIP_TRY(hipMemcpyAsync(grads_out, w.grads, NP * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (apply_adam) return launch_adam(params, m, v, w.grads, NP, adam_step, lr, s);
    return 0;
}

int generic_train_epoch(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                        const uint16_t* msb, const int64_t* perm, int64_t n, int bs, float* params,
                        float* m, float* v, int64_t step0, double lr, float* losses, void* ws,
                        size_t ws_bytes, hipStream_t s)
{
    if (!ws || ws_bytes < generic_train_workspace(net, bs)) {
        set_error("train workspace too small: %zu < %zu", ws_bytes, generic_train_workspace(net, bs));
        return LBDRN_E_WORKSPACE;
    }
    TrainWs w;
    carve_train(net, bs, ws, &w);
    int64_t step = step0;
    int si = 0;
    for (int64_t first = 0; first < n; first += bs, ++si) {
        int B = (int)std::min<int64_t>(bs, n - first);
        if (int rc = generic_features(g, msb, perm + first, B, 0, w.x, s)) return rc;
        if (int rc = generic_labels(img, g.C, g.H, g.W, g.K, perm + first, B, w.t, s)) return rc;
        ++step;
        if (int rc = generic_train_step(net, w.x, w.t, B, params, m, v, step, lr, 1,
                                        losses ? losses + si : nullptr, nullptr, ws, ws_bytes, s))
            return rc;
    }
    return 0;
}

}  // namespace lbdrn
