// "LBB2": lossless payload of the MSB planes, coded and decoded on the GPU.
//
// Where it stands: the reference stores the MSB raster with an external lossless codec (JPEG 2000 through the
// gdal_translate CLI, ref encode.py:137, decode.py:69-73).  Neither GDAL nor OpenJPEG is in this image, the
// stand-in used so far (plane predictor + LZMA on the host) needs 50 s for one 8 x 2048^2 tile -- 300 times the
// whole fit -- and the planes already sit in HBM.  This codec is byte/integer work shaped for a 64-wide
// wavefront; the format is this package's own (tag LBB2 in container.py), so parity with the reference's
// JPEG 2000 bytes is unpinned by construction -- the contract is the lossless round trip, and byte identity
// with the sequential restatement in oracle/plane_codec.c (tests/test_gpu_plane_codec.py).
//
// Format (full text in oracle/plane_codec.c): a band is cut into strips of 64 columns over the full height;
// one wave codes a strip in passes of 64 rows, lane i owning row 64p+i.  Per pass one of four predictors
// (MED / mean / plane / their mix, the one with the smallest residual sum), residuals folded to unsigned,
// adaptive Golomb-Rice per lane with zero-group flags for flat areas.  Every lane builds a private bit
// stream; the strip's 32-bit words are interleaved in the order the decoding wave asks for them while it
// walks the anti-diagonals of a pass (lane i decodes column d-i at step d: exactly the order in which the
// predictor's left / up / up-left neighbours become available), so decoding is one coalesced word stream per
// wave, the per-step hand-out is a ballot + mbcnt, and no lane ever waits for another lane's bits.
//
// Encoder, per strip (one 64-thread workgroup): phase A walks the passes -- stage the 65 x 64 pixel window in
// LDS, residual sums of the four predictors (wave reduction), code the lane's row into its private stream
// (global scratch) and record each symbol's length; phase B replays the decoder's requests from the recorded
// lengths and emits the words in that order.  A scan of the strips' word counts and a compaction kernel then
// pack the strips back to back.  Decoder: scan of the counts, then one wave per strip with the strip's next
// words and the pixel window in LDS.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace lbdrn {

constexpr int PL_T = 64;      // strip width = lanes = rows per pass
constexpr int PL_NMODES = 4;
constexpr int PL_QESC = 15;   // unary length that announces a raw 16-bit value

struct PlaneGeom {
    int C, H, W, TX, NP;
    int64_t nstrips;
    size_t pw;    // private words per lane (worst case + tail)
    size_t capw;  // words per strip (worst case)
};

static PlaneGeom plane_geom(int C, int H, int W)
{
    PlaneGeom g;
    g.C = C; g.H = H; g.W = W;
    g.TX = (W + PL_T - 1) / PL_T;
    g.NP = (H + PL_T - 1) / PL_T;
    g.nstrips = (int64_t)C * g.TX;
    g.pw = (size_t)g.NP * PL_T + 1;
    g.capw = 1 + (size_t)g.NP + PL_T * g.pw;
    return g;
}

struct PlaneWs {
    uint32_t* priv;     // [nstrips][64][pw]
    uint8_t* lens;      // [nstrips][NP*64][64]
    int* modes;         // [nstrips][NP]
    uint32_t* words;    // [nstrips][capw]   (encoder: strips before compaction)
    uint64_t* offsets;  // [nstrips + 1]
    void* scan_tmp;
    size_t scan_bytes, total;
};

static int carve_plane(const PlaneGeom& g, void* ws, PlaneWs* w)
{
    size_t scan_bytes = 0;
    uint32_t* in = nullptr;
    uint64_t* out = nullptr;
    if (rocprim::exclusive_scan(nullptr, scan_bytes, in, out, (uint64_t)0, (size_t)std::max<int64_t>(g.nstrips, 1),
                                rocprim::plus<uint64_t>()) != hipSuccess)
        return LBDRN_E_DEVICE;
    char* p = (char*)ws;
    w->priv = (uint32_t*)p; p += align_up((size_t)g.nstrips * PL_T * g.pw * 4, 256);
    w->lens = (uint8_t*)p; p += align_up((size_t)g.nstrips * g.NP * PL_T * PL_T, 256);
    w->modes = (int*)p; p += align_up((size_t)g.nstrips * g.NP * 4, 256);
    w->words = (uint32_t*)p; p += align_up((size_t)g.nstrips * g.capw * 4, 256);
    w->offsets = (uint64_t*)p; p += align_up((size_t)(g.nstrips + 1) * 8, 256);
    w->scan_tmp = p; p += align_up(scan_bytes, 256);
    w->scan_bytes = scan_bytes;
    w->total = (size_t)(p - (char*)ws);
    return 0;
}

// ------------------------------------------------------------------ the symbol coder (both directions)

__device__ __forceinline__ int pl_inner(int a, int b, int c, int mode)
{
    if (mode == 0) {
        const int mn = min(a, b), mx = max(a, b);
        return c >= mx ? mn : (c <= mn ? mx : a + b - c);
    }
    if (mode == 1) return (a + b) >> 1;
    if (mode == 2) return a + b - c;
    return ((3 * (a + b) - 2 * c + (1 << 20)) >> 2) - (1 << 18);
}
// r: row in the image, j: column in the strip; a left, b up, c up-left
__device__ __forceinline__ int pl_predict(int r, int j, int a, int b, int c, int mode)
{
    if (r == 0) return j == 0 ? 0 : a;
    if (j == 0) return b;
    return pl_inner(a, b, c, mode);
}
__device__ __forceinline__ uint32_t pl_fold(int x, int pred)
{
    const int e = (int)(short)(unsigned short)(x - pred);
    return e >= 0 ? 2u * (uint32_t)e : (uint32_t)(-2 * e - 1);
}

struct LaneCoder {
    uint32_t A, N;
    int left;
    bool zero;
    __device__ void init(int k0, int z0)
    {
        N = 4;
        A = z0 == 2 ? 0u : (z0 == 1 ? 1u : 6u << k0);
        left = 0;
        zero = false;
    }
    __device__ int k() const
    {
        int kk = 0;
        while (kk < 15 && (N << (kk + 1)) < A) ++kk;
        return kk;
    }
    __device__ int group() const { return 16 * A < N ? 16 : (2 * A < N ? 4 : 1); }
    __device__ void update(uint32_t v)
    {
        left -= 1;
        A += v;
        N += 1;
        if (N == 32) { A = (A + 1) >> 1; N = 16; }
    }
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// ------------------------------------------------------------------ encoder

__global__ void __launch_bounds__(PL_T) k_plane_encode(const uint16_t* __restrict__ planes, PlaneGeom g, PlaneWs w,
                                                        uint32_t* __restrict__ counts)
{
    __shared__ uint16_t px[PL_T + 1][PL_T + 2];   // row 0 = the row above the pass
    __shared__ uint16_t vrow[PL_T][PL_T + 2];     // folded residuals of the lane's row
    __shared__ __attribute__((aligned(16))) uint8_t lensw[PL_T][PL_T];
    __shared__ uint32_t pwin[PL_T][PL_T + 1];
    const int lane = threadIdx.x;
    const int64_t strip = blockIdx.x;
    const int c = (int)(strip / g.TX), tx = (int)(strip % g.TX);
    const int x0 = tx * PL_T, tw = min(PL_T, g.W - x0);
    const uint16_t* band = planes + (size_t)c * g.H * g.W + x0;
    uint32_t* priv = w.priv + ((size_t)strip * PL_T + lane) * g.pw;
    uint8_t* glens = w.lens + (size_t)strip * g.NP * PL_T * PL_T;
    int* gmodes = w.modes + (size_t)strip * g.NP;
    uint32_t* wout = w.words + (size_t)strip * g.capw;

    // ---- phase A: private streams
    LaneCoder st;
    st.init(0, 0);
    int k0 = 0, z0 = 0;
    unsigned long long acc = 0;
    int nb = 0;
    size_t pwords = 0;
    for (int p = 0; p < g.NP; ++p) {
        const int rows = min(PL_T, g.H - p * PL_T);
        __syncthreads();
        for (int rr = (p == 0 ? 1 : 0); rr <= rows; ++rr)
            px[rr][lane] = lane < tw ? band[(size_t)(p * PL_T - 1 + rr) * g.W + lane] : (uint16_t)0;
        __syncthreads();
        const int r = p * PL_T + lane;
        uint32_t sum[PL_NMODES] = {0, 0, 0, 0};
        if (lane < rows)
            for (int j = 0; j < tw; ++j) {
                const int a = j ? px[lane + 1][j - 1] : 0, b = r ? px[lane][j] : 0, cc = (j && r) ? px[lane][j - 1] : 0;
                const int x = px[lane + 1][j];
#pragma unroll
                for (int m = 0; m < PL_NMODES; ++m) sum[m] += pl_fold(x, pl_predict(r, j, a, b, cc, m));
            }
        int mode = 0;
        uint32_t best = wave_sum(sum[0]);
#pragma unroll
        for (int m = 1; m < PL_NMODES; ++m) {
            const uint32_t s = wave_sum(sum[m]);
            if (s < best) { best = s; mode = m; }
        }
        if (lane == 0) gmodes[p] = mode;
        if (p == 0) {
            const unsigned long long count = (unsigned long long)rows * tw;
            while (k0 < 15 && (count << (k0 + 1)) < best) ++k0;
            z0 = 16ull * best < count ? 2 : (2ull * best < count ? 1 : 0);
            st.init(k0, z0);
        }
        if (lane < rows) {
            for (int j = 0; j < tw; ++j) {
                const int a = j ? px[lane + 1][j - 1] : 0, b = r ? px[lane][j] : 0, cc = (j && r) ? px[lane][j - 1] : 0;
                vrow[lane][j] = (uint16_t)pl_fold(px[lane + 1][j], pl_predict(r, j, a, b, cc, mode));
            }
            st.left = 0;  // groups do not cross rows
            for (int j = 0; j < tw; ++j) {
                const uint32_t v = vrow[lane][j];
                uint32_t code = 0;
                int len = 0;
                if (st.left == 0) {
                    int gsz = st.group();
                    st.zero = false;
                    if (gsz > 1) {
                        gsz = min(gsz, tw - j);
                        bool all0 = true;
                        for (int u = 0; u < gsz; ++u) all0 = all0 && vrow[lane][j + u] == 0;
                        code = all0 ? 0u : 1u;
                        len = 1;
                        st.zero = all0;
                    }
                    st.left = gsz;
                }
                if (!st.zero) {
                    const int kk = st.k();
                    const uint32_t q = v >> kk;
                    if (q < (uint32_t)PL_QESC) {
                        code = (code << (q + 1 + kk)) | (((1u << q) - 1u) << (kk + 1)) | (v & ((1u << kk) - 1u));
                        len += (int)q + 1 + kk;
                    } else {
                        code = (code << 31) | (0x7FFFu << 16) | v;
                        len += 31;
                    }
                }
                lensw[lane][j] = (uint8_t)len;
                if (len) {
                    acc = (acc << len) | code;
                    nb += len;
                    if (nb >= 32) {
                        priv[pwords++] = (uint32_t)(acc >> (nb - 32));
                        nb -= 32;
                    }
                }
                st.update(v);
            }
        }
        __syncthreads();
        {   // the pass's symbol lengths, 4 KB, coalesced
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&lensw[0][0]);
            uint32_t* dst = reinterpret_cast<uint32_t*>(glens + (size_t)p * PL_T * PL_T);
            for (int idx = lane; idx < PL_T * PL_T / 4; idx += PL_T) dst[idx] = src[idx];
        }
    }
    if (lane < g.H) priv[pwords++] = nb ? (uint32_t)(acc << (32 - nb)) : 0u;  // tail, zero padded
    __threadfence_block();
    __syncthreads();

    // ---- phase B: the decoder's request order
    uint32_t nw = 0;
    if (lane == 0) wout[0] = (uint32_t)k0 | ((uint32_t)z0 << 12);
    nw = 1;
    int fill = 0;
    size_t taken = 0;
    for (int p = 0; p < g.NP; ++p) {
        const int rows = min(PL_T, g.H - p * PL_T);
        if (lane == 0) wout[nw] = (uint32_t)gmodes[p];
        nw += 1;
        __syncthreads();
        {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(glens + (size_t)p * PL_T * PL_T);
            uint32_t* dst = reinterpret_cast<uint32_t*>(&lensw[0][0]);
            for (int idx = lane; idx < PL_T * PL_T / 4; idx += PL_T) dst[idx] = src[idx];
        }
        const size_t taken0 = taken;
        for (int u = 0; u <= PL_T; ++u) pwin[lane][u] = taken0 + u < pwords ? priv[taken0 + u] : 0u;
        __syncthreads();
        for (int d = 0; d < rows + tw - 1; ++d) {
            const int j = d - lane;
            const bool active = lane < rows && j >= 0 && j < tw;
            const bool need = active && fill < 32;
            const unsigned long long mask = __ballot(need);
            if (need) {
                wout[nw + lanes_below(mask)] = pwin[lane][taken - taken0];
                taken += 1;
                fill += 32;
            }
            nw += (uint32_t)__popcll(mask);
            if (active) fill -= lensw[lane][j];
        }
    }
    if (lane == 0) counts[strip] = nw;
}

__global__ void __launch_bounds__(256) k_plane_compact(const uint32_t* __restrict__ strips, size_t capw,
                                                        const uint32_t* __restrict__ counts,
                                                        const uint64_t* __restrict__ offsets, int64_t nstrips,
                                                        uint32_t* __restrict__ out, uint64_t* __restrict__ body_bytes)
{
    const int64_t s = blockIdx.x;
    const uint32_t n = counts[s];
    const uint32_t* src = strips + (size_t)s * capw;
    uint32_t* dst = out + offsets[s];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    if (s == nstrips - 1 && threadIdx.x == 0) *body_bytes = ((uint64_t)nstrips + offsets[s] + n) * 4;
}

// ------------------------------------------------------------------ decoder

constexpr int PL_WWIN = PL_T * PL_T + 1;  // most words one pass can take: its mode word is read before staging

__global__ void __launch_bounds__(PL_T) k_plane_decode(const uint32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ words,
                                                        const uint64_t* __restrict__ offsets, PlaneGeom g,
                                                        uint64_t total_words, uint16_t* __restrict__ planes,
                                                        int* __restrict__ status)
{
    __shared__ uint16_t px[PL_T + 1][PL_T + 2];
    __shared__ uint32_t wwin[PL_WWIN];
    const int lane = threadIdx.x;
    const int64_t strip = blockIdx.x;
    const int c = (int)(strip / g.TX), tx = (int)(strip % g.TX);
    const int x0 = tx * PL_T, tw = min(PL_T, g.W - x0);
    uint16_t* band = planes + (size_t)c * g.H * g.W + x0;
    const uint64_t off = offsets[strip];
    uint32_t nws = counts[strip];
    bool bad = false;
    if (off + nws > total_words) { nws = 0; bad = true; }   // counts that overrun the payload
    const uint32_t* ws = words + off;
    const uint32_t w0 = nws ? ws[0] : 0u;
    int k0 = (int)(w0 & 0xFF), z0 = (int)(w0 >> 12);
    if (nws < 1 || k0 > 15 || z0 > 2 || (w0 & 0xF00)) { bad = true; k0 = 0; z0 = 0; }
    LaneCoder st;
    st.init(k0, z0);
    unsigned long long buf = 0;
    int nb = 0;
    uint32_t cur = 1;
    for (int p = 0; p < g.NP; ++p) {
        const int rows = min(PL_T, g.H - p * PL_T);
        int mode = 0;
        if (cur < nws) mode = (int)ws[cur]; else bad = true;
        if (mode < 0 || mode >= PL_NMODES) { bad = true; mode = 0; }
        cur += 1;
        const uint32_t cur0 = cur;
        __syncthreads();
        for (int idx = lane; idx < PL_WWIN; idx += PL_T) wwin[idx] = (uint64_t)cur0 + idx < nws ? ws[cur0 + idx] : 0u;
        __syncthreads();
        st.left = 0;
        const int r = p * PL_T + lane;
        for (int d = 0; d < rows + tw - 1; ++d) {
            const int j = d - lane;
            const bool active = lane < rows && j >= 0 && j < tw;
            const bool need = active && nb < 32;
            const unsigned long long mask = __ballot(need);
            if (need) {
                const uint32_t idx = cur + (uint32_t)lanes_below(mask);
                if (idx >= nws) bad = true;
                buf |= (unsigned long long)wwin[min(idx - cur0, (uint32_t)PL_WWIN - 1)] << (32 - nb);
                nb += 32;
            }
            cur += (uint32_t)__popcll(mask);
            if (active) {
                uint32_t top = (uint32_t)(buf >> 32), v = 0;
                int len = 0;
                if (st.left == 0) {
                    int gsz = st.group();
                    st.zero = false;
                    if (gsz > 1) {
                        gsz = min(gsz, tw - j);
                        st.zero = !(top >> 31);
                        top <<= 1;
                        len = 1;
                    }
                    st.left = gsz;
                }
                if (!st.zero) {
                    const int kk = st.k();
                    const int q = min(__clz((int)~top), PL_QESC);   // leading ones, at most the escape length
                    if (q >= PL_QESC) { v = (top >> 1) & 0xFFFFu; len += 31; }
                    else { v = ((uint32_t)q << kk) | ((top >> (31 - q - kk)) & ((1u << kk) - 1u)); len += q + 1 + kk; }
                }
                buf <<= len;
                nb -= len;
                st.update(v);
                const int e = (v & 1u) ? -(int)((v + 1u) >> 1) : (int)(v >> 1);
                const int a = j ? px[lane + 1][j - 1] : 0, b = r ? px[lane][j] : 0, cc = (j && r) ? px[lane][j - 1] : 0;
                px[lane + 1][j] = (uint16_t)(pl_predict(r, j, a, b, cc, mode) + e);
            }
            __syncthreads();   // one wave: orders this step's LDS write before the neighbours' reads of the next
        }
        for (int rr = 1; rr <= rows; ++rr)
            if (lane < tw) band[(size_t)(p * PL_T - 1 + rr) * g.W + lane] = px[rr][lane];
        if (lane < tw) px[0][lane] = px[PL_T][lane];   // the next pass's row above (unused after a short pass)
        __syncthreads();
    }
    if (cur != nws) bad = true;
    if (bad) atomicOr(status, 1);
}

// ------------------------------------------------------------------ host entry points

size_t plane_bound(int C, int H, int W)
{
    if (C < 1 || H < 1 || W < 1) return 0;
    const PlaneGeom g = plane_geom(C, H, W);
    return ((size_t)g.nstrips + (size_t)g.nstrips * g.capw) * 4;
}

size_t plane_workspace(int C, int H, int W)
{
    if (C < 1 || H < 1 || W < 1) return 0;
    PlaneWs w;
    if (carve_plane(plane_geom(C, H, W), nullptr, &w)) return 0;
    return w.total;
}

int plane_encode(const uint16_t* planes, int C, int H, int W, void* body, size_t body_cap, uint64_t* body_bytes,
                 void* ws, size_t ws_bytes, hipStream_t s)
{
    LBDRN_REQUIRE(planes && body && body_bytes && C >= 1 && H >= 1 && W >= 1, "null pointer or empty raster");
    const PlaneGeom g = plane_geom(C, H, W);
    PlaneWs w;
    if (int rc = carve_plane(g, ws, &w)) return rc;
    if (!ws || ws_bytes < w.total) {
        set_error("plane codec workspace too small: %zu < %zu", ws_bytes, w.total);
        return LBDRN_E_WORKSPACE;
    }
    if (body_cap < plane_bound(C, H, W)) {
        set_error("plane payload buffer too small: %zu < %zu", body_cap, plane_bound(C, H, W));
        return LBDRN_E_WORKSPACE;
    }
    uint32_t* counts = (uint32_t*)body;
    k_plane_encode<<<(unsigned)g.nstrips, PL_T, 0, s>>>(planes, g, w, counts);
    LBDRN_LAUNCH_CHECK();
    LBDRN_HIP_TRY(rocprim::exclusive_scan(w.scan_tmp, w.scan_bytes, counts, w.offsets, (uint64_t)0, (size_t)g.nstrips,
                                          rocprim::plus<uint64_t>(), s));
    k_plane_compact<<<(unsigned)g.nstrips, 256, 0, s>>>(w.words, g.capw, counts, w.offsets, g.nstrips,
                                                         counts + g.nstrips, body_bytes);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

int plane_decode(const void* body, size_t body_bytes, int C, int H, int W, uint16_t* planes, int* status, void* ws,
                 size_t ws_bytes, hipStream_t s)
{
    LBDRN_REQUIRE(planes && body && status && C >= 1 && H >= 1 && W >= 1, "null pointer or empty raster");
    const PlaneGeom g = plane_geom(C, H, W);
    PlaneWs w;
    if (int rc = carve_plane(g, ws, &w)) return rc;
    if (!ws || ws_bytes < w.total) {
        set_error("plane codec workspace too small: %zu < %zu", ws_bytes, w.total);
        return LBDRN_E_WORKSPACE;
    }
    if (body_bytes % 4 || body_bytes < (size_t)g.nstrips * 8) {   // counts + at least one word per strip
        set_error("plane payload of %zu bytes cannot hold %lld strips", body_bytes, (long long)g.nstrips);
        return LBDRN_E_ARG;
    }
    const uint32_t* counts = (const uint32_t*)body;
    const uint64_t total_words = body_bytes / 4 - (uint64_t)g.nstrips;
    LBDRN_HIP_TRY(hipMemsetAsync(status, 0, sizeof(int), s));
    LBDRN_HIP_TRY(rocprim::exclusive_scan(w.scan_tmp, w.scan_bytes, counts, w.offsets, (uint64_t)0, (size_t)g.nstrips,
                                          rocprim::plus<uint64_t>(), s));
    k_plane_decode<<<(unsigned)g.nstrips, PL_T, 0, s>>>(counts, counts + g.nstrips, w.offsets, g, total_words, planes,
                                                        status);
    LBDRN_LAUNCH_CHECK();
    return 0;
}

}  // namespace lbdrn
