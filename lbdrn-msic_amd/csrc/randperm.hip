// torch.randperm(n, generator=Generator().manual_seed(seed)) for CPU generators, bit for bit, on the
// GPU (a4: the minibatch order of DataLoader(shuffle=True), ref encode.py:69-70 via
// torch/utils/data/sampler.py:163-183).
//
// What torch does on the host (aten TensorFactories.cpp randperm_cpu, n < 2^32/20): r = 0..n-1, then a
// forward Fisher-Yates pass "for i in 0..n-2: swap(r[i], r[i + mt19937() % (n - i)])" -- 4.19 M
// dependent swaps with random access, 60-300 ms per epoch on a host core, which made the whole fit
// host-bound (ten permutations per image against ~200 ms of GPU work).
//
// Parallel reformulation (exact): position i is final after step i and holds the value that sat at
// j_i = i + z_i just before step i.  That value is p itself if no earlier step targeted p, else it is
// whatever sat at position i' before step i', where i' is the latest step < i with j_i' = p -- a
// chain that walks to ever earlier steps (one hop on average).  So:
//   1. one wave runs the MT19937 recurrence and emits the raw words; the targets j_i are made of them by
//      4 M threads;  2. the steps that share a target are linked into a list per target (one atomic exchange per step:
//      round 4 -- rounds 2-3 counted, scanned and scattered them into buckets: six random accesses per step, now four,
//      and no scan; the pipeline costs a tile 7 ms of the chip with four fits in flight, scripts/perm_cost_probe.py);
//   3. every position walks its own (short) list once and leaves, for every step in it, the latest earlier step with the
//      same target (the first hop of that step's chain) and, for itself, the latest step below it (every later hop);
//   4. every position chases its chain, independently.
// List order is irrelevant (each hop takes the maximum step below a bound), so the order in which the atomics
// land does not affect the result.  tests/test_gpu_randperm.py checks equality with torch.randperm for many (seed, n).
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "mt_jump.inc"

namespace lbdrn {

constexpr int MT_N = 624, MT_M = 397;

__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v)
{
    uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// raw[i] = word i of the mt19937 state sequence (untempered) for i < n-1, one permutation per workgroup.
// The recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1]) is serial at a distance of 227 words, so a permutation is
// ONE wave walking the state in ten 64-word chunks, with no barrier anywhere (4.19 M words: 5.5 ms as a 256-thread
// workgroup with four barriers per 624 words -- the first thing a fit waits for -- against 3 ms here):
//  * a lane keeps "its" ten state words in registers from block to block (word 64c + lane of chunk c);
//  * x[k+1] and x[k+397] / the already renewed x[k-227] come from an LDS copy of the state, written as each chunk is
//    renewed and read three chunks ahead of their use (LDS operations of one wave complete in order; a DPP wave shift
//    for x[k+1] costs more issue slots than the read).
// Tempering and the "% (n - i)" are left to the first consumer, which has 4 M threads to do them with.
//
// Round 4: a permutation longer than MT_SEG words is generated as SEGMENTS side by side (blockIdx.y), each by one wave as
// above: segment 0 from the seeded state, segment s >= 1 from the state window x[s MT_SEG .. +624) that k_mt_jump has
// combined out of the first 20,560 words (mt_jump.inc) -- those come from a first launch of this kernel that stops at
// `limit` and leaves the seeded state in x0.  27 waves for 71 us each instead of one for 1.86 ms.
struct SeedList { uint32_t s[32]; };
struct MtPlan {
    uint32_t seg;         // raw words per segment (a multiple of 624); with one segment in the grid it is not looked at
    uint32_t limit;       // no raw index at or beyond this one is generated (the prefix launch: MT_PREFIX)
    const uint32_t* win;  // [seed][MT_MAX_SEG - 1][624]: the states of segments 1.. (null: one segment)
    uint32_t* x0;         // [seed][624]: the seeded state, for k_mt_jump (the prefix launch; else null)
};
__global__ void __launch_bounds__(64)
    k_mt19937_raw(SeedList seeds, uint32_t n, uint32_t* __restrict__ jall, size_t jstride, MtPlan P)
{
    __shared__ uint32_t st[MT_N + 16];
    const int lane = threadIdx.x;
    const uint32_t sg = blockIdx.y;
    uint32_t* __restrict__ raw = jall + (size_t)blockIdx.x * jstride;
    if (sg == 0) {
        if (lane == 0) {  // init_genrand, as at::mt19937(seed)
            uint32_t x = seeds.s[blockIdx.x];
            st[0] = x;
            for (int k = 1; k < MT_N; ++k) {
                x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)k;
                st[k] = x;
            }
            if (n > 0) raw[n - 1] = n - 1;   // (position n-1 is its own target; the consumers leave it alone)
        }
    } else {
        const uint32_t* __restrict__ w = P.win + ((size_t)blockIdx.x * (MT_MAX_SEG - 1) + (sg - 1)) * MT_N;
        for (int k = lane; k < MT_N; k += 64) st[k] = w[k];
    }
    __syncthreads();
    if (P.x0)
        for (int k = lane; k < MT_N; k += 64) P.x0[(size_t)blockIdx.x * MT_N + k] = st[k];
    const uint32_t all = n > 0 ? n - 1 : 0;
    const uint32_t first = sg * P.seg;
    uint32_t steps = (sg + 1 == gridDim.y || all - first < P.seg) ? all : first + P.seg;   // (the last segment takes what is left)
    steps = steps < P.limit ? steps : P.limit;
    constexpr int NC = 10;   // chunks: words 64c + lane, the last one 48 wide
    uint32_t a[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) a[c] = (64 * c + lane) < MT_N ? st[64 * c + lane] : 0u;
    int midx[NC], bidx[NC];  // where chunk c finds x[k+397] (not renewed yet) or the renewed x[k-227]; and x[k+1]
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = 64 * c + lane;
        midx[c] = k + MT_M < MT_N ? k + MT_M : (k < MT_N ? k - (MT_N - MT_M) : 0);
        bidx[c] = k + 1 < MT_N ? k + 1 : 0;   // x[k+1]; word 623 takes the renewed word 0
    }
    auto block = [&](uint32_t base, bool whole) {   // renew the 624 words, store those below `steps`
        uint32_t m[NC], b[NC];
        uint32_t* __restrict__ dst = raw + base + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) { m[c] = st[midx[c]]; b[c] = st[64 * c + lane + 1]; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t y = (a[c] & 0x80000000u) | (b[c] & 0x7fffffffu);
            const uint32_t v = m[c] ^ (y >> 1) ^ ((b[c] & 1u) ? 0x9908b0dfu : 0u);
            a[c] = v;
            const int k = 64 * c + lane;
            if (c < NC - 1 || k < MT_N) {
                st[k] = v;
                if (whole || base + (uint32_t)k < steps) dst[64 * c] = v;
            }
            // the look-ahead reads below take words OTHER lanes have just stored (x[k+1], the renewed x[k-227]):
            // within one thread the addresses are disjoint, so without this the compiler may hoist them above the
            // store they depend on -- the hardware keeps a wave's LDS operations in order, the compiler has to be
            // told (wave-scope fence + scheduling barrier: no instruction is emitted for either)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (c + 3 < NC) { m[c + 3] = st[midx[c + 3]]; b[c + 3] = st[bidx[c + 3]]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // block boundary: the next block's first reads
        __builtin_amdgcn_wave_barrier();
    };
    uint32_t base = first;
    for (; base + MT_N <= steps; base += MT_N) block(base, true);
    if (base < steps) block(base, false);
}

// the states the segments 1.. start from: win[seed][s-1][m] = XOR over the set bits i of g_s of x[i + m] (mt_jump.inc),
// x[0..624) = the seeded state, x[624 + r] = raw[r].  blockIdx = (segment - 1, seed, eighth of the polynomial): an eighth
// is 78 words of g_s = 2,496 values of i and needs 3,119 words of x -- 15 KB of LDS with the zeros behind them, so that these workgroups fit on a
// CU beside a training workgroup (146.5 of 160 KB) instead of waiting for a free one; the eighths meet in win[] by
// atomic XOR (exact in any order).  The bits of g_s are wave-uniform: the loop over them is scalar, its body one LDS read.
constexpr int JUMP_SPLIT = 8, JUMP_PW = MT_N / JUMP_SPLIT, JUMP_X = JUMP_PW * 32 + MT_N - 1, JUMP_THREADS = 640,
              JUMP_ZERO = (JUMP_X + 31) & ~31;
static_assert(MT_N % JUMP_SPLIT == 0 && (JUMP_SPLIT - 1) * JUMP_PW * 32 + JUMP_X <= MT_N + (int)MT_PREFIX, "k_mt_jump reads x[0 .. 624 + MT_PREFIX)");
__global__ void __launch_bounds__(JUMP_THREADS)
    k_mt_jump(const uint32_t* __restrict__ x0, const uint32_t* __restrict__ jall, size_t jstride,
              const uint32_t* __restrict__ polys, uint32_t* __restrict__ win)
{
    __shared__ uint32_t xs[JUMP_ZERO + JUMP_THREADS];   // the words of x, then zeros: what an absent bit reads
    const int tid = threadIdx.x;
    xs[JUMP_ZERO + tid] = 0u;
    const uint32_t* __restrict__ raw = jall + (size_t)blockIdx.y * jstride;
    const int i0 = blockIdx.z * JUMP_PW * 32;
    for (int k = tid; k < JUMP_X; k += JUMP_THREADS) {
        const int idx = i0 + k;
        xs[k] = idx < MT_N ? x0[(size_t)blockIdx.y * MT_N + idx] : raw[idx - MT_N];
    }
    __syncthreads();
    const uint32_t* __restrict__ g = polys + (size_t)blockIdx.x * MT_N + blockIdx.z * JUMP_PW;
    // the 78 words of the polynomial sit in two registers across the wave and are picked by v_readlane: a load per
    // word would put a memory wait (the counter LDS reads share) into every turn of the loop
    static_assert(JUMP_PW <= 128, "two registers hold the eighth");
    const int lane = tid & 63;
    const uint32_t g0 = lane < JUMP_PW ? g[lane] : 0u, g1 = 64 + lane < JUMP_PW ? g[64 + lane] : 0u;
    uint32_t acc = 0;
    for (int w = 0; w < JUMP_PW; ++w) {
        uint32_t gw = w < 64 ? __builtin_amdgcn_readlane(g0, w) : __builtin_amdgcn_readlane(g1, w - 64);
        while (gw) {   // four set bits per turn, so that four reads are in flight; a bit that is not there reads a zero
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int off = gw ? 32 * w + __builtin_ctz(gw) : JUMP_ZERO;
                gw &= gw - 1;   // (0 stays 0)
                v[u] = xs[off + tid];
            }
            acc ^= (v[0] ^ v[1]) ^ (v[2] ^ v[3]);
        }
    }
    if (tid < MT_N) atomicXor(&win[((size_t)blockIdx.y * (MT_MAX_SEG - 1) + blockIdx.x) * MT_N + tid], acc);   // (every lane stays for the readlanes)
}

__device__ __forceinline__ uint32_t mt_target(uint32_t raw, uint32_t i, uint32_t n) { return i + mt_temper(raw) % (n - i); }

// j[i] = i + mt19937_output(i) % (n - i) for i < n-1, in place over the raw words; step i is pushed onto the list of
// its target: head[t] = i, next[i] = the previous head (-1: end).  ONE random access per step (the exchange); next[] is
// written where the step sits.
__global__ void __launch_bounds__(256)
    k_link_targets(uint32_t* __restrict__ j, uint32_t steps, uint32_t n, int32_t* __restrict__ head, int32_t* __restrict__ next)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < steps) {
        const uint32_t t = mt_target(j[i], i, n);
        j[i] = t;
        next[i] = atomicExch(&head[t], (int32_t)i);
    }
}

// The chains: position p walks its own list once (the steps that target it: one on average, ~ln n at the far end) and
// leaves, for every step x in it, the latest earlier step with the same target (pred[x]: the first hop of x's chain),
// and for itself the latest step below p that targets it (last[p]: every later hop -- the steps that target p are <= p).
// last overwrites head in place (a thread reads only its own head; the walks read next[] alone).
constexpr int LINK_REG = 8;   // list lengths handled in registers; longer lists (a vanishing share) walk again per element
__global__ void __launch_bounds__(256)
    k_links(uint32_t n, int32_t* head_last, const int32_t* __restrict__ next, int32_t* __restrict__ pred)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t h = head_last[p];
    int32_t e[LINK_REG];
    int len = 0;
    int32_t last = -1;
    for (int32_t x = h; x >= 0; x = next[x]) {
        if (len < LINK_REG) e[len] = x;
        ++len;
        if ((uint32_t)x < p && x > last) last = x;
    }
    if (len <= LINK_REG) {
#pragma unroll
        for (int a = 0; a < LINK_REG; ++a) {
            if (a < len) {
                int32_t best = -1;
#pragma unroll
                for (int b = 0; b < LINK_REG; ++b)
                    if (b < len && e[b] < e[a] && e[b] > best) best = e[b];
                pred[e[a]] = best;
            }
        }
    } else {
        for (int32_t x = h; x >= 0; x = next[x]) {
            int32_t best = -1;
            for (int32_t y = h; y >= 0; y = next[y])
                if (y < x && y > best) best = y;
            pred[x] = best;
        }
    }
    head_last[p] = last;
    if (p == n - 1) pred[p] = last;   // the last position is nobody's step: its chain starts at its own list
}

// ---- the partitioned path (n <= PART_MAX_N; round 4).  The list building above costs one memory-side atomic and two more
// random accesses per step, and with fits in flight every one of them competes with the training launches' gathers
// (the pipeline cost a tile 7 ms of 63).  Here the steps are first routed to the partition of 2048 POSITIONS their target
// falls into -- a counting pass, a scan of the (partition, workgroup) counts and a scatter of (step, target) pairs, all
// coalesced or in 64-byte pieces --, and one workgroup per partition then builds that partition's lists with LDS atomics
// (its heads are 8 KB of LDS), walking them through the partition's own pairs (L2-resident).  What is left of the
// random traffic is one 4-byte write per step (pred[step]) and the chase.
// Partition = 2048 positions: the far end of the array is dense with targets (position p is hit ln(n / (n - p)) times on
// average: 8.6 per position in the last partition against 1 overall), and a partition's workgroup walks its lists one
// dependent load at a time -- 8192-position partitions left one workgroup with 59 k pairs and the launch at 0.7 ms.
constexpr int PART_SHIFT = 11, PART_SIZE = 1 << PART_SHIFT, PART_MAX = 2048;
constexpr int64_t PART_MAX_N = (int64_t)PART_MAX << PART_SHIFT;   // 4,194,304: the 2048^2 tile
constexpr int PART_CHUNK = 16384;   // steps per workgroup of the counting / scattering passes: eight pairs = 64 bytes per partition on average

// The permutations of one call go through every launch TOGETHER (blockIdx.y = permutation, each with its own arrays at
// a fixed stride): a launch boundary is an L2 write-back + invalidate across the XCDs, felt by the training launches of
// the fits in flight -- six launches per batch of nine permutations instead of fifty-four.
struct PartArrays {
    uint32_t *j, *histT, *part_start;   // per permutation: + c * stride
    uint2* pairs;
    int2* links;
    int32_t *pred, *last;
    size_t sj, shist, sstart, sn;       // strides in elements: targets, histT, part_start, everything of n entries
};

// targets (in place over the raw words) + this workgroup's count per partition: histT[partition][workgroup]
__global__ void __launch_bounds__(256) k_part_count(PartArrays A, uint32_t steps, uint32_t n, int npart, int nwg)
{
    uint32_t* __restrict__ j = A.j + blockIdx.y * A.sj;
    uint32_t* __restrict__ histT = A.histT + blockIdx.y * A.shist;
    __shared__ uint32_t hist[PART_MAX];
    for (int k = threadIdx.x; k < PART_MAX; k += 256) hist[k] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * PART_CHUNK;
#pragma unroll 4
    for (int u = 0; u < PART_CHUNK / 256; ++u) {
        const uint32_t i = base + u * 256 + threadIdx.x;
        if (i < steps) {
            const uint32_t t = mt_target(j[i], i, n);
            j[i] = t;
            atomicAdd(&hist[t >> PART_SHIFT], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < npart; k += 256) histT[(size_t)k * nwg + blockIdx.x] = hist[k];
}

// exclusive scan of every partition's row of workgroup counts (in place) and the row totals; then the totals themselves:
// a pair of (partition p, workgroup w) starts at part_start[p] + histT[p][w]
__global__ void __launch_bounds__(1024) k_part_scan_rows(uint32_t* __restrict__ histT0, size_t sh, int nwg, uint32_t* __restrict__ totals0, size_t st)
{
    __shared__ uint32_t sums[1024];
    const int tid = threadIdx.x;
    uint32_t* __restrict__ histT = histT0 + blockIdx.y * sh;
    uint32_t* __restrict__ totals = totals0 + blockIdx.y * st;
    uint32_t* row = histT + (size_t)blockIdx.x * nwg;
    uint32_t run = 0;   // (nwg <= 1024 at the sizes this path takes; the loop keeps it general)
    for (int base = 0; base < nwg; base += 1024) {
        const int k = base + tid;
        const uint32_t c = k < nwg ? row[k] : 0u;
        sums[tid] = c;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t v = tid >= d ? sums[tid - d] : 0u;
            __syncthreads();
            sums[tid] += v;
            __syncthreads();
        }
        if (k < nwg) row[k] = run + sums[tid] - c;
        run += sums[1023];
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = run;   // (second use, one row = the partitions' totals: totals = &part_start[npart])
}
// (step, target) pairs in partition order
__global__ void __launch_bounds__(256)
    k_part_scatter(PartArrays A, uint32_t steps, int npart, int nwg)
{
    const uint32_t* __restrict__ j = A.j + blockIdx.y * A.sj;
    const uint32_t* __restrict__ offs = A.histT + blockIdx.y * A.shist;
    const uint32_t* __restrict__ part_start = A.part_start + blockIdx.y * A.sstart;
    uint2* __restrict__ pairs = A.pairs + blockIdx.y * A.sn;
    __shared__ uint32_t cursor[PART_MAX];
    for (int k = threadIdx.x; k < npart; k += 256) cursor[k] = part_start[k] + offs[(size_t)k * nwg + blockIdx.x];
    __syncthreads();
    const uint32_t base = blockIdx.x * PART_CHUNK;
#pragma unroll 4
    for (int u = 0; u < PART_CHUNK / 256; ++u) {
        const uint32_t i = base + u * 256 + threadIdx.x;
        if (i < steps) {
            const uint32_t t = j[i];
            pairs[atomicAdd(&cursor[t >> PART_SHIFT], 1u)] = make_uint2(i, t);
        }
    }
}

// one workgroup per partition: lists of its positions out of its pairs (heads in LDS; a link = (previous head, step) per
// pair index, so that a walk is one 8-byte load per entry), then for every position the predecessors of its list's steps
// (pred[step]) and the latest step below it (last[position]).  Lists up to LINK_REG entries are ordered in registers,
// longer ones (the far end of the array: ~ln n entries) in a per-thread strip of LDS.
#ifndef LBDRN_LINK_LDS
#define LBDRN_LINK_LDS 24
#endif
constexpr int LINK_LDS = LBDRN_LINK_LDS;   // entries of a thread's LDS strip; longer lists (a handful per permutation) re-walk
constexpr int LINK_THREADS = 512;   // the walks are chains of dependent loads: sixteen waves per CU keep more of them in flight
                                    // (256 threads: 204 us per launch, 512: 141; 1024 with strips of 12 entries: 206)
__global__ void __launch_bounds__(LINK_THREADS)
    k_part_links(PartArrays A, uint32_t n)
{
    const uint32_t* __restrict__ part_start = A.part_start + blockIdx.y * A.sstart;
    const uint2* __restrict__ pairs = A.pairs + blockIdx.y * A.sn;
    int2* __restrict__ links = A.links + blockIdx.y * A.sn;
    int32_t* __restrict__ pred = A.pred + blockIdx.y * A.sn;
    int32_t* __restrict__ last = A.last + blockIdx.y * A.sn;
    __shared__ int32_t head[PART_SIZE];
    __shared__ int32_t strip[LINK_THREADS][LINK_LDS + 1];   // (+1: the strips of neighbouring threads start in different banks)
    const uint32_t part = gridDim.x - 1 - blockIdx.x;   // the dense partitions (the far end of the array) first: they take ten times as long
    const uint32_t base = part << PART_SHIFT;
    const uint32_t k0 = part_start[part], k1 = part_start[part + 1];
    for (int k = threadIdx.x; k < PART_SIZE; k += LINK_THREADS) head[k] = -1;
    __syncthreads();
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += LINK_THREADS) {
        const uint2 pr = pairs[k];
        links[k] = make_int2(atomicExch(&head[pr.y - base], (int32_t)k), (int32_t)pr.x);
    }
    __threadfence_block();
    __syncthreads();
    int32_t* mine = strip[threadIdx.x];
    for (uint32_t lp = threadIdx.x; lp < PART_SIZE; lp += LINK_THREADS) {
        const uint32_t p = base + lp;
        if (p >= n) break;
        const int32_t h = head[lp];
        int32_t e[LINK_REG];
        int len = 0;
        int32_t lastv = -1;
        for (int32_t k = h; k >= 0;) {
            const int2 l = links[k];
            const int32_t x = l.y;
            if (len < LINK_REG) e[len] = x;
            if (len < LINK_LDS) mine[len] = x;
            ++len;
            if ((uint32_t)x < p && x > lastv) lastv = x;
            k = l.x;
        }
        if (len <= LINK_REG) {
#pragma unroll
            for (int a = 0; a < LINK_REG; ++a) {
                if (a < len) {
                    int32_t best = -1;
#pragma unroll
                    for (int b = 0; b < LINK_REG; ++b)
                        if (b < len && e[b] < e[a] && e[b] > best) best = e[b];
                    if (best >= 0) pred[e[a]] = best;   // (pred[] starts at -1: two steps in three have no earlier step with their target)
                }
            }
        } else if (len <= LINK_LDS) {
            for (int a = 0; a < len; ++a) {
                const int32_t x = mine[a];
                int32_t best = -1;
                for (int b = 0; b < len; ++b) {
                    const int32_t y = mine[b];
                    if (y < x && y > best) best = y;
                }
                if (best >= 0) pred[x] = best;
            }
        } else {
            for (int32_t k = h; k >= 0; k = links[k].x) {
                const int32_t x = links[k].y;
                int32_t best = -1;
                for (int32_t m = h; m >= 0; m = links[m].x) {
                    const int32_t y = links[m].y;
                    if (y < x && y > best) best = y;
                }
                if (best >= 0) pred[x] = best;
            }
        }
        last[p] = lastv;
        if (p == n - 1) pred[p] = lastv;   // the last position is nobody's step: its chain starts at its own list
    }
}

__global__ void __launch_bounds__(256)
    k_chase(const uint32_t* __restrict__ j, uint32_t n, const int32_t* __restrict__ pred,
            const int32_t* __restrict__ last, int64_t* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t s = pred[i];
    uint32_t p = j[i];
    while (s >= 0) {
        p = (uint32_t)s;
        s = last[p];
    }
    out[i] = p;
}

// the chase of every permutation of the call (the partitioned path's arrays)
__global__ void __launch_bounds__(256) k_chase_batch(PartArrays A, uint32_t n, int64_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* __restrict__ j = A.j + blockIdx.y * A.sj;
    const int32_t* __restrict__ pred = A.pred + blockIdx.y * A.sn;
    const int32_t* __restrict__ last = A.last + blockIdx.y * A.sn;
    int32_t s = pred[i];
    uint32_t p = j[i];
    while (s >= 0) {
        p = (uint32_t)s;
        s = last[p];
    }
    out[(size_t)blockIdx.y * n + i] = p;
}

struct PermWs {
    uint32_t *j, *cnt, *off, *cursor;
    uint2* pairs;            // partitioned path: (step, target) in partition order
    int2* links;             // (previous head of the target's list, step) per pair
    int32_t *pred, *last;
    uint32_t *histT, *part_start;
    uint32_t *x0, *win;      // seeded states [count][624]; segment states [count][MT_MAX_SEG - 1][624] (mt_jump.inc)
    size_t sn, shist, sstart;
    size_t arr, total;
};
static bool perm_partitioned(int64_t n)
{
#ifdef LBDRN_EXP_RANDPERM_ATOMIC   // (A/B build: the memory-side atomic path for every n)
    return false;
#else
    return n > 16 * PART_SIZE && n <= PART_MAX_N;
#endif
}

static int carve_perm(int64_t n, int count, void* ws, PermWs* w)
{
    const size_t arr = align_up((size_t)std::max<int64_t>(n, 1) * sizeof(uint32_t), 256);
    char* p = (char*)ws;
    w->j = (uint32_t*)p; p += arr * count;     // one target array per permutation
    w->cnt = (uint32_t*)p; p += arr;            // head[], then last[]
    w->cursor = (uint32_t*)p; p += arr;         // pred[]
    w->off = (uint32_t*)p; p += arr;            // next[]
    w->pairs = nullptr; w->links = nullptr; w->histT = nullptr; w->part_start = nullptr; w->pred = nullptr; w->last = nullptr;
    if (perm_partitioned(n)) {   // every permutation of the call has its own arrays: they go through the launches together
        const size_t nwg = (size_t)((n + PART_CHUNK - 1) / PART_CHUNK), npart = (size_t)((n + PART_SIZE - 1) >> PART_SHIFT);
        w->sn = align_up((size_t)n * sizeof(uint2), 256) / sizeof(uint2);          // stride of the n-entry arrays, in entries
        w->shist = align_up(nwg * npart * sizeof(uint32_t), 256) / sizeof(uint32_t);
        w->sstart = align_up((npart + 1) * sizeof(uint32_t), 256) / sizeof(uint32_t);
        w->pairs = (uint2*)p; p += w->sn * sizeof(uint2) * count;
        w->links = (int2*)p; p += w->sn * sizeof(int2) * count;
        w->pred = (int32_t*)p; p += w->sn * sizeof(int32_t) * count;
        w->last = (int32_t*)p; p += w->sn * sizeof(int32_t) * count;
        w->histT = (uint32_t*)p; p += w->shist * sizeof(uint32_t) * count;
        w->part_start = (uint32_t*)p; p += w->sstart * sizeof(uint32_t) * count;
    }
    w->x0 = (uint32_t*)p; p += align_up((size_t)count * MT_N * sizeof(uint32_t), 256);
    w->win = (uint32_t*)p; p += align_up((size_t)count * (MT_MAX_SEG - 1) * MT_N * sizeof(uint32_t), 256);
    w->arr = arr;
    w->total = (size_t)(p - (char*)ws);
    return 0;
}

int64_t mt19937_jump_poly(int segment, uint32_t* out)
{
    LBDRN_REQUIRE(segment >= 1 && segment < MT_MAX_SEG, "segment must be 1 .. 31");
    const std::vector<uint32_t>& tab = mt_jump_polys();
    memcpy(out, tab.data() + (size_t)(segment - 1) * MT_N, MT_N * sizeof(uint32_t));
    return (int64_t)MT_SEG;
}

size_t randperm_workspace(int64_t n, int count)
{
    PermWs w;
    if (count < 1 || carve_perm(n, count, nullptr, &w)) return 0;
    return w.total;
}

// out[c][0..n) = torch.randperm(n, generator=manual_seed(seeds[c])) for c < count (count <= 32)
int randperm_batch(const uint64_t* seeds, int count, int64_t n, int64_t* out, void* ws, size_t ws_bytes,
                   hipStream_t s)
{
    LBDRN_REQUIRE(n >= 0 && count >= 1 && count <= 32 && seeds, "bad n / count (1..32) / seeds");
    if (n >= (int64_t)(0xffffffffu / 20)) {
        set_error("n=%lld: torch switches to another algorithm at n >= 2^32/20", (long long)n);
        return LBDRN_E_UNSUPPORTED;
    }
    if (n == 0) return 0;
    PermWs w;
    if (int rc = carve_perm(n, count, ws, &w)) return rc;
    if (!ws || ws_bytes < w.total) {
        set_error("randperm workspace too small: %zu < %zu", ws_bytes, w.total);
        return LBDRN_E_WORKSPACE;
    }
    const uint32_t un = (uint32_t)n, steps = un - 1;
    SeedList sl;
    for (int c = 0; c < 32; ++c) sl.s[c] = c < count ? (uint32_t)(seeds[c] & 0xffffffffu) : 0u;
    // timing-only builds (never the shipped library: the permutations are then garbage; the training kernels clamp what
    // they read): -DLBDRN_EXP_RANDPERM_DIAG=1 no MT19937 launch, =2 none of the launches behind it, =3 neither
#ifdef LBDRN_EXP_RANDPERM_DIAG
    constexpr int diag = LBDRN_EXP_RANDPERM_DIAG;
#else
    constexpr int diag = 0;
#endif
#ifdef LBDRN_EXP_RANDPERM_NOJUMP   // (A/B build: one wave per permutation from end to end)
    constexpr bool no_jump = true;
#else
    constexpr bool no_jump = false;
#endif
    const int nseg = no_jump ? 1 : (int)std::min<uint32_t>(MT_MAX_SEG, (steps + MT_SEG - 1) / MT_SEG);
    const size_t jstride = w.arr / sizeof(uint32_t);
    if (!(diag & 1)) {
        if (nseg > 1) {
            const uint32_t* polys = nullptr;
            if (int rc = mt_jump_table_device(&polys)) return rc;
            k_mt19937_raw<<<dim3(count, 1), 64, 0, s>>>(sl, un, w.j, jstride, MtPlan{0u, MT_PREFIX, nullptr, w.x0});
            LBDRN_LAUNCH_CHECK();
            LBDRN_HIP_TRY(hipMemsetAsync(w.win, 0, (size_t)count * (MT_MAX_SEG - 1) * MT_N * sizeof(uint32_t), s));
            k_mt_jump<<<dim3(nseg - 1, count, JUMP_SPLIT), JUMP_THREADS, 0, s>>>(w.x0, w.j, jstride, polys, w.win);
            LBDRN_LAUNCH_CHECK();
            k_mt19937_raw<<<dim3(count, nseg), 64, 0, s>>>(sl, un, w.j, jstride, MtPlan{MT_SEG, 0xffffffffu, w.win, nullptr});
        } else {
            k_mt19937_raw<<<dim3(count, 1), 64, 0, s>>>(sl, un, w.j, jstride, MtPlan{0u, 0xffffffffu, nullptr, nullptr});
        }
        LBDRN_LAUNCH_CHECK();
    }
    if (diag & 2) return 0;
    if (w.pairs && steps) {   // the partitioned path: all permutations of the call per launch
        const int nwg = (int)((steps + PART_CHUNK - 1) / PART_CHUNK), npart = (int)((un + PART_SIZE - 1) >> PART_SHIFT);
        PartArrays A;
        A.j = w.j; A.histT = w.histT; A.part_start = w.part_start; A.pairs = w.pairs; A.links = w.links; A.pred = w.pred; A.last = w.last;
        A.sj = w.arr / sizeof(uint32_t); A.shist = w.shist; A.sstart = w.sstart; A.sn = w.sn;
        LBDRN_HIP_TRY(hipMemsetAsync(w.pred, 0xFF, w.sn * sizeof(int32_t) * count, s));   // pred[] = -1: only the steps that HAVE a predecessor are written (at random)
        k_part_count<<<dim3(nwg, count), 256, 0, s>>>(A, steps, un, npart, nwg);
        LBDRN_LAUNCH_CHECK();
        k_part_scan_rows<<<dim3(npart, count), 1024, 0, s>>>(w.histT, w.shist, nwg, w.part_start, w.sstart);   // row totals -> part_start[partition]
        LBDRN_LAUNCH_CHECK();
        k_part_scan_rows<<<dim3(1, count), 1024, 0, s>>>(w.part_start, w.sstart, npart, w.part_start + npart, w.sstart);   // their exclusive scan in place, the grand total behind
        LBDRN_LAUNCH_CHECK();
        k_part_scatter<<<dim3(nwg, count), 256, 0, s>>>(A, steps, npart, nwg);
        LBDRN_LAUNCH_CHECK();
        k_part_links<<<dim3(npart, count), LINK_THREADS, 0, s>>>(A, un);
        LBDRN_LAUNCH_CHECK();
        k_chase_batch<<<dim3((un + 255) / 256, count), 256, 0, s>>>(A, un, out);
        LBDRN_LAUNCH_CHECK();
        return 0;
    }
    for (int c = 0; c < count; ++c) {
        uint32_t* j = w.j + (size_t)c * (w.arr / sizeof(uint32_t));
        LBDRN_HIP_TRY(hipMemsetAsync(w.cnt, 0xFF, w.arr, s));   // head[] = -1
        if (steps) {
            k_link_targets<<<(steps + 255) / 256, 256, 0, s>>>(j, steps, un, (int32_t*)w.cnt, (int32_t*)w.off);
            LBDRN_LAUNCH_CHECK();
        }
        k_links<<<(un + 255) / 256, 256, 0, s>>>(un, (int32_t*)w.cnt, (const int32_t*)w.off, (int32_t*)w.cursor);
        LBDRN_LAUNCH_CHECK();
        k_chase<<<(un + 255) / 256, 256, 0, s>>>(j, un, (const int32_t*)w.cursor, (const int32_t*)w.cnt, out + (size_t)c * n);
        LBDRN_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace lbdrn
