// Diagnostic microbenchmark (not part of the product): do f32 MFMAs and f32 VALU work overlap on a SIMD?
// Four kernels on 256 workgroups x 512 threads (two waves per SIMD), wall-clocked:
//   M: every wave issues only v_mfma_f32_32x32x2_f32 (two accumulator chains)
//   V: every wave issues only v_fma_f32 (32 independent chains) -- or v_pk_fma_f32 with PK
//   S: the waves selected by `sel` run M's loop, the others V's loop (sel 0: wave >= 4, sel 1: wave & 1)
//   I: every wave interleaves 1 MFMA with NV VALU instructions
// If the pipes were independent, S and I would take max(M, V); if they share a datapath, about M + V.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int MF_PER_IT = 8;     // MFMAs per loop trip
constexpr int VA_PER_IT = 128;   // VALU fmas per loop trip (M-only time ~ 8*64 = 512 cyc/wave-alone; V-only ~ 128*4)

__device__ __forceinline__ void mfma_loop(f32x16 (&acc)[2], float a, float b, int iters)
{
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < MF_PER_IT / 2; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
        }
    }
}
template <bool PK>
__device__ __forceinline__ void valu_loop(float (&v)[32], float a, float b, int iters)
{
    for (int it = 0; it < iters; ++it) {
        if constexpr (PK) {
#pragma unroll
            for (int u = 0; u < VA_PER_IT / 16; ++u)
#pragma unroll
                for (int c = 0; c < 32; c += 2) {
                    f32x2 x = {v[c], v[c + 1]}, aa = {a, a}, bb = {b, b};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(aa), "v"(bb));
                    v[c] = x[0]; v[c + 1] = x[1];
                }
        } else {
#pragma unroll
            for (int u = 0; u < VA_PER_IT / 32; ++u)
#pragma unroll
                for (int c = 0; c < 32; ++c) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(v[c]) : "v"(a), "v"(b));
        }
    }
}

// mode 0: M   1: V   2: S(sel 0)   3: S(sel 1)   4: I
template <int MODE, bool PK>
__global__ void __launch_bounds__(512) k(float* out, int iters)
{
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float v[32];
    for (int c = 0; c < 32; ++c) v[c] = threadIdx.x * 1e-3f + c;
    float a = 0.999f + threadIdx.x * 1e-6f, b = 1e-3f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MODE == 0) mfma_loop(acc, a, b, iters);
    else if (MODE == 1) valu_loop<PK>(v, a, b, iters);
    else if (MODE == 2) { if (wave >= 4) mfma_loop(acc, a, b, iters); else valu_loop<PK>(v, a, b, iters); }
    else if (MODE == 3) { if (wave & 1) mfma_loop(acc, a, b, iters); else valu_loop<PK>(v, a, b, iters); }
    else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < MF_PER_IT; ++u) {
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < VA_PER_IT / MF_PER_IT; ++c)
                    asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(v[(u * 16 + c) & 31]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0;
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int c = 0; c < 32; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
float wall(K kern, float* out, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<256, 512>>>(out, 50);
    hipEventRecord(a);
    kern<<<256, 512>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main()
{
    float* out; hipMalloc(&out, sizeof(float) * 512 * 256);
    const int iters = 4000;
    const double clk = 2.4e9;
    auto rep = [&](const char* n, float ms, double mf_waves, double va_waves) {
        double cyc = ms * 1e-3 * clk;
        printf("%-34s %7.3f ms = %9.0f cycles", n, ms, cyc);
        if (mf_waves > 0) printf("  | %.1f cyc per MFMA per SIMD", cyc / (iters * MF_PER_IT * mf_waves));
        if (va_waves > 0) printf("  | %.2f cyc per VALU per SIMD", cyc / (iters * (double)VA_PER_IT * va_waves));
        printf("\n");
    };
    rep("M  (8 waves mfma)", wall(k<0, false>, out, iters), 2, 0);
    rep("V  (8 waves v_fma_f32)", wall(k<1, false>, out, iters), 0, 2);
    rep("Vp (8 waves v_pk_fma_f32)", wall(k<1, true>, out, iters), 0, 2);
    rep("S0 (waves>=4 mfma, others fma)", wall(k<2, false>, out, iters), 1, 1);
    rep("S1 (odd waves mfma, others fma)", wall(k<3, false>, out, iters), 1, 1);
    rep("S1p (odd waves mfma, others pk)", wall(k<3, true>, out, iters), 1, 1);
    rep("I  (every wave: 1 mfma + 16 fma)", wall(k<4, false>, out, iters), 2, 2);
    return 0;
}
