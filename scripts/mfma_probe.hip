// Diagnostic microbenchmark (not part of the product): cycles per v_mfma_f32_16x16x4_f32 /
// v_mfma_f32_32x32x2_f32 for 1, 2, 4 independent accumulator chains, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void k16(float* out, unsigned long long* cyc, int iters)
{
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS>
__global__ void k32(float* out, unsigned long long* cyc, int iters)
{
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class K>
void run(const char* name, K kern, int threads, int chains, int nblk)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * nblk); hipMalloc(&cyc, 8 * nblk);
    const int iters = 2000;
    kern<<<nblk, threads>>>(out, cyc, iters);
    kern<<<nblk, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, 8 * nblk, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < nblk; ++i) m += h[i]; m /= nblk;
    int waves_per_simd = threads / 256;
    printf("%-10s threads=%4d chains=%d blocks=%4d : %.1f cycles per MFMA per wave, %.1f per MFMA per SIMD\n", name, threads,
           chains, nblk, m / (iters * 8.0 * chains), m / (iters * 8.0 * chains) / (waves_per_simd ? waves_per_simd : 1));
    hipFree(out); hipFree(cyc);
}
template <class K>
void wall(const char* name, K kern, int threads, int chains, double flop_per_mfma)
{
    float* out; unsigned long long* cyc;
    const int nblk = 256 * (1024 / threads > 2 ? 2 : 1) ;
    hipMalloc(&out, sizeof(float) * threads * nblk); hipMalloc(&cyc, 8 * nblk);
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<nblk, threads>>>(out, cyc, 100);
    hipEventRecord(a);
    kern<<<nblk, threads>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double mfmas = (double)nblk * (threads / 64) * iters * 8.0 * chains;
    printf("WALL %-10s threads=%4d chains=%d blocks=%d : %.2f ms -> %.1f TFLOP/s\n", name, threads, chains, nblk, ms,
           mfmas * flop_per_mfma / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main()
{
    wall("16x16x4", k16<2>, 256, 2, 2048.0);
    wall("16x16x4", k16<2>, 512, 2, 2048.0);
    wall("16x16x4", k16<1>, 1024, 1, 2048.0);
    wall("32x32x2", k32<2>, 256, 2, 4096.0);
    wall("32x32x2", k32<2>, 512, 2, 4096.0);
    for (int nblk : {1, 256}) {
        run("16x16x4", k16<1>, 256, 1, nblk);
        run("16x16x4", k16<2>, 256, 2, nblk);
        run("16x16x4", k16<4>, 256, 4, nblk);
        run("16x16x4", k16<1>, 512, 1, nblk);
        run("16x16x4", k16<2>, 512, 2, nblk);
        run("32x32x2", k32<1>, 256, 1, nblk);
        run("32x32x2", k32<2>, 256, 2, nblk);
        run("32x32x2", k32<2>, 512, 2, nblk);
    }
    return 0;
}
