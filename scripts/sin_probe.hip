// Probe: accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against double sin/cos over the
// argument range the training step sees (30 z, |z| up to a few units).  hipcc --offload-arch=gfx950 -O2 -o sin_probe sin_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = x[i] * 0.15915494309189535f;          // revolutions
    const float f = __builtin_amdgcn_fractf(r);
    s[i] = __builtin_amdgcn_sinf(f);
    c[i] = __builtin_amdgcn_cosf(f);
}
int main()
{
    const int n = 1 << 22;
  for (float R : {3.0f, 20.0f, 120.0f}) {
    std::vector<float> x(n), s(n), c(n);
    for (int i = 0; i < n; ++i) x[i] = -R + 2 * R * (float)i / n;
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, rs = 0;
    for (int i = 0; i < n; ++i) {
        es = std::fmax(es, std::fabs((double)s[i] - std::sin((double)x[i])));
        ec = std::fmax(ec, std::fabs((double)c[i] - std::cos((double)x[i])));
        rs += ((double)s[i] - std::sin((double)x[i])) * ((double)s[i] - std::sin((double)x[i]));
    }
    printf("|x| <= %g: ", R);
    printf("max abs err sin %.3e cos %.3e  rms sin %.3e\n", es, ec, std::sqrt(rs / n));
  }
    return 0;
}
