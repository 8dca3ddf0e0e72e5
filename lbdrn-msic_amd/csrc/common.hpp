// Shared host-side helpers of liblbdrn_hip: error string, HIP call checking, geometry helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lbdrn_hip.h"

namespace lbdrn {

void set_error(const char* fmt, ...);

#define LBDRN_HIP_TRY(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ::lbdrn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                               __LINE__);                                                    \
            return LBDRN_E_DEVICE;                                                           \
        }                                                                                    \
    } while (0)

#define LBDRN_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            ::lbdrn::set_error(__VA_ARGS__); \
            return LBDRN_E_ARG;             \
        }                                   \
    } while (0)

// kernels are launched with <<<>>>; this turns a launch failure into a status
#define LBDRN_LAUNCH_CHECK() LBDRN_HIP_TRY(hipGetLastError())

inline int64_t param_count(const lbdrn_net& n)
{
    int64_t c = 0;
    for (int l = 0; l < n.nl; ++l) c += (int64_t)n.bc * (l ? n.bc : n.F) + n.bc;
    return c + (int64_t)n.C * n.bc + n.C;
}
// offset of layer l's weight matrix in the flat parameter vector (l == nl: last layer)
inline int64_t layer_offset(const lbdrn_net& n, int l)
{
    int64_t c = 0;
    for (int i = 0; i < l; ++i) c += (int64_t)n.bc * (i ? n.bc : n.F) + n.bc;
    return c;
}
inline int feature_dim(const lbdrn_geom& g)
{
    const int side = 2 * g.D + 1;
    return 2 * g.P + (g.use_colors ? g.C * side * side : 0);
}
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int check_geom(const lbdrn_geom* g);
int check_net(const lbdrn_net* n);

// ---- generic path (generic.hip)
int generic_split_bits(const uint16_t* img, int C, int H, int W, int K, uint16_t* msb, int32_t* mx,
                       hipStream_t s);
int generic_labels(const uint16_t* img, int C, int H, int W, int K, const int64_t* idx, int64_t n,
                   float* labels, hipStream_t s);
int generic_features(const lbdrn_geom& g, const uint16_t* msb, const int64_t* idx, int64_t n,
                     int64_t first, float* out, hipStream_t s);
size_t generic_forward_workspace(const lbdrn_net& net, int64_t B);
int generic_forward(const lbdrn_net& net, const float* params, const float* x, int64_t B, float* y,
                    void* ws, size_t ws_bytes, hipStream_t s);
size_t generic_apply_workspace(const lbdrn_geom& g, const lbdrn_net& net);
int generic_decode(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* msb,
                   const float* params, uint16_t* out, float* y_out, void* ws, size_t ws_bytes,
                   hipStream_t s);
int generic_eval_sse(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                     const uint16_t* msb, const float* params, double* sse, void* ws,
                     size_t ws_bytes, hipStream_t s);
size_t generic_train_workspace(const lbdrn_net& net, int B);
int generic_train_step(const lbdrn_net& net, const float* x, const float* t, int B, float* params,
                       float* m, float* v, int64_t adam_step, double lr, int apply_adam,
                       float* loss, float* grads, void* ws, size_t ws_bytes, hipStream_t s);
int generic_train_epoch(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                        const uint16_t* msb, const int64_t* perm, int64_t n, int bs, float* params,
                        float* m, float* v, int64_t step0, double lr, float* losses, void* ws,
                        size_t ws_bytes, hipStream_t s);

// ---- fused MFMA path (apply_mfma.hip)
bool mfma_apply_supported(const lbdrn_geom& g, const lbdrn_net& net);
size_t mfma_apply_workspace(const lbdrn_geom& g, const lbdrn_net& net);
int mfma_decode(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* msb, const float* params,
                uint16_t* out, float* y_out, void* ws, size_t ws_bytes, hipStream_t s);
int mfma_eval_sse(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                  const uint16_t* msb, const float* params, double* sse, void* ws, size_t ws_bytes,
                  bool background, bool fast, bool x16, hipStream_t s);

int mfma_train_epoch_group(int count, const lbdrn_geom& g, const lbdrn_net& net, const int64_t* const* perm, int64_t n,
                           int bs, float* const* params, float* const* m, float* const* v, int64_t step0, double lr,
                           float* const* losses, void* const* ws, size_t ws_bytes, hipStream_t s, bool alone);

}  // namespace lbdrn
