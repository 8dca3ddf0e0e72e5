// extern "C" entry points of liblbdrn_hip.so (declared in include/lbdrn_hip.h).
// Argument checking, path selection (fused MFMA kernels vs generic kernels) and nothing else.
#include <string>

#include "common.hpp"

namespace lbdrn {
const char* last_error();
bool mfma_train_supported(const lbdrn_geom& g, const lbdrn_net& net, int bs = 0);
int mfma_train_step_features(const lbdrn_geom& g, const lbdrn_net& net);
bool mfma_train_takes_groups(const lbdrn_geom& g, const lbdrn_net& net);
size_t mfma_train_workspace(const lbdrn_geom& g, const lbdrn_net& net, int bs);
int mfma_train_prepare(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                       const uint16_t* msb, int bs, void* ws, size_t ws_bytes, hipStream_t s);
int mfma_train_epoch(const lbdrn_geom& g, const lbdrn_net& net, const uint16_t* img,
                     const uint16_t* msb, const int64_t* perm, int64_t n, int bs, float* params,
                     float* m, float* v, int64_t step0, double lr, float* losses, void* ws,
                     size_t ws_bytes, hipStream_t s, bool alone);
int train_profile_mode(int mode);
size_t randperm_workspace(int64_t n, int count);
size_t plane_bound(int C, int H, int W);
size_t plane_workspace(int C, int H, int W);
int plane_encode(const uint16_t* planes, int C, int H, int W, void* body, size_t body_cap, uint64_t* body_bytes,
                 void* ws, size_t ws_bytes, hipStream_t s);
int plane_decode(const void* body, size_t body_bytes, int C, int H, int W, uint16_t* planes, int* status, void* ws,
                 size_t ws_bytes, hipStream_t s);
int64_t mt19937_jump_poly(int segment, uint32_t* out);
int randperm_batch(const uint64_t* seeds, int count, int64_t n, int64_t* out, void* ws, size_t ws_bytes,
                   hipStream_t s);
size_t weights_bound(int64_t n);
int weights_encode(const float* values, int64_t n, int precision, uint8_t* out, size_t cap, size_t* nbytes);
int weights_info(const uint8_t* in, size_t nbytes, int64_t* n, int* precision);
int weights_decode(const uint8_t* in, size_t nbytes, float* values, int64_t cap);
}  // namespace lbdrn

using namespace lbdrn;

static int device_ok()
{
    static thread_local int cached = 1;  // 1 = unknown, 0 = ok, <0 = error
    if (cached != 1) return cached;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) {
        set_error("no HIP device available (%s); liblbdrn_hip has no CPU path",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return cached = LBDRN_E_DEVICE;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_error("cannot query the current HIP device");
        return cached = LBDRN_E_DEVICE;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return cached = LBDRN_E_DEVICE;
    }
    return cached = 0;
}

#define NEED_DEVICE()                  \
    do {                               \
        if (int rc_ = device_ok()) return rc_; \
    } while (0)

static int net_matches(const lbdrn_geom* g, const lbdrn_net* net)
{
    LBDRN_REQUIRE(net->F == feature_dim(*g), "net.F=%d but the geometry yields %d features", net->F,
                  feature_dim(*g));
    LBDRN_REQUIRE(net->C == g->C, "net.C=%d but the image has %d bands", net->C, g->C);
    LBDRN_REQUIRE(g->msb_max >= 0 && g->msb_max <= 65535, "msb_max=%d out of range", g->msb_max);
    return 0;
}

extern "C" {

const char* lbdrn_last_error(void) { return last_error(); }
int lbdrn_abi_version(void) { return LBDRN_ABI_VERSION; }
int lbdrn_device_check(void) { return device_ok(); }

int64_t lbdrn_param_count(const lbdrn_net* net)
{
    if (check_net(net)) return LBDRN_E_ARG;
    return param_count(*net);
}

int32_t lbdrn_feature_dim(const lbdrn_geom* g)
{
    if (!g) return LBDRN_E_ARG;
    return feature_dim(*g);
}

int lbdrn_split_bits(const uint16_t* img, int32_t C, int32_t H, int32_t W, int32_t K, uint16_t* msb,
                     int32_t* msb_max, void* stream)
{
    LBDRN_REQUIRE(img && msb_max, "img and msb_max must not be null");
    LBDRN_REQUIRE(C >= 1 && H >= 1 && W >= 1 && K >= 1 && K <= 15, "bad arguments C=%d H=%d W=%d K=%d", C, H, W, K);
    NEED_DEVICE();
    return generic_split_bits(img, C, H, W, K, msb, msb_max, (hipStream_t)stream);
}

int lbdrn_labels(const uint16_t* img, int32_t C, int32_t H, int32_t W, int32_t K, const int64_t* idx,
                 int64_t n, float* labels, void* stream)
{
    LBDRN_REQUIRE(img && labels, "img and labels must not be null");
    LBDRN_REQUIRE(C >= 1 && H >= 1 && W >= 1 && K >= 1 && K <= 15 && n >= 0, "bad arguments");
    LBDRN_REQUIRE(idx || n <= (int64_t)H * W, "n exceeds the pixel count");
    NEED_DEVICE();
    return generic_labels(img, C, H, W, K, idx, n, labels, (hipStream_t)stream);
}

int lbdrn_features(const lbdrn_geom* g, const uint16_t* msb, const int64_t* idx, int64_t n,
                   float* features, void* stream)
{
    if (int rc = check_geom(g)) return rc;
    LBDRN_REQUIRE(msb && features && n >= 0, "msb/features null or n negative");
    LBDRN_REQUIRE(idx || n <= (int64_t)g->H * g->W, "n exceeds the pixel count");
    NEED_DEVICE();
    return generic_features(*g, msb, idx, n, 0, features, (hipStream_t)stream);
}

size_t lbdrn_forward_workspace(const lbdrn_net* net, int64_t B)
{
    if (check_net(net) || B < 0) return 0;
    return generic_forward_workspace(*net, B);
}

int lbdrn_forward(const lbdrn_net* net, const float* params, const float* x, int64_t B, float* y,
                  void* workspace, size_t workspace_bytes, void* stream)
{
    if (int rc = check_net(net)) return rc;
    LBDRN_REQUIRE(params && x && y && B >= 0, "null pointer or negative batch");
    NEED_DEVICE();
    return generic_forward(*net, params, x, B, y, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t lbdrn_apply_workspace(const lbdrn_geom* g, const lbdrn_net* net)
{
    if (check_geom(g) || check_net(net)) return 0;
    size_t a = generic_apply_workspace(*g, *net);
    size_t b = mfma_apply_supported(*g, *net) ? mfma_apply_workspace(*g, *net) : 0;
    return a > b ? a : b;
}

static int pick_apply(const lbdrn_geom* g, const lbdrn_net* net, int32_t path, bool* use_mfma)
{
    const bool ok = mfma_apply_supported(*g, *net);
    if (path == LBDRN_PATH_MFMA && !ok) {
        set_error("fused MFMA apply kernel does not support bc=%d nl=%d C=%d D=%d", net->bc, net->nl,
                  net->C, g->D);
        return LBDRN_E_UNSUPPORTED;
    }
    LBDRN_REQUIRE(path >= LBDRN_PATH_AUTO && path <= LBDRN_PATH_MFMA, "unknown path %d", path);
    *use_mfma = ok && path != LBDRN_PATH_GENERIC;
    return 0;
}

int lbdrn_decode_fused(const lbdrn_geom* g, const lbdrn_net* net, const uint16_t* msb,
                       const float* params, uint16_t* out, float* y_out, void* workspace,
                       size_t workspace_bytes, int32_t path, void* stream)
{
    if (int rc = check_geom(g)) return rc;
    if (int rc = check_net(net)) return rc;
    if (int rc = net_matches(g, net)) return rc;
    LBDRN_REQUIRE(msb && params && out, "msb, params and out must not be null");
    NEED_DEVICE();
    bool use_mfma = false;
    if (int rc = pick_apply(g, net, path, &use_mfma)) return rc;
    if (use_mfma)
        return mfma_decode(*g, *net, msb, params, out, y_out, workspace, workspace_bytes, (hipStream_t)stream);
    return generic_decode(*g, *net, msb, params, out, y_out, workspace, workspace_bytes, (hipStream_t)stream);
}

int lbdrn_eval_sse(const lbdrn_geom* g, const lbdrn_net* net, const uint16_t* img, const uint16_t* msb,
                   const float* params, double* sse, void* workspace, size_t workspace_bytes,
                   int32_t path, void* stream)
{
    if (int rc = check_geom(g)) return rc;
    if (int rc = check_net(net)) return rc;
    if (int rc = net_matches(g, net)) return rc;
    LBDRN_REQUIRE(img && msb && params && sse, "img, msb, params and sse must not be null");
    NEED_DEVICE();
    const bool background = (path & LBDRN_EVAL_BACKGROUND) != 0, fast = (path & LBDRN_EVAL_FAST) != 0;
    const bool x16 = fast && (path & LBDRN_EVAL_X16) != 0;
    path &= ~(LBDRN_EVAL_BACKGROUND | LBDRN_EVAL_FAST | LBDRN_EVAL_X16);
    bool use_mfma = false;
    if (int rc = pick_apply(g, net, path, &use_mfma)) return rc;
    if (use_mfma)
        return mfma_eval_sse(*g, *net, img, msb, params, sse, workspace, workspace_bytes, background, fast, x16,
                             (hipStream_t)stream);
    return generic_eval_sse(*g, *net, img, msb, params, sse, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t lbdrn_train_workspace(const lbdrn_geom* g, const lbdrn_net* net, int32_t batch_size)
{
    if (check_geom(g) || check_net(net) || batch_size < 1) return 0;
    size_t a = generic_train_workspace(*net, batch_size);
    size_t b = mfma_train_supported(*g, *net, batch_size) ? mfma_train_workspace(*g, *net, batch_size) : 0;
    return a > b ? a : b;
}

int lbdrn_train_prepare(const lbdrn_geom* g, const lbdrn_net* net, const uint16_t* img,
                        const uint16_t* msb, int32_t batch_size, void* workspace, size_t workspace_bytes,
                        int32_t path, void* stream)
{
    if (int rc = check_geom(g)) return rc;
    if (int rc = check_net(net)) return rc;
    if (int rc = net_matches(g, net)) return rc;
    LBDRN_REQUIRE(img && msb && batch_size >= 1, "null pointer or bad batch size");
    LBDRN_REQUIRE(path >= LBDRN_PATH_AUTO && path <= LBDRN_PATH_MFMA, "unknown path %d", path);
    NEED_DEVICE();
    const bool ok = mfma_train_supported(*g, *net, batch_size);
    if (path == LBDRN_PATH_MFMA && !ok) {
        set_error("fused MFMA train kernel does not support this shape or minibatch size");
        return LBDRN_E_UNSUPPORTED;
    }
    if (ok && path != LBDRN_PATH_GENERIC)
        return mfma_train_prepare(*g, *net, img, msb, batch_size, workspace, workspace_bytes, (hipStream_t)stream);
    return 0;
}

int lbdrn_train_epoch(const lbdrn_geom* g, const lbdrn_net* net, const uint16_t* img,
                      const uint16_t* msb, const int64_t* perm, int64_t n, int32_t batch_size,
                      float* params, float* exp_avg, float* exp_avg_sq, int64_t adam_step0, double lr,
                      float* losses, void* workspace, size_t workspace_bytes, int32_t path,
                      void* stream)
{
    if (int rc = check_geom(g)) return rc;
    if (int rc = check_net(net)) return rc;
    if (int rc = net_matches(g, net)) return rc;
    LBDRN_REQUIRE(img && msb && perm && params && exp_avg && exp_avg_sq, "null pointer");
    LBDRN_REQUIRE(n >= 0 && batch_size >= 1 && adam_step0 >= 0, "bad n/batch_size/adam_step0");
    NEED_DEVICE();
    const bool alone = (path & LBDRN_TRAIN_ALONE) != 0;   // a hint (lbdrn_hip.h): performance only
    path &= ~LBDRN_TRAIN_ALONE;
    const bool ok = mfma_train_supported(*g, *net, batch_size);
    if (path == LBDRN_PATH_MFMA && !ok) {
        set_error("fused MFMA train kernel does not support this shape or minibatch size");
        return LBDRN_E_UNSUPPORTED;
    }
    LBDRN_REQUIRE(path >= LBDRN_PATH_AUTO && path <= LBDRN_PATH_MFMA, "unknown path %d", path);
    if (ok && path != LBDRN_PATH_GENERIC)
        return mfma_train_epoch(*g, *net, img, msb, perm, n, batch_size, params, exp_avg, exp_avg_sq,
                                adam_step0, lr, losses, workspace, workspace_bytes, (hipStream_t)stream, alone);
    return generic_train_epoch(*g, *net, img, msb, perm, n, batch_size, params, exp_avg, exp_avg_sq,
                               adam_step0, lr, losses, workspace, workspace_bytes, (hipStream_t)stream);
}

int lbdrn_train_group_max(void) { return 4; }

int32_t lbdrn_train_group_size(const lbdrn_geom* g, const lbdrn_net* net)
{
    if (!g || !net) return 1;
    return mfma_train_takes_groups(*g, *net) ? lbdrn_train_group_max() : 1;
}

int32_t lbdrn_train_step_features(const lbdrn_geom* g, const lbdrn_net* net)
{
    if (!g || !net) return 0;
    return mfma_train_step_features(*g, *net);
}

int lbdrn_train_epoch_group(int32_t count, const lbdrn_geom* const* g, const lbdrn_net* net, const uint16_t* const* img,
                            const uint16_t* const* msb, const int64_t* const* perm, int64_t n, int32_t batch_size,
                            float* const* params, float* const* exp_avg, float* const* exp_avg_sq, int64_t adam_step0,
                            double lr, float* const* losses, void* const* workspace, size_t workspace_bytes, int32_t path,
                            void* stream)
{
    LBDRN_REQUIRE(count >= 1 && count <= lbdrn_train_group_max(), "group of %d fits (1..%d)", count, lbdrn_train_group_max());
    LBDRN_REQUIRE(g && img && msb && perm && params && exp_avg && exp_avg_sq && workspace, "null pointer array");
    if (int rc = check_net(net)) return rc;
    for (int f = 0; f < count; ++f) {
        LBDRN_REQUIRE(g[f] && img[f] && msb[f] && perm[f] && params[f] && exp_avg[f] && exp_avg_sq[f], "null pointer (fit %d)", f);
        if (int rc = check_geom(g[f])) return rc;
        if (int rc = net_matches(g[f], net)) return rc;
        LBDRN_REQUIRE(g[f]->C == g[0]->C && g[f]->H == g[0]->H && g[f]->W == g[0]->W && g[f]->K == g[0]->K &&
                      g[f]->D == g[0]->D && g[f]->P == g[0]->P && g[f]->use_colors == g[0]->use_colors &&
                      g[f]->relative == g[0]->relative, "the fits of a group must have one shape (fit %d differs)", f);
    }
    LBDRN_REQUIRE(n >= 0 && batch_size >= 1 && adam_step0 >= 0, "bad n/batch_size/adam_step0");
    const int32_t hint = count == 1 ? (path & LBDRN_TRAIN_ALONE) : 0;   // (a group is not alone)
    path &= ~LBDRN_TRAIN_ALONE;
    LBDRN_REQUIRE(path >= LBDRN_PATH_AUTO && path <= LBDRN_PATH_MFMA, "unknown path %d", path);
    NEED_DEVICE();
    // side by side on the fused step where the shape has one; otherwise (and for any shape the group launch does not
    // take) one after another: same numbers either way
    if (count > 1 && path != LBDRN_PATH_GENERIC && mfma_train_supported(*g[0], *net, batch_size)) {
        const int rc = mfma_train_epoch_group(count, *g[0], *net, perm, n, batch_size, params, exp_avg, exp_avg_sq,
                                              adam_step0, lr, losses, workspace, workspace_bytes, (hipStream_t)stream, false);
        if (rc != LBDRN_E_UNSUPPORTED) return rc;
    }
    for (int f = 0; f < count; ++f)
        if (int rc = lbdrn_train_epoch(g[f], net, img[f], msb[f], perm[f], n, batch_size, params[f], exp_avg[f],
                                       exp_avg_sq[f], adam_step0, lr, losses ? losses[f] : nullptr, workspace[f],
                                       workspace_bytes, path | hint, stream))
            return rc;
    return 0;
}

int lbdrn_train_profile_mode(int32_t mode)
{
    LBDRN_REQUIRE(mode >= 0 && mode <= 5, "mode must be 0 .. 5");
    return train_profile_mode(mode);
}

size_t lbdrn_randperm_workspace(int64_t n, int32_t count)
{
    if (n < 0 || count < 1 || count > 32) return 0;
    return randperm_workspace(n, count);
}

int lbdrn_randperm(const uint64_t* seeds, int32_t count, int64_t n, int64_t* perm, void* workspace,
                   size_t workspace_bytes, void* stream)
{
    LBDRN_REQUIRE(n >= 0 && (perm || n == 0), "bad n or null output");
    NEED_DEVICE();
    return randperm_batch(seeds, count, n, perm, workspace, workspace_bytes, (hipStream_t)stream);
}

int64_t lbdrn_mt19937_jump_poly(int32_t segment, uint32_t* poly624)
{
    LBDRN_REQUIRE(poly624, "null output");
    return mt19937_jump_poly(segment, poly624);
}

size_t lbdrn_plane_bound(int32_t C, int32_t H, int32_t W) { return plane_bound(C, H, W); }
size_t lbdrn_plane_workspace(int32_t C, int32_t H, int32_t W) { return plane_workspace(C, H, W); }

int lbdrn_plane_encode(const uint16_t* planes, int32_t C, int32_t H, int32_t W, void* body, size_t body_capacity,
                       uint64_t* body_bytes, void* workspace, size_t workspace_bytes, void* stream)
{
    LBDRN_REQUIRE(C >= 1 && H >= 1 && W >= 1 && C <= 65535, "bad raster geometry");
    NEED_DEVICE();
    return plane_encode(planes, C, H, W, body, body_capacity, body_bytes, workspace, workspace_bytes,
                        (hipStream_t)stream);
}

int lbdrn_plane_decode(const void* body, size_t body_bytes, int32_t C, int32_t H, int32_t W, uint16_t* planes,
                       int32_t* status, void* workspace, size_t workspace_bytes, void* stream)
{
    LBDRN_REQUIRE(C >= 1 && H >= 1 && W >= 1 && C <= 65535, "bad raster geometry");
    NEED_DEVICE();
    return plane_decode(body, body_bytes, C, H, W, planes, status, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t lbdrn_weights_bound(int64_t n) { return n < 0 ? 0 : weights_bound(n); }

int lbdrn_weights_encode(const float* values, int64_t n, int32_t precision, void* out, size_t capacity, size_t* nbytes)
{
    LBDRN_REQUIRE((values || n == 0) && out && nbytes && n >= 0 && n < ((int64_t)1 << 32), "null pointer or bad count");
    LBDRN_REQUIRE(precision == 0 || (precision >= 2 && precision <= 32), "precision %d is not 0 or 2..32", precision);
    return weights_encode(values, n, precision, (uint8_t*)out, capacity, nbytes);
}

int lbdrn_weights_info(const void* stream, size_t nbytes, int64_t* n, int32_t* precision)
{
    LBDRN_REQUIRE(stream && n && precision, "null pointer");
    int prec = 0;
    const int rc = weights_info((const uint8_t*)stream, nbytes, n, &prec);
    *precision = prec;
    return rc;
}

int lbdrn_weights_decode(const void* stream, size_t nbytes, float* values, int64_t capacity)
{
    LBDRN_REQUIRE(stream && (values || capacity == 0) && capacity >= 0, "null pointer or bad capacity");
    return weights_decode((const uint8_t*)stream, nbytes, values, capacity);
}

int lbdrn_train_step(const lbdrn_net* net, const float* x, const float* t, int32_t B, float* params,
                     float* exp_avg, float* exp_avg_sq, int64_t adam_step, double lr,
                     int32_t apply_adam, float* loss, float* grads, void* workspace,
                     size_t workspace_bytes, void* stream)
{
    if (int rc = check_net(net)) return rc;
    LBDRN_REQUIRE(x && t && params, "x, t and params must not be null");
    LBDRN_REQUIRE(!apply_adam || (exp_avg && exp_avg_sq && adam_step >= 1), "Adam state missing or adam_step < 1");
    NEED_DEVICE();
    return generic_train_step(*net, x, t, B, params, exp_avg, exp_avg_sq, adam_step, lr, apply_adam,
                              loss, grads, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
