/*
 * lbdrn_hip.h -- C ABI of liblbdrn_hip.so, the MI355X (gfx950) implementation of the
 * LBDRN-MSIC per-image fit/apply hot path.
 *
 * The reference (/root/reference, pure Python) has no FFI of its own; its boundary is the
 * Python surface encode.py / decode.py / LBDRNmodel.py / LBDRNdataset.py.  Each entry point
 * below names the reference lines whose arithmetic it replaces; the Python host code in
 * lbdrn-msic_amd/ binds them with ctypes (lbdrn_hip/_lib.py) and INTEGRATION.md shows the stub
 * a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C, no exceptions: every call returns 0 on success or a negative lbdrn_status;
 *    lbdrn_last_error() returns a per-thread message for the last failure.
 *  - all data pointers are caller-owned DEVICE pointers (HBM); the library never frees or
 *    retains them.  Scalars and the two config structs are passed by value / host pointer.
 *    (One exception, named where it is declared: the weight payload codec works on host buffers.)
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream) and re-entrant; there is no global mutable state besides the error string.
 *  - there is no CPU path: without a gfx950 device every compute call returns LBDRN_E_DEVICE.
 *  - images are channel-planar [C][H][W] uint16, as GDAL's ReadAsArray() returns them
 *    (ref LBDRNdataset.py:93-94); feature / label matrices are row-major [N][F] / [N][C]
 *    float32, row n = pixel y*W+x (ref LBDRNdataset.py:129-131).
 */
#ifndef LBDRN_HIP_H
#define LBDRN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the functions declared here are its only exports. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* 2 (round 5): the `path` word of lbdrn_train_epoch gained LBDRN_TRAIN_ALONE and of lbdrn_eval_sse LBDRN_EVAL_FAST /
 * LBDRN_EVAL_BACKGROUND, lbdrn_train_group_size / lbdrn_train_step_features were added, and the fused step's
 * weight-gradient summation tree changed (still one fixed tree: see lbdrn_train_epoch); struct lbdrn_net gained `act`
 * (20 bytes instead of 16: a caller built against version 1 passes a shorter struct and must be rebuilt). */
#define LBDRN_ABI_VERSION 2

typedef enum lbdrn_status {
    LBDRN_OK = 0,
    LBDRN_E_ARG = -1,     /* invalid argument (shape, range, null pointer) */
    LBDRN_E_DEVICE = -2,  /* no usable gfx950 device / HIP runtime error */
    LBDRN_E_UNSUPPORTED = -3,
    LBDRN_E_WORKSPACE = -4 /* workspace too small */
} lbdrn_status;

/* Image + feature geometry: constants.py:3-14 plus the per-image numbers process() derives. */
typedef struct lbdrn_geom {
    int32_t C, H, W;      /* bands, rows, columns */
    int32_t K;            /* low bits dropped (encode.py:178) */
    int32_t D;            /* neighbourhood radius, window (2D+1)^2 (encode.py:184) */
    int32_t msb_max;      /* MSB.max(): LBDRNdataset.py:120, decode.py:93 */
    int32_t use_colors;   /* constants.py:11 */
    int32_t relative;     /* constants.py:14 */
    int32_t P;            /* positional features per axis: 0, 1 or 1+2*N_FREQ (constants.py:3-8) */
    int32_t reserved;
    const float *rowtab;  /* device [H][P]: ph-derived features (LBDRNdataset.py:108-118) */
    const float *coltab;  /* device [W][P]: pw-derived features */
} lbdrn_geom;

/* LBDRNModel(dim_in=F, dim_hidden=bc, dim_out=C, num_layers=nl): LBDRNmodel.py:58-77.
 * Parameters travel as one float32 vector in state_dict order (encode.py:124-128):
 * for l in 0..nl-1: W_l[bc][in_l], b_l[bc];  then W_last[C][bc], b_last[C]. */
typedef struct lbdrn_net {
    int32_t F, bc, C, nl;
    int32_t act;          /* hidden activation (LBDRNmodel.py:37,75): LBDRN_ACT_SINE = the default Sine(w0 = 30);
                           * LBDRN_ACT_RELU = LBDRNModel(activation=torch.nn.ReLU()), the alternative the reference names
                           * (encode.py:75, decode.py:108).  Since round 6 the fused kernels take it as a template argument
                           * wherever they take the Sine network at bc <= 128 (apply: bc 32 / 64 / 128; training: bc = 64, one
                           * or two hidden layers); the bc >= 128 training / bc = 256 apply kernels and the nl = 3 training
                           * kernel are the Sine network's: there LBDRN_PATH_MFMA answers LBDRN_E_UNSUPPORTED and
                           * LBDRN_PATH_AUTO takes the LDS-tiled generic kernels.  The head is always Sigmoid. */
} lbdrn_net;
enum { LBDRN_ACT_SINE = 0, LBDRN_ACT_RELU = 1 };

const char *lbdrn_last_error(void);
int lbdrn_abi_version(void);
/* 0 if a gfx950 device is present and usable, else LBDRN_E_DEVICE. */
int lbdrn_device_check(void);
int64_t lbdrn_param_count(const lbdrn_net *net);
int32_t lbdrn_feature_dim(const lbdrn_geom *g);

/* a1 -- LBDRNdataset.py:95-101: msb = img >> K (uint16 plane, same layout) and the running
 * maximum of msb into *msb_max (device int32, must be zeroed by the caller or hold a previous
 * maximum).  msb may be NULL (max only). */
int lbdrn_split_bits(const uint16_t *img, int32_t C, int32_t H, int32_t W, int32_t K,
                     uint16_t *msb, int32_t *msb_max, void *stream);

/* a1 -- LBDRNdataset.py:96-97,131: labels[i][c] = float(img - (msb<<K)) / (2^K-1) for the pixels
 * idx[0..n) (int64 device indices) or, when idx is NULL, for pixels 0..n-1 in raster order. */
int lbdrn_labels(const uint16_t *img, int32_t C, int32_t H, int32_t W, int32_t K,
                 const int64_t *idx, int64_t n, float *labels, void *stream);

/* a2/a3 -- LBDRNdataset.py:104-130 (dup. decode.py:77-102): feature rows for idx[0..n) or raster
 * order.  msb is the [C][H][W] uint16 MSB plane. */
int lbdrn_features(const lbdrn_geom *g, const uint16_t *msb, const int64_t *idx, int64_t n,
                   float *features, void *stream);

/* a5 -- LBDRNModel.forward, LBDRNmodel.py:79-82: y[B][C] = sigmoid(W_last sin(30(...)) + b).
 * workspace: device scratch of lbdrn_forward_workspace(net, B) bytes. */
size_t lbdrn_forward_workspace(const lbdrn_net *net, int64_t B);
int lbdrn_forward(const lbdrn_net *net, const float *params, const float *x, int64_t B, float *y,
                  void *workspace, size_t workspace_bytes, void *stream);

/* a2+a5+a11 -- decode.py:73-134 in one pass: features from the MSB plane, forward,
 * r = round_half_even(y*(2^K-1)), out = (msb<<K) + r as uint16 [C][H][W].
 * y_out (optional, may be NULL) receives the sigmoid outputs [H*W][C].
 * flags: LBDRN_PATH_AUTO picks the MFMA kernel when the shape supports it. */
#define LBDRN_PATH_AUTO 0
#define LBDRN_PATH_GENERIC 1 /* tiled f32 FMA kernels, any shape */
#define LBDRN_PATH_MFMA 2    /* fused MFMA kernel; LBDRN_E_UNSUPPORTED if the shape does not fit */
size_t lbdrn_apply_workspace(const lbdrn_geom *g, const lbdrn_net *net);
int lbdrn_decode_fused(const lbdrn_geom *g, const lbdrn_net *net, const uint16_t *msb,
                       const float *params, uint16_t *out, float *y_out, void *workspace,
                       size_t workspace_bytes, int32_t path, void *stream);

/* a9 -- evaluator + LBDRNPerformance (modified_ignite_engine.py:38-43, LBDRNperformance.py:18-21):
 * *sse (device float64) = sum over all pixels and bands of (y - label)^2, fixed summation order
 * (bitwise reproducible, on any launch shape).  img is the original [C][H][W] image the labels derive from.
 * path may carry LBDRN_EVAL_BACKGROUND (path | LBDRN_EVAL_BACKGROUND): the pass is launched on half as many
 * workgroups, for running on a second stream beside a fit's training steps (which hold the other half of the CUs).
 * Same *sse bit for bit. */
#define LBDRN_EVAL_BACKGROUND 0x200
/* path may also carry LBDRN_EVAL_FAST: the pass is the reference's PER-EPOCH evaluation (encode.py:104-117), whose
 * result only ranks the epochs -- an encode-time float, held to the 1e-5 relative tolerance like the training loss, not
 * to a bit pattern.  With the flag the fused kernels take sin and the sigmoid from the hardware's transcendental
 * instructions behind a compensated reduction (4.5e-7 absolute per activation, the training step's arithmetic) instead
 * of the canonical polynomials the decode kernels must use: *sse within 1e-6 relative of the flagless call (tested),
 * the pass 20 % shorter.  Still a fixed summation order: bitwise reproducible, same sum on any launch shape.  The
 * generic path ignores the flag.
 * With the flag the fused kernels take the colour window from img >> g->K and DO NOT READ msb (one plane read per pass
 * instead of two): msb must be exactly that plane -- as lbdrn_split_bits leaves it -- or NULL.  A caller whose MSB plane is
 * something else (a decoded or modified base) must leave the flag out. */
#define LBDRN_EVAL_FAST 0x400
/* With LBDRN_EVAL_FAST, path may also carry LBDRN_EVAL_X16 (round 6; OPT-IN: no caller of this package sets it by default): the
 * colour features of layer 0 run on the f16 matrix pipe with EXACT operands.  A relative colour feature is
 * (msb_nbr - msb_ctr) / max -- its numerator an integer of at most 11 bits, exactly an fp16 -- and a float32 weight is exactly
 * the sum of three fp16 pieces once a power of two has brought it into range; v_mfma_f32_32x32x16_f16 multiplies them exactly
 * and sums in float32, so the pre-activations are those of the float32 MFMA with FEWER roundings (nothing of a weight or of a
 * feature is dropped), at 3/16 of its matrix-pipe cycles.  A hint: it takes effect where the shape and the image qualify
 * (relative colours, no positional features, bc <= 64, D in 1..3, an even band count, g->msb_max <= 2047: any 16-bit image at
 * K >= 5) and is ignored elsewhere.  *sse within 1e-6 relative of the flagless call like LBDRN_EVAL_FAST (measured: DESIGN.md 10). */
#define LBDRN_EVAL_X16 0x1000
int lbdrn_eval_sse(const lbdrn_geom *g, const lbdrn_net *net, const uint16_t *img,
                   const uint16_t *msb, const float *params, double *sse, void *workspace,
                   size_t workspace_bytes, int32_t path, void *stream);

/* a4+a5+a7+a8 -- one trainer epoch (encode.py:69-70,157; modified_ignite_engine.py:18-27;
 * torch.optim.Adam defaults, encode.py:84): for each minibatch perm[s*bs .. min((s+1)*bs,n))
 * of pixel indices: gather features and labels, forward, MSE loss, backward, Adam update of
 * params/exp_avg/exp_avg_sq in place.  adam_step0 = number of Adam steps already taken;
 * losses (optional) receives one float32 minibatch loss per step.  The last minibatch may be
 * short (no drop_last, encode.py:69) -- down to ONE row; no byte of perm beyond its n elements is read.
 * batch_size: any value >= 1 (the reference takes any -bs).  The fused bc = 64 step addresses one step's gradient slabs
 * with 32-bit offsets: where they would pass 2 GiB (about 0.9 million rows per minibatch at the headline shape)
 * LBDRN_PATH_MFMA answers LBDRN_E_UNSUPPORTED and LBDRN_PATH_AUTO runs the generic step (same tolerance contract). */
size_t lbdrn_train_workspace(const lbdrn_geom *g, const lbdrn_net *net, int32_t batch_size);
/* path of lbdrn_train_epoch may carry LBDRN_TRAIN_ALONE (path | LBDRN_TRAIN_ALONE): a HINT that nothing else of weight
 * runs on the device beside this fit's steps (the reference's own situation: one image after another, run.sh:29-42).
 * The fused bc = 64 step then also touches the head of the NEXT minibatch's rows before its loader wave ends -- a lone
 * chain of short launches leaves the memory system idle most of the time, and the next launch's first requests end in
 * a cache (-1..2 % per tile); with several fits in flight the same reads cost more than they return, so callers that
 * keep fits in flight leave the flag out.  Performance only: every number is the same bit for bit with and without. */
#define LBDRN_TRAIN_ALONE 0x800
/* Once per image, before the first lbdrn_train_epoch on this workspace: builds the per-image state
 * the fused path keeps in the workspace (the [N][F+C] feature|label row matrix that replaces the
 * reference's host-side LBDRNDataset tensors, LBDRNdataset.py:141-142).  The workspace contents
 * must then be left untouched between the epochs of that image.  No-op for the generic path. */
int lbdrn_train_prepare(const lbdrn_geom *g, const lbdrn_net *net, const uint16_t *img,
                        const uint16_t *msb, int32_t batch_size, void *workspace,
                        size_t workspace_bytes, int32_t path, void *stream);
int lbdrn_train_epoch(const lbdrn_geom *g, const lbdrn_net *net, const uint16_t *img,
                      const uint16_t *msb, const int64_t *perm, int64_t n, int32_t batch_size,
                      float *params, float *exp_avg, float *exp_avg_sq, int64_t adam_step0,
                      double lr, float *losses, void *workspace, size_t workspace_bytes,
                      int32_t path, void *stream);

/* The same epoch for `count` (1 .. lbdrn_train_group_max()) INDEPENDENT fits of one shape -- same raster dimensions,
 * K, D, feature switches, network, n, batch size, Adam step count and learning rate; every array holds one entry per
 * fit: its geometry (msb_max and tables may differ), image, MSB plane, permutation, optimiser state, losses (array or
 * entries may be NULL) and its own prepared workspace of workspace_bytes each.  Where the shape has a fused step that
 * takes groups, minibatch s of every fit runs in ONE launch (two 8192-row minibatches = 256 workgroups fill the
 * chip exactly; independent chains of 128-workgroup launches meet each other only by chance), otherwise the fits
 * run one after another.  The reference has no such call (run.sh:29-42 encodes image after image): results are those
 * of `count` calls of lbdrn_train_epoch, bit for bit. */
int lbdrn_train_group_max(void);
/* How many fits of this shape lbdrn_train_epoch_group steps in ONE launch per minibatch: lbdrn_train_group_max() where
 * the shape runs on the streamed step, 1 where the fits of a group run one after another (the caller then gains nothing
 * from grouping and should give every fit its own stream).  Only the scalar fields of g are read (tables may be NULL). */
int32_t lbdrn_train_group_size(const lbdrn_geom *g, const lbdrn_net *net);
/* Diagnostic (bench.py's FLOP count): the number of input features the fused training step of this shape multiplies.
 * With RELATIVE and D > 0 the window centre minus itself is an exact 0.0f for every band (LBDRNdataset.py:126-128):
 * the streamed step leaves those C columns of W_0 out of its products -- they add nothing, their gradient is exactly
 * 0 and Adam leaves such a weight at its initial value, as in the reference -- and this returns F - C; F otherwise. */
int32_t lbdrn_train_step_features(const lbdrn_geom *g, const lbdrn_net *net);
int lbdrn_train_epoch_group(int32_t count, const lbdrn_geom *const *g, const lbdrn_net *net,
                            const uint16_t *const *img, const uint16_t *const *msb, const int64_t *const *perm,
                            int64_t n, int32_t batch_size, float *const *params, float *const *exp_avg,
                            float *const *exp_avg_sq, int64_t adam_step0, double lr, float *const *losses,
                            void *const *workspace, size_t workspace_bytes, int32_t path, void *stream);

/* DIAGNOSTIC, not part of the codec path: a measurement aid of bench.py's roofline leg, and the one piece of mutable
 * state this library keeps besides the error string -- thread-local, off (0) unless a caller sets it.  mode 1: every step
 * of lbdrn_train_epoch on this thread launches its reduce/Adam kernel twice (the second with a zero step size); mode 2:
 * its training kernel twice (the launch is idempotent: it writes its gradient slabs and loss partials); mode 0: normal.
 * Timing one epoch in each mode with a single HIP-event pair gives, per step, t_reduce = t(mode 1) - t(mode 0) and
 * t_train = t(mode 2) - t(mode 0) -- what one more launch of that kernel costs inside the real dependent sequence,
 * its launch boundary included -- without per-launch event packets (which cost more than the gaps they would
 * measure).  mode 3: the reduce/Adam launch is left out -- an epoch of training launches back to back, every one on its
 * own slice of the permutation (cold rows, unlike the repeated launch of mode 2): t(mode 3) / steps is the training
 * kernel's own average duration, the figure bench.py's `roofline.achieved` is computed from.  Where a step has TWO
 * launches in front of its reduce/Adam launch (the bc >= 128 step: forward/backward, then the weight-gradient GEMM) mode 3
 * runs both, mode 4 the forward/backward launch alone and mode 5 doubles the weight-gradient launch; elsewhere mode 4 is
 * mode 3 and mode 5 is mode 0.  Use on scratch optimiser state. */
int lbdrn_train_profile_mode(int32_t mode);

/* a4 -- the minibatch order: perm[0..n) = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
 * of a CPU generator, bit for bit (the call RandomSampler.__iter__ makes for DataLoader(shuffle=True),
 * torch/utils/data/sampler.py:163-183, behind encode.py:69-70), computed on the GPU.  Supported for
 * n < 2^32/20 (torch's Fisher-Yates branch); larger n returns LBDRN_E_UNSUPPORTED. */
size_t lbdrn_randperm_workspace(int64_t n, int32_t count);
/* count permutations at once (1 <= count <= 32; seeds is a HOST array): perm[c][0..n) for seeds[c].
 * The MT19937 recurrence is serial, so a fit generates the orders of all its epochs in one call. */
int lbdrn_randperm(const uint64_t *seeds, int32_t count, int64_t n, int64_t *perm, void *workspace,
                   size_t workspace_bytes, void *stream);
/* Host only, for tests: the jump polynomial of segment `segment` (1..31) of a long permutation's MT19937 stream --
 * g(t) = t^(segment * words_per_segment) mod the generator's characteristic polynomial, as 624 words (coefficient of t^i
 * = bit i % 32 of word i / 32).  A permutation longer than a segment is generated as segments side by side, each from
 * the state x[J + k] = XOR over the set bits i of g of x[i + k] (x: the raw MT19937 word sequence, seeded state first).
 * Returns words_per_segment, or a negative LBDRN_E_* code. */
int64_t lbdrn_mt19937_jump_poly(int32_t segment, uint32_t *poly624);

/* MSB-plane payload "LBB2" -- stands where the reference calls an external lossless codec for the MSB raster
 * (gdal_translate -of JP2OpenJPEG QUALITY=100 REVERSIBLE=YES, encode.py:137; read back at decode.py:69-73).
 * The byte format is this library's own (oracle/plane_codec.c states it in full): lossless by construction,
 * not JPEG 2000.  planes: [C][H][W] uint16 in HBM.  body: counts[C*ceil(W/64)] uint32 then the 32-bit words.
 *   lbdrn_plane_bound      worst-case body size in bytes (size the output buffer with it)
 *   lbdrn_plane_workspace  scratch bytes for either direction
 *   lbdrn_plane_encode     writes the body and its length in bytes (*body_bytes, device memory)
 *   lbdrn_plane_decode     rebuilds the planes; *status (device int32) is 0 for a well-formed body, non-zero
 *                          when the stream was inconsistent (the planes are then unspecified, never out of
 *                          bounds).  LBDRN_E_ARG when body_bytes cannot belong to this geometry. */
size_t lbdrn_plane_bound(int32_t C, int32_t H, int32_t W);
size_t lbdrn_plane_workspace(int32_t C, int32_t H, int32_t W);
int lbdrn_plane_encode(const uint16_t *planes, int32_t C, int32_t H, int32_t W, void *body, size_t body_capacity,
                       uint64_t *body_bytes, void *workspace, size_t workspace_bytes, void *stream);
int lbdrn_plane_decode(const void *body, size_t body_bytes, int32_t C, int32_t H, int32_t W, uint16_t *planes,
                       int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/* Weight payload -- stands where the reference calls fpzip (encode.py:129 `fpzip.compress(params,
 * precision=args.precision, order='C')`, decode.py:113 `fpzip.decompress(...)[0][0][0]`): the float32 parameter
 * vector (state_dict order, a10) as a 1-D fpzip stream at `precision` bits (2..32; 0 = 32).  The lossy value map is
 * "keep the top `precision` bits of the IEEE pattern".  fpzip's source is absent here: the stream syntax is a
 * restatement of the published algorithm, PARITY UNPINNED (csrc/weights_codec.hip says what is and is not verified).
 * A 35 KB serial entropy code runs on the host: these three calls take HOST pointers (the only ones in this header).
 *   lbdrn_weights_bound   buffer size that always suffices for n values
 *   lbdrn_weights_encode  writes the stream, *nbytes = its length
 *   lbdrn_weights_info    reads the header: *n values at *precision bits
 *   lbdrn_weights_decode  values[0..n) (capacity in values); LBDRN_E_ARG for a foreign or truncated stream */
size_t lbdrn_weights_bound(int64_t n);
int lbdrn_weights_encode(const float *values, int64_t n, int32_t precision, void *out, size_t capacity,
                         size_t *nbytes);
int lbdrn_weights_info(const void *stream, size_t nbytes, int64_t *n, int32_t *precision);
int lbdrn_weights_decode(const void *stream, size_t nbytes, float *values, int64_t capacity);

/* a7/a8 building block exposed for teacher-forced parity tests: one update on an explicit
 * minibatch x[B][F], t[B][C]; grads (optional) receives d(loss)/d(params). */
int lbdrn_train_step(const lbdrn_net *net, const float *x, const float *t, int32_t B,
                     float *params, float *exp_avg, float *exp_avg_sq, int64_t adam_step,
                     double lr, int32_t apply_adam, float *loss, float *grads, void *workspace,
                     size_t workspace_bytes, void *stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* LBDRN_HIP_H */
