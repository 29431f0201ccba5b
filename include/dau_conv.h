/*
 * dau_conv.h -- C ABI of the MI355X-native DAU convolution operator.
 *
 * This is the drop-in boundary for the reference's DAU forward/backward path.  Every
 * entry point takes plain device pointers, sizes and a HIP stream handle; there are no
 * framework types in the signatures.  What each entry point replaces in
 * skokec/DAU-ConvNet (paths relative to the reference tree):
 *
 *   dau_conv_plan_create / _destroy   BaseDAUConvLayer::LayerSetUp + Reshape
 *                                     (src/dau_conv/base_dau_conv_layer.cpp:29-187,190-394),
 *                                     DAUConvForward / DAUConvBackward constructors
 *                                     (src/dau_conv/dau_conv_impl/dau_conv_forward.cpp:28-77,
 *                                      dau_conv_backward.cpp:11-89), DAUConvSettings
 *                                     (include/dau_conv/base_dau_conv_layer.hpp:109-130)
 *   dau_conv_workspace_bytes          DAUConvForward::get_allocation_sizes
 *                                     (include/dau_conv/dau_conv_impl/dau_conv_forward.hpp:30-33),
 *                                     DAUConvBackward::get_allocation_sizes (dau_conv_backward.hpp:41-42),
 *                                     allocate_workspace_mem (base_dau_conv_layer.hpp:263)
 *   dau_conv_forward                  DAUConvOp::Compute (plugins/tensorflow/src/dau_conv_op.cpp:150-324)
 *                                     -> BaseDAUConvLayer::Forward_gpu (src/dau_conv/base_dau_conv_layer.cu:15-127)
 *                                     -> DAUConvForward::forward_pass (dau_conv_forward.cpp:128-174)
 *   dau_conv_backward                 DAUConvGradOp::Compute (plugins/tensorflow/src/dau_conv_grad_op.cpp:115-318)
 *                                     -> BaseDAUConvLayer::Backward_gpu (base_dau_conv_layer.cu:130-363)
 *                                     -> DAUConvBackward::backward_pass (dau_conv_backward.cpp:173-232)
 *   dau_conv_backward_param_sums /    the two halves of Backward_gpu's parameter-gradient path: the raw sums of K4
 *   dau_conv_finalize_param_grads     (base_dau_conv_layer.cu:232-241) and its elementwise tail (dmu *= w, ignored units,
 *                                     NaN -> 0, :335-355; lr factor, dau_conv_grad_op.cpp:297-303), split so that a
 *                                     data-parallel caller can all-reduce the sums in between
 *   dau_conv_check_status /           the max|mu| / NaN precondition checks of the ops
 *   dau_conv_last_status              (dau_conv_op.cpp:223-262, dau_conv_grad_op.cpp:209-250); done on the
 *                                     device without a host sync, read back only on request (check_status waits for
 *                                     the stream; last_status reads what completed calls left in pinned host
 *                                     memory, without waiting; a bad status stays there until it has been reported)
 *   dau_conv_filters                  BaseDAUKernelCompute::get_kernels (base_dau_conv_layer.cu:537-710)
 *   dau_conv_unit_table               perpare_weights_and_offsets (dau_conv_forward_core.hpp:1858-2215)
 *   dau_conv_last_error               DAUException::what (include/dau_conv/util/common.hpp:40-66)
 *
 * Tensor layouts (identical to the reference): activations NCHW contiguous float32 (or,
 * with DAU_FLAG_IO_BF16, bfloat16 storage of x, y, dy, dx -- arithmetic stays fp32);
 * parameters and their gradients [1,S,G,F] contiguous float32 (f fastest); sigma is a
 * full [1,S,G,F] tensor whose element 0 is used (base_dau_conv_layer.hpp:266-275).
 *
 * Ownership: the caller owns every buffer, including one workspace per in-flight call.
 * A plan's configuration is immutable after creation, so concurrent calls on different
 * streams with different workspaces are safe.  Only dau_conv_plan_create / _destroy allocate
 * (the plan and 32 bytes of pinned host memory for its status mirror) and only
 * dau_conv_check_status synchronises (it waits for the stream); no other call allocates,
 * frees or synchronises.  A plan may be used on any device (the kernels' launch attributes are
 * set once per device, on the first call there, behind a lock).
 * dau_conv_backward OVERWRITES the gradient outputs (the reference op zero-fills them and
 * then accumulates, dau_conv_grad_op.cpp:202-205 -- same net result).
 *
 * Offset buckets.  The kernels stage a border of R pixels around every tile, R in
 * {4, 8, 16, 18, 20, 24, 32}.  The reference picks R per call from a blocking amax of mu1/mu2
 * (dau_conv_op.cpp:223-253).  Here a plan holds the kernel sets of every R up to the one
 * max_kernel_size allows; a call enqueues the set that covered the previous call's max|mu|
 * (read from the pinned mirror, no sync) plus the largest set, each guarded on the device by
 * THIS call's max|mu|, so exactly one does the work and results never depend on the hint (with
 * DAU_FLAG_DENSE_BF16 the bucket-4 member, the only one with bf16 arithmetic, is enqueued as a guarded
 * candidate on EVERY call, so that too is decided by the call's own offsets alone).
 *
 * Workspace.  Every pass stages its input before it gathers.  Where the staged copy of the
 * whole batch would exceed 12 GB (DAU_WORKSPACE_BUDGET_GB in the environment at plan
 * creation; only 512 x 512 maps get there) the pass runs over the batch in slabs of an even
 * divisor of N images, and dau_conv_workspace_bytes reports the smaller requirement.
 *
 * HIP graphs.  After one ordinary call (which sets the kernels' launch attributes) forward and
 * backward only enqueue kernels and one 16-byte memset on the caller's stream, so they can be
 * captured into a graph; the bucket candidates are frozen at capture time, the device-side
 * guards still pick the kernels the replay's offsets need.
 */
#ifndef DAU_CONV_H_
#define DAU_CONV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAU_CONV_ABI_VERSION 4

#if defined(__GNUC__)
#define DAU_API __attribute__((visibility("default")))
#else
#define DAU_API
#endif

/* Status codes; they map 1:1 onto the TensorFlow codes the reference ops raise. */
enum {
    DAU_OK = 0,
    DAU_INVALID_ARGUMENT = 1,    /* bad shape/attr, |mu| larger than the offset bucket allows   */
    DAU_FAILED_PRECONDITION = 2, /* NaN in mu1/mu2, sigma <= 0                                   */
    DAU_INTERNAL = 3             /* HIP runtime error (reference: DAUException -> INTERNAL)      */
};

/* dau_conv_desc.flags */
enum {
    DAU_FLAG_USE_INTERPOLATION = 1 << 0,    /* attr use_interpolation (default true)              */
    DAU_FLAG_UNIT_TESTING = 1 << 1,         /* attr unit_testing: drop last error row/col per the
                                               oracle rule (dau_conv_test.py:110-136)             */
    DAU_FLAG_SINGLE_DIM_KERNEL = 1 << 2,    /* attr single_dim_kernel (DAUConv1d)                 */
    DAU_FLAG_FORBID_POSITIVE_DIM1 = 1 << 3, /* attr forbid_positive_dim1                          */
    DAU_FLAG_IO_BF16 = 1 << 4,              /* x, y, dy, dx are bfloat16 arrays (passed through the
                                               float* parameters); parameters, their gradients and all
                                               arithmetic stay fp32.  Needs the tiled kernels: plan
                                               creation fails where only the direct ones apply.        */
    DAU_FLAG_STATIC_BUCKET = 1 << 5,        /* always run the kernels of the bucket max_kernel_size allows (no
                                               per-call selection from the actual offsets)             */
    DAU_FLAG_DENSE_BF16 = 1 << 6,           /* with DAU_FLAG_IO_BF16: calls whose offsets lie within +-4 run their
                                               two gather-sum passes (y, dx) as a DENSIFIED implicit GEMM on the
                                               bf16 matrix cores (units scattered into a 9x9 kernel per channel
                                               pair -- 7x7 when the call's offsets lie within +-3, decided on the
                                               device; taps and blurred activations rounded to bf16, fp32 sums),
                                               and -- from three units per channel on -- their parameter gradients as dense cross-correlations on the
                                               same matrix cores (filtered input and error rounded to bf16, fp32
                                               sums).  Otherwise the exact fp32 path.                        */
    DAU_FLAG_DENSE_WGRAD_NEVER = 1 << 7,    /* with DAU_FLAG_DENSE_BF16: keep the exact fp32 gather-dot for the
                                               parameter gradients whatever the unit count                   */
    DAU_FLAG_DENSE_WGRAD_ALWAYS = 1 << 8,   /* with DAU_FLAG_DENSE_BF16: dense parameter gradients from ONE unit
                                               per channel on (default: from three, where they start to pay) */
    DAU_FLAG_DENSE_SPLIT_F16 = 1 << 9,      /* The two gather-sum passes (y, dx) of calls whose offsets lie within +-2 / +-3 / +-4 can run
                                               as a DENSIFIED 5x5 / 7x7 / 9x9 implicit GEMM on the f16 matrix cores with BOTH operands
                                               split into two binary16 limbs (hi + lo, 22 significant bits; products hi*hi + lo*hi +
                                               hi*lo, fp32 sums, k_dense_split.hip): fp32 accuracy -- the same parity bar as the exact
                                               gather, measured margins in profiles/ -- at 3 * taps / 16 of the fp32 rate per (pixel,
                                               channel pair) instead of 4 G.  float32 or bfloat16 activations; the call's offsets decide
                                               on the device which member runs; every other call, and the parameter gradients, keep the
                                               exact kernels.  DEFAULT (neither flag): the members that pay for the plan's unit count
                                               (on whole tiles: radius 2 from two units per channel pair, radius 3 from three, radius 4 from four).
                                               This flag: all three members whatever the unit count.                        */
    DAU_FLAG_NO_DENSE_SPLIT = 1 << 10,      /* never: always the exact fp32 gather (excludes DAU_FLAG_DENSE_SPLIT_F16)      */
    DAU_FLAG_DEFAULT = DAU_FLAG_USE_INTERPOLATION
};

/* dau_conv_desc.algo */
enum {
    DAU_ALGO_AUTO = 0,      /* fastest kernels that support the shape                            */
    DAU_ALGO_DIRECT = 1,    /* plain one-thread-per-output HIP kernels (any shape)               */
    DAU_ALGO_TILED = 2      /* LDS-tiled MFMA gather / wave-reduced gradient kernels             */
};

/* which pass a workspace is sized for */
enum { DAU_PASS_FORWARD = 1, DAU_PASS_BACKWARD = 2 };

/* dau_conv_backward need-mask (params_propagate_down of Backward_gpu) */
enum {
    DAU_NEED_DX = 1 << 0,
    DAU_NEED_DW = 1 << 1,
    DAU_NEED_DMU1 = 1 << 2,
    DAU_NEED_DMU2 = 1 << 3,
    DAU_NEED_DSIGMA = 1 << 4,
    DAU_NEED_ALL = 31
};

typedef struct dau_conv_desc {
    int32_t struct_size;            /* sizeof(dau_conv_desc), for ABI evolution                   */
    int32_t batch;                  /* N                                                          */
    int32_t in_channels;            /* S                                                          */
    int32_t out_channels;           /* F  (attr num_output)                                       */
    int32_t units_per_channel;      /* G  (number_units_x * number_units_y, incl. ignored units)  */
    int32_t height, width;          /* H, W (output size == input size, dau_conv_op.cpp:185-188)  */
    int32_t max_kernel_size;        /* attr kernel_size: 9/17/33/37/41/49/65 -> largest offset
                                       bucket 4/8/16/18/20/24/32                                  */
    int32_t number_units_ignore;    /* attr number_units_ignore                                   */
    int32_t flags;                  /* DAU_FLAG_*                                                 */
    int32_t algo;                   /* DAU_ALGO_*                                                 */
    float sigma_hint;               /* host copy of sigma; sizes the blur support 2*ceil(5s)+1 as
                                       LayerSetUp does (base_dau_conv_layer.cpp:140-147); the taps
                                       themselves are computed from the device tensor each call   */
    float mu_learning_rate_factor;  /* attr mu_learning_rate_factor (grad op only)                */
} dau_conv_desc;

typedef struct dau_conv_plan dau_conv_plan; /* opaque */

typedef struct dau_conv_plan_info {
    int32_t offset_bucket;     /* largest (static) R in {4,8,16,18,20,24,32}                      */
    int32_t blur_support;      /* k of the k x k prefilter                                        */
    int32_t algo_forward;      /* DAU_ALGO_* actually used by dau_conv_forward / the dx pass      */
    int32_t algo_backward;     /* DAU_ALGO_* actually used for the parameter gradients            */
    int32_t drop_last_col;     /* unit_testing edge rule outcome for this W                       */
    int32_t drop_last_row;     /* ... and H                                                       */
    int32_t gather_patch;      /* tiled gather-sum: pixels per patch side (0: direct kernel)      */
    int32_t gather_stack;      /* ... and (image pair, patch) planes gathered per workgroup       */
    int32_t dot_windows;       /* tiled gather-dot: offset windows (1 for kernels <= 17)          */
    int32_t gather_windows;    /* tiled gather-sum: offset-window passes (1 for kernels <= 33)    */
    int32_t bucket_sets;       /* kernel sets a call can choose from (1: static bucket only)      */
    int32_t gather_dense_bf16; /* 1: the bucket-4 gather-sum passes use the densified bf16 GEMM;
                                  2: the parameter gradients too (three or more units)          */
    int32_t batch_slab_gather; /* images staged and gathered at a time by the y / dx passes and   */
    int32_t batch_slab_dot;    /* by the parameter-gradient pass of the static bucket (= batch
                                  unless the staged copy would exceed the workspace budget)      */
    int32_t dot_region;        /* tiled gather-dot of the static bucket: 100 * columns + rows of the
                                  positions one sweep covers (808, 807, 1404, 804; 0: direct)     */
    int32_t gather_fblock;     /* tiled gather-sum y pass of the static bucket: output channels per
                                  workgroup (4, 8, 12, 16; 0: direct)                             */
    int32_t gather_variant;    /* ... and the row of its kernel table (k_gather_mfma.hip kVariants) */
    int32_t dense_bf16_radius3; /* DAU_FLAG_DENSE_BF16 plans: 1 if calls within +-3 take the 7x7 members (gather-sum), 2 if the parameter
                                   gradients do too (49 displacements); 0: the 9x9 members only */
    int32_t gather_dense_split; /* bit r (r = 2, 3, 4): calls whose offsets lie within +-r can run their gather-sum passes as the
                                   two-limb f16 GEMM of that radius (0: never) */
} dau_conv_plan_info;

DAU_API int dau_conv_abi_version(void);
/* sha-256 prefix (16 hex digits) over the kernel sources this library was built from (the .hip and .hpp files of csrc, its Makefile, this
 * header), set by the Makefile: measurement files (profiles/) name the build they were taken from by it. */
DAU_API const char *dau_conv_build_id(void);
DAU_API const char *dau_conv_last_error(void); /* thread-local message of the last failing call */

/* Support k of the k x k prefilter a sigma needs: 2*ceil(5*sigma)+1 in float32 (base_dau_conv_layer.cpp:146).  A plan depends on
 * sigma_hint only through this number, so callers that cache plans key them on it (a trainable sigma changes every step). */
DAU_API int dau_conv_filter_support(float sigma);

DAU_API int dau_conv_plan_create(const dau_conv_desc *desc, dau_conv_plan **plan_out);
DAU_API int dau_conv_plan_destroy(dau_conv_plan *plan);
DAU_API int dau_conv_plan_get_info(const dau_conv_plan *plan, dau_conv_plan_info *info);
DAU_API int dau_conv_workspace_bytes(const dau_conv_plan *plan, int pass, size_t *bytes_out);

/* stream is a hipStream_t (NULL = default stream).  All pointers are device pointers. */
DAU_API int dau_conv_forward(const dau_conv_plan *plan, void *stream, const float *x, const float *w,
                     const float *mu1, const float *mu2, const float *sigma, float *y,
                     void *workspace, size_t workspace_bytes);

DAU_API int dau_conv_backward(const dau_conv_plan *plan, void *stream, const float *x, const float *dy,
                      const float *w, const float *mu1, const float *mu2, const float *sigma,
                      float *dx, float *dw, float *dmu1, float *dmu2, float *dsigma,
                      void *workspace, size_t workspace_bytes, int need_mask);

/* The parameter-gradient path of dau_conv_backward in two steps, for data-parallel callers (batch sharded over
 * ranks): _param_sums writes the raw sums r[k][s][g][f], k = {w, mu1, mu2, sigma} (4*S*G*F floats, linear in the
 * batch) into sums_out; the caller all-reduces that ONE flat buffer; _finalize_param_grads then applies the
 * elementwise tail on the reduced sums (dw = r0, dmu1 = w*r1*lr, dmu2 = w*r2*lr, dsigma = w*r3, ignored units -> 0,
 * NaN in dmu -> 0), so that every rank ends up bit-identical and a NaN on one rank is not zeroed before the exchange.
 * dau_conv_backward(need_mask of the four) == _param_sums followed by _finalize_param_grads. */
DAU_API int dau_conv_backward_param_sums(const dau_conv_plan *plan, void *stream, const float *x, const float *dy,
                                 const float *mu1, const float *mu2, const float *sigma, float *sums_out,
                                 void *workspace, size_t workspace_bytes);
DAU_API int dau_conv_finalize_param_grads(const dau_conv_plan *plan, void *stream, const float *sums, const float *w,
                                  float *dw, float *dmu1, float *dmu2, float *dsigma, int need_mask);

/* Waits for `stream`, then reports what the last call that used `workspace` found in mu1/mu2:
 * DAU_OK, DAU_FAILED_PRECONDITION (NaN) or DAU_INVALID_ARGUMENT (|mu| beyond the bucket).
 * max_abs_mu_out (may be NULL) receives max(|mu1|,|mu2|). */
DAU_API int dau_conv_check_status(const dau_conv_plan *plan, void *stream, const void *workspace,
                          float *max_abs_mu_out);

/* The same report without waiting, from pinned host memory.  A NaN or an out-of-range offset seen by ANY completed call of
 * this plan since the last report is STICKY: it stays until this function (or dau_conv_check_status) has returned it once,
 * however many good calls completed in between -- plans are shared by all layers of one shape, and a later layer's clean
 * status must not hide an earlier layer's error.  Without such an error: DAU_OK and max|mu| of the most recent completed
 * call (valid_out = 0 when none has completed yet).  A caller that checks this before each call learns of a bad offset
 * at most a few calls late, without ever stalling the stream. */
DAU_API int dau_conv_last_status(const dau_conv_plan *plan, float *max_abs_mu_out, int32_t *valid_out);

/* Building blocks exposed for parity tests (device pointers in and out).
 * filters_out: 6 planes of k*k floats in the order Gn, Dw, Dmu1, Dmu2, Dsigma, Gerr.
 * unit table: THE table the gather kernels consume, written by the very kernel every forward / backward call runs first
 * (prepare_units_kernel): S*G*F entries of 6 dwords {int32 floor(mu1'), int32 floor(mu2'), float w'*b00, w'*b01, w'*b10,
 * w'*b11} (factor index 2*dy + dx).  form 0: entry order [S][G][F], mu' = mu (forward, and with w = NULL -> w' = 1 the
 * parameter-gradient pass); form 1: entry order [F][G][S], mu' = -mu (the input-gradient pass,
 * base_dau_conv_layer.cu:299-325).  Ignored units have w' = 0. */
DAU_API int dau_conv_filters(const dau_conv_plan *plan, void *stream, const float *sigma, float *filters_out);
DAU_API int dau_conv_unit_table(const dau_conv_plan *plan, void *stream, const float *w, const float *mu1, const float *mu2,
                        int form, void *table_out);

/* Optional per-kernel timing for benchmarks (no reference counterpart; the reference only has the
 * compile-time PROFILE_CUDA block, dau_conv_forward_core.hpp:2506-2563).  Between _begin and _end every
 * dominant kernel launch is bracketed by HIP events on the caller's stream (asynchronous, no host sync).
 * Slots: 0 = gather-sum of dau_conv_forward, 1 = gather-sum of the dx pass, 2 = gather-dot (parameter
 * gradients).  _end waits for the events and returns, per slot, the summed milliseconds of those launches and the
 * number of PASSES they made up (a pass over large offsets takes several window launches).
 * Not thread-safe; one profiling session per plan at a time. */
enum { DAU_PROFILE_SLOTS = 3 };
DAU_API int dau_conv_profile_begin(dau_conv_plan *plan);
DAU_API int dau_conv_profile_end(dau_conv_plan *plan, double *ms_out, int32_t *passes_out);

#ifdef __cplusplus
}
#endif
#endif /* DAU_CONV_H_ */
