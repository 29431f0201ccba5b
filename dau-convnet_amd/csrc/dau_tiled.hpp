// Interface of the LDS-tiled kernels (DAU_ALGO_TILED).
//   gather-sum (forward y and input gradient dx): MFMA 4x4x1 outer products with
//     tap-separated accumulators over image-pair-interleaved LDS planes  (k_gather_mfma.hip)
//   gather-dot (parameter gradients): lane-per-unit packed-FMA kernel     (k_gather_dot.hip)
#pragma once
#include "dau_common.hpp"

namespace dau {

// Geometry of one gather-sum pass  (in-channels Cin -> out-channels Cout).
struct TiledConfig {
    int N, Cin, Cout, G, H, W;
    int R;            // offset bucket
    int blur_k;       // prefilter support
    int NP;           // image pairs = ceil(N/2)
    int windows;      // gather passes: 1, or 4 offset windows of radius R/2 for buckets 24 and 32 (every unit belongs to
                      // exactly one window; a pass gathers only its own units)
    int stack;        // (image pair, patch) planes gathered per workgroup
    int patches;      // patches per image: big or odd-sized images are gathered patch by patch
    int rows, pitch;  // staged plane of a patch: rows = ph + 2R + 1, pitch (in positions) >= pw + 2R + 1 with pitch % 32 == 8
    int tiles_x, tiles_y;   // position tiles of a patch
    int tile_w;             // their width: 8 (8 x 8 positions) or 32 (32 x 2)
    int fblock;       // out-channels per workgroup
    int variant;      // kernel instantiation id
    int debug;        // DAU_GATHER_DEBUG at plan creation (timing experiments)
    int bf16;         // activations in and out are bfloat16
};

bool tiled_gather_configure(int N, int Cin, int Cout, int G, int H, int W, int R, int blur_k, bool bf16, TiledConfig* cfg);
size_t tiled_gather_workspace_bytes(const TiledConfig& cfg);
// prepare: blur `in` ([N,Cin,H,W]) with the Gaussian (`filters` = output of launch_synth_filters; `mirrored`
// selects the flipped kernel of the input-gradient pass) into the staged pair-interleaved planes and pack the
// unit table (`table` is indexed [Cin][G][Cout]).  run: the gather itself, writing `out` ([N,Cout,H,W]).
int tiled_gather_windows(const TiledConfig& cfg);
// one pass per offset window: prepare(window) then run(accumulate = window > 0)
// guard: see dau_common.hpp (every kernel of the pass returns at once unless max|mu| lies in the guard's range)
void tiled_gather_prepare(hipStream_t st, const TiledConfig& cfg, const float* in, const float* filters, bool mirrored,
                          const UnitRef* table, void* workspace, int window, const Guard& guard);
void tiled_gather_run(hipStream_t st, const TiledConfig& cfg, float* out, void* workspace, bool accumulate, const Guard& guard);
// once per plan and device, before the first run: raises the kernels' dynamic-LDS limit
void tiled_gather_init(const TiledConfig& cfg);

// Densified bf16 gather-sum (k_dense_bf16.hip; DAU_FLAG_DENSE_BF16, offset bucket 4, bfloat16 activations): the units of
// every (input, output) channel pair scattered into a dense 9 x 9 kernel, implicit GEMM on the bf16 matrix cores.
struct DenseConfig {
    int N, Cin, Cout, G, H, W;
    int R, blur_k;
    int bf16;         // activations in and out are bfloat16 (required)
    int nsub;         // 8-pixel subtiles per column block = kernel instantiation
    int ftiles;       // 32-channel accumulator tiles per wave (2: four waves per workgroup, 1: eight)
};
// The functions exist once per offset radius of the dense form, in the namespaces r4 (|mu| <= 4: 9 x 9 taps) and r3 (|mu| <= 3:
// 7 x 7): the same source compiled twice (Makefile); `R` of configure must be the namespace's radius.
#define DAU_DECLARE_DENSE_GATHER(NS)                                                                                           \
    namespace NS {                                                                                                            \
    bool dense_gather_configure(int N, int Cin, int Cout, int G, int H, int W, int R, int blur_k, bool bf16, DenseConfig* cfg); \
    size_t dense_gather_workspace_bytes(const DenseConfig& cfg);                                                              \
    void dense_gather_init(const DenseConfig& cfg);                                                                           \
    /* prepare: dense kernel synthesis from the unit table ([Cin][G][Cout]) + blurred bf16 staging of `in`; run: the GEMM */  \
    void dense_gather_prepare(hipStream_t st, const DenseConfig& cfg, const float* in, const float* filters, bool mirrored,   \
                              const UnitRef* table, void* workspace, const Guard& guard);                                     \
    void dense_gather_run(hipStream_t st, const DenseConfig& cfg, float* out, void* workspace, const Guard& guard);           \
    }
DAU_DECLARE_DENSE_GATHER(r4)
DAU_DECLARE_DENSE_GATHER(r3)

// Densified gather-sum at fp32 accuracy (k_dense_split.hip): the same dense form with both operands split into two binary16
// limbs, three f16 MFMAs per tap -- fp32 or bf16 activations, inside the fp32 parity bar; the same source compiled once per
// offset radius (Makefile): namespaces s2, s3, s4 = offsets within +-2, +-3, +-4 (5 x 5, 7 x 7, 9 x 9 taps).
#define DAU_DECLARE_SPLIT_GATHER(NS)                                                                                           \
    namespace NS {                                                                                                            \
    bool split_gather_configure(int N, int Cin, int Cout, int G, int H, int W, int R, int blur_k, bool bf16, DenseConfig* cfg); \
    size_t split_gather_workspace_bytes(const DenseConfig& cfg);                                                              \
    void split_gather_init(const DenseConfig& cfg);                                                                           \
    void split_gather_prepare(hipStream_t st, const DenseConfig& cfg, const float* in, const float* filters, bool mirrored,   \
                              const UnitRef* table, void* workspace, const Guard& guard);                                     \
    void split_gather_run(hipStream_t st, const DenseConfig& cfg, float* out, void* workspace, const Guard& guard);           \
    }
DAU_DECLARE_SPLIT_GATHER(s2)
DAU_DECLARE_SPLIT_GATHER(s3)
DAU_DECLARE_SPLIT_GATHER(s4)

struct TiledDotConfig {
    Shape sh;
    int R, blur_k;
    int NP;
    int variant;
    int windows;      // offset windows of radius 8: 1 for R <= 8, 4 for R = 16, 9 for R = 24, 16 for R = 32; with more than
                      // one window the units are binned by window on the device and a window pass visits only its own
    bool as1, one_tile;   // tuning choices read from the environment at plan creation
    bool rw8;             // DAU_DOT_RW=8 at plan creation: 8-column regions only
    bool ring;            // window passes keep the error tile as a ring of rows (k_gather_dot.hip, RING)
    int region_cols, region_rows;   // positions per (item, input channel) sweep: 8 x 8, 8 x 7, 14 x 4 (or 8 x 4 in bucket 18)
    int rounds;           // DAU_DOT_ROUNDS at plan creation: workgroups per CU the chunking aims at (0: default)
    bool bf16;            // x and dy are bfloat16
    int ignore;           // number_units_ignore: binned window passes give those units no slot
    int debug;
};

bool tiled_dot_configure(const Shape& sh, int R, int blur_k, bool bf16, int ignore, TiledDotConfig* cfg);
size_t tiled_dot_workspace_bytes(const TiledDotConfig& cfg);
// r4[k][s][g][f] = sum_{n,p} dy'[n,f,p] * bilinear(x * D_k, p + o);  `filters` = output of launch_synth_filters.
void tiled_dot_prepare(hipStream_t st, const TiledDotConfig& cfg, const float* x, const float* dy,
                       const float* filters, const UnitRef* table_bare, int drop_col, int drop_row, void* workspace,
                       const Guard& guard);
// accumulate: add this batch slab's sums to r4 instead of overwriting it
void tiled_dot_run(hipStream_t st, const TiledDotConfig& cfg, float* r4, void* workspace, const Guard& guard, bool accumulate);
void tiled_dot_init(const TiledDotConfig& cfg);

// The four derivative-filtered copies of x, staged position-major (k_gather_dot.hip): x[N,C,H,W] -> xk[NP][cstride][Hp][Wp][4][2]
void launch_blur4_pack(hipStream_t st, const float* x, const float* filters, int N, int C, int cstride, int H, int W, int Hp,
                       int Wp, int blur_k, bool bf16, float* xk, const Guard& guard);
void blur4_pack_init(int blur_k);
bool blur4_pack_fits(int blur_k, int Hp, int Wp);

// Densified parameter gradients on the bf16 matrix cores (k_dense_wgrad.hip; DAU_FLAG_DENSE_BF16, bucket 4, bfloat16
// activations, three or more units): C_k[d][s][f] = sum_{n,q} Xk[n,s,q+d] * E'[n,f,q] for the 9 x 9 displacements d as a GEMM
// with K = (image, position), then r_k[u] = sum_taps b_t(u) * C_k[o_u + t].
struct WgradConfig {
    Shape sh;
    int blur_k;
    int SB, FB, NC;       // 32-channel blocks of S and F, 16-image chunks of N
    int HsT, WsT, WT, nseg;   // staged Xk plane (H+8 rows, nseg*WT+8 columns rounded up to 8); a row is walked in nseg segments of WT
                          // columns (an instantiated length)
    int splits;           // the image chunks are cut into `splits` ranges (partial sums per range)
    int Hp, Wp;           // plane of the intermediate fp32 copy (blur4_pack)
    bool fused;           // XkT is written from x through the transposed bf16 copy (instantiated prefilter supports); else
                          // through the fp32 copy of blur4_pack
};
// (once per radius, as the dense gather-sum: namespaces r4 and r3)
// r4[k][s][g][f] = raw parameter-gradient sums; x, dy bfloat16 NCHW; table = bare unit table [S][G][F].  nk: kinds computed (w, mu1,
// mu2, sigma in this order): 3 leaves the sigma block of r4 untouched (a caller that does not want dsigma; the reference's
// last_k_optional: base_dau_conv_layer.hpp:213, src/dau_conv/dau_conv_impl/dau_conv_backward.cpp:219)
#define DAU_DECLARE_DENSE_WGRAD(NS)                                                                                            \
    namespace NS {                                                                                                            \
    bool dense_wgrad_configure(const Shape& sh, int blur_k, bool bf16, WgradConfig* cfg);                                     \
    size_t dense_wgrad_workspace_bytes(const WgradConfig& cfg);                                                               \
    void dense_wgrad_init(const WgradConfig& cfg);                                                                            \
    void dense_wgrad_run(hipStream_t st, const WgradConfig& cfg, const float* x, const float* dy, const float* filters,       \
                         const UnitRef* table, int drop_col, int drop_row, float* r4, void* workspace, const Guard& guard, int nk); \
    }
DAU_DECLARE_DENSE_WGRAD(r4)
DAU_DECLARE_DENSE_WGRAD(r3)

}  // namespace dau
