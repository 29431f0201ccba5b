// Tiled gather-sum on the matrix cores: forward output and input gradient.
//
//   out[n,f,y,x] = sum_{c,g} sum_{t in 2x2 taps} w_t(c,g,f) * Xb[n,c, y+oy+dy_t, x+ox+dx_t]
//
// Replaces the reference's DAUConv_forward_pipeline_kernel + interleave_input_data_kernel +
// perpare_weights_and_offsets (include/dau_conv/dau_conv_impl/dau_conv_forward_core.hpp:804-1605,
// 1607-1732, 1858-2215) and the prefilter pass caffe_gpu_convolve2 (src/dau_conv/util/convolve.cu:48-131).
// It is a different algorithm, designed for CDNA4:
//
//  * Tap-separated accumulation.  For a unit u=(c,g,f) with integer displacement o_u, ONE value
//    Xb[q+o_u] feeds all four bilinear taps:  Z_t[q] += w_t(u) * Xb[q+o_u]  (t = 0..3), and the output
//    is assembled once per (n,f) at the end:  out[p] = Z_0[p] + Z_1[p+(0,1)] + Z_2[p+(1,0)] + Z_3[p+(1,1)].
//    So the inner loop needs one LDS float per four MACs, at any (unaligned) displacement.
//  * The four MACs are one row of v_mfma_f32_4x4x1_16b_f32: a lane's own X value is the B operand, the
//    four tap weights sit in the four lanes of its quad as the A operand (lane 4b+i holds w_i), and
//    D register i of lane 4b+j accumulates Z_i for that lane's position.  Exact fp32 FMA numerics.
//  * Two images are interleaved element-wise in the staged planes, so one ds_read_b64 (256 B/clk LDS
//    rate, naturally aligned whatever the displacement) feeds two MFMAs.
//  * Positions are grouped in 8x8 tiles whose LDS addresses differ from the lane base by compile-time
//    immediates; the staged plane pitch is = 8 (mod 32) positions so every ds_read_b64 of a tile is
//    bank-conflict free.  The (H+1)x(W+1) domain of Z needs one extra row/column: two "edge" tiles.
//  * A workgroup owns (image pair, FB output channels) and streams all input channels through a
//    double-buffered LDS plane filled by global_load_lds (no VGPR staging) from a pre-blurred,
//    zero-bordered, pair-interleaved copy written by blur_pack_kernel.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "dau_tiled.hpp"

namespace dau {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

namespace {

constexpr int kFB = 4;            // output channels per workgroup (default; small-map variants take 8 or 16)
constexpr int kUnitDwords = 8;    // packed unit: {w00,off,w01,off,w10,offT,w11,offT} (offT: displacement in the transposed strip)

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// An image is cut into patches of ph x pw pixels; every patch is gathered by one workgroup pass from its own staged
// plane (the patch plus a border of R on the left/top and R+1 on the right/bottom, taken from the neighbouring pixels of
// the blurred image, zero outside it).  EDGE variants have ph = 8*ty, pw = 8*tx whatever H and W are (pixels beyond the
// image are computed and dropped at the store) and two extra "edge" tiles for the last row / column of Z; non-EDGE
// variants have patches of at most 8*ty - 1 by 8*tx - 1 pixels (a whole image when it is that small), whose
// (ph+1) x (pw+1) domain of Z fits the regular tiles.
struct Geometry {
    int H, W, R;              // image, offset bucket
    int Rt;                   // offset radius one staged plane covers: R for R <= 16, R/2 for R = 24, 32
    int nwin1;                // R > 16: the offset range is cut into nwin1 x nwin1 windows of radius Rt, one gather pass per
                              // window (shifted planes; the units are binned by window, a pass gathers only its own),
                              // outputs accumulate
    int ph, pw;               // patch (pixels)
    int npx, npy;             // patches per image
    int rows, pitch, cols;    // staged plane of one patch: rows x pitch positions, the first `cols` columns carry data
    int tx, ty, tw;           // regular tiles of a patch (tw x 64/tw positions each)
    int edge;                 // 1: separate edge tiles for the extra row / column of Z; 0: regular tiles cover (H+1)x(W+1)
    int variant;              // row of kVariants, -1: no instantiated kernel fits
    int sk;                   // (image pair, patch) planes stacked per workgroup
    int fb;                   // output channels per workgroup
    int nb;                   // plane buffers in LDS (2, or 3 for the lagged kernels)
    int strip_pitch;          // EDGE: rows of the transposed strip (columns pw .. pw+2R of the plane, column-major)
    size_t strip_off;         // byte offset of the strip inside a staged plane
    size_t plane_bytes;       // padded to 1 KiB (one global_load_lds wave instruction)
};

struct Variant { int tx, ty, pitch, edge, split, sk, plane_bytes, fb, tuning, nb = 2, tw = 8; };   // tw: tile width (8 x 8 or 32 x 2 positions)
// instantiated kernels (add rows here and in the dispatch of tiled_gather_run)
const Variant kVariants[] = {
    {7, 7, 72, 1, 2, 1, 0, 4, 0},       // 0: 56x56 patches, R=4
    {7, 7, 104, 1, 2, 1, 0, 4, 0},      // 1: 56x56 patches, R=8/16
    {4, 4, 72, 1, 1, 1, 0, 4, 0},       // 2: 32x32 patches, R<=16
    {2, 2, 40, 1, 1, 1, 0, 4, 0},       // 3: 16x16 patches, R<=8
    {3, 3, 40, 1, 1, 1, 0, 4, 0},       // 4: 24x24 patches, R=4
    {1, 1, 40, 1, 1, 1, 0, 4, 0},       // 5: 8x8 patches, R<=8
    {4, 4, 40, 0, 1, 1, 0, 4, 0},       // 6: one 25..31 pixel image (27x27, 28x28), R=4
    {7, 7, 72, 1, 3, 1, 0, 4, 1},       // 7: 56x56, R=4, three waves per output channel (tuning alternative, DAU_GATHER_SPLIT=3)
    // Small feature maps, R=4: several planes stacked per workgroup (plane size is part of the instantiation) and, for
    // the smallest, more output channels per workgroup so that a staged plane is fetched from L2 less often per MFMA.
    {4, 4, 72, 1, 2, 2, 26624, 4, 0},   // 8: 2 x 32x32 patches
    {3, 3, 40, 1, 2, 4, 13312, 4, 0},   // 9: 4 x 24x24
    {2, 2, 40, 1, 1, 4, 10240, 8, 0},   // 10: 4 x 16x16, 8 channels
    {1, 1, 40, 1, 1, 4, 7168, 16, 0},   // 11: 4 x 8x8, 16 channels
    {4, 4, 40, 0, 2, 3, 13312, 4, 0},   // 12: 3 x one 25..31 pixel image
    // whole images of at most 23 / 15 / 7 pixels: the (H+1) x (W+1) domain of Z fits the regular tiles, no edge tiles
    {3, 3, 40, 0, 1, 2, 10240, 8, 0},   // 13: 2 x one 17..23 pixel image, 8 channels
    {2, 2, 40, 0, 1, 4, 8192, 8, 0},    // 14: 4 x one 9..15 pixel image (14x14), 8 channels
    {1, 1, 40, 0, 1, 8, 5120, 16, 0},   // 15: 8 x one <=7 pixel image (7x7), 16 channels
    // Large offsets on large images: edge-free 31 pixel patches (the 32 x 32 domain of Z is the 4 x 4 regular tiles, so
    // no strip, and the plane is just (32 + 2R)^2 positions): R <= 20 under pitch 72, R <= 28 under pitch 104 with eight
    // output channels per workgroup sharing the bigger plane.  Bucket 24 therefore needs no offset windows.
    {4, 4, 72, 0, 2, 1, 0, 4, 0},       // 16: 31 pixel patches, R <= 20
    {4, 4, 104, 0, 1, 1, 0, 8, 0},      // 17: 31 pixel patches, R <= 28, 8 channels
    {4, 4, 40, 0, 1, 1, 0, 8, 0},       // 18: one 25..31 pixel image, eight channels on its plane (28x28 at 512 channels:
                                        //     73.7 / 69.4 TF against 71.1 / 67.8 for the three stacked images of row 12)
    {4, 4, 40, 0, 2, 2, 13312, 4, 2},   // 19: two stacked 25..31 pixel images (tuning alternative, DAU_GATHER_VARIANT=19: no gain)
    {7, 7, 72, 1, 2, 1, 0, 4, 2, 3},    // 20: row 0 with three plane buffers and lagged partner waves (DAU_GATHER_VARIANT=20)
    {4, 4, 40, 0, 1, 1, 0, 12, 0},      // 21: one 25..31 pixel image, twelve channels on its plane: a plane is fetched from L2 a
                                        //     third as often as with four (27x27, 96 -> 256 channels, 64 images: dx 0.78 -> 0.54 ms;
                                        //     28x28 at 512 channels: 11.0 / 11.6 -> 10.5 / 10.9 ms).  Sixteen channels = 1024
                                        //     threads spill (20x slower): not instantiated.
    {4, 4, 104, 0, 1, 1, 0, 12, 0},     // 22: 31 pixel patches, R <= 28, twelve channels (512x512, bucket 18: 250 -> 242 ms per pass)
    {4, 4, 72, 0, 1, 1, 0, 12, 0},      // 23: 31 pixel patches, R <= 20, twelve channels on a pitch-72 plane: a third fewer staged
                                        //     bytes to write, fetch and hold than row 22 for the buckets that fit (16, 18, 20)
    // Tiles of 32 x 2 positions (a half wave reads 32 consecutive positions: conflict free under any pitch): the 29 x 29 domain
    // of a 28 pixel image is 15 such tiles (30 rows) instead of 16 tiles of 8 x 8 (32 rows), a 27 pixel image 14.  Measured
    // (same box, 28x28 at 512 channels): 10.77 / 11.49 ms against 10.43 / 11.04 ms for row 21 although a sixteenth of the MFMAs is
    // gone and ds_read_b64 runs at the same 2.25 clk for both lane maps (tools/microbench/lds_tile_shapes.hip): explicit request
    // only (DAU_GATHER_VARIANT=24 / 25 in the tuning build)
    {1, 15, 40, 0, 1, 1, 0, 12, 2, 2, 32},   // 24: one 28 / 29 pixel image, twelve channels
    {1, 14, 40, 0, 1, 1, 0, 12, 2, 2, 32},   // 25: one 25..27 pixel image, twelve channels
    // the twelve-channel rows with three plane buffers and lagged partner waves (one workgroup per CU: all its waves meet at every
    // channel's barrier, nothing else on the CU fills the start-up after it)
    {4, 4, 40, 0, 1, 1, 0, 12, 2, 3},        // 26: row 21 lagged (measured 2 % slower than row 21 at 28x28 / 512 channels)
    {4, 4, 72, 0, 1, 1, 0, 12, 2, 3},        // 27: row 23 lagged (correct, but 13 x slower than row 23 on 512x512 maps: explicit request only)
};

// one (channel block, input channel) slice of the packed unit table: [G slots][fb channels][8 dwords]; window passes
// (binned) append the number of slots each channel uses in this pass: [fb] dwords
size_t ut_stride_bytes(int G, int fb, bool binned) { return round_up((size_t)G * fb * kUnitDwords * 4 + (binned ? fb * 4 : 0), 1024); }

// N, Cout only steer the choice between stacked and plain variants (enough workgroups to fill the chip)
// only: >= 0 pins the row of kVariants (the plan's choice, so that every later call sees the same layout)
Geometry make_geometry(int H, int W, int R, int G, int N, int Cout, int only = -1) {
    Geometry g{};
    g.H = H; g.W = W; g.R = R; g.variant = -1;
    // offset windows only where no instantiated kernel can stage the whole bucket around a patch (R = 32 today)
    const int Rfull = R;
    for (int nwin1 = 1; nwin1 <= 2 && g.variant < 0; ++nwin1) {
    if (Rfull % nwin1) continue;
    g.nwin1 = nwin1; g.Rt = Rfull / nwin1;
    const bool binned = g.nwin1 > 1;
    R = g.Rt;                                                // everything below sizes ONE plane
    // The environment is consulted at plan creation only (only < 0); afterwards the plan's row is passed in.
    //   DAU_GATHER_VARIANT=<row>  pins the kernel (tests of the small-map variants on small batches)
    //   DAU_GATHER_SPLIT=3        the three-waves-per-channel tuning alternative;  DAU_GATHER_STACK=0  never stack
    int want_split = 0;
    bool never_stack = false;
    if (only < 0) {
        want_split = DAU_TUNE_INT("DAU_GATHER_SPLIT", 0);
        never_stack = DAU_TUNE_INT("DAU_GATHER_STACK", 1) == 0;
        only = DAU_TUNE_INT("DAU_GATHER_VARIANT", -1);
    }
    double best = 0.0;
    for (int i = 0; i < (int)(sizeof(kVariants) / sizeof(kVariants[0])); ++i) {
        const Variant& v = kVariants[i];
        if (only >= 0 && i != only) continue;
#ifndef DAU_TUNING
        if (v.tuning) continue;                                                       // not instantiated in the release build
#endif
        if (only < 0 && v.tuning == 2) continue;                                      // explicit request only
        if (only < 0 && v.tuning && v.split != want_split) continue;
        int ph, pw, cols, rows;
        if (v.edge) {
            ph = v.ty * 8; pw = v.tx * 8;
            cols = pw + 1 + 2 * R; rows = ph + 1 + 2 * R;
        } else {
            // regular tiles cover the (ph+1) x (pw+1) domain of Z: a whole image, or patches of 8*t - 1 pixels
            const int th = 64 / v.tw;
            ph = H < v.ty * th - 1 ? H : v.ty * th - 1; pw = W < v.tx * v.tw - 1 ? W : v.tx * v.tw - 1;
            cols = v.tx * v.tw + 2 * R; rows = v.ty * th + 2 * R;
        }
        if (cols > v.pitch) continue;
        const size_t plane = round_up(((size_t)rows * v.pitch + (v.edge ? (size_t)(2 * R + 1) * rows : 0)) * 8, 1024);
        if (v.plane_bytes && plane != (size_t)v.plane_bytes) continue;
        // two buffers of sk planes + two unit slices must fit the 160 KiB of LDS
        if (v.nb * v.sk * plane + v.nb * ut_stride_bytes(G, v.fb, binned) > 160 * 1024) continue;
        if (v.nb == 3 && binned) continue;                                           // lagged kernels: every channel has G units
        const int npx = (W + pw - 1) / pw, npy = (H + ph - 1) / ph;
        const long planes = (long)((N + 1) / 2) * npx * npy, groups = (planes + v.sk - 1) / v.sk;
        if (never_stack && v.sk > 1) continue;
        const long blocks = groups * ((Cout + v.fb - 1) / v.fb);
        // relative cost in tile units: MFMA tiles + the DMA of the planes + a fixed part per workgroup and channel
        // (barrier, unit fetch, exposed LDS latency: ~12 tiles' worth, fitted to 28x28 stacked vs plain)
        // (per four output channels: a workgroup with more channels fetches its planes and pays its barriers once for all)
        const int edge_tiles = v.edge ? ((v.tx + v.ty) * 8 + 1 <= 64 ? 1 : 2) : 0;
        double cost = (double)groups * (v.sk * (v.tx * v.ty + edge_tiles + 0.02 * rows * v.pitch / 8.0 * (4.0 / v.fb)) + 12.0 * (4.0 / v.fb));
        // a grid that ends with a nearly empty round of workgroups wastes the chip: price the rounds, not the blocks
        cost *= (double)((blocks + 255) / 256 * 256) / (double)blocks;
        // channel slots of the last, partly filled channel block
        cost *= (double)((Cout + v.fb - 1) / v.fb * v.fb) / (double)Cout;
        if (v.tuning && only < 0) cost = 0.0;                // explicitly requested
        if (g.variant >= 0 && cost >= best) continue;
        best = cost;
        g.variant = i; g.ph = ph; g.pw = pw; g.npx = npx; g.npy = npy; g.rows = rows; g.cols = cols; g.pitch = v.pitch;
        g.tx = v.tx; g.ty = v.ty; g.tw = v.tw; g.edge = v.edge; g.sk = v.sk; g.fb = v.fb; g.nb = v.nb;
    }
    }
    if (g.variant < 0) return g;
    R = g.Rt;
    g.strip_pitch = g.rows;
    g.strip_off = (size_t)g.rows * g.pitch * 8;
    g.plane_bytes = round_up(g.strip_off + (g.edge ? (size_t)(2 * R + 1) * g.strip_pitch * 8 : 0), 1024);
    return g;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// blur + pack: in[N,C,H,W] -> staged[NP][patch][C][rows][pitch][2].  Staged position (row, col) of patch (py, px) is the
// blurred image at (py*ph - R + row, px*pw - R + col), zero outside the image, blurred with the separable Gaussian
// gx (x) gy.  One workgroup per (image pair, patch, channel): raw window -> LDS, horizontal pass -> LDS, vertical pass
// -> coalesced rows.  HBM bound: reads the NCHW input once (plus patch halos), writes the staged copy once.
// ------------------------------------------------------------------------------------------------
struct BlurPackArgs {
    const float* in;
    const float* taps;
    float* staged;
    int mirrored, N, C, H, W, R, k;
    int bf16;                   // input is bfloat16 (DAU_FLAG_IO_BF16)
    int cy, cx;                 // offset-window centre (0 unless the bucket is cut into windows)
    int ph, pw, npx, npy;
    int rows, pitch, cols, strip_cols;
    size_t plane_floats;
    int ppb;                    // planes per workgroup (small planes: every group of 8/ppb waves blurs its own plane)
    int bands, band_rows;       // big planes: a workgroup writes band_rows staged rows of its plane (less LDS, more
                                // workgroups per CU to hide the memory latency of this HBM-bound kernel)
    int planes;                 // (image pair, patch, channel, band) work items in total
    unsigned lds_plane_floats;  // LDS floats per plane
    Guard guard;
};

// K: compile-time prefilter support (taps live in SGPRs, tap loops unrolled); K = 0: any support, taps re-read per use
template <int K>
#ifndef DAU_BLUR_WAVES_PER_EU
#define DAU_BLUR_WAVES_PER_EU 6
#endif
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(DAU_BLUR_WAVES_PER_EU))) blur_pack_kernel(const BlurPackArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!guard_pass(a.guard)) return;
    const int C = a.C, H = a.H, W = a.W, R = a.R, k = K ? K : a.k;
    const int lane = threadIdx.x & 63;
    const int nw = (blockDim.x >> 6) / a.ppb;     // waves per plane
    const int sub = (threadIdx.x >> 6) / nw, wave = (threadIdx.x >> 6) % nw;
    int pid = blockIdx.x * a.ppb + sub;
    const bool active = pid < a.planes;           // the last workgroup may have idle wave groups (they still reach the barriers)
    if (!active) pid = a.planes - 1;
    const int band = pid % a.bands; pid /= a.bands;
    const int row0 = band * a.band_rows, row1 = row0 + a.band_rows < a.rows ? row0 + a.band_rows : a.rows;   // staged rows of this band
    const int c = pid % C;
    const int npp = pid / C;                       // (image pair, patch)
    const int npatch = a.npx * a.npy;
    const int np = npp / npatch, patch = npp % npatch;
    const int wy0 = (patch / a.npx) * a.ph - R + a.cy, wx0 = (patch % a.npx) * a.pw - R + a.cx;   // image coordinates of staged (0, 0)
    // the part of the window that lies inside the image
    const int ya0 = wy0 + row0 > 0 ? wy0 + row0 : 0;
    const int ya1r = wy0 + row1 < H ? wy0 + row1 : H, ya1 = ya1r > ya0 ? ya1r : ya0;
    const int xa0 = wx0 > 0 ? wx0 : 0, xa1 = wx0 + a.cols < W ? wx0 + a.cols : W;
    const int kr = (k - 1) / 2;
    const int bw = xa1 - xa0;
    const int lw = bw + 2 * kr, lh = (ya1 - ya0) + 2 * kr;
    f2* A = reinterpret_cast<f2*>(lds + (size_t)sub * a.lds_plane_floats);   // raw window, zero outside the image   [lh][lw]
    f2* B = A + (size_t)lh * lw;                  // after the horizontal pass            [lh][bw]
    const int tp = a.band_rows | 1;
    f2* T = B + (size_t)lh * bw;                  // strip columns of this band, column-major  [strip_cols][tp]
    const float* gxp = a.taps + (a.mirrored ? kTapGXR : kTapGX) * kTapPitch;
    const float* gyp = a.taps + (a.mirrored ? kTapGYR : kTapGY) * kTapPitch;
    float gxr[K ? K : 1], gyr[K ? K : 1];
    if (K) {
#pragma unroll
        for (int i = 0; i < K; ++i) { gxr[i] = gxp[i]; gyr[i] = gyp[i]; }
    }
    auto gx = [&](int i) { return K ? gxr[i] : gxp[i]; };
    auto gy = [&](int i) { return K ? gyr[i] : gyp[i]; };
    const int n0 = 2 * np, n1 = 2 * np + 1;
    const long p0 = ((long)n0 * C + c) * H * W, p1 = ((long)(n1 < a.N ? n1 : n0) * C + c) * H * W;   // element offsets
    const bool bf16 = a.bf16 != 0;
    const float m1 = n1 < a.N ? 1.0f : 0.0f;      // odd batch: the second image of the last pair is zero
    // rows x cols of work for this plane's waves: a wave per row when the rows are wide, a flat index when they are narrow
    // (a 7-pixel row would leave most of a wave idle)
    // (nor a 72-position row: its second wave instruction would run with 8 of 64 lanes)
    auto for_each = [&](int rows_, int cols_, auto&& body) {
        if (cols_ >= 56 && (cols_ % 64 == 0 || cols_ % 64 >= 48)) {
            for (int r = wave; r < rows_; r += nw)
                for (int x = lane; x < cols_; x += 64) body(r, x);
        } else {
            for (int t = wave * 64 + lane; t < rows_ * cols_; t += nw * 64) { const int r = t / cols_; body(r, t - r * cols_); }
        }
    };
    // raw window -> LDS, the loads of a batch in flight together (load_phase, dau_common.hpp)
    auto fill = [&](auto bfc) {
        constexpr bool BF = decltype(bfc)::value;
        struct Raw2 { typename RawAct<BF>::type v0, v1; };
        load_phase<Raw2>(lh, lw, wave, nw, lane,
            [&](int r, int xl) {
                const int yy = ya0 - kr + r, xx = xa0 - kr + xl;
                const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
                const long off = in ? (long)yy * W + xx : 0;            // outside the image: element 0 (valid), discarded
                return Raw2{load_raw<BF>(a.in, p0 + off), load_raw<BF>(a.in, p1 + off)};
            },
            [&](int r, int xl, Raw2 v) {
                const int yy = ya0 - kr + r, xx = xa0 - kr + xl;
                const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
                A[r * lw + xl] = f2{mask_act(act_of(v.v0), in), mask_act(m1 * act_of(v.v1), in)};
            });
    };
    if (bf16) fill(std::true_type{}); else fill(std::false_type{});
    __syncthreads();
    for_each(lh, bw, [&](int r, int x) {
        const int yy = ya0 - kr + r;
        f2 acc = {0.0f, 0.0f};
        if (yy >= 0 && yy < H) {
#pragma unroll
            for (int i = 0; i < k; ++i) acc = __builtin_elementwise_fma(A[r * lw + x + i], f2{gx(i), gx(i)}, acc);
        }
        B[r * bw + x] = acc;
    });
    __syncthreads();
    f2* out = reinterpret_cast<f2*>(a.staged + ((size_t)npp * C + c) * a.plane_floats);
    f2* strip = out + (size_t)a.rows * a.pitch;
    for_each(active ? row1 - row0 : 0, a.pitch, [&](int brow, int col) {
        const int row = row0 + brow;
        const int iy = wy0 + row, ix = wx0 + col;
        f2 acc = {0.0f, 0.0f};
        if (iy >= ya0 && iy < ya1 && ix >= xa0 && ix < xa1) {
#pragma unroll
            for (int j = 0; j < k; ++j) acc = __builtin_elementwise_fma(B[(iy - ya0 + j) * bw + (ix - xa0)], f2{gy(j), gy(j)}, acc);
        }
        out[row * a.pitch + col] = acc;
        // columns pw .. pw+2R are stored a second time column-major: the edge-column tile of the gather reads a
        // vertical run of positions, which is bank-conflict free only in this orientation.  They are collected in LDS
        // and written as runs of rows below (straight from here they were nine scattered 8-byte writes per row, three
        // times the write transactions of the row itself)
        if (a.strip_cols > 0 && col >= a.pw && col < a.pw + a.strip_cols) T[(col - a.pw) * tp + brow] = acc;
    });
    if (a.strip_cols > 0) {
        __syncthreads();
        const int nr = active ? row1 - row0 : 0;
        for (int t = wave * 64 + lane; t < a.strip_cols * nr; t += nw * 64) {
            const int cidx = t / nr, r = t - cidx * nr;
            strip[(size_t)cidx * a.rows + row0 + r] = T[cidx * tp + r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// unit packing: UnitRef[Cin][G][Cout] -> packed[FBn][Cin] slices of [G][kFB][8 dwords], each slice padded to 1 KiB
// ------------------------------------------------------------------------------------------------
// R: offset bucket; Rt, nwin1, window: the pass's offset window (Rt == R, nwin1 == 1: the whole bucket)
__global__ void pack_units_kernel(const UnitRef* __restrict__ table, int Cin, int G, int Cout, int pitch, int R, int Rt,
                                  int nwin1, int window, int strip_pitch, int ut_stride_dwords, int kFB,
                                  unsigned int* __restrict__ packed, const Guard guard) {
    if (!guard_pass(guard)) return;
    const int nfb = (Cout + kFB - 1) / kFB;
    const long total = (long)nfb * Cin * G * kFB;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int fi = (int)(idx % kFB);
        const int g = (int)((idx / kFB) % G);
        const int c = (int)((idx / ((long)kFB * G)) % Cin);
        const int fb = (int)(idx / ((long)kFB * G * Cin));
        const int f = fb * kFB + fi;
        UnitRef u{0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
        if (f < Cout) u = table[((long)c * G + g) * Cout + f];
        // a unit belongs to exactly one offset window; in the other passes it contributes nothing
        int wy = (u.oy + R) / (2 * Rt), wx = (u.ox + R) / (2 * Rt);
        wy = wy < nwin1 ? wy : nwin1 - 1; wx = wx < nwin1 ? wx : nwin1 - 1;
        const int cy = -R + Rt + 2 * Rt * (window / nwin1), cx = -R + Rt + 2 * Rt * (window % nwin1);
        if (wy != window / nwin1 || wx != window % nwin1) u = UnitRef{cx, cy, 0.0f, 0.0f, 0.0f, 0.0f};
        const int ox = u.ox - cx, oy = u.oy - cy;    // displacement relative to the window centre, |.| <= Rt
        const int off = (oy * pitch + ox) * 8;       // byte displacement inside a staged plane
        const int offt = ((ox + Rt) * strip_pitch + oy) * 8;   // ... and inside its transposed strip
        unsigned int* dst = packed + ((long)fb * Cin + c) * ut_stride_dwords + (g * kFB + fi) * kUnitDwords;
        dst[0] = __float_as_uint(u.w00); dst[1] = (unsigned)off;
        dst[2] = __float_as_uint(u.w01); dst[3] = (unsigned)off;
        dst[4] = __float_as_uint(u.w10); dst[5] = (unsigned)offt;
        dst[6] = __float_as_uint(u.w11); dst[7] = (unsigned)offt;
    }
}

// Window passes (R > 16): the units of (input channel c, output channel f) that fall into this pass's offset window are
// compacted into the first slots of the slice and their number goes into the slice's count word, so that the gather
// visits only them (the reference splits the work of its large-offset kernels by K instead, dau_conv_backward.cpp:194-231;
// round 1 here visited every unit in every window with zeroed weights).  Units whose four weights are all zero (ignored
// units, number_units_ignore) are dropped.  One thread per (channel block, c, channel of the block).
__global__ void pack_units_binned_kernel(const UnitRef* __restrict__ table, int Cin, int G, int Cout, int pitch, int R,
                                         int Rt, int nwin1, int window, int strip_pitch, int ut_stride_dwords, int kFB,
                                         unsigned int* __restrict__ packed, const Guard guard) {
    if (!guard_pass(guard)) return;
    const int nfb = (Cout + kFB - 1) / kFB;
    const long total = (long)nfb * Cin * kFB;
    const int cy = -R + Rt + 2 * Rt * (window / nwin1), cx = -R + Rt + 2 * Rt * (window % nwin1);
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int fi = (int)(idx % kFB);
        const int c = (int)((idx / kFB) % Cin);
        const int fb = (int)(idx / ((long)kFB * Cin));
        const int f = fb * kFB + fi;
        unsigned int* slice = packed + ((long)fb * Cin + c) * ut_stride_dwords;
        int cnt = 0;
        if (f < Cout) {
            for (int g = 0; g < G; ++g) {
                const UnitRef u = table[((long)c * G + g) * Cout + f];
                int wy = (u.oy + R) / (2 * Rt), wx = (u.ox + R) / (2 * Rt);
                wy = wy < nwin1 ? wy : nwin1 - 1; wx = wx < nwin1 ? wx : nwin1 - 1;
                if (wy != window / nwin1 || wx != window % nwin1) continue;
                if (u.w00 == 0.0f && u.w01 == 0.0f && u.w10 == 0.0f && u.w11 == 0.0f) continue;
                const int ox = u.ox - cx, oy = u.oy - cy;    // displacement relative to the window centre, |.| <= Rt
                const int off = (oy * pitch + ox) * 8;
                const int offt = ((ox + Rt) * strip_pitch + oy) * 8;
                unsigned int* dst = slice + (cnt * kFB + fi) * kUnitDwords;
                dst[0] = __float_as_uint(u.w00); dst[1] = (unsigned)off;
                dst[2] = __float_as_uint(u.w01); dst[3] = (unsigned)off;
                dst[4] = __float_as_uint(u.w10); dst[5] = (unsigned)offt;
                dst[6] = __float_as_uint(u.w11); dst[7] = (unsigned)offt;
                ++cnt;
            }
        }
        slice[G * kFB * kUnitDwords + fi] = (unsigned)cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
    const char* staged;        // [NP][Cin][plane_bytes]
    const char* packed;        // [NFB][Cin][ut_stride]
    float* out;                // [N][Cout][H][W]
    int N, Cin, Cout, G, H, W, R;
    int npx, npy;              // patches per image (EDGE variants: 8*TY x 8*TX pixels each)
    int ph, pw;                // patch size in pixels (non-EDGE variants: at most 8*TY - 1 by 8*TX - 1)
    int nfb;                   // ceil(Cout / kFB)
    unsigned plane_bytes, ut_stride;
    unsigned strip_off;        // byte offset of the transposed strip inside a plane
    unsigned zpitch;           // epilogue Z-plane pitch (floats)
    int accumulate;            // 1: add to out (second and later offset-window passes)
    int bf16;                  // out is bfloat16 (DAU_FLAG_IO_BF16)
    int debug;                 // timing experiments only (DAU_GATHER_DEBUG): 1 = no plane refills after the first two
    int binned;                // window pass: every channel's slot count comes from its slice's count words
    Guard guard;
};

// TX, TY : regular 8x8 tiles of one plane;  PITCH: staged pitch (positions);  EDGE: two extra edge tiles per plane
// SPLIT  : waves sharing one output channel (tiles are dealt out in contiguous ranges)
// SK, PB : SK consecutive (image pair, patch) planes of PB bytes each are stacked in one LDS buffer and gathered by one
//          workgroup (small feature maps: more tiles per wave against the per-channel fixed cost).  PB = 0 with SK = 1:
//          plane size taken at run time.
// FB     : output channels per workgroup (one wave, or SPLIT waves, each)
// NB     : plane buffers in LDS.  2: every wave gathers channel c between barriers c and c+1.  3 ("lagged"): the second
//          half of the waves (the SIMD partners of the first half) runs ONE UNIT behind -- between barriers c and c+1 they
//          gather the last unit of channel c-1 and all but the last unit of channel c -- so that after a barrier they start
//          with tile reads whose addresses they already hold while their partners fetch table entries, and the two waves
//          of a SIMD do not sit in their start-up at the same time (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
// TW     : tile = TW x (64 / TW) positions: 8 x 8, or 32 x 2 (non-EDGE only)
template <int TX_, int TY_, int PITCH_, bool EDGE_, int SPLIT_, int SK_ = 1, int PB_ = 0, int FB_ = kFB, int NB_ = 2, int TW_ = 8>
struct GatherTraits {
    static constexpr int TX = TX_, TY = TY_, PITCH = PITCH_, SPLIT = SPLIT_, SK = SK_, PB = PB_, FB = FB_, NB = NB_;
    static constexpr int TW = TW_, TH = 64 / TW_;
    static_assert(TW_ == 8 || (TW_ == 32 && !EDGE_), "tile shapes: 8 x 8, or 32 x 2 without edge tiles");
    static constexpr bool LAGGED = NB_ == 3;
    static constexpr bool EDGE = EDGE_;
    static constexpr int kRegular = TX * TY;
    // the extra row and column of Z share one tile when they fit its 64 lanes (patches up to 24 pixels)
    static constexpr bool kMergeEdge = EDGE && (TX * 8 + TY * 8 + 1 <= 64);
    static constexpr int kPlaneTiles = kRegular + (EDGE ? (kMergeEdge ? 1 : 2) : 0);
    static constexpr int kTiles = SK * kPlaneTiles;
    static constexpr int kPerPart = (kTiles + SPLIT - 1) / SPLIT;
    static constexpr int kWaves = FB * SPLIT;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kEpiF = 2;   // output channels assembled per epilogue round
    static_assert(SK == 1 || PB > 0, "stacked planes need a compile-time plane size");
    static_assert((SK - 1) * PB + ((TY * TH - 1) * PITCH + TX * TW) * 8 < 65536, "LDS immediates are 16 bits");
};

template <int TX, int PITCH, int TW>
__device__ __forceinline__ constexpr unsigned tile_imm(int tile) {
    return (unsigned)(((tile / TX) * (64 / TW) * PITCH + (tile % TX) * TW) * 8);
}

// One unit for one part: every tile index (hence every LDS immediate) is a compile-time constant.
// Tiles go in batches of B: the ds_read_b64 of batch b+1 are issued before the MFMAs of batch b and the
// wait is counted (lgkmcnt = size of batch b+1), so one batch is always in flight.  The reads are
// inline asm on purpose: hipcc would fuse pairs into ds_read2_b64, which runs at half the LDS rate,
// and waits lgkmcnt(0) for the prefetched batch.  The sched_barrier after each wait keeps the
// register-only MFMAs from being hoisted above it.
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#ifndef DAU_GATHER_BATCH
#define DAU_GATHER_BATCH 4      // (tools/build_variant.sh ... -DDAU_GATHER_BATCH=5: timing experiments)
#endif
constexpr int kBatch = DAU_GATHER_BATCH;   // tiles per batch

// TILE: index into the wave's flat tile list = stacked plane * kPlaneTiles + tile of that plane
template <class T, int TILE>
__device__ __forceinline__ void load_tile(f2& dst, unsigned addr, unsigned addr_e0, unsigned addr_e1) {
#ifdef DAU_DIAG_NOLDS   // timing diagnosis only: no LDS tile reads (results are garbage)
    asm volatile("" : "=v"(dst) : "v"(addr), "v"(addr_e0), "v"(addr_e1));
    return;
#endif
    constexpr int plane = TILE / T::kPlaneTiles, t = TILE % T::kPlaneTiles;
    constexpr unsigned poff = (unsigned)(plane * T::PB);
    if constexpr (t < T::kRegular)
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(poff + tile_imm<T::TX, T::PITCH, T::TW>(t)) : "memory");
    else if constexpr (t == T::kRegular)
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr_e0), "n"(poff) : "memory");
    else
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr_e1), "n"(poff) : "memory");
}

template <class T, int FIRST, int COUNT, int BATCH, int... J>
__device__ __forceinline__ void load_batch(f2 (&dst)[kBatch], unsigned addr, unsigned addr_e0, unsigned addr_e1,
                                           std::integer_sequence<int, J...>) {
    (([&] {
         if constexpr (BATCH * kBatch + J < COUNT)
             load_tile<T, FIRST + BATCH * kBatch + J>(dst[J], addr, addr_e0, addr_e1);
     }()),
     ...);
}

// A group of U consecutive units (same input channel, unit indices g .. g+U-1) runs as one straight-line block:
// the batch list of all U units is flattened and the ds_read_b64 of flattened batch n+1 are issued before the MFMAs
// of batch n (counted lgkmcnt), so the LDS latency is exposed once per group instead of once per unit.
template <class T, int FIRST, int COUNT, int KP, int U, int FB>
__device__ __forceinline__ void group_batches(f4 (&acc)[KP][2], f2 (&xv)[2][kBatch], const float (&wv)[U],
                                              const unsigned (&addr)[U], const unsigned (&addr_e0)[U],
                                              const unsigned (&addr_e1)[U]) {
    constexpr int NB = (COUNT + kBatch - 1) / kBatch;
    constexpr int u = FB / NB, b = FB % NB;
    constexpr bool more = FB + 1 < U * NB;
    constexpr int nu = more ? (FB + 1) / NB : 0, nb = more ? (FB + 1) % NB : 0;
    constexpr int left = COUNT - nb * kBatch;
    constexpr int next = more ? (left < kBatch ? left : kBatch) : 0;
    if constexpr (more)
        load_batch<T, FIRST, COUNT, nb>(xv[(FB + 1) & 1], addr[nu], addr_e0[nu], addr_e1[nu], std::make_integer_sequence<int, kBatch>{});
    lds_wait<next>();
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
        const int i = b * kBatch + j;
        if (i < COUNT) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[u], xv[FB & 1][j].x, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[u], xv[FB & 1][j].y, acc[i][1], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (more) group_batches<T, FIRST, COUNT, KP, U, FB + 1>(acc, xv, wv, addr, addr_e0, addr_e1);
}

// ut_addr0 / pbase0: LDS address of this lane's slot in the FIRST unit's table entry and the plane buffer that unit reads;
// ut_addr / pbase: the same for the group's second unit, the following ones are unit_pitch bytes apart.  Ordinarily the
// first unit is just the entry before the second; for a lagged wave's first group of a channel it is the previous
// channel's last unit, in the previous channel's buffers (all four values are wave-uniform run-time numbers: one code path).
template <class T, int PART, int KP, int U>
__device__ __forceinline__ void unit_group(f4 (&acc)[KP][2], unsigned ut_addr0, unsigned pbase0, unsigned ut_addr,
                                           unsigned unit_pitch, unsigned pbase, unsigned lane_base, unsigned ebase0,
                                           unsigned ebase1) {
    constexpr int first = PART * T::kPerPart;
    constexpr int count = (first + KP <= T::kTiles) ? KP : (T::kTiles > first ? T::kTiles - first : 0);
    if constexpr (count > 0) {
        f2 wo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned ua = u == 0 ? ut_addr0 : ut_addr + (u - 1) * unit_pitch;
            asm volatile("ds_read_b64 %0, %1" : "=v"(wo[u]) : "v"(ua) : "memory");
        }
        lds_wait<0>();
        float wv[U];
        unsigned addr[U], addr_e0[U], addr_e1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            wv[u] = wo[u].x;
            // lanes 0,1 of every quad hold the plane displacement, lanes 2,3 the strip displacement: broadcast both
            const unsigned oraw = __float_as_uint(wo[u].y);
            const unsigned pb = u == 0 ? pbase0 : pbase;
            const unsigned off = (unsigned)__builtin_amdgcn_mov_dpp((int)oraw, 0x00, 0xf, 0xf, true) + pb;
            const unsigned offt = (unsigned)__builtin_amdgcn_mov_dpp((int)oraw, 0xAA, 0xf, 0xf, true) + pb;
            addr[u] = lane_base + off;
            if constexpr (T::kMergeEdge) {
                // one edge tile: ebase1 is a lane mask, set for the lanes that walk the column strip
                addr_e0[u] = ebase0 + ((off & ~ebase1) | (offt & ebase1)); addr_e1[u] = 0;
            } else {
                addr_e0[u] = ebase0 + off; addr_e1[u] = ebase1 + offt;
            }
        }
        f2 xv[2][kBatch];
        load_batch<T, first, count, 0>(xv[0], addr[0], addr_e0[0], addr_e1[0], std::make_integer_sequence<int, kBatch>{});
        group_batches<T, first, count, KP, U, 0>(acc, xv, wv, addr, addr_e0, addr_e1);
    }
}

// Whole per-wave program for one PART (the part only selects which tiles the wave owns, so that all
// LDS immediates are compile-time constants; every wave runs the same number of barriers).
template <class T, int PART>
__device__ __forceinline__ void gather_body(const GatherArgs& a, char* smem, int lane, int wave, int fi) {
    // lagged kernels: the waves of the second half (SIMD partners of the first) run one unit behind (GatherTraits::NB)
    const bool lag = T::LAGGED && wave >= T::kWaves / 2;
    constexpr int KP = T::kPerPart;
    constexpr int NB = T::NB;
    constexpr int TX = T::TX, TY = T::TY, PITCH = T::PITCH, SK = T::SK;
    constexpr bool EDGE = T::EDGE;

    // workgroup -> (group of SK consecutive (image pair, patch) planes, channel block); blocks that share planes are
    // made consecutive on one XCD (blocks b and b+8 share an XCD) so the staged planes are fetched into one L2.
    const int nblk = gridDim.x;
    int logical;
    {
        const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        const int chunk = nblk / 8, rem = nblk % 8;
        logical = (xcd < rem ? xcd * (chunk + 1) : rem * (chunk + 1) + (xcd - rem) * chunk) + idx;
    }
    const int npp0 = (logical / a.nfb) * SK, fb = logical % a.nfb;
    const int npatch = a.npx * a.npy;
    const int npp_total = ((a.N + 1) / 2) * npatch;

    // H, W: the pixels one plane covers (a patch; the whole image when it is small enough)
    const int R = a.R, H = EDGE ? TY * 8 : a.ph, W = EDGE ? TX * 8 : a.pw;
    const int ly = lane / T::TW, lx = lane % T::TW;
    const unsigned lane_base = (unsigned)(((ly + R) * PITCH + (lx + R)) * 8);
    // Edge tiles (the extra row / column of Z): tile E0 = the row y = H, x = 0..W-1, read from the plane (consecutive
    // lanes, consecutive addresses); tile E1 = the column x = W, y = 0..H, read from the column-major strip so that its
    // vertical run of positions is conflict-free as well (from the row-major plane it was an 8-way bank conflict and
    // made LDS, not the matrix pipe, the busiest resource).
    int ey[2] = {0, 0}, ex[2] = {0, 0};
    bool evalid[2] = {false, false};
    unsigned ebase0, ebase1;
    if (T::kMergeEdge) {
        // lanes 0..W-1: the row y = H; lanes W..W+H: the column x = W (from the strip); ebase1 = mask of the latter
        const bool col = lane >= W;
        if (!col) { ey[0] = H; ex[0] = lane; evalid[0] = true; }
        else if (lane <= W + H) { ey[0] = lane - W; ex[0] = W; evalid[0] = true; }
        ebase0 = col ? (unsigned)(a.strip_off + (ey[0] + R) * 8) : (unsigned)(((ey[0] + R) * PITCH + ex[0] + R) * 8);
        ebase1 = col ? 0xffffffffu : 0u;
    } else {
        if (EDGE) {
            if (lane < W) { ey[0] = H; ex[0] = lane; evalid[0] = true; }
            if (lane <= H) { ey[1] = lane; ex[1] = W; evalid[1] = true; }
        }
        ebase0 = (unsigned)(((ey[0] + R) * PITCH + ex[0] + R) * 8);
        ebase1 = (unsigned)(a.strip_off + (ey[1] + R) * 8);
    }

    const unsigned plane_bytes = T::PB ? (unsigned)T::PB : a.plane_bytes, ut_stride = a.ut_stride;
    const unsigned buf_bytes = SK * plane_bytes;
    const char* src_units = a.packed + (size_t)fb * a.Cin * ut_stride;
    const unsigned ut_base = NB * buf_bytes;

    auto issue = [&](int c, int buf) {
        const unsigned pieces = plane_bytes >> 10;
#pragma unroll
        for (int k = 0; k < SK; ++k) {
            // the last group may be short: its missing planes re-read the last valid one (their output is not stored)
            const int npp = npp0 + k < npp_total ? npp0 + k : npp_total - 1;
            const char* ps = a.staged + ((size_t)npp * a.Cin + c) * plane_bytes;
            for (unsigned piece = wave; piece < pieces; piece += T::kWaves)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(ps + (size_t)piece * 1024 + lane * 16),
                                                 (lds_ptr_t)(smem + buf * buf_bytes + k * plane_bytes + piece * 1024), 16, 0, 0);
        }
        const char* us = src_units + (size_t)c * ut_stride;
        const unsigned upieces = ut_stride >> 10;
        for (unsigned piece = wave; piece < upieces; piece += T::kWaves)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(us + (size_t)piece * 1024 + lane * 16),
                                             (lds_ptr_t)(smem + ut_base + buf * ut_stride + piece * 1024), 16, 0, 0);
    };

    f4 acc[KP][2];
#pragma unroll
    for (int i = 0; i < KP; ++i) { acc[i][0] = f4{0, 0, 0, 0}; acc[i][1] = f4{0, 0, 0, 0}; }

    issue(0, 0);
    int buf = 0, pbuf = NB - 1;        // buffers of channel c and of channel c - 1
    for (int c = 0; c < a.Cin; ++c, pbuf = buf, buf = (buf + 1 == NB ? 0 : buf + 1)) {
        const int nbuf = buf + 1 == NB ? 0 : buf + 1;
#ifndef DAU_DIAG_NOBARRIER
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // plane c is in LDS for everyone; everyone is done reading plane c-1 (lagged kernels: c-2)
#endif
#ifdef DAU_DIAG_FUSED_BLUR
#include "diag_fused_blur.inc"   // timing diagnosis of DESIGN.md 5.3 (results are garbage); tools/ab_fused_blur.sh
#else
        if (c + 1 < a.Cin && !((a.debug & 1) && c >= 1)) issue(c + 1, nbuf);
#endif
        const unsigned pbase = buf * buf_bytes;
        const unsigned ut_addr = ut_base + buf * ut_stride + (fi * kUnitDwords + (lane & 3) * 2) * 4;
        constexpr unsigned unit_pitch = T::FB * kUnitDwords * 4;
        // slots this wave's output channel uses in this input channel: all G, or (window pass) the slice's count word
        int ng = a.G;
        if (a.binned) {
            unsigned cw;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(cw) : "v"(ut_base + buf * ut_stride + (unsigned)(a.G * T::FB * kUnitDwords + fi) * 4) : "memory");
            ng = __builtin_amdgcn_readfirstlane((int)cw);
        }
        // this interval's units: the first one (a0, p0), then consecutive entries of THIS channel from `rest` on
        int n = ng;
        unsigned a0 = ut_addr, p0 = pbase, rest = ut_addr + unit_pitch;
        if (lag) {
            if (c == 0) {
                n = ng - 1;                    // units 0 .. ng-2; the last one waits for the next interval
            } else {                           // [last unit of channel c-1] + units 0 .. ng-2 of channel c
                a0 = ut_base + pbuf * ut_stride + (fi * kUnitDwords + (lane & 3) * 2) * 4 + (ng - 1) * unit_pitch;
                p0 = pbuf * buf_bytes;
                rest = ut_addr;
            }
        }
        int g = 0;
        for (; g + 4 <= n; g += 4) {
            unit_group<T, PART, KP, 4>(acc, a0, p0, rest, unit_pitch, pbase, lane_base, ebase0, ebase1);
            a0 = rest + 3 * unit_pitch; p0 = pbase; rest = a0 + unit_pitch;
        }
        if (g + 2 <= n) {
            unit_group<T, PART, KP, 2>(acc, a0, p0, rest, unit_pitch, pbase, lane_base, ebase0, ebase1);
            g += 2; a0 = rest + unit_pitch; p0 = pbase; rest = a0 + unit_pitch;
        }
        if (g < n)
            unit_group<T, PART, KP, 1>(acc, a0, p0, rest, unit_pitch, pbase, lane_base, ebase0, ebase1);
    }
    if (lag) {
        // the last unit of the last channel (its plane stays in LDS: the partners wait at the epilogue's first barrier)
        constexpr unsigned unit_pitch = T::FB * kUnitDwords * 4;
        const unsigned a0 = ut_base + pbuf * ut_stride + (fi * kUnitDwords + (lane & 3) * 2) * 4 + (a.G - 1) * unit_pitch;
        unit_group<T, PART, KP, 1>(acc, a0, pbuf * buf_bytes, a0, unit_pitch, pbuf * buf_bytes, lane_base, ebase0, ebase1);
    }

    // ---- epilogue: out[p] = Z0[p] + Z1[p+(0,1)] + Z2[p+(1,0)] + Z3[p+(1,1)] through LDS ---------------
    // per round: one image of the pair, kEpiF output channels, all SK planes
    const unsigned zpitch = a.zpitch;
    const unsigned zplane = (unsigned)(H + 1) * zpitch;       // floats per tap plane
    const unsigned zchan = 4 * zplane;                        // floats per (plane, output channel)
    float* zs = reinterpret_cast<float*>(smem);
    const int HW = H * W;
    const long plane_out = (long)a.H * a.W;
#pragma unroll
    for (int img = 0; img < 2; ++img) {   // unrolled: acc[i][img] must be a static register index
#pragma unroll 1
        for (int fh = 0; fh < T::FB / T::kEpiF; ++fh) {
            __syncthreads();
            if (fi / T::kEpiF == fh) {
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    constexpr int first = PART * KP;
                    const int flat = first + i;
                    const int k = flat / T::kPlaneTiles, tile = flat % T::kPlaneTiles;
                    int y, x; bool ok;
                    if (flat >= T::kTiles) { y = 0; x = 0; ok = false; }
                    else if (tile < T::kRegular) { y = (tile / TX) * T::TH + ly; x = (tile % TX) * T::TW + lx; ok = (y <= H) && (x <= W); }
                    else if (tile == T::kRegular) { y = ey[0]; x = ex[0]; ok = evalid[0]; }
                    else { y = ey[1]; x = ex[1]; ok = evalid[1]; }
                    if (ok) {
                        const f4 v = acc[i][img];
                        float* q = zs + (size_t)(k * T::kEpiF + fi % T::kEpiF) * zchan + (unsigned)y * zpitch + x;
                        q[0] = v[0]; q[zplane] = v[1]; q[2 * zplane] = v[2]; q[3 * zplane] = v[3];
                    }
                }
            }
            __syncthreads();
            for (int o = threadIdx.x; o < SK * T::kEpiF * HW; o += T::kThreads) {
                const int kf = o / HW;                     // (plane, channel of the round)
                const int k = kf / T::kEpiF, fl = kf % T::kEpiF;
                const int p = o % HW, y = p / W, x = p % W;
                const int f = fb * T::FB + fh * T::kEpiF + fl;
                const int npp = npp0 + k;
                const int n = 2 * (npp / npatch) + img, patch = npp % npatch;
                const int gy = (patch / a.npx) * H + y, gx = (patch % a.npx) * W + x;
                const float* zf = zs + (size_t)kf * zchan + (unsigned)y * zpitch + x;
                const float v = zf[0] + zf[zplane + 1] + zf[2 * zplane + zpitch] + zf[3 * zplane + zpitch + 1];
                if (npp < npp_total && n < a.N && f < a.Cout && gy < a.H && gx < a.W)
                    store_act(a.out, ((long)n * a.Cout + f) * plane_out + (long)gy * a.W + gx, v, a.bf16 != 0, a.accumulate != 0);
            }
        }
    }
}

template <class T>
__global__ void __launch_bounds__(T::kThreads) gather_mfma_kernel(const GatherArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!guard_pass(a.guard)) return;
    constexpr int SPLIT = T::SPLIT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int part = wave % SPLIT, fi = wave / SPLIT;
    if (SPLIT == 1 || part == 0) gather_body<T, 0>(a, smem, lane, wave, fi);
    else if (SPLIT == 2 || part == 1) { if constexpr (SPLIT > 1) gather_body<T, 1>(a, smem, lane, wave, fi); }
    else if (SPLIT == 3 || part == 2) { if constexpr (SPLIT > 2) gather_body<T, 2>(a, smem, lane, wave, fi); }
    else { if constexpr (SPLIT > 3) gather_body<T, 3>(a, smem, lane, wave, fi); }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

// a == nullptr: raise the kernel's dynamic-LDS limit (once per plan and device, tiled_gather_init); else launch
template <class T>
void launch_variant(hipStream_t st, const GatherArgs* a, int grid, size_t lds) {
    auto kern = gather_mfma_kernel<T>;
    if (!a) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::kThreads), lds, st, *a);
}

void dispatch_variant(int variant, hipStream_t st, const GatherArgs* a, int grid, size_t lds) {
    switch (variant) {
        case 0: launch_variant<GatherTraits<7, 7, 72, true, 2>>(st, a, grid, lds); break;
        case 1: launch_variant<GatherTraits<7, 7, 104, true, 2>>(st, a, grid, lds); break;
        case 2: launch_variant<GatherTraits<4, 4, 72, true, 1>>(st, a, grid, lds); break;
        case 3: launch_variant<GatherTraits<2, 2, 40, true, 1>>(st, a, grid, lds); break;
        case 4: launch_variant<GatherTraits<3, 3, 40, true, 1>>(st, a, grid, lds); break;
        case 5: launch_variant<GatherTraits<1, 1, 40, true, 1>>(st, a, grid, lds); break;
        case 6: launch_variant<GatherTraits<4, 4, 40, false, 1>>(st, a, grid, lds); break;
#ifdef DAU_TUNING
        case 7: launch_variant<GatherTraits<7, 7, 72, true, 3>>(st, a, grid, lds); break;
#endif
        case 8: launch_variant<GatherTraits<4, 4, 72, true, 2, 2, 26624>>(st, a, grid, lds); break;
        case 9: launch_variant<GatherTraits<3, 3, 40, true, 2, 4, 13312>>(st, a, grid, lds); break;
        case 10: launch_variant<GatherTraits<2, 2, 40, true, 1, 4, 10240, 8>>(st, a, grid, lds); break;
        case 11: launch_variant<GatherTraits<1, 1, 40, true, 1, 4, 7168, 16>>(st, a, grid, lds); break;
        case 12: launch_variant<GatherTraits<4, 4, 40, false, 2, 3, 13312>>(st, a, grid, lds); break;
        case 13: launch_variant<GatherTraits<3, 3, 40, false, 1, 2, 10240, 8>>(st, a, grid, lds); break;
        case 14: launch_variant<GatherTraits<2, 2, 40, false, 1, 4, 8192, 8>>(st, a, grid, lds); break;
        case 15: launch_variant<GatherTraits<1, 1, 40, false, 1, 8, 5120, 16>>(st, a, grid, lds); break;
        case 16: launch_variant<GatherTraits<4, 4, 72, false, 2>>(st, a, grid, lds); break;
        case 17: launch_variant<GatherTraits<4, 4, 104, false, 1, 1, 0, 8>>(st, a, grid, lds); break;
        case 18: launch_variant<GatherTraits<4, 4, 40, false, 1, 1, 0, 8>>(st, a, grid, lds); break;
#ifdef DAU_TUNING
        case 19: launch_variant<GatherTraits<4, 4, 40, false, 2, 2, 13312, 4>>(st, a, grid, lds); break;
#endif
#ifdef DAU_TUNING
        case 20: launch_variant<GatherTraits<7, 7, 72, true, 2, 1, 0, 4, 3>>(st, a, grid, lds); break;
#endif
        case 21: launch_variant<GatherTraits<4, 4, 40, false, 1, 1, 0, 12>>(st, a, grid, lds); break;
        case 22: launch_variant<GatherTraits<4, 4, 104, false, 1, 1, 0, 12>>(st, a, grid, lds); break;
        case 23: launch_variant<GatherTraits<4, 4, 72, false, 1, 1, 0, 12>>(st, a, grid, lds); break;
#ifdef DAU_TUNING                // explicit-request rows (Variant::tuning != 0) exist in the tuning build only
        case 24: launch_variant<GatherTraits<1, 15, 40, false, 1, 1, 0, 12, 2, 32>>(st, a, grid, lds); break;
        case 25: launch_variant<GatherTraits<1, 14, 40, false, 1, 1, 0, 12, 2, 32>>(st, a, grid, lds); break;
        case 26: launch_variant<GatherTraits<4, 4, 40, false, 1, 1, 0, 12, 3>>(st, a, grid, lds); break;
        case 27: launch_variant<GatherTraits<4, 4, 72, false, 1, 1, 0, 12, 3>>(st, a, grid, lds); break;
#endif
        default: break;
    }
}

auto blur_pack_for(int blur_k) {
    // sigma = 0.5 (the reference's default) gives a 7-tap prefilter; other supports take the generic instantiation
    return blur_k == 7 ? blur_pack_kernel<7> : blur_k == 5 ? blur_pack_kernel<5> : blur_k == 9 ? blur_pack_kernel<9> : blur_pack_kernel<0>;
}

// largest window any patch (band of band_rows staged rows) needs: raw [lh][lw] + horizontally filtered [lh][bw]
size_t blur_pack_lds_bytes(const Geometry& g, int k, int band_rows = 0) {
    const int rows = band_rows > 0 && band_rows < g.rows ? band_rows : g.rows;
    const size_t wh = rows < g.H ? rows : g.H, ww = g.cols < g.W ? g.cols : g.W;
    const size_t strip = g.edge ? (size_t)(2 * g.Rt + 1) * (rows | 1) : 0;      // column-major copy of the strip columns
    return ((wh + k - 1) * (ww + k - 1) + (wh + k - 1) * ww + strip) * 8;
}

// planes that need more than 80 KiB of LDS (two workgroups per CU) are staged in two (or more) row bands.  (Round 1 found
// 40 KiB bands faster; with the loads of a workgroup in flight together the whole plane per workgroup is: 56-pixel planes
// 382 -> 350 us, the 68 x 104 planes of bucket 18 1784 -> 1594 us -- every band re-reads its blur halo.)
int blur_pack_bands(const Geometry& g, int k) {
    int bands = 1;
    static const size_t limit = (size_t)DAU_TUNE_INT("DAU_BLUR_LDS_KB", 80) * 1024;
    while (bands < 8 && blur_pack_lds_bytes(g, k, (g.rows + bands - 1) / bands) > limit) ++bands;
    return bands;
}

size_t lds_bytes(const TiledConfig& c, const Geometry& g) {
    const size_t main_b = (size_t)g.nb * g.sk * g.plane_bytes + g.nb * ut_stride_bytes(c.G, g.fb, g.nwin1 > 1);
    const size_t zpitch = g.pw + 2;
    const size_t epi_b = (size_t)g.sk * 2 /*kEpiF*/ * 4 * (g.ph + 1) * zpitch * 4;
    return main_b > epi_b ? main_b : epi_b;
}

}  // namespace

bool tiled_gather_configure(int N, int Cin, int Cout, int G, int H, int W, int R, int blur_k, bool bf16, TiledConfig* cfg) {
    const Geometry g = make_geometry(H, W, R, G, N, Cout);
    if (g.variant < 0) return false;
    TiledConfig c{};
    c.N = N; c.Cin = Cin; c.Cout = Cout; c.G = G; c.H = H; c.W = W; c.R = R; c.blur_k = blur_k;
    c.NP = (N + 1) / 2;
    c.rows = g.rows; c.pitch = g.pitch; c.tiles_x = g.tx; c.tiles_y = g.ty; c.tile_w = g.tw; c.fblock = g.fb; c.variant = g.variant;
    c.patches = g.npx * g.npy; c.stack = g.sk; c.windows = g.nwin1 * g.nwin1;
    c.debug = DAU_TUNE_INT("DAU_GATHER_DEBUG", 0);
    c.bf16 = bf16 ? 1 : 0;
    if (lds_bytes(c, g) > 160 * 1024) return false;
    // blur_pack keeps both raw planes (+ blur halo) and the horizontally filtered rows in LDS
    if (blur_pack_lds_bytes(g, blur_k, (g.rows + blur_pack_bands(g, blur_k) - 1) / blur_pack_bands(g, blur_k)) > 150 * 1024) return false;
    *cfg = c;
    return true;
}

size_t tiled_gather_workspace_bytes(const TiledConfig& c) {
    const Geometry g = make_geometry(c.H, c.W, c.R, c.G, c.N, c.Cout, c.variant);
    const size_t nfb = (c.Cout + g.fb - 1) / g.fb;
    return round_up((size_t)c.NP * c.patches * c.Cin * g.plane_bytes, 256) + round_up(nfb * c.Cin * ut_stride_bytes(c.G, g.fb, g.nwin1 > 1), 256);
}

void tiled_gather_init(const TiledConfig& c) {
    dispatch_variant(c.variant, nullptr, nullptr, 0, 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blur_pack_for(c.blur_k)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

int tiled_gather_windows(const TiledConfig& c) { return c.windows; }

void tiled_gather_prepare(hipStream_t st, const TiledConfig& c, const float* in, const float* filters, bool mirrored,
                          const UnitRef* table, void* workspace, int window, const Guard& guard) {
    const Geometry g = make_geometry(c.H, c.W, c.R, c.G, c.N, c.Cout, c.variant);
    char* staged = static_cast<char*>(workspace);
    char* packed = staged + round_up((size_t)c.NP * c.patches * c.Cin * g.plane_bytes, 256);
    const int bands = blur_pack_bands(g, c.blur_k);
    const int band_rows = (g.rows + bands - 1) / bands;
    const size_t blur_lds = blur_pack_lds_bytes(g, c.blur_k, band_rows);
    auto kern = blur_pack_for(c.blur_k);
    BlurPackArgs b{};
    b.guard = guard;
    b.in = in; b.taps = filters + kTaps1dOffset; b.staged = reinterpret_cast<float*>(staged);
    // centre of this pass's offset window: staged (row, col) of a patch is the image at (py*ph - Rt + cy + row, ...)
    const int cy = -c.R + g.Rt + 2 * g.Rt * (window / g.nwin1), cx = -c.R + g.Rt + 2 * g.Rt * (window % g.nwin1);
    b.mirrored = mirrored ? 1 : 0; b.N = c.N; b.C = c.Cin; b.H = c.H; b.W = c.W; b.R = g.Rt; b.k = c.blur_k;
    b.cy = cy; b.cx = cx; b.bf16 = c.bf16;
    b.ph = g.ph; b.pw = g.pw; b.npx = g.npx; b.npy = g.npy;
    b.rows = g.rows; b.pitch = g.pitch; b.cols = g.cols; b.strip_cols = g.edge ? 2 * g.Rt + 1 : 0;
    b.plane_floats = g.plane_bytes / 4;
    // small planes: several per workgroup, so that the 512 threads have rows to share (7x7 maps: 8 planes)
    const int elems = g.rows * g.pitch;
    b.ppb = elems >= 4096 ? 1 : elems >= 2048 ? 2 : elems >= 1024 ? 4 : 8;
    while (b.ppb > 1 && b.ppb * blur_lds > 64 * 1024) b.ppb /= 2;
    b.bands = bands; b.band_rows = band_rows;
    if (bands > 1) b.ppb = 1;
    b.planes = c.NP * c.patches * c.Cin * bands;
    b.lds_plane_floats = (unsigned)(blur_lds / 4);
    static const int blur_threads = DAU_TUNE_INT("DAU_BLUR_THREADS", 512);
    hipLaunchKernelGGL(kern, dim3((b.planes + b.ppb - 1) / b.ppb), dim3(b.ppb > 1 ? 512 : blur_threads), b.ppb * blur_lds, st, b);
    const int nfb = (c.Cout + g.fb - 1) / g.fb;
    const bool binned = g.nwin1 > 1;
    const size_t uts = ut_stride_bytes(c.G, g.fb, binned);
    if (binned) {
        // slots beyond a channel's count and the KiB padding are never read
        const long total = (long)nfb * c.Cin * g.fb;
        const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
        hipLaunchKernelGGL(pack_units_binned_kernel, dim3(grid), dim3(256), 0, st, table, c.Cin, c.G, c.Cout, g.pitch, c.R, g.Rt,
                           g.nwin1, window, g.strip_pitch, (int)(uts / 4), g.fb, reinterpret_cast<unsigned int*>(packed), guard);
        return;
    }
    // Packed slices are padded to whole KiB; the padding is zeroed together with the payload.  The memset is not guarded:
    // the candidate bucket sets of a call run one after the other in the same workspace, so it is harmless for the other.
    (void)hipMemsetAsync(packed, 0, (size_t)nfb * c.Cin * uts, st);
    const long total = (long)nfb * c.Cin * c.G * g.fb;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(pack_units_kernel, dim3(grid), dim3(256), 0, st, table, c.Cin, c.G, c.Cout, g.pitch, c.R, g.Rt, g.nwin1,
                       window, g.strip_pitch,
                       (int)(uts / 4), g.fb,
                       reinterpret_cast<unsigned int*>(packed), guard);
}

void tiled_gather_run(hipStream_t st, const TiledConfig& c, float* out, void* workspace, bool accumulate, const Guard& guard) {
    const Geometry g = make_geometry(c.H, c.W, c.R, c.G, c.N, c.Cout, c.variant);
    GatherArgs a{};
    a.staged = static_cast<const char*>(workspace);
    a.packed = a.staged + round_up((size_t)c.NP * c.patches * c.Cin * g.plane_bytes, 256);
    a.out = out;
    a.npx = g.npx; a.npy = g.npy; a.ph = g.ph; a.pw = g.pw;
    a.N = c.N; a.Cin = c.Cin; a.Cout = c.Cout; a.G = c.G; a.H = c.H; a.W = c.W; a.R = g.Rt;
    a.accumulate = accumulate ? 1 : 0; a.bf16 = c.bf16;
    a.nfb = (c.Cout + g.fb - 1) / g.fb;
    a.plane_bytes = (unsigned)g.plane_bytes;
    a.strip_off = (unsigned)g.strip_off;
    a.binned = g.nwin1 > 1 ? 1 : 0;
    a.guard = guard;
    a.ut_stride = (unsigned)ut_stride_bytes(c.G, g.fb, g.nwin1 > 1);
    a.zpitch = (unsigned)(g.pw + 2);
    a.debug = c.debug;
    const int grid = ((c.NP * c.patches + g.sk - 1) / g.sk) * a.nfb;
    const size_t lds = lds_bytes(c, g);
    dispatch_variant(c.variant, st, &a, grid, lds);
}

}  // namespace dau
