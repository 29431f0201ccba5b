// Densified gather-sum at fp32 accuracy on the f16 matrix cores: two-limb ("split") operands.
//
//   out[n,f,y,x] = sum_{c} sum_{ty,tx < K} Wd[f][c][ty][tx] * Xb[n,c, y+ty-R, x+tx-R]          (K = 2R+1, offsets within +-R)
//
// is the dense form of k_dense_bf16.hip -- the G units of every (input channel, output channel) pair scattered into one K x K
// kernel, the pass an implicit GEMM  M = output channels, N = pixels, K = input channels x taps -- but with both operands kept
// to fp32 accuracy: every fp32 value v (blurred activation, dense tap) is scaled by a power of two and split into two
// binary16 limbs  hi = f16(v), lo = f16(v - hi)  (22 significant bits together), and the product is accumulated as
//      hi_w * hi_x  +  lo_w * hi_x  +  hi_w * lo_x          (the dropped lo * lo term is 2^-22 of the product)
// -- three v_mfma_f32_32x32x16_f16 per tap and 16 input channels, products exact, fp32 accumulation.  At radius 3 that is
// 3 * 49 / 16 = 9.2 fp32-rate MACs per (pixel, channel pair) against the 4 G = 16 of the exact gather at four units
// (k_gather_mfma.hip, v_mfma_f32_4x4x1 at the fp32 vector rate), inside the SAME parity bar (1e-4 relative + 1e-6 of the
// max-norm against the oracle: tests/test_gpu_dense_split.py records the margins).  It replaces the same reference code as
// the gather: DAUConv_forward_pipeline_kernel + interleave_input_data_kernel + perpare_weights_and_offsets
// (include/dau_conv/dau_conv_impl/dau_conv_forward_core.hpp:804-1605, 1607-1732, 1858-2215) and caffe_gpu_convolve2
// (src/dau_conv/util/convolve.cu:48-131), for calls whose offsets lie within +-R (the call's device guard decides).
//
// Power-of-two scales (binary16 has five exponent bits): sx brings max|x| to [2^13, 2^14), sw the bound G * max|w| of a dense
// tap likewise; both come from a two-launch max reduction over the pass's input and its unit table, the epilogue multiplies
// by 1 / (sx * sw).  Scaling by a power of two is exact, so the only effect is that the limbs cannot overflow and that values
// down to 2^-17 of the maximum keep all 22 bits (smaller ones: an absolute error below 2^-39 of the maximum).
//
// Layouts (HBM, all in the pass's workspace):
//   header      SplitScales (max bits, scales)
//   XS[n][chunk][limb][half][Hs][Ws][8]  f16: Gaussian-blurred, scaled input; 16 input channels per chunk as two halves of 8
//       (one 16-byte unit per position = the B fragment of one lane), limb 0 = hi, 1 = lo; staged position (r, c) = image
//       (r-R, c-R), zero outside the image.
//   WS[chunk][tap][limb][CoutP][16]      f16: the dense kernel (A fragments of the two lane halves).
// Workgroup = 128 output channels x 8 rows x NSUB*8 columns as 8 waves = 4 (32 channels) x 2 (4 rows), two per SIMD; a wave
// owns NSUB <= 4 tiles.  Per chunk the window of the four (limb, half) planes is copied into LDS by global_load_lds
// (double buffered, no registers); per tap a wave reads NSUB hi and NSUB lo B fragments (ds_read_b128) and two A fragments
// (global memory, two taps ahead) and issues 3 * NSUB MFMAs.
//
// Accumulation is hierarchical.  v_mfma_f32_32x32x16_f16 aligns its sixteen products to the accumulator's exponent and drops
// what falls more than about two bits below its last place (tools/microbench/mfma_f16_accum: sixteen products of 1/16 ulp each
// vanish), so a long chain through one accumulator loses bits of the RUNNING SUM with every instruction: 2352 chained MFMAs at
// S = 256 measured 3.9e-6 of the max-norm against the oracle, four times the exact gather.  So a tile has TWO accumulators:
// kFlushRows rows of taps (4 x 21 MFMAs) are chained into `part`, and `part` is added to the tile's running sum by v_pk_add_f32
// (round to nearest).  Both ends lose: every external add costs half an ulp of the running sum, every chained MFMA an ulp of the
// chain -- measured max error / max-norm of y at S = F = 256 (and at S = F = 512, 28 x 28) per chain length:
//   3 MFMAs 1.5e-6 (2.5e-6), 9: 8.7e-7 (1.3e-6), 21 (one row): 5.8e-7 (7.9e-7), 42: 4.3e-7 (5.9e-7), 84: 4.0e-7 (4.5e-7),
//   147 (one chunk): 5.1e-7 (4.9e-7), 294: 8.3e-7 (6.4e-7)                        -- profiles/r4_ab_split_chain_length.txt
// at the same speed.  Two accumulators of 16 registers per tile is why a wave owns at most four tiles (a 56-pixel row is a block of
// four and a block of three).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "dau_tiled.hpp"

#ifndef DAU_SPLIT_R
#define DAU_SPLIT_R 3
#endif
#ifndef DAU_SPLIT_NS
#define DAU_SPLIT_NS s3
#endif

namespace dau {
namespace DAU_SPLIT_NS {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

namespace {

constexpr int kDR = DAU_SPLIT_R;        // offset radius of the form
constexpr int kDK = 2 * kDR + 1;        // taps per axis (an offset of exactly +R has fraction 0: the tap at R + 1 carries weight 0)
constexpr int kDSpan = kDK - 1;
constexpr int kDTaps = kDK * kDK;
constexpr int kDRows = 8;               // output rows per workgroup
constexpr int kDFB = 128;               // output channels per workgroup
constexpr int kAhead = 2;               // taps the A stream runs ahead
#ifndef DAU_SPLIT_FLUSH_TAPS
#define DAU_SPLIT_FLUSH_TAPS 0          // 0: a whole row of taps per chain
#endif
constexpr int kFlushTaps = DAU_SPLIT_FLUSH_TAPS > 0 ? DAU_SPLIT_FLUSH_TAPS : kDK;   // taps chained through `part` before it joins the running sum
#ifndef DAU_SPLIT_FLUSH_ROWS
#define DAU_SPLIT_FLUSH_ROWS 4          // rows of taps per chain (> 1: the chain runs across rows and chunks, `part` is zeroed by moves)
#endif
constexpr int kFlushRows = DAU_SPLIT_FLUSH_ROWS;

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// device-side scales of one pass (head of its workspace)
struct SplitScales {
    unsigned max_x_bits, max_w_bits;    // float bits of max|input| and of max over the units of |w| (sum of the four tap weights)
    float sx, sw, inv;                  // power-of-two scales of the activations / the dense taps, 1 / (sx * sw)
    float pad[3];
};
constexpr int kPartials = 1024;         // workgroups of the max reduction

constexpr int kMaxSub = 4;              // 8-pixel tiles per column block (two accumulators per tile: see the header)
struct SplitGeom {
    int sub;                 // 8-pixel tiles per row
    // column blocks: nb_a blocks of nsub_a tiles from column 0, then nb_b blocks of nsub_b = nsub_a - 1 tiles (0 blocks when even)
    int nsub_a, nb_a, nsub_b, nb_b;
    // row blocks: nrb8 blocks of eight rows from row 0, then -- where 1 .. 4 rows are left -- one block of four (nrb4 = 1)
    int nrb8, nrb4;
    int tall;                // > 0: the eight-row blocks run as ONE column block of `tall` tiles of 8 rows x 4 columns (5 or 7)
    int Hs, Ws, nchunk, CoutP;
    size_t hdr_bytes, xs_bytes, ws_bytes;
};

SplitGeom split_geometry(const DenseConfig& c) {
    SplitGeom g{};
    g.sub = (c.W + 7) / 8;
    const int nblocks = (g.sub + kMaxSub - 1) / kMaxSub, base = g.sub / nblocks, rem = g.sub % nblocks;
    if (rem) { g.nsub_a = base + 1; g.nb_a = rem; g.nsub_b = base; g.nb_b = nblocks - rem; }
    else { g.nsub_a = base; g.nb_a = nblocks; g.nsub_b = 0; g.nb_b = 0; }
    const int left = c.H % kDRows;
    // the block of four rows is a launch of its own: it pays where the launches are many rounds of workgroups long (C3, 2048 workgroups:
    // gather 6.6 -> 6.1 ms per pass), not where a pass is two rounds and the split makes it three (C1, 512 workgroups: 0.34 -> 0.41 ms)
    const long wgs = (long)c.N * ((c.H + kDRows - 1) / kDRows) * nblocks * ((c.Cout + kDFB - 1) / kDFB);
    const int rows4 = DAU_TUNE_INT("DAU_SPLIT_ROWS4", 1);    // tuning build: 0 never, 2 always (the variant tests)
    g.nrb4 = (left >= 1 && left <= 4 && rows4 != 0 && (wgs >= 1024 || rows4 == 2)) ? 1 : 0;
    g.nrb8 = c.H / kDRows + (left && !g.nrb4 ? 1 : 0);
    g.Hs = g.nrb8 * kDRows + g.nrb4 * 4 + kDSpan;
    {
        // tall tiles where they compute at least 5 % fewer tile positions than the 4 x 8 ones (whose waves skip dead groups of four rows):
        // 28 x 28 with a block of four rows: 24 x 28 + 4 x 32 against 28 x 32
        const int t4 = (c.W + 3) / 4;
        const long rows8 = (long)g.nrb8 * kDRows;
        const long flat = (long)(g.nrb4 ? rows8 + 4 : (c.H + 3) / 4 * 4) * g.sub * 8;
        const long tall = rows8 * t4 * 4 + (g.nrb4 ? 4L * g.sub * 8 : 0);
        const int want = DAU_TUNE_INT("DAU_SPLIT_TALL", 1);  // tuning build: 0 never, 2 wherever the width allows (the variant tests)
        g.tall = (want != 0 && g.nrb8 > 0 && (t4 == 5 || t4 == 7) && (tall * 20 <= flat * 19 || want == 2)) ? t4 : 0;
    }
    g.Ws = g.sub * 8 + kDSpan;
    g.nchunk = (c.Cin + 15) / 16;
    g.CoutP = (int)round_up(c.Cout, kDFB);
    g.hdr_bytes = round_up(sizeof(SplitScales) + (size_t)2 * kPartials * sizeof(unsigned), 256);
    // + 4 KiB: the window copy reads whole LDS-pitch rows, i.e. a few units past the last plane's last row
    g.xs_bytes = round_up((size_t)c.N * g.nchunk * 4 * g.Hs * g.Ws * 16 + 4096, 256);
    g.ws_bytes = round_up(((size_t)g.nchunk * kDTaps + 8) * 2 * g.CoutP * 32, 256);   // + look-ahead taps of the A stream
    return g;
}

constexpr int lds_pitch(int nsub) {       // positions; = 8 (mod 16) and >= nsub*8 + span: the four rows of a B fragment hit different banks
    int p = nsub * 8 + kDSpan;
    while (p % 16 != 8) ++p;
    return p;
}
// narrow tiles (8 rows x 4 columns): sixteen lanes read four columns of four rows -- conflict free at a pitch = 4 or 12 (mod 16)
constexpr int lds_pitch_narrow(int ntiles) {
    int p = ntiles * 4 + kDSpan;
    while (p % 16 != 4 && p % 16 != 12) ++p;
    return p;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// scales: max|input| and max|w| (two launches: per-workgroup maxima, then one workgroup reduces them and derives the scales)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) split_absmax_kernel(const float* __restrict__ in, long count, int bf16, const UnitRef* __restrict__ table,
                                                           long units, unsigned* __restrict__ partial, const Guard guard) {
    if (!guard_pass(guard)) return;
    unsigned mx = 0, mw = 0;
    const long stride = (long)gridDim.x * blockDim.x, t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (bf16) {
        const uint2* p = reinterpret_cast<const uint2*>(in);          // four bf16 per load
        const long n4 = reinterpret_cast<uintptr_t>(in) % 8 == 0 ? count / 4 : 0;
        for (long i = t0; i < n4; i += stride) {
            const uint2 v = p[i];
            mx = max(mx, max(max((v.x << 16) & 0x7fffffffu, v.x & 0x7fff0000u), max((v.y << 16) & 0x7fffffffu, v.y & 0x7fff0000u)));
        }
        for (long i = n4 * 4 + t0; i < count; i += stride) mx = max(mx, ((unsigned)reinterpret_cast<const unsigned short*>(in)[i] << 16) & 0x7fffffffu);
    } else {
        const uint4* p = reinterpret_cast<const uint4*>(in);
        const long n4 = reinterpret_cast<uintptr_t>(in) % 16 == 0 ? count / 4 : 0;
        for (long i = t0; i < n4; i += stride) {
            const uint4 v = p[i];
            mx = max(mx, max(max(v.x & 0x7fffffffu, v.y & 0x7fffffffu), max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
        }
        for (long i = n4 * 4 + t0; i < count; i += stride) mx = max(mx, __float_as_uint(in[i]) & 0x7fffffffu);
    }
    for (long u = t0; u < units; u += stride) {
        const UnitRef r = table[u];
        mw = max(mw, __float_as_uint(fabsf(r.w00) + fabsf(r.w01) + fabsf(r.w10) + fabsf(r.w11)));
    }
    __shared__ unsigned sx[4], sw[4];
    for (int m = 32; m >= 1; m >>= 1) { mx = max(mx, (unsigned)__shfl_xor((int)mx, m)); mw = max(mw, (unsigned)__shfl_xor((int)mw, m)); }
    if ((threadIdx.x & 63) == 0) { sx[threadIdx.x >> 6] = mx; sw[threadIdx.x >> 6] = mw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = max(max(sx[0], sx[1]), max(sx[2], sx[3]));
        partial[kPartials + blockIdx.x] = max(max(sw[0], sw[1]), max(sw[2], sw[3]));
    }
}

// 2^(13 - floor(log2 m)) for a finite m > 0 (m * scale in [2^13, 2^14)), 1 otherwise (zeros; Inf / NaN pass through the arithmetic)
__device__ __forceinline__ float limb_scale(float m) {
    if (!(m > 0.0f) || !(m < INFINITY)) return 1.0f;
    int e = ilogbf(m);
    e = 13 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.0f, e);
}

__global__ void __launch_bounds__(256) split_scales_kernel(const unsigned* __restrict__ partial, int nparts, int G, SplitScales* __restrict__ out,
                                                           const Guard guard) {
    if (!guard_pass(guard)) return;
    unsigned mx = 0, mw = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) { mx = max(mx, partial[i]); mw = max(mw, partial[kPartials + i]); }
    __shared__ unsigned sx[4], sw[4];
    for (int m = 32; m >= 1; m >>= 1) { mx = max(mx, (unsigned)__shfl_xor((int)mx, m)); mw = max(mw, (unsigned)__shfl_xor((int)mw, m)); }
    if ((threadIdx.x & 63) == 0) { sx[threadIdx.x >> 6] = mx; sw[threadIdx.x >> 6] = mw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = max(max(sx[0], sx[1]), max(sx[2], sx[3]));
        mw = max(max(sw[0], sw[1]), max(sw[2], sw[3]));
        const float ax = __uint_as_float(mx), aw = __uint_as_float(mw);
        const float s_x = limb_scale(ax), s_w = limb_scale(aw * (float)G);    // a dense tap sums at most G units
        out->max_x_bits = mx; out->max_w_bits = mw;
        out->sx = s_x; out->sw = s_w; out->inv = (1.0f / s_x) * (1.0f / s_w);
    }
}

// ------------------------------------------------------------------------------------------------
// dense kernel synthesis: WS[chunk][tap][limb][f][sl] = limbs of sw * (sum over the units g of (c = 16*chunk + sl, f) of the
// bilinear weights that land on the tap).  One thread per (input channel, output channel): its K x K kernel is summed in LDS
// ([tap][thread] floats; the four taps of each of its G units added in unit order) and written out tap by tap.
// ------------------------------------------------------------------------------------------------
constexpr int kScT = 64;
__global__ void __launch_bounds__(kScT) split_densify_kernel(const UnitRef* __restrict__ table, int Cin, int G, int Cout, int CoutP, int nchunk,
                                                             const SplitScales* __restrict__ sc, _Float16* __restrict__ wsd, const Guard guard) {
    __shared__ float acc[kDTaps * kScT];
    if (!guard_pass(guard)) return;
    constexpr int FPB = kScT / 16;                         // output channels per workgroup
    const int tid = threadIdx.x, sl = tid & 15;
    const int fq = blockIdx.x % (CoutP / FPB), chunk = blockIdx.x / (CoutP / FPB);
    const int f = fq * FPB + (tid >> 4), c = chunk * 16 + sl;
#pragma unroll 7
    for (int tap = 0; tap < kDTaps; ++tap) acc[tap * kScT + tid] = 0.0f;
    if (c < Cin && f < Cout) {
        for (int g = 0; g < G; ++g) {
            const UnitRef u = table[((long)c * G + g) * Cout + f];
            const int ty = u.oy + kDR, tx = u.ox + kDR;
            // (a tap outside the kernel belongs to an offset of exactly +R, weight 0, or to a call whose guard does not pass)
            const bool y0 = ty >= 0 && ty < kDK, y1 = ty + 1 >= 0 && ty + 1 < kDK, x0 = tx >= 0 && tx < kDK, x1 = tx + 1 >= 0 && tx + 1 < kDK;
            if (y0 && x0) acc[(ty * kDK + tx) * kScT + tid] += u.w00;
            if (y0 && x1) acc[(ty * kDK + tx + 1) * kScT + tid] += u.w01;
            if (y1 && x0) acc[((ty + 1) * kDK + tx) * kScT + tid] += u.w10;
            if (y1 && x1) acc[((ty + 1) * kDK + tx + 1) * kScT + tid] += u.w11;
        }
    }
    const float sw = sc->sw;
    _Float16* dst = wsd + ((long)chunk * kDTaps * 2 * CoutP + f) * 16 + sl;
    const long limb = (long)CoutP * 16, tapstride = 2 * limb;
#pragma unroll 7
    for (int tap = 0; tap < kDTaps; ++tap) {
        const float v = acc[tap * kScT + tid] * sw;
        const _Float16 hi = (_Float16)v;
        dst[tap * tapstride] = hi;
        dst[tap * tapstride + limb] = (_Float16)(v - (float)hi);
    }
}

// ------------------------------------------------------------------------------------------------
// staging: in[N,C,H,W] (f32 or bf16) -> XS (separable Gaussian, scaled, two f16 limbs, chunked, zero border).
// Workgroup = (image, group of 8 channels, band of rows, segment of <= 64 columns).  The raw window of the eight channels goes
// to LDS as fp32; then every wave walks its share of the band's rows with lane = column: horizontal pass from LDS, vertical
// pass over a register ring of K rows, two 16-byte units (hi, lo) = 8 channels per position.
// ------------------------------------------------------------------------------------------------
constexpr int kSP = 80;                   // LDS row of one channel: [8 spare | 64 columns | 8 spare] floats
struct SplitStageArgs {
    const float* in;
    const float* taps;
    const SplitScales* sc;
    _Float16* xs;
    int N, C, H, W, mirrored;
    int Hs, Ws, nchunk;
    int RB, nbands, nsegs;   // rows per band, bands and 64-column segments per plane
    int vec;                 // rows are whole 4-element pieces (W % 4 == 0, base aligned)
    Guard guard;
};

// i / d for 0 <= i < 2^22 and a runtime d
__device__ __forceinline__ int fast_div(int i, int d, float inv) {
    int q = (int)(((float)i + 0.5f) * inv);
    const int r = i - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

// Horizontal pass, IN PLACE: the waves share the rows of the band's window (row r = wave, wave + nw, ...); a lane filters its column
// of NCH channels and writes the results over the raw values.  In place is safe because a row segment belongs to ONE wave: every read
// of a channel's row precedes the write of that channel in the wave's instruction stream (the written value depends on all of them)
// and a wave's LDS operations execute in order.  (Before: every wave filtered the K - 1 halo rows of its own rows again -- ten rows
// for four at the north-star shape.)
template <int K, int NCH>
__device__ __forceinline__ void split_stage_rows(float* rawl, const float* px, int col, int ch0, int wave, int nw, int lh, int ncols) {
    constexpr int kr = (K - 1) / 2;
    if (col >= ncols) return;
    float gx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) gx[j] = px[j];
    for (int r = wave; r < lh; r += nw) {
        float* row = rawl + (r * 8 + ch0) * kSP + 8 + col;
        float acc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            acc[ch] = 0.0f;
#pragma unroll
            for (int i = 0; i < K; ++i) acc[ch] = fmaf(row[ch * kSP + i - kr], gx[i], acc[ch]);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) row[ch * kSP] = acc[ch];
    }
}

// the walk of one wave over its rows ya .. yb: lane column `col`, channels ch0 .. ch0 + NCH - 1 of the group's eight; vertical pass over
// a register ring of K horizontally filtered rows (rawl after split_stage_rows), scale, split, store
template <int K, int NCH>
__device__ __forceinline__ void split_stage_walk(const float* rawl, const float* py, float sx, int col, int ch0, int ya, int yb,
                                                 int y0, int x0, int x1, int Ws, u32x4* xhi, u32x4* xlo) {
    constexpr int kr = (K - 1) / 2;
    const int x = x0 + col;
    if (x >= x1) return;
    float gy[K];
#pragma unroll
    for (int j = 0; j < K; ++j) gy[j] = py[j];
    float ring[K][NCH];
    const int nin = yb - ya + 2 * kr;                        // input rows ya - kr .. yb + kr - 1
    char* dhi = reinterpret_cast<char*>(xhi + (long)kDR * Ws + kDR + x) + ch0 * 2;
    char* dlo = reinterpret_cast<char*>(xlo + (long)kDR * Ws + kDR + x) + ch0 * 2;
    for (int s0 = 0; s0 < nin; s0 += K) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int s = s0 + j;
            if (s < nin) {
                const float* row = rawl + ((ya - y0 + s) * 8 + ch0) * kSP + 8 + col;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) ring[j][ch] = row[ch * kSP];
                if (s >= K - 1) {
                    typedef _Float16 f16xn __attribute__((ext_vector_type(NCH)));
                    f16xn oh, ol;
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        float acc = 0.0f;
#pragma unroll
                        for (int i = 0; i < K; ++i) acc = fmaf(ring[(j + 1 + i) % K][ch], gy[i], acc);
                        acc *= sx;
                        const _Float16 hi = (_Float16)acc;
                        oh[ch] = hi;
                        ol[ch] = (_Float16)(acc - (float)hi);
                    }
                    const long o = (long)(ya + s - (K - 1)) * Ws * 16;
                    *reinterpret_cast<f16xn*>(dhi + o) = oh;
                    *reinterpret_cast<f16xn*>(dlo + o) = ol;
                }
            }
        }
    }
}

#ifndef DAU_SPLIT_STAGE_THREADS
#define DAU_SPLIT_STAGE_THREADS 512        // eight waves share a band (same box: 375 us per pass at the north-star shape with 256 threads, 362 with 512)
#endif
constexpr int kStageThreads = DAU_SPLIT_STAGE_THREADS;
template <int K, bool BF>
__global__ void __launch_bounds__(kStageThreads) split_stage_kernel(const SplitStageArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rawl[];   // [row][8 channels][kSP]
    if (!guard_pass(a.guard)) return;
    constexpr int kr = (K - 1) / 2;
    constexpr int PPR = kSP / 4;                                   // 4-element pieces per LDS row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int t = blockIdx.x;
    const int seg = t % a.nsegs; t /= a.nsegs;
    const int band = t % a.nbands; t /= a.nbands;
    const int grp = t % (2 * a.nchunk);                             // chunk = grp / 2, half = grp & 1
    const int n = t / (2 * a.nchunk);
    const int y0 = band * a.RB, y1 = y0 + a.RB < a.H ? y0 + a.RB : a.H;
    const int x0 = seg * 64, x1 = x0 + 64 < a.W ? x0 + 64 : a.W;
    const int lh = y1 - y0 + 2 * kr;
    const long plane = (long)a.H * a.W;
    // ---- raw window -> LDS: piece (r, ch, q) covers image row y0 - kr + r, columns x0 - 8 + 4q .. + 3 of channel grp*8 + ch
    {
        const int pieces = lh * 8 * PPR;
        constexpr int UB = (13 * 256 + kStageThreads - 1) / kStageThreads;   // loads in flight per thread: the whole window of a 14-row band (12.5 pieces per thread of 256) in one batch
        for (int i0 = threadIdx.x; i0 < pieces; i0 += kStageThreads * UB) {
            float4 v[UB];
            // branch-free loads (clamped address, masked value): a branch around a load makes hipcc wait for it at the join
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * kStageThreads, rc = i / PPR, q = i - rc * PPR, r = rc >> 3, ch = rc & 7;
                const int c = grp * 8 + ch, y = y0 - kr + r, xs = x0 - 8 + 4 * q;
                const bool row_in = c < a.C && y >= 0 && y < a.H && i < pieces;
                const long base = ((long)n * a.C + (c < a.C ? c : 0)) * plane;
                if (a.vec) {
                    const bool ok = row_in && xs >= 0 && xs + 4 <= a.W;
                    const long idx = base + (ok ? (long)y * a.W + xs : 0);
                    if constexpr (BF) {
                        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.in) + idx);
                        const unsigned m = ok ? 0xffffffffu : 0u;
                        v[u] = make_float4(__uint_as_float((w.x << 16) & m), __uint_as_float(w.x & 0xffff0000u & m),
                                           __uint_as_float((w.y << 16) & m), __uint_as_float(w.y & 0xffff0000u & m));
                    } else {
                        const float4 w = *reinterpret_cast<const float4*>(a.in + idx);
                        v[u] = make_float4(mask_act(w.x, ok), mask_act(w.y, ok), mask_act(w.z, ok), mask_act(w.w, ok));
                    }
                } else {
                    float e[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool ok = row_in && xs + k >= 0 && xs + k < a.W;
                        e[k] = mask_act(load_act_t<BF>(a.in, base + (ok ? (long)y * a.W + xs + k : 0)), ok);
                    }
                    v[u] = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * kStageThreads;
                if (i < pieces) *reinterpret_cast<float4*>(rawl + (long)i * 4) = v[u];      // piece i sits at [r][ch][4q]: i * 4 floats
            }
        }
    }
    // ---- zero border of the staged planes (both limbs): what this workgroup's rows / columns touch outside the image
    const int chunk = grp >> 1, half = grp & 1;
    const long splane = (long)a.Hs * a.Ws;
    u32x4* xhi = reinterpret_cast<u32x4*>(a.xs) + ((((long)n * a.nchunk + chunk) * 2 + 0) * 2 + half) * splane;
    u32x4* xlo = xhi + 2 * splane;
    {
        const int rs0 = band == 0 ? 0 : y0 + kDR, rs1 = band == a.nbands - 1 ? a.Hs : y1 + kDR;
        const int cs0 = seg == 0 ? 0 : x0 + kDR, cs1 = seg == a.nsegs - 1 ? a.Ws : x1 + kDR;
        const int cw = cs1 - cs0, cnt = (rs1 - rs0) * cw;
        const float inv = 1.0f / (float)cw;
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int rr = fast_div(i, cw, inv), r = rs0 + rr, cc = cs0 + i - rr * cw;
            if (r >= y0 + kDR && r < y1 + kDR && cc >= x0 + kDR && cc < x1 + kDR) continue;
            xhi[(long)r * a.Ws + cc] = z;
            xlo[(long)r * a.Ws + cc] = z;
        }
    }
    __syncthreads();
    // lane = column; segments of at most 32 columns (maps up to 32 pixels wide) give the two half waves four channels each instead of
    // leaving half of the lanes idle
    const float* px = a.taps + (a.mirrored ? kTapGXR : kTapGX) * kTapPitch;
    const float* py = a.taps + (a.mirrored ? kTapGYR : kTapGY) * kTapPitch;
    // ---- horizontal pass over every row of the window, in place (the waves share the rows)
    if (x1 - x0 <= 32) split_stage_rows<K, 4>(rawl, px, lane & 31, (lane >> 5) * 4, wave, nw, lh, x1 - x0);
    else split_stage_rows<K, 8>(rawl, px, lane, 0, wave, nw, lh, x1 - x0);
    __syncthreads();
    // ---- vertical pass, scale, split, store: rows ya .. yb of this wave
    const int SR = (y1 - y0 + nw - 1) / nw;
    const int ya = y0 + wave * SR, yb = ya + SR < y1 ? ya + SR : y1;
    if (ya >= yb) return;
    const float sx = a.sc->sx;
    if (x1 - x0 <= 32) split_stage_walk<K, 4>(rawl, py, sx, lane & 31, (lane >> 5) * 4, ya, yb, y0, x0, x1, a.Ws, xhi, xlo);
    else split_stage_walk<K, 8>(rawl, py, sx, lane, 0, ya, yb, y0, x0, x1, a.Ws, xhi, xlo);
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct SplitArgs {
    const _Float16* xs;
    const _Float16* wsd;
    const SplitScales* sc;
    float* out;               // [N][Cout][H][W], f32 or bf16
    int N, Cout, CoutP, H, W, Hs, Ws, nchunk, ncb, nrb, out_bf16;
    int col0;                 // first column of this launch's blocks (a row is covered by blocks of NSUB and of NSUB - 1 tiles)
    int row0;                 // first row of this launch's row blocks (a map whose height leaves 1 .. 4 rows after its 8-row blocks
                              // ends with one block of FOUR rows: RG = 1)
    Guard guard;
};

// RG: row groups (of four rows) per workgroup.  2: the eight waves are 4 (32 channels) x 2 (row groups), a wave owns the NSUB tiles of its
// row group.  1: a block of four rows -- 4 (32 channels) x 2 (column halves), a wave owns (NSUB + 1) / 2 tiles; used for the last
// 1 .. 4 rows of a map (28- and 27-pixel maps: 24 + 4 rows instead of 32).
// TT ("tall tiles", RG = 2 only): a tile is 8 rows x 4 columns instead of 4 rows x 8 columns and NSUB counts those; the workgroup is
// 8 rows x NSUB*4 columns, its eight waves 4 (32 channels) x 2 (column halves of (NSUB + 1) / 2 and NSUB / 2 tiles -- the two waves of a
// SIMD, so the SIMD's work is NSUB tiles).  For maps whose width is 1 .. 4 columns more than a multiple of eight (28 = 7 x 4: no padded
// columns where 4 x 8 tiles pad to 32).
template <int NSUB, int RG = 2, bool TT = false>
__global__ void __launch_bounds__(512) split_gather_kernel(const SplitArgs a) {
    static_assert(!TT || RG == 2, "tall tiles: blocks of eight rows");
    constexpr int TW = TT ? 4 : 8;                           // columns of a tile
    constexpr int P = TT ? lds_pitch_narrow(NSUB) : lds_pitch(NSUB);   // LDS pitch (positions)
    constexpr int kRowsWG = 4 * RG;                          // output rows per workgroup
    constexpr int NT0 = (RG == 2 && !TT) ? NSUB : (NSUB + 1) / 2;      // tiles per wave
    constexpr int NT1 = TT ? NSUB / 2 : NT0;                 // ... of the second column half (tall tiles)
    constexpr int WR = kRowsWG + kDSpan;                     // window rows
    constexpr int HALF = WR * P * 16;                        // bytes of one (limb, half) plane window
    constexpr int BUFU = 4 * WR * P;                         // 16-byte units of a window: [limb][half][row][P]
    constexpr int NPIECE = (BUFU + 63) / 64;                 // 1 KiB pieces (one global_load_lds wave instruction each)
    constexpr int BUF = NPIECE * 1024;
    constexpr int PPW = (NPIECE + 7) / 8;                    // pieces per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!guard_pass(a.guard)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fw = wave & 3, pw = wave >> 2;                 // channel part / row group (RG = 2) or column half (RG = 1, TT) of the workgroup tile
    const int prow = (RG == 2 && !TT) ? 4 * pw : 0, ptile = (RG == 2 && !TT) ? 0 : pw * NT0;   // first row / first tile of this wave inside the workgroup tile
    // Workgroups are dealt to the eight XCDs round robin (block b runs on XCD b % 8) and every XCD has its own L2: consecutive
    // LOGICAL ids -- the channel blocks of one window, then the next column block, the next row block (both share halo with it),
    // the same image -- are mapped to one XCD, so that a window is fetched from HBM once, not once per channel block
    int t;
#ifdef DAU_SPLIT_NO_XCD_MAP                // (timing experiment: tools/build_variant.sh)
    t = blockIdx.x;
#else
    {
        const int nblk = gridDim.x, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        const int per = nblk / 8, rem = nblk % 8;
        t = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + idx;
    }
#endif
    const int fb = t % (a.CoutP / kDFB); t /= (a.CoutP / kDFB);       // channel blocks fastest: they share the window
    const int cb = t % a.ncb; t /= a.ncb;
    const int rb = t % a.nrb;
    const int n = t / a.nrb;
    const int h = lane >> 5, nn = lane & 31;

    // window copy: LDS unit L = ((limb*2 + half) * WR + r) * P + c  <-  plane (limb, half), staged position (rb*8 + r, cb*NSUB*8 + c).
    // Whole pitch rows are copied (c up to P - 1 reads past the window, into the row's tail or the next row: never used).
    const long xs_plane = (long)a.Hs * a.Ws;                 // 16-byte units per (n, chunk, limb, half)
    const int colb = a.col0 + cb * NSUB * TW;                // first column of this block
    const int rowb = a.row0 + rb * kRowsWG;                  // first row of this block
    const u32x4* xsrc = reinterpret_cast<const u32x4*>(a.xs) + ((long)n * a.nchunk * 4) * xs_plane + (long)rowb * a.Ws + colb;
    int goff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int piece = wave + 8 * i;
        piece = piece < NPIECE ? piece : NPIECE - 1;         // surplus pieces repeat the last one
        int L = piece * 64 + lane;
        L = L < BUFU ? L : BUFU - 1;
        const int ph = L / (WR * P), rem = L - ph * (WR * P), r = rem / P, c = rem - r * P;
        goff[i] = (int)(ph * xs_plane + (long)r * a.Ws + c);
    }
    auto issue = [&](int chunk, int buf) {
        const u32x4* src = xsrc + (long)chunk * 4 * xs_plane;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            int piece = wave + 8 * i;
            piece = piece < NPIECE ? piece : NPIECE - 1;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + goff[i]), (lds_ptr_t)(smem + buf * BUF + piece * 1024), 16, 0, 0);
        }
    };

    const int prr = TT ? nn >> 2 : nn >> 3, pcc = TT ? nn & 3 : nn & 7;   // this lane's position inside a tile
    // (kernel arguments the epilogue uses, read once: inside the lambda hipcc reloads them from the argument segment at every store)
    float* const out_ptr = a.out;
    const bool out_bf16 = a.out_bf16 != 0;
    const int out_c = a.Cout, out_h = a.H, out_w = a.W;
    auto body = [&](auto ntc) __attribute__((always_inline)) {
    constexpr int NT = decltype(ntc)::value;                 // tiles of this wave
    f32x16 sum[NT], acc[NT];                                 // running sums; the rows of taps being chained
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < NT; ++j) { sum[j] = zero; acc[j] = zero; }
    int rows_chained = 0;

    // A fragments: lane (nn, h) reads 16 bytes of channel fb*128 + fw*32 + nn; lo limb CoutP*2 units further
    const f16x8* wp = reinterpret_cast<const f16x8*>(a.wsd) + ((long)(fb * kDFB + fw * 32 + nn)) * 2 + h;
    const long wlo = (long)a.CoutP * 2, wtap = 2 * wlo;      // f16x8 units
    const unsigned lane_base = (unsigned)(h * HALF + ((prow + prr) * P + ptile * TW + pcc) * 16);

    // A wave whose four rows lie below the image (H = 28: the second half of the fourth row block) or whose 32 output channels lie
    // beyond Cout (96 channels padded to 128) has nothing to compute: it keeps copying its share of the windows and meeting the
    // barriers, and leaves the matrix pipe to its SIMD partner -- the kernel is bound by that pipe, so the workgroup finishes sooner.
#ifdef DAU_SPLIT_NO_IDLE_WAVES           // (timing experiment: tools/build_variant.sh)
    const bool live = true;
#else
    const bool live = rowb + prow < a.H && ptile < NSUB && colb + ptile * TW < a.W && fb * kDFB + fw * 32 < a.Cout;
#endif
    issue(0, 0);
    f16x8 ah[kDK], al[kDK], an[2], bn[2];                   // A fragments (hi, lo) of a row of taps; the next row's first two
    static_assert(kAhead == 2 && kDK >= 5 && kDK <= 9, "the A ring below is written for two taps ahead");
#pragma unroll
    for (int i = 0; i < kAhead; ++i) { ah[i] = wp[i * wtap]; al[i] = wp[i * wtap + wlo]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < a.nchunk) issue(chunk + 1, buf ^ 1);   // lands under this chunk's taps
        const unsigned bbase = lane_base + buf * BUF;
        // One row of taps per iteration, and NOTHING in flight across the back-edge: hipcc merges the wait-counter states at a loop
        // header conservatively (an LDS read or an A fragment carried over costs a wait for everything at its first use), so a row
        // starts with its own first hi read, and the next row's first two A fragments -- requested at taps 2 and 3, long landed --
        // are copied into place at the row's end.  Within a row: tap tx requests the A fragments of tap tx + 2 after its first
        // MFMA group, reads its lo fragments under the hi MFMAs and the next tap's hi fragments under the lo MFMAs.
#pragma unroll 1
        for (int ty = 0; ty < (live ? kDK : 0); ++ty) {
            const unsigned brow = bbase + ty * P * 16;
            f16x8 xh[NT], xl[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) xh[j] = *reinterpret_cast<const f16x8*>(smem + brow + (TW * j) * 16);
#pragma unroll
            for (int tx = 0; tx < kDK; ++tx) {
#pragma unroll
                for (int j = 0; j < NT; ++j) xl[j] = *reinterpret_cast<const f16x8*>(smem + brow + 2 * HALF + (tx + TW * j) * 16);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tx], xh[j], (kFlushRows == 1 && tx % kFlushTaps == 0) ? zero : acc[j], 0, 0, 0);
                if (tx + kAhead < kDK) { ah[tx + kAhead] = wp[(tx + kAhead) * wtap]; al[tx + kAhead] = wp[(tx + kAhead) * wtap + wlo]; }
                if (tx == 2 || tx == 3) { an[tx - 2] = wp[(kDK + tx - 2) * wtap]; bn[tx - 2] = wp[(kDK + tx - 2) * wtap + wlo]; }
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tx], xh[j], acc[j], 0, 0, 0);
                if (tx + 1 < kDK) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) xh[j] = *reinterpret_cast<const f16x8*>(smem + brow + (tx + 1 + TW * j) * 16);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tx], xl[j], acc[j], 0, 0, 0);
                // the order hipcc must keep (left alone it sinks every LDS read to just before its MFMA and waits for it there)
                if (tx == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                if (tx == 2 || tx == 3) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
                else if (tx + kAhead < kDK) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                if (tx + 1 < kDK) __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                if ((tx + 1) % kFlushTaps == 0 && tx + 1 < kDK) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) sum[j] += acc[j];
                }
            }
            wp += kDK * wtap;
            ah[0] = an[0]; al[0] = bn[0]; ah[1] = an[1]; al[1] = bn[1];
            if constexpr (kFlushRows == 1) {
#pragma unroll
                for (int j = 0; j < NT; ++j) sum[j] += acc[j];    // round-to-nearest adds of the row's partial sums
            } else {
                if (++rows_chained == kFlushRows) {
                    rows_chained = 0;
#pragma unroll
                    for (int j = 0; j < NT; ++j) { sum[j] += acc[j]; acc[j] = zero; }
                }
            }
        }
        if (chunk + 1 < a.nchunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next window has landed (this wave's pieces; the barrier joins the rest)
            __syncthreads();
        }
    }

    if constexpr (kFlushRows > 1) {
#pragma unroll
        for (int j = 0; j < NT; ++j) sum[j] += acc[j];     // the chain in progress
    }
    // epilogue: C/D layout of the 32x32 tile: column (pixel) = lane & 31, row (channel) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const float inv = a.sc->inv;
    const int y = rowb + prow + prr;
    const long plane = (long)out_h * out_w;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int x = colb + TW * (ptile + j) + pcc;
        if (y < out_h && x < out_w && ptile + j < NSUB) {      // (RG = 1, odd NSUB: the second column half's last tile does not exist)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int f = fb * kDFB + fw * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (f < out_c) store_act(out_ptr, ((long)n * out_c + f) * plane + (long)y * out_w + x, sum[j][i] * inv, out_bf16, false);
            }
        }
    }
    };
    // (both branches meet the same barriers: one per chunk)
    if constexpr (TT) {
        if (pw == 0) body(std::integral_constant<int, NT0>{});
        else body(std::integral_constant<int, NT1>{});
    } else {
        body(std::integral_constant<int, NT0>{});
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

template <int NSUB, int RG, bool TT>
constexpr size_t split_lds_bytes() { return 2 * (size_t)((4 * (4 * RG + kDSpan) * (TT ? lds_pitch_narrow(NSUB) : lds_pitch(NSUB)) + 63) / 64) * 1024; }

template <int NSUB, int RG, bool TT = false>
void launch_split(hipStream_t st, const SplitArgs* a, int grid) {
    auto kern = split_gather_kernel<NSUB, RG, TT>;
    if (!a) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return; }
    constexpr size_t lds = split_lds_bytes<NSUB, RG, TT>();      // (a comma inside the launch macro's arguments would split them)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, *a);
}

void dispatch_split(int nsub, int rg, hipStream_t st, const SplitArgs* a, int grid, bool tall = false) {
    if (tall) {                                              // nsub = tiles of four columns: 5 or 7 (split_geometry)
        if (nsub == 5) launch_split<5, 2, true>(st, a, grid);
        else launch_split<7, 2, true>(st, a, grid);
        return;
    }
    if (rg == 1) {
        switch (nsub) {
            case 1: launch_split<1, 1>(st, a, grid); break;
            case 2: launch_split<2, 1>(st, a, grid); break;
            case 3: launch_split<3, 1>(st, a, grid); break;
            default: launch_split<4, 1>(st, a, grid); break;
        }
        return;
    }
    switch (nsub) {
        case 1: launch_split<1, 2>(st, a, grid); break;
        case 2: launch_split<2, 2>(st, a, grid); break;
        case 3: launch_split<3, 2>(st, a, grid); break;
        default: launch_split<4, 2>(st, a, grid); break;
    }
}

const void* stage_for(int blur_k, bool bf16) {
#define DAU_SPLIT_STAGE(K) case K: return bf16 ? reinterpret_cast<const void*>(split_stage_kernel<K, true>) : reinterpret_cast<const void*>(split_stage_kernel<K, false>)
    switch (blur_k) {
        DAU_SPLIT_STAGE(3);
        DAU_SPLIT_STAGE(5);
        DAU_SPLIT_STAGE(7);
        DAU_SPLIT_STAGE(9);
        DAU_SPLIT_STAGE(11);
        default: return nullptr;                 // wider prefilters: no split form (the exact gather runs)
    }
#undef DAU_SPLIT_STAGE
}
// bands of rows whose raw window (eight channels, rows of kSP floats) stays below ~52 KiB of LDS: three workgroups per CU
void stage_plan(const DenseConfig& c, int* RB, int* nbands, size_t* lds) {
    const int kr = (c.blur_k - 1) / 2;
    int rows = 52 * 1024 / (8 * kSP * 4) - 2 * kr;
    const int rb = DAU_TUNE_INT("DAU_SPLIT_STAGE_RB", 0);
    if (rb > 0) rows = rb;
    rows = rows < 4 ? 4 : rows;
    const int nb = (c.H + rows - 1) / rows;
    *nbands = nb; *RB = (c.H + nb - 1) / nb;
    *lds = (size_t)(*RB + 2 * kr) * 8 * kSP * 4;
}

}  // namespace

bool split_gather_configure(int N, int Cin, int Cout, int G, int H, int W, int R, int blur_k, bool bf16, DenseConfig* cfg) {
    if (R != kDR || !stage_for(blur_k, bf16)) return false;
    DenseConfig c{};
    c.N = N; c.Cin = Cin; c.Cout = Cout; c.G = G; c.H = H; c.W = W; c.R = R; c.blur_k = blur_k; c.bf16 = bf16 ? 1 : 0;
    const SplitGeom g = split_geometry(c);
    c.nsub = g.nsub_a;
    c.ftiles = 1;
    // 32-bit unit offsets inside one image's staged planes
    if ((size_t)g.nchunk * 4 * g.Hs * g.Ws > (size_t)1 << 30) return false;
    *cfg = c;
    return true;
}

size_t split_gather_workspace_bytes(const DenseConfig& c) {
    const SplitGeom g = split_geometry(c);
    return g.hdr_bytes + g.xs_bytes + g.ws_bytes;
}

void split_gather_init(const DenseConfig& c) {
    const SplitGeom g = split_geometry(c);
    for (int rg = 1; rg <= 2; ++rg) {
        if (!(rg == 2 ? g.nrb8 : g.nrb4)) continue;
        if (rg == 2 && g.tall) { dispatch_split(g.tall, 2, nullptr, nullptr, 0, true); continue; }
        dispatch_split(g.nsub_a, rg, nullptr, nullptr, 0);
        if (g.nb_b) dispatch_split(g.nsub_b, rg, nullptr, nullptr, 0);
    }
    (void)hipFuncSetAttribute(stage_for(c.blur_k, c.bf16 != 0), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void split_gather_prepare(hipStream_t st, const DenseConfig& c, const float* in, const float* filters, bool mirrored,
                          const UnitRef* table, void* workspace, const Guard& guard) {
    const SplitGeom g = split_geometry(c);
    char* ws = static_cast<char*>(workspace);
    SplitScales* sc = reinterpret_cast<SplitScales*>(ws);
    unsigned* partial = reinterpret_cast<unsigned*>(ws + sizeof(SplitScales));
    _Float16* xs = reinterpret_cast<_Float16*>(ws + g.hdr_bytes);
    _Float16* wsd = reinterpret_cast<_Float16*>(ws + g.hdr_bytes + g.xs_bytes);
    const long count = (long)c.N * c.Cin * c.H * c.W, units = (long)c.Cin * c.G * c.Cout;
    const int nparts = (int)std::min<long>(kPartials, (count / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(split_absmax_kernel, dim3(nparts), dim3(256), 0, st, in, count, c.bf16, table, units, partial, guard);
    hipLaunchKernelGGL(split_scales_kernel, dim3(1), dim3(256), 0, st, partial, nparts, c.G, sc, guard);
    hipLaunchKernelGGL(split_densify_kernel, dim3(g.nchunk * (g.CoutP / (kScT / 16))), dim3(kScT), 0, st, table, c.Cin, c.G, c.Cout, g.CoutP,
                       g.nchunk, sc, wsd, guard);
    SplitStageArgs s{};
    s.in = in; s.taps = filters + kTaps1dOffset; s.sc = sc; s.xs = xs;
    s.N = c.N; s.C = c.Cin; s.H = c.H; s.W = c.W; s.mirrored = mirrored ? 1 : 0;
    s.Hs = g.Hs; s.Ws = g.Ws; s.nchunk = g.nchunk; s.guard = guard;
    size_t lds;
    stage_plan(c, &s.RB, &s.nbands, &lds);
    s.nsegs = (c.W + 63) / 64;
    s.vec = c.W % 4 == 0 && reinterpret_cast<uintptr_t>(in) % 16 == 0;
    void* args[] = {&s};
    (void)hipLaunchKernel(stage_for(c.blur_k, c.bf16 != 0), dim3(c.N * 2 * g.nchunk * s.nbands * s.nsegs), dim3(kStageThreads), args, lds, st);
}

void split_gather_run(hipStream_t st, const DenseConfig& c, float* out, void* workspace, const Guard& guard) {
    const SplitGeom g = split_geometry(c);
    char* ws = static_cast<char*>(workspace);
    SplitArgs a{};
    a.sc = reinterpret_cast<const SplitScales*>(ws);
    a.xs = reinterpret_cast<const _Float16*>(ws + g.hdr_bytes);
    a.wsd = reinterpret_cast<const _Float16*>(ws + g.hdr_bytes + g.xs_bytes);
    a.out = out;
    a.N = c.N; a.Cout = c.Cout; a.CoutP = g.CoutP; a.H = c.H; a.W = c.W; a.Hs = g.Hs; a.Ws = g.Ws; a.nchunk = g.nchunk;
    a.out_bf16 = c.bf16; a.guard = guard;
    for (int rg = 2; rg >= 1; --rg) {                        // the eight-row blocks, then the block of four rows where there is one
        a.nrb = rg == 2 ? g.nrb8 : g.nrb4;
        if (!a.nrb) continue;
        a.row0 = rg == 2 ? 0 : g.nrb8 * kDRows;
        if (rg == 2 && g.tall) {                             // one column block of tall tiles
            a.ncb = 1; a.col0 = 0;
            dispatch_split(g.tall, 2, st, &a, c.N * a.nrb * (g.CoutP / kDFB), true);
            continue;
        }
        a.ncb = g.nb_a; a.col0 = 0;
        dispatch_split(g.nsub_a, rg, st, &a, c.N * a.nrb * g.nb_a * (g.CoutP / kDFB));
        if (g.nb_b) {
            a.ncb = g.nb_b; a.col0 = g.nb_a * g.nsub_a * 8;
            dispatch_split(g.nsub_b, rg, st, &a, c.N * a.nrb * g.nb_b * (g.CoutP / kDFB));
        }
    }
}

}  // namespace DAU_SPLIT_NS
}  // namespace dau
