// Densified gather-sum on the bf16 matrix cores (DAU_FLAG_DENSE_BF16; offsets within +-4 only).
//
//   out[n,f,y,x] = sum_{c} sum_{ty,tx < 9} Wd[f][c][ty][tx] * Xb[n,c, y+ty-4, x+tx-4]
//
// The G units of every (input channel c, output channel f) pair are scattered into a dense 9 x 9 kernel
// (integer offsets -4..4 plus the second bilinear tap: Wd[f][c][oy+dy+4][ox+dx+4] += w * b_dydx) and the pass becomes an
// implicit GEMM  M = output channels, N = pixels, K = input channels x 81 taps  on v_mfma_f32_32x32x16_bf16 (fp32
// accumulation).  That is 81 / (4 G) times the FLOPs of the exact gather (k_gather_mfma.hip) at 16 x its matrix rate:
// SURVEY.md section 7 hard part (A), measured with the library convolution in tools/probe_densified_bf16.py (1.9 - 2.7 x
// faster than the gather for the forward pass of BASELINE config 2).  It replaces the same reference code as the gather:
// DAUConv_forward_pipeline_kernel + interleave_input_data_kernel + perpare_weights_and_offsets
// (include/dau_conv/dau_conv_impl/dau_conv_forward_core.hpp:804-1605, 1607-1732, 1858-2215) and caffe_gpu_convolve2
// (src/dau_conv/util/convolve.cu:48-131).  Numerics: taps and blurred activations are rounded to bfloat16, products are
// exact, sums are fp32 -- inside the 2e-2 bar of the bf16 configuration, not the 1e-4 bar of the fp32 one, hence opt-in
// and only together with DAU_FLAG_IO_BF16.
//
// Layouts (HBM):
//   XD[n][chunk][half][Hs][Ws][8]  bf16: Gaussian-blurred input, 16 input channels per chunk as two halves of 8 (one
//       16-byte unit per position and half = the B fragment of one lane), staged position (r, c) = image (r-R, c-R),
//       zero outside the image; Hs, Ws cover whole row / column blocks plus the tap border (2 R positions).
//   WD[chunk][tap][CoutP][16]      bf16: the dense kernel, 32 bytes per output channel (A fragments of the two lane halves).
// Workgroup = 128 output channels x 8 rows x NSUB*8 columns, as 8 waves = 4 (32 channels each) x 2 (4 rows each), two per
// SIMD (or 4 waves with two channel tiles each, FT = 2); a wave owns FT x NSUB accumulator tiles of 32 channels x (4 rows x
// 8 columns).  Per chunk the 16 x (NSUB*8+8) window of
// both halves sits in LDS (pitch = 8 mod 16 positions: the four rows of a B fragment fall on different banks); per tap a
// wave loads two A fragments from global memory (L1/L2 resident: 8 KB per tap and chunk for 256 channels) and NSUB B
// fragments with ds_read_b128 at the tap's displacement, and issues 2*NSUB MFMAs.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "dau_tiled.hpp"

// The file is compiled once per offset radius of the dense form (Makefile): R = 4 (namespace r4, the bucket-4 member: |mu| <= 4)
// and R = 3 (namespace r3: calls whose |mu| <= 3 -- 49 taps instead of 81; the device guard of the call picks one of the two).
#ifndef DAU_DENSE_R
#define DAU_DENSE_R 4                   // (tools/build_variant.sh ... -DDAU_DENSE_R=8: the 18 x 18 form of bucket 8, timing experiment)
#endif
#ifndef DAU_DENSE_NS
#define DAU_DENSE_NS r4
#endif

namespace dau {
namespace DAU_DENSE_NS {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

constexpr int kDR = DAU_DENSE_R;        // offset radius of the dense form
// Taps per axis: integer offsets -R .. R plus the second bilinear tap.  An offset of exactly +R has fraction 0, so the taps at
// R + 1 carry the weight 0 for every unit a guarded call can see (|mu| <= R): the production form (R = 4) leaves that row and
// column out -- 81 taps instead of 100.  (The R = 8 timing variant keeps the 18 x 18 form it was measured with.)
constexpr int kDK = kDR <= 4 ? 2 * kDR + 1 : 2 * kDR + 2;
constexpr int kDSpan = kDK - 1;         // border of the staged plane and of the LDS window (positions)
constexpr int kDTaps = kDK * kDK;
constexpr int kDRows = 8;               // output rows per workgroup
constexpr int kDFB = 128;               // output channels per workgroup

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DenseGeom {
    int nsub;                // 8-pixel subtiles per column block (kernel instantiation): 2, 4, 7 or 8
    int ncb, nrb;            // column / row blocks per image
    int Hs, Ws;              // staged plane (positions)
    int nchunk, CoutP;
    size_t xd_bytes, wd_bytes;
};

DenseGeom dense_geometry(const DenseConfig& c) {
    DenseGeom g{};
    const int sub = (c.W + 7) / 8;
    // fewest wasted columns first, then fewest blocks
    int best = -1; long best_cost = 0;
    for (int ns : {2, 4, 7, 8}) {
        const int ncb = (sub + ns - 1) / ns;
        const long cost = (long)ncb * ns * 1000 + ncb;
        if (best < 0 || cost < best_cost) { best = ns; best_cost = cost; }
    }
    g.nsub = best;
    g.ncb = (sub + g.nsub - 1) / g.nsub;
    g.nrb = (c.H + kDRows - 1) / kDRows;
    g.Hs = g.nrb * kDRows + kDSpan;
    g.Ws = g.ncb * g.nsub * 8 + kDSpan;
    g.nchunk = (c.Cin + 15) / 16;
    g.CoutP = (int)round_up(c.Cout, kDFB);
    g.xd_bytes = round_up((size_t)c.N * g.nchunk * 2 * g.Hs * g.Ws * 16, 256);
    g.wd_bytes = round_up(((size_t)g.nchunk * kDTaps + 8) * g.CoutP * 32, 256);   // + look-ahead taps of the A stream
    return g;
}

constexpr int lds_pitch(int nsub) {       // positions; = 8 (mod 16) and >= nsub*8 + 8
    int p = nsub * 8 + kDSpan;
    while (p % 16 != 8) ++p;
    return p;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// dense kernel synthesis: WD[chunk][tap][f][sl] = sum over the units g of (c = 16*chunk + sl, f) of the bilinear weights
// that land on tap (ty, tx).  table is indexed [Cin][G][Cout] like every gather-sum unit table (forward: w-scaled
// factors; dx pass: the transposed table with negated offsets).  One thread per output element, coalesced writes.
// ------------------------------------------------------------------------------------------------
__global__ void densify_units_kernel(const UnitRef* __restrict__ table, int Cin, int G, int Cout, int CoutP, int nchunk,
                                     __bf16* __restrict__ wd, const Guard guard) {
    if (!guard_pass(guard)) return;
    const long total = (long)nchunk * kDTaps * CoutP * 16;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int sl = (int)(idx & 15);
        const int f = (int)((idx >> 4) % CoutP);
        const int tap = (int)((idx / ((long)CoutP * 16)) % kDTaps);
        const int chunk = (int)(idx / ((long)CoutP * 16 * kDTaps));
        const int c = chunk * 16 + sl, ty = tap / kDK, tx = tap % kDK;
        float v = 0.0f;
        if (c < Cin && f < Cout) {
            for (int g = 0; g < G; ++g) {
                const UnitRef u = table[((long)c * G + g) * Cout + f];
                const int ry = ty - (u.oy + kDR), rx = tx - (u.ox + kDR);     // which of the unit's 2 x 2 taps this is
                if (ry == 0 && rx == 0) v += u.w00;
                if (ry == 0 && rx == 1) v += u.w01;
                if (ry == 1 && rx == 0) v += u.w10;
                if (ry == 1 && rx == 1) v += u.w11;
            }
        }
        wd[idx] = (__bf16)v;
    }
}

// The same table, one thread per (input channel, output channel): its dense kernel is summed in LDS ([tap][thread] floats, zeroed,
// then the four taps of each of its G units added) and written out tap by tap -- a wave = 4 output channels x the 16 input channels
// of a chunk writes 128 contiguous bytes per tap.  G table reads per thread instead of G per OUTPUT ELEMENT (81 x fewer).
constexpr int kScT = kDTaps <= 128 ? 64 : 32;             // threads per workgroup (the accumulators fill the static LDS limit otherwise)
__global__ void __launch_bounds__(kScT) densify_units_scatter_kernel(const UnitRef* __restrict__ table, int Cin, int G, int Cout, int CoutP,
                                                                     int nchunk, __bf16* __restrict__ wd, const Guard guard) {
    __shared__ float acc[kDTaps * kScT];
    if (!guard_pass(guard)) return;
    constexpr int FPB = kScT / 16;                         // output channels per workgroup
    const int tid = threadIdx.x, sl = tid & 15;
    const int fq = blockIdx.x % (CoutP / FPB), chunk = blockIdx.x / (CoutP / FPB);
    const int f = fq * FPB + (tid >> 4), c = chunk * 16 + sl;
#pragma unroll 4
    for (int tap = 0; tap < kDTaps; ++tap) acc[tap * kScT + tid] = 0.0f;
    if (c < Cin && f < Cout) {
        for (int g = 0; g < G; ++g) {                       // in the order of densify_units_kernel: the same fp32 sums
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
    __bf16* dst = wd + ((long)chunk * kDTaps * CoutP + f) * 16 + sl;
#pragma unroll 4
    for (int tap = 0; tap < kDTaps; ++tap) dst[(long)tap * CoutP * 16] = (__bf16)acc[tap * kScT + tid];
}

// ------------------------------------------------------------------------------------------------
// staging: in[N,C,H,W] (bf16 or f32) -> XD (blurred with the separable Gaussian, bf16, chunked, zero border).
// One workgroup per (image, group of 8 channels = one half of a chunk, tile of TR x TC staged positions): raw window ->
// LDS, horizontal pass -> LDS, vertical pass -> one 16-byte unit per position.  HBM bound.
// ------------------------------------------------------------------------------------------------
struct DenseStageArgs {
    const float* in;
    const float* taps;       // 1-D factor arrays (dau_common.hpp)
    __bf16* xd;
    int N, C, H, W, k, mirrored, bf16;
    int Hs, Ws, nchunk;
    int TR, TC, ntr, ntc;    // tile of staged positions and tiles per plane
    Guard guard;
};

// i / d for 0 <= i < 2^22 and a runtime d (the loops below index flat tiles; integer division proper costs ~40 instructions)
__device__ __forceinline__ int fast_div(int i, int d, float inv) {
    int q = (int)(((float)i + 0.5f) * inv);
    const int r = i - q * d;
    q += (r >= d) - (r < 0);
    return q;
}

__global__ void __launch_bounds__(256) dense_stage_kernel(const DenseStageArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!guard_pass(a.guard)) return;
    const int k = a.k, kr = (k - 1) / 2;
    int t = blockIdx.x;
    const int tc = t % a.ntc; t /= a.ntc;
    const int tr = t % a.ntr; t /= a.ntr;
    const int grp = t % (2 * a.nchunk);        // group of 8 channels: chunk = grp / 2, half = grp & 1
    const int n = t / (2 * a.nchunk);
    const int r0 = tr * a.TR, c0 = tc * a.TC;  // staged origin of the tile
    const int rows = r0 + a.TR < a.Hs ? a.TR : a.Hs - r0, cols = c0 + a.TC < a.Ws ? a.TC : a.Ws - c0;
    const int lh = rows + k - 1, lw = cols + k - 1;
    // four channels at a time (LDS: raw [4][lh][lw] + horizontally filtered [4][lh][cols]), two rounds per group
    float* raw = lds;
    float* hor = lds + 4 * lh * lw;
    const float* gx = a.taps + (a.mirrored ? kTapGXR : kTapGX) * kTapPitch;
    const float* gy = a.taps + (a.mirrored ? kTapGYR : kTapGY) * kTapPitch;
    const long plane = (long)a.H * a.W;
    const float inv_lw = 1.0f / (float)lw, inv_cols = 1.0f / (float)cols;
    const int npos = rows * cols;
    constexpr int kMaxPos = 5;                  // positions per thread: TR * TC <= 5 * 256
    float res[kMaxPos][8];
    const int plane_raw = lh * lw, plane_hor = lh * cols;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (round) __syncthreads();             // the vertical pass of the first round is done with `hor`
        for (int i = threadIdx.x; i < 4 * plane_raw; i += 256) {
            const int ch = i >= 2 * plane_raw ? (i >= 3 * plane_raw ? 3 : 2) : (i >= plane_raw ? 1 : 0);
            const int rem = i - ch * plane_raw, y = fast_div(rem, lw, inv_lw), x = rem - y * lw;
            const int c = grp * 8 + round * 4 + ch, iy = r0 - kDR - kr + y, ix = c0 - kDR - kr + x;
            float v = 0.0f;
            if (c < a.C && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = load_act(a.in, ((long)n * a.C + c) * plane + (long)iy * a.W + ix, a.bf16 != 0);
            raw[i] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * plane_hor; i += 256) {
            const int ch = i >= 2 * plane_hor ? (i >= 3 * plane_hor ? 3 : 2) : (i >= plane_hor ? 1 : 0);
            const int rem = i - ch * plane_hor, y = fast_div(rem, cols, inv_cols), x = rem - y * cols;
            const float* src = raw + ch * plane_raw + y * lw + x;
            float acc = 0.0f;
            for (int j = 0; j < k; ++j) acc = fmaf(src[j], gx[j], acc);
            hor[i] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kMaxPos; ++q) {
            const int i = threadIdx.x + q * 256;
            if (i < npos) {
                const int y = fast_div(i, cols, inv_cols), x = i - y * cols;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    float acc = 0.0f;
                    const float* src = hor + ch * plane_hor + y * cols + x;
                    for (int j = 0; j < k; ++j) acc = fmaf(src[j * cols], gy[j], acc);
                    res[q][round * 4 + ch] = acc;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kMaxPos; ++q) {
        const int i = threadIdx.x + q * 256;
        if (i < npos) {
            const int y = fast_div(i, cols, inv_cols), x = i - y * cols;
            const int iy = r0 + y - kDR, ix = c0 + x - kDR;
            const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;      // the blurred image is zero outside the image
            bf16x8 o;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) o[ch] = (__bf16)(inside ? res[q][ch] : 0.0f);
            bf16x8* dst = reinterpret_cast<bf16x8*>(a.xd) + (((long)n * 2 * a.nchunk + grp) * a.Hs + (r0 + y)) * a.Ws + (c0 + x);
            *dst = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// staging, fast form: bfloat16 input and an instantiated prefilter support K (taps in scalar registers, tap loops unrolled).
// Workgroup = (image, group of 8 channels, band of rows, segment of <= 64 columns).  The raw window of the eight channels goes to
// LDS as it is stored (bf16 bits, 16-byte pieces: a wave per channel, the pieces of consecutive rows are contiguous in HBM); then
// every wave walks its share of the band's rows with lane = column: horizontal pass from LDS (one ds_read_u16 per tap and channel,
// a wave reads 128 contiguous bytes), vertical pass over a register ring of K rows, one 16-byte unit = 8 channels per position.
// Loads and stores are 16 bytes per lane and contiguous; the sums run in the order of dense_stage_kernel (bit-identical output).
// ------------------------------------------------------------------------------------------------
constexpr int kSP = 80;                   // LDS row of one channel: [8 spare | 64 columns | 8 spare] bf16 = ten 16-byte pieces
struct DenseStageRowsArgs {
    const unsigned short* in;
    const float* taps;
    __bf16* xd;
    int N, C, H, W, mirrored;
    int Hs, Ws, nchunk;
    int RB, nbands, nsegs;   // rows per band, bands and 64-column segments per plane
    int vec;                 // rows are whole 16-byte pieces (W % 8 == 0, base aligned)
    Guard guard;
};

template <int K>
__global__ void __launch_bounds__(512) dense_stage_rows_kernel(const DenseStageRowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short rawl[];   // [row][8 channels][kSP]
    if (!guard_pass(a.guard)) return;
    constexpr int kr = (K - 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int t = blockIdx.x;
    const int seg = t % a.nsegs; t /= a.nsegs;
    const int band = t % a.nbands; t /= a.nbands;
    const int grp = t % (2 * a.nchunk);
    const int n = t / (2 * a.nchunk);
    const int y0 = band * a.RB, y1 = y0 + a.RB < a.H ? y0 + a.RB : a.H;
    const int x0 = seg * 64, x1 = x0 + 64 < a.W ? x0 + 64 : a.W;
    const int lh = y1 - y0 + 2 * kr;
    const long plane = (long)a.H * a.W;
    // ---- raw window -> LDS: piece (r, q) of channel ch covers image row y0 - kr + r, columns x0 - 8 + 8q .. + 7
    for (int ch = wave; ch < 8; ch += nw) {
        const int c = grp * 8 + ch;
        const unsigned short* src = a.in + ((long)n * a.C + (c < a.C ? c : 0)) * plane;
        const int pieces = lh * (kSP / 8);
        for (int i0 = lane; i0 < pieces; i0 += 64 * 4) {
            uint4 v[4];
            // branch-free loads (clamped address, masked value): a branch around a load makes hipcc wait for it at the join
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 64, r = i / (kSP / 8), q = i - r * (kSP / 8);
                const int y = y0 - kr + r, xs = x0 - 8 + 8 * q;
                const bool row_in = c < a.C && y >= 0 && y < a.H && i < pieces;
                if (a.vec) {
                    const bool ok = row_in && xs >= 0 && xs + 8 <= a.W;
                    const uint4 w = *reinterpret_cast<const uint4*>(src + (ok ? (long)y * a.W + xs : 0));
                    const unsigned m = ok ? 0xffffffffu : 0u;
                    v[u] = make_uint4(w.x & m, w.y & m, w.z & m, w.w & m);
                } else {
                    unsigned e[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool ok = row_in && xs + k >= 0 && xs + k < a.W;
                        e[k] = (unsigned)src[ok ? (long)y * a.W + xs + k : 0] & (ok ? 0xffffu : 0u);
                    }
                    v[u] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 64, r = i / (kSP / 8), q = i - r * (kSP / 8);
                if (i < pieces) *reinterpret_cast<uint4*>(rawl + ((r * 8 + ch) * kSP + 8 * q)) = v[u];
            }
        }
    }
    // ---- zero border of the staged plane: what this workgroup's rows / columns touch outside the image
    {
        const int rs0 = band == 0 ? 0 : y0 + kDR, rs1 = band == a.nbands - 1 ? a.Hs : y1 + kDR;
        const int cs0 = seg == 0 ? 0 : x0 + kDR, cs1 = seg == a.nsegs - 1 ? a.Ws : x1 + kDR;
        const int cw = cs1 - cs0, cnt = (rs1 - rs0) * cw;
        const float inv = 1.0f / (float)cw;
        bf16x8* dst = reinterpret_cast<bf16x8*>(a.xd) + ((long)n * 2 * a.nchunk + grp) * a.Hs * a.Ws;
        bf16x8 z;
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = (__bf16)0.0f;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int rr = fast_div(i, cw, inv), r = rs0 + rr, cc = cs0 + i - rr * cw;
            if (r >= y0 + kDR && r < y1 + kDR && cc >= x0 + kDR && cc < x1 + kDR) continue;
            dst[(long)r * a.Ws + cc] = z;
        }
    }
    __syncthreads();
    // ---- rows ya .. yb of this wave, lane = column
    const int SR = (y1 - y0 + nw - 1) / nw;
    const int ya = y0 + wave * SR, yb = ya + SR < y1 ? ya + SR : y1;
    const int x = x0 + lane;
    if (ya >= yb || x >= x1) return;
    float gx[K], gy[K];
    {
        const float* px = a.taps + (a.mirrored ? kTapGXR : kTapGX) * kTapPitch;
        const float* py = a.taps + (a.mirrored ? kTapGYR : kTapGY) * kTapPitch;
#pragma unroll
        for (int j = 0; j < K; ++j) { gx[j] = px[j]; gy[j] = py[j]; }
    }
    float ring[K][8];
    const int nin = yb - ya + 2 * kr;                        // input rows ya - kr .. yb + kr - 1
    bf16x8* dst = reinterpret_cast<bf16x8*>(a.xd) + (((long)n * 2 * a.nchunk + grp) * a.Hs + kDR) * a.Ws + kDR + x;
    for (int s0 = 0; s0 < nin; s0 += K) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int s = s0 + j;
            if (s < nin) {
                const unsigned short* row = rawl + ((ya - y0 + s) * 8) * kSP + 8 + lane - kr;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    float acc = 0.0f;
#pragma unroll
                    for (int i = 0; i < K; ++i) acc = fmaf(__uint_as_float((unsigned)row[ch * kSP + i] << 16), gx[i], acc);
                    ring[j][ch] = acc;
                }
                if (s >= K - 1) {
                    bf16x8 o;
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        float acc = 0.0f;
#pragma unroll
                        for (int i = 0; i < K; ++i) acc = fmaf(ring[(j + 1 + i) % K][ch], gy[i], acc);
                        o[ch] = (__bf16)acc;
                    }
                    dst[(long)(ya + s - (K - 1)) * a.Ws] = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct DenseArgs {
    const __bf16* xd;
    const __bf16* wd;
    float* out;               // [N][Cout][H][W], bf16 or f32
    int N, Cout, CoutP, H, W, Hs, Ws, nchunk, ncb, nrb, out_bf16;
    Guard guard;
};

// FT: 32-channel accumulator tiles per wave: 2 -> 4 waves per workgroup (one per SIMD), 1 -> 8 waves (two per SIMD, half
// the accumulators each, every B fragment read by twice as many waves)
template <int NSUB, int FT>
__global__ void __launch_bounds__(FT == 2 ? 256 : 512) dense_gather_kernel(const DenseArgs a) {
    constexpr int kThreads = FT == 2 ? 256 : 512;
    constexpr int P = lds_pitch(NSUB);                       // LDS pitch (positions)
    constexpr int WC = NSUB * 8 + kDSpan;                    // window columns
    constexpr int WR = kDRows + kDSpan;                      // window rows
    constexpr int HALF = WR * P * 16;                        // bytes of one half-plane window
    constexpr int BUF = 2 * HALF;
    constexpr int PIECES = 2 * WR * WC;                      // 16-byte pieces of a window
    constexpr int PER = (PIECES + kThreads - 1) / kThreads;
    constexpr int FW = 4 / FT;                               // waves along the 128 output channels
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!guard_pass(a.guard)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fw = wave % FW, pw = wave / FW;                // channel part / row half of the workgroup tile
    int t = blockIdx.x;
    const int fb = t % (a.CoutP / kDFB); t /= (a.CoutP / kDFB);       // channel blocks fastest: they share the window
    const int cb = t % a.ncb; t /= a.ncb;
    const int rb = t % a.nrb;
    const int n = t / a.nrb;
    const int h = lane >> 5, nn = lane & 31;

    // window pieces of this thread: piece p -> (half, row, col)
    const long xd_plane = (long)a.Hs * a.Ws;                 // 16-byte units per (n, chunk, half)
    const u32x4* xsrc = reinterpret_cast<const u32x4*>(a.xd) + ((long)n * a.nchunk * 2) * xd_plane + (long)(rb * kDRows) * a.Ws + cb * NSUB * 8;
    int goff[PER]; unsigned loff[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int p = threadIdx.x + i * kThreads;
        if (p >= PIECES) p = PIECES - 1;                      // surplus threads repeat the last piece
        const int ph = p / (WR * WC), rem = p - ph * (WR * WC), r = rem / WC, c = rem - r * WC;
        goff[i] = (int)(ph * xd_plane + (long)r * a.Ws + c);
        loff[i] = (unsigned)(ph * HALF + (r * P + c) * 16);
    }
    auto fetch = [&](int chunk, u32x4 (&regs)[PER]) {
        const u32x4* src = xsrc + (long)chunk * 2 * xd_plane;
#pragma unroll
        for (int i = 0; i < PER; ++i) regs[i] = src[goff[i]];
    };
    auto deposit = [&](int buf, const u32x4 (&regs)[PER]) {
#pragma unroll
        for (int i = 0; i < PER; ++i) *reinterpret_cast<u32x4*>(smem + buf * BUF + loff[i]) = regs[i];
    };

    f32x16 acc[FT][NSUB];
#pragma unroll
    for (int t2 = 0; t2 < FT; ++t2)
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t2][j][i] = 0.0f;

    // A fragments: lane (nn, h) of tile t2 reads 16 bytes of channel fb*128 + fw*32*FT + t2*32 + nn
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.wd) + ((long)(fb * kDFB + fw * 32 * FT + nn)) * 2 + h;
    const long wtap = (long)a.CoutP * 2;                      // bf16x8 units per tap
    const unsigned lane_base = (unsigned)(h * HALF + ((4 * pw + (nn >> 3)) * P + (nn & 7)) * 16);

    // Software pipeline (one wave per SIMD: nothing else hides a latency): the A fragments of tap t+4 are requested while
    // tap t runs (five register buffers; 4 x 448 MFMA cycles cover an L2 / Infinity-Cache round trip), the B fragments of
    // tap t+1 are read from LDS while tap t runs (two buffers).  Taps are consecutive in WD across rows, and across
    // chunks, so the A stream is one running pointer; the last chunk's look-ahead reads the padding tap behind WD.
    u32x4 win[PER];
    fetch(0, win);
    deposit(0, win);
    // A buffers: a divisor of the taps per row (the ring index restarts per row); nine taps per row: a whole row ahead where the
    // registers allow it (seven accumulator tiles: 236 of 256), else two taps ahead
#ifndef DAU_DENSE_RING9
#define DAU_DENSE_RING9 1
#endif
    constexpr int kRing = kDK % 5 == 0 ? 5 : kDK % 6 == 0 ? 6 : kDK == 7 ? 7 : (DAU_DENSE_RING9 && NSUB * FT <= 7 ? 9 : 3);
    static_assert(kDK % kRing == 0, "A ring");
    bf16x8 af[kRing][FT];
    const bf16x8* wp = wsrc;                                  // tap 0 of chunk 0
#pragma unroll
    for (int i = 0; i < kRing - 1; ++i)
#pragma unroll
        for (int t2 = 0; t2 < FT; ++t2) af[i][t2] = wp[i * wtap + 64 * t2];
    wp += (kRing - 1) * wtap;                                 // next tap to request
    __syncthreads();
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < a.nchunk) fetch(chunk + 1, win);      // lands under this chunk's hundred taps
        const unsigned bbase = lane_base + buf * BUF;
        bf16x8 bf[2][NSUB];
#pragma unroll
        for (int j = 0; j < NSUB; ++j) bf[0][j] = *reinterpret_cast<const bf16x8*>(smem + bbase + (8 * j) * 16);
        // one row of taps; PH = which of the two B buffers its first tap reads (rows of an odd number of taps alternate)
        auto row = [&](int ty, auto ph) {
            constexpr int PH = decltype(ph)::value;
            const unsigned brow = bbase + ty * P * 16;
#pragma unroll
            for (int tx = 0; tx < kDK; ++tx) {
                constexpr int kAhead = kRing - 1;
                const int cur = tx % kRing, nxt = (tx + kAhead) % kRing, pb = (tx + PH) & 1;
#pragma unroll
                for (int t2 = 0; t2 < FT; ++t2) af[nxt][t2] = wp[64 * t2];
                wp += wtap;
                // B fragments of the next tap of this chunk (the last tap of a chunk has no successor in this window)
                if (tx + 1 < kDK) {
#pragma unroll
                    for (int j = 0; j < NSUB; ++j) bf[pb ^ 1][j] = *reinterpret_cast<const bf16x8*>(smem + brow + (tx + 1 + 8 * j) * 16);
                } else if (ty + 1 < kDK) {
#pragma unroll
                    for (int j = 0; j < NSUB; ++j) bf[pb ^ 1][j] = *reinterpret_cast<const bf16x8*>(smem + brow + P * 16 + (8 * j) * 16);
                }
#pragma unroll
                for (int j = 0; j < NSUB; ++j)
#pragma unroll
                    for (int t2 = 0; t2 < FT; ++t2)
                        acc[t2][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][t2], bf[pb][j], acc[t2][j], 0, 0, 0);
            }
        };
        if constexpr (kDK & 1) {
#pragma unroll 1
            for (int ty = 0; ty + 1 < kDK; ty += 2) {
                row(ty, std::integral_constant<int, 0>{});
                row(ty + 1, std::integral_constant<int, 1>{});
            }
            row(kDK - 1, std::integral_constant<int, 0>{});      // (the next chunk starts with a fresh read into buffer 0)
        } else {
#pragma unroll 1
            for (int ty = 0; ty < kDK; ++ty) row(ty, std::integral_constant<int, 0>{});
        }
        if (chunk + 1 < a.nchunk) {
            deposit(buf ^ 1, win);                             // nobody reads that buffer during this chunk
            __syncthreads();
        }
    }

    // epilogue: C/D layout of the 32x32 tile: column (pixel) = lane & 31, row (channel) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const int y = rb * kDRows + 4 * pw + (nn >> 3);
    const long plane = (long)a.H * a.W;
#pragma unroll
    for (int t2 = 0; t2 < FT; ++t2)
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            const int x = cb * NSUB * 8 + 8 * j + (nn & 7);
            if (y < a.H && x < a.W) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int f = fb * kDFB + fw * 32 * FT + t2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (f < a.Cout) store_act(a.out, ((long)n * a.Cout + f) * plane + (long)y * a.W + x, acc[t2][j][i], a.out_bf16 != 0, false);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

template <int NSUB, int FT>
void launch_dense(hipStream_t st, const DenseArgs* a, int grid) {
    constexpr size_t lds = 2 * 2 * (size_t)(kDRows + kDSpan) * lds_pitch(NSUB) * 16;
    auto kern = dense_gather_kernel<NSUB, FT>;
    if (!a) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FT == 2 ? 256 : 512), lds, st, *a);
}

void dispatch_dense(int nsub, int ft, hipStream_t st, const DenseArgs* a, int grid) {
    if (ft == 1) {
        switch (nsub) {
            case 2: launch_dense<2, 1>(st, a, grid); break;
            case 4: launch_dense<4, 1>(st, a, grid); break;
            case 7: launch_dense<7, 1>(st, a, grid); break;
            default: launch_dense<8, 1>(st, a, grid); break;
        }
        return;
    }
#ifdef DAU_TUNING                 // the four-wave form (DAU_DENSE_FT=2) is a tuning alternative: not instantiated in the release build
    switch (nsub) {
        case 2: launch_dense<2, 2>(st, a, grid); break;
        case 4: launch_dense<4, 2>(st, a, grid); break;
        case 7: launch_dense<7, 2>(st, a, grid); break;
        default: launch_dense<8, 2>(st, a, grid); break;
    }
#endif
}

void stage_tile(const DenseConfig& c, const DenseGeom& g, int* TR, int* TC, size_t* lds) {
    // tile of at most 5 * 256 staged positions (the kernel keeps its results in registers) whose raw window of four
    // channels and horizontally filtered rows fit ~50 KiB of LDS (three workgroups per CU)
    // The kernel is latency bound: small tiles (about 22 x 22 positions, ~22 KiB of LDS, six workgroups per CU) beat big
    // ones although they filter more halo -- 65 x 65 planes: 16 x 65 tiles 1.32 ms, 22 x 22 tiles 0.70 ms (same box).
    int tr = (g.Hs + (g.Hs + 23) / 24 - 1) / ((g.Hs + 23) / 24), tc = (g.Ws + (g.Ws + 23) / 24 - 1) / ((g.Ws + 23) / 24);
    const int want_tr = DAU_TUNE_INT("DAU_DENSE_STAGE_TR", 0), want_tc = DAU_TUNE_INT("DAU_DENSE_STAGE_TC", 0);
    if (want_tr > 0) tr = want_tr;
    if (want_tc > 0) tc = want_tc;
    auto bytes = [&](int r, int cc) { return (size_t)4 * (r + c.blur_k - 1) * ((cc + c.blur_k - 1) + cc) * 4; };
    while ((bytes(tr, tc) > 52 * 1024 || tr * tc > 5 * 256) && tr > 4) tr -= 4;
    while ((bytes(tr, tc) > 52 * 1024 || tr * tc > 5 * 256) && tc > 16) tc -= 16;
    *TR = tr; *TC = tc; *lds = bytes(tr, tc);
}

const void* stage_rows_for(int blur_k) {
    switch (blur_k) {
        case 3: return reinterpret_cast<const void*>(dense_stage_rows_kernel<3>);
        case 5: return reinterpret_cast<const void*>(dense_stage_rows_kernel<5>);
        case 7: return reinterpret_cast<const void*>(dense_stage_rows_kernel<7>);
        case 9: return reinterpret_cast<const void*>(dense_stage_rows_kernel<9>);
        default: return nullptr;                 // wider prefilters: dense_stage_kernel
    }
}
// bands of rows whose raw window (eight channels, rows of kSP) stays below ~48 KiB of LDS: three workgroups per CU
void stage_rows_plan(const DenseConfig& c, int* RB, int* nbands, size_t* lds) {
    const int kr = (c.blur_k - 1) / 2;
    int rows = 48 * 1024 / (8 * kSP * 2) - 2 * kr;
    const int rb = DAU_TUNE_INT("DAU_DENSE_STAGE_RB", 0);
    if (rb > 0) rows = rb;
    rows = rows < 4 ? 4 : rows;
    const int nb = (c.H + rows - 1) / rows;
    *nbands = nb; *RB = (c.H + nb - 1) / nb;
    *lds = (size_t)(*RB + 2 * kr) * 8 * kSP * 2;
}

}  // namespace

bool dense_gather_configure(int N, int Cin, int Cout, int G, int H, int W, int R, int blur_k, bool bf16, DenseConfig* cfg) {
    if (R != kDR || !bf16) return false;          // the 9 x 9 dense kernel covers offsets within +-4; bf16 layers only
    DenseConfig c{};
    c.N = N; c.Cin = Cin; c.Cout = Cout; c.G = G; c.H = H; c.W = W; c.R = R; c.blur_k = blur_k; c.bf16 = 1;
    const DenseGeom g = dense_geometry(c);
    c.nsub = g.nsub;
    // accumulator tiles per wave: 1 = eight waves per workgroup, two per SIMD (default: 3.71 ms against 4.24 ms with four
    // waves at BASELINE config 2, same box); DAU_DENSE_FT=2 at plan creation selects the four-wave form
    c.ftiles = DAU_TUNE_INT("DAU_DENSE_FT", 1);
    if (c.ftiles != 2) c.ftiles = 1;
    int tr, tc; size_t lds;
    stage_tile(c, g, &tr, &tc, &lds);
    if (lds > 150 * 1024) return false;
    // 32-bit unit offsets inside one image's staged planes
    if ((size_t)g.nchunk * 2 * g.Hs * g.Ws > (size_t)1 << 30) return false;
    *cfg = c;
    return true;
}

size_t dense_gather_workspace_bytes(const DenseConfig& c) {
    const DenseGeom g = dense_geometry(c);
    return g.xd_bytes + g.wd_bytes;
}

void dense_gather_init(const DenseConfig& c) {
    dispatch_dense(c.nsub, c.ftiles, nullptr, nullptr, 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (stage_rows_for(c.blur_k)) (void)hipFuncSetAttribute(stage_rows_for(c.blur_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void dense_gather_prepare(hipStream_t st, const DenseConfig& c, const float* in, const float* filters, bool mirrored,
                          const UnitRef* table, void* workspace, const Guard& guard) {
    const DenseGeom g = dense_geometry(c);
    char* ws = static_cast<char*>(workspace);
    __bf16* xd = reinterpret_cast<__bf16*>(ws);
    __bf16* wd = reinterpret_cast<__bf16*>(ws + g.xd_bytes);
    {
        const long total = (long)g.nchunk * kDTaps * g.CoutP * 16;
        const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        if (DAU_TUNE_INT("DAU_DENSE_SCATTER", 1))
            hipLaunchKernelGGL(densify_units_scatter_kernel, dim3(g.nchunk * (g.CoutP / (kScT / 16))), dim3(kScT), 0, st, table, c.Cin, c.G, c.Cout,
                               g.CoutP, g.nchunk, wd, guard);
        else
            hipLaunchKernelGGL(densify_units_kernel, dim3(grid), dim3(256), 0, st, table, c.Cin, c.G, c.Cout, g.CoutP, g.nchunk, wd, guard);
    }
    if (stage_rows_for(c.blur_k) && c.bf16 && DAU_TUNE_INT("DAU_DENSE_STAGE_FAST", 1)) {
        DenseStageRowsArgs s{};
        s.in = reinterpret_cast<const unsigned short*>(in); s.taps = filters + kTaps1dOffset; s.xd = xd;
        s.N = c.N; s.C = c.Cin; s.H = c.H; s.W = c.W; s.mirrored = mirrored ? 1 : 0;
        s.Hs = g.Hs; s.Ws = g.Ws; s.nchunk = g.nchunk; s.guard = guard;
        size_t lds;
        stage_rows_plan(c, &s.RB, &s.nbands, &lds);
        s.nsegs = (c.W + 63) / 64;
        s.vec = c.W % 8 == 0 && reinterpret_cast<uintptr_t>(in) % 16 == 0;
        const int threads = DAU_TUNE_INT("DAU_DENSE_STAGE_THREADS", 256);
        void* args[] = {&s};
        (void)hipLaunchKernel(stage_rows_for(c.blur_k), dim3(c.N * 2 * g.nchunk * s.nbands * s.nsegs), dim3(threads), args, lds, st);
    } else {
        DenseStageArgs s{};
        s.in = in; s.taps = filters + kTaps1dOffset; s.xd = xd;
        s.N = c.N; s.C = c.Cin; s.H = c.H; s.W = c.W; s.k = c.blur_k; s.mirrored = mirrored ? 1 : 0; s.bf16 = c.bf16;
        s.Hs = g.Hs; s.Ws = g.Ws; s.nchunk = g.nchunk; s.guard = guard;
        size_t lds;
        stage_tile(c, g, &s.TR, &s.TC, &lds);
        s.ntr = (g.Hs + s.TR - 1) / s.TR; s.ntc = (g.Ws + s.TC - 1) / s.TC;
        hipLaunchKernelGGL(dense_stage_kernel, dim3(c.N * 2 * g.nchunk * s.ntr * s.ntc), dim3(256), lds, st, s);
    }
}

void dense_gather_run(hipStream_t st, const DenseConfig& c, float* out, void* workspace, const Guard& guard) {
    const DenseGeom g = dense_geometry(c);
    char* ws = static_cast<char*>(workspace);
    const __bf16* xd = reinterpret_cast<const __bf16*>(ws);
    const __bf16* wd = reinterpret_cast<const __bf16*>(ws + g.xd_bytes);
    DenseArgs a{};
    a.xd = xd; a.wd = wd; a.out = out;
    a.N = c.N; a.Cout = c.Cout; a.CoutP = g.CoutP; a.H = c.H; a.W = c.W; a.Hs = g.Hs; a.Ws = g.Ws; a.nchunk = g.nchunk;
    a.ncb = g.ncb; a.nrb = g.nrb; a.out_bf16 = c.bf16; a.guard = guard;
    const int grid = c.N * g.nrb * g.ncb * (g.CoutP / kDFB);
    dispatch_dense(c.nsub, c.ftiles, st, &a, grid);
}

}  // namespace DAU_DENSE_NS
}  // namespace dau
