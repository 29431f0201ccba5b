// Densified parameter gradients on the bf16 matrix cores (DAU_FLAG_DENSE_BF16; offsets within +-R, three or more units).  The file is
// compiled for R = 4 and for R = 3 (below); the numbers in this header are those of R = 4 (R = 3: 49 displacements, 196 GEMMs).
//
//   r_k[s,g,f] = sum_{n,p} E'[n,f,p] * sum_{t in 2x2} b_t(s,g,f) * Xk[n,s, p + o + t]          k = w, mu1, mu2, sigma
//             = sum_t b_t * C_k[o + t][s][f],      C_k[d][s][f] = sum_{n,p} Xk[n,s,p+d] * E'[n,f,p],   d in [-4, 4]^2 (the second tap of an offset of exactly +4 has weight 0)
//
// The 81 cross-correlations C_k[d] do not depend on the units: 324 GEMMs  M = input channels, N = output channels,
// K = (image, position)  on v_mfma_f32_32x32x16_bf16, 2*324*N*H*W*S*F FLOP whatever the unit count -- at the rate of the
// dense gather-sum (k_dense_bf16.hip) that is the time the exact gather-dot (k_gather_dot.hip) needs for four units, so the
// form is used from three units on (BASELINE config 2 has six).  It replaces the same reference code as the gather-dot:
// DAUConv_bwd_multi_pipeline_kernel and its three preparation kernels
// (include/dau_conv/dau_conv_impl/dau_conv_backward_core.hpp:1017-1820, 1824-2380) and the 4-filter prefilter pass
// (src/dau_conv/util/convolve.cu:48-131).  Numerics: Xk and E' are rounded to bfloat16, products are exact, sums are fp32:
// the 2e-2 bar of the bf16 configuration (opt-in with DAU_FLAG_DENSE_BF16, as the dense gather-sum).
//
// K runs over the IMAGES innermost, so that a displacement only changes the position and every matrix fragment is one
// aligned KiB whatever d is:
//   XkT[k][sb][nc][H+8][WsT][2][32 s][8 n] bf16 the four derivative-filtered copies of x, staged position (r, c) = image
//                                               (r-R, c-R), zero outside the image (wg_transpose_x + wg_filter<K> from x; prefilters
//                                               wider than 9 taps: wg_stage_x from blur4_pack's fp32 copy)
//   ET [fb][nc][H][WT'][32 f][16 n]      bf16   the error (unit_testing edge rule applied), zero for columns W..WT'-1 (WT' = whole
//                                               row segments of an instantiated length)
//   C  [split][k][9][9][SB*32][FB*32]    fp32   partial correlations of one range of image chunks
// wg_gemm: workgroup = (32 input channels, kind k, row displacement oy, range of image chunks) x 8 waves = 8 blocks of 32
// output channels; a wave keeps the nine column displacements ox as nine 32 x 32 accumulators and walks (image chunk, row,
// column): per column one new Xk fragment (a window of nine slides along the row; the row sits in LDS, loaded by
// global_load_lds one row ahead and shared by the eight waves), one E' fragment (from global memory, six columns ahead)
// and nine MFMAs.
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "dau_tiled.hpp"

#ifndef DAU_DENSE_R
#define DAU_DENSE_R 4          // compiled once per radius, as k_dense_bf16.hip: namespaces r4 (|mu| <= 4) and r3 (|mu| <= 3: 49 displacements)
#endif
#ifndef DAU_DENSE_NS
#define DAU_DENSE_NS r4
#endif

namespace dau {
namespace DAU_DENSE_NS {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

namespace {

constexpr int kWR = DAU_DENSE_R;   // offset radius
constexpr int kWD = 2 * kWR + 1;   // displacements per axis: -R .. R (the tap at R + 1 belongs to an offset of exactly +R: fraction 0,
                                   // weight 0 -- wg_finish_kernel never reads it)
constexpr int kWMaxSteps = 60;     // widest row (columns per row are straight-line code: see wg_gemm_kernel)
constexpr int kWAhead = 5;         // E' fragments in flight ahead of the one in use
constexpr int kWSlots = kWAhead + 1;
constexpr int kWWaves = 8;

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WgLayout { size_t xk_off, xkt_off, et_off, c_off, total; };
WgLayout wg_layout(const WgradConfig& c) {
    WgLayout l{};
    size_t off = 0;
    // fused staging: xT (one KiB per pixel of a 32-channel x 16-image block); else the fp32 copy of blur4_pack
    l.xk_off = off;  off += c.fused ? round_up((size_t)c.SB * c.NC * c.sh.H * c.sh.W * 1024, 256)
                                    : round_up((size_t)((c.sh.N + 1) / 2) * c.SB * 32 * c.Hp * c.Wp * 32, 256);
    l.xkt_off = off; off += round_up((size_t)kNumK * c.SB * c.NC * c.HsT * c.WsT * 1024, 256);
    l.et_off = off;  off += round_up((size_t)c.FB * c.NC * c.sh.H * c.nseg * c.WT * 1024 + (size_t)kWSlots * 1024, 256);   // + look-ahead past the end
    l.c_off = off;   off += round_up((size_t)c.splits * kNumK * kWD * kWD * c.SB * 32 * c.FB * 32 * 4, 256);
    l.total = off;
    return l;
}

// ------------------------------------------------------------------------------------------------
// staging
// ------------------------------------------------------------------------------------------------
// xk[NP][SB*32][Hp][Wp][4][2] fp32 -> XkT.  One workgroup per (sb, nc, staged row, run of 8 staged columns): the 8 x 32 bytes
// of 8 image pairs x 32 channels -> LDS -> 32 fragments (4 kinds x 8 positions) of [32 s][16 n] bf16.
struct WgStageXArgs {
    const float* xk;
    __bf16* xkt;
    int N, NP, SB, NC, H, W, Hp, Wp, HsT, WsT;
    Guard guard;
};
__global__ void __launch_bounds__(512) wg_stage_x_kernel(const WgStageXArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [pair*32 + s][8 positions x 8 floats | 1] = 65 KiB
    if (!guard_pass(a.guard)) return;
    int t = blockIdx.x;
    const int runs = a.WsT / 8;
    const int xr = t % runs; t /= runs;
    const int yy = t % a.HsT; t /= a.HsT;
    const int nc = t % a.NC;
    const int sb = t / a.NC;
    const int y = yy - kWR, x0 = xr * 8 - kWR;             // image coordinates of the run
    const bool row_in = y >= 0 && y < a.H;
    // load: 256 (pair, channel) rows of 64 floats; a thread takes float4 pieces (16 per row)
    for (int i = threadIdx.x; i < 256 * 16; i += blockDim.x) {
        const int row = i >> 4, piece = i & 15;            // piece: position = piece / 2, half of its 8 floats
        const int pr = row >> 5, s = row & 31;
        const int np = nc * 8 + pr, x = x0 + (piece >> 1);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (row_in && np < a.NP && x >= 0 && x < a.Wp)
            v = *reinterpret_cast<const float4*>(a.xk + ((((size_t)np * a.SB * 32 + sb * 32 + s) * a.Hp + y) * a.Wp + x) * 8 + (piece & 1) * 4);
        float* d = lds + row * 65 + piece * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // store: fragment (kind k, position p) = 512 bf16; a thread writes 32 of them = two channel rows of 16 images
    const int frag = threadIdx.x >> 4, part = threadIdx.x & 15;     // 32 fragments x 16 threads
    const int k = frag >> 3, p = frag & 7;
    __bf16* out = a.xkt + (((((size_t)k * a.SB + sb) * a.NC + nc) * a.HsT + yy) * a.WsT + xr * 8 + p) * 512;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int s = part * 2 + h;
        bf16x8 o0, o1;
#pragma unroll
        for (int n = 0; n < 8; ++n) {                       // images 0..7 of the chunk = pairs 0..3, images 8..15 = pairs 4..7
            o0[n] = (__bf16)lds[((n >> 1) * 32 + s) * 65 + p * 8 + k * 2 + (n & 1)];
            o1[n] = (__bf16)lds[((4 + (n >> 1)) * 32 + s) * 65 + p * 8 + k * 2 + (n & 1)];
        }
        // fragment order [half of the images][32 s][8 n]: the 16 lanes ds_read_b128 serves per cycle then read 256 contiguous
        // bytes (with [32 s][16 n] rows r and r+8 shared banks: half of the kernel's LDS cycles were two-way conflicts)
        *reinterpret_cast<bf16x8*>(out + s * 8) = o0;
        *reinterpret_cast<bf16x8*>(out + 256 + s * 8) = o1;
    }
}

// ------------------------------------------------------------------------------------------------
// staging, fused form (instantiated prefilter supports): x -> xT (images innermost, bf16 as stored) -> XkT, without the fp32 copy.
//   wg_transpose_x_kernel  x[N,S,H,W] bf16 -> xT[sb][nc][H][W][2][32 s][8 n]: the fragment order of XkT, one KiB per pixel
//   wg_filter_kernel<K>    the four derivative filters applied IN that order: a fragment of XkT is an elementwise combination of
//                          K x K fragments of xT, so every load and store is a contiguous half fragment (8 bytes per lane) whatever
//                          the tap.  A wave owns (32 channels x 16 images, column, half of the images) and walks down the rows:
//                          horizontal pass over the K neighbouring columns (read through L1/L2: the waves of the neighbouring columns
//                          run on the same XCD), vertical pass over a register ring of K rows of the three horizontal results.
// The sums run in the order of blur4_pack_kernel and are rounded to bfloat16 once, as wg_stage_x_kernel did: bit-identical XkT.
// ------------------------------------------------------------------------------------------------
struct WgTransposeArgs {
    const unsigned short* x;
    unsigned short* xt;
    int N, S, SB, NC, H, W, nruns, vec;
    Guard guard;
};
constexpr int kTPitch = 66;        // LDS row of one (image, channel): 64 columns + 2 (33 dwords: the 32 channels of a column read
                                   // 32 different banks)
__global__ void __launch_bounds__(512) wg_transpose_x_kernel(const WgTransposeArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[512 * kTPitch];   // [n*32 + s][column]
    if (!guard_pass(a.guard)) return;
    int t = blockIdx.x;
    const int run = t % a.nruns; t /= a.nruns;
    const int y = t % a.H; t /= a.H;
    const int nc = t % a.NC;
    const int sb = t / a.NC;
    const int x0 = run * 64, cols = x0 + 64 < a.W ? 64 : a.W - x0;
    // load: 512 rows of up to 64 values = 8 pieces of 16 bytes; consecutive lanes take consecutive pieces of a row
    {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = threadIdx.x + u * 512, row = i >> 3, q = i & 7;
            const int n = nc * 16 + (row >> 5), s = sb * 32 + (row & 31), xs = x0 + q * 8;
            const bool in = n < a.N && s < a.S;
            const unsigned short* src = a.x + (((long)(in ? n : 0) * a.S + (in ? s : 0)) * a.H + y) * a.W;
            if (a.vec) {                                         // branch free: clamped address, masked value
                const bool ok = in && xs + 8 <= a.W;
                const uint4 w = *reinterpret_cast<const uint4*>(src + (ok ? xs : 0));
                const unsigned m = ok ? 0xffffffffu : 0u;
                v[u] = make_uint4(w.x & m, w.y & m, w.z & m, w.w & m);
            } else {
                unsigned e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool ok = in && xs + k < a.W;
                    e[k] = (unsigned)src[ok ? xs + k : 0] & (ok ? 0xffffu : 0u);
                }
                v[u] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = threadIdx.x + u * 512, row = i >> 3, q = i & 7;
            unsigned* d = reinterpret_cast<unsigned*>(lds + row * kTPitch + q * 8);    // 4-byte aligned
            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
    }
    __syncthreads();
    // store: a wave writes one fragment per column: lane = (half, s) writes the 8 images of its half = 16 bytes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, s = lane & 31;
    for (int c = wave; c < cols; c += 8) {
        unsigned e[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) e[n] = lds[((half * 8 + n) * 32 + s) * kTPitch + c];
        uint4 o = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
        *reinterpret_cast<uint4*>(a.xt + ((((long)sb * a.NC + nc) * a.H + y) * a.W + x0 + c) * 512 + lane * 8) = o;
    }
}

struct WgFilterArgs {
    const unsigned short* xt;
    unsigned short* xkt;
    const float* taps;
    int SB, NC, H, W, HsT, WsT;
    int RB, nbands, ncolblk, nblocks;   // rows per band, bands, blocks of 2 staged columns, workgroups that have work
    int nk;                             // kinds wanted: 4, or 3 = without the sigma kind (the last one)
    Guard guard;
};

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const __bf16 l = (__bf16)lo, h = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, l) | (unsigned)__builtin_bit_cast(unsigned short, h) << 16;
}

template <int K>
__global__ void __launch_bounds__(256) wg_filter_kernel(const WgFilterArgs a) {
    if (!guard_pass(a.guard)) return;
    constexpr int kr = (K - 1) / 2;
    // workgroups go to the eight XCDs round robin: give every XCD a contiguous range of the logical order, so that the column blocks
    // of one (sb, nc, band) share an L2
    const int per = gridDim.x >> 3;                              // the grid is a multiple of 8
    int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= a.nblocks) return;
    const int cb = t % a.ncolblk; t /= a.ncolblk;
    const int band = t % a.nbands; t /= a.nbands;
    const int nc = t % a.NC;
    const int sb = t / a.NC;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // everything below but `voff` is wave-uniform: scalar registers
    const int half = wave & 1, cs = cb * 2 + (wave >> 1);        // staged column; image column x = cs - 4
    if (cs >= a.WsT) return;
    const int x = cs - kWR;
    const int y0 = band * a.RB, y1 = y0 + a.RB < a.H ? y0 + a.RB : a.H;
    const int rs0 = band == 0 ? 0 : y0 + kWR, rs1 = band == a.nbands - 1 ? a.HsT : y1 + kWR;   // staged rows this wave writes
    const unsigned voff = (unsigned)(half * 512 + lane * 8);     // this lane's 8 bytes of a fragment
    const long kstride = (long)a.SB * a.NC * a.HsT * a.WsT * 1024;                         // one kind of XkT (bytes)
    char* out = reinterpret_cast<char*>(a.xkt) + ((((long)sb * a.NC + nc) * a.HsT) * a.WsT + cs) * 1024;
    const long rstride = (long)a.WsT * 1024;
    const bool want_sigma = a.nk > 3;
    auto store4 = [&](int r, uint2 w, uint2 m1, uint2 m2, uint2 sg) {
        char* o = out + (long)r * rstride;
        *reinterpret_cast<uint2*>(o + voff) = w;
        *reinterpret_cast<uint2*>(o + kstride + voff) = m1;
        *reinterpret_cast<uint2*>(o + 2 * kstride + voff) = m2;
        if (want_sigma) *reinterpret_cast<uint2*>(o + 3 * kstride + voff) = sg;
    };
    const uint2 zero = make_uint2(0u, 0u);
    if (x < 0 || x >= a.W) {                                     // border column
        for (int r = rs0; r < rs1; ++r) store4(r, zero, zero, zero, zero);
        return;
    }
    for (int r = rs0; r < y0 + kWR; ++r) store4(r, zero, zero, zero, zero);
    for (int r = y1 + kWR; r < rs1; ++r) store4(r, zero, zero, zero, zero);
    float gxt[K], axt[K], cxt[K], gyt[K], ayt[K], byt[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        gxt[j] = a.taps[kTapGX * kTapPitch + j]; axt[j] = a.taps[kTapAX * kTapPitch + j]; cxt[j] = a.taps[kTapCX * kTapPitch + j];
        gyt[j] = a.taps[kTapGY * kTapPitch + j]; ayt[j] = a.taps[kTapAY * kTapPitch + j]; byt[j] = a.taps[kTapBY * kTapPitch + j];
    }
    // the K columns of the horizontal pass: clamped into the image, zero through a mask where they fall outside
    const char* xbase = reinterpret_cast<const char*>(a.xt) + (((long)sb * a.NC + nc) * a.H) * a.W * 1024;
    int colx[K];
    unsigned cmask[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int xj = x + j - kr;
        const bool in = xj >= 0 && xj < a.W;
        colx[j] = in ? xj : x;
        cmask[j] = in ? 0xffffffffu : 0u;
    }
    // input row s = image row y0 - kr + s; rows outside the image give zero horizontal results (their loads go to a valid row)
    auto load_row = [&](int s, uint2 (&v)[K]) {
        int yy = y0 - kr + s;
        yy = yy < 0 ? 0 : (yy >= a.H ? a.H - 1 : yy);
        const char* rowp = xbase + (long)yy * a.W * 1024;
#pragma unroll
        for (int j = 0; j < K; ++j) v[j] = *reinterpret_cast<const uint2*>(rowp + (long)colx[j] * 1024 + voff);
    };
    float ring[K][3][4];
    auto horizontal = [&](int s, const uint2 (&cur)[K], float (&h)[3][4]) {
        const int yy = y0 - kr + s;
        const unsigned rmask = (yy >= 0 && yy < a.H) ? 0xffffffffu : 0u;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) h[q][e] = 0.0f;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const unsigned m = cmask[i] & rmask;
            const unsigned lo = cur[i].x & m, hi = cur[i].y & m;
            const float v[4] = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                                __uint_as_float(hi & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[0][e] = fmaf(v[e], gxt[i], h[0][e]);
                h[1][e] = fmaf(v[e], axt[i], h[1][e]);
                h[2][e] = fmaf(v[e], cxt[i], h[2][e]);
            }
        }
    };
    // two rows of loads in flight ahead of the one in use (a row's arithmetic is shorter than a trip to the L2)
    uint2 nxt[K], nx2[K];
    load_row(0, nxt);
    load_row(1, nx2);
    // prologue: the first K - 1 input rows only fill the ring
#pragma unroll
    for (int s = 0; s < K - 1; ++s) {
        uint2 cur[K];
#pragma unroll
        for (int i = 0; i < K; ++i) { cur[i] = nxt[i]; nxt[i] = nx2[i]; }
        load_row(s + 2, nx2);
        horizontal(s, cur, ring[s]);
    }
    // K rows per trip: ring slot and tap order are compile-time; rows past the band are computed and not stored (RB is a multiple of K
    // wherever the image allows)
    const int nout = y1 - y0;
    for (int o0 = 0; o0 < nout; o0 += K) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int s = o0 + j + K - 1;                        // input row; output row o0 + j
            const int slot = (K - 1 + j) % K;
            uint2 cur[K];
#pragma unroll
            for (int i = 0; i < K; ++i) { cur[i] = nxt[i]; nxt[i] = nx2[i]; }
            load_row(s + 2, nx2);
            horizontal(s, cur, ring[slot]);
            float dw[4] = {0.0f, 0.0f, 0.0f, 0.0f}, d1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, d2[4] = {0.0f, 0.0f, 0.0f, 0.0f},
                  ds[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int q = (slot + 1 + i) % K;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dw[e] = fmaf(ring[q][0][e], gyt[i], dw[e]);
                    d1[e] = fmaf(ring[q][1][e], gyt[i], d1[e]);
                    d2[e] = fmaf(ring[q][0][e], ayt[i], d2[e]);
                    ds[e] = fmaf(ring[q][2][e], gyt[i], ds[e]);
                    ds[e] = fmaf(ring[q][0][e], byt[i], ds[e]);
                }
            }
            if (o0 + j < nout)
                store4(y0 + o0 + j + kWR, make_uint2(pack_bf16x2(dw[0], dw[1]), pack_bf16x2(dw[2], dw[3])),
                       make_uint2(pack_bf16x2(d1[0], d1[1]), pack_bf16x2(d1[2], d1[3])),
                       make_uint2(pack_bf16x2(d2[0], d2[1]), pack_bf16x2(d2[2], d2[3])),
                       make_uint2(pack_bf16x2(ds[0], ds[1]), pack_bf16x2(ds[2], ds[3])));
        }
    }
}

// dy[N,F,H,W] bf16 -> ET.  One workgroup per (fb, nc, row, run of 64 columns): 512 (image, channel) rows of 64 values (16-byte
// pieces, as wg_transpose_x_kernel) -> LDS -> one fragment of [32 f][16 n] per column.  The unit_testing edge rule (last column / row
// of the error dropped) is applied here; the columns W .. WT'-1 are written as zeros.
struct WgStageEArgs {
    const unsigned short* dy;
    __bf16* et;
    int N, F, FB, NC, H, W, WT, drop_col, drop_row, nruns, vec;
    Guard guard;
};
__global__ void __launch_bounds__(512) wg_stage_e_kernel(const WgStageEArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[512 * kTPitch];   // [n*32 + f][column]
    if (!guard_pass(a.guard)) return;
    int t = blockIdx.x;
    const int run = t % a.nruns; t /= a.nruns;
    const int y = t % a.H; t /= a.H;
    const int nc = t % a.NC;
    const int fb = t / a.NC;
    const int x0 = run * 64;
    const bool row_in = !(a.drop_row && y == a.H - 1);
    const int wlim = a.drop_col ? a.W - 1 : a.W;
    {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = threadIdx.x + u * 512, row = i >> 3, q = i & 7;
            const int n = nc * 16 + (row >> 5), f = fb * 32 + (row & 31), xs = x0 + q * 8;
            const bool in = row_in && n < a.N && f < a.F;
            const unsigned short* src = a.dy + (((long)(in ? n : 0) * a.F + (in ? f : 0)) * a.H + y) * a.W;
            if (a.vec) {
                const bool ok = in && xs + 8 <= a.W;
                const uint4 w = *reinterpret_cast<const uint4*>(src + (ok ? xs : 0));
                unsigned d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {                    // elements 2k, 2k+1: columns xs + 2k, xs + 2k + 1
                    const unsigned m = (ok && xs + 2 * k < wlim ? 0xffffu : 0u) | (ok && xs + 2 * k + 1 < wlim ? 0xffff0000u : 0u);
                    d[k] &= m;
                }
                v[u] = make_uint4(d[0], d[1], d[2], d[3]);
            } else {
                unsigned e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool ok = in && xs + k < wlim;
                    e[k] = (unsigned)src[ok ? xs + k : 0] & (ok ? 0xffffu : 0u);
                }
                v[u] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = threadIdx.x + u * 512, row = i >> 3, q = i & 7;
            unsigned* d = reinterpret_cast<unsigned*>(lds + row * kTPitch + q * 8);
            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
    }
    __syncthreads();
    // fragment of column c: [32 f][16 n]; lane = (f, half of the images) writes 16 bytes, a wave one fragment
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = lane >> 1, nh = lane & 1;
    const int cols = x0 + 64 < a.WT ? 64 : a.WT - x0;
    for (int c = wave; c < cols; c += 8) {
        unsigned e[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) e[n] = lds[((nh * 8 + n) * 32 + f) * kTPitch + c];
        const uint4 o = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(a.et) + ((((long)fb * a.NC + nc) * a.H + y) * a.WT + x0 + c) * 512 + lane * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------------
struct WgGemmArgs {
    const char* xkt;
    const char* et;
    float* c;
    int SB, FB, NC, H, HsT, WsT, WT, nseg, rowf, splits, fgroups;   // WT: columns of a row segment, rowf: its Xk fragments (WT + 8)
    int nk;                                                         // kinds computed: the first nk of kNumK
    Guard guard;
};

// E' fragments come through inline asm with counted waits (vector memory returns in order): hipcc's own bookkeeping waited for
// all but one of them.  The row of a wave is straight-line code, so that no register with a load in
// flight ever crosses a loop back-edge (where a register copy would read it too early); the only back-edge is the row loop,
// which drains everything for its barrier anyway.  NSTEP = columns of a row = the instantiation (window slot = column mod 10
// and E' slot = column mod 6 are then compile-time constants for any row length).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// (the base goes through an s_mov_b64 inside the block: an SGPR that the register allocator reloads by v_readlane_b32 right before
// the block must not be read by a vector memory instruction within 5 cycles -- see x_load in k_gather_dot.hip)
#define WG_ELOAD(dst, voff, sbase)                                                                                         \
    do {                                                                                                                   \
        unsigned long long sb_;                                                                                            \
        asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1" : "=v"(dst), "=&s"(sb_) : "v"(voff), "s"(sbase) : "memory"); \
    } while (0)
template <int N>
__device__ __forceinline__ void wg_vmwait() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
constexpr int kWDma = 10;          // global_load_lds instructions per wave and row (rows of up to 80 fragments; short rows repeat
                                   // their last piece, so that the counted waits can step over a constant number)

template <int NSTEP>
__global__ void __launch_bounds__(kWWaves * 64) wg_gemm_kernel(const WgGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two row segments of WT + 8 Xk fragments
    if (!guard_pass(a.guard)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int t = blockIdx.x;
    const int sb = t % a.SB; t /= a.SB;
    const int fg = t % a.fgroups; t /= a.fgroups;
    const int oy = t % kWD; t /= kWD;
    const int k = t % a.nk;
    const int split = t / a.nk;
    const int per = (a.NC + a.splits - 1) / a.splits;
    const int nc0 = split * per, nc1 = nc0 + per < a.NC ? nc0 + per : a.NC;
    const int fb = fg * kWWaves + wave;
    const bool active = fb < a.FB;
    const int fbc = active ? fb : a.FB - 1;                         // idle waves read a valid block and store nothing

    f32x16 acc[kWD];
#pragma unroll
    for (int i = 0; i < kWD; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;

    // a "row" of the walk = one segment of WT columns of an image row (rows of more than 60 pixels are cut into segments;
    // a segment needs the WT + 8 staged columns from its first one)
    const unsigned row_bytes = (unsigned)a.rowf * 1024;
    const int per_chunk = a.H * a.nseg;
    const int rows = (nc1 - nc0) * per_chunk;
    auto row_src = [&](int r) -> const char* {
        const int nc = nc0 + r / per_chunk, y = (r % per_chunk) / a.nseg, seg = r % a.nseg;
        return a.xkt + (((((size_t)k * a.SB + sb) * a.NC + nc) * a.HsT + (y + oy)) * a.WsT + (size_t)seg * a.WT) * 1024;
    };
    auto issue_row = [&](int r, int buf) {
        const char* src = row_src(r);
#pragma unroll
        for (int i = 0; i < kWDma; ++i) {
            int p = wave + i * kWWaves;
            p = p < a.rowf ? p : a.rowf - 1;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)p * 1024 + lane * 16),
                                             (lds_ptr_t)(smem + buf * row_bytes + p * 1024), 16, 0, 0);
        }
    };
    // a lane's 16 bytes of a [32][16] fragment (E'): row lane & 31, images 8 * (lane >> 5) .. + 7
    const unsigned lfrag = (unsigned)((lane & 31) * 32 + (lane >> 5) * 16);
    if (rows > 0) issue_row(0, 0);
    for (int r = 0; r < rows; ++r) {
        const int buf = r & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                            // row r has landed; everybody is done with row r-1
        const int nc = nc0 + r / per_chunk, y = (r % per_chunk) / a.nseg, seg = r % a.nseg;
        const char* arow = smem + buf * row_bytes + (unsigned)((lane >> 5) * 512 + (lane & 31) * 16);   // Xk fragments: [half][32][8]
        const char* erow = a.et + ((((size_t)fbc * a.NC + nc) * a.H + y) * a.nseg + seg) * (size_t)a.WT * 1024;     // wave-uniform
        bf16x8 win[kWD];
        u32x4 eq[kWSlots];
        // E' of the first columns, THEN the next row's Xk (kWDma instructions): the waits of the first kWAhead columns step over them
#pragma unroll
        for (int i = 0; i < kWAhead; ++i) WG_ELOAD(eq[i], lfrag, erow + i * 1024);
        if (r + 1 < rows) issue_row(r + 1, buf ^ 1);
        else issue_row(r, buf ^ 1);                                 // (keeps the number of operations in flight constant)
#pragma unroll
        for (int i = 0; i < kWD - 1; ++i) win[i] = *reinterpret_cast<const bf16x8*>(arow + i * 1024);
#pragma unroll
        for (int x = 0; x < NSTEP; ++x) {
            // the window holds staged columns x .. x+8 (slot = column mod 9), eq[x mod 6] = E'(x)
            win[(x + kWD - 1) % kWD] = *reinterpret_cast<const bf16x8*>(arow + (x + kWD - 1) * 1024);
            // No look-ahead past the row's end: a register that receives a load nobody will read is free for hipcc to give to
            // something else right after the asm statement -- and the data that lands in it later corrupts that (seen with seven
            // accumulators: the E' fragments of the next row's first columns ended up in the Xk window).  The counted waits follow
            // the loads that are really in flight behind E'(x).
            if (x + kWAhead < NSTEP) WG_ELOAD(eq[(x + kWAhead) % kWSlots], lfrag, erow + (x + kWAhead) * 1024);
            const int behind = NSTEP - 1 - x;                 // E' loads of this row issued after E'(x) (x is a constant once unrolled)
            static_assert(kWAhead == 5, "the tail below counts down from four");
            if (x < kWAhead) wg_vmwait<kWAhead + kWDma>();
            else if (behind >= kWAhead) wg_vmwait<kWAhead>();
            else if (behind == 4) wg_vmwait<4>();
            else if (behind == 3) wg_vmwait<3>();
            else if (behind == 2) wg_vmwait<2>();
            else if (behind == 1) wg_vmwait<1>();
            else wg_vmwait<0>();
            const bf16x8 e = __builtin_bit_cast(bf16x8, eq[x % kWSlots]);
#pragma unroll
            for (int i = 0; i < kWD; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(win[(x + i) % kWD], e, acc[i], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!active) return;
    // C[split][k][oy][ox][s][f]: accumulator register j of a lane = row (j&3) + 8*(j>>2) + 4*(lane>>5), column lane&31
    const int SP = a.SB * 32, FP = a.FB * 32;
    float* cb = a.c + ((((size_t)split * kNumK + k) * kWD + oy) * kWD) * SP * FP;
#pragma unroll
    for (int i = 0; i < kWD; ++i) {
        float* ci = cb + (size_t)i * SP * FP;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int srow = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
            ci[(size_t)(sb * 32 + srow) * FP + fb * 32 + (lane & 31)] = acc[i][j];
        }
    }
}

// r4[k][(s*G+g)*F+f] = sum over the image ranges and the four bilinear taps of the unit; one thread per (kind, unit).  The loads of
// four ranges (sixteen values) are issued together; the sums run in double in the fixed order (range, tap), whatever the batching.
__global__ void wg_finish_kernel(const float* __restrict__ c, const UnitRef* __restrict__ table, int S, int G, int F, int SP,
                                 int FP, int splits, int nk, float* __restrict__ r4, const Guard guard) {
    if (!guard_pass(guard)) return;
    const long units = (long)S * G * F;
    const size_t plane = (size_t)SP * FP;                    // one displacement
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < units * nk; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i / units);
        const long u = i - (long)k * units;
        const int f = (int)(u % F), s = (int)(u / ((long)F * G));
        const UnitRef ur = table[u];
        int dyi = ur.oy + kWR, dxi = ur.ox + kWR;
        dyi = dyi < 0 ? 0 : (dyi > kWD - 1 ? kWD - 1 : dyi);        // (a guarded call never clamps: offsets within +-4)
        dxi = dxi < 0 ? 0 : (dxi > kWD - 1 ? kWD - 1 : dxi);
        // displacement +5 = second tap of an offset of exactly +4: its weight is 0 and C does not hold it (the load goes to the
        // unit's first tap instead and is multiplied by a zero weight)
        const bool yin = dyi + 1 < kWD, xin = dxi + 1 < kWD;
        const float b[4] = {ur.w00, xin ? ur.w01 : 0.0f, yin ? ur.w10 : 0.0f, (xin && yin) ? ur.w11 : 0.0f};
        const size_t o00 = (size_t)(dyi * kWD + dxi) * plane;
        const size_t off[4] = {o00, xin ? o00 + plane : o00, yin ? o00 + kWD * plane : o00, (xin && yin) ? o00 + (kWD + 1) * plane : o00};
        const float* cu = c + (size_t)s * FP + f;
        double sum = 0.0;
        for (int sp0 = 0; sp0 < splits; sp0 += 4) {
            float v[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sp = sp0 + q < splits ? sp0 + q : splits - 1;          // (past the end: a valid range, not added)
                const float* ck = cu + (((size_t)sp * kNumK + k) * kWD * kWD) * plane;
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) v[q][tp] = ck[off[tp]];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (sp0 + q < splits)
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) sum += (double)b[tp] * (double)v[q][tp];
        }
        r4[i] = (float)sum;
    }
}

const void* wg_gemm_for(int wt) {
    switch (wt) {
        case 14: return reinterpret_cast<const void*>(wg_gemm_kernel<14>);
        case 28: return reinterpret_cast<const void*>(wg_gemm_kernel<28>);
        case 30: return reinterpret_cast<const void*>(wg_gemm_kernel<30>);
        case 42: return reinterpret_cast<const void*>(wg_gemm_kernel<42>);
        case 56: return reinterpret_cast<const void*>(wg_gemm_kernel<56>);
        default: return reinterpret_cast<const void*>(wg_gemm_kernel<60>);
    }
}

}  // namespace

namespace {
const void* wg_filter_for(int blur_k) {
    switch (blur_k) {
        case 3: return reinterpret_cast<const void*>(wg_filter_kernel<3>);
        case 5: return reinterpret_cast<const void*>(wg_filter_kernel<5>);
        case 7: return reinterpret_cast<const void*>(wg_filter_kernel<7>);
        case 9: return reinterpret_cast<const void*>(wg_filter_kernel<9>);
        default: return nullptr;                 // wider prefilters: blur4_pack_kernel + wg_stage_x_kernel
    }
}
}  // namespace

bool dense_wgrad_configure(const Shape& sh, int blur_k, bool bf16, WgradConfig* cfg) {
    if (!bf16) return false;
    WgradConfig c{};
    c.sh = sh; c.blur_k = blur_k;
    c.SB = (sh.S + 31) / 32; c.FB = (sh.F + 31) / 32; c.NC = (sh.N + 15) / 16;
    // instantiated segment lengths (wg_gemm_kernel<NSTEP>): the fewest padded columns, then the fewest segments
    c.WT = 0; c.nseg = 0;
    for (int w : {14, 28, 30, 42, 56, 60}) {
        const int n = (sh.W + w - 1) / w;
        if (c.WT == 0 || n * w < c.nseg * c.WT || (n * w == c.nseg * c.WT && n < c.nseg)) { c.WT = w; c.nseg = n; }
    }
    c.HsT = sh.H + kWD - 1;
    c.WsT = (int)round_up((size_t)c.nseg * c.WT + kWD - 1, 8);
    c.Hp = sh.H; c.Wp = (sh.W + 7) / 8 * 8;
    if (c.WT + kWD - 1 > kWDma * kWWaves || c.WT > kWMaxSteps) return false;   // two row segments of WT + 8 fragments in LDS
    c.fused = wg_filter_for(blur_k) != nullptr && DAU_TUNE_INT("DAU_WGRAD_FUSED_STAGE", 1) != 0;
    if (!c.fused && !blur4_pack_fits(blur_k, c.Hp, c.Wp)) return false;
    const int fgroups = (c.FB + kWWaves - 1) / kWWaves;
    const int base = c.SB * fgroups * kWD * kNumK;
    // ranges of image chunks: about four workgroups per CU or more, and among the counts that divide the chunks evenly the one whose
    // grid wastes the least of its last round of 256 workgroups (one per CU: two rows of Xk fill the LDS)
    int splits = (1024 + base - 1) / base;
    splits = splits < 1 ? 1 : (splits > c.NC ? c.NC : splits);
    {
        double best = 1e30;
        int pick = splits;
        for (int sp = splits; sp <= c.NC && sp <= 4 * splits; ++sp) {
            if (c.NC % sp) continue;
            const double wgs = (double)base * sp, waste = std::ceil(wgs / 256.0) * 256.0 / wgs;
            if (waste < best - 0.02) { best = waste; pick = sp; }
        }
        if (best < 1e30) splits = pick;
    }
    c.splits = splits;
    *cfg = c;
    return true;
}

size_t dense_wgrad_workspace_bytes(const WgradConfig& c) { return wg_layout(c).total; }

void dense_wgrad_init(const WgradConfig& c) {
    if (!c.fused) blur4_pack_init(c.blur_k);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wg_stage_x_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 65 * 4);
    (void)hipFuncSetAttribute(wg_gemm_for(c.WT), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void dense_wgrad_run(hipStream_t st, const WgradConfig& c, const float* x, const float* dy, const float* filters,
                     const UnitRef* table, int drop_col, int drop_row, float* r4, void* workspace, const Guard& guard, int nk) {
    nk = nk < 1 ? 1 : (nk > kNumK ? kNumK : nk);
    const WgLayout l = wg_layout(c);
    char* ws = static_cast<char*>(workspace);
    const Shape& s = c.sh;
    if (c.fused) {
        unsigned short* xt = reinterpret_cast<unsigned short*>(ws + l.xk_off);
        WgTransposeArgs t{};
        t.x = reinterpret_cast<const unsigned short*>(x); t.xt = xt;
        t.N = s.N; t.S = s.S; t.SB = c.SB; t.NC = c.NC; t.H = s.H; t.W = s.W; t.nruns = (s.W + 63) / 64;
        t.vec = s.W % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0;
        t.guard = guard;
        hipLaunchKernelGGL(wg_transpose_x_kernel, dim3(c.SB * c.NC * s.H * t.nruns), dim3(512), 0, st, t);
        WgFilterArgs f{};
        f.xt = xt; f.xkt = reinterpret_cast<unsigned short*>(ws + l.xkt_off); f.taps = filters + kTaps1dOffset;
        f.SB = c.SB; f.NC = c.NC; f.H = s.H; f.W = s.W; f.HsT = c.HsT; f.WsT = c.WsT;
        // bands of rows: enough workgroups for a few rounds of the chip (768 resident), at least 8 rows each
        const int kr = (c.blur_k - 1) / 2;
        f.ncolblk = c.WsT / 2;
        int nbands = DAU_TUNE_INT("DAU_WGRAD_FILTER_BANDS", 0);
        if (nbands <= 0) {
            nbands = (4 * 768 + c.SB * c.NC * f.ncolblk - 1) / (c.SB * c.NC * f.ncolblk);
            while (nbands > 1 && (s.H + nbands - 1) / nbands < 4 * kr + 2) --nbands;     // the halo rows are filtered twice
        }
        nbands = nbands < 1 ? 1 : (nbands > s.H ? s.H : nbands);
        f.RB = (s.H + nbands - 1) / nbands; f.nbands = (s.H + f.RB - 1) / f.RB;
        f.nblocks = c.SB * c.NC * f.nbands * f.ncolblk;
        f.nk = nk;
        f.guard = guard;
        void* args[] = {&f};
        (void)hipLaunchKernel(wg_filter_for(c.blur_k), dim3((f.nblocks + 7) / 8 * 8), dim3(256), args, 0, st);
    } else {
    float* xk = reinterpret_cast<float*>(ws + l.xk_off);
    launch_blur4_pack(st, x, filters, s.N, s.S, c.SB * 32, s.H, s.W, c.Hp, c.Wp, c.blur_k, true, xk, guard);
    {
        WgStageXArgs a{};
        a.xk = xk; a.xkt = reinterpret_cast<__bf16*>(ws + l.xkt_off);
        a.N = s.N; a.NP = (s.N + 1) / 2; a.SB = c.SB; a.NC = c.NC; a.H = s.H; a.W = s.W; a.Hp = c.Hp; a.Wp = c.Wp;
        a.HsT = c.HsT; a.WsT = c.WsT; a.guard = guard;
        hipLaunchKernelGGL(wg_stage_x_kernel, dim3(c.SB * c.NC * c.HsT * (c.WsT / 8)), dim3(512), 256 * 65 * 4, st, a);
    }
    }
    {
        WgStageEArgs a{};
        a.dy = reinterpret_cast<const unsigned short*>(dy); a.et = reinterpret_cast<__bf16*>(ws + l.et_off);
        a.N = s.N; a.F = s.F; a.FB = c.FB; a.NC = c.NC; a.H = s.H; a.W = s.W; a.WT = c.nseg * c.WT; a.drop_col = drop_col; a.drop_row = drop_row;
        a.nruns = (c.nseg * c.WT + 63) / 64;
        a.vec = s.W % 8 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0;
        a.guard = guard;
        hipLaunchKernelGGL(wg_stage_e_kernel, dim3(c.FB * c.NC * s.H * a.nruns), dim3(512), 0, st, a);
    }
    {
        WgGemmArgs a{};
        a.xkt = ws + l.xkt_off; a.et = ws + l.et_off; a.c = reinterpret_cast<float*>(ws + l.c_off);
        a.SB = c.SB; a.FB = c.FB; a.NC = c.NC; a.H = s.H; a.HsT = c.HsT; a.WsT = c.WsT; a.WT = c.WT; a.nseg = c.nseg; a.rowf = c.WT + kWD - 1; a.splits = c.splits;
        a.fgroups = (c.FB + kWWaves - 1) / kWWaves; a.nk = nk; a.guard = guard;
        const int grid = c.SB * a.fgroups * kWD * nk * c.splits;
        void* args[] = {&a};
        (void)hipLaunchKernel(wg_gemm_for(c.WT), dim3(grid), dim3(kWWaves * 64), args, (size_t)2 * (c.WT + kWD - 1) * 1024, st);
    }
    {
        const long units = (long)s.S * s.G * s.F * nk;
        const int grid = (int)((units + 255) / 256 < 8192 ? (units + 255) / 256 : 8192);
        hipLaunchKernelGGL(wg_finish_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float*>(ws + l.c_off), table, s.S,
                           s.G, s.F, c.SB * 32, c.FB * 32, c.splits, nk, r4, guard);
    }
}

}  // namespace DAU_DENSE_NS
}  // namespace dau
