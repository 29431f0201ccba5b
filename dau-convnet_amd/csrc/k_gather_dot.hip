// Tiled gather-dot: parameter gradients
//
//   r_k[s,g,f] = sum_{n,y,x} E'[n,f,y,x] * sum_{dy,dx} b_{dy,dx}(s,g,f) * Xk[n,s, y+oy+dy, x+ox+dx],   k = w, mu1, mu2, sigma
//
// Replaces the reference's DAUConv_bwd_multi_pipeline_kernel + interleave_input_data_kernel +
// interleave_error_data_kernel + perpare_weights_and_offsets_bw_multi
// (include/dau_conv/dau_conv_impl/dau_conv_backward_core.hpp:1017-1820, 2248-2380, 2118-2246, 1824-2115)
// and the 4-filter prefilter pass (src/dau_conv/util/convolve.cu:48-131).  Different algorithm:
//
//  * Re-indexed over the position q of the prefiltered input:
//        r_k[u] = sum_{n,q} Xk[n,s,q] * Et_u[n,f,q],   Et_u[q] = sum_{dy,dx} b_{dy,dx} E'[q - o_u - (dy,dx)]
//    so the error is interpolated once (4 MACs) and multiplied by the four k-planes (4 MACs): 8 MACs
//    per (n, q, unit), the minimum (dau_conv_backward_core.hpp:846-975 does the same split).
//  * LANE = UNIT.  A wave owns AS input channels s and GP pairs of units g; its 64 lanes are two units x 32 output
//    channels.  Every lane keeps its own four accumulators per (s, pair, image) for the whole kernel, so there is
//    no cross-lane reduction (the reference needs cub::WarpReduce + atomics, :1747-1811).
//  * The error tile sits in LDS position-major with the 32 output channels fastest ([row][col][f][image]), so lane f
//    always hits bank pair f: conflict-free ds_read_b64 at ANY per-lane displacement.  Two images are interleaved
//    element-wise, so every packed operand is a natural register pair (image n, image n+1).
//  * Per (position, unit pair, image pair): 4 v_pk_*_f32 interpolate Et (bilinear factors broadcast with op_sel) and
//    2 v_mfma_f32_4x4x1 contract it with the four kinds: the position q is wave-uniform; one 64-lane load fetches
//    Xk of 16 positions (lane 4b+i = kind i of position b) and the MFMA of position b broadcasts block b's A operand
//    to all blocks (CBSZ/ABID), D register k of a lane = its gradient kind k.
//  * A workgroup owns (32 output channels, a block of input channels, a block of four units) and walks (image pair,
//    8x8 region) items with the LDS error tile filled by global_load_lds (two tiles for bucket 4, one for bucket 8).
//    Buckets beyond 8 run as offset-window passes over a WORK LIST of half-sweeps (dot_worklist_kernel below).
//    Work is split in chunks over the items; a small deterministic pass sums the partials (no float atomics).
//  * Accuracy: a lane's fp32 accumulators take at most ~1024 products; then they are added to the workgroup's own slot of the
//    partial sums (kFlushTerms below) and start again from zero.  The slot is a float where it takes at most 16 such additions
//    per chunk (round 4: half the flush traffic; the chunks are summed in double) and a double otherwise (512 x 512 maps).  The rounding error of a parameter gradient --
//    a sum of N*H*W signed products -- therefore does not grow with the batch, the map size or the chunking: 6-7e-7 of the
//    max-norm at every size measured (before: 1.4e-6 at 4 x 512 x 512; the floor of SURVEY.md 8d is 1e-6).  Cost at the
//    north-star shape, same box: 17.17 ms without, 17.34 ms with a flush every 1024 products, 17.62 ms every 512 (4.5e-7).
//  Tuning knobs (timing experiments only): -DDAU_DOT_WAVES=8, DAU_DOT_NBUF=1, DAU_DOT_AS1, DAU_DOT_DEBUG.
#include <cstdlib>
#include <algorithm>
#ifndef DAU_DOT_FLUSH_TERMS
#define DAU_DOT_FLUSH_TERMS 1024      // products per fp32 accumulator chain (see "Accuracy" above)
#endif
#include <type_traits>
#include <utility>

#include "dau_tiled.hpp"

namespace dau {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

namespace {

constexpr int kDF = 32;        // output channels per workgroup (half a wave; the two halves are two units g)
constexpr int kRW = 8;           // region of positions q handled per item: kRW columns x RH rows (RH = 8, or 7 for heights
                                 // such as 7, 14, 28 that waste fewer rows that way); bucket 4 also has 14 x 4 regions
                                 // (DotGeometry::RW) for widths such as 27 and 28 that pad to 32 in steps of 8
#ifndef DAU_DOT_WAVES
#define DAU_DOT_WAVES 16
#endif
constexpr int kDWaves = DAU_DOT_WAVES;
#ifndef DAU_DOT_STRIDE
#define DAU_DOT_STRIDE 1
#endif
#ifndef DAU_DOT_PRIO
#define DAU_DOT_PRIO 0          // timing experiment: 1 = s_setprio 2 around the MFMA burst, 2 = 1 / 3 / 0 for interpolation / MFMA / reads
#endif
constexpr int kParamDwords = 8;  // per lane per (s, g-pair): b00,b01,b10,b11, base, pad x3

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DotGeometry {
    int RW;                     // columns per region: 8, or 14 (with RH = 4; bucket 4, unit blocks of four only)
    int RH;                     // rows per region: 8 or 7 (4 with RW = 14 and in the wide-window form of bucket 18)
    int epitch, erows;          // LDS error tile (positions)
    int rx, ry;                 // regions per image
    int EX, EY;                 // staged error plane (positions): regions*8 + 2R + 1
    int Hp, Wp;                 // staged Xk plane (positions): regions*8
    // The units of a channel are covered by up to two kernel passes: blocks of four units (two unit pairs per wave, two
    // input channels per wave) and, for a remainder of one or two units, one unit pair per wave with four input channels
    // per wave -- every wave carries four (channel, pair) slots either way.  A remainder of three takes a four-unit block.
    // Several offset windows (R > 8, "binned"): one pass over the WORK LIST of every (window, fb, channel group)
    // (dot_worklist_kernel): AS = 4 entries per wave and round, ngb = rounds allocated, nsb = channel groups.
    struct Pass { int g_begin, GP, AS, ngb, sblock, nsb, chunks; size_t params_off, params_bytes; };
    int rounds_launched;        // binned: workgroups per work list in the grid (a workgroup strides over the list's rounds)
    int sgroup;                 // binned: input channels per channel group (their Xk planes span less than 2 GiB, so that a
                                // lane's channel offset fits the 32-bit VGPR offset of a global load); 0: shape not supported
    int npass;
    Pass pass[2];
    int nbuf;                   // error tiles resident in LDS: 2 (load under compute) or 1 (R = 8: one tile fills the LDS)
    int Rp;                     // the bucket rounded up to whole windows (R = 20 is staged and binned like R = 24)
    int Rt;                     // offset radius one LDS tile covers: min(R, 8)
    int nsub1;                  // R > 8: the offset range is cut into nsub1 x nsub1 windows of radius Rt; one workgroup
                                // pass per window, units outside the window contribute zero (their factors are zeroed)
    int s_pad;                  // input channels padded to whole workgroups of every pass
    int nfb, chunks, items;
    size_t tile_bytes;          // padded to 1 KiB
};

// as1 / one_tile: tuning choices fixed at plan creation (TiledDotConfig), so that every later call sees the same layout
constexpr int kWlEntries = 4;                              // work list: entries per wave and round = AS of the binned gather-dot
constexpr int kWlSlots = kDWaves * 2 * kWlEntries;         // half-sweeps per round

DotGeometry make_dot_geometry(const Shape& sh, int R, bool as1 = false, bool one_tile = false, int rounds = 0, bool rw8 = false) {
    DotGeometry g{};
    // Bucket 18 (offsets within +-18: BASELINE config 4 has +-17): 2 x 2 windows of radius 9 over regions of 4 x 8
    // positions (a 23 x 27 tile fills the LDS like the 25 x 25 one) instead of 3 x 3 windows of radius 8 over 8 x 8
    // regions -- fewer, fuller windows (about 2.25 units per (s, f, window) instead of 1)
    const bool wide = R == 18;
    g.Rt = wide ? 9 : (R < 8 ? R : 8);
    g.nsub1 = (R + g.Rt - 1) / g.Rt;
    g.Rp = g.nsub1 * g.Rt;
    R = g.Rp;
    const bool binned = g.nsub1 > 1;
    // rows per region: 7 when that pads the height less (7, 14, 21, 27, 28, ...)
    g.RW = kRW;
    // (bucket 18: regions of 4 rows x 8 columns, the most a radius-9 tile leaves room for)
    // (window passes: 8 rows -- their Xk ring holds row r of a sweep in slot r % 4, which needs whole multiples of four rows)
    g.RH = wide ? 4 : binned ? 8 : (((sh.H + 6) / 7) * 7 < ((sh.H + 7) / 8) * 8 ? 7 : 8);
    {
        // 14 x 4 regions (same 56 positions as 8 x 7; a 13 x 23 tile, two of them fit the LDS) where they pad the map less:
        // 27 and 28 pixel maps become 28 x 28 instead of 28 x 32 positions.  Bucket 4 only (a bucket 8 tile would not fit),
        // and only when every pass has two unit pairs per wave (the interpolation block is written for four chains =
        // two positions x two pairs; one pair per wave takes four positions, which does not divide 14)
        const long area8 = (long)((sh.W + kRW - 1) / kRW * kRW) * ((sh.H + g.RH - 1) / g.RH * g.RH);
        const long area14 = (long)((sh.W + 13) / 14 * 14) * ((sh.H + 3) / 4 * 4);
        const int rem = sh.G % 4;
        const bool pairs_only = sh.G >= 3 && (rem == 0 || rem == 3);
        if (g.Rt == 4 && !binned && !as1 && !one_tile && pairs_only && !rw8 && area14 * 20 <= area8 * 19) { g.RW = 14; g.RH = 4; }   // at least 5 % fewer positions
    }
    g.epitch = g.RW + 2 * g.Rt + 1;
    g.erows = g.RH + 2 * g.Rt + 1;
    g.rx = (sh.W + g.RW - 1) / g.RW;
    g.ry = (sh.H + g.RH - 1) / g.RH;
    g.EX = g.rx * g.RW + 2 * R + 1;
    g.EY = g.ry * g.RH + 2 * R + 1;
    g.Hp = g.ry * g.RH;
    g.Wp = g.rx * g.RW;
    g.nfb = (sh.F + kDF - 1) / kDF;
    {
        const int full4 = sh.G / 4, rem = sh.G % 4;
        const int blocks4 = full4 + (rem == 3 ? 1 : 0);
        g.npass = 0;
        if (binned) {
            const size_t plane_bytes = (size_t)g.Hp * g.Wp * 32;
            const long fit = (long)(((size_t)1 << 31) / plane_bytes) / 32 * 32;          // whole tiles of 32 channels
            g.sgroup = fit >= 32 ? (int)(fit < (sh.S + 31) / 32 * 32 ? fit : (sh.S + 31) / 32 * 32) : 0;
            const int nsg = g.sgroup ? (sh.S + g.sgroup - 1) / g.sgroup : 1;
            // every input channel of a group needs at most G half-sweeps
            const int rounds_max = ((g.sgroup ? g.sgroup : 32) * sh.G + kWlSlots - 1) / kWlSlots;
            g.pass[g.npass++] = DotGeometry::Pass{0, 1, kWlEntries, rounds_max, 0, nsg, 0, 0, 0};
        } else {
            if (blocks4 > 0) g.pass[g.npass++] = DotGeometry::Pass{0, 2, as1 ? 1 : 2, blocks4, 0, 0, 0, 0, 0};
            if (rem == 1 || rem == 2) g.pass[g.npass++] = DotGeometry::Pass{4 * full4, 1, 4, 1, 0, 0, 0, 0, 0};
        }
        g.s_pad = binned ? sh.S : 0;
        for (int i = 0; i < g.npass && !binned; ++i) {
            DotGeometry::Pass& ps = g.pass[i];
            ps.sblock = kDWaves * ps.AS;
            ps.nsb = (sh.S + ps.sblock - 1) / ps.sblock;
            if (ps.nsb * ps.sblock > g.s_pad) g.s_pad = ps.nsb * ps.sblock;
        }
        size_t off = 0;
        for (int i = 0; i < g.npass; ++i) {
            DotGeometry::Pass& ps = g.pass[i];
            ps.params_off = off;
            ps.params_bytes = binned ? round_up((size_t)g.nsub1 * g.nsub1 * g.nfb * ps.nsb * ps.ngb * kWlSlots * 32 * kParamDwords * 4, 256)
                                     : round_up((size_t)g.nsub1 * g.nsub1 * g.s_pad * ps.ngb * ps.GP * g.nfb * 64 * kParamDwords * 4, 256);
            off += ps.params_bytes;
        }
    }
    g.items = ((sh.N + 1) / 2) * g.rx * g.ry;
    // per pass: enough workgroups to fill the chip several times over, but no more chunks than items
    g.chunks = 1;
    for (int i = 0; i < g.npass; ++i) {
        // at most four full rounds of 256 workgroups (one more workgroup would add a whole, nearly empty round), no
        // more chunks than items, and no chunk without items
        // binned: how many rounds a work list uses depends on the offsets and is known on the device only.  The grid holds ONE
        // workgroup per (chunk, work list), which walks the list's rounds one after the other: no workgroup without work
        // unless a window is empty.  That matters more than it seems: workgroups that leave at once MIXED with ones that
        // run for milliseconds cost C4 14 % (same box: 794 ms with 7 workgroups per list where every list has 6 rounds,
        // 681 ms with 6, 685 ms with 3 -- each taking two rounds --, 803 ms with 20) although a grid of nothing but such
        // workgroups passes in 35 us: blocks go to the XCDs round robin and in order, so XCDs that draw the idle ones run
        // ahead and then wait for the dispatcher, which is held up by a block bound for an XCD that is still full.
        // Chunks for 32 workgroups per CU, so that the tail stays short.
        int per_chunk = g.nfb * g.pass[i].nsb * g.pass[i].ngb;
        if (binned) {
            const int want = DAU_TUNE_INT("DAU_DOT_RLAUNCH", 1);
            g.rounds_launched = want < g.pass[i].ngb ? (want > 1 ? want : 1) : g.pass[i].ngb;
            per_chunk = g.nfb * g.pass[i].nsb * g.nsub1 * g.nsub1 * g.rounds_launched;
        }
        // rounds: workgroups per CU the grid is sized for.  More chunks = a finer tail, at 32 B of partial sums (double) per
        // unit and chunk; at the north-star shape 4, 8 and 16 rounds take the same time (same-box A/B,
        // profiles/r2_ab_dot_rounds.txt).  (Round 2 used 16 rounds to shorten the fp32 chains; the in-kernel flush to the
        // double partial sums has taken that role.)
        const int nrounds = rounds > 0 ? rounds : (binned ? 32 : 8);
        int chunks = (256 * nrounds + (binned ? per_chunk - 1 : 0)) / per_chunk;
        if (!binned && chunks > 32) chunks = 32;            // the reduction pass reads every chunk's slab: keep it cheap on small layers
        if (chunks > g.items) chunks = g.items;
        if (chunks < 1) chunks = 1;
        const int per = (g.items + chunks - 1) / chunks;
        chunks = (g.items + per - 1) / per;
        g.pass[i].chunks = chunks;
        if (chunks > g.chunks) g.chunks = chunks;           // slabs allocated for the partial sums
    }
    g.tile_bytes = round_up((size_t)g.erows * g.epitch * kDF * 8, 1024);
    g.nbuf = 2 * g.tile_bytes <= 160 * 1024 ? 2 : 1;
    if (one_tile) g.nbuf = 1;
    return g;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// staging kernels
// ------------------------------------------------------------------------------------------------
// dy[N,F,H,W] -> EP[NP][nfb][EY][EX][32][2]; padded coordinate Y = y + R + 1, zero elsewhere; the
// unit_testing edge rule (last column / row of the error dropped) is applied here.
// One workgroup per (pair, channel block, padded row, chunk of XC padded columns): 64 coalesced row reads -> LDS ->
// one contiguous write of up to XC*256 B (a transpose from channel-major to position-major).  HBM bound.
constexpr int kPackErrorChunk = 512;   // padded columns per workgroup (LDS: 64 rows of up to 513 floats)

__global__ void __launch_bounds__(256) pack_error_kernel(const float* __restrict__ dy, int N, int F, int H, int W, int R,
                                                         int EX, int EY, int nfb, int drop_col, int drop_row, int bf16,
                                                         float* __restrict__ ep, const Guard guard) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [64 = 32 f x 2 images][chunk width | 1]
    if (!guard_pass(guard)) return;
    const int nxc = (EX + kPackErrorChunk - 1) / kPackErrorChunk;
    int t0 = blockIdx.x;
    const int xc = t0 % nxc; t0 /= nxc;
    const int Y = t0 % EY; t0 /= EY;
    const int fb = t0 % nfb;
    const int np = t0 / nfb;
    const int y = Y - (R + 1);
    const int X0 = xc * kPackErrorChunk, X1 = X0 + kPackErrorChunk < EX ? X0 + kPackErrorChunk : EX;   // padded columns
    // image columns of this chunk: x = X - (R + 1), clipped to the image
    const int xa0 = X0 - (R + 1) > 0 ? X0 - (R + 1) : 0, xa1 = X1 - (R + 1) < W ? X1 - (R + 1) : W;
    const int cw = xa1 > xa0 ? xa1 - xa0 : 0;
    const int wp = (cw > 0 ? cw : 1) | 1;                   // odd pitch: conflict-free transposed reads
    const bool rowin = y >= 0 && y < H && !(drop_row && y == H - 1);
    if (rowin && cw > 0) {
        // flat over (r = fl*2 + image, x): narrow maps keep all lanes busy (a 7-pixel row per wave instruction did not)
        // (the loads of kLoadBatch items go out before the first is stored, branch free: see load_phase in dau_common.hpp)
        auto fill = [&](auto bfc) {
            constexpr bool BF = decltype(bfc)::value;
            const int total = 64 * cw;
            for (int t0 = threadIdx.x; t0 < total; t0 += blockDim.x * kLoadBatch) {
                typename RawAct<BF>::type v[kLoadBatch];
                int rr[kLoadBatch];
                bool okk[kLoadBatch];
#pragma unroll
                for (int u = 0; u < kLoadBatch; ++u) {
                    const int t = t0 + u * blockDim.x, tc = t < total ? t : total - 1;
                    const int r = tc / cw, x = tc - r * cw;
                    rr[u] = r;
                    const int f = fb * kDF + (r >> 1), n = 2 * np + (r & 1);
                    const bool ok = f < F && n < N;
                    const long src = (((long)(ok ? n : 0) * F + (ok ? f : 0)) * H + y) * W + xa0;
                    v[u] = load_raw<BF>(dy, src + x);
                    okk[u] = ok;
                }
#pragma unroll
                for (int u = 0; u < kLoadBatch; ++u) {
                    const int t = t0 + u * blockDim.x;
                    if (t < total) lds[rr[u] * wp + (t - rr[u] * cw)] = mask_act(act_of(v[u]), okk[u]);
                }
            }
        };
        if (bf16) fill(std::true_type{}); else fill(std::false_type{});
    }
    __syncthreads();
    float* out = ep + ((((size_t)np * nfb + fb) * EY + Y) * EX + X0) * (kDF * 2);
    const int wlim = drop_col ? W - 1 : W;
    for (int t = threadIdx.x; t < (X1 - X0) * 64; t += blockDim.x) {
        const int X = X0 + (t >> 6), r = t & 63, x = X - (R + 1);
        out[t] = (rowin && x >= xa0 && x < xa1 && x < wlim) ? lds[r * wp + (x - xa0)] : 0.0f;
    }
}

// x[N,S,H,W] -> XK[NP][S][Hp][Wp][4][2]: the four derivative-filtered copies (separable form, see dau_common.hpp),
// zero padded to whole regions.  One workgroup per (pair, output window, channel): raw window (+ blur halo) -> LDS,
// three horizontal passes (gx, ax, cx) -> LDS, five vertical 1-D passes -> 32 B per position.  HBM bound.
struct Blur4Args {
    const float* in;
    const float* taps;
    float* xk;
    int N, C, cstride, H, W, k, Hp, Wp;
    int bf16;                   // input is bfloat16
    int ppb, items;             // windows per workgroup (small maps) and windows in total
    unsigned lds_item_floats;   // LDS floats per window
    int WY, WX, nwy, nwx;       // output window (rows x columns of the Hp x Wp plane) and windows per plane
    Guard guard;
};

// K: compile-time prefilter support (taps live in SGPRs, tap loops unrolled); K = 0: any support
template <int K>
__global__ void __launch_bounds__(512) blur4_pack_kernel(const Blur4Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!guard_pass(a.guard)) return;
    // C counts the channel slots of the staged copy (cstride = input channels padded to whole workgroups of the
    // gather-dot); slots beyond the real channels are written as zero planes
    const int C = a.cstride, H = a.H, W = a.W, k = K ? K : a.k;
    const int Creal = a.C;
    const int lane = threadIdx.x & 63;
    const int nw = (blockDim.x >> 6) / a.ppb;     // waves per window
    const int sub = (threadIdx.x >> 6) / nw, wave = (threadIdx.x >> 6) % nw;
    int t = blockIdx.x * a.ppb + sub;
    const bool active = t < a.items;              // idle wave groups of the last workgroup still reach the barriers
    if (!active) t = a.items - 1;
    const int c = t % C; t /= C;
    const int wx = t % a.nwx; t /= a.nwx;
    const int wy = t % a.nwy;
    const int np = t / a.nwy;
    const int oy0 = wy * a.WY, ox0 = wx * a.WX;
    const int oh = oy0 + a.WY < a.Hp ? a.WY : a.Hp - oy0, ow = ox0 + a.WX < a.Wp ? a.WX : a.Wp - ox0;
    const int kr = (k - 1) / 2;
    const int lw = ow + 2 * kr, lh = oh + 2 * kr;
    f2* A = reinterpret_cast<f2*>(lds + (size_t)sub * a.lds_item_floats);   // raw [lh][lw], image (oy0 - kr + r, ox0 - kr + xl)
    f2* B = A + (size_t)lh * lw;                             // [3][lh][ow]
    const float* tp[6] = {a.taps + kTapGX * kTapPitch, a.taps + kTapAX * kTapPitch, a.taps + kTapCX * kTapPitch,
                          a.taps + kTapGY * kTapPitch, a.taps + kTapAY * kTapPitch, a.taps + kTapBY * kTapPitch};
    float tr[6][K ? K : 1];
    if (K) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < K; ++i) tr[q][i] = tp[q][i];
    }
    auto tap = [&](int q, int i) { return K ? tr[q][i] : tp[q][i]; };
    const int n0 = 2 * np, n1 = 2 * np + 1;
    const bool real = c < Creal;
    const int cs = real ? c : 0;
    const long p0 = ((long)n0 * Creal + cs) * H * W, p1 = ((long)(n1 < a.N ? n1 : n0) * Creal + cs) * H * W;   // element offsets
    const bool bf16 = a.bf16 != 0;
    const float m0 = real ? 1.0f : 0.0f;
    const float m1 = (real && n1 < a.N) ? 1.0f : 0.0f;
    // rows x cols of work for this window's waves: a wave per row when the rows are wide, a flat index when they are narrow
    auto for_each = [&](int rows_, int cols_, auto&& body) {
        if (cols_ >= 56) {
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
                const int yy = oy0 - kr + r, xx = ox0 - kr + xl;
                const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
                const long off = in ? (long)yy * W + xx : 0;            // outside the image: element 0 (valid), discarded
                return Raw2{load_raw<BF>(a.in, p0 + off), load_raw<BF>(a.in, p1 + off)};
            },
            [&](int r, int xl, Raw2 v) {
                const int yy = oy0 - kr + r, xx = ox0 - kr + xl;
                const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
                A[r * lw + xl] = f2{mask_act(m0 * act_of(v.v0), in), mask_act(m1 * act_of(v.v1), in)};
            });
    };
    if (bf16) fill(std::true_type{}); else fill(std::false_type{});
    __syncthreads();
    for_each(lh, ow, [&](int r, int x) {
        const int yy = oy0 - kr + r;
        f2 h1 = {0.0f, 0.0f}, h2 = {0.0f, 0.0f}, h3 = {0.0f, 0.0f};
        if (yy >= 0 && yy < H) {
#pragma unroll
            for (int i = 0; i < k; ++i) {
                const f2 v = A[r * lw + x + i];
                h1 = __builtin_elementwise_fma(v, f2{tap(0, i), tap(0, i)}, h1);
                h2 = __builtin_elementwise_fma(v, f2{tap(1, i), tap(1, i)}, h2);
                h3 = __builtin_elementwise_fma(v, f2{tap(2, i), tap(2, i)}, h3);
            }
        }
        B[(0 * lh + r) * ow + x] = h1; B[(1 * lh + r) * ow + x] = h2; B[(2 * lh + r) * ow + x] = h3;
    });
    __syncthreads();
    f8* out = reinterpret_cast<f8*>(a.xk) + ((size_t)np * a.cstride + c) * a.Hp * a.Wp;
    for_each(active ? oh : 0, ow, [&](int yr, int xc) {
        const int yy = oy0 + yr, xx = ox0 + xc;
        f2 dw = {0.0f, 0.0f}, d1 = {0.0f, 0.0f}, d2 = {0.0f, 0.0f}, ds = {0.0f, 0.0f};
        if (yy < H && xx < W) {
#pragma unroll
            for (int j = 0; j < k; ++j) {
                const f2 b1 = B[(0 * lh + yr + j) * ow + xc], b2 = B[(1 * lh + yr + j) * ow + xc], b3 = B[(2 * lh + yr + j) * ow + xc];
                dw = __builtin_elementwise_fma(b1, f2{tap(3, j), tap(3, j)}, dw);
                d1 = __builtin_elementwise_fma(b2, f2{tap(3, j), tap(3, j)}, d1);
                d2 = __builtin_elementwise_fma(b1, f2{tap(4, j), tap(4, j)}, d2);
                ds = __builtin_elementwise_fma(b3, f2{tap(3, j), tap(3, j)}, ds);
                ds = __builtin_elementwise_fma(b1, f2{tap(5, j), tap(5, j)}, ds);
            }
        }
        out[(size_t)yy * a.Wp + xx] = f8{dw.x, dw.y, d1.x, d1.y, d2.x, d2.y, ds.x, ds.y};
    });
}

// per-lane parameters: params[sub][s][gb][gp][fb][lane][8] = {b00, b01, b10, b11, base, 0, 0, 0}
// lane = half*32 + fl ; unit = (s, g = g_begin + gb*2*GP + 2*gp + half, f = fb*32 + fl); invalid units get zero factors.
__global__ void dot_params_kernel(const UnitRef* __restrict__ table, int S, int G, int F, int R, int Rt, int nsub1,
                                  int epitch, int g_begin, int GP, int ngb, int nfb, int s_pad, float* __restrict__ params,
                                  const Guard guard) {
    if (!guard_pass(guard)) return;
    const long total = (long)nsub1 * nsub1 * s_pad * ngb * GP * nfb * 64;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx % 64);
        long t = idx / 64;
        const int fb = (int)(t % nfb); t /= nfb;
        const int gp = (int)(t % GP); t /= GP;
        const int gb = (int)(t % ngb); t /= ngb;
        const int s = (int)(t % s_pad);
        const int sub = (int)(t / s_pad);
        const int g = g_begin + gb * 2 * GP + 2 * gp + (lane >> 5), fl = lane & 31, f = fb * kDF + fl;
        UnitRef u{0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
        if (s < S && g < G && f < F) u = table[((long)s * G + g) * F + f];
        // offset window of this pass: centre c = -R + Rt + 2*Rt*i per axis; a unit belongs to exactly one window
        int wy = (u.oy + R) / (2 * Rt), wx = (u.ox + R) / (2 * Rt);
        wy = wy < nsub1 ? wy : nsub1 - 1; wx = wx < nsub1 ? wx : nsub1 - 1;
        const int cy = -R + Rt + 2 * Rt * (sub / nsub1), cx = -R + Rt + 2 * Rt * (sub % nsub1);
        if (wy != sub / nsub1 || wx != sub % nsub1) u = UnitRef{cx, cy, 0.0f, 0.0f, 0.0f, 0.0f};
        const int base = (((Rt - (u.oy - cy)) * epitch + (Rt - (u.ox - cx))) * kDF + fl) * 8;
        float* dst = params + idx * kParamDwords;
        dst[0] = u.w00; dst[1] = u.w01; dst[2] = u.w10; dst[3] = u.w11;
        dst[4] = __int_as_float(base); dst[5] = 0.0f; dst[6] = 0.0f; dst[7] = 0.0f;
    }
}

// Window passes (several offset windows, "binned"): the WORK LIST.
// A sweep of the gather-dot gives every lane one unit; which unit is the lane's own business: its parameters carry the byte
// address of its error column (output channel included), its bilinear factors and the index of the sum it accumulates.  What
// the lanes of a HALF wave must share is the input channel s: the 4x4x1 MFMA broadcasts its A operand (Xk of a position) per
// group of 8 blocks with CBSZ = 3 (tools/microbench/mfma_abid checks that), so the two half waves of a wave are independent.
// So the units of (window, 32 output channels fb, input channel s) -- n_s of them, c_f per output channel -- are dealt into
// H_s = max(ceil(n_s / 32), ceil(max c_f / 2)) HALF-SWEEPS of up to 32 units.  Two units of ONE output channel in a half-sweep
// read the same LDS bank pair: their ds_read_b64 takes two passes instead of one, for the whole half wave -- and at two passes
// the LDS pipe, not the matrix pipe, is what the sweep waits for (measured: dealing the units round robin, every half-sweep with
// some pair of that kind, made a sweep 40 % slower).  So the assignment is greedy per input channel (one thread, counts only):
// a unit goes where its output channel is not yet present; when it must share, into a half-sweep that already pays for a
// conflict of that degree.  On BASELINE config 4's uniform offsets that leaves two clean half-sweeps and one with two-way
// conflicts per input channel: 5.7 sweeps per wave and item against 13.0 in round 2 (lane = output channel, conflict-free: an
// input channel needed max_f c_f sweeps of 2 x 32 lanes however few were used) and an ideal of 4.5 (tools/slot_histogram.py;
// the reference splits its large-offset kernels by K instead, dau_conv_backward.cpp:194-231).
// The half-sweeps of a (window, fb, channel group) are listed in ascending s and dealt out: half-sweep h -> round h / 128, and
// inside the round wave-half h % 32, entry (h % 128) / 32 -- every wave of every round but the last carries 4 full entries.
//   params[sub][fb][sg][round][wave][entry][lane][8] = {b00, b01, b10, b11, base, u, soff, 0}
//   base = byte address of the unit's error column in the tile (its output channel included); u = flat unit index
//   (s*G + g)*F + f, -1 for an empty lane; soff = byte offset of channel s inside its group's Xk planes.
//   nrounds[(sub*nfb + fb)*nsg + sg] = rounds in use (workgroups of later rounds leave at once).
// Units whose factors are all zero (ignored units) take no slot.  One workgroup per (window, fb, channel group), 32 channels x
// 32 output channels at a time; at most kWlMaxUnits units per channel pair (plans with more fall back to the direct kernels).
constexpr int kWlMaxUnits = 16;

__global__ void __launch_bounds__(1024) dot_worklist_kernel(const UnitRef* __restrict__ table, int S, int G, int F, int R, int Rt,
                                                            int nsub1, int epitch, int nfb, int nsg, int sgroup, int rounds_max,
                                                            unsigned plane_bytes, float* __restrict__ params,
                                                            int* __restrict__ nrounds, const Guard guard) {
    if (!guard_pass(guard)) return;
    __shared__ int K[32], pre[32], hbase;
    __shared__ unsigned char cnt[32][32];                       // [channel][output channel]: units in the window
    __shared__ unsigned short where[32][32][kWlMaxUnits];       // [channel][output channel][k-th unit]: half-sweep << 5 | lane
    __shared__ unsigned char mult[32][kWlMaxUnits][32];         // [channel][half-sweep][output channel]: units placed
    __shared__ unsigned char fill[32][kWlMaxUnits], degree[32][kWlMaxUnits];   // lanes used, worst bank-pair multiplicity
    __shared__ int lane0_base[32][kWlMaxUnits];                                // tile address of the unit in lane 0
    int t = blockIdx.x;
    const int sg = t % nsg; t /= nsg;
    const int fb = t % nfb;
    const int sub = t / nfb;
    const int fl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int f = fb * kDF + fl;
    const int cy = -R + Rt + 2 * Rt * (sub / nsub1), cx = -R + Rt + 2 * Rt * (sub % nsub1);
    float* const pbase = params + (size_t)blockIdx.x * rounds_max * kWlSlots * 32 * kParamDwords;
    auto slot_ptr = [&](int h, int ln) -> float* {         // lane ln of half-sweep h
        const int round = h / kWlSlots, hl = h % kWlSlots;
        const int wh = hl % (2 * kDWaves), entry = hl / (2 * kDWaves);
        return pbase + ((((size_t)round * kDWaves + (wh >> 1)) * kWlEntries + entry) * 64 + (wh & 1) * 32 + ln) * kParamDwords;
    };
    auto in_window = [&](const UnitRef& u) {
        int wy = (u.oy + R) / (2 * Rt), wx = (u.ox + R) / (2 * Rt);
        wy = wy < nsub1 ? wy : nsub1 - 1; wx = wx < nsub1 ? wx : nsub1 - 1;
        return wy == sub / nsub1 && wx == sub % nsub1 && !(u.w00 == 0.0f && u.w01 == 0.0f && u.w10 == 0.0f && u.w11 == 0.0f);
    };
    // an unused lane: zero factors, and the tile address of its half-sweep's lane 0 (identical addresses are broadcast: an address
    // of its own would put a second address on the bank pair of whichever unit has that output channel); in an entirely
    // empty half-sweep lane fl reads bank pair fl
    auto write_empty = [&](float* dst, unsigned soff, int base) {
        dst[0] = 0.0f; dst[1] = 0.0f; dst[2] = 0.0f; dst[3] = 0.0f;
        dst[4] = __int_as_float(base); dst[5] = __int_as_float(-1); dst[6] = __uint_as_float(soff); dst[7] = 0.0f;
    };
    const int own_base = ((Rt * epitch + Rt) * kDF + fl) * 8;
    if (threadIdx.x == 0) hbase = 0;
    const int s_lo = sg * sgroup, s_hi = s_lo + sgroup < S ? s_lo + sgroup : S;
    for (int s0 = s_lo; s0 < s_hi; s0 += 32) {
        const int s = s0 + sl;                              // the 32 threads of a channel are one half wave
        const bool live = s < s_hi && f < F;
        int c = 0;
        if (live)
            for (int g = 0; g < G; ++g) c += in_window(table[((long)s * G + g) * F + f]) ? 1 : 0;
        cnt[sl][fl] = (unsigned char)c;
        int ns = c, cmax = c;                               // units of the channel, most units of one output channel
        for (int m = 16; m >= 1; m >>= 1) { ns += __shfl_xor(ns, m); cmax = max(cmax, __shfl_xor(cmax, m)); }
        const int hs_n = max((ns + 31) / 32, (cmax + 1) / 2);   // half-sweeps of this channel
        if (fl == 0) K[sl] = hs_n;
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = hbase;
            for (int i = 0; i < 32; ++i) { pre[i] = acc; acc += K[i]; }
            hbase = acc;
        }
        if (fl == 0 && hs_n > 0) {
            // greedy placement of the channel's units: cost of a half-sweep = max(its matrix-pipe time, its LDS time), the
            // LDS time growing with the worst bank-pair multiplicity m (100 : 54 m, the ratio measured for this kernel)
            auto cost = [](int m) { return m <= 1 ? 100 : 54 * m; };
            for (int h = 0; h < hs_n; ++h) {
                fill[sl][h] = 0; degree[sl][h] = 1;
                for (int j = 0; j < 32; ++j) mult[sl][h][j] = 0;
            }
            for (int j = 0; j < 32; ++j)
                for (int k = 0; k < cnt[sl][j]; ++k) {
                    int best = 0, best_key = 1 << 30;
                    for (int h = 0; h < hs_n; ++h) {
                        if (fill[sl][h] >= 32) continue;
                        const int m1 = mult[sl][h][j] + 1, d = degree[sl][h];
                        const int key = (cost(m1 > d ? m1 : d) - cost(d)) * 64 + fill[sl][h];   // cheapest, then emptiest
                        if (key < best_key) { best_key = key; best = h; }
                    }
                    where[sl][j][k] = (unsigned short)(best << 5 | fill[sl][best]);
                    ++fill[sl][best];
                    const int m1 = ++mult[sl][best][j];
                    if (m1 > degree[sl][best]) degree[sl][best] = (unsigned char)m1;
                }
        }
        __syncthreads();
        if (s < s_hi && hs_n > 0) {
            const unsigned soff = (unsigned)(s - s_lo) * plane_bytes;
            int k = 0;
            if (live)
                for (int g = 0; g < G; ++g) {
                    const UnitRef u = table[((long)s * G + g) * F + f];
                    if (!in_window(u)) continue;
                    const int w_ = where[sl][fl][k++];
                    float* dst = slot_ptr(pre[sl] + (w_ >> 5), w_ & 31);
                    const int base = (((Rt - (u.oy - cy)) * epitch + (Rt - (u.ox - cx))) * kDF + fl) * 8;
                    dst[0] = u.w00; dst[1] = u.w01; dst[2] = u.w10; dst[3] = u.w11;
                    dst[4] = __int_as_float(base); dst[5] = __int_as_float((int)(((long)s * G + g) * F + f));
                    dst[6] = __uint_as_float(soff); dst[7] = 0.0f;
                    if ((w_ & 31) == 0) lane0_base[sl][w_ >> 5] = base;
                }
        }
        __syncthreads();
        if (s < s_hi)
            for (int h = 0; h < hs_n; ++h)                   // the lanes a half-sweep does not use
                if (fl >= fill[sl][h]) write_empty(slot_ptr(pre[sl] + h, fl), (unsigned)(s - s_lo) * plane_bytes, lane0_base[sl][h]);
        __syncthreads();
    }
    // the unused half-sweeps of the last round are empty (their lanes gather from channel 0 of the group with zero factors)
    const int total = hbase, rounds = (total + kWlSlots - 1) / kWlSlots;
    for (int h = total + sl; h < rounds * kWlSlots; h += 32) write_empty(slot_ptr(h, fl), 0u, own_base);
    if (threadIdx.x == 0) nrounds[blockIdx.x] = rounds;
}

// r4[k][u] = sum over the slabs of partial[slab][k][u]; units g < g_split were written by a pass with slabs0 slabs,
// the others by a pass with slabs1 (u = (s*G + g)*F + f)
// zero_from: units g >= zero_from have no partial sums (binned passes give ignored units no slot): their sums are zero
// accumulate: add to r4 (second and later batch slabs of a call) instead of overwriting it
// partial_f32: the slabs hold float sums (see DotArgs::partial_f32), else double
__global__ void dot_reduce_kernel(const void* __restrict__ partial, int partial_f32, long n, int G, int F, int g_split, int slabs0,
                                  int slabs1, int zero_from, int accumulate, float* __restrict__ r4, const Guard guard) {
    if (!guard_pass(guard)) return;
    const double* pd = static_cast<const double*>(partial);
    const float* pf = static_cast<const float*>(partial);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)((i / F) % G);
        const int slabs = g < g_split ? slabs0 : slabs1;
        double s = 0.0;
        if (g < zero_from)
            for (int c = 0; c < slabs; ++c) s += partial_f32 ? (double)pf[(long)c * n + i] : pd[(long)c * n + i];
        r4[i] = accumulate ? (float)((double)r4[i] + s) : (float)s;
    }
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct DotArgs {
    const char* ep;
    const float* xk;
    const float* params;
    void* partial;              // [chunk][4][S][G][F]: every (chunk, unit) belongs to exactly one workgroup, which accumulates into it
    int partial_f32;            // the slots are float: a slot takes at most kMaxF32Flushes additions of ~1024-product sums, which
                                // costs nothing measurable in accuracy (the chunks are summed in double) and halves the flush
                                // traffic; chunks with more flushes per slot (512 x 512 maps) keep double slots
    int N, S, F, G, R;
    int g_begin;                // first unit of this pass
    int NP, nfb, nsb, ngb, nbuf, chunks, items;   // window passes (work list): nsb = channel groups, ngb = rounds allocated
    const int* nrounds;         // window passes: rounds in use per (window, fb, channel group)
    int sgroup;                 // window passes: input channels per channel group
    int rmax;                   // window passes: rounds allocated per work list (ngb of them are launched; a workgroup strides)
    int Rt, nsub1;
    int rx, ry, EX, EY, Hp, Wp, epitch, erows, s_pad;
    unsigned tile_bytes;
    int debug;   // timing experiments only (tuning build, DAU_DOT_DEBUG): 1 = Xk always of the chunk's first item, 2 = no error-tile refills
    Guard guard;
};

// Packed fp32 helpers.  Plain vector builtins (not inline asm): hipcc selects v_pk_mul_f32 / v_pk_fma_f32,
// folds the broadcast of a bilinear factor into op_sel (or a loop-invariant register pair) and takes the
// wave-uniform Xk pair straight from SGPRs.  Inline-asm versions cost an s_nop per dependent pair because
// the hazard recognizer must assume the worst about asm results.
__device__ __forceinline__ f2 pk_mul_lo(f2 a, f2 b) { return a * f2{b.x, b.x}; }
__device__ __forceinline__ f2 pk_fma_lo(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, f2{b.x, b.x}, c); }
__device__ __forceinline__ f2 pk_fma_hi(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, f2{b.y, b.y}, c); }
__device__ __forceinline__ f2 pk_fma_s(f2 a, f2 s, f2 c) { return __builtin_elementwise_fma(a, s, c); }

// LDS reads are inline asm: hipcc would pair them into half-rate ds_read2_b64; waits are placed by hand.
#define lds_read(dst, addr, imm) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define lds_read_imm(dst, addr, imm) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
// Xk ring: the destination is a read-write operand so that the register stays put across loop back-edges.
// The base first goes through an s_mov_b64 inside the asm block: when the register allocator has to spill SGPRs (this kernel sits
// at the limit), the base arrives by v_readlane_b32 -- a VALU write of an SGPR -- and a vector memory instruction that reads such
// an SGPR within 5 cycles reads the OLD value (the manual-hazard table of the ISA; hipcc's hazard recognizer does not look into
// asm blocks).  Seen as a memory fault at "null + offset" when a loop around the kernel body raised the SGPR pressure.  A scalar
// instruction reading the SGPR is interlocked, and scalar-write -> vector-memory-read needs no wait states.
#define x_load(dst, voff, sbase, imm)                                                                                      \
    do {                                                                                                                   \
        unsigned long long sb_;                                                                                            \
        asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx2 %0, %2, %1 offset:%4"                                         \
                     : "+v"(dst), "=&s"(sb_) : "v"(voff), "s"(sbase), "n"(imm) : "memory");                                \
    } while (0)
// ring of kXSlots loads: the oldest is complete when at most kXSlots - 1 (+ tile loads in between) are outstanding
template <int N>
__device__ __forceinline__ void x_wait() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void lgkm_wait0() {
    __builtin_amdgcn_sched_barrier(0);   // the previous position's FMAs stay above the wait (they cover the reads in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // and this position's FMAs stay below it
}

// v_mfma_f32_4x4x1 with CBSZ = 4: the A operand of block ABID (lanes 4*ABID .. 4*ABID+3) feeds all 16 blocks
// (checked on the hardware by tools/microbench/mfma_abid.hip).  abid is a constant after unrolling; the switch folds.
// CBSZ = 3: two groups of 8 blocks (the two half waves), each fed by block ABID (< 8) of its own group.
template <int CBSZ, int ABID>
__device__ __forceinline__ f4 mfma_abid(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CBSZ, ABID, 0); }
template <int CBSZ>
__device__ __forceinline__ f4 mfma_bcast(float a, float b, f4 c, int abid) {
    if constexpr (CBSZ == 3) {
        switch (abid) {
            case 0: return mfma_abid<3, 0>(a, b, c);   case 1: return mfma_abid<3, 1>(a, b, c);   case 2: return mfma_abid<3, 2>(a, b, c);
            case 3: return mfma_abid<3, 3>(a, b, c);   case 4: return mfma_abid<3, 4>(a, b, c);   case 5: return mfma_abid<3, 5>(a, b, c);
            case 6: return mfma_abid<3, 6>(a, b, c);   default: return mfma_abid<3, 7>(a, b, c);
        }
    } else {
        switch (abid) {
            case 0: return mfma_abid<4, 0>(a, b, c);   case 1: return mfma_abid<4, 1>(a, b, c);   case 2: return mfma_abid<4, 2>(a, b, c);
            case 3: return mfma_abid<4, 3>(a, b, c);   case 4: return mfma_abid<4, 4>(a, b, c);   case 5: return mfma_abid<4, 5>(a, b, c);
            case 6: return mfma_abid<4, 6>(a, b, c);   case 7: return mfma_abid<4, 7>(a, b, c);   case 8: return mfma_abid<4, 8>(a, b, c);
            case 9: return mfma_abid<4, 9>(a, b, c);   case 10: return mfma_abid<4, 10>(a, b, c); case 11: return mfma_abid<4, 11>(a, b, c);
            case 12: return mfma_abid<4, 12>(a, b, c); case 13: return mfma_abid<4, 13>(a, b, c); case 14: return mfma_abid<4, 14>(a, b, c);
            default: return mfma_abid<4, 15>(a, b, c);
        }
    }
}

// RH rows per region; the Xk ring has one slot per two region rows = 16 positions (the last slot of an odd RH fetches one
// row too many, which is never used; the buffer has a spare row at its end)
// BINNED: window pass over the work list (dot_worklist_kernel): a wave's AS entries are half-sweeps -- (input channel, k-th unit)
// per HALF wave -- a lane's unit index and its half wave's Xk plane offset come with its parameters, empty entries are skipped
// by the wave, and the workgroups of rounds the list does not use leave at once.
// RW: columns per region (8, or 14 with RH = 4); the Xk ring has one slot per 16 positions of the region in row-major order
// RING (window passes; one error tile fills the LDS): the items of a chunk walk DOWN the columns of regions and the tile is a
// ring of rows, so that an item loads only the RH new rows of its tile (4 of 23 in bucket 18) instead of all of them.
template <int GP, int AS, int RH, bool BINNED = false, int RW = 8, bool RING = false>
__global__ void __launch_bounds__(kDWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) gather_dot_kernel(const DotArgs a) {
    constexpr int kRH = RH;
    constexpr int kRW = RW;                       // (shadows the namespace constant)
    // positions one Xk load covers: 16 (lane 4b+i = kind i of position b of the wave's input channel), or -- window passes --
    // 8 per half wave (lane 32h+4b+i = kind i of position b of half h's input channel): one region row
    constexpr int kPPS = BINNED ? 8 : 16;
    // ring depth: window passes keep four region rows of Xk in flight (half a sweep of an 8-row region; a ring of eight would cost
    // sixteen registers the kernel does not have); row r of a sweep lives in slot r % 4
    constexpr int kXSlots = BINNED ? (RH < 4 ? RH : 4) : (RW == 8 ? (RH + 1) / 2 : (RH * RW + 15) / 16);
    constexpr int kCBSZ = BINNED ? 3 : 4;            // A operand broadcast over all 16 blocks, or within each half wave
    static_assert(RW == 8 || (RW % 2 == 0 && !BINNED), "region width");
    static_assert(!BINNED || RH % 4 == 0, "window passes: row r of a sweep lives in ring slot r % 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (!guard_pass(a.guard)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // workgroup -> (chunk, fb, sb), sb fastest; consecutive logical ids share the error tiles and are
    // placed on one XCD (blocks b, b+8, ... share an XCD)
    const int nblk = gridDim.x;
    int logical;
    {
        const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        const int chunk_ = nblk / 8, rem = nblk % 8;
        logical = (xcd < rem ? xcd * (chunk_ + 1) : rem * (chunk_ + 1) + (xcd - rem) * chunk_) + idx;
    }
    const int sb = logical % a.nsb;                  // block of input channels (window passes: channel group)
    int gb = (logical / a.nsb) % a.ngb;              // block of units (window passes: first round of the work list)
    const int fb = (logical / (a.nsb * a.ngb)) % a.nfb;
    const int nsub = a.nsub1 * a.nsub1;
    const int sub = (logical / (a.nsb * a.ngb * a.nfb)) % nsub;
    const int chunk = logical / (a.nsb * a.ngb * a.nfb * nsub);
    // tile origin of this pass's offset window inside the staged error plane (0 when one tile covers the bucket)
    const int sub_dy = 2 * (a.R - a.Rt) - 2 * a.Rt * (sub / a.nsub1), sub_dx = 2 * (a.R - a.Rt) - 2 * a.Rt * (sub % a.nsub1);

    // Window passes: the grid holds a.ngb (= 1 unless a tuning knob says otherwise) workgroups per (chunk, window, fb, channel
    // group), and workgroup gb takes the rounds gb, gb + a.ngb, ... of the work list, however many there are (a.rmax are
    // allocated): see make_dot_geometry for why the grid is not sized for the worst case.
    int nr = 1;
    if constexpr (BINNED) {
        nr = a.nrounds[(sub * a.nfb + fb) * a.nsb + sb];
        if (gb >= nr) return;
    }
    // this chunk's contiguous range of items (image pair, region)
    const int per = (a.items + a.chunks - 1) / a.chunks;
    const int item0 = chunk * per;
    const int item1 = item0 + per < a.items ? item0 + per : a.items;
#if DAU_DOT_STRIDE
    for (;;) {
#endif

    // per-lane parameters of the wave's AS x GP units, resident in registers for the whole kernel
    f2 bw[AS][GP][2];
    unsigned base[AS][GP];      // RING: (the lane's first tile row) << 16 | byte offset inside a row
    unsigned xv[AS];            // BINNED: the lane's Xk offset: its half wave's input channel plane + (lane & 31) * 8
    int s_of[AS];
    bool act[AS];               // BINNED: some lane of the wave has a unit in entry si (wave-uniform)
    const int s_base = BINNED ? sb * a.sgroup : sb * (kDWaves * AS) + wave * AS;
#pragma unroll
    for (int si = 0; si < AS; ++si) {
        const int s = s_base + si;
        s_of[si] = s;
        act[si] = true;
#pragma unroll
        for (int gp = 0; gp < GP; ++gp) {
            const float* p = BINNED ? a.params + (((((((long)sub * a.nfb + fb) * a.nsb + sb) * a.rmax + gb) * kDWaves + wave) * AS + si) * 64 + lane) * kParamDwords
                                    : a.params + ((((((long)sub * a.s_pad + s) * a.ngb + gb) * GP + gp) * a.nfb + fb) * 64 + lane) * kParamDwords;
            bw[si][gp][0] = f2{p[0], p[1]};
            bw[si][gp][1] = f2{p[2], p[3]};
            base[si][gp] = (unsigned)__float_as_int(p[4]);
            if constexpr (RING) {
                // tile row and byte offset inside the row (the row lives in a ring slot that changes from item to item)
                const unsigned rb = (unsigned)a.epitch * kDF * 8;
                const unsigned br = base[si][gp] / rb;
                base[si][gp] = (br << 16) | (base[si][gp] - br * rb);       // a row is at most 27 positions x 256 B
            }
            if constexpr (BINNED) {
                act[si] = __ballot(__float_as_int(p[5]) >= 0) != 0ull;   // (the unit index itself is re-read at flush time)
                // slot layout of the window passes: lane 32 h + 4 b + i = kind i of position (row b % 4, column b / 4) of half h's channel
                xv[si] = __float_as_uint(p[6]) + (unsigned)((lane >> 2) & 3) * (unsigned)(a.Wp * 32) + (unsigned)((lane >> 4) & 1) * 32u + (unsigned)(lane & 3) * 8u;
            }
        }
        if constexpr (BINNED) {
            // (act[si] was set with the parameters above)
        }
    }
    int first_act = 0;          // first input channel of the wave that has units; AS: none
    if constexpr (BINNED) {
        first_act = AS;
#pragma unroll
        for (int si = AS - 1; si >= 0; --si) first_act = act[si] ? si : first_act;
    }
    // BINNED: the per-lane Xk offset of the first active entry.  (Selected by the act[] flags, here and for the next active
    // entry below: a select chain keyed on the entry INDEX is turned into a dynamically indexed array by hipcc, i.e. scratch.)
    unsigned xv_first = 0;
    if constexpr (BINNED) {
#pragma unroll
        for (int si = AS - 1; si >= 0; --si) xv_first = act[si] ? xv[si] : xv_first;
    }
    const bool wave_idle = first_act >= AS;

    f4 acc[AS][GP][2];   // [input channel][unit pair][image]: register k = gradient kind
#pragma unroll
    for (int si = 0; si < AS; ++si)
#pragma unroll
        for (int gp = 0; gp < GP; ++gp) { acc[si][gp][0] = f4{0, 0, 0, 0}; acc[si][gp][1] = f4{0, 0, 0, 0}; }

    const int regions = a.rx * a.ry;
    const unsigned tile_bytes = a.tile_bytes;
    const unsigned row_bytes = (unsigned)a.epitch * kDF * 8;

    const unsigned pieces = tile_bytes >> 10;
    auto tile_src = [&](int item) -> const char* {
        const int np = item / regions, reg = item % regions;
        const int ry = RING ? reg % a.ry : reg / a.rx, rx = RING ? reg / a.ry : reg % a.rx;
        return a.ep + ((((size_t)np * a.nfb + fb) * a.EY + (size_t)(ry * kRH + sub_dy)) * a.EX + (size_t)(rx * kRW + sub_dx)) * (kDF * 8);
    };
    // whole tile at once (first item; single-tile mode)
    auto issue = [&](int item, int buf) {
        const char* src = tile_src(item);
        for (unsigned piece = wave; piece < pieces; piece += kDWaves) {
            const unsigned b = piece * 1024 + lane * 16;
            unsigned trow = b / row_bytes;
            const unsigned within = b - trow * row_bytes;
            if (trow >= (unsigned)a.erows) trow = a.erows - 1;        // tail padding: re-read a valid row
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)trow * a.EX * (kDF * 8) + within),
                                             (lds_ptr_t)(smem + buf * tile_bytes + piece * 1024), 16, 0, 0);
        }
    };
    // RING: the RH new rows of the tile of `item` (the region below the previous item's) into the ring slots the previous
    // tile's first RH rows occupied (origin = ring slot of the previous tile's row 0).  Row by row: a row is not a whole number
    // of KiB pieces, and a piece must be contiguous in LDS.
    auto issue_rows = [&](int item, int origin) {
        const char* src = tile_src(item);
        const unsigned row_pieces = (row_bytes + 1023) >> 10;
        for (unsigned u = wave; u < (unsigned)kRH * row_pieces; u += kDWaves) {
            const unsigned rr = u / row_pieces, pc = u - rr * row_pieces;
            int slot = origin + (int)rr;
            slot = slot >= a.erows ? slot - a.erows : slot;
            const unsigned within = pc * 1024 + lane * 16;
            if (within < row_bytes)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)(a.erows - kRH + rr) * a.EX * (kDF * 8) + within),
                                                 (lds_ptr_t)(smem + (unsigned)slot * row_bytes + pc * 1024), 16, 0, 0);
        }
    };
    // Two-tile mode only exists for the 17 x 17 position tile of bucket R = 4 (host: nbuf == 2 <=> Rt == 4).  There every
    // wave issues exactly kRounds4 load instructions per tile (waves without a last piece repeat their previous one), so
    // that the counted waits of the Xk ring can step over them (see x_wait below).
    constexpr unsigned kPieces4 = ((kRW + 2 * 4 + 1) * (kRH + 2 * 4 + 1) * kDF * 8 + 1023) / 1024;
    constexpr int kRounds4 = (kPieces4 + kDWaves - 1) / kDWaves;
    auto issue_next = [&](int item, int buf) {
        const char* src = tile_src(item);
#pragma unroll 1
        for (int r = 0; r < kRounds4; ++r) {
            unsigned piece = r * kDWaves + wave;
            if (piece >= kPieces4) piece -= kDWaves;
            const unsigned b = piece * 1024 + lane * 16;
            constexpr unsigned kRowBytes4 = (kRW + 2 * 4 + 1) * kDF * 8, kRows4 = kRH + 2 * 4 + 1;   // the R = 4 tile
            unsigned trow = b / kRowBytes4;                                                           // constant divisor
            const unsigned within = b - trow * kRowBytes4;
            if (trow >= kRows4) trow = kRows4 - 1;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (size_t)trow * a.EX * (kDF * 8) + within),
                                             (lds_ptr_t)(smem + buf * tile_bytes + piece * 1024), 16, 0, 0);
        }
    };

    // Xk of (item, s) at the region origin; one position = 8 floats [kind][image].  One 64-lane load fetches 16
    // positions = two region rows: lane 4b+i gets the image pair of kind i of position b, which is the A operand of
    // block b of v_mfma_f32_4x4x1 (lane 4b+i supplies A[i]); the MFMA of position b then broadcasts block b to all
    // blocks (CBSZ/ABID).  512 useful bytes per load instruction -- the first version loaded one position per
    // instruction with every quad fetching the same 32 B, which kept the texture-address unit busy half of the time
    // (and all of the time with one unit pair per wave).  The address is a wave-uniform base (SGPR pair) + one VGPR.
    const size_t xpitch = (size_t)a.Wp * 32;
    const unsigned xlane = (unsigned)((lane >> 5) * xpitch + (lane & 31) * 8);
    // RW != 8: a slot's 16 positions are not two whole rows; lane 4b+i of slot t holds kind i of the region's position
    // 16t+b (row-major), clamped to the region's last position
    unsigned xoff[kXSlots];
#pragma unroll
    for (int t = 0; t < kXSlots; ++t) {
        int P = 16 * t + (lane >> 2);
        P = P < kRH * kRW ? P : kRH * kRW - 1;
        xoff[t] = RW == 8 ? 0u : (unsigned)((P / kRW) * xpitch + (P % kRW) * 32 + (lane & 3) * 8);
    }
    // slot t of the sweep that starts at `base`
    // (window passes: `base` is the sweep origin in the channel group's first plane, voff the lane's channel + position offset)
    auto x_fetch = [&](f2& dst, const char* base, int t, unsigned voff) {
#ifdef DAU_DIAG_FUSED_XK
        // Timing experiment (tools/build_variant.sh ... -DDAU_DIAG_FUSED_XK; DESIGN "the fused prefilter"): a LOWER BOUND of what
        // computing the four derivative-filtered kinds inside this kernel would add.  One Xk load stands for 16 positions x 4 kinds x
        // an image pair of one input channel; from raw x that is, per position, three 7-tap horizontal passes (shared by the seven rows
        // that use them: 21 packed FMAs) and five 7-tap vertical passes (35) = 56 packed FMAs, i.e. 14 wave instructions per load
        // with the work spread perfectly over the 64 lanes -- no loads of raw x, no LDS exchange, no borders, no registers.
        {
            f2 d_ = dst;
#pragma unroll
            for (int q_ = 0; q_ < 14; ++q_) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d_));
            asm volatile("" ::"v"(d_));
        }
#endif
        if constexpr (BINNED) x_load(dst, voff, base + (size_t)(t / 4) * 4 * xpitch + (t % 4) * 64, 0);   // column pair t % 4 of row block t / 4
        else if constexpr (RW == 8) x_load(dst, xlane, base + 2 * t * xpitch, 0);
        else x_load(dst, xoff[t], base, 0);
    };
    auto sweep_ptr = [&](int item, int s) -> const char* {
        if (a.debug & 1) item = item0;      // timing experiment (tuning build): every item reads the Xk of the chunk's first one
        const int np_ = item / regions, reg_ = item % regions;
        const int ry_ = RING ? reg_ % a.ry : reg_ / a.rx, rx_ = RING ? reg_ / a.ry : reg_ % a.rx;
        return reinterpret_cast<const char*>(a.xk) +
               ((((size_t)np_ * a.s_pad + s) * a.Hp + (size_t)ry_ * kRH) * a.Wp + (size_t)rx_ * kRW) * 32;
    };
    // Ring of the next 8 rows' Xk (4 slots of 2 VGPRs), filled by ordinary vector loads and retired with
    // COUNTED vmcnt waits: vector memory returns in order and is independent of the LDS counter.  The ring
    // runs continuously across sweeps and items (a finished row pair is refilled with the same rows of
    // the next sweep), so the memory latency is exposed once per kernel, not once per item.
    // (The first version fetched Xk with scalar loads: they share lgkmcnt with LDS and return out of order, so
    //  every LDS wait had to drain them; measured 6 ms of exposed scalar-miss latency.)
    f2 xr[kXSlots];
#pragma unroll
    for (int i = 0; i < kXSlots; ++i) xr[i] = f2{0.0f, 0.0f};
    if (item0 < item1) {
        issue(item0, 0);
        if (!wave_idle) {
            const char* x0 = sweep_ptr(item0, BINNED ? s_base : s_base + first_act);
            const unsigned v0 = xv_first;
#pragma unroll
            for (int i = 0; i < kXSlots; ++i) x_fetch(xr[i], x0, i, v0);
        }
    }
    // The fp32 accumulators leave for the (double) partial sums every kFlushItems items: ~1024 products per chain.
    // partial[chunk][k][(s*G+g)*F+f] += image 0 + image 1.  Every unit is written by exactly one workgroup per chunk: its own
    // (unit block) pass, or -- BINNED -- the pass of the one (window, half-sweep) it was dealt into; so these are additions by
    // ONE lane in program order (deterministic).  The first flush stores.
    constexpr int kFlushTerms = DAU_DOT_FLUSH_TERMS;
    constexpr int kFlushItems = kFlushTerms / (RH * RW) > 0 ? kFlushTerms / (RH * RW) : 1;
    const long units = (long)a.S * a.G * a.F;
    bool flushed = false;
    auto flush = [&]() {
        const int half = lane >> 5, f = fb * kDF + (lane & 31);
#pragma unroll
        for (int si = 0; si < AS; ++si)
#pragma unroll
            for (int gp = 0; gp < GP; ++gp) {
                const int s = s_of[si], g = a.g_begin + gb * 2 * GP + 2 * gp + half;
                // one entry at a time, its address computed here and now: the kernel has no registers to spare for sixteen
                // loads in flight or for addresses hoisted out of the item loop (the empty asm pins the computation here)
                int u = (s * a.G + g) * a.F + f;
                if constexpr (BINNED)
                    u = __float_as_int(a.params[(((((((long)sub * a.nfb + fb) * a.nsb + sb) * a.rmax + gb) * kDWaves + wave) * AS + si) * 64 + lane) * kParamDwords + 5]);
                asm volatile("" : "+v"(u));
                if (BINNED ? u >= 0 : (s < a.S && g < a.G && f < a.F)) {
                    // the first flush stores; the later ones add with the hardware's atomic (no return value: nothing to wait
                    // for -- a load / add / store round trip per entry cost 0.8 ms of the 17 ms north-star pass).  The slot belongs
                    // to this lane alone, so the order of the additions is the program's.
                    if (a.partial_f32) {
                        float* dst = static_cast<float*>(a.partial) + (long)chunk * kNumK * units + u;
#pragma unroll
                        for (int kk = 0; kk < kNumK; ++kk) {
                            const float v = acc[si][gp][0][kk] + acc[si][gp][1][kk];
                            if (flushed) unsafeAtomicAdd(&dst[kk * units], v);
                            else dst[kk * units] = v;
                        }
                    } else {
                        double* dst = static_cast<double*>(a.partial) + (long)chunk * kNumK * units + u;
#pragma unroll
                        for (int kk = 0; kk < kNumK; ++kk) {
                            const double v = (double)acc[si][gp][0][kk] + (double)acc[si][gp][1][kk];
                            if (flushed) unsafeAtomicAdd(&dst[kk * units], v);
                            else dst[kk * units] = v;
                        }
                    }
                }
                acc[si][gp][0] = f4{0, 0, 0, 0}; acc[si][gp][1] = f4{0, 0, 0, 0};
                __builtin_amdgcn_sched_barrier(0);
            }
        flushed = true;
    };
    int origin = 0;             // RING: ring slot of the current tile's row 0
    int since_flush = 0;
    for (int item = item0; item < item1; ++item) {
        const bool two = a.nbuf == 2;
        const int buf = two ? (item - item0) & 1 : 0;
        // Two tiles: the error tile of this item was requested one whole item ago, so everything but the kXSlots newest vector
        // memory operations (the Xk ring) has to be complete.  One tile (it fills the LDS): it was requested after the
        // previous item's last ring refill, so everything has to be complete; that load is exposed, a few us per ~100 us item.
        if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kXSlots) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // Two tiles: request the next one now (the last item re-requests its own tile into the idle buffer, so that the
        // number of loads in flight is the same for every item).  Vector memory retires in order and these kRounds4 loads
        // sit between the Xk ring entries: during the first sweep the counted waits allow kRounds4 more operations in
        // flight, after that the tile has had a whole sweep of time and the waits cover it.
        if (two) issue_next(item + 1 < item1 && !(a.debug & 2) ? item + 1 : item, buf ^ 1);
        const unsigned bufoff = buf * tile_bytes;
        // positions per group: with one unit pair per wave four positions give the same 16 packed + 8 MFMA run as two
        // positions with two pairs (pipe switches are expensive, tools/microbench/mfma_pk_grouping)
        constexpr int GS = GP == 1 ? 4 : 2;
        f2 eb[2][GP][2][GS];         // [buffer parity][unit pair][row: 0 = tile row j (dy=1), 1 = row j+1 (dy=0)][col]
        f2 epn[GP][2];               // column 0 of a row (precedes its first group)
        unsigned rowaddr[GP], rowaddr2[GP];
        int slot2[GP];               // RING: ring slot of rowaddr2's tile row
        unsigned bcol[GP];           // RING: the lane's byte offset inside a tile row
        // address of the tile row below rowaddr2's
        auto row_below = [&](int gp) -> unsigned {
            if constexpr (RING) return slot2[gp] + 1 == a.erows ? bcol[gp] : rowaddr2[gp] + row_bytes;
            else return rowaddr2[gp] + row_bytes;
        };
        // tile rows 0 and 1 of the sweep of the entry whose (packed) base is b
        auto rows_for = [&](unsigned b, int gp) {
            if constexpr (RING) {
                bcol[gp] = b & 0xffffu;
                int sl = origin + (int)(b >> 16);
                sl = sl >= a.erows ? sl - a.erows : sl;
                rowaddr[gp] = (unsigned)sl * row_bytes + bcol[gp];
                slot2[gp] = sl + 1 == a.erows ? 0 : sl + 1;
                rowaddr2[gp] = (unsigned)slot2[gp] * row_bytes + bcol[gp];
            } else {
                rowaddr[gp] = b + bufoff;
                rowaddr2[gp] = rowaddr[gp] + row_bytes;
            }
        };
        // column `col` of a sweep's first group (and, with col 0, the column that precedes it) into buffer 0
        auto prime_col = [&](int col) {
#pragma unroll
            for (int gp = 0; gp < GP; ++gp) {
                if (col == 0) {
                    lds_read(epn[gp][0], rowaddr[gp], 0);
                    lds_read(epn[gp][1], rowaddr2[gp], 0);
                }
                lds_read(eb[0][gp][0][col], rowaddr[gp], (1 + col) * (kDF * 8));
                lds_read(eb[0][gp][1][col], rowaddr2[gp], (1 + col) * (kDF * 8));
            }
        };
#define DAU_INTERP4(o0, o1, o2, o3, E0, L0, E1, L1, A0, A1, A2, A3, B0, B1, B2, B3)                                  \
    asm volatile("v_pk_mul_f32 %0, %4, %20 op_sel_hi:[1,0]\n\t"                                                     \
                 "v_pk_mul_f32 %1, %5, %21 op_sel_hi:[1,0]\n\t"                                                     \
                 "v_pk_mul_f32 %2, %6, %22 op_sel_hi:[1,0]\n\t"                                                     \
                 "v_pk_mul_f32 %3, %7, %23 op_sel_hi:[1,0]\n\t"                                                     \
                 "v_pk_fma_f32 %0, %8, %20, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                                \
                 "v_pk_fma_f32 %1, %9, %21, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                                \
                 "v_pk_fma_f32 %2, %10, %22, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                               \
                 "v_pk_fma_f32 %3, %11, %23, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                               \
                 "v_pk_fma_f32 %0, %12, %24, %0 op_sel_hi:[1,0,1]\n\t"                                              \
                 "v_pk_fma_f32 %1, %13, %25, %1 op_sel_hi:[1,0,1]\n\t"                                              \
                 "v_pk_fma_f32 %2, %14, %26, %2 op_sel_hi:[1,0,1]\n\t"                                              \
                 "v_pk_fma_f32 %3, %15, %27, %3 op_sel_hi:[1,0,1]\n\t"                                              \
                 "v_pk_fma_f32 %0, %16, %24, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                               \
                 "v_pk_fma_f32 %1, %17, %25, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                               \
                 "v_pk_fma_f32 %2, %18, %26, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                               \
                 "v_pk_fma_f32 %3, %19, %27, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]"                                    \
                 : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)                                                         \
                 : "v"(E0[0]), "v"(E0[1]), "v"(E0[2]), "v"(E0[3]), "v"(L0[0]), "v"(L0[1]), "v"(L0[2]), "v"(L0[3]),   \
                   "v"(E1[0]), "v"(E1[1]), "v"(E1[2]), "v"(E1[3]), "v"(L1[0]), "v"(L1[1]), "v"(L1[2]), "v"(L1[3]),   \
                   "v"(A0), "v"(A1), "v"(A2), "v"(A3), "v"(B0), "v"(B1), "v"(B2), "v"(B3))
#pragma unroll
        for (int si = 0; si < AS; ++si) {
            if (BINNED && !act[si]) continue;     // wave-uniform: no unit of this input channel in the wave's slots
            const char* xbase = sweep_ptr(item, BINNED ? s_base : s_of[si]);
            // where the ring continues after this sweep: next input channel of this item (BINNED: the next one that
            // has units), or the next item (or, at the very end, the last row again so that the number of loads in
            // flight stays constant)
            int nxt = AS;
            unsigned vnext = item + 1 < item1 ? xv_first : (BINNED ? xv[si] : 0u);
#pragma unroll
            for (int sj = AS - 1; sj > si; --sj) {
                nxt = act[sj] ? sj : nxt;
                if constexpr (BINNED) vnext = act[sj] ? xv[sj] : vnext;
            }
            const char* xnext_sweep = BINNED ? (nxt < AS || item + 1 >= item1 ? xbase : sweep_ptr(item + 1, s_base))
                                             : nxt < AS ? sweep_ptr(item, s_base + nxt)
                                                        : (item + 1 < item1 ? sweep_ptr(item + 1, s_base + first_act) : xbase);

            if constexpr (BINNED) {
                // ---- window passes: the sweep in COLUMN groups ---------------------------------------------------------------
                // A group is the four region rows of ONE column: its positions need the error at tile rows r .. r+4 of columns
                // c-1 and c, so a group costs five new reads (one column of five rows) where the row-wise walk of the unit-block
                // kernels costs eight -- 45 instead of 72 reads per 4 x 8 positions.  The window passes need that: their
                // half-sweeps carry units that share a bank pair, and at two passes per read the LDS pipe was their bottleneck.
                // Three column buffers (previous, current, prefetched); the Xk ring holds two columns x four rows per slot.
                constexpr int kBlocks = kRH / 4;
                f2 cb[3][5];
                unsigned ra[5];              // the lane's five tile rows of the row block (byte address of column -1)
                auto rows_of_block = [&](int rb, unsigned (&r)[5]) {
                    if constexpr (RING) {
                        const unsigned bc = base[si][0] & 0xffffu;
                        int sl = origin + (int)(base[si][0] >> 16) + 4 * rb;
                        sl = sl >= a.erows ? sl - a.erows : sl;
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const int slk = sl + k >= a.erows ? sl + k - a.erows : sl + k;
                            r[k] = (unsigned)slk * row_bytes + bc;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 5; ++k) r[k] = base[si][0] + bufoff + (unsigned)(4 * rb + k) * row_bytes;
                    }
                };
#pragma unroll
                for (int rb = 0; rb < kBlocks; ++rb) {
                    // (every row block starts with its own exposed read of columns -1 and 0: carrying the next block's row
                    // addresses through the last group cost registers the 8-row kernel does not have)
                    rows_of_block(rb, ra);
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        lds_read(cb[0][k], ra[k], 0);                    // column -1
                        lds_read(cb[1][k], ra[k], kDF * 8);              // column 0
                    }
                    lgkm_wait0();
#pragma unroll
                    for (int c = 0; c < kRW; ++c) {
                        f2(&prev)[5] = cb[c % 3];
                        f2(&cur)[5] = cb[(c + 1) % 3];
                        f2(&nxt)[5] = cb[(c + 2) % 3];
                        if (c + 1 < kRW) {
#pragma unroll
                            for (int k = 0; k < 5; ++k) lds_read(nxt[k], ra[k], (c + 2) * (kDF * 8));       // column c + 1
                        }
                        f2 et[4];
                        {
                            const f2* pe1 = &cur[0]; const f2* pl1 = &prev[0]; const f2* pe0 = &cur[1]; const f2* pl0 = &prev[1];
                            DAU_INTERP4(et[0], et[1], et[2], et[3], pe0, pl0, pe1, pl1,
                                        bw[si][0][0], bw[si][0][0], bw[si][0][0], bw[si][0][0],
                                        bw[si][0][1], bw[si][0][1], bw[si][0][1], bw[si][0][1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const int ld = rb * 4 + c / 2;                // Xk load of this column pair
                        if (c % 2 == 0) x_wait<kXSlots - 1>();
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const int abid = (c % 2) * 4 + p;
                            acc[si][0][0] = mfma_bcast<kCBSZ>(xr[ld % kXSlots].x, et[p].x, acc[si][0][0], abid);
                            acc[si][0][1] = mfma_bcast<kCBSZ>(xr[ld % kXSlots].y, et[p].y, acc[si][0][1], abid);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (c % 2 == 1) {
                            constexpr int kLoads = kRH;               // loads per sweep
                            if (ld + kXSlots < kLoads) x_fetch(xr[ld % kXSlots], xbase, ld + kXSlots, xv[si]);
                            else x_fetch(xr[ld % kXSlots], xnext_sweep, ld + kXSlots - kLoads, vnext);
                        }
                        lgkm_wait0();
                    }
                }
                continue;
            }
            // The sweep over the region is fully unrolled (no back-edge copies).  Software pipeline over groups of GS positions:
            // at the END of a group one lgkmcnt(0) retires the error columns prefetched for the next group, which flew under
            // this group's work.  (Tried in round 3: the last group of a sweep requesting the first group of the item's next
            // sweep, so that only the first sweep of an item starts with an exposed LDS round trip -- no gain at the north-star
            // shape, 18.05 vs 18.12 ms, and 3 % slower in the window passes, which have no registers for it.)
#pragma unroll
            for (int gp = 0; gp < GP; ++gp) rows_for(base[si][gp], gp);
#pragma unroll
            for (int cc = 0; cc < GS; ++cc) prime_col(cc);
            lgkm_wait0();
#pragma unroll
            for (int j = 0; j < kRH; ++j) {
#pragma unroll
                for (int gq = 0; gq < kRW / GS; ++gq) {
                    constexpr int kGroups = kRW / GS;
                    const int par = (j * kGroups + gq) & 1;   // running group parity (seven groups per row in the 14-column form)
                    // Prefetch the following group's error columns into the other buffer.  That buffer's LAST column is still
                    // needed (as the left neighbour) by this group's first position, so it is requested only after the
                    // interpolation blocks; the others go out now.
#define DAU_PREFETCH_COL(col)                                                                                       \
    if (gq + 1 < kGroups) {                                                                                         \
        _Pragma("unroll") for (int gp = 0; gp < GP; ++gp) {                                                         \
            lds_read(eb[par ^ 1][gp][0][col], rowaddr[gp], (GS * gq + GS + 1 + col) * (kDF * 8));                   \
            lds_read(eb[par ^ 1][gp][1][col], rowaddr2[gp], (GS * gq + GS + 1 + col) * (kDF * 8));                  \
        }                                                                                                           \
    } else if (j + 1 < kRH) {                                                                                       \
        /* first group of the next row: tile rows j+1 and j+2 (rowaddr2 / rowaddr2 + pitch), columns 0..GS */       \
        _Pragma("unroll") for (int gp = 0; gp < GP; ++gp) {                                                         \
            const unsigned r3 = row_below(gp);                                                                      \
            if (col == 0) {                                                                                         \
                lds_read(epn[gp][0], rowaddr2[gp], 0);                                                              \
                lds_read(epn[gp][1], r3, 0);                                                                        \
            }                                                                                                       \
            lds_read(eb[par ^ 1][gp][0][col], rowaddr2[gp], (1 + col) * (kDF * 8));                                 \
            lds_read(eb[par ^ 1][gp][1][col], r3, (1 + col) * (kDF * 8));                                           \
        }                                                                                                           \
    }
#pragma unroll
                    for (int cc = 0; cc + 1 < GS; ++cc) { DAU_PREFETCH_COL(cc) }
                    // The group's GS positions x GP unit pairs: first ALL interpolation blocks (16 packed ops), then ALL
                    // kind-contractions (8 MFMAs).  Alternating the packed-VALU and MFMA pipes in small groups is slow on
                    // gfx950 (tools/microbench/mfma_pk_grouping: 2 MFMA + 4 pk per switch 125 TF, 8 + 16: 143 TF).
                    // Et = b00*E[q-o] + b01*E[q-o-(0,1)] + b10*E[q-o-(1,0)] + b11*E[q-o-(1,1)] for the group's four
                    // (position, unit pair) elements.  The four chains are interleaved instruction by instruction, so that
                    // no packed op depends on the one issued just before it; within a chain every dependency is through the
                    // accumulator operand, which needs no wait state.  (One asm block: the hazard recognizer would put an
                    // s_nop between separate asm statements that feed each other.)
                    f2 et[GS][GP];
                    f2 e0[GS][GP], l0[GS][GP], e1[GS][GP], l1[GS][GP];
#pragma unroll
                    for (int p = 0; p < GS; ++p) {
#pragma unroll
                        for (int gp = 0; gp < GP; ++gp) {
                            e1[p][gp] = eb[par][gp][0][p]; e0[p][gp] = eb[par][gp][1][p];
                            // column to the left: this group's previous column, the row's column 0, or the previous group's last
                            l1[p][gp] = p > 0 ? eb[par][gp][0][p > 0 ? p - 1 : 0] : (gq == 0 ? epn[gp][0] : eb[par ^ 1][gp][0][GS - 1]);
                            l0[p][gp] = p > 0 ? eb[par][gp][1][p > 0 ? p - 1 : 0] : (gq == 0 ? epn[gp][1] : eb[par ^ 1][gp][1][GS - 1]);
                        }
                    }
#if DAU_DOT_PRIO == 2
                    __builtin_amdgcn_s_setprio(1);
#endif
                    {
                        // flatten [GS][GP] (GS*GP == 4 in every instantiation) into chain order
                        static_assert(GS * GP == 4, "four interleaved chains");
                        const f2* pe0 = &e0[0][0]; const f2* pl0 = &l0[0][0]; const f2* pe1 = &e1[0][0]; const f2* pl1 = &l1[0][0];
                        f2* po = &et[0][0];
                        // chain c = p * GP + gp uses the factors of unit pair gp = c % GP
                        DAU_INTERP4(po[0], po[1], po[2], po[3], pe0, pl0, pe1, pl1,
                                    bw[si][0 % GP][0], bw[si][1 % GP][0], bw[si][2 % GP][0], bw[si][3 % GP][0],
                                    bw[si][0 % GP][1], bw[si][1 % GP][1], bw[si][2 % GP][1], bw[si][3 % GP][1]);
                    }
#undef DAU_INTERP4
                    __builtin_amdgcn_sched_barrier(0);
                    // the other buffer's last column (left neighbour of this group's first position) is now consumed
                    { DAU_PREFETCH_COL(GS - 1) }
                    // Xk of this group's positions: when the group opens a new slot (16 positions), that slot is the oldest of
                    // the ring
                    const int pos0 = j * kRW + GS * gq;                   // first position of the group, row-major in the region
                    if (pos0 % kPPS == 0) {
                        if (!BINNED && si == 0 && two) x_wait<kXSlots - 1 + kRounds4>();
                        else x_wait<kXSlots - 1>();
                    }
#if DAU_DOT_PRIO
                    __builtin_amdgcn_s_setprio(DAU_DOT_PRIO == 2 ? 3 : 2);
#endif
#pragma unroll
                    for (int p = 0; p < GS; ++p) {
#pragma unroll
                        for (int gp = 0; gp < GP; ++gp) {
                            const int abid = (pos0 + p) % kPPS, slot = ((pos0 + p) / kPPS) % kXSlots;     // position within the slot's 16 (8)
                            acc[si][gp][0] = mfma_bcast<kCBSZ>(xr[slot].x, et[p][gp].x, acc[si][gp][0], abid);
                            acc[si][gp][1] = mfma_bcast<kCBSZ>(xr[slot].y, et[p][gp].y, acc[si][gp][1], abid);
                        }
                    }
#if DAU_DOT_PRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    // the slot is consumed: refill it with the same positions of the next sweep
                    if ((pos0 + GS) % kPPS == 0 || pos0 + GS == kRH * kRW) {
                        constexpr int kLoads = BINNED ? RH : kXSlots;        // loads per sweep
                        const int ld = pos0 / kPPS + kXSlots;                // the load that takes the freed slot
                        if (ld < kLoads) x_fetch(xr[(pos0 / kPPS) % kXSlots], xbase, ld, BINNED ? xv[si] : 0u);
                        else x_fetch(xr[(pos0 / kPPS) % kXSlots], xnext_sweep, ld - kLoads, vnext);
                    }
                    lgkm_wait0();   // the prefetched group has landed
                }
                // next row: tile rows shift down by one
#pragma unroll
                for (int gp = 0; gp < GP; ++gp) {
                    if constexpr (RING) {
                        const unsigned below = row_below(gp);
                        rowaddr[gp] = rowaddr2[gp]; rowaddr2[gp] = below;
                        slot2[gp] = slot2[gp] + 1 == a.erows ? 0 : slot2[gp] + 1;
                    } else {
                        rowaddr[gp] = rowaddr2[gp]; rowaddr2[gp] += row_bytes;
                    }
                }
            }
        }
        if (!two && item + 1 < item1) {
            __syncthreads();            // every wave is done with the only tile
            if constexpr (RING) {
                // the next item is the region below (same image pair, same column of regions): its tile shares all but RH
                // rows with this one
                const bool below = (item + 1) % regions != 0 && ((item + 1) % regions) % a.ry != 0;
                if (a.debug & 2) { origin = 0; }
                else if (below) { issue_rows(item + 1, origin); origin = origin + kRH >= a.erows ? origin + kRH - a.erows : origin + kRH; }
                else { issue(item + 1, 0); origin = 0; }
            } else {
                if (!(a.debug & 2)) issue(item + 1, 0);
            }
        }
        if (++since_flush == kFlushItems && item + 1 < item1) { flush(); since_flush = 0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flush();
#if DAU_DOT_STRIDE
    if constexpr (!BINNED) break;
    else {
        gb += a.ngb;                // the next round of this workgroup's stride
        if (gb >= nr) break;
        __syncthreads();            // every wave is done with the tile before the next round refills it
    }
    }
#endif
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct DotLayout {
    size_t ep_off, xk_off, params_off, nrounds_off, partial_off, total;
};

DotLayout dot_layout(const TiledDotConfig& c, const DotGeometry& g) {
    DotLayout l{};
    const size_t NP = (c.sh.N + 1) / 2;
    const size_t s_pad = (size_t)g.s_pad;
    size_t off = 0;
    l.ep_off = off; off += round_up(NP * g.nfb * g.EY * g.EX * kDF * 8, 256);
    l.xk_off = off; off += round_up(NP * s_pad * g.Hp * g.Wp * 32 + (size_t)g.Wp * 32, 256);   // + one spare row
    l.params_off = off;
    for (int i = 0; i < g.npass; ++i) off += g.pass[i].params_bytes;
    l.nrounds_off = off; off += round_up((size_t)g.nsub1 * g.nsub1 * g.nfb * g.pass[0].nsb * 4, 256);   // work lists: rounds in use
    l.partial_off = off; off += round_up((size_t)g.chunks * kNumK * c.sh.S * c.sh.G * c.sh.F * 8, 256);   // double
    l.total = off;
    return l;
}

// output window of blur4_pack_kernel: up to 64 columns, and as many rows (a multiple of 8) as keep the raw window plus
// the three filtered copies below ~74 KiB, so that two workgroups share a CU
void blur4_plan(int k, int Hp, int Wp, int* wy, int* wx, size_t* lds) {
    const int WX = Wp < 64 ? Wp : 64;
    auto bytes = [&](int WY) { return ((size_t)(WY + k - 1) * (WX + k - 1) + (size_t)3 * (WY + k - 1) * WX) * 8; };
    int WY = Hp;
    static const size_t limit = (size_t)DAU_TUNE_INT("DAU_BLUR4_LDS_KB", 74) * 1024;
    while (WY > 8 && bytes(WY) > limit) WY -= 8;
    *wy = WY; *wx = WX; *lds = bytes(WY);
}

// a == nullptr: raise the kernel's dynamic-LDS limit (once per plan and device, tiled_dot_init); else launch
template <int GP, int AS, int RH, bool BINNED = false, int RW = 8, bool RING = false>
void launch_dot(hipStream_t st, const DotArgs* a, int grid, size_t lds) {
    auto kern = gather_dot_kernel<GP, AS, RH, BINNED, RW, RING>;
    if (!a) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kDWaves * 64), lds, st, *a);
}

void dispatch_dot(bool binned, bool ring, int RW, int RH, int GP, int AS, hipStream_t st, const DotArgs* a, int grid, size_t lds) {
    if (RW == 14) {                                  // 14 x 4 regions: unit blocks of four only (make_dot_geometry)
        if (AS == 2) launch_dot<2, 2, 4, false, 14>(st, a, grid, lds);
        else launch_dot<2, 1, 4, false, 14>(st, a, grid, lds);
        return;
    }
    if (binned) {
        // window passes: the tile is a ring of rows (DAU_DOT_RING=0 at plan creation: whole tiles, for A/B)
        if (ring) {
            if (RH == 8) launch_dot<1, 4, 8, true, 8, true>(st, a, grid, lds);
            else launch_dot<1, 4, 4, true, 8, true>(st, a, grid, lds);
            return;
        }
        if (RH == 8) launch_dot<1, 4, 8, true>(st, a, grid, lds);
        else launch_dot<1, 4, 4, true>(st, a, grid, lds);
    } else if (RH == 8) {
        if (GP == 1) launch_dot<1, 4, 8>(st, a, grid, lds);
        else if (AS == 2) launch_dot<2, 2, 8>(st, a, grid, lds);
        else launch_dot<2, 1, 8>(st, a, grid, lds);
    } else {
        if (GP == 1) launch_dot<1, 4, 7>(st, a, grid, lds);
        else if (AS == 2) launch_dot<2, 2, 7>(st, a, grid, lds);
        else launch_dot<2, 1, 7>(st, a, grid, lds);
    }
}

auto blur4_pack_for(int blur_k) {
    return blur_k == 7 ? blur4_pack_kernel<7> : blur_k == 5 ? blur4_pack_kernel<5> : blur_k == 9 ? blur4_pack_kernel<9> : blur4_pack_kernel<0>;
}

}  // namespace

// x[N,C,H,W] -> xk[NP][cstride][Hp][Wp][4 kinds][2 images] (fp32), zero beyond the image and in the channel slots C..cstride-1
void launch_blur4_pack(hipStream_t st, const float* x, const float* filters, int N, int C, int cstride, int H, int W, int Hp,
                       int Wp, int blur_k, bool bf16, float* xk, const Guard& guard) {
    int wy, wx; size_t blur_lds;
    blur4_plan(blur_k, Hp, Wp, &wy, &wx, &blur_lds);
    auto kern = blur4_pack_for(blur_k);
    Blur4Args b{};
    b.guard = guard;
    b.in = x; b.taps = filters + kTaps1dOffset; b.xk = xk;
    b.N = N; b.C = C; b.cstride = cstride; b.H = H; b.W = W; b.k = blur_k; b.Hp = Hp; b.Wp = Wp; b.bf16 = bf16 ? 1 : 0;
    b.WY = wy; b.WX = wx; b.nwy = (Hp + wy - 1) / wy; b.nwx = (Wp + wx - 1) / wx;
    // small windows: several per workgroup, so that the 512 threads have rows to share
    const int elems = wy * wx;
    b.ppb = elems >= 2048 ? 1 : elems >= 1024 ? 2 : elems >= 512 ? 4 : 8;
    while (b.ppb > 1 && b.ppb * blur_lds > 64 * 1024) b.ppb /= 2;
    b.items = ((N + 1) / 2) * b.nwy * b.nwx * cstride;
    b.lds_item_floats = (unsigned)(blur_lds / 4);
    hipLaunchKernelGGL(kern, dim3((b.items + b.ppb - 1) / b.ppb), dim3(512), b.ppb * blur_lds, st, b);
}
void blur4_pack_init(int blur_k) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blur4_pack_for(blur_k)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
bool blur4_pack_fits(int blur_k, int Hp, int Wp) {
    int wy, wx; size_t blur_lds;
    blur4_plan(blur_k, Hp, Wp, &wy, &wx, &blur_lds);
    return blur_lds <= 150 * 1024;
}

bool tiled_dot_configure(const Shape& sh, int R, int blur_k, bool bf16, int ignore, TiledDotConfig* cfg) {
    // timing experiments: DAU_DOT_AS1 (one input channel per wave), DAU_DOT_NBUF=1 (one error tile), DAU_DOT_DEBUG
    const bool as1 = DAU_TUNE_SET("DAU_DOT_AS1");
    const bool one_tile = DAU_TUNE_INT("DAU_DOT_NBUF", 2) == 1;
    const int rounds = DAU_TUNE_INT("DAU_DOT_ROUNDS", 0);
    const bool rw8 = DAU_TUNE_INT("DAU_DOT_RW", 0) == 8;      // 8-column regions only (A/B of the 14 x 4 form)
    const DotGeometry g = make_dot_geometry(sh, R, as1, one_tile, rounds, rw8);
    if (g.nbuf * g.tile_bytes > 160 * 1024) return false;
    if (g.nsub1 > 1 && g.sgroup == 0) return false;          // one Xk plane of 2 GiB and more: maps beyond ~1400 x 1400
    if (g.nsub1 > 1 && sh.G > kWlMaxUnits) return false;     // the work list places at most 16 units per channel pair
    // immediates of the unrolled column walk must fit 16 bits
    if ((size_t)g.epitch * kDF * 8 + (g.RW + 1) * kDF * 8 > 65535) return false;
    {
        int wy, wx; size_t blur_lds;
        blur4_plan(blur_k, g.Hp, g.Wp, &wy, &wx, &blur_lds);
        if (blur_lds > 150 * 1024) return false;
    }
    TiledDotConfig c{};
    c.sh = sh; c.R = R; c.blur_k = blur_k; c.NP = (sh.N + 1) / 2; c.variant = g.npass; c.windows = g.nsub1 * g.nsub1;
    c.bf16 = bf16; c.ignore = ignore;
    c.ring = g.nsub1 > 1 && g.nbuf == 1 && DAU_TUNE_INT("DAU_DOT_RING", 1) != 0;
    c.as1 = as1; c.one_tile = one_tile; c.rounds = rounds; c.rw8 = rw8; c.region_cols = g.RW; c.region_rows = g.RH; c.debug = DAU_TUNE_INT("DAU_DOT_DEBUG", 0);
    *cfg = c;
    return true;
}

size_t tiled_dot_workspace_bytes(const TiledDotConfig& c) {
    return dot_layout(c, make_dot_geometry(c.sh, c.R, c.as1, c.one_tile, c.rounds, c.rw8)).total;
}

void tiled_dot_init(const TiledDotConfig& c) {
    const DotGeometry g = make_dot_geometry(c.sh, c.R, c.as1, c.one_tile, c.rounds, c.rw8);
    for (int i = 0; i < g.npass; ++i) dispatch_dot(g.nsub1 > 1, c.ring, g.RW, g.RH, g.pass[i].GP, g.pass[i].AS, nullptr, nullptr, 0, 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pack_error_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blur4_pack_for(c.blur_k)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void tiled_dot_prepare(hipStream_t st, const TiledDotConfig& c, const float* x, const float* dy, const float* filters,
                       const UnitRef* table_bare, int drop_col, int drop_row, void* workspace, const Guard& guard) {
    const DotGeometry g = make_dot_geometry(c.sh, c.R, c.as1, c.one_tile, c.rounds, c.rw8);
    const DotLayout l = dot_layout(c, g);
    char* ws = static_cast<char*>(workspace);
    const Shape& s = c.sh;
    const int s_pad = g.s_pad;
    {
        const int cwmax = s.W < kPackErrorChunk ? s.W : kPackErrorChunk;
        const size_t lds = (size_t)64 * (cwmax | 1) * 4;
        const int nxc = (g.EX + kPackErrorChunk - 1) / kPackErrorChunk;
        hipLaunchKernelGGL(pack_error_kernel, dim3(c.NP * g.nfb * g.EY * nxc), dim3(256), lds, st, dy, s.N, s.F, s.H, s.W, g.Rp, g.EX,
                           g.EY, g.nfb, drop_col, drop_row, c.bf16 ? 1 : 0, reinterpret_cast<float*>(ws + l.ep_off), guard);
    }
    // channel slots beyond S (padding of the last input-channel block) are written as zero planes by the kernel
    launch_blur4_pack(st, x, filters, s.N, s.S, s_pad, s.H, s.W, g.Hp, g.Wp, c.blur_k, c.bf16, reinterpret_cast<float*>(ws + l.xk_off), guard);
    if (g.nsub1 > 1) {
        const DotGeometry::Pass& ps = g.pass[0];
        hipLaunchKernelGGL(dot_worklist_kernel, dim3(g.nsub1 * g.nsub1 * g.nfb * ps.nsb), dim3(1024), 0, st, table_bare, s.S, s.G, s.F,
                           g.Rp, g.Rt, g.nsub1, g.epitch, g.nfb, ps.nsb, g.sgroup, ps.ngb, (unsigned)((size_t)g.Hp * g.Wp * 32),
                           reinterpret_cast<float*>(ws + l.params_off + ps.params_off), reinterpret_cast<int*>(ws + l.nrounds_off), guard);
        return;
    }
    for (int i = 0; i < g.npass; ++i) {
        const DotGeometry::Pass& ps = g.pass[i];
        const long total = (long)g.nsub1 * g.nsub1 * s_pad * ps.ngb * ps.GP * g.nfb * 64;
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(dot_params_kernel, dim3(grid), dim3(256), 0, st, table_bare, s.S, s.G, s.F, g.Rp, g.Rt, g.nsub1,
                           g.epitch, ps.g_begin, ps.GP, ps.ngb, g.nfb, s_pad,
                           reinterpret_cast<float*>(ws + l.params_off + ps.params_off), guard);
    }
}

void tiled_dot_run(hipStream_t st, const TiledDotConfig& c, float* r4, void* workspace, const Guard& guard, bool accumulate) {
    const DotGeometry g = make_dot_geometry(c.sh, c.R, c.as1, c.one_tile, c.rounds, c.rw8);
    const DotLayout l = dot_layout(c, g);
    char* ws = static_cast<char*>(workspace);
    const Shape& s = c.sh;
    DotArgs a{};
    a.ep = ws + l.ep_off;
    a.xk = reinterpret_cast<const float*>(ws + l.xk_off);
    a.partial = ws + l.partial_off;
    a.N = s.N; a.S = s.S; a.F = s.F; a.G = s.G; a.R = g.Rp;
    a.NP = c.NP; a.nfb = g.nfb; a.nbuf = g.nbuf; a.Rt = g.Rt; a.nsub1 = g.nsub1; a.chunks = g.chunks; a.items = g.items;
    a.rx = g.rx; a.ry = g.ry; a.EX = g.EX; a.EY = g.EY; a.Hp = g.Hp; a.Wp = g.Wp; a.epitch = g.epitch; a.erows = g.erows;
    a.s_pad = g.s_pad;
    a.nrounds = reinterpret_cast<const int*>(ws + l.nrounds_off);
    a.sgroup = g.sgroup;
    a.tile_bytes = (unsigned)g.tile_bytes;
    a.debug = c.debug;
    a.guard = guard;
    const bool binned = g.nsub1 > 1;
    const size_t lds = (size_t)g.nbuf * g.tile_bytes;
    {
        // float slots where a slot takes few additions (DotArgs::partial_f32): flushes per (chunk, slot) of the busiest pass
        constexpr int kMaxF32Flushes = 16;
        const int flush_items = std::max(1, DAU_DOT_FLUSH_TERMS / (g.RH * g.RW));
        int flushes = 0;
        for (int i = 0; i < g.npass; ++i) {
            const int per = (g.items + g.pass[i].chunks - 1) / g.pass[i].chunks;
            flushes = std::max(flushes, (per + flush_items - 1) / flush_items);
        }
        a.partial_f32 = (flushes <= kMaxF32Flushes && DAU_TUNE_INT("DAU_DOT_PARTIAL_F32", 1) != 0) ? 1 : 0;
    }
    for (int i = 0; i < g.npass; ++i) {           // every pass writes its own units' slabs of the partial sums
        const DotGeometry::Pass& ps = g.pass[i];
        a.params = reinterpret_cast<const float*>(ws + l.params_off + ps.params_off);
        a.g_begin = ps.g_begin; a.nsb = ps.nsb; a.chunks = ps.chunks;
        a.rmax = ps.ngb; a.ngb = binned ? g.rounds_launched : ps.ngb;
        const int grid = ps.chunks * g.nsub1 * g.nsub1 * g.nfb * a.ngb * ps.nsb;
        dispatch_dot(binned, c.ring, g.RW, g.RH, ps.GP, ps.AS, st, &a, grid, lds);
    }
    const long n = (long)kNumK * s.S * s.G * s.F;
    const int rgrid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    const int g_split = g.npass == 2 ? g.pass[1].g_begin : s.G;
    hipLaunchKernelGGL(dot_reduce_kernel, dim3(rgrid), dim3(256), 0, st, (const void*)a.partial, a.partial_f32, n, s.G, s.F, g_split,
                       g.pass[0].chunks, g.pass[g.npass - 1].chunks, binned ? s.G - c.ignore : s.G, accumulate ? 1 : 0, r4, guard);
}

}  // namespace dau
