// Densified parameter gradients on the bf16 matrix cores (DAU_FLAG_DENSE_BF16; offsets within +-4, three or more units).
//
//   r_k[s,g,f] = sum_{n,p} E'[n,f,p] * sum_{t in 2x2} b_t(s,g,f) * Xk[n,s, p + o + t]          k = w, mu1, mu2, sigma
//             = sum_t b_t * C_k[o + t][s][f],      C_k[d][s][f] = sum_{n,p} Xk[n,s,p+d] * E'[n,f,p],   d in [-4, 5]^2
//
// The 100 cross-correlations C_k[d] do not depend on the units: 400 GEMMs  M = input channels, N = output channels,
// K = (image, position)  on v_mfma_f32_32x32x16_bf16, 2*400*N*H*W*S*F FLOP whatever the unit count -- at the rate of the
// dense gather-sum (k_dense_bf16.hip) that is the time the exact gather-dot (k_gather_dot.hip) needs for four units, so the
// form is used from three units on (BASELINE config 2 has six).  It replaces the same reference code as the gather-dot:
// DAUConv_bwd_multi_pipeline_kernel and its three preparation kernels
// (include/dau_conv/dau_conv_impl/dau_conv_backward_core.hpp:1017-1820, 1824-2380) and the 4-filter prefilter pass
// (src/dau_conv/util/convolve.cu:48-131).  Numerics: Xk and E' are rounded to bfloat16, products are exact, sums are fp32:
// the 2e-2 bar of the bf16 configuration (opt-in with DAU_FLAG_DENSE_BF16, as the dense gather-sum).
//
// K runs over the IMAGES innermost, so that a displacement only changes the position and every matrix fragment is one
// aligned KiB whatever d is:
//   XkT[k][sb][nc][H+9][WsT][2][32 s][8 n] bf16 the four derivative-filtered copies of x, staged position (r, c) = image
//                                               (r-4, c-4), zero outside the image (wg_stage_x from blur4_pack's fp32 copy)
//   ET [fb][nc][H][WT'][32 f][16 n]      bf16   the error (unit_testing edge rule applied), zero for columns W..WT'-1 (WT' = whole
//                                               row segments of an instantiated length)
//   C  [split][k][10][10][SB*32][FB*32]  fp32   partial correlations of one range of image chunks
// wg_gemm: workgroup = (32 input channels, kind k, row displacement oy, range of image chunks) x 8 waves = 8 blocks of 32
// output channels; a wave keeps the ten column displacements ox as ten 32 x 32 accumulators and walks (image chunk, row,
// column): per column one new Xk fragment (a window of ten slides along the row; the row sits in LDS, loaded by
// global_load_lds one row ahead and shared by the eight waves), one E' fragment (from global memory, six columns ahead)
// and ten MFMAs.
#include <cstdlib>

#include "dau_tiled.hpp"

namespace dau {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

namespace {

constexpr int kWD = 10;            // displacements per axis: -4 .. 5
constexpr int kWMaxSteps = 60;     // widest row (columns per row are straight-line code: see wg_gemm_kernel)
constexpr int kWAhead = 5;         // E' fragments in flight ahead of the one in use
constexpr int kWSlots = kWAhead + 1;
constexpr int kWWaves = 8;

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WgLayout { size_t xk_off, xkt_off, et_off, c_off, total; };
WgLayout wg_layout(const WgradConfig& c) {
    WgLayout l{};
    size_t off = 0;
    l.xk_off = off;  off += round_up((size_t)((c.sh.N + 1) / 2) * c.SB * 32 * c.Hp * c.Wp * 32, 256);
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
    const int y = yy - 4, x0 = xr * 8 - 4;                 // image coordinates of the run
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

// dy[N,F,H,W] bf16 -> ET.  One workgroup per (fb, nc, row, run of 32 columns): 512 (image, channel) rows of 32 values -> LDS
// -> 32 fragments of [32 f][16 n].  The unit_testing edge rule (last column / row of the error dropped) is applied here.
struct WgStageEArgs {
    const unsigned short* dy;
    __bf16* et;
    int N, F, FB, NC, H, W, WT, drop_col, drop_row;
    Guard guard;
};
__global__ void __launch_bounds__(512) wg_stage_e_kernel(const WgStageEArgs a) {
    __shared__ unsigned short lds[512 * 34];               // [n*32 + f][32 columns | 2]
    if (!guard_pass(a.guard)) return;
    int t = blockIdx.x;
    const int runs = (a.WT + 31) / 32;
    const int xr = t % runs; t /= runs;
    const int y = t % a.H; t /= a.H;
    const int nc = t % a.NC;
    const int fb = t / a.NC;
    const bool row_in = !(a.drop_row && y == a.H - 1);
    const int wlim = a.drop_col ? a.W - 1 : a.W;
    for (int i = threadIdx.x; i < 512 * 32; i += blockDim.x) {
        const int row = i >> 5, c = i & 31;
        const int nl = row >> 5, fl = row & 31;
        const int n = nc * 16 + nl, f = fb * 32 + fl, x = xr * 32 + c;
        unsigned short v = 0;
        if (row_in && n < a.N && f < a.F && x < wlim) v = a.dy[(((size_t)n * a.F + f) * a.H + y) * a.W + x];
        lds[row * 34 + c] = v;
    }
    __syncthreads();
    // fragment of column c: [32 f][16 n]; a thread writes one channel row of 16 images
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) {
        const int c = i >> 5, fl = i & 31;
        const int x = xr * 32 + c;
        if (x >= a.WT) continue;
        unsigned short o[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) o[n] = lds[(n * 32 + fl) * 34 + c];
        unsigned short* dst = reinterpret_cast<unsigned short*>(a.et) + ((((size_t)fb * a.NC + nc) * a.H + y) * a.WT + x) * 512 + fl * 16;
#pragma unroll
        for (int n = 0; n < 16; ++n) dst[n] = o[n];
    }
}

// ------------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------------
struct WgGemmArgs {
    const char* xkt;
    const char* et;
    float* c;
    int SB, FB, NC, H, HsT, WsT, WT, nseg, rowf, splits, fgroups;   // WT: columns of a row segment, rowf: its Xk fragments (WT + 9)
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
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two row segments of WT + 9 Xk fragments
    if (!guard_pass(a.guard)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int t = blockIdx.x;
    const int sb = t % a.SB; t /= a.SB;
    const int fg = t % a.fgroups; t /= a.fgroups;
    const int oy = t % kWD; t /= kWD;
    const int k = t % kNumK;
    const int split = t / kNumK;
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
    // a segment needs the WT + 9 staged columns from its first one)
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
            // the window holds staged columns x .. x+9 (slot = column mod 10), eq[x mod 6] = E'(x)
            win[(x + kWD - 1) % kWD] = *reinterpret_cast<const bf16x8*>(arow + (x + kWD - 1) * 1024);
            WG_ELOAD(eq[(x + kWAhead) % kWSlots], lfrag, erow + (x + kWAhead) * 1024);     // (the buffer has slack past its end)
            if (x < kWAhead) wg_vmwait<kWAhead + kWDma>(); else wg_vmwait<kWAhead>();
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

// r4[k][(s*G+g)*F+f] = sum over the image ranges and the four bilinear taps of the unit
__global__ void wg_finish_kernel(const float* __restrict__ c, const UnitRef* __restrict__ table, int S, int G, int F, int SP,
                                 int FP, int splits, float* __restrict__ r4, const Guard guard) {
    if (!guard_pass(guard)) return;
    const long units = (long)S * G * F;
    for (long u = blockIdx.x * (long)blockDim.x + threadIdx.x; u < units; u += (long)gridDim.x * blockDim.x) {
        const int f = (int)(u % F), s = (int)(u / ((long)F * G));
        const UnitRef ur = table[u];
        const float b[4] = {ur.w00, ur.w01, ur.w10, ur.w11};
        int dyi = ur.oy + 4, dxi = ur.ox + 4;
        dyi = dyi < 0 ? 0 : (dyi > kWD - 2 ? kWD - 2 : dyi);        // (a guarded call never clamps: offsets within +-4)
        dxi = dxi < 0 ? 0 : (dxi > kWD - 2 ? kWD - 2 : dxi);
        for (int k = 0; k < kNumK; ++k) {
            double sum = 0.0;
            for (int sp = 0; sp < splits; ++sp) {
                const float* ck = c + (((size_t)sp * kNumK + k) * kWD * kWD) * SP * FP + (size_t)s * FP + f;
#pragma unroll
                for (int tp = 0; tp < 4; ++tp)
                    sum += (double)b[tp] * (double)ck[(size_t)((dyi + (tp >> 1)) * kWD + dxi + (tp & 1)) * SP * FP];
            }
            r4[(size_t)k * units + u] = (float)sum;
        }
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
    if (c.WT + kWD - 1 > kWDma * kWWaves || c.WT > kWMaxSteps) return false;   // two row segments of WT + 9 fragments in LDS
    if (!blur4_pack_fits(blur_k, c.Hp, c.Wp)) return false;
    const int fgroups = (c.FB + kWWaves - 1) / kWWaves;
    const int base = c.SB * fgroups * kWD * kNumK;
    int splits = (1024 + base - 1) / base;
    splits = splits < 1 ? 1 : (splits > c.NC ? c.NC : splits);
    c.splits = splits;
    *cfg = c;
    return true;
}

size_t dense_wgrad_workspace_bytes(const WgradConfig& c) { return wg_layout(c).total; }

void dense_wgrad_init(const WgradConfig& c) {
    blur4_pack_init(c.blur_k);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wg_stage_x_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 65 * 4);
    (void)hipFuncSetAttribute(wg_gemm_for(c.WT), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void dense_wgrad_run(hipStream_t st, const WgradConfig& c, const float* x, const float* dy, const float* filters,
                     const UnitRef* table, int drop_col, int drop_row, float* r4, void* workspace, const Guard& guard) {
    const WgLayout l = wg_layout(c);
    char* ws = static_cast<char*>(workspace);
    const Shape& s = c.sh;
    float* xk = reinterpret_cast<float*>(ws + l.xk_off);
    launch_blur4_pack(st, x, filters, s.N, s.S, c.SB * 32, s.H, s.W, c.Hp, c.Wp, c.blur_k, true, xk, guard);
    {
        WgStageXArgs a{};
        a.xk = xk; a.xkt = reinterpret_cast<__bf16*>(ws + l.xkt_off);
        a.N = s.N; a.NP = (s.N + 1) / 2; a.SB = c.SB; a.NC = c.NC; a.H = s.H; a.W = s.W; a.Hp = c.Hp; a.Wp = c.Wp;
        a.HsT = c.HsT; a.WsT = c.WsT; a.guard = guard;
        hipLaunchKernelGGL(wg_stage_x_kernel, dim3(c.SB * c.NC * c.HsT * (c.WsT / 8)), dim3(512), 256 * 65 * 4, st, a);
    }
    {
        WgStageEArgs a{};
        a.dy = reinterpret_cast<const unsigned short*>(dy); a.et = reinterpret_cast<__bf16*>(ws + l.et_off);
        a.N = s.N; a.F = s.F; a.FB = c.FB; a.NC = c.NC; a.H = s.H; a.W = s.W; a.WT = c.nseg * c.WT; a.drop_col = drop_col; a.drop_row = drop_row;
        a.guard = guard;
        hipLaunchKernelGGL(wg_stage_e_kernel, dim3(c.FB * c.NC * s.H * ((c.nseg * c.WT + 31) / 32)), dim3(512), 0, st, a);
    }
    {
        WgGemmArgs a{};
        a.xkt = ws + l.xkt_off; a.et = ws + l.et_off; a.c = reinterpret_cast<float*>(ws + l.c_off);
        a.SB = c.SB; a.FB = c.FB; a.NC = c.NC; a.H = s.H; a.HsT = c.HsT; a.WsT = c.WsT; a.WT = c.WT; a.nseg = c.nseg; a.rowf = c.WT + kWD - 1; a.splits = c.splits;
        a.fgroups = (c.FB + kWWaves - 1) / kWWaves; a.guard = guard;
        const int grid = c.SB * a.fgroups * kWD * kNumK * c.splits;
        void* args[] = {&a};
        (void)hipLaunchKernel(wg_gemm_for(c.WT), dim3(grid), dim3(kWWaves * 64), args, (size_t)2 * (c.WT + kWD - 1) * 1024, st);
    }
    {
        const long units = (long)s.S * s.G * s.F;
        const int grid = (int)((units + 255) / 256 < 2048 ? (units + 255) / 256 : 2048);
        hipLaunchKernelGGL(wg_finish_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float*>(ws + l.c_off), table, s.S,
                           s.G, s.F, c.SB * 32, c.FB * 32, c.splits, r4, guard);
    }
}

}  // namespace dau
