// cconv16dc_kernels.hip -- DECODE-order group-causal masked convolution of the latent entropy nets' hidden / last layers
// (A10 for cin = 4, cout <= 4: test/lic360_demo.py:198-206) on v_mfma_f32_16x16x4_f32, input-stationary.
//
// Arithmetic contract (extension/cconv_dc_cuda.cu:313-364): per output scalar 128 virtual lanes, lane
// i = gid*25 + 5 kh + kw runs ONE fmaf chain over the input groups tc = 0 .. L-1 (channel 4 tc + gid at tap (kh, kw)),
// L = min(G, g + 4 + hidden - kh - kw), then the fixed tree p[i]+p[i+64]; +32; ...; +1.  v_mfma_f32_16x16x4_f32 is a k-ordered
// fmaf chain of 4 terms, so K = four consecutive input groups of one lane's chain (cconv16_kernels.hip).
//
// Why not the encode-order mapping: in decode order the groups of a plane sit on DIFFERENT anti-diagonals (group g on
// s = psum - g), so 16 MFMA rows that share a B operand cannot be "4 groups x 4 channels at one tap".  What they can be:
// every lane (g, o, kh, kw) that reads the SAME INPUT DIAGONAL d = s - 4 + kh + kw of the same input channel.  All those chains
// have the same length (the causal rule is "input group tc at diagonal d is known iff tc + d <= psum"): L = psum + hidden - d.
//   * MFMA cols = 16 INPUT rows (th) of one input diagonal, B[k][n] = x[4 (tc0+k) + gid][d][th_n]  -- no halo columns;
//   * MFMA rows = 4 output channels x 4 SLOTS, a slot = (group q of the task's 3 groups, kh) with kw = c + q - kh, c = the
//     input diagonal relative to the task (c = -2..8: 11 diagonals, 75 slots in 24 row tiles);  D row m = 4 o + slot, so the
//     output channel lies in the lane group (lane >> 4) and the SLOT IN THE REGISTER INDEX: every accumulator register holds
//     one chain kind for 4 channels x 16 rows, and everything the reduction tree does is plain register arithmetic;
//   * the accumulator of slot (q, kh) at input row th belongs to output row th + 2 - kh.  A wave owns 32 consecutive rows as
//     two row tiles interleaved by two (row = base + 2 n + t): the realignment by kh - 2 in {-2..2} is a register renaming
//     plus one DPP row shift, whose edge lane comes from the partner wave of the other 32 rows through LDS (4 lanes per
//     register); rows outside the image hold exact zeros;
//   * lane classes as in the other leaf-resident kernels (class = i mod 4 stays together until the last two tree levels),
//     SKEWED by the group: wave CLS owns class (CLS + q) mod 4 of group q, which is what makes the input channel
//     gid = (CLS - c) mod 4 the same for every slot of a diagonal;
//   * a workgroup = 8 waves = 4 classes x 2 row halves of one (sample, 3-group block) -- or two samples side by side when the
//     block's diagonals fit 32 rows; 24 tiles x 2 x 4 = 192 accumulator registers per wave, two waves per SIMD;
//   * NO LDS and NO barrier in the K loop: A (packed weights, one 4..16-byte load per lane for all row tiles of a diagonal)
//     and B (8 bytes per lane) go from L2 straight to registers, 4..6 diagonals ahead of their MFMAs; loop order = K blocks
//     outer, the LIVE diagonals of the block inner (a diagonal's chains run exactly ceil(L/4) blocks), one straight-line body
//     per number of live diagonals -- ~5 other instructions per diagonal step: the first version of this loop spent more
//     time issuing address arithmetic than the matrix pipe spent on the MFMAs.
// Activations: the zero-padded diagonal-major layout of cconv4v3_dc.inc ([n][c][S+12][H+4], cell (s, th) at row s+6, col th+2).
#include "common.h"
#include "conv_plan.h"
#include "cconv_tree.h"

#define XD_GB 3                              // groups per task
#define XD_ND (8 + XD_GB)                    // input diagonals per task: c = -(XD_GB-1) .. 8
#define XD_C0 (XD_GB - 1)                    // dc = c + XD_C0
#define XD_THREADS 512
#define XD_ROW0 6                            // == D3_S0 of cconv4v3_dc.inc (lic360_dc4_layout)
#define XD_COL0 2
#ifndef XD_RA
#define XD_RA 4                              // weight operand slots of the K loop (L2-resident data)
#endif
#ifndef XD_RB
#define XD_RB 4                              // activation operand slots (deeper rings were measured: no gain)
#endif

// ------------------------------------------------------------------------------------------------ slot tables (compile time)
__host__ __device__ constexpr int xd_nslots(int dc) {
    int c = dc - XD_C0, n = 0;
    for (int q = 0; q < XD_GB; ++q)
        for (int kh = 0; kh < 5; ++kh) { const int kw = c + q - kh; if (kw >= 0 && kw <= 4) ++n; }
    return n;
}
__host__ __device__ constexpr int xd_ntiles(int dc) { return (xd_nslots(dc) + 3) / 4; }
__host__ __device__ constexpr int xd_tbase(int dc) { int t = 0; for (int d = 0; d < dc; ++d) t += xd_ntiles(d); return t; }
#define XD_NT 24
static_assert(xd_tbase(XD_ND) == XD_NT, "row tiles per (class, group block)");
// idx-th slot of diagonal dc -> q * 8 + kh, or -1
__host__ __device__ constexpr int xd_slot(int dc, int idx) {
    int c = dc - XD_C0, n = 0;
    for (int q = 0; q < XD_GB; ++q)
        for (int kh = 0; kh < 5; ++kh) {
            const int kw = c + q - kh;
            if (kw >= 0 && kw <= 4) { if (n == idx) return q * 8 + kh; ++n; }
        }
    return -1;
}
__host__ __device__ constexpr int xd_slot_index(int dc, int q, int kh) {
    for (int i = 0; i < xd_nslots(dc); ++i) if (xd_slot(dc, i) == q * 8 + kh) return i;
    return -1;
}
__host__ __device__ constexpr int xd_dc_of_tile(int tile) { int dc = 0; while (xd_tbase(dc + 1) <= tile) ++dc; return dc; }
// halo registers published across the row halves: slots shifted towards lower rows (kh >= 3) need the partner's first column,
// slots shifted towards higher rows (kh <= 1) its last one.  id of (dc, idx, t) in the enumeration of direction UP (1) / DOWN (0)
__host__ __device__ constexpr bool xd_halo_has(int kh, int t, bool up) {
    return up ? (kh == 3 ? t == 0 : kh == 4) : (kh == 1 ? t == 1 : kh == 0);
}
__host__ __device__ constexpr int xd_halo_id(int dc, int idx, int t, bool up) {
    int n = 0;
    for (int d = 0; d < XD_ND; ++d)
        for (int i = 0; i < xd_nslots(d); ++i)
            for (int tt = 0; tt < 2; ++tt) {
                if (d == dc && i == idx && tt == t) return n;
                if (xd_halo_has(xd_slot(d, i) & 7, tt, up)) ++n;
            }
    return n;
}
#define XD_NHALO 45
static_assert(xd_halo_id(XD_ND, 0, 0, true) == XD_NHALO && xd_halo_id(XD_ND, 0, 0, false) == XD_NHALO, "halo registers per direction");

static inline bool conv16dc_ok(const lic360_conv_plan *p) {
    return p->ksz == 5 && p->cin == 4 && p->cout >= 1 && p->cout <= 4 && p->ngroup >= 4 && p->ngroup <= 64 && p->ngroup % 4 == 0;
}
static inline int conv16dc_ngb(const lic360_conv_plan *p) { return (p->ngroup + XD_GB - 1) / XD_GB; }
static inline int conv16dc_nkb(const lic360_conv_plan *p) { return (p->ngroup + 3) / 4; }

// ------------------------------------------------------------------------------------------------ weight packing
// packed[net][gb][class][kb][diagonal dc][lane][t < ntiles(dc)] (the diagonal's block starts at float tbase(dc) * 64): lane
// l = 16 k + i carries A[row i][k] of the MFMA of tile tbase(dc) + t in K block kb -- the tiles of a diagonal side by side, so that
// one 4..16-byte load per lane fetches a whole K step's weights (the kernel is bound by the NUMBER of vector memory instructions:
// the CU's address unit takes ~16 cycles per wave instruction whatever its width):
//   row i = 4 o + r, slot r of the tile = (q, kh) with kw = c + q - kh, group g = 3 gb + q, input channel 4 (4 kb + k) + gid,
//   gid = (class - c) mod 4.  Zero where the chain has ended (tc >= L), for o >= cout, g >= G and unused slots.
__global__ void k_conv16dc_pack(const float *__restrict__ weight, float *__restrict__ packed, int nb, int G, int cout, int hidden, int n_gb, int NKB) {
    const long total = (long)nb * n_gb * 4 * NKB * XD_NT * 64;
    const int C = G * 4, nout = G * cout;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long t = e / (XD_NT * 64);
        const int w = (int)(e - t * (XD_NT * 64));                          // position inside the (class, K block) record
        int dcp = 0;
        while (dcp + 1 < XD_ND && xd_tbase(dcp + 1) * 64 <= w) ++dcp;
        const int nt = xd_ntiles(dcp), wl = w - xd_tbase(dcp) * 64, l = wl / nt, tile = xd_tbase(dcp) + wl % nt;
        const int kb = (int)(t % NKB); t /= NKB;
        const int cls = (int)(t & 3); t >>= 2;
        const int gb = (int)(t % n_gb), b = (int)(t / n_gb);
        const int i = l & 15, k = l >> 4, o = i >> 2, r = i & 3;
        const int dc = xd_dc_of_tile(tile), idx = (tile - xd_tbase(dc)) * 4 + r, sl = xd_slot(dc, idx);
        float v = 0.0f;
        if (sl >= 0 && o < cout) {
            const int q = sl >> 3, kh = sl & 7, c = dc - XD_C0, kw = c + q - kh, g = gb * XD_GB + q, gid = (cls - c + 16) & 3, tc = kb * 4 + k;
            int L = g + 4 + hidden - kh - kw;                                // extension/cconv_dc_cuda.cu:336-338
            if (L > G) L = G;
            if (g < G && tc < L) v = weight[(((long)b * nout + g * cout + o) * C + tc * 4 + gid) * 25 + kh * 5 + kw];
        }
        packed[e] = v;
    }
}

LIC360_API int lic360_conv16dc_supported(const lic360_conv_plan *p) { return p && conv16dc_ok(p) ? 1 : 0; }
LIC360_API long lic360_conv16dc_packed_floats(const lic360_conv_plan *p) {
    return p && conv16dc_ok(p) ? (long)conv16dc_ngb(p) * 4 * conv16dc_nkb(p) * XD_NT * 64 : 0;
}
LIC360_API int lic360_conv16dc_pack(void *stream, const lic360_conv_plan *p, const float *weight, int nb, float *packed) {
    ARG_CHECK(p && conv16dc_ok(p) && weight && packed && nb > 0);
    const long total = lic360_conv16dc_packed_floats(p) * nb;
    hipLaunchKernelGGL(k_conv16dc_pack, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, weight, packed, nb, p->ngroup, p->cout,
                       p->constrain == 5 ? 0 : 1, conv16dc_ngb(p), conv16dc_nkb(p));
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ kernel
struct XdArgs {
    const float *x, *packed, *bias, *act, *residual;
    float *out;
    int G, cout, hidden, H, W, npb, x_mod, N, psum;
    int ngb_all, gb_hi, n_gbv, NKB, HP;
    long SKP;
    int can_pair;                              // samples n and n + 8 always belong to the same stacked net (16 | samples per net)
};

typedef float xd_f2 __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ f32x4 xd_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

#define XD_ROW_SHL1 0x101                      // out[n] = in[n + 1] inside a row of 16 lanes
#define XD_ROW_SHR1 0x111                      // out[n] = in[n - 1]
template <int CTRL>
__device__ __forceinline__ float xd_row_shift(float edge, float v) {     // lanes without a source inside the row keep `edge`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

struct XdOps { float a[4]; };

// reference tree over the leaves of group Q's class CQ: F(i, 128) = leaf i, F(i, s) = F(i, 2s) + F(i + s, 2s), result F(CQ, 4)
template <int I, int S>
struct XdTree {
    static constexpr bool live = XdTree<I, S * 2>::live || XdTree<I + S, S * 2>::live;
    template <class F>
    static __device__ __forceinline__ float eval(F &&leaf) {
        if constexpr (!XdTree<I + S, S * 2>::live) return XdTree<I, S * 2>::eval(leaf);
        else if constexpr (!XdTree<I, S * 2>::live) return XdTree<I + S, S * 2>::eval(leaf);
        else return XdTree<I, S * 2>::eval(leaf) + XdTree<I + S, S * 2>::eval(leaf);
    }
};
template <int I>
struct XdTree<I, 128> {
    static constexpr bool live = I < 100;
    template <class F>
    static __device__ __forceinline__ float eval(F &&leaf) { return leaf(IC<I>{}); }
};

#ifdef XD_STAMP
// diagnostic build only: cycles per phase, summed per wave over the launch (no product code reads these)
__device__ unsigned long long xd_stamps[256 * 8 * 8];
#define XD_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[i] += t_ - t0; t0 = t_; } while (0)
LIC360_API int lic360_xd_stamps(unsigned long long *host_out, int clear) {
    if (host_out) HIP_TRY(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(xd_stamps), sizeof(xd_stamps)));
    if (clear) { static unsigned long long z[256 * 8 * 8]; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(xd_stamps), z, sizeof(z))); }
    return 0;
}
#else
#define XD_T(i)
#endif

template <int CLS>
__device__ __forceinline__ void xd_body(const XdArgs &a, float *halo, float *comb, const int lane, const int whalf) {
    const int G = a.G, H = a.H, W = a.W, S = H + W - 1, C = G * 4, nout = G * a.cout, HP = a.HP;
    const long SKP = a.SKP;
    const int SKP4 = (int)(4 * SKP);
    const int n16 = lane & 15, kl = lane >> 4;
    // ---- task list of this workgroup: XCD-aware static walk, heaviest group blocks first, boustrophedon (cconv4v6_dc.inc)
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x - xcd + 7) >> 3;
    const int ns_x = (a.N - xcd + 7) >> 3;                                  // samples of this XCD: n = xcd + 8 m
    // window of group block gb: the input rows its (up to) three diagonals read.  They fit 32 rows -> the task takes TWO samples,
    // one per row half, on the window [T0, T0 + 32) (return value = T0); otherwise one sample on rows 0..63 (return value -1).
    // (computed here from scalars: indexing a table in the kernel arguments with a run-time index makes hipcc treat the whole
    // task state as divergent)
    auto window_of = [&](int gb) __attribute__((always_inline)) {
        int lo = 1 << 30, hi = -1;
#pragma unroll
        for (int q = 0; q < XD_GB; ++q) {
            const int g = gb * XD_GB + q, sq = a.psum - g;
            const int l = sq >= W ? sq - W + 1 : 0, hh = sq < H ? sq : H - 1;
            if (g < G && sq >= 0 && sq < S) { lo = l < lo ? l : lo; hi = hh > hi ? hh : hi; }
        }
        const int lo_in = lo - 2 > 0 ? lo - 2 : 0, hi_in = hi + 2 < H - 1 ? hi + 2 : H - 1, t0 = lo_in & ~1;
        return (a.can_pair && hi_in - t0 + 1 <= 32) ? t0 : -1;
    };
    unsigned span_mask = 0;                                                 // bit j: block gb_hi - j takes one sample per task
    for (int j = 0; j < a.n_gbv; ++j) span_mask |= (window_of(a.gb_hi - j) < 0 ? 1u : 0u) << j;
    span_mask = __builtin_amdgcn_readfirstlane(span_mask);
#ifdef XD_EXP_HALFWG                                                          // experiment (timing only): 4-wave workgroups, one row half per task, no halo
    auto units_of = [&](int j) __attribute__((always_inline)) { return ((span_mask >> j) & 1u) ? 2 * ns_x : ns_x; };
#else
    auto units_of = [&](int j) __attribute__((always_inline)) { return ((span_mask >> j) & 1u) ? ns_x : (ns_x + 1) >> 1; };
#endif
    int n_my = 0;
    for (int j = 0; j < a.n_gbv; ++j) n_my += units_of(j);
    const float *const act_p = a.act ? a.act : a.bias, *const res_p = a.residual ? a.residual : a.x;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#ifdef XD_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- task descriptors.  The walk is software-pipelined: task k + 1 is decoded and its first operand loads are issued BEFORE
    // the epilogue of task k (the operand slots are free then), so no task starts by waiting for memory.
    struct Task {
        int tc0, s0, n_w, net, pbase, X, nKmax, half;
        bool span, valid_w;
        const char *xs, *ws;
    };
    int scan_j = 0, scan_base = 0;                                          // block of the walk's current task and its first task index
    auto decode = [&](int kt, Task &t) __attribute__((always_inline)) {
        const int u = kt * wgs_per_xcd + ((kt & 1) ? wgs_per_xcd - 1 - wg_in_xcd : wg_in_xcd);
        if (u >= n_my) return false;
        while (scan_j < a.n_gbv - 1 && u >= scan_base + units_of(scan_j)) { scan_base += units_of(scan_j); ++scan_j; }   // (u grows with kt)
        const int rem = u - scan_base, gb = a.gb_hi - scan_j;
        t.span = (span_mask >> scan_j) & 1u;
        t.tc0 = gb * XD_GB; t.s0 = a.psum - t.tc0;
        const int T0 = t.span ? 0 : window_of(gb);
#ifdef XD_EXP_HALFWG
        const int half = t.span ? (rem & 1) : 0;
        int n_w = t.span ? xcd + 8 * (rem >> 1) : xcd + 8 * rem;
        t.valid_w = n_w < a.N;
#else
        const int half = whalf;
        int n_w = t.span ? xcd + 8 * rem : xcd + 16 * rem + 8 * half;
        t.valid_w = n_w < a.N;
        if (!t.valid_w) n_w = xcd + 16 * rem;                               // the idle half of an odd pair recomputes sample A and stores nothing
#endif
        t.half = half;
        t.n_w = n_w;
        int net = 0;                                                        // n_w / npb without a division (few stacked nets)
        for (int q = a.npb; q <= n_w; q += a.npb) ++net;
        t.net = net;
        t.pbase = T0 + (t.span ? 32 * half : 0);
        t.X = t.tc0 + 4 + a.hidden + XD_C0;                                 // chain length of diagonal dc: min(G, X - dc)
        t.nKmax = ((t.X < G ? t.X : G) + 3) >> 2;
#ifdef XD_EXP_SAMEX                                                           // ablation: every sample reads sample xcd's activations (L2-resident)
        t.xs = (const char *)(a.x + (long)(xcd) * C * SKP + (long)t.s0 * HP + XD_COL0);
#else
        t.xs = (const char *)(a.x + (long)(n_w < a.x_mod ? n_w : n_w % a.x_mod) * C * SKP + (long)t.s0 * HP + XD_COL0);
#endif
        t.ws = (const char *)(a.packed + ((((long)net * a.ngb_all + gb) * 4 + CLS) * a.NKB) * (XD_NT * 64));
        return true;
    };
    // operand addresses = scalar base + 32-bit lane offset (no vector address arithmetic in the K loop): lane (k, n) reads channel
    // plane 4 (4 kb + k) + gid at rows pe, pe + 1 (4 | G: a K block never leaves the sample's planes)
    unsigned offx[4], offw[4];                                              // per input channel gid / per tile count of a diagonal
#pragma unroll
    for (int g = 0; g < 4; ++g) offw[g] = (unsigned)lane * 4u * (unsigned)(g + 1);
    int pe = 0;                                                             // first of this lane's two input rows
    auto lane_rows = [&](const Task &t) __attribute__((always_inline)) {
        pe = t.pbase + 2 * n16;
        { const int pmax = (H + 1) & ~1; if (pe > pmax) pe = pmax; }         // rows >= H: the zero columns behind the image
#pragma unroll
        for (int g = 0; g < 4; ++g) offx[g] = (unsigned)((kl * 4 + g) * (int)SKP + pe) * 4u;
    };
    typedef const __attribute__((address_space(1))) char *gptr;               // (explicitly global: a pointer that went through asm would be flat)
    const unsigned kbx = 16u * (unsigned)SKP4;                              // bytes between K blocks of x (< 2^32: one sample's planes)
    const unsigned hp4 = (unsigned)HP * 4u;
    f32x4 acc[XD_NT][2];
    // ---- K loop machinery: K blocks outer, the live diagonals of a block inner.  Diagonal dc is live in block kb while 4 kb < its
    // chain length, i.e. the live ones are the prefix dc < D(kb) = min(11, X - 4 kb).
        auto loadB = [&](auto dd, xd_f2 &b, gptr xk) __attribute__((always_inline)) {
            constexpr int dc = decltype(dd)::value, gid = (CLS - (dc - XD_C0) + 16) & 3;
            gptr xb = xk + (unsigned)dc * hp4;
            asm volatile("" : "+s"(xb), "+v"(offx[gid]));                   // scalar base + 32-bit lane offset: global_load v, v_off, s[base]
#ifdef XD_EXP_NOB                                                             // ablations (timing only, results are garbage)
            if (a.N < 0)
#endif
            b = *(const __attribute__((address_space(1))) xd_f2 *)(xb + offx[gid]);
        };
        auto loadA = [&](auto dd, XdOps &o, gptr wk) __attribute__((always_inline)) {
            constexpr int dc = decltype(dd)::value, T = xd_ntiles(dc), tb = xd_tbase(dc);
            typedef float xd_fT __attribute__((ext_vector_type(T == 3 ? 3 : T), aligned(T == 3 ? 4 : 4 * T)));
            gptr wb = wk + tb * 256;
            asm volatile("" : "+s"(wb), "+v"(offw[T - 1]));                 // (else hipcc widens the lane offset to 64 bits once and adds in the VALU)
#ifdef XD_EXP_NOA
            if (a.N < 0)
#endif
            {
                if constexpr (T == 1) o.a[0] = *(const __attribute__((address_space(1))) float *)(wb + offw[0]);
                else {
                    const xd_fT v = *(const __attribute__((address_space(1))) xd_fT *)(wb + offw[T - 1]);
#pragma unroll
                    for (int t = 0; t < T; ++t) o.a[t] = v[t];
                }
            }
        };
        auto fma = [&](auto dd, const XdOps &o, const xd_f2 &b) __attribute__((always_inline)) {
            constexpr int dc = decltype(dd)::value, T = xd_ntiles(dc), tb = xd_tbase(dc);
#pragma unroll
            for (int t = 0; t < T; ++t) {
#ifdef XD_EXP_NOMFMA
                acc[tb + t][0][0] += o.a[t] * b.x;
#else
                acc[tb + t][0] = xd_mfma(o.a[t], b.x, acc[tb + t][0]);
                acc[tb + t][1] = xd_mfma(o.a[t], b.y, acc[tb + t][1]);
#endif
            }
        };
        // operand slots: weights of diagonal dc in sa[dc % XD_RA], activations in sb[dc % XD_RB]; after the MFMAs of (kb, dc) a slot is
        // refilled with its next content, (kb, dc + R) or (kb + 1, dc % R): activations (may miss every cache) run further ahead
        XdOps sa[XD_RA];
        xd_f2 sb[XD_RB];
        // Three straight-line block bodies, for up to 11, 7 and 3 live diagonals: every load is unconditional, so hipcc's wait counts
        // are exact.  A block with D live diagonals runs in the smallest body >= D; a dead diagonal inside it multiplies zero weights
        // (the packed array is zero past a chain's end) with whatever its rows hold: exact, ~5 % more MFMAs.  Since D drops by 4 per
        // block, a task is [body 11]* [body 7]? [body 3]?  (Scalar conditions per diagonal instead: every step waited for ALL loads.
        // A switch over eleven bodies: hipcc spilled every accumulator around every MFMA.)
        auto body = [&](auto NN, auto NX, gptr xk, gptr xk1, gptr wk, gptr wk1) __attribute__((always_inline)) {
            constexpr int N = decltype(NN)::value, NEXT = decltype(NX)::value;   // NEXT: diagonals the following block's body runs
            constexpr int NA = NEXT < XD_RA ? NEXT : XD_RA, NB = NEXT < XD_RB ? NEXT : XD_RB;     // slots it expects filled
#ifndef XD_INTERLEAVE
            static_for<N>([&](auto dd) {
                constexpr int dc = decltype(dd)::value, la = dc % XD_RA, lb = dc % XD_RB;
                fma(dd, sa[la], sb[lb]);
                if constexpr (dc + XD_RB < N) loadB(IC<dc + XD_RB>{}, sb[lb], xk); else if constexpr (lb < NB) loadB(IC<lb>{}, sb[lb], xk1);
                if constexpr (dc + XD_RA < N) loadA(IC<dc + XD_RA>{}, sa[la], wk); else if constexpr (la < NA) loadA(IC<la>{}, sa[la], wk1);
                __builtin_amdgcn_sched_barrier(0);
            });
#else
            // experiment (tools/xd_lone.sh): the refill of a diagonal's operand slots is issued BETWEEN the first MFMAs of the following
            // diagonal, one load group behind each -- ~3 instructions hide in the 32-cycle shadow of an MFMA, nine in a row after the
            // last MFMA of a diagonal do not (tools/micro/mfma_issue.hip).  No effect with two waves per SIMD (116.3-119.5 us against 116.8).
            auto refillB = [&](auto dd) __attribute__((always_inline)) {
                constexpr int d = decltype(dd)::value, lb = d % XD_RB;
                if constexpr (d + XD_RB < N) loadB(IC<d + XD_RB>{}, sb[lb], xk); else if constexpr (lb < NB) loadB(IC<lb>{}, sb[lb], xk1);
            };
            auto refillA = [&](auto dd) __attribute__((always_inline)) {
                constexpr int d = decltype(dd)::value, la = d % XD_RA;
                if constexpr (d + XD_RA < N) loadA(IC<d + XD_RA>{}, sa[la], wk); else if constexpr (la < NA) loadA(IC<la>{}, sa[la], wk1);
            };
            static_for<N>([&](auto dd) {
                constexpr int dc = decltype(dd)::value, la = dc % XD_RA, lb = dc % XD_RB, T = xd_ntiles(dc), tb = xd_tbase(dc);
                acc[tb][0] = xd_mfma(sa[la].a[0], sb[lb].x, acc[tb][0]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (dc >= 1) refillB(IC<(dc >= 1 ? dc - 1 : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
                acc[tb][1] = xd_mfma(sa[la].a[0], sb[lb].y, acc[tb][1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (dc >= 1) refillA(IC<(dc >= 1 ? dc - 1 : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 1; t < T; ++t) {
                    acc[tb + t][0] = xd_mfma(sa[la].a[t], sb[lb].x, acc[tb + t][0]);
                    acc[tb + t][1] = xd_mfma(sa[la].a[t], sb[lb].y, acc[tb + t][1]);
                }
                if constexpr (dc == N - 1) { __builtin_amdgcn_sched_barrier(0); refillB(IC<N - 1>{}); refillA(IC<N - 1>{}); }
                __builtin_amdgcn_sched_barrier(0);
            });
#endif
        };
        auto fill = [&](const Task &t) __attribute__((always_inline)) {       // operands of block 0, the first diagonals (rows always exist)
            static_for<XD_RB>([&](auto ss) { loadB(ss, sb[decltype(ss)::value], (gptr)t.xs); });
            static_for<XD_RA>([&](auto ss) { loadA(ss, sa[decltype(ss)::value], (gptr)t.ws); });
        };
    Task cur, nxt;
    if (!decode(0, cur)) return;                                            // (uniform over the workgroup)
    lane_rows(cur);
    fill(cur);
    for (int kt = 0;; ++kt) {
        const int tc0 = cur.tc0, s0 = cur.s0, n_w = cur.n_w, net = cur.net, pbase = cur.pbase;
#ifdef XD_EXP_HALFWG
        const bool span = false, valid_w = cur.valid_w;
        const int half = 0;
#else
        const bool span = cur.span, valid_w = cur.valid_w;
        const int half = cur.half;
#endif
#pragma unroll
        for (int i = 0; i < XD_NT; ++i) { acc[i][0] = zero4; acc[i][1] = zero4; }
        XD_T(0);
        {
            gptr xk = (gptr)cur.xs, wk = (gptr)cur.ws;
            const int X = cur.X;
            int nKmax = cur.nKmax;
            asm volatile("" : "+s"(nKmax));
            int kb = 0, D = X;                                                  // live diagonals of block kb (uncapped)
            auto next_ptrs = [&](gptr &xk1, gptr &wk1) __attribute__((always_inline)) {
                const bool more = kb + 1 < nKmax;                               // the task's last block re-reads itself
                xk1 = xk + (more ? kbx : 0u); wk1 = wk + (more ? (unsigned)(XD_NT * 256) : 0u);
                asm volatile("" : "+s"(xk1), "+s"(wk1));
            };
            for (; kb < nKmax && D >= 8; ++kb, D -= 4) {
                gptr xk1, wk1;
                next_ptrs(xk1, wk1);
                body(IC<XD_ND>{}, IC<XD_ND>{}, xk, xk1, wk, wk1);                // next: body 11 or 7 (a dead diagonal's operands: harmless)
                xk = xk1; wk = wk1;
            }
            if (kb < nKmax && D >= 4) {
                gptr xk1, wk1;
                next_ptrs(xk1, wk1);
                body(IC<7>{}, IC<3>{}, xk, xk1, wk, wk1);
                xk = xk1; wk = wk1; ++kb; D -= 4;
            }
            if (kb < nKmax) body(IC<3>{}, IC<0>{}, xk, xk, wk, wk);
        }
        XD_T(1);
        // ---- the next task: decode, lane offsets, first operand loads (in flight during this task's epilogue)
        const int pe_e = pe;                                                // this task's rows, for the epilogue
        const bool have_next = decode(kt + 1, nxt);
        if (have_next) { lane_rows(nxt); fill(nxt); }
        // ---- epilogue operands of the waves that finish a group (class q < 3 finishes group q of its half): fetched after the K loop (they would cost 7 registers inside it), used after two barriers
        float e_bias = 0.f, e_act = 0.f;
        xd_f2 e_res = {0.f, 0.f};
        long e_oi = 0;
        bool e_ok0 = false, e_ok1 = false;
        if constexpr (CLS < XD_GB) {
            const int q = CLS, g = tc0 + q, sq = s0 - q, o = kl;
            const bool vq = valid_w && g < G && sq >= 0 && sq < S && o < a.cout;
            const int lo = sq >= W ? sq - W + 1 : 0, hi = sq < H ? sq : H - 1;
            const int p0 = pbase + 2 * n16;
            e_ok0 = vq && p0 >= lo && p0 <= hi;
            e_ok1 = vq && p0 + 1 >= lo && p0 + 1 <= hi;
            const int gc = g < G ? g : G - 1, oc = o < a.cout ? o : a.cout - 1, sc = sq < 0 ? 0 : (sq >= S ? S - 1 : sq);
            const int bid = net * nout + gc * a.cout + oc;
            e_oi = ((long)n_w * nout + gc * a.cout + oc) * SKP + (long)(sc + XD_ROW0) * HP + pe_e + XD_COL0;
            e_bias = a.bias[bid];
            e_act = act_p[bid];
            e_res = *(const xd_f2 *)(res_p + (a.residual ? e_oi : 0));
        }
#ifdef XD_EXP_NOEPI
        if (a.N < 0)
#endif
        {
        // ---- halo: the partner half's edge column of every shifted slot (only when the two halves are one sample)
        if (span) {
            if (half == 1) {
                if (n16 == 0) {
                    static_for<XD_ND>([&](auto dd) {
                        constexpr int dc = decltype(dd)::value;
                        static_for<xd_nslots(dc)>([&](auto ii) {
                            constexpr int idx = decltype(ii)::value, kh = xd_slot(dc, idx) & 7, tile = xd_tbase(dc) + idx / 4, reg = idx & 3;
                            static_for<2>([&](auto tt) {
                                constexpr int t = decltype(tt)::value;
                                if constexpr (xd_halo_has(kh, t, true)) {
                                    constexpr int hid = ((1 * 4 + CLS) * XD_NHALO + xd_halo_id(dc, idx, t, true)) * 4;   // (constexpr: else evaluated at run time)
                                    halo[hid + kl] = acc[tile][t][reg];
                                }
                            });
                        });
                    });
                }
            } else {
                if (n16 == 15) {
                    static_for<XD_ND>([&](auto dd) {
                        constexpr int dc = decltype(dd)::value;
                        static_for<xd_nslots(dc)>([&](auto ii) {
                            constexpr int idx = decltype(ii)::value, kh = xd_slot(dc, idx) & 7, tile = xd_tbase(dc) + idx / 4, reg = idx & 3;
                            static_for<2>([&](auto tt) {
                                constexpr int t = decltype(tt)::value;
                                if constexpr (xd_halo_has(kh, t, false)) {
                                    constexpr int hid = ((0 * 4 + CLS) * XD_NHALO + xd_halo_id(dc, idx, t, false)) * 4;
                                    halo[hid + kl] = acc[tile][t][reg];
                                }
                            });
                        });
                    });
                }
            }
        }
        XD_T(2);
        __syncthreads();
        XD_T(3);
        // ---- realignment + the reference tree inside the class, per group and output row tile
        // this wave has a partner above / below: its edge columns come from the halo (all ones), else they are zeros
        // (a wave without a partner reads the block of zeros instead: one address select per task, not one AND per register)
        const float *const halo_up = halo + ((span && half == 0) ? (1 * 4 + CLS) * XD_NHALO * 4 : 2 * 4 * XD_NHALO * 4) + kl;
        const float *const halo_dn = halo + ((span && half == 1) ? (0 * 4 + CLS) * XD_NHALO * 4 : 2 * 4 * XD_NHALO * 4) + kl;
        static_for<XD_GB>([&](auto qq) {
            constexpr int Q = decltype(qq)::value, CQ = (CLS + Q) & 3;
            static_for<2>([&](auto tt) {
                constexpr int TP = decltype(tt)::value;                      // output row = base + 2 n + TP
                auto leaf = [&](auto ii) __attribute__((always_inline)) -> float {
                    constexpr int i = decltype(ii)::value, tap = i % 25, kh = tap / 5, kw = tap % 5, c = kh + kw - Q, dc = c + XD_C0;
                    static_assert(dc >= 0 && dc < XD_ND && ((CLS - c + 16) & 3) == i / 25, "leaf i of class CQ lies on diagonal c in channel gid");
                    constexpr int idx = xd_slot_index(dc, Q, kh), tile = xd_tbase(dc) + idx / 4, reg = idx & 3, dl = kh - 2;
                    static_assert(idx >= 0, "slot table");
                    // output row r <- input row r + dl;  rows are (n, t) with row = base + 2 n + t
                    if constexpr (dl == 0) return acc[tile][TP][reg];
                    else if constexpr (dl == 1 && TP == 0) return acc[tile][1][reg];
                    else if constexpr (dl == -1 && TP == 1) return acc[tile][0][reg];
                    else if constexpr (dl > 0) {                              // needs column n + 1 of source tile TS
                        constexpr int TS = dl == 1 ? 0 : TP, hid = xd_halo_id(dc, idx, TS, true) * 4;
                        const float e = halo_up[hid];
                        return xd_row_shift<XD_ROW_SHL1>(e, acc[tile][TS][reg]);
                    } else {                                                  // column n - 1
                        constexpr int TS = dl == -1 ? 1 : TP, hid = xd_halo_id(dc, idx, TS, false) * 4;
                        const float e = halo_dn[hid];
                        return xd_row_shift<XD_ROW_SHR1>(e, acc[tile][TS][reg]);
                    }
                };
                const float part = XdTree<CQ, 4>::eval(leaf);
                comb[((((Q * 2 + TP) * 4 + CQ) * 2 + half) * 64) + lane] = part;
            });
        });
        XD_T(4);
        __syncthreads();
        XD_T(5);
        // ---- last two tree levels across the classes + bias / PReLU / residual / store: class q finishes group q of its half
        if constexpr (CLS < XD_GB) {
            constexpr int Q = CLS;
            float sv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float *cb = comb + (((Q * 2 + t) * 4) * 2 + half) * 64 + lane;
                const float f0 = cb[0], f1 = cb[2 * 64], f2 = cb[4 * 64], f3 = cb[6 * 64];
                float v = ((f0 + f2) + (f1 + f3)) + e_bias;
                if (a.act) { if (v < 0) v = v * e_act; }                      // cconv_dc_cuda.cu:360-362
                if (a.residual) v = v + (t == 0 ? e_res.x : e_res.y);        // fused TileAdd
                sv[t] = v;
            }
            if (e_ok0 && e_ok1) *(xd_f2 *)(a.out + e_oi) = (xd_f2){sv[0], sv[1]};
            else if (e_ok0) a.out[e_oi] = sv[0];
            else if (e_ok1) a.out[e_oi + 1] = sv[1];
        }
        }
        XD_T(6);
        if (!have_next) break;
        cur = nxt;
#ifdef XD_EXP_NOEPI
        if (acc[0][0][0] == 1.2345f && acc[23][1][3] == 5.f) a.out[0] = acc[5][0][1] + acc[11][1][2] + acc[17][0][0];   // keep the K loops alive
#endif
    }
#ifdef XD_STAMP
    if (lane == 0) {
        st[7] += __builtin_amdgcn_s_memtime() - t0;
        for (int i = 0; i < 8; ++i) xd_stamps[((blockIdx.x & 255) * 8 + whalf * 4 + CLS) * 8 + i] += st[i];
    }
#endif
}

#ifdef XD_EXP_HALFWG
#undef XD_THREADS
#define XD_THREADS 256
#endif
__global__ __launch_bounds__(XD_THREADS, 2) void k_cconv16dc(XdArgs a) {
    __shared__ float halo[(2 * 4 + 1) * XD_NHALO * 4];                      // [direction][class][register][channel] + a block of zeros
    for (int i = threadIdx.x; i < XD_NHALO * 4; i += XD_THREADS) halo[2 * 4 * XD_NHALO * 4 + i] = 0.f;
    __syncthreads();
    __shared__ float comb[XD_GB * 2 * 4 * 2 * 64];
#ifdef XD_EXP_LONE                                                            // experiment: one workgroup per CU (LDS ballast)
    __shared__ float ballast[24 * 1024];
    if (a.N < 0) { ballast[threadIdx.x] = 1.f; __syncthreads(); comb[threadIdx.x] = ballast[threadIdx.x ^ 1]; }
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cls = wave & 3, half = wave >> 2;
#ifdef XD_EXP_EMPTY
    if (a.N > 0) return;
#endif
#ifdef XD_ONE_CLASS
    (void)cls;
    xd_body<1>(a, halo, comb, lane, half);
#else
    switch (cls) {
        case 0: xd_body<0>(a, halo, comb, lane, half); break;
        case 1: xd_body<1>(a, halo, comb, lane, half); break;
        case 2: xd_body<2>(a, halo, comb, lane, half); break;
        default: xd_body<3>(a, halo, comb, lane, half); break;
    }
#endif
}

// x / residual / out: the zero-padded diagonal-major layout of lic360_dc4_layout (row0 = 6, col0 = 2); h <= 64.
LIC360_API int lic360_cconv16_dc_plane(void *stream, const lic360_conv_plan *p, const float *x, const float *packed, const float *bias,
                                       const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod) {
    ARG_CHECK(p && conv16dc_ok(p) && x && packed && bias && out && n > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n && h > 0 && h <= 64 && w > 0);
    const int G = p->ngroup, S = h + w - 1;
    if (psum < 0 || psum >= h + w + G - 2) return 0;
    int rows, pitch, row0, col0;
    if (lic360_dc4_layout(h, w, &rows, &pitch, &row0, &col0)) return 2;
    ARG_CHECK(row0 == XD_ROW0 && col0 == XD_COL0);
    XdArgs a;
    a.x = x; a.packed = packed; a.bias = bias; a.act = act; a.residual = residual; a.out = out;
    a.G = G; a.cout = p->cout; a.hidden = p->constrain == 5 ? 0 : 1; a.H = h; a.W = w; a.npb = n / nb; a.x_mod = x_mod; a.N = n; a.psum = psum;
    a.ngb_all = conv16dc_ngb(p); a.NKB = conv16dc_nkb(p); a.HP = pitch; a.SKP = (long)rows * pitch;
    a.can_pair = (a.npb % 16) == 0 ? 1 : 0;                                // samples n, n + 8 of an XCD's list then share a net
    int gb_lo = 1 << 30, gb_hi = -1;
    for (int gb = 0; gb < a.ngb_all; ++gb) {
        bool live = false;
        for (int q = 0; q < XD_GB; ++q) {
            const int g = gb * XD_GB + q, sq = psum - g;
            live = live || (g < G && sq >= 0 && sq < S);
        }
        if (!live) continue;
        if (gb < gb_lo) gb_lo = gb;
        if (gb > gb_hi) gb_hi = gb;
    }
    if (gb_hi < 0) return 0;
    a.gb_hi = gb_hi; a.n_gbv = gb_hi - gb_lo + 1;
    ARG_CHECK(a.n_gbv <= 32);
#ifdef XD_EXP_HALFWG
    hipLaunchKernelGGL(k_cconv16dc, dim3(512), dim3(XD_THREADS), 0, (hipStream_t)stream, a);
#else
    hipLaunchKernelGGL(k_cconv16dc, dim3(256), dim3(XD_THREADS), 0, (hipStream_t)stream, a);
#endif
    LAUNCH_CHECK();
    return 0;
}
