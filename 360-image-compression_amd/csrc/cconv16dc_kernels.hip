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
//   * K loop (round 4): operands come through LDS by LDS-DMA.  Measured on the round-3 form, which fetched A (one 4..16-byte
//     load per lane) and B (8 bytes per lane) straight into registers: the launch took as long with every MFMA removed as with
//     them (116 us), 93 us without the B loads -- the CU's vector-memory ADDRESS unit was the bound: a wave instruction costs
//     it ~18 cycles whatever its width (tools/micro/vmem_rate.hip: dword 9, dwordx2 / dwordx4 / LDS-DMA 18 cycles per CU), and a
//     K block needed 8 waves x 22 narrow loads = 2900 of the 3072 cycles its MFMAs take (profiles/r04_dc16_ablations.txt).
//     Now a STAGE = one K block of one class: the packed weights (6 KB, shared by the two row halves) and the 11 input
//     diagonals (1 KB each: 4 channel planes x 64 rows, both halves) arrive as 17 full-width global_load_lds_dwordx4 -- 1224
//     address-unit cycles per K block instead of 2900 --, issued one stage ahead into a double-buffered LDS image and spread
//     between the MFMAs of the running stage; the MFMA waves read their operands with ds_read (b32..b128 for A, b64 for B);
//     one barrier per stage.  Loop order = K blocks outer, the LIVE diagonals of the block inner (a diagonal's chains run
//     exactly ceil(L/4) blocks), one straight-line body per number of live diagonals.
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
#define XD_STAGE_A (XD_NT * 64)                // floats of packed weights per (class, K block)
#define XD_STAGE (XD_STAGE_A + XD_ND * 256)   // + 11 input diagonals x (4 channel planes x 64 rows): 17 KB per (buffer, class)
#ifndef XD_SPLIT
#define XD_SPLIT 0                            // 1: row half 0 issues all 17 LDS-DMAs of its class's next stage, half 1 none; 0: 9 + 8
#endif
#define XD_NDMA (XD_SPLIT ? 6 + XD_ND : 9)    // LDS-DMAs per issuing wave and stage (0: 3 weight chunks + 6 diagonals; row half 1: the 6th repeats its 5th)
#ifndef XD_PF
#define XD_PF 3                               // operand reads run this many diagonals ahead of their MFMAs
#endif
#ifndef XD_DSTRIDE
#define XD_DSTRIDE 2                          // one LDS-DMA of the next stage behind every XD_DSTRIDE-th MFMA, from the stage's first on
#endif
#ifndef XD_PRIO
#define XD_PRIO 1                             // issue priority of the two row halves (they share a SIMD): 0 none, 1 static for half 1, 2 swapped mid-stage, 3 alternating per diagonal
#endif
#ifndef XD_ORDER
#define XD_ORDER 1                            // 1: a stage walks its diagonals longest first (5, 4, 6, 3, 7, ...), 0: in index order
#endif
// the i-th diagonal a body with N live diagonals (dc < N) walks
__host__ __device__ constexpr int xd_walk(int N, int i) {
#if XD_ORDER
    constexpr int ord[11] = {5, 4, 6, 3, 7, 2, 8, 1, 9, 0, 10};
    int n = 0;
    for (int j = 0; j < 11; ++j) if (ord[j] < N) { if (n == i) return ord[j]; ++n; }
    return -1;
#else
    (void)N;
    return i;
#endif
}

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
// packed[net][gb][class][kb][diagonal dc][lane][t < ntiles(dc)] (the diagonal's block starts at float tbase(dc) * 64; a diagonal of THREE
// tiles is stored as [lane][2] followed by [lane][1], so that every LDS read of the kernel is a naturally aligned b32 / b64 / b128):
// lane l = 16 k + i carries A[row i][k] of the MFMA of tile tbase(dc) + t in K block kb.  One (class, K block) record = 24 tiles =
// 6 KB, copied verbatim into LDS by six 1 KB LDS-DMAs:
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
        const int nt = xd_ntiles(dcp), wl = w - xd_tbase(dcp) * 64;
        int l, tt;
        if (nt == 3) { if (wl < 128) { l = wl >> 1; tt = wl & 1; } else { l = wl - 128; tt = 2; } }
        else { l = wl / nt; tt = wl % nt; }
        const int tile = xd_tbase(dcp) + tt;
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
    int rs;                                    // samples per XCD and ROUND of the task walk (0: one round with every sample)
};

typedef float xd_f2 __attribute__((ext_vector_type(2)));
typedef float xd_f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ f32x4 xd_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

#define XD_ROW_SHL1 0x101                      // out[n] = in[n + 1] inside a row of 16 lanes
#define XD_ROW_SHR1 0x111                      // out[n] = in[n - 1]
template <int CTRL>
__device__ __forceinline__ float xd_row_shift(float edge, float v) {     // lanes without a source inside the row keep `edge`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// LDS-DMA (the idiom of cconv16_kernels.hip): lane l's 16 bytes at sbase + voff land at LDS byte lds + 16 l.  Scalar base + 32-bit
// lane offset: no vector address arithmetic.  hipcc does not see these VMEM operations: completion is enforced by hand (XD_WAIT0
// before the barrier that publishes a stage).  M0 is written in the statement that reads it.
typedef const __attribute__((address_space(1))) char *xd_gptr;            // (explicitly global: a pointer that went through asm would be flat)
__device__ __forceinline__ void xd_dma(unsigned voff, xd_gptr sbase, unsigned lds) {
    // (the base is wave-uniform by construction; where hipcc cannot prove it -- it keeps such a value in a VGPR -- the two
    // v_readfirstlane make it so; where it can, they fold away)
    const unsigned long long pb = (unsigned long long)sbase;
    sbase = (xd_gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)pb));
    lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
#define XD_WAIT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
__device__ __forceinline__ unsigned xd_lds_addr(const float *p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const float *)p; }

struct XdOps { float a[4]; xd_f2 b; };

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
__device__ unsigned long long xd_stamps[256 * 8 * 10];
#define XD_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[i] += t_ - t0; t0 = t_; } while (0)
LIC360_API int lic360_xd_stamps(unsigned long long *host_out, int clear) {
    if (host_out) HIP_TRY(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(xd_stamps), sizeof(xd_stamps)));
    if (clear) { static unsigned long long z[256 * 8 * 10]; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(xd_stamps), z, sizeof(z))); }
    return 0;
}
#else
#define XD_T(i)
#endif

template <int CLS, int HALF>
__device__ __forceinline__ void xd_body(const XdArgs &a, float *ring, float *halo, float *comb, const int lane) {
    constexpr int whalf = HALF;
    const int G = a.G, H = a.H, W = a.W, S = H + W - 1, C = G * 4, nout = G * a.cout, HP = a.HP;
    const long SKP = a.SKP;
    const int n16 = lane & 15, kl = lane >> 4;
    // ---- task list of this workgroup: XCD-aware static walk, heaviest group blocks first, boustrophedon (cconv4v6_dc.inc)
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x - xcd + 7) >> 3;
    const int ns_x = (a.N - xcd + 7) >> 3;                                  // samples of this XCD: n = xcd + 8 m
    // ROUNDS: the walk below (heaviest group blocks first) runs over RS samples of the XCD at a time.  Every group block of a sample
    // reads the same band of input diagonals (block gb the channels of groups < 3 gb + 8); inside a round the bands stay in the
    // XCD's L2 between the sweeps of the group blocks.
    const int RS = a.rs > 0 ? a.rs : ns_x, n_rounds = a.rs > 0 ? ns_x / a.rs : 1;   // (host: 8 rs | N)
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
    auto units_of = [&](int j) __attribute__((always_inline)) { return ((span_mask >> j) & 1u) ? RS : (RS + 1) >> 1; };
    int n_round = 0;                                                        // tasks of one round
    for (int j = 0; j < a.n_gbv; ++j) n_round += units_of(j);
    const int n_my = n_round * n_rounds;
    const float *const act_p = a.act ? a.act : a.bias, *const res_p = a.residual ? a.residual : a.x;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#ifdef XD_STAMP
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- task descriptors.  The walk is software-pipelined: task k + 1 is decoded BEFORE the K loop of task k, whose last stage
    // issues the LDS-DMAs of task k + 1's first stage.
    struct Task {
        int tc0, s0, n_w, net, pbase, X, nKmax, half;
        unsigned d0, d1;                                                    // byte distance of row half 0's / 1's sample from xs
        bool span, valid_w;
        xd_gptr xs, ws;
    };
    int scan_j = 0, scan_base = 0, round = 0;                               // block of the walk's current task, its first task index, the round
    auto decode = [&](int kt, Task &t) __attribute__((always_inline)) {
        const int u = kt * wgs_per_xcd + ((kt & 1) ? wgs_per_xcd - 1 - wg_in_xcd : wg_in_xcd);
        if (u >= n_my) return false;
        while (u >= (round + 1) * n_round) { ++round; scan_j = 0; scan_base = round * n_round; }                          // (u grows with kt)
        while (scan_j < a.n_gbv - 1 && u >= scan_base + units_of(scan_j)) { scan_base += units_of(scan_j); ++scan_j; }   // (u grows with kt)
        const int rem = u - scan_base, gb = a.gb_hi - scan_j;
        t.span = (span_mask >> scan_j) & 1u;
        t.tc0 = gb * XD_GB; t.s0 = a.psum - t.tc0;
        const int T0 = t.span ? 0 : window_of(gb);
        const int half = whalf;
        // samples of the two row halves: one sample (span) or the pair (n, n + 8); the idle half of an odd pair recomputes sample A
        // and stores nothing
        const int m0 = round * RS;                                          // the round's samples: n = xcd + 8 (m0 + i), i < RS
        const bool lone = !t.span && 2 * rem + 1 >= RS;                     // the last pair of an odd round has one sample
        const int nA = xcd + 8 * (m0 + (t.span ? rem : 2 * rem)), nB = (t.span || lone || nA + 8 >= a.N) ? nA : nA + 8;
        int n_w = half ? nB : nA;
        t.valid_w = t.span || half == 0 || nB != nA;
        t.half = half;
        t.n_w = n_w;
        int net = 0;                                                        // n_w / npb without a division (few stacked nets)
        for (int q = a.npb; q <= n_w; q += a.npb) ++net;
        t.net = net;
        t.pbase = T0 + (t.span ? 32 * half : 0);
        t.X = t.tc0 + 4 + a.hidden + XD_C0;                                 // chain length of diagonal dc: min(G, X - dc)
        t.nKmax = ((t.X < G ? t.X : G) + 3) >> 2;
        // the task's input: both halves' samples relative to the lower of the two (the lane offsets of a DMA are unsigned)
        const int iA = nA < a.x_mod ? nA : nA % a.x_mod, iB = nB < a.x_mod ? nB : nB % a.x_mod, i0 = iA < iB ? iA : iB;
        const long sample_bytes = (long)C * SKP * 4;
        t.d0 = (unsigned)((iA - i0) * sample_bytes); t.d1 = (unsigned)((iB - i0) * sample_bytes);    // (< 2^32: checked by the host)
        t.xs = (xd_gptr)(a.x + (long)i0 * C * SKP + (long)t.s0 * HP + XD_COL0);
        t.ws = (xd_gptr)(a.packed + ((((long)net * a.ngb_all + gb) * 4 + CLS) * a.NKB) * XD_STAGE_A);
        t.pbase = T0 + (t.span ? 32 * half : 0);
        // rows of the two halves inside the 64-row LDS image: span: rows 0..63 of one sample; pair: the window [T0, T0 + 32) of each
        t.tc0 = gb * XD_GB;
        return true;
    };
    // ---- LDS-DMA lane offsets.  A: lane l copies bytes [16 l, 16 l + 16) of a 1 KB chunk.  B (one input diagonal of one K block,
    // both row halves): lane l = 32 kk + 16 h + 8 k1 + m fetches rows r0(h) + 4 m .. + 3 of channel plane k = 2 kk + k1 of half h's
    // sample, and its 16 bytes land in slot l -- the slot order that makes the MFMA waves' ds_read_b64 conflict-free (a 32-lane group
    // reads the 16 slots {16 kk' + 8 k1 + m} of its half, distinct modulo 16).  Rows >= H + 2 hold whatever lies behind the row in
    // the layout (finite activations or zeros): they only reach outputs of rows >= H, which are never stored.
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma_lane_offset = [&](const Task &t) __attribute__((always_inline)) -> unsigned {
        const int h = (lane >> 4) & 1, k = 2 * (lane >> 5) + ((lane >> 3) & 1), m = lane & 7;
        int r = (t.span ? 32 * h : t.pbase) + 4 * m;                         // (pair: pbase = T0 for both halves)
        if (r >= H + 2) r = H - 2 > 0 ? H - 2 : 0;                           // stay inside the sample's planes
        return (unsigned)((4 * k * (int)SKP + r) * 4) + (h ? t.d1 : t.d0);
    };
    const unsigned hp4 = (unsigned)HP * 4u, skp4 = (unsigned)SKP * 4u;
    const unsigned kbx = 16u * skp4;                                        // bytes between K blocks of x (< 2^32: one sample's planes)
    // the LDS-DMAs of one wave for one stage (weights of K block at wk, activations at xk) into the stage image at LDS byte `dst`
    auto dma_one = [&](auto jj, xd_gptr xk, xd_gptr wk, unsigned voff, unsigned dst) __attribute__((always_inline)) {
        constexpr int j = decltype(jj)::value;
#if XD_SPLIT
        // The two row halves of a class share a SIMD and the hardware arbitrates its issue winner-takes-all (by priority, then age): one
        // of them runs ahead, the other finishes alone.  So the roles are made explicit: half 0 issues ALL the class's DMAs at the
        // start of a stage, while half 1 -- at priority 1 -- has the matrix pipe to itself; half 0's MFMAs fill in behind.
        if constexpr (HALF == 0) {
            if constexpr (j < 6) xd_dma(lane16, wk + j * 1024u, dst + j * 1024u);
            else {
                constexpr int dc = j - 6;
                constexpr unsigned gid = (unsigned)(CLS + XD_C0 + 16 - dc) & 3u;    // input channel inside a group: (class - c) mod 4, c = dc - XD_C0
                xd_dma(voff, xk + ((unsigned)dc * hp4 + gid * skp4), dst + (unsigned)(XD_STAGE_A + dc * 256) * 4u);
            }
        }
#else
        if constexpr (j < 3) {
            const unsigned ch = (3u * (unsigned)whalf + j) * 1024u;
            xd_dma(lane16, wk + ch, dst + ch);
        } else {
            int dc = 6 * whalf + (j - 3);
            dc = dc > XD_ND - 1 ? XD_ND - 1 : dc;
            const unsigned gid = (unsigned)(CLS + XD_C0 + 16 - dc) & 3u;    // input channel inside a group: (class - c) mod 4, c = dc - XD_C0
            xd_dma(voff, xk + ((unsigned)dc * hp4 + gid * skp4), dst + (unsigned)(XD_STAGE_A + dc * 256) * 4u);
        }
#endif
    };
    // ---- MFMA-side operand reads: lane (k, n) of half h reads its two rows (8 bytes) of plane k from slot 32 (k >> 1) + 16 h + 8 (k & 1) + (n >> 1)
    const int boff = 4 * (32 * (kl >> 1) + 16 * whalf + 8 * (kl & 1) + (n16 >> 1)) + 2 * (n16 & 1);   // floats inside a diagonal's 1 KB chunk
    f32x4 acc[XD_NT][2];
    auto load_ops = [&](auto dd, XdOps &o, const float *sA) __attribute__((always_inline)) {
        constexpr int dc = decltype(dd)::value, T = xd_ntiles(dc), tb = xd_tbase(dc);
        const float *p = sA + tb * 64;
        if constexpr (T == 1) o.a[0] = p[lane];
        else if constexpr (T == 2) { const xd_f2 v = *(const xd_f2 *)(p + 2 * lane); o.a[0] = v.x; o.a[1] = v.y; }
        else if constexpr (T == 3) { const xd_f2 v = *(const xd_f2 *)(p + 2 * lane); o.a[0] = v.x; o.a[1] = v.y; o.a[2] = p[128 + lane]; }
        else { const f32x4 v = *(const f32x4 *)(p + 4 * lane); o.a[0] = v[0]; o.a[1] = v[1]; o.a[2] = v[2]; o.a[3] = v[3]; }
        o.b = *(const xd_f2 *)(sA + XD_STAGE_A + dc * 256 + boff);
    };
    // the MFMAs of diagonal dc; `hook(IC<m>{})` runs behind the m-th MFMA of the stage (the LDS-DMAs of the next stage go there, one
    // per MFMA from the stage's first on: a DMA issued that early has the whole stage to land, and its ~6 scalar instructions sit
    // in the shadow of the MFMA in front of it)
    auto fma = [&](auto dd, auto mm0, const XdOps &o, auto &&hook) __attribute__((always_inline)) {
        constexpr int dc = decltype(dd)::value, T = xd_ntiles(dc), tb = xd_tbase(dc), M0 = decltype(mm0)::value;   // M0: MFMAs of the stage before this diagonal
        static_for<T>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            acc[tb + t][0] = xd_mfma(o.a[t], o.b.x, acc[tb + t][0]);
            hook(IC<M0 + 2 * t>{});
            acc[tb + t][1] = xd_mfma(o.a[t], o.b.y, acc[tb + t][1]);
            hook(IC<M0 + 2 * t + 1>{});
        });
    };
    // One stage = one K block: N live diagonals (three straight-line bodies, for up to 11, 7 and 3 live diagonals; a block with D live
    // diagonals runs in the smallest body >= D: a dead diagonal inside it multiplies zero weights -- the packed array is zero past a
    // chain's end -- with whatever its rows hold: exact, ~5 % more MFMAs).  Operand reads run two diagonals ahead of their MFMAs; the
    // LDS-DMAs of the NEXT stage go behind the stage's first nine MFMAs (the workgroup's very last stage fetches itself again:
    // straight-line code, nobody reads that image); the stage ends with the wait + barrier that publishes the next stage's image and
    // frees this one (with the DMAs spread over the whole stage, that wait was 28 % of a wave's time: the late ones had not landed).
    unsigned stage = 0;                                                     // parity = the LDS buffer the current stage reads
    auto stage_image = [&](unsigned par) __attribute__((always_inline)) { return ring + ((par & 1u) * 4 + CLS) * XD_STAGE; };
    auto body = [&](auto NN, xd_gptr xk1, xd_gptr wk1, unsigned voff1) __attribute__((always_inline)) {
        constexpr int N = decltype(NN)::value, NM = 2 * xd_tbase(N);        // MFMAs of the stage
        const float *sA = stage_image(stage);
        const unsigned dst = xd_lds_addr(stage_image(stage + 1));
        auto hook = [&](auto mm) __attribute__((always_inline)) {
            constexpr int m = decltype(mm)::value;
            if constexpr (XD_DSTRIDE > 0 && m % (XD_DSTRIDE > 0 ? XD_DSTRIDE : 1) == 0 && m / (XD_DSTRIDE > 0 ? XD_DSTRIDE : 1) < XD_NDMA) {
                __builtin_amdgcn_sched_barrier(0);
                dma_one(IC<m / XD_DSTRIDE>{}, xk1, wk1, voff1, dst);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        XdOps ops[XD_PF + 1];
        if constexpr (XD_DSTRIDE == 0) static_for<XD_NDMA>([&](auto jj) { dma_one(jj, xk1, wk1, voff1, dst); });   // all of them up front
        static_for<(XD_PF < N ? XD_PF : N)>([&](auto ii) { load_ops(IC<xd_walk(N, decltype(ii)::value)>{}, ops[decltype(ii)::value], sA); });
        static_for<N>([&](auto ii) {
            constexpr int i = decltype(ii)::value, dc = xd_walk(N, i);
            constexpr int m0 = [] { int m = 0; for (int j = 0; j < i; ++j) m += 2 * xd_ntiles(xd_walk(N, j)); return m; }();
            if constexpr (i + XD_PF < N) load_ops(IC<xd_walk(N, i + XD_PF)>{}, ops[(i + XD_PF) % (XD_PF + 1)], sA);
#if XD_PRIO == 3
            if ((i + whalf) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#elif XD_PRIO == 2
            if constexpr (i == 0) { if (whalf) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
            if constexpr (i == (N > 7 ? 4 : (N > 3 ? 3 : 1))) { if (whalf) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
            __builtin_amdgcn_sched_barrier(0);
            fma(IC<dc>{}, IC<m0>{}, ops[i % (XD_PF + 1)], hook);
            __builtin_amdgcn_sched_barrier(0);
        });
        constexpr int HOOKED = XD_DSTRIDE > 0 ? (NM + (XD_DSTRIDE > 0 ? XD_DSTRIDE : 1) - 1) / (XD_DSTRIDE > 0 ? XD_DSTRIDE : 1) : XD_NDMA;   // DMAs the hooks issued
        static_for<(HOOKED < XD_NDMA ? XD_NDMA - HOOKED : 0)>([&](auto rr) { dma_one(IC<HOOKED + decltype(rr)::value>{}, xk1, wk1, voff1, dst); });
        XD_T(1);
        XD_WAIT0();
        XD_T(7);                                                            // (diagnostic build: the wait for this wave's own DMAs ...
        __syncthreads();
        XD_T(8);                                                            //  ... and the stage barrier)
        ++stage;
    };
    Task cur, nxt;
    if (!decode(0, cur)) return;                                            // (uniform over the workgroup)
    unsigned voff = dma_lane_offset(cur);
    {   // the first stage of the first task
        const unsigned dst = xd_lds_addr(stage_image(0));
        static_for<XD_NDMA>([&](auto jj) { dma_one(jj, cur.xs, cur.ws, voff, dst); });
        XD_WAIT0();
        __syncthreads();
    }
    for (int kt = 0;; ++kt) {
        const int tc0 = cur.tc0, s0 = cur.s0, n_w = cur.n_w, net = cur.net, pbase = cur.pbase;
        const bool span = cur.span, valid_w = cur.valid_w;
        const int half = cur.half;
        const bool have_next = decode(kt + 1, nxt);
        const unsigned voff_n = have_next ? dma_lane_offset(nxt) : voff;
#pragma unroll
        for (int i = 0; i < XD_NT; ++i) { acc[i][0] = zero4; acc[i][1] = zero4; }
        XD_T(0);
        {
            xd_gptr xk = cur.xs, wk = cur.ws;
            const int X = cur.X;
            int nKmax = cur.nKmax;
            asm volatile("" : "+s"(nKmax));
            int kb = 0, D = X;                                                  // live diagonals of block kb (uncapped)
            // operands of the stage after block kb: block kb + 1 of this task, or block 0 of the next one (none: this block again)
            auto next_stage = [&](xd_gptr &xk1, xd_gptr &wk1, unsigned &v1) __attribute__((always_inline)) {
                const bool inner = kb + 1 < nKmax;
                xk1 = inner ? xk + kbx : (have_next ? nxt.xs : xk);
                wk1 = inner ? wk + (unsigned)(XD_STAGE_A * 4) : (have_next ? nxt.ws : wk);
                v1 = inner ? voff : voff_n;
                asm volatile("" : "+s"(xk1), "+s"(wk1));
            };
            for (; kb < nKmax && D >= 8; ++kb, D -= 4) {
                xd_gptr xk1, wk1; unsigned v1;
                next_stage(xk1, wk1, v1);
                body(IC<XD_ND>{}, xk1, wk1, v1);
                xk = xk1; wk = wk1;
            }
            if (kb < nKmax && D >= 4) {
                xd_gptr xk1, wk1; unsigned v1;
                next_stage(xk1, wk1, v1);
                body(IC<7>{}, xk1, wk1, v1);
                xk = xk1; wk = wk1; ++kb; D -= 4;
            }
            if (kb < nKmax) {
                xd_gptr xk1, wk1; unsigned v1;
                next_stage(xk1, wk1, v1);
                body(IC<3>{}, xk1, wk1, v1);
            }
        }
        XD_T(1);
        // ---- epilogue operands of the waves that finish a group (class q < 3 finishes group q of its half): fetched after the K loop (they would cost 7 registers inside it), used after two barriers
        const int pe_e = pbase + 2 * n16;                                   // first of this lane's two rows
        float e_bias = 0.f, e_act = 0.f;
        xd_f2 e_res = {0.f, 0.f};
        long e_oi = 0;
        bool e_ok0 = false, e_ok1 = false;
        if constexpr (CLS < XD_GB) {
            const int q = CLS, g = tc0 + q, sq = s0 - q, o = kl;
            const bool vq = valid_w && g < G && sq >= 0 && sq < S && o < a.cout;
            const int lo = sq >= W ? sq - W + 1 : 0, hi = sq < H ? sq : H - 1;
            const int p0 = pe_e;
            e_ok0 = vq && p0 >= lo && p0 <= hi;
            e_ok1 = vq && p0 + 1 >= lo && p0 + 1 <= hi;
            const int gc = g < G ? g : G - 1, oc = o < a.cout ? o : a.cout - 1, sc = sq < 0 ? 0 : (sq >= S ? S - 1 : sq);
            const int bid = net * nout + gc * a.cout + oc;
            const int pc = pe_e > ((H + 1) & ~1) ? ((H + 1) & ~1) : pe_e;    // (rows behind the image: any address inside the row)
            e_oi = ((long)n_w * nout + gc * a.cout + oc) * SKP + (long)(sc + XD_ROW0) * HP + pc + XD_COL0;
            e_bias = a.bias[bid];
            e_act = act_p[bid];
            e_res = *(const xd_f2u *)(res_p + (a.residual ? e_oi : 0));
        }
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
            if (e_ok0 && e_ok1) *(xd_f2u *)(a.out + e_oi) = (xd_f2u){sv[0], sv[1]};
            else if (e_ok0) a.out[e_oi] = sv[0];
            else if (e_ok1) a.out[e_oi + 1] = sv[1];
        }
        XD_T(6);
        if (!have_next) break;
        cur = nxt;
        voff = voff_n;
    }
#ifdef XD_STAMP
    if (lane == 0) {
        for (int i = 0; i < 10; ++i) xd_stamps[((blockIdx.x & 255) * 8 + HALF * 4 + CLS) * 10 + i] += st[i];
    }
#endif
}

__global__ __launch_bounds__(XD_THREADS, 2) void k_cconv16dc(XdArgs a) {
    __shared__ __attribute__((aligned(16))) float ring[2 * 4 * XD_STAGE];   // [buffer][class][weights 6 KB | 11 diagonals x 1 KB]: 136 KB
    __shared__ float halo[(2 * 4 + 1) * XD_NHALO * 4];                      // [direction][class][register][channel] + a block of zeros
    __shared__ float comb[XD_GB * 2 * 4 * 2 * 64];
    for (int i = threadIdx.x; i < XD_NHALO * 4; i += XD_THREADS) halo[2 * 4 * XD_NHALO * 4 + i] = 0.f;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cls = wave & 3, half = wave >> 2;
#if XD_PRIO == 1
    // The two row halves of a class share a SIMD and every stage ends in a barrier: with equal priorities the older wave (half 0) wins
    // the issue arbitration, runs its 48 MFMAs in 2400 cycles and then waits 1700 at the barrier while the younger one finishes alone
    // at a lone wave's pace (stamps: K loops 153 k / 233 k cycles per launch, barrier waits 110 k / 9 k).  Static priority for the
    // younger half (MI355X_MICROARCH.md, "Two waves per SIMD", item 4).
    if (half) __builtin_amdgcn_s_setprio(1);
#endif
    switch (wave) {
        case 0: xd_body<0, 0>(a, ring, halo, comb, lane); break;
        case 1: xd_body<1, 0>(a, ring, halo, comb, lane); break;
        case 2: xd_body<2, 0>(a, ring, halo, comb, lane); break;
        case 3: xd_body<3, 0>(a, ring, halo, comb, lane); break;
        case 4: xd_body<0, 1>(a, ring, halo, comb, lane); break;
        case 5: xd_body<1, 1>(a, ring, halo, comb, lane); break;
        case 6: xd_body<2, 1>(a, ring, halo, comb, lane); break;
        default: xd_body<3, 1>(a, ring, halo, comb, lane); break;
    }
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
    ARG_CHECK(8L * G * 4 * rows * pitch * 4 < (1L << 32));                  // a pair's samples (n, n + 8) inside one 32-bit lane offset
    XdArgs a;
    a.x = x; a.packed = packed; a.bias = bias; a.act = act; a.residual = residual; a.out = out;
    a.G = G; a.cout = p->cout; a.hidden = p->constrain == 5 ? 0 : 1; a.H = h; a.W = w; a.npb = n / nb; a.x_mod = x_mod; a.N = n; a.psum = psum;
    a.ngb_all = conv16dc_ngb(p); a.NKB = conv16dc_nkb(p); a.HP = pitch; a.SKP = (long)rows * pitch;
    a.can_pair = (a.npb % 16) == 0 && x_mod == n ? 1 : 0;                  // samples n, n + 8 of an XCD's list then share a net (and x is not shared)
    // rounds of the task walk: LIC360_DC_RS=<k>: k samples per XCD and round (default 0: one round)
    static const int rs_env = [] { const char *e = getenv("LIC360_DC_RS"); return e ? atoi(e) : 0; }();
    a.rs = 0;
    if (rs_env > 0 && n % 8 == 0 && (n / 8) % rs_env == 0 && (rs_env % 2 == 0 || !a.can_pair)) a.rs = rs_env;
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
    hipLaunchKernelGGL(k_cconv16dc, dim3(256), dim3(XD_THREADS), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return 0;
}

// ================================================================================================ class-sequential form (round 4)
// k_cconv16dq: the same mapping, arithmetic, packed weights and LDS-DMA stages as k_cconv16dc, with the work of a task laid out in TIME
// instead of over waves.  In k_cconv16dc the eight waves of a workgroup are 4 lane classes x 2 row halves of one (sample, group block): they
// run in lockstep (a barrier per stage, two per task), the two waves of a SIMD are the two halves of one class -- and the hardware
// arbitrates a SIMD's issue winner-takes-all, so one half runs ahead and waits at every barrier while the other finishes alone at a
// lone wave's pace (stamps, profiles/r04_dc16_ldsdma_counters.txt: 17 % of a wave's time in stage barriers, the matrix pipe 53 % busy).
// Here a workgroup has FOUR waves = 2 samples x 2 row halves (or 4 samples where a block's diagonals fit 32 rows) of one group block,
// and every wave walks the four lane classes one after the other on the same 192 accumulators:
//   * two workgroups per CU: the two waves of a SIMD belong to DIFFERENT workgroups (or kernels of different streams), drift apart, and
//     the epilogue / barrier / stage-start bubbles of one are filled with the other's MFMAs;
//   * the packed weights of a stage are shared by the workgroup's four waves (28 LDS-DMAs per stage and workgroup: 6 + 2 x 11);
//   * no class combine through LDS: a wave meets all four class partials of its outputs itself, (F0 + F2) + (F1 + F3) in registers;
//     what is left between waves is the halo of the two row halves of a sample, once per class pass.
#define XQ_THREADS 256
#define XQ_IMG (XD_STAGE_A + 2 * XD_ND * 256)  // floats of one stage image: weights | input diagonals of wave pair 0 | of wave pair 1 (28 KB)
#ifndef XQ_PF
#define XQ_PF 2
#endif
#ifndef XQ_DSTRIDE
#define XQ_DSTRIDE 2
#endif

template <int WV>
__device__ __forceinline__ void xq_body(const XdArgs &a, float *ringA, float *ringB, float *halo, float *hs_all, const int lane) {
    float *const hs = hs_all + WV * (XD_GB * 2 * 2 * 64) + lane;            // this lane's class partials [group][output row][class parity] (LDS: 12 registers too many)
    constexpr int PAIR = WV >> 1, HALF = WV & 1;                            // wave = (sample slot pair, row half)
    constexpr int NA_DMA = WV < 2 ? 2 : 1;                                  // this wave's share of a stage's six weight chunks (chunk WV + 4 j)
    const int G = a.G, H = a.H, W = a.W, S = H + W - 1, C = G * 4, nout = G * a.cout, HP = a.HP;
    const long SKP = a.SKP;
    const int n16 = lane & 15, kl = lane >> 4;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (gridDim.x - xcd + 7) >> 3;
    const int spn = a.npb >> 3;                                             // samples per stacked net and XCD (host: 8 | npb, 8 | N): n = xcd + 8 i
    const int nb = a.N / a.npb;
    auto window_of = [&](int gb) __attribute__((always_inline)) {             // as in xd_body
        int lo = 1 << 30, hi = -1;
#pragma unroll
        for (int q = 0; q < XD_GB; ++q) {
            const int g = gb * XD_GB + q, sq = a.psum - g;
            const int l = sq >= W ? sq - W + 1 : 0, hh = sq < H ? sq : H - 1;
            if (g < G && sq >= 0 && sq < S) { lo = l < lo ? l : lo; hi = hh > hi ? hh : hi; }
        }
        const int lo_in = lo - 2 > 0 ? lo - 2 : 0, hi_in = hi + 2 < H - 1 ? hi + 2 : H - 1, t0 = lo_in & ~1;
        return hi_in - t0 + 1 <= 32 ? t0 : -1;
    };
    unsigned span_mask = 0;                                                 // bit j: block gb_hi - j has 64-row diagonals: 2 samples per task (else 4)
    for (int j = 0; j < a.n_gbv; ++j) span_mask |= (window_of(a.gb_hi - j) < 0 ? 1u : 0u) << j;
    span_mask = __builtin_amdgcn_readfirstlane(span_mask);
    const int upn_span = (spn + 1) >> 1, upn_pair = (spn + 3) >> 2;          // tasks per net and group block
#ifdef XQ_NETMAJOR                                                            // (experiment) walk net by net: the bands of a net's 6 samples stay in L2
    auto units_of = [&](int j) __attribute__((always_inline)) { return ((span_mask >> j) & 1u) ? upn_span : upn_pair; };
    int per_net = 0;
    for (int j = 0; j < a.n_gbv; ++j) per_net += units_of(j);
    const int n_my = per_net * nb;
    int wnet = 0;
#else
    auto units_of = [&](int j) __attribute__((always_inline)) { return nb * (((span_mask >> j) & 1u) ? upn_span : upn_pair); };
    int n_my = 0;
    for (int j = 0; j < a.n_gbv; ++j) n_my += units_of(j);
#endif
    const float *const act_p = a.act ? a.act : a.bias, *const res_p = a.residual ? a.residual : a.x;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#ifdef XD_STAMP
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
#endif
    struct Task {
        int tc0, s0, n_w, net, pbase, X, nKmax, gb;
        bool span, valid_w;
        xd_gptr xs, ws;                                                     // xs: this wave's sample; ws: packed weights of (net, group block), class 0
    };
    int scan_j = 0, scan_base = 0;
    auto decode = [&](int kt, Task &t) __attribute__((always_inline)) {
        const int u = kt * wgs_per_xcd + ((kt & 1) ? wgs_per_xcd - 1 - wg_in_xcd : wg_in_xcd);
        if (u >= n_my) return false;
#ifdef XQ_NETMAJOR
        while (u >= (wnet + 1) * per_net) { ++wnet; scan_j = 0; scan_base = wnet * per_net; }
#endif
        while (scan_j < a.n_gbv - 1 && u >= scan_base + units_of(scan_j)) { scan_base += units_of(scan_j); ++scan_j; }   // (u grows with kt)
        const int gb = a.gb_hi - scan_j;
        t.span = (span_mask >> scan_j) & 1u;
        const int upn = t.span ? upn_span : upn_pair;
        int rem = u - scan_base, net = 0;
#ifdef XQ_NETMAJOR
        net = wnet;
#else
        while (rem >= upn) { rem -= upn; ++net; }
#endif
        t.net = net; t.gb = gb;
        t.tc0 = gb * XD_GB; t.s0 = a.psum - t.tc0;
        const int T0 = t.span ? 0 : window_of(gb);
        // samples of this wave pair's two row halves (index inside the net's list of the XCD; a slot past the list recomputes the
        // unit's first sample and stores nothing)
        int jA, jB;
        if (t.span) { jA = jB = 2 * rem + PAIR; }
        else { jA = 4 * rem + 2 * PAIR; jB = jA + 1; }
        const bool vA = jA < spn, vB = jB < spn;
        const int j0 = t.span ? 2 * rem : 4 * rem;
        if (!vA) jA = j0;
        if (!vB) jB = vA ? jA : j0;
        const int nA = xcd + 8 * (net * spn + jA), nB = xcd + 8 * (net * spn + jB);
        t.n_w = HALF ? nB : nA;
        t.valid_w = HALF ? vB : vA;
        t.pbase = T0 + (t.span ? 32 * HALF : 0);
        t.X = t.tc0 + 4 + a.hidden + XD_C0;
        t.nKmax = ((t.X < G ? t.X : G) + 3) >> 2;
        const int iw = t.n_w < a.x_mod ? t.n_w : t.n_w % a.x_mod;
        t.xs = (xd_gptr)(a.x + (long)iw * C * SKP + (long)t.s0 * HP + XD_COL0);
        t.ws = (xd_gptr)(a.packed + (((long)net * a.ngb_all + gb) * 4 * a.NKB) * XD_STAGE_A);
        return true;
    };
    const unsigned lane16 = (unsigned)lane * 16u;
    // ---- staging (round 4, second form).  WEIGHTS: one 6 KB image per stage, shared by the workgroup, double-buffered, fetched one
    // stage ahead (they come from L2), published by the stage barrier.  ACTIVATIONS: every wave fetches ITS OWN 32 rows into a PRIVATE
    // ring of two stages -- unit u of a stage = the two diagonals (5 + u, 4 - u) of the walk 5, 4, 6, 3, ... (1 KB: lanes 0..31 the first,
    // lanes 32..63 the second diagonal; 4 channel planes x 8 row quads each) -- and refills a unit's slot right after its MFMAs with the
    // unit of the stage AFTER NEXT.  Nothing but the wave's own vmcnt orders those reads, so the activations -- which come from beyond L2
    // -- have two full stages to land and no barrier waits for them (with both operands one stage ahead in a shared image a wave waited
    // 18 % of its time for its own DMAs and 7 % at stage barriers behind the others').
    constexpr int NU = 6;                                                   // units per stage
    float *const bring = ringB + WV * (2 * NU * 256);
    const unsigned bring_lds = xd_lds_addr(bring);
    auto dma_lane_offset = [&](const Task &t) __attribute__((always_inline)) -> unsigned {        // lane l = 32 w + 8 k + m: rows pbase + 4 m .. + 3 of plane k
        const int k = (lane >> 3) & 3, m = lane & 7;
        int r = t.pbase + 4 * m;
        if (r >= H + 2) r = H - 2 > 0 ? H - 2 : 0;                           // stay inside the sample's planes (rows behind the image reach no stored output)
        return (unsigned)((4 * k * (int)SKP + r) * 4);
    };
    const unsigned hp4 = (unsigned)HP * 4u, skp4 = (unsigned)SKP * 4u, kbx = 16u * skp4;
    const unsigned cls_bytes = (unsigned)a.NKB * (XD_STAGE_A * 4);          // packed weights of one class of a (net, group block)
    // a stage = (task: 0 current / 1 next, class, K block)
    struct Stage { int sel, cls, kb; };
    // ---- MFMA-side operand addresses
    const int boffA = 4 * (8 * kl + (n16 >> 1)) + 2 * (n16 & 1), boffB = boffA + 128;   // floats inside a unit's slot: first / second diagonal
    f32x4 acc[XD_NT][2];
    struct Ops2 { float a[7]; xd_f2 ba, bb; };                              // weights of the unit's two diagonals (<= 4 + 3 tiles), its two B operands
    constexpr auto unit_da = [](int u) { return 5 + u; };
    constexpr auto unit_db = [](int u) { return u < 5 ? 4 - u : -1; };
    auto load_a = [&](auto dd, float *o, const float *sA) __attribute__((always_inline)) {
        constexpr int dc = decltype(dd)::value, T = xd_ntiles(dc), tb = xd_tbase(dc);
        const float *p = sA + tb * 64;
        if constexpr (T == 1) o[0] = p[lane];
        else if constexpr (T == 2) { const xd_f2 v = *(const xd_f2 *)(p + 2 * lane); o[0] = v.x; o[1] = v.y; }
        else if constexpr (T == 3) { const xd_f2 v = *(const xd_f2 *)(p + 2 * lane); o[0] = v.x; o[1] = v.y; o[2] = p[128 + lane]; }
        else { const f32x4 v = *(const f32x4 *)(p + 4 * lane); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
    };
    // operands of unit U of a body with N live diagonals: weights from the shared image sA, activations from this wave's ring slot
    auto load_unit = [&](auto NN, auto uu, Ops2 &o, const float *sA, const float *slot) __attribute__((always_inline)) {
        constexpr int N = decltype(NN)::value, U = decltype(uu)::value, DA = unit_da(U), DB = unit_db(U);
        if constexpr (DA < N) { load_a(IC<DA>{}, o.a, sA); o.ba = *(const xd_f2 *)(slot + boffA); }
        if constexpr (DB >= 0 && DB < N) { load_a(IC<(DB >= 0 ? DB : 0)>{}, o.a + (DA < N ? xd_ntiles(DA) : 0), sA); o.bb = *(const xd_f2 *)(slot + boffB); }
    };
    auto fma_d = [&](auto dd, auto ff, const float *w, const xd_f2 &b) __attribute__((always_inline)) {
        constexpr int dc = decltype(dd)::value, T = xd_ntiles(dc), tb = xd_tbase(dc);
        constexpr bool FIRST = decltype(ff)::value;                         // the first stage of a class pass starts every chain from C = 0
        static_for<T>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            acc[tb + t][0] = xd_mfma(w[t], b.x, FIRST ? zero4 : acc[tb + t][0]);
            acc[tb + t][1] = xd_mfma(w[t], b.y, FIRST ? zero4 : acc[tb + t][1]);
        });
    };
    auto fma_unit = [&](auto NN, auto uu, auto ff, const Ops2 &o) __attribute__((always_inline)) {
        constexpr int N = decltype(NN)::value, U = decltype(uu)::value, DA = unit_da(U), DB = unit_db(U);
        if constexpr (DA < N) fma_d(IC<DA>{}, ff, o.a, o.ba);
        if constexpr (DB >= 0 && DB < N) fma_d(IC<(DB >= 0 ? DB : 0)>{}, ff, o.a + (DA < N ? xd_ntiles(DA) : 0), o.bb);
    };
    unsigned stage = 0;                                                     // parity = the weight image and the ring half the current stage reads
    Task cur, nxt;
    bool have_next = false;
    unsigned voff = 0, voff_n = 0;
    auto task_of = [&](int sel) __attribute__((always_inline)) -> const Task & { return sel ? nxt : cur; };
    auto stage_next = [&](Stage st) __attribute__((always_inline)) {          // the stage after st (the workgroup's last stage: itself)
        const int nK = st.sel ? nxt.nKmax : cur.nKmax;
        Stage r = st;
        if (st.kb + 1 < nK) r.kb = st.kb + 1;
        else if (st.cls < 3) { r.cls = st.cls + 1; r.kb = 0; }
        else if (st.sel == 0 && have_next) { r.sel = 1; r.cls = 0; r.kb = 0; }
        r.sel = __builtin_amdgcn_readfirstlane(r.sel); r.cls = __builtin_amdgcn_readfirstlane(r.cls); r.kb = __builtin_amdgcn_readfirstlane(r.kb);
        return r;
    };
    // weight chunks of stage st into image `par`
    auto dma_weights = [&](Stage st, unsigned par) __attribute__((always_inline)) {
        const Task &t = task_of(st.sel);
        xd_gptr wk = t.ws + ((unsigned)st.cls * cls_bytes + (unsigned)st.kb * (unsigned)(XD_STAGE_A * 4));
        const unsigned dst = xd_lds_addr(ringA + (par & 1u) * XD_STAGE_A);
        static_for<NA_DMA>([&](auto jj) {
            constexpr unsigned ch = (unsigned)(WV + 4 * decltype(jj)::value) * 1024u;
            xd_dma(lane16, wk + ch, dst + ch);
        });
    };
    // unit U of stage st into this wave's ring half `par`
    auto dma_unit = [&](auto uu, Stage st, unsigned par) __attribute__((always_inline)) {
        constexpr int U = decltype(uu)::value, DA = unit_da(U), DB = unit_db(U) >= 0 ? unit_db(U) : unit_da(U);   // (the sixth unit fetches diagonal 10 twice)
        const Task &t = task_of(st.sel);
        xd_gptr xk = t.xs + (unsigned)st.kb * kbx;
        const unsigned ga = (unsigned)(st.cls + XD_C0 + 16 - DA) & 3u, gb2 = (unsigned)(st.cls + XD_C0 + 16 - DB) & 3u;   // input channel inside a group: (class - c) mod 4
        const unsigned sa = (unsigned)DA * hp4 + ga * skp4, sb = (unsigned)DB * hp4 + gb2 * skp4;
        const unsigned vo = (st.sel ? voff_n : voff) + (lane < 32 ? sa : sb);
        xd_dma(vo, xk, bring_lds + ((par & 1u) * NU + U) * 1024u);
    };
#define XQ_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    // One stage.  Issue order per wave: [weights of stage + 1] then, behind each unit's MFMAs, [unit of stage + 2]: 6 + NA_DMA per stage, so
    // "all but the 10 + NA_DMA youngest" covers every unit this stage reads, and "all but the 6 youngest" at its end the weights of the next
    auto body = [&](auto NN, auto ff, Stage st1, Stage st2) __attribute__((always_inline)) {
        constexpr int N = decltype(NN)::value;
        constexpr bool FIRST = decltype(ff)::value;
        if constexpr (FIRST && N < XD_ND) {                                 // the tiles of the diagonals this body does not walk
#pragma unroll
            for (int i = xd_tbase(N); i < XD_NT; ++i) { acc[i][0] = zero4; acc[i][1] = zero4; }
        }
        const float *sA = ringA + (stage & 1u) * XD_STAGE_A;
        const float *rb = bring + (stage & 1u) * (NU * 256);
        Ops2 ops[2];
        XQ_VMWAIT(10 + NA_DMA);
        load_unit(NN, IC<0>{}, ops[0], sA, rb);
        __builtin_amdgcn_sched_barrier(0);
        dma_weights(st1, stage + 1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NU>([&](auto uu) {
            constexpr int U = decltype(uu)::value;
            if constexpr (U + 1 < NU) {
                XQ_VMWAIT(10 + NA_DMA);
                __builtin_amdgcn_sched_barrier(0);
                load_unit(NN, IC<U + 1>{}, ops[(U + 1) & 1], sA, rb + (U + 1) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            fma_unit(NN, uu, ff, ops[U & 1]);
            __builtin_amdgcn_sched_barrier(0);
            dma_unit(uu, st2, stage);                                       // the slot just read: stage + 2 has this stage's parity
            __builtin_amdgcn_sched_barrier(0);
        });
        XD_T(1);
        XQ_VMWAIT(6);
        XD_T(7);
        __syncthreads();
        XD_T(8);
        ++stage;
    };
    if (!decode(0, cur)) return;                                            // (uniform over the workgroup)
    voff = dma_lane_offset(cur);
    have_next = decode(1, nxt);
    voff_n = have_next ? dma_lane_offset(nxt) : voff;
    {   // stages 0 and 1 of the first task: weights of stage 0, activations of both
        const Stage s0 = {0, 0, 0}, s1 = stage_next(s0);
        dma_weights(s0, 0);
        static_for<NU>([&](auto uu) { dma_unit(uu, s0, 0); });
        static_for<NU>([&](auto uu) { dma_unit(uu, s1, 1); });
        XD_WAIT0();
        __syncthreads();
    }
    for (int kt = 0;; ++kt) {
        const int tc0 = cur.tc0, s0 = cur.s0, n_w = cur.n_w, net = cur.net, pbase = cur.pbase;
        const bool span = cur.span, valid_w = cur.valid_w;
#ifndef XQ_NCLS
#define XQ_NCLS 4
#endif
        static_for<XQ_NCLS>([&](auto cc) {
            constexpr int CLS = decltype(cc)::value;
            XD_T(0);
            {
                const int X = cur.X;
                int nKmax = cur.nKmax;
                asm volatile("" : "+s"(nKmax));
                int kb = 0, D = X;
                asm volatile("" : "+s"(D));                                     // (uniform: the first block's body is chosen by a scalar branch)
                auto run = [&](auto NN, auto ff) __attribute__((always_inline)) {
                    const Stage st = {0, CLS, kb}, st1 = stage_next(st), st2 = stage_next(st1);
                    body(NN, ff, st1, st2);
                };
                // the first block starts the chains (X >= 7: eleven live diagonals, or seven for group block 0 of a hidden layer)
                if (D >= 8) run(IC<XD_ND>{}, IC<true>{});
                else run(IC<7>{}, IC<true>{});
                ++kb; D -= 4;
                for (; kb < nKmax && D >= 8; ++kb, D -= 4) run(IC<XD_ND>{}, IC<false>{});
                if (kb < nKmax && D >= 4) { run(IC<7>{}, IC<false>{}); ++kb; D -= 4; }
                if (kb < nKmax) run(IC<3>{}, IC<false>{});
            }
            // ---- halo of the two row halves of a sample (span tasks), then realignment + the reference tree inside the class
            if (span) {
                if constexpr (HALF == 1) {
                    if (n16 == 0) {
                        static_for<XD_ND>([&](auto dd) {
                            constexpr int dc = decltype(dd)::value;
                            static_for<xd_nslots(dc)>([&](auto ii) {
                                constexpr int idx = decltype(ii)::value, kh = xd_slot(dc, idx) & 7, tile = xd_tbase(dc) + idx / 4, reg = idx & 3;
                                static_for<2>([&](auto tt) {
                                    constexpr int t = decltype(tt)::value;
                                    if constexpr (xd_halo_has(kh, t, true)) {
                                        constexpr int hid = ((1 * 2 + PAIR) * XD_NHALO + xd_halo_id(dc, idx, t, true)) * 4;
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
                                        constexpr int hid = ((0 * 2 + PAIR) * XD_NHALO + xd_halo_id(dc, idx, t, false)) * 4;
                                        halo[hid + kl] = acc[tile][t][reg];
                                    }
                                });
                            });
                        });
                    }
                }
                XD_T(2);
                __syncthreads();
                XD_T(3);
            }
            const float *const halo_up = halo + ((span && HALF == 0) ? (1 * 2 + PAIR) * XD_NHALO * 4 : 2 * 2 * XD_NHALO * 4) + kl;
            const float *const halo_dn = halo + ((span && HALF == 1) ? (0 * 2 + PAIR) * XD_NHALO * 4 : 2 * 2 * XD_NHALO * 4) + kl;
            static_for<XD_GB>([&](auto qq) {
                constexpr int Q = decltype(qq)::value, CQ = (CLS + Q) & 3;
                static_for<2>([&](auto tt) {
                    constexpr int TP = decltype(tt)::value;
                    auto leaf = [&](auto ii) __attribute__((always_inline)) -> float {
                        constexpr int i = decltype(ii)::value, tap = i % 25, kh = tap / 5, kw = tap % 5, c = kh + kw - Q, dc = c + XD_C0;
                        static_assert(dc >= 0 && dc < XD_ND && ((CLS - c + 16) & 3) == i / 25, "leaf i of class CQ lies on diagonal c in channel gid");
                        constexpr int idx = xd_slot_index(dc, Q, kh), tile = xd_tbase(dc) + idx / 4, reg = idx & 3, dl = kh - 2;
                        static_assert(idx >= 0, "slot table");
                        if constexpr (dl == 0) return acc[tile][TP][reg];
                        else if constexpr (dl == 1 && TP == 0) return acc[tile][1][reg];
                        else if constexpr (dl == -1 && TP == 1) return acc[tile][0][reg];
                        else if constexpr (dl > 0) {
                            constexpr int TS = dl == 1 ? 0 : TP, hid = xd_halo_id(dc, idx, TS, true) * 4;
                            const float e = halo_up[hid];
                            return xd_row_shift<XD_ROW_SHL1>(e, acc[tile][TS][reg]);
                        } else {
                            constexpr int TS = dl == -1 ? 1 : TP, hid = xd_halo_id(dc, idx, TS, false) * 4;
                            const float e = halo_dn[hid];
                            return xd_row_shift<XD_ROW_SHR1>(e, acc[tile][TS][reg]);
                        }
                    };
                    __builtin_amdgcn_sched_barrier(0);                          // (else hipcc hoists every tree's halo reads and spills accumulators)
                    const float part = XdTree<CQ, 4>::eval(leaf);               // F(CQ, 4) of group Q
                    // class order of group Q over the passes: Q, Q+1, Q+2, Q+3 (mod 4): the second class of a parity adds to the first
                    constexpr int hi = ((Q * 2 + TP) * 2 + (CQ & 1)) * 64;
                    if constexpr (CLS < 2) hs[hi] = part;
                    else hs[hi] = hs[hi] + part;                                // F0 + F2 / F1 + F3 (IEEE addition commutes exactly)
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        XD_T(4);
        // ---- (F0 + F2) + (F1 + F3) + bias / PReLU / residual / store: this wave's two rows of channel o = lane >> 4 of the three groups
        {
            const int pe_e = pbase + 2 * n16, o = kl;
            const int pc = pe_e > ((H + 1) & ~1) ? ((H + 1) & ~1) : pe_e;
            static_for<XD_GB>([&](auto qq) {
                constexpr int Q = decltype(qq)::value;
                const int g = tc0 + Q, sq = s0 - Q;
                const bool vq = valid_w && g < G && sq >= 0 && sq < S && o < a.cout;
                const int lo = sq >= W ? sq - W + 1 : 0, hi = sq < H ? sq : H - 1;
                const bool ok0 = vq && pe_e >= lo && pe_e <= hi, ok1 = vq && pe_e + 1 >= lo && pe_e + 1 <= hi;
                if (ok0 || ok1) {
                    const int bid = net * nout + g * a.cout + o;
                    const long oi = ((long)n_w * nout + g * a.cout + o) * SKP + (long)(sq + XD_ROW0) * HP + pc + XD_COL0;
                    const float e_bias = a.bias[bid], e_act = act_p[bid];
                    xd_f2 e_res = {0.f, 0.f};
                    if (a.residual) e_res = *(const xd_f2u *)(res_p + oi);
                    float sv[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float v = (hs[((Q * 2 + t) * 2 + 0) * 64] + hs[((Q * 2 + t) * 2 + 1) * 64]) + e_bias;
                        if (a.act) { if (v < 0) v = v * e_act; }                  // cconv_dc_cuda.cu:360-362
                        if (a.residual) v = v + (t == 0 ? e_res.x : e_res.y);    // fused TileAdd
                        sv[t] = v;
                    }
                    if (ok0 && ok1) *(xd_f2u *)(a.out + oi) = (xd_f2u){sv[0], sv[1]};
                    else if (ok0) a.out[oi] = sv[0];
                    else a.out[oi + 1] = sv[1];
                }
            });
        }
        XD_T(6);
        if (!have_next) break;
        cur = nxt;
        voff = voff_n;
        have_next = decode(kt + 2, nxt);
        voff_n = have_next ? dma_lane_offset(nxt) : voff;
    }
    XD_WAIT0();                                                             // no DMA may outlive the workgroup's LDS
#ifdef XD_STAMP
    if (lane == 0) for (int i = 0; i < 10; ++i) xd_stamps[((blockIdx.x & 255) * 8 + (blockIdx.x >> 8) * 4 + WV) * 10 + i] += st[i];
#endif
}

__global__ __launch_bounds__(XQ_THREADS, 2) void k_cconv16dq(XdArgs a) {
    __shared__ __attribute__((aligned(16))) float ringA[2 * XD_STAGE_A];     // two weight images: 12 KB
    __shared__ __attribute__((aligned(16))) float ringB[4 * 2 * 6 * 256];    // per wave: two stages x six 1 KB units of its own input rows: 48 KB
    __shared__ float halo[(2 * 2 + 1) * XD_NHALO * 4];                      // [direction][wave pair][register][channel] + a block of zeros
    __shared__ float hs[4 * XD_GB * 2 * 2 * 64];                            // [wave][group][output row][class parity][lane]
    for (int i = threadIdx.x; i < XD_NHALO * 4; i += XQ_THREADS) halo[2 * 2 * XD_NHALO * 4 + i] = 0.f;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    switch (wave) {
        case 0: xq_body<0>(a, ringA, ringB, halo, hs, lane); break;
        case 1: xq_body<1>(a, ringA, ringB, halo, hs, lane); break;
        case 2: xq_body<2>(a, ringA, ringB, halo, hs, lane); break;
        default: xq_body<3>(a, ringA, ringB, halo, hs, lane); break;
    }
}

// Same contract as lic360_cconv16_dc_plane; additionally 8 | n and 16 | n / nb and x_mod == n (the samples of a task share a net and sit
// 8 apart): returns 3 when the shape does not qualify (the caller falls back to lic360_cconv16_dc_plane).
LIC360_API int lic360_cconv16_dq_plane(void *stream, const lic360_conv_plan *p, const float *x, const float *packed, const float *bias,
                                       const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod) {
    ARG_CHECK(p && conv16dc_ok(p) && x && packed && bias && out && n > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n && h > 0 && h <= 64 && w > 0);
    if (n % 8 || (n / nb) % 16 || x_mod != n) return 3;
    const int G = p->ngroup, S = h + w - 1;
    if (psum < 0 || psum >= h + w + G - 2) return 0;
    int rows, pitch, row0, col0;
    if (lic360_dc4_layout(h, w, &rows, &pitch, &row0, &col0)) return 2;
    ARG_CHECK(row0 == XD_ROW0 && col0 == XD_COL0);
    ARG_CHECK(8L * G * 4 * rows * pitch * 4 < (1L << 32));
    XdArgs a;
    a.x = x; a.packed = packed; a.bias = bias; a.act = act; a.residual = residual; a.out = out;
    a.G = G; a.cout = p->cout; a.hidden = p->constrain == 5 ? 0 : 1; a.H = h; a.W = w; a.npb = n / nb; a.x_mod = x_mod; a.N = n; a.psum = psum;
    a.ngb_all = conv16dc_ngb(p); a.NKB = conv16dc_nkb(p); a.HP = pitch; a.SKP = (long)rows * pitch;
    a.can_pair = 1; a.rs = 0;
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
    static const int grid = [] { const char *e = getenv("LIC360_DQ_GRID"); const int g = e ? atoi(e) : 0; return g >= 8 && g <= 2048 ? g : 512; }();   // (diagnostic: 256 = one workgroup per CU)
    hipLaunchKernelGGL(k_cconv16dq, dim3(grid), dim3(XQ_THREADS), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return 0;
}
