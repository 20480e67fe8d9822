// cconv16_kernels.hip -- ENCODE-order group-causal masked convolution of the latent entropy nets (A9 for the shapes
// test/lic360_demo.py:104-112 uses: ngroup groups, cin in {1,4}, cout in {3,4}) on v_mfma_f32_16x16x4_f32.
//
// Arithmetic contract (extension/cconv_ec_cuda.cu:268-315): per output scalar 128 virtual lanes, lane
// (gid, kh, kw) runs ONE fmaf chain over ti = gid, gid+cin, ... (= input groups tc = 0, 1, ...), then the fixed tree
// p[i]+p[i+64]; +32; ...; +1.  v_mfma_f32_16x16x4_f32 is, bit for bit, a k-ordered fmaf chain of 4 terms
// (D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))), so with K = four CONSECUTIVE input groups of one lane's chain an
// MFMA advances 256 chains by four terms each, in order -- at the full fp32 matrix rate (the 4x4x1 form used in decode
// order tops out at 74 % of it and blocks the VALU while it issues).
//
// Mapping ("leaf-resident" like cconv4_kernels.hip: every chain of every output owns an accumulator until the tree):
//   * MFMA rows  = 16 outputs that read the SAME input at the SAME tap: 4 consecutive groups g0..g0+3 x their 4 output
//     channels.  In encode order all groups are evaluated at the same positions, so the B operand is shared exactly; the
//     groups' chains differ in length by one each (L = g+4+hidden-kh-kw) -- the packed A operand carries zero weights
//     past a chain's end (fma(0, x, acc) == acc).
//   * MFMA cols  = 16 consecutive positions of one image row.
//   * K          = input groups tc0..tc0+3 of the chain's channel gid (cin = 4), or tc0+4sq..+3 (cin = 1).
//   * a workgroup (8 waves, two per SIMD) owns a 4-row x 16-column tile of one (sample, group block): wave (ps, c) keeps the
//     chains of lane class c (lanes = c mod 4: 25 chains for cin = 4, 7/6 for cin = 1) of rows 2ps, 2ps+1 resident: 2 x 25
//     accumulators of 4 registers = 200 of the wave's 256.  Lanes of equal index mod 4 stay together until the last two
//     tree levels, so each wave reduces in registers and one tile per wave and row crosses LDS for the final (F0+F2)+(F1+F3).
//     (One wave per SIMD with all four rows -- 400 accumulator registers -- was tried first: hipcc cannot keep more
//     accumulators than one half of the unified register file holds without copying or spilling them around every MFMA.)
//   * a STEP = 16 input channel planes (cin = 4: four input groups; cin = 1: sixteen): their 8 x 20 halo tiles (10 KB) and
//     the step's packed weights (4 classes x 28 slots x 64 lanes, 28 KB) go to a double-buffered LDS image by LDS-DMA
//     (global_load_lds_dwordx4, 5 per wave and step, no staging registers: the accumulators leave none to spare)
//     two steps ahead of the MFMAs (three LDS buffers); one barrier per step; per wave and step 25 weight reads + 50 operand reads
//     (ds_read_b32, immediate offsets only) feed 50 MFMAs of 32 cycles.
//   * workgroups are persistent; tasks = (sample, 4 consecutive tiles, group block), group block fastest, pulled from a
//     per-XCD counter so that the workgroups of an XCD sweep one input region together (it stays in that XCD's L2) and the
//     group blocks' 6x different lengths balance; the staging pipeline runs across tile and task boundaries.
// Activations: zero-haloed NCHW planes [n][c][hp][wp], cell (r, c) at [(r+2)*wp + c+2] (lic360_ec16_layout); rows / columns
// beyond the image are never written and stay zero, so no load is conditional.
#include "common.h"
#include "conv_plan.h"
#include "cconv_tree.h"
#include "gmm_tables.h"
#include "need.h"

#define C16_SLOTS 28                       // weight slots per lane class and step (cin = 4: 25 taps; cin = 1: 4 sub-quads x 7 taps)
#define C16_TH 4                           // tile rows
#define C16_TW 16                          // tile columns = MFMA columns
#define C16_NT 2                           // tile rows per wave
#define C16_HR (C16_TH + 4)
#define C16_HC (C16_TW + 4)
#define C16_PLANE 164                      // floats per staged plane: 8 x 20 + one pad quad, so that 4 planes = 16 banks (mod 32)
#define C16_WFL (4 * C16_SLOTS * 64)       // floats of packed weights per step (28 KB)
#define C16_XWIN 11                        // 1 KB LDS-DMA windows of x per step: 11 x 64 quads >= 16 planes x 41 quads
#define C16_WWIN (C16_WFL / 256)           // ... of weights (28)
#define C16_XFL (C16_XWIN * 256)
#define C16_NDMA 5                         // DMAs per wave and step: 8 x 5 = 40 windows = 11 + 28 + 1 dump
#define C16_BUF (C16_XFL + C16_WFL + 256)  // floats per LDS buffer: x | weights | dump window
#define C16_THREADS 512
#define C16_COMB (C16_TH * 4 * 4 * 64)     // tile rows x classes x registers x lanes
#define C16_TPT 4                          // tiles per task
#define C16_NBUF 3                         // LDS step buffers: the DMAs run two steps ahead of the MFMAs
#define C16_FGPB 5                         // groups per block of the FUSED last layer (cout = 3: row = 3 q + r)
#define C16_FTPT 2                         // tiles per task of the FUSED last layer: its three nets run as PHASES over them (y parked in LDS)
#define C16_YFL (C16_FTPT * 3 * C16_FGPB * 3 * 64)   // floats of the parked y: [tile][net][group][channel][position]

static inline bool conv16_ok(const lic360_conv_plan *p) {
    return p->ksz == 5 && (p->cin == 1 || p->cin == 4) && p->cout >= 1 && p->cout <= 4 && p->ngroup >= 1 && p->ngroup <= 256;
}
static inline int conv16_tcs(int cin) { return cin == 4 ? 4 : 16; }                       // input groups per step
static inline int conv16_nsteps_max(const lic360_conv_plan *p) { return (p->ngroup + conv16_tcs(p->cin) - 1) / conv16_tcs(p->cin); }
static inline int conv16_ngb(const lic360_conv_plan *p) { return (p->ngroup + 3) / 4; }

// ------------------------------------------------------------------------------------------------ weight packing
// packed16[net][gb][step][class c][slot][lane l]:  lane l = 16 k + i carries A[row i][k] of the slot's MFMA:
//   row i = 4 q + r  (group g0 + q, output channel r),  k = input group offset inside the quad;
//   cin = 4: slot = tap, gid = (c - tap) mod 4, input channel (tc0 + k) * 4 + gid;
//   cin = 1: slot = 7 sq + j, tap = c + 4 j, input channel tc0 + 4 sq + k.
// Zero where the chain has ended (tc >= L = g + 4 + hidden - kh - kw, capped at ngroup), for r >= cout, g >= ngroup, unused slots.
// FUSED last layer (cout = 3, lic360_conv16_pack_tables): FIVE groups per block, row i = 3 q + r (15 of the 16 MFMA rows carry an output instead
// of 12: 10 group blocks instead of 12 for 48 groups).
__global__ void k_conv16_pack(const float *__restrict__ weight, float *__restrict__ packed, int nb, int G, int cin, int cout, int hidden, int NS, int gpb, int rpg) {
    const int n_gb = (G + gpb - 1) / gpb, tcs = cin == 4 ? 4 : 16;
    const long per_net = (long)n_gb * NS * C16_WFL, total = per_net * nb;
    const int C = G * cin, nout = G * cout;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        long t = e >> 6;
        const int slot = (int)(t % C16_SLOTS); t /= C16_SLOTS;
        const int c = (int)(t & 3); t >>= 2;
        const int step = (int)(t % NS); t /= NS;
        const int gb = (int)(t % n_gb), b = (int)(t / n_gb);
        const int i = l & 15, k = l >> 4, q = i / rpg, r = i - q * rpg, g = gb * gpb + q;
        int tap, tc, ci;
        if (cin == 4) { tap = slot; tc = step * tcs + k; ci = tc * 4 + ((c - tap) & 3); }
        else { const int sq = slot / 7, j = slot % 7; tap = c + 4 * j; tc = step * tcs + 4 * sq + k; ci = tc; }
        float v = 0.0f;
        if (tap < 25 && slot < (cin == 4 ? 25 : 28) && q < gpb && g < G && r < cout) {
            const int kh = tap / 5, kw = tap % 5;
            int L = g + 4 - kh - kw + hidden;                               // extension/cconv_ec_cuda.cu:288-290
            if (L > G) L = G;
            if (tc < L) v = weight[(((long)b * nout + g * cout + r) * C + ci) * 25 + tap];
        }
        packed[e] = v;
    }
}

LIC360_API int lic360_conv16_supported(const lic360_conv_plan *p) { return p && conv16_ok(p) ? 1 : 0; }
LIC360_API long lic360_conv16_packed_floats(const lic360_conv_plan *p) {
    return p && conv16_ok(p) ? (long)conv16_ngb(p) * conv16_nsteps_max(p) * C16_WFL : 0;
}
LIC360_API int lic360_conv16_pack(void *stream, const lic360_conv_plan *p, const float *weight, int nb, float *packed) {
    ARG_CHECK(p && conv16_ok(p) && weight && packed && nb > 0);
    const long total = lic360_conv16_packed_floats(p) * nb;
    hipLaunchKernelGGL(k_conv16_pack, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, weight, packed, nb, p->ngroup, p->cin,
                       p->cout, p->constrain == 5 ? 0 : 1, conv16_nsteps_max(p), 4, 4);
    LAUNCH_CHECK();
    return 0;
}
// the packing lic360_cconv16_ec_tables reads (cin = 4, cout = 3: five groups per block); fits the lic360_conv16_packed_floats allocation
LIC360_API int lic360_conv16_pack_tables(void *stream, const lic360_conv_plan *p, const float *weight, int nb, float *packed) {
    ARG_CHECK(p && conv16_ok(p) && p->cin == 4 && p->cout == 3 && weight && packed && nb > 0);
    const long total = (long)((p->ngroup + C16_FGPB - 1) / C16_FGPB) * conv16_nsteps_max(p) * C16_WFL * nb;
    hipLaunchKernelGGL(k_conv16_pack, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, weight, packed, nb, p->ngroup, p->cin,
                       p->cout, p->constrain == 5 ? 0 : 1, conv16_nsteps_max(p), C16_FGPB, 3);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_ec16_layout(int h, int w, int *hp, int *wp) {
    ARG_CHECK(hp && wp && h > 0 && w > 0);
    *hp = (h + C16_TH - 1) / C16_TH * C16_TH + 4;
    *wp = (w + C16_TW - 1) / C16_TW * C16_TW + 4;
    return 0;
}

// ------------------------------------------------------------------------------------------------ kernel
// LDS-DMA through inline asm (the idiom of cconv4v3_dc.inc): destination = M0 + lane * 16; the compiler does not see these
// VMEM operations, completion is enforced by hand (C16_WAIT0 before the barrier that publishes the buffer).
#define C16_WAIT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define C16_STR2(x) #x
#define C16_STR(x) C16_STR2(x)
#define C16_WAIT_PREV() asm volatile("s_waitcnt vmcnt(" C16_STR(C16_NDMA) ")" ::: "memory")   // all but the youngest step's DMAs
#define C16_NDMA4 10                       // cin = 4: the four waves of set 0 issue the whole step image, 10 windows each
#define C16_WAIT_PREV4() asm volatile("s_waitcnt vmcnt(" C16_STR(C16_NDMA4) ")" ::: "memory")
// KIND: 0 = a window of x halo tiles, 1 = a window of packed weights
template <int KIND = 1>
__device__ __forceinline__ void c16_dma_x4(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory");
}
// the form the kernels use: 64-bit UNIFORM base (SGPR pair) + 32-bit per-lane byte offset.  The per-lane offsets are loop-invariant registers and a
// window's position goes into the scalar base, so a DMA costs scalar instructions only: with per-lane 64-bit pointers every DMA carried 2-8
// VALU instructions (two quarter-rate multiplies for the x windows) into the MFMA stream -- stamps with the DMA instruction removed but its
// address arithmetic kept ran as slowly as the full kernel (tools/ec_stamp.sh -DC16_EXP_DMAADDR).
template <int KIND = 1>
__device__ __forceinline__ void c16_dma_s(unsigned voff, const float *sbase, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ const float *c16_uniform(const float *p) {     // the value IS wave-uniform; this tells the compiler
    const unsigned long v = (unsigned long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const float *)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned c16_lds_addr(const float *p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const float *)p;
}

struct C16Args {
    const float *x, *packed, *bias, *act, *residual;
    float *out;
    int *ctr;                                  // 8 task counters (one per XCD), zeroed by the host before the launch
    int G, cout, hidden, H, W, hp, wp, npb, x_mod, N;
    int n_gb, ntx, ntiles, n_chunks, NS;
    int gbk;                                   // task order: blocks of gbk (sample, chunk) units, group block slowest inside a block
    // FUSE (last layer of the latent net + CDF-table build, SURVEY.md §7 k_cconv_ec_last_gmm): N = images, the three stacked
    // nets [weight, sigma, mu] of an image are swept one after the other inside a task
    const float *code, *mask;                  // [N, G, H, W] symbols / importance mask
    const int *pidx, *plane_start;             // scan-order prefix tables (code_contex_cuda.cu:19-31) / first record of a plane
    uint2 *rec;                                // [N][G*H*W] (cdf[sym], cdf[sym+1]) in coding order
    // dead-cone skip (round 6, need.h): per XCD the live tasks of this layer in launch order, entry = u | tile mask << 28 (list[xcd * list_cap + k], cnt[xcd]
    // entries); NULL: every task, every tile
    const int *list, *cnt;
    int list_cap;
};
#define C16_TASK_END 0x0fffffff              // task number of a ring entry past the end of the XCD's list

#ifdef C16_STAMP
// diagnostic build only (tools/ec_stamp.sh): cycles per phase of the hidden-layer instantiation, summed per wave over the launches
__device__ unsigned long long c16_stamps[256 * 8 * 10];
#define C16_T(i) do { if constexpr (CIN == 4 && !FUSE) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[i] += t_ - t0; t0 = t_; } } while (0)
LIC360_API int lic360_c16_stamps(unsigned long long *host_out, int clear) {
    if (host_out) HIP_TRY(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(c16_stamps), sizeof(c16_stamps)));
    if (clear) { static unsigned long long z[256 * 8 * 10]; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c16_stamps), z, sizeof(z))); }
    return 0;
}
#else
#define C16_T(i)
#endif
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// taps of the three tap-diagonal ranges (d = kh + kw <= 3, 4..5, 6..8), in tap order
__host__ __device__ constexpr int c16_range_count(int R) { return R == 0 ? 10 : (R == 1 ? 9 : 6); }
__host__ __device__ constexpr int c16_range_tap(int R, int i) {
    int n = 0;
    for (int tap = 0; tap < 25; ++tap) {
        const int d = tap / 5 + tap % 5, r = d <= 3 ? 0 : (d <= 5 ? 1 : 2);
        if (r == R) { if (n == i) return tap; ++n; }
    }
    return -1;
}
// operands of one chain (cin = 4): its weight register and the operand registers of the wave's two tile rows
struct C16Ops { float a, b[C16_NT]; };
template <int CLS, int TAP>
__device__ __forceinline__ void c16_load4(C16Ops &o, const float *xs, const float *ws) {
    constexpr int kh = TAP / 5, kw = TAP % 5, gid = (CLS - TAP) & 3;
    o.a = ws[TAP * 64];
#pragma unroll
    for (int t = 0; t < C16_NT; ++t) o.b[t] = xs[gid * C16_PLANE + (kh + t) * C16_HC + kw];
}
// cin = 4, one range of one step: per chain 1 weight read + 2 operand reads + 2 MFMAs, software-pipelined C16_PF chains deep
// (the reads of chain g + C16_PF are issued before the MFMAs of chain g; the two waves of a SIMD run this code in lockstep, so
// a partner's MFMAs do not cover a wave's LDS latency); sched_barriers pin that order -- left alone, the scheduler hoists every
// read of the range above the first MFMA and spills the accumulators.  ops[] is a ring indexed by the chain's position g in
// the step (ranges back to back); chains of the NEXT range are prefetched when that range runs (`more`).
#ifndef C16_PF
#define C16_PF 2
#endif
__host__ __device__ constexpr int c16_range_base(int R) { return R == 0 ? 0 : (R == 1 ? 10 : 19); }
template <int CLS, int R, bool FIRST, class Hook>
__device__ __forceinline__ void c16_range4(f32x4 (*acc)[25], const float *xs, const float *ws, C16Ops (&ops)[C16_PF + 1], bool more, Hook &&hook) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};                                 // FIRST: the first step of a tile starts every chain from C = 0
    constexpr int N = c16_range_count(R), G0 = c16_range_base(R);
    static_for<N>([&](auto ii) {
        constexpr int i = decltype(ii)::value, tap = c16_range_tap(R, i), g = G0 + i, slot = g % (C16_PF + 1), nslot = (g + C16_PF) % (C16_PF + 1);
        if constexpr (i + C16_PF < N) c16_load4<CLS, c16_range_tap(R, i + C16_PF)>(ops[nslot], xs, ws);
        else if constexpr (R < 2) { if (more) c16_load4<CLS, c16_range_tap(R + 1, i + C16_PF - N)>(ops[nslot], xs, ws); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < C16_NT; ++t) acc[t][tap] = mfma16(ops[slot].a, ops[slot].b[t], FIRST ? zero4 : acc[t][tap]);
        __builtin_amdgcn_sched_barrier(0);
        hook(IC<i>{});                                                      // (range 0: the step's LDS-DMAs, spread between the chains)
    });
}

// FUSE: table build of a finished task -- all three nets' outputs of the task's tiles are in LDS (comb + C16_COMB): a thread per coded symbol builds
// the 9-entry CDF (softmax, sigma floor, erf CDF, fix-up: entropy_gmm_table_cuda.cu:29-107,138-159) and writes the symbol's (cdf[sym], cdf[sym+1])
// record at its place in coding order.  Two passes over the task's (tile, group, position) symbols: masked ones get their (0, 0) record at
// once, the coded ones are COMPACTED into a list first (the partial-sum half of comb is dead here), so that the CDF arithmetic -- 9 % of the
// kernel when every thread that owned a symbol ran it, masked or not, in lockstep with its wave -- runs on dense waves.
// NOT inlined: one copy of its ~10 KB of code in the kernel instead of one per call site and lane class (instruction cache, see c16_body).
typedef __attribute__((address_space(3))) float c16_lds_f;
typedef __attribute__((address_space(3))) int c16_lds_i;
template <int GPB, int TPT>
__device__ __attribute__((noinline)) void c16_tables_phase(c16_lds_f *comb, int tid, int tb_T, int tb_nt, int tb_n, int tb_gb, const float *__restrict__ mask,
                                                           const float *__restrict__ code, uint2 *__restrict__ rec, const int *__restrict__ pidx,
                                                           const int *__restrict__ plane_start, int ntx, int H, int W, int G) {
    c16_lds_i *const tl = (c16_lds_i *)comb;                               // [0, 64 GPB TPT): item list, then the counter
    c16_lds_i *const tcnt = tl + 64 * GPB * TPT;
    __syncthreads();                                                        // the last tile's y is complete, its partial sums have been read
    if (tid == 0) *tcnt = 0;
    __syncthreads();
    const long HW = (long)H * W;
    auto locate = [&](int item, int &g, int &th, int &tw, int &pos, int &tt, int &q) __attribute__((always_inline)) {
        tt = item / (64 * GPB);
        const int w = item - tt * (64 * GPB);
        q = w >> 6; pos = w & 63;
        const int T = tb_T + tt, ty = T / ntx, tx = T - ty * ntx;
        th = ty * C16_TH + (pos >> 4); tw = tx * C16_TW + (pos & 15); g = tb_gb * GPB + q;
    };
    auto record_at = [&](int g, int th, int tw) __attribute__((always_inline)) -> long {      // the symbol's place in coding order (tile_extract_cuda.cu:36-41)
        const int sd = th + tw, p = sd + g, la = p >= G ? p - G + 1 : 0;
        return (long)tb_n * G * HW + plane_start[p] + (pidx[sd] - pidx[la]) + (th - (sd >= W ? sd - W + 1 : 0));
    };
    for (int item = tid; item < tb_nt * 64 * GPB; item += C16_THREADS) {
        int g, th, tw, pos, tt, q;
        locate(item, g, th, tw, pos, tt, q);
        if (g < G && th < H && tw < W) {
            if (mask[(((long)tb_n * G + g) * H + th) * W + tw] < 0.5f) rec[record_at(g, th, tw)] = make_uint2(0u, 0u);   // coder.cpp:79
            else tl[__hip_atomic_fetch_add(tcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] = item;
        }
    }
    __syncthreads();
    const int ncoded = *tcnt;
    for (int t = tid; t < ncoded; t += C16_THREADS) {
        int g, th, tw, pos, tt, q;
        locate(tl[t], g, th, tw, pos, tt, q);
        float v[9];
#pragma unroll
        for (int net = 0; net < 3; ++net)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[net * 3 + c] = comb[C16_COMB + (((tt * 3 + net) * GPB + q) * 3 + c) * 64 + pos];
        int Tb[9];
        gmm_cdf9(v, v + 3, v + 6, Tb);                                      // softmax, sigma floor, erf CDF, fix-up: entropy_gmm_table_cuda.cu:29-107,138-159
        int sym = (int)code[(((long)tb_n * G + g) * H + th) * W + tw];
        sym = sym < 0 ? 0 : (sym > 7 ? 7 : sym);
        rec[record_at(g, th, tw)] = make_uint2((unsigned)Tb[sym], (unsigned)Tb[sym + 1]);
    }
    __syncthreads();                                                        // (the list lies where the next tile's partial sums go)
}

// PS (the wave's pair of tile rows): compile-time (PS_T >= 0) in the unfused kernels, a RUN-TIME value in the fused one (four copies of this
// code instead of eight).  With eight copies a kernel is 72-96 KB of code, every wave looping over ~12 KB of its own -- more than the 64 KB
// instruction cache two CUs share: SQC_ICACHE_MISSES 4.2 M + 3.8 M duplicates per hidden-layer launch, ~4.6 per wave and tile, all in the
// once-per-tile code (tree, epilogue, task set-up); 11 M + 4.6 M for the fused kernel, which was 250 KB with the table phase inlined at two
// sites of eight copies.  Measured (round 4, same box): the run-time form removes the misses (14 K per launch) and costs the hidden layer
// +0.7 % and the first layer +4 % (their DMA window selection becomes scalar branches), so only the fused kernel keeps it -- there, with the
// table phase as ONE noinline function (63 KB in all), it is part of a 6 % gain (see DMA0 below).
template <int CIN, int CLS, bool FUSE, int PS_T = -1>
__device__ __forceinline__ void c16_body(const C16Args &a, float *lds, float *comb, int *tq, const int tid, const int lane, const int ps_rt) {
    const int PS = PS_T >= 0 ? PS_T : ps_rt;
    static_assert(!FUSE || CIN == 4, "the fused table build belongs to the last (cin = 4) layer");
    constexpr int NSUB = FUSE ? 3 : 1;                                      // FUSE: the 3 stacked nets of an image, one after the other per tile
    constexpr int GPB = FUSE ? C16_FGPB : 4;                                // groups per block (MFMA row = 4 q + r, fused: 3 q + r)
    // tiles per task.  FUSE: the task's tiles are swept once per net (net 0 over all of them, then net 1, then net 2), so that the
    // workgroups of an XCD -- which start together and run equal task shapes -- stream ONE net's weights at a time (cycling the nets per
    // tile tripled the weight working set: 33 GB of fabric traffic per launch for 1.2 GB of algorithmic bytes); the nets' outputs wait
    // in LDS for the table phase
    constexpr int TPT = FUSE ? C16_FTPT : C16_TPT;
    constexpr int NA = NAcc<CIN>::value;
    constexpr int TCS = CIN == 4 ? 4 : 16;                                  // input groups per step
    const int WAVE = PS * 4 + CLS;
    const int G = a.G, C = G * CIN, nout = G * a.cout;
    const long PL = (long)a.hp * a.wp;
    const int xcd = blockIdx.x & 7;
    const int ns_x = (a.N - xcd + 7) >> 3;                                  // samples of this XCD: n = xcd + 8 m
    const int n_my = ns_x * a.n_chunks * a.n_gb;
#ifdef C16_STAMP
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t_entry = t0;
#endif
    // ---- task queue: thread 0 pulls task numbers four tasks ahead of the compute cursor into an 8-slot LDS ring
    // (the issue cursor runs two steps ahead: with one-step tasks it reads task c + 3 while the compute cursor is in task c,
    // and a barrier must lie between a pull and its first read)
    // ring entry = task number u | tile mask << 28 (bit t: tile t of the task's chunk is live).  With a list (a.list: the dead-cone skip, need.h) the
    // counter indexes the XCD's compacted list of live tasks; without one every task is live with all its tiles.
    const int n_live = a.list ? __builtin_amdgcn_readfirstlane(a.cnt[xcd]) : n_my;   // entries of this XCD's list
    auto pull = [&](int k) __attribute__((always_inline)) {
        if (tid == 0) {
            const int i = atomicAdd(a.ctr + xcd, 1);
            tq[k & 7] = i < n_live ? (a.list ? a.list[(long)xcd * a.list_cap + i] : (int)((unsigned)i | 0xfu << 28)) : C16_TASK_END;
        }
    };
    auto task = [&](int k) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(tq[k & 7]) & 0x0fffffff; };
    // the live tiles of ring entry k (clamped to the tiles that exist: tile0 = the chunk's first tile)
    auto tmask = [&](int k, int tile0) __attribute__((always_inline)) {
        const int nt = a.ntiles - tile0 < TPT ? a.ntiles - tile0 : TPT;
        return (int)(((unsigned)__builtin_amdgcn_readfirstlane(tq[k & 7]) >> 28) & ((1u << nt) - 1u));
    };
    // task u -> sample n, first tile, group block (group block fastest: the workgroups of an XCD share the input region)
    auto decode = [&](int u, int &n, int &tile0, int &gb) __attribute__((always_inline)) {
        const int units = ns_x * a.n_chunks, per = a.gbk * a.n_gb, blk = u / per, r = u - blk * per;
        const int left = units - blk * a.gbk, kk = left < a.gbk ? left : a.gbk;
        gb = a.n_gb - 1 - r / kk;
        const int v = blk * a.gbk + r % kk;
        tile0 = (v % a.n_chunks) * TPT;
        n = xcd + 8 * (v / a.n_chunks);
    };
    auto steps_of = [&](int gb) __attribute__((always_inline)) {
        int gl = gb * GPB + GPB - 1;
        if (gl > G - 1) gl = G - 1;
        int L = gl + 4 + a.hidden;
        if (L > G) L = G;
        return (L + TCS - 1) / TCS;
    };
    pull(0);
    pull(1);
    pull(2);
    pull(3);
    pull(4);
    __syncthreads();
    // ---- LDS-DMA windows of this wave: window j = WAVE + 8 m (m < 5) of the step image [11 x | 28 weights | 1 dump]; lane l
    // of an x window moves quad e = 64 j + l of [16 slots][41 quads] (quad 40 of a plane is padding, quads >= 656 are slack:
    // both re-fetch a valid quad); a weight window moves 1 KB of the step's packed weights
    constexpr int QPR = C16_HC / 4, QPP = C16_PLANE / 4;                    // quads per row / per plane incl. pad (5, 41)
    static_assert(C16_XWIN * 64 >= 16 * QPP && C16_XWIN + C16_WWIN + 1 == 8 * C16_NDMA, "DMA windows cover the step image");
    // DMA0 (fused kernel): ONLY the waves of set 0 issue DMAs -- window j = CLS + 4 m, m < 10; set 1 runs no issue cursor.  Set 0 (older waves: they
    // win the SIMD's issue arbitration) reaches the step barrier ~1000 cycles before set 1 (tools/ec_stamp.sh), so the DMA issue costs the
    // workgroup less there.  Where the 40 DMAs of a step are issued was measured per kernel (tools/ec_variants.sh, same box; ms per 48 images):
    //   five per wave in both sets, between the chains of range 0 (round 2/3):   hidden 12.37   fused 12.05
    //   ten per wave of set 0, behind the step's MFMAs (barrier wait):           hidden 12.42   fused 11.80
    //   ten per wave of set 0, one behind each chain of range 0 (SPREAD):        hidden 12.97   fused 11.32
    // (the cost of the DMAs -- ~1 ms per launch, NODMA ablation -- moves between the sets but is conserved in the hidden layers).
    constexpr bool DMA0 = FUSE;
#ifdef C16_DMA_SPREAD
    constexpr bool SPREAD = DMA0 && (C16_DMA_SPREAD != 0);
#else
    constexpr bool SPREAD = DMA0;
#endif
    constexpr int NDW = DMA0 ? C16_NDMA4 : C16_NDMA;                        // DMAs per issuing wave and step
    const bool issuer = !DMA0 || PS == 0;
    constexpr int NXD = DMA0 ? 3 : (PS_T >= 0 ? (C16_XWIN - (PS_T * 4 + CLS) + 7) / 8 : 2);   // x windows of a wave (j < 11)
    auto window = [&](int m) __attribute__((always_inline)) { return DMA0 ? CLS + 4 * m : WAVE + 8 * m; };
    int xpl[NXD], xg[NXD];
    unsigned xoff[NXD];
#pragma unroll
    for (int m = 0; m < NXD; ++m) {
        int e = window(m) * 64 + lane;
        if (e >= 16 * QPP) e = 0;
        const int slot = e / QPP;
        int rem = e - slot * QPP;
        if (rem >= C16_HR * QPR) rem = C16_HR * QPR - 1;
        const int row = rem / QPR, cq = rem - row * QPR;
        xpl[m] = CIN == 4 ? slot : ((slot & 3) * 4 + (slot >> 2));          // cin = 1: slot 4 k + sq holds channel tc0 + 4 sq + k
        xg[m] = row * a.wp + cq * 4;
        xoff[m] = (unsigned)(((long)xpl[m] * PL + xg[m]) * 4);              // byte offset from the step's first plane at the tile origin
    }
    const unsigned woff = lane * 16;
    const bool ragged = C % (TCS * CIN) != 0;                               // the last step holds fewer than 16 planes
    const unsigned lds_base = c16_lds_addr(lds);
    // ---- issue cursor
    int iq = 0, i_rem = 0, i_mask0 = 0, istep = 0, inet = 0, i_nsteps = 0, i_tile0 = 0, i_n = 0, i_gb = 0;   // i_rem: live tiles of the task still to issue
    bool ivalid = false;
    const float *ixb = a.x, *iwb = a.packed;                                // x of (sample, tile), weights of (net, group block)
    auto issue_task = [&]() __attribute__((always_inline)) {
        const int u = task(iq);
        ivalid = u < n_my;
        if (ivalid) {
            decode(u, i_n, i_tile0, i_gb);
            i_mask0 = i_rem = tmask(iq, i_tile0);
            i_nsteps = steps_of(i_gb);
            if constexpr (!FUSE) iwb = a.packed + ((long)(i_n / a.npb) * a.n_gb + i_gb) * a.NS * C16_WFL;
        }
    };
    auto issue_tile = [&]() __attribute__((always_inline)) {
        const int T = i_tile0 + __builtin_ctz(i_rem), ty = T / a.ntx, tx = T - ty * a.ntx;
        const int sample = FUSE ? inet * a.N + i_n : i_n;                   // FUSE: sample of net `inet` = inet * images + image
        ixb = a.x + (long)(sample % a.x_mod) * C * PL + (long)(ty * C16_TH) * a.wp + tx * C16_TW;
        if constexpr (FUSE) iwb = a.packed + ((long)inet * a.n_gb + i_gb) * a.NS * C16_WFL;
    };
    // LDS-DMA m (of 5) of the issue cursor's step into buffer `buf`; issue_advance() then moves the cursor (an exhausted cursor
    // keeps re-reading its last addresses: every DMA stays unconditional and inside the tensors, 5 per wave and step)
    auto issue_dma = [&](auto mm, int buf) __attribute__((always_inline)) {
        constexpr int m = decltype(mm)::value;
        const int j = window(m);
        const unsigned lb = lds_base + (unsigned)buf * (C16_BUF * 4);
        if constexpr (!DMA0) {
            // per-lane 64-bit addresses.  (The scalar-base form below was measured here too: hidden layers 12.76 against 12.40 ms -- the
            // base arithmetic of 5 DMAs per wave lands in the scalar unit between the chains of range 0 and spills more SGPRs.)
            const float *wsrc = iwb + (long)istep * C16_WFL + lane * 4;
            if (m < NXD && j < C16_XWIN) {
                constexpr int mx = m < NXD ? m : 0;
                int c = istep * (TCS * CIN) + xpl[mx];
                if (c > C - 1) c = C - 1;                                   // planes past the last channel only meet zero weights
                c16_dma_x4<0>(ixb + (long)c * PL + xg[mx], lb + j * 1024);
            } else c16_dma_x4(wsrc + (j < C16_XWIN + C16_WWIN ? (j - C16_XWIN) * 256 : 0), lb + j * 1024);   // weights; the 40th window: dump
            return;
        }
        const float *wsrc = iwb + (long)istep * C16_WFL;
        if (m < NXD && j < C16_XWIN) {                                      // (uniform)
            constexpr int mx = m < NXD ? m : 0;
            const int c0 = istep * (TCS * CIN);
            unsigned vo = xoff[mx];
            if (ragged && c0 + TCS * CIN > C) {                             // planes past the last channel only meet zero weights: re-read the last one
                const int c = c0 + xpl[mx] > C - 1 ? C - 1 - c0 : xpl[mx];
                vo = (unsigned)(((long)c * PL + xg[mx]) * 4);
            }
            c16_dma_s<0>(vo, c16_uniform(ixb + (long)c0 * PL), lb + j * 1024);
        } else c16_dma_s(woff, c16_uniform(wsrc + (j < C16_XWIN + C16_WWIN ? (j - C16_XWIN) * 256 : 0)), lb + j * 1024);   // weights; the 40th window: dump
    };
    auto issue_advance = [&]() __attribute__((always_inline)) {
        if (ivalid) {
            if (++istep == i_nsteps) {
                istep = 0;
                i_rem &= i_rem - 1;
                if constexpr (FUSE) {                                       // tile fastest, then net, then the next task
                    if (!i_rem) {
                        i_rem = i_mask0;
                        if (++inet == NSUB) {
                            inet = 0;
                            ++iq;
                            issue_task();
                        }
                    }
                } else {
                    if (!i_rem) {
                        ++iq;
                        issue_task();
                    }
                }
                if (ivalid) issue_tile();
                else { istep = i_nsteps - 1; inet = NSUB - 1; }             // idle: stay on the last step
            }
        }
    };
    auto issue = [&](int buf) __attribute__((always_inline)) {
        static_for<NDW>([&](auto mm) { issue_dma(mm, buf); });
        issue_advance();
    };
    if (task(0) >= n_my) return;                                            // uniform: the whole workgroup has no task
    if (issuer) {
        issue_task();
        issue_tile();
        issue(0);
        issue(1);
    }
    // ---- compute cursor
    int cq = 0, ctile = 0, cstep = 0, cnet = 0, c_nsteps, c_tile0, c_n, c_gb;   // ctile: live tiles of the task done (FUSE: the y slot of the current one)
    decode(task(0), c_n, c_tile0, c_gb);
    int c_mask0 = tmask(0, c_tile0), c_rem = c_mask0;                       // c_rem: live tiles of the task still to compute; the current one is its lowest bit
    c_nsteps = steps_of(c_gb);
    f32x4 acc[C16_NT][NA];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CIN == 1) {                                               // cin = 4 starts its chains with C = 0 in the first step
#pragma unroll
        for (int t = 0; t < C16_NT; ++t)
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[t][i] = zero4;
    }
    // operand addresses: lane (k = l >> 4, j = l & 15) reads plane slot 4 k + {gid | sq}, row 2 PS + t + kh, column j + kw
    const int xlane = (lane >> 4) * 4 * C16_PLANE + (C16_NT * PS) * C16_HC + (lane & 15);
    const int wlane = C16_XFL + CLS * C16_SLOTS * 64 + lane;
    if constexpr (DMA0) C16_WAIT_PREV4();
    else C16_WAIT_PREV();
    __syncthreads();
    float e_bias[2] = {0.f, 0.f}, e_act[2] = {0.f, 0.f}, e_res[2] = {0.f, 0.f};
    long e_oi[2] = {0, 0};
    bool e_ok = false;
    int e_cq = -1;
    const float *const act_p = a.act ? a.act : a.bias, *const res_p = a.residual ? a.residual : a.x;
    // FUSE: table build of a finished tile.  It runs at the top of the NEXT tile's first step -- where every accumulator is dead
    // (the first step starts all chains from C = 0), so its ~90 registers do not compete with them -- or after the loop.
    bool tb_pending = false;
    int tb_T = 0, tb_nt = 0, tb_n = 0, tb_gb = 0;                           // first tile and tile count of the finished task
    auto tables = [&]() __attribute__((always_inline)) {
        c16_tables_phase<GPB, TPT>((c16_lds_f *)comb, tid, tb_T, tb_nt, tb_n, tb_gb, a.mask, a.code, a.rec, a.pidx, a.plane_start, a.ntx, a.H, a.W, G);
        tb_pending = false;
    };
    // (comb is double-buffered by tile parity: with one-step tiles the next tile's partial sums are written before the
    // barrier that would separate them from this tile's reads)
    int cur = 0, ntile = 0;                                                 // cur: LDS buffer of the step being computed
    bool done = false;
    C16_T(8);
    do {
        const float *xs = lds + cur * C16_BUF + xlane, *ws = lds + cur * C16_BUF + wlane;
        const int nbuf = cur >= 1 ? cur - 1 : C16_NBUF - 1;                  // DMAs of the step after next go to (cur + 2) mod 3
        if constexpr (CIN == 1) issue(nbuf);
        const int g0 = c_gb * GPB, dl = g0 + GPB + 3 + a.hidden - cstep * TCS;   // tap diagonals d >= dl carry only zero weights (the block's last group has the longest chains)
        // bias / PReLU slope / residual of the tile are fetched at the START of its last step, from clamped addresses (they
        // are the oldest loads in flight when the epilogue needs them: waiting for them there stalled all eight waves for
        // three dependent memory round trips per tile, ~25 % of the kernel)
        const bool last = cstep + 1 == c_nsteps;
        if (last) {
            const int T = c_tile0 + __builtin_ctz(c_rem), ty = T / a.ntx, tx = T - ty * a.ntx;
            const int q = lane >> 4, j = lane & 15;
            const int y = ty * C16_TH + (WAVE >> 1), x = tx * C16_TW + j, g = c_gb * 4 + q;
            e_ok = g < G && y < a.H && x < a.W;
            const int gc = g < G ? g : G - 1;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * (WAVE & 1) + rr;
                if constexpr (FUSE) {                                       // MFMA row m = 4 q + r = 3 (group in block) + channel
                    const int m = 4 * q + r, gq = m / 3, ch = m - 3 * gq, gg = c_gb * GPB + gq;
                    e_bias[rr] = a.bias[cnet * nout + (gg < G ? gg : G - 1) * 3 + ch];
                } else {
                    const int rc = r < a.cout ? r : a.cout - 1;
                    const int o = gc * a.cout + rc, bid = (c_n / a.npb) * nout + o;
                    e_oi[rr] = ((long)c_n * nout + o) * PL + (long)((y < a.H ? y : a.H - 1) + 2) * a.wp + (x < a.W ? x : a.W - 1) + 2;
                    if (e_cq != cq) { e_bias[rr] = a.bias[bid]; e_act[rr] = act_p[bid]; }   // once per task: the same for its four tiles (4 of a tile's 6 epilogue loads: -1 %)
                    e_res[rr] = res_p[a.residual ? e_oi[rr] : 0];
                }
            }
            e_cq = cq;
        }
        C16_T(0);
        if constexpr (CIN == 4) {
            // chains in three ranges of tap diagonals (d <= 3, 4..5, 6..8): the later ranges die first as tc grows
            C16Ops ops[C16_PF + 1];
            static_for<C16_PF>([&](auto ii) { c16_load4<CLS, c16_range_tap(0, decltype(ii)::value)>(ops[decltype(ii)::value], xs, ws); });
            // the step's 5 DMAs are issued between the chains of range 0 (one after every second chain): issued together at the
            // top of the step they keep both waves of a SIMD off the matrix pipe for their whole issue time
            auto nohook = [](auto) __attribute__((always_inline)) {};
            // SPREAD: set 0's ten DMAs go between the ten chains of range 0 (one each) instead of behind the step's MFMAs
            auto dmahook = [&](auto ii) __attribute__((always_inline)) {
                constexpr int i = decltype(ii)::value;
                if constexpr (SPREAD && i < NDW) { if (PS == 0) issue_dma(IC<i>{}, nbuf); }
                // not DMA0: every wave issues its five windows here, one after every second chain (issued together at the top of the step they
                // keep both waves of a SIMD off the matrix pipe for their whole issue time)
                else if constexpr (!DMA0 && (i & 1) == 0 && i / 2 < C16_NDMA) issue_dma(IC<i / 2>{}, nbuf);
            };
            if (cstep == 0) {                                               // every chain starts here (dead ones on zero weights): no zeroing pass
                if constexpr (FUSE) { if (tb_pending) tables(); }
                c16_range4<CLS, 0, true>(acc, xs, ws, ops, true, dmahook);
                if constexpr (!DMA0) issue_advance();
                c16_range4<CLS, 1, true>(acc, xs, ws, ops, true, nohook);
                c16_range4<CLS, 2, true>(acc, xs, ws, ops, false, nohook);
            } else {
                c16_range4<CLS, 0, false>(acc, xs, ws, ops, dl > 4, dmahook);
                if constexpr (!DMA0) issue_advance();
                if (dl > 4) c16_range4<CLS, 1, false>(acc, xs, ws, ops, dl > 6, nohook);
                if (dl > 6) c16_range4<CLS, 2, false>(acc, xs, ws, ops, false, nohook);
            }
        } else {
            static_for<4>([&](auto ss) {
                constexpr int sq = decltype(ss)::value;
                const int dls = dl - 4 * sq;
                static_for<NA>([&](auto ii) {
                    constexpr int j = decltype(ii)::value, tap = CLS + 4 * j, kh = tap / 5, kw = tap % 5, d = kh + kw;
                    if constexpr (tap < 25) {
                        if (d < dls) {
                            const float av = ws[(sq * 7 + j) * 64];
                            float bv[C16_NT];
#pragma unroll
                            for (int t = 0; t < C16_NT; ++t) bv[t] = xs[sq * C16_PLANE + (kh + t) * C16_HC + kw];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int t = 0; t < C16_NT; ++t) acc[t][j] = mfma16(av, bv[t], acc[t][j]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                });
            });
        }
        C16_T(1);
        if (last) {
#pragma unroll
            for (int t = 0; t < C16_NT; ++t) {
                const f32x4 part = tree4_eval<CIN, CLS>(acc[t]);
#pragma unroll
                for (int r = 0; r < 4; ++r) comb[(FUSE ? 0 : (ntile & 1)) * C16_COMB + (((PS * C16_NT + t) * 4 + CLS) * 4 + r) * 64 + lane] = part[r];
                if constexpr (CIN == 1) {
#pragma unroll
                    for (int i = 0; i < NA; ++i) acc[t][i] = zero4;
                }
            }
        }
        C16_T(2);
        if constexpr (DMA0) {
            if (PS == 0) {
                // the tile's epilogue operands (ordinary loads, issued at the top of this step) are waited for HERE by the compiler -- not behind the
                // DMAs below, which it cannot see: its vmcnt for them after the barrier would count the just-issued DMAs as well
                if (last) asm volatile("" : "+v"(e_bias[0]), "+v"(e_bias[1]), "+v"(e_act[0]), "+v"(e_act[1]), "+v"(e_res[0]), "+v"(e_res[1]));
                if constexpr (SPREAD) issue_advance();                      // (the step's windows went out between the chains of range 0)
                else issue(nbuf);                                           // the step after next: 10 windows, then the cursor moves
                C16_WAIT_PREV4();                                           // the NEXT step's windows (issued a step ago) have landed
            }
        } else C16_WAIT_PREV();                                             // own DMAs of the NEXT step (issued a step ago) have landed
        C16_T(3);
        __syncthreads();
        C16_T(4);
#ifdef C16_STAMP
        st[6] += 1; st[7] += last;
#endif
        if (last) {
            // wave w finishes tile row w >> 1, output channels 2 (w & 1), 2 (w & 1) + 1 of the four groups: (F0 + F2) + (F1 + F3)
            const int trow = WAVE >> 1;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int rbase = 2 * (WAVE & 1);
                const int r = rbase + rr;
                const float *cb = comb + (FUSE ? 0 : (ntile & 1)) * C16_COMB + ((trow * 4) * 4 + r) * 64 + lane;
                const float f0 = cb[0], f1 = cb[4 * 64], f2 = cb[8 * 64], f3 = cb[12 * 64];
                float sv = ((f0 + f2) + (f1 + f3)) + e_bias[rr];
                if (a.act) sv = sv > 0 ? sv : sv * e_act[rr];                // cconv_ec_cuda.cu:311-312
                if (a.residual) sv = sv + e_res[rr];
                if constexpr (FUSE) {                                       // y of (net, group gq, channel ch) at tile position (trow, j): MFMA row m = 3 gq + ch
                    const int m = 4 * (lane >> 4) + r, gq = m / 3, ch = m - 3 * gq;
                    if (m < 3 * GPB) comb[C16_COMB + (((ctile * 3 + cnet) * GPB + gq) * 3 + ch) * 64 + trow * 16 + (lane & 15)] = sv;
                } else if (e_ok && r < a.cout) a.out[e_oi[rr]] = sv;
            }
            cstep = 0;
            ++ntile;
            if constexpr (FUSE) {
                if (c_nsteps == 1) __syncthreads();                         // one-step tiles: the single comb buffer is rewritten before the next barrier
            }
            ++ctile;
            c_rem &= c_rem - 1;
            bool task_done = c_rem == 0;
            if constexpr (FUSE) {
                static_assert(TPT == 2, "the live tiles of a fused task are consecutive (tables phase: tile tb_T + slot)");
                if (task_done) {                                            // this net's sweep over the task's tiles is complete
                    const int nt = ctile;
                    ctile = 0;
                    c_rem = c_mask0;
                    if (++cnet < NSUB) task_done = false;                   // the same tiles, next net
                    else { cnet = 0; tb_pending = true; tb_T = c_tile0 + __builtin_ctz(c_mask0); tb_nt = nt; tb_n = c_n; tb_gb = c_gb; }
                }
            } else if (task_done) ctile = 0;
            if (task_done) {
                ++cq;
                pull(cq + 4);                                               // first read at least one barrier later
                const int u = task(cq);
                if (u < n_my) { decode(u, c_n, c_tile0, c_gb); c_nsteps = steps_of(c_gb); c_mask0 = c_rem = tmask(cq, c_tile0); }
                else done = true;
            }
        } else ++cstep;
        cur = cur + 1 == C16_NBUF ? 0 : cur + 1;
        C16_T(5);
    } while (!done);
    if constexpr (FUSE) { if (tb_pending) tables(); }
    C16_WAIT0();                                                            // no DMA may outlive the workgroup's LDS
#ifdef C16_STAMP
    if constexpr (CIN == 4 && !FUSE) {
        st[9] = __builtin_amdgcn_s_memtime() - t_entry;
        if (lane == 0) for (int i = 0; i < 10; ++i) c16_stamps[((blockIdx.x & 255) * 8 + WAVE) * 10 + i] += st[i];
    }
#endif
}

template <int CIN, bool FUSE>
__global__ __launch_bounds__(C16_THREADS, 2) void k_cconv16(C16Args a) {
    __shared__ float lds[C16_NBUF * C16_BUF];
    __shared__ float comb[FUSE ? (C16_COMB + C16_YFL) : 2 * C16_COMB];     // FUSE: [partial sums | y of the task's tiles x three nets]
    __shared__ int tq[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ps = wave >> 2;
    if constexpr (!FUSE) {
        switch (wave) {
            case 0: c16_body<CIN, 0, FUSE, 0>(a, lds, comb, tq, tid, lane, 0); break;
            case 1: c16_body<CIN, 1, FUSE, 0>(a, lds, comb, tq, tid, lane, 0); break;
            case 2: c16_body<CIN, 2, FUSE, 0>(a, lds, comb, tq, tid, lane, 0); break;
            case 3: c16_body<CIN, 3, FUSE, 0>(a, lds, comb, tq, tid, lane, 0); break;
            case 4: c16_body<CIN, 0, FUSE, 1>(a, lds, comb, tq, tid, lane, 1); break;
            case 5: c16_body<CIN, 1, FUSE, 1>(a, lds, comb, tq, tid, lane, 1); break;
            case 6: c16_body<CIN, 2, FUSE, 1>(a, lds, comb, tq, tid, lane, 1); break;
            default: c16_body<CIN, 3, FUSE, 1>(a, lds, comb, tq, tid, lane, 1); break;
        }
    } else {
        switch (wave & 3) {
            case 0: c16_body<CIN, 0, FUSE>(a, lds, comb, tq, tid, lane, ps); break;
            case 1: c16_body<CIN, 1, FUSE>(a, lds, comb, tq, tid, lane, ps); break;
            case 2: c16_body<CIN, 2, FUSE>(a, lds, comb, tq, tid, lane, ps); break;
            default: c16_body<CIN, 3, FUSE>(a, lds, comb, tq, tid, lane, ps); break;
        }
    }
}

// ------------------------------------------------------------------------------------------------ hidden layers, SKEWED wave sets
// k_cconv16s: the cin = 4 / unfused kernel with the two wave sets of a workgroup HALF A STEP APART.  Stamps of k_cconv16<4,false>
// (tools/ec_stamp.sh) showed what its one-barrier-per-step lockstep costs: both waves of a SIMD run their MFMAs in the same ~3500
// cycles of a 4600-cycle step and their barrier wait, task bookkeeping, epilogue prefetch, tree and final sums in the same ~1000,
// during which the matrix pipe has nothing.  Here wave set 1 passes the step barrier after tap range 0 of ITS step (10 of 25 chains)
// while set 0 passes it at the end of its step: between two barriers ("epoch" e) set 0 runs step e and set 1 runs ranges 1-2 of step
// e-1, the tile bookkeeping and range 0 of step e -- so each set's non-MFMA work sits next to the other set's MFMAs.
//   * LDS ring: during epoch e buffers e-1 (set 1), e (both) are read, so the epoch's DMAs fill buffer e+1: ONE epoch ahead (the
//     lockstep form runs two steps ahead on the same three buffers); every wave waits for all its DMAs before the barrier.
//   * WHO issues the 40 DMAs of a step decides whether the skew pays (tools/ec_variants.sh, ms per 48 images, lockstep kernel 12.3-12.4 on
//     the same box): five per wave in both sets between their chains 12.7-13.3; all 40 in set 0, one behind each chain of range 0: 12.8;
//     all 40 in set 0 as ONE BURST before its range 0 (C16S_N0 = 40, C16S_PLACE = 0): **11.9**; 36 / 32 / 28 / 20 / 12 / 0 of them in set 0 and
//     the rest as a burst in set 1 behind the barrier: 12.0 / 12.0 / 12.0 / 12.8 / 12.8 / 13.2.  A wave with DMAs between its chains runs the
//     matrix pipe at half speed when it is alone on its SIMD (stamps: 2 800 cycles for 50 MFMAs, 1 365 without DMAs), so the DMAs go
//     where no MFMA waits behind them: set 0 issues the burst while set 1 runs ranges 1-2 of the previous step, undisturbed.
//   * a tile's last two tree levels cross LDS inside ONE wave set (comb rows 2 PS, 2 PS + 1): set 1 writes its partial sums in the
//     middle of an epoch, passes the next barrier, then finishes the tile (bias, PReLU, residual, store) -- while it is already
//     inside the next tile.  The epilogue operands therefore load one half-step earlier than they are overwritten: tiles need >= 2
//     steps (ngroup >= 5; the host falls back to k_cconv16 otherwise).
//   * one extra barrier after the loop lets set 1 finish its last tile.
template <int CLS, int PS>
__device__ __forceinline__ void c16s_body(const C16Args &a, float *lds, float *comb, int *tq, const int tid, const int lane) {
    constexpr int CIN = 4, GPB = 4, TPT = C16_TPT, NA = NAcc<4>::value, TCS = 4, WAVE = PS * 4 + CLS;
    constexpr bool FUSE = false;                                            // (C16_T)
    const int G = a.G, C = G * CIN, nout = G * a.cout;
    const long PL = (long)a.hp * a.wp;
    const int xcd = blockIdx.x & 7;
    const int ns_x = (a.N - xcd + 7) >> 3;
    const int n_my = ns_x * a.n_chunks * a.n_gb;
#ifdef C16_STAMP
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t_entry = t0;
#endif
    // ring entries, the optional list of live tasks and the tile masks: as in c16_body
    const int n_live = a.list ? __builtin_amdgcn_readfirstlane(a.cnt[xcd]) : n_my;   // entries of this XCD's list
    auto pull = [&](int k) __attribute__((always_inline)) {
        if (tid == 0) {
            const int i = atomicAdd(a.ctr + xcd, 1);
            tq[k & 7] = i < n_live ? (a.list ? a.list[(long)xcd * a.list_cap + i] : (int)((unsigned)i | 0xfu << 28)) : C16_TASK_END;
        }
    };
    auto task = [&](int k) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(tq[k & 7]) & 0x0fffffff; };
    auto tmask = [&](int k, int tile0) __attribute__((always_inline)) {
        const int nt = a.ntiles - tile0 < TPT ? a.ntiles - tile0 : TPT;
        return (int)(((unsigned)__builtin_amdgcn_readfirstlane(tq[k & 7]) >> 28) & ((1u << nt) - 1u));
    };
    auto decode = [&](int u, int &n, int &tile0, int &gb) __attribute__((always_inline)) {
        const int units = ns_x * a.n_chunks, per = a.gbk * a.n_gb, blk = u / per, r = u - blk * per;
        const int left = units - blk * a.gbk, kk = left < a.gbk ? left : a.gbk;
        gb = a.n_gb - 1 - r / kk;
        const int v = blk * a.gbk + r % kk;
        tile0 = (v % a.n_chunks) * TPT;
        n = xcd + 8 * (v / a.n_chunks);
    };
    auto steps_of = [&](int gb) __attribute__((always_inline)) {
        int gl = gb * GPB + GPB - 1;
        if (gl > G - 1) gl = G - 1;
        int L = gl + 4 + a.hidden;
        if (L > G) L = G;
        return (L + TCS - 1) / TCS;
    };
    pull(0);
    pull(1);
    pull(2);
    pull(3);
    pull(4);
    __syncthreads();
    // ---- LDS-DMA windows of this wave (as in c16_body)
    constexpr int QPR = C16_HC / 4, QPP = C16_PLANE / 4;
    // all DMAs of a step are issued by wave set 0 (ten windows per wave: j = CLS + 4 m), set 1's stream holds MFMAs and LDS reads only
#ifndef C16S_PLACE
#define C16S_PLACE 0                            // 0: a set issues its DMAs as a burst at the start of its epoch part, 1: set 0 one behind each chain of range 0
#endif
#ifndef C16S_N0
#define C16S_N0 40                              // windows 0 .. N0-1 of a step image are issued by wave set 0, the rest by set 1 (multiples of 4)
#endif
    constexpr int NDW = PS == 0 ? C16S_N0 / 4 : (40 - C16S_N0) / 4, NXD = 3;
    auto window = [](int m) constexpr { return (PS == 0 ? 0 : C16S_N0) + CLS + 4 * m; };
    int xpl[NXD], xg[NXD];
    unsigned xoff[NXD];
#pragma unroll
    for (int m = 0; m < NXD; ++m) {
        int e = window(m) * 64 + lane;
        if (e >= 16 * QPP) e = 0;
        const int slot = e / QPP;
        int rem = e - slot * QPP;
        if (rem >= C16_HR * QPR) rem = C16_HR * QPR - 1;
        const int row = rem / QPR, cq = rem - row * QPR;
        xpl[m] = slot;
        xg[m] = row * a.wp + cq * 4;
        xoff[m] = (unsigned)(((long)xpl[m] * PL + xg[m]) * 4);
    }
    const unsigned woff = lane * 16;
    const bool ragged = C % (TCS * CIN) != 0;
    const unsigned lds_base = c16_lds_addr(lds);
    // ---- issue cursor: one step image per epoch, the same step in both wave sets
    int iq = 0, i_rem = 0, istep = 0, i_nsteps = 0, i_tile0 = 0, i_n = 0, i_gb = 0;   // i_rem: live tiles of the task still to issue
    bool ivalid = false;
    const float *ixb = a.x, *iwb = a.packed;
    auto issue_task = [&]() __attribute__((always_inline)) {
        const int u = task(iq);
        ivalid = u < n_my;
        if (ivalid) {
            decode(u, i_n, i_tile0, i_gb);
            i_rem = tmask(iq, i_tile0);
            i_nsteps = steps_of(i_gb);
            iwb = a.packed + ((long)(i_n / a.npb) * a.n_gb + i_gb) * a.NS * C16_WFL;
        }
    };
    auto issue_tile = [&]() __attribute__((always_inline)) {
        const int T = i_tile0 + __builtin_ctz(i_rem), ty = T / a.ntx, tx = T - ty * a.ntx;
        ixb = a.x + (long)(i_n % a.x_mod) * C * PL + (long)(ty * C16_TH) * a.wp + tx * C16_TW;
    };
    auto issue_dma = [&](auto mm, int buf) __attribute__((always_inline)) {
        constexpr int m = decltype(mm)::value, j = window(m);
        const unsigned lb = lds_base + (unsigned)buf * (C16_BUF * 4);
        const float *wsrc = iwb + (long)istep * C16_WFL;
        if constexpr (m < NXD && j < C16_XWIN) {
            const int c0 = istep * (TCS * CIN);
            unsigned vo = xoff[m];
            if (ragged && c0 + TCS * CIN > C) {
                const int c = c0 + xpl[m] > C - 1 ? C - 1 - c0 : xpl[m];
                vo = (unsigned)(((long)c * PL + xg[m]) * 4);
            }
            c16_dma_s<0>(vo, c16_uniform(ixb + (long)c0 * PL), lb + j * 1024);
        } else if constexpr (j < C16_XWIN + C16_WWIN) c16_dma_s(woff, c16_uniform(wsrc + (j - C16_XWIN) * 256), lb + j * 1024);
        else c16_dma_s(woff, c16_uniform(wsrc), lb + j * 1024);
    };
    auto issue_advance = [&]() __attribute__((always_inline)) {
        if (ivalid) {
            if (++istep == i_nsteps) {
                istep = 0;
                i_rem &= i_rem - 1;
                if (!i_rem) {
                    ++iq;
                    issue_task();
                }
                if (ivalid) issue_tile();
                else istep = i_nsteps - 1;                                  // idle: stay on the last step
            }
        }
    };
    auto issue_all = [&](int buf) __attribute__((always_inline)) {
        static_for<NDW>([&](auto mm) { issue_dma(mm, buf); });
        issue_advance();
    };
    if (task(0) >= n_my) return;                                            // uniform: the whole workgroup has no task
    issue_task();
    issue_tile();
    issue_all(0);                                                           // step 0 -> buffer 0 (each set its share of the windows)
    // ---- compute cursor
    int cq = 0, cstep = 0, c_nsteps, c_tile0, c_n, c_gb;
    decode(task(0), c_n, c_tile0, c_gb);
    int c_rem = tmask(0, c_tile0);                                          // live tiles of the task still to compute; the current one is its lowest bit
    c_nsteps = steps_of(c_gb);
    f32x4 acc[C16_NT][NA];
    const int xlane = (lane >> 4) * 4 * C16_PLANE + (C16_NT * PS) * C16_HC + (lane & 15);
    const int wlane = C16_XFL + CLS * C16_SLOTS * 64 + lane;
    C16_WAIT0();
    __syncthreads();                                                        // barrier "-1": step 0 is in LDS
    float e_bias[2] = {0.f, 0.f}, e_act[2] = {0.f, 0.f}, e_res[2] = {0.f, 0.f};
    long e_oi[2] = {0, 0};
    bool e_ok = false;
    const float *const act_p = a.act ? a.act : a.bias, *const res_p = a.residual ? a.residual : a.x;
    // bias / PReLU slope / residual of the current tile (clamped addresses), fetched one epoch before the tile is finished
    auto prefetch_epilogue = [&]() __attribute__((always_inline)) {
        const int T = c_tile0 + __builtin_ctz(c_rem), ty = T / a.ntx, tx = T - ty * a.ntx;
        const int q = lane >> 4, j = lane & 15;
        const int y = ty * C16_TH + (WAVE >> 1), x = tx * C16_TW + j, g = c_gb * 4 + q;
        e_ok = g < G && y < a.H && x < a.W;
        const int gc = g < G ? g : G - 1;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * (WAVE & 1) + rr;
            const int rc = r < a.cout ? r : a.cout - 1;
            const int o = gc * a.cout + rc, bid = (c_n / a.npb) * nout + o;
            e_oi[rr] = ((long)c_n * nout + o) * PL + (long)((y < a.H ? y : a.H - 1) + 2) * a.wp + (x < a.W ? x : a.W - 1) + 2;
            e_bias[rr] = a.bias[bid];
            e_act[rr] = act_p[bid];
            e_res[rr] = res_p[a.residual ? e_oi[rr] : 0];
        }
    };
    auto tree_part = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < C16_NT; ++t) {
            const f32x4 part = tree4_eval<CIN, CLS>(acc[t]);
#pragma unroll
            for (int r = 0; r < 4; ++r) comb[par * C16_COMB + (((PS * C16_NT + t) * 4 + CLS) * 4 + r) * 64 + lane] = part[r];
        }
    };
    // wave w finishes tile row w >> 1, output channels 2 (w & 1), 2 (w & 1) + 1 of the four groups: (F0 + F2) + (F1 + F3)
    auto finalize = [&](int par) __attribute__((always_inline)) {
        constexpr int trow = WAVE >> 1;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            constexpr int rbase = 2 * (WAVE & 1);
            const int r = rbase + rr;
            const float *cb = comb + par * C16_COMB + ((trow * 4) * 4 + r) * 64 + lane;
            const float f0 = cb[0], f1 = cb[4 * 64], f2 = cb[8 * 64], f3 = cb[12 * 64];
            float sv = ((f0 + f2) + (f1 + f3)) + e_bias[rr];
            if (a.act) sv = sv > 0 ? sv : sv * e_act[rr];                    // cconv_ec_cuda.cu:311-312
            if (a.residual) sv = sv + e_res[rr];
            if (e_ok && r < a.cout) a.out[e_oi[rr]] = sv;
        }
    };
    int cur = 0, ntile = 0;                                                 // cur: LDS buffer of the compute cursor's step
    bool done = false;
    // the compute cursor leaves a step (`last`: it was its tile's last one)
    auto advance_compute = [&](bool last) __attribute__((always_inline)) {
        if (last) {
            cstep = 0;
            ++ntile;
            c_rem &= c_rem - 1;
            if (!c_rem) {
                ++cq;
                if constexpr (PS == 0) pull(cq + 4);                        // first read at least one barrier later
                const int u = task(cq);
                if (u < n_my) { decode(u, c_n, c_tile0, c_gb); c_nsteps = steps_of(c_gb); c_rem = tmask(cq, c_tile0); }
                else done = true;
            }
        } else ++cstep;
        cur = cur + 1 == C16_NBUF ? 0 : cur + 1;
    };
    auto nohook = [](auto) __attribute__((always_inline)) {};
    C16Ops ops[C16_PF + 1];
    C16_T(8);
    if constexpr (PS == 0) {
        do {
            const float *xs = lds + cur * C16_BUF + xlane, *ws = lds + cur * C16_BUF + wlane;
            const int nbuf = cur + 1 == C16_NBUF ? 0 : cur + 1;             // this epoch's DMAs: the next step
            const int dl = c_gb * GPB + GPB + 3 + a.hidden - cstep * TCS;   // tap diagonals d >= dl carry only zero weights
            const bool last = cstep + 1 == c_nsteps;
            if (last) prefetch_epilogue();
            C16_T(0);
            static_for<C16_PF>([&](auto ii) { c16_load4<CLS, c16_range_tap(0, decltype(ii)::value)>(ops[decltype(ii)::value], xs, ws); });
            auto dmahook = [&](auto ii) __attribute__((always_inline)) {
                constexpr int i = decltype(ii)::value;
                if constexpr (C16S_PLACE == 1 && i < NDW) issue_dma(IC<i>{}, nbuf);
            };
            if constexpr (C16S_PLACE == 0) issue_all(nbuf);
            if (cstep == 0) {                                               // every chain starts here (dead ones on zero weights)
                c16_range4<CLS, 0, true>(acc, xs, ws, ops, true, dmahook);
                if constexpr (C16S_PLACE == 1) issue_advance();
                c16_range4<CLS, 1, true>(acc, xs, ws, ops, true, nohook);
                c16_range4<CLS, 2, true>(acc, xs, ws, ops, false, nohook);
            } else {
                c16_range4<CLS, 0, false>(acc, xs, ws, ops, dl > 4, dmahook);
                if constexpr (C16S_PLACE == 1) issue_advance();
                if (dl > 4) c16_range4<CLS, 1, false>(acc, xs, ws, ops, dl > 6, nohook);
                if (dl > 6) c16_range4<CLS, 2, false>(acc, xs, ws, ops, false, nohook);
            }
            C16_T(1);
            if (last) tree_part(ntile & 1);
            C16_T(2);
            C16_WAIT0();
            C16_T(3);
            __syncthreads();
            C16_T(4);
#ifdef C16_STAMP
            st[6] += 1; st[7] += last;
#endif
            if (last) finalize(ntile & 1);
            advance_compute(last);
            C16_T(5);
        } while (!done);
        C16_WAIT0();
        __syncthreads();                                                    // set 1 finishes its last tile behind this one
    } else {
        // epoch 0: the DMAs of step 1, range 0 of step 0
        issue_all(1);
        int dl = c_gb * GPB + GPB + 3 + a.hidden;
        bool last = false;                                                  // (tiles have >= 2 steps)
        bool fin_pending = false;
        int fin_par = 0;
        {
            const float *xs = lds + xlane, *ws = lds + wlane;
            static_for<C16_PF>([&](auto ii) { c16_load4<CLS, c16_range_tap(0, decltype(ii)::value)>(ops[decltype(ii)::value], xs, ws); });
            c16_range4<CLS, 0, true>(acc, xs, ws, ops, true, nohook);
        }
        C16_T(1);
        while (true) {
            C16_WAIT0();
            C16_T(3);
            __syncthreads();
            C16_T(4);
            if (fin_pending) { finalize(fin_par); fin_pending = false; }
            // ---- second half of the step: ranges 1-2 on buffer cur, this epoch's DMAs (the step after next) between the chains of range 1
            {
                const float *xs = lds + cur * C16_BUF + xlane, *ws = lds + cur * C16_BUF + wlane;
                const int nbuf = cur >= 1 ? cur - 1 : C16_NBUF - 1;         // (cur + 2) mod 3
                if (cstep + 2 == c_nsteps) prefetch_epilogue();             // the tile ends with the next step
                C16_T(5);
                issue_all(nbuf);                                            // set 1's share of the step after next, as a burst behind the barrier
                if (cstep == 0) {
                    c16_range4<CLS, 1, true>(acc, xs, ws, ops, true, nohook);
                    c16_range4<CLS, 2, true>(acc, xs, ws, ops, false, nohook);
                } else if (dl > 4) {
                    c16_range4<CLS, 1, false>(acc, xs, ws, ops, dl > 6, nohook);
                    if (dl > 6) c16_range4<CLS, 2, false>(acc, xs, ws, ops, false, nohook);
                }
            }
            C16_T(1);
            if (last) { tree_part(ntile & 1); fin_pending = true; fin_par = ntile & 1; }
            C16_T(2);
#ifdef C16_STAMP
            st[6] += 1; st[7] += last;
#endif
            advance_compute(last);
            if (done) break;
            // ---- first half of the next step: range 0 on buffer cur
            {
                const float *xs = lds + cur * C16_BUF + xlane, *ws = lds + cur * C16_BUF + wlane;
                dl = c_gb * GPB + GPB + 3 + a.hidden - cstep * TCS;
                last = cstep + 1 == c_nsteps;
                C16_T(0);
                static_for<C16_PF>([&](auto ii) { c16_load4<CLS, c16_range_tap(0, decltype(ii)::value)>(ops[decltype(ii)::value], xs, ws); });
                if (cstep == 0) c16_range4<CLS, 0, true>(acc, xs, ws, ops, true, nohook);
                else c16_range4<CLS, 0, false>(acc, xs, ws, ops, dl > 4, nohook);
            }
            C16_T(1);
        }
        C16_WAIT0();
        __syncthreads();
        if (fin_pending) finalize(fin_par);
    }
    C16_WAIT0();                                                            // no DMA may outlive the workgroup's LDS
#ifdef C16_STAMP
    st[9] = __builtin_amdgcn_s_memtime() - t_entry;
    if (lane == 0) for (int i = 0; i < 10; ++i) c16_stamps[((blockIdx.x & 255) * 8 + WAVE) * 10 + i] += st[i];
#endif
}

__global__ __launch_bounds__(C16_THREADS, 2) void k_cconv16s(C16Args a) {
    __shared__ float lds[C16_NBUF * C16_BUF];
    __shared__ float comb[2 * C16_COMB];
    __shared__ int tq[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    switch (wave) {
        case 0: c16s_body<0, 0>(a, lds, comb, tq, tid, lane); break;
        case 1: c16s_body<1, 0>(a, lds, comb, tq, tid, lane); break;
        case 2: c16s_body<2, 0>(a, lds, comb, tq, tid, lane); break;
        case 3: c16s_body<3, 0>(a, lds, comb, tq, tid, lane); break;
        case 4: c16s_body<0, 1>(a, lds, comb, tq, tid, lane); break;
        case 5: c16s_body<1, 1>(a, lds, comb, tq, tid, lane); break;
        case 6: c16s_body<2, 1>(a, lds, comb, tq, tid, lane); break;
        default: c16s_body<3, 1>(a, lds, comb, tq, tid, lane); break;
    }
}

static int c16_fill_args(C16Args &a, const lic360_conv_plan *p, int h, int w, int tpt = C16_TPT) {
    a.G = p->ngroup; a.cout = p->cout; a.hidden = p->constrain == 5 ? 0 : 1; a.H = h; a.W = w;
    if (lic360_ec16_layout(h, w, &a.hp, &a.wp)) return 2;
    a.n_gb = conv16_ngb(p);
    a.ntx = (w + C16_TW - 1) / C16_TW;
    a.ntiles = a.ntx * ((h + C16_TH - 1) / C16_TH);
    a.n_chunks = (a.ntiles + tpt - 1) / tpt;
    a.NS = conv16_nsteps_max(p);
    a.gbk = lic360_ec_gbk();                                                // task-order block size: 16 (blocks of 4 .. all measured: DESIGN 4.1 a)
    a.code = a.mask = nullptr; a.pidx = a.plane_start = nullptr; a.rec = nullptr;
    a.list = a.cnt = nullptr; a.list_cap = 0;
    return 0;
}

// x / residual / out: [n][C | nout][hp][wp] zero-haloed planes (lic360_ec16_layout).  ctr: 8 ints of device scratch.
// list / cnt / cap: the layer's live tasks per XCD (need.h, lic360_ec_lists_build: [8][cap] entries, [8] counts), or NULL = every task.  The cells of
// a skipped (tile, group block) are not written.  Internal entry (hidden visibility): the fused codec's dead-cone skip.
int lic360_cconv16_ec_list(void *stream, const lic360_conv_plan *p, const float *x, const float *packed16, const float *bias,
                           const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod, int *ctr,
                           const int *list, const int *cnt, int cap) {
    ARG_CHECK(p && conv16_ok(p) && x && packed16 && bias && out && ctr && n > 0 && h > 0 && w > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n);
    ARG_CHECK((list != nullptr) == (cnt != nullptr) && (!list || (cap > 0 && p->cin == 4)));
    C16Args a;
    a.x = x; a.packed = packed16; a.bias = bias; a.act = act; a.residual = residual; a.out = out; a.ctr = ctr;
    a.npb = n / nb; a.x_mod = x_mod; a.N = n;
    if (c16_fill_args(a, p, h, w)) return 2;
    a.list = list; a.cnt = cnt; a.list_cap = cap;
    ARG_CHECK((long)((n + 7) / 8) * a.n_chunks * a.n_gb < C16_TASK_END);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(ctr, 0, 8 * sizeof(int), s));
    // cin = 4: the skewed form (k_cconv16s) when every tile has at least two steps; nets of fewer than five groups keep the lockstep kernel
    const int l0 = 3 + 4 + a.hidden, l0c = l0 < p->ngroup ? l0 : p->ngroup;     // chain length of the first group block
    if (p->cin == 4 && l0c > 4) hipLaunchKernelGGL(k_cconv16s, dim3(256), dim3(C16_THREADS), 0, s, a);
    else if (p->cin == 4) hipLaunchKernelGGL((k_cconv16<4, false>), dim3(256), dim3(C16_THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_cconv16<1, false>), dim3(256), dim3(C16_THREADS), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_cconv16_ec(void *stream, const lic360_conv_plan *p, const float *x, const float *packed16, const float *bias,
                                 const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod, int *ctr) {
    return lic360_cconv16_ec_list(stream, p, x, packed16, bias, act, residual, out, n, h, w, nb, x_mod, ctr, nullptr, nullptr, 0);
}

// Last layer of the latent entropy model fused with the CDF-table build (SURVEY.md §7 `k_cconv_ec_last_gmm`; replaces the last
// CconvEcBatch + TileExtractBatch + EntropyBatchGmmTable of EntEncoderFast.forward, test/lic360_demo.py:132-140,
// extension/entropy_gmm_table_cuda.cu:138-191): x = [3 * images][C][hp][wp] activations of the three stacked nets
// [weight, sigma, mu] (net-major), cout must be 3; the nets' outputs of a tile meet in LDS and never reach HBM; per symbol
// only (cdf[sym], cdf[sym+1]) is written, at the symbol's place in coding order.  pidx / plane_start: device copies of the
// CodeContex prefix table and of the first-record-of-plane table.
// list / cnt / cap as in lic360_cconv16_ec_list (five groups per block, two tiles per task, N = images).  The records of a skipped (tile, group block) are
// NOT written -- all its symbols are masked, their records are (0, 0): the caller clears `rec` before a launch with a list.
int lic360_cconv16_ec_tables_list(void *stream, const lic360_conv_plan *p, const float *x, const float *packed16, const float *bias,
                                  const float *code, const float *mask, const int *pidx_dev, const int *plane_start_dev,
                                  void *rec, int images, int h, int w, int *ctr, const int *list, const int *cnt, int cap) {
    ARG_CHECK(p && conv16_ok(p) && p->cin == 4 && p->cout == 3 && x && packed16 && bias && code && mask && pidx_dev && plane_start_dev && rec &&
              ctr && images > 0 && h > 0 && w > 0);
    ARG_CHECK((list != nullptr) == (cnt != nullptr) && (!list || cap > 0));
    C16Args a;
    a.x = x; a.packed = packed16; a.bias = bias; a.act = nullptr; a.residual = nullptr; a.out = nullptr; a.ctr = ctr;
    a.npb = images; a.x_mod = 3 * images; a.N = images;
    if (c16_fill_args(a, p, h, w, C16_FTPT)) return 2;
    a.n_gb = (p->ngroup + C16_FGPB - 1) / C16_FGPB;                         // five groups per block (lic360_conv16_pack_tables)
    a.code = code; a.mask = mask; a.pidx = pidx_dev; a.plane_start = plane_start_dev; a.rec = (uint2 *)rec;
    a.list = list; a.cnt = cnt; a.list_cap = cap;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(ctr, 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL((k_cconv16<4, true>), dim3(256), dim3(C16_THREADS), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_cconv16_ec_tables(void *stream, const lic360_conv_plan *p, const float *x, const float *packed16, const float *bias,
                                        const float *code, const float *mask, const int *pidx_dev, const int *plane_start_dev,
                                        void *rec, int images, int h, int w, int *ctr) {
    return lic360_cconv16_ec_tables_list(stream, p, x, packed16, bias, code, mask, pidx_dev, plane_start_dev, rec, images, h, w, ctr, nullptr, nullptr, 0);
}
