// cconv144_kernels.hip -- masked convolution of the IMPORTANCE-MAP entropy net (test/lic360_demo.py:153-161, 254-262: one group,
// 144 hidden channels: layers 144 -> 144 and 144 -> 49) on v_mfma_f32_16x16x4_f32, encode order and decode order.
//
// Arithmetic contract (extension/cconv_ec_cuda.cu:54-96) with ngroup = 1: virtual lane l of an output scalar
// walks the flat indices  l, l+128, l+256, ... < 25*C  (index -> kw = index%5, kh = (index/5)%5, gid = index/25) and adds ONE
// term x[gid][pos + tap] * w[o][gid][tap] per causal index (kh + kw <= 3 + hidden; the others contribute nothing), all in one
// fmaf chain; then the fixed tree p[i]+p[i+64]; +32; ...; +1.  With K = four CONSECUTIVE causal terms of one lane's chain an
// MFMA advances 16 output channels x 16 positions of that lane by four terms in order (bit for bit an fmaf chain).
//
// Mapping (leaf-resident: all 128 chains of a 16 x 16 output tile own accumulators until the tree):
//   * everything about a chain -- which (channel, tap) its q-th quad of terms reads -- is a compile-time constant of
//     (C, hidden, l, q); the kernel is fully unrolled over it, so the B operand of an MFMA is ONE ds_read_b32 at
//     `select(k; 4 literals) + lane` and the A operand ONE coalesced 256-byte global load from a stream packed in the same order;
//   * wave c of a workgroup (8 waves) owns lanes l = c (mod 8): 16 chains, the top four tree levels (+64, +32, +16, +8) in
//     registers; the last three levels cross the waves through LDS;
//   * a workgroup keeps the zero-haloed x tile of ALL 144 input channels of NT position tiles in LDS (LDS-DMA once per task)
//     and sweeps the output-channel tiles over it; weights go straight to registers, prefetched four blocks ahead;
//   * EC layout: zero-haloed NCHW planes, a task = NT consecutive rows x 16 columns of one map;
//     DC layout: zero-padded diagonal-major planes (cell (th, tw) at row th+tw+4, column th+2), a task = the NT x 16
//     positions th0.. of anti-diagonal s of one map: what decode plane s touches.
#include "common.h"
#include "conv_plan.h"
#include "cconv_tree.h"

#define I144_THREADS 512

// ------------------------------------------------------------------------------------------------ compile-time chain tables
template <int C, int HIDDEN>
struct ITerms {
    static constexpr int NIDX = 25 * C;
    static constexpr bool causal(int tap) { return tap / 5 + tap % 5 <= 3 + HIDDEN; }        // cconv_ec_cuda.cu:71-73 with ngroup = 1
    static constexpr int nterms(int l) {
        int n = 0;
        for (int i = l; i < NIDX; i += 128) n += causal(i % 25) ? 1 : 0;
        return n;
    }
    static constexpr int term(int l, int t) {                // flat index of lane l's t-th causal term, -1 past the end
        int n = 0;
        for (int i = l; i < NIDX; i += 128)
            if (causal(i % 25)) { if (n == t) return i; ++n; }
        return -1;
    }
    static constexpr int nquads(int l) { return (nterms(l) + 3) / 4; }
    static constexpr int qbase(int l) {                      // position of lane l's first quad in the packed stream
        int s = 0;
        for (int j = 0; j < l; ++j) s += nquads(j);
        return s;
    }
    static constexpr int NQ = qbase(128);
    // wave c of a workgroup owns lanes l = c + 8 a (a < 16) and runs their quads in (a, q) order: its "blocks" b = 0 .. nblocks(c) - 1
    static constexpr int nblocks(int c) { int s = 0; for (int a = 0; a < 16; ++a) s += nquads(c + 8 * a); return s; }
    static constexpr int blk_a(int c, int b) { int a = 0; while (b >= nquads(c + 8 * a)) { b -= nquads(c + 8 * a); ++a; } return a; }
    static constexpr int blk_q(int c, int b) { int a = 0; while (b >= nquads(c + 8 * a)) { b -= nquads(c + 8 * a); ++a; } return b; }
    static constexpr int ngroups() { int m = 0; for (int c = 0; c < 8; ++c) { const int g = (nblocks(c) + 3) / 4; m = g > m ? g : m; } return m; }
    static constexpr int NG = ngroups();                     // groups of four blocks per wave (padded to the longest wave)
};

static inline bool conv144_ok(const lic360_conv_plan *p) {
    // constrain 6 only: the one-group net's first layer has a single input channel (the generic kernel serves it)
    return p->ksz == 5 && p->ngroup == 1 && p->C == 144 && p->nout >= 1 && p->nout <= 144 && p->constrain == 6;
}
template <int HIDDEN> static int i144_nq() { return ITerms<144, HIDDEN>::NQ; }
static int conv144_nq(const lic360_conv_plan *p) { (void)p; return i144_nq<1>(); }
static int conv144_otiles(const lic360_conv_plan *p) { return (p->nout + 15) / 16; }

// packed144[otile][wave c][group][lane 16 k + i][4]: word j of a lane's quad = w[16 otile + i][gid][tap] of the k-th term of the quad Q that wave c
// runs as its block 4 group + j (0 past a chain's end / past nout / past the wave's last block).  A wave fetches the A operands of FOUR consecutive
// MFMA blocks with one 16-byte load per lane (round 4: one global_load_dword per block kept the CU's address unit -- 9 cycles per wave instruction
// of that width, 18 for a 16-byte one -- busy for more than half of the kernel; qmap: block -> Q per wave).
__global__ void k_conv144_pack(const float *__restrict__ weight, const int *__restrict__ src, const int *__restrict__ qmap, float *__restrict__ packed,
                               int nout, int C, int NG, long total) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int jw = (int)(e & 3), l = (int)((e >> 2) & 63), i = l & 15, k = l >> 4;
        const long t = e >> 8;
        const int g = (int)(t % NG), c = (int)((t / NG) & 7), ot = (int)(t / (8 * NG)), o = ot * 16 + i;
        const int Q = qmap[(c * NG + g) * 4 + jw];
        const int s = Q >= 0 ? src[Q * 4 + k] : -1;
        packed[e] = (s >= 0 && o < nout) ? weight[(long)o * C * 25 + s] : 0.0f;         // s = gid * 25 + tap = the flat index itself
    }
}
template <int HIDDEN>
static void i144_fill_src(int *src, int *qmap) {
    using T = ITerms<144, HIDDEN>;
    int Q = 0;
    for (int l = 0; l < 128; ++l) {
        const int nt = T::nterms(l);
        for (int q = 0; q < T::nquads(l); ++q, ++Q)
            for (int k = 0; k < 4; ++k) src[Q * 4 + k] = 4 * q + k < nt ? T::term(l, 4 * q + k) : -1;
    }
    for (int c = 0; c < 8; ++c)
        for (int b = 0; b < 4 * T::NG; ++b)
            qmap[c * 4 * T::NG + b] = b < T::nblocks(c) ? T::qbase(c + 8 * T::blk_a(c, b)) + T::blk_q(c, b) : -1;
}
LIC360_API int lic360_conv144_supported(const lic360_conv_plan *p) { return p && conv144_ok(p) ? 1 : 0; }
LIC360_API long lic360_conv144_packed_floats(const lic360_conv_plan *p) {
    return p && conv144_ok(p) ? (long)conv144_otiles(p) * 8 * ITerms<144, 1>::NG * 256 : 0;
}
LIC360_API int lic360_conv144_pack(void *stream, const lic360_conv_plan *p, const float *weight, float *packed) {
    ARG_CHECK(p && conv144_ok(p) && weight && packed);
    const int NQ = conv144_nq(p), NG = ITerms<144, 1>::NG, nint = 4 * NQ + 8 * 4 * NG;
    int *h = (int *)malloc(sizeof(int) * nint), *d = nullptr;
    i144_fill_src<1>(h, h + 4 * NQ);
    hipError_t e = hipMalloc((void **)&d, sizeof(int) * nint);
    if (e == hipSuccess) e = hipMemcpyAsync(d, h, sizeof(int) * nint, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) {
        const long total = lic360_conv144_packed_floats(p);
        hipLaunchKernelGGL(k_conv144_pack, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, weight, d, d + 4 * NQ, packed, p->nout, p->C, NG,
                           total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);          // h / d are released below
    free(h);
    if (d) (void)hipFree(d);
    HIP_TRY(e);
    return 0;
}

// ------------------------------------------------------------------------------------------------ layouts
// EC: staged tile = (NT + 4) rows x 20 columns of the zero-haloed NCHW plane; position tile t = output row r0 + t, columns c0..c0+15;
//     tap (kh, kw) of tile t reads staged (t + kh, j + kw).
// DC: staged tile = 5 anti-diagonals s-4..s x (16 NT + 4) columns of the diagonal-major plane; position tile t = rows th0+16t..+15 of
//     diagonal s; tap (kh, kw) (kh + kw <= 4) reads staged (kh + kw, 16 t + j + kh).
template <bool DC, int NT>
struct ILay {
    static constexpr int ROWS = DC ? 5 : NT + 4, COLS = DC ? 16 * NT + 4 : 20, QPR = COLS / 4, PLANE = ROWS * COLS, TOFF = DC ? 16 : COLS;
    static constexpr int tap_off(int kh, int kw) { return DC ? (kh + kw) * COLS + kh : kh * COLS + kw; }
};
#define I144_R0 4                               // DC layout: zero diagonals before s = 0
#define I144_C0 2                               // DC layout: zero columns before th = 0

struct I144Args {
    const float *x, *packed, *bias, *act, *residual;
    float *out;
    int N, H, W, nout, n_ot;
    int grid;                                   // workgroups of the launch (reading gridDim would add HIP's 256 bytes of implicit kernel arguments to every launch:
                                                //   argument bytes cost time when several streams launch at once, DESIGN.md 4.1 b'')
    long xplane, xsample;                       // floats per input plane / sample
    int xpitch;
    long oplane, osample;                       // output (and residual) addressing: plane / sample strides,
    int opitch, ooff;                           //   cell (r, c) at [(r + ooff) * opitch + c + ooff] (EC)
    int tiles_r, tiles_c;                       // EC: row blocks / column blocks per map
    int s, th_lo, th_hi, th0;                   // DC: anti-diagonal, its valid rows, first row of the task's window (multiple of 4: 16-byte DMAs)
    int n_seg;                                  // DC: 16 NT-row segments of the anti-diagonal (maps taller than one task's window)
    int og, n_og;                               // DC: output-channel tiles per task / tasks per map (a plane of one map is little work:
                                                //     several workgroups share it, each staging the map's x tile for its own tiles)
};

#define I144_WAIT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
__device__ __forceinline__ void i144_dma_x4(const float *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ unsigned i144_lds_addr(const float *p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const float *)p;
}
__device__ __forceinline__ f32x4 i144_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

#define I144_PFG 2                              // weight prefetch depth (groups of four blocks)

template <int HIDDEN, bool DC, int NT, int WAVE>
__device__ __forceinline__ void i144_body(const I144Args &a, float *xs, float *comb, const int tid, const int lane) {
    using T = ITerms<144, HIDDEN>;
    using L = ILay<DC, NT>;
    constexpr int C = 144;
    constexpr int XQ = C * L::ROWS * L::QPR;                                // 16-byte quads of the x tile
    constexpr int NDMA = (XQ + 511) / 512;                                  // DMAs per wave (8 waves x 64 lanes per round)
    const unsigned lds_x = i144_lds_addr(xs);
    // per-lane source offsets (floats) of this wave's DMA windows, relative to the task's tile origin
    int soff[NDMA];
#pragma unroll
    for (int m = 0; m < NDMA; ++m) {
        int e = (WAVE + 8 * m) * 64 + lane;
        if (e >= XQ) e = XQ - 1;                                            // slack lanes of the last window re-fetch the last quad
        const int pl = e / (L::ROWS * L::QPR), rem = e - pl * (L::ROWS * L::QPR), row = rem / L::QPR, cq = rem - row * L::QPR;
        soff[m] = (int)(pl * a.xplane) + row * a.xpitch + cq * 4;
    }
    const int kq = lane >> 4, j = lane & 15;
    const int ntasks = DC ? a.N * a.n_og * a.n_seg : a.N * a.tiles_r * a.tiles_c;
    int parity = 0;
    for (int task = blockIdx.x; task < ntasks; task += a.grid) {
        // ---- task geometry
        int n, r0 = 0, c0 = 0, ot_lo = 0, ot_hi = a.n_ot, th0 = 0;
        if constexpr (DC) {
            const int per = a.n_og * a.n_seg;
            n = task / per;
            const int rem = task - n * per, seg = rem / a.n_og;
            ot_lo = (rem - seg * a.n_og) * a.og;
            ot_hi = ot_lo + a.og < a.n_ot ? ot_lo + a.og : a.n_ot;
            th0 = a.th0 + seg * (16 * NT);
        } else {
            n = task / (a.tiles_r * a.tiles_c);
            const int rem = task - n * (a.tiles_r * a.tiles_c);
            r0 = (rem / a.tiles_c) * NT;
            c0 = (rem % a.tiles_c) * 16;
        }
        // tile origin in the input plane: EC padded (r0, c0) [= image (r0-2, c0-2)]; DC row s - 4 + R0, column th0 (= image row th0 - 2)
        const float *xt = a.x + (long)n * a.xsample + (DC ? (long)(a.s - 4 + I144_R0) * a.xpitch + th0 : (long)r0 * a.xpitch + c0);
        __syncthreads();                                                    // every wave is done with the previous x tile
        static_for<NDMA>([&](auto mm) {
            constexpr int m = decltype(mm)::value;
            if constexpr ((WAVE + 8 * m) * 64 < XQ) i144_dma_x4(xt + soff[m], lds_x + (unsigned)((WAVE + 8 * m) * 1024));
        });
        I144_WAIT0();
        __syncthreads();
        for (int ot = ot_lo; ot < ot_hi; ++ot) {
            const float *wot = a.packed + ((long)ot * 8 + WAVE) * (T::NG * 256) + lane * 4;
            f32x4 acc[NT][16];
            // ---- the 16 chains of this wave, quad by quad; block index b runs over (a = lane slot, q) in order; the A operands of blocks
            // 4 g .. 4 g + 3 are one f32x4 per lane, fetched I144_PFG groups ahead
            constexpr int NB = T::nblocks(WAVE), NGW = (NB + 3) / 4, RING = I144_PFG + 1;
            f32x4 aring[RING];
            static_for<(I144_PFG < NGW ? I144_PFG : NGW)>([&](auto gg) {
                constexpr int g = decltype(gg)::value;
                aring[g % RING] = *(const f32x4 *)(wot + g * 256);
            });
            static_for<NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value, aa = T::blk_a(WAVE, b), q = T::blk_q(WAVE, b), l = WAVE + 8 * aa;
                if constexpr (b % 4 == 0 && b / 4 + I144_PFG < NGW) aring[(b / 4 + I144_PFG) % RING] = *(const f32x4 *)(wot + (b / 4 + I144_PFG) * 256);
                const float av = aring[(b / 4) % RING][b % 4];
                // B operand address: term k of the quad reads channel gid_k, tap_k (compile-time); past the chain's end any valid cell
                constexpr int nt = T::nterms(l);
                auto off_of = [](int k) constexpr {
                    const int t = 4 * q + k < nt ? 4 * q + k : nt - 1, idx = T::term(l, t), tap = idx % 25, gid = idx / 25;
                    return (gid * L::PLANE + L::tap_off(tap / 5, tap % 5)) * 4;
                };
                constexpr int o0 = off_of(0), o1 = off_of(1), o2 = off_of(2), o3 = off_of(3);
                int voff = kq == 1 ? o1 : o0;
                voff = kq == 2 ? o2 : voff;
                voff = kq == 3 ? o3 : voff;
                const float *bp = (const float *)((const char *)xs + voff) + j;
                float bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = bp[t * L::TOFF];
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][aa] = i144_mfma(av, bv[t], q == 0 ? zero4 : acc[t][aa]);
                __builtin_amdgcn_sched_barrier(0);
            });
            // ---- tree levels +64, +32, +16, +8 in registers (lane slots aa = l >> 3), then one tile per wave and position tile to LDS
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 lv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) lv[i] = acc[t][i] + acc[t][i + 8];            // p[i] + p[i + 64]
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = lv[i] + lv[i + 4];                    // + 32
#pragma unroll
                for (int i = 0; i < 2; ++i) lv[i] = lv[i] + lv[i + 2];                    // + 16
                const f32x4 r = lv[0] + lv[1];                                            // + 8
#pragma unroll
                for (int v = 0; v < 4; ++v) comb[((parity * NT + t) * 8 + WAVE) * 256 + v * 64 + lane] = r[v];
            }
            __syncthreads();
            // ---- last three levels (+4, +2, +1 over the waves' residues c), bias, PReLU, residual, store
            for (int e = tid; e < NT * 256; e += I144_THREADS) {
                const int t = e >> 8, rem = e & 255, v = rem >> 6, ln = rem & 63;
                const float *cb = comb + ((parity * NT + t) * 8) * 256 + v * 64 + ln;
                const float r0_ = cb[0], r1 = cb[256], r2 = cb[512], r3 = cb[768], r4 = cb[1024], r5 = cb[1280], r6 = cb[1536], r7 = cb[1792];
                float sv = ((r0_ + r4) + (r2 + r6)) + ((r1 + r5) + (r3 + r7));
                const int o = ot * 16 + 4 * (ln >> 4) + v, jj = ln & 15;
                bool ok = o < a.nout;
                long oi;
                if constexpr (DC) {
                    const int th = th0 + 16 * t + jj;
                    ok = ok && th >= a.th_lo && th <= a.th_hi;
                    oi = (long)n * a.osample + (long)o * a.oplane + (long)(a.s + I144_R0) * a.opitch + th + I144_C0;
                } else {
                    const int y = r0 + t, xx = c0 + jj;
                    ok = ok && y < a.H && xx < a.W;
                    oi = (long)n * a.osample + (long)o * a.oplane + (long)(y + a.ooff) * a.opitch + xx + a.ooff;
                }
                if (ok) {
                    sv = sv + a.bias[o];
                    if (a.act) sv = DC ? (sv < 0 ? sv * a.act[o] : sv) : (sv > 0 ? sv : sv * a.act[o]);   // cconv_dc_cuda.cu:360-362 / cconv_ec_cuda.cu:311-312
                    if (a.residual) sv = sv + a.residual[oi];
                    a.out[oi] = sv;
                }
            }
            parity ^= 1;
        }
    }
}

template <int HIDDEN, bool DC, int NT>
__global__ __launch_bounds__(I144_THREADS, 2) void k_cconv144(I144Args a) {
    using L = ILay<DC, NT>;
    __shared__ float xs[144 * L::PLANE + 256];                             // + slack of the last DMA window
    __shared__ float comb[2 * NT * 8 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    switch (wave) {
        case 0: i144_body<HIDDEN, DC, NT, 0>(a, xs, comb, tid, lane); break;
        case 1: i144_body<HIDDEN, DC, NT, 1>(a, xs, comb, tid, lane); break;
        case 2: i144_body<HIDDEN, DC, NT, 2>(a, xs, comb, tid, lane); break;
        case 3: i144_body<HIDDEN, DC, NT, 3>(a, xs, comb, tid, lane); break;
        case 4: i144_body<HIDDEN, DC, NT, 4>(a, xs, comb, tid, lane); break;
        case 5: i144_body<HIDDEN, DC, NT, 5>(a, xs, comb, tid, lane); break;
        case 6: i144_body<HIDDEN, DC, NT, 6>(a, xs, comb, tid, lane); break;
        default: i144_body<HIDDEN, DC, NT, 7>(a, xs, comb, tid, lane); break;
    }
}

#define I144_NT_EC 2
#define I144_NT_DC 2

// Encode order.  x: zero-haloed NCHW planes [n][144][hp][wp], cell (r, c) at [(r+2)*wp + c+2] with hp >= 2*ceil(h/2) + 4 and
// wp >= 16*ceil(w/16) + 4 (lic360_ec144_layout); out / residual: [n][nout] planes of `opitch`-float rows, cell (r, c) at
// [(r+ooff)*opitch + c+ooff] with plane stride oplane (ooff = 2: the same haloed layout; ooff = 0: plain NCHW).
LIC360_API int lic360_ec144_layout(int h, int w, int *hp, int *wp) {
    ARG_CHECK(hp && wp && h > 0 && w > 0);
    *hp = (h + I144_NT_EC - 1) / I144_NT_EC * I144_NT_EC + 4;
    *wp = (w + 15) / 16 * 16 + 4;
    return 0;
}
LIC360_API int lic360_cconv144_ec(void *stream, const lic360_conv_plan *p, const float *x, const float *packed144, const float *bias,
                                  const float *act, const float *residual, float *out, int n, int h, int w, long oplane, int opitch, int ooff) {
    ARG_CHECK(p && conv144_ok(p) && x && packed144 && bias && out && n > 0 && h > 0 && w > 0 && oplane > 0 && opitch > 0 && (ooff == 0 || ooff == 2));
    I144Args a;
    int hp, wp;
    if (lic360_ec144_layout(h, w, &hp, &wp)) return 2;
    a.x = x; a.packed = packed144; a.bias = bias; a.act = act; a.residual = residual; a.out = out;
    a.N = n; a.H = h; a.W = w; a.nout = p->nout; a.n_ot = conv144_otiles(p);
    a.xplane = (long)hp * wp; a.xsample = 144 * a.xplane; a.xpitch = wp;
    a.oplane = oplane; a.osample = (long)p->nout * oplane; a.opitch = opitch; a.ooff = ooff;
    a.tiles_r = (h + I144_NT_EC - 1) / I144_NT_EC; a.tiles_c = (w + 15) / 16;
    a.s = a.th_lo = a.th_hi = a.th0 = 0; a.og = a.n_ot; a.n_og = 1; a.n_seg = 1;
    const long ntasks = (long)n * a.tiles_r * a.tiles_c;
    const dim3 grid((unsigned)(ntasks < 256 ? ntasks : 256));
    a.grid = (int)grid.x;
    hipLaunchKernelGGL((k_cconv144<1, false, I144_NT_EC>), grid, dim3(I144_THREADS), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return 0;
}

// Decode order, plane (= anti-diagonal) s of every map: x / out / residual are zero-padded diagonal-major planes
// [n][c][h + w - 1 + 8][h + 4] (lic360_dc144_layout): cell (th, tw) at row th + tw + 4, column th + 2.
LIC360_API int lic360_dc144_layout(int h, int w, int *rows, int *pitch) {
    ARG_CHECK(rows && pitch && h > 0 && w > 0);
    *rows = h + w - 1 + 2 * I144_R0;
    *pitch = (h + 2 * I144_C0 + 16 * I144_NT_DC + 3) / 4 * 4;               // a task's 36-column window stays inside the row
    return 0;
}
LIC360_API int lic360_cconv144_dc_plane(void *stream, const lic360_conv_plan *p, const float *x, const float *packed144, const float *bias,
                                        const float *act, const float *residual, float *out, int n, int h, int w, int s) {
    ARG_CHECK(p && conv144_ok(p) && x && packed144 && bias && out && n > 0 && h > 0 && w > 0);
    if (s < 0 || s >= h + w - 1) return 0;
    I144Args a;
    int rows, pitch;
    if (lic360_dc144_layout(h, w, &rows, &pitch)) return 2;
    a.x = x; a.packed = packed144; a.bias = bias; a.act = act; a.residual = residual; a.out = out;
    a.N = n; a.H = h; a.W = w; a.nout = p->nout; a.n_ot = conv144_otiles(p);
    a.xplane = (long)rows * pitch; a.xsample = 144 * a.xplane; a.xpitch = pitch;
    a.oplane = a.xplane; a.osample = (long)p->nout * a.xplane; a.opitch = pitch; a.ooff = 0;
    a.tiles_r = a.tiles_c = 1;
    a.s = s; a.th_lo = s >= w ? s - w + 1 : 0; a.th_hi = s < h ? s : h - 1; a.th0 = a.th_lo & ~3;
    // split a map's output tiles over enough workgroups to occupy the chip (each re-stages the map's 104 KB x tile)
    a.n_seg = (a.th_hi - a.th0) / (16 * I144_NT_DC) + 1;
    a.og = a.n_ot;
    while (a.og > 1 && (long)n * a.n_seg * ((a.n_ot + a.og - 1) / a.og) < 192) --a.og;
    a.n_og = (a.n_ot + a.og - 1) / a.og;
    const long ntasks = (long)n * a.n_og * a.n_seg;
    const dim3 grid((unsigned)(ntasks < 256 ? ntasks : 256));
    a.grid = (int)grid.x;
    hipLaunchKernelGGL((k_cconv144<1, true, I144_NT_DC>), grid, dim3(I144_THREADS), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return 0;
}
