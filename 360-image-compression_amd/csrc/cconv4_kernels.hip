// cconv4_kernels.hip -- "leaf-resident" group-causal masked convolution for the latent entropy nets
// (A9/A10 for the shapes test/lic360_demo.py:104-112 uses: ngroup groups, cin in {1,4}, cout in {3,4}).
//
// Same arithmetic contract as cconv_kernels.hip (extension/cconv_ec_cuda.cu:268-315: one fmaf chain per
// virtual lane over ti = gid, gid+cin, ...; then the fixed 128-leaf tree), different mapping:
//
//   * 64 output positions of ONE output group g (its <=4 output channels) are owned by FOUR waves; wave c
//     keeps the partial sums of the virtual lanes l = c (mod 4) resident in registers, one 4x4x1 MFMA
//     accumulator per lane (4 output channels x this thread's position).  Because 25 = 1 (mod 4), for
//     cin = 4 every wave gets exactly one lane per tap (gid = (c - tap) mod 4): 25 accumulators, 100 VGPRs;
//   * the K loop runs over the input GROUP index tc (outermost), so a step needs only the cin channels of
//     group tc: they are staged in LDS once per step and shared by all taps, rows and lane classes; the
//     chain of lane (gid,kh,kw) receives exactly its own terms tc = 0..L-1 in order, i.e. the reference's
//     fmaf chain with NO padding work (v_mfma_f32_4x4x1_16b_f32 has K = 1: D = fma(A, B, C), measured
//     bit-exact on gfx950, tools/mfma_probe.hip);
//   * the 4 weights of a lane (one per output channel) sit in lanes 0-3 of the A operand and are broadcast
//     to all 16 4x4 blocks with cbsz = 4; one ds_read_b128 fetches them for 4 taps;
//   * causality removes whole taps per step (lane active iff kh+kw < g+4+hidden-tc): one scalar branch per tap;
//   * lanes of equal index mod 4 stay together until the last two levels of the reference's tree
//     (p[i]+p[i+64], +32, ..., +4 happen inside a class), so each wave reduces its own 25 accumulators in
//     registers and only ONE tile per wave crosses LDS for the final (F0+F2)+(F1+F3).
//
// Encode order (EC4): workgroup = 12 waves = 3 consecutive rows x 64 columns of one (sample, group); the
// 7 x 68 halo tile of the step's cin channels and the step's 2 KB of weights are double-buffered in LDS.
// Decode order (DC4, cconv4v3_dc.inc): persistent workgroups of 12 waves = 3 adjacent groups (adjacent
// anti-diagonals of the current plane) of one sample; lanes run along the diagonal in a zero-padded diagonal-major
// activation layout, so the 11 x 72 halo band shared by the 3 diagonals is fetched with unconditional 16-byte
// LDS-DMA loads and the output stores are contiguous.
#include "common.h"
#include "conv_plan.h"

#include "cconv_tree.h"

#define C4_COLS 68                       // 64 positions + 2*2 halo
#define C4_WSLOTS 128                    // weight slots per step (>= 25*cin), 4 floats each

static inline bool conv4_ok(const lic360_conv_plan *p) {
    return p->ksz == 5 && (p->cin == 1 || p->cin == 4) && p->cout >= 1 && p->cout <= 4 && p->ngroup <= 64;
}

// ------------------------------------------------------------------------------------------------
// packed4[net][g][tc][c][leaf][r]: class c = virtual lane mod 4, leaf i (0..31) inside the class
// (cin = 4: tap = i, gid = (c - tap) mod 4;  cin = 1: tap = c + 4*i), r = output channel within the group.
// A wave reads its class as two 64-float registers: lane (i%16)*4 + r of register i/16 holds the weight of leaf i, row r --
// exactly the 4-lane block that `abid = i%16` selects for broadcast (cbsz = 4) in v_mfma_f32_4x4x1_16b_f32.
// QUAD part (read by the v6 kernels, cconv4v6_dc.inc): the same weights with the registers a wave needs for one DOUBLE step adjacent per lane, so
// that it fetches them with one (cin = 4) or two (cin = 1) 16-byte loads instead of four / eight 4-byte ones -- the CU's address unit takes a wave
// instruction at a time whatever its width, and the decode kernel's twelve waves kept it busy 864 of a double step's ~2900 cycles:
//   quads[net][g][blk][c][lane][4]   cin = 4: blk = tc / 2, word = 2 (tc & 1) + register;   cin = 1: blk = tc / 4, word = tc & 3 (register 0 only)
// wblk 4 KB blocks per output group (zero-filled past the last input group).
__global__ void k_conv4_pack(const float *__restrict__ weight, float *__restrict__ packed, float *__restrict__ quads, int wblk, int nb, int ngroup, int cin,
                             int cout, int hidden) {
    const long per_net = (long)ngroup * ngroup * C4_WSLOTS * 4, total = per_net * nb;
    const int C = ngroup * cin, nout = ngroup * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int r = (int)(i & 3), leaf = (int)((i >> 2) & 31), c = (int)((i >> 7) & 3);
        long t = i / (C4_WSLOTS * 4);
        int tc = (int)(t % ngroup), g = (int)((t / ngroup) % ngroup), b = (int)(t / ((long)ngroup * ngroup));
        int tap = cin == 4 ? leaf : c + 4 * leaf, gid = cin == 4 ? ((c - tap) & 3) : 0;
        float v = 0.0f;
        if (tap < 25 && r < cout) {
            int kh = tap / 5, kw = tap % 5;
            int L = g + 4 - kh - kw + hidden;               // chain length of this lane (extension/cconv_ec_cuda.cu:288-290)
            if (L > ngroup) L = ngroup;
            if (tc < L) v = weight[(((long)b * nout + g * cout + r) * C + tc * cin + gid) * 25 + tap];
        }
        packed[i] = v;
        const long qb = (long)(b * ngroup + g) * wblk;
        const int ln = (leaf & 15) * 4 + r;
        if (cin == 4) quads[(qb + (tc >> 1)) * 1024 + c * 256 + ln * 4 + (tc & 1) * 2 + (leaf >> 4)] = v;
        else if (leaf < 16) quads[(qb + (tc >> 2)) * 1024 + c * 256 + ln * 4 + (tc & 3)] = v;
    }
}

LIC360_API int lic360_conv4_supported(const lic360_conv_plan *p) { return p && conv4_ok(p) ? 1 : 0; }
// a packed4 buffer of nb nets = [nb x slot part (ngroup^2 x 512 floats per net)][nb x quad part (ngroup x wblk x 1024 floats per net)]
static inline long conv4_slot_floats(const lic360_conv_plan *p) { return (long)p->ngroup * p->ngroup * C4_WSLOTS * 4; }
static inline int conv4_wblk(const lic360_conv_plan *p) { return p->cin == 4 ? (p->ngroup + 1) / 2 : 2 * ((p->ngroup + 7) / 8); }
static inline long conv4_quad_floats(const lic360_conv_plan *p) { return (long)p->ngroup * conv4_wblk(p) * 1024; }
LIC360_API long lic360_conv4_packed_floats(const lic360_conv_plan *p) {
    return p && conv4_ok(p) ? conv4_slot_floats(p) + conv4_quad_floats(p) : 0;
}
LIC360_API int lic360_conv4_pack(void *stream, const lic360_conv_plan *p, const float *weight, int nb, float *packed) {
    ARG_CHECK(p && conv4_ok(p) && weight && packed && nb > 0);
    const long total = conv4_slot_floats(p) * nb;
    float *quads = packed + total;
    HIP_TRY(hipMemsetAsync(quads, 0, (size_t)conv4_quad_floats(p) * nb * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(k_conv4_pack, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, weight, packed, quads, conv4_wblk(p), nb, p->ngroup,
                       p->cin, p->cout, p->constrain == 5 ? 0 : 1);
    LAUNCH_CHECK();
    return 0;
}

// PART 0 = the first chunk of 8 accumulators, PART 1 = the rest, PART 2 = everything.  The persistent kernels run part 0,
// then stage the NEXT step (global loads, ds_writes into the other LDS half), then part 1: the staging instructions
// issue in the shadow of the first chunk's MFMAs instead of in front of them.
template <int CIN, int CLS, int XPLANE, int COLS, bool DIAG, bool FULL, int PART = 2>
__device__ __forceinline__ void conv4_step(f32x4 *acc, const float *xs, const f32x4 *ws4, int dlim, int lane, int xbase) {
    constexpr int NA = NAcc<CIN>::value;
    static_assert(FULL, "every step is a full straight-line step (zero weights past a lane's chain end)");
    (void)dlim;
    // the class's weights for this step: 2 registers of 16 leaves x 4 rows (2 ds_read_b32 instead of 7 ds_read_b128)
    const float *wl = (const float *)ws4 + CLS * 128 + lane;
    const float w0 = wl[0], w1 = (NA > 16 && PART != 0) ? wl[64] : 0.0f;
    const float *xl = xs + xbase;
    // register-bounded chunks of 8 lanes: the LDS reads of a chunk are issued together, then its MFMAs
    constexpr int CH = 8;
    static_for<(NA + CH - 1) / CH>([&](auto cc) {
        constexpr int c0 = decltype(cc)::value * CH;
        if constexpr (PART == 2 || (PART == 0) == (c0 == 0)) {
            float bv[CH];
            static_for<CH>([&](auto kk) {
                constexpr int k = decltype(kk)::value, i = c0 + k, tap = CIN == 4 ? i : CLS + 4 * i;
                constexpr int kh = tap / 5, kw = tap % 5, gid = CIN == 4 ? ((CLS - tap) & 3) : 0;
                if constexpr (i < NA && tap < 25) bv[k] = xl[gid * XPLANE + (DIAG ? (kh + kw) * COLS + kh : kh * COLS + kw)];
            });
            static_for<CH>([&](auto kk) {
                constexpr int k = decltype(kk)::value, i = c0 + k, tap = CIN == 4 ? i : CLS + 4 * i;
                if constexpr (i < NA && tap < 25) acc[i] = mfma4<i % 16>(i < 16 ? w0 : w1, bv[k], acc[i]);
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}

#define C4_PS 3                          // position sets (rows / diagonals) per workgroup
#define C4_THREADS (C4_PS * 4 * 64)      // 12 waves: 3 per SIMD -> 168 VGPRs each, no spills with 25 resident accumulators
// final two tree levels across the 4 lane classes of a position set + epilogue handled by the caller
#define C4_COMB_FLOATS (C4_PS * 4 * 4 * 64)

// ------------------------------------------------------------------------------------------------ EC4
#define EC4_ROWS (C4_PS + 4)              // output rows + 2*2 halo
template <int CIN, int CLS>
__device__ __forceinline__ f32x4 ec4_body(const float *__restrict__ xn, const f32x4 *__restrict__ wp, float (*xs)[CIN * EC4_ROWS * C4_COLS],
                                          f32x4 (*ws4)[C4_WSLOTS], int tid, int lane, int ps, int g, int hidden, int Lmax,
                                          int r0, int c0, int H, int W) {
    constexpr int XPLANE = EC4_ROWS * C4_COLS, XS = CIN * XPLANE;
    constexpr int XLD = (XS + C4_THREADS - 1) / C4_THREADS;
    f32x4 acc[NAcc<CIN>::value];
#pragma unroll
    for (int i = 0; i < NAcc<CIN>::value; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float xr[XLD];
    f32x4 wr = {0.f, 0.f, 0.f, 0.f};
    // staging addresses are step-invariant up to + tc*CIN*H*W: resolve the index math once (-1 = outside the image)
    int xo[XLD];
#pragma unroll
    for (int k = 0; k < XLD; ++k) {
        int e = tid + k * C4_THREADS;
        xo[k] = -1;
        if (e < XS) {
            int gid = e / XPLANE, q = e % XPLANE, rr = q / C4_COLS, cc = q % C4_COLS;
            int gr = r0 - 2 + rr, gc = c0 - 2 + cc;
            if (gr >= 0 && gr < H && gc >= 0 && gc < W) xo[k] = (gid * H + gr) * W + gc;
        }
    }
    const int tcs = CIN * H * W;
    auto gload = [&](int tc) {
        const float *xt = xn + (long)tc * tcs;
#pragma unroll
        for (int k = 0; k < XLD; ++k) xr[k] = xo[k] >= 0 ? xt[xo[k]] : 0.0f;
        if (tid < C4_WSLOTS) wr = wp[(long)tc * C4_WSLOTS + tid];
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int k = 0; k < XLD; ++k) {
            int e = tid + k * C4_THREADS;
            if (e < XS) xs[buf][e] = xr[k];
        }
        if (tid < C4_WSLOTS) ws4[buf][tid] = wr;
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int xbase = ps * C4_COLS + lane;
    // Lanes whose chain has ended (kh+kw >= g+4+hidden-tc) carry a zero weight in packed4, so every step is the same
    // straight-line block of 25 MFMAs: fma(0, x, acc) == acc leaves their sums untouched.
    for (int tc = 0; tc < Lmax; ++tc) {
        const int cur = tc & 1;
        if (tc + 1 < Lmax) gload(tc + 1);
        conv4_step<CIN, CLS, XPLANE, C4_COLS, false, true>(acc, xs[cur], ws4[cur], 9, lane, xbase);
        if (tc + 1 < Lmax) lstore(cur ^ 1);
        __syncthreads();
    }
    return tree4_eval<CIN, CLS>(acc);
}

template <int CIN>
__global__ __launch_bounds__(C4_THREADS) void k_cconv4_ec(
    const float *__restrict__ x, const float *__restrict__ packed, const float *__restrict__ bias, const float *__restrict__ act,
    const float *__restrict__ residual, float *__restrict__ out, int ngroup, int cout, int hidden, int H, int W, int npb, int x_mod) {
    constexpr int XS = CIN * EC4_ROWS * C4_COLS;
    __shared__ float xs[2][XS];
    __shared__ f32x4 ws4[2][C4_WSLOTS];
    __shared__ float comb[C4_COMB_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), ps = wave >> 2, cls = wave & 3;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * C4_PS;
    const int g = blockIdx.z % ngroup, n = blockIdx.z / ngroup, nbatch = n / npb;
    const int C = ngroup * CIN, nout = ngroup * cout;
    int Lmax = g + 4 + hidden;
    if (Lmax > ngroup) Lmax = ngroup;
    const float *xn = x + (long)(n % x_mod) * C * H * W;
    const f32x4 *wp = (const f32x4 *)(packed + (((long)nbatch * ngroup + g) * ngroup) * C4_WSLOTS * 4);
    f32x4 part;
    // every wave runs the same number of barriers; the four lane classes differ only in compile-time offsets
    switch (cls) {
        case 0: part = ec4_body<CIN, 0>(xn, wp, xs, ws4, tid, lane, ps, g, hidden, Lmax, r0, c0, H, W); break;
        case 1: part = ec4_body<CIN, 1>(xn, wp, xs, ws4, tid, lane, ps, g, hidden, Lmax, r0, c0, H, W); break;
        case 2: part = ec4_body<CIN, 2>(xn, wp, xs, ws4, tid, lane, ps, g, hidden, Lmax, r0, c0, H, W); break;
        default: part = ec4_body<CIN, 3>(xn, wp, xs, ws4, tid, lane, ps, g, hidden, Lmax, r0, c0, H, W); break;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) comb[((ps * 4 + cls) * 4 + r) * 64 + lane] = part[r];
    __syncthreads();
    // wave (ps, cls) finishes output channel r = cls of its position set: (F0 + F2) + (F1 + F3)
    const int r = cls;
    const int gr = r0 + ps, gc = c0 + lane;
    if (r < cout && gr < H && gc < W) {
        const float f0 = comb[((ps * 4 + 0) * 4 + r) * 64 + lane], f1 = comb[((ps * 4 + 1) * 4 + r) * 64 + lane];
        const float f2 = comb[((ps * 4 + 2) * 4 + r) * 64 + lane], f3 = comb[((ps * 4 + 3) * 4 + r) * 64 + lane];
        const int o = g * cout + r, bid = nbatch * nout + o;
        float sv = ((f0 + f2) + (f1 + f3)) + bias[bid];
        if (act) sv = sv > 0 ? sv : sv * act[bid];                      // cconv_ec_cuda.cu:311-312
        const long oi = (((long)n * nout + o) * H + gr) * W + gc;
        if (residual) sv = sv + residual[oi];
        out[oi] = sv;
    }
}

LIC360_API int lic360_cconv4_ec(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod) {
    ARG_CHECK(p && conv4_ok(p) && x && packed4 && bias && out && n > 0 && h > 0 && w > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n);
    ARG_CHECK((long)n * p->ngroup < 65536);
    dim3 grid((w + 63) / 64, (h + C4_PS - 1) / C4_PS, n * p->ngroup);
    const int hidden = p->constrain == 5 ? 0 : 1;
    if (p->cin == 4)
        hipLaunchKernelGGL(k_cconv4_ec<4>, grid, dim3(C4_THREADS), 0, (hipStream_t)stream, x, packed4, bias, act, residual, out, p->ngroup,
                           p->cout, hidden, h, w, n / nb, x_mod);
    else
        hipLaunchKernelGGL(k_cconv4_ec<1>, grid, dim3(C4_THREADS), 0, (hipStream_t)stream, x, packed4, bias, act, residual, out, p->ngroup,
                           p->cout, hidden, h, w, n / nb, x_mod);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ DC4
#include "cconv4v3_dc.inc"
#include "cconv4v6_dc.inc"
#include "cconv4v3_ec.inc"

// mode: bit 0 = the LDS-DMA kernel of the previous generation (A/B runs), bit 1 = no two-samples-per-wave packing.
// Internal entry (hidden visibility) so that the fused codec passes the switches it read ONCE at create time.
int lic360_cconv4_dc_plane_mode(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod, int mode) {
    ARG_CHECK(p && conv4_ok(p) && x && packed4 && bias && out && n > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n);
    if (psum < 0 || psum >= h + w + p->ngroup - 2) return 0;
    if ((mode & 1) && (h <= 64 || w <= 64)) return launch_cconv4v3_dc((hipStream_t)stream, p, x, packed4, bias, act, residual, out, n, h, w, nb, psum, x_mod);
    return launch_cconv4v6_dc((hipStream_t)stream, p, x, packed4, bias, act, residual, out, n, h, w, nb, psum, x_mod, (mode & 2) != 0,
                              (mode >> 2) & 3);
}
// LIC360_DC4=3 / LIC360_NOPACK select the A/B variants; the environment is read once per process, not per launch
int lic360_dc4_env_mode(void) {
    const char *v = getenv("LIC360_DC4");
    const char *gsm = getenv("LIC360_DC_GSTEP");                     // "1": one group per task always, "3": never (default: by task count)
    return ((v && v[0] == '3') ? 1 : 0) | (getenv("LIC360_NOPACK") ? 2 : 0) | (gsm && gsm[0] == '1' ? 4 : (gsm && gsm[0] == '3' ? 8 : 0));
}
LIC360_API int lic360_cconv4_dc_plane(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                      const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod) {
    static const int mode = lic360_dc4_env_mode();
    return lic360_cconv4_dc_plane_mode(stream, p, x, packed4, bias, act, residual, out, n, h, w, nb, psum, x_mod, mode);
}

// Floats to allocate for `planes` activation planes (samples x channels) of one of this file's padded layouts: the planes plus
// the slack the band fetches of the last plane may touch (they are whole 16-byte quads of 11-row bands and are not clamped at
// the end of the tensor).  Callers size and zero buffers with this instead of knowing the slack.
//   layout: 0 = decode order (lic360_dc4_layout), 1 = encode order NCHW (lic360_ec4_layout), 2 = encode order wrapped diagonals
//   (lic360_ec6_layout)
#define C4_TAIL_FLOATS 4096
LIC360_API long lic360_conv4_buffer_floats(int layout, long planes, int h, int w) {
    if (planes <= 0 || h <= 0 || w <= 0) return 0;
    long per = 0;
    if (layout == 0) per = (long)D3_SP(h, w) * D3_HP(h);
    else if (layout == 1) per = (long)E3_HP(h) * E3_WP(w);
    else if (layout == 2 && w >= 7) per = (long)E6_ROWS(w) * D3_HP(h);
    return per ? planes * per + C4_TAIL_FLOATS : 0;
}
LIC360_API int lic360_dc4_layout(int h, int w, int *rows, int *pitch, int *row0, int *col0) {
    ARG_CHECK(rows && pitch && row0 && col0 && h > 0 && w > 0);
    *rows = D3_SP(h, w); *pitch = D3_HP(h); *row0 = D3_S0; *col0 = D3_C0;
    return 0;
}

// encode order on zero-padded activations [n][c][E3_HP(h)][E3_WP(w)] (cell (r, c) at [(r+2)*WP + c+2]); used by the fused codec
LIC360_API int lic360_ec4_layout(int h, int w, int *hp, int *wp) {
    ARG_CHECK(hp && wp && h > 0 && w > 0);
    *hp = E3_HP(h); *wp = E3_WP(w);
    return 0;
}
// encode order on the wrapped diagonal-major layout (cconv4v6_dc.inc): planes of `rows` x `pitch` floats, cell (th, tw) on
// wrapped diagonal sg = (th + tw + 2) % wpp at [(sg + row0) * pitch + th + 2]; diagonals < rows - wpp - row0 are stored a
// second time wpp rows further down and diagonals >= wpp - row0 a second time wpp rows further up.
LIC360_API int lic360_ec6_layout(int h, int w, int *rows, int *pitch, int *row0, int *wpp) {
    ARG_CHECK(h > 0 && w >= 7 && rows && pitch && row0 && wpp);                        // the 11-row band must not overlap itself
    *rows = E6_ROWS(w); *pitch = D3_HP(h); *row0 = E6_R0; *wpp = E6_WPP(w);
    return 0;
}
LIC360_API int lic360_cconv4_ec_diag(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                     const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod) {
    ARG_CHECK(p && conv4_ok(p) && x && packed4 && bias && out && n > 0 && nb > 0 && n % nb == 0 && h > 0 && w >= 7 && x_mod > 0);
    return launch_cconv4v6_ec((hipStream_t)stream, p, x, packed4, bias, act, residual, out, n, h, w, nb, x_mod);
}
LIC360_API int lic360_cconv4_ec_padded(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                       const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod) {
    ARG_CHECK(p && conv4_ok(p) && x && packed4 && bias && out && n > 0 && h > 0 && w > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n);
    return launch_cconv4v3_ec((hipStream_t)stream, p, x, packed4, bias, act, residual, out, n, h, w, nb, x_mod);
}
