// cconv_kernels.hip -- group-causal masked 5x5 convolution (A9 cconv_ec, A10 cconv_dc) for gfx950.
//
// Reference: one 128-thread block per output scalar, shared-memory tree reduce
// (extension/cconv_ec_cuda.cu:268-315, extension/cconv_dc_cuda.cu:313-364).
//
// Here: one wave computes a 16-output-channel x (NT*16)-position tile with
// v_mfma_f32_16x16x4_f32.  On gfx950 that instruction is bit-for-bit a k-ordered fmaf chain
// (one rounding per product, subnormals kept), so the reference's per-lane chain
//     sum = fmaf(x[ti], w[ti], sum),  ti = gid, gid+cin, ...
// becomes the K loop of the MFMA, one accumulator tile per virtual lane, and the reference's
// fixed 128-leaf tree (p[i]+p[i+64]; +32; ...; +1) becomes a binary-counter merge over lanes
// visited in bit-reversed order (conv_plan.cpp).  A/B operands:
//     A[i][k] = packed weight of output row o0+i for term k   (global, coalesced 256 B / K-step,
//               software-prefetched 3 K-steps ahead; shared by every position tile -> L2 resident)
//     B[k][j] = x[ti_k][row + kh_k][col + kw_k + j]            (EC: LDS tile; DC: global gather)
// Encode (EC): workgroup = 4 x 16 output positions x ALL output channels of one sample; the
// zero-padded input tile (all C channels x 8 x 20, plane stride padded to 164 floats so the four
// K-slices of a B read land on disjoint bank halves) is staged once in LDS (<=126 KB) and reused by
// every output tile and every tap.  Waves pull 16-channel output tiles from an LDS counter in
// descending work order (causality makes late groups ~10x more expensive than early ones).
// Decode (DC): one wave per (output tile, 16 plane positions) task; operands gathered from global.
#include "common.h"
#include "conv_plan.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define EC_TH 4            // output rows per workgroup (= N-tiles per wave)
#define EC_TW 16           // output cols per workgroup (= MFMA N)
#define EC_ROWS (EC_TH + 4)
#define EC_COLS (EC_TW + 4)
#define EC_PL 164          // padded plane stride (floats): 4*164 = 656 = 16 (mod 32)

// ------------------------------------------------------------------------------------------------
// weight packing: packed[b][i] = wsrc[i] >= 0 ? weight[b][wsrc[i]] : 0
__global__ void k_conv_pack(const float *__restrict__ weight, const int *__restrict__ wsrc, float *__restrict__ packed,
                            long nreal, long nper, long wstride, int nb) {
    long total = nper * nb;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long b = i / nper, j = i % nper;
        int s = j < nreal ? wsrc[j] : -1;
        packed[i] = s >= 0 ? weight[b * wstride + s] : 0.0f;
    }
}

LIC360_API int lic360_conv_pack(void *stream, const lic360_conv_plan *p, const float *weight, int nb, float *packed) {
    ARG_CHECK(p && weight && packed && nb > 0);
    long nper = lic360_conv_plan_packed_floats(p);
    long nreal = p->total_rec * 64;
    long wstride = (long)p->nout * p->C * p->ksz * p->ksz;
    hipLaunchKernelGGL(k_conv_pack, dim3(lic360_blocks(nper * nb, 4)), dim3(256), 0, (hipStream_t)stream, weight, p->d_wsrc, packed, nreal, nper, wstride, nb);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The reference's 128-leaf tree as a binary-counter merge over the visiting index r (bit-reversed
// lane order): finishing leaf r adds one stacked partial tile per trailing 1-bit of r, then stacks
// the sum.  Levels 0-3 (hot: one merge per leaf on average) live in registers; levels 4-6 (7 pushes
// and 7 pops per output tile in total) live in a small private-memory array so that the kernel
// fits 3 waves per SIMD.
#define TREE_DECL(NT) f32x4 cur[NT], s0[NT], s1[NT], s2[NT], s3[NT]; volatile float hi[3][NT * 4]
#define TREE_LEVEL(NT, L, S)                                                         \
    if (!((r >> L) & 1)) {                                                           \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) S[j] = cur[j];                \
        break;                                                                       \
    }                                                                                \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) cur[j] = S[j] + cur[j];
#define TREE_MERGE(NT)                                                               \
    do {                                                                             \
        TREE_LEVEL(NT, 0, s0) TREE_LEVEL(NT, 1, s1) TREE_LEVEL(NT, 2, s2) TREE_LEVEL(NT, 3, s3) \
        int lev = 0;                                                                 \
        for (int blk = r >> 4; blk & 1; blk >>= 1, ++lev) {                          \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) cur[j][e] = hi[lev][j * 4 + e] + cur[j][e]; \
        }                                                                            \
        if (lev < 3) {                                                               \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) hi[lev][j * 4 + e] = cur[j][e]; \
        }                                                                            \
    } while (0)

// ------------------------------------------------------------------------------------------------
// EC kernel: 768 threads = 12 waves (3 per SIMD: while one wave waits for its next A fragment the
// other two keep the SIMD's matrix pipe busy; one K-step is prefetched per wave on top of that).
#define EC_THREADS 768
#define EC_WAVES (EC_THREADS / 64)
template <int NT>
__global__ __launch_bounds__(EC_THREADS) void k_cconv_ec(
    const float *__restrict__ x, const float *__restrict__ packed, const float *__restrict__ bias, const float *__restrict__ act,
    const float *__restrict__ residual, float *__restrict__ out, const int *__restrict__ mt_rec_start, const int *__restrict__ leaf_cnt,
    const int *__restrict__ term, int C, int H, int W, int nout, int n_mtiles, int npb, long packed_per_net, int x_mod) {
    extern __shared__ float xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * EC_TW, r0 = blockIdx.y * EC_TH, n = blockIdx.z;
    const int nbatch = n / npb;
    // ---- stage the zero-padded input tile: xs[ch][rr][cc] = x[n][ch][r0-2+rr][c0-2+cc]
    {
        const float *xn = x + (long)(n % x_mod) * C * H * W;       // x_mod < N: the stacked nets share one input
        const int per = EC_ROWS * EC_COLS;
        for (int e = tid; e < C * per; e += EC_THREADS) {
            int ch = e / per, q = e % per, rr = q / EC_COLS, cc = q % EC_COLS;
            int gr = r0 - 2 + rr, gc = c0 - 2 + cc;
            float v = 0.0f;
            if (gr >= 0 && gr < H && gc >= 0 && gc < W) v = xn[((long)ch * H + gr) * W + gc];
            xs[ch * EC_PL + q] = v;
        }
    }
    __syncthreads();
    const float *wp = packed + (long)nbatch * packed_per_net;
    const int kq = lane >> 4, col = lane & 15;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);       // wave-uniform -> scalar control flow
    // Output tiles are dealt heaviest-first (late groups have the longest chains) in snake order over
    // the 4 SIMDs (waves w, w+4, w+8 share a SIMD), so every SIMD gets about the same MFMA count.
    for (int slot = wave_s; slot < n_mtiles; slot += EC_WAVES) {
        const int round = slot >> 2, pos = slot & 3;
        int rank = (round & 1) ? (round * 4 + 3 - pos) : slot;     // rank in descending-work order
        if (rank >= n_mtiles) rank = slot;                          // ragged last round
        const int mi = n_mtiles - 1 - rank;
        int rec = mt_rec_start[mi];
        const int *cnt = leaf_cnt + mi * 128;
        TREE_DECL(NT);
        // pipeline: A fragment 1 K-step ahead, term word 2 ahead, B (LDS) operands 1 ahead
        const float *wl = wp + lane;
        const int *tl = term + kq;
        float a_cur = wl[rec * 64];
        int t_nxt = tl[(rec + 1) * 4];
        float b_cur[NT];
        {
            int t0 = tl[rec * 4];
            int off = (t0 & 0xffff) * EC_PL + ((t0 >> 16) & 0xff) * EC_COLS + ((t0 >> 24) & 0xff) + col;
#pragma unroll
            for (int j = 0; j < NT; ++j) b_cur[j] = xs[off + j * EC_COLS];
        }
        for (int r = 0; r < 128; ++r) {
            const int nk = cnt[r];
#pragma unroll
            for (int j = 0; j < NT; ++j) cur[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < nk; ++s) {
                float a_nxt = wl[(rec + 1) * 64];                   // <= LIC360_REC_PAD records past the end
                int t_nn = tl[(rec + 2) * 4];
                int off = (t_nxt & 0xffff) * EC_PL + ((t_nxt >> 16) & 0xff) * EC_COLS + ((t_nxt >> 24) & 0xff) + col;
                float b_nxt[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) b_nxt[j] = xs[off + j * EC_COLS];
#pragma unroll
                for (int j = 0; j < NT; ++j) cur[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[j], cur[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j) b_cur[j] = b_nxt[j];
                a_cur = a_nxt; t_nxt = t_nn; ++rec;
            }
            TREE_MERGE(NT);
        }
        // ---- epilogue: lane holds rows o0 + kq*4 + reg at column `col`, image rows r0 + j
        const int gc = c0 + col;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            int o = mi * 16 + kq * 4 + reg;
            if (o < nout && gc < W) {
                int bid = nbatch * nout + o;
                float bsv = bias[bid];
                float av = act ? act[bid] : 0.0f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    int gr = r0 + j;
                    if (gr < H) {
                        float sv = cur[j][reg] + bsv;
                        if (act) sv = sv > 0 ? sv : sv * av;        // cconv_ec_cuda.cu:311-312
                        const long oi = (((long)n * nout + o) * H + gr) * W + gc;
                        if (residual) sv = sv + residual[oi];       // fused `conv2(conv1(x)) + x` (lic360_demo.py:41)
                        out[oi] = sv;
                    }
                }
            }
        }
    }
}

LIC360_API int lic360_cconv_ec(void *stream, const lic360_conv_plan *p, const float *x, const float *packed, const float *bias,
                               const float *act, float *out, int n, int h, int w, int nb) {
    return lic360_cconv_ec_ex(stream, p, x, packed, bias, act, nullptr, out, n, h, w, nb, n);
}

LIC360_API int lic360_cconv_ec_ex(void *stream, const lic360_conv_plan *p, const float *x, const float *packed, const float *bias,
                                  const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod) {
    ARG_CHECK(p && x && packed && bias && out && n > 0 && h > 0 && w > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n);
    ARG_CHECK(p->ksz == 5);
    size_t lds = (size_t)p->C * EC_PL * sizeof(float);
    ARG_CHECK(lds <= 160 * 1024);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_cconv_ec<EC_TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((w + EC_TW - 1) / EC_TW, (h + EC_TH - 1) / EC_TH, n);
    hipLaunchKernelGGL(k_cconv_ec<EC_TH>, grid, dim3(EC_THREADS), lds, (hipStream_t)stream, x, packed, bias, act, residual, out,
                       p->d_mt_rec_start, p->d_leaf_cnt, p->d_term, p->C, h, w, p->nout, p->n_mtiles, n / nb,
                       lic360_conv_plan_packed_floats(p), x_mod);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// DC kernel: grid = (chunks of 16 positions, n_mtiles, N), block = one wave.
// Positions of output tile mi on plane psum: diagonals s = psum - g, g in [glo, ghi] -> contiguous
// range of the scan order.  Only rows whose group equals psum - th - tw are stored.
// Decode order: 8 waves per (output tile, 16 plane positions) task.  Wave w evaluates the leaves r = 16w .. 16w+15 of the
// visiting order -- a complete sub-tree of the reference's reduction -- so the serial K chain (operands gathered from
// global memory, one step prefetched) is 8x shorter than with one wave per task; the three top levels of the binary-counter
// merge are then applied to the 8 partial tiles in the same operand order ((B0+B1)+(B2+B3))+((B4+B5)+(B6+B7)).
__global__ __launch_bounds__(512) void k_cconv_dc(
    const float *__restrict__ x, const float *__restrict__ packed, const float *__restrict__ bias, const float *__restrict__ act,
    float *__restrict__ out, const int *__restrict__ mt_rec_start, const int *__restrict__ leaf_cnt, const int *__restrict__ term,
    const int *__restrict__ mt_glo, const int *__restrict__ mt_ghi, const int *__restrict__ idx, const int *__restrict__ plane_idx,
    int C, int H, int W, int nout, int cout, int half, int npb, long packed_per_net, int psum,
    const float *__restrict__ residual, int x_mod, long x_cs, long x_hs, long x_ws, long o_cs, long o_hs, long o_ws) {
    __shared__ f32x4 part[8][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), kq = lane >> 4, col = lane & 15;
    const int mi = blockIdx.y, n = blockIdx.z, nbatch = n / npb;
    // diagonal range of this tile on this plane
    int s_lo = psum - mt_ghi[mi], s_hi = psum - mt_glo[mi];
    if (s_lo < 0) s_lo = 0;
    if (s_hi > H + W - 2) s_hi = H + W - 2;
    if (s_lo > s_hi) return;
    const int qbeg = plane_idx[s_lo], qend = plane_idx[s_hi + 1];
    if (qbeg + (int)blockIdx.x * 16 >= qend) return;
    const int q = qbeg + blockIdx.x * 16 + col;
    const bool live = q < qend;
    const int th = live ? idx[q] : 0, tw = live ? idx[q + H * W] : 0;
    const float *xn = x + (long)(n % x_mod) * C * x_cs;
    const float *wp = packed + (long)nbatch * packed_per_net;
    const int *cnt = leaf_cnt + mi * 128;
    // first record of this wave's 16 leaves
    int before = 0;
    for (int i = lane; i < 16 * wv; i += 64) before += cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    long rec = mt_rec_start[mi] + before;
    f32x4 cur[1], s0[1], s1[1], s2[1], s3[1];
#define DC_LOADB(T, dst)                                                            \
    {                                                                               \
        int ph = th + ((T >> 16) & 0xff) - half, pw = tw + ((T >> 24) & 0xff) - half; \
        dst = 0.0f;                                                                 \
        if (live && ph >= 0 && ph < H && pw >= 0 && pw < W) dst = xn[(long)(T & 0xffff) * x_cs + ph * x_hs + pw * x_ws]; \
    }
    // operands of records rec .. rec+3 in flight (LIC360_REC_PAD = 4 records are readable past the end)
    float a0 = wp[rec * 64 + lane], a1 = wp[(rec + 1) * 64 + lane], a2 = wp[(rec + 2) * 64 + lane], b0, b1, b2;
    {
        const int t0 = term[rec * 4 + kq], t1 = term[(rec + 1) * 4 + kq], t2 = term[(rec + 2) * 4 + kq];
        DC_LOADB(t0, b0);
        DC_LOADB(t1, b1);
        DC_LOADB(t2, b2);
    }
    for (int rr = 0; rr < 16; ++rr) {
        const int r = rr, nk = cnt[16 * wv + rr];
        cur[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nk; ++s) {
            const float a3 = wp[(rec + 3) * 64 + lane];
            float b3;
            const int t3 = term[(rec + 3) * 4 + kq];
            DC_LOADB(t3, b3);
            cur[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, cur[0], 0, 0, 0);
            a0 = a1; a1 = a2; a2 = a3; b0 = b1; b1 = b2; b2 = b3; ++rec;
        }
        do { TREE_LEVEL(1, 0, s0) TREE_LEVEL(1, 1, s1) TREE_LEVEL(1, 2, s2) TREE_LEVEL(1, 3, s3) } while (0);
    }
#undef DC_LOADB
    part[wv][lane] = cur[0];
    __syncthreads();
    if (wv != 0 || !live) return;
    const f32x4 tot = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) + ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
    const int g = psum - th - tw;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        int o = mi * 16 + kq * 4 + reg;
        if (o < nout && o / cout == g) {
            int bid = nbatch * nout + o;
            float sv = tot[reg] + bias[bid];
            if (act) { if (sv < 0) sv = sv * act[bid]; }            // cconv_dc_cuda.cu:360-362
            const long oi = ((long)n * nout + o) * o_cs + th * o_hs + tw * o_ws;
            if (residual) sv = sv + residual[oi];                   // fused TileAdd (tile_add_cuda.cu:35)
            out[oi] = sv;
        }
    }
}

LIC360_API int lic360_cconv_dc_plane(void *stream, const lic360_conv_plan *p, const float *x, const float *packed, const float *bias,
                                     const float *act, float *out, int n, int h, int w, int nb,
                                     const int *idx_dev, const int *plane_idx_dev, const int *plane_idx_host, int psum) {
    return lic360_cconv_dc_plane_ex(stream, p, x, packed, bias, act, nullptr, out, n, h, w, nb, idx_dev, plane_idx_dev, plane_idx_host,
                                    psum, n, 0);
}

LIC360_API int lic360_cconv_dc_plane_ex(void *stream, const lic360_conv_plan *p, const float *x, const float *packed, const float *bias,
                                        const float *act, const float *residual, float *out, int n, int h, int w, int nb,
                                        const int *idx_dev, const int *plane_idx_dev, const int *plane_idx_host, int psum,
                                        int x_mod, int skewed) {
    // activation layout: NCHW, or diagonal-major [n][c][s=th+tw][th] (positions of one anti-diagonal contiguous)
    long cs = skewed ? (long)(h + w - 1) * h : (long)h * w, hs = skewed ? h + 1 : w, ws = skewed ? h : 1;
    return lic360_cconv_dc_plane_strided(stream, p, x, packed, bias, act, residual, out, n, h, w, nb, idx_dev, plane_idx_dev, plane_idx_host,
                                         psum, x_mod, cs, hs, ws, cs, hs, ws);
}

// internal (hidden visibility): explicit strides -- cell (th, tw) of plane c of sample n at [(n*C + c)*cs + th*hs + tw*ws] from the
// pointer passed (the caller adds a layout's constant offset to x / residual / out); residual uses the output strides
int lic360_cconv_dc_plane_strided(void *stream, const lic360_conv_plan *p, const float *x, const float *packed, const float *bias,
                                  const float *act, const float *residual, float *out, int n, int h, int w, int nb,
                                  const int *idx_dev, const int *plane_idx_dev, const int *plane_idx_host, int psum, int x_mod,
                                  long x_cs, long x_hs, long x_ws, long o_cs, long o_hs, long o_ws) {
    ARG_CHECK(p && x && packed && bias && out && idx_dev && plane_idx_dev && plane_idx_host && n > 0 && nb > 0 && n % nb == 0);
    ARG_CHECK(x_mod > 0 && x_mod <= n);
    if (psum < 0 || psum >= h + w + p->ngroup - 2) return 0;
    // widest tile range on this plane -> grid.x (blocks past a tile's range exit immediately)
    int maxpos = 0;
    for (int mi = 0; mi < p->n_mtiles; ++mi) {
        int s_lo = psum - p->mt_ghi[mi], s_hi = psum - p->mt_glo[mi];
        if (s_lo < 0) s_lo = 0;
        if (s_hi > h + w - 2) s_hi = h + w - 2;
        if (s_lo > s_hi) continue;
        int cntp = plane_idx_host[s_hi + 1] - plane_idx_host[s_lo];
        if (cntp > maxpos) maxpos = cntp;
    }
    if (maxpos == 0) return 0;
    dim3 grid((maxpos + 15) / 16, p->n_mtiles, n);
    hipLaunchKernelGGL(k_cconv_dc, grid, dim3(512), 0, (hipStream_t)stream, x, packed, bias, act, out, p->d_mt_rec_start, p->d_leaf_cnt,
                       p->d_term, p->d_mt_glo, p->d_mt_ghi, idx_dev, plane_idx_dev, p->C, h, w, p->nout, p->cout, p->half, n / nb,
                       lic360_conv_plan_packed_floats(p), psum, residual, x_mod, x_cs, x_hs, x_ws, o_cs, o_hs, o_ws);
    LAUNCH_CHECK();
    return 0;
}
