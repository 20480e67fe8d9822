// conv3x3_kernels.hip -- the 3x3 stride-1 convolutions of the analysis / synthesis transforms on sphere-apron maps,
// hand-written for gfx950: the tile loader reads the SPHERE APRON BY INDEX (no padded copy, no SpherePad launch), the epilogue applies
// bias + PReLU + the block's residual add, and the SphereTrim that follows every such convolution becomes the output window (cells outside
// it are simply not computed).  Replaces, per layer, nn.Conv2d + SpherePad + nn.PReLU + SphereTrim (+ the residual add) of
// test/model_zoo.py:45-62 of the reference (ResidualBlockV2), :8-23 (ResidualBlock.conv2), :64-94 (ResidualBlockDown.conv2),
// :144-169 (ResidualBlockUp.conv1 / conv2); apron rule: extension/sphere_pad_cuda.cu:48-65.
//
// Arithmetic: fp32 throughout, v_mfma_f32_16x16x4_f32 (bit for bit a k-ordered fmaf chain of 4 terms); the summation order over
// (input channel, tap) is this kernel's own -- a library convolution fixes none either -- so parity with the oracle's conv2d is to 1e-4.
//
// Mapping.  Implicit GEMM with M = output channels, N = positions, K = (input channel, tap).  A workgroup (8 waves, two per SIMD) owns
// a 16 x 16 tile of output positions of one image and NQ * 48 output channels: wave (mq, nh) keeps 48 channels (3 MFMA row tiles) x
// RW rows x 16 columns in 12 RW accumulator registers (96 at NQ = 4, RW = 8).  Input channels arrive in chunks of 16: their 18 x 18 halo tiles
// go to a double-buffered LDS image by per-lane LDS-DMA (`global_load_lds_dword`: each lane fetches ONE cell from wherever the sphere
// rule says it lives -- longitude wrap, pole rows reflected and mirrored -- so interior and apron cells cost the same), one chunk ahead;
// one barrier per chunk (864 MFMAs per wave).  For a (4-channel group, kw) pair a wave reads RW + 2 B operands (one per input row:
// output row r at tap row kh reads input row r + kh) and 9 A operands (3 kh x 3 row tiles, three 16-byte loads per lane from a stream
// packed in exactly this order, one pair ahead) for 9 RW MFMAs: 0.26 operand fetches per MFMA.  LDS plane pitch 336 = 16 (mod 64)
// banks: the four k-planes of a B read fall on disjoint bank quarters.
#include "common.h"
#include <cstdint>

typedef float s3_f4 __attribute__((ext_vector_type(4)));

#define S3_T 16                                      // tile columns (and rows of the main tile shape)
#define S3_THREADS 512
// tile shapes: TR rows x 16 columns, kernel size KS (3, or 1: the transforms' 1x1 layers on the same body).  LDS plane of a channel:
// (TR + KS - 1) x (16 + KS - 1) floats, pitch rounded up to 16 (mod 64) banks; CK input channels per LDS chunk (16 at KS = 3, 32 at KS = 1)
constexpr int s3_pitch(int tr, int ks) { int p = (tr + ks - 1) * (S3_T + ks - 1); while (p % 64 != 16) ++p; return p; }   // TR = 16, KS = 3: 324 -> 336
constexpr int s3_ck(int ks) { return ks == 3 ? 16 : 32; }
constexpr int s3_ndma(int tr, int ks) { return (s3_ck(ks) * s3_pitch(tr, ks) + 511) / 512; }       // DMA instructions per wave and chunk (8 waves x 64 lanes)
constexpr int s3_na4(int ks) { return (3 * ks + 3) / 4; }                                          // 16-byte A loads per lane and (channel group, kw) pair: 3 ks row-tile operands

struct S3Args {
    const float *x, *w, *bias, *slope, *res;
    float *out;
    int n, cin, cout, hp, wp;                        // x: [n][cin][hp][wp]; cout = output channels of this launch (all blocks)
    int pad, sphere;                                 // 1: cells of the `pad`-wide apron are read from the interior by the sphere rule; 2: longitude wrap only
                                                     // (rows as they are: a map whose pole rows hold computed values, the 1-ring output of another launch)
    int ring, ringw;                                 // output window = rows [ring, hp - ring) x columns [ringw, wp - ringw) of the input grid
    int rw, tall_last;                               // rows per wave of a tile; tall_last != 0: the last tile row runs rw + 1 rows per wave (the window's remainder rows)
    int ohp, owp, ooff;                              // out: [n][cout][ohp][owp], window cell (ph, pw) at (ph - ooff, pw - ooff)
    int shuffle;                                     // != 0: out is the x2 pixel shuffle of that (Dtow(2, d2w), dtow_cuda.cu:38-75): [n][cout / 4][2 ohp][2 owp], channel 4 p + v of
                                                     // cell (y, x) at channel p, cell (2 y + v / 2, 2 x + v % 2) -- a lane's four accumulator registers are one 2 x 2 block
    int tiles_x, tiles_y;
};

__device__ __forceinline__ void s3_dma(unsigned voff, const float *sbase, unsigned lds_byte_addr) {
    // LDS destination = M0 + lane * 4; M0 is written inside the statement (tests/test_asm_m0.py: the compiler itself never reads M0 here)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ const float *s3_uniform(const float *p) {       // the value IS wave-uniform; this tells the compiler
    const unsigned long v = (unsigned long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const float *)(((unsigned long)hi << 32) | lo);
}
// padded cell (ph, pw) of a sphere map -> the padded cell that holds its value (interior cells map to themselves); sphere_pad_cuda.cu:48-65
__device__ __forceinline__ void s3_sphere(int &ph, int &pw, int hp, int wp, int pad) {
    const int H = hp - 2 * pad, W = wp - 2 * pad;
    int th = ph - pad, tw = pw - pad;
    tw = tw < 0 ? tw + W : (tw >= W ? tw - W : tw);
    if (th < 0) { th = -1 - th; tw = W - 1 - tw; }
    else if (th >= H) { th = 2 * H - 1 - th; tw = W - 1 - tw; }
    ph = th + pad; pw = tw + pad;
}

// weights: [cout block of NQ * 48][cin / 4][kw < ks][mq][j < na4][lane] x 4 floats; lane l = 16 k + i, element e = 4 j + t = 3 kh + mt (e < 3 ks):
// W[co = 48 mq + 16 mt + i][ci = 4 cg + k][kh][kw] -- the A operand of the MFMA for (kh, row tile mt)
__global__ void k_sconv3x3_pack(const float *__restrict__ w, float *__restrict__ packed, int cin, int cout, int nq, int ks, long total) {
    const int na4 = (3 * ks + 3) / 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int t = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        long r = idx >> 8;
        const int j = (int)(r % na4); r /= na4;
        const int mq = (int)(r % nq); r /= nq;
        const int kw = (int)(r % ks); r /= ks;
        const int cg = (int)(r % (cin / 4)), blk = (int)(r / (cin / 4));
        const int e = 4 * j + t, kh = e / 3, mt = e - 3 * kh;
        float v = 0.0f;
        if (e < 3 * ks) {
            const int co = blk * nq * 48 + 48 * mq + 16 * mt + (lane & 15), ci = 4 * cg + (lane >> 4);
            v = w[(((long)co * cin + ci) * ks + kh) * ks + kw];
        }
        packed[idx] = v;
    }
}

template <int NQ, int RW, int PD, int KS>                                   // NQ * 48 output channels per workgroup, RW rows per wave, A operands PD pairs ahead, KS x KS taps
__device__ __forceinline__ void s3_body(const S3Args &a, float *lds, int ty, int tx, int img) {
    constexpr int NR = 8 / NQ, TR = NR * RW;                                // row groups per workgroup, tile rows
    constexpr int S3_CK = s3_ck(KS), NPAIR = S3_CK / 4 * KS, NA4 = s3_na4(KS), S3_XC = S3_T + KS - 1;
    constexpr int S3_PL = s3_pitch(TR, KS), S3_NDMA = s3_ndma(TR, KS), S3_BUF = 8 * S3_NDMA * 64, S3_XR = TR + KS - 1;
    static_assert(NPAIR % (PD + 1) == 0, "the operand ring's index must be static across chunks");
    float (*xs)[S3_BUF] = (float (*)[S3_BUF])lds;
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), mq = wave % NQ, nh = wave / NQ;
    const int tr0 = a.ring + ty * (NR * a.rw), tc0 = a.ringw + tx * S3_T;    // input-grid cell of the tile's first output (a.rw: rows per wave of the ordinary tile rows)
    const int blk = blockIdx.y, cblk = NQ * 48;
    const long PLg = (long)a.hp * a.wp;
    // ---- this lane's cells of a chunk's LDS image: LDS float q = (i * 8 + wave) * 64 + lane <-> (channel q / 336, halo row, halo column)
    unsigned voff[S3_NDMA];
#pragma unroll
    for (int i = 0; i < S3_NDMA; ++i) {
        const int q = (i * 8 + wave) * 64 + lane;
        int ch = q / S3_PL, rem = q - ch * S3_PL;
        if (ch >= S3_CK || rem >= S3_XR * S3_XC) { ch = 0; rem = 0; }      // pitch padding and the slack behind the last plane: any valid cell
        const int r = rem / S3_XC, c = rem - r * S3_XC;
        int ph = tr0 - KS / 2 + r, pw = tc0 - KS / 2 + c;
        ph = ph < 0 ? 0 : (ph > a.hp - 1 ? a.hp - 1 : ph);                  // (only cells of outputs outside the window reach past the map)
        pw = pw < 0 ? 0 : (pw > a.wp - 1 ? a.wp - 1 : pw);
        if (a.sphere == 1) s3_sphere(ph, pw, a.hp, a.wp, a.pad);
        else if (a.sphere == 2) { const int W = a.wp - 2 * a.pad; pw = pw < a.pad ? pw + W : (pw >= a.pad + W ? pw - W : pw); }
        voff[i] = (unsigned)(((long)ch * PLg + (long)ph * a.wp + pw) * 4);
    }
    const float *xb = a.x + (long)img * a.cin * PLg;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const float *)&xs[0][0];
    auto issue_dma = [&](int ck) __attribute__((always_inline)) {
        const float *sb = s3_uniform(xb + (long)ck * S3_CK * PLg);
        const unsigned lb = lds0 + (unsigned)((ck & 1) * S3_BUF + wave * 64) * 4u;
#pragma unroll
        for (int i = 0; i < S3_NDMA; ++i) s3_dma(voff[i], sb, lb + (unsigned)(i * 8 * 64 * 4));
    };
    const int nck = a.cin / S3_CK, niter = nck * NPAIR;
    // A operands: asm loads + counted waits (hipcc sinks visible loads to just before their first use to save registers, which leaves their
    // L2 latency exposed twice per pair; here the three 16-byte loads of pair it + 1 are issued at the top of pair it and waited for at the
    // top of pair it + 1).  In-order counter: a chunk's 11 DMAs are issued BEHIND the A loads of its first pair, so `vmcnt(11)` at the second
    // pair waits for the operands only and the DMAs have two pairs (~9000 cycles) before a `vmcnt(0)` asks for them.
    const char *wl = (const char *)((const s3_f4 *)a.w + ((long)blk * (a.cin / 4) * KS * NQ + mq) * NA4 * 64 + lane);   // + it * NQ * NA4 KB per (cg, kw) pair
    auto load_a = [&](int it, s3_f4 (&A)[3]) __attribute__((always_inline)) {
        const char *p = wl + (long)it * (NQ * NA4 * 1024);
        if constexpr (NA4 == 3)
            asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:1024\n\tglobal_load_dwordx4 %2, %3, off offset:2048"
                         : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]) : "v"(p));
        else
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(A[0]) : "v"(p));
    };
#define S3_WAIT_A(N, A_)                                                                                              \
    do {                                                                                                              \
        if constexpr (NA4 == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(A_[0]), "+v"(A_[1]), "+v"(A_[2]) : "n"(N)); \
        else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(A_[0]) : "n"(N));                                              \
    } while (0)
    s3_f4 acc[3][RW];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[m][r] = (s3_f4){0.f, 0.f, 0.f, 0.f};
    issue_dma(0);
    // operand ring: PD + 1 sets, pair `it` in set it % (PD + 1).  The wide tiles run one pair ahead (a pair is 72 MFMAs = 2304 cycles per wave);
    // the 2-row remainder tiles have 9 MFMAs per pair and would wait a full L2 round trip per pair: they run five ahead.
    s3_f4 A[PD + 1][3];
#pragma unroll
    for (int d = 0; d < PD; ++d) load_a(d < niter ? d : niter - 1, A[d]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int ck = 0; ck < nck; ++ck) {
        const float *xl = &xs[ck & 1][kq * S3_PL + nh * RW * S3_XC + col];
        float b[2][RW + KS - 1];
#pragma unroll
        for (int j = 0; j < RW + KS - 1; ++j) b[0][j] = xl[j * S3_XC];       // pair 0 of the chunk (the later pairs are read one pair ahead)
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {                                   // (4-channel group, kw) pairs of the chunk
            const int cur = p & 1, nxt = cur ^ 1, sa = p % (PD + 1), sn = (p + PD) % (PD + 1);
            // in-order counter: behind set sa's loads came PD - 1 younger sets and -- for the PD pairs that follow a chunk's first -- its DMAs
            if (p >= 1 && p <= PD) S3_WAIT_A(NA4 * (PD - 1) + S3_NDMA, A[sa]);  // (registers written by an asm load are only read behind the wait that names them)
            else S3_WAIT_A(NA4 * (PD - 1), A[sa]);
            const int itn = ck * NPAIR + p + PD;
            load_a(itn < niter ? itn : niter - 1, A[sn]);
            if (p == 0 && ck + 1 < nck) issue_dma(ck + 1);
            if (p == 0 && ck + 1 >= nck) {                                   // keep the counts of the waits at p = 1 .. PD right: harmless loads into the idle buffer
#pragma unroll
                for (int i = 0; i < S3_NDMA; ++i) s3_dma(voff[i], s3_uniform(xb), lds0 + (unsigned)(((ck & 1) ^ 1) * S3_BUF + (i * 8 + wave) * 64) * 4u);
            }
            if (p + 1 < NPAIR) {
                const int cgn = (p + 1) / KS, kwn = (p + 1) - KS * cgn;
#pragma unroll
                for (int j = 0; j < RW + KS - 1; ++j) b[nxt][j] = xl[cgn * 4 * S3_PL + j * S3_XC + kwn];
            }
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const int e = 3 * kh + m;
                        acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[sa][e >> 2][e & 3], b[cur][r + kh], acc[m][r], 0, 0, 0);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the next chunk's DMAs have landed (and the prefetched A operands)
        __syncthreads();
    }
#undef S3_WAIT_A
    // ---- epilogue: bias, PReLU, residual, store.  Accumulator m, row r, register v: channel 48 mq + 16 m + 4 kq + v, position (row, col)
    const int pw = tc0 + col;
    const long oPL = (long)a.ohp * a.owp;
    const float *__restrict__ resp = a.res;
    float *__restrict__ outp = a.out;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int co = blk * cblk + 48 * mq + 16 * m + 4 * kq;
        const s3_f4 bs = *(const s3_f4 *)(a.bias + co);
        s3_f4 sl = {1.f, 1.f, 1.f, 1.f};
        if (a.slope) sl = *(const s3_f4 *)(a.slope + co);
        s3_f4 rv[RW];
        if (resp) {                                                         // all residual loads of the row tile in flight before its first store
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int ph = tr0 + nh * RW + r;
                const bool ok = ph < a.hp - a.ring && pw < a.wp - a.ringw;
                const int rh = ok ? ph : a.ring, rw_ = ok ? pw : a.ringw;
                if (a.shuffle) {                                           // the residual has the OUTPUT's (shuffled) geometry
                    typedef float s3_f2 __attribute__((ext_vector_type(2)));
                    const long ri = (((long)img * (a.cout >> 2) + (co >> 2)) * (2 * a.ohp) + 2 * (rh - a.ooff)) * (2 * a.owp) + 2 * (rw_ - a.ooff);
                    const s3_f2 lo = *(const s3_f2 *)(resp + ri), hi = *(const s3_f2 *)(resp + ri + 2 * a.owp);
                    rv[r] = (s3_f4){lo[0], lo[1], hi[0], hi[1]};
                } else {
                    const long ri = ((long)img * a.cout + co) * PLg + (long)rh * a.wp + rw_;
#pragma unroll
                    for (int v = 0; v < 4; ++v) rv[r][v] = resp[ri + v * PLg];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int ph = tr0 + nh * RW + r;
            if (ph < a.hp - a.ring && pw < a.wp - a.ringw) {
                float y[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    y[v] = acc[m][r][v] + bs[v];
                    if (a.slope) y[v] = y[v] > 0.f ? y[v] : y[v] * sl[v];
                    if (resp) y[v] = y[v] + rv[r][v];
                }
                if (a.shuffle) {                                           // two 8-byte stores per lane: 16 lanes write 128 contiguous bytes of each of two rows
                    typedef float s3_f2 __attribute__((ext_vector_type(2)));
                    const long o = (((long)img * (a.cout >> 2) + (co >> 2)) * (2 * a.ohp) + 2 * (ph - a.ooff)) * (2 * a.owp) + 2 * (pw - a.ooff);
                    *(s3_f2 *)(outp + o) = (s3_f2){y[0], y[1]};
                    *(s3_f2 *)(outp + o + 2 * a.owp) = (s3_f2){y[2], y[3]};
                } else {
                    const long o = ((long)img * a.cout + co) * oPL + (long)(ph - a.ooff) * a.owp + (pw - a.ooff);
#pragma unroll
                    for (int v = 0; v < 4; ++v) outp[o + v * oPL] = y[v];
                }
            }
        }
    }
}

// A window of 16 k + 2 rows (every 1-ring window of these maps) would need a seventeenth tile row with 14 dead rows; instead its LAST tile row
// runs one more row per wave (18 rows at 192 channels, 20 at 96): the workgroup picks its body by its tile row (uniform per workgroup).
template <int NQ, int RW, int KS>
__global__ __launch_bounds__(S3_THREADS) void k_sconv3x3(S3Args a) {
    constexpr int NR = 8 / NQ;
    __shared__ float lds[2 * 8 * s3_ndma(NR * (RW + (KS == 3 ? 1 : 0)), KS) * 64];
    const int tpi = a.tiles_x * a.tiles_y, img = blockIdx.x / tpi, trem = blockIdx.x - img * tpi, ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    if constexpr (KS == 3) {
        if (a.tall_last && ty == a.tiles_y - 1) { s3_body<NQ, RW + 1, 1, 3>(a, lds, ty, tx, img); return; }
    }
    s3_body<NQ, RW, KS == 3 ? 1 : 3, KS>(a, lds, ty, tx, img);            // a 1x1 pair is 24 MFMAs (768 cycles): its A operand is fetched three pairs ahead
}

static inline bool s3_ok(int cin, int cout, int ks = 3) { return (ks == 3 || ks == 1) && cin >= s3_ck(ks) && cin % s3_ck(ks) == 0 && cout >= 96 && (cout % 192 == 0 || cout == 96); }
static inline long s3_packed(int cin, int cout, int ks) { return s3_ok(cin, cout, ks) ? (long)cout / 48 * (cin / 4) * ks * s3_na4(ks) * 256 : 0; }
static int s3_pack(void *stream, const float *weight, float *packed, int cin, int cout, int ks) {
    ARG_CHECK(weight && packed && s3_ok(cin, cout, ks));
    const long total = s3_packed(cin, cout, ks);
    hipLaunchKernelGGL(k_sconv3x3_pack, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, weight, packed, cin, cout, cout % 192 == 0 ? 4 : 2, ks, total);
    LAUNCH_CHECK();
    return 0;
}
static int s3_launch(void *stream, const float *x, const float *packed, const float *bias, const float *slope, const float *residual, float *out,
                     int n, int cin, int cout, int hp, int wp, int pad, int sphere, int ring, int ring_w, int out_crop, int ks, int shuffle = 0) {
    ARG_CHECK(x && packed && bias && out && n > 0 && s3_ok(cin, cout, ks) && pad >= 0 && ring >= ks / 2 && ring_w >= ring && hp > 2 * ring && wp > 2 * ring_w && out_crop >= 0 &&
              out_crop <= ring && sphere >= 0 && sphere <= 2);
    ARG_CHECK(!sphere || (pad >= 1 && hp >= 4 * pad && wp >= 4 * pad));     // the wrapped / reflected source of an apron cell is an interior cell
    ARG_CHECK((double)s3_ck(ks) * hp * wp * 4.0 < 4294967296.0 && ((uintptr_t)bias & 15) == 0 && (!slope || ((uintptr_t)slope & 15) == 0));
    ARG_CHECK(!residual || out_crop == 0 || shuffle);                       // the residual has the input's geometry -- or, shuffled, the output's
    ARG_CHECK(!shuffle || (((uintptr_t)out & 7) == 0 && ((uintptr_t)residual & 7) == 0));   // the shuffled store / residual load move aligned pairs
    S3Args a;
    a.x = x; a.w = packed; a.bias = bias; a.slope = slope; a.res = residual; a.out = out;
    a.n = n; a.cin = cin; a.cout = cout; a.hp = hp; a.wp = wp; a.pad = pad; a.sphere = sphere; a.ring = ring; a.ringw = ring_w;
    a.ooff = out_crop; a.ohp = hp - 2 * out_crop; a.owp = wp - 2 * out_crop; a.shuffle = shuffle;
    a.tiles_x = (wp - 2 * ring_w + S3_T - 1) / S3_T;
    const int nq = cout % 192 == 0 ? 4 : 2, nrg = 8 / nq, nr = hp - 2 * ring, full = nr / S3_T, rem = nr - full * S3_T;
    a.rw = S3_T / nrg;
    a.tall_last = ks == 3 && rem > 0 && rem <= nrg && full > 0;             // the remainder fits one more row per wave of the last tile row
    a.tiles_y = a.tall_last ? full : (nr + S3_T - 1) / S3_T;
    const long tiles = (long)n * a.tiles_x * a.tiles_y;
    ARG_CHECK(tiles < (1L << 31));
    const dim3 grid((unsigned)tiles, nq == 4 ? cout / 192 : 1);
    if (ks == 3 && nq == 4) hipLaunchKernelGGL((k_sconv3x3<4, 8, 3>), grid, dim3(S3_THREADS), 0, (hipStream_t)stream, a);
    else if (ks == 3) hipLaunchKernelGGL((k_sconv3x3<2, 4, 3>), grid, dim3(S3_THREADS), 0, (hipStream_t)stream, a);
    else if (nq == 4) hipLaunchKernelGGL((k_sconv3x3<4, 8, 1>), grid, dim3(S3_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_sconv3x3<2, 4, 1>), grid, dim3(S3_THREADS), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_sconv3x3_supported(int cin, int cout) { return s3_ok(cin, cout, 3) ? 1 : 0; }
LIC360_API long lic360_sconv3x3_packed_floats(int cin, int cout) { return s3_packed(cin, cout, 3); }
LIC360_API int lic360_sconv3x3_pack(void *stream, const float *weight, float *packed, int cin, int cout) { return s3_pack(stream, weight, packed, cin, cout, 3); }
LIC360_API int lic360_sconv3x3(void *stream, const float *x, const float *packed, const float *bias, const float *slope, const float *residual, float *out,
                               int n, int cin, int cout, int hp, int wp, int pad, int sphere, int ring, int ring_w, int out_crop, int shuffle) {
    return s3_launch(stream, x, packed, bias, slope, residual, out, n, cin, cout, hp, wp, pad, sphere, ring, ring_w, out_crop, 3, shuffle);
}
// the transforms' 1x1 layers on the same body (K = input channels only, no halo): bias + PReLU + residual in the epilogue, the window as above
LIC360_API int lic360_sconv1x1_supported(int cin, int cout) { return s3_ok(cin, cout, 1) ? 1 : 0; }
LIC360_API long lic360_sconv1x1_packed_floats(int cin, int cout) { return s3_packed(cin, cout, 1); }
LIC360_API int lic360_sconv1x1_pack(void *stream, const float *weight, float *packed, int cin, int cout) { return s3_pack(stream, weight, packed, cin, cout, 1); }
LIC360_API int lic360_sconv1x1(void *stream, const float *x, const float *packed, const float *bias, const float *slope, const float *residual, float *out,
                               int n, int cin, int cout, int hp, int wp, int ring, int ring_w, int crop, int shuffle) {
    return s3_launch(stream, x, packed, bias, slope, residual, out, n, cin, cout, hp, wp, 0, 0, ring, ring_w, crop, 1, shuffle);
}
