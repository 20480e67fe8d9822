// need_kernels.hip -- the dead cone of the latent entropy nets (need.h): per-layer "highest live group" maps from the importance mask, and
// the task lists they give the two conv kernels -- encode order: the live (sample, chunk of 4 x 16 tiles, group block) tasks of
// cconv16_kernels.hip, compacted per layer and XCD in launch order; decode order: per (layer, plane, XCD) the task records of
// cconv4v6_dc.inc with every sample's row window trimmed to the hull of its live rows and the windows of a chunk of samples packed end to
// end over the lanes (the tape packing of round 5 with a window per sample).
//
// What the reference does here: nothing -- cconv_ec / cconv_dc evaluate every output (extension/cconv_ec_cuda.cu:317-339,
// cconv_dc_cuda.cu:367-398) and coder.cpp:79 then skips the masked symbols.  Outputs are bit-identical: a skipped cell is only read by
// chains whose own outputs are dead, or with zero weights.
#include "common.h"
#include "need.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------ need maps
// one workgroup per image; layer l is computed from layer l + 1 of the same image, a workgroup barrier between the layers
#define NEED_T 1024
__global__ __launch_bounds__(NEED_T) void k_need(const float *__restrict__ mask, signed char *__restrict__ need, signed char *__restrict__ need_d,
                                                 signed char *__restrict__ tmax, int G, int H, int W, int nty, int ntx) {
    const int b = blockIdx.x, tid = threadIdx.x, HW = H * W, S = H + W - 1;
    signed char *nd = need + (long)b * NEED_LAYERS * HW, *dd = need_d + (long)b * NEED_LAYERS * S * H;
    const float *m = mask + (long)b * G * HW;
    for (int i = tid; i < NEED_LAYERS * S * H; i += NEED_T) dd[i] = -1;                // diagonal-major copy: cells outside the image
    for (int i = tid; i < HW; i += NEED_T) {
        int top = -1;
        for (int g = 0; g < G; ++g) if (!(m[(long)g * HW + i] < 0.5f)) top = g;       // coder.cpp:79
        nd[(NEED_LAYERS - 1) * HW + i] = (signed char)top;
    }
    for (int l = NEED_LAYERS - 2; l >= 0; --l) {
        __syncthreads();
        const signed char *up = nd + (l + 1) * HW;
        for (int i = tid; i < HW; i += NEED_T) {
            const int y = i / W, x = i - y * W;
            int best = -1;
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int dx = -2; dx <= 2; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    const int n = up[yy * W + xx];
                    if (n >= 0 && n + dy + dx > best) best = n + dy + dx;
                }
            }
            nd[l * HW + i] = (signed char)(best > G - 1 ? G - 1 : best);
        }
    }
    __syncthreads();
    for (int i = tid; i < NEED_LAYERS * HW; i += NEED_T) {
        const int l = i / HW, r = i - l * HW, y = r / W, x = r - y * W;
        dd[((long)l * S + y + x) * H + y] = nd[i];
    }
    for (int t = tid; t < NEED_LAYERS * nty * ntx; t += NEED_T) {
        const int l = t / (nty * ntx), r = t - l * nty * ntx, ty = r / ntx, tx = r - ty * ntx;
        int best = -1;
        for (int yy = ty * 4; yy < ty * 4 + 4 && yy < H; ++yy)
            for (int xx = tx * 16; xx < tx * 16 + 16 && xx < W; ++xx) { const int n = nd[l * HW + yy * W + xx]; best = n > best ? n : best; }
        tmax[(long)b * NEED_LAYERS * nty * ntx + t] = (signed char)best;
    }
}

// The same maps with the layer being dilated held in LDS and the 5 x 5 maximum taken separably (a row pass, then a column pass: max over (dy, dx) of
// need + dy + dx = max over dy of (max over dx of need + dx) + dy), for maps of at most 32 K cells: k_need's eleven layers of 25 byte loads from global
// memory behind workgroup barriers took 0.58 ms per call whatever the batch -- 5 % of a single image's encode.  -100 stands for "nothing live here".
__global__ __launch_bounds__(NEED_T) void k_need_lds(const float *__restrict__ mask, signed char *__restrict__ need, signed char *__restrict__ need_d,
                                                     signed char *__restrict__ tmax, int G, int H, int W, int nty, int ntx) {
    extern __shared__ signed char sm[];
    const int b = blockIdx.x, tid = threadIdx.x, HW = H * W, S = H + W - 1;
    signed char *cur = sm, *tmp = sm + HW;
    signed char *nd = need + (long)b * NEED_LAYERS * HW, *dd = need_d + (long)b * NEED_LAYERS * S * H;
    const float *m = mask + (long)b * G * HW;
    for (int i = tid; i < NEED_LAYERS * S * H; i += NEED_T) dd[i] = -1;
    for (int i = tid; i < HW; i += NEED_T) {
        int top = -1;
        for (int g = 0; g < G; ++g) if (!(m[(long)g * HW + i] < 0.5f)) top = g;
        cur[i] = (signed char)top;
    }
    __syncthreads();
    for (int l = NEED_LAYERS - 1; l >= 0; --l) {
        // cur = need_l: store it (row-major, diagonal-major), then dilate it into need_{l-1}
        for (int i = tid; i < HW; i += NEED_T) {
            const int y = i / W, x = i - y * W;
            const signed char v = cur[i];
            nd[l * HW + i] = v;
            dd[((long)l * S + y + x) * H + y] = v;
        }
        for (int t = tid; t < nty * ntx; t += NEED_T) {
            const int ty = t / ntx, tx = t - ty * ntx;
            int best = -1;
            for (int yy = ty * 4; yy < ty * 4 + 4 && yy < H; ++yy)
                for (int xx = tx * 16; xx < tx * 16 + 16 && xx < W; ++xx) { const int n = cur[yy * W + xx]; best = n > best ? n : best; }
            tmax[((long)b * NEED_LAYERS + l) * nty * ntx + t] = (signed char)best;
        }
        if (l == 0) break;
        for (int i = tid; i < HW; i += NEED_T) {                            // row pass
            const int y = i / W, x = i - y * W;
            int best = -100;
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const int n = cur[y * W + xx];
                if (n >= 0 && n + dx > best) best = n + dx;
            }
            tmp[i] = (signed char)best;
        }
        __syncthreads();
        for (int i = tid; i < HW; i += NEED_T) {                            // column pass
            const int y = i / W, x = i - y * W;
            int best = -100;
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                const int n = tmp[yy * W + x];
                if (n > -100 && n + dy > best) best = n + dy;
            }
            cur[i] = (signed char)(best < 0 ? -1 : (best > G - 1 ? G - 1 : best));
        }
        __syncthreads();
    }
}

int lic360_need_build(void *stream, const float *mask, int B, int G, int H, int W, signed char *need, signed char *need_d, signed char *tmax) {
    ARG_CHECK(mask && need && need_d && tmax && B > 0 && G > 0 && G <= 120 && H > 0 && W > 0);
    if (2L * H * W <= 65536) {
        hipLaunchKernelGGL(k_need_lds, dim3(B), dim3(NEED_T), (size_t)2 * H * W, (hipStream_t)stream, mask, need, need_d, tmax, G, H, W, (H + 3) / 4, (W + 15) / 16);
        LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_need, dim3(B), dim3(NEED_T), 0, (hipStream_t)stream, mask, need, need_d, tmax, G, H, W, (H + 3) / 4, (W + 15) / 16);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ encode order
// The launch order of cconv16_kernels.hip (c16_body::decode): task u of an XCD -> (sample n, first tile, group block); a task is live when one of its
// tiles holds a position that needs a group of the block.  One workgroup per (XCD, layer) compacts the live tasks in that order.
struct EcListArgs {
    const signed char *tmax;
    int *list, *cnt;
    unsigned long long *stats;
    int cap, l0, N, npb, n_chunks, n_gb, gbk, ntiles, gpb, tpt;
};
__global__ __launch_bounds__(256) void k_ec_tasks(const EcListArgs a) {
    __shared__ int wsum[4], s_stat[NEED_STAT_G];
    const int xcd = blockIdx.x, l = a.l0 + blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ns_x = (a.N - xcd + 7) >> 3, n_my = ns_x * a.n_chunks * a.n_gb;
    int *out = a.list + (long)(l * 8 + xcd) * a.cap;
    if (tid < NEED_STAT_G) s_stat[tid] = 0;
    int base = 0;
    for (int u0 = 0; u0 < n_my; u0 += 256) {
        const int u = u0 + tid;
        int tm = 0, gb = 0;
        if (u < n_my) {
            const int units = ns_x * a.n_chunks, per = a.gbk * a.n_gb, blk = u / per, r = u - blk * per;
            const int left = units - blk * a.gbk, kk = left < a.gbk ? left : a.gbk;
            gb = a.n_gb - 1 - r / kk;
            const int v = blk * a.gbk + r % kk, tile0 = (v % a.n_chunks) * a.tpt, n = xcd + 8 * (v / a.n_chunks), img = n % a.npb;
            const signed char *tmx = a.tmax + ((long)img * NEED_LAYERS + l) * a.ntiles;
            for (int t = 0; t < a.tpt; ++t)
                if (tile0 + t < a.ntiles && tmx[tile0 + t] >= gb * a.gpb) tm |= 1 << t;
        }
        const unsigned long long bal = __ballot(tm != 0);
        __syncthreads();                                                    // (wsum of the previous round has been read)
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (tm) {
            if (off < a.cap) out[off] = (int)((unsigned)u | (unsigned)tm << 28);
            if (a.stats) atomicAdd(&s_stat[gb & (NEED_STAT_G - 1)], __popc(tm));
        }
        base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    __syncthreads();
    if (tid == 0) a.cnt[l * 8 + xcd] = base < a.cap ? base : a.cap;
    if (a.stats && tid < NEED_STAT_G && s_stat[tid]) atomicAdd(&a.stats[l * NEED_STAT_G + tid], (unsigned long long)s_stat[tid]);
}

// hidden layers 1..10 (four groups per block, four tiles per task, the 3 B samples of the stacked nets) and the fused last layer (five groups,
// two tiles, B images); the first layer (cin = 1) has next to nothing to skip and keeps its full task list
int lic360_ec_lists_build(void *stream, const signed char *tmax, int B, int G, int H, int W, const lic360_ec_lists &l, unsigned long long *stats) {
    ARG_CHECK(tmax && l.list && l.cnt && B > 0);
    const int ntx = (W + 15) / 16, ntiles = ntx * ((H + 3) / 4);
    EcListArgs a;
    a.tmax = tmax; a.list = l.list; a.cnt = l.cnt; a.stats = stats; a.cap = l.cap; a.npb = B; a.ntiles = ntiles; a.gbk = lic360_ec_gbk();
    a.l0 = 1; a.N = 3 * B; a.gpb = 4; a.tpt = 4; a.n_chunks = (ntiles + 3) / 4; a.n_gb = (G + 3) / 4;
    ARG_CHECK(((a.N + 7) / 8) * a.n_chunks * a.n_gb <= l.cap && ((a.N + 7) / 8) * a.n_chunks * a.n_gb < (1 << 28));
    hipLaunchKernelGGL(k_ec_tasks, dim3(8, 10), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    a.l0 = 11; a.N = B; a.gpb = 5; a.tpt = 2; a.n_chunks = (ntiles + 1) / 2; a.n_gb = (G + 4) / 5;
    ARG_CHECK(((a.N + 7) / 8) * a.n_chunks * a.n_gb <= l.cap);
    hipLaunchKernelGGL(k_ec_tasks, dim3(8, 1), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ decode order
// One workgroup per (XCD, plane, layer): the group blocks of the plane as launch_cconv4v6_dc sees them (three groups on the staggered diagonals
// s0, s0 - 1, s0 - 2); per block and image of the XCD's list the hull of the rows that need one of the block's groups; per block, net and chunk of
// <= 8 samples the windows packed into waves (dcl_wave_pieces) -- or one plain record per live sample where packing saves nothing (plain tasks are
// cheaper to set up).  Order: group block major (heaviest first), net, chunk: the launch order of the unpacked kernel.
#define DCL_MAXB 24                                                         // group blocks of a plane (ngroup <= 72)
#define DCL_MAXM 64                                                         // images of an XCD's list (B <= 512)
struct DcListArgs {
    const signed char *need_d;
    uint4 *list;
    int *cnt;
    unsigned long long *stats;
    int cap, l0, npb, G, H, W, P;
};
__global__ __launch_bounds__(128) void k_dc_tasks(const DcListArgs a) {
    __shared__ short wlo[DCL_MAXB][DCL_MAXM], whi[DCL_MAXB][DCL_MAXM];
    __shared__ unsigned char wgm[DCL_MAXB][DCL_MAXM];                     // which of the block's three groups have a live row (of that image, on this plane)
    __shared__ int nrec[DCL_MAXB], plain[DCL_MAXB], offs[DCL_MAXB + 1], s_stat[DCL_MAXB * 3];
    const int xcd = blockIdx.x, p = blockIdx.y, l = a.l0 + blockIdx.z, tid = threadIdx.x;
    const int G = a.G, H = a.H, W = a.W, S = H + W - 1, m = a.npb / 8;
    const long li = ((long)l * a.P + p) * 8 + xcd;
    uint4 *out = a.list + li * a.cap;
    int gb_lo = 1 << 30, gb_hi = -1;
    for (int gb = 0; gb < (G + 2) / 3; ++gb) {
        const int s0 = p - gb * 3;
        if (s0 - 2 >= S || s0 < 0) continue;
        if (gb < gb_lo) gb_lo = gb;
        if (gb > gb_hi) gb_hi = gb;
    }
    if (gb_hi < 0) { if (tid == 0) a.cnt[li] = 0; return; }
    const int n_gb = gb_hi + 1, n_gbv = gb_hi - gb_lo + 1;
    if (tid < DCL_MAXB * 3) s_stat[tid] = 0;
    for (int job = tid; job < n_gbv * m; job += 128) {
        const int j = job / m, k = job - j * m, img = xcd + 8 * k, g0 = (n_gb - 1 - j) * 3, s0 = p - g0;
        int lo = 1 << 20, hi = -1, gm = 0;
        for (int q = 0; q < 3; ++q) {
            const int g = g0 + q, sq = s0 - q;
            if (g >= G || sq < 0 || sq >= S) continue;
            const int ya = sq >= W ? sq - W + 1 : 0, yb = sq < H ? sq : H - 1;
            const signed char *row = a.need_d + (((long)img * NEED_LAYERS + l) * S + sq) * H;
            int y = ya;
            while (y <= yb && row[y] < g) ++y;
            if (y > yb) continue;
            int z = yb;
            while (row[z] < g) --z;
            lo = y < lo ? y : lo;
            hi = z > hi ? z : hi;
            gm |= 1 << q;
        }
        wlo[j][k] = (short)(hi < 0 ? 0 : lo);
        whi[j][k] = (short)hi;
        wgm[j][k] = (unsigned char)gm;
    }
    __syncthreads();
    // pass 1 (one thread per block): records of one net's list, packed against plain
    if (tid < n_gbv) {
        const int j = tid;
        int live = 0, packed = 0;
        for (int c0 = 0; c0 < m; c0 += DCL_CHUNK) {
            const int c = m - c0 < DCL_CHUNK ? m - c0 : DCL_CHUNK;
            int lo[DCL_CHUNK], hi[DCL_CHUNK];
            for (int k = 0; k < c; ++k) { lo[k] = wlo[j][c0 + k]; hi[k] = whi[j][c0 + k]; live += hi[k] >= lo[k]; }
            int k = 0, slo = lo[0];
            unsigned pc[3];
            while (k < c && dcl_wave_pieces(lo, hi, H, c, k, slo, pc) > 0) ++packed;
        }
        plain[j] = packed >= live;
        nrec[j] = packed < live ? packed : live;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int j = 0; j < n_gbv; ++j) { offs[j] = acc; acc += 3 * nrec[j]; }
        offs[n_gbv] = acc;
        a.cnt[li] = acc < a.cap ? acc : a.cap;
    }
    __syncthreads();
    // pass 2 (one thread per block and net): the records
    if (tid < n_gbv * 3) {
        const int j = tid / 3, net = tid - j * 3, g0 = (n_gb - 1 - j) * 3, s0 = p - g0;
        int o = offs[j] + net * nrec[j];
        int cells[3] = {0, 0, 0};
        int dlo[3], dhi[3];
        for (int q = 0; q < 3; ++q) {
            const int sq = s0 - q;
            const bool ok = g0 + q < G && sq >= 0 && sq < S;
            dlo[q] = ok ? (sq >= W ? sq - W + 1 : 0) : 1;
            dhi[q] = ok ? (sq < H ? sq : H - 1) : 0;
        }
        auto count = [&](int rlo, int rhi, int gm) {
            for (int q = 0; q < 3; ++q) { const int u = rlo > dlo[q] ? rlo : dlo[q], v = rhi < dhi[q] ? rhi : dhi[q]; if (v >= u && (gm >> q & 1)) cells[q] += v - u + 1; }
        };
        for (int c0 = 0; c0 < m; c0 += DCL_CHUNK) {
            const int c = m - c0 < DCL_CHUNK ? m - c0 : DCL_CHUNK, nb = xcd + 8 * (net * m + c0);
            int lo[DCL_CHUNK], hi[DCL_CHUNK];
            for (int k = 0; k < c; ++k) { lo[k] = wlo[j][c0 + k]; hi[k] = whi[j][c0 + k]; }
            if (plain[j]) {
                for (int k = 0; k < c; ++k)
                    if (hi[k] >= lo[k]) {
                        const unsigned gm = wgm[j][c0 + k];
                        if (o < a.cap) out[o] = make_uint4((unsigned)g0 | (unsigned)(nb + 8 * k) << 10, gm << 22, 0u, 0u);
                        ++o;
                        count(0, H - 1, (int)gm);
                    }
            } else {
                int k = 0, slo = lo[0];
                unsigned pc[3];
                while (k < c && dcl_wave_pieces(lo, hi, H, c, k, slo, pc) > 0) {
                    unsigned gm = 0u;                                       // groups live in ANY of the record's samples (bits 22..24 of the first piece word)
                    for (int i = 0; i < 3; ++i) if (pc[i]) gm |= wgm[j][c0 + (pc[i] & 7)];
                    if (o < a.cap) out[o] = make_uint4((unsigned)g0 | 4u << 7 | (unsigned)nb << 10, pc[0] | gm << 22, pc[1], pc[2]);
                    ++o;
                    for (int i = 0; i < 3; ++i) if (pc[i]) count((pc[i] >> 3) & 63, (pc[i] >> 9) & 63, (int)gm);
                }
            }
        }
        if (a.stats) for (int q = 0; q < 3; ++q) if (cells[q]) atomicAdd(&s_stat[j * 3 + q], cells[q]);
    }
    if (a.stats) {
        __syncthreads();
        if (tid < n_gbv * 3 && s_stat[tid]) {
            const int j = tid / 3, g = (n_gb - 1 - j) * 3 + tid % 3;
            if (g < G) atomicAdd(&a.stats[l * NEED_STAT_G + g], (unsigned long long)s_stat[tid]);
        }
    }
}

// Balance the eight lists of a launch.  Each XCD owns the samples n = xcd (mod 8) -- their bands stay in that XCD's L2 from layer to layer -- and with the
// dead-cone hulls the lists' work differs with the images' masks: a launch waits for its most loaded XCD, +3.9 % over the mean on SURVEY 8d's masks
// (12 % on the last layer; tools/list_balance.py).  A record carries its absolute sample index, so any XCD can run it: the lightest records (the tail: lists
// are heaviest block first) of the most loaded list move to the least loaded one while that lowers the pair's maximum.  Work of a record = the steps of its
// block's longest chain.  One workgroup per (plane, layer); the moved records (a few per cent) read their bands through another XCD's L2.
__global__ __launch_bounds__(64) void k_dc_balance(const DcListArgs a) {
    __shared__ int w[8], n[8];
    const int p = blockIdx.x, l = a.l0 + blockIdx.y, tid = threadIdx.x;
    const long li = ((long)l * a.P + p) * 8;
    auto steps = [&](unsigned x) { const int g0 = (int)(x & 127u), s = g0 + 2 + 4 + 1; return s < a.G ? s : a.G; };
    if (tid < 8) {
        const int c = a.cnt[li + tid];
        const uint4 *r = a.list + (li + tid) * a.cap;
        int acc = 0;
        for (int i = 0; i < c; ++i) acc += steps(r[i].x);
        n[tid] = c; w[tid] = acc;
    }
    __syncthreads();
    if (tid != 0) return;
    for (int it = 0; it < 8 * a.cap; ++it) {
        int hx = 0, lx = 0;
        for (int x = 1; x < 8; ++x) { if (w[x] > w[hx]) hx = x; if (w[x] < w[lx]) lx = x; }
        if (hx == lx || n[hx] == 0 || n[lx] >= a.cap) break;
        const uint4 rec = a.list[(li + hx) * a.cap + n[hx] - 1];
        const int s = steps(rec.x);
        if (w[lx] + s >= w[hx]) break;                                      // the move would not lower the pair's maximum
        a.list[(li + lx) * a.cap + n[lx]] = rec;
        ++n[lx]; --n[hx];
        w[lx] += s; w[hx] -= s;
    }
    for (int x = 0; x < 8; ++x) a.cnt[li + x] = n[x];
}

// layers 1..11 (cin = 4: hidden and last layers; the first layer keeps its full task list)
int lic360_dc_lists_build(void *stream, const signed char *need_d, int B, int G, int H, int W, const lic360_dc_lists &l, unsigned long long *stats) {
    ARG_CHECK(need_d && l.list && l.cnt && B > 0 && B % 8 == 0 && B / 8 <= DCL_MAXM && H <= 64 && G <= 3 * DCL_MAXB && l.P == H + W + G - 2);
    ARG_CHECK(l.cap >= ((G + 2) / 3 < DCL_MAXB ? (G + 2) / 3 : DCL_MAXB) * 3 * (B / 8) && 3L * B < (1L << 22));
    DcListArgs a;
    a.need_d = need_d; a.list = l.list; a.cnt = l.cnt; a.stats = stats; a.cap = l.cap; a.l0 = 1; a.npb = B; a.G = G; a.H = H; a.W = W; a.P = l.P;
    hipLaunchKernelGGL(k_dc_tasks, dim3(8, l.P, NEED_LAYERS - 1), dim3(128), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    static const bool balance = !getenv("LIC360_NOBALANCE");                // (A/B switch of the XCD balancing pass)
    if (balance) {
        hipLaunchKernelGGL(k_dc_balance, dim3(l.P, NEED_LAYERS - 1), dim3(64), 0, (hipStream_t)stream, a);
        LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ test hooks (C ABI)
// need maps of a batch of masks [B, G, H, W] (device) -> need [B][12][H][W] int8 (device): tests/test_gpu_need.py checks them against a brute-force
// reachability statement of the dead cone
LIC360_API int lic360_need_maps(void *stream, const float *mask, int B, int G, int H, int W, signed char *need_out) {
    ARG_CHECK(mask && need_out && B > 0 && G > 0 && G <= 127 && H > 0 && W > 0);
    signed char *dd = nullptr, *tm = nullptr;
    const long S = H + W - 1, nt = (long)((H + 3) / 4) * ((W + 15) / 16);
    HIP_TRY(hipMalloc((void **)&dd, (size_t)B * NEED_LAYERS * S * H));
    HIP_TRY(hipMalloc((void **)&tm, (size_t)B * NEED_LAYERS * nt));
    const int rc = lic360_need_build(stream, mask, B, G, H, W, need_out, dd, tm);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    (void)hipFree(dd); (void)hipFree(tm);
    return rc;
}

// Host-only view of the list packing (no GPU work): the waves into which dcl_wave_pieces lays the row windows lo[k]..hi[k] (k < c <= 8; hi < lo: sample k
// has no live row) of ONE chunk of samples on an image of h rows; pieces[3 w + i] = piece i of wave w (k | slo << 3 | shi << 9 | a0 << 15 | 1 << 21, 0 = none),
// room for 3 * 2 * c words; *n_waves = waves used.  For tests of the packing rules on the CPU (tests/test_dcl_pack.py).
LIC360_API int lic360_dcl_pack_layout(int h, int c, const int *lo, const int *hi, unsigned *pieces, int *n_waves) {
    ARG_CHECK(h > 0 && h <= 64 && c > 0 && c <= DCL_CHUNK && lo && hi && pieces && n_waves);
    for (int k = 0; k < c; ++k) ARG_CHECK(hi[k] < lo[k] || (lo[k] >= 0 && hi[k] < h));
    int k = 0, slo = lo[0], nw = 0;
    unsigned pc[3];
    while (k < c && dcl_wave_pieces(lo, hi, h, c, k, slo, pc) > 0) {
        ARG_CHECK(nw < 2 * c);
        for (int i = 0; i < 3; ++i) pieces[3 * nw + i] = pc[i];
        ++nw;
    }
    *n_waves = nw;
    return 0;
}
