// need.h -- the "dead cone" of the latent entropy nets (round 6), shared by need_kernels.hip, the two conv kernels and the fused codec.
//
// A symbol (g, y, x) with mask < 0.5 is never coded (extension/coder.cpp:79), so nothing can observe the last layer's outputs at it; and
// an activation of layer l at (g, q) is dead unless some live output of layer l + 1 reads it with a non-zero weight.  Layer l + 1 at
// position p, group gp reads q = p + d, d in [-2, 2]^2, input groups <= gp - d_y - d_x (extension/cconv_ec_cuda.cu:288-290 with hidden = 1,
// the same rule in decode order: cconv_dc_cuda.cu:313-364), and the residual blocks read (g, q) itself (covered by d = 0).  With the
// importance mask a prefix in g (g < L[y/2, x/2]: ImpMap -> Dtow) the live groups of a position are a prefix too:
//   need_11(p)  = highest group with mask >= 0.5 at p                      (-1: none)
//   need_l(q)   = min(G - 1, max over p in q + [-2, 2]^2 of need_{l+1}(p) + (p_y - q_y) + (p_x - q_x))   (need_{l+1}(p) >= 0 only; < 0: -1)
// i.e. "plane of the last consumer" t = need + y + x is dilated 5 x 5 per layer.  The reference computes the dead cells too; skipping them
// changes no bitstream and no decoded symbol (they are only read by chains whose own outputs are dead, or with zero weights: the cells
// hold finite values -- outputs of earlier passes -- and fma(0, finite, acc) == acc).
#pragma once
#include <cstdint>

#define NEED_LAYERS 12
#define NEED_STAT_G 64                     // statistics rows: [layer][group] stored cells (decode order), [layer][group block] live tiles (encode order)

// ---- decode order: per (layer, plane, XCD) lists of task records for cconv4v6_dc.inc's LIST kernels.
// record = uint4: x = g0 | packed << 7 | n << 10   (packed 4: up to three pieces y / z / w, sample n + 8 k each; packed 0: sample n, whole diagonals);
//                 bits 22..24 of y: which of the block's three groups g0, g0 + 1, g0 + 2 are live on some row of the record (the others' wave sets idle)
// piece  = k | slo << 3 | shi << 9 | a0 << 15 | 1 << 21   (rows slo..shi of sample k of the chunk in lanes a0.., as the tape's pieces)
#define DCL_CHUNK 8                        // samples per packing chunk (k has 3 bits)

// The pieces of ONE wave: cconv4v6_dc.inc's dc6_wave_pieces with a row window PER SAMPLE (lo[k]..hi[k], hi < lo: nothing of that sample
// is live on this plane).  Same rules: (a0 - slo) % 4 == 0; the next piece starts in the quad behind this piece's last band column (last lane + 4);
// a0 >= 2 unless the piece starts with image row 0, last lane <= 61 unless it ends with the image's last row; a window may be cut between two
// waves; at most three pieces.  (k, slo) advance to the start of the next wave (k == c: all samples placed).
__host__ __device__ inline int dcl_wave_pieces(const int *lo, const int *hi, int h, int c, int &k, int &slo, unsigned out[3]) {
    int nwin = 0, pos = 0;
    out[0] = out[1] = out[2] = 0u;
    while (nwin < 3 && k < c) {
        if (hi[k] < lo[k]) { ++k; if (k < c) slo = lo[k]; continue; }
        int a0 = pos;
        if (slo != 0 && a0 < 2) a0 = 2;
        a0 += (slo - a0) & 3;
        const int top = hi[k] == h - 1 ? 63 : 61;
        int shi = hi[k];
        if (a0 + (hi[k] - slo) > top) shi = slo + (61 - a0);               // cut
        if (a0 > 61 || (shi < hi[k] && (a0 > 57 || shi - slo + 1 < 4))) break;   // no room (for a useful piece of a cut window)
        out[nwin++] = (unsigned)k | (unsigned)slo << 3 | (unsigned)shi << 9 | (unsigned)a0 << 15 | 1u << 21;
        pos = ((a0 + (shi - slo) + 4) / 4 + 1) * 4;
        slo = shi + 1;
        if (slo > hi[k]) { ++k; if (k < c) slo = lo[k]; }
    }
    return nwin;
}

// units per block of the encode kernels' task order (c16_fill_args and the list builder must agree): 16 measured best without lists (DESIGN 4.1 a);
// LIC360_EC_GBK overrides it for A/B runs
#include <cstdlib>
static inline int lic360_ec_gbk(void) {
    static const int v = [] { const char *e = getenv("LIC360_EC_GBK"); const int x = e ? atoi(e) : 0; return x > 0 && x <= 4096 ? x : 16; }();
    return v;
}

// internal entries (hidden visibility)
// need [B][12][H][W] and its diagonal-major copy need_d [B][12][H+W-1][H] (cell (y, x) at [(y + x) * H + y]; cells outside the image: -1),
// tile maxima tmax [B][12][nty][ntx] over the encode kernels' 4 x 16 tiles; all int8
int lic360_need_build(void *stream, const float *mask, int B, int G, int H, int W, signed char *need, signed char *need_d, signed char *tmax);
struct lic360_ec_lists {                   // encode order: compact lists of the live (sample, chunk of tiles, group block) tasks per layer and XCD
    int *list = nullptr;                   // [12][8][cap] entries u | tile mask << 28 in launch order
    int *cnt = nullptr;                    // [12][8]
    int cap = 0;
};
int lic360_ec_lists_build(void *stream, const signed char *tmax, int B, int G, int H, int W, const lic360_ec_lists &l, unsigned long long *stats);
struct lic360_dc_lists {                   // decode order
    uint4 *list = nullptr;                 // [12][P][8][cap]
    int *cnt = nullptr;                    // [12][P][8]
    int cap = 0, P = 0;
};
int lic360_dc_lists_build(void *stream, const signed char *need_d, int B, int G, int H, int W, const lic360_dc_lists &l, unsigned long long *stats);
