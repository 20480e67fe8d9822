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
// Decode order only (cconv4v6_dc.inc): persistent workgroups of 12 waves = 3 adjacent groups (adjacent anti-diagonals of the
// current plane) of one sample.  Encode order runs on the 16x16x4 kernels of cconv16_kernels.hip; the 4x4x1 encode kernels and
// the LDS-DMA generation of the decode kernel were retired in round 5 (git show c0e481c).
#include "common.h"
#include "conv_plan.h"

#include "cconv_tree.h"
#include "need.h"

#define C4_COLS 68                       // 64 positions + 2*2 halo
#define C4_WSLOTS 128                    // weight slots per step (>= 25*cin), 4 floats each

static inline bool conv4_ok(const lic360_conv_plan *p) {
    return p->ksz == 5 && (p->cin == 1 || p->cin == 4) && p->cout >= 1 && p->cout <= 4 && p->ngroup <= 64;
}

// ------------------------------------------------------------------------------------------------
// Weight element (net b, output group g, input group tc, class c, leaf i, row r): class c = virtual lane mod 4, leaf i (0..31) inside
// the class (cin = 4: tap = i, gid = (c - tap) mod 4;  cin = 1: tap = c + 4*i), r = output channel within the group; lane (i%16)*4 + r of
// register i/16 holds it -- exactly the 4-lane block that `abid = i%16` selects for broadcast (cbsz = 4) in v_mfma_f32_4x4x1_16b_f32.
// Packed so that the registers a wave needs for one DOUBLE step are adjacent per lane: it fetches them with one (cin = 4) or two (cin = 1)
// 16-byte loads instead of four / eight 4-byte ones -- the CU's address unit takes a wave instruction at a time whatever its width, and the
// decode kernel's twelve waves kept it busy 864 of a double step's ~2900 cycles:
//   packed4[net][g][blk][c][lane][4]   cin = 4: blk = tc / 2, word = 2 (tc & 1) + register;   cin = 1: blk = tc / 4, word = tc & 3 (register 0 only)
// wblk 4 KB blocks per output group (zero-filled past the last input group).
__global__ void k_conv4_pack(const float *__restrict__ weight, float *__restrict__ quads, int wblk, int nb, int ngroup, int cin, int cout, int hidden) {
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
        const long qb = (long)(b * ngroup + g) * wblk;
        const int ln = (leaf & 15) * 4 + r;
        if (cin == 4) quads[(qb + (tc >> 1)) * 1024 + c * 256 + ln * 4 + (tc & 1) * 2 + (leaf >> 4)] = v;
        else if (leaf < 16) quads[(qb + (tc >> 2)) * 1024 + c * 256 + ln * 4 + (tc & 3)] = v;
    }
}

LIC360_API int lic360_conv4_supported(const lic360_conv_plan *p) { return p && conv4_ok(p) ? 1 : 0; }
// a packed4 buffer of nb nets = nb x (ngroup x wblk x 1024 floats)
static inline long conv4_slot_floats(const lic360_conv_plan *p) { return (long)p->ngroup * p->ngroup * C4_WSLOTS * 4; }   // elements the pack kernel walks
static inline int conv4_wblk(const lic360_conv_plan *p) { return p->cin == 4 ? (p->ngroup + 1) / 2 : 2 * ((p->ngroup + 7) / 8); }
static inline long conv4_quad_floats(const lic360_conv_plan *p) { return (long)p->ngroup * conv4_wblk(p) * 1024; }
LIC360_API long lic360_conv4_packed_floats(const lic360_conv_plan *p) {
    return p && conv4_ok(p) ? conv4_quad_floats(p) : 0;
}
LIC360_API int lic360_conv4_pack(void *stream, const lic360_conv_plan *p, const float *weight, int nb, float *packed) {
    ARG_CHECK(p && conv4_ok(p) && weight && packed && nb > 0);
    const long total = conv4_slot_floats(p) * nb;
    HIP_TRY(hipMemsetAsync(packed, 0, (size_t)conv4_quad_floats(p) * nb * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(k_conv4_pack, dim3(lic360_blocks(total, 4)), dim3(256), 0, (hipStream_t)stream, weight, packed, conv4_wblk(p), nb, p->ngroup,
                       p->cin, p->cout, p->constrain == 5 ? 0 : 1);
    LAUNCH_CHECK();
    return 0;
}

#define C4_PS 3                          // position sets (rows / diagonals) per workgroup
#define C4_THREADS (C4_PS * 4 * 64)      // 12 waves: 3 per SIMD -> 168 VGPRs each, no spills with 25 resident accumulators
// final two tree levels across the 4 lane classes of a position set + epilogue handled by the caller
#define C4_COMB_FLOATS (C4_PS * 4 * 4 * 64)

// ------------------------------------------------------------------------------------------------ DC4
#include "cconv4v6_dc.inc"

// mode: bit 1 = no several-samples-per-wave packing, bits 2-3 = task granularity (4: one group per task always, 8: never; 0: by task count).
// Internal entry (hidden visibility) so that the fused codec passes the switches it read ONCE at create time.
int lic360_cconv4_dc_plane_mode(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod, int mode) {
    ARG_CHECK(p && conv4_ok(p) && x && packed4 && bias && out && n > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n);
    if (psum < 0 || psum >= h + w + p->ngroup - 2) return 0;
    return launch_cconv4v6_dc((hipStream_t)stream, p, x, packed4, bias, act, residual, out, n, h, w, nb, psum, x_mod, (mode & 2) != 0,
                              (mode >> 2) & 3);
}
int lic360_cconv4_dc_plane_list(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod,
                                const void *list, const int *cnt, int cap) {
    ARG_CHECK(p && conv4_ok(p) && x && packed4 && bias && out && n > 0 && nb > 0 && n % nb == 0 && x_mod > 0 && x_mod <= n);
    if (psum < 0 || psum >= h + w + p->ngroup - 2) return 0;
    return launch_cconv4v6_dc_list((hipStream_t)stream, p, x, packed4, bias, act, residual, out, n, h, w, nb, psum, x_mod, (const uint4 *)list, cnt, cap);
}
// LIC360_NOPACK / LIC360_DC_GSTEP force the schedule variants (both are tested against the oracle); read once per process, not per launch
int lic360_dc4_env_mode(void) {
    const char *gsm = getenv("LIC360_DC_GSTEP");                     // "1": one group per task always, "3": never (default: by task count)
    return (getenv("LIC360_NOPACK") ? 2 : 0) | (gsm && gsm[0] == '1' ? 4 : (gsm && gsm[0] == '3' ? 8 : 0));
}
LIC360_API int lic360_cconv4_dc_plane(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                      const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod) {
    static const int mode = lic360_dc4_env_mode();
    return lic360_cconv4_dc_plane_mode(stream, p, x, packed4, bias, act, residual, out, n, h, w, nb, psum, x_mod, mode);
}

// Floats to allocate for `planes` activation planes (samples x channels) of the padded decode-order layout (lic360_dc4_layout): the
// planes plus the slack the band fetches of the last plane may touch (they are whole 16-byte quads of 11-row bands and are not clamped
// at the end of the tensor).  Callers size and zero buffers with this instead of knowing the slack.  layout: 0 (the only one).
#define C4_TAIL_FLOATS 4096
LIC360_API long lic360_conv4_buffer_floats(int layout, long planes, int h, int w) {
    if (planes <= 0 || h <= 0 || w <= 0 || layout != 0) return 0;
    return planes * ((long)D3_SP(h, w) * D3_HP(h)) + C4_TAIL_FLOATS;
}
LIC360_API int lic360_dc4_layout(int h, int w, int *rows, int *pitch, int *row0, int *col0) {
    ARG_CHECK(rows && pitch && row0 && col0 && h > 0 && w > 0);
    *rows = D3_SP(h, w); *pitch = D3_HP(h); *row0 = D3_S0; *col0 = D3_C0;
    return 0;
}
