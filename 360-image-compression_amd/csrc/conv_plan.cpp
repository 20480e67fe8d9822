// conv_plan.cpp -- builds the term schedule for the MFMA masked-conv kernels.
//
// Canonical evaluation order being reproduced (extension/cconv_ec_cuda.cu:268-315):
// 128 virtual lanes; lane l walks flat tap indices l, l+128, ... < k*k*cin, index -> (kw, kh, gid);
// for each index a chain over ti = gid, gid+cin, ... < nchannel(o, kh, kw); one running sum per lane;
// then the fixed tree  p[i]+p[i+64]; +32; +16; +8; +4; +2; +1.
//
// Mapping to v_mfma_f32_16x16x4_f32 (bit-for-bit a k-ordered fmaf chain): for a tile of 16 output
// channels the chain of one lane becomes the K loop; terms that are causally masked for EVERY row
// of the tile are dropped (they are skipped by the reference too), terms masked for some rows
// get a zero weight (fma(x, 0, s) == s).  Lanes are visited in 7-bit bit-reversed order so that
// the tree is a binary-counter merge with at most 7 live partial tiles.
#include "common.h"
#include "conv_plan.h"
#include <algorithm>

static int bitrev7(int r) {
    int l = 0;
    for (int b = 0; b < 7; ++b) if (r & (1 << b)) l |= 1 << (6 - b);
    return l;
}

static int nchannel_for_group(const lic360_conv_plan &p, int g, int kh, int kw) {
    // psum - ph - pw = g + 2*half - kh - kw   (cconv_ec_cuda.cu:280-288)
    int v = g + 2 * p.half - kh - kw + (p.constrain == 5 ? 0 : 1);
    long n = (long)v * p.cin;
    if (n > p.C) n = p.C;
    return (int)n;
}

template <class T>
static int upload(const std::vector<T> &v, T **dst) {
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void **)dst, bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

LIC360_API int lic360_conv_plan_create(int channel, int ngroup, int nout, int ksz, int constrain, lic360_conv_plan **out) {
    ARG_CHECK(out && channel > 0 && ngroup > 0 && nout > 0 && ksz > 0 && (ksz & 1));
    ARG_CHECK(channel % ngroup == 0 && nout % ngroup == 0 && (constrain == 5 || constrain == 6));
    ARG_CHECK(channel < 65536 && ksz < 256);
    lic360_conv_plan *p = new lic360_conv_plan();
    p->C = channel; p->ngroup = ngroup; p->nout = nout; p->ksz = ksz; p->constrain = constrain;
    p->cin = channel / ngroup; p->cout = nout / ngroup; p->half = ksz / 2;
    p->n_mtiles = (nout + 15) / 16;
    const int nblock = ksz * ksz * p->cin;
    p->mt_rec_start.assign(p->n_mtiles + 1, 0);
    p->leaf_cnt.assign((size_t)p->n_mtiles * 128, 0);
    long rec = 0;
    for (int mi = 0; mi < p->n_mtiles; ++mi) {
        p->mt_rec_start[mi] = (int)rec;
        int o0 = mi * 16, o1 = std::min(o0 + 15, nout - 1);
        int glo = o0 / p->cout, ghi = o1 / p->cout;
        p->mt_glo.push_back(glo);
        p->mt_ghi.push_back(ghi);
        for (int r = 0; r < 128; ++r) {
            int lane = bitrev7(r);
            std::vector<int> terms;                 // packed (ti, kh, kw) in canonical order
            for (int index = lane; index < nblock; index += 128) {
                int kw = index % ksz, kh = (index / ksz) % ksz, gid = index / ksz / ksz;
                int nmax = nchannel_for_group(*p, ghi, kh, kw);
                for (int ti = gid; ti < nmax; ti += p->cin) terms.push_back(ti | (kh << 16) | (kw << 24));
            }
            int nrec = ((int)terms.size() + 3) / 4;
            p->leaf_cnt[(size_t)mi * 128 + r] = nrec;
            for (int s = 0; s < nrec; ++s) {
                for (int k = 0; k < 4; ++k) {
                    int idx = s * 4 + k;
                    bool live = idx < (int)terms.size();
                    int tm = live ? terms[idx] : (0 | (p->half << 16) | (p->half << 24));
                    p->term.push_back(tm);
                }
                for (int k = 0; k < 4; ++k) {
                    int idx = s * 4 + k;
                    bool live = idx < (int)terms.size();
                    int tm = live ? terms[idx] : 0;
                    int ti = tm & 0xffff, kh = (tm >> 16) & 0xff, kw = (tm >> 24) & 0xff;
                    for (int i = 0; i < 16; ++i) {
                        int o = o0 + i, src = -1;
                        if (live && o < nout && ti < nchannel_for_group(*p, o / p->cout, kh, kw))
                            src = ((o * channel + ti) * ksz + kh) * ksz + kw;
                        (void)k;
                        p->wsrc.push_back(src);
                    }
                }
                ++rec;
            }
        }
    }
    p->mt_rec_start[p->n_mtiles] = (int)rec;
    p->total_rec = rec;
    // wsrc was pushed as [rec][k][i] = slot k*16+i : matches the A-fragment lane map of 16x16x4 (lane = k*16 + i)
    for (int i = 0; i < LIC360_REC_PAD * 4; ++i) p->term.push_back(0 | (p->half << 16) | (p->half << 24));
    int rc = 0;
    rc |= upload(p->mt_rec_start, &p->d_mt_rec_start);
    rc |= upload(p->leaf_cnt, &p->d_leaf_cnt);
    rc |= upload(p->term, &p->d_term);
    rc |= upload(p->wsrc, &p->d_wsrc);
    rc |= upload(p->mt_glo, &p->d_mt_glo);
    rc |= upload(p->mt_ghi, &p->d_mt_ghi);
    if (rc) { lic360_conv_plan_destroy(p); return 1; }
    *out = p;
    return 0;
}

LIC360_API void lic360_conv_plan_destroy(lic360_conv_plan *p) {
    if (!p) return;
    (void)hipFree(p->d_mt_rec_start); (void)hipFree(p->d_leaf_cnt); (void)hipFree(p->d_term); (void)hipFree(p->d_wsrc);
    (void)hipFree(p->d_mt_glo); (void)hipFree(p->d_mt_ghi);
    delete p;
}

LIC360_API long lic360_conv_plan_packed_floats(const lic360_conv_plan *p) {
    return p ? (p->total_rec + LIC360_REC_PAD) * 64 : 0;
}
