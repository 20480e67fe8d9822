// conv_plan.h -- weight-independent schedule of one group-causal masked conv layer shape.
#pragma once
#include <vector>
#include <cstdint>

// One K-step record = 4 consecutive terms of one virtual lane's fmaf chain (K dim of
// v_mfma_f32_16x16x4_f32) for a 16-row output tile.
struct lic360_conv_plan {
    int C, ngroup, nout, ksz, constrain, cin, cout, half;
    int n_mtiles;
    long total_rec;                       // K-step records over all output tiles (per stacked net)
    std::vector<int> mt_rec_start;        // [n_mtiles+1]
    std::vector<int> leaf_cnt;            // [n_mtiles][128] K-steps per leaf, VISITING order r (lane = bitrev7(r))
    std::vector<int> term;                // [total_rec+PAD][4]  ti | kh<<16 | kw<<24
    std::vector<int> wsrc;                // [total_rec][64]     flat index into weight[nout][C][k][k], -1 = zero
    std::vector<int> mt_glo, mt_ghi;      // group range covered by each output tile
    int *d_mt_rec_start = nullptr, *d_leaf_cnt = nullptr, *d_term = nullptr, *d_wsrc = nullptr;
    int *d_mt_glo = nullptr, *d_mt_ghi = nullptr;
};
static const int LIC360_REC_PAD = 4;       // records readable past the end (software prefetch)

// internal entries shared between translation units (hidden visibility; not part of include/lic360_hip.h)
int lic360_cconv4_dc_plane_mode(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod, int mode);
int lic360_dc4_env_mode(void);
int lic360_cconv_dc_plane_strided(void *stream, const lic360_conv_plan *p, const float *x, const float *packed, const float *bias,
                                  const float *act, const float *residual, float *out, int n, int h, int w, int nb,
                                  const int *idx_dev, const int *plane_idx_dev, const int *plane_idx_host, int psum, int x_mod,
                                  long x_cs, long x_hs, long x_ws, long o_cs, long o_hs, long o_ws);
// the dead-cone skip's entries (round 6; need.h): the encode-order kernels over a compacted list of live tasks, the decode-order kernel over the
// (layer, plane)'s task records
int lic360_cconv16_ec_list(void *stream, const lic360_conv_plan *p, const float *x, const float *packed16, const float *bias,
                           const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod, int *ctr,
                           const int *list, const int *cnt, int cap);
int lic360_cconv16_ec_tables_list(void *stream, const lic360_conv_plan *p, const float *x, const float *packed16, const float *bias,
                                  const float *code, const float *mask, const int *pidx_dev, const int *plane_start_dev,
                                  void *rec, int images, int h, int w, int *ctr, const int *list, const int *cnt, int cap);
int lic360_cconv4_dc_plane_list(void *stream, const lic360_conv_plan *p, const float *x, const float *packed4, const float *bias,
                                const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod,
                                const void *list, const int *cnt, int cap);
