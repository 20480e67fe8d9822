/* lic360_hip.h -- C ABI of liblic360_hip.so, the MI355X (gfx950) implementation of the
 * LIC360 spherical-tiling + latent entropy-coding hot path.
 *
 * This is the drop-in boundary: each entry point replaces one method of a
 * class bound in the reference's pybind11 module `lic360` (extension/main.cpp:4-178).  The
 * citation after each declaration names the reference interface it replaces.  Conventions:
 *   - every pointer named *_dev / x / out ... is DEVICE memory (fp32, contiguous NCHW) unless
 *     the name says host; `stream` is the caller's hipStream_t passed as void* (the reference
 *     captured the ATen stream at construction, extension/base_opt.hpp:21-23);
 *   - plain C types only -- no torch / ATen types cross this boundary;
 *   - return value 0 = ok, non-zero = error; lic360_last_error() returns the message (the
 *     reference printed CUDA errors and carried on, extension/caffe_cuda_macro.h:21-33);
 *   - nothing here falls back to the CPU: if no HIP device is usable the call fails.
 */
#ifndef LIC360_HIP_H
#define LIC360_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *lic360_last_error(void);
int lic360_version(void);

/* ---- A1-A3 sphere ops ------------------------------------------------------------------ */
/* SpherePadOp.forward, not in place          extension/sphere_pad_cuda.cu:67-105 (kernel :29-46) */
int lic360_sphere_pad(void *stream, const float *x, float *out, int nc, int h, int w, int pad);
/* SpherePadOp.forward, inplace=true          extension/sphere_pad_cuda.cu:48-65; hp,wp = padded dims */
int lic360_sphere_pad_inplace(void *stream, float *x, int nc, int hp, int wp, int pad);
/* SphereTrimOp.forward then SpherePadOp.forward(inplace) of the same width on the same tensor, as the reference's blocks run them back to back
 * (test/model_zoo.py:83-84,90-91,160-161): the refresh overwrites every cell the trim zeroed, so the pair is ONE pass over the apron */
int lic360_sphere_trim_pad_inplace(void *stream, float *x, int nc, int hp, int wp, int pad);
/* SphereTrimOp.forward                       extension/sphere_trim_cuda.cu:28-46 */
int lic360_sphere_trim(void *stream, float *x, int nc, int h, int w, int pad);
/* SphereCutEdgeOp.forward                    extension/sphere_cut_edge_cuda.cu:43-62 */
int lic360_sphere_cut_edge(void *stream, const float *x, float *out, int nc, int h, int w, int pad);
/* SpherePadOp.backward: gradient of the un-padded tensor [nc][h][w] from top_diff [nc][h+2p][w+2p]: own cell + wrap column + mirrored pole
 * row + pole corner, added in that order     extension/sphere_pad_cuda.cu:107-136,181-198 */
int lic360_sphere_pad_backward(void *stream, float *in_diff, const float *top_diff, int nc, int h, int w, int pad);
/* SpherePadOp.backward, inplace=true: interior cells of diff [nc][hp][wp] accumulate their apron copies, the apron stays
 *                                            extension/sphere_pad_cuda.cu:138-180 */
int lic360_sphere_pad_backward_inplace(void *stream, float *diff, int nc, int hp, int wp, int pad);
/* SphereCutEdgeOp.backward: in_diff [nc][h][w] = top_diff [nc][h-2p][w-2p] inside a zero apron   extension/sphere_cut_edge_cuda.cu:63-96 */
int lic360_sphere_cut_edge_backward(void *stream, float *in_diff, const float *top_diff, int nc, int h, int w, int pad);
/* SphereLatScaleOp.forward / backward (same product)  extension/sphere_lat_scale_cuda.cu:40-58,69-87 */
int lic360_sphere_lat_scale(void *stream, const float *x, const float *weight, float *out, int nc, int h, int w, int npart);

/* ---- A4-A7, A18 pointwise ops ---------------------------------------------------------- */
/* ImpMapOp.forward (mask may be NULL)        extension/imp_map_cuda.cu:112-136 */
int lic360_imp_map(void *stream, const float *x, const float *imp, float *out, float *mask, int n, int c, int h, int w, int levels);
/* ImpMapOp constraint tensor top[1]          extension/imp_map_cuda.cu:27-71 (host computes, device write) */
int lic360_imp_map_constrain(void *stream, float *constrain, int n, int h, float rt, float scale_constrain);
/* alpha_t [h] of ImpMapOp: alpha / (|cos((0.5-(i+0.5)/h) pi)| / max * scale_weight + 1 - scale_weight) (host computes, device write)
 *                                            extension/imp_map_cuda.cu:27-36,49-52 */
int lic360_imp_map_alpha(void *stream, float *alpha_t, int h, float alpha, float scale_weight);
/* ImpMapOp.backward: data_diff [n,c,h,w] = top_diff under the mask floor(imp*levels); imp_diff [n,1,h,w] by rule imp_kernel 0..3
 * (kernels v1..v4), channel sums in ascending order; sphere_constrain [n,h], alpha_t [h]
 *                                            extension/imp_map_cuda.cu:138-298 */
int lic360_imp_map_backward(void *stream, const float *top_diff, const float *imp, const float *sphere_constrain, const float *alpha_t,
                            float *data_diff, float *imp_diff, int n, int c, int h, int w, int levels, int imp_kernel, float gamma);
/* Imp2maskOp.forward                         extension/imp2mask_cuda.cu:41-57 */
int lic360_imp2mask(void *stream, const float *x, float *out, int n, int c, int h, int w, int cpn);
/* MaskConstrainOp.forward / .backward (in place on a conv weight or its gradient [nout][channel][ksz][ksz]; constrain 5: taps with
 * tw + th + tc >= tn + ksz - 1 are zeroed, 6: tw + th + tc > tn + ksz - 1; tc / tn = input / output group)
 *                                            extension/mask_constrain_cuda.cu:17-41,47-91 */
int lic360_mask_constrain(void *stream, float *weight, int nout, int channel, int ksz, int ngroup, int constrain);
/* ScaleOp.forward                            extension/scale_cuda.cu:32-48 */
int lic360_scale(void *stream, const float *x, float *out, long count, float bias, float scale);
/* QuantOp.forward (train=false path); qidx may be NULL; wq_scratch/count are [c,levels] device buffers
 *                                            extension/quant_cuda.cu:136-169 */
int lic360_quant(void *stream, const float *x, const float *weight_b, float *wq_scratch, float *top, float *qidx, float *count,
                 int n, int c, int h, int w, int levels);
/* QuantOp training side.  update_weight: trailing levels without samples share one increment, an empty first level moves the first
 * centre, then the counts decay by weight_decay (in place on weight [c,levels] and ncount [c,levels])
 *                                            extension/quant_cuda.cu:87-133 */
int lic360_quant_update_weight(void *stream, float *weight, float *ncount, int c, int levels, float weight_decay);
/* QuantOp.backward: weight_diff [c,levels] = per-channel sums of (top_data - bottom_data) over the elements with index >= j (times the
 * level increment for j > 0; summed per workgroup in a fixed order, the reference uses float atomics), data_diff = top_diff0
 * (+ top_alpha * top_diff1 / beta when top_diff1 != NULL); qidx = the indices of the forward pass as floats, wq = its level increments
 *                                            extension/quant_cuda.cu:170-262 */
int lic360_quant_backward(void *stream, const float *top_diff0, const float *top_diff1, const float *bottom_data, const float *top_data,
                          const float *qidx, const float *wq, float *data_diff, float *weight_diff, int n, int c, int h, int w, int levels,
                          float top_alpha);
/* DquantOp.forward                           extension/dquant_cuda.cu:49-68 */
int lic360_dquant(void *stream, const float *x, const float *mask, const float *weight_b, float *wc_scratch, float *out,
                  int n, int c, int h, int w, int levels);
/* DtowOp.forward (d2w: pixel shuffle, else unshuffle)   extension/dtow_cuda.cu:77-102 */
int lic360_dtow(void *stream, const float *x, float *out, int n, int c, int h, int w, int stride, int d2w);

/* ---- A17 layout ops ---------------------------------------------------------------------- */
/* ContextReshapeOp.forward / backward(inverse=1)  extension/context_reshape_cuda.cu:42-60,74-92 */
int lic360_context_reshape(void *stream, const float *x, float *out, int n, int c, int h, int w, int ngroup, int inverse);
/* ContexShiftOp.forward (inv=0 zero-fills the skewed tensor first)  extension/contex_shift_cuda.cu:65-90 */
int lic360_contex_shift(void *stream, const float *x, float *out, int n, int c, int hin, int w, int cpn, int inv);

/* ---- A8 scan order ----------------------------------------------------------------------- */
/* CodeContexOp.forward: HOST tables idx[2*h*w], plane_idx[h+w]   extension/code_contex_cuda.cu:11-38 */
int lic360_code_contex(int h, int w, int *idx_host, int *plane_idx_host);
/* window [start,len) of plane psum in the scan order (extension/cconv_dc_cuda.cu:374-376); len 0 past the end */
int lic360_plane_window(int psum, int ngroup, int h, int w, const int *plane_idx_host, int *start, int *len);

/* ---- A11-A13 plane gather / scatter / add -------------------------------------------------- */
/* TileExtractOp.forward gather of plane psum -> out[n][len][cpn]   extension/tile_extract_cuda.cu:31-45,48-98 */
int lic360_tile_extract(void *stream, const float *x, float *out, int n, int c, int h, int w, int ngroup,
                        const int *idx_dev, int start, int len, int psum);
/* TileExtractOp.forward_batch (3 stacked nets, slab stride cpn*h*w*(n/3))  extension/tile_extract_cuda.cu:101-151 */
int lic360_tile_extract_batch(void *stream, const float *x, float *out, int n, int c, int h, int w, int ngroup,
                              const int *idx_dev, int start, int len, int psum);
/* TileInputOp.forward scatter of plane psum (already decremented) -> out[rep*n,g,h,w]  extension/tile_input_cuda.cu:27-76 */
int lic360_tile_input(void *stream, const float *sym, float *out, int n, int g, int h, int w, float bias, float scale, int rep,
                      const int *idx_dev, int start, int len, int psum);
/* TileAddOp.forward y += x on plane psum      extension/tile_add_cuda.cu:22-60 */
int lic360_tile_add(void *stream, float *y, const float *x, int n, int c, int h, int w, int ngroup,
                    const int *idx_dev, int start, int len, int psum);

/* ---- A14-A16 probability tables ------------------------------------------------------------ */
/* EntropyGmmTableOp.forward / forward_batch: w,d are rewritten in place (softmax, sigma floor);
 * out float[tn][nstep+1].  For forward_batch pass w=data, d=data+stride, m=data+2*stride.
 *                                            extension/entropy_gmm_table_cuda.cu:109-135,161-191 */
int lic360_gmm_table(void *stream, float *w, float *d, const float *m, float *out, int tn, int ng, int nstep,
                     float bias, float total, float beta);
/* EntropyTableOp.forward                     extension/entropy_table_cuda.cu:78-96 */
int lic360_entropy_table(void *stream, const float *logits, float *out, int count, int nstep, float total);
/* EntropyGmmOp.forward (loss + stored grads) / backward (scale stored grads by top_diff)
 *                                            extension/entropy_gmm_cuda.cu:71-91,108-124 */
int lic360_entropy_gmm(void *stream, const float *w, const float *d, const float *m, const float *label, float *loss,
                       float *wd, float *dd, float *md, float *ld, int count, int ng);
int lic360_entropy_gmm_backward(void *stream, float *wd, float *dd, float *md, float *ld, const float *top_diff, int count, int ng);

/* ---- A9/A10 group-causal masked convolution ------------------------------------------------ */
/* A plan holds the weight-independent schedule of one layer shape (term lists per 16-channel
 * output tile in the reference's 128-lane tree order) in host and device memory. */
typedef struct lic360_conv_plan lic360_conv_plan;
/* shape of CconvEcOp/CconvDcOp ctor (channel, ngroup, nout, kernel_size, constrain)  extension/cconv_ec.hpp:7-16 */
int lic360_conv_plan_create(int channel, int ngroup, int nout, int ksz, int constrain, lic360_conv_plan **plan);
void lic360_conv_plan_destroy(lic360_conv_plan *plan);
/* floats of packed weights per stacked net */
long lic360_conv_plan_packed_floats(const lic360_conv_plan *plan);
/* re-layout weight[nb][nout][channel][k][k] into MFMA A-fragments packed[nb][packed_floats] */
int lic360_conv_pack(void *stream, const lic360_conv_plan *plan, const float *weight, int nb, float *packed);
/* CconvEcOp.forward / forward_act / forward_batch / forward_act_batch (act NULL = no PReLU; nb = weight.size(0) or 1)
 *                                            extension/cconv_ec_cuda.cu:99-121,170-192,242-265,317-339 */
int lic360_cconv_ec(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed, const float *bias,
                    const float *act, float *out, int n, int h, int w, int nb);
/* CconvDcOp.forward* for ONE plane psum: writes the plane's outputs into the persistent out[n,nout,h,w]
 * (caller zero-fills at psum==0 like :385).  plane_idx_dev = device copy of CodeContex's prefix table.
 *                                            extension/cconv_dc_cuda.cu:108-138,193-224,279-310,367-398 */
int lic360_cconv_dc_plane(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed, const float *bias,
                          const float *act, float *out, int n, int h, int w, int nb,
                          const int *idx_dev, const int *plane_idx_dev, const int *plane_idx_host, int psum);

/* Extended forms used by the fused codec: `residual` (same layout as out, may be NULL) is added after the
 * activation (EntropyResidualBlock*: `conv2(conv1(x)) + x`, test/lic360_demo.py:29-41, extension/tile_add_cuda.cu:35);
 * x_mod < n lets the stacked nets share one input (the reference concatenates three copies, lic360_demo.py:131);
 * skewed != 0 selects the diagonal-major activation layout [n][c][th+tw][th] for x, residual and out. */
int lic360_cconv_ec_ex(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed, const float *bias,
                       const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod);
int lic360_cconv_dc_plane_ex(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed, const float *bias,
                             const float *act, const float *residual, float *out, int n, int h, int w, int nb,
                             const int *idx_dev, const int *plane_idx_dev, const int *plane_idx_host, int psum, int x_mod, int skewed);

/* Decode order, leaf-resident variant (v_mfma_f32_4x4x1, csrc/cconv4_kernels.hip + cconv4v6_dc.inc) for the latent-net shapes cin in {1,4},
 * cout <= 4, ngroup <= 64: replaces CconvDcOp.forward_act_batch / forward_batch of one plane (extension/cconv_dc_cuda.cu:313-398), same
 * results bit for bit as lic360_cconv_dc_plane, own weight layout.  Activations: zero-padded diagonal-major [n][c][rows][pitch] with cell
 * (s = th+tw, th) at [(s + row0) * pitch + th + col0] (lic360_dc4_layout; the padding must be zero).  Buffers are sized with
 * lic360_conv4_buffer_floats(0, planes, h, w) -- planes = samples x channels -- which includes the slack the 16-byte band fetches of the
 * last plane may touch (whole quads of 11-row bands, not clamped at the end of the tensor); the whole buffer starts zeroed. */
long lic360_conv4_buffer_floats(int layout, long planes, int h, int w);
int lic360_dc4_layout(int h, int w, int *rows, int *pitch, int *row0, int *col0);
int lic360_conv4_supported(const lic360_conv_plan *plan);
/* packed4: nb * lic360_conv4_packed_floats(plan) floats: per (net, output group) blocks of 4 KB in which the registers a wave needs for one
 * double step are adjacent per lane (one or two 16-byte loads per wave and double step) */
long lic360_conv4_packed_floats(const lic360_conv_plan *plan);
int lic360_conv4_pack(void *stream, const lic360_conv_plan *plan, const float *weight, int nb, float *packed4);
int lic360_cconv4_dc_plane(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed4, const float *bias,
                           const float *act, const float *residual, float *out, int n, int h, int w, int nb, int psum, int x_mod);
/* Host-only (no GPU work): the sample packing lic360_cconv4_dc_plane uses on plane psum for a layer of ngroup groups (cin = 4: hidden / last
 * layers, 1: first layer) over n samples of nb nets -- "tape" packing: the row windows of *tape_c consecutive samples of an XCD's list laid end to
 * end over the 64 lanes of nwaves[j] tasks per group block j (a window may be cut between two tasks).  *tape_c = 0: one sample per task on this
 * plane.  blocks[j] = first group of block j (launch order); windows[(j * 6 + t) * 3 + i] = piece i of task t of a tape:
 * k | slo << 3 | shi << 9 | a0 << 15 | 1 << 21 (sample k of the tape stores rows slo..shi from lanes a0..; 0 = none).  Arrays of 24, 24 and
 * 24 * 6 * 3 entries (the launch itself carries only where each task of a tape starts; the pieces follow from one rule on both sides).  For tests of the packing rules (tests/test_dc_tape.py). */
int lic360_dc4_tape_layout(int ngroup, int cin, int n, int nb, int h, int w, int psum, int x_mod, int *tape_c, int *n_blocks, int *blocks,
                           int *nwaves, unsigned *windows);

/* Encode-order variant on v_mfma_f32_16x16x4_f32 (csrc/cconv16_kernels.hip): rows = 4 consecutive groups x 4 output channels, K = 4
 * consecutive input groups of a lane's chain; same results bit for bit as lic360_cconv_ec (extension/cconv_ec_cuda.cu:271-331).
 * Shapes: cin in {1,4}, cout <= 4 (the latent nets).  Activations are zero-haloed NCHW planes [n][c][hp][wp], cell (r, c) at
 * [(r+2)*wp + c+2] (lic360_ec16_layout: hp = 4*ceil(h/4)+4, wp = 16*ceil(w/16)+4); halo and round-up cells must be zero
 * and are never written.  ctr: 8 ints of device scratch (per-XCD task counters, zeroed on the stream by the call). */
int lic360_ec16_layout(int h, int w, int *hp, int *wp);
int lic360_conv16_supported(const lic360_conv_plan *plan);
long lic360_conv16_packed_floats(const lic360_conv_plan *plan);
int lic360_conv16_pack(void *stream, const lic360_conv_plan *plan, const float *weight, int nb, float *packed16);
/* the weight layout lic360_cconv16_ec_tables reads (cin = 4, cout = 3): five groups per block, MFMA row = 3 (group in block) + channel,
 * so that 15 of 16 MFMA rows carry an output; fits a lic360_conv16_packed_floats allocation */
int lic360_conv16_pack_tables(void *stream, const lic360_conv_plan *plan, const float *weight, int nb, float *packed16);
int lic360_cconv16_ec(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed16, const float *bias,
                      const float *act, const float *residual, float *out, int n, int h, int w, int nb, int x_mod, int *ctr);

/* Last layer of the latent entropy model with the CDF-table build fused into its epilogue (SURVEY.md §7 k_cconv_ec_last_gmm;
 * replaces the last CconvEcBatch.forward + TileExtractBatch + EntropyBatchGmmTable.forward_batch of
 * test/lic360_demo.py:132-140, extension/entropy_gmm_table_cuda.cu:138-191): x = activations of the three stacked nets
 * [weight, sigma, mu], [3*images][C][hp][wp] net-major in the lic360_ec16_layout; code / mask [images, G, h, w]; pidx_dev /
 * plane_start_dev = device copies of CodeContex's prefix table [h+w] and of the index of each plane's first record
 * [h+w+G-1]; rec = uint32 pairs [images][G*h*w] (cdf[sym], cdf[sym+1]) in coding order, (0,0) where masked.
 * packed16: from lic360_conv16_pack_tables (NOT lic360_conv16_pack). */
int lic360_cconv16_ec_tables(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed16, const float *bias,
                             const float *code, const float *mask, const int *pidx_dev, const int *plane_start_dev,
                             void *rec, int images, int h, int w, int *ctr);

/* The importance-map net's hidden / last layers (one group, C = 144, constrain 6: test/lic360_demo.py:153-161, 254-262) on
 * v_mfma_f32_16x16x4_f32 (csrc/cconv144_kernels.hip): same results bit for bit as lic360_cconv_ec / lic360_cconv_dc_plane.
 * Encode order: x = zero-haloed NCHW planes [n][144][hp][wp] (lic360_ec144_layout, cell (r, c) at [(r+2)*wp + c+2]); out /
 * residual = [n][nout] planes with stride oplane, rows of opitch floats, cell (r, c) at [(r+ooff)*opitch + c+ooff] (ooff 2 =
 * the same haloed layout, 0 = plain NCHW).  Decode order (plane = anti-diagonal s of every map): x / residual / out are
 * zero-padded diagonal-major planes [n][c][rows][pitch] (lic360_dc144_layout), cell (th, tw) at [(th+tw+4)*pitch + th+2]. */
int lic360_conv144_supported(const lic360_conv_plan *plan);
long lic360_conv144_packed_floats(const lic360_conv_plan *plan);
int lic360_conv144_pack(void *stream, const lic360_conv_plan *plan, const float *weight, float *packed144);
int lic360_ec144_layout(int h, int w, int *hp, int *wp);
int lic360_cconv144_ec(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed144, const float *bias,
                       const float *act, const float *residual, float *out, int n, int h, int w, long oplane, int opitch, int ooff);
int lic360_dc144_layout(int h, int w, int *rows, int *pitch);
int lic360_cconv144_dc_plane(void *stream, const lic360_conv_plan *plan, const float *x, const float *packed144, const float *bias,
                             const float *act, const float *residual, float *out, int n, int h, int w, int s);

/* ---- A19/A20 arithmetic coder (host side, as in the reference) ----------------------------- */
/* Coder operates on HOST int32 tables exactly like the reference's Coder (extension/coder.h:10-63,
 * extension/coder.cpp:30-113): the op API hands it CPU tensors. */
typedef struct lic360_coder lic360_coder;
lic360_coder *lic360_coder_enc_open(void);                                   /* Coder::start_encoder  coder.h:15-21 */
int lic360_coder_encode_slice(lic360_coder *c, const int *table, int ncode, const int *label, const float *mask, int num);
                                                                             /* encodes / encodes_mask coder.cpp:30-48,70-89 */
long lic360_coder_enc_finish(lic360_coder *c);                               /* Coder::end_encoder    coder.h:22-26 */
const uint8_t *lic360_coder_bytes(const lic360_coder *c);
lic360_coder *lic360_coder_dec_open(const uint8_t *bytes, long n);           /* Coder::start_decoder  coder.h:30-35 */
int lic360_coder_decode_slice(lic360_coder *c, const int *table, int ncode, const float *mask, float file_value, float *out, int num);
                                                                             /* decodes / decodes_mask coder.cpp:49-69,90-113 */
void lic360_coder_close(lic360_coder *c);

/* ---- D1/D2 device-resident latent codec ------------------------------------------------------- */
/* One object per (ngroup, h, w) entropy-domain shape and GPU; replaces EntEncoderFast / EntDecoder
 * (test/lic360_demo.py:95-141,191-238) for batches of up to max_batch images.  All pointers are device
 * memory; bitstreams live in HBM as bytes[b*cap .. b*cap+nbytes[b]) and are byte-identical to what
 * Coder::end_encoder would have written to `<code>` (extension/coder.h:22-26). */
typedef struct lic360_codec lic360_codec;
int lic360_codec_create(int ngroup, int h, int w, int max_batch, lic360_codec **codec);
void lic360_codec_destroy(lic360_codec *codec);
/* layer 0..11 of the 3 stacked nets [weight, sigma, mu]: weight [3,nout,cin_total,5,5], bias/act [3,nout]
 * (act NULL for layer 11) -- the tensors cast_entropy_parameter fills (test/lic360_demo.py:296-311) */
int lic360_codec_set_layer(void *stream, lic360_codec *codec, int layer, const float *weight, const float *bias, const float *act);
/* code/mask [b,ngroup,h,w] fp32 (symbols 0..7, mask 0/1) -> bitstreams; err[b] != 0 flags a coder fault or cap overflow */
int lic360_codec_encode(void *stream, lic360_codec *codec, const float *code, const float *mask, int b,
                        uint8_t *bytes, long cap, int *nbytes, int *err);
/* bitstreams + mask -> code_out [b,ngroup,h,w] (decoded symbol where mask, 0 elsewhere: `b[0:1] + 3.5*mask`, lic360_demo.py:236-237) */
int lic360_codec_decode(void *stream, lic360_codec *codec, const uint8_t *bytes, long cap, const int *nbytes,
                        const float *mask, int b, float *code_out, int *err);
/* Device-resident importance-map stream (ImpEntEncoderFast / ImpEntDecoder, test/lic360_demo.py:143-189, 241-290): one
 * group, 12 spatially causal layers with `hidden_channels` channels, nsym-way softmax tables (entropy_table_cuda.cu:24-96),
 * symbols = importance levels.  levels / levels_out: float [B,1,h,w]; weights per layer as [nout][C][5][5] (one net).
 * Bitstream buffers as in lic360_codec_encode / _decode. */
typedef struct lic360_impcodec lic360_impcodec;
int lic360_impcodec_create(int h, int w, int hidden_channels, int nsym, int max_batch, lic360_impcodec **out);
void lic360_impcodec_destroy(lic360_impcodec *c);
int lic360_impcodec_set_layer(void *stream, lic360_impcodec *c, int layer, const float *weight, const float *bias, const float *act);
int lic360_impcodec_encode(void *stream, lic360_impcodec *c, const float *levels, int B, uint8_t *bytes, long cap, int *nbytes, int *err);
int lic360_impcodec_decode(void *stream, lic360_impcodec *c, const uint8_t *bytes, long cap, const int *nbytes, int B, float *levels_out, int *err);
/* The map's decode AND the latent codec's mask, plane by plane: after plane p, mask_out = Dtow(stride)(Imp2mask(levels_out)) -- the
 * last three lines of ImpEntDecoder.forward (test/lic360_demo.py:283-287) -- is refreshed and the codec-owned event p is recorded,
 * so that lic360_codec_decode_gated on ANOTHER stream runs behind the map's decode instead of after it.  mask_out:
 * [B][mask_c / stride^2][stride h][stride w] (LIC360: mask_c = 192, stride = 2); *generation_out: the ticket of this masked decode.
 * ORDERING CONTRACT: (1) the events are re-recorded by every call, so the gated latent decode of a step must be enqueued AFTER this call
 * of the same step (enforced: the ticket); (2) the next step's masked decode overwrites mask_out -- order it after the latent decode
 * that still reads the buffer (an event on the latent stream, or alternate between two mask buffers as bench.py does). */
int lic360_impcodec_decode_masked(void *stream, lic360_impcodec *c, const uint8_t *bytes, long cap, const int *nbytes, int B,
                                  float *levels_out, int *err, float *mask_out, int mask_c, int stride, long *generation_out);
/* lic360_codec_decode behind that map decode: the convolutions of a latent plane read no mask; the plane's table kernel waits for the
 * map codec's event min(P - 1, plane / stride).  Same results as lic360_codec_decode on the finished mask.  `generation` must be the
 * ticket of map_codec's LATEST lic360_impcodec_decode_masked into this very `mask`, not yet used: anything else (a gate of an earlier
 * step, a gate used twice, the latent decode enqueued first) returns an error instead of decoding against a stale mask.  map_codec
 * must outlive the call.  No reference counterpart: EntDecoder.forward takes the finished mask (test/lic360_demo.py:218-238); this
 * only moves the 18 ms of a map's decode under the 97 ms of the latent's. */
int lic360_codec_decode_gated(void *stream, lic360_codec *codec, const uint8_t *bytes, long cap, const int *nbytes,
                              const float *mask, int b, float *code_out, int *err, lic360_impcodec *map_codec, long generation);

/* ---- f3 viewport projection (ProjectsOp, the sampling stage of VPSNR / VSSIM) -------------------------------------------------
 * Sampling coordinates tf [14][h_out*w_out][2] = (x, y) in ERP pixels of the 14 rectilinear viewports (yaw theta*pi, pitch phi*pi, field
 * of view fov*pi) on an ERP of height x width; host computes in fp32, device write
 *                                            extension/projects.hpp:8-20, projects_cuda.cu:7-67,101-153 */
int lic360_projects_tf(void *stream, float *tf_dev, int h_out, int w_out, const float *theta14, const float *phi14, float fov,
                       int height, int width);
/* ProjectsOp.forward: x [nc][h][w] -> out [14][nc][h_out][w_out] (viewport-major), bilinear or nearest
 *                                            extension/projects_cuda.cu:181-232 */
int lic360_projects_forward(void *stream, const float *x, const float *tf, float *out, int nc, int h, int w, int h_out, int w_out, int nearest);
/* ProjectsOp.backward: scatter-add of the viewport gradients (in_diff) and of their interpolation weights (count), both [nc][h][w]
 *                                            extension/projects_cuda.cu:234-329 */
int lic360_projects_backward(void *stream, const float *top_diff, const float *tf, float *in_diff, float *count, int nc, int h, int w,
                             int h_out, int w_out, int nearest);

/* CppOp: ERP -> Craster parabolic projection.  lic360_cpp_rows fills the per-row column window ws [h][2] = (first column, count) and the
 * row angle theta [h] (host computes, device write); lic360_cpp_forward resamples x [nc][h][w] row by row, zero outside the window; mask
 * (may be NULL) gets 1 inside / 0 outside     extension/CPP.hpp:6-11, CPP_cuda.cu:11-22,46-122 */
int lic360_cpp_rows(void *stream, int *ws_dev, float *theta_dev, int h, int w);
int lic360_cpp_forward(void *stream, const float *x, float *out, float *mask, const int *ws_dev, const float *theta_dev, int nc, int h, int w);

/* ViewportOp: one rectilinear viewport per sample looking at theta_phi [n,2] (radians, device), field of view fov_deg.
 * cal_rota_matrix -> rota [n,9]; forward -> out [n,c,ho,wo], rays0 / rays [n,ho,wo,3] (camera frame / rotated), tf [n,ho,wo,2] = (longitude,
 * latitude) of every viewport pixel; get_viewport_xy -> where the directions theta_phi_next fall in the current viewports, in pixels
 *                                            extension/viewport.hpp:7-13, viewport_cuda.cu:8-289 (its backward returns nothing) */
int lic360_viewport_rota(void *stream, const float *theta_phi, float *rota, int n);
int lic360_viewport_forward(void *stream, const float *x, const float *theta_phi, float *out, float *rays0, float *rota, float *rays, float *tf,
                            int n, int c, int h, int w, int ho, int wo, float fov_deg);
int lic360_viewport_xy(void *stream, const float *theta_phi_next, const float *rota, float *xy, int n, int ho, int wo, float fov_deg);

/* ---- f1 generalised divisive normalisation, one pass (the reference's GDN is four torch kernels: lic360_operator/GDN.py:66-100) ------
 * out[n,i,p] = x / sqrt(beta[i] + sum_j gamma[i,j] x[n,j,p]^2)   (inverse != 0: x * sqrt(...)); x / out [n][c][p] contiguous,
 * gamma [c][c] and beta [c] are the effective (reparametrised) parameters; c in {16,32,48,64,96,128,192}; j summed in ascending order */
int lic360_gdn(void *stream, const float *x, const float *gamma, const float *beta, float *out, int n, int c, long p, int inverse);

/* ---- f1 3x3 stride-1 convolution on sphere-apron maps with the apron read BY INDEX in the tile loader and bias + PReLU + residual in the
 * epilogue (csrc/conv3x3_kernels.hip): replaces, per layer, nn.Conv2d(c, c', 3, 1, 1 | 0) + the in-place SpherePad in front of it + the
 * nn.PReLU + SphereTrim (+ residual add) behind it in test/model_zoo.py:8-23,45-62,64-94,144-169 (sphere rule: extension/sphere_pad_cuda.cu:48-65).
 * x [n][cin][hp][wp]; sphere = 1: only the interior (pad cells in from every edge) is read, apron cells come from the interior by the sphere
 * rule; sphere = 2: longitude wrap only (columns of the apron come from the interior, rows are read as they are: the input is the 1-ring output
 * of another launch, whose wrapped columns need not be computed twice); sphere = 0: read as it is.
 * out [n][cout][hp - 2 crop][wp - 2 crop]: the cells of rows [ring, hp - ring) x columns [ring_w, wp - ring_w) of the input grid are
 * written (out = conv + bias; PReLU if slope; + residual [n][cout][hp][wp] if given), the others are NOT touched (SphereTrim(ring) = leave or
 * zero them: lic360_sphere_trim / lic360_sphere_apron_from).  crop = 1 is the unpadded nn.Conv2d(.., 3, 1) of ResidualBlockUp.conv1;
 * shuffle != 0 stores through Dtow(2, d2w) (extension/dtow_cuda.cu:38-75): out [n][cout / 4][2 (hp - 2 crop)][2 (wp - 2 crop)], channel 4 p + v of a
 * cell (y, x) at channel p, cell (2 y + v / 2, 2 x + v % 2) -- the conv -> PReLU -> Dtow chain of test/model_zoo.py:160-162 in one launch.
 * fp32 MFMA, this kernel's own summation order (1e-4 against a library convolution).  cin % 16 == 0, cout in {96} or a multiple of 192;
 * packed = lic360_sconv3x3_pack of the [cout][cin][3][3] weight; bias / slope 16-byte aligned. */
int lic360_sconv3x3_supported(int cin, int cout);
long lic360_sconv3x3_packed_floats(int cin, int cout);
int lic360_sconv3x3_pack(void *stream, const float *weight, float *packed, int cin, int cout);
int lic360_sconv3x3(void *stream, const float *x, const float *packed, const float *bias, const float *slope, const float *residual, float *out,
                    int n, int cin, int cout, int hp, int wp, int pad, int sphere, int ring, int ring_w, int crop, int shuffle);
/* the transforms' 1x1 layers on the same kernel body (K = input channels only): replaces nn.Conv2d(c, c', 1) + nn.PReLU (+ residual add) of
 * test/model_zoo.py:8-23 (ResidualBlock.conv1 / conv3) and, with crop = 1 and shuffle != 0, SphereCutEdge(1) + nn.Conv2d(c, 4c, 1) + Dtow(2) (+ add) of
 * the shortcut of ResidualBlockUp (test/model_zoo.py:165-168).  x [n][cin][hp][wp]; out (and residual) [n][cout][hp - 2 crop][wp - 2 crop], or shuffled
 * [n][cout / 4][2 (hp - 2 crop)][2 (wp - 2 crop)]; the cells of rows [ring, hp - ring) x columns [ring_w, wp - ring_w) of the input grid are written, the
 * others not touched; cin % 32 == 0, cout = 96 or a multiple of 192 */
int lic360_sconv1x1_supported(int cin, int cout);
long lic360_sconv1x1_packed_floats(int cin, int cout);
int lic360_sconv1x1_pack(void *stream, const float *weight, float *packed, int cin, int cout);
int lic360_sconv1x1(void *stream, const float *x, const float *packed, const float *bias, const float *slope, const float *residual, float *out,
                    int n, int cin, int cout, int hp, int wp, int ring, int ring_w, int crop, int shuffle);
/* apron of dst <- sphere-wrapped interior of src (src == dst: lic360_sphere_pad_inplace); [nc][hp][wp] planes      sphere_pad_cuda.cu:48-65 */
int lic360_sphere_apron_from(void *stream, const float *src, float *dst, int nc, int hp, int wp, int pad);

/* Test hooks of the DEVICE arithmetic coder (A19/A20 as they run inside the fused codec): raw int32 tables [n][ncode+1]
 * (every table totals 65536), labels and an optional mask, all in device memory, through the same kernels the codec
 * launches -- encode: k_ac_encode; decode: k_dec_init + k_dec_plane (ncode == 8) or k_imp_dec_plane (other alphabets),
 * `chunk` symbols per launch with the coder state carried between launches like between planes.  Reference behaviour:
 * extension/coder.cpp:30-113 over extension/ArithmeticCoder.cpp:34-116.  Synchronises the stream before returning.
 * The 8-symbol decoder carries the 7 inner table entries as 16-bit values (they are < 65536 whenever the last symbol has a
 * non-zero frequency, as every table of this codec does); a table that does not fit sets error bit 64. */
int lic360_devcoder_encode(void *stream, const int *tables, int ncode, const int *labels, const float *mask, long n,
                           uint8_t *bytes, long cap, int *nbytes, int *err);
int lic360_devcoder_decode(void *stream, const int *tables, int ncode, const float *mask, long n, int chunk,
                           const uint8_t *bytes, long cap, const int *nbytes, float *out, int *err);

/* Where the serial arithmetic-coder phases of the fused latent codec run.  The reference runs its coder on the CPU (extension/coder.cpp:70-113, called per
 * plane from test/lic360_demo.py:139-140,234); the fused codec keeps it on the GPU (one wave per image) where many images per call hide it, and hands it
 * to host threads -- records D2H / per-plane tables through pinned memory, polled flags -- when a call holds few images.  mode 0: device, 1: host
 * (<= 64 images per call), 2 (default): host for calls of at most 8 images.  LIC360_HOST_CODER=0|1 sets the mode at create.  Bitstreams are identical. */
int lic360_codec_set_coder(lic360_codec *codec, int mode);

/* ---- dead-cone skip of the fused latent codec (round 6; csrc/need.h) -------------------------------------------------------------
 * The reference evaluates every output of the entropy nets (extension/cconv_ec_cuda.cu:317-339, cconv_dc_cuda.cu:367-398) and coder.cpp:79 then
 * skips the masked symbols.  The fused codec does not compute what no coded symbol can observe: need_l(y, x) = highest group of layer l whose
 * output some coded symbol reads (-1: none), need_11 = the mask's highest coded group, need_l = min(G - 1, 5 x 5 dilation of need_{l+1} + dy + dx).
 * Bitstreams and decoded symbols are unchanged.  lic360_need_maps: masks [b, g, h, w] -> need [b][12][h][w] int8 (device), the kernel the codec runs. */
int lic360_need_maps(void *stream, const float *mask, int b, int g, int h, int w, signed char *need_out);
/* Host-only (no GPU work): how the decode-order task lists pack the live row windows lo[k]..hi[k] (k < c <= 8 samples of one chunk; hi < lo: none) of an
 * image of h <= 64 rows into waves -- round 5's tape rules with a window per sample; pieces[3 w + i] = piece i of wave w
 * (k | slo << 3 | shi << 9 | a0 << 15 | 1 << 21, 0 = none; room for 6 c words), *n_waves = waves used.  tests/test_dcl_pack.py. */
int lic360_dcl_pack_layout(int h, int c, const int *lo, const int *hi, unsigned *pieces, int *n_waves);
/* enable > 0: the following encodes / decodes count what they execute (0: stop; < 0: leave as it is).  out (host, 2 * 12 * 64 values, may be NULL; reading clears):
 * [0][layer][group block] live (tile, group block) pairs of the encode-order launches (64 positions x the block's groups, per sample),
 * [1][layer][group] cells the decode-order launches stored.  *skip_active_out (may be NULL): 0 the codec does not skip (generic kernels or
 * LIC360_NOSKIP), 1 encode order only, 2 both orders (decode order: batches of >= 16 images, 8 | batch, h <= 64). */
int lic360_codec_skip_stats(lic360_codec *codec, int enable, unsigned long long *out, int *skip_active_out);
/* test hooks: every interior cell of the codec's activation buffers <- value (finite); what the last encode / decode scheduled
 * (which 0: need maps, 1 / 2: encode-order list counts / entries, 3 / 4: decode-order list counts / records; see csrc/codec_fused.hip) */
int lic360_codec_debug_fill(void *stream, lic360_codec *codec, float value);
int lic360_codec_debug_lists(lic360_codec *codec, int which, void *host_out, long bytes, int *cap_out);

/* calibration for bench.py (no reference counterpart): the fp32 MFMA rate (v_mfma_f32_16x16x4_f32, 8 waves per CU) this device sustains right now, in
 * TFLOP/s, and its compute-unit count; ~50 ms.  bench.py prints it beside the headline so that figures of different boxes of a pool can be normalised. */
int lic360_calib_mfma_f32(void *stream, double *tflops, int *cus_out);

/* timing hooks for bench.py's instrumented pass: HIP events around every launch of each kernel class of the codec
 * (classes() names them, comma separated: first / hidden / last conv layers in both orders, table builds, coder kernels),
 * recorded on the caller's stream; read() fills ms[k] / launches[k] per class (n >= number of classes) and resets */
/* the kernels behind each kernel class of the timing hooks, "class=kernel+kernel;..." (names as rocprofv3 prints them) */
const char *lic360_codec_kernel_names(void);
int lic360_codec_profile_enable(lic360_codec *codec, int on);
const char *lic360_codec_profile_classes(void);
int lic360_codec_profile_read(lic360_codec *codec, int n, double *ms, long *launches);

#ifdef __cplusplus
}
#endif
#endif /* LIC360_HIP_H */
