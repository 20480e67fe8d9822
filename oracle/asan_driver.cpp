// asan_driver.cpp -- CPU-only sanitizer run (the reference has none) of the host-side product code and of the
// oracle: csrc/coder_host.cpp + csrc/ac_core.h (the drop-in Coder) and oracle/lic360_oracle.c, built with
// -fsanitize=address,undefined by `make -C oracle asan`.  Exercises: encode/decode of random and near-degenerate tables (masked
// and unmasked, empty slices, multi-slice streams), product <-> oracle interoperability in both directions, truncated and
// corrupted streams (must report, never read out of bounds), the oracle's conv / table / sphere ops on ragged shapes.
// TEST INFRASTRUCTURE: never linked into the product.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdarg>
#include "../include/lic360_hip.h"

void lic360_set_error(const char *, ...) {}          // the product's definition lives in a HIP translation unit

extern "C" {
typedef struct orc_ac orc_ac;
orc_ac *orc_ac_enc_open(void);
orc_ac *orc_ac_dec_open(const uint8_t *bytes, size_t n);
void orc_ac_close(orc_ac *a);
int orc_ac_error(orc_ac *a);
void orc_ac_encode_slice(orc_ac *a, const int *table, int ncode, const int *label, const float *mask, int num);
size_t orc_ac_enc_finish(orc_ac *a);
const uint8_t *orc_ac_bytes(orc_ac *a);
void orc_ac_decode_slice(orc_ac *a, const int *table, int ncode, const float *mask, float file_value, float *out, int num);
void orc_cconv_ec(const float *input, const float *weight, const float *bias, const float *act, float *out, int N, int C, int H, int W, int nout,
                  int ngroup, int ksz, int constrain, int nb);
void orc_gmm_table(float *weight, float *delta, const float *mean, float *out, int tn, int ng, int nstep, float bias, float total, float beta);
void orc_entropy_table(const float *data, float *out, int count, int nstep, float total);
void orc_sphere_pad(const float *in, float *out, int NC, int H, int W, int pad);
void orc_code_contex(int H, int W, int *idx, int *plane_idx);
}

static uint32_t rs = 12345;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }
static float frand() { return (float)(rnd() % 20001) / 10000.0f - 1.0f; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "asan_driver: check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static void make_tables(std::vector<int> &t, int n, int ncode, bool skew) {
    t.resize((size_t)n * (ncode + 1));
    for (int i = 0; i < n; ++i) {
        std::vector<int> f(ncode, 1);
        int left = 65536 - ncode;
        if (skew) f[rnd() % ncode] += left;
        else for (int k = 0; k < ncode; ++k) { int a = k + 1 == ncode ? left : (int)(rnd() % (left + 1)) / (ncode - k); f[k] += a; left -= a; }
        int acc = 0;
        for (int k = 0; k <= ncode; ++k) { t[(size_t)i * (ncode + 1) + k] = acc; if (k < ncode) acc += f[k]; }
    }
}

static int coder_case(int n, int ncode, bool masked, bool skew, int nslices) {
    std::vector<int> tab, lab(n);
    make_tables(tab, n, ncode, skew);
    std::vector<float> mask(n, 1.0f);
    for (int i = 0; i < n; ++i) { lab[i] = (int)(rnd() % ncode); if (masked && rnd() % 3 == 0) mask[i] = 0.0f; }
    const float *mk = masked ? mask.data() : nullptr;
    // product encoder, sliced
    lic360_coder *e = lic360_coder_enc_open();
    for (int s = 0; s < nslices; ++s) {
        const int a = (long)n * s / nslices, b = (long)n * (s + 1) / nslices;
        CHECK(lic360_coder_encode_slice(e, tab.data() + (size_t)a * (ncode + 1), ncode, lab.data() + a, mk ? mk + a : nullptr, b - a) == 0);
    }
    const long len = lic360_coder_enc_finish(e);
    std::vector<uint8_t> bytes(lic360_coder_bytes(e), lic360_coder_bytes(e) + len);
    lic360_coder_close(e);
    // oracle encoder: same bytes
    orc_ac *oe = orc_ac_enc_open();
    orc_ac_encode_slice(oe, tab.data(), ncode, lab.data(), mk, n);
    const size_t olen = orc_ac_enc_finish(oe);
    CHECK(olen == (size_t)len && memcmp(orc_ac_bytes(oe), bytes.data(), olen) == 0);
    orc_ac_close(oe);
    // product decoder and oracle decoder
    std::vector<float> out(n), out2(n);
    lic360_coder *d = lic360_coder_dec_open(bytes.data(), len);
    CHECK(d && lic360_coder_decode_slice(d, tab.data(), ncode, mk, 3.5f, out.data(), n) == 0);
    lic360_coder_close(d);
    orc_ac *od = orc_ac_dec_open(bytes.data(), bytes.size());
    orc_ac_decode_slice(od, tab.data(), ncode, mk, 3.5f, out2.data(), n);
    CHECK(orc_ac_error(od) == 0);
    orc_ac_close(od);
    for (int i = 0; i < n; ++i) {
        const float want = mask[i] < 0.5f && masked ? 3.5f : (float)lab[i];
        CHECK(out[i] == want && out2[i] == want);
    }
    // truncated / corrupted streams: any outcome but an out-of-bounds access
    if (len > 8) {
        lic360_coder *t = lic360_coder_dec_open(bytes.data(), len / 2);
        (void)lic360_coder_decode_slice(t, tab.data(), ncode, mk, 3.5f, out.data(), n);
        lic360_coder_close(t);
        for (auto &b : bytes) b ^= 0x5a;
        t = lic360_coder_dec_open(bytes.data(), len);
        (void)lic360_coder_decode_slice(t, tab.data(), ncode, mk, 3.5f, out.data(), n);
        lic360_coder_close(t);
    }
    return 0;
}

int main() {
    if (coder_case(0, 8, false, false, 1)) return 1;
    if (coder_case(1, 8, false, false, 1)) return 1;
    if (coder_case(5000, 8, true, false, 7)) return 1;
    if (coder_case(5000, 8, false, true, 3)) return 1;
    if (coder_case(3000, 49, false, false, 5)) return 1;
    if (coder_case(3000, 49, false, true, 1)) return 1;
    // oracle ops on ragged shapes
    {
        const int N = 3, G = 3, cin = 4, cout = 4, H = 5, W = 7, C = G * cin, nout = G * cout;
        std::vector<float> x((size_t)N * C * H * W), w((size_t)3 * nout * C * 25), b(3 * nout), a(3 * nout), o((size_t)N * nout * H * W);
        for (auto &v : x) v = frand();
        for (auto &v : w) v = 0.1f * frand();
        for (auto &v : b) v = 0.1f * frand();
        for (auto &v : a) v = 0.25f;
        orc_cconv_ec(x.data(), w.data(), b.data(), a.data(), o.data(), N, C, H, W, nout, G, 5, 6, 3);
        orc_cconv_ec(x.data(), w.data(), b.data(), nullptr, o.data(), N, C, H, W, nout, G, 5, 5, 3);
        std::vector<float> p((size_t)N * C * (H + 4) * (W + 4));
        orc_sphere_pad(x.data(), p.data(), N * C, H, W, 2);
        const int tn = 257;
        std::vector<float> gw(tn * 3), gd(tn * 3), gm(tn * 3), gt(tn * 9), lg((size_t)tn * 49), et((size_t)tn * 50);
        for (auto &v : gw) v = 3 * frand();
        for (auto &v : gd) v = frand();                      // negative and tiny sigmas: the floor and the fix-up walk
        for (auto &v : gm) v = 6 * frand();
        orc_gmm_table(gw.data(), gd.data(), gm.data(), gt.data(), tn, 3, 8, 3.5f, 65536.0f, 1e-6f);
        for (int i = 0; i < tn; ++i) { CHECK(gt[i * 9] == 0 && gt[i * 9 + 8] == 65536); for (int k = 0; k < 8; ++k) CHECK(gt[i * 9 + k + 1] > gt[i * 9 + k]); }
        for (auto &v : lg) v = 30 * frand();
        orc_entropy_table(lg.data(), et.data(), tn, 49, 65536.0f);
        for (int i = 0; i < tn; ++i) CHECK(et[i * 50] == 0 && et[i * 50 + 49] == 65536);
        std::vector<int> idx(2 * H * W), pidx(H + W);
        orc_code_contex(H, W, idx.data(), pidx.data());
        CHECK(pidx[H + W - 1] == H * W);
    }
    printf("asan_driver: ok\n");
    return 0;
}
