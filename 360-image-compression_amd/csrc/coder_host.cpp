// coder_host.cpp -- host-side Coder of the drop-in API (A19/A20).  The reference's Coder is a CPU
// object fed with CPU int32 tensors (extension/coder.h:10-63, extension/coder.cpp:30-113); this is
// its counterpart on top of ac_core.h.  The fused device-resident codec uses the same core in
// ac_kernels.hip.
#include "common.h"
#include "ac_core.h"
#include <vector>
#include <cstring>

struct lic360_coder {
    AcState st;
    AcBitWriter bw;
    AcBitReader br;
    std::vector<uint8_t> buf;
    bool encoding;
};

static void grow(lic360_coder *c, long need) {
    if ((long)c->buf.size() < c->bw.len + need) {
        c->buf.resize(std::max<size_t>(c->buf.size() * 2, (size_t)(c->bw.len + need)));
    }
    c->bw.buf = c->buf.data();
    c->bw.cap = (long)c->buf.size();
}

LIC360_API lic360_coder *lic360_coder_enc_open(void) {
    lic360_coder *c = new lic360_coder();
    c->encoding = true;
    c->buf.resize(1 << 16);
    ac_init(c->st);
    ac_bw_init(c->bw, c->buf.data(), (long)c->buf.size());
    return c;
}

LIC360_API int lic360_coder_encode_slice(lic360_coder *c, const int *table, int ncode, const int *label, const float *mask, int num) {
    ARG_CHECK(c && c->encoding && ncode > 0 && num >= 0);
    if (num == 0) return 0;
    ARG_CHECK(table && label);
    for (int i = 0; i < num; ++i) {
        if (mask && mask[i] < 0.5f) continue;
        const int *t = table + (long)i * (ncode + 1);
        int sym = label[i];
        if (sym < 0 || sym >= ncode) { lic360_set_error("symbol %d out of range [0,%d) at %d", sym, ncode, i); return 3; }
        grow(c, 64 + (long)(c->st.underflow / 8));
        ac_encode_symbol(c->st, c->bw, (uint32_t)t[sym], (uint32_t)t[sym + 1], (uint32_t)t[ncode]);
        if (c->st.error) { lic360_set_error("arithmetic encoder error %d at symbol %d (zero-frequency symbol or corrupt table)", c->st.error, i); return 3; }
    }
    return 0;
}

LIC360_API long lic360_coder_enc_finish(lic360_coder *c) {
    if (!c || !c->encoding) return -1;
    grow(c, 16);
    ac_encode_finish(c->st, c->bw);
    return c->bw.len;
}
LIC360_API const uint8_t *lic360_coder_bytes(const lic360_coder *c) { return c ? c->buf.data() : nullptr; }

LIC360_API lic360_coder *lic360_coder_dec_open(const uint8_t *bytes, long n) {
    if (n < 0 || (n > 0 && !bytes)) { lic360_set_error("bad bitstream buffer"); return nullptr; }
    lic360_coder *c = new lic360_coder();
    c->encoding = false;
    c->buf.assign(bytes, bytes + n);
    ac_init(c->st);
    ac_br_init(c->br, c->buf.data(), n);
    ac_decode_start(c->st, c->br);
    return c;
}

LIC360_API int lic360_coder_decode_slice(lic360_coder *c, const int *table, int ncode, const float *mask, float file_value, float *out, int num) {
    ARG_CHECK(c && !c->encoding && ncode > 0 && num >= 0);
    if (num == 0) return 0;
    ARG_CHECK(table && out);
    for (int i = 0; i < num; ++i) {
        if (mask && mask[i] < 0.5f) { out[i] = file_value; continue; }
        const int *t = table + (long)i * (ncode + 1);
        uint32_t total = (uint32_t)t[ncode];
        uint32_t value = ac_decode_target(c->st, total);
        // highest symbol with table[s] <= value  (binary search of ArithmeticDecoder::read)
        uint32_t start = 0, end = (uint32_t)ncode;
        while (end - start > 1) {
            uint32_t mid = (start + end) >> 1;
            if ((uint32_t)t[mid] > value) end = mid; else start = mid;
        }
        ac_decode_consume(c->st, c->br, (uint32_t)t[start], (uint32_t)t[start + 1], total);
        if (c->st.error) { lic360_set_error("arithmetic decoder error %d at symbol %d (corrupt stream or table)", c->st.error, i); return 3; }
        out[i] = (float)start;
    }
    return 0;
}

LIC360_API void lic360_coder_close(lic360_coder *c) { delete c; }
