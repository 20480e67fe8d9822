// ac_core.h -- 32-bit binary arithmetic coder core, shared by the host Coder (coder_host.cpp) and
// the device-resident coder kernels (codec_fused.hip).  Written from the behaviour of the
// reference coder (extension/ArithmeticCoder.cpp:34-69,82-116,152-170; bit order
// extension/BitIoStream.cpp:19-34,52-71), NOT from its code: the per-bit renormalisation loops
// are replaced by closed forms (count-leading-zeros on low^high for the shift run, count of
// leading 01/10 pairs for the underflow run) and bits move through a 64-bit accumulator, so a
// symbol costs O(1) instead of O(bits).  Output is byte-identical (tests/test_host_coder.py and, on the
// device, tests/test_gpu_device_coder.py against tests/golden/ac_golden.npz).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AC_HD __host__ __device__ __forceinline__
#else
#define AC_HD inline
#endif
// In device code one wave runs one coder with wave-uniform state; only lane 0 touches the byte buffer.
#if defined(__HIP_DEVICE_COMPILE__)
#define AC_STORE_LANE ((threadIdx.x & 63) == 0)
#else
#define AC_STORE_LANE true
#endif

struct AcBitWriter {
    uint8_t *buf;      // destination
    long cap, len;     // capacity / bytes written (len may exceed cap: then bytes are dropped and overflow is reported)
    uint64_t acc;      // pending bits, right-aligned
    int nacc;          // number of pending bits (< 8 after every put)
};
struct AcBitReader {
    const uint8_t *buf;
    long len, pos;     // pos = next byte
    uint64_t acc;      // buffered bits, right-aligned
    int nacc;
};
struct AcState {
    uint32_t low, high, code;
    uint64_t underflow;
    int error;         // 0 ok; 1 zero-frequency symbol; 2 range invariant; 3 decoder consistency (last writer wins in these host/device
                       // functions).  The DEVICE ENCODER's serial core (k_ac_encode, codec_fused.hip) keeps a sticky BIT MASK instead: 1 | 2 = 3
                       // there means "empty symbol and range fault", not the decoder's code 3; both read as "stream invalid" (non-zero)
};

AC_HD void ac_bw_init(AcBitWriter &w, uint8_t *buf, long cap) { w.buf = buf; w.cap = cap; w.len = 0; w.acc = 0; w.nacc = 0; }
// append the low n bits of v (n <= 32), MSB first
AC_HD void ac_bw_put(AcBitWriter &w, uint32_t v, int n) {
    if (n == 0) return;
    w.acc = (w.acc << n) | (uint64_t)(n == 32 ? v : (v & ((1u << n) - 1u)));
    w.nacc += n;
    while (w.nacc >= 8) {
        w.nacc -= 8;
        uint8_t b = (uint8_t)(w.acc >> w.nacc);
        if (w.len < w.cap && AC_STORE_LANE) w.buf[w.len] = b;
        ++w.len;
    }
}
AC_HD void ac_bw_put_run(AcBitWriter &w, int bit, uint64_t n) {   // n copies of `bit`
    uint32_t pat = bit ? 0xffffffffu : 0u;
    while (n >= 32) { ac_bw_put(w, pat, 32); n -= 32; }
    ac_bw_put(w, pat, (int)n);
}
AC_HD void ac_bw_finish(AcBitWriter &w) {                          // zero-pad to a byte boundary
    if (w.nacc) ac_bw_put(w, 0, 8 - w.nacc);
}
AC_HD void ac_br_init(AcBitReader &r, const uint8_t *buf, long len) { r.buf = buf; r.len = len; r.pos = 0; r.acc = 0; r.nacc = 0; }
// read n bits (n <= 32), MSB first; past the end the stream reads as zeros
AC_HD uint32_t ac_br_get(AcBitReader &r, int n) {
    if (n == 0) return 0;
    while (r.nacc < n) {
        uint8_t b = r.pos < r.len ? r.buf[r.pos] : 0;
        ++r.pos;
        r.acc = (r.acc << 8) | b;
        r.nacc += 8;
    }
    r.nacc -= n;
    uint64_t v = r.acc >> r.nacc;
    return (uint32_t)(n == 32 ? v : (v & ((1ull << n) - 1ull)));
}

AC_HD int ac_clz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return v ? __clz((int)v) : 32;
#else
    return v ? __builtin_clz(v) : 32;
#endif
}

AC_HD void ac_init(AcState &s) { s.low = 0; s.high = 0xffffffffu; s.code = 0; s.underflow = 0; s.error = 0; }

// narrow [low,high] to the symbol's sub-range; returns the shift run n1 and underflow run n2
AC_HD void ac_narrow(AcState &s, uint32_t symLow, uint32_t symHigh, uint32_t total, int &n1, int &n2, uint32_t &low_before_shift) {
    uint64_t range = (uint64_t)s.high - (uint64_t)s.low + 1;
    if (symLow >= symHigh) s.error = 1;
    if (s.low >= s.high || range < ((1ull << 30) + 2)) s.error = 2;
    uint64_t nl, nh;
    if (total == 65536u) {                       // every table of this codec sums to 2^16
        nl = (uint64_t)s.low + (((uint64_t)symLow * range) >> 16);
        nh = (uint64_t)s.low + (((uint64_t)symHigh * range) >> 16) - 1;
    } else {
        nl = (uint64_t)s.low + (uint64_t)symLow * range / total;
        nh = (uint64_t)s.low + (uint64_t)symHigh * range / total - 1;
    }
    uint32_t low = (uint32_t)nl, high = (uint32_t)nh;
    low_before_shift = low;
    n1 = ac_clz32(low ^ high);                   // leading bits on which low and high agree
    if (n1 >= 32) { n1 = 31; s.error = 2; }
    if (n1) { low <<= n1; high = (high << n1) | ((1u << n1) - 1u); }
    uint32_t m = (low & ~high) << 1;             // bit 30 downwards: low=1, high=0
    n2 = ac_clz32(~m);
    if (n2 > 30) n2 = 30;
    if (n2) {
        low = (low << n2) & 0x7fffffffu;
        high = ((high << n2) & 0x7fffffffu) | 0x80000000u | ((1u << n2) - 1u);
    }
    s.low = low; s.high = high;
}

AC_HD void ac_encode_symbol(AcState &s, AcBitWriter &w, uint32_t symLow, uint32_t symHigh, uint32_t total) {
    int n1, n2;
    uint32_t lowb;
    ac_narrow(s, symLow, symHigh, total, n1, n2, lowb);
    if (n1) {
        int bit = (int)(lowb >> 31);
        ac_bw_put(w, (uint32_t)bit, 1);
        if (s.underflow) { ac_bw_put_run(w, bit ^ 1, s.underflow); s.underflow = 0; }
        if (n1 > 1) ac_bw_put(w, lowb >> (32 - n1), n1 - 1);   // bits 30 .. 32-n1 of low
    }
    s.underflow += (uint64_t)n2;
}
AC_HD void ac_encode_finish(AcState &s, AcBitWriter &w) {
    (void)s;
    ac_bw_put(w, 1u, 1);                         // ArithmeticEncoder::finish writes a single 1
    ac_bw_finish(w);
}

AC_HD void ac_decode_start(AcState &s, AcBitReader &r) { s.code = ac_br_get(r, 32); }

// floor(num / den) for num < 2^48, 2^30 < den <= 2^32, quotient < 2^17: one fp32 divide plus an exact
// integer correction (the fp32 estimate is off by at most 1), instead of a 64-bit integer division.
AC_HD uint32_t ac_div48(uint64_t num, uint64_t den) {
    uint32_t q = (uint32_t)((float)num / (float)den);
    int64_t r = (int64_t)num - (int64_t)((uint64_t)q * den);
    while (r < 0) { --q; r += (int64_t)den; }
    while (r >= (int64_t)den) { ++q; r -= (int64_t)den; }
    return q;
}
// value in [0,total) that the current code points at
AC_HD uint32_t ac_decode_target(const AcState &s, uint32_t total) {
    uint64_t range = (uint64_t)s.high - (uint64_t)s.low + 1;
    uint64_t offset = (uint64_t)s.code - (uint64_t)s.low;
    if (total == 65536u) return ac_div48(((offset + 1) << 16) - 1, range);
    return (uint32_t)(((offset + 1) * total - 1) / range);
}
// R: any bit source with  uint32_t get(int n)  returning the next n (<= 32) bits MSB-first, zeros past the end
struct AcHostBits {
    AcBitReader &r;
    AC_HD uint32_t get(int n) { return ac_br_get(r, n); }
};
template <class R>
AC_HD void ac_decode_consume_from(AcState &s, R &bits, uint32_t symLow, uint32_t symHigh, uint32_t total) {
    int n1, n2;
    uint32_t lowb;
    ac_narrow(s, symLow, symHigh, total, n1, n2, lowb);
    uint32_t code = s.code;
    if (n1) code = (n1 == 32 ? 0u : (code << n1)) | bits.get(n1);
    if (n2) code = (code & 0x80000000u) | ((code << n2) & 0x7fffffffu) | bits.get(n2);
    s.code = code;
    if (code < s.low || code > s.high) s.error = 3;
}
AC_HD void ac_decode_consume(AcState &s, AcBitReader &r, uint32_t symLow, uint32_t symHigh, uint32_t total) {
    int n1, n2;
    uint32_t lowb;
    ac_narrow(s, symLow, symHigh, total, n1, n2, lowb);
    uint32_t code = s.code;
    if (n1) code = (n1 == 32 ? 0u : (code << n1)) | ac_br_get(r, n1);
    if (n2) code = (code & 0x80000000u) | ((code << n2) & 0x7fffffffu) | ac_br_get(r, n2);
    s.code = code;
    if (code < s.low || code > s.high) s.error = 3;
}
