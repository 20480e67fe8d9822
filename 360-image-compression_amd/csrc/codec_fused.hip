// codec_fused.hip -- device-resident latent codec (rows D1/D2 of SURVEY.md §8a as ONE pipeline per
// batch of images): the whole of EntEncoderFast.forward / EntDecoder.forward
// (test/lic360_demo.py:119-141, 215-238) without the reference's per-plane GPU<->CPU round trips.
//
// encode:  prep (code-3.5)*mask  ->  12 masked-conv layers (MFMA, residual adds fused)  ->
//          per-symbol GMM CDF build in coding order, keeping only (cdf[sym], cdf[sym+1])  ->
//          one arithmetic-coder wave per image writing the bitstream into HBM.
// decode:  per anti-diagonal plane p: 12 plane-restricted conv layers on a diagonal-major
//          activation layout  ->  one wave per image: 64 CDF tables in parallel, then the serial
//          range decoder on wave-uniform state, symbols scattered straight into the next plane's input.
// Bitstreams are byte-identical to the per-plane drivers and to the CPU oracle.
#include "common.h"
#include "conv_plan.h"
#include "ac_core.h"
#include "lic360_exact_math.h"
#include "gmm_tables.h"
#include "need.h"
#include <vector>
#include <cstring>
#include <algorithm>
#include <thread>
#include <atomic>
#include <immintrin.h>

struct AcDevState {            // per-image decoder state carried across planes
    uint32_t low, high, code;
    int nacc, error, pad;
    long pos;
    unsigned long long acc;
};

enum { PROF_EC_FIRST, PROF_EC_HIDDEN, PROF_EC_LAST, PROF_ENC_TABLES, PROF_AC_ENCODE,
       PROF_DC_FIRST, PROF_DC_HIDDEN, PROF_DC_LAST, PROF_DEC_TABLES, PROF_DEC_PLANE, PROF_NCLS };
static const char *const PROF_NAMES = "ec_first,ec_hidden,ec_last,enc_tables,ac_encode,dc_first,dc_hidden,dc_last,dec_tables,dec_plane";

struct lic360_codec {
    int G, H, W, maxB, S, P, HW;
    int sk_rows, sk_pitch, sk_row0, sk_col0;
    int e_hp, e_wp, e_off;                     // encode activation planes: [e_hp][e_wp], cell (r, c) at [(r+e_off)*e_wp + c+e_off]
    lic360_conv_plan *plan[3];                 // first, hidden, last
    float *packed[12], *bias[12], *act[12];
    float *packed4[12];                        // leaf-resident (4x4x1 MFMA) weight layout, when the shape allows it
    float *packed16[12];                       // 16x16x4 MFMA weight layout of the encode-order kernel (csrc/cconv16_kernels.hip)
    bool use4;                                 // the nets' shapes fit the specialised kernels (4x4x1 decode order, 16x16x4 encode order with the last layer
                                               // fused with the CDF tables); otherwise -- or under LIC360_FUSED_CONV=16 -- the generic kernels of cconv_kernels.hip
    int dc_mode = 0;                           // schedule switches of the decode kernel (LIC360_NOPACK, LIC360_DC_GSTEP), read from the environment once, at create
    int *e_ctr = nullptr;                      // 8 task counters of the encode kernel
    std::vector<int> h_idx, h_pidx, h_plane_start;
    int *d_idx, *d_pidx, *d_plane_start;
    float *e_x0, *e_buf[3];
    uint2 *e_rec;
    float *d_x0, *d_act[11], *d_y;
    AcDevState *d_state;
    uint4 *d_tab = nullptr;                    // per-plane CDF tables [maxB][tab_pitch][2] (k_dec_tables -> k_dec_plane)
    int tab_pitch = 0;
    bool layer_set[12];
    // Dead-cone skip (round 6, need.h): outputs no coded symbol can observe are not computed -- per-layer need maps from the mask, compacted task lists
    // for the encode-order kernels (layers 1..11), per-plane task records for the decode-order kernel (layers 1..11; batches of >= 16 images with
    // 8 | batch on images of at most 64 rows -- below that a list could only drop whole three-group tasks, which almost never happens, and the decode
    // would have to wait for the whole importance map instead of running behind it).  LIC360_NOSKIP=1 (read at create) turns it off.
    bool skip = false;
    signed char *need = nullptr, *need_d = nullptr, *tmax = nullptr;
    lic360_ec_lists ecl;
    lic360_dc_lists dcl;
    unsigned long long *stats = nullptr;       // [2][12][NEED_STAT_G]: live tiles per (layer, group block) of the encodes, stored cells per (layer, group) of the decodes
    bool stats_on = false;
    // Host leg of the arithmetic coder (round 6; the reference's own division of labour: extension/coder.cpp:70-113 runs on the CPU).  One wave per image
    // is the slowest possible coder -- 27.6 ms per image to encode, ~170 us per plane to decode -- and only hides behind other images' convolutions; a
    // host thread does 21 / 29 ns per symbol.  With few images per call (<= coder_auto_max = 8) the serial phases therefore run on host threads:
    // encode -- records D2H, one thread per image, bitstreams H2D; decode -- per plane the packed tables go to pinned memory, one persistent thread per
    // image decodes its symbols, a GPU kernel waits for them on polled flags (no API call, no stream synchronisation per plane).
    int coder_mode = 2;                        // 0 device, 1 host, 2 auto (host when B <= coder_auto_max); LIC360_HOST_CODER=0|1 overrides at create
    int coder_auto_max = 8;                    // (16 images per stream on three streams lose: config 4 as written 52 -> 36 Mpixel/s, config 5 47 -> 18 -- every plane then waits for its slowest host thread while the GPU coder of one stream hides behind the other streams' convolutions)
    struct HostLeg *hl = nullptr;
    // optional per-kernel-class timing (bench.py's instrumented pass; off in the timed region): HIP event pairs around
    // every launch of a class, recorded on the launch stream
    bool prof = false;
    std::vector<hipEvent_t> ev[PROF_NCLS];     // start/stop pairs
    size_t n_ev[PROF_NCLS] = {};
};

static int prof_mark(lic360_codec *c, int cls, hipStream_t s) {
    if (!c->prof) return 0;
    std::vector<hipEvent_t> &pool = c->ev[cls];
    if (c->n_ev[cls] >= pool.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        pool.push_back(e);
    }
    HIP_TRY(hipEventRecord(pool[c->n_ev[cls]++], s));
    return 0;
}
// brackets one launch (or launch group) of class CLS
#define PROF(c, CLS, s, ...) do { prof_mark(c, CLS, s); __VA_ARGS__; prof_mark(c, CLS, s); } while (0)

static int plan_of(int layer) { return layer == 0 ? 0 : (layer == 11 ? 2 : 1); }

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------ encode
// offset of cell (r, c) inside one encode activation plane (zero-haloed NCHW, or plain NCHW with off = 0 for the generic kernels)
__device__ __forceinline__ long e_cell(int r, int c, int wp, int off) { return (long)(r + off) * wp + c + off; }

__global__ void k_enc_prep(const float *__restrict__ code, const float *__restrict__ mask, float *__restrict__ x0, long total,
                           int H, int W, int hp, int wp, int off) {
    GRID_STRIDE(i, total) {
        int c = (int)(i % W), r = (int)((i / W) % H);
        long pl = i / ((long)H * W);
        x0[pl * hp * wp + e_cell(r, c, wp, off)] = (code[i] - 3.5f) * mask[i];       // lic360_demo.py:130
    }
}

// one thread per latent element (NCHW-linear, coalesced reads); the (cdf[sym], cdf[sym+1]) pair is
// written at the symbol's position in coding order: plane p = g+th+tw, diagonals ascending inside a
// plane, rows ascending inside a diagonal (extension/code_contex_cuda.cu:19-31, tile_extract_cuda.cu:36-41).
__global__ void k_enc_tables(const float *__restrict__ y, const float *__restrict__ code, const float *__restrict__ mask,
                             const int *__restrict__ pidx, const int *__restrict__ plane_start, uint2 *__restrict__ rec,
                             int B, int G, int H, int W, int hp, int wp, int off) {
    const long HW = (long)H * W, per = (long)G * HW, total = per * B;
    GRID_STRIDE(i, total) {
        int tw = (int)(i % W), th = (int)((i / W) % H), g = (int)((i / HW) % G), b = (int)(i / per);
        int s = th + tw, p = s + g;
        int la = p >= G ? p - G + 1 : 0;
        long k = plane_start[p] + (pidx[s] - pidx[la]) + (th - (s >= W ? s - W + 1 : 0));
        uint2 r = make_uint2(0u, 0u);
        if (!(mask[i] < 0.5f)) {                                     // coder.cpp:79
            float v[9];
            const long cell = e_cell(th, tw, wp, off);
#pragma unroll
            for (int net = 0; net < 3; ++net)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    v[net * 3 + c] = y[((long)(net * B + b) * (3 * G) + g * 3 + c) * hp * wp + cell];
            int T[9];
            gmm_cdf9(v, v + 3, v + 6, T);
            int sym = (int)code[i];
            sym = sym < 0 ? 0 : (sym > 7 ? 7 : sym);
            r = make_uint2((unsigned)T[sym], (unsigned)T[sym + 1]);
        }
        rec[(long)b * per + k] = r;
    }
}

// Wave-resident bit sink of the encoder: pending bits collect in a 64-bit scalar; every full 32-bit word goes (big-endian) into
// lane `wpos` of ONE VGPR (one v_cndmask), and the 256-byte window is stored with one coalesced dword store per 64 words --
// no byte stores and no divergent code on the serial chain.  Words past the stream's slot are dropped, `len` keeps counting
// (overflow is reported as error bit 16).  Streams start 4-byte aligned (cap % 4 == 0).
struct DevBitSink {
    uint8_t *buf;
    long cap, wbase;            // wbase: byte offset of the window
    unsigned long long acc;     // pending bits, right-aligned
    int nacc, wpos, lane;       // nacc < 32 between puts; wpos: words in the window
    uint32_t win;
    __device__ __forceinline__ void init(uint8_t *b, long c, int l) { buf = b; cap = c; wbase = 0; acc = 0; nacc = 0; wpos = 0; lane = l; win = 0; }
    __device__ __forceinline__ void flush() {
        if (lane < wpos && wbase + 4 * lane + 4 <= cap) *(uint32_t *)(buf + wbase + 4 * lane) = win;
        wbase += 4 * wpos;
        wpos = 0;
    }
    __device__ __forceinline__ void put(uint32_t v, int n) {              // the low n bits of v (n <= 32), MSB first
        acc = (acc << n) | (unsigned long long)(n == 32 ? v : (v & ((1u << n) - 1u)));
        nacc += n;
        if (nacc >= 32) {
            nacc -= 32;
            const uint32_t word = __builtin_bswap32((uint32_t)(acc >> nacc));
            win = lane == wpos ? word : win;
            if (++wpos == 64) flush();
        }
    }
    __device__ __forceinline__ void put_run(int bit, unsigned long long n) {
        const uint32_t pat = bit ? 0xffffffffu : 0u;
        while (n >= 32) { put(pat, 32); n -= 32; }
        if (n) put(pat, (int)n);
    }
    // zero-pad to a byte boundary, store what is left; returns the stream length in bytes
    __device__ __forceinline__ long finish() {
        const int tail = (nacc + 7) >> 3;                                 // bytes of the last, partial word
        if (tail) {
            const uint32_t word = __builtin_bswap32((uint32_t)(acc << (32 - nacc)));
            win = lane == wpos ? word : win;
        }
        const long len = wbase + 4 * wpos + tail;
        if (lane < wpos && wbase + 4 * lane + 4 <= cap) *(uint32_t *)(buf + wbase + 4 * lane) = win;
        if (lane == wpos && tail) {
            for (int i = 0; i < tail; ++i)
                if (wbase + 4 * lane + i < cap) buf[wbase + 4 * lane + i] = (uint8_t)(win >> (8 * i));
        }
        return len;
    }
};
// Two waves per image, a two-stage pipeline over groups of 64 records: wave 0 runs the interval recurrence (ac_narrow: the only part
// that is inherently serial) and leaves, per coded symbol, the shift run n1, the underflow run n2 and the interval's low word in lane j
// of two registers; wave 1 turns the previous group's triples into bits (ArithmeticEncoder::shift / underflow bookkeeping is a pure
// function of that sequence).  Each wave's chain is about half of the single-wave coder's; one workgroup barrier per group.
__global__ __launch_bounds__(128) void k_ac_encode(const uint2 *__restrict__ rec, long n, uint8_t *__restrict__ bytes, long cap,
                                                   int *__restrict__ nbytes, int *__restrict__ err) {
    __shared__ uint2 ring[2][64];                                         // x: low word before the shift; y: n1 | n2 << 8 | coded << 16
    __shared__ int s_error;
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint2 *r = rec + (long)b * n;
    const long ngroups = (n + 63) / 64;
    if (wave == 0) {
        AcState st;
        ac_init(st);
        auto load_rec = [&](long base) __attribute__((always_inline)) {
            uint2 v = make_uint2(0u, 0u);
            if (base + lane < n) v = r[base + lane];
            return v;
        };
        uint2 v = load_rec(0), v1 = load_rec(64);                          // records are fetched two groups ahead of the chain
        for (long g = 0; g <= ngroups; ++g) {
            if (g < ngroups) {
                const uint2 v2 = load_rec(64 * g + 128);
                unsigned long long todo = __ballot(v.y != 0u);            // hi == 0: not coded (mask < 0.5, coder.cpp:79)
                uint2 o = make_uint2(0u, 0u);
                uint32_t low = st.low, high = st.high, serr = st.error;
                while (todo) {
                    const int j = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v.x, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)v.y, j);
                    // ac_narrow (ac_core.h) for total = 2^16 as 41 scalar instructions (hipcc: 60+): interval starts as one s_mul_hi_u32 of
                    // (T << 16) and range (range == 2^32: T << 16 itself; T = 65536 = the end of the table: the whole range), high carried
                    // complemented through the two shift runs, the faults as sticky bits (1: empty symbol, 2: range too small).
                    uint32_t lowb, meta, r1, rng, full, sa, sb, t0, tt;
                    asm("s_sub_u32 %5, %1, %0\n\t"                 // r1 = high - low
                        "s_cmp_lt_u32 %5, 0x40000001\n\t"          // range < 2^30 + 2 (ArithmeticCoder.cpp:44-45)
                        "s_cselect_b32 %10, 2, 0\n\t"
                        "s_or_b32 %2, %2, %10\n\t"
                        "s_add_u32 %6, %5, 1\n\t"                  // range (mod 2^32); SCC = (range == 2^32)
                        "s_cselect_b32 %7, -1, 0\n\t"
                        "s_lshl_b32 %11, %12, 16\n\t"
                        "s_mul_hi_u32 %8, %11, %6\n\t"
                        "s_and_b32 %10, %11, %7\n\t"
                        "s_or_b32 %8, %8, %10\n\t"                 // floor(symLow * range >> 16)
                        "s_lshl_b32 %11, %13, 16\n\t"
                        "s_mul_hi_u32 %9, %11, %6\n\t"
                        "s_and_b32 %10, %11, %7\n\t"
                        "s_or_b32 %9, %9, %10\n\t"
                        "s_add_u32 %9, %9, -1\n\t"                 // floor(symHigh * range >> 16) - 1
                        "s_cmp_eq_u32 %11, 0\n\t"                  // symHigh = 65536
                        "s_cselect_b32 %9, %5, %9\n\t"
                        "s_cmp_ge_u32 %12, %13\n\t"                // symLow >= symHigh (ArithmeticCoder.cpp:41-42)
                        "s_cselect_b32 %10, 1, 0\n\t"
                        "s_or_b32 %2, %2, %10\n\t"
                        "s_add_u32 %9, %0, %9\n\t"                 // high'
                        "s_add_u32 %0, %0, %8\n\t"                 // low'
                        "s_mov_b32 %3, %0\n\t"                     // the low word before the shift: its top bits are the output
                        "s_not_b32 %10, %9\n\t"
                        "s_xor_b32 %8, %0, %9\n\t"
                        "s_or_b32 %8, %8, 1\n\t"
                        "s_flbit_i32_b32 %4, %8\n\t"               // n1
                        "s_lshl_b32 %0, %0, %4\n\t"
                        "s_lshl_b32 %10, %10, %4\n\t"
                        "s_and_b32 %8, %0, %10\n\t"
                        "s_lshl_b32 %8, %8, 1\n\t"
                        "s_not_b32 %8, %8\n\t"
                        "s_flbit_i32_b32 %8, %8\n\t"               // n2
                        "s_min_u32 %8, %8, 30\n\t"
                        "s_lshl_b32 %0, %0, %8\n\t"
                        "s_bitset0_b32 %0, 31\n\t"
                        "s_lshl_b32 %10, %10, %8\n\t"
                        "s_orn2_b32 %1, 0x80000000, %10\n\t"
                        "s_lshl_b32 %8, %8, 8\n\t"
                        "s_or_b32 %4, %4, %8\n\t"
                        "s_bitset1_b32 %4, 16"                     // meta = n1 | n2 << 8 | coded << 16
                        : "+s"(low), "+s"(high), "+s"(serr), "=&s"(lowb), "=&s"(meta), "=&s"(r1), "=&s"(rng), "=&s"(full), "=&s"(sa), "=&s"(sb),
                          "=&s"(t0), "=&s"(tt)
                        : "s"(lo), "s"(hi)
                        : "scc");
                    // lane j of o <- (lowb, meta)  (two SGPR sources per v_writelane: the lane select goes through m0; the s_nop covers m0's hazard)
                    asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                                 : "+v"(o.x), "+v"(o.y) : "s"(lowb), "s"(meta), "s"(j));   // (m0 is not in the clobber list: hipcc treats it as reserved and uses it for nothing in this kernel -- gfx9 LDS instructions do not read it)
                }
                st.low = low; st.high = high; st.error = serr;
                ring[g & 1][lane] = o;
                v = v1; v1 = v2;
            }
            __syncthreads();
        }
        if (lane == 0) s_error = st.error;
        __syncthreads();
    } else {
        DevBitSink bw;
        bw.init(bytes + (long)b * cap, cap, lane);
        unsigned long long underflow = 0;
        for (long g = 0; g <= ngroups; ++g) {
            if (g > 0) {
                const uint2 o = ring[(g - 1) & 1][lane];
                unsigned long long todo = __ballot((o.y >> 16) != 0u);
                while (todo) {
                    const int j = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint32_t lowb = (uint32_t)__builtin_amdgcn_readlane((int)o.x, j), meta = (uint32_t)__builtin_amdgcn_readlane((int)o.y, j);
                    const int n1 = (int)(meta & 0xffu), n2 = (int)((meta >> 8) & 0xffu);
                    if (n1) {                                               // ArithmeticCoder.cpp:53-69 on the closed forms
                        if (underflow == 0) bw.put(lowb >> (32 - n1), n1);
                        else {
                            const int bit = (int)(lowb >> 31);
                            bw.put((uint32_t)bit, 1);
                            bw.put_run(bit ^ 1, underflow);
                            underflow = 0;
                            if (n1 > 1) bw.put(lowb >> (32 - n1), n1 - 1);
                        }
                    }
                    underflow += (unsigned long long)n2;
                }
            }
            __syncthreads();
        }
        bw.put(1u, 1);                                                    // ArithmeticEncoder::finish writes a single 1
        const long len = bw.finish();
        __syncthreads();
        if (lane == 0) {
            nbytes[b] = (int)len;
            err[b] = s_error | (len > cap ? 16 : 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ decode
// the device-side stream length is never trusted: it is clamped to the stream's slot (error bit 32 flags a clamp)
__device__ __forceinline__ long dev_stream_len(int nb, long cap) { return nb < 0 ? 0 : ((long)nb > cap ? cap : (long)nb); }
__global__ void k_dec_init(const uint8_t *__restrict__ bytes, long cap, const int *__restrict__ nbytes, AcDevState *__restrict__ state, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    AcBitReader rd;
    const long len = dev_stream_len(nbytes[b], cap);
    ac_br_init(rd, bytes + (long)b * cap, len);
    AcState st;
    ac_init(st);
    ac_decode_start(st, rd);                                          // consumes exactly 4 bytes: pos = 4, no pending bits
    AcDevState d;
    d.low = st.low; d.high = st.high; d.code = st.code; d.error = (len != (long)nbytes[b]) ? 32 : 0; d.pad = 0;
    d.pos = rd.pos; d.acc = 0; d.nacc = 0;
    state[b] = d;
}

// Wave-resident bit source for the decoder: lane i of `win` holds stream bytes [wbase+4i, wbase+4i+4) as a big-endian
// word (zeros past the end), so a 32-bit refill is one v_readlane instead of dependent byte loads from global memory;
// the 256-byte window is re-fetched with one coalesced load when the read position leaves it.
struct DevBits {
    const uint8_t *buf;
    int len, pos;               // pos: next unread byte (multiple of 4); streams are < 2 GB
    unsigned long long acc;
    int nacc;
    int wbase;
    uint32_t win;
    int lane;
    __device__ __forceinline__ void fetch_window() {
        wbase = pos;
        const int o = wbase + 4 * lane;
        uint32_t w = 0;
        if (o < len) {
            w = *(const uint32_t *)(buf + o);                           // streams start 4-byte aligned (cap % 4 == 0)
            w = __builtin_bswap32(w);
            if (o + 4 > len) w &= 0xffffffffu << (8 * (o + 4 - len));
        }
        win = w;
    }
    __device__ __forceinline__ uint32_t get(int n) {
        if (nacc < n) {
            if (pos >= wbase + 256) fetch_window();
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win, (pos - wbase) >> 2);
            acc = (acc << 32) | w;
            nacc += 32;
            pos += 4;
        }
        nacc -= n;
        const unsigned long long v = acc >> nacc;
        return (uint32_t)(n == 32 ? v : (v & ((1ull << n) - 1ull)));
    }
};

// 8-symbol tables travel as EIGHT words per symbol, word k = T[k] << 16 (T[0] = 0; T[8] = 65536 is implicit; an uncoded symbol is
// all zeros, a coded one has T[7] >= 7): the decoder loads 8 symbols per VGPR -- lane 8q + k holds word k of the q-th -- so that ONE
// v_mul_hi_u32 yields all eight interval starts of a symbol and one v_cmp finds it (ac_decode_symbol8_v).  The inner entries are
// <= 65535 because T[8] = 65536 and the entries increase.
__device__ __forceinline__ void dec_pack8(const int *T, uint4 &a, uint4 &b) {
    a = make_uint4(0u, (unsigned)T[1] << 16, (unsigned)T[2] << 16, (unsigned)T[3] << 16);
    b = make_uint4((unsigned)T[4] << 16, (unsigned)T[5] << 16, (unsigned)T[6] << 16, (unsigned)T[7] << 16);
}
// One symbol of the serial decode chain, wave-uniform, without a division.  ArithmeticDecoder::read (ArithmeticCoder.cpp:82-116)
// finds value = ((offset+1)*total - 1) / range and the symbol with T[sym] <= value < T[sym+1]; since
//   value >= T[k]  <=>  (offset+1)*total > T[k]*range  <=>  offset >= floor(T[k]*range / total)
// the symbol is the number of k in 1..7 whose interval start  floor(T[k]*range >> 16)  is <= offset, and the starts that bracket
// it ARE the new low / high of ac_narrow.
// The chain is a lone wave issuing one instruction every 4-5 cycles, so its length in instructions IS the decode latency of an
// image.  ~72 instructions per coded symbol (round 2: 125 from C++; hipcc materialises every comparison as a lane mask, assembles
// small integers on the VALU ...), the serial core as inline assembly:
//   * all eight starts at once: v_mul_hi_u32 of the symbol's eight lanes (T[k] << 16) with range -- floor(T*range >> 16) exactly,
//     range < 2^32; range == 2^32 (whenever low = 0, high = 2^32 - 1 recurs) selects T << 16 itself -- one v_cmp against offset,
//     a popcount of the symbol's 8 flag bits, two v_readlane for the bracketing starts (T[8]: him1 = range - 1);
//   * the underflow run of the code register as one shift and one XOR: while low = 01.., high = 10.. and low <= code <= high, the
//     second bit of code is the complement of its first, so  (code & TOP) | ((code << 1) & ~TOP)  ==  (code << 1) ^ TOP, and n2
//     such steps are (code << n2) ^ TOP; with the n1 plain shifts before them: code' = ((code << n) | next n bits) ^ (n2 ? TOP : 0);
//   * high is carried complemented through both shift runs ( ~((~h) << n) == (h << n) | ones(n) );
//   * n = n1 + n2 <= 18: a selected symbol has T[sym+1] - T[sym] >= 1 of 65536 (a zero-width entry is never selected: offset >= b
//     and offset < b cannot both hold) and range > 2^30 before the step, so the new range is >= 2^14: low' and high' agree on at most
//     18 leading bits, and every underflow bit halves what is left of that budget.  One branch-free 32-bit refill therefore covers
//     every read, whatever the stream holds (a conditional update of the reader state costs a dozen register copies per symbol).
// The reference's range / consistency assertions (errors 2 and 3 of ac_core.h) cannot fire inside this function: the symbol is
// chosen so that low+lo <= code <= low+him1, and the renormalisation maps that interval and the code by the same shifts.  They
// are checked once per launch on the state that enters and leaves (ac_state_check) instead of per symbol.
// tv: the table words of 8 symbols (dec_pack8), L: first lane of this symbol's eight.
__device__ __forceinline__ int ac_decode_symbol8_v(AcState &s, DevBits &rd, uint32_t tv, int L) {
    uint32_t low = s.low, high = s.high;
    int sym, n1, n2;
    uint32_t r1, off, rng, lo, hi, t0, t1, t2, stv;
    unsigned long long fm, fl;
    // (three statements: hipcc treats EVERY result of an asm statement as divergent as soon as one of them is a vector register)
    asm("s_sub_u32 %0, %5, %4\n\t"                 // r1 = high - low
        "s_sub_u32 %1, %6, %4\n\t"                 // offset = code - low
        "s_add_u32 %2, %0, 1\n\t"                  // range (mod 2^32); SCC = (range == 2^32)
        "s_cselect_b64 %3, -1, 0"
        : "=&s"(r1), "=&s"(off), "=&s"(rng), "=&s"(fm) : "s"(low), "s"(high), "s"(s.code) : "scc");
    asm("v_mul_hi_u32 %0, %1, %2\n\t"              // the eight interval starts floor(T[k] * range >> 16) of every symbol in tv
        "v_cndmask_b32_e64 %0, %0, %1, %3"         // range == 2^32: T[k] << 16
        : "=&v"(stv) : "v"(tv), "s"(rng), "s"(fm));
    asm("v_cmp_le_u32_e64 vcc, %12, %13\n\t"
        "s_or_b32 %7, %14, 0x80000\n\t"            // bit field: offset L, width 8
        "s_bfe_u64 %10, vcc, %7\n\t"
        "s_bcnt1_i32_b64 %8, %10\n\t"              // 1 + sym (start 0 = 0 always passes; the starts increase)
        "s_add_u32 %9, %14, %8\n\t"                // lane of start[sym + 1]
        "s_add_u32 %7, %9, -1\n\t"
        "v_readlane_b32 %5, %12, %7\n\t"           // lo = start[sym]
        "v_readlane_b32 %6, %12, %9\n\t"
        "s_add_u32 %6, %6, -1\n\t"                 // him1 = start[sym + 1] - 1 ...
        "s_cmp_eq_u32 %8, 8\n\t"
        "s_cselect_b32 %6, %11, %6\n\t"            // ... or range - 1 for the last symbol
        "s_add_u32 %2, %8, -1\n\t"
        "s_add_u32 %6, %0, %6\n\t"                 // ---- high' = low + him1, low' = low + lo
        "s_add_u32 %0, %0, %5\n\t"
        "s_not_b32 %7, %6\n\t"                     // high travels complemented
        "s_xor_b32 %5, %0, %6\n\t"
        "s_or_b32 %5, %5, 1\n\t"                   // (low == high: n1 = 31, flagged by ac_state_check at the end of the launch)
        "s_flbit_i32_b32 %3, %5\n\t"               // n1: leading bits on which low and high agree
        "s_lshl_b32 %0, %0, %3\n\t"
        "s_lshl_b32 %7, %7, %3\n\t"
        "s_and_b32 %5, %0, %7\n\t"                 // low = 01.., high = 10..
        "s_lshl_b32 %5, %5, 1\n\t"
        "s_not_b32 %5, %5\n\t"
        "s_flbit_i32_b32 %4, %5\n\t"               // n2: underflow run
        "s_min_u32 %4, %4, 30\n\t"
        "s_lshl_b32 %0, %0, %4\n\t"
        "s_bitset0_b32 %0, 31\n\t"
        "s_lshl_b32 %7, %7, %4\n\t"
        "s_orn2_b32 %1, 0x80000000, %7"
        : "+s"(low), "=&s"(high), "=&s"(sym), "=&s"(n1), "=&s"(n2), "=&s"(lo), "=&s"(hi), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(fl)
        : "s"(r1), "v"(stv), "s"(off), "s"(L)
        : "scc", "vcc");
    const uint32_t code = s.code;
    s.low = low; s.high = high;
    if (__builtin_expect(rd.pos >= rd.wbase + 256, 0)) rd.fetch_window();
    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)rd.win, (rd.pos - rd.wbase) >> 2);
    unsigned long long acc = rd.acc, bits, t64;
    int nacc = rd.nacc, pos = rd.pos, n, u0, u1;
    asm("s_lshl_b64 %4, %0, 32\n\t"                // acc << 32 | w
        "s_or_b64 %4, %4, %10\n\t"
        "s_add_u32 %5, %8, %9\n\t"                 // n
        "s_cmp_lt_i32 %1, %5\n\t"                  // fewer than n bits in the accumulator: append the next stream word
        "s_cselect_b64 %0, %4, %0\n\t"
        "s_cselect_b32 %6, 32, 0\n\t"
        "s_cselect_b32 %7, 4, 0\n\t"
        "s_add_u32 %2, %2, %7\n\t"                 // pos
        "s_add_u32 %1, %1, %6\n\t"
        "s_sub_u32 %1, %1, %5\n\t"                 // nacc
        "s_lshr_b64 %3, %0, %1\n\t"                // the next n bits: (acc >> nacc) & ones(n)
        "s_lshl_b64 %4, -1, %5\n\t"
        "s_andn2_b64 %3, %3, %4"
        : "+s"(acc), "+s"(nacc), "+s"(pos), "=&s"(bits), "=&s"(t64), "=&s"(n), "=&s"(u0), "=&s"(u1)
        : "s"(n1), "s"(n2), "s"((unsigned long long)w)
        : "scc");
    // code' = ((code << n) | bits) ^ (n2 ? TOP : 0), see above (n1 <= 31 and n2 <= 30: two 32-bit shifts)
    s.code = (((code << n1) << n2) | (uint32_t)bits) ^ ((uint32_t)(n2 != 0) << 31);
    rd.acc = acc; rd.nacc = nacc; rd.pos = pos;
    return sym;
}
__device__ __forceinline__ void ac_state_check(AcState &s) {
    const uint32_t r1 = s.high - s.low;
    if (s.low >= s.high || r1 < (1u << 30) + 1u) s.error = 2;
    if (s.code < s.low || s.code > s.high) s.error = 3;
}

// Decode of one plane runs as two kernels:
//  k_dec_tables -- one thread per (image, plane position): the 7 inner CDF entries of the symbol from the three nets' outputs,
//                  7 x int32 + a "coded" flag as two uint4 (massively parallel, register-hungry, short);
//  k_dec_plane  -- one wave per image: serial range decode on the wave-uniform state from those tables, then scatter of
//                  (sym-3.5 | 0) into the diagonal-major input of the next plane and of the decoded symbol into the NCHW
//                  output (= TileInput + `b[0:1] + 3.5*mask`, lic360_demo.py:222,236-237).  It needs < 40 VGPRs, so its
//                  waves fit next to the 12-wave workgroups of the conv kernels of the other streams (456 of 512 VGPRs
//                  per SIMD) instead of keeping whole CUs away from them for the ~0.4 ms the serial chain takes.
// one struct by value instead of 16 scalar kernel arguments: the number of arguments of a per-plane launch costs like their bytes do (DESIGN.md 4.1 b'')
struct DecTablesArgs {
    const float *y;
    const float *mask;
    const int *idx;
    int start;
    int len;
    int p;
    uint4 *tab;
    int tab_pitch;
    int B;
    int G;
    int H;
    int W;
    int sk_rows;
    int sk_pitch;
    int sk_row0;
    int sk_col0;
};
__global__ __launch_bounds__(64) void k_dec_tables(const DecTablesArgs a) {
    const float *__restrict__ y = a.y;
    const float *__restrict__ mask = a.mask;
    const int *__restrict__ idx = a.idx;
    const int start = a.start;
    const int len = a.len;
    const int p = a.p;
    uint4 *__restrict__ tab = a.tab;
    const int tab_pitch = a.tab_pitch;
    const int B = a.B;
    const int G = a.G;
    const int H = a.H;
    const int W = a.W;
    const int sk_rows = a.sk_rows;
    const int sk_pitch = a.sk_pitch;
    const int sk_row0 = a.sk_row0;
    const int sk_col0 = a.sk_col0;
    const int b = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    if (i >= len) return;
    const int HW = H * W;
    const long SK = (long)sk_rows * sk_pitch;
    const int q = start + i;
    const int th = idx[q], tw = idx[q + HW], g = p - th - tw;
    const long nchw = (((long)b * G + g) * H + th) * W + tw;
    uint4 r = make_uint4(0u, 0u, 0u, 0u), r2 = r;
    if (!(mask[nchw] < 0.5f)) {                                      // coder.cpp:79
        float v[9];
#pragma unroll
        for (int net = 0; net < 3; ++net)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[net * 3 + c] = y[((long)(net * B + b) * (3 * G) + g * 3 + c) * SK + (long)(th + tw + sk_row0) * sk_pitch + th + sk_col0];
        int T[9];
        gmm_cdf9(v, v + 3, v + 6, T);                                // 0 = T[0] < T[1] < ... < T[8] = 65536
        dec_pack8(T, r, r2);
    }
    tab[((long)b * tab_pitch + i) * 2] = r;
    tab[((long)b * tab_pitch + i) * 2 + 1] = r2;
}

// LINEAR (test hook lic360_devcoder_decode): symbols go to code_out[b*G + start + i] instead of the latent layouts
// one struct by value instead of 19 scalar kernel arguments: the number of arguments of a per-plane launch costs like their bytes do (DESIGN.md 4.1 b'')
struct DecPlaneArgs {
    const uint4 *tab;
    int tab_pitch;
    const int *idx;
    int start;
    int len;
    int p;
    AcDevState *state;
    const uint8_t *bytes;
    long cap;
    const int *nbytes;
    float *x0;
    float *code_out;
    int G;
    int H;
    int W;
    int sk_rows;
    int sk_pitch;
    int sk_row0;
    int sk_col0;
};
template <bool LINEAR>
__global__ __launch_bounds__(64) void k_dec_plane(const DecPlaneArgs a) {
    const uint4 *__restrict__ tab = a.tab;
    const int tab_pitch = a.tab_pitch;
    const int *__restrict__ idx = a.idx;
    const int start = a.start;
    const int len = a.len;
    const int p = a.p;
    AcDevState *__restrict__ state = a.state;
    const uint8_t *__restrict__ bytes = a.bytes;
    const long cap = a.cap;
    const int *__restrict__ nbytes = a.nbytes;
    float *__restrict__ x0 = a.x0;
    float *__restrict__ code_out = a.code_out;
    const int G = a.G;
    const int H = a.H;
    const int W = a.W;
    const int sk_rows = a.sk_rows;
    const int sk_pitch = a.sk_pitch;
    const int sk_row0 = a.sk_row0;
    const int sk_col0 = a.sk_col0;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int HW = H * W;
    const long SK = (long)sk_rows * sk_pitch;
    AcDevState ds = state[b];
    AcState st;
    // (the state goes through inline assembly with scalar-register operands: make sure it IS scalar)
    auto sgpr = [](uint32_t v) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    st.low = sgpr(ds.low); st.high = sgpr(ds.high); st.code = sgpr(ds.code); st.underflow = 0; st.error = 0;
    DevBits rd;
    rd.buf = bytes + (long)b * cap; rd.len = dev_stream_len(nbytes[b], cap); rd.pos = (int)sgpr((uint32_t)ds.pos);
    rd.acc = ((unsigned long long)sgpr((uint32_t)(ds.acc >> 32)) << 32) | sgpr((uint32_t)ds.acc); rd.nacc = (int)sgpr((uint32_t)ds.nacc); rd.lane = lane;
    rd.fetch_window();
    // Tables: eight VGPRs hold the 8 x 8 words of a group's 64 symbols (lane 8q + k of register r = word k of symbol 8r + q); a
    // register is refilled with the next group's words as soon as its eight symbols are decoded, so that no memory latency sits
    // between groups.  The scan positions of the next group are fetched before the serial chain of the current one runs.
    const uint32_t *const tw32 = (const uint32_t *)(tab + (long)b * tab_pitch * 2);
    auto load_tab = [&](int base, int r) __attribute__((always_inline)) {
        uint32_t t = 0u;
        if (base + 8 * r + (lane >> 3) < len) t = tw32[(long)(base + 8 * r) * 8 + lane];
        return t;
    };
    auto load_pos = [&](int base, int &th, int &tw) __attribute__((always_inline)) {
        th = tw = 0;
        if constexpr (!LINEAR) { if (base + lane < len) { th = idx[start + base + lane]; tw = idx[start + base + lane + HW]; } }
    };
    uint32_t tv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) tv[r] = load_tab(0, r);
    int th_cur, tw_cur;
    load_pos(0, th_cur, tw_cur);
    for (int base = 0; base < len; base += 64) {
        const bool live = base + lane < len;
        int th_next, tw_next;
        load_pos(base + 64, th_next, tw_next);
        int symv = 0;
        unsigned long long coded_m = 0;                                 // bit j: symbol j of the group is coded
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // a coded symbol has T[7] != 0 (lane 8q + 7); its bit is moved to lane 8q
            unsigned long long todo = (__ballot(tv[r] != 0u) >> 7) & 0x0101010101010101ull;
            while (todo) {                                              // the coded symbols of this register, in order
                const int L = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int sym = ac_decode_symbol8_v(st, rd, tv[r], L);
                const int j = 8 * r + (L >> 3);
                coded_m |= 1ull << j;
                // (lane j of symv <- sym; clang exposes no writelane builtin; two SGPR sources: the lane select goes through m0, the s_nop covers its hazard)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(symv) : "s"(sym), "s"(j));
            }
            tv[r] = load_tab(base + 64, r);
        }
        const bool coded = (coded_m >> lane) & 1ull;
        if (live) {
            if constexpr (LINEAR) code_out[(long)b * G + start + base + lane] = coded ? (float)symv : 0.0f;
            else {
                const int th = th_cur, tw = tw_cur, g = p - th - tw;
                x0[((long)b * G + g) * SK + (long)(th + tw + sk_row0) * sk_pitch + th + sk_col0] = coded ? (float)symv - 3.5f : 0.0f;
                code_out[(((long)b * G + g) * H + th) * W + tw] = coded ? (float)symv : 0.0f;
            }
        }
        th_cur = th_next; tw_cur = tw_next;
    }
    ac_state_check(st);
    if (lane == 0) {
        ds.low = st.low; ds.high = st.high; ds.code = st.code; ds.error |= st.error;     // sticky: coder faults 1..3, clamp flag 32
        ds.pos = rd.pos; ds.acc = rd.acc; ds.nacc = rd.nacc;
        state[b] = ds;
    }
}

__global__ void k_collect_err(const AcDevState *__restrict__ state, int *__restrict__ err, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) err[b] = state[b].error;
}

// ------------------------------------------------------------------------------------------------ host side
template <class T>
static int dmalloc(T **p, size_t n) {
    HIP_TRY(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return 0;
}

// ------------------------------------------------------------------------------------------------ host leg of the coder (latency regime)
// Pinned, host-coherent buffers + polled flags between the GPU and one host thread per image; see lic360_codec::coder_mode.
#define HL_FB 16                               // flags: [0] tables of plane `seq` are in tab_h, [1] bitstreams of decode `gen` are in bytes_h, [2] abort,
#define HL_MAXB 64                             //        [HL_FB + i] image i's symbols of plane `seq` are in sym_h
#define HL_SPIN_LIMIT 2000000L                 // polls of a GPU-side wait before it gives up (~4 s: a host thread that lost its core on a busy box gets it back long before): a dead
                                               // host thread is an error, never a hang
struct HostLeg {
    int maxB = 0, tab_pitch = 0;
    long cap = 0, nsym = 0;
    uint2 *rec_h = nullptr;                    // [maxB][nsym] (cdf[sym], cdf[sym + 1]) records of an encode
    uint8_t *bytes_h = nullptr;                // [maxB][cap]
    int *nbytes_h = nullptr, *err_h = nullptr;
    unsigned short *tab_h = nullptr;           // [maxB][tab_pitch][8]: coded flag, T[1..7] of the current plane's symbols
    float *sym_h = nullptr;                    // [maxB][tab_pitch]: decoded symbols of the current plane (-1: not coded)
    int *flags = nullptr;
    int *d_ctr = nullptr;                      // device: arrival counter of the table kernel's workgroups
    int gen = 0;
    std::vector<std::thread> workers;
};
static inline int hl_ld(const int *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void hl_st(int *p, int v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static void hl_join(HostLeg *h) {
    for (std::thread &t : h->workers) if (t.joinable()) t.join();
    h->workers.clear();
}
static void hl_free(HostLeg *h) {
    if (!h) return;
    if (h->flags) hl_st(h->flags + 2, 1);                                 // release whoever still polls
    hl_join(h);
    (void)hipDeviceSynchronize();
    (void)hipHostFree(h->rec_h); (void)hipHostFree(h->bytes_h); (void)hipHostFree(h->nbytes_h); (void)hipHostFree(h->err_h);
    (void)hipHostFree(h->tab_h); (void)hipHostFree(h->sym_h); (void)hipHostFree(h->flags); (void)hipFree(h->d_ctr);
    delete h;
}
template <class T>
static int hl_alloc(T **p, size_t n) {
    HIP_TRY(hipHostMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent));
    return 0;
}
// buffers for B images of nsym symbols, bitstream slots of cap bytes, planes of at most tab_pitch symbols (grown on demand; growing synchronises)
static int hl_get(lic360_codec *c, int B, long cap, HostLeg **out) {
    ARG_CHECK(B > 0 && B <= HL_MAXB);
    HostLeg *h = c->hl;
    const long nsym = (long)c->G * c->HW;
    if (h && (h->maxB < B || h->cap < cap)) { hl_free(h); h = c->hl = nullptr; }
    if (!h) {
        h = new HostLeg();
        c->hl = h;
        h->maxB = std::max(B, std::min(c->maxB, c->coder_auto_max)); h->cap = cap; h->nsym = nsym; h->tab_pitch = c->tab_pitch;
        int rc = hl_alloc(&h->rec_h, (size_t)h->maxB * nsym) | hl_alloc(&h->bytes_h, (size_t)h->maxB * cap) | hl_alloc(&h->nbytes_h, h->maxB) |
                 hl_alloc(&h->err_h, h->maxB) | hl_alloc(&h->tab_h, (size_t)h->maxB * h->tab_pitch * 8) | hl_alloc(&h->sym_h, (size_t)h->maxB * h->tab_pitch) |
                 hl_alloc(&h->flags, HL_FB + HL_MAXB) | dmalloc(&h->d_ctr, 1);
        if (rc) return 1;
        memset(h->flags, 0, (HL_FB + HL_MAXB) * sizeof(int));
        HIP_TRY(hipMemset(h->d_ctr, 0, sizeof(int)));
    }
    *out = h;
    return 0;
}

// ---- encode: the records of the fused last layer, one host thread per image (ac_core.h: the coder the drop-in `Coder` runs)
struct HlEncJob { HostLeg *h; int B; long n, cap; };
static void hl_encode_image(HostLeg *h, int i, long n, long cap) {
    const uint2 *r = h->rec_h + (long)i * n;
    AcState st;
    ac_init(st);
    AcBitWriter bw;
    ac_bw_init(bw, h->bytes_h + (long)i * h->cap, cap);
    for (long k = 0; k < n; ++k) {
        const uint2 v = r[k];
        if (v.y) ac_encode_symbol(st, bw, v.x, v.y, 65536u);             // hi == 0: not coded (mask < 0.5, coder.cpp:79)
    }
    ac_encode_finish(st, bw);
    h->nbytes_h[i] = (int)bw.len;
    h->err_h[i] = st.error | (bw.len > cap ? 16 : 0);
}
static void hl_encode_all(void *ud) {                                   // (a stream callback: no HIP call in here)
    HlEncJob *j = (HlEncJob *)ud;
    std::vector<std::thread> th;
    int started = 1;
    try {
        for (int i = 1; i < j->B; ++i, ++started) th.emplace_back(hl_encode_image, j->h, i, j->n, j->cap);
    } catch (...) {}                                                    // (no more threads to be had: the rest runs here, one after the other)
    hl_encode_image(j->h, 0, j->n, j->cap);
    for (int i = started; i < j->B; ++i) hl_encode_image(j->h, i, j->n, j->cap);
    for (std::thread &t : th) t.join();
    delete j;
}
static int hl_encode(lic360_codec *c, hipStream_t s, int B, uint8_t *bytes, long cap, int *nbytes, int *err) {
    HostLeg *h;
    if (hl_get(c, B, cap, &h)) return 1;
    const long n = (long)c->G * c->HW;
    HIP_TRY(hipMemcpyAsync(h->rec_h, c->e_rec, (size_t)B * n * sizeof(uint2), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipLaunchHostFunc(s, hl_encode_all, new HlEncJob{h, B, n, cap}));
    HIP_TRY(hipMemcpy2DAsync(bytes, (size_t)cap, h->bytes_h, (size_t)h->cap, (size_t)cap, (size_t)B, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(nbytes, h->nbytes_h, (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(err, h->err_h, (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
    return 0;
}

// ---- decode: per plane  k_dec_tables_host -> (host threads) -> k_dec_wait_scatter
struct DecTablesHostArgs {
    const float *y, *mask;
    const int *idx;
    int start, len, p;
    uint4 *tab;                                 // pinned: [B][tab_pitch] x 8 halves
    int tab_pitch, B, G, H, W, sk_rows, sk_pitch, sk_row0, sk_col0;
    int *ctr, *flag;
    int seq, nblocks;
};
__global__ __launch_bounds__(64) void k_dec_tables_host(const DecTablesHostArgs a) {
    const int b = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    if (i < a.len) {
        const int HW = a.H * a.W;
        const long SK = (long)a.sk_rows * a.sk_pitch;
        const int q = a.start + i;
        const int th = a.idx[q], tw = a.idx[q + HW], g = a.p - th - tw;
        const long nchw = (((long)b * a.G + g) * a.H + th) * a.W + tw;
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (!(a.mask[nchw] < 0.5f)) {                                    // coder.cpp:79
            float v[9];
#pragma unroll
            for (int net = 0; net < 3; ++net)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    v[net * 3 + c] = a.y[((long)(net * a.B + b) * (3 * a.G) + g * 3 + c) * SK + (long)(th + tw + a.sk_row0) * a.sk_pitch + th + a.sk_col0];
            int T[9];
            gmm_cdf9(v, v + 3, v + 6, T);                                // 0 = T[0] < T[1] < ... < T[8] = 65536: the inner entries fit 16 bits
            r = make_uint4(1u | (unsigned)T[1] << 16, (unsigned)T[2] | (unsigned)T[3] << 16, (unsigned)T[4] | (unsigned)T[5] << 16, (unsigned)T[6] | (unsigned)T[7] << 16);
        }
        a.tab[(long)b * a.tab_pitch + i] = r;
    }
    // the last workgroup to arrive publishes the plane (lane 0 of a workgroup always owns a symbol)
    __threadfence_system();
    if (threadIdx.x == 0) {
        if (atomicAdd(a.ctr, 1) == a.nblocks - 1) {
            atomicExch(a.ctr, 0);
            __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
struct DecWaitArgs {
    int *flags;
    const float *sym;
    int seq, B, tab_pitch;
    const int *idx;
    int start, len, p;
    float *x0, *code_out;
    int G, H, W, sk_rows, sk_pitch, sk_row0, sk_col0;
};
__global__ __launch_bounds__(1024) void k_dec_wait_scatter(const DecWaitArgs a) {
    __shared__ int s_abort;
    const int tid = threadIdx.x;
    if (tid == 0) s_abort = 0;
    __syncthreads();
    if (tid < a.B) {                                                    // thread i waits for image i's host thread; every poll is a read of pinned host memory
        long spins = 0;
        while (__hip_atomic_load(a.flags + HL_FB + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
            if (__hip_atomic_load(a.flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || ++spins > HL_SPIN_LIMIT) { s_abort = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (s_abort) {                                                      // sticky: every later wait of this decode returns at once, err[] reports it
        if (tid == 0) __hip_atomic_store(a.flags + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const int HW = a.H * a.W;
    const long SK = (long)a.sk_rows * a.sk_pitch;
    for (int e = tid; e < a.B * a.len; e += 1024) {
        const int b = e / a.len, j = e - b * a.len;
        const float v = a.sym[(long)b * a.tab_pitch + j];
        const bool coded = v >= 0.0f;
        const int th = a.idx[a.start + j], tw = a.idx[a.start + j + HW], g = a.p - th - tw;
        a.x0[((long)b * a.G + g) * SK + (long)(th + tw + a.sk_row0) * a.sk_pitch + th + a.sk_col0] = coded ? v - 3.5f : 0.0f;   // = TileInput + `b[0:1] + 3.5*mask` (lic360_demo.py:222,236-237)
        a.code_out[(((long)b * a.G + g) * a.H + th) * a.W + tw] = coded ? v : 0.0f;
    }
}
__global__ void k_hl_signal(int *flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_hl_err(const int *err_h, const int *flags, int *err, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) err[b] = err_h[b] | (__hip_atomic_load((int *)flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? 128 : 0);   // 128: the host leg was aborted
}
// the thread of image i: ArithmeticDecoder::read (ArithmeticCoder.cpp:82-116) on the tables the GPU publishes plane by plane
static void hl_decode_worker(HostLeg *h, int i, long cap, int gen, std::vector<int> lens) {
    int *flags = h->flags;
    auto wait = [&](const int *f, int v) {
        while (hl_ld(f) < v) {
            if (hl_ld(flags + 2)) return false;
            _mm_pause();
        }
        return true;
    };
    if (!wait(flags + 1, gen)) return;
    long len = h->nbytes_h[i];
    int e0 = 0;
    if (len < 0) { len = 0; e0 = 32; }
    if (len > cap) { len = cap; e0 = 32; }                               // the stream length is never trusted (error bit 32 flags a clamp)
    AcBitReader rd;
    ac_br_init(rd, h->bytes_h + (long)i * h->cap, len);
    AcState st;
    ac_init(st);
    ac_decode_start(st, rd);
    const int seq0 = gen << 13;
    for (int p = 0; p < (int)lens.size(); ++p) {
        const int ln = lens[p];
        if (ln <= 0) continue;
        if (!wait(flags, seq0 + p + 1)) return;
        const unsigned short *t = h->tab_h + (long)i * h->tab_pitch * 8;
        float *o = h->sym_h + (long)i * h->tab_pitch;
        for (int j = 0; j < ln; ++j, t += 8) {
            if (!t[0]) { o[j] = -1.0f; continue; }
            const uint32_t target = ac_decode_target(st, 65536u);
            int sym = 0;
#pragma GCC unroll 7
            for (int k = 1; k < 8; ++k) sym += (uint32_t)t[k] <= target;
            ac_decode_consume(st, rd, sym ? t[sym] : 0u, sym == 7 ? 65536u : t[sym + 1], 65536u);
            o[j] = (float)sym;
        }
        h->err_h[i] = e0 | st.error;
        hl_st(flags + HL_FB + i, seq0 + p + 1);
    }
}

LIC360_API int lic360_codec_create(int ngroup, int h, int w, int max_batch, lic360_codec **out) {
    ARG_CHECK(out && ngroup > 0 && ngroup < 128 && h > 0 && w > 0 && h < 4096 && w < 4096 && max_batch > 0);
    lic360_codec *c = new lic360_codec();
    memset(c->layer_set, 0, sizeof(c->layer_set));
    c->G = ngroup; c->H = h; c->W = w; c->maxB = max_batch; c->S = h + w - 1; c->P = h + w + ngroup - 2; c->HW = h * w;
    for (int i = 0; i < 12; ++i) c->packed[i] = c->bias[i] = c->act[i] = c->packed4[i] = c->packed16[i] = nullptr;
    int rc = 0;
    rc |= lic360_conv_plan_create(ngroup * 1, ngroup, ngroup * 4, 5, 5, &c->plan[0]);
    rc |= lic360_conv_plan_create(ngroup * 4, ngroup, ngroup * 4, 5, 6, &c->plan[1]);
    rc |= lic360_conv_plan_create(ngroup * 4, ngroup, ngroup * 3, 5, 6, &c->plan[2]);
    if (rc) return 1;
    const char *force = getenv("LIC360_FUSED_CONV");                  // "16" forces the generic 16x16x4 kernels (the fall-back path, kept tested)
    c->use4 = lic360_conv4_supported(c->plan[0]) && lic360_conv4_supported(c->plan[1]) && lic360_conv4_supported(c->plan[2]) &&
              lic360_conv16_supported(c->plan[0]) && lic360_conv16_supported(c->plan[1]) && lic360_conv16_supported(c->plan[2]) &&
              !(force && force[0] == '1' && force[1] == '6');
    c->h_idx.resize(2 * (size_t)c->HW);
    c->h_pidx.resize(h + w);
    lic360_code_contex(h, w, c->h_idx.data(), c->h_pidx.data());
    c->h_plane_start.resize(c->P + 1);
    int acc = 0;
    for (int p = 0; p < c->P; ++p) {
        int st, ln;
        lic360_plane_window(p, ngroup, h, w, c->h_pidx.data(), &st, &ln);
        c->h_plane_start[p] = acc;
        acc += ln;
    }
    c->h_plane_start[c->P] = acc;
    if (acc != ngroup * c->HW) { lic360_set_error("internal: plane schedule does not cover the latent"); return 1; }
    rc |= dmalloc(&c->d_idx, c->h_idx.size());
    rc |= dmalloc(&c->d_pidx, c->h_pidx.size());
    rc |= dmalloc(&c->d_plane_start, c->h_plane_start.size());
    if (rc) return 1;
    HIP_TRY(hipMemcpy(c->d_idx, c->h_idx.data(), c->h_idx.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_pidx, c->h_pidx.data(), c->h_pidx.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_plane_start, c->h_plane_start.data(), c->h_plane_start.size() * 4, hipMemcpyHostToDevice));
    if (c->use4) { if (lic360_dc4_layout(h, w, &c->sk_rows, &c->sk_pitch, &c->sk_row0, &c->sk_col0)) return 1; }
    else { c->sk_rows = c->S; c->sk_pitch = h; c->sk_row0 = 0; c->sk_col0 = 0; }
    const size_t B = max_batch, G = ngroup, HW = c->HW, SK = (size_t)c->sk_rows * c->sk_pitch, TAIL = (size_t)lic360_conv4_buffer_floats(0, 1, h, w) - (size_t)c->sk_rows * c->sk_pitch;   // slack for the band fetches of the last plane
    c->dc_mode = lic360_dc4_env_mode();
    // encode order: 16x16x4 MFMA kernels on zero-haloed NCHW planes, or plain NCHW for the generic kernels
    if (c->use4) { if (lic360_ec16_layout(h, w, &c->e_hp, &c->e_wp)) return 1; c->e_off = 2; }
    else { c->e_hp = h; c->e_wp = w; c->e_off = 0; }
    rc |= dmalloc(&c->e_ctr, 8);
    const size_t EPL = (size_t)c->e_hp * c->e_wp;
    rc |= dmalloc(&c->e_x0, B * G * EPL + TAIL);
    for (int i = 0; i < 3; ++i) rc |= dmalloc(&c->e_buf[i], 3 * B * 4 * G * EPL + TAIL);
    rc |= dmalloc(&c->e_rec, B * G * HW);
    rc |= dmalloc(&c->d_x0, B * G * SK + TAIL);
    for (int i = 0; i < 11; ++i) rc |= dmalloc(&c->d_act[i], 3 * B * 4 * G * SK + TAIL);
    rc |= dmalloc(&c->d_y, 3 * B * 3 * G * SK + TAIL);
    rc |= dmalloc(&c->d_state, B);
    for (int p = 0; p < c->P; ++p) c->tab_pitch = std::max(c->tab_pitch, c->h_plane_start[p + 1] - c->h_plane_start[p]);
    c->tab_pitch = (c->tab_pitch + 63) / 64 * 64;
    rc |= dmalloc(&c->d_tab, 2 * B * (size_t)c->tab_pitch);                 // two uint4 per symbol (dec_pack8)
    if (const char *hc = getenv("LIC360_HOST_CODER")) c->coder_mode = hc[0] == '0' ? 0 : (hc[0] == '1' ? 1 : 2);
    c->skip = c->use4 && !getenv("LIC360_NOSKIP");
    if (c->skip) {
        const size_t S = c->S, nt = (size_t)((h + 3) / 4) * ((w + 15) / 16);
        rc |= dmalloc(&c->need, B * NEED_LAYERS * HW);
        rc |= dmalloc(&c->need_d, B * NEED_LAYERS * S * h);
        rc |= dmalloc(&c->tmax, B * NEED_LAYERS * nt);
        rc |= dmalloc(&c->stats, 2 * NEED_LAYERS * NEED_STAT_G);
        c->ecl.cap = (int)(((3 * B + 7) / 8) * ((nt + 3) / 4) * ((G + 3) / 4));          // tasks of an XCD's list, hidden layers (the fused layer's is shorter)
        const size_t cap11 = ((B + 7) / 8) * ((nt + 1) / 2) * ((G + 4) / 5);
        if ((size_t)c->ecl.cap < cap11) c->ecl.cap = (int)cap11;
        rc |= dmalloc(&c->ecl.list, (size_t)NEED_LAYERS * 8 * c->ecl.cap);
        rc |= dmalloc(&c->ecl.cnt, (size_t)NEED_LAYERS * 8);
        if (h <= 64 && max_batch % 8 == 0 && max_batch >= 16 && max_batch <= 512 && ngroup <= 72) {
            c->dcl.P = c->P;
            c->dcl.cap = (int)(((G + 2) / 3) * 3 * (B / 8));
            rc |= dmalloc(&c->dcl.list, (size_t)NEED_LAYERS * c->P * 8 * c->dcl.cap);
            rc |= dmalloc(&c->dcl.cnt, (size_t)NEED_LAYERS * c->P * 8);
        }
        if (rc) return 1;
        HIP_TRY(hipMemset(c->stats, 0, 2 * NEED_LAYERS * NEED_STAT_G * sizeof(unsigned long long)));
    }
    if (rc) return 1;
    // decode activations are only ever read where already written or with a zero weight; they must be finite
    HIP_TRY(hipMemset(c->e_x0, 0, (B * G * EPL + TAIL) * 4));
    for (int i = 0; i < 3; ++i) HIP_TRY(hipMemset(c->e_buf[i], 0, (3 * B * 4 * G * EPL + TAIL) * 4));
    HIP_TRY(hipMemset(c->d_x0, 0, (B * G * SK + TAIL) * 4));
    for (int i = 0; i < 11; ++i) HIP_TRY(hipMemset(c->d_act[i], 0, (3 * B * 4 * G * SK + TAIL) * 4));
    HIP_TRY(hipMemset(c->d_y, 0, (3 * B * 3 * G * SK + TAIL) * 4));
    *out = c;
    return 0;
}

LIC360_API void lic360_codec_destroy(lic360_codec *c) {
    if (!c) return;
    hl_free(c->hl);
    for (int i = 0; i < 3; ++i) lic360_conv_plan_destroy(c->plan[i]);
    for (int i = 0; i < 12; ++i) { (void)hipFree(c->packed[i]); (void)hipFree(c->bias[i]); (void)hipFree(c->act[i]); (void)hipFree(c->packed4[i]); }
    (void)hipFree(c->d_idx); (void)hipFree(c->d_pidx); (void)hipFree(c->d_plane_start);
    (void)hipFree(c->e_x0); for (int i = 0; i < 3; ++i) (void)hipFree(c->e_buf[i]);
    (void)hipFree(c->e_rec); (void)hipFree(c->d_x0); for (int i = 0; i < 11; ++i) (void)hipFree(c->d_act[i]);
    (void)hipFree(c->d_y); (void)hipFree(c->d_state); (void)hipFree(c->d_tab); (void)hipFree(c->e_ctr);
    (void)hipFree(c->need); (void)hipFree(c->need_d); (void)hipFree(c->tmax); (void)hipFree(c->stats);
    (void)hipFree(c->ecl.list); (void)hipFree(c->ecl.cnt); (void)hipFree(c->dcl.list); (void)hipFree(c->dcl.cnt);
    for (int i = 0; i < 12; ++i) (void)hipFree(c->packed16[i]);
    for (int k = 0; k < PROF_NCLS; ++k)
        for (hipEvent_t e : c->ev[k]) (void)hipEventDestroy(e);
    delete c;
}

LIC360_API int lic360_codec_set_layer(void *stream, lic360_codec *c, int layer, const float *weight, const float *bias, const float *act) {
    ARG_CHECK(c && layer >= 0 && layer < 12 && weight && bias);
    ARG_CHECK((act != nullptr) == (layer != 11));
    lic360_conv_plan *p = c->plan[plan_of(layer)];
    long nper = lic360_conv_plan_packed_floats(p);
    if (!c->packed[layer]) {
        if (dmalloc(&c->packed[layer], 3 * (size_t)nper)) return 1;
        if (dmalloc(&c->bias[layer], 3 * (size_t)p->nout)) return 1;
        if (act && dmalloc(&c->act[layer], 3 * (size_t)p->nout)) return 1;
        if (c->use4 && dmalloc(&c->packed4[layer], 3 * (size_t)lic360_conv4_packed_floats(p))) return 1;
        if (c->use4 && dmalloc(&c->packed16[layer], 3 * (size_t)lic360_conv16_packed_floats(p))) return 1;
    }
    if (lic360_conv_pack(stream, p, weight, 3, c->packed[layer])) return 1;
    if (c->use4 && lic360_conv4_pack(stream, p, weight, 3, c->packed4[layer])) return 1;
    // (the fused last layer + CDF tables reads a packing of its own: five groups per block)
    if (c->use4 && (layer == 11 ? lic360_conv16_pack_tables(stream, p, weight, 3, c->packed16[layer])
                                : lic360_conv16_pack(stream, p, weight, 3, c->packed16[layer]))) return 1;
    HIP_TRY(hipMemcpyAsync(c->bias[layer], bias, 3 * (size_t)p->nout * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (act) HIP_TRY(hipMemcpyAsync(c->act[layer], act, 3 * (size_t)p->nout * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    c->layer_set[layer] = true;
    return 0;
}

static bool coder_on_host(const lic360_codec *c, int B) {
    return c->coder_mode == 1 ? B <= HL_MAXB : (c->coder_mode == 2 && B <= c->coder_auto_max);
}
// 0: the serial coder phases run on the GPU (one wave per image), 1: on host threads (B <= 64), 2: auto -- host for calls of at most 8 images (the
// latency regime: the GPU coder only pays where other images' convolutions hide it)
LIC360_API int lic360_codec_set_coder(lic360_codec *c, int mode) {
    ARG_CHECK(c && mode >= 0 && mode <= 2);
    c->coder_mode = mode;
    return 0;
}
static int check_ready(const lic360_codec *c, int B) {
    ARG_CHECK(c && B > 0 && B <= c->maxB);
    for (int i = 0; i < 12; ++i)
        if (!c->layer_set[i]) { lic360_set_error("codec layer %d has no weights (call lic360_codec_set_layer)", i); return 2; }
    return 0;
}

LIC360_API int lic360_codec_encode(void *stream, lic360_codec *c, const float *code, const float *mask, int B,
                                   uint8_t *bytes, long cap, int *nbytes, int *err) {
    if (check_ready(c, B)) return 2;
    ARG_CHECK(code && mask && bytes && nbytes && err && cap > 0 && cap < (1L << 31) && cap % 4 == 0 && ((uintptr_t)bytes & 3) == 0);   // DevBitSink: dword stores at bytes + b*cap
    hipStream_t s = (hipStream_t)stream;
    const int G = c->G, H = c->H, W = c->W;
    const long total = (long)B * G * c->HW;
    hipLaunchKernelGGL(k_enc_prep, dim3(lic360_blocks(total, 4)), dim3(256), 0, s, code, mask, c->e_x0, total, H, W, c->e_hp, c->e_wp, c->e_off);
    LAUNCH_CHECK();
    float *cur = c->e_buf[0], *t1 = c->e_buf[1], *nxt = c->e_buf[2];
    if (c->skip) {                                                      // need maps of this batch's masks, the live tasks of layers 1..11
        if (lic360_need_build(stream, mask, B, G, H, W, c->need, c->need_d, c->tmax)) return 1;
        if (lic360_ec_lists_build(stream, c->tmax, B, G, H, W, c->ecl, c->stats_on ? c->stats : nullptr)) return 1;
    }
    auto ec = [&](int layer, const float *xin, const float *res, float *dst, int x_mod) -> int {
        lic360_conv_plan *p = c->plan[plan_of(layer)];
        if (c->use4 && c->skip && layer > 0)
            return lic360_cconv16_ec_list(stream, p, xin, c->packed16[layer], c->bias[layer], c->act[layer], res, dst, 3 * B, H, W, 3, x_mod, c->e_ctr,
                                          c->ecl.list + (size_t)layer * 8 * c->ecl.cap, c->ecl.cnt + layer * 8, c->ecl.cap);
        if (c->use4) return lic360_cconv16_ec(stream, p, xin, c->packed16[layer], c->bias[layer], c->act[layer], res, dst, 3 * B, H, W, 3, x_mod, c->e_ctr);
        return lic360_cconv_ec_ex(stream, p, xin, c->packed[layer], c->bias[layer], c->act[layer], res, dst, 3 * B, H, W, 3, x_mod);
    };
    int rc = 0;
    PROF(c, PROF_EC_FIRST, s, rc |= ec(0, c->e_x0, nullptr, cur, B));
    for (int blk = 0; blk < 5 && !rc; ++blk) {
        int a = 1 + 2 * blk, b2 = 2 + 2 * blk;
        PROF(c, PROF_EC_HIDDEN, s, rc |= ec(a, cur, nullptr, t1, 3 * B));
        PROF(c, PROF_EC_HIDDEN, s, rc |= ec(b2, t1, cur, nxt, 3 * B));
        float *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (rc) return 1;
    if (c->use4) {
        // last layer + CDF tables in one kernel: the nets' outputs never reach HBM (no y buffer, no k_enc_tables)
        if (c->skip) {
            // a skipped (tile, group block) holds masked symbols only: their records are (0, 0) and nobody writes them
            HIP_TRY(hipMemsetAsync(c->e_rec, 0, (size_t)B * G * c->HW * sizeof(uint2), s));
            PROF(c, PROF_EC_LAST, s, rc |= lic360_cconv16_ec_tables_list(stream, c->plan[2], cur, c->packed16[11], c->bias[11], code, mask, c->d_pidx, c->d_plane_start,
                                                                         c->e_rec, B, H, W, c->e_ctr, c->ecl.list + (size_t)11 * 8 * c->ecl.cap, c->ecl.cnt + 11 * 8, c->ecl.cap));
        } else
        PROF(c, PROF_EC_LAST, s, rc |= lic360_cconv16_ec_tables(stream, c->plan[2], cur, c->packed16[11], c->bias[11], code, mask, c->d_pidx,
                                                                c->d_plane_start, c->e_rec, B, H, W, c->e_ctr));
        if (rc) return 1;
        if (coder_on_host(c, B)) {
            PROF(c, PROF_AC_ENCODE, s, rc |= hl_encode(c, s, B, bytes, cap, nbytes, err));
            return rc ? 1 : 0;
        }
        PROF(c, PROF_AC_ENCODE, s, hipLaunchKernelGGL(k_ac_encode, dim3(B), dim3(128), 0, s, c->e_rec, (long)G * c->HW, bytes, cap, nbytes, err));
        LAUNCH_CHECK();
        return 0;
    }
    PROF(c, PROF_EC_LAST, s, rc |= ec(11, cur, nullptr, t1, 3 * B));
    if (rc) return 1;
    PROF(c, PROF_ENC_TABLES, s, hipLaunchKernelGGL(k_enc_tables, dim3(lic360_blocks(total, 1)), dim3(256), 0, s, t1, code, mask, c->d_pidx,
                                                   c->d_plane_start, c->e_rec, B, G, H, W, c->e_hp, c->e_wp, c->e_off));
    LAUNCH_CHECK();
    if (coder_on_host(c, B)) {
        PROF(c, PROF_AC_ENCODE, s, rc |= hl_encode(c, s, B, bytes, cap, nbytes, err));
        return rc ? 1 : 0;
    }
    PROF(c, PROF_AC_ENCODE, s, hipLaunchKernelGGL(k_ac_encode, dim3(B), dim3(128), 0, s, c->e_rec, (long)G * c->HW, bytes, cap, nbytes, err));
    LAUNCH_CHECK();
    return 0;
}

static int codec_decode_impl(void *stream, lic360_codec *c, const uint8_t *bytes, long cap, const int *nbytes,
                             const float *mask, int B, float *code_out, int *err, void *const *gate, int n_gate, int gate_stride) {
    if (check_ready(c, B)) return 2;
    ARG_CHECK(bytes && nbytes && mask && code_out && err && cap > 0 && cap < (1L << 31) && cap % 4 == 0 && ((uintptr_t)bytes & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    const int G = c->G, H = c->H, W = c->W;
    const int *pih = c->h_pidx.data();
    // the serial decoder: one wave per image on the GPU, or one host thread per image fed through pinned memory (few images per call)
    const bool host = coder_on_host(c, B);
    HostLeg *h = nullptr;
    int seq0 = 0;
    if (host) {
        ARG_CHECK(c->P < (1 << 13));
        if (hl_get(c, B, cap, &h)) return 1;
        hl_join(h);                                                     // the previous decode's threads end with its last plane
        if (hl_ld(h->flags + 2) || h->gen >= (1 << 17)) {               // an aborted decode, or the sequence numbers run out: nothing polls after a synchronisation
            HIP_TRY(hipDeviceSynchronize());
            memset(h->flags, 0, (HL_FB + HL_MAXB) * sizeof(int));
            h->gen = 0;
        }
        const int gen = ++h->gen;
        seq0 = gen << 13;
        HIP_TRY(hipMemcpy2DAsync(h->bytes_h, (size_t)h->cap, bytes, (size_t)cap, (size_t)cap, (size_t)B, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h->nbytes_h, nbytes, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(k_hl_signal, dim3(1), dim3(1), 0, s, h->flags + 1, gen);
        LAUNCH_CHECK();
        std::vector<int> lens(c->P);
        for (int p = 0; p < c->P; ++p) { int st0, ln; lic360_plane_window(p, G, H, W, pih, &st0, &ln); lens[p] = ln; }
        try {
            for (int i = 0; i < B; ++i) h->workers.emplace_back(hl_decode_worker, h, i, cap, gen, lens);
        } catch (...) {                                                 // (thread creation failed: release the ones that started, report)
            hl_st(h->flags + 2, 1);
            hl_join(h);
            lic360_set_error("host leg of the coder: cannot start %d decode threads (lic360_codec_set_coder(codec, 0) keeps the coder on the GPU)", B);
            return 1;
        }
    } else {
        hipLaunchKernelGGL(k_dec_init, dim3((B + 63) / 64), dim3(64), 0, s, bytes, cap, nbytes, c->d_state, B);
        LAUNCH_CHECK();
    }
    // dead-cone skip: the whole mask must be final before the first plane (a gated decode waits for the map's LAST event instead of plane by plane)
    const bool lists = c->skip && c->dcl.list && B % 8 == 0 && B >= 16;
    if (lists) {
        if (gate) { HIP_TRY(hipStreamWaitEvent(s, (hipEvent_t)gate[n_gate - 1], 0)); gate = nullptr; }
        if (lic360_need_build(stream, mask, B, G, H, W, c->need, c->need_d, c->tmax)) return 1;
        if (lic360_dc_lists_build(stream, c->need_d, B, G, H, W, c->dcl, c->stats_on ? c->stats + NEED_LAYERS * NEED_STAT_G : nullptr)) return 1;
    }
    auto dc = [&](int layer, const float *xin, const float *res, float *dst, int x_mod, int p) -> int {
        lic360_conv_plan *pl = c->plan[plan_of(layer)];
        if (lists && layer > 0) {
            const size_t li = ((size_t)layer * c->P + p) * 8;
            return lic360_cconv4_dc_plane_list(stream, pl, xin, c->packed4[layer], c->bias[layer], c->act[layer], res, dst, 3 * B, H, W, 3, p, x_mod,
                                               c->dcl.list + li * c->dcl.cap, c->dcl.cnt + li, c->dcl.cap);
        }
        if (c->use4) return lic360_cconv4_dc_plane_mode(stream, pl, xin, c->packed4[layer], c->bias[layer], c->act[layer], res, dst, 3 * B, H, W, 3, p, x_mod, c->dc_mode);
        return lic360_cconv_dc_plane_ex(stream, pl, xin, c->packed[layer], c->bias[layer], c->act[layer], res, dst, 3 * B, H, W, 3,
                                        c->d_idx, c->d_pidx, pih, p, x_mod, 1);
    };
    int gate_q = -1;
    for (int p = 0; p < c->P; ++p) {
        // plane p of all 12 layers (x0 already holds planes < p)
        int rc = 0;
        PROF(c, PROF_DC_FIRST, s, rc |= dc(0, c->d_x0, nullptr, c->d_act[0], B, p));
        for (int blk = 0; blk < 5 && !rc; ++blk) {
            int a = 1 + 2 * blk, b2 = 2 + 2 * blk;
            PROF(c, PROF_DC_HIDDEN, s, rc |= dc(a, c->d_act[a - 1], nullptr, c->d_act[a], 3 * B, p));
            PROF(c, PROF_DC_HIDDEN, s, rc |= dc(b2, c->d_act[a], c->d_act[a - 1], c->d_act[b2], 3 * B, p));
        }
        if (rc) return 1;
        PROF(c, PROF_DC_LAST, s, rc |= dc(11, c->d_act[10], nullptr, c->d_y, 3 * B, p));
        if (rc) return 1;
        int start, len;
        lic360_plane_window(p, G, H, W, pih, &start, &len);
        if (len <= 0) continue;
        if (gate) {                                                     // the mask of this plane's positions: map planes <= p / stride (see lic360_impcodec_decode_masked)
            const int q = std::min(n_gate - 1, p / gate_stride);
            if (q != gate_q) { HIP_TRY(hipStreamWaitEvent(s, (hipEvent_t)gate[q], 0)); gate_q = q; }
        }
        if (host) {
            const int nblk = (len + 63) / 64 * B;
            PROF(c, PROF_DEC_TABLES, s, hipLaunchKernelGGL(k_dec_tables_host, dim3((len + 63) / 64, B), dim3(64), 0, s, DecTablesHostArgs{c->d_y, mask, c->d_idx, start, len, p,
                                                           (uint4 *)h->tab_h, h->tab_pitch, B, G, H, W, c->sk_rows, c->sk_pitch, c->sk_row0, c->sk_col0, h->d_ctr, h->flags, seq0 + p + 1, nblk}));
            LAUNCH_CHECK();
            PROF(c, PROF_DEC_PLANE, s, hipLaunchKernelGGL(k_dec_wait_scatter, dim3(1), dim3(1024), 0, s, DecWaitArgs{h->flags, h->sym_h, seq0 + p + 1, B, h->tab_pitch, c->d_idx, start, len, p,
                                                          c->d_x0, code_out, G, H, W, c->sk_rows, c->sk_pitch, c->sk_row0, c->sk_col0}));
            LAUNCH_CHECK();
            continue;
        }
        PROF(c, PROF_DEC_TABLES, s, hipLaunchKernelGGL(k_dec_tables, dim3((len + 63) / 64, B), dim3(64), 0, s, DecTablesArgs{c->d_y, mask, c->d_idx, start, len, p,
                                                       c->d_tab, c->tab_pitch, B, G, H, W, c->sk_rows, c->sk_pitch, c->sk_row0, c->sk_col0}));
        LAUNCH_CHECK();
        PROF(c, PROF_DEC_PLANE, s, hipLaunchKernelGGL(k_dec_plane<false>, dim3(B), dim3(64), 0, s, DecPlaneArgs{c->d_tab, c->tab_pitch, c->d_idx, start, len, p,
                                                      c->d_state, bytes, cap, nbytes, c->d_x0, code_out, G, H, W, c->sk_rows, c->sk_pitch,
                                                      c->sk_row0, c->sk_col0}));
        LAUNCH_CHECK();
    }
    if (host) hipLaunchKernelGGL(k_hl_err, dim3((B + 63) / 64), dim3(64), 0, s, h->err_h, h->flags, err, B);
    else hipLaunchKernelGGL(k_collect_err, dim3((B + 63) / 64), dim3(64), 0, s, c->d_state, err, B);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_codec_decode(void *stream, lic360_codec *c, const uint8_t *bytes, long cap, const int *nbytes,
                                   const float *mask, int B, float *code_out, int *err) {
    return codec_decode_impl(stream, c, bytes, cap, nbytes, mask, B, code_out, err, nullptr, 0, 1);
}
// ------------------------------------------------------------------------------------------------ importance-map stream
// Device-resident counterpart of ImpEntEncoderFast / ImpEntDecoder (test/lic360_demo.py:143-189, 241-290): one group, a
// 12-layer spatially causal net with `cpg` hidden channels, an nsym-way softmax table per position
// (entropy_table_cuda.cu:24-96), symbols = importance levels, input scale 2/(nsym-2) and bias -1.  The convolutions run on
// the generic 16x16x4 kernels (cin = cpg is outside the leaf-resident kernel's shapes); tables and the coder on the GPU.
struct lic360_impcodec {
    int H, W, HW, P, cpg, nsym, maxB;
    float sc;
    lic360_conv_plan *plan[3];
    float *packed[12], *bias[12], *act[12];
    bool layer_set[12];
    std::vector<int> h_idx, h_pidx;
    int *d_idx, *d_pidx;
    float *e_x0, *e_buf[3];
    uint2 *e_rec;
    float *d_x0, *d_act[11], *d_y;
    // leaf-resident 16x16x4 kernels for the 144-channel layers (csrc/cconv144_kernels.hip): encode on zero-haloed NCHW planes,
    // decode on zero-padded diagonal-major planes [rows = sk_rows][sk_pitch], cell (th, tw) at (th + tw + sk_row0, th + sk_col0)
    bool use144 = false;
    float *packed144[12];
    int e_hp = 0, e_wp = 0;
    float *e_pad[3] = {nullptr, nullptr, nullptr}, *e_plain = nullptr;
    int sk_rows, sk_pitch, sk_row0, sk_col0;
    int *d_tab;                                 // [maxB][tab_pitch][IMP_TW] tables of the current plane
    int tab_pitch;
    AcDevState *d_state;
    // lic360_impcodec_decode_masked: one event per plane ("the latent mask of every map cell of planes <= p is final").  The events are
    // re-recorded by every masked decode and hipStreamWaitEvent latches whichever record came last, so a gated latent decode is only
    // valid when it is enqueued AFTER the masked decode of the same step: decode_masked bumps gate_gen and arms the gate, decode_gated
    // must present that generation and consumes it (a stale, reused or out-of-order gate is an error, not a silent wrong mask).
    std::vector<hipEvent_t> plane_ev;
    long gate_gen = 0;
    bool gate_armed = false;
    int gate_stride = 0, gate_B = 0;           // gate_B: the batch whose mask rows the armed masked decode fills
    const float *gate_mask = nullptr;
};
#define IMP_TW 64                              // ints per table row (nsym + 1 <= 64), one per lane

// plain NCHW planes -> interior of zero-haloed planes [hp][wp] (cell (r, c) at (r + 2, c + 2))
__global__ void k_imp_halo(const float *__restrict__ in, float *__restrict__ out, long total, int H, int W, int hp, int wp) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % W), r = (int)((i / W) % H);
        const long pl = i / ((long)H * W);
        out[pl * hp * wp + (long)(r + 2) * wp + c + 2] = in[i];
    }
}
__global__ void k_imp_prep(const float *__restrict__ lv, float *__restrict__ x0, long total, float sc) {
    GRID_STRIDE(i, total) x0[i] = lic360_affine(lv[i], sc, -1.0f);                 // Scale(-1, 2/47): lic360_demo.py:168
}
// nsym-way softmax CDF of one position (entropy_table_cuda.cu:24-96)
__device__ __forceinline__ void imp_table(const float *__restrict__ y, long base, long cstride, int nsym, float *T) {
    float tmp[64], lg[64];
    for (int i = 0; i < nsym; ++i) lg[i] = y[base + i * cstride];
    lic360_softmax_cdf(lg, T, tmp, nsym, 65536.0f);
    lic360_cdf_fixup(T, nsym, 1);
}
// The same table for the compile-time alphabet of every LIC360 importance net: static indices, registers (gmm_tables.h)
template <int NSYM>
__device__ __forceinline__ void imp_table_t(const float *__restrict__ y, long base, long cstride, float (&T)[NSYM + 1]) {
    float lg[NSYM];
#pragma unroll
    for (int i = 0; i < NSYM; ++i) lg[i] = y[base + i * cstride];
    softmax_table_static<NSYM>(lg, 65536.0f, T);
}
#define IMP_NSYM_FAST 49                                              // the alphabet of every LIC360 importance net (model_zoo.py)
template <bool FAST>
__global__ void k_imp_enc_tables(const float *__restrict__ y, const float *__restrict__ lv, const int *__restrict__ pidx, uint2 *__restrict__ rec,
                                 int B, int H, int W, int nsym) {
    const long HW = (long)H * W, total = HW * B;
    GRID_STRIDE(i, total) {
        const int tw = (int)(i % W), th = (int)((i / W) % H), b = (int)(i / HW), s = th + tw;
        int sym = (int)lv[i];
        sym = sym < 0 ? 0 : (sym > nsym - 1 ? nsym - 1 : sym);
        unsigned lo = 0u, hi = 0u;
        if constexpr (FAST) {
            float T[IMP_NSYM_FAST + 1];
            imp_table_t<IMP_NSYM_FAST>(y, (long)b * nsym * HW + (long)th * W + tw, HW, T);
#pragma unroll
            for (int k = 0; k < IMP_NSYM_FAST; ++k) if (k == sym) { lo = (unsigned)(int)T[k]; hi = (unsigned)(int)T[k + 1]; }
        } else {
            float T[65];
            imp_table(y, (long)b * nsym * HW + (long)th * W + tw, HW, nsym, T);
            lo = (unsigned)(int)T[sym]; hi = (unsigned)(int)T[sym + 1];
        }
        rec[(long)b * HW + pidx[s] + (th - (s >= W ? s - W + 1 : 0))] = make_uint2(lo, hi);
    }
}
// decode activations are diagonal-major [n][c][H+W-1][H] (cell (th, tw) at (th+tw)*H + th): the 16 plane positions a
// conv wave gathers are contiguous
// one struct by value instead of 13 scalar kernel arguments: the number of arguments of a per-plane launch costs like their bytes do (DESIGN.md 4.1 b'')
struct ImpDecTablesArgs {
    const float *y;
    const int *idx;
    int start;
    int len;
    int *tab;
    int tab_pitch;
    int H;
    int W;
    int nsym;
    int sk_rows;
    int sk_pitch;
    int sk_row0;
    int sk_col0;
};
template <bool FAST>
__global__ __launch_bounds__(64) void k_imp_dec_tables(const ImpDecTablesArgs a) {
    const float *__restrict__ y = a.y;
    const int *__restrict__ idx = a.idx;
    const int start = a.start;
    const int len = a.len;
    int *__restrict__ tab = a.tab;
    const int tab_pitch = a.tab_pitch;
    const int H = a.H;
    const int W = a.W;
    const int nsym = a.nsym;
    const int sk_rows = a.sk_rows;
    const int sk_pitch = a.sk_pitch;
    const int sk_row0 = a.sk_row0;
    const int sk_col0 = a.sk_col0;
    const int b = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    if (i >= len) return;
    const long HW = (long)H * W, SK = (long)sk_rows * sk_pitch;
    const int th = idx[start + i], tw = idx[start + i + HW];
    int *row = tab + ((long)b * tab_pitch + i) * IMP_TW;
    if constexpr (FAST) {
        float T[IMP_NSYM_FAST + 1];
        imp_table_t<IMP_NSYM_FAST>(y, (long)b * nsym * SK + (long)(th + tw + sk_row0) * sk_pitch + th + sk_col0, SK, T);
#pragma unroll
        for (int k = 0; k <= IMP_NSYM_FAST; ++k) row[k] = (int)T[k];
    } else {
        float T[65];
        imp_table(y, (long)b * nsym * SK + (long)(th + tw + sk_row0) * sk_pitch + th + sk_col0, SK, nsym, T);
        for (int k = 0; k <= nsym; ++k) row[k] = (int)T[k];
    }
}
// one wave per image: lane k holds T[k] of the current symbol; the symbol is the number of inner entries <= target
// one struct by value instead of 19 scalar kernel arguments: the number of arguments of a per-plane launch costs like their bytes do (DESIGN.md 4.1 b'')
struct ImpDecPlaneArgs {
    const int *tab;
    int tab_pitch;
    const int *idx;
    int start;
    int len;
    AcDevState *state;
    const uint8_t *bytes;
    long cap;
    const int *nbytes;
    float *x0;
    float *out;
    int H;
    int W;
    int nsym;
    float sc;
    int sk_rows = 0;
    int sk_pitch = 0;
    int sk_row0 = 0;
    int sk_col0 = 0;
};
template <bool LINEAR>
__global__ __launch_bounds__(64) void k_imp_dec_plane(const ImpDecPlaneArgs a) {
    const int *__restrict__ tab = a.tab;
    const int tab_pitch = a.tab_pitch;
    const int *__restrict__ idx = a.idx;
    const int start = a.start;
    const int len = a.len;
    AcDevState *__restrict__ state = a.state;
    const uint8_t *__restrict__ bytes = a.bytes;
    const long cap = a.cap;
    const int *__restrict__ nbytes = a.nbytes;
    float *__restrict__ x0 = a.x0;
    float *__restrict__ out = a.out;
    const int H = a.H;
    const int W = a.W;
    const int nsym = a.nsym;
    const float sc = a.sc;
    const int sk_rows = a.sk_rows;
    const int sk_pitch = a.sk_pitch;
    const int sk_row0 = a.sk_row0;
    const int sk_col0 = a.sk_col0;
    const int b = blockIdx.x, lane = threadIdx.x;
    const long HW = (long)H * W;
    AcDevState ds = state[b];
    AcState st;
    st.low = ds.low; st.high = ds.high; st.code = ds.code; st.underflow = 0; st.error = 0;
    DevBits rd;
    rd.buf = bytes + (long)b * cap; rd.len = dev_stream_len(nbytes[b], cap); rd.pos = ds.pos; rd.acc = ds.acc; rd.nacc = ds.nacc; rd.lane = lane;
    rd.fetch_window();
    const int *rows = tab + (long)b * tab_pitch * IMP_TW;
    int tnext = len > 0 ? rows[lane] : 0;
    for (int j = 0; j < len; ++j) {
        const int tl = tnext;
        if (j + 1 < len) tnext = rows[(long)(j + 1) * IMP_TW + lane];               // next row is in flight while this symbol decodes
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(tl, nsym);
        const uint32_t target = ac_decode_target(st, total);
        const int sym = __popcll(__ballot(lane >= 1 && lane < nsym && target >= (uint32_t)tl));
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(tl, sym), hi = (uint32_t)__builtin_amdgcn_readlane(tl, sym + 1);
        ac_decode_consume_from(st, rd, lo, hi, total);
        if (lane == 0) {
            if constexpr (LINEAR) out[(long)b * HW + start + j] = (float)sym;
            else {
                const int th = idx[start + j], tw = idx[start + j + HW];
                x0[(long)b * sk_rows * sk_pitch + (long)(th + tw + sk_row0) * sk_pitch + th + sk_col0] = lic360_affine((float)sym, sc, -1.0f);   // TileInput(1, -1, 2/47, 1): lic360_demo.py:264
                out[(long)b * HW + (long)th * W + tw] = (float)sym;
            }
        }
    }
    if (lane == 0) {
        ds.low = st.low; ds.high = st.high; ds.code = st.code; ds.error |= st.error;     // sticky: coder faults 1..3, clamp flag 32
        ds.pos = rd.pos; ds.acc = rd.acc; ds.nacc = rd.nacc;
        state[b] = ds;
    }
}

LIC360_API int lic360_impcodec_create(int h, int w, int hidden_channels, int nsym, int max_batch, lic360_impcodec **out) {
    ARG_CHECK(out && h > 0 && w > 0 && h < 4096 && w < 4096 && hidden_channels > 0 && nsym >= 3 && nsym < IMP_TW && max_batch > 0);
    lic360_impcodec *c = new lic360_impcodec();
    memset(c->layer_set, 0, sizeof(c->layer_set));
    c->H = h; c->W = w; c->HW = h * w; c->P = h + w - 1; c->cpg = hidden_channels; c->nsym = nsym; c->maxB = max_batch;
    c->sc = 2.0f / (float)(nsym - 2);
    for (int i = 0; i < 12; ++i) c->packed[i] = c->bias[i] = c->act[i] = c->packed144[i] = nullptr;
    int rc = 0;
    rc |= lic360_conv_plan_create(1, 1, hidden_channels, 5, 5, &c->plan[0]);
    rc |= lic360_conv_plan_create(hidden_channels, 1, hidden_channels, 5, 6, &c->plan[1]);
    rc |= lic360_conv_plan_create(hidden_channels, 1, nsym, 5, 6, &c->plan[2]);
    if (rc) return 1;
    c->h_idx.resize(2 * (size_t)c->HW);
    c->h_pidx.resize(h + w);
    lic360_code_contex(h, w, c->h_idx.data(), c->h_pidx.data());
    c->tab_pitch = ((h < w ? h : w) + 63) / 64 * 64;
    const size_t B = max_batch, HW = c->HW, C = hidden_channels, CE = hidden_channels > nsym ? hidden_channels : nsym;   // the last layer writes nsym planes
    rc |= dmalloc(&c->d_idx, c->h_idx.size());
    rc |= dmalloc(&c->d_pidx, c->h_pidx.size());
    rc |= dmalloc(&c->e_x0, B * HW);
    for (int i = 0; i < 3; ++i) rc |= dmalloc(&c->e_buf[i], B * CE * HW);
    rc |= dmalloc(&c->e_rec, B * HW);
    // 144-channel layers on the leaf-resident 16x16x4 kernels (other widths keep the generic kernels)
    c->use144 = lic360_conv144_supported(c->plan[1]) && lic360_conv144_supported(c->plan[2]);
    if (c->use144) {
        if (lic360_ec144_layout(h, w, &c->e_hp, &c->e_wp) || lic360_dc144_layout(h, w, &c->sk_rows, &c->sk_pitch)) return 1;
        c->sk_row0 = 4; c->sk_col0 = 2;
        for (int i = 0; i < 3; ++i) rc |= dmalloc(&c->e_pad[i], B * C * (size_t)c->e_hp * c->e_wp);
        rc |= dmalloc(&c->e_plain, B * CE * HW);
        if (rc) return 1;
        for (int i = 0; i < 3; ++i) HIP_TRY(hipMemset(c->e_pad[i], 0, B * C * (size_t)c->e_hp * c->e_wp * 4));   // the halo stays zero
    } else { c->sk_rows = h + w - 1; c->sk_pitch = h; c->sk_row0 = 0; c->sk_col0 = 0; }
    const size_t SK = (size_t)c->sk_rows * c->sk_pitch;                   // diagonal-major decode planes
    rc |= dmalloc(&c->d_x0, B * SK);
    for (int i = 0; i < 11; ++i) rc |= dmalloc(&c->d_act[i], B * C * SK);
    rc |= dmalloc(&c->d_y, B * (size_t)nsym * SK);
    rc |= dmalloc(&c->d_tab, B * (size_t)c->tab_pitch * IMP_TW);
    rc |= dmalloc(&c->d_state, B);
    if (rc) return 1;
    HIP_TRY(hipMemcpy(c->d_idx, c->h_idx.data(), c->h_idx.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_pidx, c->h_pidx.data(), c->h_pidx.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(c->d_x0, 0, B * SK * 4));
    for (int i = 0; i < 11; ++i) HIP_TRY(hipMemset(c->d_act[i], 0, B * C * SK * 4));
    HIP_TRY(hipMemset(c->d_y, 0, B * (size_t)nsym * SK * 4));
    *out = c;
    return 0;
}
LIC360_API void lic360_impcodec_destroy(lic360_impcodec *c) {
    if (!c) return;
    for (int i = 0; i < 3; ++i) lic360_conv_plan_destroy(c->plan[i]);
    for (int i = 0; i < 12; ++i) { (void)hipFree(c->packed[i]); (void)hipFree(c->bias[i]); (void)hipFree(c->act[i]); (void)hipFree(c->packed144[i]); }
    for (int i = 0; i < 3; ++i) (void)hipFree(c->e_pad[i]);
    (void)hipFree(c->e_plain);
    (void)hipFree(c->d_idx); (void)hipFree(c->d_pidx); (void)hipFree(c->e_x0);
    for (int i = 0; i < 3; ++i) (void)hipFree(c->e_buf[i]);
    (void)hipFree(c->e_rec); (void)hipFree(c->d_x0);
    for (int i = 0; i < 11; ++i) (void)hipFree(c->d_act[i]);
    (void)hipFree(c->d_y); (void)hipFree(c->d_tab); (void)hipFree(c->d_state);
    for (hipEvent_t e : c->plane_ev) (void)hipEventDestroy(e);
    delete c;
}
LIC360_API int lic360_impcodec_set_layer(void *stream, lic360_impcodec *c, int layer, const float *weight, const float *bias, const float *act) {
    ARG_CHECK(c && layer >= 0 && layer < 12 && weight && bias);
    ARG_CHECK((act != nullptr) == (layer != 11));
    lic360_conv_plan *p = c->plan[plan_of(layer)];
    if (!c->packed[layer]) {
        if (dmalloc(&c->packed[layer], (size_t)lic360_conv_plan_packed_floats(p))) return 1;
        if (dmalloc(&c->bias[layer], (size_t)p->nout)) return 1;
        if (act && dmalloc(&c->act[layer], (size_t)p->nout)) return 1;
    }
    if (lic360_conv_pack(stream, p, weight, 1, c->packed[layer])) return 1;
    if (c->use144 && layer > 0) {
        if (!c->packed144[layer] && dmalloc(&c->packed144[layer], (size_t)lic360_conv144_packed_floats(p))) return 1;
        if (lic360_conv144_pack(stream, p, weight, c->packed144[layer])) return 1;
    }
    HIP_TRY(hipMemcpyAsync(c->bias[layer], bias, (size_t)p->nout * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (act) HIP_TRY(hipMemcpyAsync(c->act[layer], act, (size_t)p->nout * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    c->layer_set[layer] = true;
    return 0;
}
static int imp_ready(const lic360_impcodec *c, int B) {
    ARG_CHECK(c && B > 0 && B <= c->maxB);
    for (int i = 0; i < 12; ++i)
        if (!c->layer_set[i]) { lic360_set_error("importance codec layer %d has no weights (call lic360_impcodec_set_layer)", i); return 2; }
    return 0;
}
LIC360_API int lic360_impcodec_encode(void *stream, lic360_impcodec *c, const float *levels, int B, uint8_t *bytes, long cap, int *nbytes, int *err) {
    if (imp_ready(c, B)) return 2;
    ARG_CHECK(levels && bytes && nbytes && err && cap > 0 && cap < (1L << 31) && cap % 4 == 0 && ((uintptr_t)bytes & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    const int H = c->H, W = c->W;
    const long total = (long)B * c->HW;
    hipLaunchKernelGGL(k_imp_prep, dim3(lic360_blocks(total, 4)), dim3(256), 0, s, levels, c->e_x0, total, c->sc);
    LAUNCH_CHECK();
    if (c->use144) {
        // layer 0 (1 -> 144) on the generic kernel into plain NCHW, copied into the zero-haloed planes; layers 1..10 haloed -> haloed;
        // layer 11 (144 -> nsym) haloed -> plain NCHW for the table kernel
        const long PLh = (long)c->e_hp * c->e_wp;
        if (lic360_cconv_ec_ex(stream, c->plan[0], c->e_x0, c->packed[0], c->bias[0], c->act[0], nullptr, c->e_plain, B, H, W, 1, B)) return 1;
        hipLaunchKernelGGL(k_imp_halo, dim3(lic360_blocks((long)B * c->cpg * c->HW, 4)), dim3(256), 0, s, c->e_plain, c->e_pad[0], (long)B * c->cpg * c->HW, H, W, c->e_hp, c->e_wp);
        LAUNCH_CHECK();
        float *cur = c->e_pad[0], *t1 = c->e_pad[1], *nxt = c->e_pad[2];
        auto ec = [&](int layer, const float *xin, const float *res, float *dst) -> int {
            return lic360_cconv144_ec(stream, c->plan[1], xin, c->packed144[layer], c->bias[layer], c->act[layer], res, dst, B, H, W, PLh, c->e_wp, 2);
        };
        for (int blk = 0; blk < 5; ++blk) {
            if (ec(1 + 2 * blk, cur, nullptr, t1)) return 1;
            if (ec(2 + 2 * blk, t1, cur, nxt)) return 1;
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        if (lic360_cconv144_ec(stream, c->plan[2], cur, c->packed144[11], c->bias[11], nullptr, nullptr, c->e_plain, B, H, W, (long)c->HW, W, 0)) return 1;
        if (c->nsym == IMP_NSYM_FAST) hipLaunchKernelGGL(k_imp_enc_tables<true>, dim3(lic360_blocks(total, 1)), dim3(64), 0, s, c->e_plain, levels, c->d_pidx, c->e_rec, B, H, W, c->nsym);
        else hipLaunchKernelGGL(k_imp_enc_tables<false>, dim3(lic360_blocks(total, 1)), dim3(64), 0, s, c->e_plain, levels, c->d_pidx, c->e_rec, B, H, W, c->nsym);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_ac_encode, dim3(B), dim3(128), 0, s, c->e_rec, (long)c->HW, bytes, cap, nbytes, err);
        LAUNCH_CHECK();
        return 0;
    }
    float *cur = c->e_buf[0], *t1 = c->e_buf[1], *nxt = c->e_buf[2];
    auto ec = [&](int layer, const float *xin, const float *res, float *dst) -> int {
        return lic360_cconv_ec_ex(stream, c->plan[plan_of(layer)], xin, c->packed[layer], c->bias[layer], c->act[layer], res, dst, B, H, W, 1, B);
    };
    if (ec(0, c->e_x0, nullptr, cur)) return 1;
    for (int blk = 0; blk < 5; ++blk) {
        if (ec(1 + 2 * blk, cur, nullptr, t1)) return 1;
        if (ec(2 + 2 * blk, t1, cur, nxt)) return 1;
        float *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (ec(11, cur, nullptr, t1)) return 1;                                         // [B, nsym, H, W] (uses the first nsym planes of the buffer)
    if (c->nsym == IMP_NSYM_FAST) hipLaunchKernelGGL(k_imp_enc_tables<true>, dim3(lic360_blocks(total, 1)), dim3(64), 0, s, t1, levels, c->d_pidx, c->e_rec, B, H, W, c->nsym);
    else hipLaunchKernelGGL(k_imp_enc_tables<false>, dim3(lic360_blocks(total, 1)), dim3(64), 0, s, t1, levels, c->d_pidx, c->e_rec, B, H, W, c->nsym);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ac_encode, dim3(B), dim3(128), 0, s, c->e_rec, (long)c->HW, bytes, cap, nbytes, err);
    LAUNCH_CHECK();
    return 0;
}
// mask_out cells of the map positions idx[start .. start+len): Dtow(stride)(Imp2mask(levels)) restricted to them --
//   tmask(tc, th, tw) = tc < int(level + 1e-5) * cpn                                 (extension/imp2mask_cuda.cu:31)
//   out(pc, ph, pw)   = tmask(pc s^2 + (ph % s) s + pw % s, ph / s, pw / s)          (extension/dtow_cuda.cu:38-56)
// one struct by value instead of 11 scalar kernel arguments: the number of arguments of a per-plane launch costs like their bytes do (DESIGN.md 4.1 b'')
struct ImpMaskPlaneArgs {
    const float *levels;
    const int *idx;
    int start;
    int len;
    float *out;
    int B;
    int H;
    int W;
    int C;
    int s;
    int cpn;
};
__global__ void k_imp_mask_plane(const ImpMaskPlaneArgs a) {
    const float *__restrict__ levels = a.levels;
    const int *__restrict__ idx = a.idx;
    const int start = a.start;
    const int len = a.len;
    float *__restrict__ out = a.out;
    const int B = a.B;
    const int H = a.H;
    const int W = a.W;
    const int C = a.C;
    const int s = a.s;
    const int cpn = a.cpn;
    const long HW = (long)H * W, total = (long)B * len * C;
    const int Co = C / (s * s), Ho = H * s, Wo = W * s;
    // one thread per element, 256 threads per workgroup (launched once per plane: no gridDim / blockDim, so no implicit kernel arguments)
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < total) {
        const int tc = (int)(e % C), i = (int)((e / C) % len), b = (int)(e / C / len);
        const int th = idx[start + i], tw = idx[start + i + HW];
        const int imp = (int)((double)levels[(long)b * HW + (long)th * W + tw] + 1e-5) * cpn;
        const int pc = tc / (s * s), r = tc % (s * s), ph = th * s + r / s, pw = tw * s + r % s;
        out[(((long)b * Co + pc) * Ho + ph) * Wo + pw] = tc < imp ? 1.0f : 0.0f;
    }
}
static int impcodec_decode_impl(void *stream, lic360_impcodec *c, const uint8_t *bytes, long cap, const int *nbytes, int B,
                                float *levels_out, int *err, float *mask_out, int mask_c, int stride) {
    if (imp_ready(c, B)) return 2;
    ARG_CHECK(bytes && nbytes && levels_out && err && cap > 0 && cap < (1L << 31) && cap % 4 == 0 && ((uintptr_t)bytes & 3) == 0);
    hipStream_t s = (hipStream_t)stream;
    const int H = c->H, W = c->W;
    hipLaunchKernelGGL(k_dec_init, dim3((B + 63) / 64), dim3(64), 0, s, bytes, cap, nbytes, c->d_state, B);
    LAUNCH_CHECK();
    const int *pih = c->h_pidx.data();
    const long SK = (long)c->sk_rows * c->sk_pitch, off0 = (long)c->sk_row0 * c->sk_pitch + c->sk_col0;
    auto dc = [&](int layer, const float *xin, const float *res, float *dst, int p) -> int {
        if (c->use144 && layer > 0)
            return lic360_cconv144_dc_plane(stream, c->plan[plan_of(layer)], xin, c->packed144[layer], c->bias[layer], c->act[layer], res, dst, B, H, W, p);
        // generic kernel on the diagonal-major planes: cell (th, tw) at th * (pitch + 1) + tw * pitch from the layout's origin
        return lic360_cconv_dc_plane_strided(stream, c->plan[plan_of(layer)], xin + off0, c->packed[layer], c->bias[layer], c->act[layer],
                                             res ? res + off0 : nullptr, dst + off0, B, H, W, 1, c->d_idx, c->d_pidx, pih, p, B,
                                             SK, c->sk_pitch + 1, c->sk_pitch, SK, c->sk_pitch + 1, c->sk_pitch);
    };
    for (int p = 0; p < c->P; ++p) {
        if (dc(0, c->d_x0, nullptr, c->d_act[0], p)) return 1;
        for (int blk = 0; blk < 5; ++blk) {
            const int a = 1 + 2 * blk, b2 = 2 + 2 * blk;
            if (dc(a, c->d_act[a - 1], nullptr, c->d_act[a], p)) return 1;
            if (dc(b2, c->d_act[a], c->d_act[a - 1], c->d_act[b2], p)) return 1;
        }
        if (dc(11, c->d_act[10], nullptr, c->d_y, p)) return 1;
        const int start = pih[p], len = pih[p + 1] - pih[p];
        if (len <= 0) continue;
        if (c->nsym == IMP_NSYM_FAST)
            hipLaunchKernelGGL(k_imp_dec_tables<true>, dim3((len + 63) / 64, B), dim3(64), 0, s, ImpDecTablesArgs{c->d_y, c->d_idx, start, len, c->d_tab, c->tab_pitch, H, W, c->nsym,
                               c->sk_rows, c->sk_pitch, c->sk_row0, c->sk_col0});
        else
            hipLaunchKernelGGL(k_imp_dec_tables<false>, dim3((len + 63) / 64, B), dim3(64), 0, s, ImpDecTablesArgs{c->d_y, c->d_idx, start, len, c->d_tab, c->tab_pitch, H, W, c->nsym,
                               c->sk_rows, c->sk_pitch, c->sk_row0, c->sk_col0});
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_imp_dec_plane<false>, dim3(B), dim3(64), 0, s, ImpDecPlaneArgs{c->d_tab, c->tab_pitch, c->d_idx, start, len, c->d_state, bytes, cap, nbytes,
                           c->d_x0, levels_out, H, W, c->nsym, c->sc, c->sk_rows, c->sk_pitch, c->sk_row0, c->sk_col0});
        LAUNCH_CHECK();
        if (mask_out) {
            // the latent mask of the map cells decoded in this plane (cells of later planes hold whatever the buffer held: nobody
            // reads their mask before their plane's event)
            const long tot = (long)B * len * mask_c;
            hipLaunchKernelGGL(k_imp_mask_plane, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ImpMaskPlaneArgs{levels_out, c->d_idx, start, len, mask_out, B, H, W,
                               mask_c, stride, mask_c / (c->nsym - 1)});
            LAUNCH_CHECK();
            HIP_TRY(hipEventRecord(c->plane_ev[p], s));
        }
    }
    hipLaunchKernelGGL(k_collect_err, dim3((B + 63) / 64), dim3(64), 0, s, c->d_state, err, B);
    LAUNCH_CHECK();
    return 0;
}
LIC360_API int lic360_impcodec_decode(void *stream, lic360_impcodec *c, const uint8_t *bytes, long cap, const int *nbytes, int B,
                                      float *levels_out, int *err) {
    return impcodec_decode_impl(stream, c, bytes, cap, nbytes, B, levels_out, err, nullptr, 0, 0);
}
// Decode + the latent codec's mask, plane by plane: after plane p of the maps, the cells of mask_out = Dtow(stride)(Imp2mask(levels_out))
// (ImpEntDecoder.forward's last three lines, test/lic360_demo.py:283-287) that plane p decides are written and event p is recorded, so
// that a latent decode on ANOTHER stream can run behind the map's decode instead of after it (lic360_codec_decode_gated): the latent
// plane q only reads the mask of map cells (h >> 1, w >> 1) with h + w <= q, i.e. of map planes <= q / stride.
// mask_out: [B][mask_c / stride^2][stride h][stride w]; *generation_out identifies this masked decode to lic360_codec_decode_gated.
// Caller's contract (the events are re-recorded by every call): enqueue this BEFORE the gated latent decode of the same step, and order
// the NEXT step's masked decode into the same mask_out after the latent decode that still reads it (an event, or two mask buffers).
LIC360_API int lic360_impcodec_decode_masked(void *stream, lic360_impcodec *c, const uint8_t *bytes, long cap, const int *nbytes, int B,
                                             float *levels_out, int *err, float *mask_out, int mask_c, int stride, long *generation_out) {
    ARG_CHECK(c && mask_out && generation_out && mask_c > 0 && stride > 0 && mask_c % (stride * stride) == 0 && c->nsym > 1 &&
              mask_c % (c->nsym - 1) == 0 && B > 0 && B <= c->maxB);
    while ((int)c->plane_ev.size() < c->P) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));   // (same-device consumers only: no system-scope cache writeback per plane)
        c->plane_ev.push_back(e);
    }
    const int rc = impcodec_decode_impl(stream, c, bytes, cap, nbytes, B, levels_out, err, mask_out, mask_c, stride);
    c->gate_armed = rc == 0;                                            // (a failed enqueue arms nothing)
    if (rc) return rc;
    c->gate_stride = stride;
    c->gate_B = B;
    c->gate_mask = mask_out;
    *generation_out = ++c->gate_gen;
    return 0;
}

// The latent decode behind an importance-map decode that runs on another stream and fills `mask` plane by plane
// (lic360_impcodec_decode_masked on `map_codec`): the convolutions of a latent plane need no mask at all, its table kernel waits for
// the map's event min(P - 1, plane / stride) before it reads the mask.  Results are those of lic360_codec_decode on the finished mask.
// `generation` is what that masked decode returned: it must be the map codec's LATEST masked decode, not yet consumed, into this `mask`.
LIC360_API int lic360_codec_decode_gated(void *stream, lic360_codec *c, const uint8_t *bytes, long cap, const int *nbytes,
                                         const float *mask, int B, float *code_out, int *err, lic360_impcodec *map_codec, long generation) {
    ARG_CHECK(map_codec && generation > 0);
    if (!map_codec->gate_armed || map_codec->gate_gen != generation) {
        lic360_set_error("lic360_codec_decode_gated: stale gate (generation %ld, the map codec is at %ld, %s): enqueue "
                         "lic360_impcodec_decode_masked of THIS step first, and use each gate once",
                         generation, map_codec->gate_gen, map_codec->gate_armed ? "armed" : "already consumed");
        return 2;
    }
    if (map_codec->gate_mask != mask) {
        lic360_set_error("lic360_codec_decode_gated: `mask` is not the buffer the masked decode of generation %ld fills", generation);
        return 2;
    }
    if (B > map_codec->gate_B) {
        lic360_set_error("lic360_codec_decode_gated: batch %d, but the masked decode of generation %ld fills the mask of %d images", B, generation, map_codec->gate_B);
        return 2;
    }
    // a call that is rejected for its own arguments must not burn the ticket: the checks codec_decode_impl starts with, first
    if (check_ready(c, B)) return 2;
    ARG_CHECK(bytes && nbytes && mask && code_out && err && cap > 0 && cap < (1L << 31) && cap % 4 == 0 && ((uintptr_t)bytes & 3) == 0);
    map_codec->gate_armed = false;
    return codec_decode_impl(stream, c, bytes, cap, nbytes, mask, B, code_out, err, (void *const *)map_codec->plane_ev.data(), map_codec->P,
                             map_codec->gate_stride);
}

// ------------------------------------------------------------------------------------------------ device-coder test hooks
// Raw tables + symbols through the DEVICE coder kernels (k_ac_encode, k_dec_init, k_dec_plane, k_imp_dec_plane), so that the
// fixtures generated by the reference coder (tests/golden/ac_golden.npz) exercise the v_readlane windows, the fp32-divide
// target and the chunked state hand-over directly.  Every table must total 65536 (what this codec's tables do).
__global__ void k_test_records(const int *__restrict__ tables, int ncode, const int *__restrict__ labels, const float *__restrict__ mask,
                               long n, uint2 *__restrict__ rec) {
    GRID_STRIDE(i, n) {
        uint2 r = make_uint2(0u, 0u);
        if (!mask || !(mask[i] < 0.5f)) {
            const int *T = tables + i * (ncode + 1);
            int sym = labels[i];
            sym = sym < 0 ? 0 : (sym > ncode - 1 ? ncode - 1 : sym);
            r = make_uint2((unsigned)T[sym], (unsigned)T[sym + 1]);
        }
        rec[i] = r;
    }
}
__global__ void k_test_tab8(const int *__restrict__ tables, const float *__restrict__ mask, long start, int len, uint4 *__restrict__ tab, AcDevState *st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const int *T = tables + (start + i) * 9;
    const bool coded = !mask || !(mask[start + i] < 0.5f);
    if (coded && ((T[1] | T[2] | T[3] | T[4] | T[5] | T[6] | T[7]) & ~0xffff)) atomicOr(&st->error, 64);   // inner entries must fit 16 bits
    // dec_pack8's words carry "coded" as T[7] << 16 != 0 (k_dec_plane skips a symbol whose word 7 is zero): a coded symbol with all mass
    // on symbol 7 (T[1..7] == 0) would be skipped and the stream would desynchronise silently.  The codec's own tables are strictly
    // increasing (lic360_cdf_fixup), this hook's callers must be told.
    if (coded && T[7] == 0) atomicOr(&st->error, 64);
    uint4 r = make_uint4(0u, 0u, 0u, 0u), r2 = r;
    if (coded) dec_pack8(T, r, r2);
    tab[2 * i] = r;
    tab[2 * i + 1] = r2;
}
__global__ void k_test_tabn(const int *__restrict__ tables, int ncode, long start, int len, int *__restrict__ tab) {
    const int i = blockIdx.x, k = threadIdx.x;
    if (i < len && k <= ncode) tab[(long)i * IMP_TW + k] = tables[(start + i) * (ncode + 1) + k];
}

LIC360_API int lic360_devcoder_encode(void *stream, const int *tables, int ncode, const int *labels, const float *mask, long n,
                                      uint8_t *bytes, long cap, int *nbytes, int *err) {
    ARG_CHECK(ncode >= 1 && n >= 0 && bytes && nbytes && err && cap > 0 && cap < (1L << 31) && cap % 4 == 0 && ((uintptr_t)bytes & 3) == 0 &&
              (n == 0 || (tables && labels)));
    hipStream_t s = (hipStream_t)stream;
    uint2 *rec = nullptr;
    if (dmalloc(&rec, (size_t)n)) return 1;
    if (n) {
        hipLaunchKernelGGL(k_test_records, dim3(lic360_blocks(n, 1)), dim3(256), 0, s, tables, ncode, labels, mask, n, rec);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_ac_encode, dim3(1), dim3(128), 0, s, rec, n, bytes, cap, nbytes, err);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(s));
    (void)hipFree(rec);
    return 0;
}

// decodes n symbols in consecutive chunks of `chunk` symbols -- the role planes play in the codec: coder state and bit
// position are handed from launch to launch through AcDevState.  ncode == 8 runs k_dec_plane (8-symbol GMM tables as two
// uint4 + coded flag), any other alphabet (< 64) k_imp_dec_plane (one table entry per lane).  out[i] = symbol, 0 where masked.
LIC360_API int lic360_devcoder_decode(void *stream, const int *tables, int ncode, const float *mask, long n, int chunk,
                                      const uint8_t *bytes, long cap, const int *nbytes, float *out, int *err) {
    ARG_CHECK(ncode >= 1 && ncode < IMP_TW && n >= 0 && chunk > 0 && bytes && nbytes && out && err && cap > 0 && cap < (1L << 31) && cap % 4 == 0 &&
              ((uintptr_t)bytes & 3) == 0 && (n == 0 || tables) && (ncode == 8 || !mask));
    hipStream_t s = (hipStream_t)stream;
    AcDevState *st = nullptr;
    uint4 *tab8 = nullptr;
    int *tabn = nullptr;
    if (dmalloc(&st, 1) || dmalloc(&tab8, 2 * (size_t)chunk) || dmalloc(&tabn, (size_t)chunk * IMP_TW)) return 1;
    hipLaunchKernelGGL(k_dec_init, dim3(1), dim3(64), 0, s, bytes, cap, nbytes, st, 1);
    LAUNCH_CHECK();
    for (long start = 0; start < n; start += chunk) {
        const int len = (int)std::min<long>(chunk, n - start);
        if (ncode == 8) {
            hipLaunchKernelGGL(k_test_tab8, dim3((len + 63) / 64), dim3(64), 0, s, tables, mask, start, len, tab8, st);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(k_dec_plane<true>, dim3(1), dim3(64), 0, s, DecPlaneArgs{tab8, chunk, (const int *)nullptr, (int)start, len, 0, st, bytes, cap, nbytes,
                               (float *)nullptr, out, 0, 1, 1, 0, 0, 0, 0});
        } else {
            hipLaunchKernelGGL(k_test_tabn, dim3(len), dim3(64), 0, s, tables, ncode, start, len, tabn);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(k_imp_dec_plane<true>, dim3(1), dim3(64), 0, s, ImpDecPlaneArgs{tabn, chunk, (const int *)nullptr, (int)start, len, st, bytes, cap, nbytes,
                               (float *)nullptr, out, 0, 0, ncode, 0.0f});
        }
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_collect_err, dim3(1), dim3(64), 0, s, st, err, 1);
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(s));
    (void)hipFree(st); (void)hipFree(tab8); (void)hipFree(tabn);
    return 0;
}

// ------------------------------------------------------------------------------------------------ timing hooks
LIC360_API int lic360_codec_profile_enable(lic360_codec *c, int on) {
    ARG_CHECK(c);
    c->prof = on != 0;
    for (int k = 0; k < PROF_NCLS; ++k) c->n_ev[k] = 0;
    return 0;
}
// comma-separated names of the kernel classes, in the order lic360_codec_profile_read fills its arrays
LIC360_API const char *lic360_codec_profile_classes(void) { return PROF_NAMES; }

// Per kernel class: sum of HIP-event elapsed times (ms) and number of launches recorded since the last call; the events
// were recorded on the stream the kernels ran on.  ms / launches must hold `n` entries (n >= number of classes, else error).
LIC360_API int lic360_codec_profile_read(lic360_codec *c, int n, double *ms, long *launches) {
    ARG_CHECK(c && ms && launches && n >= PROF_NCLS);
    for (int k = 0; k < PROF_NCLS; ++k) {
        double acc = 0;
        const size_t cnt = c->n_ev[k];
        for (size_t i = 0; i + 1 < cnt; i += 2) {
            HIP_TRY(hipEventSynchronize(c->ev[k][i + 1]));
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, c->ev[k][i], c->ev[k][i + 1]));
            acc += t;
        }
        ms[k] = acc;
        launches[k] = (long)(cnt / 2);
        c->n_ev[k] = 0;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ dead-cone skip: statistics and test hooks
// enable > 0: the following encodes / decodes count what they execute (0: stop, < 0: unchanged); out (host, 2 * 12 * 64 values, may be NULL): [0][layer][group block] live
// (tile, group block) pairs of the encode-order launches -- 64 positions x the block's groups each, per sample --, [1][layer][group] cells the
// decode-order launches stored; reading clears the counters.  skip_active_out (may be NULL): 1 if this codec skips at all (0: generic kernels or LIC360_NOSKIP).
LIC360_API int lic360_codec_skip_stats(lic360_codec *c, int enable, unsigned long long *out, int *skip_active_out) {
    ARG_CHECK(c);
    if (skip_active_out) *skip_active_out = c->skip ? 1 + (c->dcl.list ? 1 : 0) : 0;
    if (!c->skip) { if (out) memset(out, 0, 2 * NEED_LAYERS * NEED_STAT_G * sizeof(unsigned long long)); return 0; }
    if (enable >= 0) c->stats_on = enable != 0;                        // (< 0: leave the switch alone)
    if (out) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, c->stats, 2 * NEED_LAYERS * NEED_STAT_G * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(c->stats, 0, 2 * NEED_LAYERS * NEED_STAT_G * sizeof(unsigned long long)));
    }
    return 0;
}
// Test hook: every interior cell of every activation buffer of the codec (encode-order ping-pong buffers, decode-order layer buffers and y) <- value;
// halo / padding cells stay zero.  A following encode / decode must produce the same bytes / symbols whatever the value: live cells are written
// before they are read, dead cells are only read with zero weights or by dead chains (tests/test_gpu_need.py; finite values only: 0 * NaN is NaN).
__global__ void k_fill_enc(float *buf, long planes, int H, int W, int hp, int wp, int off, float v) {
    GRID_STRIDE(i, planes * H * W) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        buf[(i / ((long)H * W)) * hp * wp + e_cell(y, x, wp, off)] = v;
    }
}
__global__ void k_fill_dec(float *buf, long planes, int H, int W, int sk_rows, int sk_pitch, int row0, int col0, float v) {
    GRID_STRIDE(i, planes * H * W) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        buf[(i / ((long)H * W)) * sk_rows * sk_pitch + (long)(y + x + row0) * sk_pitch + y + col0] = v;
    }
}
LIC360_API int lic360_codec_debug_fill(void *stream, lic360_codec *c, float value) {
    ARG_CHECK(c && value == value && value - value == 0.0f);
    hipStream_t s = (hipStream_t)stream;
    const long B = c->maxB, G = c->G;
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(k_fill_enc, dim3(lic360_blocks(3 * B * 4 * G * c->HW, 4)), dim3(256), 0, s, c->e_buf[i], 3 * B * 4 * G, c->H, c->W, c->e_hp, c->e_wp, c->e_off, value);
        LAUNCH_CHECK();
    }
    for (int i = 0; i < 11; ++i) {
        hipLaunchKernelGGL(k_fill_dec, dim3(lic360_blocks(3 * B * 4 * G * c->HW, 4)), dim3(256), 0, s, c->d_act[i], 3 * B * 4 * G, c->H, c->W, c->sk_rows, c->sk_pitch, c->sk_row0, c->sk_col0, value);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_fill_dec, dim3(lic360_blocks(3 * B * 3 * G * c->HW, 4)), dim3(256), 0, s, c->d_y, 3 * B * 3 * G, c->H, c->W, c->sk_rows, c->sk_pitch, c->sk_row0, c->sk_col0, value);
    LAUNCH_CHECK();
    return 0;
}
// Test hook: what the last encode / decode of the codec scheduled, copied to the host.  which 0: need maps [maxB][12][H][W] int8; 1: encode list counts
// [12][8] int; 2: encode lists [12][8][cap] int; 3: decode list counts [12][P][8] int; 4: decode records [12][P][8][cap] x 4 unsigned.  *cap_out: cap.
LIC360_API int lic360_codec_debug_lists(lic360_codec *c, int which, void *host_out, long bytes, int *cap_out) {
    ARG_CHECK(c && host_out && bytes >= 0 && which >= 0 && which <= 4 && c->skip);
    HIP_TRY(hipDeviceSynchronize());
    const void *src = nullptr;
    long have = 0;
    if (which == 0) { src = c->need; have = (long)c->maxB * NEED_LAYERS * c->HW; }
    else if (which == 1) { src = c->ecl.cnt; have = NEED_LAYERS * 8 * 4L; }
    else if (which == 2) { src = c->ecl.list; have = (long)NEED_LAYERS * 8 * c->ecl.cap * 4; }
    else if (which == 3) { src = c->dcl.cnt; have = c->dcl.list ? (long)NEED_LAYERS * c->P * 8 * 4 : 0; }
    else { src = c->dcl.list; have = c->dcl.list ? (long)NEED_LAYERS * c->P * 8 * c->dcl.cap * 16 : 0; }
    if (cap_out) *cap_out = which <= 2 ? c->ecl.cap : c->dcl.cap;
    ARG_CHECK(src && bytes <= have);
    HIP_TRY(hipMemcpy(host_out, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}

// the kernels the fused codecs run per bench kernel class (names as rocprofv3 prints them, templates included): bench.py attaches a committed PMC entry
// to a row only if every kernel that entry names is listed here for the row's class (tests/test_abi.py checks each base name against the library's symbols)
LIC360_API const char *lic360_codec_kernel_names(void) {
    return "ec_first=k_cconv16<1, false>;ec_hidden=k_cconv16s+k_cconv16<4, false>;ec_last=k_cconv16<4, true>;"
           "dc_first=k_cconv4v6<1, false, false>+k_cconv4v6t<1>;"
           "dc_hidden=k_cconv4v6l<4>+k_cconv4v6t<4>+k_cconv4v6<4, false, false>+k_cconv4v6<4, false, true>;"
           "dc_last=k_cconv4v6l<4>+k_cconv4v6t<4>+k_cconv4v6<4, false, false>+k_cconv4v6<4, false, true>;"
           "imp_ec=k_cconv144<1, false, 2>;imp_dc=k_cconv144<1, true, 2>";
}
