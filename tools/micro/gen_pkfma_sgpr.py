#!/usr/bin/env python3
"""Generator of tools/micro/pkfma_sgpr.hip (VERDICT r4, next #1 d): is a decode-order masked convolution whose WEIGHTS are scalar
operands feasible on gfx950?  In decode order the weights of a wave (one net, one group, one lane class) are wave-uniform, so the 100
fmaf per position, input channel and class could be 50 `v_pk_fma_f32 acc[2], x (op_sel broadcast), s[w:w+1]` — full fp32 vector rate
(157 TF, the same as the f32 MFMA peak), no MFMA row padding.  The open question is the weight stream: 400 bytes of scalar loads per
wave and step, SMEM returns out of order (only `lgkmcnt(0)` is a safe wait, and it also drains the LDS reads), 102 SGPRs per wave.

Body of a wave: groups of G weights in a two-half SGPR ring; per group `s_waitcnt lgkmcnt(0)`, the next group's `s_load_dwordx16/x8`,
the LDS operand reads of the next group (9 per 50 packed fmas, as the 4x4x1 kernel's one read per tap diagonal), G/2 `v_pk_fma_f32`.
Variants: G = 32 / 40, 12 or 16 waves per CU, every wave its own weight stream (L2-resident) or all waves of a CU the same one
(scalar-cache hits), no scalar loads at all (the VALU ceiling), no LDS reads.
    python3 tools/micro/gen_pkfma_sgpr.py > tools/micro/pkfma_sgpr.hip
    hipcc --offload-arch=gfx950 -O3 tools/micro/pkfma_sgpr.hip -o /tmp/pkfma_sgpr && /tmp/pkfma_sgpr
"""
import sys

NACC = 100          # accumulators: 25 taps x 4 output channels
XBASE = 100         # x operand ring v[100:119]
RING = 20           # first SGPR of the weight ring


def chunks(G):
    out, o = [], 0
    for c in (16, 8, 4):
        while G - o >= c:
            out.append((o, c))
            o += c
    assert o == G
    return out


def body(G, ngrp, nosl, nolds):
    """one inner-loop iteration: ngrp groups (even) of G weights"""
    L = []
    nrd = -(-9 * (G // 2) // 50)        # LDS reads per group
    fma = 0
    for g in range(ngrp):
        half = RING + (g % 2) * G
        nxt = RING + ((g + 1) % 2) * G
        L.append("s_waitcnt lgkmcnt(0)")
        if not nosl:
            for (o, c) in chunks(G):
                L.append(f"s_load_dwordx{c} s[{nxt + o}:{nxt + o + c - 1}], s[16:17], {((g + 1) * G + o) * 4}")
        if not nolds:
            for r in range(nrd):
                L.append(f"ds_read_b32 v{XBASE + ((g + 1) % 2) * 10 + r}, %[lds] offset:{(g * nrd + r) * 256}")
        for i in range(G // 2):
            a = 2 * (fma % (NACC // 2))
            x = XBASE + (g % 2) * 10 + 2 * (i % max(1, nrd // 2 + (nrd & 1)))
            hi = (i // 5) & 1
            L.append(f"v_pk_fma_f32 v[{a}:{a + 1}], v[{x}:{x + 1}], s[{half + 2 * i}:{half + 2 * i + 1}], v[{a}:{a + 1}] "
                     f"op_sel:[{hi},0,0] op_sel_hi:[{hi},1,1]")
            fma += 1
    return L, fma


def kernel(name, G, ngrp, nosl=False, nolds=False):
    L, fma = body(G, ngrp, nosl, nolds)
    pro = [f"v_mov_b32 v{i}, 0" for i in range(NACC)] + [f"v_mov_b32 v{XBASE + i}, 1.0" for i in range(20)]
    pro += [f"s_mov_b32 s{RING + i}, 1.0" for i in range(2 * G)]
    pro += ["s_mov_b32 s19, %[reps]", "1:", "s_mov_b64 s[16:17], %[b]", "s_mov_b32 s18, %[nit]"]
    if not nosl:
        pro += [f"s_load_dwordx{c} s[{RING + o}:{RING + o + c - 1}], s[16:17], {o * 4}" for (o, c) in chunks(G)]
    pro += ["2:"]
    epi = [f"s_add_u32 s16, s16, {ngrp * G * 4}", "s_addc_u32 s17, s17, 0", "s_sub_u32 s18, s18, 1", "s_cmp_lg_u32 s18, 0",
           "s_cbranch_scc1 2b", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s19, s19, 1", "s_cmp_lg_u32 s19, 0", "s_cbranch_scc1 1b"]
    epi += [f"v_add_f32 v0, v0, v{i}" for i in range(1, NACC)] + ["v_mov_b32 %[res], v0"]
    text = "\\n\"\n        \"".join(pro + L + epi)
    clob = ", ".join([f'"v{i}"' for i in range(XBASE + 20)] + [f'"s{i}"' for i in range(16, RING + 2 * G)] + ['"scc"', '"memory"'])
    return fma, f"""
// {name}: G = {G}, {ngrp} groups per loop iteration, {fma} v_pk_fma_f32 per iteration{', no scalar loads' if nosl else ''}{', no LDS reads' if nolds else ''}
__global__ __launch_bounds__(1024) void {name}(const float *__restrict__ w, float *__restrict__ out, int nit, int reps, int share, long stride) {{
    __shared__ float xs[16 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) xs[i] = 1.0f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *base = w + (share == 0 ? ((long)blockIdx.x * (blockDim.x >> 6) + wave) * stride : share == 3 ? (long)((blockIdx.x & 7) * 4 + (wave & 3)) * stride : (long)(blockIdx.x & 31) * stride);
    const unsigned lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)&xs[threadIdx.x & 63];
    float res;
    asm volatile("{text}\\n"
        : [res] "=v"(res)
        : [b] "s"(base), [lds] "v"(lds), [nit] "s"(nit), [reps] "s"(reps)
        : {clob});
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = res;
}}
"""


VARIANTS = [
    ("k_g32", 32, 6, False, False),
    ("k_g40", 40, 10, False, False),
    ("k_g32_nosl", 32, 6, True, False),
    ("k_g32_nolds", 32, 6, False, True),
    ("k_g40_nolds", 40, 10, False, True),
    ("k_g16", 16, 12, False, False),
]


def main():
    o = sys.stdout
    o.write("// GENERATED by tools/micro/gen_pkfma_sgpr.py -- do not edit.  See that file for what this measures.\n")
    o.write("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <vector>\n")
    meta = []
    for (name, G, ngrp, nosl, nolds) in VARIANTS:
        fma, src = kernel(name, G, ngrp, nosl, nolds)
        o.write(src)
        meta.append((name, G, ngrp, fma))
    o.write("""
typedef void (*kern_t)(const float *, float *, int, int, int, long);
struct V { const char *name; kern_t k; int G, ngrp, fma; };
static const V variants[] = {
""")
    for (name, G, ngrp, fma) in meta:
        o.write(f'    {{"{name}", {name}, {G}, {ngrp}, {fma}}},\n')
    o.write("""};
int main() {
    const long stride = 128 * 1024;                      // floats between weight streams (a stream is nit * ngrp * G floats + slack)
    float *w, *out;
    hipMalloc(&w, sizeof(float) * stride * (256 * 16 + 8) + (1 << 20));
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    std::vector<float> h(stride * 8, 1.0f / 1024.f);
    for (int i = 0; i < 256 * 16 + 8; i += 8) (void)hipMemcpy(w + (long)i * stride, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const V &v : variants) for (int waves : {4, 8, 12, 16}) for (int share : {0, 1, 2, 3}) {
        // 0: a 12 KB stream per wave, re-read (misses: the CU's waves together exceed the scalar cache); 1: one 12 KB stream per CU (hits);
        // 2: one LONG stream per CU, read once per repeat (first touch misses, the other waves hit); 3: four long streams per CU (wave & 3)
        const int nit = share >= 2 ? 256 * 192 / (v.G * v.ngrp) : 16, reps = share >= 2 ? 25 : 400;
        hipLaunchKernelGGL(v.k, dim3(256), dim3(waves * 64), 0, 0, w, out, nit, 4, share, stride);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\\n", v.name); return 1; }
        hipEventRecord(e0);
        hipLaunchKernelGGL(v.k, dim3(256), dim3(waves * 64), 0, 0, w, out, nit, reps, share, stride);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fmas = (double)v.fma * nit * reps * waves * 256;      // wave instructions
        const double tf = fmas * 64 * 4 / (ms * 1e-3) / 1e12;
        const double bpc = (double)v.G * v.ngrp * 4 * nit * reps * waves / (ms * 1e-3 * 2.4e9);   // scalar bytes per clock and CU at 2.4 GHz
        printf("%-12s waves/CU %2d  %s  %7.3f ms  %6.1f TF-equivalent  (%.1f scalar B/clk/CU, %.2f cyc per pk_fma and SIMD)\\n", v.name, waves,
               (share == 0 ? "short stream per wave" : share == 1 ? "short stream per CU  " : share == 2 ? "long stream per CU   " : "4 long streams per CU"), ms, tf, bpc, ms * 1e-3 * 2.4e9 / (fmas / 1024.0));
        fflush(stdout);
    }
    return 0;
}
""")


if __name__ == "__main__":
    main()
