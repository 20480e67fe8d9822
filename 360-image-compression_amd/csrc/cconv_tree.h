// cconv_tree.h -- pieces shared by the leaf-resident conv kernels (cconv4_kernels.hip: 4x4x1 MFMA, both orders;
// cconv16_kernels.hip: 16x16x4 MFMA, encode order): the reference's 128-leaf reduction tree on registers, compile-time loops.
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// the reference's reduction tree on registers: F(i, 128) = p[i];  F(i, s) = F(i, 2s) + F(i + s, 2s);  result F(0, 1)
// (p[i] + p[i+64] first, ..., + p[i+1] last; cconv_ec_cuda.cu:299-309).  A wave of class c evaluates F(c, 4).
// Lane i of class c = i%4 lives in accumulator i%25 (cin = 4) or i/4 (cin = 1).
template <int CIN, int I, int S>
struct Tree4 {
    static constexpr bool live = Tree4<CIN, I, S * 2>::live || Tree4<CIN, I + S, S * 2>::live;
    static __device__ __forceinline__ f32x4 eval(const f32x4 *acc) {
        if constexpr (!Tree4<CIN, I + S, S * 2>::live) return Tree4<CIN, I, S * 2>::eval(acc);     // x + 0 == x
        else if constexpr (!Tree4<CIN, I, S * 2>::live) return Tree4<CIN, I + S, S * 2>::eval(acc);
        else return Tree4<CIN, I, S * 2>::eval(acc) + Tree4<CIN, I + S, S * 2>::eval(acc);
    }
};
template <int CIN, int I>
struct Tree4<CIN, I, 128> {
    static constexpr bool live = I < 25 * CIN;
    static __device__ __forceinline__ f32x4 eval(const f32x4 *acc) {
        if constexpr (live) return acc[CIN == 4 ? I % 25 : I / 4];
        else return (f32x4){0.f, 0.f, 0.f, 0.f};
    }
};

// The tree on accumulator quads.  Its result goes through an empty asm statement as ONE 128-bit value: without that, hipcc sees
// that the callers consume it element by element and scalarises the whole tree into four per-component chains of v_add_f32; with
// it the adds stay <4 x float> and are selected as v_pk_add_f32 (two independent IEEE fp32 adds per instruction, full rate on
// gfx90a+) -- half the vector-ALU instructions of a phase in which no MFMA runs, bit-identical results, and the compiler keeps
// its own MFMA -> VALU hazard handling (an inline-asm v_pk_add_f32 would hide the reads from it).
template <int CIN, int CLS>
__device__ __forceinline__ f32x4 tree4_eval(const f32x4 *acc) {
    f32x4 r = Tree4<CIN, CLS, 4>::eval(acc);
    asm volatile("" : "+v"(r));
    return r;
}

template <int CIN> struct NAcc { static constexpr int value = CIN == 4 ? 25 : 7; };

// one K step of lane class CLS: every lane gets  acc = fma(w, x, acc)  (lanes whose chain has ended carry w = 0).
// DIAG: staged tile rows are anti-diagonals (row = kh+kw, col = kh + lane) instead of image rows (row = kh, col = kw + lane).
template <int ABID>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);   // D = fma(A[block ABID], B, C), A broadcast to all 16 blocks
}
template <int I> struct IC { static constexpr int value = I; };
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<N, F, I + 1>(static_cast<F &&>(f)); }
}

