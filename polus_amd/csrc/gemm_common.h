// Shared by the two MFMA GEMM kernels (gemm.hip: 128x128 general; gemm256.hip: 256x256 bf16
// direct-to-LDS): launch arguments and the fused epilogue.
#pragma once
#include "common.h"

namespace pgemm {

struct GemmArgs {
    const void* A; const void* B; void* C;
    long lda, ldb, ldc;
    int M, N, K;
    int k_per_split;  // multiple of BK
    float alpha;
    const float* bias;
    const void* resid; long ldr;
    void* aux; long ldaux;
    int act, flags;
    float* partial;  // split-K slabs [splits][M][N] or null
    int a_vec, b_vec, epi_vec;
};

template <typename TC> __device__ __forceinline__ void ld4x(const TC* p, float (&v)[4], int vec, int nvalid) {
    if (vec && nvalid >= 4) load4<TC>(p, v);
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = r < nvalid ? to_f<TC>(p[r]) : 0.0f;
    }
}
template <typename TC> __device__ __forceinline__ void st4x(TC* p, const float (&v)[4], int vec, int nvalid) {
    if (vec && nvalid >= 4) store4<TC>(p, v);
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r < nvalid) p[r] = from_f<TC>(v[r]);
    }
}


// Epilogue for one 16x16 accumulator tile issued as D[n][m]: the lane holds C[m][n..n+3]
// (m = tile row lane&15, n = 4*(lane>>4) + r).  Order: alpha, bias, ACT_FWD (aux = pre-activation),
// ACT_BWD (* act'(aux)), residual, ACCUM_C.
template <typename T, typename TC>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& p, const f32x4& acc, int m, int n, int split) {
    if (m >= p.M) return;
    const int nvalid = p.N - n;
    if (nvalid <= 0) return;
    const int ev = p.epi_vec;
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (p.partial) {
        float* dst = p.partial + ((long)split * p.M + m) * p.N + n;
        st4x<float>(dst, v, (p.N & 3) == 0, nvalid);
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
    if (p.bias) {
        float b[4]; ld4x<float>(p.bias + n, b, ev, nvalid);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += b[r];
    }
    if (p.flags & POLUS_GEMM_ACT_FWD) {
        if (p.aux) st4x<T>(static_cast<T*>(p.aux) + (long)m * p.ldaux + n, v, ev, nvalid);
        apply_act_n<4>(p.act, v);
    }
    if (p.flags & POLUS_GEMM_ACT_BWD) {
        float u[4]; ld4x<T>(static_cast<const T*>(p.aux) + (long)m * p.ldaux + n, u, ev, nvalid);
        apply_act_grad_n<4>(p.act, v, u);
    }
    if (p.resid) {
        float rr[4]; ld4x<T>(static_cast<const T*>(p.resid) + (long)m * p.ldr + n, rr, ev, nvalid);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rr[r];
    }
    TC* c = static_cast<TC*>(p.C) + (long)m * p.ldc + n;
    if (p.flags & POLUS_GEMM_ACCUM_C) {
        float o[4]; ld4x<TC>(c, o, ev, nvalid);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += o[r];
    }
    st4x<TC>(c, v, ev, nvalid);
}

// XCD-aware bijective remap: blocks b and b+8 share an XCD (and its L2); give each XCD a
// contiguous run of tiles so that neighbours reuse the same A row panel.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

}  // namespace pgemm

// gemm256.hip: bf16, both operands K-contiguous, whole 16-byte chunks.  c_is_f32 selects TC.
int polus_launch_gemm256(const pgemm::GemmArgs& a, int c_is_f32, hipStream_t st);
