// bf16 MFMA GEMM, 256x256 tile, direct-to-LDS staging (gfx950).  C[M,N] = A[M,K] . B[N,K]^T with
// both operands K-contiguous: the forward Dense layers and, through the transposed weight shadow,
// dX = dY . W.
//
// One 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave = 8x4 MFMA 16x16x32 tiles,
// 128 accumulator registers) per 256x256 tile, one workgroup per CU.  K-step 64.
// LDS: 2 stages x (A 256 rows x 128 B | B 256 rows x 128 B) = 128 KiB, filled by
// global_load_lds_dwordx4 (no VGPR round trip).  LDS-DMA writes lane-linear (8 rows x 128 B per
// wave-instruction), so the bank swizzle lives on the SOURCE address: LDS chunk pc of row r holds
// K-chunk pc ^ ((r >> 1) & 7); fragment reads apply the same XOR, which puts the 16 rows of every
// ds_read_b128 lane group on 16 distinct 16-byte slots.
//
// Schedule per K-step t (4 phases = the 4 quadrants of the wave tile, 16 MFMAs each).  The tile
// is split by FIRST USE into four 16 KiB groups, loaded one per phase for step t+1 into the other
// stage:
//     G1a = A rows with (row & 64) == 0   first used in phase 1      issued in phase 1 of step t
//     G1b = B rows with (row & 32) == 0   phases 1 and 4             issued in phase 2
//     G2  = B rows with (row & 32) != 0   phases 2 and 3             issued in phase 3
//     G3  = A rows with (row & 64) != 0   phases 3 and 4             issued in phase 4
// so every load has >= 2.5 phases (~1.3k cycles of MFMA time) before its first reader, and the
// only waits are counted: `s_waitcnt vmcnt(4)` (two groups stay in flight) followed by a raw
// s_barrier at the end of phases 1, 2 and 4; data is read only in a phase after the wait+barrier
// that retired it.  WAR: a group of step t+1 overwrites bytes last read one full step earlier,
// with at least one barrier in between.
#include "gemm_common.h"

using namespace pgemm;

namespace {

constexpr int TM = 256, TN = 256, TK = 64, NTHR = 512;
constexpr int REGION = 256 * 128;  // one operand, one stage
constexpr int STAGE = 2 * REGION;
constexpr int SMEM_BYTES = 2 * STAGE;

__device__ const uint4 g_zero_chunk256[1] = {{0u, 0u, 0u, 0u}};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

// first row (within the 256-row tile) of the 8-row block q (0..15) of each group
__device__ __forceinline__ int a_lo_row0(int q) { return (q >> 3) * 128 + (q & 7) * 8; }
__device__ __forceinline__ int a_hi_row0(int q) { return (q >> 3) * 128 + 64 + (q & 7) * 8; }
__device__ __forceinline__ int b_lo_row0(int q) { return (q >> 2) * 64 + (q & 3) * 8; }
__device__ __forceinline__ int b_hi_row0(int q) { return (q >> 2) * 64 + 32 + (q & 3) * 8; }

// One 1-KiB LDS-DMA piece: 8 rows x 128 B starting at tile row `row0`.
__device__ __forceinline__ void dma_piece(const bf16_t* __restrict__ base, long ld, int rows_total, int tile_r0,
                                          int row0, int k0, int kend, unsigned char* region, int lane) {
    const int row = row0 + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);
    const int gr = tile_r0 + row, gk = k0 + lc * 8;
    const bf16_t* src = (gr < rows_total && gk < kend) ? base + (long)gr * ld + gk
                                                       : reinterpret_cast<const bf16_t*>(g_zero_chunk256);
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_void_t*)(region + row0 * 128), 16, 0, 0);
}

#define POLUS_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <typename TC, int MODE = -1>
__global__ __launch_bounds__(NTHR, 2) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int wm = wid >> 2, wn = wid & 3;

    const int tiles_n = (p.N + TN - 1) / TN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * TM, n0 = (wg % tiles_n) * TN;
    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);
    const int K = p.K;
    const int nk = (K + TK - 1) / TK;

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // each wave issues pieces 2*wid and 2*wid+1 of a group
    auto issue_a_lo = [&](int st, int k0) {
        unsigned char* reg = smem + st * STAGE;
        dma_piece(A, p.lda, p.M, m0, a_lo_row0(2 * wid), k0, K, reg, lane);
        dma_piece(A, p.lda, p.M, m0, a_lo_row0(2 * wid + 1), k0, K, reg, lane);
    };
    auto issue_a_hi = [&](int st, int k0) {
        unsigned char* reg = smem + st * STAGE;
        dma_piece(A, p.lda, p.M, m0, a_hi_row0(2 * wid), k0, K, reg, lane);
        dma_piece(A, p.lda, p.M, m0, a_hi_row0(2 * wid + 1), k0, K, reg, lane);
    };
    auto issue_b_lo = [&](int st, int k0) {
        unsigned char* reg = smem + st * STAGE + REGION;
        dma_piece(B, p.ldb, p.N, n0, b_lo_row0(2 * wid), k0, K, reg, lane);
        dma_piece(B, p.ldb, p.N, n0, b_lo_row0(2 * wid + 1), k0, K, reg, lane);
    };
    auto issue_b_hi = [&](int st, int k0) {
        unsigned char* reg = smem + st * STAGE + REGION;
        dma_piece(B, p.ldb, p.N, n0, b_hi_row0(2 * wid), k0, K, reg, lane);
        dma_piece(B, p.ldb, p.N, n0, b_hi_row0(2 * wid + 1), k0, K, reg, lane);
    };

    // fragment addresses: row R = base16 + i, logical chunk sub*4 + g, physical ^ ((R>>1)&7) = ^ (i>>1)
    const int swz = (i >> 1) & 7;
    const int a_off = (wm * 128 + i) * 128;            // + mt*2048
    const int b_off = REGION + (wn * 64 + i) * 128;    // + nt*2048
    const int c0 = ((0 + g) ^ swz) * 16, c1 = ((4 + g) ^ swz) * 16;
    auto lda_frag = [&](Frag<bf16_t> (&f)[2], const unsigned char* stage, int mt) {
        f[0].v = *reinterpret_cast<const bf16x8*>(stage + a_off + mt * 2048 + c0);
        f[1].v = *reinterpret_cast<const bf16x8*>(stage + a_off + mt * 2048 + c1);
    };
    auto ldb_frag = [&](Frag<bf16_t> (&f)[2], const unsigned char* stage, int nt) {
        f[0].v = *reinterpret_cast<const bf16x8*>(stage + b_off + nt * 2048 + c0);
        f[1].v = *reinterpret_cast<const bf16x8*>(stage + b_off + nt * 2048 + c1);
    };

    // prologue: whole first tile; G1 must have landed before phase 1
    issue_a_lo(0, 0); issue_b_lo(0, 0); issue_b_hi(0, 0); issue_a_hi(0, 0);
    POLUS_VMCNT(4);
    __builtin_amdgcn_s_barrier();

    for (int t = 0; t < nk; ++t) {
        const unsigned char* st = smem + (t & 1) * STAGE;
        const int nst = (t + 1) & 1, nk0 = (t + 1) * TK;
        const bool more = (t + 1 < nk) && !(p.ablate & 1);
        const bool nomma = p.ablate & 2;
        Frag<bf16_t> af[4][2], bfr[2][2];

        // ---- phase 1: quadrant (m 0..3, n 0..1)
        if (more) issue_a_lo(nst, nk0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) ldb_frag(bfr[nt], st, nt);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) lda_frag(af[mt], st, mt);
        if (!nomma) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                mma16(acc[mt][nt], bfr[nt][0], af[mt][0]);
                mma16(acc[mt][nt], bfr[nt][1], af[mt][1]);
            }
        __builtin_amdgcn_s_setprio(0);
        } else { asm volatile("" :: "v"(af[0][0].v), "v"(af[1][0].v), "v"(af[2][0].v), "v"(af[3][0].v), "v"(af[0][1].v), "v"(af[1][1].v), "v"(af[2][1].v), "v"(af[3][1].v), "v"(bfr[0][0].v), "v"(bfr[1][0].v), "v"(bfr[0][1].v), "v"(bfr[1][1].v)); }
        if (more) POLUS_VMCNT(4); else POLUS_VMCNT(2);   // retires G2(t) = B hi
        __builtin_amdgcn_s_barrier();

        // ---- phase 2: quadrant (m 0..3, n 2..3)
        if (more) issue_b_lo(nst, nk0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) ldb_frag(bfr[nt], st, 2 + nt);
        if (!nomma) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                mma16(acc[mt][2 + nt], bfr[nt][0], af[mt][0]);
                mma16(acc[mt][2 + nt], bfr[nt][1], af[mt][1]);
            }
        __builtin_amdgcn_s_setprio(0);
        } else { asm volatile("" :: "v"(af[0][0].v), "v"(af[1][0].v), "v"(af[2][0].v), "v"(af[3][0].v), "v"(af[0][1].v), "v"(af[1][1].v), "v"(af[2][1].v), "v"(af[3][1].v), "v"(bfr[0][0].v), "v"(bfr[1][0].v), "v"(bfr[0][1].v), "v"(bfr[1][1].v)); }
        if (more) POLUS_VMCNT(4); else POLUS_VMCNT(0);   // retires G3(t) = A hi
        __builtin_amdgcn_s_barrier();

        // ---- phase 3: quadrant (m 4..7, n 2..3)
        if (more) issue_b_hi(nst, nk0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) lda_frag(af[mt], st, 4 + mt);
        if (!nomma) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                mma16(acc[4 + mt][2 + nt], bfr[nt][0], af[mt][0]);
                mma16(acc[4 + mt][2 + nt], bfr[nt][1], af[mt][1]);
            }
        __builtin_amdgcn_s_setprio(0);
        } else { asm volatile("" :: "v"(af[0][0].v), "v"(af[1][0].v), "v"(af[2][0].v), "v"(af[3][0].v), "v"(af[0][1].v), "v"(af[1][1].v), "v"(af[2][1].v), "v"(af[3][1].v), "v"(bfr[0][0].v), "v"(bfr[1][0].v), "v"(bfr[0][1].v), "v"(bfr[1][1].v)); }

        // ---- phase 4: quadrant (m 4..7, n 0..1)
        if (more) issue_a_hi(nst, nk0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) ldb_frag(bfr[nt], st, nt);
        if (!nomma) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                mma16(acc[4 + mt][nt], bfr[nt][0], af[mt][0]);
                mma16(acc[4 + mt][nt], bfr[nt][1], af[mt][1]);
            }
        __builtin_amdgcn_s_setprio(0);
        } else { asm volatile("" :: "v"(af[0][0].v), "v"(af[1][0].v), "v"(af[2][0].v), "v"(af[3][0].v), "v"(af[0][1].v), "v"(af[1][1].v), "v"(af[2][1].v), "v"(af[3][1].v), "v"(bfr[0][0].v), "v"(bfr[1][0].v), "v"(bfr[0][1].v), "v"(bfr[1][1].v)); }
        if (more) {
            POLUS_VMCNT(4);                              // retires G1(t+1) = A lo, B lo
            __builtin_amdgcn_s_barrier();
        } else if (t + 1 < nk) {
            __builtin_amdgcn_s_barrier();                // ablation build only
        }
    }

    // every wave is done with the operand stages (and no DMA is in flight): reuse LDS for the C staging
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (MODE >= 0) epilogue_wave<TC, 64, false, MODE < 0 ? 0 : MODE, false>(p, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wid * 8704);
    else epilogue_wave_128x64_lds<TC, false>(p, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wid * 8704);
}

template <typename TC, int MODE = -1>
int launch256(const GemmArgs& a, hipStream_t st) {
    static bool attr_done = false;
    auto kern = gemm256_kernel<TC, MODE>;
    if (!attr_done) {
        POLUS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
        attr_done = true;
    }
    const int tiles = ((a.M + TM - 1) / TM) * ((a.N + TN - 1) / TN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHR), SMEM_BYTES, st, a);
    POLUS_CHECK_LAUNCH("polus_gemm(256x256)");
    return POLUS_OK;
}

}  // namespace

int polus_launch_gemm256(const GemmArgs& a, int c_is_f32, hipStream_t st) {
    if (!c_is_f32) {
        switch (polus_gemm_p_mode(a, 0, 0)) {
            case 0: return launch256<bf16_t, 0>(a, st);
            case 1: return launch256<bf16_t, 1>(a, st);
            case 2: return launch256<bf16_t, 2>(a, st);
            case 3: return launch256<bf16_t, 3>(a, st);
            default: break;
        }
    }
    return c_is_f32 ? launch256<float>(a, st) : launch256<bf16_t>(a, st);
}
